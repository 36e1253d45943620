// Round-5 probe for VERDICT r4 item 1: can the streaming half of the step (k_flush_lw's read-modify-write of the inverse planes) be
// made independent of the wave slots it holds with gfx950's LDS-DMA loads (global_load_lds_dwordx4), and does it then run beside a
// kernel that is bound by the fp64 pipe at 2 waves / SIMD (k_orb's situation) without taking its time?
//   A  k_rmw_rows      k_flush_lw's mapping, plain loads into registers (tools/scratch/layout_probe.hip), at its natural occupancy and
//                      throttled to 2 / 1 blocks per CU by a dynamic-LDS reservation
//   B  k_rmw_dma       one loader wave per block issues global_load_lds_dwordx4 into a ring of 16-KB LDS slots (16 walkers x 4 rows of
//                      the 32 x 32 inverse), four consumer waves read a slot from LDS, update, and store to global memory
//   C  k_pipe          fp64 FMA + MFMA loop, 256 threads, 120 VGPRs + 58 KB LDS (two blocks per CU): alone, beside A, beside B
// hipcc --offload-arch=gfx950 -O3 tools/scratch/dma_probe.hip -o /tmp/dma_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef double d4 __attribute__((ext_vector_type(4)));
#define CHK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

__global__ __launch_bounds__(256) void k_rmw_rows(double* __restrict__ T, long W) {
  extern __shared__ double pad_[];
  const int wl = threadIdx.x & 15, rg = threadIdx.x >> 4;
  const long w = (long)blockIdx.x * 16 + wl;
  double* base = T + w;
  for (int i = rg; i < 32; i += 16) {
    double t[32];
#pragma unroll
    for (int k = 0; k < 32; ++k) t[k] = base[(long)(i * 32 + k) * W];
#pragma unroll
    for (int k = 0; k < 32; ++k) t[k] = t[k] * 1.0000001 + 1.0;
#pragma unroll
    for (int k = 0; k < 32; ++k) base[(long)(i * 32 + k) * W] = t[k];
  }
  if (pad_[0] == 1.2345e300) T[0] = 0.0;
}

// ring of NS slots; slot = 4 rows x 32 columns x 16 walkers = 128 planes x 128 B = 16 KB = 16 wave-instructions of 1 KB
// persistent: block b walks walker groups b, b + gridDim.x, ...; items of a group = 8 (rows 4 q .. 4 q + 3)
template <int NS>
__global__ __launch_bounds__(320) void k_rmw_dma(double* __restrict__ T, long W, long ngroups) {
  extern __shared__ double ring[];  // [NS][128][16]
  const int tid = threadIdx.x, wv = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;
  const long nitem = ((ngroups - blockIdx.x + gridDim.x - 1) / gridDim.x) * 8;  // items of this block
  auto item_base = [&](long it) -> double* {  // first plane of the item, walker 0 of the group
    const long g = blockIdx.x + (it >> 3) * gridDim.x;
    return T + (long)((it & 7) * 128) * W + g * 16;
  };
  if (wv == 4) {  // loader: lane -> (plane within an 8-plane piece, 16-byte pair of walkers)
    const int pl = lane >> 3, pr = lane & 7;
    auto issue = [&](long it) {
      const double* src = item_base(it) + (long)pl * W + 2 * pr;
      double* dst = ring + (size_t)(it % NS) * 2048;
#pragma unroll
      for (int piece = 0; piece < 16; ++piece)
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src + (long)piece * 8 * W),
                                         (__attribute__((address_space(3))) void*)(dst + piece * 128), 16, 0, 0);
    };
    for (long it = 0; it < NS - 1 && it < nitem; ++it) issue(it);
    for (long it = 0; it < nitem; ++it) {
      // slot it % NS must have landed: at most NS - 2 later items (16 loads each) may still be in flight
      const long ahead = (nitem - 1 - it < NS - 2) ? nitem - 1 - it : NS - 2;
      if (ahead >= 2) __builtin_amdgcn_s_waitcnt(0x0f70 | 32 & 0xf | ((32 >> 4) & 3) << 14);  // vmcnt(32)
      else if (ahead == 1) __builtin_amdgcn_s_waitcnt(0x0f70 | (16 & 0xf) | ((16 >> 4) & 3) << 14);  // vmcnt(16)
      else __builtin_amdgcn_s_waitcnt(0x0f70);  // vmcnt(0)
      __builtin_amdgcn_s_barrier();             // consumers may read slot it % NS; they have finished slot (it - 1) % NS
      if (it + NS - 1 < nitem) issue(it + NS - 1);
    }
  } else {  // consumers: wave = row of the slot, lane = (walker, column quarter)
    const int wl = lane & 15, q = lane >> 4;
    for (long it = 0; it < nitem; ++it) {
      __builtin_amdgcn_s_barrier();
      const double* s = ring + (size_t)(it % NS) * 2048 + (size_t)(wv * 32 + q * 8) * 16 + wl;
      double t[8];
#pragma unroll
      for (int k = 0; k < 8; ++k) t[k] = s[k * 16];
#pragma unroll
      for (int k = 0; k < 8; ++k) t[k] = t[k] * 1.0000001 + 1.0;
      double* o = item_base(it) + (long)(wv * 32 + q * 8) * W + wl;
#pragma unroll
      for (int k = 0; k < 8; ++k) o[(long)k * W] = t[k];
    }
  }
}

__global__ __launch_bounds__(256, 2) void k_pipe(double* out, int iters) {
  extern __shared__ double lp[];
  d4 acc[12];
  double a[24];
  for (int i = 0; i < 12; ++i) acc[i] = (d4){0, 0, 0, 0};
  for (int i = 0; i < 24; ++i) a[i] = i + threadIdx.x * 1e-3;
  double x = 1.0 + 1e-9 * threadIdx.x, y = 1e-9, b = blockIdx.x * 1e-3 + 1.0;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < 12; ++i) acc[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[i], b, acc[i], 0, 0, 0);
#pragma unroll
    for (int i = 0; i < 24; ++i) a[i] = fma(a[i], x, y);
  }
  double s = lp[threadIdx.x & 7] * 0.0;
  for (int i = 0; i < 12; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
  for (int i = 0; i < 24; ++i) s += a[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

static float timed(hipStream_t st, int reps, auto&& launch) {
  hipEvent_t a, b;
  (void)hipEventCreate(&a); (void)hipEventCreate(&b);
  float best = 1e9f;
  for (int r = 0; r < reps; ++r) {
    (void)hipEventRecord(a, st); launch(); (void)hipEventRecord(b, st); (void)hipEventSynchronize(b);
    float ms; (void)hipEventElapsedTime(&ms, a, b);
    if (ms < best) best = ms;
  }
  return best;
}

int main() {
  const long W = 65536, ng = W / 16;
  double *T, *out;
  CHK(hipMalloc(&T, 1024 * W * 8)); CHK(hipMalloc(&out, 1 << 24)); CHK(hipMemset(T, 0, 1024 * W * 8));
  hipStream_t s1, s2;
  CHK(hipStreamCreate(&s1)); CHK(hipStreamCreate(&s2));
  const double bytes = 2.0 * 1024 * W * 8;
  CHK(hipFuncSetAttribute((const void*)k_rmw_rows, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
  CHK(hipFuncSetAttribute((const void*)k_rmw_dma<4>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
  CHK(hipFuncSetAttribute((const void*)k_rmw_dma<8>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
  CHK(hipFuncSetAttribute((const void*)k_pipe, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
  printf("== A: plain loads, k_flush_lw's mapping (2 x %.0f MB)\n", bytes / 2e6);
  for (int lds : {0, 40 * 1024, 80 * 1024, 159 * 1024}) {
    float ms = timed(s1, 5, [&] { hipLaunchKernelGGL(k_rmw_rows, dim3(ng), dim3(256), lds, s1, T, W); });
    printf("dynamic LDS %6d B (blocks per CU <= %s): %7.1f us %6.0f GB/s\n", lds, lds == 0 ? "regs" : (lds < 50000 ? "4" : (lds < 90000 ? "2" : "1")), ms * 1e3, bytes / ms * 1e-6);
  }
  printf("== B: LDS-DMA loader wave + 4 consumer waves, persistent blocks\n");
  for (int bpc : {1, 2}) {
    {
      float ms = timed(s1, 5, [&] { hipLaunchKernelGGL(k_rmw_dma<4>, dim3(256 * bpc), dim3(320), 4 * 16384, s1, T, W, ng); });
      printf("ring 4 x 16 KB, %d block(s) per CU: %7.1f us %6.0f GB/s\n", bpc, ms * 1e3, bytes / ms * 1e-6);
    }
    if (bpc == 1) {
      float ms = timed(s1, 5, [&] { hipLaunchKernelGGL(k_rmw_dma<8>, dim3(256 * bpc), dim3(320), 8 * 16384, s1, T, W, ng); });
      printf("ring 8 x 16 KB, %d block(s) per CU: %7.1f us %6.0f GB/s\n", bpc, ms * 1e3, bytes / ms * 1e-6);
    }
  }
  printf("== C: fp64-pipe kernel (2 blocks of 256 threads per CU, 58 KB LDS each) alone and beside the streams\n");
  const int iters = 1400;
  auto pipe = [&] { hipLaunchKernelGGL(k_pipe, dim3(512 * 4), dim3(256), 58 * 1024, s2, out, iters); };
  float p0 = timed(s2, 5, pipe);
  printf("k_pipe alone: %7.1f us\n", p0 * 1e3);
  struct V { const char* name; int kind; int lds; int bpc; };
  for (V v : {V{"A plain, natural occupancy", 0, 0, 0}, V{"A plain, <= 1 block per CU", 0, 159 * 1024 - 58 * 2048, 0}, V{"B DMA ring 4, 1 block per CU", 1, 4 * 16384, 1}, V{"B DMA ring 2.. (32 KB), 1 block per CU", 2, 0, 1}}) {
    // both on their own streams, started together; each timed by its own events
    hipEvent_t a1, b1, a2, b2;
    (void)hipEventCreate(&a1); (void)hipEventCreate(&b1); (void)hipEventCreate(&a2); (void)hipEventCreate(&b2);
    float best1 = 1e9f, best2 = 1e9f;
    for (int r = 0; r < 4; ++r) {
      (void)hipDeviceSynchronize();
      (void)hipEventRecord(a2, s2); pipe(); (void)hipEventRecord(b2, s2);
      (void)hipEventRecord(a1, s1);
      for (int rep = 0; rep < 2; ++rep) {
        if (v.kind == 0) hipLaunchKernelGGL(k_rmw_rows, dim3(ng), dim3(256), v.lds, s1, T, W);
        else if (v.kind == 1) hipLaunchKernelGGL(k_rmw_dma<4>, dim3(256 * v.bpc), dim3(320), v.lds, s1, T, W, ng);
        else hipLaunchKernelGGL(k_rmw_dma<4>, dim3(256 * v.bpc), dim3(320), 4 * 16384, s1, T, W, ng);
      }
      (void)hipEventRecord(b1, s1);
      (void)hipDeviceSynchronize();
      float m1, m2; (void)hipEventElapsedTime(&m1, a1, b1); (void)hipEventElapsedTime(&m2, a2, b2);
      if (m1 + m2 < best1 + best2) { best1 = m1; best2 = m2; }
    }
    printf("%-40s: 2 x stream %7.1f us (%6.0f GB/s each)   k_pipe %7.1f us (%.2fx alone)\n", v.name, best1 * 1e3, 2 * bytes / best1 * 1e-6, best2 * 1e3, best2 / p0);
  }
  return 0;
}
