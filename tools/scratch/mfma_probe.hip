// Machine parameters behind the resident sweep's schedule (round 6): what ONE wave per SIMD gets out of the fp64 pipe.
//   (a) v_mfma_f64_16x16x4_f64 with 1..6 independent accumulators, 1 or 2 waves per SIMD: cycles per MFMA (s_memtime);
//   (b) the same with the A operand read from LDS one step ahead (the contraction of k_sweep_r8);
//   (c) a chain of DEPENDENT v_fma_f64 (latency) and 2 / 4 / 8 independent chains (issue rate).
// Build: hipcc --offload-arch=gfx950 -O3 tools/scratch/mfma_probe.hip -o tools/scratch/bin/mfma_probe ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef double d4 __attribute__((ext_vector_type(4)));

template <int NACC, bool LDSA>
__global__ void k_mfma(double* out, unsigned long long* cyc, int iters) {
  extern __shared__ double lds[];
  for (int k = threadIdx.x; k < 4096; k += blockDim.x) lds[k] = 1e-3 * k;
  __syncthreads();
  d4 acc[NACC];
  for (int i = 0; i < NACC; ++i) acc[i] = (d4){0, 0, 0, 0};
  double a[NACC], b = blockIdx.x * 1e-3 + 1.0;
  for (int i = 0; i < NACC; ++i) a[i] = threadIdx.x * 1e-3 + i;
  const int lane = threadIdx.x & 63;
  const unsigned long long t0 = __builtin_readcyclecounter();
  for (int it = 0; it < iters; ++it) {
    double an[NACC];
    if (LDSA) {
#pragma unroll
      for (int i = 0; i < NACC; ++i) an[i] = lds[((it + 1) * 32 + i * 512 + lane) & 4095];
    }
#pragma unroll
    for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[i], b, acc[i], 0, 0, 0);
    if (LDSA) {
#pragma unroll
      for (int i = 0; i < NACC; ++i) a[i] = an[i];
    }
  }
  const unsigned long long t1 = __builtin_readcyclecounter();
  double s = 0;
  for (int i = 0; i < NACC; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}
template <int NCH>
__global__ void k_fma(double* out, unsigned long long* cyc, int iters) {
  double a[NCH], x = 1.0 + 1e-9 * threadIdx.x, y = 1e-9;
  for (int i = 0; i < NCH; ++i) a[i] = i;
  const unsigned long long t0 = __builtin_readcyclecounter();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int r = 0; r < 8; ++r)
#pragma unroll
      for (int i = 0; i < NCH; ++i) a[i] = fma(a[i], x, y);
  }
  const unsigned long long t1 = __builtin_readcyclecounter();
  double s = 0;
  for (int i = 0; i < NCH; ++i) s += a[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}
template <class K>
static void run(const char* name, K kern, int threads, size_t lds, int iters, double per_iter, double* out, unsigned long long* cyc) {
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL(kern, dim3(256), dim3(threads), lds, 0, out, cyc, iters);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  hipLaunchKernelGGL(kern, dim3(256), dim3(threads), lds, 0, out, cyc, iters);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  std::vector<unsigned long long> h(256);
  hipMemcpy(h.data(), cyc, 256 * 8, hipMemcpyDeviceToHost);
  double c = 0; for (auto v : h) c += (double)v; c /= 256;
  printf("%-44s %8.3f ms  %10.0f counter ticks  -> %.2f ns per op per wave, %.2f ticks\n", name, ms, c, 1e6 * ms / (iters * per_iter), c / (iters * per_iter));
}
int main() {
  double* out; unsigned long long* cyc;
  hipMalloc(&out, 256 * 1024 * 8); hipMalloc(&cyc, 256 * 8);
  const int it = 20000;
  const size_t big = 100 * 1024;  // one block per CU
#define M(N, L, T, S, nm) hipFuncSetAttribute((const void*)k_mfma<N, L>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024); run(nm, k_mfma<N, L>, T, S, it, N, out, cyc)
  M(1, false, 256, big, "mfma 1 acc, 1 wave/SIMD");
  M(2, false, 256, big, "mfma 2 acc, 1 wave/SIMD");
  M(3, false, 256, big, "mfma 3 acc, 1 wave/SIMD");
  M(4, false, 256, big, "mfma 4 acc, 1 wave/SIMD");
  M(6, false, 256, big, "mfma 6 acc, 1 wave/SIMD");
  M(3, true, 256, big, "mfma 3 acc + LDS A, 1 wave/SIMD");
  M(3, false, 512, big, "mfma 3 acc, 2 waves/SIMD (one block)");
  M(3, true, 512, big, "mfma 3 acc + LDS A, 2 waves/SIMD");
  M(1, false, 512, big, "mfma 1 acc, 2 waves/SIMD");
#define F(N, T, nm) run(nm, k_fma<N>, T, big, it, 8.0 * N, out, cyc)
  F(1, 256, "fma 1 chain, 1 wave/SIMD");
  F(2, 256, "fma 2 chains, 1 wave/SIMD");
  F(4, 256, "fma 4 chains, 1 wave/SIMD");
  F(8, 256, "fma 8 chains, 1 wave/SIMD");
  F(1, 512, "fma 1 chain, 2 waves/SIMD");
  F(2, 512, "fma 2 chains, 2 waves/SIMD");
  F(4, 512, "fma 4 chains, 2 waves/SIMD");
  return 0;
}
