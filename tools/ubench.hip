// Micro-benchmarks for the two ceilings of the orbital kernel on gfx950:
//   (a) v_mfma_f64_16x16x4_f64 issue rate (fp64 matrix peak), (b) fp64 exp() throughput (ocml).
// Build: hipcc --offload-arch=gfx950 -O3 tools/ubench.hip -o tools/ubench ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double d4 __attribute__((ext_vector_type(4)));

template <int NACC>
__global__ __launch_bounds__(256) void k_mfma(double* out, int iters) {
  d4 acc[NACC];
  for (int i = 0; i < NACC; ++i) acc[i] = (d4){0, 0, 0, 0};
  double a = threadIdx.x * 1e-3, b = blockIdx.x * 1e-3 + 1.0;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[i], 0, 0, 0);
  }
  double s = 0;
  for (int i = 0; i < NACC; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

__global__ __launch_bounds__(256) void k_exp(double* out, int iters) {
  double x = -1e-3 * (threadIdx.x + 1), s = 0;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < 8; ++i) s += exp(x * (it + i + 1));
  }
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

__global__ __launch_bounds__(256) void k_fma(double* out, int iters) {
  double a[8], x = 1.0 + 1e-9 * threadIdx.x, y = 1e-9;
  for (int i = 0; i < 8; ++i) a[i] = i;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < 8; ++i) a[i] = fma(a[i], x, y);
  }
  double s = 0;
  for (int i = 0; i < 8; ++i) s += a[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

// Co-execution test: blocks of 8 waves (2 per SIMD).  mode 0: waves 0-3 MFMA, 4-7 idle; mode 1: waves 0-3 idle, 4-7 FMA;
// mode 2: waves 0-3 MFMA and 4-7 FMA at the same time; mode 3: every wave alternates MFMA and FMA bursts itself.
__global__ __launch_bounds__(512) void k_coexec(double* out, int iters, int mode) {
  const int wv = threadIdx.x >> 6;
  d4 acc[8];
  for (int i = 0; i < 8; ++i) acc[i] = (d4){0, 0, 0, 0};
  double f[8], x = 1.0 + 1e-9 * threadIdx.x, y = 1e-9, a = threadIdx.x * 1e-3, b = blockIdx.x * 1e-3 + 1.0;
  for (int i = 0; i < 8; ++i) f[i] = i;
  const bool do_mfma = (mode == 0 || mode == 2) ? wv < 4 : (mode == 3);
  const bool do_fma = (mode == 1 || mode == 2) ? wv >= 4 : (mode == 3);
  for (int it = 0; it < iters; ++it) {
    if (do_mfma) {
#pragma unroll
      for (int i = 0; i < 8; ++i) acc[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[i], 0, 0, 0);
    }
    if (do_fma) {
#pragma unroll
      for (int r = 0; r < 16; ++r)
#pragma unroll
        for (int i = 0; i < 8; ++i) f[i] = fma(f[i], x, y);
    }
  }
  double s = 0;
  for (int i = 0; i < 8; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3] + f[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <class F>
static float timeit(F f) {
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  f();
  hipDeviceSynchronize();
  hipEventRecord(e0);
  f();
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  return ms;
}

int main() {
  double* out; hipMalloc(&out, 4096 * 256 * sizeof(double));
  const int iters = 4000;
  for (int blocks : {256, 512, 1024, 2048}) {
    float ms = timeit([&] { hipLaunchKernelGGL(k_mfma<4>, dim3(blocks), dim3(256), 0, 0, out, iters); });
    double flops = (double)blocks * 4 /*waves*/ * iters * 4 * 2048.0;
    printf("mfma_f64_16x16x4 x4acc blocks=%d: %.3f ms  %.2f TFLOP/s\n", blocks, ms, flops / ms / 1e9);
  }
  {
    float ms = timeit([&] { hipLaunchKernelGGL(k_mfma<8>, dim3(1024), dim3(256), 0, 0, out, iters); });
    printf("mfma_f64_16x16x4 x8acc blocks=1024: %.3f ms  %.2f TFLOP/s\n", ms, 1024.0 * 4 * iters * 8 * 2048.0 / ms / 1e9);
  }
  for (int blocks : {1024, 4096}) {
    float ms = timeit([&] { hipLaunchKernelGGL(k_exp, dim3(blocks), dim3(256), 0, 0, out, 500); });
    printf("exp f64 blocks=%d: %.3f ms  %.1f Gexp/s\n", blocks, ms, (double)blocks * 256 * 500 * 8 / ms / 1e6);
  }
  {
    float ms = timeit([&] { hipLaunchKernelGGL(k_fma, dim3(2048), dim3(256), 0, 0, out, 4000); });
    printf("fma f64 blocks=2048: %.3f ms  %.2f TFLOP/s\n", ms, 2048.0 * 256 * 4000 * 8 * 2 / ms / 1e9);
  }
  for (int mode = 0; mode < 4; ++mode) {
    const int it2 = 2000;
    float ms = timeit([&] { hipLaunchKernelGGL(k_coexec, dim3(256), dim3(512), 0, 0, out, it2, mode); });
    const double nm = (mode == 3 ? 8.0 : 4.0) * 256 * it2 * 8 * 2048.0;          // MFMA flops when active
    const double nf = (mode == 3 ? 8.0 : 4.0) * 256 * 64 * it2 * 16.0 * 8 * 2;   // FMA flops when active
    const double fl = (mode == 0 ? nm : mode == 1 ? nf : nm + nf);
    printf("coexec mode %d (0 mfma only, 1 fma only, 2 mfma||fma on different waves, 3 both in every wave): %.3f ms  %.2f TFLOP/s total\n",
           mode, ms, fl / ms / 1e9);
  }
  return 0;
}
