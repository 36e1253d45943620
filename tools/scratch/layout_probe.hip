// Does the walker-fastest plane layout [n][n][W] of the inverse cost HBM efficiency against walker tiles [W/16][n][n][16]?
// Read-modify-write of a 32 x 32 x W array with k_flush_lw's thread mapping (block = 16 walkers x 16 row groups, a row of 32 doubles in
// registers per pass) in both layouts; and a read pass with k_step_lw's mapping (64 walkers x 4 groups, 8 slots of ONE row per thread).
// hipcc --offload-arch=gfx950 -O3 tools/scratch/layout_probe.hip -o /tmp/layout_probe
#include <hip/hip_runtime.h>
#include <cstdio>
template <int TS>
__global__ __launch_bounds__(256) void k_rmw_rows(double* __restrict__ T, long W) {
  const int wl = threadIdx.x & 15, rg = threadIdx.x >> 4;
  const long b = blockIdx.x, w = b * 16 + wl;
  double* base = TS ? T + (w / TS) * 1024 * TS + (w % TS) : T + w;
  const long str = TS ? TS : W;
  for (int i = rg; i < 32; i += 16) {
    double t[32];
#pragma unroll
    for (int k = 0; k < 32; ++k) t[k] = base[(long)(i * 32 + k) * str];
#pragma unroll
    for (int k = 0; k < 32; ++k) t[k] = t[k] * 1.0000001 + 1.0;
#pragma unroll
    for (int k = 0; k < 32; ++k) base[(long)(i * 32 + k) * str] = t[k];
  }
}
// TS = walkers per tile (0: planes)
template <int TS>
__global__ __launch_bounds__(256) void k_read_row(const double* __restrict__ T, long W, int i, double* out) {
  const int lane = threadIdx.x & 63, g = threadIdx.x >> 6;
  const long w = (long)blockIdx.x * 64 + lane;
  double s = 0;
#pragma unroll
  for (int k = 8 * g; k < 8 * g + 8; ++k) s += TS ? T[((w / TS) * 1024 + i * 32 + k) * TS + (w % TS)] : T[(long)(i * 32 + k) * W + w];
  if (s == 1.2345e300) out[0] = s;
}
#define TM(name, launch, bytes) do { hipEvent_t a, b; (void)hipEventCreate(&a); (void)hipEventCreate(&b); float best = 1e9; \
  for (int r = 0; r < 5; ++r) { (void)hipEventRecord(a); launch; (void)hipEventRecord(b); (void)hipEventSynchronize(b); float ms; (void)hipEventElapsedTime(&ms, a, b); if (ms < best) best = ms; } \
  printf("%-44s %8.1f us %6.0f GB/s\n", name, best * 1e3, (bytes) / best * 1e-6); } while (0)
int main() {
  for (long W : {65536L, 16384L}) {
    double *T, *out;
    (void)hipMalloc(&T, 1024 * W * 8); (void)hipMalloc(&out, 64); (void)hipMemset(T, 0, 1024 * W * 8);
    printf("W = %ld\n", W);
    TM("flush mapping, planes [n][n][W]", hipLaunchKernelGGL(k_rmw_rows<0>, dim3(W / 16), dim3(256), 0, 0, T, W), 2.0 * 1024 * W * 8);
    TM("flush mapping, tiles [W/16][n][n][16]", hipLaunchKernelGGL(k_rmw_rows<16>, dim3(W / 16), dim3(256), 0, 0, T, W), 2.0 * 1024 * W * 8);
    TM("flush mapping, tiles [W/64][n][n][64]", hipLaunchKernelGGL(k_rmw_rows<64>, dim3(W / 16), dim3(256), 0, 0, T, W), 2.0 * 1024 * W * 8);
    TM("flush mapping, tiles [W/256][n][n][256]", hipLaunchKernelGGL(k_rmw_rows<256>, dim3(W / 16), dim3(256), 0, 0, T, W), 2.0 * 1024 * W * 8);
    // 32 launches, one row each (what 32 moves of a spin read of the inverse): 256 B per walker and launch
    TM("step mapping x32 rows, planes", for (int i = 0; i < 32; ++i) hipLaunchKernelGGL(k_read_row<0>, dim3(W / 64), dim3(256), 0, 0, T, W, i, out), 1024.0 * W * 8);
    TM("step mapping x32 rows, tiles of 16", for (int i = 0; i < 32; ++i) hipLaunchKernelGGL(k_read_row<16>, dim3(W / 64), dim3(256), 0, 0, T, W, i, out), 1024.0 * W * 8);
    TM("step mapping x32 rows, tiles of 64", for (int i = 0; i < 32; ++i) hipLaunchKernelGGL(k_read_row<64>, dim3(W / 64), dim3(256), 0, 0, T, W, i, out), 1024.0 * W * 8);
    (void)hipFree(T); (void)hipFree(out);
  }
  return 0;
}
