// Calibration of rocprofv3's FETCH_SIZE / WRITE_SIZE on gfx950 for THIS code's access patterns
// (/opt/skills/guides/MI355X_MICROARCH.md "HBM": FETCH_SIZE reports 1/2 of a wide 16-B/lane streaming read; other widths
// and WRITE_SIZE are uncalibrated — "calibrate on a known byte count in your own access pattern").
// Each kernel moves a KNOWN number of bytes over a 2 GiB buffer (8x the 256 MiB Infinity Cache):
//   k_read8   8 B/lane coalesced loads  (the lane-per-walker kernels: one double per lane, walker index fastest)
//   k_read16  16 B/lane coalesced loads (the guide's calibrated pattern, factor 2 expected)
//   k_write8  8 B/lane coalesced stores
//   k_rmw8    read + write of the same 8 B/lane (the Sherman-Morrison flush)
// tools/refresh_evidence.sh runs this under `rocprofv3 --pmc FETCH_SIZE` and `--pmc WRITE_SIZE` (separate passes) and
// tools/pmc_summary.py turns counter / known bytes into the correction factors it applies to the bench's kernels.
// Build: hipcc --offload-arch=gfx950 -O3 tools/pmc_calib.hip -o tools/pmc_calib
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double d2 __attribute__((ext_vector_type(2)));

__global__ __launch_bounds__(256) void k_read8(const double* __restrict__ in, size_t n, double* __restrict__ out) {
  double s = 0.0;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) s += in[i];
  if (s == 1.2345e300) out[0] = s;
}
__global__ __launch_bounds__(256) void k_read16(const d2* __restrict__ in, size_t n2, double* __restrict__ out) {
  double s = 0.0;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n2; i += (size_t)gridDim.x * 256) { const d2 v = in[i]; s += v.x + v.y; }
  if (s == 1.2345e300) out[0] = s;
}
__global__ __launch_bounds__(256) void k_write8(double* __restrict__ out, size_t n) {
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) out[i] = (double)i;
}
__global__ __launch_bounds__(256) void k_rmw8(double* __restrict__ io, size_t n) {
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) io[i] = io[i] * 1.0000001 + 1.0;
}

int main() {
  const size_t n = (size_t)1 << 28;  // doubles: 2 GiB
  double *buf, *out;
  if (hipMalloc(&buf, n * sizeof(double)) != hipSuccess || hipMalloc(&out, 64) != hipSuccess) { printf("alloc failed\n"); return 1; }
  (void)hipMemset(buf, 0, n * sizeof(double));
  const dim3 g(256 * 32), b(256);
  hipEvent_t e0, e1;
  (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  float ms;
  for (int rep = 0; rep < 2; ++rep) {
    (void)hipEventRecord(e0); hipLaunchKernelGGL(k_read8, g, b, 0, 0, (const double*)buf, n, out); (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    (void)hipEventElapsedTime(&ms, e0, e1); printf("k_read8  %.3f ms  %.0f GB/s (known %zu B)\n", ms, n * 8 / ms * 1e-6, n * 8);
    (void)hipEventRecord(e0); hipLaunchKernelGGL(k_read16, g, b, 0, 0, (const d2*)buf, n / 2, out); (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    (void)hipEventElapsedTime(&ms, e0, e1); printf("k_read16 %.3f ms  %.0f GB/s\n", ms, n * 8 / ms * 1e-6);
    (void)hipEventRecord(e0); hipLaunchKernelGGL(k_write8, g, b, 0, 0, buf, n); (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    (void)hipEventElapsedTime(&ms, e0, e1); printf("k_write8 %.3f ms  %.0f GB/s\n", ms, n * 8 / ms * 1e-6);
    (void)hipEventRecord(e0); hipLaunchKernelGGL(k_rmw8, g, b, 0, 0, buf, n); (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    (void)hipEventElapsedTime(&ms, e0, e1); printf("k_rmw8   %.3f ms  %.0f GB/s (read + write)\n", ms, 2 * n * 8 / ms * 1e-6);
  }
  printf("KNOWN_BYTES %zu\n", n * 8);
  return 0;
}
