// HBM streaming-bandwidth probe: how fast can this box stream with 8 / 16 B per lane, 1-8 independent loads in flight per thread,
// various grid sizes?  (decides whether the lane-per-walker kernels' ~3.5-3.9 TB/s is the DRAM limit or a memory-level-parallelism limit)
// hipcc --offload-arch=gfx950 -O3 tools/scratch/bw_probe.hip -o tools/scratch/bw_probe
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double d2 __attribute__((ext_vector_type(2)));
template <int U>
__global__ __launch_bounds__(256) void k_r8(const double* __restrict__ in, size_t n, double* out) {
  double s[U];
  for (int u = 0; u < U; ++u) s[u] = 0;
  const size_t stride = (size_t)gridDim.x * 256;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i + (U - 1) * stride < n; i += U * stride) {
#pragma unroll
    for (int u = 0; u < U; ++u) s[u] += in[i + u * stride];
  }
  double t = 0; for (int u = 0; u < U; ++u) t += s[u];
  if (t == 1.2345e300) out[0] = t;
}
template <int U>
__global__ __launch_bounds__(256) void k_r16(const d2* __restrict__ in, size_t n2, double* out) {
  double s[U];
  for (int u = 0; u < U; ++u) s[u] = 0;
  const size_t stride = (size_t)gridDim.x * 256;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i + (U - 1) * stride < n2; i += U * stride) {
#pragma unroll
    for (int u = 0; u < U; ++u) { const d2 v = in[i + u * stride]; s[u] += v.x + v.y; }
  }
  double t = 0; for (int u = 0; u < U; ++u) t += s[u];
  if (t == 1.2345e300) out[0] = t;
}
template <int U>
__global__ __launch_bounds__(256) void k_rmw16(d2* __restrict__ io, size_t n2) {
  const size_t stride = (size_t)gridDim.x * 256;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i + (U - 1) * stride < n2; i += U * stride) {
    d2 v[U];
#pragma unroll
    for (int u = 0; u < U; ++u) v[u] = io[i + u * stride];
#pragma unroll
    for (int u = 0; u < U; ++u) { v[u].x = v[u].x * 1.0000001 + 1.0; v[u].y += 1.0; io[i + u * stride] = v[u]; }
  }
}
#define T(name, launch, bytes) do { hipEvent_t a, b; (void)hipEventCreate(&a); (void)hipEventCreate(&b); float best = 1e9; \
  for (int r = 0; r < 3; ++r) { (void)hipEventRecord(a); launch; (void)hipEventRecord(b); (void)hipEventSynchronize(b); float ms; (void)hipEventElapsedTime(&ms, a, b); if (ms < best) best = ms; } \
  printf("%-28s %7.3f ms %6.0f GB/s\n", name, best, (bytes) / best * 1e-6); } while (0)
int main() {
  const size_t n = (size_t)1 << 28;
  double *buf, *out;
  (void)hipMalloc(&buf, n * 8); (void)hipMalloc(&out, 64); (void)hipMemset(buf, 0, n * 8);
  for (int g : {2048, 8192, 32768}) {
    char nm[64];
    snprintf(nm, 64, "r8 U1 grid %d", g); T(nm, hipLaunchKernelGGL(k_r8<1>, dim3(g), dim3(256), 0, 0, buf, n, out), n * 8.0);
    snprintf(nm, 64, "r8 U4 grid %d", g); T(nm, hipLaunchKernelGGL(k_r8<4>, dim3(g), dim3(256), 0, 0, buf, n, out), n * 8.0);
    snprintf(nm, 64, "r8 U8 grid %d", g); T(nm, hipLaunchKernelGGL(k_r8<8>, dim3(g), dim3(256), 0, 0, buf, n, out), n * 8.0);
    snprintf(nm, 64, "r16 U1 grid %d", g); T(nm, hipLaunchKernelGGL(k_r16<1>, dim3(g), dim3(256), 0, 0, (const d2*)buf, n / 2, out), n * 8.0);
    snprintf(nm, 64, "r16 U4 grid %d", g); T(nm, hipLaunchKernelGGL(k_r16<4>, dim3(g), dim3(256), 0, 0, (const d2*)buf, n / 2, out), n * 8.0);
    snprintf(nm, 64, "r16 U8 grid %d", g); T(nm, hipLaunchKernelGGL(k_r16<8>, dim3(g), dim3(256), 0, 0, (const d2*)buf, n / 2, out), n * 8.0);
    snprintf(nm, 64, "rmw16 U4 grid %d", g); T(nm, hipLaunchKernelGGL(k_rmw16<4>, dim3(g), dim3(256), 0, 0, (d2*)buf, n / 2), 2 * n * 8.0);
  }
  return 0;
}
