// Does hipEventElapsedTime work on events recorded by captured graph nodes (ROCm 7.2, gfx950)?
#include <hip/hip_runtime.h>
#include <cstdio>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s -> %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)
__global__ void spin(double* a, int n) { double s = a[threadIdx.x]; for (int i = 0; i < n; ++i) s = s * 1.0000001 + 1e-9; a[threadIdx.x] = s; }
int main() {
  hipStream_t st; CK(hipStreamCreate(&st));
  double* d; CK(hipMalloc(&d, 4096));
  hipEvent_t e[8];
  for (auto& x : e) CK(hipEventCreate(&x));
  hipGraph_t g; hipGraphExec_t ge;
  CK(hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal));
  for (int k = 0; k < 4; ++k) {
    CK(hipEventRecord(e[2 * k], st));
    hipLaunchKernelGGL(spin, dim3(64), dim3(64), 0, st, d, 20000 * (k + 1));
    CK(hipEventRecord(e[2 * k + 1], st));
  }
  CK(hipStreamEndCapture(st, &g));
  CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
  for (int rep = 0; rep < 3; ++rep) {
    CK(hipGraphLaunch(ge, st));
    CK(hipStreamSynchronize(st));
    for (int k = 0; k < 4; ++k) {
      float ms = -1.f;
      hipError_t er = hipEventElapsedTime(&ms, e[2 * k], e[2 * k + 1]);
      printf("rep %d kernel %d: %s ms=%f\n", rep, k, hipGetErrorString(er), ms);
    }
  }
  // launch overhead comparison: 400 tiny kernels, stream vs graph
  hipGraph_t g2; hipGraphExec_t ge2;
  CK(hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal));
  for (int k = 0; k < 400; ++k) hipLaunchKernelGGL(spin, dim3(64), dim3(64), 0, st, d, 200);
  CK(hipStreamEndCapture(st, &g2));
  CK(hipGraphInstantiate(&ge2, g2, nullptr, nullptr, 0));
  hipEvent_t t0, t1; CK(hipEventCreate(&t0)); CK(hipEventCreate(&t1));
  for (int rep = 0; rep < 3; ++rep) {
    CK(hipEventRecord(t0, st));
    for (int k = 0; k < 400; ++k) hipLaunchKernelGGL(spin, dim3(64), dim3(64), 0, st, d, 200);
    CK(hipEventRecord(t1, st)); CK(hipEventSynchronize(t1));
    float a; CK(hipEventElapsedTime(&a, t0, t1));
    CK(hipEventRecord(t0, st));
    CK(hipGraphLaunch(ge2, st));
    CK(hipEventRecord(t1, st)); CK(hipEventSynchronize(t1));
    float b; CK(hipEventElapsedTime(&b, t0, t1));
    printf("400 tiny kernels: stream %.3f ms, graph %.3f ms\n", a, b);
  }
  return 0;
}
