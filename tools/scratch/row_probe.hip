// How fast can the energy pass's row cache be read?  65 536 walkers x 64 electrons x 1 280-byte rows in a two-slot cache
// ([electron][slot][walker][160 doubles]; the slot of a (walker, electron) is random), wave = (64 walkers, electron):
//   own:  every lane streams its own row, 16 bytes per load (k_kinetic_lw's pattern; 64 lines per instruction)
//   quad: the four lanes of a quad read the four walkers' rows a 64-byte line at a time
//   wave: the wave reads one walker's row after the other, 1 KB contiguous per instruction (+ the fifth component of four walkers)
// Build: hipcc --offload-arch=gfx950 -O3 tools/scratch/row_probe.hip -o tools/scratch/bin/row_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
#define ROW 160
__global__ __launch_bounds__(64) void k_own(const double* rc, const uint8_t* sel, long W, int N, double* out, int depth) {
  const long w = (long)(blockIdx.x % (W / 64)) * 64 + threadIdx.x;
  const int i = blockIdx.x / (W / 64);
  const double* row = rc + (((size_t)i * 2 + sel[(size_t)i * W + w]) * W + w) * ROW;
  double s = 0.0;
  for (int k = 0; k < ROW / 2; k += 8) {
    double2 v[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) v[u] = *reinterpret_cast<const double2*>(row + 2 * (k + u));
#pragma unroll
    for (int u = 0; u < 8; ++u) s += v[u].x + v[u].y;
  }
  out[(size_t)i * W + w] = s;
}
template <int MAP, int INV>
__global__ __launch_bounds__(64) void k_own2(const double* rc, const uint8_t* sel, long W, int N, double* out, const double* Tt) {
  long w; int i;
  if (MAP) {  // k_kinetic_lw's order: the 64 electron blocks of a walker group 8 apart
    const long chunk = (long)blockIdx.x / (8 * N);
    const int rem = (int)((long)blockIdx.x % (8 * N));
    w = (chunk * 8 + (rem & 7)) * 64 + threadIdx.x; i = rem >> 3;
  } else { w = (long)(blockIdx.x % (W / 64)) * 64 + threadIdx.x; i = blockIdx.x / (W / 64); }
  const double* row = rc + (((size_t)i * 2 + sel[(size_t)i * W + w]) * W + w) * ROW;
  double s = 0.0;
  double t[32];
  if (INV) {
    if (INV == 2) {  // tile-blocked inverse [i][W / 64][j][64]
      const double* Ti = Tt + ((size_t)i * (W / 64) + w / 64) * 32 * 64 + (w & 63);
#pragma unroll
      for (int u = 0; u < 32; ++u) t[u] = Ti[u * 64];
    } else {
    const double* Ti = Tt + (size_t)i * 32 * W + w;
#pragma unroll
    for (int u = 0; u < 32; ++u) t[u] = Ti[(size_t)u * W];
    }
  }
  for (int k = 0; k < ROW / 2; k += 8) {
    double2 v[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) v[u] = *reinterpret_cast<const double2*>(row + 2 * (k + u));
#pragma unroll
    for (int u = 0; u < 8; ++u) s += INV ? v[u].x * t[(2 * (k + u)) & 31] + v[u].y * t[(2 * (k + u) + 1) & 31] : v[u].x + v[u].y;
  }
  out[(size_t)i * W + w] = s;
}
// t re-read with every batch (few registers, all the occupancy the hardware has)
template <int MAP>
__global__ __launch_bounds__(64) void k_own3(const double* rc, const uint8_t* sel, long W, int N, double* out, const double* Tt) {
  long w; int i;
  if (MAP) {
    const long chunk = (long)blockIdx.x / (8 * N);
    const int rem = (int)((long)blockIdx.x % (8 * N));
    w = (chunk * 8 + (rem & 7)) * 64 + threadIdx.x; i = rem >> 3;
  } else { w = (long)(blockIdx.x % (W / 64)) * 64 + threadIdx.x; i = blockIdx.x / (W / 64); }
  const double* row = rc + (((size_t)i * 2 + sel[(size_t)i * W + w]) * W + w) * ROW;
  const double* Ti = Tt + (size_t)i * 32 * W + w;
  double s = 0.0;
  for (int c = 0; c < 5; ++c)
    for (int h = 0; h < 2; ++h) {
      double2 v[8]; double t[16];
#pragma unroll
      for (int u = 0; u < 8; ++u) v[u] = *reinterpret_cast<const double2*>(row + c * 32 + h * 16 + 2 * u);
#pragma unroll
      for (int u = 0; u < 16; ++u) t[u] = Ti[(size_t)(h * 16 + u) * W];
#pragma unroll
      for (int u = 0; u < 8; ++u) s += v[u].x * t[2 * u] + v[u].y * t[2 * u + 1];
    }
  out[(size_t)i * W + w] = s;
}
// rows with non-temporal loads / inverse with non-temporal loads / both; t first or rows first
template <int NTR, int NTI, int ORDER>
__global__ __launch_bounds__(64) void k_own4(const double* rc, const uint8_t* sel, long W, int N, double* out, const double* Tt) {
  const long w = (long)(blockIdx.x % (W / 64)) * 64 + threadIdx.x; const int i = blockIdx.x / (W / 64);
  const double* row = rc + (((size_t)i * 2 + sel[(size_t)i * W + w]) * W + w) * ROW;
  const double* Ti = Tt + (size_t)i * 32 * W + w;
  double s = 0.0, t[32];
  if (ORDER == 0) {
#pragma unroll
    for (int u = 0; u < 32; ++u) t[u] = NTI ? __builtin_nontemporal_load(Ti + (size_t)u * W) : Ti[(size_t)u * W];
  }
  double acc[5] = {0, 0, 0, 0, 0};
  for (int c = 0; c < 5; ++c) {
    double v[32];
#pragma unroll
    for (int u = 0; u < 32; ++u) v[u] = NTR ? __builtin_nontemporal_load(row + c * 32 + u) : row[c * 32 + u];
    if (ORDER == 0) {
#pragma unroll
      for (int u = 0; u < 32; ++u) s += v[u] * t[u];
    } else {
#pragma unroll
      for (int u = 0; u < 32; ++u) acc[c] += v[u] * (double)(u + 1);
    }
  }
  if (ORDER == 1) {  // (not the same arithmetic: only the traffic matters here)
#pragma unroll
    for (int u = 0; u < 32; ++u) s += (NTI ? __builtin_nontemporal_load(Ti + (size_t)u * W) : Ti[(size_t)u * W]) * acc[u % 5];
  }
  out[(size_t)i * W + w] = s;
}
// independent blocks: even blocks stream rows, odd blocks read the inverse planes (no wave does both)
__global__ __launch_bounds__(64) void k_roles(const double* rc, const uint8_t* sel, long W, int N, double* out, const double* Tt) {
  const unsigned b = blockIdx.x >> 1;
  const long w = (long)(b % (W / 64)) * 64 + threadIdx.x; const int i = b / (W / 64);
  double s = 0.0;
  if (blockIdx.x & 1) {
    const double* Ti = Tt + (size_t)i * 32 * W + w;
#pragma unroll
    for (int u = 0; u < 32; ++u) s += Ti[(size_t)u * W];
    out[(size_t)i * W + w] = s;
  } else {
    const double* row = rc + (((size_t)i * 2 + sel[(size_t)i * W + w]) * W + w) * ROW;
    for (int k = 0; k < ROW / 2; k += 8) {
      double2 v[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) v[u] = *reinterpret_cast<const double2*>(row + 2 * (k + u));
#pragma unroll
      for (int u = 0; u < 8; ++u) s += v[u].x + v[u].y;
    }
    out[(size_t)(N - 1 - i) * W + w] = s;
  }
}
// the inverse alone: [i][j][W] planes (PAT 0), tile-blocked (1), quad pattern on planes (2)
template <int PAT>
__global__ __launch_bounds__(64) void k_inv(const double* rc, const uint8_t* sel, long W, int N, double* out, const double* Tt) {
  const long w = (long)(blockIdx.x % (W / 64)) * 64 + threadIdx.x; const int i = blockIdx.x / (W / 64);
  double s = 0.0;
  if (PAT == 0) {
    const double* Ti = Tt + (size_t)i * 32 * W + w;
#pragma unroll
    for (int u = 0; u < 32; ++u) s += Ti[(size_t)u * W];
  } else if (PAT == 1) {
    const double* Ti = Tt + ((size_t)i * (W / 64) + w / 64) * 32 * 64 + (w & 63);
#pragma unroll
    for (int u = 0; u < 32; ++u) s += Ti[u * 64];
  } else {
    const int q = threadIdx.x & 3;
    const double* Tq = Tt + ((size_t)i * 32 + 2 * q) * W + (w - q);
#pragma unroll
    for (int j = 0; j < 32; j += 8) {
      const double4 ta = *reinterpret_cast<const double4*>(Tq + (size_t)j * W), tb = *reinterpret_cast<const double4*>(Tq + (size_t)(j + 1) * W);
      s += ta.x + ta.y + ta.z + ta.w + tb.x + tb.y + tb.z + tb.w;
    }
  }
  out[(size_t)i * W + w] = s;
}
// quad-cooperative rows, the quad's share of the inverse (k_kinetic_lw V == 3)
template <int MAP>
__global__ __launch_bounds__(64) void k_quad3(const double* rc, const uint8_t* sel, long W, int N, double* out, const double* Tt) {
  long w; int i;
  if (MAP) {
    const long chunk = (long)blockIdx.x / (8 * N);
    const int rem = (int)((long)blockIdx.x % (8 * N));
    w = (chunk * 8 + (rem & 7)) * 64 + threadIdx.x; i = rem >> 3;
  } else { w = (long)(blockIdx.x % (W / 64)) * 64 + threadIdx.x; i = blockIdx.x / (W / 64); }
  const int q = threadIdx.x & 3;
  const long wq = w - q;
  const double* rq[4];
  for (int t = 0; t < 4; ++t) rq[t] = rc + (((size_t)i * 2 + sel[(size_t)i * W + wq + t]) * W + wq + t) * ROW + 2 * q;
  const double* Tq = Tt + ((size_t)i * 32 + 2 * q) * W + wq;
  double s = 0.0;
  for (int j = 0; j < 32; j += 8) {
    const double4 ta = *reinterpret_cast<const double4*>(Tq + (size_t)j * W), tb = *reinterpret_cast<const double4*>(Tq + (size_t)(j + 1) * W);
    const double t0[4] = {ta.x, ta.y, ta.z, ta.w}, t1[4] = {tb.x, tb.y, tb.z, tb.w};
    double2 v[20];
#pragma unroll
    for (int c = 0; c < 5; ++c)
#pragma unroll
      for (int t = 0; t < 4; ++t) v[c * 4 + t] = *reinterpret_cast<const double2*>(rq[t] + c * 32 + j);
#pragma unroll
    for (int c = 0; c < 5; ++c)
#pragma unroll
      for (int t = 0; t < 4; ++t) s += v[c * 4 + t].x * t0[t] + v[c * 4 + t].y * t1[t];
  }
  out[(size_t)i * W + w] = s;
}
__global__ __launch_bounds__(64) void k_quad(const double* rc, const uint8_t* sel, long W, int N, double* out, int depth) {
  const long w = (long)(blockIdx.x % (W / 64)) * 64 + threadIdx.x;
  const int i = blockIdx.x / (W / 64);
  const int q = threadIdx.x & 3;
  const long wq = w - q;
  const double* rq[4];
  for (int t = 0; t < 4; ++t) rq[t] = rc + (((size_t)i * 2 + sel[(size_t)i * W + wq + t]) * W + wq + t) * ROW + 2 * q;
  double s = 0.0;
  for (int k = 0; k < ROW; k += 16) {
    double2 v[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) v[u] = *reinterpret_cast<const double2*>(rq[u & 3] + k + 8 * (u >> 2));
#pragma unroll
    for (int u = 0; u < 8; ++u) s += v[u].x + v[u].y;
  }
  out[(size_t)i * W + w] = s;
}
__global__ __launch_bounds__(64) void k_wave(const double* rc, const uint8_t* sel, long W, int N, double* out, int depth) {
  const long w0 = (long)(blockIdx.x % (W / 64)) * 64;
  const int i = blockIdx.x / (W / 64);
  const int lane = threadIdx.x;
  const int myslot = sel[(size_t)i * W + w0 + lane];
  double s = 0.0;
  for (int wl = 0; wl < 64; wl += 8) {
    double2 v[10];
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      const double* r4[4];
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        const int sl = __builtin_amdgcn_readlane(myslot, wl + 4 * h + t);
        r4[t] = rc + (((size_t)i * 2 + sl) * W + w0 + wl + 4 * h + t) * ROW;
      }
#pragma unroll
      for (int t = 0; t < 4; ++t) v[5 * h + t] = *reinterpret_cast<const double2*>(r4[t] + 2 * lane);  // components 0-3: 1 KB
      const double* r5 = (lane >> 4) == 0 ? r4[0] : (lane >> 4) == 1 ? r4[1] : (lane >> 4) == 2 ? r4[2] : r4[3];
      v[5 * h + 4] = *reinterpret_cast<const double2*>(r5 + 128 + 2 * (lane & 15));  // component 4 of the four walkers
    }
#pragma unroll
    for (int u = 0; u < 10; ++u) s += v[u].x + v[u].y;
  }
  out[(size_t)i * W + w0 + lane] = s;
}
int main() {
  const long W = 65536; const int N = 64;
  const size_t nrc = (size_t)N * 2 * W * ROW;
  double *rc, *out; uint8_t* sel;
  hipMalloc(&rc, nrc * 8); hipMalloc(&out, (size_t)N * W * 8); hipMalloc(&sel, (size_t)N * W);
  hipMemset(rc, 0, nrc * 8);
  std::vector<uint8_t> hs((size_t)N * W);
  uint32_t x = 12345;
  for (auto& b : hs) { x = x * 1664525u + 1013904223u; b = (x >> 16) & 1; }
  hipMemcpy(sel, hs.data(), hs.size(), hipMemcpyHostToDevice);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  const double bytes = (double)N * W * ROW * 8;
  auto run = [&](const char* name, void (*k)(const double*, const uint8_t*, long, int, double*, int)) {
    for (int rep = 0; rep < 3; ++rep) {
      hipEventRecord(e0);
      hipLaunchKernelGGL(k, dim3((unsigned)(W / 64 * N)), dim3(64), 0, 0, rc, sel, W, N, out, 0);
      hipEventRecord(e1); hipEventSynchronize(e1);
      float ms; hipEventElapsedTime(&ms, e0, e1);
      if (rep == 2) printf("%-6s %8.1f us  %6.0f GB/s\n", name, ms * 1e3, bytes / ms / 1e6);
    }
  };
  run("own", k_own); run("quad", k_quad); run("wave", k_wave);
  double* Tt; hipMalloc(&Tt, (size_t)N * 32 * W * 8); hipMemset(Tt, 0, (size_t)N * 32 * W * 8);
  auto run2 = [&](const char* name, void (*k)(const double*, const uint8_t*, long, int, double*, const double*), double by) {
    for (int rep = 0; rep < 3; ++rep) {
      hipEventRecord(e0);
      hipLaunchKernelGGL(k, dim3((unsigned)(W / 64 * N)), dim3(64), 0, 0, rc, sel, W, N, out, Tt);
      hipEventRecord(e1); hipEventSynchronize(e1);
      float ms; hipEventElapsedTime(&ms, e0, e1);
      if (rep == 2) printf("%-22s %8.1f us  %6.0f GB/s\n", name, ms * 1e3, by / ms / 1e6);
    }
  };
  const double binv = (double)N * 32 * W * 8;
  run2("own, kernel order", k_own2<1, 0>, bytes);
  run2("own + inverse", k_own2<0, 1>, bytes + binv);
  run2("own + inverse, k.order", k_own2<1, 1>, bytes + binv);
  run2("own4 plain", k_own4<0, 0, 0>, bytes + binv);
  run2("own4 nt rows", k_own4<1, 0, 0>, bytes + binv);
  run2("own4 nt inverse", k_own4<0, 1, 0>, bytes + binv);
  run2("own4 nt both", k_own4<1, 1, 0>, bytes + binv);
  run2("own4 inverse last", k_own4<0, 0, 1>, bytes + binv);
  for (int rep = 0; rep < 3; ++rep) {
    hipEventRecord(e0);
    hipLaunchKernelGGL(k_roles, dim3((unsigned)(W / 64 * N * 2)), dim3(64), 0, 0, rc, sel, W, N, out, Tt);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    if (rep == 2) printf("%-22s %8.1f us  %6.0f GB/s\n", "roles: rows | inverse", ms * 1e3, (bytes + binv) / ms / 1e6);
  }
  run2("inverse alone, planes", k_inv<0>, binv);
  run2("inverse alone, tiled", k_inv<1>, binv);
  run2("inverse alone, quad", k_inv<2>, binv);
  run2("own, t per batch", k_own3<0>, bytes + binv);
  run2("own, t per batch, k.ord", k_own3<1>, bytes + binv);
  run2("quad + quad inverse", k_quad3<0>, bytes + binv);
  run2("quad + quad inv, k.ord", k_quad3<1>, bytes + binv);
  run2("own + tiled inverse", k_own2<0, 2>, bytes + binv);
  run2("own + tiled inv, k.ord", k_own2<1, 2>, bytes + binv);
  for (auto& b : hs) b = 0;  // every row in slot 0: contiguous
  hipMemcpy(sel, hs.data(), hs.size(), hipMemcpyHostToDevice);
  printf("all rows in slot 0:\n");
  run("own", k_own); run("quad", k_quad); run("wave", k_wave);
  return 0;
}
