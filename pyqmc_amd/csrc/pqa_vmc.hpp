// Fused single-electron-move kernels for the VMC sweep: the body of vmc_worker's electron loop
// (pyqmc/method/mc.py:115-137) split around the one orbital evaluation a move needs.
//
//   k_propose : drift at the current position (Slater part from the per-electron orbital cache,
//               Jastrow part recomputed), proposal r' = r + sqrt(tau) z + tau limdrift(grad)
//   k_orb<5>  : orbitals (value, gradient, laplacian) of electron e at r' for all walkers
//   k_accept  : ratio and reverse drift at r', Metropolis test, and for accepted walkers the
//               Sherman-Morrison update, Jastrow sums, coordinate and cache commit
//
// The reference evaluates the orbitals twice per move (old and new position, mc.py:117,124); the
// cached rows of the last accepted position are the same numbers, so only the new position is
// evaluated here, and the cached Laplacian row makes the kinetic energy free of AO work.
#pragma once
#include "pqa_common.hpp"
#include "pqa_jastrow.hpp"
#include "pqa_slater.hpp"
#include "pqa_cslater.hpp"

struct MoveBuf {
  double* newpos;   // [W][3]
  double* aux;      // [W][8]: gauss*sqrt(tau) (3), limited drift (3), U_old, unused
  const double* gauss;  // tape [N][W][3] for this step or NULL
  const double* unif;   // tape [N][W] for this step or NULL
  uint8_t* accept;  // [W] accept flags of this move
  uint8_t* accept_rec;  // [N][W] record for this step or NULL
  int* dwrap;       // [W][3] wrap removed when the proposal was folded into the periodic cell (NULL: open system)
  int* wrap;        // [W][N][3] wraps accumulated by accepted moves since the last recompute (NULL: open system)
  int* acc_w;       // [W] accepted moves of this walker in this step (no same-address atomics: they serialise at ~12 ns each)
  uint64_t seed;
  uint32_t step;
  double tstep;
  // DMC drift-diffusion (dmc.py:38-70): Umrigar's limited drift, fixed-node rejection, per-walker diffusion statistics
  int dmc;
  double* r2_prop;  // [W] sum over electrons of |gauss + drift|^2 of every proposal
  double* r2_acc;   // [W] the same for accepted proposals
};

__device__ __forceinline__ void limdrift3(double& gx, double& gy, double& gz) {  // mc.py:76-89, cutoff 1
  const double tot = sqrt(gx * gx + gy * gy + gz * gz);
  if (tot > 1.0) { gx /= tot; gy /= tot; gz /= tot; }
}
__device__ __forceinline__ double finite_or(double v, double alt) { return (v >= -DBL_MAX && v <= DBL_MAX) ? v : alt; }
// Umrigar's limiter (dmc.py:22-35, acyrus = 0.5): g -> g * tau_eff, tau_eff = (sqrt(1 + 2 tau a |g|^2) - 1) / (a |g|^2)
__device__ __forceinline__ void limdrift_dmc(double& gx, double& gy, double& gz, double tau) {
  const double v2 = gx * gx + gy * gy + gz * gz, a = 0.5;
  const double te = (v2 > 1e-8) ? (sqrt(1.0 + 2.0 * tau * a * v2) - 1.0) / (a * v2) : tau;
  gx *= te; gy *= te; gz *= te;
}

// Slater part of a move: gradient of log|Psi_S| (real part for complex orbitals: the drift uses np.real(grad),
// mc.py:118,126) and |ratio|^2, sanitised like gradient_value (slater.py:414-417).  CX: complex determinants.
template <bool CX>
__device__ __forceinline__ void slater_move_terms(const SysDev& S, const SlaterState& st, int s, int i, long w,
                                                  const double* __restrict__ row, double* lds, double& gx, double& gy,
                                                  double& gz, double& val2, double* sgn = nullptr) {
  if (CX) {
    cx r[5];
    slater_ratios_c<5>(S, st, s, i, w, row, r, lds);
    const double d = 1.0 / cabs2(r[0]);  // Re(r_c / r_0) = (r_c . conj r_0) / |r_0|^2
    gx += finite_or((r[1].r * r[0].r + r[1].i * r[0].i) * d, 0.0);
    gy += finite_or((r[2].r * r[0].r + r[2].i * r[0].i) * d, 0.0);
    gz += finite_or((r[3].r * r[0].r + r[3].i * r[0].i) * d, 0.0);
    val2 = finite_or(cabs2(r[0]), 1.0);
  } else {
    double r[5];
    slater_ratios<5>(S, st, s, i, w, row, r, lds);
    gx += finite_or(r[1] / r[0], 0.0); gy += finite_or(r[2] / r[0], 0.0); gz += finite_or(r[3] / r[0], 0.0);
    const double v = finite_or(r[0], 1.0);
    val2 = v * v;
    if (sgn) *sgn = (v > 0.0) ? 1.0 : ((v < 0.0) ? -1.0 : 0.0);  // np.sign of the determinant ratio (the Jastrow ratio is positive)
  }
}

#ifdef PQA_WW_CLK  // timing build only (tools/scratch/ww_clk.py): 100 MHz stamps inside k_propose (0-3) and k_accept (4-7)
static __device__ unsigned long long pqa_ww_clk[1024 * 8];
#define PQA_WCLK(k) do { if (blockIdx.x < 1024 && threadIdx.x == 0) pqa_ww_clk[blockIdx.x * 8 + (k)] = wall_clock64(); } while (0)
#else
#define PQA_WCLK(k) do { } while (0)
#endif
template <bool CX>
static __global__ __launch_bounds__(64) void k_propose(SysDev S, SlaterState st, JastrowState js, MoveBuf mb, int e,
                                                int has_slater, int has_jastrow, long W) {
  extern __shared__ double lds[];
  const long w = blockIdx.x;
  const double* xw = js.x + (size_t)w * S.nelec * 3;
  const double ex = xw[3 * e], ey = xw[3 * e + 1], ez = xw[3 * e + 2];
  double gx = 0.0, gy = 0.0, gz = 0.0, U0 = 0.0;
  PQA_WCLK(0);
  if (has_slater) {
    const int s = e >= S.nup, i = e - s * S.nup, n = s ? S.ndn : S.nup, nmo = S.nmo[s];
    double v2;
    slater_move_terms<CX>(S, st, s, i, w, st.cache[s] + ((size_t)w * n + i) * 5 * nmo, lds, gx, gy, gz, v2);
  }
  PQA_WCLK(1);
  if (has_jastrow) {
    double g[3], lp;
#ifdef PQA_WW_CLK
    jas_eval<1>(S, xw, e, ex, ey, ez, U0, g, lp, 1, lds + S.j3_off);
    PQA_WCLK(2);
    { double u3 = 0.0, g3[3] = {0.0, 0.0, 0.0}; jas_eval<1>(S, xw, e, ex, ey, ez, u3, g3, lp, 2, lds + S.j3_off); U0 += u3; g[0] += g3[0]; g[1] += g3[1]; g[2] += g3[2]; }
#else
    jas_eval<1>(S, xw, e, ex, ey, ez, U0, g, lp, 3, lds + S.j3_off);
#endif
    gx += g[0]; gy += g[1]; gz += g[2];
  }
  PQA_WCLK(3);
  if (mb.dmc) limdrift_dmc(gx, gy, gz, mb.tstep);  // the drift vector itself (already times tau_eff)
  else limdrift3(gx, gy, gz);
  if (threadIdx.x == 0) {
    double z0, z1, z2, z3;
    if (mb.gauss) {
      const double* zt = mb.gauss + ((size_t)e * W + w) * 3;
      z0 = zt[0]; z1 = zt[1]; z2 = zt[2];
    } else {
      normal2(philox(mb.seed, (uint32_t)w, (uint32_t)e, PQA_STREAM_GAUSS_A, mb.step), z0, z1);
      normal2(philox(mb.seed, (uint32_t)w, (uint32_t)e, PQA_STREAM_GAUSS_B, mb.step), z2, z3);
    }
    const double sq = sqrt(mb.tstep);
    z0 *= sq; z1 *= sq; z2 *= sq;
    const double df = mb.dmc ? 1.0 : mb.tstep;
    double* np_ = mb.newpos + 3 * w;
    np_[0] = ex + z0 + gx * df;  // mc.py:120 / dmc.py:52
    np_[1] = ey + z1 + gy * df;
    np_[2] = ez + z2 + gz * df;
    if (mb.dwrap) fold_cell(S, np_[0], np_[1], np_[2], mb.dwrap + 3 * w);  // make_irreducible, mc.py:121
    double* a = mb.aux + 8 * w;
    a[0] = z0; a[1] = z1; a[2] = z2; a[3] = gx; a[4] = gy; a[5] = gz; a[6] = U0;
  }
}

// motmp: [W][5][nmo_s] orbitals at the proposed position.  LDS: n(n+1)+2n doubles (+ multi-det scratch).
template <bool CX>
static __global__ __launch_bounds__(64) void k_accept(SysDev S, SlaterState st, JastrowState js, MoveBuf mb, int e,
                                               int has_slater, int has_jastrow, const double* __restrict__ motmp, long W) {
  extern __shared__ double lds[];
  const long w = blockIdx.x;
  const int lane = threadIdx.x;
  const int s = e >= S.nup, i = e - s * S.nup, n = s ? S.ndn : S.nup, nmo = S.nmo[s];
  const double* xw = js.x + (size_t)w * S.nelec * 3;
  const double* a = mb.aux + 8 * w;
  const double nx = mb.newpos[3 * w], ny = mb.newpos[3 * w + 1], nz = mb.newpos[3 * w + 2];
  double val2 = 1.0, gx = 0.0, gy = 0.0, gz = 0.0;  // val2 = |Psi(new)/Psi|^2 (mc.py:131)
  const double* row = motmp + (size_t)w * 5 * nmo;
  double sgn = 1.0;
  PQA_WCLK(4);
  if (has_slater) slater_move_terms<CX>(S, st, s, i, w, row, lds, gx, gy, gz, val2, &sgn);
  PQA_WCLK(5);
  if (has_jastrow) {
    double g[3], lp, U;
    jas_eval<1>(S, xw, e, nx, ny, nz, U, g, lp, 3, lds + S.j3_off);
    gx += g[0]; gy += g[1]; gz += g[2];
    const double ej = exp(U - a[6]);
    val2 *= ej * ej;
  }
  double bx, by, bz;
  if (mb.dmc) {  // dmc.py:57-60: backward = gauss + drift(old) + drift(new)
    limdrift_dmc(gx, gy, gz, mb.tstep);
    bx = a[0] + a[3] + gx; by = a[1] + a[4] + gy; bz = a[2] + a[5] + gz;
  } else {
    limdrift3(gx, gy, gz);
    bx = a[0] + mb.tstep * (a[3] + gx); by = a[1] + mb.tstep * (a[4] + gy); bz = a[2] + mb.tstep * (a[5] + gz);
  }
  const double fwd = a[0] * a[0] + a[1] * a[1] + a[2] * a[2];
  const double bwd = bx * bx + by * by + bz * bz;
  const double t_prob = exp(1.0 / (2.0 * mb.tstep) * (fwd - bwd));  // mc.py:130
  double ratio = val2 * t_prob;
  if (mb.dmc && !CX) ratio *= sgn;  // fixed node: a sign change is never accepted (dmc.py:64-66)
  double u;
  if (mb.unif) u = mb.unif[(size_t)e * W + w];
  else {
    const Philox p = philox(mb.seed, (uint32_t)w, (uint32_t)e, PQA_STREAM_ACCEPT, mb.step);
    u = u01(p.c[0], p.c[1]);
  }
  const bool acc = ratio > u;
  PQA_WCLK(6);
  if (mb.dmc && lane == 0) {  // dmc.py:68 r2 = |gauss + drift|^2
    const double r2 = (a[0] + a[3]) * (a[0] + a[3]) + (a[1] + a[4]) * (a[1] + a[4]) + (a[2] + a[5]) * (a[2] + a[5]);
    mb.r2_prop[w] += r2;
    if (acc) mb.r2_acc[w] += r2;
  }
  if (lane == 0) {
    mb.accept[w] = acc;
    if (mb.accept_rec) mb.accept_rec[(size_t)e * W + w] = acc;
    if (acc) mb.acc_w[w] += 1;
  }
  if (!acc) return;
  if (has_slater) {
    if (CX) sm_update_wave_c(S, st, s, i, w, row, lds);
    else sm_update_wave(S, st, s, i, w, row, lds);
    double* c = st.cache[s] + ((size_t)w * n + i) * 5 * nmo;
    for (int k = lane; k < 5 * nmo; k += 64) c[k] = row[k];
  }
  // The public Jastrow sums (_avalues/_bvalues) are not needed by the sweep itself (ratios come from
  // U_e(new) - U_e(old)); the host marks them stale and rebuilds them from x on demand.
  if (lane == 0) {
    double* x = js.x + (size_t)w * S.nelec * 3 + 3 * e;
    x[0] = nx; x[1] = ny; x[2] = nz;
    if (mb.wrap) {
      int* wr = mb.wrap + ((size_t)w * S.nelec + e) * 3;
      wr[0] += mb.dwrap[3 * w]; wr[1] += mb.dwrap[3 * w + 1]; wr[2] += mb.dwrap[3 * w + 2];
    }
  }
  PQA_WCLK(7);
}

// accepted-move count of one sweep: sum acc_w[0..W) -> *out, and reset acc_w.  One block, deterministic.
template <int PQA_UNIT = 0>  // (a template so that only the units that launch it compile it)
static __global__ __launch_bounds__(1024) void k_sum_reset_int(int* __restrict__ acc_w, long W, int* __restrict__ out) {
  __shared__ int part[1024];
  int s = 0;
  long i = threadIdx.x;
  for (; i + 7 * 1024 < W; i += 8 * 1024) {  // eight loads in flight per thread
    int v[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) v[u] = acc_w[i + u * 1024];
#pragma unroll
    for (int u = 0; u < 8; ++u) { s += v[u]; acc_w[i + u * 1024] = 0; }
  }
  for (; i < W; i += 1024) { s += acc_w[i]; acc_w[i] = 0; }
  part[threadIdx.x] = s;
  __syncthreads();
  for (int off = 512; off > 0; off >>= 1) {
    if ((int)threadIdx.x < off) part[threadIdx.x] += part[threadIdx.x + off];
    __syncthreads();
  }
  if (threadIdx.x == 0) *out = part[0];
}
