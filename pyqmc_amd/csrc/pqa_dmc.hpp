// pqa_dmc.hpp — the DMC step on the device (real and, since round 3, complex wave functions; open and periodic systems).
//
// One DMC step of the reference's dmc_propagate (pyqmc/method/dmc.py:123-221) is
//   (1) per electron, one T-move: compute_tmoves (eval_ecp.py:43-80) -> propose_tmoves (dmc.py:73-120) -> accept -> update
//   (2) per electron, one drift-diffusion move with Umrigar's limited drift and fixed-node rejection (dmc.py:38-70):
//       the VMC kernels k_propose / k_accept in their `dmc` mode (pqa_vmc.hpp)
//   (3) local energy, branching factor compute_S (dmc.py:224-235), weight update and weighted averages.
// The kernels here are (1) and (3).  Unlike the host-driven pqa_tmoves (dense [W][P] candidate table, kept for the
// protocol-level API) the fused path compacts the candidates of the walkers that pass the stochastic ECP mask, the same
// count -> scan -> fill -> orbitals -> ratio pipeline as the energy's ECP term: dead candidates carry weight 0 and can never
// be the first crossing of the cumulative distribution, so leaving them out selects the same move.  The orbital
// evaluation is batched over all electrons of the step (two launches per spin and step instead of two per electron:
// a launch over a few hundred points costs the kernel's ~0.2-0.3 ms latency floor whatever its size).
#pragma once
#include "pqa_energy.hpp"
#include "pqa_vmc.hpp"
#include "pqa_lw.hpp"  // jas_eval_lane (tm_ratio_part)

#define PQA_STREAM_TMMASK 6u
#define PQA_STREAM_TM_U1 7u
#define PQA_STREAM_TM_U2 8u

struct TmBuf {
  const double* rot;    // [N][necp][3][3] rotations of the quadrature grids of this step
  const double* unif;   // [N][necp][W] mask uniforms or NULL -> Philox
  const double* u1;     // [N][W] selection uniforms or NULL
  const double* u2;     // [N][W] acceptance uniforms or NULL
  const double* quad;   // [6+12][3]
  uint64_t seed;
  uint32_t step;
  double tau, threshold;
  int* cnt;             // [N][W] live candidates of (electron, walker)
  long* off;            // [N*W+1] exclusive scan of cnt: electron-major, so one spin's candidates are contiguous
  unsigned long long* pass;  // [N][W][ceil(necp/64)] ECP atoms whose mask the (electron, walker) passed
  double* pts;          // [ncand][3]
  double* wgt;          // [ncand] sum_l (exp(-tau v_l/prob) - 1)(2l+1) P_l(cos) w_i
  double* amp;          // [ncand] ratio * weight
  double* rat;          // [ncand] Psi(candidate)/Psi
  int* ptw;             // [ncand] walker of the candidate
  int* acc;             // [N][W] accepted T-moves of this step (0/1)
  long* acc_off;        // [N*W+1] exclusive scan of acc
  int* acc_idx;         // [N*W] their indices e*W + w, ascending (spin-up first)
  double* acc_pos;      // [N*W][3] their new positions
  int nofold;           // twisted handles keep TRUE (unfolded) coordinates: an accepted T-move is not folded into the cell
};

// pass A: which ECP atoms pass the mask for electron e of each walker, and how many candidates that makes.
// The candidates of ALL electrons are laid out before the sequential T-move loop starts: electron e has not moved
// when its turn comes, and the mask / rotation draws do not depend on the state, so positions, weights and orbital
// values are those the reference computes one electron at a time (dmc.py:160-168); only the ratios need the loop.
// grid = (ceil(W/256), N), block = 256.
template <int PQA_UNIT = 0>  // (a template so that only the units that launch it compile it)
static __global__ __launch_bounds__(256) void k_tm_count(SysDev S, JastrowState js, TmBuf B, long W) {
  const long w = (long)blockIdx.x * 256 + threadIdx.x;
  const int e = blockIdx.y;
  if (w >= W) return;
  const double* xw = js.x + (size_t)w * S.nelec * 3;
  const double ex = xw[3 * e], ey = xw[3 * e + 1], ez = xw[3 * e + 2];
  const int nkw = (S.necp + 63) / 64;
  int c = 0;
  for (int kw = 0; kw < nkw; ++kw) {
    unsigned long long m = 0ull;
    for (int k = kw * 64; k < S.necp && k < kw * 64 + 64; ++k) {
      const int ia = S.ecp_atom[k];
      double dx = ex - S.atom_xyz[3 * ia], dy = ey - S.atom_xyz[3 * ia + 1], dz = ez - S.atom_xyz[3 * ia + 2];
      min_image(S, dx, dy, dz);
      const double r = sqrt(dx * dx + dy * dy + dz * dz);
      double v[PQA_MAXCHAN], prob;
      int nch;
      ecp_radial(S, k, r, B.threshold, v, nch, prob);
      double u;
      if (B.unif) u = B.unif[((size_t)e * S.necp + k) * W + w];
      else {
        const Philox p = philox(B.seed, (uint32_t)w, (uint32_t)(e * S.necp + k), PQA_STREAM_TMMASK, B.step);
        u = u01(p.c[0], p.c[1]);
      }
      if (nch > 1 && prob > u) {
        m |= 1ull << (k - kw * 64);
        c += (nch <= 2) ? 6 : 12;
      }
    }
    B.pass[((size_t)e * W + w) * nkw + kw] = m;
  }
  B.cnt[(size_t)e * W + w] = c;
}

// Device-wide exclusive scan of c[n] -> o[n+1] in three small launches (a single block walking a million counters costs
// ~2 ms; this is ~20 us): k_scan_local scans 1024-element tiles and emits tile totals, k_scan_tiles scans those (one block),
// k_scan_add adds the tile offsets and copies o[k*W] (k = 0..n/W) to marks[] — the per-electron totals the host reads back
// in one small copy to size the launches.
template <int PQA_UNIT = 0>  // (a template so that only the units that launch it compile it)
static __global__ __launch_bounds__(1024) void k_scan_local(const int* __restrict__ c, long* __restrict__ o, long n, long* __restrict__ tile_sum) {
  __shared__ long wsum[16];
  const long i = (long)blockIdx.x * 1024 + threadIdx.x;
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const long v = (i < n) ? c[i] : 0;
  long x = v;  // inclusive scan within the wave
#pragma unroll
  for (int d = 1; d < 64; d <<= 1) {
    const long y = __shfl_up(x, d, 64);
    if (lane >= d) x += y;
  }
  if (lane == 63) wsum[wv] = x;
  __syncthreads();
  if (threadIdx.x == 0) {
    long run = 0;
    for (int t = 0; t < 16; ++t) { const long y = wsum[t]; wsum[t] = run; run += y; }
    tile_sum[blockIdx.x] = run;
  }
  __syncthreads();
  if (i < n) o[i] = wsum[wv] + x - v;
}
template <int PQA_UNIT = 0>  // (a template so that only the units that launch it compile it)
static __global__ __launch_bounds__(1024) void k_scan_tiles(long* __restrict__ t, long nt) {  // in place, t[nt] = total
  // (the 1 024 per-thread sums are scanned by the waves — shuffle scan inside a wave, the 16 wave totals by one thread; until round 6 thread 0
  // walked all 1 024 through LDS: 110 us per call, 5 % of the C5 DMC step at 4 096 walkers)
  __shared__ long wsum[16];
  const long per = (nt + 1023) / 1024;
  const long b = (long)threadIdx.x * per, e = (b + per < nt) ? b + per : nt;
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  long sum = 0;
  for (long i = b; i < e; ++i) sum += t[i];
  long x = sum;  // inclusive scan within the wave
#pragma unroll
  for (int d = 1; d < 64; d <<= 1) {
    const long y = __shfl_up(x, d, 64);
    if (lane >= d) x += y;
  }
  if (lane == 63) wsum[wv] = x;
  __syncthreads();
  if (threadIdx.x == 0) {
    long run = 0;
    for (int k = 0; k < 16; ++k) { const long v = wsum[k]; wsum[k] = run; run += v; }
    t[nt] = run;
  }
  __syncthreads();
  long run = wsum[wv] + x - sum;
  for (long i = b; i < e; ++i) { const long v = t[i]; t[i] = run; run += v; }
}
template <int PQA_UNIT = 0>  // (a template so that only the units that launch it compile it)
static __global__ __launch_bounds__(1024) void k_scan_add(long* __restrict__ o, long n, const long* __restrict__ tile_off, long nt, long W,
                                                   long* __restrict__ marks) {
  const long i = (long)blockIdx.x * 1024 + threadIdx.x;
  if (i < n) {
    const long v = o[i] + tile_off[blockIdx.x];
    o[i] = v;
    if (i % W == 0) marks[i / W] = v;
  }
  if (i == 0) { o[n] = tile_off[nt]; marks[n / W] = tile_off[nt]; }
}

// The same scans for small inputs in ONE launch: two count arrays of n <= 16384 entries each (the ECP point counts of the two spin channels of
// a small shard), one block; o0 / o1 get n + 1 entries (exclusive prefix, total last).  Three launches per array were 45 us of a 0.8 ms step.
template <int PQA_UNIT = 0>
static __global__ __launch_bounds__(1024) void k_scan_small2(const int* __restrict__ c0, long* __restrict__ o0, const int* __restrict__ c1,
                                                             long* __restrict__ o1, long n, long* __restrict__ totals) {
  __shared__ long part[2][1024];
  const long per = (n + 1023) / 1024;
  const long b = (long)threadIdx.x * per, e = (b + per < n) ? b + per : n;
  long s0 = 0, s1 = 0;
  for (long i = b; i < e; ++i) { s0 += c0[i]; s1 += c1[i]; }
  part[0][threadIdx.x] = s0; part[1][threadIdx.x] = s1;
  __syncthreads();
  for (int d = 1; d < 1024; d <<= 1) {  // inclusive scan of the per-thread sums (integers: any order is exact)
    const long a0 = (int)threadIdx.x >= d ? part[0][threadIdx.x - d] : 0, a1 = (int)threadIdx.x >= d ? part[1][threadIdx.x - d] : 0;
    __syncthreads();
    part[0][threadIdx.x] += a0; part[1][threadIdx.x] += a1;
    __syncthreads();
  }
  long r0 = part[0][threadIdx.x] - s0, r1 = part[1][threadIdx.x] - s1;
  for (long i = b; i < e; ++i) { o0[i] = r0; r0 += c0[i]; o1[i] = r1; r1 += c1[i]; }
  if (threadIdx.x == 1023) { o0[n] = part[0][1023]; o1[n] = part[1][1023]; totals[0] = part[0][1023]; totals[1] = part[1][1023]; }
}

// pass B: candidate positions and T-move weights, atom-major in quadrature order (the order of the dense table).
// grid = (W, N), block = 64.
template <int PQA_UNIT = 0>  // (a template so that only the units that launch it compile it)
static __global__ __launch_bounds__(64) void k_tm_fill(SysDev S, JastrowState js, TmBuf B, long W) {
  const long w = blockIdx.x;
  const int e = blockIdx.y;
  const int lane = threadIdx.x;
  if (B.cnt[(size_t)e * W + w] == 0) return;
  const double* xw = js.x + (size_t)w * S.nelec * 3;
  const double ex = xw[3 * e], ey = xw[3 * e + 1], ez = xw[3 * e + 2];
  const int nkw = (S.necp + 63) / 64;
  long run = B.off[(size_t)e * W + w];
  for (int kw = 0; kw < nkw; ++kw) {
    unsigned long long m = B.pass[((size_t)e * W + w) * nkw + kw];
    while (m) {
      const int k = kw * 64 + __ffsll((long long)m) - 1;
      m &= m - 1;
      const int ia = S.ecp_atom[k];
      double dx = ex - S.atom_xyz[3 * ia], dy = ey - S.atom_xyz[3 * ia + 1], dz = ez - S.atom_xyz[3 * ia + 2];
      min_image(S, dx, dy, dz);
      const double r = sqrt(dx * dx + dy * dy + dz * dz);
      double v[PQA_MAXCHAN], prob;
      int nch;
      ecp_radial(S, k, r, B.threshold, v, nch, prob);
      const int naip = (nch <= 2) ? 6 : 12;
      if (lane < naip) {
        const double* qd = B.quad + ((nch <= 2) ? 0 : 18) + 3 * lane;
        const double* R = B.rot + ((size_t)e * S.necp + k) * 9;
        const double vx = R[0] * qd[0] + R[1] * qd[1] + R[2] * qd[2];
        const double vy = R[3] * qd[0] + R[4] * qd[1] + R[5] * qd[2];
        const double vz = R[6] * qd[0] + R[7] * qd[1] + R[8] * qd[2];
        const double rix = r * vx, riy = r * vy, riz = r * vz;
        const double cosv = (dx * rix + dy * riy + dz * riz) / (r * sqrt(rix * rix + riy * riy + riz * riz));
        double wt = 0.0;
        for (int c = 0; c < nch - 1; ++c) wt += (exp(-B.tau * (v[c] / prob)) - 1.0) * (2 * c + 1) * legendre_l(c, cosv);
        wt *= 1.0 / naip;
        double* p = B.pts + 3 * (size_t)(run + lane);
        p[0] = (ex - dx) + rix; p[1] = (ey - dy) + riy; p[2] = (ez - dz) + riz;
        B.wgt[run + lane] = wt;
        B.ptw[run + lane] = (int)w;
      }
      run += naip;
    }
  }
}

// Ratios of ALL T-move candidates against the state at the start of the T-move phase, one THREAD per candidate (single
// determinant, two-body Jastrow, real orbitals): the wave-per-walker loop of k_tm_walker spent ~35 us per candidate on
// dependent loads and wave reductions — 2.1 ms of a 19 ms step at 4096 walkers for ~60 candidates per walker — where this
// kernel takes < 1 ns per candidate in aggregate.  The ratios stay valid for a walker until one of its T-moves is accepted
// (0.3 % of the electrons per step); from there on k_tm_walker recomputes that walker's ratios against the updated state.
// mo: [npts][nmo_s] orbital values of this spin's candidates, p_base: their first candidate.
// Psi(candidate p of electron e, walker w) / Psi by ONE thread: the candidate's orbital row against row e of the inverse,
// and the two-body Jastrow sums at the candidate and at the electron's position (u_old < 0: computed here too).
__device__ __forceinline__ double tm_candidate_ratio(const SysDev& S, const SlaterState& st, const JastrowState& js, const TmBuf& B, int s, int e,
                                                    long w, long p, const double* __restrict__ row, int has_slater, int has_jastrow,
                                                    bool have_uold, double u_old) {
  const int n = s ? S.ndn : S.nup, i = e - s * S.nup;
  double ratio = 1.0;
  if (has_slater) {
    const double* Ti = st.T[s] + ((size_t)w * n + i) * n;
    const int* occ = S.det_occ[s];
    double r = 0.0;
    if (S.occ_ident[s] && ((n | S.nmo[s]) & 3) == 0) {  // ground-state occupation: both rows 32 bytes per load (same order of additions)
      for (int k = 0; k < n; k += 4) {
        const double4 a = *reinterpret_cast<const double4*>(row + k), t = *reinterpret_cast<const double4*>(Ti + k);
        r += a.x * t.x; r += a.y * t.y; r += a.z * t.z; r += a.w * t.w;
      }
    } else
      for (int k = 0; k < n; ++k) r += row[occ[k]] * Ti[k];
    ratio = r;
  }
  if (has_jastrow) {
    // the partner / ion sums by jas_eval_lane on the walker-major coordinates (function tables in registers, coordinates several partners
    // ahead; the same terms in the same order as the plain loops this replaced, which read every table entry inside the pair loop)
    const double* xw = js.x + (size_t)w * S.nelec * 3;
    double un = 0.0, uo = 0.0, g_[3], lp_, ee_, ei_;
    // (k_tm_ratio is launched per spin channel: with the merged Pade records the pair's record is a scalar select although the electron and the
    // walker differ between lanes — jas_eval_lane_m ELANE, as in k_ecp_point_lw)
    if (S.jq_on) {
      jas_eval_lane_m<0, true, true, true>(S, xw, 1L, 0L, e, B.pts[3 * p], B.pts[3 * p + 1], B.pts[3 * p + 2], 1, 0, 1, un, g_, lp_, ee_, ei_, -1, true);
      if (!have_uold) jas_eval_lane_m<0, true, true, true>(S, xw, 1L, 0L, e, xw[3 * e], xw[3 * e + 1], xw[3 * e + 2], 1, 0, 1, uo, g_, lp_, ee_, ei_, -1, true);
    } else {
      jas_eval_lane<0, true>(S, xw, 1L, 0L, e, B.pts[3 * p], B.pts[3 * p + 1], B.pts[3 * p + 2], 1, 0, 1, un, g_, lp_, ee_, ei_);
      if (!have_uold) jas_eval_lane<0, true>(S, xw, 1L, 0L, e, xw[3 * e], xw[3 * e + 1], xw[3 * e + 2], 1, 0, 1, uo, g_, lp_, ee_, ei_);
    }
    ratio *= exp(un - (have_uold ? u_old : uo));
  }
  return ratio;
}

// Share c of ngrp of a candidate's sums (k_tm_walker after a walker's first accepted T-move: lanes = (candidate, share), the shares added by
// shuffles): orbital slots k = c, c + ngrp, ... of the determinant dot, partners j and ions I = c, c + ngrp, ... of the Jastrow exponent at the
// candidate (jas_eval_lane on the walker-major coordinates: function tables in registers, coordinates four partners ahead).
__device__ __forceinline__ void tm_ratio_part(const SysDev& S, const SlaterState& st, const JastrowState& js, const TmBuf& B, int s, int e, long w, long p,
                                              const double* __restrict__ row, int c, int ngrp, int has_slater, int has_jastrow, double& rpart,
                                              double& upart) {
  const int n = s ? S.ndn : S.nup, i = e - s * S.nup;
  rpart = 0.0; upart = 0.0;
  if (has_slater) {
    const double* Ti = st.T[s] + ((size_t)w * n + i) * n;
    const int* occ = S.det_occ[s];
    double r = 0.0;
    for (int k = c; k < n; k += ngrp) r += row[occ[k]] * Ti[k];
    rpart = r;
  }
  if (has_jastrow) {
    double g_[3], lp_, ee_, ei_;
    jas_eval_lane<0, true>(S, js.x + (size_t)w * S.nelec * 3, 1L, 0L, e, B.pts[3 * p], B.pts[3 * p + 1], B.pts[3 * p + 2], 1, c, ngrp, upart, g_, lp_, ee_, ei_);
  }
}

// U_e at the CURRENT position of every (electron, walker) that has candidates: -log of the denominator all of its candidates
// share (computed once here instead of once per candidate).  uold: [N][W].  grid = (ceil(W/256), N), block = 256.
template <int PQA_UNIT = 0>  // (a template so that only the units that launch it compile it)
static __global__ __launch_bounds__(256) void k_tm_uold(SysDev S, JastrowState js, TmBuf B, long W, double* __restrict__ uold) {
  const long w = (long)blockIdx.x * 256 + threadIdx.x;
  const int e = blockIdx.y;
  if (w >= W || B.cnt[(size_t)e * W + w] == 0) return;
  const double* xw = js.x + (size_t)w * S.nelec * 3;
  const double ox = xw[3 * e], oy = xw[3 * e + 1], oz = xw[3 * e + 2];
  const int edown = e >= S.nup;
  const double irb = 1.0 / S.rcut_b, ira = 1.0 / S.rcut_a;
  double uo = 0.0;
  for (int j = 0; j < S.nelec; ++j) {
    if (j == e) continue;
    double dx = ox - xw[3 * j], dy = oy - xw[3 * j + 1], dz = oz - xw[3 * j + 2];
    min_image_j(S, dx, dy, dz);
    const double rn = sqrt(dx * dx + dy * dy + dz * dz);
    if (rn < S.rcut_b) {
      const RadShared sh = rad_shared<0>(rn, irb);
      const int col = edown + (j >= S.nup);
      double u = 0.0;
      for (int l = 0; l < S.nb; ++l) {
        double v, g, lp;
        rad_fn<0>(S.b_kind[l], S.b_param[l], S.b_aux[l], S.rcut_b, sh, v, g, lp);
        u += S.bcoeff[l * 3 + col] * v;
      }
      uo += u;
    }
  }
  for (int I = 0; I < S.natom; ++I) {
    double dx = ox - S.atom_xyz[3 * I], dy = oy - S.atom_xyz[3 * I + 1], dz = oz - S.atom_xyz[3 * I + 2];
    min_image_j(S, dx, dy, dz);
    const double rn = sqrt(dx * dx + dy * dy + dz * dz);
    if (rn < S.rcut_a) {
      const RadShared sh = rad_shared<0>(rn, ira);
      double u = 0.0;
      for (int k = 0; k < S.na; ++k) {
        double v, g, lp;
        rad_fn<0>(S.a_kind[k], S.a_param[k], S.a_aux[k], S.rcut_a, sh, v, g, lp);
        u += S.acoeff[(I * S.na + k) * 2 + edown] * v;
      }
      uo += u;
    }
  }
  uold[(size_t)e * W + w] = uo;
}

template <int PQA_UNIT = 0>  // (a template so that only the units that launch it compile it)
static __global__ __launch_bounds__(256) void k_tm_ratio(SysDev S, SlaterState st, JastrowState js, TmBuf B, int s, int has_slater, int has_jastrow,
                                                  const double* __restrict__ mo, long p_base, long npts, long W, const double* __restrict__ uold) {
  const long q = (long)blockIdx.x * 256 + threadIdx.x;
  if (q >= npts) return;
  const long p = p_base + q, w = B.ptw[p];
  int lo = s ? S.nup : 0, hi = (s ? S.nelec : S.nup) - 1;  // the candidate's electron: off is electron-major and ascending
  while (lo < hi) {
    const int mid = (lo + hi + 1) >> 1;
    if (B.off[(size_t)mid * W + w] <= p) lo = mid; else hi = mid - 1;
  }
  const double ratio = tm_candidate_ratio(S, st, js, B, s, lo, w, p, mo + (size_t)q * S.nmo[s], has_slater, has_jastrow, has_jastrow != 0,
                                          has_jastrow ? uold[(size_t)lo * W + w] : 0.0);
  B.rat[p] = ratio;
  B.amp[p] = ratio * B.wgt[p];
}

// The sequential part of the T-move phase, for ALL electrons of one walker in one block: walkers are independent, so
// nothing forces a launch per electron (64 x 2 launches of latency-bound kernels were 8-11 % of a DMC step); a walker has
// candidates for ~2-3 of its 64 electrons, the rest cost one offset compare.  Per electron e with candidates:
//   ratios Psi(candidate)/Psi against the CURRENT inverse and Jastrow state (the walker's earlier T-moves included);
//   heat-bath selection and detailed-balance acceptance of dmc.py:73-120,
//     fwd_q = max(ratio_q weight_q, 0), norm = 1 + sum fwd, move q chosen with probability fwd_q / norm (else stay);
//     backward amplitudes seen from the chosen point: ratio_q weight_q / ratio_sel for the other candidates and
//     weight_sel / ratio_sel for the way back; accept with probability norm / back_norm;
//   for an accepted move the commit (updateinternals with mask, dmc.py:167-168): Sherman-Morrison update with the
//   candidate's orbital VALUE row (already evaluated) and the coordinate.  The gradient / Laplacian rows of the cache are
//   refreshed for all of the step's accepted T-moves at once afterwards (k_tm_cache): nothing reads them in between.
// mo_up / mo_dn: [ncand of the spin][nmo_s] orbital values; tot_up: first spin-down candidate.  grid = W, block = 64.
// CX: complex determinants (pqa_cslater.hpp).  The reference's selection lines order complex amplitudes with `>` / `<` and keep
// 1 / ratio in a real array (dmc.py:83-101), which has no defined meaning; the rule followed — pinned by golden g30, generated
// from the reference with the real part of the ratios handed to those very lines — is amplitudes from Re[Psi(R')/Psi(R)]:
// `rat` below is that real part, the commit is the full complex Sherman-Morrison update.
template <bool CX>
static __global__ __launch_bounds__(64) void k_tm_walker(SysDev S, SlaterState st, JastrowState js, TmBuf B, int has_slater, int has_jastrow,
                                                  const double* __restrict__ mo_up, const double* __restrict__ mo_dn, long tot_up, long W, int precomputed) {
  extern __shared__ double lds[];
  const long w = blockIdx.x;
  const int lane = threadIdx.x;
  double* xw = js.x + (size_t)w * S.nelec * 3;
  bool fresh = precomputed != 0;  // B.rat / B.amp hold this walker's ratios (k_tm_ratio) until one of its T-moves is accepted
  // every electron's candidate range once (lane e: electrons e and e + 64), and a mask of the electrons that have any: read one
  // electron at a time they were two dependent loads per electron, 128 round trips per walker for the ~15 that matter
  long offs[2][2];
  unsigned long long has[2];
#pragma unroll
  for (int hh = 0; hh < 2; ++hh) {
    const int el = lane + 64 * hh;
    offs[hh][0] = el < S.nelec ? B.off[(size_t)el * W + w] : 0;
    offs[hh][1] = el < S.nelec ? B.off[(size_t)el * W + w + 1] : 0;
    has[hh] = __ballot(offs[hh][1] > offs[hh][0]);
  }
  for (int e = 0; e < S.nelec; ++e) {
    if (!((has[e >> 6] >> (e & 63)) & 1ull)) continue;  // (acc was cleared for the whole step)
    const long p0 = __shfl(e < 64 ? offs[0][0] : offs[1][0], e & 63, 64), p1 = __shfl(e < 64 ? offs[0][1] : offs[1][1], e & 63, 64);
    const int n = (int)(p1 - p0);
    const int s = e >= S.nup, nmo = S.nmo[s];
    const double* mo = s ? mo_dn : mo_up;
    const long p_base = s ? tot_up : 0;
    const bool fast = fresh && n <= 64;  // ratios from k_tm_ratio, one candidate per lane
    double U0 = 0.0, g[3], lp;
    if (has_jastrow && !fast) jas_eval<0>(S, xw, e, xw[3 * e], xw[3 * e + 1], xw[3 * e + 2], U0, g, lp, 3, lds + S.j3_off);
    // candidate q's ratio and amplitude stay in lane q's registers (the ratio loop leaves them wave-uniform): the heat-bath
    // sums below used to re-read them from global memory one dependent load at a time, 3 n round trips per electron
    const bool in_regs = n <= 64;
    double my_rat = 1.0, my_amp = 0.0, my_wgt = 0.0;
    if (fast) {
      if (lane < n) { my_rat = B.rat[p0 + lane]; my_amp = B.amp[p0 + lane]; my_wgt = B.wgt[p0 + lane]; }
    } else if (!CX && precomputed && n <= 64) {
      // the walker has accepted a T-move: the ratios of k_tm_ratio are stale.  All of this electron's candidates at once, lanes = (candidate,
      // share of its sums) — one candidate after the other on the whole wave (the loop below: two wave reductions and a dependent chain per
      // candidate) made the walkers with accepted T-moves the tail of the launch, 710 us against 114 for the others
      int ngrp = 1, lg = 0;
      while (ngrp * 2 * n <= 64) { ngrp *= 2; ++lg; }
      const int q = lane >> lg, c = lane & (ngrp - 1);
      const long p = p0 + (q < n ? q : 0);
      double rpart, upart;
      tm_ratio_part(S, st, js, B, s, e, w, p, mo + (size_t)(p - p_base) * nmo, c, ngrp, has_slater, has_jastrow, rpart, upart);
      for (int o = 1; o < ngrp; o <<= 1) { rpart += __shfl_xor(rpart, o, 64); upart += __shfl_xor(upart, o, 64); }
      double rat = has_slater ? rpart : 1.0;
      if (has_jastrow) rat *= exp(upart - U0);
      const double ratq = __shfl(rat, (lane < n ? lane : 0) << lg, 64);  // candidate q's ratio to lane q
      if (lane < n) {
        my_rat = ratq; my_wgt = B.wgt[p0 + lane]; my_amp = ratq * my_wgt;
        B.rat[p0 + lane] = ratq; B.amp[p0 + lane] = my_amp;
      }
    } else
    for (long p = p0; p < p1; ++p) {
      double rat = 1.0;
      if (has_slater) {
        if (CX) {
          cx r1[1];
          slater_ratios_c<1>(S, st, s, e - s * S.nup, w, mo + (size_t)(p - p_base) * nmo, r1, lds);
          rat = r1[0].r;  // (the Jastrow ratio below is real and positive: Re[ratio] = Re[determinant ratio] * exp(dU))
        } else {
          double r1[1];
          slater_ratios<1>(S, st, s, e - s * S.nup, w, mo + (size_t)(p - p_base) * nmo, r1, lds);
          rat = r1[0];
        }
      }
      if (has_jastrow) {
        double U;
        jas_eval<0>(S, xw, e, B.pts[3 * p], B.pts[3 * p + 1], B.pts[3 * p + 2], U, g, lp, 3, lds + S.j3_off);
        rat *= exp(U - U0);
      }
      const double wg = B.wgt[p];
      if (lane == (int)(p - p0)) { my_rat = rat; my_wgt = wg; my_amp = rat * wg; }
      if (lane == 0) { B.rat[p] = rat; B.amp[p] = rat * wg; }
    }
    // values of candidate q as seen by lane 0: register of lane q (or global memory for more than 64 candidates)
    auto amp_of = [&](int q) { return in_regs ? __shfl(my_amp, q, 64) : B.amp[p0 + q]; };
    auto rat_of = [&](int q) { return in_regs ? __shfl(my_rat, q, 64) : B.rat[p0 + q]; };
    auto wgt_of = [&](int q) { return in_regs ? __shfl(my_wgt, q, 64) : B.wgt[p0 + q]; };
    int sel = n, acc = 0;
    if (in_regs) {
      // the sums run in candidate order as before (same additions, same order); the terms — a division per candidate in the
      // selection, a product in the way back — are formed by the candidates' own lanes at once instead of one per iteration
      const double my_fwd = fmax(my_amp, 0.0);
      double norm = 1.0;
      for (int q = 0; q < n; ++q) norm += __shfl(my_fwd, q, 64);
      double u1, u2;
      if (B.u1) { u1 = B.u1[(size_t)e * W + w]; u2 = B.u2[(size_t)e * W + w]; }
      else {
        const Philox a = philox(B.seed, (uint32_t)w, (uint32_t)e, PQA_STREAM_TM_U1, B.step);
        const Philox b = philox(B.seed, (uint32_t)w, (uint32_t)e, PQA_STREAM_TM_U2, B.step);
        u1 = u01(a.c[0], a.c[1]); u2 = u01(b.c[0], b.c[1]);
      }
      const double my_t = my_fwd / norm;
      sel = 0;
      double cdf = 0.0;
      for (int q = 0; q < n; ++q) {
        cdf += __shfl(my_t, q, 64);
        if (cdf < u1) ++sel;
      }
      if (sel < n) {
        const double rr = 1.0 / __shfl(my_rat, sel, 64);
        const double my_b = fmax((lane == sel) ? rr * my_wgt : my_amp * rr, 0.0);
        double back = 1.0;
        for (int q = 0; q < n; ++q) back += __shfl(my_b, q, 64);
        acc = norm / back > u2;
      }
      if (lane == 0) B.acc[(size_t)e * W + w] = acc;
    } else
    {  // (all lanes run the loops so that the shuffles are convergent; lane 0's results are the ones used)
      double norm = 1.0;
      for (int q = 0; q < n; ++q) norm += fmax(amp_of(q), 0.0);
      double u1, u2;
      if (B.u1) { u1 = B.u1[(size_t)e * W + w]; u2 = B.u2[(size_t)e * W + w]; }
      else {
        const Philox a = philox(B.seed, (uint32_t)w, (uint32_t)e, PQA_STREAM_TM_U1, B.step);
        const Philox b = philox(B.seed, (uint32_t)w, (uint32_t)e, PQA_STREAM_TM_U2, B.step);
        u1 = u01(a.c[0], a.c[1]); u2 = u01(b.c[0], b.c[1]);
      }
      sel = 0;
      double cdf = 0.0;
      for (int q = 0; q < n; ++q) {
        cdf += fmax(amp_of(q), 0.0) / norm;
        if (cdf < u1) ++sel;
      }
      if (sel < n) {
        const double rr = 1.0 / rat_of(sel);
        double back = 1.0;
        for (int q = 0; q < n; ++q) back += fmax((q == sel) ? rr * wgt_of(q) : amp_of(q) * rr, 0.0);
        acc = norm / back > u2;
      }
      if (lane == 0) B.acc[(size_t)e * W + w] = acc;
    }
    acc = __shfl(acc, 0, 64);
    sel = __shfl(sel, 0, 64);
    if (!acc) continue;
    fresh = false;
    __syncthreads();
    if (has_slater) {
      if (CX) sm_update_wave_c(S, st, s, e - s * S.nup, w, mo + (size_t)(p0 + sel - p_base) * nmo, lds);
      else sm_update_wave(S, st, s, e - s * S.nup, w, mo + (size_t)(p0 + sel - p_base) * nmo, lds);
    }
    if (lane == 0) {
      // compute_tmoves folds the candidates (eval_ecp.py:66 make_irreducible) and propose_tmoves then takes only their
      // folded coordinates (dmc.py:100), so the reference's wrap counters do not see a T-move across the cell boundary;
      // identical results means the same here: fold, and leave the counters alone.
      double nx = B.pts[3 * (p0 + sel)], ny = B.pts[3 * (p0 + sel) + 1], nz = B.pts[3 * (p0 + sel) + 2];
      if (!B.nofold) fold_cell(S, nx, ny, nz);
      xw[3 * e] = nx; xw[3 * e + 1] = ny; xw[3 * e + 2] = nz;
    }
    __syncthreads();  // the next electron's ratios see the new coordinate and inverse
  }
}

// ascending list of the (electron, walker) pairs whose T-move was accepted in this step, with their (new) positions:
// entry acc_off[i] of the list for every flagged i = e*W + w.  grid = ceil(N*W/256), block = 256.
template <int PQA_UNIT = 0>  // (a template so that only the units that launch it compile it)
static __global__ __launch_bounds__(256) void k_tm_gather(TmBuf B, const double* __restrict__ x, int nelec, long W) {
  const long i = (long)blockIdx.x * 256 + threadIdx.x;
  if (i >= (long)nelec * W || !B.acc[i]) return;
  const long el = i / W, w = i - el * W, a = B.acc_off[i];
  const double* xe = x + ((size_t)w * nelec + el) * 3;
  B.acc_idx[a] = (int)i;
  B.acc_pos[3 * a] = xe[0]; B.acc_pos[3 * a + 1] = xe[1]; B.acc_pos[3 * a + 2] = xe[2];
}

// orbital-row cache (value, gradient, Laplacian) of the accepted T-moves of one spin.  mo5: [count][5][nmo_s] rows at the
// new positions, idx: the matching (e*W + w) entries.  rc / sel: the two-slot row cache of the lane-per-walker sweep when that
// holds the live cache (the row replaces the CURRENT slot's), else NULL (st.cache).  grid = count, block = 64.
template <int PQA_UNIT = 0>  // (a template so that only the units that launch it compile it)
static __global__ __launch_bounds__(64) void k_tm_cache(SysDev S, SlaterState st, const int* __restrict__ idx, const double* __restrict__ mo5,
                                                 int s, long W, double* __restrict__ rc, const uint8_t* __restrict__ sel) {
  const long a = blockIdx.x;
  const long i = idx[a];
  const int e = (int)(i / W);
  const long w = i - (long)e * W;
  const int n = s ? S.ndn : S.nup, nmo = S.nmo[s], ie = e - s * S.nup;
  const double* row = mo5 + (size_t)a * 5 * nmo;
  double* c = rc ? rc + (((size_t)ie * 2 + sel[(size_t)ie * W + w]) * W + w) * 5 * nmo : st.cache[s] + ((size_t)w * n + ie) * 5 * nmo;
  for (int k = threadIdx.x; k < 5 * nmo; k += 64) c[k] = row[k];
}

// ---------------------------------------------------------------- weights and averages
__device__ __forceinline__ double dmc_S(double e_trial, double e_est, double branchcut, double v2, double tau, double eloc, int nelec) {
  const double d = e_est - eloc;  // dmc.py:224-235
  const double e_cut = fmin(fmax(d, -branchcut), branchcut);
  const double t = v2 * tau / nelec;
  return e_trial - e_est + e_cut / sqrt(1.0 + t * t);
}

// weights *= exp(tau (r2_acc / r2_prop) (S_new + S_old)/2); the new energy becomes the old one; statistics reset.
// en: rows ke, ee, ei, ecp, grad2, total of the NEW configuration.  eold/v2old: [W].
template <int PQA_UNIT = 0>  // (a template so that only the units that launch it compile it)
static __global__ __launch_bounds__(256) void k_dmc_weights(const double* __restrict__ en, double* __restrict__ eold, double* __restrict__ v2old,
                                                     double* __restrict__ r2_acc, double* __restrict__ r2_prop,
                                                     double* __restrict__ weights, double tau, double branchcut, double e_trial,
                                                     double e_est, int nelec, long W) {
  const long w = (long)blockIdx.x * 256 + threadIdx.x;
  if (w >= W) return;
  const double eloc = en[5 * W + w], v2 = en[4 * W + w];
  const double Sm = 0.5 * (dmc_S(e_trial, e_est, branchcut, v2, tau, eloc, nelec) +
                           dmc_S(e_trial, e_est, branchcut, v2old[w], tau, eold[w], nelec));
  weights[w] *= exp(tau * (r2_acc[w] / r2_prop[w]) * Sm);
  eold[w] = eloc; v2old[w] = v2;
  r2_acc[w] = 0.0; r2_prop[w] = 0.0;
}

template <int PQA_UNIT = 0>  // (a template so that only the units that launch it compile it)
static __global__ __launch_bounds__(256) void k_dmc_keep(const double* __restrict__ en, double* __restrict__ eold, double* __restrict__ v2old, long W) {
  const long w = (long)blockIdx.x * 256 + threadIdx.x;
  if (w >= W) return;
  eold[w] = en[5 * W + w]; v2old[w] = en[4 * W + w];
}

// out[0..5] = sum_w weights[w] en[k][w] / sum_w weights[w] (the reference's dot(weights, v)/(W wavg), dmc.py:205-209),
// out[6] = mean weight; complex wave functions (nrow = 7: the energy buffer's row 6 is Im ecp = Im total) also out[7] = the
// weighted mean of that row.  One block of 1024 threads, deterministic.
template <int PQA_UNIT = 0>  // (a template so that only the units that launch it compile it)
static __global__ __launch_bounds__(1024) void k_dmc_averages(const double* __restrict__ en, const double* __restrict__ weights, long W,
                                                       double* __restrict__ out, int nrow) {
  __shared__ double part[8][1024];
  double s[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  for (long i = threadIdx.x; i < W; i += 1024) {
    const double wt = weights[i];
#pragma unroll
    for (int k = 0; k < 6; ++k) s[k] += wt * en[(size_t)k * W + i];
    s[6] += wt;
    if (nrow > 6) s[7] += wt * en[(size_t)6 * W + i];
  }
#pragma unroll
  for (int k = 0; k < 8; ++k) part[k][threadIdx.x] = s[k];
  __syncthreads();
  for (int off = 512; off > 0; off >>= 1) {
    if ((int)threadIdx.x < off)
#pragma unroll
      for (int k = 0; k < 8; ++k) part[k][threadIdx.x] += part[k][threadIdx.x + off];
    __syncthreads();
  }
  if (threadIdx.x < 6) out[threadIdx.x] = part[threadIdx.x][0] / part[6][0];
  if (threadIdx.x == 6) out[6] = part[6][0] / (double)W;
  if (threadIdx.x == 7 && nrow > 6) out[7] = part[7][0] / part[6][0];
}
