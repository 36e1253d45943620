// Slater-determinant kernels: one walker (and one unique spin determinant) per wavefront,
// inverse tile staged in LDS.
//
// Reference semantics (pyqmc/wf/slater.py): recompute :227-260 (slogdet + inv per unique
// determinant), value via determinant_tools.compute_value :74-88, row ratios
// _testrow/_testrowderiv :301-380, Sherman-Morrison sherman_morrison_ms :88-94 inside
// updateinternals :262-291.
//
// Storage differs from the reference on purpose: the inverse is kept ELECTRON-major,
//   T[w][d][i][k] = inverse[w][d][k][i]   (k orbital slot, i electron),
// so the column of the inverse needed for a ratio of electron i is one contiguous 8n-byte row.
#pragma once
#include <float.h>
#include "pqa_common.hpp"
inline namespace PQA_SYNC_NS {  // (PQA_WSYNC flavour: pqa_common.hpp)

struct SlaterState {
  double* T[2];      // [W][ndet_s][n_s][n_s]
  double* dsign[2];  // [W][ndet_s]
  double* dlog[2];   // [W][ndet_s]
  double* cache[2];  // [W][n_s][5][nmo_s]  MO value, grad, laplacian of every electron at its current position
};

__device__ __forceinline__ double clamp_nan_to_num(double v) {
  if (v != v) return 0.0;
  if (v > DBL_MAX) return DBL_MAX;
  if (v < -DBL_MAX) return -DBL_MAX;
  return v;
}

// ---------------------------------------------------------------- build + invert (recompute)
// grid = W * ndet_s blocks of 64 threads; dynamic LDS: n*(n+1) doubles + n ints.
// B[j][i] = mo(electron i, orbital occ[d][j]); T = B^{-1}, det(B) = det(reference matrix).
template <int PQA_UNIT = 0>  // (a template so that only the units that launch it compile it)
static __global__ __launch_bounds__(64) void k_build_invert(SysDev S, SlaterState st, int s, long W) {
  extern __shared__ double lds[];
  const int n = s ? S.ndn : S.nup, nmo = S.nmo[s], D = S.ndet_s[s];
  if (n == 0) return;
  const long w = blockIdx.x / D;
  const int d = blockIdx.x % D;
  const int lane = threadIdx.x, ld = n + 1;
  double* M = lds;
  int* perm = (int*)(lds + (size_t)n * ld);
  const int* occ = S.det_occ[s] + (size_t)d * n;
  const double* cw = st.cache[s] + (size_t)w * n * 5 * nmo;
  for (int idx = lane; idx < n * n; idx += 64) {
    const int i = idx / n, j = idx % n;  // coalesced over j within an electron's row
    M[j * ld + i] = cw[(size_t)i * 5 * nmo + occ[j]];
  }
  __syncthreads();
  double sign = 1.0, logd = 0.0;
  bool singular = false;
  // (a lane owns the rows / columns lane, lane + 64, ...: one each up to 64 electrons, two up to 128)
  for (int k = 0; k < n; ++k) {
    double v = -1.0;
    int idx = lane;
    for (int r = lane; r < n; r += 64) {
      const double c = (r >= k) ? fabs(M[r * ld + k]) : -1.0;
      if (c > v) { v = c; idx = r; }  // (ascending r: the lowest index among equals, like the butterfly below)
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
      const double ov = __shfl_xor(v, off, 64);
      const int oi = __shfl_xor(idx, off, 64);
      if (ov > v || (ov == v && oi < idx)) { v = ov; idx = oi; }
    }
    const int p = idx;
    if (!(v > 0.0) || !(v <= DBL_MAX)) { singular = true; break; }
    if (p != k)
      for (int c = lane; c < n; c += 64) {
        const double t = M[k * ld + c];
        M[k * ld + c] = M[p * ld + c];
        M[p * ld + c] = t;
      }
    if (p != k) sign = -sign;
    if (lane == 0) perm[k] = p;
    __syncthreads();
    const double piv = M[k * ld + k];
    logd += log(fabs(piv));
    if (piv < 0.0) sign = -sign;
    __syncthreads();
    double rk[PQA_MAXN / 64];
#pragma unroll
    for (int q = 0; q < PQA_MAXN / 64; ++q) {
      const int c = lane + 64 * q;
      rk[q] = 0.0;
      if (c < n) {
        rk[q] = (c == k) ? 1.0 / piv : M[k * ld + c] / piv;
        M[k * ld + c] = rk[q];
      }
    }
    for (int r = 0; r < n; ++r) {
      if (r == k) continue;
      const double f = M[r * ld + k];
      __builtin_amdgcn_wave_barrier();  // (n > 64: column k belongs to one lane, read by all before that lane rewrites it)
#pragma unroll
      for (int q = 0; q < PQA_MAXN / 64; ++q) {
        const int c = lane + 64 * q;
        if (c < n) {
          const double cur = (c == k) ? 0.0 : M[r * ld + c];
          M[r * ld + c] = cur - f * rk[q];
        }
      }
    }
    __syncthreads();
  }
  double* Tw = st.T[s] + ((size_t)w * D + d) * n * n;
  if (singular) {  // slater.py:246-258: inverse left at zero where logdet is not finite
    for (int idx = lane; idx < n * n; idx += 64) Tw[idx] = 0.0;
    if (lane == 0) { st.dsign[s][w * D + d] = 0.0; st.dlog[s][w * D + d] = -INFINITY; }
    return;
  }
  for (int k = n - 1; k >= 0; --k) {  // undo the row pivoting: column swaps in reverse order
    const int p = perm[k];
    if (p != k)
      for (int r = lane; r < n; r += 64) {
        const double t = M[r * ld + k];
        M[r * ld + k] = M[r * ld + p];
        M[r * ld + p] = t;
      }
    __syncthreads();
  }
  for (int idx = lane; idx < n * n; idx += 64) Tw[idx] = M[(idx / n) * ld + idx % n];
  if (lane == 0) { st.dsign[s][w * D + d] = sign; st.dlog[s][w * D + d] = logd; }
}

// ---------------------------------------------------------------- multi-determinant bookkeeping
// weight of full determinant Dd for walker w, relative to exp(ref): c_D s_up s_dn exp(l_up + l_dn - ref)
__device__ __forceinline__ double det_logsum(const SysDev& S, const SlaterState& st, long w, int Dd) {
  return st.dlog[0][w * S.ndet_s[0] + S.det_map[Dd]] + st.dlog[1][w * S.ndet_s[1] + S.det_map[S.ndet + Dd]];
}
__device__ __forceinline__ double det_ref(const SysDev& S, const SlaterState& st, long w) {
  double m = -INFINITY;
  for (int Dd = threadIdx.x & 63; Dd < S.ndet; Dd += 64) m = fmax(m, det_logsum(S, st, w, Dd));
  return wave_max(m);
}
__device__ __forceinline__ double det_weight(const SysDev& S, const SlaterState& st, long w, int Dd, double ref) {
  const double su = st.dsign[0][w * S.ndet_s[0] + S.det_map[Dd]];
  const double sd = st.dsign[1][w * S.ndet_s[1] + S.det_map[S.ndet + Dd]];
  const double l = det_logsum(S, st, w, Dd);
  const double ex = (l == -INFINITY) ? 0.0 : exp(l - ref);
  return S.det_coeff[Dd] * su * sd * ex;
}

// (sign, log|Psi_S|) of one walker — determinant_tools.compute_value (:74-88)
__device__ __forceinline__ void slater_value_wave(const SysDev& S, const SlaterState& st, long w, double& sign,
                                                  double& logv) {
  const double ref = det_ref(S, st, w);
  double t = 0.0;
  for (int Dd = threadIdx.x & 63; Dd < S.ndet; Dd += 64) t += det_weight(S, st, w, Dd, ref);
  t = wave_sum(t);
  sign = clamp_nan_to_num(t / fabs(t));
  logv = clamp_nan_to_num(log(fabs(t)) + ref);
}

template <int PQA_UNIT = 0>  // (a template so that only the units that launch it compile it)
static __global__ __launch_bounds__(64) void k_slater_value(SysDev S, SlaterState st, double* sign, double* logv) {
  const long w = blockIdx.x;
  double sg, lv;
  slater_value_wave(S, st, w, sg, lv);
  if (threadIdx.x == 0) { sign[w] = sg; logv[w] = lv; }
}

// Ratios (new row)/(current) of electron i (index within spin s) for NCOMP stacked rows
// mo[c][nmo] — slater.py:301-380.  scratch: >= ndet_s*NCOMP doubles of LDS (multi-det only).
template <int NCOMP>
__device__ __forceinline__ void slater_ratios(const SysDev& S, const SlaterState& st, int s, int i, long w,
                                              const double* __restrict__ mo, double (&out)[NCOMP], double* scratch) {
  const int lane = threadIdx.x & 63;
  const int n = s ? S.ndn : S.nup, nmo = S.nmo[s], D = S.ndet_s[s];
  if (S.ndet == 1) {
    double part[NCOMP];
#pragma unroll
    for (int c = 0; c < NCOMP; ++c) part[c] = 0.0;
    const double* Trow = st.T[s] + ((size_t)w * n + i) * n;
    const int* occ = S.det_occ[s];
    for (int j = lane; j < n; j += 64) {
      const double t = Trow[j];
      const int o = occ[j];
#pragma unroll
      for (int c = 0; c < NCOMP; ++c) part[c] += mo[c * nmo + o] * t;
    }
#pragma unroll
    for (int c = 0; c < NCOMP; ++c) out[c] = wave_sum(part[c]);
    return;
  }
  if (n <= 16) {
    // Few electrons per spin (the 50-determinant water molecule has 4): one determinant per pass left 60 of the 64 lanes idle and
    // cost a wave-wide reduction per determinant and component — k_ecp_accum walked ~30 unique determinants x 36 points one after
    // the other (0.9 ms per evaluation at 2 048 walkers).  Lanes = (determinant of the pass, slot): 64 / GS determinants at a
    // time, a butterfly over the GS = 2^k >= n lanes of a determinant.  It adds the same pairs as wave_sum's scan does for lanes
    // 0 .. n-1 (adjacent pairs, then pairs of pairs; the other lanes hold zeros there): the ratios are bitwise the same.
    const int GS = n <= 1 ? 1 : (n <= 2 ? 2 : (n <= 4 ? 4 : (n <= 8 ? 8 : 16)));
    const int DP = 64 / GS, g = lane / GS, j = lane & (GS - 1);
#pragma unroll 2
    for (int d0 = 0; d0 < D; d0 += DP) {  // (passes are independent: two in flight)
      const int d = d0 + g;
      const bool act = d < D && j < n;
      double part[NCOMP];
#pragma unroll
      for (int c = 0; c < NCOMP; ++c) part[c] = 0.0;
      if (act) {
        const double t = st.T[s][(((size_t)w * D + d) * n + i) * n + j];
        const int o = S.det_occ[s][(size_t)d * n + j];
#pragma unroll
        for (int c = 0; c < NCOMP; ++c) part[c] += mo[c * nmo + o] * t;
      }
      for (int off = 1; off < GS; off <<= 1) {
#pragma unroll
        for (int c = 0; c < NCOMP; ++c) part[c] += __shfl_xor(part[c], off, 64);
      }
      if (act && j == 0) {
#pragma unroll
        for (int c = 0; c < NCOMP; ++c) scratch[d * NCOMP + c] = part[c];
      }
    }
  } else
  for (int d = 0; d < D; ++d) {
    const double* Trow = st.T[s] + (((size_t)w * D + d) * n + i) * n;
    const int* occ = S.det_occ[s] + (size_t)d * n;
    double part[NCOMP];
#pragma unroll
    for (int c = 0; c < NCOMP; ++c) part[c] = 0.0;
    for (int j = lane; j < n; j += 64) {
      const double t = Trow[j];
      const int o = occ[j];
#pragma unroll
      for (int c = 0; c < NCOMP; ++c) part[c] += mo[c * nmo + o] * t;
    }
#pragma unroll
    for (int c = 0; c < NCOMP; ++c) {
      const double r = wave_sum(part[c]);
      if (lane == 0) scratch[d * NCOMP + c] = r;
    }
  }
  PQA_WSYNC();
  const double ref = det_ref(S, st, w);
  double num[NCOMP], den = 0.0;
#pragma unroll
  for (int c = 0; c < NCOMP; ++c) num[c] = 0.0;
  for (int Dd = lane; Dd < S.ndet; Dd += 64) {
    const double wt = det_weight(S, st, w, Dd, ref);
    const int ds = S.det_map[s * S.ndet + Dd];
    den += wt;
#pragma unroll
    for (int c = 0; c < NCOMP; ++c) num[c] += wt * scratch[ds * NCOMP + c];
  }
  den = wave_sum(den);
#pragma unroll
  for (int c = 0; c < NCOMP; ++c) out[c] = wave_sum(num[c]) / den;
  PQA_WSYNC();
}

// out (NCOMP, nrow*npt); mo rows [(r*npt+q)][NCOMP][nmo]; dynamic LDS: max(ndet_s)*NCOMP doubles
template <int NCOMP>
static __global__ __launch_bounds__(64) void k_slater_eval(SysDev S, SlaterState st, int e, const double* __restrict__ mo,
                                                    long nrow, int npt, const int* __restrict__ widx,
                                                    double* __restrict__ out) {
  extern __shared__ double lds[];
  const long r = blockIdx.x;
  const long w = widx ? widx[r] : r;
  const int s = e >= S.nup, i = e - s * S.nup, nmo = S.nmo[s];
  for (int q = 0; q < npt; ++q) {
    double rat[NCOMP];
    slater_ratios<NCOMP>(S, st, s, i, w, mo + ((size_t)(r * npt + q) * NCOMP) * nmo, rat, lds);
    if (threadIdx.x == 0)
#pragma unroll
      for (int c = 0; c < NCOMP; ++c) out[(size_t)c * nrow * npt + r * npt + q] = rat[c];
  }
}

// ---------------------------------------------------------------- Sherman-Morrison
// Replace the row of electron i by the orbitals `morow` (value component, [nmo]) for every unique
// determinant of spin s of walker w.  slater.py:88-94, :286-291.
//   tmp[j]   = sum_k vec[k] inv[k][j]          (j = electron)       -> sum_k V[k] T[j][k]
//   ratio    = tmp[i]
//   T[j][k] -= (T[i][k] / ratio) tmp[j]  (j != i),   T[i][k] = T[i][k] / ratio
// The n x n tile is staged global -> LDS with coalesced 512-B wave accesses (row stride n+1, odd, so
// the row-per-lane accesses below are bank-conflict free); every row is split over R = 64/n lane
// groups so all 64 lanes work for n <= 32.  LDS: n(n+1) + 2n + 64 doubles.
// dpart / dparts: this wave updates the passes dpart, dpart + dparts, ... of the small-determinant branch (n <= 8; the fused
// wave-per-walker sweep deals a walker's determinants to its three waves); elsewhere every determinant (dparts 1).
__device__ __forceinline__ void sm_update_wave(const SysDev& S, const SlaterState& st, int s, int i, long w,
                                               const double* __restrict__ morow, double* lds, int dpart = 0, int dparts = 1) {
  const int lane = threadIdx.x & 63;
  const int n = s ? S.ndn : S.nup, D = S.ndet_s[s], ld = n + 1;
  if (n <= 8) {
    // Small determinants (the 50-determinant molecule: 4 x 4, ~30 unique per spin): the staged update below is five block barriers
    // per determinant, one determinant at a time — most of k_accept's 77 us.  Here a lane holds ONE element T[r][c] of a determinant
    // and 64 / GS^2 determinants (GS = 2^k >= n) go through at once; the row dots, the ratio and the pivot row travel by shuffles.
    // Same products, same order of additions as below (for n <= 8 every lane group there holds one column: the row dot is the
    // sequential sum of the n rounded products).
    const int GS = n <= 1 ? 1 : (n <= 2 ? 2 : (n <= 4 ? 4 : 8)), E = GS * GS, DP = 64 / E;
    const int g = lane / E, r = (lane & (E - 1)) / GS, c = lane & (GS - 1);
#pragma unroll 2
    for (int d0 = dpart * DP; d0 < D; d0 += DP * dparts) {  // (passes are independent: two in flight)
      const int d = d0 + g;
      const bool act = d < D && r < n && c < n;
      double* Tw = st.T[s] + ((size_t)w * D + (d < D ? d : 0)) * n * n;
      double t = act ? Tw[r * n + c] : 0.0;
      const double v = act ? morow[S.det_occ[s][(size_t)d * n + c]] : 0.0;
      const double p = v * t;
      double tmp = 0.0;
      for (int k = 0; k < GS; ++k) tmp += __shfl(p, g * E + r * GS + k, 64);
      const double ratio = __shfl(tmp, g * E + i * GS, 64);
      const double rr = __shfl(t, g * E + i * GS + c, 64) / ratio;  // inv_ratio[c] = inv[c][i] / ratio
      if (r == i) t = rr;
      else t -= rr * tmp;
      if (act) Tw[r * n + c] = t;
      if (d < D && r == 0 && c == 0) {
        const size_t o = (size_t)w * D + d;
        st.dsign[s][o] *= (ratio > 0.0) ? 1.0 : ((ratio < 0.0) ? -1.0 : ratio);  // np.sign (0 and nan propagate)
        st.dlog[s][o] += log(fabs(ratio));
      }
    }
    return;
  }
  if (dpart != 0) return;  // (the branches below work on one determinant at a time through the wave's LDS tile: one wave does them all)
  if (n > PQA_MAXN_FAST) {
    // More than 64 electrons of this spin: the tile does not fit the staging scheme below (one lane per row, n (n + 1) doubles of
    // LDS), so the update runs on the inverse where it lies: one wave sum per row for tmp, then the rank-1 update row by row.
    // LDS: 3 n doubles.  Same formulas (slater.py:88-94); the row dot is a wave sum over lanes that hold columns lane, lane + 64.
    double* V = lds;
    double* TMP = lds + n;
    double* Rr = TMP + n;
    for (int d = 0; d < D; ++d) {
      double* Tw = st.T[s] + ((size_t)w * D + d) * n * n;
      const int* occ = S.det_occ[s] + (size_t)d * n;
      for (int k = lane; k < n; k += 64) V[k] = morow[occ[k]];
      PQA_WSYNC();
      for (int j = 0; j < n; ++j) {
        double p = 0.0;
        for (int k = lane; k < n; k += 64) p += V[k] * Tw[(size_t)j * n + k];
        p = wave_sum(p);
        if (lane == 0) TMP[j] = p;
      }
      PQA_WSYNC();
      const double ratio = TMP[i];
      for (int k = lane; k < n; k += 64) Rr[k] = Tw[(size_t)i * n + k] / ratio;
      PQA_WSYNC();
      for (int j = 0; j < n; ++j) {
        const double tj = TMP[j];
        if (j == i) { for (int k = lane; k < n; k += 64) Tw[(size_t)j * n + k] = Rr[k]; }
        else { for (int k = lane; k < n; k += 64) Tw[(size_t)j * n + k] -= Rr[k] * tj; }
      }
      if (lane == 0) {
        const size_t o = (size_t)w * D + d;
        st.dsign[s][o] *= (ratio > 0.0) ? 1.0 : ((ratio < 0.0) ? -1.0 : ratio);
        st.dlog[s][o] += log(fabs(ratio));
      }
      PQA_WSYNC();
    }
    return;
  }
  double* L = lds;
  double* V = lds + (size_t)n * ld;
  double* Rr = V + n;
  double* Pt = Rr + n;  // [R][n] partial dot products
  const int R = (n >= 64) ? 1 : 64 / n;          // lane groups per row
  const int chunk = (n + R - 1) / R;             // columns per lane group
  const int part = lane / n, row = lane - part * n;
  const bool active = part < R;
  const int kb = part * chunk, ke = (kb + chunk < n) ? kb + chunk : n;
  const int di = 64 / n, dk = 64 - di * n;       // (row, col) increment of a 64-element stride
  for (int d = 0; d < D; ++d) {
    double* Tw = st.T[s] + ((size_t)w * D + d) * n * n;
    const int* occ = S.det_occ[s] + (size_t)d * n;
    {
      int r = part, c = row;  // element index lane = r*n + c
      for (int idx = lane; idx < n * n; idx += 64) {
        L[r * ld + c] = Tw[idx];
        r += di; c += dk;
        if (c >= n) { c -= n; ++r; }
      }
    }
    for (int k = lane; k < n; k += 64) V[k] = morow[occ[k]];
    PQA_WSYNC();
    double tmp = 0.0;
    if (active) {
      const double* Lr = L + row * ld;
      for (int k = kb; k < ke; ++k) tmp += V[k] * Lr[k];
      Pt[part * n + row] = tmp;
    }
    PQA_WSYNC();
    tmp = 0.0;
    if (active)
      for (int q = 0; q < R; ++q) tmp += Pt[q * n + row];  // same order in every lane group: bitwise equal
    const double ratio = __shfl(tmp, i, 64);               // lane i is (part 0, row i)
    if (lane < n) Rr[lane] = L[i * ld + lane] / ratio;     // inv_ratio[k] = inv[k][i] / ratio
    PQA_WSYNC();
    if (active) {
      double* Lr = L + row * ld;
      if (row == i) {
        for (int k = kb; k < ke; ++k) Lr[k] = Rr[k];
      } else {
        for (int k = kb; k < ke; ++k) Lr[k] -= Rr[k] * tmp;
      }
    }
    PQA_WSYNC();
    {
      int r = part, c = row;
      for (int idx = lane; idx < n * n; idx += 64) {
        Tw[idx] = L[r * ld + c];
        r += di; c += dk;
        if (c >= n) { c -= n; ++r; }
      }
    }
    if (lane == 0) {
      const size_t o = (size_t)w * D + d;
      st.dsign[s][o] *= (ratio > 0.0) ? 1.0 : ((ratio < 0.0) ? -1.0 : ratio);  // np.sign (0 and nan propagate)
      st.dlog[s][o] += log(fabs(ratio));
    }
    PQA_WSYNC();
  }
}

// grid = W; mo rows [w][NCOMP_STRIDE][nmo] (value component first); mask (W) bytes
template <int PQA_UNIT = 0>  // (a template so that only the units that launch it compile it)
static __global__ __launch_bounds__(64) void k_sm_update(SysDev S, SlaterState st, int e, const double* __restrict__ mo,
                                                  int row_stride, const uint8_t* __restrict__ mask, int to_cache) {
  extern __shared__ double lds[];
  const long w = blockIdx.x;
  if (mask && !mask[w]) return;
  const int s = e >= S.nup, i = e - s * S.nup, n = s ? S.ndn : S.nup, nmo = S.nmo[s];
  const double* row = mo + (size_t)w * row_stride;
  sm_update_wave(S, st, s, i, w, row, lds);
  if (to_cache) {  // keep the per-electron orbital cache current (value, grad, lap rows)
    double* c = st.cache[s] + ((size_t)w * n + i) * 5 * nmo;
    for (int k = threadIdx.x; k < 5 * nmo; k += 64) c[k] = row[k];
  }
}

// any non-finite log-determinant? (slater.py:269-275 trigger for a full recompute)
template <int PQA_UNIT = 0>  // (a template so that only the units that launch it compile it)
static __global__ void k_has_zero(const double* dlog, long count, int* flag) {
  const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx < count) {
    const double v = dlog[idx];
    if (!(v >= -DBL_MAX && v <= DBL_MAX)) atomicOr(flag, 1);
  }
}

// ---------------------------------------------------------------- parameter gradients (Slater.pgradient, slater.py:462-542)
// d_det[w][di] = D_up D_dn / Psi for determinant di (:495-505).
template <int PQA_UNIT = 0>  // (a template so that only the units that launch it compile it)
static __global__ void k_pgrad_det(SysDev S, SlaterState st, const double* __restrict__ psi_sign, const double* __restrict__ psi_log,
                            long W, double* __restrict__ out) {
  const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= W * S.ndet) return;
  const long w = idx / S.ndet;
  const int di = (int)(idx % S.ndet), u0 = S.det_map[di], u1 = S.det_map[S.ndet + di];
  const double sg = psi_sign[w];
  double v = 0.0;
  if (sg != 0.0) {
    double sgn = 1.0, lg = 0.0;
    if (S.nup > 0) { sgn *= st.dsign[0][w * S.ndet_s[0] + u0]; lg += st.dlog[0][w * S.ndet_s[0] + u0]; }
    if (S.ndn > 0) { sgn *= st.dsign[1][w * S.ndet_s[1] + u1]; lg += st.dlog[1][w * S.ndet_s[1] + u1]; }
    v = sgn * exp(lg - psi_log[w]) / sg;
  }
  out[idx] = v;
}

// d_mo[w][a][m] = sum_di coeff[di] d_det[w][di] * sum_e ao[w][e][a] inverse_det[col(m)][e]   for m occupied in the
// determinant (:507-533; _testcol :382-388).  ao: [W*N][nao] values of all electrons (walker-major, electron order of x);
// colmap: [ndet_s][nmo_s] column of orbital m in unique determinant u or -1.  grid = (W), block = 256.
template <int PQA_UNIT = 0>  // (a template so that only the units that launch it compile it)
static __global__ __launch_bounds__(256) void k_pgrad_mo(SysDev S, SlaterState st, int s, const double* __restrict__ ao,
                                                  const double* __restrict__ d_det, const int* __restrict__ colmap,
                                                  double* __restrict__ out) {
  extern __shared__ double wu[];  // [ndet_s] weight of every unique determinant of this spin
  const long w = blockIdx.x;
  const int n = s ? S.ndn : S.nup, D = S.ndet_s[s], nmo = S.nmo[s], nao = S.nao;
  for (int u = threadIdx.x; u < D; u += blockDim.x) {
    double acc = 0.0;
    for (int di = 0; di < S.ndet; ++di)
      if (S.det_map[s * S.ndet + di] == u) acc += S.det_coeff[di] * d_det[w * S.ndet + di];
    wu[u] = acc;
  }
  __syncthreads();
  const double* aow = ao + ((size_t)w * S.nelec + (size_t)s * S.nup) * nao;
  const double* Tw = st.T[s] + (size_t)w * D * n * n;
  for (int idx = threadIdx.x; idx < nao * nmo; idx += blockDim.x) {
    const int a = idx / nmo, m = idx % nmo;
    double acc = 0.0;
    for (int u = 0; u < D; ++u) {
      const int col = colmap[u * nmo + m];
      if (col < 0) continue;
      const double* Tu = Tw + (size_t)u * n * n;
      double t = 0.0;
      for (int e = 0; e < n; ++e) t += aow[(size_t)e * nao + a] * Tu[e * n + col];
      acc += wu[u] * t;
    }
    out[((size_t)w * nao + a) * nmo + m] = acc;
  }
}
}  // inline namespace PQA_SYNC_NS
