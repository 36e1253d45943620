// Complex Slater determinants (Bloch orbitals at k-points off the time-reversal-invariant set), wave-per-walker.
//
// Same algorithms as pqa_slater.hpp in complex arithmetic (slater.py:212-216 dtype = complex, get_phase = z/|z|):
// build + invert :227-260, value determinant_tools.py:74-88, row-replacement ratios :301-380, Sherman-Morrison :88-94.
// Conventions of the complex mode (SysDev.cplx != 0):
//   * the orbital kernel sees REAL coefficient matrices [Re C | Im C] (nao x 2 nmo) — S.nmo[s] counts those real
//     columns, the number of orbitals is S.nmo[s] / 2; an MO row is [ncomp][2 nmo] with the real parts first;
//   * determinant occupations index orbitals (0 .. nmo-1);
//   * T[s] is [W][D][n][n] complex, (re, im) interleaved, electron-major like the real layout;
//   * dsign[s] is [W][D] complex unit phases (interleaved), dlog[s] [W][D] real.
#pragma once
#include "pqa_slater.hpp"

struct cx {
  double r, i;
};
__device__ __forceinline__ cx cmul(cx a, cx b) { return {a.r * b.r - a.i * b.i, a.r * b.i + a.i * b.r}; }
__device__ __forceinline__ cx cadd(cx a, cx b) { return {a.r + b.r, a.i + b.i}; }
__device__ __forceinline__ cx csub(cx a, cx b) { return {a.r - b.r, a.i - b.i}; }
__device__ __forceinline__ cx cscale(cx a, double f) { return {a.r * f, a.i * f}; }
__device__ __forceinline__ double cabs2(cx a) { return a.r * a.r + a.i * a.i; }
__device__ __forceinline__ cx cdiv(cx a, cx b) {
  const double d = 1.0 / cabs2(b);
  return {(a.r * b.r + a.i * b.i) * d, (a.i * b.r - a.r * b.i) * d};
}
__device__ __forceinline__ cx cphase(cx a) {  // z / |z| (nan for 0, like numpy)
  const double m = sqrt(cabs2(a));
  return {a.r / m, a.i / m};
}
__device__ __forceinline__ cx wave_sum_cx(cx a) { return {wave_sum(a.r), wave_sum(a.i)}; }

// ---------------------------------------------------------------- build + invert
// grid = W * ndet_s blocks of 64 threads; dynamic LDS: 2 * n * (n+1) doubles + n ints.
template <int PQA_UNIT = 0>  // (a template so that only the units that launch it compile it)
static __global__ __launch_bounds__(64) void k_build_invert_c(SysDev S, SlaterState st, int s, long W) {
  extern __shared__ double lds[];
  const int n = s ? S.ndn : S.nup, nmo2 = S.nmo[s], nmo = nmo2 / 2, D = S.ndet_s[s];
  if (n == 0) return;
  const long w = blockIdx.x / D;
  const int d = blockIdx.x % D;
  const int lane = threadIdx.x, ld = n + 1;
  double* Mr = lds;
  double* Mi = lds + (size_t)n * ld;
  int* perm = (int*)(lds + (size_t)2 * n * ld);
  const int* occ = S.det_occ[s] + (size_t)d * n;
  const double* cw = st.cache[s] + (size_t)w * n * 5 * nmo2;
  for (int idx = lane; idx < n * n; idx += 64) {
    const int i = idx / n, j = idx % n;
    Mr[j * ld + i] = cw[(size_t)i * 5 * nmo2 + occ[j]];
    Mi[j * ld + i] = cw[(size_t)i * 5 * nmo2 + nmo + occ[j]];
  }
  __syncthreads();
  cx phase = {1.0, 0.0};
  double logd = 0.0;
  bool singular = false;
  for (int k = 0; k < n; ++k) {
    double v = (lane >= k && lane < n) ? Mr[lane * ld + k] * Mr[lane * ld + k] + Mi[lane * ld + k] * Mi[lane * ld + k] : -1.0;
    int idx = lane;
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
      const double ov = __shfl_xor(v, off, 64);
      const int oi = __shfl_xor(idx, off, 64);
      if (ov > v || (ov == v && oi < idx)) { v = ov; idx = oi; }
    }
    const int p = idx;
    if (!(v > 0.0) || !(v <= DBL_MAX)) { singular = true; break; }
    if (p != k && lane < n) {
      double t = Mr[k * ld + lane]; Mr[k * ld + lane] = Mr[p * ld + lane]; Mr[p * ld + lane] = t;
      t = Mi[k * ld + lane]; Mi[k * ld + lane] = Mi[p * ld + lane]; Mi[p * ld + lane] = t;
    }
    if (p != k) phase = cscale(phase, -1.0);
    if (lane == 0) perm[k] = p;
    __syncthreads();
    const cx piv = {Mr[k * ld + k], Mi[k * ld + k]};
    logd += 0.5 * log(cabs2(piv));
    phase = cmul(phase, cphase(piv));
    __syncthreads();
    cx rk = {0.0, 0.0};
    if (lane < n) {
      const cx one = {1.0, 0.0};
      rk = (lane == k) ? cdiv(one, piv) : cdiv(cx{Mr[k * ld + lane], Mi[k * ld + lane]}, piv);
      Mr[k * ld + lane] = rk.r; Mi[k * ld + lane] = rk.i;
    }
    for (int r = 0; r < n; ++r) {
      if (r == k) continue;
      const cx f = {Mr[r * ld + k], Mi[r * ld + k]};
      if (lane < n) {
        const cx cur = (lane == k) ? cx{0.0, 0.0} : cx{Mr[r * ld + lane], Mi[r * ld + lane]};
        const cx nv = csub(cur, cmul(f, rk));
        Mr[r * ld + lane] = nv.r; Mi[r * ld + lane] = nv.i;
      }
    }
    __syncthreads();
  }
  double* Tw = st.T[s] + ((size_t)w * D + d) * n * n * 2;
  double* ph = st.dsign[s] + ((size_t)w * D + d) * 2;
  if (singular) {
    for (int idx = lane; idx < 2 * n * n; idx += 64) Tw[idx] = 0.0;
    if (lane == 0) { ph[0] = 0.0; ph[1] = 0.0; st.dlog[s][w * D + d] = -INFINITY; }
    return;
  }
  for (int k = n - 1; k >= 0; --k) {
    const int p = perm[k];
    if (p != k && lane < n) {
      double t = Mr[lane * ld + k]; Mr[lane * ld + k] = Mr[lane * ld + p]; Mr[lane * ld + p] = t;
      t = Mi[lane * ld + k]; Mi[lane * ld + k] = Mi[lane * ld + p]; Mi[lane * ld + p] = t;
    }
    __syncthreads();
  }
  for (int idx = lane; idx < n * n; idx += 64) {
    Tw[2 * idx] = Mr[(idx / n) * ld + idx % n];
    Tw[2 * idx + 1] = Mi[(idx / n) * ld + idx % n];
  }
  if (lane == 0) { ph[0] = phase.r; ph[1] = phase.i; st.dlog[s][w * D + d] = logd; }
}

// More than 64 electrons of a spin: the complex tile (2 n (n + 1) doubles: 188 KB at n = 108) does not fit LDS, so the elimination runs
// on a scratch matrix in global memory (M: [block][2][n][n + 1], volatile accesses: every element is written by one lane and read by
// others of the same wave between block barriers), lanes owning columns lane, lane + 64.  Same pivoting rule and arithmetic as above.
// grid = nw * ndet_s blocks of 64 threads for walkers [w0, w0 + nw); dynamic LDS: n ints.
template <int PQA_UNIT = 0>
static __global__ __launch_bounds__(64) void k_build_invert_cg(SysDev S, SlaterState st, int s, long w0, double* __restrict__ scratch) {
  extern __shared__ double lds[];
  const int n = s ? S.ndn : S.nup, nmo2 = S.nmo[s], nmo = nmo2 / 2, D = S.ndet_s[s];
  if (n == 0) return;
  const long w = w0 + blockIdx.x / D;
  const int d = blockIdx.x % D;
  const int lane = threadIdx.x, ld = n + 1;
  volatile double* Mr = scratch + (size_t)blockIdx.x * 2 * n * ld;
  volatile double* Mi = Mr + (size_t)n * ld;
  int* perm = (int*)lds;
  const int* occ = S.det_occ[s] + (size_t)d * n;
  const double* cw = st.cache[s] + (size_t)w * n * 5 * nmo2;
  for (int idx = lane; idx < n * n; idx += 64) {
    const int i = idx / n, j = idx % n;
    Mr[j * ld + i] = cw[(size_t)i * 5 * nmo2 + occ[j]];
    Mi[j * ld + i] = cw[(size_t)i * 5 * nmo2 + nmo + occ[j]];
  }
  __syncthreads();
  cx phase = {1.0, 0.0};
  double logd = 0.0;
  bool singular = false;
  for (int k = 0; k < n; ++k) {
    double v = -1.0;
    int idx = lane;
    for (int r = lane; r < n; r += 64) {
      const double a = Mr[r * ld + k], b = Mi[r * ld + k];
      const double c = (r >= k) ? a * a + b * b : -1.0;
      if (c > v) { v = c; idx = r; }
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
        double t = Mr[k * ld + c]; Mr[k * ld + c] = Mr[p * ld + c]; Mr[p * ld + c] = t;
        t = Mi[k * ld + c]; Mi[k * ld + c] = Mi[p * ld + c]; Mi[p * ld + c] = t;
      }
    if (p != k) phase = cscale(phase, -1.0);
    if (lane == 0) perm[k] = p;
    __syncthreads();
    const cx piv = {Mr[k * ld + k], Mi[k * ld + k]};
    logd += 0.5 * log(cabs2(piv));
    phase = cmul(phase, cphase(piv));
    __syncthreads();
    cx rk[PQA_MAXN / 64];
#pragma unroll
    for (int q = 0; q < PQA_MAXN / 64; ++q) {
      const int c = lane + 64 * q;
      rk[q] = {0.0, 0.0};
      if (c < n) {
        const cx one = {1.0, 0.0};
        rk[q] = (c == k) ? cdiv(one, piv) : cdiv(cx{Mr[k * ld + c], Mi[k * ld + c]}, piv);
        Mr[k * ld + c] = rk[q].r; Mi[k * ld + c] = rk[q].i;
      }
    }
    __syncthreads();
    for (int r = 0; r < n; ++r) {
      if (r == k) continue;
      const cx f = {Mr[r * ld + k], Mi[r * ld + k]};
      __syncthreads();  // (every lane has column k's entry of row r before its owner overwrites it)
#pragma unroll
      for (int q = 0; q < PQA_MAXN / 64; ++q) {
        const int c = lane + 64 * q;
        if (c < n) {
          const cx cur = (c == k) ? cx{0.0, 0.0} : cx{Mr[r * ld + c], Mi[r * ld + c]};
          const cx nv = csub(cur, cmul(f, rk[q]));
          Mr[r * ld + c] = nv.r; Mi[r * ld + c] = nv.i;
        }
      }
    }
    __syncthreads();
  }
  double* Tw = st.T[s] + ((size_t)w * D + d) * n * n * 2;
  double* ph = st.dsign[s] + ((size_t)w * D + d) * 2;
  if (singular) {
    for (int idx = lane; idx < 2 * n * n; idx += 64) Tw[idx] = 0.0;
    if (lane == 0) { ph[0] = 0.0; ph[1] = 0.0; st.dlog[s][w * D + d] = -INFINITY; }
    return;
  }
  for (int k = n - 1; k >= 0; --k) {
    const int p = perm[k];
    if (p != k)
      for (int r = lane; r < n; r += 64) {
        double t = Mr[r * ld + k]; Mr[r * ld + k] = Mr[r * ld + p]; Mr[r * ld + p] = t;
        t = Mi[r * ld + k]; Mi[r * ld + k] = Mi[r * ld + p]; Mi[r * ld + p] = t;
      }
    __syncthreads();
  }
  for (int idx = lane; idx < n * n; idx += 64) {
    Tw[2 * idx] = Mr[(idx / n) * ld + idx % n];
    Tw[2 * idx + 1] = Mi[(idx / n) * ld + idx % n];
  }
  if (lane == 0) { ph[0] = phase.r; ph[1] = phase.i; st.dlog[s][w * D + d] = logd; }
}

// ---------------------------------------------------------------- multi-determinant bookkeeping
__device__ __forceinline__ cx det_weight_c(const SysDev& S, const SlaterState& st, long w, int Dd, double ref) {
  cx ph = {1.0, 0.0};
  if (S.nup > 0) { const double* p = st.dsign[0] + ((size_t)w * S.ndet_s[0] + S.det_map[Dd]) * 2; ph = cmul(ph, cx{p[0], p[1]}); }
  if (S.ndn > 0) { const double* p = st.dsign[1] + ((size_t)w * S.ndet_s[1] + S.det_map[S.ndet + Dd]) * 2; ph = cmul(ph, cx{p[0], p[1]}); }
  const double l = det_logsum(S, st, w, Dd);
  const double ex = (l == -INFINITY) ? 0.0 : exp(l - ref);
  return cscale(ph, S.det_coeff[Dd] * ex);
}

// (phase, log|Psi_S|) of one walker — determinant_tools.compute_value (:74-88) with get_phase = z/|z|
__device__ __forceinline__ void slater_value_wave_c(const SysDev& S, const SlaterState& st, long w, cx& phase, double& logv) {
  const double ref = det_ref(S, st, w);
  cx t = {0.0, 0.0};
  for (int Dd = threadIdx.x & 63; Dd < S.ndet; Dd += 64) t = cadd(t, det_weight_c(S, st, w, Dd, ref));
  t = wave_sum_cx(t);
  const double m = sqrt(cabs2(t));
  phase = {clamp_nan_to_num(t.r / m), clamp_nan_to_num(t.i / m)};
  logv = clamp_nan_to_num(log(m) + ref);
}

template <int PQA_UNIT = 0>  // (a template so that only the units that launch it compile it)
static __global__ __launch_bounds__(64) void k_slater_value_c(SysDev S, SlaterState st, double* sign, double* logv) {
  const long w = blockIdx.x;
  cx ph;
  double lv;
  slater_value_wave_c(S, st, w, ph, lv);
  if (threadIdx.x == 0) { sign[2 * w] = ph.r; sign[2 * w + 1] = ph.i; logv[w] = lv; }
}

// Ratios (new row)/(current) of electron i for NCOMP stacked rows mo[c][2 nmo] (re block, im block).
// scratch: >= ndet_s * NCOMP * 2 doubles of LDS (multi-determinant only).
template <int NCOMP>
__device__ __forceinline__ void slater_ratios_c(const SysDev& S, const SlaterState& st, int s, int i, long w,
                                                const double* __restrict__ mo, cx (&out)[NCOMP], double* scratch) {
  const int lane = threadIdx.x & 63;
  const int n = s ? S.ndn : S.nup, nmo2 = S.nmo[s], nmo = nmo2 / 2, D = S.ndet_s[s];
  auto dots = [&](int d, cx (&part)[NCOMP]) {
    const double* Trow = st.T[s] + ((((size_t)w * D + d) * n + i) * n) * 2;
    const int* occ = S.det_occ[s] + (size_t)d * n;
#pragma unroll
    for (int c = 0; c < NCOMP; ++c) part[c] = {0.0, 0.0};
    for (int j = lane; j < n; j += 64) {
      const cx t = {Trow[2 * j], Trow[2 * j + 1]};
      const int o = occ[j];
#pragma unroll
      for (int c = 0; c < NCOMP; ++c) part[c] = cadd(part[c], cmul(cx{mo[c * nmo2 + o], mo[c * nmo2 + nmo + o]}, t));
    }
#pragma unroll
    for (int c = 0; c < NCOMP; ++c) part[c] = wave_sum_cx(part[c]);
  };
  if (S.ndet == 1) {
    dots(0, out);
    return;
  }
  for (int d = 0; d < D; ++d) {
    cx part[NCOMP];
    dots(d, part);
    if (lane == 0)
#pragma unroll
      for (int c = 0; c < NCOMP; ++c) { scratch[(d * NCOMP + c) * 2] = part[c].r; scratch[(d * NCOMP + c) * 2 + 1] = part[c].i; }
  }
  __syncthreads();
  const double ref = det_ref(S, st, w);
  cx num[NCOMP], den = {0.0, 0.0};
#pragma unroll
  for (int c = 0; c < NCOMP; ++c) num[c] = {0.0, 0.0};
  for (int Dd = lane; Dd < S.ndet; Dd += 64) {
    const cx wt = det_weight_c(S, st, w, Dd, ref);
    const int ds = S.det_map[s * S.ndet + Dd];
    den = cadd(den, wt);
#pragma unroll
    for (int c = 0; c < NCOMP; ++c) num[c] = cadd(num[c], cmul(wt, cx{scratch[(ds * NCOMP + c) * 2], scratch[(ds * NCOMP + c) * 2 + 1]}));
  }
  den = wave_sum_cx(den);
#pragma unroll
  for (int c = 0; c < NCOMP; ++c) out[c] = cdiv(wave_sum_cx(num[c]), den);
  __syncthreads();
}

// out (NCOMP, nrow*npt) complex (interleaved); mo rows [(r*npt+q)][NCOMP][2 nmo]; LDS: max(ndet_s)*NCOMP*2 doubles
template <int NCOMP>
static __global__ __launch_bounds__(64) void k_slater_eval_c(SysDev S, SlaterState st, int e, const double* __restrict__ mo, long nrow,
                                                      int npt, const int* __restrict__ widx, double* __restrict__ out) {
  extern __shared__ double lds[];
  const long r = blockIdx.x;
  const long w = widx ? widx[r] : r;
  const int s = e >= S.nup, i = e - s * S.nup, nmo2 = S.nmo[s];
  for (int q = 0; q < npt; ++q) {
    cx rat[NCOMP];
    slater_ratios_c<NCOMP>(S, st, s, i, w, mo + ((size_t)(r * npt + q) * NCOMP) * nmo2, rat, lds);
    if (threadIdx.x == 0)
#pragma unroll
      for (int c = 0; c < NCOMP; ++c) {
        const size_t o = ((size_t)c * nrow * npt + r * npt + q) * 2;
        out[o] = rat[c].r; out[o + 1] = rat[c].i;
      }
  }
}

// ---------------------------------------------------------------- Sherman-Morrison (slater.py:88-94, :286-291)
// lane = row j of the tile.  LDS: 2 n (n+1) + 2 n + 2 n doubles.
__device__ __forceinline__ void sm_update_wave_c(const SysDev& S, const SlaterState& st, int s, int i, long w,
                                                 const double* __restrict__ morow, double* lds) {
  const int lane = threadIdx.x & 63;
  const int n = s ? S.ndn : S.nup, nmo = S.nmo[s] / 2, D = S.ndet_s[s], ld = n + 1;
  if (n > PQA_MAXN_FAST) {  // on the inverse in place (see sm_update_wave): LDS 6 n doubles
    double* V = lds;       // [n][2]
    double* TMP = V + 2 * n;
    double* R = TMP + 2 * n;
    for (int d = 0; d < D; ++d) {
      double* Tw = st.T[s] + ((size_t)w * D + d) * n * n * 2;
      const int* occ = S.det_occ[s] + (size_t)d * n;
      for (int k = lane; k < n; k += 64) { V[2 * k] = morow[occ[k]]; V[2 * k + 1] = morow[nmo + occ[k]]; }
      __syncthreads();
      for (int j = 0; j < n; ++j) {
        cx p = {0.0, 0.0};
        for (int k = lane; k < n; k += 64) p = cadd(p, cmul(cx{V[2 * k], V[2 * k + 1]}, cx{Tw[2 * ((size_t)j * n + k)], Tw[2 * ((size_t)j * n + k) + 1]}));
        p = wave_sum_cx(p);
        if (lane == 0) { TMP[2 * j] = p.r; TMP[2 * j + 1] = p.i; }
      }
      __syncthreads();
      const cx ratio = {TMP[2 * i], TMP[2 * i + 1]};
      for (int k = lane; k < n; k += 64) {
        const cx q = cdiv(cx{Tw[2 * ((size_t)i * n + k)], Tw[2 * ((size_t)i * n + k) + 1]}, ratio);
        R[2 * k] = q.r; R[2 * k + 1] = q.i;
      }
      __syncthreads();
      for (int j = 0; j < n; ++j) {
        const cx tj = {TMP[2 * j], TMP[2 * j + 1]};
        for (int k = lane; k < n; k += 64) {
          const cx rk = {R[2 * k], R[2 * k + 1]};
          double* t = Tw + 2 * ((size_t)j * n + k);
          const cx nv = (j == i) ? rk : csub(cx{t[0], t[1]}, cmul(rk, tj));
          t[0] = nv.r; t[1] = nv.i;
        }
      }
      if (lane == 0) {
        double* ph = st.dsign[s] + ((size_t)w * D + d) * 2;
        const cx np_ = cmul(cx{ph[0], ph[1]}, cphase(ratio));
        ph[0] = np_.r; ph[1] = np_.i;
        st.dlog[s][(size_t)w * D + d] += 0.5 * log(cabs2(ratio));
      }
      __syncthreads();
    }
    return;
  }
  double* Lr = lds;
  double* Li = Lr + (size_t)n * ld;
  double* Vr = Li + (size_t)n * ld;
  double* Vi = Vr + n;
  double* Rr = Vi + n;
  double* Ri = Rr + n;
  for (int d = 0; d < D; ++d) {
    double* Tw = st.T[s] + ((size_t)w * D + d) * n * n * 2;
    const int* occ = S.det_occ[s] + (size_t)d * n;
    for (int idx = lane; idx < n * n; idx += 64) {
      Lr[(idx / n) * ld + idx % n] = Tw[2 * idx];
      Li[(idx / n) * ld + idx % n] = Tw[2 * idx + 1];
    }
    for (int k = lane; k < n; k += 64) { Vr[k] = morow[occ[k]]; Vi[k] = morow[nmo + occ[k]]; }
    __syncthreads();
    cx tmp = {0.0, 0.0};
    if (lane < n)
      for (int k = 0; k < n; ++k) tmp = cadd(tmp, cmul(cx{Vr[k], Vi[k]}, cx{Lr[lane * ld + k], Li[lane * ld + k]}));
    const cx ratio = {__shfl(tmp.r, i, 64), __shfl(tmp.i, i, 64)};
    if (lane < n) {
      const cx q = cdiv(cx{Lr[i * ld + lane], Li[i * ld + lane]}, ratio);
      Rr[lane] = q.r; Ri[lane] = q.i;
    }
    __syncthreads();
    if (lane < n) {
      for (int k = 0; k < n; ++k) {
        const cx rk = {Rr[k], Ri[k]};
        const cx nv = (lane == i) ? rk : csub(cx{Lr[lane * ld + k], Li[lane * ld + k]}, cmul(rk, tmp));
        Lr[lane * ld + k] = nv.r; Li[lane * ld + k] = nv.i;
      }
    }
    __syncthreads();
    for (int idx = lane; idx < n * n; idx += 64) {
      Tw[2 * idx] = Lr[(idx / n) * ld + idx % n];
      Tw[2 * idx + 1] = Li[(idx / n) * ld + idx % n];
    }
    if (lane == 0) {
      double* ph = st.dsign[s] + ((size_t)w * D + d) * 2;
      const cx np_ = cmul(cx{ph[0], ph[1]}, cphase(ratio));
      ph[0] = np_.r; ph[1] = np_.i;
      st.dlog[s][(size_t)w * D + d] += 0.5 * log(cabs2(ratio));
    }
    __syncthreads();
  }
}

template <int PQA_UNIT = 0>  // (a template so that only the units that launch it compile it)
static __global__ __launch_bounds__(64) void k_sm_update_c(SysDev S, SlaterState st, int e, const double* __restrict__ mo, int row_stride,
                                                    const uint8_t* __restrict__ mask, int to_cache) {
  extern __shared__ double lds[];
  const long w = blockIdx.x;
  if (mask && !mask[w]) return;
  const int s = e >= S.nup, i = e - s * S.nup, n = s ? S.ndn : S.nup, nmo2 = S.nmo[s];
  const double* row = mo + (size_t)w * row_stride;
  sm_update_wave_c(S, st, s, i, w, row, lds);
  if (to_cache) {
    double* c = st.cache[s] + ((size_t)w * n + i) * 5 * nmo2;
    for (int k = threadIdx.x; k < 5 * nmo2; k += 64) c[k] = row[k];
  }
}

// ---------------------------------------------------------------- parameter gradients of complex determinants (slater.py:462-542)
// d_det[w][di] = D_up D_dn / Psi (complex, interleaved): phase_up phase_dn e^{log_up + log_dn - log|Psi|} / phase(Psi).
template <int PQA_UNIT = 0>  // (a template so that only the units that launch it compile it)
static __global__ void k_pgrad_det_c(SysDev S, SlaterState st, const double* __restrict__ psi_phase, const double* __restrict__ psi_log,
                                     long W, double* __restrict__ out) {
  const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= W * S.ndet) return;
  const long w = idx / S.ndet;
  const int di = (int)(idx % S.ndet), u0 = S.det_map[di], u1 = S.det_map[S.ndet + di];
  const cx pp = {psi_phase[2 * w], psi_phase[2 * w + 1]};
  cx v = {0.0, 0.0};
  if (pp.r != 0.0 || pp.i != 0.0) {
    cx ph = {1.0, 0.0};
    double lg = 0.0;
    if (S.nup > 0) { const double* q = st.dsign[0] + 2 * (w * S.ndet_s[0] + u0); ph = cmul(ph, cx{q[0], q[1]}); lg += st.dlog[0][w * S.ndet_s[0] + u0]; }
    if (S.ndn > 0) { const double* q = st.dsign[1] + 2 * (w * S.ndet_s[1] + u1); ph = cmul(ph, cx{q[0], q[1]}); lg += st.dlog[1][w * S.ndet_s[1] + u1]; }
    v = cdiv(cscale(ph, exp(lg - psi_log[w])), pp);
  }
  out[2 * idx] = v.r; out[2 * idx + 1] = v.i;
}

// d_mo[w][a][m] = sum_di coeff[di] d_det[w][di] * sum_e ao[w][e][a] inverse_u[col(m)][e]  (holomorphic: no conjugation; _testcol
// slater.py:382-388 in complex arithmetic).  ao: [W*N][nao] real AOs, or with AOCX (twisted cells) the real plane followed by
// the imaginary plane, each [W*N][nao].  out: [W][nao][nmo] complex interleaved, nmo = orbitals.  grid = W, block = 256.
template <bool AOCX>
static __global__ __launch_bounds__(256) void k_pgrad_mo_c(SysDev S, SlaterState st, int s, const double* __restrict__ ao, long ao_plane,
                                                           const double* __restrict__ d_det, const int* __restrict__ colmap,
                                                           double* __restrict__ out) {
  extern __shared__ double wu[];  // [ndet_s][2]
  const long w = blockIdx.x;
  const int n = s ? S.ndn : S.nup, D = S.ndet_s[s], nmo2 = S.nmo[s], nmo = nmo2 / 2, nao = S.nao;
  for (int u = threadIdx.x; u < D; u += blockDim.x) {
    cx acc = {0.0, 0.0};
    for (int di = 0; di < S.ndet; ++di)
      if (S.det_map[s * S.ndet + di] == u) {
        const double* q = d_det + 2 * (w * S.ndet + di);
        acc = cadd(acc, cscale(cx{q[0], q[1]}, S.det_coeff[di]));
      }
    wu[2 * u] = acc.r; wu[2 * u + 1] = acc.i;
  }
  __syncthreads();
  const double* aow = ao + ((size_t)w * S.nelec + (size_t)s * S.nup) * nao;
  const double* Tw = st.T[s] + (size_t)w * D * n * n * 2;
  for (int idx = threadIdx.x; idx < nao * nmo; idx += blockDim.x) {
    const int a = idx / nmo, m = idx % nmo;
    cx acc = {0.0, 0.0};
    for (int u = 0; u < D; ++u) {
      const int col = colmap[u * nmo2 + m];
      if (col < 0) continue;
      const double* Tu = Tw + (size_t)u * n * n * 2;
      cx t = {0.0, 0.0};
      for (int e = 0; e < n; ++e) {
        const cx tv = {Tu[2 * (e * n + col)], Tu[2 * (e * n + col) + 1]};
        const cx av = {aow[(size_t)e * nao + a], AOCX ? aow[ao_plane + (size_t)e * nao + a] : 0.0};
        t = cadd(t, cmul(av, tv));
      }
      acc = cadd(acc, cmul(cx{wu[2 * u], wu[2 * u + 1]}, t));
    }
    out[2 * (((size_t)w * nao + a) * nmo + m)] = acc.r;
    out[2 * (((size_t)w * nao + a) * nmo + m) + 1] = acc.i;
  }
}
