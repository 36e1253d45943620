// Density-matrix sampling on the device: the auxiliary one-electron Metropolis walk of the
// one- and two-body density-matrix estimators and their per-walker contractions.
//
// What is computed follows pyqmc/observables/obdm.py:215-250 (sample_onebody: walkers distributed as
// f(r) = sum_i |phi_i(r)|^2), obdm.py:139-193 (one-body estimator) and tbdm.py:188-283 (two-body estimator, Eq. 10 of
// DOI:10.1063/1.4793531).  How it is computed is not the reference's: the walk never leaves the device (one k_orb launch
// and one accept kernel per sample, random numbers from tapes or Philox), and the two-body contraction is factorised,
//   value[(i,j,k,l)] = M[i][k] conj(phi_j(r1')) conj(phi_l(r2')) / (f1 f2),   M = Phi_a^T (wfratio) Phi_b,
// two small matrix products per walker instead of a (walkers x electron pairs x index tuples) tensor.
//
// Orbital rows are k_orb's output layout: [point][nmo2], nmo2 = nmo (real) or 2 nmo (complex: real block | imaginary block).
#pragma once
#include "pqa_common.hpp"

struct cplx2 { double r, i; };
__device__ __forceinline__ cplx2 dm_mul(cplx2 a, cplx2 b) { return {a.r * b.r - a.i * b.i, a.r * b.i + a.i * b.r}; }
__device__ __forceinline__ cplx2 dm_conj(cplx2 a) { return {a.r, -a.i}; }
// element j of an orbital row
__device__ __forceinline__ cplx2 dm_row(const double* __restrict__ row, int j, int nmo, int oc) { return {row[j], oc ? row[nmo + j] : 0.0}; }
// element of a ratio array that is real (rc = 0) or interleaved complex (rc = 1)
__device__ __forceinline__ cplx2 dm_rat(const double* __restrict__ a, size_t idx, int rc) { return rc ? cplx2{a[2 * idx], a[2 * idx + 1]} : cplx2{a[idx], 0.0}; }

// f[p] = sum_j |row[p][j]|^2
template <int PQA_UNIT = 0>  // (a template so that only the units that launch it compile it)
static __global__ __launch_bounds__(256) void k_dm_density(const double* __restrict__ rows, long n, int nmo2, double* __restrict__ f) {
  const long p = (long)blockIdx.x * 256 + threadIdx.x;
  if (p >= n) return;
  const double* r = rows + (size_t)p * nmo2;
  double s = 0.0;
  for (int j = 0; j < nmo2; ++j) s += r[j] * r[j];
  f[p] = s;
}

// proposal of sample s: r' = r + sqrt(tstep) z (obdm.py:232-237; the walkers are kept in unfolded coordinates — the orbital
// kernel folds every point itself, which also gives a twisted cell its wrap phase)
template <int PQA_UNIT = 0>  // (a template so that only the units that launch it compile it)
static __global__ __launch_bounds__(256) void k_dm_propose(const double* __restrict__ pos, const double* __restrict__ gauss, uint64_t seed, uint32_t s,
                                                    double sq, long n, double* __restrict__ newpos) {
  const long p = (long)blockIdx.x * 256 + threadIdx.x;
  if (p >= n) return;
  double z0, z1, z2, z3;
  if (gauss) { z0 = gauss[3 * p]; z1 = gauss[3 * p + 1]; z2 = gauss[3 * p + 2]; }
  else {
    normal2(philox(seed, (uint32_t)p, s, PQA_STREAM_GAUSS_A, 0x444du), z0, z1);
    normal2(philox(seed, (uint32_t)p, s, PQA_STREAM_GAUSS_B, 0x444du), z2, z3);
  }
  newpos[3 * p] = pos[3 * p] + sq * z0; newpos[3 * p + 1] = pos[3 * p + 1] + sq * z1; newpos[3 * p + 2] = pos[3 * p + 2] + sq * z2;
}

// accept with probability f(r')/f(r) (obdm.py:240-246); accepted walkers take the new position, orbital row and density.
// keep_*: where this sample's walkers are recorded (NULL: not kept).  One wave per walker, lanes over the row.
template <int PQA_UNIT = 0>  // (a template so that only the units that launch it compile it)
static __global__ __launch_bounds__(64) void k_dm_accept(double* __restrict__ pos, double* __restrict__ rows, double* __restrict__ f,
                                                  const double* __restrict__ newpos, const double* __restrict__ newrows,
                                                  const double* __restrict__ unif, uint64_t seed, uint32_t s, long n, int nmo2,
                                                  double* __restrict__ accept, double* __restrict__ keep_pos,
                                                  double* __restrict__ keep_rows, double* __restrict__ keep_f) {
  const long p = blockIdx.x;
  const int lane = threadIdx.x;
  const double* nr = newrows + (size_t)p * nmo2;
  double* cr = rows + (size_t)p * nmo2;
  double part = 0.0;
  for (int j = lane; j < nmo2; j += 64) part += nr[j] * nr[j];
  const double fn = wave_sum(part), fo = f[p];
  double u;
  if (unif) u = unif[p];
  else {
    const Philox ph = philox(seed, (uint32_t)p, s, PQA_STREAM_ACCEPT, 0x444du);
    u = u01(ph.c[0], ph.c[1]);
  }
  const bool acc = fn / fo > u;
  if (acc) {
    for (int j = lane; j < nmo2; j += 64) cr[j] = nr[j];
    if (lane < 3) pos[3 * p + lane] = newpos[3 * p + lane];
    if (lane == 0) f[p] = fn;
  }
  if (lane == 0 && accept) accept[p] = acc ? 1.0 : 0.0;
  if (keep_rows) {
    for (int j = lane; j < nmo2; j += 64) keep_rows[(size_t)p * nmo2 + j] = acc ? nr[j] : cr[j];
    if (lane < 3) keep_pos[3 * p + lane] = acc ? newpos[3 * p + lane] : pos[3 * p + lane];
    if (lane == 0) keep_f[p] = acc ? fn : fo;
  }
}

// One-body estimator, one block per configuration n (obdm.py:170-190):
//   value[n][j][k] (+)= (phi_j(r')/F) conj( sum_e ratio[n][e] phi_k(r_e) ),  norm[n][j] (+)= |phi_j(r')|^2 / F,  F = f(r') / norb
// with r' the auxiliary walker assign[n] of the kept sample.  cfg: [nconf][nelec][nmo2] orbitals at the electrons.
// LDS: 2 norb doubles.
template <int PQA_UNIT = 0>  // (a template so that only the units that launch it compile it)
static __global__ __launch_bounds__(256) void k_obdm_acc(const double* __restrict__ aux_rows, const double* __restrict__ aux_f,
                                                  const int* __restrict__ assign, const double* __restrict__ cfg,
                                                  const double* __restrict__ ratio, int rc, int oc, int nelec, int norb, int first,
                                                  double* __restrict__ value, double* __restrict__ norm) {
  extern __shared__ double t[];  // conj(sum_e ratio phi_k(r_e)) as (re, im) pairs
  const long n = blockIdx.x;
  const int a = assign[n], nmo2 = oc ? 2 * norb : norb, cx = rc | oc;
  const double* brow = aux_rows + (size_t)a * nmo2;
  const double F = aux_f[a] / norb;
  for (int k = threadIdx.x; k < norb; k += 256) {
    cplx2 s = {0.0, 0.0};
    for (int e = 0; e < nelec; ++e) {
      const cplx2 r = dm_rat(ratio, (size_t)n * nelec + e, rc);
      const cplx2 ph = dm_row(cfg + ((size_t)n * nelec + e) * nmo2, k, norb, oc);
      const cplx2 pr = dm_mul(dm_conj(r), dm_conj(ph));
      s.r += pr.r; s.i += pr.i;
    }
    t[2 * k] = s.r; t[2 * k + 1] = s.i;
  }
  __syncthreads();
  for (int idx = threadIdx.x; idx < norb * norb; idx += 256) {
    const int j = idx / norb, k = idx - j * norb;
    cplx2 b = dm_row(brow, j, norb, oc);
    b.r /= F; b.i /= F;
    const cplx2 v = dm_mul(b, cplx2{t[2 * k], t[2 * k + 1]});
    const size_t o = ((size_t)n * norb + j) * norb + k;
    if (cx) {
      value[2 * o] = (first ? 0.0 : value[2 * o]) + v.r;
      value[2 * o + 1] = (first ? 0.0 : value[2 * o + 1]) + v.i;
    } else value[o] = (first ? 0.0 : value[o]) + v.r;
  }
  for (int j = threadIdx.x; j < norb; j += 256) {
    const cplx2 b = dm_row(brow, j, norb, oc);
    const size_t o = (size_t)n * norb + j;
    norm[o] = (first ? 0.0 : norm[o]) + (b.r * b.r + b.i * b.i) / F;
  }
}

// Two-body estimator, one block per configuration (tbdm.py:232-277).  ratio[n][a][b] = Psi(r_a -> r1', r_b -> r2') / Psi
// (0 for a pair that would move the same electron twice); cfg_a [nconf][nea][nmo2a], cfg_b [nconf][neb][nmo2b];
// ijkl [4][ntuple].  LDS: 2 (nea nb + na nb) doubles.
template <int PQA_UNIT = 0>  // (a template so that only the units that launch it compile it)
static __global__ __launch_bounds__(256) void k_tbdm_acc(const double* __restrict__ auxa_rows, const double* __restrict__ auxa_f,
                                                  const double* __restrict__ auxb_rows, const double* __restrict__ auxb_f,
                                                  const int* __restrict__ assign_a, const int* __restrict__ assign_b,
                                                  const double* __restrict__ cfg_a, const double* __restrict__ cfg_b,
                                                  const double* __restrict__ ratio, int rc, int oc, int nea, int neb, int na, int nb,
                                                  const int* __restrict__ ijkl, int ntuple, int first, double* __restrict__ value,
                                                  double* __restrict__ norm_a, double* __restrict__ norm_b) {
  extern __shared__ double lds[];
  double* u = lds;                         // [nea][nb] complex: sum_b ratio[a][b] phi_k(r_b)
  double* M = lds + (size_t)2 * nea * nb;  // [na][nb] complex
  const long n = blockIdx.x;
  const int cx = rc | oc, na2 = oc ? 2 * na : na, nb2 = oc ? 2 * nb : nb;
  const int aa = assign_a[n], ab = assign_b[n];
  for (int idx = threadIdx.x; idx < nea * nb; idx += 256) {
    const int a = idx / nb, k = idx - a * nb;
    cplx2 s = {0.0, 0.0};
    for (int b = 0; b < neb; ++b) {
      const cplx2 pr = dm_mul(dm_rat(ratio, ((size_t)n * nea + a) * neb + b, rc), dm_row(cfg_b + ((size_t)n * neb + b) * nb2, k, nb, oc));
      s.r += pr.r; s.i += pr.i;
    }
    u[2 * idx] = s.r; u[2 * idx + 1] = s.i;
  }
  __syncthreads();
  for (int idx = threadIdx.x; idx < na * nb; idx += 256) {
    const int i = idx / nb, k = idx - i * nb;
    cplx2 s = {0.0, 0.0};
    for (int a = 0; a < nea; ++a) {
      const cplx2 pr = dm_mul(dm_row(cfg_a + ((size_t)n * nea + a) * na2, i, na, oc), cplx2{u[2 * (a * nb + k)], u[2 * (a * nb + k) + 1]});
      s.r += pr.r; s.i += pr.i;
    }
    M[2 * idx] = s.r; M[2 * idx + 1] = s.i;
  }
  __syncthreads();
  const double* ra = auxa_rows + (size_t)aa * na2;
  const double* rb = auxb_rows + (size_t)ab * nb2;
  const double fa = auxa_f[aa], fb = auxb_f[ab], rho = 1.0 / (fa * fb);
  for (int o = threadIdx.x; o < ntuple; o += 256) {
    const int i = ijkl[o], j = ijkl[ntuple + o], k = ijkl[2 * ntuple + o], l = ijkl[3 * ntuple + o];
    cplx2 v = dm_mul(cplx2{M[2 * (i * nb + k)], M[2 * (i * nb + k) + 1]}, dm_mul(dm_conj(dm_row(ra, j, na, oc)), dm_conj(dm_row(rb, l, nb, oc))));
    v.r *= rho; v.i *= rho;
    const size_t q = (size_t)n * ntuple + o;
    if (cx) {
      value[2 * q] = (first ? 0.0 : value[2 * q]) + v.r;
      value[2 * q + 1] = (first ? 0.0 : value[2 * q + 1]) + v.i;
    } else value[q] = (first ? 0.0 : value[q]) + v.r;
  }
  for (int j = threadIdx.x; j < na; j += 256) {
    const cplx2 b = dm_row(ra, j, na, oc);
    const size_t o = (size_t)n * na + j;
    norm_a[o] = (first ? 0.0 : norm_a[o]) + (b.r * b.r + b.i * b.i) / fa;
  }
  for (int j = threadIdx.x; j < nb; j += 256) {
    const cplx2 b = dm_row(rb, j, nb, oc);
    const size_t o = (size_t)n * nb + j;
    norm_b[o] = (first ? 0.0 : norm_b[o]) + (b.r * b.r + b.i * b.i) / fb;
  }
}

// out[c] = scale * mean over rows of in[rows][cols] (deterministic: one block per column, fixed tree)
template <int PQA_UNIT = 0>  // (a template so that only the units that launch it compile it)
static __global__ __launch_bounds__(256) void k_col_means(const double* __restrict__ in, long rows, long cols, double scale, double* __restrict__ out) {
  __shared__ double sh[256];
  const long c = blockIdx.x;
  double s = 0.0;
  for (long r = threadIdx.x; r < rows; r += 256) s += in[r * cols + c];
  sh[threadIdx.x] = s;
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) {
    if ((int)threadIdx.x < o) sh[threadIdx.x] += sh[threadIdx.x + o];
    __syncthreads();
  }
  if (threadIdx.x == 0) out[c] = scale * sh[0] / (double)rows;
}
template <int PQA_UNIT = 0>  // (a template so that only the units that launch it compile it)
static __global__ __launch_bounds__(256) void k_scale_copy(const double* __restrict__ in, long n, double scale, double* __restrict__ out) {
  const long i = (long)blockIdx.x * 256 + threadIdx.x;
  if (i < n) out[i] = scale * in[i];
}

// ---------------------------------------------------------------- Gram matrix on the matrix cores
// C[p][q] = sum_n A[n][p] B[n][q]  (the moment matrix <dp_i dp_j> of stochastic reconfiguration,
// stochastic_reconfiguration.py:106-114: A = dp, B = weights * f * dp).  One wave per 16x16 tile of C and slice of n,
// v_mfma_f64_16x16x4_f64 over 4 configurations at a time; the slices' partial tiles are summed by k_gram_reduce in slice
// order (deterministic).  A, B row-major [n][p], [n][q]: a lane's operand A[n0 + kq][p0 + i16] is contiguous over i16.
template <int PQA_UNIT = 0>  // (a template so that only the units that launch it compile it)
static __global__ __launch_bounds__(64) void k_gram_mfma(const double* __restrict__ A, const double* __restrict__ B, long n, int P, int Q,
                                                  int nslice, double* __restrict__ part) {
  const int lane = threadIdx.x, i16 = lane & 15, kq = lane >> 4;
  const int p0 = blockIdx.x * 16, q0 = blockIdx.y * 16, sl = blockIdx.z;
  const long per = ((n + nslice - 1) / nslice + 3) & ~3L;
  const long n_lo = sl * per, n_hi = (n_lo + per < n) ? n_lo + per : n;
  d4 acc = {0.0, 0.0, 0.0, 0.0};
  const bool pin = p0 + i16 < P, qin = q0 + i16 < Q;
  for (long r = n_lo; r < n_hi; r += 4) {
    const long rr = r + kq;
    const double a = (pin && rr < n_hi) ? A[rr * P + p0 + i16] : 0.0;
    const double b = (qin && rr < n_hi) ? B[rr * Q + q0 + i16] : 0.0;
    acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc, 0, 0, 0);
  }
  // lane holds D[row = kq + 4 r][col = i16]
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const int p = p0 + kq + 4 * r, q = q0 + i16;
    if (p < P && q < Q) part[((size_t)sl * P + p) * Q + q] = acc[r];
  }
}
template <int PQA_UNIT = 0>  // (a template so that only the units that launch it compile it)
static __global__ __launch_bounds__(256) void k_gram_reduce(const double* __restrict__ part, long PQ, int nslice, double* __restrict__ C) {
  const long i = (long)blockIdx.x * 256 + threadIdx.x;
  if (i >= PQ) return;
  double s = 0.0;
  for (int sl = 0; sl < nslice; ++sl) s += part[(size_t)sl * PQ + i];
  C[i] = s;
}
