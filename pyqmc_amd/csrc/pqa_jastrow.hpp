// One- and two-body spin Jastrow kernels — one walker per wavefront, lanes over the other
// electrons / the ions ("distance sweep").
//
// Reference semantics: pyqmc/wf/jastrowspin.py (recompute :56-109, value :251-255, testvalue
// :387-419, gradient_value :296-340, gradient_laplacian :342-385, updateinternals :111-137) with
// the radial functions of pyqmc/wf/func3d.py (PolyPade :25-49, CutoffCusp :125-182, zero outside
// rcut :299-324).
//
// Deliberate difference: the reference keeps per-electron partial sums (_a_partial, _b_partial:
// 53 KB per walker at 64 electrons) and patches N rows of them on every accepted move.  Here the
// one-electron sum U_e(r) = sum_I c^a a(|r-R_I|) + sum_{j!=e} c^b b(|r-r_j|) is re-evaluated at the
// old position from the stored walker coordinates instead (same arithmetic, no drift, 100x less
// state to stream), and only the reference's public sums _avalues/_bvalues are maintained.
#pragma once
#include "pqa_common.hpp"
inline namespace PQA_SYNC_NS {  // (PQA_WSYNC flavour: pqa_common.hpp)

struct JastrowState {
  double* x;        // [W][N][3] walker coordinates (the reference's _configscurrent)
  double* avalues;  // [W][natom][na][2]
  double* bvalues;  // [W][nb][3]
};

// ---------------------------------------------------------------- radial basis functions
// All functions of one basis share rcut (func3d.py:289-291), so everything that depends only on r is
// computed once per pair (RadShared) and each function costs one reciprocal:
//   PolyPade(beta)   func3d.py:25-49    z1 = r/rcut-1, p = 3 z1^4 + 4 z1^3 + 1
//        value = (1-p)/(1+beta p),  (dU/dr)/r = -12 (1+beta) z1^2 / (rcut^2 (1+beta p)^2),
//        lap   = gfac (5 + 2/z1 - 24 beta (z1+1)^2 z1^2 / (1+beta p))
//   CutoffCusp(gamma) func3d.py:125-182  y = r/rcut, a = (y-1)^2, b = (a(y-1)+1)/3
//        value = rcut(-b/(1+gamma b) + 1/(3+gamma)),  (dU/dr)/r = -a/((1+gamma b)^2 r),
//        lap   = -2/((1+gamma b)^2 r) ((y-1 - a^2 gamma/(1+gamma b)) y + a)
// Reciprocals use v_rcp_f64 + two Newton steps (relative error ~1e-16) instead of the 12-instruction
// IEEE division sequence; r/rcut is r * (1/rcut).  Differences from the reference's arithmetic are at
// the 1e-16 level, far inside the 1e-12 parity tolerance of the Jastrow tests.
__device__ __forceinline__ double fast_rcp(double x) {
  double r = __builtin_amdgcn_rcp(x);
  r = fma(r, fma(-x, r, 1.0), r);
  r = fma(r, fma(-x, r, 1.0), r);
  return r;
}

struct RadShared {
  double r, y, z1, z12, p, omp, c0, t5, q, b, inv_r;
};

template <int MODE>
__device__ __forceinline__ RadShared rad_shared(double r, double inv_rcut) {
  RadShared s;
  s.r = r;
  s.y = r * inv_rcut;
  s.z1 = s.y - 1.0;
  s.z12 = s.z1 * s.z1;
  s.p = (3.0 * s.z12 + 4.0 * s.z1) * s.z12 + 1.0;
  s.omp = 1.0 - s.p;
  s.c0 = -12.0 * inv_rcut * inv_rcut * s.z12;
  s.b = (s.z12 * s.z1 + 1.0) * (1.0 / 3.0);
  s.inv_r = fast_rcp(r);
  s.t5 = 0.0; s.q = 0.0;
  if (MODE == 2) {
    s.t5 = 5.0 + 2.0 * fast_rcp(s.z1);
    s.q = 24.0 * s.y * s.y * s.z12;
  }
  return s;
}

// aux = 1/(3+gamma) for the cusp function (host-computed)
template <int MODE>
__device__ __forceinline__ void rad_fn(int kind, double par, double aux, double rcut, const RadShared& s, double& val,
                                       double& gfac, double& lap) {
  if (kind == 0) {
    const double obp = fast_rcp(1.0 + par * s.p);
    val = s.omp * obp;
    gfac = s.c0 * (1.0 + par) * obp * obp;
    lap = (MODE == 2) ? gfac * (s.t5 - par * s.q * obp) : 0.0;
  } else {
    const double ogb = fast_rcp(1.0 + par * s.b);
    const double c = ogb * ogb * s.inv_r;
    val = (aux - s.b * ogb) * rcut;
    gfac = -s.z12 * c;
    lap = (MODE == 2) ? -2.0 * c * ((s.z1 - s.z12 * s.z12 * par * ogb) * s.y + s.z12) : 0.0;
  }
}

// ---------------------------------------------------------------- merged Pade functions (round 4)
// The K PolyPade functions of a basis share p(r), so their coefficient-weighted sums are rational functions of p with the common
// denominator D(p) = prod_k (1 + beta_k p):
//   S1 = sum_k c_k / (1 + beta_k p)                        = N1(p) / D(p)      value      = (1 - p) S1
//   S2 = sum_k c_k (1 + beta_k) / (1 + beta_k p)^2         = N2(p) / D(p)^2    (dU/dr)/r  = c0 S2
//   S3 = sum_k c_k beta_k (1 + beta_k) / (1 + beta_k p)^3  = N3(p) / D(p)^3    lap        = c0 (t5 S2 - q S3)
// (c0, t5, q of RadShared; rad_fn's formulas summed over k).  ONE reciprocal and 3K .. 6K - 3 multiply-adds per pair instead of K
// reciprocals with two Newton steps each: the lane-per-walker pair loops are instruction bound and the reciprocals were a third of
// them.  The numerator polynomials depend on the coefficient set (spin channel; atom and spin) and are tabulated by the host
// whenever coefficients change (pqa_capi.hip: jas_merge_tables, long-double products rounded once).  All beta_k > -1 and
// 0 <= p <= 1: D has no zero, its coefficients are positive for the usual positive beta; the numerators carry the same
// cancellation between functions as the function-by-function sums, so the results agree to rounding (a few 1e-16 of the sum of
// magnitudes: tests/test_gpu_jastrow_merge.py).  KD = 3: K <= 3 (zero-padded), KD = 4: K = 4.
// The numerator record is WAVE-UNIFORM at every call site (scalar loads).  A Horner step acc * p + c with c in scalar registers is
// one v_fma_f64 with a scalar addend; left to itself the compiler copies each freshly loaded coefficient into a vector register
// pair first (its two-address v_fmac form wants the addend in the destination): three instructions per step instead of one — which
// ate most of what the merge saves.  horner_s pins the three-address form.
__device__ __forceinline__ double horner_s(double acc, double p, double c_uniform) {
  double d;
  asm("v_fma_f64 %0, %1, %2, %3" : "=v"(d) : "v"(acc), "v"(p), "s"(c_uniform));
  return d;
}
struct MergedSums { double S1, S2, S3; };
// UNI: the record pointer is wave-uniform (scalar loads, horner_s); otherwise it may differ between lanes (vector loads, plain fma)
template <int MODE, int KD, bool UNI = true>
__device__ __forceinline__ MergedSums pade_merged(const double (&Dc)[5], const double* __restrict__ q, double p) {
  auto step = [&](double acc, double c) { return UNI ? horner_s(acc, p, c) : fma(acc, p, c); };
  double D = Dc[KD], n1 = q[KD - 1];
#pragma unroll
  for (int i = KD - 1; i >= 0; --i) D = fma(D, p, Dc[i]);
#pragma unroll
  for (int i = KD - 2; i >= 0; --i) n1 = step(n1, q[i]);
  const double id = fast_rcp(D);
  MergedSums m;
  m.S1 = n1 * id; m.S2 = 0.0; m.S3 = 0.0;
  if (MODE >= 1) {
    double n2 = q[4 + 2 * KD - 2];
#pragma unroll
    for (int i = 2 * KD - 3; i >= 0; --i) n2 = step(n2, q[4 + i]);
    const double id2 = id * id;
    m.S2 = n2 * id2;
    if (MODE == 2) {
      double n3 = q[11 + 3 * KD - 3];
#pragma unroll
      for (int i = 3 * KD - 4; i >= 0; --i) n3 = step(n3, q[11 + i]);
      m.S3 = n3 * (id2 * id);
    }
  }
  return m;
}

// r = sqrt(x) by the compiler's own sequence for v_rsq_f64 (one coupled Goldschmidt step, two corrections: bitwise the library
// result for normal x) without its scaling of arguments below 2^-767 and its zero / infinity fix-up, and 1 / r from the
// half-reciprocal the sequence carries (one Newton step: relative error ~1e-16) instead of v_rcp_f64 + two steps.  x = 0 (two
// particles on one point) gives r = 0, 1 / r = +inf as sqrt + fast_rcp do.
__device__ __forceinline__ void sqrt_rinv(double x, double& r, double& rinv) {
  const double y = __builtin_amdgcn_rsq(x);
  double g = x * y, h = 0.5 * y;
  const double e = fma(-h, g, 0.5);
  g = fma(g, e, g); h = fma(h, e, h);
  double d = fma(-g, g, x);
  g = fma(d, h, g);
  d = fma(-g, g, x);
  g = fma(d, h, g);
  const double t = h + h;
  double ri = fma(t, fma(-g, t, 1.0), t);
  if (x == 0.0) { g = 0.0; ri = INFINITY; }
  r = g; rinv = ri;
}
// rad_shared with 1 / r handed in
template <int MODE>
__device__ __forceinline__ RadShared rad_shared_ri(double r, double inv_r, double inv_rcut) {
  RadShared s;
  s.r = r;
  s.y = r * inv_rcut;
  s.z1 = s.y - 1.0;
  s.z12 = s.z1 * s.z1;
  s.p = (3.0 * s.z12 + 4.0 * s.z1) * s.z12 + 1.0;
  s.omp = 1.0 - s.p;
  s.c0 = -12.0 * inv_rcut * inv_rcut * s.z12;
  s.b = (s.z12 * s.z1 + 1.0) * (1.0 / 3.0);
  s.inv_r = inv_r;
  s.t5 = 0.0; s.q = 0.0;
  if (MODE == 2) {
    s.t5 = 5.0 + 2.0 * fast_rcp(s.z1);
    s.q = 24.0 * s.y * s.y * s.z12;
  }
  return s;
}

// value of one radial function for r < rcut (sums maintenance: recompute / protocol update)
__device__ __forceinline__ double jas_value1(int kind, double par, double aux, double rcut, double r) {
  const RadShared s = rad_shared<0>(r, 1.0 / rcut);
  double v, g, l;
  rad_fn<0>(kind, par, aux, rcut, s, v, g, l);
  return v;
}

// ---------------------------------------------------------------- three-body term
// P_e(r) = sum_{j != e} sum_I sum_{klm} C_{Iklm,s} a_k(|r-R_I|) a_l(|r_j-R_I|) b_m(|r-r_j|),  s = [e down] + [j down]
// (three_body_jastrow.py:66-147; moving electron e changes U by P_e(new) - P_e(old), :323-341), with
// grad/lap w.r.t. r (:374-655):  grad = sum (grad a_k) a_l b_m + a_k a_l grad b_m,
//                                 lap  = sum (lap a_k) a_l b_m + 2 grad a_k . grad b_m a_l + a_k a_l lap b_m.
// Phase 1 (lanes over ions): contract C with the moving electron's a-functions into LDS tables
//   E0/Eg/El[I][l][m][s'] = sum_k C[I][k][l][m][edown+s'] * {a_k, (da_k/dr)/r, lap a_k}(r_eI)   and d_eI.
// Phase 2 (lanes over the other electrons j): loop ions, a_l(r_jI) recomputed from the stored coordinates.
// scr: natom * (3 + 6*na3*nb3) doubles of LDS.  Adds into U, g, lapU.  Block = one wave.
__device__ __forceinline__ int j3_stride(const SysDev& S) { return 3 + 6 * S.na3 * S.nb3; }

// NB3: length of the fully unrolled register arrays (>= na3, nb3)
template <int MODE, int NB3>
__device__ __forceinline__ void jas3_eval_n(const SysDev& S, const double* __restrict__ xw, int e, double rx, double ry,
                                          double rz, double* scr, double& U, double (&g)[3], double& lapU) {
  const int lane = threadIdx.x & 63;
  const int edown = e >= S.nup, na = S.na3, nb = S.nb3, str = j3_stride(S), nlm = na * nb * 2;
  const double ira = 1.0 / S.rcut_a3, irb = 1.0 / S.rcut_b3;
  if (S.natom <= 64) {
    // Phase 1 over ALL lanes.  With lanes = ions only, a molecule of three atoms had three lanes walk the na nb 2 contractions
    // one coefficient load after the other (96 dependent round trips, ~40 us per call: nearly all of k_propose / k_accept /
    // k_ecp_accum for the 50-determinant water molecule).  Now lane I evaluates its ion's radial functions, and the
    // (ion, l, m, spin) contractions are spread over the wave, each taking its ion's function values by shuffle and its na
    // coefficients with independent loads.  Same sums in the same order.
    const int I0 = lane < S.natom ? lane : 0;
    double dx = rx - S.atom_xyz[3 * I0], dy = ry - S.atom_xyz[3 * I0 + 1], dz = rz - S.atom_xyz[3 * I0 + 2];
    min_image_j(S, dx, dy, dz);
    const double r = sqrt(dx * dx + dy * dy + dz * dz);
    if (lane < S.natom) { double* row = scr + (size_t)lane * str; row[0] = dx; row[1] = dy; row[2] = dz; }
    double av[NB3], ag[NB3], al[NB3];
    const bool in = r < S.rcut_a3;
    const RadShared sh = rad_shared<2>(in ? r : 0.5 * S.rcut_a3, ira);
#pragma unroll
    for (int k = 0; k < NB3; ++k) {
      av[k] = ag[k] = al[k] = 0.0;
      if (k < na && in) rad_fn<2>(S.a3_kind[k], S.a3_param[k], S.a3_aux[k], S.rcut_a3, sh, av[k], ag[k], al[k]);
    }
    const int nitem = S.natom * nlm;
    for (int t0 = 0; t0 < nitem; t0 += 64) {
      const int t = t0 + lane;
      const bool live = t < nitem;
      const int I = live ? t / nlm : 0, rem = live ? t - I * nlm : 0;
      const int l = rem / (2 * nb), m = (rem >> 1) % nb, sp = rem & 1;
      const double* CI = S.c3 + (size_t)I * na * na * nb * 3;
      double cc[NB3];
#pragma unroll
      for (int k = 0; k < NB3; ++k) cc[k] = (k < na) ? CI[((k * na + l) * nb + m) * 3 + edown + sp] : 0.0;
      double e0 = 0.0, eg = 0.0, el = 0.0;
#pragma unroll
      for (int k = 0; k < NB3; ++k) {
        const double a0 = __shfl(av[k], I, 64), a1 = __shfl(ag[k], I, 64), a2 = __shfl(al[k], I, 64);
        if (k < na) { e0 += cc[k] * a0; eg += cc[k] * a1; el += cc[k] * a2; }
      }
      if (live) {
        double* row = scr + (size_t)I * str;
        row[3 + rem] = e0; row[3 + rem + nlm] = eg; row[3 + rem + 2 * nlm] = el;
      }
    }
  } else
  for (int I = lane; I < S.natom; I += 64) {
    double dx = rx - S.atom_xyz[3 * I], dy = ry - S.atom_xyz[3 * I + 1], dz = rz - S.atom_xyz[3 * I + 2];
    min_image_j(S, dx, dy, dz);
    const double r = sqrt(dx * dx + dy * dy + dz * dz);
    double* row = scr + (size_t)I * str;
    row[0] = dx; row[1] = dy; row[2] = dz;
    double av[NB3], ag[NB3], al[NB3];
    const bool in = r < S.rcut_a3;
    const RadShared sh = rad_shared<2>(in ? r : 0.5 * S.rcut_a3, ira);
#pragma unroll
    for (int k = 0; k < NB3; ++k) {
      av[k] = ag[k] = al[k] = 0.0;
      if (k < na && in) rad_fn<2>(S.a3_kind[k], S.a3_param[k], S.a3_aux[k], S.rcut_a3, sh, av[k], ag[k], al[k]);
    }
    const double* CI = S.c3 + (size_t)I * na * na * nb * 3;
    for (int l = 0; l < na; ++l)
      for (int m = 0; m < nb; ++m)
        for (int sp = 0; sp < 2; ++sp) {
          double e0 = 0.0, eg = 0.0, el = 0.0;
#pragma unroll
          for (int k = 0; k < NB3; ++k) {
            if (k < na) {
              const double c = CI[((k * na + l) * nb + m) * 3 + edown + sp];
              e0 += c * av[k]; eg += c * ag[k]; el += c * al[k];
            }
          }
          const int o = 3 + (l * nb + m) * 2 + sp;
          row[o] = e0; row[o + nlm] = eg; row[o + 2 * nlm] = el;
        }
  }
  PQA_WSYNC();
  double u = 0.0, gx = 0.0, gy = 0.0, gz = 0.0, lp = 0.0;
  // Phase 2, one (partner electron j, ion I) term: a_l(r_jI) against the contracted tables of ion I and the b-functions of the pair
  auto pair_ion = [&](int I, int sp, double jx, double jy, double jz, double dx, double dy, double dz, const double (&bv)[NB3],
                      const double (&bg)[NB3], const double (&bl)[NB3]) {
    const double rj = mi_norm(S, jx - S.atom_xyz[3 * I], jy - S.atom_xyz[3 * I + 1], jz - S.atom_xyz[3 * I + 2]);
    if (!(rj < S.rcut_a3)) return;
    const RadShared sha = rad_shared<0>(rj, ira);
    const double* row = scr + (size_t)I * str;
    double s0 = 0.0, sga = 0.0, sgb = 0.0, sla = 0.0, scr_ = 0.0, slb = 0.0;
    for (int l = 0; l < na; ++l) {
      double aj, t1, t2;
      rad_fn<0>(S.a3_kind[l], S.a3_param[l], S.a3_aux[l], S.rcut_a3, sha, aj, t1, t2);
#pragma unroll
      for (int m = 0; m < NB3; ++m) {
        if (m < nb) {
          const int o = 3 + (l * nb + m) * 2 + sp;
          const double e0 = row[o] * aj;
          s0 += e0 * bv[m];
          if (MODE >= 1) { sga += row[o + nlm] * aj * bv[m]; sgb += e0 * bg[m]; }
          if (MODE == 2) { sla += row[o + 2 * nlm] * aj * bv[m]; scr_ += row[o + nlm] * aj * bg[m]; slb += e0 * bl[m]; }
        }
      }
    }
    u += s0;
    if (MODE >= 1) {
      gx += row[0] * sga + dx * sgb; gy += row[1] * sga + dy * sgb; gz += row[2] * sga + dz * sgb;
    }
    if (MODE == 2) lp += sla + 2.0 * (row[0] * dx + row[1] * dy + row[2] * dz) * scr_ + slb;
  };
  // Lanes over the partner electrons; with few electrons (a molecule of 8: seven lanes busy, each walking every ion) over
  // (partner, ion) pairs instead — the pair's b-functions are then evaluated once per ion, a third more arithmetic on a chain
  // that is natom times shorter (k_propose of the 50-determinant water molecule: 8.5 us of its 16 in this loop).
  const bool by_pair = S.nelec <= 32 && S.natom > 1;
  const int nit = by_pair ? S.nelec * S.natom : S.nelec;
  for (int t = lane; t < nit; t += 64) {
    const int j = by_pair ? t / S.natom : t;
    if (j == e) continue;
    const double jx = xw[3 * j], jy = xw[3 * j + 1], jz = xw[3 * j + 2];
    double dx = rx - jx, dy = ry - jy, dz = rz - jz;
    min_image_j(S, dx, dy, dz);
    const double r = sqrt(dx * dx + dy * dy + dz * dz);
    if (!(r < S.rcut_b3)) continue;
    double bv[NB3], bg[NB3], bl[NB3];
    const RadShared shb = rad_shared<2>(r, irb);
#pragma unroll
    for (int m = 0; m < NB3; ++m) {
      bv[m] = bg[m] = bl[m] = 0.0;
      if (m < nb) rad_fn<2>(S.b3_kind[m], S.b3_param[m], S.b3_aux[m], S.rcut_b3, shb, bv[m], bg[m], bl[m]);
    }
    const int sp = j >= S.nup;
    const int i0 = by_pair ? t - j * S.natom : 0, i1 = by_pair ? i0 + 1 : S.natom;
    for (int I = i0; I < i1; ++I) pair_ion(I, sp, jx, jy, jz, dx, dy, dz, bv, bg, bl);
  }
  if (MODE <= 1) U += wave_sum(u);
  if (MODE >= 1) { g[0] += wave_sum(gx); g[1] += wave_sum(gy); g[2] += wave_sum(gz); }
  if (MODE == 2) lapU += wave_sum(lp);
  PQA_WSYNC();
}

// The usual three-body expansions have at most four functions per kind: the arrays and unrolled loops of that instantiation are
// half as long (C4 at 2 048 walkers: 1.84 -> 2.03 M walker-steps/s); same operations in the same order either way.
template <int MODE>
__device__ __forceinline__ void jas3_eval(const SysDev& S, const double* __restrict__ xw, int e, double rx, double ry,
                                          double rz, double* scr, double& U, double (&g)[3], double& lapU) {
  if (S.na3 <= 4 && S.nb3 <= 4) jas3_eval_n<MODE, 4>(S, xw, e, rx, ry, rz, scr, U, g, lapU);
  else jas3_eval_n<MODE, PQA_MAXBAS3>(S, xw, e, rx, ry, rz, scr, U, g, lapU);
}

// U_e(r), grad U_e, lap U_e (bare laplacian, without |grad|^2) for electron e placed at r, against
// the walker coordinates xw (electron e itself skipped).  MODE 0: value; 1: value+grad; 2: grad+lap.
// parts: bit 0 = one/two-body terms (JastrowSpin), bit 1 = three-body term (needs scr, see jas3_eval).
// PBC = false compiles the minimal-image code out (open-boundary instantiations of the hot kernels)
template <int MODE, bool PBC = true>
__device__ __forceinline__ void jas_eval(const SysDev& S, const double* __restrict__ xw, int e, double rx, double ry,
                                         double rz, double& U, double (&g)[3], double& lapU, int parts = 1,
                                         double* scr = nullptr) {
  const int lane = threadIdx.x & 63;
  const int edown = e >= S.nup;
  const double irb = 1.0 / S.rcut_b, ira = 1.0 / S.rcut_a;
  double u = 0.0, gx = 0.0, gy = 0.0, gz = 0.0, lp = 0.0;
  if (parts & 1) {
  for (int j = lane; j < S.nelec; j += 64) {
    if (j == e) continue;
    double dx = rx - xw[3 * j], dy = ry - xw[3 * j + 1], dz = rz - xw[3 * j + 2];
    if (PBC) min_image_j(S, dx, dy, dz);
    const double r = sqrt(dx * dx + dy * dy + dz * dz);
    if (r < S.rcut_b) {
      const RadShared sh = rad_shared<MODE>(r, irb);
      const int col = edown + (j >= S.nup);
      double sg = 0.0;
      for (int l = 0; l < S.nb; ++l) {
        double v, gf, lpl;
        rad_fn<MODE>(S.b_kind[l], S.b_param[l], S.b_aux[l], S.rcut_b, sh, v, gf, lpl);
        const double c = S.bcoeff[l * 3 + col];
        u += c * v;
        if (MODE >= 1) sg += c * gf;
        if (MODE == 2) lp += c * lpl;
      }
      if (MODE >= 1) { gx += sg * dx; gy += sg * dy; gz += sg * dz; }
    }
  }
  for (int I = lane; I < S.natom; I += 64) {
    double dx = rx - S.atom_xyz[3 * I], dy = ry - S.atom_xyz[3 * I + 1], dz = rz - S.atom_xyz[3 * I + 2];
    if (PBC) min_image_j(S, dx, dy, dz);
    const double r = sqrt(dx * dx + dy * dy + dz * dz);
    if (r < S.rcut_a) {
      const RadShared sh = rad_shared<MODE>(r, ira);
      double sg = 0.0;
      for (int k = 0; k < S.na; ++k) {
        double v, gf, lpl;
        rad_fn<MODE>(S.a_kind[k], S.a_param[k], S.a_aux[k], S.rcut_a, sh, v, gf, lpl);
        const double c = S.acoeff[(I * S.na + k) * 2 + edown];
        u += c * v;
        if (MODE >= 1) sg += c * gf;
        if (MODE == 2) lp += c * lpl;
      }
      if (MODE >= 1) { gx += sg * dx; gy += sg * dy; gz += sg * dz; }
    }
  }
  }
  U = (MODE <= 1) ? wave_sum(u) : 0.0;
  g[0] = g[1] = g[2] = 0.0;
  if (MODE >= 1) { g[0] = wave_sum(gx); g[1] = wave_sum(gy); g[2] = wave_sum(gz); }
  lapU = (MODE == 2) ? wave_sum(lp) : 0.0;
  if ((parts & 2) && S.na3 > 0 && scr) jas3_eval<MODE>(S, xw, e, rx, ry, rz, scr, U, g, lapU);
}

// Commit the move of electron e of walker w to rn: patch _avalues/_bvalues with (new - old) and move
// the stored coordinate.  jastrowspin.py:131-137.
__device__ __forceinline__ void jas_commit(const SysDev& S, const JastrowState& js, long w, int e, double rx, double ry,
                                           double rz) {
  const int lane = threadIdx.x & 63;
  const int edown = e >= S.nup;
  double* xw = js.x + (size_t)w * S.nelec * 3;
  const double ox = xw[3 * e], oy = xw[3 * e + 1], oz = xw[3 * e + 2];
  if (S.nb > 0) {
    double diff[2][PQA_MAXBAS];
#pragma unroll
    for (int l = 0; l < PQA_MAXBAS; ++l) diff[0][l] = diff[1][l] = 0.0;
    for (int j = lane; j < S.nelec; j += 64) {
      if (j == e) continue;
      const double jx = xw[3 * j], jy = xw[3 * j + 1], jz = xw[3 * j + 2];
      const double rn = mi_norm(S, rx - jx, ry - jy, rz - jz);
      const double ro = mi_norm(S, ox - jx, oy - jy, oz - jz);
      const int grp = j >= S.nup;
#pragma unroll
      for (int l = 0; l < PQA_MAXBAS; ++l) {
        if (l < S.nb) {
          double vn = 0.0, vo = 0.0;
          if (rn < S.rcut_b) vn = jas_value1(S.b_kind[l], S.b_param[l], S.b_aux[l], S.rcut_b, rn);
          if (ro < S.rcut_b) vo = jas_value1(S.b_kind[l], S.b_param[l], S.b_aux[l], S.rcut_b, ro);
          if (grp) diff[1][l] += vn - vo; else diff[0][l] += vn - vo;
        }
      }
    }
    double* bv = js.bvalues + (size_t)w * S.nb * 3;
#pragma unroll
    for (int l = 0; l < PQA_MAXBAS; ++l) {
      if (l < S.nb) {
        const double d0 = wave_sum(diff[0][l]), d1 = wave_sum(diff[1][l]);
        if (lane == 0) { bv[l * 3 + edown] += d0; bv[l * 3 + edown + 1] += d1; }
      }
    }
  }
  for (int I = lane; I < S.natom; I += 64) {
    const double ax = S.atom_xyz[3 * I], ay = S.atom_xyz[3 * I + 1], az = S.atom_xyz[3 * I + 2];
    const double rn = mi_norm(S, rx - ax, ry - ay, rz - az);
    const double ro = mi_norm(S, ox - ax, oy - ay, oz - az);
    double* av = js.avalues + ((size_t)w * S.natom + I) * S.na * 2;
    for (int k = 0; k < S.na; ++k) {
      double vn = 0.0, vo = 0.0;
      if (rn < S.rcut_a) vn = jas_value1(S.a_kind[k], S.a_param[k], S.a_aux[k], S.rcut_a, rn);
      if (ro < S.rcut_a) vo = jas_value1(S.a_kind[k], S.a_param[k], S.a_aux[k], S.rcut_a, ro);
      av[k * 2 + edown] += vn - vo;
    }
  }
  __syncthreads();
  if (lane == 0) { xw[3 * e] = rx; xw[3 * e + 1] = ry; xw[3 * e + 2] = rz; }
  __syncthreads();
}

// U = sum bvalues*bcoeff + sum avalues*acoeff  (jastrowspin.py:251-255)
__device__ __forceinline__ double jas_value_wave(const SysDev& S, const JastrowState& js, long w) {
  const int lane = threadIdx.x & 63;
  double u = 0.0;
  const double* bv = js.bvalues + (size_t)w * S.nb * 3;
  for (int i = lane; i < S.nb * 3; i += 64) u += bv[i] * S.bcoeff[i];
  const double* av = js.avalues + (size_t)w * S.natom * S.na * 2;
  for (int i = lane; i < S.natom * S.na * 2; i += 64) u += av[i] * S.acoeff[i];
  return wave_sum(u);
}

template <int PQA_UNIT = 0>  // (a template so that only the units that launch it compile it)
static __global__ __launch_bounds__(64) void k_jastrow_value(SysDev S, JastrowState js, double* out) {
  const double u = jas_value_wave(S, js, blockIdx.x);
  if (threadIdx.x == 0) out[blockIdx.x] = u;
}

// _avalues / _bvalues from scratch for walker w (jastrowspin.py:80-105)
template <int PQA_UNIT = 0>  // (a template so that only the units that launch it compile it)
static __global__ __launch_bounds__(64) void k_jastrow_recompute(SysDev S, JastrowState js) {
  const long w = blockIdx.x;
  const int lane = threadIdx.x;
  const double* xw = js.x + (size_t)w * S.nelec * 3;
  for (int I = lane; I < S.natom; I += 64) {
    const double ax = S.atom_xyz[3 * I], ay = S.atom_xyz[3 * I + 1], az = S.atom_xyz[3 * I + 2];
    double* av = js.avalues + ((size_t)w * S.natom + I) * S.na * 2;
    for (int k = 0; k < S.na; ++k) {
      double sum[2] = {0.0, 0.0};
      for (int e = 0; e < S.nelec; ++e) {
        double dx = xw[3 * e] - ax, dy = xw[3 * e + 1] - ay, dz = xw[3 * e + 2] - az;
        min_image_j(S, dx, dy, dz);
        const double r = sqrt(dx * dx + dy * dy + dz * dz);
        if (r < S.rcut_a) sum[e >= S.nup] += jas_value1(S.a_kind[k], S.a_param[k], S.a_aux[k], S.rcut_a, r);
      }
      av[k * 2] = sum[0];
      av[k * 2 + 1] = sum[1];
    }
  }
  double acc[3][PQA_MAXBAS];
#pragma unroll
  for (int l = 0; l < PQA_MAXBAS; ++l) acc[0][l] = acc[1][l] = acc[2][l] = 0.0;
  for (int i = 0; i < S.nelec; ++i) {
    const double ix = xw[3 * i], iy = xw[3 * i + 1], iz = xw[3 * i + 2];
    for (int j = i + 1 + lane; j < S.nelec; j += 64) {
      double dx = ix - xw[3 * j], dy = iy - xw[3 * j + 1], dz = iz - xw[3 * j + 2];
      min_image_j(S, dx, dy, dz);
      const double r = sqrt(dx * dx + dy * dy + dz * dz);
      if (r < S.rcut_b) {
        const int t = (i >= S.nup) + (j >= S.nup);  // 0 upup, 1 updown, 2 downdown
#pragma unroll
        for (int l = 0; l < PQA_MAXBAS; ++l) {
          if (l < S.nb) {
            const double v = jas_value1(S.b_kind[l], S.b_param[l], S.b_aux[l], S.rcut_b, r);
            if (t == 0) acc[0][l] += v; else if (t == 1) acc[1][l] += v; else acc[2][l] += v;
          }
        }
      }
    }
  }
  double* bv = js.bvalues + (size_t)w * S.nb * 3;
#pragma unroll
  for (int l = 0; l < PQA_MAXBAS; ++l) {
    if (l < S.nb) {
      const double s0 = wave_sum(acc[0][l]), s1 = wave_sum(acc[1][l]), s2 = wave_sum(acc[2][l]);
      if (lane == 0) { bv[l * 3] = s0; bv[l * 3 + 1] = s1; bv[l * 3 + 2] = s2; }
    }
  }
}

// Protocol evaluations at caller-supplied points.  pts (nrow, npt, 3); walker of row r = widx[r] or r.
// mode 0: out[r*npt+q] = exp(U_e(pt) - U_e(x_e))                       (testvalue)
// mode 1: out (4,nrow): grad U_e(pt), exp(U_e(pt) - U_e(x_e))          (gradient_value)
// mode 2: out (4,nrow): grad U_e(pt), lap U_e + |grad U_e|^2           (gradient_laplacian)
template <int PQA_UNIT = 0>  // (a template so that only the units that launch it compile it)
static __global__ __launch_bounds__(64) void k_jastrow_eval(SysDev S, JastrowState js, int e, const double* __restrict__ pts,
                                                     long nrow, int npt, const int* __restrict__ widx, int mode,
                                                     int parts, double* __restrict__ out) {
  extern __shared__ double lds[];
  double* scr = lds + S.j3_off;
  const long r = blockIdx.x;
  const long w = widx ? widx[r] : r;
  const double* xw = js.x + (size_t)w * S.nelec * 3;
  double g[3], lp, U0 = 0.0, U;
  if (mode <= 1) jas_eval<0>(S, xw, e, xw[3 * e], xw[3 * e + 1], xw[3 * e + 2], U0, g, lp, parts, scr);
  for (int q = 0; q < npt; ++q) {
    const double* p = pts + (size_t)(r * npt + q) * 3;
    if (mode == 0) {
      jas_eval<0>(S, xw, e, p[0], p[1], p[2], U, g, lp, parts, scr);
      if (threadIdx.x == 0) out[r * npt + q] = exp(U - U0);
    } else if (mode == 1) {
      jas_eval<1>(S, xw, e, p[0], p[1], p[2], U, g, lp, parts, scr);
      if (threadIdx.x == 0) { out[r] = g[0]; out[nrow + r] = g[1]; out[2 * nrow + r] = g[2]; out[3 * nrow + r] = exp(U - U0); }
    } else {
      jas_eval<2>(S, xw, e, p[0], p[1], p[2], U, g, lp, parts, scr);
      if (threadIdx.x == 0) {
        out[r] = g[0]; out[nrow + r] = g[1]; out[2 * nrow + r] = g[2];
        out[3 * nrow + r] = lp + g[0] * g[0] + g[1] * g[1] + g[2] * g[2];
      }
    }
  }
}

template <int PQA_UNIT = 0>  // (a template so that only the units that launch it compile it)
static __global__ __launch_bounds__(64) void k_jastrow_update(SysDev S, JastrowState js, int e, const double* __restrict__ epos,
                                                       const uint8_t* __restrict__ mask) {
  const long w = blockIdx.x;
  if (mask && !mask[w]) return;
  jas_commit(S, js, w, e, epos[3 * w], epos[3 * w + 1], epos[3 * w + 2]);
}

// three-body log value U3 = 1/2 sum_e P_e(x_e)  (three_body_jastrow.py:98-101)
template <int PQA_UNIT = 0>  // (a template so that only the units that launch it compile it)
static __global__ __launch_bounds__(64) void k_j3_value(SysDev S, JastrowState js, double* __restrict__ out) {
  extern __shared__ double lds[];
  const long w = blockIdx.x;
  const double* xw = js.x + (size_t)w * S.nelec * 3;
  double tot = 0.0;
  for (int e = 0; e < S.nelec; ++e) {
    double U = 0.0, g[3] = {0.0, 0.0, 0.0}, lp = 0.0;
    jas3_eval<0>(S, xw, e, xw[3 * e], xw[3 * e + 1], xw[3 * e + 2], lds + S.j3_off, U, g, lp);
    tot += 0.5 * U;
  }
  if (threadIdx.x == 0) out[w] = tot;
}

// move the stored coordinate of electron e for the masked walkers (handles without a two-body factor)
template <int PQA_UNIT = 0>  // (a template so that only the units that launch it compile it)
static __global__ void k_move_x(JastrowState js, int N, int e, const double* __restrict__ epos, const uint8_t* __restrict__ mask, long W) {
  const long w = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (w >= W || (mask && !mask[w])) return;
  double* x = js.x + ((size_t)w * N + e) * 3;
  x[0] = epos[3 * w]; x[1] = epos[3 * w + 1]; x[2] = epos[3 * w + 2];
}

// ---------------------------------------------------------------- three-body parameter gradient
// dU/dc[I][k][l][m][sp] = 1/2 (X + X^T_kl),  X[I][k][l][m][sp] = sum over pairs (i,j) of spin class sp
//   a_k(r_iI) a_l(r_jI) b_m(r_ij):  sp 0 = up-up (i<j), 1 = up (i) - down (j), 2 = down-down (i<j)
// (ThreeBodyJastrow.pgradient, three_body_jastrow.py:657-719).  One wave per walker; LDS: a-values of every electron
// [N][natom][na] followed by b-values of every pair [N(N-1)/2][nb] (row-major upper triangle); out (W, natom, na, na, nb, 3).
template <int PQA_UNIT = 0>  // (a template so that only the units that launch it compile it)
static __global__ __launch_bounds__(64) void k_j3_pgrad(SysDev S, JastrowState js, double* __restrict__ out) {
  extern __shared__ double lds[];
  const long w = blockIdx.x;
  const int lane = threadIdx.x, N = S.nelec, A = S.natom, na = S.na3, nb = S.nb3;
  const double* xw = js.x + (size_t)w * N * 3;
  double* av = lds;
  double* bv = lds + (size_t)N * A * na;
  for (int q = lane; q < N * A; q += 64) {
    const int e = q / A, I = q % A;
    const double r = mi_norm(S, xw[3 * e] - S.atom_xyz[3 * I], xw[3 * e + 1] - S.atom_xyz[3 * I + 1], xw[3 * e + 2] - S.atom_xyz[3 * I + 2]);
    for (int k = 0; k < na; ++k)
      av[(size_t)q * na + k] = (r < S.rcut_a3) ? jas_value1(S.a3_kind[k], S.a3_param[k], S.a3_aux[k], S.rcut_a3, r) : 0.0;
  }
  const int npair = N * (N - 1) / 2;
  for (int p = lane; p < npair; p += 64) {
    int i = 0, rem = p;
    while (rem >= N - 1 - i) { rem -= N - 1 - i; ++i; }
    const int j = i + 1 + rem;
    const double r = mi_norm(S, xw[3 * i] - xw[3 * j], xw[3 * i + 1] - xw[3 * j + 1], xw[3 * i + 2] - xw[3 * j + 2]);
    for (int m = 0; m < nb; ++m)
      bv[(size_t)p * nb + m] = (r < S.rcut_b3) ? jas_value1(S.b3_kind[m], S.b3_param[m], S.b3_aux[m], S.rcut_b3, r) : 0.0;
  }
  __syncthreads();
  const int E = A * na * na * nb * 3;
  for (int idx = lane; idx < E; idx += 64) {
    int t = idx;
    const int sp = t % 3; t /= 3;
    const int m = t % nb; t /= nb;
    const int l = t % na; t /= na;
    const int k = t % na; t /= na;
    const int I = t;
    // pair ranges of the spin class: i in [i0,i1), j in [max(i+1,j0), j1)
    const int i0 = (sp == 2) ? S.nup : 0, i1 = (sp == 0 || sp == 1) ? S.nup : N;
    const int j0 = (sp == 0) ? 0 : S.nup, j1 = (sp == 0) ? S.nup : N;
    double acc = 0.0;
    for (int i = i0; i < i1; ++i) {
      const double aik = av[((size_t)i * A + I) * na + k], ail = av[((size_t)i * A + I) * na + l];
      const long rowoff = (long)i * (N - 1) - (long)i * (i - 1) / 2 - (i + 1);  // pair (i,j) -> rowoff + j
      for (int j = (j0 > i + 1 ? j0 : i + 1); j < j1; ++j) {
        const double b = bv[(size_t)(rowoff + j) * nb + m];
        acc += 0.5 * (aik * av[((size_t)j * A + I) * na + l] + ail * av[((size_t)j * A + I) * na + k]) * b;
      }
    }
    out[(size_t)w * E + idx] = acc;
  }
}
}  // inline namespace PQA_SYNC_NS
