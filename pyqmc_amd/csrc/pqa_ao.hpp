// GTO atomic-orbital evaluation and the fused AO->MO contraction (fp64 MFMA).
//
// What is computed follows the reference's in-repo evaluator pyqmc/wf/numba/gto.py:
//   value  mol_eval_gto      :89-136     chi = S_lm(r-R_A) * R(|r-R_A|^2)
//   grad   mol_eval_gto_grad :139-194    d chi = dS R + S dR,  dR_i = -2 a x_i c e^{-a r^2}
//   lap    mol_eval_gto_lap  :197-254    lap chi = S sum 2a(2a r^2-3) c e^{-a r^2} + 2 dS.dR
// with real solid harmonics as numba/spherical_harmonics.py:40-200 (l=1 ordered x,y,z) and the
// contraction orbitals.py:95-96 (ao.dot(C)).  How it is computed is MI355X-specific:
// one lane per point for the transcendental work, shell tables read through the scalar
// cache, AO tile staged in LDS (XOR-swizzled, conflict-free for ds_read_b64), contraction on
// v_mfma_f64_16x16x4_f64.
#pragma once
#include "pqa_common.hpp"
#include "pqa_sph_high.hpp"

#define PQA_STR_(x) #x
#define PQA_UNROLL_N(n) _Pragma(PQA_STR_(unroll n))  // (a macro inside '#pragma unroll' does not survive -save-temps builds)
#ifndef PQA_PRIM_UNROLL
#define PQA_PRIM_UNROLL 1
#endif
// exp of a non-positive argument (the Gaussians' -alpha r^2): round(x log2 e) and a two-term Cody-Waite reduction, a degree-11
// polynomial on |r| <= ln 2 / 2 (near-minimax fit of (e^r - 1 - r) / r^2, so e^0 = 1 exactly), ldexp.  20 instructions where the
// library routine takes 24 (it also serves positive arguments: two compare-and-select pairs for overflow / underflow); max error
// 1.3 ulp over [-745, 0] against long-double expl (host check with the same constants and fma order, 4e7 arguments).
__device__ __forceinline__ double exp_neg(double x) {
  x = fmax(x, -800.0);
  const double k = __builtin_rint(x * 1.4426950408889634074);
  double r = fma(-k, 6.93147180369123816490e-01, x);
  r = fma(-k, 1.90821492927058770002e-10, r);
  double p = 2.519062899249436e-08;
  p = fma(p, r, 2.761249479944321e-07);
  p = fma(p, r, 2.7557019442003614e-06);
  p = fma(p, r, 2.480153911822949e-05);
  p = fma(p, r, 0.0001984127010654779);
  p = fma(p, r, 0.0013888888904182626);
  p = fma(p, r, 0.008333333333235474);
  p = fma(p, r, 0.04166666666665442);
  p = fma(p, r, 0.1666666666666677);
  p = fma(p, r, 0.5);
  p = fma(p, r, 1.0);
  p = fma(p, r, 1.0);
  return ldexp(p, (int)k);
}
// Three exponentials side by side, every step of exp_neg for the three arguments in turn: a kernel with one or two waves per SIMD hides
// the latency of the dependent fma chain only with independent chains in the SAME thread, and the compiler does not interleave the
// unrolled calls by itself (pqa_res8.hpp).  Bit for bit exp_neg of each argument.
__device__ __forceinline__ void exp_neg3(double x0, double x1, double x2, double& e0, double& e1, double& e2) {
  x0 = fmax(x0, -800.0); x1 = fmax(x1, -800.0); x2 = fmax(x2, -800.0);
  const double k0 = __builtin_rint(x0 * 1.4426950408889634074), k1 = __builtin_rint(x1 * 1.4426950408889634074),
               k2 = __builtin_rint(x2 * 1.4426950408889634074);
  double r0 = fma(-k0, 6.93147180369123816490e-01, x0), r1 = fma(-k1, 6.93147180369123816490e-01, x1), r2 = fma(-k2, 6.93147180369123816490e-01, x2);
  r0 = fma(-k0, 1.90821492927058770002e-10, r0); r1 = fma(-k1, 1.90821492927058770002e-10, r1); r2 = fma(-k2, 1.90821492927058770002e-10, r2);
  double p0 = 2.519062899249436e-08, p1 = 2.519062899249436e-08, p2 = 2.519062899249436e-08;
#define PQA_E3(c) p0 = fma(p0, r0, c); p1 = fma(p1, r1, c); p2 = fma(p2, r2, c);
  PQA_E3(2.761249479944321e-07) PQA_E3(2.7557019442003614e-06) PQA_E3(2.480153911822949e-05) PQA_E3(0.0001984127010654779)
  PQA_E3(0.0013888888904182626) PQA_E3(0.008333333333235474) PQA_E3(0.04166666666665442) PQA_E3(0.1666666666666677)
  PQA_E3(0.5) PQA_E3(1.0) PQA_E3(1.0)
#undef PQA_E3
  e0 = ldexp(p0, (int)k0); e1 = ldexp(p1, (int)k1); e2 = ldexp(p2, (int)k2);
}
#ifndef PQA_EXP_NEG
#define PQA_EXP_NEG 1  // 0: library exp (A/B: k_orb<5> 129.2 -> 127.3 us, k_orb<1> 821 -> 790, periodic k_orb_wide 193 -> 187)
#endif
#if PQA_EXP_NEG
#define PQA_EXP(x) exp_neg(x)
#else
#define PQA_EXP(x) exp(x)
#endif
// Primitive screening: skip a primitive with alpha r^2 > PQA_PRIM_CUT (it contributes < 2e-22 of its coefficient).  The test
// is per lane and the exp sequence is only saved when EVERY lane of the wave skips.
//  * open systems (-DPQA_PRIM_SCREEN=1, off): the 64 points of a wave are 64 different walkers, some lane is nearly always
//    close, and the compare + branch cost more than the rare skip saves (round 2, tools/scratch/ab_screen.sh: k_orb<5>
//    136.7 -> 141.4 us per 65536 points).  Round 3 tried it on the values-only launches with the ECP points listed atom by atom
//    (every tile within a few bohr of one atom, so distant atoms' tight primitives drop out for the whole wave): 826 -> 968 us
//    per launch of 1.2 M points — the branch per primitive breaks up the interleaved exp sequences of neighbouring primitives,
//    which costs more than the skipped ones save (walker-major points with the test: 1 035 us).
//  * periodic cells (always on, SCREEN template argument): every lane walks the images of an atom NEAREST FIRST, so from the
//    second image on all lanes sit at r^2 >~ (half the cell)^2 and the tight primitives of a contracted shell drop out for
//    the whole wave.
#ifndef PQA_PRIM_SCREEN
#define PQA_PRIM_SCREEN 0
#endif
#define PQA_PRIM_CUT 50.0

// p-th point lives at base + (p / group) * group_stride + (p % group) * 3
// count (optional): the number of points lives on the device — the launch covers an upper bound P and the blocks beyond *count leave at
// once (the ECP point lists of small shards: sizing the launch from the host costs a device -> host round trip per energy evaluation)
struct PointAddr {
  const double* base;
  int group;
  long group_stride;
  const long* count = nullptr;
};
__device__ __forceinline__ long point_count(const PointAddr& a, long P) {
  if (!a.count) return P;
  const long c = *a.count;
  return c < P ? c : P;
}
__device__ __forceinline__ void load_point(const PointAddr& a, long p, double& x, double& y, double& z) {
  const double* q = a.base + (p / a.group) * a.group_stride + (p % a.group) * 3;
  x = q[0]; y = q[1]; z = q[2];
}

// normalisation constants of the orthonormal real spherical harmonics
#define SH_S0 0.28209479177387814   // 1/(2 sqrt(pi))
#define SH_P1 0.4886025119029199    // sqrt(3/(4 pi))
#define SH_DXY 1.0925484305920792   // 1/2 sqrt(15/pi)
#define SH_DZ2 0.31539156525252005  // 1/4 sqrt(5/pi)
#define SH_DX2 0.5462742152960396   // 1/4 sqrt(15/pi)
#define SH_F3 0.5900435899266435    // 1/4 sqrt(35/(2 pi))
#define SH_F2 2.890611442640554     // 1/2 sqrt(105/pi)
#define SH_F1 0.4570457994644658    // 1/4 sqrt(21/(2 pi))
#define SH_F0 0.3731763325901154    // 1/4 sqrt(7/pi)
#define SH_F2C 1.445305721320277    // 1/4 sqrt(105/pi)

// g and h shells (l = 4, 5; the reference supports l <= 5, numba/gto.py:107-118): real solid harmonic m of shell l and
// its gradient from the generated monomial tables (tools/gen_solid_harmonics.py).  Deliberately a compact run-time loop
// instead of unrolled expressions: these shells appear only in quadruple-zeta and larger basis sets, and written out they
// would set the register high-water mark of every orbital kernel.  l, m are wave-uniform: the table comes through the
// scalar cache.
__device__ __forceinline__ double sph_ipow(double x, int n) {
  double r = 1.0;
#pragma unroll 1
  for (int q = 0; q < n; ++q) r *= x;
  return r;
}
__device__ __forceinline__ void sph_high(int l, int m, double x, double y, double z, double& s, double& sx, double& sy, double& sz) {
  s = sx = sy = sz = 0.0;
  const int t1 = SPH_HI_OFF[l - 4][m + 1];
#pragma unroll 1
  for (int t = SPH_HI_OFF[l - 4][m]; t < t1; ++t) {
    const SphTerm q = SPH_HI_TERM[t];
    const double xa = sph_ipow(x, q.i - 1), ya = sph_ipow(y, q.j - 1), za = sph_ipow(z, q.k - 1);  // x^(i-1) ... (1 for i <= 1)
    const double xi = q.i ? xa * x : 1.0, yj = q.j ? ya * y : 1.0, zk = q.k ? za * z : 1.0;
    s += q.c * xi * yj * zk;
    if (q.i) sx += q.c * q.i * xa * yj * zk;
    if (q.j) sy += q.c * q.j * xi * ya * zk;
    if (q.k) sz += q.c * q.k * xi * yj * za;
  }
}

// Evaluate one contracted shell at displacement (x,y,z) from its centre and hand each of its
// 2l+1 functions to sink(m, value, dx, dy, dz, lap).  NCOMP = 1 | 4 | 5 selects how much is computed.
// LMAX < 3 compiles the f-shell branch out (callers that know the basis has none: its seven functions set the register
// high-water mark of the routine); LMAX < 5 the g/h branch (the periodic kernels: their register budget is exhausted — with it
// k_orb<5,..,PBC=1> went from 213 to 228 VGPRs, 2 to 1 waves per SIMD and +70 % time; periodic cells take l <= 3).
// Radial value of a contracted shell from its table (SysDev::rtab): x = r^2 is located by the binary exponent and the top three mantissa bits
// of y = x + 2^-7 — interval (octave, eighth), local variable u in [-1, 1) — and R(x) = sum c e^{-a x} is a degree-9 polynomial in u (Horner).
// One exponential per primitive (20 instructions each, 9 primitives in a cc-pVDZ s or p contraction) becomes ~12 instructions of indexing, five
// 16-byte gathers from an L2-resident table and 9 fused multiply-adds.  Beyond the last interval every primitive is below 1e-20 of its
// coefficient: zero.  Values only — the value-only orbital kernel of the ECP quadrature (k_orb<1>) is where the exponentials were 40 % of the
// instructions.  With the derivative sums as well (30 coefficients, 15 gathers per lane and shell) the texture path takes as long as the
// exponentials did: k_orb<5> 3.83 -> 3.9 ms, the resident sweep unchanged, so those keep the primitive sums.  (With every
// lane on ONE record k_orb<1> is only 5 % faster still: the gathers are not what bounds it now.)
#define PQA_RT_X0 0.0078125  // 2^-7
#define PQA_RT_NSUB 8
#define PQA_RT_DEG 9
#define PQA_RT_REC (PQA_RT_DEG + 1)  // doubles per interval
#define PQA_RT_MINP 3        // shells with fewer primitives keep their exponentials
#define PQA_RT_MAXERR 2e-15  // largest fit error (relative to sum |c|) a table may have (build_radial_tables)
__device__ __forceinline__ double radial_tab(const double* __restrict__ tab, int nint, double r2) {
  const double y = r2 + PQA_RT_X0;
  const int E = __builtin_amdgcn_frexp_exp(y);       // y = m 2^E, m in [0.5, 1)
  const double m = __builtin_amdgcn_frexp_mant(y);
  const double t = fma(m, 2.0 * PQA_RT_NSUB, -(double)PQA_RT_NSUB);  // (2 m - 1) NSUB in [0, NSUB)
  const double tj = floor(t);
  const double u = fma(t - tj, 2.0, -1.0);
  int idx = (E + 6) * PQA_RT_NSUB + (int)tj;
  const bool in = idx < nint;
  idx = in ? idx : nint - 1;
  const double2* rec = reinterpret_cast<const double2*>(tab + (size_t)idx * PQA_RT_REC);
  double c[PQA_RT_REC];
#pragma unroll
  for (int q = 0; q < PQA_RT_REC / 2; ++q) { const double2 v = rec[q]; c[2 * q] = v.x; c[2 * q + 1] = v.y; }
  double f0 = c[PQA_RT_DEG];
#pragma unroll
  for (int q = PQA_RT_DEG - 1; q >= 0; --q) f0 = fma(f0, u, c[q]);
  return in ? f0 : 0.0;
}
// The angular half: the shell's 2l+1 functions from its radial sums R = sum c e^{-a r^2}, dRs = sum a c e^{-a r^2}, lapR = sum 2a (2a r^2 - 3) c e^{-a r^2}.
template <int NCOMP, int LMAX = 5, class Sink>
__device__ __forceinline__ void shell_angular(int l, double x, double y, double z, double R, double dRs, double lapR, Sink&& sink);
template <int NCOMP, int LMAX = 5, bool SCREEN = false, class Sink>
__device__ __forceinline__ void shell_eval(int l, double x, double y, double z, const double* __restrict__ pexp,
                                           const double* __restrict__ pcoef, int np, Sink&& sink) {
  const double r2 = x * x + y * y + z * z;
  double R = 0.0, dRs = 0.0, lapR = 0.0;
  PQA_UNROLL_N(PQA_PRIM_UNROLL)
  for (int p = 0; p < np; ++p) {
    const double a = pexp[p];
    if ((SCREEN || PQA_PRIM_SCREEN) && a * r2 > PQA_PRIM_CUT) continue;
#ifdef PQA_ABL_NOEXP  // ablation builds only (tools/scratch/abl_pbc.sh): wrong values, kernel timing only
    const double t = pcoef[p] * (1.0 - 1e-3 * a * r2);
#else
    const double t = pcoef[p] * PQA_EXP(-a * r2);
#endif
    R += t;
    if (NCOMP > 1) dRs += a * t;
    if (NCOMP == 5) lapR += t * (2.0 * a) * (2.0 * a * r2 - 3.0);
  }
  shell_angular<NCOMP, LMAX>(l, x, y, z, R, dRs, lapR, sink);
}
// The values of a contracted shell through its radial table (open systems; callers check SysDev::shell_rt)
template <int LMAX = 5, class Sink>
__device__ __forceinline__ void shell_eval_tab(int l, double x, double y, double z, const double* __restrict__ tab, int nint, Sink&& sink) {
  shell_angular<1, LMAX>(l, x, y, z, radial_tab(tab, nint, x * x + y * y + z * z), 0.0, 0.0, sink);
}
// The same sums with the primitives three at a time (exp_neg3: independent chains for kernels with one or two waves per SIMD); the
// primitives are added in the same order as in shell_eval.
template <int NCOMP, int LMAX = 5, class Sink>
__device__ __forceinline__ void shell_eval3(int l, double x, double y, double z, const double* __restrict__ pexp,
                                            const double* __restrict__ pcoef, int np, Sink&& sink) {
  const double r2 = x * x + y * y + z * z;
  double R = 0.0, dRs = 0.0, lapR = 0.0;
  auto add = [&](double a, double c, double e) __attribute__((always_inline)) {
    const double t = c * e;
    R += t;
    if (NCOMP > 1) dRs += a * t;
    if (NCOMP == 5) lapR += t * (2.0 * a) * (2.0 * a * r2 - 3.0);
  };
  int p = 0;
  if (np >= 3) {  // (the next triple's exponents and coefficients are requested before this one's exponentials: their LDS round trip runs under the chains)
    double a0 = pexp[0], a1 = pexp[1], a2 = pexp[2], c0 = pcoef[0], c1 = pcoef[1], c2 = pcoef[2];
#pragma unroll 1
    for (; p + 3 <= np; p += 3) {
      const int pn = min(p + 3, np - 3);
      const double na0 = pexp[pn], na1 = pexp[pn + 1], na2 = pexp[pn + 2], nc0 = pcoef[pn], nc1 = pcoef[pn + 1], nc2 = pcoef[pn + 2];
      double e0, e1, e2;
      exp_neg3(-a0 * r2, -a1 * r2, -a2 * r2, e0, e1, e2);
      add(a0, c0, e0); add(a1, c1, e1); add(a2, c2, e2);
      a0 = na0; a1 = na1; a2 = na2; c0 = nc0; c1 = nc1; c2 = nc2;
    }
  }
  if (p + 2 == np) {  // two left: the three-way routine with the spare argument at zero (e^0 discarded)
    const double a0 = pexp[p], c0 = pcoef[p], a1 = pexp[p + 1], c1 = pcoef[p + 1];
    double e0, e1, e2;
    exp_neg3(-a0 * r2, -a1 * r2, 0.0, e0, e1, e2);
    add(a0, c0, e0); add(a1, c1, e1);
  } else if (p < np) add(pexp[p], pcoef[p], exp_neg(-pexp[p] * r2));
  shell_angular<NCOMP, LMAX>(l, x, y, z, R, dRs, lapR, sink);
}
template <int NCOMP, int LMAX, class Sink>
__device__ __forceinline__ void shell_angular(int l, double x, double y, double z, double R, double dRs, double lapR, Sink&& sink) {
  dRs *= -2.0;  // grad R = dRs * (x,y,z)
  const double Rx = dRs * x, Ry = dRs * y, Rz = dRs * z;
#define EMIT(m, S, Sx, Sy, Sz)                                                                             \
  {                                                                                                        \
    const double s_ = (S), sx_ = (Sx), sy_ = (Sy), sz_ = (Sz);                                             \
    double gx_ = 0, gy_ = 0, gz_ = 0, lp_ = 0;                                                             \
    if (NCOMP > 1) { gx_ = sx_ * R + s_ * Rx; gy_ = sy_ * R + s_ * Ry; gz_ = sz_ * R + s_ * Rz; }          \
    if (NCOMP == 5) lp_ = s_ * lapR + 2.0 * (sx_ * Rx + sy_ * Ry + sz_ * Rz);                              \
    sink(m, s_ * R, gx_, gy_, gz_, lp_);                                                                   \
  }
  switch (l) {
    case 0:
      EMIT(0, SH_S0, 0.0, 0.0, 0.0);
      break;
    case 1:
      EMIT(0, SH_P1 * x, SH_P1, 0.0, 0.0);
      EMIT(1, SH_P1 * y, 0.0, SH_P1, 0.0);
      EMIT(2, SH_P1 * z, 0.0, 0.0, SH_P1);
      break;
    case 2:
      EMIT(0, SH_DXY * x * y, SH_DXY * y, SH_DXY * x, 0.0);
      EMIT(1, SH_DXY * y * z, 0.0, SH_DXY * z, SH_DXY * y);
      EMIT(2, SH_DZ2 * (2.0 * z * z - x * x - y * y), -2.0 * SH_DZ2 * x, -2.0 * SH_DZ2 * y, 4.0 * SH_DZ2 * z);
      EMIT(3, SH_DXY * x * z, SH_DXY * z, 0.0, SH_DXY * x);
      EMIT(4, SH_DX2 * (x * x - y * y), 2.0 * SH_DX2 * x, -2.0 * SH_DX2 * y, 0.0);
      break;
    default: if (LMAX >= 3) {
      if (LMAX >= 5 && l > 3) {
#pragma unroll 1
        for (int m = 0; m < 2 * l + 1; ++m) {
          double s4, s4x, s4y, s4z;
          sph_high(l, m, x, y, z, s4, s4x, s4y, s4z);
          EMIT(m, s4, s4x, s4y, s4z);
        }
        break;
      }
      const double x2 = x * x, y2 = y * y, z2 = z * z;
      EMIT(0, SH_F3 * y * (3.0 * x2 - y2), SH_F3 * 6.0 * x * y, SH_F3 * 3.0 * (x2 - y2), 0.0);
      EMIT(1, SH_F2 * x * y * z, SH_F2 * y * z, SH_F2 * x * z, SH_F2 * x * y);
      EMIT(2, SH_F1 * y * (4.0 * z2 - x2 - y2), -2.0 * SH_F1 * x * y, SH_F1 * (4.0 * z2 - x2 - 3.0 * y2),
           8.0 * SH_F1 * y * z);
      EMIT(3, SH_F0 * z * (2.0 * z2 - 3.0 * x2 - 3.0 * y2), -6.0 * SH_F0 * x * z, -6.0 * SH_F0 * y * z,
           SH_F0 * (6.0 * z2 - 3.0 * x2 - 3.0 * y2));
      EMIT(4, SH_F1 * x * (4.0 * z2 - x2 - y2), SH_F1 * (4.0 * z2 - 3.0 * x2 - y2), -2.0 * SH_F1 * x * y,
           8.0 * SH_F1 * x * z);
      EMIT(5, SH_F2C * z * (x2 - y2), 2.0 * SH_F2C * x * z, -2.0 * SH_F2C * y * z, SH_F2C * (x2 - y2));
      EMIT(6, SH_F3 * x * (x2 - 3.0 * y2), SH_F3 * 3.0 * (x2 - y2), -SH_F3 * 6.0 * x * y, 0.0);
    }
  }
#undef EMIT
}

// ---------------------------------------------------------------- lattice-summed shells
// Periodic AO = sum over the images of its centre that lie within the atom's and the shell's r^2 cut-offs
// (numba/pbcgto.py:205-225, :340-365, :470-506).  The displacement point - atom is first folded to the cell-centred
// parallelepiped (|d0| <= half the longest body diagonal), so only the cell translations Ls[j] with
// |Ls[j]| <= sqrt(atom_cut) + that radius can contribute: num_Ls[atom] of them, a few dozen instead of hundreds.
// Which of those pass the atom test (and the reference's membership rule, see include/pyqmc_amd.h) is worked out
// ONCE per (point, atom) into a 64-bit mask that all shells of the atom reuse.  sh / ia / l / j are wave-uniform at
// every call site: tables come through the scalar cache, only the tests diverge between lanes.
struct PrimWrap { int w0, w1, w2; };
__device__ __forceinline__ PrimWrap prim_wrap(const SysDev& S, double px, double py, double pz) {
  PrimWrap w = {0, 0, 0};  // W = floor(r . inv(lattice_prim)) (enforce_pbc, pbc/pbc.py:37-43): membership rule only
  if (S.pb->member) {
    w.w0 = (int)floor(px * S.pb->lprim_inv[0] + py * S.pb->lprim_inv[3] + pz * S.pb->lprim_inv[6]);
    w.w1 = (int)floor(px * S.pb->lprim_inv[1] + py * S.pb->lprim_inv[4] + pz * S.pb->lprim_inv[7]);
    w.w2 = (int)floor(px * S.pb->lprim_inv[2] + py * S.pb->lprim_inv[5] + pz * S.pb->lprim_inv[8]);
  }
  return w;
}

// Image lists.  k_pbc_prepass works out, once per (point, atom), which of the atom's candidate images are admitted (atom
// cut-off and membership rule) and SORTS them by distance, nearest first, into a packed list of 16-bit image indices (4 per
// 64-bit word, PQA_IMG_END after the last).  Every shell of the atom then walks a PREFIX of that list — the images inside
// its own cut-off — and stops at the first one outside: a contracted shell with a short range looks at its 1-2 images
// instead of every image the atom's most diffuse shell needs (13 in the 2x2x2 diamond cell), and because the k-th entries of
// all lanes are their k-th NEAREST images the wave's lanes leave the loop together and the primitive screening above works.
// A lane whose list does not fit (or an atom with more than 128 candidates) carries PQA_IMG_OVF and tests every candidate
// image directly, as does the thread-per-point test kernel k_ao.
#define PQA_IMG_END 0xFFFFu
#define PQA_IMG_OVF 0xFFFEu
struct PbcCtx {
  int ia = -1;
  double x0, y0, z0;        // folded displacement point - atom
  int b0, b1, b2;           // membership index of image j = b + img_n[j] (direct tests only)
  const unsigned long long* lp = nullptr;  // this lane's list: word w at lp[w * lstride]
  long lstride = 0;
  int lcap = 0;             // entries the list can hold (bounds the walk)
  bool ovf = true;          // no list: test every candidate image
  double cf = 1.0, sf = 0.0;   // twisted: (cos, sin)(k_t . f . lattice) of the fold f applied to point - atom
};

__device__ __forceinline__ bool pbc_image_ok(const SysDev& S, const PbcCtx& c, int j, double r2) {
  if (r2 > S.pb->atom_cut[c.ia]) return false;
  if (S.pb->member) {  // would the reference have looked at this image?
    const int side = 2 * S.pb->member_M + 1;
    const int n0 = c.b0 + S.pb->img_n[3 * j], n1 = c.b1 + S.pb->img_n[3 * j + 1], n2 = c.b2 + S.pb->img_n[3 * j + 2];
    if ((unsigned)n0 >= (unsigned)side || (unsigned)n1 >= (unsigned)side || (unsigned)n2 >= (unsigned)side) return false;
    if (!S.pb->member[((size_t)S.pb->member_class[c.ia] * side + n0) * side * side + n1 * side + n2]) return false;
  }
  return true;
}

// fold point - atom into the cell-centred parallelepiped; membership base and twist phase of that fold
__device__ __forceinline__ void pbc_ctx_base(const SysDev& S, PbcCtx& c, int ia, double x, double y, double z, PrimWrap pw) {
  c.ia = ia;
  const double f0 = floor(x * S.pb->linv[0] + y * S.pb->linv[3] + z * S.pb->linv[6] + 0.5);
  const double f1 = floor(x * S.pb->linv[1] + y * S.pb->linv[4] + z * S.pb->linv[7] + 0.5);
  const double f2 = floor(x * S.pb->linv[2] + y * S.pb->linv[5] + z * S.pb->linv[8] + 0.5);
  c.x0 = x - (f0 * S.pb->lat[0] + f1 * S.pb->lat[3] + f2 * S.pb->lat[6]);
  c.y0 = y - (f0 * S.pb->lat[1] + f1 * S.pb->lat[4] + f2 * S.pb->lat[7]);
  c.z0 = z - (f0 * S.pb->lat[2] + f1 * S.pb->lat[5] + f2 * S.pb->lat[8]);
  if (S.pb->twist) sincos(f0 * S.pb->ktl[0] + f1 * S.pb->ktl[1] + f2 * S.pb->ktl[2], &c.sf, &c.cf);
  if (S.pb->member) {  // atom image R_A + (f + m) . lattice  <->  primitive translation atom_n + (f + m) . supercell
    const int i0 = (int)f0, i1 = (int)f1, i2 = (int)f2;
    c.b0 = S.pb->atom_n[3 * ia] + i0 * S.pb->supercell[0] + i1 * S.pb->supercell[3] + i2 * S.pb->supercell[6] - pw.w0 + S.pb->member_M;
    c.b1 = S.pb->atom_n[3 * ia + 1] + i0 * S.pb->supercell[1] + i1 * S.pb->supercell[4] + i2 * S.pb->supercell[7] - pw.w1 + S.pb->member_M;
    c.b2 = S.pb->atom_n[3 * ia + 2] + i0 * S.pb->supercell[2] + i1 * S.pb->supercell[5] + i2 * S.pb->supercell[8] - pw.w2 + S.pb->member_M;
  }
}
// list-less context (k_ao; lanes flagged PQA_IMG_OVF)
__device__ __forceinline__ void pbc_ctx_update(const SysDev& S, PbcCtx& c, int ia, double x, double y, double z, PrimWrap pw) {
  if (c.ia == ia) return;
  pbc_ctx_base(S, c, ia, x, y, z, pw);
  c.ovf = true;
}

// Twisted cells (TW): the lattice sum sum_L exp(i k_t . L) phi(r - R - L) is complex; one walk over the admitted images
// accumulates its real and imaginary parts (the shell's values are evaluated once per image and weighted by cos / sin of
// the image phase) and hands them to sink(m, ...) and sink_im(m, ...).  Untwisted: sink only.
// ls(j, lx, ly, lz, ph): lattice vector (and, twisted, the (cos, sin) phase) of image j — from the kernel's LDS copy where it has one:
// the image walk alone (list decode, a per-lane gather of the vector, r^2, the range test) was a third of k_orb<5> and two thirds
// of k_orb<1> in a periodic cell with the vectors gathered from global memory (compile-time ablation, tools/scratch/abl_pbc.sh).
template <int NCOMP, bool TW = false, int LMAX = 3, class Sink, class SinkIm, class LsGet>
__device__ __forceinline__ void shell_eval_pbc(const SysDev& S, const PbcCtx& c, int sh, int l, const double* __restrict__ pexp,
                                               const double* __restrict__ pcoef, int np, Sink&& sink, SinkIm&& sink_im, bool& accumulate,
                                               LsGet&& ls) {
  // The lattice sum is accumulated where the functions live (the lane's own column of the LDS tile; `accumulate` tells the
  // sinks to add instead of store): 7 x NCOMP running sums in registers — twice that for a twisted cell — were 70 / 140 of
  // the kernel's ~255 registers, pinned it at 2 (twisted: 1) waves per SIMD and spilled.
  accumulate = false;
#ifndef PQA_ABL_NOZERO
#pragma unroll
  for (int m = 0; m < 2 * LMAX + 1; ++m)
    if (m < 2 * l + 1) {
      sink(m, 0.0, 0.0, 0.0, 0.0, 0.0);
      if (TW) sink_im(m, 0.0, 0.0, 0.0, 0.0, 0.0);
    }
#endif
  accumulate = true;
  const int nimg = S.pb->num_Ls[c.ia];
  const double scut = S.pb->shell_cut[sh];
  auto add = [&](double xj, double yj, double zj, int j, double cj, double sj) {
    double pr = 1.0, pi = 0.0;
    if (TW) {  // exp(i k_t . (f . lattice + Ls[j])): cos and sin of the summed angle
      pr = c.cf * cj - c.sf * sj;
      pi = c.sf * cj + c.cf * sj;
    }
    shell_eval<NCOMP, LMAX, true>(l, xj, yj, zj, pexp, pcoef, np, [&](int m, double v, double gx, double gy, double gz, double lp) {
      if (TW) {
        sink_im(m, pi * v, pi * gx, pi * gy, pi * gz, pi * lp);
        v *= pr; gx *= pr; gy *= pr; gz *= pr; lp *= pr;
      }
      sink(m, v, gx, gy, gz, lp);
    });
  };
  // Each lane walks ITS OWN list of admitted images, nearest first, as far as this shell's cut-off reaches (the points of a
  // wave sit anywhere in the cell: iterating over image indices in lock-step would make every lane wait for the union of
  // all lanes' images).  Iterations = max over lanes of the number of images inside the shell's range.
  {
#ifdef PQA_ABL_NOWALK
    bool alive = false;
#else
    bool alive = !c.ovf;
#endif
    unsigned long long cur = 0ull;
    int k = 0;
    while (__any(alive)) {
      if (k >= c.lcap) alive = false;
      if (alive) {
        if ((k & 3) == 0) cur = c.lp[(size_t)(k >> 2) * c.lstride];
        const int j = (int)(cur & 0xFFFFull);
        cur >>= 16;
        if (j == (int)PQA_IMG_END) alive = false;
        else {
          double lx, ly, lz, cj = 1.0, sj = 0.0;
          ls(j, lx, ly, lz, cj, sj);
          const double xj = c.x0 - lx, yj = c.y0 - ly, zj = c.z0 - lz;
#ifdef PQA_ABL_NOADD
          if (xj * xj + yj * yj + zj * zj <= scut) { if (xj == 1.2345e300) add(xj, yj, zj, j, cj, sj); }
#else
          if (xj * xj + yj * yj + zj * zj <= scut) add(xj, yj, zj, j, cj, sj);
#endif
          else alive = false;  // sorted by distance: nothing further can be inside
        }
      }
      ++k;
    }
  }
  if (__any(c.ovf)) {  // list-less lanes: direct tests of every candidate image
    for (int j = 0; j < nimg; ++j) {
      const double xj = c.x0 - S.pb->Ls[3 * j], yj = c.y0 - S.pb->Ls[3 * j + 1], zj = c.z0 - S.pb->Ls[3 * j + 2];
      const double r2 = xj * xj + yj * yj + zj * zj;
      if (c.ovf && r2 <= scut && pbc_image_ok(S, c, j, r2))
        add(xj, yj, zj, j, TW ? S.pb->img_phase[2 * j] : 1.0, TW ? S.pb->img_phase[2 * j + 1] : 0.0);
    }
  }
  accumulate = false;
}
template <int NCOMP, int LMAX = 3, class Sink>
__device__ __forceinline__ void shell_eval_pbc(const SysDev& S, const PbcCtx& c, int sh, int l, const double* __restrict__ pexp,
                                               const double* __restrict__ pcoef, int np, Sink&& sink, bool& accumulate) {
  shell_eval_pbc<NCOMP, false, LMAX>(S, c, sh, l, pexp, pcoef, np, sink, sink, accumulate,
                               [&](int j, double& lx, double& ly, double& lz, double&, double&) { lx = S.pb->Ls[3 * j]; ly = S.pb->Ls[3 * j + 1]; lz = S.pb->Ls[3 * j + 2]; });
}

// Pre-pass of a periodic k_orb launch: thread = (point, atom).  Folds the point into the cell, folds point - atom into
// the cell-centred parallelepiped, works out which of the atom's candidate images are admitted (atom cut-off and
// membership rule) and writes them sorted by distance as the packed list described at PbcCtx — once, instead of once per
// lane group inside k_orb, and in a kernel that is not register-bound.  lst: [natom][NW][P] words.
#define PQA_PRE_CAP 32   // admitted images per (point, atom) the pre-pass can order (their 4-bit classes fill two words)
#define PQA_PRE_NT 256
#define PQA_MAXCLS 15    // distinct shell cut-offs per atom the tables hold
#define PQA_PRE_NCUT 10  // ... and the pre-pass handles (cut-offs in registers, class populations packed 6 bits each); more: direct tests
// Ordering without a sort: a shell only needs the images inside ITS cut-off to come first, and an atom's shells have a
// handful of distinct cut-offs (five in the diamond basis).  So every admitted image gets the class of the smallest shell
// cut-off that contains it, and the list is written class by class, with the lane's nearest image moved to the very front
// (so that from the second entry on all lanes are far from the centre and the primitive screening bites).  Everything
// lives in registers: an LDS sort cut the kernel's occupancy to 1.5 waves per SIMD and a selection sort in registers
// recomputed n^2 distances — both 2.4 x slower than this.
#define PQA_PRE_NWMAX 8   // list words the pre-pass can assemble in LDS (32 entries: PQA_PRE_CAP)
#define PQA_PRE_MEMB 2048 // bytes of one atom class of the membership table staged in LDS (side^3; 729 for M = 4)
template <int NCUT> struct PreFields { typedef unsigned long long type; };
template <> struct PreFields<5> { typedef unsigned type; };
// NCUT: distinct shell cut-offs per atom the instantiation handles (5: the usual basis sets; PQA_PRE_NCUT otherwise).
// Dynamic LDS: 2 x (4 NW) x PQA_PRE_NT 16-bit entries (records in arrival order, entries in list order).
template <int NCUT = PQA_PRE_NCUT, int PQA_UNIT = 0>
static __global__ __launch_bounds__(PQA_PRE_NT) void k_pbc_prepass(SysDev S, PointAddr pa, long P, int NW, double* __restrict__ d0,
                                                     unsigned long long* __restrict__ lst, double* __restrict__ theta) {
  // The candidates' lattice vectors, their membership offsets and the atom class's membership bytes are staged in LDS: as
  // per-candidate scalar / gather loads (load, wait, test) they were 79 + 4 x 13 dependent round trips per thread in the
  // 2x2x2 diamond cell — this kernel runs one wave per SIMD in a 4096-walker launch and is all latency.
  __shared__ double s_Ls[128][3];
  __shared__ int s_imgn[128][3];
  __shared__ unsigned char s_memb[PQA_PRE_MEMB];
  __shared__ double s_cut[PQA_MAXCLS + 1];
  extern __shared__ unsigned short s_pre[];   // records / entries of the list being assembled, one column per thread each
  const int ia = blockIdx.y, tid = threadIdx.x;
  const int ncls = S.pb->ncls[ia];
  const bool has_member = S.pb->member != nullptr;
  const int side = 2 * S.pb->member_M + 1, side3 = side * side * side;
  const bool memb_lds = has_member && side3 <= PQA_PRE_MEMB;
  const unsigned char* gmemb = has_member ? S.pb->member + (size_t)S.pb->member_class[ia] * side3 : nullptr;
  {
    const int nl = min(S.pb->num_Ls[ia], 128);
    const bool by_candidate = has_member && !S.pb->memb_mask;  // (only the table-less membership tests read these two)
    for (int q = tid; q < 3 * nl; q += PQA_PRE_NT) {
      s_Ls[q / 3][q % 3] = S.pb->Ls[q];
      if (by_candidate) s_imgn[q / 3][q % 3] = S.pb->img_n[q];
    }
    if (memb_lds && by_candidate) for (int q = tid; q < side3; q += PQA_PRE_NT) s_memb[q] = gmemb[q];
    if (tid < PQA_MAXCLS) s_cut[tid] = tid < ncls ? S.pb->cls_cut[ia * PQA_MAXCLS + tid] : 0.0;
    __syncthreads();
  }
  const long p = (long)blockIdx.x * PQA_PRE_NT + tid;
  if (p >= P) return;
  double px, py, pz;
  load_point(pa, p, px, py, pz);
  int dw[3];
  fold_cell(S, px, py, pz, dw);
  if (S.pb->twist && ia == 0) {  // wrap phase exp(i k_t . wrap . lattice) of the point (orbitals.py:203-213)
    double sn, cs;
    sincos(dw[0] * S.pb->ktl[0] + dw[1] * S.pb->ktl[1] + dw[2] * S.pb->ktl[2], &sn, &cs);
    theta[2 * p] = cs; theta[2 * p + 1] = sn;
  }
  PbcCtx c;
  pbc_ctx_base(S, c, ia, px - S.atom_xyz[3 * ia], py - S.atom_xyz[3 * ia + 1], pz - S.atom_xyz[3 * ia + 2], prim_wrap(S, px, py, pz));
  d0[((size_t)ia * 3 + 0) * P + p] = c.x0;
  d0[((size_t)ia * 3 + 1) * P + p] = c.y0;
  d0[((size_t)ia * 3 + 2) * P + p] = c.z0;
  if (S.pb->twist) { d0[((size_t)S.natom * 3 + 2 * ia) * P + p] = c.cf; d0[((size_t)S.natom * 3 + 2 * ia + 1) * P + p] = c.sf; }
  unsigned long long* out = lst + (size_t)ia * NW * P + p;  // word w at out[w * P]
  const int nimg = S.pb->num_Ls[ia];
  const int nw = min(NW, PQA_PRE_NWMAX), cap = min(4 * nw - 1, PQA_PRE_CAP);
  unsigned short* s_rec = s_pre;
  unsigned short* s_ent = s_pre + (size_t)4 * nw * PQA_PRE_NT;
  if (nimg > 128 || ncls <= 0 || ncls > NCUT) { out[0] = (unsigned long long)PQA_IMG_OVF; return; }
  // 1. candidates worth a distance test: the ones near this sub-cell of the folded displacement (create: near_masks) that the
  //    membership rule admits (one look-up of the pre-tabulated mask of this (atom class, fold): member_masks)
  unsigned long long m0 = nimg >= 64 ? ~0ull : (1ull << nimg) - 1ull;
  unsigned long long m1 = nimg <= 64 ? 0ull : (nimg >= 128 ? ~0ull : (1ull << (nimg - 64)) - 1ull);
  if (S.pb->near_mask) {
    const int G = S.pb->near_G;
    const double u0 = c.x0 * S.pb->linv[0] + c.y0 * S.pb->linv[3] + c.z0 * S.pb->linv[6];
    const double u1 = c.x0 * S.pb->linv[1] + c.y0 * S.pb->linv[4] + c.z0 * S.pb->linv[7];
    const double u2 = c.x0 * S.pb->linv[2] + c.y0 * S.pb->linv[5] + c.z0 * S.pb->linv[8];
    const int g0 = min(G - 1, max(0, (int)((u0 + 0.5) * G))), g1 = min(G - 1, max(0, (int)((u1 + 0.5) * G))),
              g2 = min(G - 1, max(0, (int)((u2 + 0.5) * G)));
    const unsigned long long* nm = S.pb->near_mask + 2 * ((((size_t)ia * G + g0) * G + g1) * G + g2);
    m0 &= nm[0]; m1 &= nm[1];
  }
  const double acut = S.pb->atom_cut[ia];
  if (has_member) {
    if (S.pb->memb_mask) {
      const int E = S.pb->memb_E, T = side + 2 * E;
      const int i0 = c.b0 + E, i1 = c.b1 + E, i2 = c.b2 + E;
      if ((unsigned)i0 < (unsigned)T && (unsigned)i1 < (unsigned)T && (unsigned)i2 < (unsigned)T) {
        const unsigned long long* mm = S.pb->memb_mask + 2 * ((((size_t)S.pb->member_class[ia] * T + i0) * T + i1) * T + i2);
        m0 &= mm[0]; m1 &= mm[1];
      } else { m0 = 0ull; m1 = 0ull; }
    } else {  // no table: candidate by candidate (distance first: the rule costs four dependent look-ups)
      unsigned long long k0 = 0ull, k1 = 0ull, b0_ = m0, b1_ = m1;
      while (b0_ | b1_) {
        int j;
        if (b0_) { j = __ffsll((long long)b0_) - 1; b0_ &= b0_ - 1; }
        else { j = 64 + __ffsll((long long)b1_) - 1; b1_ &= b1_ - 1; }
        const double xj = c.x0 - s_Ls[j][0], yj = c.y0 - s_Ls[j][1], zj = c.z0 - s_Ls[j][2];
        if (xj * xj + yj * yj + zj * zj > acut) continue;
        const int n0 = c.b0 + s_imgn[j][0], n1 = c.b1 + s_imgn[j][1], n2 = c.b2 + s_imgn[j][2];
        if ((unsigned)n0 >= (unsigned)side || (unsigned)n1 >= (unsigned)side || (unsigned)n2 >= (unsigned)side) continue;
        const int mi = (n0 * side + n1) * side + n2;
        if (!(memb_lds ? s_memb[mi] : gmemb[mi])) continue;
        if (j < 64) k0 |= 1ull << j; else k1 |= 1ull << (j - 64);
      }
      m0 = k0; m1 = k1;
    }
  }
  // 2. distance test of those.  Every admitted image gets the class of the smallest shell cut-off that contains it; its record
  //    (index, class) goes to the thread's LDS column in arrival order, the class populations are counted in 6-bit fields.
  double cut_r[NCUT];
#pragma unroll
  for (int q = 0; q < NCUT; ++q) cut_r[q] = q < ncls ? s_cut[q] : INFINITY;  // (classes past the last are never exceeded)
  typedef typename PreFields<NCUT>::type fld_t;   // NCUT 6-bit fields: 32 bits for five classes, 64 for ten
  fld_t cnt = 0;
  int n = 0, kmin = -1;
  double rmin = 1e300;
  bool over = false;
#pragma unroll
  for (int half = 0; half < 2; ++half) {
    unsigned long long m = over ? 0ull : (half ? m1 : m0);
    while (m) {
      const int j = 64 * half + __ffsll((long long)m) - 1;
      m &= m - 1;
      const double xj = c.x0 - s_Ls[j][0], yj = c.y0 - s_Ls[j][1], zj = c.z0 - s_Ls[j][2];
      const double r2 = xj * xj + yj * yj + zj * zj;
      int cls = 0;
#pragma unroll
      for (int q = 0; q < NCUT; ++q) cls += r2 > cut_r[q] ? 1 : 0;
      if (r2 > acut || cls >= ncls) continue;  // outside the atom's cut-off, or inside it but outside every shell's
      if (n >= cap) { over = true; m = 0ull; continue; }
      s_rec[n * PQA_PRE_NT + tid] = (unsigned short)(j | (cls << 8));
      cnt += (fld_t)1 << (6 * cls);
      if (r2 < rmin) { rmin = r2; kmin = n; }
      ++n;
    }
  }
  if (over) { out[0] = (unsigned long long)PQA_IMG_OVF; return; }
  // 3. nearest image first, then class by class (index order inside a class): counting scatter of the records into the
  //    thread's second LDS column, 16-bit entries (no read-modify-write of packed words)
  const int nwords = n / 4 + 1;  // n entries + at least one terminator
  if (n > 0) {
    const int cmin = s_rec[kmin * PQA_PRE_NT + tid] >> 8;
    cnt -= (fld_t)1 << (6 * cmin);  // the nearest image leaves its class ...
    fld_t off = 0;                  // ... and takes position 0; off: next free position of every class
    int run = 1;
#pragma unroll
    for (int q = 0; q < NCUT; ++q) { off |= (fld_t)run << (6 * q); run += (int)((cnt >> (6 * q)) & 63); }
    for (int k = 0; k < n; ++k) {
      const int rec = s_rec[k * PQA_PRE_NT + tid];
      const int sh = 6 * (rec >> 8);
      int pos = 0;
      if (k != kmin) { pos = (int)((off >> sh) & 63); off += (fld_t)1 << sh; }
      s_ent[pos * PQA_PRE_NT + tid] = (unsigned short)(rec & 255);
    }
  }
  for (int k = n; k < 4 * nwords; ++k) s_ent[k * PQA_PRE_NT + tid] = (unsigned short)PQA_IMG_END;
  for (int w = 0; w < nwords; ++w) {
    const unsigned e0 = s_ent[(4 * w) * PQA_PRE_NT + tid], e1 = s_ent[(4 * w + 1) * PQA_PRE_NT + tid],
                   e2 = s_ent[(4 * w + 2) * PQA_PRE_NT + tid], e3 = s_ent[(4 * w + 3) * PQA_PRE_NT + tid];
    out[(size_t)w * P] = (unsigned long long)(e0 | (e1 << 16)) | ((unsigned long long)(e2 | (e3 << 16)) << 32);
  }
}

// per (point, atom) data of k_pbc_prepass -> context of the shells of atom ia (k_orb, k_orb_wide)
template <int PBC, class Tab>
__device__ __forceinline__ void pbc_ctx_load(const SysDev& S, const Tab& T, PbcCtx& ctx, int ia, long P, long p, double x, double y,
                                             double z, PrimWrap pw) {
  if (ctx.ia == ia) return;
  ctx.ia = ia;
  ctx.x0 = T.pbc_d0[((size_t)ia * 3 + 0) * P + p];
  ctx.y0 = T.pbc_d0[((size_t)ia * 3 + 1) * P + p];
  ctx.z0 = T.pbc_d0[((size_t)ia * 3 + 2) * P + p];
  if (PBC == 2) {
    ctx.cf = T.pbc_d0[((size_t)S.natom * 3 + 2 * ia) * P + p];
    ctx.sf = T.pbc_d0[((size_t)S.natom * 3 + 2 * ia + 1) * P + p];
  }
  ctx.lp = T.pbc_list + (size_t)ia * T.pbc_nw * P + p;
  ctx.lstride = P;
  ctx.lcap = 4 * T.pbc_nw;
  ctx.ovf = (ctx.lp[0] & 0xFFFFull) == (unsigned long long)PQA_IMG_OVF;
  if (ctx.ovf) pbc_ctx_base(S, ctx, ia, x, y, z, pw);  // membership base for the direct tests (same fold as the pre-pass)
}

// Twisted cells: multiply every orbital row [ncomp][2 nmo] (re block | im block) by the point's wrap phase.
// rows of a two-slot output cleared before a K-split launch accumulates into them.  grid = (P, ceil(row / 256)), block = 256
template <int PQA_UNIT = 0>  // (a template so that only the units that launch it compile it)
static __global__ __launch_bounds__(256) void k_zero_rows(double* __restrict__ out, int row, const unsigned char* __restrict__ sel, long slot_stride) {
  const long p = blockIdx.x;
  const int k = blockIdx.y * 256 + threadIdx.x;
  if (k < row) out[(size_t)(sel[p] ^ 1) * slot_stride + (size_t)p * row + k] = 0.0;
}
// sel / slot_stride: the two-slot output of ChunkTab (nullptr: plain rows)
template <int PQA_UNIT = 0>  // (a template so that only the units that launch it compile it)
static __global__ void k_row_phase(double* __restrict__ out, long P, int ncomp, int nmo2, const double* __restrict__ theta,
                            const unsigned char* __restrict__ sel, long slot_stride) {
  const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
  const int nmo = nmo2 / 2;
  if (idx >= P * ncomp * nmo) return;
  const long p = idx / ((long)ncomp * nmo);
  const int c = (int)((idx / nmo) % ncomp), j = (int)(idx % nmo);
  double* row = out + (sel ? (size_t)(sel[p] ^ 1) * slot_stride : (size_t)0) + ((size_t)p * ncomp + c) * nmo2;
  const double cs = theta[2 * p], sn = theta[2 * p + 1], re = row[j], im = row[nmo + j];
  row[j] = re * cs - im * sn;
  row[nmo + j] = re * sn + im * cs;
}

// ---------------------------------------------------------------- AO only (test / A-B entry)
// out (NCOMP, P, nao); one thread per point.
// PLMAX: largest l of the lattice-summed shells (3: the instantiation every periodic handle without g / h shells uses)
template <int NCOMP, int PLMAX = 3>
static __global__ void k_ao(SysDev S, PointAddr pa, long P, double* __restrict__ out) {
  const long p = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (p >= P) return;
  double px, py, pz;
  load_point(pa, p, px, py, pz);
  if (S.nL > 0) fold_cell(S, px, py, pz);  // periodic orbitals are tabulated for points inside the cell
  PbcCtx ctx;
  const PrimWrap pw = S.nL > 0 ? prim_wrap(S, px, py, pz) : PrimWrap{0, 0, 0};  // S.pb is null for open systems
  for (int sh = 0; sh < S.nshell; ++sh) {
    const int ia = S.shell_atom[sh], p0 = S.shell_prim_off[sh], ao0 = S.shell_ao_off[sh];
    const double x = px - S.atom_xyz[3 * ia], y = py - S.atom_xyz[3 * ia + 1], z = pz - S.atom_xyz[3 * ia + 2];
    bool accum = false;  // periodic: the lattice sum accumulates in place (shell_eval_pbc)
    auto store = [&](int m, double v, double gx, double gy, double gz, double lp) {
      double* o = out + p * S.nao + ao0 + m;
      const long cs = P * (long)S.nao;
      if (accum) {
        o[0] += v;
        if (NCOMP > 1) { o[cs] += gx; o[2 * cs] += gy; o[3 * cs] += gz; }
        if (NCOMP == 5) o[4 * cs] += lp;
      } else {
        o[0] = v;
        if (NCOMP > 1) { o[cs] = gx; o[2 * cs] = gy; o[3 * cs] = gz; }
        if (NCOMP == 5) o[4 * cs] = lp;
      }
    };
    if (S.nL > 0) {
      pbc_ctx_update(S, ctx, ia, x, y, z, pw);
      shell_eval_pbc<NCOMP, PLMAX>(S, ctx, sh, S.shell_l[sh], S.prim_exp + p0, S.prim_coef + p0, S.shell_prim_off[sh + 1] - p0, store, accum);
    } else
      shell_eval<NCOMP>(S.shell_l[sh], x, y, z, S.prim_exp + p0, S.prim_coef + p0, S.shell_prim_off[sh + 1] - p0, store);
  }
}

// AO values of a TWISTED cell at arbitrary (unfolded) points: sum_L e^{i k_t.L} phi(r - R - L) times the wrap phase of the fold
// (orbitals.py:203-213).  out: real plane [P][nao] followed by the imaginary plane.  One thread per point, direct image tests.
// Used by the parameter gradient of complex determinants (pqa_slater_pgradient).
template <int PQA_UNIT = 0>  // (a template so that only the units that launch it compile it)
static __global__ void k_ao_tw(SysDev S, const double* __restrict__ pts, long P, double* __restrict__ out) {
  const long p = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (p >= P) return;
  double px = pts[3 * p], py = pts[3 * p + 1], pz = pts[3 * p + 2];
  int dw[3];
  fold_cell(S, px, py, pz, dw);
  double wsn, wcs;
  sincos(dw[0] * S.pb->ktl[0] + dw[1] * S.pb->ktl[1] + dw[2] * S.pb->ktl[2], &wsn, &wcs);
  PbcCtx ctx;
  const PrimWrap pw = prim_wrap(S, px, py, pz);
  double* ore = out + p * S.nao;
  double* oim = out + (P + p) * (long)S.nao;
  for (int sh = 0; sh < S.nshell; ++sh) {
    const int ia = S.shell_atom[sh], p0 = S.shell_prim_off[sh], ao0 = S.shell_ao_off[sh];
    const double x = px - S.atom_xyz[3 * ia], y = py - S.atom_xyz[3 * ia + 1], z = pz - S.atom_xyz[3 * ia + 2];
    bool accum = false;
    auto st_re = [&](int m, double v, double, double, double, double) { if (accum) ore[ao0 + m] += v; else ore[ao0 + m] = v; };
    auto st_im = [&](int m, double v, double, double, double, double) { if (accum) oim[ao0 + m] += v; else oim[ao0 + m] = v; };
    pbc_ctx_update(S, ctx, ia, x, y, z, pw);
    shell_eval_pbc<1, true>(S, ctx, sh, S.shell_l[sh], S.prim_exp + p0, S.prim_coef + p0, S.shell_prim_off[sh + 1] - p0, st_re, st_im, accum,
                            [&](int j, double& lx, double& ly, double& lz, double& cj, double& sj) {
                              lx = S.pb->Ls[3 * j]; ly = S.pb->Ls[3 * j + 1]; lz = S.pb->Ls[3 * j + 2];
                              cj = S.pb->img_phase[2 * j]; sj = S.pb->img_phase[2 * j + 1];
                            });
    for (int m = 0; m < 2 * S.shell_l[sh] + 1; ++m) {  // wrap phase of the point
      const double re = ore[ao0 + m], im = oim[ao0 + m];
      ore[ao0 + m] = re * wcs - im * wsn;
      oim[ao0 + m] = re * wsn + im * wcs;
    }
  }
}

// Contraction of AO planes [ncomp][P][nao] (k_ao) with C [nao][nmo] into the orbital kernels' row layout out[p][ncomp][nmo]
// (two-slot output like ChunkTab::out_sel when sel != nullptr).  The general path of periodic cells with g / h shells — the MFMA
// kernels' lattice-sum phase is built for l <= 3 (registers); correctness first, one thread per output value.
template <int PQA_UNIT = 0>
static __global__ void k_mo_rows(const double* __restrict__ ao, const double* __restrict__ C, long P, int ncomp, int nao, int nmo,
                                 double* __restrict__ out, const unsigned char* __restrict__ sel, long slot_stride) {
  const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= P * ncomp * nmo) return;
  const long p = idx / ((long)ncomp * nmo);
  const int c = (int)((idx / nmo) % ncomp), j = (int)(idx % nmo);
  const double* a = ao + ((size_t)c * P + p) * nao;
  double s = 0.0;
  for (int k = 0; k < nao; ++k) s += a[k] * C[(size_t)k * nmo + j];
  out[(sel ? (size_t)(sel[p] ^ 1) * slot_stride : (size_t)0) + ((size_t)p * ncomp + c) * nmo + j] = s;
}

// plain contraction out[c][p][j] = sum_a ao[c][p][a] C[a][j]  (A/B check of the MFMA kernel only)
template <int PQA_UNIT = 0>  // (a template so that only the units that launch it compile it)
static __global__ void k_mo_valu(const double* __restrict__ ao, const double* __restrict__ C, long rows, int nao, int nmo,
                          double* __restrict__ out) {
  const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= rows * nmo) return;
  const long r = idx / nmo;
  const int j = idx % nmo;
  double s = 0.0;
  for (int a = 0; a < nao; ++a) s += ao[r * nao + a] * C[(long)a * nmo + j];
  out[idx] = s;
}

// ---------------------------------------------------------------- fused AO -> MO, MFMA
// The shells are packed into chunks of <= KC functions and, inside a chunk, into the G = 256/TP lane groups of a
// block so that EVERY chunk gives every group the same amount of work (a barrier ends each chunk, so an unbalanced
// chunk idles three waves).  The contraction does not care in which order the AOs arrive: a chunk's tile rows are
// its shells in packing order (shell_kb), and the coefficient matrices are stored in that permuted row order.
struct ChunkTab {
  int nchunk;
  const int* chunk_nk;    // AOs in chunk
  const int* shell_kb;    // [nshell] first tile row of the shell inside its chunk
  const int* chunk_row0;  // first row in the zero-padded coefficient matrices
  const int* cw_off[3];   // [nchunk*G+1] for G = 4 (TP=64), G = 8 (TP=32) and G = 16 (TP=16)
  const int* cw_shell[3]; // shells for (chunk, group)
  const double* cpad[2];  // per spin [rows_pad][ldc[s]], rows padded to x4 per chunk, cols to x16
  int ldc[2];
  int col0;               // first orbital column of THIS launch (k_orb): handles with more than 64 orbitals of a spin contract them in
                          // windows of 64 = four 16-column MFMA tiles, one launch per window (0 everywhere else)
  // periodic launches only: per (atom, point) folded displacement [natom][3][P] and sorted image list [natom][pbc_nw][P],
  // written by k_pbc_prepass for the points of THIS launch
  const double* pbc_d0;
  const unsigned long long* pbc_list;
  int pbc_nw;
  // two-slot output (the lane-per-walker sweep's row cache, pqa_lw.hpp): point p's rows go to out + (out_sel[p] ^ 1) *
  // out_slot_stride + p * NCOMP * nmo, i.e. into the slot the walker is not using.  nullptr: plain out[p][c][j].
  const unsigned char* out_sel;
  long out_slot_stride;
};
__device__ __forceinline__ double* orb_out(const ChunkTab& T, double* out, long p) {
  return T.out_sel ? out + (size_t)(T.out_sel[p] ^ 1) * T.out_slot_stride : out;
}

#define PQA_LS_MAX 128     // candidate lattice vectors the periodic kernels keep in LDS
#define PQA_WS_MAXSH 160   // shells / primitives that fit the LDS-resident basis tables
#define PQA_WS_MAXP 640

// out[p][c][j], p < P, c < NCOMP, j < nmo.  Block = 256 threads (4 waves), TP = 64, 32 or 16 points.
//  phase 1 (VALU/exp bound): thread = (point, lane group); each group evaluates its share of the chunk's
//          shells and writes the XOR-swizzled LDS tile [comp][k][point ^ ((k&1)<<4)]
//  phase 2 (MFMA): TP=64: wave wv owns the 16-point tile wv and all NT orbital tiles;
//                  TP=32: wave wv owns point tile wv&1 and orbital tiles (wv>>1), (wv>>1)+2, ...
//                  TP=16: one point tile, wave wv owns orbital tiles wv, wv+4, ... (small launches: more blocks, 16 lane groups)
//          D[point][orb] += A[point][k] B[k][orb] with v_mfma_f64_16x16x4_f64; B straight from L2.
// PBC: 0 open system, 1 periodic (real lattice sums), 2 periodic with a twist (complex lattice sums: real and imaginary tile rows per shell)
template <int NCOMP, int NT, int KC, int TP, bool LDSTAB, int PBC = 0>
static __global__ __launch_bounds__(256) void k_orb(SysDev S, ChunkTab T, int spin, PointAddr pa, long P,
                                             double* __restrict__ out) {
  P = point_count(pa, P);
  if ((long)blockIdx.x * TP >= P) return;
  constexpr int G = 256 / TP;                         // lane groups in phase 1
  constexpr int NU = (TP == 64) ? NT : ((TP == 32) ? (NT + 1) / 2 : (NT + 3) / 4);  // orbital tiles per wave in phase 2
  constexpr int KS = KC / 4;
  __shared__ double tile[NCOMP][KC][TP];
  // LDSTAB: basis tables staged once per block so phase 1 never waits on chains of dependent scalar loads
  __shared__ double sh_xyz[LDSTAB ? PQA_WS_MAXSH : 1][3];
  __shared__ int sh_meta[LDSTAB ? PQA_WS_MAXSH : 1][7];  // l, nprim, first primitive, tile row in chunk, atom, radial table offset (-1: none) and intervals
  __shared__ double pr_exp[LDSTAB ? PQA_WS_MAXP : 1], pr_coef[LDSTAB ? PQA_WS_MAXP : 1];
  __shared__ double sh_Ls[PBC ? PQA_LS_MAX : 1][3], sh_ph[PBC == 2 ? PQA_LS_MAX : 1][2];  // lattice vectors (phases) of the candidate images
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  const bool ls_lds = PBC && S.nL <= PQA_LS_MAX;
  if (PBC && ls_lds) {
    for (int q = tid; q < 3 * S.nL; q += 256) sh_Ls[q / 3][q % 3] = S.pb->Ls[q];
    if (PBC == 2) for (int q = tid; q < 2 * S.nL; q += 256) sh_ph[q / 2][q % 2] = S.pb->img_phase[q];
    if (!LDSTAB) __syncthreads();
  }
  if (LDSTAB) {
    for (int sh = tid; sh < S.nshell; sh += 256) {
      const int ia = S.shell_atom[sh];
      sh_xyz[sh][0] = S.atom_xyz[3 * ia]; sh_xyz[sh][1] = S.atom_xyz[3 * ia + 1]; sh_xyz[sh][2] = S.atom_xyz[3 * ia + 2];
      sh_meta[sh][0] = S.shell_l[sh];
      sh_meta[sh][1] = S.shell_prim_off[sh + 1] - S.shell_prim_off[sh];
      sh_meta[sh][2] = S.shell_prim_off[sh];
      sh_meta[sh][3] = T.shell_kb[sh];
      sh_meta[sh][4] = ia;
      sh_meta[sh][5] = (PBC || NCOMP > 1) ? -1 : S.shell_rt[2 * sh]; sh_meta[sh][6] = (PBC || NCOMP > 1) ? 0 : S.shell_rt[2 * sh + 1];
    }
    for (int p = tid; p < S.nprim; p += 256) { pr_exp[p] = S.prim_exp[p]; pr_coef[p] = S.prim_coef[p]; }
    __syncthreads();
  }
  // lane group: with 64-point tiles a wave IS one group, so the group index — and through it the shell list, every shell's
  // table entries and the primitive loop's trip count — is wave-uniform: tell the compiler (scalar loads and a scalar loop
  // instead of per-lane loads of the same value and an exec-masked loop; the per-shell look-ups were two dependent vector
  // memory round trips each, ~22 shells per group and tile)
  const int pl = tid & (TP - 1), grp = (TP == 64) ? __builtin_amdgcn_readfirstlane(tid / TP) : tid / TP;
  const long p0 = (long)blockIdx.x * TP;
  const long pmine = (p0 + pl < P) ? p0 + pl : P - 1;
  double px, py, pz;
  load_point(pa, pmine, px, py, pz);
  if (PBC) fold_cell(S, px, py, pz);  // callers may hand over quadrature / proposal points outside the cell
  PbcCtx ctx;
  const PrimWrap pw = PBC ? prim_wrap(S, px, py, pz) : PrimWrap{0, 0, 0};

  d4 acc[NU][NCOMP];
#pragma unroll
  for (int u = 0; u < NU; ++u)
#pragma unroll
    for (int c = 0; c < NCOMP; ++c) acc[u][c] = (d4){0.0, 0.0, 0.0, 0.0};

  const double* __restrict__ C = T.cpad[spin] + T.col0;
  const int ldc = T.ldc[spin];
  const int i16 = lane & 15, kq = lane >> 4;
  const int ptile = (TP == 64) ? wv : ((TP == 32) ? (wv & 1) : 0);
  const int u0 = (TP == 64) ? 0 : ((TP == 32) ? (wv >> 1) : wv), ustep = (TP == 64) ? 1 : ((TP == 32) ? 2 : 4);
  constexpr int TI = (TP == 64) ? 0 : ((TP == 32) ? 1 : 2);
  const int* __restrict__ cw_off = T.cw_off[TI];
  const int* __restrict__ cw_shell = T.cw_shell[TI];
  auto ls_from_lds = [&](int j, double& lx, double& ly, double& lz, double& cj, double& sj) {
    lx = sh_Ls[j][0]; ly = sh_Ls[j][1]; lz = sh_Ls[j][2];
    if (PBC == 2) { cj = sh_ph[PBC == 2 ? j : 0][0]; sj = sh_ph[PBC == 2 ? j : 0][1]; }
  };
  auto ls_from_global = [&](int j, double& lx, double& ly, double& lz, double& cj, double& sj) {
    lx = S.pb->Ls[3 * j]; ly = S.pb->Ls[3 * j + 1]; lz = S.pb->Ls[3 * j + 2];
    if (PBC == 2) { cj = S.pb->img_phase[2 * j]; sj = S.pb->img_phase[2 * j + 1]; }
  };

  // gridDim.y > 1: the chunks (the K dimension) are split over that many blocks per point tile, each adding its partial
  // sums to a zeroed output with hardware fp64 atomics — for small launches, where a block's serial chain over the chunks,
  // not throughput, sets the time.  Two partial sums commute, so the result is deterministic for a split of 2.
  const int nsplit = gridDim.y, ch_lo = (int)((long)blockIdx.y * T.nchunk / nsplit), ch_hi = (int)((long)(blockIdx.y + 1) * T.nchunk / nsplit);
  for (int ch = ch_lo; ch < ch_hi; ++ch) {
    const int nk = T.chunk_nk[ch], row0 = T.chunk_row0[ch];
    const int nk4 = (nk + 3) & ~3;
    // B operand of this chunk: issue the L2 loads now, consume them after phase 1
    double bq[KS][NU];
    {
      const double* crow = C + (long)(row0 + kq) * ldc + i16;
#pragma unroll
      for (int ks = 0; ks < KS; ++ks)
#pragma unroll
        for (int u = 0; u < NU; ++u) {
          const int ut = u0 + u * ustep;
          bq[ks][u] = (ut < NT) ? crow[(long)ks * 4 * ldc + 16 * ut] : 0.0;
        }
    }
#ifdef PQA_ABL_NOP1
    const int s_end = 0;
#else
    const int s_end = cw_off[ch * G + grp + 1];
#endif
    for (int si = cw_off[ch * G + grp]; si < s_end; ++si) {
      const int sh = cw_shell[si];
      int l_, np_, q0, kb, ia_ = 0, rt_ = -1, rn_ = 0;
      double x, y, z;
      const double *pe, *pc;
      if (LDSTAB) {
        l_ = sh_meta[sh][0]; np_ = sh_meta[sh][1]; q0 = sh_meta[sh][2]; kb = sh_meta[sh][3]; ia_ = sh_meta[sh][4];
        if (!PBC && NCOMP == 1) { rt_ = sh_meta[sh][5]; rn_ = sh_meta[sh][6]; }
        x = px - sh_xyz[sh][0]; y = py - sh_xyz[sh][1]; z = pz - sh_xyz[sh][2];
        pe = pr_exp + q0; pc = pr_coef + q0;
      } else {
        const int ia = S.shell_atom[sh];
        ia_ = ia;
        q0 = S.shell_prim_off[sh]; kb = T.shell_kb[sh]; l_ = S.shell_l[sh]; np_ = S.shell_prim_off[sh + 1] - q0;
        x = px - S.atom_xyz[3 * ia]; y = py - S.atom_xyz[3 * ia + 1]; z = pz - S.atom_xyz[3 * ia + 2];
        pe = S.prim_exp + q0; pc = S.prim_coef + q0;
        if (!PBC && NCOMP == 1) { rt_ = S.shell_rt[2 * sh]; rn_ = S.shell_rt[2 * sh + 1]; }
      }
      bool accum = false;  // periodic: the lattice sum accumulates in the tile (shell_eval_pbc)
      auto to_tile = [&](int m, double v, double gx, double gy, double gz, double lp) {
        const int k = kb + m;
        const int col = (TP >= 32) ? (pl ^ ((k & 1) << 4)) : pl;
        if (PBC && accum) {  // (LDS add without return: one LDS operation per value instead of a read and a write; only this lane touches the entry)
          unsafeAtomicAdd(&tile[0][k][col], v);
          if (NCOMP > 1) { unsafeAtomicAdd(&tile[1 % NCOMP][k][col], gx); unsafeAtomicAdd(&tile[2 % NCOMP][k][col], gy); unsafeAtomicAdd(&tile[3 % NCOMP][k][col], gz); }
          if (NCOMP == 5) unsafeAtomicAdd(&tile[4 % NCOMP][k][col], lp);
        } else {
          tile[0][k][col] = v;
          if (NCOMP > 1) { tile[1 % NCOMP][k][col] = gx; tile[2 % NCOMP][k][col] = gy; tile[3 % NCOMP][k][col] = gz; }
          if (NCOMP == 5) tile[4 % NCOMP][k][col] = lp;
        }
      };
      if (PBC) {
        pbc_ctx_load<PBC>(S, T, ctx, ia_, P, pmine, x, y, z, pw);
        if (PBC == 2) {  // twisted: the shell's imaginary rows follow its real rows in the tile
          const int kbi = kb + 2 * l_ + 1;
          auto to_tile_im = [&](int m, double v, double gx, double gy, double gz, double lp) {
            const int k = kbi + m;
            const int col = (TP >= 32) ? (pl ^ ((k & 1) << 4)) : pl;
            if (accum) {
              unsafeAtomicAdd(&tile[0][k][col], v);
              if (NCOMP > 1) { unsafeAtomicAdd(&tile[1 % NCOMP][k][col], gx); unsafeAtomicAdd(&tile[2 % NCOMP][k][col], gy); unsafeAtomicAdd(&tile[3 % NCOMP][k][col], gz); }
              if (NCOMP == 5) unsafeAtomicAdd(&tile[4 % NCOMP][k][col], lp);
            } else {
              tile[0][k][col] = v;
              if (NCOMP > 1) { tile[1 % NCOMP][k][col] = gx; tile[2 % NCOMP][k][col] = gy; tile[3 % NCOMP][k][col] = gz; }
              if (NCOMP == 5) tile[4 % NCOMP][k][col] = lp;
            }
          };
          if (ls_lds) shell_eval_pbc<NCOMP, true>(S, ctx, sh, l_, pe, pc, np_, to_tile, to_tile_im, accum, ls_from_lds);
          else shell_eval_pbc<NCOMP, true>(S, ctx, sh, l_, pe, pc, np_, to_tile, to_tile_im, accum, ls_from_global);
        } else if (ls_lds) shell_eval_pbc<NCOMP, false>(S, ctx, sh, l_, pe, pc, np_, to_tile, to_tile, accum, ls_from_lds);
        else shell_eval_pbc<NCOMP, false>(S, ctx, sh, l_, pe, pc, np_, to_tile, to_tile, accum, ls_from_global);
      } else if (NCOMP == 1 && rt_ >= 0) shell_eval_tab(l_, x, y, z, S.rtab + rt_, rn_, to_tile);  // contracted shell, values: the radial sum from its table
      else shell_eval<NCOMP>(l_, x, y, z, pe, pc, np_, to_tile);
    }
    // zero the rows behind the chunk's last function up to KC: every chunk then takes all KS k-steps, without a branch.  (Guarded by
    // `ks * 4 < nk4` every k-step was a basic block of its own and the accumulators changed register class at each of them: 16 v_accvgpr_write,
    // the MFMAs, s_nop 15, 16 v_accvgpr_read per step — half the vector instructions of the value-only kernel.  The rows' B operands are the next
    // chunk's coefficients or the zero rows behind the table: finite, times zero.)
    // (The periodic value-only kernels keep the guard — with it removed k_orb<1, 2, 16, 64, true, 1> went from 399 to 480 us per launch of C5's
    // T-move candidates — and so do five components: their 80 accumulator registers live in AGPRs throughout, and one basic block of 40 MFMAs lets the scheduler
    // hoist all their LDS operands: 204 -> 260 registers.)
    for (int idx = tid; idx < (((NCOMP == 1 && PBC == 0) ? KC : nk4) - nk) * NCOMP * TP; idx += 256) {
      const int rc = idx / TP;
      tile[rc % NCOMP][nk + rc / NCOMP][idx & (TP - 1)] = 0.0;
    }
    __syncthreads();
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
      if ((NCOMP == 1 && PBC == 0) || ks * 4 < nk4) {
        const int k = ks * 4 + kq;
        const int col = (TP >= 32) ? ((16 * ptile + i16) ^ ((k & 1) << 4)) : i16;
#pragma unroll
        for (int c = 0; c < NCOMP; ++c) {
          const double a = tile[c][k][col];
#pragma unroll
#ifdef PQA_ABL_NOMFMA
          for (int u = 0; u < NU; ++u) acc[u][c][0] += a * bq[ks][u];
#else
          for (int u = 0; u < NU; ++u) acc[u][c] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, bq[ks][u], acc[u][c], 0, 0, 0);
#endif
        }
      }
    }
    __syncthreads();
  }
  // epilogue: lane holds D[row = (lane>>4) + 4r][col = lane & 15]
  const int nmo = S.nmo[spin];
  double* orow[4];  // the lane's four output rows (one selector look-up each, before the store loops: inside them every store
                    // waited for its own dependent byte load)
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const long pp = p0 + 16 * ptile + kq + 4 * r;
    orow[r] = (pp < P) ? orb_out(T, out, pp) + pp * NCOMP * nmo : nullptr;
  }
#pragma unroll
  for (int u = 0; u < NU; ++u) {
    const int ut = u0 + u * ustep;
    const int j = T.col0 + 16 * ut + i16;
    if (ut >= NT || j >= nmo) continue;
#pragma unroll
    for (int c = 0; c < NCOMP; ++c)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        if (orow[r]) {
          double* o = orow[r] + c * nmo + j;
          if (nsplit > 1) unsafeAtomicAdd(o, acc[u][c][r]);
          else *o = acc[u][c][r];
        }
      }
  }
}

// ---------------------------------------------------------------- wave-specialised variant
// Same math and tables as k_orb<.., TP=64>, different schedule: a block has 8 waves; waves 0-3 are
// PRODUCERS (phase-1 work: exp-bound AO evaluation of chunk c into LDS buffer c&1), waves 4-7 are
// CONSUMERS (phase-2 work: MFMA contraction of chunk c-1 from buffer (c-1)&1, B operand prefetched
// one chunk ahead from L2).  One barrier per chunk; a producer and a consumer wave share each SIMD,
// so the VALU/transcendental pipe and the matrix pipe run concurrently inside ONE block — which is
// what a launch of only W/64 = 256 blocks (one per CU) needs.
template <int NCOMP, int NT, int KC>
static __global__ __launch_bounds__(512) void k_orb_ws(SysDev S, ChunkTab T, int spin, PointAddr pa, long P,
                                                double* __restrict__ out) {
  P = point_count(pa, P);
  if ((long)blockIdx.x * 64 >= P) return;
  constexpr int KS = KC / 4;
  __shared__ double tile[2][NCOMP][KC][64];
  // basis tables staged once per block: the producer loop then never waits on dependent global/scalar
  // loads (shell -> atom -> coordinates -> primitives), which is what bounded it before
  __shared__ double sh_xyz[PQA_WS_MAXSH][3];
  __shared__ int sh_meta[PQA_WS_MAXSH][NCOMP == 1 ? 6 : 4];  // l, nprim, first primitive, first AO; values only: radial table offset (-1: none), intervals
  __shared__ double pr_exp[PQA_WS_MAXP], pr_coef[PQA_WS_MAXP];
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  for (int sh = tid; sh < S.nshell; sh += 512) {
    const int ia = S.shell_atom[sh];
    sh_xyz[sh][0] = S.atom_xyz[3 * ia]; sh_xyz[sh][1] = S.atom_xyz[3 * ia + 1]; sh_xyz[sh][2] = S.atom_xyz[3 * ia + 2];
    sh_meta[sh][0] = S.shell_l[sh];
    sh_meta[sh][1] = S.shell_prim_off[sh + 1] - S.shell_prim_off[sh];
    sh_meta[sh][2] = S.shell_prim_off[sh];
    sh_meta[sh][3] = T.shell_kb[sh];
    if (NCOMP == 1) { sh_meta[sh][4 % (NCOMP == 1 ? 6 : 4)] = S.shell_rt[2 * sh]; sh_meta[sh][5 % (NCOMP == 1 ? 6 : 4)] = S.shell_rt[2 * sh + 1]; }
  }
  for (int p = tid; p < S.nprim; p += 512) { pr_exp[p] = S.prim_exp[p]; pr_coef[p] = S.prim_coef[p]; }
  __syncthreads();
  const bool producer = wv < 4;
  const int grp = __builtin_amdgcn_readfirstlane(wv & 3);  // producer: shell group; consumer: 16-point tile (wave-uniform)
  const long p0 = (long)blockIdx.x * 64;
  const long pmine = (p0 + lane < P) ? p0 + lane : P - 1;
  double px = 0.0, py = 0.0, pz = 0.0;
  if (producer) load_point(pa, pmine, px, py, pz);

  d4 acc[NT][NCOMP];
#pragma unroll
  for (int u = 0; u < NT; ++u)
#pragma unroll
    for (int c = 0; c < NCOMP; ++c) acc[u][c] = (d4){0.0, 0.0, 0.0, 0.0};

  const double* __restrict__ C = T.cpad[spin];
  const int ldc = T.ldc[spin];
  const int i16 = lane & 15, kq = lane >> 4;
  const int* __restrict__ cw_off = T.cw_off[0];
  const int* __restrict__ cw_shell = T.cw_shell[0];
  double bq[KS][NT];  // B operand of the chunk the consumer will contract NEXT iteration
#pragma unroll
  for (int ks = 0; ks < KS; ++ks)
#pragma unroll
    for (int u = 0; u < NT; ++u) bq[ks][u] = 0.0;

  for (int ch = 0; ch <= T.nchunk; ++ch) {
    if (producer) {
      if (ch < T.nchunk) {
        double (*tb)[KC][64] = tile[ch & 1];
        const int nk = T.chunk_nk[ch];
        const int nk4 = (nk + 3) & ~3;
        const int s_end = cw_off[ch * 4 + grp + 1];
        for (int si = cw_off[ch * 4 + grp]; si < s_end; ++si) {
          const int sh = cw_shell[si];
          const int q0 = sh_meta[sh][2], kb = sh_meta[sh][3];
          const double x = px - sh_xyz[sh][0], y = py - sh_xyz[sh][1], z = pz - sh_xyz[sh][2];
          auto to_tile = [&](int m, double v, double gx, double gy, double gz, double lp) {
            const int k = kb + m;
            const int col = lane ^ ((k & 1) << 4);
            tb[0][k][col] = v;
            if (NCOMP > 1) { tb[1 % NCOMP][k][col] = gx; tb[2 % NCOMP][k][col] = gy; tb[3 % NCOMP][k][col] = gz; }
            if (NCOMP == 5) tb[4 % NCOMP][k][col] = lp;
          };
          const int rt_ = NCOMP == 1 ? sh_meta[sh][4 % (NCOMP == 1 ? 6 : 4)] : -1;
          if (NCOMP == 1 && rt_ >= 0) shell_eval_tab(sh_meta[sh][0], x, y, z, S.rtab + rt_, sh_meta[sh][5 % (NCOMP == 1 ? 6 : 4)], to_tile);  // radial_tab
          else shell_eval<NCOMP>(sh_meta[sh][0], x, y, z, pr_exp + q0, pr_coef + q0, sh_meta[sh][1], to_tile);
        }
        for (int idx = tid; idx < ((NCOMP == 1 ? KC : nk4) - nk) * NCOMP * 64; idx += 256) {  // zero the K padding rows; values only: up to KC (k_orb: branch-free k-steps); 256 producer threads
          const int rc = idx >> 6;
          tb[rc % NCOMP][nk + rc / NCOMP][idx & 63] = 0.0;
        }
      }
    } else {
      // contract chunk ch-1 with the B values fetched during the previous iteration ...
      double bcur[KS][NT];
#pragma unroll
      for (int ks = 0; ks < KS; ++ks)
#pragma unroll
        for (int u = 0; u < NT; ++u) bcur[ks][u] = bq[ks][u];
      // ... and start fetching B for chunk ch (rows beyond its nk4 belong to the next chunk or the tail pad: unused)
      if (ch < T.nchunk) {
        const double* crow = C + (long)(T.chunk_row0[ch] + kq) * ldc + i16;
#pragma unroll
        for (int ks = 0; ks < KS; ++ks)
#pragma unroll
          for (int u = 0; u < NT; ++u) bq[ks][u] = crow[(long)ks * 4 * ldc + 16 * u];
      }
      if (ch > 0) {
        double (*tb)[KC][64] = tile[(ch - 1) & 1];
        const int nk4 = (T.chunk_nk[ch - 1] + 3) & ~3;
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
          if (NCOMP == 1 || ks * 4 < nk4) {
            const int k = ks * 4 + kq;
            const int col = (16 * grp + i16) ^ ((k & 1) << 4);
#pragma unroll
            for (int c = 0; c < NCOMP; ++c) {
              const double a = tb[c][k][col];
#pragma unroll
              for (int u = 0; u < NT; ++u) acc[u][c] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, bcur[ks][u], acc[u][c], 0, 0, 0);
            }
          }
        }
      }
    }
    __syncthreads();
  }
  if (producer) return;
  const int nmo = S.nmo[spin];
  double* orow[4];
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const long pp = p0 + 16 * grp + kq + 4 * r;
    orow[r] = (pp < P) ? orb_out(T, out, pp) + pp * NCOMP * nmo : nullptr;
  }
#pragma unroll
  for (int u = 0; u < NT; ++u) {
    const int j = 16 * u + i16;
    if (j >= nmo) continue;
#pragma unroll
    for (int c = 0; c < NCOMP; ++c)
#pragma unroll
      for (int r = 0; r < 4; ++r)
        if (orow[r]) orow[r][c * nmo + j] = acc[u][c][r];
  }
}

// ---------------------------------------------------------------- whole-K variant for SMALL launches
// A sweep launches the orbital kernel on one point per walker; at the BASELINE walker counts of the periodic configurations
// (4096 - 8192 per GPU) that is a few hundred 16-point tiles, and k_orb's time there is the LATENCY of one block's chain —
// chunks x (phase 1, barrier, MFMA, barrier), each phase-1 thread walking ~45 (shell, image) lattice sums — not throughput.
// This variant shortens the chain 4x: a block of 1024 threads owns ONE 16-point tile, 64 lane groups split ALL shells of the
// basis between them (a thread evaluates 1-2 shells, not 5-7), the AO tile of the whole basis [NCOMP][rows][16] sits in LDS
// (up to ~150 KB: one block per CU), and after a single barrier the waves contract (component, orbital tile) pairs over the
// full K with v_mfma_f64_16x16x4_f64.  Same shell routines, same coefficient layout (the chunk table's padded row order), so
// rows are identical to k_orb's up to the MFMA accumulation order (one K loop instead of per-chunk partial sums).
#ifdef PQA_WIDE_CLK  // timing build only (tools/scratch/wide_clk.py): 100 MHz stamps of the phases of the first blocks
static __device__ unsigned long long pqa_wide_clk[1024 * 8];
#define PQA_CLK(k) do { if (blockIdx.x < 1024 && threadIdx.x == 0) pqa_wide_clk[blockIdx.x * 8 + (k)] = wall_clock64(); } while (0)
#else
#define PQA_CLK(k) do { } while (0)
#endif
struct WideTab {
  const int* off;     // [65] shells of lane group g: shell[off[g] .. off[g+1]); 64 groups (1024 threads) or 32 (512 threads)
  const int* shell;
  const int* row;     // [nshell] (twisted: [2 nshell], imaginary rows second) first padded tile row of the shell
  int rows_pad;       // K: multiple of 4
};
// nls: doubles of the periodic kernels' lattice-vector / phase table (5 per candidate image, 0 for open systems)
__host__ __device__ inline size_t wide_lds_bytes(int ncomp, int rows_pad, int nshell, int nprim, int nls = 0) {
  return ((size_t)ncomp * rows_pad * 16 + 3 * (size_t)nshell + 2 * (size_t)nprim + (size_t)nls) * sizeof(double) + (size_t)5 * nshell * sizeof(int);
}

// NTH threads: 1024 for open systems (94 VGPRs); periodic lattice sums need > 128 registers (at 1024 threads they spilled
// 384 B, twisted 1024 B per lane, and lost to k_orb), so periodic launches take 512 threads = 32 lane groups.
template <int NCOMP, int NT, int PBC, int NTH>
static __global__ __launch_bounds__(NTH) void k_orb_wide(SysDev S, ChunkTab T, WideTab Wt, int spin, PointAddr pa, long P, double* __restrict__ out) {
  extern __shared__ double wl[];
  PQA_CLK(0);
  const int K = Wt.rows_pad;
  double* tile = wl;                                  // [NCOMP][K][16]
  double* sh_xyz = tile + (size_t)NCOMP * K * 16;     // [nshell][3]
  double* pr_exp = sh_xyz + 3 * (size_t)S.nshell;
  double* pr_coef = pr_exp + S.nprim;
  const bool ls_lds = PBC && S.nL <= PQA_LS_MAX;
  double* w_Ls = pr_coef + S.nprim;                   // periodic: [nL][3] lattice vectors, [nL][2] phases
  double* w_ph = w_Ls + (ls_lds ? 3 * (size_t)S.nL : 0);
  int* sh_meta = (int*)(w_ph + (ls_lds ? 2 * (size_t)S.nL : 0));  // [nshell][5]: l, nprim, first primitive, tile row, atom
  if (PBC && ls_lds) {
    for (int q = threadIdx.x; q < 3 * S.nL; q += NTH) w_Ls[q] = S.pb->Ls[q];
    if (PBC == 2) for (int q = threadIdx.x; q < 2 * S.nL; q += NTH) w_ph[q] = S.pb->img_phase[q];
  }
  auto ls_get = [&](int j, double& lx, double& ly, double& lz, double& cj, double& sj) {
    if (ls_lds) {
      lx = w_Ls[3 * j]; ly = w_Ls[3 * j + 1]; lz = w_Ls[3 * j + 2];
      if (PBC == 2) { cj = w_ph[2 * j]; sj = w_ph[2 * j + 1]; }
    } else {
      lx = S.pb->Ls[3 * j]; ly = S.pb->Ls[3 * j + 1]; lz = S.pb->Ls[3 * j + 2];
      if (PBC == 2) { cj = S.pb->img_phase[2 * j]; sj = S.pb->img_phase[2 * j + 1]; }
    }
  };
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  for (int sh = tid; sh < S.nshell; sh += NTH) {
    const int ia = S.shell_atom[sh];
    sh_xyz[3 * sh] = S.atom_xyz[3 * ia]; sh_xyz[3 * sh + 1] = S.atom_xyz[3 * ia + 1]; sh_xyz[3 * sh + 2] = S.atom_xyz[3 * ia + 2];
    sh_meta[5 * sh] = S.shell_l[sh];
    sh_meta[5 * sh + 1] = S.shell_prim_off[sh + 1] - S.shell_prim_off[sh];
    sh_meta[5 * sh + 2] = S.shell_prim_off[sh];
    sh_meta[5 * sh + 3] = Wt.row[sh];
    sh_meta[5 * sh + 4] = ia;
  }
  for (int p = tid; p < S.nprim; p += NTH) { pr_exp[p] = S.prim_exp[p]; pr_coef[p] = S.prim_coef[p]; }
  for (int k = tid; k < NCOMP * K * 16; k += NTH) tile[k] = 0.0;  // (the K padding rows stay zero)
  __syncthreads();
  PQA_CLK(1);
  const int pl = tid & 15, grp = tid >> 4;
  const long p0 = (long)blockIdx.x * 16;
  const long pmine = (p0 + pl < P) ? p0 + pl : P - 1;
  double px, py, pz;
  load_point(pa, pmine, px, py, pz);
  if (PBC) fold_cell(S, px, py, pz);
  PbcCtx ctx;
  const PrimWrap pw = PBC ? prim_wrap(S, px, py, pz) : PrimWrap{0, 0, 0};
  const int s_end = Wt.off[grp + 1];
  for (int si = Wt.off[grp]; si < s_end; ++si) {
    const int sh = Wt.shell[si];
    const int l_ = sh_meta[5 * sh], np_ = sh_meta[5 * sh + 1], q0 = sh_meta[5 * sh + 2], kb = sh_meta[5 * sh + 3], ia_ = sh_meta[5 * sh + 4];
    const double x = px - sh_xyz[3 * sh], y = py - sh_xyz[3 * sh + 1], z = pz - sh_xyz[3 * sh + 2];
    const double *pe = pr_exp + q0, *pc = pr_coef + q0;
    bool accum = false;  // periodic: the lattice sum accumulates in the tile (shell_eval_pbc)
    auto to_tile = [&](int m, double v, double gx, double gy, double gz, double lp) {
      double* t = tile + (size_t)(kb + m) * 16 + pl;
      if (PBC && accum) {
        unsafeAtomicAdd(t, v);
        if (NCOMP > 1) { unsafeAtomicAdd(t + (size_t)(1 % NCOMP) * K * 16, gx); unsafeAtomicAdd(t + (size_t)(2 % NCOMP) * K * 16, gy); unsafeAtomicAdd(t + (size_t)(3 % NCOMP) * K * 16, gz); }
        if (NCOMP == 5) unsafeAtomicAdd(t + (size_t)(4 % NCOMP) * K * 16, lp);
      } else {
        t[0] = v;
        if (NCOMP > 1) { t[(size_t)(1 % NCOMP) * K * 16] = gx; t[(size_t)(2 % NCOMP) * K * 16] = gy; t[(size_t)(3 % NCOMP) * K * 16] = gz; }
        if (NCOMP == 5) t[(size_t)(4 % NCOMP) * K * 16] = lp;
      }
    };
    if (PBC) {
      pbc_ctx_load<PBC>(S, T, ctx, ia_, P, pmine, x, y, z, pw);
      if (PBC == 2) {
        const int kbi = Wt.row[sh + S.nshell];
        auto to_tile_im = [&](int m, double v, double gx, double gy, double gz, double lp) {
          double* t = tile + (size_t)(kbi + m) * 16 + pl;
          if (accum) {
            unsafeAtomicAdd(t, v);
            if (NCOMP > 1) { unsafeAtomicAdd(t + (size_t)(1 % NCOMP) * K * 16, gx); unsafeAtomicAdd(t + (size_t)(2 % NCOMP) * K * 16, gy); unsafeAtomicAdd(t + (size_t)(3 % NCOMP) * K * 16, gz); }
            if (NCOMP == 5) unsafeAtomicAdd(t + (size_t)(4 % NCOMP) * K * 16, lp);
          } else {
            t[0] = v;
            if (NCOMP > 1) { t[(size_t)(1 % NCOMP) * K * 16] = gx; t[(size_t)(2 % NCOMP) * K * 16] = gy; t[(size_t)(3 % NCOMP) * K * 16] = gz; }
            if (NCOMP == 5) t[(size_t)(4 % NCOMP) * K * 16] = lp;
          }
        };
        shell_eval_pbc<NCOMP, true>(S, ctx, sh, l_, pe, pc, np_, to_tile, to_tile_im, accum, ls_get);
      } else shell_eval_pbc<NCOMP, false>(S, ctx, sh, l_, pe, pc, np_, to_tile, to_tile, accum, ls_get);
    } else shell_eval<NCOMP>(l_, x, y, z, pe, pc, np_, to_tile);
  }
  PQA_CLK(2);
#ifdef PQA_WIDE_CLK
  if (blockIdx.x < 1024 && threadIdx.x == NTH - 64) pqa_wide_clk[blockIdx.x * 8 + 6] = wall_clock64();  // (last wave's phase 1)
#endif
  __syncthreads();
  PQA_CLK(3);
  // contraction: wave <-> (component c, orbital tile ut); D[point][orbital] += A[point][k] B[k][orbital], B straight from L2
  const double* __restrict__ C = T.cpad[spin];
  const int ldc = T.ldc[spin], nmo = S.nmo[spin];
  const int i16 = lane & 15, kq = lane >> 4;
  double* orow[4];
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const long pp = p0 + kq + 4 * r;
    orow[r] = (pp < P) ? orb_out(T, out, pp) + pp * NCOMP * nmo : nullptr;
  }
  for (int role = wv; role < NCOMP * NT; role += NTH / 64) {
    const int c = role % NCOMP, ut = role / NCOMP;
    const double* a_ = tile + (size_t)c * K * 16 + (size_t)kq * 16 + i16;
    const double* b_ = C + (size_t)kq * ldc + 16 * ut + i16;
    d4 acc = {0.0, 0.0, 0.0, 0.0};
    int ks = 0;
    for (; ks + 8 <= K / 4; ks += 8) {
      double av[8], bv[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) { bv[u] = b_[(size_t)(ks + u) * 4 * ldc]; av[u] = a_[(size_t)(ks + u) * 64]; }
#pragma unroll
      for (int u = 0; u < 8; ++u) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(av[u], bv[u], acc, 0, 0, 0);
    }
    for (; ks < K / 4; ++ks) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a_[(size_t)ks * 64], b_[(size_t)ks * 4 * ldc], acc, 0, 0, 0);
    PQA_CLK(4);
    const int j = 16 * ut + i16;
    if (j < nmo) {
#pragma unroll
      for (int r = 0; r < 4; ++r)
        if (orow[r]) orow[r][c * nmo + j] = acc[r];
    }
  }
  PQA_CLK(5);
}
