// Local-energy kernels on device-resident walkers: kinetic + open-boundary Coulomb in one
// per-walker pass, and the semi-local ECP integrator as a compact-list pipeline.
//
// Reference semantics: pyqmc/observables/energy.py (kinetic :57-65, ee/ei :28-45),
// pyqmc/wf/multiplywf.py:121-129 (product Laplacian), pyqmc/observables/eval_ecp.py
// (ecp :21-40, ecp_ea :83-132, ecp_mask :135-146, rnExp :182-200, P_l :203-225, get_P_l :228-252,
// get_rot :255-275, grids :278-336), harness pyqmc/observables/accumulators.py:60-75.
#pragma once
#include "pqa_erfc_tab.hpp"
#include "pqa_common.hpp"
#include "pqa_jastrow.hpp"
#include "pqa_slater.hpp"
#include "pqa_cslater.hpp"

// ---------------------------------------------------------------- kinetic + Coulomb
// out rows: ke, ee, ei, grad2 each (W).  LDS: max(ndet_s)*5 doubles (multi-determinant scratch).
// CX: complex determinants — ke = -1/2 Re(lap Psi / Psi), grad2 = sum |grad Psi / Psi|^2 (energy.py:57-65).
// NWV waves per walker: wave v takes the electrons v, v + NWV, ... (own LDS slice of lds_stride doubles) and the pair / ion
// rows i = v, v + NWV, ... of the Coulomb sums; the block adds the waves' sums in wave order.  All waves run the same number of
// rounds (the helpers use block barriers): a surplus round repeats the last electron and adds nothing.
template <bool CX, int NWV = 1>
static __global__ __launch_bounds__(64 * NWV) void k_kinetic_coulomb(SysDev S, SlaterState st, JastrowState js, int has_slater,
                                                              int has_jastrow, long W, double* __restrict__ out, int lds_stride) {
  extern __shared__ double lds_all[];
  __shared__ double wsum[4][NWV];
  const int wv = (NWV > 1) ? __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)) : 0;
  double* lds = lds_all + (size_t)wv * lds_stride;
  const long w = blockIdx.x;
  const int lane = threadIdx.x & 63;
  const double* xw = js.x + (size_t)w * S.nelec * 3;
  double ke = 0.0, grad2 = 0.0;
  for (int eb = 0; eb < S.nelec; eb += NWV) {
    const bool valid = eb + wv < S.nelec;
    const int e = valid ? eb + wv : S.nelec - 1;
    double gs[3] = {0.0, 0.0, 0.0}, ls = 0.0, gsi[3] = {0.0, 0.0, 0.0};  // gsi: imaginary part of the Slater gradient (CX)
    if (has_slater) {
      const int s = e >= S.nup, i = e - s * S.nup, n = s ? S.ndn : S.nup, nmo = S.nmo[s];
      if (CX) {
        cx r[5];
        slater_ratios_c<5>(S, st, s, i, w, st.cache[s] + ((size_t)w * n + i) * 5 * nmo, r, lds);
        for (int c = 0; c < 3; ++c) { const cx g = cdiv(r[1 + c], r[0]); gs[c] = g.r; gsi[c] = g.i; }
        ls = cdiv(r[4], r[0]).r;  // only the real part of the Laplacian enters (gj is real)
      } else {
        double r[5];
        slater_ratios<5>(S, st, s, i, w, st.cache[s] + ((size_t)w * n + i) * 5 * nmo, r, lds);
        gs[0] = r[1] / r[0]; gs[1] = r[2] / r[0]; gs[2] = r[3] / r[0];
        ls = r[4] / r[0];
      }
    }
    double gj[3] = {0.0, 0.0, 0.0}, lj = 0.0, U;
    if (has_jastrow) {
      jas_eval<2>(S, xw, e, xw[3 * e], xw[3 * e + 1], xw[3 * e + 2], U, gj, lj, 3, lds + S.j3_off);
      lj += gj[0] * gj[0] + gj[1] * gj[1] + gj[2] * gj[2];
    }
    const double gx = gs[0] + gj[0], gy = gs[1] + gj[1], gz = gs[2] + gj[2];
    const double lap = ls + lj + 2.0 * (gs[0] * gj[0] + gs[1] * gj[1] + gs[2] * gj[2]);
    if (valid) {
      ke += -0.5 * lap;
      grad2 += gx * gx + gy * gy + gz * gz;
      if (CX) grad2 += gsi[0] * gsi[0] + gsi[1] * gsi[1] + gsi[2] * gsi[2];
    }
  }
  double ee = 0.0, ei = 0.0;
  for (int i = wv; i < (S.pbc ? 0 : S.nelec); i += NWV) {  // periodic cells: k_ewald fills ee / ei
    const double ix = xw[3 * i], iy = xw[3 * i + 1], iz = xw[3 * i + 2];
    for (int j = i + 1 + lane; j < S.nelec; j += 64) {
      const double dx = ix - xw[3 * j], dy = iy - xw[3 * j + 1], dz = iz - xw[3 * j + 2];
      ee += 1.0 / sqrt(dx * dx + dy * dy + dz * dz);
    }
    for (int I = lane; I < S.natom; I += 64) {
      const double dx = ix - S.atom_xyz[3 * I], dy = iy - S.atom_xyz[3 * I + 1], dz = iz - S.atom_xyz[3 * I + 2];
      ei -= S.atom_charge[I] / sqrt(dx * dx + dy * dy + dz * dz);
    }
  }
  ee = wave_sum(ee);
  ei = wave_sum(ei);
  if (NWV > 1) {
    if (lane == 0) { wsum[0][wv] = ke; wsum[1][wv] = ee; wsum[2][wv] = ei; wsum[3][wv] = grad2; }
    __syncthreads();
    if (threadIdx.x == 0) {
      ke = ee = ei = grad2 = 0.0;
      for (int v = 0; v < NWV; ++v) { ke += wsum[0][v]; ee += wsum[1][v]; ei += wsum[2][v]; grad2 += wsum[3][v]; }
    }
  }
  if (threadIdx.x == 0) { out[w] = ke; out[W + w] = ee; out[2 * W + w] = ei; out[3 * W + w] = grad2; }
}

// ---------------------------------------------------------------- Ewald (observables/ewald.py:238-354)
// ee = sum_{i<j} sum_n erfc(a r_ijn)/r_ijn + sum_G w_G |sum_i e^{iG.x_i}|^2 + ee_const          (:262-275, :293-300)
// ei = sum_{i,I} -Z_I sum_n erfc(a r_iIn)/r_iIn + 2 sum_G w_G (-Re(rho_I) C_G - Im(rho_I) S_G) + ei_const  (:255-259, :301-304)
// with r_n = |minimal-image displacement + n . lattice|, n in {-1,0,1}^3 (real_cij :391-398, nlatvec = 1), G over the
// positive half space with w_G > 1e-10 (:372-388), rho_I = sum_I Z_I e^{iG.R_I} (:233-234) and the self + charged
// constants of :186-190.  One wave per walker; x is [W][N][3] (stride form lets the lane-per-walker state pass its
// transposed coordinates: element (w,e,c) at x[w*sw + e*se + c*sc]).
struct EwaldDev {
  int ng;
  const double* g;        // [ng][3]
  const double* gweight;  // [ng]
  const double* ion_cos;  // [ng] Re rho_I
  const double* ion_sin;  // [ng] Im rho_I
  double alpha, ee_const, ei_const;
  const int* gn;          // [ng][3] integer coordinates of g in the reciprocal basis, or nullptr (direct sincos)
  double recip[9];        // rows: reciprocal basis vectors (g = gn . recip)
  int nmax;               // max |gn| component
};
#define PQA_EWALD_T 256  // threads per walker: the phase tables cost ~20 KB of LDS per block, so one wave per block left 1-2 waves per SIMD
template <int PQA_UNIT = 0>  // (a template so that only the units that launch it compile it)
static __global__ __launch_bounds__(PQA_EWALD_T) void k_ewald(SysDev S, EwaldDev E, const double* __restrict__ x, long sw, long se, long sc,
                                              long W, double* __restrict__ out) {
  extern __shared__ double lds[];  // [N][3] coordinates of this walker
  // erfc(x) = erfcx(x) exp(-x^2), erfcx from the generated piecewise polynomials (tools/gen_erfc_table.py: 52 intervals, degree 9,
  // 2.6e-15 relative): the library erfc was ~80 % of the real-space sum's instructions
  __shared__ double erfc_tab[PQA_ERFC_N][PQA_ERFC_DEG + 1];
  for (int k = threadIdx.x; k < PQA_ERFC_N * (PQA_ERFC_DEG + 1); k += PQA_EWALD_T) erfc_tab[k / (PQA_ERFC_DEG + 1)][k % (PQA_ERFC_DEG + 1)] = PQA_ERFC_TAB[k / (PQA_ERFC_DEG + 1)][k % (PQA_ERFC_DEG + 1)];
  auto erfc_fast = [&](double x) {  // 0 <= x < PQA_ERFC_XMAX (the callers cut at x^2 <= 40)
    const int i = min((int)(x * (1.0 / PQA_ERFC_H)), PQA_ERFC_N - 1);
    const double u = 2.0 * (x - i * PQA_ERFC_H) * (1.0 / PQA_ERFC_H) - 1.0;
    const double* c = erfc_tab[i];
    double p = c[PQA_ERFC_DEG];
#pragma unroll
    for (int k = PQA_ERFC_DEG - 1; k >= 0; --k) p = p * u + c[k];
    return p * exp(-x * x);
  };
  const long w = blockIdx.x;
  const int lane = threadIdx.x;  // 0 .. PQA_EWALD_T-1: the block's threads share the pair / ion / g-point loops
  __shared__ double img_l2[27];  // |L|^2 of the 27 images
  if (lane < 27) {
    const int a = lane / 9 - 1, b = (lane / 3) % 3 - 1, c = lane % 3 - 1;
    const double lx = a * S.pb->lat[0] + b * S.pb->lat[3] + c * S.pb->lat[6], ly = a * S.pb->lat[1] + b * S.pb->lat[4] + c * S.pb->lat[7];
    const double lz = a * S.pb->lat[2] + b * S.pb->lat[5] + c * S.pb->lat[8];
    img_l2[lane] = lx * lx + ly * ly + lz * lz;
  }
  for (int k = lane; k < S.nelec * 3; k += PQA_EWALD_T) lds[k] = x[w * sw + (k / 3) * se + (k % 3) * sc];
  __syncthreads();
  // 27-image real-space sum of one pair.  Each lane first marks which of ITS 27 images are close enough to matter, then walks
  // its own marks: the lanes of a wave hold different pairs, so stepping through the 27 images in lock-step evaluated erfc
  // for nearly every image (some lane always needed it) although a lane needs ~12 of them.  Same terms in the same order.
  auto real_sum = [&](double dx, double dy, double dz) {
    min_image(S, dx, dy, dz);
    const double a2 = E.alpha * E.alpha;
    unsigned m = 0u;
    // the marking pass takes |d + L|^2 = |d|^2 + 2 d.L + |L|^2 (three dot products per pair, then three additions per image
    // instead of nine multiply-adds); an image this estimate puts on the other side of the threshold than the direct form would
    // carries erfc(x)/r < 4e-19, below the last bit of the pair's sum.  The admitted images are evaluated as before.
    const double d2 = dx * dx + dy * dy + dz * dz;
    const double p0 = 2.0 * (dx * S.pb->lat[0] + dy * S.pb->lat[1] + dz * S.pb->lat[2]);
    const double p1 = 2.0 * (dx * S.pb->lat[3] + dy * S.pb->lat[4] + dz * S.pb->lat[5]);
    const double p2 = 2.0 * (dx * S.pb->lat[6] + dy * S.pb->lat[7] + dz * S.pb->lat[8]);
#pragma unroll
    for (int idx = 0; idx < 27; ++idx) {
      const int a = idx / 9 - 1, b = (idx / 3) % 3 - 1, c = idx % 3 - 1;
      const double r2 = d2 + (a * p0 + b * p1 + c * p2) + img_l2[idx];
      // erfc(x)/r < 4e-19 for x^2 > 40: below the last bit of the sum.  (r2 comes from |d|^2 + 2 d.L + |L|^2 and can cancel to
      // a tiny NEGATIVE number next to a lattice point: such an image is marked — its true r2, recomputed below, is ~0.)
      if (!(a2 * r2 > 40.0)) m |= 1u << idx;
    }
    double acc = 0.0;
    while (__any(m != 0u)) {
      if (m) {
        const int idx = __ffs((int)m) - 1;
        m &= m - 1u;
        const int a = idx / 9 - 1, b = (idx / 3) % 3 - 1, c = idx % 3 - 1;
        const double rx = dx + a * S.pb->lat[0] + b * S.pb->lat[3] + c * S.pb->lat[6];
        const double ry = dy + a * S.pb->lat[1] + b * S.pb->lat[4] + c * S.pb->lat[7];
        const double rz = dz + a * S.pb->lat[2] + b * S.pb->lat[5] + c * S.pb->lat[8];
        // 1/r from v_rsq_f64 + two Newton steps (1-2 ulp), r = r^2 (1/r): a third of the IEEE sqrt + division sequences
        const double r2 = rx * rx + ry * ry + rz * rz, hr2 = 0.5 * r2;
        double ir = __builtin_amdgcn_rsq(r2);
        ir = ir * fma(-hr2 * ir, ir, 1.5);
        ir = ir * fma(-hr2 * ir, ir, 1.5);
        // coincident particles (r2 == 0: rsq = inf and the Newton step makes inf * 0 = NaN): erfc(0)/0 = +inf, as the
        // IEEE sequence gave; the term never enters an accepted configuration but must not poison the sum with NaN
        acc += (r2 > 0.0) ? erfc_fast(E.alpha * (r2 * ir)) * ir : __builtin_inf();
      }
    }
    return acc;
  };
  double ee = 0.0, ei = 0.0;
  const int npair = S.nelec * (S.nelec - 1) / 2;
#ifndef PQA_EW_NOREAL
  for (int p = lane; p < npair; p += PQA_EWALD_T) {  // pair p -> (i<j), row-major upper triangle
    int i = 0, rem = p;
    while (rem >= S.nelec - 1 - i) { rem -= S.nelec - 1 - i; ++i; }
    const int j = i + 1 + rem;
    ee += real_sum(lds[3 * i] - lds[3 * j], lds[3 * i + 1] - lds[3 * j + 1], lds[3 * i + 2] - lds[3 * j + 2]);
  }
  for (int q = lane; q < S.nelec * S.natom; q += PQA_EWALD_T) {
    const int e = q / S.natom, I = q % S.natom;
    ei -= S.atom_charge[I] * real_sum(lds[3 * e] - S.atom_xyz[3 * I], lds[3 * e + 1] - S.atom_xyz[3 * I + 1],
                                      lds[3 * e + 2] - S.atom_xyz[3 * I + 2]);
  }
#endif
#ifndef PQA_EW_NORECIP
  if (E.gn) {
    // e^{i g.x_e} = prod_a (e^{i b_a.x_e})^{n_a}: powers 0..nmax of the three base phases of every electron go to LDS
    // (complex multiplication recurrence), a (g, electron) term is then two complex products instead of a sincos.
    const int M = E.nmax + 1;
    double* ph = lds + S.nelec * 3;  // [N][3][M][2]
    for (int q = lane; q < S.nelec * 3; q += PQA_EWALD_T) {
      const int e = q / 3, a = q % 3;
      double sn, cs;
      sincos(E.recip[3 * a] * lds[3 * e] + E.recip[3 * a + 1] * lds[3 * e + 1] + E.recip[3 * a + 2] * lds[3 * e + 2], &sn, &cs);
      double* t = ph + (size_t)q * M * 2;
      double cr = 1.0, ci = 0.0;
      for (int m = 0; m < M; ++m) {
        t[2 * m] = cr; t[2 * m + 1] = ci;
        const double nr = cr * cs - ci * sn;
        ci = cr * sn + ci * cs;
        cr = nr;
      }
    }
    __syncthreads();
    for (int g = lane; g < E.ng; g += PQA_EWALD_T) {
      const int n0 = E.gn[3 * g], n1 = E.gn[3 * g + 1], n2 = E.gn[3 * g + 2];
      const int m0 = abs(n0), m1 = abs(n1), m2 = abs(n2);
      const double f0 = n0 < 0 ? -1.0 : 1.0, f1 = n1 < 0 ? -1.0 : 1.0, f2 = n2 < 0 ? -1.0 : 1.0;
      double sc_ = 0.0, ss_ = 0.0;
      for (int e = 0; e < S.nelec; ++e) {
        const double* t = ph + (size_t)e * 3 * M * 2;
        const double ar = t[2 * m0], ai = f0 * t[2 * m0 + 1];
        const double br = t[2 * (M + m1)], bi = f1 * t[2 * (M + m1) + 1];
        const double cr = t[2 * (2 * M + m2)], ci = f2 * t[2 * (2 * M + m2) + 1];
        const double abr = ar * br - ai * bi, abi = ar * bi + ai * br;
        sc_ += abr * cr - abi * ci;
        ss_ += abr * ci + abi * cr;
      }
      ee += E.gweight[g] * (ss_ * ss_ + sc_ * sc_);
      ei += 2.0 * E.gweight[g] * (-E.ion_cos[g] * sc_ - E.ion_sin[g] * ss_);
    }
  } else {
    for (int g = lane; g < E.ng; g += PQA_EWALD_T) {
      const double gx = E.g[3 * g], gy = E.g[3 * g + 1], gz = E.g[3 * g + 2];
      double sc_ = 0.0, ss_ = 0.0;
      for (int e = 0; e < S.nelec; ++e) {
        double sn, cs;
        sincos(gx * lds[3 * e] + gy * lds[3 * e + 1] + gz * lds[3 * e + 2], &sn, &cs);
        sc_ += cs; ss_ += sn;
      }
      ee += E.gweight[g] * (ss_ * ss_ + sc_ * sc_);
      ei += 2.0 * E.gweight[g] * (-E.ion_cos[g] * sc_ - E.ion_sin[g] * ss_);
    }
  }
#endif
  ee = wave_sum(ee);
  ei = wave_sum(ei);
  __shared__ double part[2][PQA_EWALD_T / 64];
  if ((lane & 63) == 0) { part[0][lane >> 6] = ee; part[1][lane >> 6] = ei; }
  __syncthreads();
  if (lane == 0) {
    double se = 0.0, si = 0.0;
    for (int k = 0; k < PQA_EWALD_T / 64; ++k) { se += part[0][k]; si += part[1][k]; }
    out[W + w] = se + E.ee_const; out[2 * W + w] = si + E.ei_const;
  }
}

// ---------------------------------------------------------------- ECP
struct EcpBuf {
  const double* rot;     // [N][necp][3][3]
  const double* unif;    // [N][necp][W] or NULL -> Philox
  const double* quad;    // [144][3] quadrature directions: rows 0-5 octahedral OA, 6-17 icosahedral IAB, then OAB (18), OABC (26), IABC (32), OABCD (50)
  const double* quadw;   // [144] their weights (eval_ecp.py:325-334)
  uint64_t seed;
  uint32_t step;
  double threshold;
  double* local;         // [W] sum of local channels
  int* cnt;              // [2][nseg W] aux points per (segment, walker) per spin
  long* off;             // [2][nseg W + 1] exclusive scan of cnt
  int nseg;              // 1: the points of a walker are contiguous; necp: ATOM-major — segment k holds the points around ECP atom k of
                         // all walkers, walker by walker, so that a tile of the orbital kernel sits within a few bohr of ONE atom: in a
                         // periodic cell the lanes' image walks (shell_eval_pbc: iterations = the longest list in the wave) then have
                         // similar lengths.  A walker's sum still runs atom by atom, i.e. in the same order.
  double* pts[2];        // [npts_s][3]
  double* wgt[2];        // [npts_s]  sum_l (v_l/prob)(2l+1)P_l(cos) w_i
  int* pte[2];           // [npts_s]  electron index of the point
  int* ptw[2];           // [npts_s]  walker index of the point (thread-per-point accumulation)
  const long* ptot[2];   // device-side totals of the two lists where the host launched over an upper bound (pqa_energy.hip: defer), else nullptr
  double* u0[2];         // [npts_s]  two-body Jastrow exponent U_e at the electron's CURRENT position (same for the 6/12 points of an entry)
  int has_j2;            // fill u0
  const double* ue;      // [N][W] U_e of every electron at its own position (k_kinetic_lw), or NULL: the fill pass sums it itself
  unsigned long long* passbits;  // [W][ceil(N*necp/64)] (electron, atom) pairs that passed the stochastic mask
};

__device__ __forceinline__ double legendre_l(int l, double x) {
  switch (l) {
    case 0: return 1.0;
    case 1: return x;
    case 2: return 0.5 * (3.0 * x * x - 1.0);
    case 3: return 0.5 * (5.0 * x * x * x - 3.0 * x);
    default: return 0.125 * (35.0 * x * x * x * x - 30.0 * x * x + 3.0);
  }
}

// v_l(r) for every channel of ECP atom k (local channel last) and the acceptance probability
__device__ __forceinline__ void ecp_radial(const SysDev& S, int k, double r, double threshold, double (&v)[PQA_MAXCHAN],
                                           int& nch, double& prob) {
  const int c0 = S.ecp_chan_off[k];
  nch = S.ecp_chan_off[k + 1] - c0;
  double pr = 0.0;
  for (int c = 0; c < nch; ++c) {
    double sum = 0.0;
    for (int t = S.ecp_term_off[c0 + c]; t < S.ecp_term_off[c0 + c + 1]; ++t) {
      const int n = S.ecp_term_n[t];
      // integer powers by multiplication (PySCF's r^(n-2), n = 0 ... 4, gives -2 ... 2): `pow` inlined here cost the kernels that
      // call this ~80 vector registers (k_ecp_count: 157 -> the occupancy of 3 waves per SIMD)
      double rn = 1.0;
      if (n != 0) {
        const int an = n < 0 ? -n : n;
        double rp = r;
        for (int q = 1; q < an; ++q) rp *= r;
        rn = n < 0 ? 1.0 / rp : rp;
      }
      sum += rn * S.ecp_term_coef[t] * exp(-S.ecp_term_exp[t] * r * r);
    }
    v[c] = sum;
    if (c < nch - 1) pr += fabs(sum) * threshold * (2.0 * (2 * c + 1) + 1.0);  // eval_ecp.py:139-141
  }
  prob = (threshold > 0.0) ? fmin(1.0, pr) : 1.0;
}

__device__ __forceinline__ bool ecp_pass(const SysDev& S, const EcpBuf& B, long w, long W, int e, int k, double prob) {
  if (!(prob > 0.0)) return false;  // u >= 0: nothing to draw (atoms whose non-local channels vanish)
  double u;
  if (B.unif) u = B.unif[((size_t)e * S.necp + k) * W + w];
  else {
    const Philox p = philox(B.seed, (uint32_t)w, (uint32_t)(e * S.necp + k), PQA_STREAM_ECPMASK, B.step);
    u = u01(p.c[0], p.c[1]);
  }
  return prob > u;
}

// pass A: local part + number of auxiliary points per spin.  grid = W, block = 64.
// Lanes run over the electrons, the loop over the ECP atoms: the atom (its channel / term tables, its coordinates) is
// wave-uniform, so the tables come through the scalar cache and the channel loops do not diverge between O and H.
// passbits[w][k][e-block]: which electrons passed the stochastic mask at atom k (k_ecp_fill walks them atom-major).
// PBC = false compiles the minimal-image code out: its register demand cost the open-system launches a wave per SIMD.
template <bool PBC>
static __global__ __launch_bounds__(64) void k_ecp_count(SysDev S, JastrowState js, EcpBuf B, long W) {
  const long w = blockIdx.x;
  const int lane = threadIdx.x;
  const double* xw = js.x + (size_t)w * S.nelec * 3;
  double loc = 0.0;
  int c_up = 0, c_dn = 0;
  const int neb = (S.nelec + 63) / 64;
  __shared__ unsigned long long pb[64];  // near-atom path: electrons of this block that passed the mask at atom k
  for (int eb = 0; eb < neb; ++eb) {
    const int e = eb * 64 + lane;
    const bool live = e < S.nelec;
    const double ex = live ? xw[3 * e] : 0.0, ey = live ? xw[3 * e + 1] : 0.0, ez = live ? xw[3 * e + 2] : 0.0;
    if (S.necp <= 64) {
      // Every lane (electron) marks the ECP atoms within range of it (S.ecp_rc2: beyond it every term is < 1e-22) and walks ITS
      // OWN marks: an electron is in range of 0-2 atoms of the 24 of the water cluster, and with the atom loop in lock-step all
      // 24 radial evaluations ran for every electron (the kernel was bound by their exp).  What the far atoms would have added to
      // the local energy is below its last bit, and their mask probability is 0.
      unsigned long long near = 0ull;
      for (int k = 0; k < S.necp; ++k) {
        const int ia = S.ecp_atom[k];
        double dx = ex - S.atom_xyz[3 * ia], dy = ey - S.atom_xyz[3 * ia + 1], dz = ez - S.atom_xyz[3 * ia + 2];
        if (PBC) min_image(S, dx, dy, dz);
        if (live && dx * dx + dy * dy + dz * dz < S.ecp_rc2[k]) near |= 1ull << k;
      }
      if (lane < S.necp) pb[lane] = 0ull;
      __syncthreads();
      while (__any(near != 0ull)) {
        if (near) {
          const int k = __ffsll((long long)near) - 1;
          near &= near - 1;
          const int ia = S.ecp_atom[k];
          double dx = ex - S.atom_xyz[3 * ia], dy = ey - S.atom_xyz[3 * ia + 1], dz = ez - S.atom_xyz[3 * ia + 2];
          if (PBC) min_image(S, dx, dy, dz);
          const double r = sqrt(dx * dx + dy * dy + dz * dz);
          double v[PQA_MAXCHAN], prob;
          int nch;
          ecp_radial(S, k, r, B.threshold, v, nch, prob);
          loc += v[nch - 1];
          if (nch > 1 && ecp_pass(S, B, w, W, e, k, prob)) {
            const int naip = S.ecp_naip[k];
            if (e < S.nup) c_up += naip; else c_dn += naip;
            atomicOr(&pb[k], 1ull << lane);
          }
        }
      }
      __syncthreads();
      if (lane < S.necp) B.passbits[((size_t)w * S.necp + lane) * neb + eb] = pb[lane];
      __syncthreads();
      continue;
    }
    for (int k = 0; k < S.necp; ++k) {
      const int ia = S.ecp_atom[k];
      bool pass = false;
      if (live) {
        double dx = ex - S.atom_xyz[3 * ia], dy = ey - S.atom_xyz[3 * ia + 1], dz = ez - S.atom_xyz[3 * ia + 2];
        if (PBC) min_image(S, dx, dy, dz);  // configs.dist.dist_i, eval_ecp.py:95
        const double r = sqrt(dx * dx + dy * dy + dz * dz);
        double v[PQA_MAXCHAN], prob;
        int nch;
        ecp_radial(S, k, r, B.threshold, v, nch, prob);
        loc += v[nch - 1];
        pass = nch > 1 && ecp_pass(S, B, w, W, e, k, prob);
        if (pass) {
          const int naip = S.ecp_naip[k];
          if (e < S.nup) c_up += naip; else c_dn += naip;
        }
      }
      const unsigned long long m = __ballot(pass);
      if (lane == 0) B.passbits[((size_t)w * S.necp + k) * neb + eb] = m;
    }
  }
  loc = wave_sum(loc);
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) { c_up += __shfl_xor(c_up, off, 64); c_dn += __shfl_xor(c_dn, off, 64); }
  if (lane == 0) { B.local[w] = loc; B.cnt[w] = c_up; B.cnt[W + w] = c_dn; }  // (walker-major: nseg = 1)
}

// exclusive scan of cnt[2][W] -> off[2][W+1]; one block of 1024 threads
template <int PQA_UNIT = 0>  // (a template so that only the units that launch it compile it)
static __global__ __launch_bounds__(1024) void k_scan2(const int* __restrict__ cnt, long* __restrict__ off, long W) {
  __shared__ long part[1024];
  for (int s = 0; s < 2; ++s) {
    const int* c = cnt + (size_t)s * W;
    long* o = off + (size_t)s * (W + 1);
    const long per = (W + 1023) / 1024;
    const long b = (long)threadIdx.x * per, e = (b + per < W) ? b + per : W;
    long sum = 0;
    for (long i = b; i < e; ++i) sum += c[i];
    part[threadIdx.x] = sum;
    __syncthreads();
    if (threadIdx.x == 0) {
      long run = 0;
      for (int t = 0; t < 1024; ++t) { const long v = part[t]; part[t] = run; run += v; }
      o[W] = run;
    }
    __syncthreads();
    long run = part[threadIdx.x];
    for (long i = b; i < e; ++i) { o[i] = run; run += c[i]; }
    __syncthreads();
  }
}

// pass B: emit auxiliary points, per-point weights and electron index.  grid = W, block = 64.
template <bool PBC>
static __global__ __launch_bounds__(64) void k_ecp_fill(SysDev S, JastrowState js, EcpBuf B, long W) {
  const long w = blockIdx.x;
  const int lane = threadIdx.x;
  const double* xw = js.x + (size_t)w * S.nelec * 3;
  long run[2] = {B.off[w], B.off[(W + 1) + w]};
  const int neb = (S.nelec + 63) / 64;
  for (int q0 = 0; q0 < S.necp * neb; ++q0) {
    const int k = q0 / neb, eb = q0 % neb;
    unsigned long long m = B.passbits[(size_t)w * S.necp * neb + q0];  // the mask k_ecp_count drew
    while (m) {  // whole wave cooperates on one (electron, atom) entry at a time, atom-major
      const int src = __ffsll((long long)m) - 1;
      m &= m - 1;
      const int e = eb * 64 + src, ia = S.ecp_atom[k], s = e >= S.nup;
      const double ax = S.atom_xyz[3 * ia], ay = S.atom_xyz[3 * ia + 1], az = S.atom_xyz[3 * ia + 2];
      double dx = xw[3 * e] - ax, dy = xw[3 * e + 1] - ay, dz = xw[3 * e + 2] - az;
      if (PBC) min_image(S, dx, dy, dz);
      const double r = sqrt(dx * dx + dy * dy + dz * dz);
      double v[PQA_MAXCHAN], prob;
      int nch;
      ecp_radial(S, k, r, B.threshold, v, nch, prob);
      const int naip = S.ecp_naip[k], qoff = S.ecp_qoff[k];  // (at most 50 points: one lane each)
      double U0 = 0.0;
      if (B.has_j2) {  // once per (electron, atom) entry, by the whole wave, instead of once per point later
        double g_[3], lp_;
        jas_eval<0, PBC>(S, xw, e, xw[3 * e], xw[3 * e + 1], xw[3 * e + 2], U0, g_, lp_, 1);
      }
      if (lane < naip) {
        const double* qd = B.quad + 3 * (qoff + lane);
        const double* R = B.rot + ((size_t)e * S.necp + k) * 9;
        const double vx = R[0] * qd[0] + R[1] * qd[1] + R[2] * qd[2];
        const double vy = R[3] * qd[0] + R[4] * qd[1] + R[5] * qd[2];
        const double vz = R[6] * qd[0] + R[7] * qd[1] + R[8] * qd[2];
        const double rix = r * vx, riy = r * vy, riz = r * vz;  // eval_ecp.py:242
        const double cosv = (dx * rix + dy * riy + dz * riz) / (r * sqrt(rix * rix + riy * riy + riz * riz));
        double wsum = 0.0;
        for (int c = 0; c < nch - 1; ++c) wsum += (v[c] / prob) * (2 * c + 1) * legendre_l(c, cosv);
        const long slot = run[s] + lane;
        B.pts[s][3 * slot] = (xw[3 * e] - dx) + rix;  // eval_ecp.py:110
        B.pts[s][3 * slot + 1] = (xw[3 * e + 1] - dy) + riy;
        B.pts[s][3 * slot + 2] = (xw[3 * e + 2] - dz) + riz;
        B.wgt[s][slot] = wsum * B.quadw[qoff + lane];
        B.pte[s][slot] = e;
        B.ptw[s][slot] = (int)w;
        B.u0[s][slot] = U0;
      }
      run[s] += naip;
    }
  }
}

// pass C: ecp[w] = local + sum_points weight * Psi(aux)/Psi.  mo[s]: [npts_s][nmo_s] orbital values.
// Block = NWV waves on ONE walker: wave v takes the points first + v, first + v + NWV, ... of each spin's list, with its own
// LDS slice (lds_stride doubles: multi-determinant scratch / three-body tables), and the block adds the waves' sums in wave
// order.  The points of a walker are independent, and one wave walking all ~36 of them one after the other — multi-determinant
// ratio, three-body factor — was 0.7 ms per evaluation whatever the walker count below ~4 000 (the 50-determinant molecule at
// 2 048 walkers per GPU).  The helpers synchronise with block barriers, so all waves make the same calls: every wave runs the
// same number of rounds (surplus rounds repeat the last point with weight 0) and re-evaluates the old-position exponent when
// ANY wave meets a new electron.
// CX: complex determinants; the imaginary part of the walker's sum goes to ecp[W + w].
template <bool PBC, bool CX = false, int NWV = 1>
static __global__ __launch_bounds__(64 * NWV) void k_ecp_accum(SysDev S, SlaterState st, JastrowState js, EcpBuf B, int has_slater,
                                                        int has_jastrow, const double* __restrict__ mo_up,
                                                        const double* __restrict__ mo_dn, long W, double* __restrict__ ecp,
                                                        int lds_stride) {
  extern __shared__ double lds_all[];
  __shared__ double wsum[2][NWV];
  const int wv = (NWV > 1) ? __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)) : 0;
  double* lds = lds_all + (size_t)wv * lds_stride;
  const long w = blockIdx.x;
  const double* xw = js.x + (size_t)w * S.nelec * 3;
  double tot = 0.0, tot_im = 0.0;
  for (int s = 0; s < 2; ++s) {
    const double* mo = s ? mo_dn : mo_up;
    const int nmo = S.nmo[s];
    int last_e = -1;
    double U0 = 0.0;
    for (int seg = 0; seg < B.nseg; ++seg) {
    const size_t so = (size_t)s * ((size_t)B.nseg * W + 1) + (size_t)seg * W + w;
    const long p0 = B.off[so], p1 = B.off[so + 1];
    for (long pb = p0; pb < p1; pb += NWV) {
      const bool valid = pb + wv < p1;
      const long p = valid ? pb + wv : p1 - 1;
      const int e = B.pte[s][p];
      double ratio = 1.0, ratio_im = 0.0;
      if (has_slater) {
        if (CX) {
          cx r1[1];
          slater_ratios_c<1>(S, st, s, e - s * S.nup, w, mo + (size_t)p * nmo, r1, lds);
          ratio = r1[0].r; ratio_im = r1[0].i;
        } else {
          double r1[1];
          slater_ratios<1>(S, st, s, e - s * S.nup, w, mo + (size_t)p * nmo, r1, lds);
          ratio = r1[0];
        }
      }
      if (has_jastrow) {
        double g[3], lp, U;
        const bool fresh = (NWV > 1) ? (bool)__syncthreads_or(e != last_e) : (e != last_e);
        if (fresh) { jas_eval<0, PBC>(S, xw, e, xw[3 * e], xw[3 * e + 1], xw[3 * e + 2], U0, g, lp, 3, lds + S.j3_off); last_e = e; }
        jas_eval<0, PBC>(S, xw, e, B.pts[s][3 * p], B.pts[s][3 * p + 1], B.pts[s][3 * p + 2], U, g, lp, 3, lds + S.j3_off);
        const double ej = exp(U - U0);
        ratio *= ej; ratio_im *= ej;
      }
      if (valid) {
        tot += ratio * B.wgt[s][p];
        if (CX) tot_im += ratio_im * B.wgt[s][p];
      }
    }
    }
  }
  if (NWV > 1) {
    if ((threadIdx.x & 63) == 0) { wsum[0][wv] = tot; wsum[1][wv] = tot_im; }
    __syncthreads();
    if (threadIdx.x == 0) {
      tot = 0.0; tot_im = 0.0;
      for (int v = 0; v < NWV; ++v) { tot += wsum[0][v]; tot_im += wsum[1][v]; }
    }
  }
  if (threadIdx.x == 0) {
    ecp[w] = B.local[w] + tot;
    if (CX) ecp[W + w] = tot_im;
  }
}

// uniformly random rotations from a normalised Gaussian quaternion (one per (electron, ECP atom))
template <int PQA_UNIT = 0>  // (a template so that only the units that launch it compile it)
static __global__ void k_gen_rot(int count, uint64_t seed, uint32_t step, double* __restrict__ rot) {
  const int idx = blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= count) return;
  double q0, q1, q2, q3;
  normal2(philox(seed, (uint32_t)idx, 0u, PQA_STREAM_ECPROT, step), q0, q1);
  normal2(philox(seed, (uint32_t)idx, 1u, PQA_STREAM_ECPROT, step), q2, q3);
  const double inv = 1.0 / sqrt(q0 * q0 + q1 * q1 + q2 * q2 + q3 * q3);
  const double w = q0 * inv, x = q1 * inv, y = q2 * inv, z = q3 * inv;
  double* R = rot + 9 * (size_t)idx;
  R[0] = 1 - 2 * (y * y + z * z); R[1] = 2 * (x * y - z * w); R[2] = 2 * (x * z + y * w);
  R[3] = 2 * (x * y + z * w); R[4] = 1 - 2 * (x * x + z * z); R[5] = 2 * (y * z - x * w);
  R[6] = 2 * (x * z - y * w); R[7] = 2 * (y * z + x * w); R[8] = 1 - 2 * (x * x + y * y);
}

// total = ke + ee + ei + ecp + ii ; rows of out: ke, ee, ei, ecp, grad2, total  (accumulators.py:68-75)
// complex determinants: a 7th row holds Im(ecp) = Im(total) (eval_ecp.py:89, accumulators.py:74).
template <int PQA_UNIT = 0>  // (a template so that only the units that launch it compile it)
static __global__ void k_energy_assemble(const double* __restrict__ kc /*ke,ee,ei,grad2*/, const double* __restrict__ ecp,
                                  double ii, long W, double* __restrict__ out, int cplx) {
  const long w = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (w >= W) return;
  const double ke = kc[w], ee = kc[W + w], ei = kc[2 * W + w], g2 = kc[3 * W + w], ec = ecp ? ecp[w] : 0.0;
  out[w] = ke; out[W + w] = ee; out[2 * W + w] = ei; out[3 * W + w] = ec; out[4 * W + w] = g2;
  out[5 * W + w] = ke + ee + ei + ec + ii;
  if (cplx) out[6 * W + w] = ecp ? ecp[W + w] : 0.0;
}

// deterministic column means of a (nrow, W) array: one block of 256 threads per row
template <int PQA_UNIT = 0>  // (a template so that only the units that launch it compile it)
static __global__ __launch_bounds__(256) void k_row_means(const double* __restrict__ a, long W, double* __restrict__ out) {
  __shared__ double part[256];
  const double* row = a + (size_t)blockIdx.x * W;
  double s = 0.0;
  // eight loads in flight per thread (one at a time the 256 dependent round trips of a 65 536-walker row took 64 us); the thread's sum keeps
  // its order: element i, i + 256, ...
  long i = threadIdx.x;
  for (; i + 7 * 256 < W; i += 8 * 256) {
    double v[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) v[u] = row[i + u * 256];
#pragma unroll
    for (int u = 0; u < 8; ++u) s += v[u];
  }
  for (; i < W; i += 256) s += row[i];
  part[threadIdx.x] = s;
  __syncthreads();
  for (int off = 128; off > 0; off >>= 1) {
    if ((int)threadIdx.x < off) part[threadIdx.x] += part[threadIdx.x + off];
    __syncthreads();
  }
  if (threadIdx.x == 0) out[blockIdx.x] = part[0] / (double)W;
}

// k_energy_assemble + k_row_means (+ k_sum_reset_int of the sweep before) in one launch for small shards: block r < nen assembles row r of out
// and sums it the way k_row_means does (the same values in the same order: the same means); block nen adds the walkers' accepted moves.
template <int PQA_UNIT = 0>
static __global__ __launch_bounds__(256) void k_energy_finish(const double* __restrict__ kc, const double* __restrict__ ecp, double ii, long W,
                                                               double* __restrict__ out, double* __restrict__ means, int nen,
                                                               int* __restrict__ acc_w, int* __restrict__ acc_out) {
  const int r = blockIdx.x;
  if (r == nen) {
    __shared__ int ipart[256];
    int s = 0;
    for (long i = threadIdx.x; i < W; i += 256) { s += acc_w[i]; acc_w[i] = 0; }
    ipart[threadIdx.x] = s;
    __syncthreads();
    for (int off = 128; off > 0; off >>= 1) {
      if ((int)threadIdx.x < off) ipart[threadIdx.x] += ipart[threadIdx.x + off];
      __syncthreads();
    }
    if (threadIdx.x == 0) *acc_out = ipart[0];
    return;
  }
  __shared__ double part[256];
  double s = 0.0;
  for (long w = threadIdx.x; w < W; w += 256) {
    double v;
    if (r == 0) v = kc[w];
    else if (r == 1) v = kc[W + w];
    else if (r == 2) v = kc[2 * W + w];
    else if (r == 3) v = ecp ? ecp[w] : 0.0;
    else if (r == 4) v = kc[3 * W + w];
    else if (r == 5) { const double ke = kc[w], ee = kc[W + w], ei = kc[2 * W + w], ec = ecp ? ecp[w] : 0.0; v = ke + ee + ei + ec + ii; }
    else v = ecp ? ecp[W + w] : 0.0;
    out[(size_t)r * W + w] = v;
    s += v;
  }
  part[threadIdx.x] = s;
  __syncthreads();
  for (int off = 128; off > 0; off >>= 1) {
    if ((int)threadIdx.x < off) part[threadIdx.x] += part[threadIdx.x + off];
    __syncthreads();
  }
  if (threadIdx.x == 0) means[r] = part[0] / (double)W;
}

// pass C, thread-per-point variant (single determinant, no three-body factor): contrib[p] = weight_p * Psi(aux_p)/Psi.
// One wave per walker (k_ecp_accum) walks ~38 points one after another with every load and reduction latency
// exposed; here each point is one thread: a 32-term dot with the walker's inverse row for the determinant ratio and
// one loop over the other electrons and the ions for U(new) - U(old).  k_ecp_sum then adds a walker's points in slot
// order, so the result does not depend on scheduling.
template <bool PBC>
static __global__ __launch_bounds__(256) void k_ecp_point(SysDev S, SlaterState st, JastrowState js, EcpBuf B, int s, int has_slater,
                                                   int has_jastrow, const double* __restrict__ mo, long npts,
                                                   double* __restrict__ contrib, const double* __restrict__ Tbase, long sw, long si, long sk) {
  // Tbase / sw / si / sk: element (walker, electron row, column) of the inverse at Tbase[w sw + i si + k sk] — the
  // walker-major array (sw = n^2, si = n, sk = 1) or the lane-per-walker planes (sw = 1, si = n W, sk = W), which spares the
  // fused sweep a transpose of every walker's inverse per energy evaluation (35 KB per walker moved for ~2 KB read here)
  const long p = (long)blockIdx.x * 256 + threadIdx.x;
  if (p >= npts || (B.ptot[s] && p >= *B.ptot[s])) return;
  const int e = B.pte[s][p];
  const long w = B.ptw[s][p];
  const int n = s ? S.ndn : S.nup, i = e - s * S.nup, nmo = S.nmo[s];
  double ratio = 1.0;
  if (has_slater) {
    const double* Ti = Tbase + (size_t)w * sw + (size_t)i * si;
    const double* row = mo + (size_t)p * nmo;
    const int* occ = S.det_occ[s];
    double r = 0.0;
    // The lanes of a wave walk 64 different 256-byte rows: with 8-byte loads that is 64 cache lines per load instruction and no
    // reuse in L1 (0.26 of this kernel's 0.83 ms, compile-time ablation).  Where the determinant occupies the first n orbitals
    // in order (the usual ground-state list) the row is read 32 bytes at a time instead.
    if (S.occ_ident[s] && (n % 4) == 0 && (nmo % 4) == 0) {
      const double4* row4 = reinterpret_cast<const double4*>(row);
      for (int k4 = 0; k4 < n / 4; ++k4) {
        const double4 q = row4[k4];
        const double* Tk = Ti + (size_t)(4 * k4) * sk;
        r += q.x * Tk[0]; r += q.y * Tk[sk]; r += q.z * Tk[2 * sk]; r += q.w * Tk[3 * sk];
      }
    } else
      for (int k = 0; k < n; ++k) r += row[occ[k]] * Ti[(size_t)k * sk];
    ratio = r;
  }
  if (has_jastrow) {
    const double* xw = js.x + (size_t)w * S.nelec * 3;
    const double nx = B.pts[s][3 * p], ny = B.pts[s][3 * p + 1], nz = B.pts[s][3 * p + 2];
    const int edown = e >= S.nup;
    const double irb = 1.0 / S.rcut_b, ira = 1.0 / S.rcut_a;
    double du = 0.0;
    auto norm = [&](double dx, double dy, double dz) {
      if (PBC) min_image_j(S, dx, dy, dz);
      return sqrt(dx * dx + dy * dy + dz * dz);
    };
    for (int j = 0; j < S.nelec; ++j) {
      if (j == e) continue;
      const double jx = xw[3 * j], jy = xw[3 * j + 1], jz = xw[3 * j + 2];
      const double rn = norm(nx - jx, ny - jy, nz - jz);
      const int col = edown + (j >= S.nup);
      if (rn < S.rcut_b) {
        const RadShared sh = rad_shared<0>(rn, irb);
        for (int l = 0; l < S.nb; ++l) {
          double v, g, lp;
          rad_fn<0>(S.b_kind[l], S.b_param[l], S.b_aux[l], S.rcut_b, sh, v, g, lp);
          du += S.bcoeff[l * 3 + col] * v;
        }
      }
    }
    for (int I = 0; I < S.natom; ++I) {
      const double ax = S.atom_xyz[3 * I], ay = S.atom_xyz[3 * I + 1], az = S.atom_xyz[3 * I + 2];
      const double rn = norm(nx - ax, ny - ay, nz - az);
      if (rn < S.rcut_a) {
        const RadShared sh = rad_shared<0>(rn, ira);
        for (int k = 0; k < S.na; ++k) {
          double v, g, lp;
          rad_fn<0>(S.a_kind[k], S.a_param[k], S.a_aux[k], S.rcut_a, sh, v, g, lp);
          du += S.acoeff[(I * S.na + k) * 2 + edown] * v;
        }
      }
    }
    ratio *= exp(du - B.u0[s][p]);  // U_e(new) - U_e(old); the old-position sum comes from k_ecp_fill
  }
  contrib[p] = ratio * B.wgt[s][p];
}

// ecp[w] = local + sum of the walker's point contributions, spin up then spin down, in slot order
// n_up / n_dn > 0: complex contributions, imaginary parts at c[n + p]; their sum goes to ecp[W + w]
template <int PQA_UNIT = 0>  // (a template so that only the units that launch it compile it)
static __global__ __launch_bounds__(256) void k_ecp_sum(EcpBuf B, const double* __restrict__ c_up, const double* __restrict__ c_dn, long W,
                                                 double* __restrict__ ecp, long n_up = 0, long n_dn = 0) {
  const long w = (long)blockIdx.x * 256 + threadIdx.x;
  if (w >= W) return;
  double tot = 0.0, tim = 0.0;
  const size_t SS = (size_t)B.nseg * W + 1;
  for (int seg = 0; seg < B.nseg; ++seg)
    for (long p = B.off[(size_t)seg * W + w]; p < B.off[(size_t)seg * W + w + 1]; ++p) { tot += c_up[p]; if (n_up > 0) tim += c_up[n_up + p]; }
  for (int seg = 0; seg < B.nseg; ++seg)
    for (long p = B.off[SS + (size_t)seg * W + w]; p < B.off[SS + (size_t)seg * W + w + 1]; ++p) { tot += c_dn[p]; if (n_dn > 0) tim += c_dn[n_dn + p]; }
  ecp[w] = B.local[w] + tot;
  if (n_up > 0 || n_dn > 0) ecp[W + w] = tim;
}

// ---------------------------------------------------------------- T-move candidates (DMC)
// eval_ecp.compute_tmoves (eval_ecp.py:43-80) for ONE electron e: every ECP atom's quadrature points.
// Point q of walker w belongs to ECP atom pt_k[q], quadrature index pt_i[q].  Outputs per (w,q):
//   pos (3)   position of the candidate (the current position of e where the walker fails the ECP mask)
//   weight    sum_l (exp(-tau v_l/prob) - 1)(2l+1) P_l(cos) w_i   (0 where masked out)
//   live      1 where the walker passed the mask for that atom
// rot [necp][3][3], unif [necp][W].  grid = W, block = 64.
template <int PQA_UNIT = 0>  // (a template so that only the units that launch it compile it)
static __global__ __launch_bounds__(64) void k_tmove_points(SysDev S, JastrowState js, int e, double tau, double threshold,
                                                     const double* __restrict__ rot, const double* __restrict__ unif,
                                                     const double* __restrict__ quad, const int* __restrict__ pt_k,
                                                     const int* __restrict__ pt_i, int P, long W, double* __restrict__ pos,
                                                     double* __restrict__ weight, uint8_t* __restrict__ live) {
  const long w = blockIdx.x;
  const double* xw = js.x + (size_t)w * S.nelec * 3;
  const double ex = xw[3 * e], ey = xw[3 * e + 1], ez = xw[3 * e + 2];
  for (int q = threadIdx.x; q < P; q += 64) {
    const int k = pt_k[q], i = pt_i[q], ia = S.ecp_atom[k];
    double dx = ex - S.atom_xyz[3 * ia], dy = ey - S.atom_xyz[3 * ia + 1], dz = ez - S.atom_xyz[3 * ia + 2];
    min_image(S, dx, dy, dz);
    const double r = sqrt(dx * dx + dy * dy + dz * dz);
    double v[PQA_MAXCHAN], prob;
    int nch;
    ecp_radial(S, k, r, threshold, v, nch, prob);
    const bool pass = nch > 1 && prob > unif[(size_t)k * W + w];
    double px = ex, py = ey, pz = ez, wt = 0.0;
    if (pass) {
      const int naip = (nch <= 2) ? 6 : 12;
      const double* qd = quad + ((nch <= 2) ? 0 : 18) + 3 * i;
      const double* R = rot + (size_t)k * 9;
      const double vx = R[0] * qd[0] + R[1] * qd[1] + R[2] * qd[2];
      const double vy = R[3] * qd[0] + R[4] * qd[1] + R[5] * qd[2];
      const double vz = R[6] * qd[0] + R[7] * qd[1] + R[8] * qd[2];
      const double rix = r * vx, riy = r * vy, riz = r * vz;
      const double cosv = (dx * rix + dy * riy + dz * riz) / (r * sqrt(rix * rix + riy * riy + riz * riz));
      for (int c = 0; c < nch - 1; ++c) wt += (exp(-tau * (v[c] / prob)) - 1.0) * (2 * c + 1) * legendre_l(c, cosv);
      wt *= 1.0 / naip;
      px = (ex - dx) + rix; py = (ey - dy) + riy; pz = (ez - dz) + riz;
    }
    const size_t o = (size_t)w * P + q;
    pos[3 * o] = px; pos[3 * o + 1] = py; pos[3 * o + 2] = pz;
    weight[o] = wt;
    live[o] = pass;
  }
}

// ratio[w][q] = Psi(candidate)/Psi for live candidates, 1 otherwise.  mo: [W*P][nmo_s].  LDS: max(ndet_s) doubles.
template <int PQA_UNIT = 0>  // (a template so that only the units that launch it compile it)
static __global__ __launch_bounds__(64) void k_tmove_ratio(SysDev S, SlaterState st, JastrowState js, int e, int has_slater,
                                                    int has_jastrow, const double* __restrict__ mo,
                                                    const double* __restrict__ pos, const uint8_t* __restrict__ live, int P,
                                                    double* __restrict__ ratio) {
  extern __shared__ double lds[];
  const long w = blockIdx.x;
  const int s = e >= S.nup, nmo = S.nmo[s];
  const double* xw = js.x + (size_t)w * S.nelec * 3;
  double U0 = 0.0, g[3], lp;
  if (has_jastrow) jas_eval<0>(S, xw, e, xw[3 * e], xw[3 * e + 1], xw[3 * e + 2], U0, g, lp, 3, lds + S.j3_off);
  for (int q = 0; q < P; ++q) {
    const size_t o = (size_t)w * P + q;
    double rat = 1.0;
    if (live[o]) {  // wave-uniform
      if (has_slater) {
        double r1[1];
        slater_ratios<1>(S, st, s, e - s * S.nup, w, mo + o * nmo, r1, lds);
        rat = r1[0];
      }
      if (has_jastrow) {
        double U;
        jas_eval<0>(S, xw, e, pos[3 * o], pos[3 * o + 1], pos[3 * o + 2], U, g, lp, 3, lds + S.j3_off);
        rat *= exp(U - U0);
      }
    }
    if (threadIdx.x == 0) ratio[o] = rat;
  }
}

// ---------------------------------------------------------------- testvalue_many (density-matrix accumulators)
// out[r][idx] = Psi(electron es[idx] moved to pts[r]) / Psi for every listed electron, ONE auxiliary position per row
// (Slater.testvalue_many slater.py:448-460, JastrowSpin.testvalue_many jastrowspin.py:421-455, ThreeBodyJastrow
// three_body_jastrow.py:343-372, product multiplywf.py:112-114).  factors: bit 0 Slater, bit 1 two-body, bit 2 three-body.
// mo_up / mo_dn: [nrow][nmo_s] orbital values at the auxiliary positions.  Block = one wave per row.
template <bool CX>
static __global__ __launch_bounds__(64) void k_testvalue_many(SysDev S, SlaterState st, JastrowState js, const int* __restrict__ es, int ne,
                                                       const double* __restrict__ pts, const double* __restrict__ mo_up,
                                                       const double* __restrict__ mo_dn, long nrow,
                                                       const int* __restrict__ widx, int factors, double* __restrict__ out) {
  extern __shared__ double lds[];
  const long r = blockIdx.x;
  const long w = widx ? widx[r] : r;
  const double* xw = js.x + (size_t)w * S.nelec * 3;
  const double px = pts[3 * r], py = pts[3 * r + 1], pz = pts[3 * r + 2];
  const int parts = (factors >> 1) & 3;
  for (int idx = 0; idx < ne; ++idx) {
    const int e = es[idx], s = e >= S.nup, i = e - s * S.nup;
    double vr = 1.0, vi = 0.0;  // CX: complex determinant ratio (rows [Re | Im]), real Jastrow factor
    if (factors & 1) {
      if (CX) {
        cx r1[1];
        slater_ratios_c<1>(S, st, s, i, w, (s ? mo_dn : mo_up) + (size_t)r * S.nmo[s], r1, lds);
        vr = r1[0].r; vi = r1[0].i;
      } else {
        double r1[1];
        slater_ratios<1>(S, st, s, i, w, (s ? mo_dn : mo_up) + (size_t)r * S.nmo[s], r1, lds);
        vr = r1[0];
      }
      __syncthreads();
    }
    if (parts) {
      double g[3], lp, U0, U;
      jas_eval<0>(S, xw, e, xw[3 * e], xw[3 * e + 1], xw[3 * e + 2], U0, g, lp, parts, lds + S.j3_off);
      __syncthreads();
      jas_eval<0>(S, xw, e, px, py, pz, U, g, lp, parts, lds + S.j3_off);
      __syncthreads();
      const double f = exp(U - U0);
      vr *= f; vi *= f;
    }
    if (threadIdx.x == 0) {
      if (CX) { out[2 * ((size_t)r * ne + idx)] = vr; out[2 * ((size_t)r * ne + idx) + 1] = vi; }
      else out[(size_t)r * ne + idx] = vr;
    }
  }
}
