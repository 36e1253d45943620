// Lane-per-walker ("LW") kernels for the fused VMC sweep and the kinetic energy.
//
// Measured on MI355X (profiles/r01_*): with one walker per wavefront the per-move kernels are
// VALU-issue bound — ~2.3-2.6 k instructions per walker-move, most of them replicated scalar work,
// cross-lane reductions and half-empty lanes (32 orbitals, 24 ions on 64 lanes) — while moving
// only ~20 KB per walker.  Mapping one WALKER per LANE over structure-of-arrays state removes the
// reductions and the idle lanes (every load is a coalesced 512-B wave access) and cuts the issue
// cost per walker-move by more than 10x.  The arithmetic and its reference semantics are unchanged:
//   propose / accept : vmc_worker body, pyqmc/method/mc.py:115-137 (limdrift :76-89)
//   Slater ratios    : slater.py:342-418 (single determinant), Sherman-Morrison slater.py:88-94
//   Jastrow          : jastrowspin.py:296-385 with func3d.py radial functions
//   kinetic, Coulomb : observables/energy.py:28-65, product Laplacian multiplywf.py:121-129
// These kernels handle the single-determinant case; multi-determinant handles use the
// wave-per-walker kernels of pqa_vmc.hpp.
//
// SoA state (walker index fastest):
//   xt   [N][3][W]            coordinates
//   Tt   [s] [n][n][W]        inverse, electron-major: Tt[i][k][w] = inverse[k][i]
//   rc   [s] [n][2][W][5][nmo] cached MO value/grad/lap rows of every electron, TWO slots of point-major rows (the layout
//                              the orbital kernel writes), sel [s][n][W] = the slot that holds the current position's row.
//        A proposal of electron i is written by the orbital kernel straight into the walker's OTHER slot and accepting it
//        flips the selector byte: no copy (round 2 kept the cache walker-fastest and copied / transposed every accepted row
//        into it: 40 us of the ~160 us a move's bookkeeping kernel took at 65536 walkers, 3.8 ms of a 31.7 ms step).
// Complex determinants (CX instantiations; conventions of pqa_cslater.hpp): nmo counts REAL columns [Re | Im], so orbital o of a
// row is (row[o], row[nmo/2 + o]); the inverse is (re, im) interleaved in the walker-major layout, i.e. planes
// Tt[i][k][2][W] after the transpose; dsign is a unit phase [W][2]; V / R buffers hold 2 n planes.  The drift uses
// Re(grad Psi / Psi) and the acceptance |ratio|^2 (mc.py:118,131).
#pragma once
#include "pqa_common.hpp"
#include "pqa_jastrow.hpp"
#include "pqa_vmc.hpp"

struct LwState {
  double* xt;
  double* Tt[2];
  double* rc[2];
  uint8_t* sel[2];
  double* dsign[2];  // [W] (single determinant) — shared with SlaterState
  double* dlog[2];
  double* auxt;      // [8][W]: scaled gaussian (3), limited drift (3), U_old, ratio
};
// cached / proposed orbital row [5][nmo] of electron i (of spin s) of walker w in slot `slot`
__device__ __forceinline__ double* lw_row(const LwState& L, int s, int i, int slot, long w, long W, int nmo) {
  return L.rc[s] + (((size_t)i * 2 + slot) * W + w) * 5 * nmo;
}

// ---------------------------------------------------------------- layout transposes
// in [R][C] -> out [C][R] through a 32x33 LDS tile; block (32,8)
template <int PQA_UNIT = 0>  // (a template so that only the units that launch it compile it)
static __global__ __launch_bounds__(256) void k_transpose(const double* __restrict__ in, double* __restrict__ out, long R, long C) {
  __shared__ double tile[32][33];
  const long c0 = (long)blockIdx.x * 32, r0 = (long)blockIdx.y * 32;
  for (int j = threadIdx.y; j < 32; j += 8) {
    const long r = r0 + j, c = c0 + threadIdx.x;
    if (r < R && c < C) tile[j][threadIdx.x] = in[r * C + c];
  }
  __syncthreads();
  for (int j = threadIdx.y; j < 32; j += 8) {
    const long c = c0 + j, r = r0 + threadIdx.x;
    if (r < R && c < C) out[c * R + r] = tile[threadIdx.x][j];
  }
}

// walker-major orbital cache [W][n][5 nmo] <-> the two-slot row cache.  to_rc: everything lands in slot 0 (selectors cleared);
// from_rc: every electron's CURRENT slot.  grid = (W, n, ceil(row / 256)), block = 256.
template <int PQA_UNIT = 0>  // (a template so that only the units that launch it compile it)
static __global__ __launch_bounds__(256) void k_cache_to_rc(const double* __restrict__ aos, double* __restrict__ rc, uint8_t* __restrict__ sel, int n,
                                                     int row, long W) {
  const long w = blockIdx.x;
  const int i = blockIdx.y, k = blockIdx.z * 256 + threadIdx.x;
  if (k == 0) sel[(size_t)i * W + w] = 0;
  if (k < row) rc[(((size_t)i * 2) * W + w) * row + k] = aos[((size_t)w * n + i) * row + k];
}
template <int PQA_UNIT = 0>  // (a template so that only the units that launch it compile it)
static __global__ __launch_bounds__(256) void k_cache_from_rc(const double* __restrict__ rc, const uint8_t* __restrict__ sel, double* __restrict__ aos,
                                                       int n, int row, long W) {
  const long w = blockIdx.x;
  const int i = blockIdx.y, k = blockIdx.z * 256 + threadIdx.x;
  if (k < row) aos[((size_t)w * n + i) * row + k] = rc[(((size_t)i * 2 + sel[(size_t)i * W + w]) * W + w) * row + k];
}

// ---------------------------------------------------------------- per-lane Jastrow
// Contribution of the pairs j = j0, j0+dj, ... and ions I = j0, j0+dj, ... to U_e, grad U_e, (bare) lap U_e
// of electron e of walker w at (rx,ry,rz); MODE 2 also returns that share of the Coulomb sums
// ee = sum_{j>e} 1/r, ei = -sum Z/r.  (j0,dj) = (0,1) gives the full sums.
#ifndef PQA_JAS_PF
#define PQA_JAS_PF 4
#endif
#define PQA_JAS_NF 4  // basis functions per kind whose tables the fast path keeps in scalar registers
// FAST (nb, na <= PQA_JAS_NF, the reference's default Jastrow has 4 + 4): the function tables (kind, parameter, cusp constant,
// the two coefficient columns electron e can meet) are read ONCE into scalar registers.  Indexed by the loop variable they
// were scalar loads inside the innermost loop — two dependent load-and-wait pairs per function and pair, ~90 per thread —
// and with four waves per SIMD those waits, not the arithmetic, set the kernel's time.  Same operations in the same order.
template <int MODE, bool PBC, bool FAST, int PF = PQA_JAS_PF>
__device__ __forceinline__ void jas_eval_lane_t(const SysDev& S, const double* xt, long W, long w, int e,
                                                double rx, double ry, double rz, int has_jastrow, int j0, int dj, double& U,
                                                double (&g)[3], double& lapU, double& ee, double& ei, int skip = -1, bool ions = true) {
  constexpr int NF = PQA_JAS_NF;
  const int edown = e >= S.nup;
  const double irb = 1.0 / S.rcut_b, ira = 1.0 / S.rcut_a;
  double u_ = 0.0, gx = 0.0, gy = 0.0, gz = 0.0, lp = 0.0, see = 0.0, sei = 0.0;
  int bk[NF], ak[NF];
  double bp[NF], ba[NF], bc0[NF], bc1[NF], ap[NF], aa[NF];
  if (FAST) {
#pragma unroll
    for (int l = 0; l < NF; ++l) {
      bk[l] = S.b_kind[l]; bp[l] = S.b_param[l]; ba[l] = S.b_aux[l];
      ak[l] = S.a_kind[l]; ap[l] = S.a_param[l]; aa[l] = S.a_aux[l];
      bc0[l] = (has_jastrow && l < S.nb) ? S.bcoeff[l * 3 + edown] : 0.0;
      bc1[l] = (has_jastrow && l < S.nb) ? S.bcoeff[l * 3 + edown + 1] : 0.0;
    }
  }
  // The partners' coordinates are fetched PF at a time before any of them is used.
  for (int jb = j0; jb < S.nelec; jb += PF * dj) {
    double cx[PF], cy[PF], cz[PF];
#pragma unroll
    for (int u = 0; u < PF; ++u) {
      const int j = jb + u * dj;
      const double* xj = xt + (size_t)(j < S.nelec ? j : e) * 3 * W + w;  // past the end: any valid address, never used
      cx[u] = xj[0]; cy[u] = xj[W]; cz[u] = xj[2 * W];
    }
#pragma unroll
    for (int u = 0; u < PF; ++u) {
      const int j = jb + u * dj;
      if (j >= S.nelec || j == e || j == skip) continue;
      double dx = rx - cx[u], dy = ry - cy[u], dz = rz - cz[u];
      if (PBC) min_image_j(S, dx, dy, dz);  // compiled out of the open-boundary instantiation (the hot path of the headline bench)
      const double r = sqrt(dx * dx + dy * dy + dz * dz);
      if (MODE == 2 && j > e) see += fast_rcp(r);
      if (has_jastrow && r < S.rcut_b) {
        const RadShared sh = rad_shared<MODE>(r, irb);
        const bool hi = j >= S.nup;
        double sg = 0.0;
        if (FAST) {
#pragma unroll
          for (int l = 0; l < NF; ++l) {
            if (l < S.nb) {
              double v, gf, lpl;
              rad_fn<MODE>(bk[l], bp[l], ba[l], S.rcut_b, sh, v, gf, lpl);
              const double c = hi ? bc1[l] : bc0[l];
              u_ += c * v;
              if (MODE >= 1) sg += c * gf;
              if (MODE == 2) lp += c * lpl;
            }
          }
        } else {
          const int col = edown + (hi ? 1 : 0);
          for (int l = 0; l < S.nb; ++l) {
            double v, gf, lpl;
            rad_fn<MODE>(S.b_kind[l], S.b_param[l], S.b_aux[l], S.rcut_b, sh, v, gf, lpl);
            const double c = S.bcoeff[l * 3 + col];
            u_ += c * v;
            if (MODE >= 1) sg += c * gf;
            if (MODE == 2) lp += c * lpl;
          }
        }
        if (MODE >= 1) { gx += sg * dx; gy += sg * dy; gz += sg * dz; }
      }
    }
  }
  for (int I = ions ? j0 : S.natom; I < S.natom; I += dj) {
    double dx = rx - S.atom_xyz[3 * I], dy = ry - S.atom_xyz[3 * I + 1], dz = rz - S.atom_xyz[3 * I + 2];
    double ac[NF];
    if (FAST) {
#pragma unroll
      for (int k = 0; k < NF; ++k) ac[k] = (has_jastrow && k < S.na) ? S.acoeff[(I * S.na + k) * 2 + edown] : 0.0;
    }
    if (PBC) min_image_j(S, dx, dy, dz);
    const double r = sqrt(dx * dx + dy * dy + dz * dz);
    if (MODE == 2) sei -= S.atom_charge[I] * fast_rcp(r);
    if (has_jastrow && r < S.rcut_a) {
      const RadShared sh = rad_shared<MODE>(r, ira);
      double sg = 0.0;
      if (FAST) {
#pragma unroll
        for (int k = 0; k < NF; ++k) {
          if (k < S.na) {
            double v, gf, lpl;
            rad_fn<MODE>(ak[k], ap[k], aa[k], S.rcut_a, sh, v, gf, lpl);
            u_ += ac[k] * v;
            if (MODE >= 1) sg += ac[k] * gf;
            if (MODE == 2) lp += ac[k] * lpl;
          }
        }
      } else {
        for (int k = 0; k < S.na; ++k) {
          double v, gf, lpl;
          rad_fn<MODE>(S.a_kind[k], S.a_param[k], S.a_aux[k], S.rcut_a, sh, v, gf, lpl);
          const double c = S.acoeff[(I * S.na + k) * 2 + edown];
          u_ += c * v;
          if (MODE >= 1) sg += c * gf;
          if (MODE == 2) lp += c * lpl;
        }
      }
      if (MODE >= 1) { gx += sg * dx; gy += sg * dy; gz += sg * dz; }
    }
  }
  U = u_; g[0] = gx; g[1] = gy; g[2] = gz; lapU = lp; ee = see; ei = sei;
}
// MERGED (S.jq_on; callers whose electron AND partner index are wave-uniform, UJ): the Pade functions of a basis as one rational function of p
// per pair (pade_merged), the numerator record picked by the pair's spin channel / the ion with scalar selects and loads; a cusp
// function at index 0 is evaluated as before.  r and 1 / r from one v_rsq_f64 (sqrt_rinv).  Same sums as the FAST route up to
// rounding (the reciprocals are not the same operations), a third fewer instructions per pair.
// UNI = false: electron and partner index may differ between lanes (the narrow step kernel, the ECP passes): the same code with
// per-lane table reads — taken only where the FAST route does not apply (more than PQA_JAS_NF functions in a basis, i.e. the
// ion-cusp function of all-electron atoms next to four Pade functions), instead of the table walk in the innermost loop.
// ELANE (with UNI): the electron differs between the lanes of a wave but its SPIN does not (the thread-per-point ECP kernel: one launch, or one
// grid row, per spin channel) — the records are still picked with scalar selects, only the j == e test is per lane.
template <int MODE, bool PBC, bool UNI = true, bool ELANE = false, int PF = PQA_JAS_PF>
__device__ __forceinline__ void jas_eval_lane_m(const SysDev& S, const double* xt, long W, long w, int e,
                                                double rx, double ry, double rz, int has_jastrow, int j0, int dj, double& U,
                                                double (&g)[3], double& lapU, double& ee, double& ei, int skip, bool ions) {
  if (UNI) {
    j0 = __builtin_amdgcn_readfirstlane(j0); dj = __builtin_amdgcn_readfirstlane(dj); skip = __builtin_amdgcn_readfirstlane(skip);
    if (!ELANE) e = __builtin_amdgcn_readfirstlane(e);  // UJ callers: the electron is wave-uniform too (kernel argument / block index)
  }
  const int edown = (UNI && ELANE) ? __builtin_amdgcn_readfirstlane((int)(e >= S.nup)) : (int)(e >= S.nup);
  const double irb = 1.0 / S.rcut_b, ira = 1.0 / S.rcut_a;
  double u_ = 0.0, gx = 0.0, gy = 0.0, gz = 0.0, lp = 0.0, see = 0.0, sei = 0.0;
  const bool jb_on = has_jastrow && S.nb > 0, ja_on = has_jastrow && S.na > 0;
  const bool bcusp = S.b_kind[0] == 1, acusp = S.a_kind[0] == 1;
  const double bcp = S.b_param[0], bca = S.b_aux[0], acp = S.a_param[0], aca = S.a_aux[0];
  const double bcc0 = (jb_on && bcusp) ? S.bcoeff[edown] : 0.0, bcc1 = (jb_on && bcusp) ? S.bcoeff[edown + 1] : 0.0;
  const double* qb0 = S.bq + edown * PQA_JQ;
  double Db[5], Da[5];
#pragma unroll
  for (int i = 0; i < 5; ++i) { Db[i] = S.b_D[i]; Da[i] = S.a_D[i]; }
  const bool kb4 = S.jq_b > 3, ka4 = S.jq_a > 3;
  for (int jb = j0; jb < S.nelec; jb += PF * dj) {
    double cx[PF], cy[PF], cz[PF];
#pragma unroll
    for (int u = 0; u < PF; ++u) {
      const int j = jb + u * dj;
      const double* xj = xt + (size_t)(j < S.nelec ? j : e) * 3 * W + w;
      cx[u] = xj[0]; cy[u] = xj[W]; cz[u] = xj[2 * W];
    }
#pragma unroll
    for (int u = 0; u < PF; ++u) {
      const int j = jb + u * dj;
      if (j >= S.nelec || j == e || j == skip) continue;
      double dx = rx - cx[u], dy = ry - cy[u], dz = rz - cz[u];
      if (PBC) min_image_j(S, dx, dy, dz);
      double r, ri;
      sqrt_rinv(dx * dx + dy * dy + dz * dz, r, ri);
      if (MODE == 2 && j > e) see += ri;
      if (jb_on && r < S.rcut_b) {
        const RadShared sh = rad_shared_ri<MODE>(r, ri, irb);
        const bool hi = j >= S.nup;
        const double* q = qb0 + (hi ? PQA_JQ : 0);
        const MergedSums m = kb4 ? pade_merged<MODE, 4, UNI>(Db, q, sh.p) : pade_merged<MODE, 3, UNI>(Db, q, sh.p);
        u_ += sh.omp * m.S1;
        double sg = sh.c0 * m.S2;
        if (MODE == 2) lp += sh.c0 * (sh.t5 * m.S2 - sh.q * m.S3);
        if (bcusp) {
          double v, gf, lpl;
          rad_fn<MODE>(1, bcp, bca, S.rcut_b, sh, v, gf, lpl);
          const double c = hi ? bcc1 : bcc0;
          u_ += c * v;
          if (MODE >= 1) sg += c * gf;
          if (MODE == 2) lp += c * lpl;
        }
        if (MODE >= 1) { gx += sg * dx; gy += sg * dy; gz += sg * dz; }
      }
    }
  }
  for (int I = ions ? j0 : S.natom; I < S.natom; I += dj) {
    double dx = rx - S.atom_xyz[3 * I], dy = ry - S.atom_xyz[3 * I + 1], dz = rz - S.atom_xyz[3 * I + 2];
    const double* q = S.aq + (size_t)(I * 2 + edown) * PQA_JQ;
    const double acc = (ja_on && acusp) ? S.acoeff[(I * S.na) * 2 + edown] : 0.0;
    if (PBC) min_image_j(S, dx, dy, dz);
    double r, ri;
    sqrt_rinv(dx * dx + dy * dy + dz * dz, r, ri);
    if (MODE == 2) sei -= S.atom_charge[I] * ri;
    if (ja_on && r < S.rcut_a) {
      const RadShared sh = rad_shared_ri<MODE>(r, ri, ira);
      const MergedSums m = ka4 ? pade_merged<MODE, 4, UNI>(Da, q, sh.p) : pade_merged<MODE, 3, UNI>(Da, q, sh.p);
      u_ += sh.omp * m.S1;
      double sg = sh.c0 * m.S2;
      if (MODE == 2) lp += sh.c0 * (sh.t5 * m.S2 - sh.q * m.S3);
      if (acusp) {
        double v, gf, lpl;
        rad_fn<MODE>(1, acp, aca, S.rcut_a, sh, v, gf, lpl);
        u_ += acc * v;
        if (MODE >= 1) sg += acc * gf;
        if (MODE == 2) lp += acc * lpl;
      }
      if (MODE >= 1) { gx += sg * dx; gy += sg * dy; gz += sg * dz; }
    }
  }
  U = u_; g[0] = gx; g[1] = gy; g[2] = gz; lapU = lp; ee = see; ei = sei;
}
// PF: partners whose coordinates are requested ahead (4; the kinetic pass of large shards, whose walk comes out of L2, takes 8: 1.68 -> 1.60 ms at
// 65 536 walkers — at the price of 17 registers, a resident wave the small shards need: 195 -> 211 us at 4 096 walkers of C5)
template <int MODE, bool PBC, bool UJ = false, int PF = PQA_JAS_PF>
__device__ __forceinline__ void jas_eval_lane(const SysDev& S, const double* xt, long W, long w, int e,
                                              double rx, double ry, double rz, int has_jastrow, int j0, int dj, double& U,
                                              double (&g)[3], double& lapU, double& ee, double& ei, int skip = -1, bool ions = true) {
  const bool fast = S.nb <= PQA_JAS_NF && S.na <= PQA_JAS_NF;
  if (UJ && S.jq_on) jas_eval_lane_m<MODE, PBC, true, false, PF>(S, xt, W, w, e, rx, ry, rz, has_jastrow, j0, dj, U, g, lapU, ee, ei, skip, ions);
  else if (!UJ && S.jq_on && !fast) jas_eval_lane_m<MODE, PBC, false, false, PF>(S, xt, W, w, e, rx, ry, rz, has_jastrow, j0, dj, U, g, lapU, ee, ei, skip, ions);
  else if (fast) jas_eval_lane_t<MODE, PBC, true, PF>(S, xt, W, w, e, rx, ry, rz, has_jastrow, j0, dj, U, g, lapU, ee, ei, skip, ions);
  else jas_eval_lane_t<MODE, PBC, false, PF>(S, xt, W, w, e, rx, ry, rz, has_jastrow, j0, dj, U, g, lapU, ee, ei, skip, ions);
}
// Jastrow part of group g's partial sums of electron e at (px, py, pz): value and gradient over the partners j = g, g + G, ...
// and the ions I = g, g + G, ...  `skip` (>= 0): the pair with electron `skip` is left out of the strided sum and ADDED LAST by
// the group that owns it (skip mod G) — the old-position sums of the next electron to be proposed leave out the electron that
// has just been decided, the one partner whose position is not known before that decision: everything else can be (and, for
// large shards, is) summed ahead by k_jas_pre while the orbital kernel runs.  `pre` (non-null): that partial, [G][4][W].
template <bool PBC, bool UJ = false>
__device__ __forceinline__ void lw_jastrow_part(const SysDev& S, const LwState& L, int e, int has_jastrow, double px, double py, double pz, long W, long w,
                                                int g, int G, int skip, const double* __restrict__ pre, double& U, double (&gg)[3]) {
  double lp, ee, ei;
  if (pre) {
    U = pre[((size_t)g * 4 + 0) * W + w]; gg[0] = pre[((size_t)g * 4 + 1) * W + w]; gg[1] = pre[((size_t)g * 4 + 2) * W + w]; gg[2] = pre[((size_t)g * 4 + 3) * W + w];
  } else jas_eval_lane<1, PBC, UJ>(S, L.xt, W, w, e, px, py, pz, has_jastrow, g, G, U, gg, lp, ee, ei, skip);
  if (skip >= 0 && skip % G == g) {  // the one pair, at the partner's settled position
    double u1, g1[3];
    jas_eval_lane<1, PBC, UJ>(S, L.xt, W, w, e, px, py, pz, has_jastrow, skip, S.nelec, u1, g1, lp, ee, ei, -1, false);
    U += u1; gg[0] += g1[0]; gg[1] += g1[1]; gg[2] += g1[2];
  }
}

// ---------------------------------------------------------------- move kernels
// Every per-walker loop (orbital slots for the Slater ratios, other electrons and ions for the Jastrow) is
// split over G thread groups (grid.y) so that even W/64 < #SIMDs fills the chip and no thread walks a long
// chain of dependent loads; partial sums land in part[G][8][W] and a finish kernel (thread = walker) adds
// them in group order (deterministic) and does the per-walker scalar work.
//   part rows: 0..3 Slater sums (value, d/dx, d/dy, d/dz), 4 U, 5..7 grad U

// pos: proposal [W][3] (accept) or NULL = current position of e from xt (propose)
// rows: [W][5][nmo] orbital rows at `pos` (accept) or NULL = cached rows ct (propose)
#define PQA_LW_PART_ROWS(CX) ((CX) ? 12 : 8)
// (Forcing 6 or 8 waves per SIMD spills — 76 / 120 us against 66 — and twice the groups at the same occupancy changes nothing:
// the kernel moves ~4 KB per walker at ~4 TB/s.)
// Share of thread group g (of G) in the sums of electron e of walker w at (px, py, pz): Slater sums of the orbital row
// `row` ([5][nmo], point-major: the proposal's row or the cached one) against the inverse row, Jastrow sums against the
// walker's coordinates; p[] in the row order above.
template <bool PBC, bool CX, bool UJ = false>
__device__ __forceinline__ void lw_move_sums(const SysDev& S, const LwState& L, int e, int has_jastrow, double px, double py, double pz,
                                             const double* row, long W, long w, int g, int G,
                                             double (&p)[PQA_LW_PART_ROWS(CX)], int jskip = -1, const double* __restrict__ jpre = nullptr) {
  const int s = e >= S.nup, i = e - s * S.nup, n = s ? S.ndn : S.nup, nmo = S.nmo[s];
  constexpr int CF = CX ? 2 : 1;
  double r0 = 0.0, r1 = 0.0, r2 = 0.0, r3 = 0.0, q0 = 0.0, q1 = 0.0, q2 = 0.0, q3 = 0.0;  // q: imaginary parts (CX)
#ifndef PQA_MP_NOSLATER
  {
    const double* Ti = L.Tt[s] + (size_t)i * n * CF * W + w;
    const int* occ = S.det_occ[s];
    const int nh = nmo / CF;  // orbitals
    // group g takes a CONTIGUOUS range of orbital slots: the rows are point-major (1280 B per walker at 32 orbitals), so a
    // thread's consecutive slots share 64-byte lines
    const int nj = (n + G - 1) / G, jb = g * nj, je = (jb + nj < n) ? jb + nj : n;
    if (!CX && S.occ_ident[s] && ((jb | (je - jb) | nmo) & 7) == 0) {
      // ground-state occupation, 8 slots at a time: a lane's slice of a component row is ONE 64-byte line, fetched by two
      // adjacent 32-byte loads and used up at once (64 lanes sit on 64 different rows: nothing is shared between lanes, and a
      // half-used line does not survive in L1 until the next slots come round).  Same operations in the same order per sum.
      for (int j = jb; j < je; j += 8) {
        double t[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) t[u] = Ti[(size_t)(j + u) * W];
#define PQA_ROW8(C, ACC) do { const double4 lo = *reinterpret_cast<const double4*>(row + (C) * nmo + j), hi = *reinterpret_cast<const double4*>(row + (C) * nmo + j + 4); \
          ACC += lo.x * t[0]; ACC += lo.y * t[1]; ACC += lo.z * t[2]; ACC += lo.w * t[3]; ACC += hi.x * t[4]; ACC += hi.y * t[5]; ACC += hi.z * t[6]; ACC += hi.w * t[7]; } while (0)
        PQA_ROW8(0, r0); PQA_ROW8(1, r1); PQA_ROW8(2, r2); PQA_ROW8(3, r3);
#undef PQA_ROW8
      }
    } else if (!CX && S.occ_ident[s] && ((jb | (je - jb) | nmo) & 3) == 0) {
      // ... 4 slots (32 bytes per load) where the slices are shorter
      for (int j = jb; j < je; j += 4) {
        const double4 a0 = *reinterpret_cast<const double4*>(row + j), a1 = *reinterpret_cast<const double4*>(row + nmo + j);
        const double4 a2 = *reinterpret_cast<const double4*>(row + 2 * nmo + j), a3 = *reinterpret_cast<const double4*>(row + 3 * nmo + j);
        const double t0 = Ti[(size_t)j * W], t1 = Ti[(size_t)(j + 1) * W], t2 = Ti[(size_t)(j + 2) * W], t3 = Ti[(size_t)(j + 3) * W];
        r0 += a0.x * t0; r1 += a1.x * t0; r2 += a2.x * t0; r3 += a3.x * t0;
        r0 += a0.y * t1; r1 += a1.y * t1; r2 += a2.y * t1; r3 += a3.y * t1;
        r0 += a0.z * t2; r1 += a1.z * t2; r2 += a2.z * t2; r3 += a3.z * t2;
        r0 += a0.w * t3; r1 += a1.w * t3; r2 += a2.w * t3; r3 += a3.w * t3;
      }
    } else if (CX && S.occ_ident[s] && ((jb | (je - jb) | nh) & 3) == 0) {
      // complex rows, 4 slots (32 bytes of the real and of the imaginary block per component) at a time: element by element every
      // 8-byte load of the wave is 64 line requests (see k_kinetic_lw)
      for (int j = jb; j < je; j += 4) {
        double tr[4], ti[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) { tr[u] = Ti[(size_t)(2 * (j + u)) * W]; ti[u] = Ti[(size_t)(2 * (j + u) + 1) * W]; }
        const double4 a0 = *reinterpret_cast<const double4*>(row + j), b0 = *reinterpret_cast<const double4*>(row + nh + j);
        const double4 a1 = *reinterpret_cast<const double4*>(row + nmo + j), b1 = *reinterpret_cast<const double4*>(row + nmo + nh + j);
        const double4 a2 = *reinterpret_cast<const double4*>(row + 2 * nmo + j), b2 = *reinterpret_cast<const double4*>(row + 2 * nmo + nh + j);
        const double4 a3 = *reinterpret_cast<const double4*>(row + 3 * nmo + j), b3 = *reinterpret_cast<const double4*>(row + 3 * nmo + nh + j);
#define PQA_CXS(U, X) do { r0 += a0.X * tr[U] - b0.X * ti[U]; q0 += a0.X * ti[U] + b0.X * tr[U]; r1 += a1.X * tr[U] - b1.X * ti[U]; q1 += a1.X * ti[U] + b1.X * tr[U]; \
          r2 += a2.X * tr[U] - b2.X * ti[U]; q2 += a2.X * ti[U] + b2.X * tr[U]; r3 += a3.X * tr[U] - b3.X * ti[U]; q3 += a3.X * ti[U] + b3.X * tr[U]; } while (0)
        PQA_CXS(0, x); PQA_CXS(1, y); PQA_CXS(2, z); PQA_CXS(3, w);
#undef PQA_CXS
      }
    } else {
#pragma unroll 4
      for (int j = jb; j < je; ++j) {
        const int o = occ[j];
        if (CX) {
          const double tr = Ti[(size_t)(2 * j) * W], ti = Ti[(size_t)(2 * j + 1) * W];
          const double a0 = row[o], b0 = row[nh + o], a1 = row[nmo + o], b1 = row[nmo + nh + o];
          const double a2 = row[2 * nmo + o], b2 = row[2 * nmo + nh + o], a3 = row[3 * nmo + o], b3 = row[3 * nmo + nh + o];
          r0 += a0 * tr - b0 * ti; q0 += a0 * ti + b0 * tr; r1 += a1 * tr - b1 * ti; q1 += a1 * ti + b1 * tr;
          r2 += a2 * tr - b2 * ti; q2 += a2 * ti + b2 * tr; r3 += a3 * tr - b3 * ti; q3 += a3 * ti + b3 * tr;
        } else {
          const double t = Ti[(size_t)j * W];
          r0 += row[o] * t; r1 += row[nmo + o] * t; r2 += row[2 * nmo + o] * t; r3 += row[3 * nmo + o] * t;
        }
      }
    }
  }
#endif
  double U = 0.0, gg[3] = {0.0, 0.0, 0.0};
#ifndef PQA_MP_NOJAS
  lw_jastrow_part<PBC, UJ>(S, L, e, has_jastrow, px, py, pz, W, w, g, G, jskip, jpre, U, gg);
#endif
  if (CX) {  // rows: Re r0, Im r0, Re r1, Im r1, ..., then U, grad U
    p[0] = r0; p[1] = q0; p[2] = r1; p[3] = q1; p[4] = r2; p[5] = q2; p[6] = r3; p[7] = q3;
    p[PQA_LW_PART_ROWS(CX) - 4] = U; p[PQA_LW_PART_ROWS(CX) - 3] = gg[0]; p[PQA_LW_PART_ROWS(CX) - 2] = gg[1]; p[PQA_LW_PART_ROWS(CX) - 1] = gg[2];
  } else {
    p[0] = r0; p[1] = r1; p[2] = r2; p[3] = r3; p[4] = U; p[5] = gg[0]; p[6] = gg[1]; p[7] = gg[2];
  }
}
// Slater part of the summed partials -> gradient of log|Psi_S| (real part for complex orbitals), determinant ratio (re, im)
template <bool CX, int PR>
__device__ __forceinline__ void lw_slater_terms(const double (&v)[PR], double& gx, double& gy, double& gz, double& dr, double& di) {
  if (CX) {
    dr = v[0]; di = v[1];
    const double d = 1.0 / (dr * dr + di * di);  // Re(r_c / r_0) = Re(r_c conj r_0) / |r_0|^2
    gx = finite_or((v[2] * dr + v[3] * di) * d, 0.0); gy = finite_or((v[4] * dr + v[5] * di) * d, 0.0); gz = finite_or((v[6] * dr + v[7] * di) * d, 0.0);
  } else {
    dr = v[0]; di = 0.0;
    gx = finite_or(v[1] / v[0], 0.0); gy = finite_or(v[2] / v[0], 0.0); gz = finite_or(v[3] / v[0], 0.0);
  }
}

// ---------------------------------------------------------------- Sherman-Morrison (slater.py:88-94), blocked
// Electrons of one spin are moved in index order, so a ratio or drift only ever needs the inverse rows of electrons that
// have not moved yet in this sweep plus the current one.  The electrons are grouped in blocks of KB; an accepted move of
// electron i updates immediately only the KB rows of its block (in k_step_lw)
//   T[j][k] -= R[k] * (V . T[j])   (j != i),     T[i][k] = R[k],     R = T_old[i]/ratio, V = new orbital row
// and leaves (V, R) in the block buffers Vb/Rb[q][n][W] (q = position in the block, act[q][W] = accepted).  After the last
// electron of a block k_flush_lw applies the block's accepted updates, in order, to every row outside the block while that
// row sits in registers.  Per row the arithmetic and its order are exactly those of updating after every move, so the
// inverse is bitwise identical — but it crosses HBM once per block instead of once per move.
// rows outside [j_lo, j_hi) of spin s: apply the block's nq buffered updates in order.  Vb/Rb: [KB][n][W], act: [KB][W].
// Block = 16 walkers x 16 row groups: the block's update vectors (V_q, R_q of its 16 walkers, 2 nq n 16 doubles) are staged
// in LDS once and shared by the row groups — read per row from global memory they were 4x the traffic of the inverse
// itself (1.26 ms per flush at 65 536 walkers).  A row's arithmetic and its order are unchanged (bitwise identical).
// 16 consecutive walkers are 128 contiguous bytes of every (row, column) plane: two full cache lines per access.
#ifndef PQA_FLUSH_WB
#define PQA_FLUSH_WB 16
#endif
// PQA_ROWDOT — the order of a row's dot product V . T[j] in EVERY lane-per-walker kernel that updates rows (k_flush_lw, the commit
// halves of k_step_lw and k_step_pre): four partial sums over the quarters [q NMAX/4, (q + 1) NMAX/4) of the row's NMAX-padded
// columns, each in ascending k, combined as ((p0 + p1) + p2) + p3 (complex rows: quarters of the complex columns, real and imaginary
// part alike).  One convention everywhere keeps the inverse independent of the block size KB bit for bit, and it is what lets the
// small-shard kernel give a block row to FOUR threads, a quarter each (k_step_pre, GW > 16), instead of one thread walking 32 columns
// while 24 of the 32 thread groups wait.  (Until the last session of round 4 the dot was one running sum over k.)
// WB: walkers per block (16 rows groups at 16, 32 at 8: small walker counts get twice the blocks and one row per thread —
// at 4 096 walkers a flush was 256 blocks of two-row threads, 40 us for 17 us of traffic)
template <int NMAX, bool CX = false, int WB = PQA_FLUSH_WB>
static __global__ __launch_bounds__(256) void k_flush_lw(SysDev S, LwState L, int s, const double* __restrict__ Vb,
                                                  const double* __restrict__ Rb, const uint8_t* __restrict__ act, long W,
                                                  long wlo, long whi, int j_lo, int j_hi, int nq) {
  extern __shared__ double sh[];
  const int n = s ? S.ndn : S.nup;
  const int L_ = CX ? 2 * n : n;  // doubles per row
  double* shV = sh;
  double* shR = sh + (size_t)nq * L_ * WB;
  const int wl = threadIdx.x & (WB - 1), g = threadIdx.x / WB;
  const long w0 = wlo + (long)blockIdx.x * WB;  // walkers [wlo, whi) of the shard; W is the plane stride
  // eight elements of each vector per pass, all sixteen loads in flight before the first LDS store (one element per iteration was
  // a round trip per iteration: 8-10 of them at the head of a ~30-us launch for small shards)
  for (int base = threadIdx.x; base < nq * L_ * WB; base += 8 * 256) {
    double v8[8], r8[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const int idx = base + u * 256;
      const int ic = (idx < nq * L_ * WB) ? idx : base;
      const long ws = (w0 + (ic & (WB - 1)) < whi) ? w0 + (ic & (WB - 1)) : whi - 1;
      const size_t src = (size_t)(ic / WB) * W + ws;  // idx / WB = q * L_ + k
      v8[u] = Vb[src]; r8[u] = Rb[src];
    }
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const int idx = base + u * 256;
      if (idx < nq * L_ * WB) { shV[idx] = v8[u]; shR[idx] = r8[u]; }
    }
  }
  __syncthreads();
  const long w = w0 + wl;
  if (w >= whi) return;
  unsigned mask = 0;
  for (int q = 0; q < nq; ++q) mask |= act[(size_t)q * W + w] ? (1u << q) : 0u;
  if (!mask) return;
  double* T = L.Tt[s] + w;
  const int nout = n - (j_hi - j_lo);
  for (int jj = g; jj < nout; jj += 256 / WB) {
    const int j = (jj < j_lo) ? jj : jj + (j_hi - j_lo);
    double* Tj = T + (size_t)j * L_ * W;
    double t[NMAX];
#pragma unroll
    for (int k = 0; k < NMAX; ++k) t[k] = (k < L_) ? Tj[(size_t)k * W] : 0.0;
    for (int q = 0; q < nq; ++q) {
      if (!((mask >> q) & 1u)) continue;
      const double* Vq = shV + (size_t)q * L_ * WB + wl;
      const double* Rq = shR + (size_t)q * L_ * WB + wl;
      if (CX) {
        double pr[4] = {0.0, 0.0, 0.0, 0.0}, pi[4] = {0.0, 0.0, 0.0, 0.0};  // row dot in four partial sums: PQA_ROWDOT below
#pragma unroll
        for (int k = 0; k < NMAX / 2; ++k)
          if (2 * k < L_) {
            const double vr = Vq[(2 * k) * WB], vi = Vq[(2 * k + 1) * WB];
            pr[k / (NMAX / 8)] += vr * t[2 * k] - vi * t[2 * k + 1];
            pi[k / (NMAX / 8)] += vr * t[2 * k + 1] + vi * t[2 * k];
          }
        const double tmp = ((pr[0] + pr[1]) + pr[2]) + pr[3], tmi = ((pi[0] + pi[1]) + pi[2]) + pi[3];
#pragma unroll
        for (int k = 0; k < NMAX / 2; ++k)
          if (2 * k < L_) {
            const double rr = Rq[(2 * k) * WB], ri = Rq[(2 * k + 1) * WB];
            t[2 * k] -= rr * tmp - ri * tmi;
            t[2 * k + 1] -= rr * tmi + ri * tmp;
          }
      } else {
        double p4[4] = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
        for (int k = 0; k < NMAX; ++k)
          if (k < L_) p4[k / (NMAX / 4)] += Vq[k * WB] * t[k];
        const double tmp = ((p4[0] + p4[1]) + p4[2]) + p4[3];
#pragma unroll
        for (int k = 0; k < NMAX; ++k)
          if (k < L_) t[k] = t[k] - Rq[k * WB] * tmp;
      }
    }
#pragma unroll
    for (int k = 0; k < NMAX; ++k)
      if (k < L_) Tj[(size_t)k * W] = t[k];
  }
}

// ---------------------------------------------------------------- one launch per move (round 3)
// k_step_lw = [decide electron e_acc: new-position sums, Metropolis test, Sherman-Morrison commit of the block rows, cache row,
// coordinate]  followed by  [propose electron e_prop: old-position sums, drift, proposal] — what round 2 did in six launches
// per move (two partial-sum launches, two thread-per-walker finish kernels, the commit) is ONE launch between two orbital
// evaluations.  Block = NW walkers x G thread groups: the groups' partial sums meet in LDS (added in group order, exactly as
// the finish kernels added the part[G] planes: bit-identical trajectories, checked against the six-launch build before it was
// removed), every group repeats the per-walker scalar work (decision, proposal) and group 0 stores
// it; the accepted walkers' update vectors V, R sit in LDS for the commit and go to the block buffers for the flush.  What a
// move streams through HBM: the walker's coordinates once (were 2x + L2 re-reads by the finish kernels), no part[] planes, no
// auxiliary round trips — and 2 dependent launches per move instead of 6 (the chain that sets the step time of small shards).
// Either half can be switched off (e_acc < 0 / e_prop < 0): at a Sherman-Morrison block boundary the flush has to run
// between the two halves.
struct StepArgs {
  int e_acc, e_prop;   // electron to decide / to propose (-1: skip that half)
  int has_jastrow, G, NW;  // thread groups per walker, walkers per block (NW * G <= 256)
  int j_lo, j_hi;      // Sherman-Morrison block of e_acc (rows of its spin)
  long W;              // walkers of the shard = stride of every plane
  long w0, w1;         // this launch covers walkers [w0, w1) (the whole shard, or one half-ensemble of the pipelined sweep)
  int j_skip;          // propose half: electron decided just before e_prop (its pair is summed last, lw_jastrow_part), or -1
  const double* jnew;  // Jastrow partial sums summed ahead by k_jas_pre, [G][4][W]: at the proposal of e_acc / at the current
  const double* jold;  //   position of e_prop without the pair j_skip (nullptr: summed here)
  double* Rbuf;        // block buffers, slot of e_acc: [L_][W]
  double* Vbuf;
  uint8_t* act;        // [W]
};

// Block = NW walkers x G groups, <= 256 threads.  WIDE: NW = 64, a wave is one group (g wave-uniform: table look-ups stay scalar
// loads).  Otherwise NW = 16 or 32 and a wave holds 4 or 2 groups of the same walkers: small shards then spread over 4x / 2x
// as many blocks — with 64 walkers per block 4096 walkers are 64 blocks on 64 of the 256 CUs, each issuing all 16 groups' work.
#ifndef PQA_STEP_MINW
#define PQA_STEP_MINW 0  // > 0: minimum waves per SIMD the register allocation of k_step_lw must allow (A/B: tools/scratch/r3_ab_simple.sh)
#endif
#if PQA_STEP_MINW > 0
#define PQA_STEP_BOUNDS __launch_bounds__(256, PQA_STEP_MINW)
#else
#define PQA_STEP_BOUNDS __launch_bounds__(256)
#endif
template <bool PBC, bool CX, int NMAX, bool WIDE>
static __global__ PQA_STEP_BOUNDS void k_step_lw(SysDev S, LwState L, MoveBuf mb, StepArgs a) {
  extern __shared__ double sh[];
  constexpr int PR = PQA_LW_PART_ROWS(CX), JU = CX ? 8 : 4, CF = CX ? 2 : 1;
  const int NW = WIDE ? 64 : a.NW;
  const int lane = WIDE ? (int)(threadIdx.x & 63) : (int)threadIdx.x % NW;  // walker within the block
  const int g = WIDE ? __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)) : (int)threadIdx.x / NW;
  const int G = a.G;
  const long W = a.W;
  const long wr = a.w0 + (long)blockIdx.x * NW + lane;
  const bool live = wr < a.w1;
  const long w = live ? wr : a.w1 - 1;  // lanes past the end shadow the last walker and store nothing
  const bool lead = live && g == 0;
  if (a.e_acc >= 0) {
    const int e = a.e_acc;
    const int s = e >= S.nup, i = e - s * S.nup, n = s ? S.ndn : S.nup, nmo = S.nmo[s];
    const int L_ = CF * n;  // doubles per inverse row
    // the proposal's orbital row: the orbital kernel wrote it into the slot the walker is NOT using
    const int cur = L.sel[s][(size_t)i * W + w];
    const double* row = lw_row(L, s, i, cur ^ 1, w, W, nmo);
    double v[PR];
    {
      double p[PR];
      lw_move_sums<PBC, CX, WIDE>(S, L, e, a.has_jastrow, mb.newpos[3 * w], mb.newpos[3 * w + 1], mb.newpos[3 * w + 2], row, W, w, g, G, p, -1, a.jnew);
#pragma unroll
      for (int c = 0; c < PR; ++c) sh[(c * G + g) * NW + lane] = p[c];
    }
    __syncthreads();
#pragma unroll
    for (int c = 0; c < PR; ++c) v[c] = 0.0;
    for (int gg = 0; gg < G; ++gg) {
#pragma unroll
      for (int c = 0; c < PR; ++c) v[c] += sh[(c * G + gg) * NW + lane];
    }
    // ---- Metropolis decision (mc.py:124-132; dmc.py:57-70), every group the same numbers
    double gx, gy, gz, dr, di;
    lw_slater_terms<CX, PR>(v, gx, gy, gz, dr, di);
    gx += v[JU + 1]; gy += v[JU + 2]; gz += v[JU + 3];
    const double* ax = L.auxt + w;
    double val2;  // |Psi(new)/Psi|^2 (mc.py:131)
    if (CX) val2 = finite_or(dr * dr + di * di, 1.0);
    else { const double val = finite_or(dr, 1.0); val2 = val * val; }
    if (a.has_jastrow) { const double ej = exp(v[JU] - ax[6 * W]); val2 *= ej * ej; }
    const double a0 = ax[0], a1 = ax[W], a2 = ax[2 * W], d0 = ax[3 * W], d1 = ax[4 * W], d2 = ax[5 * W];
    const double fwd = a0 * a0 + a1 * a1 + a2 * a2;
    double bx, by, bz;
    if (mb.dmc) {  // dmc.py:57-60: backward = gauss + drift(old) + drift(new)
      limdrift_dmc(gx, gy, gz, mb.tstep);
      bx = a0 + d0 + gx; by = a1 + d1 + gy; bz = a2 + d2 + gz;
    } else {
      limdrift3(gx, gy, gz);
      bx = a0 + mb.tstep * (d0 + gx); by = a1 + mb.tstep * (d1 + gy); bz = a2 + mb.tstep * (d2 + gz);
    }
    const double bwd = bx * bx + by * by + bz * bz;
    const double t_prob = exp(1.0 / (2.0 * mb.tstep) * (fwd - bwd));
    double ratio = val2 * t_prob;
    if (mb.dmc && !CX) {
      const double dv = finite_or(dr, 1.0);  // the Jastrow ratio is positive: np.sign(psi_ratio) is the determinant's
      ratio *= (dv > 0.0) ? 1.0 : ((dv < 0.0) ? -1.0 : 0.0);  // fixed node (dmc.py:64-66)
    }
    double u;
    if (mb.unif) u = mb.unif[(size_t)e * W + w];
    else {
      const Philox ph = philox(mb.seed, (uint32_t)w, (uint32_t)e, PQA_STREAM_ACCEPT, mb.step);
      u = u01(ph.c[0], ph.c[1]);
    }
    const bool acc = ratio > u;
    if (lead) {
      if (mb.dmc) {  // dmc.py:68 r2 = |gauss + drift|^2
        const double rx = a0 + d0, ry = a1 + d1, rz = a2 + d2;
        const double r2 = rx * rx + ry * ry + rz * rz;
        mb.r2_prop[w] += r2;
        if (acc) mb.r2_acc[w] += r2;
      }
      a.act[w] = acc;
      if (mb.accept_rec) mb.accept_rec[(size_t)e * W + w] = acc;
      if (acc) {
        mb.acc_w[w] += 1;
        L.sel[s][(size_t)i * W + w] = (uint8_t)(cur ^ 1);  // the proposal's row becomes the cached row
        double* xe = L.xt + (size_t)e * 3 * W + w;
        xe[0] = mb.newpos[3 * w]; xe[W] = mb.newpos[3 * w + 1]; xe[2 * W] = mb.newpos[3 * w + 2];
        if (mb.wrap) {
          int* wp = mb.wrap + ((size_t)w * S.nelec + e) * 3;
          wp[0] += mb.dwrap[3 * w]; wp[1] += mb.dwrap[3 * w + 1]; wp[2] += mb.dwrap[3 * w + 2];
        }
        if (CX) {  // dsign *= ratio / |ratio|, dlog += log|ratio|
          const double m2 = dr * dr + di * di, m = sqrt(m2), ur = dr / m, ui = di / m;
          double* ds = L.dsign[s] + 2 * w;
          const double sr = ds[0], si = ds[1];
          ds[0] = sr * ur - si * ui; ds[1] = sr * ui + si * ur;
          L.dlog[s][w] += 0.5 * log(m2);
        } else {
          L.dsign[s][w] *= (dr > 0.0) ? 1.0 : ((dr < 0.0) ? -1.0 : dr);
          L.dlog[s][w] += log(fabs(dr));
        }
      }
    }
    __syncthreads();  // the partial sums have been read: their LDS becomes the update vectors
    // ---- V = new orbital row on the occupied slots, R = T_old[i] / ratio (slater.py:88-94); each group its slot range
    double* shV = sh;
    double* shR = sh + (size_t)L_ * NW;
    {
      const int* occ = S.det_occ[s];
      const int nh = nmo / CF;
      const double* Ti = L.Tt[s] + (size_t)i * L_ * W + w;
      const int nj = (n + G - 1) / G, jb = g * nj, je = (jb + nj < n) ? jb + nj : n;
      const bool st = acc && live;
      if (CX) {
        const double m2 = dr * dr + di * di;
        const double ir = dr / m2, ii = -di / m2;  // 1 / ratio
        for (int k = jb; k < je; ++k) {
          const double vr = acc ? row[occ[k]] : 0.0, vi = acc ? row[nh + occ[k]] : 0.0;
          const double tr = Ti[(size_t)(2 * k) * W], ti = Ti[(size_t)(2 * k + 1) * W];
          const double rr = acc ? tr * ir - ti * ii : 0.0, ri = acc ? tr * ii + ti * ir : 0.0;
          shV[(2 * k) * NW + lane] = vr; shV[(2 * k + 1) * NW + lane] = vi;
          shR[(2 * k) * NW + lane] = rr; shR[(2 * k + 1) * NW + lane] = ri;
          if (st) {
            a.Vbuf[(size_t)(2 * k) * W + w] = vr; a.Vbuf[(size_t)(2 * k + 1) * W + w] = vi;
            a.Rbuf[(size_t)(2 * k) * W + w] = rr; a.Rbuf[(size_t)(2 * k + 1) * W + w] = ri;
          }
        }
      } else {
        const double inv = 1.0 / dr;
        for (int k = jb; k < je; ++k) {
          const double vv = acc ? row[occ[k]] : 0.0;
          const double rr = acc ? Ti[(size_t)k * W] * inv : 0.0;
          shV[k * NW + lane] = vv;
          shR[k * NW + lane] = rr;
          if (st) { a.Vbuf[(size_t)k * W + w] = vv; a.Rbuf[(size_t)k * W + w] = rr; }
        }
      }
    }
    __syncthreads();
    // ---- rows of the electron block (k_flush_lw brings the others up to date when the block ends).  Rejected walkers take
    // part in the loads and stores (writing back what they read): a cache line holds 8 walkers, so it is fetched and written
    // whenever one of them accepted anyway, and with every lane storing the wave writes whole lines.
#ifndef PQA_ST_NOCOMMIT
    if (__any(acc)) {
      double* T = L.Tt[s] + w;
      for (int j = a.j_lo + g; j < a.j_hi; j += G) {
        double* Tj = T + (size_t)j * L_ * W;
        if (j == i) {
          if (acc && live) {
#pragma unroll
            for (int k = 0; k < NMAX; ++k)
              if (k < L_) Tj[(size_t)k * W] = shR[k * NW + lane];
          }
          continue;
        }
        double t[NMAX];
#pragma unroll
        for (int k = 0; k < NMAX; ++k) t[k] = (k < L_) ? Tj[(size_t)k * W] : 0.0;
        if (CX) {
          double pr[4] = {0.0, 0.0, 0.0, 0.0}, pi[4] = {0.0, 0.0, 0.0, 0.0};  // PQA_ROWDOT
#pragma unroll
          for (int k = 0; k < NMAX / 2; ++k)
            if (2 * k < L_) {  // tmp = sum_k V_k t_k (complex, no conjugation)
              const double vr = shV[(2 * k) * NW + lane], vi = shV[(2 * k + 1) * NW + lane];
              pr[k / (NMAX / 8)] += vr * t[2 * k] - vi * t[2 * k + 1];
              pi[k / (NMAX / 8)] += vr * t[2 * k + 1] + vi * t[2 * k];
            }
          const double tmp = ((pr[0] + pr[1]) + pr[2]) + pr[3], tmi = ((pi[0] + pi[1]) + pi[2]) + pi[3];
          if (live) {
#pragma unroll
            for (int k = 0; k < NMAX / 2; ++k)
              if (2 * k < L_) {
                const double rr = shR[(2 * k) * NW + lane], ri = shR[(2 * k + 1) * NW + lane];
                const double ur = rr * tmp - ri * tmi, ui = rr * tmi + ri * tmp;
                Tj[(size_t)(2 * k) * W] = acc ? t[2 * k] - ur : t[2 * k];
                Tj[(size_t)(2 * k + 1) * W] = acc ? t[2 * k + 1] - ui : t[2 * k + 1];
              }
          }
        } else {
          double p4[4] = {0.0, 0.0, 0.0, 0.0};  // PQA_ROWDOT
#pragma unroll
          for (int k = 0; k < NMAX; ++k)
            if (k < L_) p4[k / (NMAX / 4)] += shV[k * NW + lane] * t[k];
          const double tmp = ((p4[0] + p4[1]) + p4[2]) + p4[3];
          if (live) {
#pragma unroll
            for (int k = 0; k < NMAX; ++k)
              if (k < L_) Tj[(size_t)k * W] = acc ? t[k] - shR[k * NW + lane] * tmp : t[k];
          }
        }
      }
    }
#endif
    __syncthreads();  // inverse rows, coordinate: visible to the other groups of this walker block
  }
  if (a.e_prop >= 0) {
    // ---- drift at the current position, proposal r' = r + sqrt(tau) z + tau limdrift(grad)   (mc.py:117-121)
    const int e = a.e_prop;
    double v[PR];
    {
      double p[PR];
      const int s = e >= S.nup, i = e - s * S.nup;
      const double* xe = L.xt + (size_t)e * 3 * W + w;
      const double* row = lw_row(L, s, i, L.sel[s][(size_t)i * W + w], w, W, S.nmo[s]);  // cached row of the current position
      lw_move_sums<PBC, CX, WIDE>(S, L, e, a.has_jastrow, xe[0], xe[W], xe[2 * W], row, W, w, g, G, p, a.j_skip, a.jold);
#pragma unroll
      for (int c = 0; c < PR; ++c) sh[(c * G + g) * NW + lane] = p[c];
    }
    __syncthreads();
    if (!lead) return;
#pragma unroll
    for (int c = 0; c < PR; ++c) v[c] = 0.0;
    for (int gg = 0; gg < G; ++gg) {
#pragma unroll
      for (int c = 0; c < PR; ++c) v[c] += sh[(c * G + gg) * NW + lane];
    }
    double gx, gy, gz, dr, di;
    lw_slater_terms<CX, PR>(v, gx, gy, gz, dr, di);
    gx += v[JU + 1]; gy += v[JU + 2]; gz += v[JU + 3];
    if (mb.dmc) limdrift_dmc(gx, gy, gz, mb.tstep);  // the drift vector itself (dmc.py:50-52)
    else limdrift3(gx, gy, gz);
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
    const double* xe = L.xt + (size_t)e * 3 * W + w;
    double* np_ = mb.newpos + 3 * w;
    const double df = mb.dmc ? 1.0 : mb.tstep;
    np_[0] = xe[0] + z0 + gx * df;
    np_[1] = xe[W] + z1 + gy * df;
    np_[2] = xe[2 * W] + z2 + gz * df;
    if (mb.dwrap) fold_cell(S, np_[0], np_[1], np_[2], mb.dwrap + 3 * w);  // make_irreducible, mc.py:121
    double* ao = L.auxt + w;
    ao[0] = z0; ao[W] = z1; ao[2 * W] = z2; ao[3 * W] = gx; ao[4 * W] = gy; ao[5 * W] = gz; ao[6 * W] = v[JU];
  }
}

// ---------------------------------------------------------------- Jastrow sums of a move, ahead of its orbitals (round 4)
// Of k_step_lw's 117 us per move at 65536 walkers ~48 are the two Jastrow pair loops (value + gradient at the proposal of
// e_acc, and at the current position of e_prop) — fp64 VALU work on 1.5 KB of coordinates per walker that needs nothing from
// the orbital kernel: the proposal is known when the previous step launch ends, and of e_prop's partners only e_acc can still
// move.  k_jas_pre forms exactly the partial sums k_step_lw would (same block geometry, same strided partner lists, the pair
// (e_prop, e_acc) left to k_step_lw: lw_jastrow_part) and is launched on a side stream NEXT TO k_orb, whose waves leave the
// SIMDs' fp64 pipe idle a quarter of the time and 96 registers per SIMD free: the sums cost the move nothing, k_step_lw
// loads 2 x 4 doubles per thread instead.  OPT-IN experiment (PQA_JPRE=1, off by default, DESIGN.md section 4: it lost): the sums here go
// function by function, the wide k_step_lw's own go through the merged Pade route, so the two routes agree to rounding, not bitwise.
template <bool PBC>
static __global__ __launch_bounds__(256) void k_jas_pre(SysDev S, LwState L, MoveBuf mb, StepArgs a, double* __restrict__ jnew, double* __restrict__ jold) {
  const int NW = a.NW, G = a.G;
  const int lane = (int)threadIdx.x % NW, g = (int)threadIdx.x / NW;
  const long W = a.W;
  const long w = a.w0 + (long)blockIdx.x * NW + lane;
  if (w >= a.w1) return;
  double U, gg[3], lp, ee, ei;
  if (a.e_acc >= 0) {
    jas_eval_lane<1, PBC>(S, L.xt, W, w, a.e_acc, mb.newpos[3 * w], mb.newpos[3 * w + 1], mb.newpos[3 * w + 2], a.has_jastrow, g, G, U, gg, lp, ee, ei);
    jnew[((size_t)g * 4 + 0) * W + w] = U; jnew[((size_t)g * 4 + 1) * W + w] = gg[0]; jnew[((size_t)g * 4 + 2) * W + w] = gg[1]; jnew[((size_t)g * 4 + 3) * W + w] = gg[2];
  }
  if (a.e_prop >= 0) {
    const double* xe = L.xt + (size_t)a.e_prop * 3 * W + w;
    jas_eval_lane<1, PBC>(S, L.xt, W, w, a.e_prop, xe[0], xe[W], xe[2 * W], a.has_jastrow, g, G, U, gg, lp, ee, ei, a.j_skip);
    jold[((size_t)g * 4 + 0) * W + w] = U; jold[((size_t)g * 4 + 1) * W + w] = gg[0]; jold[((size_t)g * 4 + 2) * W + w] = gg[1]; jold[((size_t)g * 4 + 3) * W + w] = gg[2];
  }
}

// ---------------------------------------------------------------- k_step_lw for small shards: every load up front
// At 4 096 walkers k_step_lw is 256 blocks of one wave per SIMD and its 30 us are a chain of ~11 dependent memory round trips
// (selector -> row, inverse slice, partner coordinates, ion tables, auxiliaries, the block row of the commit, then the same
// again for the proposal of the next electron) between 5 barriers — the step time of a small shard is 77 of those chains.
// Almost none of the addresses depend on anything computed in the kernel: k_step_pre issues ALL loads of BOTH halves at entry
// (one round trip; the orbital rows, whose slot depends on the selector byte, one more), keeps the partner coordinates of the
// walker in registers across the two halves (the accepted proposal is patched in), takes the update vectors from the slices
// already in registers, and hands the inverse row of the next electron — a block row this launch updates — to the proposal
// half through LDS instead of a store / barrier / load.  Registers are no object here (one wave per SIMD).
// Every sum is formed from the same operands in the same order as in k_step_lw / lw_move_sums / jas_eval_lane_t<.., FAST>; the
// compiler contracts multiply-adds differently in the two inlining contexts, so a group's Jastrow gradient sum can differ in its
// last bit: same decisions, walkers equal to ~1e-14 after a sweep (measured with the -DPQA_PRE_DBG builds, which swap single
// parts back to the k_step_lw routines; tests/test_gpu_fullsize.py::test_prefetching_step_kernel_against_the_general_one).
// Scope (the host falls back to k_step_lw otherwise): real orbitals, ground-state occupation lists, Jastrow tables of at most
// PQA_JAS_NF functions per kind, G >= 8 groups with N <= PQA_PRE_NP G electrons, natom <= PQA_PRE_NA G ions, and a
// Sherman-Morrison block of at most G rows.
#define PQA_PRE_NP 8
#ifndef PQA_PRE_DBG
#define PQA_PRE_DBG 0  // bisecting aid: 1 / 2 decide / propose sums by lw_move_sums, 4 / 8 only their Jastrow part by jas_eval_lane
#endif
#define PQA_PRE_NA 4
struct JasTabs {  // wave-uniform function tables (scalar registers)
  int bk[PQA_JAS_NF], ak[PQA_JAS_NF];
  double bp[PQA_JAS_NF], ba[PQA_JAS_NF], ap[PQA_JAS_NF], aa[PQA_JAS_NF];
};
// jas_eval_lane_t<1, PBC, true> for the partners j = g, g + G, ... and ions I = g, g + G, ... on coordinates and ion
// coefficients that are already in registers
template <bool PBC, int NP, int NA>
__device__ __forceinline__ void jas_pre(const SysDev& S, const JasTabs& J, int e, double rx, double ry, double rz, int has_jastrow, int g,
                                        int G, const double (&pcx)[NP], const double (&pcy)[NP],
                                        const double (&pcz)[NP], const double (&atx)[NA], const double (&aty)[NA],
                                        const double (&atz)[NA], const double (&bc0)[PQA_JAS_NF], const double (&bc1)[PQA_JAS_NF],
                                        const double (&ac)[NA][PQA_JAS_NF], double& U, double (&gr)[3]) {
  constexpr int NF = PQA_JAS_NF;
  const double irb = 1.0 / S.rcut_b, ira = 1.0 / S.rcut_a;
  double u_ = 0.0, gx = 0.0, gy = 0.0, gz = 0.0;
#pragma unroll
  for (int m = 0; m < NP; ++m) {
    const int j = g + m * G;
    if (j >= S.nelec || j == e) continue;
    double dx = rx - pcx[m], dy = ry - pcy[m], dz = rz - pcz[m];
    if (PBC) min_image_j(S, dx, dy, dz);
    const double r = sqrt(dx * dx + dy * dy + dz * dz);
    if (has_jastrow && r < S.rcut_b) {
      const RadShared sh = rad_shared<1>(r, irb);
      const bool hi = j >= S.nup;
      double sg = 0.0;
#pragma unroll
      for (int l = 0; l < NF; ++l) {
        if (l < S.nb) {
          double v, gf, lpl;
          rad_fn<1>(J.bk[l], J.bp[l], J.ba[l], S.rcut_b, sh, v, gf, lpl);
          const double c = hi ? bc1[l] : bc0[l];
          u_ += c * v;
          sg += c * gf;
        }
      }
      gx += sg * dx; gy += sg * dy; gz += sg * dz;
    }
  }
#pragma unroll
  for (int m = 0; m < NA; ++m) {
    const int I = g + m * G;
    if (I >= S.natom) continue;
    double dx = rx - atx[m], dy = ry - aty[m], dz = rz - atz[m];
    if (PBC) min_image_j(S, dx, dy, dz);
    const double r = sqrt(dx * dx + dy * dy + dz * dz);
    if (has_jastrow && r < S.rcut_a) {
      const RadShared sh = rad_shared<1>(r, ira);
      double sg = 0.0;
#pragma unroll
      for (int k = 0; k < NF; ++k) {
        if (k < S.na) {
          double v, gf, lpl;
          rad_fn<1>(J.ak[k], J.ap[k], J.aa[k], S.rcut_a, sh, v, gf, lpl);
          u_ += ac[m][k] * v;
          sg += ac[m][k] * gf;
        }
      }
      gx += sg * dx; gy += sg * dy; gz += sg * dz;
    }
  }
  U = u_; gr[0] = gx; gr[1] = gy; gr[2] = gz;
}

// GW = 16: the kernel as described (256 threads: NW walkers x G >= 8 groups, up to 8 partners and 4 ions per thread).
// GW = 32 / 64 (round 4): 16 walkers x GW groups = 512 / 1024 threads — two / ONE partner electron and ion per thread, so the two
// Jastrow pair loops that made up most of the ~3000-instruction dependent chain of a launch become one or two pair evaluations;
// the groups' partial sums are totalled by eight threads per walker (one per row of the sums, group order) instead of by every thread.
#ifdef PQA_PRE_CLK  // timing build only (tools/scratch/pre_clk.py): 100 MHz stamps of the phases of the first 256 blocks of the LAST launch
static __device__ unsigned long long pqa_pre_clk[256 * 16];
#define PQA_PCLK(k) do { if (blockIdx.x < 256 && threadIdx.x == 0) pqa_pre_clk[blockIdx.x * 16 + (k)] = wall_clock64(); } while (0)
#else
#define PQA_PCLK(k) do { } while (0)
#endif
template <bool PBC, int NMAX, int GW = 16>
static __global__ __launch_bounds__(GW == 16 ? 256 : 16 * GW) void k_step_pre(SysDev S, LwState L, MoveBuf mb, StepArgs a) {
  extern __shared__ double sh[];
  constexpr int PR = 8, JU = 4, NF = PQA_JAS_NF;
  constexpr int NS = GW == 16 ? (NMAX + 7) / 8 : (NMAX + GW - 1) / GW, NP = GW == 16 ? PQA_PRE_NP : 64 / GW, NA = GW == 16 ? PQA_PRE_NA : 64 / GW;
  PQA_PCLK(0);
  const int NW = a.NW, G = a.G;
  const int lane = (int)threadIdx.x % NW, g = (int)threadIdx.x / NW;
  const long W = a.W;
  const long wr = a.w0 + (long)blockIdx.x * NW + lane;
  const bool live = wr < a.w1;
  const long w = live ? wr : a.w1 - 1;  // lanes past the end shadow the last walker and store nothing
  const bool lead = live && g == 0;
  double* shP = sh;                           // [PR][G][NW] partial sums
  double* shV = sh + (size_t)PR * G * NW;     // [NMAX][NW] update vectors
  double* shR = shV + (size_t)NMAX * NW;
  double* shT = shR + (size_t)NMAX * NW;      // [NMAX][NW] inverse row of the next electron after the commit
  double* shTot = shT + (size_t)NMAX * NW;    // [PR][NW] (GW > 16) the totals of the groups' partial sums
  const int ea = a.e_acc, ep = a.e_prop;
  const bool has_a = ea >= 0, has_p = ep >= 0;
  const int s = has_a ? (ea >= S.nup) : 0, i = ea - s * S.nup, n = s ? S.ndn : S.nup, nmo = S.nmo[s];
  const int s2 = has_p ? (ep >= S.nup) : 0, i2 = ep - s2 * S.nup, n2 = s2 ? S.ndn : S.nup, nmo2 = S.nmo[s2];
  const int nj = (n + G - 1) / G, jb = g * nj, je = (jb + nj < n) ? jb + nj : n;
  const int nj2 = (n2 + G - 1) / G, jb2 = g * nj2, je2 = (jb2 + nj2 < n2) ? jb2 + nj2 : n2;
  const bool handoff = has_a && has_p && s2 == s;  // the proposal's inverse row is a block row of this launch's commit
  // ---------------------------------------------------------------- all loads whose addresses are known now
  JasTabs J;
#pragma unroll
  for (int l = 0; l < NF; ++l) {
    J.bk[l] = S.b_kind[l]; J.bp[l] = S.b_param[l]; J.ba[l] = S.b_aux[l];
    J.ak[l] = S.a_kind[l]; J.ap[l] = S.a_param[l]; J.aa[l] = S.a_aux[l];
  }
  double pcx[NP], pcy[NP], pcz[NP], atx[NA], aty[NA], atz[NA];
#pragma unroll
  for (int m = 0; m < NP; ++m) {
    const int j = g + m * G;
    const double* xj = L.xt + (size_t)(j < S.nelec ? j : 0) * 3 * W + w;
    pcx[m] = xj[0]; pcy[m] = xj[W]; pcz[m] = xj[2 * W];
  }
#pragma unroll
  for (int m = 0; m < NA; ++m) {
    const int I = (g + m * G < S.natom) ? g + m * G : 0;
    atx[m] = S.atom_xyz[3 * I]; aty[m] = S.atom_xyz[3 * I + 1]; atz[m] = S.atom_xyz[3 * I + 2];
  }
  double bcA0[NF], bcA1[NF], bcP0[NF], bcP1[NF], acA[NA][NF], acP[NA][NF];
#pragma unroll
  for (int l = 0; l < NF; ++l) {
    bcA0[l] = (a.has_jastrow && l < S.nb) ? S.bcoeff[l * 3 + s] : 0.0;
    bcA1[l] = (a.has_jastrow && l < S.nb) ? S.bcoeff[l * 3 + s + 1] : 0.0;
    bcP0[l] = (a.has_jastrow && l < S.nb) ? S.bcoeff[l * 3 + s2] : 0.0;
    bcP1[l] = (a.has_jastrow && l < S.nb) ? S.bcoeff[l * 3 + s2 + 1] : 0.0;
  }
#pragma unroll
  for (int m = 0; m < NA; ++m) {
    const int I = (g + m * G < S.natom) ? g + m * G : 0;
#pragma unroll
    for (int k = 0; k < NF; ++k) {
      acA[m][k] = (a.has_jastrow && k < S.na) ? S.acoeff[(I * S.na + k) * 2 + s] : 0.0;
      acP[m][k] = (a.has_jastrow && k < S.na) ? S.acoeff[(I * S.na + k) * 2 + s2] : 0.0;
    }
  }
  int cur = 0, cur2 = 0;
  double npx = 0.0, npy = 0.0, npz = 0.0, ax7[7] = {0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0}, uu = 0.0;
  // Block rows.  GW = 16: one row per group (the host checks j_hi - j_lo <= G).  GW > 16: one row per QUARTET of groups, group
  // g holding the quarter g % 4 of the row's NMAX-padded columns (j_hi - j_lo <= G / 4): the row's dot product is the four
  // quarter sums combined in the PQA_ROWDOT order, so the inverse stays bitwise what k_step_lw / k_flush_lw make of it.
  constexpr bool QUART = GW > 16;
  constexpr int QL = NMAX / 4, TBN = QUART ? QL : NMAX;
  const int qd = QUART ? (g & 3) : 0, kq0 = qd * QL;
  double tinv[NS], tb[TBN], tinv2[NS], xe2[3] = {0.0, 0.0, 0.0}, zt[3] = {0.0, 0.0, 0.0};
  const int jrow = a.j_lo + (QUART ? (g >> 2) : g);
  const bool has_row = has_a && jrow < a.j_hi;
  if (has_a) {
    cur = L.sel[s][(size_t)i * W + w];
    npx = mb.newpos[3 * w]; npy = mb.newpos[3 * w + 1]; npz = mb.newpos[3 * w + 2];
    const double* ax = L.auxt + w;
#pragma unroll
    for (int q = 0; q < 7; ++q) ax7[q] = ax[(size_t)q * W];
    if (mb.unif) uu = mb.unif[(size_t)ea * W + w];
    const double* Ti = L.Tt[s] + (size_t)i * n * W + w;
#pragma unroll
    for (int u = 0; u < NS; ++u) tinv[u] = (jb + u < je) ? Ti[(size_t)(jb + u) * W] : 0.0;
    if (has_row) {
      const double* Tj = L.Tt[s] + (size_t)jrow * n * W + w;
#pragma unroll
      for (int u = 0; u < TBN; ++u) tb[u] = (kq0 + u < n) ? Tj[(size_t)(kq0 + u) * W] : 0.0;
    }
  }
  if (has_p) {
    cur2 = L.sel[s2][(size_t)i2 * W + w];
    const double* xe = L.xt + (size_t)ep * 3 * W + w;
    xe2[0] = xe[0]; xe2[1] = xe[W]; xe2[2] = xe[2 * W];
    if (mb.gauss) { const double* z = mb.gauss + ((size_t)ep * W + w) * 3; zt[0] = z[0]; zt[1] = z[1]; zt[2] = z[2]; }
    if (!handoff) {
      const double* Ti2 = L.Tt[s2] + (size_t)i2 * n2 * W + w;
#pragma unroll
      for (int u = 0; u < NS; ++u) tinv2[u] = (jb2 + u < je2) ? Ti2[(size_t)(jb2 + u) * W] : 0.0;
    }
  }
  // ---------------------------------------------------------------- second round trip: the orbital rows (slot = selector byte)
  double rv[4][NS], rv2[4][NS];
  if (has_a) {
    const double* row = lw_row(L, s, i, cur ^ 1, w, W, nmo);  // the proposal's row: the slot the walker is NOT using
#pragma unroll
    for (int c = 0; c < 4; ++c)
#pragma unroll
      for (int u = 0; u < NS; ++u) rv[c][u] = (jb + u < je) ? row[c * nmo + jb + u] : 0.0;
  }
  if (has_p) {
    const double* row2 = lw_row(L, s2, i2, cur2, w, W, nmo2);  // cached row of the current position
#pragma unroll
    for (int c = 0; c < 4; ++c)
#pragma unroll
      for (int u = 0; u < NS; ++u) rv2[c][u] = (jb2 + u < je2) ? row2[c * nmo2 + jb2 + u] : 0.0;
  }
  bool acc = false;
  PQA_PCLK(1);
  if (has_a) {
    const int e = ea;
    {
      double r0 = 0.0, r1 = 0.0, r2 = 0.0, r3 = 0.0;
#pragma unroll
      for (int u = 0; u < NS; ++u)
        if (jb + u < je) { const double t = tinv[u]; r0 += rv[0][u] * t; r1 += rv[1][u] * t; r2 += rv[2][u] * t; r3 += rv[3][u] * t; }
#ifdef PQA_PRE_CLK
      if (blockIdx.x < 256 && threadIdx.x == 0) pqa_pre_clk[blockIdx.x * 16 + 2] = wall_clock64() + (r0 == 1.2345e300 ? 1 : 0);  // the Slater sums (= every load of the first two round trips) done
#endif
      double U, gg[3];
      jas_pre<PBC, NP, NA>(S, J, e, npx, npy, npz, a.has_jastrow, g, G, pcx, pcy, pcz, atx, aty, atz, bcA0, bcA1, acA, U, gg);
#ifdef PQA_PRE_CLK
      if (blockIdx.x < 256 && threadIdx.x == 0) pqa_pre_clk[blockIdx.x * 16 + 3] = wall_clock64() + (U == 1.2345e300 ? 1 : 0);  // this thread's Jastrow pairs done
#endif
#if PQA_PRE_DBG & 1
      double p[PR];
      lw_move_sums<PBC, false>(S, L, e, a.has_jastrow, npx, npy, npz, lw_row(L, s, i, cur ^ 1, w, W, nmo), W, w, g, G, p);
#elif PQA_PRE_DBG & 4
      double lp_, ee_, ei_;
      jas_eval_lane<1, PBC>(S, L.xt, W, w, e, npx, npy, npz, a.has_jastrow, g, G, U, gg, lp_, ee_, ei_);
      const double p[PR] = {r0, r1, r2, r3, U, gg[0], gg[1], gg[2]};
#else
      const double p[PR] = {r0, r1, r2, r3, U, gg[0], gg[1], gg[2]};
#endif
#pragma unroll
      for (int c = 0; c < PR; ++c) shP[(c * G + g) * NW + lane] = p[c];
    }
    __syncthreads();
    PQA_PCLK(4);
    double v[PR];
    if (GW > 16) {
      if (g < PR) {
        double tsum = 0.0;
        for (int gg = 0; gg < G; ++gg) tsum += shP[(g * G + gg) * NW + lane];
        shTot[g * NW + lane] = tsum;
      }
      __syncthreads();
      PQA_PCLK(5);
#pragma unroll
      for (int c = 0; c < PR; ++c) v[c] = shTot[c * NW + lane];
    } else {
#pragma unroll
      for (int c = 0; c < PR; ++c) v[c] = 0.0;
      for (int gg = 0; gg < G; ++gg) {
#pragma unroll
        for (int c = 0; c < PR; ++c) v[c] += shP[(c * G + gg) * NW + lane];
      }
    }
    // ---- Metropolis decision (mc.py:124-132; dmc.py:57-70), every group the same numbers: k_step_lw's lines
    double gx, gy, gz, dr, di;
    lw_slater_terms<false, PR>(v, gx, gy, gz, dr, di);
    gx += v[JU + 1]; gy += v[JU + 2]; gz += v[JU + 3];
    double val2;
    { const double val = finite_or(dr, 1.0); val2 = val * val; }
    if (a.has_jastrow) { const double ej = exp(v[JU] - ax7[6]); val2 *= ej * ej; }
    const double a0 = ax7[0], a1 = ax7[1], a2 = ax7[2], d0 = ax7[3], d1 = ax7[4], d2 = ax7[5];
    const double fwd = a0 * a0 + a1 * a1 + a2 * a2;
    double bx, by, bz;
    if (mb.dmc) {
      limdrift_dmc(gx, gy, gz, mb.tstep);
      bx = a0 + d0 + gx; by = a1 + d1 + gy; bz = a2 + d2 + gz;
    } else {
      limdrift3(gx, gy, gz);
      bx = a0 + mb.tstep * (d0 + gx); by = a1 + mb.tstep * (d1 + gy); bz = a2 + mb.tstep * (d2 + gz);
    }
    const double bwd = bx * bx + by * by + bz * bz;
    const double t_prob = exp(1.0 / (2.0 * mb.tstep) * (fwd - bwd));
    double ratio = val2 * t_prob;
    if (mb.dmc) {
      const double dv = finite_or(dr, 1.0);
      ratio *= (dv > 0.0) ? 1.0 : ((dv < 0.0) ? -1.0 : 0.0);  // fixed node (dmc.py:64-66)
    }
    double u;
    if (mb.unif) u = uu;
    else {
      const Philox ph = philox(mb.seed, (uint32_t)w, (uint32_t)e, PQA_STREAM_ACCEPT, mb.step);
      u = u01(ph.c[0], ph.c[1]);
    }
    acc = ratio > u;
#ifdef PQA_PRE_CLK
    if (blockIdx.x < 256 && threadIdx.x == 0) pqa_pre_clk[blockIdx.x * 16 + 6] = wall_clock64() + (acc ? 0 : 0);  // decided
#endif
    if (lead) {
      if (mb.dmc) {
        const double rx = a0 + d0, ry = a1 + d1, rz = a2 + d2;
        const double r2 = rx * rx + ry * ry + rz * rz;
        mb.r2_prop[w] += r2;
        if (acc) mb.r2_acc[w] += r2;
      }
      a.act[w] = acc;
#if !(PQA_PRE_DBG & 32)
      if (mb.accept_rec) mb.accept_rec[(size_t)e * W + w] = acc;
#endif
      if (acc) {
        mb.acc_w[w] += 1;
        L.sel[s][(size_t)i * W + w] = (uint8_t)(cur ^ 1);
        double* xe = L.xt + (size_t)e * 3 * W + w;
        xe[0] = npx; xe[W] = npy; xe[2 * W] = npz;
        if (mb.wrap) {
          int* wp = mb.wrap + ((size_t)w * S.nelec + e) * 3;
          wp[0] += mb.dwrap[3 * w]; wp[1] += mb.dwrap[3 * w + 1]; wp[2] += mb.dwrap[3 * w + 2];
        }
        L.dsign[s][w] *= (dr > 0.0) ? 1.0 : ((dr < 0.0) ? -1.0 : dr);
        L.dlog[s][w] += log(fabs(dr));
      }
    }
    // ---- V = new orbital row, R = T_old[i] / ratio (slater.py:88-94): this group's slots, from the registers
    {
      const double inv = 1.0 / dr;
      const bool st = acc && live;
#pragma unroll
      for (int u = 0; u < NS; ++u) {
        const int k = jb + u;
        if (k < je) {
          const double vv = acc ? rv[0][u] : 0.0;
          const double rr = acc ? tinv[u] * inv : 0.0;
          shV[k * NW + lane] = vv;
          shR[k * NW + lane] = rr;
          if (st) { a.Vbuf[(size_t)k * W + w] = vv; a.Rbuf[(size_t)k * W + w] = rr; }
        }
      }
    }
    __syncthreads();
    PQA_PCLK(7);
    // ---- the block rows
    if (QUART) {
      // quarter sums of V . T[jrow] -> LDS (the partial-sum planes are free: every group has read its totals), then every thread of
      // the quartet combines the four in the PQA_ROWDOT order and updates its own quarter of the row
      double pq = 0.0;
      if (has_row && jrow != i) {
#pragma unroll
        for (int u = 0; u < TBN; ++u)
          if (kq0 + u < n) pq += shV[(kq0 + u) * NW + lane] * tb[u];
      }
      shP[g * NW + lane] = pq;
      __syncthreads();
#ifdef PQA_PRE_CLK
      if (blockIdx.x < 256 && threadIdx.x == 0) pqa_pre_clk[blockIdx.x * 16 + 12] = wall_clock64();
#endif
      if (has_row) {
        double* Tj = L.Tt[s] + (size_t)jrow * n * W + w;
        const bool any = __any(acc);  // k_step_lw stores nothing where no lane of the wave accepted
        if (jrow == i) {
          if (acc && live) {
#pragma unroll
            for (int u = 0; u < TBN; ++u)
              if (kq0 + u < n) Tj[(size_t)(kq0 + u) * W] = shR[(kq0 + u) * NW + lane];
          }
        } else {
          const int gq = g & ~3;
          const double tmp = ((shP[gq * NW + lane] + shP[(gq + 1) * NW + lane]) + shP[(gq + 2) * NW + lane]) + shP[(gq + 3) * NW + lane];
#pragma unroll
          for (int u = 0; u < TBN; ++u)
            if (kq0 + u < n) tb[u] = acc ? tb[u] - shR[(kq0 + u) * NW + lane] * tmp : tb[u];
#ifdef PQA_PRE_CLK
          if (blockIdx.x < 256 && threadIdx.x == 0) pqa_pre_clk[blockIdx.x * 16 + 13] = wall_clock64() + (tb[0] == 1.2345e300 ? 1 : 0);
#endif
          if (live && any) {
#pragma unroll
            for (int u = 0; u < TBN; ++u)
              if (kq0 + u < n) Tj[(size_t)(kq0 + u) * W] = tb[u];
          }
          PQA_PCLK(14);
          if (handoff && jrow == i2) {
#pragma unroll
            for (int u = 0; u < TBN; ++u)
              if (kq0 + u < n) shT[(kq0 + u) * NW + lane] = tb[u];
          }
        }
      }
    } else if (has_row) {
      double* Tj = L.Tt[s] + (size_t)jrow * n * W + w;
      const bool any = __any(acc);  // k_step_lw stores nothing where no lane of the wave accepted
      if (jrow == i) {
        if (acc && live) {
#pragma unroll
          for (int k = 0; k < NMAX; ++k)
            if (k < n) Tj[(size_t)k * W] = shR[k * NW + lane];
        }
      } else {
        double p4[4] = {0.0, 0.0, 0.0, 0.0};  // PQA_ROWDOT
#pragma unroll
        for (int k = 0; k < TBN; ++k)
          if (k < n) p4[k / (NMAX / 4)] += shV[k * NW + lane] * tb[k];
        const double tmp = ((p4[0] + p4[1]) + p4[2]) + p4[3];
#pragma unroll
        for (int k = 0; k < TBN; ++k)
          if (k < n) tb[k] = acc ? tb[k] - shR[k * NW + lane] * tmp : tb[k];
        if (live && any) {
#pragma unroll
          for (int k = 0; k < TBN; ++k)
            if (k < n) Tj[(size_t)k * W] = tb[k];
        }
        if (handoff && jrow == i2) {
#pragma unroll
          for (int k = 0; k < TBN; ++k)
            if (k < n) shT[k * NW + lane] = tb[k];
        }
      }
    }
    __syncthreads();
    PQA_PCLK(8);
    // the accepted proposal replaces the electron's coordinate in the register copy
#if PQA_PRE_DBG & 16
#pragma unroll
    for (int m = 0; m < NP; ++m) {
      const int j = g + m * G;
      const double* xj = L.xt + (size_t)(j < S.nelec ? j : 0) * 3 * W + w;
      pcx[m] = xj[0]; pcy[m] = xj[W]; pcz[m] = xj[2 * W];
    }
#else
#pragma unroll
    for (int m = 0; m < NP; ++m)
      if (acc && g + m * G == e) { pcx[m] = npx; pcy[m] = npy; pcz[m] = npz; }
#endif
  }
  if (has_p) {
    // ---- drift at the current position, proposal r' = r + sqrt(tau) z + tau limdrift(grad)   (mc.py:117-121)
    const int e = ep;
    {
      if (handoff) {
#pragma unroll
        for (int u = 0; u < NS; ++u) tinv2[u] = (jb2 + u < je2) ? shT[(jb2 + u) * NW + lane] : 0.0;
      }
      double r0 = 0.0, r1 = 0.0, r2 = 0.0, r3 = 0.0;
#pragma unroll
      for (int u = 0; u < NS; ++u)
        if (jb2 + u < je2) { const double t = tinv2[u]; r0 += rv2[0][u] * t; r1 += rv2[1][u] * t; r2 += rv2[2][u] * t; r3 += rv2[3][u] * t; }
      double U, gg[3];
      jas_pre<PBC, NP, NA>(S, J, e, xe2[0], xe2[1], xe2[2], a.has_jastrow, g, G, pcx, pcy, pcz, atx, aty, atz, bcP0, bcP1, acP, U, gg);
#if PQA_PRE_DBG & 32
      {
        double U2, g2[3], lp_, ee_, ei_;
        jas_eval_lane<1, PBC>(S, L.xt, W, w, e, xe2[0], xe2[1], xe2[2], a.has_jastrow, g, G, U2, g2, lp_, ee_, ei_);
        const int code = (U != U2 ? 1 : 0) | (gg[0] != g2[0] ? 2 : 0) | (gg[1] != g2[1] ? 4 : 0) | (gg[2] != g2[2] ? 8 : 0);
        shV[g * NW + lane] = (double)(code ? (code | (g << 4)) : 0);
      }
#endif
#if PQA_PRE_DBG & 2
      double p[PR];
      lw_move_sums<PBC, false>(S, L, e, a.has_jastrow, xe2[0], xe2[1], xe2[2], lw_row(L, s2, i2, cur2, w, W, nmo2), W, w, g, G, p);
#elif PQA_PRE_DBG & 8
      double lp_, ee_, ei_;
      jas_eval_lane<1, PBC>(S, L.xt, W, w, e, xe2[0], xe2[1], xe2[2], a.has_jastrow, g, G, U, gg, lp_, ee_, ei_);
      const double p[PR] = {r0, r1, r2, r3, U, gg[0], gg[1], gg[2]};
#else
      const double p[PR] = {r0, r1, r2, r3, U, gg[0], gg[1], gg[2]};
#endif
#pragma unroll
      for (int c = 0; c < PR; ++c) shP[(c * G + g) * NW + lane] = p[c];
    }
    __syncthreads();
    PQA_PCLK(9);
    if (GW > 16) {
      if (g < PR) {
        double tsum = 0.0;
        for (int gg = 0; gg < G; ++gg) tsum += shP[(g * G + gg) * NW + lane];
        shTot[g * NW + lane] = tsum;
      }
      __syncthreads();
    }
    PQA_PCLK(10);
    if (!lead) return;
    double v[PR];
    if (GW > 16) {
#pragma unroll
      for (int c = 0; c < PR; ++c) v[c] = shTot[c * NW + lane];
    } else {
#pragma unroll
      for (int c = 0; c < PR; ++c) v[c] = 0.0;
      for (int gg = 0; gg < G; ++gg) {
#pragma unroll
        for (int c = 0; c < PR; ++c) v[c] += shP[(c * G + gg) * NW + lane];
      }
    }
    double gx, gy, gz, dr, di;
    lw_slater_terms<false, PR>(v, gx, gy, gz, dr, di);
    gx += v[JU + 1]; gy += v[JU + 2]; gz += v[JU + 3];
    if (mb.dmc) limdrift_dmc(gx, gy, gz, mb.tstep);
    else limdrift3(gx, gy, gz);
    double z0, z1, z2, z3;
    if (mb.gauss) { z0 = zt[0]; z1 = zt[1]; z2 = zt[2]; }
    else {
      normal2(philox(mb.seed, (uint32_t)w, (uint32_t)e, PQA_STREAM_GAUSS_A, mb.step), z0, z1);
      normal2(philox(mb.seed, (uint32_t)w, (uint32_t)e, PQA_STREAM_GAUSS_B, mb.step), z2, z3);
    }
    const double sq = sqrt(mb.tstep);
    z0 *= sq; z1 *= sq; z2 *= sq;
    double* np_ = mb.newpos + 3 * w;
    const double df = mb.dmc ? 1.0 : mb.tstep;
    np_[0] = xe2[0] + z0 + gx * df;
    np_[1] = xe2[1] + z1 + gy * df;
    np_[2] = xe2[2] + z2 + gz * df;
    if (mb.dwrap) fold_cell(S, np_[0], np_[1], np_[2], mb.dwrap + 3 * w);  // make_irreducible, mc.py:121
    double* ao = L.auxt + w;
    ao[0] = z0; ao[W] = z1; ao[2 * W] = z2; ao[3 * W] = gx; ao[4 * W] = gy; ao[5 * W] = gz; ao[6 * W] = v[JU];
    PQA_PCLK(11);
  }
}

// ---------------------------------------------------------------- kinetic + Coulomb
// thread = (walker, electron), walker fastest.  part [5][N][W]: ke_e, grad2_e, ee_e, ei_e, U_e (Jastrow exponent of electron e)
// block = (64 walkers, PQA_KIN_EB electrons): one wave per electron, so the electron index — and with it the spin, the orbital
// occupation list and every Jastrow table address — must stay wave-uniform (scalar loads): threadIdx.y goes through
// readfirstlane.  (As a plain per-lane value it turned the occupation look-up of the inner loop into a dependent vector load:
// 2.0 -> 3.3 ms per evaluation.)  Every electron's thread walks ALL coordinates of its walker for the Jastrow and Coulomb sums,
// so the 64 electron-waves of a walker group pull the group's coordinates through the fabric 64 times (the counters show
// 195 KB per walker against 96 KB of inverse + cache rows; the re-reads hit the Infinity Cache).
// Where the time goes at 65 536 walkers (round 6, tools/scratch/row_probe.hip + kernel variants): the row cache alone streams at 5.0 (a lane per
// row) to 5.8 TB/s (a quad per line), the inverse planes alone at 5.8 TB/s — 1.1 ms for both — but read in the same kernel, even by different
// blocks, the two streams take 1.5-1.9 ms (tile-blocked inverse: 1.55): the mix, not either pattern, costs the bandwidth.  The coordinate walk
// of the Jastrow / Coulomb sums (96 KB per wave out of L2) adds 0.28 ms on top although its arithmetic is 0.07 ms.  PQA_KIN_V: 1 = a lane
// streams its own row (1.86 ms), 3 = quad-cooperative lines in the QUAD instantiation (1.79 ms, default; shards below 16 384 walkers take the
// other one: 163 against 121 registers, and 4 096 walkers are one round of waves only at four per SIMD — 195 vs 231 us for C5), 4 = V1 with
// half a component requested ahead (1.84 ms).
#ifndef PQA_KIN_EB
#define PQA_KIN_EB 1
#endif
#ifndef PQA_KIN_V
#define PQA_KIN_V 3
#endif
// PQA_KIN_C0 = 1 (default): the quad-cooperative instantiation does not read the VALUE block of the cached rows.  The reference divides the
// derivative sums by sum_j phi_j(r_i) T_ji (slater.py gradient_laplacian: ratios[1:] / ratios[0]) — at the electron's own position that is row i of
// the Slater matrix times column i of its inverse: 1, up to the rounding the inverse has accumulated since the last recompute.  Taking it as 1
// saves a fifth of the row traffic (1.78 -> 1.68 ms at 65 536 walkers); measured on (H2O)8, 16 384 walkers, 40 sweeps without a recompute, the
// steps' kinetic-energy means agree with the dividing build to the last bit or one ulp (<= 4e-16 relative; tools/scratch/c0_check.py).
// -DPQA_KIN_C0=0 restores the division.
#ifndef PQA_KIN_C0
#define PQA_KIN_C0 1
#endif
template <bool PBC, bool CX = false, bool QUAD = false>  // QUAD: quad-cooperative row reads (large shards; real determinants, W % 4 == 0)
static __global__ __launch_bounds__(64 * PQA_KIN_EB) void k_kinetic_lw(SysDev S, LwState L, int has_jastrow, long W, double* __restrict__ part) {
  // Block b -> (walker group, electron block): the electron blocks of ONE walker group sit 8 apart in the linear block order, so
  // they land on the same XCD (blocks go to the XCDs round-robin) and run at about the same time: the group's coordinates, which
  // every one of them walks, come out of that XCD's L2 after the first.  (With the electron on grid.y the 64 blocks of a group were
  // 1 024 blocks apart and every one fetched the coordinates again: 195 KB per walker through the fabric against 98 KB of rows and
  // inverses, at 6.2 TB/s — the kernel was bound by re-reads.)
  const int neb_ = (S.nelec + PQA_KIN_EB - 1) / PQA_KIN_EB;
  const long chunk = (long)blockIdx.x / (8 * neb_);
  const int rem = (int)((long)blockIdx.x % (8 * neb_));
  const long w = (chunk * 8 + (rem & 7)) * 64 + threadIdx.x;
  const int e = (rem >> 3) * PQA_KIN_EB + (PQA_KIN_EB > 1 ? __builtin_amdgcn_readfirstlane((int)threadIdx.y) : 0);
  if (w >= W || e >= S.nelec) return;
  const int s = e >= S.nup, i = e - s * S.nup, n = s ? S.ndn : S.nup, nmo = S.nmo[s];
  constexpr int CF = CX ? 2 : 1;
  double r[5] = {0.0, 0.0, 0.0, 0.0, 0.0}, q[5] = {0.0, 0.0, 0.0, 0.0, 0.0};
  {
    const double* Ti = L.Tt[s] + (size_t)i * n * CF * W + w;
    const double* row = lw_row(L, s, i, L.sel[s][(size_t)i * W + w], w, W, nmo);  // [5][nmo], this lane's own 1280-B row
    const int* occ = S.det_occ[s];
    const int nh = nmo / CF;
    if (!CX && S.occ_ident[s] && ((n | nmo) & 7) == 0) {
      // ground-state occupation: whole 64-byte lines of the lane's own row, two adjacent 32-byte loads each, used up at once
      // (walking 4 slots of all five components first left every line half used until the next round: 320 lines per wave in
      // flight, more than L1 keeps with 16 waves per CU — the kernel took 3.5 ms instead of 1.9)
#if PQA_KIN_V == 3 || PQA_KIN_V == 1
      // Quad-cooperative rows: the four lanes of a quad (walkers wq .. wq + 3) read the row of each of the four walkers together, 16 bytes per
      // lane = one whole 64-byte line per quad and instruction, 16 lines per wave-instruction.  (A lane streaming its own row touches a line of
      // its own per load: 64 lines per instruction, and the address path — one line per cycle and CU — not HBM set the kernel's time: 3.7 TB/s.)
      // Lane q of the quad holds orbital slots j + 2q, j + 2q + 1 of each 8-slot line, so it takes those two slots of the inverse row of ALL four
      // walkers (two 32-byte loads of T[slot][wq .. wq + 3]) and accumulates its share of all four walkers' sums; the quad adds the shares at the
      // end (DPP) and lane q keeps walker wq + q's.
      if (QUAD && PQA_KIN_V == 3 && n <= 32 && (W & 3) == 0) {
        const int q = (int)threadIdx.x & 3;
        const long wq = w - q;
        const uint32_t s4 = *reinterpret_cast<const uint32_t*>(L.sel[s] + (size_t)i * W + wq);
        const double* rq[4];
#pragma unroll
        for (int t = 0; t < 4; ++t) rq[t] = lw_row(L, s, i, (int)((s4 >> (8 * t)) & 0xffu), wq + t, W, nmo) + 2 * q;
        const double* Tq = L.Tt[s] + ((size_t)i * n + 2 * q) * W + wq;
        double p[4][5];
#pragma unroll
        for (int t = 0; t < 4; ++t)
#pragma unroll
          for (int c = 0; c < 5; ++c) p[t][c] = 0.0;
        for (int j = 0; j < n; j += 8) {  // one line of every component of the four rows (20 loads) + the quad's share of the inverse in flight
          const double4 ta = *reinterpret_cast<const double4*>(Tq + (size_t)j * W), tb = *reinterpret_cast<const double4*>(Tq + (size_t)(j + 1) * W);
          double2 v[20];
#pragma unroll
          for (int c = PQA_KIN_C0; c < 5; ++c)
#pragma unroll
            for (int t = 0; t < 4; ++t) v[c * 4 + t] = *reinterpret_cast<const double2*>(rq[t] + c * nmo + j);
          __builtin_amdgcn_sched_barrier(0);
          const double t0[4] = {ta.x, ta.y, ta.z, ta.w}, t1[4] = {tb.x, tb.y, tb.z, tb.w};
#pragma unroll
          for (int c = PQA_KIN_C0; c < 5; ++c)
#pragma unroll
            for (int t = 0; t < 4; ++t) { p[t][c] += v[c * 4 + t].x * t0[t]; p[t][c] += v[c * 4 + t].y * t1[t]; }
          __builtin_amdgcn_sched_barrier(0);
        }
#pragma unroll
        for (int c = PQA_KIN_C0; c < 5; ++c) {
          double mine = 0.0;
#pragma unroll
          for (int t = 0; t < 4; ++t) {
            double x = p[t][c];
            x += quad_dpp<0xb1>(x);  // quad_perm [1, 0, 3, 2]
            x += quad_dpp<0x4e>(x);  // quad_perm [2, 3, 0, 1]: the quad's total in every lane
            mine = (q == t) ? x : mine;
          }
          r[c] = mine;
        }
        if (PQA_KIN_C0) r[0] = 1.0;
      } else
#endif
#if PQA_KIN_V == 4
      // component-major, a whole component (n doubles = n / 8 lines of the lane's own row) requested before the previous component's products:
      // 256 bytes per lane in flight all the time (the compiler's own schedule of V == 1 waits for every line before it asks for the next:
      // one 64-byte line per lane in flight, 3.7 TB/s)
      if (n <= 32) {
        double t[32];
#pragma unroll
        for (int u = 0; u < 32; ++u) t[u] = (u < n) ? Ti[(size_t)u * W] : 0.0;
        // half a component (16 slots = 128 bytes) per buffer: request the next half, then use the previous one
        double4 A[4], B[4];
#define PQA_KIN_SB __builtin_amdgcn_sched_barrier(0);
#define PQA_KIN_LOAD(BUF, H) _Pragma("unroll") for (int k = 0; k < 4; ++k) BUF[k] = *reinterpret_cast<const double4*>(row + ((H) >> 1) * nmo + ((((H) & 1) * 16 + 4 * k) < n ? ((H) & 1) * 16 + 4 * k : 0));  /* (slots >= n: t = 0) */
#define PQA_KIN_DOT(BUF, H)                                                                                       \
  _Pragma("unroll") for (int k = 0; k < 4; ++k) {                                                                 \
    const int j_ = ((H) & 1) * 16 + 4 * k;                                                                        \
    r[(H) >> 1] += BUF[k].x * t[j_]; r[(H) >> 1] += BUF[k].y * t[j_ + 1]; r[(H) >> 1] += BUF[k].z * t[j_ + 2]; r[(H) >> 1] += BUF[k].w * t[j_ + 3]; \
  }
        PQA_KIN_LOAD(A, 0)
        PQA_KIN_LOAD(B, 1)
        PQA_KIN_SB PQA_KIN_DOT(A, 0) PQA_KIN_SB PQA_KIN_LOAD(A, 2)
        PQA_KIN_SB PQA_KIN_DOT(B, 1) PQA_KIN_SB PQA_KIN_LOAD(B, 3)
        PQA_KIN_SB PQA_KIN_DOT(A, 2) PQA_KIN_SB PQA_KIN_LOAD(A, 4)
        PQA_KIN_SB PQA_KIN_DOT(B, 3) PQA_KIN_SB PQA_KIN_LOAD(B, 5)
        PQA_KIN_SB PQA_KIN_DOT(A, 4) PQA_KIN_SB PQA_KIN_LOAD(A, 6)
        PQA_KIN_SB PQA_KIN_DOT(B, 5) PQA_KIN_SB PQA_KIN_LOAD(B, 7)
        PQA_KIN_SB PQA_KIN_DOT(A, 6) PQA_KIN_SB PQA_KIN_LOAD(A, 8)
        PQA_KIN_SB PQA_KIN_DOT(B, 7) PQA_KIN_SB PQA_KIN_LOAD(B, 9)
        PQA_KIN_SB PQA_KIN_DOT(A, 8)
        PQA_KIN_DOT(B, 9)
#undef PQA_KIN_LOAD
#undef PQA_KIN_DOT
#undef PQA_KIN_SB
      } else
#elif PQA_KIN_V == 1 || PQA_KIN_V == 3  // component-major: the lane streams its row front to back (adjacent lines back to back), inverse row in registers
      if (n <= 32) {
        double t[32];
#pragma unroll
        for (int u = 0; u < 32; ++u) t[u] = (u < n) ? Ti[(size_t)u * W] : 0.0;
#pragma unroll
        for (int c = 0; c < 5; ++c)
#pragma unroll
          for (int j = 0; j < 32; j += 8)
            if (j < n) {
              const double4 lo = *reinterpret_cast<const double4*>(row + c * nmo + j), hi = *reinterpret_cast<const double4*>(row + c * nmo + j + 4);
              r[c] += lo.x * t[j]; r[c] += lo.y * t[j + 1]; r[c] += lo.z * t[j + 2]; r[c] += lo.w * t[j + 3];
              r[c] += hi.x * t[j + 4]; r[c] += hi.y * t[j + 5]; r[c] += hi.z * t[j + 6]; r[c] += hi.w * t[j + 7];
            }
      } else
#elif PQA_KIN_V == 2  // component-major, the inverse row re-read per component (coalesced, cache hits)
      if (true) {
#pragma unroll
        for (int c = 0; c < 5; ++c)
          for (int j = 0; j < n; j += 8) {
            double t[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) t[u] = Ti[(size_t)(j + u) * W];
            const double4 lo = *reinterpret_cast<const double4*>(row + c * nmo + j), hi = *reinterpret_cast<const double4*>(row + c * nmo + j + 4);
            r[c] += lo.x * t[0]; r[c] += lo.y * t[1]; r[c] += lo.z * t[2]; r[c] += lo.w * t[3];
            r[c] += hi.x * t[4]; r[c] += hi.y * t[5]; r[c] += hi.z * t[6]; r[c] += hi.w * t[7];
          }
      } else
#endif
      for (int j = 0; j < n; j += 8) {
        double t[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) t[u] = Ti[(size_t)(j + u) * W];
#pragma unroll
        for (int c = 0; c < 5; ++c) {
          const double4 lo = *reinterpret_cast<const double4*>(row + c * nmo + j), hi = *reinterpret_cast<const double4*>(row + c * nmo + j + 4);
          r[c] += lo.x * t[0]; r[c] += lo.y * t[1]; r[c] += lo.z * t[2]; r[c] += lo.w * t[3];
          r[c] += hi.x * t[4]; r[c] += hi.y * t[5]; r[c] += hi.z * t[6]; r[c] += hi.w * t[7];
        }
      }
    } else
    if (CX && S.occ_ident[s] && ((n | nh) & 7) == 0) {
      // complex rows, ground-state occupation: whole 64-byte lines of the lane's own row — per component the real and the
      // imaginary block of 8 slots are one line each (two adjacent 32-byte loads).  Element by element every 8-byte load of the
      // wave touched 64 different lines: 12 k line requests per wave, and the kernel was bound by them (0.72 ms per evaluation of
      // the twisted 32-electron cell at 8 192 walkers against 0.30 for the real 64-electron cell with as many threads).
      for (int j0 = 0; j0 < n; j0 += 8) {
        double tr[8], ti[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) { tr[u] = Ti[(size_t)(2 * (j0 + u)) * W]; ti[u] = Ti[(size_t)(2 * (j0 + u) + 1) * W]; }
#pragma unroll
        for (int c = 0; c < 5; ++c) {
          const double4 al = *reinterpret_cast<const double4*>(row + c * nmo + j0), ah = *reinterpret_cast<const double4*>(row + c * nmo + j0 + 4);
          const double4 bl = *reinterpret_cast<const double4*>(row + c * nmo + nh + j0), bh = *reinterpret_cast<const double4*>(row + c * nmo + nh + j0 + 4);
          const double a[8] = {al.x, al.y, al.z, al.w, ah.x, ah.y, ah.z, ah.w}, b[8] = {bl.x, bl.y, bl.z, bl.w, bh.x, bh.y, bh.z, bh.w};
#pragma unroll
          for (int u = 0; u < 8; ++u) { r[c] += a[u] * tr[u] - b[u] * ti[u]; q[c] += a[u] * ti[u] + b[u] * tr[u]; }
        }
      }
    } else
    if (CX) {
      // four slots' worth of loads (2 inverse planes + 10 row elements each) in flight before the first product: one slot at a time
      // every iteration waited for its own loads — 0.76 ms per evaluation of the twisted 32-electron cell at 8 192 walkers
      for (int j0 = 0; j0 < n; j0 += 4) {
        double tr[4], ti[4], a[4][5], b[4][5];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          const int j = (j0 + u < n) ? j0 + u : n - 1;
          const double* cj = row + occ[j];
          tr[u] = Ti[(size_t)(2 * j) * W]; ti[u] = Ti[(size_t)(2 * j + 1) * W];
#pragma unroll
          for (int c = 0; c < 5; ++c) { a[u][c] = cj[c * nmo]; b[u][c] = cj[c * nmo + nh]; }
        }
#pragma unroll
        for (int u = 0; u < 4; ++u)
          if (j0 + u < n) {
#pragma unroll
            for (int c = 0; c < 5; ++c) { r[c] += a[u][c] * tr[u] - b[u][c] * ti[u]; q[c] += a[u][c] * ti[u] + b[u][c] * tr[u]; }
          }
      }
    } else
    for (int j = 0; j < n; ++j) {
      const double* cj = row + occ[j];
      if (CX) {
        const double tr = Ti[(size_t)(2 * j) * W], ti = Ti[(size_t)(2 * j + 1) * W];
#pragma unroll
        for (int c = 0; c < 5; ++c) {
          const double a = cj[c * nmo], b = cj[c * nmo + nh];
          r[c] += a * tr - b * ti; q[c] += a * ti + b * tr;
        }
      } else {
        const double t = Ti[(size_t)j * W];
#pragma unroll
        for (int c = 0; c < 5; ++c) r[c] += cj[c * nmo] * t;
      }
    }
  }
  double gs0, gs1, gs2, ls, gi2 = 0.0;
  if (CX) {  // grad Psi / Psi complex; only Re(lap) enters (the Jastrow gradient is real): energy.py:57-65 on complex ratios
    const double d = 1.0 / (r[0] * r[0] + q[0] * q[0]);
    gs0 = (r[1] * r[0] + q[1] * q[0]) * d; gs1 = (r[2] * r[0] + q[2] * q[0]) * d; gs2 = (r[3] * r[0] + q[3] * q[0]) * d;
    const double i0 = (q[1] * r[0] - r[1] * q[0]) * d, i1 = (q[2] * r[0] - r[2] * q[0]) * d, i2 = (q[3] * r[0] - r[3] * q[0]) * d;
    gi2 = i0 * i0 + i1 * i1 + i2 * i2;
    ls = (r[4] * r[0] + q[4] * q[0]) * d;
  } else {
    gs0 = r[1] / r[0]; gs1 = r[2] / r[0]; gs2 = r[3] / r[0]; ls = r[4] / r[0];
  }
  const double* xe = L.xt + (size_t)e * 3 * W + w;
  double U, gj[3], lj, ee, ei;
  jas_eval_lane<2, PBC, true, QUAD ? 8 : PQA_JAS_PF>(S, L.xt, W, w, e, xe[0], xe[W], xe[2 * W], has_jastrow, 0, 1, U, gj, lj, ee, ei);
  lj += gj[0] * gj[0] + gj[1] * gj[1] + gj[2] * gj[2];
  const double gx = gs0 + gj[0], gy = gs1 + gj[1], gz = gs2 + gj[2];
  const double lap = ls + lj + 2.0 * (gs0 * gj[0] + gs1 * gj[1] + gs2 * gj[2]);
  const size_t o = (size_t)e * W + w, NW = (size_t)S.nelec * W;
  part[o] = -0.5 * lap;
  part[NW + o] = gx * gx + gy * gy + gz * gz + gi2;
  part[2 * NW + o] = ee;
  part[3 * NW + o] = ei;
  part[4 * NW + o] = U;  // U_e at the electron's own position: the ECP pass needs it for every (electron, atom) entry it integrates
}

// out rows ke, ee, ei, grad2 (layout of k_kinetic_coulomb) = sums over electrons of part
template <int PQA_UNIT = 0>  // (a template so that only the units that launch it compile it)
static __global__ void k_kinetic_reduce(const double* __restrict__ part, int N, long W, double* __restrict__ out) {
  const long w = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (w >= W) return;
  const size_t NW = (size_t)N * W;
  double ke = 0.0, g2 = 0.0, ee = 0.0, ei = 0.0;
  for (int e = 0; e < N; ++e) {
    const size_t o = (size_t)e * W + w;
    ke += part[o]; g2 += part[NW + o]; ee += part[2 * NW + o]; ei += part[3 * NW + o];
  }
  out[w] = ke; out[W + w] = ee; out[2 * W + w] = ei; out[3 * W + w] = g2;
}

