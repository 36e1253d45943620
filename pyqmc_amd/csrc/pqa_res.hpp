// pqa_res.hpp — the RESIDENT electron sweep: all N single-electron moves of a walker tile in ONE launch, state on chip.
//
// north_star's prescription ("one walker per wavefront with LDS-staged MO-inverse and electron-electron distance tiles") in the
// form the measurements of rounds 2-4 point to.  The lane-per-walker sweep (pqa_lw.hpp) streams ~1 MB per walker-step through
// HBM in ~155 dependent launches (k_orb -> k_step_lw per move, k_flush_lw per electron block); both kernel families are bound by
// the wave slots they hold, and small shards are a 128-link chain of latency-bound launches.  Here a block of 512 threads owns
// 16 walkers — exactly one 16-point fp64 MFMA tile — for the whole sweep:
//   * the transposed inverse of the current spin lives in REGISTERS: thread (walker wl = tid / 32, row r = tid % 32) holds the
//     32 columns of row r (64 VGPRs; a 512-thread block has 256 per thread), the walker's coordinates two electrons per thread;
//     a walker is 32 consecutive lanes, so every per-walker sum is a DPP row reduction plus one 16-lane swizzle — no barrier;
//   * per move the block evaluates the 16 proposals' atomic orbitals cooperatively (thread = (point, one of 32 lane groups),
//     shell lists balanced by the phase-1 cost model) into ONE LDS tile holding the whole basis (5 x K x 16 doubles: 118 KB for
//     the 184-AO (H2O)8 basis; several passes when it does not fit), contracts it on v_mfma_f64_16x16x4_f64 with the K dimension
//     split over the waves (partials meet in LDS and are added in a fixed order), and decides / commits / proposes the next
//     electron without leaving the CU: Slater ratio sums against the register inverse, Jastrow pair sums from register
//     coordinates, Metropolis test, Sherman-Morrison update of the 32 register rows, cache row of accepted moves;
//   * HBM is touched once per sweep for the inverse and the coordinates, once per move for the cached orbital row (1 KB read,
//     1.3 KB written on acceptance) and the random-number tapes: ~150 KB per walker-step instead of ~1 MB, one launch instead of
//     ~155.
// Arithmetic per walker is the reference's (vmc_worker body mc.py:112-137, limdrift :76-89; dmc.py:38-70 in DMC mode; Slater
// ratios slater.py:342-418, Sherman-Morrison slater.py:88-94, Jastrow jastrowspin.py:296-385 with func3d.py radial functions).
// Sums run in a different order than in the lane-per-walker kernels: trajectories agree to rounding, decisions are identical off
// measure-zero ties; the random numbers are the same Philox streams (drawn ahead by k_tile_draws), so the oracle replays apply.
// State in and out: the lane-per-walker SoA planes (xt, Tt, two-slot row cache + selectors, dsign / dlog) — the energy kernels
// run on them unchanged.
// Scope: single-determinant Slater factor with <= 32 electrons and <= 32 orbitals per spin, optional two-body Jastrow factor, l <= 3; open
// boundary conditions or (PBC) a periodic cell — lattice-summed AOs from per-(point, atom) image lists built in the block (phase 0) and
// accumulated in the tile with ds_add_f64 (phase 1), minimal-image Jastrow pairs, proposals folded into the cell.  CX: complex determinants
// in periodic cells (<= 16 electrons and orbitals per spin: a row of the inverse is 16 (re, im) pairs in the same 32 registers), with or
// without a twist — a twisted cell's tile holds the real AO rows and behind them the imaginary ones (coefficient rows [-C_im | C_re]), every
// image weighted by exp(i k_t . (fold + L_j)), the orbital rows by the wrap phase of the folded proposal; the walkers stay unfolded.
// Everything else keeps the lane-per-walker sweep.
#pragma once
#include "pqa_ao.hpp"
#include "pqa_jastrow.hpp"
#include "pqa_lw.hpp"
#include "pqa_vmc.hpp"

#define PQA_RES_NT 512     // threads per block
#define PQA_RES_NW 16      // walkers per block = points of the MFMA tile
#define PQA_RES_G 32       // lane groups of the AO phase
#define PQA_RES_MAXKS 16   // k-steps (4 AO rows each) of one pass a wave contracts at most
#define PQA_RES_MAXPASS 8
#define PQA_RES_WS 24     // doubles per walker of the per-walker scalars (wsc): 16 of the move + the cell wraps of a folded proposal (or, in a
                          // twisted cell, the folded proposal) + the imaginary part of the determinant phase + the wrap phase of the proposal
#define PQA_JQP (PQA_JQ + 1)  // doubles per (spin, ion) record of the block's LDS copy of the merged electron-ion numerators: [spin][ion][25] — the
                            // lanes of a wave read THEIR ion's record, and an odd stride spreads them over the banks (the global layout [ion][spin][24]
                            // put all 64 lanes on one bank pair: 11 reads of ~64 cycles per pair); slot 24: the ion's cusp coefficient of that spin
#define PQA_RES_RS 176     // doubles per walker of the combined orbital rows [5][32] (+16: walkers of a wave on disjoint LDS banks)

struct ResTab {
  int npass;                               // AO passes per move (1 when the whole basis fits the LDS tile)
  int kt;                                  // rows of the LDS tile (largest pass), multiple of 4
  int pass_row0[PQA_RES_MAXPASS + 1];      // padded coefficient rows [pass_row0[p], pass_row0[p + 1]) form pass p
  const int* grp_off;                      // [npass * 32 + 1]
  const int* grp_shell;                    // shells of (pass, lane group)
  const int* shell_row;                    // [nshell] padded coefficient row of the shell's first function
  int nlist;                               // entries of grp_shell
  int region;                              // doubles of the tile / partial-sum / orbital-row region
  int part_off;                            // offset of the K-partials in the region: 0 (one pass: they reuse the tile) or 80 kt
  int nprim_u;                             // distinct primitives: shells with the same (exponent, coefficient) sequence share one copy in LDS
  const double* prim_exp_u;                // [nprim_u]
  const double* prim_coef_u;
  const int* shell_q0;                     // [nshell] first primitive of the shell in those tables
  int pbc_off, icap, twist;                // (twist: complex AOs — the tile holds the real rows, then from row im_off on the imaginary rows)
  int im_off;                       // periodic: byte offset of the lattice vectors / image lists in the dynamic LDS, list capacity
};
// doubles per point of a K-partial [5][16 nt] (+ padding: the four point quartets of a wave on different banks)
__host__ __device__ inline int res_ps(int nt) { return 80 * nt + 16; }
// (ResTab::icap: admitted images per (point, atom) the block's image lists hold — 32, 24, 16 or 12, the most the LDS budget allows with the
// whole basis in one tile; a pair with more takes the direct tests)
// periodic instantiation: candidate lattice vectors [nL][3], image lists [natom][16][icap] and their lengths [natom][16]
#define PQA_RES_NCUT 6   // distinct shell cut-offs per atom the block's image lists sort by (more: the pair walks the candidate masks)
// + shell cut-offs [nshell], per atom: cut-off + PQA_RES_NCUT class cut-offs (doubles), candidates / classes / membership class (ints),
// + the cell (inverse lattice, lattice, inverse primitive lattice: 27 doubles, the two mask-table addresses) and the membership rule's
// integers (supercell matrix, M, E, G, flags, atom_n[natom][3])
// twisted cells: + image phases [nL][2], the fold phase of every (point, atom) pair [natom][16][2]
__host__ __device__ inline size_t res_lds_pbc(int natom, int nL, int icap, int nshell, int twist = 0) {
  return ((size_t)3 * nL + nshell + (size_t)natom * (1 + PQA_RES_NCUT) + 36 + (twist ? 2 * (size_t)nL + 32 * (size_t)natom : 0)) * sizeof(double) + (((size_t)(natom * 6 + 16) * sizeof(int) + 7) & ~(size_t)7) +
         (((size_t)natom * 16 * (icap + 1) + 7) & ~(size_t)7);
}
__host__ __device__ inline size_t res_lds_fixed(int nshell, int nprim, int natom, int na, int nlist, int npass) {
  const size_t d = 16 * 32 + 16 * PQA_RES_WS + 2 * (size_t)nprim + 3 * (size_t)natom + 2 * (size_t)natom * (na > 0 ? na : 1) +
                   2 * (size_t)natom * PQA_JQP + (3 * PQA_JQ + 24);
  const size_t i = 5 * (size_t)nshell + (size_t)nlist + (size_t)npass * 32 + 1 + 64;
  return d * sizeof(double) + i * sizeof(int);
}

template <int CTRL>
__device__ __forceinline__ double res_dpp(double v) {
  const int lo = __builtin_amdgcn_update_dpp(0, __double2loint(v), CTRL, 0xf, 0xf, false);
  const int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(v), CTRL, 0xf, 0xf, false);
  return __hiloint2double(hi, lo);
}
// Sum over the 32 lanes of a walker (half a wave), the SAME bits in every lane: four rotate-and-add steps inside each DPP row of
// 16 lanes (after the step by d the values have period d along the row, and x + y = y + x), then the two rows are exchanged with
// ds_swizzle (xor 16; no LDS memory access).  15 instructions per sum.
__device__ __forceinline__ double res_sum32(double v) {
  v += res_dpp<0x128>(v);  // row_ror:8
  v += res_dpp<0x124>(v);  // row_ror:4
  v += res_dpp<0x122>(v);  // row_ror:2
  v += res_dpp<0x121>(v);  // row_ror:1
  const int lo = __builtin_amdgcn_ds_swizzle(__double2loint(v), 0x401F);  // bit mode: and 0x1f, or 0, xor 0x10
  const int hi = __builtin_amdgcn_ds_swizzle(__double2hiint(v), 0x401F);
  return v + __hiloint2double(hi, lo);
}
// Block barrier for LDS hand-overs: __syncthreads() carries a fence that waits for vmcnt(0) — every cached-row / tape prefetch and
// every cache-row store in flight (HBM round trips) at each of the four barriers of a move.  Nothing a barrier of this kernel
// orders goes through global memory (a thread only re-reads global data written by earlier launches), so: this wave's LDS
// operations done, then the barrier.
__device__ __forceinline__ void res_block_sync() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }
__device__ __forceinline__ void res_lds_add(double* p, double v) { __hip_atomic_fetch_add(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); }
// LDS written by other lanes of the SAME wave: the hardware executes a wave's DS instructions in order; this keeps the compiler
// from moving accesses across the point.
__device__ __forceinline__ void res_wave_sync() {
  __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
  __builtin_amdgcn_wave_barrier();
}

// This thread's share of U_e, grad U_e at (rx, ry, rz).  Partner slots of thread r: the up electron r, the down electron nup + r
// (coordinates in registers) and the ions r, r + 32 — so the partner's spin, and with it the coefficient set, is wave-uniform.
// Merged route (S.jq_on, pqa_jastrow.hpp: the Pade functions of a basis as one rational function per pair; electron-electron
// numerators through scalar registers, the per-ion numerators from the block's LDS copy) or function by function (any basis).
template <bool PBC>
__device__ __forceinline__ void res_jas_part(const SysDev& S, int e, int r, double rx, double ry, double rz, const double (&cx)[2],
                                             const double (&cy)[2], const double (&cz)[2], const double* __restrict__ at_xyz,
                                             const double* __restrict__ acoef, const double* __restrict__ aq, double& U, double (&g)[3], int excl = -1) {
  const int edown = e >= S.nup;
  const double irb = 1.0 / S.rcut_b, ira = 1.0 / S.rcut_a;
  double u_ = 0.0, gx = 0.0, gy = 0.0, gz = 0.0;
  if (S.jq_on) {
    const bool bcusp = S.b_kind[0] == 1, acusp = S.a_kind[0] == 1, kb4 = S.jq_b > 3, ka4 = S.jq_a > 3;
    const double bcp = S.b_param[0], bca = S.b_aux[0], acp = S.a_param[0], aca = S.a_aux[0];
    double Db[5], Da[5];
#pragma unroll
    for (int i = 0; i < 5; ++i) { Db[i] = S.b_D[i]; Da[i] = S.a_D[i]; }
#pragma unroll
    for (int q = 0; q < 2; ++q) {
      const int j = (q ? S.nup : 0) + r;
      if (S.nb > 0 && r < (q ? S.ndn : S.nup) && j != e && j != excl) {
        double dx = rx - cx[q], dy = ry - cy[q], dz = rz - cz[q];
        if (PBC) min_image_j(S, dx, dy, dz);
        double rr, ri;
        sqrt_rinv(dx * dx + dy * dy + dz * dz, rr, ri);
        if (rr < S.rcut_b) {
          const RadShared sh = rad_shared_ri<1>(rr, ri, irb);
          const double* qq = S.bq + (edown + q) * PQA_JQ;
          const MergedSums m = kb4 ? pade_merged<1, 4, true>(Db, qq, sh.p) : pade_merged<1, 3, true>(Db, qq, sh.p);
          u_ += sh.omp * m.S1;
          double sg = sh.c0 * m.S2;
          if (bcusp) {
            double v, gf, lpl;
            rad_fn<1>(1, bcp, bca, S.rcut_b, sh, v, gf, lpl);
            const double c = S.bcoeff[edown + q];
            u_ += c * v;
            sg += c * gf;
          }
          gx += sg * dx; gy += sg * dy; gz += sg * dz;
        }
      }
    }
#pragma unroll
    for (int q = 0; q < 2; ++q) {
      const int I = r + 32 * q;
      if (q == 1 && S.natom <= 32) break;
      if (S.na > 0 && I < S.natom) {
        double dx = rx - at_xyz[3 * I], dy = ry - at_xyz[3 * I + 1], dz = rz - at_xyz[3 * I + 2];
        if (PBC) min_image_j(S, dx, dy, dz);
        double rr, ri;
        sqrt_rinv(dx * dx + dy * dy + dz * dz, rr, ri);
        if (rr < S.rcut_a) {
          const RadShared sh = rad_shared_ri<1>(rr, ri, ira);
          const double* qq = aq + ((size_t)edown * S.natom + I) * PQA_JQP;
          const MergedSums m = ka4 ? pade_merged<1, 4, false>(Da, qq, sh.p) : pade_merged<1, 3, false>(Da, qq, sh.p);
          u_ += sh.omp * m.S1;
          double sg = sh.c0 * m.S2;
          if (acusp) {
            double v, gf, lpl;
            rad_fn<1>(1, acp, aca, S.rcut_a, sh, v, gf, lpl);
            const double c = acoef[(I * S.na) * 2 + edown];
            u_ += c * v;
            sg += c * gf;
          }
          gx += sg * dx; gy += sg * dy; gz += sg * dz;
        }
      }
    }
  } else {
#pragma unroll 1
    for (int q = 0; q < 2; ++q) {
      const int j = (q ? S.nup : 0) + r;
      if (S.nb > 0 && r < (q ? S.ndn : S.nup) && j != e && j != excl) {
        double dx = rx - (q ? cx[1] : cx[0]), dy = ry - (q ? cy[1] : cy[0]), dz = rz - (q ? cz[1] : cz[0]);
        if (PBC) min_image_j(S, dx, dy, dz);
        const double rr = sqrt(dx * dx + dy * dy + dz * dz);
        if (rr < S.rcut_b) {
          const RadShared sh = rad_shared<1>(rr, irb);
          double sg = 0.0;
#pragma unroll 1
          for (int l = 0; l < S.nb; ++l) {
            double v, gf, lpl;
            rad_fn<1>(S.b_kind[l], S.b_param[l], S.b_aux[l], S.rcut_b, sh, v, gf, lpl);
            const double c = S.bcoeff[l * 3 + edown + q];
            u_ += c * v;
            sg += c * gf;
          }
          gx += sg * dx; gy += sg * dy; gz += sg * dz;
        }
      }
    }
#pragma unroll 1
    for (int q = 0; q < 2; ++q) {
      const int I = r + 32 * q;
      if (S.na > 0 && I < S.natom) {
        double dx = rx - at_xyz[3 * I], dy = ry - at_xyz[3 * I + 1], dz = rz - at_xyz[3 * I + 2];
        if (PBC) min_image_j(S, dx, dy, dz);
        const double rr = sqrt(dx * dx + dy * dy + dz * dz);
        if (rr < S.rcut_a) {
          const RadShared sh = rad_shared<1>(rr, ira);
          double sg = 0.0;
#pragma unroll 1
          for (int k = 0; k < S.na; ++k) {
            double v, gf, lpl;
            rad_fn<1>(S.a_kind[k], S.a_param[k], S.a_aux[k], S.rcut_a, sh, v, gf, lpl);
            const double c = acoef[(I * S.na + k) * 2 + edown];
            u_ += c * v;
            sg += c * gf;
          }
          gx += sg * dx; gy += sg * dy; gz += sg * dz;
        }
      }
    }
  }
  U = u_; g[0] = gx; g[1] = gy; g[2] = gz;
}

// ---- merged route, branch-free: the kernel runs two waves per SIMD, so a chain of dependent fp64 instructions runs at a quarter of
// the pipe's rate; what hides the latency is independent chains in ONE thread.  res_pair_m has no branch (masked pairs are evaluated
// at a harmless distance and discarded by a select), so the compiler interleaves the eight pairs res_jas_dual_m evaluates.
// K <= 4 Pade functions through the KD = 4 polynomials (records and denominators are zero padded); a basis without cusp function
// passes cusp parameter and coefficient 0.
#ifndef PQA_RES_JCHAIN
#define PQA_RES_JCHAIN 2
#endif
// Minimal image of a Jastrow pair from the block's LDS copy of the cell: where every Jastrow cut-off is at most the inradius of the
// cell-centred parallelepiped (PbcDev::jas_fold, flag at pbt[29]) the fold IS the minimal image inside the cut-off (min_image_j);
// otherwise the general reduction on the global tables.
__device__ __forceinline__ void res_min_image(const SysDev& S, const double* __restrict__ pbt, double& dx, double& dy, double& dz) {
  if (pbt[29] != 0.0) {
    double f0 = dx * pbt[0] + dy * pbt[3] + dz * pbt[6], f1 = dx * pbt[1] + dy * pbt[4] + dz * pbt[7], f2 = dx * pbt[2] + dy * pbt[5] + dz * pbt[8];
    f0 -= floor(f0 + 0.5); f1 -= floor(f1 + 0.5); f2 -= floor(f2 + 0.5);
    dx = f0 * pbt[9] + f1 * pbt[12] + f2 * pbt[15];
    dy = f0 * pbt[10] + f1 * pbt[13] + f2 * pbt[16];
    dz = f0 * pbt[11] + f1 * pbt[14] + f2 * pbt[17];
  } else min_image_j(S, dx, dy, dz);
}
struct ResJ { double u, x, y, z; };
template <bool UNI>
__device__ __forceinline__ void res_pair_m(bool valid, double dx, double dy, double dz, double rcut, double ircut, const double (&D)[5],
                                           const double* __restrict__ q, double cpar, double caux, double ccoef, ResJ& a) {
  double rr, ri;
  sqrt_rinv(dx * dx + dy * dy + dz * dz, rr, ri);
  const bool in = valid && rr < rcut;
  rr = in ? rr : 0.5 * rcut; ri = in ? ri : 2.0 * ircut;
  const RadShared sh = rad_shared_ri<1>(rr, ri, ircut);
  const MergedSums m = pade_merged<1, 4, UNI>(D, q, sh.p);
  double du = sh.omp * m.S1, sg = sh.c0 * m.S2;
  {
    double v, gf, lpl;
    rad_fn<1>(1, cpar, caux, rcut, sh, v, gf, lpl);
    du += ccoef * v;
    sg += ccoef * gf;
  }
  du = in ? du : 0.0; sg = in ? sg : 0.0;
  a.u += du; a.x += sg * dx; a.y += sg * dy; a.z += sg * dz;
}
// This thread's share of U, grad U of electron e at (px, py, pz), merged route: the two electron partners interleaved, then the ion(s).
// Every table comes from the block's LDS copy jt (electron-electron numerators [3][PQA_JQ], denominators b_D, a_D, cusp coefficients
// bcoeff[0][0..2]) and aq: a vector load from global memory in here makes the compiler wait for vmcnt(0), i.e. for the row / tape
// prefetches and the cache-row stores still in flight (3.7 us per evaluation instead of ~1).
#define PQA_RES_JT (3 * PQA_JQ + 24)
template <bool PBC>
__device__ __forceinline__ void res_jas_m(const SysDev& S, int r, const double (&cx)[2], const double (&cy)[2], const double (&cz)[2],
                                          const double* __restrict__ at_xyz, const double* __restrict__ acoef, const double* __restrict__ aq,
                                          const double* __restrict__ jt, int e, double px, double py, double pz, ResJ& j,
                                          const double* __restrict__ pbt = nullptr) {
  const int se = e >= S.nup;
  const double irb = jt[3 * PQA_JQ + 13], ira = jt[3 * PQA_JQ + 14];  // (loop-invariant VALU results would be hoisted and spilled)
  const bool bcusp = S.nb > 0 && S.b_kind[0] == 1, acusp = S.na > 0 && S.a_kind[0] == 1;
  const double bcp = bcusp ? S.b_param[0] : 0.0, bca = bcusp ? S.b_aux[0] : 0.0, acp = acusp ? S.a_param[0] : 0.0, aca = acusp ? S.a_aux[0] : 0.0;
  {
    double Db[5];
#pragma unroll
    for (int i = 0; i < 5; ++i) Db[i] = jt[3 * PQA_JQ + i];
#pragma unroll
    for (int q = 0; q < 2; ++q) {  // partner slot q: spin q
      const int jj = (q ? S.nup : 0) + r;
      double dx = px - cx[q], dy = py - cy[q], dz = pz - cz[q];
      if (PBC) res_min_image(S, pbt, dx, dy, dz);  // (distance.py:83-159; inside the cut-off the folded vector where the cell allows)
      res_pair_m<false>(S.nb > 0 && r < (q ? S.ndn : S.nup) && jj != e, dx, dy, dz, S.rcut_b, irb, Db,
                        jt + (se + q) * PQA_JQ, bcp, bca, jt[3 * PQA_JQ + 10 + se + q], j);
    }
  }
#if PQA_RES_JCHAIN < 3
  __builtin_amdgcn_sched_barrier(0);  // (three pairs interleaved need more registers than the kernel has to spare)
#endif
  {
    double Da[5];
#pragma unroll
    for (int i = 0; i < 5; ++i) Da[i] = jt[3 * PQA_JQ + 5 + i];
    for (int q = 0; q < (S.natom > 32 ? 2 : 1); ++q) {
      const int I = r + 32 * q, Ic = I < S.natom ? I : 0;
      double dx = px - at_xyz[3 * Ic], dy = py - at_xyz[3 * Ic + 1], dz = pz - at_xyz[3 * Ic + 2];
      if (PBC) res_min_image(S, pbt, dx, dy, dz);
      const double* qrec = aq + ((size_t)se * S.natom + Ic) * PQA_JQP;
      res_pair_m<false>(S.na > 0 && I < S.natom, dx, dy, dz, S.rcut_a, ira, Da, qrec, acp, aca, qrec[PQA_JQ], j);
    }
  }
}
template <int KWC>
__device__ __forceinline__ void res_combine(const double* __restrict__ pb, int PS, int cstride, double* __restrict__ rn_r) {
  double v[5][KWC];
#pragma unroll
  for (int c = 0; c < 5; ++c)
#pragma unroll
    for (int k = 0; k < KWC; ++k) v[c][k] = pb[(size_t)k * 16 * PS + c * cstride];
#pragma unroll
  for (int c = 0; c < 5; ++c) {
    double sum = v[c][0];
#pragma unroll
    for (int k = 1; k < KWC; ++k) sum += v[c][k];
    rn_r[c * 32] = sum;
  }
}

// The periodic instantiation's view of the cell, from the block's LDS copy (pbt / pbi, see the kernel): fold of point - atom into the
// cell-centred parallelepiped with the membership base of the pair (pbc_ctx_base + prim_wrap, pqa_ao.hpp), the fold alone, and the
// candidate images of a (point, atom) pair that are worth a distance test (k_pbc_prepass, step 1: the candidates near the sub-cell of
// the folded displacement, near_masks, that the reference's membership rule admits, member_masks).  Handles without the mask tables,
// with more than 128 candidates or more than PQA_RES_NCUT shell cut-offs per atom keep the launch-per-move sweep (pbc_lists_ok).
__device__ __forceinline__ void res_fold(const double* __restrict__ pbt, double x, double y, double z, double& x0, double& y0, double& z0,
                                         double& f0, double& f1, double& f2) {
  f0 = floor(x * pbt[0] + y * pbt[3] + z * pbt[6] + 0.5);
  f1 = floor(x * pbt[1] + y * pbt[4] + z * pbt[7] + 0.5);
  f2 = floor(x * pbt[2] + y * pbt[5] + z * pbt[8] + 0.5);
  x0 = x - (f0 * pbt[9] + f1 * pbt[12] + f2 * pbt[15]);
  y0 = y - (f0 * pbt[10] + f1 * pbt[13] + f2 * pbt[16]);
  z0 = z - (f0 * pbt[11] + f1 * pbt[14] + f2 * pbt[17]);
}
struct ResPair { double x0, y0, z0; int b0, b1, b2; double f0, f1, f2; };
__device__ __forceinline__ ResPair res_pair_base(const double* __restrict__ pbt, const int* __restrict__ pbi, int a, double px, double py, double pz,
                                                 double ax, double ay, double az) {
  ResPair c;
  double f0, f1, f2;
  res_fold(pbt, px - ax, py - ay, pz - az, c.x0, c.y0, c.z0, f0, f1, f2);
  c.f0 = f0; c.f1 = f1; c.f2 = f2;
  c.b0 = c.b1 = c.b2 = 0;
  if (pbi[12]) {
    const int w0 = (int)floor(px * pbt[18] + py * pbt[21] + pz * pbt[24]), w1 = (int)floor(px * pbt[19] + py * pbt[22] + pz * pbt[25]),
              w2 = (int)floor(px * pbt[20] + py * pbt[23] + pz * pbt[26]);
    const int i0 = (int)f0, i1 = (int)f1, i2 = (int)f2, M = pbi[9];
    c.b0 = pbi[14 + 3 * a] + i0 * pbi[0] + i1 * pbi[3] + i2 * pbi[6] - w0 + M;
    c.b1 = pbi[14 + 3 * a + 1] + i0 * pbi[1] + i1 * pbi[4] + i2 * pbi[7] - w1 + M;
    c.b2 = pbi[14 + 3 * a + 2] + i0 * pbi[2] + i1 * pbi[5] + i2 * pbi[8] - w2 + M;
  }
  return c;
}
__device__ __forceinline__ void res_image_masks(const double* __restrict__ pbt, const int* __restrict__ pbi, const ResPair& c, int a, int nimg,
                                                int mclass, unsigned long long& m0, unsigned long long& m1) {
  m0 = nimg >= 64 ? ~0ull : (1ull << nimg) - 1ull;
  m1 = nimg <= 64 ? 0ull : (nimg >= 128 ? ~0ull : (1ull << (nimg - 64)) - 1ull);
  const unsigned long long* near_mask = reinterpret_cast<const unsigned long long*>(reinterpret_cast<const unsigned long long*>(pbt + 27)[0]);
  const unsigned long long* memb_mask = reinterpret_cast<const unsigned long long*>(reinterpret_cast<const unsigned long long*>(pbt + 27)[1]);
  // (both table entries requested before either is used: two dependent round trips to L2 otherwise)
  const unsigned long long* nm = nullptr;
  const unsigned long long* mm = nullptr;
  bool outside = false;
  if (near_mask) {
    const int G = pbi[11];
    const double u0 = c.x0 * pbt[0] + c.y0 * pbt[3] + c.z0 * pbt[6];
    const double u1 = c.x0 * pbt[1] + c.y0 * pbt[4] + c.z0 * pbt[7];
    const double u2 = c.x0 * pbt[2] + c.y0 * pbt[5] + c.z0 * pbt[8];
    const int g0 = min(G - 1, max(0, (int)((u0 + 0.5) * G))), g1 = min(G - 1, max(0, (int)((u1 + 0.5) * G))),
              g2 = min(G - 1, max(0, (int)((u2 + 0.5) * G)));
    nm = near_mask + 2 * ((((size_t)a * G + g0) * G + g1) * G + g2);
  }
  if (pbi[12]) {
    const int side = 2 * pbi[9] + 1, E = pbi[10], Tm = side + 2 * E;
    const int i0 = c.b0 + E, i1 = c.b1 + E, i2 = c.b2 + E;
    if ((unsigned)i0 < (unsigned)Tm && (unsigned)i1 < (unsigned)Tm && (unsigned)i2 < (unsigned)Tm)
      mm = memb_mask + 2 * ((((size_t)mclass * Tm + i0) * Tm + i1) * Tm + i2);
    else outside = true;
  }
  const ulonglong2 vn = nm ? *reinterpret_cast<const ulonglong2*>(nm) : make_ulonglong2(~0ull, ~0ull);
  const ulonglong2 vm = mm ? *reinterpret_cast<const ulonglong2*>(mm) : make_ulonglong2(~0ull, ~0ull);
  m0 &= vn.x & vm.x; m1 &= vn.y & vm.y;
  if (outside) { m0 = 0ull; m1 = 0ull; }
}

#ifdef PQA_RES_CLK  // timing build only: 100 MHz stamps of thread 0 of the first blocks, last move of the sweep
static __device__ unsigned long long pqa_res_clk[64 * 16];
static __device__ unsigned long long pqa_res_clk3[64 * 8];
static __device__ unsigned long long pqa_res_clk2[64 * 16];  // thread 0's AO phase: cycles in [0] list header + zeroing, [1] fold, [2] walk + evaluation, [3] shells, [4] images evaluated
#define PQA_RCLK2(k, v) do { if (blockIdx.x < 64 && threadIdx.x == 0) pqa_res_clk2[blockIdx.x * 16 + (k)] = (v); } while (0)
#define PQA_RCLK(k) do { if (blockIdx.x < 64 && threadIdx.x == 0) pqa_res_clk[blockIdx.x * 16 + (k)] = wall_clock64(); } while (0)
#else
#define PQA_RCLK(k) do { } while (0)
#define PQA_RCLK2(k, v) do { } while (0)
#endif

// grid = ceil((w_hi - w_lo) / 16) blocks of 512 threads; dynamic LDS = RT.region doubles + res_lds_fixed(...).
// mb.gauss [N][W][3] and mb.unif [N][W] must be set (the caller draws them ahead from the Philox streams when there is no tape).
// Register budget: 256 per thread (two waves per SIMD).  What is carried across the orbital phase is the inverse row (64), the
// two coordinates (12) and a few indices; everything a proposal hands to its decision waits in LDS (wsc), the accumulators live
// only across the MFMA loop, and the loads a decision / the next proposal need are issued after the AO phase.
#ifndef PQA_RES_LB
#define PQA_RES_LB PQA_RES_NT
#endif
template <bool DMC, int LMAX, bool PBC = false, bool CX = false>
static __global__ __launch_bounds__(PQA_RES_LB) void k_sweep_res(SysDev S, LwState L, MoveBuf mb, ChunkTab T, ResTab RT, int has_jastrow,
                                                                  long W, long w_lo, long w_hi) {
  extern __shared__ double lds[];
  const int tid = threadIdx.x, lane = tid & 63, wv = __builtin_amdgcn_readfirstlane(tid >> 6);  // (wave-uniform: scalar)
  const int wl = tid >> 5, r = tid & 31;      // step phases: walker of the block, inverse row / partner slot
  const int pl = tid & 15, grp = tid >> 4;    // AO phase: point, lane group
  const int i16 = lane & 15, kq = lane >> 4;  // MFMA phase
  const int N = S.nelec, KT = RT.kt;
  double* region = lds;
  double* rowE = region + RT.region;           // [16][32] inverse row of the electron being moved
  double* wsc = rowE + 16 * 32;                // [16][16] per walker: 0..2 proposal, 3..5 scaled gaussians, 6..8 drift, 9 U at the old position,
                                               // 10..12 determinant sign / log / running |ratio| product, 13..14 r^2 sums (DMC), 15 accepted moves, 16..18 cells the
                                               // folded proposal crossed (periodic)
  double* pr_exp = wsc + 16 * PQA_RES_WS;      // [RT.nprim_u] distinct primitives (the 16 carbon atoms of a cell share 14, not 224)
  double* pr_coef = pr_exp + RT.nprim_u;
  double* at_xyz = pr_coef + RT.nprim_u;
  double* acoef = at_xyz + 3 * (size_t)S.natom;
  double* aql = acoef + 2 * (size_t)S.natom * (S.na > 0 ? S.na : 1);          // merged Pade numerators per (ion, spin)
  double* jt = aql + 2 * (size_t)S.natom * PQA_JQP;                            // electron-electron Jastrow tables (res_jas_m)
  int* sh_meta = (int*)(jt + PQA_RES_JT);  // l, primitives, first primitive, padded row, atom
  int* glist = sh_meta + 5 * (size_t)S.nshell;
  int* goff = glist + RT.nlist;
  int* occ = goff + RT.npass * 32 + 1;
  // periodic: behind everything else (res_lds_pbc; 8-byte aligned: the int block above holds an even number of entries or is padded by the host)
  double* LsL = lds + (RT.pbc_off >> 3);
  double* sh_cut = LsL + 3 * (PBC ? S.nL : 0);                     // [nshell] shell cut-offs
  double* at_cut = sh_cut + (PBC ? S.nshell : 0);                  // [natom][1 + PQA_RES_NCUT]: atom cut-off, class cut-offs (ascending)
  double* pbt = at_cut + (PBC ? S.natom * (1 + PQA_RES_NCUT) : 0);   // [36]: linv, lat, lprim_inv, the addresses of near_mask / memb_mask, jas_fold, ktl, twist
  double* phL = pbt + (PBC ? 36 : 0);                                // twisted: [nL][2] (cos, sin)(k_t . Ls[j])
  double* pcs = phL + ((PBC && CX && RT.twist) ? 2 * S.nL : 0);      // twisted: [natom][16][2] fold phase of (point, atom)
  int* at_int = reinterpret_cast<int*>(pcs + ((PBC && CX && RT.twist) ? 32 * S.natom : 0));  // [natom][3]: candidates, classes, membership class
  int* pbi = at_int + (PBC ? 3 * S.natom : 0);                       // [14 + 3 natom]: supercell[9], M, E, G, has_member, jas_fold, atom_n
  unsigned char* imgl = reinterpret_cast<unsigned char*>(pbi + ((PBC ? 3 * S.natom + 16 : 0) & ~1));
  unsigned char* imgn = imgl + (size_t)S.natom * 16 * RT.icap;
  double* ws = wsc + wl * PQA_RES_WS;

  const long wraw = w_lo + (long)blockIdx.x * PQA_RES_NW + wl;
  const bool live = wraw < w_hi;
  const long wg = live ? wraw : w_hi - 1;  // walkers past the end shadow the last one and store nothing

  // ---- tables and the walker's coordinates
  for (int sh = tid; sh < S.nshell; sh += PQA_RES_NT) {
    const int ia = S.shell_atom[sh];
    sh_meta[5 * sh] = S.shell_l[sh];
    sh_meta[5 * sh + 1] = S.shell_prim_off[sh + 1] - S.shell_prim_off[sh];
    sh_meta[5 * sh + 2] = RT.shell_q0[sh];
    sh_meta[5 * sh + 3] = RT.shell_row[sh];
    sh_meta[5 * sh + 4] = ia;
  }
  for (int p = tid; p < RT.nprim_u; p += PQA_RES_NT) { pr_exp[p] = RT.prim_exp_u[p]; pr_coef[p] = RT.prim_coef_u[p]; }
  for (int k = tid; k < 3 * S.natom; k += PQA_RES_NT) at_xyz[k] = S.atom_xyz[k];
  for (int k = tid; k < 2 * S.natom * S.na; k += PQA_RES_NT) acoef[k] = has_jastrow ? S.acoeff[k] : 0.0;
  for (int k = tid; k < 2 * S.natom * PQA_JQP; k += PQA_RES_NT) {  // [spin][ion][PQA_JQP] from the global [ion][spin][PQA_JQ]; slot 24: cusp coefficient
    const int j = k % PQA_JQP, I = (k / PQA_JQP) % S.natom, sp = k / (PQA_JQP * S.natom);
    double v = 0.0;
    if (has_jastrow && S.jq_on && S.na > 0) v = j < PQA_JQ ? S.aq[(size_t)(I * 2 + sp) * PQA_JQ + j] : (S.a_kind[0] == 1 ? S.acoeff[(I * S.na) * 2 + sp] : 0.0);
    aql[k] = v;
  }
  for (int k = tid; k < PQA_RES_JT; k += PQA_RES_NT) {
    double v = 0.0;
    if (has_jastrow && S.jq_on) {
      if (k < 3 * PQA_JQ) v = S.nb > 0 ? S.bq[k] : 0.0;
      else if (k < 3 * PQA_JQ + 5) v = S.b_D[k - 3 * PQA_JQ];
      else if (k < 3 * PQA_JQ + 10) v = S.a_D[k - 3 * PQA_JQ - 5];
      else if (k < 3 * PQA_JQ + 13) v = (S.nb > 0 && S.b_kind[0] == 1) ? S.bcoeff[k - 3 * PQA_JQ - 10] : 0.0;
    }
    if (k == 3 * PQA_JQ + 13) v = 1.0 / S.rcut_b;
    if (k == 3 * PQA_JQ + 14) v = 1.0 / S.rcut_a;
    if (k == 3 * PQA_JQ + 15) v = sqrt(mb.tstep);
    if (k == 3 * PQA_JQ + 16) v = 1.0 / (2.0 * mb.tstep);
    jt[k] = v;
  }
  for (int k = tid; k < RT.nlist; k += PQA_RES_NT) glist[k] = RT.grp_shell[k];
  for (int k = tid; k < RT.npass * 32 + 1; k += PQA_RES_NT) goff[k] = RT.grp_off[k];
  for (int k = tid; k < 64; k += PQA_RES_NT) {
    const int s = k >> 5, q = k & 31, n = s ? S.ndn : S.nup;
    occ[k] = q < n ? (s ? S.det_occ[1][q] : S.det_occ[0][q]) : 0;
  }
  if (PBC) {  // (no vector load from global memory inside the AO phase: each one waits for everything in flight, and the per-atom /
              // per-shell table look-ups were dependent round trips — 12 of the phase's 38 us)
    for (int k = tid; k < 3 * S.nL; k += PQA_RES_NT) LsL[k] = S.pb->Ls[k];
    for (int k = tid; k < S.nshell; k += PQA_RES_NT) sh_cut[k] = S.pb->shell_cut[k];
    if (CX && RT.twist) for (int k = tid; k < 2 * S.nL; k += PQA_RES_NT) phL[k] = S.pb->img_phase[k];
    for (int k = tid; k < S.natom; k += PQA_RES_NT) {
      const int ncl = S.pb->ncls[k];
      at_cut[k * (1 + PQA_RES_NCUT)] = S.pb->atom_cut[k];
      for (int q = 0; q < PQA_RES_NCUT; ++q) at_cut[k * (1 + PQA_RES_NCUT) + 1 + q] = q < ncl ? S.pb->cls_cut[k * PQA_MAXCLS + q] : INFINITY;
      at_int[3 * k] = S.pb->num_Ls[k]; at_int[3 * k + 1] = ncl; at_int[3 * k + 2] = S.pb->member ? S.pb->member_class[k] : 0;
      for (int q = 0; q < 3; ++q) pbi[14 + 3 * k + q] = S.pb->member ? S.pb->atom_n[3 * k + q] : 0;
    }
    // (a load from S.pb inside the electron loop is a VECTOR load — after the loop's stores the compiler cannot keep it scalar — and
    // waits for everything in flight: the cell and the membership rule's scalars from LDS instead)
    if (tid < 9) { pbt[tid] = S.pb->linv[tid]; pbt[9 + tid] = S.pb->lat[tid]; pbt[18 + tid] = S.pb->lprim_inv[tid]; pbi[tid] = S.pb->supercell[tid]; }
    if (tid == 9) {
      pbi[9] = S.pb->member_M; pbi[10] = S.pb->memb_E; pbi[11] = S.pb->near_G; pbi[12] = S.pb->member != nullptr; pbi[13] = S.pb->jas_fold;
      reinterpret_cast<unsigned long long*>(pbt + 27)[0] = (unsigned long long)S.pb->near_mask;
      reinterpret_cast<unsigned long long*>(pbt + 27)[1] = (unsigned long long)S.pb->memb_mask;
      pbt[29] = (S.pbc == 1 || S.pb->jas_fold) ? 1.0 : 0.0;  // (pbc 1: orthogonal cell, the fold is the minimal image)
      pbt[30] = S.pb->ktl[0]; pbt[31] = S.pb->ktl[1]; pbt[32] = S.pb->ktl[2]; pbt[33] = S.pb->twist ? 1.0 : 0.0;
    }
  }
  for (int k = tid; k < RT.region; k += PQA_RES_NT) region[k] = 0.0;  // (K-padding rows of the tile stay finite)
  double cx[2], cy[2], cz[2];
#pragma unroll
  for (int q = 0; q < 2; ++q) {  // slot 0: up electron r, slot 1: down electron nup + r
    const int j = (q ? S.nup : 0) + r;
    const double* xj = L.xt + (size_t)((r < (q ? S.ndn : S.nup)) ? j : 0) * 3 * W + wg;
    cx[q] = xj[0]; cy[q] = xj[W]; cz[q] = xj[2 * W];
  }
  if (r == 0) { ws[13] = 0.0; ws[14] = 0.0; ws[15] = 0.0; }  // r^2 sums of the proposals / the accepted ones (DMC), accepted moves
  __syncthreads();

#pragma unroll 1
  for (int s = 0; s < 2; ++s) {
    const int n = s ? S.ndn : S.nup;
    if (n == 0) continue;
    const int nmo = s ? S.nmo[1] : S.nmo[0], ldc = s ? T.ldc[1] : T.ldc[0], nt = ldc >> 4, e0 = s ? S.nup : 0;
    const double* __restrict__ cpad = s ? T.cpad[1] : T.cpad[0];
    double* Tg = (s ? L.Tt[1] : L.Tt[0]) + wg;
    double* rcs = s ? L.rc[1] : L.rc[0];
    uint8_t* sels = s ? L.sel[1] : L.sel[0];
    const int KW = 8 / nt, u = wv % nt, kw = wv / nt;  // MFMA roles: orbital tile u, K-split part kw of KW
    const int PS = res_ps(nt);
    double* part = region + RT.part_off;        // [KW][16][PS]
    double* rn = part + (size_t)KW * 16 * PS + (size_t)wl * PQA_RES_RS;  // this walker's combined rows [5][32]
    const int* occs = occ + 32 * s;
    const bool ident = (s ? S.occ_ident[1] : S.occ_ident[0]) != 0;
    const int oc = occs[r];
    const int ncol = CX ? 2 * n : n;  // doubles per inverse row (complex: (re, im) pairs — 16 electrons of a spin at most)
    const int nh = CX ? nmo / 2 : nmo;  // orbitals (complex: the rows are [re block | im block])
    // ---- the transposed inverse of this spin: row r of walker wl
    // (all 32 loads in flight at once — clamped addresses, the entries outside the block multiplied by zero: with a select per element the
    // compiler branches around each load, and the opaque copy used against that until round 6 made it wait for every load before it issued
    // the next: 32 dependent HBM round trips per spin)
    double t[32];
    {
      const double* trow = Tg + (size_t)(r < n ? r : 0) * ncol * W;
      const double rm = r < n ? 1.0 : 0.0;
#pragma unroll
      for (int k = 0; k < 32; ++k) t[k] = trow[(size_t)(k < ncol ? k : 0) * W];
#pragma unroll
      for (int k = 0; k < 32; ++k) t[k] *= (k < ncol) ? rm : 0.0;
    }
    int selr = (r < n) ? (int)sels[(size_t)r * W + wg] : 0;  // slot of electron r's cached row
    if (r == 0) {  // per-walker scalars of this spin's sweep live in LDS (wsc 10..12: sign, log, running product of |ratio|)
      const double* dsg = s ? L.dsign[1] : L.dsign[0];
      if (CX) { ws[10] = dsg[2 * wg]; ws[19] = dsg[2 * wg + 1]; } else ws[10] = dsg[wg];  // (complex: the determinant's phase)
      ws[11] = (s ? L.dlog[1] : L.dlog[0])[wg]; ws[12] = 1.0;
    }

#pragma unroll 1
    for (int i = -1; i < n; ++i) {  // iteration i: [orbitals at the proposals of electron i, decide i], then propose i + 1
      const int e = e0 + i;
      // Everything derived from the thread index is re-derived here from an OPAQUE copy: the loop body is ~10 000 instructions, and
      // whatever the compiler proves loop-invariant (hundreds of address and mask values) it hoists in front of the loop, keeps live
      // across it and spills — together with a good part of the inverse row.  A dozen integer instructions per move instead.
      int tido = tid;
      asm volatile("" : "+v"(tido));
      const int lane = tido & 63, wl = tido >> 5, r = tido & 31, pl = tido & 15, grp = tido >> 4, i16 = lane & 15, kq = lane >> 4;
      const long wraw = w_lo + (long)blockIdx.x * PQA_RES_NW + wl;
      const bool live = wraw < w_hi;
      const long wg = live ? wraw : w_hi - 1;
      double* ws = wsc + wl * PQA_RES_WS;
      double* rn = part + (size_t)KW * 16 * PS + (size_t)wl * PQA_RES_RS;
      const int oc = occs[r];
      double ro[CX ? 8 : 4], uacc = 0.0, g0 = 0.0, g1 = 0.0, g2 = 0.0;
#pragma unroll
      for (int q = 0; q < (CX ? 8 : 4); ++q) ro[q] = 0.0;
      double p0 = 1.0, p1 = 0.0, p2 = 0.0, p3 = 0.0;  // Slater sums at the proposal of electron i
      double q0i = 0.0, q1i = 0.0, q2i = 0.0, q3i = 0.0;  // (complex: their imaginary parts)
      // loads the decision / the next proposal need (cached row of electron i + 1, tape entries): issued behind the AO phase, used
      // after the contraction
      auto prefetch = [&]() {
        if (i + 1 < n) {
          const int slot = __shfl(selr, (lane & 32) | (i + 1), 64);
          const double* row = rcs + (((size_t)(i + 1) * 2 + slot) * W + wg) * 5 * nmo;
          if (r < n) {
            ro[0] = row[oc]; ro[1] = row[nmo + oc]; ro[2] = row[2 * nmo + oc]; ro[3] = row[3 * nmo + oc];
            if (CX) { ro[4] = row[nh + oc]; ro[5] = row[nmo + nh + oc]; ro[6] = row[2 * nmo + nh + oc]; ro[7] = row[3 * nmo + nh + oc]; }
          }
          const double* zt = mb.gauss + ((size_t)(e + 1) * W + wg) * 3;
          g0 = zt[0]; g1 = zt[1]; g2 = zt[2];
        }
        if (i >= 0) uacc = mb.unif[(size_t)e * W + wg];
      };
      if (i < 0) prefetch();
      if (i >= 0) {
        PQA_RCLK(0);
        res_block_sync();  // proposals of all 16 walkers are in wsc; the previous move's reads of the region are done
        // ================= orbital rows at the 16 proposals
        // (opaque copies: the addresses of the unrolled contraction below depend on them, so the compiler cannot hoist the ~100
        // loop-invariant address values out of the electron loop — it did, and spilled them and a third of the inverse row)
        int kwv = kw, ktv = KT;
        asm volatile("" : "+s"(kwv), "+s"(ktv));
        if (PBC) {
          // ---- images of every (point, atom) pair the reference's rule admits inside the atom's cut-off (numba/pbcgto.py:565-604; the
          // work of k_pbc_prepass, in the block): candidates from the pre-tabulated near / membership masks, a distance test each,
          // indices into the pair's LDS list.  Pairs the lists cannot hold (no mask table, more images than a list holds) are flagged 255 and
          // take the direct tests of shell_eval_pbc; 254: the list is too short, the pair's shells walk the candidate masks themselves.
          const int po = (CX && RT.twist) ? 16 : 0;  // (twisted: the folded copy of the proposal)
          const double ppx = wsc[pl * PQA_RES_WS + po], ppy = wsc[pl * PQA_RES_WS + po + 1], ppz = wsc[pl * PQA_RES_WS + po + 2];
          int NS = 1;  // threads per (point, atom) pair: the largest power of two with natom NS <= 32
          while (2 * NS * S.natom <= 32) NS *= 2;
          if (NS > 1) {
            // Few atoms: the pair's candidates dealt to NS threads (image j to thread j mod NS), each with its own class populations;
            // the populations meet in LDS (the tile region is free until phase 1 zeroes its rows; the values are finite as doubles, so rows of
            // K padding they land on still contract to zero) and every thread writes its entries at its own offsets — class by class, inside a
            // class slice by slice.  One thread per pair was 11 of the move's 50 us in the 8-atom cell (8 of 32 lane groups busy).
            unsigned long long* pcnt = reinterpret_cast<unsigned long long*>(region);  // [natom][16][NS], 8-bit fields
#ifdef PQA_RES_CLK
            const unsigned long long tA = clock64(); unsigned long long tB = tA, tC = tA, tD = tA, tE = tA;
#endif
            const int a = grp % S.natom, q = grp / S.natom;
            const bool on = q < NS;
            unsigned long long k0 = 0ull, k1 = 0ull;
            double cx0 = 0.0, cy0 = 0.0, cz0 = 0.0;
            if (on) {
              const ResPair c = res_pair_base(pbt, pbi, a, ppx, ppy, ppz, at_xyz[3 * a], at_xyz[3 * a + 1], at_xyz[3 * a + 2]);
              cx0 = c.x0; cy0 = c.y0; cz0 = c.z0;
              if (CX && RT.twist && q == 0) {
                double sf_, cf_;
                sincos(c.f0 * pbt[30] + c.f1 * pbt[31] + c.f2 * pbt[32], &sf_, &cf_);
                pcs[(a * 16 + pl) * 2] = cf_; pcs[(a * 16 + pl) * 2 + 1] = sf_;
              }
              unsigned long long m0 = 0ull, m1 = 0ull;
              const int ncl = at_int[3 * a + 1];
              res_image_masks(pbt, pbi, c, a, at_int[3 * a], at_int[3 * a + 2], m0, m1);
#ifdef PQA_RES_CLK
              asm volatile("" : "+v"(m0), "+v"(m1)); tB = clock64();
#endif
              double cut_r[PQA_RES_NCUT];
#pragma unroll
              for (int k = 0; k < PQA_RES_NCUT; ++k) cut_r[k] = at_cut[a * (1 + PQA_RES_NCUT) + 1 + k];
              const double acut = at_cut[a * (1 + PQA_RES_NCUT)];
              unsigned long long cnt = 0ull;
              // bits j = q mod NS (NS divides 64: the same stripe in both words)
              const unsigned long long stripe = (NS == 2 ? 0x5555555555555555ull : NS == 4 ? 0x1111111111111111ull : NS == 8 ? 0x0101010101010101ull
                                                 : NS == 16 ? 0x0001000100010001ull : 0x0000000100000001ull) << q;
#pragma unroll 1
              for (int half = 0; half < 2; ++half) {
                unsigned long long m = (half ? m1 : m0) & stripe, keep = 0ull;
                while (m) {
                  const int b = __ffsll((long long)m) - 1, j = 64 * half + b;
                  m &= m - 1;
                  const double xj = c.x0 - LsL[3 * j], yj = c.y0 - LsL[3 * j + 1], zj = c.z0 - LsL[3 * j + 2];
                  const double r2 = xj * xj + yj * yj + zj * zj;
                  int cls = 0;
#pragma unroll
                  for (int k = 0; k < PQA_RES_NCUT; ++k) cls += r2 > cut_r[k] ? 1 : 0;
                  if (r2 > acut || cls >= ncl) continue;
                  cnt += 1ull << (8 * cls);
                  keep |= 1ull << b;
                }
                if (half) k1 = keep; else k0 = keep;
              }
              pcnt[((size_t)a * 16 + pl) * NS + q] = cnt;
            }
#ifdef PQA_RES_CLK
            tC = clock64();
#endif
            res_block_sync();
#ifdef PQA_RES_CLK
            tD = clock64();
#endif
            if (on) {
              const unsigned long long* pc = pcnt + ((size_t)a * 16 + pl) * NS;
              int n = 0;
              unsigned long long off = 0ull;
              {
                int tc[PQA_RES_NCUT], bq[PQA_RES_NCUT];
#pragma unroll
                for (int k = 0; k < PQA_RES_NCUT; ++k) { tc[k] = 0; bq[k] = 0; }
                for (int q2 = 0; q2 < NS; ++q2) {  // (one LDS read per thread of the pair)
                  const unsigned long long v = pc[q2];
#pragma unroll
                  for (int k = 0; k < PQA_RES_NCUT; ++k) {
                    const int f = (int)((v >> (8 * k)) & 255);
                    bq[k] += q2 < q ? f : 0;
                    tc[k] += f;
                  }
                }
                int run = 0;
#pragma unroll
                for (int k = 0; k < PQA_RES_NCUT; ++k) {
                  off |= (unsigned long long)((run + bq[k]) & 255) << (8 * k);
                  run += tc[k];
                }
                n = run;
              }
              if (n > RT.icap) n = 254;
              else {
                double cut_r[PQA_RES_NCUT];
#pragma unroll
                for (int k = 0; k < PQA_RES_NCUT; ++k) cut_r[k] = at_cut[a * (1 + PQA_RES_NCUT) + 1 + k];
                unsigned char* lst = imgl + ((size_t)a * 16 + pl) * RT.icap;
#pragma unroll 1
                for (int half = 0; half < 2; ++half) {
                  unsigned long long m = half ? k1 : k0;
                  while (m) {
                    const int j = 64 * half + __ffsll((long long)m) - 1;
                    m &= m - 1;
                    const double xj = cx0 - LsL[3 * j], yj = cy0 - LsL[3 * j + 1], zj = cz0 - LsL[3 * j + 2];
                    const double r2 = xj * xj + yj * yj + zj * zj;
                    int cls = 0;
#pragma unroll
                    for (int k = 0; k < PQA_RES_NCUT; ++k) cls += r2 > cut_r[k] ? 1 : 0;
                    const int pos = (int)((off >> (8 * cls)) & 255);
                    off += 1ull << (8 * cls);
                    lst[pos] = (unsigned char)j;
                  }
                }
              }
              if (q == 0) imgn[a * 16 + pl] = (unsigned char)n;
            }
#ifdef PQA_RES_CLK
            tE = clock64();
            PQA_RCLK2(5, tB - tA); PQA_RCLK2(6, tC - tB); PQA_RCLK2(7, tD - tC); PQA_RCLK2(8, tE - tD);
#endif
          } else
          for (int a = grp; a < S.natom; a += 32) {
            const ResPair c = res_pair_base(pbt, pbi, a, ppx, ppy, ppz, at_xyz[3 * a], at_xyz[3 * a + 1], at_xyz[3 * a + 2]);
            if (CX && RT.twist) {  // exp(i k_t . f . lattice) of the fold of point - atom (pbc_ctx_base)
              double sf_, cf_;
              sincos(c.f0 * pbt[30] + c.f1 * pbt[31] + c.f2 * pbt[32], &sf_, &cf_);
              pcs[(a * 16 + pl) * 2] = cf_; pcs[(a * 16 + pl) * 2 + 1] = sf_;
            }
            int n = 0;
            unsigned long long m0 = 0ull, m1 = 0ull;
            const int ncl = at_int[3 * a + 1];
            res_image_masks(pbt, pbi, c, a, at_int[3 * a], at_int[3 * a + 2], m0, m1);
            {
              // Every admitted image gets the class of the smallest shell cut-off of this atom that contains it (at most PQA_RES_NCUT
              // distinct cut-offs here) and the list is written class by class — a counting sort in two walks over the candidate bits — so that a
              // shell's walk ends at the first image outside ITS cut-off: the lanes of a wave run the union of their walks, and unsorted
              // lists made every shell walk every image of the atom (158 us per move instead of ~15).
              double cut_r[PQA_RES_NCUT];
#pragma unroll
              for (int q = 0; q < PQA_RES_NCUT; ++q) cut_r[q] = at_cut[a * (1 + PQA_RES_NCUT) + 1 + q];
              const double acut = at_cut[a * (1 + PQA_RES_NCUT)];
              unsigned long long cnt = 0ull;  // 6-bit class populations
#pragma unroll 1
              for (int half = 0; half < 2; ++half) {
                unsigned long long m = half ? m1 : m0, keep = 0ull;
                while (m) {
                  const int b = __ffsll((long long)m) - 1, j = 64 * half + b;
                  m &= m - 1;
                  const double xj = c.x0 - LsL[3 * j], yj = c.y0 - LsL[3 * j + 1], zj = c.z0 - LsL[3 * j + 2];
                  const double r2 = xj * xj + yj * yj + zj * zj;
                  int cls = 0;
#pragma unroll
                  for (int q = 0; q < PQA_RES_NCUT; ++q) cls += r2 > cut_r[q] ? 1 : 0;
                  if (r2 > acut || cls >= ncl) continue;
                  cnt += 1ull << (6 * cls);
                  keep |= 1ull << b;
                  ++n;
                }
                if (half) m1 = keep; else m0 = keep;  // (the second walk visits the admitted ones only)
              }
              if (n > RT.icap) n = 254;  // more than the list holds: the shells of this pair walk the candidate masks themselves
              else {
                unsigned long long off = 0ull;
                int run = 0;
#pragma unroll
                for (int q = 0; q < PQA_RES_NCUT; ++q) { off |= (unsigned long long)run << (6 * q); run += (int)((cnt >> (6 * q)) & 63); }
                unsigned char* lst = imgl + ((size_t)a * 16 + pl) * RT.icap;
#pragma unroll 1
                for (int half = 0; half < 2; ++half) {
                  unsigned long long m = half ? m1 : m0;
                  while (m) {
                    const int j = 64 * half + __ffsll((long long)m) - 1;
                    m &= m - 1;
                    const double xj = c.x0 - LsL[3 * j], yj = c.y0 - LsL[3 * j + 1], zj = c.z0 - LsL[3 * j + 2];
                    const double r2 = xj * xj + yj * yj + zj * zj;
                    int cls = 0;
#pragma unroll
                    for (int q = 0; q < PQA_RES_NCUT; ++q) cls += r2 > cut_r[q] ? 1 : 0;
                    const int pos = (int)((off >> (6 * cls)) & 63);
                    off += 1ull << (6 * cls);
                    lst[pos] = (unsigned char)j;
                  }
                }
              }
            }
            imgn[a * 16 + pl] = (unsigned char)n;
          }
          res_block_sync();
          PQA_RCLK(14);
#ifdef PQA_RES_CLK
          if (blockIdx.x < 64 && threadIdx.x == 0) { int c255 = 0, tot = 0; for (int q = 0; q < S.natom * 16; ++q) { c255 += imgn[q] == 255; tot += imgn[q] == 255 ? 0 : imgn[q]; } pqa_res_clk[blockIdx.x * 16 + 15] = ((unsigned long long)c255 << 32) | (unsigned)tot; }
#endif
        }
        for (int ps = 0; ps < RT.npass; ++ps) {
          const int row_base = RT.pass_row0[ps], nks = (RT.pass_row0[ps + 1] - row_base) >> 2;
          if (ps > 0) res_block_sync();  // the previous pass's MFMA reads of the tile are done
          {
            const int po1 = (PBC && CX && RT.twist) ? 16 : 0;
            const double px = wsc[pl * PQA_RES_WS + po1], py = wsc[pl * PQA_RES_WS + po1 + 1], pz = wsc[pl * PQA_RES_WS + po1 + 2];
#ifndef PQA_RES_ABL_NOAO
#ifdef PQA_RES_CLK
            unsigned long long c_a = 0, c_b = 0, c_c = 0, n_sh = 0, n_im = 0;
#endif
            for (int it = goff[ps * 32 + grp]; it < goff[ps * 32 + grp + 1]; ++it) {
#ifdef PQA_RES_CLK
              const unsigned long long t_0 = clock64();
#endif
              const int sh = glist[it];
              const int l_ = sh_meta[5 * sh], np_ = sh_meta[5 * sh + 1], q0 = sh_meta[5 * sh + 2], krow = sh_meta[5 * sh + 3] - row_base;
              if (PBC) {
                // lattice sum over the admitted images inside this shell's cut-off (numba/pbcgto.py:99-506): the tile element belongs
                // to this thread, the sum accumulates in place
                const int a_ = sh_meta[5 * sh + 4], nim = imgn[a_ * 16 + pl];
                {
                  double* tl = region + (size_t)krow * 16 + pl;
                  // (the lattice sum accumulates in the tile: 15-25 running sums in registers beside the inverse row end up in scratch —
                  // tried, 39 k -> 62 k cycles for thread 0's three shells)
                  const bool tw = CX && RT.twist;
                  double* tli = tl + (size_t)RT.im_off * 16;  // twisted: the imaginary rows of the shell
#pragma unroll
                  for (int m = 0; m < 2 * LMAX + 1; ++m)
                    if (m < 2 * l_ + 1) {
                      tl[(size_t)m * 16] = 0.0; tl[((size_t)KT + m) * 16] = 0.0; tl[((size_t)2 * KT + m) * 16] = 0.0; tl[((size_t)3 * KT + m) * 16] = 0.0;
                      tl[((size_t)4 * KT + m) * 16] = 0.0;
                      if (tw) {
                        tli[(size_t)m * 16] = 0.0; tli[((size_t)KT + m) * 16] = 0.0; tli[((size_t)2 * KT + m) * 16] = 0.0; tli[((size_t)3 * KT + m) * 16] = 0.0;
                        tli[((size_t)4 * KT + m) * 16] = 0.0;
                      }
                    }
                  // (the five component planes as restrict pointers: the compiler could not tell that tl + c KT 16 are different addresses and
                  // ran the 5 (2 l + 1) read-add-write sequences of an image one after the other)
                  double* __restrict__ pl0 = tl;
                  double* __restrict__ pl1 = tl + (size_t)KT * 16;
                  double* __restrict__ pl2 = tl + (size_t)2 * KT * 16;
                  double* __restrict__ pl3 = tl + (size_t)3 * KT * 16;
                  double* __restrict__ pl4 = tl + (size_t)4 * KT * 16;
                  double* __restrict__ qi0 = tli;
                  double* __restrict__ qi1 = tli + (size_t)KT * 16;
                  double* __restrict__ qi2 = tli + (size_t)2 * KT * 16;
                  double* __restrict__ qi3 = tli + (size_t)3 * KT * 16;
                  double* __restrict__ qi4 = tli + (size_t)4 * KT * 16;
                  const double cf_ = tw ? pcs[(a_ * 16 + pl) * 2] : 1.0, sf_ = tw ? pcs[(a_ * 16 + pl) * 2 + 1] : 0.0;
                  auto add_image = [&](double xj, double yj, double zj, int j) __attribute__((always_inline)) {
                    // twisted: exp(i k_t . (f . lattice + Ls[j])) weights the image (shell_eval_pbc, TW)
                    const double cj = tw ? phL[2 * j] : 1.0, sj = tw ? phL[2 * j + 1] : 0.0;
                    const double wr = cf_ * cj - sf_ * sj, wi = sf_ * cj + cf_ * sj;
                    shell_eval<5, LMAX, true>(l_, xj, yj, zj, pr_exp + q0, pr_coef + q0, np_,
                                              [&](int m, double v, double ax, double ay, double az, double lp) __attribute__((always_inline)) {
                                                // ds_add_f64 without return value: one LDS operation per element instead of a dependent read / add / write; every
                                                // address has ONE owner thread, so the order of the sum stays the program's
                                                if (tw) {
                                                  res_lds_add(qi0 + m * 16, wi * v); res_lds_add(qi1 + m * 16, wi * ax); res_lds_add(qi2 + m * 16, wi * ay);
                                                  res_lds_add(qi3 + m * 16, wi * az); res_lds_add(qi4 + m * 16, wi * lp);
                                                  res_lds_add(pl0 + m * 16, wr * v); res_lds_add(pl1 + m * 16, wr * ax); res_lds_add(pl2 + m * 16, wr * ay);
                                                  res_lds_add(pl3 + m * 16, wr * az); res_lds_add(pl4 + m * 16, wr * lp);
                                                } else {
                                                  res_lds_add(pl0 + m * 16, v); res_lds_add(pl1 + m * 16, ax); res_lds_add(pl2 + m * 16, ay);
                                                  res_lds_add(pl3 + m * 16, az); res_lds_add(pl4 + m * 16, lp);
                                                }
                                              });
                  };
#ifdef PQA_RES_CLK
                  const unsigned long long t_1 = clock64();
#endif
                  double x0, y0, z0, f0_, f1_, f2_;
                  res_fold(pbt, px - at_xyz[3 * a_], py - at_xyz[3 * a_ + 1], pz - at_xyz[3 * a_ + 2], x0, y0, z0, f0_, f1_, f2_);
                  const double scut = sh_cut[sh];
#ifdef PQA_RES_CLK
                  const unsigned long long t_2 = clock64();
#endif
                  const unsigned char* lst = imgl + ((size_t)a_ * 16 + pl) * RT.icap;
                  if (nim == 254) {  // (rare: same images in index order, found again from the masks)
                    const ResPair c2 = res_pair_base(pbt, pbi, a_, px, py, pz, at_xyz[3 * a_], at_xyz[3 * a_ + 1], at_xyz[3 * a_ + 2]);
                    unsigned long long m0 = 0ull, m1 = 0ull;
                    res_image_masks(pbt, pbi, c2, a_, at_int[3 * a_], at_int[3 * a_ + 2], m0, m1);
                    const double cut2 = fmin(scut, at_cut[a_ * (1 + PQA_RES_NCUT)]);
#pragma unroll 1
                    for (int half = 0; half < 2; ++half) {
                      unsigned long long m = half ? m1 : m0;
                      while (m) {
                        const int j = 64 * half + __ffsll((long long)m) - 1;
                        m &= m - 1;
                        const double xj = x0 - LsL[3 * j], yj = y0 - LsL[3 * j + 1], zj = z0 - LsL[3 * j + 2];
                        if (xj * xj + yj * yj + zj * zj <= cut2) add_image(xj, yj, zj, j);
                      }
                    }
                    continue;
                  }
#pragma unroll 1
                  for (int k = 0; k < nim; ++k) {
                    const int j = lst[k];
                    const double xj = x0 - LsL[3 * j], yj = y0 - LsL[3 * j + 1], zj = z0 - LsL[3 * j + 2];
                    if (xj * xj + yj * yj + zj * zj > scut) break;  // (class-ordered list: nothing further is inside this shell's cut-off)
                    add_image(xj, yj, zj, j);
#ifdef PQA_RES_CLK
                    ++n_im;
#endif
                  }
#ifdef PQA_RES_CLK
                  { const unsigned long long t_3 = clock64(); c_a += t_1 - t_0; c_b += t_2 - t_1; c_c += t_3 - t_2; ++n_sh; }
#endif
                  continue;
                }
              } else
              shell_eval<5, LMAX>(l_, px - at_xyz[3 * sh_meta[5 * sh + 4]], py - at_xyz[3 * sh_meta[5 * sh + 4] + 1], pz - at_xyz[3 * sh_meta[5 * sh + 4] + 2], pr_exp + q0, pr_coef + q0, np_,
                                  [&](int m, double v, double ax, double ay, double az, double lp) {
                                    double* tl = region + (size_t)(krow + m) * 16 + pl;
                                    tl[0] = v; tl[(size_t)KT * 16] = ax; tl[(size_t)2 * KT * 16] = ay; tl[(size_t)3 * KT * 16] = az;
                                    tl[(size_t)4 * KT * 16] = lp;
                                  });
            }
#endif
#ifdef PQA_RES_CLK
            if (PBC) { PQA_RCLK2(0, c_a); PQA_RCLK2(1, c_b); PQA_RCLK2(2, c_c); PQA_RCLK2(3, n_sh); PQA_RCLK2(4, n_im); }
            if (PBC && blockIdx.x < 64 && (threadIdx.x & 63) == 0) pqa_res_clk3[blockIdx.x * 8 + (threadIdx.x >> 6)] = c_a + c_b + c_c;  // per wave: its phase 1
#endif
          }
          // B operand of this wave's k-steps (L2-resident coefficient rows): a ring of four, the first three requested behind the AO
          // phase — in flight while the waves meet at the barrier — and k-step q + 3 while k-step q is contracted (all of them at
          // once would be 32 registers across the contraction)
          const double* cb = cpad + (size_t)(row_base + kq) * ldc + 16 * u + i16;
          double bq[4];
#pragma unroll
          for (int q = 0; q < 3; ++q) bq[q] = (kwv + q * KW < nks) ? cb[(size_t)(kwv + q * KW) * 4 * ldc] : 0.0;
          bq[3] = 0.0;
          PQA_RCLK(1);
          res_block_sync();
          d4 acc[5];
#pragma unroll
          for (int c = 0; c < 5; ++c) acc[c] = (d4){0.0, 0.0, 0.0, 0.0};
#ifndef PQA_RES_ABL_NOMFMA
          {
            const double* a_ = region + (size_t)kq * 16 + i16;
#pragma unroll 1
            for (int q4 = 0; q4 < PQA_RES_MAXKS; q4 += 4) {  // four k-steps per trip (the ring's static indices)
              if (kwv + q4 * KW >= nks) break;
#pragma unroll
              for (int qq = 0; qq < 4; ++qq) {
                const int ks = kwv + (q4 + qq) * KW;
                if (ks + 3 * KW < nks) bq[(qq + 3) & 3] = cb[(size_t)(ks + 3 * KW) * 4 * ldc];
                if (ks < nks) {
                  double ac[5];
#pragma unroll
                  for (int c = 0; c < 5; ++c) ac[c] = a_[((size_t)c * ktv + 4 * ks) * 16];
#pragma unroll
                  for (int c = 0; c < 5; ++c) acc[c] = __builtin_amdgcn_mfma_f64_16x16x4f64(ac[c], bq[qq], acc[c], 0, 0, 0);
                }
                __builtin_amdgcn_sched_barrier(0);
              }
            }
          }
#endif
          if (ps == RT.npass - 1) prefetch();
          PQA_RCLK(2);
          // K-partials of this wave: lane holds D[point = kq + 4 rr][orbital = 16 u + i16].  One pass: they take the tile's place
          // (every wave has to be done reading it); several passes: own memory behind the tile, accumulated pass by pass by the
          // same lane
          if (RT.part_off == 0) res_block_sync();
          {
            double* pw = part + ((size_t)kw * 16 + kq) * PS + 16 * u + i16;
            if (ps == 0) {
#pragma unroll
              for (int c = 0; c < 5; ++c)
#pragma unroll
                for (int rr = 0; rr < 4; ++rr) pw[(size_t)4 * rr * PS + c * 16 * nt] = acc[c][rr];
            } else {
#pragma unroll
              for (int c = 0; c < 5; ++c)
#pragma unroll
                for (int rr = 0; rr < 4; ++rr) pw[(size_t)4 * rr * PS + c * 16 * nt] += acc[c][rr];
            }
          }
        }
        res_block_sync();
        PQA_RCLK(3);
        // ---- this walker's rows: the KW partials added in a fixed order (thread r: orbital r, five components)
        if (r < 16 * nt) {
          if (KW == 4) res_combine<4>(part + (size_t)wl * PS + r, PS, 16 * nt, rn + r);
          else res_combine<8>(part + (size_t)wl * PS + r, PS, 16 * nt, rn + r);
        }
        res_wave_sync();
        PQA_RCLK(7);
        // Slater sums at the proposal: ratio and gradient rows against T[i] (zero beyond n)
        if (CX) {
          if (RT.twist) {  // wrap phase of the folded proposal (orbitals.py:203-213; k_row_phase): every orbital row times exp(i theta)
            const double cs = ws[20], sn = ws[21];
            if (r < nh) {
#pragma unroll
              for (int c = 0; c < 5; ++c) {
                const double re = rn[c * 32 + r], im = rn[c * 32 + nh + r];
                rn[c * 32 + r] = re * cs - im * sn; rn[c * 32 + nh + r] = re * sn + im * cs;
              }
            }
            res_wave_sync();
          }
          // complex row sums: lane r < n holds T[i][r] = (rowE[2 r], rowE[2 r + 1]) and the row's entries of ITS orbital
          const double tr = r < n ? rowE[wl * 32 + 2 * r] : 0.0, ti = r < n ? rowE[wl * 32 + 2 * r + 1] : 0.0;
          const double a0 = rn[oc], b0 = rn[nh + oc], a1 = rn[32 + oc], b1 = rn[32 + nh + oc];
          const double a2 = rn[64 + oc], b2 = rn[64 + nh + oc], a3 = rn[96 + oc], b3 = rn[96 + nh + oc];
          p0 = a0 * tr - b0 * ti; q0i = a0 * ti + b0 * tr; p1 = a1 * tr - b1 * ti; q1i = a1 * ti + b1 * tr;
          p2 = a2 * tr - b2 * ti; q2i = a2 * ti + b2 * tr; p3 = a3 * tr - b3 * ti; q3i = a3 * ti + b3 * tr;
          p0 = res_sum32(p0); p1 = res_sum32(p1); p2 = res_sum32(p2); p3 = res_sum32(p3);
          q0i = res_sum32(q0i); q1i = res_sum32(q1i); q2i = res_sum32(q2i); q3i = res_sum32(q3i);
        } else {
        const double te = rowE[wl * 32 + r];
        p0 = rn[oc] * te; p1 = rn[32 + oc] * te; p2 = rn[64 + oc] * te; p3 = rn[96 + oc] * te;
        p0 = res_sum32(p0); p1 = res_sum32(p1); p2 = res_sum32(p2); p3 = res_sum32(p3);
        }
        PQA_RCLK(8);
      }
      const bool have_dec = i >= 0, have_prop = i + 1 < n;
      bool accd = false;
      if (have_dec) {
        // ================= decide electron i (mc.py:124-132; dmc.py:57-70): the same numbers in all lanes of the walker
        const double npx = ws[0], npy = ws[1], npz = ws[2];
        const double dr = p0, di = q0i, m2 = CX ? dr * dr + di * di : dr * dr;
        double hx, hy, hz;
        if (CX) {  // Re(grad / value) = Re(grad conj(value)) / |value|^2 (lw_slater_terms)
          const double d_ = 1.0 / m2;
          hx = finite_or((p1 * dr + q1i * di) * d_, 0.0); hy = finite_or((p2 * dr + q2i * di) * d_, 0.0); hz = finite_or((p3 * dr + q3i * di) * d_, 0.0);
        } else { hx = finite_or(p1 / p0, 0.0); hy = finite_or(p2 / p0, 0.0); hz = finite_or(p3 / p0, 0.0); }
        const double val = finite_or(dr, 1.0);
        double val2 = CX ? finite_or(m2, 1.0) : val * val;
#ifndef PQA_RES_ABL_NOJAS
        if (has_jastrow) {
          ResJ jn{0.0, 0.0, 0.0, 0.0};
          if (S.jq_on) res_jas_m<PBC>(S, r, cx, cy, cz, at_xyz, acoef, aql, jt, e, npx, npy, npz, jn, pbt);
          else {
            double g3[3];
            res_jas_part<PBC>(S, e, r, npx, npy, npz, cx, cy, cz, at_xyz, acoef, aql, jn.u, g3);
            jn.x = g3[0]; jn.y = g3[1]; jn.z = g3[2];
          }
          jn.u = res_sum32(jn.u); jn.x = res_sum32(jn.x); jn.y = res_sum32(jn.y); jn.z = res_sum32(jn.z);
          hx += jn.x; hy += jn.y; hz += jn.z;
          const double ej = exp(jn.u - ws[9]);
          val2 *= ej * ej;
        }
#endif
        PQA_RCLK(9);
        {
          const double z0 = ws[3], z1 = ws[4], z2 = ws[5], d0 = ws[6], d1 = ws[7], d2 = ws[8];
          const double fwd = z0 * z0 + z1 * z1 + z2 * z2;
          double bx, by, bz;
          if (DMC) {
            limdrift_dmc(hx, hy, hz, mb.tstep);
            bx = z0 + d0 + hx; by = z1 + d1 + hy; bz = z2 + d2 + hz;
          } else {
            limdrift3(hx, hy, hz);
            bx = z0 + mb.tstep * (d0 + hx); by = z1 + mb.tstep * (d1 + hy); bz = z2 + mb.tstep * (d2 + hz);
          }
          const double bwd = bx * bx + by * by + bz * bz;
          double ratio = val2 * exp(jt[3 * PQA_JQ + 16] * (fwd - bwd));
          if (DMC && !CX) ratio *= (val > 0.0) ? 1.0 : ((val < 0.0) ? -1.0 : 0.0);  // fixed node (dmc.py:64-66)
          accd = ratio > uacc;
          if (DMC && r == 0) {
            const double rx = z0 + d0, ry = z1 + d1, rz = z2 + d2, r2 = rx * rx + ry * ry + rz * rz;
            ws[13] += r2;
            if (accd) ws[14] += r2;
          }
        }
        if (live && r == 0 && mb.accept_rec) mb.accept_rec[(size_t)e * W + wg] = accd;
        PQA_RCLK(4);
        if (accd) {
          // Sherman-Morrison on the register rows (slater.py:88-94): R = T_old[i] / ratio, T[j] -= R (V . T[j]), T[i] = R;
          // the row's dot product in the PQA_ROWDOT order of the lane-per-walker kernels.  Eight columns at a time (the
          // scheduling fences keep the compiler from requesting all 64 LDS operands at once)
#ifndef PQA_RES_ABL_NOSM
          if (CX) {
            // complex rows: 16 (re, im) columns; the row's dot product V . T[r] in the PQA_ROWDOT order of the complex lane-per-walker
            // kernels (quarters of the complex columns), R = T_old[i] / ratio
            const double ir = dr / m2, ii = -di / m2;
            double pr4[4] = {0.0, 0.0, 0.0, 0.0}, pi4[4] = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
            for (int k4 = 0; k4 < 4; ++k4) {
#pragma unroll
              for (int kc = 4 * k4; kc < 4 * k4 + 4; ++kc) {
                const int o = occs[kc];
                const double vr = rn[o], vi = rn[nh + o];
                pr4[k4] += vr * t[2 * kc] - vi * t[2 * kc + 1];
                pi4[k4] += vr * t[2 * kc + 1] + vi * t[2 * kc];
              }
              __builtin_amdgcn_sched_barrier(0);
            }
            const double tmp = ((pr4[0] + pr4[1]) + pr4[2]) + pr4[3], tmi = ((pi4[0] + pi4[1]) + pi4[2]) + pi4[3];
            const double* Re = rowE + wl * 32;
#pragma unroll
            for (int k4 = 0; k4 < 4; ++k4) {
#pragma unroll
              for (int kc = 4 * k4; kc < 4 * k4 + 4; ++kc) {
                const double2 e2 = *reinterpret_cast<const double2*>(Re + 2 * kc);
                const double rr = e2.x * ir - e2.y * ii, ri = e2.x * ii + e2.y * ir;
                t[2 * kc] = (r == i) ? rr : t[2 * kc] - (rr * tmp - ri * tmi);
                t[2 * kc + 1] = (r == i) ? ri : t[2 * kc + 1] - (rr * tmi + ri * tmp);
              }
              __builtin_amdgcn_sched_barrier(0);
            }
          } else {
          const double inv = 1.0 / dr;
          double p4[4] = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
          for (int k8 = 0; k8 < 4; ++k8) {
            if (ident) {
#pragma unroll
              for (int k = 8 * k8; k < 8 * k8 + 8; k += 2) {
                const double2 v2 = *reinterpret_cast<const double2*>(rn + k);
                p4[k8] += v2.x * t[k];
                p4[k8] += v2.y * t[k + 1];
              }
            } else {
#pragma unroll
              for (int k = 8 * k8; k < 8 * k8 + 8; ++k) p4[k8] += rn[occs[k]] * t[k];
            }
            __builtin_amdgcn_sched_barrier(0);
          }
          const double tmp = ((p4[0] + p4[1]) + p4[2]) + p4[3];
          const double* Re = rowE + wl * 32;
#pragma unroll
          for (int k8 = 0; k8 < 4; ++k8) {
#pragma unroll
            for (int k = 8 * k8; k < 8 * k8 + 8; k += 2) {
              const double2 r2 = *reinterpret_cast<const double2*>(Re + k);
              const double R0 = r2.x * inv, R1 = r2.y * inv;
              t[k] = (r == i) ? R0 : t[k] - R0 * tmp;
              t[k + 1] = (r == i) ? R1 : t[k + 1] - R1 * tmp;
            }
            __builtin_amdgcn_sched_barrier(0);
          }
          }
#endif
          PQA_RCLK(13);
          // sign and log of the determinant (per walker, in LDS; lane 0): the ratios' magnitudes as a running product, its
          // logarithm taken when it leaves [1e-60, 1e60] and at the end of the spin's sweep (log of a product = sum of logs to
          // rounding; one log per move was ~1 us)
          if (r == 0) {
            const double mag = CX ? sqrt(m2) : fabs(dr);
            if (CX) {  // phase *= ratio / |ratio|
              const double ur = dr / mag, ui = di / mag, sr = ws[10], si = ws[19];
              ws[10] = sr * ur - si * ui; ws[19] = sr * ui + si * ur;
            } else ws[10] *= (dr > 0.0) ? 1.0 : ((dr < 0.0) ? -1.0 : dr);
            double lpr = ws[12] * mag;
            if (!(lpr > 1e-60 && lpr < 1e60)) { ws[11] += log(lpr); lpr = 1.0; }
            ws[12] = lpr;
            ws[15] += 1.0;
          }
          if (r == i) {
            if (s) { cx[1] = npx; cy[1] = npy; cz[1] = npz; } else { cx[0] = npx; cy[0] = npy; cz[0] = npz; }
            if (PBC && live && mb.wrap) {  // PeriodicConfigs.move: the electron's wrap counters follow (coord.py:180-189)
              int* wp = mb.wrap + ((size_t)wg * S.nelec + e) * 3;
              wp[0] += (int)ws[16]; wp[1] += (int)ws[17]; wp[2] += (int)ws[18];
            }
          }
          // the proposal's rows become the cached rows of electron i: into the walker's other slot, selector flipped
          const int cur = __shfl(selr, (lane & 32) | i, 64);
          if (live && r < nmo) {
            double* out = rcs + (((size_t)i * 2 + (cur ^ 1)) * W + wg) * 5 * nmo;
#pragma unroll
            for (int c = 0; c < 5; ++c) out[c * nmo + r] = rn[c * 32 + r];
          }
          if (r == i) {
            selr = cur ^ 1;
            if (live) sels[(size_t)i * W + wg] = (uint8_t)selr;
          }
        }
        PQA_RCLK(5);
      }
      // ================= propose electron i + 1 (mc.py:117-121): drift at its current position
      if (have_prop) {
        const int ip = i + 1, ep = e0 + ip;
        res_wave_sync();  // (the decision's reads of rowE and wsc are done)
        if (r == ip) {
#pragma unroll
          for (int k = 0; k < 32; ++k) rowE[wl * 32 + k] = t[k];
        }
        res_wave_sync();
        PQA_RCLK(10);
        double gx, gy, gz;
        if (CX) {
          const double tr = r < n ? rowE[wl * 32 + 2 * r] : 0.0, ti = r < n ? rowE[wl * 32 + 2 * r + 1] : 0.0;
          double q0 = ro[0] * tr - ro[4] * ti, s0 = ro[0] * ti + ro[4] * tr, q1 = ro[1] * tr - ro[5] * ti, s1 = ro[1] * ti + ro[5] * tr;
          double q2 = ro[2] * tr - ro[6] * ti, s2 = ro[2] * ti + ro[6] * tr, q3 = ro[3] * tr - ro[7] * ti, s3 = ro[3] * ti + ro[7] * tr;
          q0 = res_sum32(q0); q1 = res_sum32(q1); q2 = res_sum32(q2); q3 = res_sum32(q3);
          s0 = res_sum32(s0); s1 = res_sum32(s1); s2 = res_sum32(s2); s3 = res_sum32(s3);
          const double d_ = 1.0 / (q0 * q0 + s0 * s0);
          gx = finite_or((q1 * q0 + s1 * s0) * d_, 0.0); gy = finite_or((q2 * q0 + s2 * s0) * d_, 0.0); gz = finite_or((q3 * q0 + s3 * s0) * d_, 0.0);
        } else {
          const double te = rowE[wl * 32 + r];
          double q0 = ro[0] * te, q1 = ro[1] * te, q2 = ro[2] * te, q3 = ro[3] * te;
          q0 = res_sum32(q0); q1 = res_sum32(q1); q2 = res_sum32(q2); q3 = res_sum32(q3);
          gx = finite_or(q1 / q0, 0.0); gy = finite_or(q2 / q0, 0.0); gz = finite_or(q3 / q0, 0.0);
        }
        const int src = (lane & 32) | ip;
        const double pox = __shfl(s ? cx[1] : cx[0], src, 64), poy = __shfl(s ? cy[1] : cy[0], src, 64), poz = __shfl(s ? cz[1] : cz[0], src, 64);
        double U0 = 0.0;
        PQA_RCLK(11);
#ifndef PQA_RES_ABL_NOJAS
        if (has_jastrow) {
          ResJ jo{0.0, 0.0, 0.0, 0.0};
          if (S.jq_on) res_jas_m<PBC>(S, r, cx, cy, cz, at_xyz, acoef, aql, jt, ep, pox, poy, poz, jo, pbt);
          else {
            double g3[3];
            res_jas_part<PBC>(S, ep, r, pox, poy, poz, cx, cy, cz, at_xyz, acoef, aql, jo.u, g3);
            jo.x = g3[0]; jo.y = g3[1]; jo.z = g3[2];
          }
          jo.u = res_sum32(jo.u); jo.x = res_sum32(jo.x); jo.y = res_sum32(jo.y); jo.z = res_sum32(jo.z);
          U0 = jo.u; gx += jo.x; gy += jo.y; gz += jo.z;
        }
#endif
        PQA_RCLK(12);
        if (DMC) limdrift_dmc(gx, gy, gz, mb.tstep); else limdrift3(gx, gy, gz);
        const double sq = jt[3 * PQA_JQ + 15], df = DMC ? 1.0 : mb.tstep;
        const double z0 = g0 * sq, z1 = g1 * sq, z2 = g2 * sq;
        if (r == 0) {
          double nx = pox + z0 + gx * df, ny = poy + z1 + gy * df, nz = poz + z2 + gz * df;
          if (PBC && CX && RT.twist) {
            // twisted cell: the walker stays unfolded (include/pyqmc_amd.h); the orbitals are tabulated inside the cell — the folded copy
            // for the AO phase and the wrap phase exp(i k_t . wrap . lattice) of that fold (orbitals.py:203-213) beside it
            double f0 = nx * pbt[0] + ny * pbt[3] + nz * pbt[6], f1 = nx * pbt[1] + ny * pbt[4] + nz * pbt[7], f2 = nx * pbt[2] + ny * pbt[5] + nz * pbt[8];
            const double w0 = floor(f0), w1 = floor(f1), w2 = floor(f2);
            f0 -= w0; f1 -= w1; f2 -= w2;
            ws[16] = f0 * pbt[9] + f1 * pbt[12] + f2 * pbt[15];
            ws[17] = f0 * pbt[10] + f1 * pbt[13] + f2 * pbt[16];
            ws[18] = f0 * pbt[11] + f1 * pbt[14] + f2 * pbt[17];
            double sn_, cs_;
            sincos(w0 * pbt[30] + w1 * pbt[31] + w2 * pbt[32], &sn_, &cs_);
            ws[20] = cs_; ws[21] = sn_;
          } else
          if (PBC) {  // make_irreducible (mc.py:121, coord.py:164-178): the proposal inside the cell, the cells it crossed kept for the accept
            double f0 = nx * pbt[0] + ny * pbt[3] + nz * pbt[6], f1 = nx * pbt[1] + ny * pbt[4] + nz * pbt[7], f2 = nx * pbt[2] + ny * pbt[5] + nz * pbt[8];
            const double w0 = floor(f0), w1 = floor(f1), w2 = floor(f2);  // (fold_cell, pqa_common.hpp, on the block's copy of the cell)
            f0 -= w0; f1 -= w1; f2 -= w2;
            nx = f0 * pbt[9] + f1 * pbt[12] + f2 * pbt[15];
            ny = f0 * pbt[10] + f1 * pbt[13] + f2 * pbt[16];
            nz = f0 * pbt[11] + f1 * pbt[14] + f2 * pbt[17];
            ws[16] = w0; ws[17] = w1; ws[18] = w2;
          }
          ws[0] = nx; ws[1] = ny; ws[2] = nz;
          ws[3] = z0; ws[4] = z1; ws[5] = z2; ws[6] = gx; ws[7] = gy; ws[8] = gz; ws[9] = U0;
        }
        res_wave_sync();
        PQA_RCLK(6);
      }
    }
    // ---- this spin's state back to the planes
    if (live && r < n) {
#pragma unroll
      for (int k8 = 0; k8 < 4; ++k8) {
#pragma unroll
        for (int k = 8 * k8; k < 8 * k8 + 8; ++k)
          if (k < ncol) Tg[((size_t)r * ncol + k) * W] = t[k];
        __builtin_amdgcn_sched_barrier(0);
      }
    }
    if (live && r == 0) {
      double* dsg = s ? L.dsign[1] : L.dsign[0];
      if (CX) { dsg[2 * wg] = ws[10]; dsg[2 * wg + 1] = ws[19]; } else dsg[wg] = ws[10];
      (s ? L.dlog[1] : L.dlog[0])[wg] = ws[11] + log(ws[12]);
    }
    __syncthreads();  // (rowE / region reads of this spin's last decision before the next spin's first proposal)
  }
  if (live) {
#pragma unroll
    for (int q = 0; q < 2; ++q) {
      const int j = (q ? S.nup : 0) + r;
      if (r < (q ? S.ndn : S.nup)) {
        double* xj = L.xt + (size_t)j * 3 * W + wg;
        xj[0] = cx[q]; xj[W] = cy[q]; xj[2 * W] = cz[q];
      }
    }
    if (r == 0) {
      mb.acc_w[wg] += (int)ws[15];
      if (DMC) { mb.r2_prop[wg] += ws[13]; mb.r2_acc[wg] += ws[14]; }
    }
  }
}
