// pqa_res8.hpp — the resident electron sweep for open-boundary, real, single-determinant handles, second generation (round 6):
// EIGHT walkers per 256-thread block, TWO independent blocks per CU, wave-uniform atomic-orbital phase.
//
// Why.  k_sweep_res (pqa_res.hpp) keeps a 16-walker tile on one CU as ONE 512-thread block whose eight waves walk through the
// seven phases of a move in lock step (AO 6.6 us -> contraction 4.9 -> Jastrow 2.4 -> ... = 20.7 us per move, 32 % of the fp64
// pipe): while the block waits at a barrier, in a dependent chain or for its MFMA results, nothing else is resident to use the
// pipe, and its AO phase runs FOUR different shells in every wave (thread = (point, lane group): 16 points x 4 groups), i.e. up to
// four divergent passes of shell_eval at a quarter of the lanes each.  Here:
//   * a block owns 8 walkers (thread (walker wl = tid / 32, row r = tid % 32), the inverse row in 64 registers as before), so
//     TWO blocks of four waves fit a CU (2 x 256 threads x 256 registers, 2 x <= 80 KB of LDS).  They are independent
//     workgroups: one block's AO / Jastrow instructions issue under the other's MFMA chain, barrier waits and dependent sums.
//     Barriers span four waves instead of eight;
//   * the AO phase is WAVE-UNIFORM: a work item is one shell TYPE (same l, same primitive sequence — the same shell of every
//     atom of a species) on up to eight atoms; lane = (point = lane % 8, atom slot = lane / 8).  l, the primitive count and the
//     exponent / coefficient addresses are scalars, the switch over l is a scalar branch, every lane runs the same
//     instructions on its own (point, atom) displacement: no divergence, no exec-masked replays.  Items are dealt to the four
//     waves by descending cost (host, r8_setup);
//   * the contraction pairs two components in the 16 rows of v_mfma_f64_16x16x4_f64 (rows 0-7: component c of the 8 points,
//     rows 8-15: component c + 1), three MFMAs per k-step for the five components (the third half empty), K split over the
//     waves as before.
// Decisions, drift, Sherman-Morrison, Jastrow sums, cache rows, tapes: the statements of k_sweep_res, unchanged (same order of
// every per-walker sum; the orbital rows themselves sum the AOs in a different K order than k_sweep_res: rounding-level
// differences, identical decisions off measure-zero ties).  Reference: mc.py:112-137, slater.py:88-94,342-418,
// jastrowspin.py:296-385, numba/gto.py:89-254.
#pragma once
#include "pqa_res.hpp"
#include "pqa_res8_tab.hpp"

// One Jastrow pair of the merged route, branch-free (res_pair_m), handed back instead of accumulated: the terms u and g = sg (dx, dy, dz).
struct R8Pair { double du, sg, dx, dy, dz; };
// (no fix-up of r = 0 and no early replacement of out-of-range distances: whatever an excluded pair produces — the self pair sits at r = 0
// exactly, 1 / r = inf, NaN products — is discarded by the final select, which is all the branch-free evaluation needs)
__device__ __forceinline__ R8Pair r8_pair(bool valid, double dx, double dy, double dz, double rcut, double ircut, const double (&D)[5],
                                          const double* __restrict__ q, double cpar, double caux, double ccoef) {
  const double x = dx * dx + dy * dy + dz * dz;
  double rr, ri;
  {  // sqrt_rinv without its x == 0 branch
    const double y = __builtin_amdgcn_rsq(x);
    double g = x * y, h = 0.5 * y;
    const double e = fma(-h, g, 0.5);
    g = fma(g, e, g); h = fma(h, e, h);
    double d = fma(-g, g, x);
    g = fma(d, h, g);
    d = fma(-g, g, x);
    g = fma(d, h, g);
    const double t = h + h;
    rr = g; ri = fma(t, fma(-g, t, 1.0), t);
  }
  const bool in = valid && rr < rcut;
  const RadShared sh = rad_shared_ri<1>(rr, ri, ircut);
  const MergedSums m = pade_merged<1, 4, false>(D, q, sh.p);
  double du = sh.omp * m.S1, sg = sh.c0 * m.S2;
  {
    double v, gf, lpl;
    rad_fn<1>(1, cpar, caux, rcut, sh, v, gf, lpl);
    du += ccoef * v;
    sg += ccoef * gf;
  }
  R8Pair p;
  p.du = in ? du : 0.0; p.sg = in ? sg : 0.0; p.dx = dx; p.dy = dy; p.dz = dz;
  return p;
}
__device__ __forceinline__ void r8_acc(ResJ& a, const R8Pair& p) { a.u += p.du; a.x += p.sg * p.dx; a.y += p.sg * p.dy; a.z += p.sg * p.dz; }
__device__ __forceinline__ R8Pair r8_sel(bool c, const R8Pair& a, const R8Pair& b) {
  R8Pair p;
  p.du = c ? a.du : b.du; p.sg = c ? a.sg : b.sg; p.dx = c ? a.dx : b.dx; p.dy = c ? a.dy : b.dy; p.dz = c ? a.dz : b.dz;
  return p;
}
// Both Jastrow evaluations of a move in ONE batch of independent pair chains (merged route): jn = this thread's share of U, grad U of
// the decided electron e at its proposal n; joR / joA = the share of the NEXT electron ep (same spin) at its current position o with
// electron e left where it is / moved to its proposal.  Nothing here needs the orbitals, so the kernel runs it ahead of the AO phase, where
// the two serial evaluations of k_sweep_res (2.3 + 2.1 us of dependent chains behind the contraction) are out of the move's critical path
// and the registers are free.  Per lane the pairs enter the sums in the order of res_jas_m (up partner, down partner, ion): the sums are
// the ones k_sweep_res forms.
template <bool NEXT>
__device__ __forceinline__ void r8_jas_dual(const SysDev& S, int r, const double (&cx)[2], const double (&cy)[2], const double (&cz)[2],
                                            const double* __restrict__ at_xyz, const double* __restrict__ acoef, const double* __restrict__ aq,
                                            const double* __restrict__ jt, int e, double nx, double ny, double nz, int ep, double ox, double oy,
                                            double oz, ResJ& jn, ResJ& joR, ResJ& joA) {
  const int se = e >= S.nup;
  const double irb = jt[3 * PQA_JQ + 13], ira = jt[3 * PQA_JQ + 14];
  const bool bcusp = S.nb > 0 && S.b_kind[0] == 1, acusp = S.na > 0 && S.a_kind[0] == 1;
  const double bcp = bcusp ? S.b_param[0] : 0.0, bca = bcusp ? S.b_aux[0] : 0.0, acp = acusp ? S.a_param[0] : 0.0, aca = acusp ? S.a_aux[0] : 0.0;
  double Db[5], Da[5];
#pragma unroll
  for (int i = 0; i < 5; ++i) { Db[i] = jt[3 * PQA_JQ + i]; Da[i] = jt[3 * PQA_JQ + 5 + i]; }
  const int j0 = r, j1 = S.nup + r;
  const bool v0 = S.nb > 0 && r < S.nup, v1 = S.nb > 0 && r < S.ndn;
  const double* q0 = jt + se * PQA_JQ;        // numerators of the (se, up) channel; (se, down) follows
  const double c0 = jt[3 * PQA_JQ + 10 + se], c1 = jt[3 * PQA_JQ + 11 + se];
  // ion partner(s) of this lane: atom r, and atom r + 32 of molecules with more than 32 atoms (second pass below)
  auto ion = [&](int I, double x, double y, double z) __attribute__((always_inline)) {
    const int Ic = I < S.natom ? I : 0;
    const double* qrec = aq + ((size_t)se * S.natom + Ic) * PQA_JQP;  // (conflict-free record layout of the LDS copy, pqa_res.hpp)
    return r8_pair(S.na > 0 && I < S.natom, x - at_xyz[3 * Ic], y - at_xyz[3 * Ic + 1], z - at_xyz[3 * Ic + 2], S.rcut_a, ira, Da, qrec, acp, aca, qrec[PQA_JQ]);
  };
  // The seven pairs of the move (three of the decided electron, four of the next one), ONE AT A TIME, the ion pairs inside (one-trip) loops:
  // a pair's own chains (denominator and two numerator polynomials, the cusp function) already run side by side, and every attempt to run
  // pairs side by side — two, three or all seven in one basic block — made the compiler hold their LDS operands early and spill a quarter of
  // the inverse row across this block (17-22 scratch stores and reloads per move: 19-21 us per move of two resident blocks against 17.2).
  const int nion = S.natom > 32 ? 2 : 1;
  const R8Pair na_ = r8_pair(v0 && j0 != e, nx - cx[0], ny - cy[0], nz - cz[0], S.rcut_b, irb, Db, q0, bcp, bca, c0);
  r8_acc(jn, na_);
  __builtin_amdgcn_sched_barrier(0);
  const R8Pair nb_ = r8_pair(v1 && j1 != e, nx - cx[1], ny - cy[1], nz - cz[1], S.rcut_b, irb, Db, q0 + PQA_JQ, bcp, bca, c1);
  r8_acc(jn, nb_);
  __builtin_amdgcn_sched_barrier(0);
  for (int q = 0; q < nion; ++q) { const R8Pair nc_ = ion(r + 32 * q, nx, ny, nz); r8_acc(jn, nc_); }
  if (NEXT) {  // ---- the next electron at its current position: partner e at its old place (R) or at its proposal (A)
    __builtin_amdgcn_sched_barrier(0);
    // the partner slot of e's spin twice: e where it is (R) and at its proposal (A; the same pair again in every lane but e's own)
    const bool mine = (se ? j1 : j0) == e;
    const double px_ = mine ? nx : (se ? cx[1] : cx[0]), py_ = mine ? ny : (se ? cy[1] : cy[0]), pz_ = mine ? nz : (se ? cz[1] : cz[0]);
    const R8Pair y = r8_pair((se ? v1 : v0) && (se ? j1 : j0) != ep, ox - px_, oy - py_, oz - pz_, S.rcut_b, irb, Db, q0 + se * PQA_JQ, bcp, bca, se ? c1 : c0);
    __builtin_amdgcn_sched_barrier(0);
    const R8Pair a = r8_pair(v0 && j0 != ep, ox - cx[0], oy - cy[0], oz - cz[0], S.rcut_b, irb, Db, q0, bcp, bca, c0);
    r8_acc(joR, a);
    if (se == 0) r8_acc(joA, y); else r8_acc(joA, a);  // (wave-uniform)
    __builtin_amdgcn_sched_barrier(0);
    const R8Pair b = r8_pair(v1 && j1 != ep, ox - cx[1], oy - cy[1], oz - cz[1], S.rcut_b, irb, Db, q0 + PQA_JQ, bcp, bca, c1);
    r8_acc(joR, b);
    if (se == 1) r8_acc(joA, y); else r8_acc(joA, b);
    __builtin_amdgcn_sched_barrier(0);
    for (int q = 0; q < nion; ++q) { const R8Pair c = ion(r + 32 * q, ox, oy, oz); r8_acc(joR, c); r8_acc(joA, c); }
  }
}

// limdrift3 (mc.py:76-89) with one division
__device__ __forceinline__ void r8_limdrift3(double& gx, double& gy, double& gz) {
  const double tot = sqrt(gx * gx + gy * gy + gz * gz);
  if (tot > 1.0) { const double it = 1.0 / tot; gx *= it; gy *= it; gz *= it; }
}
template <int KWC>
__device__ __forceinline__ void r8_combine(const double* __restrict__ pb, int PS, int cstride, double* __restrict__ rn_r) {
  double v[5][KWC];
#pragma unroll
  for (int c = 0; c < 5; ++c)
#pragma unroll
    for (int k = 0; k < KWC; ++k) v[c][k] = pb[(size_t)k * PQA_R8_NW * PS + c * cstride];
#pragma unroll
  for (int c = 0; c < 5; ++c) {
    double sum = v[c][0];
#pragma unroll
    for (int k = 1; k < KWC; ++k) sum += v[c][k];
    rn_r[c * 32] = sum;
  }
}

#ifndef PQA_R8_UNIT0
#define PQA_R8_UNIT0 1  // the value sum at an electron's own position taken as 1 (see the next-proposal drift below)
#endif
// grid = ceil((w_hi - w_lo) / 8) blocks of 256 threads, two per CU (256 registers per thread, dynamic LDS <= 80 KB).
// timing builds: the phase stamps of a move in the MIDDLE of the second spin's sweep (the last move has no next electron: half the Jastrow
// work, no prefetch)
#define PQA_R8CLK(k) do { if (s == 1 && i == 16) PQA_RCLK(k); } while (0)
// (timing builds) wall-clock stamps of a block's prologue / epilogue: 0 tables in LDS, 1 + 4 s: spin s rows loaded, 2 + 4 s: its moves done,
// 3 + 4 s: its state stored, 4 + 4 s: past the spin's closing barrier
#ifdef PQA_RES_CLK
#define PQA_R8T(k) PQA_RCLK2(k, wall_clock64())
#else
#define PQA_R8T(k) do { } while (0)
#endif
template <bool DMC, int LMAX>
static __global__ __launch_bounds__(PQA_R8_NT, 2) void k_sweep_r8(SysDev S, LwState L, MoveBuf mb, ChunkTab T, R8Tab RT, int has_jastrow,
                                                                  long W, long w_lo, long w_hi) {
  extern __shared__ double lds[];
  PQA_RCLK(14);
  const int tid = threadIdx.x, wv = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wl = tid >> 5, r = tid & 31;
  const int KT = RT.kt, CS = RT.cstride;
  double* region = lds;
  double* rowE = region + RT.region;           // [8][32] inverse row of the electron being moved
  double* wsc = rowE + PQA_R8_NW * 32;         // [8][PQA_RES_WS] per-walker scalars (layout of k_sweep_res)
  double* ixyz = wsc + PQA_R8_NW * PQA_R8_WS;  // [nitem][8][3] atom position of every (item, slot)
  double* pr_exp = ixyz + 24 * (size_t)RT.nitem;
  double* pr_coef = pr_exp + RT.nprim_u;
  double* at_xyz = pr_coef + RT.nprim_u;
  double* acoef = at_xyz + 3 * (size_t)S.natom;
  double* aql = acoef + 2 * (size_t)S.natom * (S.na > 0 ? S.na : 1);
  double* jt = aql + 2 * (size_t)S.natom * PQA_JQP;
  int* ihdr = (int*)(jt + PQA_RES_JT);         // [nitem][4]
  int* ilane = ihdr + 4 * (size_t)RT.nitem;    // [nitem][8][2]
  int* occ = ilane + 16 * (size_t)RT.nitem;    // [2][32]
  int* woff = occ + 64;                        // [5]
  double* ws = wsc + wl * PQA_R8_WS;

  const long wraw = w_lo + (long)blockIdx.x * PQA_R8_NW + wl;
  const bool live = wraw < w_hi;
  const long wg = live ? wraw : w_hi - 1;  // walkers past the end shadow the last one and store nothing

  // ---- tables and the walker's coordinates
  for (int p = tid; p < RT.nprim_u; p += PQA_R8_NT) { pr_exp[p] = RT.prim_exp_u[p]; pr_coef[p] = RT.prim_coef_u[p]; }
  for (int k = tid; k < 3 * S.natom; k += PQA_R8_NT) at_xyz[k] = S.atom_xyz[k];
  for (int k = tid; k < 2 * S.natom * S.na; k += PQA_R8_NT) acoef[k] = has_jastrow ? S.acoeff[k] : 0.0;
  for (int k = tid; k < 2 * S.natom * PQA_JQP; k += PQA_R8_NT) {  // [spin][ion][PQA_JQP] from the global [ion][spin][PQA_JQ]; slot 24: cusp coefficient
    const int j = k % PQA_JQP, I = (k / PQA_JQP) % S.natom, sp = k / (PQA_JQP * S.natom);
    double v = 0.0;
    if (has_jastrow && S.jq_on && S.na > 0) v = j < PQA_JQ ? S.aq[(size_t)(I * 2 + sp) * PQA_JQ + j] : (S.a_kind[0] == 1 ? S.acoeff[(I * S.na) * 2 + sp] : 0.0);
    aql[k] = v;
  }
  for (int k = tid; k < PQA_RES_JT; k += PQA_R8_NT) {
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
  for (int k = tid; k < 4 * RT.nitem; k += PQA_R8_NT) ihdr[k] = RT.item_hdr[k];
  for (int k = tid; k < 16 * RT.nitem; k += PQA_R8_NT) ilane[k] = RT.item_lane[k];
  for (int k = tid; k < 24 * RT.nitem; k += PQA_R8_NT) ixyz[k] = RT.item_xyz[k];
  for (int k = tid; k < 64; k += PQA_R8_NT) {
    const int s = k >> 5, q = k & 31, n = s ? S.ndn : S.nup;
    occ[k] = q < n ? (s ? S.det_occ[1][q] : S.det_occ[0][q]) : 0;
  }
  if (tid < 5) woff[tid] = RT.wave_off[tid];
  for (int k = tid; k < RT.region; k += PQA_R8_NT) region[k] = 0.0;  // (K-padding rows of the tile stay finite)
  double cx[2], cy[2], cz[2];
#pragma unroll
  for (int q = 0; q < 2; ++q) {  // slot 0: up electron r, slot 1: down electron nup + r
    const int j = (q ? S.nup : 0) + r;
    const double* xj = L.xt + (size_t)((r < (q ? S.ndn : S.nup)) ? j : 0) * 3 * W + wg;
    cx[q] = xj[0]; cy[q] = xj[W]; cz[q] = xj[2 * W];
  }
  if (r == 0) { ws[13] = 0.0; ws[14] = 0.0; ws[15] = 0.0; }
  __syncthreads();
  PQA_R8T(0);
  if (RT.stagger > 0 && (__builtin_amdgcn_s_getreg(0x3806) & 0xff) != 0) {  // HW_REG_LDS_ALLOC.LDS_BASE: not the first workgroup on this CU
    for (int k = 0; k < RT.stagger; ++k) { __builtin_amdgcn_s_sleep(100); }  // (64 x 100 cycles each)
  }

#pragma unroll 1
  for (int s = 0; s < 2; ++s) {
    const int n = s ? S.ndn : S.nup;
    if (n == 0) continue;
    const int nmo = s ? S.nmo[1] : S.nmo[0], ldc = s ? T.ldc[1] : T.ldc[0], nt = ldc >> 4, e0 = s ? S.nup : 0;
    const double* __restrict__ cpad = s ? T.cpad[1] : T.cpad[0];
    double* Tg = (s ? L.Tt[1] : L.Tt[0]) + wg;
    double* rcs = s ? L.rc[1] : L.rc[0];
    uint8_t* sels = s ? L.sel[1] : L.sel[0];
    const int KW = 4 / nt, u = wv % nt, kw = wv / nt;  // MFMA roles: orbital tile u, K-split part kw of KW
    const int PS = res_ps(nt);
    const int nks = KT >> 2;
    double* part = region;                      // [KW][8][PS]: the K-partials take the tile's place
    const int* occs = occ + 32 * s;
    const bool ident = (s ? S.occ_ident[1] : S.occ_ident[0]) != 0;
    // ---- the transposed inverse of this spin: row r of walker wl
    // (all 32 loads in flight at once — clamped addresses, the entries outside the n x n block multiplied by zero: with a select per element
    // the compiler branches around each load, and the opaque copy k_sweep_res used against that made it wait for every load before it issued
    // the next: 32 dependent HBM round trips per spin, ~40 us of a block's 1.07 ms)
    double t[32];
    {
      const double* trow = Tg + (size_t)(r < n ? r : 0) * n * W;
      const double rm = r < n ? 1.0 : 0.0;
#pragma unroll
      for (int k = 0; k < 32; ++k) t[k] = trow[(size_t)(k < n ? k : 0) * W];
#pragma unroll
      for (int k = 0; k < 32; ++k) t[k] *= (k < n) ? rm : 0.0;
    }
    int selr = (r < n) ? (int)sels[(size_t)r * W + wg] : 0;
    if (r == 0) {
      ws[10] = (s ? L.dsign[1] : L.dsign[0])[wg];
      ws[11] = (s ? L.dlog[1] : L.dlog[0])[wg]; ws[12] = 1.0;
    }

#ifdef PQA_RES_CLK
    { double chk = t[0] + t[31]; asm volatile("" :: "v"(chk)); }  // (the rows have arrived)
#endif
    PQA_R8T(1 + 4 * s);
#pragma unroll 1
    for (int i = -1; i < n; ++i) {  // iteration i: [orbitals at the proposals of electron i, decide i], then propose i + 1
      const int e = e0 + i;
      // (thread-index derived values from an OPAQUE copy: see k_sweep_res — whatever the compiler proves loop-invariant it hoists in
      // front of this ~8 000-instruction body and spills, together with part of the inverse row)
      int tido = tid;
      asm volatile("" : "+v"(tido));
      const int lane = tido & 63, wl = tido >> 5, r = tido & 31, i16 = lane & 15, kq = lane >> 4;
      const long wraw = w_lo + (long)blockIdx.x * PQA_R8_NW + wl;
      const bool live = wraw < w_hi;
      const long wg = live ? wraw : w_hi - 1;
      double* ws = wsc + wl * PQA_R8_WS;
      double* rn = part + (size_t)KW * PQA_R8_NW * PS + (size_t)wl * PQA_RES_RS;  // this walker's combined rows [5][32]
      const int oc = occs[r];
      double ro[4], uacc = 0.0, g0 = 0.0, g1 = 0.0, g2 = 0.0;
#pragma unroll
      for (int q = 0; q < 4; ++q) ro[q] = 0.0;
      double p0 = 1.0, p1 = 0.0, p2 = 0.0, p3 = 0.0;  // Slater sums at the proposal of electron i
      auto prefetch = [&]() {
        if (!PQA_R8_ON(8)) return;
        if (i + 1 < n) {
          const int slot = __shfl(selr, (lane & 32) | (i + 1), 64);
          const double* row = rcs + (((size_t)(i + 1) * 2 + slot) * W + wg) * 5 * nmo;
          if (r < n) { if (!PQA_R8_UNIT0) ro[0] = row[oc]; ro[1] = row[nmo + oc]; ro[2] = row[2 * nmo + oc]; ro[3] = row[3 * nmo + oc]; }
          const double* zt = mb.gauss + ((size_t)(e + 1) * W + wg) * 3;
          g0 = zt[0]; g1 = zt[1]; g2 = zt[2];
        }
        if (i >= 0) uacc = mb.unif[(size_t)e * W + wg];
      };
      if (i < 0) prefetch();
      if (i >= 0) {
        PQA_R8CLK(0);
        if (has_jastrow && PQA_R8_ON(4)) {
          // ---- both Jastrow evaluations of this move, ahead of the orbitals (r8_jas_dual): totals to wsc 16..27
          const bool nxt = i + 1 < n;
          const double npx = ws[0], npy = ws[1], npz = ws[2];
          const int srcn = (lane & 32) | (nxt ? i + 1 : i);
          const double pox = __shfl(s ? cx[1] : cx[0], srcn, 64), poy = __shfl(s ? cy[1] : cy[0], srcn, 64), poz = __shfl(s ? cz[1] : cz[0], srcn, 64);
          ResJ jn{0.0, 0.0, 0.0, 0.0}, jr{0.0, 0.0, 0.0, 0.0}, ja{0.0, 0.0, 0.0, 0.0};
          if (S.jq_on) {
            if (nxt) r8_jas_dual<true>(S, r, cx, cy, cz, at_xyz, acoef, aql, jt, e, npx, npy, npz, e + 1, pox, poy, poz, jn, jr, ja);
            else r8_jas_dual<false>(S, r, cx, cy, cz, at_xyz, acoef, aql, jt, e, npx, npy, npz, e + 1, pox, poy, poz, jn, jr, ja);
          } else {  // any basis, function by function: three plain evaluations (the third with e at its proposal)
            double g3[3];
            res_jas_part<false>(S, e, r, npx, npy, npz, cx, cy, cz, at_xyz, acoef, aql, jn.u, g3);
            jn.x = g3[0]; jn.y = g3[1]; jn.z = g3[2];
            if (nxt) {
              res_jas_part<false>(S, e + 1, r, pox, poy, poz, cx, cy, cz, at_xyz, acoef, aql, jr.u, g3);
              jr.x = g3[0]; jr.y = g3[1]; jr.z = g3[2];
              double ax[2] = {cx[0], cx[1]}, ay[2] = {cy[0], cy[1]}, az[2] = {cz[0], cz[1]};
              if (r == i) { if (s) { ax[1] = npx; ay[1] = npy; az[1] = npz; } else { ax[0] = npx; ay[0] = npy; az[0] = npz; } }
              res_jas_part<false>(S, e + 1, r, pox, poy, poz, ax, ay, az, at_xyz, acoef, aql, ja.u, g3);
              ja.x = g3[0]; ja.y = g3[1]; ja.z = g3[2];
            }
          }
          // The walker's totals of the twelve (four) sums through LDS instead of twelve 32-lane butterflies (13 vector instructions each, 160 of
          // the move's ~1300 Jastrow instructions): every lane writes its terms, lane v adds the 32 terms of sum v in a fixed order and leaves
          // the total in wsc 16 + v.  The staging area is the part of the region behind the partials and orbital rows, which nothing uses
          // between the combination of the previous move and this move's AO phase; only this walker's lanes touch its slice (wave-local order).
          {
            double* st = region + RT.jstage + (size_t)wl * (12 * 33);
            const int nv = nxt ? 12 : 4;
            st[r] = jn.u; st[33 + r] = jn.x; st[66 + r] = jn.y; st[99 + r] = jn.z;
            if (nxt) {
              st[132 + r] = jr.u; st[165 + r] = jr.x; st[198 + r] = jr.y; st[231 + r] = jr.z;
              st[264 + r] = ja.u; st[297 + r] = ja.x; st[330 + r] = ja.y; st[363 + r] = ja.z;
            }
            res_wave_sync();
            if (r < nv) {
              const double* sv = st + 33 * r;
              double a0 = sv[0], a1 = sv[1], a2 = sv[2], a3 = sv[3];
#pragma unroll
              for (int j = 4; j < 32; j += 4) { a0 += sv[j]; a1 += sv[j + 1]; a2 += sv[j + 2]; a3 += sv[j + 3]; }
              ws[16 + r] = (a0 + a1) + (a2 + a3);
            }
            res_wave_sync();
          }
        }
        PQA_R8CLK(9);
        res_block_sync();  // proposals of all 8 walkers are in wsc; the previous move's reads of the region are done
        int kwv = kw, csv = CS;
        asm volatile("" : "+s"(kwv), "+s"(csv));
        // ================= atomic orbitals at the 8 proposals: wave-uniform items (shell type x 8 atoms), lane = (point, atom slot)
        {
          const int pt = lane & 7, slot = lane >> 3;
          const double px = wsc[pt * PQA_R8_WS], py = wsc[pt * PQA_R8_WS + 1], pz = wsc[pt * PQA_R8_WS + 2];
          const int it1 = __builtin_amdgcn_readfirstlane(woff[wv + 1]), it0 = __builtin_amdgcn_readfirstlane(woff[wv]);
          // (an item's header — shell type, the slot's tile row and atom position — is requested while the previous item is evaluated: two
          // dependent LDS round trips in front of every item otherwise, three items per wave)
          auto hdr = [&](int it, int& h0, int& h1, int& h2, int& at, int& krow, double& ax, double& ay, double& az) __attribute__((always_inline)) {
            h0 = ihdr[4 * it]; h1 = ihdr[4 * it + 1]; h2 = ihdr[4 * it + 2];
            at = ilane[(it * 8 + slot) * 2]; krow = ilane[(it * 8 + slot) * 2 + 1];
            ax = ixyz[(it * 8 + slot) * 3]; ay = ixyz[(it * 8 + slot) * 3 + 1]; az = ixyz[(it * 8 + slot) * 3 + 2];
          };
          int h0 = 0, h1 = 0, h2 = 0, at = -1, krow = 0;
          double ax_ = 0.0, ay_ = 0.0, az_ = 0.0;
          if (it0 < it1) hdr(it0, h0, h1, h2, at, krow, ax_, ay_, az_);
#pragma unroll 1
          for (int it = it0; it < it1 && PQA_R8_ON(1); ++it) {
            int n0, n1, n2, nat, nkrow;
            double nax, nay, naz;
            hdr(min(it + 1, it1 - 1), n0, n1, n2, nat, nkrow, nax, nay, naz);
            const int l_ = __builtin_amdgcn_readfirstlane(h0), np_ = __builtin_amdgcn_readfirstlane(h1), q0 = __builtin_amdgcn_readfirstlane(h2);
            const bool on = at >= 0;
            double* tl = region + (size_t)krow * 8 + pt;
            shell_eval3<5, LMAX>(l_, px - ax_, py - ay_, pz - az_, pr_exp + q0, pr_coef + q0, np_,
                                          [&](int m, double v, double ax, double ay, double az, double lp) __attribute__((always_inline)) {
                                            if (on) {
                                              double* q = tl + m * 8;
                                              q[0] = v; q[csv] = ax; q[2 * csv] = ay; q[3 * csv] = az; q[4 * csv] = lp;
                                            }
                                          });
            h0 = n0; h1 = n1; h2 = n2; at = nat; krow = nkrow; ax_ = nax; ay_ = nay; az_ = naz;
          }
        }
        // B operand of this wave's k-steps (L2-resident coefficient rows, zero rows behind the basis: d_cres is padded to x16): k-step
        // ks = kw + q KW for q < nq = ceil(nks / KW) in EVERY wave — a wave whose last step lies behind the tile contracts the tile's last
        // rows against zeros.  Every load of the loop is unconditional: with a load inside a branch the compiler cannot count what is in
        // flight and waits for vmcnt(0), i.e. for the load it has just issued (k_sweep_res: 4.9 us per contraction against 2 of MFMA time).
        const int nq = (nks + KW - 1) / KW;  // k-steps of a wave (six per trip: the operand rings' static indices; a wave's last one may lie behind the tile)
        const size_t bstep = (size_t)4 * KW * ldc;
        const double* cb = cpad + (size_t)(4 * kwv + kq) * ldc + 16 * u + i16;
        double bq[6];
#pragma unroll
        for (int q = 0; q < 5; ++q) bq[q] = cb[(size_t)min(q, nq - 1) * bstep];
        bq[5] = 0.0;
        PQA_R8CLK(1);
        res_block_sync();
        d4 acc[3];
#pragma unroll
        for (int c = 0; c < 3; ++c) acc[c] = (d4){0.0, 0.0, 0.0, 0.0};
        {
          // A operand: row m = 8 h + p of MFMA j is component 2 j + h of point p (j = 2: component 4 in both halves, the upper one unused);
          // the next k-step's three operands are read while this one's MFMAs run, the B operand five steps ahead
          const int h8 = i16 >> 3, p8 = i16 & 7;
          const double* a0 = region + (size_t)h8 * csv + kq * 8 + p8;
          const double* a1 = region + (size_t)(2 + h8) * csv + kq * 8 + p8;
          const double* a2 = region + (size_t)4 * csv + kq * 8 + p8;
          double xa[2][3];
          {
            const int ka = min(kwv, nks - 1) * 32;
            xa[0][0] = a0[ka]; xa[0][1] = a1[ka]; xa[0][2] = a2[ka];
          }
#pragma unroll 1
          for (int q6 = 0; q6 < nq && PQA_R8_ON(2); q6 += 6) {
#pragma unroll
            for (int qq = 0; qq < 6; ++qq) {
              const int q = q6 + qq;
              bq[(qq + 5) % 6] = cb[(size_t)min(q + 5, nq - 1) * bstep];
              const int ka = min(kwv + (q + 1) * KW, nks - 1) * 32;
              xa[(qq + 1) & 1][0] = a0[ka]; xa[(qq + 1) & 1][1] = a1[ka]; xa[(qq + 1) & 1][2] = a2[ka];
              if (q < nq) {  // (wave-uniform; the loads above stay unconditional so that the compiler can count them)
                acc[0] = __builtin_amdgcn_mfma_f64_16x16x4f64(xa[qq & 1][0], bq[qq], acc[0], 0, 0, 0);
                acc[1] = __builtin_amdgcn_mfma_f64_16x16x4f64(xa[qq & 1][1], bq[qq], acc[1], 0, 0, 0);
                acc[2] = __builtin_amdgcn_mfma_f64_16x16x4f64(xa[qq & 1][2], bq[qq], acc[2], 0, 0, 0);
              }
            }
          }
        }
        prefetch();
        PQA_R8CLK(2);
        res_block_sync();  // every wave is done reading the tile: the K-partials take its place
        {
          // lane holds D[m = kq + 4 rr][orbital = 16 u + i16]: rr 0, 1 -> component 2 j of points kq, kq + 4; rr 2, 3 -> component 2 j + 1
          double* pw = part + ((size_t)kw * PQA_R8_NW + kq) * PS + 16 * u + i16;
#pragma unroll
          for (int j = 0; j < 3; ++j)
#pragma unroll
            for (int rr = 0; rr < 4; ++rr) {
              const int c = 2 * j + (rr >> 1);
              if (c < 5) pw[(size_t)4 * (rr & 1) * PS + c * 16 * nt] = acc[j][rr];
            }
        }
        res_block_sync();
        PQA_R8CLK(3);
        // ---- this walker's rows: the KW partials added in a fixed order (thread r: orbital r, five components)
        if (r < 16 * nt) {
          if (KW == 2) r8_combine<2>(part + (size_t)wl * PS + r, PS, 16 * nt, rn + r);
          else r8_combine<4>(part + (size_t)wl * PS + r, PS, 16 * nt, rn + r);
        }
        res_wave_sync();
        PQA_R8CLK(7);
        const double te = rowE[wl * 32 + r];
        p0 = rn[oc] * te; p1 = rn[32 + oc] * te; p2 = rn[64 + oc] * te; p3 = rn[96 + oc] * te;
        p0 = res_sum32(p0); p1 = res_sum32(p1); p2 = res_sum32(p2); p3 = res_sum32(p3);
        PQA_R8CLK(8);
      }
      const bool have_dec = i >= 0, have_prop = i + 1 < n;
      bool accd = false;
      if (have_dec) {
        // ================= decide electron i (mc.py:124-132; dmc.py:57-70): the same numbers in all lanes of the walker
        const double npx = ws[0], npy = ws[1], npz = ws[2];
        const double dr = p0;
        const double ip0 = 1.0 / p0;  // (one division for the three drift components)
        double hx = finite_or(p1 * ip0, 0.0), hy = finite_or(p2 * ip0, 0.0), hz = finite_or(p3 * ip0, 0.0);
        const double val = finite_or(dr, 1.0);
        double val2 = val * val;
        double jexp = 0.0;  // exponent of the Jastrow factor's part of the acceptance ratio: |Psi_new / Psi_old|^2 = val^2 exp(2 (U_new - U_old))
        if (has_jastrow) {  // (summed ahead of the orbitals: wsc 16..19)
          hx += ws[17]; hy += ws[18]; hz += ws[19];
          jexp = 2.0 * (ws[16] - ws[9]);
        }
        {
          const double z0 = ws[3], z1 = ws[4], z2 = ws[5], d0 = ws[6], d1 = ws[7], d2 = ws[8];
          const double fwd = z0 * z0 + z1 * z1 + z2 * z2;
          double bx, by, bz;
          if (DMC) {
            limdrift_dmc(hx, hy, hz, mb.tstep);
            bx = z0 + d0 + hx; by = z1 + d1 + hy; bz = z2 + d2 + hz;
          } else {
            r8_limdrift3(hx, hy, hz);
            bx = z0 + mb.tstep * (d0 + hx); by = z1 + mb.tstep * (d1 + hy); bz = z2 + mb.tstep * (d2 + hz);
          }
          const double bwd = bx * bx + by * by + bz * bz;
          double ratio = val2 * exp(jexp + jt[3 * PQA_JQ + 16] * (fwd - bwd));  // (one exponential for the Jastrow ratio and the Green's function ratio)
          if (DMC) ratio *= (val > 0.0) ? 1.0 : ((val < 0.0) ? -1.0 : 0.0);  // fixed node (dmc.py:64-66)
          accd = ratio > uacc;
          if (DMC && r == 0) {
            const double rx = z0 + d0, ry = z1 + d1, rz = z2 + d2, r2 = rx * rx + ry * ry + rz * rz;
            ws[13] += r2;
            if (accd) ws[14] += r2;
          }
        }
        if (live && r == 0 && mb.accept_rec) mb.accept_rec[(size_t)e * W + wg] = accd;
        PQA_R8CLK(4);
        if (accd) {
          // Sherman-Morrison on the register rows (slater.py:88-94): R = T_old[i] / ratio, T[j] -= R (V . T[j]), T[i] = R;
          // the row's dot product in the PQA_ROWDOT order of the lane-per-walker kernels
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
          PQA_R8CLK(13);
          if (r == 0) {  // sign and log of the determinant: running product of |ratio|, its logarithm taken when it leaves [1e-60, 1e60]
            const double mag = fabs(dr);
            ws[10] *= (dr > 0.0) ? 1.0 : ((dr < 0.0) ? -1.0 : dr);
            double lpr = ws[12] * mag;
            if (!(lpr > 1e-60 && lpr < 1e60)) { ws[11] += log(lpr); lpr = 1.0; }
            ws[12] = lpr;
            ws[15] += 1.0;
          }
          if (r == i) {
            if (s) { cx[1] = npx; cy[1] = npy; cz[1] = npz; } else { cx[0] = npx; cy[0] = npy; cz[0] = npz; }
          }
          // the proposal's rows become the cached rows of electron i: into the walker's other slot, selector flipped
          const int cur = __shfl(selr, (lane & 32) | i, 64);
          if (live && r < nmo && PQA_R8_ON(16)) {
            double* out = rcs + (((size_t)i * 2 + (cur ^ 1)) * W + wg) * 5 * nmo;
#pragma unroll
            for (int c = 0; c < 5; ++c) out[c * nmo + r] = rn[c * 32 + r];
          }
          if (r == i) {
            selr = cur ^ 1;
            if (live) sels[(size_t)i * W + wg] = (uint8_t)selr;
          }
        }
        PQA_R8CLK(5);
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
        PQA_R8CLK(10);
        double gx, gy, gz;
        {
          const double te = rowE[wl * 32 + r];
          double q1 = ro[1] * te, q2 = ro[2] * te, q3 = ro[3] * te;
          q1 = res_sum32(q1); q2 = res_sum32(q2); q3 = res_sum32(q3);
          if (PQA_R8_UNIT0) {
            // The reference divides the gradient sums by the value sum (slater.py gradient: ratios[1:] / ratios[0]), which at the electron's own
            // position is row e of the Slater matrix times column e of its inverse: 1 up to the rounding of the inverse.  Taken as 1: one load,
            // one 32-lane sum and a division less per move (-DPQA_R8_UNIT0=0 restores them; the drift changes by that rounding, ~1e-13 relative).
            gx = finite_or(q1, 0.0); gy = finite_or(q2, 0.0); gz = finite_or(q3, 0.0);
          } else {
            double q0 = ro[0] * te;
            q0 = res_sum32(q0);
            const double iq0 = 1.0 / q0;
            gx = finite_or(q1 * iq0, 0.0); gy = finite_or(q2 * iq0, 0.0); gz = finite_or(q3 * iq0, 0.0);
          }
        }
        const int src = (lane & 32) | ip;
        const double pox = __shfl(s ? cx[1] : cx[0], src, 64), poy = __shfl(s ? cy[1] : cy[0], src, 64), poz = __shfl(s ? cz[1] : cz[0], src, 64);
        double U0 = 0.0;
        PQA_R8CLK(11);
        if (has_jastrow && i >= 0) {  // (summed ahead for both outcomes of the decision: wsc 20..23 rejected, 24..27 accepted)
          const double* jq = ws + (accd ? 24 : 20);
          U0 = jq[0]; gx += jq[1]; gy += jq[2]; gz += jq[3];
        } else if (has_jastrow) {
          ResJ jo{0.0, 0.0, 0.0, 0.0};
          if (S.jq_on) res_jas_m<false>(S, r, cx, cy, cz, at_xyz, acoef, aql, jt, ep, pox, poy, poz, jo);
          else {
            double g3[3];
            res_jas_part<false>(S, ep, r, pox, poy, poz, cx, cy, cz, at_xyz, acoef, aql, jo.u, g3);
            jo.x = g3[0]; jo.y = g3[1]; jo.z = g3[2];
          }
          jo.u = res_sum32(jo.u); jo.x = res_sum32(jo.x); jo.y = res_sum32(jo.y); jo.z = res_sum32(jo.z);
          U0 = jo.u; gx += jo.x; gy += jo.y; gz += jo.z;
        }
        PQA_R8CLK(12);
        if (DMC) limdrift_dmc(gx, gy, gz, mb.tstep); else r8_limdrift3(gx, gy, gz);
        const double sq = jt[3 * PQA_JQ + 15], df = DMC ? 1.0 : mb.tstep;
        const double z0 = g0 * sq, z1 = g1 * sq, z2 = g2 * sq;
        if (r == 0) {
          ws[0] = pox + z0 + gx * df; ws[1] = poy + z1 + gy * df; ws[2] = poz + z2 + gz * df;
          ws[3] = z0; ws[4] = z1; ws[5] = z2; ws[6] = gx; ws[7] = gy; ws[8] = gz; ws[9] = U0;
        }
        res_wave_sync();
        PQA_R8CLK(6);
      }
    }
    PQA_R8T(2 + 4 * s);
    // ---- this spin's state back to the planes
    if (live && r < n) {
#pragma unroll
      for (int k8 = 0; k8 < 4; ++k8) {
#pragma unroll
        for (int k = 8 * k8; k < 8 * k8 + 8; ++k)
          if (k < n) Tg[((size_t)r * n + k) * W] = t[k];
        __builtin_amdgcn_sched_barrier(0);
      }
    }
    if (live && r == 0) {
      (s ? L.dsign[1] : L.dsign[0])[wg] = ws[10];
      (s ? L.dlog[1] : L.dlog[0])[wg] = ws[11] + log(ws[12]);
    }
    PQA_R8T(3 + 4 * s);
    __syncthreads();  // (rowE / region reads of this spin's last decision before the next spin's first proposal)
    PQA_R8T(4 + 4 * s);
  }
  PQA_RCLK(15);
  if (live) {
#pragma unroll
    for (int q = 0; q < 2; ++q) {
      const int j = (q ? S.nup : 0) + r;
      if (r < (q ? S.ndn : S.nup)) {
        double* xj = L.xt + (size_t)j * 3 * W + wg;
        xj[0] = cx[q]; xj[W] = cy[q]; xj[2 * W] = cz[q];
        if (RT.xaos) {
          double* xa = RT.xaos + ((size_t)wg * S.nelec + j) * 3;
          xa[0] = cx[q]; xa[1] = cy[q]; xa[2] = cz[q];
        }
      }
    }
    if (r == 0) {
      mb.acc_w[wg] += (int)ws[15];
      if (DMC) { mb.r2_prop[wg] += ws[13]; mb.r2_acc[wg] += ws[14]; }
    }
  }
}
