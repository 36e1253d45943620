// pqa_ww.hpp — the wave-per-walker electron sweep in ONE launch.
//
// Wave functions outside the lane-per-walker kernels' scope (multi-determinant expansions, three-body Jastrow factors) move an electron
// with k_propose -> orbital kernel -> k_accept (pqa_vmc.hpp): three launches per move, each a single wave per walker running one dependent
// chain after the other — for the 50-determinant water molecule of BASELINE config C4 (2 048 walkers, 8 electrons) 21 + 6 + 27 us per move
// of which the chains are 14 and 22 (tools/scratch/ww_clk.py: Slater terms 4.6, two-body Jastrow 3.4, three-body 5.9; at the proposal 4.1,
// 9.3 + decision, commit 7.6).  Here a block owns the walker for the whole sweep:
//   phase A  Slater drift at the current position (cached orbital row), one/two-body Jastrow, three-body Jastrow
//   phase B  the proposal (mc.py:117-121, dmc.py:46-52)
//   phase C  the proposal's orbital row (AOs of ONE point: lanes over shells, then lanes over (component, orbital) dot products against
//            the block's LDS copy of the coefficient matrix) and the Slater ratios; the Jastrow terms at the proposal
//   phase D  Metropolis test (mc.py:124-132, dmc.py:57-70); accepted: the determinants' Sherman-Morrison updates (slater.py:88-94),
//            cache row, coordinate
// NWV = 1 (default for small shards): one wave per walker runs the parts one after the other — the launches' chains without the 3 N launches,
// their drain / fill and the separate orbital kernel: C4 step 0.884 -> 0.801 ms at 2 048 walkers (2.32 -> 2.56 M walker-steps/s).
// NWV = 3 (PQA_WW=3): the three parts of phases A and C on three waves side by side, the determinant updates dealt to the waves.  MEASURED
// (tools/scratch/ww1_clk.py, profiles/r05_ww_one_launch.txt): a move takes 23 us in a block that has the CU to itself (phase A 7.4, proposal 1,
// phase C 10.7 of which the orbital row 6.2 before the LDS copy of the coefficients, decision + commit 4) — but 6 144 waves at 168 registers
// are two rounds of 12 waves per CU, and with every SIMD holding three waves the same move takes ~37 us: 575 us per sweep against 431 us of the
// launches.  The three chains of a move are not latency alone — two waves per SIMD already keep the fp64 pipe about half busy — so splitting
// them over waves buys less than the registers and barriers cost.  Kept as the measured answer to the round-4 review's "fuse k_propose ->
// k_accept -> next k_propose into one persistent launch".
// The device functions are the ones k_propose / k_accept call, in the same order of operations — the same numbers except for the orbital
// row, whose contraction is a sequential sum here and an MFMA tile sum in the orbital kernels (relative differences of 1e-16).
// Scope: open boundary conditions, real orbitals, l <= 5.  Requires PQA_WSYNC to be the wave-level fence (pqa_sweep_ww.hip).
#pragma once
#include "pqa_ao.hpp"
#include "pqa_vmc.hpp"

#ifdef PQA_WW_CLK  // timing build only (tools/scratch/ww1_clk.py): 100 MHz stamps of lane 0 of each wave of the first 256 blocks, last move
static __device__ unsigned long long pqa_ww1_clk[256 * 3 * 8];
#define PQA_W1CLK(k) do { if (blockIdx.x < 256 && (threadIdx.x & 63) == 0) pqa_ww1_clk[(blockIdx.x * 3 + (threadIdx.x >> 6)) * 8 + (k)] = wall_clock64(); } while (0)
#else
#define PQA_W1CLK(k) do { } while (0)
#endif
#define PQA_WW_XCH 32  // doubles of the exchange area in front of the orbital row

// Orbital row [5][nmo] of spin s at one point by one wave.  aov: [5][nao] LDS scratch.
// C: the coefficient matrix [nao][nmo] of the spin (the block's LDS copy where it fits).
template <int LMAX>
__device__ __forceinline__ void ww_orb_point(const SysDev& S, int s, double x, double y, double z, double* __restrict__ aov,
                                             double* __restrict__ row, const double* __restrict__ C) {
  const int lane = threadIdx.x & 63, nao = S.nao, nmo = S.nmo[s];
  for (int sh = lane; sh < S.nshell; sh += 64) {
    const int ia = S.shell_atom[sh], q0 = S.shell_prim_off[sh], np_ = S.shell_prim_off[sh + 1] - q0;
    double* a = aov + S.shell_ao_off[sh];
    shell_eval<5, LMAX>(S.shell_l[sh], x - S.atom_xyz[3 * ia], y - S.atom_xyz[3 * ia + 1], z - S.atom_xyz[3 * ia + 2], S.prim_exp + q0,
                        S.prim_coef + q0, np_, [&](int m, double v, double gx, double gy, double gz, double lp) {
                          a[m] = v; a[nao + m] = gx; a[2 * nao + m] = gy; a[3 * nao + m] = gz; a[4 * nao + m] = lp;
                        });
  }
  PQA_WSYNC();
  for (int idx = lane; idx < 5 * nmo; idx += 64) {
    const int c = idx / nmo, j = idx - c * nmo;
    const double* __restrict__ av = aov + (size_t)c * nao;
    const double* __restrict__ cj = C + j;
    double acc = 0.0;
    int k = 0;
    for (; k + 4 <= nao; k += 4) {  // (four products in flight; the sum itself stays sequential)
      const double a0 = av[k], a1 = av[k + 1], a2 = av[k + 2], a3 = av[k + 3];
      const double c0 = cj[(size_t)k * nmo], c1 = cj[(size_t)(k + 1) * nmo], c2 = cj[(size_t)(k + 2) * nmo], c3 = cj[(size_t)(k + 3) * nmo];
      acc += a0 * c0; acc += a1 * c1; acc += a2 * c2; acc += a3 * c3;
    }
    for (; k < nao; ++k) acc += av[k] * cj[(size_t)k * nmo];
    row[idx] = acc;
  }
  PQA_WSYNC();
}

// grid = W blocks of 192 threads.  Dynamic LDS: [0, xoff) the scratch of the Slater functions and (from S.j3_off) of the three-body
// Jastrow term, then PQA_WW_XCH exchange doubles, the orbital row [5][max nmo] and the AO values [5][nao].
// NWV = 3: the three roles on three waves; NWV = 1: one wave per walker runs them one after the other (the launches' chains without the
// launches: the orbital row from the block itself, no drain / fill between the parts of a move)
template <int LMAX, int NWV>
static __global__ __launch_bounds__(64 * NWV, NWV == 3 ? 3 : 2) void k_sweep_ww(SysDev S, SlaterState st, JastrowState js, MoveBuf mb, int has_slater,
                                                                  int has_jastrow, int xoff, int cstage, long W) {
  extern __shared__ double lds[];
  const long w = blockIdx.x;
  const int tid = threadIdx.x, lane = tid & 63, wv = __builtin_amdgcn_readfirstlane(tid >> 6);
  // exchange area: [0..4] / [16..20] Slater terms (gx, gy, gz, val2, sign) at the current position / at the proposal, [5..8] / [21..24]
  // one/two-body Jastrow (U, g), [9..12] / [25..28] three-body, [13..15] the scaled gaussians
  double* xch = lds + xoff;
  double* row = xch + PQA_WW_XCH;
  const int nmo_max = S.nmo[0] > S.nmo[1] ? S.nmo[0] : S.nmo[1];
  double* aov = row + 5 * (size_t)nmo_max;
  double* cl = aov + 5 * (size_t)S.nao;  // cstage: the coefficient matrices of both spins [nao][nmo_up], [nao][nmo_dn]
  if (cstage) {
    const int n0 = S.nao * S.nmo[0], n1 = S.nao * S.nmo[1];
    for (int k = tid; k < n0; k += 64 * NWV) cl[k] = S.mo[0][k];
    for (int k = tid; k < n1; k += 64 * NWV) cl[n0 + k] = S.mo[1][k];
    __syncthreads();
  }
  const int N = S.nelec;
  const bool has_j3 = has_jastrow && S.na3 > 0;
  double* xg = js.x + (size_t)w * N * 3;
#pragma unroll 1
  for (int e = 0; e < N; ++e) {
    const double* xw = xg;
    asm volatile("" : "+s"(xw) :: "memory");  // (the coordinates change under the loop: no load of them may be carried across a move)
    const int s = e >= S.nup, i = e - s * S.nup, n = s ? S.ndn : S.nup, nmo = S.nmo[s];
    const double ex = xw[3 * e], ey = xw[3 * e + 1], ez = xw[3 * e + 2];
    double dgx = 0.0, dgy = 0.0, dgz = 0.0, U0 = 0.0, z0 = 0.0, z1 = 0.0, z2 = 0.0, nx = ex, ny = ey, nz = ez;
#pragma unroll 1
    for (int ph = 0; ph < 2; ++ph) {  // phase A (current position), phase C (proposal): the three parts side by side
      double* xo = xch + 16 * ph;
      PQA_W1CLK(ph ? 3 : 0);
      if (wv == 0) {  // (NWV 1: the only wave)
        double gx = 0.0, gy = 0.0, gz = 0.0, v2 = 1.0, sgn = 1.0;
        if (has_slater) {
          const double* r_ = st.cache[s] + ((size_t)w * n + i) * 5 * nmo;
          if (ph) { ww_orb_point<LMAX>(S, s, nx, ny, nz, aov, row, cstage ? cl + (s ? S.nao * S.nmo[0] : 0) : S.mo[s]); r_ = row; PQA_W1CLK(7); }
          slater_move_terms<false>(S, st, s, i, w, r_, lds, gx, gy, gz, v2, &sgn);
        }
        if (lane == 0) {
          xo[0] = gx; xo[1] = gy; xo[2] = gz; xo[3] = v2; xo[4] = sgn;
          if (ph == 0) {
            double g0, g1, g2, g3;
            if (mb.gauss) {
              const double* zt = mb.gauss + ((size_t)e * W + w) * 3;
              g0 = zt[0]; g1 = zt[1]; g2 = zt[2];
            } else {
              normal2(philox(mb.seed, (uint32_t)w, (uint32_t)e, PQA_STREAM_GAUSS_A, mb.step), g0, g1);
              normal2(philox(mb.seed, (uint32_t)w, (uint32_t)e, PQA_STREAM_GAUSS_B, mb.step), g2, g3);
            }
            const double sq = sqrt(mb.tstep);
            xch[13] = g0 * sq; xch[14] = g1 * sq; xch[15] = g2 * sq;
          }
        }
      }
      if (NWV == 1) {
        double U = 0.0, g[3] = {0.0, 0.0, 0.0}, lp;
        if (has_jastrow) jas_eval<1, false>(S, xw, e, nx, ny, nz, U, g, lp, 1, nullptr);
        if (lane == 0) { xo[5] = U; xo[6] = g[0]; xo[7] = g[1]; xo[8] = g[2]; }
        U = 0.0; g[0] = g[1] = g[2] = 0.0;
        if (has_j3) jas_eval<1, false>(S, xw, e, nx, ny, nz, U, g, lp, 2, lds + S.j3_off);
        if (lane == 0) { xo[9] = U; xo[10] = g[0]; xo[11] = g[1]; xo[12] = g[2]; }
      } else if (wv != 0) {
        double U = 0.0, g[3] = {0.0, 0.0, 0.0}, lp;
        if (wv == 1) { if (has_jastrow) jas_eval<1, false>(S, xw, e, nx, ny, nz, U, g, lp, 1, nullptr); }
        else if (has_j3) jas_eval<1, false>(S, xw, e, nx, ny, nz, U, g, lp, 2, lds + S.j3_off);
        if (lane == 0) { double* o = xo + (wv == 1 ? 5 : 9); o[0] = U; o[1] = g[0]; o[2] = g[1]; o[3] = g[2]; }
      }
      PQA_W1CLK(ph ? 4 : 1);
      __syncthreads();
      PQA_W1CLK(ph ? 5 : 2);
      if (ph == 0) {
        // ---- phase B: the proposal (every thread; k_propose's arithmetic: the three-body terms join the two-body ones, the sum the Slater part)
        double j0 = xch[6], j1 = xch[7], j2 = xch[8];
        U0 = xch[5];
        if (has_j3) { U0 += xch[9]; j0 += xch[10]; j1 += xch[11]; j2 += xch[12]; }
        dgx = xch[0]; dgy = xch[1]; dgz = xch[2];
        if (has_jastrow) { dgx += j0; dgy += j1; dgz += j2; }
        if (mb.dmc) limdrift_dmc(dgx, dgy, dgz, mb.tstep); else limdrift3(dgx, dgy, dgz);
        z0 = xch[13]; z1 = xch[14]; z2 = xch[15];
        const double df = mb.dmc ? 1.0 : mb.tstep;
        nx = ex + z0 + dgx * df; ny = ey + z1 + dgy * df; nz = ez + z2 + dgz * df;
      }
    }
    // ---- phase D: decision (every thread; k_accept's arithmetic), commit
    bool acc;
    {
      double val2 = xch[19], gx = xch[16], gy = xch[17], gz = xch[18];
      if (has_jastrow) {
        double U = xch[21], j0 = xch[22], j1 = xch[23], j2 = xch[24];
        if (has_j3) { U += xch[25]; j0 += xch[26]; j1 += xch[27]; j2 += xch[28]; }
        gx += j0; gy += j1; gz += j2;
        const double ej = exp(U - U0);
        val2 *= ej * ej;
      }
      double bx, by, bz;
      if (mb.dmc) {
        limdrift_dmc(gx, gy, gz, mb.tstep);
        bx = z0 + dgx + gx; by = z1 + dgy + gy; bz = z2 + dgz + gz;
      } else {
        limdrift3(gx, gy, gz);
        bx = z0 + mb.tstep * (dgx + gx); by = z1 + mb.tstep * (dgy + gy); bz = z2 + mb.tstep * (dgz + gz);
      }
      const double fwd = z0 * z0 + z1 * z1 + z2 * z2;
      const double bwd = bx * bx + by * by + bz * bz;
      const double t_prob = exp(1.0 / (2.0 * mb.tstep) * (fwd - bwd));
      double ratio = val2 * t_prob;
      if (mb.dmc) ratio *= xch[20];  // fixed node (dmc.py:64-66)
      double u;
      if (mb.unif) u = mb.unif[(size_t)e * W + w];
      else {
        const Philox p = philox(mb.seed, (uint32_t)w, (uint32_t)e, PQA_STREAM_ACCEPT, mb.step);
        u = u01(p.c[0], p.c[1]);
      }
      acc = ratio > u;
    }
    if (tid == 0) {
      if (mb.dmc) {
        const double r2 = (z0 + dgx) * (z0 + dgx) + (z1 + dgy) * (z1 + dgy) + (z2 + dgz) * (z2 + dgz);
        mb.r2_prop[w] += r2;
        if (acc) mb.r2_acc[w] += r2;
      }
      mb.accept[w] = acc;
      if (mb.accept_rec) mb.accept_rec[(size_t)e * W + w] = acc;
      if (acc) mb.acc_w[w] += 1;
    }
    if (acc) {
      if (has_slater) {
        sm_update_wave(S, st, s, i, w, row, lds, wv, NWV);
        double* c = st.cache[s] + ((size_t)w * n + i) * 5 * nmo;
        for (int k = tid; k < 5 * nmo; k += 64 * NWV) c[k] = row[k];
      }
      if (tid == (NWV == 3 ? 64 : 0)) { xg[3 * e] = nx; xg[3 * e + 1] = ny; xg[3 * e + 2] = nz; }
    }
    PQA_W1CLK(6);
    __syncthreads();  // (the coordinate, the inverses and the exchange area before the next electron)
  }
}
