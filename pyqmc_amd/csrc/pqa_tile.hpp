// pqa_tile.hpp — the whole VMC / DMC electron sweep of a walker tile in ONE kernel, state on chip.
//
// The lane-per-walker sweep (pqa_lw.hpp) launches six kernels per electron and streams every walker's inverse through
// HBM once per move (k_commit_lw, 31 % of the step) and its coordinates and orbital rows twice (k_move_part_lw, 25 %).
// Walkers are independent Markov chains, so nothing forces a global synchronisation between electrons: here a block of
// 16 waves owns 16 walkers for the whole sweep,
//   * wave = walker: the transposed inverse T[j][k] of the current spin lives in REGISTERS (lane = electron row j and
//     column half h, 16 doubles per lane for n <= 32), the walker's coordinates in LDS;
//   * block = one 16-point MFMA tile: the orbital rows at the 16 proposals are evaluated cooperatively exactly like
//     k_orb does it (phase 1: thread = (shell, point) -> LDS AO tile; phase 2: v_mfma_f64_16x16x4_f64 against the padded
//     coefficient matrix), the result never leaves LDS;
//   * per electron: drift from the cached orbital row (HBM, 1.3 KB) and the register inverse, Jastrow sums from LDS
//     coordinates, proposal, orbitals, Metropolis, Sherman-Morrison update in registers, cache row of accepted moves.
// HBM traffic per walker-step drops from ~1.4 MB to ~0.25 MB and the ~400 launches of the sweep become one.
// Arithmetic per walker is the reference's (mc.py:112-137 / dmc.py:38-70); sums run in a different order than in the
// other two kernel families, so trajectories agree to rounding, decisions are identical off measure-zero ties.
// Scope: open systems, real single-determinant Slater factor with n_up, n_dn <= 32 and <= 32 orbitals per spin,
// optional two-body Jastrow (no three-body factor).
#pragma once
#include "pqa_ao.hpp"
#include "pqa_jastrow.hpp"
#include "pqa_vmc.hpp"

#define PQA_TILE_NW 16      // walkers (waves) per block
#define PQA_TILE_KT 96      // AO rows per pass of the orbital evaluation
#define PQA_TILE_MAXPASS 8

struct TileTab {
  int npass;
  int pass_chunk[PQA_TILE_MAXPASS + 1];  // chunk range of each pass (chunks of the ncomp = 5 table, rows <= KT per pass)
  int nmo_pad;                           // 16 * nt
};

// dynamic LDS layout (doubles unless noted)
struct TileLds {
  double* xs;      // [NW][3 N]        coordinates
  double* tile;    // [5][KT][16]      AO tile of one pass
  double* rnew;    // [NW][5][nmo_pad] orbital rows at the proposals
  double* rold;    // [NW][5][nmo_pad] cached rows of the current positions (staging)
  double* sh_xyz;  // [nshell][3]
  double* pr_exp;  // [nprim]
  double* pr_coef; // [nprim]
  int* sh_meta;    // [nshell][4]: l, nprim, first primitive, tile row (chunk row0 + row in chunk)
  int* occ;        // [2][32]
};
__host__ __device__ inline size_t tile_lds_bytes(int N, int nmo_pad, int nshell, int nprim) {
  size_t d = (size_t)PQA_TILE_NW * 3 * N + 5 * PQA_TILE_KT * 16 + 2 * (size_t)PQA_TILE_NW * 5 * nmo_pad + 3 * (size_t)nshell + 2 * (size_t)nprim;
  return d * sizeof(double) + ((size_t)4 * nshell + 64) * sizeof(int);
}

// dot of an orbital row (slot order through occ) with the register inverse: returns sum_k row[occ[k]] T[j][k] for the
// lane's own row j (both halves of the wave hold the full sum)
__device__ __forceinline__ double tile_rowdot(const double* __restrict__ row, const int* __restrict__ occ, const double (&t)[16], int h) {
  double s = 0.0;
#pragma unroll
  for (int q = 0; q < 16; ++q) s += row[occ[16 * h + q]] * t[q];
  return s + __shfl_xor(s, 32, 64);
}

template <bool DMC>
__global__ __launch_bounds__(1024) void k_sweep_tile(SysDev S, SlaterState st, JastrowState js, MoveBuf mb, ChunkTab T, TileTab TT,
                                                     int has_jastrow, long W) {
  extern __shared__ double lds_raw[];
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  const int N = S.nelec, nmo_pad = TT.nmo_pad;
  TileLds L;
  L.xs = lds_raw;
  L.tile = L.xs + (size_t)PQA_TILE_NW * 3 * N;
  L.rnew = L.tile + 5 * PQA_TILE_KT * 16;
  L.rold = L.rnew + (size_t)PQA_TILE_NW * 5 * nmo_pad;
  L.sh_xyz = L.rold + (size_t)PQA_TILE_NW * 5 * nmo_pad;
  L.pr_exp = L.sh_xyz + 3 * (size_t)S.nshell;
  L.pr_coef = L.pr_exp + S.nprim;
  L.sh_meta = (int*)(L.pr_coef + S.nprim);
  L.occ = L.sh_meta + 4 * (size_t)S.nshell;

  const long w_raw = (long)blockIdx.x * PQA_TILE_NW + wv;
  const bool live = w_raw < W;
  const long w = live ? w_raw : W - 1;  // tail waves shadow the last walker and write nothing
  // ---- stage tables, coordinates; clear the AO tile (its K-padding rows are never written again)
  for (int sh = tid; sh < S.nshell; sh += 1024) {
    const int ia = S.shell_atom[sh];
    L.sh_xyz[3 * sh] = S.atom_xyz[3 * ia]; L.sh_xyz[3 * sh + 1] = S.atom_xyz[3 * ia + 1]; L.sh_xyz[3 * sh + 2] = S.atom_xyz[3 * ia + 2];
    L.sh_meta[4 * sh] = S.shell_l[sh];
    L.sh_meta[4 * sh + 1] = S.shell_prim_off[sh + 1] - S.shell_prim_off[sh];
    L.sh_meta[4 * sh + 2] = S.shell_prim_off[sh];
    L.sh_meta[4 * sh + 3] = T.shell_kb[sh];  // row inside its chunk; the chunk's row0 is added per pass
  }
  for (int p = tid; p < S.nprim; p += 1024) { L.pr_exp[p] = S.prim_exp[p]; L.pr_coef[p] = S.prim_coef[p]; }
  for (int k = tid; k < 64; k += 1024) {
    const int s = k >> 5, q = k & 31, n = s ? S.ndn : S.nup;
    L.occ[k] = q < n ? S.det_occ[s][q] : 0;
  }
  for (int k = tid; k < 5 * PQA_TILE_KT * 16; k += 1024) L.tile[k] = 0.0;
  double* xw = L.xs + (size_t)wv * 3 * N;
  {
    const double* xg = js.x + (size_t)w * N * 3;
    for (int k = lane; k < 3 * N; k += 64) xw[k] = xg[k];
  }
  __syncthreads();

  const int j = lane & 31, h = lane >> 5;
  double* rnew = L.rnew + (size_t)wv * 5 * nmo_pad;
  double* rold = L.rold + (size_t)wv * 5 * nmo_pad;
  const int i16 = lane & 15, kq = lane >> 4;
  const int mc = wv % 5, mu = wv / 5;  // MFMA role of waves 0..9: component mc, orbital tile mu
  int n_acc = 0;
  double r2p = 0.0, r2a = 0.0;

  for (int s = 0; s < 2; ++s) {
    const int n = s ? S.ndn : S.nup, nmo = S.nmo[s];
    if (n == 0) continue;
    const int* occ = L.occ + 32 * s;
    // ---- the transposed inverse of this spin into registers: lane (j, h) holds T[j][16h .. 16h+15]
    double t[16];
    {
      const double* Tg = st.T[s] + (size_t)w * n * n;
#pragma unroll
      for (int q = 0; q < 16; ++q) t[q] = (j < n && 16 * h + q < n) ? Tg[(size_t)j * n + 16 * h + q] : 0.0;
    }
    double dsign = st.dsign[s][w], dlog = st.dlog[s][w];
    const double* __restrict__ C = T.cpad[s];
    const int ldc = T.ldc[s];
    const bool mfma_wave = wv < 10 && 16 * mu < nmo_pad;

    for (int i = 0; i < n; ++i) {
      const int e = s * S.nup + i;
      // ================= proposal: drift at the current position
      {
        const double* cg = st.cache[s] + ((size_t)w * n + i) * 5 * nmo;
        for (int k = lane; k < 5 * nmo; k += 64) rold[(k / nmo) * nmo_pad + (k % nmo)] = cg[k];
      }
      const double ex = xw[3 * e], ey = xw[3 * e + 1], ez = xw[3 * e + 2];
      double gx, gy, gz, U0 = 0.0;
      {
        const double r0 = __shfl(tile_rowdot(rold, occ, t, h), i, 64);
        const double r1 = __shfl(tile_rowdot(rold + nmo_pad, occ, t, h), i, 64);
        const double r2 = __shfl(tile_rowdot(rold + 2 * nmo_pad, occ, t, h), i, 64);
        const double r3 = __shfl(tile_rowdot(rold + 3 * nmo_pad, occ, t, h), i, 64);
        gx = finite_or(r1 / r0, 0.0); gy = finite_or(r2 / r0, 0.0); gz = finite_or(r3 / r0, 0.0);
      }
#ifndef PQA_TILE_ABL_NOJAS
      if (has_jastrow) {
        double g[3], lp;
        jas_eval<1, false>(S, xw, e, ex, ey, ez, U0, g, lp, 1);
        gx += g[0]; gy += g[1]; gz += g[2];
      }
#endif
      if (DMC) limdrift_dmc(gx, gy, gz, mb.tstep); else limdrift3(gx, gy, gz);
      double z0, z1, z2, z3;
      if (mb.gauss) {
        const double* zt = mb.gauss + ((size_t)e * W + w) * 3;
        z0 = zt[0]; z1 = zt[1]; z2 = zt[2];
      } else {
        normal2(philox(mb.seed, (uint32_t)w, (uint32_t)e, PQA_STREAM_GAUSS_A, mb.step), z0, z1);
        normal2(philox(mb.seed, (uint32_t)w, (uint32_t)e, PQA_STREAM_GAUSS_B, mb.step), z2, z3);
      }
      const double sq = sqrt(mb.tstep), df = DMC ? 1.0 : mb.tstep;
      z0 *= sq; z1 *= sq; z2 *= sq;
      const double nx = ex + z0 + gx * df, ny = ey + z1 + gy * df, nz = ez + z2 + gz * df;
      if (lane == 0) { rnew[0] = nx; rnew[1] = ny; rnew[2] = nz; }  // the proposal travels through rnew's first slots
      __syncthreads();
      // ================= orbital rows at the 16 proposals of the block (k_orb's two phases on a 16-point tile)
      double px, py, pz;
      {
        const double* pr = L.rnew + (size_t)(tid & 15) * 5 * nmo_pad;
        px = pr[0]; py = pr[1]; pz = pr[2];
      }
      __syncthreads();  // every thread holds its point: rnew may now be overwritten
      d4 acc = (d4){0.0, 0.0, 0.0, 0.0};
      for (int ps = 0; ps < TT.npass; ++ps) {
        const int ch0 = TT.pass_chunk[ps], ch1 = TT.pass_chunk[ps + 1];
        const int row_base = T.chunk_row0[ch0];
        const int s_lo = T.cw_off[0][4 * ch0], s_hi = T.cw_off[0][4 * ch1];
#ifndef PQA_TILE_ABL_NOAO
        for (int it = tid; it < (s_hi - s_lo) * 16; it += 1024) {
          const int sh = T.cw_shell[0][s_lo + (it >> 4)];
          // chunk of the shell: its rows start at chunk_row0[chunk]; shell_kb is relative to the chunk.  The shells of a
          // pass are listed chunk by chunk, so the chunk index is recovered from the running offsets.
          int ch = ch0;
          while (s_lo + (it >> 4) >= T.cw_off[0][4 * (ch + 1)]) ++ch;
          const int krow = T.chunk_row0[ch] - row_base + L.sh_meta[4 * sh + 3];
          const int l_ = L.sh_meta[4 * sh], np_ = L.sh_meta[4 * sh + 1], q0 = L.sh_meta[4 * sh + 2];
          const int pl = it & 15;
          shell_eval<5>(l_, px - L.sh_xyz[3 * sh], py - L.sh_xyz[3 * sh + 1], pz - L.sh_xyz[3 * sh + 2], L.pr_exp + q0, L.pr_coef + q0, np_,
                        [&](int m, double v, double ax, double ay, double az, double lp) {
                          double* tl = L.tile + (size_t)(krow + m) * 16 + pl;
                          tl[0] = v; tl[PQA_TILE_KT * 16] = ax; tl[2 * PQA_TILE_KT * 16] = ay; tl[3 * PQA_TILE_KT * 16] = az;
                          tl[4 * PQA_TILE_KT * 16] = lp;
                        });
        }
#endif
        __syncthreads();
#ifndef PQA_TILE_ABL_NOMFMA
        if (mfma_wave) {
          const int nrow = T.chunk_row0[ch1 - 1] + ((T.chunk_nk[ch1 - 1] + 3) & ~3) - row_base;  // padded rows of this pass
          const double* a_ = L.tile + (size_t)mc * PQA_TILE_KT * 16 + (size_t)kq * 16 + i16;
          const double* b_ = C + (size_t)(row_base + kq) * ldc + 16 * mu + i16;
#pragma unroll 4
          for (int ks = 0; ks < nrow / 4; ++ks)
            acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a_[(size_t)ks * 64], b_[(size_t)ks * 4 * ldc], acc, 0, 0, 0);
        }
#endif
        __syncthreads();
      }
      if (mfma_wave) {  // lane holds D[point = kq + 4r][orbital = 16 mu + i16]
#pragma unroll
        for (int r = 0; r < 4; ++r) L.rnew[((size_t)(kq + 4 * r) * 5 + mc) * nmo_pad + 16 * mu + i16] = acc[r];
      }
      __syncthreads();
      // ================= Metropolis at the proposal
      const double tmp0 = tile_rowdot(rnew, occ, t, h);  // sum_k V[k] T[j][k] for the lane's row: reused by the update
      double val2, sgn = 1.0, hx, hy, hz;
      const double dr = __shfl(tmp0, i, 64);
      {
        const double r1 = __shfl(tile_rowdot(rnew + nmo_pad, occ, t, h), i, 64);
        const double r2 = __shfl(tile_rowdot(rnew + 2 * nmo_pad, occ, t, h), i, 64);
        const double r3 = __shfl(tile_rowdot(rnew + 3 * nmo_pad, occ, t, h), i, 64);
        hx = finite_or(r1 / dr, 0.0); hy = finite_or(r2 / dr, 0.0); hz = finite_or(r3 / dr, 0.0);
        const double v = finite_or(dr, 1.0);
        val2 = v * v;
        sgn = (v > 0.0) ? 1.0 : ((v < 0.0) ? -1.0 : 0.0);
      }
#ifndef PQA_TILE_ABL_NOJAS
      if (has_jastrow) {
        double g[3], lp, U;
        jas_eval<1, false>(S, xw, e, nx, ny, nz, U, g, lp, 1);
        hx += g[0]; hy += g[1]; hz += g[2];
        const double ej = exp(U - U0);
        val2 *= ej * ej;
      }
#endif
      double bx, by, bz;
      if (DMC) {
        limdrift_dmc(hx, hy, hz, mb.tstep);
        bx = z0 + gx + hx; by = z1 + gy + hy; bz = z2 + gz + hz;
      } else {
        limdrift3(hx, hy, hz);
        bx = z0 + mb.tstep * (gx + hx); by = z1 + mb.tstep * (gy + hy); bz = z2 + mb.tstep * (gz + hz);
      }
      const double fwd = z0 * z0 + z1 * z1 + z2 * z2, bwd = bx * bx + by * by + bz * bz;
      double ratio = val2 * exp(1.0 / (2.0 * mb.tstep) * (fwd - bwd));
      if (DMC) ratio *= sgn;
      double u;
      if (mb.unif) u = mb.unif[(size_t)e * W + w];
      else {
        const Philox p = philox(mb.seed, (uint32_t)w, (uint32_t)e, PQA_STREAM_ACCEPT, mb.step);
        u = u01(p.c[0], p.c[1]);
      }
      const bool accd = ratio > u;  // wave-uniform: every lane computed the same numbers
      if (DMC) {
        const double r2 = (z0 + gx) * (z0 + gx) + (z1 + gy) * (z1 + gy) + (z2 + gz) * (z2 + gz);
        r2p += r2;
        if (accd) r2a += r2;
      }
      if (live && lane == 0 && mb.accept_rec) mb.accept_rec[(size_t)e * W + w] = accd;
      if (accd) {
        ++n_acc;
        // Sherman-Morrison in registers: R[k] = T[i][k] / ratio, T[j][k] -= R[k] tmp[j] (j != i), T[i][k] = R[k]
        const double inv = 1.0 / dr;
#pragma unroll
        for (int q = 0; q < 16; ++q) {
          const double R = __shfl(t[q], i + 32 * h, 64) * inv;
          t[q] = (j == i) ? R : t[q] - R * tmp0;
        }
        dsign *= (dr > 0.0) ? 1.0 : ((dr < 0.0) ? -1.0 : dr);
        dlog += log(fabs(dr));
        if (live) {
          double* cg = st.cache[s] + ((size_t)w * n + i) * 5 * nmo;
          for (int k = lane; k < 5 * nmo; k += 64) cg[k] = rnew[(k / nmo) * nmo_pad + (k % nmo)];
        }
        if (lane == 0) { xw[3 * e] = nx; xw[3 * e + 1] = ny; xw[3 * e + 2] = nz; }
      }
      __syncthreads();  // rnew / xw settled before the next electron's proposal reuses them
    }
    if (live) {
      double* Tg = st.T[s] + (size_t)w * n * n;
#pragma unroll
      for (int q = 0; q < 16; ++q)
        if (j < n && 16 * h + q < n) Tg[(size_t)j * n + 16 * h + q] = t[q];
      if (lane == 0) { st.dsign[s][w] = dsign; st.dlog[s][w] = dlog; }
    }
  }
  if (live) {
    double* xg = js.x + (size_t)w * N * 3;
    for (int k = lane; k < 3 * N; k += 64) xg[k] = xw[k];
    if (lane == 0) {
      mb.acc_w[w] += n_acc;
      if (DMC) { mb.r2_prop[w] += r2p; mb.r2_acc[w] += r2a; }
    }
  }
}
