// pqa_tile.hpp — the whole VMC / DMC electron sweep of a walker tile in ONE kernel, state on chip.
//
// The lane-per-walker sweep (pqa_lw.hpp) launches six kernels per electron and streams every walker's inverse through
// HBM once per move (k_commit_lw, 31 % of the step) and its coordinates and orbital rows twice (k_move_part_lw, 25 %).
// Walkers are independent Markov chains, so nothing forces a global synchronisation between electrons: here a block of
// PQA_TILE_NW waves owns as many walkers for the whole sweep,
//   * wave = walker: the transposed inverse T[j][k] of the current spin lives in REGISTERS (lane = electron row j and
//     column half h, 16 doubles per lane for n <= 32), the walker's coordinates in LDS;
//   * block = one (partly filled) 16-point MFMA tile: the orbital rows at the block's proposals are evaluated cooperatively like
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

#ifndef PQA_TILE_NW
#define PQA_TILE_NW 12      // walkers (waves) per block: 16 -> 128 VGPRs per lane (spills), 12 -> 168 (best measured), 8 -> 256, no spills (A/B via -DPQA_TILE_NW)
#endif
#define PQA_TILE_NT (64 * PQA_TILE_NW)
#define PQA_TILE_KT 64      // AO rows per pass of the orbital evaluation
#define PQA_TILE_MAXPASS 8

struct TileTab {
  int npass;
  int pass_chunk[PQA_TILE_MAXPASS + 1];  // chunk range of each pass (chunks of the ncomp = 5 table, rows <= KT per pass)
  int nmo_pad;                           // 16 * nt
  int rows_pad;                          // padded AO rows of the coefficient matrices (all chunks)
};

// dynamic LDS layout (doubles unless noted)
struct TileLds {
  double* xs;      // [NW][3 N]        coordinates
  double* tile;    // [5][KT][16]      AO tile of one pass
  double* rnew;    // [NW][5][nmo_pad] orbital rows at the proposals
  double* rold;    // [NW][5][nmo_pad] cached rows of the current positions: ALIASES the AO tile (free during the proposal)
  double* Cs;      // [rows_pad][nmo_pad] coefficient matrix of the current spin (B operand of the MFMA phase)
  double* wsc;     // [NW][16] per-wave scalars parked across the orbital phase: z (3), drift (3), U0, proposal (3), sign, log, r2 sums
  double* sh_xyz;  // [nshell][3]
  double* pr_exp;  // [nprim]
  double* pr_coef; // [nprim]
  int* sh_meta;    // [nshell][4]: l, nprim, first primitive, absolute padded AO row (chunk row0 + row in chunk)
  int* sh_list;    // [nshell] shells in chunk order (the order of ChunkTab::cw_shell[0])
  int* occ;        // [2][32]
};
__host__ __device__ inline size_t tile_lds_bytes(int N, int nmo_pad, int nshell, int nprim, int rows_pad) {
  size_t d = (size_t)PQA_TILE_NW * 3 * N + 5 * PQA_TILE_KT * 16 + (size_t)PQA_TILE_NW * 5 * nmo_pad + (size_t)rows_pad * nmo_pad +
             3 * (size_t)nshell + 2 * (size_t)nprim + (size_t)PQA_TILE_NW * 16;
  return d * sizeof(double) + ((size_t)5 * nshell + 64) * sizeof(int);
}

// dots of the value and gradient rows (slot order through occ) with the register inverse: d[c] = sum_k row_c[occ[k]] T[j][k]
// for the lane's own electron row j, c = value, d/dx, d/dy, d/dz (both halves of the wave end up with the full sums).
// One pass over the 16 columns serves the four rows: four separate passes let the scheduler hoist 4 x 32 LDS loads at once,
// which alone overflowed the 128-register budget.
__device__ __forceinline__ void tile_rowdots4(const double* __restrict__ rows, int nmo_pad, const int* __restrict__ occ, const double (&t)[16],
                                              int h, double (&d)[4]) {
  double s0 = 0.0, s1 = 0.0, s2 = 0.0, s3 = 0.0;
#pragma unroll
  for (int q = 0; q < 16; ++q) {
    const double* r = rows + occ[16 * h + q];
    const double tq = t[q];
    s0 += r[0] * tq; s1 += r[nmo_pad] * tq; s2 += r[2 * nmo_pad] * tq; s3 += r[3 * nmo_pad] * tq;
  }
  d[0] = s0 + __shfl_xor(s0, 32, 64); d[1] = s1 + __shfl_xor(s1, 32, 64);
  d[2] = s2 + __shfl_xor(s2, 32, 64); d[3] = s3 + __shfl_xor(s3, 32, 64);
}

// The sweep's random numbers, drawn ahead of it from the same Philox streams the other sweep kernels use (so the three
// paths make the same decisions): gauss [N][W][3] standard normals, unif [N][W].  Keeping Box-Muller (sincos, log) out of
// k_sweep_tile matters: its argument reduction alone costs that kernel dozens of spilled registers.
template <int PQA_UNIT = 0>  // (a template so that only the units that launch it compile it)
static __global__ __launch_bounds__(256) void k_tile_draws(uint64_t seed, uint32_t step, int N, long W, double* __restrict__ gauss, double* __restrict__ unif) {
  const long idx = (long)blockIdx.x * 256 + threadIdx.x;
  if (idx >= (long)N * W) return;
  const int e = (int)(idx / W);
  const long w = idx - (long)e * W;
  double z0, z1, z2, z3;
  normal2(philox(seed, (uint32_t)w, (uint32_t)e, PQA_STREAM_GAUSS_A, step), z0, z1);
  normal2(philox(seed, (uint32_t)w, (uint32_t)e, PQA_STREAM_GAUSS_B, step), z2, z3);
  gauss[3 * idx] = z0; gauss[3 * idx + 1] = z1; gauss[3 * idx + 2] = z2;
  const Philox p = philox(seed, (uint32_t)w, (uint32_t)e, PQA_STREAM_ACCEPT, step);
  unif[idx] = u01(p.c[0], p.c[1]);
}

template <bool DMC, int LMAX>
static __global__ __launch_bounds__(PQA_TILE_NT) void k_sweep_tile(SysDev S, SlaterState st, JastrowState js, MoveBuf mb, ChunkTab T, TileTab TT,
                                                     int has_jastrow, long W) {
  extern __shared__ double lds_raw[];
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  const int N = S.nelec, nmo_pad = TT.nmo_pad;
  TileLds L;
  L.xs = lds_raw;
  L.tile = L.xs + (size_t)PQA_TILE_NW * 3 * N;
  L.rnew = L.tile + 5 * PQA_TILE_KT * 16;
  L.rold = L.tile;
  L.Cs = L.rnew + (size_t)PQA_TILE_NW * 5 * nmo_pad;
  L.wsc = L.Cs + (size_t)TT.rows_pad * nmo_pad;
  L.sh_xyz = L.wsc + (size_t)PQA_TILE_NW * 16;
  L.pr_exp = L.sh_xyz + 3 * (size_t)S.nshell;
  L.pr_coef = L.pr_exp + S.nprim;
  L.sh_meta = (int*)(L.pr_coef + S.nprim);
  L.sh_list = L.sh_meta + 4 * (size_t)S.nshell;
  L.occ = L.sh_list + S.nshell;

  const long w_raw = (long)blockIdx.x * PQA_TILE_NW + wv;
  const bool live = w_raw < W;
  const long w = live ? w_raw : W - 1;  // tail waves shadow the last walker and write nothing
  // ---- stage tables, coordinates; clear the AO tile (its K-padding rows are never written again)
  for (int sh = tid; sh < S.nshell; sh += PQA_TILE_NT) {
    const int ia = S.shell_atom[sh];
    L.sh_xyz[3 * sh] = S.atom_xyz[3 * ia]; L.sh_xyz[3 * sh + 1] = S.atom_xyz[3 * ia + 1]; L.sh_xyz[3 * sh + 2] = S.atom_xyz[3 * ia + 2];
    L.sh_meta[4 * sh] = S.shell_l[sh];
    L.sh_meta[4 * sh + 1] = S.shell_prim_off[sh + 1] - S.shell_prim_off[sh];
    L.sh_meta[4 * sh + 2] = S.shell_prim_off[sh];
  }
  for (int pos = tid; pos < S.nshell; pos += PQA_TILE_NT) {  // cw_shell[0] lists every shell once, chunk by chunk
    const int sh = T.cw_shell[0][pos];
    int ch = 0;
    while (pos >= T.cw_off[0][4 * (ch + 1)]) ++ch;
    L.sh_list[pos] = sh;
    L.sh_meta[4 * sh + 3] = T.chunk_row0[ch] + T.shell_kb[sh];
  }
  for (int p = tid; p < S.nprim; p += PQA_TILE_NT) { L.pr_exp[p] = S.prim_exp[p]; L.pr_coef[p] = S.prim_coef[p]; }
  for (int k = tid; k < 64; k += PQA_TILE_NT) {
    const int s = k >> 5, q = k & 31, n = s ? S.ndn : S.nup;
    L.occ[k] = q < n ? S.det_occ[s][q] : 0;
  }
  for (int k = tid; k < 5 * PQA_TILE_KT * 16; k += PQA_TILE_NT) L.tile[k] = 0.0;
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
  // MFMA roles: (component c, orbital tile u) = (role % 5, role / 5), role = wv, wv + NW, ...  (10 roles for 32 orbitals)
  constexpr int NROLE = (10 + PQA_TILE_NW - 1) / PQA_TILE_NW;
  int n_acc = 0;
  double* ws = L.wsc + (size_t)wv * 16;
  if (lane == 0) { ws[12] = 0.0; ws[13] = 0.0; }  // r2 sums (DMC)

#pragma unroll
  for (int s = 0; s < 2; ++s) {  // unrolled: st.T[s], T.cpad[s], S.nmo[s] ... become static selections of kernel arguments
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
    if (lane == 0) { ws[10] = st.dsign[s][w]; ws[11] = st.dlog[s][w]; }
    const int ldc = T.ldc[s];
    __syncthreads();  // (the previous spin's MFMA reads of Cs are done)
    for (int k = tid; k < TT.rows_pad * ldc; k += PQA_TILE_NT) L.Cs[k] = T.cpad[s][k];
    __syncthreads();
    const double* __restrict__ C = L.Cs;
    const int nrole = 5 * (nmo_pad / 16);

    for (int i = 0; i < n; ++i) {
      const int e = s * S.nup + i;
      // ================= proposal: drift at the current position
      {
        const double* cg = st.cache[s] + ((size_t)w * n + i) * 5 * nmo;
        for (int k = lane; k < 5 * nmo; k += 64) rold[(k / nmo) * nmo_pad + (k % nmo)] = cg[k];
      }
      {  // scope: everything the later phases need is parked in ws (LDS), nothing stays in registers
      const double ex = xw[3 * e], ey = xw[3 * e + 1], ez = xw[3 * e + 2];
      double gx, gy, gz, U0 = 0.0;
      {
        double d4_[4];
        tile_rowdots4(rold, nmo_pad, occ, t, h, d4_);
        const double r0 = __shfl(d4_[0], i, 64), r1 = __shfl(d4_[1], i, 64), r2 = __shfl(d4_[2], i, 64), r3 = __shfl(d4_[3], i, 64);
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
      double z0, z1, z2;  // standard normals of this move: the replay tape, or what k_tile_draws generated from the Philox streams
      {
        const double* zt = mb.gauss + ((size_t)e * W + w) * 3;
        z0 = zt[0]; z1 = zt[1]; z2 = zt[2];
      }
      const double sq = sqrt(mb.tstep), df = DMC ? 1.0 : mb.tstep;
      z0 *= sq; z1 *= sq; z2 *= sq;
      if (lane == 0) {
        const double nx = ex + z0 + gx * df, ny = ey + z1 + gy * df, nz = ez + z2 + gz * df;
        rnew[0] = nx; rnew[1] = ny; rnew[2] = nz;  // the proposal travels through rnew's first slots
        ws[0] = z0; ws[1] = z1; ws[2] = z2; ws[3] = gx; ws[4] = gy; ws[5] = gz; ws[6] = U0; ws[7] = nx; ws[8] = ny; ws[9] = nz;
      }
      }
      __syncthreads();
      // ================= orbital rows at the 16 proposals of the block (k_orb's two phases on a 16-point tile)
      double px, py, pz;
      {
        const double* pr = L.rnew + (size_t)(tid % PQA_TILE_NW) * 5 * nmo_pad;  // thread's point in phase 1: tid mod NW
        px = pr[0]; py = pr[1]; pz = pr[2];
      }
      __syncthreads();  // every thread holds its point: rnew may now be overwritten
      d4 acc[NROLE];
#pragma unroll
      for (int k = 0; k < NROLE; ++k) acc[k] = (d4){0.0, 0.0, 0.0, 0.0};
      for (int ps = 0; ps < TT.npass; ++ps) {
        const int ch0 = TT.pass_chunk[ps], ch1 = TT.pass_chunk[ps + 1];
        const int row_base = T.chunk_row0[ch0];
        const int s_lo = T.cw_off[0][4 * ch0], s_hi = T.cw_off[0][4 * ch1];
#ifndef PQA_TILE_ABL_NOAO
        for (int it = tid; it < (s_hi - s_lo) * PQA_TILE_NW; it += PQA_TILE_NT) {  // NT is a multiple of NW: it mod NW == tid mod NW
          const int sh = L.sh_list[s_lo + it / PQA_TILE_NW];
          const int krow = L.sh_meta[4 * sh + 3] - row_base;
          const int l_ = L.sh_meta[4 * sh], np_ = L.sh_meta[4 * sh + 1], q0 = L.sh_meta[4 * sh + 2];
          const int pl = it % PQA_TILE_NW;
          shell_eval<5, LMAX>(l_, px - L.sh_xyz[3 * sh], py - L.sh_xyz[3 * sh + 1], pz - L.sh_xyz[3 * sh + 2], L.pr_exp + q0, L.pr_coef + q0, np_,
                        [&](int m, double v, double ax, double ay, double az, double lp) {
                          double* tl = L.tile + (size_t)(krow + m) * 16 + pl;
                          tl[0] = v; tl[PQA_TILE_KT * 16] = ax; tl[2 * PQA_TILE_KT * 16] = ay; tl[3 * PQA_TILE_KT * 16] = az;
                          tl[4 * PQA_TILE_KT * 16] = lp;
                        });
        }
#endif
        __syncthreads();
#ifndef PQA_TILE_ABL_NOMFMA
        {
          const int nrow = T.chunk_row0[ch1 - 1] + ((T.chunk_nk[ch1 - 1] + 3) & ~3) - row_base;  // padded rows of this pass
#pragma unroll
          for (int k = 0; k < NROLE; ++k) {
            const int role = wv + k * PQA_TILE_NW;
            if (role < nrole) {
              const double* a_ = L.tile + (size_t)(role % 5) * PQA_TILE_KT * 16 + (size_t)kq * 16 + i16;
              const double* b_ = C + (size_t)(row_base + kq) * ldc + 16 * (role / 5) + i16;
#pragma unroll 4
              for (int ks = 0; ks < nrow / 4; ++ks)
                acc[k] = __builtin_amdgcn_mfma_f64_16x16x4f64(a_[(size_t)ks * 64], b_[(size_t)ks * 4 * ldc], acc[k], 0, 0, 0);
            }
          }
        }
#endif
        __syncthreads();
      }
#pragma unroll
      for (int k = 0; k < NROLE; ++k) {  // lane holds D[point = kq + 4r][orbital = 16 u + i16] of its role
        const int role = wv + k * PQA_TILE_NW;
        if (role < nrole) {
#pragma unroll
          for (int r = 0; r < 4; ++r)
            if (kq + 4 * r < PQA_TILE_NW) L.rnew[((size_t)(kq + 4 * r) * 5 + role % 5) * nmo_pad + 16 * (role / 5) + i16] = acc[k][r];
        }
      }
      __syncthreads();
      // ================= Metropolis at the proposal
      double dn[4];
      tile_rowdots4(rnew, nmo_pad, occ, t, h, dn);
      const double tmp0 = dn[0];  // sum_k V[k] T[j][k] for the lane's row: reused by the update
      double val2, sgn = 1.0, hx, hy, hz;
      const double dr = __shfl(tmp0, i, 64);
      {
        const double r1 = __shfl(dn[1], i, 64), r2 = __shfl(dn[2], i, 64), r3 = __shfl(dn[3], i, 64);
        hx = finite_or(r1 / dr, 0.0); hy = finite_or(r2 / dr, 0.0); hz = finite_or(r3 / dr, 0.0);
        const double v = finite_or(dr, 1.0);
        val2 = v * v;
        sgn = (v > 0.0) ? 1.0 : ((v < 0.0) ? -1.0 : 0.0);
      }
      const double nx = ws[7], ny = ws[8], nz = ws[9];
#ifndef PQA_TILE_ABL_NOJAS
      if (has_jastrow) {
        double g[3], lp, U;
        jas_eval<1, false>(S, xw, e, nx, ny, nz, U, g, lp, 1);
        hx += g[0]; hy += g[1]; hz += g[2];
        const double ej = exp(U - ws[6]);
        val2 *= ej * ej;
      }
#endif
      const double z0 = ws[0], z1 = ws[1], z2 = ws[2], gx = ws[3], gy = ws[4], gz = ws[5];
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
      const double u = mb.unif[(size_t)e * W + w];
      const bool accd = ratio > u;  // wave-uniform: every lane computed the same numbers
      if (DMC && lane == 0) {
        const double r2 = (z0 + gx) * (z0 + gx) + (z1 + gy) * (z1 + gy) + (z2 + gz) * (z2 + gz);
        ws[12] += r2;
        if (accd) ws[13] += r2;
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
        if (lane == 0) {
          ws[10] *= (dr > 0.0) ? 1.0 : ((dr < 0.0) ? -1.0 : dr);
          ws[11] += log(fabs(dr));
          xw[3 * e] = nx; xw[3 * e + 1] = ny; xw[3 * e + 2] = nz;
        }
        if (live) {
          double* cg = st.cache[s] + ((size_t)w * n + i) * 5 * nmo;
          for (int k = lane; k < 5 * nmo; k += 64) cg[k] = rnew[(k / nmo) * nmo_pad + (k % nmo)];
        }
      }
      __syncthreads();  // rnew / xw settled before the next electron's proposal reuses them
    }
    if (live) {
      double* Tg = st.T[s] + (size_t)w * n * n;
#pragma unroll
      for (int q = 0; q < 16; ++q)
        if (j < n && 16 * h + q < n) Tg[(size_t)j * n + 16 * h + q] = t[q];
      if (lane == 0) { st.dsign[s][w] = ws[10]; st.dlog[s][w] = ws[11]; }
    }
  }
  if (live) {
    double* xg = js.x + (size_t)w * N * 3;
    for (int k = lane; k < 3 * N; k += 64) xg[k] = xw[k];
    if (lane == 0) {
      mb.acc_w[w] += n_acc;
      if (DMC) { mb.r2_prop[w] += ws[12]; mb.r2_acc[w] += ws[13]; }
    }
  }
}
