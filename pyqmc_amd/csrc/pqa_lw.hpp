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
// State seen by these kernels:
//   xt   [N][3][W]            coordinates, SoA (walker fastest): every Jastrow load is a coalesced wave access
//   T    [s] [W][n][n]        inverse, canonical AoS: a partial-sum thread only gathers its slice of ONE 256-B row
//   cache[s] [W][n][5][nmo]   cached MO value/grad/lap rows, canonical AoS (same argument)
// Keeping T and the cache in AoS lets the Sherman-Morrison commit run as a register-resident wave-per-walker
// kernel that touches accepted walkers only (an SoA commit drags every walker's inverse through HBM because
// accepted and rejected walkers share cache lines: 307 MB vs 184 MB per launch at W = 16384, PMC-measured).
#pragma once
#include "pqa_common.hpp"
#include "pqa_jastrow.hpp"
#include "pqa_vmc.hpp"

struct LwState {
  double* xt;
  double* T[2];      // AoS [W][n][n]   (single determinant)
  double* cache[2];  // AoS [W][n][5][nmo]
  double* dsign[2];  // [W] (single determinant) — shared with SlaterState
  double* dlog[2];
  double* auxt;      // [8][W]: scaled gaussian (3), limited drift (3), U_old, ratio
};

// ---------------------------------------------------------------- layout transposes
// in [R][C] -> out [C][R] through a 32x33 LDS tile; block (32,8)
__global__ __launch_bounds__(256) void k_transpose(const double* __restrict__ in, double* __restrict__ out, long R, long C) {
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

// ---------------------------------------------------------------- per-lane Jastrow
// Contribution of the pairs j = j0, j0+dj, ... and ions I = j0, j0+dj, ... to U_e, grad U_e, (bare) lap U_e
// of electron e of walker w at (rx,ry,rz); MODE 2 also returns that share of the Coulomb sums
// ee = sum_{j>e} 1/r, ei = -sum Z/r.  (j0,dj) = (0,1) gives the full sums.
template <int MODE>
__device__ __forceinline__ void jas_eval_lane(const SysDev& S, const double* __restrict__ xt, long W, long w, int e,
                                              double rx, double ry, double rz, int has_jastrow, int j0, int dj, double& U,
                                              double (&g)[3], double& lapU, double& ee, double& ei) {
  const int edown = e >= S.nup;
  const double irb = 1.0 / S.rcut_b, ira = 1.0 / S.rcut_a;
  double u = 0.0, gx = 0.0, gy = 0.0, gz = 0.0, lp = 0.0, see = 0.0, sei = 0.0;
#pragma unroll 4
  for (int j = j0; j < S.nelec; j += dj) {
    const double* xj = xt + (size_t)j * 3 * W + w;
    const double dx = rx - xj[0], dy = ry - xj[W], dz = rz - xj[2 * W];
    const double r = sqrt(dx * dx + dy * dy + dz * dz);
    if (j == e) continue;
    if (MODE == 2 && j > e) see += fast_rcp(r);
    if (has_jastrow && r < S.rcut_b) {
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
  for (int I = j0; I < S.natom; I += dj) {
    const double dx = rx - S.atom_xyz[3 * I], dy = ry - S.atom_xyz[3 * I + 1], dz = rz - S.atom_xyz[3 * I + 2];
    const double r = sqrt(dx * dx + dy * dy + dz * dz);
    if (MODE == 2) sei -= S.atom_charge[I] * fast_rcp(r);
    if (has_jastrow && r < S.rcut_a) {
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
  U = u; g[0] = gx; g[1] = gy; g[2] = gz; lapU = lp; ee = see; ei = sei;
}

// ---------------------------------------------------------------- move kernels
// Every per-walker loop (orbital slots for the Slater ratios, other electrons and ions for the Jastrow) is
// split over G thread groups (grid.y) so that even W/64 < #SIMDs fills the chip and no thread walks a long
// chain of dependent loads; partial sums land in part[G][8][W] and a finish kernel (thread = walker) adds
// them in group order (deterministic) and does the per-walker scalar work.
//   part rows: 0..3 Slater sums (value, d/dx, d/dy, d/dz), 4 U, 5..7 grad U

// pos: proposal [W][3] (accept) or NULL = current position of e from xt (propose)
// rows: [W][5][nmo] orbital rows at `pos` (accept) or NULL = cached rows ct (propose)
__global__ __launch_bounds__(64) void k_move_part_lw(SysDev S, LwState L, int e, int has_jastrow, const double* __restrict__ pos,
                                                     const double* __restrict__ rows, long W, int G, double* __restrict__ part) {
  const long w = (long)blockIdx.x * 64 + threadIdx.x;
  const int g = blockIdx.y;
  if (w >= W) return;
  const int s = e >= S.nup, i = e - s * S.nup, n = s ? S.ndn : S.nup, nmo = S.nmo[s];
  double px, py, pz;
  if (pos) { px = pos[3 * w]; py = pos[3 * w + 1]; pz = pos[3 * w + 2]; }
  else { const double* xe = L.xt + (size_t)e * 3 * W + w; px = xe[0]; py = xe[W]; pz = xe[2 * W]; }
  double r0 = 0.0, r1 = 0.0, r2 = 0.0, r3 = 0.0;
  {
    const double* Ti = L.T[s] + ((size_t)w * n + i) * n;
    const int* occ = S.det_occ[s];
    const double* row = rows ? rows + (size_t)w * 5 * nmo : L.cache[s] + ((size_t)w * n + i) * 5 * nmo;
#pragma unroll 4
    for (int j = g; j < n; j += G) {
      const double t = Ti[j];
      const int o = occ[j];
      r0 += row[o] * t; r1 += row[nmo + o] * t; r2 += row[2 * nmo + o] * t; r3 += row[3 * nmo + o] * t;
    }
  }
  double U, gg[3], lp, ee, ei;
  jas_eval_lane<1>(S, L.xt, W, w, e, px, py, pz, has_jastrow, g, G, U, gg, lp, ee, ei);
  double* p = part + (size_t)g * 8 * W + w;
  p[0] = r0; p[W] = r1; p[2 * W] = r2; p[3 * W] = r3; p[4 * W] = U; p[5 * W] = gg[0]; p[6 * W] = gg[1]; p[7 * W] = gg[2];
}

__device__ __forceinline__ void lw_sum_parts(const double* __restrict__ part, long W, long w, int G, double (&v)[8]) {
#pragma unroll
  for (int c = 0; c < 8; ++c) v[c] = 0.0;
  for (int g = 0; g < G; ++g) {
    const double* p = part + (size_t)g * 8 * W + w;
#pragma unroll
    for (int c = 0; c < 8; ++c) v[c] += p[(size_t)c * W];
  }
}

// drift at the current position, proposal r' = r + sqrt(tau) z + tau limdrift(grad)   (mc.py:117-121)
__global__ __launch_bounds__(64) void k_propose_fin_lw(SysDev S, LwState L, MoveBuf mb, int e, long W, int G,
                                                       const double* __restrict__ part) {
  const long w = (long)blockIdx.x * 64 + threadIdx.x;
  if (w >= W) return;
  double v[8];
  lw_sum_parts(part, W, w, G, v);
  double gx = finite_or(v[1] / v[0], 0.0) + v[5], gy = finite_or(v[2] / v[0], 0.0) + v[6], gz = finite_or(v[3] / v[0], 0.0) + v[7];
  limdrift3(gx, gy, gz);
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
  np_[0] = xe[0] + z0 + gx * mb.tstep;
  np_[1] = xe[W] + z1 + gy * mb.tstep;
  np_[2] = xe[2 * W] + z2 + gz * mb.tstep;
  double* a = L.auxt + w;
  a[0] = z0; a[W] = z1; a[2 * W] = z2; a[3 * W] = gx; a[4 * W] = gy; a[5 * W] = gz; a[6 * W] = v[4];
}

// Metropolis decision (mc.py:124-132); accepted walkers: move the coordinate, update sign/log of the
// determinant and leave the determinant ratio in auxt[7] for the commit kernel.
__global__ __launch_bounds__(64) void k_accept_fin_lw(SysDev S, LwState L, MoveBuf mb, int e, int has_jastrow, long W, int G,
                                                      const double* __restrict__ part) {
  const long w = (long)blockIdx.x * 64 + threadIdx.x;
  if (w >= W) return;
  const int s = e >= S.nup;
  double v[8];
  lw_sum_parts(part, W, w, G, v);
  double gx = finite_or(v[1] / v[0], 0.0) + v[5], gy = finite_or(v[2] / v[0], 0.0) + v[6], gz = finite_or(v[3] / v[0], 0.0) + v[7];
  const double* a = L.auxt + w;
  double val = finite_or(v[0], 1.0);
  if (has_jastrow) val *= exp(v[4] - a[6 * W]);
  limdrift3(gx, gy, gz);
  const double a0 = a[0], a1 = a[W], a2 = a[2 * W];
  const double fwd = a0 * a0 + a1 * a1 + a2 * a2;
  const double bx = a0 + mb.tstep * (a[3 * W] + gx), by = a1 + mb.tstep * (a[4 * W] + gy), bz = a2 + mb.tstep * (a[5 * W] + gz);
  const double bwd = bx * bx + by * by + bz * bz;
  const double t_prob = exp(1.0 / (2.0 * mb.tstep) * (fwd - bwd));
  const double ratio = val * val * t_prob;
  double u;
  if (mb.unif) u = mb.unif[(size_t)e * W + w];
  else {
    const Philox p = philox(mb.seed, (uint32_t)w, (uint32_t)e, PQA_STREAM_ACCEPT, mb.step);
    u = u01(p.c[0], p.c[1]);
  }
  const bool acc = ratio > u;
  mb.accept[w] = acc;
  if (mb.accept_rec) mb.accept_rec[(size_t)e * W + w] = acc;
  if (!acc) return;
  mb.acc_w[w] += 1;
  double* xe = L.xt + (size_t)e * 3 * W + w;
  xe[0] = mb.newpos[3 * w]; xe[W] = mb.newpos[3 * w + 1]; xe[2 * W] = mb.newpos[3 * w + 2];
  const double dr = v[0];  // determinant ratio
  L.dsign[s][w] *= (dr > 0.0) ? 1.0 : ((dr < 0.0) ? -1.0 : dr);
  L.dlog[s][w] += log(fabs(dr));
  L.auxt[7 * W + w] = dr;
}

// ---------------------------------------------------------------- commit (Sherman-Morrison, slater.py:88-94)
// n = 32: the 8-KB inverse of one accepted walker lives in the registers of ONE wave — lane l holds the
// contiguous half-row T[l/2][16(l&1) .. +15] (8 x global_load_dwordx4, consecutive lanes = consecutive 128 B:
// perfectly coalesced, no LDS):
//   tmp[r] = V . T[r]         (16 FMAs per lane + one exchange with the partner lane)
//   ratio  = tmp[i],  R = T[i]/ratio  (read straight from global: row i is only rewritten by its own lanes)
//   T[r]  -= R tmp[r]  (r != i),  T[i] = R
// and the 5*nmo cached orbital values of electron i are refreshed.  Rejected walkers exit at once, so HBM traffic
// is 16 KB per ACCEPTED walker.  (A blocked/delayed variant that touches T once per 8 moves was measured 9 %
// slower in the SoA form — its flush re-reads the block's V/R vectors from cache for every row — and was dropped.)
__global__ __launch_bounds__(64) void k_commit_ww32(SysDev S, LwState L, MoveBuf mb, int e, const double* __restrict__ motmp,
                                                    long W) {
  const long w = blockIdx.x;
  if (!mb.accept[w]) return;
  const int lane = threadIdx.x;
  const int s = e >= S.nup, i = e - s * S.nup, nmo = S.nmo[s];
  const int r = lane >> 1, h = lane & 1;
  double* Tw = L.T[s] + (size_t)w * 1024;
  const double* row = motmp + (size_t)w * 5 * nmo;
  const int* occ = S.det_occ[s];
  double t[16], V[16], R[16];
  const double4_* src = reinterpret_cast<const double4_*>(Tw + (size_t)lane * 16);
  const double4_* ri = reinterpret_cast<const double4_*>(Tw + (size_t)i * 32 + 16 * h);
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const double4_ a = src[q], b = ri[q];
    t[4 * q] = a.x; t[4 * q + 1] = a.y; t[4 * q + 2] = a.z; t[4 * q + 3] = a.w;
    R[4 * q] = b.x; R[4 * q + 1] = b.y; R[4 * q + 2] = b.z; R[4 * q + 3] = b.w;
  }
#pragma unroll
  for (int k = 0; k < 16; ++k) V[k] = row[occ[16 * h + k]];
  double tmp = 0.0;
#pragma unroll
  for (int k = 0; k < 16; ++k) tmp += V[k] * t[k];
  tmp += __shfl_xor(tmp, 1, 64);  // partner half-row
  const double inv = 1.0 / L.auxt[7 * W + w];  // determinant ratio = tmp of row i (computed identically upstream)
  __syncthreads();                 // every lane has read row i before its owners overwrite it
#pragma unroll
  for (int k = 0; k < 16; ++k) {
    const double Rk = R[k] * inv;
    t[k] = (r == i) ? Rk : t[k] - Rk * tmp;
  }
  double4_* dst = reinterpret_cast<double4_*>(Tw + (size_t)lane * 16);
#pragma unroll
  for (int q = 0; q < 4; ++q) dst[q] = double4_{t[4 * q], t[4 * q + 1], t[4 * q + 2], t[4 * q + 3]};
  double* c = L.cache[s] + ((size_t)w * 32 + i) * 5 * nmo;
  for (int k = lane; k < 5 * nmo; k += 64) c[k] = row[k];
}

// Slater part of the kinetic energy, wave per walker: r_c(i) = sum_j cache[i][c][occ_j] T[i][j] for all electrons of
// spin s at once (lane = (electron, half-row), 128-B loads).  out [W][N][5].  n = 32, single determinant, occ = identity
// is NOT assumed (gathers through occ).
__global__ __launch_bounds__(64) void k_slater_rows_ww32(SysDev S, LwState L, int s, long W, double* __restrict__ out) {
  const long w = blockIdx.x;
  const int lane = threadIdx.x, r = lane >> 1, h = lane & 1, nmo = S.nmo[s];
  const double* Tw = L.T[s] + (size_t)w * 1024 + (size_t)lane * 16;
  const double* cw = L.cache[s] + ((size_t)w * 32 + r) * 5 * nmo;
  const int* occ = S.det_occ[s];
  double t[16], acc[5] = {0.0, 0.0, 0.0, 0.0, 0.0};
#pragma unroll
  for (int k = 0; k < 16; ++k) t[k] = Tw[k];
#pragma unroll
  for (int k = 0; k < 16; ++k) {
    const int o = occ[16 * h + k];
#pragma unroll
    for (int c = 0; c < 5; ++c) acc[c] += cw[c * nmo + o] * t[k];
  }
#pragma unroll
  for (int c = 0; c < 5; ++c) acc[c] += __shfl_xor(acc[c], 1, 64);
  if (h == 0) {
    double* o = out + ((size_t)w * S.nelec + (s ? S.nup : 0) + r) * 5;
#pragma unroll
    for (int c = 0; c < 5; ++c) o[c] = acc[c];
  }
}

// ---------------------------------------------------------------- kinetic + Coulomb
// thread = (walker, electron), walker fastest.  srow [W][N][5]: Slater row sums from k_slater_rows_ww32 (or NULL:
// computed here).  part [4][N][W]: ke_e, grad2_e, ee_e, ei_e
__global__ __launch_bounds__(64) void k_kinetic_lw(SysDev S, LwState L, int has_jastrow, long W, const double* __restrict__ srow,
                                                   double* __restrict__ part) {
  const long w = (long)blockIdx.x * 64 + threadIdx.x;
  const int e = blockIdx.y;
  if (w >= W) return;
  const int s = e >= S.nup, i = e - s * S.nup, n = s ? S.ndn : S.nup, nmo = S.nmo[s];
  double r[5] = {0.0, 0.0, 0.0, 0.0, 0.0};
  if (srow) {
    const double* q = srow + ((size_t)w * S.nelec + e) * 5;
#pragma unroll
    for (int c = 0; c < 5; ++c) r[c] = q[c];
  } else {
    const double* Ti = L.T[s] + ((size_t)w * n + i) * n;
    const double* ci = L.cache[s] + ((size_t)w * n + i) * 5 * nmo;
    const int* occ = S.det_occ[s];
    for (int j = 0; j < n; ++j) {
      const double t = Ti[j];
      const int o = occ[j];
#pragma unroll
      for (int c = 0; c < 5; ++c) r[c] += ci[c * nmo + o] * t;
    }
  }
  const double gs0 = r[1] / r[0], gs1 = r[2] / r[0], gs2 = r[3] / r[0], ls = r[4] / r[0];
  const double* xe = L.xt + (size_t)e * 3 * W + w;
  double U, gj[3], lj, ee, ei;
  jas_eval_lane<2>(S, L.xt, W, w, e, xe[0], xe[W], xe[2 * W], has_jastrow, 0, 1, U, gj, lj, ee, ei);
  lj += gj[0] * gj[0] + gj[1] * gj[1] + gj[2] * gj[2];
  const double gx = gs0 + gj[0], gy = gs1 + gj[1], gz = gs2 + gj[2];
  const double lap = ls + lj + 2.0 * (gs0 * gj[0] + gs1 * gj[1] + gs2 * gj[2]);
  const size_t o = (size_t)e * W + w, NW = (size_t)S.nelec * W;
  part[o] = -0.5 * lap;
  part[NW + o] = gx * gx + gy * gy + gz * gz;
  part[2 * NW + o] = ee;
  part[3 * NW + o] = ei;
}

// out rows ke, ee, ei, grad2 (layout of k_kinetic_coulomb) = sums over electrons of part
__global__ void k_kinetic_reduce(const double* __restrict__ part, int N, long W, double* __restrict__ out) {
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
