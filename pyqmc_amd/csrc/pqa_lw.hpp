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
//   ct   [s] [n][5][nmo][W]   cached MO value/grad/lap rows of every electron
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
  double* ct[2];
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
#ifndef PQA_JAS_PF
#define PQA_JAS_PF 4
#endif
#define PQA_JAS_NF 4  // basis functions per kind whose tables the fast path keeps in scalar registers
// FAST (nb, na <= PQA_JAS_NF, the reference's default Jastrow has 4 + 4): the function tables (kind, parameter, cusp constant,
// the two coefficient columns electron e can meet) are read ONCE into scalar registers.  Indexed by the loop variable they
// were scalar loads inside the innermost loop — two dependent load-and-wait pairs per function and pair, ~90 per thread —
// and with four waves per SIMD those waits, not the arithmetic, set the kernel's time.  Same operations in the same order.
template <int MODE, bool PBC, bool FAST>
__device__ __forceinline__ void jas_eval_lane_t(const SysDev& S, const double* __restrict__ xt, long W, long w, int e,
                                                double rx, double ry, double rz, int has_jastrow, int j0, int dj, double& U,
                                                double (&g)[3], double& lapU, double& ee, double& ei) {
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
  // The partners' coordinates are fetched PQA_JAS_PF at a time before any of them is used.
  for (int jb = j0; jb < S.nelec; jb += PQA_JAS_PF * dj) {
    double cx[PQA_JAS_PF], cy[PQA_JAS_PF], cz[PQA_JAS_PF];
#pragma unroll
    for (int u = 0; u < PQA_JAS_PF; ++u) {
      const int j = jb + u * dj;
      const double* xj = xt + (size_t)(j < S.nelec ? j : e) * 3 * W + w;  // past the end: any valid address, never used
      cx[u] = xj[0]; cy[u] = xj[W]; cz[u] = xj[2 * W];
    }
#pragma unroll
    for (int u = 0; u < PQA_JAS_PF; ++u) {
      const int j = jb + u * dj;
      if (j >= S.nelec || j == e) continue;
      double dx = rx - cx[u], dy = ry - cy[u], dz = rz - cz[u];
      if (PBC) min_image(S, dx, dy, dz);  // compiled out of the open-boundary instantiation (the hot path of the headline bench)
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
  for (int I = j0; I < S.natom; I += dj) {
    double dx = rx - S.atom_xyz[3 * I], dy = ry - S.atom_xyz[3 * I + 1], dz = rz - S.atom_xyz[3 * I + 2];
    double ac[NF];
    if (FAST) {
#pragma unroll
      for (int k = 0; k < NF; ++k) ac[k] = (has_jastrow && k < S.na) ? S.acoeff[(I * S.na + k) * 2 + edown] : 0.0;
    }
    if (PBC) min_image(S, dx, dy, dz);
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
template <int MODE, bool PBC>
__device__ __forceinline__ void jas_eval_lane(const SysDev& S, const double* __restrict__ xt, long W, long w, int e,
                                              double rx, double ry, double rz, int has_jastrow, int j0, int dj, double& U,
                                              double (&g)[3], double& lapU, double& ee, double& ei) {
  if (S.nb <= PQA_JAS_NF && S.na <= PQA_JAS_NF) jas_eval_lane_t<MODE, PBC, true>(S, xt, W, w, e, rx, ry, rz, has_jastrow, j0, dj, U, g, lapU, ee, ei);
  else jas_eval_lane_t<MODE, PBC, false>(S, xt, W, w, e, rx, ry, rz, has_jastrow, j0, dj, U, g, lapU, ee, ei);
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
template <bool PBC, bool CX = false>
__global__ __launch_bounds__(64) void k_move_part_lw(SysDev S, LwState L, int e, int has_jastrow, const double* __restrict__ pos,
                                                     const double* __restrict__ rows, long W, int G, double* __restrict__ part) {
  const long w = (long)blockIdx.x * 64 + threadIdx.x;
  const int g = blockIdx.y;
  if (w >= W) return;
  const int s = e >= S.nup, i = e - s * S.nup, n = s ? S.ndn : S.nup, nmo = S.nmo[s];
  constexpr int CF = CX ? 2 : 1;
  double px, py, pz;
  if (pos) { px = pos[3 * w]; py = pos[3 * w + 1]; pz = pos[3 * w + 2]; }
  else { const double* xe = L.xt + (size_t)e * 3 * W + w; px = xe[0]; py = xe[W]; pz = xe[2 * W]; }
  double r0 = 0.0, r1 = 0.0, r2 = 0.0, r3 = 0.0, q0 = 0.0, q1 = 0.0, q2 = 0.0, q3 = 0.0;  // q: imaginary parts (CX)
#ifndef PQA_MP_NOSLATER
  {
    const double* Ti = L.Tt[s] + (size_t)i * n * CF * W + w;
    const int* occ = S.det_occ[s];
    const int nh = nmo / CF;  // orbitals
    // group g takes a CONTIGUOUS range of orbital slots: the proposal's rows are point-major (k_orb's output, 1280 B per
    // walker), so a thread's consecutive slots share 64-byte lines (4 lines per thread and component instead of 8
    // lines used 8 bytes each when the slots were dealt round-robin: 53 -> 22 us of this kernel)
    const int nj = (n + G - 1) / G, jb = g * nj, je = (jb + nj < n) ? jb + nj : n;
    if (rows && !CX && S.occ_ident[s] && ((jb | (je - jb) | nmo) & 3) == 0) {
      // ground-state occupation: the slots are the orbitals themselves, so the thread's slice of each component row is read 32
      // bytes at a time (64 lanes on 64 different rows: a quarter of the load instructions / cache-line visits); same
      // operations in the same order
      const double* row = rows + (size_t)w * 5 * nmo;
      for (int j = jb; j < je; j += 4) {
        const double4 a0 = *reinterpret_cast<const double4*>(row + j), a1 = *reinterpret_cast<const double4*>(row + nmo + j);
        const double4 a2 = *reinterpret_cast<const double4*>(row + 2 * nmo + j), a3 = *reinterpret_cast<const double4*>(row + 3 * nmo + j);
        const double t0 = Ti[(size_t)j * W], t1 = Ti[(size_t)(j + 1) * W], t2 = Ti[(size_t)(j + 2) * W], t3 = Ti[(size_t)(j + 3) * W];
        r0 += a0.x * t0; r1 += a1.x * t0; r2 += a2.x * t0; r3 += a3.x * t0;
        r0 += a0.y * t1; r1 += a1.y * t1; r2 += a2.y * t1; r3 += a3.y * t1;
        r0 += a0.z * t2; r1 += a1.z * t2; r2 += a2.z * t2; r3 += a3.z * t2;
        r0 += a0.w * t3; r1 += a1.w * t3; r2 += a2.w * t3; r3 += a3.w * t3;
      }
    } else if (rows) {
      const double* row = rows + (size_t)w * 5 * nmo;
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
    } else {
      const double* ci = L.ct[s] + (size_t)i * 5 * nmo * W + w;
#pragma unroll 4
      for (int j = jb; j < je; ++j) {
        const double* cj = ci + (size_t)occ[j] * W;
        if (CX) {
          const double tr = Ti[(size_t)(2 * j) * W], ti = Ti[(size_t)(2 * j + 1) * W];
          const double* dj = cj + (size_t)nh * W;  // imaginary parts
          const double a0 = cj[0], b0 = dj[0], a1 = cj[(size_t)nmo * W], b1 = dj[(size_t)nmo * W];
          const double a2 = cj[(size_t)2 * nmo * W], b2 = dj[(size_t)2 * nmo * W], a3 = cj[(size_t)3 * nmo * W], b3 = dj[(size_t)3 * nmo * W];
          r0 += a0 * tr - b0 * ti; q0 += a0 * ti + b0 * tr; r1 += a1 * tr - b1 * ti; q1 += a1 * ti + b1 * tr;
          r2 += a2 * tr - b2 * ti; q2 += a2 * ti + b2 * tr; r3 += a3 * tr - b3 * ti; q3 += a3 * ti + b3 * tr;
        } else {
          const double t = Ti[(size_t)j * W];
          r0 += cj[0] * t; r1 += cj[(size_t)nmo * W] * t; r2 += cj[(size_t)2 * nmo * W] * t; r3 += cj[(size_t)3 * nmo * W] * t;
        }
      }
    }
  }
#endif
  double U = 0.0, gg[3] = {0.0, 0.0, 0.0}, lp, ee, ei;
#ifndef PQA_MP_NOJAS
  jas_eval_lane<1, PBC>(S, L.xt, W, w, e, px, py, pz, has_jastrow, g, G, U, gg, lp, ee, ei);
#endif
  constexpr int PR = PQA_LW_PART_ROWS(CX);
  double* p = part + (size_t)g * PR * W + w;
  if (CX) {  // rows: Re r0, Im r0, Re r1, Im r1, ..., then U, grad U
    p[0] = r0; p[W] = q0; p[2 * W] = r1; p[3 * W] = q1; p[4 * W] = r2; p[5 * W] = q2; p[6 * W] = r3; p[7 * W] = q3;
    p[8 * W] = U; p[9 * W] = gg[0]; p[10 * W] = gg[1]; p[11 * W] = gg[2];
  } else {
    p[0] = r0; p[W] = r1; p[2 * W] = r2; p[3 * W] = r3; p[4 * W] = U; p[5 * W] = gg[0]; p[6 * W] = gg[1]; p[7 * W] = gg[2];
  }
}

template <int PR>
__device__ __forceinline__ void lw_sum_parts(const double* __restrict__ part, long W, long w, int G, double (&v)[PR]) {
#pragma unroll
  for (int c = 0; c < PR; ++c) v[c] = 0.0;
  for (int g = 0; g < G; ++g) {
    const double* p = part + (size_t)g * PR * W + w;
#pragma unroll
    for (int c = 0; c < PR; ++c) v[c] += p[(size_t)c * W];
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

// drift at the current position, proposal r' = r + sqrt(tau) z + tau limdrift(grad)   (mc.py:117-121)
template <bool CX = false>
__global__ __launch_bounds__(64) void k_propose_fin_lw(SysDev S, LwState L, MoveBuf mb, int e, long W, int G,
                                                       const double* __restrict__ part) {
  const long w = (long)blockIdx.x * 64 + threadIdx.x;
  if (w >= W) return;
  constexpr int PR = PQA_LW_PART_ROWS(CX), JU = CX ? 8 : 4;
  double v[PR];
  lw_sum_parts<PR>(part, W, w, G, v);
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
  double* a = L.auxt + w;
  a[0] = z0; a[W] = z1; a[2 * W] = z2; a[3 * W] = gx; a[4 * W] = gy; a[5 * W] = gz; a[6 * W] = v[JU];
}

// Metropolis decision (mc.py:124-132); accepted walkers: move the coordinate, update sign/log of the
// determinant, and stage R[k] = T[i][k]/ratio in Rbuf[n][W] (complex: [n][2][W]) for the commit kernel.
template <bool CX = false>
__global__ __launch_bounds__(64) void k_accept_fin_lw(SysDev S, LwState L, MoveBuf mb, int e, int has_jastrow, long W, int G,
                                                      const double* __restrict__ part, double* __restrict__ Rbuf,
                                                      double* __restrict__ Vbuf, uint8_t* __restrict__ act,
                                                      const double* __restrict__ motmp) {
  const long w = (long)blockIdx.x * 64 + threadIdx.x;
  if (w >= W) return;
  const int s = e >= S.nup, i = e - s * S.nup, n = s ? S.ndn : S.nup, nmo = S.nmo[s];
  constexpr int PR = PQA_LW_PART_ROWS(CX), JU = CX ? 8 : 4, CF = CX ? 2 : 1;
  double v[PR];
  lw_sum_parts<PR>(part, W, w, G, v);
  double gx, gy, gz, dr, di;
  lw_slater_terms<CX, PR>(v, gx, gy, gz, dr, di);
  gx += v[JU + 1]; gy += v[JU + 2]; gz += v[JU + 3];
  const double* a = L.auxt + w;
  double val2;  // |Psi(new)/Psi|^2 (mc.py:131)
  if (CX) val2 = finite_or(dr * dr + di * di, 1.0);
  else { const double val = finite_or(dr, 1.0); val2 = val * val; }
  if (has_jastrow) { const double ej = exp(v[JU] - a[6 * W]); val2 *= ej * ej; }
  const double a0 = a[0], a1 = a[W], a2 = a[2 * W];
  const double fwd = a0 * a0 + a1 * a1 + a2 * a2;
  double bx, by, bz;
  if (mb.dmc) {  // dmc.py:57-60: backward = gauss + drift(old) + drift(new)
    limdrift_dmc(gx, gy, gz, mb.tstep);
    bx = a0 + a[3 * W] + gx; by = a1 + a[4 * W] + gy; bz = a2 + a[5 * W] + gz;
  } else {
    limdrift3(gx, gy, gz);
    bx = a0 + mb.tstep * (a[3 * W] + gx); by = a1 + mb.tstep * (a[4 * W] + gy); bz = a2 + mb.tstep * (a[5 * W] + gz);
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
    const Philox p = philox(mb.seed, (uint32_t)w, (uint32_t)e, PQA_STREAM_ACCEPT, mb.step);
    u = u01(p.c[0], p.c[1]);
  }
  const bool acc = ratio > u;
  if (mb.dmc) {  // dmc.py:68 r2 = |gauss + drift|^2
    const double rx = a0 + a[3 * W], ry = a1 + a[4 * W], rz = a2 + a[5 * W];
    const double r2 = rx * rx + ry * ry + rz * rz;
    mb.r2_prop[w] += r2;
    if (acc) mb.r2_acc[w] += r2;
  }
  mb.accept[w] = acc;
  act[w] = acc;
  if (mb.accept_rec) mb.accept_rec[(size_t)e * W + w] = acc;
  if (!acc) return;
  mb.acc_w[w] += 1;
  double* xe = L.xt + (size_t)e * 3 * W + w;
  xe[0] = mb.newpos[3 * w]; xe[W] = mb.newpos[3 * w + 1]; xe[2 * W] = mb.newpos[3 * w + 2];
  if (mb.wrap) {
    int* wr = mb.wrap + ((size_t)w * S.nelec + e) * 3;
    wr[0] += mb.dwrap[3 * w]; wr[1] += mb.dwrap[3 * w + 1]; wr[2] += mb.dwrap[3 * w + 2];
  }
  {
    const double* row = motmp + (size_t)w * 5 * nmo;
    const int* occ = S.det_occ[s];
    const int nh = nmo / CF;
    if (!CX && S.occ_ident[s] && ((n | nmo) & 3) == 0) {  // ground-state occupation: the value row 32 bytes per load (k_move_part_lw)
#pragma unroll 2
      for (int k = 0; k < n; k += 4) {
        const double4 a = *reinterpret_cast<const double4*>(row + k);
        Vbuf[(size_t)k * W + w] = a.x; Vbuf[(size_t)(k + 1) * W + w] = a.y; Vbuf[(size_t)(k + 2) * W + w] = a.z; Vbuf[(size_t)(k + 3) * W + w] = a.w;
      }
    } else
#pragma unroll 8
    for (int k = 0; k < n; ++k) {
      if (CX) { Vbuf[(size_t)(2 * k) * W + w] = row[occ[k]]; Vbuf[(size_t)(2 * k + 1) * W + w] = row[nh + occ[k]]; }
      else Vbuf[(size_t)k * W + w] = row[occ[k]];
    }
  }
  const double* Ti = L.Tt[s] + (size_t)i * n * CF * W + w;
  if (CX) {  // dsign *= ratio / |ratio|, dlog += log|ratio|; R = T_old[i] / ratio (complex)
    const double m2 = dr * dr + di * di, m = sqrt(m2), ur = dr / m, ui = di / m;
    double* ds = L.dsign[s] + 2 * w;
    const double sr = ds[0], si = ds[1];
    ds[0] = sr * ur - si * ui; ds[1] = sr * ui + si * ur;
    L.dlog[s][w] += 0.5 * log(m2);
    const double ir = dr / m2, ii = -di / m2;  // 1 / ratio
#pragma unroll 8
    for (int k = 0; k < n; ++k) {
      const double tr = Ti[(size_t)(2 * k) * W], ti = Ti[(size_t)(2 * k + 1) * W];
      Rbuf[(size_t)(2 * k) * W + w] = tr * ir - ti * ii;
      Rbuf[(size_t)(2 * k + 1) * W + w] = tr * ii + ti * ir;
    }
  } else {
    L.dsign[s][w] *= (dr > 0.0) ? 1.0 : ((dr < 0.0) ? -1.0 : dr);
    L.dlog[s][w] += log(fabs(dr));
    const double inv = 1.0 / dr;
#pragma unroll 8
    for (int k = 0; k < n; ++k) Rbuf[(size_t)k * W + w] = Ti[(size_t)k * W] * inv;
  }
}

// ---------------------------------------------------------------- commit (Sherman-Morrison, slater.py:88-94)
// Blocked update.  Electrons of one spin are moved in index order, so a ratio or drift only ever needs the
// inverse rows of electrons that have not moved yet in this sweep plus the current one.  The electrons are
// grouped in blocks of KB; an accepted move of electron i updates immediately only the KB rows of its block
//   T[j][k] -= R[k] * (V . T[j])   (j != i),     T[i][k] = R[k],     R = T_old[i]/ratio, V = new orbital row
// and leaves (V, R) in the block buffers Vb/Rb[q][n][W] (q = position in the block, act[q][W] = accepted).
// After the last electron of a block k_flush_lw applies the block's accepted updates, in order, to every
// row outside the block while that row sits in registers.  Per row the arithmetic and its order are exactly
// those of updating after every move, so the inverse is bitwise identical — but it crosses HBM once per
// block instead of once per move (512*KB + 16384/KB bytes per move at n = 32: 2.7x less at KB = 8).
// thread = (walker, row group g of G).  The 5*nmo cached orbital values of electron i are refreshed in slices
// by the same groups.  NMAX >= n.
// NMAX >= doubles per row (n real, 2 n complex)
template <int NMAX, bool FULLLINE, bool CX = false>
__global__ __launch_bounds__(64) void k_commit_lw(SysDev S, LwState L, MoveBuf mb, int e, const double* __restrict__ motmp,
                                                  const double* __restrict__ Rbuf, const double* __restrict__ Vbuf, long W,
                                                  int G, int j_lo, int j_hi) {
  const long w = (long)blockIdx.x * 64 + threadIdx.x;
  const int g = blockIdx.y;
  if (w >= W) return;
  // Rejected walkers take part in the loads and stores of the inverse rows (writing back what they read): a cache
  // line holds 8 walkers, so it is fetched and written whenever one of them accepted anyway, and with every lane
  // storing, the wave writes whole lines instead of byte-masked fragments.
  const bool acc = mb.accept[w] != 0;
  if (FULLLINE ? !__any(acc) : !acc) return;
  const int s = e >= S.nup, i = e - s * S.nup, n = s ? S.ndn : S.nup, nmo = S.nmo[s];
  const int L_ = CX ? 2 * n : n;  // doubles per row
  double* T = L.Tt[s] + w;
  double V[NMAX], R[NMAX];
#pragma unroll
  for (int k = 0; k < NMAX; ++k) {
    V[k] = (k < L_ && acc) ? Vbuf[(size_t)k * W + w] : 0.0;
    R[k] = (k < L_ && acc) ? Rbuf[(size_t)k * W + w] : 0.0;
  }
  for (int j = j_lo + g; j < j_hi; j += G) {
    double* Tj = T + (size_t)j * L_ * W;
    if (j == i) {
      if (acc) {
#pragma unroll
        for (int k = 0; k < NMAX; ++k)
          if (k < L_) Tj[(size_t)k * W] = R[k];
      }
      continue;
    }
    double t[NMAX];
    double tmp = 0.0, tmi = 0.0;
#pragma unroll
    for (int k = 0; k < NMAX; ++k) t[k] = (k < L_) ? Tj[(size_t)k * W] : 0.0;
    if (CX) {
#pragma unroll
      for (int k = 0; k < NMAX / 2; ++k) {  // tmp = sum_k V_k t_k (complex, no conjugation)
        tmp += V[2 * k] * t[2 * k] - V[2 * k + 1] * t[2 * k + 1];
        tmi += V[2 * k] * t[2 * k + 1] + V[2 * k + 1] * t[2 * k];
      }
#pragma unroll
      for (int k = 0; k < NMAX / 2; ++k)
        if (2 * k < L_) {
          const double ur = R[2 * k] * tmp - R[2 * k + 1] * tmi, ui = R[2 * k] * tmi + R[2 * k + 1] * tmp;
          Tj[(size_t)(2 * k) * W] = acc ? t[2 * k] - ur : t[2 * k];
          Tj[(size_t)(2 * k + 1) * W] = acc ? t[2 * k + 1] - ui : t[2 * k + 1];
        }
    } else {
#pragma unroll
      for (int k = 0; k < NMAX; ++k) tmp += V[k] * t[k];
#pragma unroll
      for (int k = 0; k < NMAX; ++k)
        if (k < L_) Tj[(size_t)k * W] = acc ? t[k] - R[k] * tmp : t[k];
    }
  }
  if (!acc) return;
  const double* row = motmp + (size_t)w * 5 * nmo;
  double* c = L.ct[s] + (size_t)i * 5 * nmo * W + w;
  const int nk = (5 * nmo + G - 1) / G, kb = g * nk, ke = (kb + nk < 5 * nmo) ? kb + nk : 5 * nmo;  // contiguous slice: whole lines of the point-major row
#pragma unroll 8
  for (int k = kb; k < ke; ++k) c[(size_t)k * W] = row[k];
}

// rows outside [j_lo, j_hi) of spin s: apply the block's nq buffered updates in order.  Vb/Rb: [KB][n][W], act: [KB][W].
// Block = 16 walkers x 16 row groups: the block's update vectors (V_q, R_q of its 16 walkers, 2 nq n 16 doubles) are staged
// in LDS once and shared by the row groups — read per row from global memory they were 4x the traffic of the inverse
// itself (1.26 ms per flush at 65 536 walkers).  A row's arithmetic and its order are unchanged (bitwise identical).
// 16 consecutive walkers are 128 contiguous bytes of every (row, column) plane: two full cache lines per access.
#ifndef PQA_FLUSH_WB
#define PQA_FLUSH_WB 16
#endif
template <int NMAX, bool CX = false>
__global__ __launch_bounds__(256) void k_flush_lw(SysDev S, LwState L, int s, const double* __restrict__ Vb,
                                                  const double* __restrict__ Rb, const uint8_t* __restrict__ act, long W,
                                                  int j_lo, int j_hi, int nq) {
  extern __shared__ double sh[];
  const int n = s ? S.ndn : S.nup;
  const int L_ = CX ? 2 * n : n;  // doubles per row
  double* shV = sh;
  double* shR = sh + (size_t)nq * L_ * PQA_FLUSH_WB;
  const int wl = threadIdx.x & (PQA_FLUSH_WB - 1), g = threadIdx.x / PQA_FLUSH_WB;
  const long w0 = (long)blockIdx.x * PQA_FLUSH_WB;
  for (int idx = threadIdx.x; idx < nq * L_ * PQA_FLUSH_WB; idx += 256) {
    const long ws = (w0 + (idx & (PQA_FLUSH_WB - 1)) < W) ? w0 + (idx & (PQA_FLUSH_WB - 1)) : W - 1;
    const size_t src = (size_t)(idx / PQA_FLUSH_WB) * W + ws;  // idx / WB = q * L_ + k
    shV[idx] = Vb[src];
    shR[idx] = Rb[src];
  }
  __syncthreads();
  const long w = w0 + wl;
  if (w >= W) return;
  unsigned mask = 0;
  for (int q = 0; q < nq; ++q) mask |= act[(size_t)q * W + w] ? (1u << q) : 0u;
  if (!mask) return;
  double* T = L.Tt[s] + w;
  const int nout = n - (j_hi - j_lo);
  for (int jj = g; jj < nout; jj += 256 / PQA_FLUSH_WB) {
    const int j = (jj < j_lo) ? jj : jj + (j_hi - j_lo);
    double* Tj = T + (size_t)j * L_ * W;
    double t[NMAX];
#pragma unroll
    for (int k = 0; k < NMAX; ++k) t[k] = (k < L_) ? Tj[(size_t)k * W] : 0.0;
    for (int q = 0; q < nq; ++q) {
      if (!((mask >> q) & 1u)) continue;
      const double* Vq = shV + (size_t)q * L_ * PQA_FLUSH_WB + wl;
      const double* Rq = shR + (size_t)q * L_ * PQA_FLUSH_WB + wl;
      if (CX) {
        double tmp = 0.0, tmi = 0.0;
#pragma unroll
        for (int k = 0; k < NMAX / 2; ++k)
          if (2 * k < L_) {
            const double vr = Vq[(2 * k) * PQA_FLUSH_WB], vi = Vq[(2 * k + 1) * PQA_FLUSH_WB];
            tmp += vr * t[2 * k] - vi * t[2 * k + 1];
            tmi += vr * t[2 * k + 1] + vi * t[2 * k];
          }
#pragma unroll
        for (int k = 0; k < NMAX / 2; ++k)
          if (2 * k < L_) {
            const double rr = Rq[(2 * k) * PQA_FLUSH_WB], ri = Rq[(2 * k + 1) * PQA_FLUSH_WB];
            t[2 * k] -= rr * tmp - ri * tmi;
            t[2 * k + 1] -= rr * tmi + ri * tmp;
          }
      } else {
        double tmp = 0.0;
#pragma unroll
        for (int k = 0; k < NMAX; ++k)
          if (k < L_) tmp += Vq[k * PQA_FLUSH_WB] * t[k];
#pragma unroll
        for (int k = 0; k < NMAX; ++k)
          if (k < L_) t[k] = t[k] - Rq[k * PQA_FLUSH_WB] * tmp;
      }
    }
#pragma unroll
    for (int k = 0; k < NMAX; ++k)
      if (k < L_) Tj[(size_t)k * W] = t[k];
  }
}

// ---------------------------------------------------------------- kinetic + Coulomb
// thread = (walker, electron), walker fastest.  part [4][N][W]: ke_e, grad2_e, ee_e, ei_e
// block = (64 walkers, PQA_KIN_EB electrons): one wave per electron, so the electron index — and with it the spin, the orbital
// occupation list and every Jastrow table address — must stay wave-uniform (scalar loads): threadIdx.y goes through
// readfirstlane.  (As a plain per-lane value it turned the occupation look-up of the inner loop into a dependent vector load:
// 2.0 -> 3.3 ms per evaluation.)  Every electron's thread walks ALL coordinates of its walker for the Jastrow and Coulomb sums,
// so the 64 electron-waves of a walker group pull the group's coordinates through the fabric 64 times (the counters show
// 195 KB per walker against 96 KB of inverse + cache rows; the re-reads hit the Infinity Cache).
#ifndef PQA_KIN_EB
#define PQA_KIN_EB 1
#endif
template <bool PBC, bool CX = false>
__global__ __launch_bounds__(64 * PQA_KIN_EB) void k_kinetic_lw(SysDev S, LwState L, int has_jastrow, long W, double* __restrict__ part) {
  const long w = (long)blockIdx.x * 64 + threadIdx.x;
  const int e = blockIdx.y * PQA_KIN_EB + (PQA_KIN_EB > 1 ? __builtin_amdgcn_readfirstlane((int)threadIdx.y) : 0);
  if (w >= W || e >= S.nelec) return;
  const int s = e >= S.nup, i = e - s * S.nup, n = s ? S.ndn : S.nup, nmo = S.nmo[s];
  constexpr int CF = CX ? 2 : 1;
  double r[5] = {0.0, 0.0, 0.0, 0.0, 0.0}, q[5] = {0.0, 0.0, 0.0, 0.0, 0.0};
  {
    const double* Ti = L.Tt[s] + (size_t)i * n * CF * W + w;
    const double* ci = L.ct[s] + (size_t)i * 5 * nmo * W + w;
    const int* occ = S.det_occ[s];
    const int nh = nmo / CF;
    for (int j = 0; j < n; ++j) {
      const double* cj = ci + (size_t)occ[j] * W;
      if (CX) {
        const double tr = Ti[(size_t)(2 * j) * W], ti = Ti[(size_t)(2 * j + 1) * W];
#pragma unroll
        for (int c = 0; c < 5; ++c) {
          const double a = cj[(size_t)c * nmo * W], b = cj[((size_t)c * nmo + nh) * W];
          r[c] += a * tr - b * ti; q[c] += a * ti + b * tr;
        }
      } else {
        const double t = Ti[(size_t)j * W];
#pragma unroll
        for (int c = 0; c < 5; ++c) r[c] += cj[(size_t)c * nmo * W] * t;
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
  jas_eval_lane<2, PBC>(S, L.xt, W, w, e, xe[0], xe[W], xe[2 * W], has_jastrow, 0, 1, U, gj, lj, ee, ei);
  lj += gj[0] * gj[0] + gj[1] * gj[1] + gj[2] * gj[2];
  const double gx = gs0 + gj[0], gy = gs1 + gj[1], gz = gs2 + gj[2];
  const double lap = ls + lj + 2.0 * (gs0 * gj[0] + gs1 * gj[1] + gs2 * gj[2]);
  const size_t o = (size_t)e * W + w, NW = (size_t)S.nelec * W;
  part[o] = -0.5 * lap;
  part[NW + o] = gx * gx + gy * gy + gz * gz + gi2;
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
