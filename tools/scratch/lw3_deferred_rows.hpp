// EXPERIMENT RECORD (round 2, not compiled): deferred-row lane-per-walker sweep, 3 launches per move.
// Measured slower than the six-launch sweep at every walker count (profiles/r02_lw3_experiment_*): the thread-per-walker
// finish kernel (160 us) and the doubled Jastrow partial-sum kernel (161 us) cost more than the launches and bytes they save.

// ================================================================ deferred-row sweep (three launches per move)
// The kernels above stream the walker state through HBM six launches per move: the coordinates twice (old-position and
// new-position Jastrow sums), the KB block rows of the inverse on every commit, the orbital rows three times.  Here the
// inverse is not touched between flushes at all:
//   * a row of the inverse is needed exactly once per sweep — when its own electron moves.  It is MATERIALISED then from
//     the flushed inverse T0 by applying, in order, the block's buffered accepted updates (V_q, R_q) — per row exactly
//     the operations, in the order, of updating after every move (slater.py:88-94), so the inverse stays bit-identical
//     for every block size — and kept in Tc[n][W] for the move's ratio sums and its R = Tc / ratio;
//   * k_flush2_lw applies the block's updates to ALL rows once per block (rows inside the block are replaced by R_q at
//     their own update);
//   * the Jastrow sums at the OLD position of the next electron ride along in the partial-sum kernel of the current
//     move (same x_j loads): they are summed without the pair (next, current), whose two candidates (current electron
//     at its old / proposed position) are stored separately and selected by the Metropolis decision;
//   * one thread-per-walker kernel does the decision for electron e, stages (V, R), materialises the row of e+1, takes
//     the Slater drift of e+1 from the orbital cache and makes the proposal for e+1.
// Per move: k_orb<5>, k_part2_lw, k_fin2_lw (+ one flush and a split k_fin2_lw per block of KB electrons).
#define PQA_PART2_ROWS 12  // 0..3 Slater sums at the proposal; 4..7 U, grad U at the proposal; 8..11 U, grad U of the next electron (old position)

// b-function pair term of an electron at displacement d from electron j: adds c.b(r), c.(db/dr)/r.d
__device__ __forceinline__ void lw_pair_term(const SysDev& S, double dx, double dy, double dz, int col, double irb, double& u, double& gx,
                                             double& gy, double& gz) {
  const double r = sqrt(dx * dx + dy * dy + dz * dz);
  if (r < S.rcut_b) {
    const RadShared sh = rad_shared<1>(r, irb);
    double sg = 0.0;
    for (int l = 0; l < S.nb; ++l) {
      double v, gf, lpl;
      rad_fn<1>(S.b_kind[l], S.b_param[l], S.b_aux[l], S.rcut_b, sh, v, gf, lpl);
      const double c = S.bcoeff[l * 3 + col];
      u += c * v;
      sg += c * gf;
    }
    gx += sg * dx; gy += sg * dy; gz += sg * dz;
  }
}
__device__ __forceinline__ void lw_ion_term(const SysDev& S, double dx, double dy, double dz, int I, int edown, double ira, double& u,
                                            double& gx, double& gy, double& gz) {
  const double r = sqrt(dx * dx + dy * dy + dz * dz);
  if (r < S.rcut_a) {
    const RadShared sh = rad_shared<1>(r, ira);
    double sg = 0.0;
    for (int k = 0; k < S.na; ++k) {
      double v, gf, lpl;
      rad_fn<1>(S.a_kind[k], S.a_param[k], S.a_aux[k], S.rcut_a, sh, v, gf, lpl);
      const double c = S.acoeff[(I * S.na + k) * 2 + edown];
      u += c * v;
      sg += c * gf;
    }
    gx += sg * dx; gy += sg * dy; gz += sg * dz;
  }
}

// thread = (walker, group g of G).  ec: electron being moved (proposal newpos[W][3], orbital rows `rows` [W][5][nmo], current
// inverse row Tc[n][W]) or -1; en: next electron (old-position Jastrow sums) or -1; er: electron whose cache row is refreshed
// from rows_prev for the walkers with act_prev set, or -1.  pair[8][W]: (U, grad U) of the pair (en, ec at its old
// position), then of (en, ec at the proposal).
template <bool PBC>
__global__ __launch_bounds__(64) void k_part2_lw(SysDev S, LwState L, int ec, int en, int er, int has_jastrow,
                                                 const double* __restrict__ newpos, const double* __restrict__ rows,
                                                 const double* __restrict__ rows_prev, const uint8_t* __restrict__ act_prev,
                                                 const double* __restrict__ Tc, long W, int G, double* __restrict__ part,
                                                 double* __restrict__ pair) {
  const long w = (long)blockIdx.x * 64 + threadIdx.x;
  const int g = blockIdx.y;
  if (w >= W) return;
  if (er >= 0 && act_prev[w]) {  // cached orbital rows of the electron accepted one move ago
    const int s = er >= S.nup, i = er - s * S.nup, nmo = S.nmo[s];
    const double* row = rows_prev + (size_t)w * 5 * nmo;
    double* c = L.ct[s] + (size_t)i * 5 * nmo * W + w;
    const int nk = (5 * nmo + G - 1) / G, kb = g * nk, ke = (kb + nk < 5 * nmo) ? kb + nk : 5 * nmo;  // whole lines of the point-major row
#pragma unroll 8
    for (int k = kb; k < ke; ++k) c[(size_t)k * W] = row[k];
  }
  if (ec < 0 && en < 0) return;
  double* p = part + (size_t)g * PQA_PART2_ROWS * W + w;
  double px = 0.0, py = 0.0, pz = 0.0, qx = 0.0, qy = 0.0, qz = 0.0;
  if (ec >= 0) {
    px = newpos[3 * w]; py = newpos[3 * w + 1]; pz = newpos[3 * w + 2];
    const int s = ec >= S.nup, n = s ? S.ndn : S.nup, nmo = S.nmo[s];
    const int* occ = S.det_occ[s];
    const int nj = (n + G - 1) / G, jb = g * nj, je = (jb + nj < n) ? jb + nj : n;  // contiguous slots: whole lines of the row
    const double* row = rows + (size_t)w * 5 * nmo;
    const double* Ti = Tc + w;
    double r0 = 0.0, r1 = 0.0, r2 = 0.0, r3 = 0.0;
#pragma unroll 4
    for (int j = jb; j < je; ++j) {
      const double t = Ti[(size_t)j * W];
      const int o = occ[j];
      r0 += row[o] * t; r1 += row[nmo + o] * t; r2 += row[2 * nmo + o] * t; r3 += row[3 * nmo + o] * t;
    }
    p[0] = r0; p[W] = r1; p[2 * W] = r2; p[3 * W] = r3;
  }
  if (en >= 0) { const double* xe = L.xt + (size_t)en * 3 * W + w; qx = xe[0]; qy = xe[W]; qz = xe[2 * W]; }
  double cu = 0.0, cx = 0.0, cy = 0.0, cz = 0.0, nu = 0.0, nx = 0.0, ny = 0.0, nz = 0.0;
  if (has_jastrow) {
    const int cdown = ec >= S.nup, ndown = en >= S.nup;
    const double irb = 1.0 / S.rcut_b, ira = 1.0 / S.rcut_a;
#pragma unroll 2
    for (int j = g; j < S.nelec; j += G) {
      const double* xj = L.xt + (size_t)j * 3 * W + w;
      const double jx = xj[0], jy = xj[W], jz = xj[2 * W];
      const int jdown = j >= S.nup;
      if (ec >= 0 && j != ec) {
        double dx = px - jx, dy = py - jy, dz = pz - jz;
        if (PBC) min_image(S, dx, dy, dz);
        lw_pair_term(S, dx, dy, dz, cdown + jdown, irb, cu, cx, cy, cz);
      }
      if (en >= 0 && j != en) {
        double dx = qx - jx, dy = qy - jy, dz = qz - jz;
        if (PBC) min_image(S, dx, dy, dz);
        if (j == ec) {  // (wave-uniform) the pair whose partner is being moved: both candidates, chosen by the decision
          double ou = 0.0, ox = 0.0, oy = 0.0, oz = 0.0, mu = 0.0, mx = 0.0, my = 0.0, mz = 0.0;
          lw_pair_term(S, dx, dy, dz, ndown + jdown, irb, ou, ox, oy, oz);
          double ex = qx - px, ey = qy - py, ez = qz - pz;
          if (PBC) min_image(S, ex, ey, ez);
          lw_pair_term(S, ex, ey, ez, ndown + jdown, irb, mu, mx, my, mz);
          double* pr = pair + w;
          pr[0] = ou; pr[W] = ox; pr[2 * W] = oy; pr[3 * W] = oz; pr[4 * W] = mu; pr[5 * W] = mx; pr[6 * W] = my; pr[7 * W] = mz;
        } else lw_pair_term(S, dx, dy, dz, ndown + jdown, irb, nu, nx, ny, nz);
      }
    }
    for (int I = g; I < S.natom; I += G) {
      const double ax = S.atom_xyz[3 * I], ay = S.atom_xyz[3 * I + 1], az = S.atom_xyz[3 * I + 2];
      if (ec >= 0) {
        double dx = px - ax, dy = py - ay, dz = pz - az;
        if (PBC) min_image(S, dx, dy, dz);
        lw_ion_term(S, dx, dy, dz, I, cdown, ira, cu, cx, cy, cz);
      }
      if (en >= 0) {
        double dx = qx - ax, dy = qy - ay, dz = qz - az;
        if (PBC) min_image(S, dx, dy, dz);
        lw_ion_term(S, dx, dy, dz, I, ndown, ira, nu, nx, ny, nz);
      }
    }
  }
  if (ec >= 0) { p[4 * W] = cu; p[5 * W] = cx; p[6 * W] = cy; p[7 * W] = cz; }
  if (en >= 0) { p[8 * W] = nu; p[9 * W] = nx; p[10 * W] = ny; p[11 * W] = nz; }
}

// thread = walker.  DOA: Metropolis decision for electron ec (position q of its block; mc.py:124-137 / dmc.py:57-70): accepted
// walkers move the coordinate, update sign/log of the determinant and stage V = new orbital row (slot order), R = Tc / ratio
// in Vb/Rb[q][n][W]; act[q][W] = decision.  DOB: proposal for electron en (mc.py:117-121): its inverse row from T0 and the
// block's buffered updates (with DOA: positions 0..q, the last one from registers; without: none — the block was flushed),
// written to Tc; Slater drift from the orbital cache, Jastrow drift from the partial sums (+ the pair with the electron
// just moved, chosen by the decision: this kernel's, or act_sel[W] of the previous launch, or none at the sweep's start).
template <bool DOA, bool DOB, int NMAX>
__global__ __launch_bounds__(64) void k_fin2_lw(SysDev S, LwState L, MoveBuf mb, int ec, int en, int q, int has_jastrow, long W, int G,
                                                const double* __restrict__ part, const double* __restrict__ pair,
                                                const double* __restrict__ motmp, double* __restrict__ Tc, double* __restrict__ Vb,
                                                double* __restrict__ Rb, uint8_t* __restrict__ act, const uint8_t* __restrict__ act_sel) {
  const long w = (long)blockIdx.x * 64 + threadIdx.x;
  if (w >= W) return;
  bool acc = false;
  double V[NMAX], R[NMAX];
  if (DOA) {
    const int s = ec >= S.nup, i = ec - s * S.nup, n = s ? S.ndn : S.nup, nmo = S.nmo[s];
    double v[8];
#pragma unroll
    for (int c = 0; c < 8; ++c) v[c] = 0.0;
    for (int g = 0; g < G; ++g) {
      const double* pp = part + (size_t)g * PQA_PART2_ROWS * W + w;
#pragma unroll
      for (int c = 0; c < 8; ++c) v[c] += pp[(size_t)c * W];
    }
    double gx = finite_or(v[1] / v[0], 0.0) + v[5], gy = finite_or(v[2] / v[0], 0.0) + v[6], gz = finite_or(v[3] / v[0], 0.0) + v[7];
    const double* a = L.auxt + w;
    double val = finite_or(v[0], 1.0);
    if (has_jastrow) val *= exp(v[4] - a[6 * W]);
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
    double ratio = val * val * t_prob;
    if (mb.dmc) {
      const double dv = finite_or(v[0], 1.0);  // the Jastrow ratio is positive: np.sign(psi_ratio) is the determinant's
      ratio *= (dv > 0.0) ? 1.0 : ((dv < 0.0) ? -1.0 : 0.0);  // fixed node (dmc.py:64-66)
    }
    double u;
    if (mb.unif) u = mb.unif[(size_t)ec * W + w];
    else {
      const Philox ph = philox(mb.seed, (uint32_t)w, (uint32_t)ec, PQA_STREAM_ACCEPT, mb.step);
      u = u01(ph.c[0], ph.c[1]);
    }
    acc = ratio > u;
    if (mb.dmc) {  // dmc.py:68 r2 = |gauss + drift|^2
      const double rx = a0 + a[3 * W], ry = a1 + a[4 * W], rz = a2 + a[5 * W];
      const double r2 = rx * rx + ry * ry + rz * rz;
      mb.r2_prop[w] += r2;
      if (acc) mb.r2_acc[w] += r2;
    }
    mb.accept[w] = acc;
    act[(size_t)q * W + w] = acc;
    if (mb.accept_rec) mb.accept_rec[(size_t)ec * W + w] = acc;
    if (acc) {
      mb.acc_w[w] += 1;
      double* xe = L.xt + (size_t)ec * 3 * W + w;
      xe[0] = mb.newpos[3 * w]; xe[W] = mb.newpos[3 * w + 1]; xe[2 * W] = mb.newpos[3 * w + 2];
      if (mb.wrap) {
        int* wr = mb.wrap + ((size_t)w * S.nelec + ec) * 3;
        wr[0] += mb.dwrap[3 * w]; wr[1] += mb.dwrap[3 * w + 1]; wr[2] += mb.dwrap[3 * w + 2];
      }
      const double dr = v[0];  // determinant ratio
      L.dsign[s][w] *= (dr > 0.0) ? 1.0 : ((dr < 0.0) ? -1.0 : dr);
      L.dlog[s][w] += log(fabs(dr));
      const double inv = 1.0 / dr;
      const double* row = motmp + (size_t)w * 5 * nmo;
      const int* occ = S.det_occ[s];
      double* vq = Vb + (size_t)q * n * W + w;
      double* rq = Rb + (size_t)q * n * W + w;
#pragma unroll
      for (int k = 0; k < NMAX; ++k) {
        V[k] = (k < n) ? row[occ[k]] : 0.0;
        R[k] = (k < n) ? Tc[(size_t)k * W + w] * inv : 0.0;
        if (k < n) { vq[(size_t)k * W] = V[k]; rq[(size_t)k * W] = R[k]; }
      }
    }
    (void)i;
  }
  if (DOB) {
    const int s = en >= S.nup, i = en - s * S.nup, n = s ? S.ndn : S.nup, nmo = S.nmo[s];
    double t[NMAX];
    const double* T0 = L.Tt[s] + (size_t)i * n * W + w;
#pragma unroll
    for (int k = 0; k < NMAX; ++k) t[k] = (k < n) ? T0[(size_t)k * W] : 0.0;
    if (DOA) {
      for (int qq = 0; qq < q; ++qq) {
        if (!act[(size_t)qq * W + w]) continue;
        const double* vq = Vb + (size_t)qq * n * W + w;
        const double* rq = Rb + (size_t)qq * n * W + w;
        double tmp = 0.0;
#pragma unroll
        for (int k = 0; k < NMAX; ++k)
          if (k < n) tmp += vq[(size_t)k * W] * t[k];
#pragma unroll
        for (int k = 0; k < NMAX; ++k)
          if (k < n) t[k] = t[k] - rq[(size_t)k * W] * tmp;
      }
      if (acc) {
        double tmp = 0.0;
#pragma unroll
        for (int k = 0; k < NMAX; ++k) tmp += V[k] * t[k];
#pragma unroll
        for (int k = 0; k < NMAX; ++k) t[k] = t[k] - R[k] * tmp;
      }
    }
    double r0 = 0.0, r1 = 0.0, r2 = 0.0, r3 = 0.0;
    {
      const double* ci = L.ct[s] + (size_t)i * 5 * nmo * W + w;
      const int* occ = S.det_occ[s];
#pragma unroll
      for (int k = 0; k < NMAX; ++k) {
        if (k < n) {
          Tc[(size_t)k * W + w] = t[k];
          const double* cj = ci + (size_t)occ[k] * W;
          r0 += cj[0] * t[k]; r1 += cj[(size_t)nmo * W] * t[k]; r2 += cj[(size_t)2 * nmo * W] * t[k]; r3 += cj[(size_t)3 * nmo * W] * t[k];
        }
      }
    }
    double U0 = 0.0, jx = 0.0, jy = 0.0, jz = 0.0;
    if (has_jastrow) {
      for (int g = 0; g < G; ++g) {
        const double* pp = part + ((size_t)g * PQA_PART2_ROWS + 8) * W + w;
        U0 += pp[0]; jx += pp[W]; jy += pp[2 * W]; jz += pp[3 * W];
      }
      const bool have_pair = DOA || act_sel != nullptr;
      if (have_pair) {
        const bool accp = DOA ? acc : (act_sel[w] != 0);
        const double* pr = pair + (accp ? 4 : 0) * W + w;
        U0 += pr[0]; jx += pr[W]; jy += pr[2 * W]; jz += pr[3 * W];
      }
    }
    double gx = finite_or(r1 / r0, 0.0) + jx, gy = finite_or(r2 / r0, 0.0) + jy, gz = finite_or(r3 / r0, 0.0) + jz;
    if (mb.dmc) limdrift_dmc(gx, gy, gz, mb.tstep);  // the drift vector itself (dmc.py:50-52)
    else limdrift3(gx, gy, gz);
    double z0, z1, z2, z3;
    if (mb.gauss) {
      const double* zt = mb.gauss + ((size_t)en * W + w) * 3;
      z0 = zt[0]; z1 = zt[1]; z2 = zt[2];
    } else {
      normal2(philox(mb.seed, (uint32_t)w, (uint32_t)en, PQA_STREAM_GAUSS_A, mb.step), z0, z1);
      normal2(philox(mb.seed, (uint32_t)w, (uint32_t)en, PQA_STREAM_GAUSS_B, mb.step), z2, z3);
    }
    const double sq = sqrt(mb.tstep);
    z0 *= sq; z1 *= sq; z2 *= sq;
    const double* xe = L.xt + (size_t)en * 3 * W + w;
    double* np_ = mb.newpos + 3 * w;
    const double df = mb.dmc ? 1.0 : mb.tstep;
    np_[0] = xe[0] + z0 + gx * df;
    np_[1] = xe[W] + z1 + gy * df;
    np_[2] = xe[2 * W] + z2 + gz * df;
    if (mb.dwrap) fold_cell(S, np_[0], np_[1], np_[2], mb.dwrap + 3 * w);  // make_irreducible, mc.py:121
    double* a = L.auxt + w;
    a[0] = z0; a[W] = z1; a[2 * W] = z2; a[3 * W] = gx; a[4 * W] = gy; a[5 * W] = gz; a[6 * W] = U0;
  }
}

// All rows of spin s: apply the nq buffered updates of the block [j_lo, j_lo + nq) in order; row j_lo + q' is REPLACED by
// R_q' at its own update.  Block = 16 walkers x 16 row groups; the update vectors of QC block positions at a time
// (2 QC n 16 doubles) are staged in LDS and shared by the row groups.  16 consecutive walkers are 128 contiguous bytes of
// every (row, column) plane: whole cache lines.
template <int NMAX>
__global__ __launch_bounds__(256) void k_flush2_lw(SysDev S, LwState L, int s, const double* __restrict__ Vb,
                                                   const double* __restrict__ Rb, const uint8_t* __restrict__ act, long W,
                                                   int j_lo, int nq, int QC) {
  extern __shared__ double sh[];
  const int n = s ? S.ndn : S.nup;
  double* shV = sh;
  double* shR = sh + (size_t)QC * n * PQA_FLUSH_WB;
  const int wl = threadIdx.x & (PQA_FLUSH_WB - 1), g = threadIdx.x / PQA_FLUSH_WB;
  const long w0 = (long)blockIdx.x * PQA_FLUSH_WB;
  const long w = w0 + wl;
  unsigned long long mask = 0ull;
  if (w < W)
    for (int qq = 0; qq < nq; ++qq) mask |= act[(size_t)qq * W + w] ? (1ull << qq) : 0ull;
  for (int q0 = 0; q0 < nq; q0 += QC) {
    const int qc = (nq - q0 < QC) ? nq - q0 : QC;
    __syncthreads();
    for (int idx = threadIdx.x; idx < qc * n * PQA_FLUSH_WB; idx += 256) {
      const long ws = (w0 + (idx & (PQA_FLUSH_WB - 1)) < W) ? w0 + (idx & (PQA_FLUSH_WB - 1)) : W - 1;
      const size_t src = ((size_t)q0 * n + idx / PQA_FLUSH_WB) * W + ws;  // idx / WB = q * n + k
      shV[idx] = Vb[src];
      shR[idx] = Rb[src];
    }
    __syncthreads();
    if (w >= W || !((mask >> q0) & ((qc < 64 ? (1ull << qc) : 0ull) - 1ull))) continue;
    double* T = L.Tt[s] + w;
    for (int j = g; j < n; j += 256 / PQA_FLUSH_WB) {
      double* Tj = T + (size_t)j * n * W;
      double t[NMAX];
#pragma unroll
      for (int k = 0; k < NMAX; ++k) t[k] = (k < n) ? Tj[(size_t)k * W] : 0.0;
      for (int qq = 0; qq < qc; ++qq) {
        if (!((mask >> (q0 + qq)) & 1ull)) continue;
        const double* Rq = shR + (size_t)qq * n * PQA_FLUSH_WB + wl;
        if (j == j_lo + q0 + qq) {
#pragma unroll
          for (int k = 0; k < NMAX; ++k)
            if (k < n) t[k] = Rq[k * PQA_FLUSH_WB];
          continue;
        }
        const double* Vq = shV + (size_t)qq * n * PQA_FLUSH_WB + wl;
        double tmp = 0.0;
#pragma unroll
        for (int k = 0; k < NMAX; ++k)
          if (k < n) tmp += Vq[k * PQA_FLUSH_WB] * t[k];
#pragma unroll
        for (int k = 0; k < NMAX; ++k)
          if (k < n) t[k] = t[k] - Rq[k * PQA_FLUSH_WB] * tmp;
      }
#pragma unroll
      for (int k = 0; k < NMAX; ++k)
        if (k < n) Tj[(size_t)k * W] = t[k];
    }
  }
}
