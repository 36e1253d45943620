// Semi-local ECP integrator, second generation of the list-building passes and the point kernel (round 3).
//
// Reference semantics: pyqmc/observables/eval_ecp.py (ecp_ea :83-132, ecp_mask :135-146, rnExp :182-200, get_P_l :228-252);
// the kernels of pqa_energy.hpp (k_ecp_count / k_ecp_fill / k_ecp_point) stay as the general path and as the A/B partner.
//
// What was wrong with the first generation at 65 536 walkers of the 64-electron cluster (rocprofv3 + PMC, profiles/r03_*):
// k_ecp_count 0.47 ms and k_ecp_fill 0.81 ms moved 0.12 / 0.29 GB — 0.3 TB/s — and k_ecp_point 0.63 ms per launch at
// 2.2 TB/s: none of them was near any throughput limit.  Each was one wave walking a chain of dependent round trips at 3-4
// waves per SIMD: the ECP tables (channel offsets -> term offsets -> terms, three dependent vector loads per channel because
// the atom index differs between lanes), the mask words one at a time, one (electron, atom) entry after the other, one
// partner coordinate at a time.  Here
//   * the ECP tables and the ECP atoms' coordinates are staged in LDS once per block of four walkers,
//   * a block handles four walkers (one wave each),
//   * k_ecp_fill_t reads all mask words of its walker with one load, and FOUR entries are worked on at a time by 16-lane
//     groups (12 or 6 lanes make the quadrature points; the 16 lanes split the partners of the Jastrow exponent),
//   * k_ecp_point_lw fetches 16 slots of the inverse column and the orbital row before the first product and takes the
//     Jastrow exponent from jas_eval_lane (coordinates four partners ahead, function tables in registers).
// Entry order, slot offsets and every per-point quantity are those of the first generation; the old-position Jastrow
// exponent u0 is summed in a different order (16-lane groups instead of one 64-lane tree): last-bit differences.
#pragma once
#include "pqa_energy.hpp"
#include "pqa_lw.hpp"

#ifndef PQA_ECP_WB
#define PQA_ECP_WB 4  // walkers (waves) per block of the list-building passes
#endif

struct EcpTab {  // same member names as the SysDev tables: ecp_radial_t works on either
  const int* ecp_chan_off;
  const int* ecp_term_off;
  const int* ecp_term_n;
  const double* ecp_term_exp;
  const double* ecp_term_coef;
  const double* atom;  // [necp][4]: x, y, z of the ECP atom, r^2 range
  const int* naip;     // [necp] quadrature points of the atom's rule, and the rule's first row in the direction / weight tables
  const int* qoff;
};

__host__ __device__ inline size_t ecp_tab_bytes(int necp, int nchan, int nterm) {
  const size_t d = (size_t)4 * necp + 2 * (size_t)nterm, i = (size_t)(necp + 1) + (nchan + 1) + nterm + 2 * (size_t)necp;
  return d * sizeof(double) + ((i + 1) / 2) * 2 * sizeof(int);
}

// cooperative copy global -> LDS; the caller synchronises
__device__ __forceinline__ EcpTab ecp_stage(const SysDev& S, int nchan, int nterm, double* lds, int tid, int nthreads) {
  double* atom = lds;
  double* ex = atom + 4 * S.necp;
  double* co = ex + nterm;
  int* ci = reinterpret_cast<int*>(co + nterm);
  int* to = ci + (S.necp + 1);
  int* tn = to + (nchan + 1);
  int* na = tn + nterm;
  int* qo = na + S.necp;
  for (int k = tid; k < S.necp; k += nthreads) {
    const int ia = S.ecp_atom[k];
    atom[4 * k] = S.atom_xyz[3 * ia]; atom[4 * k + 1] = S.atom_xyz[3 * ia + 1]; atom[4 * k + 2] = S.atom_xyz[3 * ia + 2];
    atom[4 * k + 3] = S.ecp_rc2[k];
    na[k] = S.ecp_naip[k]; qo[k] = S.ecp_qoff[k];
  }
  for (int t = tid; t < nterm; t += nthreads) { ex[t] = S.ecp_term_exp[t]; co[t] = S.ecp_term_coef[t]; tn[t] = S.ecp_term_n[t]; }
  for (int k = tid; k <= S.necp; k += nthreads) ci[k] = S.ecp_chan_off[k];
  for (int c = tid; c <= nchan; c += nthreads) to[c] = S.ecp_term_off[c];
  EcpTab T;
  T.ecp_chan_off = ci; T.ecp_term_off = to; T.ecp_term_n = tn; T.ecp_term_exp = ex; T.ecp_term_coef = co; T.atom = atom; T.naip = na; T.qoff = qo;
  return T;
}

// ecp_radial (pqa_energy.hpp) on any table holder
template <class TT>
__device__ __forceinline__ void ecp_radial_t(const TT& T, int k, double r, double threshold, double (&v)[PQA_MAXCHAN], int& nch,
                                             double& prob) {
  const int c0 = T.ecp_chan_off[k];
  nch = T.ecp_chan_off[k + 1] - c0;
  double pr = 0.0;
  for (int c = 0; c < nch; ++c) {
    double sum = 0.0;
    for (int t = T.ecp_term_off[c0 + c]; t < T.ecp_term_off[c0 + c + 1]; ++t) {
      const int n = T.ecp_term_n[t];
      double rn = 1.0;
      if (n != 0) {
        const int an = n < 0 ? -n : n;
        double rp = r;
        for (int q = 1; q < an; ++q) rp *= r;
        rn = n < 0 ? 1.0 / rp : rp;
      }
      sum += rn * T.ecp_term_coef[t] * exp(-T.ecp_term_exp[t] * r * r);
    }
    v[c] = sum;
    if (c < nch - 1) pr += fabs(sum) * threshold * (2.0 * (2 * c + 1) + 1.0);  // eval_ecp.py:139-141
  }
  prob = (threshold > 0.0) ? fmin(1.0, pr) : 1.0;
}

__device__ __forceinline__ int wave_excl_scan(int v, int lane, int& total) {
  int x = v;
#pragma unroll
  for (int off = 1; off < 64; off <<= 1) {
    const int y = __shfl_up(x, off, 64);
    if (lane >= off) x += y;
  }
  total = __shfl(x, 63, 64);
  return x - v;
}

// pass A: k_ecp_count for necp <= 64 with the tables in LDS.  grid = ceil(W / PQA_ECP_WB), block = 64 * PQA_ECP_WB, dynamic LDS = ecp_tab_bytes.
template <bool PBC>
static __global__ __launch_bounds__(64 * PQA_ECP_WB) void k_ecp_count_t(SysDev S, JastrowState js, EcpBuf B, int nchan, int nterm, long W) {
  extern __shared__ double lds[];
  __shared__ unsigned long long pb_[PQA_ECP_WB][64];  // electrons of the current block that passed the mask at atom k
  const int lane = threadIdx.x & 63, wv = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  const long w_ = (long)blockIdx.x * PQA_ECP_WB + wv;
  const bool wlive = w_ < W;
  const long w = wlive ? w_ : W - 1;  // surplus waves of the last block go through the barriers on a valid walker and store nothing
  const double* xw = js.x + (size_t)w * S.nelec * 3;
  const int neb = (S.nelec + 63) / 64;
  double cx = 0.0, cy = 0.0, cz = 0.0;
  if (lane < S.nelec) { cx = xw[3 * lane]; cy = xw[3 * lane + 1]; cz = xw[3 * lane + 2]; }  // in flight while the tables are staged
  const EcpTab T = ecp_stage(S, nchan, nterm, lds, (int)threadIdx.x, 64 * PQA_ECP_WB);
  unsigned long long* pb = pb_[wv];
  if (lane < S.necp) pb[lane] = 0ull;
  __syncthreads();
  double loc = 0.0;
  int c_up = 0, c_dn = 0, k_up = 0, k_dn = 0;  // k_*: entries of ECP atom `lane` (atom-major lists)
  for (int eb = 0; eb < neb; ++eb) {
    const int e = eb * 64 + lane;
    const bool live = e < S.nelec;
    if (eb > 0 && live) { cx = xw[3 * e]; cy = xw[3 * e + 1]; cz = xw[3 * e + 2]; }
    const double ex = live ? cx : 0.0, ey = live ? cy : 0.0, ez = live ? cz : 0.0;
    unsigned long long near = 0ull;
    for (int k = 0; k < S.necp; ++k) {
      double dx = ex - T.atom[4 * k], dy = ey - T.atom[4 * k + 1], dz = ez - T.atom[4 * k + 2];
      if (PBC) min_image(S, dx, dy, dz);
      if (live && dx * dx + dy * dy + dz * dz < T.atom[4 * k + 3]) near |= 1ull << k;
    }
    while (__any(near != 0ull)) {
      if (near) {
        const int k = __ffsll((long long)near) - 1;
        near &= near - 1;
        double dx = ex - T.atom[4 * k], dy = ey - T.atom[4 * k + 1], dz = ez - T.atom[4 * k + 2];
        if (PBC) min_image(S, dx, dy, dz);
        const double r = sqrt(dx * dx + dy * dy + dz * dz);
        double v[PQA_MAXCHAN], prob;
        int nch;
        ecp_radial_t(T, k, r, B.threshold, v, nch, prob);
        loc += v[nch - 1];
        if (nch > 1 && ecp_pass(S, B, w, W, e, k, prob)) {
          const int naip = T.naip[k];
          if (e < S.nup) c_up += naip; else c_dn += naip;
          atomicOr(&pb[k], 1ull << lane);
        }
      }
    }
    __syncthreads();
    if (lane < S.necp) {
      const unsigned long long mk = pb[lane];
      if (wlive) B.passbits[((size_t)w * S.necp + lane) * neb + eb] = mk;
      const int nu = S.nup - eb * 64;  // bits below it are spin-up electrons
      const unsigned long long um = (nu >= 64) ? ~0ull : ((nu <= 0) ? 0ull : ((1ull << nu) - 1ull));
      k_up += __popcll(mk & um); k_dn += __popcll(mk & ~um);
      pb[lane] = 0ull;
    }
    __syncthreads();
  }
  loc = wave_sum(loc);
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) { c_up += __shfl_xor(c_up, off, 64); c_dn += __shfl_xor(c_dn, off, 64); }
  if (lane == 0 && wlive) B.local[w] = loc;
  if (B.nseg > 1) {  // atom-major: points of (atom lane, walker w), spin up and spin down
    if (lane < S.necp && wlive) {
      const int naip = T.naip[lane];
      B.cnt[(size_t)lane * W + w] = naip * k_up;
      B.cnt[(size_t)B.nseg * W + (size_t)lane * W + w] = naip * k_dn;
    }
  } else if (lane == 0 && wlive) { B.cnt[w] = c_up; B.cnt[W + w] = c_dn; }
}

// pass B: k_ecp_fill for necp * ceil(N / 64) <= 64.  Same launch shape as k_ecp_count_t.
// UE: the old-position Jastrow exponents come from B.ue (k_kinetic_lw evaluates U_e of every electron anyway and ran just before);
// otherwise the 16 lanes of a group sum the entry's exponent themselves.
template <bool PBC, bool UE>
static __global__ __launch_bounds__(64 * PQA_ECP_WB) void k_ecp_fill_t(SysDev S, JastrowState js, EcpBuf B, int nchan, int nterm, long W) {
  extern __shared__ double lds[];
  __shared__ int el_[PQA_ECP_WB][64];   // entries of the current window: atom << 16 | electron
  __shared__ long eo_[PQA_ECP_WB][64];  // their first slot in the spin's point list
  const int lane = threadIdx.x & 63, wv = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  const int grp = lane >> 4, l16 = lane & 15;
  const long w_ = (long)blockIdx.x * PQA_ECP_WB + wv;
  const bool wlive = w_ < W;
  const long w = wlive ? w_ : W - 1;
  const double* xw = js.x + (size_t)w * S.nelec * 3;
  const int neb = (S.nelec + 63) / 64, nq = S.necp * neb;
  unsigned long long m = (wlive && lane < nq) ? B.passbits[(size_t)w * nq + lane] : 0ull;  // the mask k_ecp_count drew
  const EcpTab T = ecp_stage(S, nchan, nterm, lds, (int)threadIdx.x, 64 * PQA_ECP_WB);
  __syncthreads();
  int* el = el_[wv];
  long* eo = eo_[wv];
  // lane q = (atom k, electron block eb): its entries, atom-major, electrons ascending
  const int kq = (lane < nq) ? lane / neb : 0, ebq = (lane < nq) ? lane % neb : 0;
  const int naipq = T.naip[kq];
  const int nup_here = S.nup - ebq * 64;  // bits below it are spin-up electrons
  const unsigned long long upmask = (nup_here >= 64) ? ~0ull : ((nup_here <= 0) ? 0ull : ((1ull << nup_here) - 1ull));
  const int cu = __popcll(m & upmask), cd = __popcll(m & ~upmask);
  int nent, tu, td;
  const int ebase = wave_excl_scan(cu + cd, lane, nent);
  // first slot of this lane's (atom, electron block) word in the two spins' lists: walker-major — after the walker's earlier
  // words; atom-major — segment (atom, walker), after the atom's earlier electron blocks
  const int urel = wave_excl_scan(cu * naipq, lane, tu), drel = wave_excl_scan(cd * naipq, lane, td);
  const size_t SS = (size_t)B.nseg * W + 1;
  long ubase, dbase;
  if (B.nseg > 1) {
    const int first = kq * neb;  // the atom's first word
    ubase = B.off[(size_t)kq * W + w] + (urel - __shfl(urel, first, 64));
    dbase = B.off[SS + (size_t)kq * W + w] + (drel - __shfl(drel, first, 64));
  } else {
    ubase = B.off[w] + urel;
    dbase = B.off[SS + w] + drel;
  }
  for (int base = 0; base < nent; base += 64) {
    {
      unsigned long long mm = m;
      int idx = ebase, iu = 0, id = 0;
      while (mm) {
        const int b = __ffsll((long long)mm) - 1;
        mm &= mm - 1;
        const bool up = (upmask >> b) & 1ull;
        const long off = up ? ubase + (long)naipq * iu : dbase + (long)naipq * id;
        if (up) ++iu; else ++id;
        if (idx >= base && idx < base + 64) { el[idx - base] = (kq << 16) | (ebq * 64 + b); eo[idx - base] = off; }
        ++idx;
      }
    }
    // el / eo belong to this wave alone and the trip counts differ between the waves of the block: no block barrier in here.
    // A wave's LDS operations complete in program order; the fence keeps the compiler from moving the reads above the writes.
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
    const int cnt = (nent - base < 64) ? nent - base : 64;
#pragma unroll 1
    for (int r0 = 0; r0 < cnt; r0 += 4) {
      const bool act = r0 + grp < cnt;
      const int ei = act ? r0 + grp : cnt - 1;
      const int ke = el[ei], k = ke >> 16, e = ke & 0xffff, s = e >= S.nup;
      const long off = eo[ei];
      // lane l16 < 9 of the group fetches one element of the entry's rotation now; the group reads them by shuffle after the Jastrow sum
      const double Rl = B.rot[((size_t)e * S.necp + k) * 9 + (l16 < 9 ? l16 : 0)];
      const double x0 = xw[3 * e], y0 = xw[3 * e + 1], z0 = xw[3 * e + 2];
      double U0 = 0.0;
      if (UE) U0 = B.ue[(size_t)e * W + w];
      else if (B.has_j2) {  // the 16 lanes of the group split the partners and the ions; fixed-order tree over the group
        double g_[3], lp_, ee_, ei_;
        jas_eval_lane<0, PBC>(S, xw, 1L, 0L, e, x0, y0, z0, 1, l16, 16, U0, g_, lp_, ee_, ei_);
#pragma unroll
        for (int o = 1; o < 16; o <<= 1) U0 += __shfl_xor(U0, o, 64);
      }
      double dx = x0 - T.atom[4 * k], dy = y0 - T.atom[4 * k + 1], dz = z0 - T.atom[4 * k + 2];
      if (PBC) min_image(S, dx, dy, dz);
      const double r = sqrt(dx * dx + dy * dy + dz * dz);
      double v[PQA_MAXCHAN], prob;
      int nch;
      ecp_radial_t(T, k, r, B.threshold, v, nch, prob);
      const int naip = T.naip[k], qoff = T.qoff[k];
      double Rm[9];
#pragma unroll
      for (int q = 0; q < 9; ++q) Rm[q] = __shfl(Rl, (lane & 48) + q, 64);
      if (act && wlive)
        for (int ip = l16; ip < naip; ip += 16) {  // (6 or 12 points: one trip; the 18- to 50-point rules: up to four)
          const double* qd = B.quad + 3 * (qoff + ip);
          const double vx = Rm[0] * qd[0] + Rm[1] * qd[1] + Rm[2] * qd[2];
          const double vy = Rm[3] * qd[0] + Rm[4] * qd[1] + Rm[5] * qd[2];
          const double vz = Rm[6] * qd[0] + Rm[7] * qd[1] + Rm[8] * qd[2];
          const double rix = r * vx, riy = r * vy, riz = r * vz;  // eval_ecp.py:242
          const double cosv = (dx * rix + dy * riy + dz * riz) / (r * sqrt(rix * rix + riy * riy + riz * riz));
          double wsum = 0.0;
          for (int c = 0; c < nch - 1; ++c) wsum += (v[c] / prob) * (2 * c + 1) * legendre_l(c, cosv);
          const long slot = off + ip;
          B.pts[s][3 * slot] = (x0 - dx) + rix;  // eval_ecp.py:110
          B.pts[s][3 * slot + 1] = (y0 - dy) + riy;
          B.pts[s][3 * slot + 2] = (z0 - dz) + riz;
          B.wgt[s][slot] = wsum * B.quadw[qoff + ip];
          B.pte[s][slot] = e;
          B.ptw[s][slot] = (int)w;
          B.u0[s][slot] = U0;
        }
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
  }
}

// ---------------------------------------------------------------- thread per auxiliary point on the lane-per-walker state
// k_ecp_point (pqa_energy.hpp) with its two latency chains taken apart.  There every one of the 8 slices of the determinant dot
// waited for its own loads, and the loop over the 63 partners loaded one partner's coordinates, waited, and then waited again for
// every coefficient (indexed by the partner's spin: a vector load inside the innermost loop): ~270 us per wave for ~20 us of
// arithmetic.  Here the inverse column and the orbital row are fetched 16 slots at a time before any product, and the Jastrow
// exponent comes from jas_eval_lane on the SoA coordinate planes.  Same operations in the same order as k_ecp_point.
// CX: complex determinants (inverse planes [row][2 k + re|im][W], orbital rows [re block | im block]); the imaginary parts of the
// contributions go to contrib[npts + p].  (k_ecp_accum walked a walker's points one after the other on one wave, 16 of its lanes
// busy in the 16-electron dots: 0.99 ms per evaluation of the twisted 32-electron cell at 8 192 walkers, 11 % of its step.)
template <bool PBC, bool CX = false>
static __global__ __launch_bounds__(256) void k_ecp_point_lw(SysDev S, LwState L, EcpBuf B, int s, int has_slater, int has_jastrow,
                                                      const double* __restrict__ mo, long npts, long W, double* __restrict__ contrib,
                                                      const double* __restrict__ mo1 = nullptr, long npts1 = 0, double* __restrict__ contrib1 = nullptr) {
  if (blockIdx.y) { s = 1; mo = mo1; npts = npts1; contrib = contrib1; }  // (small shards: both spin channels in one launch, grid rows 0 / 1)
  const long p = (long)blockIdx.x * 256 + threadIdx.x;
  if (p >= npts || (B.ptot[s] && p >= *B.ptot[s])) return;
  const int e = B.pte[s][p];
  const long w = B.ptw[s][p];
  const int n = s ? S.ndn : S.nup, i = e - s * S.nup, nmo = S.nmo[s];
  double ratio = 1.0, ratio_im = 0.0;
  if (has_slater) {
    const double* row = mo + (size_t)p * nmo;
    if (CX) {
      const double* Ti = L.Tt[s] + (size_t)i * 2 * n * W + w;
      const int* occ = S.det_occ[s];
      const int nh = nmo / 2;
      double rr = 0.0, ri = 0.0;
      const bool lines = S.occ_ident[s] && ((n | nh) & 7) == 0;  // whole 64-byte lines of the point's row (see k_kinetic_lw)
      for (int k0 = 0; k0 < n; k0 += 8) {
        double tr[8], ti[8], a[8], b[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
          const int k = (k0 + u < n) ? k0 + u : n - 1;
          tr[u] = Ti[(size_t)(2 * k) * W]; ti[u] = Ti[(size_t)(2 * k + 1) * W];
        }
        if (lines) {
          const double4 al = *reinterpret_cast<const double4*>(row + k0), ah = *reinterpret_cast<const double4*>(row + k0 + 4);
          const double4 bl = *reinterpret_cast<const double4*>(row + nh + k0), bh = *reinterpret_cast<const double4*>(row + nh + k0 + 4);
          a[0] = al.x; a[1] = al.y; a[2] = al.z; a[3] = al.w; a[4] = ah.x; a[5] = ah.y; a[6] = ah.z; a[7] = ah.w;
          b[0] = bl.x; b[1] = bl.y; b[2] = bl.z; b[3] = bl.w; b[4] = bh.x; b[5] = bh.y; b[6] = bh.z; b[7] = bh.w;
        } else {
#pragma unroll
          for (int u = 0; u < 8; ++u) {
            const int o = occ[(k0 + u < n) ? k0 + u : n - 1];
            a[u] = row[o]; b[u] = row[nh + o];
          }
        }
#pragma unroll
        for (int u = 0; u < 8; ++u)
          if (k0 + u < n) { rr += a[u] * tr[u] - b[u] * ti[u]; ri += a[u] * ti[u] + b[u] * tr[u]; }
      }
      ratio = rr; ratio_im = ri;
    } else {
    const double* Ti = L.Tt[s] + (size_t)i * n * W + w;
    double r = 0.0;
    if (S.occ_ident[s] && (n % 16) == 0 && (nmo % 4) == 0) {
      for (int k0 = 0; k0 < n; k0 += 16) {
        double t[16];
        double4 q[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) q[u] = *reinterpret_cast<const double4*>(row + k0 + 4 * u);
#pragma unroll
        for (int u = 0; u < 16; ++u) t[u] = Ti[(size_t)(k0 + u) * W];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          r += q[u].x * t[4 * u]; r += q[u].y * t[4 * u + 1]; r += q[u].z * t[4 * u + 2]; r += q[u].w * t[4 * u + 3];
        }
      }
    } else {
      const int* occ = S.det_occ[s];
      for (int k = 0; k < n; ++k) r += row[occ[k]] * Ti[(size_t)k * W];
    }
    ratio = r;
    }
  }
  if (has_jastrow) {
    double U, g[3], lp, ee, ei;
    // every point of this launch (grid row) belongs to an electron of spin s: the merged Pade records are picked by scalar selects although the
    // electron differs between lanes (jas_eval_lane_m ELANE: a third fewer instructions per pair than the function-by-function route)
    if (S.jq_on) jas_eval_lane_m<0, PBC, true, true>(S, L.xt, W, w, e, B.pts[s][3 * p], B.pts[s][3 * p + 1], B.pts[s][3 * p + 2], 1, 0, 1, U, g, lp, ee, ei, -1, true);
    else jas_eval_lane<0, PBC>(S, L.xt, W, w, e, B.pts[s][3 * p], B.pts[s][3 * p + 1], B.pts[s][3 * p + 2], 1, 0, 1, U, g, lp, ee, ei);
    const double ej = exp(U - B.u0[s][p]);  // U_e(new) - U_e(old); the old-position sum comes from the fill pass
    ratio *= ej; ratio_im *= ej;
  }
  contrib[p] = ratio * B.wgt[s][p];
  if (CX) contrib[npts + p] = ratio_im * B.wgt[s][p];
}
