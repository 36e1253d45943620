// The batched ECP integrator: every ECP atom's quadrature points of one electron in one table, the largest terms evaluated
// deterministically and the rest sampled (EnergyAccumulator(use_old_ecp=False) -> pyqmc/observables/jax_ecp.py).
//
// Reference semantics, per electron e (jax_ecp.py):
//   evaluate_vl :160-222      every atom with naip > 0 contributes its naip points: v_l(r) (2l+1) P_l(cos) w_i per channel, the
//                             point's "probability" = sum over the atom's non-local channels of v_l(r)^2; the local channels of all
//                             atoms are summed (no range cut-off, no stochastic mask)
//   downselect_move_info :225-290   if nselect_deterministic + nselect_random < npoints: the nselect_deterministic points of largest
//                             probability are kept with weight 1 (numpy argsort ascending, last entries), the others' probabilities
//                             are normalised (uniform over all points where they vanish), nselect_random points are drawn from the
//                             cumulative sum with one uniform each and weighted by 1 / (nselect_random p)
//   ECPAccumulator.__call__ :72-105   ecp = sum_e [ local + sum_selected ratio * weight ]
//   ECPAccumulator.nonlocal_tmoves :110-135   weight = sum_l [P_l > 0] (exp(-tau v_l / P_l) - 1) P_l over the selected points
// Ties in the probabilities (the points of one atom share it) are ordered by index, as a stable sort orders them: among equal
// probabilities the larger point index is kept first.  (numpy's default argsort is not stable: where a tie straddles the cut the
// reference's own choice depends on the sort implementation; every such choice is a valid sample of the same estimator.)
//
// One wave per walker walks the electrons of [e0, e1); lane k holds ECP atom k (necp <= 64).  The selected points go to the same
// per-spin lists the semi-local integrator fills (EcpBuf: pts / wgt / pte / ptw / u0, nsel slots per electron, walker-major), so the
// orbital launch, the ratio kernels and the ordered per-walker sum are shared with it.
#pragma once
#include "pqa_energy.hpp"

#define PQA_STREAM_ECPSEL 9u

struct EcpbArgs {
  const int* naip;     // [necp] points of atom k (0: the atom has no non-local channel to integrate)
  const int* qoff;     // [necp] first row of its rule in EcpBuf::quad / quadw
  const int* pstart;   // [necp] index of its first point in the electron's table
  int npoints, nsd, nsr, nsel;  // table size, deterministic / random selections, slots per electron (npoints when nothing is dropped)
  const double* selu;  // [N][W][nsr] selection uniforms, or NULL -> Philox
  int e0, e1;
  double tau;          // > 0: T-move weights of electron e0 into out_pos / out_w (nothing goes to the EcpBuf lists)
  double* out_pos;     // [W][nsel][3]
  double* out_w;       // [W][nsel]
};

#define PQA_ECPB_WB 4
#define PQA_ECPB_MAXSEL 256  // slots of one electron held in LDS per pass

template <bool PBC>
static __global__ __launch_bounds__(64 * PQA_ECPB_WB) void k_ecpb_fill(SysDev S, JastrowState js, EcpBuf B, EcpbArgs A, long W) {
  __shared__ int sl_atom_[PQA_ECPB_WB][PQA_ECPB_MAXSEL];
  __shared__ int sl_ip_[PQA_ECPB_WB][PQA_ECPB_MAXSEL];
  __shared__ double sl_sc_[PQA_ECPB_WB][PQA_ECPB_MAXSEL];
  const int lane = threadIdx.x & 63, wv = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  const long w = (long)blockIdx.x * PQA_ECPB_WB + wv;
  if (w >= W) return;  // (no block barrier below: every LDS slice belongs to one wave)
  int* sl_atom = sl_atom_[wv];
  int* sl_ip = sl_ip_[wv];
  double* sl_sc = sl_sc_[wv];
  const double* xw = js.x + (size_t)w * S.nelec * 3;
  const bool atom = lane < S.necp;
  const int kk = atom ? lane : 0;
  const int ia = S.ecp_atom[kk];
  const double ax = S.atom_xyz[3 * ia], ay = S.atom_xyz[3 * ia + 1], az = S.atom_xyz[3 * ia + 2];
  const int naip = atom ? A.naip[kk] : 0, qoff = A.qoff[kk], pstart = A.pstart[kk];
  const bool down = A.nsel < A.npoints;
  double loc = 0.0;
  for (int e = A.e0; e < A.e1; ++e) {
    const int s = e >= S.nup, i_s = e - s * S.nup, n_s = s ? S.ndn : S.nup;
    const double x0 = xw[3 * e], y0 = xw[3 * e + 1], z0 = xw[3 * e + 2];
    double dx = x0 - ax, dy = y0 - ay, dz = z0 - az;
    if (PBC) min_image(S, dx, dy, dz);  // configs.dist.dist_i (jax_ecp.py:170-172)
    const double r = sqrt(dx * dx + dy * dy + dz * dz);
    double v[PQA_MAXCHAN] = {}, pr_;
    int nch = 1;
    if (atom) ecp_radial(S, kk, r, 0.0, v, nch, pr_);
    double vloc = atom ? v[nch - 1] : 0.0, prob = 0.0;
    for (int c = 0; c < nch - 1; ++c) prob += v[c] * v[c];  // jax_ecp.py:214
    if (naip == 0) prob = 0.0;
    loc += wave_sum(vloc);
    double U0 = 0.0;
    if (A.tau <= 0.0) {
      if (B.ue) U0 = B.ue[(size_t)e * W + w];
      else if (B.has_j2) {
        double g_[3], lp_;
        jas_eval<0, PBC>(S, xw, e, x0, y0, z0, U0, g_, lp_, 1);
      }
    }
    for (int base = 0; base < A.nsel; base += PQA_ECPB_MAXSEL) {  // (one pass unless nothing is dropped from a table of > 256 points)
      const int nhere = (A.nsel - base < PQA_ECPB_MAXSEL) ? A.nsel - base : PQA_ECPB_MAXSEL;
      if (!down) {
        for (int ip = 0; ip < naip; ++ip) {
          const int sl = pstart + ip - base;
          if (sl >= 0 && sl < nhere) { sl_atom[sl] = kk; sl_ip[sl] = ip; sl_sc[sl] = 1.0; }
        }
      } else {
        // points of atoms that come before this one in descending (probability, index) order
        int before = 0;
        for (int j = 0; j < S.necp; ++j) {
          const double pj = __shfl(prob, j, 64);
          const int nj = __shfl(naip, j, 64);
          if (j != kk && (pj > prob || (pj == prob && j > kk))) before += nj;
        }
        int taken = A.nsd - before;
        taken = taken < 0 ? 0 : (taken > naip ? naip : taken);
        for (int t = 0; t < taken; ++t) {  // ascending order of the reference's index list: the largest entry last
          const int sl = A.nsd - 1 - (before + t);
          sl_atom[sl] = kk; sl_ip[sl] = naip - 1 - t; sl_sc[sl] = 1.0;
        }
        const int live0 = naip - taken;
        const double norm = wave_sum((double)live0 * prob);
        const bool flat = !(norm > 0.0);  // jax_ecp.py:246-247: uniform over ALL points of the table
        const double pn = flat ? 1.0 / (double)(A.npoints - A.nsd) : prob / norm;
        const int live = flat ? naip : live0;
        const double mass = (double)live * pn;
        double pre = mass;  // inclusive scan over the atoms in index order
#pragma unroll
        for (int off = 1; off < 64; off <<= 1) {
          const double y = __shfl_up(pre, off, 64);
          if (lane >= off) pre += y;
        }
        pre -= mass;
        for (int j = 0; j < A.nsr; ++j) {
          double u;
          if (A.selu) u = A.selu[((size_t)e * W + w) * A.nsr + j];
          else {
            const Philox p = philox(B.seed, (uint32_t)w, (uint32_t)(e * A.nsr + j), PQA_STREAM_ECPSEL, B.step);
            u = u01(p.c[0], p.c[1]);
          }
          int c = 0;  // this atom's points whose cumulative probability lies below u (jax_ecp.py:254-257)
          for (int ip = 0; ip < naip; ++ip) {
            const double cdf = pre + (double)((ip < live ? ip + 1 : live)) * pn;
            c += (u > cdf) ? 1 : 0;
          }
#pragma unroll
          for (int off = 32; off > 0; off >>= 1) c += __shfl_xor(c, off, 64);
          const int idx = c < A.npoints ? c : A.npoints - 1;
          if (naip > 0 && idx >= pstart && idx < pstart + naip) {
            const int ip = idx - pstart;
            const bool det = ip >= naip - taken;  // (only where the uniform fallback can land on a kept point) weight 1, jax_ecp.py:262-265
            const double ps = det ? 1.0 : (double)A.nsr * pn;
            sl_atom[A.nsd + j] = kk; sl_ip[A.nsd + j] = ip; sl_sc[A.nsd + j] = ps > 0.0 ? 1.0 / ps : 0.0;
          }
        }
      }
      __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
      __builtin_amdgcn_wave_barrier();
      for (int sl0 = 0; sl0 < nhere; sl0 += 64) {
        const int sl = sl0 + lane;
        const bool on = sl < nhere;
        const int k = on ? sl_atom[sl] : 0, ip = on ? sl_ip[sl] : 0;
        const double sc = on ? sl_sc[sl] : 0.0;
        // the atom's displacement, distance and channel values live in lane k
        const double kdx = __shfl(dx, k, 64), kdy = __shfl(dy, k, 64), kdz = __shfl(dz, k, 64), kr = __shfl(r, k, 64);
        const int knch = __shfl(nch, k, 64), kq = __shfl(qoff, k, 64);
        double kv[PQA_MAXCHAN];
#pragma unroll
        for (int c = 0; c < PQA_MAXCHAN; ++c) kv[c] = __shfl(v[c], k, 64);
        if (on) {
          const double* qd = B.quad + 3 * (kq + ip);
          const double* R = B.rot + ((size_t)e * S.necp + k) * 9;
          const double vx = R[0] * qd[0] + R[1] * qd[1] + R[2] * qd[2];
          const double vy = R[3] * qd[0] + R[4] * qd[1] + R[5] * qd[2];
          const double vz = R[6] * qd[0] + R[7] * qd[1] + R[8] * qd[2];
          const double rix = kr * vx, riy = kr * vy, riz = kr * vz;  // eval_ecp.py:242
          const double cosv = (kdx * rix + kdy * riy + kdz * riz) / (kr * sqrt(rix * rix + riy * riy + riz * riz));
          const double qw = B.quadw[kq + ip];
          const double px = (x0 - kdx) + rix, py = (y0 - kdy) + riy, pz = (z0 - kdz) + riz;  // jax_ecp.py:84-86
          double wsum = 0.0;
          if (A.tau > 0.0) {
            for (int c = 0; c < knch - 1; ++c) {
              const double pl = (2 * c + 1) * legendre_l(c, cosv) * qw;
              if (pl > 0.0) wsum += (exp(-A.tau * (kv[c] * sc)) - 1.0) * pl;  // jax_ecp.py:124-131
            }
            const size_t o = (size_t)w * A.nsel + base + sl;
            A.out_pos[3 * o] = px; A.out_pos[3 * o + 1] = py; A.out_pos[3 * o + 2] = pz;
            A.out_w[o] = wsum;
          } else {
            for (int c = 0; c < knch - 1; ++c) wsum += kv[c] * ((2 * c + 1) * legendre_l(c, cosv) * qw);
            const size_t slot = ((size_t)w * n_s + i_s) * A.nsel + base + sl;
            B.pts[s][3 * slot] = px; B.pts[s][3 * slot + 1] = py; B.pts[s][3 * slot + 2] = pz;
            B.wgt[s][slot] = wsum * sc;
            B.pte[s][slot] = e;
            B.ptw[s][slot] = (int)w;
            B.u0[s][slot] = U0;
          }
        }
      }
      __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
      __builtin_amdgcn_wave_barrier();
    }
  }
  if (lane == 0 && A.tau <= 0.0) {
    B.local[w] = loc;
    const size_t SS = (size_t)W + 1;  // (walker-major lists: nseg = 1)
    B.off[w] = (long)w * S.nup * A.nsel;
    B.off[SS + w] = (long)w * S.ndn * A.nsel;
    if (w == W - 1) { B.off[W] = (long)W * S.nup * A.nsel; B.off[SS + W] = (long)W * S.ndn * A.nsel; }
  }
}
