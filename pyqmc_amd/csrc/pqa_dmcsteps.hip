// pyqmc_amd C ABI implementation (host side): the fused DMC step loop (pqa_dmc_steps) and the T-move entry (pqa_tmoves).
// See include/pyqmc_amd.h for the contract and pqa_internal.hpp for what the units share.
#include "pqa_internal.hpp"
int scan_ints(pqa_handle* h, const int* c, long* o, long n, long Wm, long* marks) {
  const long nt = (n + 1023) / 1024;
  TRY(ensure(h, h->b_tmtile, (size_t)(nt + 1) * sizeof(long)));
  long* tile = (long*)h->b_tmtile.p;
  hipLaunchKernelGGL((k_scan_local<>), dim3((unsigned)nt), dim3(1024), 0, h->stream, c, o, n, tile);
  hipLaunchKernelGGL((k_scan_tiles<>), dim3(1), dim3(1024), 0, h->stream, tile, nt);
  hipLaunchKernelGGL((k_scan_add<>), dim3((unsigned)nt), dim3(1024), 0, h->stream, o, n, (const long*)tile, nt, Wm, marks);
  return check_launch(h, "k_scan_local/tiles/add");
}

// ---------------------------------------------------------------- fused DMC propagation
// nsteps steps of dmc_propagate (pyqmc/method/dmc.py:123-221) without leaving the device: T-moves, drift-diffusion with
// fixed-node rejection, local energy, weight update, weighted step averages.  Walker-per-wave kernels (the AoS state).
extern "C" int pqa_dmc_steps(pqa_handle_t* h, double tstep, int nsteps, double branchcut, double e_trial, double e_est, double threshold,
                             double* weights, const pqa_dmc_tapes_t* tp, uint64_t seed, double* step_avg, double* step_acc) {
  TRY(sync_aos(h));  // (the starting energy and the first T-moves read the walker-major state)
  HIPCHK(hipSetDevice(h->device));
  if (h->W == 0) FAIL("state not initialised (call pqa_wf_recompute)");
  if (nsteps <= 0) return 0;
  if (!weights || !step_avg || !step_acc) FAIL("pqa_dmc_steps: weights / step_avg / step_acc must not be NULL");
  const int navg = h->cplx ? 8 : 7;  // numbers per step in step_avg (complex: + the weighted mean of Im ecp = Im total)
  const long W = h->W;
  const int N = h->N, necp = h->necp, P = h->tm_P;
  const bool tmoves = necp > 0 && P > 0;
  if (tp && (!tp->gauss || !tp->unif)) FAIL("pqa_dmc_steps: a tape set needs gauss and unif");
  if (tp && h->ecpb_on) FAIL("pqa_dmc_steps: the tapes have the semi-local ECP integrator's layout (pqa_set_ecp_batched is on)");
  if (tp && necp > 0 && (!tp->ecp_rot || !tp->ecp_unif)) FAIL("pqa_dmc_steps: a tape set needs ecp_rot and ecp_unif for ECP systems");
  if (tp && tmoves && (!tp->tm_rot || !tp->tm_unif || !tp->tm_u1 || !tp->tm_u2)) FAIL("pqa_dmc_steps: a tape set needs the four T-move tapes");
  h->saved_valid = false;
  const int nmo_max = std::max(std::max(h->nmo[0], h->nmo[1]), 1);
  TRY(ensure(h, h->b_newpos, (size_t)W * 3 * sizeof(double)));
  TRY(ensure(h, h->b_aux, (size_t)W * 8 * sizeof(double)));
  TRY(ensure(h, h->b_accept, (size_t)W));
  TRY(ensure(h, h->b_acccnt, (size_t)nsteps * 2 * sizeof(int)));
  TRY(ensure(h, h->b_motmp, (size_t)W * 5 * nmo_max * sizeof(double)));
  TRY(ensure(h, h->b_accw, (size_t)W * sizeof(int)));
  TRY(ensure(h, h->b_dmcw, (size_t)W * sizeof(double)));
  TRY(ensure(h, h->b_dmcold, (size_t)2 * W * sizeof(double)));
  TRY(ensure(h, h->b_dmcr2, (size_t)2 * W * sizeof(double)));
  TRY(ensure(h, h->b_dmcout, (size_t)nsteps * navg * sizeof(double)));
  HIPCHK(hipMemsetAsync(h->b_acccnt.p, 0, (size_t)nsteps * 2 * sizeof(int), h->stream));
  HIPCHK(hipMemsetAsync(h->b_accw.p, 0, (size_t)W * sizeof(int), h->stream));
  HIPCHK(hipMemsetAsync(h->b_dmcr2.p, 0, (size_t)2 * W * sizeof(double), h->stream));
  TRY(copy_in(h, h->b_dmcw.p, weights, (size_t)W * sizeof(double)));
  if (h->S.pbc) {
    TRY(ensure(h, h->b_dwrap, (size_t)W * 3 * sizeof(int)));
    TRY(ensure(h, h->b_wrap, (size_t)W * N * 3 * sizeof(int)));
    HIPCHK(hipMemsetAsync(h->b_wrap.p, 0, (size_t)W * N * 3 * sizeof(int), h->stream));
    h->wrap_W = W;
  }
  if (tp) {
    TRY(ensure(h, h->b_gauss, (size_t)N * W * 3 * sizeof(double)));
    TRY(ensure(h, h->b_unif, (size_t)N * W * sizeof(double)));
  }
  const size_t nrot = (size_t)N * std::max(necp, 1);
  const int nkw = (std::max(necp, 1) + 63) / 64;
  if (tmoves) {
    const size_t NW = (size_t)N * W;
    TRY(ensure(h, h->b_tmcnt, NW * sizeof(int)));
    TRY(ensure(h, h->b_tmoff, (NW + 1) * sizeof(long)));
    TRY(ensure(h, h->b_tmpass, NW * nkw * sizeof(unsigned long long)));
    TRY(ensure(h, h->b_tmacc, NW * sizeof(int)));
    TRY(ensure(h, h->b_tmaoff, (NW + 1) * sizeof(long)));
    TRY(ensure(h, h->b_tmmarks, (size_t)(N + 1) * sizeof(long)));
    TRY(ensure(h, h->b_tmidx, NW * sizeof(int)));
    TRY(ensure(h, h->b_tmapos, NW * 3 * sizeof(double)));
    if (tp) TRY(ensure(h, h->b_tmu, (size_t)(2 + necp) * NW * sizeof(double)));
    TRY(ensure(h, h->b_rot, nrot * 9 * sizeof(double)));
  }
  const bool lw = h->lw_mode != 0 && h->has_slater && h->ndet == 1 && !h->has_j3 && (!h->cplx || std::max(h->nup, h->ndn) <= 32);
  LwCtx lc;
  TRY(lw_setup(h, lw, lc));
  const dim3 gw256((unsigned)((W + 255) / 256));
  double* eold = (double*)h->b_dmcold.p;
  double* r2 = (double*)h->b_dmcr2.p;
  std::vector<long> tm_accepted((size_t)nsteps, 0);
  // energy of the starting configuration (dmc.py:146-149) — or, after pqa_dmc_continue, the energies the previous call's last step
  // left in eold / v2old: what the reference itself carries from step to step (dmc.py:148-149, :199-200)
  const bool cont = h->dmc_continue;
  h->dmc_continue = false;
  if (cont && !(h->dmc_old_valid && h->dmc_old_W == W)) FAIL("pqa_dmc_continue: no previous pqa_dmc_steps call on the resident walkers' state");
  if (!cont) {
    TRY(energy_dev(h, threshold, (tp && necp) ? tp->ecp_rot : nullptr, (tp && necp) ? tp->ecp_unif : nullptr, seed, 0u, false));
    hipLaunchKernelGGL((k_dmc_keep<>), gw256, dim3(256), 0, h->stream, (const double*)h->b_en.p, eold, eold + W, W);
  }
  for (int step = 0; step < nsteps; ++step) {
    MoveBuf mb{};
    mb.newpos = (double*)h->b_newpos.p; mb.aux = (double*)h->b_aux.p; mb.accept = (uint8_t*)h->b_accept.p;
    mb.acc_w = (int*)h->b_accw.p; mb.seed = seed; mb.step = (uint32_t)step; mb.tstep = tstep;
    mb.dmc = 1; mb.r2_acc = r2; mb.r2_prop = r2 + W;
    if (h->S.pbc && !h->twist) { mb.dwrap = (int*)h->b_dwrap.p; mb.wrap = (int*)h->b_wrap.p; }  // twisted handles keep the walkers unfolded
    if (tmoves) {
      const size_t NW = (size_t)N * W;
      TmBuf B{};
      B.quad = h->d_quad; B.seed = seed; B.step = (uint32_t)step; B.tau = tstep; B.threshold = threshold; B.nofold = h->twist ? 1 : 0;
      B.cnt = (int*)h->b_tmcnt.p; B.off = (long*)h->b_tmoff.p; B.pass = (unsigned long long*)h->b_tmpass.p;
      long* d_marks = (long*)h->b_tmmarks.p;
      B.acc = (int*)h->b_tmacc.p; B.acc_off = (long*)h->b_tmaoff.p;
      B.acc_idx = (int*)h->b_tmidx.p; B.acc_pos = (double*)h->b_tmapos.p;
      if (tp) {
        double* u = (double*)h->b_tmu.p;
        TRY(copy_in(h, h->b_rot.p, tp->tm_rot + (size_t)step * nrot * 9, nrot * 9 * sizeof(double)));
        TRY(copy_in(h, u, tp->tm_u1 + (size_t)step * NW, NW * sizeof(double)));
        TRY(copy_in(h, u + NW, tp->tm_u2 + (size_t)step * NW, NW * sizeof(double)));
        TRY(copy_in(h, u + 2 * NW, tp->tm_unif + (size_t)step * NW * necp, NW * necp * sizeof(double)));
        B.u1 = u; B.u2 = u + NW; B.unif = u + 2 * NW;
      } else {
        hipLaunchKernelGGL((k_gen_rot<>), dim3((unsigned)((nrot + 63) / 64)), dim3(64), 0, h->stream, (int)nrot, seed ^ 0x9E3779B97F4A7C15ull,
                           (uint32_t)step, (double*)h->b_rot.p);
        TRY(check_launch(h, "k_gen_rot"));
      }
      B.rot = (const double*)h->b_rot.p;
      HIPCHK(hipMemsetAsync(B.acc, 0, NW * sizeof(int), h->stream));
      hipLaunchKernelGGL((k_tm_count<>), dim3(gw256.x, (unsigned)N), dim3(256), 0, h->stream, h->S, h->js, B, W);
      TRY(check_launch(h, "k_tm_count"));
      TRY(scan_ints(h, (const int*)B.cnt, B.off, (long)NW, W, d_marks));
      std::vector<long> eoff((size_t)N + 1);  // first candidate of every electron
      TRY(copy_out(h, eoff.data(), d_marks, eoff.size() * sizeof(long)));
      const long tot = eoff[N], tot_up = eoff[h->nup];
      if (tot > 0) {
        TRY(ensure(h, h->b_tpos, (size_t)tot * 3 * sizeof(double)));
        TRY(ensure(h, h->b_twgt, (size_t)tot * sizeof(double)));
        TRY(ensure(h, h->b_tmamp, (size_t)tot * 2 * sizeof(double)));
        TRY(ensure(h, h->b_tmptw, (size_t)tot * sizeof(int)));
        B.pts = (double*)h->b_tpos.p; B.wgt = (double*)h->b_twgt.p; B.amp = (double*)h->b_tmamp.p; B.rat = B.amp + tot;
        B.ptw = (int*)h->b_tmptw.p;
        hipLaunchKernelGGL((k_tm_fill<>), dim3((unsigned)W, (unsigned)N), dim3(64), 0, h->stream, h->S, h->js, B, W);
        TRY(check_launch(h, "k_tm_fill"));
        const long cnt_s[2] = {tot_up, tot - tot_up}, base_s[2] = {0, tot_up};
        if (h->has_slater)
          for (int s = 0; s < 2; ++s) {
            if (cnt_s[s] == 0) continue;
            TRY(ensure(h, h->b_emo[s], (size_t)cnt_s[s] * nmo_max * sizeof(double)));
            TRY(launch_orb(h, s, plain_points(B.pts + 3 * base_s[s], cnt_s[s]), cnt_s[s], 1, (double*)h->b_emo[s].p));
          }
        // ratios of all candidates against the state before the first T-move: one thread per candidate (k_tm_ratio)
        const bool pre = h->ndet == 1 && !h->has_j3 && !h->cplx && h->tm_pre;
        // U_e of every electron at its current position: from the second step of a call on, the energy evaluation that closed the
        // previous step left exactly that ([N][W], k_kinetic_lw) — the walkers have not moved since
        const double* d_uold = nullptr;
        if (pre && h->has_jastrow) {
          if (lw && step > 0 && h->has_j2 && !h->has_j3) d_uold = (const double*)h->b_kpart.p + (size_t)4 * NW;
          else {
            TRY(ensure(h, h->b_tmuold, (size_t)NW * sizeof(double)));
            hipLaunchKernelGGL((k_tm_uold<>), dim3(gw256.x, (unsigned)N), dim3(256), 0, h->stream, h->S, h->js, B, W, (double*)h->b_tmuold.p);
            d_uold = (const double*)h->b_tmuold.p;
          }
        }
        if (pre)
          for (int s = 0; s < 2; ++s) {
            if (cnt_s[s] == 0) continue;
            const dim3 g((unsigned)((cnt_s[s] + 255) / 256));
            hipLaunchKernelGGL((k_tm_ratio<>), g, dim3(256), 0, h->stream, h->S, h->st, h->js, B, s, (int)h->has_slater, (int)h->has_jastrow,
                               (const double*)h->b_emo[s].p, base_s[s], cnt_s[s], W, d_uold);
          }
        const size_t lds_tm = std::max(lds_sm(h), lds_det(h, 1));
        if (h->cplx) hipLaunchKernelGGL(k_tm_walker<true>, dim3((unsigned)W), dim3(64), 2 * lds_tm, h->stream, h->S, h->st, h->js, B, (int)h->has_slater,
                                        (int)h->has_jastrow, (const double*)h->b_emo[0].p, (const double*)h->b_emo[1].p, tot_up, W, 0);
        else hipLaunchKernelGGL(k_tm_walker<false>, dim3((unsigned)W), dim3(64), lds_tm, h->stream, h->S, h->st, h->js, B, (int)h->has_slater,
                                (int)h->has_jastrow, (const double*)h->b_emo[0].p, (const double*)h->b_emo[1].p, tot_up, W, pre ? 1 : 0);
        TRY(check_launch(h, "k_tm_walker"));
        TRY(scan_ints(h, (const int*)B.acc, B.acc_off, (long)NW, W, d_marks));
        hipLaunchKernelGGL((k_tm_gather<>), dim3((unsigned)((NW + 255) / 256)), dim3(256), 0, h->stream, B, (const double*)h->js.x, N, W);
        TRY(check_launch(h, "k_tm_gather"));
        TRY(copy_out(h, eoff.data(), d_marks, eoff.size() * sizeof(long)));
        const long nacc[2] = {eoff[N], eoff[h->nup]};
        tm_accepted[step] = nacc[0];
        if (h->has_slater) {  // gradient / Laplacian rows of the moved electrons, one launch per spin
          const long na_s[2] = {nacc[1], nacc[0] - nacc[1]}, a0_s[2] = {0, nacc[1]};
          for (int s = 0; s < 2; ++s) {
            if (na_s[s] == 0) continue;
            TRY(ensure(h, h->b_emo[s], (size_t)na_s[s] * 5 * nmo_max * sizeof(double)));
            TRY(launch_orb(h, s, plain_points(B.acc_pos + 3 * a0_s[s], na_s[s]), na_s[s], 5, (double*)h->b_emo[s].p));
            hipLaunchKernelGGL((k_tm_cache<>), dim3((unsigned)na_s[s]), dim3(64), 0, h->stream, h->S, h->st, (const int*)(B.acc_idx + a0_s[s]),
                               (const double*)h->b_emo[s].p, s, W, lw ? (double*)h->b_rc[s].p : (double*)nullptr, lw ? (const uint8_t*)h->b_sel[s].p : (const uint8_t*)nullptr);
          }
          TRY(check_launch(h, "k_tm_cache"));
        }
      }
    }
    if (tp) {
      TRY(copy_in(h, h->b_gauss.p, tp->gauss + (size_t)step * N * W * 3, (size_t)N * W * 3 * sizeof(double)));
      TRY(copy_in(h, h->b_unif.p, tp->unif + (size_t)step * N * W, (size_t)N * W * sizeof(double)));
      mb.gauss = (const double*)h->b_gauss.p; mb.unif = (const double*)h->b_unif.p;
    }
    if (lw && tmoves) TRY(lw_from_aos(h, false));  // the T-moves worked on the AoS coordinates and inverses
    TRY(sweep_electrons(h, mb, lw, lc));
    hipLaunchKernelGGL((k_sum_reset_int<>), dim3(1), dim3(1024), 0, h->stream, (int*)h->b_accw.p, W, (int*)h->b_acccnt.p + 2 * step);
    TRY(check_launch(h, "k_propose/k_accept (dmc)"));
    TRY(energy_dev(h, threshold, (tp && necp) ? tp->ecp_rot + (size_t)(step + 1) * nrot * 9 : nullptr,
                   (tp && necp) ? tp->ecp_unif + (size_t)(step + 1) * nrot * W : nullptr, seed, (uint32_t)(step + 1), lw));
    hipLaunchKernelGGL((k_dmc_weights<>), gw256, dim3(256), 0, h->stream, (const double*)h->b_en.p, eold, eold + W, r2, r2 + W,
                       (double*)h->b_dmcw.p, tstep, branchcut, e_trial, e_est, N, W);
    hipLaunchKernelGGL((k_dmc_averages<>), dim3(1), dim3(1024), 0, h->stream, (const double*)h->b_en.p, (const double*)h->b_dmcw.p, W,
                       (double*)h->b_dmcout.p + (size_t)step * navg, h->cplx ? 7 : 6);
    TRY(check_launch(h, "k_dmc_weights/k_dmc_averages"));
  }
  if (lw) TRY(lw_to_aos(h, true));
  h->jas_stale = h->has_j2;
  std::vector<int> cnt((size_t)nsteps * 2);
  TRY(copy_in(h, step_avg, h->b_dmcout.p, (size_t)nsteps * navg * sizeof(double)));
  TRY(copy_in(h, weights, h->b_dmcw.p, (size_t)W * sizeof(double)));
  TRY(copy_out(h, cnt.data(), h->b_acccnt.p, cnt.size() * sizeof(int)));
  for (int i = 0; i < nsteps; ++i) {
    step_acc[2 * i] = (double)cnt[2 * i] / ((double)W * N);
    step_acc[2 * i + 1] = (double)tm_accepted[i] / ((double)W * N);
  }
  h->dmc_old_valid = true; h->dmc_old_W = W;
  return 0;
}

extern "C" int pqa_dmc_continue(pqa_handle_t* h, int on) {
  h->dmc_continue = on != 0;
  return 0;
}

extern "C" int pqa_dmc_can_continue(pqa_handle_t* h) { return (h->dmc_old_valid && h->dmc_old_W == h->W) ? 1 : 0; }

extern "C" int pqa_tmove_npoints(pqa_handle_t* h) { return h->tm_P; }

extern "C" int pqa_tmoves(pqa_handle_t* h, int e, double tau, double threshold, const double* rot, const double* unif,
                          double* ratio, double* weight, double* pos) {
  TRY(sync_aos(h));
  HIPCHK(hipSetDevice(h->device));
  if (h->cplx && ratio) FAIL("pqa_tmoves: complex orbitals — pass ratio = NULL (positions and weights only) and take the ratios from pqa_wf_testvalue");
  if (h->W == 0) FAIL("state not initialised (call recompute)");
  if (e < 0 || e >= h->N) FAIL("electron index out of range");
  const long W = h->W;
  const int P = h->tm_P, s = e >= h->nup;
  if (P == 0) return 0;
  if (!rot || !unif) FAIL("pqa_tmoves needs the rotation and mask-uniform tapes");
  h->saved_valid = false;
  const size_t np = (size_t)W * P;
  TRY(ensure(h, h->b_rot, (size_t)h->necp * 9 * sizeof(double)));
  TRY(ensure(h, h->b_eunif, (size_t)h->necp * W * sizeof(double)));
  TRY(ensure(h, h->b_tpos, np * 3 * sizeof(double)));
  TRY(ensure(h, h->b_twgt, np * sizeof(double)));
  TRY(ensure(h, h->b_tlive, np));
  TRY(ensure(h, h->b_trat, np * sizeof(double)));
  TRY(copy_in(h, h->b_rot.p, rot, (size_t)h->necp * 9 * sizeof(double)));
  TRY(copy_in(h, h->b_eunif.p, unif, (size_t)h->necp * W * sizeof(double)));
  hipLaunchKernelGGL((k_tmove_points<>), dim3((unsigned)W), dim3(64), 0, h->stream, h->S, h->js, e, tau, threshold,
                     (const double*)h->b_rot.p, (const double*)h->b_eunif.p, (const double*)h->d_quad, (const int*)h->d_ptk,
                     (const int*)h->d_pti, P, W, (double*)h->b_tpos.p, (double*)h->b_twgt.p, (uint8_t*)h->b_tlive.p);
  TRY(check_launch(h, "k_tmove_points"));
  if (!ratio) {  // candidate positions and weights only (dead candidates carry weight 0)
    TRY(copy_in(h, weight, h->b_twgt.p, np * sizeof(double)));
    return copy_out(h, pos, h->b_tpos.p, np * 3 * sizeof(double));
  }
  if (h->has_slater) {
    TRY(ensure(h, h->b_motmp, np * std::max(h->nmo[s], 1) * sizeof(double)));
    TRY(launch_orb(h, s, plain_points((const double*)h->b_tpos.p, (long)np), (long)np, 1, (double*)h->b_motmp.p));
  }
  hipLaunchKernelGGL((k_tmove_ratio<>), dim3((unsigned)W), dim3(64), lds_det(h, 1), h->stream, h->S, h->st, h->js, e, (int)h->has_slater,
                     (int)h->has_jastrow, (const double*)h->b_motmp.p, (const double*)h->b_tpos.p, (const uint8_t*)h->b_tlive.p, P,
                     (double*)h->b_trat.p);
  TRY(check_launch(h, "k_tmove_ratio"));
  TRY(copy_in(h, ratio, h->b_trat.p, np * sizeof(double)));
  TRY(copy_in(h, weight, h->b_twgt.p, np * sizeof(double)));
  return copy_out(h, pos, h->b_tpos.p, np * 3 * sizeof(double));
}

// ---------------------------------------------------------------- measurement
