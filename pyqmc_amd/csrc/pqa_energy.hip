// pyqmc_amd C ABI implementation (host side): local energy on the device (kinetic, Coulomb / Ewald, ECP), pqa_energy, pqa_set_ewald, pqa_get_wrap.
// See include/pyqmc_amd.h for the contract and pqa_internal.hpp for what the units share.
#include "pqa_internal.hpp"

int energy_dev(pqa_handle* h, double threshold, const double* rot, const double* unif, uint64_t seed, uint32_t step,
               bool soa_current, bool aos_T_needed, bool assemble) {
  const long W = h->W;
  struct JsxOnce { pqa_handle* h; ~JsxOnce() { h->jsx_current = false; } } jsx_once{h};  // (valid for the evaluation that follows the sweep only)
  if (!soa_current) h->jsx_current = false;
  bool soa_T = false;
  struct Side { pqa_handle* h = nullptr; hipStream_t main = nullptr; ~Side() { if (h) h->stream = main; } } side;  // (launches go to h->stream: restored on every exit)
  bool side_join = false;
  TRY(ensure(h, h->b_kc, (size_t)4 * W * sizeof(double)));
  TRY(ensure(h, h->b_en, (size_t)(h->cplx ? 7 : 6) * W * sizeof(double)));
  // will the ECP passes below leave their point totals on the device (no host synchronisation inside this evaluation)?
  const bool will_defer = h->necp > 0 && h->ecpb_on == 0 && h->ecp_defer != 0 && !h->S.pbc && h->ecp_hint_valid && (h->ecp_evals % 16) != 0 &&
                          (W * (long)h->N * h->necp * std::max(h->S.ecp_naip_max, 1)) * (long)(64 + 8 * std::max(h->nmo[0], h->nmo[1])) <= (long)256 << 20;
  auto side_begin = [&]() -> int {  // the kinetic / Coulomb pass goes to the side stream; the caller's stream carries on with the ECP passes
    if (!h->en_stream) {
      HIPCHK(hipStreamCreateWithFlags(&h->en_stream, hipStreamNonBlocking));
      for (hipEvent_t& e : h->en_ev) HIPCHK(hipEventCreateWithFlags(&e, hipEventDisableTiming));
    }
    HIPCHK(hipEventRecord(h->en_ev[0], h->stream));
    HIPCHK(hipStreamWaitEvent(h->en_stream, h->en_ev[0], 0));
    side.main = h->stream; side.h = h;
    h->stream = h->en_stream;
    return 0;
  };
  if (soa_current) {
    // shards whose ECP passes read their point totals back: the host waits for the counting passes while the kinetic pass runs beside them on
    // the side stream, and has the rest of the evaluation enqueued before the device gets there (the read-back was an idle gap of ~50 us)
    if (h->necp > 0 && h->ecpb_on == 0 && h->en_overlap && (W <= 16384 || h->en_overlap > 1) && !will_defer) TRY(side_begin());  // (PQA_EN_OVERLAP=2: at every size, A/B)
    const dim3 gk((unsigned)((((W + 63) / 64 + 7) / 8) * 8 * ((h->N + PQA_KIN_EB - 1) / PQA_KIN_EB))), bk(64, PQA_KIN_EB);  // see k_kinetic_lw
    const bool kin_quad = W >= 16384 && (W & 3) == 0;  // quad-cooperative row reads (k_kinetic_lw): large shards
    if (h->cplx) {
      if (h->S.pbc) hipLaunchKernelGGL((k_kinetic_lw<true, true>), gk, bk, 0, h->stream, h->S, lw_state(h), (int)h->has_jastrow, W, (double*)h->b_kpart.p);
      else hipLaunchKernelGGL((k_kinetic_lw<false, true>), gk, bk, 0, h->stream, h->S, lw_state(h), (int)h->has_jastrow, W, (double*)h->b_kpart.p);
    } else if (h->S.pbc)
      { if (kin_quad) hipLaunchKernelGGL((k_kinetic_lw<true, false, true>), gk, bk, 0, h->stream, h->S, lw_state(h), (int)h->has_jastrow, W, (double*)h->b_kpart.p);
        else hipLaunchKernelGGL(k_kinetic_lw<true>, gk, bk, 0, h->stream, h->S, lw_state(h), (int)h->has_jastrow, W, (double*)h->b_kpart.p); }
    else if (kin_quad)
      hipLaunchKernelGGL((k_kinetic_lw<false, false, true>), gk, bk, 0, h->stream, h->S, lw_state(h), (int)h->has_jastrow, W, (double*)h->b_kpart.p);
    else
      hipLaunchKernelGGL(k_kinetic_lw<false>, gk, bk, 0, h->stream, h->S, lw_state(h), (int)h->has_jastrow, W, (double*)h->b_kpart.p);
    hipLaunchKernelGGL((k_kinetic_reduce<>), dim3((unsigned)((W + 255) / 256)), dim3(256), 0, h->stream, (const double*)h->b_kpart.p,
                       h->N, W, (double*)h->b_kc.p);
    TRY(check_launch(h, "k_kinetic_lw"));
    // the ECP kernels read walker-major coordinates; the inverse only when the wave-per-walker accumulation runs (or the
    // caller works on the walker-major state next: the DMC step's T-moves) — the thread-per-point kernel takes the planes
    soa_T = !aos_T_needed && (!h->cplx || h->ecp_point_lw) && h->ndet == 1 && !h->has_j3 && h->ecp_wave == 0 && h->ecp_soa_t;
    if (h->necp > 0) {
      hipStream_t cur = h->stream;
      if (side.h) h->stream = side.main;  // (the ECP passes' input: in their stream)
      if (soa_T) {
        if (!h->jsx_current) { transpose(h, (const double*)h->b_xt.p, h->js.x, (long)h->N * 3, W); TRY(check_launch(h, "k_transpose")); }  // (else: k_sweep_r8 wrote them)
      }
      else TRY(lw_to_aos(h, false));
      if (side.h) {  // the Ewald pass (side stream) reads the walker-major coordinates too
        HIPCHK(hipEventRecord(h->en_ev[2], h->stream));
        HIPCHK(hipStreamWaitEvent(cur, h->en_ev[2], 0));
      }
      h->stream = cur;
    }
  } else {
    // small shards with ECPs: the kinetic / Coulomb pass on a side stream beside the ECP passes — both are a few thousand latency-bound waves
    // (50-determinant water molecule, 2 048 walkers: 125 us and 245 us one after the other)
    if (h->necp > 0 && h->en_overlap && W * h->N <= 32768) TRY(side_begin());
    {  // four waves per walker while the launch is too small to fill the chip with one
      const bool kc4 = h->ecp_acc_waves == 4 || (h->ecp_acc_waves == 0 && W * h->N <= 32768);
      const size_t st_ = (size_t)(h->cplx ? 2 : 1) * lds_det(h, 5);
      const int str_ = (int)(st_ / sizeof(double));
#define PQA_KC(CXF, NV) hipLaunchKernelGGL((k_kinetic_coulomb<CXF, NV>), dim3((unsigned)W), dim3(64 * NV), NV * st_, h->stream, h->S, h->st, h->js, \
                                           (int)h->has_slater, (int)h->has_jastrow, W, (double*)h->b_kc.p, str_)
      if (h->cplx) { if (kc4) PQA_KC(true, 4); else PQA_KC(true, 1); }
      else { if (kc4) PQA_KC(false, 4); else PQA_KC(false, 1); }
#undef PQA_KC
    }
    TRY(check_launch(h, "k_kinetic_coulomb"));
  }
  if (h->S.pbc) {
    if (!h->ew_set) FAIL("periodic Coulomb energy needs the Ewald tables (pqa_set_ewald)");
    const bool soa = soa_current && h->necp == 0;  // with ECPs the coordinates were just transposed back
    const double* x = soa ? (const double*)h->b_xt.p : h->js.x;
    const size_t lds_ew = ((size_t)h->N * 3 + (h->ew.gn ? (size_t)h->N * 3 * (h->ew.nmax + 1) * 2 : 0)) * sizeof(double);
    hipLaunchKernelGGL((k_ewald<>), dim3((unsigned)W), dim3(PQA_EWALD_T), lds_ew, h->stream, h->S, h->ew, x,
                       soa ? 1L : (long)h->N * 3, soa ? 3 * W : 3L, soa ? W : 1L, W, (double*)h->b_kc.p);
    TRY(check_launch(h, "k_ewald"));
  }
  if (side.h) {  // the ECP passes go on in the caller's stream; k_energy_assemble waits for the side stream
    HIPCHK(hipEventRecord(h->en_ev[1], h->stream));
    h->stream = side.main;
    side.h = nullptr;
    side_join = true;
  }
  const double* d_ecp = nullptr;
  h->last_ecp_points = 0;
  if (h->necp > 0) {
    const size_t nrot = (size_t)h->N * h->necp;
    TRY(ensure(h, h->b_rot, nrot * 9 * sizeof(double)));
    if (rot) TRY(copy_in(h, h->b_rot.p, rot, nrot * 9 * sizeof(double)));
    else {
      hipLaunchKernelGGL((k_gen_rot<>), dim3((unsigned)((nrot + 63) / 64)), dim3(64), 0, h->stream, (int)nrot, seed, step, (double*)h->b_rot.p);
      TRY(check_launch(h, "k_gen_rot"));
    }
    EcpBuf B{};
    B.rot = (const double*)h->b_rot.p;
    const bool batched = h->ecpb_on != 0;
    if (unif && !batched) {
      TRY(ensure(h, h->b_eunif, nrot * W * sizeof(double)));
      TRY(copy_in(h, h->b_eunif.p, unif, nrot * W * sizeof(double)));
      B.unif = (const double*)h->b_eunif.p;
    }
    B.quad = h->d_quad; B.quadw = h->d_quadw; B.seed = seed; B.step = step; B.threshold = threshold;
    TRY(ensure(h, h->b_elocal, W * sizeof(double)));
    // second-generation list passes (pqa_ecp.hpp): tables in LDS, four walkers per block, ATOM-major point lists
    const size_t tab_b = ecp_tab_bytes(h->necp, h->ecp_nchan, h->ecp_nterm);
    const bool ecp_t = h->ecp_lds && h->necp <= 64 && (long)h->necp * ((h->N + 63) / 64) <= 64 && tab_b <= 32768;
    // (atom-major lists where the orbital kernel gains from them: periodic cells, whose per-lane image walks then have similar
    // lengths within a tile — 2x2x2 diamond VMC +3 % at 32768 walkers; open systems gain nothing and pay a longer scan and sum)
    const long nseg = (!batched && ecp_t && h->ecp_atom_major && h->S.pbc) ? h->necp : 1, nsw = nseg * W;
    B.nseg = (int)nseg;
    TRY(ensure(h, h->b_ecnt, 2 * nsw * sizeof(int)));
    TRY(ensure(h, h->b_eoff, 2 * (nsw + 1) * sizeof(long)));
    TRY(ensure(h, h->b_ecp, (h->cplx ? 2 : 1) * W * sizeof(double)));
    TRY(ensure(h, h->b_epass, (size_t)W * h->necp * ((h->N + 63) / 64) * sizeof(unsigned long long)));
    B.local = (double*)h->b_elocal.p; B.cnt = (int*)h->b_ecnt.p; B.off = (long*)h->b_eoff.p;
    B.passbits = (unsigned long long*)h->b_epass.p;
    B.has_j2 = h->has_j2 ? 1 : 0;
    B.ue = (soa_current && h->has_j2) ? (const double*)h->b_kpart.p + (size_t)4 * h->N * W : nullptr;  // k_kinetic_lw left U_e there
    const dim3 g_t((unsigned)((W + PQA_ECP_WB - 1) / PQA_ECP_WB)), b_t(64 * PQA_ECP_WB);
    long tot[2];
    bool defer = false;  // totals stay on the device (below)
    const long* cnt_dev[2] = {nullptr, nullptr};
    EcpbArgs A{};
    if (batched) {  // every (walker, electron) has ecpb_nsel slots: no counting pass, no scan (pqa_ecpb.hpp)
      A.naip = h->d_ecpb_naip; A.qoff = h->d_ecpb_qoff; A.pstart = h->d_ecpb_pstart;
      A.npoints = h->ecpb_npoints; A.nsd = h->ecpb_nsd; A.nsr = h->ecpb_nsr; A.nsel = h->ecpb_nsel;
      A.e0 = 0; A.e1 = h->N; A.tau = 0.0;
      if (unif && A.nsel < A.npoints) {  // here: the (N, W, nselect_random) selection uniforms
        const size_t nb = (size_t)h->N * W * A.nsr * sizeof(double);
        TRY(ensure(h, h->b_eunif, nb));
        TRY(copy_in(h, h->b_eunif.p, unif, nb));
        A.selu = (const double*)h->b_eunif.p;
      }
      tot[0] = W * (long)h->nup * A.nsel; tot[1] = W * (long)h->ndn * A.nsel;
    } else {
    if (ecp_t) {
      if (h->S.pbc) hipLaunchKernelGGL(k_ecp_count_t<true>, g_t, b_t, tab_b, h->stream, h->S, h->js, B, h->ecp_nchan, h->ecp_nterm, W);
      else hipLaunchKernelGGL(k_ecp_count_t<false>, g_t, b_t, tab_b, h->stream, h->S, h->js, B, h->ecp_nchan, h->ecp_nterm, W);
    } else
    if (h->S.pbc) hipLaunchKernelGGL(k_ecp_count<true>, dim3((unsigned)W), dim3(64), 0, h->stream, h->S, h->js, B, W);
    else hipLaunchKernelGGL(k_ecp_count<false>, dim3((unsigned)W), dim3(64), 0, h->stream, h->S, h->js, B, W);
    // device-wide scans of the two spins' point counts (the one-block k_scan2 took 0.26 ms at 65536 walkers; small shards: one launch)
    // (both totals land in the handle's pinned, device-visible host words: one stream synchronisation is the whole read-back)
    if (!h->pin_tot) HIPCHK(hipHostMalloc((void**)&h->pin_tot, 4 * sizeof(long), hipHostMallocMapped));
    if (nsw <= 16384) hipLaunchKernelGGL((k_scan_small2<>), dim3(1), dim3(1024), 0, h->stream, (const int*)B.cnt, B.off, (const int*)B.cnt + nsw, B.off + (nsw + 1), nsw, h->pin_tot);
    else {
      TRY(ensure(h, h->b_tmmarks, 4 * sizeof(long)));
      TRY(scan_ints(h, (const int*)B.cnt, B.off, nsw, nsw, (long*)h->b_tmmarks.p));
      TRY(scan_ints(h, (const int*)B.cnt + nsw, B.off + (nsw + 1), nsw, nsw, (long*)h->b_tmmarks.p + 2));
    }
    TRY(check_launch(h, "k_ecp_count/k_scan2"));
    // Small shards: no device -> host round trip for the totals (two copies and the host's wake-up: ~80 us of a 0.3 - 0.8 ms step).  The
    // lists are sized for every point of every ECP atom, the orbital launch and the per-point pass cover that bound and their blocks beyond
    // the device-side count leave at once (PointAddr::count, EcpBuf::ptot; k_ecp_accum / k_ecp_sum read the list offsets from the device
    // anyway); the kernel choice follows the totals of the last evaluation that did read them (every 16th does).
    {
      const long pmax = (long)h->necp * std::max(h->S.ecp_naip_max, 1);  // quadrature points of one electron over all ECP atoms, at most
      const long ub0 = W * (long)h->nup * pmax, ub1 = W * (long)h->ndn * pmax;
      defer = h->ecp_defer != 0 && !h->S.pbc && h->ecp_hint_valid && (h->ecp_evals % 16) != 0 &&
              (ub0 + ub1) * (long)(64 + 8 * std::max(h->nmo[0], h->nmo[1])) <= (long)256 << 20;
      if (defer) { tot[0] = ub0; tot[1] = ub1; }
      else {
        if (nsw <= 16384) {
          HIPCHK(hipStreamSynchronize(h->stream));
          tot[0] = h->pin_tot[0]; tot[1] = h->pin_tot[1];
        } else {  // (marks[1], marks[3] of the two scans: one copy)
          long mk[4];
          TRY(copy_out(h, mk, h->b_tmmarks.p, 4 * sizeof(long)));
          tot[0] = mk[1]; tot[1] = mk[3];
        }
        h->ecp_hint[0] = tot[0]; h->ecp_hint[1] = tot[1]; h->ecp_hint_valid = true;
      }
      ++h->ecp_evals;
      cnt_dev[0] = B.off + nsw; cnt_dev[1] = B.off + (nsw + 1) + nsw;
      B.ptot[0] = defer ? cnt_dev[0] : nullptr; B.ptot[1] = defer ? cnt_dev[1] : nullptr;
    }
    }
    h->last_ecp_points = defer ? -1 : tot[0] + tot[1];
    h->last_ecp_dev[0] = defer ? cnt_dev[0] : nullptr; h->last_ecp_dev[1] = defer ? cnt_dev[1] : nullptr;
    for (int s = 0; s < 2; ++s) {
      const size_t n = (size_t)std::max<long>(tot[s], 1);
      TRY(ensure(h, h->b_epts[s], n * 3 * sizeof(double)));
      TRY(ensure(h, h->b_ewgt[s], n * sizeof(double)));
      TRY(ensure(h, h->b_epte[s], n * sizeof(int)));
      TRY(ensure(h, h->b_eptw[s], n * sizeof(int)));
      TRY(ensure(h, h->b_econ[s], (h->cplx ? 2 : 1) * n * sizeof(double)));
      TRY(ensure(h, h->b_eu0[s], n * sizeof(double)));
      B.ptw[s] = (int*)h->b_eptw[s].p;
      B.u0[s] = (double*)h->b_eu0[s].p;
      TRY(ensure(h, h->b_emo[s], n * std::max(h->nmo[s], 1) * sizeof(double)));
      B.pts[s] = (double*)h->b_epts[s].p; B.wgt[s] = (double*)h->b_ewgt[s].p; B.pte[s] = (int*)h->b_epte[s].p;
    }
    if (batched) {  // (also with no point at all: the local channels and the list offsets come from this launch)
      const dim3 g_b((unsigned)((W + PQA_ECPB_WB - 1) / PQA_ECPB_WB)), b_b(64 * PQA_ECPB_WB);
      if (h->S.pbc) hipLaunchKernelGGL(k_ecpb_fill<true>, g_b, b_b, 0, h->stream, h->S, h->js, B, A, W);
      else hipLaunchKernelGGL(k_ecpb_fill<false>, g_b, b_b, 0, h->stream, h->S, h->js, B, A, W);
      TRY(check_launch(h, "k_ecpb_fill"));
    }
    if (side_join && soa_current) HIPCHK(hipStreamWaitEvent(h->stream, h->en_ev[1], 0));  // (the fill pass reads U_e from the kinetic pass's output)
    if (tot[0] + tot[1] > 0) {
      if (batched) {
      } else
      if (ecp_t) {
        if (B.ue) {
          if (h->S.pbc) hipLaunchKernelGGL((k_ecp_fill_t<true, true>), g_t, b_t, tab_b, h->stream, h->S, h->js, B, h->ecp_nchan, h->ecp_nterm, W);
          else hipLaunchKernelGGL((k_ecp_fill_t<false, true>), g_t, b_t, tab_b, h->stream, h->S, h->js, B, h->ecp_nchan, h->ecp_nterm, W);
        } else {
          if (h->S.pbc) hipLaunchKernelGGL((k_ecp_fill_t<true, false>), g_t, b_t, tab_b, h->stream, h->S, h->js, B, h->ecp_nchan, h->ecp_nterm, W);
          else hipLaunchKernelGGL((k_ecp_fill_t<false, false>), g_t, b_t, tab_b, h->stream, h->S, h->js, B, h->ecp_nchan, h->ecp_nterm, W);
        }
      } else
      if (h->S.pbc) hipLaunchKernelGGL(k_ecp_fill<true>, dim3((unsigned)W), dim3(64), 0, h->stream, h->S, h->js, B, W);
      else hipLaunchKernelGGL(k_ecp_fill<false>, dim3((unsigned)W), dim3(64), 0, h->stream, h->S, h->js, B, W);
      TRY(check_launch(h, "k_ecp_fill"));
      if (h->has_slater)
        for (int s = 0; s < 2; ++s) {
          PointAddr pa = plain_points(B.pts[s], tot[s]);
          if (defer) { pa.count = cnt_dev[s]; h->orb_p_hint = std::max<long>(h->ecp_hint[s], 1); }
          const int rc = launch_orb(h, s, pa, tot[s], 1, (double*)h->b_emo[s].p);
          h->orb_p_hint = 0;
          if (rc != 0) return rc;
        }
    }
    // wave-per-walker accumulation (complex determinants, several determinants, three-body factor): four waves share a walker's
    // points while the launch is too small to fill the chip with one (measured after the three-body / determinant-pass fixes: C4
    // +7 % at 2048 walkers, even at 4096, -7 % at 8192; the 32-electron twisted cell +1 % at 1024, -1.5 % at 2048, -6 % at 8192)
    const bool acc4 = h->ecp_acc_waves == 4 || (h->ecp_acc_waves == 0 && W * h->N <= 32768);
#define PQA_ECP_ACC(PB, CXF, SC) do { const size_t st_ = (size_t)(SC) * lds_det(h, 1); const int str_ = (int)(st_ / sizeof(double)); \
      if (acc4) hipLaunchKernelGGL((k_ecp_accum<PB, CXF, 4>), dim3((unsigned)W), dim3(256), 4 * st_, h->stream, h->S, h->st, h->js, B, (int)h->has_slater, \
                                   (int)h->has_jastrow, (const double*)h->b_emo[0].p, (const double*)h->b_emo[1].p, W, (double*)h->b_ecp.p, str_); \
      else hipLaunchKernelGGL((k_ecp_accum<PB, CXF, 1>), dim3((unsigned)W), dim3(64), st_, h->stream, h->S, h->st, h->js, B, (int)h->has_slater, \
                              (int)h->has_jastrow, (const double*)h->b_emo[0].p, (const double*)h->b_emo[1].p, W, (double*)h->b_ecp.p, str_); } while (0)
    const bool cx_points = h->cplx && soa_current && h->ndet == 1 && !h->has_j3 && h->ecp_wave == 0 && h->ecp_point_lw;  // thread per point on the complex planes
    if (h->cplx && !cx_points) {  // complex determinants: wave-per-walker accumulation in complex arithmetic
      if (h->S.pbc) PQA_ECP_ACC(true, true, 2); else PQA_ECP_ACC(false, true, 2);
    } else
    if (h->ndet == 1 && !h->has_j3 && h->ecp_wave == 0) {  // thread per point, then an ordered per-walker sum
      // small shards: the two spin channels of the per-point pass in one launch (two launches of ~11 us for a few thousand points each)
      const bool both = (cx_points || (soa_current && !h->cplx && h->ecp_point_lw)) && tot[0] > 0 && tot[1] > 0 &&
                        std::max(h->ecp_hint[0], h->ecp_hint[1]) <= 262144 && h->ecp_hint_valid;
      if (both) {
        const dim3 g2((unsigned)((std::max(tot[0], tot[1]) + 255) / 256), 2);
#define PQA_PT2(PB, CXF) hipLaunchKernelGGL((k_ecp_point_lw<PB, CXF>), g2, dim3(256), 0, h->stream, h->S, lw_state(h), B, 0, (int)h->has_slater, (int)h->has_jastrow, \
                                            (const double*)h->b_emo[0].p, tot[0], W, (double*)h->b_econ[0].p, (const double*)h->b_emo[1].p, tot[1], (double*)h->b_econ[1].p)
        if (cx_points) { if (h->S.pbc) PQA_PT2(true, true); else PQA_PT2(false, true); }
        else { if (h->S.pbc) PQA_PT2(true, false); else PQA_PT2(false, false); }
#undef PQA_PT2
      } else
      for (int s = 0; s < 2; ++s) {
        if (tot[s] <= 0) continue;
        const dim3 g((unsigned)((tot[s] + 255) / 256));
        const long n_s = s ? h->ndn : h->nup;
        const double* Tb = soa_T ? (const double*)h->b_Tt[s].p : (const double*)h->st.T[s];
        const long sw = soa_T ? 1 : n_s * n_s, si = soa_T ? n_s * W : n_s, sk = soa_T ? W : 1;
        // (the planes are the live state whenever this evaluation follows a lane-per-walker sweep, also where the walker-major copy
        // was refreshed for the caller's next step — the DMC loop's T-moves)
        if (cx_points) {
          if (h->S.pbc)
            hipLaunchKernelGGL((k_ecp_point_lw<true, true>), g, dim3(256), 0, h->stream, h->S, lw_state(h), B, s, (int)h->has_slater,
                               (int)h->has_jastrow, (const double*)h->b_emo[s].p, tot[s], W, (double*)h->b_econ[s].p);
          else
            hipLaunchKernelGGL((k_ecp_point_lw<false, true>), g, dim3(256), 0, h->stream, h->S, lw_state(h), B, s, (int)h->has_slater,
                               (int)h->has_jastrow, (const double*)h->b_emo[s].p, tot[s], W, (double*)h->b_econ[s].p);
        } else
        if (soa_current && !h->cplx && h->ecp_point_lw) {
          if (h->S.pbc)
            hipLaunchKernelGGL(k_ecp_point_lw<true>, g, dim3(256), 0, h->stream, h->S, lw_state(h), B, s, (int)h->has_slater,
                               (int)h->has_jastrow, (const double*)h->b_emo[s].p, tot[s], W, (double*)h->b_econ[s].p);
          else
            hipLaunchKernelGGL(k_ecp_point_lw<false>, g, dim3(256), 0, h->stream, h->S, lw_state(h), B, s, (int)h->has_slater,
                               (int)h->has_jastrow, (const double*)h->b_emo[s].p, tot[s], W, (double*)h->b_econ[s].p);
        } else
        if (h->S.pbc)
          hipLaunchKernelGGL(k_ecp_point<true>, g, dim3(256), 0, h->stream, h->S, h->st, h->js, B, s, (int)h->has_slater,
                             (int)h->has_jastrow, (const double*)h->b_emo[s].p, tot[s], (double*)h->b_econ[s].p, Tb, sw, si, sk);
        else
          hipLaunchKernelGGL(k_ecp_point<false>, g, dim3(256), 0, h->stream, h->S, h->st, h->js, B, s, (int)h->has_slater,
                             (int)h->has_jastrow, (const double*)h->b_emo[s].p, tot[s], (double*)h->b_econ[s].p, Tb, sw, si, sk);
      }
      hipLaunchKernelGGL((k_ecp_sum<>), dim3((unsigned)((W + 255) / 256)), dim3(256), 0, h->stream, B, (const double*)h->b_econ[0].p,
                         (const double*)h->b_econ[1].p, W, (double*)h->b_ecp.p, cx_points ? std::max<long>(tot[0], 1) : 0L,
                         cx_points ? std::max<long>(tot[1], 1) : 0L);
    } else {
      if (h->S.pbc) PQA_ECP_ACC(true, false, 1); else PQA_ECP_ACC(false, false, 1);
    }
#undef PQA_ECP_ACC
    TRY(check_launch(h, "k_ecp_accum"));
    d_ecp = (const double*)h->b_ecp.p;
  }
  if (side_join) HIPCHK(hipStreamWaitEvent(h->stream, h->en_ev[1], 0));
  h->en_d_ecp = d_ecp;
  if (!assemble) return 0;  // (the caller finishes: k_energy_finish)
  hipLaunchKernelGGL((k_energy_assemble<>), dim3((unsigned)((W + 255) / 256)), dim3(256), 0, h->stream, (const double*)h->b_kc.p, d_ecp,
                     h->ii_energy, W, (double*)h->b_en.p, (int)h->cplx);
  return check_launch(h, "k_energy_assemble");
}

// ---------------------------------------------------------------- batched ECP integrator (jax_ecp.py)
extern "C" int pqa_set_ecp_batched(pqa_handle_t* h, int32_t enable, const int32_t* naip, int32_t nsd, int32_t nsr) {
  HIPCHK(hipSetDevice(h->device));
  if (!enable) { h->ecpb_on = 0; return 0; }
  if (h->necp == 0) { h->ecpb_on = 0; return 0; }  // nothing to integrate
  if (h->necp > 64) FAIL("the batched ECP integrator holds one ECP atom per lane: at most 64 ECP atoms");
  if (!naip || nsd < 0 || nsr < 0) FAIL("bad arguments");
  std::vector<int> na((size_t)h->necp), qo((size_t)h->necp), ps((size_t)h->necp);
  int np = 0;
  for (int k = 0; k < h->necp; ++k) {
    static const int offs[][2] = {{6, 0}, {12, 6}, {18, 18}, {26, 36}, {32, 62}, {50, 94}};
    int off = naip[k] == 0 ? 0 : -1;
    for (auto& o : offs) if (o[0] == naip[k]) off = o[1];
    if (off < 0) FAIL("naip must be 0 or one of 6, 12, 18, 26, 32, 50 for every atom (eval_ecp.py:266-267)");
    if (naip[k] > 0 && h->ecp_nch[k] < 2) FAIL("quadrature points on an atom without a non-local channel");
    na[k] = naip[k]; qo[k] = off; ps[k] = np; np += naip[k];
  }
  HIPCHK(hipStreamSynchronize(h->stream));
  if (!h->d_ecpb_naip) {
    TRY(upload_table(h, na.data(), na.size(), &h->d_ecpb_naip));
    TRY(upload_table(h, qo.data(), qo.size(), &h->d_ecpb_qoff));
    TRY(upload_table(h, ps.data(), ps.size(), &h->d_ecpb_pstart));
  } else {
    HIPCHK(hipMemcpy(h->d_ecpb_naip, na.data(), na.size() * sizeof(int), hipMemcpyHostToDevice));
    HIPCHK(hipMemcpy(h->d_ecpb_qoff, qo.data(), qo.size() * sizeof(int), hipMemcpyHostToDevice));
    HIPCHK(hipMemcpy(h->d_ecpb_pstart, ps.data(), ps.size() * sizeof(int), hipMemcpyHostToDevice));
  }
  h->ecpb_npoints = np; h->ecpb_nsd = nsd; h->ecpb_nsr = nsr;
  h->ecpb_nsel = (nsd + nsr >= np) ? np : nsd + nsr;  // jax_ecp.py:236-237: nothing is dropped
  if (h->ecpb_nsel < np && nsd + nsr > PQA_ECPB_MAXSEL) FAIL("more than 256 selected points per electron");
  h->ecpb_on = 1;
  return 0;
}

extern "C" int pqa_ecp_batched_nselected(pqa_handle_t* h) { return h->ecpb_on ? h->ecpb_nsel : 0; }

extern "C" int pqa_ecp_batched_moves(pqa_handle_t* h, int e, double tau, const double* rot, const double* unif, uint64_t seed,
                                     double* weight, double* pos) {
  TRY(sync_aos(h));
  HIPCHK(hipSetDevice(h->device));
  if (h->W == 0) FAIL("state not initialised (call pqa_wf_recompute)");
  if (!h->ecpb_on) FAIL("pqa_set_ecp_batched first");
  if (e < 0 || e >= h->N || !(tau > 0.0) || !rot) FAIL("bad arguments");
  const long W = h->W;
  const int nsel = h->ecpb_nsel;
  if (nsel == 0) return 0;
  // the kernel indexes the rotations by (electron, atom): place this electron's necp matrices at its row
  const size_t nrot = (size_t)h->N * h->necp;
  TRY(ensure(h, h->b_rot, nrot * 9 * sizeof(double)));
  TRY(copy_in(h, (double*)h->b_rot.p + (size_t)e * h->necp * 9, rot, (size_t)h->necp * 9 * sizeof(double)));
  EcpBuf B{};
  B.rot = (const double*)h->b_rot.p; B.quad = h->d_quad; B.quadw = h->d_quadw; B.seed = seed; B.step = 0x7d0u;
  EcpbArgs A{};
  A.naip = h->d_ecpb_naip; A.qoff = h->d_ecpb_qoff; A.pstart = h->d_ecpb_pstart;
  A.npoints = h->ecpb_npoints; A.nsd = h->ecpb_nsd; A.nsr = h->ecpb_nsr; A.nsel = nsel;
  A.e0 = e; A.e1 = e + 1; A.tau = tau;
  if (unif && nsel < A.npoints) {
    const size_t nb = (size_t)h->N * W * A.nsr * sizeof(double);  // (indexed by (electron, walker, draw) like the energy pass)
    TRY(ensure(h, h->b_eunif, nb));
    TRY(copy_in(h, (double*)h->b_eunif.p + (size_t)e * W * A.nsr, unif, (size_t)W * A.nsr * sizeof(double)));
    A.selu = (const double*)h->b_eunif.p;
  }
  TRY(ensure(h, h->b_epts[0], (size_t)W * nsel * 3 * sizeof(double)));
  TRY(ensure(h, h->b_ewgt[0], (size_t)W * nsel * sizeof(double)));
  A.out_pos = (double*)h->b_epts[0].p; A.out_w = (double*)h->b_ewgt[0].p;
  const dim3 g_b((unsigned)((W + PQA_ECPB_WB - 1) / PQA_ECPB_WB)), b_b(64 * PQA_ECPB_WB);
  if (h->S.pbc) hipLaunchKernelGGL(k_ecpb_fill<true>, g_b, b_b, 0, h->stream, h->S, h->js, B, A, W);
  else hipLaunchKernelGGL(k_ecpb_fill<false>, g_b, b_b, 0, h->stream, h->S, h->js, B, A, W);
  TRY(check_launch(h, "k_ecpb_fill"));
  TRY(copy_in(h, weight, A.out_w, (size_t)W * nsel * sizeof(double)));
  return copy_out(h, pos, A.out_pos, (size_t)W * nsel * 3 * sizeof(double));
}

extern "C" int pqa_set_ewald(pqa_handle_t* h, double alpha, int32_t ng, const double* gpoints, const double* gweight,
                             const double* ion_cos, const double* ion_sin, double ee_const, double ei_const, double ii,
                             const int32_t* gidx, const double* recip) {
  HIPCHK(hipSetDevice(h->device));
  if (!h->S.pbc) FAIL("Ewald tables on an open-boundary handle");
  if (ng < 0 || !(alpha > 0.0)) FAIL("bad Ewald parameters");
  HIPCHK(hipStreamSynchronize(h->stream));
  double* d;
  TRY(upload_table(h, gpoints, (size_t)ng * 3, &d)); h->ew.g = d;
  TRY(upload_table(h, gweight, (size_t)ng, &d)); h->ew.gweight = d;
  TRY(upload_table(h, ion_cos, (size_t)ng, &d)); h->ew.ion_cos = d;
  TRY(upload_table(h, ion_sin, (size_t)ng, &d)); h->ew.ion_sin = d;
  h->ew.ng = ng; h->ew.alpha = alpha; h->ew.ee_const = ee_const; h->ew.ei_const = ei_const;
  h->ew.gn = nullptr; h->ew.nmax = 0;
  if (gidx && recip && ng > 0) {
    std::vector<int> gi((size_t)ng * 3);
    HIPCHK(hipMemcpy(gi.data(), gidx, gi.size() * sizeof(int), hipMemcpyDefault));
    int nmax = 0;
    for (int v : gi) nmax = std::max(nmax, std::abs(v));
    const size_t lds = ((size_t)h->N * 3 + (size_t)h->N * 3 * (nmax + 1) * 2) * sizeof(double);
    if (lds <= 64 * 1024) {  // otherwise stay with the direct sincos form
      int* dgi;
      TRY(upload_table(h, gi.data(), gi.size(), &dgi));
      h->ew.gn = dgi; h->ew.nmax = nmax;
      HIPCHK(hipMemcpy(h->ew.recip, recip, 9 * sizeof(double), hipMemcpyDefault));
    }
  }
  h->ii_energy = ii;
  h->ew_set = true;
  return 0;
}

extern "C" int pqa_get_wrap(pqa_handle_t* h, int32_t* wrap) {
  HIPCHK(hipSetDevice(h->device));
  if (!h->S.pbc) FAIL("open-boundary handle has no wrap counters");
  if (h->wrap_W != h->W || h->W == 0) FAIL("no fused sweep has run on the resident walkers");
  return copy_out(h, wrap, h->b_wrap.p, (size_t)h->W * h->N * 3 * sizeof(int));
}

extern "C" int pqa_energy(pqa_handle_t* h, double threshold, const double* rot, const double* unif, uint64_t seed, double* out) {
  TRY(sync_aos(h));
  HIPCHK(hipSetDevice(h->device));
  if (h->W == 0) FAIL("state not initialised (call pqa_wf_recompute)");
  h->saved_valid = false;
  TRY(energy_dev(h, threshold, rot, unif, seed, 0u));
  return copy_out(h, out, h->b_en.p, (size_t)(h->cplx ? 7 : 6) * h->W * sizeof(double));
}

