// pyqmc_amd C ABI implementation (host side): layout conversions, the fused VMC sweep (pqa_vmc_sweeps), the walker-tile sweep.
// See include/pyqmc_amd.h for the contract and pqa_internal.hpp for what the units share.
#include "pqa_sweep_launch.hpp"

void launch_step_real(pqa_handle* h, const LwState& L, const MoveBuf& mb, const StepArgs& a, int rowlen) {
  if (h->S.pbc) launch_step_lw<true, false>(h, L, mb, a, rowlen); else launch_step_lw<false, false>(h, L, mb, a, rowlen);
}
void launch_jas_pre(pqa_handle* h, const LwState& L, const MoveBuf& mb, const StepArgs& a, double* jnew, double* jold) {
  const dim3 grid((unsigned)((a.w1 - a.w0 + a.NW - 1) / a.NW)), block((unsigned)(a.NW * a.G));
  if (h->S.pbc) hipLaunchKernelGGL(k_jas_pre<true>, grid, block, 0, h->stream, h->S, L, mb, a, jnew, jold);
  else hipLaunchKernelGGL(k_jas_pre<false>, grid, block, 0, h->stream, h->S, L, mb, a, jnew, jold);
}
void launch_flush_real(pqa_handle* h, const LwState& L, int s, long W, long w0, long w1, int j_lo, int j_hi, int nq, int rowlen, int n_s) {
  launch_flush_lw<false>(h, L, s, W, w0, w1, j_lo, j_hi, nq, rowlen, n_s);
}

// energy of the resident walkers into device buffer b_en (6,W)
void transpose(pqa_handle* h, const double* in, double* out, long R, long C) {  // in [R][C] -> out [C][R]
  if (R <= 0 || C <= 0) return;
  hipLaunchKernelGGL((k_transpose<>), dim3((unsigned)((C + 31) / 32), (unsigned)((R + 31) / 32)), dim3(32, 8), 0, h->stream, in, out, R, C);
}

LwState lw_state(pqa_handle* h) {
  LwState L{};
  L.xt = (double*)h->b_xt.p;
  for (int s = 0; s < 2; ++s) {
    L.Tt[s] = (double*)h->b_Tt[s].p; L.rc[s] = (double*)h->b_rc[s].p; L.sel[s] = (uint8_t*)h->b_sel[s].p;
    L.dsign[s] = h->st.dsign[s]; L.dlog[s] = h->st.dlog[s];
  }
  L.auxt = (double*)h->b_auxt.p;
  return L;
}

// AoS (canonical, wave-per-walker kernels) -> SoA mirrors for the lane-per-walker kernels
int lw_from_aos(pqa_handle* h, bool with_cache) {
  const long W = h->W;
  const int nel[2] = {h->nup, h->ndn};
  TRY(ensure(h, h->b_xt, (size_t)W * h->N * 3 * sizeof(double)));
  TRY(ensure(h, h->b_auxt, (size_t)W * 8 * sizeof(double)));
  TRY(ensure(h, h->b_kpart, (size_t)W * h->N * 5 * sizeof(double)));
  transpose(h, h->js.x, (double*)h->b_xt.p, W, (long)h->N * 3);
  for (int s = 0; s < 2; ++s) {
    const size_t n = nel[s], cf = h->cplx ? 2 : 1;
    TRY(ensure(h, h->b_Tt[s], cf * W * n * n * sizeof(double)));
    const int row = 5 * h->nmo[s];
    TRY(ensure(h, h->b_rc[s], (size_t)2 * W * n * row * sizeof(double)));  // two slots per electron (pqa_lw.hpp)
    TRY(ensure(h, h->b_sel[s], (size_t)W * n));
    transpose(h, h->st.T[s], (double*)h->b_Tt[s].p, W, (long)(cf * n * n));
    if (n > 0 && row > 0) {
      // (without the cache: the caller knows the row cache and its selectors are live — the T-move phase of the DMC step)
      if (with_cache) hipLaunchKernelGGL((k_cache_to_rc<>), dim3((unsigned)W, (unsigned)n, (unsigned)((row + 255) / 256)), dim3(256), 0, h->stream,
                                         (const double*)h->st.cache[s], (double*)h->b_rc[s].p, (uint8_t*)h->b_sel[s].p, (int)n, row, W);
    }
  }
  return check_launch(h, "k_transpose");
}
// SoA -> AoS: coordinates and inverses (what the ECP kernels read); with_cache also the orbital cache
int lw_to_aos(pqa_handle* h, bool with_cache) {
  const long W = h->W;
  const int nel[2] = {h->nup, h->ndn};
  transpose(h, (const double*)h->b_xt.p, h->js.x, (long)h->N * 3, W);
  for (int s = 0; s < 2; ++s) {
    const long n = nel[s];
    transpose(h, (const double*)h->b_Tt[s].p, h->st.T[s], (h->cplx ? 2 : 1) * n * n, W);
    const int row = 5 * h->nmo[s];
    if (with_cache && n > 0 && row > 0)
      hipLaunchKernelGGL((k_cache_from_rc<>), dim3((unsigned)W, (unsigned)n, (unsigned)((row + 255) / 256)), dim3(256), 0, h->stream,
                         (const double*)h->b_rc[s].p, (const uint8_t*)h->b_sel[s].p, h->st.cache[s], (int)n, row, W);
  }
  return check_launch(h, "k_transpose");
}

// A fused call on the lane-per-walker kernels leaves the live state in the SoA planes and only marks the walker-major arrays
// stale: back-to-back fused calls (the blocks of a VMC run) then skip both layout conversions (~13 GB of traffic per call at
// 65536 walkers of the 64-electron system, 7 ms), and whoever needs the walker-major state — every protocol entry, the
// energy entry, branching — converts it back first.
int sync_aos(pqa_handle* h) {
  if (!h || !h->aos_stale) return 0;
  HIPCHK(hipSetDevice(h->device));
  h->aos_stale = false;
  return lw_to_aos(h, true);
}


// ---------------------------------------------------------------- one sweep over the electrons (shared by VMC and DMC)
// Geometry of the lane-per-walker kernels and, when `lw`, the SoA copy of the state and its scratch.
int lw_setup(pqa_handle* h, bool lw, LwCtx& c) {
  const long W = h->W;
  // thread groups per walker in k_step_lw.  Measured (tools/scratch/r3_step_abl*.sh, (H2O)8 step in ms at 4 / 8 / 16 groups):
  // 4096 walkers 6.38 / 5.15 / 4.67, 8192: 7.71 / 6.54 / 6.39, 16384: 10.7 / 9.8 / 11.1, 32768: 16.5 / 17.3 / 18.8, 65536: 30.8 / 32.7 / 37.3
  // 4 groups (a wave per group, k_step_lw's wide form) from 26624 walkers, 8 below, 16 below 16384 (re-measured at the end of round 4:
  // at 28672 walkers the wide form takes 13.1 ms per (H2O)8 step against 14.6 with 8 groups; at 24576 8 groups win 11.6 vs 12.2)
  c.Gm = 4;
  if ((long)4 * W < 1664L * 64) c.Gm = 8;
  if (c.Gm == 8 && (long)8 * W < 2048L * 64) c.Gm = 16;
  if (h->lw_gm > 0) c.Gm = std::min(h->lw_gm, 16);
  c.nmax = std::max(h->nup, h->ndn);
  // block size of the delayed Sherman-Morrison update: 4 from 16 electrons per spin (8 flushes at 32), 5 from 24 (7 flushes at 32:
  // 35.60 -> 35.32 ms per step of the 64-electron benchmark; 6 is slower again — the per-move commit touches KB rows)
  // small shards (every launch a latency chain, one block row per thread group): 8 — fewer flush launches and split step launches
  // ((H2O)8 at 4096 walkers: 4.08 ms per step with 5, 3.97 with 8, 3.94 with 11, 4.01 with 16; 8192: 5.97 / 5.80 / 5.90 / 6.00)
  const int kb = h->lw_kb < 0 ? (c.nmax >= 24 ? (W <= 8192 ? 8 : 5) : (c.nmax >= 16 ? 4 : 0)) : h->lw_kb;
  c.KB = (kb > 0) ? std::min(kb, std::max(c.nmax, 1)) : std::max(c.nmax, 1);  // KB = n: plain per-move update
  if (!lw) TRY(sync_aos(h));
  if (lw) {
    if (!h->aos_stale) TRY(lw_from_aos(h));  // (stale walker-major arrays: the planes ARE the state)
    const size_t cf = h->cplx ? 2 : 1;
    TRY(ensure(h, h->b_rbuf, cf * c.KB * std::max(c.nmax, 1) * W * sizeof(double)));
    TRY(ensure(h, h->b_vbuf, cf * c.KB * std::max(c.nmax, 1) * W * sizeof(double)));
    TRY(ensure(h, h->b_act, (size_t)c.KB * W));
  }
  return 0;
}
// ---- pipelined half-ensembles (round 4) ---------------------------------------------------------------------------------
// A move is k_orb (fp64 MFMA / VALU pipe bound, HBM idle) followed by k_step_lw (+ k_flush_lw; HBM / latency bound, the matrix
// pipe idle), and one walker's chain is strictly serial.  Walkers are independent, so the shard is cut into two half-ensembles
// A = [0, wm) and B = [wm, W) whose chains run side by side: while A's orbitals are evaluated, B's state is streamed, and
// vice versa.  Every kernel takes a walker window [w0, w1) on the SAME planes (stride W) and keys its Philox streams by the
// walker's index in the shard, so a trajectory does not depend on the cut: bit-identical to the single-stream sweep.
//   mode 1: one stream per half, free running
//   mode 2: one stream per half; the orbital launches of the two halves form one chain (orb A(e) -> orb B(e) -> orb A(e+1) ...),
//           so two orbital kernels never compete for the pipe and the other half's step kernel fills the rest of the chip
//   mode 3: one stream per kernel FAMILY (orbitals / state streaming) with the dependencies as events: the same schedule as
//           mode 2, and the two streams can be given disjoint CU masks (PQA_SPLIT_CUS = CUs of the orbital stream)
// PQA_SPLIT selects the mode (0: off), PQA_SPLIT_MIN the smallest shard that is cut.
struct HalfPipe {
  int mode = 0;
  long wm = 0;
  hipStream_t main = nullptr, s[2] = {nullptr, nullptr};
  hipEvent_t orb_done[2] = {nullptr, nullptr}, step_done[2] = {nullptr, nullptr};
};
static int pipe_event(pqa_handle* h, hipEvent_t* ev) {  // events from a ring: a wait keeps the record it saw when it was enqueued
  if (h->pipe_events.size() < 64) {
    hipEvent_t e;
    HIPCHK(hipEventCreateWithFlags(&e, hipEventDisableTiming));
    h->pipe_events.push_back(e);
    *ev = e;
    return 0;
  }
  *ev = h->pipe_events[h->pipe_next++ % h->pipe_events.size()];
  return 0;
}
static int pipe_streams(pqa_handle* h) {
  if (h->pipe_stream[0]) return 0;
  const int ncu = h->split_cus;
  for (int k = 0; k < 2; ++k) {
    if (h->split_mode == 3 && ncu > 0 && ncu < h->cu_count) {  // family streams on disjoint CU sets: orbitals on the first ncu mask bits
      std::vector<uint32_t> mask((h->cu_count + 31) / 32, 0u);
      for (int c = 0; c < h->cu_count; ++c)
        if ((c < ncu) == (k == 0)) mask[c / 32] |= 1u << (c % 32);
      HIPCHK(hipExtStreamCreateWithCUMask(&h->pipe_stream[k], (uint32_t)mask.size(), mask.data()));
    } else HIPCHK(hipStreamCreateWithFlags(&h->pipe_stream[k], hipStreamNonBlocking));
  }
  return 0;
}

// The next step's draws on the side stream, behind the sweep that has just been enqueued (the tape set of step + 1 was last read by the sweep
// of step - 1): they run beside this step's energy pass (k_tile_draws: 64 us of Philox + Box-Muller arithmetic at 65 536 walkers, next to
// memory- and latency-bound kernels) instead of in front of the next sweep.
int draws_ahead(pqa_handle* h, uint64_t seed, uint32_t next_step) {
  const long W = h->W;
  const size_t NW = (size_t)h->N * W;
  if (!h->draw_stream) {
    HIPCHK(hipStreamCreateWithFlags(&h->draw_stream, hipStreamNonBlocking));
    for (hipEvent_t& e : h->draw_ev) HIPCHK(hipEventCreateWithFlags(&e, hipEventDisableTiming));
  }
  DevBuf& bg = (next_step & 1) ? h->b_gauss_b : h->b_gauss;
  DevBuf& bu = (next_step & 1) ? h->b_unif_b : h->b_unif;
  TRY(ensure(h, bg, NW * 3 * sizeof(double)));
  TRY(ensure(h, bu, NW * sizeof(double)));
  HIPCHK(hipEventRecord(h->draw_ev[0], h->stream));
  HIPCHK(hipStreamWaitEvent(h->draw_stream, h->draw_ev[0], 0));
  hipLaunchKernelGGL((k_tile_draws<>), dim3((unsigned)((NW + 255) / 256)), dim3(256), 0, h->draw_stream, seed, next_step, h->N, W, (double*)bg.p, (double*)bu.p);
  HIPCHK(hipEventRecord(h->draw_ev[1], h->draw_stream));
  h->draw_ahead_valid = true; h->draw_ahead_step = next_step; h->draw_ahead_seed = seed; h->draw_ahead_W = W;
  return 0;
}

static bool N_ok(const pqa_handle* h) { return h->N <= 64 && h->natom <= 64 && std::max(h->nup, h->ndn) <= 64; }
static int sweep_electrons_fused(pqa_handle* h, const MoveBuf& mb_in, const LwCtx& lc) {
  const long W = h->W;
  MoveBuf mb = mb_in;
  // the resident sweep (pqa_res.hpp: one launch per sweep, state on chip) where the system is in its scope; it reads both tapes
  const bool tapes_ok = (mb.gauss != nullptr) == (mb.unif != nullptr);
  const bool r8 = tapes_ok && r8_eligible(h, W);  // (second generation, open-boundary real handles: pqa_res8.hpp)
  const bool res = r8 || (tapes_ok && res_eligible(h, W));
  if (!mb.gauss && !mb.unif && (res || W <= h->draws_max)) {
    // small shards: the sweep's normals and uniforms drawn ahead by one launch from the same Philox streams (k_tile_draws) — in
    // k_step_lw the lead group's Box-Muller pairs are ~600 dependent instructions of every move's chain with one wave per SIMD
    const size_t NW = (size_t)h->N * W;
    // two tape sets, step s in set s & 1: pqa_vmc_sweeps draws step s + 1 on a side stream while step s's energy pass runs (draws_ahead)
    DevBuf& bg = (mb.step & 1) ? h->b_gauss_b : h->b_gauss;
    DevBuf& bu = (mb.step & 1) ? h->b_unif_b : h->b_unif;
    if (h->draw_ahead_valid && h->draw_ahead_step == mb.step && h->draw_ahead_seed == mb.seed && h->draw_ahead_W == W) {
      HIPCHK(hipStreamWaitEvent(h->stream, h->draw_ev[1], 0));
    } else {
      TRY(ensure(h, bg, NW * 3 * sizeof(double)));
      TRY(ensure(h, bu, NW * sizeof(double)));
      hipLaunchKernelGGL((k_tile_draws<>), dim3((unsigned)((NW + 255) / 256)), dim3(256), 0, h->stream, mb.seed, mb.step, h->N, W, (double*)bg.p, (double*)bu.p);
    }
    h->draw_ahead_valid = false;
    h->draws_on_device = true;
    mb.gauss = (const double*)bg.p; mb.unif = (const double*)bu.p;
  }
  if (r8) return sweep_r8(h, mb);
  if (res) return sweep_res(h, mb);
  const int N = h->N, KB = lc.KB, nmax = lc.nmax;
  const LwState L = lw_state(h);
  const int cfi = h->cplx ? 2 : 1, rowlen = cfi * nmax;  // doubles per inverse row
  // thread groups per walker (lc.Gm: ~4 waves per SIMD's worth of threads) and walkers per block: 256 threads at most, so more
  // than 4 groups narrow the block to 32 or 16 walkers — which is also what spreads a small shard over the chip
  int G = std::min(lc.Gm, 16);
  int NW = (G <= 4) ? 64 : 256 / G;
  if (h->lw_nw > 0 && h->lw_nw * G <= 256) NW = h->lw_nw;
  // small shards (one 16-walker block per CU at most): 32 thread groups per walker — k_step_pre<.., 32>, 512 threads, two Jastrow
  // partners per thread: (H2O)8 step 3.93 -> 3.40 ms at 4096 walkers, 3.38 -> 2.88 at 1024; 64 groups (one partner per thread, 1024
  // threads) spill at the 128-register limit: 4.47 / 3.53 ms.  PQA_STEP_GW = 16 / 32 / 64 pins it.
  {
    const int gw = h->step_gw ? h->step_gw : 32;
    if ((gw == 32 || gw == 64) && W <= h->step_pre_max && N_ok(h) && step_pre_system_ok(h, rowlen) && KB <= gw / 4) { G = gw; NW = 16; }  // (a block row per quartet of groups)
  }

  HalfPipe P;
  P.main = h->stream;
  P.mode = (W >= h->split_min && !(h->split_mode == 1 && h->S.pbc)) ? h->split_mode : 0;  // (free-running halves would share the periodic pre-pass scratch)
  const int nh = P.mode ? 2 : 1;
  P.wm = P.mode ? std::min(W, ((W / 2 + 255) / 256) * 256) : W;
  if (P.mode) {
    TRY(pipe_streams(h));
    P.s[0] = h->pipe_stream[0]; P.s[1] = h->pipe_stream[1];
    hipEvent_t fork;
    TRY(pipe_event(h, &fork));
    HIPCHK(hipEventRecord(fork, P.main));
    HIPCHK(hipStreamWaitEvent(P.s[0], fork, 0));
    HIPCHK(hipStreamWaitEvent(P.s[1], fork, 0));
  }
  // Jastrow sums ahead of the orbitals (k_jas_pre): on for the k_step_lw launches of large shards with a Jastrow factor
  const bool jpre = h->has_jastrow && (h->jpre < 0 ? W >= h->jpre_min : h->jpre != 0) && !(NW < 64 && h->step_pre && W <= h->step_pre_max);
  double* jbuf = nullptr;
  hipEvent_t jas_done[2] = {nullptr, nullptr}, steps_done[2] = {nullptr, nullptr};
  if (jpre) {
    TRY(ensure(h, h->b_jpre, (size_t)2 * G * 4 * W * sizeof(double)));
    jbuf = (double*)h->b_jpre.p;
    for (int k = 0; k < nh; ++k)
      if (!h->jas_stream[k]) HIPCHK(hipStreamCreateWithFlags(&h->jas_stream[k], hipStreamNonBlocking));
  }
  struct Guard { pqa_handle* h; hipStream_t s; ~Guard() { h->stream = s; } } guard{h, P.main};  // launches go to h->stream: restored on every exit
  const long wlo[2] = {0, P.wm}, whi[2] = {P.wm, W};
  // stream of a launch: half hh, family 0 = orbitals, 1 = state streaming
  auto on = [&](int hh, int fam) { h->stream = !P.mode ? P.main : (P.mode == 3 ? P.s[fam] : P.s[hh]); };
  auto step = [&](int hh, int e_acc, int e_prop, bool use_pre = false) {
    StepArgs a{};
    a.e_acc = e_acc; a.e_prop = e_prop; a.has_jastrow = (int)h->has_jastrow; a.G = G; a.NW = NW; a.W = W; a.w0 = wlo[hh]; a.w1 = whi[hh];
    a.j_skip = e_prop > 0 ? e_prop - 1 : -1;
    if (use_pre) { a.jnew = e_acc >= 0 ? jbuf : nullptr; a.jold = e_prop >= 0 ? jbuf + (size_t)G * 4 * W : nullptr; }
    if (e_acc >= 0) {
      const int s = e_acc >= h->nup, n_s = s ? h->ndn : h->nup, i_s = e_acc - (s ? h->nup : 0);
      const int q = i_s % KB;
      a.j_lo = i_s - q; a.j_hi = std::min(a.j_lo + KB, n_s);
      a.Rbuf = (double*)h->b_rbuf.p + (size_t)q * cfi * n_s * W;
      a.Vbuf = (double*)h->b_vbuf.p + (size_t)q * cfi * n_s * W;
      a.act = (uint8_t*)h->b_act.p + (size_t)q * W;
    }
    if (h->cplx) launch_step_cx(h, L, mb, a, rowlen); else launch_step_real(h, L, mb, a, rowlen);
  };
  // what the state-streaming launches of half hh wait for / leave behind in modes 2 and 3
  auto after_steps = [&](int hh) -> int {
    if (P.mode == 3) { TRY(pipe_event(h, &P.step_done[hh])); HIPCHK(hipEventRecord(P.step_done[hh], h->stream)); }
    return 0;
  };
  for (int hh = 0; hh < nh; ++hh) {
    on(hh, 1);
    step(hh, -1, 0);
    TRY(after_steps(hh));
  }
  for (int e = 0; e < N; ++e) {
    const int s = e >= h->nup, n_s = s ? h->ndn : h->nup, i_s = e - (s ? h->nup : 0);
    const int q = i_s % KB, j_lo = i_s - q, j_hi = std::min(j_lo + KB, n_s);
    const bool block_done = (i_s == j_hi - 1);
    const bool need_flush = block_done && (j_hi - j_lo < n_s);
    // the next electron's inverse row is current after this move's commit unless it opens a new block of the SAME spin
    const bool fuse_next = (e + 1 < N) && !(need_flush && i_s + 1 < n_s);
    for (int hh = 0; hh < nh; ++hh) {
      const long w0 = wlo[hh], Wn = whi[hh] - wlo[hh];
      if (Wn <= 0) continue;
      // ---- Jastrow sums of this move on the side stream: they need the proposal (previous step launch) and nothing of the orbitals
      if (jpre) {
        hipStream_t st_steps = !P.mode ? P.main : (P.mode == 3 ? P.s[1] : P.s[hh]);
        TRY(pipe_event(h, &steps_done[hh]));
        HIPCHK(hipEventRecord(steps_done[hh], st_steps));
        HIPCHK(hipStreamWaitEvent(h->jas_stream[hh], steps_done[hh], 0));
        StepArgs ja{};
        ja.e_acc = e; ja.e_prop = (e + 1 < N) ? e + 1 : -1; ja.has_jastrow = 1; ja.G = G; ja.NW = NW; ja.W = W; ja.w0 = w0; ja.w1 = whi[hh];
        ja.j_skip = e;
        h->stream = h->jas_stream[hh];
        launch_jas_pre(h, L, mb, ja, jbuf, jbuf + (size_t)G * 4 * W);
        TRY(pipe_event(h, &jas_done[hh]));
        HIPCHK(hipEventRecord(jas_done[hh], h->stream));
      }
      // ---- orbitals at the proposals: the rows go straight into the slot of electron i_s the walker is not using (accepting
      // flips the selector)
      on(hh, 0);
      if (P.mode == 2 && P.orb_done[1 - hh]) HIPCHK(hipStreamWaitEvent(h->stream, P.orb_done[1 - hh], 0));
      if (P.mode == 3 && P.step_done[hh]) HIPCHK(hipStreamWaitEvent(h->stream, P.step_done[hh], 0));
      TRY(launch_orb(h, s, plain_points(mb.newpos + 3 * w0, Wn), Wn, 5, (double*)h->b_rc[s].p + ((size_t)i_s * 2 * W + w0) * 5 * h->nmo[s],
                     (const unsigned char*)h->b_sel[s].p + (size_t)i_s * W + w0, (long)W * 5 * h->nmo[s]));
      if (P.mode >= 2) { TRY(pipe_event(h, &P.orb_done[hh])); HIPCHK(hipEventRecord(P.orb_done[hh], h->stream)); }
      // ---- decide e, commit, propose e + 1
      on(hh, 1);
      if (P.mode == 3) HIPCHK(hipStreamWaitEvent(h->stream, P.orb_done[hh], 0));
      if (jpre) HIPCHK(hipStreamWaitEvent(h->stream, jas_done[hh], 0));
      hipEvent_t pe1 = nullptr;
      if (h->profile && hh == 0 && (e % (4 * (int)h->prof_stride)) == 1) {  // sparsely sampled full (decide + propose) launches: an event pair costs ~2 us of stream time
        if (h->prof3_used == h->prof3_events.size()) {
          hipEvent_t a, b;
          HIPCHK(hipEventCreate(&a));
          HIPCHK(hipEventCreate(&b));
          h->prof3_events.emplace_back(a, b);
        }
        if (fuse_next) {
          HIPCHK(hipEventRecord(h->prof3_events[h->prof3_used].first, h->stream));
          pe1 = h->prof3_events[h->prof3_used].second;
          ++h->prof3_used;
        }
      }
      step(hh, e, fuse_next ? e + 1 : -1, jpre);
      if (pe1) { HIPCHK(hipEventRecord(pe1, h->stream)); h->prof3_launches += 1; }
      if (need_flush) {  // block finished: bring every other row of this spin up to date
        const int nq = j_hi - j_lo;
        hipEvent_t ce1 = nullptr;
        if (h->profile && hh == 0 && ((j_lo / std::max(KB, 1)) % 4) == 0) {  // every 4th flush of a spin
          if (h->prof2_used == h->prof2_events.size()) {
            hipEvent_t a, b;
            HIPCHK(hipEventCreate(&a));
            HIPCHK(hipEventCreate(&b));
            h->prof2_events.emplace_back(a, b);
          }
          HIPCHK(hipEventRecord(h->prof2_events[h->prof2_used].first, h->stream));
          ce1 = h->prof2_events[h->prof2_used].second;
          ++h->prof2_used;
        }
        if (h->cplx) launch_flush_cx(h, L, s, W, w0, whi[hh], j_lo, j_hi, nq, rowlen, n_s);
        else launch_flush_real(h, L, s, W, w0, whi[hh], j_lo, j_hi, nq, rowlen, n_s);
        if (ce1) { HIPCHK(hipEventRecord(ce1, h->stream)); h->prof2_launches += 1; }
      }
      if (!fuse_next && e + 1 < N) step(hh, -1, e + 1, jpre);
      TRY(after_steps(hh));
    }
  }
  if (P.mode) {  // join: the caller's stream continues after both chains
    for (int k = 0; k < 2; ++k) {
      hipEvent_t j;
      TRY(pipe_event(h, &j));
      HIPCHK(hipEventRecord(j, P.s[k]));
      HIPCHK(hipStreamWaitEvent(P.main, j, 0));
    }
  }
  return 0;
}

// One proposal per electron, in index order, on the SoA state (lw: two launches per move, above) or the AoS state with the
// wave-per-walker kernels (multi-determinant, three-body, large complex determinants); mb.dmc selects the DMC variant.
int sweep_electrons(pqa_handle* h, const MoveBuf& mb, bool lw, const LwCtx& lc) {
  if (lw) return sweep_electrons_fused(h, mb, lc);
  const long W = h->W;
  if (ww_eligible(h, W)) return sweep_ww(h, mb);  // small shards: the whole sweep of a walker in one launch, three waves per walker (pqa_ww.hpp)
  const size_t lds_acc = std::max(lds_sm(h), lds_det(h, 5));
  for (int e = 0; e < h->N; ++e) {
    const int s = e >= h->nup;
    const double* mo = (const double*)h->b_motmp.p;
    if (h->cplx) {
      hipLaunchKernelGGL(k_propose<true>, dim3((unsigned)W), dim3(64), 2 * lds_det(h, 5), h->stream, h->S, h->st, h->js, mb, e,
                         (int)h->has_slater, (int)h->has_jastrow, W);
      if (h->has_slater) TRY(launch_orb(h, s, plain_points(mb.newpos, W), W, 5, (double*)h->b_motmp.p));
      hipLaunchKernelGGL(k_accept<true>, dim3((unsigned)W), dim3(64), 2 * lds_acc, h->stream, h->S, h->st, h->js, mb, e,
                         (int)h->has_slater, (int)h->has_jastrow, mo, W);
      continue;
    }
    hipLaunchKernelGGL(k_propose<false>, dim3((unsigned)W), dim3(64), lds_det(h, 5), h->stream, h->S, h->st, h->js, mb, e,
                       (int)h->has_slater, (int)h->has_jastrow, W);
    if (h->has_slater) TRY(launch_orb(h, s, plain_points(mb.newpos, W), W, 5, (double*)h->b_motmp.p));
    hipLaunchKernelGGL(k_accept<false>, dim3((unsigned)W), dim3(64), lds_acc, h->stream, h->S, h->st, h->js, mb, e, (int)h->has_slater,
                       (int)h->has_jastrow, mo, W);
  }
  return 0;
}

extern "C" int pqa_vmc_sweeps(pqa_handle_t* h, double tstep, int nsteps, const double* gauss, const double* unif, double threshold,
                              const double* ecp_rot, const double* ecp_unif, uint64_t seed, double* acceptance,
                              double* energy_mean, uint8_t* accept_rec) {
  h->dmc_old_valid = false;  // (the state pqa_dmc_continue refers to is gone)
  HIPCHK(hipSetDevice(h->device));
  if (h->W == 0) FAIL("state not initialised (call pqa_wf_recompute)");
  if (nsteps <= 0) return 0;
  if (h->ecpb_on && (ecp_rot || ecp_unif)) FAIL("the ECP tapes of pqa_vmc_sweeps have the semi-local integrator's layout: with pqa_set_ecp_batched the draws come from the device streams (replay through pqa_energy)");
  const long W = h->W;
  const int N = h->N;
  h->saved_valid = false;
  const int nmo_max = std::max(h->nmo[0], h->nmo[1]);
  TRY(ensure(h, h->b_newpos, (size_t)W * 3 * sizeof(double)));
  TRY(ensure(h, h->b_aux, (size_t)W * 8 * sizeof(double)));
  TRY(ensure(h, h->b_accept, (size_t)W));
  TRY(ensure(h, h->b_acccnt, (size_t)nsteps * sizeof(int)));
  TRY(ensure(h, h->b_motmp, (size_t)W * 5 * std::max(nmo_max, 1) * sizeof(double)));
  const int nen = h->cplx ? 7 : 6;  // energy rows: complex determinants add Im(ecp) = Im(total)
  TRY(ensure(h, h->b_means, (size_t)nsteps * nen * sizeof(double)));
  TRY(ensure(h, h->b_accw, (size_t)W * sizeof(int)));
  HIPCHK(hipMemsetAsync(h->b_acccnt.p, 0, (size_t)nsteps * sizeof(int), h->stream));
  HIPCHK(hipMemsetAsync(h->b_accw.p, 0, (size_t)W * sizeof(int), h->stream));
  if (h->S.pbc) {  // wrap counters of this call's accepted moves (pqa_get_wrap)
    TRY(ensure(h, h->b_dwrap, (size_t)W * 3 * sizeof(int)));
    TRY(ensure(h, h->b_wrap, (size_t)W * N * 3 * sizeof(int)));
    HIPCHK(hipMemsetAsync(h->b_wrap.p, 0, (size_t)W * N * 3 * sizeof(int), h->stream));
    h->wrap_W = W;
  }
  if (gauss) TRY(ensure(h, h->b_gauss, (size_t)N * W * 3 * sizeof(double)));
  if (unif) TRY(ensure(h, h->b_unif, (size_t)N * W * sizeof(double)));
  if (accept_rec) TRY(ensure(h, h->b_accrec, (size_t)N * W));
  const size_t nrot = (size_t)N * std::max(h->necp, 1);
  const bool tile = tile_eligible(h);
  const bool lw = !tile && h->lw_mode != 0 && h->has_slater && h->ndet == 1 && !h->has_j3 && (!h->cplx || std::max(h->nup, h->ndn) <= 32);
  LwCtx lc;
  TRY(lw_setup(h, lw, lc));
  h->draw_ahead_valid = false;
  for (int step = 0; step < nsteps; ++step) {
    MoveBuf mb{};
    mb.newpos = (double*)h->b_newpos.p; mb.aux = (double*)h->b_aux.p; mb.accept = (uint8_t*)h->b_accept.p;
    mb.acc_w = (int*)h->b_accw.p; mb.seed = seed; mb.step = (uint32_t)step; mb.tstep = tstep;
    if (h->S.pbc && !h->twist) { mb.dwrap = (int*)h->b_dwrap.p; mb.wrap = (int*)h->b_wrap.p; }  // twisted handles keep the walkers unfolded
    if (gauss) {
      TRY(copy_in(h, h->b_gauss.p, gauss + (size_t)step * N * W * 3, (size_t)N * W * 3 * sizeof(double)));
      mb.gauss = (const double*)h->b_gauss.p;
    }
    if (unif) {
      TRY(copy_in(h, h->b_unif.p, unif + (size_t)step * N * W, (size_t)N * W * sizeof(double)));
      mb.unif = (const double*)h->b_unif.p;
    }
    if (accept_rec) mb.accept_rec = (uint8_t*)h->b_accrec.p;
    h->jsx_current = false;
    h->r8_xaos_next = energy_mean != nullptr && h->necp > 0;  // (consumed by sweep_r8 only)
    h->draws_on_device = false;
    if (tile) TRY(sweep_tile(h, mb));
    else TRY(sweep_electrons(h, mb, lw, lc));
    h->r8_xaos_next = false;
    if (h->draws_on_device && energy_mean && step + 1 < nsteps && W >= 16384 && h->draws_ahead_on) TRY(draws_ahead(h, seed, (uint32_t)(step + 1)));
    // small shards: the accepted-move count, the energy rows and their means in one launch at the end of the step (three launches of ~5 us
    // otherwise — 2 % of the 50-determinant molecule's step at 2 048 walkers)
    const bool finish1 = energy_mean && W <= 16384;
    if (!finish1) hipLaunchKernelGGL((k_sum_reset_int<>), dim3(1), dim3(1024), 0, h->stream, (int*)h->b_accw.p, W, (int*)h->b_acccnt.p + step);
    TRY(check_launch(h, "k_propose/k_accept"));
    if (accept_rec) TRY(copy_in(h, accept_rec + (size_t)step * N * W, h->b_accrec.p, (size_t)N * W));
    if (energy_mean) {
      TRY(energy_dev(h, threshold, ecp_rot ? ecp_rot + (size_t)step * nrot * 9 : nullptr,
                     ecp_unif ? ecp_unif + (size_t)step * nrot * W : nullptr, seed, (uint32_t)step, lw, /*aos_T_needed=*/false, /*assemble=*/!finish1));
      if (finish1)
        hipLaunchKernelGGL((k_energy_finish<>), dim3(nen + 1), dim3(256), 0, h->stream, (const double*)h->b_kc.p, h->en_d_ecp, h->ii_energy, W,
                           (double*)h->b_en.p, (double*)h->b_means.p + (size_t)step * nen, nen, (int*)h->b_accw.p, (int*)h->b_acccnt.p + step);
      else
        hipLaunchKernelGGL((k_row_means<>), dim3(nen), dim3(256), 0, h->stream, (const double*)h->b_en.p, W, (double*)h->b_means.p + (size_t)step * nen);
      TRY(check_launch(h, "k_row_means"));
    }
  }
  h->aos_stale = lw;  // converted back on demand (sync_aos)
  h->jas_stale = h->has_j2;
  std::vector<int> cnt(nsteps);
  TRY(copy_out(h, cnt.data(), h->b_acccnt.p, (size_t)nsteps * sizeof(int)));
  if (acceptance) {
    std::vector<double> acc(nsteps);
    for (int i = 0; i < nsteps; ++i) acc[i] = (double)cnt[i] / ((double)W * N);
    HIPCHK(hipMemcpy(acceptance, acc.data(), nsteps * sizeof(double), hipMemcpyDefault));
  }
  if (energy_mean) HIPCHK(hipMemcpy(energy_mean, h->b_means.p, (size_t)nsteps * nen * sizeof(double), hipMemcpyDefault));
  return 0;
}

#ifdef PQA_PRE_CLK  // timing build only (tools/scratch/pre_clk.py)
extern "C" int pqa_debug_pre_clk(unsigned long long* dst, int n) {
  return (int)hipMemcpyFromSymbol(dst, HIP_SYMBOL(pqa_pre_clk), (size_t)n * sizeof(unsigned long long));
}
#endif
#ifdef PQA_WW_CLK  // timing build only (tools/scratch/ww_clk.py)
extern "C" int pqa_debug_ww_clk(unsigned long long* dst, int n) {
  return (int)hipMemcpyFromSymbol(dst, HIP_SYMBOL(pqa_ww_clk), (size_t)n * sizeof(unsigned long long));
}
#endif
