// pyqmc_amd C ABI implementation (host side): orbital kernel launches (k_orb, k_orb_ws, k_orb_wide, k_pbc_prepass).
// See include/pyqmc_amd.h for the contract and pqa_internal.hpp for what the units share.
#include "pqa_orb_common.hpp"
// ---------------------------------------------------------------- orbital kernel launch
int launch_orb_pbc_any(pqa_handle* h, int ncomp, int spin, PointAddr pa, long P, double* out);  // pqa_orb_pbc.hip
int launch_orb_general(pqa_handle* h, int ncomp, int spin, PointAddr pa, long P, double* out);  // pqa_orb_pbc.hip

template <int NCOMP, int KC>
static void launch_orb_ws(pqa_handle* h, int tabi, int spin, PointAddr pa, long P, double* out) {
  const dim3 grid((unsigned)((P + 63) / 64)), block(512);
  switch (h->nt[spin]) {
    case 1: hipLaunchKernelGGL((k_orb_ws<NCOMP, 1, KC>), grid, block, 0, h->stream, h->S, tabx(h, tabi), spin, pa, P, out); break;
    case 2: hipLaunchKernelGGL((k_orb_ws<NCOMP, 2, KC>), grid, block, 0, h->stream, h->S, tabx(h, tabi), spin, pa, P, out); break;
    default: hipLaunchKernelGGL((k_orb_ws<NCOMP, 4, KC>), grid, block, 0, h->stream, h->S, tabx(h, tabi), spin, pa, P, out); break;
  }
}

template <int NCOMP, int KC, int TP, bool LT>
static void launch_orb_t2(pqa_handle* h, int tabi, int spin, PointAddr pa, long P, double* out) {
  const dim3 grid((unsigned)((P + TP - 1) / TP)), block(256);
  const int left = (h->nmo[spin] - h->orb_col0 + 15) / 16;  // 16-column tiles from this launch's first column on (at most four per launch)
  switch (left) {
    case 1: hipLaunchKernelGGL((k_orb<NCOMP, 1, KC, TP, LT>), grid, block, 0, h->stream, h->S, tabx(h, tabi), spin, pa, P, out); break;
    case 2: hipLaunchKernelGGL((k_orb<NCOMP, 2, KC, TP, LT>), grid, block, 0, h->stream, h->S, tabx(h, tabi), spin, pa, P, out); break;
    default: hipLaunchKernelGGL((k_orb<NCOMP, 4, KC, TP, LT>), grid, block, 0, h->stream, h->S, tabx(h, tabi), spin, pa, P, out); break;
  }
}
template <int NCOMP, int KC, int TP>
static void launch_orb_t(pqa_handle* h, int tabi, int spin, PointAddr pa, long P, double* out) {
  if (h->nshell <= PQA_WS_MAXSH && (int)h->S.nprim <= PQA_WS_MAXP && !h->orb_notab) launch_orb_t2<NCOMP, KC, TP, true>(h, tabi, spin, pa, P, out);
  else launch_orb_t2<NCOMP, KC, TP, false>(h, tabi, spin, pa, P, out);
}

// out[p][ncomp][nmo_spin]
// out_sel / slot_stride: two-slot output (ChunkTab::out_sel), else plain rows
static int launch_orb_impl(pqa_handle* h, int spin, PointAddr pa, long P, int ncomp, double* out);
int launch_orb(pqa_handle* h, int spin, PointAddr pa, long P, int ncomp, double* out, const unsigned char* out_sel, long slot_stride) {
  h->out_sel = out_sel; h->out_slot_stride = slot_stride;
  const int rc = launch_orb_impl(h, spin, pa, P, ncomp, out);
  h->out_sel = nullptr; h->out_slot_stride = 0;
  return rc;
}
static int launch_orb_impl(pqa_handle* h, int spin, PointAddr pa, long P, int ncomp, double* out) {
  if (P <= 0 || h->nmo[spin] == 0) return 0;
  hipEvent_t e0 = nullptr, e1 = nullptr;
  // account the dominant (move) launches only, and only a 1-in-prof_stride sample of them: an event pair costs ~2 us of
  // stream time, 512 pairs per step were 1.2 ms of a 27 ms step
  const bool prof = h->profile && ncomp == 5 && (h->prof_tick++ % h->prof_stride) == 0;
  if (prof) {
    if (h->prof_used == h->prof_events.size()) {
      hipEvent_t a, b;
      HIPCHK(hipEventCreate(&a));
      HIPCHK(hipEventCreate(&b));
      h->prof_events.emplace_back(a, b);
    }
    e0 = h->prof_events[h->prof_used].first;
    e1 = h->prof_events[h->prof_used].second;
    ++h->prof_used;
    HIPCHK(hipEventRecord(e0, h->stream));
  }
  // 64-point tiles need >= ~4 blocks per CU to overlap their exp and MFMA phases across blocks; below
  // that, 32-point tiles double the number of resident blocks (PQA_ORB_TP overrides for A/B runs)
  const long Ps = h->orb_p_hint > 0 ? std::min(h->orb_p_hint, P) : P;  // points the launch is expected to work on (PointAddr::count: P is a bound)
  int tp = (Ps >= (long)64 * 448) ? 64 : 32;  // (28672 points: 15.4 -> 14.5 ms per (H2O)8 step with 64-point tiles; 24576: 11.7 vs 12.2 for 32)
  if (h->orb_tp == 32 || h->orb_tp == 64) tp = h->orb_tp;
  // measured on MI355X: below ~2 blocks per CU the wave-specialised schedule wins (its
  // producer and consumer waves overlap inside one block); with >= 4 resident blocks per CU the plain kernel does
  // (one 64-point block per CU at most: with a second round of blocks the plain kernel wins — (H2O)8 step 12.3 -> 10.6 ms at 18432
  // walkers, 12.9 -> 11.3 at 22528, measured at the end of round 4; at 16384 the two are level)
  const bool want_ws = h->orb_ws < 0 ? (Ps <= (long)64 * 256) : (h->orb_ws != 0);
  if (h->S.nL > 0) TRY(launch_orb_pbc_any(h, ncomp, spin, pa, P, out));
  else if (h->big && h->orb_general) TRY(launch_orb_general(h, ncomp, spin, pa, P, out));
  else if (h->big) {  // more than 64 orbitals of a spin: windows of 64 columns, one k_orb launch each (the AO phase runs once per window)
    for (int col0 = 0; col0 < 16 * h->nt[spin] && col0 < h->nmo[spin]; col0 += 64) {
      h->orb_col0 = col0;
      if (ncomp == 5) { if (tp == 64) launch_orb_t<5, 16, 64>(h, 0, spin, pa, P, out); else launch_orb_t<5, 16, 32>(h, 0, spin, pa, P, out); }
      else if (ncomp == 1) { if (tp == 64) launch_orb_t<1, 32, 64>(h, 1, spin, pa, P, out); else launch_orb_t<1, 32, 32>(h, 1, spin, pa, P, out); }
      else { h->orb_col0 = 0; FAIL("orbital kernel supports ncomp 1 or 5"); }
    }
    h->orb_col0 = 0;
  } else
  if (wide_wanted(h, 0, P, ncomp)) {
    TRY((launch_orb_wide<0, 1024>(h, tabx(h, 0), 0, spin, pa, P, out)));
  } else
  if (want_ws && h->nshell <= PQA_WS_MAXSH && (int)h->S.nprim <= PQA_WS_MAXP) {
    if (ncomp == 5) launch_orb_ws<5, 16>(h, 0, spin, pa, P, out);
    else if (ncomp == 1) launch_orb_ws<1, 32>(h, 1, spin, pa, P, out);
    else FAIL("orbital kernel supports ncomp 1 or 5");
  } else
  if (ncomp == 5) { if (tp == 64) launch_orb_t<5, 16, 64>(h, 0, spin, pa, P, out); else launch_orb_t<5, 16, 32>(h, 0, spin, pa, P, out); }
  else if (ncomp == 1) { if (tp == 64) launch_orb_t<1, 32, 64>(h, 1, spin, pa, P, out); else launch_orb_t<1, 32, 32>(h, 1, spin, pa, P, out); }
  else FAIL("orbital kernel supports ncomp 1 or 5");
  TRY(check_launch(h, "k_orb"));
  if (prof) {
    HIPCHK(hipEventRecord(e1, h->stream));
    h->prof_launches += 1;
    h->prof_pc += (double)P * ncomp;
  }
  return 0;
}

PointAddr plain_points(const double* base, long P) {
  PointAddr pa;
  pa.base = base;
  pa.group = (int)std::max<long>(P, 1);
  pa.group_stride = 0;
  return pa;
}
