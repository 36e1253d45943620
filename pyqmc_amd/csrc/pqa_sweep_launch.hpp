// Launch wrappers of the lane-per-walker step and flush kernels (pqa_lw.hpp), shared by pqa_sweep.hip (real determinants)
// and pqa_sweep_cx.hip (complex ones): the two families are separate translation units only to compile in parallel.
#pragma once
#include "pqa_internal.hpp"

// Lane-per-walker sweep, two launches per move: k_orb at the proposal, then k_step_lw = decide electron e + propose electron
// e + 1 (pqa_lw.hpp).  The two halves are launched apart where the blocked Sherman-Morrison update has to flush in between
// (e + 1 opens a new electron block of the same spin: its inverse row is only current after k_flush_lw).
// what k_step_pre's scope asks of the system whatever its group count (pqa_lw.hpp)
static inline bool step_pre_system_ok(const pqa_handle* h, int rowlen) {
  return !h->cplx && h->step_pre && h->S.occ_ident[0] && h->S.occ_ident[1] && h->S.nb <= PQA_JAS_NF && h->S.na <= PQA_JAS_NF && rowlen <= 64;
}
template <bool PBC, bool CX>
static void launch_step_lw(pqa_handle* h, const LwState& L, const MoveBuf& mb, const StepArgs& a, int rowlen) {
  const dim3 grid((unsigned)((a.w1 - a.w0 + a.NW - 1) / a.NW)), block((unsigned)(a.NW * a.G));
  // small shards: the variant with every load issued at entry (k_step_pre, pqa_lw.hpp) where its scope covers the system
  // (one block per CU at most: the kernel holds ~360 registers per lane, one wave per SIMD)
  const bool pre_ok = !CX && step_pre_system_ok(h, rowlen) && (a.e_acc < 0 || a.j_hi - a.j_lo <= (a.G >= 32 ? a.G / 4 : a.G));  // (32 / 64 groups: a block row per quartet of groups)
  if (pre_ok && a.NW == 16 && (a.G == 32 || a.G == 64) && a.W <= h->step_pre_max && h->N <= 64 && h->S.natom <= 64) {  // 512 / 1024 threads per 16 walkers
#define PQA_STEP_W(NM) do { const size_t lds_p = ((size_t)8 * a.G + 3 * NM + 8) * a.NW * sizeof(double); \
      if (a.G == 64) hipLaunchKernelGGL((k_step_pre<PBC, NM, 64>), grid, block, lds_p, h->stream, h->S, L, mb, a); \
      else hipLaunchKernelGGL((k_step_pre<PBC, NM, 32>), grid, block, lds_p, h->stream, h->S, L, mb, a); } while (0)
    if (rowlen <= 8) PQA_STEP_W(8); else if (rowlen <= 16) PQA_STEP_W(16); else if (rowlen <= 32) PQA_STEP_W(32); else PQA_STEP_W(64);
#undef PQA_STEP_W
    return;
  }
  if (pre_ok && a.NW < 64 && a.G >= 8 && a.G <= 16 && a.W <= h->step_pre_max && h->N <= PQA_PRE_NP * a.G && h->S.natom <= PQA_PRE_NA * a.G) {
#define PQA_STEP_P(NM) do { const size_t lds_p = ((size_t)8 * a.G + 3 * NM + 8) * a.NW * sizeof(double); \
      hipLaunchKernelGGL((k_step_pre<PBC, NM>), grid, block, lds_p, h->stream, h->S, L, mb, a); } while (0)
    if (rowlen <= 8) PQA_STEP_P(8); else if (rowlen <= 16) PQA_STEP_P(16); else if (rowlen <= 32) PQA_STEP_P(32); else PQA_STEP_P(64);
#undef PQA_STEP_P
    return;
  }
  const size_t lds = (size_t)std::max(PQA_LW_PART_ROWS(CX) * a.G, 2 * rowlen) * a.NW * sizeof(double);
#define PQA_STEP(NM) do { if (a.NW == 64) hipLaunchKernelGGL((k_step_lw<PBC, CX, NM, true>), grid, block, lds, h->stream, h->S, L, mb, a); \
                          else hipLaunchKernelGGL((k_step_lw<PBC, CX, NM, false>), grid, block, lds, h->stream, h->S, L, mb, a); } while (0)
  if (rowlen <= 8) PQA_STEP(8); else if (rowlen <= 16) PQA_STEP(16); else if (rowlen <= 32) PQA_STEP(32); else PQA_STEP(64);
#undef PQA_STEP
}
// k_flush_lw on walkers [w0, w1): rows outside [j_lo, j_hi) of spin s take the block's nq buffered updates
template <bool CX>
static void launch_flush_lw(pqa_handle* h, const LwState& L, int s, long W, long w0, long w1, int j_lo, int j_hi, int nq, int rowlen, int n_s) {
  constexpr int cfi = CX ? 2 : 1;
  const long Wn = w1 - w0;
#define PQA_FLUSH_W(NM, WB_) do { const size_t lds_f = (size_t)2 * nq * cfi * n_s * WB_ * sizeof(double); const dim3 gf((unsigned)((Wn + WB_ - 1) / WB_)); \
      hipLaunchKernelGGL((k_flush_lw<NM, CX, WB_>), gf, dim3(256), lds_f, h->stream, h->S, L, s, (const double*)h->b_vbuf.p, (const double*)h->b_rbuf.p, (const uint8_t*)h->b_act.p, W, w0, w1, j_lo, j_hi, nq); } while (0)
#define PQA_FLUSH(NM) do { if (W <= h->flush_wb8_max) PQA_FLUSH_W(NM, 8); else PQA_FLUSH_W(NM, PQA_FLUSH_WB); } while (0)
  if (rowlen <= 8) PQA_FLUSH(8); else if (rowlen <= 16) PQA_FLUSH(16); else if (rowlen <= 32) PQA_FLUSH(32); else PQA_FLUSH(64);
#undef PQA_FLUSH_W
#undef PQA_FLUSH
}
