// Shared by pqa_orb.hip (open systems) and pqa_orb_pbc.hip (periodic cells): the chunk table of a launch and the whole-K
// small-launch kernel k_orb_wide.
#pragma once
#include "pqa_internal.hpp"

static inline ChunkTab tabx(const pqa_handle* h, int tabi) {  // the chunk table + where this launch's rows go
  ChunkTab T = h->tab[tabi];
  T.out_sel = h->out_sel;
  T.out_slot_stride = h->out_slot_stride;
  T.col0 = h->orb_col0;
  return T;
}
// whole-K kernel for small 5-component launches (k_orb_wide, pqa_ao.hpp)
static inline bool wide_wanted(const pqa_handle* h, int tabi, long P, int ncomp) {
  if (ncomp != 5 || h->orb_wide == 0 || h->wide[tabi].rows_pad <= 0) return false;
  if (wide_lds_bytes(5, h->wide[tabi].rows_pad, h->nshell, (int)h->S.nprim, (h->S.pbc && h->S.nL <= PQA_LS_MAX) ? 5 * h->S.nL : 0) > (size_t)160 * 1024 - 256) return false;
  if (h->orb_wide == 1) return true;
  // measured (tools/scratch/ab_wide*.sh, 1 MI355X): (H2O)8 step 6.65 -> 4.73 ms at 1024 walkers, 7.64 -> 5.86 at 4096, 9.08 -> 7.84
  // at 8192, even at 16384, slower at 32768 (one 1024-thread block per CU cannot overlap AO and MFMA phases of different
  // tiles); periodic cells (512 threads, two lane-group chains per point like the K-split k_orb): 2x2x2 diamond +5 / +8 / +2.5 %
  // at 1024 / 4096 / 8192 walkers, but the 8-atom cell (40 shells on 32 groups) and twisted cells (528 B of spills) lose
  // after the image lists / in-tile accumulation (no spills any more): twisted 8-atom cell 451k -> 580k walker-steps/s at 4096
  // walkers, 708k -> 756k at 8192; untwisted 8-atom cell even
  // ... and with the image walk / accumulation as they are now it wins up to 32768 points (C3 +18 % at 24576 walkers, +7 % at
  // 16384 and 32768; C5 +6 % at 12288, +1-2 % at 16384 and 32768): periodic threshold 4 x orb_wide_max
  if (h->S.pbc) return (h->twist || h->nshell >= 64) && P <= 4 * h->orb_wide_max;
  return P <= h->orb_wide_max + h->orb_wide_max / 2;
}
template <int PBCV, int NTH>
static int launch_orb_wide(pqa_handle* h, const ChunkTab& T, int tabi, int spin, PointAddr pa, long P, double* out) {
  const size_t lds = wide_lds_bytes(5, h->wide[tabi].rows_pad, h->nshell, (int)h->S.nprim, (h->S.pbc && h->S.nL <= PQA_LS_MAX) ? 5 * h->S.nL : 0);
  const dim3 grid((unsigned)((P + 15) / 16)), block(NTH);
#define PQA_WIDE(NT) do { const void* fn = (const void*)k_orb_wide<5, NT, PBCV, NTH>; \
    if (std::find(h->wide_attr.begin(), h->wide_attr.end(), fn) == h->wide_attr.end()) { \
      HIPCHK(hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024)); h->wide_attr.push_back(fn); } \
    hipLaunchKernelGGL((k_orb_wide<5, NT, PBCV, NTH>), grid, block, lds, h->stream, h->S, T, h->wide[tabi], spin, pa, P, out); } while (0)
  switch (h->nt[spin]) {
    case 1: PQA_WIDE(1); break;
    case 2: PQA_WIDE(2); break;
    default: PQA_WIDE(4); break;
  }
#undef PQA_WIDE
  return 0;
}

