// pyqmc_amd C ABI implementation (host side): the walker-tile sweep (opt-in, PQA_LW=2; pqa_tile.hpp).
// An experiment kept for A/B runs (on par with the lane-per-walker sweep up to ~3000 walkers, behind it
// above): compiled only into -DPQA_AB builds (python __graft_entry__.py --ab), the default library carries neither the
// kernel nor its launch code.
#include "pqa_internal.hpp"
#ifndef PQA_AB
bool tile_eligible(const pqa_handle*) { return false; }
int sweep_tile(pqa_handle* h, const MoveBuf&) { FAIL("the walker-tile sweep is only part of -DPQA_AB builds"); }
#else
bool tile_eligible(const pqa_handle* h) {
  if (h->lw_mode != 2 || !h->has_slater || h->ndet != 1 || h->has_j3 || h->cplx || h->S.pbc) return false;
  if (h->nup > 32 || h->ndn > 32 || h->nmo[0] > 32 || h->nmo[1] > 32) return false;
  for (int l : h->shell_l)
    if (l > 3) return false;
  const int nmo_pad = 16 * std::max(h->nt[0], h->nt[1]);
  return tile_lds_bytes(h->N, nmo_pad, h->nshell, (int)h->S.nprim, h->chunks[0].rows_pad) <= 160 * 1024 - 512;
}
// One sweep over all electrons for every walker, in one launch.  mb carries the step's tapes / seeds as for the other paths.
int sweep_tile(pqa_handle* h, const MoveBuf& mb_in) {
  MoveBuf mb = mb_in;
  if (!mb.gauss || !mb.unif) {  // no replay tapes: draw this sweep's numbers from the Philox streams first
    const size_t NW = (size_t)h->N * h->W;
    TRY(ensure(h, h->b_gauss, NW * 3 * sizeof(double)));
    TRY(ensure(h, h->b_unif, NW * sizeof(double)));
    hipLaunchKernelGGL((k_tile_draws<>), dim3((unsigned)((NW + 255) / 256)), dim3(256), 0, h->stream, mb.seed, mb.step, h->N, h->W,
                       (double*)h->b_gauss.p, (double*)h->b_unif.p);
    mb.gauss = (const double*)h->b_gauss.p; mb.unif = (const double*)h->b_unif.p;
  }
  const ChunkHost& c = h->chunks[0];
  TileTab TT{};
  TT.nmo_pad = 16 * std::max(h->nt[0], h->nt[1]);
  TT.rows_pad = c.rows_pad;
  TT.pass_chunk[0] = 0;
  const int nch = (int)c.nk.size();
  int ch = 0;
  while (ch < nch) {  // greedy: consecutive chunks while their padded rows fit the LDS tile
    if (TT.npass == PQA_TILE_MAXPASS) FAIL("walker-tile sweep: too many AO passes for this basis");
    const int base = c.row0[ch];
    int end = ch;
    while (end < nch && c.row0[end] + ((c.nk[end] + 3) & ~3) - base <= PQA_TILE_KT) ++end;
    if (end == ch) FAIL("walker-tile sweep: a chunk does not fit the AO tile");
    ch = end;
    TT.pass_chunk[++TT.npass] = ch;
  }
  const size_t lds = tile_lds_bytes(h->N, TT.nmo_pad, h->nshell, (int)h->S.nprim, TT.rows_pad);
  const dim3 grid((unsigned)((h->W + PQA_TILE_NW - 1) / PQA_TILE_NW)), block(PQA_TILE_NT);
  int lmax = 0;
  for (int sh = 0; sh < h->nshell; ++sh) lmax = std::max(lmax, h->shell_l[sh]);
  if (!h->tile_attr_set) {
    HIPCHK(hipFuncSetAttribute((const void*)k_sweep_tile<false, 2>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    HIPCHK(hipFuncSetAttribute((const void*)k_sweep_tile<true, 2>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    HIPCHK(hipFuncSetAttribute((const void*)k_sweep_tile<false, 3>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    HIPCHK(hipFuncSetAttribute((const void*)k_sweep_tile<true, 3>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    h->tile_attr_set = true;
  }
#define PQA_TILE_LAUNCH(D, LM) hipLaunchKernelGGL((k_sweep_tile<D, LM>), grid, block, lds, h->stream, h->S, h->st, h->js, mb, h->tab[0], TT, (int)h->has_jastrow, h->W)
  if (mb.dmc) { if (lmax <= 2) PQA_TILE_LAUNCH(true, 2); else PQA_TILE_LAUNCH(true, 3); }
  else { if (lmax <= 2) PQA_TILE_LAUNCH(false, 2); else PQA_TILE_LAUNCH(false, 3); }
#undef PQA_TILE_LAUNCH
  return check_launch(h, "k_sweep_tile");
}
#endif
