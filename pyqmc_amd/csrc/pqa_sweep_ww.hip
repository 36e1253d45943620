// pyqmc_amd C ABI implementation (host side): the wave-per-walker electron sweep in one launch (pqa_ww.hpp) — eligibility, launch.
// Its three waves run different device functions of one move side by side, so the synchronisation INSIDE those functions (PQA_WSYNC,
// pqa_common.hpp) is the wave-level fence here: LDS operations of a wave execute in order, the fence keeps the compiler from moving them.
#define PQA_WSYNC() do { __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront"); __builtin_amdgcn_wave_barrier(); } while (0)
#define PQA_SYNC_NS pqa_sync_wave
#include "pqa_internal.hpp"
#include "pqa_ww.hpp"

static size_t ww_xoff(const pqa_handle* h) {  // doubles in front of the exchange area: what k_propose / k_accept keep in dynamic LDS
  const size_t b = std::max(std::max(lds_sm(h), lds_det(h, 5)), lds_j3(h));
  return (b + 7) / 8;
}
static bool ww_cstage(const pqa_handle* h) { return (size_t)h->nao * (h->nmo[0] + h->nmo[1]) * sizeof(double) <= 16 * 1024; }  // coefficient matrices in LDS
static size_t ww_lds(const pqa_handle* h) {
  return (ww_xoff(h) + PQA_WW_XCH + (size_t)5 * std::max(std::max(h->nmo[0], h->nmo[1]), 1) + (size_t)5 * h->nao +
          (ww_cstage(h) ? (size_t)h->nao * (h->nmo[0] + h->nmo[1]) : 0)) * sizeof(double);
}

bool ww_eligible(pqa_handle* h, long W) {
  if (h->ww_mode == 0 || h->cplx || h->S.pbc) return false;
  if (ww_lds(h) > 64 * 1024) return false;
  for (int l : h->shell_l)
    if (l > 5) return false;
  return h->ww_mode > 0 || W <= h->ww_max;
}

int sweep_ww(pqa_handle* h, const MoveBuf& mb) {
  const long W = h->W;
  int lmax = 0;
  for (int l : h->shell_l) lmax = std::max(lmax, l);
  const int nwv = h->ww_mode == 3 ? 3 : 1;  // PQA_WW=3: three waves per walker
  const dim3 grid((unsigned)W), block(64 * nwv);
  const size_t lds = ww_lds(h);
  const int xoff = (int)ww_xoff(h);
#define PQA_WW_LAUNCH(LM, NW) hipLaunchKernelGGL((k_sweep_ww<LM, NW>), grid, block, lds, h->stream, h->S, h->st, h->js, mb, (int)h->has_slater, (int)h->has_jastrow, xoff, (int)ww_cstage(h), W)
  if (nwv == 3) { if (lmax <= 2) PQA_WW_LAUNCH(2, 3); else if (lmax <= 3) PQA_WW_LAUNCH(3, 3); else PQA_WW_LAUNCH(5, 3); }
  else { if (lmax <= 2) PQA_WW_LAUNCH(2, 1); else if (lmax <= 3) PQA_WW_LAUNCH(3, 1); else PQA_WW_LAUNCH(5, 1); }
#undef PQA_WW_LAUNCH
  return check_launch(h, "k_sweep_ww");
}

#ifdef PQA_WW_CLK  // timing build only
extern "C" int pqa_debug_ww1_clk(unsigned long long* dst, int n) {
  return (int)hipMemcpyFromSymbol(dst, HIP_SYMBOL(pqa_ww1_clk), (size_t)n * sizeof(unsigned long long));
}
#endif
