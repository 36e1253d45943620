// pyqmc_amd C ABI implementation (host side): periodic orbital launches (k_pbc_prepass, k_orb<.., PBC>, k_orb_wide<.., PBC>).
#include "pqa_orb_common.hpp"

// periodic orbitals: lattice-summed shells, 64-point tiles, tables through the scalar cache
template <int NCOMP, int KC>
static int launch_orb_pbc(pqa_handle* h, int tabi, int spin, PointAddr pa, long P, double* out) {
  TRY(ensure(h, h->b_pbcd0, (size_t)h->natom * (h->twist ? 5 : 3) * P * sizeof(double)));
  const int NW = h->pbc_nw;
  TRY(ensure(h, h->b_pbcmask, (size_t)h->natom * NW * P * sizeof(unsigned long long)));
  if (h->twist) TRY(ensure(h, h->b_pbcth, (size_t)2 * P * sizeof(double)));
  {
    const dim3 grid((unsigned)((P + PQA_PRE_NT - 1) / PQA_PRE_NT), (unsigned)h->natom);
    const size_t lds = (size_t)2 * 4 * std::min(NW, PQA_PRE_NWMAX) * PQA_PRE_NT * sizeof(unsigned short);
    if (h->pbc_maxcls <= 5)
      hipLaunchKernelGGL((k_pbc_prepass<5>), grid, dim3(PQA_PRE_NT), lds, h->stream, h->S, pa, P, NW, (double*)h->b_pbcd0.p,
                         (unsigned long long*)h->b_pbcmask.p, (double*)h->b_pbcth.p);
    else
      hipLaunchKernelGGL((k_pbc_prepass<PQA_PRE_NCUT>), grid, dim3(PQA_PRE_NT), lds, h->stream, h->S, pa, P, NW, (double*)h->b_pbcd0.p,
                         (unsigned long long*)h->b_pbcmask.p, (double*)h->b_pbcth.p);
  }
  ChunkTab T = tabx(h, tabi);
  T.pbc_d0 = (const double*)h->b_pbcd0.p;
  T.pbc_list = (const unsigned long long*)h->b_pbcmask.p;
  T.pbc_nw = NW;
  if (wide_wanted(h, tabi, P, NCOMP)) {  // small launch: one 1024-thread block per 16-point tile, the whole basis in LDS
    if (h->twist) TRY((launch_orb_wide<2, 512>(h, T, tabi, spin, pa, P, out)));
    else if (h->wide_nth == 1024) TRY((launch_orb_wide<1, 1024>(h, T, tabi, spin, pa, P, out)));
    else TRY((launch_orb_wide<1, 512>(h, T, tabi, spin, pa, P, out)));
    if (h->twist) {
      const long nel = P * NCOMP * (h->nmo[spin] / 2);
      hipLaunchKernelGGL((k_row_phase<>), dim3((unsigned)((nel + 255) / 256)), dim3(256), 0, h->stream, out, P, NCOMP, h->nmo[spin],
                         (const double*)h->b_pbcth.p, h->out_sel, h->out_slot_stride);
    }
    return 0;
  }
  // Tile width.  32-point tiles: twice the blocks, and the 8 lane groups halve each thread's share of a chunk's lattice
  // sums; 64-point tiles: half the B-operand and table traffic per point.  Which wins depends on cell and launch size
  // (2x2x2 diamond supercell, 16 atoms: 32 wins at every size, 28.5 -> 21.9 ms/step at 1024 walkers, 107.5 -> 100.0 at
  // 32768; 8-atom cubic cell: 32 wins up to 16384 points, 64 wins by 14 % at 32768), both give bit-identical rows, so
  // large launches time each twice per size class (four stream synchronisations in the handle's lifetime per class) and
  // keep the faster; small ones take 32.  PQA_ORB_TP pins it.
  // small launches: 16-point tiles (2x2x2 diamond: 19.8 -> 15.6 ms/step at 1024 walkers, 25.5 -> 21.9 at 4096, +3.5 % at 8192;
  // the twisted 8-atom cell loses 13 % at 8192, hence the threshold)
  int tp = (P <= 4096) ? 16 : 32;
  pqa_handle::TpTune* tune = nullptr;
  int tune_slot = -1;
  hipEvent_t te0 = nullptr, te1 = nullptr;
  if (h->orb_tp == 16 || h->orb_tp == 32 || h->orb_tp == 64) tp = h->orb_tp;
  else if (P >= 16384) {
    int b = 0;
    while ((2L << b) <= P && b < 46) ++b;
    tune = &h->tp_tune[tabi & 1][b];
    if (tune->choice) tp = tune->choice;
    else {
      tune_slot = tune->n[0] <= tune->n[1] ? 0 : 1;  // alternate; best of two samples each
      tp = tune_slot ? 64 : 32;
      HIPCHK(hipEventCreate(&te0));
      HIPCHK(hipEventCreate(&te1));
      HIPCHK(hipEventRecord(te0, h->stream));
    }
  }
  // small launches: split the chunk loop over two blocks per point tile (k_orb: gridDim.y), output accumulated atomically
  const int nsplit = (P <= h->orb_split_max && T.nchunk >= 4 && !h->orb_nosplit) ? 2 : 1;
  if (nsplit > 1) {
    if (h->out_sel) hipLaunchKernelGGL((k_zero_rows<>), dim3((unsigned)P, (unsigned)((NCOMP * h->nmo[spin] + 255) / 256)), dim3(256), 0, h->stream, out,
                                       NCOMP * h->nmo[spin], h->out_sel, h->out_slot_stride);
    else HIPCHK(hipMemsetAsync(out, 0, (size_t)P * NCOMP * h->nmo[spin] * sizeof(double), h->stream));
  }
  const dim3 grid((unsigned)((P + tp - 1) / tp), (unsigned)nsplit), block(256);
  // basis tables in LDS when they fit: besides the faster table reads, the larger LDS footprint makes the compiler
  // budget registers for 2 blocks per CU instead of 4 (128 registers + 800 B of scratch spills otherwise)
  const bool lt = h->nshell <= PQA_WS_MAXSH && (int)h->S.nprim <= PQA_WS_MAXP && !h->orb_notab;
#define PQA_ORB_PBC2(NT, LT, TPV) do { if (h->twist) hipLaunchKernelGGL((k_orb<NCOMP, NT, KC, TPV, LT, 2>), grid, block, 0, h->stream, h->S, T, spin, pa, P, out); \
                                       else hipLaunchKernelGGL((k_orb<NCOMP, NT, KC, TPV, LT, 1>), grid, block, 0, h->stream, h->S, T, spin, pa, P, out); } while (0)
#define PQA_ORB_PBC(NT, LT) do { if (tp == 64) PQA_ORB_PBC2(NT, LT, 64); else if (tp == 16) PQA_ORB_PBC2(NT, LT, 16); else PQA_ORB_PBC2(NT, LT, 32); } while (0)
  switch (h->nt[spin]) {
    case 1: if (lt) PQA_ORB_PBC(1, true); else PQA_ORB_PBC(1, false); break;
    case 2: if (lt) PQA_ORB_PBC(2, true); else PQA_ORB_PBC(2, false); break;
    default: if (lt) PQA_ORB_PBC(4, true); else PQA_ORB_PBC(4, false); break;
  }
#undef PQA_ORB_PBC2
  if (tune_slot >= 0) {
    HIPCHK(hipEventRecord(te1, h->stream));
    HIPCHK(hipEventSynchronize(te1));
    float ms = 0.f;
    HIPCHK(hipEventElapsedTime(&ms, te0, te1));
    HIPCHK(hipEventDestroy(te0));
    HIPCHK(hipEventDestroy(te1));
    tune->ms[tune_slot] = std::min(tune->ms[tune_slot], ms);
    if (++tune->n[tune_slot] >= 2 && tune->n[1 - tune_slot] >= 2) tune->choice = tune->ms[1] < tune->ms[0] ? 64 : 32;
  }
#undef PQA_ORB_PBC
  if (h->twist) {
    const long nel = P * NCOMP * (h->nmo[spin] / 2);
    hipLaunchKernelGGL((k_row_phase<>), dim3((unsigned)((nel + 255) / 256)), dim3(256), 0, h->stream, out, P, NCOMP, h->nmo[spin],
                       (const double*)h->b_pbcth.p, h->out_sel, h->out_slot_stride);
  }
  return 0;
}
int launch_ao(pqa_handle* h, PointAddr pa, long P, int ncomp, double* out) {
  const dim3 grid((unsigned)((P + 63) / 64)), block(64);
#define PQA_AO(NC) do { if (h->pbc_high_l) hipLaunchKernelGGL((k_ao<NC, 5>), grid, block, 0, h->stream, h->S, pa, P, out); \
                        else hipLaunchKernelGGL((k_ao<NC, 3>), grid, block, 0, h->stream, h->S, pa, P, out); } while (0)
  if (ncomp == 1) PQA_AO(1); else if (ncomp == 4) PQA_AO(4); else if (ncomp == 5) PQA_AO(5); else FAIL("AO evaluation: ncomp must be 1, 4 or 5");
#undef PQA_AO
  return check_launch(h, "k_ao");
}
// periodic cells with g / h shells: AO planes by the thread-per-point evaluator (direct image tests, the reference's cut-offs and
// membership rule like every other path), contracted by k_mo_rows into the row layout the callers expect
int launch_orb_general(pqa_handle* h, int ncomp, int spin, PointAddr pa, long P, double* out) {
  long chunk = std::max<long>(1, ((long)1 << 28) / ((long)ncomp * h->nao));  // <= 2 GiB of AO planes per pass
  if (pa.group_stride != 0 && P > chunk) {  // grouped lists (an electron block per walker): whole groups per pass
    if (pa.group > chunk) FAIL("general orbital path: a point group longer than one pass");
    chunk -= chunk % pa.group;
  }
  for (long p0 = 0; p0 < P; p0 += chunk) {
    const long n = std::min(chunk, P - p0);
    TRY(ensure(h, h->b_ao, (size_t)n * ncomp * h->nao * sizeof(double)));
    PointAddr pp = pa;
    if (pa.group_stride != 0) pp.base = pa.base + (p0 / pa.group) * pa.group_stride;
    else if (p0) { pp.base = pa.base + 3 * p0; pp.group = (int)n; }  // (plain lists: group = P, stride 0)
    TRY(launch_ao(h, pp, n, ncomp, (double*)h->b_ao.p));
    const long tot = n * ncomp * h->nmo[spin];
    hipLaunchKernelGGL((k_mo_rows<>), dim3((unsigned)((tot + 255) / 256)), dim3(256), 0, h->stream, (const double*)h->b_ao.p, (const double*)h->d_mo[spin], n,
                       ncomp, h->nao, h->nmo[spin], out + (size_t)p0 * ncomp * h->nmo[spin], h->out_sel ? h->out_sel + p0 : nullptr, h->out_slot_stride);
    TRY(check_launch(h, "k_mo_rows"));
  }
  return 0;
}
int launch_orb_pbc_any(pqa_handle* h, int ncomp, int spin, PointAddr pa, long P, double* out) {
  if (h->pbc_high_l || h->big) return launch_orb_general(h, ncomp, spin, pa, P, out);
  if (ncomp == 5) {
    // AO rows per chunk of the 5-component launch.  Automatic (PQA_ORB_KC5 unset): 16 for launches of at least 16384 points, 32 below —
    // measured at the end of round 4: whole-ensemble launches (recompute, the DMC step's refresh of accepted T-moves: 131 k points per
    // spin at 4096 walkers) run ~2x faster with 16-row chunks (C5 DMC 33.5 -> 31.4 ms per step at 16384 walkers, 62.8 -> 58.0 at 32768,
    // 10.5 -> 10.05 at 4096; 8-atom cubic cell VMC 12.7 -> 11.9 at 32768), 8192-point move launches of that cell lose 16 % with them.
    const int kc = (h->orb_kc5 == 16 || h->orb_kc5 == 32) ? h->orb_kc5 : (P >= 16384 ? 16 : 32);
    return (kc == 32) ? launch_orb_pbc<5, 32>(h, 1, spin, pa, P, out) : launch_orb_pbc<5, 16>(h, 0, spin, pa, P, out);
  }
  // value-only launches (ECP quadrature points, T-move candidates): 16-row chunks on the 5-component launch's chunk table
  // (C5 DMC 10.1 -> 9.8 ms per step at 4096 walkers, 31.2 -> 30.3 at 16384; C3 6.62 -> 6.47 at 8192; 2x2x2 VMC + 1.5-2 %; the 8-atom
  // cubic cell loses 1 %); PQA_ORB_KC1=32 restores the 32-row chunks
  if (ncomp == 1) return (h->orb_kc1 == 32) ? launch_orb_pbc<1, 32>(h, 1, spin, pa, P, out) : launch_orb_pbc<1, 16>(h, 0, spin, pa, P, out);
  FAIL("orbital kernel supports ncomp 1 or 5");
}

#ifdef PQA_WIDE_CLK  // timing build only (tools/scratch/wide_clk.py)
extern "C" int pqa_debug_wide_clk(unsigned long long* dst, int n) {
  return (int)hipMemcpyFromSymbol(dst, HIP_SYMBOL(pqa_wide_clk), (size_t)n * sizeof(unsigned long long));
}
#endif
