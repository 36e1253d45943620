// pyqmc_amd C ABI implementation (host side): the second-generation resident sweep (pqa_res8.hpp) — work items of the wave-uniform AO
// phase, eligibility, launch.  Called by sweep_electrons_fused (pqa_sweep.hip) ahead of k_sweep_res and the launch-per-move sweep.
#include "pqa_internal.hpp"
#include "pqa_res8.hpp"

// Once per handle.  Work item = one shell type (same l and the same exponent / coefficient sequence: the same shell of every atom of a
// species) on up to eight atoms; items go to the four waves by descending cost (longest processing time first).  The tile holds the AOs in
// their own order (rows padded to x4) and the contraction reads the dense coefficient copy d_cres (shared with k_sweep_res's dense mode;
// res_refresh_coeff keeps it current after set_mo).
static int r8_setup(pqa_handle* h) {
  h->r8_ready = true;
  h->r8_ok = false;
  if (h->r8_mode == 0) return 0;
  if (!h->has_slater || h->ndet != 1 || h->has_j3 || h->cplx || h->twist || h->S.pbc || h->big) return 0;
  if (h->nup > 32 || h->ndn > 32 || h->nmo[0] > 32 || h->nmo[1] > 32 || h->N > 64 || h->N < 1 || h->natom > 64) return 0;
  int lmax = 0;
  for (int l : h->shell_l) lmax = std::max(lmax, l);
  if (lmax > 3) return 0;
  for (int s = 0; s < 2; ++s)
    if ((s ? h->ndn : h->nup) > 0 && (h->nt[s] < 1 || h->nt[s] > 2)) return 0;
  // deduplicated primitives (as res_setup) and the shells' atoms
  std::vector<double> pe_u, pc_u;
  std::vector<int> q0_u((size_t)h->nshell, 0), sat((size_t)h->nshell, 0);
  {
    std::vector<double> pe((size_t)h->S.nprim), pc((size_t)h->S.nprim);
    std::vector<int> po((size_t)h->nshell + 1);
    HIPCHK(hipMemcpy(pe.data(), h->S.prim_exp, pe.size() * sizeof(double), hipMemcpyDeviceToHost));
    HIPCHK(hipMemcpy(pc.data(), h->S.prim_coef, pc.size() * sizeof(double), hipMemcpyDeviceToHost));
    HIPCHK(hipMemcpy(po.data(), h->S.shell_prim_off, po.size() * sizeof(int), hipMemcpyDeviceToHost));
    HIPCHK(hipMemcpy(sat.data(), h->S.shell_atom, sat.size() * sizeof(int), hipMemcpyDeviceToHost));
    for (int sh = 0; sh < h->nshell; ++sh) {
      const int n = po[sh + 1] - po[sh];
      int found = -1;
      for (int prev = 0; prev < sh && found < 0; ++prev)
        if (po[prev + 1] - po[prev] == n && std::equal(pe.begin() + po[sh], pe.begin() + po[sh + 1], pe.begin() + po[prev]) &&
            std::equal(pc.begin() + po[sh], pc.begin() + po[sh + 1], pc.begin() + po[prev])) found = q0_u[prev];
      if (found < 0) { found = (int)pe_u.size(); pe_u.insert(pe_u.end(), pe.begin() + po[sh], pe.begin() + po[sh + 1]); pc_u.insert(pc_u.end(), pc.begin() + po[sh], pc.begin() + po[sh + 1]); }
      q0_u[sh] = found;
    }
  }
  struct Item { int l, np, q0, cost; std::vector<int> shells; };
  std::vector<Item> items;
  {
    std::vector<char> done((size_t)h->nshell, 0);
    for (int sh = 0; sh < h->nshell; ++sh) {
      if (done[sh]) continue;
      std::vector<int> mem;
      for (int q = sh; q < h->nshell; ++q)
        if (!done[q] && h->shell_l[q] == h->shell_l[sh] && q0_u[q] == q0_u[sh] && h->shell_np[q] == h->shell_np[sh]) { mem.push_back(q); done[q] = 1; }
      for (size_t k = 0; k < mem.size(); k += 8) {
        Item it{h->shell_l[sh], h->shell_np[sh], q0_u[sh], h->shell_cost[sh], {}};
        it.shells.assign(mem.begin() + k, mem.begin() + std::min(k + 8, mem.size()));
        items.push_back(it);
      }
    }
  }
  const int nitem = (int)items.size();
  if (nitem == 0) return 0;
  double used = 0.0;
  for (const Item& it : items) used += (double)it.shells.size();
  h->r8_util = used / (8.0 * nitem);
  std::vector<int> order((size_t)nitem);
  for (int k = 0; k < nitem; ++k) order[k] = k;
  std::stable_sort(order.begin(), order.end(), [&](int a, int b) { return items[a].cost > items[b].cost; });
  std::vector<std::vector<int>> wave(4);
  long load[4] = {0, 0, 0, 0};
  for (int k : order) {
    int w = 0;
    for (int q = 1; q < 4; ++q)
      if (load[q] < load[w]) w = q;
    wave[w].push_back(k);
    load[w] += items[k].cost;
  }
  R8Tab RT{};
  std::vector<int> hdr, lane;
  std::vector<double> axyz((size_t)3 * h->natom), ixyz;
  HIPCHK(hipMemcpy(axyz.data(), h->S.atom_xyz, axyz.size() * sizeof(double), hipMemcpyDeviceToHost));
  int pos = 0;
  for (int w = 0; w < 4; ++w) {
    RT.wave_off[w] = pos;
    for (int k : wave[w]) {
      const Item& it = items[k];
      hdr.push_back(it.l); hdr.push_back(it.np); hdr.push_back(it.q0); hdr.push_back(0);
      for (int q = 0; q < 8; ++q) {
        const bool on = q < (int)it.shells.size();
        lane.push_back(on ? sat[it.shells[q]] : -1);
        lane.push_back(on ? h->shell_ao[it.shells[q]] : 0);
        for (int d = 0; d < 3; ++d) ixyz.push_back(on ? axyz[3 * sat[it.shells[q]] + d] : 0.0);
      }
      ++pos;
    }
  }
  RT.wave_off[4] = pos;
  RT.nitem = nitem;
  const int rows4 = (h->nao + 3) & ~3;
  RT.kt = rows4;
  RT.cstride = 8 * rows4;
  while (RT.cstride % 32 != 16) RT.cstride += 8;  // planes c, c + 1 start 128 B apart mod 256: the two halves of an A operand on different banks
  size_t part_rn = 0;
  for (int s = 0; s < 2; ++s) {
    if ((s ? h->ndn : h->nup) == 0) continue;
    const int nt = h->nt[s], KW = 4 / nt;
    if (rows4 / 4 > PQA_R8_MAXQ * KW) return 0;
    part_rn = std::max(part_rn, (size_t)KW * PQA_R8_NW * res_ps(nt) + (size_t)PQA_R8_NW * PQA_RES_RS);
  }
  RT.jstage = (int)part_rn;
  RT.region = (int)std::max((size_t)5 * RT.cstride, part_rn + (size_t)PQA_R8_NW * 12 * 33);
  RT.nprim_u = (int)pe_u.size();
  h->r8_lds = (size_t)RT.region * sizeof(double) + r8_lds_fixed(RT.nprim_u, h->natom, h->na, nitem);
  h->r8_lds = (h->r8_lds + 15) & ~(size_t)15;
  if (h->r8_lds > 80 * 1024) return 0;  // two blocks per CU
  // dense coefficient copy
  h->res_rows4 = rows4;
  for (int s = 0; s < 2; ++s) {
    if (h->nmo[s] == 0 || h->d_cres[s]) continue;
    std::vector<double> mo((size_t)h->nao * h->nmo[s]);
    HIPCHK(hipMemcpy(mo.data(), h->d_mo[s], mo.size() * sizeof(double), hipMemcpyDeviceToHost));
    TRY(upload_table<double>(h, nullptr, (size_t)res_rows_alloc(rows4) * 16 * h->nt[s], &h->d_cres[s]));
    TRY(res_refresh_coeff(h, s, mo.data()));
  }
  int* tmp_i = nullptr;
  TRY(upload_table(h, hdr.data(), hdr.size(), &tmp_i)); RT.item_hdr = tmp_i;
  TRY(upload_table(h, lane.data(), lane.size(), &tmp_i)); RT.item_lane = tmp_i;
  double* tmp_d = nullptr;
  TRY(upload_table(h, ixyz.data(), ixyz.size(), &tmp_d)); RT.item_xyz = tmp_d;
  TRY(upload_table(h, pe_u.data(), pe_u.size(), &tmp_d)); RT.prim_exp_u = tmp_d;
  TRY(upload_table(h, pc_u.data(), pc_u.size(), &tmp_d)); RT.prim_coef_u = tmp_d;
  HIPCHK(hipFuncSetAttribute((const void*)k_sweep_r8<false, 2>, hipFuncAttributeMaxDynamicSharedMemorySize, 80 * 1024));
  HIPCHK(hipFuncSetAttribute((const void*)k_sweep_r8<true, 2>, hipFuncAttributeMaxDynamicSharedMemorySize, 80 * 1024));
  HIPCHK(hipFuncSetAttribute((const void*)k_sweep_r8<false, 3>, hipFuncAttributeMaxDynamicSharedMemorySize, 80 * 1024));
  HIPCHK(hipFuncSetAttribute((const void*)k_sweep_r8<true, 3>, hipFuncAttributeMaxDynamicSharedMemorySize, 80 * 1024));
  if (getenv("PQA_RES_DEBUG")) {
    fprintf(stderr, "[pqa_res8] %d items (slot use %.2f), tile rows %d, plane stride %d, LDS %zu B, wave loads %ld %ld %ld %ld\n", nitem, h->r8_util,
            RT.kt, RT.cstride, h->r8_lds, load[0], load[1], load[2], load[3]);
    int nb = 0;
    (void)hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, (const void*)k_sweep_r8<false, 2>, PQA_R8_NT, h->r8_lds);
    fprintf(stderr, "[pqa_res8] resident blocks per CU (k_sweep_r8<false, 2>): %d\n", nb);
    if (atoi(getenv("PQA_RES_DEBUG")) > 1)
      for (int w = 0; w < 4; ++w)
        for (int k : wave[w]) fprintf(stderr, "[pqa_res8] wave %d: l %d, np %d, %zu atoms, cost %d\n", w, items[k].l, items[k].np, items[k].shells.size(), items[k].cost);
  }
  if (const char* a = getenv("PQA_R8_STAGGER")) RT.stagger = atoi(a);
  if (const char* a = getenv("PQA_R8_ABL")) RT.abl = atoi(a);  // (only timing builds read it)
  h->r8_tab = RT;
  h->res_lmax = lmax;
  h->r8_ok = true;
  return 0;
}

bool r8_eligible(pqa_handle* h, long W) {
  if (h->res_mode == 0) return false;  // PQA_RES=0: the launch-per-move sweep
  if (!h->r8_ready) {
    if (r8_setup(h) != 0) { h->r8_ok = false; h->err.clear(); }
  }
  if (!h->r8_ok) return false;
  if (h->r8_mode > 0) return true;
  if (W < h->res_min || W > h->res_max) return false;
  return h->r8_util >= 0.5 && std::max(h->nup, h->ndn) >= 16;
}

int sweep_r8(pqa_handle* h, const MoveBuf& mb) {
  if (!mb.gauss || !mb.unif) FAIL("resident sweep: the random-number tapes are missing");
  const long W = h->W;
  const LwState L = lw_state(h);
  ChunkTab Tc = h->tab[0];
  Tc.cpad[0] = h->d_cres[0]; Tc.cpad[1] = h->d_cres[1];
  const dim3 grid((unsigned)((W + PQA_R8_NW - 1) / PQA_R8_NW)), block(PQA_R8_NT);
  h->r8_tab.xaos = h->r8_xaos_next ? h->js.x : nullptr;  // (pqa_vmc_sweeps: an energy evaluation with ECP passes follows)
  h->jsx_current = h->r8_xaos_next;
  h->r8_xaos_next = false;
  hipEvent_t e1 = nullptr;
  if (h->profile) {  // every launch is bracketed (one launch per sweep)
    if (h->prof_used == h->prof_events.size()) {
      hipEvent_t a, b;
      HIPCHK(hipEventCreate(&a));
      HIPCHK(hipEventCreate(&b));
      h->prof_events.emplace_back(a, b);
    }
    HIPCHK(hipEventRecord(h->prof_events[h->prof_used].first, h->stream));
    e1 = h->prof_events[h->prof_used].second;
    ++h->prof_used;
    h->prof_launches += 1;
    h->prof_pc += (double)W * h->N * 5;
  }
#define PQA_R8_LAUNCH(D, LM) hipLaunchKernelGGL((k_sweep_r8<D, LM>), grid, block, h->r8_lds, h->stream, h->S, L, mb, Tc, h->r8_tab, (int)h->has_jastrow, W, 0L, W)
  if (mb.dmc) { if (h->res_lmax <= 2) PQA_R8_LAUNCH(true, 2); else PQA_R8_LAUNCH(true, 3); }
  else { if (h->res_lmax <= 2) PQA_R8_LAUNCH(false, 2); else PQA_R8_LAUNCH(false, 3); }
#undef PQA_R8_LAUNCH
  if (e1) HIPCHK(hipEventRecord(e1, h->stream));
  return check_launch(h, "k_sweep_r8");
}

#ifdef PQA_RES_CLK  // timing build only (tools/scratch/res_clk.py --r8)
extern "C" int pqa_debug_r8_clk(unsigned long long* dst, int n) {
  return (int)hipMemcpyFromSymbol(dst, HIP_SYMBOL(pqa_res_clk), (size_t)n * sizeof(unsigned long long));
}
extern "C" int pqa_debug_r8_clk2(unsigned long long* dst, int n) {
  return (int)hipMemcpyFromSymbol(dst, HIP_SYMBOL(pqa_res_clk2), (size_t)n * sizeof(unsigned long long));
}
#endif
