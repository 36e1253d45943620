// pyqmc_amd C ABI implementation (host side): the resident electron sweep (pqa_res.hpp) — tables, eligibility, launch.
// Called by sweep_electrons_fused (pqa_sweep.hip) in place of the per-move launches of the lane-per-walker sweep.
#include "pqa_internal.hpp"

// Passes over the padded coefficient rows of the 16-row chunk table (h->chunks[0]: the coefficient matrices cpad[0] are shared
// with k_orb), shell lists per (pass, lane group), LDS budget.  Once per handle; leaves res_ok = false when the system is outside
// the kernel's scope.
// dense coefficient copy [rows4][ldc] of spin s in AO order (rows beyond nao and columns beyond nmo zero)
int res_refresh_coeff(pqa_handle* h, int s, const double* mo_host) {
  if (!h->d_cres[s] || h->nmo[s] == 0) return 0;
  const int ldc = 16 * h->nt[s], nmo = h->nmo[s];
  std::vector<double> pad((size_t)res_rows_alloc(h->res_rows4) * ldc, 0.0);  // (zero rows behind the basis: k_sweep_r8's K split)
  for (int a = 0; a < h->nao; ++a)
    for (int j = 0; j < nmo; ++j) pad[(size_t)a * ldc + j] = mo_host[(size_t)a * nmo + j];
  if (h->twist) {  // rows nao .. 2 nao: the imaginary AO parts, (i AO_im)(C_re + i C_im) = AO_im (-C_im + i C_re), columns [re | im] (upload_cpad)
    const int nr = nmo / 2;
    for (int a = 0; a < h->nao; ++a)
      for (int j = 0; j < nr; ++j) {
        pad[(size_t)(h->nao + a) * ldc + j] = -mo_host[(size_t)a * nmo + nr + j];
        pad[(size_t)(h->nao + a) * ldc + nr + j] = mo_host[(size_t)a * nmo + j];
      }
  }
  HIPCHK(hipStreamSynchronize(h->stream));
  HIPCHK(hipMemcpy(h->d_cres[s], pad.data(), pad.size() * sizeof(double), hipMemcpyHostToDevice));
  return 0;
}

static int res_setup(pqa_handle* h) {
  h->res_ready = true;
  h->res_ok = false;
  if (h->res_mode == 0) return 0;
  if (!h->has_slater || h->ndet != 1 || h->has_j3) return 0;
  // complex determinants: periodic cells (twisted or not), 16 electrons and 16 orbitals per spin — a row of the inverse is 32 doubles
  if (h->cplx && (!h->S.pbc || h->nup > 16 || h->ndn > 16 || h->res_cx == 0)) return 0;
  if (h->twist && !h->cplx) return 0;
  if (h->S.pbc && (h->S.nL <= 0 || h->pbc_high_l || h->res_pbc == 0 || !h->pbc_lists_ok)) return 0;  // periodic: lattice-summed orbitals, l <= 3 (PQA_RES_PBC=0 keeps the launches)
  if (h->nup > 32 || h->ndn > 32 || h->nmo[0] > 32 || h->nmo[1] > 32 || h->N > 64 || h->N < 1 || h->natom > 64) return 0;
  int lmax = 0;
  for (int l : h->shell_l) lmax = std::max(lmax, l);
  if (lmax > 3) return 0;
  h->res_lmax = lmax;
  const ChunkHost& c = h->chunks[0];
  const int nch = (int)c.nk.size();
  if (nch == 0) return 0;
  // primitives: shells with the same (exponent, coefficient) sequence — the same shell of every atom of a species — share one LDS copy
  std::vector<double> pe_u, pc_u;
  std::vector<int> q0_u((size_t)h->nshell, 0);
  {
    std::vector<double> pe((size_t)h->S.nprim), pc((size_t)h->S.nprim);
    std::vector<int> po((size_t)h->nshell + 1);
    HIPCHK(hipMemcpy(pe.data(), h->S.prim_exp, pe.size() * sizeof(double), hipMemcpyDeviceToHost));
    HIPCHK(hipMemcpy(pc.data(), h->S.prim_coef, pc.size() * sizeof(double), hipMemcpyDeviceToHost));
    HIPCHK(hipMemcpy(po.data(), h->S.shell_prim_off, po.size() * sizeof(int), hipMemcpyDeviceToHost));
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
  const int nprim_u = (int)pe_u.size();
  int rows_cap = 1 << 30;
  size_t part_rn = 0;
  for (int s = 0; s < 2; ++s) {
    if ((s ? h->ndn : h->nup) == 0) continue;
    const int nt = h->nt[s];
    if (nt < 1 || nt > 2) return 0;
    rows_cap = std::min(rows_cap, 4 * PQA_RES_MAXKS * (8 / nt));
    part_rn = std::max(part_rn, (size_t)(8 / nt) * 16 * res_ps(nt) + (size_t)16 * PQA_RES_RS);
  }
  // periodic: the image lists take what is left beside a one-pass tile — up to 32 entries per (point, atom), at least 10
  int icap = 0;
  size_t pbc_b = 0;
  auto pick_icap = [&](size_t rest) {  // largest capacity whose lists fit `rest` bytes (0: none)
    for (int cand = 32; cand >= 10; --cand)
      if (res_lds_pbc(h->natom, h->S.nL, cand, h->nshell, h->twist) + 8 <= rest) return cand;
    return 0;
  };
  if (h->S.pbc) {
    const size_t f0 = res_lds_fixed(h->nshell, nprim_u, h->natom, h->na, h->nshell, PQA_RES_MAXPASS);
    const size_t tile = (size_t)80 * c.rows_pad * sizeof(double), budget0 = (size_t)160 * 1024 - 256;
    icap = f0 + tile < budget0 ? pick_icap(budget0 - f0 - tile) : 0;
    if (icap == 0) icap = 10;
    pbc_b = res_lds_pbc(h->natom, h->S.nL, icap, h->nshell, h->twist) + 8;
  }
  const size_t fixed = res_lds_fixed(h->nshell, nprim_u, h->natom, h->na, h->nshell, PQA_RES_MAXPASS) + pbc_b;
  const size_t budget = 160 * 1024 - 256;
  if (fixed + part_rn * sizeof(double) > budget) return 0;
  const size_t avail = (budget - fixed) / sizeof(double);
  // one pass if the whole basis fits (the partials then reuse the tile's memory); otherwise the tile shares the region with them
  const int rows_all = c.rows_pad;
  // (twisted cells: dense rows only — the real rows of the whole basis, then the imaginary rows)
  const bool one = !h->twist && rows_all <= rows_cap && (size_t)80 * rows_all <= avail;
  // dense mode: the chunk padding (16-row chunks: 224 rows for the 208 AOs of the 2x2x2 diamond cell) is what keeps the basis out of one
  // tile, and the AOs in their own order (padded to x4) fit
  const int rows4 = ((h->twist ? 2 : 1) * h->nao + 3) & ~3;
  const size_t fixed1 = res_lds_fixed(h->nshell, nprim_u, h->natom, h->na, h->nshell, 1);
  bool dense = false;
  if (!one && rows4 <= rows_cap) {
    const size_t tile = std::max((size_t)80 * rows4, part_rn) * sizeof(double) + 8;
    if (!h->S.pbc) dense = fixed1 + tile <= budget;
    else if (fixed1 + tile < budget) {
      const int cand = pick_icap(budget - fixed1 - tile);
      if (cand > 0) { dense = true; icap = cand; pbc_b = res_lds_pbc(h->natom, h->S.nL, cand, h->nshell, h->twist) + 8; }
    }
  }
  if (h->twist && !dense) return 0;
  const int kt_cap = (one || dense) ? (dense ? rows4 : rows_all) : std::min(rows_cap, (int)(((avail - part_rn) / 80) & ~(size_t)3));
  if (kt_cap < 20) return 0;
  ResTab RT{};
  h->res_dense = dense;
  h->res_rows4 = rows4;
  if (dense) {
    for (int s = 0; s < 2; ++s) {
      if (h->nmo[s] == 0) continue;
      std::vector<double> mo((size_t)h->nao * h->nmo[s]);
      HIPCHK(hipMemcpy(mo.data(), h->d_mo[s], mo.size() * sizeof(double), hipMemcpyDeviceToHost));
      if (!h->d_cres[s]) TRY(upload_table<double>(h, nullptr, (size_t)res_rows_alloc(rows4) * 16 * h->nt[s], &h->d_cres[s]));
      TRY(res_refresh_coeff(h, s, mo.data()));
    }
  }
  int ch = 0, kt = 0;
  std::vector<int> pass_of_chunk((size_t)nch, 0);
  if (dense) {
    RT.npass = 1; RT.pass_row0[0] = 0; kt = rows4; ch = nch;
  }
  while (ch < nch) {  // greedy: consecutive chunks while their padded rows fit the tile
    if (RT.npass == PQA_RES_MAXPASS) return 0;
    const int base = c.row0[ch];
    int end = ch;
    while (end < nch && c.row0[end] + ((c.nk[end] + 3) & ~3) - base <= kt_cap) ++end;
    if (end == ch) return 0;
    for (int q = ch; q < end; ++q) pass_of_chunk[q] = RT.npass;
    RT.pass_row0[RT.npass] = base;
    kt = std::max(kt, c.row0[end - 1] + ((c.nk[end - 1] + 3) & ~3) - base);
    ch = end;
    ++RT.npass;
  }
  RT.pass_row0[RT.npass] = dense ? rows4 : c.rows_pad;
  RT.kt = kt;
  RT.part_off = (RT.npass == 1) ? 0 : 80 * kt;
  RT.region = (RT.npass == 1) ? (int)std::max((size_t)80 * kt, part_rn) : (int)((size_t)80 * kt + part_rn);
  // shell lists: per pass the shells by descending phase-1 cost, dealt to the 32 lane groups in snake order — neighbours in cost
  // (the same kind of shell) land in neighbouring groups, i.e. in one wave, and the groups' totals stay balanced
  std::vector<int> off(1, 0), list, srow((size_t)h->nshell, 0);
  for (int sh = 0; sh < h->nshell; ++sh) srow[sh] = dense ? h->shell_ao[sh] : c.row0[c.shell_chunk[sh]] + c.shell_kb[sh];
  for (int p = 0; p < RT.npass; ++p) {
    std::vector<int> mem;
    for (int sh = 0; sh < h->nshell; ++sh)
      if (pass_of_chunk[c.shell_chunk[sh]] == p) mem.push_back(sh);
    std::stable_sort(mem.begin(), mem.end(), [&](int a, int b) {
      if (h->shell_cost[a] != h->shell_cost[b]) return h->shell_cost[a] > h->shell_cost[b];
      return h->shell_l[a] > h->shell_l[b];
    });
    std::vector<std::vector<int>> grp(PQA_RES_G);
    for (size_t k = 0; k < mem.size(); ++k) {
      const int round = (int)(k / PQA_RES_G), pos = (int)(k % PQA_RES_G);
      grp[(round & 1) ? PQA_RES_G - 1 - pos : pos].push_back(mem[k]);
    }
    for (int g = 0; g < PQA_RES_G; ++g) {
      for (int sh : grp[g]) list.push_back(sh);
      off.push_back((int)list.size());
    }
  }
  RT.nlist = (int)list.size();
  int* tmp_i = nullptr;
  TRY(upload_table(h, off.data(), off.size(), &tmp_i)); RT.grp_off = tmp_i;
  TRY(upload_table(h, list.data(), list.size(), &tmp_i)); RT.grp_shell = tmp_i;
  TRY(upload_table(h, srow.data(), srow.size(), &tmp_i)); RT.shell_row = tmp_i;
  TRY(upload_table(h, q0_u.data(), q0_u.size(), &tmp_i)); RT.shell_q0 = tmp_i;
  double* tmp_d = nullptr;
  TRY(upload_table(h, pe_u.data(), pe_u.size(), &tmp_d)); RT.prim_exp_u = tmp_d;
  TRY(upload_table(h, pc_u.data(), pc_u.size(), &tmp_d)); RT.prim_coef_u = tmp_d;
  RT.nprim_u = nprim_u;
  h->res_lds = (size_t)RT.region * sizeof(double) + res_lds_fixed(h->nshell, nprim_u, h->natom, h->na, RT.nlist, RT.npass);
  h->res_lds = (h->res_lds + 7) & ~(size_t)7;
  RT.pbc_off = (int)h->res_lds; RT.icap = icap;
  RT.twist = h->twist ? 1 : 0; RT.im_off = h->nao;
  if (const char* e = getenv("PQA_RES_ICAP")) RT.icap = std::max(1, std::min(icap, atoi(e)));  // (tests: short lists, the pairs that overflow walk the masks)
  h->res_lds += pbc_b;
  if (h->res_lds > 160 * 1024) return 0;
  HIPCHK(hipFuncSetAttribute((const void*)k_sweep_res<false, 2>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
  HIPCHK(hipFuncSetAttribute((const void*)k_sweep_res<true, 2>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
  HIPCHK(hipFuncSetAttribute((const void*)k_sweep_res<false, 3>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
  HIPCHK(hipFuncSetAttribute((const void*)k_sweep_res<true, 3>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
  HIPCHK(hipFuncSetAttribute((const void*)k_sweep_res<false, 2, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
  HIPCHK(hipFuncSetAttribute((const void*)k_sweep_res<true, 2, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
  HIPCHK(hipFuncSetAttribute((const void*)k_sweep_res<false, 3, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
  HIPCHK(hipFuncSetAttribute((const void*)k_sweep_res<true, 3, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
  HIPCHK(hipFuncSetAttribute((const void*)k_sweep_res<false, 3, true, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
  HIPCHK(hipFuncSetAttribute((const void*)k_sweep_res<true, 3, true, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
  HIPCHK(hipFuncSetAttribute((const void*)k_sweep_res<false, 2, true, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
  HIPCHK(hipFuncSetAttribute((const void*)k_sweep_res<true, 2, true, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
  if (const char* dbg = getenv("PQA_RES_DEBUG"); dbg && atoi(dbg) > 1) {  // the lane groups' shell lists with the cost model's figures
    for (int g = 0; g < PQA_RES_G * RT.npass; ++g) {
      fprintf(stderr, "[pqa_res] group %2d:", g);
      for (int k = off[g]; k < off[g + 1]; ++k) fprintf(stderr, " sh %d (l %d, np %d, cost %d)", list[k], h->shell_l[list[k]], h->shell_np[list[k]], h->shell_cost[list[k]]);
      fprintf(stderr, "\n");
    }
  }
  if (getenv("PQA_RES_DEBUG")) fprintf(stderr, "[pqa_res] passes %d, tile rows %d (padded basis %d), LDS %zu B, image-list capacity %d\n", RT.npass, RT.kt, c.rows_pad, h->res_lds, RT.icap);
  h->res_tab = RT;
  h->res_ok = true;
  return 0;
}

bool res_eligible(pqa_handle* h, long W) {
  if (!h->res_ready) {
    if (res_setup(h) != 0) { h->res_ok = false; h->err.clear(); }
  }
  if (!h->res_ok) return false;
  if (h->res_mode > 0) return true;
  // automatic: measured against the launch-per-move sweep (gpurun_out/res_scan.jsonl, round 5; sweep only): (H2O)8 1.66x at 512
  // walkers, 1.88x at 4096, 1.38x at 16384, 1.06x at 32768, 1.02x at 49152, 0.97x at 65536; H2O (8 electrons: a walker's 32 lanes
  // are mostly idle) 1.2x up to 4096 walkers, 0.49x at 16384.  One round of blocks (16 walkers per CU) always wins.
  if (W < h->res_min || W > h->res_max) return false;
  // periodic cells (lattice-summed AO phase in the block, round 5; 2x2x2 diamond cell, sweep alone, launches -> resident): 4.10 -> 2.71 ms at
  // 2048 walkers, 4.43 -> 2.74 at 4096, 7.56 -> 5.48 at 8192, 12.5 -> 10.9 at 16384, 17.2 -> 16.4 at 24576, even at 32768
  // (after the image lists were dealt to several threads per pair and the lattice sums went to ds_add_f64: DMC step 13.6 -> 12.1 ms at 4096,
  // 43.3 -> 41.4 at 16384, 82.8 -> 82.1 at 32768)
  // complex determinants (twisted 8-atom cell, VMC step with energy, launches -> resident): 4.29 -> 2.90 ms at 4096 walkers, 6.69 -> 5.33 at 8192,
  // 8.57 -> 7.76 at 12288, 10.8 -> 10.2 at 16384
  // (with the instantiation for bases without f shells: 15.3 -> 14.5 ms at 24 576 walkers, 19.7 -> 19.3 at 32 768)
  if (h->cplx) return W <= 32768;
  if (h->S.pbc) return W <= 32768;
  return W <= 4096 || (std::max(h->nup, h->ndn) >= 16 && W <= 49152);
}

// One sweep over all electrons of walkers [0, W): a single launch.  mb carries both tapes (the caller drew them if there were none).
int sweep_res(pqa_handle* h, const MoveBuf& mb) {
  if (!mb.gauss || !mb.unif) FAIL("resident sweep: the random-number tapes are missing");
  const long W = h->W;
  const LwState L = lw_state(h);
  ChunkTab Tc = h->tab[0];
  if (h->res_dense) { Tc.cpad[0] = h->d_cres[0]; Tc.cpad[1] = h->d_cres[1]; }
  const dim3 grid((unsigned)((W + PQA_RES_NW - 1) / PQA_RES_NW)), block(PQA_RES_NT);
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
    h->prof_pc += (double)W * h->N * 5;  // point-components of this launch (as launch_orb counts them)
  }
#define PQA_RES_LAUNCH(D, LM) hipLaunchKernelGGL((k_sweep_res<D, LM>), grid, block, h->res_lds, h->stream, h->S, L, mb, Tc, h->res_tab, (int)h->has_jastrow, W, 0L, W)
#define PQA_RES_LAUNCH_P(D, LM) hipLaunchKernelGGL((k_sweep_res<D, LM, true>), grid, block, h->res_lds, h->stream, h->S, L, mb, Tc, h->res_tab, (int)h->has_jastrow, W, 0L, W)
#define PQA_RES_LAUNCH_C(D, LM) hipLaunchKernelGGL((k_sweep_res<D, LM, true, true>), grid, block, h->res_lds, h->stream, h->S, L, mb, Tc, h->res_tab, (int)h->has_jastrow, W, 0L, W)
  if (h->cplx) {
    if (mb.dmc) { if (h->res_lmax <= 2) PQA_RES_LAUNCH_C(true, 2); else PQA_RES_LAUNCH_C(true, 3); }
    else { if (h->res_lmax <= 2) PQA_RES_LAUNCH_C(false, 2); else PQA_RES_LAUNCH_C(false, 3); }
  } else
#undef PQA_RES_LAUNCH_C
  if (h->S.pbc) {  // (s, p, d shells: 25 running sums of a shell's lattice sum in registers; with f shells 35)
    if (mb.dmc) { if (h->res_lmax <= 2) PQA_RES_LAUNCH_P(true, 2); else PQA_RES_LAUNCH_P(true, 3); }
    else { if (h->res_lmax <= 2) PQA_RES_LAUNCH_P(false, 2); else PQA_RES_LAUNCH_P(false, 3); }
  } else
#undef PQA_RES_LAUNCH_P
  if (mb.dmc) { if (h->res_lmax <= 2) PQA_RES_LAUNCH(true, 2); else PQA_RES_LAUNCH(true, 3); }
  else { if (h->res_lmax <= 2) PQA_RES_LAUNCH(false, 2); else PQA_RES_LAUNCH(false, 3); }
#undef PQA_RES_LAUNCH
  if (e1) HIPCHK(hipEventRecord(e1, h->stream));
  return check_launch(h, "k_sweep_res");
}

#ifdef PQA_RES_CLK  // timing build only
extern "C" int pqa_debug_res_clk3(unsigned long long* dst, int n) {
  return (int)hipMemcpyFromSymbol(dst, HIP_SYMBOL(pqa_res_clk3), (size_t)n * sizeof(unsigned long long));
}
extern "C" int pqa_debug_res_clk2(unsigned long long* dst, int n) {
  return (int)hipMemcpyFromSymbol(dst, HIP_SYMBOL(pqa_res_clk2), (size_t)n * sizeof(unsigned long long));
}
extern "C" int pqa_debug_res_clk(unsigned long long* dst, int n) {
  return (int)hipMemcpyFromSymbol(dst, HIP_SYMBOL(pqa_res_clk), (size_t)n * sizeof(unsigned long long));
}
#endif
