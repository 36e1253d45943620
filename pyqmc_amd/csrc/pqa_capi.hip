// pyqmc_amd C ABI implementation (host side).  See include/pyqmc_amd.h for the contract.
// Single translation unit: the device code lives in the headers included below.
#include "pqa_internal.hpp"

static thread_local std::string g_create_error;


// ---------------------------------------------------------------- membership masks (k_pbc_prepass)
// The reference's image-membership rule asks, for candidate image j of an atom, whether member[class][b + img_n[j]] is set,
// b being the membership base of the (point, atom) pair (pbc_ctx_base).  Tabulated here for every base in an extended grid
// (side + 2 E per axis, E = side >= every |img_n|) as a 128-bit mask over the candidates, so that the pre-pass replaces ~10
// four-load tests per thread by one 16-byte look-up.  315 KB per atom class for M = 4.
static int member_masks(pqa_handle* h, const pqa_system_t* sys, PbcDev& P) {
  P.memb_mask = nullptr;
  P.memb_E = 0;
  const int side = 2 * sys->member_M + 1, E = side, T = side + 2 * E, nc = sys->n_member_class, nj = std::min(sys->nL, 128);
  if ((size_t)nc * T * T * T * 16 > ((size_t)64 << 20)) return 0;  // (absurdly large rule: candidate-by-candidate tests)
  std::vector<unsigned char> mem((size_t)nc * side * side * side);
  std::vector<int> imgn((size_t)sys->nL * 3);
  HIPCHK(hipMemcpy(mem.data(), sys->member, mem.size(), hipMemcpyDefault));
  HIPCHK(hipMemcpy(imgn.data(), sys->img_n, imgn.size() * sizeof(int), hipMemcpyDefault));
  for (int j = 0; j < nj; ++j)
    for (int c = 0; c < 3; ++c)
      if (std::abs(imgn[3 * j + c]) > E) return 0;  // a base outside the grid could still reach a member: no table
  std::vector<unsigned long long> mask((size_t)nc * T * T * T * 2, 0ull);
  for (int cl = 0; cl < nc; ++cl)
    for (int i0 = 0; i0 < T; ++i0)
      for (int i1 = 0; i1 < T; ++i1)
        for (int i2 = 0; i2 < T; ++i2) {
          unsigned long long* m = &mask[2 * ((((size_t)cl * T + i0) * T + i1) * T + i2)];
          for (int j = 0; j < nj; ++j) {
            const int n0 = i0 - E + imgn[3 * j], n1 = i1 - E + imgn[3 * j + 1], n2 = i2 - E + imgn[3 * j + 2];
            if (n0 < 0 || n0 >= side || n1 < 0 || n1 >= side || n2 < 0 || n2 >= side) continue;
            if (mem[(((size_t)cl * side + n0) * side + n1) * side + n2]) m[j >> 6] |= 1ull << (j & 63);
          }
        }
  unsigned long long* d = nullptr;
  TRY(upload_table(h, mask.data(), mask.size(), &d));
  P.memb_mask = d;
  P.memb_E = E;
  return 0;
}

// ---------------------------------------------------------------- near-candidate masks (k_pbc_prepass)
// The pre-pass folds point - atom into the cell-centred parallelepiped and then needs the lattice vectors L_j with
// |d - L_j|^2 <= atom_cut.  The candidate list (num_Ls[atom] vectors, 79 in the 2x2x2 diamond cell) is what ANY point of the
// cell may need; a given point needs about a sixth of it.  Tabulated here, for every sub-cell of a G^3 grid over the fractional
// coordinates [-1/2, 1/2)^3, the candidates whose distance to the sub-cell's centre is at most sqrt(atom_cut) + the sub-cell's
// half diagonal (padded): a superset of what any point inside it can admit.  16 bytes per (atom, sub-cell).
static int near_masks(pqa_handle* h, const pqa_system_t* sys, const std::vector<int>& nl, const std::vector<double>& ac, PbcDev& P) {
  P.near_mask = nullptr;
  P.near_G = 0;
  int G = (size_t)h->natom * 16 * 16 * 16 * 16 <= ((size_t)2 << 20) ? 16 : 8;  // (the table should stay in an XCD's L2)
  if (const char* e = getenv("PQA_PRE_GRID")) G = std::max(0, std::min(16, atoi(e)));
  while (G > 1 && (size_t)h->natom * G * G * G * 16 > ((size_t)64 << 20)) G /= 2;
  if (G < 2) return 0;
  const int nj = std::min(sys->nL, 128);
  std::vector<double> ls((size_t)nj * 3);
  HIPCHK(hipMemcpy(ls.data(), sys->Ls, ls.size() * sizeof(double), hipMemcpyDefault));
  const double* a = sys->lattice;
  const double hw = 0.5 / G + 1e-6;  // half width of a sub-cell in fractional coordinates, padded for the rounding of the fold
  double rho = 0.0;
  for (int sg = 0; sg < 4; ++sg) {   // half of the longest body diagonal
    const double s1 = (sg & 1) ? -hw : hw, s2 = (sg & 2) ? -hw : hw;
    double d2 = 0.0;
    for (int c = 0; c < 3; ++c) { const double v = hw * a[c] + s1 * a[3 + c] + s2 * a[6 + c]; d2 += v * v; }
    rho = std::max(rho, std::sqrt(d2));
  }
  std::vector<unsigned long long> mask((size_t)h->natom * G * G * G * 2, 0ull);
  for (int ia = 0; ia < h->natom; ++ia) {
    const double reach = std::sqrt(std::max(ac[ia], 0.0)) * (1.0 + 1e-9) + rho + 1e-9;
    const int n = std::min(nl[ia], nj);
    for (int g0 = 0; g0 < G; ++g0)
      for (int g1 = 0; g1 < G; ++g1)
        for (int g2 = 0; g2 < G; ++g2) {
          const double f[3] = {(g0 + 0.5) / G - 0.5, (g1 + 0.5) / G - 0.5, (g2 + 0.5) / G - 0.5};
          double ctr[3];
          for (int c = 0; c < 3; ++c) ctr[c] = f[0] * a[c] + f[1] * a[3 + c] + f[2] * a[6 + c];
          unsigned long long* m = &mask[2 * ((((size_t)ia * G + g0) * G + g1) * G + g2)];
          for (int j = 0; j < n; ++j) {
            const double dx = ctr[0] - ls[3 * j], dy = ctr[1] - ls[3 * j + 1], dz = ctr[2] - ls[3 * j + 2];
            if (dx * dx + dy * dy + dz * dz <= reach * reach) m[j >> 6] |= 1ull << (j & 63);
          }
        }
  }
  unsigned long long* d = nullptr;
  TRY(upload_table(h, mask.data(), mask.size(), &d));
  P.near_mask = d;
  P.near_G = G;
  return 0;
}

// ---------------------------------------------------------------- Voronoi-relevant lattice vectors (min_image)
// v is relevant iff v/2 is strictly closer to 0 (and v) than to every other lattice point.  Candidates: coefficients in
// {-2..2}^3 (all relevant vectors of any cell that is not absurdly skewed), tested against the points with coefficients in
// {-4..4}^3.  One of each +- pair; three-dimensional lattices have at most 7 pairs.
static int voronoi_vectors(const double* a, PbcDev& P) {
  P.nvor = 0;
  for (int q = 0; q < 7; ++q) { P.vor[q][0] = P.vor[q][1] = P.vor[q][2] = 0.0; P.vorh[q] = 1.0; }  // padding: never violated
  auto vec = [&](int i, int j, int k, double* v) {
    for (int c = 0; c < 3; ++c) v[c] = i * a[c] + j * a[3 + c] + k * a[6 + c];
  };
  for (int i = -2; i <= 2; ++i)
    for (int j = -2; j <= 2; ++j)
      for (int k = -2; k <= 2; ++k) {
        if (i < 0 || (i == 0 && (j < 0 || (j == 0 && k <= 0)))) continue;  // one of each pair, not the origin
        double v[3];
        vec(i, j, k, v);
        const double half = 0.5 * std::sqrt(v[0] * v[0] + v[1] * v[1] + v[2] * v[2]);
        bool relevant = true;
        for (int p = -4; p <= 4 && relevant; ++p)
          for (int q = -4; q <= 4 && relevant; ++q)
            for (int r = -4; r <= 4; ++r) {
              if ((p == 0 && q == 0 && r == 0) || (p == i && q == j && r == k)) continue;
              double u[3];
              vec(p, q, r, u);
              const double d = std::sqrt((0.5 * v[0] - u[0]) * (0.5 * v[0] - u[0]) + (0.5 * v[1] - u[1]) * (0.5 * v[1] - u[1]) +
                                         (0.5 * v[2] - u[2]) * (0.5 * v[2] - u[2]));
              if (d <= half * (1.0 + 1e-9)) { relevant = false; break; }
            }
        if (!relevant) continue;
        if (P.nvor >= 7) return 1;
        for (int c = 0; c < 3; ++c) P.vor[P.nvor][c] = v[c];
        P.vorh[P.nvor] = 2.0 * half * half;  // |v|^2 / 2
        ++P.nvor;
      }
  return P.nvor >= 3 ? 0 : 1;
}

// ---------------------------------------------------------------- phase-1 cost model of the shells
// One evaluation of a shell costs a radial part per primitive and an angular part / tile stores per function.  An open
// system evaluates every shell once per point.  A periodic one evaluates it once per image inside the SHELL's cut-off — the
// wave walks max-over-lanes of that count, about 1.4 x the mean V_sphere(shell_cut) / V_cell plus one — and from the second
// (farther) image on only the primitives that survive the screening at half the shortest lattice vector are evaluated.
// Packing the lane groups with the per-evaluation cost alone gave a group holding two diffuse p shells (13 images each in
// the 2x2x2 diamond cell) 2.3 x the average load, and a barrier ends every chunk.
// Radial tables of the contracted shells (SysDev::rtab, radial_tab in pqa_ao.hpp): per distinct (exponent, coefficient) sequence of at least
// PQA_RT_MINP primitives the sum R(x) = sum_p c_p exp(-a_p x) as degree-9 polynomials on the intervals y = x + 2^-7 in
// 2^(o-7) [1 + j/8, 1 + (j+1)/8), o = 0 .. until every primitive is below exp(-46).  Chebyshev interpolation at the 10 nodes of every interval
// in long double, converted to powers of the local variable u in [-1, 1]; the largest error found at 33 points per interval (in the device's
// arithmetic: double Horner) is kept in h->rt_err, relative to sum_p |c_p| (1e-15 for cc-pVDZ-shaped contractions; pqa_debug_radtab_err, and
// the device tests compare the orbitals with the primitive sums).  Open systems only (the lattice sums keep their exponentials), value-only
// orbital kernel only; PQA_RADTAB=0 turns it off.
static int build_radial_tables(pqa_handle* h, const pqa_system_t* sys, SysDev& S) {
  std::vector<int> rt((size_t)2 * sys->nshell, -1);
  std::vector<double> tab;
  h->rt_err = 0.0;
  const char* env = getenv("PQA_RADTAB");
  const bool on = !(env && atoi(env) == 0) && sys->pbc == 0;
  std::vector<double> pe_((size_t)std::max(sys->nprim, 1)), pc_((size_t)std::max(sys->nprim, 1));
  std::vector<int> po_((size_t)sys->nshell + 1);
  HIPCHK(hipMemcpy(pe_.data(), sys->prim_exp, (size_t)sys->nprim * sizeof(double), hipMemcpyDefault));
  HIPCHK(hipMemcpy(pc_.data(), sys->prim_coef, (size_t)sys->nprim * sizeof(double), hipMemcpyDefault));
  HIPCHK(hipMemcpy(po_.data(), sys->shell_prim_off, po_.size() * sizeof(int), hipMemcpyDefault));
  const double* prim_exp = pe_.data();
  const double* prim_coef = pc_.data();
  const int* shell_prim_off = po_.data();
  if (on) {
    constexpr int n = PQA_RT_DEG + 1;
    long double nodes[n], Tm[n][n];  // Chebyshev nodes; T_q(u) in powers of u
    const long double pi = acosl(-1.0L);
    for (int k = 0; k < n; ++k) nodes[k] = cosl(pi * (k + 0.5L) / n);
    for (int q = 0; q < n; ++q)
      for (int d = 0; d < n; ++d) Tm[q][d] = 0.0L;
    Tm[0][0] = 1.0L; Tm[1][1] = 1.0L;
    for (int q = 2; q < n; ++q)
      for (int d = 0; d < n; ++d) Tm[q][d] = (d > 0 ? 2.0L * Tm[q - 1][d - 1] : 0.0L) - Tm[q - 2][d];
    for (int sh = 0; sh < sys->nshell; ++sh) {
      const int p0 = shell_prim_off[sh], np = shell_prim_off[sh + 1] - p0;
      if (np < PQA_RT_MINP) continue;
      int same = -1;
      for (int prev = 0; prev < sh && same < 0; ++prev) {
        const int q0 = shell_prim_off[prev];
        if (shell_prim_off[prev + 1] - q0 == np && rt[2 * prev] >= 0 && std::equal(prim_exp + p0, prim_exp + p0 + np, prim_exp + q0) &&
            std::equal(prim_coef + p0, prim_coef + p0 + np, prim_coef + q0)) same = prev;
      }
      if (same >= 0) { rt[2 * sh] = rt[2 * same]; rt[2 * sh + 1] = rt[2 * same + 1]; continue; }
      double amin = prim_exp[p0];
      for (int p = 0; p < np; ++p) amin = std::min(amin, prim_exp[p0 + p]);
      if (!(amin > 0.0)) continue;
      const int noct = std::max(1, (int)std::ceil(std::log2((46.0 / amin + PQA_RT_X0) / PQA_RT_X0)));
      if (noct > 40) continue;
      const int nint = noct * PQA_RT_NSUB;
      const size_t tab0 = tab.size();
      double err_sh = 0.0;
      rt[2 * sh] = (int)tab.size(); rt[2 * sh + 1] = nint;
      long double scale = 0.0L;
      for (int p = 0; p < np; ++p) scale += fabsl((long double)prim_coef[p0 + p]);
      auto F = [&](long double x) {
        long double f = 0.0L;
        for (int p = 0; p < np; ++p) f += (long double)prim_coef[p0 + p] * expl(-(long double)prim_exp[p0 + p] * x);
        return f;
      };
      for (int o = 0; o < noct; ++o)
        for (int j = 0; j < PQA_RT_NSUB; ++j) {
          const long double ylo = (long double)PQA_RT_X0 * ldexpl(1.0L, o) * (1.0L + (long double)j / PQA_RT_NSUB);
          const long double yhi = (long double)PQA_RT_X0 * ldexpl(1.0L, o) * (1.0L + (long double)(j + 1) / PQA_RT_NSUB);
          const long double xc = 0.5L * (ylo + yhi) - (long double)PQA_RT_X0, hw = 0.5L * (yhi - ylo);
          {
            long double fv[n], cc[n], mono[n];
            for (int q = 0; q < n; ++q) fv[q] = F(xc + hw * nodes[q]);
            for (int q = 0; q < n; ++q) {
              long double sum = 0.0L;
              for (int m = 0; m < n; ++m) sum += fv[m] * cosl(q * pi * (m + 0.5L) / n);
              cc[q] = (q == 0 ? 1.0L : 2.0L) * sum / n;
            }
            for (int d = 0; d < n; ++d) { mono[d] = 0.0L; for (int q = 0; q < n; ++q) mono[d] += cc[q] * Tm[q][d]; }
            double m64[n];
            for (int d = 0; d < n; ++d) { m64[d] = (double)mono[d]; tab.push_back(m64[d]); }
            for (int t = 0; t <= 32; ++t) {  // the table against the sums, in the arithmetic the device uses (double Horner)
              const double u = -1.0 + t / 16.0;
              double pv = m64[n - 1];
              for (int d = n - 2; d >= 0; --d) pv = std::fma(pv, u, m64[d]);
              const long double ex = F(xc + hw * (long double)u);
              err_sh = std::max(err_sh, (double)(fabsl((long double)pv - ex) / scale));
            }
          }
        }
      // A table is kept only if it reproduces the primitive sum to rounding everywhere: the tight primitives of all-electron sets (exponent
      // 11 720 in cc-pVDZ oxygen: e^{-a x} falls by e^-11 across the first interval) are beyond a degree-9 fit — 1e-5 of sum |c| — and such
      // shells keep their exponentials.
      if (err_sh > PQA_RT_MAXERR) { tab.resize(tab0); rt[2 * sh] = -1; rt[2 * sh + 1] = 0; }
      else h->rt_err = std::max(h->rt_err, err_sh);
    }
  }
  double* td = nullptr; int* ti = nullptr;
  TRY(upload_table(h, tab.data(), tab.size(), &td)); S.rtab = td;
  TRY(upload_table(h, rt.data(), rt.size(), &ti)); S.shell_rt = ti;
  h->rt_shells.assign(rt.begin(), rt.end());
  if (getenv("PQA_RES_DEBUG")) fprintf(stderr, "[pqa] radial tables: %zu doubles, largest error %.2e of sum |c|\n", tab.size(), h->rt_err);
  return 0;
}

static void shell_costs(pqa_handle* h, const pqa_system_t* sys) {
  const int tw = h->twist ? 2 : 1;
  h->shell_cost.assign((size_t)h->nshell, 0);
  std::vector<double> scut, pexp;
  double vol = 0.0, half2 = 0.0;
  if (sys->pbc && sys->nL > 0 && sys->shell_cut) {
    scut.resize((size_t)h->nshell);
    hipMemcpy(scut.data(), sys->shell_cut, scut.size() * sizeof(double), hipMemcpyDefault);
    pexp.resize((size_t)sys->nprim);
    hipMemcpy(pexp.data(), sys->prim_exp, pexp.size() * sizeof(double), hipMemcpyDefault);
    const double* a = sys->lattice;
    vol = fabs(a[0] * (a[4] * a[8] - a[5] * a[7]) - a[1] * (a[3] * a[8] - a[5] * a[6]) + a[2] * (a[3] * a[7] - a[4] * a[6]));
    half2 = 1e300;
    for (int i = 0; i < 3; ++i) half2 = std::min(half2, 0.25 * (a[3 * i] * a[3 * i] + a[3 * i + 1] * a[3 * i + 1] + a[3 * i + 2] * a[3 * i + 2]));
  }
  std::vector<int> poff((size_t)h->nshell + 1);
  hipMemcpy(poff.data(), sys->shell_prim_off, poff.size() * sizeof(int), hipMemcpyDefault);
  for (int s = 0; s < h->nshell; ++s) {
    const int ang = 25 * tw * (2 * h->shell_l[s] + 1) + 40;
    if (scut.empty() || !(vol > 0.0)) { h->shell_cost[s] = 45 * h->shell_np[s] + ang; continue; }
    const double mean = 4.18879020478639 * scut[s] * std::sqrt(scut[s]) / vol;
    const double iters = std::max(1.0, 1.4 * mean + 1.0);
    int far = 0;  // primitives still evaluated beyond the nearest image
    for (int q = poff[s]; q < poff[s + 1]; ++q) far += (pexp[q] * half2 <= 50.0) ? 1 : 0;
    h->shell_cost[s] = (int)((45 * h->shell_np[s] + ang + 30) + (iters - 1.0) * (45 * far + ang + 30));
  }
}

// ---------------------------------------------------------------- chunk tables for k_orb
static void build_chunks(const pqa_handle* h, int KC, ChunkHost& c) {
  c = ChunkHost();
  for (int g = 0; g < 3; ++g) c.cw_off[g].push_back(0);
  // twisted cells: a shell's complex lattice sum occupies 2 (2l+1) tile rows, real parts then imaginary parts, in ONE
  // chunk, so that a single walk over the images fills both (evaluating the parts as two separate shells doubled the
  // exp work).  shell_kb / shell_chunk keep an entry sh + nshell for the imaginary rows (coefficient upload).
  const int nsx = h->nshell, tw = h->twist ? 2 : 1;
  c.shell_kb.assign((size_t)tw * h->nshell, 0);
  c.shell_chunk.assign((size_t)tw * h->nshell, 0);
  // phase-1 cost of a shell: radial part per primitive + angular part / tile stores per function
  auto cost = [&](int s) { return h->shell_cost[s]; };
  auto nfun = [&](int s) { return tw * (2 * h->shell_l[s] + 1); };
  int nao = 0;
  for (int s = 0; s < nsx; ++s) nao += nfun(s);
  // Longest-processing-time packing over (chunk, group) slots under the chunk's row capacity; if a shell does not
  // fit anywhere a chunk is added.  4 groups per chunk (64-point tiles) is the layout that is balanced; the
  // 8-group lists (32-point tiles) are a second LPT inside each chunk.
  std::vector<int> order((size_t)nsx);
  for (int s = 0; s < nsx; ++s) order[s] = s;
  std::stable_sort(order.begin(), order.end(), [&](int a, int b) { return cost(a) > cost(b); });
  int nchunk = std::max((nao + KC - 1) / KC, 1);
  std::vector<std::vector<int>> load;
  std::vector<int> rows;
  std::vector<std::vector<std::vector<int>>> slot;  // [chunk][group] -> shells
  for (;;) {
    load.assign((size_t)nchunk, std::vector<int>(4, 0));
    rows.assign((size_t)nchunk, 0);
    slot.assign((size_t)nchunk, std::vector<std::vector<int>>(4));
    bool ok = true;
    for (int s : order) {
      int bc = -1, bg = -1;
      for (int ch = 0; ch < nchunk; ++ch) {
        if (rows[ch] + nfun(s) > KC) continue;
        for (int g = 0; g < 4; ++g)
          if (bc < 0 || load[ch][g] < load[bc][bg]) { bc = ch; bg = g; }
      }
      if (bc < 0) { ok = false; break; }
      slot[bc][bg].push_back(s);
      load[bc][bg] += cost(s);
      rows[bc] += nfun(s);
    }
    if (ok) break;
    ++nchunk;
  }
  int row0 = 0;
  for (int ch = 0; ch < nchunk; ++ch) {
    if (rows[ch] == 0) continue;  // (possible after a capacity retry)
    const int ci = (int)c.nk.size();
    c.nk.push_back(rows[ch]);
    c.row0.push_back(row0);
    row0 += (rows[ch] + 3) & ~3;
    int kb = 0;
    std::vector<int> members;
    for (int g = 0; g < 4; ++g)
      for (int s : slot[ch][g]) {
        c.shell_kb[s] = kb; c.shell_chunk[s] = ci;
        if (tw == 2) { c.shell_kb[s + h->nshell] = kb + nfun(s) / 2; c.shell_chunk[s + h->nshell] = ci; }
        kb += nfun(s);
        members.push_back(s);
      }
    for (int g = 0; g < 4; ++g) {
      for (int s : slot[ch][g]) c.cw_shell[0].push_back(s);
      c.cw_off[0].push_back((int)c.cw_shell[0].size());
    }
    std::stable_sort(members.begin(), members.end(), [&](int a, int b) { return cost(a) > cost(b); });
    std::vector<std::vector<int>> l8(8);
    std::vector<int> load8(8, 0);
    for (int s : members) {
      int best = 0;
      for (int q = 1; q < 8; ++q)
        if (load8[q] < load8[best]) best = q;
      l8[best].push_back(s);
      load8[best] += cost(s);
    }
    for (int q = 0; q < 8; ++q) {
      for (int s : l8[q]) c.cw_shell[1].push_back(s);
      c.cw_off[1].push_back((int)c.cw_shell[1].size());
    }
    std::vector<std::vector<int>> l16(16);  // 16-point tiles: 16 lane groups
    std::vector<int> load16(16, 0);
    for (int s : members) {
      int best = 0;
      for (int q = 1; q < 16; ++q)
        if (load16[q] < load16[best]) best = q;
      l16[best].push_back(s);
      load16[best] += cost(s);
    }
    for (int q = 0; q < 16; ++q) {
      for (int s : l16[q]) c.cw_shell[2].push_back(s);
      c.cw_off[2].push_back((int)c.cw_shell[2].size());
    }
  }
  c.rows_pad = row0;
}

// zero-padded coefficient matrix for one chunk table / spin
static int upload_cpad(pqa_handle* h, int t, int s, const double* mo_host) {
  const ChunkHost& c = h->chunks[t];
  const int ldc = 16 * h->nt[s], nmo = h->nmo[s];
  std::vector<double> pad((size_t)std::max(c.rows_pad, 1) * ldc, 0.0);
  for (int sh = 0; sh < h->nshell; ++sh)  // tile row (chunk, shell_kb + m)  <-  AO shell_ao[sh] + m
    for (int m = 0; m < 2 * h->shell_l[sh] + 1; ++m)
      for (int j = 0; j < nmo; ++j)
        pad[(size_t)(c.row0[c.shell_chunk[sh]] + c.shell_kb[sh] + m) * ldc + j] = mo_host[(size_t)(h->shell_ao[sh] + m) * nmo + j];
  if (h->twist) {  // rows of the imaginary AO parts: (i AO_im)(C_re + i C_im) = AO_im (-C_im + i C_re), columns [re | im]
    const int nr = nmo / 2;
    for (int sh = 0; sh < h->nshell; ++sh) {
      const int sx = sh + h->nshell;
      for (int m = 0; m < 2 * h->shell_l[sh] + 1; ++m)
        for (int j = 0; j < nr; ++j) {
          const double* src = mo_host + (size_t)(h->shell_ao[sh] + m) * nmo;
          double* dst = pad.data() + (size_t)(c.row0[c.shell_chunk[sx]] + c.shell_kb[sx] + m) * ldc;
          dst[j] = -src[nr + j];
          dst[nr + j] = src[j];
        }
    }
  }
  HIPCHK(hipMemcpy(h->d_cpad[t][s], pad.data(), pad.size() * sizeof(double), hipMemcpyHostToDevice));
  return 0;
}

static int set_mo(pqa_handle* h, int s, const double* mo_host) {
  if (h->nmo[s] == 0 || !mo_host) return 0;  // an empty spin channel (fully polarised systems): nothing to upload
  HIPCHK(hipMemcpy(h->d_mo[s], mo_host, (size_t)h->nao * std::max(h->nmo[s], 1) * sizeof(double), hipMemcpyHostToDevice));
  for (int t = 0; t < 2; ++t) TRY(upload_cpad(h, t, s, mo_host));
  TRY(res_refresh_coeff(h, s, mo_host));  // (the resident sweep's dense coefficient copy, if it keeps one)
  return 0;
}

// ---------------------------------------------------------------- create / destroy
extern "C" int pqa_device_count(void) {
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess) return 0;
  return n;
}

extern "C" const char* pqa_last_error(const pqa_handle_t* h) { return h ? h->err.c_str() : g_create_error.c_str(); }

static int jas_merge_tables(pqa_handle* h);
// ccoeff (natom,na3,na3,nb3,3) -> C = (c + c^T_kl)/2 (three_body_jastrow.py:94-96)
static int set_c3(pqa_handle* h, const double* c) {
  const int A = h->natom, na = h->na3, nb = h->nb3;
  std::vector<double> sym((size_t)A * na * na * nb * 3);
  for (int I = 0; I < A; ++I)
    for (int k = 0; k < na; ++k)
      for (int l = 0; l < na; ++l)
        for (int m = 0; m < nb * 3; ++m) {
          const size_t a = (((size_t)I * na + k) * na + l) * nb * 3 + m, b = (((size_t)I * na + l) * na + k) * nb * 3 + m;
          sym[a] = 0.5 * (c[a] + c[b]);
        }
  HIPCHK(hipMemcpy(h->d_c3, sym.data(), sym.size() * sizeof(double), hipMemcpyHostToDevice));
  return 0;
}

// The reference's quadrature grids (eval_ecp.py:278-336, generate_quadrature_grids), every rule in the reference's point order.
// Octahedral families from the 27 points of {-1,0,1}^3 (x slowest, z fastest: numpy's mgrid order) by their count of non-zero
// coordinates: OA (1, the axes), OB (2, / sqrt 2), OC (3, / sqrt 3), OD = the three cyclic column rolls of (+-1, +-1, +-3) / sqrt 11.
// Icosahedral families from polar angles: A the poles, B the ten points at atan 2 / pi - atan 2, C the twenty at c_1, c_2.
static int ecp_quadrature_offset(int naip) {
  switch (naip) { case 6: return 0; case 12: return 6; case 18: return 18; case 26: return 36; case 32: return 62; case 50: return 94; default: return -1; }
}
static void ecp_quadrature_tables(std::vector<double>& quad, std::vector<double>& quadw) {
  std::vector<std::array<double, 3>> O[4], I[3];
  for (int x = -1; x <= 1; ++x)
    for (int y = -1; y <= 1; ++y)
      for (int z = -1; z <= 1; ++z) {
        const int nz = (x != 0) + (y != 0) + (z != 0);
        if (nz == 0) continue;
        const double sc = nz == 1 ? 1.0 : std::sqrt((double)nz);
        O[nz - 1].push_back({x / sc, y / sc, z / sc});
      }
  {
    const double f = std::sqrt(3.0 / 11.0);
    std::vector<std::array<double, 3>> d1;
    for (auto& p : O[2]) d1.push_back({p[0] * f, p[1] * f, p[2] * f * 3.0});
    for (int roll = 0; roll < 3; ++roll)  // np.roll(d1, roll, axis=1): column j moves to column (j + roll) % 3
      for (auto& p : d1) {
        std::array<double, 3> q;
        for (int j = 0; j < 3; ++j) q[(j + roll) % 3] = p[j];
        O[3].push_back(q);
      }
  }
  {
    const double pi = std::acos(-1.0), b1 = std::atan(2.0), s5 = std::sqrt(5.0);
    const double c1 = std::acos((2.0 + s5) / std::sqrt(15.0 + 6.0 * s5)), c2 = std::acos(1.0 / std::sqrt(15.0 + 6.0 * s5));
    auto sph = [](double t, double p) { return std::array<double, 3>{std::sin(t) * std::cos(p), std::sin(t) * std::sin(p), std::cos(t)}; };
    I[0].push_back(sph(0.0, 0.0)); I[0].push_back(sph(pi, 0.0));
    for (int k = 0; k < 10; ++k) I[1].push_back(sph(k % 2 == 0 ? b1 : pi - b1, k * pi / 5.0));
    for (int k = 0; k < 10; ++k) I[2].push_back(sph(k % 2 == 0 ? pi - c1 : c1, k * pi / 5.0));
    for (int k = 0; k < 10; ++k) I[2].push_back(sph(k % 2 == 0 ? pi - c2 : c2, k * pi / 5.0));
  }
  auto emit = [&](const std::vector<std::array<double, 3>>* fam, int nfam, const double* w) {
    for (int f = 0; f < nfam; ++f)
      for (auto& p : fam[f]) { quad.insert(quad.end(), p.begin(), p.end()); quadw.push_back(w[f]); }
  };
  const double w6[] = {1.0 / 6}, w12[] = {1.0 / 12, 1.0 / 12}, w18[] = {1.0 / 30, 1.0 / 15}, w26[] = {1.0 / 21, 4.0 / 105, 27.0 / 840};
  const double w32[] = {5.0 / 168, 5.0 / 168, 27.0 / 840}, w50[] = {4.0 / 315, 64.0 / 2835, 27.0 / 1280, 14641.0 / 725760};
  emit(O, 1, w6); emit(I, 2, w12); emit(O, 2, w18); emit(O, 3, w26); emit(I, 3, w32); emit(O, 4, w50);
}

static int create_impl(pqa_handle* h, const pqa_system_t* sys) {
  HIPCHK(hipSetDevice(h->device));
  HIPCHK(hipStreamCreateWithFlags(&h->stream, hipStreamNonBlocking));
  HIPCHK(hipEventCreate(&h->ev0));
  HIPCHK(hipEventCreate(&h->ev1));
  // A/B switches for the schedule variants compared in DESIGN.md sections 3-4 (all default to the measured best;
  // none of them changes results beyond summation order, tests/test_gpu_parity.py cross-checks the pairs):
  //   PQA_ORB_TP 16|32|64 point tile of k_orb (periodic: pins the automatic choice), PQA_ORB_WS 0|1 wave-specialised k_orb,
  //   PQA_ORB_NOTAB 1 basis tables from global memory, PQA_ORB_KC5 16|32 AO rows per chunk of the periodic 5-component
  //   launch, PQA_ORB_NOSPLIT 1 / PQA_ORB_SPLIT_MAX n chunk loop of small periodic launches on one block,
  //   PQA_LW 0 wave-per-walker sweep | 1 lane-per-walker (default) | 2 walker-tile kernel, PQA_LW_KB k electrons per
  //   Sherman-Morrison block (0: update every row per move; default 4), PQA_LW_GM g thread groups per walker and PQA_LW_NW
  //   walkers per block of k_step_lw, PQA_ECP_WAVE 1 wave-per-walker ECP accumulation,
  //   PQA_PROF_STRIDE n event brackets on every n-th orbital launch when profiling is enabled,
  //   PQA_PRE_GRID g   sub-cells per axis of the near-candidate masks of the periodic pre-pass (default 16, 8 beyond 32 atoms; 0: every candidate tested)
  //   PQA_PRE_NCUT n   at least n shell cut-off classes in the pre-pass instantiation (5 or 10 are compiled; tests)
  //   PQA_PBC_NW n words (4 image indices each) per (point, atom) image list of the periodic pre-pass (default from the cell;
  //   1 forces the direct-test fallback: tests), PQA_WIDE_NTH 512 k_orb_wide with 512 threads in untwisted periodic cells,
  //   PQA_TM_PRE 0 T-move ratios by the wave-per-walker loop only;
  //   round 3 (each documented at its field above): PQA_STEP_PRE 0 k_step_lw for small shards too, PQA_DRAWS_MAX n walker count up to
  //   which a sweep's random numbers are drawn ahead, PQA_FLUSH_WB8_MAX n 8-walker flush blocks up to n walkers, PQA_ECP_LDS 0 /
  //   PQA_ECP_POINT_LW 0 first-generation ECP list passes / point kernel, PQA_ECP_ACC_WAVES 1|4 waves per walker in the
  //   wave-per-walker energy kernels, PQA_ECP_ATOM_MAJOR 0 walker-major ECP lists in periodic cells, PQA_JAS_FOLD 0 Voronoi
  //   reduction in every periodic Jastrow pair;
  //   round 4: PQA_STEP_GW 16|32|64 thread groups per walker of k_step_pre, PQA_STEP_PRE_MAX n largest shard that runs it (8192),
  //   PQA_SPLIT 0..3 / PQA_SPLIT_MIN / PQA_SPLIT_CUS pipelined half-ensembles, PQA_JPRE 1 Jastrow sums ahead on a side stream,
  //   PQA_JAS_MERGE 0 Pade functions one by one instead of the merged rational function (jas_merge_tables),
  //   PQA_ORB_KC5 / PQA_ORB_KC1 16|32 AO rows per chunk of the periodic 5-component / value-only orbital launches.
  //   round 5: PQA_RES 0|1 resident sweep off / forced (default: by shard size, pqa_res.hip res_eligible), PQA_RES_MIN / PQA_RES_MAX walker
  //   window of the automatic choice, PQA_RES_PBC 0 periodic handles keep the launch-per-move sweep, PQA_RES_ICAP n shorter image lists in
  //   the periodic resident sweep (tests), PQA_RES_DEBUG 1 prints the tile / LDS plan, PQA_ORB_GENERAL 1 orbitals of handles beyond 64 per
  //   spin by k_ao + k_mo_rows instead of the windowed k_orb, PQA_RES_CX 0 complex determinants keep the launch-per-move sweep, PQA_WW 0|1|3
  //   wave-per-walker sweep in one launch off / forced with one / three waves per walker (default: one wave up to 4096 walkers), PQA_ECP_DEFER 0
  //   the ECP point totals are read back at every evaluation, PQA_EN_OVERLAP 0 the kinetic pass of small shards stays in line.
  if (const char* tp = getenv("PQA_ORB_TP")) h->orb_tp = atoi(tp);
  if (const char* ns = getenv("PQA_ORB_NOSPLIT")) h->orb_nosplit = atoi(ns);
  if (const char* sm = getenv("PQA_ORB_SPLIT_MAX")) h->orb_split_max = atol(sm);
  if (const char* kc = getenv("PQA_ORB_KC5")) h->orb_kc5 = atoi(kc);
  if (const char* kc = getenv("PQA_ORB_KC1")) h->orb_kc1 = atoi(kc);
  if (const char* ps = getenv("PQA_PROF_STRIDE")) h->prof_stride = (unsigned)std::max(1, atoi(ps));
  if (const char* lw = getenv("PQA_LW")) h->lw_mode = atoi(lw);
  if (const char* rs = getenv("PQA_RES")) h->res_mode = atoi(rs);
  if (const char* rs = getenv("PQA_R8")) h->r8_mode = atoi(rs);
  if (const char* rs = getenv("PQA_RES_PBC")) h->res_pbc = atoi(rs);
  if (const char* rs = getenv("PQA_RES_CX")) h->res_cx = atoi(rs);
  if (const char* rs = getenv("PQA_WW")) h->ww_mode = atoi(rs);
  if (const char* rs = getenv("PQA_ECP_DEFER")) h->ecp_defer = atoi(rs);
  if (const char* rs = getenv("PQA_EN_OVERLAP")) h->en_overlap = atoi(rs);
  if (const char* rs = getenv("PQA_DRAWS_AHEAD")) h->draws_ahead_on = atoi(rs) != 0;
  if (const char* rs = getenv("PQA_RES_MIN")) h->res_min = atol(rs);
  if (const char* rs = getenv("PQA_RES_MAX")) h->res_max = atol(rs);
  if (const char* ws = getenv("PQA_ORB_WS")) h->orb_ws = atoi(ws);
  if (const char* wd = getenv("PQA_ORB_WIDE")) h->orb_wide = atoi(wd);
  if (const char* wm = getenv("PQA_ORB_WIDE_MAX")) h->orb_wide_max = atol(wm);
  if (const char* kb = getenv("PQA_LW_KB")) h->lw_kb = atoi(kb);
  if (const char* nt = getenv("PQA_ORB_NOTAB")) h->orb_notab = atoi(nt);
  if (const char* gm = getenv("PQA_LW_GM")) h->lw_gm = atoi(gm);
  if (const char* nw = getenv("PQA_LW_NW")) { const int v = atoi(nw); h->lw_nw = (v == 16 || v == 32 || v == 64) ? v : 0; }
  if (const char* ew = getenv("PQA_ECP_WAVE")) h->ecp_wave = atoi(ew);
  if (const char* es = getenv("PQA_ECP_SOA_T")) h->ecp_soa_t = atoi(es);
  if (const char* ep = getenv("PQA_ECP_POINT_LW")) h->ecp_point_lw = atoi(ep);
  if (const char* el = getenv("PQA_ECP_LDS")) h->ecp_lds = atoi(el);
  if (const char* jf = getenv("PQA_JAS_FOLD")) h->jas_fold_allowed = atoi(jf);
  if (const char* am = getenv("PQA_ECP_ATOM_MAJOR")) h->ecp_atom_major = atoi(am);
  if (const char* ea = getenv("PQA_ECP_ACC_WAVES")) h->ecp_acc_waves = atoi(ea);
  if (const char* sp = getenv("PQA_STEP_PRE")) h->step_pre = atoi(sp);
  if (const char* sp = getenv("PQA_STEP_GW")) h->step_gw = atoi(sp);
  if (const char* sp = getenv("PQA_STEP_PRE_MAX")) h->step_pre_max = atol(sp);
  if (const char* dm = getenv("PQA_DRAWS_MAX")) h->draws_max = atol(dm);
  if (const char* fw = getenv("PQA_FLUSH_WB8_MAX")) h->flush_wb8_max = atol(fw);
  if (const char* sp = getenv("PQA_SPLIT")) h->split_mode = std::max(0, std::min(3, atoi(sp)));
  if (const char* sp = getenv("PQA_SPLIT_MIN")) h->split_min = std::max(512L, atol(sp));
  if (const char* sp = getenv("PQA_SPLIT_CUS")) h->split_cus = atoi(sp);
  if (const char* sp = getenv("PQA_JPRE")) h->jpre = atoi(sp);
  if (const char* sp = getenv("PQA_JAS_MERGE")) h->jas_merge = atoi(sp);
  if (const char* sp = getenv("PQA_JPRE_MIN")) h->jpre_min = std::max(64L, atol(sp));
  {
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, h->device) == hipSuccess && prop.multiProcessorCount > 0) h->cu_count = prop.multiProcessorCount;
  }
  h->natom = sys->natom; h->nup = sys->nelec_up; h->ndn = sys->nelec_dn; h->N = h->nup + h->ndn;
  h->nao = sys->nao; h->nshell = sys->nshell;
  h->has_slater = sys->has_slater != 0;
  h->cplx = h->has_slater && sys->complex_orbitals != 0;
  h->twist = sys->twisted != 0;
  // k_orb_wide: 1024 threads (64 lane groups) per 16-point tile; twisted cells need > 128 registers per thread: 512.  Untwisted
  // periodic cells fit 128 since the lattice sums accumulate in the tile: C5 +3 % at 1024-8192 walkers over 512 threads.
  h->wide_nth = (sys->pbc && h->twist) ? 512 : 1024;
  if (const char* e = getenv("PQA_TM_PRE")) h->tm_pre = atoi(e) != 0;
  if (const char* e = getenv("PQA_WIDE_NTH")) { if (sys->pbc && atoi(e) == 512) h->wide_nth = 512; }
  if (h->twist && !(h->cplx && sys->pbc && sys->nL > 0)) FAIL("twisted boundary conditions need pbc, complex_orbitals and the periodic orbital tables");
  if (h->cplx && ((sys->nmo_up | sys->nmo_dn) & 1)) FAIL("complex orbitals: nmo_up / nmo_dn count the real columns [Re C | Im C] and must be even");
  h->has_j2 = sys->na > 0 || sys->nb > 0;
  h->has_j3 = sys->na3 > 0 && sys->nb3 > 0;
  h->has_jastrow = h->has_j2 || h->has_j3;
  h->na = sys->na; h->nb = sys->nb; h->necp = sys->necp; h->na3 = h->has_j3 ? sys->na3 : 0; h->nb3 = h->has_j3 ? sys->nb3 : 0;
  if (h->na > PQA_MAXBAS || h->nb > PQA_MAXBAS) FAIL("more than 16 two-body Jastrow basis functions per kind");
  if (h->na3 > PQA_MAXBAS3 || h->nb3 > PQA_MAXBAS3) FAIL("more than 8 three-body Jastrow basis functions per kind");
  if (h->nup > PQA_MAXN || h->ndn > PQA_MAXN) FAIL("more than 128 electrons per spin channel are not supported");
  // More than 64 electrons or orbitals of a spin (slater.py:155-260 takes any number): the handle runs on the kernels that are
  // general in n — orbitals by the thread-per-point evaluator + k_mo_rows, determinants by the wave-per-walker kernels with two
  // columns per lane (k_build_invert, slater_ratios, sm_update_wave on the inverse in place), no lane-per-walker planes.
  h->big = h->nup > PQA_MAXN_FAST || h->ndn > PQA_MAXN_FAST || (sys->has_slater && (sys->nmo_up > PQA_MAXN_FAST || sys->nmo_dn > PQA_MAXN_FAST));
  if (h->big && h->twist) FAIL("twisted cells: at most 64 electrons and 64 orbitals per spin channel (the general orbital path evaluates real AOs)");
  if (h->big && h->cplx && !(sys->pbc && sys->nL > 0)) FAIL("complex orbitals beyond 64 per spin: periodic handles only");
  if (h->big) h->lw_mode = 0;
  if (const char* e = getenv("PQA_ORB_GENERAL")) h->orb_general = atoi(e) != 0;
  SysDev& S = h->S;
  S.natom = h->natom; S.nup = h->nup; S.ndn = h->ndn; S.nelec = h->N;
  S.pbc = sys->pbc;
  PbcDev P{};
  if (S.pbc < 0 || S.pbc > 2) FAIL("pbc must be 0 (open), 1 (orthogonal cell) or 2 (general cell)");
  if (S.pbc) {
    const double* a = sys->lattice;
    const double det = a[0] * (a[4] * a[8] - a[5] * a[7]) - a[1] * (a[3] * a[8] - a[5] * a[6]) + a[2] * (a[3] * a[7] - a[4] * a[6]);
    if (!(fabs(det) > 1e-12)) FAIL("singular lattice");
    for (int i = 0; i < 9; ++i) P.lat[i] = a[i];
    if (S.pbc == 2 && voronoi_vectors(a, P)) FAIL("could not determine the Voronoi-relevant vectors of the lattice");
    const double id = 1.0 / det;  // inverse by cofactors: linv[r][c] = cof(c,r) / det
    P.linv[0] = (a[4] * a[8] - a[5] * a[7]) * id; P.linv[1] = (a[2] * a[7] - a[1] * a[8]) * id; P.linv[2] = (a[1] * a[5] - a[2] * a[4]) * id;
    P.linv[3] = (a[5] * a[6] - a[3] * a[8]) * id; P.linv[4] = (a[0] * a[8] - a[2] * a[6]) * id; P.linv[5] = (a[2] * a[3] - a[0] * a[5]) * id;
    P.linv[6] = (a[3] * a[7] - a[4] * a[6]) * id; P.linv[7] = (a[1] * a[6] - a[0] * a[7]) * id; P.linv[8] = (a[0] * a[4] - a[1] * a[3]) * id;
    {  // inradius of {frac in [-1/2, 1/2)^3}: the face frac_c = 1/2 is 1 / (2 |column c of linv|) away from the origin
      double rho = 1e300;
      for (int c = 0; c < 3; ++c) rho = std::min(rho, 0.5 / sqrt(P.linv[c] * P.linv[c] + P.linv[3 + c] * P.linv[3 + c] + P.linv[6 + c] * P.linv[6 + c]));
      double rmax = 0.0;
      if (sys->na > 0) rmax = std::max(rmax, sys->rcut_a);
      if (sys->nb > 0) rmax = std::max(rmax, sys->rcut_b);
      if (sys->na3 > 0 && sys->nb3 > 0) rmax = std::max(rmax, std::max(sys->rcut_a3, sys->rcut_b3));
      P.jas_fold = (h->jas_fold_allowed && rmax <= rho * (1.0 + 1e-12)) ? 1 : 0;
    }
  }
  double* tmp_d; int* tmp_i;
  TRY(upload_table(h, sys->atom_xyz, (size_t)h->natom * 3, &tmp_d)); S.atom_xyz = tmp_d;
  TRY(upload_table(h, sys->atom_charge, (size_t)h->natom, &tmp_d)); S.atom_charge = tmp_d;
  for (int i = 0; i < h->natom; ++i)
    for (int j = i + 1; j < h->natom; ++j) {
      double d2 = 0;
      for (int k = 0; k < 3; ++k) { const double d = sys->atom_xyz[3 * i + k] - sys->atom_xyz[3 * j + k]; d2 += d * d; }
      h->ii_energy += sys->atom_charge[i] * sys->atom_charge[j] / std::sqrt(d2);
    }
  if (h->has_slater) {
    S.nshell = sys->nshell; S.nprim = sys->nprim; S.nao = sys->nao;
    for (int s = 0; s < sys->nshell; ++s) {
      if (sys->shell_l[s] < 0 || sys->shell_l[s] > 5) FAIL("shells up to h (l <= 5, as numba/gto.py:107-118) are implemented");
      if (sys->nL > 0 && sys->shell_l[s] > 3) {
        if (h->twist) FAIL("twisted cells: shells up to f (l <= 3); g and h shells are implemented for open systems and untwisted cells");
        h->pbc_high_l = true;  // the general (thread-per-point) orbital path, pqa_orb_pbc.hip
      }
      h->shell_l.push_back(sys->shell_l[s]);
      h->shell_np.push_back(sys->shell_prim_off[s + 1] - sys->shell_prim_off[s]);
      h->shell_ao.push_back(sys->shell_ao_off[s]);
    }
    TRY(upload_table(h, sys->shell_atom, (size_t)sys->nshell, &tmp_i)); S.shell_atom = tmp_i;
    TRY(upload_table(h, sys->shell_l, (size_t)sys->nshell, &tmp_i)); S.shell_l = tmp_i;
    TRY(upload_table(h, sys->shell_prim_off, (size_t)sys->nshell + 1, &tmp_i)); S.shell_prim_off = tmp_i;
    TRY(upload_table(h, sys->shell_ao_off, (size_t)sys->nshell, &tmp_i)); S.shell_ao_off = tmp_i;
    TRY(upload_table(h, sys->prim_exp, (size_t)sys->nprim, &tmp_d)); S.prim_exp = tmp_d;
    TRY(upload_table(h, sys->prim_coef, (size_t)sys->nprim, &tmp_d)); S.prim_coef = tmp_d;
    TRY(build_radial_tables(h, sys, S));
    S.nL = 0;
    if (S.pbc) {
      if (sys->nL <= 0 || !sys->Ls || !sys->num_Ls || !sys->atom_cut || !sys->shell_cut) FAIL("periodic orbitals need the lattice-sum tables (Ls, num_Ls, atom_cut, shell_cut)");
      std::vector<int> nl((size_t)h->natom);
      HIPCHK(hipMemcpy(nl.data(), sys->num_Ls, nl.size() * sizeof(int), hipMemcpyDefault));
      for (int v : nl)
        if (v < 1 || v > sys->nL) FAIL("num_Ls out of range");
      S.nL = sys->nL;
      TRY(upload_table(h, sys->Ls, (size_t)sys->nL * 3, &tmp_d)); P.Ls = tmp_d;
      TRY(upload_table(h, sys->num_Ls, (size_t)h->natom, &tmp_i)); P.num_Ls = tmp_i;
      TRY(upload_table(h, sys->atom_cut, (size_t)h->natom, &tmp_d)); P.atom_cut = tmp_d;
      TRY(upload_table(h, sys->shell_cut, (size_t)sys->nshell, &tmp_d)); P.shell_cut = tmp_d;
      {  // distinct shell cut-offs per atom, ascending: the classes k_pbc_prepass orders an atom's images by
        std::vector<double> sc((size_t)sys->nshell), cc((size_t)h->natom * PQA_MAXCLS, 0.0);
        std::vector<int> sa((size_t)sys->nshell), nc((size_t)h->natom, 0);
        HIPCHK(hipMemcpy(sc.data(), sys->shell_cut, sc.size() * sizeof(double), hipMemcpyDefault));
        HIPCHK(hipMemcpy(sa.data(), sys->shell_atom, sa.size() * sizeof(int), hipMemcpyDefault));
        for (int a = 0; a < h->natom; ++a) {
          std::vector<double> u;
          for (int q = 0; q < sys->nshell; ++q)
            if (sa[q] == a) u.push_back(sc[q]);
          std::sort(u.begin(), u.end());
          u.erase(std::unique(u.begin(), u.end()), u.end());
          if ((int)u.size() > PQA_MAXCLS) continue;  // nc = 0: this atom's images are tested directly
          nc[a] = (int)u.size();
          for (size_t q = 0; q < u.size(); ++q) cc[(size_t)a * PQA_MAXCLS + q] = u[q];
        }
        TRY(upload_table(h, cc.data(), cc.size(), &tmp_d)); P.cls_cut = tmp_d;
        TRY(upload_table(h, nc.data(), nc.size(), &tmp_i)); P.ncls = tmp_i;
        h->pbc_maxcls = *std::max_element(nc.begin(), nc.end());
        h->pbc_mincls = *std::min_element(nc.begin(), nc.end());
        if (const char* e = getenv("PQA_PRE_NCUT")) h->pbc_maxcls = std::max(h->pbc_maxcls, atoi(e));  // (tests: the ten-class instantiation)
      }
      {  // capacity of the per-(atom, point) image lists: the lattice points inside a sphere of the largest atom cut-off number
         // V_sphere / V_cell on average; 1.5 x that + 8 with room for a terminator (lanes beyond it test images directly)
        std::vector<double> ac((size_t)h->natom);
        HIPCHK(hipMemcpy(ac.data(), sys->atom_cut, ac.size() * sizeof(double), hipMemcpyDefault));
        const double* a = sys->lattice;
        const double vol = fabs(a[0] * (a[4] * a[8] - a[5] * a[7]) - a[1] * (a[3] * a[8] - a[5] * a[6]) + a[2] * (a[3] * a[7] - a[4] * a[6]));
        const double r2 = *std::max_element(ac.begin(), ac.end());
        const double mean = 4.18879020478639 * r2 * std::sqrt(r2) / std::max(vol, 1e-12);
        const int cap = (int)std::min(127.0, std::ceil(1.5 * mean + 8.0));
        h->pbc_nw = cap / 4 + 1;
        if (const char* e = getenv("PQA_PBC_NW")) h->pbc_nw = std::max(1, std::min(32, atoi(e)));
        TRY(near_masks(h, sys, nl, ac, P));
      }
      P.twist = h->twist ? 1 : 0;
      if (h->twist) {
        std::vector<double> ls((size_t)sys->nL * 3), ph((size_t)sys->nL * 2);
        HIPCHK(hipMemcpy(ls.data(), sys->Ls, ls.size() * sizeof(double), hipMemcpyDefault));
        for (int j = 0; j < sys->nL; ++j) {
          const double a = sys->twist_k[0] * ls[3 * j] + sys->twist_k[1] * ls[3 * j + 1] + sys->twist_k[2] * ls[3 * j + 2];
          ph[2 * j] = std::cos(a); ph[2 * j + 1] = std::sin(a);
        }
        TRY(upload_table(h, ph.data(), ph.size(), &tmp_d)); P.img_phase = tmp_d;
        for (int a = 0; a < 3; ++a)
          P.ktl[a] = sys->twist_k[0] * sys->lattice[3 * a] + sys->twist_k[1] * sys->lattice[3 * a + 1] + sys->twist_k[2] * sys->lattice[3 * a + 2];
      }
      P.member = nullptr;
      if (sys->member) {
        if (!sys->img_n || !sys->atom_n || !sys->member_class || sys->member_M < 0 || sys->n_member_class < 1) FAIL("incomplete image-membership tables");
        const double* a = sys->lattice_prim;
        const double det = a[0] * (a[4] * a[8] - a[5] * a[7]) - a[1] * (a[3] * a[8] - a[5] * a[6]) + a[2] * (a[3] * a[7] - a[4] * a[6]);
        if (!(fabs(det) > 1e-12)) FAIL("singular primitive lattice");
        const double id = 1.0 / det;
        double* v = P.lprim_inv;
        v[0] = (a[4] * a[8] - a[5] * a[7]) * id; v[1] = (a[2] * a[7] - a[1] * a[8]) * id; v[2] = (a[1] * a[5] - a[2] * a[4]) * id;
        v[3] = (a[5] * a[6] - a[3] * a[8]) * id; v[4] = (a[0] * a[8] - a[2] * a[6]) * id; v[5] = (a[2] * a[3] - a[0] * a[5]) * id;
        v[6] = (a[3] * a[7] - a[4] * a[6]) * id; v[7] = (a[1] * a[6] - a[0] * a[7]) * id; v[8] = (a[0] * a[4] - a[1] * a[3]) * id;
        const size_t side = 2 * (size_t)sys->member_M + 1;
        unsigned char* tmp_b;
        TRY(upload_table(h, sys->member, (size_t)sys->n_member_class * side * side * side, &tmp_b)); P.member = tmp_b;
        TRY(upload_table(h, sys->member_class, (size_t)h->natom, &tmp_i)); P.member_class = tmp_i;
        TRY(upload_table(h, sys->img_n, (size_t)sys->nL * 3, &tmp_i)); P.img_n = tmp_i;
        TRY(upload_table(h, sys->atom_n, (size_t)h->natom * 3, &tmp_i)); P.atom_n = tmp_i;
        P.member_M = sys->member_M;
        TRY(member_masks(h, sys, P));
        for (int i = 0; i < 9; ++i) {  // supercell matrix = lattice . inv(lattice_prim), must be integer
          double v_ = 0.0;
          for (int k = 0; k < 3; ++k) v_ += sys->lattice[3 * (i / 3) + k] * P.lprim_inv[3 * k + (i % 3)];
          P.supercell[i] = (int)lround(v_);
          if (fabs(v_ - P.supercell[i]) > 1e-6) FAIL("lattice is not an integer multiple of lattice_prim");
        }
      }
    }
    h->nmo[0] = sys->nmo_up; h->nmo[1] = sys->nmo_dn;
    h->ndet = sys->ndet; h->ndet_s[0] = sys->ndet_up; h->ndet_s[1] = sys->ndet_dn;
    S.ndet = h->ndet;
    const int* occ_src[2] = {sys->det_occ_up, sys->det_occ_dn};
    const double* mo_src[2] = {sys->mo_up, sys->mo_dn};
    const int nel[2] = {h->nup, h->ndn};
    shell_costs(h, sys);
    build_chunks(h, 16, h->chunks[0]);
    build_chunks(h, 32, h->chunks[1]);
    for (int s = 0; s < 2; ++s) {
      if (h->nmo[s] > (h->cplx ? 2 : 1) * PQA_MAXN) FAIL("more than 128 orbitals per spin are not supported");
      const int nt = (h->nmo[s] + 15) / 16;
      h->nt[s] = nt <= 1 ? 1 : (nt == 2 ? 2 : (nt <= 4 ? 4 : (nt <= 8 ? 8 : 16)));  // (8: padded coefficient rows of 128 columns, contracted in two windows of four tiles; periodic big handles through k_mo_rows)
      S.nmo[s] = h->nmo[s]; S.ndet_s[s] = h->ndet_s[s];
      TRY(upload_table(h, occ_src[s], (size_t)h->ndet_s[s] * nel[s], &tmp_i)); S.det_occ[s] = tmp_i;
      {
        std::vector<int> oc((size_t)std::max(nel[s], 1));
        if (nel[s] > 0) HIPCHK(hipMemcpy(oc.data(), occ_src[s], (size_t)nel[s] * sizeof(int), hipMemcpyDefault));
        S.occ_ident[s] = 1;
        for (int k = 0; k < nel[s]; ++k) S.occ_ident[s] &= (oc[k] == k) ? 1 : 0;
      }
      {
        std::vector<int> occ_h((size_t)h->ndet_s[s] * nel[s]), cm((size_t)h->ndet_s[s] * std::max(h->nmo[s], 1), -1);
        if (!occ_h.empty()) HIPCHK(hipMemcpy(occ_h.data(), occ_src[s], occ_h.size() * sizeof(int), hipMemcpyDefault));
        for (int u = 0; u < h->ndet_s[s]; ++u)
          for (int k = 0; k < nel[s]; ++k) {
            const int m = occ_h[(size_t)u * nel[s] + k];
            if (m < 0 || m >= h->nmo[s]) FAIL("determinant occupation outside the orbital range");
            cm[(size_t)u * h->nmo[s] + m] = k;
          }
        TRY(upload_table(h, cm.data(), cm.size(), &h->d_colmap[s]));
      }
      TRY(upload_table<double>(h, nullptr, (size_t)h->nao * std::max(h->nmo[s], 1), &h->d_mo[s])); S.mo[s] = h->d_mo[s];
      for (int t = 0; t < 2; ++t)
        TRY(upload_table<double>(h, nullptr, (size_t)(std::max(h->chunks[t].rows_pad, 1) + 32) * 16 * h->nt[s], &h->d_cpad[t][s]));
      if (h->nmo[s] > 0) TRY(set_mo(h, s, mo_src[s]));
    }
    TRY(upload_table(h, sys->det_coeff, (size_t)h->ndet, &h->d_detcoeff)); S.det_coeff = h->d_detcoeff;
    TRY(upload_table(h, sys->det_map, (size_t)2 * h->ndet, &tmp_i)); S.det_map = tmp_i;
    for (int t = 0; t < 2; ++t) {
      const ChunkHost& c = h->chunks[t];
      ChunkTab& T = h->tab[t];
      T.nchunk = (int)c.nk.size();
      TRY(upload_table(h, c.nk.data(), c.nk.size(), &tmp_i)); T.chunk_nk = tmp_i;
      TRY(upload_table(h, c.shell_kb.data(), c.shell_kb.size(), &tmp_i)); T.shell_kb = tmp_i;
      TRY(upload_table(h, c.row0.data(), c.row0.size(), &tmp_i)); T.chunk_row0 = tmp_i;
      for (int g = 0; g < 3; ++g) {
        TRY(upload_table(h, c.cw_off[g].data(), c.cw_off[g].size(), &tmp_i)); T.cw_off[g] = tmp_i;
        TRY(upload_table(h, c.cw_shell[g].data(), c.cw_shell[g].size(), &tmp_i)); T.cw_shell[g] = tmp_i;
      }
      for (int s = 0; s < 2; ++s) { T.cpad[s] = h->d_cpad[t][s]; T.ldc[s] = 16 * h->nt[s]; }
      // k_orb_wide: all shells dealt to 64 lane groups (longest processing time first), tile row of a shell = its padded row
      const int tw = h->twist ? 2 : 1;
      const int ngrp = h->wide_nth / 16;  // lane groups of k_orb_wide (16 points per block)
      std::vector<int> order((size_t)h->nshell), wrow((size_t)tw * h->nshell), woff(65, 0), wsh;
      auto cost = [&](int s) { return h->shell_cost[s]; };
      for (int s = 0; s < h->nshell; ++s) order[s] = s;
      std::stable_sort(order.begin(), order.end(), [&](int a, int b) { return cost(a) > cost(b); });
      std::vector<std::vector<int>> grp(64);
      std::vector<int> load(64, 0);
      for (int s : order) {
        int best = 0;
        for (int g = 1; g < ngrp; ++g)
          if (load[g] < load[best]) best = g;
        grp[best].push_back(s);
        load[best] += cost(s);
      }
      for (int g = 0; g < 64; ++g) {
        // shells of one atom next to each other: a thread re-reads the per-(point, atom) fold / mask only when the atom changes
        std::stable_sort(grp[g].begin(), grp[g].end(), [&](int a, int b) { return sys->shell_atom[a] < sys->shell_atom[b]; });
        for (int s : grp[g]) wsh.push_back(s);
        woff[g + 1] = (int)wsh.size();
      }
      for (int s = 0; s < tw * h->nshell; ++s) wrow[s] = c.row0[c.shell_chunk[s]] + c.shell_kb[s];
      TRY(upload_table(h, woff.data(), woff.size(), &tmp_i)); h->wide[t].off = tmp_i;
      TRY(upload_table(h, wsh.data(), wsh.size(), &tmp_i)); h->wide[t].shell = tmp_i;
      TRY(upload_table(h, wrow.data(), wrow.size(), &tmp_i)); h->wide[t].row = tmp_i;
      h->wide[t].rows_pad = c.rows_pad;
    }
  }
  S.na = h->na; S.nb = h->nb; S.rcut_a = sys->rcut_a; S.rcut_b = sys->rcut_b;
  for (int k = 0; k < h->na; ++k) { S.a_kind[k] = sys->a_kind[k]; S.a_param[k] = sys->a_param[k]; S.a_aux[k] = 1.0 / (3.0 + sys->a_param[k]); }
  for (int k = 0; k < h->nb; ++k) { S.b_kind[k] = sys->b_kind[k]; S.b_param[k] = sys->b_param[k]; S.b_aux[k] = 1.0 / (3.0 + sys->b_param[k]); }
  if (S.pbc) {  // all periodic tables sit behind one pointer (see SysDev)
    // (the resident sweep's in-block image lists need the mask tables, at most 128 candidates and a handful of shell cut-offs per atom)
    h->pbc_lists_ok = sys->nL > 0 && sys->nL <= 128 && (P.member == nullptr || P.memb_mask != nullptr) && h->pbc_mincls >= 1 && h->pbc_maxcls <= PQA_RES_NCUT;
    PbcDev* dp;
    TRY(upload_table(h, &P, (size_t)1, &dp));
    S.pb = dp;
  }
  TRY(upload_table(h, sys->acoeff, (size_t)h->natom * h->na * 2, &h->d_acoeff)); S.acoeff = h->d_acoeff;
  TRY(upload_table(h, sys->bcoeff, (size_t)h->nb * 3, &h->d_bcoeff)); S.bcoeff = h->d_bcoeff;
  TRY(jas_merge_tables(h));
  S.na3 = h->na3; S.nb3 = h->nb3; S.rcut_a3 = sys->rcut_a3; S.rcut_b3 = sys->rcut_b3;
  for (int k = 0; k < h->na3; ++k) { S.a3_kind[k] = sys->a3_kind[k]; S.a3_param[k] = sys->a3_param[k]; S.a3_aux[k] = 1.0 / (3.0 + sys->a3_param[k]); }
  for (int k = 0; k < h->nb3; ++k) { S.b3_kind[k] = sys->b3_kind[k]; S.b3_param[k] = sys->b3_param[k]; S.b3_aux[k] = 1.0 / (3.0 + sys->b3_param[k]); }
  TRY(upload_table<double>(h, nullptr, (size_t)h->natom * h->na3 * h->na3 * h->nb3 * 3, &h->d_c3)); S.c3 = h->d_c3;
  if (h->has_j3 && sys->ccoeff) TRY(set_c3(h, sys->ccoeff));
  {  // the three-body scratch sits behind whatever else a kernel keeps in dynamic LDS
    const size_t n = std::max(sys->nelec_up, sys->nelec_dn);
    const size_t other = std::max((n > PQA_MAXN_FAST ? 3 * n + 64 : n * (n + 1) + 3 * n + 64) * sizeof(double),
                                  (size_t)std::max(sys->has_slater ? std::max(sys->ndet_up, sys->ndet_dn) : 1, 1) * 5 * sizeof(double));
    S.j3_off = (int)((other + 7) / 8);
  }
  S.necp = h->necp;
  if (h->necp > 0) {
    const int nchan = sys->ecp_chan_off[h->necp];
    const int nterm = sys->ecp_term_off[nchan];
    h->ecp_nchan = nchan; h->ecp_nterm = nterm;
    for (int k = 0; k < h->necp; ++k)
      if (sys->ecp_chan_off[k + 1] - sys->ecp_chan_off[k] > PQA_MAXCHAN) FAIL("ECP with more than 5 non-local channels (the reference's Legendre functions end at l = 4, eval_ecp.py:203-225)");
    TRY(upload_table(h, sys->ecp_atom, (size_t)h->necp, &tmp_i)); S.ecp_atom = tmp_i;
    TRY(upload_table(h, sys->ecp_chan_off, (size_t)h->necp + 1, &tmp_i)); S.ecp_chan_off = tmp_i;
    TRY(upload_table(h, sys->ecp_term_off, (size_t)nchan + 1, &tmp_i)); S.ecp_term_off = tmp_i;
    TRY(upload_table(h, sys->ecp_term_n, (size_t)nterm, &tmp_i)); S.ecp_term_n = tmp_i;
    TRY(upload_table(h, sys->ecp_term_exp, (size_t)nterm, &tmp_d)); S.ecp_term_exp = tmp_d;
    TRY(upload_table(h, sys->ecp_term_coef, (size_t)nterm, &tmp_d)); S.ecp_term_coef = tmp_d;
    {  // range of every ECP atom: r^2 beyond which all of its terms |c| r^n exp(-a r^2) stay below 1e-22 (k_ecp_count visits an
       // electron's near atoms only; what it leaves out is below the last bit of the local energy and can never pass the mask)
      std::vector<int> co((size_t)h->necp + 1), to((size_t)nchan + 1), tn((size_t)std::max(nterm, 1));
      std::vector<double> te(tn.size()), tc(tn.size()), rc2((size_t)std::max(h->necp, 1), 0.0);
      HIPCHK(hipMemcpy(co.data(), sys->ecp_chan_off, co.size() * sizeof(int), hipMemcpyDefault));
      HIPCHK(hipMemcpy(to.data(), sys->ecp_term_off, to.size() * sizeof(int), hipMemcpyDefault));
      if (nterm > 0) {
        HIPCHK(hipMemcpy(tn.data(), sys->ecp_term_n, (size_t)nterm * sizeof(int), hipMemcpyDefault));
        HIPCHK(hipMemcpy(te.data(), sys->ecp_term_exp, (size_t)nterm * sizeof(double), hipMemcpyDefault));
        HIPCHK(hipMemcpy(tc.data(), sys->ecp_term_coef, (size_t)nterm * sizeof(double), hipMemcpyDefault));
      }
      for (int k = 0; k < h->necp; ++k) {
        double rc = 0.0;
        for (int t = to[co[k]]; t < to[co[k + 1]]; ++t) {
          if (tc[t] == 0.0) continue;
          if (!(te[t] > 0.0)) { rc = 1e150; break; }  // no decay: never out of range
          double r = 60.0;  // walk inwards until the term is visible
          while (r > 0.02 && fabs(tc[t]) * std::pow(r, (double)tn[t]) * std::exp(-te[t] * r * r) < 1e-22) r -= 0.01;
          rc = std::max(rc, r + 0.02);
        }
        rc2[k] = rc * rc;
      }
      TRY(upload_table(h, rc2.data(), rc2.size(), &tmp_d)); S.ecp_rc2 = tmp_d;
    }
  }
  // quadrature grids (eval_ecp.py:278-336): all six rules of Mitas, Shirley & Ceperley in one table — rows 0-5 OA (6), 6-17 IAB (12),
  // 18-35 OAB (18), 36-61 OABC (26), 62-93 IABC (32), 94-143 OABCD (50) — in the reference's point order, with their weights
  {
    std::vector<double> quad, quadw;
    ecp_quadrature_tables(quad, quadw);
    TRY(upload_table(h, quad.data(), quad.size(), &h->d_quad));
    TRY(upload_table(h, quadw.data(), quadw.size(), &h->d_quadw));
    std::vector<int> na((size_t)std::max(h->necp, 1), 0), qo((size_t)std::max(h->necp, 1), 0);
    h->ecp_nch.assign((size_t)h->necp, 0);
    for (int k = 0; k < h->necp; ++k) {
      h->ecp_nch[k] = sys->ecp_chan_off[k + 1] - sys->ecp_chan_off[k];
      na[k] = h->ecp_nch[k] <= 2 ? 6 : 12;  // eval_ecp.py:239-240
      qo[k] = ecp_quadrature_offset(na[k]);
    }
    TRY(upload_table(h, na.data(), na.size(), &h->d_ecp_naip));
    TRY(upload_table(h, qo.data(), qo.size(), &h->d_ecp_qoff));
    S.ecp_naip = h->d_ecp_naip; S.ecp_qoff = h->d_ecp_qoff;
    S.ecp_naip_max = 0;
    for (int k = 0; k < h->necp; ++k) S.ecp_naip_max = std::max(S.ecp_naip_max, na[k]);
  }
  {  // flat (atom, quadrature index) list of the T-move candidates of one electron
    std::vector<int> ptk, pti;
    for (int k = 0; k < h->necp; ++k) {
      const int nch = sys->ecp_chan_off[k + 1] - sys->ecp_chan_off[k];
      const int naip = nch <= 2 ? 6 : 12;
      for (int i = 0; i < naip; ++i) { ptk.push_back(k); pti.push_back(i); }
    }
    h->tm_P = (int)ptk.size();
    TRY(upload_table(h, ptk.data(), ptk.size(), &h->d_ptk));
    TRY(upload_table(h, pti.data(), pti.size(), &h->d_pti));
  }
  return 0;
}

extern "C" int pqa_create(const pqa_system_t* sys, int device, pqa_handle_t** out) {
  *out = nullptr;
  pqa_handle* h = new pqa_handle();
  h->device = device;
  int rc = create_impl(h, sys);
  if (rc) {
    g_create_error = h->err;
    pqa_destroy(h);
    return rc;
  }
  *out = h;
  return 0;
}

extern "C" void pqa_destroy(pqa_handle_t* h) {
  if (!h) return;
  (void)hipSetDevice(h->device);
  if (h->stream) (void)hipStreamSynchronize(h->stream);
  for (void* p : h->owned) (void)hipFree(p);
  DevBuf* bufs[] = {&h->b_x, &h->b_T[0], &h->b_T[1], &h->b_dsign[0], &h->b_dsign[1], &h->b_dlog[0], &h->b_dlog[1],
                    &h->b_cache[0], &h->b_cache[1], &h->b_aval, &h->b_bval, &h->b_pts, &h->b_motmp, &h->b_out, &h->b_widx,
                    &h->b_mask, &h->b_ao, &h->b_flag, &h->b_newpos, &h->b_aux, &h->b_accept, &h->b_accrec, &h->b_acccnt, &h->b_accw,
                    &h->b_gauss, &h->b_unif, &h->b_kc, &h->b_en, &h->b_means, &h->b_sign, &h->b_log, &h->b_ju, &h->b_rot,
                    &h->b_eunif, &h->b_elocal, &h->b_ecnt, &h->b_eoff, &h->b_epts[0], &h->b_epts[1], &h->b_ewgt[0],
                    &h->b_ewgt[1], &h->b_epte[0], &h->b_epte[1], &h->b_emo[0], &h->b_emo[1], &h->b_ecp, &h->b_xt, &h->b_Tt[0], &h->b_Tt[1], &h->b_rc[0], &h->b_rc[1], &h->b_sel[0], &h->b_sel[1], &h->b_auxt, &h->b_kpart, &h->b_rbuf, &h->b_vbuf, &h->b_act, &h->b_alt_x, &h->b_alt_T[0], &h->b_alt_T[1], &h->b_alt_dsign[0], &h->b_alt_dsign[1], &h->b_alt_dlog[0], &h->b_alt_dlog[1], &h->b_alt_cache[0], &h->b_alt_cache[1], &h->b_alt_aval, &h->b_alt_bval, &h->b_alt_j3u, &h->b_rsidx, &h->b_tpos, &h->b_twgt, &h->b_tlive, &h->b_trat, &h->b_tmcnt, &h->b_tmoff, &h->b_tmpass, &h->b_tmamp, &h->b_tmacc, &h->b_tmidx, &h->b_tmapos, &h->b_tmu, &h->b_tmtile, &h->b_tmaoff, &h->b_tmptw, &h->b_tmmarks, &h->b_dmcw, &h->b_dmcold, &h->b_dmcr2, &h->b_dmcout, &h->b_j3u, &h->b_dwrap, &h->b_wrap, &h->b_epass, &h->b_eptw[0], &h->b_eptw[1], &h->b_econ[0], &h->b_econ[1], &h->b_eu0[0], &h->b_eu0[1], &h->b_tves, &h->b_pgdet, &h->b_pbcd0, &h->b_pbcmask, &h->b_pbcth, &h->b_tmuold, &h->b_gauss_b, &h->b_unif_b};
  for (DevBuf* b : bufs)
    if (b->p) (void)hipFree(b->p);
  for (auto& d : h->dm)
    for (DevBuf* b : {&d.pos, &d.row, &d.f, &d.newpos, &d.keep_pos, &d.keep_row, &d.keep_f, &d.cfg})
      if (b->p) (void)hipFree(b->p);
  for (DevBuf* b : {&h->dm_val, &h->dm_norm[0], &h->dm_norm[1], &h->dm_tmp, &h->dm_ijkl, &h->dm_assign[0], &h->dm_assign[1], &h->dm_ratio, &h->dm_acc})
    if (b->p) (void)hipFree(b->p);
  for (auto& pr : h->prof_events) { (void)hipEventDestroy(pr.first); (void)hipEventDestroy(pr.second); }
  for (auto& pr : h->prof2_events) { (void)hipEventDestroy(pr.first); (void)hipEventDestroy(pr.second); }
  for (auto& pr : h->prof3_events) { (void)hipEventDestroy(pr.first); (void)hipEventDestroy(pr.second); }
  if (h->ev0) (void)hipEventDestroy(h->ev0);
  if (h->ev1) (void)hipEventDestroy(h->ev1);
  for (hipEvent_t e : h->pipe_events) (void)hipEventDestroy(e);
  for (hipStream_t s : h->pipe_stream)
    if (s) { (void)hipStreamSynchronize(s); (void)hipStreamDestroy(s); }
  for (hipStream_t s : h->jas_stream)
    if (s) { (void)hipStreamSynchronize(s); (void)hipStreamDestroy(s); }
  if (h->b_jpre.p) (void)hipFree(h->b_jpre.p);
  if (h->pin_tot) (void)hipHostFree(h->pin_tot);
  if (h->en_stream) { (void)hipStreamSynchronize(h->en_stream); (void)hipStreamDestroy(h->en_stream); }
  if (h->draw_stream) { (void)hipStreamSynchronize(h->draw_stream); (void)hipStreamDestroy(h->draw_stream); }
  for (hipEvent_t e : h->draw_ev) if (e) (void)hipEventDestroy(e);
  for (hipEvent_t e : h->en_ev)
    if (e) (void)hipEventDestroy(e);
  if (h->stream) (void)hipStreamDestroy(h->stream);
  delete h;
}

// ---------------------------------------------------------------- merged Pade numerators
// Tables for pade_merged (pqa_jastrow.hpp): with D_k = 1 + beta_k p over the PolyPade functions k of a basis and a coefficient set c,
//   N1 = sum_k c_k prod_{j != k} D_j,  N2 = sum_k c_k (1 + beta_k) prod_{j != k} D_j^2,  N3 = sum_k c_k beta_k (1 + beta_k) prod_{j != k} D_j^3
// as ascending coefficients at offsets 0 / 4 / 11 of a PQA_JQ-double record, one record per (atom, spin of the electron) and per
// electron-electron spin channel; D = prod_k D_k in S.a_D / S.b_D.  Products in long double, rounded once.  Called at create and
// after every change of acoeff / bcoeff.  Route available (S.jq_on) when every non-empty basis is [cusp]? + 1..4 Pade functions.
typedef std::vector<long double> Poly;
static Poly poly_mul(const Poly& a, const Poly& b) {
  Poly c(a.size() + b.size() - 1, 0.0L);
  for (size_t i = 0; i < a.size(); ++i)
    for (size_t j = 0; j < b.size(); ++j) c[i + j] += a[i] * b[j];
  return c;
}
static void jas_merge_record(const std::vector<double>& beta, const double* c, size_t cstride, double* rec) {
  const int K = (int)beta.size();
  Poly n1(1, 0.0L), n2(1, 0.0L), n3(1, 0.0L);
  auto add = [](Poly& acc, const Poly& t, long double f) {
    if (acc.size() < t.size()) acc.resize(t.size(), 0.0L);
    for (size_t i = 0; i < t.size(); ++i) acc[i] += f * t[i];
  };
  for (int k = 0; k < K; ++k) {
    Poly o(1, 1.0L);
    for (int j = 0; j < K; ++j)
      if (j != k) o = poly_mul(o, Poly{1.0L, (long double)beta[j]});
    const Poly o2 = poly_mul(o, o), o3 = poly_mul(o2, o);
    const long double ck = c[(size_t)k * cstride], bk = beta[k];
    add(n1, o, ck); add(n2, o2, ck * (1.0L + bk)); add(n3, o3, ck * bk * (1.0L + bk));
  }
  for (int i = 0; i < PQA_JQ; ++i) rec[i] = 0.0;
  for (size_t i = 0; i < n1.size() && i < 4; ++i) rec[i] = (double)n1[i];
  for (size_t i = 0; i < n2.size() && i < 7; ++i) rec[4 + i] = (double)n2[i];
  for (size_t i = 0; i < n3.size() && i < 10; ++i) rec[11 + i] = (double)n3[i];
}
static int jas_merge_tables(pqa_handle* h) {
  SysDev& S = h->S;
  S.jq_on = S.jq_a = S.jq_b = 0;
  if (!h->jas_merge || !h->has_j2 || (h->na == 0 && h->nb == 0)) return 0;
  auto pades = [](int n, const int* kind, const double* par, std::vector<double>& beta, int& first) {
    first = (n > 0 && kind[0] == 1) ? 1 : 0;
    beta.clear();
    for (int k = first; k < n; ++k) {
      if (kind[k] != 0 || !(par[k] > -1.0)) return false;
      beta.push_back(par[k]);
    }
    return n == 0 || (beta.size() >= 1 && beta.size() <= 4);
  };
  std::vector<double> ba, bb;
  int fa = 0, fb = 0;
  if (!pades(h->na, S.a_kind, S.a_param, ba, fa) || !pades(h->nb, S.b_kind, S.b_param, bb, fb)) return 0;
  auto denom = [](const std::vector<double>& beta, double* D) {
    Poly d(1, 1.0L);
    for (double b : beta) d = poly_mul(d, Poly{1.0L, (long double)b});
    for (int i = 0; i < 5; ++i) D[i] = i < (int)d.size() ? (double)d[i] : 0.0;
  };
  denom(ba, S.a_D); denom(bb, S.b_D);
  std::vector<double> ac((size_t)h->natom * h->na * 2 + 1), bc((size_t)h->nb * 3 + 1);
  HIPCHK(hipMemcpy(ac.data(), h->d_acoeff, (size_t)h->natom * h->na * 2 * sizeof(double), hipMemcpyDeviceToHost));
  HIPCHK(hipMemcpy(bc.data(), h->d_bcoeff, (size_t)h->nb * 3 * sizeof(double), hipMemcpyDeviceToHost));
  std::vector<double> aq((size_t)h->natom * 2 * PQA_JQ + PQA_JQ, 0.0), bq((size_t)3 * PQA_JQ, 0.0);
  if (h->na > 0)
    for (int I = 0; I < h->natom; ++I)
      for (int sp = 0; sp < 2; ++sp) jas_merge_record(ba, ac.data() + ((size_t)I * h->na + fa) * 2 + sp, 2, aq.data() + ((size_t)I * 2 + sp) * PQA_JQ);
  if (h->nb > 0)
    for (int ch = 0; ch < 3; ++ch) jas_merge_record(bb, bc.data() + (size_t)fb * 3 + ch, 3, bq.data() + (size_t)ch * PQA_JQ);
  if (!h->d_aq) {
    TRY(upload_table<double>(h, nullptr, aq.size(), &h->d_aq));
    TRY(upload_table<double>(h, nullptr, bq.size(), &h->d_bq));
  }
  HIPCHK(hipMemcpy(h->d_aq, aq.data(), aq.size() * sizeof(double), hipMemcpyHostToDevice));
  HIPCHK(hipMemcpy(h->d_bq, bq.data(), bq.size() * sizeof(double), hipMemcpyHostToDevice));
  S.aq = h->d_aq; S.bq = h->d_bq;
  S.jq_a = (int)ba.size(); S.jq_b = (int)bb.size(); S.jq_on = 1;
  return 0;
}

// ---------------------------------------------------------------- parameters
extern "C" int pqa_set_param(pqa_handle_t* h, const char* name, const double* data, int64_t n) {
  h->dmc_old_valid = false;  // (the state pqa_dmc_continue refers to is gone)
  HIPCHK(hipSetDevice(h->device));
  HIPCHK(hipStreamSynchronize(h->stream));
  const std::string k(name);
  auto expect = [&](int64_t want) { return n == want; };
  if (k == "acoeff") {
    if (!expect((int64_t)h->natom * h->na * 2)) FAIL("acoeff size mismatch");
    HIPCHK(hipMemcpy(h->d_acoeff, data, n * sizeof(double), hipMemcpyDefault));
    TRY(jas_merge_tables(h));
  } else if (k == "bcoeff") {
    if (!expect((int64_t)h->nb * 3)) FAIL("bcoeff size mismatch");
    HIPCHK(hipMemcpy(h->d_bcoeff, data, n * sizeof(double), hipMemcpyDefault));
    TRY(jas_merge_tables(h));
  } else if (k == "ccoeff") {
    if (!h->has_j3 || !expect((int64_t)h->natom * h->na3 * h->na3 * h->nb3 * 3)) FAIL("ccoeff size mismatch");
    std::vector<double> host((size_t)n);
    HIPCHK(hipMemcpy(host.data(), data, n * sizeof(double), hipMemcpyDefault));
    TRY(set_c3(h, host.data()));
  } else if (k == "det_coeff") {
    if (!h->has_slater || !expect(h->ndet)) FAIL("det_coeff size mismatch");
    HIPCHK(hipMemcpy(h->d_detcoeff, data, n * sizeof(double), hipMemcpyDefault));
  } else if (k == "mo_coeff_alpha" || k == "mo_coeff_beta") {
    const int s = k == "mo_coeff_beta";
    if (!h->has_slater || !expect((int64_t)h->nao * h->nmo[s])) FAIL("mo_coeff size mismatch");
    std::vector<double> host((size_t)n);
    HIPCHK(hipMemcpy(host.data(), data, n * sizeof(double), hipMemcpyDefault));
    TRY(set_mo(h, s, host.data()));
  } else
    FAIL("unknown parameter name");
  h->saved_valid = false;
  return 0;
}

extern "C" int pqa_get_param(pqa_handle_t* h, const char* name, double* out, int64_t n) {
  HIPCHK(hipSetDevice(h->device));
  HIPCHK(hipStreamSynchronize(h->stream));
  const std::string k(name);
  const double* src = nullptr;
  int64_t want = 0;
  if (k == "acoeff") { src = h->d_acoeff; want = (int64_t)h->natom * h->na * 2; }
  else if (k == "bcoeff") { src = h->d_bcoeff; want = (int64_t)h->nb * 3; }
  else if (k == "det_coeff") { src = h->d_detcoeff; want = h->ndet; }
  else if (k == "mo_coeff_alpha") { src = h->d_mo[0]; want = (int64_t)h->nao * h->nmo[0]; }
  else if (k == "mo_coeff_beta") { src = h->d_mo[1]; want = (int64_t)h->nao * h->nmo[1]; }
  else if (k == "radial_table_info") {  // [number of table doubles, largest fit error relative to sum |c|] (build_radial_tables)
    if (n != 2) FAIL("parameter size mismatch");
    long used = 0;
    for (size_t q = 0; q + 1 < h->rt_shells.size(); q += 2)
      if (h->rt_shells[q] >= 0) used = std::max(used, (long)h->rt_shells[q] + (long)h->rt_shells[q + 1] * PQA_RT_REC);
    out[0] = (double)used; out[1] = h->rt_err;
    return 0;
  }
  else FAIL("unknown parameter name");
  if (n != want) FAIL("parameter size mismatch");
  HIPCHK(hipMemcpy(out, src, n * sizeof(double), hipMemcpyDefault));
  return 0;
}


extern "C" int pqa_eval_ao(pqa_handle_t* h, const double* pts, int64_t npts, int ncomp, double* out) {
  HIPCHK(hipSetDevice(h->device));
  if (!h->has_slater) FAIL("handle has no orbital tables");
  if (ncomp != 1 && ncomp != 4 && ncomp != 5) FAIL("ncomp must be 1, 4 or 5");
  if (h->twist) FAIL("AO-only evaluation is not available for twisted cells (complex AOs); use pqa_eval_mo");
  if (npts <= 0) return 0;
  const size_t nout = (size_t)ncomp * npts * h->nao;
  TRY(ensure(h, h->b_pts, (size_t)npts * 3 * sizeof(double)));
  TRY(ensure(h, h->b_ao, nout * sizeof(double)));
  TRY(copy_in(h, h->b_pts.p, pts, (size_t)npts * 3 * sizeof(double)));
  const dim3 grid((unsigned)((npts + 63) / 64)), block(64);
  TRY(launch_ao(h, plain_points((const double*)h->b_pts.p, npts), npts, ncomp, (double*)h->b_ao.p));
  TRY(check_launch(h, "k_ao"));
  return copy_out(h, out, h->b_ao.p, nout * sizeof(double));
}

extern "C" int pqa_eval_mo(pqa_handle_t* h, int spin, const double* pts, int64_t npts, int ncomp, int use_mfma, double* out) {
  HIPCHK(hipSetDevice(h->device));
  if (!h->has_slater) FAIL("handle has no orbital tables");
  if (ncomp != 1 && ncomp != 5) FAIL("ncomp must be 1 or 5");
  if (spin < 0 || spin > 1) FAIL("spin must be 0 or 1");
  if (npts <= 0 || h->nmo[spin] == 0) return 0;
  const int nmo = h->nmo[spin];
  const size_t nout = (size_t)ncomp * npts * nmo;
  h->saved_valid = false;
  TRY(ensure(h, h->b_pts, (size_t)npts * 3 * sizeof(double)));
  TRY(ensure(h, h->b_out, nout * sizeof(double)));
  TRY(copy_in(h, h->b_pts.p, pts, (size_t)npts * 3 * sizeof(double)));
  std::vector<double> host(nout);
  if (use_mfma) {
    TRY(ensure(h, h->b_motmp, nout * sizeof(double)));
    TRY(launch_orb(h, spin, plain_points((const double*)h->b_pts.p, npts), npts, ncomp, (double*)h->b_motmp.p));
    TRY(copy_out(h, host.data(), h->b_motmp.p, nout * sizeof(double)));
    std::vector<double> tr(nout);  // [p][c][j] -> [c][p][j]
    for (int64_t p = 0; p < npts; ++p)
      for (int c = 0; c < ncomp; ++c)
        memcpy(&tr[((size_t)c * npts + p) * nmo], &host[((size_t)p * ncomp + c) * nmo], nmo * sizeof(double));
    HIPCHK(hipMemcpy(out, tr.data(), nout * sizeof(double), hipMemcpyDefault));
    return 0;
  }
  const size_t nao_out = (size_t)ncomp * npts * h->nao;
  TRY(ensure(h, h->b_ao, nao_out * sizeof(double)));
  const dim3 grid((unsigned)((npts + 63) / 64)), block(64);
  TRY(launch_ao(h, plain_points((const double*)h->b_pts.p, npts), npts, ncomp, (double*)h->b_ao.p));
  const long rows = (long)ncomp * npts;
  hipLaunchKernelGGL((k_mo_valu<>), dim3((unsigned)((rows * nmo + 255) / 256)), dim3(256), 0, h->stream, (const double*)h->b_ao.p,
                     (const double*)h->d_mo[spin], rows, h->nao, nmo, (double*)h->b_out.p);
  TRY(check_launch(h, "k_mo_valu"));
  return copy_out(h, out, h->b_out.p, nout * sizeof(double));
}

// ---------------------------------------------------------------- walker state allocation
static int ensure_walkers(pqa_handle* h, long W) {
  if (W <= 0) FAIL("number of walkers must be positive");
  if (W != h->W) {  // (the deferred ECP point totals and the kernel-choice hints describe the shard they were measured on: ADVICE r5)
    h->ecp_hint_valid = false; h->last_ecp_dev[0] = h->last_ecp_dev[1] = nullptr;
    if (h->last_ecp_points < 0) h->last_ecp_points = 0;
  }
  h->W = W;
  h->saved_valid = false;
  TRY(ensure(h, h->b_x, (size_t)W * h->N * 3 * sizeof(double)));
  h->js.x = (double*)h->b_x.p;
  if (h->has_slater) {
    const int nel[2] = {h->nup, h->ndn};
    for (int s = 0; s < 2; ++s) {
      const size_t D = h->ndet_s[s], n = nel[s];
      const size_t cf = h->cplx ? 2 : 1;
      TRY(ensure(h, h->b_T[s], cf * W * D * n * n * sizeof(double)));
      TRY(ensure(h, h->b_dsign[s], cf * W * D * sizeof(double)));
      TRY(ensure(h, h->b_dlog[s], W * D * sizeof(double)));
      TRY(ensure(h, h->b_cache[s], W * n * 5 * h->nmo[s] * sizeof(double)));
      h->st.T[s] = (double*)h->b_T[s].p;
      h->st.dsign[s] = (double*)h->b_dsign[s].p;
      h->st.dlog[s] = (double*)h->b_dlog[s].p;
      h->st.cache[s] = (double*)h->b_cache[s].p;
    }
  }
  TRY(ensure(h, h->b_j3u, W * sizeof(double)));
  if (h->has_j2) {
    TRY(ensure(h, h->b_aval, (size_t)W * h->natom * h->na * 2 * sizeof(double)));
    TRY(ensure(h, h->b_bval, (size_t)W * h->nb * 3 * sizeof(double)));
    h->js.avalues = (double*)h->b_aval.p;
    h->js.bvalues = (double*)h->b_bval.p;
  }
  TRY(ensure(h, h->b_sign, (h->cplx ? 2 : 1) * W * sizeof(double)));
  TRY(ensure(h, h->b_log, W * sizeof(double)));
  TRY(ensure(h, h->b_ju, W * sizeof(double)));
  TRY(ensure(h, h->b_mask, W));
  return 0;
}

static int jas_refresh(pqa_handle* h) {
  if (h->has_j2 && h->jas_stale && h->W > 0) {
    hipLaunchKernelGGL((k_jastrow_recompute<>), dim3((unsigned)h->W), dim3(64), 0, h->stream, h->S, h->js);
    TRY(check_launch(h, "k_jastrow_recompute"));
  }
  h->jas_stale = false;
  return 0;
}


static int slater_rebuild(pqa_handle* h) {  // cache + inverse + determinants from js.x
  const int nel[2] = {h->nup, h->ndn};
  for (int s = 0; s < 2; ++s) {
    if (nel[s] == 0) {  // an empty spin channel (fully polarised system): the determinant of the 0 x 0 matrix is 1
      const size_t nd = (size_t)h->W * std::max(h->ndet_s[s], 1);
      std::vector<double> one((h->cplx ? 2 : 1) * nd, 0.0);
      for (size_t k = 0; k < nd; ++k) one[(h->cplx ? 2 : 1) * k] = 1.0;
      HIPCHK(hipStreamSynchronize(h->stream));
      HIPCHK(hipMemcpy(h->st.dsign[s], one.data(), one.size() * sizeof(double), hipMemcpyHostToDevice));
      HIPCHK(hipMemset(h->st.dlog[s], 0, nd * sizeof(double)));
      continue;
    }
    PointAddr pa;
    pa.base = h->js.x + (size_t)(s ? h->nup : 0) * 3;
    pa.group = nel[s];
    pa.group_stride = (long)h->N * 3;
    TRY(launch_orb(h, s, pa, h->W * nel[s], 5, h->st.cache[s]));
    const size_t lds = (h->cplx ? 2 : 1) * ((size_t)nel[s] * (nel[s] + 1)) * sizeof(double) + (size_t)nel[s] * sizeof(int) + 16;
    if (!h->cplx && lds > 64 * 1024 && !h->invert_attr) {  // (91 electrons of a spin and more: past the default dynamic-LDS limit)
      HIPCHK(hipFuncSetAttribute((const void*)k_build_invert<0>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
      h->invert_attr = true;
    }
    if (h->cplx && nel[s] > PQA_MAXN_FAST) {  // the complex tile does not fit LDS: scratch matrices in global memory, <= 512 MB per pass
      const size_t per = (size_t)2 * nel[s] * (nel[s] + 1) * sizeof(double) * h->ndet_s[s];
      const long wchunk = std::max<long>(1, std::min<long>(h->W, (long)(((size_t)512 << 20) / per)));
      TRY(ensure(h, h->b_ao, (size_t)wchunk * per));
      for (long w0 = 0; w0 < h->W; w0 += wchunk) {
        const long nw = std::min(wchunk, h->W - w0);
        hipLaunchKernelGGL((k_build_invert_cg<>), dim3((unsigned)(nw * h->ndet_s[s])), dim3(64), (size_t)nel[s] * sizeof(int) + 16, h->stream, h->S, h->st, s, w0,
                           (double*)h->b_ao.p);
      }
    } else
    if (h->cplx) hipLaunchKernelGGL((k_build_invert_c<>), dim3((unsigned)(h->W * h->ndet_s[s])), dim3(64), lds, h->stream, h->S, h->st, s, h->W);
    else hipLaunchKernelGGL((k_build_invert<>), dim3((unsigned)(h->W * h->ndet_s[s])), dim3(64), lds, h->stream, h->S, h->st, s, h->W);
    TRY(check_launch(h, "k_build_invert"));
  }
  return 0;
}

static int slater_value_dev(pqa_handle* h) {
  if (h->cplx) hipLaunchKernelGGL((k_slater_value_c<>), dim3((unsigned)h->W), dim3(64), 0, h->stream, h->S, h->st, (double*)h->b_sign.p, (double*)h->b_log.p);
  else hipLaunchKernelGGL((k_slater_value<>), dim3((unsigned)h->W), dim3(64), 0, h->stream, h->S, h->st, (double*)h->b_sign.p, (double*)h->b_log.p);
  return check_launch(h, "k_slater_value");
}

extern "C" int pqa_slater_recompute(pqa_handle_t* h, const double* configs, int64_t W, double* sign, double* logabs) {
  h->dmc_old_valid = false;  // (the state pqa_dmc_continue refers to is gone)
  TRY(sync_aos(h));
  HIPCHK(hipSetDevice(h->device));
  if (!h->has_slater) FAIL("handle has no Slater factor");
  if (!h->has_jastrow || h->W != W) {
    TRY(ensure_walkers(h, W));
    TRY(copy_in(h, h->js.x, configs, (size_t)W * h->N * 3 * sizeof(double)));
    TRY(slater_rebuild(h));
  } else {
    // the Jastrow factor owns the stored walker coordinates; evaluate from a scratch copy
    h->saved_valid = false;
    double* keep = h->js.x;
    TRY(ensure(h, h->b_pts, (size_t)W * h->N * 3 * sizeof(double)));
    TRY(copy_in(h, h->b_pts.p, configs, (size_t)W * h->N * 3 * sizeof(double)));
    h->js.x = (double*)h->b_pts.p;
    int rc = slater_rebuild(h);
    h->js.x = keep;
    if (rc) return rc;
  }
  return pqa_slater_value(h, sign, logabs);
}

extern "C" int pqa_slater_value(pqa_handle_t* h, double* sign, double* logabs) {
  TRY(sync_aos(h));
  HIPCHK(hipSetDevice(h->device));
  if (!h->has_slater || h->W == 0) FAIL("Slater state not initialised (call recompute)");
  TRY(slater_value_dev(h));
  TRY(copy_in(h, sign, h->b_sign.p, (h->cplx ? 2 : 1) * h->W * sizeof(double)));
  return copy_out(h, logabs, h->b_log.p, h->W * sizeof(double));
}

extern "C" int pqa_slater_eval(pqa_handle_t* h, int e, const double* pts, int64_t nrow, int npt, const int32_t* widx,
                               int ncomp, int keep_saved, double* out) {
  TRY(sync_aos(h));
  HIPCHK(hipSetDevice(h->device));
  if (!h->has_slater || h->W == 0) FAIL("Slater state not initialised (call recompute)");
  if (e < 0 || e >= h->N) FAIL("electron index out of range");
  if (ncomp != 1 && ncomp != 5) FAIL("ncomp must be 1 or 5");
  if (nrow <= 0 || npt <= 0) return 0;
  if (!widx && nrow != h->W) FAIL("nrow must equal the number of walkers when widx is NULL");
  const int s = e >= h->nup, nmo = h->nmo[s];
  const long P = nrow * npt;
  h->saved_valid = false;
  TRY(ensure(h, h->b_pts, (size_t)P * 3 * sizeof(double)));
  TRY(ensure(h, h->b_motmp, (size_t)P * ncomp * nmo * sizeof(double)));
  const size_t cf = h->cplx ? 2 : 1;
  TRY(ensure(h, h->b_out, cf * (size_t)P * ncomp * sizeof(double)));
  TRY(copy_in(h, h->b_pts.p, pts, (size_t)P * 3 * sizeof(double)));
  const int* dw = nullptr;
  if (widx) {
    TRY(ensure(h, h->b_widx, (size_t)nrow * sizeof(int)));
    TRY(copy_in(h, h->b_widx.p, widx, (size_t)nrow * sizeof(int)));
    dw = (const int*)h->b_widx.p;
  }
  TRY(launch_orb(h, s, plain_points((const double*)h->b_pts.p, P), P, ncomp, (double*)h->b_motmp.p));
  const dim3 grid((unsigned)nrow), block(64);
  if (h->cplx) {
    if (ncomp == 1)
      hipLaunchKernelGGL(k_slater_eval_c<1>, grid, block, 2 * lds_det(h, 1), h->stream, h->S, h->st, e, (const double*)h->b_motmp.p,
                         (long)nrow, npt, dw, (double*)h->b_out.p);
    else
      hipLaunchKernelGGL(k_slater_eval_c<5>, grid, block, 2 * lds_det(h, 5), h->stream, h->S, h->st, e, (const double*)h->b_motmp.p,
                         (long)nrow, npt, dw, (double*)h->b_out.p);
  } else
  if (ncomp == 1)
    hipLaunchKernelGGL(k_slater_eval<1>, grid, block, lds_det(h, 1), h->stream, h->S, h->st, e, (const double*)h->b_motmp.p,
                       (long)nrow, npt, dw, (double*)h->b_out.p);
  else
    hipLaunchKernelGGL(k_slater_eval<5>, grid, block, lds_det(h, 5), h->stream, h->S, h->st, e, (const double*)h->b_motmp.p,
                       (long)nrow, npt, dw, (double*)h->b_out.p);
  TRY(check_launch(h, "k_slater_eval"));
  TRY(copy_out(h, out, h->b_out.p, cf * (size_t)P * ncomp * sizeof(double)));
  if (keep_saved && npt == 1 && !widx && ncomp == 5) { h->saved_valid = true; h->saved_e = e; }
  return 0;
}

extern "C" int pqa_testvalue_many(pqa_handle_t* h, const int32_t* es, int ne, const double* pts, int64_t nrow, const int32_t* widx,
                                  int factors, double* out) {
  TRY(sync_aos(h));
  HIPCHK(hipSetDevice(h->device));
  if (h->W == 0) FAIL("state not initialised (call recompute)");
  if (nrow <= 0 || ne <= 0) return 0;
  if (!widx && nrow != h->W) FAIL("nrow must equal the number of walkers when widx is NULL");
  if ((factors & 1) && !h->has_slater) FAIL("handle has no Slater factor");
  if ((factors & 2) && !h->has_j2) FAIL("handle has no two-body Jastrow factor");
  if ((factors & 4) && !h->has_j3) FAIL("handle has no three-body Jastrow factor");
  if (!(factors & 7)) FAIL("no factor selected");
  std::vector<int> he((size_t)ne);
  HIPCHK(hipMemcpy(he.data(), es, (size_t)ne * sizeof(int), hipMemcpyDefault));
  for (int e : he)
    if (e < 0 || e >= h->N) FAIL("electron index out of range");
  h->saved_valid = false;
  TRY(ensure(h, h->b_pts, (size_t)nrow * 3 * sizeof(double)));
  const size_t cf = h->cplx ? 2 : 1;  // complex handles return (re, im) pairs
  TRY(ensure(h, h->b_out, cf * nrow * ne * sizeof(double)));
  TRY(ensure(h, h->b_tves, (size_t)ne * sizeof(int)));
  TRY(copy_in(h, h->b_pts.p, pts, (size_t)nrow * 3 * sizeof(double)));
  TRY(copy_in(h, h->b_tves.p, he.data(), (size_t)ne * sizeof(int)));
  const int* dw = nullptr;
  if (widx) {
    TRY(ensure(h, h->b_widx, (size_t)nrow * sizeof(int)));
    TRY(copy_in(h, h->b_widx.p, widx, (size_t)nrow * sizeof(int)));
    dw = (const int*)h->b_widx.p;
  }
  if (factors & 1)
    for (int s = 0; s < 2; ++s) {
      TRY(ensure(h, h->b_emo[s], (size_t)nrow * std::max(h->nmo[s], 1) * sizeof(double)));
      TRY(launch_orb(h, s, plain_points((const double*)h->b_pts.p, nrow), nrow, 1, (double*)h->b_emo[s].p));
    }
  if (h->cplx)
    hipLaunchKernelGGL(k_testvalue_many<true>, dim3((unsigned)nrow), dim3(64), 2 * lds_det(h, 1), h->stream, h->S, h->st, h->js,
                       (const int*)h->b_tves.p, ne, (const double*)h->b_pts.p, (const double*)h->b_emo[0].p,
                       (const double*)h->b_emo[1].p, (long)nrow, dw, factors, (double*)h->b_out.p);
  else
    hipLaunchKernelGGL(k_testvalue_many<false>, dim3((unsigned)nrow), dim3(64), lds_det(h, 1), h->stream, h->S, h->st, h->js,
                       (const int*)h->b_tves.p, ne, (const double*)h->b_pts.p, (const double*)h->b_emo[0].p,
                       (const double*)h->b_emo[1].p, (long)nrow, dw, factors, (double*)h->b_out.p);
  TRY(check_launch(h, "k_testvalue_many"));
  return copy_out(h, out, h->b_out.p, cf * nrow * ne * sizeof(double));
}

extern "C" int pqa_slater_pgradient(pqa_handle_t* h, double* d_det, double* d_mo_up, double* d_mo_dn) {
  TRY(sync_aos(h));
  HIPCHK(hipSetDevice(h->device));
  if (!h->has_slater || h->W == 0) FAIL("Slater state not initialised (call recompute)");
  const long W = h->W;
  const size_t cf = h->cplx ? 2 : 1;  // complex handles: every output is complex, (re, im) interleaved; nmo below counts orbitals
  TRY(slater_value_dev(h));  // sign (complex: phase) / log of the determinant expansion -> b_sign, b_log
  TRY(ensure(h, h->b_pgdet, cf * W * h->ndet * sizeof(double)));
  const dim3 gd((unsigned)((W * h->ndet + 255) / 256));
  if (h->cplx) hipLaunchKernelGGL((k_pgrad_det_c<>), gd, dim3(256), 0, h->stream, h->S, h->st, (const double*)h->b_sign.p, (const double*)h->b_log.p, W, (double*)h->b_pgdet.p);
  else hipLaunchKernelGGL((k_pgrad_det<>), gd, dim3(256), 0, h->stream, h->S, h->st, (const double*)h->b_sign.p, (const double*)h->b_log.p, W, (double*)h->b_pgdet.p);
  TRY(check_launch(h, "k_pgrad_det"));
  if (d_det) TRY(copy_out(h, d_det, h->b_pgdet.p, cf * W * h->ndet * sizeof(double)));
  double* outs[2] = {d_mo_up, d_mo_dn};
  if (!d_mo_up && !d_mo_dn) return 0;
  // AO values of every electron at its position: real, or for a twisted cell complex (two planes; the walkers of a twisted
  // handle are unfolded: k_ao_tw folds every point and gives it its wrap phase)
  const long ao_plane = W * h->N * (long)h->nao;
  const size_t nao_all = (size_t)ao_plane * (h->twist ? 2 : 1);
  if (nao_all * sizeof(double) > ((size_t)16 << 30)) FAIL("orbital-coefficient gradients need the AO values of all electrons: too many walkers for one call");
  TRY(ensure(h, h->b_ao, nao_all * sizeof(double)));
  const dim3 ga((unsigned)((W * h->N + 63) / 64));
  if (h->twist) hipLaunchKernelGGL((k_ao_tw<>), ga, dim3(64), 0, h->stream, h->S, (const double*)h->js.x, W * h->N, (double*)h->b_ao.p);
  else TRY(launch_ao(h, plain_points((const double*)h->js.x, W * h->N), W * h->N, 1, (double*)h->b_ao.p));
  TRY(check_launch(h, "k_ao"));
  for (int s = 0; s < 2; ++s) {
    const int n = s ? h->ndn : h->nup;
    if (!outs[s] || n == 0 || h->nmo[s] == 0) continue;
    const size_t nout = (size_t)W * h->nao * h->nmo[s];  // (complex: nmo[s] = 2 x orbitals, i.e. already the doubles of the complex result)
    TRY(ensure(h, h->b_out, nout * sizeof(double)));
    const size_t lds = cf * (size_t)h->ndet_s[s] * sizeof(double);
    if (!h->cplx) hipLaunchKernelGGL((k_pgrad_mo<>), dim3((unsigned)W), dim3(256), lds, h->stream, h->S, h->st, s,
                                     (const double*)h->b_ao.p, (const double*)h->b_pgdet.p, (const int*)h->d_colmap[s], (double*)h->b_out.p);
    else if (h->twist) hipLaunchKernelGGL(k_pgrad_mo_c<true>, dim3((unsigned)W), dim3(256), lds, h->stream, h->S, h->st, s, (const double*)h->b_ao.p, ao_plane,
                                          (const double*)h->b_pgdet.p, (const int*)h->d_colmap[s], (double*)h->b_out.p);
    else hipLaunchKernelGGL(k_pgrad_mo_c<false>, dim3((unsigned)W), dim3(256), lds, h->stream, h->S, h->st, s, (const double*)h->b_ao.p, ao_plane,
                            (const double*)h->b_pgdet.p, (const int*)h->d_colmap[s], (double*)h->b_out.p);
    TRY(check_launch(h, "k_pgrad_mo"));
    TRY(copy_out(h, outs[s], h->b_out.p, nout * sizeof(double)));
  }
  return 0;
}

extern "C" int pqa_slater_has_zero(pqa_handle_t* h, int spin, int* flag) {
  TRY(sync_aos(h));
  HIPCHK(hipSetDevice(h->device));
  if (!h->has_slater || h->W == 0) FAIL("Slater state not initialised (call recompute)");
  TRY(ensure(h, h->b_flag, sizeof(int)));
  HIPCHK(hipMemsetAsync(h->b_flag.p, 0, sizeof(int), h->stream));
  const long count = h->W * h->ndet_s[spin];
  hipLaunchKernelGGL((k_has_zero<>), dim3((unsigned)((count + 255) / 256)), dim3(256), 0, h->stream, (const double*)h->st.dlog[spin], count, (int*)h->b_flag.p);
  TRY(check_launch(h, "k_has_zero"));
  return copy_out(h, flag, h->b_flag.p, sizeof(int));
}

extern "C" int pqa_slater_update(pqa_handle_t* h, int e, const double* epos, const uint8_t* mask, int use_saved) {
  h->dmc_old_valid = false;  // (the state pqa_dmc_continue refers to is gone)
  TRY(sync_aos(h));
  HIPCHK(hipSetDevice(h->device));
  if (!h->has_slater || h->W == 0) FAIL("Slater state not initialised (call recompute)");
  if (e < 0 || e >= h->N) FAIL("electron index out of range");
  const int s = e >= h->nup, nmo = h->nmo[s];
  const long W = h->W;
  if (!(use_saved && h->saved_valid && h->saved_e == e)) {
    TRY(ensure(h, h->b_pts, (size_t)W * 3 * sizeof(double)));
    TRY(ensure(h, h->b_motmp, (size_t)W * 5 * nmo * sizeof(double)));
    TRY(copy_in(h, h->b_pts.p, epos, (size_t)W * 3 * sizeof(double)));
    TRY(launch_orb(h, s, plain_points((const double*)h->b_pts.p, W), W, 5, (double*)h->b_motmp.p));
  }
  h->saved_valid = false;
  const uint8_t* dm = nullptr;
  if (mask) {
    TRY(copy_in(h, h->b_mask.p, mask, (size_t)W));
    dm = (const uint8_t*)h->b_mask.p;
  }
  if (h->cplx) hipLaunchKernelGGL((k_sm_update_c<>), dim3((unsigned)W), dim3(64), 2 * lds_sm(h), h->stream, h->S, h->st, e,
                                  (const double*)h->b_motmp.p, 5 * nmo, dm, 1);
  else hipLaunchKernelGGL((k_sm_update<>), dim3((unsigned)W), dim3(64), lds_sm(h), h->stream, h->S, h->st, e, (const double*)h->b_motmp.p,
                          5 * nmo, dm, 1);
  TRY(check_launch(h, "k_sm_update"));
  HIPCHK(hipStreamSynchronize(h->stream));
  return 0;
}

extern "C" int pqa_slater_get_state(pqa_handle_t* h, int spin, double* inverse, double* dets) {
  TRY(sync_aos(h));
  HIPCHK(hipSetDevice(h->device));
  if (!h->has_slater || h->W == 0) FAIL("Slater state not initialised (call recompute)");
  const size_t W = h->W, D = h->ndet_s[spin], n = spin ? h->ndn : h->nup;
  HIPCHK(hipStreamSynchronize(h->stream));
  const size_t cf = h->cplx ? 2 : 1;  // complex: (re, im) interleaved in every output
  if (inverse) {
    std::vector<double> T(cf * W * D * n * n), inv(cf * W * D * n * n);
    HIPCHK(hipMemcpy(T.data(), h->st.T[spin], T.size() * sizeof(double), hipMemcpyDeviceToHost));
    for (size_t m = 0; m < W * D; ++m)
      for (size_t i = 0; i < n; ++i)
        for (size_t k = 0; k < n; ++k)
          for (size_t q = 0; q < cf; ++q) inv[((m * n + k) * n + i) * cf + q] = T[((m * n + i) * n + k) * cf + q];
    HIPCHK(hipMemcpy(inverse, inv.data(), inv.size() * sizeof(double), hipMemcpyDefault));
  }
  if (dets) {  // [phase (W,D) (complex: interleaved)] followed by [log (W,D)]
    std::vector<double> d((cf + 1) * W * D);
    HIPCHK(hipMemcpy(d.data(), h->st.dsign[spin], cf * W * D * sizeof(double), hipMemcpyDeviceToHost));
    HIPCHK(hipMemcpy(d.data() + cf * W * D, h->st.dlog[spin], W * D * sizeof(double), hipMemcpyDeviceToHost));
    HIPCHK(hipMemcpy(dets, d.data(), d.size() * sizeof(double), hipMemcpyDefault));
  }
  return 0;
}

// ---------------------------------------------------------------- Jastrow
extern "C" int pqa_jastrow_recompute(pqa_handle_t* h, const double* configs, int64_t W, double* logval) {
  h->dmc_old_valid = false;  // (the state pqa_dmc_continue refers to is gone)
  TRY(sync_aos(h));
  HIPCHK(hipSetDevice(h->device));
  if (!h->has_j2) FAIL("handle has no two-body Jastrow factor");
  if (h->W != W) TRY(ensure_walkers(h, W));
  h->jas_stale = false;
  TRY(copy_in(h, h->js.x, configs, (size_t)W * h->N * 3 * sizeof(double)));
  hipLaunchKernelGGL((k_jastrow_recompute<>), dim3((unsigned)W), dim3(64), 0, h->stream, h->S, h->js);
  TRY(check_launch(h, "k_jastrow_recompute"));
  return pqa_jastrow_value(h, logval);
}

extern "C" int pqa_jastrow_value(pqa_handle_t* h, double* logval) {
  TRY(sync_aos(h));
  HIPCHK(hipSetDevice(h->device));
  if (!h->has_j2 || h->W == 0) FAIL("Jastrow state not initialised (call recompute)");
  TRY(jas_refresh(h));
  hipLaunchKernelGGL((k_jastrow_value<>), dim3((unsigned)h->W), dim3(64), 0, h->stream, h->S, h->js, (double*)h->b_ju.p);
  TRY(check_launch(h, "k_jastrow_value"));
  return copy_out(h, logval, h->b_ju.p, h->W * sizeof(double));
}

static int jastrow_eval_parts(pqa_handle_t* h, int parts, int e, const double* pts, int64_t nrow, int npt, const int32_t* widx,
                              int mode, double* out) {
  HIPCHK(hipSetDevice(h->device));
  if (!((parts & 1) ? h->has_j2 : h->has_j3) || h->W == 0) FAIL("Jastrow state not initialised (call recompute)");
  if (e < 0 || e >= h->N) FAIL("electron index out of range");
  if (mode < 0 || mode > 2 || (mode > 0 && npt != 1)) FAIL("bad mode / npt combination");
  if (nrow <= 0 || npt <= 0) return 0;
  if (!widx && nrow != h->W) FAIL("nrow must equal the number of walkers when widx is NULL");
  const long P = nrow * npt;
  const size_t nout = mode == 0 ? (size_t)P : (size_t)4 * nrow;
  TRY(ensure(h, h->b_pts, (size_t)P * 3 * sizeof(double)));
  TRY(ensure(h, h->b_out, nout * sizeof(double)));
  TRY(copy_in(h, h->b_pts.p, pts, (size_t)P * 3 * sizeof(double)));
  const int* dw = nullptr;
  if (widx) {
    TRY(ensure(h, h->b_widx, (size_t)nrow * sizeof(int)));
    TRY(copy_in(h, h->b_widx.p, widx, (size_t)nrow * sizeof(int)));
    dw = (const int*)h->b_widx.p;
  }
  hipLaunchKernelGGL((k_jastrow_eval<>), dim3((unsigned)nrow), dim3(64), lds_j3(h), h->stream, h->S, h->js, e, (const double*)h->b_pts.p,
                     (long)nrow, npt, dw, mode, parts, (double*)h->b_out.p);
  TRY(check_launch(h, "k_jastrow_eval"));
  return copy_out(h, out, h->b_out.p, nout * sizeof(double));
}

extern "C" int pqa_jastrow_eval(pqa_handle_t* h, int e, const double* pts, int64_t nrow, int npt, const int32_t* widx,
                                int mode, double* out) {
  TRY(sync_aos(h));
  return jastrow_eval_parts(h, 1, e, pts, nrow, npt, widx, mode, out);
}

extern "C" int pqa_j3_eval(pqa_handle_t* h, int e, const double* pts, int64_t nrow, int npt, const int32_t* widx, int mode,
                           double* out) {
  TRY(sync_aos(h));
  return jastrow_eval_parts(h, 2, e, pts, nrow, npt, widx, mode, out);
}

static int j3_value_dev(pqa_handle* h) {
  hipLaunchKernelGGL((k_j3_value<>), dim3((unsigned)h->W), dim3(64), lds_j3(h), h->stream, h->S, h->js, (double*)h->b_j3u.p);
  return check_launch(h, "k_j3_value");
}

extern "C" int pqa_j3_value(pqa_handle_t* h, double* logval) {
  TRY(sync_aos(h));
  HIPCHK(hipSetDevice(h->device));
  if (!h->has_j3 || h->W == 0) FAIL("three-body Jastrow state not initialised (call recompute)");
  TRY(j3_value_dev(h));
  return copy_out(h, logval, h->b_j3u.p, h->W * sizeof(double));
}

extern "C" int pqa_j3_pgradient(pqa_handle_t* h, double* d_ccoeff) {
  TRY(sync_aos(h));
  HIPCHK(hipSetDevice(h->device));
  if (!h->has_j3 || h->W == 0) FAIL("three-body Jastrow state not initialised (call recompute)");
  const long W = h->W;
  const size_t E = (size_t)h->natom * h->na3 * h->na3 * h->nb3 * 3;
  const size_t lds = ((size_t)h->N * h->natom * h->na3 + (size_t)h->N * (h->N - 1) / 2 * h->nb3) * sizeof(double);
  if (lds > 150 * 1024) FAIL("three-body parameter gradient: the a/b value tables of one walker do not fit LDS");
  TRY(ensure(h, h->b_out, (size_t)W * E * sizeof(double)));
  HIPCHK(hipFuncSetAttribute(reinterpret_cast<const void*>(k_j3_pgrad<>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  hipLaunchKernelGGL((k_j3_pgrad<>), dim3((unsigned)W), dim3(64), lds, h->stream, h->S, h->js, (double*)h->b_out.p);
  TRY(check_launch(h, "k_j3_pgrad"));
  return copy_out(h, d_ccoeff, h->b_out.p, (size_t)W * E * sizeof(double));
}

extern "C" int pqa_j3_recompute(pqa_handle_t* h, const double* configs, int64_t W, double* logval) {
  h->dmc_old_valid = false;  // (the state pqa_dmc_continue refers to is gone)
  TRY(sync_aos(h));
  HIPCHK(hipSetDevice(h->device));
  if (!h->has_j3) FAIL("handle has no three-body Jastrow factor");
  if (h->has_j2 && h->W == W) {  // the two-body factor owns the stored coordinates: evaluate from a scratch copy
    double* keep = h->js.x;
    TRY(ensure(h, h->b_pts, (size_t)W * h->N * 3 * sizeof(double)));
    TRY(copy_in(h, h->b_pts.p, configs, (size_t)W * h->N * 3 * sizeof(double)));
    h->js.x = (double*)h->b_pts.p;
    int rc = j3_value_dev(h);
    h->js.x = keep;
    if (rc) return rc;
    return copy_out(h, logval, h->b_j3u.p, W * sizeof(double));
  }
  if (h->W != W) TRY(ensure_walkers(h, W));
  TRY(copy_in(h, h->js.x, configs, (size_t)W * h->N * 3 * sizeof(double)));
  return pqa_j3_value(h, logval);
}

extern "C" int pqa_j3_update(pqa_handle_t* h, int e, const double* epos, const uint8_t* mask) {
  h->dmc_old_valid = false;  // (the state pqa_dmc_continue refers to is gone)
  TRY(sync_aos(h));
  HIPCHK(hipSetDevice(h->device));
  if (!h->has_j3 || h->W == 0) FAIL("three-body Jastrow state not initialised (call recompute)");
  if (e < 0 || e >= h->N) FAIL("electron index out of range");
  if (h->has_j2) return 0;  // coordinates are moved by the two-body factor's update
  const long W = h->W;
  TRY(ensure(h, h->b_newpos, (size_t)W * 3 * sizeof(double)));
  TRY(copy_in(h, h->b_newpos.p, epos, (size_t)W * 3 * sizeof(double)));
  const uint8_t* dm = nullptr;
  if (mask) {
    TRY(copy_in(h, h->b_mask.p, mask, (size_t)W));
    dm = (const uint8_t*)h->b_mask.p;
  }
  hipLaunchKernelGGL((k_move_x<>), dim3((unsigned)((W + 255) / 256)), dim3(256), 0, h->stream, h->js, h->N, e, (const double*)h->b_newpos.p, dm, W);
  TRY(check_launch(h, "k_move_x"));
  HIPCHK(hipStreamSynchronize(h->stream));
  return 0;
}

extern "C" int pqa_jastrow_update(pqa_handle_t* h, int e, const double* epos, const uint8_t* mask) {
  h->dmc_old_valid = false;  // (the state pqa_dmc_continue refers to is gone)
  TRY(sync_aos(h));
  HIPCHK(hipSetDevice(h->device));
  if (!h->has_j2 || h->W == 0) FAIL("Jastrow state not initialised (call recompute)");
  if (e < 0 || e >= h->N) FAIL("electron index out of range");
  const long W = h->W;
  TRY(jas_refresh(h));
  TRY(ensure(h, h->b_newpos, (size_t)W * 3 * sizeof(double)));
  TRY(copy_in(h, h->b_newpos.p, epos, (size_t)W * 3 * sizeof(double)));
  const uint8_t* dm = nullptr;
  if (mask) {
    TRY(copy_in(h, h->b_mask.p, mask, (size_t)W));
    dm = (const uint8_t*)h->b_mask.p;
  }
  hipLaunchKernelGGL((k_jastrow_update<>), dim3((unsigned)W), dim3(64), 0, h->stream, h->S, h->js, e, (const double*)h->b_newpos.p, dm);
  TRY(check_launch(h, "k_jastrow_update"));
  HIPCHK(hipStreamSynchronize(h->stream));
  return 0;
}

// ---------------------------------------------------------------- product wave function: one call per protocol method
// MultiplyWF.gradient / gradient_value / gradient_laplacian and updateinternals (multiplywf.py:102-129) of a Slater x two-body-Jastrow
// product living on this handle.  The per-factor entries above cost a copy-in, a launch chain, a copy-out and a synchronisation EACH
// (tools/protocol_profile.py: ~60 us per call at 4096 walkers, seven calls per electron move of pyqmc.method.mc.vmc_worker); here
// the proposal goes in once and both factors' rows come back in one transfer.
extern "C" int pqa_wf_eval(pqa_handle_t* h, int e, const double* pts, int jmode, int keep_saved, double* out) {
  TRY(sync_aos(h));
  HIPCHK(hipSetDevice(h->device));
  if (!h->has_slater || !h->has_j2 || h->has_j3 || h->cplx || h->W == 0) FAIL("pqa_wf_eval: a real Slater x two-body-Jastrow product with resident walkers");
  if (e < 0 || e >= h->N) FAIL("electron index out of range");
  if (jmode != 1 && jmode != 2) FAIL("jmode: 1 (Jastrow gradient + value) or 2 (gradient + laplacian)");
  const int s = e >= h->nup, nmo = h->nmo[s];
  const long W = h->W;
  h->saved_valid = false;
  TRY(ensure(h, h->b_pts, (size_t)W * 3 * sizeof(double)));
  TRY(ensure(h, h->b_motmp, (size_t)W * 5 * nmo * sizeof(double)));
  TRY(ensure(h, h->b_out, (size_t)9 * W * sizeof(double)));
  TRY(copy_in(h, h->b_pts.p, pts, (size_t)W * 3 * sizeof(double)));
  TRY(launch_orb(h, s, plain_points((const double*)h->b_pts.p, W), W, 5, (double*)h->b_motmp.p));
  hipLaunchKernelGGL(k_slater_eval<5>, dim3((unsigned)W), dim3(64), lds_det(h, 5), h->stream, h->S, h->st, e, (const double*)h->b_motmp.p, W, 1,
                     (const int*)nullptr, (double*)h->b_out.p);
  hipLaunchKernelGGL((k_jastrow_eval<>), dim3((unsigned)W), dim3(64), lds_j3(h), h->stream, h->S, h->js, e, (const double*)h->b_pts.p, W, 1,
                     (const int*)nullptr, jmode, 1, (double*)h->b_out.p + (size_t)5 * W);
  TRY(check_launch(h, "k_slater_eval / k_jastrow_eval"));
  TRY(copy_out(h, out, h->b_out.p, (size_t)9 * W * sizeof(double)));
  if (keep_saved) { h->saved_valid = true; h->saved_e = e; }
  return 0;
}

extern "C" int pqa_wf_update(pqa_handle_t* h, int e, const double* epos, const uint8_t* mask, int use_saved, int* has_zero) {
  h->dmc_old_valid = false;  // (the state pqa_dmc_continue refers to is gone)
  TRY(sync_aos(h));
  HIPCHK(hipSetDevice(h->device));
  if (!h->has_slater || !h->has_j2 || h->has_j3 || h->cplx || h->W == 0) FAIL("pqa_wf_update: a real Slater x two-body-Jastrow product with resident walkers");
  if (e < 0 || e >= h->N) FAIL("electron index out of range");
  const int s = e >= h->nup, nmo = h->nmo[s];
  const long W = h->W;
  TRY(jas_refresh(h));
  TRY(ensure(h, h->b_newpos, (size_t)W * 3 * sizeof(double)));
  TRY(copy_in(h, h->b_newpos.p, epos, (size_t)W * 3 * sizeof(double)));
  if (!(use_saved && h->saved_valid && h->saved_e == e)) {
    TRY(ensure(h, h->b_motmp, (size_t)W * 5 * nmo * sizeof(double)));
    TRY(launch_orb(h, s, plain_points((const double*)h->b_newpos.p, W), W, 5, (double*)h->b_motmp.p));
  }
  h->saved_valid = false;
  const uint8_t* dm = nullptr;
  if (mask) {
    TRY(copy_in(h, h->b_mask.p, mask, (size_t)W));
    dm = (const uint8_t*)h->b_mask.p;
  }
  hipLaunchKernelGGL((k_sm_update<>), dim3((unsigned)W), dim3(64), lds_sm(h), h->stream, h->S, h->st, e, (const double*)h->b_motmp.p, 5 * nmo, dm, 1);
  hipLaunchKernelGGL((k_jastrow_update<>), dim3((unsigned)W), dim3(64), 0, h->stream, h->S, h->js, e, (const double*)h->b_newpos.p, dm);
  // slater.py:269-275 looks for a vanished determinant BEFORE an update; the flag of the state this call leaves is what the next
  // update of this spin needs (the caller keeps it), and it travels with the synchronisation the call ends with anyway
  TRY(ensure(h, h->b_flag, sizeof(int)));
  HIPCHK(hipMemsetAsync(h->b_flag.p, 0, sizeof(int), h->stream));
  const long count = W * h->ndet_s[s];
  hipLaunchKernelGGL((k_has_zero<>), dim3((unsigned)((count + 255) / 256)), dim3(256), 0, h->stream, (const double*)h->st.dlog[s], count, (int*)h->b_flag.p);
  TRY(check_launch(h, "k_sm_update / k_jastrow_update"));
  return copy_out(h, has_zero, h->b_flag.p, sizeof(int));
}

extern "C" int pqa_jastrow_get_state(pqa_handle_t* h, double* avalues, double* bvalues, double* configs) {
  TRY(sync_aos(h));
  HIPCHK(hipSetDevice(h->device));
  if (h->W == 0) FAIL("state not initialised (call recompute)");
  TRY(jas_refresh(h));
  HIPCHK(hipStreamSynchronize(h->stream));
  if (avalues && h->has_j2) HIPCHK(hipMemcpy(avalues, h->js.avalues, (size_t)h->W * h->natom * h->na * 2 * sizeof(double), hipMemcpyDefault));
  if (bvalues && h->has_j2) HIPCHK(hipMemcpy(bvalues, h->js.bvalues, (size_t)h->W * h->nb * 3 * sizeof(double), hipMemcpyDefault));
  if (configs) HIPCHK(hipMemcpy(configs, h->js.x, (size_t)h->W * h->N * 3 * sizeof(double), hipMemcpyDefault));
  return 0;
}

// ---------------------------------------------------------------- fused path
static int wf_value_host(pqa_handle* h, double* sign, double* logabs) {
  const long W = h->W;
  const size_t cf = h->cplx ? 2 : 1;
  std::vector<double> sg(cf * W, 1.0), lg(W, 0.0), ju(W, 0.0);
  if (h->has_slater) {
    TRY(slater_value_dev(h));
    TRY(copy_in(h, sg.data(), h->b_sign.p, cf * W * sizeof(double)));
    TRY(copy_in(h, lg.data(), h->b_log.p, W * sizeof(double)));
  }
  std::vector<double> j3u(W, 0.0);
  if (h->has_j2) {
    TRY(jas_refresh(h));
    hipLaunchKernelGGL((k_jastrow_value<>), dim3((unsigned)W), dim3(64), 0, h->stream, h->S, h->js, (double*)h->b_ju.p);
    TRY(check_launch(h, "k_jastrow_value"));
    TRY(copy_in(h, ju.data(), h->b_ju.p, W * sizeof(double)));
  }
  if (h->has_j3) {
    TRY(j3_value_dev(h));
    TRY(copy_in(h, j3u.data(), h->b_j3u.p, W * sizeof(double)));
  }
  HIPCHK(hipStreamSynchronize(h->stream));
  for (long w = 0; w < W; ++w) lg[w] += ju[w] + j3u[w];
  if (sign) HIPCHK(hipMemcpy(sign, sg.data(), cf * W * sizeof(double), hipMemcpyDefault));
  if (logabs) HIPCHK(hipMemcpy(logabs, lg.data(), W * sizeof(double), hipMemcpyDefault));
  return 0;
}

extern "C" int pqa_wf_recompute(pqa_handle_t* h, const double* configs, int64_t W, double* sign, double* logabs) {
  h->dmc_old_valid = false;  // (the state pqa_dmc_continue refers to is gone)
  TRY(sync_aos(h));
  HIPCHK(hipSetDevice(h->device));
  TRY(ensure_walkers(h, W));
  h->jas_stale = false;
  TRY(copy_in(h, h->js.x, configs, (size_t)W * h->N * 3 * sizeof(double)));
  if (h->has_slater) TRY(slater_rebuild(h));
  if (h->has_j2) {
    hipLaunchKernelGGL((k_jastrow_recompute<>), dim3((unsigned)W), dim3(64), 0, h->stream, h->S, h->js);
    TRY(check_launch(h, "k_jastrow_recompute"));
  }
  return wf_value_host(h, sign, logabs);
}

extern "C" int pqa_wf_value(pqa_handle_t* h, double* sign, double* logabs) {
  TRY(sync_aos(h));
  HIPCHK(hipSetDevice(h->device));
  if (h->W == 0) FAIL("state not initialised (call recompute)");
  return wf_value_host(h, sign, logabs);
}

extern "C" int pqa_get_configs(pqa_handle_t* h, double* configs) { return pqa_jastrow_get_state(h, nullptr, nullptr, configs); }


// ---------------------------------------------------------------- branching on the device
// dst row w <- src row idx[w]; rows of `row` doubles.  grid = (W, ceil(row / 1024)), block = 256 (4 doubles per thread)
__global__ __launch_bounds__(256) void k_gather_rows(const double* __restrict__ src, double* __restrict__ dst, const int* __restrict__ idx,
                                                     long row) {
  const long w = blockIdx.x;
  const double* s = src + (size_t)idx[w] * row;
  double* d = dst + (size_t)w * row;
  for (long k = (long)blockIdx.y * 1024 + threadIdx.x; k < row && k < ((long)blockIdx.y + 1) * 1024; k += 256) d[k] = s[k];
}
static int gather_swap(pqa_handle* h, DevBuf& cur, DevBuf& alt, const int* d_idx, size_t row_doubles) {
  if (row_doubles == 0 || !cur.p) return 0;
  TRY(ensure(h, alt, (size_t)h->W * row_doubles * sizeof(double)));
  hipLaunchKernelGGL(k_gather_rows, dim3((unsigned)h->W, (unsigned)((row_doubles + 1023) / 1024)), dim3(256), 0, h->stream,
                     (const double*)cur.p, (double*)alt.p, d_idx, (long)row_doubles);
  std::swap(cur, alt);
  return 0;
}
extern "C" int pqa_resample(pqa_handle_t* h, const int32_t* newinds) {
  h->dmc_old_valid = false;  // (the state pqa_dmc_continue refers to is gone)
  TRY(sync_aos(h));
  HIPCHK(hipSetDevice(h->device));
  if (h->W == 0) FAIL("state not initialised (call recompute)");
  if (!newinds) FAIL("pqa_resample: newinds must not be NULL");
  const long W = h->W;
  for (long w = 0; w < W; ++w)
    if (newinds[w] < 0 || newinds[w] >= W) FAIL("pqa_resample: index out of range");
  h->saved_valid = false;
  TRY(ensure(h, h->b_rsidx, (size_t)W * sizeof(int)));
  TRY(copy_in(h, h->b_rsidx.p, newinds, (size_t)W * sizeof(int)));
  const int* idx = (const int*)h->b_rsidx.p;
  const size_t cf = h->cplx ? 2 : 1;
  TRY(gather_swap(h, h->b_x, h->b_alt_x, idx, (size_t)h->N * 3));
  h->js.x = (double*)h->b_x.p;
  if (h->has_slater) {
    const int nel[2] = {h->nup, h->ndn};
    for (int s = 0; s < 2; ++s) {
      const size_t D = h->ndet_s[s], n = nel[s];
      TRY(gather_swap(h, h->b_T[s], h->b_alt_T[s], idx, cf * D * n * n));
      TRY(gather_swap(h, h->b_dsign[s], h->b_alt_dsign[s], idx, cf * D));
      TRY(gather_swap(h, h->b_dlog[s], h->b_alt_dlog[s], idx, D));
      TRY(gather_swap(h, h->b_cache[s], h->b_alt_cache[s], idx, n * 5 * h->nmo[s]));
      h->st.T[s] = (double*)h->b_T[s].p;
      h->st.dsign[s] = (double*)h->b_dsign[s].p;
      h->st.dlog[s] = (double*)h->b_dlog[s].p;
      h->st.cache[s] = (double*)h->b_cache[s].p;
    }
  }
  if (h->has_j2) {
    if (!h->jas_stale) {
      TRY(gather_swap(h, h->b_aval, h->b_alt_aval, idx, (size_t)h->natom * h->na * 2));
      TRY(gather_swap(h, h->b_bval, h->b_alt_bval, idx, (size_t)h->nb * 3));
    }
    h->js.avalues = (double*)h->b_aval.p;
    h->js.bvalues = (double*)h->b_bval.p;
  }
  TRY(gather_swap(h, h->b_j3u, h->b_alt_j3u, idx, 1));
  TRY(check_launch(h, "k_gather_rows"));
  HIPCHK(hipStreamSynchronize(h->stream));  // newinds may be freed by the caller
  return 0;
}

// ---------------------------------------------------------------- distributed branching (one walker exchange per block)
// dst row k <- src row idx[k] for k < n (k_gather_rows with a destination that is NOT one of the handle's buffers)
extern "C" int pqa_get_walkers(pqa_handle_t* h, const int32_t* idx, int64_t n, double* out) {
  TRY(sync_aos(h));
  HIPCHK(hipSetDevice(h->device));
  if (h->W == 0) FAIL("state not initialised (call recompute)");
  if (n <= 0) return 0;
  for (int64_t k = 0; k < n; ++k)
    if (idx[k] < 0 || idx[k] >= h->W) FAIL("pqa_get_walkers: index out of range");
  const size_t row = (size_t)h->N * 3;
  TRY(ensure(h, h->b_rsidx, (size_t)n * sizeof(int)));
  TRY(ensure(h, h->b_pts, (size_t)n * row * sizeof(double)));
  TRY(copy_in(h, h->b_rsidx.p, idx, (size_t)n * sizeof(int)));
  hipLaunchKernelGGL(k_gather_rows, dim3((unsigned)n, (unsigned)((row + 1023) / 1024)), dim3(256), 0, h->stream, (const double*)h->js.x,
                     (double*)h->b_pts.p, (const int*)h->b_rsidx.p, (long)row);
  TRY(check_launch(h, "k_gather_rows"));
  return copy_out(h, out, h->b_pts.p, (size_t)n * row * sizeof(double));
}

// Wave-function state of walkers [w0, w0 + n) from their coordinates: the recompute pipeline run on a view of the state.
static int recompute_range(pqa_handle* h, long w0, long n) {
  if (n <= 0) return 0;
  const JastrowState js0 = h->js;
  const SlaterState st0 = h->st;
  const long W0 = h->W;
  const size_t cf = h->cplx ? 2 : 1;
  const int nel[2] = {h->nup, h->ndn};
  h->js.x += (size_t)w0 * h->N * 3;
  if (h->has_j2) { h->js.avalues += (size_t)w0 * h->natom * h->na * 2; h->js.bvalues += (size_t)w0 * h->nb * 3; }
  if (h->has_slater)
    for (int s = 0; s < 2; ++s) {
      const size_t D = h->ndet_s[s], ne = nel[s];
      h->st.T[s] += cf * w0 * D * ne * ne; h->st.dsign[s] += cf * w0 * D; h->st.dlog[s] += (size_t)w0 * D;
      h->st.cache[s] += (size_t)w0 * ne * 5 * h->nmo[s];
    }
  h->W = n;
  int rc = 0;
  if (h->has_slater) rc = slater_rebuild(h);
  if (!rc && h->has_j2 && !h->jas_stale) {
    hipLaunchKernelGGL((k_jastrow_recompute<>), dim3((unsigned)n), dim3(64), 0, h->stream, h->S, h->js);
    rc = check_launch(h, "k_jastrow_recompute");
  }
  if (!rc && h->has_j3) {
    hipLaunchKernelGGL((k_j3_value<>), dim3((unsigned)n), dim3(64), lds_j3(h), h->stream, h->S, h->js, (double*)h->b_j3u.p + w0);
    rc = check_launch(h, "k_j3_value");
  }
  h->js = js0; h->st = st0; h->W = W0;
  return rc;
}

extern "C" int pqa_branch_exchange(pqa_handle_t* h, const int32_t* keep_src, int64_t nkeep, const double* recv_x, int64_t nrecv) {
  h->dmc_old_valid = false;  // (the state pqa_dmc_continue refers to is gone)
  TRY(sync_aos(h));
  HIPCHK(hipSetDevice(h->device));
  if (h->W == 0) FAIL("state not initialised (call recompute)");
  if (nkeep < 0 || nrecv < 0 || nkeep + nrecv != h->W) FAIL("pqa_branch_exchange: kept + received walkers must equal the resident count");
  std::vector<int32_t> idx((size_t)h->W, 0);
  for (int64_t k = 0; k < nkeep; ++k) idx[k] = keep_src[k];
  TRY(pqa_resample(h, idx.data()));  // received slots gather walker 0's state: overwritten below
  if (nrecv > 0) {
    TRY(copy_in(h, h->js.x + (size_t)nkeep * h->N * 3, recv_x, (size_t)nrecv * h->N * 3 * sizeof(double)));
    TRY(recompute_range(h, nkeep, nrecv));
  }
  HIPCHK(hipStreamSynchronize(h->stream));
  return 0;
}

// device-wide exclusive scan c[n] -> o[n+1]; marks[k] = o[k*Wm] for k = 0..n/Wm (pqa_dmc.hpp)
// ---------------------------------------------------------------- density-matrix sampling (pqa_dm.hpp)
extern "C" int pqa_dm_walk(pqa_handle_t* h, int slot, int spin, int64_t n, int nsamples, double tstep, double* pos, const double* gauss,
                           const double* unif, uint64_t seed, int nkeep, double* keep_pos, double* accept) {
  HIPCHK(hipSetDevice(h->device));
  if (!h->has_slater) FAIL("handle has no orbital tables");
  if (slot < 0 || slot > 1 || spin < 0 || spin > 1) FAIL("pqa_dm_walk: slot and spin must be 0 or 1");
  if (n <= 0 || nsamples < 0 || nkeep < 0 || nkeep > nsamples) FAIL("pqa_dm_walk: bad sizes");
  if ((gauss == nullptr) != (unif == nullptr)) FAIL("pqa_dm_walk: give both tapes or neither");
  if (h->nmo[spin] == 0) FAIL("pqa_dm_walk: no orbitals for this spin");
  auto& d = h->dm[slot];
  const int nmo2 = h->nmo[spin];  // complex handles count [Re | Im] columns
  d.n = n; d.nkeep = nkeep; d.spin = spin;
  h->saved_valid = false;
  TRY(ensure(h, d.pos, (size_t)n * 3 * sizeof(double)));
  TRY(ensure(h, d.newpos, (size_t)n * 3 * sizeof(double)));
  TRY(ensure(h, d.row, (size_t)n * nmo2 * sizeof(double)));
  TRY(ensure(h, d.f, (size_t)n * sizeof(double)));
  TRY(ensure(h, d.keep_pos, (size_t)std::max(nkeep, 1) * n * 3 * sizeof(double)));
  TRY(ensure(h, d.keep_row, (size_t)std::max(nkeep, 1) * n * nmo2 * sizeof(double)));
  TRY(ensure(h, d.keep_f, (size_t)std::max(nkeep, 1) * n * sizeof(double)));
  TRY(ensure(h, h->b_motmp, (size_t)n * nmo2 * sizeof(double)));
  const int CH = 64;  // samples per tape upload
  if (gauss) {
    TRY(ensure(h, h->b_gauss, (size_t)CH * n * 3 * sizeof(double)));
    TRY(ensure(h, h->b_unif, (size_t)CH * n * sizeof(double)));
  }
  if (accept) TRY(ensure(h, h->dm_acc, (size_t)CH * n * sizeof(double)));
  TRY(copy_in(h, d.pos.p, pos, (size_t)n * 3 * sizeof(double)));
  const dim3 g256((unsigned)((n + 255) / 256));
  TRY(launch_orb(h, spin, plain_points((const double*)d.pos.p, n), n, 1, (double*)d.row.p));
  hipLaunchKernelGGL((k_dm_density<>), g256, dim3(256), 0, h->stream, (const double*)d.row.p, (long)n, nmo2, (double*)d.f.p);
  TRY(check_launch(h, "k_dm_density"));
  const double sq = sqrt(tstep);
  for (int s0 = 0; s0 < nsamples; s0 += CH) {
    const int ns = std::min(CH, nsamples - s0);
    if (gauss) {
      TRY(copy_in(h, h->b_gauss.p, gauss + (size_t)s0 * n * 3, (size_t)ns * n * 3 * sizeof(double)));
      TRY(copy_in(h, h->b_unif.p, unif + (size_t)s0 * n, (size_t)ns * n * sizeof(double)));
    }
    for (int k = 0; k < ns; ++k) {
      const int s = s0 + k, kk = s - (nsamples - nkeep);
      hipLaunchKernelGGL((k_dm_propose<>), g256, dim3(256), 0, h->stream, (const double*)d.pos.p,
                         gauss ? (const double*)h->b_gauss.p + (size_t)k * n * 3 : (const double*)nullptr, seed, (uint32_t)s, sq, (long)n,
                         (double*)d.newpos.p);
      TRY(launch_orb(h, spin, plain_points((const double*)d.newpos.p, n), n, 1, (double*)h->b_motmp.p));
      hipLaunchKernelGGL((k_dm_accept<>), dim3((unsigned)n), dim3(64), 0, h->stream, (double*)d.pos.p, (double*)d.row.p, (double*)d.f.p,
                         (const double*)d.newpos.p, (const double*)h->b_motmp.p,
                         unif ? (const double*)h->b_unif.p + (size_t)k * n : (const double*)nullptr, seed, (uint32_t)s, (long)n, nmo2,
                         accept ? (double*)h->dm_acc.p + (size_t)k * n : (double*)nullptr,
                         kk >= 0 ? (double*)d.keep_pos.p + (size_t)kk * n * 3 : (double*)nullptr,
                         kk >= 0 ? (double*)d.keep_row.p + (size_t)kk * n * nmo2 : (double*)nullptr,
                         kk >= 0 ? (double*)d.keep_f.p + (size_t)kk * n : (double*)nullptr);
    }
    TRY(check_launch(h, "k_dm_propose/k_dm_accept"));
    if (accept) TRY(copy_out(h, accept + (size_t)s0 * n, h->dm_acc.p, (size_t)ns * n * sizeof(double)));
  }
  TRY(copy_out(h, pos, d.pos.p, (size_t)n * 3 * sizeof(double)));
  if (keep_pos && nkeep > 0) TRY(copy_out(h, keep_pos, d.keep_pos.p, (size_t)nkeep * n * 3 * sizeof(double)));
  return 0;
}

extern "C" int pqa_dm_points(pqa_handle_t* h, int slot, int spin, const double* pts, int64_t npts) {
  HIPCHK(hipSetDevice(h->device));
  if (!h->has_slater) FAIL("handle has no orbital tables");
  if (slot < 0 || slot > 1 || spin < 0 || spin > 1 || npts <= 0) FAIL("pqa_dm_points: bad arguments");
  auto& d = h->dm[slot];
  const int nmo2 = h->nmo[spin];
  h->saved_valid = false;
  TRY(ensure(h, h->b_pts, (size_t)npts * 3 * sizeof(double)));
  TRY(ensure(h, d.cfg, (size_t)npts * nmo2 * sizeof(double)));
  TRY(copy_in(h, h->b_pts.p, pts, (size_t)npts * 3 * sizeof(double)));
  d.ncfg = npts;
  return launch_orb(h, spin, plain_points((const double*)h->b_pts.p, npts), npts, 1, (double*)d.cfg.p);
}

static int dm_prepare(pqa_handle* h, long nconf, long nval, int cx, int first, long nnorm_a, long nnorm_b) {
  if (first) { h->dm_nconf = nconf; h->dm_nval = nval; h->dm_cx = cx; }
  else if (h->dm_nconf != nconf || h->dm_nval != nval || h->dm_cx != cx) FAIL("density-matrix accumulation: shape changed since the first sweep");
  TRY(ensure(h, h->dm_val, (size_t)nconf * nval * (cx ? 2 : 1) * sizeof(double)));
  TRY(ensure(h, h->dm_norm[0], (size_t)nconf * std::max(nnorm_a, 1L) * sizeof(double)));
  TRY(ensure(h, h->dm_norm[1], (size_t)nconf * std::max(nnorm_b, 1L) * sizeof(double)));
  return 0;
}

extern "C" int pqa_obdm_accumulate(pqa_handle_t* h, int slot, int k, int64_t nconf, int nelec, const int32_t* assign, const double* ratio,
                                   int ratio_complex, int first) {
  HIPCHK(hipSetDevice(h->device));
  if (slot < 0 || slot > 1) FAIL("pqa_obdm_accumulate: slot must be 0 or 1");
  auto& d = h->dm[slot];
  if (k < 0 || k >= d.nkeep) FAIL("pqa_obdm_accumulate: sample was not kept by pqa_dm_walk");
  if (d.ncfg != nconf * nelec) FAIL("pqa_obdm_accumulate: pqa_dm_points was called with another number of points");
  const int oc = h->cplx ? 1 : 0, rc = ratio_complex ? 1 : 0, nmo2 = h->nmo[d.spin], norb = nmo2 / (oc ? 2 : 1);
  TRY(dm_prepare(h, nconf, (long)norb * norb, rc | oc, first, norb, 0));
  TRY(ensure(h, h->dm_assign[0], (size_t)nconf * sizeof(int)));
  TRY(ensure(h, h->dm_ratio, (size_t)nconf * nelec * (rc ? 2 : 1) * sizeof(double)));
  TRY(copy_in(h, h->dm_assign[0].p, assign, (size_t)nconf * sizeof(int)));
  TRY(copy_in(h, h->dm_ratio.p, ratio, (size_t)nconf * nelec * (rc ? 2 : 1) * sizeof(double)));
  hipLaunchKernelGGL((k_obdm_acc<>), dim3((unsigned)nconf), dim3(256), (size_t)2 * norb * sizeof(double), h->stream,
                     (const double*)d.keep_row.p + (size_t)k * d.n * nmo2, (const double*)d.keep_f.p + (size_t)k * d.n,
                     (const int*)h->dm_assign[0].p, (const double*)d.cfg.p, (const double*)h->dm_ratio.p, rc, oc, nelec, norb, first,
                     (double*)h->dm_val.p, (double*)h->dm_norm[0].p);
  return check_launch(h, "k_obdm_acc");
}

extern "C" int pqa_tbdm_accumulate(pqa_handle_t* h, int k, int64_t nconf, int nea, int neb, const int32_t* assign_a, const int32_t* assign_b,
                                   const double* ratio, int ratio_complex, const int32_t* ijkl, int ntuple, int first) {
  HIPCHK(hipSetDevice(h->device));
  auto& da = h->dm[0];
  auto& db = h->dm[1];
  if (k < 0 || k >= da.nkeep || k >= db.nkeep) FAIL("pqa_tbdm_accumulate: sample was not kept by pqa_dm_walk");
  if (da.ncfg != nconf * nea || db.ncfg != nconf * neb) FAIL("pqa_tbdm_accumulate: pqa_dm_points was called with other numbers of points");
  const int oc = h->cplx ? 1 : 0, rc = ratio_complex ? 1 : 0;
  const int na2 = h->nmo[da.spin], nb2 = h->nmo[db.spin], na = na2 / (oc ? 2 : 1), nb = nb2 / (oc ? 2 : 1);
  const size_t lds = (size_t)2 * ((size_t)nea * nb + (size_t)na * nb) * sizeof(double);
  if (lds > 64 * 1024) FAIL("pqa_tbdm_accumulate: orbital basis too large for the per-walker LDS tiles");
  TRY(dm_prepare(h, nconf, ntuple, rc | oc, first, na, nb));
  TRY(ensure(h, h->dm_assign[0], (size_t)nconf * sizeof(int)));
  TRY(ensure(h, h->dm_assign[1], (size_t)nconf * sizeof(int)));
  TRY(ensure(h, h->dm_ratio, (size_t)nconf * nea * neb * (rc ? 2 : 1) * sizeof(double)));
  TRY(ensure(h, h->dm_ijkl, (size_t)4 * ntuple * sizeof(int)));
  TRY(copy_in(h, h->dm_assign[0].p, assign_a, (size_t)nconf * sizeof(int)));
  TRY(copy_in(h, h->dm_assign[1].p, assign_b, (size_t)nconf * sizeof(int)));
  TRY(copy_in(h, h->dm_ratio.p, ratio, (size_t)nconf * nea * neb * (rc ? 2 : 1) * sizeof(double)));
  TRY(copy_in(h, h->dm_ijkl.p, ijkl, (size_t)4 * ntuple * sizeof(int)));
  hipLaunchKernelGGL((k_tbdm_acc<>), dim3((unsigned)nconf), dim3(256), lds, h->stream,
                     (const double*)da.keep_row.p + (size_t)k * da.n * na2, (const double*)da.keep_f.p + (size_t)k * da.n,
                     (const double*)db.keep_row.p + (size_t)k * db.n * nb2, (const double*)db.keep_f.p + (size_t)k * db.n,
                     (const int*)h->dm_assign[0].p, (const int*)h->dm_assign[1].p, (const double*)da.cfg.p, (const double*)db.cfg.p,
                     (const double*)h->dm_ratio.p, rc, oc, nea, neb, na, nb, (const int*)h->dm_ijkl.p, ntuple, first,
                     (double*)h->dm_val.p, (double*)h->dm_norm[0].p, (double*)h->dm_norm[1].p);
  return check_launch(h, "k_tbdm_acc");
}

// which: 0 value (dm_nval entries per configuration, interleaved complex if any input was), 1 norm (first / a), 2 norm b of
// `ncol` entries; mean != 0: average over the configurations on the device
extern "C" int pqa_dm_fetch(pqa_handle_t* h, int which, int ncol, double scale, int mean, double* out) {
  HIPCHK(hipSetDevice(h->device));
  if (h->dm_nconf <= 0) FAIL("pqa_dm_fetch: nothing accumulated");
  const double* src;
  long cols;
  if (which == 0) { src = (const double*)h->dm_val.p; cols = h->dm_nval * (h->dm_cx ? 2 : 1); }
  else if (which == 1 || which == 2) { src = (const double*)h->dm_norm[which - 1].p; cols = ncol; }
  else FAIL("pqa_dm_fetch: which must be 0, 1 or 2");
  if (which == 0 && ncol != cols) FAIL("pqa_dm_fetch: ncol does not match the accumulated value");
  const long nout = mean ? cols : h->dm_nconf * cols;
  TRY(ensure(h, h->dm_tmp, (size_t)nout * sizeof(double)));
  if (mean) hipLaunchKernelGGL((k_col_means<>), dim3((unsigned)cols), dim3(256), 0, h->stream, src, h->dm_nconf, cols, scale, (double*)h->dm_tmp.p);
  else hipLaunchKernelGGL((k_scale_copy<>), dim3((unsigned)((nout + 255) / 256)), dim3(256), 0, h->stream, src, nout, scale, (double*)h->dm_tmp.p);
  TRY(check_launch(h, "pqa_dm_fetch"));
  return copy_out(h, out, h->dm_tmp.p, (size_t)nout * sizeof(double));
}

extern "C" int pqa_gram(pqa_handle_t* h, int64_t n, int P, int Q, const double* A, const double* B, double* C) {
  HIPCHK(hipSetDevice(h->device));
  if (n <= 0 || P <= 0 || Q <= 0) FAIL("pqa_gram: bad sizes");
  const int tiles = ((P + 15) / 16) * ((Q + 15) / 16);
  int nslice = (int)std::min<long>(std::max<long>(1, 1024 / tiles), std::max<long>(1, n / 64));
  DevBuf &a = h->b_pts, &b = h->b_out, &part = h->dm_tmp, &c = h->dm_acc;
  TRY(ensure(h, a, (size_t)n * P * sizeof(double)));
  TRY(ensure(h, b, (size_t)n * Q * sizeof(double)));
  TRY(ensure(h, part, (size_t)nslice * P * Q * sizeof(double)));
  TRY(ensure(h, c, (size_t)P * Q * sizeof(double)));
  TRY(copy_in(h, a.p, A, (size_t)n * P * sizeof(double)));
  TRY(copy_in(h, b.p, B, (size_t)n * Q * sizeof(double)));
  hipLaunchKernelGGL((k_gram_mfma<>), dim3((unsigned)((P + 15) / 16), (unsigned)((Q + 15) / 16), (unsigned)nslice), dim3(64), 0, h->stream,
                     (const double*)a.p, (const double*)b.p, (long)n, P, Q, nslice, (double*)part.p);
  hipLaunchKernelGGL((k_gram_reduce<>), dim3((unsigned)(((long)P * Q + 255) / 256)), dim3(256), 0, h->stream, (const double*)part.p, (long)P * Q,
                     nslice, (double*)c.p);
  TRY(check_launch(h, "k_gram_mfma"));
  return copy_out(h, C, c.p, (size_t)P * Q * sizeof(double));
}

// The standard normals and Metropolis uniforms the fused sweeps draw for (seed, step): gauss (N,W,3), unif (N,W) for
// walkers 0..W-1 (the streams are keyed by walker index, so any prefix of an ensemble can be asked for).
extern "C" int pqa_philox_tapes(pqa_handle_t* h, uint64_t seed, int step, int64_t W, double* gauss, double* unif) {
  HIPCHK(hipSetDevice(h->device));
  if (W <= 0 || step < 0 || !gauss || !unif) FAIL("pqa_philox_tapes: bad arguments");
  const size_t NW = (size_t)h->N * W;
  TRY(ensure(h, h->b_gauss, NW * 3 * sizeof(double)));
  TRY(ensure(h, h->b_unif, NW * sizeof(double)));
  hipLaunchKernelGGL((k_tile_draws<>), dim3((unsigned)((NW + 255) / 256)), dim3(256), 0, h->stream, seed, (uint32_t)step, h->N, (long)W,
                     (double*)h->b_gauss.p, (double*)h->b_unif.p);
  TRY(check_launch(h, "k_tile_draws"));
  TRY(copy_in(h, gauss, h->b_gauss.p, NW * 3 * sizeof(double)));
  return copy_out(h, unif, h->b_unif.p, NW * sizeof(double));
}

// out[c][w] = the uniform of Philox(seed; walker w, counter c, stream, step) — what ecp_pass / k_tm_count / k_tm_walker draw
static __global__ __launch_bounds__(256) void k_uniform_plane(uint64_t seed, uint32_t stream, uint32_t step, long ncount, long W, double* __restrict__ out) {
  const long idx = (long)blockIdx.x * 256 + threadIdx.x;
  if (idx >= ncount * W) return;
  const long c = idx / W, w = idx - c * W;
  const Philox p = philox(seed, (uint32_t)w, (uint32_t)c, stream, step);
  out[idx] = u01(p.c[0], p.c[1]);
}
extern "C" int pqa_philox_dmc_tapes(pqa_handle_t* h, uint64_t seed, int nsteps, int64_t W, pqa_dmc_tapes_t* out) {
  HIPCHK(hipSetDevice(h->device));
  if (W <= 0 || nsteps <= 0 || !out || !out->gauss || !out->unif) FAIL("pqa_philox_dmc_tapes: bad arguments");
  const int N = h->N, necp = h->necp;
  const size_t NW = (size_t)N * W, nrot = (size_t)N * std::max(necp, 1);
  if (necp > 0 && (!out->ecp_rot || !out->ecp_unif)) FAIL("pqa_philox_dmc_tapes: ECP systems need ecp_rot and ecp_unif");
  const bool tm = necp > 0 && out->tm_rot && out->tm_unif && out->tm_u1 && out->tm_u2;
  TRY(ensure(h, h->b_gauss, NW * 3 * sizeof(double)));
  TRY(ensure(h, h->b_unif, std::max(NW, nrot * (size_t)W) * sizeof(double)));
  TRY(ensure(h, h->b_rot, nrot * 9 * sizeof(double)));
  auto plane = [&](uint32_t stream, uint32_t step, size_t ncount, const double* dst) -> int {
    hipLaunchKernelGGL(k_uniform_plane, dim3((unsigned)((ncount * W + 255) / 256)), dim3(256), 0, h->stream, seed, stream, step, (long)ncount, (long)W,
                       (double*)h->b_unif.p);
    TRY(check_launch(h, "k_uniform_plane"));
    return copy_out(h, const_cast<double*>(dst), h->b_unif.p, ncount * W * sizeof(double));
  };
  auto rots = [&](uint64_t sd, uint32_t step, const double* dst) -> int {
    hipLaunchKernelGGL((k_gen_rot<>), dim3((unsigned)((nrot + 63) / 64)), dim3(64), 0, h->stream, (int)nrot, sd, step, (double*)h->b_rot.p);
    TRY(check_launch(h, "k_gen_rot"));
    return copy_out(h, const_cast<double*>(dst), h->b_rot.p, nrot * 9 * sizeof(double));
  };
  for (int i = 0; i <= nsteps; ++i) {
    if (necp > 0) {  // energy evaluation i (0: the starting configuration; energy_dev's draws)
      TRY(rots(seed, (uint32_t)i, out->ecp_rot + (size_t)i * nrot * 9));
      TRY(plane(PQA_STREAM_ECPMASK, (uint32_t)i, nrot, out->ecp_unif + (size_t)i * nrot * W));
    }
    if (i == nsteps) break;
    if (tm) {
      TRY(rots(seed ^ 0x9E3779B97F4A7C15ull, (uint32_t)i, out->tm_rot + (size_t)i * nrot * 9));
      TRY(plane(PQA_STREAM_TMMASK, (uint32_t)i, nrot, out->tm_unif + (size_t)i * nrot * W));
      TRY(plane(PQA_STREAM_TM_U1, (uint32_t)i, (size_t)N, out->tm_u1 + (size_t)i * NW));
      TRY(plane(PQA_STREAM_TM_U2, (uint32_t)i, (size_t)N, out->tm_u2 + (size_t)i * NW));
    }
    hipLaunchKernelGGL((k_tile_draws<>), dim3((unsigned)((NW + 255) / 256)), dim3(256), 0, h->stream, seed, (uint32_t)i, N, (long)W,
                       (double*)h->b_gauss.p, (double*)h->b_unif.p);
    TRY(check_launch(h, "k_tile_draws"));
    TRY(copy_in(h, const_cast<double*>(out->gauss) + (size_t)i * NW * 3, h->b_gauss.p, NW * 3 * sizeof(double)));
    TRY(copy_out(h, const_cast<double*>(out->unif) + (size_t)i * NW, h->b_unif.p, NW * sizeof(double)));
  }
  return 0;
}

extern "C" int pqa_sync(pqa_handle_t* h) {
  HIPCHK(hipSetDevice(h->device));
  HIPCHK(hipStreamSynchronize(h->stream));
  return 0;
}
extern "C" int pqa_timer_start(pqa_handle_t* h) {
  HIPCHK(hipSetDevice(h->device));
  HIPCHK(hipEventRecord(h->ev0, h->stream));
  return 0;
}
extern "C" int pqa_timer_stop(pqa_handle_t* h, double* elapsed_ms) {
  HIPCHK(hipSetDevice(h->device));
  HIPCHK(hipEventRecord(h->ev1, h->stream));
  HIPCHK(hipEventSynchronize(h->ev1));
  float ms = 0.f;
  HIPCHK(hipEventElapsedTime(&ms, h->ev0, h->ev1));
  *elapsed_ms = ms;
  return 0;
}
extern "C" int pqa_profile_enable(pqa_handle_t* h, int enable) {
  HIPCHK(hipSetDevice(h->device));
  HIPCHK(hipStreamSynchronize(h->stream));
  h->profile = enable != 0;
  h->prof_used = 0; h->prof_launches = 0; h->prof_ms = 0.0; h->prof_pc = 0.0;
  h->prof2_used = 0; h->prof2_launches = 0; h->prof2_ms = 0.0;
  h->prof3_used = 0; h->prof3_launches = 0; h->prof3_ms = 0.0;
  return 0;
}
extern "C" int pqa_profile_query(pqa_handle_t* h, int64_t* launches, double* total_ms, double* point_comps) {
  HIPCHK(hipSetDevice(h->device));
  HIPCHK(hipStreamSynchronize(h->stream));
  for (size_t i = 0; i < h->prof_used; ++i) {
    float ms = 0.f;
    HIPCHK(hipEventElapsedTime(&ms, h->prof_events[i].first, h->prof_events[i].second));
    h->prof_ms += ms;
  }
  h->prof_used = 0;
  if (launches) *launches = h->prof_launches;
  if (total_ms) *total_ms = h->prof_ms;
  if (point_comps) *point_comps = h->prof_pc;
  return 0;
}
extern "C" int pqa_profile_query_commit(pqa_handle_t* h, int64_t* launches, double* total_ms) {
  HIPCHK(hipSetDevice(h->device));
  HIPCHK(hipStreamSynchronize(h->stream));
  for (size_t i = 0; i < h->prof2_used; ++i) {
    float ms = 0.f;
    HIPCHK(hipEventElapsedTime(&ms, h->prof2_events[i].first, h->prof2_events[i].second));
    h->prof2_ms += ms;
  }
  h->prof2_used = 0;
  if (launches) *launches = h->prof2_launches;
  if (total_ms) *total_ms = h->prof2_ms;
  return 0;
}
extern "C" int pqa_profile_query_part(pqa_handle_t* h, int64_t* launches, double* total_ms, int* groups) {
  HIPCHK(hipSetDevice(h->device));
  HIPCHK(hipStreamSynchronize(h->stream));
  for (size_t i = 0; i < h->prof3_used; ++i) {
    float ms = 0.f;
    HIPCHK(hipEventElapsedTime(&ms, h->prof3_events[i].first, h->prof3_events[i].second));
    h->prof3_ms += ms;
  }
  h->prof3_used = 0;
  if (launches) *launches = h->prof3_launches;
  if (total_ms) *total_ms = h->prof3_ms;
  if (groups) {
    LwCtx lc;
    TRY(lw_setup(h, false, lc));
    *groups = lc.Gm;
  }
  return 0;
}
extern "C" int pqa_set_ecp_naip(pqa_handle_t* h, int32_t naip) {
  HIPCHK(hipSetDevice(h->device));
  if (naip != 0 && ecp_quadrature_offset(naip) < 0) FAIL("naip must be one of 6, 12, 18, 26, 32, 50 (eval_ecp.py:266-267), or 0 for the per-atom default");
  if (h->necp == 0) { h->ecp_naip = naip; return 0; }
  std::vector<int> na((size_t)h->necp), qo((size_t)h->necp);
  for (int k = 0; k < h->necp; ++k) {
    na[k] = naip ? naip : (h->ecp_nch[k] <= 2 ? 6 : 12);
    qo[k] = ecp_quadrature_offset(na[k]);
  }
  HIPCHK(hipStreamSynchronize(h->stream));
  h->ecp_hint_valid = false; h->last_ecp_dev[0] = h->last_ecp_dev[1] = nullptr;  // (another rule: other point totals)
  if (h->last_ecp_points < 0) h->last_ecp_points = 0;
  h->S.ecp_naip_max = 0;
  for (int k = 0; k < h->necp; ++k) h->S.ecp_naip_max = std::max(h->S.ecp_naip_max, na[k]);
  HIPCHK(hipMemcpy(h->d_ecp_naip, na.data(), na.size() * sizeof(int), hipMemcpyHostToDevice));
  HIPCHK(hipMemcpy(h->d_ecp_qoff, qo.data(), qo.size() * sizeof(int), hipMemcpyHostToDevice));
  h->ecp_naip = naip;
  return 0;
}
extern "C" int pqa_last_ecp_points(pqa_handle_t* h, int64_t* npoints) {
  if (h->last_ecp_points < 0 && h->last_ecp_dev[0]) {  // the last evaluation left its totals on the device (pqa_energy.hip)
    HIPCHK(hipSetDevice(h->device));
    long t[2];
    TRY(copy_in(h, &t[0], h->last_ecp_dev[0], sizeof(long)));
    TRY(copy_out(h, &t[1], h->last_ecp_dev[1], sizeof(long)));
    h->last_ecp_points = t[0] + t[1];
  }
  *npoints = h->last_ecp_points;
  return 0;
}
