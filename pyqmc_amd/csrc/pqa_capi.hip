// pyqmc_amd C ABI implementation (host side).  See include/pyqmc_amd.h for the contract.
// Single translation unit: the device code lives in the headers included below.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "../../include/pyqmc_amd.h"
#include "pqa_ao.hpp"
#include "pqa_common.hpp"
#include "pqa_cslater.hpp"
#include "pqa_dmc.hpp"
#include "pqa_energy.hpp"
#include "pqa_ecp.hpp"
#include "pqa_jastrow.hpp"
#include "pqa_lw.hpp"
#include "pqa_slater.hpp"
#include "pqa_tile.hpp"
#include "pqa_dm.hpp"
#include "pqa_vmc.hpp"

static thread_local std::string g_create_error;

struct DevBuf {
  void* p = nullptr;
  size_t cap = 0;
};

struct ChunkHost {
  std::vector<int> nk, row0;
  std::vector<int> shell_kb, shell_chunk;  // per shell: first tile row inside its chunk, chunk index
  std::vector<int> cw_off[3], cw_shell[3];  // shell lists per (chunk, lane group) for 4, 8 and 16 groups
  int rows_pad = 0;
};

struct pqa_handle {
  int device = 0;
  hipStream_t stream = nullptr;
  std::string err;
  std::vector<void*> owned;  // table allocations freed at destroy
  // host copies needed after create
  int natom = 0, nup = 0, ndn = 0, N = 0, nao = 0, nshell = 0;
  int nmo[2] = {0, 0}, nt[2] = {1, 1}, ndet = 1, ndet_s[2] = {1, 1};
  int na = 0, nb = 0, necp = 0;
  bool tm_pre = true;   // T-move ratios of all candidates in one thread-per-candidate launch (PQA_TM_PRE=0: wave-per-walker loop only)
  bool aos_stale = false;  // the lane-per-walker planes hold the live state; the walker-major arrays are converted back on demand (sync_aos)
  int wide_nth = 1024;  // threads per block of k_orb_wide (PQA_WIDE_NTH; periodic default 512)
  int pbc_nw = 2;  // words per (atom, point) of the sorted image lists k_pbc_prepass writes (4 entries each)
  bool twist = false;  // twisted boundary conditions: complex lattice-summed AOs, unfolded positions (include/pyqmc_amd.h)
  bool cplx = false;  // complex orbitals: mo_* hold [Re C | Im C], see pqa_cslater.hpp
  bool has_slater = false, has_jastrow = false;  // has_jastrow: any Jastrow factor (two- and/or three-body)
  bool has_j2 = false, has_j3 = false;
  int na3 = 0, nb3 = 0;
  double* d_c3 = nullptr;
  DevBuf b_j3u;
  double ii_energy = 0.0;
  EwaldDev ew{};  // periodic Coulomb tables (pqa_set_ewald)
  bool ew_set = false;
  std::vector<int> shell_l, shell_np, shell_ao;
  std::vector<int> shell_cost;  // phase-1 cost model of a shell (shell_costs): balances the lane groups of the orbital kernels
  SysDev S{};
  ChunkHost chunks[2];  // [0]: KC=16 (5 components), [1]: KC=32 (value only)
  ChunkTab tab[2]{};
  const unsigned char* out_sel = nullptr;  // two-slot output of the NEXT orbital launch (ChunkTab::out_sel; set by launch_orb)
  long out_slot_stride = 0;
  double* d_mo[2] = {nullptr, nullptr};       // [nao][nmo]
  double* d_cpad[2][2] = {{nullptr, nullptr}, {nullptr, nullptr}};  // [tab][spin]
  double *d_acoeff = nullptr, *d_bcoeff = nullptr, *d_detcoeff = nullptr, *d_quad = nullptr;
  // walker state
  long W = 0;
  SlaterState st{};
  JastrowState js{};
  DevBuf b_x, b_T[2], b_dsign[2], b_dlog[2], b_cache[2], b_aval, b_bval;
  DevBuf b_alt_x, b_alt_T[2], b_alt_dsign[2], b_alt_dlog[2], b_alt_cache[2], b_alt_aval, b_alt_bval, b_alt_j3u, b_rsidx;  // pqa_resample's other halves
  // scratch
  DevBuf b_pts, b_motmp, b_out, b_widx, b_mask, b_ao, b_flag, b_newpos, b_aux, b_accept, b_accrec, b_acccnt, b_accw, b_dwrap, b_wrap, b_epass, b_eptw[2], b_econ[2], b_eu0[2], b_tves, b_pgdet, b_pbcd0, b_pbcmask, b_pbcth, b_tmuold;
  int* d_colmap[2] = {nullptr, nullptr};  // [ndet_s][nmo_s] column of an orbital in a unique determinant, or -1
  int ecp_wave = 0;  // PQA_ECP_WAVE=1: wave-per-walker ECP accumulation (A/B)
  int ecp_soa_t = 1;  // PQA_ECP_SOA_T=0: transpose the inverse back for the ECP point kernel (A/B)
  int ecp_point_lw = 1;  // PQA_ECP_POINT_LW=0: k_ecp_point on the planes instead of k_ecp_point_lw (A/B)
  long flush_wb8_max = 8192;  // PQA_FLUSH_WB8_MAX: walker counts up to which k_flush_lw runs with 8 walkers per block
  long draws_max = 16384;  // PQA_DRAWS_MAX: walker counts up to which a fused sweep draws its random numbers ahead (k_tile_draws)
  int step_pre = 1;      // PQA_STEP_PRE=0: k_step_lw for small shards too (A/B, bitwise check)
  int ecp_acc_waves = 0; // PQA_ECP_ACC_WAVES: 1 / 4 waves per walker in k_ecp_accum / k_kinetic_coulomb (0: 4 while walkers x electrons <= 32768)
  int jas_fold_allowed = 1;  // PQA_JAS_FOLD=0: Voronoi reduction in every periodic Jastrow pair (A/B, bitwise check)
  int ecp_atom_major = 1;  // PQA_ECP_ATOM_MAJOR=0: walker-major ECP point lists in periodic cells too (A/B)
  int ecp_lds = 1;       // PQA_ECP_LDS=0: first-generation k_ecp_count / k_ecp_fill (A/B)
  int ecp_nchan = 0, ecp_nterm = 0;
  long wrap_W = 0;
  DevBuf b_gauss, b_unif, b_kc, b_en, b_means, b_sign, b_log, b_ju;
  DevBuf b_tpos, b_twgt, b_tlive, b_trat;
  DevBuf b_tmcnt, b_tmoff, b_tmpass, b_tmamp, b_tmacc, b_tmidx, b_tmapos, b_tmu, b_tmtile, b_tmaoff, b_tmptw, b_tmmarks, b_dmcw, b_dmcold, b_dmcr2, b_dmcout;
  int tm_P = 0;
  int *d_ptk = nullptr, *d_pti = nullptr;
  DevBuf b_xt, b_Tt[2], b_rc[2], b_sel[2], b_auxt, b_kpart, b_rbuf, b_vbuf, b_act;
  // electrons per Sherman-Morrison block (PQA_LW_KB): -1 automatic (4 for >= 16 electrons per spin), 0 = update every row on
  // every move.  Blocking is bitwise identical and cuts the inverse's HBM traffic ~3x; it pays since k_flush_lw stages the
  // block's update vectors in LDS (1.26 -> 0.27 ms per flush at 65536 walkers): commit + flush 15.5 -> 8.4 ms per step.
  int lw_kb = -1;
  int lw_nw = 0;  // PQA_LW_NW: walkers per block of k_step_lw (16, 32, 64; 0 = automatic)
  int lw_gm = 0;  // thread groups of the move kernels (PQA_LW_GM; 0 = automatic)  // lane-per-walker SoA mirrors (pqa_lw.hpp)
  DevBuf b_rot, b_eunif, b_elocal, b_ecnt, b_eoff, b_epts[2], b_ewgt[2], b_epte[2], b_emo[2], b_ecp;
  int orb_tp = 0;  // 0 = automatic
  int orb_nosplit = 0;  // PQA_ORB_NOSPLIT=1: never split the chunk loop of small periodic launches (A/B)
  long orb_split_max = 8192;  // largest periodic launch whose chunk loop is split over two blocks (PQA_ORB_SPLIT_MAX)
  // AO rows per chunk of the PERIODIC 5-component launch: 32 halves the number of (phase 1, barrier, MFMA, barrier)
  // rounds of a block's latency chain — 2x2x2 diamond supercell +4.5-10 % at every walker count, 8-atom cell +11 % at 8192
  // walkers, -4 % at 32768 (PQA_ORB_KC5=16 restores the 16-row chunks; the open-system kernel keeps 16: 0.36 vs 0.29 of peak)
  int orb_kc5 = 32;
  struct TpTune { float ms[2] = {1e30f, 1e30f}; int n[2] = {0, 0}; int choice = 0; };  // periodic k_orb: [0] 32-point, [1] 64-point tiles
  TpTune tp_tune[2][48];  // per chunk table (5 / 1 components) and log2 bucket of the point count
  WideTab wide[2]{};  // lane-group shell lists of the whole-K small-launch kernel (k_orb_wide), per chunk table (64 groups; periodic: 32)
  int orb_wide = -1;  // PQA_ORB_WIDE: -1 automatic (5-component launches of <= orb_wide_max points), 0 never, 1 whenever the tile fits LDS
  long orb_wide_max = 8192;  // PQA_ORB_WIDE_MAX
  std::vector<const void*> wide_attr;  // kernels whose dynamic-LDS limit has been raised
  int orb_ws = -1;  // -1 automatic; 1 wave-specialised orbital kernel; 0 phase-alternating k_orb (PQA_ORB_WS)
  int orb_notab = 0;  // PQA_ORB_NOTAB=1: basis tables from global memory (A/B)
  int lw_mode = 1;  // 1: lane-per-walker fused sweep (single determinant); 0: wave-per-walker kernels; 2: walker-tile sweep (PQA_LW)
  // density-matrix sampling (pqa_dm.hpp): per slot the auxiliary walkers (position, orbital row, density), the kept samples
  // and the orbitals at the configurations' electrons; accumulators of the estimator in dm_val / dm_norm
  struct DmSlot { DevBuf pos, row, f, newpos, keep_pos, keep_row, keep_f, cfg; long n = 0, ncfg = 0; int nkeep = 0, spin = 0; };
  DmSlot dm[2];
  DevBuf dm_val, dm_norm[2], dm_tmp, dm_ijkl, dm_assign[2], dm_ratio, dm_acc;
  long dm_nconf = 0, dm_nval = 0;
  int dm_cx = 0;
  bool tile_attr_set = false;
  bool saved_valid = false;
  bool jas_stale = false;  // fused sweeps move x without patching avalues/bvalues
  int saved_e = -1;
  long last_ecp_points = 0;
  // measurement
  hipEvent_t ev0 = nullptr, ev1 = nullptr;
  bool profile = false;
  std::vector<std::pair<hipEvent_t, hipEvent_t>> prof_events;
  std::vector<std::pair<hipEvent_t, hipEvent_t>> prof2_events;  // Sherman-Morrison commit launches of the fused sweep
  size_t prof2_used = 0;
  std::vector<std::pair<hipEvent_t, hipEvent_t>> prof3_events;  // partial-sum launches (k_move_part_lw) of the fused sweep
  size_t prof3_used = 0;
  long prof3_launches = 0;
  double prof3_ms = 0.0;
  long prof2_launches = 0;
  double prof2_ms = 0.0;
  size_t prof_used = 0;
  unsigned prof_tick = 0, prof2_tick = 0;  // the event pairs bracket every 4th eligible launch (PQA_PROF_STRIDE)
  unsigned prof_stride = 4;
  long prof_launches = 0;
  double prof_ms = 0.0, prof_pc = 0.0;
};

#define HIPCHK(call)                                                                                     \
  do {                                                                                                   \
    hipError_t e_ = (call);                                                                              \
    if (e_ != hipSuccess) {                                                                              \
      char buf_[512];                                                                                    \
      snprintf(buf_, sizeof buf_, "%s failed: %s (%s:%d)", #call, hipGetErrorString(e_), __FILE__, __LINE__); \
      h->err = buf_;                                                                                     \
      return -1;                                                                                         \
    }                                                                                                    \
  } while (0)
#define FAIL(msg)      \
  do {                 \
    h->err = (msg);    \
    return -2;         \
  } while (0)
#define TRY(x)          \
  do {                  \
    int rc_ = (x);      \
    if (rc_) return rc_; \
  } while (0)

struct pqa_handle;
static int sync_aos(pqa_handle* h);
static int scan_ints(pqa_handle* h, const int* c, long* o, long n, long Wm, long* marks);

static int ensure(pqa_handle* h, DevBuf& b, size_t bytes) {
  if (bytes <= b.cap && b.p) return 0;
  // A buffer that has to GROW holds data-dependent sizes (ECP / T-move point lists: ~38 points per walker +- sqrt(N)
  // from step to step).  Exact-size regrowth made every new maximum a hipFree + hipMalloc pair, i.e. a device
  // synchronisation and milliseconds of driver time in the first dozens of steps (the first timed steps on a fresh box
  // ran 15 % slow); 25 % headroom on regrowth ends that after the second step.  First allocations stay exact.
  const bool regrow = b.p != nullptr;
  if (b.p) HIPCHK(hipFree(b.p));
  b.p = nullptr;
  b.cap = 0;
  size_t want = std::max<size_t>(regrow ? bytes + bytes / 4 : bytes, 256);
  HIPCHK(hipMalloc(&b.p, want));
  b.cap = want;
  return 0;
}

template <class T>
static int upload_table(pqa_handle* h, const T* src, size_t n, T** dst) {
  *dst = nullptr;
  if (n == 0) n = 1;
  void* p = nullptr;
  HIPCHK(hipMalloc(&p, n * sizeof(T)));
  h->owned.push_back(p);
  if (src) HIPCHK(hipMemcpy(p, src, n * sizeof(T), hipMemcpyHostToDevice));
  else HIPCHK(hipMemset(p, 0, n * sizeof(T)));
  *dst = (T*)p;
  return 0;
}

static int copy_in(pqa_handle* h, void* dst, const void* src, size_t bytes) {
  if (bytes == 0) return 0;
  HIPCHK(hipMemcpyAsync(dst, src, bytes, hipMemcpyDefault, h->stream));
  return 0;
}
static int copy_out(pqa_handle* h, void* dst, const void* src, size_t bytes) {
  if (bytes) HIPCHK(hipMemcpyAsync(dst, src, bytes, hipMemcpyDefault, h->stream));
  HIPCHK(hipStreamSynchronize(h->stream));
  return 0;
}
static int check_launch(pqa_handle* h, const char* what) {
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) {
    h->err = std::string(what) + " launch failed: " + hipGetErrorString(e);
    return -1;
  }
  return 0;
}

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
  HIPCHK(hipMemcpy(h->d_mo[s], mo_host, (size_t)h->nao * std::max(h->nmo[s], 1) * sizeof(double), hipMemcpyHostToDevice));
  for (int t = 0; t < 2; ++t) TRY(upload_cpad(h, t, s, mo_host));
  return 0;
}

// ---------------------------------------------------------------- create / destroy
extern "C" int pqa_device_count(void) {
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess) return 0;
  return n;
}

extern "C" const char* pqa_last_error(const pqa_handle_t* h) { return h ? h->err.c_str() : g_create_error.c_str(); }

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
  //   PQA_PBC_NW n words (4 image indices each) per (point, atom) image list of the periodic pre-pass (default from the cell;
  //   1 forces the direct-test fallback: tests), PQA_WIDE_NTH 512 k_orb_wide with 512 threads in untwisted periodic cells,
  //   PQA_TM_PRE 0 T-move ratios by the wave-per-walker loop only;
  //   round 3 (each documented at its field above): PQA_STEP_PRE 0 k_step_lw for small shards too, PQA_DRAWS_MAX n walker count up to
  //   which a sweep's random numbers are drawn ahead, PQA_FLUSH_WB8_MAX n 8-walker flush blocks up to n walkers, PQA_ECP_LDS 0 /
  //   PQA_ECP_POINT_LW 0 first-generation ECP list passes / point kernel, PQA_ECP_ACC_WAVES 1|4 waves per walker in the
  //   wave-per-walker energy kernels, PQA_ECP_ATOM_MAJOR 0 walker-major ECP lists in periodic cells, PQA_JAS_FOLD 0 Voronoi
  //   reduction in every periodic Jastrow pair.
  if (const char* tp = getenv("PQA_ORB_TP")) h->orb_tp = atoi(tp);
  if (const char* ns = getenv("PQA_ORB_NOSPLIT")) h->orb_nosplit = atoi(ns);
  if (const char* sm = getenv("PQA_ORB_SPLIT_MAX")) h->orb_split_max = atol(sm);
  if (const char* kc = getenv("PQA_ORB_KC5")) h->orb_kc5 = atoi(kc);
  if (const char* ps = getenv("PQA_PROF_STRIDE")) h->prof_stride = (unsigned)std::max(1, atoi(ps));
  if (const char* lw = getenv("PQA_LW")) h->lw_mode = atoi(lw);
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
  if (const char* dm = getenv("PQA_DRAWS_MAX")) h->draws_max = atol(dm);
  if (const char* fw = getenv("PQA_FLUSH_WB8_MAX")) h->flush_wb8_max = atol(fw);
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
  if (h->nup > PQA_MAXN || h->ndn > PQA_MAXN) FAIL("more than 64 electrons per spin channel is not supported by the one-wave determinant tile");
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
      if (sys->nL > 0 && sys->shell_l[s] > 3) FAIL("periodic orbitals: shells up to f (l <= 3); g and h shells are implemented for open systems only");
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
      if (h->nmo[s] > 64) FAIL("more than 64 orbitals per spin are not supported by the contraction tiles");
      const int nt = (h->nmo[s] + 15) / 16;
      h->nt[s] = nt <= 1 ? 1 : (nt == 2 ? 2 : 4);
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
    PbcDev* dp;
    TRY(upload_table(h, &P, (size_t)1, &dp));
    S.pb = dp;
  }
  TRY(upload_table(h, sys->acoeff, (size_t)h->natom * h->na * 2, &h->d_acoeff)); S.acoeff = h->d_acoeff;
  TRY(upload_table(h, sys->bcoeff, (size_t)h->nb * 3, &h->d_bcoeff)); S.bcoeff = h->d_bcoeff;
  S.na3 = h->na3; S.nb3 = h->nb3; S.rcut_a3 = sys->rcut_a3; S.rcut_b3 = sys->rcut_b3;
  for (int k = 0; k < h->na3; ++k) { S.a3_kind[k] = sys->a3_kind[k]; S.a3_param[k] = sys->a3_param[k]; S.a3_aux[k] = 1.0 / (3.0 + sys->a3_param[k]); }
  for (int k = 0; k < h->nb3; ++k) { S.b3_kind[k] = sys->b3_kind[k]; S.b3_param[k] = sys->b3_param[k]; S.b3_aux[k] = 1.0 / (3.0 + sys->b3_param[k]); }
  TRY(upload_table<double>(h, nullptr, (size_t)h->natom * h->na3 * h->na3 * h->nb3 * 3, &h->d_c3)); S.c3 = h->d_c3;
  if (h->has_j3 && sys->ccoeff) TRY(set_c3(h, sys->ccoeff));
  {  // the three-body scratch sits behind whatever else a kernel keeps in dynamic LDS
    const size_t n = std::max(sys->nelec_up, sys->nelec_dn);
    const size_t other = std::max((n * (n + 1) + 3 * n + 64) * sizeof(double),
                                  (size_t)std::max(sys->has_slater ? std::max(sys->ndet_up, sys->ndet_dn) : 1, 1) * 5 * sizeof(double));
    S.j3_off = (int)((other + 7) / 8);
  }
  S.necp = h->necp;
  if (h->necp > 0) {
    const int nchan = sys->ecp_chan_off[h->necp];
    const int nterm = sys->ecp_term_off[nchan];
    h->ecp_nchan = nchan; h->ecp_nterm = nterm;
    for (int k = 0; k < h->necp; ++k)
      if (sys->ecp_chan_off[k + 1] - sys->ecp_chan_off[k] > PQA_MAXCHAN) FAIL("ECP with more than 4 non-local channels");
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
  // quadrature directions (eval_ecp.py:278-336): octahedral 6, icosahedral 12
  std::vector<double> quad;
  const double oa[6][3] = {{-1, 0, 0}, {0, -1, 0}, {0, 0, -1}, {0, 0, 1}, {0, 1, 0}, {1, 0, 0}};
  for (auto& p : oa) quad.insert(quad.end(), p, p + 3);
  {
    const double b1 = std::atan(2.0), pi = std::acos(-1.0);
    std::vector<double> th = {0.0, pi}, ph = {0.0, 0.0};
    for (int k = 0; k < 10; ++k) { th.push_back(k % 2 == 0 ? b1 : pi - b1); ph.push_back(k * pi / 5.0); }
    for (int i = 0; i < 12; ++i) {
      quad.push_back(std::sin(th[i]) * std::cos(ph[i]));
      quad.push_back(std::sin(th[i]) * std::sin(ph[i]));
      quad.push_back(std::cos(th[i]));
    }
  }
  TRY(upload_table(h, quad.data(), quad.size(), &h->d_quad));
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
                    &h->b_ewgt[1], &h->b_epte[0], &h->b_epte[1], &h->b_emo[0], &h->b_emo[1], &h->b_ecp, &h->b_xt, &h->b_Tt[0], &h->b_Tt[1], &h->b_rc[0], &h->b_rc[1], &h->b_sel[0], &h->b_sel[1], &h->b_auxt, &h->b_kpart, &h->b_rbuf, &h->b_vbuf, &h->b_act, &h->b_alt_x, &h->b_alt_T[0], &h->b_alt_T[1], &h->b_alt_dsign[0], &h->b_alt_dsign[1], &h->b_alt_dlog[0], &h->b_alt_dlog[1], &h->b_alt_cache[0], &h->b_alt_cache[1], &h->b_alt_aval, &h->b_alt_bval, &h->b_alt_j3u, &h->b_rsidx, &h->b_tpos, &h->b_twgt, &h->b_tlive, &h->b_trat, &h->b_tmcnt, &h->b_tmoff, &h->b_tmpass, &h->b_tmamp, &h->b_tmacc, &h->b_tmidx, &h->b_tmapos, &h->b_tmu, &h->b_tmtile, &h->b_tmaoff, &h->b_tmptw, &h->b_tmmarks, &h->b_dmcw, &h->b_dmcold, &h->b_dmcr2, &h->b_dmcout, &h->b_j3u, &h->b_dwrap, &h->b_wrap, &h->b_epass, &h->b_eptw[0], &h->b_eptw[1], &h->b_econ[0], &h->b_econ[1], &h->b_eu0[0], &h->b_eu0[1], &h->b_tves, &h->b_pgdet, &h->b_pbcd0, &h->b_pbcmask, &h->b_pbcth, &h->b_tmuold};
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
  if (h->stream) (void)hipStreamDestroy(h->stream);
  delete h;
}

// ---------------------------------------------------------------- parameters
extern "C" int pqa_set_param(pqa_handle_t* h, const char* name, const double* data, int64_t n) {
  HIPCHK(hipSetDevice(h->device));
  HIPCHK(hipStreamSynchronize(h->stream));
  const std::string k(name);
  auto expect = [&](int64_t want) { return n == want; };
  if (k == "acoeff") {
    if (!expect((int64_t)h->natom * h->na * 2)) FAIL("acoeff size mismatch");
    HIPCHK(hipMemcpy(h->d_acoeff, data, n * sizeof(double), hipMemcpyDefault));
  } else if (k == "bcoeff") {
    if (!expect((int64_t)h->nb * 3)) FAIL("bcoeff size mismatch");
    HIPCHK(hipMemcpy(h->d_bcoeff, data, n * sizeof(double), hipMemcpyDefault));
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
  else FAIL("unknown parameter name");
  if (n != want) FAIL("parameter size mismatch");
  HIPCHK(hipMemcpy(out, src, n * sizeof(double), hipMemcpyDefault));
  return 0;
}

// ---------------------------------------------------------------- orbital kernel launch
static ChunkTab tabx(const pqa_handle* h, int tabi) {  // the chunk table + where this launch's rows go
  ChunkTab T = h->tab[tabi];
  T.out_sel = h->out_sel;
  T.out_slot_stride = h->out_slot_stride;
  return T;
}
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
  switch (h->nt[spin]) {
    case 1: hipLaunchKernelGGL((k_orb<NCOMP, 1, KC, TP, LT>), grid, block, 0, h->stream, h->S, tabx(h, tabi), spin, pa, P, out); break;
    case 2: hipLaunchKernelGGL((k_orb<NCOMP, 2, KC, TP, LT>), grid, block, 0, h->stream, h->S, tabx(h, tabi), spin, pa, P, out); break;
    default: hipLaunchKernelGGL((k_orb<NCOMP, 4, KC, TP, LT>), grid, block, 0, h->stream, h->S, tabx(h, tabi), spin, pa, P, out); break;
  }
}
// whole-K kernel for small 5-component launches (k_orb_wide, pqa_ao.hpp)
static bool wide_wanted(const pqa_handle* h, int tabi, long P, int ncomp) {
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

// periodic orbitals: lattice-summed shells, 64-point tiles, tables through the scalar cache
template <int NCOMP, int KC>
static int launch_orb_pbc(pqa_handle* h, int tabi, int spin, PointAddr pa, long P, double* out) {
  TRY(ensure(h, h->b_pbcd0, (size_t)h->natom * (h->twist ? 5 : 3) * P * sizeof(double)));
  const int NW = h->pbc_nw;
  TRY(ensure(h, h->b_pbcmask, (size_t)h->natom * NW * P * sizeof(unsigned long long)));
  if (h->twist) TRY(ensure(h, h->b_pbcth, (size_t)2 * P * sizeof(double)));
  hipLaunchKernelGGL(k_pbc_prepass, dim3((unsigned)((P + PQA_PRE_NT - 1) / PQA_PRE_NT), (unsigned)h->natom), dim3(PQA_PRE_NT), 0, h->stream, h->S, pa, P, NW,
                     (double*)h->b_pbcd0.p, (unsigned long long*)h->b_pbcmask.p, (double*)h->b_pbcth.p);
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
      hipLaunchKernelGGL(k_row_phase, dim3((unsigned)((nel + 255) / 256)), dim3(256), 0, h->stream, out, P, NCOMP, h->nmo[spin],
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
    if (h->out_sel) hipLaunchKernelGGL(k_zero_rows, dim3((unsigned)P, (unsigned)((NCOMP * h->nmo[spin] + 255) / 256)), dim3(256), 0, h->stream, out,
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
    hipLaunchKernelGGL(k_row_phase, dim3((unsigned)((nel + 255) / 256)), dim3(256), 0, h->stream, out, P, NCOMP, h->nmo[spin],
                       (const double*)h->b_pbcth.p, h->out_sel, h->out_slot_stride);
  }
  return 0;
}
template <int NCOMP, int KC, int TP>
static void launch_orb_t(pqa_handle* h, int tabi, int spin, PointAddr pa, long P, double* out) {
  if (h->nshell <= PQA_WS_MAXSH && (int)h->S.nprim <= PQA_WS_MAXP && !h->orb_notab) launch_orb_t2<NCOMP, KC, TP, true>(h, tabi, spin, pa, P, out);
  else launch_orb_t2<NCOMP, KC, TP, false>(h, tabi, spin, pa, P, out);
}

// out[p][ncomp][nmo_spin]
// out_sel / slot_stride: two-slot output (ChunkTab::out_sel), else plain rows
static int launch_orb_impl(pqa_handle* h, int spin, PointAddr pa, long P, int ncomp, double* out);
static int launch_orb(pqa_handle* h, int spin, PointAddr pa, long P, int ncomp, double* out, const unsigned char* out_sel = nullptr,
                      long slot_stride = 0) {
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
  int tp = (P >= (long)64 * 512) ? 64 : 32;
  if (h->orb_tp == 32 || h->orb_tp == 64) tp = h->orb_tp;
  // measured on MI355X: below ~2 blocks per CU the wave-specialised schedule wins (its
  // producer and consumer waves overlap inside one block); with >= 4 resident blocks per CU the plain kernel does
  const bool want_ws = h->orb_ws < 0 ? (P < (long)64 * 512) : (h->orb_ws != 0);
  if (h->S.nL > 0) {
    if (ncomp == 5) { if (h->orb_kc5 == 32) TRY((launch_orb_pbc<5, 32>(h, 1, spin, pa, P, out))); else TRY((launch_orb_pbc<5, 16>(h, 0, spin, pa, P, out))); }
    else if (ncomp == 1) TRY((launch_orb_pbc<1, 32>(h, 1, spin, pa, P, out)));
    else FAIL("orbital kernel supports ncomp 1 or 5");
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

static PointAddr plain_points(const double* base, long P) {
  PointAddr pa;
  pa.base = base;
  pa.group = (int)std::max<long>(P, 1);
  pa.group_stride = 0;
  return pa;
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
  if (ncomp == 1) hipLaunchKernelGGL(k_ao<1>, grid, block, 0, h->stream, h->S, (const double*)h->b_pts.p, (long)npts, (double*)h->b_ao.p);
  else if (ncomp == 4) hipLaunchKernelGGL(k_ao<4>, grid, block, 0, h->stream, h->S, (const double*)h->b_pts.p, (long)npts, (double*)h->b_ao.p);
  else hipLaunchKernelGGL(k_ao<5>, grid, block, 0, h->stream, h->S, (const double*)h->b_pts.p, (long)npts, (double*)h->b_ao.p);
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
  if (ncomp == 1) hipLaunchKernelGGL(k_ao<1>, grid, block, 0, h->stream, h->S, (const double*)h->b_pts.p, (long)npts, (double*)h->b_ao.p);
  else hipLaunchKernelGGL(k_ao<5>, grid, block, 0, h->stream, h->S, (const double*)h->b_pts.p, (long)npts, (double*)h->b_ao.p);
  const long rows = (long)ncomp * npts;
  hipLaunchKernelGGL(k_mo_valu, dim3((unsigned)((rows * nmo + 255) / 256)), dim3(256), 0, h->stream, (const double*)h->b_ao.p,
                     (const double*)h->d_mo[spin], rows, h->nao, nmo, (double*)h->b_out.p);
  TRY(check_launch(h, "k_mo_valu"));
  return copy_out(h, out, h->b_out.p, nout * sizeof(double));
}

// ---------------------------------------------------------------- walker state allocation
static int ensure_walkers(pqa_handle* h, long W) {
  if (W <= 0) FAIL("number of walkers must be positive");
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
    hipLaunchKernelGGL(k_jastrow_recompute, dim3((unsigned)h->W), dim3(64), 0, h->stream, h->S, h->js);
    TRY(check_launch(h, "k_jastrow_recompute"));
  }
  h->jas_stale = false;
  return 0;
}

static size_t lds_j3(const pqa_handle* h) {  // bytes needed by kernels that call jas_eval with the three-body term
  return h->has_j3 ? ((size_t)h->S.j3_off + (size_t)h->natom * (3 + 6 * h->na3 * h->nb3)) * sizeof(double) : 0;
}
static size_t lds_sm(const pqa_handle* h) {
  const size_t n = std::max(h->nup, h->ndn);
  return std::max((n * (n + 1) + 2 * n + 64 + n) * sizeof(double), lds_j3(h));
}
static size_t lds_det(const pqa_handle* h, int ncomp) {
  return std::max((size_t)std::max(h->ndet_s[0], h->ndet_s[1]) * ncomp * sizeof(double), lds_j3(h));
}

static int slater_rebuild(pqa_handle* h) {  // cache + inverse + determinants from js.x
  const int nel[2] = {h->nup, h->ndn};
  for (int s = 0; s < 2; ++s) {
    if (nel[s] == 0) continue;
    PointAddr pa;
    pa.base = h->js.x + (size_t)(s ? h->nup : 0) * 3;
    pa.group = nel[s];
    pa.group_stride = (long)h->N * 3;
    TRY(launch_orb(h, s, pa, h->W * nel[s], 5, h->st.cache[s]));
    const size_t lds = (h->cplx ? 2 : 1) * ((size_t)nel[s] * (nel[s] + 1)) * sizeof(double) + (size_t)nel[s] * sizeof(int) + 16;
    if (h->cplx) hipLaunchKernelGGL(k_build_invert_c, dim3((unsigned)(h->W * h->ndet_s[s])), dim3(64), lds, h->stream, h->S, h->st, s, h->W);
    else hipLaunchKernelGGL(k_build_invert, dim3((unsigned)(h->W * h->ndet_s[s])), dim3(64), lds, h->stream, h->S, h->st, s, h->W);
    TRY(check_launch(h, "k_build_invert"));
  }
  return 0;
}

static int slater_value_dev(pqa_handle* h) {
  if (h->cplx) hipLaunchKernelGGL(k_slater_value_c, dim3((unsigned)h->W), dim3(64), 0, h->stream, h->S, h->st, (double*)h->b_sign.p, (double*)h->b_log.p);
  else hipLaunchKernelGGL(k_slater_value, dim3((unsigned)h->W), dim3(64), 0, h->stream, h->S, h->st, (double*)h->b_sign.p, (double*)h->b_log.p);
  return check_launch(h, "k_slater_value");
}

extern "C" int pqa_slater_recompute(pqa_handle_t* h, const double* configs, int64_t W, double* sign, double* logabs) {
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
  if (h->cplx) FAIL("complex orbitals: only the wave-function protocol entry points are implemented so far");
  HIPCHK(hipSetDevice(h->device));
  if (!h->has_slater || h->W == 0) FAIL("Slater state not initialised (call recompute)");
  const long W = h->W;
  TRY(slater_value_dev(h));  // sign / log of the determinant expansion -> b_sign, b_log
  TRY(ensure(h, h->b_pgdet, (size_t)W * h->ndet * sizeof(double)));
  hipLaunchKernelGGL(k_pgrad_det, dim3((unsigned)((W * h->ndet + 255) / 256)), dim3(256), 0, h->stream, h->S, h->st,
                     (const double*)h->b_sign.p, (const double*)h->b_log.p, W, (double*)h->b_pgdet.p);
  TRY(check_launch(h, "k_pgrad_det"));
  if (d_det) TRY(copy_out(h, d_det, h->b_pgdet.p, (size_t)W * h->ndet * sizeof(double)));
  double* outs[2] = {d_mo_up, d_mo_dn};
  if (!d_mo_up && !d_mo_dn) return 0;
  const size_t nao_all = (size_t)W * h->N * h->nao;
  if (nao_all * sizeof(double) > ((size_t)16 << 30)) FAIL("orbital-coefficient gradients need the AO values of all electrons: too many walkers for one call");
  TRY(ensure(h, h->b_ao, nao_all * sizeof(double)));
  hipLaunchKernelGGL(k_ao<1>, dim3((unsigned)((W * h->N + 63) / 64)), dim3(64), 0, h->stream, h->S, (const double*)h->js.x, W * h->N,
                     (double*)h->b_ao.p);
  TRY(check_launch(h, "k_ao"));
  for (int s = 0; s < 2; ++s) {
    const int n = s ? h->ndn : h->nup;
    if (!outs[s] || n == 0 || h->nmo[s] == 0) continue;
    const size_t nout = (size_t)W * h->nao * h->nmo[s];
    TRY(ensure(h, h->b_out, nout * sizeof(double)));
    hipLaunchKernelGGL(k_pgrad_mo, dim3((unsigned)W), dim3(256), (size_t)h->ndet_s[s] * sizeof(double), h->stream, h->S, h->st, s,
                       (const double*)h->b_ao.p, (const double*)h->b_pgdet.p, (const int*)h->d_colmap[s], (double*)h->b_out.p);
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
  hipLaunchKernelGGL(k_has_zero, dim3((unsigned)((count + 255) / 256)), dim3(256), 0, h->stream, (const double*)h->st.dlog[spin], count, (int*)h->b_flag.p);
  TRY(check_launch(h, "k_has_zero"));
  return copy_out(h, flag, h->b_flag.p, sizeof(int));
}

extern "C" int pqa_slater_update(pqa_handle_t* h, int e, const double* epos, const uint8_t* mask, int use_saved) {
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
  if (h->cplx) hipLaunchKernelGGL(k_sm_update_c, dim3((unsigned)W), dim3(64), 2 * lds_sm(h), h->stream, h->S, h->st, e,
                                  (const double*)h->b_motmp.p, 5 * nmo, dm, 1);
  else hipLaunchKernelGGL(k_sm_update, dim3((unsigned)W), dim3(64), lds_sm(h), h->stream, h->S, h->st, e, (const double*)h->b_motmp.p,
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
  TRY(sync_aos(h));
  HIPCHK(hipSetDevice(h->device));
  if (!h->has_j2) FAIL("handle has no two-body Jastrow factor");
  if (h->W != W) TRY(ensure_walkers(h, W));
  h->jas_stale = false;
  TRY(copy_in(h, h->js.x, configs, (size_t)W * h->N * 3 * sizeof(double)));
  hipLaunchKernelGGL(k_jastrow_recompute, dim3((unsigned)W), dim3(64), 0, h->stream, h->S, h->js);
  TRY(check_launch(h, "k_jastrow_recompute"));
  return pqa_jastrow_value(h, logval);
}

extern "C" int pqa_jastrow_value(pqa_handle_t* h, double* logval) {
  TRY(sync_aos(h));
  HIPCHK(hipSetDevice(h->device));
  if (!h->has_j2 || h->W == 0) FAIL("Jastrow state not initialised (call recompute)");
  TRY(jas_refresh(h));
  hipLaunchKernelGGL(k_jastrow_value, dim3((unsigned)h->W), dim3(64), 0, h->stream, h->S, h->js, (double*)h->b_ju.p);
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
  hipLaunchKernelGGL(k_jastrow_eval, dim3((unsigned)nrow), dim3(64), lds_j3(h), h->stream, h->S, h->js, e, (const double*)h->b_pts.p,
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
  hipLaunchKernelGGL(k_j3_value, dim3((unsigned)h->W), dim3(64), lds_j3(h), h->stream, h->S, h->js, (double*)h->b_j3u.p);
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
  HIPCHK(hipFuncSetAttribute(reinterpret_cast<const void*>(k_j3_pgrad), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  hipLaunchKernelGGL(k_j3_pgrad, dim3((unsigned)W), dim3(64), lds, h->stream, h->S, h->js, (double*)h->b_out.p);
  TRY(check_launch(h, "k_j3_pgrad"));
  return copy_out(h, d_ccoeff, h->b_out.p, (size_t)W * E * sizeof(double));
}

extern "C" int pqa_j3_recompute(pqa_handle_t* h, const double* configs, int64_t W, double* logval) {
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
  hipLaunchKernelGGL(k_move_x, dim3((unsigned)((W + 255) / 256)), dim3(256), 0, h->stream, h->js, h->N, e, (const double*)h->b_newpos.p, dm, W);
  TRY(check_launch(h, "k_move_x"));
  HIPCHK(hipStreamSynchronize(h->stream));
  return 0;
}

extern "C" int pqa_jastrow_update(pqa_handle_t* h, int e, const double* epos, const uint8_t* mask) {
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
  hipLaunchKernelGGL(k_jastrow_update, dim3((unsigned)W), dim3(64), 0, h->stream, h->S, h->js, e, (const double*)h->b_newpos.p, dm);
  TRY(check_launch(h, "k_jastrow_update"));
  HIPCHK(hipStreamSynchronize(h->stream));
  return 0;
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
    hipLaunchKernelGGL(k_jastrow_value, dim3((unsigned)W), dim3(64), 0, h->stream, h->S, h->js, (double*)h->b_ju.p);
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
  TRY(sync_aos(h));
  HIPCHK(hipSetDevice(h->device));
  TRY(ensure_walkers(h, W));
  h->jas_stale = false;
  TRY(copy_in(h, h->js.x, configs, (size_t)W * h->N * 3 * sizeof(double)));
  if (h->has_slater) TRY(slater_rebuild(h));
  if (h->has_j2) {
    hipLaunchKernelGGL(k_jastrow_recompute, dim3((unsigned)W), dim3(64), 0, h->stream, h->S, h->js);
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

// energy of the resident walkers into device buffer b_en (6,W)
static void transpose(pqa_handle* h, const double* in, double* out, long R, long C) {  // in [R][C] -> out [C][R]
  if (R <= 0 || C <= 0) return;
  hipLaunchKernelGGL(k_transpose, dim3((unsigned)((C + 31) / 32), (unsigned)((R + 31) / 32)), dim3(32, 8), 0, h->stream, in, out, R, C);
}

static LwState lw_state(pqa_handle* h) {
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
static int lw_from_aos(pqa_handle* h, bool with_cache = true) {
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
      if (with_cache) hipLaunchKernelGGL(k_cache_to_rc, dim3((unsigned)W, (unsigned)n, (unsigned)((row + 255) / 256)), dim3(256), 0, h->stream,
                                         (const double*)h->st.cache[s], (double*)h->b_rc[s].p, (uint8_t*)h->b_sel[s].p, (int)n, row, W);
    }
  }
  return check_launch(h, "k_transpose");
}
// SoA -> AoS: coordinates and inverses (what the ECP kernels read); with_cache also the orbital cache
static int lw_to_aos(pqa_handle* h, bool with_cache) {
  const long W = h->W;
  const int nel[2] = {h->nup, h->ndn};
  transpose(h, (const double*)h->b_xt.p, h->js.x, (long)h->N * 3, W);
  for (int s = 0; s < 2; ++s) {
    const long n = nel[s];
    transpose(h, (const double*)h->b_Tt[s].p, h->st.T[s], (h->cplx ? 2 : 1) * n * n, W);
    const int row = 5 * h->nmo[s];
    if (with_cache && n > 0 && row > 0)
      hipLaunchKernelGGL(k_cache_from_rc, dim3((unsigned)W, (unsigned)n, (unsigned)((row + 255) / 256)), dim3(256), 0, h->stream,
                         (const double*)h->b_rc[s].p, (const uint8_t*)h->b_sel[s].p, h->st.cache[s], (int)n, row, W);
  }
  return check_launch(h, "k_transpose");
}

// A fused call on the lane-per-walker kernels leaves the live state in the SoA planes and only marks the walker-major arrays
// stale: back-to-back fused calls (the blocks of a VMC run) then skip both layout conversions (~13 GB of traffic per call at
// 65536 walkers of the 64-electron system, 7 ms), and whoever needs the walker-major state — every protocol entry, the
// energy entry, branching — converts it back first.
static int sync_aos(pqa_handle* h) {
  if (!h || !h->aos_stale) return 0;
  HIPCHK(hipSetDevice(h->device));
  h->aos_stale = false;
  return lw_to_aos(h, true);
}

static int energy_dev(pqa_handle* h, double threshold, const double* rot, const double* unif, uint64_t seed, uint32_t step,
                      bool soa_current = false, bool aos_T_needed = true) {
  const long W = h->W;
  bool soa_T = false;
  TRY(ensure(h, h->b_kc, (size_t)4 * W * sizeof(double)));
  TRY(ensure(h, h->b_en, (size_t)(h->cplx ? 7 : 6) * W * sizeof(double)));
  if (soa_current) {
    const dim3 gk((unsigned)((((W + 63) / 64 + 7) / 8) * 8 * ((h->N + PQA_KIN_EB - 1) / PQA_KIN_EB))), bk(64, PQA_KIN_EB);  // see k_kinetic_lw
    if (h->cplx) {
      if (h->S.pbc) hipLaunchKernelGGL((k_kinetic_lw<true, true>), gk, bk, 0, h->stream, h->S, lw_state(h), (int)h->has_jastrow, W, (double*)h->b_kpart.p);
      else hipLaunchKernelGGL((k_kinetic_lw<false, true>), gk, bk, 0, h->stream, h->S, lw_state(h), (int)h->has_jastrow, W, (double*)h->b_kpart.p);
    } else if (h->S.pbc)
      hipLaunchKernelGGL(k_kinetic_lw<true>, gk, bk, 0, h->stream, h->S, lw_state(h), (int)h->has_jastrow, W, (double*)h->b_kpart.p);
    else
      hipLaunchKernelGGL(k_kinetic_lw<false>, gk, bk, 0, h->stream, h->S, lw_state(h), (int)h->has_jastrow, W, (double*)h->b_kpart.p);
    hipLaunchKernelGGL(k_kinetic_reduce, dim3((unsigned)((W + 255) / 256)), dim3(256), 0, h->stream, (const double*)h->b_kpart.p,
                       h->N, W, (double*)h->b_kc.p);
    TRY(check_launch(h, "k_kinetic_lw"));
    // the ECP kernels read walker-major coordinates; the inverse only when the wave-per-walker accumulation runs (or the
    // caller works on the walker-major state next: the DMC step's T-moves) — the thread-per-point kernel takes the planes
    soa_T = !aos_T_needed && (!h->cplx || h->ecp_point_lw) && h->ndet == 1 && !h->has_j3 && h->ecp_wave == 0 && h->ecp_soa_t;
    if (h->necp > 0) {
      if (soa_T) { transpose(h, (const double*)h->b_xt.p, h->js.x, (long)h->N * 3, W); TRY(check_launch(h, "k_transpose")); }
      else TRY(lw_to_aos(h, false));
    }
  } else {
    {  // four waves per walker while the launch is too small to fill the chip with one
      const bool kc4 = h->ecp_acc_waves == 4 || (h->ecp_acc_waves == 0 && W * h->N <= 32768);
      const size_t st_ = (size_t)(h->cplx ? 2 : 1) * lds_det(h, 5);
      const int str_ = (int)(st_ / sizeof(double));
#define PQA_KC(CXF, NV) hipLaunchKernelGGL((k_kinetic_coulomb<CXF, NV>), dim3((unsigned)W), dim3(64 * NV), NV * st_, h->stream, h->S, h->st, h->js, \
                                           (int)h->has_slater, (int)h->has_jastrow, W, (double*)h->b_kc.p, str_)
      if (h->cplx) { if (kc4) PQA_KC(true, 4); else PQA_KC(true, 1); }
      else { if (kc4) PQA_KC(false, 4); else PQA_KC(false, 1); }
#undef PQA_KC
    }
    TRY(check_launch(h, "k_kinetic_coulomb"));
  }
  if (h->S.pbc) {
    if (!h->ew_set) FAIL("periodic Coulomb energy needs the Ewald tables (pqa_set_ewald)");
    const bool soa = soa_current && h->necp == 0;  // with ECPs the coordinates were just transposed back
    const double* x = soa ? (const double*)h->b_xt.p : h->js.x;
    const size_t lds_ew = ((size_t)h->N * 3 + (h->ew.gn ? (size_t)h->N * 3 * (h->ew.nmax + 1) * 2 : 0)) * sizeof(double);
    hipLaunchKernelGGL(k_ewald, dim3((unsigned)W), dim3(PQA_EWALD_T), lds_ew, h->stream, h->S, h->ew, x,
                       soa ? 1L : (long)h->N * 3, soa ? 3 * W : 3L, soa ? W : 1L, W, (double*)h->b_kc.p);
    TRY(check_launch(h, "k_ewald"));
  }
  const double* d_ecp = nullptr;
  h->last_ecp_points = 0;
  if (h->necp > 0) {
    const size_t nrot = (size_t)h->N * h->necp;
    TRY(ensure(h, h->b_rot, nrot * 9 * sizeof(double)));
    if (rot) TRY(copy_in(h, h->b_rot.p, rot, nrot * 9 * sizeof(double)));
    else {
      hipLaunchKernelGGL(k_gen_rot, dim3((unsigned)((nrot + 63) / 64)), dim3(64), 0, h->stream, (int)nrot, seed, step, (double*)h->b_rot.p);
      TRY(check_launch(h, "k_gen_rot"));
    }
    EcpBuf B{};
    B.rot = (const double*)h->b_rot.p;
    if (unif) {
      TRY(ensure(h, h->b_eunif, nrot * W * sizeof(double)));
      TRY(copy_in(h, h->b_eunif.p, unif, nrot * W * sizeof(double)));
      B.unif = (const double*)h->b_eunif.p;
    }
    B.quad = h->d_quad; B.seed = seed; B.step = step; B.threshold = threshold;
    TRY(ensure(h, h->b_elocal, W * sizeof(double)));
    // second-generation list passes (pqa_ecp.hpp): tables in LDS, four walkers per block, ATOM-major point lists
    const size_t tab_b = ecp_tab_bytes(h->necp, h->ecp_nchan, h->ecp_nterm);
    const bool ecp_t = h->ecp_lds && h->necp <= 64 && (long)h->necp * ((h->N + 63) / 64) <= 64 && tab_b <= 32768;
    // (atom-major lists where the orbital kernel gains from them: periodic cells, whose per-lane image walks then have similar
    // lengths within a tile — 2x2x2 diamond VMC +3 % at 32768 walkers; open systems gain nothing and pay a longer scan and sum)
    const long nseg = (ecp_t && h->ecp_atom_major && h->S.pbc) ? h->necp : 1, nsw = nseg * W;
    B.nseg = (int)nseg;
    TRY(ensure(h, h->b_ecnt, 2 * nsw * sizeof(int)));
    TRY(ensure(h, h->b_eoff, 2 * (nsw + 1) * sizeof(long)));
    TRY(ensure(h, h->b_ecp, (h->cplx ? 2 : 1) * W * sizeof(double)));
    TRY(ensure(h, h->b_epass, (size_t)W * h->necp * ((h->N + 63) / 64) * sizeof(unsigned long long)));
    B.local = (double*)h->b_elocal.p; B.cnt = (int*)h->b_ecnt.p; B.off = (long*)h->b_eoff.p;
    B.passbits = (unsigned long long*)h->b_epass.p;
    B.has_j2 = h->has_j2 ? 1 : 0;
    B.ue = (soa_current && h->has_j2) ? (const double*)h->b_kpart.p + (size_t)4 * h->N * W : nullptr;  // k_kinetic_lw left U_e there
    const dim3 g_t((unsigned)((W + PQA_ECP_WB - 1) / PQA_ECP_WB)), b_t(64 * PQA_ECP_WB);
    if (ecp_t) {
      if (h->S.pbc) hipLaunchKernelGGL(k_ecp_count_t<true>, g_t, b_t, tab_b, h->stream, h->S, h->js, B, h->ecp_nchan, h->ecp_nterm, W);
      else hipLaunchKernelGGL(k_ecp_count_t<false>, g_t, b_t, tab_b, h->stream, h->S, h->js, B, h->ecp_nchan, h->ecp_nterm, W);
    } else
    if (h->S.pbc) hipLaunchKernelGGL(k_ecp_count<true>, dim3((unsigned)W), dim3(64), 0, h->stream, h->S, h->js, B, W);
    else hipLaunchKernelGGL(k_ecp_count<false>, dim3((unsigned)W), dim3(64), 0, h->stream, h->S, h->js, B, W);
    // device-wide scans of the two spins' point counts (the one-block k_scan2 took 0.26 ms at 65536 walkers)
    TRY(ensure(h, h->b_tmmarks, 4 * sizeof(long)));
    TRY(scan_ints(h, (const int*)B.cnt, B.off, nsw, nsw, (long*)h->b_tmmarks.p));
    TRY(scan_ints(h, (const int*)B.cnt + nsw, B.off + (nsw + 1), nsw, nsw, (long*)h->b_tmmarks.p + 2));
    TRY(check_launch(h, "k_ecp_count/k_scan2"));
    long tot[2];
    TRY(copy_in(h, &tot[0], B.off + nsw, sizeof(long)));
    TRY(copy_out(h, &tot[1], B.off + (nsw + 1) + nsw, sizeof(long)));
    h->last_ecp_points = tot[0] + tot[1];
    for (int s = 0; s < 2; ++s) {
      const size_t n = (size_t)std::max<long>(tot[s], 1);
      TRY(ensure(h, h->b_epts[s], n * 3 * sizeof(double)));
      TRY(ensure(h, h->b_ewgt[s], n * sizeof(double)));
      TRY(ensure(h, h->b_epte[s], n * sizeof(int)));
      TRY(ensure(h, h->b_eptw[s], n * sizeof(int)));
      TRY(ensure(h, h->b_econ[s], (h->cplx ? 2 : 1) * n * sizeof(double)));
      TRY(ensure(h, h->b_eu0[s], n * sizeof(double)));
      B.ptw[s] = (int*)h->b_eptw[s].p;
      B.u0[s] = (double*)h->b_eu0[s].p;
      TRY(ensure(h, h->b_emo[s], n * std::max(h->nmo[s], 1) * sizeof(double)));
      B.pts[s] = (double*)h->b_epts[s].p; B.wgt[s] = (double*)h->b_ewgt[s].p; B.pte[s] = (int*)h->b_epte[s].p;
    }
    if (tot[0] + tot[1] > 0) {
      if (ecp_t) {
        if (B.ue) {
          if (h->S.pbc) hipLaunchKernelGGL((k_ecp_fill_t<true, true>), g_t, b_t, tab_b, h->stream, h->S, h->js, B, h->ecp_nchan, h->ecp_nterm, W);
          else hipLaunchKernelGGL((k_ecp_fill_t<false, true>), g_t, b_t, tab_b, h->stream, h->S, h->js, B, h->ecp_nchan, h->ecp_nterm, W);
        } else {
          if (h->S.pbc) hipLaunchKernelGGL((k_ecp_fill_t<true, false>), g_t, b_t, tab_b, h->stream, h->S, h->js, B, h->ecp_nchan, h->ecp_nterm, W);
          else hipLaunchKernelGGL((k_ecp_fill_t<false, false>), g_t, b_t, tab_b, h->stream, h->S, h->js, B, h->ecp_nchan, h->ecp_nterm, W);
        }
      } else
      if (h->S.pbc) hipLaunchKernelGGL(k_ecp_fill<true>, dim3((unsigned)W), dim3(64), 0, h->stream, h->S, h->js, B, W);
      else hipLaunchKernelGGL(k_ecp_fill<false>, dim3((unsigned)W), dim3(64), 0, h->stream, h->S, h->js, B, W);
      TRY(check_launch(h, "k_ecp_fill"));
      if (h->has_slater)
        for (int s = 0; s < 2; ++s)
          TRY(launch_orb(h, s, plain_points(B.pts[s], tot[s]), tot[s], 1, (double*)h->b_emo[s].p));
    }
    // wave-per-walker accumulation (complex determinants, several determinants, three-body factor): four waves share a walker's
    // points while the launch is too small to fill the chip with one (measured after the three-body / determinant-pass fixes: C4
    // +7 % at 2048 walkers, even at 4096, -7 % at 8192; the 32-electron twisted cell +1 % at 1024, -1.5 % at 2048, -6 % at 8192)
    const bool acc4 = h->ecp_acc_waves == 4 || (h->ecp_acc_waves == 0 && W * h->N <= 32768);
#define PQA_ECP_ACC(PB, CXF, SC) do { const size_t st_ = (size_t)(SC) * lds_det(h, 1); const int str_ = (int)(st_ / sizeof(double)); \
      if (acc4) hipLaunchKernelGGL((k_ecp_accum<PB, CXF, 4>), dim3((unsigned)W), dim3(256), 4 * st_, h->stream, h->S, h->st, h->js, B, (int)h->has_slater, \
                                   (int)h->has_jastrow, (const double*)h->b_emo[0].p, (const double*)h->b_emo[1].p, W, (double*)h->b_ecp.p, str_); \
      else hipLaunchKernelGGL((k_ecp_accum<PB, CXF, 1>), dim3((unsigned)W), dim3(64), st_, h->stream, h->S, h->st, h->js, B, (int)h->has_slater, \
                              (int)h->has_jastrow, (const double*)h->b_emo[0].p, (const double*)h->b_emo[1].p, W, (double*)h->b_ecp.p, str_); } while (0)
    const bool cx_points = h->cplx && soa_current && h->ndet == 1 && !h->has_j3 && h->ecp_wave == 0 && h->ecp_point_lw;  // thread per point on the complex planes
    if (h->cplx && !cx_points) {  // complex determinants: wave-per-walker accumulation in complex arithmetic
      if (h->S.pbc) PQA_ECP_ACC(true, true, 2); else PQA_ECP_ACC(false, true, 2);
    } else
    if (h->ndet == 1 && !h->has_j3 && h->ecp_wave == 0) {  // thread per point, then an ordered per-walker sum
      for (int s = 0; s < 2; ++s) {
        if (tot[s] <= 0) continue;
        const dim3 g((unsigned)((tot[s] + 255) / 256));
        const long n_s = s ? h->ndn : h->nup;
        const double* Tb = soa_T ? (const double*)h->b_Tt[s].p : (const double*)h->st.T[s];
        const long sw = soa_T ? 1 : n_s * n_s, si = soa_T ? n_s * W : n_s, sk = soa_T ? W : 1;
        // (the planes are the live state whenever this evaluation follows a lane-per-walker sweep, also where the walker-major copy
        // was refreshed for the caller's next step — the DMC loop's T-moves)
        if (cx_points) {
          if (h->S.pbc)
            hipLaunchKernelGGL((k_ecp_point_lw<true, true>), g, dim3(256), 0, h->stream, h->S, lw_state(h), B, s, (int)h->has_slater,
                               (int)h->has_jastrow, (const double*)h->b_emo[s].p, tot[s], W, (double*)h->b_econ[s].p);
          else
            hipLaunchKernelGGL((k_ecp_point_lw<false, true>), g, dim3(256), 0, h->stream, h->S, lw_state(h), B, s, (int)h->has_slater,
                               (int)h->has_jastrow, (const double*)h->b_emo[s].p, tot[s], W, (double*)h->b_econ[s].p);
        } else
        if (soa_current && !h->cplx && h->ecp_point_lw) {
          if (h->S.pbc)
            hipLaunchKernelGGL(k_ecp_point_lw<true>, g, dim3(256), 0, h->stream, h->S, lw_state(h), B, s, (int)h->has_slater,
                               (int)h->has_jastrow, (const double*)h->b_emo[s].p, tot[s], W, (double*)h->b_econ[s].p);
          else
            hipLaunchKernelGGL(k_ecp_point_lw<false>, g, dim3(256), 0, h->stream, h->S, lw_state(h), B, s, (int)h->has_slater,
                               (int)h->has_jastrow, (const double*)h->b_emo[s].p, tot[s], W, (double*)h->b_econ[s].p);
        } else
        if (h->S.pbc)
          hipLaunchKernelGGL(k_ecp_point<true>, g, dim3(256), 0, h->stream, h->S, h->st, h->js, B, s, (int)h->has_slater,
                             (int)h->has_jastrow, (const double*)h->b_emo[s].p, tot[s], (double*)h->b_econ[s].p, Tb, sw, si, sk);
        else
          hipLaunchKernelGGL(k_ecp_point<false>, g, dim3(256), 0, h->stream, h->S, h->st, h->js, B, s, (int)h->has_slater,
                             (int)h->has_jastrow, (const double*)h->b_emo[s].p, tot[s], (double*)h->b_econ[s].p, Tb, sw, si, sk);
      }
      hipLaunchKernelGGL(k_ecp_sum, dim3((unsigned)((W + 255) / 256)), dim3(256), 0, h->stream, B, (const double*)h->b_econ[0].p,
                         (const double*)h->b_econ[1].p, W, (double*)h->b_ecp.p, cx_points ? std::max<long>(tot[0], 1) : 0L,
                         cx_points ? std::max<long>(tot[1], 1) : 0L);
    } else {
      if (h->S.pbc) PQA_ECP_ACC(true, false, 1); else PQA_ECP_ACC(false, false, 1);
    }
#undef PQA_ECP_ACC
    TRY(check_launch(h, "k_ecp_accum"));
    d_ecp = (const double*)h->b_ecp.p;
  }
  hipLaunchKernelGGL(k_energy_assemble, dim3((unsigned)((W + 255) / 256)), dim3(256), 0, h->stream, (const double*)h->b_kc.p, d_ecp,
                     h->ii_energy, W, (double*)h->b_en.p, (int)h->cplx);
  return check_launch(h, "k_energy_assemble");
}

extern "C" int pqa_set_ewald(pqa_handle_t* h, double alpha, int32_t ng, const double* gpoints, const double* gweight,
                             const double* ion_cos, const double* ion_sin, double ee_const, double ei_const, double ii,
                             const int32_t* gidx, const double* recip) {
  HIPCHK(hipSetDevice(h->device));
  if (!h->S.pbc) FAIL("Ewald tables on an open-boundary handle");
  if (ng < 0 || !(alpha > 0.0)) FAIL("bad Ewald parameters");
  HIPCHK(hipStreamSynchronize(h->stream));
  double* d;
  TRY(upload_table(h, gpoints, (size_t)ng * 3, &d)); h->ew.g = d;
  TRY(upload_table(h, gweight, (size_t)ng, &d)); h->ew.gweight = d;
  TRY(upload_table(h, ion_cos, (size_t)ng, &d)); h->ew.ion_cos = d;
  TRY(upload_table(h, ion_sin, (size_t)ng, &d)); h->ew.ion_sin = d;
  h->ew.ng = ng; h->ew.alpha = alpha; h->ew.ee_const = ee_const; h->ew.ei_const = ei_const;
  h->ew.gn = nullptr; h->ew.nmax = 0;
  if (gidx && recip && ng > 0) {
    std::vector<int> gi((size_t)ng * 3);
    HIPCHK(hipMemcpy(gi.data(), gidx, gi.size() * sizeof(int), hipMemcpyDefault));
    int nmax = 0;
    for (int v : gi) nmax = std::max(nmax, std::abs(v));
    const size_t lds = ((size_t)h->N * 3 + (size_t)h->N * 3 * (nmax + 1) * 2) * sizeof(double);
    if (lds <= 64 * 1024) {  // otherwise stay with the direct sincos form
      int* dgi;
      TRY(upload_table(h, gi.data(), gi.size(), &dgi));
      h->ew.gn = dgi; h->ew.nmax = nmax;
      HIPCHK(hipMemcpy(h->ew.recip, recip, 9 * sizeof(double), hipMemcpyDefault));
    }
  }
  h->ii_energy = ii;
  h->ew_set = true;
  return 0;
}

extern "C" int pqa_get_wrap(pqa_handle_t* h, int32_t* wrap) {
  HIPCHK(hipSetDevice(h->device));
  if (!h->S.pbc) FAIL("open-boundary handle has no wrap counters");
  if (h->wrap_W != h->W || h->W == 0) FAIL("no fused sweep has run on the resident walkers");
  return copy_out(h, wrap, h->b_wrap.p, (size_t)h->W * h->N * 3 * sizeof(int));
}

extern "C" int pqa_energy(pqa_handle_t* h, double threshold, const double* rot, const double* unif, uint64_t seed, double* out) {
  TRY(sync_aos(h));
  HIPCHK(hipSetDevice(h->device));
  if (h->W == 0) FAIL("state not initialised (call pqa_wf_recompute)");
  h->saved_valid = false;
  TRY(energy_dev(h, threshold, rot, unif, seed, 0u));
  return copy_out(h, out, h->b_en.p, (size_t)(h->cplx ? 7 : 6) * h->W * sizeof(double));
}

// ---------------------------------------------------------------- walker-tile sweep (pqa_tile.hpp)
static bool tile_eligible(const pqa_handle* h) {
  if (h->lw_mode != 2 || !h->has_slater || h->ndet != 1 || h->has_j3 || h->cplx || h->S.pbc) return false;
  if (h->nup > 32 || h->ndn > 32 || h->nmo[0] > 32 || h->nmo[1] > 32) return false;
  for (int l : h->shell_l)
    if (l > 3) return false;
  const int nmo_pad = 16 * std::max(h->nt[0], h->nt[1]);
  return tile_lds_bytes(h->N, nmo_pad, h->nshell, (int)h->S.nprim, h->chunks[0].rows_pad) <= 160 * 1024 - 512;
}
// One sweep over all electrons for every walker, in one launch.  mb carries the step's tapes / seeds as for the other paths.
static int sweep_tile(pqa_handle* h, const MoveBuf& mb_in) {
  MoveBuf mb = mb_in;
  if (!mb.gauss || !mb.unif) {  // no replay tapes: draw this sweep's numbers from the Philox streams first
    const size_t NW = (size_t)h->N * h->W;
    TRY(ensure(h, h->b_gauss, NW * 3 * sizeof(double)));
    TRY(ensure(h, h->b_unif, NW * sizeof(double)));
    hipLaunchKernelGGL(k_tile_draws, dim3((unsigned)((NW + 255) / 256)), dim3(256), 0, h->stream, mb.seed, mb.step, h->N, h->W,
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

// ---------------------------------------------------------------- one sweep over the electrons (shared by VMC and DMC)
struct LwCtx {
  int Gm = 1, KB = 1, nmax = 1;
};
// Geometry of the lane-per-walker kernels and, when `lw`, the SoA copy of the state and its scratch.
static int lw_setup(pqa_handle* h, bool lw, LwCtx& c) {
  const long W = h->W;
  // thread groups per walker in k_step_lw.  Measured (tools/scratch/r3_step_abl*.sh, (H2O)8 step in ms at 4 / 8 / 16 groups):
  // 4096 walkers 6.38 / 5.15 / 4.67, 8192: 7.71 / 6.54 / 6.39, 16384: 10.7 / 9.8 / 11.1, 32768: 16.5 / 17.3 / 18.8, 65536: 30.8 / 32.7 / 37.3
  c.Gm = 4;
  while (c.Gm < 16 && (long)c.Gm * W < 2048L * 64) c.Gm *= 2;
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
// Lane-per-walker sweep, two launches per move: k_orb at the proposal, then k_step_lw = decide electron e + propose electron
// e + 1 (pqa_lw.hpp).  The two halves are launched apart where the blocked Sherman-Morrison update has to flush in between
// (e + 1 opens a new electron block of the same spin: its inverse row is only current after k_flush_lw).
template <bool PBC, bool CX>
static void launch_step_lw(pqa_handle* h, const LwState& L, const MoveBuf& mb, const StepArgs& a, int rowlen) {
  const dim3 grid((unsigned)((a.W + a.NW - 1) / a.NW)), block((unsigned)(a.NW * a.G));
  // small shards: the variant with every load issued at entry (k_step_pre, pqa_lw.hpp) where its scope covers the system
  // (one block per CU at most: the kernel holds ~360 registers per lane, one wave per SIMD)
  if (!CX && h->step_pre && a.NW < 64 && a.G >= 8 && grid.x <= 256 && h->S.occ_ident[0] && h->S.occ_ident[1] && h->S.nb <= PQA_JAS_NF && h->S.na <= PQA_JAS_NF &&
      h->N <= PQA_PRE_NP * a.G && h->S.natom <= PQA_PRE_NA * a.G && (a.e_acc < 0 || a.j_hi - a.j_lo <= a.G) && rowlen <= 64) {
#define PQA_STEP_P(NM) do { const size_t lds_p = ((size_t)8 * a.G + 3 * NM) * a.NW * sizeof(double); \
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
static int sweep_electrons_fused(pqa_handle* h, const MoveBuf& mb_in, const LwCtx& lc) {
  const long W = h->W;
  MoveBuf mb = mb_in;
  if (!mb.gauss && !mb.unif && W <= h->draws_max) {
    // small shards: the sweep's normals and uniforms drawn ahead by one launch from the same Philox streams (k_tile_draws) — in
    // k_step_lw the lead group's Box-Muller pairs are ~600 dependent instructions of every move's chain with one wave per SIMD
    const size_t NW = (size_t)h->N * W;
    TRY(ensure(h, h->b_gauss, NW * 3 * sizeof(double)));
    TRY(ensure(h, h->b_unif, NW * sizeof(double)));
    hipLaunchKernelGGL(k_tile_draws, dim3((unsigned)((NW + 255) / 256)), dim3(256), 0, h->stream, mb.seed, mb.step, h->N, W,
                       (double*)h->b_gauss.p, (double*)h->b_unif.p);
    mb.gauss = (const double*)h->b_gauss.p; mb.unif = (const double*)h->b_unif.p;
  }
  const int N = h->N, KB = lc.KB, nmax = lc.nmax;
  const LwState L = lw_state(h);
  const int cfi = h->cplx ? 2 : 1, rowlen = cfi * nmax;  // doubles per inverse row
  // thread groups per walker (lc.Gm: ~4 waves per SIMD's worth of threads) and walkers per block: 256 threads at most, so more
  // than 4 groups narrow the block to 32 or 16 walkers — which is also what spreads a small shard over the chip
  const int G = std::min(lc.Gm, 16);
  int NW = (G <= 4) ? 64 : 256 / G;
  if (h->lw_nw > 0 && h->lw_nw * G <= 256) NW = h->lw_nw;
  auto step = [&](int e_acc, int e_prop) {
    StepArgs a{};
    a.e_acc = e_acc; a.e_prop = e_prop; a.has_jastrow = (int)h->has_jastrow; a.G = G; a.NW = NW; a.W = W;
    if (e_acc >= 0) {
      const int s = e_acc >= h->nup, n_s = s ? h->ndn : h->nup, i_s = e_acc - (s ? h->nup : 0);
      const int q = i_s % KB;
      a.j_lo = i_s - q; a.j_hi = std::min(a.j_lo + KB, n_s);
      a.Rbuf = (double*)h->b_rbuf.p + (size_t)q * cfi * n_s * W;
      a.Vbuf = (double*)h->b_vbuf.p + (size_t)q * cfi * n_s * W;
      a.act = (uint8_t*)h->b_act.p + (size_t)q * W;
    }
    if (h->cplx) { if (h->S.pbc) launch_step_lw<true, true>(h, L, mb, a, rowlen); else launch_step_lw<false, true>(h, L, mb, a, rowlen); }
    else { if (h->S.pbc) launch_step_lw<true, false>(h, L, mb, a, rowlen); else launch_step_lw<false, false>(h, L, mb, a, rowlen); }
  };
  step(-1, 0);
  for (int e = 0; e < N; ++e) {
    const int s = e >= h->nup, n_s = s ? h->ndn : h->nup, i_s = e - (s ? h->nup : 0);
    const int q = i_s % KB, j_lo = i_s - q, j_hi = std::min(j_lo + KB, n_s);
    // the proposal's rows go straight into the slot of electron i_s the walker is not using (accepting flips the selector)
    TRY(launch_orb(h, s, plain_points(mb.newpos, W), W, 5, (double*)h->b_rc[s].p + (size_t)i_s * 2 * W * 5 * h->nmo[s],
                   (const unsigned char*)h->b_sel[s].p + (size_t)i_s * W, (long)W * 5 * h->nmo[s]));
    const bool block_done = (i_s == j_hi - 1);
    const bool need_flush = block_done && (j_hi - j_lo < n_s);
    // the next electron's inverse row is current after this move's commit unless it opens a new block of the SAME spin
    const bool fuse_next = (e + 1 < N) && !(need_flush && i_s + 1 < n_s);
    hipEvent_t pe1 = nullptr;
    if (h->profile && (e % (4 * (int)h->prof_stride)) == 1) {  // sparsely sampled full (decide + propose) launches: an event pair costs ~2 us of stream time
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
    step(e, fuse_next ? e + 1 : -1);
    if (pe1) { HIPCHK(hipEventRecord(pe1, h->stream)); h->prof3_launches += 1; }
    if (need_flush) {  // block finished: bring every other row of this spin up to date
      const int nq = j_hi - j_lo;
      hipEvent_t ce1 = nullptr;
      if (h->profile && ((j_lo / std::max(KB, 1)) % 4) == 0) {  // every 4th flush of a spin
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
#define PQA_FLUSH_W(NM, WB_) do { const size_t lds_f = (size_t)2 * nq * cfi * n_s * WB_ * sizeof(double); const dim3 gf((unsigned)((W + WB_ - 1) / WB_)); \
      if (h->cplx) hipLaunchKernelGGL((k_flush_lw<NM, true, WB_>), gf, dim3(256), lds_f, h->stream, h->S, L, s, (const double*)h->b_vbuf.p, (const double*)h->b_rbuf.p, (const uint8_t*)h->b_act.p, W, j_lo, j_hi, nq); \
      else hipLaunchKernelGGL((k_flush_lw<NM, false, WB_>), gf, dim3(256), lds_f, h->stream, h->S, L, s, (const double*)h->b_vbuf.p, (const double*)h->b_rbuf.p, (const uint8_t*)h->b_act.p, W, j_lo, j_hi, nq); } while (0)
#define PQA_FLUSH(NM) do { if (W <= h->flush_wb8_max) PQA_FLUSH_W(NM, 8); else PQA_FLUSH_W(NM, PQA_FLUSH_WB); } while (0)
      if (rowlen <= 8) PQA_FLUSH(8); else if (rowlen <= 16) PQA_FLUSH(16); else if (rowlen <= 32) PQA_FLUSH(32); else PQA_FLUSH(64);
#undef PQA_FLUSH_W
#undef PQA_FLUSH
      if (ce1) { HIPCHK(hipEventRecord(ce1, h->stream)); h->prof2_launches += 1; }
    }
    if (!fuse_next && e + 1 < N) step(-1, e + 1);
  }
  return 0;
}

// One proposal per electron, in index order, on the SoA state (lw: two launches per move, above) or the AoS state with the
// wave-per-walker kernels (multi-determinant, three-body, large complex determinants); mb.dmc selects the DMC variant.
static int sweep_electrons(pqa_handle* h, const MoveBuf& mb, bool lw, const LwCtx& lc) {
  if (lw) return sweep_electrons_fused(h, mb, lc);
  const long W = h->W;
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
  HIPCHK(hipSetDevice(h->device));
  if (h->W == 0) FAIL("state not initialised (call pqa_wf_recompute)");
  if (nsteps <= 0) return 0;
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
    if (tile) TRY(sweep_tile(h, mb));
    else TRY(sweep_electrons(h, mb, lw, lc));
    hipLaunchKernelGGL(k_sum_reset_int, dim3(1), dim3(1024), 0, h->stream, (int*)h->b_accw.p, W, (int*)h->b_acccnt.p + step);
    TRY(check_launch(h, "k_propose/k_accept"));
    if (accept_rec) TRY(copy_in(h, accept_rec + (size_t)step * N * W, h->b_accrec.p, (size_t)N * W));
    if (energy_mean) {
      TRY(energy_dev(h, threshold, ecp_rot ? ecp_rot + (size_t)step * nrot * 9 : nullptr,
                     ecp_unif ? ecp_unif + (size_t)step * nrot * W : nullptr, seed, (uint32_t)step, lw, /*aos_T_needed=*/false));
      hipLaunchKernelGGL(k_row_means, dim3(nen), dim3(256), 0, h->stream, (const double*)h->b_en.p, W, (double*)h->b_means.p + (size_t)step * nen);
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
    hipLaunchKernelGGL(k_jastrow_recompute, dim3((unsigned)n), dim3(64), 0, h->stream, h->S, h->js);
    rc = check_launch(h, "k_jastrow_recompute");
  }
  if (!rc && h->has_j3) {
    hipLaunchKernelGGL(k_j3_value, dim3((unsigned)n), dim3(64), lds_j3(h), h->stream, h->S, h->js, (double*)h->b_j3u.p + w0);
    rc = check_launch(h, "k_j3_value");
  }
  h->js = js0; h->st = st0; h->W = W0;
  return rc;
}

extern "C" int pqa_branch_exchange(pqa_handle_t* h, const int32_t* keep_src, int64_t nkeep, const double* recv_x, int64_t nrecv) {
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
static int scan_ints(pqa_handle* h, const int* c, long* o, long n, long Wm, long* marks) {
  const long nt = (n + 1023) / 1024;
  TRY(ensure(h, h->b_tmtile, (size_t)(nt + 1) * sizeof(long)));
  long* tile = (long*)h->b_tmtile.p;
  hipLaunchKernelGGL(k_scan_local, dim3((unsigned)nt), dim3(1024), 0, h->stream, c, o, n, tile);
  hipLaunchKernelGGL(k_scan_tiles, dim3(1), dim3(1024), 0, h->stream, tile, nt);
  hipLaunchKernelGGL(k_scan_add, dim3((unsigned)nt), dim3(1024), 0, h->stream, o, n, (const long*)tile, nt, Wm, marks);
  return check_launch(h, "k_scan_local/tiles/add");
}

// ---------------------------------------------------------------- fused DMC propagation
// nsteps steps of dmc_propagate (pyqmc/method/dmc.py:123-221) without leaving the device: T-moves, drift-diffusion with
// fixed-node rejection, local energy, weight update, weighted step averages.  Walker-per-wave kernels (the AoS state).
extern "C" int pqa_dmc_steps(pqa_handle_t* h, double tstep, int nsteps, double branchcut, double e_trial, double e_est, double threshold,
                             double* weights, const pqa_dmc_tapes_t* tp, uint64_t seed, double* step_avg, double* step_acc) {
  TRY(sync_aos(h));  // (the starting energy and the first T-moves read the walker-major state)
  HIPCHK(hipSetDevice(h->device));
  if (h->W == 0) FAIL("state not initialised (call pqa_wf_recompute)");
  if (nsteps <= 0) return 0;
  if (!weights || !step_avg || !step_acc) FAIL("pqa_dmc_steps: weights / step_avg / step_acc must not be NULL");
  const int navg = h->cplx ? 8 : 7;  // numbers per step in step_avg (complex: + the weighted mean of Im ecp = Im total)
  const long W = h->W;
  const int N = h->N, necp = h->necp, P = h->tm_P;
  const bool tmoves = necp > 0 && P > 0;
  if (tp && (!tp->gauss || !tp->unif)) FAIL("pqa_dmc_steps: a tape set needs gauss and unif");
  if (tp && necp > 0 && (!tp->ecp_rot || !tp->ecp_unif)) FAIL("pqa_dmc_steps: a tape set needs ecp_rot and ecp_unif for ECP systems");
  if (tp && tmoves && (!tp->tm_rot || !tp->tm_unif || !tp->tm_u1 || !tp->tm_u2)) FAIL("pqa_dmc_steps: a tape set needs the four T-move tapes");
  h->saved_valid = false;
  const int nmo_max = std::max(std::max(h->nmo[0], h->nmo[1]), 1);
  TRY(ensure(h, h->b_newpos, (size_t)W * 3 * sizeof(double)));
  TRY(ensure(h, h->b_aux, (size_t)W * 8 * sizeof(double)));
  TRY(ensure(h, h->b_accept, (size_t)W));
  TRY(ensure(h, h->b_acccnt, (size_t)nsteps * 2 * sizeof(int)));
  TRY(ensure(h, h->b_motmp, (size_t)W * 5 * nmo_max * sizeof(double)));
  TRY(ensure(h, h->b_accw, (size_t)W * sizeof(int)));
  TRY(ensure(h, h->b_dmcw, (size_t)W * sizeof(double)));
  TRY(ensure(h, h->b_dmcold, (size_t)2 * W * sizeof(double)));
  TRY(ensure(h, h->b_dmcr2, (size_t)2 * W * sizeof(double)));
  TRY(ensure(h, h->b_dmcout, (size_t)nsteps * navg * sizeof(double)));
  HIPCHK(hipMemsetAsync(h->b_acccnt.p, 0, (size_t)nsteps * 2 * sizeof(int), h->stream));
  HIPCHK(hipMemsetAsync(h->b_accw.p, 0, (size_t)W * sizeof(int), h->stream));
  HIPCHK(hipMemsetAsync(h->b_dmcr2.p, 0, (size_t)2 * W * sizeof(double), h->stream));
  TRY(copy_in(h, h->b_dmcw.p, weights, (size_t)W * sizeof(double)));
  if (h->S.pbc) {
    TRY(ensure(h, h->b_dwrap, (size_t)W * 3 * sizeof(int)));
    TRY(ensure(h, h->b_wrap, (size_t)W * N * 3 * sizeof(int)));
    HIPCHK(hipMemsetAsync(h->b_wrap.p, 0, (size_t)W * N * 3 * sizeof(int), h->stream));
    h->wrap_W = W;
  }
  if (tp) {
    TRY(ensure(h, h->b_gauss, (size_t)N * W * 3 * sizeof(double)));
    TRY(ensure(h, h->b_unif, (size_t)N * W * sizeof(double)));
  }
  const size_t nrot = (size_t)N * std::max(necp, 1);
  const int nkw = (std::max(necp, 1) + 63) / 64;
  if (tmoves) {
    const size_t NW = (size_t)N * W;
    TRY(ensure(h, h->b_tmcnt, NW * sizeof(int)));
    TRY(ensure(h, h->b_tmoff, (NW + 1) * sizeof(long)));
    TRY(ensure(h, h->b_tmpass, NW * nkw * sizeof(unsigned long long)));
    TRY(ensure(h, h->b_tmacc, NW * sizeof(int)));
    TRY(ensure(h, h->b_tmaoff, (NW + 1) * sizeof(long)));
    TRY(ensure(h, h->b_tmmarks, (size_t)(N + 1) * sizeof(long)));
    TRY(ensure(h, h->b_tmidx, NW * sizeof(int)));
    TRY(ensure(h, h->b_tmapos, NW * 3 * sizeof(double)));
    if (tp) TRY(ensure(h, h->b_tmu, (size_t)(2 + necp) * NW * sizeof(double)));
    TRY(ensure(h, h->b_rot, nrot * 9 * sizeof(double)));
  }
  const bool lw = h->lw_mode != 0 && h->has_slater && h->ndet == 1 && !h->has_j3 && (!h->cplx || std::max(h->nup, h->ndn) <= 32);
  LwCtx lc;
  TRY(lw_setup(h, lw, lc));
  const dim3 gw256((unsigned)((W + 255) / 256));
  double* eold = (double*)h->b_dmcold.p;
  double* r2 = (double*)h->b_dmcr2.p;
  std::vector<long> tm_accepted((size_t)nsteps, 0);
  // energy of the starting configuration (dmc.py:146-149)
  TRY(energy_dev(h, threshold, (tp && necp) ? tp->ecp_rot : nullptr, (tp && necp) ? tp->ecp_unif : nullptr, seed, 0u, false));
  hipLaunchKernelGGL(k_dmc_keep, gw256, dim3(256), 0, h->stream, (const double*)h->b_en.p, eold, eold + W, W);
  for (int step = 0; step < nsteps; ++step) {
    MoveBuf mb{};
    mb.newpos = (double*)h->b_newpos.p; mb.aux = (double*)h->b_aux.p; mb.accept = (uint8_t*)h->b_accept.p;
    mb.acc_w = (int*)h->b_accw.p; mb.seed = seed; mb.step = (uint32_t)step; mb.tstep = tstep;
    mb.dmc = 1; mb.r2_acc = r2; mb.r2_prop = r2 + W;
    if (h->S.pbc && !h->twist) { mb.dwrap = (int*)h->b_dwrap.p; mb.wrap = (int*)h->b_wrap.p; }  // twisted handles keep the walkers unfolded
    if (tmoves) {
      const size_t NW = (size_t)N * W;
      TmBuf B{};
      B.quad = h->d_quad; B.seed = seed; B.step = (uint32_t)step; B.tau = tstep; B.threshold = threshold; B.nofold = h->twist ? 1 : 0;
      B.cnt = (int*)h->b_tmcnt.p; B.off = (long*)h->b_tmoff.p; B.pass = (unsigned long long*)h->b_tmpass.p;
      long* d_marks = (long*)h->b_tmmarks.p;
      B.acc = (int*)h->b_tmacc.p; B.acc_off = (long*)h->b_tmaoff.p;
      B.acc_idx = (int*)h->b_tmidx.p; B.acc_pos = (double*)h->b_tmapos.p;
      if (tp) {
        double* u = (double*)h->b_tmu.p;
        TRY(copy_in(h, h->b_rot.p, tp->tm_rot + (size_t)step * nrot * 9, nrot * 9 * sizeof(double)));
        TRY(copy_in(h, u, tp->tm_u1 + (size_t)step * NW, NW * sizeof(double)));
        TRY(copy_in(h, u + NW, tp->tm_u2 + (size_t)step * NW, NW * sizeof(double)));
        TRY(copy_in(h, u + 2 * NW, tp->tm_unif + (size_t)step * NW * necp, NW * necp * sizeof(double)));
        B.u1 = u; B.u2 = u + NW; B.unif = u + 2 * NW;
      } else {
        hipLaunchKernelGGL(k_gen_rot, dim3((unsigned)((nrot + 63) / 64)), dim3(64), 0, h->stream, (int)nrot, seed ^ 0x9E3779B97F4A7C15ull,
                           (uint32_t)step, (double*)h->b_rot.p);
        TRY(check_launch(h, "k_gen_rot"));
      }
      B.rot = (const double*)h->b_rot.p;
      HIPCHK(hipMemsetAsync(B.acc, 0, NW * sizeof(int), h->stream));
      hipLaunchKernelGGL(k_tm_count, dim3(gw256.x, (unsigned)N), dim3(256), 0, h->stream, h->S, h->js, B, W);
      TRY(check_launch(h, "k_tm_count"));
      TRY(scan_ints(h, (const int*)B.cnt, B.off, (long)NW, W, d_marks));
      std::vector<long> eoff((size_t)N + 1);  // first candidate of every electron
      TRY(copy_out(h, eoff.data(), d_marks, eoff.size() * sizeof(long)));
      const long tot = eoff[N], tot_up = eoff[h->nup];
      if (tot > 0) {
        TRY(ensure(h, h->b_tpos, (size_t)tot * 3 * sizeof(double)));
        TRY(ensure(h, h->b_twgt, (size_t)tot * sizeof(double)));
        TRY(ensure(h, h->b_tmamp, (size_t)tot * 2 * sizeof(double)));
        TRY(ensure(h, h->b_tmptw, (size_t)tot * sizeof(int)));
        B.pts = (double*)h->b_tpos.p; B.wgt = (double*)h->b_twgt.p; B.amp = (double*)h->b_tmamp.p; B.rat = B.amp + tot;
        B.ptw = (int*)h->b_tmptw.p;
        hipLaunchKernelGGL(k_tm_fill, dim3((unsigned)W, (unsigned)N), dim3(64), 0, h->stream, h->S, h->js, B, W);
        TRY(check_launch(h, "k_tm_fill"));
        const long cnt_s[2] = {tot_up, tot - tot_up}, base_s[2] = {0, tot_up};
        if (h->has_slater)
          for (int s = 0; s < 2; ++s) {
            if (cnt_s[s] == 0) continue;
            TRY(ensure(h, h->b_emo[s], (size_t)cnt_s[s] * nmo_max * sizeof(double)));
            TRY(launch_orb(h, s, plain_points(B.pts + 3 * base_s[s], cnt_s[s]), cnt_s[s], 1, (double*)h->b_emo[s].p));
          }
        // ratios of all candidates against the state before the first T-move: one thread per candidate (k_tm_ratio)
        const bool pre = h->ndet == 1 && !h->has_j3 && !h->cplx && h->tm_pre;
        // U_e of every electron at its current position: from the second step of a call on, the energy evaluation that closed the
        // previous step left exactly that ([N][W], k_kinetic_lw) — the walkers have not moved since
        const double* d_uold = nullptr;
        if (pre && h->has_jastrow) {
          if (lw && step > 0 && h->has_j2 && !h->has_j3) d_uold = (const double*)h->b_kpart.p + (size_t)4 * NW;
          else {
            TRY(ensure(h, h->b_tmuold, (size_t)NW * sizeof(double)));
            hipLaunchKernelGGL(k_tm_uold, dim3(gw256.x, (unsigned)N), dim3(256), 0, h->stream, h->S, h->js, B, W, (double*)h->b_tmuold.p);
            d_uold = (const double*)h->b_tmuold.p;
          }
        }
        if (pre)
          for (int s = 0; s < 2; ++s) {
            if (cnt_s[s] == 0) continue;
            const dim3 g((unsigned)((cnt_s[s] + 255) / 256));
            hipLaunchKernelGGL(k_tm_ratio, g, dim3(256), 0, h->stream, h->S, h->st, h->js, B, s, (int)h->has_slater, (int)h->has_jastrow,
                               (const double*)h->b_emo[s].p, base_s[s], cnt_s[s], W, d_uold);
          }
        const size_t lds_tm = std::max(lds_sm(h), lds_det(h, 1));
        if (h->cplx) hipLaunchKernelGGL(k_tm_walker<true>, dim3((unsigned)W), dim3(64), 2 * lds_tm, h->stream, h->S, h->st, h->js, B, (int)h->has_slater,
                                        (int)h->has_jastrow, (const double*)h->b_emo[0].p, (const double*)h->b_emo[1].p, tot_up, W, 0);
        else hipLaunchKernelGGL(k_tm_walker<false>, dim3((unsigned)W), dim3(64), lds_tm, h->stream, h->S, h->st, h->js, B, (int)h->has_slater,
                                (int)h->has_jastrow, (const double*)h->b_emo[0].p, (const double*)h->b_emo[1].p, tot_up, W, pre ? 1 : 0);
        TRY(check_launch(h, "k_tm_walker"));
        TRY(scan_ints(h, (const int*)B.acc, B.acc_off, (long)NW, W, d_marks));
        hipLaunchKernelGGL(k_tm_gather, dim3((unsigned)((NW + 255) / 256)), dim3(256), 0, h->stream, B, (const double*)h->js.x, N, W);
        TRY(check_launch(h, "k_tm_gather"));
        TRY(copy_out(h, eoff.data(), d_marks, eoff.size() * sizeof(long)));
        const long nacc[2] = {eoff[N], eoff[h->nup]};
        tm_accepted[step] = nacc[0];
        if (h->has_slater) {  // gradient / Laplacian rows of the moved electrons, one launch per spin
          const long na_s[2] = {nacc[1], nacc[0] - nacc[1]}, a0_s[2] = {0, nacc[1]};
          for (int s = 0; s < 2; ++s) {
            if (na_s[s] == 0) continue;
            TRY(ensure(h, h->b_emo[s], (size_t)na_s[s] * 5 * nmo_max * sizeof(double)));
            TRY(launch_orb(h, s, plain_points(B.acc_pos + 3 * a0_s[s], na_s[s]), na_s[s], 5, (double*)h->b_emo[s].p));
            hipLaunchKernelGGL(k_tm_cache, dim3((unsigned)na_s[s]), dim3(64), 0, h->stream, h->S, h->st, (const int*)(B.acc_idx + a0_s[s]),
                               (const double*)h->b_emo[s].p, s, W, lw ? (double*)h->b_rc[s].p : (double*)nullptr, lw ? (const uint8_t*)h->b_sel[s].p : (const uint8_t*)nullptr);
          }
          TRY(check_launch(h, "k_tm_cache"));
        }
      }
    }
    if (tp) {
      TRY(copy_in(h, h->b_gauss.p, tp->gauss + (size_t)step * N * W * 3, (size_t)N * W * 3 * sizeof(double)));
      TRY(copy_in(h, h->b_unif.p, tp->unif + (size_t)step * N * W, (size_t)N * W * sizeof(double)));
      mb.gauss = (const double*)h->b_gauss.p; mb.unif = (const double*)h->b_unif.p;
    }
    if (lw && tmoves) TRY(lw_from_aos(h, false));  // the T-moves worked on the AoS coordinates and inverses
    TRY(sweep_electrons(h, mb, lw, lc));
    hipLaunchKernelGGL(k_sum_reset_int, dim3(1), dim3(1024), 0, h->stream, (int*)h->b_accw.p, W, (int*)h->b_acccnt.p + 2 * step);
    TRY(check_launch(h, "k_propose/k_accept (dmc)"));
    TRY(energy_dev(h, threshold, (tp && necp) ? tp->ecp_rot + (size_t)(step + 1) * nrot * 9 : nullptr,
                   (tp && necp) ? tp->ecp_unif + (size_t)(step + 1) * nrot * W : nullptr, seed, (uint32_t)(step + 1), lw));
    hipLaunchKernelGGL(k_dmc_weights, gw256, dim3(256), 0, h->stream, (const double*)h->b_en.p, eold, eold + W, r2, r2 + W,
                       (double*)h->b_dmcw.p, tstep, branchcut, e_trial, e_est, N, W);
    hipLaunchKernelGGL(k_dmc_averages, dim3(1), dim3(1024), 0, h->stream, (const double*)h->b_en.p, (const double*)h->b_dmcw.p, W,
                       (double*)h->b_dmcout.p + (size_t)step * navg, h->cplx ? 7 : 6);
    TRY(check_launch(h, "k_dmc_weights/k_dmc_averages"));
  }
  if (lw) TRY(lw_to_aos(h, true));
  h->jas_stale = h->has_j2;
  std::vector<int> cnt((size_t)nsteps * 2);
  TRY(copy_in(h, step_avg, h->b_dmcout.p, (size_t)nsteps * navg * sizeof(double)));
  TRY(copy_in(h, weights, h->b_dmcw.p, (size_t)W * sizeof(double)));
  TRY(copy_out(h, cnt.data(), h->b_acccnt.p, cnt.size() * sizeof(int)));
  for (int i = 0; i < nsteps; ++i) {
    step_acc[2 * i] = (double)cnt[2 * i] / ((double)W * N);
    step_acc[2 * i + 1] = (double)tm_accepted[i] / ((double)W * N);
  }
  return 0;
}

extern "C" int pqa_tmove_npoints(pqa_handle_t* h) { return h->tm_P; }

extern "C" int pqa_tmoves(pqa_handle_t* h, int e, double tau, double threshold, const double* rot, const double* unif,
                          double* ratio, double* weight, double* pos) {
  TRY(sync_aos(h));
  HIPCHK(hipSetDevice(h->device));
  if (h->cplx && ratio) FAIL("pqa_tmoves: complex orbitals — pass ratio = NULL (positions and weights only) and take the ratios from pqa_wf_testvalue");
  if (h->W == 0) FAIL("state not initialised (call recompute)");
  if (e < 0 || e >= h->N) FAIL("electron index out of range");
  const long W = h->W;
  const int P = h->tm_P, s = e >= h->nup;
  if (P == 0) return 0;
  if (!rot || !unif) FAIL("pqa_tmoves needs the rotation and mask-uniform tapes");
  h->saved_valid = false;
  const size_t np = (size_t)W * P;
  TRY(ensure(h, h->b_rot, (size_t)h->necp * 9 * sizeof(double)));
  TRY(ensure(h, h->b_eunif, (size_t)h->necp * W * sizeof(double)));
  TRY(ensure(h, h->b_tpos, np * 3 * sizeof(double)));
  TRY(ensure(h, h->b_twgt, np * sizeof(double)));
  TRY(ensure(h, h->b_tlive, np));
  TRY(ensure(h, h->b_trat, np * sizeof(double)));
  TRY(copy_in(h, h->b_rot.p, rot, (size_t)h->necp * 9 * sizeof(double)));
  TRY(copy_in(h, h->b_eunif.p, unif, (size_t)h->necp * W * sizeof(double)));
  hipLaunchKernelGGL(k_tmove_points, dim3((unsigned)W), dim3(64), 0, h->stream, h->S, h->js, e, tau, threshold,
                     (const double*)h->b_rot.p, (const double*)h->b_eunif.p, (const double*)h->d_quad, (const int*)h->d_ptk,
                     (const int*)h->d_pti, P, W, (double*)h->b_tpos.p, (double*)h->b_twgt.p, (uint8_t*)h->b_tlive.p);
  TRY(check_launch(h, "k_tmove_points"));
  if (!ratio) {  // candidate positions and weights only (dead candidates carry weight 0)
    TRY(copy_in(h, weight, h->b_twgt.p, np * sizeof(double)));
    return copy_out(h, pos, h->b_tpos.p, np * 3 * sizeof(double));
  }
  if (h->has_slater) {
    TRY(ensure(h, h->b_motmp, np * std::max(h->nmo[s], 1) * sizeof(double)));
    TRY(launch_orb(h, s, plain_points((const double*)h->b_tpos.p, (long)np), (long)np, 1, (double*)h->b_motmp.p));
  }
  hipLaunchKernelGGL(k_tmove_ratio, dim3((unsigned)W), dim3(64), lds_det(h, 1), h->stream, h->S, h->st, h->js, e, (int)h->has_slater,
                     (int)h->has_jastrow, (const double*)h->b_motmp.p, (const double*)h->b_tpos.p, (const uint8_t*)h->b_tlive.p, P,
                     (double*)h->b_trat.p);
  TRY(check_launch(h, "k_tmove_ratio"));
  TRY(copy_in(h, ratio, h->b_trat.p, np * sizeof(double)));
  TRY(copy_in(h, weight, h->b_twgt.p, np * sizeof(double)));
  return copy_out(h, pos, h->b_tpos.p, np * 3 * sizeof(double));
}

// ---------------------------------------------------------------- measurement
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
  hipLaunchKernelGGL(k_dm_density, g256, dim3(256), 0, h->stream, (const double*)d.row.p, (long)n, nmo2, (double*)d.f.p);
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
      hipLaunchKernelGGL(k_dm_propose, g256, dim3(256), 0, h->stream, (const double*)d.pos.p,
                         gauss ? (const double*)h->b_gauss.p + (size_t)k * n * 3 : (const double*)nullptr, seed, (uint32_t)s, sq, (long)n,
                         (double*)d.newpos.p);
      TRY(launch_orb(h, spin, plain_points((const double*)d.newpos.p, n), n, 1, (double*)h->b_motmp.p));
      hipLaunchKernelGGL(k_dm_accept, dim3((unsigned)n), dim3(64), 0, h->stream, (double*)d.pos.p, (double*)d.row.p, (double*)d.f.p,
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
  hipLaunchKernelGGL(k_obdm_acc, dim3((unsigned)nconf), dim3(256), (size_t)2 * norb * sizeof(double), h->stream,
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
  hipLaunchKernelGGL(k_tbdm_acc, dim3((unsigned)nconf), dim3(256), lds, h->stream,
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
  if (mean) hipLaunchKernelGGL(k_col_means, dim3((unsigned)cols), dim3(256), 0, h->stream, src, h->dm_nconf, cols, scale, (double*)h->dm_tmp.p);
  else hipLaunchKernelGGL(k_scale_copy, dim3((unsigned)((nout + 255) / 256)), dim3(256), 0, h->stream, src, nout, scale, (double*)h->dm_tmp.p);
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
  hipLaunchKernelGGL(k_gram_mfma, dim3((unsigned)((P + 15) / 16), (unsigned)((Q + 15) / 16), (unsigned)nslice), dim3(64), 0, h->stream,
                     (const double*)a.p, (const double*)b.p, (long)n, P, Q, nslice, (double*)part.p);
  hipLaunchKernelGGL(k_gram_reduce, dim3((unsigned)(((long)P * Q + 255) / 256)), dim3(256), 0, h->stream, (const double*)part.p, (long)P * Q,
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
  hipLaunchKernelGGL(k_tile_draws, dim3((unsigned)((NW + 255) / 256)), dim3(256), 0, h->stream, seed, (uint32_t)step, h->N, (long)W,
                     (double*)h->b_gauss.p, (double*)h->b_unif.p);
  TRY(check_launch(h, "k_tile_draws"));
  TRY(copy_in(h, gauss, h->b_gauss.p, NW * 3 * sizeof(double)));
  return copy_out(h, unif, h->b_unif.p, NW * sizeof(double));
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
extern "C" int pqa_last_ecp_points(pqa_handle_t* h, int64_t* npoints) {
  *npoints = h->last_ecp_points;
  return 0;
}
