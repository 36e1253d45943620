// Host-side internals shared by the translation units of libpyqmc_amd.so (pqa_capi.hip, pqa_orb.hip, pqa_sweep.hip,
// pqa_energy.hip, pqa_dmcsteps.hip): the handle, the error macros, buffer helpers and the functions one unit calls in another.
// The device code lives in the kernel headers; every kernel has internal linkage, so a unit only compiles what it launches.
#pragma once
#include <hip/hip_runtime.h>

#include <algorithm>
#include <array>
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
#include "pqa_ecpb.hpp"
#include "pqa_jastrow.hpp"
#include "pqa_lw.hpp"
#include "pqa_slater.hpp"
#include "pqa_tile.hpp"
#include "pqa_res.hpp"
#include "pqa_res8_tab.hpp"
#include "pqa_dm.hpp"
#include "pqa_vmc.hpp"



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
  int pbc_maxcls = PQA_PRE_NCUT;  // most distinct shell cut-offs any atom has (picks the pre-pass instantiation)
  int pbc_nw = 2;  // words per (atom, point) of the sorted image lists k_pbc_prepass writes (4 entries each)
  bool orb_general = false;  // PQA_ORB_GENERAL=1: big open handles evaluate orbitals by k_ao + k_mo_rows instead of the windowed k_orb (A/B, tests)
  // pqa_dmc_continue: the next pqa_dmc_steps call takes the energies its predecessor ended with as its starting energies
  bool dmc_continue = false, dmc_old_valid = false;
  long dmc_old_W = 0;
  bool invert_attr = false;  // k_build_invert's dynamic-LDS limit raised (n > 90)
  bool big = false;         // more than 64 electrons or orbitals of a spin: general orbital path, wave-per-walker kernels (pqa_create)
  bool pbc_high_l = false;  // a periodic cell with g / h shells: orbitals through k_ao<.., 5> + k_mo_rows (pqa_orb_pbc.hip)
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
  std::vector<int> rt_shells;  // [nshell][2] host copy of SysDev::shell_rt (radial tables of the contracted shells)
  double rt_err = 0.0;         // largest table error found at create, relative to sum |c| a^k
  std::vector<int> shell_cost;  // phase-1 cost model of a shell (shell_costs): balances the lane groups of the orbital kernels
  SysDev S{};
  ChunkHost chunks[2];  // [0]: KC=16 (5 components), [1]: KC=32 (value only)
  ChunkTab tab[2]{};
  const unsigned char* out_sel = nullptr;  // two-slot output of the NEXT orbital launch (ChunkTab::out_sel; set by launch_orb)
  long out_slot_stride = 0;
  int orb_col0 = 0;  // first orbital column of the NEXT k_orb launch (ChunkTab::col0; set by launch_orb for handles with > 64 orbitals)
  double* d_mo[2] = {nullptr, nullptr};       // [nao][nmo]
  double* d_cpad[2][2] = {{nullptr, nullptr}, {nullptr, nullptr}};  // [tab][spin]
  double *d_acoeff = nullptr, *d_bcoeff = nullptr, *d_detcoeff = nullptr, *d_quad = nullptr, *d_quadw = nullptr;
  int *d_ecp_naip = nullptr, *d_ecp_qoff = nullptr;  // per-atom quadrature rule (pqa_set_ecp_naip)
  int ecp_naip = 0;                                  // 0: the reference's default, 6 or 12 by channel count
  std::vector<int> ecp_nch;                          // channels (incl. local) of every ECP atom
  // batched ECP integrator (pqa_set_ecp_batched, pqa_ecpb.hpp): per-atom point counts, table size, selections, slots per electron
  int ecpb_on = 0, ecpb_npoints = 0, ecpb_nsd = 0, ecpb_nsr = 0, ecpb_nsel = 0;
  int *d_ecpb_naip = nullptr, *d_ecpb_qoff = nullptr, *d_ecpb_pstart = nullptr;
  double *d_aq = nullptr, *d_bq = nullptr;  // merged Pade numerators (jas_merge_tables); jas_merge: PQA_JAS_MERGE=0 keeps the function-by-function route (A/B)
  int jas_merge = 1;
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
  long step_pre_max = 8192;  // PQA_STEP_PRE_MAX: largest shard (walkers) that runs k_step_pre (8192: 5.90 -> 5.57 ms per (H2O)8 step since the quartet commit; 16384 loses)
  int step_gw = 0;       // PQA_STEP_GW: thread groups per walker of k_step_pre for shards <= 4096 walkers (0 automatic: 64; 16 / 32 / 64)
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
  int orb_kc5 = 0;  // 0: by launch size (launch_orb_pbc_any); PQA_ORB_KC5 16|32 pins it
  int orb_kc1 = 16;  // AO rows per chunk of the periodic value-only launch (PQA_ORB_KC1 16|32)
  struct TpTune { float ms[2] = {1e30f, 1e30f}; int n[2] = {0, 0}; int choice = 0; };  // periodic k_orb: [0] 32-point, [1] 64-point tiles
  TpTune tp_tune[2][48];  // per chunk table (5 / 1 components) and log2 bucket of the point count
  WideTab wide[2]{};  // lane-group shell lists of the whole-K small-launch kernel (k_orb_wide), per chunk table (64 groups; periodic: 32)
  int orb_wide = -1;  // PQA_ORB_WIDE: -1 automatic (5-component launches of <= orb_wide_max points), 0 never, 1 whenever the tile fits LDS
  long orb_wide_max = 8192;  // PQA_ORB_WIDE_MAX
  std::vector<const void*> wide_attr;  // kernels whose dynamic-LDS limit has been raised
  int orb_ws = -1;  // -1 automatic; 1 wave-specialised orbital kernel; 0 phase-alternating k_orb (PQA_ORB_WS)
  int orb_notab = 0;  // PQA_ORB_NOTAB=1: basis tables from global memory (A/B)
  // pipelined half-ensembles of the lane-per-walker sweep (pqa_sweep.hip): mode (PQA_SPLIT), smallest shard that is cut
  // (PQA_SPLIT_MIN), CUs of the orbital stream in mode 3 (PQA_SPLIT_CUS, 0 = no masks)
  // OFF by default: two free-running half-ensembles (PQA_SPLIT=1) give +3.3-3.7 % at 65536 walkers in same-box A/B runs (bit-identical;
  // -8 % at 32768), but the kernels of the two halves then share the chip and every launch takes about twice as long for the same
  // work — the per-kernel roofline bench.py reports from launch durations stops meaning what it says (0.40 -> 0.23 for k_orb)
  int split_mode = 0, split_cus = 0, cu_count = 256;
  long split_min = 65536;
  hipStream_t pipe_stream[2] = {nullptr, nullptr};
  // Jastrow sums of a move summed ahead on a side stream next to the orbital kernel (k_jas_pre, pqa_lw.hpp): PQA_JPRE=1 (or -1:
  // shards of at least jpre_min walkers, PQA_JPRE_MIN), one side stream per half-ensemble, partials [2 halves of a move][G][4][W].
  // OFF by default — measured: k_step_lw 117 -> 81 us per move, but k_orb next to k_jas_pre 122 -> 196 us (25.9 -> 27.0 ms per step
  // at 65536 walkers; 2x2x2 periodic cell +2 %): the fp64 pipe the two share is the step's bottleneck, not idle
  int jpre = 0;
  long jpre_min = 32768;
  hipStream_t jas_stream[2] = {nullptr, nullptr};
  DevBuf b_jpre;
  std::vector<hipEvent_t> pipe_events;
  size_t pipe_next = 0;
  // resident sweep (pqa_res.hpp / pqa_res.hip): the whole electron sweep of 16 walkers in one block, one launch per sweep.
  // PQA_RES: -1 automatic (shards of res_min .. res_max walkers: PQA_RES_MIN / PQA_RES_MAX), 0 never, 1 whenever the system is in scope
  int res_mode = -1;
  long res_min = 1, res_max = 1L << 40;
  bool res_ready = false, res_ok = false;
  // dense mode: the tile holds the AOs in their own order (rows padded to x4 only) with its own coefficient copy d_cres[s] [rows4][ldc] —
  // for bases whose chunk-padded rows do not fit one LDS tile (the 2x2x2 diamond cell: 208 AOs, 224 padded rows)
  bool res_dense = false;
  int res_rows4 = 0;
  double* d_cres[2] = {nullptr, nullptr};
  ResTab res_tab{};
  size_t res_lds = 0;
  int res_lmax = 0;
  int pbc_mincls = 0;
  bool pbc_lists_ok = false;
  // second generation of the resident sweep for open-boundary real handles (pqa_res8.hpp / pqa_res8.hip): 8 walkers per 256-thread block, two
  // blocks per CU, wave-uniform AO phase.  PQA_R8: -1 automatic (r8_eligible), 0 never (k_sweep_res / the launches), 1 whenever in scope
  int r8_mode = -1;
  bool r8_ready = false, r8_ok = false;
  R8Tab r8_tab{};
  size_t r8_lds = 0;
  double r8_util = 0.0;
  bool r8_xaos_next = false;  // the next k_sweep_r8 launch also writes the walker-major coordinates (js.x)
  bool jsx_current = false;   // ... and did: energy_dev skips its transpose of the coordinate planes
  int res_pbc = 1;  // PQA_RES_PBC=0: periodic handles keep the launch-per-move sweep (A/B)
  int res_cx = 1;   // PQA_RES_CX=0: complex determinants keep the launch-per-move sweep (A/B)
  // wave-per-walker sweep in one launch (pqa_ww.hpp; PQA_WW): -1 by shard size (one wave per walker up to ww_max walkers), 0 off, 1 always,
  // 3 always with three waves per walker (measured slower).  50-determinant water molecule, VMC step with energy, launches -> one
  // launch: 0.722 -> 0.663 ms at 1 024 walkers, 0.884 -> 0.801 at 2 048, 1.428 -> 1.382 at 4 096, 2.25 -> 2.38 at 8 192
  // ECP point totals left on the device (pqa_energy.hip: small shards on the k_ecp_accum path; PQA_ECP_DEFER=0 reads them every time)
  const double* en_d_ecp = nullptr;  // energy_dev: the ECP row(s) of its last evaluation (nullptr: no ECP)
  long* pin_tot = nullptr;  // pinned host words the scan kernels write the ECP point totals to (device-visible: hipHostMallocMapped)
  int ecp_defer = 1;
  int en_overlap = 1;  // kinetic / Coulomb pass of small wave-per-walker shards on a side stream beside the ECP passes (PQA_EN_OVERLAP=0: in line)
  // the NEXT step's sweep draws (k_tile_draws) generated beside the energy pass of the current step: second tape set, its stream and events
  DevBuf b_gauss_b, b_unif_b;
  hipStream_t draw_stream = nullptr;
  hipEvent_t draw_ev[2] = {nullptr, nullptr};
  bool draw_ahead_valid = false, draws_on_device = false, draws_ahead_on = true;  // (PQA_DRAWS_AHEAD=0: A/B)  // draws_on_device: the last sweep took its draws from k_tile_draws
  uint32_t draw_ahead_step = 0;
  uint64_t draw_ahead_seed = 0;
  long draw_ahead_W = 0;
  hipStream_t en_stream = nullptr;
  hipEvent_t en_ev[3] = {nullptr, nullptr, nullptr};
  bool ecp_hint_valid = false;
  long ecp_hint[2] = {0, 0}, ecp_evals = 0;
  const long* last_ecp_dev[2] = {nullptr, nullptr};
  long orb_p_hint = 0;  // launch_orb: points the next launch is expected to work on when its P is an upper bound (0: P)
  int ww_mode = -1;
  long ww_max = 4096;
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


static inline int ensure(pqa_handle* h, DevBuf& b, size_t bytes) {
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

static inline int copy_in(pqa_handle* h, void* dst, const void* src, size_t bytes) {
  if (bytes == 0) return 0;
  HIPCHK(hipMemcpyAsync(dst, src, bytes, hipMemcpyDefault, h->stream));
  return 0;
}
static inline int copy_out(pqa_handle* h, void* dst, const void* src, size_t bytes) {
  if (bytes) HIPCHK(hipMemcpyAsync(dst, src, bytes, hipMemcpyDefault, h->stream));
  HIPCHK(hipStreamSynchronize(h->stream));
  return 0;
}
static inline int check_launch(pqa_handle* h, const char* what) {
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) {
    h->err = std::string(what) + " launch failed: " + hipGetErrorString(e);
    return -1;
  }
  return 0;
}

static inline size_t lds_j3(const pqa_handle* h) {  // bytes needed by kernels that call jas_eval with the three-body term
  return h->has_j3 ? ((size_t)h->S.j3_off + (size_t)h->natom * (3 + 6 * h->na3 * h->nb3)) * sizeof(double) : 0;
}
static inline size_t lds_sm(const pqa_handle* h) {
  const size_t n = std::max(h->nup, h->ndn);
  if (n > PQA_MAXN_FAST) return std::max((3 * n + 64) * sizeof(double), lds_j3(h));  // sm_update_wave works on the inverse in place there
  return std::max((n * (n + 1) + 2 * n + 64 + n) * sizeof(double), lds_j3(h));
}
static inline size_t lds_det(const pqa_handle* h, int ncomp) {
  return std::max((size_t)std::max(h->ndet_s[0], h->ndet_s[1]) * ncomp * sizeof(double), lds_j3(h));
}

struct LwCtx {
  int Gm = 1, KB = 1, nmax = 1;
};

// ---- functions defined in one unit and called from others
// pqa_orb.hip: out[p][ncomp][nmo_spin]; out_sel / slot_stride: two-slot output (ChunkTab::out_sel), else plain rows
int launch_orb(pqa_handle* h, int spin, PointAddr pa, long P, int ncomp, double* out, const unsigned char* out_sel = nullptr, long slot_stride = 0);
PointAddr plain_points(const double* base, long P);
// pqa_orb_pbc.hip: AO planes out[ncomp][P][nao] at arbitrary points, thread per point (test entry, parameter gradients, the general
// periodic path); ncomp 1, 4 or 5
int launch_ao(pqa_handle* h, PointAddr pa, long P, int ncomp, double* out);
// pqa_sweep.hip
void transpose(pqa_handle* h, const double* in, double* out, long R, long C);  // in [R][C] -> out [C][R]
LwState lw_state(pqa_handle* h);
int lw_from_aos(pqa_handle* h, bool with_cache = true);
int lw_to_aos(pqa_handle* h, bool with_cache);
int sync_aos(pqa_handle* h);
int lw_setup(pqa_handle* h, bool lw, LwCtx& c);
int sweep_electrons(pqa_handle* h, const MoveBuf& mb, bool lw, const LwCtx& lc);
void launch_step_real(pqa_handle* h, const LwState& L, const MoveBuf& mb, const StepArgs& a, int rowlen);
void launch_jas_pre(pqa_handle* h, const LwState& L, const MoveBuf& mb, const StepArgs& a, double* jnew, double* jold);
void launch_flush_real(pqa_handle* h, const LwState& L, int s, long W, long w0, long w1, int j_lo, int j_hi, int nq, int rowlen, int n_s);
// pqa_sweep_cx.hip
void launch_step_cx(pqa_handle* h, const LwState& L, const MoveBuf& mb, const StepArgs& a, int rowlen);
void launch_flush_cx(pqa_handle* h, const LwState& L, int s, long W, long w0, long w1, int j_lo, int j_hi, int nq, int rowlen, int n_s);
// pqa_res.hip
bool res_eligible(pqa_handle* h, long W);
int sweep_res(pqa_handle* h, const MoveBuf& mb);
int res_refresh_coeff(pqa_handle* h, int s, const double* mo_host);  // (pqa_res.hip: dense coefficient copy follows set_mo)
// pqa_res8.hip
static inline int res_rows_alloc(int rows4) { return rows4 + 96; }  // rows of the dense coefficient copies d_cres: zero beyond the basis (k_sweep_r8 contracts six k-steps per trip in every wave)
bool r8_eligible(pqa_handle* h, long W);
int sweep_r8(pqa_handle* h, const MoveBuf& mb);
// pqa_sweep_ww.hip
bool ww_eligible(pqa_handle* h, long W);
int sweep_ww(pqa_handle* h, const MoveBuf& mb);
// pqa_tile.hip
bool tile_eligible(const pqa_handle* h);
int sweep_tile(pqa_handle* h, const MoveBuf& mb_in);
// pqa_energy.hip
// assemble = false: the rows of b_en are left to the caller (k_energy_finish, from b_kc and en_d_ecp)
int energy_dev(pqa_handle* h, double threshold, const double* rot, const double* unif, uint64_t seed, uint32_t step,
               bool soa_current = false, bool aos_T_needed = true, bool assemble = true);
// pqa_dmcsteps.hip
int scan_ints(pqa_handle* h, const int* c, long* o, long n, long Wm, long* marks);
