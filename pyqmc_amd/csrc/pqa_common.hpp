// pyqmc_amd device-side common definitions (gfx950 only; wave = 64 lanes).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#define PQA_WAVE 64
#define PQA_MAXBAS 16     // max two-body Jastrow basis functions per kind (hot kernels loop to na / nb; only the protocol kernels' register arrays have this length)
#ifndef PQA_MAXBAS3
#define PQA_MAXBAS3 8     // max three-body basis functions per kind (fully unrolled register arrays in jas3_eval)
#endif
#define PQA_JQ 24         // doubles per merged-numerator record (N1[4], N2[7], N3[10], padding)
#define PQA_MAXN 128      // max electrons / orbitals per spin (real orbitals; up to 64 on every fast path, above: the two-slot wave kernels)
#define PQA_MAXN_FAST 64  // one lane per column: LDS-staged determinant tile, lane-per-walker planes, four 16-column MFMA tiles
#define PQA_MAXCHAN 6     // ECP channels per atom incl. local: s, p, d, f, g non-local channels — the reference's Legendre table ends at l = 4 (eval_ecp.py:203-225)
#define PQA_MAXAIP 12

typedef double d4 __attribute__((ext_vector_type(4)));

// Periodic-cell tables (device memory, read through the scalar cache).
struct PbcDev {
  double lat[9], linv[9];  // rows = lattice vectors; linv = inverse (frac = d . linv)
  // general cells: the Voronoi-relevant lattice vectors (one of each +- pair, at most 7 in three dimensions) and |v|^2 / 2
  int nvor;
  double vor[7][3], vorh[7];
  // 1: every Jastrow cut-off is at most the inradius of the cell-centred parallelepiped {frac in [-1/2, 1/2)^3}: a Jastrow pair then
  // never needs the reduction above — inside the cut-off the folded vector IS the minimal image (see min_image_j)
  int jas_fold;
  // periodic Gamma-point orbitals (numba/pbcgto.py:99-653): AO = sum over the cell translations Ls[j], j < num_Ls[atom],
  // skipping images with r^2 > atom_cut[atom] or r^2 > shell_cut[shell]
  const double* Ls;
  const int* num_Ls;
  const double* atom_cut;
  const double* shell_cut;
  // distinct shell cut-offs of every atom, ascending: cls_cut[atom][PQA_MAXCLS], ncls[atom] (0: more than PQA_MAXCLS, no lists)
  const double* cls_cut;
  const int* ncls;
  // reference image-membership rule (see include/pyqmc_amd.h): member == nullptr -> every image inside the cut-offs
  const unsigned char* member;
  // membership of candidate image j for every (atom class, membership base b): masks[class][b0 + E][b1 + E][b2 + E][2] over a
  // (side + 2 E)^3 grid of bases (create: member_masks); nullptr: no table, test candidate by candidate
  const unsigned long long* memb_mask;
  int memb_E;
  // candidates that can lie inside atom_cut of SOME point of a sub-cell of the cell-centred parallelepiped the folded
  // displacement point - atom lives in: near_mask[atom][g0][g1][g2][2] over a near_G^3 grid of fractional coordinates
  // (create: near_masks; conservative, so the exact tests downstream decide).  nullptr: every candidate is looked at
  const unsigned long long* near_mask;
  int near_G;
  const int* member_class;
  const int* img_n;
  const int* atom_n;
  int member_M;
  int supercell[9];  // lattice = supercell . lattice_prim (integers)
  double lprim_inv[9];
  // twisted boundary conditions: every image carries exp(i k_t . L); ktl[a] = k_t . lattice_a, img_phase[j] = (cos, sin)(k_t . Ls[j])
  int twist;
  double ktl[3];
  const double* img_phase;
};

// Device view of the system tables (all pointers are device memory).
struct SysDev {
  int natom, nup, ndn, nelec;
  const double* atom_xyz;
  const double* atom_charge;
  int nshell, nprim, nao;
  const int* shell_atom;
  const int* shell_l;
  const int* shell_prim_off;
  const int* shell_ao_off;
  const double* prim_exp;
  const double* prim_coef;
  // Tabulated radial functions of contracted shells (round 6; open systems, the MFMA orbital kernels and the resident sweep): for a shell with
  // shell_rt[2 sh] >= 0 the three radial sums F_k(x) = sum_p a_p^k c_p exp(-a_p x), k = 0, 1, 2, x = r^2 (numba/gto.py:89-254: R, dR, lap R) are
  // piecewise degree-9 polynomials at rtab + shell_rt[2 sh], shell_rt[2 sh + 1] intervals of [interval][3][10] doubles (radial_tab, pqa_ao.hpp;
  // host: build_radial_tables, pqa_capi.hip: |error| <= 1e-15 sum_p |c_p| a_p^k).  -1: the primitives are summed (exp per primitive).
  const double* rtab;
  const int* shell_rt;
  int nmo[2];
  const double* mo[2];  // [ao][nmo]
  int ndet, ndet_s[2];
  const double* det_coeff;
  const int* det_occ[2];  // [ndet_s][n_s]
  int occ_ident[2];       // the first determinant of the spin occupies orbitals 0..n-1 in order (rows can be read with wide loads)
  const int* det_map;     // [2][ndet]
  int na, nb;
  int a_kind[PQA_MAXBAS];
  double a_param[PQA_MAXBAS];
  double a_aux[PQA_MAXBAS];  // 1/(3+gamma) for cusp functions
  int b_kind[PQA_MAXBAS];
  double b_param[PQA_MAXBAS];
  double b_aux[PQA_MAXBAS];
  double rcut_a, rcut_b;
  const double* acoeff;  // [natom][na][2]
  const double* bcoeff;  // [nb][3]
  // merged Pade functions (pqa_jastrow.hpp: pade_merged): jq_on -> every non-empty basis consists of an optional cusp function at index 0
  // followed by 2..4 PolyPade functions, whose coefficient-weighted sums are evaluated as ONE rational function of p per pair:
  // denominators a_D / b_D (ascending powers of p, zero padded), numerators per coefficient set in aq [natom][2][PQA_JQ] and
  // bq [3][PQA_JQ] (N1 at 0, N2 at 4, N3 at 11; host: jas_merge_tables).  0: function by function (rad_fn)
  int jq_on, jq_a, jq_b;  // jq_on: merged route available; jq_a / jq_b: Pade functions of the basis (0: the basis is empty)
  double a_D[5], b_D[5];
  const double* aq;
  const double* bq;
  // three-body Jastrow (three_body_jastrow.py:19-63): own a/b bases, C = (c + c^T_kl)/2 as [natom][na3][na3][nb3][3]
  int na3, nb3;
  int a3_kind[PQA_MAXBAS3];
  double a3_param[PQA_MAXBAS3];
  double a3_aux[PQA_MAXBAS3];
  int b3_kind[PQA_MAXBAS3];
  double b3_param[PQA_MAXBAS3];
  double b3_aux[PQA_MAXBAS3];
  double rcut_a3, rcut_b3;
  const double* c3;
  int j3_off;  // offset (in doubles) of the three-body scratch inside a kernel's dynamic LDS
  // periodic boundary conditions: 0 open, 1 fold fractional coordinates (orthogonal lattice vectors,
  // distance.py:143-159), 2 fold + argmin over the 27 neighbouring cells (distance.py:129-141).  Everything else a
  // periodic system needs sits behind ONE pointer, so that the kernel-argument block (and with it the scalar-register
  // footprint of every open-boundary kernel) does not grow with it.
  int pbc;
  int nL;  // > 0: periodic orbital tables present (pb->Ls ...)
  const struct PbcDev* pb;
  int necp;
  const int* ecp_atom;
  const int* ecp_chan_off;
  const int* ecp_term_off;
  const int* ecp_term_n;
  const double* ecp_term_exp;
  const double* ecp_term_coef;
  const double* ecp_rc2;  // [necp] r^2 beyond which every term of the atom's ECP is below 1e-22 in magnitude (create: ecp_ranges)
  // quadrature rule of every ECP atom (eval_ecp.py:228-252 get_P_l, :278-336 the Mitas-Shirley-Ceperley grids): naip points starting at
  // row ecp_qoff[k] of the direction / weight tables (EcpBuf::quad, quadw).  pqa_set_ecp_naip; default 6 (<= 2 channels) or 12.
  const int* ecp_naip;
  const int* ecp_qoff;
  int ecp_naip_max;
};

// ---------------------------------------------------------------- minimal image
// Displacement -> nearest periodic image (MinimalImageDistance, distance.py:83-159).  Folding the fractional
// coordinates into [-1/2, 1/2) is the reference's diagonal/orthogonal rule and exact there.  For a general cell the
// reference searches the 27 neighbours of the difference of two in-cell points; here the folded vector is REDUCED into the
// Wigner-Seitz cell instead: d is a minimal image iff |d . v| <= |v|^2 / 2 for every Voronoi-relevant lattice vector v (6
// pairs for the fcc-type supercells, 7 at most), and subtracting / adding a violated v strictly shortens d — one or two
// sweeps over the host-made list (create: voronoi_vectors) and one that changes nothing, ~150 instructions where the
// 27-candidate search took ~500 in every periodic Jastrow / ECP / Ewald pair.  Same distances (both are exact minima; at a
// tie on the cell boundary the representative may differ).
template <bool JAS>
__device__ __forceinline__ void min_image_impl(const SysDev& S, double& dx, double& dy, double& dz) {
  if (S.pbc == 0) return;
  double f0 = dx * S.pb->linv[0] + dy * S.pb->linv[3] + dz * S.pb->linv[6];
  double f1 = dx * S.pb->linv[1] + dy * S.pb->linv[4] + dz * S.pb->linv[7];
  double f2 = dx * S.pb->linv[2] + dy * S.pb->linv[5] + dz * S.pb->linv[8];
  f0 -= floor(f0 + 0.5); f1 -= floor(f1 + 0.5); f2 -= floor(f2 + 0.5);
  dx = f0 * S.pb->lat[0] + f1 * S.pb->lat[3] + f2 * S.pb->lat[6];
  dy = f0 * S.pb->lat[1] + f1 * S.pb->lat[4] + f2 * S.pb->lat[7];
  dz = f0 * S.pb->lat[2] + f1 * S.pb->lat[5] + f2 * S.pb->lat[8];
  if (S.pbc == 2 && !(JAS && S.pb->jas_fold)) {
    // table once into (scalar) registers: unused entries are zero vectors with h = 1, which never trigger
    double vx[7], vy[7], vz[7], hh[7];
#pragma unroll
    for (int q = 0; q < 7; ++q) { vx[q] = S.pb->vor[q][0]; vy[q] = S.pb->vor[q][1]; vz[q] = S.pb->vor[q][2]; hh[q] = S.pb->vorh[q]; }
    bool changed = false;
#pragma unroll
    for (int sweep = 0; sweep < 2; ++sweep) {  // two straight-line sweeps settle all but ~0.1 % of the vectors ...
      changed = false;
#pragma unroll
      for (int q = 0; q < 7; ++q) {
        const double pr = dx * vx[q] + dy * vy[q] + dz * vz[q];
        const double sg = pr > hh[q] ? 1.0 : (pr < -hh[q] ? -1.0 : 0.0);
        dx -= sg * vx[q]; dy -= sg * vy[q]; dz -= sg * vz[q];
        changed = changed || sg != 0.0;
      }
    }
    for (int sweep = 0; sweep < 6 && __any(changed); ++sweep) {  // ... the rest until a sweep changes nothing
      changed = false;
#pragma unroll
      for (int q = 0; q < 7; ++q) {
        const double pr = dx * vx[q] + dy * vy[q] + dz * vz[q];
        const double sg = pr > hh[q] ? 1.0 : (pr < -hh[q] ? -1.0 : 0.0);
        dx -= sg * vx[q]; dy -= sg * vy[q]; dz -= sg * vz[q];
        changed = changed || sg != 0.0;
      }
    }
  }
}

__device__ __forceinline__ void min_image(const SysDev& S, double& dx, double& dy, double& dz) { min_image_impl<false>(S, dx, dy, dz); }
// For the pair functions of the Jastrow factor, which vanish from their cut-off r_c on.  Where r_c <= rho, the inradius of the
// parallelepiped the fold maps into (pyqmc's periodic default is r_c = rho: wftools.py generate_jastrow), the Voronoi
// reduction is never needed: a minimal image shorter than rho lies inside the parallelepiped, so the fold — the unique
// representative in there — is that image, unchanged by the reduction (bitwise the same vector); and a folded vector of
// length >= r_c means the minimal image is >= r_c as well (were it shorter it would be the folded vector), so the pair is
// skipped either way.  ~150 of the ~250 instructions of every periodic Jastrow pair in the general (fcc-type) cells.
__device__ __forceinline__ void min_image_j(const SysDev& S, double& dx, double& dy, double& dz) { min_image_impl<true>(S, dx, dy, dz); }

// Position -> inside the cell (enforce_pbc, pbc/pbc.py:37-48: fractional coordinates split by divmod(., 1)); dw receives
// the integer wrap that was removed.
__device__ __forceinline__ void fold_cell(const SysDev& S, double& x, double& y, double& z, int* dw = nullptr) {
  if (S.pbc == 0) {
    if (dw) dw[0] = dw[1] = dw[2] = 0;
    return;
  }
  double f0 = x * S.pb->linv[0] + y * S.pb->linv[3] + z * S.pb->linv[6];
  double f1 = x * S.pb->linv[1] + y * S.pb->linv[4] + z * S.pb->linv[7];
  double f2 = x * S.pb->linv[2] + y * S.pb->linv[5] + z * S.pb->linv[8];
  const double w0 = floor(f0), w1 = floor(f1), w2 = floor(f2);
  f0 -= w0; f1 -= w1; f2 -= w2;
  x = f0 * S.pb->lat[0] + f1 * S.pb->lat[3] + f2 * S.pb->lat[6];
  y = f0 * S.pb->lat[1] + f1 * S.pb->lat[4] + f2 * S.pb->lat[7];
  z = f0 * S.pb->lat[2] + f1 * S.pb->lat[5] + f2 * S.pb->lat[8];
  if (dw) { dw[0] = (int)w0; dw[1] = (int)w1; dw[2] = (int)w2; }
}

__device__ __forceinline__ double mi_norm(const SysDev& S, double dx, double dy, double dz) {
  min_image(S, dx, dy, dz);
  return sqrt(dx * dx + dy * dy + dz * dz);
}

// ---------------------------------------------------------------- wave-level reductions
// Sum over the 64 lanes, result broadcast to every lane.  Uses DPP row shifts + row broadcasts (VALU
// only, ~20 instructions) instead of __shfl_xor, which lowers to 12 dependent ds_bpermute_b32 round
// trips through the LDS crossbar per reduction.  Fixed association order -> deterministic.
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ double dpp_add(double v) {
  const int lo = __builtin_amdgcn_update_dpp(0, __double2loint(v), CTRL, ROW_MASK, 0xf, true);
  const int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(v), CTRL, ROW_MASK, 0xf, true);
  return v + __hiloint2double(hi, lo);  // lanes whose source is out of range / row-masked add +0.0
}
// v of the lane quad_perm CTRL names, inside each quad of four lanes
template <int CTRL>
__device__ __forceinline__ double quad_dpp(double v) {
  const int lo = __builtin_amdgcn_update_dpp(0, __double2loint(v), CTRL, 0xf, 0xf, false);
  const int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(v), CTRL, 0xf, 0xf, false);
  return __hiloint2double(hi, lo);
}
// Synchronisation inside the wave-per-walker device functions (slater_ratios, sm_update_wave, jas3_eval: one wave works on one walker
// through its own LDS scratch).  In blocks whose waves all run the same function it is the block barrier; pqa_sweep_ww.hip — three
// waves of a block running DIFFERENT functions of one move — defines it as a wave-level fence before including the headers.
#ifndef PQA_WSYNC
#define PQA_WSYNC() __syncthreads()
#endif
// The headers whose device functions contain PQA_WSYNC (pqa_slater.hpp, pqa_jastrow.hpp) put them into an inline namespace named after the
// flavour, so that the two bodies of e.g. sm_update_wave are two different entities (no one-definition-rule hazard if the units were ever
// linked with relocatable device code): a unit that redefines PQA_WSYNC also defines PQA_SYNC_NS.
#ifndef PQA_SYNC_NS
#define PQA_SYNC_NS pqa_sync_block
#endif
__device__ __forceinline__ double wave_sum(double v) {
  v = dpp_add<0x111, 0xf>(v);  // row_shr:1   inclusive scan inside each row of 16 lanes
  v = dpp_add<0x112, 0xf>(v);  // row_shr:2
  v = dpp_add<0x114, 0xf>(v);  // row_shr:4
  v = dpp_add<0x118, 0xf>(v);  // row_shr:8   lane 15 of each row = row total
  v = dpp_add<0x142, 0xa>(v);  // row_bcast:15 into rows 1 and 3
  v = dpp_add<0x143, 0xc>(v);  // row_bcast:31 into rows 2 and 3 -> lane 63 = wave total
  const int lo = __builtin_amdgcn_readlane(__double2loint(v), 63);
  const int hi = __builtin_amdgcn_readlane(__double2hiint(v), 63);
  return __hiloint2double(hi, lo);
}

__device__ __forceinline__ double wave_max(double v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v = fmax(v, __shfl_xor(v, off, PQA_WAVE));
  return v;
}

__device__ __forceinline__ double wave_bcast(double v, int lane) { return __shfl(v, lane, PQA_WAVE); }

// ---------------------------------------------------------------- Philox4x32-10 counter RNG
struct Philox {
  uint32_t c[4];
};
__device__ __forceinline__ Philox philox(uint64_t seed, uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3) {
  uint32_t k0 = (uint32_t)seed, k1 = (uint32_t)(seed >> 32);
  uint32_t x0 = c0, x1 = c1, x2 = c2, x3 = c3;
#pragma unroll
  for (int r = 0; r < 10; ++r) {
    uint64_t p0 = (uint64_t)0xD2511F53u * x0;
    uint64_t p1 = (uint64_t)0xCD9E8D57u * x2;
    uint32_t y0 = (uint32_t)(p1 >> 32) ^ x1 ^ k0;
    uint32_t y1 = (uint32_t)p1;
    uint32_t y2 = (uint32_t)(p0 >> 32) ^ x3 ^ k1;
    uint32_t y3 = (uint32_t)p0;
    x0 = y0; x1 = y1; x2 = y2; x3 = y3;
    k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
  }
  Philox o; o.c[0] = x0; o.c[1] = x1; o.c[2] = x2; o.c[3] = x3;
  return o;
}
// uniform in (0,1) with 53 random bits
__device__ __forceinline__ double u01(uint32_t hi, uint32_t lo) {
  uint64_t b = (((uint64_t)hi << 32) | lo) >> 11;
  return ((double)b + 0.5) * (1.0 / 9007199254740992.0);
}
// two standard normals from one Philox block (Box-Muller)
__device__ __forceinline__ void normal2(const Philox& p, double& z0, double& z1) {
  double u1 = u01(p.c[0], p.c[1]), u2 = u01(p.c[2], p.c[3]);
  double r = sqrt(-2.0 * log(u1));
  double s, c;
  sincospi(2.0 * u2, &s, &c);
  z0 = r * c; z1 = r * s;
}

// RNG stream ids (third counter word) so the draws of different kernels never collide
#define PQA_STREAM_GAUSS_A 1u
#define PQA_STREAM_GAUSS_B 2u
#define PQA_STREAM_ACCEPT 3u
#define PQA_STREAM_ECPMASK 4u
#define PQA_STREAM_ECPROT 5u
