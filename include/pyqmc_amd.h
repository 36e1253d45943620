/* pyqmc_amd — C ABI of the MI355X (gfx950) walker-batched trial-wave-function evaluator.
 *
 * The reference (WagnerGroup/pyqmc 0.8.0) has NO FFI: its boundary is the duck-typed
 * Python wave-function protocol (doc/source/wavefunction.rst:5-40).  Each entry point
 * below names the reference method it stands in for; pyqmc_amd/wf.py binds them with
 * ctypes behind classes with the reference's names (Slater, JastrowSpin, MultiplyWF),
 * and INTEGRATION.md shows the stub a pyqmc maintainer would add.
 *
 * Conventions
 *  - return 0 = ok, <0 = error (text via pqa_last_error).
 *  - every buffer is caller-owned, C-contiguous fp64 unless typed otherwise, and may be a
 *    host pointer OR a device pointer (copies use hipMemcpyDefault).  No pointer is
 *    retained after the call returns.
 *  - one handle = one walker shard on one device, one HIP stream, not re-entrant
 *    (same as the reference objects, which hold mutable per-walker state).
 *  - electrons are ordered all spin-up then all spin-down (slater.py:236-239).
 */
#ifndef PYQMC_AMD_H
#define PYQMC_AMD_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct pqa_handle pqa_handle_t;

/* Flat description of a Slater-Jastrow trial wave function (all pointers: host memory). */
typedef struct {
  /* geometry */
  int32_t natom;
  int32_t nelec_up, nelec_dn;
  const double* atom_xyz;    /* natom*3, bohr */
  const double* atom_charge; /* natom, valence charges (energy.py:40-45) */
  /* contracted GTO shells in AO order (numba/gto.py:435-470); coefficients already
     normalised (gto.py:375-405) */
  int32_t nshell, nprim, nao;
  const int32_t* shell_atom;     /* nshell */
  const int32_t* shell_l;        /* nshell, l <= 5 (twisted cells: l <= 3) */
  const int32_t* shell_prim_off; /* nshell+1 */
  const int32_t* shell_ao_off;   /* nshell */
  const double* prim_exp;        /* nprim */
  const double* prim_coef;       /* nprim */
  /* molecular orbitals, row-major [ao][mo] (orbitals.py:56-60), truncated to used columns */
  int32_t nmo_up, nmo_dn;
  const double* mo_up;
  const double* mo_dn;
  /* determinant expansion packed like determinant_tools.create_packed_objects (:39-71) */
  int32_t ndet, ndet_up, ndet_dn;
  const double* det_coeff;   /* ndet */
  const int32_t* det_occ_up; /* ndet_up*nelec_up */
  const int32_t* det_occ_dn; /* ndet_dn*nelec_dn */
  const int32_t* det_map;    /* 2*ndet */
  /* two-body Jastrow (jastrowspin.py:31-54); basis kinds: 0 = PolyPade(beta), 1 = CutoffCusp(gamma)
     (func3d.py:52-210) */
  int32_t na, nb; /* 0/0 = no Jastrow factor */
  const int32_t* a_kind;
  const double* a_param;
  const int32_t* b_kind;
  const double* b_param;
  double rcut_a, rcut_b;
  const double* acoeff; /* natom*na*2 */
  const double* bcoeff; /* nb*3 */
  /* semi-local ECPs (eval_ecp.py:149-200).  Channels of ECP atom k are
     [ecp_chan_off[k], ecp_chan_off[k+1]): non-local l = 0,1,.. first, LOCAL channel last
     (the reference's v_l column order).  Terms: sum coef * r^n * exp(-exp r^2). */
  int32_t necp;
  const int32_t* ecp_atom;     /* necp: atom index */
  const int32_t* ecp_chan_off; /* necp+1 */
  const int32_t* ecp_term_off; /* nchan_total+1 */
  const int32_t* ecp_term_n;
  const double* ecp_term_exp;
  const double* ecp_term_coef;
  int32_t has_slater; /* 0: Jastrow-only handle */
  /* three-body (electron-electron-ion) Jastrow (three_body_jastrow.py:19-63): its own a/b radial bases and
     ccoeff (natom, na3, na3, nb3, 3) as stored in wf.parameters["ccoeff"] (symmetrised in k,l by the library,
     :94-96).  na3 = 0: no three-body factor. */
  int32_t na3, nb3;
  const int32_t* a3_kind;
  const double* a3_param;
  const int32_t* b3_kind;
  const double* b3_param;
  double rcut_a3, rcut_b3;
  const double* ccoeff;
  /* periodic boundary conditions (PeriodicConfigs coord.py:137-252, MinimalImageDistance distance.py:83-159).
     pbc = 0: open system.  1: lattice vectors mutually orthogonal — displacements are folded in fractional
     coordinates (diagonal_dist / orthogonal_dist, :143-159).  2: general cell — fold, then argmin over the 27
     neighbouring cells (general_dist, :129-141).  Walker positions handed to the library are expected inside
     the cell (enforce_pbc, pbc/pbc.py:18-49), as PeriodicConfigs keeps them. */
  int32_t pbc;
  double lattice[9]; /* rows = lattice vectors of the simulation cell, bohr */
  /* Periodic orbitals (PBCOrbitalEvaluatorKpoints orbitals.py:118-255 over the lattice-summed AOs of
     numba/pbcgto.py:99-653), evaluated as Gamma-point orbitals of the simulation cell: the displacement point - atom
     is folded into the cell-centred parallelepiped, then every AO is summed over the cell translations Ls[j],
     j < num_Ls[atom of the shell] (all with |Ls[j]| <= sqrt(atom_cut) + half the cell's longest body diagonal),
     skipping an image when r^2 > atom_cut[atom]
     (pbcgto.py:213-214) or r^2 > shell_cut[shell] (:222, :356); cut-offs as max_Ls (:565-583).  Bloch orbitals
     of a primitive cell at the k-points of a zero-twist supercell are brought to this form on the host by folding
     the (real) Bloch phases into mo_up / mo_dn (pyqmc_amd/pbc.py:fold_mo_coeff).  nL = 0 with pbc != 0: no
     orbital tables (Jastrow-only handle). */
  int32_t nL;
  const double* Ls;        /* nL*3, sorted by norm (pbcgto.py:602-603) */
  const int32_t* num_Ls;   /* natom */
  const double* atom_cut;  /* natom, r^2 */
  const double* shell_cut; /* nshell, r^2 */
  /* Optional: reproduce which images the reference looks at.  The reference folds a point into the PRIMITIVE cell
     (wrap W = floor(r . inv(lattice_prim)), orbitals.py:201) and sums only the first num_Ls[a] entries of its
     norm-sorted primitive translation list (pbcgto.py:603-616, max_Ls :549-591), which is not a superset of what
     the cut-offs admit.  With member != NULL an image of atom A displaced by the cell translation
     (f + Ls[j]) (f = the fold applied to point - atom) is kept only if the primitive translation
     atom_n[A] + n(f) + img_n[j] - W is marked in member[member_class[A]], a (2M+1)^3 byte grid indexed
     [n0+M][n1+M][n2+M].  member = NULL: every image inside the cut-offs counts. */
  double lattice_prim[9];
  const int32_t* img_n;        /* nL*3: Ls[j] in units of the primitive lattice vectors */
  const int32_t* atom_n;       /* natom*3: the primitive translation that separates atom A from its primitive-cell original */
  const uint8_t* member;       /* n_member_class * (2M+1)^3 */
  const int32_t* member_class; /* natom */
  int32_t member_M, n_member_class;
  /* Complex orbitals (slater.py:212-216: Bloch coefficients at k-points off the time-reversal-invariant set).  With
     complex_orbitals != 0, mo_up / mo_dn hold [Re C | Im C] (nao x nmo_*, nmo_* = twice the number of orbitals),
     determinant occupations still index orbitals, and every complex output of the Slater entry points (sign of
     recompute / value, ratios of pqa_slater_eval, inverse / phases of pqa_slater_get_state) is (re, im) interleaved,
     i.e. the buffers are numpy complex128 arrays of the documented shapes.  Implemented for the wave-function protocol
     entry points; the fused sweep / energy entries refuse complex handles. */
  int32_t complex_orbitals;
  /* Twisted boundary conditions (Bloch orbitals with psi(r + L) = exp(i k_t . L) psi(r) for the lattice vectors L of the
     simulation cell; orbitals.py:201-213 get_wrapphase_complex).  twisted != 0 needs pbc, complex_orbitals and the
     periodic orbital tables.  The AOs become complex lattice sums sum_L exp(i k_t . L) phi(r - R - L); a handle in this
     mode takes UNFOLDED positions everywhere (position inside the cell + wrap . lattice, i.e. the electron's true
     coordinate) — the orbital kernel folds a point itself and derives the wrap phase exp(i k_t . wrap . lattice) of
     the reference from the integer wrap it removed; Jastrow, ECP and Ewald kernels use minimal-image displacements
     and do not care.  The fused sweep leaves the walkers unfolded. */
  int32_t twisted;
  double twist_k[3]; /* cartesian, 1/bohr */
} pqa_system_t;

/* ---- lifetime --------------------------------------------------------------------- */
int pqa_create(const pqa_system_t* sys, int device, pqa_handle_t** out);
void pqa_destroy(pqa_handle_t* h);
const char* pqa_last_error(const pqa_handle_t* h); /* h may be NULL: last create error */
int pqa_device_count(void);

/* wf.parameters[...] (slater.py:193-210, jastrowspin.py:48-53): names "det_coeff",
   "mo_coeff_alpha", "mo_coeff_beta", "acoeff", "bcoeff", "ccoeff".  Takes effect immediately for
   evaluations; cached walker state is refreshed by the next recompute, as in the reference. */
int pqa_set_param(pqa_handle_t* h, const char* name, const double* data, int64_t n);
int pqa_get_param(pqa_handle_t* h, const char* name, double* out, int64_t n);
/* pqa_get_param also answers "radial_table_info" (n = 2): the size in doubles of the radial tables the value-only
   orbital kernel reads for contracted shells of open systems and the largest fit error found when they were built,
   relative to sum |c| of the contraction (0, 0 with PQA_RADTAB=0 or a periodic system). */

/* ---- orbital evaluation (orbitals.py:85-96; numba/gto.py:89-254) --------------------- */
/* out (ncomp, npts, nao); ncomp 1 = value, 4 = +gradient, 5 = +laplacian */
int pqa_eval_ao(pqa_handle_t* h, const double* pts, int64_t npts, int ncomp, double* out);
/* out (ncomp, npts, nmo_spin); ncomp 1 or 5.  use_mfma=1: fused AO->MO MFMA kernel (product
   path); 0: plain VALU contraction of pqa_eval_ao output (A/B check only). */
int pqa_eval_mo(pqa_handle_t* h, int spin, const double* pts, int64_t npts, int ncomp, int use_mfma, double* out);

/* ---- Slater factor ------------------------------------------------------------------- */
/* Slater.recompute (slater.py:227-260) -> (sign, log|psi|) each (W) */
int pqa_slater_recompute(pqa_handle_t* h, const double* configs, int64_t W, double* sign, double* logabs);
/* Slater.value (slater.py:293-299) */
int pqa_slater_value(pqa_handle_t* h, double* sign, double* logabs);
/* _testrow/_testrowderiv (slater.py:301-380): multi-determinant ratios of replacing electron e's
   row by the orbitals at pts.  pts (nrow, npt, 3); row r belongs to walker widx[r] (NULL: r);
   out (ncomp, nrow*npt) with ncomp 1 (value: testvalue :429-446) or 5 (value, d/dx,d/dy,d/dz, laplacian:
   gradient_value :403-418, gradient_laplacian :420-427).  keep_saved (npt==1, widx==NULL, ncomp==5):
   keep the MO rows on the device for the next pqa_slater_update(use_saved=1). */
int pqa_slater_eval(pqa_handle_t* h, int e, const double* pts, int64_t nrow, int npt, const int32_t* widx,
                    int ncomp, int keep_saved, double* out);
/* Slater.updateinternals (slater.py:262-291) + Sherman-Morrison (slater.py:88-94); mask (W) bytes */
int pqa_slater_update(pqa_handle_t* h, int e, const double* epos, const uint8_t* mask, int use_saved);
/* any non-finite log-determinant of spin `spin`?  (the reference's trigger for a full recompute inside
   updateinternals, slater.py:269-275) */
int pqa_slater_has_zero(pqa_handle_t* h, int spin, int* flag);
/* test access: _inverse[s] (W, ndet_s, n, n) in the reference's [orbital, electron] order and
   _dets[s] (2, W, ndet_s) */
int pqa_slater_get_state(pqa_handle_t* h, int spin, double* inverse, double* dets);

/* ---- Jastrow factor ------------------------------------------------------------------ */
/* JastrowSpin.recompute / value (jastrowspin.py:56-109, 251-255) -> U (W) */
int pqa_jastrow_recompute(pqa_handle_t* h, const double* configs, int64_t W, double* logval);
int pqa_jastrow_value(pqa_handle_t* h, double* logval);
/* mode 0: testvalue (jastrowspin.py:387-419): out (nrow*npt) ratios
   mode 1: gradient_value (:296-340): out (4, nrow): dU/dx,dU/dy,dU/dz, ratio      (npt == 1)
   mode 2: gradient_laplacian (:342-385): out (4, nrow): grad U, lap U + |grad U|^2 (npt == 1) */
int pqa_jastrow_eval(pqa_handle_t* h, int e, const double* pts, int64_t nrow, int npt, const int32_t* widx,
                     int mode, double* out);
/* JastrowSpin.updateinternals (jastrowspin.py:111-137): also moves the handle's copy of the walkers */
int pqa_jastrow_update(pqa_handle_t* h, int e, const double* epos, const uint8_t* mask);
/* test access: _avalues (W,natom,na,2), _bvalues (W,nb,3), _configscurrent (W,N,3); any may be NULL */
int pqa_jastrow_get_state(pqa_handle_t* h, double* avalues, double* bvalues, double* configs);

/* ---- three-body Jastrow factor (ThreeBodyJastrow, three_body_jastrow.py) -------------- */
/* recompute :66-104 / value :191-195 -> U3 (W) */
int pqa_j3_recompute(pqa_handle_t* h, const double* configs, int64_t W, double* logval);
int pqa_j3_value(pqa_handle_t* h, double* logval);
/* testvalue :323-341 (mode 0), gradient_value :454-539 (mode 1), gradient_laplacian :541-655 (mode 2);
   argument and output layout as pqa_jastrow_eval */
int pqa_j3_eval(pqa_handle_t* h, int e, const double* pts, int64_t nrow, int npt, const int32_t* widx, int mode, double* out);
/* updateinternals :149-189: the factor keeps no partial sums here, so this only moves the stored walker
   coordinates when the handle has no two-body factor to do it */
int pqa_j3_update(pqa_handle_t* h, int e, const double* epos, const uint8_t* mask);

/* ---- fused device-resident path --------------------------------------------------------- */
/* MultiplyWF.recompute (multiplywf.py:81-88) for all factors of the handle; also fills the
   per-electron orbital cache used by the fused sweep/energy. */
int pqa_wf_recompute(pqa_handle_t* h, const double* configs, int64_t W, double* sign, double* logabs);
int pqa_wf_value(pqa_handle_t* h, double* sign, double* logabs);
int pqa_get_configs(pqa_handle_t* h, double* configs);

/* Slater.pgradient (slater.py:462-542): d Psi / Psi with respect to the determinant coefficients, d_det (W, ndet)
   (:495-505), and to the orbital coefficients of each spin, d_mo_* (W, nao, nmo_s) (:507-533, _testcol :382-388),
   from the resident state (call after recompute / updateinternals).  Any output may be NULL.  Complex handles (complex
   orbitals / twisted cells): every output is complex, (re, im) interleaved, nmo_s counting orbitals — the holomorphic
   derivative the reference forms in complex arithmetic; twisted cells take the (complex) AO values of the unfolded
   walkers with their wrap phases. */
int pqa_slater_pgradient(pqa_handle_t* h, double* d_det, double* d_mo_up, double* d_mo_dn);

/* ThreeBodyJastrow.pgradient (three_body_jastrow.py:657-719): dU/dccoeff, d_ccoeff (W, natom, na3, na3, nb3, 3), from the
   stored walker coordinates. */
int pqa_j3_pgradient(pqa_handle_t* h, double* d_ccoeff);

/* testvalue_many (Slater slater.py:448-460, JastrowSpin jastrowspin.py:421-455, ThreeBodyJastrow
   three_body_jastrow.py:343-372, MultiplyWF multiplywf.py:112-114; used by the density-matrix accumulators
   observables/obdm.py:175, tbdm.py:239): for ONE auxiliary position per row, the ratio Psi(electron es[i] moved there)/Psi
   for each of the ne listed electrons.  pts (nrow,3); widx (nrow) walker index per row or NULL (nrow = W);
   factors: bit 0 Slater, bit 1 two-body Jastrow, bit 2 three-body Jastrow (their product is returned);
   out (nrow, ne) doubles, or (nrow, ne, 2) = (re, im) on a handle with complex orbitals. */
int pqa_testvalue_many(pqa_handle_t* h, const int32_t* es, int ne, const double* pts, int64_t nrow, const int32_t* widx,
                       int factors, double* out);

/* Periodic Coulomb energy: tables of the Ewald sum (Ewald.__init__ / set_up_reciprocal_ewald_sum / set_ewald_constants,
   observables/ewald.py:95-190; built by pyqmc_amd/ewald.py).  gpoints (ng,3): reciprocal vectors of the positive half
   space with weight > 1e-10 (:372-388); gweight (ng) = 4 pi exp(-G^2/4 alpha^2) / (V G^2); ion_cos/ion_sin (ng): real and
   imaginary part of sum_I Z_I exp(i G.R_I) (:233-234); ee_const / ei_const: self + charged-system terms for this electron
   count (:180-184); ii: ion-ion energy incl. its constants (:353).  The real-space sum runs over the 27 cells of
   nlatvec = 1 (:113-123).  gidx (ng,3) int32 / recip (3,3): optional integer decomposition gpoints = gidx . recip
   (recip = 2 pi inv(lattice)^T, :374-388); with it the structure factors e^{iG.x} are built from three base phases per
   electron by complex multiplication instead of one sincos per (G, electron).  NULL: direct sincos.
   Required before pqa_energy / energies in pqa_vmc_sweeps on a handle with pbc != 0. */
int pqa_set_ewald(pqa_handle_t* h, double alpha, int32_t ng, const double* gpoints, const double* gweight,
                  const double* ion_cos, const double* ion_sin, double ee_const, double ei_const, double ii,
                  const int32_t* gidx, const double* recip);

/* Wrap counters (PeriodicConfigs.wrap, coord.py:137-189) accumulated by the accepted moves of the LAST pqa_vmc_sweeps
   call: wrap (W, nelec, 3) int32, to be added to the caller's counters.  The walkers themselves stay folded into the
   cell (make_irreducible, mc.py:121). */
int pqa_get_wrap(pqa_handle_t* h, int32_t* wrap);

/* MultiplyWF.gradient / gradient_value / gradient_laplacian (multiplywf.py:116-129) of a real Slater x two-body-Jastrow product
   living on this handle, electron e moved to pts (W, 3), in ONE call: out (9, W) = the five Slater ratio rows of pqa_slater_eval
   (value, d/dx, d/dy, d/dz, laplacian, all relative to the current determinant) followed by the four Jastrow rows of
   pqa_jastrow_eval in mode jmode (1: grad U (3), value ratio; 2: grad U (3), laplacian).  keep_saved as pqa_slater_eval.
   pqa_wf_update: MultiplyWF.updateinternals (multiplywf.py:102-106) — Sherman-Morrison + Jastrow sums for the walkers of `mask`
   (NULL: all), and *has_zero = 1 if a determinant of e's spin is zero / not finite AFTER the update: what slater.py:269-275 tests
   before the NEXT update of that spin (the caller carries the flag; pqa_slater_has_zero is the stand-alone test). */
int pqa_wf_eval(pqa_handle_t* h, int e, const double* pts, int jmode, int keep_saved, double* out);
int pqa_wf_update(pqa_handle_t* h, int e, const double* epos, const uint8_t* mask, int use_saved, int* has_zero);

/* EnergyAccumulator.__call__ (accumulators.py:60-75) on the device-resident walkers:
   out (6, W) rows ke, ee, ei, ecp, grad2, total (kinetic energy.py:57-65, Coulomb :28-54, ECP
   eval_ecp.py:21-146).  threshold as eval_ecp.ecp_mask (:135-146).  rot (N, necp, 3, 3) and
   unif (N, necp, W) replay the reference's random draws; NULL -> device Philox stream `seed`. */
int pqa_energy(pqa_handle_t* h, double threshold, const double* rot, const double* unif, uint64_t seed, double* out);

/* EnergyAccumulator(naip=...) (accumulators.py:48-51 -> eval_ecp.ecp(..., naip), eval_ecp.py:21, :228-252 get_P_l): the
   quadrature rule of the ECP integrator of pqa_energy / the energy pass of pqa_vmc_sweeps / pqa_dmc_steps.  naip one of 6, 12,
   18, 26, 32, 50 — the grids of Mitas, Shirley & Ceperley the reference tabulates (eval_ecp.py:278-336) — for every ECP atom;
   0 restores the reference's default (naip=None: 6 points for atoms with at most one non-local channel, 12 otherwise,
   eval_ecp.py:239-240).  Any other value is refused, as get_rot does (eval_ecp.py:266-267).  The T-move candidates
   (pqa_tmoves, pqa_dmc_steps) keep the default rule: the reference's nonlocal_tmoves does not pass naip (accumulators.py:82-84). */
int pqa_set_ecp_naip(pqa_handle_t* h, int32_t naip);

/* EnergyAccumulator(use_old_ecp=False) (accumulators.py:57-58) -> jax_ecp.ECPAccumulator (jax_ecp.py:22-142): the batched
   formulation of the ECP integral.  For every electron ALL ECP atoms' quadrature points form one table (evaluate_vl,
   jax_ecp.py:160-222: naip[k] points at ECP atom k, 0 or one of the six grids; no range cut-off, no stochastic mask), of which
   nselect_deterministic points of largest sum_l v_l^2 are evaluated with weight 1 and nselect_random are sampled from the
   rest with weight 1 / (nselect_random p) (downselect_move_info, jax_ecp.py:225-290); nothing is dropped when the two add up
   to the table size.  enable = 1 switches the ECP part of pqa_energy and of the energy passes of pqa_vmc_sweeps /
   pqa_dmc_steps to it, 0 back to the semi-local integrator (eval_ecp.py).  In this mode pqa_energy's `unif` is the
   (N, W, nselect_random) table of selection uniforms (jax_ecp.py:255) and `rot` (N, necp, 3, 3) as before; pqa_vmc_sweeps /
   pqa_dmc_steps draw both from the device streams (their tapes keep the semi-local layout and are refused).
   pqa_ecp_batched_nselected: slots per electron (the P of the outputs below).
   pqa_ecp_batched_moves: ECPAccumulator.nonlocal_tmoves (jax_ecp.py:110-135) for electron e — the selected points' positions
   pos (W, P, 3) and T-move weights weight (W, P) = sum_l [P_l > 0] (exp(-tau v_l / P_l) - 1) P_l; rot (necp, 3, 3),
   unif (W, nselect_random) or NULL -> device stream `seed`.  The ratios come from pqa_wf_testvalue at `pos`. */
int pqa_set_ecp_batched(pqa_handle_t* h, int32_t enable, const int32_t* naip, int32_t nselect_deterministic, int32_t nselect_random);
int pqa_ecp_batched_nselected(pqa_handle_t* h);
int pqa_ecp_batched_moves(pqa_handle_t* h, int e, double tau, const double* rot, const double* unif, uint64_t seed,
                          double* weight, double* pos);

/* EnergyAccumulator.nonlocal_tmoves -> eval_ecp.compute_tmoves (eval_ecp.py:43-80) for electron e of the resident
   walkers: the candidate T-moves over every ECP atom's quadrature points (P = pqa_tmove_npoints() per walker).
   rot (necp,3,3), unif (necp,W): the reference's per-atom random rotation / mask uniforms.  Outputs: ratio (W,P)
   Psi(candidate)/Psi (1 where the walker fails the ECP mask for that atom), weight (W,P) = sum_l (exp(-tau v_l)-1)
   (2l+1)P_l w_i (0 there), pos (W,P,3) candidate positions (current position there).  ratio = NULL: positions and weights
   only — complex handles take their (complex) ratios from pqa_wf_testvalue at those positions. */
/* The next pqa_dmc_steps call starts from the energies (E_L, |grad|^2) its predecessor's last step left on the device instead of
   evaluating the starting configuration again — what dmc_propagate carries from step to step (dmc.py:148-149, :199-200); for callers
   that make one call per step to run host accumulators in between (pyqmc_amd.dmc with OBDM / TBDM accumulators).  One-shot; refused
   when the wave-function state changed since that call (recompute, update, resample, sweep, parameter change).  With tapes the
   index-0 energy draws of the call are unused. */
int pqa_dmc_continue(pqa_handle_t* h, int on);
/* 1 while the energies of the last pqa_dmc_steps call still describe the resident walkers' state (pqa_dmc_continue would be honoured), else 0:
   a per-step caller whose host accumulators may have touched the state asks before it continues (pyqmc_amd.dmc; dmc.py:196-212). */
int pqa_dmc_can_continue(pqa_handle_t* h);
int pqa_tmove_npoints(pqa_handle_t* h);
int pqa_tmoves(pqa_handle_t* h, int e, double tau, double threshold, const double* rot, const double* unif, double* ratio,
               double* weight, double* pos);

/* vmc_worker move loop (mc.py:112-137) fused on the device, nsteps sweeps over all electrons,
   optionally followed each sweep by the energy accumulator (mc.py:142-148).
   gauss (nsteps,N,W,3) standard normals and unif (nsteps,N,W): replay tapes, or NULL -> Philox(seed).
   ecp_rot (nsteps,N,necp,3,3) / ecp_unif (nsteps,N,necp,W): same for the ECP draws.
   acceptance (nsteps): mean acceptance per sweep.  energy_mean (nsteps,6): walker means of
   ke,ee,ei,ecp,grad2,total (NULL: no energy evaluation).  accept_rec (nsteps,N,W) bytes, may be NULL. */
int pqa_vmc_sweeps(pqa_handle_t* h, double tstep, int nsteps, const double* gauss, const double* unif,
                   double threshold, const double* ecp_rot, const double* ecp_unif, uint64_t seed,
                   double* acceptance, double* energy_mean, uint8_t* accept_rec);

/* branch (pyqmc/method/dmc.py:342-376) without the recompute that follows it in the reference (:155): walker w of the
   resident ensemble becomes a copy of walker newinds[w] — coordinates, inverses, determinant signs / logs, orbital-row
   cache and Jastrow sums are gathered on the device (the counterpart of configs.resample(newinds) coord.py:64-70,
   191-198 for the wave-function state).  newinds (W) int32 in [0, W). */
int pqa_resample(pqa_handle_t* h, const int32_t* newinds);

/* Distributed branching (pyqmc/method/dmc.py:342-376 over ranks): after every rank has computed the
   identical global comb, a walker whose new owner differs from its old one travels as coordinates only.
   pqa_get_walkers: coordinates (n,N,3) of the resident walkers idx[k] (duplicates allowed) into `out` — host memory or a
   device buffer handed straight to RCCL.
   pqa_branch_exchange: the rank's new ensemble = the resident walkers keep_src[0..nkeep) (their whole wave-function state is
   gathered on the device, as pqa_resample) followed by nrecv received walkers with coordinates recv_x (nrecv,N,3; host or
   device), for which ALONE the state is recomputed (the reference recomputes every walker after a branch, dmc.py:155).
   nkeep + nrecv must equal the resident walker count. */
int pqa_get_walkers(pqa_handle_t* h, const int32_t* idx, int64_t n, double* out);
int pqa_branch_exchange(pqa_handle_t* h, const int32_t* keep_src, int64_t nkeep, const double* recv_x, int64_t nrecv);

/* dmc_propagate's step loop (pyqmc/method/dmc.py:123-221) fused on the device, open or periodic systems, real and
   (round 3) complex wave functions.  Complex: no node constraint in the drift-diffusion (dmc.py:64-66 is real-only),
   weights from Re E_L, T-move amplitudes from Re[Psi(R')/Psi(R)] (the reference's propose_tmoves orders complex
   amplitudes, which is undefined; golden g30 pins this rule), and step_avg has EIGHT numbers per step: the seven below,
   then the weighted mean of Im ecp (= Im total).  Per step: per step (1) one T-move per electron (compute_tmoves eval_ecp.py:43-80, propose_tmoves dmc.py:73-120,
   masked updateinternals :160-168), (2) one drift-diffusion move per electron with Umrigar's limited drift
   (limdrift :22-35) and fixed-node rejection (propose_drift_diffusion :38-70), (3) the energy accumulator, the
   branching factor compute_S (:224-235), weights *= exp(tau r2_acc/r2_prop (S_new+S_old)/2) and the weighted step
   averages (:196-215).  No branching: the caller combs between calls, as rundmc does (:342-376).
   The walkers must be resident and current (pqa_wf_recompute); the starting energy is evaluated first (:146-149).
   weights (W) in/out.  step_avg (nsteps,7) [complex: (nsteps,8)]: sum_w w_w row_w / sum_w w_w for rows ke, ee, ei, ecp,
   grad2, total (real parts), then the mean weight.  step_acc (nsteps,2): acceptance and T-move acceptance (accepted / (W nelec)).
   tapes NULL -> device Philox streams keyed by (seed, walker, electron, step); otherwise every pointer the system
   needs must be set (the ECP ones only when the system has ECP atoms):
     gauss (nsteps,N,W,3) standard normals, unif (nsteps,N,W)                      drift-diffusion moves
     tm_rot (nsteps,N,necp,3,3), tm_unif (nsteps,N,necp,W), tm_u1, tm_u2 (nsteps,N,W)  T-move grid, mask, selection, acceptance
     ecp_rot (nsteps+1,N,necp,3,3), ecp_unif (nsteps+1,N,necp,W)                  energy draws; index 0 = starting energy */
typedef struct {
  const double* gauss;
  const double* unif;
  const double* tm_rot;
  const double* tm_unif;
  const double* tm_u1;
  const double* tm_u2;
  const double* ecp_rot;
  const double* ecp_unif;
} pqa_dmc_tapes_t;
int pqa_dmc_steps(pqa_handle_t* h, double tstep, int nsteps, double branchcut, double e_trial, double e_est,
                  double threshold, double* weights, const pqa_dmc_tapes_t* tapes, uint64_t seed, double* step_avg,
                  double* step_acc);

/* ---- density matrices and parameter-gradient moments ---------------------------- */
/* Auxiliary one-electron Metropolis walk of the density-matrix estimators (sample_onebody,
   pyqmc/observables/obdm.py:215-250) on a handle whose orbitals are the estimator's basis (built from orb_coeff alone):
   n walkers distributed as f(r) = sum_i |phi_i(r)|^2 (orbitals of `spin`), nsamples proposals r + sqrt(tstep) z each,
   accepted with probability f(r')/f(r).  The walk stays on the device (one orbital launch + one accept kernel per
   sample).  pos (n,3) in/out; periodic handles take and return UNFOLDED coordinates (the orbital kernel folds every point
   itself and gives twisted cells their wrap phase, orbitals.py:201-213).  gauss (nsamples,n,3) standard normals and unif
   (nsamples,n): replay tapes in the reference's draw order, or both NULL -> Philox(seed).  The last nkeep samples
   (positions, orbital rows, densities) stay resident in `slot` (0 or 1) for pqa_obdm_accumulate / pqa_tbdm_accumulate;
   keep_pos (nkeep,n,3) receives their positions (may be NULL), accept (nsamples,n) the decisions as 0/1 (may be NULL). */
int pqa_dm_walk(pqa_handle_t* h, int slot, int spin, int64_t n, int nsamples, double tstep, double* pos, const double* gauss,
                const double* unif, uint64_t seed, int nkeep, double* keep_pos, double* accept);
/* Basis orbitals at the configurations' electrons (OBDMAccumulator.evaluate_orbitals obdm.py:199-201, tbdm.py:203-212):
   pts (npts,3) = (nconf, nelec_listed, 3), kept on the device in `slot`. */
int pqa_dm_points(pqa_handle_t* h, int slot, int spin, const double* pts, int64_t npts);
/* One sweep of the one-body estimator (obdm.py:170-190) for kept sample k of `slot`: configuration n uses auxiliary walker
   assign[n]; ratio (nconf,nelec) = Psi(r_e -> r')/Psi from wf.testvalue_many (interleaved complex if ratio_complex).
   first != 0 starts new accumulators value (nconf,norb,norb), norm (nconf,norb); otherwise adds. */
int pqa_obdm_accumulate(pqa_handle_t* h, int slot, int k, int64_t nconf, int nelec, const int32_t* assign, const double* ratio,
                        int ratio_complex, int first);
/* One sweep of the two-body estimator (tbdm.py:232-277): slot 0 / 1 hold the walks and electron orbitals of the first /
   second spin of the sector.  ratio (nconf,nea,neb) = Psi(r_a -> r1', r_b -> r2')/Psi, 0 for a pair naming one electron
   twice; ijkl (4,ntuple) int32.  Accumulates value (nconf,ntuple), norm_a (nconf,norb_a), norm_b (nconf,norb_b). */
int pqa_tbdm_accumulate(pqa_handle_t* h, int k, int64_t nconf, int nea, int neb, const int32_t* assign_a, const int32_t* assign_b,
                        const double* ratio, int ratio_complex, const int32_t* ijkl, int ntuple, int first);
/* Read an accumulator times `scale`: which = 0 value (ncol = entries per configuration, doubled when complex), 1 norm /
   norm_a, 2 norm_b (ncol = orbitals).  mean = 0: (nconf,ncol); mean != 0: (ncol,) averaged over the configurations on
   the device (the accumulators' avg(), obdm.py:195-197). */
int pqa_dm_fetch(pqa_handle_t* h, int which, int ncol, double scale, int mean, double* out);
/* C (P,Q) = A^T B for A (n,P), B (n,Q) on the fp64 matrix cores: the moment matrix dpidpj = dp^T diag(w f) dp of
   StochasticReconfiguration.avg (stochastic_reconfiguration.py:106-114). */
int pqa_gram(pqa_handle_t* h, int64_t n, int P, int Q, const double* A, const double* B, double* C);

/* The random numbers the fused sweeps (pqa_vmc_sweeps, pqa_dmc_steps without tapes) draw for sweep `step` of `seed`:
   gauss (N,W,3) standard normals and unif (N,W) Metropolis uniforms of walkers 0..W-1 — Philox4x32-10 keyed by
   (seed; walker, electron, stream, step).  Test entry: lets the CPU oracle replay a device-RNG trajectory (the role
   np.random.seed plays for the reference's vmc_worker, mc.py:119,131). */
int pqa_philox_tapes(pqa_handle_t* h, uint64_t seed, int step, int64_t W, double* gauss, double* unif);
/* The same for pqa_dmc_steps without tapes: every array of pqa_dmc_tapes_t for `nsteps` steps of `seed`, restricted to walkers
   0..W-1, in the tape layouts above (the members of `out` point to caller-allocated HOST arrays of those shapes; tm_* may be
   NULL for systems without ECPs) — T-move grid rotations, mask uniforms, selection / acceptance uniforms, drift-diffusion
   normals and uniforms, and the energy evaluations' quadrature rotations and mask uniforms (index 0 = starting energy).
   Test entry: the CPU oracle's dmc_propagate (pyqmc/method/dmc.py:123-221) replays a device-RNG block walker by walker. */
int pqa_philox_dmc_tapes(pqa_handle_t* h, uint64_t seed, int nsteps, int64_t W, pqa_dmc_tapes_t* out);

/* ---- measurement -------------------------------------------------------------------- */
/* HIP-event timing on the handle's own stream (torch.cuda.Event only sees torch's stream). */
int pqa_timer_start(pqa_handle_t* h);
int pqa_timer_stop(pqa_handle_t* h, double* elapsed_ms); /* synchronises the stream */
int pqa_sync(pqa_handle_t* h);
/* per-kernel-class accounting: enable=1 brackets every launch of the orbital (AO->MO MFMA) kernel
   with events; query returns launches, total ms, total points*ncomp processed since enable. */
int pqa_profile_enable(pqa_handle_t* h, int enable);
int pqa_profile_query(pqa_handle_t* h, int64_t* launches, double* total_ms, double* point_comps);
/* same accounting for the streaming kernel of the fused lane-per-walker sweep: the flush launches of the blocked
   Sherman-Morrison update (rows outside the current electron block, once per block of KB moves; none when KB = n) */
int pqa_profile_query_commit(pqa_handle_t* h, int64_t* launches, double* total_ms);
/* and for the partial-sum kernel of the fused sweep (k_move_part_lw: Slater ratio sums + the Jastrow distance sums of the
   proposal, jastrowspin.py:296-340): launches bracketed (the proposal-side launch of every prof-stride-th electron), their
   total ms, and the number of partial-sum groups per walker the launch geometry uses at the resident walker count */
int pqa_profile_query_part(pqa_handle_t* h, int64_t* launches, double* total_ms, int* groups);
/* points evaluated by the ECP integrator in the last pqa_energy call (data dependent) */
int pqa_last_ecp_points(pqa_handle_t* h, int64_t* npoints);

#ifdef __cplusplus
}
#endif
#endif
