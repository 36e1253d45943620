/* TEST / MEASUREMENT INFRASTRUCTURE — compiled atomic-orbital evaluator of the CPU oracle.
 *
 * The reference's default AO back end is compiled code (PySCF's libcgto through mol.eval_gto, pyqmc/wf/orbitals.py:46-51; the
 * in-repo alternative is numba-JIT-ed, pyqmc/wf/numba/gto.py).  SURVEY.md 8(d) therefore asks the CPU baseline to evaluate its
 * AOs with a single-thread compiled routine, so that the AO share of the timed baseline is what a compiled back end costs, not
 * what NumPy temporaries cost.  This file restates pyqmc/wf/numba/gto.py in plain C, loop for loop:
 *   mol_eval_gto       gto.py:89-136   (atom loop; r^2; solid harmonics of the atom's max l; per shell: radial sum, product)
 *   mol_eval_gto_grad  gto.py:139-194  (grad = dS R + S dR)
 *   mol_eval_gto_lap   gto.py:197-254  (lap = S lapR + 2 grad S . grad R, lap S = 0)
 *   radial_gto / _grad / _lap  gto.py:257-321   (R = sum c e^{-a r^2}; dR_i = -2 a x_i c e^{..}; lapR = sum 2a(2a r^2-3) c e^{..})
 *   real solid harmonics l <= 3  numba/spherical_harmonics.py:40-200  (orthonormal Y_lm r^l; l = 1 ordered x, y, z)
 * Shell coefficients arrive normalised (oracle/gto.py: normalized_contraction, gto.py:375-405).  l >= 4 is not built here (the
 * oracle falls back to its NumPy routine).  Only oracle/gto.py loads this library (ctypes); nothing under pyqmc_amd/ does.
 *
 * out layout: [ncomp][npts][nao] (the layout of mol_eval_gto* after the final transpose).
 */
#include <math.h>
#include <stddef.h>

#define S0 0.28209479177387814   /* 1/(2 sqrt(pi)) */
#define P1 0.4886025119029199    /* sqrt(3/(4 pi)) */
#define DXY 1.0925484305920792   /* 1/2 sqrt(15/pi) */
#define DZ2 0.31539156525252005  /* 1/4 sqrt(5/pi) */
#define DX2 0.5462742152960396   /* 1/4 sqrt(15/pi) */
#define F3 0.5900435899266435    /* 1/4 sqrt(35/(2 pi)) */
#define F2 2.890611442640554     /* 1/2 sqrt(105/pi) */
#define F1 0.4570457994644658    /* 1/4 sqrt(21/(2 pi)) */
#define F0 0.3731763325901154    /* 1/4 sqrt(7/pi) */
#define F2C 1.445305721320277    /* 1/4 sqrt(105/pi) */

/* S[16], dS[16][3] for l <= lmax (<= 3) at v */
static void solid_harmonics(const double* v, int lmax, int deriv, double* S, double (*dS)[3]) {
  const double x = v[0], y = v[1], z = v[2];
  const int n = (lmax + 1) * (lmax + 1);
  if (deriv)
    for (int i = 0; i < n; ++i) dS[i][0] = dS[i][1] = dS[i][2] = 0.0;
  S[0] = S0;
  if (lmax >= 1) {
    S[1] = P1 * x; S[2] = P1 * y; S[3] = P1 * z;
    if (deriv) { dS[1][0] = P1; dS[2][1] = P1; dS[3][2] = P1; }
  }
  if (lmax >= 2) {
    S[4] = DXY * x * y; S[5] = DXY * y * z; S[6] = DZ2 * (2 * z * z - x * x - y * y); S[7] = DXY * x * z; S[8] = DX2 * (x * x - y * y);
    if (deriv) {
      dS[4][0] = DXY * y; dS[4][1] = DXY * x;
      dS[5][1] = DXY * z; dS[5][2] = DXY * y;
      dS[6][0] = -2 * DZ2 * x; dS[6][1] = -2 * DZ2 * y; dS[6][2] = 4 * DZ2 * z;
      dS[7][0] = DXY * z; dS[7][2] = DXY * x;
      dS[8][0] = 2 * DX2 * x; dS[8][1] = -2 * DX2 * y;
    }
  }
  if (lmax >= 3) {
    const double x2 = x * x, y2 = y * y, z2 = z * z;
    S[9] = F3 * y * (3 * x2 - y2); S[10] = F2 * x * y * z; S[11] = F1 * y * (4 * z2 - x2 - y2);
    S[12] = F0 * z * (2 * z2 - 3 * x2 - 3 * y2); S[13] = F1 * x * (4 * z2 - x2 - y2); S[14] = F2C * z * (x2 - y2); S[15] = F3 * x * (x2 - 3 * y2);
    if (deriv) {
      dS[9][0] = F3 * 6 * x * y; dS[9][1] = F3 * (3 * x2 - 3 * y2);
      dS[10][0] = F2 * y * z; dS[10][1] = F2 * x * z; dS[10][2] = F2 * x * y;
      dS[11][0] = F1 * (-2 * x * y); dS[11][1] = F1 * (4 * z2 - x2 - 3 * y2); dS[11][2] = F1 * 8 * y * z;
      dS[12][0] = F0 * (-6 * x * z); dS[12][1] = F0 * (-6 * y * z); dS[12][2] = F0 * (6 * z2 - 3 * x2 - 3 * y2);
      dS[13][0] = F1 * (4 * z2 - 3 * x2 - y2); dS[13][1] = F1 * (-2 * x * y); dS[13][2] = F1 * 8 * x * z;
      dS[14][0] = F2C * 2 * x * z; dS[14][1] = -F2C * 2 * y * z; dS[14][2] = F2C * (x2 - y2);
      dS[15][0] = F3 * (3 * x2 - 3 * y2); dS[15][1] = -F3 * 6 * x * y;
    }
  }
}

/* shells must be grouped by atom (as AOTable builds them).  Returns 0, or -1 if a shell has l > 3.
 * Loop structure of the reference: per atom the displacement / r^2 arrays of all points, per shell the radial sums with the
 * POINT loop innermost (radial_gto*: `for c in coeffs: for a in range(npts)`), which is what lets the compiler vectorise the
 * exponential (numba compiles these loops with fastmath=True; gcc -O3 here — -ffast-math was measured and changes nothing: oracle/Makefile).
 * Points go in blocks that fit L1. */
#define BLK 128
int ao_eval(int ncomp, long npts, const double* pts, int nshell, const int* shell_atom, const int* shell_l, const int* prim_off,
            const double* exps, const double* coefs, const int* ao_off, const double* atom_xyz, int nao, double* out) {
  const int deriv = ncomp > 1;
  const size_t plane = (size_t)npts * nao;
  for (int s = 0; s < nshell; ++s)
    if (shell_l[s] > 3) return -1;
  double vx[BLK], vy[BLK], vz[BLK], r2[BLK], R[BLK], dRs[BLK], lapR[BLK];
  double S[BLK][16], dS[BLK][16][3];
  for (long p0 = 0; p0 < npts; p0 += BLK) {
    const int nb_ = (int)((npts - p0 < BLK) ? npts - p0 : BLK);
    for (int s0 = 0; s0 < nshell;) {
      const int ia = shell_atom[s0];
      int s1 = s0, lmax = 0;
      while (s1 < nshell && shell_atom[s1] == ia) { if (shell_l[s1] > lmax) lmax = shell_l[s1]; ++s1; }
      for (int p = 0; p < nb_; ++p) {
        vx[p] = pts[3 * (p0 + p)] - atom_xyz[3 * ia]; vy[p] = pts[3 * (p0 + p) + 1] - atom_xyz[3 * ia + 1]; vz[p] = pts[3 * (p0 + p) + 2] - atom_xyz[3 * ia + 2];
        r2[p] = vx[p] * vx[p] + vy[p] * vy[p] + vz[p] * vz[p];
        const double v[3] = {vx[p], vy[p], vz[p]};
        solid_harmonics(v, lmax, deriv, S[p], dS[p]);
      }
      for (int s = s0; s < s1; ++s) {
        const int l = shell_l[s], nb = 2 * l + 1, off = ao_off[s], sl = l * l;
        for (int p = 0; p < nb_; ++p) R[p] = dRs[p] = lapR[p] = 0.0;
        for (int q = prim_off[s]; q < prim_off[s + 1]; ++q) {
          const double a = exps[q], c = coefs[q];
          if (ncomp == 5) {
            for (int p = 0; p < nb_; ++p) { const double t = c * exp(-r2[p] * a); R[p] += t; dRs[p] += a * t; lapR[p] += t * (2.0 * a) * (2.0 * a * r2[p] - 3.0); }
          } else if (deriv) {
            for (int p = 0; p < nb_; ++p) { const double t = c * exp(-r2[p] * a); R[p] += t; dRs[p] += a * t; }
          } else {
            for (int p = 0; p < nb_; ++p) R[p] += c * exp(-r2[p] * a);
          }
        }
        for (int p = 0; p < nb_; ++p) {
          const double v[3] = {vx[p], vy[p], vz[p]}, d = -2.0 * dRs[p]; /* dR/dx_i = d * x_i */
          double* o = out + (size_t)(p0 + p) * nao + off;
          for (int m = 0; m < nb; ++m) {
            o[m] = S[p][sl + m] * R[p];
            if (deriv)
              for (int i = 0; i < 3; ++i) o[(size_t)(1 + i) * plane + m] = dS[p][sl + m][i] * R[p] + S[p][sl + m] * d * v[i];
            if (ncomp == 5)
              o[4 * plane + m] = S[p][sl + m] * lapR[p] + 2.0 * d * (dS[p][sl + m][0] * v[0] + dS[p][sl + m][1] * v[1] + dS[p][sl + m][2] * v[2]);
          }
        }
      }
      s0 = s1;
    }
  }
  return 0;
}

/* ---------------------------------------------------------------------------------------------------------------------------
 * Periodic (lattice-summed, Bloch-phased) AOs: pyqmc/wf/numba/pbcgto.py:99-506 restated like the molecular routine above —
 *   pbc_eval_gto / _grad / _lap   pbcgto.py:205-225, 300-330, 400-440: per atom, per translation j < num_Ls[atom] of the
 *   norm-sorted list Ls, v = r - R_atom - Ls[j]; points with r^2 > atom_cut[atom] are skipped (:217); per shell points pass
 *   r^2 < shell_cut (values, :254) / not r^2 > shell_cut (derivatives, :356); the shell's functions are added to every
 *   k-point's output weighted by the Bloch phase e^{i k.Ls[j]} (:222-224, phases precomputed :620-621).
 * The same tests, the same image order, the same sums as oracle/pbc.py:eval_ao_pbc (the NumPy restatement this is checked
 * against to 1e-12).  phases: [nL][nk] real (cplx = 0) or [nL][nk][2] (re, im); out: [nk][ncomp][npts][nao] doubles, or
 * interleaved (re, im) pairs when cplx.  pts must already be folded into the primitive cell (as PeriodicOrbitals.aos does). */
int ao_eval_pbc(int ncomp, long npts, const double* pts, int nshell, const int* shell_atom, const int* shell_l, const int* prim_off,
                const double* exps, const double* coefs, const int* ao_off, const double* atom_xyz, int nao,
                int nk, const double* Ls, const int* num_Ls, const double* atom_cut, const double* shell_cut, const double* phases, int cplx,
                double* out) {
  const int deriv = ncomp > 1, cf = cplx ? 2 : 1;
  const size_t plane = (size_t)npts * nao * cf, kplane = plane * ncomp;
  for (int s = 0; s < nshell; ++s)
    if (shell_l[s] > 3) return -1;
  for (size_t i = 0; i < kplane * nk; ++i) out[i] = 0.0;
  double vx[BLK], vy[BLK], vz[BLK], r2[BLK], R[BLK], dRs[BLK], lapR[BLK];
  double S[BLK][16], dS[BLK][16][3];
  int idx[BLK], sidx[BLK];
  double val[5][7];
  for (long p0 = 0; p0 < npts; p0 += BLK) {
    const int nb_ = (int)((npts - p0 < BLK) ? npts - p0 : BLK);
    for (int s0 = 0; s0 < nshell;) {
      const int ia = shell_atom[s0];
      int s1 = s0, lmax = 0;
      while (s1 < nshell && shell_atom[s1] == ia) { if (shell_l[s1] > lmax) lmax = shell_l[s1]; ++s1; }
      for (int j = 0; j < num_Ls[ia]; ++j) {
        int n = 0;
        for (int p = 0; p < nb_; ++p) {  /* points inside the atom's cut-off for this image */
          const double x = pts[3 * (p0 + p)] - atom_xyz[3 * ia] - Ls[3 * j], y = pts[3 * (p0 + p) + 1] - atom_xyz[3 * ia + 1] - Ls[3 * j + 1],
                       z = pts[3 * (p0 + p) + 2] - atom_xyz[3 * ia + 2] - Ls[3 * j + 2];
          const double rr = x * x + y * y + z * z;
          if (rr > atom_cut[ia]) continue;
          vx[n] = x; vy[n] = y; vz[n] = z; r2[n] = rr; idx[n] = p;
          ++n;
        }
        if (n == 0) continue;
        for (int q = 0; q < n; ++q) {
          const double v[3] = {vx[q], vy[q], vz[q]};
          solid_harmonics(v, lmax, deriv, S[q], dS[q]);
        }
        for (int s = s0; s < s1; ++s) {
          const int l = shell_l[s], nb = 2 * l + 1, off = ao_off[s], sl = l * l;
          int m_ = 0;
          for (int q = 0; q < n; ++q)
            if (ncomp == 1 ? (r2[q] < shell_cut[s]) : !(r2[q] > shell_cut[s])) sidx[m_++] = q;
          if (m_ == 0) continue;
          for (int t = 0; t < m_; ++t) R[t] = dRs[t] = lapR[t] = 0.0;
          for (int pq = prim_off[s]; pq < prim_off[s + 1]; ++pq) {
            const double a = exps[pq], c = coefs[pq];
            if (ncomp == 5) {
              for (int t = 0; t < m_; ++t) { const double rr = r2[sidx[t]], e = c * exp(-rr * a); R[t] += e; dRs[t] += a * e; lapR[t] += e * (2.0 * a) * (2.0 * a * rr - 3.0); }
            } else if (deriv) {
              for (int t = 0; t < m_; ++t) { const double e = c * exp(-r2[sidx[t]] * a); R[t] += e; dRs[t] += a * e; }
            } else {
              for (int t = 0; t < m_; ++t) R[t] += c * exp(-r2[sidx[t]] * a);
            }
          }
          for (int t = 0; t < m_; ++t) {
            const int q = sidx[t];
            const double v[3] = {vx[q], vy[q], vz[q]}, d = -2.0 * dRs[t];
            for (int m = 0; m < nb; ++m) {
              val[0][m] = S[q][sl + m] * R[t];
              if (deriv)
                for (int i = 0; i < 3; ++i) val[1 + i][m] = dS[q][sl + m][i] * R[t] + S[q][sl + m] * d * v[i];
              if (ncomp == 5)
                val[4][m] = S[q][sl + m] * lapR[t] + 2.0 * d * (dS[q][sl + m][0] * v[0] + dS[q][sl + m][1] * v[1] + dS[q][sl + m][2] * v[2]);
            }
            for (int k = 0; k < nk; ++k) {
              double* o = out + (size_t)k * kplane + ((size_t)(p0 + idx[q]) * nao + off) * cf;
              if (cplx) {
                const double pr = phases[((size_t)j * nk + k) * 2], pi = phases[((size_t)j * nk + k) * 2 + 1];
                for (int c = 0; c < ncomp; ++c)
                  for (int m = 0; m < nb; ++m) { o[c * plane + 2 * m] += pr * val[c][m]; o[c * plane + 2 * m + 1] += pi * val[c][m]; }
              } else {
                const double pr = phases[(size_t)j * nk + k];
                for (int c = 0; c < ncomp; ++c)
                  for (int m = 0; m < nb; ++m) o[c * plane + m] += pr * val[c][m];
              }
            }
          }
        }
      }
      s0 = s1;
    }
  }
  return 0;
}
