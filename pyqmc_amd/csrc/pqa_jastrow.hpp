// One- and two-body spin Jastrow kernels — one walker per wavefront, lanes over the other
// electrons / the ions ("distance sweep").
//
// Reference semantics: pyqmc/wf/jastrowspin.py (recompute :56-109, value :251-255, testvalue
// :387-419, gradient_value :296-340, gradient_laplacian :342-385, updateinternals :111-137) with
// the radial functions of pyqmc/wf/func3d.py (PolyPade :25-49, CutoffCusp :125-182, zero outside
// rcut :299-324).
//
// Deliberate difference: the reference keeps per-electron partial sums (_a_partial, _b_partial:
// 53 KB per walker at 64 electrons) and patches N rows of them on every accepted move.  Here the
// one-electron sum U_e(r) = sum_I c^a a(|r-R_I|) + sum_{j!=e} c^b b(|r-r_j|) is re-evaluated at the
// old position from the stored walker coordinates instead (same arithmetic, no drift, 100x less
// state to stream), and only the reference's public sums _avalues/_bvalues are maintained.
#pragma once
#include "pqa_common.hpp"

struct JastrowState {
  double* x;        // [W][N][3] walker coordinates (the reference's _configscurrent)
  double* avalues;  // [W][natom][na][2]
  double* bvalues;  // [W][nb][3]
};

// value, (dU/dr)/r and laplacian of one radial basis function for r < rcut
__device__ __forceinline__ void jas_radial(int kind, double par, double rcut, double r, double& val, double& gfac,
                                           double& lap) {
  if (kind == 0) {  // PolyPade(beta)  func3d.py:25-49
    const double z1 = r / rcut - 1.0, z12 = z1 * z1;
    const double p = (3.0 * z12 + 4.0 * z1) * z12 + 1.0;
    const double obp = 1.0 / (1.0 + par * p);
    val = (1.0 - p) * obp;
    gfac = -(1.0 + par) * 12.0 / (rcut * rcut) * obp * obp * z12;
    lap = gfac * (5.0 + 2.0 / z1 - 24.0 * par * (z1 + 1.0) * (z1 + 1.0) * z12 * obp);
  } else {  // CutoffCusp(gamma)  func3d.py:125-182
    const double y = r / rcut, y1 = y - 1.0, a = y1 * y1;
    const double b = (a * y1 + 1.0) / 3.0;
    const double ogb = 1.0 / (1.0 + par * b);
    const double c = ogb * ogb / r;
    val = (-b * ogb + 1.0 / (3.0 + par)) * rcut;
    gfac = -a * c;
    lap = -2.0 * c * ((y1 - a * a * par * ogb) * y + a);
  }
}

// U_e(r), grad U_e, lap U_e (bare laplacian, without |grad|^2) for electron e placed at r, against
// the walker coordinates xw (electron e itself skipped).  MODE 0: value; 1: value+grad; 2: grad+lap.
template <int MODE>
__device__ __forceinline__ void jas_eval(const SysDev& S, const double* __restrict__ xw, int e, double rx, double ry,
                                         double rz, double& U, double (&g)[3], double& lapU) {
  const int lane = threadIdx.x & 63;
  const int edown = e >= S.nup;
  double u = 0.0, gx = 0.0, gy = 0.0, gz = 0.0, lp = 0.0;
  for (int j = lane; j < S.nelec; j += 64) {
    if (j == e) continue;
    const double dx = rx - xw[3 * j], dy = ry - xw[3 * j + 1], dz = rz - xw[3 * j + 2];
    const double r = sqrt(dx * dx + dy * dy + dz * dz);
    if (r < S.rcut_b) {
      const int col = edown + (j >= S.nup);
      for (int l = 0; l < S.nb; ++l) {
        double v, gf, lpl;
        jas_radial(S.b_kind[l], S.b_param[l], S.rcut_b, r, v, gf, lpl);
        const double c = S.bcoeff[l * 3 + col];
        u += c * v;
        if (MODE >= 1) { gx += c * gf * dx; gy += c * gf * dy; gz += c * gf * dz; }
        if (MODE == 2) lp += c * lpl;
      }
    }
  }
  for (int I = lane; I < S.natom; I += 64) {
    const double dx = rx - S.atom_xyz[3 * I], dy = ry - S.atom_xyz[3 * I + 1], dz = rz - S.atom_xyz[3 * I + 2];
    const double r = sqrt(dx * dx + dy * dy + dz * dz);
    if (r < S.rcut_a) {
      for (int k = 0; k < S.na; ++k) {
        double v, gf, lpl;
        jas_radial(S.a_kind[k], S.a_param[k], S.rcut_a, r, v, gf, lpl);
        const double c = S.acoeff[(I * S.na + k) * 2 + edown];
        u += c * v;
        if (MODE >= 1) { gx += c * gf * dx; gy += c * gf * dy; gz += c * gf * dz; }
        if (MODE == 2) lp += c * lpl;
      }
    }
  }
  U = (MODE <= 1) ? wave_sum(u) : 0.0;
  if (MODE >= 1) { g[0] = wave_sum(gx); g[1] = wave_sum(gy); g[2] = wave_sum(gz); }
  lapU = (MODE == 2) ? wave_sum(lp) : 0.0;
}

// Commit the move of electron e of walker w to rn: patch _avalues/_bvalues with (new - old) and move
// the stored coordinate.  jastrowspin.py:131-137.
__device__ __forceinline__ void jas_commit(const SysDev& S, const JastrowState& js, long w, int e, double rx, double ry,
                                           double rz) {
  const int lane = threadIdx.x & 63;
  const int edown = e >= S.nup;
  double* xw = js.x + (size_t)w * S.nelec * 3;
  const double ox = xw[3 * e], oy = xw[3 * e + 1], oz = xw[3 * e + 2];
  if (S.nb > 0) {
    double diff[2][PQA_MAXBAS];
#pragma unroll
    for (int l = 0; l < PQA_MAXBAS; ++l) diff[0][l] = diff[1][l] = 0.0;
    for (int j = lane; j < S.nelec; j += 64) {
      if (j == e) continue;
      const double jx = xw[3 * j], jy = xw[3 * j + 1], jz = xw[3 * j + 2];
      const double rn = sqrt((rx - jx) * (rx - jx) + (ry - jy) * (ry - jy) + (rz - jz) * (rz - jz));
      const double ro = sqrt((ox - jx) * (ox - jx) + (oy - jy) * (oy - jy) + (oz - jz) * (oz - jz));
      const int grp = j >= S.nup;
#pragma unroll
      for (int l = 0; l < PQA_MAXBAS; ++l) {
        if (l < S.nb) {
          double vn = 0.0, vo = 0.0, t1, t2;
          if (rn < S.rcut_b) jas_radial(S.b_kind[l], S.b_param[l], S.rcut_b, rn, vn, t1, t2);
          if (ro < S.rcut_b) jas_radial(S.b_kind[l], S.b_param[l], S.rcut_b, ro, vo, t1, t2);
          if (grp) diff[1][l] += vn - vo; else diff[0][l] += vn - vo;
        }
      }
    }
    double* bv = js.bvalues + (size_t)w * S.nb * 3;
#pragma unroll
    for (int l = 0; l < PQA_MAXBAS; ++l) {
      if (l < S.nb) {
        const double d0 = wave_sum(diff[0][l]), d1 = wave_sum(diff[1][l]);
        if (lane == 0) { bv[l * 3 + edown] += d0; bv[l * 3 + edown + 1] += d1; }
      }
    }
  }
  for (int I = lane; I < S.natom; I += 64) {
    const double ax = S.atom_xyz[3 * I], ay = S.atom_xyz[3 * I + 1], az = S.atom_xyz[3 * I + 2];
    const double rn = sqrt((rx - ax) * (rx - ax) + (ry - ay) * (ry - ay) + (rz - az) * (rz - az));
    const double ro = sqrt((ox - ax) * (ox - ax) + (oy - ay) * (oy - ay) + (oz - az) * (oz - az));
    double* av = js.avalues + ((size_t)w * S.natom + I) * S.na * 2;
    for (int k = 0; k < S.na; ++k) {
      double vn = 0.0, vo = 0.0, t1, t2;
      if (rn < S.rcut_a) jas_radial(S.a_kind[k], S.a_param[k], S.rcut_a, rn, vn, t1, t2);
      if (ro < S.rcut_a) jas_radial(S.a_kind[k], S.a_param[k], S.rcut_a, ro, vo, t1, t2);
      av[k * 2 + edown] += vn - vo;
    }
  }
  __syncthreads();
  if (lane == 0) { xw[3 * e] = rx; xw[3 * e + 1] = ry; xw[3 * e + 2] = rz; }
  __syncthreads();
}

// U = sum bvalues*bcoeff + sum avalues*acoeff  (jastrowspin.py:251-255)
__device__ __forceinline__ double jas_value_wave(const SysDev& S, const JastrowState& js, long w) {
  const int lane = threadIdx.x & 63;
  double u = 0.0;
  const double* bv = js.bvalues + (size_t)w * S.nb * 3;
  for (int i = lane; i < S.nb * 3; i += 64) u += bv[i] * S.bcoeff[i];
  const double* av = js.avalues + (size_t)w * S.natom * S.na * 2;
  for (int i = lane; i < S.natom * S.na * 2; i += 64) u += av[i] * S.acoeff[i];
  return wave_sum(u);
}

__global__ __launch_bounds__(64) void k_jastrow_value(SysDev S, JastrowState js, double* out) {
  const double u = jas_value_wave(S, js, blockIdx.x);
  if (threadIdx.x == 0) out[blockIdx.x] = u;
}

// _avalues / _bvalues from scratch for walker w (jastrowspin.py:80-105)
__global__ __launch_bounds__(64) void k_jastrow_recompute(SysDev S, JastrowState js) {
  const long w = blockIdx.x;
  const int lane = threadIdx.x;
  const double* xw = js.x + (size_t)w * S.nelec * 3;
  for (int I = lane; I < S.natom; I += 64) {
    const double ax = S.atom_xyz[3 * I], ay = S.atom_xyz[3 * I + 1], az = S.atom_xyz[3 * I + 2];
    double* av = js.avalues + ((size_t)w * S.natom + I) * S.na * 2;
    for (int k = 0; k < S.na; ++k) {
      double sum[2] = {0.0, 0.0};
      for (int e = 0; e < S.nelec; ++e) {
        const double dx = xw[3 * e] - ax, dy = xw[3 * e + 1] - ay, dz = xw[3 * e + 2] - az;
        const double r = sqrt(dx * dx + dy * dy + dz * dz);
        if (r < S.rcut_a) {
          double v, t1, t2;
          jas_radial(S.a_kind[k], S.a_param[k], S.rcut_a, r, v, t1, t2);
          sum[e >= S.nup] += v;
        }
      }
      av[k * 2] = sum[0];
      av[k * 2 + 1] = sum[1];
    }
  }
  double acc[3][PQA_MAXBAS];
#pragma unroll
  for (int l = 0; l < PQA_MAXBAS; ++l) acc[0][l] = acc[1][l] = acc[2][l] = 0.0;
  for (int i = 0; i < S.nelec; ++i) {
    const double ix = xw[3 * i], iy = xw[3 * i + 1], iz = xw[3 * i + 2];
    for (int j = i + 1 + lane; j < S.nelec; j += 64) {
      const double dx = ix - xw[3 * j], dy = iy - xw[3 * j + 1], dz = iz - xw[3 * j + 2];
      const double r = sqrt(dx * dx + dy * dy + dz * dz);
      if (r < S.rcut_b) {
        const int t = (i >= S.nup) + (j >= S.nup);  // 0 upup, 1 updown, 2 downdown
#pragma unroll
        for (int l = 0; l < PQA_MAXBAS; ++l) {
          if (l < S.nb) {
            double v, t1, t2;
            jas_radial(S.b_kind[l], S.b_param[l], S.rcut_b, r, v, t1, t2);
            if (t == 0) acc[0][l] += v; else if (t == 1) acc[1][l] += v; else acc[2][l] += v;
          }
        }
      }
    }
  }
  double* bv = js.bvalues + (size_t)w * S.nb * 3;
#pragma unroll
  for (int l = 0; l < PQA_MAXBAS; ++l) {
    if (l < S.nb) {
      const double s0 = wave_sum(acc[0][l]), s1 = wave_sum(acc[1][l]), s2 = wave_sum(acc[2][l]);
      if (lane == 0) { bv[l * 3] = s0; bv[l * 3 + 1] = s1; bv[l * 3 + 2] = s2; }
    }
  }
}

// Protocol evaluations at caller-supplied points.  pts (nrow, npt, 3); walker of row r = widx[r] or r.
// mode 0: out[r*npt+q] = exp(U_e(pt) - U_e(x_e))                       (testvalue)
// mode 1: out (4,nrow): grad U_e(pt), exp(U_e(pt) - U_e(x_e))          (gradient_value)
// mode 2: out (4,nrow): grad U_e(pt), lap U_e + |grad U_e|^2           (gradient_laplacian)
__global__ __launch_bounds__(64) void k_jastrow_eval(SysDev S, JastrowState js, int e, const double* __restrict__ pts,
                                                     long nrow, int npt, const int* __restrict__ widx, int mode,
                                                     double* __restrict__ out) {
  const long r = blockIdx.x;
  const long w = widx ? widx[r] : r;
  const double* xw = js.x + (size_t)w * S.nelec * 3;
  double g[3], lp, U0 = 0.0, U;
  if (mode <= 1) jas_eval<0>(S, xw, e, xw[3 * e], xw[3 * e + 1], xw[3 * e + 2], U0, g, lp);
  for (int q = 0; q < npt; ++q) {
    const double* p = pts + (size_t)(r * npt + q) * 3;
    if (mode == 0) {
      jas_eval<0>(S, xw, e, p[0], p[1], p[2], U, g, lp);
      if (threadIdx.x == 0) out[r * npt + q] = exp(U - U0);
    } else if (mode == 1) {
      jas_eval<1>(S, xw, e, p[0], p[1], p[2], U, g, lp);
      if (threadIdx.x == 0) { out[r] = g[0]; out[nrow + r] = g[1]; out[2 * nrow + r] = g[2]; out[3 * nrow + r] = exp(U - U0); }
    } else {
      jas_eval<2>(S, xw, e, p[0], p[1], p[2], U, g, lp);
      if (threadIdx.x == 0) {
        out[r] = g[0]; out[nrow + r] = g[1]; out[2 * nrow + r] = g[2];
        out[3 * nrow + r] = lp + g[0] * g[0] + g[1] * g[1] + g[2] * g[2];
      }
    }
  }
}

__global__ __launch_bounds__(64) void k_jastrow_update(SysDev S, JastrowState js, int e, const double* __restrict__ epos,
                                                       const uint8_t* __restrict__ mask) {
  const long w = blockIdx.x;
  if (mask && !mask[w]) return;
  jas_commit(S, js, w, e, epos[3 * w], epos[3 * w + 1], epos[3 * w + 2]);
}
