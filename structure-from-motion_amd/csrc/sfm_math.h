// sfm_math.h — per-observation arithmetic shared by every kernel (device) and by the host-side
// argument checks.  Expressions follow SURVEY.md Appendix A, which restates the reference:
//   R(q)                 utils.py:83-91
//   verify_rotation_mat  utils.py:101-105
//   q(R)                 utils.py:47-56
//   Jp = [J_C | J_R J_q] campose_processor.py:462-482, 636-808
//   Jx                   triangulation_processor.py:261-269
// All float64.  R is row-major R[3*i+j].
#pragma once

#include <hip/hip_runtime.h>

#include "../../include/sfm_hip.h"

#define SFM_HD __host__ __device__ __forceinline__

namespace sfm {

constexpr double kRotTol = 1e-8;   // utils.py:102
constexpr double kQwMin = 1e-6;    // utils.py:49

// Camera block expanded once per iteration (kernel ba_cam_prep): 19 doubles.
struct CamPrep {
  double C[3];   // centre
  double R[9];   // R(q), row-major
  double t[3];   // R^T (-C): last column of the K-free projection [R^T | t] (ba_processor.py:328)
  double q[4];   // canonical quaternion re-derived from R (campose_processor.py:464, quirk Q7)
};

SFM_HD void quat_to_rot(const double* q, double* R) {
  const double w = q[0], x = q[1], y = q[2], z = q[3];
  R[0] = 1 - 2 * (z * z) - 2 * (y * y);
  R[1] = -2 * z * w + 2 * y * x;
  R[2] = 2 * y * w + 2 * z * x;
  R[3] = 2 * x * y + 2 * w * z;
  R[4] = 1 - 2 * (z * z) - 2 * (x * x);
  R[5] = 2 * z * y - 2 * x * w;
  R[6] = 2 * x * z - 2 * w * y;
  R[7] = 2 * y * z + 2 * w * x;
  R[8] = 1 - 2 * (y * y) - 2 * (x * x);
}

SFM_HD double det3(const double* m) {
  return m[0] * (m[4] * m[8] - m[5] * m[7]) - m[1] * (m[3] * m[8] - m[5] * m[6]) +
         m[2] * (m[3] * m[7] - m[4] * m[6]);
}

// The reference's one-sided predicate: false if det(R) - 1 >= 1e-8 or any (inv(R) - R^T) > 1e-8.
SFM_HD bool verify_rotation(const double* R) {
  const double d = det3(R);
  if (d - 1 >= kRotTol) return false;            // NaN passes, exactly as the NumPy comparison does
  const double id = 1.0 / d;
  // inverse by adjugate: inv[i][j] = cof[j][i] / det
  double inv[9];
  inv[0] = (R[4] * R[8] - R[5] * R[7]) * id;
  inv[1] = (R[2] * R[7] - R[1] * R[8]) * id;
  inv[2] = (R[1] * R[5] - R[2] * R[4]) * id;
  inv[3] = (R[5] * R[6] - R[3] * R[8]) * id;
  inv[4] = (R[0] * R[8] - R[2] * R[6]) * id;
  inv[5] = (R[2] * R[3] - R[0] * R[5]) * id;
  inv[6] = (R[3] * R[7] - R[4] * R[6]) * id;
  inv[7] = (R[1] * R[6] - R[0] * R[7]) * id;
  inv[8] = (R[0] * R[4] - R[1] * R[3]) * id;
  for (int i = 0; i < 3; ++i)
    for (int j = 0; j < 3; ++j)
      if (inv[3 * i + j] - R[3 * j + i] > kRotTol) return false;
  return true;
}

// R -> canonical quaternion (qw >= 0).  Returns SFM_OK or the status the reference would raise.
SFM_HD int rot_to_quat(const double* R, double* q) {
  if (!verify_rotation(R)) return SFM_E_BAD_ROTATION;
  const double tr1 = 1 + R[0] + R[4] + R[8];
  if (tr1 < 0) return SFM_E_SQRT_DOMAIN;
  const double qw = sqrt(tr1) / 2.0;
  if (fabs(qw) < kQwMin) return SFM_E_QW_ZERO;
  q[0] = qw;
  q[1] = (R[7] - R[5]) / (4 * qw);
  q[2] = (R[2] - R[6]) / (4 * qw);
  q[3] = (R[3] - R[1]) / (4 * qw);
  return SFM_OK;
}

// Expand a camera block [C, q] (q as stored, not re-normalised here).  Status as the reference's
// convert_quaternion_to_rotation (ba_processor.py:323) followed by convert_rotation_to_quaternion
// (campose_processor.py:464) would raise.
SFM_HD int cam_prepare(const double* cam7, CamPrep* out) {
  out->C[0] = cam7[0]; out->C[1] = cam7[1]; out->C[2] = cam7[2];
  quat_to_rot(cam7 + 3, out->R);
  int st = rot_to_quat(out->R, out->q);
  const double* R = out->R;
  for (int j = 0; j < 3; ++j)   // t = R^T @ (-C)
    out->t[j] = R[0 + j] * -cam7[0] + R[3 + j] * -cam7[1] + R[6 + j] * -cam7[2];
  return st;
}

// Same, from (R, C) given directly (PnP iteration 0 uses R0 itself, campose_processor.py:367).
SFM_HD int cam_prepare_rc(const double* R, const double* C, CamPrep* out) {
  for (int i = 0; i < 9; ++i) out->R[i] = R[i];
  for (int i = 0; i < 3; ++i) out->C[i] = C[i];
  int st = rot_to_quat(out->R, out->q);
  for (int j = 0; j < 3; ++j)
    out->t[j] = R[0 + j] * -C[0] + R[3 + j] * -C[1] + R[6 + j] * -C[2];
  return st;
}

// Camera-frame point p = [R^T | t] @ [X,Y,Z,W]  (campose_processor.py:727-728).
SFM_HD void project_cam(const CamPrep& c, double X, double Y, double Z, double W, double* p) {
  const double* R = c.R;
  p[0] = R[0] * X + R[3] * Y + R[6] * Z + c.t[0] * W;
  p[1] = R[1] * X + R[4] * Y + R[7] * Z + c.t[1] * W;
  p[2] = R[2] * X + R[5] * Y + R[8] * Z + c.t[2] * W;
}

// Jp (2x7, row-major Jp[7*row + col]) = [J_C | J_R J_q] for one (camera, point), given iz = 1/pz.
// The reference forms J_R (2x9, campose_processor.py:742-766) and J_q (9x4, campose:654-700) and
// multiplies them; here the product is factored as
//   Jp_q[0][k] = (pz A_k - px C_k) / pz^2,  Jp_q[1][k] = (pz B_k - py C_k) / pz^2,
//   (A,B,C)_k = sum_i d_i dR_{i0,i1,i2}/dq_k   with d = X - C (campose:735),
// which is the same sum with the 11 structural zeros of J_q dropped (52 FMA-class ops instead of 96).
SFM_HD void jac_cam_iz(const CamPrep& c, double X, double Y, double Z, const double* p, double iz, int quirks,
                       double* Jp) {
  const double* R = c.R;
  const double px = p[0], py = p[1], pz = p[2];
  const double iz2 = iz * iz;
  const double d0 = X - c.C[0], d1 = Y - c.C[1], d2 = Z - c.C[2];
  // J_C (campose_processor.py:798-806); Q2 keeps the reference's sign in the v-row
  const double sgn = (quirks & SFM_Q2_LOC_JAC_SIGN) ? 1.0 : -1.0;
  for (int i = 0; i < 3; ++i) {
    Jp[i] = (px * R[3 * i + 2] - pz * R[3 * i + 0]) * iz2;
    Jp[7 + i] = (-pz * R[3 * i + 1] - py * (sgn * R[3 * i + 2])) * iz2;
  }
  const double w2 = 2 * c.q[0], x2 = 2 * c.q[1], y2 = 2 * c.q[2], z2 = 2 * c.q[3];
  const double x4 = 4 * c.q[1], y4 = 4 * c.q[2], z4 = 4 * c.q[3];
  double A[4], B[4], C[4];
  A[0] = d1 * z2 - d2 * y2;            B[0] = d2 * x2 - d0 * z2;            C[0] = d0 * y2 - d1 * x2;
  A[1] = d1 * y2 + d2 * z2;            B[1] = d0 * y2 - d1 * x4 + d2 * w2;  C[1] = d0 * z2 - d1 * w2 - d2 * x4;
  A[2] = d1 * x2 - d0 * y4 - d2 * w2;  B[2] = d0 * x2 + d2 * z2;            C[2] = d0 * w2 + d1 * z2 - d2 * y4;
  A[3] = d1 * w2 - d0 * z4 + d2 * x2;  B[3] = d2 * y2 - d0 * w2 - d1 * z4;  C[3] = d0 * x2 + d1 * y2;
  for (int k = 0; k < 4; ++k) {
    Jp[3 + k] = (pz * A[k] - px * C[k]) * iz2;
    Jp[10 + k] = (pz * B[k] - py * C[k]) * iz2;
  }
}

// The u-row of Jp alone (Jp[0..6]): all that the reference's PnP keeps of every point but the last one under its row-stacking
// quirk Q1 (campose_processor.py:404-405) -- the B sums and the seven v-row entries are not formed.
SFM_HD void jac_cam_iz_urow(const CamPrep& c, double X, double Y, double Z, const double* p, double iz, double* Jp) {
  const double* R = c.R;
  const double px = p[0], pz = p[2];
  const double iz2 = iz * iz;
  const double d0 = X - c.C[0], d1 = Y - c.C[1], d2 = Z - c.C[2];
  for (int i = 0; i < 3; ++i) Jp[i] = (px * R[3 * i + 2] - pz * R[3 * i + 0]) * iz2;
  const double w2 = 2 * c.q[0], x2 = 2 * c.q[1], y2 = 2 * c.q[2], z2 = 2 * c.q[3];
  const double x4 = 4 * c.q[1], y4 = 4 * c.q[2], z4 = 4 * c.q[3];
  double A[4], C[4];
  A[0] = d1 * z2 - d2 * y2;            C[0] = d0 * y2 - d1 * x2;
  A[1] = d1 * y2 + d2 * z2;            C[1] = d0 * z2 - d1 * w2 - d2 * x4;
  A[2] = d1 * x2 - d0 * y4 - d2 * w2;  C[2] = d0 * w2 + d1 * z2 - d2 * y4;
  A[3] = d1 * w2 - d0 * z4 + d2 * x2;  C[3] = d0 * x2 + d1 * y2;
  for (int k = 0; k < 4; ++k) Jp[3 + k] = (pz * A[k] - px * C[k]) * iz2;
}

SFM_HD void jac_cam(const CamPrep& c, double X, double Y, double Z, const double* p, int quirks,
                    double* Jp) {
  jac_cam_iz(c, X, Y, Z, p, 1.0 / p[2], quirks, Jp);
}

// Jx (2x3, row-major) for a general 3x4 projection P (row-major P[4*i+j]) and s = P @ X~
// (triangulation_processor.py:261-269).
SFM_HD void jac_pt(const double* P, const double* s, double* Jx) {
  const double iz2 = 1.0 / (s[2] * s[2]);
  for (int j = 0; j < 3; ++j) {
    Jx[j] = (s[2] * P[j] - s[0] * P[8 + j]) * iz2;
    Jx[3 + j] = (s[2] * P[4 + j] - s[1] * P[8 + j]) * iz2;
  }
}

// Jx for the K-free projection [R^T | t] of a prepared camera (ba_processor.py:328, 333), iz = 1/pz.
SFM_HD void jac_pt_cam_iz(const CamPrep& c, const double* p, double iz, double* Jx) {
  const double* R = c.R;
  const double iz2 = iz * iz;
  for (int j = 0; j < 3; ++j) {
    Jx[j] = (p[2] * R[3 * j + 0] - p[0] * R[3 * j + 2]) * iz2;
    Jx[3 + j] = (p[2] * R[3 * j + 1] - p[1] * R[3 * j + 2]) * iz2;
  }
}

}  // namespace sfm
