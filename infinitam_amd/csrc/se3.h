// se3.h -- rigid-motion algebra for the host side of the ICP tracker, in double precision.
//
// The tracker optimises over a 6-vector (translation part v, rotation part w) tied to the model-view matrix by the SE(3)
// exponential: M = [ exp([w]x)  V(w) v ; 0 1 ].  The reference keeps both representations in ITMPose and converts with
// closed forms in float (Objects/ITMPose.cpp:84-236); what has to be matched here is the FUNCTION, not its evaluation
// order -- the tracker's parity bar is 2e-5 on the pose.  This file is an independent implementation:
//   * everything in double, matrices row-major;
//   * log: rotation angle from atan2(|a|, c) with a = vee(R - R^T)/2 and c = (tr R - 1)/2 -- for the slightly
//     non-orthonormal matrices the update produces, a and c are what the reference's formulas read as well, so the
//     projection onto SO(3) agrees to first order -- and the axis near pi from the dominant column of the symmetric part;
//   * translation part through the closed-form inverse of V(w) (coefficient 1/12 + theta^2/720 + ... near zero);
//   * exp: Rodrigues with series for the three coefficient functions below theta = 1e-3.
#pragma once

#include <cmath>

namespace itm {
namespace se3 {

struct Rigid {            // x_out = R x_in + t
  double R[9];            // row-major
  double t[3];
};
struct Twist { double v[3], w[3]; };

inline void cross(const double* a, const double* b, double* o) {
  o[0] = a[1] * b[2] - a[2] * b[1];
  o[1] = a[2] * b[0] - a[0] * b[2];
  o[2] = a[0] * b[1] - a[1] * b[0];
}
inline double dot(const double* a, const double* b) { return a[0] * b[0] + a[1] * b[1] + a[2] * b[2]; }

// column-major float[16] (ORUtils::Matrix4 storage, m[col * 4 + row]) <-> Rigid
inline Rigid from_matrix(const float* m) {
  Rigid g;
  for (int r = 0; r < 3; ++r) {
    for (int c = 0; c < 3; ++c) g.R[3 * r + c] = (double)m[4 * c + r];
    g.t[r] = (double)m[12 + r];
  }
  return g;
}
inline void to_matrix(const Rigid& g, float* m) {
  for (int r = 0; r < 3; ++r) {
    for (int c = 0; c < 3; ++c) m[4 * c + r] = (float)g.R[3 * r + c];
    m[12 + r] = (float)g.t[r];
    m[4 * r + 3] = 0.0f;
  }
  m[15] = 1.0f;
}

inline Rigid compose(const Rigid& a, const Rigid& b) {   // a after b
  Rigid g;
  for (int r = 0; r < 3; ++r) {
    for (int c = 0; c < 3; ++c) g.R[3 * r + c] = a.R[3 * r] * b.R[c] + a.R[3 * r + 1] * b.R[3 + c] + a.R[3 * r + 2] * b.R[6 + c];
    g.t[r] = a.R[3 * r] * b.t[0] + a.R[3 * r + 1] * b.t[1] + a.R[3 * r + 2] * b.t[2] + a.t[r];
  }
  return g;
}

// inverse of an affine map whose linear part need not be orthonormal (adjugate / determinant); false if singular
inline bool invert(const Rigid& g, Rigid& out) {
  const double* a = g.R;
  const double c00 = a[4] * a[8] - a[5] * a[7], c01 = a[5] * a[6] - a[3] * a[8], c02 = a[3] * a[7] - a[4] * a[6];
  const double det = a[0] * c00 + a[1] * c01 + a[2] * c02;
  if (det == 0.0 || !std::isfinite(det)) return false;
  const double k = 1.0 / det;
  out.R[0] = c00 * k; out.R[1] = (a[2] * a[7] - a[1] * a[8]) * k; out.R[2] = (a[1] * a[5] - a[2] * a[4]) * k;
  out.R[3] = c01 * k; out.R[4] = (a[0] * a[8] - a[2] * a[6]) * k; out.R[5] = (a[2] * a[3] - a[0] * a[5]) * k;
  out.R[6] = c02 * k; out.R[7] = (a[1] * a[6] - a[0] * a[7]) * k; out.R[8] = (a[0] * a[4] - a[1] * a[3]) * k;
  for (int r = 0; r < 3; ++r) out.t[r] = -(out.R[3 * r] * g.t[0] + out.R[3 * r + 1] * g.t[1] + out.R[3 * r + 2] * g.t[2]);
  return true;
}

// coefficient functions of the exponential: sin(x)/x, (1 - cos x)/x^2, (x - sin x)/x^3
inline void exp_coefficients(double th, double& A, double& B, double& C) {
  const double t2 = th * th;
  if (th < 1e-3) {
    A = 1.0 - t2 / 6.0 * (1.0 - t2 / 20.0);
    B = 0.5 - t2 / 24.0 * (1.0 - t2 / 30.0);
    C = 1.0 / 6.0 - t2 / 120.0 * (1.0 - t2 / 42.0);
  } else {
    const double s = std::sin(th), c = std::cos(th);
    A = s / th;
    B = (1.0 - c) / t2;
    C = (th - s) / (t2 * th);
  }
}

inline Rigid exp(const Twist& x) {
  const double* w = x.w;
  const double th = std::sqrt(dot(w, w));
  double A, B, C;
  exp_coefficients(th, A, B, C);
  Rigid g;
  // R = I + A [w]x + B [w]x^2, with [w]x^2 = w w^T - |w|^2 I
  const double ww = th * th;
  for (int r = 0; r < 3; ++r)
    for (int c = 0; c < 3; ++c) g.R[3 * r + c] = B * w[r] * w[c] + ((r == c) ? (1.0 - B * ww) : 0.0);
  g.R[1] -= A * w[2]; g.R[2] += A * w[1];
  g.R[3] += A * w[2]; g.R[5] -= A * w[0];
  g.R[6] -= A * w[1]; g.R[7] += A * w[0];
  // t = V v = v + B (w x v) + C (w x (w x v))
  double wv[3], wwv[3];
  cross(w, x.v, wv); cross(w, wv, wwv);
  for (int i = 0; i < 3; ++i) g.t[i] = x.v[i] + B * wv[i] + C * wwv[i];
  return g;
}

inline Twist log(const Rigid& g) {
  const double* R = g.R;
  Twist x;
  double a[3] = {0.5 * (R[7] - R[5]), 0.5 * (R[2] - R[6]), 0.5 * (R[3] - R[1])};   // vee of the antisymmetric part
  const double c = 0.5 * (R[0] + R[4] + R[8] - 1.0);
  const double s = std::sqrt(dot(a, a));
  const double th = std::atan2(s, c);
  if (c > -0.70710678118654752) {
    const double k = (s > 1e-12) ? th / s : 1.0;      // th/s -> 1 as the angle vanishes
    for (int i = 0; i < 3; ++i) x.w[i] = k * a[i];
  } else {
    // near pi the antisymmetric part vanishes: take the axis from the symmetric part S = (R + R^T)/2 - c I = (1 - c) n n^T
    double S[9];
    for (int r = 0; r < 3; ++r)
      for (int q = 0; q < 3; ++q) S[3 * r + q] = 0.5 * (R[3 * r + q] + R[3 * q + r]) - ((r == q) ? c : 0.0);
    int j = 0;
    if (std::fabs(S[4]) > std::fabs(S[3 * j + j])) j = 1;
    if (std::fabs(S[8]) > std::fabs(S[3 * j + j])) j = 2;
    double n[3] = {S[j], S[3 + j], S[6 + j]};
    if (dot(n, a) < 0.0) { n[0] = -n[0]; n[1] = -n[1]; n[2] = -n[2]; }
    const double len = std::sqrt(dot(n, n));
    for (int i = 0; i < 3; ++i) x.w[i] = (len > 0.0) ? th * n[i] / len : 0.0;
  }
  // v = V^-1 t = t - w x t / 2 + kappa w x (w x t),  kappa = (1 - (theta/2) cot(theta/2)) / theta^2
  const double t2 = dot(x.w, x.w), theta = std::sqrt(t2);
  double kappa;
  if (theta < 1e-3) kappa = 1.0 / 12.0 + t2 / 720.0 + t2 * t2 / 30240.0;
  else kappa = (1.0 - 0.5 * theta * std::cos(0.5 * theta) / std::sin(0.5 * theta)) / t2;
  double wt[3], wwt[3];
  cross(x.w, g.t, wt); cross(x.w, wt, wwt);
  for (int i = 0; i < 3; ++i) x.v[i] = g.t[i] - 0.5 * wt[i] + kappa * wwt[i];
  return x;
}

// Symmetric positive definite solve A x = b for n <= 6 through A = L D L^T (no square roots); `a` is row-major with row
// stride `lda`.  Returns false (x = 0) when a pivot is not positive, i.e. the system carries no information.
inline bool solve_spd(const double* a, int lda, int n, const double* b, double* x) {
  double L[36], D[6], y[6];
  for (int j = 0; j < n; ++j) {
    double d = a[j * lda + j];
    for (int k = 0; k < j; ++k) d -= L[j * 6 + k] * L[j * 6 + k] * D[k];
    if (!(d > 0.0) || !std::isfinite(d)) { for (int i = 0; i < n; ++i) x[i] = 0.0; return false; }
    D[j] = d;
    for (int i = j + 1; i < n; ++i) {
      double v = a[i * lda + j];
      for (int k = 0; k < j; ++k) v -= L[i * 6 + k] * L[j * 6 + k] * D[k];
      L[i * 6 + j] = v / d;
    }
  }
  for (int i = 0; i < n; ++i) { double v = b[i]; for (int k = 0; k < i; ++k) v -= L[i * 6 + k] * y[k]; y[i] = v; }
  for (int i = n - 1; i >= 0; --i) { double v = y[i] / D[i]; for (int k = i + 1; k < n; ++k) v -= L[k * 6 + i] * x[k]; x[i] = v; }
  return true;
}

}  // namespace se3
}  // namespace itm
