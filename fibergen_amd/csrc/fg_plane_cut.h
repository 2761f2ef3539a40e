// Volume fraction of a box cut by a plane, closed form and unconditionally stable (host + device).
// Used by the GPU voxeliser (fg_voxelize.hip); tests/emulate/emu_cut.cpp compiles it for the host, where it is checked
// against exact rational arithmetic (tests/test_plane_cut.py).
#pragma once

#include <cmath>

#include "fg_common.h"

namespace fg {

// Fraction of the box [0, d]^3 (edges d = (dx, dy, dz)) on the side  n . (y - xs) < 0  of the plane through xs (given
// relative to the box origin) with unit normal n.  After mirroring the axes with negative normal components the region
// is  sum_i m_i y_i < alpha  with m_i = |n_i| >= 0; with A_i = m_i d_i sorted ascending the fraction is a repeated
// moving average of a ramp,
//     V1(u) = clamp(u / A3, 0, 1),   V2(s) = mean of V1 over [s - A2, s],   V3(t) = mean of V2 over [t - A1, t],
// (an A_i below the spacing of the floating-point numbers at its argument: the mean is the point value).  V1 is piecewise linear with kinks at {0, A3}, V2 piecewise quadratic
// with kinks at {0, A2, A3, A2 + A3}: each mean is taken piece by piece with the rule that is exact on the piece
// (trapezoid / Simpson).  Every term is non-negative -- no cancellation however small a normal component is, where
// the expanded polynomial  [a^3 - sum (a - A_i)_+^3 + sum (a - A_i - A_j)_+^3] / (6 A1 A2 A3)  loses all digits.
FG_HD double clampd(double x, double lo, double hi) { return x < lo ? lo : (x > hi ? hi : x); }

FG_HD double cut_v1(double u, double A3) { return u <= 0.0 ? 0.0 : (u >= A3 ? 1.0 : u / A3); }

FG_HD double cut_v2(double s, double A2, double A3) {
  const double lo = s - A2, h = s - lo;   // h: the width that exists in floating point (weights then sum to one)
  if (!(h > 0.0)) return cut_v1(s, A3);
  if (s <= 0.0) return 0.0;
  if (lo >= A3) return 1.0;
  const double brk[4] = {lo, 0.0, A3, s};   // pieces of [lo, s] between the kinks of V1
  double v = 0.0, a = lo;
  for (int k = 1; k < 4; ++k) {
    const double b = clampd(brk[k], lo, s);
    if (b > a) {
      v += ((b - a) / h) * 0.5 * (cut_v1(a, A3) + cut_v1(b, A3));
      a = b;
    }
  }
  return v;
}

FG_HD double cut_v3(double t, double A1, double A2, double A3) {
  const double lo = t - A1, h = t - lo;
  if (!(h > 0.0)) return cut_v2(t, A2, A3);
  if (t <= 0.0) return 0.0;
  if (lo >= A2 + A3) return 1.0;
  const double brk[6] = {lo, 0.0, A2, A3, A2 + A3, t};   // A2 <= A3: ascending kinks of V2
  double v = 0.0, a = lo;
  for (int k = 1; k < 6; ++k) {
    const double b = clampd(brk[k], lo, t);
    if (b > a) {
      const double m = 0.5 * (a + b);
      v += ((b - a) / h) * ((cut_v2(a, A2, A3) + 4.0 * cut_v2(m, A2, A3) + cut_v2(b, A2, A3)) / 6.0);
      a = b;
    }
  }
  return v;
}

// xs_rel: a point of the plane relative to the box origin, n: unit normal, d: box edges
FG_HD double box_fraction_below_plane(const double xs_rel[3], const double n[3], const double d[3]) {
  double A[3] = {std::fabs(n[0]) * d[0], std::fabs(n[1]) * d[1], std::fabs(n[2]) * d[2]};
  double alpha = n[0] * xs_rel[0] + n[1] * xs_rel[1] + n[2] * xs_rel[2];
  for (int i = 0; i < 3; ++i)
    if (n[i] < 0) alpha -= n[i] * d[i];
  if (A[0] > A[1]) { const double t = A[0]; A[0] = A[1]; A[1] = t; }
  if (A[1] > A[2]) { const double t = A[1]; A[1] = A[2]; A[2] = t; }
  if (A[0] > A[1]) { const double t = A[0]; A[0] = A[1]; A[1] = t; }
  if (!(alpha > 0.0)) return 0.0;
  if (alpha >= A[0] + A[1] + A[2]) return 1.0;
  const double v = cut_v3(alpha, A[0], A[1], A[2]);
  return v < 0.0 ? 0.0 : (v > 1.0 ? 1.0 : v);
}

}  // namespace fg
