// Exact thresholds on SQUARED distances for the three boolean tests of the particle path.
//
// The reference compares a distance dist = np.sqrt(np.sum(np.square(delta))) against a constant:
//   is_collision   dist < 0.15 + 0.15                      (multi-goal_spread.py:114-118)
//   reached        -dist >= -0.05  <=>  dist <= 0.05       (multi-goal_spread.py:125-129)
//   (build) the soft contact is exactly +-0 for dist >= kSkip (particle.hip, contact_force: 0.32 for the float32 soft-plus on
//           the hardware transcendental units, 0.41 for float32 libm, 1.05 for float64)
// IEEE sqrt is correctly rounded and monotone, so for a constant c of the working precision
//   RN(sqrt(x)) <  c   <=>   x <  T_lt(c)   with T_lt(c) = min{ x : RN(sqrt(x)) >= c }
//   RN(sqrt(x)) <= c   <=>   x <  T_lt(next_up(c))
// (NaN makes both sides false; +inf is on the "not less" side of both.)  Testing x = dx*dx + dy*dy against T takes the
// N(N-1) dependent square roots per env out of the collision pass and the N-1 of the near-neighbour scan without
// changing a single result bit.  The constants are derived with exact rational arithmetic and checked EXHAUSTIVELY over
// all 2^31 non-negative floats (doubles: derivation + a window of neighbours) by tests/test_thresholds.py, which parses
// this file.
#pragma once

namespace cm3 {

template <typename R> struct Thresh;

template <> struct Thresh<float> {
  // c = 0.15f + 0.15f = 0x1.333334p-2f (0.3f):  sqrtf(x) < c  <=>  x < kColl2
  static constexpr float kColl2 = 0x1.70a3d8p-4f;   // CM3_THRESH f32 coll 0.3
  // c = Contact<float>::kSkip:                  sqrtf(x) >= c <=>  x >= kSkip2
#ifndef CM3_F32_LIBM_SOFTPLUS
  static constexpr float kSkip2 = 0x1.a36e2cp-4f;   // CM3_THRESH f32 skip 0.32
#else
  static constexpr float kSkip2 = 0x1.5844dp-3f;    // CM3_THRESH f32 skip_libm 0.41
#endif
  // c = 0.05f:                                  sqrtf(x) <= c <=>  x < kReach2
  static constexpr float kReach2 = 0x1.47ae18p-9f;  // CM3_THRESH f32 reach 0.05
};

template <> struct Thresh<double> {
  static constexpr double kColl2 = 0x1.70a3d70a3d709p-4;   // CM3_THRESH f64 coll 0.3
  static constexpr double kSkip2 = 0x1.1a3d70a3d70a4p+0;   // CM3_THRESH f64 skip 1.05
  static constexpr double kReach2 = 0x1.47ae147ae147dp-9;  // CM3_THRESH f64 reach 0.05
};

}  // namespace cm3
