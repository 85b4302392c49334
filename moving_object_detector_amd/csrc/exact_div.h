// exact_div.h — correctly rounded F64 quotients with fewer instructions than one IEEE division each.
//
// The scene-flow kernel is VALU-bound and spends most of its F64 work in three places where a division has structure:
//   * project3dToPixel divides two numerators by the same tz   (scene_flow_constructor.cpp:84, image_geometry)
//   * the velocity divides three differences by the frame's dt  (scene_flow_constructor.cpp:200-202)
// Both keep the reference's result bit for bit; what changes is how the correctly rounded quotient is reached.
#pragma once
#if defined(__HIPCC__)
#include <hip/hip_runtime.h>
#endif
#include <math.h>
#include <stdint.h>

#if defined(__HIPCC__)
#define ED_HD __host__ __device__ __forceinline__
#else
#define ED_HD inline
#endif

namespace exact_div {

// |v| in [2^-200, 2^200]: no operand scaling, no overflow/underflow anywhere in the sequences below.
ED_HD bool in_window(double v) { const double a = fabs(v); return (a >= 0x1p-200) & (a <= 0x1p200); }

// ---- n / d for a d whose correctly rounded reciprocal rd = RN(1/d) is known (computed once per frame on the host) ----
// Markstein's division step: q0 = RN(n rd) is within one ulp of n/d, rem = n - d q0 is exact in an FMA, and
// RN(q0 + rem rd) is the correctly rounded quotient.  Requires d in the window and a numerator widened from F32
// (|n| in [2^-149, 2^128) when it is an ordinary number).  `ordinary` = n is neither 0, inf nor NaN; those take n * rd,
// which has the sign and class of n / d (the correction step would lose the sign of a zero and turn inf into NaN).
// tests/cpp/exact_div_test.cpp checks the step against `/` on random and edge operands.
ED_HD bool reciprocal_usable(double d) { return in_window(d); }
ED_HD double div_by_known(double n, bool ordinary, double d, double rd) {
  const double q0 = n * rd;
  const double rem = fma(-d, q0, n);
  const double q = fma(rem, rd, q0);
  return ordinary ? q : q0;
}

#if defined(__HIPCC__)
// ---- a / d and b / d on the device, sharing the reciprocal refinement ----
// This is the instruction sequence the compiler emits for one F64 division on gfx9 (v_div_scale, v_rcp_f64, two Newton
// steps, quotient, remainder, v_div_fmas, v_div_fixup) with the scaling and fix-up steps removed, which are the identity
// when d, a, b are in the window.  A zero numerator keeps the plain product (a correctly signed zero).  The return value
// tells the caller whether the precondition held; if not, the caller divides with `/`.
__device__ __forceinline__ bool div2_shared(double a, double b, double d, double &qa, double &qb) {
  double r = __builtin_amdgcn_rcp(d);
  double e = fma(-d, r, 1.0);
  r = fma(r, e, r);
  e = fma(-d, r, 1.0);
  r = fma(r, e, r);
  const double q0 = a * r, q1 = b * r;
  const double ra = fma(-d, q0, a), rb = fma(-d, q1, b);
  const bool za = a == 0.0, zb = b == 0.0;
  qa = za ? q0 : fma(ra, r, q0);
  qb = zb ? q1 : fma(rb, r, q1);
  return in_window(d) & (in_window(a) | za) & (in_window(b) | zb);
}
#endif

}  // namespace exact_div
