// short_div.h -- correctly rounded fp64 division and square root for operands whose RANGE is known (trace_kernel.hip; also compiled
// into tools/micro/short_div.hip, which compares them bit for bit with the compiler's own sequences on the GPU).
#pragma once
#include <hip/hip_runtime.h>

namespace rayhs {
#ifndef DEV
#define DEV static __device__ __forceinline__
#endif

// IEEE division and square root WITHOUT their range fix-ups, for operands whose range is known.
// The compiler's correctly rounded  n / d  is eleven vector instructions: v_div_scale x 2 (pre-scale operands whose quotient would leave
// the normal range), v_rcp_f64, two Newton steps, the quotient and its residual, v_div_fmas (undo the scaling), v_div_fixup (zeros,
// infinities, NaNs, the sign).  When v_div_scale scales nothing -- d and n normal, 2^-767 < |n / d| < 2^767 and the like: the ISA's
// pseudo code lists the cases -- the three range instructions are identities on everything the other eight compute, so the eight by
// themselves yield THE SAME DOUBLE, bit for bit (a reciprocal needs seven: its quotient n * y is y).  sqrt likewise: ten of eighteen.
// Used where the range is certain: the packet / per-lane walks' triangle tests (|det| >= 1e-6 or the lane is rejected whatever its
// quotients; < 1e110 because begin() hands rays with |o| + |d| >= 1e30 to the reference tree and the launcher hands scenes with vertex
// coordinates beyond 1e90 to the exhaustive kernel) and a point light's arithmetic behind one range test on the squared distance.
// tools/micro/short_div.hip compares them with the compiler's sequences on the GPU (profiles/r06/micro/short_div.txt); -DRAYHS_NO_SHORT_DIV
// builds without them (profiles/r06/experiments/).
DEV double rcp_newton(double x) {
    double y = __builtin_amdgcn_rcp(x);
    double e = __builtin_fma(-x, y, 1.0);
    y = __builtin_fma(y, e, y);
    e = __builtin_fma(-x, y, 1.0);
    return __builtin_fma(y, e, y);
}
DEV double rcp_in_range(double x) {             // 1 / x for 2^-767 <= |x| < 2^1022
    const double y = rcp_newton(x);
    const double r = __builtin_fma(-x, y, 1.0);
    return __builtin_fma(r, y, y);
}
DEV double div_in_range(double n, double d) {   // n / d for normal d, 1 / d and n / d, |n| >= 2^-969, exponents less than 768 apart
    const double y = rcp_newton(d);
    const double q = n * y;
    const double r = __builtin_fma(-d, q, n);
    return __builtin_fma(r, y, q);
}
DEV double sqrt_in_range(double x) {            // sqrt x for 2^-767 <= x < inf
    const double y = __builtin_amdgcn_rsq(x);
    double g = x * y, h = y * 0.5;
    const double r = __builtin_fma(-h, g, 0.5);
    g = __builtin_fma(g, r, g);
    h = __builtin_fma(h, r, h);
    double d = __builtin_fma(-g, g, x);
    g = __builtin_fma(d, h, g);
    d = __builtin_fma(-g, g, x);
    return __builtin_fma(d, h, g);
}

}  // namespace rayhs
