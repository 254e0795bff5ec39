// Device helpers shared by the translation units of libpnl_hip.so (gfx950 only): kernel function gamma, quadrature
// order formula, hardware fp64 atomics, DPP / permlane wave reductions.
#pragma once
#include "pnl_device.h"

// read-only tables through the constant address space: loads with a wave-uniform address are scalar loads (SGPRs)
typedef const double __attribute__((address_space(4))) *pnl_const_f64_ptr;

// ---------------------------------------------------------------------------------------------
// kernel function gamma(|x-y|^2)   (KC:75-294)
// Branch-free ln and exp for the kernels with a general exponent: d2^e = exp(e ln d2), d2 a positive normal number and
// |e ln d2| far from overflow.  Straight-line code (no special cases), so the independent evaluations of a pair interleave.
// ln x: x = m 2^k with m in [sqrt(1/2), sqrt(2)); ln m = 2 atanh(f), f = (m-1)/(m+1), |f| < 0.172, odd series to f^21.
__device__ __forceinline__ double pnl_log(double x) {
    double m = __builtin_amdgcn_frexp_mant(x);           // [0.5, 1)
    int k = __builtin_amdgcn_frexp_exp(x);
    const bool low = m < 0.70710678118654752440;
    m = low ? m+m : m;
    k = low ? k-1 : k;
    const double a = m-1.0, b = m+1.0;
    double r = __builtin_amdgcn_rcp(b);
    r = __builtin_fma(__builtin_fma(-b, r, 1.0), r, r);
    r = __builtin_fma(__builtin_fma(-b, r, 1.0), r, r);
    double f = a*r;
    f = __builtin_fma(__builtin_fma(-f, b, a), r, f);
    const double z = f*f;
    double p = 1.0/21.0;
    p = __builtin_fma(p, z, 1.0/19.0);
    p = __builtin_fma(p, z, 1.0/17.0);
    p = __builtin_fma(p, z, 1.0/15.0);
    p = __builtin_fma(p, z, 1.0/13.0);
    p = __builtin_fma(p, z, 1.0/11.0);
    p = __builtin_fma(p, z, 1.0/9.0);
    p = __builtin_fma(p, z, 1.0/7.0);
    p = __builtin_fma(p, z, 1.0/5.0);
    p = __builtin_fma(p, z, 1.0/3.0);
    const double lm = __builtin_fma(f+f, z*p, f+f);       // 2 f + 2 f^3 (1/3 + ...)
    const double kd = (double)k;
    return __builtin_fma(kd, 6.93147180369123816490e-01, __builtin_fma(kd, 1.90821492927058770002e-10, lm));
}

// exp y, |y| < 700: y = n ln 2 + r, |r| <= 0.347, Taylor polynomial to r^13, scaled by 2^n
__device__ __forceinline__ double pnl_exp(double y) {
    const double n = __builtin_rint(y*1.44269504088896338700);
    double r = __builtin_fma(-n, 6.93147180369123816490e-01, y);
    r = __builtin_fma(-n, 1.90821492927058770002e-10, r);
    double p = 1.0/6227020800.0;
    p = __builtin_fma(p, r, 1.0/479001600.0);
    p = __builtin_fma(p, r, 1.0/39916800.0);
    p = __builtin_fma(p, r, 1.0/3628800.0);
    p = __builtin_fma(p, r, 1.0/362880.0);
    p = __builtin_fma(p, r, 1.0/40320.0);
    p = __builtin_fma(p, r, 1.0/5040.0);
    p = __builtin_fma(p, r, 1.0/720.0);
    p = __builtin_fma(p, r, 1.0/120.0);
    p = __builtin_fma(p, r, 1.0/24.0);
    p = __builtin_fma(p, r, 1.0/6.0);
    p = __builtin_fma(p, r, 0.5);
    p = __builtin_fma(p, r, 1.0);
    p = __builtin_fma(p, r, 1.0);
    return __builtin_amdgcn_ldexp(p, (int)n);
}

// KT: 0 general (pow / indicator / peridynamic, horizon test), 1 fractional with exponent -qm/4, qm a run-time (wave-uniform)
// value, 2 the same with qm == 6 known at compile time (s = 1/2 in 2D): no branch per evaluation, so the compiler
// interleaves the dependent chains of the independent evaluations of a pair.
template <int KT>
__device__ __forceinline__ double kern_eval(const DevKernel &k, double d2) {
    if (KT == 2) {
        // d2^(-3/2) = r^3 (1 - e)^(-3/2) with r = v_rsq_f64(d2) (~2^-23 relative), e = 1 - d2 r^2 (|e| < 3e-7):
        // r^3 (1 + e (3/2 + 15/8 e)), the next term 35/16 e^3 is below 1e-19; six operations after the rsq, chain depth five
        const double r = __builtin_amdgcn_rsq(d2);
        const double t = r*r;
        const double e = __builtin_fma(-d2, t, 1.0);
        const double g0 = r*t;
        return __builtin_fma(g0, e*__builtin_fma(1.875, e, 1.5), g0);
    } else if (KT == 1) {
        // exponent = -qm/4 (s a multiple of 1/4 in 1D / 2D): d2^(-1/2) from v_rsq_f64 (~2^-23 relative) + one Halley step
        // (cubic: r (1 + e/2 + 3 e^2/8), e = 1 - d2 r^2, five operations for full precision), for odd qm one more refined
        // rsqrt gives d2^(-1/4); then an integer power.  A few ulp instead of libm's pow at 1/8 of the cost.  The scale is
        // applied once per pair (kern_scale); qm is wave-uniform, the branches are scalar.
        double r = __builtin_amdgcn_rsq(d2);
        {
            const double e = __builtin_fma(-(d2*r), r, 1.0);
            r = __builtin_fma(r, e*__builtin_fma(0.375, e, 0.5), r);
        }
        int p = k.qm;
        if (p == 6) return (r*r)*r;                      // s = 1/2 in 2D
        double base = r;
        if (p & 1) {
            double t = __builtin_amdgcn_rsq(r);
            const double e = __builtin_fma(-(r*t), t, 1.0);
            t = __builtin_fma(t, e*__builtin_fma(0.375, e, 0.5), t);
            base = r*t;                                  // d2^(-1/4)
        } else p >>= 1;
        double res = (p & 1) ? base : 1.;
        p >>= 1;
        while (p) {
            base *= base;
            if (p & 1) res *= base;
            p >>= 1;
        }
        return res;
    } else {
        if (!(d2 <= k.horizon2)) return 0.;
        // general exponent: exp(e ln d2) instead of pow (half the instructions; |e ln d2| < 60 keeps the relative error of the
        // product below 1e-14, three orders under the parity tolerance)
        if (k.ktype == 0) return k.scale*pnl_exp(k.exponent*pnl_log(d2));
        if (k.ktype == 1) return k.scale;
        return k.scale/sqrt(d2);
    }
}

template <int KT>
__device__ __forceinline__ double kern_scale(const DevKernel &k) { return KT >= 1 ? k.scale : 1.; }

// distant quadrature order  (FL2:622-642, :1226-1243, FL1:234-253, :646-660)
__device__ __forceinline__ int quad_order(const DevFormula &F, double H0, double h1, double h2, double d) {
    double logdh1 = log(d/h1), logdh2 = log(d/h2);
    double L1 = fabs(log(h1/H0)), L2 = fabs(log(h2/H0));
    double Lm = fmax(L1, L2);
    double n1 = logdh1, n2 = logdh2;
    if (F.clip) { n1 = fmax(logdh1, 0.); n2 = fmax(logdh2, 0.); }
    double p1 = ceil((F.c0 + F.a*L2 + F.b*Lm - F.e*n2)/(fmax(logdh1, 0.) + F.den0));
    double p2 = ceil((F.c0 + F.a*L1 + F.b*Lm - F.e*n1)/(fmax(logdh2, 0.) + F.den0));
    int q1 = (int)fmax(p1, 2.), q2 = (int)fmax(p2, 2.);
    return q1 > q2 ? q1 : q2;
}

// Same order, decided in fp32 where that is safe: the fp32 value of the ceil() argument is off by < 2e-5 (v_log_f32 /
// v_rcp_f32 are 1 ulp, the operands are O(10)), so whenever it is further than 2e-4 from an integer the fp64 formula
// gives the same ceil; otherwise (0.03 % of the pairs) the exact fp64 formula decides.  lh = ln h and L = |ln(h/H0)| per
// cell are staged once per tile in fp32, Ld = |ln(h/H0)| in fp64 for the exact path.
__device__ __forceinline__ int quad_order_exact(const DevFormula &F, double h1, double h2, double Ld1, double Ld2, double d) {
    const double logdh1 = log(d/h1), logdh2 = log(d/h2);
    const double Lm = fmax(Ld1, Ld2);
    double n1 = logdh1, n2 = logdh2;
    if (F.clip) { n1 = fmax(logdh1, 0.); n2 = fmax(logdh2, 0.); }
    const double p1 = ceil((F.c0 + F.a*Ld2 + F.b*Lm - F.e*n2)/(fmax(logdh1, 0.) + F.den0));
    const double p2 = ceil((F.c0 + F.a*Ld1 + F.b*Lm - F.e*n1)/(fmax(logdh2, 0.) + F.den0));
    const int q1 = (int)fmax(p1, 2.), q2 = (int)fmax(p2, 2.);
    return q1 > q2 ? q1 : q2;
}

__device__ __forceinline__ int quad_order_fast(const DevFormula &F, double h1, double h2, float lh1, float lh2,
                                               float L1, float L2, double Ld1, double Ld2, double d2) {
    const float ld = 0.5f*0.69314718056f*__builtin_amdgcn_logf((float)d2);
    const float logdh1 = ld-lh1, logdh2 = ld-lh2;
    const float Lm = fmaxf(L1, L2);
    const float n1 = F.clip ? fmaxf(logdh1, 0.f) : logdh1, n2 = F.clip ? fmaxf(logdh2, 0.f) : logdh2;
    const float c0 = (float)F.c0, a = (float)F.a, b = (float)F.b, e = (float)F.e, den0 = (float)F.den0;
    const float a1 = (c0+a*L2+b*Lm-e*n2)*__builtin_amdgcn_rcpf(fmaxf(logdh1, 0.f)+den0);
    const float a2 = (c0+a*L1+b*Lm-e*n1)*__builtin_amdgcn_rcpf(fmaxf(logdh2, 0.f)+den0);
    const float r1 = rintf(a1), r2 = rintf(a2);
    const bool risky = (a1 > 1.5f && fabsf(a1-r1) < 2e-4f) || (a2 > 1.5f && fabsf(a2-r2) < 2e-4f) || !(a1 == a1) || !(a2 == a2);
    if (risky) return quad_order_exact(F, h1, h2, Ld1, Ld2, sqrt(d2));
    const int q1 = (int)fmaxf(ceilf(a1), 2.f), q2 = (int)fmaxf(ceilf(a2), 2.f);
    return q1 > q2 ? q1 : q2;
}

// hardware fp64 adds (global_atomic_add_f64 / ds_add_f64, no CAS loop); built with -munsafe-fp-atomics
__device__ __forceinline__ void atomic_add_f64(double *p, double v) {
    (void)__hip_atomic_fetch_add(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ void lds_add_f64(double *p, double v) {
    (void)__hip_atomic_fetch_add(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
}


// ---------------------------------------------------------------------------------------------
// wave-wide sum with DPP row shifts / broadcasts (no LDS traffic); result in every lane
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ double dpp_add(double v) {
    // full row mask: lanes without a source read 0 (bound_ctrl), so the destination needs no zero-initialised "old" value
    // (two v_mov_b32 less per step); partial row masks keep old = 0 in the disabled rows
    constexpr bool BC = ROW_MASK == 0xf;
    const int lo = __builtin_amdgcn_update_dpp(0, __double2loint(v), CTRL, ROW_MASK, 0xf, BC);
    const int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(v), CTRL, ROW_MASK, 0xf, BC);
    return v+__hiloint2double(hi, lo);
}
__device__ __forceinline__ double wave_sum(double v) {
    v = dpp_add<0x111, 0xf>(v);      // row_shr:1
    v = dpp_add<0x112, 0xf>(v);      // row_shr:2
    v = dpp_add<0x114, 0xf>(v);      // row_shr:4
    v = dpp_add<0x118, 0xf>(v);      // row_shr:8  -> lane 15 of every row holds the row sum
    v = dpp_add<0x142, 0xa>(v);      // row_bcast:15 into rows 1 and 3
    v = dpp_add<0x143, 0xc>(v);      // row_bcast:31 into rows 2 and 3 -> lane 63 holds the total
    const int lo = __builtin_amdgcn_readlane(__double2loint(v), 63);
    const int hi = __builtin_amdgcn_readlane(__double2hiint(v), 63);
    return __hiloint2double(hi, lo);
}

// Three wave-wide sums at once (the column sums of the 3-point rule): the first two butterfly stages pack the three inputs
// by lane & 3 (lanes 0, 1 -> a, b; lanes 2, 3 -> c), so that from then on ONE value per lane is reduced over the lanes of
// equal lane & 3: row rotations by 4 and 8, then the gfx950 row / half-wave swaps (v_permlane16_swap, v_permlane32_swap).
// 39 VALU operations instead of 3 x 24 for three separate wave_sum calls.
template <int CTRL>
__device__ __forceinline__ double dpp_get(double v) {
    const int lo = __builtin_amdgcn_update_dpp(0, __double2loint(v), CTRL, 0xf, 0xf, true);
    const int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(v), CTRL, 0xf, 0xf, true);
    return __hiloint2double(hi, lo);
}
__device__ __forceinline__ double add_xor16(double v) {      // v[l] + v[l ^ 16]
    const auto lo = __builtin_amdgcn_permlane16_swap(__double2loint(v), __double2loint(v), false, false);
    const auto hi = __builtin_amdgcn_permlane16_swap(__double2hiint(v), __double2hiint(v), false, false);
    return __hiloint2double(hi[0], lo[0])+__hiloint2double(hi[1], lo[1]);
}
__device__ __forceinline__ double add_xor32(double v) {      // v[l] + v[l ^ 32]
    const auto lo = __builtin_amdgcn_permlane32_swap(__double2loint(v), __double2loint(v), false, false);
    const auto hi = __builtin_amdgcn_permlane32_swap(__double2hiint(v), __double2hiint(v), false, false);
    return __hiloint2double(hi[0], lo[0])+__hiloint2double(hi[1], lo[1]);
}
__device__ __forceinline__ void wave_sum3(double a, double b, double c, double &A, double &B, double &C) {
    const int lane = threadIdx.x & 63;
    const bool o1 = (lane & 1) != 0, o2 = (lane & 2) != 0;
    double x = o1 ? b : a;
    x += dpp_get<0xB1>(o1 ? a : b);            // quad_perm [1,0,3,2]: even lanes hold a pair sum of a, odd lanes of b
    const double y = c+dpp_get<0xB1>(c);       // pair sums of c in both lanes
    double z = o2 ? y : x;
    z += dpp_get<0x4E>(o2 ? x : y);            // quad_perm [2,3,0,1]: lane & 3 = 0: quad sum of a, 1: of b, 2 and 3: of c
    z += dpp_get<0x124>(z);                    // row_ror:4
    z += dpp_get<0x128>(z);                    // row_ror:8 -> row sums, by lane & 3
    z = add_xor16(z);
    z = add_xor32(z);
    const int lo = __double2loint(z), hi = __double2hiint(z);
    A = __hiloint2double(__builtin_amdgcn_readlane(hi, 0), __builtin_amdgcn_readlane(lo, 0));
    B = __hiloint2double(__builtin_amdgcn_readlane(hi, 1), __builtin_amdgcn_readlane(lo, 1));
    C = __hiloint2double(__builtin_amdgcn_readlane(hi, 2), __builtin_amdgcn_readlane(lo, 2));
}

// sum over the 16 lanes of a DPP row, result in every lane of the row
template <int CTRL>
__device__ __forceinline__ double dpp_row_add(double v) {
    const int lo = __builtin_amdgcn_update_dpp(0, __double2loint(v), CTRL, 0xf, 0xf, false);
    const int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(v), CTRL, 0xf, 0xf, false);
    return v+__hiloint2double(hi, lo);
}
__device__ __forceinline__ double row16_sum(double v) {
    v = dpp_row_add<0xB1>(v);        // quad_perm [1,0,3,2]
    v = dpp_row_add<0x4E>(v);        // quad_perm [2,3,0,1]
    v = dpp_row_add<0x141>(v);       // row_half_mirror
    v = dpp_row_add<0x140>(v);       // row_mirror
    return v;
}

