// Device helpers shared by the translation units of libpnl_hip.so (gfx950 only): kernel function gamma, quadrature
// order formula, hardware fp64 atomics, DPP / permlane wave reductions.
#pragma once
#include "pnl_device.h"

// read-only tables through the constant address space: loads with a wave-uniform address are scalar loads (SGPRs)
typedef const double __attribute__((address_space(4))) *pnl_const_f64_ptr;

// ---- ln and exp for the general kernel exponent: d2^e = exp(e ln d2) ------------------------------------------------------
// Table-driven range reduction (64 intervals each), then short polynomials: 25 fp64 operations and three table loads per
// power instead of 55 operations with the long series; a few 1e-16 relative, like the series.
// 1 / c_j (rounded) and -ln of that rounded value, c_j the centre of [1/2 + j/128, 1/2 + (j+1)/128)
static __constant__ double PNL_LOG_TAB[64][2] = {
    {1.9844961240310077, -0.6853650401178903}, {1.9541984732824427, -0.6699801212784109},
    {1.9248120300751879, -0.6548283162578087}, {1.8962962962962964, -0.6399026660411331},
    {1.8686131386861313, -0.6251965186514375}, {1.841726618705036, -0.6107035113488708},
    {1.8156028368794326, -0.5964175541013942}, {1.7902097902097902, -0.5823328142196552},
    {1.7655172413793103, -0.5684437020589881}, {1.7414965986394557, -0.5547448577008262},
    {1.7181208053691275, -0.5412311385341033}, {1.695364238410596, -0.5278976076646381},
    {1.673202614379085, -0.514739523087127}, {1.6516129032258065, -0.5017523275603158},
    {1.6305732484076434, -0.4889316391312545}, {1.610062893081761, -0.476273242259331},
    {1.5900621118012421, -0.46377307949509944}, {1.5705521472392638, -0.4514272436728002},
    {1.5515151515151515, -0.4392319705789819}, {1.532934131736527, -0.4271836320628074},
    {1.514792899408284, -0.415278729556489}, {1.4970760233918128, -0.4035138879769026},
    {1.4797687861271676, -0.3918858499817835}, {1.4628571428571429, -0.38039147055604844},
    {1.4463276836158192, -0.3690277119057333}, {1.4301675977653632, -0.35779163863880753},
    {1.4143646408839778, -0.34668041321373666}, {1.3989071038251366, -0.33569129163814154},
    {1.3837837837837839, -0.3248216194012377}, {1.3689839572192513, -0.3140688276249758},
    {1.3544973544973544, -0.30343042941992004}, {1.3403141361256545, -0.2929040164329327},
    {1.3264248704663213, -0.28248725557467697}, {1.3128205128205128, -0.27217788591581565},
    {1.299492385786802, -0.2619737157415739}, {1.2864321608040201, -0.2518726197550701},
    {1.2736318407960199, -0.2418725364204867}, {1.2610837438423645, -0.23197146543777517},
    {1.248780487804878, -0.2221674653411543}, {1.2367149758454106, -0.21245865121419336},
    {1.2248803827751196, -0.20284319251475144}, {1.2132701421800949, -0.19331931100349606},
    {1.2018779342723005, -0.18388527877013738}, {1.1906976744186046, -0.17453941635189965},
    {1.1797235023041475, -0.16528009093910292}, {1.1689497716894977, -0.1561057146630616},
    {1.158371040723982, -0.14701474296180975}, {1.147982062780269, -0.1380056730194437},
    {1.1377777777777778, -0.12907704227514236}, {1.1277533039647578, -0.12022742699815989},
    {1.1179039301310043, -0.11145544092532278}, {1.1082251082251082, -0.10275973395776894},
    {1.0987124463519313, -0.09413899091386191}, {1.0893617021276596, -0.08559193033540353},
    {1.080168776371308, -0.0771173033444312}, {1.0711297071129706, -0.06871389254805173},
    {1.062240663900415, -0.06038051098890748}, {1.0534979423868314, -0.0521160011390141},
    {1.0448979591836736, -0.04391923393483558}, {1.0364372469635628, -0.03578910785158529},
    {1.0281124497991967, -0.02772454801485477}, {1.0199203187250996, -0.019724505347778573},
    {1.0118577075098814, -0.011787955752042173}, {1.003921568627451, -0.003913899321136315},
};
// 2^(j/64)
static __constant__ double PNL_EXP_TAB[64] = {
    1.0, 1.0108892860517005, 1.0218971486541166, 1.0330248790212284,
    1.0442737824274138, 1.0556451783605572, 1.0671404006768237, 1.0787607977571199,
    1.0905077326652577, 1.102382583307841, 1.1143867425958924, 1.1265216186082418,
    1.1387886347566916, 1.1511892299529827, 1.1637248587775775, 1.1763969916502812,
    1.189207115002721, 1.202156731452703, 1.215247359980469, 1.22848053610687,
    1.241857812073484, 1.255380757024691, 1.2690509571917332, 1.2828700160787783,
    1.2968395546510096, 1.3109612115247644, 1.3252366431597413, 1.339667524053303,
    1.3542555469368927, 1.3690024229745905, 1.383909881963832, 1.3989796725383112,
    1.4142135623730951, 1.42961333839197, 1.4451808069770467, 1.460917794180647,
    1.4768261459394993, 1.4929077282912648, 1.5091644275934228, 1.5255981507445384,
    1.5422108254079407, 1.559004400237837, 1.5759808451078865, 1.593142151342267,
    1.6104903319492543, 1.6280274218573478, 1.645755478153965, 1.6636765803267364,
    1.681792830507429, 1.7001063537185235, 1.718619298122478, 1.7373338352737062,
    1.7562521603732995, 1.7753764925265212, 1.7947090750031072, 1.8142521755003989,
    1.8340080864093424, 1.8539791250833855, 1.8741676341103, 1.8945759815869656,
    1.9152065613971474, 1.9360617934922943, 1.9571441241754002, 1.978456026387951,
};
// ln 2 / 64 = hi + lo with 36 significant bits in hi (n hi is exact for |n| < 2^17), 64 / ln 2
#define PNL_LN2_64_HI 0.010830424696223417
#define PNL_LN2_64_LO 2.572804622327669e-14
#define PNL_64_LN2 92.33248261689366

// ln x, x > 0 normal: x = m 2^k, m in [1/2, 1); j = top six fraction bits of m, u = m / c_j - 1 (one FMA, |u| <= 2^-7),
// ln x = k ln 2 + ln c_j + log1p(u), log1p by its series to u^7 (next term 2^-56 / 8).
// Exponent and mantissa are taken apart with integer operations on the high word and int <-> double conversions go through
// the 2^52 trick: v_frexp_*_f64, v_ldexp_f64, v_rndne_f64 and the f64 conversions issue at a quarter of the FMA rate.
__device__ __forceinline__ double pnl_log(double x) {
    const int hi = __double2hiint(x);
    const double m = __hiloint2double((hi & 0x800fffff) | 0x3fe00000, __double2loint(x));
    const int j = (hi >> 14) & 63;
    // k = biased exponent - 1022 as a double: 2^52 + 2^31 + k has the low word k ^ 0x80000000
    const double kd = __hiloint2double(0x43300000, (((hi >> 20) & 0x7ff)-1022) ^ 0x80000000)-4503601774854144.0;
    const double u = __builtin_fma(m, PNL_LOG_TAB[j][0], -1.0);
    double p = 1.0/7.0;
    p = __builtin_fma(p, u, -1.0/6.0);
    p = __builtin_fma(p, u, 0.2);
    p = __builtin_fma(p, u, -0.25);
    p = __builtin_fma(p, u, 1.0/3.0);
    p = __builtin_fma(p, u, -0.5);
    p = __builtin_fma(p*u, u, u);
    return __builtin_fma(kd, 6.93147180369123816490e-01, __builtin_fma(kd, 1.90821492927058770002e-10, PNL_LOG_TAB[j][1]+p));
}

// exp y, |y| < 700: y = n ln 2 / 64 + r, |r| <= ln 2 / 128, exp y = 2^(n >> 6) 2^((n & 63)/64) exp r, exp r to r^5 (next 3e-17);
// n = rint(y 64 / ln 2) from the low word of y 64 / ln 2 + 1.5 2^52, the power of two added to the exponent field (the result
// is a normal number: kernel values, |y| < 700)
__device__ __forceinline__ double pnl_exp(double y) {
    const double big = __builtin_fma(y, PNL_64_LN2, 6755399441055744.0);
    const int ni = __double2loint(big);
    const double n = big-6755399441055744.0;
    double r = __builtin_fma(-n, PNL_LN2_64_HI, y);
    r = __builtin_fma(-n, PNL_LN2_64_LO, r);
    double p = 1.0/120.0;
    p = __builtin_fma(p, r, 1.0/24.0);
    p = __builtin_fma(p, r, 1.0/6.0);
    p = __builtin_fma(p, r, 0.5);
    p = __builtin_fma(p, r, 1.0);
    p = __builtin_fma(p, r, 1.0);
    const double v = p*PNL_EXP_TAB[ni & 63];
    return __hiloint2double(__double2hiint(v)+((ni >> 6) << 20), __double2loint(v));
}

// General power through tables in LDS (the tile kernels of a general exponent): x = 2^k m, m in [1, 2), j = top seven fraction
// bits of m, u = m T[j] - 1 with T[j] = fl(1 / c_j) (one FMA, |u| <= 2^-8),
//   scale x^e = (scale c_j^e) 2^(e k) (1 + u)^e = T[128 + j] T[256 + k + 96] (1 + u)^e,
// the last factor by its binomial series to u^6 (coefficients C(e, i) of the kernel class in DevKernel::pb; the next term is
// below 1e-16 for |e| <= 2).  18 operations and three LDS gathers against 36 operations and three L1 gathers of
// exp(e ln x) above; the tables (PNL_POW_TAB_DOUBLES doubles per exponent, built in long double by pow_table in
// pnl_hip.hip) are copied to LDS by the kernel.  Exponents of x outside 2^-96 ... 2^31 are clamped (|x - y| < 1e-14).
#define PNL_POW_TAB_DOUBLES 384
__device__ __forceinline__ void pnl_pow_tab_fill(double *dst, const double *__restrict__ src, int tid, int nthreads) {
    if (src) for (int t = tid; t < PNL_POW_TAB_DOUBLES; t += nthreads) dst[t] = src[t];
}
__device__ __forceinline__ double pnl_pow_tab(double x, const DevKernel &k, const double *__restrict__ T) {
    const int hi = __double2hiint(x);
    const int j = (hi >> 13) & 127;
    const int kx = min(max(((hi >> 20) & 0x7ff)-(1023-96), 0), 127);
    const double m = __hiloint2double((hi & 0x000fffff) | 0x3ff00000, __double2loint(x));
    const double u = __builtin_fma(m, T[j], -1.0);
    double p = k.pb[5];
    p = __builtin_fma(p, u, k.pb[4]);
    p = __builtin_fma(p, u, k.pb[3]);
    p = __builtin_fma(p, u, k.pb[2]);
    p = __builtin_fma(p, u, k.pb[1]);
    p = __builtin_fma(p, u, k.pb[0]);
    p = __builtin_fma(p, u, 1.0);
    return (T[128+j]*T[256+kx])*p;
}

// ---- kernel function gamma(|x-y|^2)   (KC:75-294) ----------------------------------------------------------------------
// KT: 0 general (pow / indicator / peridynamic, horizon test), 1 fractional with exponent -qm/4, qm a run-time (wave-uniform)
// value, 2 the same with qm == 6 known at compile time (s = 1/2 in 2D): no branch per evaluation, so the compiler
// interleaves the dependent chains of the independent evaluations of a pair.
// erfc of the device library, kept out of line: it is evaluated by one boundary kernel (Gaussian, 1D) and would otherwise be inlined
// into the general branch of every KT == 0 hot loop (library 9.3 -> 13.3 MB, compile 3.2 -> 5.2 min)
__device__ __noinline__ static double pnl_erfc(double x) { return erfc(x); }

// KT == 3 is not a kernel instantiation of its own: the KT == 0 kernels enter their hot loops with it when the kernel is fractional
// with power tables in LDS and no horizon (kern_eval_pow_ok) -- the general branch below tests the horizon per lane and switches on
// the kernel type per evaluation, which costs more than the branches: no two evaluations are ever scheduled together.
// BND: the kernel may be a Gauss-theorem twin of an integrable kernel (types 5 .. 8: only boundary kernels are).  Their formulas --
// erfc above all -- stay out of the kernels that integrate element pairs: the register allocation of a kernel is the maximum over
// its paths, and with them in the general branch k_worklist_sorted<2, 3, 0> went from 240 VGPRs to 256 + 56 AGPRs, i.e. from two
// waves per SIMD to one, also for the fractional kernels that never take that branch (8.5 instead of 6.1 ms at 98,304 cells, s = 0.4).
// d2^(-QM/4) for odd QM (s = 1/4, 3/4 in 1D / 2D): y = d2^(-1/4) from the single-precision pipeline (v_log_f32, v_exp_f32: |1 - d2 y^4| =
// e < 5e-6 for d2 in [1e-10, 8]), then y^QM (1 - e)^(-QM/4) with the series to e^2 (the next term is below 4e-16).  Two quarter-rate
// single-precision operations and 8-9 double-precision ones instead of two v_rsq_f64 (17.7 cycles per wave each) with a cubic
// correction each and the integer power: 88 instead of 126 cycles per kernel value for QM = 7 (tools/probes/valu_rate_probe.hip).
template <int QM>
__device__ __forceinline__ double pnl_pow_quarter_odd(double d2) {
    static_assert(QM == 3 || QM == 5 || QM == 7, "exponents -3/4, -5/4, -7/4");
    const double y = (double)__builtin_amdgcn_exp2f(-0.25f*__builtin_amdgcn_logf((float)d2));
    const double y2 = y*y, y4 = y2*y2;
    const double e = __builtin_fma(-d2, y4, 1.0);
    const double P = QM == 3 ? y2*y : (QM == 5 ? y4*y : (y4*y2)*y);
    // c1 + c2 e with c1 = QM/4 from an opaque register pair and the three-address v_fma spelled out (see KT == 2 below)
    double c1 = 0.25*QM, q;
    asm("" : "+v"(c1));
    asm("v_fma_f64 %0, %1, %2, %3" : "=v"(q) : "v"(e), "s"(0.25*QM*(0.25*QM+1.)*0.5), "v"(c1));
    return __builtin_fma(P, e*q, P);
}

template <int KT, bool BND = false>
__device__ __forceinline__ double kern_eval(const DevKernel &k, double d2, const double *__restrict__ ltab = nullptr) {
    if (KT == 3) return pnl_pow_tab(d2, k, ltab);
    if (KT == 4) return k.scale;                        // constant kernel on a pair that lies inside the horizon (kern_dispatch<0, true>)
    if (KT >= 10) {
        // KT == 1 with the exponent -QM/4 known at compile time (QM = KT - 10; kern_dispatch): the scalar branches and the
        // square-and-multiply loop of the run-time version below split every evaluation into basic blocks of its own
        constexpr int QM = KT-10;
        if constexpr (QM == 3 || QM == 5 || QM == 7) return pnl_pow_quarter_odd<(QM == 3 || QM == 5 || QM == 7) ? QM : 3>(d2);
        double r = __builtin_amdgcn_rsq(d2);
        {
            const double e = __builtin_fma(-(d2*r), r, 1.0);
            r = __builtin_fma(r, e*__builtin_fma(0.375, e, 0.5), r);
        }
        double base = r;
        int p = QM/2;
        if (QM & 1) {
            double t = __builtin_amdgcn_rsq(r);
            const double e = __builtin_fma(-(r*t), t, 1.0);
            t = __builtin_fma(t, e*__builtin_fma(0.375, e, 0.5), t);
            base = r*t;                                  // d2^(-1/4)
            p = QM;
        }
        double res = (p & 1) ? base : 1.;
#pragma unroll
        for (int b = 1; b < 6; b++) {                   // p < 64; the trip count and every test are compile-time constants
            if ((p >> b) == 0) break;
            base *= base;
            if ((p >> b) & 1) res = ((p & ((1 << b)-1)) == 0) ? base : res*base;
        }
        return res;
    }
    if (KT == 2) {
        // d2^(-3/2) = r^3 (1 - e)^(-3/2) with r = v_rsq_f64(d2) (~2^-23 relative), e = 1 - d2 r^2 (|e| < 3e-7):
        // r^3 (1 + e (3/2 + 15/8 e)), the next term 35/16 e^3 is below 1e-19; six operations after the rsq, chain depth five
        // 3/2 is no inline constant: as an immediate the compiler rebuilds it in a register pair for every evaluation (v_fmac wants
        // the addend in its destination: two v_mov per kernel value, 18 of the 270 instructions per pair of the 3-point tile loop);
        // an opaque loop-invariant register pair and the three-address form spelled out make it one v_fma
        double c15 = 1.5, p;
        asm("" : "+v"(c15));
        const double r = __builtin_amdgcn_rsq(d2);
        const double t = r*r;
        const double e = __builtin_fma(-d2, t, 1.0);
        const double g0 = r*t;
        asm("v_fma_f64 %0, %1, %2, %3" : "=v"(p) : "v"(e), "s"(1.875), "v"(c15));
        return __builtin_fma(g0, e*p, g0);
    } else if (KT == 1) {
        // exponent = -qm/4 (s a multiple of 1/4 in 1D / 2D): d2^(-1/2) from v_rsq_f64 (~2^-23 relative) + one Halley step
        // (cubic: r (1 + e/2 + 3 e^2/8), e = 1 - d2 r^2, five operations for full precision), for odd qm one more refined
        // rsqrt gives d2^(-1/4); then an integer power.  A few ulp instead of libm's pow at 1/8 of the cost.  The scale is
        // applied once per pair (kern_scale); qm is wave-uniform, the branches are scalar.
        int p = k.qm;
        if (p == 7) return pnl_pow_quarter_odd<7>(d2);   // wave-uniform branches
        if (p == 5) return pnl_pow_quarter_odd<5>(d2);
        if (p == 3) return pnl_pow_quarter_odd<3>(d2);
        double r = __builtin_amdgcn_rsq(d2);
        {
            const double e = __builtin_fma(-(d2*r), r, 1.0);
            r = __builtin_fma(r, e*__builtin_fma(0.375, e, 0.5), r);
        }
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
        if (k.ktype == 0) return ltab ? pnl_pow_tab(d2, k, ltab) : k.scale*pnl_exp(k.exponent*pnl_log(d2));
        if (k.ktype == 1) return k.scale;
        if (k.ktype == 2) return k.scale/sqrt(d2);
        if (k.ktype == 3) return k.scale*pnl_exp(k.exponent*d2);              // Gaussian: exponent = -1 / (2 variance^d) or -9 / horizon^2
        if (k.ktype == 4 || !BND) return k.scale*pnl_exp(k.exponent*sqrt(d2));       // exponential: exponent = -rate
        // Gauss-theorem twins of the integrable kernels on the full space (kernelsCy.pyx:418-477; gammainc(a, x) there is the
        // unnormalised upper incomplete Gamma function, :39-40): 5 Gaussian 1D, 6 exponential; 7 / 8 the 2D forms with the
        // 1 / |x-y| of the normal factor folded in (DevProblem::bkn)
        if (k.ktype == 5) return k.scale*sqrt(3.14159265358979323846/(-k.exponent))*pnl_erfc(sqrt(-k.exponent*d2));
        if (k.ktype == 6) return 2.*k.scale*pnl_exp(k.exponent*sqrt(d2))/(-k.exponent);
        if (k.ktype == 7) return k.scale*pnl_exp(k.exponent*d2)/(-k.exponent*d2);
        return 2.*k.scale*pnl_exp(k.exponent*sqrt(d2))/(-k.exponent*sqrt(d2));
    }
}

template <int KT>
__device__ __forceinline__ double kern_scale(const DevKernel &k) { return (KT == 1 || KT == 2 || KT >= 10) ? k.scale : 1.; }
// may a KT == 0 kernel run its hot loop with KT == 3 (wave-uniform)?
__device__ __forceinline__ bool kern_eval_pow_ok(const DevKernel &k, const double *ltab) {
    return ltab != nullptr && k.ktype == 0 && !(k.horizon2 < 1e300);
}
// hot(tag): a generic lambda whose body uses decltype(tag)::value as the KT of its kern_eval calls
template <int KT> struct KTag { static constexpr int value = KT; };
// INSIDE: the caller knows that every point pair it evaluates lies inside the horizon (the lists of a finite-horizon tile hold the
// pairs that rel_position classified as interacting; the cut ones take eval_distant_cut) -- the horizon test is void, and the
// constant kernel needs no distance at all
template <int KT, bool INSIDE = false, typename F>
__device__ __forceinline__ void kern_dispatch(const DevKernel &k, const double *ltab, F &&hot) {
    if constexpr (KT == 0) {
        if (kern_eval_pow_ok(k, ltab) || (INSIDE && ltab != nullptr && k.ktype == 0)) { hot(KTag<3>{}); return; }
        if constexpr (INSIDE) {
            if (k.ktype == 1) { hot(KTag<4>{}); return; }
        }
    }
    if constexpr (KT == 1) {
        // s = 1/4 and 3/4: 2D d2^(-5/4), d2^(-7/4); 1D d2^(-3/4), d2^(-5/4); (s = 1/2 in 1D: d2^(-1))
        switch (k.qm) {
        case 3: hot(KTag<13>{}); return;
        case 4: hot(KTag<14>{}); return;
        case 5: hot(KTag<15>{}); return;
        case 7: hot(KTag<17>{}); return;
        default: break;
        }
    }
    hot(KTag<KT>{});
}

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

// the fp32 evaluation alone: -1 where it is within 2e-4 of a change of the order (the caller then evaluates quad_order_exact)
__device__ __forceinline__ int quad_order_try(const DevFormula &F, float lh1, float lh2, float L1, float L2, double d2) {
    const float ld = 0.5f*0.69314718056f*__builtin_amdgcn_logf((float)d2);
    const float logdh1 = ld-lh1, logdh2 = ld-lh2;
    const float Lm = fmaxf(L1, L2);
    const float n1 = F.clip ? fmaxf(logdh1, 0.f) : logdh1, n2 = F.clip ? fmaxf(logdh2, 0.f) : logdh2;
    const float c0 = (float)F.c0, a = (float)F.a, b = (float)F.b, e = (float)F.e, den0 = (float)F.den0;
    const float a1 = (c0+a*L2+b*Lm-e*n2)*__builtin_amdgcn_rcpf(fmaxf(logdh1, 0.f)+den0);
    const float a2 = (c0+a*L1+b*Lm-e*n1)*__builtin_amdgcn_rcpf(fmaxf(logdh2, 0.f)+den0);
    const float r1 = rintf(a1), r2 = rintf(a2);
    const bool risky = (a1 > 1.5f && fabsf(a1-r1) < 2e-4f) || (a2 > 1.5f && fabsf(a2-r2) < 2e-4f) || !(a1 == a1) || !(a2 == a2);
    const int q1 = (int)fmaxf(ceilf(a1), 2.f), q2 = (int)fmaxf(ceilf(a2), 2.f);
    return risky ? -1 : (q1 > q2 ? q1 : q2);
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

// stores into the block-slot storage (written once by the tile kernels, read once by the fold pass) and of the fold pass into A
#ifndef PNL_SLOT_NT
#define PNL_SLOT_NT 0
#endif
__device__ __forceinline__ void slot_store(double *p, double v) {
    if (PNL_SLOT_NT) __builtin_nontemporal_store(v, p); else *p = v;
}
__device__ __forceinline__ void slot_store2(double *p, double x, double y) {
    typedef double pnl_d2 __attribute__((ext_vector_type(2)));
    const pnl_d2 v = {x, y};
    if (PNL_SLOT_NT) __builtin_nontemporal_store(v, (pnl_d2*)p); else *(pnl_d2*)p = v;
}

// workgroup barrier that orders LDS traffic only.  __syncthreads() also waits for every global atomic / store of the wave
// (s_waitcnt vmcnt(0)): the flush of a tile would have to retire before the next tile may start.  The tile kernels below
// never read global memory that the same launch writes, so their barriers only have to order the LDS.
__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

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

// DoFs per vertex / per edge of the elements the kernels are instantiated for: P0 (one DoF per cell), P1, P2, P3 on intervals
// (two vertices + two cell DoFs); the merged local DoFs of a touching pair follow from them (FL2:662-811, FL1:255-330)
__host__ __device__ constexpr int elem_dpv(int dpe) { return dpe == 1 ? 0 : 1; }
__host__ __device__ constexpr int elem_dped(int dim, int dpe) { return (dim == 2 && dpe == 6) ? 1 : 0; }

// ---- distant pairs, factorised accumulation (see pnl_kernels.h: eval_distant) ---------------------------------------------
template <int DIM, int DPE>
struct PairAcc {
    static constexpr int ND = DPE*(DPE+1)/2;
    double G[DPE][DPE];
    double S1[ND], S2[ND];
    __device__ __forceinline__ void clear() {
#pragma unroll
        for (int a = 0; a < DPE; a++)
#pragma unroll
            for (int b = 0; b < DPE; b++) G[a][b] = 0.;
#pragma unroll
        for (int e = 0; e < ND; e++) { S1[e] = 0.; S2[e] = 0.; }
    }
};

// Blocked evaluation for a rule of run-time length (the work-list kernels).  eval_distant_lds above pays the DPE (DPE + 1) / 2
// FMAs of S2 per point pair (21 for P2: as much as the kernel value itself); here the columns j are walked in blocks of PNL_WL_JB
// whose column sums c_j = sum_i w_i g_ij stay in registers, the rows i = i_first, i_first + i_step, ... < n inside a block: per
// point pair the kernel value, one FMA each for the row and the column sum and DPE - 1 for u_b (the shape functions sum to one:
// u_{DPE-1} = r - sum_b u_b).  G and S1 are linear in the partial row sums of a block and take them once per (i, block), S2 takes
// the column sums once per block.  The points are x_i - y_j = (a_0 - b_0) + sum_k lambda_k(i) (a_k - a_0) - sum_k lambda_k(j)
// (b_k - b_0): four FMAs per point pair in 2D.  The columns are unrolled and guarded in groups of four by wave-uniform branches;
// the caller pads the rule copy to n4 = a multiple of four points with ZERO-WEIGHT copies of point 0 (finite kernel values that
// enter nothing).  Work per point pair for P2: 14 + kernel value + about 5 amortised, against 56 + kernel value.
// POWTAB: fractional kernel with the LDS power tables and no horizon -- no branch per evaluation.
#ifndef PNL_WL_JB
#define PNL_WL_JB 16        // columns per block (column sums in registers)
#endif
#ifndef PNL_WL_JG
#define PNL_WL_JG 4         // columns per guarded group (1, 2 or 4: the rule copy is padded to a multiple of four points)
#endif
// CM: where the column sums of a finished block go.  0: straight into S2 (the accumulators of S2 are live in the hot loop: fine for
// P1).  P2 carries 78 accumulators per lane, more than the hot loop leaves room for in 256 VGPRs, and the register allocator then
// serialises every LDS read of the loop behind a wait (measured: issue utilisation 0.14); so the column sums leave through LDS and
// S2 is formed after the last block, when the registers of the hot loop are free: 1: sc[j * cstride] is private to the lane (one
// pair per lane), 2: sc[j] is shared by the i_step lanes of the pair (ds_add_f64), which then split the columns among them.
// ONEBLOCK: the caller guarantees n4 <= PNL_WL_JB (the rules a tile kernel integrates in its lanes): no loop over blocks, S2 comes
// alive after the hot loop.
template <int DIM, int DPE, int KT, bool POWTAB, int CM, bool ONEBLOCK = false, int JG = PNL_WL_JG>
__device__ __forceinline__ void eval_distant_blocked(const DevKernel &kern, const double *__restrict__ tab, int stp, int n, int n4,
                                                     int i_first, int i_step, const double *av, const double *bv,
                                                     PairAcc<DIM, DPE> &R, const double *__restrict__ lpow, double *sc, int cstride) {
    constexpr int JB = PNL_WL_JB;
    if (CM == 2) {
        for (int j = i_first; j < n4; j += i_step) sc[j] = 0.;
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();
    }
    double ea[DIM][DIM], eb[DIM][DIM], ab[DIM];
#pragma unroll
    for (int d = 0; d < DIM; d++) {
        ab[d] = av[d]-bv[d];
#pragma unroll
        for (int k = 0; k < DIM; k++) { ea[k][d] = av[(k+1)*DIM+d]-av[d]; eb[k][d] = bv[(k+1)*DIM+d]-bv[d]; }
    }
#pragma unroll 1
    for (int j0 = 0; j0 < (ONEBLOCK ? 1 : n4); j0 += JB) {
        const double *__restrict__ tj0 = tab+j0*stp;
        double c[JB];
#pragma unroll
        for (int jj = 0; jj < JB; jj++) c[jj] = 0.;
#pragma unroll 1
        for (int i = i_first; i < n; i += i_step) {
            const double *__restrict__ ti = tab+i*stp;
            double t0[DIM];
#pragma unroll
            for (int d = 0; d < DIM; d++) {
                double sx = ab[d];
#pragma unroll
                for (int k = 0; k < DIM; k++) sx = __builtin_fma(ti[1+k], ea[k][d], sx);
                t0[d] = sx;
            }
            const double wi = ti[3];
            double r = 0., u[DPE];
#pragma unroll
            for (int b = 0; b < DPE; b++) u[b] = 0.;
#pragma unroll
            for (int jq = 0; jq < JB; jq += JG) {
                if (j0+jq < n4) {
                    // the JG columns of a group advance together, stage by stage (rule data, distances, table look-ups of the power,
                    // series, accumulation): every wait for LDS covers JG independent reads.  The empty asm statements pin the loaded
                    // values where they stand -- left alone, the scheduler sinks each read to its first use and waits for it there.
                    double lam[JG][DIM], wj[JG], ph[JG][DPE > 1 ? DPE-1 : 1], d2[JG], g[JG];
#pragma unroll
                    for (int jj = 0; jj < JG; jj++) {
                        const double *__restrict__ tj = tj0+(jq+jj)*stp;
#pragma unroll
                        for (int k = 0; k < DIM; k++) lam[jj][k] = tj[1+k];
                        wj[jj] = tj[3];
#pragma unroll
                        for (int b = 0; b+1 < DPE; b++) ph[jj][b] = tj[4+b];
                    }
#pragma unroll
                    for (int jj = 0; jj < JG; jj++) {
#pragma unroll
                        for (int k = 0; k < DIM; k++) asm volatile("" : "+v"(lam[jj][k]));
                        asm volatile("" : "+v"(wj[jj]));
#pragma unroll
                        for (int b = 0; b+1 < DPE; b++) asm volatile("" : "+v"(ph[jj][b]));
                    }
#pragma unroll
                    for (int jj = 0; jj < JG; jj++) {
                        double dd = 0.;
#pragma unroll
                        for (int d = 0; d < DIM; d++) {
                            double t = t0[d];
#pragma unroll
                            for (int k = 0; k < DIM; k++) t = __builtin_fma(-lam[jj][k], eb[k][d], t);
                            dd = __builtin_fma(t, t, dd);
                        }
                        d2[jj] = dd;
                    }
                    if (POWTAB) {
                        // pnl_pow_tab (pnl_common.h), its three look-ups issued for all columns before the first series
                        double m[JG], T0[JG], T1[JG], T2[JG];
#pragma unroll
                        for (int jj = 0; jj < JG; jj++) {
                            const int hi = __double2hiint(d2[jj]);
                            const int j = (hi >> 13) & 127;
                            const int kx = min(max(((hi >> 20) & 0x7ff)-(1023-96), 0), 127);
                            m[jj] = __hiloint2double((hi & 0x000fffff) | 0x3ff00000, __double2loint(d2[jj]));
                            T0[jj] = lpow[j]; T1[jj] = lpow[128+j]; T2[jj] = lpow[256+kx];
                        }
#pragma unroll
                        for (int jj = 0; jj < JG; jj++) { asm volatile("" : "+v"(T0[jj])); asm volatile("" : "+v"(T1[jj])); asm volatile("" : "+v"(T2[jj])); }
#pragma unroll
                        for (int jj = 0; jj < JG; jj++) {
                            const double uu = __builtin_fma(m[jj], T0[jj], -1.0);
                            double p = kern.pb[5];
                            p = __builtin_fma(p, uu, kern.pb[4]);
                            p = __builtin_fma(p, uu, kern.pb[3]);
                            p = __builtin_fma(p, uu, kern.pb[2]);
                            p = __builtin_fma(p, uu, kern.pb[1]);
                            p = __builtin_fma(p, uu, kern.pb[0]);
                            p = __builtin_fma(p, uu, 1.0);
                            g[jj] = (T1[jj]*T2[jj])*p;
                        }
                    } else {
#pragma unroll
                        for (int jj = 0; jj < JG; jj++) g[jj] = kern_eval<KT>(kern, d2[jj], lpow);
                    }
#pragma unroll
                    for (int jj = 0; jj < JG; jj++) {
                        c[jq+jj] = __builtin_fma(wi, g[jj], c[jq+jj]);
                        const double gw = g[jj]*wj[jj];
                        r += gw;
#pragma unroll
                        for (int b = 0; b+1 < DPE; b++) u[b] = __builtin_fma(gw, ph[jj][b], u[b]);
                    }
                }
            }
            u[DPE-1] = r;
#pragma unroll
            for (int b = 0; b+1 < DPE; b++) u[DPE-1] -= u[b];
            int e = 0;
#pragma unroll
            for (int a = 0; a < DPE; a++) {
                const double pa = wi*ti[4+a];
#pragma unroll
                for (int b = 0; b < DPE; b++) R.G[a][b] = __builtin_fma(pa, u[b], R.G[a][b]);
                const double pr = pa*r;
#pragma unroll
                for (int b = a; b < DPE; b++) { R.S1[e] = __builtin_fma(pr, ti[4+b], R.S1[e]); e++; }
            }
        }
#pragma unroll
        for (int jq = 0; jq < JB; jq += JG) {
            if (j0+jq < n4) {
#pragma unroll
                for (int jj = jq; jj < jq+JG; jj++) {
                    if (CM == 1) sc[(j0+jj)*cstride] = c[jj];
                    else if (CM == 2) atomic_add_f64(&sc[j0+jj], c[jj]);
                    else {
                        const double *__restrict__ tj = tj0+jj*stp;
                        const double cw = tj[3]*c[jj];
                        int e = 0;
#pragma unroll
                        for (int a = 0; a < DPE; a++) {
                            const double pc = tj[4+a]*cw;
#pragma unroll
                            for (int b = a; b < DPE; b++) { R.S2[e] = __builtin_fma(pc, tj[4+b], R.S2[e]); e++; }
                        }
                    }
                }
            }
        }
    }
    if (CM != 0) {
        if (CM == 2) {
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
            __builtin_amdgcn_wave_barrier();
        }
#pragma unroll 1
        for (int j = (CM == 2 ? i_first : 0); j < n; j += (CM == 2 ? i_step : 1)) {
            const double *__restrict__ tj = tab+j*stp;
            const double cw = tj[3]*sc[j*(CM == 2 ? 1 : cstride)];
            int e = 0;
#pragma unroll
            for (int a = 0; a < DPE; a++) {
                const double pc = tj[4+a]*cw;
#pragma unroll
                for (int b = a; b < DPE; b++) { R.S2[e] = __builtin_fma(pc, tj[4+b], R.S2[e]); e++; }
            }
        }
    }
}
