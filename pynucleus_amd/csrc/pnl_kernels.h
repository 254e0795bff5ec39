// Hand-written HIP kernels (CDNA4 / gfx950, wave64) for the nonlocal element-pair assembly.
//
// Work decomposition:
//   k_tile_distant   one workgroup per (cell-block a, cell-block b) tile of TILE x TILE cell pairs;
//                    pairs are classified, bucketed by quadrature order in LDS so that a wave runs
//                    one order at a time, integrated one pair per lane, and their local matrices are
//                    accumulated in an LDS sub-block of A before one coalesced atomic flush to HBM.
//   k_singular_pairs one wave per touching cell pair (common vertex / edge / identical), lanes over
//                    the Duffy-type quadrature points, butterfly reduction, atomic scatter.
//   k_boundary_*     Omega x Omega^c term (cell x boundary facet).
//   k_scatter_diag   adds the per-cell diagonal blocks, k_mirror symmetrises the cross part.
//
// Reference routines restated per kernel are cited at each kernel (paths under
// /root/reference/nl/PyNucleus_nl; NO = nonlocalOperator_{SCALAR}.pxi, NA = nonlocalAssembly_{SCALAR}.pxi,
// FL2 = fractionalLaplacian2D.pyx, FL1 = fractionalLaplacian1D.pyx, KC = kernelsCy.pyx).
#pragma once
#include "pnl_device.h"

// ---------------------------------------------------------------------------------------------
// kernel function gamma(|x-y|^2)   (KC:75-294)
template <int KT>
__device__ __forceinline__ double kern_eval(const DevKernel &k, double d2) {
    if (KT == 1) {
        // s = 1/2 in 2D: C * d2^(-3/2)
        double r = __builtin_amdgcn_rsq(d2);            // ~2^-26 relative
        double h = 0.5*d2;
        r = r*__builtin_fma(-h*r, r, 1.5);              // Newton, quadratic convergence
        r = r*__builtin_fma(-h*r, r, 1.5);
        r = r*__builtin_fma(-h*r, r, 1.5);
        return (r*r)*(r*k.scale);
    } else {
        if (!(d2 <= k.horizon2)) return 0.;
        if (k.ktype == 0) return k.scale*pow(d2, k.exponent);
        if (k.ktype == 1) return k.scale;
        return k.scale/sqrt(d2);
    }
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

// Same order, decided in fp32 where that is safe: the fp32 value of the ceil() argument is off by < 1e-4, so
// whenever it is further than 1e-3 from an integer the fp64 formula gives the same ceil; otherwise (rare) the
// exact formula above is evaluated.  lh = ln(h), L = |ln(h/H0)| per cell are staged once per tile.
__device__ __forceinline__ int quad_order_fast(const DevFormula &F, double H0, double h1, double h2, float lh1, float lh2,
                                               float L1, float L2, double d2) {
    const float ld = 0.5f*0.69314718056f*__builtin_amdgcn_logf((float)d2);
    const float logdh1 = ld-lh1, logdh2 = ld-lh2;
    const float Lm = fmaxf(L1, L2);
    const float n1 = F.clip ? fmaxf(logdh1, 0.f) : logdh1, n2 = F.clip ? fmaxf(logdh2, 0.f) : logdh2;
    const float c0 = (float)F.c0, a = (float)F.a, b = (float)F.b, e = (float)F.e, den0 = (float)F.den0;
    const float a1 = (c0+a*L2+b*Lm-e*n2)*__builtin_amdgcn_rcpf(fmaxf(logdh1, 0.f)+den0);
    const float a2 = (c0+a*L1+b*Lm-e*n1)*__builtin_amdgcn_rcpf(fmaxf(logdh2, 0.f)+den0);
    const float r1 = rintf(a1), r2 = rintf(a2);
    const bool risky = (a1 > 1.5f && fabsf(a1-r1) < 1e-3f) || (a2 > 1.5f && fabsf(a2-r2) < 1e-3f) || !(a1 == a1) || !(a2 == a2);
    if (risky) return quad_order(F, H0, h1, h2, sqrt(d2));
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
// Distant pair, one pair per lane (NO:722-789, uncut branch).  The reference forms
// contrib[IJ] = vol * sum_k temp[k] PSI[I,k] PSI[J,k] over the tensor rule; with
// PSI = (phi(x_i), -phi(y_j)) this factors into
//   block11[a,b] =  sum_i phi_a(x_i) phi_b(x_i) (sum_j K_ij)
//   block22[a,b] =  sum_j phi_a(y_j) phi_b(y_j) (sum_i K_ij)
//   block12[a,b] = -sum_i phi_a(x_i) sum_j K_ij phi_b(y_j)
// which is what is accumulated here (same numbers, fewer flops).
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

// runtime number of points n; rule tables are read with wave-uniform indices (scalar loads)
template <int DIM, int DPE, int KT>
__device__ __forceinline__ void eval_distant_generic(const DevProblem &P, int off, int n, const double *av, const double *bv,
                                                     PairAcc<DIM, DPE> &R) {
    constexpr int NV = DIM+1;
    const double *__restrict__ bary = P.bary+3*(size_t)off;
    const double *__restrict__ w = P.w+off;
    const double *__restrict__ phi = P.phi+(size_t)off*DPE;
    for (int i = 0; i < n; i++) {
        double x[DIM];
#pragma unroll
        for (int d = 0; d < DIM; d++) {
            double s = 0.;
#pragma unroll
            for (int k = 0; k < NV; k++) s = __builtin_fma(bary[3*i+k], av[k*DIM+d], s);
            x[d] = s;
        }
        const double wi = w[i];
        double r = 0., u[DPE];
#pragma unroll
        for (int b = 0; b < DPE; b++) u[b] = 0.;
        for (int j = 0; j < n; j++) {
            double d2 = 0.;
#pragma unroll
            for (int d = 0; d < DIM; d++) {
                double s = 0.;
#pragma unroll
                for (int k = 0; k < NV; k++) s = __builtin_fma(bary[3*j+k], bv[k*DIM+d], s);
                double t = x[d]-s;
                d2 = __builtin_fma(t, t, d2);
            }
            const double K = (wi*w[j])*kern_eval<KT>(P.k, d2);
            r += K;
            double t[DPE];
#pragma unroll
            for (int b = 0; b < DPE; b++) { t[b] = K*phi[j*DPE+b]; u[b] += t[b]; }
            int e = 0;
#pragma unroll
            for (int a = 0; a < DPE; a++)
#pragma unroll
                for (int b = a; b < DPE; b++) { R.S2[e] = __builtin_fma(t[a], phi[j*DPE+b], R.S2[e]); e++; }
        }
        int e = 0;
#pragma unroll
        for (int a = 0; a < DPE; a++) {
            const double pa = phi[i*DPE+a];
#pragma unroll
            for (int b = 0; b < DPE; b++) R.G[a][b] = __builtin_fma(pa, u[b], R.G[a][b]);
            const double pr = pa*r;
#pragma unroll
            for (int b = a; b < DPE; b++) { R.S1[e] = __builtin_fma(pr, phi[i*DPE+b], R.S1[e]); e++; }
        }
    }
}

// compile-time number of points: y_j and the column sums stay in registers
template <int DIM, int DPE, int KT, int N>
__device__ __forceinline__ void eval_distant_fixed(const DevProblem &P, int off, const double *av, const double *bv,
                                                   PairAcc<DIM, DPE> &R) {
    constexpr int NV = DIM+1;
    const double *__restrict__ bary = P.bary+3*(size_t)off;
    const double *__restrict__ w = P.w+off;
    const double *__restrict__ phi = P.phi+(size_t)off*DPE;
    double y[N][DIM], c[N];
#pragma unroll
    for (int j = 0; j < N; j++) {
        c[j] = 0.;
#pragma unroll
        for (int d = 0; d < DIM; d++) {
            double s = 0.;
#pragma unroll
            for (int k = 0; k < NV; k++) s = __builtin_fma(bary[3*j+k], bv[k*DIM+d], s);
            y[j][d] = s;
        }
    }
#pragma unroll 1
    for (int i = 0; i < N; i++) {
        double x[DIM];
#pragma unroll
        for (int d = 0; d < DIM; d++) {
            double s = 0.;
#pragma unroll
            for (int k = 0; k < NV; k++) s = __builtin_fma(bary[3*i+k], av[k*DIM+d], s);
            x[d] = s;
        }
        const double wi = w[i];
        double r = 0., u[DPE];
#pragma unroll
        for (int b = 0; b < DPE; b++) u[b] = 0.;
#pragma unroll
        for (int j = 0; j < N; j++) {
            double d2 = 0.;
#pragma unroll
            for (int d = 0; d < DIM; d++) { double t = x[d]-y[j][d]; d2 = __builtin_fma(t, t, d2); }
            const double K = (wi*w[j])*kern_eval<KT>(P.k, d2);
            r += K;
            c[j] += K;
#pragma unroll
            for (int b = 0; b < DPE; b++) u[b] = __builtin_fma(K, phi[j*DPE+b], u[b]);
        }
        int e = 0;
#pragma unroll
        for (int a = 0; a < DPE; a++) {
            const double pa = phi[i*DPE+a];
#pragma unroll
            for (int b = 0; b < DPE; b++) R.G[a][b] = __builtin_fma(pa, u[b], R.G[a][b]);
            const double pr = pa*r;
#pragma unroll
            for (int b = a; b < DPE; b++) { R.S1[e] = __builtin_fma(pr, phi[i*DPE+b], R.S1[e]); e++; }
        }
    }
#pragma unroll
    for (int j = 0; j < N; j++) {
        int e = 0;
#pragma unroll
        for (int a = 0; a < DPE; a++) {
            const double pc = phi[j*DPE+a]*c[j];
#pragma unroll
            for (int b = a; b < DPE; b++) { R.S2[e] = __builtin_fma(pc, phi[j*DPE+b], R.S2[e]); e++; }
        }
    }
}

// ---------------------------------------------------------------------------------------------
// Tile kernel: classification (NO:280-378 vertex test, NO:493-540 + FL2:622-642 order) and distant
// evaluation for one TILE x TILE block of cell pairs.
template <int DIM, int DPE, int TILE>
struct TileSmem {
    static constexpr int NV = DIM+1, NC = NV*DIM, ND = DPE*(DPE+1)/2;
    // doubles
    static constexpr int o_v = 0;                          // [2][NC][TILE]
    static constexpr int o_cen = o_v+2*NC*TILE;            // [2][DIM][TILE]
    static constexpr int o_vol = o_cen+2*DIM*TILE;         // [2][TILE]
    static constexpr int o_h = o_vol+2*TILE;               // [2][TILE]
    static constexpr int o_D = o_h+2*TILE;                 // [2][TILE][ND]
    static constexpr int n_dbl = o_D+2*TILE*ND;
    // ints after the doubles
    static constexpr int o_vid = 0;                        // [2][NV][TILE]
    static constexpr int o_cnt = o_vid+2*NV*TILE;          // [PNL_MAXQ+2]
    static constexpr int o_cur = o_cnt+PNL_MAXQ+2;         // [PNL_MAXQ+2]
    static constexpr int o_misc = o_cur+PNL_MAXQ+2;        // [4]: list length, work-list base, #eligible buckets
    static constexpr int o_el = o_misc+4;                  // [2][16]: order and list end of the populated eligible buckets
    static constexpr int o_lh = o_el+32;                   // float [2][2][TILE]: ln h, |ln(h/H0)|
    static constexpr int o_q = o_lh+4*TILE;                // unsigned char [TILE*TILE] order of each pair
    static constexpr int n_int = o_q+TILE*TILE/4;
    // shorts after the ints
    static constexpr int o_slot = 0;                       // [2][DPE][TILE]
    static constexpr int o_list = o_slot+2*DPE*TILE;       // [TILE*TILE]
    static constexpr int n_short = o_list+TILE*TILE;
    static constexpr size_t fixed_bytes = sizeof(double)*n_dbl+sizeof(int)*n_int+sizeof(short)*((n_short+3)/4*4);
};

// number of points of a distant rule the tile kernel integrates one pair per lane (fully unrolled);
// every other order goes to the global work list and is integrated one pair per wave.
__device__ __forceinline__ bool tile_eligible(int n) { return n == 2 || n == 3 || n == 4 || n == 6 || n == 7; }

#ifndef PNL_TILE_WAVES
#define PNL_TILE_WAVES 2      // waves per SIMD the tile kernel is register-limited to (measured: 2 beats 3 and 4)
#endif
template <int DIM, int DPE, int TILE, int KT>
__global__ void __launch_bounds__(PNL_NTHREADS, PNL_TILE_WAVES)
k_tile_distant(const DevProblem P, const int2 *__restrict__ tiles, double *__restrict__ A, long long ldA,
               double *__restrict__ Dglob, int cell_begin, int cell_end, int acc_stride, int4 *__restrict__ worklist,
               unsigned *__restrict__ wl_count, unsigned wl_cap, int ablate) {
    using S = TileSmem<DIM, DPE, TILE>;
    constexpr int NV = S::NV, NC = S::NC, ND = S::ND;
    constexpr int PAIRS = TILE*TILE, PER_THREAD = PAIRS/PNL_NTHREADS;
    extern __shared__ double smem[];
    double *s_dbl = smem;
    int *s_int = (int*)(s_dbl+S::n_dbl);
    short *s_short = (short*)(s_int+S::n_int);
    double *s_acc = (double*)(s_short+(S::n_short+3)/4*4);   // [nA][acc_stride]
    double *s_v = s_dbl+S::o_v, *s_cen = s_dbl+S::o_cen, *s_vol = s_dbl+S::o_vol, *s_h = s_dbl+S::o_h, *s_D = s_dbl+S::o_D;
    int *s_vid = s_int+S::o_vid, *s_cnt = s_int+S::o_cnt, *s_cur = s_int+S::o_cur, *s_misc = s_int+S::o_misc, *s_el = s_int+S::o_el;
    short *s_slot = s_short+S::o_slot;
    unsigned short *s_list = (unsigned short*)(s_short+S::o_list);
    float *s_lh = (float*)(s_int+S::o_lh);
    unsigned char *s_q = (unsigned char*)(s_int+S::o_q);

    const int tid = threadIdx.x;
    const int2 tl = tiles[blockIdx.x];
    const int ta = tl.x, tb = tl.y;
    const int nA = P.blk_ndof[ta], nB = P.blk_ndof[tb];

    // ---- stage cell data of both blocks in LDS (SoA: conflict-free per-lane reads) -------------
    for (int t = tid; t < 2*TILE; t += PNL_NTHREADS) {
        const int side = t/TILE, l = t%TILE;
        const int c = (side ? tb : ta)*TILE+l;
#pragma unroll
        for (int k = 0; k < NC; k++) s_v[(side*NC+k)*TILE+l] = P.cellv[(size_t)k*P.ncp+c];
#pragma unroll
        for (int d = 0; d < DIM; d++) s_cen[(side*DIM+d)*TILE+l] = P.ccen[(size_t)d*P.ncp+c];
        s_vol[side*TILE+l] = P.cvol[c];
        const double hc = P.ch[c];
        s_h[side*TILE+l] = hc;
        s_lh[(side*2+0)*TILE+l] = (float)log(hc);
        s_lh[(side*2+1)*TILE+l] = (float)fabs(log(hc/P.H0));
#pragma unroll
        for (int k = 0; k < NV; k++) s_vid[(side*NV+k)*TILE+l] = P.cvid[(size_t)k*P.ncp+c];
#pragma unroll
        for (int k = 0; k < DPE; k++) s_slot[(side*DPE+k)*TILE+l] = P.cslot[(size_t)k*P.ncp+c];
    }
    for (int t = tid; t < nA*acc_stride; t += PNL_NTHREADS) s_acc[t] = 0.;
    for (int t = tid; t < 2*TILE*ND; t += PNL_NTHREADS) s_D[t] = 0.;
    for (int t = tid; t < 2*(PNL_MAXQ+2); t += PNL_NTHREADS) s_cnt[t] = 0;      // s_cnt and s_cur are adjacent
    __syncthreads();

    // ---- classification ------------------------------------------------------------------------
    // pair p -> (i, j) along wrapped diagonals: consecutive lanes get distinct a-cells AND distinct
    // b-cells, so the per-cell LDS accumulators below see (almost) no same-address conflicts.
    int overflow = 0;
#pragma unroll 1
    for (int it = 0; it < PER_THREAD; it++) {
        const int p = it*PNL_NTHREADS+tid;
        const int j = p%TILE, i = (p/TILE+j)%TILE;
        int q = 0;
        const int va0 = s_vid[(0*NV+0)*TILE+i], vb0 = s_vid[(1*NV+0)*TILE+j];
        const int ca = ta*TILE+i;
        bool ok = (va0 >= 0) && (vb0 >= 0) && (ta < tb || i < j) && (ca >= cell_begin) && (ca < cell_end);
        if (ok) {
            // NA:138-150: skip pairs with boundary DoFs only;  NO:311-323: shared vertices -> singular pair
            bool any_dof = false, shared = false;
#pragma unroll
            for (int k = 0; k < DPE; k++)
                any_dof = any_dof || (s_slot[(0*DPE+k)*TILE+i] >= 0) || (s_slot[(1*DPE+k)*TILE+j] >= 0);
#pragma unroll
            for (int k = 0; k < NV; k++) {
                const int va = s_vid[(0*NV+k)*TILE+i];
#pragma unroll
                for (int m = 0; m < NV; m++) shared = shared || (va == s_vid[(1*NV+m)*TILE+j]);
            }
            if (any_dof && !shared) {
                double d2 = 0.;
#pragma unroll
                for (int d = 0; d < DIM; d++) {
                    const double t = s_cen[(0*DIM+d)*TILE+i]-s_cen[(1*DIM+d)*TILE+j];
                    d2 += t*t;
                }
                q = quad_order_fast(P.qo, P.H0, s_h[i], s_h[TILE+j], s_lh[i], s_lh[2*TILE+j], s_lh[TILE+i], s_lh[3*TILE+j], d2);
                if (q > P.qmax || q > PNL_MAXQ) { overflow++; q = 0; }
            }
        }
        s_q[p] = (unsigned char)q;
        if (q) atomicAdd(&s_cnt[q], 1);
    }
    if (overflow) atomicAdd(&P.counters[5], (unsigned long long)overflow);
    __syncthreads();
    if (tid == 0) {
        // exclusive prefixes over the orders: tile-eligible orders index the LDS list, the others a slice of
        // the global work list reserved with one atomic
        int run = 0, far = 0, nel = 0;
        unsigned long long evals = 0;
        for (int q = 2; q <= P.qmax; q++) {
            const int cq = s_cnt[q];
            if (!cq) continue;
            const int n = P.off[q+1]-P.off[q];
            if (tile_eligible(n) && nel < 16) { s_cur[q] = run; run += cq; s_el[nel] = q; s_el[16+nel] = run; nel++; }
            else { s_cur[q] = far; far += cq; s_cnt[q] = -cq; }
            evals += (unsigned long long)n*n*cq;
            atomicAdd(&P.counters[8+q], (unsigned long long)cq);
        }
        s_misc[0] = run;
        s_misc[2] = nel;
        unsigned base = 0;
        if (far) base = atomicAdd(wl_count, (unsigned)far);
        s_misc[1] = (int)base;
        if (run+far) {
            atomicAdd(&P.counters[1], (unsigned long long)(run+far));
            atomicAdd(&P.counters[2], evals);
        }
    }
    __syncthreads();
    {
        const unsigned base = (unsigned)s_misc[1];
#pragma unroll 1
        for (int it = 0; it < PER_THREAD; it++) {
            const int p = it*PNL_NTHREADS+tid;
            const int q = s_q[p];
            if (q) {
                const int pos = atomicAdd(&s_cur[q], 1);
                if (s_cnt[q] > 0) s_list[pos] = (unsigned short)p;
                else {
                    const unsigned g = base+(unsigned)pos;
                    const int j = p%TILE, i = (p/TILE+j)%TILE;
                    if (g < wl_cap) worklist[g] = make_int4(ta*TILE+i, tb*TILE+j, q, 0);
                }
            }
        }
    }
    __syncthreads();
    // after the fill s_cur[q] = end of bucket q in the LDS list (for eligible q)

    // ---- evaluation: waves take 64-pair chunks of the order-sorted list; inside a chunk the (at most few)
    //      orders present run one after the other so that every lane of a pass has the same trip count ----
    const int total = __builtin_amdgcn_readfirstlane(s_misc[0]);
    const int nel = __builtin_amdgcn_readfirstlane(s_misc[2]);
    const int wave = tid >> 6, lane = tid & 63;
    if (!(ablate & 2))
    for (int c0 = wave*64; c0 < total; c0 += PNL_NTHREADS) {
        const int idx = c0+lane;
        const bool act = idx < total;
        const int p = act ? s_list[idx] : 0;
        const int j = p%TILE, i = (p/TILE+j)%TILE;
        // order of this entry: first q whose bucket end exceeds idx
        int myq = 0;
        if (act)
            for (int k = nel-1; k >= 0; k--)
                if (idx < s_el[16+k]) myq = s_el[k];
        double av[NC], bv[NC];
#pragma unroll
        for (int k = 0; k < NC; k++) { av[k] = s_v[(0*NC+k)*TILE+i]; bv[k] = s_v[(1*NC+k)*TILE+j]; }
        PairAcc<DIM, DPE> R;
        R.clear();
        unsigned long long todo = __ballot(act);
        while (todo) {
            const int src = __ffsll((long long)todo)-1;
            const int q = __shfl(myq, src, 64);
            const bool mine = act && (myq == q);
            todo &= ~__ballot(mine);
            const int off = __builtin_amdgcn_readfirstlane(P.off[q]);
            const int n = __builtin_amdgcn_readfirstlane(P.off[q+1])-off;
            if (mine) {
                if (n == 3) eval_distant_fixed<DIM, DPE, KT, 3>(P, off, av, bv, R);
                else if (n == 6) eval_distant_fixed<DIM, DPE, KT, 6>(P, off, av, bv, R);
                else if (n == 7) eval_distant_fixed<DIM, DPE, KT, 7>(P, off, av, bv, R);
                else if (n == 2) eval_distant_fixed<DIM, DPE, KT, 2>(P, off, av, bv, R);
                else eval_distant_fixed<DIM, DPE, KT, 4>(P, off, av, bv, R);
            }
        }
        if (!act) continue;
        // NA:1405-1410: symmetric cell pairs count twice
        const double vv = 2.*s_vol[i]*s_vol[TILE+j];
        if (ablate & 1) {
            double keep = 0.;
#pragma unroll
            for (int a = 0; a < DPE; a++)
#pragma unroll
                for (int b = 0; b < DPE; b++) keep += R.G[a][b];
#pragma unroll
            for (int e2 = 0; e2 < ND; e2++) keep += R.S1[e2]+R.S2[e2];
            if (keep == 1.2345e300) s_D[0] = keep*vv;
            continue;
        }
        int e = 0;
#pragma unroll
        for (int a = 0; a < DPE; a++) {
            const int sa = s_slot[(0*DPE+a)*TILE+i];
#pragma unroll
            for (int b = 0; b < DPE; b++) {
                const int sb = s_slot[(1*DPE+b)*TILE+j];
                if (sa >= 0 && sb >= 0) lds_add_f64(&s_acc[sa*acc_stride+sb], -vv*R.G[a][b]);
            }
#pragma unroll
            for (int b = a; b < DPE; b++) {
                lds_add_f64(&s_D[(0*TILE+i)*ND+e], vv*R.S1[e]);
                lds_add_f64(&s_D[(1*TILE+j)*ND+e], vv*R.S2[e]);
                e++;
            }
        }
    }
    __syncthreads();

    // ---- flush: sub-block of A' (rows = DoFs of block a, cols = DoFs of block b) and diagonal blocks ----
    const int *__restrict__ dofA = P.blk_dofs+(size_t)ta*P.blk_stride;
    const int *__restrict__ dofB = P.blk_dofs+(size_t)tb*P.blk_stride;
    if (ablate & 4) return;
    for (int t = tid; t < nA*nB; t += PNL_NTHREADS) {
        const int r = t/nB, c = t-r*nB;
        const double v = s_acc[r*acc_stride+c];
        if (v != 0.) atomic_add_f64(&A[(long long)dofA[r]*ldA+dofB[c]], v);
    }
    for (int t = tid; t < 2*TILE*ND; t += PNL_NTHREADS) {
        const double v = s_D[t];
        if (v != 0.) {
            const int side = t/(TILE*ND), rem = t-side*TILE*ND;
            const int c = (side ? tb : ta)*TILE+rem/ND;
            atomic_add_f64(&Dglob[(size_t)c*ND+rem%ND], v);
        }
    }
}

// ---------------------------------------------------------------------------------------------
// wave-wide sum (butterfly), result in every lane
__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) v += __shfl_xor(v, m, 64);
    return v;
}

// Distant pairs of the orders the tile kernel does not unroll (NO:722-789): one wave per pair, lanes over
// the n*n point pairs of the tensor rule, butterfly reduction of the local matrix, atomic scatter.
// A' receives the cross block on the (c1-DoF, c2-DoF) side only, like the tile kernel.
template <int DIM, int DPE, int KT>
__global__ void __launch_bounds__(PNL_NTHREADS)
k_worklist_pairs(const DevProblem P, const int4 *__restrict__ worklist, const unsigned *__restrict__ wl_count, unsigned wl_cap,
                 double *__restrict__ A, long long ldA, double *__restrict__ Dglob) {
    constexpr int NV = DIM+1, NC = NV*DIM, ND = DPE*(DPE+1)/2, NG = DPE*DPE, NACC = NG+2*ND;
    const int lane = threadIdx.x & 63;
    const unsigned nwaves = gridDim.x*(PNL_NTHREADS/64);
    const unsigned count = min(*wl_count, wl_cap);
    for (unsigned item = (blockIdx.x*PNL_NTHREADS+threadIdx.x) >> 6; item < count; item += nwaves) {
        const int4 ent = worklist[item];
        const int c1 = __builtin_amdgcn_readfirstlane(ent.x), c2 = __builtin_amdgcn_readfirstlane(ent.y);
        const int q = __builtin_amdgcn_readfirstlane(ent.z);
        const int off = P.off[q], n = P.off[q+1]-off, nn = n*n;
        const double *__restrict__ bary = P.bary+3*(size_t)off;
        const double *__restrict__ w = P.w+off;
        const double *__restrict__ phi = P.phi+(size_t)off*DPE;
        double av[NC], bv[NC];
#pragma unroll
        for (int k = 0; k < NC; k++) { av[k] = P.cellv[(size_t)k*P.ncp+c1]; bv[k] = P.cellv[(size_t)k*P.ncp+c2]; }
        double acc[NACC];
#pragma unroll
        for (int e = 0; e < NACC; e++) acc[e] = 0.;
        const int di = 64/n, dj = 64-di*n;
        int i = lane/n, j = lane-i*n;
        for (int k = lane; k < nn; k += 64) {
            double d2 = 0.;
#pragma unroll
            for (int d = 0; d < DIM; d++) {
                double x = 0., y = 0.;
#pragma unroll
                for (int m = 0; m < NV; m++) {
                    x = __builtin_fma(bary[3*i+m], av[m*DIM+d], x);
                    y = __builtin_fma(bary[3*j+m], bv[m*DIM+d], y);
                }
                d2 = __builtin_fma(x-y, x-y, d2);
            }
            const double K = (w[i]*w[j])*kern_eval<KT>(P.k, d2);
            double pa[DPE], pb[DPE];
#pragma unroll
            for (int a = 0; a < DPE; a++) { pa[a] = phi[i*DPE+a]; pb[a] = phi[j*DPE+a]; }
            int e = 0;
#pragma unroll
            for (int a = 0; a < DPE; a++) {
                const double ka = K*pa[a];
#pragma unroll
                for (int b = 0; b < DPE; b++) acc[a*DPE+b] = __builtin_fma(ka, pb[b], acc[a*DPE+b]);
                const double kb = K*pb[a];
#pragma unroll
                for (int b = a; b < DPE; b++) {
                    acc[NG+e] = __builtin_fma(ka, pa[b], acc[NG+e]);
                    acc[NG+ND+e] = __builtin_fma(kb, pb[b], acc[NG+ND+e]);
                    e++;
                }
            }
            i += di; j += dj;
            if (j >= n) { j -= n; i++; }
        }
        // reduce; lane (e mod 64) keeps entry e (two per lane at most: NACC <= 128)
        double mine0 = 0., mine1 = 0.;
#pragma unroll
        for (int e = 0; e < NACC; e++) {
            const double s = wave_sum(acc[e]);
            if (e < 64) mine0 = (lane == e) ? s : mine0;
            else mine1 = (lane == e-64) ? s : mine1;
        }
        const double vv = 2.*P.cvol[c1]*P.cvol[c2];
#pragma unroll
        for (int rep = 0; rep < (NACC+63)/64; rep++) {
            const int e = lane+64*rep;
            const double val = rep ? mine1 : mine0;
            if (e < NG) {
                const int a = e/DPE, b = e-a*DPE;
                const int I = P.cdof[(size_t)a*P.ncp+c1], J = P.cdof[(size_t)b*P.ncp+c2];
                if (I >= 0 && J >= 0) atomic_add_f64(&A[(long long)I*ldA+J], -vv*val);
            } else if (e < NG+ND) atomic_add_f64(&Dglob[(size_t)c1*ND+(e-NG)], vv*val);
            else if (e < NACC) atomic_add_f64(&Dglob[(size_t)c2*ND+(e-NG-ND)], vv*val);
        }
    }
}

__device__ __forceinline__ int perm_rank(const int *perm, int n) {
    const int fact[4] = {1, 1, 2, 6};
    int index = 0;
    for (int i = 0; i < n; i++) {
        int smaller = 0;
        for (int j = 0; j < i; j++) smaller += (perm[j] < perm[i]);
        index += (perm[i]-smaller)*fact[n-1-i];
    }
    return index;
}

// Singular pairs: FL2:823-891 / FL1:349-407 with the permutations of NO:280-378 and the symmetric
// scatter NA:204-221.  One wave per pair; SLOT 0 common vertex, 1 common edge, 2 common face.
// ROWS is the number of merged local DoFs (rows of PSI).
template <int DIM, int DPE, int SLOT, int KT>
__global__ void __launch_bounds__(PNL_NTHREADS)
k_singular_pairs(const DevProblem P, const int2 *__restrict__ pairs, int npairs, double *__restrict__ A, long long ldA,
                 int cell_begin, int cell_end) {
    constexpr int NV = DIM+1;
    constexpr int DPV = 1, DPED = (DIM == 2 && DPE == 6) ? 1 : 0;
    constexpr int COMMON = SLOT+1;
    constexpr int ROWS = (COMMON == NV) ? DPE : (COMMON == 1 ? 2*DPE-DPV : 2*DPE-2*DPV-DPED);
    constexpr int NE = ROWS*(ROWS+1)/2;
    const int lane = threadIdx.x & 63;
    const int wid = (blockIdx.x*PNL_NTHREADS+threadIdx.x) >> 6;
    if (wid >= npairs) return;
    const int2 pr = pairs[wid];
    const int c1 = pr.x, c2 = pr.y;
    if (c1 < cell_begin || c1 >= cell_end) return;
    // NA:138-150
    int ld[2*DPE];
    bool any = false;
#pragma unroll
    for (int k = 0; k < DPE; k++) {
        ld[k] = P.cdof[(size_t)k*P.ncp+c1];
        ld[DPE+k] = P.cdof[(size_t)k*P.ncp+c2];
        any = any || ld[k] >= 0 || ld[DPE+k] >= 0;
    }
    if (!any) return;
    // vertex permutations, shared vertices first (NO:311-346)
    int perm1[NV], perm2[NV], perm[2*DPE];
#pragma unroll
    for (int k = 0; k < NV; k++) { perm1[k] = k; perm2[k] = k; }
#pragma unroll
    for (int k = 0; k < 2*DPE; k++) perm[k] = k;
    if (c1 != c2) {
        int mask1 = 0, mask2 = 0, common = 0;
        for (int a = 0; a < NV; a++) {
            const int v1 = P.cvid[(size_t)a*P.ncp+c1];
            for (int b = 0; b < NV; b++) {
                if (mask2 & (1 << b)) continue;
                if (v1 == P.cvid[(size_t)b*P.ncp+c2]) {
                    perm1[common] = a; perm2[common] = b;
                    mask1 += (1 << a); mask2 += (1 << b);
                    common++;
                    break;
                }
            }
        }
        int i = 0;
        for (int k = common; k < NV; k++) { while (mask1 & (1 << i)) i++; perm1[k] = i; mask1 += (1 << i); }
        i = 0;
        for (int k = common; k < NV; k++) { while (mask2 & (1 << i)) i++; perm2[k] = i; mask2 += (1 << i); }
        const int *t1 = P.perm_table+perm_rank(perm1, NV)*DPE;
        const int *t2 = P.perm_table+perm_rank(perm2, NV)*DPE;
        for (int k = 0; k < DPE; k++) perm[k] = t1[k];
        if (COMMON == 1) {
            for (int k = DPV; k < DPE; k++) perm[DPE+k-DPV] = DPE+t2[k];
        } else if (COMMON == 2) {
            for (int k = 2*DPV; k < NV*DPV; k++) perm[DPE+k-2*DPV] = DPE+t2[k];
            for (int k = NV*DPV+DPED; k < DPE; k++) perm[DPE+k-2*DPV-DPED] = DPE+t2[k];
        }
    }
    // permuted simplices
    double s1[NV][DIM], s2[NV][DIM];
#pragma unroll
    for (int k = 0; k < NV; k++)
#pragma unroll
        for (int d = 0; d < DIM; d++) {
            // runtime permutation index -> select with compares to stay in registers
            double a = 0., b = 0.;
#pragma unroll
            for (int m = 0; m < NV; m++) {
                const double va = P.cellv[(size_t)(m*DIM+d)*P.ncp+c1], vb = P.cellv[(size_t)(m*DIM+d)*P.ncp+c2];
                a = (perm1[k] == m) ? va : a;
                b = (perm2[k] == m) ? vb : b;
            }
            s1[k][d] = a; s2[k][d] = b;
        }
    const int M = P.sM[SLOT];
    const double *__restrict__ nodes = P.sNodes[SLOT];
    const double *__restrict__ w = P.sW[SLOT];
    const double *__restrict__ psi = P.sPsi[SLOT];
    double acc[NE];
#pragma unroll
    for (int e = 0; e < NE; e++) acc[e] = 0.;
    for (int m = lane; m < M; m += 64) {
        double d2 = 0.;
#pragma unroll
        for (int d = 0; d < DIM; d++) {
            double x = 0., y = 0.;
#pragma unroll
            for (int k = 0; k < NV; k++) {
                x = __builtin_fma(s1[k][d], nodes[(size_t)k*M+m], x);
                y = __builtin_fma(s2[k][d], nodes[(size_t)(NV+k)*M+m], y);
            }
            d2 = __builtin_fma(x-y, x-y, d2);
        }
        const double t = w[m]*kern_eval<KT>(P.k, d2);
        double ps[ROWS];
#pragma unroll
        for (int r = 0; r < ROWS; r++) ps[r] = psi[(size_t)r*M+m];
        int e = 0;
#pragma unroll
        for (int I = 0; I < ROWS; I++) {
            const double tI = t*ps[I];
#pragma unroll
            for (int J = I; J < ROWS; J++) { acc[e] = __builtin_fma(tI, ps[J], acc[e]); e++; }
        }
    }
    const double vol = P.sFac*P.cvol[c1]*P.cvol[c2]*(c1 == c2 ? 1. : 2.);
    // reduce and let lane e scatter entry e
    double mine = 0.;
    int myI = 0, myJ = 0;
    {
        int e = 0;
#pragma unroll
        for (int I = 0; I < ROWS; I++)
#pragma unroll
            for (int J = I; J < ROWS; J++) {
                const double s = wave_sum(acc[e]);
                if (lane == e) { mine = s; myI = I; myJ = J; }
                e++;
            }
    }
    if (lane < NE) {
        int gi = -1, gj = -1;
#pragma unroll
        for (int k = 0; k < 2*DPE; k++) {
            // perm[] and ld[] with runtime indices -> selects
            int pk = perm[k];
            int g = -1;
#pragma unroll
            for (int m = 0; m < 2*DPE; m++) g = (pk == m) ? ld[m] : g;
            gi = (myI == k) ? g : gi;
            gj = (myJ == k) ? g : gj;
        }
        const double v = mine*vol;
        if (gi >= 0 && gj >= 0) {
            if (myI == myJ) atomic_add_f64(&A[(long long)gi*ldA+gi], v);
            else {
                atomic_add_f64(&A[(long long)gi*ldA+gj], v);
                atomic_add_f64(&A[(long long)gj*ldA+gi], v);
            }
        }
    }
    if (lane == 0) {
        atomicAdd(&P.counters[1], 1ull);
        atomicAdd(&P.counters[2], (unsigned long long)M);
        atomicAdd(&P.counters[128+SLOT], 1ull);
    }
}

// ---------------------------------------------------------------------------------------------
// Omega x Omega^c, distant part: one thread per cell, loops over a chunk of boundary facets
// (NA:1430-1448 loop, NO:1022-1108 eval_distant_boundary, order FL2:1226-1243 / FL1:646-660).
template <int DIM, int DPE, int KT>
__global__ void __launch_bounds__(PNL_NTHREADS)
k_boundary_distant(const DevProblem P, double *__restrict__ Dglob, int cell_begin, int cell_end, int facets_per_chunk) {
    constexpr int NV = DIM+1, NC = NV*DIM, NF = DIM, ND = DPE*(DPE+1)/2;
    const int c = cell_begin+blockIdx.x*PNL_NTHREADS+threadIdx.x;
    const bool active = c < cell_end;
    const int cc = active ? c : cell_begin;
    double av[NC], cen[DIM];
    int vid[NV];
#pragma unroll
    for (int k = 0; k < NC; k++) av[k] = P.cellv[(size_t)k*P.ncp+cc];
#pragma unroll
    for (int d = 0; d < DIM; d++) cen[d] = P.ccen[(size_t)d*P.ncp+cc];
#pragma unroll
    for (int k = 0; k < NV; k++) vid[k] = P.cvid[(size_t)k*P.ncp+cc];
    const double h1 = P.ch[cc], vol1 = P.cvol[cc];
    double D[ND];
#pragma unroll
    for (int e = 0; e < ND; e++) D[e] = 0.;
    unsigned long long npairs = 0, nevals = 0;
    int overflow = 0;
    const int f0 = blockIdx.y*facets_per_chunk;
    const int f1 = min(P.nb, f0+facets_per_chunk);
    for (int f = f0; f < f1; f++) {
        // facet data: wave-uniform
        double fv[NF*DIM], fc[DIM], nrm[DIM];
        int fvid[NF];
#pragma unroll
        for (int k = 0; k < NF; k++) fvid[k] = P.bvid[(size_t)k*P.nb+f];
#pragma unroll
        for (int k = 0; k < NF*DIM; k++) fv[k] = P.bv[(size_t)k*P.nb+f];
        double vol2 = 1.;
#pragma unroll
        for (int d = 0; d < DIM; d++) {
            double s = 0.;
#pragma unroll
            for (int k = 0; k < NF; k++) s += fv[k*DIM+d];
            fc[d] = s*(1./NF);
        }
        if (DIM == 2) {
            nrm[0] = fv[1*DIM+1]-fv[0*DIM+1];
            nrm[1] = fv[0*DIM+0]-fv[1*DIM+0];
            const double l2 = nrm[0]*nrm[0]+nrm[1]*nrm[1];
            const double inv = 1./sqrt(l2);
            nrm[0] *= inv; nrm[1] *= inv;
            vol2 = sqrt((fv[2]-fv[0])*(fv[2]-fv[0])+(fv[3]-fv[1])*(fv[3]-fv[1]));
        }
        bool shared = false;
#pragma unroll
        for (int k = 0; k < NV; k++)
#pragma unroll
            for (int m = 0; m < NF; m++) shared = shared || (vid[k] == fvid[m]);
        if (!active || shared) continue;
        double dc2 = 0.;
#pragma unroll
        for (int d = 0; d < DIM; d++) dc2 += (cen[d]-fc[d])*(cen[d]-fc[d]);
        const int q = quad_order(P.bqo, P.H0, h1, vol2, sqrt(dc2));
        if (q > P.qmax || q > PNL_MAXQ) { overflow++; continue; }
        const int off = P.off[q], n = P.off[q+1]-off;
        const int foff = P.foff[q], nf = P.foff[q+1]-foff;
        const double *__restrict__ bary = P.bary+3*(size_t)off;
        const double *__restrict__ w = P.w+off;
        const double *__restrict__ phi = P.phi+(size_t)off*DPE;
        const double *__restrict__ fb = P.fbary+2*(size_t)foff;
        const double *__restrict__ fw = P.fw+foff;
        npairs++;
        nevals += (unsigned long long)n*nf;
        const double vol = vol1*vol2;
        for (int k = 0; k < n; k++) {
            double x[DIM];
#pragma unroll
            for (int d = 0; d < DIM; d++) {
                double s = 0.;
#pragma unroll
                for (int m = 0; m < NV; m++) s = __builtin_fma(bary[3*k+m], av[m*DIM+d], s);
                x[d] = s;
            }
            double r = 0.;
            for (int m = 0; m < nf; m++) {
                double d2 = 0., nw = 0.;
#pragma unroll
                for (int d = 0; d < DIM; d++) {
                    double y = 0.;
#pragma unroll
                    for (int t = 0; t < NF; t++) y = __builtin_fma(fb[2*m+t], fv[t*DIM+d], y);
                    const double wv = y-x[d];
                    d2 = __builtin_fma(wv, wv, d2);
                    if (DIM == 2) nw = __builtin_fma(nrm[d], wv, nw);
                }
                if (DIM == 2) nw *= 1./sqrt(d2); else nw = 1.;
                r = __builtin_fma(fw[m]*nw, kern_eval<KT>(P.bk, d2), r);
            }
            r *= w[k]*vol;
            int e = 0;
#pragma unroll
            for (int a = 0; a < DPE; a++) {
                const double pa = phi[k*DPE+a]*r;
#pragma unroll
                for (int b = a; b < DPE; b++) { D[e] = __builtin_fma(pa, phi[k*DPE+b], D[e]); e++; }
            }
        }
    }
    if (active) {
#pragma unroll
        for (int e = 0; e < ND; e++)
            if (D[e] != 0.) atomic_add_f64(&Dglob[(size_t)c*ND+e], D[e]);
        if (npairs) {
            atomicAdd(&P.counters[3], npairs);
            atomicAdd(&P.counters[4], nevals);
        }
        if (overflow) atomicAdd(&P.counters[5], (unsigned long long)overflow);
    }
}

// Omega x Omega^c, touching cell/facet pairs (FL2:1324-1407, FL1:726-785); one wave per pair.
template <int DIM, int DPE, int SLOT, int KT>
__global__ void __launch_bounds__(PNL_NTHREADS)
k_boundary_singular(const DevProblem P, const int2 *__restrict__ pairs, int npairs, double *__restrict__ Dglob,
                    int cell_begin, int cell_end) {
    constexpr int NV = DIM+1, NF = DIM, ND = DPE*(DPE+1)/2;
    const int lane = threadIdx.x & 63;
    const int wid = (blockIdx.x*PNL_NTHREADS+threadIdx.x) >> 6;
    if (wid >= npairs) return;
    const int c1 = pairs[wid].x, f = pairs[wid].y;
    if (c1 < cell_begin || c1 >= cell_end) return;
    int perm1[NV], perm2[NF], perm[DPE];
#pragma unroll
    for (int k = 0; k < NV; k++) perm1[k] = k;
#pragma unroll
    for (int k = 0; k < NF; k++) perm2[k] = k;
    {
        int mask1 = 0, mask2 = 0, common = 0;
        for (int a = 0; a < NV; a++) {
            const int v1 = P.cvid[(size_t)a*P.ncp+c1];
            for (int b = 0; b < NF; b++) {
                if (mask2 & (1 << b)) continue;
                if (v1 == P.bvid[(size_t)b*P.nb+f]) {
                    perm1[common] = a; perm2[common] = b;
                    mask1 += (1 << a); mask2 += (1 << b);
                    common++;
                    break;
                }
            }
        }
        int i = 0;
        for (int k = common; k < NV; k++) { while (mask1 & (1 << i)) i++; perm1[k] = i; mask1 += (1 << i); }
        i = 0;
        for (int k = common; k < NF; k++) { while (mask2 & (1 << i)) i++; perm2[k] = i; mask2 += (1 << i); }
        const int *t1 = P.perm_table+perm_rank(perm1, NV)*DPE;
        for (int k = 0; k < DPE; k++) perm[k] = t1[k];
    }
    double s1[NV][DIM], s2[NF][DIM], fv[NF][DIM], nrm[DIM];
#pragma unroll
    for (int k = 0; k < NF; k++)
#pragma unroll
        for (int d = 0; d < DIM; d++) fv[k][d] = P.bv[(size_t)(k*DIM+d)*P.nb+f];
#pragma unroll
    for (int k = 0; k < NV; k++)
#pragma unroll
        for (int d = 0; d < DIM; d++) {
            double a = 0.;
#pragma unroll
            for (int m = 0; m < NV; m++) a = (perm1[k] == m) ? P.cellv[(size_t)(m*DIM+d)*P.ncp+c1] : a;
            s1[k][d] = a;
        }
#pragma unroll
    for (int k = 0; k < NF; k++)
#pragma unroll
        for (int d = 0; d < DIM; d++) {
            double b = 0.;
#pragma unroll
            for (int m = 0; m < NF; m++) b = (perm2[k] == m) ? fv[m][d] : b;
            s2[k][d] = b;
        }
    double vol2 = 1.;
    if (DIM == 2) {
        nrm[0] = fv[1][1]-fv[0][1];
        nrm[1] = fv[0][0]-fv[1][0];
        const double inv = 1./sqrt(nrm[0]*nrm[0]+nrm[1]*nrm[1]);
        nrm[0] *= inv; nrm[1] *= inv;
        vol2 = sqrt((fv[1][0]-fv[0][0])*(fv[1][0]-fv[0][0])+(fv[1][1]-fv[0][1])*(fv[1][1]-fv[0][1]));
    }
    const int M = P.bM[SLOT];
    const double *__restrict__ nodes = P.bNodes[SLOT];
    const double *__restrict__ w = P.bW[SLOT];
    const double *__restrict__ PHI = P.bPhi[SLOT];
    double acc[ND];
#pragma unroll
    for (int e = 0; e < ND; e++) acc[e] = 0.;
    for (int m = lane; m < M; m += 64) {
        double d2 = 0., nw = 0.;
#pragma unroll
        for (int d = 0; d < DIM; d++) {
            double x = 0., y = 0.;
#pragma unroll
            for (int k = 0; k < NV; k++) x = __builtin_fma(s1[k][d], nodes[(size_t)k*M+m], x);
#pragma unroll
            for (int k = 0; k < NF; k++) y = __builtin_fma(s2[k][d], nodes[(size_t)(NV+k)*M+m], y);
            const double wv = x-y;
            d2 = __builtin_fma(wv, wv, d2);
            if (DIM == 2) nw = __builtin_fma(nrm[d], wv, nw);
        }
        if (DIM == 2) nw *= 1./sqrt(d2); else nw = 1.;
        const double t = w[m]*nw*kern_eval<KT>(P.bk, d2);
        double ps[DPE];
#pragma unroll
        for (int r = 0; r < DPE; r++) ps[r] = PHI[(size_t)r*M+m];
        int e = 0;
#pragma unroll
        for (int I = 0; I < DPE; I++) {
            const double tI = t*ps[I];
#pragma unroll
            for (int J = I; J < DPE; J++) { acc[e] = __builtin_fma(tI, ps[J], acc[e]); e++; }
        }
    }
    const double vol = (DIM == 2) ? P.bFac*P.cvol[c1]*vol2 : P.bFac*P.cvol[c1];
    double mine = 0.;
    int myI = 0, myJ = 0;
    {
        int e = 0;
#pragma unroll
        for (int I = 0; I < DPE; I++)
#pragma unroll
            for (int J = I; J < DPE; J++) {
                const double s = wave_sum(acc[e]);
                if (lane == e) { mine = s; myI = I; myJ = J; }
                e++;
            }
    }
    if (lane < ND) {
        // local (cell) indices i = perm[I], j = perm[J]; flattened index of (min, max)
        int i = 0, j = 0;
#pragma unroll
        for (int k = 0; k < DPE; k++) { i = (myI == k) ? perm[k] : i; j = (myJ == k) ? perm[k] : j; }
        const int lo = min(i, j), hi = max(i, j);
        const int kk = DPE*lo-(lo*(lo+1) >> 1)+hi;
        atomic_add_f64(&Dglob[(size_t)c1*ND+kk], mine*vol);
    }
    if (lane == 0) {
        atomicAdd(&P.counters[3], 1ull);
        atomicAdd(&P.counters[4], (unsigned long long)M);
    }
}

// ---------------------------------------------------------------------------------------------
// A[dof_a, dof_b] += D_c[a,b] for every cell (NA:152-168 / the diagonal blocks of NA:204-221)
template <int DPE>
__global__ void __launch_bounds__(PNL_NTHREADS)
k_scatter_diag(const DevProblem P, const double *__restrict__ Dglob, double *__restrict__ A, long long ldA) {
    constexpr int ND = DPE*(DPE+1)/2;
    const int t = blockIdx.x*PNL_NTHREADS+threadIdx.x;
    const int c = t/(DPE*DPE);
    if (c >= P.nc) return;
    const int ab = t-c*DPE*DPE, a = ab/DPE, b = ab-a*DPE;
    const int lo = min(a, b), hi = max(a, b);
    const double v = Dglob[(size_t)c*ND+DPE*lo-(lo*(lo+1) >> 1)+hi];
    const int I = P.cdof[(size_t)a*P.ncp+c], J = P.cdof[(size_t)b*P.ncp+c];
    if (v != 0. && I >= 0 && J >= 0) atomic_add_f64(&A[(long long)I*ldA+J], v);
}

// A <- A + A^T on the strict off-diagonal (cross contributions were written on one side only)
__global__ void __launch_bounds__(PNL_NTHREADS)
k_mirror(double *__restrict__ A, long long ldA, int N) {
    __shared__ double t1[32][33], t2[32][33];
    const int nb = (N+31)/32;
    // linear block id over the upper block triangle
    int bid = blockIdx.x;
    int bi = 0;
    {
        // solve bi from bid = bi*nb - bi(bi-1)/2 + (bj-bi)
        double fb = ((2.*nb+1.)-sqrt((2.*nb+1.)*(2.*nb+1.)-8.*bid))*0.5;
        bi = (int)fb;
        while (bi > 0 && (long long)bi*nb-(long long)bi*(bi-1)/2 > bid) bi--;
        while ((long long)(bi+1)*nb-(long long)(bi+1)*bi/2 <= bid) bi++;
    }
    const int bj = bi+(bid-(int)((long long)bi*nb-(long long)bi*(bi-1)/2));
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;     // 32 x 8
    for (int r = ty; r < 32; r += 8) {
        const int I = bi*32+r, J = bj*32+tx;
        t1[r][tx] = (I < N && J < N) ? A[(long long)I*ldA+J] : 0.;
        const int I2 = bj*32+r, J2 = bi*32+tx;
        t2[r][tx] = (I2 < N && J2 < N) ? A[(long long)I2*ldA+J2] : 0.;
    }
    __syncthreads();
    for (int r = ty; r < 32; r += 8) {
        const int I = bi*32+r, J = bj*32+tx;
        if (I < N && J < N) {
            if (bi != bj) A[(long long)I*ldA+J] = t1[r][tx]+t2[tx][r];
            else if (r != tx) A[(long long)I*ldA+J] = t1[r][tx]+t1[tx][r];
        }
        const int I2 = bj*32+r, J2 = bi*32+tx;
        if (bi != bj && I2 < N && J2 < N) A[(long long)I2*ldA+J2] = t2[r][tx]+t1[tx][r];
    }
}
