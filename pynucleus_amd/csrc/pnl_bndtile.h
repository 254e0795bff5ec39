// Omega x Omega^c, distant part, tiled (gfx950 only; 2D).
//
// Reference: the 'zeroExterior' loop of getDense (nonlocalAssembly_{SCALAR}.pxi:1430-1448: every cell against every facet of the
// domain boundary), eval_distant_boundary (nonlocalOperator_{SCALAR}.pxi:1022-1108), the order of the pair FL2:1226-1243.
//
// k_boundary_distant (pnl_kernels.h) reads the rule of every (cell, facet) pair from global memory point by point -- lanes of one
// wave in different orders, every read a dependent round trip through the L2 -- and so lives on waves in flight: 11 ms beside the
// HBM-bound fold pass at 98,304 cells x 768 facets for 1e10 useful operations (VALU issue utilisation 0.065).  Here a
// workgroup owns 256 cells (lane = cell, its vertices and logs in registers for the whole launch) and walks its share of the
// facets in chunks of 64 staged in LDS (vertices, unit normal, centre, length, logs, vertex ids: broadcast reads); the rules up to
// order PNL_BT_QI (cell points with weights and shape functions, facet points) are staged once per workgroup.  The local matrix
// factorises like the tile kernels' diagonal blocks: D[a,b] = vol sum_i w_i phi_a phi_b(x_i) r_i with r_i = sum_m w_m n.(y_m - x_i)
// Gamma(x_i, y_m) summed over ALL facets that take the same cell rule -- orders 2 and 3 (nearly every pair) keep their row sums in
// registers and form D once at the end; other orders add into D per point.  Pairs beyond the staged rules or with more than
// defer_evals point pairs go to the list k_boundary_items integrates one per wave, as before.
#pragma once
#include "pnl_kernels.h"

#define PNL_BT_FB 64
#define PNL_BT_QI 8

struct BndTileLds {          // offsets in doubles into the dynamic LDS block (host: bnd_tile_lds)
    int rule, frule, facets, ints, cls, total;
};

__host__ __device__ inline BndTileLds bnd_tile_layout(int npts, int nfp, int st, int ncls) {
    BndTileLds L;
    L.rule = 0;
    L.frule = L.rule+npts*st;
    L.facets = L.frule+nfp*3;
    L.facets += L.facets & 1;                                        // 16-byte alignment of everything behind (int4 table)
    L.ints = L.facets+10*PNL_BT_FB;                                  // float lh2[FB], int fvid[2][FB], int blab[FB]: 4 FB words = 2 FB doubles;
    L.cls = L.ints+2*PNL_BT_FB+2*(PNL_BT_QI+1);                      // then int4 qtab[QI+1]: points, rule offset, facet points, facet-rule offset per order
    L.total = L.cls+ncls*(int)((sizeof(DevKernel)+sizeof(DevFormula)+7)/8);
    return L;
}

template <int DPE, int KTAG>
__device__ __forceinline__ double bt_point(const double *__restrict__ rk, const double *__restrict__ s_fr, int foff, int nf,
                                           const double *av, double f0x, double f0y, double f1x, double f1y, double nx, double ny,
                                           const DevKernel &bkn) {
    // r = sum_m w_m n.(y_m - x) Gamma_b(|y_m - x|^2) at the cell point with barycentric coordinates rk[0..2]
    const double x0 = __builtin_fma(rk[2], av[4], __builtin_fma(rk[1], av[2], rk[0]*av[0]));
    const double x1 = __builtin_fma(rk[2], av[5], __builtin_fma(rk[1], av[3], rk[0]*av[1]));
    double r = 0.;
    for (int m = 0; m < nf; m++) {
        const double *fr = s_fr+3*(foff+m);
        const double w0 = __builtin_fma(fr[1], f1x, fr[0]*f0x)-x0, w1 = __builtin_fma(fr[1], f1y, fr[0]*f0y)-x1;
        const double d2 = __builtin_fma(w1, w1, w0*w0);
        const double nw = __builtin_fma(ny, w1, nx*w0);
        r = __builtin_fma(fr[2]*nw, kern_eval<KTAG, true>(bkn, d2), r);
    }
    return r;
}

template <int DIM, int DPE, int KT>
__global__ void __launch_bounds__(PNL_NTHREADS)
k_boundary_tile(const DevProblem P, double *__restrict__ Dglob, int cell_begin, int cell_end, int facets_per_block, int qi,
                const DevKernel *__restrict__ bkcls, const DevFormula *__restrict__ bfcls, int ncls, int defer_evals,
                int *__restrict__ dcells, int *__restrict__ dfacets, unsigned *__restrict__ dslots, unsigned *__restrict__ dcount,
                unsigned dcap, int *__restrict__ dcls) {
    static_assert(DIM == 2, "tiled boundary term: 2D");
    constexpr int NV = 3, NF = 2, ND = DPE*(DPE+1)/2, ST = 4+DPE, FB = PNL_BT_FB;
    extern __shared__ double s_mem[];
    const int tid = threadIdx.x;
    const int roff0 = P.off[2], npts = P.off[qi+1]-roff0, foff0 = P.foff[2], nfp = P.foff[qi+1]-foff0;
    const BndTileLds LY = bnd_tile_layout(npts, nfp, ST, bkcls ? ncls : 0);
    double *s_rule = s_mem+LY.rule, *s_fr = s_mem+LY.frule, *s_fd = s_mem+LY.facets;
    float *s_lh2 = (float*)(s_mem+LY.ints);
    int *s_fvid = (int*)(s_lh2+FB), *s_blab = s_fvid+2*FB;
    int4 *s_qtab = (int4*)(s_mem+LY.ints+2*FB);          // per order q <= qi: (n, rule offset in s_rule, nf, offset in s_fr): one LDS read per pair
    DevKernel *s_bk = (DevKernel*)(s_mem+LY.cls);
    DevFormula *s_bf = (DevFormula*)(s_bk+(bkcls ? ncls : 0));
    for (int t = tid; t < npts*ST; t += PNL_NTHREADS) {
        const int pt = t/ST, k = t-pt*ST;
        s_rule[t] = k < 3 ? P.bary[3*(size_t)(roff0+pt)+k] : (k == 3 ? P.w[roff0+pt] : P.phi[(size_t)(roff0+pt)*DPE+k-4]);
    }
    for (int t = tid; t < nfp*3; t += PNL_NTHREADS) {
        const int pt = t/3, k = t-pt*3;
        s_fr[t] = k < 2 ? P.fbary[2*(size_t)(foff0+pt)+k] : P.fw[foff0+pt];
    }
    if (bkcls)
        for (int t = tid; t < ncls; t += PNL_NTHREADS) { s_bk[t] = bkcls[t]; s_bf[t] = bfcls[t]; }
    for (int t = tid; t <= PNL_BT_QI; t += PNL_NTHREADS)
        s_qtab[t] = (t >= 2 && t <= qi) ? make_int4(P.off[t+1]-P.off[t], P.off[t]-roff0, P.foff[t+1]-P.foff[t], P.foff[t]-foff0) : make_int4(0, 0, 0, 0);
    const int c = cell_begin+blockIdx.x*PNL_NTHREADS+tid;
    bool active = c < cell_end;
    const int cc = active ? c : cell_begin;
    double av[6];
    int vid[NV];
#pragma unroll
    for (int k = 0; k < 6; k++) av[k] = P.cellv[(size_t)k*P.ncp+cc];
    const double cen0 = P.ccen[cc], cen1 = P.ccen[(size_t)P.ncp+cc];
#pragma unroll
    for (int k = 0; k < NV; k++) vid[k] = P.cvid[(size_t)k*P.ncp+cc];
    active = active && vid[0] >= 0;                      // zero-volume padding cells inside the mesh meet no facet
    const double h1 = P.ch[cc], vol1 = P.cvol[cc];
    const double Ld1 = fabs(log(h1/P.H0));
    const float lh1 = (float)log(h1), L1 = (float)Ld1;
    const int lab1 = P.cur_class >= 0 ? P.clabel[cc] : 0;
    // the cell rules of orders 2 and 3 have 3 and 6 points on triangles (checked on the host: qi2 / qi3 say so)
    const bool q2reg = P.off[3]-P.off[2] == 3, q3reg = qi >= 3 && P.off[4]-P.off[3] == 6;
    double R2[3] = {0., 0., 0.}, R3[6] = {0., 0., 0., 0., 0., 0.}, D[ND];
#pragma unroll
    for (int e = 0; e < ND; e++) D[e] = 0.;
    unsigned long long npairs = 0, nevals = 0;
    int overflow = 0;
    const int f_begin = blockIdx.y*facets_per_block, f_end = min(P.nb, f_begin+facets_per_block);
    for (int fb0 = f_begin; fb0 < f_end; fb0 += FB) {
        const int cnt = min(FB, f_end-fb0);
        __syncthreads();                                 // the previous chunk (and the rule staging) is done with
        for (int t = tid; t < cnt; t += PNL_NTHREADS) {
            const int f = fb0+t;
            s_fd[0*FB+t] = P.bv[(size_t)0*P.nb+f]; s_fd[1*FB+t] = P.bv[(size_t)1*P.nb+f];
            s_fd[2*FB+t] = P.bv[(size_t)2*P.nb+f]; s_fd[3*FB+t] = P.bv[(size_t)3*P.nb+f];
            s_fd[4*FB+t] = P.bgeo[(size_t)2*P.nb+f]; s_fd[5*FB+t] = P.bgeo[(size_t)3*P.nb+f];          // unit normal
            s_fd[6*FB+t] = P.bgeo[(size_t)0*P.nb+f]; s_fd[7*FB+t] = P.bgeo[(size_t)1*P.nb+f];          // centre
            s_fd[8*FB+t] = P.bgeo[(size_t)4*P.nb+f]; s_fd[9*FB+t] = P.bgeo[(size_t)5*P.nb+f];          // length, |ln(len / H0)|
            s_lh2[t] = (float)P.bgeo[(size_t)6*P.nb+f];                                                 // ln(len)
            s_fvid[t] = P.bvid[f]; s_fvid[FB+t] = P.bvid[(size_t)P.nb+f];
            s_blab[t] = P.cur_class >= 0 ? P.blabel[f] : 0;
        }
        __syncthreads();
        if (!active) continue;
        for (int j = 0; j < cnt; j++) {
            // variable order: with class tables ONE launch integrates every (cell, facet) with the kernel and order formula of the
            // pair's class; without them this launch handles class P.cur_class and skips the other pairs
            int kc = -1;
            if (P.cur_class >= 0) {
                kc = P.cls_of[lab1*P.nlab+s_blab[j]];
                if (!bkcls && kc != P.cur_class) continue;
            }
            const DevKernel &bkn = bkcls ? s_bk[kc] : P.bkn;
            const DevFormula &bqo = bkcls ? s_bf[kc] : P.bqo;
            const int v0 = s_fvid[j], v1 = s_fvid[FB+j];
            if (vid[0] == v0 || vid[1] == v0 || vid[2] == v0 || vid[0] == v1 || vid[1] == v1 || vid[2] == v1) continue;
            const double f0x = s_fd[0*FB+j], f0y = s_fd[1*FB+j], f1x = s_fd[2*FB+j], f1y = s_fd[3*FB+j];
            const double nx = s_fd[4*FB+j], ny = s_fd[5*FB+j];
            const double u0 = cen0-s_fd[6*FB+j], u1 = cen1-s_fd[7*FB+j];
            const double vol2 = s_fd[8*FB+j], Ld2 = s_fd[9*FB+j];
            const int q = quad_order_fast(bqo, h1, vol2, lh1, s_lh2[j], L1, (float)Ld2, Ld1, Ld2, __builtin_fma(u1, u1, u0*u0));
            if (q > P.qmax || q > PNL_MAXQ) { overflow++; continue; }
            int n, nf, ro = 0, fo = 0;
            if (q <= qi) { const int4 qt = s_qtab[q]; n = qt.x; ro = qt.y; nf = qt.z; fo = qt.w; }
            else { n = P.off[q+1]-P.off[q]; nf = P.foff[q+1]-P.foff[q]; }
            if (q > qi || (dcells && n*nf > defer_evals)) {
                // beyond the staged rules, or a rule of hundreds of point pairs next to the boundary: one per wave (k_boundary_items)
                const unsigned idx = dcells ? atomicAdd(dcount, 1u) : dcap;
                if (idx < dcap) {
                    dcells[idx] = c;
                    dslots[idx] = (unsigned)c;
                    if (bkcls) dcls[idx] = kc;
                    dfacets[(size_t)idx*NF] = v0; dfacets[(size_t)idx*NF+1] = v1;
                    continue;
                }
                if (q > qi) {
                    // no room in the list: in place from the global tables (as k_boundary_distant does)
                    npairs++;
                    nevals += (unsigned long long)n*nf;
                    const int off = P.off[q], foff = P.foff[q];
                    const double vs = vol2*kern_scale<KT>(bkn);
                    for (int k = 0; k < n; k++) {
                        const double b0 = P.bary[3*(size_t)(off+k)], b1 = P.bary[3*(size_t)(off+k)+1], b2 = P.bary[3*(size_t)(off+k)+2];
                        const double x0 = b0*av[0]+b1*av[2]+b2*av[4], x1 = b0*av[1]+b1*av[3]+b2*av[5];
                        double r = 0.;
                        for (int m = 0; m < nf; m++) {
                            const double g0 = P.fbary[2*(size_t)(foff+m)], g1 = P.fbary[2*(size_t)(foff+m)+1];
                            const double w0 = g0*f0x+g1*f1x-x0, w1 = g0*f0y+g1*f1y-x1;
                            r = __builtin_fma(P.fw[foff+m]*(nx*w0+ny*w1), kern_eval<KT, true>(bkn, w0*w0+w1*w1), r);
                        }
                        r *= P.w[off+k]*vs;
                        int e = 0;
#pragma unroll
                        for (int a = 0; a < DPE; a++) {
                            const double pa = P.phi[(size_t)(off+k)*DPE+a]*r;
#pragma unroll
                            for (int b = a; b < DPE; b++) { D[e] = __builtin_fma(pa, P.phi[(size_t)(off+k)*DPE+b], D[e]); e++; }
                        }
                    }
                    continue;
                }
            }
            npairs++;
            nevals += (unsigned long long)n*nf;
            const double vs = vol2*kern_scale<KT>(bkn);
            // s = 1/2 in 2D: the Gauss-theorem kernel with the 1 / |y - x| of the normal factor folded in is d2^(-3/2), exponent known at
            // compile time (kern_eval<2>: no branch per evaluation); wave-uniform test
            const bool e32 = KT == 1 && bkn.qm == 6;
            if (q == 2 && q2reg) {
                if (e32) {
#pragma unroll
                    for (int k = 0; k < 3; k++) R2[k] = __builtin_fma(vs, bt_point<DPE, 2>(s_rule+(ro+k)*ST, s_fr, fo, nf, av, f0x, f0y, f1x, f1y, nx, ny, bkn), R2[k]);
                } else {
#pragma unroll
                    for (int k = 0; k < 3; k++) R2[k] = __builtin_fma(vs, bt_point<DPE, KT>(s_rule+(ro+k)*ST, s_fr, fo, nf, av, f0x, f0y, f1x, f1y, nx, ny, bkn), R2[k]);
                }
            } else if (q == 3 && q3reg) {
                if (e32) {
#pragma unroll
                    for (int k = 0; k < 6; k++) R3[k] = __builtin_fma(vs, bt_point<DPE, 2>(s_rule+(ro+k)*ST, s_fr, fo, nf, av, f0x, f0y, f1x, f1y, nx, ny, bkn), R3[k]);
                } else {
#pragma unroll
                    for (int k = 0; k < 6; k++) R3[k] = __builtin_fma(vs, bt_point<DPE, KT>(s_rule+(ro+k)*ST, s_fr, fo, nf, av, f0x, f0y, f1x, f1y, nx, ny, bkn), R3[k]);
                }
            } else {
                for (int k = 0; k < n; k++) {
                    const double *rk = s_rule+(ro+k)*ST;
                    const double r = (e32 ? bt_point<DPE, 2>(rk, s_fr, fo, nf, av, f0x, f0y, f1x, f1y, nx, ny, bkn)
                                          : bt_point<DPE, KT>(rk, s_fr, fo, nf, av, f0x, f0y, f1x, f1y, nx, ny, bkn))*rk[3]*vs;
                    int e = 0;
#pragma unroll
                    for (int a = 0; a < DPE; a++) {
                        const double pa = rk[4+a]*r;
#pragma unroll
                        for (int b = a; b < DPE; b++) { D[e] = __builtin_fma(pa, rk[4+b], D[e]); e++; }
                    }
                }
            }
        }
    }
    if (active) {
        // the row sums of orders 2 and 3 become local entries once
        if (q2reg) {
            const int ro = P.off[2]-roff0;
#pragma unroll
            for (int k = 0; k < 3; k++) {
                const double *rk = s_rule+(ro+k)*ST;
                const double r = R2[k]*rk[3];
                int e = 0;
#pragma unroll
                for (int a = 0; a < DPE; a++) {
                    const double pa = rk[4+a]*r;
#pragma unroll
                    for (int b = a; b < DPE; b++) { D[e] = __builtin_fma(pa, rk[4+b], D[e]); e++; }
                }
            }
        }
        if (q3reg) {
            const int ro = P.off[3]-roff0;
#pragma unroll
            for (int k = 0; k < 6; k++) {
                const double *rk = s_rule+(ro+k)*ST;
                const double r = R3[k]*rk[3];
                int e = 0;
#pragma unroll
                for (int a = 0; a < DPE; a++) {
                    const double pa = rk[4+a]*r;
#pragma unroll
                    for (int b = a; b < DPE; b++) { D[e] = __builtin_fma(pa, rk[4+b], D[e]); e++; }
                }
            }
        }
#pragma unroll
        for (int e = 0; e < ND; e++)
            if (D[e] != 0.) atomic_add_f64(&Dglob[(size_t)c*ND+e], vol1*D[e]);
        if (overflow) atomicAdd(&P.counters[5], (unsigned long long)overflow);
    }
    {
        const double sp = wave_sum((double)npairs), se = wave_sum((double)nevals);
        if ((threadIdx.x & 63) == 0 && sp > 0.) {
            atomicAdd(&P.counters[3], (unsigned long long)sp);
            atomicAdd(&P.counters[4], (unsigned long long)se);
        }
    }
}
