// Non-symmetric kernels with a fractional order s(x) evaluated per quadrature point (gfx950 only).
//
// Reference path: fractionalLaplacian{1,2}D_nonsym (FL2:894-1184, FL1:410-604) with piecewise == False kernels
// (kernels.py:147-149): gamma(x, y) = C(s(x)) |x-y|^(-d-2 s(x)) (updateAndEvalFractional, KC:596-622), local matrix
//   int_K1 int_K2 [u(x) gamma(x,y) - u(y) gamma(y,x)] [v(x) - v(y)]            (NO:849-930, contrib[(2 dpe)^2])
// assembled for both orientations of every pair with addToMatrixElemElem (NA:1411-1428, 222-253).  For a kernel that is
// evaluated per point the two orientations of a DISTANT pair are the same sums (the tensor rule is symmetric under the
// swap), so they are computed once and added twice; touching pairs run both orientations because the singular rules are
// not symmetric under the swap (the two differ at quadrature-error level, 1e-7).
// The pair's order max(s over both centres and all vertices) (evalParamsOnSimplices, KC:1825-1846) enters the order
// formula and selects the near rule; the host precomputes it per cell / facet and per touching pair.
#pragma once
#include "pnl_kernels.h"

// struct PwDev: pnl_device.h

// Near field of these kernels (assembleClusters, NA:1812-1832 with symmetricCells == symmetricLocalMatrix == False): the work
// items are ORDERED cell pairs -- buildMasksForClusters walks cellsUnion x cellsUnion (NA:322-349) --, item t carries a mask over the
// (2 dpe)^2 entries of its local matrix (bit p (2 dpe) + q, getElemElemMask NA:425-440) and is scattered with fac = 1 into an
// unsymmetric CSR pattern by addToMatrixElemElemMasked (NA:520-532); entries absent from the pattern are dropped (addToEntry).
struct PwNear {
    const int *indptr, *indices;
    double *data;
    const unsigned long long *masks;     // [items][4]
};

__device__ __forceinline__ void pw_near_add(const PwNear &S, const unsigned long long *mask, int bit, int I, int J, double v) {
    if (I < 0 || J < 0 || !((mask[bit >> 6] >> (bit & 63)) & 1ull)) return;
    int lo = S.indptr[I];
    const int end = S.indptr[I+1];
    int hi = end;
    while (lo < hi) {
        const int mid = (lo+hi) >> 1;
        if (S.indices[mid] < J) lo = mid+1; else hi = mid;
    }
    if (lo < end && S.indices[lo] == J) atomic_add_f64(&S.data[lo], v);
}

template <int DIM>
__device__ __forceinline__ double pw_order(const PwDev &W, const double *x) {
    const double *p = W.p;
    if (W.type == 1) return p[0];
    double t;
    if (W.type == 4) {
        double r = 0.;
#pragma unroll
        for (int k = 0; k < DIM; k++) r += x[k]*x[k];
        t = sqrt(r);
    } else t = x[0];
    if (t < p[3]-p[2]) return p[0];
    if (t > p[3]+p[2]) return p[1];
    if (W.type == 3) return p[0]+p[4]*(t-p[3]+p[2]);
    const double u = (t-p[3])*p[4]+0.5;
    return p[0]+(p[1]-p[0])*(3.0*(u*u)-2.0*(u*u*u));
}

// order at a quadrature point that is known by its cell: lam = barycentric coordinates in the cell, sv = the values of a P1 order
// function (type 5, feFractionalOrder) at the cell's vertices in the same vertex order; every other type evaluates s(x)
template <int DIM>
__device__ __forceinline__ double pw_order_at(const PwDev &W, const double *x, const double *lam, const double *sv) {
    if (W.type == 5) {
        double s = 0.;
#pragma unroll
        for (int k = 0; k < DIM+1; k++) s = __builtin_fma(lam[k], sv[k], s);
        return s;
    }
    return pw_order<DIM>(W, x);
}

// variableFractionalLaplacianScaling (kernelNormalization.pyx:416-440), times 1/s for the boundary twin (KC:1990-1994)
template <int DIM>
__device__ __forceinline__ double pw_scaling(const PwDev &W, double s, bool boundary) {
    double C = 0.5;
    if (W.normalized && W.scal_n > 0) {
        // Clenshaw on the Chebyshev series of C(s) (host: local_matrix._setup_pointwise, checked there to 1e-14)
        const double t = (s-W.scal_mid)*W.scal_inv_half, t2 = t+t;
        double b1 = 0., b2 = 0.;
        for (int k = W.scal_n-1; k >= 1; k--) { const double b0 = __builtin_fma(t2, b1, W.scal_cheb[k]-b2); b2 = b1; b1 = b0; }
        C = __builtin_fma(t, b1, W.scal_cheb[0]-b2);
    } else if (W.normalized) {
        const double pi_pow = DIM == 1 ? 0.56418958354775628695 : 0.31830988618379067154;       // pi^(-d/2)
        C = exp2(2.0*s)*s*tgamma(s+0.5*DIM)*pi_pow/tgamma(1.0-s)*0.5;
    }
    return boundary ? C/s : C;
}

__device__ __forceinline__ DevFormula pw_formula(const PwDev &W, int dim, double sv) {
    DevFormula F;
    F.c0 = W.c0; F.clip = 0; F.pad = 0;
    if (dim == 2) { F.a = sv-1.; F.b = 1.; F.e = sv; F.den0 = 0.4; }
    else { F.a = 2.*sv-1.; F.b = 0.; F.e = 2.*sv; F.den0 = 0.8; }
    return F;
}

__device__ __forceinline__ DevFormula pw_formula_boundary(const PwDev &W, int dim, double sv) {
    DevFormula F;
    F.c0 = W.bc0; F.clip = 0; F.pad = 0;
    const double st = fmax(0.5*(-(1.-dim-2.*sv)-1.), 0.);
    if (dim == 2) { F.a = st-1.; F.b = 1.; F.e = st; F.den0 = 0.35; F.clip = 1; }
    else { F.a = 2.*st-1.; F.b = 0.; F.e = 2.*st; F.den0 = 0.8; }
    return F;
}

// ---- distant pairs: classification of all cell pairs c1 < c2 without a common vertex into the work list --------------
// workgroup per 64x64 block of the tile list (the tiles that are not uniform); entry = (c1, c2, rule offset, n | order << 16)
template <int DIM, int DPE>
__global__ void __launch_bounds__(PNL_NTHREADS)
k_pw_classify(const DevProblem P, const PwDev W, const int2 *__restrict__ tiles, int4 *__restrict__ wl,
              unsigned *__restrict__ wl_count, unsigned wl_cap, int cell_begin, int cell_end) {
    constexpr int NV = DIM+1, T = 64;
    const int ta = tiles[blockIdx.x].x, tb = tiles[blockIdx.x].y;
    const int i = threadIdx.x & 63, jw = threadIdx.x >> 6;
    const int c1 = ta*T+i;
    int vid1[NV];
    double cen1[DIM];
    bool any1 = false;
    const bool ok1 = c1 < P.nc && c1 >= cell_begin && c1 < cell_end;
    const int cc1 = c1 < P.nc ? c1 : 0;
#pragma unroll
    for (int k = 0; k < NV; k++) vid1[k] = P.cvid[(size_t)k*P.ncp+cc1];
#pragma unroll
    for (int d = 0; d < DIM; d++) cen1[d] = P.ccen[(size_t)d*P.ncp+cc1];
#pragma unroll
    for (int k = 0; k < DPE; k++) any1 = any1 || P.cdof[(size_t)k*P.ncp+cc1] >= 0;
    const double h1 = P.ch[cc1], sm1 = W.cell_smax[cc1];
    unsigned long long visited = 0, assembled = 0, evals = 0;
    for (int jj = 0; jj < 16; jj++) {
        const int c2 = tb*T+jw*16+jj;
        bool push = false;
        int q = 0, off = 0, n = 0;
        if (ok1 && c2 < P.nc && c2 > c1) {
            visited++;
            bool any = any1;
#pragma unroll
            for (int k = 0; k < DPE; k++) any = any || P.cdof[(size_t)k*P.ncp+c2] >= 0;
            bool shared = false;
#pragma unroll
            for (int a = 0; a < NV; a++)
#pragma unroll
                for (int b = 0; b < NV; b++) shared = shared || (vid1[a] == P.cvid[(size_t)b*P.ncp+c2]);
            if (any && !shared) {
                double d2 = 0.;
#pragma unroll
                for (int d = 0; d < DIM; d++) { const double u = cen1[d]-P.ccen[(size_t)d*P.ncp+c2]; d2 += u*u; }
                const DevFormula F = pw_formula(W, DIM, fmax(sm1, W.cell_smax[c2]));
                q = quad_order(F, P.H0, h1, P.ch[c2], sqrt(d2));
                if (q > P.qmax || q > PNL_MAXQ) atomicAdd(&P.counters[5], 1ull);
                else { off = P.off[q]; n = P.off[q+1]-off; push = true; }
            }
        }
        // wave-aggregated append
        const unsigned long long m = __ballot(push);
        if (m) {
            const int lane = threadIdx.x & 63;
            unsigned base = 0;
            if (lane == __builtin_ctzll(m)) base = atomicAdd(wl_count, (unsigned)__popcll(m));
            base = __shfl(base, __builtin_ctzll(m));
            if (push) {
                const unsigned pos = base+__popcll(m & ((1ull << lane)-1ull));
                if (pos < wl_cap) wl[pos] = make_int4(c1, c2, off, n | (q << 16));
            }
        }
    }
    (void)visited; (void)assembled; (void)evals; // the host knows the number of visited pairs, k_pw_stats counts the rest
}

// statistics from the histogram of the sorted list: pairs per order, both orientations' kernel evaluations
__global__ void k_pw_stats(const DevProblem P, const unsigned *__restrict__ hist) {
    const int q = threadIdx.x;
    if (q < 2 || q > P.qmax || q > PNL_MAXQ) return;
    const unsigned long long c = hist[q];
    if (!c) return;
    const unsigned long long n = (unsigned long long)(P.off[q+1]-P.off[q]);
    atomicAdd(&P.counters[8+q], c);
    atomicAdd(&P.counters[1], c);
    atomicAdd(&P.counters[2], 2ull*c*n*n);
}

// ---- distant pairs from the sorted list, rules with more than PNL_PW_LANE_MAXPTS points: 16 lanes per pair split the rows
// per point pair: L = ln d2 once, K1 = w_i w_j C(s(x_i)) exp(e(x_i) L), K2 = w_i w_j C(s(y_j)) exp(e(y_j) L);
// order and scaling of the points of the second cell are computed once per pair and kept in LDS.
#define PNL_PW_MAXPTS 128
// NEAR: entries (c1, c2, item, n | order << 16) of ORDERED pairs, the local matrix of this orientation alone, masked CSR scatter
template <int DIM, int DPE, bool NEAR = false>
__global__ void __launch_bounds__(PNL_NTHREADS)
k_pw_distant(const DevProblem P, const PwDev W, const int4 *__restrict__ sorted, const unsigned *__restrict__ offs,
             double *__restrict__ A, long long ldA, double *__restrict__ Dglob, int tab_max_pts, int nmin, const PwNear NR = PwNear{}) {
    constexpr int NV = DIM+1, NC = NV*DIM, ND = DPE*(DPE+1)/2, NG = DPE*DPE, NACC = 2*NG+2*ND, LPP = 16,
                  PPC = PNL_NTHREADS/LPP, NREP = (NACC+LPP-1)/LPP, ST = 4+DPE;
    extern __shared__ double s_mem[];            // rule [tab_max_pts][ST], then per pair of the chunk [PPC][tab_max_pts][2]: e(y_j), w_j C(y_j)
    double *s_rule = s_mem, *s_y = s_mem+(size_t)tab_max_pts*ST;
    __shared__ unsigned s_coff[PNL_WL_BINS+1];
    const int tid = threadIdx.x, sub = tid & (LPP-1), g = tid/LPP;
    if (tid == 0) {
        unsigned run = 0;
        for (int q = 0; q < PNL_WL_BINS; q++) {
            s_coff[q] = run;
            const int nq = (q >= 2 && q <= P.qmax && q <= PNL_MAXQ) ? P.off[q+1]-P.off[q] : 0;
            if (nq > 0 && nq >= nmin) run += (offs[q+1]-offs[q]+PPC-1u)/PPC;   // smaller rules: k_pw_lane
        }
        s_coff[PNL_WL_BINS] = run;
    }
    __syncthreads();
    const unsigned nchunks = s_coff[PNL_WL_BINS];
    int staged_q = -1;
    for (unsigned chunk = blockIdx.x; chunk < nchunks; chunk += gridDim.x) {
        int lo = 0, hi = PNL_WL_BINS-1;
        while (lo < hi) {
            const int mid = (lo+hi+1) >> 1;
            if (s_coff[mid] <= chunk) lo = mid; else hi = mid-1;
        }
        const int q = lo;
        const unsigned first = offs[q]+(unsigned)PPC*(chunk-s_coff[q]);
        const int cnt = (int)min((unsigned)PPC, offs[q+1]-first);
        const int4 e0 = sorted[first];
        const int n = e0.w & 0xffff, off = P.off[q];
        const bool in_lds = n <= tab_max_pts;    // larger rules (orders > 30): tables from L2, second-cell data recomputed
        auto rule = [&](int pt, int k) -> double {
            if (in_lds) return s_rule[pt*ST+k];
            return k < 3 ? P.bary[3*(size_t)(off+pt)+k] : (k == 3 ? P.w[off+pt] : P.phi[(size_t)(off+pt)*DPE+k-4]);
        };
        __syncthreads();                         // previous chunk's s_y reads are done
        if (q != staged_q && in_lds) {
            for (int t = tid; t < n*ST; t += PNL_NTHREADS) {
                const int pt = t/ST, k = t-pt*ST;
                s_rule[t] = k < 3 ? P.bary[3*(size_t)(off+pt)+k] : (k == 3 ? P.w[off+pt] : P.phi[(size_t)(off+pt)*DPE+k-4]);
            }
            staged_q = q;
        }
        __syncthreads();
        const bool valid = g < cnt;
        const int4 ent = valid ? sorted[first+g] : e0;
        const int c1 = ent.x, c2 = ent.y;
        double av[NC], bv[NC], sva[NV], svb[NV];
#pragma unroll
        for (int k = 0; k < NC; k++) { av[k] = P.cellv[(size_t)k*P.ncp+c1]; bv[k] = P.cellv[(size_t)k*P.ncp+c2]; }
#pragma unroll
        for (int k = 0; k < NV; k++) {
            sva[k] = W.type == 5 ? W.cell_sv[(size_t)k*W.sv_stride+c1] : 0.;
            svb[k] = W.type == 5 ? W.cell_sv[(size_t)k*W.sv_stride+c2] : 0.;
        }
        double *my_y = s_y+(size_t)g*tab_max_pts*2;
        for (int j = sub; j < n && in_lds; j += LPP) {
            double y[DIM];
#pragma unroll
            for (int d = 0; d < DIM; d++) {
                double sy = 0.;
#pragma unroll
                for (int m = 0; m < NV; m++) sy = __builtin_fma(s_rule[j*ST+m], bv[m*DIM+d], sy);
                y[d] = sy;
            }
            const double s = pw_order_at<DIM>(W, y, s_rule+j*ST, svb);
            my_y[2*j] = -0.5*DIM-s;
            my_y[2*j+1] = s_rule[j*ST+3]*pw_scaling<DIM>(W, s, false);
        }
        __syncthreads();
        double G1[DPE][DPE], G2[DPE][DPE], S1[ND], S2[ND];
#pragma unroll
        for (int a = 0; a < DPE; a++)
#pragma unroll
            for (int b = 0; b < DPE; b++) { G1[a][b] = 0.; G2[a][b] = 0.; }
#pragma unroll
        for (int e = 0; e < ND; e++) { S1[e] = 0.; S2[e] = 0.; }
#pragma unroll 1
        for (int i = sub; i < n; i += LPP) {
            double ti[ST];
#pragma unroll
            for (int m = 0; m < ST; m++) ti[m] = rule(i, m);
            double x[DIM];
#pragma unroll
            for (int d = 0; d < DIM; d++) {
                double sx = 0.;
#pragma unroll
                for (int m = 0; m < NV; m++) sx = __builtin_fma(ti[m], av[m*DIM+d], sx);
                x[d] = sx;
            }
            const double sx_ = pw_order_at<DIM>(W, x, ti, sva);
            const double ex = -0.5*DIM-sx_, cx = ti[3]*pw_scaling<DIM>(W, sx_, false);
            double r1 = 0., u1[DPE], u2[DPE];
#pragma unroll
            for (int b = 0; b < DPE; b++) { u1[b] = 0.; u2[b] = 0.; }
#pragma unroll 1
            for (int j = 0; j < n; j++) {
                double tj[ST];
#pragma unroll
                for (int m = 0; m < ST; m++) tj[m] = rule(j, m);
                double d2 = 0., y[DIM];
#pragma unroll
                for (int d = 0; d < DIM; d++) {
                    double sy = 0.;
#pragma unroll
                    for (int m = 0; m < NV; m++) sy = __builtin_fma(tj[m], bv[m*DIM+d], sy);
                    y[d] = sy;
                    const double t = x[d]-sy;
                    d2 = __builtin_fma(t, t, d2);
                }
                double ey, cy;
                if (in_lds) { ey = my_y[2*j]; cy = my_y[2*j+1]; }
                else { const double s = pw_order_at<DIM>(W, y, tj, svb); ey = -0.5*DIM-s; cy = tj[3]*pw_scaling<DIM>(W, s, false); }
                const double L = pnl_log(d2);
                const double K1 = (cx*tj[3])*pnl_exp(ex*L), K2 = (ti[3]*cy)*pnl_exp(ey*L);
                r1 += K1;
                double t2[DPE];
#pragma unroll
                for (int b = 0; b < DPE; b++) { u1[b] = __builtin_fma(K1, tj[4+b], u1[b]); t2[b] = K2*tj[4+b]; u2[b] += t2[b]; }
                int e = 0;
#pragma unroll
                for (int a = 0; a < DPE; a++)
#pragma unroll
                    for (int b = a; b < DPE; b++) { S2[e] = __builtin_fma(t2[a], tj[4+b], S2[e]); e++; }
            }
            int e = 0;
#pragma unroll
            for (int a = 0; a < DPE; a++) {
                const double pa = ti[4+a];
#pragma unroll
                for (int b = 0; b < DPE; b++) {
                    G1[a][b] = __builtin_fma(pa, u1[b], G1[a][b]);           // XY[a][b]: phi_a(x_i) sum_j K1 phi_b(y_j)
                    G2[a][b] = __builtin_fma(u2[a], ti[4+b], G2[a][b]);      // YX[a][b]: sum_j K2 phi_a(y_j) phi_b(x_i)
                }
                const double pr = pa*r1;
#pragma unroll
                for (int b = a; b < DPE; b++) { S1[e] = __builtin_fma(pr, ti[4+b], S1[e]); e++; }
            }
        }
        double acc[NACC];
#pragma unroll
        for (int a = 0; a < DPE; a++)
#pragma unroll
            for (int b = 0; b < DPE; b++) { acc[a*DPE+b] = G1[a][b]; acc[NG+a*DPE+b] = G2[a][b]; }
#pragma unroll
        for (int e = 0; e < ND; e++) { acc[2*NG+e] = S1[e]; acc[2*NG+ND+e] = S2[e]; }
        double mine[NREP];
#pragma unroll
        for (int r = 0; r < NREP; r++) mine[r] = 0.;
#pragma unroll
        for (int e = 0; e < NACC; e++) {
            const double s = row16_sum(acc[e]);
            mine[e/LPP] = (sub == (e & (LPP-1))) ? s : mine[e/LPP];
        }
        if constexpr (NEAR) {
            if (valid) {
                const double vv = P.cvol[c1]*P.cvol[c2];
                const unsigned long long *mask = NR.masks+4*(size_t)ent.z;
                constexpr int N2 = 2*DPE;
#pragma unroll
                for (int rep = 0; rep < NREP; rep++) {
                    const int e = sub+LPP*rep;
                    const double val = mine[rep];
                    if (e < NG) {                                   // XY: row a of c1, column b of c2
                        const int a = e/DPE, b = e-a*DPE;
                        pw_near_add(NR, mask, a*N2+DPE+b, P.cdof[(size_t)a*P.ncp+c1], P.cdof[(size_t)b*P.ncp+c2], -vv*val);
                    } else if (e < 2*NG) {                          // YX: row a of c2, column b of c1
                        const int a = (e-NG)/DPE, b = (e-NG)-a*DPE;
                        pw_near_add(NR, mask, (DPE+a)*N2+b, P.cdof[(size_t)a*P.ncp+c2], P.cdof[(size_t)b*P.ncp+c1], -vv*val);
                    } else if (e < NACC) {                          // XX / YY: symmetric blocks, entry (a, b) and (b, a)
                        const bool second = e >= 2*NG+ND;
                        const int k = e-2*NG-(second ? ND : 0), cc = second ? c2 : c1, sh = second ? DPE : 0;
                        int a = 0, rem = k;
                        while (rem >= DPE-a) { rem -= DPE-a; a++; }
                        const int b = a+rem;
                        const int I = P.cdof[(size_t)a*P.ncp+cc], J = P.cdof[(size_t)b*P.ncp+cc];
                        pw_near_add(NR, mask, (sh+a)*N2+sh+b, I, J, vv*val);
                        if (a != b) pw_near_add(NR, mask, (sh+b)*N2+sh+a, J, I, vv*val);
                    }
                }
            }
        } else
        if (valid) {
            const double vv = 2.*P.cvol[c1]*P.cvol[c2];          // both orientations (see the header of this file)
#pragma unroll
            for (int rep = 0; rep < NREP; rep++) {
                const int e = sub+LPP*rep;
                const double val = mine[rep];
                if (e < NG) {
                    const int a = e/DPE, b = e-a*DPE;
                    const int I = P.cdof[(size_t)a*P.ncp+c1], J = P.cdof[(size_t)b*P.ncp+c2];
                    if (I >= 0 && J >= 0) atomic_add_f64(&A[(long long)I*ldA+J], -vv*val);
                } else if (e < 2*NG) {
                    const int a = (e-NG)/DPE, b = (e-NG)-a*DPE;
                    const int I = P.cdof[(size_t)a*P.ncp+c2], J = P.cdof[(size_t)b*P.ncp+c1];
                    if (I >= 0 && J >= 0) atomic_add_f64(&A[(long long)I*ldA+J], -vv*val);
                } else if (e < 2*NG+ND) atomic_add_f64(&Dglob[(size_t)c1*ND+(e-2*NG)], vv*val);
                else if (e < NACC) atomic_add_f64(&Dglob[(size_t)c2*ND+(e-2*NG-ND)], vv*val);
            }
        }
    }
}

// ---- uniform tiles: every one of the 64 x 64 pairs is a distant pair of order 2 (host-side bound on the order formula over the
// two blocks and the range of the pair order, see pw_tile_is_uniform).  Same decomposition as k_tile_pure: lane = cell i of
// block a, the waves split the cells j of block b, both cross blocks accumulate in LDS sub-blocks (XY as A'[dofs a, dofs b], YX
// transposed into the same shape) and are flushed row-wise, so a pair costs LDS atomics instead of 30 global ones.  Order and
// scaled weight of the quadrature points are computed once per tile and cell, not per pair.
template <int DIM>
__global__ void __launch_bounds__(PNL_NTHREADS, 2)
k_pw_tile(const DevProblem P, const PwDev W, const int2 *__restrict__ tiles, int ntiles, double *__restrict__ A, long long ldA,
          double *__restrict__ Dglob, int acc_stride) {
    constexpr int TILE = 64, NV = DIM+1, DPE = NV, NC = NV*DIM, ND = DPE*(DPE+1)/2, NP = (DIM == 2) ? 3 : 2;
    constexpr int JW = TILE/(PNL_NTHREADS/64);
    extern __shared__ double smem[];
    double *s_y = smem;                                 // [TILE][NP*DIM]
    double *s_ey = s_y+TILE*NP*DIM;                     // [TILE][NP] exponent at the points of the b-cells
    double *s_cy = s_ey+TILE*NP;                        // [TILE][NP] weight * scaling
    double *s_volb = s_cy+TILE*NP;                      // [TILE]
    double *s_Da = s_volb+TILE;                         // [TILE][ND]
    double *s_Db = s_Da+TILE*ND;                        // [TILE][ND]
    int *s_slotb = (int*)(s_Db+TILE*ND);                // [TILE][DPE]
    int *s_hb = s_slotb+TILE*DPE;                       // [TILE]
    double *s_acc1 = (double*)(s_hb+TILE);              // [nA+1][acc_stride]  XY
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    double wq[NP], ph[NP][DPE], bary[NP][NV];
    {
        const int off = P.off[2];
        const double *__restrict__ gb = P.bary+3*(size_t)off, *__restrict__ gw = P.w+off, *__restrict__ gp = P.phi+(size_t)off*DPE;
#pragma unroll
        for (int i = 0; i < NP; i++) {
            wq[i] = gw[i];
#pragma unroll
            for (int a = 0; a < DPE; a++) ph[i][a] = gp[i*DPE+a];
#pragma unroll
            for (int k = 0; k < NV; k++) bary[i][k] = gb[3*i+k];
        }
    }
    unsigned long long npairs = 0;
#pragma unroll 1
    for (int tile_idx = blockIdx.x; tile_idx < ntiles; tile_idx += gridDim.x) {
        const int2 tl = tiles[tile_idx];
        const int ta = tl.x, tb = tl.y;
        const int nA = P.blk_ndof[ta], nB = P.blk_ndof[tb];
        double *s_acc2 = s_acc1+(size_t)(nA+1)*acc_stride;   // YX transposed: [slot in a][slot in b]
        __syncthreads();
        if (tid < TILE) {
            const int c = tb*TILE+tid;
            double bv[NC];
#pragma unroll
            for (int k = 0; k < NC; k++) bv[k] = P.cellv[(size_t)k*P.ncp+c];
#pragma unroll
            for (int jp = 0; jp < NP; jp++) {
                double y[DIM];
#pragma unroll
                for (int d = 0; d < DIM; d++) {
                    double sy = 0.;
#pragma unroll
                    for (int k = 0; k < NV; k++) sy = __builtin_fma(bary[jp][k], bv[k*DIM+d], sy);
                    y[d] = sy;
                    s_y[tid*NP*DIM+jp*DIM+d] = sy;
                }
                const double sv = pw_order<DIM>(W, y);
                s_ey[tid*NP+jp] = -0.5*DIM-sv;
                s_cy[tid*NP+jp] = wq[jp]*pw_scaling<DIM>(W, sv, false);
            }
            s_volb[tid] = P.cvol[c];
            int any = 0;
#pragma unroll
            for (int k = 0; k < DPE; k++) {
                const int sl = P.cslot[(size_t)k*P.ncp+c];
                s_slotb[tid*DPE+k] = sl >= 0 ? sl : nB;
                any |= (sl >= 0);
            }
            s_hb[tid] = any;
        }
        for (int t = tid; t < 2*(nA+1)*acc_stride; t += PNL_NTHREADS) s_acc1[t] = 0.;
        for (int t = tid; t < 2*TILE*ND; t += PNL_NTHREADS) s_Da[t] = 0.;
        const int ca = ta*TILE+lane;
        double x[NP][DIM], ex[NP], cx[NP];
        {
            double av[NC];
#pragma unroll
            for (int k = 0; k < NC; k++) av[k] = P.cellv[(size_t)k*P.ncp+ca];
#pragma unroll
            for (int ip = 0; ip < NP; ip++) {
#pragma unroll
                for (int d = 0; d < DIM; d++) {
                    double sx = 0.;
#pragma unroll
                    for (int k = 0; k < NV; k++) sx = __builtin_fma(bary[ip][k], av[k*DIM+d], sx);
                    x[ip][d] = sx;
                }
                const double sv = pw_order<DIM>(W, x[ip]);
                ex[ip] = -0.5*DIM-sv;
                cx[ip] = wq[ip]*pw_scaling<DIM>(W, sv, false);
            }
        }
        int sa[DPE];
        bool ha = false;
#pragma unroll
        for (int k = 0; k < DPE; k++) {
            const int sl = P.cslot[(size_t)k*P.ncp+ca];
            sa[k] = (sl >= 0 ? sl : nA)*acc_stride;
            ha = ha || sl >= 0;
        }
        const double vola = P.cvol[ca];
        double rr[NP];
#pragma unroll
        for (int ip = 0; ip < NP; ip++) rr[ip] = 0.;
        __syncthreads();
#pragma unroll 1
        for (int jj = 0; jj < JW; jj++) {
            const int j = wave*JW+jj;
            const bool valid = ha || (s_hb[j] != 0);
            npairs += (unsigned long long)__popcll(__ballot(valid));
            const double volb = valid ? s_volb[j] : 0.;
            double c2[NP], G1[DPE][DPE], G2[DPE][DPE];
#pragma unroll
            for (int jp = 0; jp < NP; jp++) c2[jp] = 0.;
#pragma unroll
            for (int a = 0; a < DPE; a++)
#pragma unroll
                for (int b = 0; b < DPE; b++) { G1[a][b] = 0.; G2[a][b] = 0.; }
#pragma unroll
            for (int ip = 0; ip < NP; ip++) {
                double r1 = 0., u1[DPE], u2[DPE];
#pragma unroll
                for (int b = 0; b < DPE; b++) { u1[b] = 0.; u2[b] = 0.; }
#pragma unroll
                for (int jp = 0; jp < NP; jp++) {
                    double d2 = 0.;
#pragma unroll
                    for (int d = 0; d < DIM; d++) { const double t = x[ip][d]-s_y[j*NP*DIM+jp*DIM+d]; d2 = __builtin_fma(t, t, d2); }
                    const double L = pnl_log(d2);
                    const double K1 = (cx[ip]*wq[jp])*pnl_exp(ex[ip]*L), K2 = (wq[ip]*s_cy[j*NP+jp])*pnl_exp(s_ey[j*NP+jp]*L);
                    r1 += K1;
                    c2[jp] += K2;
#pragma unroll
                    for (int b = 0; b < DPE; b++) { u1[b] = __builtin_fma(K1, ph[jp][b], u1[b]); u2[b] = __builtin_fma(K2, ph[jp][b], u2[b]); }
                }
                rr[ip] = __builtin_fma(volb, r1, rr[ip]);
#pragma unroll
                for (int a = 0; a < DPE; a++)
#pragma unroll
                    for (int b = 0; b < DPE; b++) {
                        G1[a][b] = __builtin_fma(ph[ip][a], u1[b], G1[a][b]);      // XY[a][b]
                        G2[a][b] = __builtin_fma(u2[a], ph[ip][b], G2[a][b]);      // YX[a][b]: row dof_j[a], column dof_i[b]
                    }
            }
            const double vv = 2.*vola*volb;                                       // both orientations
#pragma unroll
            for (int b = 0; b < DPE; b++) {
                const int sb = s_slotb[j*DPE+b];
#pragma unroll
                for (int a = 0; a < DPE; a++) {
                    lds_add_f64(&s_acc1[sa[a]+sb], -vv*G1[a][b]);
                    lds_add_f64(&s_acc2[sa[a]+sb], -vv*G2[b][a]);                 // transposed: [slot of i's DoF][slot of j's DoF]
                }
            }
            const double wa = valid ? vola : 0.;
            double cw[NP];
#pragma unroll
            for (int jp = 0; jp < NP; jp++) cw[jp] = wave_sum(wa*c2[jp]);
            if (lane < ND) {
                int a = 0, idx = lane;
                while (idx >= DPE-a) { idx -= DPE-a; a++; }
                const int b = a+idx;
                double s2 = 0.;
#pragma unroll
                for (int jp = 0; jp < NP; jp++) {
                    double pa = 0., pb = 0.;
#pragma unroll
                    for (int k = 0; k < DPE; k++) { pa = (a == k) ? ph[jp][k] : pa; pb = (b == k) ? ph[jp][k] : pb; }
                    s2 = __builtin_fma(pa*pb, cw[jp], s2);
                }
                s_Db[j*ND+lane] = 2.*s_volb[j]*s2;
            }
        }
        {
            int e = 0;
#pragma unroll
            for (int a = 0; a < DPE; a++)
#pragma unroll
                for (int b = a; b < DPE; b++) {
                    double s1 = 0.;
#pragma unroll
                    for (int ip = 0; ip < NP; ip++) s1 = __builtin_fma(ph[ip][a]*ph[ip][b], rr[ip], s1);
                    lds_add_f64(&s_Da[lane*ND+e], 2.*vola*s1);
                    e++;
                }
        }
        __syncthreads();
        const int *__restrict__ dofA = P.blk_dofs+(size_t)ta*P.blk_stride;
        const int *__restrict__ dofB = P.blk_dofs+(size_t)tb*P.blk_stride;
        for (int t = tid; t < nA*nB; t += PNL_NTHREADS) {
            const int r = t/nB, cc = t-r*nB;
            const double v = s_acc1[r*acc_stride+cc];
            if (v != 0.) atomic_add_f64(&A[(long long)dofA[r]*ldA+dofB[cc]], v);
        }
        for (int t = tid; t < nA*nB; t += PNL_NTHREADS) {
            const int cc = t/nA, r = t-cc*nA;
            const double v = s_acc2[r*acc_stride+cc];
            if (v != 0.) atomic_add_f64(&A[(long long)dofB[cc]*ldA+dofA[r]], v);
        }
        for (int t = tid; t < 2*TILE*ND; t += PNL_NTHREADS) {
            const double v = s_Da[t];
            if (v != 0.) {
                const int side = t/(TILE*ND), rem = t-side*TILE*ND;
                const int cc = (side ? tb : ta)*TILE+rem/ND;
                atomic_add_f64(&Dglob[(size_t)cc*ND+rem%ND], v);
            }
        }
    }
    if (lane == 0 && npairs) {
        atomicAdd(&P.counters[8+2], npairs);
        atomicAdd(&P.counters[1], npairs);
        atomicAdd(&P.counters[2], npairs*(unsigned long long)(2*NP*NP));
        atomicAdd(&P.counters[6], npairs);
    }
}

// ---- the other tiles: classification, counting sort by order and evaluation inside the tile ------------------------------
// workgroup per 64 x 64 tile: every thread classifies 16 pairs (shared vertices -> skipped here, touching pairs come from the host
// list; order from the pair-order formula); pairs whose rule has at most PNL_PW_LANE_MAXPTS points are bucketed by order in LDS
// and integrated ONE PAIR PER LANE, a wave taking 64 pairs of one order at a time, with both cross blocks and the diagonal
// blocks accumulated in LDS (ds_add_f64) and flushed once per tile; the few pairs of higher order go to the global work list
// (k_pw_distant).  Exponent and scaled weight at the rule's points of the 64 cells of block b are tabulated in LDS per order.
#define PNL_PW_LANE_MAXPTS 16
#define PNL_PW_NBUCK 16
template <int DIM>
__global__ void __launch_bounds__(PNL_NTHREADS, 1)
k_pw_mixed(const DevProblem P, const PwDev W, const int2 *__restrict__ tiles, int ntiles, double *__restrict__ A, long long ldA,
           double *__restrict__ Dglob, int acc_stride, int4 *__restrict__ wl, unsigned *__restrict__ wl_count, unsigned wl_cap,
           int cell_begin, int cell_end, unsigned *__restrict__ tile_ctr) {
    constexpr int T = 64, NV = DIM+1, DPE = NV, NC = NV*DIM, ND = DPE*(DPE+1)/2, ST = 4+DPE, MAXN = PNL_PW_LANE_MAXPTS;
    extern __shared__ double smem[];
    double *s_rule = smem;                               // [MAXN][ST]
    double *s_tab = s_rule+MAXN*ST;                      // [T][MAXN][2]: exponent, weight * scaling at the points of the b-cells
    double *s_Da = s_tab+T*MAXN*2;                       // [T][ND]
    double *s_Db = s_Da+T*ND;                            // [T][ND]
    unsigned short *s_list = (unsigned short*)(s_Db+T*ND);   // [T*T] pair codes i | j << 8, grouped by bucket
    int *s_cnt = (int*)(s_list+T*T);                     // [PNL_PW_NBUCK] counts, then [PNL_PW_NBUCK] offsets, [PNL_PW_NBUCK] cursors
    int *s_slota = s_cnt+3*PNL_PW_NBUCK;                 // [T][DPE]
    int *s_slotb = s_slota+T*DPE;                        // [T][DPE]
    double *s_acc1 = (double*)(s_slotb+T*DPE);           // [nA+1][acc_stride] XY, then the same for YX transposed
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    unsigned long long nass = 0, nev = 0;
    // first tile by block index, the following ones from a global counter (tile costs differ widely, see k_tile_distant)
    __shared__ int s_tile_next;
#pragma unroll 1
    for (int tile_idx = blockIdx.x; tile_idx < ntiles; tile_idx = s_tile_next) {
        const int ta = tiles[tile_idx].x, tb = tiles[tile_idx].y;
        const int nA = P.blk_ndof[ta], nB = P.blk_ndof[tb];
        double *s_acc2 = s_acc1+(size_t)(nA+1)*acc_stride;
        __syncthreads();
        if (tid == 0) s_tile_next = (int)(gridDim.x+atomicAdd(tile_ctr, 1u));      // published by the barriers below
        if (tid < 3*PNL_PW_NBUCK) s_cnt[tid] = 0;
        for (int t = tid; t < 2*(nA+1)*acc_stride; t += PNL_NTHREADS) s_acc1[t] = 0.;
        for (int t = tid; t < 2*T*ND; t += PNL_NTHREADS) s_Da[t] = 0.;
        for (int t = tid; t < T*DPE; t += PNL_NTHREADS) {
            const int c = t/DPE, k = t-c*DPE;
            const int sla = (ta*T+c < P.nc) ? P.cslot[(size_t)k*P.ncp+ta*T+c] : -1, slb = (tb*T+c < P.nc) ? P.cslot[(size_t)k*P.ncp+tb*T+c] : -1;
            s_slota[t] = sla >= 0 ? sla : nA;
            s_slotb[t] = slb >= 0 ? slb : nB;
        }
        __syncthreads();
        // ---- classification: thread -> cell i = tid & 63 of block a, 16 cells j of block b ----
        const int ci = tid & 63, c1 = ta*T+ci;
        const bool ok1 = c1 < P.nc && c1 >= cell_begin && c1 < cell_end;
        const int cc1 = c1 < P.nc ? c1 : 0;
        int vid1[NV];
        double cen1[DIM];
        bool any1 = false;
#pragma unroll
        for (int k = 0; k < NV; k++) vid1[k] = P.cvid[(size_t)k*P.ncp+cc1];
#pragma unroll
        for (int d = 0; d < DIM; d++) cen1[d] = P.ccen[(size_t)d*P.ncp+cc1];
#pragma unroll
        for (int k = 0; k < DPE; k++) any1 = any1 || P.cdof[(size_t)k*P.ncp+cc1] >= 0;
        const double h1 = P.ch[cc1], sm1 = W.cell_smax[cc1];
        int myq[16];
#pragma unroll 1
        for (int jj = 0; jj < 16; jj++) {
            const int cj = (tid >> 6)*16+jj, c2 = tb*T+cj;
            int q = 0;                                   // 0: nothing to do here
            bool push = false;
            int off = 0, n = 0;
            if (ok1 && c2 < P.nc && c2 > c1) {
                bool any = any1;
#pragma unroll
                for (int k = 0; k < DPE; k++) any = any || P.cdof[(size_t)k*P.ncp+c2] >= 0;
                bool shared = false;
#pragma unroll
                for (int a = 0; a < NV; a++)
#pragma unroll
                    for (int b = 0; b < NV; b++) shared = shared || (vid1[a] == P.cvid[(size_t)b*P.ncp+c2]);
                if (any && !shared) {
                    double d2 = 0.;
#pragma unroll
                    for (int d = 0; d < DIM; d++) { const double u = cen1[d]-P.ccen[(size_t)d*P.ncp+c2]; d2 += u*u; }
                    const DevFormula F = pw_formula(W, DIM, fmax(sm1, W.cell_smax[c2]));
                    const int qq = quad_order(F, P.H0, h1, P.ch[c2], sqrt(d2));
                    if (qq > P.qmax || qq > PNL_MAXQ) atomicAdd(&P.counters[5], 1ull);
                    else {
                        off = P.off[qq]; n = P.off[qq+1]-off;
                        if (n <= MAXN && qq < PNL_PW_NBUCK) { q = qq; atomicAdd(&s_cnt[qq], 1); }
                        else { push = true; q = -qq; }
                    }
                }
            }
            myq[jj] = q > 0 ? q : 0;
            const unsigned long long m = __ballot(push);
            if (m) {
                unsigned base = 0;
                const int leader = __builtin_ctzll(m);
                if (lane == leader) base = atomicAdd(wl_count, (unsigned)__popcll(m));
                base = __shfl(base, leader);
                if (push) {
                    const unsigned pos = base+__popcll(m & ((1ull << lane)-1ull));
                    if (pos < wl_cap) wl[pos] = make_int4(c1, c2, off, n | ((-q) << 16));
                }
            }
        }
        __syncthreads();
        if (tid == 0) {
            int run = 0;
            for (int q = 0; q < PNL_PW_NBUCK; q++) { s_cnt[PNL_PW_NBUCK+q] = run; run += s_cnt[q]; }
        }
        __syncthreads();
#pragma unroll 1
        for (int jj = 0; jj < 16; jj++)
            if (myq[jj] > 0) {
                const int pos = s_cnt[PNL_PW_NBUCK+myq[jj]]+atomicAdd(&s_cnt[2*PNL_PW_NBUCK+myq[jj]], 1);
                s_list[pos] = (unsigned short)(ci | (((tid >> 6)*16+jj) << 8));
            }
        __syncthreads();
        // ---- evaluation, order by order ----
#pragma unroll 1
        for (int q = 2; q < PNL_PW_NBUCK; q++) {
            const int cntq = s_cnt[q];
            if (cntq == 0) continue;                     // uniform over the workgroup
            const int off = P.off[q], n = P.off[q+1]-off, first = s_cnt[PNL_PW_NBUCK+q];
            __syncthreads();
            for (int t = tid; t < n*ST; t += PNL_NTHREADS) {
                const int pt = t/ST, k = t-pt*ST;
                s_rule[t] = k < 3 ? P.bary[3*(size_t)(off+pt)+k] : (k == 3 ? P.w[off+pt] : P.phi[(size_t)(off+pt)*DPE+k-4]);
            }
            __syncthreads();
            for (int t = tid; t < T*n; t += PNL_NTHREADS) {
                const int cj = t/n, j = t-cj*n, c2 = tb*T+cj;
                if (c2 < P.nc) {
                    double y[DIM];
#pragma unroll
                    for (int d = 0; d < DIM; d++) {
                        double sy = 0.;
#pragma unroll
                        for (int m = 0; m < NV; m++) sy = __builtin_fma(s_rule[j*ST+m], P.cellv[(size_t)(m*DIM+d)*P.ncp+c2], sy);
                        y[d] = sy;
                    }
                    const double sv = pw_order<DIM>(W, y);
                    s_tab[(cj*MAXN+j)*2] = -0.5*DIM-sv;
                    s_tab[(cj*MAXN+j)*2+1] = s_rule[j*ST+3]*pw_scaling<DIM>(W, sv, false);
                }
            }
            __syncthreads();
#pragma unroll 1
            for (int base = wave*64; base < cntq; base += PNL_NTHREADS) {
                const bool valid = base+lane < cntq;
                const unsigned code = s_list[first+(valid ? base+lane : 0)];
                const int li = code & 255, lj = code >> 8, c1p = ta*T+li, c2p = tb*T+lj;
                double av[NC], bv[NC];
#pragma unroll
                for (int k = 0; k < NC; k++) { av[k] = P.cellv[(size_t)k*P.ncp+c1p]; bv[k] = P.cellv[(size_t)k*P.ncp+c2p]; }
                const double *tabj = s_tab+(size_t)lj*MAXN*2;
                double G1[DPE][DPE], G2[DPE][DPE], S1[ND], S2[ND];
#pragma unroll
                for (int a = 0; a < DPE; a++)
#pragma unroll
                    for (int b = 0; b < DPE; b++) { G1[a][b] = 0.; G2[a][b] = 0.; }
#pragma unroll
                for (int e = 0; e < ND; e++) { S1[e] = 0.; S2[e] = 0.; }
#pragma unroll 1
                for (int i = 0; i < n; i++) {
                    double ti[ST];
#pragma unroll
                    for (int m = 0; m < ST; m++) ti[m] = s_rule[i*ST+m];
                    double x[DIM];
#pragma unroll
                    for (int d = 0; d < DIM; d++) {
                        double sx = 0.;
#pragma unroll
                        for (int m = 0; m < NV; m++) sx = __builtin_fma(ti[m], av[m*DIM+d], sx);
                        x[d] = sx;
                    }
                    const double sx_ = pw_order<DIM>(W, x);
                    const double ex = -0.5*DIM-sx_, cx = ti[3]*pw_scaling<DIM>(W, sx_, false);
                    double r1 = 0., u1[DPE], u2[DPE];
#pragma unroll
                    for (int b = 0; b < DPE; b++) { u1[b] = 0.; u2[b] = 0.; }
#pragma unroll 3
                    for (int j = 0; j < n; j++) {
                        double tj[ST];
#pragma unroll
                        for (int m = 0; m < ST; m++) tj[m] = s_rule[j*ST+m];
                        double d2 = 0.;
#pragma unroll
                        for (int d = 0; d < DIM; d++) {
                            double sy = 0.;
#pragma unroll
                            for (int m = 0; m < NV; m++) sy = __builtin_fma(tj[m], bv[m*DIM+d], sy);
                            const double t = x[d]-sy;
                            d2 = __builtin_fma(t, t, d2);
                        }
                        const double L = pnl_log(d2);
                        const double K1 = (cx*tj[3])*pnl_exp(ex*L), K2 = (ti[3]*tabj[2*j+1])*pnl_exp(tabj[2*j]*L);
                        r1 += K1;
                        double t2[DPE];
#pragma unroll
                        for (int b = 0; b < DPE; b++) { u1[b] = __builtin_fma(K1, tj[4+b], u1[b]); t2[b] = K2*tj[4+b]; u2[b] += t2[b]; }
                        int e = 0;
#pragma unroll
                        for (int a = 0; a < DPE; a++)
#pragma unroll
                            for (int b = a; b < DPE; b++) { S2[e] = __builtin_fma(t2[a], tj[4+b], S2[e]); e++; }
                    }
                    int e = 0;
#pragma unroll
                    for (int a = 0; a < DPE; a++) {
                        const double pa = ti[4+a];
#pragma unroll
                        for (int b = 0; b < DPE; b++) {
                            G1[a][b] = __builtin_fma(pa, u1[b], G1[a][b]);
                            G2[a][b] = __builtin_fma(u2[a], ti[4+b], G2[a][b]);
                        }
                        const double pr = pa*r1;
#pragma unroll
                        for (int b = a; b < DPE; b++) { S1[e] = __builtin_fma(pr, ti[4+b], S1[e]); e++; }
                    }
                }
                if (valid) {
                    const double vv = 2.*P.cvol[c1p]*P.cvol[c2p];        // both orientations
#pragma unroll
                    for (int a = 0; a < DPE; a++) {
                        const int ra = s_slota[li*DPE+a]*acc_stride;
#pragma unroll
                        for (int b = 0; b < DPE; b++) {
                            const int cb = s_slotb[lj*DPE+b];
                            lds_add_f64(&s_acc1[ra+cb], -vv*G1[a][b]);        // XY[a][b]: row dof_i[a], column dof_j[b]
                            lds_add_f64(&s_acc2[ra+cb], -vv*G2[b][a]);        // YX[b][a]: row dof_j[b], column dof_i[a], stored transposed
                        }
                    }
#pragma unroll
                    for (int e = 0; e < ND; e++) { lds_add_f64(&s_Da[li*ND+e], vv*S1[e]); lds_add_f64(&s_Db[lj*ND+e], vv*S2[e]); }
                }
            }
            if (wave == 0 && lane == 0) { nass += (unsigned long long)cntq; nev += 2ull*cntq*n*n; atomicAdd(&P.counters[8+q], (unsigned long long)cntq); }
        }
        __syncthreads();
        const int *__restrict__ dofA = P.blk_dofs+(size_t)ta*P.blk_stride;
        const int *__restrict__ dofB = P.blk_dofs+(size_t)tb*P.blk_stride;
        for (int t = tid; t < nA*nB; t += PNL_NTHREADS) {
            const int r = t/nB, cc = t-r*nB;
            const double v = s_acc1[r*acc_stride+cc];
            if (v != 0.) atomic_add_f64(&A[(long long)dofA[r]*ldA+dofB[cc]], v);
        }
        for (int t = tid; t < nA*nB; t += PNL_NTHREADS) {
            const int cc = t/nA, r = t-cc*nA;
            const double v = s_acc2[r*acc_stride+cc];
            if (v != 0.) atomic_add_f64(&A[(long long)dofB[cc]*ldA+dofA[r]], v);
        }
        for (int t = tid; t < 2*T*ND; t += PNL_NTHREADS) {
            const double v = s_Da[t];
            if (v != 0.) {
                const int side = t/(T*ND), rem = t-side*T*ND;
                const int cc = (side ? tb : ta)*T+rem/ND;
                atomic_add_f64(&Dglob[(size_t)cc*ND+rem%ND], v);
            }
        }
    }
    if (tid == 0 && nass) {
        atomicAdd(&P.counters[1], nass);
        atomicAdd(&P.counters[2], nev);
    }
}

// ---- distant pairs with at most PNL_PW_LANE_MAXPTS points per cell (orders 2-8 on triangles: almost all pairs): ONE PAIR PER
// LANE.  The list is sorted by order, so the 64 pairs of a wave's chunk run the same trip counts; the rule sits in the wave's
// LDS copy (broadcast reads), exponent and scaled weight of the second cell's points in a per-lane LDS column (conflict
// free), the 2 NG + 2 ND accumulators in registers, no cross-lane work.
template <int DIM>
__global__ void __launch_bounds__(PNL_NTHREADS)
k_pw_lane(const DevProblem P, const PwDev W, const int4 *__restrict__ sorted, const unsigned *__restrict__ offs,
          double *__restrict__ A, long long ldA, double *__restrict__ Dglob) {
    constexpr int NV = DIM+1, DPE = NV, NC = NV*DIM, ND = DPE*(DPE+1)/2, ST = 4+DPE, NW = PNL_NTHREADS/64, MAXN = PNL_PW_LANE_MAXPTS;
    __shared__ double s_rule[NW][MAXN*ST];
    __shared__ double s_y[NW][MAXN*2*64];        // [j][e | wC][lane]
    __shared__ unsigned s_coff[PNL_WL_BINS+1];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    if (tid == 0) {
        unsigned run = 0;
        for (int q = 0; q < PNL_WL_BINS; q++) {
            s_coff[q] = run;
            const int nq = (q >= 2 && q <= P.qmax && q <= PNL_MAXQ) ? P.off[q+1]-P.off[q] : 0;
            if (nq > 0 && nq <= MAXN) run += (offs[q+1]-offs[q]+63u)/64u;
        }
        s_coff[PNL_WL_BINS] = run;
    }
    __syncthreads();
    const unsigned nchunks = s_coff[PNL_WL_BINS];
    double *rule = s_rule[wave], *ycol = s_y[wave]+lane;
    int staged_q = -1;
    for (unsigned chunk = blockIdx.x*NW+wave; chunk < nchunks; chunk += gridDim.x*NW) {
        int lo = 0, hi = PNL_WL_BINS-1;
        while (lo < hi) {
            const int mid = (lo+hi+1) >> 1;
            if (s_coff[mid] <= chunk) lo = mid; else hi = mid-1;
        }
        const int q = lo;
        const unsigned first = offs[q]+64u*(chunk-s_coff[q]);
        const int cnt = (int)min(64u, offs[q+1]-first);
        const int off = P.off[q], n = P.off[q+1]-off;
        if (q != staged_q) {
            __builtin_amdgcn_wave_barrier();
            for (int t = lane; t < n*ST; t += 64) {
                const int pt = t/ST, k = t-pt*ST;
                rule[t] = k < 3 ? P.bary[3*(size_t)(off+pt)+k] : (k == 3 ? P.w[off+pt] : P.phi[(size_t)(off+pt)*DPE+k-4]);
            }
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
            __builtin_amdgcn_wave_barrier();
            staged_q = q;
        }
        const bool valid = lane < cnt;
        const int4 ent = sorted[first+(valid ? lane : 0)];
        const int c1 = ent.x, c2 = ent.y;
        double av[NC], bv[NC];
#pragma unroll
        for (int k = 0; k < NC; k++) { av[k] = P.cellv[(size_t)k*P.ncp+c1]; bv[k] = P.cellv[(size_t)k*P.ncp+c2]; }
        // order and scaling at the points of the second cell, once per pair
        for (int j = 0; j < n; j++) {
            double y[DIM];
#pragma unroll
            for (int d = 0; d < DIM; d++) {
                double sy = 0.;
#pragma unroll
                for (int m = 0; m < NV; m++) sy = __builtin_fma(rule[j*ST+m], bv[m*DIM+d], sy);
                y[d] = sy;
            }
            const double s = pw_order<DIM>(W, y);
            ycol[(2*j)*64] = -0.5*DIM-s;
            ycol[(2*j+1)*64] = rule[j*ST+3]*pw_scaling<DIM>(W, s, false);
        }
        double G1[DPE][DPE], G2[DPE][DPE], S1[ND], S2[ND];
#pragma unroll
        for (int a = 0; a < DPE; a++)
#pragma unroll
            for (int b = 0; b < DPE; b++) { G1[a][b] = 0.; G2[a][b] = 0.; }
#pragma unroll
        for (int e = 0; e < ND; e++) { S1[e] = 0.; S2[e] = 0.; }
#pragma unroll 1
        for (int i = 0; i < n; i++) {
            double ti[ST];
#pragma unroll
            for (int m = 0; m < ST; m++) ti[m] = rule[i*ST+m];
            double x[DIM];
#pragma unroll
            for (int d = 0; d < DIM; d++) {
                double sx = 0.;
#pragma unroll
                for (int m = 0; m < NV; m++) sx = __builtin_fma(ti[m], av[m*DIM+d], sx);
                x[d] = sx;
            }
            const double sx_ = pw_order<DIM>(W, x);
            const double ex = -0.5*DIM-sx_, cx = ti[3]*pw_scaling<DIM>(W, sx_, false);
            double r1 = 0., u1[DPE], u2[DPE];
#pragma unroll
            for (int b = 0; b < DPE; b++) { u1[b] = 0.; u2[b] = 0.; }
#pragma unroll 3
            for (int j = 0; j < n; j++) {       // three independent log / exp chains in flight (the rules have 3, 6, 12, 15 points)
                double tj[ST];
#pragma unroll
                for (int m = 0; m < ST; m++) tj[m] = rule[j*ST+m];
                double d2 = 0.;
#pragma unroll
                for (int d = 0; d < DIM; d++) {
                    double sy = 0.;
#pragma unroll
                    for (int m = 0; m < NV; m++) sy = __builtin_fma(tj[m], bv[m*DIM+d], sy);
                    const double t = x[d]-sy;
                    d2 = __builtin_fma(t, t, d2);
                }
                const double L = pnl_log(d2);
                const double K1 = (cx*tj[3])*pnl_exp(ex*L), K2 = (ti[3]*ycol[(2*j+1)*64])*pnl_exp(ycol[(2*j)*64]*L);
                r1 += K1;
                double t2[DPE];
#pragma unroll
                for (int b = 0; b < DPE; b++) { u1[b] = __builtin_fma(K1, tj[4+b], u1[b]); t2[b] = K2*tj[4+b]; u2[b] += t2[b]; }
                int e = 0;
#pragma unroll
                for (int a = 0; a < DPE; a++)
#pragma unroll
                    for (int b = a; b < DPE; b++) { S2[e] = __builtin_fma(t2[a], tj[4+b], S2[e]); e++; }
            }
            int e = 0;
#pragma unroll
            for (int a = 0; a < DPE; a++) {
                const double pa = ti[4+a];
#pragma unroll
                for (int b = 0; b < DPE; b++) {
                    G1[a][b] = __builtin_fma(pa, u1[b], G1[a][b]);
                    G2[a][b] = __builtin_fma(u2[a], ti[4+b], G2[a][b]);
                }
                const double pr = pa*r1;
#pragma unroll
                for (int b = a; b < DPE; b++) { S1[e] = __builtin_fma(pr, ti[4+b], S1[e]); e++; }
            }
        }
        if (valid) {
            const double vv = 2.*P.cvol[c1]*P.cvol[c2];          // both orientations
            int d1[DPE], d2_[DPE];
#pragma unroll
            for (int a = 0; a < DPE; a++) { d1[a] = P.cdof[(size_t)a*P.ncp+c1]; d2_[a] = P.cdof[(size_t)a*P.ncp+c2]; }
#pragma unroll
            for (int a = 0; a < DPE; a++)
#pragma unroll
                for (int b = 0; b < DPE; b++) {
                    if (d1[a] >= 0 && d2_[b] >= 0) atomic_add_f64(&A[(long long)d1[a]*ldA+d2_[b]], -vv*G1[a][b]);
                    if (d2_[a] >= 0 && d1[b] >= 0) atomic_add_f64(&A[(long long)d2_[a]*ldA+d1[b]], -vv*G2[a][b]);
                }
#pragma unroll
            for (int e = 0; e < ND; e++) {
                atomic_add_f64(&Dglob[(size_t)c1*ND+e], vv*S1[e]);
                atomic_add_f64(&Dglob[(size_t)c2*ND+e], vv*S2[e]);
            }
        }
    }
}

// ---- touching pairs (FL2:1133-1184, FL1:548-604): one wave per (pair, orientation), rule of the pair's order key ------
// pairs[t] = (c1 <= c2, common, key); full (rows x rows) non-symmetric local matrix, scatter NA:222-253
// NEAR: pairs[t] = (c1, c2, common, key) of item t, ORDERED (one wave per item, this orientation alone), masked CSR scatter
template <int DIM, int DPE, int SLOT, bool NEAR = false>
__global__ void __launch_bounds__(PNL_NTHREADS)
k_pw_singular(const DevProblem P, const PwDev W, const int4 *__restrict__ pairs, int npairs, double *__restrict__ A, long long ldA,
              int cell_begin, int cell_end, const PwNear NR = PwNear{}, const int *__restrict__ item_of = nullptr) {
    constexpr int NV = DIM+1, DPV = elem_dpv(DPE), DPED = elem_dped(DIM, DPE);
    constexpr int COMMON = SLOT+1;
    // merged local DoFs: shared vertices (and the shared edge) first (FL2:965-1075, FL1:466-530)
    constexpr int ROWS = (COMMON == NV) ? DPE : (COMMON == 1 ? 2*DPE-DPV : 2*DPE-2*DPV-DPED);
    // the ROWS x ROWS local matrix in groups of RG rows (P2: up to 11 x 11 entries do not fit the registers of a lane at once; the
    // point loop runs once per group -- touching pairs are few)
    constexpr int NGRP = ROWS > 8 ? 2 : 1, RG = (ROWS+NGRP-1)/NGRP, NE = RG*ROWS;
    const int lane = threadIdx.x & 63;
    const int wid = (blockIdx.x*PNL_NTHREADS+threadIdx.x) >> 6;
    if (wid >= (NEAR ? npairs : 2*npairs)) return;
    const int4 pr = pairs[NEAR ? wid : wid >> 1];
    if (pr.z != COMMON) return;
    const int orient = NEAR ? 0 : wid & 1;
    const int p1 = __builtin_amdgcn_readfirstlane(pr.x), p2 = __builtin_amdgcn_readfirstlane(pr.y);
    if (!NEAR && (p1 < cell_begin || p1 >= cell_end)) return;
    if (orient && p1 == p2) return;
    const int c1 = orient ? p2 : p1, c2 = orient ? p1 : p2;
    const int key = __builtin_amdgcn_readfirstlane(pr.w);
    int ld[2*DPE];
    bool any = false;
#pragma unroll
    for (int k = 0; k < DPE; k++) {
        ld[k] = P.cdof[(size_t)k*P.ncp+c1];
        ld[DPE+k] = P.cdof[(size_t)k*P.ncp+c2];
        any = any || ld[k] >= 0 || ld[DPE+k] >= 0;
    }
    if (!any) return;
    int perm1[NV], perm2[NV], perm[2*DPE];
#pragma unroll
    for (int k = 0; k < NV; k++) { perm1[k] = k; perm2[k] = k; }
#pragma unroll
    for (int k = 0; k < 2*DPE; k++) perm[k] = k;
    if (c1 != c2) {
        int mask1 = 0, mask2 = 0, common = 0;
        int vid1[NV], vid2[NV];
#pragma unroll
        for (int a = 0; a < NV; a++) { vid1[a] = P.cvid[(size_t)a*P.ncp+c1]; vid2[a] = P.cvid[(size_t)a*P.ncp+c2]; }
#pragma unroll
        for (int a = 0; a < NV; a++) {
            const int v1 = vid1[a];
#pragma unroll
            for (int b = 0; b < NV; b++) {
                if (mask2 & (1 << b)) continue;
                if (v1 == vid2[b]) {
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
    double s1[NV][DIM], s2[NV][DIM];
#pragma unroll
    for (int k = 0; k < NV; k++)
#pragma unroll
        for (int d = 0; d < DIM; d++) {
            double a = 0., b = 0.;
#pragma unroll
            for (int m = 0; m < NV; m++) {
                const double va = P.cellv[(size_t)(m*DIM+d)*P.ncp+c1], vb = P.cellv[(size_t)(m*DIM+d)*P.ncp+c2];
                a = (perm1[k] == m) ? va : a;
                b = (perm2[k] == m) ? vb : b;
            }
            s1[k][d] = a; s2[k][d] = b;
        }
    // type 5: the order function at the (permuted) vertices of both cells
    double sv1[NV], sv2[NV];
#pragma unroll
    for (int k = 0; k < NV; k++) {
        double a = 0., b = 0.;
#pragma unroll
        for (int m = 0; m < NV; m++) {
            const double va = W.type == 5 ? W.cell_sv[(size_t)m*W.sv_stride+c1] : 0., vb = W.type == 5 ? W.cell_sv[(size_t)m*W.sv_stride+c2] : 0.;
            a = (perm1[k] == m) ? va : a;
            b = (perm2[k] == m) ? vb : b;
        }
        sv1[k] = a; sv2[k] = b;
    }
    const int M = W.M[SLOT];
    const double *__restrict__ nodes = W.nodes[SLOT]+(size_t)key*2*NV*M;
    const double *__restrict__ w = W.w[SLOT]+(size_t)key*M;
    const double *__restrict__ phi0 = W.phi0[SLOT]+(size_t)key*ROWS*M;
    const double *__restrict__ phi1 = W.phi1[SLOT]+(size_t)key*ROWS*M;
    const double vol = W.sfac*P.cvol[c1]*P.cvol[c2];
#pragma unroll 1
    for (int grp = 0; grp < NGRP; grp++) {
        const int r0 = grp*RG;
        double acc[NE];
#pragma unroll
        for (int e = 0; e < NE; e++) acc[e] = 0.;
        for (int m = lane; m < M; m += 64) {
            double x[DIM], y[DIM], d2 = 0., lx[NV], ly[NV];
#pragma unroll
            for (int k = 0; k < NV; k++) { lx[k] = nodes[(size_t)k*M+m]; ly[k] = nodes[(size_t)(NV+k)*M+m]; }
#pragma unroll
            for (int d = 0; d < DIM; d++) {
                double xx = 0., yy = 0.;
#pragma unroll
                for (int k = 0; k < NV; k++) {
                    xx = __builtin_fma(s1[k][d], lx[k], xx);
                    yy = __builtin_fma(s2[k][d], ly[k], yy);
                }
                x[d] = xx; y[d] = yy;
                d2 = __builtin_fma(xx-yy, xx-yy, d2);
            }
            const double L = pnl_log(d2);
            const double sx = pw_order_at<DIM>(W, x, lx, sv1), sy = pw_order_at<DIM>(W, y, ly, sv2);
            const double t1 = w[m]*pw_scaling<DIM>(W, sx, false)*pnl_exp((-0.5*DIM-sx)*L);
            const double t2 = w[m]*pw_scaling<DIM>(W, sy, false)*pnl_exp((-0.5*DIM-sy)*L);
            double ps[ROWS];
#pragma unroll
            for (int r = 0; r < ROWS; r++) ps[r] = phi0[(size_t)r*M+m]-phi1[(size_t)r*M+m];
#pragma unroll
            for (int I = 0; I < RG; I++) {
                const int r = min(r0+I, ROWS-1);
                const double f = t1*phi0[(size_t)r*M+m]-t2*phi1[(size_t)r*M+m];
#pragma unroll
                for (int J = 0; J < ROWS; J++) acc[I*ROWS+J] = __builtin_fma(f, ps[J], acc[I*ROWS+J]);
            }
        }
        // reduce; lane (e mod 64) scatters entry e of the group
        double mine[(NE+63)/64];
#pragma unroll
        for (int r = 0; r < (NE+63)/64; r++) mine[r] = 0.;
#pragma unroll
        for (int e = 0; e < NE; e++) {
            const double sum = wave_sum(acc[e]);
            if (lane == (e & 63)) mine[e >> 6] = sum;
        }
#pragma unroll
        for (int rep = 0; rep < (NE+63)/64; rep++) {
            const int e = lane+64*rep;
            if (e >= NE) continue;
            const int myI = r0+e/ROWS, myJ = e-(e/ROWS)*ROWS;
            if (myI >= ROWS) continue;
            int gi = -1, gj = -1;
#pragma unroll
            for (int k = 0; k < 2*DPE; k++) {
                const int pk = perm[k];
                int g = -1;
#pragma unroll
                for (int m = 0; m < 2*DPE; m++) g = (pk == m) ? ld[m] : g;
                gi = (myI == k) ? g : gi;
                gj = (myJ == k) ? g : gj;
            }
            if constexpr (NEAR) {
                // local (unmerged) indices of the entry: row perm[myI], column perm[myJ] of the (2 dpe)^2 local matrix
                int li = 0, lj = 0;
#pragma unroll
                for (int k = 0; k < 2*DPE; k++) { li = (myI == k) ? perm[k] : li; lj = (myJ == k) ? perm[k] : lj; }
                pw_near_add(NR, NR.masks+4*(size_t)item_of[wid], li*(2*DPE)+lj, gi, gj, mine[rep]*vol);
            } else
            if (gi >= 0 && gj >= 0) atomic_add_f64(&A[(long long)gi*ldA+gj], mine[rep]*vol);
        }
    }
    if (lane == 0) {
        if (!orient) { atomicAdd(&P.counters[1], 1ull); atomicAdd(&P.counters[128+SLOT], 1ull); }
        atomicAdd(&P.counters[2], (unsigned long long)M);
    }
}

// ---- Omega x Omega^c with the pointwise boundary kernel C(s(x))/s(x) |x-y|^(1-d-2 s(x)), x in the cell ----------------
template <int DIM, int DPE>
__global__ void __launch_bounds__(PNL_NTHREADS)
k_pw_boundary_distant(const DevProblem P, const PwDev W, double *__restrict__ Dglob, int cell_begin, int cell_end, int facets_per_chunk) {
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
    const double h1 = P.ch[cc], vol1 = P.cvol[cc], sm1 = W.cell_smax[cc];
    double sv[NV];
#pragma unroll
    for (int k = 0; k < NV; k++) sv[k] = W.type == 5 ? W.cell_sv[(size_t)k*W.sv_stride+cc] : 0.;
    double D[ND];
#pragma unroll
    for (int e = 0; e < ND; e++) D[e] = 0.;
    unsigned long long npairs = 0, nevals = 0;
    int overflow = 0;
    const int f0 = blockIdx.y*facets_per_chunk;
    const int f1 = min(P.nb, f0+facets_per_chunk);
    for (int f = f0; f < f1; f++) {
        double fv[NF*DIM], fc[DIM], nrm[DIM];
        int fvid[NF];
#pragma unroll
        for (int k = 0; k < NF; k++) fvid[k] = P.bvid[(size_t)k*P.nb+f];
#pragma unroll
        for (int k = 0; k < NF*DIM; k++) fv[k] = P.bv[(size_t)k*P.nb+f];
#pragma unroll
        for (int d = 0; d < DIM; d++) { fc[d] = P.bgeo[(size_t)d*P.nb+f]; nrm[d] = P.bgeo[(size_t)(DIM+d)*P.nb+f]; }
        const double vol2 = P.bgeo[(size_t)(2*DIM)*P.nb+f];
        bool shared = false;
#pragma unroll
        for (int k = 0; k < NV; k++)
#pragma unroll
            for (int m = 0; m < NF; m++) shared = shared || (vid[k] == fvid[m]);
        if (!active || shared) continue;
        double dc2 = 0.;
#pragma unroll
        for (int d = 0; d < DIM; d++) dc2 += (cen[d]-fc[d])*(cen[d]-fc[d]);
        const DevFormula F = pw_formula_boundary(W, DIM, fmax(sm1, W.facet_smax[f]));
        const int q = quad_order(F, P.H0, h1, vol2, sqrt(dc2));
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
            const double sx = pw_order_at<DIM>(W, x, bary+3*k, sv);
            // Gamma_b = (C/s) d2^(0.5 (1-d) - s); the 1/|y-x| of the normal factor is folded into the exponent (2D)
            const double ex = 0.5*(1-DIM)-sx-(DIM == 2 ? 0.5 : 0.), cx = pw_scaling<DIM>(W, sx, true);
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
                if (DIM != 2) nw = 1.;
                r = __builtin_fma(fw[m]*nw, pnl_exp(ex*pnl_log(d2)), r);
            }
            r *= w[k]*vol*cx;
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
        if (overflow) atomicAdd(&P.counters[5], (unsigned long long)overflow);
    }
    const double sp = wave_sum((double)npairs), se = wave_sum((double)nevals);
    if ((threadIdx.x & 63) == 0 && sp > 0.) {
        atomicAdd(&P.counters[3], (unsigned long long)sp);
        atomicAdd(&P.counters[4], (unsigned long long)se);
    }
}

// touching (cell, facet) pairs: pairs[t] = (cell, facet, common, key); one wave per pair
template <int DIM, int DPE, int SLOT>
__global__ void __launch_bounds__(PNL_NTHREADS)
k_pw_boundary_singular(const DevProblem P, const PwDev W, const int4 *__restrict__ pairs, int npairs, double *__restrict__ Dglob,
                       int cell_begin, int cell_end) {
    constexpr int NV = DIM+1, NF = DIM, ND = DPE*(DPE+1)/2;
    const int lane = threadIdx.x & 63;
    const int wid = (blockIdx.x*PNL_NTHREADS+threadIdx.x) >> 6;
    if (wid >= npairs) return;
    const int4 pr = pairs[wid];
    if (pr.z != SLOT+1) return;
    const int c1 = __builtin_amdgcn_readfirstlane(pr.x), f = __builtin_amdgcn_readfirstlane(pr.y);
    const int key = __builtin_amdgcn_readfirstlane(pr.w);
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
    double sv1[NV];
#pragma unroll
    for (int k = 0; k < NV; k++) {
        double a = 0.;
#pragma unroll
        for (int m = 0; m < NV; m++) a = (perm1[k] == m) ? (W.type == 5 ? W.cell_sv[(size_t)m*W.sv_stride+c1] : 0.) : a;
        sv1[k] = a;
    }
    double vol2 = 1.;
    if (DIM == 2) {
        nrm[0] = fv[1][1]-fv[0][1];
        nrm[1] = fv[0][0]-fv[1][0];
        const double inv = 1./sqrt(nrm[0]*nrm[0]+nrm[1]*nrm[1]);
        nrm[0] *= inv; nrm[1] *= inv;
        vol2 = sqrt((fv[1][0]-fv[0][0])*(fv[1][0]-fv[0][0])+(fv[1][1]-fv[0][1])*(fv[1][1]-fv[0][1]));
    }
    const int M = W.bM[SLOT];
    const double *__restrict__ nodes = W.bnodes[SLOT]+(size_t)key*(NV+NF)*M;
    const double *__restrict__ w = W.bw[SLOT]+(size_t)key*M;
    const double *__restrict__ PHI = W.bphi[SLOT]+(size_t)key*DPE*M;
    double acc[ND];
#pragma unroll
    for (int e = 0; e < ND; e++) acc[e] = 0.;
    for (int m = lane; m < M; m += 64) {
        double x[DIM], d2 = 0., nw = 0., lx[NV];
#pragma unroll
        for (int k = 0; k < NV; k++) lx[k] = nodes[(size_t)k*M+m];
#pragma unroll
        for (int d = 0; d < DIM; d++) {
            double xx = 0., y = 0.;
#pragma unroll
            for (int k = 0; k < NV; k++) xx = __builtin_fma(s1[k][d], lx[k], xx);
#pragma unroll
            for (int k = 0; k < NF; k++) y = __builtin_fma(s2[k][d], nodes[(size_t)(NV+k)*M+m], y);
            x[d] = xx;
            const double wv = xx-y;
            d2 = __builtin_fma(wv, wv, d2);
            if (DIM == 2) nw = __builtin_fma(nrm[d], wv, nw);
        }
        if (DIM != 2) nw = 1.;
        const double sx = pw_order_at<DIM>(W, x, lx, sv1);
        const double ex = 0.5*(1-DIM)-sx-(DIM == 2 ? 0.5 : 0.);
        const double t = w[m]*nw*pw_scaling<DIM>(W, sx, true)*pnl_exp(ex*pnl_log(d2));
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
    const double vol = (DIM == 2) ? W.bfac*P.cvol[c1]*vol2 : W.bfac*P.cvol[c1];
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

// ---- near field (assembleClusters / getH2 with these kernels) -------------------------------------------------------------
// distant items of the masked list: thread per item t = (c1, c2) ORDERED, without a common vertex (the host lists the touching
// items apart, with the key of their near rule); entry (c1, c2, t, n | order << 16) of the work list
template <int DIM, int DPE>
__global__ void __launch_bounds__(PNL_NTHREADS)
k_pw_classify_near(const DevProblem P, const PwDev W, const int *__restrict__ pairs, const int *__restrict__ item_of, int nitems,
                   int4 *__restrict__ wl, unsigned *__restrict__ wl_count, unsigned wl_cap) {
    constexpr int NV = DIM+1;
    const int t = blockIdx.x*PNL_NTHREADS+threadIdx.x;
    bool push = false;
    int c1 = 0, c2 = 0, q = 0, off = 0, n = 0, item = 0;
    if (t < nitems) {
        item = item_of[t];
        c1 = pairs[2*(size_t)item]; c2 = pairs[2*(size_t)item+1];
        bool shared = false;
#pragma unroll
        for (int a = 0; a < NV; a++)
#pragma unroll
            for (int b = 0; b < NV; b++) shared = shared || (P.cvid[(size_t)a*P.ncp+c1] == P.cvid[(size_t)b*P.ncp+c2]);
        if (shared) atomicAdd(&P.counters[5], 1ull);                // a touching pair in the distant list: the call fails (pnl_synchronize)
        else {
            double d2 = 0.;
#pragma unroll
            for (int d = 0; d < DIM; d++) { const double u = P.ccen[(size_t)d*P.ncp+c1]-P.ccen[(size_t)d*P.ncp+c2]; d2 += u*u; }
            const DevFormula F = pw_formula(W, DIM, fmax(W.cell_smax[c1], W.cell_smax[c2]));
            q = quad_order(F, P.H0, P.ch[c1], P.ch[c2], sqrt(d2));
            if (q > P.qmax || q > PNL_MAXQ) atomicAdd(&P.counters[5], 1ull);
            else { off = P.off[q]; n = P.off[q+1]-off; push = n > 0; if (!push) atomicAdd(&P.counters[5], 1ull); }
        }
    }
    (void)off;
    const unsigned long long m = __ballot(push);
    if (m) {
        const int lane = threadIdx.x & 63;
        unsigned base = 0;
        if (lane == __builtin_ctzll(m)) base = atomicAdd(wl_count, (unsigned)__popcll(m));
        base = __shfl(base, __builtin_ctzll(m));
        if (push) {
            const unsigned pos = base+__popcll(m & ((1ull << lane)-1ull));
            if (pos < wl_cap) wl[pos] = make_int4(c1, c2, item, n | (q << 16));
        }
    }
}

// statistics of the near-field work list: pairs per order, kernel evaluations of ONE orientation per item
__global__ void k_pw_stats_near(const DevProblem P, const unsigned *__restrict__ hist) {
    const int q = threadIdx.x;
    if (q < 2 || q > P.qmax || q > PNL_MAXQ) return;
    const unsigned long long c = hist[q];
    if (!c) return;
    const unsigned long long n = (unsigned long long)(P.off[q+1]-P.off[q]);
    atomicAdd(&P.counters[8+q], c);
    atomicAdd(&P.counters[1], c);
    atomicAdd(&P.counters[2], c*n*n);
}

// Gauss-theorem term over explicit (cell, facet) items with the pointwise boundary kernel C(s(x))/s(x) |x-y|^(1-d-2 s(x)), x in the
// cell: the cluster exterior of assembleClusters (local_matrix_surface, NA:1966-2028; no shift of the facet centre for orders of
// one variable) and the global Omega x Omega^c term with fac = -1 (NA:2126-2156).  One wave per item; rule[t] = -1: distant pair,
// order from the pair's largest order sv[t] (FL2:1226-1243 / FL1:644-660); otherwise the key of the near rule (FL2:1255-1314,
// FL1:672-712).  Scatter NA:534-546 (addToMatrixElemSymMasked) into CSR / SSS.
template <int DIM, int DPE>
__global__ void __launch_bounds__(PNL_NTHREADS)
k_pw_boundary_items(const DevProblem P, const PwDev W, const double *__restrict__ verts, const int *__restrict__ cells,
                    const int *__restrict__ facets, const unsigned *__restrict__ masks, const int *__restrict__ rule,
                    const double *__restrict__ svs, int ni, double fac, const SparseOut S) {
    constexpr int NV = DIM+1, NF = DIM, ND = DPE*(DPE+1)/2;
    const int lane = threadIdx.x & 63;
    const int nwaves = gridDim.x*(PNL_NTHREADS/64);
    unsigned long long npairs = 0, nevals = 0;
    for (int wid = (blockIdx.x*PNL_NTHREADS+threadIdx.x) >> 6; wid < ni; wid += nwaves) {
        const int c1 = __builtin_amdgcn_readfirstlane(cells[wid]);
        const unsigned mask = (unsigned)__builtin_amdgcn_readfirstlane((int)masks[wid]);
        const int key = __builtin_amdgcn_readfirstlane(rule[wid]);
        int fvid[NF];
        double fv[NF][DIM], cv[NV][DIM], csv[NV];
#pragma unroll
        for (int k = 0; k < NF; k++) {
            fvid[k] = __builtin_amdgcn_readfirstlane(facets[(size_t)wid*NF+k]);
#pragma unroll
            for (int d = 0; d < DIM; d++) fv[k][d] = verts[(size_t)fvid[k]*DIM+d];
        }
#pragma unroll
        for (int k = 0; k < NV; k++) {
#pragma unroll
            for (int d = 0; d < DIM; d++) cv[k][d] = P.cellv[(size_t)(k*DIM+d)*P.ncp+c1];
            csv[k] = W.type == 5 ? W.cell_sv[(size_t)k*W.sv_stride+c1] : 0.;
        }
        int perm1[NV], perm2[NF], perm[DPE];
#pragma unroll
        for (int k = 0; k < NV; k++) perm1[k] = k;
#pragma unroll
        for (int k = 0; k < NF; k++) perm2[k] = k;
#pragma unroll
        for (int k = 0; k < DPE; k++) perm[k] = k;
        int mask1 = 0, mask2 = 0, common = 0;
        for (int a = 0; a < NV; a++) {
            const int v1 = P.cvid[(size_t)a*P.ncp+c1];
            for (int b = 0; b < NF; b++) {
                if (mask2 & (1 << b)) continue;
                if (v1 == fvid[b]) {
                    perm1[common] = a; perm2[common] = b;
                    mask1 += (1 << a); mask2 += (1 << b);
                    common++;
                    break;
                }
            }
        }
        double nrm[DIM], vol2 = 1.;
#pragma unroll
        for (int d = 0; d < DIM; d++) nrm[d] = 0.;
        if (DIM == 2) {
            nrm[0] = fv[1][1]-fv[0][1];
            nrm[1] = fv[0][0]-fv[1][0];
            const double inv = 1./sqrt(nrm[0]*nrm[0]+nrm[1]*nrm[1]);
            nrm[0] *= inv; nrm[1] *= inv;
            vol2 = sqrt((fv[1][0]-fv[0][0])*(fv[1][0]-fv[0][0])+(fv[1][1]-fv[0][1])*(fv[1][1]-fv[0][1]));
        }
        double acc[ND];
#pragma unroll
        for (int e = 0; e < ND; e++) acc[e] = 0.;
        double vol;
        if (common == 0) {
            double dc2 = 0.;
#pragma unroll
            for (int d = 0; d < DIM; d++) {
                double fc = 0.;
#pragma unroll
                for (int k = 0; k < NF; k++) fc += fv[k][d];
                const double u = P.ccen[(size_t)d*P.ncp+c1]-fc*(1./NF);
                dc2 += u*u;
            }
            const DevFormula F = pw_formula_boundary(W, DIM, svs[wid]);
            const int q = quad_order(F, P.H0, P.ch[c1], vol2, sqrt(dc2));
            if (key >= 0 || q > P.qmax || q > PNL_MAXQ || P.off[q+1] == P.off[q] || P.foff[q+1] == P.foff[q]) {
                if (lane == 0) atomicAdd(&P.counters[5], 1ull);
                continue;
            }
            const int off = P.off[q], n = P.off[q+1]-off, foff = P.foff[q], nf = P.foff[q+1]-foff;
            for (int k = lane; k < n*nf; k += 64) {
                const int i = k/nf, m = k-i*nf;
                double d2 = 0., nw = 0., x[DIM], lam[NV];
#pragma unroll
                for (int t = 0; t < NV; t++) lam[t] = P.bary[3*(size_t)(off+i)+t];
#pragma unroll
                for (int d = 0; d < DIM; d++) {
                    double xx = 0., y = 0.;
#pragma unroll
                    for (int t = 0; t < NV; t++) xx = __builtin_fma(lam[t], cv[t][d], xx);
#pragma unroll
                    for (int t = 0; t < NF; t++) y = __builtin_fma(P.fbary[2*(size_t)(foff+m)+t], fv[t][d], y);
                    x[d] = xx;
                    const double wv = y-xx;
                    d2 = __builtin_fma(wv, wv, d2);
                    if (DIM == 2) nw = __builtin_fma(nrm[d], wv, nw);
                }
                if (DIM != 2) nw = 1.;
                const double sx = pw_order_at<DIM>(W, x, lam, csv);
                const double ex = 0.5*(1-DIM)-sx-(DIM == 2 ? 0.5 : 0.);
                const double t = (P.w[off+i]*P.fw[foff+m])*nw*pw_scaling<DIM>(W, sx, true)*pnl_exp(ex*pnl_log(d2));
                int e = 0;
#pragma unroll
                for (int a = 0; a < DPE; a++) {
                    const double ta = t*P.phi[(size_t)(off+i)*DPE+a];
#pragma unroll
                    for (int b = a; b < DPE; b++) { acc[e] = __builtin_fma(ta, P.phi[(size_t)(off+i)*DPE+b], acc[e]); e++; }
                }
            }
            vol = P.cvol[c1]*vol2;
            nevals += (unsigned long long)n*nf;
        } else {
            if (key < 0 || common > NF) {
                if (lane == 0) atomicAdd(&P.counters[5], 1ull);
                continue;
            }
            int i = 0;
            for (int k = common; k < NV; k++) { while (mask1 & (1 << i)) i++; perm1[k] = i; mask1 += (1 << i); }
            i = 0;
            for (int k = common; k < NF; k++) { while (mask2 & (1 << i)) i++; perm2[k] = i; mask2 += (1 << i); }
            const int *t1 = P.perm_table+perm_rank(perm1, NV)*DPE;
            for (int k = 0; k < DPE; k++) perm[k] = t1[k];
            double s1[NV][DIM], s2[NF][DIM], sv1[NV];
#pragma unroll
            for (int k = 0; k < NV; k++) {
                double sa = 0.;
#pragma unroll
                for (int d = 0; d < DIM; d++) {
                    double a = 0.;
#pragma unroll
                    for (int m = 0; m < NV; m++) a = (perm1[k] == m) ? cv[m][d] : a;
                    s1[k][d] = a;
                }
#pragma unroll
                for (int m = 0; m < NV; m++) sa = (perm1[k] == m) ? csv[m] : sa;
                sv1[k] = sa;
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
            const int slot = common-1;
            const int M = W.bM[slot];
            const double *__restrict__ nodes = W.bnodes[slot]+(size_t)key*(NV+NF)*M;
            const double *__restrict__ w = W.bw[slot]+(size_t)key*M;
            const double *__restrict__ PHI = W.bphi[slot]+(size_t)key*DPE*M;
            for (int m = lane; m < M; m += 64) {
                double x[DIM], d2 = 0., nw = 0., lx[NV];
#pragma unroll
                for (int k = 0; k < NV; k++) lx[k] = nodes[(size_t)k*M+m];
#pragma unroll
                for (int d = 0; d < DIM; d++) {
                    double xx = 0., y = 0.;
#pragma unroll
                    for (int k = 0; k < NV; k++) xx = __builtin_fma(s1[k][d], lx[k], xx);
#pragma unroll
                    for (int k = 0; k < NF; k++) y = __builtin_fma(s2[k][d], nodes[(size_t)(NV+k)*M+m], y);
                    x[d] = xx;
                    const double wv = xx-y;
                    d2 = __builtin_fma(wv, wv, d2);
                    if (DIM == 2) nw = __builtin_fma(nrm[d], wv, nw);
                }
                if (DIM != 2) nw = 1.;
                const double sx = pw_order_at<DIM>(W, x, lx, sv1);
                const double ex = 0.5*(1-DIM)-sx-(DIM == 2 ? 0.5 : 0.);
                const double t = w[m]*nw*pw_scaling<DIM>(W, sx, true)*pnl_exp(ex*pnl_log(d2));
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
            vol = (DIM == 2) ? W.bfac*P.cvol[c1]*vol2 : W.bfac*P.cvol[c1];
            nevals += (unsigned long long)M;
        }
        npairs++;
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
            int i = 0, j = 0;
#pragma unroll
            for (int k = 0; k < DPE; k++) { i = (myI == k) ? perm[k] : i; j = (myJ == k) ? perm[k] : j; }
            const int lo = min(i, j), hi = max(i, j);
            const int kk = DPE*lo-(lo*(lo+1) >> 1)+hi;
            if ((mask >> kk) & 1u) {
                const int I = P.cdof[(size_t)lo*P.ncp+c1], J = P.cdof[(size_t)hi*P.ncp+c1];
                const double v = fac*vol*mine;
                if (lo == hi) sparse_add(S, I, I, v);
                else { sparse_add(S, I, J, v); sparse_add(S, J, I, v); }
            }
        }
    }
    if (lane == 0 && npairs) {
        atomicAdd(&P.counters[3], npairs);
        atomicAdd(&P.counters[4], nevals);
    }
}

// far field of these kernels (assembleFarFieldInteractions, clusterMethodCy.pyx:2153-2238 with kernel.variable: evalParamsPtr(x, y)
// before evalPtr): K[i][j] = -2 C(s(x_i)) |x_i - y_j|^(-d-2 s(x_i)), x_i the Chebyshev nodes of the ROW cluster n1
template <int DIM>
__global__ void __launch_bounds__(PNL_NTHREADS)
k_h2_kernel_interp_pw(const H2Dev H, const PwDev W) {
    const int pr = blockIdx.x, n1 = H.far[2*pr], n2 = H.far[2*pr+1];
    const double *b1 = H.box+(size_t)n1*DIM*2, *b2 = H.box+(size_t)n2*DIM*2;
    for (int t = threadIdx.x; t < H.M*H.M; t += PNL_NTHREADS) {
        const int i = t/H.M, j = t-i*H.M;
        double d2 = 0., x[DIM];
        int ii = i, jj = j;
#pragma unroll
        for (int d = 0; d < DIM; d++) {
            x[d] = cheb_node(b1[2*d], b1[2*d+1], H.m, ii % H.m);
            const double y = cheb_node(b2[2*d], b2[2*d+1], H.m, jj % H.m);
            ii /= H.m; jj /= H.m;
            d2 += (x[d]-y)*(x[d]-y);
        }
        const double s = pw_order<DIM>(W, x);
        H.K[(size_t)pr*H.M*H.M+t] = -2.*pw_scaling<DIM>(W, s, false)*pnl_exp((-0.5*DIM-s)*pnl_log(d2));
    }
}
