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
#include "pnl_common.h"

// ---------------------------------------------------------------------------------------------
// Distant pair, one pair per lane (NO:722-789, uncut branch).  The reference forms
// contrib[IJ] = vol * sum_k temp[k] PSI[I,k] PSI[J,k] over the tensor rule; with
// PSI = (phi(x_i), -phi(y_j)) this factors into
//   block11[a,b] =  sum_i phi_a(x_i) phi_b(x_i) (sum_j K_ij)
//   block22[a,b] =  sum_j phi_a(y_j) phi_b(y_j) (sum_i K_ij)
//   block12[a,b] = -sum_i phi_a(x_i) sum_j K_ij phi_b(y_j)
// which is what is accumulated here (same numbers, fewer flops).
// struct PairAcc: pnl_common.h

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

// compile-time number of points, one order for the whole wave: y_j and the column sums stay in registers.  What is live over
// the whole loop -- w_j and w_j phi_b(y_j) -- comes from the global table gwp (layout per point: w, w phi_0 .. w phi_{DPE-2}) with
// wave-uniform addresses, i.e. scalar loads into SGPRs instead of 2 N DPE VGPRs; the per-i constants come from the tile's LDS
// copy tab (layout per point: bary[3], w, phi[DPE]; same address in all lanes: broadcast reads).  The shape functions sum to one,
// so u_{DPE-1}(i) = r_i - sum_{b < DPE-1} u_b(i) saves one FMA per point pair.
// (The table is read through the constant address space: it is never written by a kernel, and loads from that address space
// with a uniform address are scalar loads wherever they stand -- global loads after the first store of a kernel are not.)
template <int DIM, int DPE, int KT, int N>
__device__ __forceinline__ void eval_distant_fixed(const DevProblem &P, const double *__restrict__ tab, const double *gwp_global,
                                                   const double *av, const double *bv, PairAcc<DIM, DPE> &R,
                                                   const double *__restrict__ lpow = nullptr) {
    constexpr int NV = DIM+1, ST = 4+DPE;
    const pnl_const_f64_ptr gwp = (pnl_const_f64_ptr)(unsigned long long)gwp_global;
    // the weights are folded into the accumulations instead of being multiplied into every kernel value: with g = gamma(x_i, y_j)
    //   row sum r_i = sum_j w_j g, column sum c_j = sum_i w_i g, u_b(i) = sum_j g (w_j phi_b(y_j)); w_i enters once per i
    double y[N][DIM], c[N];
#pragma unroll
    for (int j = 0; j < N; j++) {
        c[j] = 0.;
#pragma unroll
        for (int d = 0; d < DIM; d++) {
            double s = 0.;
#pragma unroll
            for (int k = 0; k < NV; k++) s = __builtin_fma(tab[j*ST+k], bv[k*DIM+d], s);
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
            for (int k = 0; k < NV; k++) s = __builtin_fma(tab[i*ST+k], av[k*DIM+d], s);
            x[d] = s;
        }
        const double wi = tab[i*ST+3];
        double r = 0., u[DPE];
#pragma unroll
        for (int b = 0; b < DPE; b++) u[b] = 0.;
#pragma unroll
        for (int j = 0; j < N; j++) {
            double d2 = 0.;
#pragma unroll
            for (int d = 0; d < DIM; d++) { double t = x[d]-y[j][d]; d2 = __builtin_fma(t, t, d2); }
            const double g = kern_eval<KT>(P.k, d2, lpow);
            r = __builtin_fma(gwp[j*DPE], g, r);
            c[j] = __builtin_fma(wi, g, c[j]);
#pragma unroll
            for (int b = 0; b+1 < DPE; b++) u[b] = __builtin_fma(g, gwp[j*DPE+1+b], u[b]);
        }
        u[DPE-1] = r;
#pragma unroll
        for (int b = 0; b+1 < DPE; b++) u[DPE-1] -= u[b];
        int e = 0;
#pragma unroll
        for (int a = 0; a < DPE; a++) {
            const double pa = wi*tab[i*ST+4+a];
#pragma unroll
            for (int b = 0; b < DPE; b++) R.G[a][b] = __builtin_fma(pa, u[b], R.G[a][b]);
            const double pr = pa*r;
#pragma unroll
            for (int b = a; b < DPE; b++) { R.S1[e] = __builtin_fma(pr, tab[i*ST+4+b], R.S1[e]); e++; }
        }
    }
#pragma unroll
    for (int j = 0; j < N; j++) {
        int e = 0;
#pragma unroll
        for (int a = 0; a < DPE; a++) {
            const double pc = tab[j*ST+4+a]*(tab[j*ST+3]*c[j]);
#pragma unroll
            for (int b = a; b < DPE; b++) { R.S2[e] = __builtin_fma(pc, tab[j*ST+4+b], R.S2[e]); e++; }
        }
    }
}

// runtime number of points n, the same for all lanes of the wave; the rule is read from an LDS copy (layout per point:
// bary[3], w, phi[DPE], stride stp) with wave-uniform addresses, i.e. broadcast reads.  x_i, the row sums and u_b are kept
// per i; S2 is accumulated directly per point pair (no per-lane column sums of runtime length).
template <int DIM, int DPE, int KT>
__device__ __forceinline__ void eval_distant_lds(const DevProblem &P, const double *__restrict__ tab, int stp, int n, const double *av,
                                                 const double *bv, PairAcc<DIM, DPE> &R, const double *__restrict__ lpow = nullptr) {
    constexpr int NV = DIM+1;
#pragma unroll 1
    for (int i = 0; i < n; i++) {
        const double *__restrict__ ti = tab+i*stp;
        double x[DIM];
#pragma unroll
        for (int d = 0; d < DIM; d++) {
            double sx = 0.;
#pragma unroll
            for (int k = 0; k < NV; k++) sx = __builtin_fma(ti[k], av[k*DIM+d], sx);
            x[d] = sx;
        }
        const double wi = ti[3];
        double r = 0., u[DPE];
#pragma unroll
        for (int b = 0; b < DPE; b++) u[b] = 0.;
#pragma unroll 2
        for (int j = 0; j < n; j++) {
            const double *__restrict__ tj = tab+j*stp;
            double d2 = 0.;
#pragma unroll
            for (int d = 0; d < DIM; d++) {
                double sy = 0.;
#pragma unroll
                for (int k = 0; k < NV; k++) sy = __builtin_fma(tj[k], bv[k*DIM+d], sy);
                const double t = x[d]-sy;
                d2 = __builtin_fma(t, t, d2);
            }
            const double K = (wi*tj[3])*kern_eval<KT>(P.k, d2, lpow);
            r += K;
            double t[DPE];
#pragma unroll
            for (int b = 0; b < DPE; b++) { t[b] = K*tj[4+b]; u[b] += t[b]; }
            int e = 0;
#pragma unroll
            for (int a = 0; a < DPE; a++)
#pragma unroll
                for (int b = a; b < DPE; b++) { R.S2[e] = __builtin_fma(t[a], tj[4+b], R.S2[e]); e++; }
        }
        int e = 0;
#pragma unroll
        for (int a = 0; a < DPE; a++) {
            const double pa = ti[4+a];
#pragma unroll
            for (int b = 0; b < DPE; b++) R.G[a][b] = __builtin_fma(pa, u[b], R.G[a][b]);
            const double pr = pa*r;
#pragma unroll
            for (int b = a; b < DPE; b++) { R.S1[e] = __builtin_fma(pr, ti[4+b], R.S1[e]); e++; }
        }
    }
}

// eval_distant_blocked: pnl_common.h
// the evaluator of a work-list kernel: the branch-free power where it applies
template <int DIM, int DPE, int KT, int CM>
__device__ __forceinline__ void eval_distant_worklist(const DevProblem &P, const double *__restrict__ tab, int stp, int n, int n4,
                                                      int i_first, int i_step, const double *av, const double *bv,
                                                      PairAcc<DIM, DPE> &R, const double *__restrict__ lpow, double *sc, int cstride) {
    // kern_dispatch spelled out: through the generic lambda (captures by reference) the callers took 40 .. 70 registers more --
    // k_worklist_sorted<2, 3, 0> 256 VGPRs + 32 AGPRs instead of 240, one wave per SIMD instead of two
#define PNL_WL_EVAL(KTE, PT) eval_distant_blocked<DIM, DPE, KTE, PT, CM>(P.k, tab, stp, n, n4, i_first, i_step, av, bv, R, lpow, sc, cstride)
    if constexpr (KT == 0) {
        if (kern_eval_pow_ok(P.k, lpow)) PNL_WL_EVAL(3, true);
        else PNL_WL_EVAL(0, false);
    } else if constexpr (KT == 1) {
        switch (P.k.qm) {
        case 3: PNL_WL_EVAL(13, false); break;
        case 4: PNL_WL_EVAL(14, false); break;
        case 5: PNL_WL_EVAL(15, false); break;
        case 7: PNL_WL_EVAL(17, false); break;
        default: PNL_WL_EVAL(1, false); break;
        }
    } else PNL_WL_EVAL(KT, false);
#undef PNL_WL_EVAL
}
// P2: column sums through LDS (see eval_distant_blocked); bytes of dynamic LDS the one-pair-per-lane kernel needs for them
__host__ __device__ constexpr bool wl_csum_lds(int dpe) { return dpe > 3; }
__host__ __device__ constexpr bool wl_lane_blocked(int dpe, int kt) { return dpe > 3 && kt == 0; }

// ---- sparse output (H2 near field, NA:1663-1964) -------------------------------------------------------------------
// CSR or SSS target with the reference's addToEntry semantics (CSR_LinearOperator_{SCALAR}.pxi:150-170,
// SSS_LinearOperator_{SCALAR}.pxi:104-130): binary search in the row, entries that are not in the pattern are dropped;
// SSS (diag != nullptr) keeps I > J in data and the diagonal in its own vector.
struct SparseOut {
    const int *indptr, *indices;
    double *data, *diag;
    const int *pairs;                    // [np][2] cell pairs, c1 <= c2
    const unsigned long long *masks;     // [np][4] requested entries of the symmetric local matrix (256-bit MASK_t)
};

__device__ __forceinline__ void sparse_add(const SparseOut &S, int I, int J, double v) {
    if (I < 0 || J < 0) return;
    if (S.diag) {
        if (I == J) { atomic_add_f64(&S.diag[I], v); return; }
        if (I < J) return;
    }
    int lo = S.indptr[I];
    const int end = S.indptr[I+1];
    int hi = end;
    while (lo < hi) {
        const int mid = (lo+hi) >> 1;
        if (S.indices[mid] < J) lo = mid+1; else hi = mid;
    }
    if (lo < end && S.indices[lo] == J) atomic_add_f64(&S.data[lo], v);
}


// ---- tiled near-field assembly (clusters.nearFieldPlan): cluster-pair tiles, node membership ---------------------------
struct ClusterTiles {
    const int *chunkA, *chunkB, *pair, *flags;      // [ntiles]; flags bit 0: n1 == n2 (unordered pairs once, both S parts)
    const int *dslotA, *dslotB;                     // [ntiles][64]: slot of the cell in the diagonal-block buffer D or -1
    const int *chunk_cells;                         // [nchunks][64], -1 = padding
    const int *chunk_ndof;                          // [nchunks]
    const int *chunk_dofs;                          // [nchunks][chunk_stride] DoFs of the chunk that belong to its node
    const short *chunk_slot;                        // [nchunks][dpe][64] slot in chunk_dofs or -1
    int chunk_stride, npairs;
    double *D;                                      // [num_dslots][dpe(dpe+1)/2]
    const int *pair_nodes;                          // [npairs][2]
    const int *node_off, *node_dofs;                // sorted DoFs of every node
    int2 *wl_ds;                                    // [wl_cap] diagonal-block slots of the work-list entries (-1: not wanted)
    int *wl_pair;                                   // [wl_cap] cluster pair of the work-list entries
    const int *sing_pair;                           // cluster pair of the touching element pairs of the current launch
    SparseOut S;                                    // the near-field matrix
};

__device__ __forceinline__ bool in_node(const ClusterTiles &CT, int node, int I) {
    int lo = CT.node_off[node];
    const int end = CT.node_off[node+1];
    int hi = end;
    while (lo < hi) {
        const int mid = (lo+hi) >> 1;
        if (CT.node_dofs[mid] < I) lo = mid+1; else hi = mid;
    }
    return lo < end && CT.node_dofs[lo] == I;
}

// does the DoF pair {I, J} belong to the cluster pair k = {n1, n2}: (I in n1, J in n2) or (I in n2, J in n1)
__device__ __forceinline__ bool pair_has(const ClusterTiles &CT, int k, int I, int J) {
    if (I < 0 || J < 0) return false;
    const int n1 = CT.pair_nodes[2*k], n2 = CT.pair_nodes[2*k+1];
    if (in_node(CT, n1, I) && in_node(CT, n2, J)) return true;
    if (n1 == n2) return false;
    return in_node(CT, n2, I) && in_node(CT, n1, J);
}

// ---------------------------------------------------------------------------------------------
// Tile kernel: classification (NO:280-378 vertex test, NO:493-540 + FL2:622-642 order) and distant
// evaluation for one TILE x TILE block of cell pairs.
#define PNL_TT_MAXPTS 96
#define PNL_GEN_MAXPTS 16      // other orders with at most this many points are integrated in the tile kernel too (list C)
#define PNL_GEN_MAXCHUNKS 128
// Pairs whose order is not one of the two unrolled point counts but has at most PNL_GEN_MAXPTS points (orders 5-8 on
// triangles, 92 % of the remaining pairs) stay in the tile as list C: it is counting-sorted by order in LDS so that a wave
// runs 64 pairs of ONE order (same trip counts), one pair per lane, and their local matrices go through the same LDS
// sub-block as lists A and B.  (Unsorted, per-lane trip counts: 13.1 -> 23.2 ms; through the global work list with global
// atomics per pair: 3.8 ms of which 2.5 ms are the atomics.)  Only higher orders go to the global work list.
// lanes walk the tile along generalised diagonals i = (s + m*j) mod TILE (m odd): distinct a- and b-cells per lane, and
// neighbouring b-cells are paired with non-neighbouring a-cells, which keeps same-address LDS atomics rare
#ifndef PNL_DIAG_MULT
#define PNL_DIAG_MULT 21
#endif
template <int DIM, int DPE, int TILE, bool POW = false>
struct TileSmem {
    static constexpr int NV = DIM+1, NC = NV*DIM, ND = DPE*(DPE+1)/2;
    // doubles
    static constexpr int o_v = 0;                          // [2][NC][TILE]
    static constexpr int o_cen = o_v+2*NC*TILE;            // [2][DIM][TILE]
    static constexpr int o_vol = o_cen+2*DIM*TILE;         // [2][TILE]
    static constexpr int o_h = o_vol+2*TILE;               // [2][TILE]
    static constexpr int o_D = o_h+2*TILE;                 // [2][TILE][ND]
    static constexpr int o_Ld = o_D+2*TILE*ND;             // [2][TILE] |ln(h/H0)| in fp64
    static constexpr int o_tt = o_Ld+2*TILE;               // [PNL_TT_MAXPTS][4+DPE] rules integrated one pair per lane
    static constexpr int o_pow = o_tt+PNL_TT_MAXPTS*(4+DPE);   // [PNL_POW_TAB_DOUBLES] tables of the general power (POW: KT == 0 only)
    static constexpr int n_dbl = o_pow+(POW ? PNL_POW_TAB_DOUBLES : 0);
    // ints after the doubles
    static constexpr int o_vid = 0;                        // [2][NV][TILE]
    static constexpr int o_cnt = o_vid+2*NV*TILE;          // [PNL_MAXQ+2]
    static constexpr int o_cur = o_cnt+PNL_MAXQ+2;         // [PNL_MAXQ+2]
    static constexpr int o_misc = o_cur+PNL_MAXQ+2;        // [4]: list length, work-list base, #eligible buckets
    static constexpr int o_cf = o_misc+4;                  // int2 [2][TILE]: bit 0 real cell, bit 1 has a DoF; radius centre -- vertex (float)
    static constexpr int o_lh = o_cf+4*TILE;               // float [2][2][TILE]: ln h, |ln(h/H0)|
    static constexpr int o_ttn = o_lh+4*TILE;          // [PNL_MAXQ+2] points of order q if the tile kernel integrates it, else 0
    static constexpr int o_tto = o_ttn+PNL_MAXQ+2;         // [PNL_MAXQ+2] its offset (points) in the table blob
    static constexpr int o_chunk = o_tto+PNL_MAXQ+2;       // [PNL_GEN_MAXCHUNKS] list C chunks: order << 20 | start << 7 | count-1
    static constexpr int o_dslot = o_chunk+PNL_GEN_MAXCHUNKS;   // [2][TILE] cluster tiles
    static constexpr int o_cell = o_dslot+2*TILE;          // [2][TILE]
    static constexpr int n_int = o_cell+2*TILE;
    // shorts after the ints
    static constexpr int o_slot = 0;                       // [2][DPE][TILE]
    static constexpr int o_list = o_slot+2*DPE*TILE;       // [TILE*TILE] 16 bit: list A from the front, list C from the back;
                                                           // [TILE*TILE] 32 bit: list B from the front, far list from the back
    static constexpr int n_short = o_list+3*TILE*TILE+2;
    static constexpr size_t fixed_bytes = sizeof(double)*n_dbl+sizeof(int)*n_int+sizeof(short)*((n_short+3)/4*4);
};

// Wave-aggregated bucket counter: lanes with the same key q (> 0) are handed consecutive positions from ONE LDS atomic
// per distinct key instead of one atomic per lane (most lanes of a wave share the key).
__device__ __forceinline__ int wave_bucket_add(int *counters, int q, bool fetch) {
    int pos = 0;
    unsigned long long todo = __ballot(q > 0);
    const unsigned long long lt = (1ull << (threadIdx.x & 63))-1ull;
    while (todo) {
        const int leader = __ffsll((long long)todo)-1;
        const int qL = __builtin_amdgcn_readlane(q, leader);
        const unsigned long long same = __ballot(q == qL);
        int base = 0;
        if ((int)(threadIdx.x & 63) == leader) base = atomicAdd(&counters[qL], __popcll(same));
        if (fetch) {
            base = __builtin_amdgcn_readlane(base, leader);
            if (q == qL) pos = base+__popcll(same & lt);
        }
        todo &= ~same;
    }
    return pos;
}

// number of points of a distant rule the tile kernel integrates one pair per lane (fully unrolled);
// every other order goes to the global work list and is integrated one pair per wave.
#ifdef PNL_ONLY_N3
__device__ __forceinline__ bool tile_eligible(int n) { return n == 3; }
#else
__device__ __forceinline__ bool tile_eligible(int n) { return n == 2 || n == 3 || n == 4 || n == 6 || n == 7; }
#endif

#ifndef PNL_PURE_WAVES
#define PNL_PURE_WAVES 4
#endif
// finite horizon (defined further down): relative position of two simplices, sorted-list keys of cut pairs
#define PNL_CUT_SHIFT 60        // sorted-list bin of a cut pair = order + PNL_CUT_SHIFT (orders <= 60)
#define PNL_INTERACT 0
#define PNL_REMOTE 1
#define PNL_CUT 2
#define PNL_WL_LANE_MAXPTS 40
template <int DIM>
__device__ __forceinline__ int rel_position(double h2, const double *av, const double *bv);
template <int DIM, int DPE>
__device__ __forceinline__ unsigned eval_distant_cut(const DevProblem &P, const double *__restrict__ tab, int stp, int n, const double *av,
                                                     const double *bv, PairAcc<DIM, DPE> &R);

// Occupancy of the general tile kernel: its LDS (about 72 KB) lets two workgroups share a CU, so the waves per SIMD come from the
// workgroup size: 512 threads at <= 128 VGPRs give 4 waves per SIMD (the unrolled 6-point evaluator stays spill-free because
// its loop-invariant rule constants live in SGPRs); measured at noRef 7: 256 x 2 waves 69.7 ms, 384 x 3 101.6 ms (spills in
// the hot loop), 512 x 4 59.9 ms.
#ifndef PNL_TILE_THREADS
#define PNL_TILE_THREADS 512  // threads of a tile workgroup (P1 / P0)
#endif
#ifndef PNL_TILE_WAVES
#define PNL_TILE_WAVES 4      // waves per SIMD the tile kernel is register-limited to
#endif
// P2 (78 local entries per pair) needs more than 256 VGPRs: one wave per SIMD without spills beats two with 600 B of scratch,
// hence 256 threads per workgroup there; the general-exponent kernel (KT = 0: exp / log chains) spills at 128 VGPRs (57 ms against
// 31 ms at noRef 6, s = 0.4) and keeps 256 threads x 2 waves as well
// (the finite-horizon variant carries the sub-simplex loops of the cut pairs: 256 threads x 2 waves as well)
// general exponent (KT == 0): 256 threads x 2 waves per SIMD while exp(e ln x) needed the registers (57 against 31 ms at 128
// VGPRs); with the table-driven power (pnl_pow_tab) the 512 x 4 configuration of the other kernels is faster again (P1, s = 0.4,
// 98,304 cells: 46.5 -> 35.6 ms).  -DPNL_KT0_WIDE=0 restores the narrow one.
#ifndef PNL_KT0_WIDE
#define PNL_KT0_WIDE 1
#endif
__host__ __device__ constexpr int tile_threads(int dpe, int kt, bool fh = false) { return (dpe > 3 || (kt == 0 && !PNL_KT0_WIDE) || fh) ? 256 : PNL_TILE_THREADS; }
__host__ __device__ constexpr int tile_waves(int dpe, int kt, bool fh = false) { return dpe > 3 ? 1 : (((kt == 0 && !PNL_KT0_WIDE) || fh) ? 2 : PNL_TILE_WAVES); }
template <int DIM, int DPE, int TILE, int KT, bool CLUSTER, bool FH = false>
__global__ void __launch_bounds__(tile_threads(DPE, KT, FH), tile_waves(DPE, KT, FH))
k_tile_distant(const DevProblem P, const int2 *__restrict__ tiles, double *__restrict__ A, long long ldA,
               double *__restrict__ Dglob, int cell_begin, int cell_end, int acc_stride, int4 *__restrict__ worklist,
               unsigned *__restrict__ wl_count, unsigned wl_cap, int ablate, int ntiles, const ClusterTiles CT,
               unsigned *__restrict__ tile_ctr, const SlotOut SO) {
    // debug switches that skip work (evaluation, accumulation, flush) exist only in PNL_DEBUG_ABLATE builds; bit 256 is
    // PNL_FLAG_SYMMETRIC_FLUSH
#ifdef PNL_DEBUG_ABLATE
    const int abl = ablate;
#else
    const int abl = ablate & 256;
#endif
    using S = TileSmem<DIM, DPE, TILE, KT == 0>;
    constexpr int NV = S::NV, NC = S::NC, ND = S::ND, NT = tile_threads(DPE, KT, FH);
    constexpr int PAIRS = TILE*TILE, PER_THREAD = (PAIRS+NT-1)/NT;
    extern __shared__ double smem[];
    double *s_dbl = smem;
    int *s_int = (int*)(s_dbl+S::n_dbl);
    short *s_short = (short*)(s_int+S::n_int);
    double *s_acc = (double*)(s_short+(S::n_short+3)/4*4);   // [nA][acc_stride]
    double *s_v = s_dbl+S::o_v, *s_cen = s_dbl+S::o_cen, *s_vol = s_dbl+S::o_vol, *s_h = s_dbl+S::o_h, *s_D = s_dbl+S::o_D;
    int *s_vid = s_int+S::o_vid, *s_cnt = s_int+S::o_cnt, *s_cur = s_int+S::o_cur, *s_misc = s_int+S::o_misc;
    int2 *s_cf = (int2*)(s_int+S::o_cf);
    short *s_slot = s_short+S::o_slot;
    unsigned short *s_list = (unsigned short*)(s_short+S::o_list);
    float *s_lh = (float*)(s_int+S::o_lh);
    int *s_ttn = s_int+S::o_ttn, *s_tto = s_int+S::o_tto;
    double *s_tt = s_dbl+S::o_tt, *s_Ld = s_dbl+S::o_Ld;

    const int tid = threadIdx.x;
    int *s_chunk = s_int+S::o_chunk;
    int *s_dslot = s_int+S::o_dslot, *s_cell = s_int+S::o_cell;     // cluster tiles: D slots and cell ids of both sides
    // statistics of order q = 2 + tid (+256) are kept in registers over all tiles of this workgroup: hot counters
    // see one atomic per workgroup, not one per tile
    unsigned long long st_cnt[(PNL_MAXQ+NT)/NT] = {0}, st_ev[(PNL_MAXQ+NT)/NT] = {0};
    // rules integrated inside the tile (the two unrolled point counts and the generic ones): staged once per workgroup
    for (int t = tid; t < PNL_MAXQ+2; t += NT) { s_ttn[t] = P.tt_n[t]; s_tto[t] = P.tt_off[t]; }
    for (int t = tid; t < P.tt_npts*(4+DPE); t += NT) s_tt[t] = P.tt_tab[t];
    double *s_pow = s_dbl+S::o_pow;
    if (KT == 0) pnl_pow_tab_fill(s_pow, P.k.ptab, tid, NT);
    const double *__restrict__ lpow = (KT == 0 && P.k.ptab) ? s_pow : nullptr;
    // the two orders integrated by the unrolled evaluators (lists A and B): the lowest ones with NA / NB points -- nearly all
    // pairs; every wave of those lists works on ONE order, so the rule constants are wave-uniform (scalar loads).  Other orders
    // with a packed rule go through list C, which is sorted by order
    constexpr bool fh = !CLUSTER && FH;               // finite horizon: sparse output CT.S, far list into the sparse pipeline
    int qA0 = 0, qB0 = 0;
    for (int q = 17; q >= 2; q--) {
        const int n = P.tt_n[q];
        qA0 = (n == ((DIM == 2) ? 3 : 2)) ? q : qA0;
        qB0 = (n == ((DIM == 2) ? 6 : 3)) ? q : qB0;
    }
    // persistent workgroups: the first tile by block index, every further one from a global counter (heavy tiles come first in
    // the list and tile costs differ by an order of magnitude: a static stride leaves 14 % of the wave slots idle at the end)
    __shared__ int s_tile_next;
    // cell data of both blocks of a tile -> LDS
    auto stage = [&](int tile_idx) {
    const int ta = CLUSTER ? CT.chunkA[tile_idx] : tiles[tile_idx].x, tb = CLUSTER ? CT.chunkB[tile_idx] : tiles[tile_idx].y;
    const int nA = CLUSTER ? CT.chunk_ndof[ta] : P.blk_ndof[ta], nB = CLUSTER ? CT.chunk_ndof[tb] : P.blk_ndof[tb];
    // ---- stage cell data of both blocks in LDS (SoA: conflict-free per-lane reads) -------------
    for (int t = tid; t < 2*TILE; t += NT) {
        const int side = t/TILE, l = t%TILE;
        const int craw = CLUSTER ? CT.chunk_cells[(size_t)(side ? tb : ta)*TILE+l] : (side ? tb : ta)*TILE+l;
        const bool real = craw >= 0;
        const int c = real ? craw : 0;
        if (CLUSTER) {
            s_cell[side*TILE+l] = craw;
            s_dslot[side*TILE+l] = (side ? CT.dslotB : CT.dslotA)[(size_t)tile_idx*TILE+l];
        }
#pragma unroll
        for (int k = 0; k < NC; k++) s_v[(side*NC+k)*TILE+l] = P.cellv[(size_t)k*P.ncp+c];
#pragma unroll
        for (int d = 0; d < DIM; d++) s_cen[(side*DIM+d)*TILE+l] = P.ccen[(size_t)d*P.ncp+c];
        s_vol[side*TILE+l] = P.cvol[c];
        s_h[side*TILE+l] = P.ch[c];
        // ln h and |ln(h/H0)| per cell are precomputed on the host (finalize)
        const double lh = P.clog[c], Ld = P.clog[(size_t)P.ncp+c];
        s_Ld[side*TILE+l] = Ld;
        s_lh[(side*2+0)*TILE+l] = (float)lh;
        s_lh[(side*2+1)*TILE+l] = (float)Ld;
        int vid0 = -1;
#pragma unroll
        for (int k = 0; k < NV; k++) {
            const int v = real ? P.cvid[(size_t)k*P.ncp+c] : -1-k;
            s_vid[(side*NV+k)*TILE+l] = v;
            vid0 = k == 0 ? v : vid0;
        }
        bool has_dof = false;
#pragma unroll
        for (int k = 0; k < DPE; k++) {
            // boundary DoFs (no slot) are sent to a trash row / column of the LDS sub-block: no branches in the hot loop
            const short sl = CLUSTER ? (real ? CT.chunk_slot[((size_t)(side ? tb : ta)*DPE+k)*TILE+l] : (short)-1) : P.cslot[(size_t)k*P.ncp+c];
            s_slot[(side*DPE+k)*TILE+l] = sl >= 0 ? sl : (short)(side ? nB : nA);
            has_dof = has_dof || sl >= 0;
        }
        // what the classification reads per cell: one 8-byte word (padding and volume-zero cells carry negative vertex ids)
        s_cf[side*TILE+l] = make_int2((vid0 >= 0 ? 1 : 0) | (has_dof ? 2 : 0), __float_as_int((float)P.clog[2*(size_t)P.ncp+c]));
    }
    };
    // The tiles of a workgroup run as a software pipeline: the cell data of the NEXT tile is staged before the flush of this one is
    // issued (a load issued behind the stores / atomics would wait for all of them), the flush zeroes the accumulators it has
    // read, and the barriers order the LDS only -- no wave waits for the flush of a tile to retire.  (Not with
    // PNL_FLAG_SYMMETRIC_FLUSH, whose second sweep reads the sub-block again.)
    const bool pipe = !(abl & 256);
    if (pipe) {
        for (int t = tid; t < ((CLUSTER ? CT.chunk_stride : SO.nU)+1)*acc_stride; t += NT) s_acc[t] = 0.;
        for (int t = tid; t < 2*TILE*ND; t += NT) s_D[t] = 0.;
        for (int t = tid; t < 2*(PNL_MAXQ+2)+4; t += NT) s_cnt[t] = 0;
        if ((int)blockIdx.x < ntiles) stage(blockIdx.x);
    }
#pragma unroll 1
    for (int tile_idx = blockIdx.x; tile_idx < ntiles; tile_idx = s_tile_next) {
    // dense: (block a, block b) of consecutive cells; cluster tiles: (chunk of n1.cells, chunk of n2.cells) of a cluster pair
    const int ta = CLUSTER ? CT.chunkA[tile_idx] : tiles[tile_idx].x, tb = CLUSTER ? CT.chunkB[tile_idx] : tiles[tile_idx].y;
    const int nA = CLUSTER ? CT.chunk_ndof[ta] : P.blk_ndof[ta], nB = CLUSTER ? CT.chunk_ndof[tb] : P.blk_ndof[tb];
    const bool sym = CLUSTER ? (CT.flags[tile_idx] & 1) != 0 : true;
    if (!pipe) {
        stage(tile_idx);
        for (int t = tid; t < (nA+1)*acc_stride; t += NT) s_acc[t] = 0.;
        for (int t = tid; t < 2*TILE*ND; t += NT) s_D[t] = 0.;
        for (int t = tid; t < 2*(PNL_MAXQ+2)+4; t += NT) s_cnt[t] = 0;    // s_cnt, s_cur and s_misc are adjacent
    }
    lds_barrier();

    // ---- classification ------------------------------------------------------------------------
    // pair p -> (i, j) along wrapped diagonals: consecutive lanes get distinct a-cells AND distinct
    // b-cells, so the per-cell LDS accumulators below see (almost) no same-address conflicts.
    // Pairs whose order has NA points (the far-field order 2) go to list A, those with NB points to list B, other orders
    // packed into the tile's rule table to list C, everything else to the global work list.  One barrier.
    constexpr int NA = (DIM == 2) ? 3 : 2, NB = (DIM == 2) ? 6 : 3;
    // list A grows from the front of s_list, list C from its back; list B from the front of s_l32, the far list (pairs for
    // the global work list) from its back
    int *s_l32 = (int*)(s_list+PAIRS+((((size_t)(s_list+PAIRS)) & 2) ? 1 : 0));
    int overflow = 0;
    const int lane = tid & 63;
    const unsigned long long lt = (1ull << lane)-1ull;
    int cnt234[3] = {0, 0, 0};
    // The workgroup size is a multiple of the tile edge: a thread keeps its cell j of block b for the whole tile, so what depends
    // on j alone is read once.  Two passes: the first computes (list, order key) of every pair of the thread -- 16 bits each, kept
    // in registers -- and counts the wave's pairs per list on the scalar unit; then ONE reservation per wave and list (instead
    // of one returning LDS atomic per list and 64 pairs), and the second pass writes the entries.  Vertex ids are compared only
    // where the centres are within the sum of the two radii (a handful of pairs per tile away from the diagonal).
    static_assert(NT%TILE == 0 && PAIRS%NT == 0 && PER_THREAD <= 16, "tile workgroup: whole rows of the tile per sweep");
    constexpr int ROWS = NT/TILE, NG = (PER_THREAD+3)/4;
    const int j = tid%TILE, row0 = tid/TILE;
    double cenb[DIM];
#pragma unroll
    for (int d = 0; d < DIM; d++) cenb[d] = s_cen[(1*DIM+d)*TILE+j];
    const int2 cfb = s_cf[TILE+j];
    const float lhb = s_lh[2*TILE+j], Llb = s_lh[3*TILE+j];
    const int labb = (!CLUSTER && P.cur_class >= 0) ? P.clabel[tb*TILE+j] : 0;
    int nw0 = 0, nw1 = 0, nw2 = 0, nw3 = 0;                // this wave's pairs in list A, list B, the far list, list C
    unsigned long long recs[NG];
#pragma unroll
    for (int g = 0; g < NG; g++) {
    recs[g] = 0ull;
#pragma unroll 2
    for (int it4 = 0; it4 < (PER_THREAD-4*g < 4 ? PER_THREAD-4*g : 4); it4++) {
        const int r = (4*g+it4)*ROWS+row0;
        const int i = (r+PNL_DIAG_MULT*j)%TILE;
        int q = 0, fkey = 0;
        const int2 cfa = s_cf[i];
        const int ca = ta*TILE+i;
        // dense: upper triangle of the cell pairs, a-cells of the caller's range.  Cluster tiles: n1 == n2 -> unordered pairs
        // once (chunk pair a <= b); n1 != n2 -> every ordered pair (X in n1.cells, Y in n2.cells); identical cells share all
        // vertices and are left to the touching-pair lists like every other touching pair
        bool ok = (cfa.x & cfb.x & 1) && !(abl & 8) &&
                  (CLUSTER ? (!sym || ta < tb || i < j) : ((ta < tb || i < j || (fh && i == j)) && (ca >= cell_begin) && (ca < cell_end)));
        // variable order: this launch assembles the pairs of one order class only
        if (!CLUSTER && P.cur_class >= 0 && ok) {
            const int la = P.clabel[ca];
            ok = P.cls_of[P.orient ? labb*P.nlab+la : la*P.nlab+labb] == P.cur_class;
        }
        if (ok) {
            // NA:138-150: skip pairs with boundary DoFs only;  NO:311-323: shared vertices -> singular pair
            const bool any_dof = ((cfa.x | cfb.x) & 2) != 0;
            double d2 = 0.;
#pragma unroll
            for (int d = 0; d < DIM; d++) {
                const double t = s_cen[(0*DIM+d)*TILE+i]-cenb[d];
                d2 += t*t;
            }
            bool shared = false;
            const float rs = __int_as_float(cfa.y)+__int_as_float(cfb.y);
            if ((float)d2 <= rs*rs) {
#pragma unroll
                for (int k = 0; k < NV; k++) {
                    const int va = s_vid[(0*NV+k)*TILE+i];
#pragma unroll
                    for (int m = 0; m < NV; m++) shared = shared || (va == s_vid[(1*NV+m)*TILE+j]);
                }
            }
            // finite horizon (getSparse): REMOTE pairs are dropped, pairs CUT by the horizon and touching pairs go to the
            // sorted sparse pipeline through the far list (keys order + PNL_CUT_SHIFT / 121 + shared vertices - 1, like
            // k_fh_pairs), pairs inside the horizon are integrated here
            int rel = PNL_INTERACT;
            if (!CLUSTER && fh && any_dof) {
                if (shared) {
                    int common = 0;
#pragma unroll
                    for (int k = 0; k < NV; k++)
#pragma unroll
                        for (int m = 0; m < NV; m++) common += (s_vid[(0*NV+k)*TILE+i] == s_vid[(1*NV+m)*TILE+j]);
                    fkey = 121+common-1;
                } else {
                    double av[NC], bv[NC];
#pragma unroll
                    for (int k = 0; k < NC; k++) { av[k] = s_v[(0*NC+k)*TILE+i]; bv[k] = s_v[(1*NC+k)*TILE+j]; }
                    rel = rel_position<DIM>(P.k.horizon2, av, bv);
                }
            }
            if (any_dof && !shared && rel != PNL_REMOTE) {
                if (abl & 16) q = 2;
                else {
                    q = quad_order_try(P.qo, s_lh[i], lhb, s_lh[TILE+i], Llb, d2);
                    if (q < 0) q = quad_order_exact(P.qo, s_h[i], s_h[TILE+j], s_Ld[i], s_Ld[TILE+j], sqrt(d2));
                }
                if (q > P.qmax || q > PNL_MAXQ) { overflow++; q = 0; }
                else if (rel == PNL_CUT) {
                    if (q > PNL_CUT_SHIFT || P.off[q+1]-P.off[q] > PNL_WL_LANE_MAXPTS) { overflow++; q = 0; }
                    else fkey = q+PNL_CUT_SHIFT;
                }
            }
        }
        // statistics: the three lowest orders (nearly all pairs) are counted with ballots on the scalar unit and added to the
        // LDS histogram once per tile; the LDS atomics of the hot loop are left to the rare higher orders
        const int nq = q ? s_ttn[q] : 0;
        const int key = fkey ? fkey : q;
        const int cls = !key ? 0 : (fkey ? 4 : (q == qA0 ? 1 : (q == qB0 ? 2 : (nq > 0 ? 3 : 4))));
        const int qs = (fh && cls == 4) ? 0 : q;         // finite horizon: the sparse pipeline counts what it is handed
        if (!(abl & 32)) {
        cnt234[0] += __popcll(__ballot(qs == 2)); cnt234[1] += __popcll(__ballot(qs == 3)); cnt234[2] += __popcll(__ballot(qs == 4));
        wave_bucket_add(s_cnt, qs > 4 ? qs : 0, false);
        }
        nw0 += __popcll(__ballot(cls == 1)); nw1 += __popcll(__ballot(cls == 2));
        nw2 += __popcll(__ballot(cls == 4)); nw3 += __popcll(__ballot(cls == 3));
        // finite horizon: cut pairs whose order has a packed rule are integrated in this tile (bit 11 -> bit 20 of the entry)
        const unsigned rec = (unsigned)cls | ((unsigned)key << 3) | ((fh && fkey > PNL_CUT_SHIFT+1 && fkey < 121 && nq > 0) ? (1u << 11) : 0u);
        recs[g] |= (unsigned long long)rec << (16*it4);
    }
    }
    {
        int mybase = 0;
        const int mine = lane == 0 ? nw0 : (lane == 1 ? nw1 : (lane == 2 ? nw2 : nw3));
        if (lane < 4 && mine) mybase = atomicAdd(&s_misc[lane], mine);
        int bA = __builtin_amdgcn_readlane(mybase, 0), bB = __builtin_amdgcn_readlane(mybase, 1);
        int bF = __builtin_amdgcn_readlane(mybase, 2), bC = __builtin_amdgcn_readlane(mybase, 3);
        if (!(abl & 128))
#pragma unroll
        for (int g = 0; g < NG; g++)
#pragma unroll
            for (int it4 = 0; it4 < (PER_THREAD-4*g < 4 ? PER_THREAD-4*g : 4); it4++) {
                const unsigned rec = (unsigned)(recs[g] >> (16*it4)) & 0xffffu;
                const int cls = rec & 7, key = (rec >> 3) & 255;
                const int p = ((4*g+it4)*ROWS+row0)*TILE+j;
                if (nw0) {
                    const unsigned long long m = __ballot(cls == 1);
                    if (cls == 1) s_list[bA+__popcll(m & lt)] = (unsigned short)(p | ((qA0-2) << 12));
                    bA += __popcll(m);
                }
                if (nw1) {
                    const unsigned long long m = __ballot(cls == 2);
                    if (cls == 2) s_l32[bB+__popcll(m & lt)] = p | ((qB0-2) << 12);
                    bB += __popcll(m);
                }
                if (nw3) {
                    const unsigned long long m = __ballot(cls == 3);
                    if (cls == 3) s_list[PAIRS-1-(bC+__popcll(m & lt))] = (unsigned short)(p | ((key-2) << 12));
                    bC += __popcll(m);
                }
                if (nw2) {
                    const unsigned long long m = __ballot(cls == 4);
                    if (cls == 4) s_l32[PAIRS-1-(bF+__popcll(m & lt))] = p | (key << 12) | ((rec & (1u << 11)) ? (1 << 20) : 0);
                    bF += __popcll(m);
                }
            }
    }
    if (overflow) atomicAdd(&P.counters[5], (unsigned long long)overflow);
    if (lane == 0) {
#pragma unroll
        for (int k = 0; k < 3; k++) if (cnt234[k]) atomicAdd(&s_cnt[2+k], cnt234[k]);
    }
    lds_barrier();
    // next tile of this workgroup (every thread has read the previous value before the barrier above; the barrier at the end
    // of the tile publishes this one)
    int next_tile = 0;                                    // asked for here, published behind the evaluation: the counter's latency is hidden
    if (tid == 0) next_tile = (int)(gridDim.x+atomicAdd(tile_ctr, 1u));
    // list C: counting sort by order into the free middle of the 32-bit list (behind list B, before the far list), chunks of 64
    // pairs of ONE order, highest order first.  s_cnt[0] = number of these chunks, s_cnt[1] = the chunk queue of the evaluation
    const int nC = __builtin_amdgcn_readfirstlane(s_misc[3]), nBl = __builtin_amdgcn_readfirstlane(s_misc[1]);
    if (nC && tid == 64 && !(abl & 2)) {
        int run = 0, nch = 0;
        for (int q = min(17, P.qmax); q >= 2; q--) {
            const int nq = s_ttn[q], c = s_cnt[q];
            if (!c || nq == 0 || q == qA0 || q == qB0) continue;
            s_cur[q] = run;
            for (int st = 0; st < c && nch < PNL_GEN_MAXCHUNKS; st += 64) s_chunk[nch++] = (q << 20) | ((run+st) << 7) | (min(64, c-st)-1);
            run += c;
        }
        s_cnt[0] = nch;
    }
    bool published = false;                                // has a barrier followed the chunk table (workgroup-uniform)
    {
        // far pairs: one reservation in the global work list per tile, then a coalesced copy
        const int nF = s_misc[2];
        if (fh) {
            // entries of the sorted sparse pipeline, one reservation per wave and 64 entries: (pair index, 0, rule offset,
            // n | key << 16) + the pair itself, what k_fh_pairs writes
#pragma unroll 1
            for (int t0 = (tid >> 6)*64; t0 < nF; t0 += NT) {
                const int t = t0+lane;
                const int ent = t < nF ? s_l32[PAIRS-1-t] : (1 << 20);
                const bool exp = !(ent & (1 << 20));
                const unsigned long long m = __ballot(exp);
                if (m) {
                    unsigned base = 0;
                    const int leader = __ffsll((long long)m)-1;
                    if (lane == leader) base = atomicAdd(wl_count, (unsigned)__popcll(m));
                    base = (unsigned)__builtin_amdgcn_readlane((int)base, leader);
                    const unsigned pos = base+(unsigned)__popcll(m & lt);
                    if (exp && pos < wl_cap) {
                        const int p = ent & 4095, key = (ent >> 12) & 255;
                        const int j = p%TILE, i = (p/TILE+PNL_DIAG_MULT*j)%TILE;
                        const int qq = key >= 121 ? 0 : (key > PNL_CUT_SHIFT+1 ? key-PNL_CUT_SHIFT : key);
                        const int o2 = qq ? P.off[qq] : 0, n2 = qq ? P.off[qq+1]-o2 : 0;
                        worklist[pos] = make_int4((int)pos, 0, o2, n2 | (key << 16));
                        CT.wl_ds[pos] = make_int2(ta*TILE+i, tb*TILE+j);
                    }
                }
            }
        } else if (nF) {
            if (tid == 0) s_cur[0] = (int)atomicAdd(wl_count, (unsigned)nF);
            lds_barrier();
            published = true;
            const unsigned base = (unsigned)s_cur[0];
            for (int t = tid; t < nF; t += NT) {
                const int ent = s_l32[PAIRS-1-t];
                const int p = ent & 4095, q = ent >> 12;
                const int j = p%TILE, i = (p/TILE+PNL_DIAG_MULT*j)%TILE;
                const int off = P.off[q];
                if (base+t < wl_cap) {
                    if (CLUSTER) {
                        worklist[base+t] = make_int4(s_cell[i], s_cell[TILE+j], (int)(base+t), (P.off[q+1]-off) | (q << 16));
                        const int dA = s_dslot[i], dB = s_dslot[TILE+j];
                        CT.wl_ds[base+t] = make_int2(dA, (dB >= 0 && (sym || dA < 0)) ? dB : -1);
                        CT.wl_pair[base+t] = CT.pair[tile_idx];
                    } else
                        worklist[base+t] = make_int4(ta*TILE+i, tb*TILE+j, off, (P.off[q+1]-off) | (q << 16));
                }
            }
        }
    }
    // statistics: one thread per order
#pragma unroll
    for (int r = 0; r < (PNL_MAXQ+NT)/NT; r++) {
        const int q = 2+tid+r*NT;
        if (q <= P.qmax) {
            const int cq = s_cnt[q];
            if (cq) {
                const int ne = s_ttn[q];
                const unsigned long long n = (unsigned long long)(ne ? ne : P.off[q+1]-P.off[q]);
                st_cnt[r] += (unsigned long long)cq;
                st_ev[r] += n*n*cq;
            }
        }
    }

    // ---- evaluation: waves take 64-pair chunks of list A, then of list B ---------------------------------------
    const int wave = tid >> 6;
    // local matrix of pair (i, j) -> LDS sub-block of A' and the per-cell diagonal blocks
    auto accumulate = [&](const PairAcc<DIM, DPE> &R, int i, int j) {
        // NA:1405-1410: symmetric cell pairs count twice
        const double vv = 2.*s_vol[i]*s_vol[TILE+j]*kern_scale<KT>(P.k);
        if (abl & 1) {
            double keep = 0.;
#pragma unroll
            for (int a = 0; a < DPE; a++)
#pragma unroll
                for (int b = 0; b < DPE; b++) keep += R.G[a][b];
#pragma unroll
            for (int e2 = 0; e2 < ND; e2++) keep += R.S1[e2]+R.S2[e2];
            if (keep == 1.2345e300) s_D[0] = keep*vv;
            return;
        }
        int e = 0;
#pragma unroll
        for (int a = 0; a < DPE; a++) {
            const int sa = s_slot[(0*DPE+a)*TILE+i];
#pragma unroll
            for (int b = 0; b < DPE; b++) {
                const int sb = s_slot[(1*DPE+b)*TILE+j];
                lds_add_f64(&s_acc[sa*acc_stride+sb], -vv*R.G[a][b]);
            }
#pragma unroll
            for (int b = a; b < DPE; b++) {
                if (CLUSTER) {
                    // X's diagonal block if X is in cellsInter; Y's if Y is and the pair (Y, X) is not enumerated itself
                    const int dA = s_dslot[i], dB = s_dslot[TILE+j];
                    if (dA >= 0) lds_add_f64(&s_D[(0*TILE+i)*ND+e], vv*R.S1[e]);
                    if (dB >= 0 && (sym || dA < 0)) lds_add_f64(&s_D[(1*TILE+j)*ND+e], vv*R.S2[e]);
                } else if (!(abl & 64)) {
                    lds_add_f64(&s_D[(0*TILE+i)*ND+e], vv*R.S1[e]);
                    lds_add_f64(&s_D[(1*TILE+j)*ND+e], vv*R.S2[e]);
                } else if (R.S1[e]+R.S2[e] == 1.2345e300) s_D[0] = 1.;
                e++;
            }
        }
    };
    if (nC && !(abl & 2)) {
        if (!published) lds_barrier();
        for (int t = tid; t < nC; t += NT) {
            const int ent = s_list[PAIRS-1-t];
            const int q = (ent >> 12)+2;
            s_l32[nBl+atomicAdd(&s_cur[q], 1)] = ent & 4095;
        }
        lds_barrier();
    }
    // One queue of 64-pair chunks for the three lists, the expensive ones first (list C from its highest order down, then list B,
    // then list A); a wave takes the next chunk when it is done with its last one -- no barrier between the lists, and the few
    // chunks of high orders no longer decide when the tile ends
    if (!(abl & 2))
    kern_dispatch<KT, fh>(P.k, lpow, [&](auto ktag) {
    constexpr int KTE = decltype(ktag)::value;              // KT, or 3: the branch-free general power (pnl_common.h)
    const int nAl = __builtin_amdgcn_readfirstlane(s_misc[0]);
    const int nchC = nC ? __builtin_amdgcn_readfirstlane(s_cnt[0]) : 0, nchB = (nBl+63) >> 6, nchA = (nAl+63) >> 6;
    const int total = nchC+nchB+nchA;
#pragma unroll 1
    for (;;) {
        int c = 0;
        if (lane == 0) c = atomicAdd(&s_cnt[1], 1);
        c = __builtin_amdgcn_readfirstlane(c);
        if (c >= total) break;
        int q, p;
        bool act;
        if (c < nchC) {
            const int desc = __builtin_amdgcn_readfirstlane(s_chunk[c]);
            const int start = (desc >> 7) & 8191, cnt = (desc & 127)+1;
            q = desc >> 20;
            act = lane < cnt;
            p = s_l32[nBl+start+(act ? lane : 0)];
        } else if (c < nchC+nchB) {
            const int idx = (c-nchC)*64+lane;
            q = qB0;
            act = idx < nBl;
            p = s_l32[act ? idx : 0] & 4095;
        } else {
            const int idx = (c-nchC-nchB)*64+lane;
            q = qA0;
            act = idx < nAl;
            p = (int)s_list[act ? idx : 0] & 4095;
        }
        const int j = p%TILE, i = (p/TILE+PNL_DIAG_MULT*j)%TILE;
        double av[NC], bv[NC];
#pragma unroll
        for (int k = 0; k < NC; k++) { av[k] = s_v[(0*NC+k)*TILE+i]; bv[k] = s_v[(1*NC+k)*TILE+j]; }
        PairAcc<DIM, DPE> R;
        R.clear();
        const int nq = __builtin_amdgcn_readfirstlane(s_ttn[q]), to = __builtin_amdgcn_readfirstlane(s_tto[q]);
        if (nq == NB) eval_distant_fixed<DIM, DPE, KTE, NB>(P, s_tt+to*(4+DPE), P.tt_wphi+to*DPE, av, bv, R, lpow);
        else if (nq == NA) eval_distant_fixed<DIM, DPE, KTE, NA>(P, s_tt+to*(4+DPE), P.tt_wphi+to*DPE, av, bv, R, lpow);
        else eval_distant_lds<DIM, DPE, KTE>(P, s_tt+to*(4+DPE), 4+DPE, nq, av, bv, R, lpow);
        if (act) accumulate(R, i, j);
    }
    });
    if (fh) {
        // ---- pairs cut by the horizon whose order has a packed rule: sub-simplex loops (eval_distant NO:790-847), one pair
        // per lane, into the same LDS sub-block; the far list still holds them (bit 20)
        const int nF = __builtin_amdgcn_readfirstlane(s_misc[2]);
        unsigned long long ncutp = 0, ncute = 0;
#pragma unroll 1
        for (int t0 = wave*64; t0 < nF; t0 += NT) {
            const int t = t0+lane;
            const int ent = t < nF ? s_l32[PAIRS-1-t] : 0;
            const bool cutl = (ent & (1 << 20)) != 0;
            if (!__ballot(cutl)) continue;
            const int p = ent & 4095, q = cutl ? ((ent >> 12) & 255)-PNL_CUT_SHIFT : 2;
            const int j = p%TILE, i = (p/TILE+PNL_DIAG_MULT*j)%TILE;
            double av[NC], bv[NC];
#pragma unroll
            for (int k = 0; k < NC; k++) { av[k] = s_v[(0*NC+k)*TILE+i]; bv[k] = s_v[(1*NC+k)*TILE+j]; }
            PairAcc<DIM, DPE> R;
            R.clear();
            if (cutl) {
                ncute += eval_distant_cut<DIM, DPE>(P, s_tt+s_tto[q]*(4+DPE), 4+DPE, s_ttn[q], av, bv, R);
                ncutp++;
                accumulate(R, i, j);
            }
            // order histogram: one global atomic per wave and distinct order (one or two)
            unsigned long long todo = __ballot(cutl);
            while (todo) {
                const int leader = __ffsll((long long)todo)-1;
                const int qL = __builtin_amdgcn_readlane(q, leader);
                const unsigned long long same = __ballot(cutl && q == qL);
                if (lane == leader) atomicAdd(&P.counters[8+qL], (unsigned long long)__popcll(same));
                todo &= ~same;
            }
        }
        if (ncutp) { atomicAdd(&P.counters[1], ncutp); atomicAdd(&P.counters[2], ncute); }
    }
    if (tid == 0) s_tile_next = next_tile;
    lds_barrier();

    // ---- flush: sub-block of A' (rows = DoFs of block a, cols = DoFs of block b) and diagonal blocks ----
    if (pipe) {
        const int nx = s_tile_next;
        // diagonal blocks: read and zeroed; cluster tiles before the next tile's slots replace s_dslot
        auto flush_D = [&]() {
            for (int t = tid; t < 2*TILE*ND; t += NT) {
                const double v = s_D[t];
                s_D[t] = 0.;
                if (v != 0. && !(abl & 4)) {
                    const int side = t/(TILE*ND), rem = t-side*TILE*ND;
                    if (CLUSTER) {
                        const int ds = s_dslot[side*TILE+rem/ND];
                        if (ds >= 0) atomic_add_f64(&CT.D[(size_t)ds*ND+rem%ND], v);
                    } else {
                        const int c = (side ? tb : ta)*TILE+rem/ND;
                        atomic_add_f64(&Dglob[(size_t)c*ND+rem%ND], v);
                    }
                }
            }
        };
        if (CLUSTER) { flush_D(); lds_barrier(); }             // every wave has read s_dslot before the next tile's slots arrive
        if (nx < ntiles) stage(nx);
        for (int t = tid; t < 2*(PNL_MAXQ+2)+4; t += NT) s_cnt[t] = 0;
        // rows 0 .. nA (the last one collects the DoFs without a slot) of the LDS sub-block are read and zeroed.  Block-slot storage
        // (pnl_tile2.h): this tile owns its nA x nB sub-block of the storage, plain stores of every entry
        const bool slots = !CLUSTER && !fh && SO.A2 != nullptr;
        const int ca = slots ? SO.colbase[ta] : 0, W = slots ? SO.S-ca : 0;
        double *__restrict__ base = slots ? SO.A2+SO.rowoff[ta]+(SO.colbase[tb]-ca) : nullptr;
        const int *__restrict__ dofA = CLUSTER ? CT.chunk_dofs+(size_t)ta*CT.chunk_stride : P.blk_dofs+(size_t)ta*P.blk_stride;
        const int *__restrict__ dofB = CLUSTER ? CT.chunk_dofs+(size_t)tb*CT.chunk_stride : P.blk_dofs+(size_t)tb*P.blk_stride;
        const unsigned inv = 0xFFFFFFFFu/(unsigned)acc_stride+1u;          // t / acc_stride = umulhi(t, inv) for t < 2^32 / acc_stride
        for (int t = tid; t < (nA+1)*acc_stride; t += NT) {
            const int r = (int)__umulhi((unsigned)t, inv), c = t-r*acc_stride;
            const double v = s_acc[t];
            s_acc[t] = 0.;
            if (r < nA && c < nB && !(abl & 4)) {
                if (slots) slot_store(base+(long long)r*W+c, v);
                else if (v != 0.) {
                    if (CLUSTER || fh) {
                        // entry (I in n1, J in n2) of the near-field matrix and its mirror image (SSS keeps the one with I > J, a
                        // rank-local CSR the ones of its own blocks: addToEntry semantics)
                        sparse_add(CT.S, dofA[r], dofB[c], v);
                        sparse_add(CT.S, dofB[c], dofA[r], v);
                    } else atomic_add_f64(&A[pnl_row(P, dofA[r])*ldA+pnl_col(P, dofB[c])], v);
                }
            }
        }
        if (!CLUSTER) flush_D();
    } else {
    const int *__restrict__ dofA = CLUSTER ? CT.chunk_dofs+(size_t)ta*CT.chunk_stride : P.blk_dofs+(size_t)ta*P.blk_stride;
    const int *__restrict__ dofB = CLUSTER ? CT.chunk_dofs+(size_t)tb*CT.chunk_stride : P.blk_dofs+(size_t)tb*P.blk_stride;
    if (!(abl & 4)) {
    if (!CLUSTER && !fh && SO.A2) {
        // block-slot storage (pnl_tile2.h): this tile owns its nA x nB sub-block of the storage, plain stores of every entry
        const int ca = SO.colbase[ta], W = SO.S-ca;
        double *__restrict__ base = SO.A2+SO.rowoff[ta]+(SO.colbase[tb]-ca);
        for (int t = tid; t < nA*nB; t += NT) {
            const int r = t/nB, c = t-r*nB;
            slot_store(base+(long long)r*W+c, s_acc[r*acc_stride+c]);
        }
    } else
    for (int t = tid; t < nA*nB; t += NT) {
        const int r = t/nB, c = t-r*nB;
        const double v = s_acc[r*acc_stride+c];
        if (v != 0.) {
            if (CLUSTER) {
                // entry (I in n1, J in n2) of the near-field matrix and its mirror image (SSS keeps the one with I > J, a
                // rank-local CSR the ones of its own blocks: addToEntry semantics)
                sparse_add(CT.S, dofA[r], dofB[c], v);
                sparse_add(CT.S, dofB[c], dofA[r], v);
            } else if (fh) {
                sparse_add(CT.S, dofA[r], dofB[c], v);
                sparse_add(CT.S, dofB[c], dofA[r], v);
            } else {
                atomic_add_f64(&A[pnl_row(P, dofA[r])*ldA+pnl_col(P, dofB[c])], v);
            }
        }
    }
    // PNL_FLAG_SYMMETRIC_FLUSH (no mirror pass): the transposed image in its own sweep, consecutive threads along a row of A
    if (!CLUSTER && (abl & 256))
        for (int t = tid; t < nA*nB; t += NT) {
            const int c = t/nA, r = t-c*nA;
            const double v = s_acc[r*acc_stride+c];
            if (v != 0.) atomic_add_f64(&A[(long long)dofB[c]*ldA+dofA[r]], v);
        }
    for (int t = tid; t < 2*TILE*ND; t += NT) {
        const double v = s_D[t];
        if (v != 0.) {
            const int side = t/(TILE*ND), rem = t-side*TILE*ND;
            if (CLUSTER) {
                const int ds = s_dslot[side*TILE+rem/ND];
                if (ds >= 0) atomic_add_f64(&CT.D[(size_t)ds*ND+rem%ND], v);
            } else {
                const int c = (side ? tb : ta)*TILE+rem/ND;
                atomic_add_f64(&Dglob[(size_t)c*ND+rem%ND], v);
            }
        }
    }
    }
    }
    lds_barrier();
    }   // tile loop
#pragma unroll
    for (int r = 0; r < (PNL_MAXQ+NT)/NT; r++) {
        const int q = 2+tid+r*NT;
        if (st_cnt[r]) {
            atomicAdd(&P.counters[8+q], st_cnt[r]);
            atomicAdd(&P.counters[1], st_cnt[r]);
            atomicAdd(&P.counters[2], st_ev[r]);
        }
    }
}

// ---------------------------------------------------------------------------------------------
// Uniform tiles: every one of the 64 x 64 cell pairs of the tile is a distant pair of the lowest order (host-side
// conservative bound on the order formula over the two blocks, see classify_tiles in pnl_hip.hip) -- two thirds of all
// pairs at noRef 6, more on finer meshes.  No classification, no lists: lane = cell i of block a, the four waves split the
// cells j of block b, so everything that depends on j alone (its quadrature points, volume, DoF slots) is a broadcast LDS
// read, the points of cell i and the row sums that feed its diagonal block stay in registers for the whole tile, and the
// column sums are reduced over the wave with DPP.  Same numbers as eval_distant_fixed (NO:722-789), same LDS sub-block.
template <int DIM, int DPE, int KT>
__global__ void __launch_bounds__(PNL_NTHREADS, PNL_PURE_WAVES)
k_tile_pure(const DevProblem P, const int2 *__restrict__ tiles, int ntiles, double *__restrict__ A, long long ldA,
            double *__restrict__ Dglob, int acc_stride, int q_uniform, int symflush, const SlotOut SO) {
#ifndef PNL_DEBUG_ABLATE
    symflush &= 1;                                      // the other bits skip work (debug builds only)
#endif
    constexpr int TILE = 64, NV = DIM+1, NC = NV*DIM, ND = DPE*(DPE+1)/2, NP = (DIM == 2) ? 3 : 2, ST = 4+DPE;
    constexpr int JW = TILE/(PNL_NTHREADS/64);          // cells j per wave
    extern __shared__ double smem[];
    double *s_y = smem;                                 // [TILE][NP*DIM] quadrature points of the b-cells
    double *s_volb = s_y+TILE*NP*DIM;                   // [TILE]
    double *s_Da = s_volb+TILE;                         // [TILE][ND]
    double *s_Db = s_Da+TILE*ND;                        // [TILE][ND]
    double *s_rule = s_Db+TILE*ND;                      // [NP][ST]
    double *s_pow = s_rule+NP*ST;                       // tables of the general power (KT == 0, pnl_pow_tab)
    int *s_slotb = (int*)(s_pow+(KT == 0 ? PNL_POW_TAB_DOUBLES : 0));   // [TILE][DPE] (+ [TILE] has-DoF flags)
    int *s_hb = s_slotb+TILE*DPE;
    double *s_acc = (double*)(s_hb+TILE);               // [nA+1][acc_stride]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    // rule constants: wave-uniform global reads (scalar loads), so the weights products and shape functions at the points
    // live in SGPRs instead of 54 VGPRs
    double wq[NP], wph[NP][DPE], ph[NP][DPE], bary[NP][NV];
    {
        const int off = P.off[q_uniform];
        const double *__restrict__ gb = P.bary+3*(size_t)off, *__restrict__ gw = P.w+off, *__restrict__ gp = P.phi+(size_t)off*DPE;
#pragma unroll
        for (int i = 0; i < NP; i++) {
            wq[i] = gw[i];
#pragma unroll
            for (int a = 0; a < DPE; a++) { ph[i][a] = gp[i*DPE+a]; wph[i][a] = gw[i]*gp[i*DPE+a]; }
#pragma unroll
            for (int k = 0; k < NV; k++) bary[i][k] = gb[3*i+k];
        }
    }
    (void)s_rule;
    const double scale2 = 2.*kern_scale<KT>(P.k);
    if (KT == 0) pnl_pow_tab_fill(s_pow, P.k.ptab, threadIdx.x, PNL_NTHREADS);      // the first barrier of the tile loop publishes it
    const double *__restrict__ ptab = (KT == 0 && P.k.ptab) ? s_pow : nullptr;
    unsigned long long npairs = 0;
#pragma unroll 1
    for (int tile_idx = blockIdx.x; tile_idx < ntiles; tile_idx += gridDim.x) {
        const int2 tl = tiles[tile_idx];
        const int ta = tl.x, tb = tl.y;
        const int nA = P.blk_ndof[ta], nB = P.blk_ndof[tb];
        __syncthreads();                                 // the previous tile's flush is done with the LDS buffers
        if (tid < TILE) {
            const int c = tb*TILE+tid;
            double bv[NC];
#pragma unroll
            for (int k = 0; k < NC; k++) bv[k] = P.cellv[(size_t)k*P.ncp+c];
#pragma unroll
            for (int jp = 0; jp < NP; jp++)
#pragma unroll
                for (int d = 0; d < DIM; d++) {
                    double sy = 0.;
#pragma unroll
                    for (int k = 0; k < NV; k++) sy = __builtin_fma(bary[jp][k], bv[k*DIM+d], sy);
                    s_y[tid*NP*DIM+jp*DIM+d] = sy;
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
        for (int t = tid; t < (nA+1)*acc_stride; t += PNL_NTHREADS) s_acc[t] = 0.;
        for (int t = tid; t < 2*TILE*ND; t += PNL_NTHREADS) s_Da[t] = 0.;
        // a side: lane = cell i
        const int ca = ta*TILE+lane;
        double x[NP][DIM];
        {
            double av[NC];
#pragma unroll
            for (int k = 0; k < NC; k++) av[k] = P.cellv[(size_t)k*P.ncp+ca];
#pragma unroll
            for (int ip = 0; ip < NP; ip++)
#pragma unroll
                for (int d = 0; d < DIM; d++) {
                    double sx = 0.;
#pragma unroll
                    for (int k = 0; k < NV; k++) sx = __builtin_fma(bary[ip][k], av[k*DIM+d], sx);
                    x[ip][d] = sx;
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
        double rr[NP];                                   // sum_j vol_j * (row sums of K): feeds the diagonal block of cell i
#pragma unroll
        for (int ip = 0; ip < NP; ip++) rr[ip] = 0.;
        __syncthreads();
#pragma unroll 1
        for (int jj = 0; jj < JW; jj++) {
            const int j = wave*JW+jj;
            double y[NP][DIM];
#pragma unroll
            for (int jp = 0; jp < NP; jp++)
#pragma unroll
                for (int d = 0; d < DIM; d++) y[jp][d] = s_y[j*NP*DIM+jp*DIM+d];
            const bool valid = ha || (s_hb[j] != 0);     // NA:138-150: pairs with boundary DoFs only are skipped
            npairs += (unsigned long long)__popcll(__ballot(valid));
            const double volb = valid ? s_volb[j] : 0.;
            double c[NP], G[DPE][DPE];
#pragma unroll
            for (int jp = 0; jp < NP; jp++) c[jp] = 0.;
#pragma unroll
            for (int a = 0; a < DPE; a++)
#pragma unroll
                for (int b = 0; b < DPE; b++) G[a][b] = 0.;
#pragma unroll
            for (int ip = 0; ip < NP; ip++) {
                double r = 0., u[DPE];
#pragma unroll
                for (int b = 0; b < DPE; b++) u[b] = 0.;
#pragma unroll
                for (int jp = 0; jp < NP; jp++) {
                    double d2 = 0.;
#pragma unroll
                    for (int d = 0; d < DIM; d++) { const double t = x[ip][d]-y[jp][d]; d2 = __builtin_fma(t, t, d2); }
                    // weights folded into the accumulations (they are scalar constants): no multiply per kernel value
                    const double g = kern_eval<KT>(P.k, d2, ptab);
                    r = __builtin_fma(wq[jp], g, r);
                    c[jp] = __builtin_fma(wq[ip], g, c[jp]);
#pragma unroll
                    for (int b = 0; b < DPE; b++) u[b] = __builtin_fma(g, wph[jp][b], u[b]);
                }
                rr[ip] = __builtin_fma(volb*wq[ip], r, rr[ip]);
#pragma unroll
                for (int a = 0; a < DPE; a++)
#pragma unroll
                    for (int b = 0; b < DPE; b++) G[a][b] = __builtin_fma(wph[ip][a], u[b], G[a][b]);
            }
            // cross block -> LDS sub-block of A'
            const double vv = scale2*vola*volb;
            if (!(symflush & 16)) {
#pragma unroll
                for (int b = 0; b < DPE; b++) {
                    const int sb = s_slotb[j*DPE+b];
#pragma unroll
                    for (int a = 0; a < DPE; a++) lds_add_f64(&s_acc[sa[a]+sb], -vv*G[a][b]);
                }
            } else if (G[0][0] == 1.2345e300) s_acc[0] = vv;
            // diagonal block of cell j: column sums over all cells i of the wave
            const double wa = valid ? vola : 0.;
            double cw[NP];
            if (NP == 3) wave_sum3(wa*c[0], wa*c[1], wa*c[NP-1], cw[0], cw[1], cw[NP-1]);
            else {
#pragma unroll
                for (int jp = 0; jp < NP; jp++) cw[jp] = wave_sum(wa*c[jp]);
            }
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
                    s2 = __builtin_fma(pa*pb*wq[jp], cw[jp], s2);
                }
                s_Db[j*ND+lane] = scale2*s_volb[j]*s2;       // this wave owns cell j: plain store
            }
        }
        // diagonal block of cell i from the accumulated row sums (the four waves hold partial sums over their j's)
        {
            int e = 0;
#pragma unroll
            for (int a = 0; a < DPE; a++)
#pragma unroll
                for (int b = a; b < DPE; b++) {
                    double s1 = 0.;
#pragma unroll
                    for (int ip = 0; ip < NP; ip++) s1 = __builtin_fma(ph[ip][a]*ph[ip][b], rr[ip], s1);
                    lds_add_f64(&s_Da[lane*ND+e], scale2*vola*s1);
                    e++;
                }
        }
        __syncthreads();
        // ---- flush ----
        const int *__restrict__ dofA = P.blk_dofs+(size_t)ta*P.blk_stride;
        const int *__restrict__ dofB = P.blk_dofs+(size_t)tb*P.blk_stride;
        if (SO.A2) {
            // block-slot storage (pnl_tile2.h): plain stores of the tile's own sub-block, every entry
            const int ca = SO.colbase[ta], W = SO.S-ca;
            double *__restrict__ base = SO.A2+SO.rowoff[ta]+(SO.colbase[tb]-ca);
            for (int t = tid; t < nA*nB; t += PNL_NTHREADS) {
                const int r = t/nB, cc = t-r*nB;
                slot_store(base+(long long)r*W+cc, s_acc[r*acc_stride+cc]);
            }
        } else if (!(symflush & 64))
        for (int t = tid; t < nA*nB; t += PNL_NTHREADS) {
            const int r = t/nB, cc = t-r*nB;
            const double v = s_acc[r*acc_stride+cc];
            if (v != 0.) atomic_add_f64(&A[pnl_row(P, dofA[r])*ldA+pnl_col(P, dofB[cc])], v);
        }
        // PNL_FLAG_SYMMETRIC_FLUSH (no mirror pass): the transposed image in its own sweep, consecutive threads along a row of A
        if (!(symflush & 64) && (symflush & 1))
            for (int t = tid; t < nA*nB; t += PNL_NTHREADS) {
                const int cc = t/nA, r = t-cc*nA;
                const double v = s_acc[r*acc_stride+cc];
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
    // statistics: every lane of a wave holds the same count
    if (lane == 0 && npairs) {
        atomicAdd(&P.counters[8+q_uniform], npairs);
        atomicAdd(&P.counters[1], npairs);
        atomicAdd(&P.counters[2], npairs*(unsigned long long)(NP*NP));
        atomicAdd(&P.counters[6], npairs);               // pairs integrated by the uniform-tile kernels (reporting)
        if (q_uniform >= 2 && q_uniform <= 4) atomicAdd(&P.counters[131+q_uniform-2], npairs);
    }
}

// ---------------------------------------------------------------------------------------------
// Finite horizon, l2 ball (interactionDomains.pyx): relative position of two simplices by their vertex distances
// (ball2_retriangulation.getRelativePosition :875-898 = ball2_barycenter :990-1013) and the evaluation of pairs that
// the horizon CUTS (eval_distant NO:790-847): sub-simplices of simplex1 (startLoopSubSimplices_Simplex, :406-567), and
// for every quadrature node x of them the sub-simplices of simplex2 inside the ball around x
// (startLoopSubSimplices_Node, :570-822).  A sub-simplex is kept as the barycentric coordinates of its vertices in a
// frame ROTATED so that the distinguished vertex (the only one inside / the only one outside) is vertex 0 -- the
// reference's index arithmetic (inside+1)%3, (inside+2)%3 is exactly that rotation -- which keeps every array index
// static; the rotation is undone on the barycentric coordinates of each quadrature point.
template <int DIM>
__device__ __forceinline__ int rel_position(double h2, const double *av, const double *bv) {
    // no FMA contraction in the geometric predicates: symmetric meshes produce exact ties (d1 == d2, |x-y|^2 == horizon^2)
    // and the branch taken must be the one plain IEEE evaluation takes
#pragma clang fp contract(off)
    constexpr int NV = DIM+1;
    double dmin2 = 1e300, dmax2 = 0.;
#pragma unroll
    for (int i = 0; i < NV; i++)
#pragma unroll
        for (int k = 0; k < NV; k++) {
            double d2 = 0.;
#pragma unroll
            for (int d = 0; d < DIM; d++) { const double t = av[i*DIM+d]-bv[k*DIM+d]; d2 += t*t; }
            dmin2 = fmin(dmin2, d2);
            dmax2 = fmax(dmax2, d2);
        }
    if (dmin2 >= h2) return PNL_REMOTE;
    if (dmax2 <= h2) return PNL_INTERACT;
    return PNL_CUT;
}

// ball2_retriangulation.findIntersections :911-938 on the segment p0 -> p1; returns the count, t[0] <= t[1]
template <int DIM>
__device__ __forceinline__ int find_intersections(double h2, const double *x, const double *p0, const double *p1, double *t) {
    // no FMA contraction in the geometric predicates: symmetric meshes produce exact ties (d1 == d2, |x-y|^2 == horizon^2)
    // and the branch taken must be the one plain IEEE evaluation takes
#pragma clang fp contract(off)
    double nn = 0., p = 0., q = 0.;
#pragma unroll
    for (int k = 0; k < DIM; k++) {
        const double A = p1[k]-p0[k], B = p0[k]-x[k];
        nn += A*A; p += A*B; q += B*B;
    }
    nn = 1./nn;
    p *= 2.*nn;
    q = (q-h2)*nn;
    const double A = -p*0.5, B = sqrt(A*A-q);
    int num = 0;
    double c = A-B;
    t[0] = t[1] = 0.;
    if (c >= 0 && c <= 1) t[num++] = c;
    c = A+B;
    if (c >= 0 && c <= 1) t[num++] = c;
    return num;
}

template <int DIM>
struct SubSimplices {
    static constexpr int NV = DIM+1;
    int n, rot;                 // number of sub-simplices; rotation of the frame the cases are written in
    // barycentric transform of the reference: node -> b + A node (A[s][k][j]: coordinate k from coordinate j), first in
    // the rotated frame, after unrotate() in the frame of the parent simplex
    double A[3][NV][NV], b[3][NV];
    double vol[3];
    __device__ __forceinline__ void clear() {
        n = 0; rot = 0;
#pragma unroll
        for (int s = 0; s < 3; s++) {
            vol[s] = 0.;
#pragma unroll
            for (int k = 0; k < NV; k++) {
                b[s][k] = 0.;
#pragma unroll
                for (int j = 0; j < NV; j++) A[s][k][j] = 0.;
            }
        }
    }
    __device__ __forceinline__ void identity() {
        clear();
#pragma unroll
        for (int k = 0; k < NV; k++) A[0][k][k] = 1.;
        vol[0] = 1.;
        n = 1;
    }
    // rotated frame -> parent frame: entry (k, j) of the parent frame is entry ((k-rot)%NV, (j-rot)%NV) of the rotated one
    __device__ __forceinline__ void unrotate() {
        if (rot == 0) return;
#pragma unroll
        for (int s = 0; s < 3; s++) {
            double Ao[NV][NV], bo[NV];
#pragma unroll
            for (int k = 0; k < NV; k++) {
                double bv = 0.;
#pragma unroll
                for (int r = 1; r < NV; r++) bv = (rot == r) ? b[s][(k+NV-r)%NV] : bv;
                bo[k] = bv;
#pragma unroll
                for (int j = 0; j < NV; j++) {
                    double av = 0.;
#pragma unroll
                    for (int r = 1; r < NV; r++) av = (rot == r) ? A[s][(k+NV-r)%NV][(j+NV-r)%NV] : av;
                    Ao[k][j] = av;
                }
            }
#pragma unroll
            for (int k = 0; k < NV; k++) {
                b[s][k] = bo[k];
#pragma unroll
                for (int j = 0; j < NV; j++) A[s][k][j] = Ao[k][j];
            }
        }
    }
};

// vertex (k + rot) % NV of a simplex, without dynamic register indexing
template <int DIM>
__device__ __forceinline__ void rotated_vertex(const double *v, int rot, int k, double *out) {
    constexpr int NV = DIM+1;
#pragma unroll
    for (int d = 0; d < DIM; d++) {
        double r = 0.;
#pragma unroll
        for (int m = 0; m < NV; m++) r = (((k+rot)%NV) == m) ? v[m*DIM+d] : r;
        out[d] = r;
    }
}

// startLoopSubSimplices_Node: sub-simplices of simplex bv inside the ball around x
template <int DIM>
__device__ __forceinline__ void subs_node(const DevKernel &K, const double *x, const double *bv, SubSimplices<DIM> &S) {
    // no FMA contraction in the geometric code: symmetric meshes produce exact ties (d1 == d2, |x-y|^2 == horizon^2) and the
    // branch taken must be the one the plain IEEE evaluation of the same expressions takes
#pragma clang fp contract(off)
    constexpr int NV = DIM+1;
    const double h2 = K.horizon2;
    S.clear();
    if (K.interaction == 2) {                           // barycenterDomain :376-392
        double c[DIM], d2 = 0.;
#pragma unroll
        for (int d = 0; d < DIM; d++) {
            double s = 0.;
#pragma unroll
            for (int m = 0; m < NV; m++) s += bv[m*DIM+d];
            c[d] = s/NV;
        }
#pragma unroll
        for (int d = 0; d < DIM; d++) d2 += (x[d]-c[d])*(x[d]-c[d]);
        if (d2 <= h2) S.identity();
        return;
    }
    int ind[NV], numInside = 0;
#pragma unroll
    for (int m = 0; m < NV; m++) {
        double d2 = 0.;
#pragma unroll
        for (int d = 0; d < DIM; d++) d2 += (x[d]-bv[m*DIM+d])*(x[d]-bv[m*DIM+d]);
        ind[m] = d2 <= h2;
        numInside += ind[m];
    }
    if (numInside == NV) { S.identity(); return; }
    double t[2], p0[DIM], p1[DIM], p2[DIM];
    if (DIM == 1) {
        if (numInside == 0) {
            if (find_intersections<DIM>(h2, x, bv, bv+DIM, t) == 2) {
                S.A[0][0][0] = 1.-t[0]; S.A[0][1 % NV][0] = t[0];
                S.A[0][1 % NV][1 % NV] = t[1]; S.A[0][0][1 % NV] = 1.-t[1];
                S.vol[0] = t[1]-t[0];
                S.n = 1;
            }
        } else {
            S.rot = ind[0] ? 0 : 1;                      // inside vertex -> 0
            rotated_vertex<DIM>(bv, S.rot, 0, p0);
            rotated_vertex<DIM>(bv, S.rot, 1, p1);
            find_intersections<DIM>(h2, x, p0, p1, t);
            S.A[0][0][0] = 1.; S.A[0][1 % NV][1 % NV] = t[0]; S.A[0][0][1 % NV] = 1.-t[0];
            S.vol[0] = t[0];
            S.n = 1;
            S.unrotate();
        }
        return;
    }
    if (numInside == 0) return;                         // ":664 There can be a nonzero intersection, but we ignore it"
    constexpr int I1 = 1 % NV, I2 = 2 % NV;            // keeps the 1D instantiation in bounds (this code is 2D only)
    if (numInside == 1) {
        S.rot = ind[0] ? 0 : (ind[I1] ? 1 : 2);
        rotated_vertex<DIM>(bv, S.rot, 0, p0);
        rotated_vertex<DIM>(bv, S.rot, 1, p1);
        rotated_vertex<DIM>(bv, S.rot, 2, p2);
        find_intersections<DIM>(h2, x, p0, p1, t);
        const double c1 = t[0];
        find_intersections<DIM>(h2, x, p0, p2, t);
        const double c2 = t[0];
        const int num = find_intersections<DIM>(h2, x, p1, p2, t);
        if (num == 0) {
            S.A[0][0][0] = 1.; S.A[0][0][I1] = 1.-c1; S.A[0][I1][I1] = c1; S.A[0][I2][I2] = c2; S.A[0][0][I2] = 1.-c2;
            S.vol[0] = c1*c2;
            S.n = 1;
        } else if (num == 2) {
            S.A[0][0][0] = 1.; S.A[0][I1][I1] = c1; S.A[0][0][I1] = 1.-c1; S.A[0][I2][I2] = t[0]; S.A[0][I1][I2] = 1.-t[0];
            S.vol[0] = c1*t[0];
            S.A[1][0][0] = 1.; S.A[1][I1][I1] = 1.-t[0]; S.A[1][I2][I1] = t[0]; S.A[1][I1][I2] = 1.-t[1]; S.A[1][I2][I2] = t[1];
            S.vol[1] = t[1]-t[0];
            S.A[2][0][0] = 1.; S.A[2][I1][I1] = 1.-t[1]; S.A[2][I2][I1] = t[1]; S.A[2][I2][I2] = c2; S.A[2][0][I2] = 1.-c2;
            S.vol[2] = c2*(1.-t[1]);
            S.n = 3;
        } else {
            S.A[0][0][0] = 1.; S.A[0][I1][I1] = c1; S.A[0][0][I1] = 1.-c1; S.A[0][I2][I2] = t[0]; S.A[0][I1][I2] = 1.-t[0];
            S.vol[0] = c1*t[0];
            S.A[1][0][0] = 1.; S.A[1][I1][I1] = 1.-t[0]; S.A[1][I2][I1] = t[0]; S.A[1][I2][I2] = c2; S.A[1][0][I2] = 1.-c2;
            S.vol[1] = c2*(1.-t[0]);
            S.n = 2;
        }
        S.unrotate();
        return;
    }
    // two vertices inside: outside vertex -> 0
    S.rot = !ind[0] ? 0 : (!ind[I1] ? 1 : 2);
    rotated_vertex<DIM>(bv, S.rot, 0, p0);
    rotated_vertex<DIM>(bv, S.rot, 1, p1);
    rotated_vertex<DIM>(bv, S.rot, 2, p2);
    find_intersections<DIM>(h2, x, p0, p1, t);
    const double c1 = t[0];
    find_intersections<DIM>(h2, x, p0, p2, t);
    const double c2 = t[0];
    double d1 = 0., d2 = 0.;
#pragma unroll
    for (int k = 0; k < DIM; k++) {
        d1 += (p2[k]-(c1*p1[k]+(1.-c1)*p0[k]))*(p2[k]-(c1*p1[k]+(1.-c1)*p0[k]));
        d2 += (p1[k]-(c2*p2[k]+(1.-c2)*p0[k]))*(p1[k]-(c2*p2[k]+(1.-c2)*p0[k]));
    }
    if (d1 < d2) {
        S.A[0][I2][I2] = 1.; S.A[0][0][0] = 1.-c2; S.A[0][I2][0] = c2; S.A[0][I1][I1] = c1; S.A[0][0][I1] = 1.-c1;
        S.vol[0] = c1*(1.-c2);
        S.A[1][I1][I1] = 1.; S.A[1][I2][I2] = 1.; S.A[1][0][0] = 1.-c1; S.A[1][I1][0] = c1;
        S.vol[1] = 1.-c1;
    } else {
        S.A[0][I1][I1] = 1.; S.A[0][I2][I2] = c2; S.A[0][0][I2] = 1.-c2; S.A[0][0][0] = 1.-c1; S.A[0][I1][0] = c1;
        S.vol[0] = c2*(1.-c1);
        S.A[1][I1][I1] = 1.; S.A[1][I2][I2] = 1.; S.A[1][0][0] = 1.-c2; S.A[1][I2][0] = c2;
        S.vol[1] = 1.-c2;
    }
    S.n = 2;
    S.unrotate();
}

// startLoopSubSimplices_Simplex: sub-simplices of simplex av that can interact with simplex bv
template <int DIM>
__device__ __forceinline__ void subs_simplex(const DevKernel &K, const double *av, const double *bv, SubSimplices<DIM> &S) {
#pragma clang fp contract(off)
    constexpr int NV = DIM+1;
    const double h2 = K.horizon2;
    S.clear();
    if (K.interaction == 2) {                           // barycenterDomain :358-374
        double c[DIM];
#pragma unroll
        for (int d = 0; d < DIM; d++) {
            double s = 0.;
#pragma unroll
            for (int m = 0; m < NV; m++) s += bv[m*DIM+d];
            c[d] = s/NV;
        }
        bool any = false;
#pragma unroll
        for (int m = 0; m < NV; m++) {
            double d2 = 0.;
#pragma unroll
            for (int d = 0; d < DIM; d++) d2 += (av[m*DIM+d]-c[d])*(av[m*DIM+d]-c[d]);
            any = any || d2 <= h2;
        }
        if (any) S.identity();
        return;
    }
    if (DIM == 1) {
        // :425-446 with nextSubSimplex_Simplex :71-87: up to two sub-intervals [l, r] of simplex1 (in units of its length)
        const double horizon = sqrt(h2);
        const bool lr = av[0] < bv[0];
        const double inv = 1./fabs(av[0]-av[1 % (NV*DIM)]);
        double iv[4];
        iv[0] = av[0]*inv; iv[3] = av[1 % (NV*DIM)]*inv;
        iv[1] = (lr ? fmax(av[0], bv[0]-horizon) : fmax(av[0], bv[0]+horizon))*inv;
        iv[2] = (lr ? fmin(av[1 % (NV*DIM)], bv[1 % (NV*DIM)]-horizon) : fmin(av[1 % (NV*DIM)], bv[1 % (NV*DIM)]+horizon))*inv;
        const int it0 = lr ? 1 : 0;
#pragma unroll
        for (int it = 0; it < 3; it++) {
            if (it < it0 || it >= it0+2) continue;
            const double l = iv[it], r = iv[it+1];
            if (r-l <= 0) continue;
            const int m = S.n++;
#pragma unroll
            for (int s = 0; s < 3; s++)
                if (s == m) {
                    S.A[s][0][0] = r-l; S.A[s][1 % NV][1 % NV] = r-l;
                    S.b[s][0] = iv[3]-r; S.b[s][1 % NV] = l-iv[0];
                    S.vol[s] = r-l;
                }
        }
        return;
    }
    constexpr int I1 = 1 % NV, I2 = 2 % NV;
    bool in[3][3], inI[3];
    int numInside = 0;
#pragma unroll
    for (int i = 0; i < 3; i++) {
        bool any = false;
#pragma unroll
        for (int k = 0; k < 3; k++) {
            double d2 = 0.;
#pragma unroll
            for (int d = 0; d < DIM; d++) d2 += (av[(i % NV)*DIM+d]-bv[(k % NV)*DIM+d])*(av[(i % NV)*DIM+d]-bv[(k % NV)*DIM+d]);
            in[i][k] = d2 <= h2;
            any = any || in[i][k];
        }
        inI[i] = any;
        numInside += any;
    }
    if (numInside == 0) return;
    if (numInside == 3) { S.identity(); return; }
    double t[2], p0[DIM], p1[DIM], p2[DIM];
    if (numInside == 1) {
        S.rot = inI[0] ? 0 : (inI[1] ? 1 : 2);
        rotated_vertex<DIM>(av, S.rot, 0, p0);
        rotated_vertex<DIM>(av, S.rot, 1, p1);
        rotated_vertex<DIM>(av, S.rot, 2, p2);
        double c1 = 0., c2 = 0.;
#pragma unroll
        for (int j = 0; j < 3; j++) {
            bool f = false;
#pragma unroll
            for (int i = 0; i < 3; i++) f = (i == S.rot) ? in[i][j] : f;
            if (f) {
                find_intersections<DIM>(h2, bv+(j % NV)*DIM, p0, p1, t);
                c1 = fmax(c1, t[0]);
                find_intersections<DIM>(h2, bv+(j % NV)*DIM, p0, p2, t);
                c2 = fmax(c2, t[0]);
            }
        }
        if (c1*c2 > 0) {
            S.A[0][0][0] = c1+c2; S.A[0][0][I1] = c2; S.A[0][0][I2] = c1; S.A[0][I1][I1] = c1; S.A[0][I2][I2] = c2;
            S.b[0][0] = 1.-c1-c2;
            S.vol[0] = c1*c2;
            S.n = 1;
            S.unrotate();
        }
        return;
    }
    S.rot = !inI[0] ? 0 : (!inI[1] ? 1 : 2);
    rotated_vertex<DIM>(av, S.rot, 0, p0);
    rotated_vertex<DIM>(av, S.rot, 1, p1);
    rotated_vertex<DIM>(av, S.rot, 2, p2);
    double c1 = 1., c2 = 1.;
#pragma unroll
    for (int j = 0; j < 3; j++) {
        bool f1 = false, f2 = false;
#pragma unroll
        for (int i = 0; i < 3; i++) { f1 = (i == (S.rot+1)%3) ? in[i][j] : f1; f2 = (i == (S.rot+2)%3) ? in[i][j] : f2; }
        if (f1) { find_intersections<DIM>(h2, bv+(j % NV)*DIM, p0, p1, t); c1 = fmin(c1, t[0]); }
        if (f2) { find_intersections<DIM>(h2, bv+(j % NV)*DIM, p0, p2, t); c2 = fmin(c2, t[0]); }
    }
    // :509-516 literally: rows outside / inside1 / inside2 of SIMPLEX2, the second sum without the square
    double q0[DIM], q1[DIM], q2[DIM], d1 = 0., d2 = 0.;
    rotated_vertex<DIM>(bv, S.rot, 0, q0);
    rotated_vertex<DIM>(bv, S.rot, 1, q1);
    rotated_vertex<DIM>(bv, S.rot, 2, q2);
#pragma unroll
    for (int k = 0; k < DIM; k++) {
        d1 += (q0[k]+c1*(q1[k]-q0[k])-q2[k])*(q0[k]+c1*(q1[k]-q0[k])-q2[k]);
        d2 += q0[k]+c2*(q2[k]-q0[k])-q1[k];
    }
    // the second sub-simplex goes to slot S.n, which is 0 or 1: write both candidates with selects
    if (d1 < d2) {
        if (1.-c1 > 0) {
            S.A[0][0][0] = 1.-c1; S.A[0][I1][I1] = 1.-c1; S.A[0][I1][I2] = -c1; S.A[0][I2][I2] = 1.; S.b[0][I1] = c1;
            S.vol[0] = 1.-c1;
            S.n = 1;
        }
        if (c1*(1.-c2) > 0.) {
            const int m = S.n++;
#pragma unroll
            for (int s = 0; s < 2; s++)
                if (s == m) {
#pragma unroll
                    for (int k = 0; k < NV; k++) { S.b[s][k] = 0.; for (int j = 0; j < NV; j++) S.A[s][k][j] = 0.; }
                    S.A[s][0][0] = 1.-c2; S.A[s][I2][I2] = 1.; S.A[s][I2][0] = c2; S.A[s][0][I1] = 1.-c1; S.A[s][I1][I1] = c1;
                    S.vol[s] = c1*(1.-c2);
                }
        }
    } else {
        if (1.-c2 > 0) {
            S.A[0][0][0] = 1.-c2; S.A[0][I2][I2] = 1.-c2; S.A[0][I2][I1] = -c2; S.A[0][I1][I1] = 1.; S.b[0][I2] = c2;
            S.vol[0] = 1.-c2;
            S.n = 1;
        }
        if (c2*(1.-c1) > 0.) {
            const int m = S.n++;
#pragma unroll
            for (int s = 0; s < 2; s++)
                if (s == m) {
#pragma unroll
                    for (int k = 0; k < NV; k++) { S.b[s][k] = 0.; for (int j = 0; j < NV; j++) S.A[s][k][j] = 0.; }
                    S.A[s][0][0] = 1.-c1; S.A[s][I1][I1] = 1.; S.A[s][I1][0] = c1; S.A[s][0][I2] = 1.-c2; S.A[s][I2][I2] = c2;
                    S.vol[s] = c2*(1.-c1);
                }
        }
    }
    S.unrotate();
}

// local shape functions at barycentric coordinates (DoFMaps.pyx:1854-2025): P1 (DPE = DIM+1), P2 on triangles (DPE = 6)
template <int DIM, int DPE>
__device__ __forceinline__ void shape_eval(const double *lam, double *phi) {
    if (DPE == DIM+1) {
#pragma unroll
        for (int k = 0; k < DPE; k++) phi[k] = lam[k];
    } else {
        phi[0] = lam[0]*(2.*lam[0]-1.); phi[1] = lam[1]*(2.*lam[1]-1.); phi[2 % DPE] = lam[2 % (DIM+1)]*(2.*lam[2 % (DIM+1)]-1.);
        phi[3 % DPE] = 4.*lam[0]*lam[1]; phi[4 % DPE] = 4.*lam[1]*lam[2 % (DIM+1)]; phi[5 % DPE] = 4.*lam[0]*lam[2 % (DIM+1)];
    }
}

// barycentric coordinates (parent frame) and global point of quadrature node mu on sub-simplex s, in the reference's order of
// operations: lam = b + A mu (transformQuadratureRule.compute, quadrature.pyx:197-206), pt = sum_k lam_k v_k (Q:76-87)
template <int DIM>
__device__ __forceinline__ void sub_point(const SubSimplices<DIM> &S, int s, const double *mu, const double *v, double *lam, double *pt) {
#pragma clang fp contract(off)
    constexpr int NV = DIM+1;
#pragma unroll
    for (int k = 0; k < NV; k++) {
        double a = 0.;
#pragma unroll
        for (int ss = 0; ss < 3; ss++)
            if (ss == s) {
                a = S.b[ss][k];
#pragma unroll
                for (int j = 0; j < NV; j++) a += S.A[ss][k][j]*mu[j];
            }
        lam[k] = a;
    }
#pragma unroll
    for (int d = 0; d < DIM; d++) {
        double a = 0.;
#pragma unroll
        for (int k = 0; k < NV; k++) a += lam[k]*v[k*DIM+d];
        pt[d] = a;
    }
}

// NO:790-847, cut branch, one pair per lane; tab = LDS copy of the rule of the pair's order (bary[3], w, ...; stride stp)
template <int DIM, int DPE>
__device__ __forceinline__ unsigned eval_distant_cut(const DevProblem &P, const double *__restrict__ tab, int stp, int n, const double *av,
                                                     const double *bv, PairAcc<DIM, DPE> &R) {
    constexpr int NV = DIM+1;
    SubSimplices<DIM> S1, S2;
    subs_simplex<DIM>(P.k, av, bv, S1);
    unsigned nevals = 0;
#pragma unroll 1
    for (int a = 0; a < S1.n; a++)
#pragma unroll 1
        for (int i = 0; i < n; i++) {
            double mu[NV], lx[NV], x[DIM], px[DPE];
#pragma unroll
            for (int k = 0; k < NV; k++) mu[k] = tab[i*stp+k];
            sub_point<DIM>(S1, a, mu, av, lx, x);
            shape_eval<DIM, DPE>(lx, px);
            const double wi = tab[i*stp+3]*(a == 0 ? S1.vol[0] : S1.vol[1]);
            subs_node<DIM>(P.k, x, bv, S2);
#pragma unroll 1
            for (int b = 0; b < S2.n; b++) {
                const double wb = wi*(b == 0 ? S2.vol[0] : (b == 1 ? S2.vol[1] : S2.vol[2]));
#pragma unroll 1
                for (int j = 0; j < n; j++) {
                    double nu[NV], ly[NV], y[DIM], py[DPE];
#pragma unroll
                    for (int k = 0; k < NV; k++) nu[k] = tab[j*stp+k];
                    sub_point<DIM>(S2, b, nu, bv, ly, y);
                    shape_eval<DIM, DPE>(ly, py);
                    double d2 = 0.;
#pragma unroll
                    for (int d = 0; d < DIM; d++) { const double t = x[d]-y[d]; d2 = __builtin_fma(t, t, d2); }
                    const double K = wb*tab[j*stp+3]*kern_eval<0>(P.k, d2);
                    nevals++;
                    int e = 0;
#pragma unroll
                    for (int aa = 0; aa < DPE; aa++) {
                        const double kx = K*px[aa], ky = K*py[aa];
#pragma unroll
                        for (int bb = 0; bb < DPE; bb++) R.G[aa][bb] = __builtin_fma(kx, py[bb], R.G[aa][bb]);
#pragma unroll
                        for (int bb = aa; bb < DPE; bb++) {
                            R.S1[e] = __builtin_fma(kx, px[bb], R.S1[e]);
                            R.S2[e] = __builtin_fma(ky, py[bb], R.S2[e]);
                            e++;
                        }
                    }
                }
            }
        }
    return nevals;
}

// NA:503-520 addToMatrixElemElemSymMasked for one entry (p <= q) of the local matrix over 2*dpe local DoFs
__device__ __forceinline__ void sparse_add_sym(const SparseOut &S, const unsigned long long *mask, int n2, int p, int q, int I, int J,
                                               double v) {
    const int k = n2*p-((p*(p-1)) >> 1)+(q-p);
    if (mask && !((mask[k >> 6] >> (k & 63)) & 1ull)) return;     // no mask list: every entry is requested (getSparse)
    if (p == q) sparse_add(S, I, I, v);
    else { sparse_add(S, I, J, v); sparse_add(S, J, I, v); }
}

// diagonal blocks of the mask-less sparse path: A[dof_a(c), dof_b(c)] += D_c[a, b] with the symmetric-entry semantics of
// sparse_add_sym (one pattern search per cell entry instead of one per pair and entry)
template <int DPE>
__global__ void __launch_bounds__(PNL_NTHREADS)
k_scatter_diag_sparse(const DevProblem P, const double *__restrict__ D, int nc, const SparseOut S) {
    constexpr int ND = DPE*(DPE+1)/2;
    const long long t = (long long)blockIdx.x*PNL_NTHREADS+threadIdx.x;
    if (t >= (long long)nc*ND) return;
    const double v = D[t];
    if (v == 0.) return;
    const int c = (int)(t/ND);
    int idx = (int)(t-(long long)c*ND), a = 0;
    while (idx >= DPE-a) { idx -= DPE-a; a++; }
    const int b = a+idx;
    const int I = P.cdof[(size_t)a*P.ncp+c], J = P.cdof[(size_t)b*P.ncp+c];
    if (a == b) sparse_add(S, I, I, v);
    else { sparse_add(S, I, J, v); sparse_add(S, J, I, v); }
}

// classification of explicit cell pairs (NO:280-378 + NO:493-540 with the exact fp64 order formula); entry =
// (pair index, 0, rule offset, n | key << 16), key = order for distant pairs, 121 + (#shared vertices - 1) for touching ones
template <int DIM, int DPE>
__global__ void __launch_bounds__(PNL_NTHREADS)
k_mp_classify(const DevProblem P, const int *__restrict__ pairs, int npairs, int4 *__restrict__ out) {
    constexpr int NV = DIM+1;
    const int t = blockIdx.x*PNL_NTHREADS+threadIdx.x;
    if (t >= npairs) return;
    const int c1 = pairs[2*t], c2 = pairs[2*t+1];
    bool any = false;
#pragma unroll
    for (int k = 0; k < DPE; k++) any = any || P.cdof[(size_t)k*P.ncp+c1] >= 0 || P.cdof[(size_t)k*P.ncp+c2] >= 0;
    // variable order: one pass per class, a pair belongs to the class of its two labels; a non-symmetric class table runs two passes
    // per class (DevProblem::orient): the second one takes the class of (label c2, label c1) and leaves identical pairs out
    if (P.cur_class >= 0 && P.cls_of[P.orient ? P.clabel[c2]*P.nlab+P.clabel[c1] : P.clabel[c1]*P.nlab+P.clabel[c2]] != P.cur_class) any = false;
    if (P.orient && c1 == c2) any = false;
    int key = 0, off = 0, n = 0;
    if (any) {
        int common = 0;
        if (c1 == c2) common = NV;
        else {
#pragma unroll
            for (int a = 0; a < NV; a++)
#pragma unroll
                for (int b = 0; b < NV; b++) common += (P.cvid[(size_t)a*P.ncp+c1] == P.cvid[(size_t)b*P.ncp+c2]);
        }
        if (common > 0) key = 121+common-1;
        else {
            // finite horizon: REMOTE pairs are ignored (NO:515-517), pairs CUT by the horizon go to their own bins
            int rel = PNL_INTERACT;
            if (P.k.horizon2 < 1e300) {
                double av[NV*DIM], bv[NV*DIM];
#pragma unroll
                for (int k = 0; k < NV*DIM; k++) { av[k] = P.cellv[(size_t)k*P.ncp+c1]; bv[k] = P.cellv[(size_t)k*P.ncp+c2]; }
                rel = rel_position<DIM>(P.k.horizon2, av, bv);
            }
            if (rel != PNL_REMOTE) {
                double d2 = 0.;
#pragma unroll
                for (int d = 0; d < DIM; d++) { const double u = P.ccen[(size_t)d*P.ncp+c1]-P.ccen[(size_t)d*P.ncp+c2]; d2 += u*u; }
                const int q = quad_order(P.qo, P.H0, P.ch[c1], P.ch[c2], sqrt(d2));
                if (q > P.qmax || q > PNL_MAXQ || (rel == PNL_CUT && (q > PNL_CUT_SHIFT || P.off[q+1]-P.off[q] > PNL_WL_LANE_MAXPTS)))
                    atomicAdd(&P.counters[5], 1ull);
                else { key = q+(rel == PNL_CUT ? PNL_CUT_SHIFT : 0); off = P.off[q]; n = P.off[q+1]-off; }
            }
        }
    }
    out[t] = make_int4(t, 0, off, n | (key << 16));
}

// Finite horizon (getSparse, NA:1062-1260): the candidate cell pairs c1 <= c2 of the block tiles that the horizon can reach
// are generated and classified on the device -- the reference walks the cells of covering cluster pairs (NA:1150-1170) and
// drops the REMOTE ones in getPanelType (NO:515-517).  One workgroup per tile of T x T cells; surviving pairs are appended to
// `pairs` and to the work list in the format of k_mp_classify (pair index, 0, rule offset, n | key << 16).
template <int DIM, int DPE>
__global__ void __launch_bounds__(PNL_NTHREADS)
k_fh_pairs(const DevProblem P, const int2 *__restrict__ tiles, int T, int2 *__restrict__ pairs, int4 *__restrict__ wl,
           unsigned *__restrict__ count, unsigned cap) {
    constexpr int NV = DIM+1;
    const int2 tl = tiles[blockIdx.x];
    const int lane = threadIdx.x & 63;
    for (int idx = threadIdx.x; idx < T*T; idx += PNL_NTHREADS) {      // T*T is a multiple of the block size: uniform trip count
        const int c1 = tl.x*T+idx%T, c2 = tl.y*T+idx/T;
        bool push = false;
        int key = 0, off = 0, n = 0;
        if (c1 < P.nc && c2 < P.nc && c1 <= c2) {
            bool any = false;
#pragma unroll
            for (int k = 0; k < DPE; k++) any = any || P.cdof[(size_t)k*P.ncp+c1] >= 0 || P.cdof[(size_t)k*P.ncp+c2] >= 0;
            if (any) {
                int common = 0;
                if (c1 == c2) common = NV;
                else {
#pragma unroll
                    for (int a = 0; a < NV; a++)
#pragma unroll
                        for (int b = 0; b < NV; b++) common += (P.cvid[(size_t)a*P.ncp+c1] == P.cvid[(size_t)b*P.ncp+c2]);
                }
                if (common > 0) { key = 121+common-1; push = true; }
                else {
                    double av[NV*DIM], bv[NV*DIM];
#pragma unroll
                    for (int k = 0; k < NV*DIM; k++) { av[k] = P.cellv[(size_t)k*P.ncp+c1]; bv[k] = P.cellv[(size_t)k*P.ncp+c2]; }
                    const int rel = rel_position<DIM>(P.k.horizon2, av, bv);
                    if (rel != PNL_REMOTE) {
                        double d2 = 0.;
#pragma unroll
                        for (int d = 0; d < DIM; d++) { const double u = P.ccen[(size_t)d*P.ncp+c1]-P.ccen[(size_t)d*P.ncp+c2]; d2 += u*u; }
                        const int q = quad_order(P.qo, P.H0, P.ch[c1], P.ch[c2], sqrt(d2));
                        if (q > P.qmax || q > PNL_MAXQ || (rel == PNL_CUT && (q > PNL_CUT_SHIFT || P.off[q+1]-P.off[q] > PNL_WL_LANE_MAXPTS)))
                            atomicAdd(&P.counters[5], 1ull);
                        else { key = q+(rel == PNL_CUT ? PNL_CUT_SHIFT : 0); off = P.off[q]; n = P.off[q+1]-off; push = true; }
                    }
                }
            }
        }
        const unsigned long long m = __ballot(push);
        if (m) {
            unsigned base = 0;
            const int leader = __builtin_ctzll(m);
            if (lane == leader) base = atomicAdd(count, (unsigned)__popcll(m));
            base = __shfl(base, leader);
            if (push) {
                const unsigned pos = base+__popcll(m & ((1ull << lane)-1ull));
                if (pos < cap) { pairs[pos] = make_int2(c1, c2); wl[pos] = make_int4((int)pos, 0, off, n | (key << 16)); }
            }
        }
    }
}

// ---- work list of the orders the tile kernel does not unroll -------------------------------------------------------
// entry = (c1, c2, rule offset, n | order << 16).  The list is counting-sorted by order so that a workgroup integrates
// pairs of ONE order at a time: the rule is staged in LDS once and all 16 pairs of a chunk run the same trip count.
#define PNL_WL_BINS 128
__global__ void __launch_bounds__(PNL_NTHREADS)
k_wl_hist(const int4 *__restrict__ wl, const unsigned *__restrict__ wl_count, unsigned wl_cap, unsigned *__restrict__ hist) {
    __shared__ unsigned h[PNL_WL_BINS];
    if (threadIdx.x < PNL_WL_BINS) h[threadIdx.x] = 0;
    __syncthreads();
    const unsigned count = min(*wl_count, wl_cap);
    for (unsigned i = blockIdx.x*PNL_NTHREADS+threadIdx.x; i < count; i += gridDim.x*PNL_NTHREADS)
        atomicAdd(&h[(wl[i].w >> 16) & (PNL_WL_BINS-1)], 1u);
    __syncthreads();
    if (threadIdx.x < PNL_WL_BINS && h[threadIdx.x]) atomicAdd(&hist[threadIdx.x], h[threadIdx.x]);
}

// offs[q] = first sorted position of order q, chunk_off[q] = first 16-pair chunk of order q; both have PNL_WL_BINS+1 entries
__global__ void k_wl_scan(const unsigned *__restrict__ hist, unsigned *__restrict__ offs, unsigned *__restrict__ chunk_off,
                          unsigned *__restrict__ cursor) {
    if (threadIdx.x == 0) {
        unsigned run = 0, crun = 0;
        for (int q = 0; q < PNL_WL_BINS; q++) {
            offs[q] = run; chunk_off[q] = crun; cursor[q] = 0;
            run += hist[q]; crun += (hist[q]+15)/16;
        }
        offs[PNL_WL_BINS] = run; chunk_off[PNL_WL_BINS] = crun;
    }
}

// Stable within a wave's slice: every wave owns a contiguous part of its workgroup's input range and hands out positions in
// input order (ballot ranks), so runs of consecutive entries with one key stay consecutive.  The producers append whole
// waves of neighbouring pairs (k_fh_pairs: 64 consecutive cells c1 against one c2), and the consumers run 64 consecutive
// sorted entries per wave: coalesced cell data, neighbouring pattern rows, shared cells that can be summed over the wave.
__global__ void __launch_bounds__(PNL_NTHREADS)
k_wl_scatter(const int4 *__restrict__ wl, const unsigned *__restrict__ wl_count, unsigned wl_cap, const unsigned *__restrict__ offs,
             unsigned *__restrict__ cursor, int4 *__restrict__ sorted) {
    constexpr int NW = PNL_NTHREADS/64;
    __shared__ unsigned h[NW][PNL_WL_BINS], base[NW][PNL_WL_BINS];
    const unsigned count = min(*wl_count, wl_cap);
    const unsigned per = (count+gridDim.x-1)/gridDim.x;
    const unsigned b0 = blockIdx.x*per, b1 = min(count, b0+per);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const unsigned perw = ((b1 > b0 ? b1-b0 : 0u)+NW-1)/NW;
    const unsigned i0 = min(b1, b0+wave*perw), i1 = min(b1, i0+perw);
    const unsigned long long lt = (1ull << lane)-1ull;
    for (int t = threadIdx.x; t < NW*PNL_WL_BINS; t += PNL_NTHREADS) (&h[0][0])[t] = 0;
    __syncthreads();
    // pass 1: keys per wave slice; pass 2 hands out positions with the same loop
    auto sweep = [&](bool place) {
        for (unsigned i = i0; i < i1; i += 64) {
            const bool act = i+lane < i1;
            int4 e = make_int4(0, 0, 0, 0);
            if (act) e = wl[i+lane];
            const int q = act ? ((e.w >> 16) & (PNL_WL_BINS-1)) : -1;
            unsigned long long todo = __ballot(act);
            while (todo) {
                const int leader = __ffsll((long long)todo)-1;
                const int qL = __builtin_amdgcn_readlane(q, leader);
                const unsigned long long same = __ballot(q == qL);
                if (place) {
                    const unsigned start = base[wave][qL]+h[wave][qL];          // the wave is the only writer of its row
                    if (q == qL) sorted[start+__popcll(same & lt)] = e;
                }
                __builtin_amdgcn_wave_barrier();
                if (lane == leader) h[wave][qL] += (unsigned)__popcll(same);
                __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
                __builtin_amdgcn_wave_barrier();
                todo &= ~same;
            }
        }
    };
    sweep(false);
    __syncthreads();
    if (threadIdx.x < PNL_WL_BINS) {
        unsigned tot = 0;
        for (int w = 0; w < NW; w++) tot += h[w][threadIdx.x];
        unsigned run = tot ? offs[threadIdx.x]+atomicAdd(&cursor[threadIdx.x], tot) : 0u;
        for (int w = 0; w < NW; w++) { base[w][threadIdx.x] = run; run += h[w][threadIdx.x]; h[w][threadIdx.x] = 0; }
    }
    __syncthreads();
    sweep(true);
}

// Distant pairs of the high orders from the sorted work list (NO:722-789; few pairs, thousands of point pairs each): a
// workgroup takes chunks of 16 pairs of one order, one DPP row (16 lanes) per pair; the lanes split the rows of the tensor
// rule (read from the LDS copy of the rule), row-wise DPP reduction of the local matrix, atomic scatter.  A' receives the cross block on the (c1-DoF, c2-DoF) side only.
template <int DIM, int DPE, int KT, bool SPARSE>
__global__ void __launch_bounds__(PNL_NTHREADS)
k_worklist_sorted(const DevProblem P, const int4 *__restrict__ sorted, const unsigned *__restrict__ offs,
                  const unsigned *__restrict__ chunk_off, double *__restrict__ A, long long ldA, double *__restrict__ Dglob,
                  int tab_max_pts, const SparseOut S, int qlast, int nmin_flags, const ClusterTiles CT) {
    const int nmin = nmin_flags & 0xffff;
    const bool symflush = (nmin_flags >> 16) & 1;  // dense output: write (I, J) and (J, I) (PNL_FLAG_SYMMETRIC_FLUSH)
    const bool cluster = CT.npairs > 0;          // work list of the cluster tiles: entry.z indexes wl_pair / wl_ds
    constexpr int NV = DIM+1, NC = NV*DIM, ND = DPE*(DPE+1)/2, NG = DPE*DPE, NACC = NG+2*ND, LPP = 16, PPC = PNL_NTHREADS/LPP, NREP = (NACC+LPP-1)/LPP, ST = 4+DPE;   // LPP lanes per pair
    extern __shared__ double s_rule[];           // [tab_max_pts][ST]: bary[3], w, phi[DPE]; P2: + [PPC][tab_max_pts] column sums
    __shared__ unsigned s_coff[PNL_WL_BINS+1];
    __shared__ double s_pow[KT == 0 ? PNL_POW_TAB_DOUBLES : 1];      // tables of the general power (pnl_pow_tab)
    const int tid = threadIdx.x, sub = tid & (LPP-1), g = tid/LPP;
    if (KT == 0) pnl_pow_tab_fill(s_pow, P.k.ptab, tid, PNL_NTHREADS);       // published by the barrier below
    const double *__restrict__ lpow = (KT == 0 && P.k.ptab) ? s_pow : nullptr;
    // chunks of PPC pairs over the orders this kernel owns: 2 <= q <= qlast (the sparse path keeps skipped pairs in bin 0 and
    // the touching pairs in the bins above qlast) with at least nmin points (the smaller rules go to k_worklist_lane)
    if (tid == 0) {
        unsigned run = 0;
        for (int q = 0; q < PNL_WL_BINS; q++) {
            s_coff[q] = run;
            const int nq = (q >= 2 && q <= P.qmax && q <= PNL_MAXQ && q <= qlast) ? P.off[q+1]-P.off[q] : 0;
            if (nq > 0 && nq >= nmin) run += (offs[q+1]-offs[q]+PPC-1u)/PPC;
        }
        s_coff[PNL_WL_BINS] = run;
    }
    __syncthreads();
    const unsigned nchunks = s_coff[PNL_WL_BINS];
    int staged_q = -1;
    // heaviest chunks first (the highest orders have thousands of point pairs per pair): the launch does not end on a few
    // workgroups that picked up the expensive pairs last
    for (unsigned rchunk = blockIdx.x; rchunk < nchunks; rchunk += gridDim.x) {
        const unsigned chunk = nchunks-1u-rchunk;
        // order of this chunk: last q with chunk_off[q] <= chunk
        int lo = 0, hi = PNL_WL_BINS-1;
        while (lo < hi) {
            const int mid = (lo+hi+1) >> 1;
            if (s_coff[mid] <= chunk) lo = mid; else hi = mid-1;
        }
        const int q = lo;
        const unsigned first = offs[q]+(unsigned)PPC*(chunk-s_coff[q]);
        const int cnt = (int)min((unsigned)PPC, offs[q+1]-first);
        const int4 e0 = sorted[first];
        const int n = e0.w & 0xffff, nn = n*n, off = P.off[q];
        const int n4 = (n+3) & ~3;                  // zero-weight copies of point 0 fill the last group of four columns
        const bool in_lds = n4 <= tab_max_pts;
        if (q != staged_q) {
            __syncthreads();
            if (in_lds)
                for (int t = tid; t < n4*ST; t += PNL_NTHREADS) {
                    const int pt = t/ST, k = t-pt*ST, src = off+(pt < n ? pt : 0);
                    s_rule[t] = k < 3 ? P.bary[3*(size_t)src+k] : (k == 3 ? (pt < n ? P.w[src] : 0.) : P.phi[(size_t)src*DPE+k-4]);
                }
            __syncthreads();
            staged_q = q;
        }
        const bool valid = g < cnt;
        const int4 ent = valid ? sorted[first+g] : e0;
        const int c1 = SPARSE ? S.pairs[2*ent.x] : ent.x, c2 = SPARSE ? S.pairs[2*ent.x+1] : ent.y;
        double av[NC], bv[NC];
#pragma unroll
        for (int k = 0; k < NC; k++) { av[k] = P.cellv[(size_t)k*P.ncp+c1]; bv[k] = P.cellv[(size_t)k*P.ncp+c2]; }
        // the LPP lanes of a pair split the rows i of the tensor rule; each lane runs the factorised accumulation of
        // eval_distant_lds over its rows and all columns j (column data are broadcast reads)
        PairAcc<DIM, DPE> R;
        R.clear();
        if (in_lds) eval_distant_worklist<DIM, DPE, KT, wl_csum_lds(DPE) ? 2 : 0>(P, s_rule, ST, n, n4, sub, LPP, av, bv, R, lpow,
                                                                                s_rule+(size_t)tab_max_pts*ST+(size_t)g*tab_max_pts, 1);
        else
        // rules beyond the LDS copy (more than tab_max_pts points): read from global memory, S2 per point pair
#pragma unroll 1
        for (int i = sub; i < n; i += LPP) {
            double ti[ST];
            {
#pragma unroll
                for (int m = 0; m < 3; m++) ti[m] = P.bary[3*(size_t)(off+i)+m];
                ti[3] = P.w[off+i];
#pragma unroll
                for (int m = 0; m < DPE; m++) ti[4+m] = P.phi[(size_t)(off+i)*DPE+m];
            }
            double x[DIM];
#pragma unroll
            for (int d = 0; d < DIM; d++) {
                double sx = 0.;
#pragma unroll
                for (int m = 0; m < NV; m++) sx = __builtin_fma(ti[m], av[m*DIM+d], sx);
                x[d] = sx;
            }
            double r = 0., u[DPE];
#pragma unroll
            for (int b = 0; b < DPE; b++) u[b] = 0.;
#pragma unroll 2
            for (int j = 0; j < n; j++) {
                double tj[ST];
                {
#pragma unroll
                    for (int m = 0; m < 3; m++) tj[m] = P.bary[3*(size_t)(off+j)+m];
                    tj[3] = P.w[off+j];
#pragma unroll
                    for (int m = 0; m < DPE; m++) tj[4+m] = P.phi[(size_t)(off+j)*DPE+m];
                }
                double d2 = 0.;
#pragma unroll
                for (int d = 0; d < DIM; d++) {
                    double sy = 0.;
#pragma unroll
                    for (int m = 0; m < NV; m++) sy = __builtin_fma(tj[m], bv[m*DIM+d], sy);
                    const double t = x[d]-sy;
                    d2 = __builtin_fma(t, t, d2);
                }
                const double K = (ti[3]*tj[3])*kern_eval<KT>(P.k, d2, lpow);
                r += K;
                double t[DPE];
#pragma unroll
                for (int b = 0; b < DPE; b++) { t[b] = K*tj[4+b]; u[b] += t[b]; }
                int e = 0;
#pragma unroll
                for (int a = 0; a < DPE; a++)
#pragma unroll
                    for (int b = a; b < DPE; b++) { R.S2[e] = __builtin_fma(t[a], tj[4+b], R.S2[e]); e++; }
            }
            int e = 0;
#pragma unroll
            for (int a = 0; a < DPE; a++) {
                const double pa = ti[4+a];
#pragma unroll
                for (int b = 0; b < DPE; b++) R.G[a][b] = __builtin_fma(pa, u[b], R.G[a][b]);
                const double pr = pa*r;
#pragma unroll
                for (int b = a; b < DPE; b++) { R.S1[e] = __builtin_fma(pr, ti[4+b], R.S1[e]); e++; }
            }
        }
        double acc[NACC];
#pragma unroll
        for (int a = 0; a < DPE; a++)
#pragma unroll
            for (int b = 0; b < DPE; b++) acc[a*DPE+b] = R.G[a][b];
#pragma unroll
        for (int e = 0; e < ND; e++) { acc[NG+e] = R.S1[e]; acc[NG+ND+e] = R.S2[e]; }
        // row-wise reduction; lane (e mod 16) of the row keeps entry e
        double mine[NREP];
#pragma unroll
        for (int r = 0; r < NREP; r++) mine[r] = 0.;
#pragma unroll
        for (int e = 0; e < NACC; e++) {
            const double s = row16_sum(acc[e]);
            mine[e/LPP] = (sub == (e & (LPP-1))) ? s : mine[e/LPP];
        }
        if (valid && cluster) {
            const double vv = 2.*P.cvol[c1]*P.cvol[c2]*kern_scale<KT>(P.k);
            const int k = CT.wl_pair[ent.z], n1 = CT.pair_nodes[2*k], n2 = CT.pair_nodes[2*k+1];
            const int2 ds = CT.wl_ds[ent.z];
#pragma unroll
            for (int rep = 0; rep < NREP; rep++) {
                const int e = sub+LPP*rep;
                const double val = mine[rep];
                if (e < NG) {
                    const int a = e/DPE, b = e-a*DPE;
                    const int I = P.cdof[(size_t)a*P.ncp+c1], J = P.cdof[(size_t)b*P.ncp+c2];
                    if (I >= 0 && J >= 0 && in_node(CT, n1, I) && in_node(CT, n2, J)) { sparse_add(CT.S, I, J, -vv*val); sparse_add(CT.S, J, I, -vv*val); }
                } else if (e < NG+ND) { if (ds.x >= 0) atomic_add_f64(&CT.D[(size_t)ds.x*ND+(e-NG)], vv*val); }
                else if (e < NACC) { if (ds.y >= 0) atomic_add_f64(&CT.D[(size_t)ds.y*ND+(e-NG-ND)], vv*val); }
            }
        } else if (valid && SPARSE) {
            const double vv = 2.*P.cvol[c1]*P.cvol[c2]*kern_scale<KT>(P.k);
            const unsigned long long *mask = S.masks ? S.masks+4*(size_t)ent.x : nullptr;
#pragma unroll
            for (int rep = 0; rep < NREP; rep++) {
                const int e = sub+LPP*rep;
                const double val = mine[rep];
                if (e < NG) {
                    const int a = e/DPE, b = e-a*DPE;
                    sparse_add_sym(S, mask, 2*DPE, a, DPE+b, P.cdof[(size_t)a*P.ncp+c1], P.cdof[(size_t)b*P.ncp+c2], -vv*val);
                } else if (e < NACC && !S.masks && Dglob != nullptr) {
                    const bool second = e >= NG+ND;
                    atomic_add_f64(&Dglob[(size_t)(second ? c2 : c1)*ND+(e-NG-(second ? ND : 0))], vv*val);
                } else if (e < NACC) {
                    // upper triangle of a diagonal block: entry index -> (a, b), a <= b
                    const bool second = e >= NG+ND;
                    int idx = e-NG-(second ? ND : 0), a = 0;
                    while (idx >= DPE-a) { idx -= DPE-a; a++; }
                    const int b = a+idx, cc = second ? c2 : c1, sh = second ? DPE : 0;
                    sparse_add_sym(S, mask, 2*DPE, sh+a, sh+b, P.cdof[(size_t)a*P.ncp+cc], P.cdof[(size_t)b*P.ncp+cc], vv*val);
                }
            }
        }
        if (valid && !SPARSE && !cluster) {
            const double vv = 2.*P.cvol[c1]*P.cvol[c2]*kern_scale<KT>(P.k);
#pragma unroll
            for (int rep = 0; rep < NREP; rep++) {
                const int e = sub+LPP*rep;
                const double val = mine[rep];
                if (e < NG) {
                    const int a = e/DPE, b = e-a*DPE;
                    const int I = P.cdof[(size_t)a*P.ncp+c1], J = P.cdof[(size_t)b*P.ncp+c2];
                    if (I >= 0 && J >= 0) {
                        atomic_add_f64(&A[pnl_row(P, I)*ldA+pnl_col(P, J)], -vv*val);
                        if (symflush) atomic_add_f64(&A[(long long)J*ldA+I], -vv*val);
                    }
                } else if (e < NG+ND) atomic_add_f64(&Dglob[(size_t)c1*ND+(e-NG)], vv*val);
                else if (e < NACC) atomic_add_f64(&Dglob[(size_t)c2*ND+(e-NG-ND)], vv*val);
            }
        }
    }
}

// Distant pairs of the orders with at most PNL_WL_LANE_MAXPTS points, ONE PAIR PER LANE (NO:722-789): the list is sorted by
// order, so all 64 pairs of a wave's chunk run the same trip counts (no divergence) and read the rule from the wave's own
// LDS copy with broadcast reads; the factorised accumulation of eval_distant_fixed is used (x_i, row sums and u_b per i,
// S2 directly per point pair), no cross-lane reduction.  Orders with more points (few pairs, thousands of point pairs each)
// go to k_worklist_sorted, which spreads one pair over 16 lanes.
template <int DIM, int DPE, int KT, bool SPARSE>
__global__ void __launch_bounds__(PNL_NTHREADS)
k_worklist_lane(const DevProblem P, const int4 *__restrict__ sorted, const unsigned *__restrict__ offs, double *__restrict__ A,
                long long ldA, double *__restrict__ Dglob, const SparseOut S, int dbg, const ClusterTiles CT) {
#ifndef PNL_DEBUG_ABLATE
    dbg &= 8;                                           // the other bits skip work (debug builds only)
#endif
    const bool cluster = CT.npairs > 0;             // work list of the cluster tiles: entry.z indexes wl_pair / wl_ds
    constexpr int NV = DIM+1, NC = NV*DIM, ND = DPE*(DPE+1)/2, ST = 4+DPE, STP = (ST+1) & ~1;
    __shared__ unsigned s_coff[PNL_WL_BINS+1];
    __shared__ double s_rule_all[PNL_NTHREADS/64][PNL_WL_LANE_MAXPTS*STP];
    extern __shared__ double s_csum[];              // P2: [PNL_WL_LANE_MAXPTS][PNL_NTHREADS] column sums (wl_lane_lds bytes)
    __shared__ double s_pow[KT == 0 ? PNL_POW_TAB_DOUBLES : 1];      // tables of the general power (pnl_pow_tab)
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    if (KT == 0) pnl_pow_tab_fill(s_pow, P.k.ptab, tid, PNL_NTHREADS);       // published by the barrier below
    const double *__restrict__ lpow = (KT == 0 && P.k.ptab) ? s_pow : nullptr;
    if (tid == 0) {
        unsigned run = 0;
        const bool finite = SPARSE && P.k.horizon2 < 1e300;
        for (int bin = 0; bin < PNL_WL_BINS; bin++) {
            s_coff[bin] = run;
            // bins above PNL_CUT_SHIFT+1 hold the pairs of order bin - PNL_CUT_SHIFT that a finite horizon cuts
            const int q = (finite && bin >= PNL_CUT_SHIFT+2 && bin <= 2*PNL_CUT_SHIFT) ? bin-PNL_CUT_SHIFT : bin;
            const bool own = finite ? bin <= 2*PNL_CUT_SHIFT : true;
            const int n = (own && q >= 2 && q <= P.qmax && q <= PNL_MAXQ) ? P.off[q+1]-P.off[q] : 0;
            if (n > 0 && n <= PNL_WL_LANE_MAXPTS) run += (offs[bin+1]-offs[bin]+63u)/64u;
        }
        s_coff[PNL_WL_BINS] = run;
    }
    __syncthreads();
    const unsigned nchunks = s_coff[PNL_WL_BINS];
    double *s_rule = s_rule_all[wave];
    int staged_q = -1;
    const unsigned nw = gridDim.x*(PNL_NTHREADS/64);
    for (unsigned chunk = blockIdx.x*(PNL_NTHREADS/64)+wave; chunk < nchunks; chunk += nw) {
        int lo = 0, hi = PNL_WL_BINS-1;
        while (lo < hi) {
            const int mid = (lo+hi+1) >> 1;
            if (s_coff[mid] <= chunk) lo = mid; else hi = mid-1;
        }
        const int bin = __builtin_amdgcn_readfirstlane(lo);
        const bool cut = SPARSE && P.k.horizon2 < 1e300 && bin >= PNL_CUT_SHIFT+2;
        const int q = cut ? bin-PNL_CUT_SHIFT : bin;
        const unsigned first = offs[bin]+64u*(chunk-s_coff[bin]);
        const int cnt = (int)min(64u, offs[bin+1]-first);
        const int off = P.off[q], n = P.off[q+1]-off;
        const int n4 = (n+3) & ~3;                  // zero-weight copies of point 0 fill the last group of four columns (n <= 40)
        if (q != staged_q) {
            for (int t = lane; t < n4*STP; t += 64) {
                const int pt = t/STP, k = t-pt*STP, src = off+(pt < n ? pt : 0);
                s_rule[t] = k < 3 ? P.bary[3*(size_t)src+k] : (k == 3 ? (pt < n ? P.w[src] : 0.) : (k < ST ? P.phi[(size_t)src*DPE+k-4] : 0.));
            }
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
            __builtin_amdgcn_wave_barrier();
            staged_q = q;
        }
        const bool valid = lane < cnt;
        const int4 ent = sorted[first+(valid ? lane : 0)];
        const int c1 = SPARSE ? S.pairs[2*(size_t)ent.x] : ent.x, c2 = SPARSE ? S.pairs[2*(size_t)ent.x+1] : ent.y;
        double av[NC], bv[NC];
#pragma unroll
        for (int k = 0; k < NC; k++) { av[k] = P.cellv[(size_t)k*P.ncp+c1]; bv[k] = P.cellv[(size_t)k*P.ncp+c2]; }
        PairAcc<DIM, DPE> R;
        R.clear();
        if (SPARSE && cut) {
            const unsigned ne = valid ? eval_distant_cut<DIM, DPE>(P, s_rule, STP, n, av, bv, R) : 0u;
            const double tot = wave_sum((double)ne);
            if (lane == 0 && tot > 0.) atomicAdd(&P.counters[2], (unsigned long long)tot);
        } else
        {
            // one pair per lane: this kernel is bound by its atomics (18 + 12 per pair, P1) and lives on waves in flight; the
            // blocked evaluator costs registers (P1: 4 -> 2 waves per SIMD, 3.9 -> 5.0 ms at 98,304 cells; P2 with the rsqrt kernel
            // 2 -> 1, 3.3 -> 4.1 ms) and pays only where the kernel value is dear and the occupancy is one wave already
            // (P2, general power: 4.6 -> 3.4 ms at 49,152 cells)
            if constexpr (wl_lane_blocked(DPE, KT))
                eval_distant_worklist<DIM, DPE, KT, 1>(P, s_rule, STP, n, n4, 0, 1, av, bv, R, lpow, s_csum+tid, PNL_NTHREADS);
            else kern_dispatch<KT>(P.k, lpow, [&](auto ktag) { eval_distant_lds<DIM, DPE, decltype(ktag)::value>(P, s_rule, STP, n, av, bv, R, lpow); });
        }
        // without masks (getSparse) the diagonal blocks go through the per-cell buffer like in the dense path and are
        // scattered once per cell (k_scatter_diag_sparse): 9 instead of 15 pattern searches per pair.  The sorted list keeps
        // the producer's runs (k_fh_pairs lays 64 consecutive cells c1 against one c2), so a wave holds a few distinct
        // cells on one side: their blocks are summed over the wave first, one atomic per distinct cell instead of one per lane
        const bool dbuf = SPARSE && !cluster && !S.masks && Dglob != nullptr;
        if (dbuf) {
            const double vz = valid ? 2.*P.cvol[c1]*P.cvol[c2]*kern_scale<KT>(P.k) : 0.;
#pragma unroll
            for (int side = 0; side < 2; side++) {
                const int cc = side ? c2 : c1;
                unsigned long long todo = __ballot(valid);
#pragma unroll 1
                for (int round = 0; round < 4 && todo; round++) {
                    const int leader = __ffsll((long long)todo)-1;
                    const int cL = __builtin_amdgcn_readlane(cc, leader);
                    const unsigned long long same = __ballot(valid && cc == cL) & todo;
                    const bool mine = (same >> lane) & 1ull;
#pragma unroll
                    for (int e = 0; e < ND; e++) {
                        const double s = wave_sum(mine ? vz*(side ? R.S2[e] : R.S1[e]) : 0.);
                        if (lane == leader) atomic_add_f64(&Dglob[(size_t)cL*ND+e], s);
                    }
                    todo &= ~same;
                }
                if ((todo >> lane) & 1ull) {
#pragma unroll
                    for (int e = 0; e < ND; e++) atomic_add_f64(&Dglob[(size_t)cc*ND+e], vz*(side ? R.S2[e] : R.S1[e]));
                }
            }
        }
        if (!valid) continue;
        const double vv = 2.*P.cvol[c1]*P.cvol[c2]*kern_scale<KT>(P.k);
        int ld1[DPE], ld2[DPE];
#pragma unroll
        for (int a = 0; a < DPE; a++) { ld1[a] = P.cdof[(size_t)a*P.ncp+c1]; ld2[a] = P.cdof[(size_t)a*P.ncp+c2]; }
        if (cluster) {
            const int k = CT.wl_pair[ent.z], n1 = CT.pair_nodes[2*k], n2 = CT.pair_nodes[2*k+1];
            const int2 ds = CT.wl_ds[ent.z];
            bool in1[DPE], in2[DPE];
#pragma unroll
            for (int a = 0; a < DPE; a++) { in1[a] = ld1[a] >= 0 && in_node(CT, n1, ld1[a]); in2[a] = ld2[a] >= 0 && in_node(CT, n2, ld2[a]); }
#pragma unroll
            for (int a = 0; a < DPE; a++)
#pragma unroll
                for (int b = 0; b < DPE; b++)
                    if (in1[a] && in2[b]) { sparse_add(CT.S, ld1[a], ld2[b], -vv*R.G[a][b]); sparse_add(CT.S, ld2[b], ld1[a], -vv*R.G[a][b]); }
#pragma unroll
            for (int e = 0; e < ND; e++) {
                if (ds.x >= 0) atomic_add_f64(&CT.D[(size_t)ds.x*ND+e], vv*R.S1[e]);
                if (ds.y >= 0) atomic_add_f64(&CT.D[(size_t)ds.y*ND+e], vv*R.S2[e]);
            }
        } else if (SPARSE) {
            const unsigned long long *mask = S.masks ? S.masks+4*(size_t)ent.x : nullptr;
            int e = 0;
#pragma unroll
            for (int a = 0; a < DPE; a++) {
#pragma unroll
                for (int b = 0; b < DPE; b++) sparse_add_sym(S, mask, 2*DPE, a, DPE+b, ld1[a], ld2[b], -vv*R.G[a][b]);
#pragma unroll
                for (int b = a; b < DPE; b++) {
                    if (!dbuf) {
                        sparse_add_sym(S, mask, 2*DPE, a, b, ld1[a], ld1[b], vv*R.S1[e]);
                        sparse_add_sym(S, mask, 2*DPE, DPE+a, DPE+b, ld2[a], ld2[b], vv*R.S2[e]);
                    }
                    e++;
                }
            }
        } else {
#pragma unroll
            for (int a = 0; a < DPE; a++)
#pragma unroll
                for (int b = 0; b < DPE; b++)
                    if (ld1[a] >= 0 && ld2[b] >= 0 && !(dbg & 1)) {
                        atomic_add_f64(&A[pnl_row(P, ld1[a])*ldA+pnl_col(P, ld2[b])], -vv*R.G[a][b]);
                        if (dbg & 8) atomic_add_f64(&A[(long long)ld2[b]*ldA+ld1[a]], -vv*R.G[a][b]);      // symmetric flush
                    }
            if (!(dbg & 2))
#pragma unroll
            for (int e = 0; e < ND; e++) {
                atomic_add_f64(&Dglob[(size_t)c1*ND+e], vv*R.S1[e]);
                atomic_add_f64(&Dglob[(size_t)c2*ND+e], vv*R.S2[e]);
            }
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
// STAGE: the rule tables (nodes, weights, PSI) are copied into LDS once per workgroup and shared by all of its
// waves over a grid-stride loop of pairs; otherwise (tables too large for LDS) they are read from L2.
#define PNL_SING_THREADS 512
template <int DIM, int DPE, int SLOT, int KT, bool STAGE, bool SPARSE>
__global__ void __launch_bounds__(PNL_SING_THREADS)
k_singular_pairs(const DevProblem P, const int2 *__restrict__ pairs, int npairs_in, double *__restrict__ A, long long ldA,
                 int cell_begin, int cell_end, const SparseOut S, const int4 *__restrict__ sorted,
                 const unsigned *__restrict__ offs, const ClusterTiles CT) {
    const bool cluster = CT.npairs > 0;          // touching element pairs of cluster pairs: pairs[wid] with cluster pair CT.sing_pair[wid]
    constexpr int NV = DIM+1;
    constexpr int DPV = elem_dpv(DPE), DPED = elem_dped(DIM, DPE);
    constexpr int COMMON = SLOT+1;
    constexpr int ROWS = (COMMON == NV) ? DPE : (COMMON == 1 ? 2*DPE-DPV : 2*DPE-2*DPV-DPED);
    constexpr int NE = ROWS*(ROWS+1)/2;
    extern __shared__ double s_tab[];
    const int lane = threadIdx.x & 63;
    const int M = P.sM[SLOT];
    const double *__restrict__ nodes = P.sNodes[SLOT];
    const double *__restrict__ w = P.sW[SLOT];
    const double *__restrict__ psi = P.sPsi[SLOT];
    if (STAGE) {
        for (int t = threadIdx.x; t < 2*NV*M; t += PNL_SING_THREADS) s_tab[t] = P.sNodes[SLOT][t];
        for (int t = threadIdx.x; t < M; t += PNL_SING_THREADS) s_tab[2*NV*M+t] = P.sW[SLOT][t];
        for (int t = threadIdx.x; t < ROWS*M; t += PNL_SING_THREADS) s_tab[(2*NV+1)*M+t] = P.sPsi[SLOT][t];
        __syncthreads();
        nodes = s_tab; w = s_tab+2*NV*M; psi = s_tab+(2*NV+1)*M;
    }
    const int nwaves = gridDim.x*(PNL_SING_THREADS/64);
    unsigned long long done = 0;        // statistics are accumulated per wave: one atomic per wave, not per pair
    // sparse path: the pairs are the segment of the sorted list with key 121 + SLOT
    const int seg0 = SPARSE ? (int)offs[121+SLOT] : 0;
    const int npairs = SPARSE ? (int)(offs[121+SLOT+1]-offs[121+SLOT]) : npairs_in;
    for (int wid = (blockIdx.x*PNL_SING_THREADS+threadIdx.x) >> 6; wid < npairs; wid += nwaves) {
    int pidx = 0;
    int2 pr;
    if (SPARSE) {
        pidx = __builtin_amdgcn_readfirstlane(sorted[seg0+wid].x);
        pr = make_int2(S.pairs[2*pidx], S.pairs[2*pidx+1]);
        // second orientation of a non-symmetric order table (swapCells, NA:1418): the singular rules are not symmetric under the swap
        if (P.orient) pr = make_int2(pr.y, pr.x);
    } else pr = pairs[wid];
    const int c1 = __builtin_amdgcn_readfirstlane(pr.x), c2 = __builtin_amdgcn_readfirstlane(pr.y);
    // the cell range of the MPI-style split applies to the smaller cell number (cellNo1 of the reference loop; the lists of
    // the second orientation of a non-symmetric order hold the pairs swapped)
    if (!SPARSE && (min(c1, c2) < cell_begin || min(c1, c2) >= cell_end)) continue;
    // NA:138-150
    int ld[2*DPE];
    bool any = false;
#pragma unroll
    for (int k = 0; k < DPE; k++) {
        ld[k] = P.cdof[(size_t)k*P.ncp+c1];
        ld[DPE+k] = P.cdof[(size_t)k*P.ncp+c2];
        any = any || ld[k] >= 0 || ld[DPE+k] >= 0;
    }
    if (!any) continue;
    // vertex permutations, shared vertices first (NO:311-346)
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
    const double vol = P.sFac*P.cvol[c1]*P.cvol[c2]*(c1 == c2 ? P.idfac : 2.)*kern_scale<KT>(P.k);
    // reduce; lane (e mod 64) scatters entry e (NE <= 66 < 128: at most two entries per lane)
    double mine[2] = {0., 0.};
    int myI[2] = {0, 0}, myJ[2] = {0, 0};
    {
        int e = 0;
#pragma unroll
        for (int I = 0; I < ROWS; I++)
#pragma unroll
            for (int J = I; J < ROWS; J++) {
                const double s = wave_sum(acc[e]);
                if (lane == (e & 63)) { mine[e >> 6] = s; myI[e >> 6] = I; myJ[e >> 6] = J; }
                e++;
            }
    }
#pragma unroll
    for (int rep = 0; rep < (NE+63)/64; rep++) {
        if (lane+64*rep >= NE) continue;
        int gi = -1, gj = -1;
#pragma unroll
        for (int k = 0; k < 2*DPE; k++) {
            // perm[] and ld[] with runtime indices -> selects
            const int pk = perm[k];
            int g = -1;
#pragma unroll
            for (int m = 0; m < 2*DPE; m++) g = (pk == m) ? ld[m] : g;
            gi = (myI[rep] == k) ? g : gi;
            gj = (myJ[rep] == k) ? g : gj;
        }
        const double v = mine[rep]*vol;
        if (cluster) {
            // the DoF pair {gi, gj} gets the entry if it belongs to the cluster pair (what is left of the masks)
            if (pair_has(CT, CT.sing_pair[wid], gi, gj)) {
                if (gi == gj) sparse_add(CT.S, gi, gi, v);
                else { sparse_add(CT.S, gi, gj, v); sparse_add(CT.S, gj, gi, v); }
            }
        } else if (SPARSE) {
            // local indices of the merged rows (FL2:874-884): p = perm[I], q = perm[J], entry (min, max)
            int pi = 0, pj = 0;
#pragma unroll
            for (int k = 0; k < 2*DPE; k++) { pi = (myI[rep] == k) ? perm[k] : pi; pj = (myJ[rep] == k) ? perm[k] : pj; }
            // the masks are indexed in the local numbering of the LISTED pair (c1 <= c2): undo the swap of the second orientation
            if (P.orient) { pi = pi < DPE ? pi+DPE : pi-DPE; pj = pj < DPE ? pj+DPE : pj-DPE; }
            const int lo = min(pi, pj), hi = max(pi, pj);
            const int glo = (lo == pi) ? gi : gj, ghi = (lo == pi) ? gj : gi;
            sparse_add_sym(S, S.masks ? S.masks+4*(size_t)pidx : nullptr, 2*DPE, lo, hi, glo, ghi, v);
        } else if (gi >= 0 && gj >= 0) {
            if (P.onesided) {
                // row slab of a rank: every symmetric contribution once, at (min, max); rows that the slab does not hold are
                // counted (counter 7) and fail the assembly
                const int lo = min(gi, gj), hi = max(gi, gj);
                const long long row = pnl_row(P, lo);
                const int col = pnl_col(P, hi);
                if (row >= 0 && col >= 0) atomic_add_f64(&A[row*ldA+col], v);
                else atomicAdd(&P.counters[7], 1ull);
            } else if (myI[rep] == myJ[rep]) atomic_add_f64(&A[(long long)gi*ldA+gi], v);
            else {
                atomic_add_f64(&A[(long long)gi*ldA+gj], v);
                atomic_add_f64(&A[(long long)gj*ldA+gi], v);
            }
        }
    }
    done++;
    }   // grid-stride loop over pairs
    if (lane == 0 && done) {
        atomicAdd(&P.counters[1], done);
        atomicAdd(&P.counters[2], done*(unsigned long long)M);
        atomicAdd(&P.counters[128+SLOT], done);
    }
}

// ---------------------------------------------------------------------------------------------
// Omega x Omega^c, distant part: one thread per cell, loops over a small chunk of boundary facets
// (NA:1430-1448 loop, NO:1022-1108 eval_distant_boundary, order FL2:1226-1243 / FL1:646-660).
// The reference multiplies the boundary kernel Gamma_b(|x-y|) by n.(y-x)/|y-x|; here the 1/|y-x| is folded
// into the kernel exponent (P.bkn = Gamma_b with exponent - 1/2), so one rsqrt-type evaluation does both.
template <int DIM, int DPE, int KT>
__global__ void __launch_bounds__(PNL_NTHREADS)
k_boundary_distant(const DevProblem P, double *__restrict__ Dglob, int cell_begin, int cell_end, int facets_per_chunk,
                   const DevKernel *__restrict__ bkcls, const DevFormula *__restrict__ bfcls, int defer_evals,
                   int *__restrict__ dcells, int *__restrict__ dfacets, unsigned *__restrict__ dslots, unsigned *__restrict__ dcount,
                   unsigned dcap, int *__restrict__ dcls) {
    constexpr int NV = DIM+1, NC = NV*DIM, NF = DIM, ND = DPE*(DPE+1)/2;
    const int c = cell_begin+blockIdx.x*PNL_NTHREADS+threadIdx.x;
    bool active = c < cell_end;
    const int cc = active ? c : cell_begin;
    double av[NC], cen[DIM];
    int vid[NV];
#pragma unroll
    for (int k = 0; k < NC; k++) av[k] = P.cellv[(size_t)k*P.ncp+cc];
#pragma unroll
    for (int d = 0; d < DIM; d++) cen[d] = P.ccen[(size_t)d*P.ncp+cc];
#pragma unroll
    for (int k = 0; k < NV; k++) vid[k] = P.cvid[(size_t)k*P.ncp+cc];
    active = active && vid[0] >= 0;                      // zero-volume padding cells inside the mesh meet no facet
    const double h1 = P.ch[cc], vol1 = P.cvol[cc];
    const double Ld1 = fabs(log(h1/P.H0));
    const float lh1 = (float)log(h1), L1 = (float)Ld1;
    double D[ND];
#pragma unroll
    for (int e = 0; e < ND; e++) D[e] = 0.;
    unsigned long long npairs = 0, nevals = 0;
    int overflow = 0;
    const int f0 = blockIdx.y*facets_per_chunk;
    const int f1 = min(P.nb, f0+facets_per_chunk);
    const int lab1 = P.cur_class >= 0 ? P.clabel[cc] : 0;
    for (int f = f0; f < f1; f++) {
        // variable order: with class tables (bkcls, bfcls) ONE launch integrates every (cell, facet) with the kernel and order
        // formula of the pair's class; without them this launch handles class P.cur_class and skips the other pairs
        DevKernel bkn = P.bkn;
        DevFormula bqo = P.bqo;
        int kc = -1;
        if (P.cur_class >= 0) {
            kc = P.cls_of[lab1*P.nlab+P.blabel[f]];
            if (bkcls) { bkn = bkcls[kc]; bqo = bfcls[kc]; }
            else if (kc != P.cur_class) continue;
        }
        // facet data: wave-uniform, precomputed once per upload (centre, unit normal, length, logs)
        double fv[NF*DIM], fc[DIM], nrm[DIM];
        int fvid[NF];
#pragma unroll
        for (int k = 0; k < NF; k++) fvid[k] = P.bvid[(size_t)k*P.nb+f];
#pragma unroll
        for (int k = 0; k < NF*DIM; k++) fv[k] = P.bv[(size_t)k*P.nb+f];
#pragma unroll
        for (int d = 0; d < DIM; d++) { fc[d] = P.bgeo[(size_t)d*P.nb+f]; nrm[d] = P.bgeo[(size_t)(DIM+d)*P.nb+f]; }
        const double vol2 = P.bgeo[(size_t)(2*DIM)*P.nb+f], Ld2 = P.bgeo[(size_t)(2*DIM+1)*P.nb+f];
        const float lh2 = (float)P.bgeo[(size_t)(2*DIM+2)*P.nb+f], L2 = (float)Ld2;
        bool shared = false;
#pragma unroll
        for (int k = 0; k < NV; k++)
#pragma unroll
            for (int m = 0; m < NF; m++) shared = shared || (vid[k] == fvid[m]);
        if (!active || shared) continue;
        double dc2 = 0.;
#pragma unroll
        for (int d = 0; d < DIM; d++) dc2 += (cen[d]-fc[d])*(cen[d]-fc[d]);
        const int q = quad_order_fast(bqo, h1, vol2, lh1, lh2, L1, L2, Ld1, Ld2, dc2);
        if (q > P.qmax || q > PNL_MAXQ) { overflow++; continue; }
        const int off = P.off[q], n = P.off[q+1]-off;
        const int foff = P.foff[q], nf = P.foff[q+1]-foff;
        // the few pairs close to the boundary need rules with hundreds of point pairs: one lane would keep its wave (and the
        // launch) waiting for it.  They go to a list of (cell, facet) items that k_boundary_items integrates one per wave,
        // lanes over the point pairs, into the same per-cell blocks.
        if (dcells && n*nf > defer_evals) {
            const unsigned idx = atomicAdd(dcount, 1u);
            if (idx < dcap) {
                dcells[idx] = c;
                dslots[idx] = (unsigned)c;
                if (bkcls) dcls[idx] = kc;
#pragma unroll
                for (int k = 0; k < NF; k++) dfacets[(size_t)idx*NF+k] = fvid[k];
                continue;
            }
        }
        const double *__restrict__ bary = P.bary+3*(size_t)off;
        const double *__restrict__ w = P.w+off;
        const double *__restrict__ phi = P.phi+(size_t)off*DPE;
        const double *__restrict__ fb = P.fbary+2*(size_t)foff;
        const double *__restrict__ fw = P.fw+foff;
        npairs++;
        nevals += (unsigned long long)n*nf;
        const double vol = vol1*vol2*kern_scale<KT>(bkn);
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
                if (DIM != 2) nw = 1.;
                r = __builtin_fma(fw[m]*nw, kern_eval<KT, true>(bkn, d2), r);
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
        if (overflow) atomicAdd(&P.counters[5], (unsigned long long)overflow);
    }
    {
        // statistics: one atomic per wave
        const double sp = wave_sum((double)npairs), se = wave_sum((double)nevals);
        if ((threadIdx.x & 63) == 0 && sp > 0.) {
            atomicAdd(&P.counters[3], (unsigned long long)sp);
            atomicAdd(&P.counters[4], (unsigned long long)se);
        }
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
    const int c1 = __builtin_amdgcn_readfirstlane(pairs[wid].x), f = __builtin_amdgcn_readfirstlane(pairs[wid].y);
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
        if (DIM != 2) nw = 1.;
        const double t = w[m]*nw*kern_eval<KT, true>(P.bkn, d2);
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
    const double vol = ((DIM == 2) ? P.bFac*P.cvol[c1]*vol2 : P.bFac*P.cvol[c1])*kern_scale<KT>(P.bkn);
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

// ---- masked pair assembly (assembleClusters, NA:1663-1964): statistics of the sorted pair list ---------------------
// hist[q] pairs of order q: numAssembledCellPairs, kernel evaluations n(q)^2 each and the order histogram (the touching
// pairs in bins 121.. are counted by k_singular_pairs itself)
__global__ void k_mp_stats(const DevProblem P, const unsigned *__restrict__ hist) {
    const int q = threadIdx.x;
    if (q < 2 || q > P.qmax || q > PNL_MAXQ) return;
    const unsigned long long c = hist[q];
    // pairs cut by a finite horizon sit in bin q + PNL_CUT_SHIFT; their kernel evaluations are counted by the kernel
    const unsigned long long ccut = (P.k.horizon2 < 1e300 && q+PNL_CUT_SHIFT < PNL_WL_BINS && q <= PNL_CUT_SHIFT) ? hist[q+PNL_CUT_SHIFT] : 0ull;
    if (!c && !ccut) return;
    const unsigned long long n = (unsigned long long)(P.off[q+1]-P.off[q]);
    atomicAdd(&P.counters[8+q], c+ccut);
    atomicAdd(&P.counters[1], c+ccut);
    if (c) atomicAdd(&P.counters[2], c*n*n);
}

// Gauss-theorem boundary term over explicit (cell, facet) items with entry masks: the cluster-local term
// NA:1842-1889 (facets = boundary of cellsUnion) and the global Omega x Omega^c term with fac = +-1 (NA:1896-1913,
// 1945-1964).  Panel by shared vertices (NO:280-378), distant evaluation NO:1022-1108, touching FL2:1324-1407 /
// FL1:726-785, scatter NA:534-546 addToMatrixElemSymMasked.  One wave per item, lanes over the quadrature points.
template <int DIM, int DPE, int KT>
__global__ void __launch_bounds__(PNL_NTHREADS)
k_boundary_items(const DevProblem P, const double *__restrict__ verts, const int *__restrict__ cells,
                 const int *__restrict__ facets, const unsigned *__restrict__ masks, int ni, double fac, const SparseOut S,
                 double *__restrict__ Dout, const unsigned *__restrict__ ni_dev, const int *__restrict__ item_cls,
                 const DevKernel *__restrict__ bkcls, const DevFormula *__restrict__ bfcls) {
    // item count produced on the device (k_boundary_distant's deferred pairs): at most the capacity ni;
    // item_cls: kernel class per item (variable order), parameters from the class tables
    if (ni_dev) ni = (int)min(*ni_dev, (unsigned)ni);
    constexpr int NV = DIM+1, NF = DIM, ND = DPE*(DPE+1)/2;
    const int lane = threadIdx.x & 63;
    const int nwaves = gridDim.x*(PNL_NTHREADS/64);
    unsigned long long npairs = 0, nevals = 0;
    for (int wid = (blockIdx.x*PNL_NTHREADS+threadIdx.x) >> 6; wid < ni; wid += nwaves) {
        const int c1 = __builtin_amdgcn_readfirstlane(cells[wid]);
        const unsigned mask = (unsigned)__builtin_amdgcn_readfirstlane((int)masks[wid]);
        DevKernel bkn = P.bkn;
        DevFormula bqo = P.bqo;
        if (item_cls) {
            const int kc = __builtin_amdgcn_readfirstlane(item_cls[wid]);
            bkn = bkcls[kc]; bqo = bfcls[kc];
        }
        int fvid[NF];
        double fv[NF][DIM], cv[NV][DIM];
#pragma unroll
        for (int k = 0; k < NF; k++) {
            fvid[k] = __builtin_amdgcn_readfirstlane(facets[(size_t)wid*NF+k]);
#pragma unroll
            for (int d = 0; d < DIM; d++) fv[k][d] = verts[(size_t)fvid[k]*DIM+d];
        }
#pragma unroll
        for (int k = 0; k < NV; k++)
#pragma unroll
            for (int d = 0; d < DIM; d++) cv[k][d] = P.cellv[(size_t)(k*DIM+d)*P.ncp+c1];
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
            const int q = quad_order(bqo, P.H0, P.ch[c1], vol2, sqrt(dc2));
            if (q > P.qmax || q > PNL_MAXQ || P.off[q+1] == P.off[q] || P.foff[q+1] == P.foff[q]) {
                if (lane == 0) atomicAdd(&P.counters[5], 1ull);
                continue;
            }
            const int off = P.off[q], n = P.off[q+1]-off, foff = P.foff[q], nf = P.foff[q+1]-foff;
            for (int k = lane; k < n*nf; k += 64) {
                const int i = k/nf, m = k-i*nf;
                double d2 = 0., nw = 0.;
#pragma unroll
                for (int d = 0; d < DIM; d++) {
                    double x = 0., y = 0.;
#pragma unroll
                    for (int t = 0; t < NV; t++) x = __builtin_fma(P.bary[3*(size_t)(off+i)+t], cv[t][d], x);
#pragma unroll
                    for (int t = 0; t < NF; t++) y = __builtin_fma(P.fbary[2*(size_t)(foff+m)+t], fv[t][d], y);
                    const double wv = y-x;
                    d2 = __builtin_fma(wv, wv, d2);
                    if (DIM == 2) nw = __builtin_fma(nrm[d], wv, nw);
                }
                if (DIM != 2) nw = 1.;
                const double t = (P.w[off+i]*P.fw[foff+m])*nw*kern_eval<KT, true>(bkn, d2);
                int e = 0;
#pragma unroll
                for (int a = 0; a < DPE; a++) {
                    const double ta = t*P.phi[(size_t)(off+i)*DPE+a];
#pragma unroll
                    for (int b = a; b < DPE; b++) { acc[e] = __builtin_fma(ta, P.phi[(size_t)(off+i)*DPE+b], acc[e]); e++; }
                }
            }
            vol = P.cvol[c1]*vol2*kern_scale<KT>(bkn);
            nevals += (unsigned long long)n*nf;
        } else {
            int i = 0;
            for (int k = common; k < NV; k++) { while (mask1 & (1 << i)) i++; perm1[k] = i; mask1 += (1 << i); }
            i = 0;
            for (int k = common; k < NF; k++) { while (mask2 & (1 << i)) i++; perm2[k] = i; mask2 += (1 << i); }
            const int *t1 = P.perm_table+perm_rank(perm1, NV)*DPE;
            for (int k = 0; k < DPE; k++) perm[k] = t1[k];
            double s1[NV][DIM], s2[NF][DIM];
#pragma unroll
            for (int k = 0; k < NV; k++)
#pragma unroll
                for (int d = 0; d < DIM; d++) {
                    double a = 0.;
#pragma unroll
                    for (int m = 0; m < NV; m++) a = (perm1[k] == m) ? cv[m][d] : a;
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
            const int slot = common-1;
            const int M = P.bM[slot];
            const double *__restrict__ nodes = P.bNodes[slot];
            const double *__restrict__ w = P.bW[slot];
            const double *__restrict__ PHI = P.bPhi[slot];
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
                if (DIM != 2) nw = 1.;
                const double t = w[m]*nw*kern_eval<KT, true>(bkn, d2);
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
            vol = ((DIM == 2) ? P.bFac*P.cvol[c1]*vol2 : P.bFac*P.cvol[c1])*kern_scale<KT>(bkn);
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
            if (Dout) atomic_add_f64(&Dout[(size_t)mask*ND+kk], fac*vol*mine);      // tiled near field: masks[] = slot in D
            else if ((mask >> kk) & 1u) {
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

// ---- near-field matvec: CSR_LinearOperator / SSS_LinearOperator matvec (CSR_LinearOperator_{SCALAR}.pxi:259-284,
// SSS_LinearOperator_{SCALAR}.pxi:146-176).  One wave per row; SSS adds the mirrored entries with atomics.
__global__ void __launch_bounds__(PNL_NTHREADS)
k_spmv(const int *__restrict__ indptr, const int *__restrict__ indices, const double *__restrict__ data,
       const double *__restrict__ diag, int n, const double *__restrict__ x, double *__restrict__ y) {
    const int lane = threadIdx.x & 63;
    const int row = (blockIdx.x*PNL_NTHREADS+threadIdx.x) >> 6;
    if (row >= n) return;
    const int b = indptr[row], e = indptr[row+1];
    const double xi = x[row];
    double s = 0.;
    int t = b+lane;
    if (!diag) {
        // four independent (index, value, x) load chains per lane in flight: rows of the near field hold ~1000 entries
        double s1 = 0., s2 = 0., s3 = 0.;
        for (; t+192 < e; t += 256) {
            const int J0 = indices[t], J1 = indices[t+64], J2 = indices[t+128], J3 = indices[t+192];
            const double a0 = data[t], a1 = data[t+64], a2 = data[t+128], a3 = data[t+192];
            s = __builtin_fma(a0, x[J0], s); s1 = __builtin_fma(a1, x[J1], s1);
            s2 = __builtin_fma(a2, x[J2], s2); s3 = __builtin_fma(a3, x[J3], s3);
        }
        s += s1+(s2+s3);
    }
    for (; t < e; t += 64) {
        const int J = indices[t];
        const double a = data[t];
        s = __builtin_fma(a, x[J], s);
        if (diag) atomic_add_f64(&y[J], a*xi);
    }
    s = wave_sum(s);
    if (lane == 0) {
        if (diag) atomic_add_f64(&y[row], __builtin_fma(diag[row], xi, s));
        else y[row] = s;
    }
}

// ---- tiled near field: scatter of the per-(cluster pair, cell) diagonal blocks -----------------------------------------
// D[d][.] of cell X = d_cell[d] in cluster pair d_pair[d]: every ordered local pair (p, q) goes to (I, J) = (ld[p], ld[q]) if
// the DoF pair belongs to the cluster pair (mirror images are their own ordered pairs; SSS keeps I >= J)
template <int DPE>
__global__ void __launch_bounds__(PNL_NTHREADS)
k_cluster_scatter_diag(const DevProblem P, const ClusterTiles CT, const int *__restrict__ d_cell, const int *__restrict__ d_pair, int nd) {
    constexpr int ND = DPE*(DPE+1)/2;
    const long long t = (long long)blockIdx.x*PNL_NTHREADS+threadIdx.x;
    const int d = (int)(t/(DPE*DPE));
    if (d >= nd) return;
    const int ab = (int)(t-(long long)d*DPE*DPE), a = ab/DPE, b = ab-a*DPE;
    const int lo = min(a, b), hi = max(a, b);
    const double v = CT.D[(size_t)d*ND+DPE*lo-(lo*(lo+1) >> 1)+hi];
    if (v == 0.) return;
    const int c = d_cell[d];
    const int I = P.cdof[(size_t)a*P.ncp+c], J = P.cdof[(size_t)b*P.ncp+c];
    if (pair_has(CT, d_pair[d], I, J)) sparse_add(CT.S, I, J, v);
}

// Cluster-local Gauss-theorem term (NA:1842-1889), distant (cell, facet) pairs: thread per diagonal-block slot d (cell
// d_cell[d] of cluster pair d_pair[d]), loop over a chunk of the facets of the boundary of the pair's cellsUnion
// (pair_foff, facet vertex ids fvid and precomputed geometry fgeo like DevProblem::bgeo).  Same evaluation as
// k_boundary_distant (NO:1022-1108, order FL2:1226-1243 / FL1:646-660).
template <int DIM, int DPE, int KT>
__global__ void __launch_bounds__(PNL_NTHREADS)
k_cluster_boundary(const DevProblem P, const double *__restrict__ verts, const int *__restrict__ d_cell, const int *__restrict__ d_pair,
                   int nd, const int *__restrict__ pair_foff, const int *__restrict__ fvid, const double *__restrict__ fgeo, int nftot,
                   double *__restrict__ D, int facets_per_chunk, int defer_evals, int *__restrict__ dcells, int *__restrict__ dfacets,
                   unsigned *__restrict__ dslots, unsigned *__restrict__ dcount, unsigned dcap) {
    constexpr int NV = DIM+1, NC = NV*DIM, NF = DIM, ND = DPE*(DPE+1)/2;
    const int d = blockIdx.x*PNL_NTHREADS+threadIdx.x;
    if (d >= nd) return;
    const int cc = d_cell[d], k = d_pair[d];
    const int f0 = pair_foff[k]+blockIdx.y*facets_per_chunk, f1 = min(pair_foff[k+1], f0+facets_per_chunk);
    if (f0 >= f1) return;
    double av[NC], cen[DIM];
    int vid[NV];
#pragma unroll
    for (int m = 0; m < NC; m++) av[m] = P.cellv[(size_t)m*P.ncp+cc];
#pragma unroll
    for (int dd = 0; dd < DIM; dd++) cen[dd] = P.ccen[(size_t)dd*P.ncp+cc];
#pragma unroll
    for (int m = 0; m < NV; m++) vid[m] = P.cvid[(size_t)m*P.ncp+cc];
    const double h1 = P.ch[cc], vol1 = P.cvol[cc];
    const double Ld1 = P.clog[(size_t)P.ncp+cc];
    const float lh1 = (float)P.clog[cc], L1 = (float)Ld1;
    double Dl[ND];
#pragma unroll
    for (int e = 0; e < ND; e++) Dl[e] = 0.;
    unsigned long long npairs = 0, nevals = 0;
    int overflow = 0;
    for (int f = f0; f < f1; f++) {
        double fv[NF*DIM], fc[DIM], nrm[DIM];
        int fvd[NF];
        bool shared = false;
#pragma unroll
        for (int m = 0; m < NF; m++) {
            fvd[m] = fvid[(size_t)f*NF+m];
#pragma unroll
            for (int dd = 0; dd < DIM; dd++) fv[m*DIM+dd] = verts[(size_t)fvd[m]*DIM+dd];
#pragma unroll
            for (int kk = 0; kk < NV; kk++) shared = shared || (vid[kk] == fvd[m]);
        }
        if (shared) continue;                       // touching (cell, facet) pairs come as explicit items
#pragma unroll
        for (int dd = 0; dd < DIM; dd++) { fc[dd] = fgeo[(size_t)dd*nftot+f]; nrm[dd] = fgeo[(size_t)(DIM+dd)*nftot+f]; }
        const double vol2 = fgeo[(size_t)(2*DIM)*nftot+f], Ld2 = fgeo[(size_t)(2*DIM+1)*nftot+f];
        const float lh2 = (float)fgeo[(size_t)(2*DIM+2)*nftot+f], L2 = (float)Ld2;
        double dc2 = 0.;
#pragma unroll
        for (int dd = 0; dd < DIM; dd++) dc2 += (cen[dd]-fc[dd])*(cen[dd]-fc[dd]);
        const int q = quad_order_fast(P.bqo, h1, vol2, lh1, lh2, L1, L2, Ld1, Ld2, dc2);
        if (q > P.qmax || q > PNL_MAXQ) { overflow++; continue; }
        const int off = P.off[q], n = P.off[q+1]-off;
        const int foff = P.foff[q], nf = P.foff[q+1]-foff;
        // the cells of a near-field cluster pair are close to its facets: pairs with many point pairs go one per wave through
        // k_boundary_items (as in k_boundary_distant) instead of keeping the other 63 lanes of this wave waiting
        if (dcells && n*nf > defer_evals) {
            const unsigned idx = atomicAdd(dcount, 1u);
            if (idx < dcap) {
                dcells[idx] = cc;
                dslots[idx] = (unsigned)d;
#pragma unroll
                for (int m = 0; m < NF; m++) dfacets[(size_t)idx*NF+m] = fvd[m];
                continue;
            }
        }
        const double *__restrict__ bary = P.bary+3*(size_t)off;
        const double *__restrict__ w = P.w+off;
        const double *__restrict__ phi = P.phi+(size_t)off*DPE;
        const double *__restrict__ fb = P.fbary+2*(size_t)foff;
        const double *__restrict__ fw = P.fw+foff;
        npairs++;
        nevals += (unsigned long long)n*nf;
        const double vol = vol1*vol2*kern_scale<KT>(P.bkn);
        for (int i = 0; i < n; i++) {
            double x[DIM];
#pragma unroll
            for (int dd = 0; dd < DIM; dd++) {
                double s = 0.;
#pragma unroll
                for (int m = 0; m < NV; m++) s = __builtin_fma(bary[3*i+m], av[m*DIM+dd], s);
                x[dd] = s;
            }
            double r = 0.;
            for (int m = 0; m < nf; m++) {
                double d2 = 0., nw = 0.;
#pragma unroll
                for (int dd = 0; dd < DIM; dd++) {
                    double y = 0.;
#pragma unroll
                    for (int t = 0; t < NF; t++) y = __builtin_fma(fb[2*m+t], fv[t*DIM+dd], y);
                    const double wv = y-x[dd];
                    d2 = __builtin_fma(wv, wv, d2);
                    if (DIM == 2) nw = __builtin_fma(nrm[dd], wv, nw);
                }
                if (DIM != 2) nw = 1.;
                r = __builtin_fma(fw[m]*nw, kern_eval<KT, true>(P.bkn, d2), r);
            }
            r *= w[i]*vol;
            int e = 0;
#pragma unroll
            for (int a = 0; a < DPE; a++) {
                const double pa = phi[i*DPE+a]*r;
#pragma unroll
                for (int b = a; b < DPE; b++) { Dl[e] = __builtin_fma(pa, phi[i*DPE+b], Dl[e]); e++; }
            }
        }
    }
#pragma unroll
    for (int e = 0; e < ND; e++)
        if (Dl[e] != 0.) atomic_add_f64(&D[(size_t)d*ND+e], Dl[e]);
    if (overflow) atomicAdd(&P.counters[5], (unsigned long long)overflow);
    if (npairs) { atomicAdd(&P.counters[3], npairs); atomicAdd(&P.counters[4], nevals); }
}

// =====================================================================================================================
// H2 far field (clusterMethodCy.pyx): Chebyshev interpolation of the kernel on admissible cluster pairs
// (assembleFarFieldInteractions :2153-2238, factor -2 for the (u(x)-u(y))(v(x)-v(y)) form), leaf values
// int phi_I L_alpha (enterLeafValues :1205-1325), upward / downward passes with the transfer operators
// (:1092-1124, :1157-1180; the transfer matrices :2004-2073 are built on the host) and H2Matrix.matvec :2269-2295.
// Tensor index alpha = alpha_0 + m alpha_1 (coordinate 0 fastest) in every array of this file.
// struct H2Dev: pnl_device.h

// j-th Chebyshev node of [a, b]: eta_j = cos((2 (m-j) - 1) pi / (2m)) (clusterMethodCy.pyx:2173, 1255)
__device__ __forceinline__ double cheb_node(double a, double b, int m, int j) {
    return (b-a)*0.5*(cos((2.0*(m-j)-1.0)/(2.0*m)*3.14159265358979323846)+1.0)+a;
}

// 1D Lagrange polynomial l on the Chebyshev nodes of [a, b] at x
__device__ __forceinline__ double lagrange1d(double a, double b, int m, int l, double x) {
    const double xl = cheb_node(a, b, m, l);
    double v = 1.;
    for (int k = 0; k < m; k++)
        if (k != l) {
            const double xk = cheb_node(a, b, m, k);
            v *= (x-xk)/(xl-xk);
        }
    return v;
}

template <int DIM>
__global__ void __launch_bounds__(PNL_NTHREADS)
k_h2_kernel_interp(const DevProblem P, const H2Dev H, const DevKernel *__restrict__ kcls, const int *__restrict__ far_class) {
    const int pr = blockIdx.x, n1 = H.far[2*pr], n2 = H.far[2*pr+1];
    const DevKernel kn = far_class ? kcls[far_class[pr]] : P.k;
    const double *b1 = H.box+(size_t)n1*DIM*2, *b2 = H.box+(size_t)n2*DIM*2;
    for (int t = threadIdx.x; t < H.M*H.M; t += PNL_NTHREADS) {
        const int i = t/H.M, j = t-i*H.M;
        double d2 = 0.;
        int ii = i, jj = j;
#pragma unroll
        for (int d = 0; d < DIM; d++) {
            const double x = cheb_node(b1[2*d], b1[2*d+1], H.m, ii % H.m), y = cheb_node(b2[2*d], b2[2*d+1], H.m, jj % H.m);
            ii /= H.m; jj /= H.m;
            d2 += (x-y)*(x-y);
        }
        H.K[(size_t)pr*H.M*H.M+t] = -2.*kern_eval<0>(kn, d2);
    }
}

// V_leaf[lcl_dof][alpha] = sum over the cells of the leaf and the quadrature points of vol w phi_k(x) L_alpha(x)
template <int DIM, int DPE>
__global__ void __launch_bounds__(PNL_NTHREADS)
k_h2_leaf_values(const DevProblem P, const H2Dev H, int nq, const double *__restrict__ qbary, const double *__restrict__ qw,
                 const double *__restrict__ qphi) {
    constexpr int NV = DIM+1;
    const int lf = blockIdx.x, node = H.leaf_node[lf];
    const double *bx = H.box+(size_t)node*DIM*2;
    const int *dofs = H.leaf_dofs+H.leaf_dof_off[lf];
    const int nd = H.leaf_dof_off[lf+1]-H.leaf_dof_off[lf];
    const int *cells = H.leaf_cells+H.leaf_cell_off[lf];
    const int ncl = H.leaf_cell_off[lf+1]-H.leaf_cell_off[lf];
    double *V = H.V+H.leaf_val_off[lf];
    // Chebyshev nodes of the leaf's box, once per workgroup (the cosines were 3/4 of the kernel: 5.2 -> 1.x ms at 49k DoFs)
    constexpr int MAXM = 64;
    __shared__ double s_node[DIM][MAXM];
    const bool tab = H.m <= MAXM;
    if (tab) {
        for (int t = threadIdx.x; t < DIM*H.m; t += PNL_NTHREADS) s_node[t/H.m][t % H.m] = cheb_node(bx[2*(t/H.m)], bx[2*(t/H.m)+1], H.m, t % H.m);
        __syncthreads();
    }
    for (int t = threadIdx.x; t < ncl*H.M; t += PNL_NTHREADS) {
        const int c = cells[t/H.M], alpha = t % H.M;
        int lcl[DPE];
        bool any = false;
#pragma unroll
        for (int k = 0; k < DPE; k++) {
            const int I = P.cdof[(size_t)k*P.ncp+c];
            int lo = 0, hi = nd;
            while (lo < hi) { const int mid = (lo+hi) >> 1; if (dofs[mid] < I) lo = mid+1; else hi = mid; }
            lcl[k] = (I >= 0 && lo < nd && dofs[lo] == I) ? lo : -1;
            any = any || lcl[k] >= 0;
        }
        if (!any) continue;
        double acc[DPE];
#pragma unroll
        for (int k = 0; k < DPE; k++) acc[k] = 0.;
        const double vol = P.cvol[c];
        for (int j = 0; j < nq; j++) {
            double L = 1.;
            int aa = alpha;
#pragma unroll
            for (int d = 0; d < DIM; d++) {
                double x = 0.;
#pragma unroll
                for (int v = 0; v < NV; v++) x = __builtin_fma(qbary[3*j+v], P.cellv[(size_t)(v*DIM+d)*P.ncp+c], x);
                const int l = aa % H.m;
                if (tab) {
                    // lagrange1d with the nodes from the table: the same operations in the same order
                    const double xl = s_node[d][l];
                    double v = 1.;
                    for (int k = 0; k < H.m; k++)
                        if (k != l) { const double xk = s_node[d][k]; v *= (x-xk)/(xl-xk); }
                    L *= v;
                } else L *= lagrange1d(bx[2*d], bx[2*d+1], H.m, l, x);
                aa /= H.m;
            }
            const double wl = vol*qw[j]*L;
#pragma unroll
            for (int k = 0; k < DPE; k++) acc[k] = __builtin_fma(wl, qphi[j*DPE+k], acc[k]);
        }
#pragma unroll
        for (int k = 0; k < DPE; k++)
            if (lcl[k] >= 0) atomic_add_f64(&V[(size_t)lcl[k]*H.M+alpha], acc[k]);
    }
}

// upward pass, leaves: cup[node][alpha] = sum_dofs x[dof] V[dof][alpha]
__global__ void __launch_bounds__(64)
k_h2_up_leaves(const H2Dev H, const double *__restrict__ x) {
    const int lf = blockIdx.x, node = H.leaf_node[lf];
    const int *dofs = H.leaf_dofs+H.leaf_dof_off[lf];
    const int nd = H.leaf_dof_off[lf+1]-H.leaf_dof_off[lf];
    const double *V = H.V+H.leaf_val_off[lf];
    for (int a = threadIdx.x; a < H.M; a += 64) {
        double s = 0.;
        for (int k = 0; k < nd; k++) s = __builtin_fma(x[dofs[k]], V[(size_t)k*H.M+a], s);
        H.cup[(size_t)node*H.M+a] = s;
    }
}

// y[i] += sum_j B[i][j] x[j] for a row-major M x M block, one wave: the lanes run over the COLUMNS, so a row is read as contiguous
// segments (a lane per row reads with a stride of M doubles: 64 cache lines per load), one wave reduction per row; lane i keeps row i and
// the results leave as one coalesced set of atomics per 64 rows
__device__ __forceinline__ void h2_block_matvec_add(const double *__restrict__ B, int M, const double *__restrict__ x, double *__restrict__ y) {
    const int lane = threadIdx.x & 63;
    for (int i0 = 0; i0 < M; i0 += 64) {
        double mine = 0.;
        const int rows = min(64, M-i0);
        for (int ii = 0; ii < rows; ii++) {
            const double *__restrict__ row = B+(size_t)(i0+ii)*M;
            double s = 0.;
            for (int j = lane; j < M; j += 64) s = __builtin_fma(row[j], x[j], s);
            s = wave_sum(s);
            mine = (lane == ii) ? s : mine;
        }
        if (lane < rows) atomic_add_f64(&y[i0+lane], mine);
    }
}

// upward pass, one level: cup[parent] += T_child cup[child] for the nodes of the level (list of children)
__global__ void __launch_bounds__(64)
k_h2_up_level(const H2Dev H, const int *__restrict__ nodes, int n) {
    const int c = nodes[blockIdx.x], p = H.parent[c];
    (void)n;
    h2_block_matvec_add(H.T+(size_t)c*H.M*H.M, H.M, H.cup+(size_t)c*H.M, H.cup+(size_t)p*H.M);
}

// far field: cdown[n1] += K cup[n2]
__global__ void __launch_bounds__(64)
k_h2_far(const H2Dev H) {
    const int pr = blockIdx.x, n1 = H.far[2*pr], n2 = H.far[2*pr+1];
    h2_block_matvec_add(H.K+(size_t)pr*H.M*H.M, H.M, H.cup+(size_t)n2*H.M, H.cdown+(size_t)n1*H.M);
}

// downward pass, one level: cdown[child] += T_child^T cdown[parent]
__global__ void __launch_bounds__(64)
k_h2_down_level(const H2Dev H, const int *__restrict__ nodes, int n) {
    const int c = nodes[blockIdx.x], p = H.parent[c];
    (void)n;
    const double *T = H.T+(size_t)c*H.M*H.M;
    for (int j = threadIdx.x; j < H.M; j += 64) {
        double s = 0.;
        for (int i = 0; i < H.M; i++) s = __builtin_fma(T[(size_t)i*H.M+j], H.cdown[(size_t)p*H.M+i], s);
        H.cdown[(size_t)c*H.M+j] += s;           // every child is written by one workgroup, after its parent's level
    }
}

// downward pass, leaves: y[dof] += sum_alpha V[dof][alpha] cdown[node][alpha]
__global__ void __launch_bounds__(64)
k_h2_down_leaves(const H2Dev H, double *__restrict__ y) {
    const int lf = blockIdx.x, node = H.leaf_node[lf];
    const int *dofs = H.leaf_dofs+H.leaf_dof_off[lf];
    const int nd = H.leaf_dof_off[lf+1]-H.leaf_dof_off[lf];
    const double *V = H.V+H.leaf_val_off[lf];
    for (int k = threadIdx.x; k < nd; k += 64) {
        double s = 0.;
        for (int a = 0; a < H.M; a++) s = __builtin_fma(V[(size_t)k*H.M+a], H.cdown[(size_t)node*H.M+a], s);
        y[dofs[k]] += s;                         // leaves partition the DoFs
    }
}

// ---------------------------------------------------------------------------------------------
// Exact order range of a tile (block a, block b), a != b, both blocks full: the order formula (FL2:622-642) of all
// TILE x TILE cell pairs, evaluated with the SAME function the general tile kernels use (quad_order_fast), and the
// shared-vertex test of getProtoPanelType (NO:280-378).  out[t] = q if every pair is a distant pair of order q, else 0.
// Run once per tile list on the tiles the host's two-sided bounds (tile_uniform_order) could not settle: the ring where
// the order steps from q to q+1 is a few cells wide, the bounds over two 64-cell blocks leave a band of several blocks.
template <int TILE>
__global__ void __launch_bounds__(256)
k_tile_order_range(const DevProblem P, const DevFormula qo, const int2 *__restrict__ tiles, int ntiles, signed char *__restrict__ out) {
    __shared__ double s_cen[4][TILE], s_h[2][TILE], s_Ld[2][TILE];
    __shared__ float s_lh[4][TILE];
    __shared__ int s_vid[6][TILE];
    __shared__ int s_red[3];
    const int tid = threadIdx.x;
#pragma unroll 1
    for (int t = blockIdx.x; t < ntiles; t += gridDim.x) {
        const int2 tl = tiles[t];
        for (int u = tid; u < 2*TILE; u += 256) {
            const int side = u/TILE, l = u-side*TILE, c = (side ? tl.y : tl.x)*TILE+l;
            s_cen[side*2+0][l] = P.ccen[c];
            s_cen[side*2+1][l] = P.ccen[(size_t)P.ncp+c];
            s_h[side][l] = P.ch[c];
            const double lh = P.clog[c], Ld = P.clog[(size_t)P.ncp+c];
            s_Ld[side][l] = Ld;
            s_lh[side*2+0][l] = (float)lh;
            s_lh[side*2+1][l] = (float)Ld;
#pragma unroll
            for (int k = 0; k < 3; k++) s_vid[side*3+k][l] = P.cvid[(size_t)k*P.ncp+c];
        }
        if (tid == 0) { s_red[0] = 1 << 30; s_red[1] = 0; s_red[2] = 0; }
        __syncthreads();
        int mn = 1 << 30, mx = 0, bad = 0;
        for (int p = tid; p < TILE*TILE; p += 256) {
            const int i = p/TILE, j = p-i*TILE;
            const int a0 = s_vid[0][i], a1 = s_vid[1][i], a2 = s_vid[2][i];
            const int b0 = s_vid[3][j], b1 = s_vid[4][j], b2 = s_vid[5][j];
            // padding cells carry negative vertex ids: their pairs do not exist (blocks with padding BEHIND the last cell never
            // come here: their coordinates are not those of a cell)
            if (a0 < 0 || b0 < 0) continue;
            bad |= (a0 == b0) | (a0 == b1) | (a0 == b2) | (a1 == b0) | (a1 == b1) | (a1 == b2) | (a2 == b0) | (a2 == b1) | (a2 == b2);
            const double dx = s_cen[0][i]-s_cen[2][j], dy = s_cen[1][i]-s_cen[3][j];
            const int q = quad_order_fast(qo, s_h[0][i], s_h[1][j], s_lh[0][i], s_lh[2][j], s_lh[1][i], s_lh[3][j], s_Ld[0][i], s_Ld[1][j],
                                          dx*dx+dy*dy);
            mn = min(mn, q); mx = max(mx, q);
        }
        atomicMin(&s_red[0], mn); atomicMax(&s_red[1], mx); atomicOr(&s_red[2], bad);
        __syncthreads();
        if (tid == 0) out[t] = (signed char)((!s_red[2] && s_red[0] == s_red[1] && s_red[0] >= 2 && s_red[0] < 120) ? s_red[0] : 0);
        __syncthreads();
    }
}
