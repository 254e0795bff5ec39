// Mixed-order tiles of the 2D P1 dense path in the layout of the uniform-tile kernels (pnl_tile2.h).
//
// A tile is "mixed" when its 64 x 64 cell pairs do not provably share one quadrature order: the ring where the order formula
// (FL2:622-642) steps from 2 to 3 (to 4), tiles next to the diagonal, tiles with touching pairs.  k_tile_distant classifies such
// a tile into per-order lists and integrates ONE PAIR PER LANE: every lane reads both cells from LDS and scatters its 21 local
// entries on its own (570 / 1000 instructions per pair of order 2 / 3 against 316 in k_tile_pure; 26 of the kernel's 37 ms at
// 98,304 cells).  Here lane = cell i of block a, the four waves split the cells j of block b like in the uniform kernels, and
// the order of the pair (i, j) is computed in the loop.  The order changes across a ring around cell j, so for most j the
// whole wave agrees: the factorised evaluator of that order runs with broadcast j data (quadrature points of cell j for the
// three unrolled rules are staged per tile), cell i in registers, row sums per (cell, point) in LDS.  Where the wave
// disagrees the evaluators of the orders that occur run one after the other, lanes of the other orders contribute nothing.
// Orders without an unrolled rule (> 4) go to the global work list like before; touching pairs are skipped (k_singular_pairs).
// Same numbers as eval_distant (NO:722-789), same classification as k_tile_distant (NO:280-378, NO:493-540), same counters.
#pragma once
#include "pnl_tile2.h"

#define MX_TILE 64
#define MX_NT 256
#define MX_NR 15          // row / column sums per (side, cell): the 3-point rule (order 2), the 6-point rules of orders 3 and 4
#define MX_NY 30          // doubles per b-cell: its 15 quadrature points

struct MxSmem {
    static constexpr int TILE = MX_TILE, NC = 6, DPE = 3, ND = 6, NR = MX_NR;
    // doubles
    static constexpr int o_y = 0;                          // [TILE][MX_NY]
    static constexpr int o_av = o_y+TILE*MX_NY;            // [TILE][NC] vertices of the a-cells
    static constexpr int o_vol = o_av+TILE*NC;             // [2][TILE]
    static constexpr int o_cen = o_vol+2*TILE;             // [2][2][TILE]
    static constexpr int o_h = o_cen+4*TILE;               // [2][TILE]
    static constexpr int o_Ld = o_h+2*TILE;                // [2][TILE]
    static constexpr int o_Ra = o_Ld+2*TILE;               // [TILE][NR]
    static constexpr int o_Rb = o_Ra+TILE*NR;              // [TILE][NR]
    static constexpr int o_PP = o_Rb+TILE*NR;              // [ND][NR]
    static constexpr int n_dbl = o_PP+ND*NR;
    // ints
    static constexpr int o_lh = 0;                         // float [2][2][TILE]: ln h, |ln(h/H0)|
    static constexpr int o_vid = o_lh+4*TILE;              // [2][3][TILE]
    static constexpr int o_slotb = o_vid+6*TILE;           // [TILE][DPE]
    static constexpr int o_sa = o_slotb+DPE*TILE;          // [TILE][DPE]
    static constexpr int o_hd = o_sa+DPE*TILE;             // [2][TILE] has-a-DoF flags
    static constexpr int o_cnt = o_hd+2*TILE;              // [PNL_MAXQ+2] pairs per order (orders > 4)
    static constexpr int o_misc = o_cnt+PNL_MAXQ+2;        // [4]: far count, work-list base, next tile
    static constexpr int o_far = o_misc+4;                 // unsigned short [TILE*TILE] far list (j << 6 | i)
    static constexpr int n_int = o_far+TILE*TILE/2;        // then [2][nUe] global DoFs of both blocks
    __host__ __device__ static constexpr size_t fixed_bytes(int nUe) { return sizeof(double)*(size_t)n_dbl+sizeof(int)*(size_t)(n_int+2*nUe); }
};

// one unrolled rule: the pairs (lane i, cell j) of the lanes with act
template <int NP, int KT>
__device__ __forceinline__ void mx_eval(const pnl_const_f64_ptr rule, const DevKernel &kk, const double (&av)[6], const int (&sa)[3],
                                        bool act, double vola, double scale2, int li, int j, int roff, const double *__restrict__ s_yj,
                                        double volb, const int *__restrict__ s_slotbj, double *__restrict__ s_Ra, double *__restrict__ s_Rbj,
                                        double *__restrict__ s_acc) {
    constexpr int DPE = 3, DIM = 2, NV = 3;
    constexpr int R_BARY = 0, R_W = 3*NP, R_WPH = R_W+NP;
    double y[NP][DIM];
#pragma unroll
    for (int jp = 0; jp < NP; jp++)
#pragma unroll
        for (int d = 0; d < DIM; d++) y[jp][d] = s_yj[jp*DIM+d];
    // NA:1405-1410: symmetric cell pairs count twice
    const double vv = act ? scale2*vola*volb : 0.;
    double c[NP], G[DPE][DPE];
#pragma unroll
    for (int jp = 0; jp < NP; jp++) c[jp] = 0.;
#pragma unroll
    for (int a = 0; a < DPE; a++)
#pragma unroll
        for (int b = 0; b < DPE; b++) G[a][b] = 0.;
#pragma unroll (NP == 3 ? 3 : 1)
    for (int ip = 0; ip < NP; ip++) {
        double x[DIM];
#pragma unroll
        for (int d = 0; d < DIM; d++) {
            double sx = 0.;
#pragma unroll
            for (int k = 0; k < NV; k++) sx = __builtin_fma(rule[R_BARY+3*ip+k], av[k*DIM+d], sx);
            x[d] = sx;
        }
        const double wi = rule[R_W+ip];
        double r = 0., u[DPE];
#pragma unroll
        for (int b = 0; b < DPE; b++) u[b] = 0.;
#pragma unroll
        for (int jp = 0; jp < NP; jp++) {
            double d2 = 0.;
#pragma unroll
            for (int d = 0; d < DIM; d++) { const double t = x[d]-y[jp][d]; d2 = __builtin_fma(t, t, d2); }
            // lanes of another order (or of a touching / skipped pair, where the cells may coincide) evaluate a harmless value
            const double g = kern_eval<KT>(kk, act ? d2 : 1.);
            r = __builtin_fma(rule[R_W+jp], g, r);
            c[jp] = __builtin_fma(wi, g, c[jp]);
#pragma unroll
            for (int b = 0; b < DPE; b++) u[b] = __builtin_fma(g, rule[R_WPH+jp*DPE+b], u[b]);
        }
        if (act) lds_add_f64(&s_Ra[li*MX_NR+roff+ip], vv*r);
#pragma unroll
        for (int a = 0; a < DPE; a++) {
            const double pa = rule[R_WPH+ip*DPE+a];
#pragma unroll
            for (int b = 0; b < DPE; b++) G[a][b] = __builtin_fma(pa, u[b], G[a][b]);
        }
    }
    if (act) {
#pragma unroll
        for (int b = 0; b < DPE; b++) {
            const int sb = s_slotbj[b];
#pragma unroll
            for (int a = 0; a < DPE; a++) lds_add_f64(&s_acc[sa[a]+sb], -vv*G[a][b]);
        }
    }
    // column sums over the lanes of this order -> scaled column sums of cell j for this rule (this wave owns cell j)
    const double sb2 = scale2*volb;
    const double wa = act ? vola : 0.;
#pragma unroll
    for (int g3 = 0; g3 < NP; g3 += 3) {
        double c0, c1, c2;
        group_sum3<64>(wa*c[g3], wa*c[g3+1], wa*c[g3+2], c0, c1, c2);
        const int k = li-g3;
        if (k >= 0 && k < 3) s_Rbj[roff+li] = sb2*(k == 0 ? c0 : (k == 1 ? c1 : c2));
    }
}

template <int KT>
__global__ void __launch_bounds__(MX_NT, 2)
k_tile_mixed(const DevProblem P, const int2 *__restrict__ tiles, int ntiles, double *__restrict__ A, long long ldA,
             double *__restrict__ Dglob, int cell_begin, int cell_end, int acc_stride, int4 *__restrict__ worklist,
             unsigned *__restrict__ wl_count, unsigned wl_cap, int flags, const double *__restrict__ rules_g, int off2, int off3,
             int off4, int nUe, unsigned *__restrict__ tile_ctr, const SlotOut SO) {
    using S = MxSmem;
    constexpr int TILE = MX_TILE, NT = MX_NT, NW = NT/64, JW = TILE/NW, DPE = 3, ND = 6, NC = 6, NV = 3, NR = MX_NR;
    const pnl_const_f64_ptr rule2 = (pnl_const_f64_ptr)(unsigned long long)(rules_g+off2);
    const pnl_const_f64_ptr rule3 = (pnl_const_f64_ptr)(unsigned long long)(rules_g+(off3 >= 0 ? off3 : 0));
    const pnl_const_f64_ptr rule4 = (pnl_const_f64_ptr)(unsigned long long)(rules_g+(off4 >= 0 ? off4 : 0));
    const bool have3 = off3 >= 0, have4 = off4 >= 0;
    extern __shared__ double smem[];
    double *s_y = smem+S::o_y, *s_av = smem+S::o_av, *s_vol = smem+S::o_vol, *s_cen = smem+S::o_cen, *s_h = smem+S::o_h;
    double *s_Ld = smem+S::o_Ld, *s_Ra = smem+S::o_Ra, *s_Rb = smem+S::o_Rb, *s_PP = smem+S::o_PP;
    int *si = (int*)(smem+S::n_dbl+(S::n_dbl & 1));
    float *s_lh = (float*)(si+S::o_lh);
    int *s_vid = si+S::o_vid, *s_slotb = si+S::o_slotb, *s_sa = si+S::o_sa, *s_hd = si+S::o_hd, *s_cnt = si+S::o_cnt, *s_misc = si+S::o_misc;
    unsigned short *s_far = (unsigned short*)(si+S::o_far);
    int *s_dof = si+S::n_int;
    double *s_acc = (double*)(s_dof+2*nUe);              // [nUe+1][acc_stride]; nUe even, n_int even
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, li = lane;
    const int wv = __builtin_amdgcn_readfirstlane(wave);
    const unsigned long long lt = (1ull << lane)-1ull;
    const DevKernel kk = P.k;
    const double scale2 = 2.*kern_scale<KT>(kk);
    unsigned long long cnt2 = 0, cnt3 = 0, cnt4 = 0;
    int overflow = 0;
    // w phi_a phi_b at the 15 points of the three rules
    for (int t = tid; t < ND*NR; t += NT) {
        const int e = t/NR, k = t-e*NR;
        double v = 0.;
        if (k < 3) v = rule2[(3*3+3+3*DPE)+e*3+k];
        else if (k < 9) { if (have3) v = rule3[(3*6+6+6*DPE)+e*6+(k-3)]; }
        else if (have4) v = rule4[(3*6+6+6*DPE)+e*6+(k-9)];
        s_PP[t] = v;
    }
    for (int t = tid; t < PNL_MAXQ+2; t += NT) s_cnt[t] = 0;
    for (int t = tid; t < (nUe+1)*acc_stride; t += NT) s_acc[t] = 0.;
#pragma unroll 1
    for (int tile_idx = blockIdx.x; tile_idx < ntiles; ) {
        const int2 tl = tiles[tile_idx];
        const int ta = tl.x, tb = tl.y;
        const int nA = P.blk_ndof[ta], nB = P.blk_ndof[tb];
        // ---- stage both blocks (b-side: threads [0, TILE), a-side: threads [TILE, 2 TILE)) ----
        if (tid < 2*TILE) {
            const bool bside = tid < TILE;
            const int side = bside ? 1 : 0, l = bside ? tid : tid-TILE, c = (bside ? tb : ta)*TILE+l;
            double v[NC];
#pragma unroll
            for (int k = 0; k < NC; k++) v[k] = P.cellv[(size_t)k*P.ncp+c];
            s_vol[side*TILE+l] = P.cvol[c];
            s_cen[(side*2+0)*TILE+l] = P.ccen[c];
            s_cen[(side*2+1)*TILE+l] = P.ccen[(size_t)P.ncp+c];
            s_h[side*TILE+l] = P.ch[c];
            const double lh = P.clog[c], Ld = P.clog[(size_t)P.ncp+c];
            s_Ld[side*TILE+l] = Ld;
            s_lh[(side*2+0)*TILE+l] = (float)lh;
            s_lh[(side*2+1)*TILE+l] = (float)Ld;
            int any = 0;
#pragma unroll
            for (int k = 0; k < NV; k++) s_vid[(side*NV+k)*TILE+l] = P.cvid[(size_t)k*P.ncp+c];
#pragma unroll
            for (int k = 0; k < DPE; k++) {
                const int sl = P.cslot[(size_t)k*P.ncp+c];
                any |= (sl >= 0);
                if (bside) s_slotb[l*DPE+k] = sl >= 0 ? sl : nUe;
                else s_sa[l*DPE+k] = (sl >= 0 ? sl : nUe)*acc_stride;
            }
            s_hd[side*TILE+l] = any;
            if (bside) {
                // quadrature points of the cell for the three rules
#pragma unroll
                for (int rr = 0; rr < 3; rr++) {
                    const pnl_const_f64_ptr rl = rr == 0 ? rule2 : (rr == 1 ? rule3 : rule4);
                    const int np = rr == 0 ? 3 : 6, yo = rr == 0 ? 0 : (rr == 1 ? 6 : 18);
                    if ((rr == 1 && !have3) || (rr == 2 && !have4)) continue;
                    for (int jp = 0; jp < np; jp++)
#pragma unroll
                        for (int d = 0; d < 2; d++) {
                            double sy = 0.;
#pragma unroll
                            for (int k = 0; k < NV; k++) sy = __builtin_fma(rl[3*jp+k], v[k*2+d], sy);
                            s_y[l*MX_NY+yo+jp*2+d] = sy;
                        }
                }
            } else {
#pragma unroll
                for (int k = 0; k < NC; k++) s_av[l*NC+k] = v[k];
            }
        }
        {
            const int *__restrict__ dofA = P.blk_dofs+(size_t)ta*P.blk_stride;
            const int *__restrict__ dofB = P.blk_dofs+(size_t)tb*P.blk_stride;
            for (int k = tid; k < nA; k += NT) s_dof[k] = P.rowmap ? P.rowmap[dofA[k]] : dofA[k];
            for (int k = tid; k < nB; k += NT) s_dof[nUe+k] = P.colmap ? P.colmap[dofB[k]] : dofB[k];
        }
        for (int t = tid; t < 2*TILE*NR; t += NT) s_Ra[t] = 0.;                   // s_Ra and s_Rb are adjacent
        if (tid == 0) { s_misc[0] = 0; s_misc[2] = (int)gridDim.x+(int)atomicAdd(tile_ctr, 1u); }
        __syncthreads();
        // ---- a side of this lane ----
        double av[NC];
#pragma unroll
        for (int k = 0; k < NC; k++) av[k] = s_av[li*NC+k];
        int sa[DPE], va[NV];
#pragma unroll
        for (int k = 0; k < DPE; k++) sa[k] = s_sa[li*DPE+k];
#pragma unroll
        for (int k = 0; k < NV; k++) va[k] = s_vid[(0*NV+k)*TILE+li];
        const bool ha = s_hd[li] != 0;
        const double vola = s_vol[li], cax = s_cen[li], cay = s_cen[TILE+li], h1 = s_h[li], Ld1 = s_Ld[li];
        const float lh1 = s_lh[li], L1 = s_lh[TILE+li];
        const int ca = ta*TILE+li;
        const bool arow = va[0] >= 0 && ca >= cell_begin && ca < cell_end;
#pragma unroll 1
        for (int jj = 0; jj < JW; jj++) {
            const int j = wv*JW+jj;
            // ---- order of the pair (i, j): NO:280-378 vertex test, NO:493-540 + FL2:622-642 ----
            const int vb0 = s_vid[(1*NV+0)*TILE+j], vb1 = s_vid[(1*NV+1)*TILE+j], vb2 = s_vid[(1*NV+2)*TILE+j];
            bool ok = arow && vb0 >= 0 && (ta < tb || li < j) && (ha || s_hd[TILE+j] != 0);
            bool shared = false;
#pragma unroll
            for (int k = 0; k < NV; k++) shared = shared || va[k] == vb0 || va[k] == vb1 || va[k] == vb2;
            int q = 0;
            if (ok && !shared) {
                const double dx = cax-s_cen[2*TILE+j], dy = cay-s_cen[3*TILE+j];
                q = quad_order_fast(P.qo, h1, s_h[TILE+j], lh1, s_lh[2*TILE+j], L1, s_lh[3*TILE+j], Ld1, s_Ld[TILE+j], dx*dx+dy*dy);
                if (q > P.qmax || q > PNL_MAXQ) { overflow++; q = 0; }
            }
            const bool a2 = q == 2, a3 = q == 3 && have3, a4 = q == 4 && have4, aF = q >= 2 && !a2 && !a3 && !a4;
            const unsigned long long m2 = __ballot(a2), m3 = __ballot(a3), m4 = __ballot(a4), mF = __ballot(aF);
            cnt2 += __popcll(m2); cnt3 += __popcll(__ballot(q == 3)); cnt4 += __popcll(__ballot(q == 4));
            if (mF) {
                // pairs for the global work list: (j, i) now, the order again when the list is written out
                int base = 0;
                const int leader = __ffsll((long long)mF)-1;
                if (lane == leader) base = atomicAdd(&s_misc[0], __popcll(mF));
                base = __builtin_amdgcn_readlane(base, leader);
                if (aF) s_far[base+__popcll(mF & lt)] = (unsigned short)((j << 6) | li);
                // histogram of the orders above 4
                unsigned long long todo = __ballot(q > 4);
                while (todo) {
                    const int ld = __ffsll((long long)todo)-1;
                    const int qL = __builtin_amdgcn_readlane(q, ld);
                    const unsigned long long same = __ballot(q == qL);
                    if (lane == ld) atomicAdd(&s_cnt[qL], __popcll(same));
                    todo &= ~same;
                }
            }
            const double volb = s_vol[TILE+j];
            if (m2) mx_eval<3, KT>(rule2, kk, av, sa, a2, vola, scale2, li, j, 0, s_y+j*MX_NY, volb, s_slotb+j*DPE, s_Ra, s_Rb+j*NR, s_acc);
            if (m3) mx_eval<6, KT>(rule3, kk, av, sa, a3, vola, scale2, li, j, 3, s_y+j*MX_NY+6, volb, s_slotb+j*DPE, s_Ra, s_Rb+j*NR, s_acc);
            if (m4) mx_eval<6, KT>(rule4, kk, av, sa, a4, vola, scale2, li, j, 9, s_y+j*MX_NY+18, volb, s_slotb+j*DPE, s_Ra, s_Rb+j*NR, s_acc);
        }
        __syncthreads();
        // ---- far pairs -> the global work list (one reservation per tile) ----
        {
            const int nF = s_misc[0];
            if (nF) {
                if (tid == 0) s_misc[1] = (int)atomicAdd(wl_count, (unsigned)nF);
                __syncthreads();
                const unsigned base = (unsigned)s_misc[1];
                for (int t = tid; t < nF; t += NT) {
                    const int p = s_far[t], j = p >> 6, i = p & 63;
                    const double dx = s_cen[i]-s_cen[2*TILE+j], dy = s_cen[TILE+i]-s_cen[3*TILE+j];
                    const int q = quad_order_fast(P.qo, s_h[i], s_h[TILE+j], s_lh[i], s_lh[2*TILE+j], s_lh[TILE+i], s_lh[3*TILE+j], s_Ld[i],
                                                  s_Ld[TILE+j], dx*dx+dy*dy);
                    const int off = P.off[q];
                    if (base+t < wl_cap) worklist[base+t] = make_int4(ta*TILE+i, tb*TILE+j, off, (P.off[q+1]-off) | (q << 16));
                }
            }
        }
        // ---- flush: sub-block of A' ----
        const bool sym = (flags & 1) != 0;
        if (SO.A2) {
            const int cb_ = SO.colbase[ta], W = SO.S-cb_;
            double *__restrict__ base = SO.A2+SO.rowoff[ta]+(SO.colbase[tb]-cb_);
#pragma unroll 1
            for (int r = wv; r < nA; r += NW) {
                double *__restrict__ row = base+(long long)r*W;
                for (int cc = 2*lane; cc < nB; cc += 128) {
                    double2 v = make_double2(0., 0.);
                    v.x = s_acc[r*acc_stride+cc]; s_acc[r*acc_stride+cc] = 0.;
                    if (cc+1 < nB) { v.y = s_acc[r*acc_stride+cc+1]; s_acc[r*acc_stride+cc+1] = 0.; }
                    *(double2*)(row+cc) = v;
                }
            }
        } else {
#pragma unroll 1
            for (int r = wv; r < nA; r += NW) {
                double *__restrict__ row = A+(long long)__builtin_amdgcn_readfirstlane(s_dof[r])*ldA;
                for (int cc = lane; cc < nB; cc += 64) {
                    const double v = s_acc[r*acc_stride+cc];
                    if (v != 0.) {
                        if (!sym) s_acc[r*acc_stride+cc] = 0.;
                        atomic_add_f64(&row[s_dof[nUe+cc]], v);
                    }
                }
            }
            if (sym) {
                __syncthreads();
#pragma unroll 1
                for (int cc = wv; cc < nB; cc += NW) {
                    double *__restrict__ row = A+(long long)__builtin_amdgcn_readfirstlane(s_dof[nUe+cc])*ldA;
                    for (int r = lane; r < nA; r += 64) {
                        const double v = s_acc[r*acc_stride+cc];
                        if (v != 0.) {
                            s_acc[r*acc_stride+cc] = 0.;
                            atomic_add_f64(&row[s_dof[r]], v);
                        }
                    }
                }
            }
        }
        // trash row / column (boundary DoFs) back to zero
        for (int t = tid; t < acc_stride; t += NT) s_acc[nUe*acc_stride+t] = 0.;
        for (int t = tid; t <= nUe; t += NT) s_acc[t*acc_stride+nUe] = 0.;
        // diagonal blocks of both sides from the scaled row / column sums of the three rules
        for (int t = tid; t < 2*TILE*ND; t += NT) {
            const int side = t/(TILE*ND), rem = t-side*TILE*ND, cl = rem/ND, e = rem-cl*ND;
            const double *__restrict__ R = side ? s_Rb+cl*NR : s_Ra+cl*NR;
            double v = 0.;
#pragma unroll
            for (int k = 0; k < NR; k++) v = __builtin_fma(s_PP[e*NR+k], R[k], v);
            if (v != 0.) atomic_add_f64(&Dglob[(size_t)((side ? tb : ta)*TILE+cl)*ND+e], v);
        }
        const int nxt = s_misc[2];
        __syncthreads();
        tile_idx = nxt;
    }
    // ---- statistics (the same counters as k_tile_distant: pairs per order incl. the deferred ones, evaluations n^2) ----
    if (overflow) atomicAdd(&P.counters[5], (unsigned long long)overflow);
    if (lane == 0) {
        const unsigned long long tot = cnt2+cnt3+cnt4;
        if (tot) {
            atomicAdd(&P.counters[1], tot);
            atomicAdd(&P.counters[2], 9ull*cnt2);
            if (cnt2) atomicAdd(&P.counters[8+2], cnt2);
        }
        if (cnt3) { atomicAdd(&P.counters[8+3], cnt3); const unsigned long long n = (unsigned long long)(P.off[4]-P.off[3]); atomicAdd(&P.counters[2], n*n*cnt3); }
        if (cnt4) { atomicAdd(&P.counters[8+4], cnt4); const unsigned long long n = (unsigned long long)(P.off[5]-P.off[4]); atomicAdd(&P.counters[2], n*n*cnt4); }
    }
    __syncthreads();
    for (int q = 5+tid; q <= PNL_MAXQ; q += NT) {
        const int cq = s_cnt[q];
        if (cq) {
            const unsigned long long n = (unsigned long long)(P.off[q+1]-P.off[q]);
            atomicAdd(&P.counters[8+q], (unsigned long long)cq);
            atomicAdd(&P.counters[1], (unsigned long long)cq);
            atomicAdd(&P.counters[2], n*n*(unsigned long long)cq);
        }
    }
}
