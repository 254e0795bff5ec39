// Second generation of the dense tile kernels (gfx950 only, 2D):
//
//   k_tile_uniform<DPE, NP, KT>  tiles whose cell pairs are ALL distant pairs of ONE quadrature order (host-side
//                                conservative bounds on the order formula, tile_uniform_order in pnl_hip.hip): no
//                                classification, no lists.  P2 (32-cell blocks): lane = (cell i of block a, half h), the two
//                                halves of a wave take two cells j of block b at a time; P1 (64-cell blocks): lane = cell i.
//                                NP = 3 (order 2) or 6 (orders 3 and 4) points per triangle.
//   k_tile_p2<KT>                the general tile kernel for P2 (78 local entries per pair).  What made the first version
//                                need 350-390 VGPRs were the 2 x 21 diagonal-block accumulators S1 / S2 beside the 36 of the
//                                cross block G.  The diagonal blocks only depend on the row sums r_i = sum_j w_j g_ij and the
//                                column sums c_j = sum_i w_i g_ij of the kernel values:
//                                   S1[a,b] = sum_i w_i phi_a phi_b(x_i) r_i,   S2[a,b] = sum_j w_j phi_a phi_b(y_j) c_j,
//                                so a pair adds N values per side to per-(cell, point) LDS buffers and the 21 entries per
//                                cell are formed once per tile at the flush.  G + c + y fit 256 VGPRs: 512 threads, two
//                                waves per SIMD.  Waves fetch their 64-pair chunks from an LDS counter, heavy chunks first.
//
// Same numbers as eval_distant (NO:722-789) in its factorised form (see pnl_kernels.h), same LDS sub-block of A'.
#pragma once
#include "pnl_common.h"

// layout of a uniform-tile rule block (doubles): bary[NP][3], w[NP], w phi[NP][DPE], w phi_a phi_b[ND][NP]
__host__ __device__ constexpr int uni_rule_size(int dpe, int np) { return 3*np+np+np*dpe+(dpe*(dpe+1)/2)*np; }

// ---- block-slot storage of the one-sided operator A' ---------------------------------------------------------------------
// A tile (block a, block b) accumulates into rows = DoFs of block a, columns = DoFs of block b.  DoFs on block borders belong
// to several blocks, so in DoF numbering the sub-blocks of different tiles overlap and the flush has to ADD (fp64 atomics: one
// 256-byte request per ~50 ns and CU, the price of a P2 tile's 89 x 89 sub-block is 12 us against 10 us for computing it).  In
// slot numbering -- every block keeps private copies of its DoFs: row (a, r), column (b, c) -- every entry belongs to exactly
// one tile: the flush is a plain coalesced store, nothing has to be zeroed first, and one gather pass (k_fold_mirror) sums
// the copies, A[I][J] = sum over copies (a, r) of I and (b, c) of J, and symmetrises in the same sweep.
//   row (a, r) of the storage holds the columns of the blocks b >= a: W_a = S - colbase[a] entries, S = sum of the padded
//   block widths; rowoff[a] = offset of row (a, 0).
// struct SlotOut: pnl_device.h

// copies of a DoF: cp[k] = (block a, global slot column colbase[a] + r), cprow[k] = offset of that slot's storage row minus
// colbase[a], so that entry (row copy k, column copy l) sits at A2[cprow[k] + cp[l].y] (stored iff cp[l].x >= cp[k].x)
//
// A = A' + A'^T with A' gathered from the block-slot storage; every entry of A is written (no zero fill needed before).
// 32 x 32 blocks of the upper block triangle, both images through an LDS transpose like k_mirror.
// The gather runs over the COPIES: the copies of the 32 row DoFs x the copies of the 32 column DoFs form a grid of about
// 42 x 42 candidate entries of the storage (1.3 copies per DoF); thread (ty, tx) takes the copy rows ty + 8 u and the copy
// columns tx, tx + 32, loads the stored ones (six independent loads in flight) and adds them to the LDS image of the block
// with ds_add_f64.  No per-entry walk whose length differs from lane to lane: the kernel was bound by instruction issue
// (1,050 VALU instructions per wave for 8 entries per thread) as much as by the chain of dependent loads.
// Tables of a range of 32 DoFs come packed (FoldEntry[1 + PNL_FOLD_TAB], pnl_device.h): header + copies, one load per thread.
#ifndef PNL_FOLD_FLAT
#define PNL_FOLD_FLAT 1
#endif
template <bool NT>
__global__ void __launch_bounds__(256)
k_fold_mirror(const double *__restrict__ A2, const FoldEntry *__restrict__ tab, const int *__restrict__ cpoff, const int2 *__restrict__ cp,
              const long long *__restrict__ cprow, double *__restrict__ A, long long ldA, int N) {
    __shared__ double t1[32][33], t2[32][33];
    __shared__ FoldEntry s_tab[2][PNL_FOLD_TAB+1];
    const unsigned nb = (unsigned)(N+31)/32;
    // 8 x 8 super-blocks of the upper block triangle, 64 consecutive workgroups each: the workgroups in flight share the
    // storage rows of 8 row ranges and 8 column ranges (pages, DRAM rows, L2 lines of the XCD) instead of sweeping one block
    // row against thousands of unrelated column ranges
    const unsigned nbs = (nb+7)/8;
    const unsigned sbid = blockIdx.x >> 6, inner = blockIdx.x & 63;
    // sbid = I nbs - I (I-1)/2 + (J - I); the grid has less than 2^31 blocks, so 32-bit arithmetic is exact
    unsigned I;
    {
        const float t = 2.f*(float)nbs+1.f;
        const float fb = (t-sqrtf(fmaxf(t*t-8.f*(float)sbid, 0.f)))*0.5f;
        I = (unsigned)fmaxf(fb, 0.f);
        if (I >= nbs) I = nbs-1;
        while (I > 0 && I*nbs-I*(I-1)/2 > sbid) I--;
        while (I+1 < nbs && (I+1)*nbs-(I+1)*I/2 <= sbid) I++;
    }
    const unsigned J = I+(sbid-(I*nbs-I*(I-1)/2));
    const unsigned bi = 8*I+(inner >> 3), bj = 8*J+(inner & 7);
    if (bi > bj || bj >= nb) return;                     // lower part of a diagonal super-block, padding of the last one
    const int tid = threadIdx.x, tx = tid & 31, ty = tid >> 5;     // 32 x 8
    if (tid < 2*(PNL_FOLD_TAB+1)) {
        const int side = tid >= PNL_FOLD_TAB+1, k = tid-side*(PNL_FOLD_TAB+1);
        s_tab[side][k] = tab[(size_t)(side ? bj : bi)*(PNL_FOLD_TAB+1)+k];
    }
    for (int k = tid; k < 32*33; k += 256) { (&t1[0][0])[k] = 0.; (&t2[0][0])[k] = 0.; }
    __syncthreads();
    const int n0 = (int)s_tab[0][0].off, n1 = (int)s_tab[1][0].off;
    if (n0 <= PNL_FOLD_TAB && n1 <= PNL_FOLD_TAB) {
#if PNL_FOLD_FLAT
        // the nr x nc candidates of an image as ONE index range over the 256 threads, eight independent loads in flight per thread:
        // the (ty, tx) grid left the lanes tx >= nc - 32 of the second column sweep idle (nc is about 42) and had five or six loads
        // per thread in flight
#pragma unroll
        for (int g = 0; g < 2; g++) {                                   // g = 0: rows of range 0 -> t1, g = 1: rows of range 1 -> t2
            if (g && bi == bj) break;
            const int nr = g ? n1 : n0, nc = g ? n0 : n1;
            double *__restrict__ tf = g ? &t2[0][0] : &t1[0][0];
            const unsigned total = (unsigned)(nr*nc), inv = ((1u << 20)+(unsigned)nc-1u)/(unsigned)nc;     // idx / nc exactly for idx < 4096, nc <= 64
            for (unsigned base = 0; base < total; base += 8*256) {
                double v[8];
                int dst[8];
#pragma unroll
                for (int u = 0; u < 8; u++) {
                    const unsigned idx = base+256u*u+(unsigned)tid;
                    dst[u] = -1;
                    v[u] = 0.;
                    if (idx < total) {
                        const unsigned ci = (idx*inv) >> 20, cj = idx-ci*(unsigned)nc;
                        const FoldEntry rw = s_tab[g][1+ci], col = s_tab[1-g][1+cj];
                        if ((col.ar >> 5) >= (rw.ar >> 5)) { v[u] = A2[rw.off+col.cy]; dst[u] = (rw.ar & 31)*33+(col.ar & 31); }
                    }
                }
#pragma unroll
                for (int u = 0; u < 8; u++) if (dst[u] >= 0) lds_add_f64(tf+dst[u], v[u]);
            }
        }
#else
#pragma unroll
        for (int g = 0; g < 2; g++) {                                   // g = 0: rows of range 0 -> t1, g = 1: rows of range 1 -> t2
            if (g && bi == bj) break;
            const int nr = g ? n1 : n0, nc = g ? n0 : n1;
            double *__restrict__ tf = g ? &t2[0][0] : &t1[0][0];
            for (int cj = tx; cj < nc; cj += 32) {
                const FoldEntry col = s_tab[1-g][1+cj];
                const int cb = col.ar >> 5, c = col.ar & 31;
                for (int ci0 = ty; ci0 < nr; ci0 += 48) {
                    double v[6];
                    int dst[6];
#pragma unroll
                    for (int u = 0; u < 6; u++) {
                        const int ci = ci0+8*u;
                        dst[u] = -1;
                        v[u] = 0.;
                        if (ci < nr) {
                            const FoldEntry rw = s_tab[g][1+ci];
                            if (cb >= (rw.ar >> 5)) { v[u] = A2[rw.off+col.cy]; dst[u] = (rw.ar & 31)*33+c; }
                        }
                    }
#pragma unroll
                    for (int u = 0; u < 6; u++) if (dst[u] >= 0) lds_add_f64(tf+dst[u], v[u]);
                }
            }
        }
#endif
    } else {
        // more copies than the packed table holds: per-entry walk over the lists in global memory
        auto gather = [&](int br, int bc, double (*t)[33]) {
            const int J = min(bc*32+tx, N), j0 = cpoff[J], j1 = cpoff[min(J+1, N)];
#pragma unroll 1
            for (int rr = 0; rr < 4; rr++) {
                const int r = ty+8*rr, I = min(br*32+r, N);
                double s = 0.;
                for (int ci = cpoff[I]; ci < cpoff[min(I+1, N)]; ci++) {
                    const int a = cp[ci].x;
                    const double *__restrict__ rowp = A2+cprow[ci];
                    for (int cj = j0; cj < j1; cj++) {
                        const int2 c2 = cp[cj];
                        if (c2.x >= a) s += rowp[c2.y];
                    }
                }
                t[r][tx] = s;
            }
        };
        gather((int)bi, (int)bj, t1);
        if (bi != bj) gather((int)bj, (int)bi, t2);
    }
    __syncthreads();
    for (int r = ty; r < 32; r += 8) {
        const int I = bi*32+r, J = bj*32+tx;
        if (I < N && J < N) {
            const double v = (bi != bj) ? t1[r][tx]+t2[tx][r] : ((r != tx) ? t1[r][tx]+t1[tx][r] : t1[r][tx]);
            if (NT) __builtin_nontemporal_store(v, A+(long long)I*ldA+J); else A[(long long)I*ldA+J] = v;
        }
        const int I2 = bj*32+r, J2 = bi*32+tx;
        if (bi != bj && I2 < N && J2 < N) {
            const double v = t2[r][tx]+t1[tx][r];
            if (NT) __builtin_nontemporal_store(v, A+(long long)I2*ldA+J2); else A[(long long)I2*ldA+J2] = v;
        }
    }
}

// sub-blocks of the tiles that are visited more than once (several order classes) are accumulated with atomics: zero them first
__global__ void __launch_bounds__(256)
k_zero_slot_tiles(const SlotOut SO, const int2 *__restrict__ tiles, const int *__restrict__ blk_ndof) {
    const int a = tiles[blockIdx.x].x, b = tiles[blockIdx.x].y;
    const int nA = blk_ndof[a], nBp = SO.colbase[b+1]-SO.colbase[b], W = SO.S-SO.colbase[a];
    double *__restrict__ base = SO.A2+SO.rowoff[a]+(SO.colbase[b]-SO.colbase[a]);
    for (int t = threadIdx.x; t < nA*nBp; t += 256) {
        const int r = t/nBp, c = t-r*nBp;
        base[(long long)r*W+c] = 0.;
    }
}

// LDS of k_tile_uniform in bytes without the sub-block of A' (host and device agree through this function)
// dof_lists: the global DoF numbers of both blocks, double-buffered (the flush into A in DoF numbering; the block-slot storage
// does not need them: 1.4 KB that let the P2 kernels of a general exponent keep their power tables next to the second workgroup)
__host__ __device__ constexpr size_t uniform_fixed_lds(int dpe, int np, int tile, int nUe, bool dof_lists = true) {
    // per-cell rows have odd strides (in doubles / ints): the lanes of a wave read the rows of 64 different cells
    // (P1; the P2 kernels, two workgroups of just under 80 KB per CU, keep compact rows and the vertices of the a-cells)
    return dpe == 3 ? sizeof(double)*(size_t)(2*tile*(np*2+1)+2*tile+2*tile*(np|1)+(dpe*(dpe+1)/2)*np)
                      +sizeof(int)*(size_t)(2*tile*(dpe|1)+2*tile+(dof_lists ? 4*nUe : 0))
                    : sizeof(double)*(size_t)(tile*np*2+tile*6+2*tile+2*tile*np+(dpe*(dpe+1)/2)*np)
                      +sizeof(int)*(size_t)(2*tile*dpe+2*tile+(dof_lists ? 4*nUe : 0));
}

// ---------------------------------------------------------------------------------------------------------------------
// Software pipeline over the tiles of a workgroup: while the flush of tile n (global atomics) is in flight the inputs of
// tile n+1 are already staged in LDS -- they are loaded BEFORE the flush is issued (the memory counter retires in order:
// a load issued behind the atomics would wait for all of them) and the barriers do not wait for global memory.  The flush
// reads the DoF numbers from LDS for the same reason and zeroes the sub-block as it goes.
// registers: two waves per SIMD for everybody (the compiler takes what it needs: 108-168 VGPRs for P1, ~250 for P2; the launcher asks
// the occupancy API how many workgroups that leaves per CU), except the P1 order-2 kernel of a general exponent, which needs 135
// and is held to 128 so that a fourth workgroup fits (order-2 tiles at 98,304 cells, s = 0.4: 67.1 -> 61.5 ms), and the P1 6-point
// kernels, held to 168 for the third (the general exponent needs 170: order-3 tiles 38.7 -> 28.7 ms)
#ifndef PNL_U33K0_WAVES
#define PNL_U33K0_WAVES 3
#endif
// STRUCT (P1, three points; pnl_context::uni_struct): equal weights w and w phi_b(y_j) = A + B delta_bj.  With the nine kernel values
// g_ab = gamma(x_a, y_b), their row sums s_a, column sums c_b and total S, the cross block is
//   G[a][b] = B^2 g_ab + A B (s_a + c_b) + A^2 S
// -- 49 instead of 87 instructions per pair behind the kernel values.
// STRUCT with six points: two orbits of three points, w phi_b(y_j) = A_o + B_o delta(b, j mod 3) in orbit o = j / 3: per row of the rule
// the orbit sums s_0, s_1 of the kernel values give r = w_0 s_0 + w_1 s_1 and u_b = A_0 s_0 + A_1 s_1 + B_0 g_b + B_1 g_{3+b}
// (14 instead of 24 instructions per row).
template <int DPE, int NP, int KT, bool STRUCT = false>
__global__ void __launch_bounds__(256, (DPE == 3 && NP == 3) ? (KT == 0 ? PNL_U33K0_WAVES : (KT == 2 ? 4 : 2)) : ((DPE == 3 && NP == 6) ? 3 : 2))
k_tile_uniform(const DevProblem P, const int2 *__restrict__ tiles, const int *__restrict__ tile_cls, const DevKernel *__restrict__ kcls,
               int ntiles, double *__restrict__ A, long long ldA, double *__restrict__ Dglob, int acc_stride, int q_uniform,
               int flags, const double *__restrict__ rule_g, int nUe, const SlotOut SO) {
    const int flags_in = flags;
    constexpr int DIM = 2, NV = 3, NC = 6, ND = DPE*(DPE+1)/2, NT = 256, NW = NT/64;
    constexpr int TILE = DPE == 6 ? 32 : 64, HALVES = 64/TILE, JW = TILE/NW, ITER = JW/HALVES;
    constexpr int R_BARY = 0, R_W = 3*NP, R_WPH = R_W+NP, R_PP = R_WPH+NP*DPE;
    // row strides of the per-cell LDS arrays: odd, so that lanes reading the rows of different cells hit different banks
    constexpr bool XPTS = DPE == 3;                      // a side: quadrature points (P1) or vertices (P2) in LDS
    constexpr int PS = XPTS ? NP*DIM+1 : NP*DIM, RS = XPTS ? (NP|1) : NP, SS = XPTS ? (DPE|1) : DPE, XS = XPTS ? PS : NC;
    // P1 (64 lanes = 64 cells): the data of the b-cell and its column sums TRAVEL from lane to lane (DPP wave rotation by one lane per
    // step) instead of being read from and added to LDS for every pair; the LDS pipe of a CU was 80-90 % busy with 24 instructions per
    // pair (12 ds_add_f64, the reads of a b-cell of its own per lane), the VALU 63 %.  P2 (two half-waves of 32 cells) reads per lane.
    constexpr bool TRAVEL = (TILE == 64);
    const pnl_const_f64_ptr rule = (pnl_const_f64_ptr)(unsigned long long)rule_g;
#ifndef PNL_DEBUG_ABLATE
    flags &= 1;                                          // the other bits skip work (debug builds only)
#endif
    extern __shared__ double smem[];
    double *s_y = smem;                                  // [TILE][PS] quadrature points of the b-cells
    double *s_x = s_y+TILE*PS;                           // [TILE][XS] quadrature points (P2: vertices) of the a-cells
    double *s_vola = s_x+TILE*XS;                        // [TILE]
    double *s_volb = s_vola+TILE;                        // [TILE]
    double *s_Ra = s_volb+TILE;                          // [TILE][NP] row sums of the a-cells, scaled
    double *s_Rb = s_Ra+TILE*RS;                         // [TILE][RS] column sums of the b-cells, scaled
    double *s_PP = s_Rb+TILE*RS;                         // [ND][NP] w phi_a phi_b at the points
    int *s_slotb = (int*)(s_PP+ND*NP);                   // [TILE][DPE] column of the sub-block (trash column nUe)
    int *s_sa = s_slotb+TILE*SS;                         // [TILE][SS] row offset in the sub-block (trash row nUe)
    int *s_ha = s_sa+TILE*SS;                            // [TILE] has-a-DoF flags
    int *s_hb = s_ha+TILE;
    int *s_dof = s_hb+TILE;                              // [2][2][nUe] global DoFs of both blocks (ping-pong over tiles);
    const bool dof_lists = SO.A2 == nullptr;             // not with the block-slot storage (uniform_fixed_lds)
    double *s_acc = (double*)(s_dof+(dof_lists ? 4*nUe : 0));    // [nUe+1][acc_stride]; nUe even
    // tables of the general power (KT == 0) behind the sub-block, if the launcher made room for them (bit 8 of flags: they
    // must not cost the second workgroup per CU)
    const bool have_pow = KT == 0 && (flags_in & 8) != 0;
    double *s_pow = s_acc+(size_t)(nUe+1)*acc_stride;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int li = lane%TILE, half = lane/TILE;
    const int wv = __builtin_amdgcn_readfirstlane(wave);
    unsigned long long npairs = 0;
    const double *cur_ptab = nullptr;                    // general power: whose tables s_pow holds

    // inputs of tile t -> LDS (b-side: threads [0, TILE), a-side: threads [TILE, 2 TILE), DoF lists: everybody)
    auto stage = [&](int t, int buf) {
        const int2 tl = tiles[t];
        const int nA = P.blk_ndof[tl.x], nB = P.blk_ndof[tl.y];
        if (tid < 2*TILE) {
            const bool bside = tid < TILE;
            const int l = bside ? tid : tid-TILE, c = (bside ? tl.y : tl.x)*TILE+l;
            double v[NC];
#pragma unroll
            for (int k = 0; k < NC; k++) v[k] = P.cellv[(size_t)k*P.ncp+c];
            const double vol = P.cvol[c];
            int sl[DPE], any = 0;
#pragma unroll
            for (int k = 0; k < DPE; k++) { sl[k] = P.cslot[(size_t)k*P.ncp+c]; any |= (sl[k] >= 0); }
            // bit 1: a real cell (zero-volume padding cells inside the mesh carry negative vertex ids: their pairs do not exist)
            any |= (P.cvid[c] >= 0) ? 2 : 0;
            // the points of both sides are formed here, once per cell and tile: as loaded values they stay in registers over the
            // pair loop (points recomputed from the vertices are rematerialised inside it when registers get tight)
            double *__restrict__ pts = (bside ? s_y : s_x)+l*PS;
            if (bside || XPTS) {
#pragma unroll
                for (int jp = 0; jp < NP; jp++)
#pragma unroll
                    for (int d = 0; d < DIM; d++) {
                        double sy = 0.;
#pragma unroll
                        for (int k = 0; k < NV; k++) sy = __builtin_fma(rule[R_BARY+3*jp+k], v[k*DIM+d], sy);
                        pts[jp*DIM+d] = sy;
                    }
            } else {
#pragma unroll
                for (int k = 0; k < NC; k++) s_x[l*XS+k] = v[k];
            }
            if (bside) {
                s_volb[l] = vol; s_hb[l] = any;
#pragma unroll
                for (int k = 0; k < DPE; k++) s_slotb[l*SS+k] = sl[k] >= 0 ? sl[k] : nUe;
            } else {
                s_vola[l] = vol; s_ha[l] = any;
#pragma unroll
                for (int k = 0; k < DPE; k++) s_sa[l*SS+k] = (sl[k] >= 0 ? sl[k] : nUe)*acc_stride;
            }
        }
        const int *__restrict__ dofA = P.blk_dofs+(size_t)tl.x*P.blk_stride;
        const int *__restrict__ dofB = P.blk_dofs+(size_t)tl.y*P.blk_stride;
        if (dof_lists) {
            for (int k = tid; k < nA; k += NT) s_dof[(buf*2+0)*nUe+k] = P.rowmap ? P.rowmap[dofA[k]] : dofA[k];
            for (int k = tid; k < nB; k += NT) s_dof[(buf*2+1)*nUe+k] = P.colmap ? P.colmap[dofB[k]] : dofB[k];
        }
        if (have_pow && tile_cls) {
            const double *pt = kcls[(tile_cls[t] & 0xffff) >> 1].ptab;
            if (pt != cur_ptab) { cur_ptab = pt; pnl_pow_tab_fill(s_pow, pt, tid, NT); }
        }
    };

    int tile_idx = blockIdx.x, buf = 0;
    if (tile_idx >= ntiles) return;
    for (int t = tid; t < ND*NP; t += NT) s_PP[t] = rule_g[R_PP+t];
    // tables of the general power: of P.k, or of the kernel class of the tile (copied when the class changes, by stage())
    if (have_pow && !tile_cls) { cur_ptab = P.k.ptab; pnl_pow_tab_fill(s_pow, cur_ptab, tid, NT); }
    for (int t = tid; t < (nUe+1)*acc_stride; t += NT) s_acc[t] = 0.;
    for (int t = tid; t < 2*TILE*RS; t += NT) s_Ra[t] = 0.;        // s_Ra and s_Rb
    stage(tile_idx, 0);
    lds_barrier();
#pragma unroll 1
    while (true) {
        const int2 tl = tiles[tile_idx];
        const int ta = tl.x, tb = tl.y;
        const int nA = P.blk_ndof[ta], nB = P.blk_ndof[tb];
        // variable order, piecewise constant: the tile's blocks carry one label each, the kernel of their class is tile-uniform
        DevKernel kk = P.k;
        if (tile_cls) kk = kcls[(tile_cls[tile_idx] & 0xffff) >> 1];
        const double scale2 = 2.*kern_scale<KT>(kk);
        const double *__restrict__ ptab = (have_pow && kk.ptab) ? s_pow : nullptr;
        double *__restrict__ Ra = s_Ra;
        // a side: lane = cell li (both halves of a P2 wave hold the same cells)
        double xa[NP == 3 ? 3 : 1][DIM];                 // P1, 3 points: in registers; 6 points: read per row of the rule
        double av[XPTS ? 1 : NC];                        // P2: the vertices, the points are formed per row of the rule
        if (XPTS && NP == 3) {
#pragma unroll
            for (int ip = 0; ip < (NP == 3 ? 3 : 1); ip++)
#pragma unroll
                for (int d = 0; d < DIM; d++) xa[ip][d] = s_x[li*XS+ip*DIM+d];
        }
        if (!XPTS) {
#pragma unroll
            for (int k = 0; k < NC; k++) av[XPTS ? 0 : k] = s_x[li*XS+k];
        }
        int sa[DPE];
#pragma unroll
        for (int k = 0; k < DPE; k++) sa[k] = s_sa[li*SS+k];
        const int fa = s_ha[li];
        const double vola = s_vola[li];
        kern_dispatch<KT>(kk, ptab, [&](auto ktag) {
        constexpr int KTE = decltype(ktag)::value;          // KT, or 3: the branch-free general power (pnl_common.h)
        // lane li meets cell (jb + li) mod TILE of block b at step jb: every lane of a group has a b-cell of its own, so the column
        // sums go to s_Rb with conflict-free ds_add_f64 (a reduction over the lanes cost 46 of the 302 instructions per pair of the
        // 3-point kernel), and two lanes add to the same entry of the sub-block only if their cells share the row DoF AND the column DoF
        double racc[NP == 3 ? 3 : 1] = {};
        // the b-cell this lane currently meets: index, points, flags, volume, columns of its DoFs, travelling column sums
        int j = (wave*JW+half+li) & (TILE-1);
        double y[NP][DIM], volb_j, cacc[TRAVEL ? NP : 1] = {};
        int fb, sbj[DPE];
        auto load_b = [&]() {
#pragma unroll
            for (int jp = 0; jp < NP; jp++)
#pragma unroll
                for (int d = 0; d < DIM; d++) y[jp][d] = s_y[j*PS+jp*DIM+d];
            fb = s_hb[j];
            volb_j = s_volb[j];
#pragma unroll
            for (int b = 0; b < DPE; b++) sbj[b] = s_slotb[j*SS+b];
        };
        if (TRAVEL) load_b();
#pragma unroll 1
        for (int jj = 0; jj < ITER; jj++) {
            if (!TRAVEL) { j = (wave*JW+jj*HALVES+half+li) & (TILE-1); load_b(); }
            else if (jj) {
                // one lane on: whichever way the rotation goes, the wave meets 16 consecutive cells of block b per lane, and the index
                // travels with the data
                constexpr int ROT = 0x134;                 // wave_rol:1
                j = __builtin_amdgcn_update_dpp(0, j, ROT, 0xf, 0xf, true);
                fb = __builtin_amdgcn_update_dpp(0, fb, ROT, 0xf, 0xf, true);
                volb_j = dpp_get<ROT>(volb_j);
#pragma unroll
                for (int b = 0; b < DPE; b++) sbj[b] = __builtin_amdgcn_update_dpp(0, sbj[b], ROT, 0xf, 0xf, true);
#pragma unroll
                for (int jp = 0; jp < NP; jp++) {
                    cacc[TRAVEL ? jp : 0] = dpp_get<ROT>(cacc[TRAVEL ? jp : 0]);
#pragma unroll
                    for (int d = 0; d < DIM; d++) y[jp][d] = dpp_get<ROT>(y[jp][d]);
                }
            }
            // NA:138-150: pairs with boundary DoFs only are skipped; pairs that hold a padding cell do not exist
            const bool valid = (((fa | fb) & 1) != 0) && (((fa & fb) & 2) != 0);
            npairs += (unsigned long long)__popcll(__ballot(valid));
            const double volb = valid ? volb_j : 0.;
            // NA:1405-1410: symmetric cell pairs count twice
            const double vv = scale2*vola*volb;
            static_assert(!STRUCT || (DPE == 3 && TRAVEL && XPTS), "the structured rules: P1");
            if constexpr (STRUCT && NP == 3) {
                double g[3][3];
#pragma unroll
                for (int a = 0; a < 3; a++)
#pragma unroll
                    for (int b = 0; b < 3; b++) {
                        double d2 = 0.;
#pragma unroll
                        for (int d = 0; d < DIM; d++) { const double t = xa[a][d]-y[b][d]; d2 = __builtin_fma(t, t, d2); }
                        g[a][b] = kern_eval<KTE>(kk, d2, ptab);
                    }
                double sr[3], sc[3];
#pragma unroll
                for (int a = 0; a < 3; a++) { sr[a] = (g[a][0]+g[a][1])+g[a][2]; sc[a] = (g[0][a]+g[1][a])+g[2][a]; }
                const double S = (sr[0]+sr[1])+sr[2];
                const double w = rule[R_W], A0 = rule[R_WPH+1], B0 = rule[R_WPH]-rule[R_WPH+1];
                const double vvw = vv*w;
#pragma unroll
                for (int a = 0; a < 3; a++) {
                    racc[a] = __builtin_fma(vvw, sr[a], racc[a]);
                    cacc[a] = __builtin_fma(vvw, sc[a], cacc[a]);
                }
                if (!(flags & 4)) {
                    const double k2 = -vv*(B0*B0), k1 = -vv*(A0*B0), T = (-vv*(A0*A0))*S;
                    double ra[3], cb[3];
#pragma unroll
                    for (int a = 0; a < 3; a++) { ra[a] = __builtin_fma(k1, sr[a], T); cb[a] = k1*sc[a]; }
#pragma unroll
                    for (int b = 0; b < 3; b++)
#pragma unroll
                        for (int a = 0; a < 3; a++) lds_add_f64(&s_acc[sa[a]+sbj[b]], __builtin_fma(k2, g[a][b], ra[a]+cb[b]));
                } else if (S == 1.2345e300) s_acc[0] = vv;
                continue;
            }
            double c[NP], G[DPE][DPE];
#pragma unroll
            for (int jp = 0; jp < NP; jp++) c[jp] = 0.;
#pragma unroll
            for (int a = 0; a < DPE; a++)
#pragma unroll
                for (int b = 0; b < DPE; b++) G[a][b] = 0.;
            // rows of the tensor rule: fully unrolled for 3 points; for 6 points the loop stays rolled (its body already holds
            // 36 independent chains) and everything indexed by ip is recomputed or read through the scalar cache
#pragma unroll (NP == 3 ? 3 : 1)
            for (int ip = 0; ip < NP; ip++) {
                double x[DIM];
#pragma unroll
                for (int d = 0; d < DIM; d++) {
                    if (!XPTS) {
                        double sx = 0.;
#pragma unroll
                        for (int k = 0; k < NV; k++) sx = __builtin_fma(rule[R_BARY+3*ip+k], av[XPTS ? 0 : k*DIM+d], sx);
                        x[d] = sx;
                    } else x[d] = NP == 3 ? xa[NP == 3 ? ip : 0][d] : s_x[li*XS+ip*DIM+d];
                }
                const double wi = rule[R_W+ip];
                double r = 0., u[DPE];
#pragma unroll
                for (int b = 0; b < DPE; b++) u[b] = 0.;
                if constexpr (STRUCT && NP == 6) {
                    double g[NP];
#pragma unroll
                    for (int jp = 0; jp < NP; jp++) {
                        double d2 = 0.;
#pragma unroll
                        for (int d = 0; d < DIM; d++) { const double t = x[d]-y[jp][d]; d2 = __builtin_fma(t, t, d2); }
                        g[jp] = kern_eval<KTE>(kk, d2, ptab);
                        c[jp] = __builtin_fma(wi, g[jp], c[jp]);
                    }
                    const double s0 = (g[0]+g[1])+g[2], s1 = (g[3]+g[4])+g[NP-1];
                    const double A0 = rule[R_WPH+1], B0 = rule[R_WPH]-rule[R_WPH+1];
                    const double A1 = rule[R_WPH+3*DPE+1], B1 = rule[R_WPH+3*DPE]-rule[R_WPH+3*DPE+1];
                    r = __builtin_fma(rule[R_W], s0, rule[R_W+3]*s1);
                    const double base = __builtin_fma(A0, s0, A1*s1);
#pragma unroll
                    for (int b = 0; b < DPE; b++) u[b] = __builtin_fma(B0, g[b], __builtin_fma(B1, g[NP == 6 ? 3+b : 0], base));
                } else {
#pragma unroll
                for (int jp = 0; jp < NP; jp++) {
                    double d2 = 0.;
#pragma unroll
                    for (int d = 0; d < DIM; d++) { const double t = x[d]-y[jp][d]; d2 = __builtin_fma(t, t, d2); }
                    const double g = kern_eval<KTE>(kk, d2, ptab);
                    r = __builtin_fma(rule[R_W+jp], g, r);
                    c[jp] = __builtin_fma(wi, g, c[jp]);
#pragma unroll
                    for (int b = 0; b < DPE; b++) u[b] = __builtin_fma(g, rule[R_WPH+jp*DPE+b], u[b]);
                }
                }
                if (NP == 3) racc[NP == 3 ? ip : 0] = __builtin_fma(vv, r, racc[NP == 3 ? ip : 0]);
                else lds_add_f64(&Ra[li*RS+ip], vv*r);
#pragma unroll
                for (int a = 0; a < DPE; a++) {
                    const double pa = rule[R_WPH+ip*DPE+a];
#pragma unroll
                    for (int b = 0; b < DPE; b++) G[a][b] = __builtin_fma(pa, u[b], G[a][b]);
                }
            }
            // cross block -> LDS sub-block of A'
            if (!(flags & 4)) {
#pragma unroll
                for (int b = 0; b < DPE; b++) {
                    const int sb = sbj[b];
#pragma unroll
                    for (int a = 0; a < DPE; a++) lds_add_f64(&s_acc[sa[a]+sb], -vv*G[a][b]);
                }
            } else if (G[0][0]+G[1][2]+G[DPE-1][DPE-1] == 1.2345e300) s_acc[0] = vv;
            // scaled column sums of cell j
#pragma unroll
            for (int jp = 0; jp < NP; jp++) {
                if (TRAVEL) cacc[TRAVEL ? jp : 0] = __builtin_fma(vv, c[jp], cacc[TRAVEL ? jp : 0]);
                else lds_add_f64(&s_Rb[j*RS+jp], vv*c[jp]);
            }
        }
        if (TRAVEL) {
            // the sums that travelled through this wave's lanes, now at the lane that met cell j last
#pragma unroll
            for (int jp = 0; jp < NP; jp++) lds_add_f64(&s_Rb[j*RS+jp], cacc[TRAVEL ? jp : 0]);
        }
        if (NP == 3) {
#pragma unroll
            for (int ip = 0; ip < (NP == 3 ? 3 : 1); ip++) lds_add_f64(&Ra[li*RS+ip], racc[ip]);
        }
        });
        lds_barrier();
        // ---- the next tile's inputs, then the flush of this one ----
        const int nxt = tile_idx+(int)gridDim.x;
        const bool more = nxt < ntiles;
        if (more) stage(nxt, buf^1);
        const int *__restrict__ dA = s_dof+(buf*2+0)*nUe, *__restrict__ dB = s_dof+(buf*2+1)*nUe;
        const bool sym = (flags & 1) != 0 && dof_lists;
        if (SO.A2) {
            // block-slot storage: this tile owns its nA x nB sub-block, plain 16-byte stores of every entry
            // (writing the padded width, whole 64-byte lines, was measured: no less traffic, the fold 1 ms slower)
            const int ca = SO.colbase[ta], W = SO.S-ca;
            double *__restrict__ base = SO.A2+SO.rowoff[ta]+(SO.colbase[tb]-ca);
#pragma unroll 1
            for (int r = wv; r < nA; r += NW) {
                double *__restrict__ row = base+(long long)r*W;
                for (int cc = 2*lane; cc < nB; cc += 128) {
                    double2 v = make_double2(0., 0.);
                    if (cc < nB) { v.x = s_acc[r*acc_stride+cc]; s_acc[r*acc_stride+cc] = 0.; }
                    if (cc+1 < nB) { v.y = s_acc[r*acc_stride+cc+1]; s_acc[r*acc_stride+cc+1] = 0.; }
                    slot_store2(row+cc, v.x, v.y);
                }
            }
        } else if (!(flags & 2))
#pragma unroll 1
        for (int r = wv; r < nA; r += NW) {
            double *__restrict__ row = A+(long long)__builtin_amdgcn_readfirstlane(dA[r])*ldA;
            for (int cc = lane; cc < nB; cc += 64) {
                const double v = s_acc[r*acc_stride+cc];
                if (v != 0.) {
                    if (!sym) s_acc[r*acc_stride+cc] = 0.;
                    atomic_add_f64(&row[dB[cc]], v);
                }
            }
        }
        // PNL_FLAG_SYMMETRIC_FLUSH (no mirror pass): the transposed image, lanes along the row of A again; it zeroes the
        // sub-block, so every wave must have finished the first sweep
        if (sym) lds_barrier();
        if (sym)
#pragma unroll 1
            for (int cc = wv; cc < nB; cc += NW) {
                double *__restrict__ row = A+(long long)__builtin_amdgcn_readfirstlane(dB[cc])*ldA;
                for (int r = lane; r < nA; r += 64) {
                    const double v = s_acc[r*acc_stride+cc];
                    if (v != 0.) {
                        s_acc[r*acc_stride+cc] = 0.;
                        atomic_add_f64(&row[dA[r]], v);
                    }
                }
            }
        if (flags & 2) for (int t = tid; t < (nUe+1)*acc_stride; t += NT) s_acc[t] = 0.;
        // diagonal blocks of both sides from the scaled row / column sums: PARTS neighbouring lanes share a (side, cell), read its sums
        // with one instruction and zero them with a later one (LDS operations of a wave complete in order), each takes every
        // PARTS-th entry
        {
            constexpr int PARTS = NT/(2*TILE);
            static_assert(PARTS*2*TILE == NT && 64%PARTS == 0, "one (side, cell) per group of PARTS lanes");
            const int item = tid/PARTS, part = tid-item*PARTS, side = item/TILE, cl = item-side*TILE;
            double *__restrict__ R = (side ? s_Rb : s_Ra)+cl*RS;
            double Rv[NP];
#pragma unroll
            for (int ip = 0; ip < NP; ip++) Rv[ip] = R[ip];
#pragma unroll
            for (int ip = 0; ip < NP; ip++) if (ip%PARTS == part) R[ip] = 0.;
#pragma unroll 1
            for (int e = part; e < ND; e += PARTS) {
                double v = 0.;
#pragma unroll
                for (int ip = 0; ip < NP; ip++) v = __builtin_fma(s_PP[e*NP+ip], Rv[ip], v);
                if (v != 0.) atomic_add_f64(&Dglob[(size_t)((side ? tb : ta)*TILE+cl)*ND+e], v);
            }
        }
        if (!more) break;
        lds_barrier();
        tile_idx = nxt; buf ^= 1;
    }
    // statistics: every lane of a wave holds the same count
    if (lane == 0 && npairs) {
        atomicAdd(&P.counters[8+q_uniform], npairs);
        atomicAdd(&P.counters[1], npairs);
        atomicAdd(&P.counters[2], npairs*(unsigned long long)(NP*NP));
        atomicAdd(&P.counters[6], npairs);
        atomicAdd(&P.counters[131+q_uniform-2], npairs);
    }
}

// =====================================================================================================================
// General P2 tile kernel
#define P2_TILE 32
#define P2_NT 512
#define P2_MAXPTS 96
#define P2_MAXCHUNKS 96
struct P2Smem {
    static constexpr int TILE = P2_TILE, NV = 3, NC = 6, DPE = 6, ND = 21, PAIRS = TILE*TILE, ST = 4+DPE;
    static constexpr int NR = 15;                          // row / column sums per (side, cell): the 3-point rule, the first and
                                                           // the second 6-point rule (orders 2, 3, 4 on triangles)
    static constexpr int MAXLAB = 16;                      // label tables up to MAXLAB x MAXLAB are kept in LDS
    // doubles
    static constexpr int o_v = 0;                          // [2][NC][TILE]
    static constexpr int o_cen = o_v+2*NC*TILE;            // [2][2][TILE]
    static constexpr int o_vol = o_cen+4*TILE;             // [2][TILE]
    static constexpr int o_h = o_vol+2*TILE;               // [2][TILE]
    static constexpr int o_Ld = o_h+2*TILE;                // [2][TILE]
    static constexpr int o_D = o_Ld+2*TILE;                // [2 buffers][2][TILE][ND] contributions formed per pair (list C)
    static constexpr int o_R = o_D+4*TILE*ND;              // [2 buffers][2][TILE][NR]
    static constexpr int o_PP = o_R+4*TILE*NR;             // [ND][NR] (+ pad): w phi_a phi_b at the points of those rules
    static constexpr int o_tt = o_PP+ND*NR+1;              // [P2_MAXPTS][ST]
    static constexpr int o_pow = o_tt+P2_MAXPTS*ST;        // [PNL_POW_TAB_DOUBLES] tables of the general power (KT == 0)
    static constexpr int n_dbl = o_pow+PNL_POW_TAB_DOUBLES;
    // ints
    static constexpr int o_vid = 0;                        // [2][NV][TILE]
    static constexpr int o_cnt = o_vid+2*NV*TILE;          // [PNL_MAXQ+2]
    static constexpr int o_cur = o_cnt+PNL_MAXQ+2;         // [PNL_MAXQ+2]
    static constexpr int o_misc = o_cur+PNL_MAXQ+2;        // [8]: |A|, |B|, |far|, |C|, #C chunks, next chunk, tile after next, wl base
    static constexpr int o_lh = o_misc+8;                  // float [2][2][TILE]
    static constexpr int o_ttn = o_lh+4*TILE;              // [PNL_MAXQ+2]
    static constexpr int o_tto = o_ttn+PNL_MAXQ+2;         // [PNL_MAXQ+2]
    static constexpr int o_off = o_tto+PNL_MAXQ+2;         // [PNL_MAXQ+2] offsets of the full rule tables (work-list entries)
    static constexpr int o_chunk = o_off+PNL_MAXQ+2;       // [P2_MAXCHUNKS] list C chunks: order << 20 | start << 7 | count-1
    static constexpr int o_lab = o_chunk+P2_MAXCHUNKS;     // [2][TILE] labels of the cells (variable order)
    static constexpr int o_clsof = o_lab+2*TILE;           // [MAXLAB*MAXLAB] class of a label pair
    static constexpr int o_l32 = o_clsof+MAXLAB*MAXLAB;    // [PAIRS] list B from the front, far list from the back
    static constexpr int n_int = o_l32+PAIRS;              // then [2 buffers][2][nUe] global DoFs of both blocks
    // shorts
    static constexpr int o_slot = 0;                       // [2][DPE][TILE]
    static constexpr int o_list = o_slot+2*DPE*TILE;       // [PAIRS] list A from the front, list C from the back
    static constexpr int o_csort = o_list+PAIRS;           // [PAIRS] list C sorted by order
    static constexpr int n_short = o_csort+PAIRS;
    __host__ __device__ static constexpr size_t fixed_bytes(int nUe) {
        return sizeof(double)*n_dbl+sizeof(int)*(size_t)(n_int+4*nUe)+sizeof(short)*n_short;
    }
};
static_assert(P2Smem::n_int % 2 == 0 && P2Smem::n_short % 4 == 0, "LDS regions must keep 8-byte alignment");

__device__ __forceinline__ int p2_wave_bucket_add(int *counters, int q) {
    unsigned long long todo = __ballot(q > 0);
    while (todo) {
        const int leader = __ffsll((long long)todo)-1;
        const int qL = __builtin_amdgcn_readlane(q, leader);
        const unsigned long long same = __ballot(q == qL);
        if ((int)(threadIdx.x & 63) == leader) atomicAdd(&counters[qL], __popcll(same));
        todo &= ~same;
    }
    return 0;
}

// one pair per lane, N points per triangle known at compile time: cross block G, column sums c; the row sums go straight to
// the per-(cell, point) LDS buffer Ra (scaled by vv).  gw: per point w, w phi[0..5] through the scalar cache; tab: the LDS
// copy of the rule (per point bary[3], w, phi[6]) for everything indexed by the rolled loop variable i.
template <int KT, int N>
__device__ __forceinline__ void p2_eval_fixed(const DevKernel &kk, const double *__restrict__ tab, const double *gw_global,
                                              const double *av, const double *bv, double vv, bool act, double *Ra,
                                              double (&G)[6][6], double (&c)[N], const double *__restrict__ ptab) {
    constexpr int ST = 10, GS = 7;
    const pnl_const_f64_ptr gw = (pnl_const_f64_ptr)(unsigned long long)gw_global;
    double y[N][2];
#pragma unroll
    for (int j = 0; j < N; j++) {
        c[j] = 0.;
#pragma unroll
        for (int d = 0; d < 2; d++) {
            double s = 0.;
#pragma unroll
            for (int k = 0; k < 3; k++) s = __builtin_fma(tab[j*ST+k], bv[k*2+d], s);
            y[j][d] = s;
        }
    }
#pragma unroll 1
    for (int i = 0; i < N; i++) {
        double x[2];
#pragma unroll
        for (int d = 0; d < 2; d++) {
            double s = 0.;
#pragma unroll
            for (int k = 0; k < 3; k++) s = __builtin_fma(tab[i*ST+k], av[k*2+d], s);
            x[d] = s;
        }
        const double wi = tab[i*ST+3];
        double r = 0., u[6];
#pragma unroll
        for (int b = 0; b < 6; b++) u[b] = 0.;
#pragma unroll
        for (int j = 0; j < N; j++) {
            double d2 = 0.;
#pragma unroll
            for (int d = 0; d < 2; d++) { const double t = x[d]-y[j][d]; d2 = __builtin_fma(t, t, d2); }
            const double g = kern_eval<KT>(kk, d2, ptab);
            r = __builtin_fma(gw[j*GS], g, r);
            c[j] = __builtin_fma(wi, g, c[j]);
#pragma unroll
            for (int b = 0; b < 6; b++) u[b] = __builtin_fma(g, gw[j*GS+1+b], u[b]);
        }
        if (act) lds_add_f64(&Ra[i], vv*r);
#pragma unroll
        for (int a = 0; a < 6; a++) {
            const double pa = wi*tab[i*ST+4+a];
#pragma unroll
            for (int b = 0; b < 6; b++) G[a][b] = __builtin_fma(pa, u[b], G[a][b]);
        }
    }
}

// runtime number of points (list C: orders with 7-16 points, a few per cent of the pairs), first sweep: G and S1
template <int KT>
__device__ __forceinline__ void p2_eval_lds_sweep1(const DevKernel &kk, const double *__restrict__ tab, int n, const double *av,
                                                   const double *bv, double (&G)[6][6], double (&S)[21], const double *__restrict__ ptab) {
    constexpr int ST = 10;
#pragma unroll 1
    for (int i = 0; i < n; i++) {
        const double *__restrict__ ti = tab+i*ST;
        double x[2];
#pragma unroll
        for (int d = 0; d < 2; d++) {
            double s = 0.;
#pragma unroll
            for (int k = 0; k < 3; k++) s = __builtin_fma(ti[k], av[k*2+d], s);
            x[d] = s;
        }
        const double wi = ti[3];
        double r = 0., u[6];
#pragma unroll
        for (int b = 0; b < 6; b++) u[b] = 0.;
#pragma unroll 2
        for (int j = 0; j < n; j++) {
            const double *__restrict__ tj = tab+j*ST;
            double d2 = 0.;
#pragma unroll
            for (int d = 0; d < 2; d++) {
                double sy = 0.;
#pragma unroll
                for (int k = 0; k < 3; k++) sy = __builtin_fma(tj[k], bv[k*2+d], sy);
                const double t = x[d]-sy;
                d2 = __builtin_fma(t, t, d2);
            }
            const double K = (wi*tj[3])*kern_eval<KT>(kk, d2, ptab);
            r += K;
#pragma unroll
            for (int b = 0; b < 6; b++) u[b] = __builtin_fma(K, tj[4+b], u[b]);
        }
        int e = 0;
#pragma unroll
        for (int a = 0; a < 6; a++) {
            const double pa = ti[4+a];
#pragma unroll
            for (int b = 0; b < 6; b++) G[a][b] = __builtin_fma(pa, u[b], G[a][b]);
            const double pr = pa*r;
#pragma unroll
            for (int b = a; b < 6; b++) { S[e] = __builtin_fma(pr, ti[4+b], S[e]); e++; }
        }
    }
}

// second sweep: the column sums c_j = sum_i w_i w_j g_ij (the kernel values once more: cheaper than 21 FMAs per point pair) -> S2
template <int KT>
__device__ __forceinline__ void p2_eval_lds_sweep2(const DevKernel &kk, const double *__restrict__ tab, int n, const double *av,
                                                   const double *bv, double (&S)[21], const double *__restrict__ ptab) {
    constexpr int ST = 10;
#pragma unroll 1
    for (int j = 0; j < n; j++) {
        const double *__restrict__ tj = tab+j*ST;
        double y[2];
#pragma unroll
        for (int d = 0; d < 2; d++) {
            double s = 0.;
#pragma unroll
            for (int k = 0; k < 3; k++) s = __builtin_fma(tj[k], bv[k*2+d], s);
            y[d] = s;
        }
        double cc = 0.;
#pragma unroll 2
        for (int i = 0; i < n; i++) {
            const double *__restrict__ ti = tab+i*ST;
            double d2 = 0.;
#pragma unroll
            for (int d = 0; d < 2; d++) {
                double sx = 0.;
#pragma unroll
                for (int k = 0; k < 3; k++) sx = __builtin_fma(ti[k], av[k*2+d], sx);
                const double t = sx-y[d];
                d2 = __builtin_fma(t, t, d2);
            }
            cc = __builtin_fma(ti[3], kern_eval<KT>(kk, d2, ptab), cc);
        }
        cc *= tj[3];
        int e = 0;
#pragma unroll
        for (int a = 0; a < 6; a++) {
            const double pc = tj[4+a]*cc;
#pragma unroll
            for (int b = a; b < 6; b++) { S[e] = __builtin_fma(pc, tj[4+b], S[e]); e++; }
        }
    }
}

// Software pipeline over the tiles of a workgroup like k_tile_uniform: the cell data of the next tile is staged BEFORE the
// flush of the current one is issued, the barriers order the LDS only, the flush takes the DoF numbers from LDS and zeroes
// the sub-block as it reads it.  Tile indices come from a global counter two tiles ahead (tickets).
template <int KT>
__global__ void __launch_bounds__(P2_NT, 2)
k_tile_p2(const DevProblem P, const int2 *__restrict__ tiles, const int *__restrict__ tile_cls, const DevKernel *__restrict__ kcls,
          const DevFormula *__restrict__ fcls, double *__restrict__ A, long long ldA, double *__restrict__ Dglob, int cell_begin,
          int cell_end, int acc_stride, int4 *__restrict__ worklist, unsigned *__restrict__ wl_count, unsigned wl_cap, int flags,
          int ntiles, unsigned *__restrict__ tile_ctr, int nUe, const SlotOut SO) {
    using S = P2Smem;
    constexpr int TILE = S::TILE, NV = 3, NC = 6, DPE = 6, ND = 21, NT = P2_NT, PAIRS = S::PAIRS, PER_THREAD = PAIRS/NT, ST = S::ST;
    constexpr int NA = 3, NB = 6, NWAVES = NT/64, NR = S::NR, MAXLAB = S::MAXLAB;
    extern __shared__ double smem[];
    double *s_dbl = smem;
    int *s_int = (int*)(s_dbl+S::n_dbl);
    int *s_dof = s_int+S::n_int;                            // [2][2][nUe]
    short *s_short = (short*)(s_dof+4*nUe);
    double *s_acc = (double*)(s_short+S::n_short);          // [nUe+1][acc_stride]
    double *s_v = s_dbl+S::o_v, *s_cen = s_dbl+S::o_cen, *s_vol = s_dbl+S::o_vol, *s_h = s_dbl+S::o_h, *s_Ld = s_dbl+S::o_Ld;
    double *s_Dall = s_dbl+S::o_D, *s_Rall = s_dbl+S::o_R, *s_PP = s_dbl+S::o_PP, *s_tt = s_dbl+S::o_tt;
    int *s_vid = s_int+S::o_vid, *s_cnt = s_int+S::o_cnt, *s_cur = s_int+S::o_cur, *s_misc = s_int+S::o_misc;
    float *s_lh = (float*)(s_int+S::o_lh);
    int *s_ttn = s_int+S::o_ttn, *s_tto = s_int+S::o_tto, *s_off = s_int+S::o_off, *s_chunk = s_int+S::o_chunk, *s_l32 = s_int+S::o_l32;
    int *s_lab = s_int+S::o_lab, *s_clsof = s_int+S::o_clsof;
    short *s_slot = s_short+S::o_slot;
    unsigned short *s_list = (unsigned short*)(s_short+S::o_list), *s_csort = (unsigned short*)(s_short+S::o_csort);

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wv = __builtin_amdgcn_readfirstlane(wave);
    const unsigned long long lt = (1ull << lane)-1ull;
    unsigned long long st_cnt = 0, st_ev = 0;                // statistics of order q = 2 + tid, kept over all tiles
    const bool lab_lds = P.nlab <= MAXLAB;
    for (int t = tid; t < PNL_MAXQ+2; t += NT) { s_ttn[t] = P.tt_n[t]; s_tto[t] = P.tt_off[t]; s_off[t] = t <= P.qmax+1 ? P.off[t] : 0; }
    for (int t = tid; t < P.tt_npts*ST; t += NT) s_tt[t] = P.tt_tab[t];
    // tables of the general power: of P.k, or of the kernel class of the tile visit (copied when the class changes, by stage())
    double *s_pow = s_dbl+S::o_pow;
    const double *cur_ptab = nullptr;
    if (KT == 0 && !tile_cls) { cur_ptab = P.k.ptab; pnl_pow_tab_fill(s_pow, cur_ptab, tid, NT); }
    if (P.nlab > 0 && lab_lds) for (int t = tid; t < P.nlab*P.nlab; t += NT) s_clsof[t] = P.cls_of[t];
    // the rules evaluated by the unrolled evaluators with row / column sums: the lowest order with 3 points and the two lowest
    // orders with 6 points (every such order has its own points, hence its own sums)
    int qA0 = 0, qB0 = 0, qB1 = 0;
    for (int q = 17; q >= 2; q--) {
        const int n = P.tt_n[q];
        qA0 = (n == NA) ? q : qA0;
        if (n == NB) { qB1 = qB0; qB0 = q; }
    }
    const bool symflush = (flags & 256) != 0;

    // cell data of both blocks of tile t -> LDS (64 threads), DoF lists -> buffer buf (everybody)
    auto stage = [&](int t, int buf) {
        const int ta = tiles[t].x, tb = tiles[t].y;
        const int nA = P.blk_ndof[ta], nB = P.blk_ndof[tb];
        if (tid < 2*TILE) {
            const int side = tid/TILE, l = tid%TILE;
            const int c = (side ? tb : ta)*TILE+l;
#pragma unroll
            for (int k = 0; k < NC; k++) s_v[(side*NC+k)*TILE+l] = P.cellv[(size_t)k*P.ncp+c];
#pragma unroll
            for (int d = 0; d < 2; d++) s_cen[(side*2+d)*TILE+l] = P.ccen[(size_t)d*P.ncp+c];
            s_vol[side*TILE+l] = P.cvol[c];
            s_h[side*TILE+l] = P.ch[c];
            const double lh = P.clog[c], Ld = P.clog[(size_t)P.ncp+c];
            s_Ld[side*TILE+l] = Ld;
            s_lh[(side*2+0)*TILE+l] = (float)lh;
            s_lh[(side*2+1)*TILE+l] = (float)Ld;
#pragma unroll
            for (int k = 0; k < NV; k++) s_vid[(side*NV+k)*TILE+l] = P.cvid[(size_t)k*P.ncp+c];
#pragma unroll
            for (int k = 0; k < DPE; k++) {
                // boundary DoFs (no slot) are sent to the trash row / column nUe of the LDS sub-block: no branches in the hot loop
                const short sl = P.cslot[(size_t)k*P.ncp+c];
                s_slot[(side*DPE+k)*TILE+l] = sl >= 0 ? sl : (short)nUe;
            }
            if (P.nlab > 0) s_lab[side*TILE+l] = P.clabel[c];
        }
        const int *__restrict__ dofA = P.blk_dofs+(size_t)ta*P.blk_stride;
        const int *__restrict__ dofB = P.blk_dofs+(size_t)tb*P.blk_stride;
        for (int k = tid; k < nA; k += NT) s_dof[(buf*2+0)*nUe+k] = P.rowmap ? P.rowmap[dofA[k]] : dofA[k];
        for (int k = tid; k < nB; k += NT) s_dof[(buf*2+1)*nUe+k] = P.colmap ? P.colmap[dofB[k]] : dofB[k];
        if (KT == 0 && tile_cls && !(tile_cls[t] & (1 << 29))) {
            const double *pt = kcls[(tile_cls[t] & 0xffff) >> 1].ptab;
            if (pt != cur_ptab) { cur_ptab = pt; pnl_pow_tab_fill(s_pow, pt, tid, NT); }
        }
    };

    int n_cur = blockIdx.x, n_nxt = (int)(gridDim.x+blockIdx.x), buf = 0;
    if (n_cur >= ntiles) return;
    unsigned pending = 0;                                    // thread 0: ticket of the tile after n_nxt
    if (tid == 0) pending = atomicAdd(tile_ctr, 1u);
    for (int t = tid; t < (nUe+1)*acc_stride; t += NT) s_acc[t] = 0.;
    for (int t = tid; t < 4*TILE*(ND+NR); t += NT) s_Dall[t] = 0.;             // s_D and s_R (both buffers) are adjacent
    for (int t = tid; t < 2*(PNL_MAXQ+2)+6; t += NT) s_cnt[t] = 0;             // s_cnt, s_cur and s_misc[0..5] are adjacent
    stage(n_cur, 0);
    lds_barrier();
    // w phi_a phi_b at the points of the unrolled rules
    for (int t = tid; t < ND*NR; t += NT) {
        const int e = t/NR, k = t-e*NR;
        const int q = k < 3 ? qA0 : (k < 9 ? qB0 : qB1), i = k < 3 ? k : (k < 9 ? k-3 : k-9);
        int a = 0, idx = e;
        while (idx >= DPE-a) { idx -= DPE-a; a++; }
        const int b = a+idx;
        const double *tp = s_tt+(s_tto[q]+i)*ST;
        s_PP[t] = q ? tp[3]*tp[4+a]*tp[4+b] : 0.;
    }
#pragma unroll 1
    while (true) {
    const int tile_idx = n_cur;
    const int ta = tiles[tile_idx].x, tb = tiles[tile_idx].y;
    const int nA = P.blk_ndof[ta], nB = P.blk_ndof[tb];
    // order class of this tile entry (variable order: kernel, order formula, work-list region of the class; bit 0: orientation).
    // Bit 29 of the class word: the tile holds pairs of SEVERAL classes, bits 0..27 are the set of them -- the classes are
    // worked through one after the other (classification, lists, evaluation per class) into the same LDS sub-block and sums,
    // ONE flush with plain stores at the end, instead of one visit per class with an atomic flush each
    const int cword = tile_cls ? tile_cls[tile_idx] : 0;
    const bool multik = tile_cls && (cword & (1 << 29)) != 0;
    unsigned cmask = multik ? ((unsigned)cword & 0x0fffffffu) : 0u;
    double *__restrict__ s_D = s_Dall+buf*2*TILE*ND, *__restrict__ s_R = s_Rall+buf*2*TILE*NR;
    bool first_class = true;
#pragma unroll 1
    do {
    int tcls;
    if (multik) { const int kc = __ffs((int)cmask)-1; cmask &= cmask-1u; tcls = 2*kc; }
    else tcls = tile_cls ? (cword & 0xffff) : (P.cur_class >= 0 ? 2*P.cur_class+P.orient : -1);
    DevKernel kk = P.k;
    DevFormula qo = P.qo;
    if (tile_cls) { kk = kcls[tcls >> 1]; qo = fcls[tcls >> 1]; }
    const int wl_region = tile_cls ? (tcls >> 1) : 0;
    if (!first_class) {
        // the lists and counters of the previous class are done with (barrier behind its evaluation)
        for (int t = tid; t < 2*(PNL_MAXQ+2)+6; t += NT) s_cnt[t] = 0;
    }
    if (KT == 0 && multik && kk.ptab != cur_ptab) { cur_ptab = kk.ptab; pnl_pow_tab_fill(s_pow, cur_ptab, tid, NT); }
    if (!first_class) lds_barrier();
    first_class = false;

    // ---- classification (NO:280-378 vertex test, NO:493-540 + FL2:622-642 order) ----
    int overflow = 0;
    int cnt234[3] = {0, 0, 0};
#pragma unroll
    for (int it = 0; it < PER_THREAD; it++) {
        const int p = it*NT+tid;
        const int j = p%TILE, i = (p/TILE+21*j)%TILE;
        int q = 0;
        const int va0 = s_vid[(0*NV+0)*TILE+i], vb0 = s_vid[(1*NV+0)*TILE+j];
        const int ca = ta*TILE+i;
        bool ok = (va0 >= 0) && (vb0 >= 0) && (ta < tb || i < j) && (ca >= cell_begin) && (ca < cell_end);
        if (tcls >= 0 && ok) {
            const int la = s_lab[i], lb = s_lab[TILE+j];
            const int idx = (tcls & 1) ? lb*P.nlab+la : la*P.nlab+lb;
            ok = (lab_lds ? s_clsof[idx] : P.cls_of[idx]) == (tcls >> 1);
        }
        if (ok) {
            bool any_dof = false, shared = false;
#pragma unroll
            for (int k = 0; k < DPE; k++)
                any_dof = any_dof || (s_slot[(0*DPE+k)*TILE+i] < nUe) || (s_slot[(1*DPE+k)*TILE+j] < nUe);
#pragma unroll
            for (int k = 0; k < NV; k++) {
                const int va = s_vid[(0*NV+k)*TILE+i];
#pragma unroll
                for (int m = 0; m < NV; m++) shared = shared || (va == s_vid[(1*NV+m)*TILE+j]);
            }
            if (any_dof && !shared) {
                double d2 = 0.;
#pragma unroll
                for (int d = 0; d < 2; d++) {
                    const double t = s_cen[(0*2+d)*TILE+i]-s_cen[(1*2+d)*TILE+j];
                    d2 += t*t;
                }
                q = quad_order_fast(qo, s_h[i], s_h[TILE+j], s_lh[i], s_lh[2*TILE+j], s_lh[TILE+i], s_lh[3*TILE+j],
                                    s_Ld[i], s_Ld[TILE+j], d2);
                if (q > P.qmax || q > PNL_MAXQ) { overflow++; q = 0; }
            }
        }
        const int nq = q ? s_ttn[q] : 0;
        const int cls = !q ? 0 : (q == qA0 ? 1 : (q == qB0 ? 2 : (nq > 0 ? 3 : 4)));
        cnt234[0] += __popcll(__ballot(q == 2)); cnt234[1] += __popcll(__ballot(q == 3)); cnt234[2] += __popcll(__ballot(q == 4));
        p2_wave_bucket_add(s_cnt, q > 4 ? q : 0);
        const unsigned short ent = (unsigned short)(p | ((q-2) << 12));
        const unsigned long long mA = __ballot(cls == 1), mB = __ballot(cls == 2), mC = __ballot(cls == 3), mF = __ballot(cls == 4);
        if (mC) {
            int base = 0;
            const int leader = __ffsll((long long)mC)-1;
            if (lane == leader) base = atomicAdd(&s_misc[3], __popcll(mC));
            base = __builtin_amdgcn_readlane(base, leader);
            if (cls == 3) s_list[PAIRS-1-(base+__popcll(mC & lt))] = ent;
        }
        if (mA) {
            int base = 0;
            const int leader = __ffsll((long long)mA)-1;
            if (lane == leader) base = atomicAdd(&s_misc[0], __popcll(mA));
            base = __builtin_amdgcn_readlane(base, leader);
            if (cls == 1) s_list[base+__popcll(mA & lt)] = ent;
        }
        if (mB) {
            int base = 0;
            const int leader = __ffsll((long long)mB)-1;
            if (lane == leader) base = atomicAdd(&s_misc[1], __popcll(mB));
            base = __builtin_amdgcn_readlane(base, leader);
            if (cls == 2) s_l32[base+__popcll(mB & lt)] = ent;
        }
        if (mF) {
            int base = 0;
            const int leader = __ffsll((long long)mF)-1;
            if (lane == leader) base = atomicAdd(&s_misc[2], __popcll(mF));
            base = __builtin_amdgcn_readlane(base, leader);
            if (cls == 4) s_l32[PAIRS-1-(base+__popcll(mF & lt))] = p | (q << 12);
        }
    }
    if (overflow) atomicAdd(&P.counters[5], (unsigned long long)overflow);
    if (lane == 0) {
#pragma unroll
        for (int k = 0; k < 3; k++) if (cnt234[k]) atomicAdd(&s_cnt[2+k], cnt234[k]);
    }
    lds_barrier();
    {
        // far pairs (orders without a packed rule): one reservation in the class's region of the global work list per tile
        const int nF = s_misc[2];
        if (nF) {
            if (tid == 0) s_misc[7] = (int)atomicAdd(wl_count+wl_region, (unsigned)nF);
            lds_barrier();
            const unsigned base = (unsigned)s_misc[7];
            int4 *__restrict__ wl = worklist+(size_t)wl_region*wl_cap;
            for (int t = tid; t < nF; t += NT) {
                const int ent = s_l32[PAIRS-1-t];
                const int p = ent & 4095, q = ent >> 12;
                const int j = p%TILE, i = (p/TILE+21*j)%TILE;
                const int off = s_off[q];
                if (base+t < wl_cap) wl[base+t] = make_int4(ta*TILE+i, tb*TILE+j, off, (s_off[q+1]-off) | (q << 16));
            }
        }
    }
    // statistics: one thread per order
    {
        const int q = 2+tid;
        if (q <= P.qmax && q <= PNL_MAXQ) {
            const int cq = s_cnt[q];
            if (cq) {
                const int ne = s_ttn[q];
                const unsigned long long n = (unsigned long long)(ne ? ne : s_off[q+1]-s_off[q]);
                st_cnt += (unsigned long long)cq;
                st_ev += n*n*cq;
            }
        }
    }
    // ---- list C: counting sort by order, 64 pairs of one order per chunk ----
    const int nC = s_misc[3];
    if (nC) {
        lds_barrier();
        if (tid == 0) {
            int run = 0, nch = 0;
            for (int q = 17; q >= 2; q--) {                    // heavy orders first
                if (q > P.qmax) continue;
                const int nq = s_ttn[q], c = s_cnt[q];
                if (!c || nq == 0 || q == qA0 || q == qB0) continue;
                s_cur[q] = run;
                for (int st = 0; st < c && nch < P2_MAXCHUNKS; st += 64) s_chunk[nch++] = (q << 20) | ((run+st) << 7) | (min(64, c-st)-1);
                run += c;
            }
            s_misc[4] = nch;
        }
        lds_barrier();
        for (int t = tid; t < nC; t += NT) {
            const int ent = s_list[PAIRS-1-t];
            const int q = (ent >> 12)+2;
            s_csort[atomicAdd(&s_cur[q], 1)] = (unsigned short)(ent & 4095);
        }
    }
    lds_barrier();

    // ---- evaluation: waves fetch chunks of 64 pairs of one order: list C (7-16 points), list B (6), list A (3) ----
    {
        const int nchC = __builtin_amdgcn_readfirstlane(s_misc[4]);
        const int totB = __builtin_amdgcn_readfirstlane(s_misc[1]), totA = __builtin_amdgcn_readfirstlane(s_misc[0]);
        const int nchB = (totB+63) >> 6, nchA = (totA+63) >> 6, nch = nchC+nchB+nchA;
        const double scale2 = 2.*kern_scale<KT>(kk);
        const double *__restrict__ ptab = (KT == 0 && kk.ptab) ? s_pow : nullptr;
        kern_dispatch<KT>(kk, ptab, [&](auto ktag) {
        constexpr int KTE = decltype(ktag)::value;          // KT, or 3: the branch-free general power (pnl_common.h)
#pragma unroll 1
        while (true) {
            int ch = 0;
            if (lane == 0) ch = atomicAdd(&s_misc[5], 1);
            ch = __builtin_amdgcn_readfirstlane(ch);
            if (ch >= nch) break;
            int p, q;
            bool act;
            if (ch < nchC) {
                const int desc = __builtin_amdgcn_readfirstlane(s_chunk[ch]);
                q = desc >> 20;
                const int start = (desc >> 7) & 8191, cnt = (desc & 127)+1;
                act = lane < cnt;
                p = s_csort[start+(act ? lane : 0)];
            } else if (ch < nchC+nchB) {
                const int idx = (ch-nchC)*64+lane;
                act = idx < totB;
                p = act ? (s_l32[idx] & 4095) : 0;
                q = qB0;
            } else {
                const int idx = (ch-nchC-nchB)*64+lane;
                act = idx < totA;
                p = act ? ((int)s_list[idx] & 4095) : 0;
                q = qA0;
            }
            const int j = p%TILE, i = (p/TILE+21*j)%TILE;
            const int nq = __builtin_amdgcn_readfirstlane(s_ttn[q]), to = __builtin_amdgcn_readfirstlane(s_tto[q]);
            const double *tab = s_tt+to*ST;
            double av[NC], bv[NC];
#pragma unroll
            for (int k = 0; k < NC; k++) { av[k] = s_v[(0*NC+k)*TILE+i]; bv[k] = s_v[(1*NC+k)*TILE+j]; }
            // NA:1405-1410: symmetric cell pairs count twice
            const double vv = act ? scale2*s_vol[i]*s_vol[TILE+j] : 0.;
            double G[6][6];
#pragma unroll
            for (int a = 0; a < 6; a++)
#pragma unroll
                for (int b = 0; b < 6; b++) G[a][b] = 0.;
            if (q == qB0 || q == qB1) {
                const int ro = q == qB0 ? 3 : 9;
                double c[NB];
                p2_eval_fixed<KTE, NB>(kk, tab, P.tt_wphif+to*7, av, bv, vv, act, s_R+(0*TILE+i)*NR+ro, G, c, ptab);
                if (act)
#pragma unroll
                    for (int jp = 0; jp < NB; jp++) lds_add_f64(&s_R[(1*TILE+j)*NR+ro+jp], vv*c[jp]);
            } else if (q == qA0) {
                double c[NA];
                p2_eval_fixed<KTE, NA>(kk, tab, P.tt_wphif+to*7, av, bv, vv, act, s_R+(0*TILE+i)*NR, G, c, ptab);
                if (act)
#pragma unroll
                    for (int jp = 0; jp < NA; jp++) lds_add_f64(&s_R[(1*TILE+j)*NR+jp], vv*c[jp]);
            } else {
                // list C (7 .. 16 points), two sweeps: the second evaluates the kernel values once more for the column sums.  (One
                // sweep with the blocked evaluator of pnl_common.h -- column sums in registers -- was measured in round 3: G, S1 and
                // 16 column sums do not fit the 256 VGPRs beside the classification state, 54 .. 170 spilled registers,
                // general tiles of C5 18.5 -> 18.7 ms.)
                double Sd[21];
#pragma unroll
                for (int e = 0; e < 21; e++) Sd[e] = 0.;
                p2_eval_lds_sweep1<KTE>(kk, tab, nq, av, bv, G, Sd, ptab);
#pragma unroll
                for (int e = 0; e < 21; e++) { if (act) lds_add_f64(&s_D[(0*TILE+i)*ND+e], vv*Sd[e]); Sd[e] = 0.; }
                p2_eval_lds_sweep2<KTE>(kk, tab, nq, av, bv, Sd, ptab);
                if (act)
#pragma unroll
                    for (int e = 0; e < 21; e++) lds_add_f64(&s_D[(1*TILE+j)*ND+e], vv*Sd[e]);
            }
            // cross block -> LDS sub-block of A'
            if (!act) continue;
            int sb[6];
#pragma unroll
            for (int b = 0; b < 6; b++) sb[b] = s_slot[(1*DPE+b)*TILE+j];
#pragma unroll
            for (int a = 0; a < 6; a++) {
                const int sa = (int)s_slot[(0*DPE+a)*TILE+i]*acc_stride;
#pragma unroll
                for (int b = 0; b < 6; b++) lds_add_f64(&s_acc[sa+sb[b]], -vv*G[a][b]);
            }
        }
        });
    }
    lds_barrier();
    } while (cmask);   // classes of a multi-class tile

    // ---- the next tile's cell data, then the flush of this one: one wave per row of the sub-block, lanes along the row of A;
    // diagonal blocks from the row / column sums ----
    const bool more = n_nxt < ntiles;
    if (more) stage(n_nxt, buf^1);
    if (tid == 0) {
        s_misc[6] = (int)(2*gridDim.x+pending);
        pending = atomicAdd(tile_ctr, 1u);
    }
    for (int t = tid; t < 2*(PNL_MAXQ+2)+6; t += NT) s_cnt[t] = 0;           // s_cnt, s_cur and s_misc[0..5] are adjacent
    const int *__restrict__ dA = s_dof+(buf*2+0)*nUe, *__restrict__ dB = s_dof+(buf*2+1)*nUe;
    if (SO.A2) {
        // block-slot storage: plain 16-byte stores of the whole sub-block; a tile that is visited once per order class
        // (bit 30 of its class word) adds instead, its sub-block has been zeroed
        const int ca = SO.colbase[ta], W = SO.S-ca;
        double *__restrict__ base = SO.A2+SO.rowoff[ta]+(SO.colbase[tb]-ca);
        const bool multi = tile_cls && (tile_cls[tile_idx] & (1 << 30));
        const int nBw = nB;
#pragma unroll 1
        for (int r = wv; r < nA; r += NWAVES) {
            double *__restrict__ row = base+(long long)r*W;
            for (int cc = 2*lane; cc < nBw; cc += 128) {
                double2 v = make_double2(0., 0.);
                if (cc < nB) { v.x = s_acc[r*acc_stride+cc]; s_acc[r*acc_stride+cc] = 0.; }
                if (cc+1 < nB) { v.y = s_acc[r*acc_stride+cc+1]; s_acc[r*acc_stride+cc+1] = 0.; }
                if (!multi) slot_store2(row+cc, v.x, v.y);
                else {
                    if (v.x != 0.) atomic_add_f64(row+cc, v.x);
                    if (v.y != 0.) atomic_add_f64(row+cc+1, v.y);
                }
            }
        }
    } else
#pragma unroll 1
    for (int r = wv; r < nA; r += NWAVES) {
        double *__restrict__ row = A+(long long)__builtin_amdgcn_readfirstlane(dA[r])*ldA;
        for (int cc = lane; cc < nB; cc += 64) {
            const double v = s_acc[r*acc_stride+cc];
            if (v != 0.) {
                if (!symflush) s_acc[r*acc_stride+cc] = 0.;
                atomic_add_f64(&row[dB[cc]], v);
            }
        }
    }
    if (symflush) lds_barrier();                              // the transposed sweep zeroes what the first one reads
    if (symflush)
#pragma unroll 1
        for (int cc = wv; cc < nB; cc += NWAVES) {
            double *__restrict__ row = A+(long long)__builtin_amdgcn_readfirstlane(dB[cc])*ldA;
            for (int r = lane; r < nA; r += 64) {
                const double v = s_acc[r*acc_stride+cc];
                if (v != 0.) {
                    s_acc[r*acc_stride+cc] = 0.;
                    atomic_add_f64(&row[dA[r]], v);
                }
            }
        }
    for (int t = tid; t < 2*TILE*ND; t += NT) {
        const int sc = t/ND, e = t-sc*ND;                    // sc = side*TILE + cell
        double v = s_D[t];
#pragma unroll
        for (int k = 0; k < NR; k++) v = __builtin_fma(s_PP[e*NR+k], s_R[sc*NR+k], v);
        if (v != 0.) {
            const int side = sc/TILE, c = (side ? tb : ta)*TILE+(sc-side*TILE);
            atomic_add_f64(&Dglob[(size_t)c*ND+e], v);
        }
    }
    // the buffers the tile before this one accumulated into (their flush is complete) are zeroed for the next tile
    for (int t = tid; t < 2*TILE*ND; t += NT) s_Dall[(buf^1)*2*TILE*ND+t] = 0.;
    for (int t = tid; t < 2*TILE*NR; t += NT) s_Rall[(buf^1)*2*TILE*NR+t] = 0.;
    if (!more) break;
    lds_barrier();
    n_cur = n_nxt; n_nxt = s_misc[6]; buf ^= 1;
    }   // tile loop
    {
        const int q = 2+tid;
        if (st_cnt) {
            atomicAdd(&P.counters[8+q], st_cnt);
            atomicAdd(&P.counters[1], st_cnt);
            atomicAdd(&P.counters[2], st_ev);
        }
    }
}
