// Launchers of the second-generation tile kernels (pnl_tile2.h): uniform-order tiles (P1 orders 3 / 4, P2 orders 2-4) and the
// general P2 tile kernel.  Called from assemble_impl / launch_tiles in pnl_hip.hip.
#include "pnl_context.h"
#include "pnl_tile2.h"

namespace {

template <int DPE, int NP, int KT, bool STRUCT = false>
int launch_uniform_t(pnl_context *ctx, const DevProblem &Pt, const int2 *tiles, const int *tile_cls, int ntiles, int q, double *A,
                     int64_t ldA, double *Dglob, const SlotOut &SO) {
    constexpr int TILE = DPE == 6 ? 32 : 64;
    const int nUe = (ctx->nU+1) & ~1;                     // even: the sub-block follows the int arrays at an 8-byte boundary
    const size_t fixed = uniform_fixed_lds(DPE, NP, TILE, nUe, SO.A2 == nullptr);
    // sub-block [nUe+1][acc_stride]: rows in different LDS banks (stride = 1 mod 32 doubles) if two workgroups still share a CU,
    // else an odd stride
    int acc_stride = nUe+1;
    while (acc_stride % 32 != 1) acc_stride++;                      // (other odd residues mod 32 measured: no difference)
    if (fixed+sizeof(double)*(size_t)(nUe+1)*acc_stride > 80*1024) acc_stride = (nUe+1) | 1;
    size_t lds = fixed+sizeof(double)*(size_t)(nUe+1)*acc_stride;
    if (lds > 160*1024)
        return fail(ctx, PNL_ERR_UNSUPPORTED, "a block of %d cells touches %d DoFs: LDS sub-block of %zu bytes exceeds 160 KiB", TILE, ctx->nU, lds);
    // general exponent: the tables of pnl_pow_tab behind the sub-block, unless they cost the second workgroup per CU
    // P1: 108-168 VGPRs allow three or four waves per SIMD and as many 38-53 KB workgroups share a CU (order-3 tiles of a general
    // exponent 36.4 -> 28.8 ms at 98,304 cells, s = 1/2: 17.8 -> 17.2; order-2 tiles of s = 1/2 with four: 42.8 -> 41.3 ms);
    // P2: two workgroups of 80 KB
    const size_t cap_cu = (size_t)std::max(1, pnl_tune("PNL_UNI_PER_CU") ? atoi(pnl_tune("PNL_UNI_PER_CU")) : 4);
    auto kfun = k_tile_uniform<DPE, NP, KT, STRUCT>;
    HIPCHK(ctx, hipFuncSetAttribute((const void*)kfun, hipFuncAttributeMaxDynamicSharedMemorySize, (int)std::min<size_t>(160*1024, lds+sizeof(double)*PNL_POW_TAB_DOUBLES)));
    // workgroups that are really resident per CU (LDS and registers): the tile loop strides by the grid, a workgroup that has
    // to wait for a slot would start its share of the tiles late
    auto resident = [&](size_t bytes) -> int {
        int occ = 0;
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, (const void*)kfun, 256, bytes) != hipSuccess || occ < 1) {
            (void)hipGetLastError();
            occ = (int)std::min<size_t>(2, (160*1024)/bytes);
        }
        return std::max(1, std::min((int)cap_cu, occ));
    };
    int pow_flag = 0;
    if (KT == 0) {
        // general exponent: the tables of pnl_pow_tab behind the sub-block, unless they cost a resident workgroup
        const size_t tab = sizeof(double)*PNL_POW_TAB_DOUBLES;
        const int per_cu0 = resident(lds);
        if (lds+tab <= 160*1024 && resident(lds+tab) == per_cu0) { lds += tab; pow_flag = 8; }
        else if (!pnl_tune("PNL_UNI_KEEP_STRIDE")) {
            // the padded row stride of the sub-block (fewer LDS bank conflicts) or the tables: the tables win (measured)
            const int odd = (nUe+1) | 1;
            const size_t alt = fixed+sizeof(double)*(size_t)(nUe+1)*odd+tab;
            if (alt <= 160*1024 && resident(alt) == per_cu0) { acc_stride = odd; lds = alt; pow_flag = 8; }
        }
    }
    HIPCHK(ctx, hipFuncSetAttribute((const void*)kfun, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    const int per_cu = resident(lds);
    const int grid = pnl_grid_cap(std::min(ntiles, 256*per_cu));
    if (pnl_tune("PNL_VERBOSE"))
        fprintf(stderr, "[pnl] uniform tiles of order %d: %d, dpe=%d np=%d kt=%d struct=%d lds=%zu bytes (%d per CU), acc_stride=%d\n", q, ntiles, DPE,
                NP, KT, (int)STRUCT, lds, per_cu, acc_stride);
    int uni_abl = 0;
#ifdef PNL_DEBUG_ABLATE
    uni_abl = pnl_tune("PNL_UNI_ABL") ? atoi(pnl_tune("PNL_UNI_ABL")) : 0;
#endif
    kt_begin(ctx, PNL_K_TILE_UNIFORM2+(q-2));
    hipLaunchKernelGGL(kfun, dim3(grid), dim3(256), lds, ctx->stream, Pt, tiles, tile_cls, (const DevKernel*)ctx->b_kcls.p, ntiles, A,
                       (long long)ldA, Dglob, acc_stride, q, (ctx->symflush ? 1 : 0) | uni_abl | pow_flag, (const double*)ctx->b_uni.p+ctx->uni_off[q],
                       nUe, SO);
    kt_end(ctx, PNL_K_TILE_UNIFORM2+(q-2));
    HIPCHK(ctx, hipGetLastError());
    return PNL_OK;
}

template <int DPE, int NP>
int launch_uniform_kt(pnl_context *ctx, int kt, const DevProblem &Pt, const int2 *tiles, const int *tile_cls, int ntiles, int q,
                      double *A, int64_t ldA, double *Dglob, const SlotOut &SO) {
    if constexpr (DPE == 3) {
        if (ctx->uni_struct[q] && !pnl_tune("PNL_UNI_GENERIC")) {
            if (kt == 2) return launch_uniform_t<DPE, NP, 2, true>(ctx, Pt, tiles, tile_cls, ntiles, q, A, ldA, Dglob, SO);
            if (kt == 1) return launch_uniform_t<DPE, NP, 1, true>(ctx, Pt, tiles, tile_cls, ntiles, q, A, ldA, Dglob, SO);
            return launch_uniform_t<DPE, NP, 0, true>(ctx, Pt, tiles, tile_cls, ntiles, q, A, ldA, Dglob, SO);
        }
    }
    if (kt == 2) return launch_uniform_t<DPE, NP, 2>(ctx, Pt, tiles, tile_cls, ntiles, q, A, ldA, Dglob, SO);
    if (kt == 1) return launch_uniform_t<DPE, NP, 1>(ctx, Pt, tiles, tile_cls, ntiles, q, A, ldA, Dglob, SO);
    return launch_uniform_t<DPE, NP, 0>(ctx, Pt, tiles, tile_cls, ntiles, q, A, ldA, Dglob, SO);
}

template <int KT>
int launch_p2_t(pnl_context *ctx, const int2 *tiles, const int *tile_cls, int ntiles, double *A, int64_t ldA, int cell_begin,
                int cell_end, unsigned wl_cap_each, const SlotOut &SO) {
    using S = P2Smem;
    const int nUe = (ctx->nU+1) & ~1;
    // one workgroup per CU: rows of the sub-block start in different LDS banks (stride = 1 mod 32 doubles) if that fits
    int stride = nUe+1;
    while (stride % 32 != 1) stride++;
    size_t lds = S::fixed_bytes(nUe)+sizeof(double)*(size_t)(nUe+1)*stride;
    if (lds > 160*1024) { stride = (nUe+1) | 1; lds = S::fixed_bytes(nUe)+sizeof(double)*(size_t)(nUe+1)*stride; }
    if (lds > 160*1024)
        return fail(ctx, PNL_ERR_UNSUPPORTED, "a block of %d cells touches %d DoFs: LDS sub-block of %zu bytes exceeds 160 KiB "
                    "(cells must be numbered with spatial locality)", P2_TILE, ctx->nU, lds);
    auto kfun = k_tile_p2<KT>;
    HIPCHK(ctx, hipFuncSetAttribute((const void*)kfun, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    // two tiles per workgroup are taken by block index, the rest through tickets: a grid of at most ntiles / 2 workgroups
    const int grid = pnl_grid_cap(std::max(1, std::min((ntiles+1)/2, 256)));
    if (pnl_tune("PNL_VERBOSE")) fprintf(stderr, "[pnl] P2 general tiles=%d nU=%d kt=%d lds=%zu bytes acc_stride=%d\n", ntiles, ctx->nU, KT, lds, stride);
    kt_begin(ctx, PNL_K_TILE_GENERAL);
    hipLaunchKernelGGL(kfun, dim3(grid), dim3(P2_NT), lds, ctx->stream, ctx->P, tiles, tile_cls, (const DevKernel*)ctx->b_kcls.p,
                       (const DevFormula*)ctx->b_fcls.p, A, (long long)ldA, (double*)ctx->b_D.p, cell_begin, cell_end, stride,
                       (int4*)ctx->b_wl.p, (unsigned*)ctx->b_wlcount.p, wl_cap_each, ctx->symflush ? 256 : 0, ntiles,
                       (unsigned*)ctx->b_tilectr.p, nUe, SO);
    kt_end(ctx, PNL_K_TILE_GENERAL);
    HIPCHK(ctx, hipGetLastError());
    return PNL_OK;
}

}  // namespace

// uniform tiles of one order; Pt / Dglob: the cell tables (and diagonal-block buffer) the tile kernels use
int pnl2_launch_uniform(pnl_context *ctx, int kt, const DevProblem &Pt, const int2 *tiles, const int *tile_cls, int ntiles, int q,
                        double *A, int64_t ldA, double *Dglob, const SlotOut &SO) {
    if (ntiles <= 0) return PNL_OK;
    if (q < 2 || q > 4 || ctx->uni_off[q] < 0) return fail(ctx, PNL_ERR_STATE, "no uniform-tile rule for order %d", q);
    const int np = ctx->uni_np[q];
    if (ctx->dpe == 6 && np == 3) return launch_uniform_kt<6, 3>(ctx, kt, Pt, tiles, tile_cls, ntiles, q, A, ldA, Dglob, SO);
    if (ctx->dpe == 6 && np == 6) return launch_uniform_kt<6, 6>(ctx, kt, Pt, tiles, tile_cls, ntiles, q, A, ldA, Dglob, SO);
    if (ctx->dpe == 3 && np == 6) return launch_uniform_kt<3, 6>(ctx, kt, Pt, tiles, tile_cls, ntiles, q, A, ldA, Dglob, SO);
    if (ctx->dpe == 3 && np == 3) return launch_uniform_kt<3, 3>(ctx, kt, Pt, tiles, tile_cls, ntiles, q, A, ldA, Dglob, SO);
    return fail(ctx, PNL_ERR_UNSUPPORTED, "uniform tiles: dpe=%d with %d points", ctx->dpe, np);
}

int pnl2_launch_p2(pnl_context *ctx, int kt, const int2 *tiles, const int *tile_cls, int ntiles, double *A, int64_t ldA,
                   int cell_begin, int cell_end, unsigned wl_cap_each, const SlotOut &SO) {
    if (ntiles <= 0) return PNL_OK;
    if (kt == 2) return launch_p2_t<2>(ctx, tiles, tile_cls, ntiles, A, ldA, cell_begin, cell_end, wl_cap_each, SO);
    if (kt == 1) return launch_p2_t<1>(ctx, tiles, tile_cls, ntiles, A, ldA, cell_begin, cell_end, wl_cap_each, SO);
    return launch_p2_t<0>(ctx, tiles, tile_cls, ntiles, A, ldA, cell_begin, cell_end, wl_cap_each, SO);
}

int pnl2_zero_slot_tiles(pnl_context *ctx, const SlotOut &SO) {
    if (ctx->n_multitiles <= 0) return PNL_OK;
    hipLaunchKernelGGL(k_zero_slot_tiles, dim3(ctx->n_multitiles), dim3(256), 0, ctx->stream, SO, (const int2*)ctx->b_multitiles.p,
                       (const int*)ctx->b_blk_ndof.p);
    HIPCHK(ctx, hipGetLastError());
    return PNL_OK;
}

int pnl2_fold_mirror(pnl_context *ctx, const SlotOut &SO, double *A, int64_t ldA) {
    const long long nb = (ctx->N+31)/32, nbs = (nb+7)/8;
    if (nbs*(nbs+1)/2*64 >= (1ll << 31)) return fail(ctx, PNL_ERR_UNSUPPORTED, "fold pass: %d DoFs exceed the grid size", ctx->N);
    kt_begin(ctx, PNL_K_FOLD_MIRROR);
    // the matrix is written once and not read again by this pass: non-temporal stores (10.9 -> 10.4 ms at 48,769 DoFs; non-temporal
    // LOADS of the storage cost 50 %: the 8-byte gathers of neighbouring threads share lines)
    hipLaunchKernelGGL(k_fold_mirror<true>, dim3((unsigned)(nbs*(nbs+1)/2*64)), dim3(256), 0, ctx->stream, (const double*)SO.A2, (const FoldEntry*)ctx->b_foldtab.p, (const int*)ctx->b_cpoff.p,
                       (const int2*)ctx->b_cpslot.p, (const long long*)ctx->b_cprow.p, A, (long long)ldA, ctx->N);
    kt_end(ctx, PNL_K_FOLD_MIRROR);
    HIPCHK(ctx, hipGetLastError());
    return PNL_OK;
}
