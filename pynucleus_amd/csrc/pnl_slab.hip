// Row-slab storage of a rank's part of the dense operator and its matvec (gfx950 only).
//
// Reference: the distributed operators of clusterMethodCy.pyx (DistributedH2Matrix_globalData.matvec :3127-3154: local
// product, Allreduce of the N-vector) on top of a row partition (tree_node.partition :1854-1896).  A rank owns a
// contiguous range of cell blocks [a0, a1) and assembles the tiles (a, b), a in its range, b >= a, one-sided:
//   slab[row(I)][col(J)] = A'[I][J],  rows = the DoFs of its cells (+ the DoFs of the cells touching them: touching pairs
//   write their symmetric local matrix once, at (min, max)),  columns = the DoFs of its cells and of all later cells,
// plus the per-cell diagonal blocks D_c it accumulated for ALL cells (its partial sums).  The operator is
//   A = sum over ranks of  A'_r + A'_r^T - diag(A'_r) + sum_c scatter(D_c),
// applied as local products and ONE all-reduce of the N-vector.  Per-rank memory: rows_r x cols_r ~ N^2 / (2 P) doubles.
#include "pnl_context.h"
#include "pnl_common.h"

namespace {

// y[rowdof[r]] += sum_j slab[r][j] x[col0 + j]  (wave per row; every row belongs to one DoF: plain accumulate through atomics
// because the transposed sweep and the diagonal blocks add into the same vector)
__global__ void __launch_bounds__(PNL_NTHREADS)
k_slab_gemv(const double *__restrict__ S, long long ld, int nrows, int ncols, const int *__restrict__ rowdof, const int *__restrict__ coldof,
            const double *__restrict__ x, double *__restrict__ y) {
    const int lane = threadIdx.x & 63;
    const int row = (blockIdx.x*PNL_NTHREADS+threadIdx.x) >> 6;
    if (row >= nrows) return;
    const double *__restrict__ a = S+(long long)row*ld;
    double s = 0.;
    for (int j = lane; j < ncols; j += 64) s = __builtin_fma(a[j], x[coldof[j]], s);
    s = wave_sum(s);
    if (lane == 0 && s != 0.) atomic_add_f64(&y[rowdof[row]], s);
}

// y[col0 + j] += sum_r slab[r][j] x[rowdof[r]] without the diagonal entries (they are counted by k_slab_gemv)
__global__ void __launch_bounds__(PNL_NTHREADS)
k_slab_gemv_t(const double *__restrict__ S, long long ld, int nrows, int ncols, const int *__restrict__ rowdof, const int *__restrict__ coldof,
              const double *__restrict__ x, double *__restrict__ y, int rows_per_block) {
    const int r0 = blockIdx.y*rows_per_block, r1 = min(nrows, r0+rows_per_block);
    const int j = blockIdx.x*PNL_NTHREADS+threadIdx.x;
    if (j >= ncols) return;
    const int J = coldof[j];
    double s = 0.;
    for (int r = r0; r < r1; r++) {
        const int I = rowdof[r];
        if (I != J) s = __builtin_fma(S[(long long)r*ld+j], x[I], s);
    }
    if (s != 0.) atomic_add_f64(&y[J], s);
}

// y[dofs(c)] += D_c x[dofs(c)] for the symmetric per-cell blocks (upper triangle, row-major)
__global__ void __launch_bounds__(PNL_NTHREADS)
k_dblocks_matvec(const int *__restrict__ cdof, int ncp, int nc, int dpe, const double *__restrict__ D, const double *__restrict__ x,
                 double *__restrict__ y) {
    const int nd = dpe*(dpe+1)/2;
    const int t = blockIdx.x*PNL_NTHREADS+threadIdx.x;
    const int c = t/dpe, a = t-c*dpe;
    if (c >= nc) return;
    const int I = cdof[(size_t)a*ncp+c];
    if (I < 0) return;
    double s = 0.;
    for (int b = 0; b < dpe; b++) {
        const int J = cdof[(size_t)b*ncp+c];
        if (J < 0) continue;
        const int lo = min(a, b), hi = max(a, b);
        s = __builtin_fma(D[(size_t)c*nd+dpe*lo-(lo*(lo+1) >> 1)+hi], x[J], s);
    }
    if (s != 0.) atomic_add_f64(&y[I], s);
}

// diag[rowdof[r]] += slab[r][rowdof[r] - col0];  diagonal of the per-cell blocks
__global__ void __launch_bounds__(PNL_NTHREADS)
k_slab_diag(const double *__restrict__ S, long long ld, int nrows, const int *__restrict__ rowdof, const int *__restrict__ colmap, double *__restrict__ diag) {
    const int r = blockIdx.x*PNL_NTHREADS+threadIdx.x;
    if (r >= nrows) return;
    const int j = colmap[rowdof[r]];
    if (j >= 0) { const double v = S[(long long)r*ld+j]; if (v != 0.) atomic_add_f64(&diag[rowdof[r]], v); }
}
__global__ void __launch_bounds__(PNL_NTHREADS)
k_dblocks_diag(const int *__restrict__ cdof, int ncp, int nc, int dpe, const double *__restrict__ D, double *__restrict__ diag) {
    const int nd = dpe*(dpe+1)/2;
    const int t = blockIdx.x*PNL_NTHREADS+threadIdx.x;
    const int c = t/dpe, a = t-c*dpe;
    if (c >= nc) return;
    const int I = cdof[(size_t)a*ncp+c];
    if (I < 0) return;
    const double v = D[(size_t)c*nd+dpe*a-(a*(a+1) >> 1)+a];
    if (v != 0.) atomic_add_f64(&diag[I], v);
}

}  // namespace

extern "C" {

int pnl_set_row_slab(pnl_context *ctx, int nrows, const int32_t *rowdofs, int ncols, const int32_t *coldofs) {
    if (!ctx) return PNL_ERR_INVALID;
    if (!ctx->have_dofs) return fail(ctx, PNL_ERR_STATE, "upload the DoF map first");
    if (nrows == 0) {
        ctx->slab_rows = 0; ctx->slab_cols = 0; ctx->slab_rowdofs.clear(); ctx->slab_coldofs.clear();
        ctx->P.rowmap = nullptr; ctx->P.colmap = nullptr; ctx->P.onesided = 0;
        return PNL_OK;
    }
    if (nrows < 0 || !rowdofs || ncols <= 0 || !coldofs) return fail(ctx, PNL_ERR_INVALID, "bad row slab");
    std::vector<int32_t> rowmap(ctx->N, -1), colmap(ctx->N, -1);
    for (int r = 0; r < nrows; r++) {
        if (rowdofs[r] < 0 || rowdofs[r] >= ctx->N || (r && rowdofs[r] <= rowdofs[r-1]))
            return fail(ctx, PNL_ERR_INVALID, "row DoFs must be increasing and lie in [0, num_dofs)");
        rowmap[rowdofs[r]] = r;
    }
    for (int c = 0; c < ncols; c++) {
        if (coldofs[c] < 0 || coldofs[c] >= ctx->N || (c && coldofs[c] <= coldofs[c-1]))
            return fail(ctx, PNL_ERR_INVALID, "column DoFs must be increasing and lie in [0, num_dofs)");
        colmap[coldofs[c]] = c;
    }
    int rc;
    if ((rc = upload(ctx, ctx->b_rowmap, rowmap.data(), rowmap.size()))) return rc;
    if ((rc = upload(ctx, ctx->b_colmap, colmap.data(), colmap.size()))) return rc;
    if ((rc = upload(ctx, ctx->b_rowdof, rowdofs, (size_t)nrows))) return rc;
    if ((rc = upload(ctx, ctx->b_coldof, coldofs, (size_t)ncols))) return rc;
    ctx->slab_rowdofs.assign(rowdofs, rowdofs+nrows);
    ctx->slab_coldofs.assign(coldofs, coldofs+ncols);
    ctx->slab_rows = nrows; ctx->slab_cols = ncols;
    ctx->P.rowmap = (const int*)ctx->b_rowmap.p; ctx->P.colmap = (const int*)ctx->b_colmap.p; ctx->P.onesided = 1;
    return PNL_OK;
}

int pnl_diag_blocks_size(pnl_context *ctx) {
    if (!ctx || !ctx->have_dofs) return PNL_ERR_INVALID;
    const int rc = pnl_tile_cells(ctx);                    // builds the padded cell tables if they are not there yet
    if (rc < 0) return rc;
    const int nd = ctx->dpe*(ctx->dpe+1)/2;
    return 2*ctx->ncp*nd;
}

int pnl_get_diag_blocks(pnl_context *ctx, double *dst) {
    if (!ctx || !dst) return PNL_ERR_INVALID;
    if (!ctx->b_D.p) return fail(ctx, PNL_ERR_STATE, "nothing assembled yet");
    { const int rc = pnl_tile_order_ready(ctx); if (rc) return rc; }
    const size_t n = (size_t)ctx->ncp*(ctx->dpe*(ctx->dpe+1)/2);
    HIPCHK(ctx, hipMemcpyAsync(dst, ctx->b_D.p, sizeof(double)*n, hipMemcpyDeviceToDevice, ctx->stream));
    if (ctx->have_tile_order) HIPCHK(ctx, hipMemcpyAsync(dst+n, ctx->b_Dt.p, sizeof(double)*n, hipMemcpyDeviceToDevice, ctx->stream));
    else HIPCHK(ctx, hipMemsetAsync(dst+n, 0, sizeof(double)*n, ctx->stream));
    return PNL_OK;
}

int pnl_slab_matvec(pnl_context *ctx, const double *slab, int64_t ld, const double *dblocks, const double *x, double *y) {
    if (!ctx || !slab || !x || !y) return PNL_ERR_INVALID;
    if (ctx->slab_rows <= 0) return fail(ctx, PNL_ERR_STATE, "no row slab set (pnl_set_row_slab)");
    const int nrows = ctx->slab_rows, ncols = ctx->slab_cols;
    { const int rc = pnl_tile_order_ready(ctx); if (rc) return rc; }
    if (ld < ncols) return fail(ctx, PNL_ERR_INVALID, "slab leading dimension %lld < %d columns", (long long)ld, ncols);
    hipStream_t st = ctx->stream;
    HIPCHK(ctx, hipMemsetAsync(y, 0, sizeof(double)*ctx->N, st));
    const int *rowdof = (const int*)ctx->b_rowdof.p, *coldof = (const int*)ctx->b_coldof.p;
    // A' x and A'^T x in ONE sweep over the slab (pnl_gemv2.hip); k_slab_gemv / k_slab_gemv_t above are the two-sweep form
    { const int rc = pnl_launch_slab_two_sided(ctx, slab, (long long)ld, nrows, ncols, rowdof, coldof, x, y); if (rc) return rc; }
    if (dblocks) {
        const size_t n = (size_t)ctx->ncp*(ctx->dpe*(ctx->dpe+1)/2);
        const int nt = ctx->nc*ctx->dpe;
        hipLaunchKernelGGL(k_dblocks_matvec, dim3((nt+PNL_NTHREADS-1)/PNL_NTHREADS), dim3(PNL_NTHREADS), 0, st, (const int*)ctx->b_cdof.p,
                           ctx->ncp, ctx->nc, ctx->dpe, dblocks, x, y);
        if (ctx->have_tile_order)
            hipLaunchKernelGGL(k_dblocks_matvec, dim3((nt+PNL_NTHREADS-1)/PNL_NTHREADS), dim3(PNL_NTHREADS), 0, st,
                               (const int*)ctx->b_cdof_t.p, ctx->ncp, ctx->nc, ctx->dpe, dblocks+n, x, y);
    }
    HIPCHK(ctx, hipGetLastError());
    return PNL_OK;
}

int pnl_slab_diagonal(pnl_context *ctx, const double *slab, int64_t ld, const double *dblocks, double *diag) {
    if (!ctx || !slab || !diag) return PNL_ERR_INVALID;
    if (ctx->slab_rows <= 0) return fail(ctx, PNL_ERR_STATE, "no row slab set (pnl_set_row_slab)");
    const int nrows = ctx->slab_rows;
    { const int rc = pnl_tile_order_ready(ctx); if (rc) return rc; }
    hipStream_t st = ctx->stream;
    HIPCHK(ctx, hipMemsetAsync(diag, 0, sizeof(double)*ctx->N, st));
    hipLaunchKernelGGL(k_slab_diag, dim3((nrows+PNL_NTHREADS-1)/PNL_NTHREADS), dim3(PNL_NTHREADS), 0, st, slab, (long long)ld, nrows,
                       (const int*)ctx->b_rowdof.p, (const int*)ctx->b_colmap.p, diag);
    if (dblocks) {
        const size_t n = (size_t)ctx->ncp*(ctx->dpe*(ctx->dpe+1)/2);
        const int nt = ctx->nc*ctx->dpe;
        hipLaunchKernelGGL(k_dblocks_diag, dim3((nt+PNL_NTHREADS-1)/PNL_NTHREADS), dim3(PNL_NTHREADS), 0, st, (const int*)ctx->b_cdof.p, ctx->ncp,
                           ctx->nc, ctx->dpe, dblocks, diag);
        if (ctx->have_tile_order)
            hipLaunchKernelGGL(k_dblocks_diag, dim3((nt+PNL_NTHREADS-1)/PNL_NTHREADS), dim3(PNL_NTHREADS), 0, st, (const int*)ctx->b_cdof_t.p,
                               ctx->ncp, ctx->nc, ctx->dpe, dblocks+n, diag);
    }
    HIPCHK(ctx, hipGetLastError());
    return PNL_OK;
}

}  // extern "C"
