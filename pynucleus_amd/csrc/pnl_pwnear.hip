// Near field and far field of the non-symmetric kernels with an order per quadrature point (gfx950 only): the part of
// assembleClusters / getH2 (nonlocalAssembly_{SCALAR}.pxi:1663-2156, 3094-3219) these kernels take -- ordered element pairs with
// masks over the (2 dpe)^2 local entries, the cluster exterior with the pointwise boundary kernel, the kernel interpolants of the
// admissible pairs with the order at the nodes of the row cluster.  Kernels: pnl_pointwise.h.
#include "pnl_context.h"
// pnl_kernels.h defines its non-template kernels without `inline`: this second translation unit keeps its copies (and the templates
// it instantiates) in an unnamed namespace, so nothing collides with pnl_hip.o at link time
namespace {
#include "pnl_pointwise.h"
}

namespace {

int near_pattern(pnl_context *ctx, double *data, PwNear &NR) {
    if (ctx->sp_nnz < 0) return fail(ctx, PNL_ERR_STATE, "upload the sparsity pattern first");
    if (!data && ctx->sp_nnz > 0) return fail(ctx, PNL_ERR_INVALID, "null output");
    NR.indptr = (const int*)ctx->b_sp_indptr.p; NR.indices = (const int*)ctx->b_sp_indices.p;
    NR.data = data; NR.masks = (const unsigned long long*)ctx->b_mp_masks.p;
    return PNL_OK;
}

// items: touching[nt] / distant[nd] index the pair list; the touching ones come as (c1, c2, common, key) in b_pw_pairs
template <int DIM, int DPE>
int pairs_near_impl(pnl_context *ctx, int np, int nt, int nd, const PwNear &NR) {
    constexpr int ST = 4+DPE;
    int rc;
    const DevProblem &P = ctx->P;
    const PwDev &W = ctx->pw;
    HIPCHK(ctx, hipMemsetAsync(ctx->b_counters.p, 0, sizeof(unsigned long long)*PNL_NCOUNTERS, ctx->stream));
    HIPCHK(ctx, hipEventRecord(ctx->ev[0], ctx->stream));
    ctx->tiles_launched = false; ctx->pure_launched = false;
    for (int e : {7, 6}) HIPCHK(ctx, hipEventRecord(ctx->ev[e], ctx->stream));
    if ((rc = ensure(ctx, ctx->b_wlcount, sizeof(unsigned)*PNL_WL_SLOTS))) return rc;
    HIPCHK(ctx, hipMemsetAsync(ctx->b_wlcount.p, 0, sizeof(unsigned), ctx->stream));
    ctx->wl_slots = 1;
    if (nd > 0) {
        // classification -> work list sorted by order -> 16 lanes per pair (k_pw_distant, this orientation alone)
        const size_t want = std::max<size_t>((size_t)nd, 1024);
        if (ctx->wl_cap < want) {
            if ((rc = ensure(ctx, ctx->b_wl, want*sizeof(int4)))) return rc;
            ctx->wl_cap = (unsigned)want;
        }
        if ((rc = ensure(ctx, ctx->b_wlsorted, (size_t)ctx->wl_cap*sizeof(int4)))) return rc;
        if ((rc = ensure(ctx, ctx->b_wlcount, sizeof(unsigned)*PNL_WL_SLOTS))) return rc;
        if ((rc = ensure(ctx, ctx->b_wlaux, sizeof(unsigned)*(4*(PNL_WL_BINS+1))))) return rc;
        HIPCHK(ctx, hipMemsetAsync(ctx->b_wlcount.p, 0, sizeof(unsigned), ctx->stream));
        ctx->wl_slots = 1; ctx->wl_cap_each = ctx->wl_cap;
        hipLaunchKernelGGL((k_pw_classify_near<DIM, DPE>), dim3((nd+PNL_NTHREADS-1)/PNL_NTHREADS), dim3(PNL_NTHREADS), 0, ctx->stream, P, W,
                           (const int*)ctx->b_mp_pairs.p, (const int*)ctx->b_mp_sorted.p, nd, (int4*)ctx->b_wl.p,
                           (unsigned*)ctx->b_wlcount.p, ctx->wl_cap);
        unsigned *hist = (unsigned*)ctx->b_wlaux.p, *offs = hist+(PNL_WL_BINS+1), *coff = offs+(PNL_WL_BINS+1), *cursor = coff+(PNL_WL_BINS+1);
        HIPCHK(ctx, hipMemsetAsync(hist, 0, sizeof(unsigned)*(PNL_WL_BINS+1), ctx->stream));
        const int4 *wl = (const int4*)ctx->b_wl.p;
        const unsigned *wlc = (const unsigned*)ctx->b_wlcount.p;
        hipLaunchKernelGGL(k_wl_hist, dim3(512), dim3(PNL_NTHREADS), 0, ctx->stream, wl, wlc, ctx->wl_cap, hist);
        hipLaunchKernelGGL(k_wl_scan, dim3(1), dim3(64), 0, ctx->stream, (const unsigned*)hist, offs, coff, cursor);
        hipLaunchKernelGGL(k_wl_scatter, dim3(512), dim3(PNL_NTHREADS), 0, ctx->stream, wl, wlc, ctx->wl_cap, (const unsigned*)offs, cursor,
                           (int4*)ctx->b_wlsorted.p);
        hipLaunchKernelGGL(k_pw_stats_near, dim3(1), dim3(PNL_WL_BINS), 0, ctx->stream, P, (const unsigned*)hist);
        const int tab_max = 256;
        const size_t lds = sizeof(double)*((size_t)tab_max*ST+(size_t)(PNL_NTHREADS/16)*tab_max*2);
        auto kfun = k_pw_distant<DIM, DPE, true>;
        HIPCHK(ctx, hipFuncSetAttribute((const void*)kfun, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        hipLaunchKernelGGL(kfun, dim3(256*4), dim3(PNL_NTHREADS), lds, ctx->stream, P, W, (const int4*)ctx->b_wlsorted.p, (const unsigned*)offs,
                           (double*)nullptr, 0ll, (double*)nullptr, tab_max, 0, NR);
        HIPCHK(ctx, hipGetLastError());
    }
    HIPCHK(ctx, hipEventRecord(ctx->ev[1], ctx->stream));
    HIPCHK(ctx, hipEventRecord(ctx->ev[2], ctx->stream));
    if (nt > 0) {
        const unsigned grid = (unsigned)(((long long)nt*64+PNL_NTHREADS-1)/PNL_NTHREADS);
        const int4 *pp = (const int4*)ctx->b_pw_pairs.p;
        const int *item = (const int*)ctx->b_mp_wl.p;
        hipLaunchKernelGGL((k_pw_singular<DIM, DPE, 0, true>), dim3(grid), dim3(PNL_NTHREADS), 0, ctx->stream, P, W, pp, nt, (double*)nullptr, 0ll, 0, 0, NR, item);
        hipLaunchKernelGGL((k_pw_singular<DIM, DPE, 1, true>), dim3(grid), dim3(PNL_NTHREADS), 0, ctx->stream, P, W, pp, nt, (double*)nullptr, 0ll, 0, 0, NR, item);
        if (DIM == 2)
            hipLaunchKernelGGL((k_pw_singular<DIM, DPE, (DIM == 2 ? 2 : 1), true>), dim3(grid), dim3(PNL_NTHREADS), 0, ctx->stream, P, W, pp, nt, (double*)nullptr, 0ll, 0, 0, NR, item);
        HIPCHK(ctx, hipGetLastError());
    }
    for (int e : {3, 4, 5}) HIPCHK(ctx, hipEventRecord(ctx->ev[e], ctx->stream));
    ctx->ev_valid = true;
    ctx->visited_pairs = (unsigned long long)np; ctx->visited_is_assembled = false;
    return PNL_OK;
}

template <int DIM, int DPE>
int boundary_near_impl(pnl_context *ctx, int ni, double fac, const SparseOut &S) {
    const int grid = std::min((ni+3)/4, 256*8);
    hipLaunchKernelGGL((k_pw_boundary_items<DIM, DPE>), dim3(grid), dim3(PNL_NTHREADS), 0, ctx->stream, ctx->P, ctx->pw,
                       (const double*)ctx->b_vertices.p, (const int*)ctx->b_bi_cells.p, (const int*)ctx->b_bi_facets.p,
                       (const unsigned*)ctx->b_bi_masks.p, (const int*)ctx->b_mp_aux.p, (const double*)ctx->b_mp_sorted.p, ni, fac, S);
    HIPCHK(ctx, hipGetLastError());
    return PNL_OK;
}

}  // namespace

int pnl_pw_h2_interp(pnl_context *ctx) {
    const H2Dev &H = ctx->h2;
    if (H.nfar <= 0) return PNL_OK;
    if (ctx->dim == 2) hipLaunchKernelGGL((k_h2_kernel_interp_pw<2>), dim3(H.nfar), dim3(PNL_NTHREADS), 0, ctx->stream, H, ctx->pw);
    else hipLaunchKernelGGL((k_h2_kernel_interp_pw<1>), dim3(H.nfar), dim3(PNL_NTHREADS), 0, ctx->stream, H, ctx->pw);
    HIPCHK(ctx, hipGetLastError());
    return PNL_OK;
}

extern "C" {

int pnl_assemble_pairs_masked_pointwise(pnl_context *ctx, int np, const int32_t *pairs, const uint64_t *masks, const int32_t *rule,
                                        double *data) {
    if (!ctx) return PNL_ERR_INVALID;
    if (np < 0 || (np && (!pairs || !masks || !rule))) return fail(ctx, PNL_ERR_INVALID, "bad pair list");
    int rc;
    if ((rc = pnl_pw_prepare(ctx, 0))) return rc;
    const int nV = ctx->dim+1;
    // split by kind on the host: touching items with the key of their near rule, the others for the device classification; every
    // index is checked here (a wrong one would be an out-of-bounds access on the device)
    std::vector<int32_t> touching, distant, titems;
    for (int t = 0; t < np; t++) {
        const int c1 = pairs[2*(size_t)t], c2 = pairs[2*(size_t)t+1];
        if (c1 < 0 || c1 >= ctx->nc || c2 < 0 || c2 >= ctx->nc) return fail(ctx, PNL_ERR_INVALID, "pair %d = (%d, %d): not cells", t, c1, c2);
        int common = 0;
        for (int a = 0; a < nV; a++)
            for (int b = 0; b < nV; b++) common += ctx->cells[(size_t)c1*nV+a] == ctx->cells[(size_t)c2*nV+b];
        if (common > 0) {
            if (rule[t] < 0 || rule[t] >= ctx->pw_nkeys[0]) return fail(ctx, PNL_ERR_INVALID, "touching pair %d: rule key %d out of range", t, rule[t]);
            touching.push_back(c1); touching.push_back(c2); touching.push_back(common); touching.push_back(rule[t]);
            titems.push_back(t);
        } else {
            if (rule[t] >= 0) return fail(ctx, PNL_ERR_INVALID, "pair %d has no common vertex but names a near rule", t);
            distant.push_back(t);
        }
    }
    if ((rc = upload(ctx, ctx->b_mp_pairs, pairs, (size_t)2*np))) return rc;
    if ((rc = upload(ctx, ctx->b_mp_masks, masks, (size_t)4*np))) return rc;
    if ((rc = upload(ctx, ctx->b_pw_pairs, touching.data(), touching.size()))) return rc;
    if ((rc = upload(ctx, ctx->b_mp_wl, titems.data(), titems.size()))) return rc;
    if ((rc = upload(ctx, ctx->b_mp_sorted, distant.data(), distant.size()))) return rc;
    PwNear NR;
    if ((rc = near_pattern(ctx, data, NR))) return rc;
    const int nt = (int)titems.size(), nd = (int)distant.size();
    if (ctx->dim == 2) return ctx->dpe == 6 ? pairs_near_impl<2, 6>(ctx, np, nt, nd, NR) : pairs_near_impl<2, 3>(ctx, np, nt, nd, NR);
    return ctx->dpe == 3 ? pairs_near_impl<1, 3>(ctx, np, nt, nd, NR) : pairs_near_impl<1, 2>(ctx, np, nt, nd, NR);
}

int pnl_assemble_boundary_masked_pointwise(pnl_context *ctx, int ni, const int32_t *cells, const int32_t *facets, const uint32_t *masks,
                                           const int32_t *rule, const double *sv, double fac, double *data, double *diag) {
    if (!ctx) return PNL_ERR_INVALID;
    if (ni < 0 || (ni && (!cells || !facets || !masks || !rule || !sv))) return fail(ctx, PNL_ERR_INVALID, "bad item list");
    int rc;
    if ((rc = pnl_pw_prepare(ctx, 0))) return rc;
    for (int s = 0; s < ctx->dim; s++)
        if (!ctx->have_pw_rules[1][s]) return fail(ctx, PNL_ERR_STATE, "pointwise boundary rule for %d common vertices not uploaded", s+1);
    const int dim = ctx->dim, nV = dim+1;
    for (int t = 0; t < ni; t++) {
        if (cells[t] < 0 || cells[t] >= ctx->nc) return fail(ctx, PNL_ERR_INVALID, "item %d: bad cell", t);
        int common = 0;
        for (int k = 0; k < dim; k++) {
            const int v = facets[(size_t)t*dim+k];
            if (v < 0 || v >= ctx->nv) return fail(ctx, PNL_ERR_INVALID, "item %d: bad facet vertex", t);
            for (int a = 0; a < nV; a++) common += ctx->cells[(size_t)cells[t]*nV+a] == v;
        }
        if (common > 0 ? (rule[t] < 0 || rule[t] >= ctx->pw_nkeys[1]) : rule[t] >= 0)
            return fail(ctx, PNL_ERR_INVALID, "item %d: %d common vertices but rule key %d", t, common, rule[t]);
        if (!(sv[t] > 0.) || !(sv[t] < 1.)) return fail(ctx, PNL_ERR_INVALID, "item %d: order %g outside (0, 1)", t, sv[t]);
    }
    if ((rc = upload(ctx, ctx->b_vertices, ctx->vertices.data(), ctx->vertices.size()))) return rc;
    if ((rc = upload(ctx, ctx->b_bi_cells, cells, (size_t)ni))) return rc;
    if ((rc = upload(ctx, ctx->b_bi_facets, facets, (size_t)ni*dim))) return rc;
    if ((rc = upload(ctx, ctx->b_bi_masks, masks, (size_t)ni))) return rc;
    if ((rc = upload(ctx, ctx->b_mp_aux, rule, (size_t)ni))) return rc;
    if ((rc = upload(ctx, ctx->b_mp_sorted, sv, (size_t)ni))) return rc;
    if (ctx->sp_nnz < 0) return fail(ctx, PNL_ERR_STATE, "upload the sparsity pattern first");
    if (!data && ctx->sp_nnz > 0) return fail(ctx, PNL_ERR_INVALID, "null output");
    SparseOut S;
    S.indptr = (const int*)ctx->b_sp_indptr.p; S.indices = (const int*)ctx->b_sp_indices.p;
    S.data = data; S.diag = diag; S.pairs = nullptr; S.masks = nullptr;
    if (ni == 0) return PNL_OK;
    if (dim == 2) return ctx->dpe == 6 ? boundary_near_impl<2, 6>(ctx, ni, fac, S) : boundary_near_impl<2, 3>(ctx, ni, fac, S);
    return ctx->dpe == 3 ? boundary_near_impl<1, 3>(ctx, ni, fac, S) : boundary_near_impl<1, 2>(ctx, ni, fac, S);
}

}  // extern "C"
