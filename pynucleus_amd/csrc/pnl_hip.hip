// libpnl_hip.so -- host side of the C ABI declared in include/pnl_hip.h (gfx950 only).
//
// Owns the HBM copies of mesh / DoF map / quadrature tables (flattened to SoA), derives the
// work lists (tiles, touching cell pairs, touching cell/facet pairs) and launches the kernels of
// pnl_kernels.h on one HIP stream.  No Python or torch types cross this boundary.
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <string>
#include <vector>
#include "pnl_hip.h"
#include "pnl_kernels.h"
#include "pnl_pointwise.h"
#include "pnl_bndtile.h"

#include <thread>
#include "pnl_context.h"

// vertex order of the cells for the tile kernels (see finalize): search on host threads, tables on first use
struct TileOrderJob {
    std::vector<int> cells, lperm;
    int nc = 0, T = 0, nblocks = 0, ncp = 0, dim = 0, dpe = 0;
    std::vector<double> cellv;
    std::vector<int32_t> cdof;
    std::vector<int16_t> cslot;
    std::thread worker;
};

namespace {


void tile_order_drop(pnl_context *ctx) {
    if (!ctx->tile_job) return;
    if (ctx->tile_job->worker.joinable()) ctx->tile_job->worker.join();
    delete ctx->tile_job;
    ctx->tile_job = nullptr;
}

int tile_order_ready(pnl_context *ctx);
}  // namespace

// other translation units (pnl_slab.hip): the Dt half of the per-cell diagonal blocks exists iff the permuted tables do; a context
// that was finalized again since the assembly (setKernel, pnl_set_cell_order) holds the flag false until the job is joined
int pnl_tile_order_ready(pnl_context *ctx) { return tile_order_ready(ctx); }

namespace {
int tile_order_ready(pnl_context *ctx) {
    TileOrderJob *job = ctx->tile_job;
    if (!job) return PNL_OK;
    if (job->worker.joinable()) job->worker.join();
    const int nV = job->dim+1, NC = nV*job->dim, ncp = job->ncp, dpe = job->dpe, dim = job->dim;
    std::vector<double> cellv_t((size_t)NC*ncp, 0.);
    std::vector<int32_t> cdof_t((size_t)dpe*ncp, -1);
    std::vector<int16_t> cslot_t((size_t)dpe*ncp, -1);
    for (int c = 0; c < job->nc; c++)
        for (int k = 0; k < nV; k++) {
            const int src = job->lperm[(size_t)c*nV+k];
            for (int d = 0; d < dim; d++) cellv_t[(size_t)(k*dim+d)*ncp+c] = job->cellv[(size_t)(src*dim+d)*ncp+c];
            cdof_t[(size_t)k*ncp+c] = job->cdof[(size_t)src*ncp+c];
            cslot_t[(size_t)k*ncp+c] = job->cslot[(size_t)src*ncp+c];
        }
    int rc2;
    if ((rc2 = upload(ctx, ctx->b_cellv_t, cellv_t.data(), cellv_t.size()))) return rc2;
    if ((rc2 = upload(ctx, ctx->b_cdof_t, cdof_t.data(), cdof_t.size()))) return rc2;
    if ((rc2 = upload(ctx, ctx->b_cslot_t, cslot_t.data(), cslot_t.size()))) return rc2;
    if ((rc2 = ensure(ctx, ctx->b_Dt, sizeof(double)*(size_t)ncp*(dpe*(dpe+1)/2)))) return rc2;
    ctx->have_tile_order = true;
    delete job;
    ctx->tile_job = nullptr;
    return PNL_OK;
}

// Build everything derived from mesh + DoF map: padded SoA cell arrays, per-block unique DoF lists,
// touching cell pairs (NO:311-323 shared-vertex test, done once through the vertex->cell adjacency),
// touching cell/facet pairs.
int finalize(pnl_context *ctx) {
    if (!ctx->dirty) return PNL_OK;
    if (!ctx->have_mesh || !ctx->have_dofs) return fail(ctx, PNL_ERR_STATE, "mesh and DoF map must be uploaded first");
    const int dim = ctx->dim, nV = dim+1, nc = ctx->nc, dpe = ctx->dpe;
    if (!((dim == 2 && (dpe == 1 || dpe == 3 || dpe == 6)) || (dim == 1 && dpe >= 1 && dpe <= 4)))          // 2D: P0, P1, P2; 1D: P0 .. P3
        return fail(ctx, PNL_ERR_UNSUPPORTED, "unsupported (dim=%d, dofs_per_element=%d)", dim, dpe);
    // P2 blocks hold about twice the DoFs per cell: half the cells per block keep the LDS sub-block in range
    const int T = ctx->tile = (dpe == 6 || (dim == 1 && dpe >= 3)) ? TILE_P2 : TILE_P1;
    const int nblocks = ctx->nblocks = (nc+T-1)/T;
    const int ncp = ctx->ncp = nblocks*T;
    const int NC = nV*dim;
    std::vector<double> cellv((size_t)NC*ncp, 0.), ccen((size_t)dim*ncp, 0.), cvol(ncp, 0.), ch(ncp, 1.);
    std::vector<int32_t> cvid((size_t)nV*ncp), cdof((size_t)dpe*ncp, -1);
    // P1: for DISTANT pairs the local vertex order of a cell is free (symmetric rules; touching pairs keep the reference's
    // order, their rules are not invariant).  The tile kernels read a copy of the cell tables in which the order is chosen
    // per cell so that within a block of T cells a vertex appears at every local position about equally often: the lanes of
    // a ds_add_f64 then hit the same LDS address ~2.3 instead of ~4.3 times (greedy + local search).
    // The search runs on host threads of its own (0.24 s on one thread at 98,304 cells, 30-50 ms on eight) and is waited for by the
    // first dense assembly (tile_order_ready): the near-field / H2 path never reads the permuted tables, and a dense assembly
    // overlaps it with the rest of this function and with the uploads of rules and kernels.
    const bool reorder = dpe == nV && dim == 2 && !pnl_tune("PNL_NO_REORDER");
    tile_order_drop(ctx);
    if (reorder) {
        TileOrderJob *job = new TileOrderJob;
        ctx->tile_job = job;
        job->cells = ctx->cells;
        job->nc = nc; job->T = T; job->nblocks = nblocks; job->ncp = ncp; job->dim = dim; job->dpe = dpe;
        job->lperm.resize((size_t)nc*nV);
        for (int c = 0; c < nc; c++) for (int k = 0; k < nV; k++) job->lperm[(size_t)c*nV+k] = k;
        job->worker = std::thread([job]() {
        static const int perms[6][3] = {{0, 1, 2}, {1, 2, 0}, {2, 0, 1}, {0, 2, 1}, {2, 1, 0}, {1, 0, 2}};
        const int nc = job->nc, T = job->T, nblocks = job->nblocks;
        const std::vector<int> &cells = job->cells;
        std::vector<int> &lperm = job->lperm;
        // the blocks are independent: a few host threads
        const int nthr = (int)std::max(1u, std::min(8u, std::thread::hardware_concurrency()));
        std::vector<std::thread> pool;
        for (int th = 0; th < nthr; th++) pool.emplace_back([&, th]() {
        for (int b = th; b < nblocks; b += nthr) {
            std::vector<std::pair<long long, int>> seen;   // (vertex*3+pos) -> count, small per block
            auto get = [&](int v, int pos) { for (auto &e : seen) if (e.first == (long long)v*3+pos) return e.second; return 0; };
            auto add = [&](int v, int pos, int d) { for (auto &e : seen) if (e.first == (long long)v*3+pos) { e.second += d; return; } seen.push_back({(long long)v*3+pos, d}); };
            // greedy pass, then a few sweeps of local search (every cell re-chooses its order given all the others)
            for (int sweep = 0; sweep < 5; sweep++)
                for (int c = b*T; c < std::min(nc, (b+1)*T); c++) {
                    if (sweep) for (int k = 0; k < 3; k++) add(cells[(size_t)c*3+lperm[(size_t)c*3+k]], k, -1);
                    int best = 0, bestmax = 1 << 30, bestsum = 1 << 30;
                    for (int p = 0; p < 6; p++) {
                        int mx = 0, sum = 0;
                        for (int k = 0; k < 3; k++) { const int g = get(cells[(size_t)c*3+perms[p][k]], k); mx = std::max(mx, g); sum += g*g; }
                        if (mx < bestmax || (mx == bestmax && sum < bestsum)) { best = p; bestmax = mx; bestsum = sum; }
                    }
                    for (int k = 0; k < 3; k++) { lperm[(size_t)c*3+k] = perms[best][k]; add(cells[(size_t)c*3+perms[best][k]], k, 1); }
                }
        }
        });
        for (auto &t : pool) t.join();
        });
    }
    // Cells of volume ZERO are padding inside the mesh (builder.label_blocks: a block of cells that would straddle an interface
    // of a piecewise-constant order is split into one block per label, filled up with zero-volume copies of its own cells).
    // They keep their geometry (every kernel value stays finite), carry no DoFs, negative vertex ids like the padding behind
    // the last cell (the tile kernels drop every pair that holds one), and belong to no touching pair.
    ctx->nreal = 0;
    std::vector<char> dummy(nc, 0);
    for (int c = 0; c < nc; c++) { dummy[c] = ctx->vol[c] == 0.; ctx->nreal += !dummy[c]; }
    ctx->real_from.assign((size_t)nc+1, 0);                  // number of real cells with index >= c
    for (int c = nc-1; c >= 0; c--) ctx->real_from[c] = ctx->real_from[c+1]+(dummy[c] ? 0 : 1);
    for (int c = 0; c < ncp; c++) {
        for (int k = 0; k < nV; k++) cvid[(size_t)k*ncp+c] = -1-k;
        if (c >= nc) continue;
        double cen[2] = {0., 0.};
        for (int k = 0; k < nV; k++) {
            const int v = ctx->cells[(size_t)c*nV+k];
            if (v < 0 || v >= ctx->nv) return fail(ctx, PNL_ERR_INVALID, "cell %d references vertex %d", c, v);
            if (!dummy[c]) cvid[(size_t)k*ncp+c] = v;
            for (int d = 0; d < dim; d++) {
                const double x = ctx->vertices[(size_t)v*dim+d];
                // the quadrature points and the cut-element geometry live in the coordinates of the interaction transform (all
                // they enter is differences x - y: |T (x - y)|, interactionDomains.pyx:1417-1470); centres, h and volumes -- the order
                // formula and the measure -- stay those of the mesh
                double xt = x;
                if (ctx->have_xform && dim == 2) xt = ctx->xform[2*d]*ctx->vertices[(size_t)v*dim]+ctx->xform[2*d+1]*ctx->vertices[(size_t)v*dim+1];
                cellv[(size_t)(k*dim+d)*ncp+c] = xt;
                cen[d] += x;
            }
        }
        const double fac = 1./nV;                    // NO:116-126
        for (int d = 0; d < dim; d++) ccen[(size_t)d*ncp+c] = cen[d]*fac;
        cvol[c] = ctx->vol[c];
        ch[c] = ctx->h[c];
        for (int k = 0; k < dpe; k++) {
            const int g = ctx->dofs[(size_t)c*dpe+k];
            if (g >= ctx->N) return fail(ctx, PNL_ERR_INVALID, "DoF id %d >= num_dofs %d", g, ctx->N);
            if (dummy[c] && g >= 0) return fail(ctx, PNL_ERR_INVALID, "cell %d has volume zero and a DoF", c);
            cdof[(size_t)k*ncp+c] = g;
        }
    }
    // unique DoFs per block
    std::vector<std::vector<int>> lists(nblocks);
    int nU = 1;
    for (int b = 0; b < nblocks; b++) {
        auto &L = lists[b];
        for (int c = b*T; c < std::min(nc, (b+1)*T); c++)
            for (int k = 0; k < dpe; k++) {
                const int g = ctx->dofs[(size_t)c*dpe+k];
                if (g >= 0) L.push_back(g);
            }
        std::sort(L.begin(), L.end());
        L.erase(std::unique(L.begin(), L.end()), L.end());
        nU = std::max<int>(nU, (int)L.size());
    }
    ctx->nU = nU;
    std::vector<int32_t> blk_ndof(nblocks), blk_dofs((size_t)nblocks*nU, 0);
    std::vector<int16_t> cslot((size_t)dpe*ncp, -1);
    for (int b = 0; b < nblocks; b++) {
        auto &L = lists[b];
        blk_ndof[b] = (int)L.size();
        std::copy(L.begin(), L.end(), blk_dofs.begin()+(size_t)b*nU);
        for (int c = b*T; c < std::min(nc, (b+1)*T); c++)
            for (int k = 0; k < dpe; k++) {
                const int g = cdof[(size_t)k*ncp+c];
                if (g >= 0) cslot[(size_t)k*ncp+c] = (int16_t)(std::lower_bound(L.begin(), L.end(), g)-L.begin());
            }
    }
    // permuted copies for the tile kernels: built and uploaded by tile_order_ready once the search above has finished
    ctx->have_tile_order = false;
    if (reorder) { ctx->tile_job->cellv = cellv; ctx->tile_job->cdof = cdof; ctx->tile_job->cslot = cslot; }
    // touching cell pairs via vertex -> cells adjacency
    std::vector<int> vptr(ctx->nv+1, 0);
    for (int c = 0; c < nc; c++)
        for (int k = 0; k < nV && !dummy[c]; k++) vptr[ctx->cells[(size_t)c*nV+k]+1]++;
    for (int v = 0; v < ctx->nv; v++) vptr[v+1] += vptr[v];
    std::vector<int> vcells(vptr[ctx->nv]), fill(vptr.begin(), vptr.end()-1);
    for (int c = 0; c < nc; c++)
        for (int k = 0; k < nV && !dummy[c]; k++) vcells[fill[ctx->cells[(size_t)c*nV+k]]++] = c;
    for (int s = 0; s < 3; s++) ctx->spairs_host[s].clear();
    {
        std::vector<int> nbr;
        for (int c1 = 0; c1 < nc; c1++) {
            nbr.clear();
            if (dummy[c1]) continue;
            for (int k = 0; k < nV; k++) {
                const int v = ctx->cells[(size_t)c1*nV+k];
                for (int t = vptr[v]; t < vptr[v+1]; t++)
                    if (vcells[t] >= c1) nbr.push_back(vcells[t]);
            }
            std::sort(nbr.begin(), nbr.end());
            for (size_t t = 0; t < nbr.size();) {
                size_t u = t;
                while (u < nbr.size() && nbr[u] == nbr[t]) u++;
                const int common = (nbr[t] == c1) ? nV : (int)(u-t);
                if (common < 1 || common > nV) return fail(ctx, PNL_ERR_INVALID, "degenerate cell pair (%d,%d)", c1, nbr[t]);
                // which cell is cellNo1 of a touching pair decides the orientation of its singular rule (NA:1386-1396: c1 <= c2 in
                // the CALLER's numbering): renumbered cells keep it (pnl_set_cell_order)
                if (!ctx->cell_orig.empty() && ctx->cell_orig[c1] > ctx->cell_orig[nbr[t]])
                    ctx->spairs_host[common-1].push_back(make_int2(nbr[t], c1));
                else
                    ctx->spairs_host[common-1].push_back(make_int2(c1, nbr[t]));
                t = u;
            }
        }
    }
    int rc;
    const int ncls = (int)ctx->cls.size(), nlab = ctx->nlab;
    if (nlab > 0 && (int)ctx->cell_labels.size() != nc) return fail(ctx, PNL_ERR_STATE, "cell labels do not match the mesh");
    // class of a cell pair / cell-facet pair (Kernel.evalParams at the two centres, NO:509-513)
    auto class_cc = [&](int c1, int c2) { return nlab ? ctx->cls_of[(size_t)ctx->cell_labels[c1]*nlab+ctx->cell_labels[c2]] : 0; };
    auto class_cf = [&](int c1, int f) { return nlab ? ctx->cls_of[(size_t)ctx->cell_labels[c1]*nlab+ctx->facet_labels[f]] : 0; };
    for (int s = 0; s < 3; s++)
        for (int k = 0; k < ncls; k++) {
            std::vector<int2> mine;
            for (const int2 &pr : ctx->spairs_host[s])
                if (class_cc(pr.x, pr.y) == k) mine.push_back(pr);
            ctx->cls[k]->n_spairs[s] = (int)mine.size();
            if ((rc = upload(ctx, ctx->cls[k]->b_spairs[s], mine.data(), mine.size()))) return rc;
            ctx->cls[k]->n_spairs1[s] = 0;
            if (ctx->nonsym) {
                // second orientation (swapCells, NA:1418): the pair (c2, c1) with the class of (label c2, label c1); identical
                // pairs are visited once
                std::vector<int2> swapped;
                for (const int2 &pr : ctx->spairs_host[s])
                    if (pr.x != pr.y && class_cc(pr.y, pr.x) == k) swapped.push_back(make_int2(pr.y, pr.x));
                ctx->cls[k]->n_spairs1[s] = (int)swapped.size();
                if ((rc = upload(ctx, ctx->cls[k]->b_spairs1[s], swapped.data(), swapped.size()))) return rc;
            }
        }
    // boundary facets
    for (int k = 0; k < ncls; k++) ctx->cls[k]->n_bpairs[0] = ctx->cls[k]->n_bpairs[1] = 0;
    if (ctx->have_boundary) {
        if (nlab > 0 && (int)ctx->facet_labels.size() != ctx->nb) return fail(ctx, PNL_ERR_STATE, "facet labels do not match the boundary");
        const int nF = dim, nb = ctx->nb;
        std::vector<int32_t> bvid((size_t)nF*nb);
        std::vector<double> bv((size_t)nF*dim*nb);
        std::vector<int2> bp[2];
        for (int f = 0; f < nb; f++) {
            for (int k = 0; k < nF; k++) {
                const int v = ctx->bcells[(size_t)f*nF+k];
                if (v < 0 || v >= ctx->nv) return fail(ctx, PNL_ERR_INVALID, "facet %d references vertex %d", f, v);
                bvid[(size_t)k*nb+f] = v;
                for (int d = 0; d < dim; d++) bv[(size_t)(k*dim+d)*nb+f] = ctx->vertices[(size_t)v*dim+d];
            }
        }
        {
            std::vector<int> nbr;
            for (int f = 0; f < nb; f++) {
                nbr.clear();
                for (int k = 0; k < nF; k++) {
                    const int v = ctx->bcells[(size_t)f*nF+k];
                    for (int t = vptr[v]; t < vptr[v+1]; t++) nbr.push_back(vcells[t]);
                }
                std::sort(nbr.begin(), nbr.end());
                for (size_t t = 0; t < nbr.size();) {
                    size_t u = t;
                    while (u < nbr.size() && nbr[u] == nbr[t]) u++;
                    const int common = (int)(u-t);
                    if (common > nF) return fail(ctx, PNL_ERR_INVALID, "degenerate cell/facet pair");
                    bp[common-1].push_back(make_int2(nbr[t], f));
                    t = u;
                }
            }
        }
        for (int s = 0; s < 2; s++)
            for (int k = 0; k < ncls; k++) {
                std::vector<int2> mine;
                for (const int2 &pr : bp[s])
                    if (class_cf(pr.x, pr.y) == k) mine.push_back(pr);
                ctx->cls[k]->n_bpairs[s] = (int)mine.size();
                if ((rc = upload(ctx, ctx->cls[k]->b_bpairs[s], mine.data(), mine.size()))) return rc;
            }
        if (nlab > 0 && (rc = upload(ctx, ctx->b_blabel, ctx->facet_labels.data(), ctx->facet_labels.size()))) return rc;
        if ((rc = upload(ctx, ctx->b_bvid, bvid.data(), bvid.size()))) return rc;
        if ((rc = upload(ctx, ctx->b_bv, bv.data(), bv.size()))) return rc;
        {
            // per-facet geometry used by every (cell, facet) pair: NO:1049-1055 normal, get_h_surface_simplex
            std::vector<double> geo((size_t)(2*dim+3)*nb, 0.);
            for (int f = 0; f < nb; f++) {
                double len = 1.;
                for (int d = 0; d < dim; d++) {
                    double sum = 0.;
                    for (int k = 0; k < nF; k++) sum += bv[(size_t)(k*dim+d)*nb+f];
                    geo[(size_t)d*nb+f] = sum*(1./nF);
                }
                if (dim == 2) {
                    double n0 = bv[(size_t)(1*dim+1)*nb+f]-bv[(size_t)(0*dim+1)*nb+f];
                    double n1 = bv[(size_t)(0*dim+0)*nb+f]-bv[(size_t)(1*dim+0)*nb+f];
                    const double inv = 1./std::sqrt(n0*n0+n1*n1);
                    geo[(size_t)(dim+0)*nb+f] = n0*inv;
                    geo[(size_t)(dim+1)*nb+f] = n1*inv;
                    const double dx = bv[(size_t)2*nb+f]-bv[(size_t)0*nb+f], dy = bv[(size_t)3*nb+f]-bv[(size_t)1*nb+f];
                    len = std::sqrt(dx*dx+dy*dy);
                }
                geo[(size_t)(2*dim)*nb+f] = len;
                geo[(size_t)(2*dim+1)*nb+f] = std::fabs(std::log(len/ctx->H0));
                geo[(size_t)(2*dim+2)*nb+f] = std::log(len);
            }
            if ((rc = upload(ctx, ctx->b_bgeo, geo.data(), geo.size()))) return rc;
        }
    }
    if ((rc = upload(ctx, ctx->b_vertices, ctx->vertices.data(), ctx->vertices.size()))) return rc;
    if (nlab > 0) {
        std::vector<int32_t> cl(ncp, 0);
        std::copy(ctx->cell_labels.begin(), ctx->cell_labels.end(), cl.begin());
        if ((rc = upload(ctx, ctx->b_clabel, cl.data(), cl.size()))) return rc;
        if ((rc = upload(ctx, ctx->b_clsof, ctx->cls_of.data(), ctx->cls_of.size()))) return rc;
    }
    if ((rc = upload(ctx, ctx->b_cellv, cellv.data(), cellv.size()))) return rc;
    if ((rc = upload(ctx, ctx->b_ccen, ccen.data(), ccen.size()))) return rc;
    if ((rc = upload(ctx, ctx->b_cvol, cvol.data(), cvol.size()))) return rc;
    if ((rc = upload(ctx, ctx->b_ch, ch.data(), ch.size()))) return rc;
    {
        std::vector<double> clog((size_t)3*ncp, 0.);
        for (int c = 0; c < ncp; c++) { clog[c] = std::log(ch[c]); clog[(size_t)ncp+c] = std::fabs(std::log(ch[c]/ctx->H0)); }
        // third row: the largest distance centre -- vertex of the cell in mesh coordinates, rounded up.  Two cells that share a
        // vertex have their centres within the sum of these radii: the tile kernels compare vertex ids only for such pairs
        for (int c = 0; c < nc; c++) {
            double r2 = 0.;
            for (int k = 0; k < nV; k++) {
                const int v = ctx->cells[(size_t)c*nV+k];
                double t2 = 0.;
                for (int d = 0; d < dim; d++) { const double t = ctx->vertices[(size_t)v*dim+d]-ccen[(size_t)d*ncp+c]; t2 += t*t; }
                r2 = std::max(r2, t2);
            }
            clog[(size_t)2*ncp+c] = std::sqrt(r2)*(1.+1e-5);
        }
        if ((rc = upload(ctx, ctx->b_clog, clog.data(), clog.size()))) return rc;
        // per-block aggregates for the host-side tile classification (uniform tiles)
        ctx->blocks.assign(nblocks, pnl_context::BlockAgg{0., 0., 0., 0., 0., 0., 0., false, 0., 0., 0.});
        for (int b = 0; b < nblocks; b++) {
            auto &B = ctx->blocks[b];
            const int c0 = b*T, c1 = std::min(nc, (b+1)*T);
            B.full = (c1-c0 == T);
            double sx = 0., sy = 0.;
            for (int c = c0; c < c1; c++) { sx += ccen[c]; if (dim == 2) sy += ccen[(size_t)ncp+c]; }
            B.cx = sx/(c1-c0); B.cy = sy/(c1-c0);
            B.rad = 0.; B.hmax = 0.; B.hmin = 1e300; B.Lmin = 1e300; B.Lmax = -1e300;
            for (int c = c0; c < c1; c++) {
                const double dx = ccen[c]-B.cx, dy = dim == 2 ? ccen[(size_t)ncp+c]-B.cy : 0.;
                B.rad = std::max(B.rad, std::sqrt(dx*dx+dy*dy));
                B.hmax = std::max(B.hmax, ch[c]);
                B.hmin = std::min(B.hmin, ch[c]);
                B.Lmin = std::min(B.Lmin, clog[(size_t)ncp+c]);
                B.Lmax = std::max(B.Lmax, clog[(size_t)ncp+c]);
            }
            // the same ball around the block in the coordinates of the interaction transform, vertices included
            B.tcx = B.cx; B.tcy = B.cy; B.trad = B.rad+B.hmax;
            if (ctx->have_xform && dim == 2) {
                double tx = 0., ty = 0.;
                int nvb = 0;
                for (int c = c0; c < c1; c++) for (int k = 0; k < nV; k++) { tx += cellv[(size_t)(k*dim)*ncp+c]; ty += cellv[(size_t)(k*dim+1)*ncp+c]; nvb++; }
                B.tcx = tx/nvb; B.tcy = ty/nvb; B.trad = 0.;
                for (int c = c0; c < c1; c++) for (int k = 0; k < nV; k++) {
                    const double dx = cellv[(size_t)(k*dim)*ncp+c]-B.tcx, dy = cellv[(size_t)(k*dim+1)*ncp+c]-B.tcy;
                    B.trad = std::max(B.trad, std::sqrt(dx*dx+dy*dy));
                }
            }
        }
        ctx->tiles_cached.clear(); ctx->tiles_cb = -1;
    }
    if ((rc = upload(ctx, ctx->b_cvid, cvid.data(), cvid.size()))) return rc;
    if ((rc = upload(ctx, ctx->b_cdof, cdof.data(), cdof.size()))) return rc;
    if ((rc = upload(ctx, ctx->b_cslot, cslot.data(), cslot.size()))) return rc;
    if ((rc = upload(ctx, ctx->b_blk_ndof, blk_ndof.data(), blk_ndof.size()))) return rc;
    if ((rc = upload(ctx, ctx->b_blk_dofs, blk_dofs.data(), blk_dofs.size()))) return rc;
    {
        // block-slot storage (pnl_tile2.h): padded column offsets, row offsets, the copies (block, slot) of every DoF
        std::vector<int32_t> colbase(nblocks+1, 0);
        for (int b = 0; b < nblocks; b++) colbase[b+1] = colbase[b]+((blk_ndof[b]+7) & ~7);
        const int S = colbase[nblocks];
        std::vector<long long> rowoff(nblocks, 0);
        long long run = 0;
        for (int a = 0; a < nblocks; a++) { rowoff[a] = run; run += (long long)blk_ndof[a]*(S-colbase[a]); }
        ctx->slot_S = S; ctx->slot_total = run;
        std::vector<int32_t> cpoff(ctx->N+1, 0);
        for (int b = 0; b < nblocks; b++) for (int g : lists[b]) cpoff[g+1]++;
        for (int g = 0; g < ctx->N; g++) cpoff[g+1] += cpoff[g];
        std::vector<int2> cp(cpoff[ctx->N]);
        std::vector<int32_t> fillp(cpoff.begin(), cpoff.end()-1);
        std::vector<long long> cprow(cp.size());
        for (int b = 0; b < nblocks; b++)
            for (size_t r = 0; r < lists[b].size(); r++) {
                const int k = fillp[lists[b][r]]++;
                cp[k] = make_int2(b, colbase[b]+(int)r);
                cprow[k] = rowoff[b]+(long long)r*(S-colbase[b])-colbase[b];
            }
        if ((rc = upload(ctx, ctx->b_cprow, cprow.data(), cprow.size()))) return rc;
        if ((rc = upload(ctx, ctx->b_scolbase, colbase.data(), colbase.size()))) return rc;
        if ((rc = upload(ctx, ctx->b_srowoff, rowoff.data(), rowoff.size()))) return rc;
        if ((rc = upload(ctx, ctx->b_cpoff, cpoff.data(), cpoff.size()))) return rc;
        if ((rc = upload(ctx, ctx->b_cpslot, cp.data(), cp.size()))) return rc;
        // the same tables packed per range of 32 DoFs (k_fold_mirror loads a range with one round trip): header (count) +
        // PNL_FOLD_TAB entries (row offset, block << 5 | local DoF, slot column); ranges with more copies use the lists above
        const int nranges = (ctx->N+31)/32;
        std::vector<FoldEntry> tab((size_t)nranges*(PNL_FOLD_TAB+1));
        for (int q = 0; q < nranges; q++) {
            FoldEntry *e = &tab[(size_t)q*(PNL_FOLD_TAB+1)];
            const int g0 = cpoff[q*32], g1 = cpoff[std::min(ctx->N, q*32+32)];
            e[0].off = g1-g0; e[0].ar = 0; e[0].cy = 0;
            if (g1-g0 > PNL_FOLD_TAB) continue;
            for (int I = q*32; I < std::min(ctx->N, q*32+32); I++)
                for (int g = cpoff[I]; g < cpoff[I+1]; g++) {
                    FoldEntry &x = e[1+g-g0];
                    x.off = cprow[g]; x.ar = (cp[g].x << 5) | (I-q*32); x.cy = cp[g].y;
                }
        }
        if ((rc = upload(ctx, ctx->b_foldtab, tab.data(), tab.size()))) return rc;
    }
    if ((rc = upload(ctx, ctx->b_perm, ctx->perm_table.data(), ctx->perm_table.size()))) return rc;
    const bool fresh_counters = !ctx->b_counters.p;
    if ((rc = ensure(ctx, ctx->b_counters, sizeof(unsigned long long)*PNL_NCOUNTERS))) return rc;
    // pnl_synchronize reads the loss counters of a context that may never assemble (an operator installed by pnl_h2_set)
    if (fresh_counters) HIPCHK(ctx, hipMemset(ctx->b_counters.p, 0, sizeof(unsigned long long)*PNL_NCOUNTERS));
    if ((rc = ensure(ctx, ctx->b_D, sizeof(double)*(size_t)ncp*(dpe*(dpe+1)/2)))) return rc;

    DevProblem &P = ctx->P;
    P.dim = dim; P.dpe = dpe; P.nc = nc; P.ncp = ncp; P.N = ctx->N; P.dpv = ctx->dpv; P.dped = ctx->dped; P.nb = ctx->nb;
    P.H0 = ctx->H0;
    P.cellv = (const double*)ctx->b_cellv.p; P.ccen = (const double*)ctx->b_ccen.p;
    P.cvol = (const double*)ctx->b_cvol.p; P.ch = (const double*)ctx->b_ch.p; P.clog = (const double*)ctx->b_clog.p;
    P.cvid = (const int*)ctx->b_cvid.p; P.cdof = (const int*)ctx->b_cdof.p; P.cslot = (const short*)ctx->b_cslot.p;
    P.blk_ndof = (const int*)ctx->b_blk_ndof.p; P.blk_dofs = (const int*)ctx->b_blk_dofs.p;
    P.blk_stride = nU; P.nblocks = nblocks;
    P.perm_table = (const int*)ctx->b_perm.p;
    P.bvid = (const int*)ctx->b_bvid.p; P.bv = (const double*)ctx->b_bv.p; P.bgeo = (const double*)ctx->b_bgeo.p;
    P.counters = (unsigned long long*)ctx->b_counters.p;
    P.nlab = nlab; P.cur_class = -1; P.orient = 0; P.pad2 = 0; P.idfac = 1.;
    P.clabel = (const int*)ctx->b_clabel.p; P.blabel = (const int*)ctx->b_blabel.p; P.cls_of = (const int*)ctx->b_clsof.p;
    ctx->dirty = false;
    return PNL_OK;
}

// Tables of pnl_pow_tab for x^exponent * scale (long double on the host, rounded once):  x = 2^k m, m in [1, 2), j = top seven
// fraction bits of m, c_j = 1 / fl(1 / (1 + (j + 1/2) / 128)), u = m fl(1/c_j) - 1 (one FMA, |u| <= 2^-8):
//   x^e = 2^(e k) c_j^e (1 + u)^e.
// also_fast: tables for an exponent -qm/4 as well -- a launch over the tiles of SEVERAL order classes runs the KT == 0 kernels for
// all of them, and a class without tables falls into the general branch there (exp(e ln x) behind the per-lane horizon test)
static const double *pow_table(pnl_context *ctx, const DevKernel &k, bool also_fast = false) {
    if (k.ktype != PNL_FRACTIONAL || (k.fast && !also_fast) || pnl_tune("PNL_NO_POWTAB")) return nullptr;
    for (auto *t : ctx->powtabs) if (t->exponent == k.exponent && t->scale == k.scale) return (const double*)t->buf.p;
    std::vector<double> tab(PNL_POW_TAB_DOUBLES);
    for (int j = 0; j < 128; j++) {
        const double invc = (double)(1.L/(1.L+((long double)j+0.5L)/128.L));
        const long double c = 1.L/(long double)invc;
        tab[j] = invc;
        tab[128+j] = (double)((long double)k.scale*powl(c, (long double)k.exponent));
        tab[256+j] = (double)exp2l((long double)k.exponent*(long double)(j-96));
    }
    auto *t = new pnl_context::PowTab;
    t->exponent = k.exponent; t->scale = k.scale;
    if (upload(ctx, t->buf, tab.data(), tab.size()) != PNL_OK) { delete t; return nullptr; }
    ctx->powtabs.push_back(t);
    return (const double*)t->buf.p;
}

void refresh_tables(pnl_context *ctx) {
    DevProblem &P = ctx->P;
    P.k = to_dev(ctx->C().kern[0], ctx->dim);
    P.bk = to_dev(ctx->C().kern[1], ctx->dim);
    {
        // n.(y-x)/|y-x| * Gamma_b(|x-y|^2): fold the normalisation into the exponent (fractional kernels only)
        pnl_kernel kn = ctx->C().kern[1];
        if (ctx->dim == 2 && kn.ktype == PNL_FRACTIONAL) kn.exponent -= 0.5;
        P.bkn = to_dev(kn, ctx->dim);
        if (ctx->dim == 2 && kn.ktype != PNL_FRACTIONAL) P.bkn.fast = 0;
        if (ctx->dim == 2 && kn.ktype == PNL_GAUSSIAN_BOUNDARY) P.bkn.ktype = 7;        // kern_eval (pnl_common.h): folded 2D forms
        if (ctx->dim == 2 && kn.ktype == PNL_EXPONENTIAL_BOUNDARY) P.bkn.ktype = 8;
    }
    P.qo = to_dev(ctx->C().form[0]);
    P.bqo = to_dev(ctx->C().form[1]);
    P.qmax = ctx->qmax;
    P.off = (const int*)ctx->b_off.p; P.bary = (const double*)ctx->b_bary.p; P.w = (const double*)ctx->b_w.p;
    P.phi = (const double*)ctx->b_phi.p; P.foff = (const int*)ctx->b_foff.p; P.fbary = (const double*)ctx->b_fbary.p;
    P.fw = (const double*)ctx->b_fw.p;
    P.tt_n = (const int*)ctx->b_ttn.p; P.tt_off = (const int*)ctx->b_ttoff.p; P.tt_tab = (const double*)ctx->b_tttab.p;
    P.tt_wphi = (const double*)ctx->b_ttwphi.p;
    P.tt_wphif = (const double*)ctx->b_ttwphif.p;
    for (int s = 0; s < 3; s++) {
        P.sNodes[s] = (const double*)ctx->C().b_sn[s].p; P.sW[s] = (const double*)ctx->C().b_sw[s].p; P.sPsi[s] = (const double*)ctx->C().b_sp[s].p;
    }
    for (int s = 0; s < 2; s++) {
        P.bNodes[s] = (const double*)ctx->C().b_bn[s].p; P.bW[s] = (const double*)ctx->C().b_bw[s].p; P.bPhi[s] = (const double*)ctx->C().b_bp[s].p;
    }
    for (int s = 0; s < 3; s++) { P.sM[s] = ctx->C().sM[s]; P.sRows[s] = ctx->C().sRows[s]; }
    for (int s = 0; s < 2; s++) P.bM[s] = ctx->C().bM[s];
    P.sFac = ctx->C().sFac; P.bFac = ctx->C().bFac;
    P.cur_class = ctx->nlab > 0 ? ctx->cur : -1;
    // non-symmetric order table: two passes per class with half the kernel each (see DevProblem::orient)
    P.orient = ctx->nonsym ? ctx->orient : 0;
    P.idfac = ctx->nonsym ? 2. : 1.;
    if (ctx->nonsym) P.k.scale *= 0.5;
    P.k.ptab = pow_table(ctx, P.k);                      // after the last change of the scale: the tables carry it
}


template <int DIM, int DPE, int KT>
int launch_pure(pnl_context *ctx, double *A, int64_t ldA, const SlotOut &SO) {
    if (ctx->n_pure == 0) return PNL_OK;
    constexpr int NP = DIM == 2 ? 3 : 2, ND = DPE*(DPE+1)/2;
    const int acc_stride = acc_stride_of(ctx->nU);
    const size_t lds = sizeof(double)*(64*NP*DIM+64+2*64*ND+NP*(4+DPE)+(KT == 0 ? PNL_POW_TAB_DOUBLES : 0))+sizeof(int)*(64*DPE+64)
                       +sizeof(double)*(size_t)(ctx->nU+1)*acc_stride;
    auto kfun = k_tile_pure<DIM, DPE, KT>;
    HIPCHK(ctx, hipFuncSetAttribute((const void*)kfun, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    int per_cu = 2;
    (void)hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, (const void*)kfun, PNL_NTHREADS, lds);
    if (pnl_tune("PNL_VERBOSE")) fprintf(stderr, "[pnl] uniform tiles=%d of %d, lds=%zu bytes, occupancy API: %d blocks/CU\n", ctx->n_pure,
                                       ctx->n_pure+ctx->n_mixed, lds, per_cu);
    const int grid = pnl_grid_cap(std::min(ctx->n_pure, 256*std::max(per_cu, 1)));
    int pure_abl = 0;
#ifdef PNL_DEBUG_ABLATE
    pure_abl = pnl_tune("PNL_PURE_ABL") ? atoi(pnl_tune("PNL_PURE_ABL")) : 0;
#endif
    kt_begin(ctx, PNL_K_TILE_UNIFORM2);
    hipLaunchKernelGGL(kfun, dim3(grid), dim3(PNL_NTHREADS), lds, ctx->stream, tile_problem(ctx), (const int2*)ctx->b_tiles.p+ctx->tile_off+ctx->n_mixed,
                       ctx->n_pure, A, (long long)ldA, (double*)(ctx->have_tile_order ? ctx->b_Dt.p : ctx->b_D.p), acc_stride, 2,
                       (ctx->symflush ? 1 : 0) | pure_abl, SO);
    kt_end(ctx, PNL_K_TILE_UNIFORM2);
    HIPCHK(ctx, hipGetLastError());
    return PNL_OK;
}

// The per-class passes of a variable order run on side streams (pnl_context::aux): fork from the caller's stream, class k on
// stream k mod NAUX (ctx->stream is redirected while a class is being launched), join back.  One class: nothing happens.
struct ClassFork {
    pnl_context *ctx;
    hipStream_t main;
    bool on, used[pnl_context::NAUX] = {};
    hipEvent_t start;
    // from: an event recorded earlier on the caller's stream (the fold pass) -- the side streams start there instead of behind
    // everything the caller's stream holds, so consecutive forked phases run back to back on every side stream
    ClassFork(pnl_context *c, int nclasses, hipEvent_t from = nullptr)
        : ctx(c), main(c->stream), on(nclasses > 1 && !pnl_tune("PNL_NO_FORK")), start(from ? from : c->ev_fork) {
        if (on && !from) (void)hipEventRecord(ctx->ev_fork, main);
    }
    // classes of very different weight (three layers: the pairs inside a layer against the few across an interface): heaviest first
    // onto the least loaded stream, so that two heavy classes do not queue behind each other while a stream of light ones runs dry
    std::vector<int> slot;
    void plan(const std::vector<int> &weight) {
        const int n = (int)weight.size();
        std::vector<int> order(n);
        for (int k = 0; k < n; k++) order[k] = k;
        std::stable_sort(order.begin(), order.end(), [&](int a, int b) { return weight[a] > weight[b]; });
        long long load[pnl_context::NAUX] = {};
        slot.assign(n, 0);
        for (int k : order) {
            int best = 0;
            for (int j = 1; j < pnl_context::NAUX; j++) if (load[j] < load[best]) best = j;
            slot[k] = best; load[best] += std::max(1, weight[k]);
        }
    }
    void use(int k) {
        if (!on) return;
        const int j = k < (int)slot.size() ? slot[k] : k % pnl_context::NAUX;
        if (!used[j]) { (void)hipStreamWaitEvent(ctx->aux[j], start, 0); used[j] = true; }
        ctx->stream = ctx->aux[j];
    }
    void join() {
        if (!on) return;
        ctx->stream = main;
        for (int j = 0; j < pnl_context::NAUX; j++)
            if (used[j]) {
                (void)hipEventRecord(ctx->ev_join[j], ctx->aux[j]);
                (void)hipStreamWaitEvent(main, ctx->ev_join[j], 0);
                used[j] = false;
            }
        on = false;
    }
    ~ClassFork() { join(); }
};

// dynamic LDS of the work-list kernels: the rule copy (+ for P2 the column sums of the PNL_NTHREADS / 16 pairs of a chunk) of
// k_worklist_sorted; for P2 the per-lane column sums of k_worklist_lane (eval_distant_blocked)
template <int DPE>
static int wl_tab_max(int wl_kb) {
    // points of the largest rule that is staged; larger rules are read from global memory, point pair by point pair (slow: at 49,152
    // P2 cells of the 12-sector disc, s = 0.7, 100,000 near pairs take the rules of 240 and 256 points -- 6e9 of the 13e9 kernel values
    // of the work lists).  P2 (one workgroup per CU for its registers anyway): 320 ... 512 points, 80 + 128 bytes of LDS per point.
    const int t = (wl_kb*1024)/((4+DPE)*(int)sizeof(double));
    return wl_csum_lds(DPE) ? std::max(320, std::min(t, 512)) : t;
}
template <int DPE>
static size_t wl_sorted_lds(int tab_max) {
    return sizeof(double)*((size_t)tab_max*(4+DPE)+(wl_csum_lds(DPE) ? (size_t)(PNL_NTHREADS/16)*tab_max : 0));
}
template <typename F>
static size_t wl_lane_lds(F fun, int dpe, int kt) {
    const size_t b = wl_lane_blocked(dpe, kt) ? sizeof(double)*PNL_WL_LANE_MAXPTS*PNL_NTHREADS : 0;
    if (b) (void)hipFuncSetAttribute((const void*)fun, hipFuncAttributeMaxDynamicSharedMemorySize, (int)b);
    return b;
}

// counting sort of a work-list region by order, then the sorted evaluation (k_worklist_lane / k_worklist_sorted);
// region: which copy of the sort buffers to use (passes that may run concurrently need their own)
template <int DIM, int DPE, int KT>
int run_worklist(pnl_context *ctx, const int4 *wl, const unsigned *wlc, unsigned cap, double *A, int64_t ldA, bool sym, int region = 0,
                 int nregions = 1) {
    int rc;
    if ((rc = ensure(ctx, ctx->b_wlsorted, (size_t)cap*nregions*sizeof(int4)))) return rc;
    if ((rc = ensure(ctx, ctx->b_wlaux, sizeof(unsigned)*(4*(PNL_WL_BINS+1))*nregions))) return rc;
    unsigned *hist = (unsigned*)ctx->b_wlaux.p+(size_t)region*4*(PNL_WL_BINS+1), *offs = hist+(PNL_WL_BINS+1), *coff = offs+(PNL_WL_BINS+1),
             *cursor = coff+(PNL_WL_BINS+1);
    int4 *wlsorted = (int4*)ctx->b_wlsorted.p+(size_t)region*cap;
    HIPCHK(ctx, hipMemsetAsync(hist, 0, sizeof(unsigned)*(PNL_WL_BINS+1), ctx->stream));
    hipLaunchKernelGGL(k_wl_hist, dim3(512), dim3(PNL_NTHREADS), 0, ctx->stream, wl, wlc, cap, hist);
    hipLaunchKernelGGL(k_wl_scan, dim3(1), dim3(64), 0, ctx->stream, (const unsigned*)hist, offs, coff, cursor);
    hipLaunchKernelGGL(k_wl_scatter, dim3(512), dim3(PNL_NTHREADS), 0, ctx->stream, wl, wlc, cap, (const unsigned*)offs, cursor,
                       wlsorted);
    const int st = 4+DPE;
    // LDS copy of the rule: 18 KB (8 workgroups per CU; rules with more points are read from global memory; 60 KB / 2 workgroups per CU was 0.6 ms slower at 98,304 cells)
    const int wl_kb = pnl_tune("PNL_WL_LDS_KB") ? std::max(4, atoi(pnl_tune("PNL_WL_LDS_KB"))) : 18;
    const int tab_max = wl_tab_max<DPE>(wl_kb);
    const int wl_grid = 256*std::max(1, std::min(8, 150/(wl_kb+(KT == 0 ? 3 : 0))));      // KT == 0: + 3 KB of power tables
    const size_t lds = wl_sorted_lds<DPE>(tab_max);
    auto wfun = k_worklist_sorted<DIM, DPE, KT, false>;
    HIPCHK(ctx, hipFuncSetAttribute((const void*)wfun, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    const int nmin = ctx->wl_lane ? PNL_WL_LANE_MAXPTS+1 : 0;
    int dbg = 0;
#ifdef PNL_DEBUG_ABLATE
    dbg = pnl_tune("PNL_WL_DBG") ? atoi(pnl_tune("PNL_WL_DBG")) : 0;
#endif
    if (ctx->wl_lane)
        hipLaunchKernelGGL((k_worklist_lane<DIM, DPE, KT, false>), dim3(256*4), dim3(PNL_NTHREADS), wl_lane_lds(k_worklist_lane<DIM, DPE, KT, false>, DPE, KT), ctx->stream, ctx->P,
                           (const int4*)wlsorted, (const unsigned*)offs, A, (long long)ldA, (double*)ctx->b_D.p, SparseOut{},
                           dbg | (sym ? 8 : 0), ClusterTiles{});
    hipLaunchKernelGGL(wfun, dim3(wl_grid), dim3(PNL_NTHREADS), lds, ctx->stream, ctx->P, (const int4*)wlsorted,
                       (const unsigned*)offs, (const unsigned*)coff, A, (long long)ldA, (double*)ctx->b_D.p, tab_max,
                       SparseOut{}, PNL_WL_BINS-1, nmin | (sym ? 1 << 16 : 0), ClusterTiles{});
    HIPCHK(ctx, hipGetLastError());
    return PNL_OK;
}

// work list for the orders that are integrated one pair per wave: `regions` regions of equal capacity, sized generously;
// an overflow is detected (check_overflow)
int ensure_worklist(pnl_context *ctx, double pairs, int regions) {
    const double frac = pnl_tune("PNL_WL_FRAC") ? atof(pnl_tune("PNL_WL_FRAC")) : 0.05;
    const double floor_entries = pnl_tune("PNL_WL_FRAC") ? 64. : (double)(1 << 20);
    const size_t each = (size_t)std::min<double>(std::max<double>(pairs*frac, floor_entries), 400e6);
    int rc;
    if ((rc = ensure(ctx, ctx->b_wl, each*(size_t)regions*sizeof(int4)))) return rc;
    ctx->wl_cap = (unsigned)std::min<size_t>(ctx->b_wl.bytes/sizeof(int4), 0xffffffffu);
    ctx->wl_cap_each = (unsigned)each;
    if ((rc = ensure(ctx, ctx->b_tilectr, sizeof(unsigned)))) return rc;
    return PNL_OK;
}

template <int DIM, int DPE, int TILE, int KT>
int launch_tiles(pnl_context *ctx, int wl_slot, double *A, int64_t ldA, int cell_begin, int cell_end, const SlotOut &SO = SlotOut{}) {
    using S = TileSmem<DIM, DPE, TILE, KT == 0>;
    HIPCHK(ctx, hipEventRecord(ctx->ev[7], ctx->stream));
    ctx->pure_launched = false;
    if (TILE == 64 && DPE <= 3) {
        int rc;
        // order-2 uniform tiles: 2D P1 the pipelined k_tile_uniform<3, 3> (with three or four workgroups per CU it beats k_tile_pure:
        // 41.3 against 45.5 ms at 98,304 cells, s = 1/2; PNL_PURE_V1=1 keeps k_tile_pure), 1D k_tile_pure
        if (DIM == 2 && DPE == 3 && !pnl_tune("PNL_PURE_V1") && ctx->uni_off[2] >= 0 && ctx->uni_np[2] == 3)
            rc = pnl2_launch_uniform(ctx, KT, tile_problem(ctx), (const int2*)ctx->b_tiles.p+ctx->tile_off+ctx->n_mixed, nullptr, ctx->n_pure, 2,
                                     A, ldA, (double*)(ctx->have_tile_order ? ctx->b_Dt.p : ctx->b_D.p), SO);
        else rc = launch_pure<DIM, (DPE <= 3 ? DPE : 3), KT>(ctx, A, ldA, SO);
        if (rc) return rc;
        ctx->pure_launched = ctx->n_pure > 0;
        if (DIM == 2 && DPE == 3) {
            // tiles whose pairs are all of order 3 / all of order 4 (6-point rules): pnl_tile2.h
            int off = ctx->tile_off+ctx->n_mixed+ctx->n_pure;
            for (int q = 3; q <= 4; q++) {
                const int n = ctx->cls_n_uni[q-2][ctx->cur];
                if ((rc = pnl2_launch_uniform(ctx, KT, tile_problem(ctx), (const int2*)ctx->b_tiles.p+off, nullptr, n, q, A, ldA,
                                              (double*)(ctx->have_tile_order ? ctx->b_Dt.p : ctx->b_D.p), SO))) return rc;
                ctx->pure_launched = ctx->pure_launched || n > 0;
                off += n;
            }
        }
        HIPCHK(ctx, hipEventRecord(ctx->ev[7], ctx->stream));
    }
    const int ntiles = ctx->n_mixed;
    const int acc_stride = acc_stride_of(ctx->nU, S::fixed_bytes);
    const size_t lds = S::fixed_bytes+sizeof(double)*(size_t)(ctx->nU+1)*acc_stride;
    if (pnl_tune("PNL_VERBOSE")) {
        int nblk = -1;
        (void)hipOccupancyMaxActiveBlocksPerMultiprocessor(&nblk, (const void*)k_tile_distant<DIM, DPE, TILE, KT, false>, tile_threads(DPE, KT), lds);
        fprintf(stderr, "[pnl] tiles=%d nU=%d lds=%zu bytes, occupancy API: %d blocks/CU\n", ntiles, ctx->nU, lds, nblk);
    }
    if (lds > 160*1024)
        return fail(ctx, PNL_ERR_UNSUPPORTED, "a block of %d cells touches %d DoFs: LDS sub-block of %zu bytes exceeds 160 KiB "
                    "(cells must be numbered with spatial locality)", TILE, ctx->nU, lds);
    // one region, reused by every class / orientation pass; the fill counter of pass p stays in slot p (check_overflow)
    unsigned *wlc = (unsigned*)ctx->b_wlcount.p+wl_slot;
    HIPCHK(ctx, hipMemsetAsync(ctx->b_tilectr.p, 0, sizeof(unsigned), ctx->stream));
    auto kfun = k_tile_distant<DIM, DPE, TILE, KT, false>;
    HIPCHK(ctx, hipFuncSetAttribute((const void*)kfun, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    int per_cu = 2;
    (void)hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, (const void*)kfun, tile_threads(DPE, KT), lds);
    const int grid_mult = pnl_tune("PNL_GRID_MULT") ? atoi(pnl_tune("PNL_GRID_MULT")) : 1;
    const int grid = pnl_grid_cap(std::min(ntiles, 256*std::max(per_cu, 1)*std::max(grid_mult, 1)));
    SlotOut SOk = SO;
    SOk.nU = ctx->nU;                                       // rows of the LDS sub-block (also without the block-slot storage)
    kt_begin(ctx, PNL_K_TILE_GENERAL);
    if (grid > 0)
        hipLaunchKernelGGL(kfun, dim3(grid), dim3(tile_threads(DPE, KT)), lds, ctx->stream, tile_problem(ctx), (const int2*)ctx->b_tiles.p+ctx->tile_off, A,
                           (long long)ldA, (double*)(ctx->have_tile_order ? ctx->b_Dt.p : ctx->b_D.p), cell_begin, cell_end, acc_stride, (int4*)ctx->b_wl.p,
                           wlc, ctx->wl_cap_each, ctx->ablate | (ctx->symflush ? 256 : 0), ntiles, ClusterTiles{},
                           (unsigned*)ctx->b_tilectr.p, SOk);
    kt_end(ctx, PNL_K_TILE_GENERAL);
    HIPCHK(ctx, hipGetLastError());
    HIPCHK(ctx, hipEventRecord(ctx->ev[6], ctx->stream));
    // block-slot storage: A = A' + A'^T is formed now (it overwrites A); the work-list kernels then write both images
    if (SO.A2) {
        int rc = pnl2_fold_mirror(ctx, SO, A, ldA);
        if (rc) return rc;
        HIPCHK(ctx, hipEventRecord(ctx->ev_fold, ctx->stream));
        ctx->fold_event_set = true;
    }
    return run_worklist<DIM, DPE, KT>(ctx, (const int4*)ctx->b_wl.p, wlc, ctx->wl_cap_each, A, ldA, ctx->symflush || SO.A2 != nullptr);
}

template <int DIM, int DPE, int SLOT, int KT>
int launch_singular_slot(pnl_context *ctx, int np, double *A, int64_t ldA, int cell_begin, int cell_end) {
    constexpr int NV = DIM+1;
    const int2 *pairs = (const int2*)(ctx->orient ? ctx->C().b_spairs1[SLOT].p : ctx->C().b_spairs[SLOT].p);
    const int M = ctx->P.sM[SLOT], rows = ctx->P.sRows[SLOT];
    const size_t lds = sizeof(double)*(size_t)(2*NV+1+rows)*M;
    const int waves_per_block = PNL_SING_THREADS/64;
    if (lds <= 150*1024) {
        auto kfun = k_singular_pairs<DIM, DPE, SLOT, KT, true, false>;
        HIPCHK(ctx, hipFuncSetAttribute((const void*)kfun, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        const int per_cu = std::max(1, (int)((160*1024)/std::max<size_t>(lds, 1)));
        const int grid = std::min((np+waves_per_block-1)/waves_per_block, 256*std::min(per_cu, 4));
        hipLaunchKernelGGL(kfun, dim3(grid), dim3(PNL_SING_THREADS), lds, ctx->stream, ctx->P, pairs, np, A, (long long)ldA,
                           cell_begin, cell_end, SparseOut{}, (const int4*)nullptr, (const unsigned*)nullptr, ClusterTiles{});
    } else {
        const int grid = std::min((np+waves_per_block-1)/waves_per_block, 256*4);
        hipLaunchKernelGGL((k_singular_pairs<DIM, DPE, SLOT, KT, false, false>), dim3(grid), dim3(PNL_SING_THREADS), 0, ctx->stream,
                           ctx->P, pairs, np, A, (long long)ldA, cell_begin, cell_end, SparseOut{}, (const int4*)nullptr,
                           (const unsigned*)nullptr, ClusterTiles{});
    }
    HIPCHK(ctx, hipGetLastError());
    return PNL_OK;
}

template <int DIM, int DPE, int KT>
int launch_singular(pnl_context *ctx, double *A, int64_t ldA, int cell_begin, int cell_end) {
    for (int s = 0; s < DIM+1; s++) {
        const int np = ctx->orient ? ctx->C().n_spairs1[s] : ctx->C().n_spairs[s];
        if (!np) continue;
        if (!ctx->C().have_sing[0][s]) return fail(ctx, PNL_ERR_STATE, "singular rule for %d common vertices not uploaded", s+1);
        int rc;
        if (s == 0) rc = launch_singular_slot<DIM, DPE, 0, KT>(ctx, np, A, ldA, cell_begin, cell_end);
        else if (s == 1) rc = launch_singular_slot<DIM, DPE, 1, KT>(ctx, np, A, ldA, cell_begin, cell_end);
        else rc = launch_singular_slot<DIM, DPE, (DIM == 2 ? 2 : 1), KT>(ctx, np, A, ldA, cell_begin, cell_end);
        if (rc) return rc;
    }
    return PNL_OK;
}

// what: 1 distant pairs, 2 touching pairs, 3 both; bkcls / bfcls: class tables for the one-launch distant pass of a variable order
template <int DIM, int DPE, int KT>
int launch_boundary(pnl_context *ctx, int cell_begin, int cell_end, int what = 3, const DevKernel *bkcls = nullptr,
                    const DevFormula *bfcls = nullptr, bool all_fast = false) {
    if (ctx->nb == 0 || cell_end <= cell_begin) return PNL_OK;
    const int ncell = cell_end-cell_begin;
    const int gx = (ncell+PNL_NTHREADS-1)/PNL_NTHREADS;
    // small facet chunks: many waves in flight hide the latency of the per-facet dependent chain
    // a rank's share of the cells may be small: shrink the facet chunks until the grid has a few thousand workgroups
    // (measured at 98,304 cells x 768 facets: 16 per chunk 5.4 ms, 32 per chunk 4.5 ms, 64 the same; 8: 6.9 ms, 1: 33.6 ms)
    int per = pnl_tune("PNL_BND_PER") ? atoi(pnl_tune("PNL_BND_PER")) : 32;
    while (per > 1 && (long long)gx*((ctx->nb+per-1)/per) < 4096) per >>= 1;
    const int chunks = (ctx->nb+per-1)/per;
    if (what & 1) {
        // pairs with more than `defer` point pairs are integrated one per wave (k_boundary_items) instead of by one lane; the
        // list holds at least 64 items per facet -- pairs that do not fit are integrated in place
        const int defer = pnl_tune("PNL_BND_DEFER") ? atoi(pnl_tune("PNL_BND_DEFER")) : 48;
        const bool use_list = defer > 0;
        const unsigned cap = (unsigned)std::min<long long>(std::max<long long>(65536, 64ll*ctx->nb), 1ll << 24);
        int *dcells = nullptr, *dfacets = nullptr, *dcls = nullptr;
        unsigned *dslots = nullptr, *dcount = nullptr;
        if (use_list) {
            int rc;
            if ((rc = ensure(ctx, ctx->b_bdefer, sizeof(int)*(size_t)cap*(3+DIM)+sizeof(unsigned)))) return rc;
            dcells = (int*)ctx->b_bdefer.p; dfacets = dcells+cap; dslots = (unsigned*)(dfacets+(size_t)cap*DIM); dcls = (int*)(dslots+cap);
            dcount = (unsigned*)(dcls+cap);
            HIPCHK(ctx, hipMemsetAsync(dcount, 0, sizeof(unsigned), ctx->stream));
        }
        const bool fast = bkcls ? all_fast : (ctx->P.bkn.fast != 0);
        bool tiled = false;
        if constexpr (DIM == 2) {
            // tiled kernel (pnl_bndtile.h): 256 cells per workgroup, facets in LDS chunks of 64, rules up to order qi in LDS
            const int qi = std::min(PNL_BT_QI, ctx->qmax);
            if (!pnl_tune("PNL_BND_OLD") && qi >= 2 && (int)ctx->rule_off.size() > qi+1) {
                const int npts = ctx->rule_off[qi+1]-ctx->rule_off[2], nfp = ctx->frule_off[qi+1]-ctx->frule_off[2];
                const int ncls = bkcls ? (int)ctx->cls.size() : 0;
                const BndTileLds LY = bnd_tile_layout(npts, nfp, 4+DPE, ncls);
                const size_t lds = sizeof(double)*(size_t)LY.total;
                if (lds <= 64*1024) {
                    // facet blocks: enough workgroups to fill the chip (a rank's share of the cells may be small), whole chunks of 64
                    const int nfb = (ctx->nb+PNL_BT_FB-1)/PNL_BT_FB;
                    const int want = pnl_tune("PNL_BT_WG") ? atoi(pnl_tune("PNL_BT_WG")) : 2048;
                    const int gy = std::max(1, std::min(nfb, (want+gx-1)/gx));
                    const int per_block = ((nfb+gy-1)/gy)*PNL_BT_FB;
                    const int gyy = (ctx->nb+per_block-1)/per_block;
                    auto launch = [&](auto kfun) {
                        (void)hipFuncSetAttribute((const void*)kfun, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
                        hipLaunchKernelGGL(kfun, dim3(gx, gyy), dim3(PNL_NTHREADS), lds, ctx->stream, ctx->P, (double*)ctx->b_D.p, cell_begin,
                                           cell_end, per_block, qi, bkcls, bfcls, ncls, defer, dcells, dfacets, dslots, dcount, cap, dcls);
                    };
                    if (fast) launch(k_boundary_tile<DIM, DPE, 1>); else launch(k_boundary_tile<DIM, DPE, 0>);
                    tiled = true;
                }
            }
        }
        if (tiled) {}
        else if (fast)
            hipLaunchKernelGGL((k_boundary_distant<DIM, DPE, 1>), dim3(gx, chunks), dim3(PNL_NTHREADS), 0, ctx->stream, ctx->P,
                               (double*)ctx->b_D.p, cell_begin, cell_end, per, bkcls, bfcls, defer, dcells, dfacets, dslots, dcount, cap, dcls);
        else
            hipLaunchKernelGGL((k_boundary_distant<DIM, DPE, 0>), dim3(gx, chunks), dim3(PNL_NTHREADS), 0, ctx->stream, ctx->P,
                               (double*)ctx->b_D.p, cell_begin, cell_end, per, bkcls, bfcls, defer, dcells, dfacets, dslots, dcount, cap, dcls);
        if (use_list) {
            const double *verts = (const double*)ctx->b_vertices.p;
            if (fast)
                hipLaunchKernelGGL((k_boundary_items<DIM, DPE, 1>), dim3(256*4), dim3(PNL_NTHREADS), 0, ctx->stream, ctx->P, verts,
                                   (const int*)dcells, (const int*)dfacets, (const unsigned*)dslots, (int)cap, 1., SparseOut{},
                                   (double*)ctx->b_D.p, (const unsigned*)dcount, bkcls ? (const int*)dcls : nullptr, bkcls, bfcls);
            else
                hipLaunchKernelGGL((k_boundary_items<DIM, DPE, 0>), dim3(256*4), dim3(PNL_NTHREADS), 0, ctx->stream, ctx->P, verts,
                                   (const int*)dcells, (const int*)dfacets, (const unsigned*)dslots, (int)cap, 1., SparseOut{},
                                   (double*)ctx->b_D.p, (const unsigned*)dcount, bkcls ? (const int*)dcls : nullptr, bkcls, bfcls);
        }
        HIPCHK(ctx, hipGetLastError());
    }
    if (what & 2)
    for (int s = 0; s < DIM; s++) {
        const int np = ctx->C().n_bpairs[s];
        if (!np) continue;
        if (!ctx->C().have_sing[1][s]) return fail(ctx, PNL_ERR_STATE, "boundary singular rule for %d common vertices not uploaded", s+1);
        const int grid = (np+3)/4;
        const int2 *pairs = (const int2*)ctx->C().b_bpairs[s].p;
        const bool fast = ctx->P.bkn.fast;
        if (s == 0 && fast) hipLaunchKernelGGL((k_boundary_singular<DIM, DPE, 0, 1>), dim3(grid), dim3(PNL_NTHREADS), 0, ctx->stream, ctx->P, pairs, np, (double*)ctx->b_D.p, cell_begin, cell_end);
        else if (s == 0) hipLaunchKernelGGL((k_boundary_singular<DIM, DPE, 0, 0>), dim3(grid), dim3(PNL_NTHREADS), 0, ctx->stream, ctx->P, pairs, np, (double*)ctx->b_D.p, cell_begin, cell_end);
        else if (fast) hipLaunchKernelGGL((k_boundary_singular<DIM, DPE, (DIM == 2 ? 1 : 0), 1>), dim3(grid), dim3(PNL_NTHREADS), 0, ctx->stream, ctx->P, pairs, np, (double*)ctx->b_D.p, cell_begin, cell_end);
        else hipLaunchKernelGGL((k_boundary_singular<DIM, DPE, (DIM == 2 ? 1 : 0), 0>), dim3(grid), dim3(PNL_NTHREADS), 0, ctx->stream, ctx->P, pairs, np, (double*)ctx->b_D.p, cell_begin, cell_end);
        HIPCHK(ctx, hipGetLastError());
    }
    return PNL_OK;
}

// block-slot storage of the one-sided operator (pnl_tile2.h): allocated when the device has room for it next to the caller's matrix
bool slot_storage_ready(pnl_context *ctx) {
    if (ctx->nonsym || pnl_tune("PNL_NO_SLOT")) return false;
    size_t free_b = 0, total_b = 0;
    const size_t need = sizeof(double)*(size_t)ctx->slot_total;
    if (ctx->b_slotA.bytes >= need) return true;
    if (hipMemGetInfo(&free_b, &total_b) != hipSuccess || free_b <= need+(size_t(2) << 30)) return false;
    return ensure(ctx, ctx->b_slotA, need) == PNL_OK;
}

// dim 2, dpe 6: ONE launch of the uniform-tile kernels per order and ONE of the general P2 tile kernel over the tiles of all
// order classes (every tile entry carries its class: kernel parameters and order formula come from per-class tables), then the
// sorted work-list evaluation per class on that class's region of the work list
template <int DIM, int DPE>
int launch_tiles_single(pnl_context *ctx, double *A, int64_t ldA, int cell_begin, int cell_end, bool slot_ok) {
    int rc;
    const int ncls = (int)ctx->cls.size();
    const bool var = ctx->nlab > 0;
    ctx->kcls_host.resize(ncls); ctx->fcls_host.resize(ncls);
    bool all_fast = true, all_half = true;
    for (int k = 0; k < ncls; k++) {
        ctx->cur = k; ctx->orient = 0;
        refresh_tables(ctx);
        ctx->kcls_host[k] = ctx->P.k; ctx->fcls_host[k] = ctx->P.qo;
        all_fast = all_fast && ctx->P.k.fast;
        all_half = all_half && ctx->P.k.fast && ctx->P.k.qm == 6;
    }
    const int kt = (all_half && !pnl_tune("PNL_NO_KT2")) ? 2 : (all_fast ? 1 : 0);
    if (kt == 0)
        // three layers with s = 0.3 .. 0.7: the class s = 1/2 (half of the pairs) is a "fast" kernel without tables of its own
        for (int k = 0; k < ncls; k++)
            if (!ctx->kcls_host[k].ptab) ctx->kcls_host[k].ptab = pow_table(ctx, ctx->kcls_host[k], true);
    if ((rc = ensure(ctx, ctx->b_kcls, sizeof(DevKernel)*ncls))) return rc;
    if ((rc = ensure(ctx, ctx->b_fcls, sizeof(DevFormula)*ncls))) return rc;
    HIPCHK(ctx, hipMemcpyAsync(ctx->b_kcls.p, ctx->kcls_host.data(), sizeof(DevKernel)*ncls, hipMemcpyHostToDevice, ctx->stream));
    HIPCHK(ctx, hipMemcpyAsync(ctx->b_fcls.p, ctx->fcls_host.data(), sizeof(DevFormula)*ncls, hipMemcpyHostToDevice, ctx->stream));
    ctx->cur = 0;
    refresh_tables(ctx);
    const int2 *tiles = (const int2*)ctx->b_tiles.p;
    const int *tcls = var ? (const int*)ctx->b_tilecls.p : nullptr;
    HIPCHK(ctx, hipMemsetAsync(ctx->b_tilectr.p, 0, sizeof(unsigned), ctx->stream));
    // block-slot storage when the whole upper block triangle is assembled by this call and mirrored afterwards
    SlotOut SO{};
    ctx->slot_used = false;
    if (slot_ok && slot_storage_ready(ctx)) {
        SO = slot_out(ctx);
        ctx->slot_used = true;
        if (var && (rc = pnl2_zero_slot_tiles(ctx, SO))) return rc;
    }
    HIPCHK(ctx, hipEventRecord(ctx->ev[7], ctx->stream));
    ctx->pure_launched = false;
    for (int q = 2; q <= 4; q++) {
        const int n = ctx->sl_n[q-1];
        if ((rc = pnl2_launch_uniform(ctx, kt, ctx->P, tiles+ctx->sl_off[q-1], tcls ? tcls+ctx->sl_off[q-1] : nullptr, n, q, A, ldA,
                                      (double*)ctx->b_D.p, SO))) return rc;
        ctx->pure_launched = ctx->pure_launched || n > 0;
    }
    if (ctx->pure_launched) HIPCHK(ctx, hipEventRecord(ctx->ev[7], ctx->stream));
    if ((rc = pnl2_launch_p2(ctx, kt, tiles+ctx->sl_off[0], tcls ? tcls+ctx->sl_off[0] : nullptr, ctx->sl_n[0], A, ldA, cell_begin, cell_end,
                             ctx->wl_cap_each, SO))) return rc;
    HIPCHK(ctx, hipEventRecord(ctx->ev[6], ctx->stream));
    // block-slot storage: A = A' + A'^T is formed now (it overwrites A); the work-list kernels then write both images
    if (ctx->slot_used) {
        if ((rc = pnl2_fold_mirror(ctx, SO, A, ldA))) return rc;
        HIPCHK(ctx, hipEventRecord(ctx->ev_fold, ctx->stream));
        ctx->fold_event_set = true;
    }
    const bool sym = ctx->symflush || ctx->slot_used;
    {
        ClassFork fork(ctx, ncls);
        if (ncls > pnl_context::NAUX && (int)ctx->cls_n_mixed.size() == ncls) fork.plan(ctx->cls_n_mixed);     // the work lists come from the mixed tiles
        for (int k = 0; k < ncls; k++) {
            ctx->cur = k;
            refresh_tables(ctx);
            fork.use(k);
            const int4 *wl = (const int4*)ctx->b_wl.p+(size_t)k*ctx->wl_cap_each;
            const unsigned *wlc = (const unsigned*)ctx->b_wlcount.p+k;
            rc = ctx->P.k.fast ? run_worklist<DIM, DPE, 1>(ctx, wl, wlc, ctx->wl_cap_each, A, ldA, sym, k, ncls)
                               : run_worklist<DIM, DPE, 0>(ctx, wl, wlc, ctx->wl_cap_each, A, ldA, sym, k, ncls);
            if (rc) { ctx->cur = 0; return rc; }
        }
    }
    ctx->cur = 0;
    return PNL_OK;
}

// the whole upper block triangle is assembled by this call and mirrored afterwards
bool slot_eligible(const pnl_context *ctx, int cell_begin, int cell_end, int flags) {
    // P2: every tile list (several order classes: multi-visit tiles accumulate); P1: one class, every tile visited by one kernel
    const bool elem = ctx->dpe == 6 || (ctx->dpe == 3 && ctx->cls.size() == 1 && !ctx->nonsym && ctx->use_pure && !pnl_tune("PNL_NO_SLOT_P1"));
    return ctx->dim == 2 && elem && ctx->slot_full_list && ctx->slab_rows == 0 && cell_begin == 0 && cell_end == ctx->nc &&
           !(flags & (PNL_FLAG_NO_MIRROR | PNL_FLAG_SYMMETRIC_FLUSH));
}

#ifndef PNL_BND_MODE_DEFAULT
#define PNL_BND_MODE_DEFAULT 3
#endif
template <int DIM, int DPE, int TILE>
int assemble_impl(pnl_context *ctx, double *A, int64_t ldA, int zero_exterior, int ntiles, int cell_begin, int cell_end, int flags) {
    int rc;
    const int ncls = (int)ctx->cls.size();
    ctx->symflush = (flags & PNL_FLAG_SYMMETRIC_FLUSH) != 0;
    for (bool &b : ctx->kev_set) b = false;
    HIPCHK(ctx, hipMemsetAsync(ctx->b_counters.p, 0, sizeof(unsigned long long)*PNL_NCOUNTERS, ctx->stream));
    HIPCHK(ctx, hipMemsetAsync(ctx->b_D.p, 0, sizeof(double)*(size_t)ctx->ncp*(DPE*(DPE+1)/2), ctx->stream));
    if (ctx->have_tile_order) HIPCHK(ctx, hipMemsetAsync(ctx->b_Dt.p, 0, sizeof(double)*(size_t)ctx->ncp*(DPE*(DPE+1)/2), ctx->stream));
    HIPCHK(ctx, hipEventRecord(ctx->ev[0], ctx->stream));
    ctx->tiles_launched = ntiles > 0;
    ctx->fold_event_set = false;
    // variable order, zero exterior: the distant (cell, facet) pairs of ALL classes run in one launch with per-class kernel /
    // order-formula tables; the tables go up now, ahead of the tile kernels, so that the launch needs nothing from the
    // caller's stream later than the fold
    const bool bnd_one_pass = zero_exterior && ncls > 1 && ctx->nlab > 0 && !pnl_tune("PNL_BND_PER_CLASS");
    bool bnd_all_fast = true;
    if (bnd_one_pass) {
        std::vector<DevKernel> &bk = ctx->bkcls_host;
        std::vector<DevFormula> &bf = ctx->bfcls_host;
        bk.resize(ncls); bf.resize(ncls);
        for (int k = 0; k < ncls; k++) {
            ctx->cur = k;
            refresh_tables(ctx);
            bk[k] = ctx->P.bkn; bf[k] = ctx->P.bqo;
            bnd_all_fast = bnd_all_fast && ctx->P.bkn.fast;
        }
        ctx->cur = 0;
        if ((rc = ensure(ctx, ctx->b_bkcls, sizeof(DevKernel)*ncls))) return rc;
        if ((rc = ensure(ctx, ctx->b_bfcls, sizeof(DevFormula)*ncls))) return rc;
        HIPCHK(ctx, hipMemcpyAsync(ctx->b_bkcls.p, bk.data(), sizeof(DevKernel)*ncls, hipMemcpyHostToDevice, ctx->stream));
        HIPCHK(ctx, hipMemcpyAsync(ctx->b_bfcls.p, bf.data(), sizeof(DevFormula)*ncls, hipMemcpyHostToDevice, ctx->stream));
    }
    // Omega x Omega^c.  Variable order: the distant (cell, facet) pairs of ALL classes in one launch with per-class kernel /
    // order-formula tables, the touching pairs per class (their rules are per class).  chain: fork over the side streams from that
    // event (the order classes' streams after a fold); nullptr: everything on ctx->stream
    auto boundary_term = [&](hipEvent_t chain) -> int {
        int rcb;
        ClassFork fork(ctx, chain ? ncls : 1, chain);
        if (chain && ncls > pnl_context::NAUX && (int)ctx->cls_n_mixed.size() == ncls) fork.plan(ctx->cls_n_mixed);
        for (int k = 0; k < ncls; k++) {
            ctx->cur = k;
            refresh_tables(ctx);
            if (!ctx->have_boundary || !ctx->C().have_kernel[1] || !ctx->C().have_form[1]) {
                ctx->cur = 0;
                return fail(ctx, PNL_ERR_STATE, "zero_exterior needs boundary facets, boundary kernel and order formula");
            }
            if (chain) fork.use(k);
            if (bnd_one_pass && k == 0 &&
                (rcb = launch_boundary<DIM, DPE, 0>(ctx, cell_begin, cell_end, 1, (const DevKernel*)ctx->b_bkcls.p, (const DevFormula*)ctx->b_bfcls.p,
                                                    bnd_all_fast))) { ctx->cur = 0; return rcb; }
            if (chain) fork.use(k);
            if ((rcb = launch_boundary<DIM, DPE, 0>(ctx, cell_begin, cell_end, bnd_one_pass ? 2 : 3))) { ctx->cur = 0; return rcb; }
        }
        ctx->cur = 0;
        return PNL_OK;
    };
    // The boundary term only adds to the per-cell diagonal blocks (b_D, scattered into A at the very end) and never touches A, so
    // its stream needs nothing but the zero fill of that buffer and the class tables.  PNL_BND_MODE 3 (default): side stream 1 behind
    // the LAST TILE KERNEL, i.e. next to the fold pass -- the fold is bound by HBM (VALU issue utilisation 0.4), the boundary kernels by
    // latency and arithmetic, and nothing else can run there (everything else adds into A, which the fold overwrites): P2 39.9 -> 39.3
    // ms, C5 52.2 -> 51.6, headline 103.3 -> 102.9 in bench.py's queued loop.  0: behind the fold (one class: side stream 1, several:
    // forked over the class streams), next to the work-list kernels.  1: behind the zero fill, submitted after the tile and
    // work-list kernels; 2: submitted first.  Measured: as soon as the boundary kernels are READY while the tile kernels still have
    // workgroups to place, their workgroups take CU slots in front of them and the tile phase loses more than the boundary term
    // costs (mode 2, P1 s = 0.4: 138 -> 173 ms).  Mode 1 wins 0.5-2 ms when the host waits for every assembly (the launches of the
    // boundary kernels arrive late) and loses 12 ms of 103 in a loop of assemblies without a host synchronisation in between
    // (everything is queued ahead, so it behaves like mode 2); a lowest-priority stream recovers half of that.
    const int bnd_mode = !zero_exterior ? -1 : pnl_tune("PNL_BND_MODE") ? atoi(pnl_tune("PNL_BND_MODE")) : PNL_BND_MODE_DEFAULT;
    bool bnd_side = false;
    auto boundary_side = [&](hipEvent_t after) -> int {
        hipStream_t const caller = ctx->stream;
        hipStream_t const side = ctx->aux[1];
        HIPCHK(ctx, hipStreamWaitEvent(side, after, 0));
        ctx->stream = side;
        const int rcb = boundary_term(nullptr);
        ctx->stream = caller;
        if (rcb) return rcb;
        HIPCHK(ctx, hipEventRecord(ctx->ev_join[1], side));
        bnd_side = true;
        return PNL_OK;
    };
    if (bnd_mode > 0) HIPCHK(ctx, hipEventRecord(ctx->ev_bnd, ctx->stream));
    if (bnd_mode == 2) {
        if ((rc = boundary_side(ctx->ev_bnd))) return rc;
        refresh_tables(ctx);
    }
    // a piecewise-constant variable order is assembled class by class: every pass sees the kernel, order formula and
    // singular rules of one order value and skips the pairs of the other classes in its classification
    const int norient = ctx->nonsym ? 2 : 1;
    if (ntiles > 0) {
        const bool single = DIM == 2 && DPE == 6;
        if (ncls*norient > PNL_WL_SLOTS) return fail(ctx, PNL_ERR_UNSUPPORTED, "more than %d order classes x orientations", PNL_WL_SLOTS);
        if ((rc = ensure(ctx, ctx->b_wlcount, sizeof(unsigned)*PNL_WL_SLOTS))) return rc;
        HIPCHK(ctx, hipMemsetAsync(ctx->b_wlcount.p, 0, sizeof(unsigned)*PNL_WL_SLOTS, ctx->stream));
        if ((rc = ensure_worklist(ctx, (double)ntiles*TILE*TILE, single ? ncls : 1))) return rc;
        ctx->wl_slots = single ? ncls : ncls*norient;
        if constexpr (DIM == 2 && DPE == 6) {
            // the block-slot storage needs every tile of the upper block triangle written by this call, then the fold + mirror pass
            const bool slot_ok = slot_eligible(ctx, cell_begin, cell_end, flags);
            if ((rc = launch_tiles_single<DIM, DPE>(ctx, A, ldA, ctx->tile_cell_filter ? cell_begin : 0, ctx->tile_cell_filter ? cell_end : ctx->nc, slot_ok)))
                { ctx->cur = 0; ctx->orient = 0; return rc; }
        } else {
        // 2D P1, one order class: block-slot storage like the P2 path (plain stores instead of 47 GB of fp64 atomics at 48,769
        // DoFs, fold + mirror instead of zero fill + mirror)
        SlotOut SO{};
        ctx->slot_used = false;
        if (slot_eligible(ctx, cell_begin, cell_end, flags) && slot_storage_ready(ctx)) { SO = slot_out(ctx); ctx->slot_used = true; }
        for (int ko = 0; ko < ncls*norient; ko++) {
            const int k = ko/norient;
            ctx->cur = k; ctx->orient = ko%norient;
            refresh_tables(ctx);
            ctx->tile_off = ctx->cls_tile_off[k]; ctx->n_mixed = ctx->cls_n_mixed[k]; ctx->n_pure = ctx->cls_n_pure[k];
            if (ctx->n_mixed+ctx->n_pure+ctx->cls_n_uni[1][k]+ctx->cls_n_uni[2][k] == 0) continue;
            const int tb0 = ctx->tile_cell_filter ? cell_begin : 0, tb1 = ctx->tile_cell_filter ? cell_end : ctx->nc;
            // s = 1/2 in 2D (exponent -6/4) has its own instantiation: branch-free evaluations
            if (ctx->P.k.fast && ctx->P.k.qm == 6 && DPE == 3 && !pnl_tune("PNL_NO_KT2")) rc = launch_tiles<DIM, DPE, TILE, (DPE == 3 ? 2 : 1)>(ctx, ko, A, ldA, tb0, tb1, SO);
            else rc = ctx->P.k.fast ? launch_tiles<DIM, DPE, TILE, 1>(ctx, ko, A, ldA, tb0, tb1, SO)
                                    : launch_tiles<DIM, DPE, TILE, 0>(ctx, ko, A, ldA, tb0, tb1, SO);
            if (rc) { ctx->cur = 0; ctx->orient = 0; return rc; }
        }
        }
    }
    ctx->orient = 0;
    HIPCHK(ctx, hipEventRecord(ctx->ev[1], ctx->stream));
    // mirror the cross part before the symmetric contributions are added on both sides
    if (ntiles > 0 && ctx->slot_used) {
        // folded and mirrored inside launch_tiles_single
    } else if (!(flags & (PNL_FLAG_NO_MIRROR | PNL_FLAG_SYMMETRIC_FLUSH))) {
        const long long nb = (ctx->N+31)/32;
        hipLaunchKernelGGL(k_mirror, dim3((unsigned)(nb*(nb+1)/2)), dim3(PNL_NTHREADS), 0, ctx->stream, A, (long long)ldA, ctx->N);
        HIPCHK(ctx, hipGetLastError());
    }
    HIPCHK(ctx, hipEventRecord(ctx->ev[2], ctx->stream));
    // one order class behind a fold pass: the touching pairs (side stream 0) and the boundary term (side stream 1) start at the
    // fold, next to the work-list kernels on the caller's stream -- everything after the fold only adds with atomics.  The
    // phase timers then show what is left of them after the work list ("singular"), the boundary phase reads 0
    hipStream_t const main_stream = ctx->stream;
    // without a fold pass (row slabs of a rank, 1D, P0) the same holds from here on: the touching pairs and the boundary term add
    // with atomics and run side by side
    bool fold_like = ctx->fold_event_set;
    if (!fold_like && ncls*norient == 1 && zero_exterior && !pnl_tune("PNL_NO_OVERLAP")) {
        HIPCHK(ctx, hipEventRecord(ctx->ev_fold, ctx->stream));
        fold_like = true;
    }
    const bool overlap = fold_like && ncls*norient == 1 && !pnl_tune("PNL_NO_OVERLAP");
    struct StreamGuard { pnl_context *c; hipStream_t s; ~StreamGuard() { c->stream = s; } } stream_guard{ctx, main_stream};
    if (overlap) { HIPCHK(ctx, hipStreamWaitEvent(ctx->aux[0], ctx->ev_fold, 0)); ctx->stream = ctx->aux[0]; }
    // several classes behind a fold pass: the side streams of the touching pairs and of the boundary term start at the fold
    // too, class k follows the work list of class k on its stream, nothing waits for the other classes
    hipEvent_t const chain = (ctx->fold_event_set && !overlap && !pnl_tune("PNL_NO_OVERLAP")) ? ctx->ev_fold : nullptr;
    {
        ClassFork fork(ctx, ncls*norient, chain);
        if (norient == 1 && ncls > pnl_context::NAUX && (int)ctx->cls_n_mixed.size() == ncls) fork.plan(ctx->cls_n_mixed);    // as the work lists
        for (int ko = 0; ko < ncls*norient; ko++) {
            ctx->cur = ko/norient; ctx->orient = ko%norient;
            refresh_tables(ctx);
            fork.use(ko);
            rc = ctx->P.k.fast ? launch_singular<DIM, DPE, 1>(ctx, A, ldA, cell_begin, cell_end)
                               : launch_singular<DIM, DPE, 0>(ctx, A, ldA, cell_begin, cell_end);
            if (rc) { ctx->cur = 0; ctx->orient = 0; return rc; }
        }
    }
    ctx->orient = 0;
    if (overlap) {
        HIPCHK(ctx, hipEventRecord(ctx->ev_join[0], ctx->aux[0]));
        ctx->stream = main_stream;
    } else HIPCHK(ctx, hipEventRecord(ctx->ev[3], ctx->stream));
    if (bnd_mode == 3 && ctx->fold_event_set) rc = boundary_side(ctx->ev[6]);          // next to the fold pass (ev[6]: tile kernels done)
    else if (bnd_mode == 1 || ((bnd_mode == 0 || bnd_mode == 3) && overlap)) rc = boundary_side(bnd_mode == 1 ? ctx->ev_bnd : ctx->ev_fold);
    else if (bnd_mode == 0 || bnd_mode == 3) rc = boundary_term(chain);
    if (rc) return rc;
    ctx->cur = 0;
    if (overlap) {
        HIPCHK(ctx, hipStreamWaitEvent(main_stream, ctx->ev_join[0], 0));
        ctx->stream = main_stream;
        HIPCHK(ctx, hipEventRecord(ctx->ev[3], ctx->stream));
    }
    if (bnd_side) HIPCHK(ctx, hipStreamWaitEvent(main_stream, ctx->ev_join[1], 0));
    HIPCHK(ctx, hipEventRecord(ctx->ev[4], ctx->stream));
    if (ctx->slab_rows == 0) {
        const long long nt = (long long)ctx->nc*DPE*DPE;
        hipLaunchKernelGGL((k_scatter_diag<DPE>), dim3((unsigned)((nt+PNL_NTHREADS-1)/PNL_NTHREADS)), dim3(PNL_NTHREADS), 0,
                           ctx->stream, ctx->P, (const double*)ctx->b_D.p, A, (long long)ldA);
        if (ctx->have_tile_order)
            hipLaunchKernelGGL((k_scatter_diag<DPE>), dim3((unsigned)((nt+PNL_NTHREADS-1)/PNL_NTHREADS)), dim3(PNL_NTHREADS), 0,
                               ctx->stream, tile_problem(ctx), (const double*)ctx->b_Dt.p, A, (long long)ldA);
        HIPCHK(ctx, hipGetLastError());
    }
    HIPCHK(ctx, hipEventRecord(ctx->ev[5], ctx->stream));
    ctx->ev_valid = true;
    return PNL_OK;
}

// ---- masked pair assembly into CSR / SSS (assembleClusters) --------------------------------------------------------
template <int DIM, int DPE, int SLOT, int KT>
int launch_singular_sparse(pnl_context *ctx, const SparseOut &S, const int4 *sorted, const unsigned *offs) {
    constexpr int NV = DIM+1;
    if (!ctx->C().have_sing[0][SLOT]) return PNL_OK;     // checked against the histogram by the caller
    const int M = ctx->P.sM[SLOT], rows = ctx->P.sRows[SLOT];
    const size_t lds = sizeof(double)*(size_t)(2*NV+1+rows)*M;
    if (lds <= 150*1024) {
        auto kfun = k_singular_pairs<DIM, DPE, SLOT, KT, true, true>;
        HIPCHK(ctx, hipFuncSetAttribute((const void*)kfun, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        const int per_cu = std::max(1, (int)((160*1024)/std::max<size_t>(lds, 1)));
        hipLaunchKernelGGL(kfun, dim3(256*std::min(per_cu, 4)), dim3(PNL_SING_THREADS), lds, ctx->stream, ctx->P, (const int2*)nullptr, 0,
                           (double*)nullptr, 0ll, 0, 0, S, sorted, offs, ClusterTiles{});
    } else {
        hipLaunchKernelGGL((k_singular_pairs<DIM, DPE, SLOT, KT, false, true>), dim3(256*4), dim3(PNL_SING_THREADS), 0, ctx->stream,
                           ctx->P, (const int2*)nullptr, 0, (double*)nullptr, 0ll, 0, 0, S, sorted, offs, ClusterTiles{});
    }
    HIPCHK(ctx, hipGetLastError());
    return PNL_OK;
}

// classify == false: the work list b_mp_wl[0..np) has been filled on the device (k_fh_pairs)
template <int DIM, int DPE, int KT>
int pairs_masked_impl(pnl_context *ctx, int np, const SparseOut &S, bool classify = true, bool first = true, bool keepD = false) {
    int rc;
    if ((rc = ensure(ctx, ctx->b_mp_wl, (size_t)np*sizeof(int4)))) return rc;
    if ((rc = ensure(ctx, ctx->b_mp_sorted, (size_t)np*sizeof(int4)))) return rc;
    if ((rc = ensure(ctx, ctx->b_mp_aux, sizeof(unsigned)*(4*(PNL_WL_BINS+1)+1)))) return rc;
    unsigned *hist = (unsigned*)ctx->b_mp_aux.p, *offs = hist+(PNL_WL_BINS+1), *coff = offs+(PNL_WL_BINS+1), *cursor = coff+(PNL_WL_BINS+1),
             *count = cursor+(PNL_WL_BINS+1);
    int4 *wl = (int4*)ctx->b_mp_wl.p, *sorted = (int4*)ctx->b_mp_sorted.p;
    const unsigned unp = (unsigned)np;
    HIPCHK(ctx, hipMemsetAsync(hist, 0, sizeof(unsigned)*(PNL_WL_BINS+1), ctx->stream));
    HIPCHK(ctx, hipMemcpyAsync(count, &unp, sizeof(unsigned), hipMemcpyHostToDevice, ctx->stream));
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));          // unp lives on this stack frame
    if (first) HIPCHK(ctx, hipEventRecord(ctx->ev[0], ctx->stream));
    if (classify)
        hipLaunchKernelGGL((k_mp_classify<DIM, DPE>), dim3((np+PNL_NTHREADS-1)/PNL_NTHREADS), dim3(PNL_NTHREADS), 0, ctx->stream, ctx->P,
                           S.pairs, np, wl);
    hipLaunchKernelGGL(k_wl_hist, dim3(512), dim3(PNL_NTHREADS), 0, ctx->stream, (const int4*)wl, (const unsigned*)count, unp, hist);
    hipLaunchKernelGGL(k_wl_scan, dim3(1), dim3(64), 0, ctx->stream, (const unsigned*)hist, offs, coff, cursor);
    hipLaunchKernelGGL(k_wl_scatter, dim3(512), dim3(PNL_NTHREADS), 0, ctx->stream, (const int4*)wl, (const unsigned*)count, unp,
                       (const unsigned*)offs, cursor, sorted);
    hipLaunchKernelGGL(k_mp_stats, dim3(1), dim3(PNL_WL_BINS), 0, ctx->stream, ctx->P, (const unsigned*)hist);
    HIPCHK(ctx, hipGetLastError());
    HIPCHK(ctx, hipEventRecord(ctx->ev[6], ctx->stream));
    {
        const int st = 4+DPE;
        // LDS copy of the rule: see run_worklist (PNL_WL_MP_KB: A/B switch of the sparse path)
        const int wl_kb = pnl_tune("PNL_WL_MP_KB") ? std::max(4, atoi(pnl_tune("PNL_WL_MP_KB"))) : 60;
        const int tab_max = wl_tab_max<DPE>(wl_kb);
        const int wl_grid = 256*std::max(1, std::min(8, 150/(wl_kb+(KT == 0 ? 3 : 0))));
        const size_t lds = wl_sorted_lds<DPE>(tab_max);
        auto wfun = k_worklist_sorted<DIM, DPE, KT, true>;
        HIPCHK(ctx, hipFuncSetAttribute((const void*)wfun, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        const int nmin = ctx->wl_lane ? PNL_WL_LANE_MAXPTS+1 : 0;
        // no masks (getSparse): diagonal blocks through the per-cell buffer, one scatter per cell at the end
        constexpr int ND = DPE*(DPE+1)/2;
        double *Dbuf = S.masks ? nullptr : (double*)ctx->b_D.p;
        if (Dbuf && !keepD) HIPCHK(ctx, hipMemsetAsync(Dbuf, 0, sizeof(double)*(size_t)ctx->ncp*ND, ctx->stream));      // keepD: the tiles of a finite horizon have been there
        if (ctx->wl_lane)
            hipLaunchKernelGGL((k_worklist_lane<DIM, DPE, KT, true>), dim3(256*4), dim3(PNL_NTHREADS), wl_lane_lds(k_worklist_lane<DIM, DPE, KT, true>, DPE, KT), ctx->stream, ctx->P,
                               (const int4*)sorted, (const unsigned*)offs, (double*)nullptr, 0ll, Dbuf, S, 0, ClusterTiles{});
        hipLaunchKernelGGL(wfun, dim3(wl_grid), dim3(PNL_NTHREADS), lds, ctx->stream, ctx->P, (const int4*)sorted, (const unsigned*)offs,
                           (const unsigned*)coff, (double*)nullptr, 0ll, Dbuf, tab_max, S, PNL_MAXQ, nmin, ClusterTiles{});
        if (Dbuf) {
            const long long n = (long long)ctx->nc*ND;
            hipLaunchKernelGGL((k_scatter_diag_sparse<DPE>), dim3((unsigned)((n+PNL_NTHREADS-1)/PNL_NTHREADS)), dim3(PNL_NTHREADS), 0,
                               ctx->stream, ctx->P, (const double*)Dbuf, ctx->nc, S);
        }
        HIPCHK(ctx, hipGetLastError());
    }
    HIPCHK(ctx, hipEventRecord(ctx->ev[1], ctx->stream));
    HIPCHK(ctx, hipEventRecord(ctx->ev[2], ctx->stream));
    // touching pairs: bins 121 (common vertex), 122 (common edge / identical in 1D), 123 (identical in 2D)
    unsigned hh[PNL_WL_BINS+1];
    HIPCHK(ctx, hipMemcpyAsync(hh, hist, sizeof(hh), hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    for (int s = 0; s < DIM+1; s++)
        if (hh[121+s] && !ctx->C().have_sing[0][s]) return fail(ctx, PNL_ERR_STATE, "singular rule for %d common vertices not uploaded", s+1);
    if (hh[121] && (rc = launch_singular_sparse<DIM, DPE, 0, KT>(ctx, S, sorted, offs))) return rc;
    if (hh[122] && (rc = launch_singular_sparse<DIM, DPE, 1, KT>(ctx, S, sorted, offs))) return rc;
    if (DIM == 2 && hh[123] && (rc = launch_singular_sparse<DIM, DPE, (DIM == 2 ? 2 : 1), KT>(ctx, S, sorted, offs))) return rc;
    HIPCHK(ctx, hipEventRecord(ctx->ev[3], ctx->stream));
    HIPCHK(ctx, hipEventRecord(ctx->ev[4], ctx->stream));
    HIPCHK(ctx, hipEventRecord(ctx->ev[5], ctx->stream));
    ctx->ev_valid = true;
    ctx->tiles_launched = true;
    return PNL_OK;
}

template <int DIM, int DPE>
int boundary_masked_impl(pnl_context *ctx, int ni, double fac, const SparseOut &S) {
    const int grid = std::min((ni+3)/4, 256*8);
    const int *cells = (const int*)ctx->b_bi_cells.p, *facets = (const int*)ctx->b_bi_facets.p;
    const unsigned *masks = (const unsigned*)ctx->b_bi_masks.p;
    const double *verts = (const double*)ctx->b_vertices.p;
    if (ctx->P.bkn.fast)
        hipLaunchKernelGGL((k_boundary_items<DIM, DPE, 1>), dim3(grid), dim3(PNL_NTHREADS), 0, ctx->stream, ctx->P, verts, cells, facets,
                           masks, ni, fac, S, (double*)nullptr, (const unsigned*)nullptr, (const int*)nullptr, (const DevKernel*)nullptr, (const DevFormula*)nullptr);
    else
        hipLaunchKernelGGL((k_boundary_items<DIM, DPE, 0>), dim3(grid), dim3(PNL_NTHREADS), 0, ctx->stream, ctx->P, verts, cells, facets,
                           masks, ni, fac, S, (double*)nullptr, (const unsigned*)nullptr, (const int*)nullptr, (const DevKernel*)nullptr, (const DevFormula*)nullptr);
    HIPCHK(ctx, hipGetLastError());
    return PNL_OK;
}


// ---- tiled near-field assembly -------------------------------------------------------------------------------------------
template <int DIM, int DPE, int TILE, int KT>
int clusters_tiled_impl(pnl_context *ctx, const pnl_cluster_plan *pl, ClusterTiles CT, int cluster_boundary, const int *d_cell,
                        const int *d_pair, const int2 *sing_dev[3], const int *sing_pair_dev[3], const int *pair_foff, const int *fvid,
                        const double *fgeo, int maxf, const int *bt_cell, const int *bt_facet, const unsigned *bt_slot) {
    using S = TileSmem<DIM, DPE, TILE, KT == 0>;
    constexpr int ND = DPE*(DPE+1)/2;
    int rc;
    const int acc_stride = pl->chunk_stride+1;
    const size_t lds = S::fixed_bytes+sizeof(double)*(size_t)(pl->chunk_stride+1)*acc_stride;
    if (lds > 160*1024) return fail(ctx, PNL_ERR_UNSUPPORTED, "cluster chunk with %d DoFs: LDS sub-block of %zu bytes exceeds 160 KiB", pl->chunk_stride, lds);
    HIPCHK(ctx, hipMemsetAsync(ctx->b_counters.p, 0, sizeof(unsigned long long)*PNL_NCOUNTERS, ctx->stream));
    HIPCHK(ctx, hipMemsetAsync(CT.D, 0, sizeof(double)*(size_t)std::max(pl->num_dslots, 1)*ND, ctx->stream));
    HIPCHK(ctx, hipEventRecord(ctx->ev[0], ctx->stream));
    HIPCHK(ctx, hipEventRecord(ctx->ev[7], ctx->stream));
    ctx->pure_launched = false;
    // work list of the orders the tiles do not integrate themselves
    {
        const double pairs = (double)pl->ntiles*TILE*TILE;
        const size_t want = (size_t)std::min<double>(std::max<double>(pairs*0.25, 1<<20), 400e6);
        if (ctx->wl_cap < want) {
            if ((rc = ensure(ctx, ctx->b_wl, want*sizeof(int4)))) return rc;
            ctx->wl_cap = (unsigned)want;
        }
        if ((rc = ensure(ctx, ctx->b_wlds, (size_t)ctx->wl_cap*sizeof(int2)))) return rc;
        if ((rc = ensure(ctx, ctx->b_wlpair, (size_t)ctx->wl_cap*sizeof(int)))) return rc;
        if ((rc = ensure(ctx, ctx->b_wlcount, sizeof(unsigned)*PNL_WL_SLOTS))) return rc;
        HIPCHK(ctx, hipMemsetAsync(ctx->b_wlcount.p, 0, sizeof(unsigned), ctx->stream));
        ctx->wl_slots = 1; ctx->wl_cap_each = ctx->wl_cap;
        if ((rc = ensure(ctx, ctx->b_tilectr, sizeof(unsigned)))) return rc;
        HIPCHK(ctx, hipMemsetAsync(ctx->b_tilectr.p, 0, sizeof(unsigned), ctx->stream));
        CT.wl_ds = (int2*)ctx->b_wlds.p; CT.wl_pair = (int*)ctx->b_wlpair.p;
    }
    if (pl->ntiles > 0) {
        auto kfun = k_tile_distant<DIM, DPE, TILE, KT, true>;
        HIPCHK(ctx, hipFuncSetAttribute((const void*)kfun, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        int per_cu = 2;
        (void)hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, (const void*)kfun, tile_threads(DPE, KT), lds);
        if (pnl_tune("PNL_VERBOSE")) fprintf(stderr, "[pnl] cluster tiles=%d nU=%d lds=%zu bytes, occupancy API: %d blocks/CU\n", pl->ntiles,
                                           pl->chunk_stride, lds, per_cu);
        const int grid = pnl_grid_cap(std::min(pl->ntiles, 256*std::max(per_cu, 1)));
        hipLaunchKernelGGL(kfun, dim3(grid), dim3(tile_threads(DPE, KT)), lds, ctx->stream, ctx->P, (const int2*)nullptr, (double*)nullptr, 0ll,
                           (double*)nullptr, 0, ctx->nc, acc_stride, (int4*)ctx->b_wl.p, (unsigned*)ctx->b_wlcount.p, ctx->wl_cap, 0,
                           pl->ntiles, CT, (unsigned*)ctx->b_tilectr.p, SlotOut{});
        HIPCHK(ctx, hipGetLastError());
    }
    HIPCHK(ctx, hipEventRecord(ctx->ev[6], ctx->stream));
    // behind the tile kernel everything only adds (sparse data, diagonal-block buffer) with atomics: the touching pairs run on side
    // stream 0 and the cluster-local boundary term on side stream 1 next to the work-list kernels; joined before the diagonal
    // blocks are scattered.  The phase timers then show what is left of them after the work list
    hipStream_t const main_stream = ctx->stream;
    const bool overlap = !pnl_tune("PNL_NO_OVERLAP");
    struct StreamGuard { pnl_context *c; hipStream_t s; ~StreamGuard() { c->stream = s; } } stream_guard{ctx, main_stream};
    if (overlap) {
        HIPCHK(ctx, hipEventRecord(ctx->ev_fold, main_stream));
        HIPCHK(ctx, hipStreamWaitEvent(ctx->aux[0], ctx->ev_fold, 0));
        HIPCHK(ctx, hipStreamWaitEvent(ctx->aux[1], ctx->ev_fold, 0));
    }
    {
        if ((rc = ensure(ctx, ctx->b_wlsorted, (size_t)ctx->wl_cap*sizeof(int4)))) return rc;
        if ((rc = ensure(ctx, ctx->b_wlaux, sizeof(unsigned)*(4*(PNL_WL_BINS+1))))) return rc;
        unsigned *hist = (unsigned*)ctx->b_wlaux.p, *offs = hist+(PNL_WL_BINS+1), *coff = offs+(PNL_WL_BINS+1), *cursor = coff+(PNL_WL_BINS+1);
        HIPCHK(ctx, hipMemsetAsync(hist, 0, sizeof(unsigned)*(PNL_WL_BINS+1), ctx->stream));
        const int4 *wl = (const int4*)ctx->b_wl.p;
        const unsigned *wlc = (const unsigned*)ctx->b_wlcount.p;
        hipLaunchKernelGGL(k_wl_hist, dim3(512), dim3(PNL_NTHREADS), 0, ctx->stream, wl, wlc, ctx->wl_cap, hist);
        hipLaunchKernelGGL(k_wl_scan, dim3(1), dim3(64), 0, ctx->stream, (const unsigned*)hist, offs, coff, cursor);
        hipLaunchKernelGGL(k_wl_scatter, dim3(512), dim3(PNL_NTHREADS), 0, ctx->stream, wl, wlc, ctx->wl_cap, (const unsigned*)offs, cursor,
                           (int4*)ctx->b_wlsorted.p);
        const int st = 4+DPE;
        // LDS copy of the rule: see run_worklist (PNL_WL_CL_KB: A/B switch of the cluster path)
        const int wl_kb = pnl_tune("PNL_WL_CL_KB") ? std::max(4, atoi(pnl_tune("PNL_WL_CL_KB"))) : 18;      // 60 KB / two workgroups per CU: + 4 ms at C4
        const int tab_max = wl_tab_max<DPE>(wl_kb);
        const size_t wlds = wl_sorted_lds<DPE>(tab_max);
        const int wl_grid = 256*std::max(1, std::min(8, 150/(wl_kb+(KT == 0 ? 3 : 0))));
        auto wfun = k_worklist_sorted<DIM, DPE, KT, false>;
        HIPCHK(ctx, hipFuncSetAttribute((const void*)wfun, hipFuncAttributeMaxDynamicSharedMemorySize, (int)wlds));
        hipLaunchKernelGGL((k_worklist_lane<DIM, DPE, KT, false>), dim3(256*4), dim3(PNL_NTHREADS), wl_lane_lds(k_worklist_lane<DIM, DPE, KT, false>, DPE, KT), ctx->stream, ctx->P,
                           (const int4*)ctx->b_wlsorted.p, (const unsigned*)offs, (double*)nullptr, 0ll, (double*)nullptr, SparseOut{}, 0, CT);
        hipLaunchKernelGGL(wfun, dim3(wl_grid), dim3(PNL_NTHREADS), wlds, ctx->stream, ctx->P, (const int4*)ctx->b_wlsorted.p,
                           (const unsigned*)offs, (const unsigned*)coff, (double*)nullptr, 0ll, (double*)nullptr, tab_max, SparseOut{},
                           PNL_WL_BINS-1, PNL_WL_LANE_MAXPTS+1, CT);
        HIPCHK(ctx, hipGetLastError());
    }
    HIPCHK(ctx, hipEventRecord(ctx->ev[1], ctx->stream));
    HIPCHK(ctx, hipEventRecord(ctx->ev[2], ctx->stream));
    // touching element pairs
    if (overlap) ctx->stream = ctx->aux[0];
    for (int s = 0; s < DIM+1; s++) {
        const int np = pl->n_sing[s];
        if (!np) continue;
        if (!ctx->C().have_sing[0][s]) return fail(ctx, PNL_ERR_STATE, "singular rule for %d common vertices not uploaded", s+1);
        ClusterTiles C2 = CT;
        C2.sing_pair = sing_pair_dev[s];
        const int M = ctx->P.sM[s], rows = ctx->P.sRows[s];
        const size_t slds = sizeof(double)*(size_t)(2*(DIM+1)+1+rows)*M;
        const int wpb = PNL_SING_THREADS/64;
        const bool stage = slds <= 150*1024;
        const int per_cu = stage ? std::max(1, (int)((160*1024)/std::max<size_t>(slds, 1))) : 4;
        const int grid = std::min((np+wpb-1)/wpb, 256*std::min(per_cu, 4));
#define PNL_LAUNCH_SING(SLOT)                                                                                                          \
        if (stage) {                                                                                                                   \
            auto kf = k_singular_pairs<DIM, DPE, SLOT, KT, true, false>;                                                               \
            HIPCHK(ctx, hipFuncSetAttribute((const void*)kf, hipFuncAttributeMaxDynamicSharedMemorySize, (int)slds));                 \
            hipLaunchKernelGGL(kf, dim3(grid), dim3(PNL_SING_THREADS), slds, ctx->stream, ctx->P, sing_dev[s], np, (double*)nullptr, 0ll, \
                               0, ctx->nc, SparseOut{}, (const int4*)nullptr, (const unsigned*)nullptr, C2);                          \
        } else                                                                                                                         \
            hipLaunchKernelGGL((k_singular_pairs<DIM, DPE, SLOT, KT, false, false>), dim3(grid), dim3(PNL_SING_THREADS), 0, ctx->stream, \
                               ctx->P, sing_dev[s], np, (double*)nullptr, 0ll, 0, ctx->nc, SparseOut{}, (const int4*)nullptr,          \
                               (const unsigned*)nullptr, C2);
        if (s == 0) { PNL_LAUNCH_SING(0) }
        else if (s == 1) { PNL_LAUNCH_SING(1) }
        else { PNL_LAUNCH_SING((DIM == 2 ? 2 : 1)) }
#undef PNL_LAUNCH_SING
        HIPCHK(ctx, hipGetLastError());
    }
    if (overlap) {
        HIPCHK(ctx, hipEventRecord(ctx->ev_join[0], ctx->aux[0]));
        ctx->stream = ctx->aux[1];
    } else HIPCHK(ctx, hipEventRecord(ctx->ev[3], ctx->stream));
    // cluster-local Gauss-theorem term into the diagonal-block buffer
    if (cluster_boundary && pl->num_dslots > 0 && pl->nfacets > 0) {
        if (!ctx->C().have_kernel[1] || !ctx->C().have_form[1]) return fail(ctx, PNL_ERR_STATE, "boundary kernel and order formula must be set");
        // few cells (those of cellsInter), many facets each: small facet chunks give the parallelism
        // (measured at C4, 9.9e6 pairs / 5.0e8 point pairs: all in place 9.0 ms; list for > 48 point pairs 4.7 ms, > 200: 4.1 ms,
        // > 200 with 4 facets per chunk 3.2 ms, > 1000: 5.4 ms)
        const int per = pnl_tune("PNL_CB_PER") ? atoi(pnl_tune("PNL_CB_PER")) : 4;
        const dim3 grid((pl->num_dslots+PNL_NTHREADS-1)/PNL_NTHREADS, (maxf+per-1)/per);
        const double *verts = (const double*)ctx->b_vertices.p;
        // (cell, facet) pairs with more than `defer` point pairs: list of items for k_boundary_items (one per wave); pairs that do
        // not fit into the list are integrated in place
        const int defer = pnl_tune("PNL_CB_DEFER") ? atoi(pnl_tune("PNL_CB_DEFER")) : 200;
        const unsigned cap = defer > 0 ? 1u << 22 : 0u;
        int *dcells = nullptr, *dfacets = nullptr;
        unsigned *dslots = nullptr, *dcount = nullptr;
        if (cap) {
            if ((rc = ensure(ctx, ctx->b_bdefer, sizeof(int)*(size_t)cap*(3+DIM)+sizeof(unsigned)))) return rc;
            dcells = (int*)ctx->b_bdefer.p; dfacets = dcells+cap; dslots = (unsigned*)(dfacets+(size_t)cap*DIM);
            dcount = (unsigned*)((int*)(dslots+cap)+cap);
            HIPCHK(ctx, hipMemsetAsync(dcount, 0, sizeof(unsigned), ctx->stream));
        }
        if (ctx->P.bkn.fast)
            hipLaunchKernelGGL((k_cluster_boundary<DIM, DPE, 1>), grid, dim3(PNL_NTHREADS), 0, ctx->stream, ctx->P, verts, d_cell, d_pair,
                               pl->num_dslots, pair_foff, fvid, fgeo, pl->nfacets, CT.D, per, defer, dcells, dfacets, dslots, dcount, cap);
        else
            hipLaunchKernelGGL((k_cluster_boundary<DIM, DPE, 0>), grid, dim3(PNL_NTHREADS), 0, ctx->stream, ctx->P, verts, d_cell, d_pair,
                               pl->num_dslots, pair_foff, fvid, fgeo, pl->nfacets, CT.D, per, defer, dcells, dfacets, dslots, dcount, cap);
        HIPCHK(ctx, hipGetLastError());
        if (cap) {
            if (ctx->P.bkn.fast)
                hipLaunchKernelGGL((k_boundary_items<DIM, DPE, 1>), dim3(256*8), dim3(PNL_NTHREADS), 0, ctx->stream, ctx->P, verts,
                                   (const int*)dcells, (const int*)dfacets, (const unsigned*)dslots, (int)cap, 1., SparseOut{}, CT.D,
                                   (const unsigned*)dcount, (const int*)nullptr, (const DevKernel*)nullptr, (const DevFormula*)nullptr);
            else
                hipLaunchKernelGGL((k_boundary_items<DIM, DPE, 0>), dim3(256*8), dim3(PNL_NTHREADS), 0, ctx->stream, ctx->P, verts,
                                   (const int*)dcells, (const int*)dfacets, (const unsigned*)dslots, (int)cap, 1., SparseOut{}, CT.D,
                                   (const unsigned*)dcount, (const int*)nullptr, (const DevKernel*)nullptr, (const DevFormula*)nullptr);
            HIPCHK(ctx, hipGetLastError());
        }
        if (pl->n_btouch > 0) {
            for (int s = 0; s < DIM; s++)
                if (!ctx->C().have_sing[1][s]) return fail(ctx, PNL_ERR_STATE, "boundary singular rule for %d common vertices not uploaded", s+1);
            const int g2 = std::min((pl->n_btouch+3)/4, 256*8);
            if (ctx->P.bkn.fast)
                hipLaunchKernelGGL((k_boundary_items<DIM, DPE, 1>), dim3(g2), dim3(PNL_NTHREADS), 0, ctx->stream, ctx->P, verts, bt_cell,
                                   bt_facet, bt_slot, pl->n_btouch, 1., SparseOut{}, CT.D, (const unsigned*)nullptr, (const int*)nullptr, (const DevKernel*)nullptr, (const DevFormula*)nullptr);
            else
                hipLaunchKernelGGL((k_boundary_items<DIM, DPE, 0>), dim3(g2), dim3(PNL_NTHREADS), 0, ctx->stream, ctx->P, verts, bt_cell,
                                   bt_facet, bt_slot, pl->n_btouch, 1., SparseOut{}, CT.D, (const unsigned*)nullptr, (const int*)nullptr, (const DevKernel*)nullptr, (const DevFormula*)nullptr);
            HIPCHK(ctx, hipGetLastError());
        }
    }
    if (overlap) {
        HIPCHK(ctx, hipEventRecord(ctx->ev_join[1], ctx->aux[1]));
        ctx->stream = main_stream;
        HIPCHK(ctx, hipStreamWaitEvent(main_stream, ctx->ev_join[0], 0));
        HIPCHK(ctx, hipStreamWaitEvent(main_stream, ctx->ev_join[1], 0));
        HIPCHK(ctx, hipEventRecord(ctx->ev[3], ctx->stream));
    }
    HIPCHK(ctx, hipEventRecord(ctx->ev[4], ctx->stream));
    if (pl->num_dslots > 0) {
        const long long nt = (long long)pl->num_dslots*DPE*DPE;
        hipLaunchKernelGGL((k_cluster_scatter_diag<DPE>), dim3((unsigned)((nt+PNL_NTHREADS-1)/PNL_NTHREADS)), dim3(PNL_NTHREADS), 0,
                           ctx->stream, ctx->P, CT, d_cell, d_pair, pl->num_dslots);
        HIPCHK(ctx, hipGetLastError());
    }
    HIPCHK(ctx, hipEventRecord(ctx->ev[5], ctx->stream));
    ctx->ev_valid = true;
    ctx->tiles_launched = true;
    return PNL_OK;
}

int check_ready(pnl_context *ctx) {
    for (auto *c : ctx->cls)
        if (!c->have_kernel[0] || !c->have_form[0] || !ctx->have_rules)
            return fail(ctx, PNL_ERR_STATE, "kernel, order formula and distant rules must be set (for every class) before assembling");
    return PNL_OK;
}

int dispatch(pnl_context *ctx, double *A, int64_t ldA, int zero_exterior, int ntiles, int cell_begin, int cell_end, int flags) {
    for (auto *c : ctx->cls)
        if (!std::isinf(c->kern[0].horizon2))
            return fail(ctx, PNL_ERR_UNSUPPORTED, "finite-horizon kernels are assembled from an explicit pair list (pnl_assemble_pairs_masked, "
                        "nonlocalBuilder.getSparse): the all-pairs dense loop has no REMOTE / CUT handling");
    refresh_tables(ctx);
    if (ctx->dim == 2 && ctx->dpe == 3) return assemble_impl<2, 3, TILE_P1>(ctx, A, ldA, zero_exterior, ntiles, cell_begin, cell_end, flags);
    if (ctx->dim == 2 && ctx->dpe == 6) return assemble_impl<2, 6, TILE_P2>(ctx, A, ldA, zero_exterior, ntiles, cell_begin, cell_end, flags);
    if (ctx->dim == 1 && ctx->dpe == 2) return assemble_impl<1, 2, TILE_P1>(ctx, A, ldA, zero_exterior, ntiles, cell_begin, cell_end, flags);
    if (ctx->dim == 1 && ctx->dpe == 3) return assemble_impl<1, 3, TILE_P2>(ctx, A, ldA, zero_exterior, ntiles, cell_begin, cell_end, flags);
    // P0 and P3 on intervals (the reference's fixtures --elementP0 / --elementP3; FL1 is generic in the DoFs per element)
    if (ctx->dim == 2 && ctx->dpe == 1) return assemble_impl<2, 1, TILE_P1>(ctx, A, ldA, zero_exterior, ntiles, cell_begin, cell_end, flags);
    if (ctx->dim == 1 && ctx->dpe == 1) return assemble_impl<1, 1, TILE_P1>(ctx, A, ldA, zero_exterior, ntiles, cell_begin, cell_end, flags);
    if (ctx->dim == 1 && ctx->dpe == 4) return assemble_impl<1, 4, TILE_P2>(ctx, A, ldA, zero_exterior, ntiles, cell_begin, cell_end, flags);
    return fail(ctx, PNL_ERR_UNSUPPORTED, "unsupported (dim=%d, dofs_per_element=%d)", ctx->dim, ctx->dpe);
}

// ---- GEMV / CG kernels ---------------------------------------------------------------------------
__global__ void __launch_bounds__(PNL_NTHREADS)
k_gemv(const double *__restrict__ A, long long ldA, int n, const double *__restrict__ x, double *__restrict__ y) {
    // one wave per row; 16 B per lane per load
    const int lane = threadIdx.x & 63;
    const int row = (blockIdx.x*PNL_NTHREADS+threadIdx.x) >> 6;
    if (row >= n) return;
    const double *__restrict__ a = A+(long long)row*ldA;
    double s0 = 0., s1 = 0., s2 = 0., s3 = 0.;
    // 16-byte loads need both the row and x aligned (x may be a storage-offset view); four of them per lane in flight
    // (6.0 TB/s at n = 48,769 against 5.4 with one)
    const bool aligned = ((((uintptr_t)a) | ((uintptr_t)x)) & 15) == 0;
    if (aligned) {
        const int n2 = n >> 1;
        const double2 *a2 = (const double2*)a;
        const double2 *x2 = (const double2*)x;
        int j = lane;
        for (; j+192 < n2; j += 256) {
            const double2 v0 = a2[j], v1 = a2[j+64], v2 = a2[j+128], v3 = a2[j+192];
            const double2 w0 = x2[j], w1 = x2[j+64], w2 = x2[j+128], w3 = x2[j+192];
            s0 = __builtin_fma(v0.x, w0.x, s0); s1 = __builtin_fma(v0.y, w0.y, s1);
            s2 = __builtin_fma(v1.x, w1.x, s2); s3 = __builtin_fma(v1.y, w1.y, s3);
            s0 = __builtin_fma(v2.x, w2.x, s0); s1 = __builtin_fma(v2.y, w2.y, s1);
            s2 = __builtin_fma(v3.x, w3.x, s2); s3 = __builtin_fma(v3.y, w3.y, s3);
        }
        for (; j < n2; j += 64) {
            const double2 av = a2[j], xv = x2[j];
            s0 = __builtin_fma(av.x, xv.x, s0);
            s1 = __builtin_fma(av.y, xv.y, s1);
        }
        if ((n & 1) && lane == 0) s0 = __builtin_fma(a[n-1], x[n-1], s0);
    } else {
        for (int j = lane; j < n; j += 64) s0 = __builtin_fma(a[j], x[j], s0);
    }
    const double s = wave_sum((s0+s1)+(s2+s3));
    if (lane == 0) y[row] = s;
}

// y += (A^T) x contribution for the one-sided storage: each wave takes a row I and adds A[I,J] x_I to y_J
__global__ void __launch_bounds__(PNL_NTHREADS)
k_gemv_t_add(const double *__restrict__ A, long long ldA, int n, const double *__restrict__ x, double *__restrict__ y,
             int rows_per_block) {
    // block handles rows [r0, r1): lane-owned column sums, then one atomic per column
    const int r0 = blockIdx.y*rows_per_block, r1 = min(n, r0+rows_per_block);
    const int j = blockIdx.x*PNL_NTHREADS+threadIdx.x;
    if (j >= n) return;
    double s = 0.;
    for (int r = r0; r < r1; r++) s = __builtin_fma(A[(long long)r*ldA+j], x[r], s);
    if (s != 0.) atomic_add_f64(&y[j], s);
}

__global__ void __launch_bounds__(PNL_NTHREADS) k_dot(const double *__restrict__ a, const double *__restrict__ b, int n, double *out) {
    double s = 0.;
    for (int i = blockIdx.x*PNL_NTHREADS+threadIdx.x; i < n; i += gridDim.x*PNL_NTHREADS) s = __builtin_fma(a[i], b[i], s);
    s = wave_sum(s);
    if ((threadIdx.x & 63) == 0) atomic_add_f64(out, s);
}

__global__ void __launch_bounds__(PNL_NTHREADS) k_diag_inv(const double *__restrict__ A, long long ldA, int n, double *__restrict__ dinv) {
    const int i = blockIdx.x*PNL_NTHREADS+threadIdx.x;
    if (i < n) dinv[i] = 1./A[(long long)i*ldA+i];
}

// r = b - Ax given Ax in t ; z = dinv*r ; p = z
__global__ void __launch_bounds__(PNL_NTHREADS)
k_cg_init(const double *__restrict__ b, const double *__restrict__ Ax, const double *__restrict__ dinv, int n, double *r, double *p) {
    const int i = blockIdx.x*PNL_NTHREADS+threadIdx.x;
    if (i < n) { const double ri = b[i]-Ax[i]; r[i] = ri; p[i] = dinv[i]*ri; }
}

// x += alpha p ; r -= alpha Ap ; z = dinv r   (alpha = scal[0]/scal[1])
__global__ void __launch_bounds__(PNL_NTHREADS)
k_cg_update(const double *__restrict__ scal, const double *__restrict__ p, const double *__restrict__ Ap,
            const double *__restrict__ dinv, int n, double *x, double *r, double *z) {
    const int i = blockIdx.x*PNL_NTHREADS+threadIdx.x;
    const double alpha = scal[0]/scal[1];
    if (i < n) {
        x[i] = __builtin_fma(alpha, p[i], x[i]);
        const double ri = __builtin_fma(-alpha, Ap[i], r[i]);
        r[i] = ri;
        z[i] = dinv[i]*ri;
    }
}

// p = z + (beta/betaOld) p   (beta = scal[2], betaOld = scal[0])
__global__ void __launch_bounds__(PNL_NTHREADS)
k_cg_dir(const double *__restrict__ scal, const double *__restrict__ z, int n, double *p) {
    const int i = blockIdx.x*PNL_NTHREADS+threadIdx.x;
    const double t = scal[2]/scal[0];
    if (i < n) p[i] = __builtin_fma(t, p[i], z[i]);
}

}  // namespace

// =================================================================================================
// Are all cell pairs of the tile (block a, block b) distant pairs of ONE quadrature order q <= qlimit?  Returns q or 0.
// Conservative bounds on the order formula (FL2:622-642 / FL1:234-253)
//   order = max(ceil f(1,2), ceil f(2,1), 2),  f(self, other) = (c0 + a L_other + b max(L_self, L_other) - e ln(d/h_other)) / (max(ln(d/h_self), 0) + den0)
// over the tile: the distance of the cell centres lies in [dmin, dmax] = |centre_a - centre_b| -+ (rad_a + rad_b), h in
// [hmin, hmax] and L = |ln(h/H0)| in [Lmin, Lmax] per block.  f <= num_max/den_min for both roles gives order <= q, and
// f >= num_min/den_max > q-1 for ONE role (for all pairs of the tile) gives order >= q.  dmin > hmax_a + hmax_b also rules
// out shared vertices (a vertex is closer than 2/3 h to its cell's centre).  Anything not provably uniform goes to the
// general kernel, whose per-pair formula decides.
static int tile_uniform_order(const pnl_context *ctx, const pnl_order_formula &F, int ta, int tb, int qlimit) {
    if (ta == tb) return 0;
    const auto &A = ctx->blocks[ta], &B = ctx->blocks[tb];
    if (!A.full || !B.full || !(F.e >= 0.) || !(F.den0 > 0.)) return 0;
    const double dx = A.cx-B.cx, dy = A.cy-B.cy, dc = std::sqrt(dx*dx+dy*dy);
    const double dmin = dc-A.rad-B.rad, dmax = dc+A.rad+B.rad;
    if (!(dmin > A.hmax+B.hmax)) return 0;
    typedef pnl_context::BlockAgg Agg;
    auto upper = [&](const Agg &S, const Agg &O, double q) {       // f(S, O) <= q for every pair
        const double l_self = std::log(dmin/S.hmax), n_other = std::log(dmin/O.hmax);      // both > 0
        const double aL = std::max(F.a*O.Lmin, F.a*O.Lmax);
        const double bL = std::max(F.b*std::max(S.Lmin, O.Lmin), F.b*std::max(S.Lmax, O.Lmax));
        const double num = F.c0+aL+bL-F.e*n_other, den = l_self+F.den0;
        return num <= q*den*(1.-1e-9)-1e-9;
    };
    auto lower = [&](const Agg &S, const Agg &O, double q) {       // f(S, O) > q for every pair
        const double l_self = std::log(dmax/S.hmin), n_other = std::log(dmax/O.hmin);
        const double aL = std::min(F.a*O.Lmin, F.a*O.Lmax);
        const double bL = std::min(F.b*std::max(S.Lmin, O.Lmin), F.b*std::max(S.Lmax, O.Lmax));
        const double num = F.c0+aL+bL-F.e*n_other, den = l_self+F.den0;
        return den > 0. && num >= q*den*(1.+1e-9)+1e-9;
    };
    for (int q = 2; q <= qlimit; q++)
        if (upper(A, B, q) && upper(B, A, q)) {
            if (q == 2) return 2;
            return (lower(A, B, q-1) || lower(B, A, q-1)) ? q : 0;
        }
    return 0;
}
static bool tile_is_uniform(const pnl_context *ctx, const pnl_order_formula &F, int ta, int tb) { return tile_uniform_order(ctx, F, ta, tb, 2) == 2; }

namespace {
template <int DIM, int DPE>
int pointwise_impl(pnl_context *ctx, double *A, int64_t ldA, int zero_exterior, int cell_begin, int cell_end, int npairs,
                   int nbpairs) {
    // P1: tile kernels with LDS sub-blocks (k_pw_tile, k_pw_mixed, k_pw_lane).  P2 (FL2:894-1184 is element-agnostic): every
    // distant pair through classification, the sorted work list and k_pw_distant (16 lanes per pair, global atomics)
    constexpr int NV = DIM+1, ND = DPE*(DPE+1)/2, ST = 4+DPE;
    constexpr bool P1el = DPE == NV;
    const bool P1 = P1el && ctx->pw.type != 5;           // a P1 order function (type 5) is known per cell: the generic kernels
    int rc;
    DevProblem &P = ctx->P;
    P.qmax = ctx->qmax;
    P.off = (const int*)ctx->b_off.p; P.bary = (const double*)ctx->b_bary.p; P.w = (const double*)ctx->b_w.p;
    P.phi = (const double*)ctx->b_phi.p; P.foff = (const int*)ctx->b_foff.p; P.fbary = (const double*)ctx->b_fbary.p;
    P.fw = (const double*)ctx->b_fw.p;
    P.cur_class = -1;
    const PwDev &W = ctx->pw;
    HIPCHK(ctx, hipMemsetAsync(ctx->b_counters.p, 0, sizeof(unsigned long long)*PNL_NCOUNTERS, ctx->stream));
    HIPCHK(ctx, hipMemsetAsync(ctx->b_D.p, 0, sizeof(double)*(size_t)ctx->ncp*ND, ctx->stream));
    HIPCHK(ctx, hipEventRecord(ctx->ev[0], ctx->stream));
    ctx->tiles_launched = true; ctx->pure_launched = false;
    const int nbk = (ctx->nc+63)/64;
    {
        // work list: only the pairs whose rule has more than 16 points arrive there when the tiles evaluate the others
        // themselves (overflow is detected by pnl_get_counters); the tile-less variant lists every pair of the cell range
        double pairs = 0.;
        for (long long c = cell_begin; c < cell_end; c++) pairs += (double)(ctx->nc-c);
        const bool tiles_evaluate = P1 && ctx->tile == 64 && !pnl_tune("PNL_PW_NOMIXED");
        const size_t want = tiles_evaluate ? (size_t)std::min<double>(std::max<double>(pairs*0.1, 1 << 20), 400e6)
                                           : (size_t)std::max<double>(pairs, 1024.);
        if (want > 1500000000ull) return fail(ctx, PNL_ERR_UNSUPPORTED, "%zu pairs exceed the work list of the pointwise path", want);
        if (ctx->wl_cap < want) {
            if ((rc = ensure(ctx, ctx->b_wl, want*sizeof(int4)))) return rc;
            ctx->wl_cap = (unsigned)want;
        }
        if ((rc = ensure(ctx, ctx->b_wlsorted, (size_t)ctx->wl_cap*sizeof(int4)))) return rc;
        if ((rc = ensure(ctx, ctx->b_wlcount, sizeof(unsigned)*PNL_WL_SLOTS))) return rc;
        if ((rc = ensure(ctx, ctx->b_wlaux, sizeof(unsigned)*(4*(PNL_WL_BINS+1))))) return rc;
        HIPCHK(ctx, hipMemsetAsync(ctx->b_wlcount.p, 0, sizeof(unsigned), ctx->stream));
        ctx->wl_slots = 1; ctx->wl_cap_each = ctx->wl_cap;
    }
    // block tiles of the upper triangle: uniform ones (every pair provably of order 2 for every pair order in the range of
    // the two blocks) go to k_pw_tile, the others through classification and the sorted work list
    std::vector<int2> mixed, uniform;
    {
        const int T = 64;
        std::vector<double> smin(nbk, 1e300), smax(nbk, -1e300);
        for (int c = 0; c < ctx->nc; c++) {
            const int b = c/T;
            smin[b] = std::min(smin[b], ctx->pw_cell_smax[c]); smax[b] = std::max(smax[b], ctx->pw_cell_smax[c]);
        }
        auto formula = [&](double sv) {
            pnl_order_formula F;
            std::memset(&F, 0, sizeof(F));
            F.c0 = W.c0;
            if (DIM == 2) { F.a = sv-1.; F.b = 1.; F.e = sv; F.den0 = 0.4; } else { F.a = 2.*sv-1.; F.b = 0.; F.e = 2.*sv; F.den0 = 0.8; }
            return F;
        };
        const bool allow = P1 && ctx->tile == T && ctx->qmax >= 2 && !pnl_tune("PNL_PW_NOTILE");
        const int a0 = cell_begin/T, a1 = (cell_end+T-1)/T;
        for (int d = 0; d < nbk; d++)
            for (int a = a0; a < a1 && a+d < nbk; a++) {
                const int b = a+d;
                // the pair order max(m_c1, m_c2) lies between the larger of the block minima and the larger of the maxima; the
                // formula is linear in it, so the two end points bound it
                const double lo = std::max(smin[a], smin[b]), hi = std::max(smax[a], smax[b]);
                bool u = allow && a*T >= cell_begin && (a+1)*T <= cell_end && tile_is_uniform(ctx, formula(lo), a, b) &&
                         tile_is_uniform(ctx, formula(hi), a, b);
                (u ? uniform : mixed).push_back(make_int2(a, b));
            }
        std::vector<int2> all(mixed);
        all.insert(all.end(), uniform.begin(), uniform.end());
        if ((rc = upload(ctx, ctx->b_tiles, all.data(), all.size()))) return rc;
        ctx->tiles_cached.clear(); ctx->tiles_cb = -1;            // b_tiles no longer holds the dense tile list
    }
    if constexpr (P1el) if (P1 && !uniform.empty()) {
        const int acc_stride = ctx->nU+1;
        constexpr int NP = DIM == 2 ? 3 : 2;
        const size_t lds = sizeof(double)*(64*NP*DIM+2*64*NP+64+2*64*ND)+sizeof(int)*(64*DPE+64)
                           +2*sizeof(double)*(size_t)(ctx->nU+1)*acc_stride;
        auto tfun = k_pw_tile<DIM>;
        HIPCHK(ctx, hipFuncSetAttribute((const void*)tfun, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        int per_cu = 1;
        (void)hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, (const void*)tfun, PNL_NTHREADS, lds);
        const int grid = pnl_grid_cap(std::min((int)uniform.size(), 256*std::max(per_cu, 1)));
        hipLaunchKernelGGL(tfun, dim3(grid), dim3(PNL_NTHREADS), lds, ctx->stream, P, W, (const int2*)ctx->b_tiles.p+mixed.size(),
                           (int)uniform.size(), A, (long long)ldA, (double*)ctx->b_D.p, acc_stride);
        HIPCHK(ctx, hipGetLastError());
        ctx->pure_launched = true;
    }
    HIPCHK(ctx, hipEventRecord(ctx->ev[7], ctx->stream));
    // the other tiles: classification, in-tile evaluation of the rules with at most 16 points (LDS sub-blocks), work list for the rest
    const bool in_tile = P1 && ctx->tile == 64 && !pnl_tune("PNL_PW_NOMIXED");
    if constexpr (P1el) if (!mixed.empty() && in_tile) {
        const int acc_stride = ctx->nU+1;
        const size_t lds = sizeof(double)*(PNL_PW_LANE_MAXPTS*ST+64*PNL_PW_LANE_MAXPTS*2+2*64*ND)+sizeof(unsigned short)*64*64
                           +sizeof(int)*(3*PNL_PW_NBUCK+2*64*DPE)+2*sizeof(double)*(size_t)(ctx->nU+1)*acc_stride;
        if (lds > 160*1024) return fail(ctx, PNL_ERR_UNSUPPORTED, "a block of 64 cells touches %d DoFs: LDS sub-blocks of %zu bytes exceed 160 KiB", ctx->nU, lds);
        auto mfun = k_pw_mixed<DIM>;
        HIPCHK(ctx, hipFuncSetAttribute((const void*)mfun, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        int per_cu = 1;
        (void)hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, (const void*)mfun, PNL_NTHREADS, lds);
        const int grid = pnl_grid_cap(std::min((int)mixed.size(), 256*std::max(per_cu, 1)));
        if ((rc = ensure(ctx, ctx->b_tilectr, sizeof(unsigned)))) return rc;
        HIPCHK(ctx, hipMemsetAsync(ctx->b_tilectr.p, 0, sizeof(unsigned), ctx->stream));
        hipLaunchKernelGGL(mfun, dim3(grid), dim3(PNL_NTHREADS), lds, ctx->stream, P, W, (const int2*)ctx->b_tiles.p, (int)mixed.size(), A,
                           (long long)ldA, (double*)ctx->b_D.p, acc_stride, (int4*)ctx->b_wl.p, (unsigned*)ctx->b_wlcount.p, ctx->wl_cap,
                           cell_begin, cell_end, (unsigned*)ctx->b_tilectr.p);
    }
    if (!mixed.empty() && !in_tile)
        hipLaunchKernelGGL((k_pw_classify<DIM, DPE>), dim3((unsigned)mixed.size()), dim3(PNL_NTHREADS), 0, ctx->stream, P, W,
                           (const int2*)ctx->b_tiles.p, (int4*)ctx->b_wl.p, (unsigned*)ctx->b_wlcount.p, ctx->wl_cap, cell_begin, cell_end);
    HIPCHK(ctx, hipGetLastError());
    HIPCHK(ctx, hipEventRecord(ctx->ev[6], ctx->stream));
    {
        unsigned *hist = (unsigned*)ctx->b_wlaux.p, *offs = hist+(PNL_WL_BINS+1), *coff = offs+(PNL_WL_BINS+1), *cursor = coff+(PNL_WL_BINS+1);
        HIPCHK(ctx, hipMemsetAsync(hist, 0, sizeof(unsigned)*(PNL_WL_BINS+1), ctx->stream));
        const int4 *wl = (const int4*)ctx->b_wl.p;
        const unsigned *wlc = (const unsigned*)ctx->b_wlcount.p;
        hipLaunchKernelGGL(k_wl_hist, dim3(512), dim3(PNL_NTHREADS), 0, ctx->stream, wl, wlc, ctx->wl_cap, hist);
        hipLaunchKernelGGL(k_wl_scan, dim3(1), dim3(64), 0, ctx->stream, (const unsigned*)hist, offs, coff, cursor);
        hipLaunchKernelGGL(k_wl_scatter, dim3(512), dim3(PNL_NTHREADS), 0, ctx->stream, wl, wlc, ctx->wl_cap, (const unsigned*)offs, cursor,
                           (int4*)ctx->b_wlsorted.p);
        hipLaunchKernelGGL(k_pw_stats, dim3(1), dim3(PNL_WL_BINS), 0, ctx->stream, P, (const unsigned*)hist);
        // LDS: rule table + order / scaling of the second cell's points for the 16 pairs of a chunk
        const int tab_max = 256;
        const size_t lds = sizeof(double)*((size_t)tab_max*ST+(size_t)(PNL_NTHREADS/16)*tab_max*2);
        auto kfun = k_pw_distant<DIM, DPE>;
        HIPCHK(ctx, hipFuncSetAttribute((const void*)kfun, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        const bool lane_kernel = P1 && !pnl_tune("PNL_PW_NOLANE");      // (with the in-tile evaluation only the rules of more than 16 points arrive here)
        if constexpr (P1el) if (lane_kernel)
            hipLaunchKernelGGL((k_pw_lane<DIM>), dim3(256*2), dim3(PNL_NTHREADS), 0, ctx->stream, P, W, (const int4*)ctx->b_wlsorted.p,
                               (const unsigned*)offs, A, (long long)ldA, (double*)ctx->b_D.p);
        hipLaunchKernelGGL(kfun, dim3(256*4), dim3(PNL_NTHREADS), lds, ctx->stream, P, W, (const int4*)ctx->b_wlsorted.p,
                           (const unsigned*)offs, A, (long long)ldA, (double*)ctx->b_D.p, tab_max, lane_kernel ? PNL_PW_LANE_MAXPTS+1 : 0, PwNear{});
        HIPCHK(ctx, hipGetLastError());
    }
    HIPCHK(ctx, hipEventRecord(ctx->ev[1], ctx->stream));
    HIPCHK(ctx, hipEventRecord(ctx->ev[2], ctx->stream));
    if (npairs > 0) {
        const unsigned grid = (unsigned)((2ll*npairs*64+PNL_NTHREADS-1)/PNL_NTHREADS);
        const int4 *pp = (const int4*)ctx->b_pw_pairs.p;
        hipLaunchKernelGGL((k_pw_singular<DIM, DPE, 0>), dim3(grid), dim3(PNL_NTHREADS), 0, ctx->stream, P, W, pp, npairs, A, (long long)ldA, cell_begin, cell_end, PwNear{}, (const int*)nullptr);
        hipLaunchKernelGGL((k_pw_singular<DIM, DPE, 1>), dim3(grid), dim3(PNL_NTHREADS), 0, ctx->stream, P, W, pp, npairs, A, (long long)ldA, cell_begin, cell_end, PwNear{}, (const int*)nullptr);
        if (DIM == 2)
            hipLaunchKernelGGL((k_pw_singular<DIM, DPE, (DIM == 2 ? 2 : 1)>), dim3(grid), dim3(PNL_NTHREADS), 0, ctx->stream, P, W, pp, npairs, A, (long long)ldA, cell_begin, cell_end, PwNear{}, (const int*)nullptr);
        HIPCHK(ctx, hipGetLastError());
    }
    HIPCHK(ctx, hipEventRecord(ctx->ev[3], ctx->stream));
    if (zero_exterior && cell_end > cell_begin) {
        const int ncell = cell_end-cell_begin, gx = (ncell+PNL_NTHREADS-1)/PNL_NTHREADS;
        int per = 16;
        while (per > 1 && (long long)gx*((ctx->nb+per-1)/per) < 4096) per >>= 1;
        hipLaunchKernelGGL((k_pw_boundary_distant<DIM, DPE>), dim3(gx, (ctx->nb+per-1)/per), dim3(PNL_NTHREADS), 0, ctx->stream, P, W,
                           (double*)ctx->b_D.p, cell_begin, cell_end, per);
        if (nbpairs > 0) {
            const unsigned grid = (unsigned)(((long long)nbpairs*64+PNL_NTHREADS-1)/PNL_NTHREADS);
            const int4 *bp = (const int4*)ctx->b_pw_bpairs.p;
            hipLaunchKernelGGL((k_pw_boundary_singular<DIM, DPE, 0>), dim3(grid), dim3(PNL_NTHREADS), 0, ctx->stream, P, W, bp, nbpairs, (double*)ctx->b_D.p, cell_begin, cell_end);
            if (DIM == 2)
                hipLaunchKernelGGL((k_pw_boundary_singular<DIM, DPE, (DIM == 2 ? 1 : 0)>), dim3(grid), dim3(PNL_NTHREADS), 0, ctx->stream, P, W, bp, nbpairs, (double*)ctx->b_D.p, cell_begin, cell_end);
        }
        HIPCHK(ctx, hipGetLastError());
    }
    HIPCHK(ctx, hipEventRecord(ctx->ev[4], ctx->stream));
    {
        const long long nt = (long long)ctx->nc*DPE*DPE;
        hipLaunchKernelGGL((k_scatter_diag<DPE>), dim3((unsigned)((nt+PNL_NTHREADS-1)/PNL_NTHREADS)), dim3(PNL_NTHREADS), 0,
                           ctx->stream, P, (const double*)ctx->b_D.p, A, (long long)ldA);
        HIPCHK(ctx, hipGetLastError());
    }
    HIPCHK(ctx, hipEventRecord(ctx->ev[5], ctx->stream));
    ctx->ev_valid = true;
    return PNL_OK;
}
}  // namespace

// getSparse without a host pair list: block tiles the horizon can reach -> k_fh_pairs -> the sorted pipeline of the masked path
namespace {
template <int DIM, int DPE, int KT>
int horizon_impl(pnl_context *ctx, SparseOut S, int cell_begin, int cell_end) {
    int rc;
    const int T = ctx->tile, nbk = ctx->nblocks;
    const double delta = std::sqrt(ctx->C().kern[0].horizon2);
    const bool whole = cell_begin <= 0 && cell_end >= ctx->nc;
    std::vector<int2> tiles;
    for (int a = 0; a < nbk; a++)
        for (int b = a; b < nbk; b++) {
            // a range of first cells (the reference's cellNo1 split, NA:1280-1285): only block rows that hold one
            if ((a+1)*T <= cell_begin || a*T >= cell_end) continue;
            const auto &A = ctx->blocks[a], &B = ctx->blocks[b];
            const double dx = A.tcx-B.tcx, dy = A.tcy-B.tcy;
            // every vertex of a block lies within trad of (tcx, tcy) (vertices within h of their cell's centre)
            if (std::sqrt(dx*dx+dy*dy)-A.trad-B.trad <= delta) tiles.push_back(make_int2(a, b));
        }
    if ((rc = upload(ctx, ctx->b_tiles, tiles.data(), tiles.size()))) return rc;
    ctx->tiles_cached.clear(); ctx->tiles_cb = -1;            // b_tiles no longer holds the dense tile list
    const size_t per_tile = (size_t)T*T, chunk_tiles = std::max<size_t>(1, (size_t)(48u << 20)/per_tile);
    const size_t cap = std::min(tiles.size(), chunk_tiles)*per_tile;
    if ((rc = ensure(ctx, ctx->b_mp_pairs, cap*sizeof(int2)))) return rc;
    if ((rc = ensure(ctx, ctx->b_mp_wl, cap*sizeof(int4)))) return rc;
    if ((rc = ensure(ctx, ctx->b_wlcount, sizeof(unsigned)*PNL_WL_SLOTS))) return rc;
    ctx->wl_slots = 1; ctx->wl_cap_each = (unsigned)cap;
    S.pairs = (const int*)ctx->b_mp_pairs.p;
    S.masks = nullptr;
    unsigned long long total = 0;
    bool first = true;
    // The pairs inside the horizon are integrated by the tile kernel (LDS sub-block, one pattern search per sub-block entry
    // instead of one per pair and entry); what it cannot do itself -- pairs cut by the horizon, touching pairs, orders
    // without a packed rule -- it hands to the sorted sparse pipeline through the far list.  PNL_FH_NOTILES=1 keeps the
    // pair generator k_fh_pairs, which sends every pair down that pipeline.
    constexpr int TILE = (DPE == 6 || (DIM == 1 && DPE == 3)) ? 32 : 64, ND = DPE*(DPE+1)/2;
    using TS = TileSmem<DIM, DPE, TILE, KT == 0>;
    const int acc_stride = acc_stride_of(ctx->nU, TS::fixed_bytes);
    const size_t lds = TS::fixed_bytes+sizeof(double)*(size_t)(ctx->nU+1)*acc_stride;
    // piecewise-constant order: the candidate pairs of a chunk once (k_fh_pairs), then the sorted pipeline once per order class and
    // orientation, whose classification keeps the pairs of the class (like pnl_assemble_pairs_masked)
    const bool var = ctx->nlab > 0;
    const bool use_tiles = T == TILE && lds <= 160*1024 && !pnl_tune("PNL_FH_NOTILES") && !var;
    if (!use_tiles && !whole) return fail(ctx, PNL_ERR_UNSUPPORTED, "a range of first cells needs the tile route of the finite-horizon assembly");
    ctx->visited_is_assembled = use_tiles;
    for (size_t t0 = 0; t0 < tiles.size(); t0 += chunk_tiles) {
        const int nt = (int)std::min(chunk_tiles, tiles.size()-t0);
        HIPCHK(ctx, hipMemsetAsync(ctx->b_wlcount.p, 0, sizeof(unsigned), ctx->stream));
        if (first) HIPCHK(ctx, hipEventRecord(ctx->ev[0], ctx->stream));
        if (use_tiles) {
            auto kfun = k_tile_distant<DIM, DPE, TILE, KT, false, true>;
            HIPCHK(ctx, hipFuncSetAttribute((const void*)kfun, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
            int per_cu = 2;
            (void)hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, (const void*)kfun, tile_threads(DPE, KT, true), lds);
            if ((rc = ensure(ctx, ctx->b_tilectr, sizeof(unsigned)))) return rc;
            HIPCHK(ctx, hipMemsetAsync(ctx->b_tilectr.p, 0, sizeof(unsigned), ctx->stream));
            HIPCHK(ctx, hipMemsetAsync(ctx->b_D.p, 0, sizeof(double)*(size_t)ctx->ncp*ND, ctx->stream));
            ClusterTiles CT{};
            CT.S = S;
            CT.wl_ds = (int2*)ctx->b_mp_pairs.p;                   // the pairs of the far-list entries
            SlotOut SOk{};
            SOk.nU = ctx->nU;                                      // rows of the LDS sub-block
            const int grid = pnl_grid_cap(std::min(nt, 256*std::max(per_cu, 1)));
            hipLaunchKernelGGL(kfun, dim3(grid), dim3(tile_threads(DPE, KT, true)), lds, ctx->stream, ctx->P, (const int2*)ctx->b_tiles.p+t0,
                               (double*)nullptr, 0ll, (double*)ctx->b_D.p, std::max(cell_begin, 0), std::min(cell_end, ctx->nc), acc_stride, (int4*)ctx->b_mp_wl.p,
                               (unsigned*)ctx->b_wlcount.p, (unsigned)cap, 0, nt, CT, (unsigned*)ctx->b_tilectr.p, SOk);
        } else
            hipLaunchKernelGGL((k_fh_pairs<DIM, DPE>), dim3(nt), dim3(PNL_NTHREADS), 0, ctx->stream, ctx->P, (const int2*)ctx->b_tiles.p+t0, T,
                               (int2*)ctx->b_mp_pairs.p, (int4*)ctx->b_mp_wl.p, (unsigned*)ctx->b_wlcount.p, (unsigned)cap);
        HIPCHK(ctx, hipGetLastError());
        unsigned np = 0;
        HIPCHK(ctx, hipMemcpyAsync(&np, ctx->b_wlcount.p, sizeof(unsigned), hipMemcpyDeviceToHost, ctx->stream));
        HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
        if (np > cap) return fail(ctx, PNL_ERR_STATE, "far list of the finite-horizon tiles overflowed (%u > %zu)", np, cap);
        total += np;
        if (np && var) {
            const int ncls = (int)ctx->cls.size(), norient = ctx->nonsym ? 2 : 1, cur0 = ctx->cur;
            for (int ko = 0; ko < ncls*norient && !rc; ko++) {
                ctx->cur = ko/norient; ctx->orient = ko%norient;
                refresh_tables(ctx);
                rc = pairs_masked_impl<DIM, DPE, 0>(ctx, (int)np, S, true, false, false);
            }
            ctx->cur = cur0; ctx->orient = 0;
            refresh_tables(ctx);
            if (rc) return rc;
        } else
        if (np && (rc = pairs_masked_impl<DIM, DPE, KT>(ctx, (int)np, S, false, false, use_tiles))) return rc;
        if (!np && use_tiles) {
            // no far entries in this chunk: the diagonal blocks of its tiles still have to reach the matrix
            const long long n = (long long)ctx->nc*ND;
            hipLaunchKernelGGL((k_scatter_diag_sparse<DPE>), dim3((unsigned)((n+PNL_NTHREADS-1)/PNL_NTHREADS)), dim3(PNL_NTHREADS), 0,
                               ctx->stream, ctx->P, (const double*)ctx->b_D.p, ctx->nc, S);
            for (int e = 1; e < 8; e++) HIPCHK(ctx, hipEventRecord(ctx->ev[e], ctx->stream));
            ctx->ev_valid = true; ctx->tiles_launched = true;
        }
        first = false;
    }
    ctx->visited_pairs = total;
    if (total == 0) {
        for (int e = 1; e < 8; e++) HIPCHK(ctx, hipEventRecord(ctx->ev[e], ctx->stream));
        ctx->ev_valid = true; ctx->tiles_launched = true;
    }
    HIPCHK(ctx, hipMemsetAsync(ctx->b_wlcount.p, 0, sizeof(unsigned), ctx->stream));     // pnl_get_counters reads it as the dense work-list fill
    return PNL_OK;
}
}  // namespace

// ---- options (pnl_context.h: pnl_tune) --------------------------------------------------------------------------------------
#include <deque>
#include <map>
#include <mutex>
namespace {
std::mutex g_opt_mutex;
// values are interned and never freed: a pointer pnl_tune() handed out stays valid while another thread sets or erases the option
// (the set of distinct values a process names is small)
std::map<std::string, const std::string*> g_options;
std::deque<std::string> g_option_values;
// the options a product build accepts: the hooks through which the parity tests reach the alternative code paths, and the
// diagnostics line
const char *const k_product_options[] = {"PNL_WL_FRAC", "PNL_FH_NOTILES", "PNL_NO_POWTAB", "PNL_VERBOSE", "PNL_PLAN_TIMING", "PNL_PLAN_THREADS", "PNL_BND_OLD",
                                         // profiling: every phase on the caller's stream, one after the other (per-kernel times that add up)
                                         "PNL_NO_OVERLAP", "PNL_NO_FORK",
                                         // tests: at most this many workgroups of a persistent tile kernel (every workgroup then walks
                                         // many tiles at test sizes: the pipelined tile loops against the oracle)
                                         "PNL_TILE_WGS", "PNL_UNI_GENERIC"};
}  // namespace

const char *pnl_tune(const char *name) {
    {
        std::lock_guard<std::mutex> lk(g_opt_mutex);
        auto it = g_options.find(name);
        if (it != g_options.end()) return it->second->c_str();
    }
#ifdef PNL_TUNING
    return getenv(name);
#else
    return nullptr;
#endif
}

extern "C" {

const char *pnl_version(void) {
#ifdef PNL_TUNING
    return "pnl_hip 0.1 (gfx950, tuning build)";
#else
    return "pnl_hip 0.1 (gfx950)";
#endif
}

int pnl_set_option(const char *name, const char *value) {
    if (!name) return PNL_ERR_INVALID;
#ifndef PNL_TUNING
    bool known = false;
    for (const char *k : k_product_options) known = known || std::strcmp(k, name) == 0;
    if (!known) return PNL_ERR_UNSUPPORTED;
#endif
    std::lock_guard<std::mutex> lk(g_opt_mutex);
    if (value) {
        const std::string *v = nullptr;
        for (const std::string &have : g_option_values) if (have == value) { v = &have; break; }
        if (!v) { g_option_values.emplace_back(value); v = &g_option_values.back(); }     // deque: earlier elements do not move
        g_options[name] = v;
    } else g_options.erase(name);
    return PNL_OK;
}

int pnl_create(int device_id, pnl_context **out) {
    if (!out) return PNL_ERR_INVALID;
    *out = nullptr;
    int ndev = 0;
    hipError_t e0;
    auto hiperr = [](const char *what, hipError_t e) { fprintf(stderr, "[pnl] pnl_create: %s failed: %s\n", what, hipGetErrorString(e)); return PNL_ERR_HIP; };
    if ((e0 = hipGetDeviceCount(&ndev)) != hipSuccess || ndev <= 0) return hiperr("hipGetDeviceCount", e0);
    if (device_id < 0 || device_id >= ndev) return PNL_ERR_INVALID;
    if ((e0 = hipSetDevice(device_id)) != hipSuccess) return hiperr("hipSetDevice", e0);
    pnl_context *ctx = new pnl_context();
    ctx->device = device_id;
    if ((e0 = hipStreamCreate(&ctx->own_stream)) != hipSuccess) { delete ctx; return hiperr("hipStreamCreate", e0); }
    ctx->stream = ctx->own_stream;
    for (auto &st : ctx->aux)
        if ((e0 = hipStreamCreateWithFlags(&st, hipStreamNonBlocking)) != hipSuccess) { delete ctx; return hiperr("hipStreamCreateWithFlags", e0); }
    if (hipEventCreateWithFlags(&ctx->ev_fork, hipEventDisableTiming) != hipSuccess) { delete ctx; return PNL_ERR_HIP; }
    if (hipEventCreateWithFlags(&ctx->ev_fold, hipEventDisableTiming) != hipSuccess) { delete ctx; return PNL_ERR_HIP; }
    if (hipEventCreateWithFlags(&ctx->ev_bnd, hipEventDisableTiming) != hipSuccess) { delete ctx; return PNL_ERR_HIP; }
    for (auto &e : ctx->ev_join)
        if (hipEventCreateWithFlags(&e, hipEventDisableTiming) != hipSuccess) { delete ctx; return PNL_ERR_HIP; }
    for (auto &e : ctx->ev)
        if (hipEventCreate(&e) != hipSuccess) { delete ctx; return PNL_ERR_HIP; }
    for (auto &pr : ctx->kev)
        for (auto &e : pr)
            if (hipEventCreate(&e) != hipSuccess) { delete ctx; return PNL_ERR_HIP; }
    std::memset(&ctx->P, 0, sizeof(ctx->P));
#ifdef PNL_DEBUG_ABLATE
    if (const char *e = pnl_tune("PNL_ABLATE")) ctx->ablate = atoi(e);      // result-changing debug switches: debug builds only
#endif
    if (const char *e = pnl_tune("PNL_WL_LANE")) ctx->wl_lane = atoi(e) != 0;
    if (const char *e = pnl_tune("PNL_PURE")) ctx->use_pure = atoi(e) != 0;
    ctx->cls.push_back(new pnl_context::ClassData());
    *out = ctx;
    return PNL_OK;
}

void pnl_destroy(pnl_context *ctx) {
    if (!ctx) return;
    (void)hipSetDevice(ctx->device);
    (void)hipStreamSynchronize(ctx->stream);
    for (auto &e : ctx->ev)
        if (e) (void)hipEventDestroy(e);
    for (auto &pr : ctx->kev)
        for (auto &e : pr)
            if (e) (void)hipEventDestroy(e);
    tile_order_drop(ctx);
    for (auto &st : ctx->aux) if (st) { (void)hipStreamSynchronize(st); (void)hipStreamDestroy(st); }
    if (ctx->ev_fork) (void)hipEventDestroy(ctx->ev_fork);
    if (ctx->ev_fold) (void)hipEventDestroy(ctx->ev_fold);
    if (ctx->ev_bnd) (void)hipEventDestroy(ctx->ev_bnd);
    for (auto *t : ctx->powtabs) delete t;
    ctx->powtabs.clear();
    for (auto &e : ctx->ev_join) if (e) (void)hipEventDestroy(e);
    if (ctx->own_stream) (void)hipStreamDestroy(ctx->own_stream);
    for (auto *c : ctx->cls) delete c;
    delete ctx;
}

const char *pnl_error_string(pnl_context *ctx) { return ctx ? ctx->err.c_str() : "null context"; }

int pnl_set_stream(pnl_context *ctx, void *hip_stream) {
    if (!ctx) return PNL_ERR_INVALID;
    ctx->stream = hip_stream ? (hipStream_t)hip_stream : ctx->own_stream;
    return PNL_OK;
}

// Entries beyond a work list's capacity and pairs whose quadrature order exceeds the uploaded tables are dropped by the
// kernels; the counters that tell (one fill counter per class pass, counter 5) are read here so that the loss fails the
// first call that waits for the assembly instead of going unnoticed.
static int check_overflow(pnl_context *ctx) {
    if (ctx->b_wlcount.p && ctx->wl_cap_each > 0) {
        unsigned wl[PNL_WL_SLOTS];
        const int n = std::max(1, std::min(ctx->wl_slots, PNL_WL_SLOTS));
        HIPCHK(ctx, hipMemcpy(wl, ctx->b_wlcount.p, sizeof(unsigned)*n, hipMemcpyDeviceToHost));
        for (int i = 0; i < n; i++)
            if (wl[i] > ctx->wl_cap_each)
                return fail(ctx, PNL_ERR_STATE, "work list overflow (pass %d): %u entries needed, capacity %u; the assembled operator is "
                            "incomplete", i, wl[i], ctx->wl_cap_each);
    }
    if (ctx->b_counters.p) {
        unsigned long long ov = 0;
        HIPCHK(ctx, hipMemcpy(&ov, (const unsigned long long*)ctx->b_counters.p+5, sizeof(ov), hipMemcpyDeviceToHost));
        if (ov) return fail(ctx, PNL_ERR_ORDER, "%llu pairs need a quadrature order beyond the uploaded tables (qmax=%d); the assembled "
                            "operator is incomplete", ov, ctx->qmax);
        HIPCHK(ctx, hipMemcpy(&ov, (const unsigned long long*)ctx->b_counters.p+7, sizeof(ov), hipMemcpyDeviceToHost));
        if (ov) return fail(ctx, PNL_ERR_STATE, "%llu entries of touching pairs have no row in the slab (pnl_set_row_slab needs the DoFs of the "
                            "cells touching the rank's cells)", ov);
    }
    return PNL_OK;
}

int pnl_synchronize(pnl_context *ctx) {
    if (!ctx) return PNL_ERR_INVALID;
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    return check_overflow(ctx);
}

int pnl_upload_mesh(pnl_context *ctx, int dim, int nv, const double *vertices, int nc, const int32_t *cells,
                    const double *vol, const double *h, double H0) {
    if (!ctx) return PNL_ERR_INVALID;
    if (dim != 1 && dim != 2) return fail(ctx, PNL_ERR_UNSUPPORTED, "dim=%d is not implemented", dim);
    if (nv <= 0 || nc <= 0 || !vertices || !cells || !vol || !h || !(H0 > 0.)) return fail(ctx, PNL_ERR_INVALID, "bad mesh arguments");
    ctx->dim = dim; ctx->nv = nv; ctx->nc = nc; ctx->H0 = H0;
    ctx->vertices.assign(vertices, vertices+(size_t)nv*dim);
    ctx->cells.assign(cells, cells+(size_t)nc*(dim+1));
    ctx->vol.assign(vol, vol+nc);
    ctx->h.assign(h, h+nc);
    ctx->cell_orig.clear();
    ctx->have_mesh = true;
    ctx->dirty = true;
    return PNL_OK;
}

int pnl_set_interaction_transform(pnl_context *ctx, int dim, const double *T) {
    if (!ctx) return PNL_ERR_INVALID;
    if (T && dim != 2) return fail(ctx, PNL_ERR_UNSUPPORTED, "interaction transform: 2D only");
    ctx->have_xform = T != nullptr;
    for (int i = 0; i < 4; i++) ctx->xform[i] = T ? T[i] : (i == 0 || i == 3 ? 1. : 0.);
    if (T && !(std::fabs(T[0]*T[3]-T[1]*T[2]) > 0.)) return fail(ctx, PNL_ERR_INVALID, "interaction transform is singular");
    ctx->dirty = true;
    return PNL_OK;
}

int pnl_set_cell_order(pnl_context *ctx, int nc, const int32_t *orig) {
    if (!ctx) return PNL_ERR_INVALID;
    if (!ctx->have_mesh) return fail(ctx, PNL_ERR_STATE, "upload the mesh first");
    if (orig && nc != ctx->nc) return fail(ctx, PNL_ERR_INVALID, "pnl_set_cell_order: %d cells expected", ctx->nc);
    if (orig) ctx->cell_orig.assign(orig, orig+nc); else ctx->cell_orig.clear();
    ctx->dirty = true;
    return PNL_OK;
}

int pnl_upload_dofmap(pnl_context *ctx, int dpe, int dofs_per_vertex, int dofs_per_edge, int num_dofs, const int32_t *dofs,
                      const int32_t *perm_table) {
    if (!ctx) return PNL_ERR_INVALID;
    if (!ctx->have_mesh) return fail(ctx, PNL_ERR_STATE, "upload the mesh first");
    if (dpe <= 0 || num_dofs <= 0 || !dofs || !perm_table) return fail(ctx, PNL_ERR_INVALID, "bad DoF map arguments");
    // NA:941: the local matrix must fit the mask type (256 bits in the reference)
    if ((2*dpe)*(2*dpe+1)/2 > 256) return fail(ctx, PNL_ERR_INVALID, "local matrix has more than 256 entries");
    ctx->dpe = dpe; ctx->dpv = dofs_per_vertex; ctx->dped = dofs_per_edge; ctx->N = num_dofs;
    ctx->dofs.assign(dofs, dofs+(size_t)ctx->nc*dpe);
    int nperm = 1;
    for (int k = 2; k <= ctx->dim+1; k++) nperm *= k;
    ctx->perm_table.assign(perm_table, perm_table+(size_t)nperm*dpe);
    ctx->have_dofs = true;
    ctx->dirty = true;
    return PNL_OK;
}

int pnl_set_classes(pnl_context *ctx, int nclasses, int num_labels, const int32_t *cell_labels, const int32_t *facet_labels,
                    const int32_t *cls_of) {
    if (!ctx) return PNL_ERR_INVALID;
    if (!ctx->have_mesh) return fail(ctx, PNL_ERR_STATE, "upload the mesh first");
    if (nclasses < 1 || nclasses > 64 || num_labels < 0 || (num_labels > 0 && (!cell_labels || !cls_of)))
        return fail(ctx, PNL_ERR_INVALID, "bad class arguments");
    for (int i = 0; i < num_labels*num_labels; i++)
        if (cls_of[i] < 0 || cls_of[i] >= nclasses) return fail(ctx, PNL_ERR_INVALID, "class table entry %d out of range", i);
    for (int c = 0; num_labels > 0 && c < ctx->nc; c++)
        if (cell_labels[c] < 0 || cell_labels[c] >= num_labels) return fail(ctx, PNL_ERR_INVALID, "label of cell %d out of range", c);
    for (auto *c : ctx->cls) delete c;
    ctx->cls.clear();
    for (int k = 0; k < nclasses; k++) ctx->cls.push_back(new pnl_context::ClassData());
    ctx->cur = 0;
    ctx->nlab = num_labels;
    ctx->cell_labels.assign(cell_labels, cell_labels+(num_labels > 0 ? ctx->nc : 0));
    ctx->cls_of.assign(cls_of, cls_of+(size_t)num_labels*num_labels);
    ctx->tiles_cached.clear(); ctx->tiles_forms.clear();
    ctx->nonsym = false;
    ctx->facet_labels.clear();
    if (num_labels > 0 && facet_labels && ctx->have_boundary) {
        for (int f = 0; f < ctx->nb; f++)
            if (facet_labels[f] < 0 || facet_labels[f] >= num_labels) return fail(ctx, PNL_ERR_INVALID, "label of facet %d out of range", f);
        ctx->facet_labels.assign(facet_labels, facet_labels+ctx->nb);
    }
    ctx->dirty = true;
    return PNL_OK;
}

int pnl_set_nonsymmetric(pnl_context *ctx, int on) {
    if (!ctx) return PNL_ERR_INVALID;
    if (on && ctx->nlab == 0) return fail(ctx, PNL_ERR_STATE, "pnl_set_classes with labels first: only a label table can be non-symmetric");
    ctx->nonsym = on != 0;
    ctx->dirty = true;
    ctx->tiles_cached.clear(); ctx->tiles_forms.clear();
    return PNL_OK;
}

int pnl_select_class(pnl_context *ctx, int k) {
    if (!ctx) return PNL_ERR_INVALID;
    if (k < 0 || k >= (int)ctx->cls.size()) return fail(ctx, PNL_ERR_INVALID, "class %d out of range", k);
    ctx->cur = k;
    return PNL_OK;
}

int pnl_set_kernel(pnl_context *ctx, int which, const pnl_kernel *k) {
    if (!ctx || !k || which < 0 || which > 1) return fail(ctx, PNL_ERR_INVALID, "bad kernel arguments");
    if (k->ktype < 0 || k->ktype > PNL_EXPONENTIAL_BOUNDARY) return fail(ctx, PNL_ERR_UNSUPPORTED, "kernel type %d is not implemented", k->ktype);
    if (!std::isinf(k->horizon2) && (k->interaction < 1 || k->interaction > 2 || !(k->horizon2 > 0.)))
        return fail(ctx, PNL_ERR_UNSUPPORTED, "finite horizon: interaction %d is not implemented (1 ball2_retriangulation, 2 ball2_barycenter)",
                    k->interaction);
    ctx->C().kern[which] = *k;
    ctx->C().have_kernel[which] = true;
    return PNL_OK;
}

int pnl_set_order_formula(pnl_context *ctx, int which, const pnl_order_formula *f) {
    if (!ctx || !f || which < 0 || which > 1) return fail(ctx, PNL_ERR_INVALID, "bad order-formula arguments");
    ctx->C().form[which] = *f;
    ctx->C().have_form[which] = true;
    return PNL_OK;
}

int pnl_upload_distant_rules(pnl_context *ctx, int qmax, const int32_t *off, const double *bary, const double *w,
                             const double *phi, const int32_t *foff, const double *fbary, const double *fw) {
    if (!ctx) return PNL_ERR_INVALID;
    if (!ctx->have_dofs) return fail(ctx, PNL_ERR_STATE, "upload the DoF map first");
    if (qmax < 2 || qmax > PNL_MAXQ || !off || !bary || !w || !phi || !foff || !fbary || !fw)
        return fail(ctx, PNL_ERR_INVALID, "bad rule arguments (2 <= qmax <= %d)", PNL_MAXQ);
    const int total = off[qmax+1], ftotal = foff[qmax+1];
    int rc;
    ctx->rule_off.assign(off, off+qmax+2); ctx->frule_off.assign(foff, foff+qmax+2);
    if ((rc = upload(ctx, ctx->b_off, off, (size_t)qmax+2))) return rc;
    if ((rc = upload(ctx, ctx->b_bary, bary, (size_t)total*3))) return rc;
    if ((rc = upload(ctx, ctx->b_w, w, (size_t)total))) return rc;
    if ((rc = upload(ctx, ctx->b_phi, phi, (size_t)total*ctx->dpe))) return rc;
    if ((rc = upload(ctx, ctx->b_foff, foff, (size_t)qmax+2))) return rc;
    if ((rc = upload(ctx, ctx->b_fbary, fbary, (size_t)ftotal*2))) return rc;
    if ((rc = upload(ctx, ctx->b_fw, fw, (size_t)ftotal))) return rc;
    // pack the orders with 2, 3, 4, 6 or 7 points (the unrolled lane-per-pair variants) for the tile kernel's LDS copy
    {
        const int dpe = ctx->dpe, st = 4+dpe;
        std::vector<int32_t> tn(PNL_MAXQ+2, 0), to(PNL_MAXQ+2, 0);
        std::vector<double> tab, wphi;
        int npts = 0, nb = 0;
        // the tile kernel unrolls exactly two point counts (3 and 6 on triangles, 2 and 3 on intervals) and integrates the
        // other orders with at most PNL_GEN_MAXPTS points through a generic loop (list C)
        const int nA = ctx->dim == 2 ? 3 : 2, nB = ctx->dim == 2 ? 6 : 3;
        for (int q = 2; q <= qmax && q < 18; q++) {
            const int n = off[q+1]-off[q];
            const bool ok = (n == nA || n == nB || (n > 0 && n <= PNL_GEN_MAXPTS));
            const int npad = n;
            if (!ok || npts+npad > PNL_TT_MAXPTS) continue;
            tn[q] = n; to[q] = npts;
            for (int i = 0; i < npad; i++) {
                const size_t p = (size_t)off[q]+(i < n ? i : 0);
                const double wp = i < n ? w[p] : 0.;
                tab.push_back(bary[3*p]); tab.push_back(bary[3*p+1]); tab.push_back(bary[3*p+2]); tab.push_back(wp);
                for (int a = 0; a < dpe; a++) tab.push_back(phi[p*dpe+a]);
                wphi.push_back(wp);
                for (int a = 0; a+1 < dpe; a++) wphi.push_back(wp*phi[p*dpe+a]);
            }
            npts += npad; nb++;
            (void)st;
        }
        if ((rc = upload(ctx, ctx->b_ttn, tn.data(), tn.size()))) return rc;
        if ((rc = upload(ctx, ctx->b_ttoff, to.data(), to.size()))) return rc;
        if ((rc = upload(ctx, ctx->b_tttab, tab.data(), tab.size()))) return rc;
        if ((rc = upload(ctx, ctx->b_ttwphi, wphi.data(), wphi.size()))) return rc;
        ctx->P.tt_npts = npts;
        // second-generation tile kernels: w, w phi[0..dpe-1] of the packed points (point order of tab)
        std::vector<double> wphif;
        for (int q = 2; q <= qmax && q < 18; q++)
            for (int i = 0; i < tn[q]; i++) {
                const size_t p = (size_t)off[q]+(i < tn[q] ? i : 0);
                const double wp = i < tn[q] ? w[p] : 0.;
                wphif.push_back(wp);
                for (int a = 0; a < dpe; a++) wphif.push_back(wp*phi[p*dpe+a]);
            }
        if ((rc = upload(ctx, ctx->b_ttwphif, wphif.data(), wphif.size()))) return rc;
        // rule blocks of the uniform-order tiles (2D, orders 2-4 with 3 or 6 points): bary[n][3], w[n], w phi[n][dpe],
        // w phi_a phi_b[nd][n] (a <= b, row-major upper triangle)
        std::vector<double> uni;
        for (int q = 0; q < 5; q++) { ctx->uni_off[q] = -1; ctx->uni_np[q] = 0; }
        for (int q = 2; q <= 4 && q <= qmax && ctx->dim == 2; q++) {
            const int n = off[q+1]-off[q];
            if (n != 3 && n != 6) continue;
            ctx->uni_off[q] = (int)uni.size(); ctx->uni_np[q] = n;
            // point order of the block: P1 with three points of equal weight whose shape values are A + B delta(b, sigma(i)) for a
            // permutation sigma (the symmetric degree-2 rule) -> the points in the order sigma^-1, so that w phi_b(y_i) = A + B delta_bi
            std::vector<int> ord(n);
            for (int i = 0; i < n; i++) ord[i] = i;
            ctx->uni_struct[q] = false;
            if (n == 3 && dpe == 3) {
                int sig[3] = {-1, -1, -1};
                bool ok = w[off[q]] == w[off[q]+1] && w[off[q]] == w[off[q]+2];
                for (int i = 0; i < 3 && ok; i++) {
                    const double *ph = &phi[((size_t)off[q]+i)*dpe];
                    int big = 0;
                    for (int b = 1; b < 3; b++) if (ph[b] > ph[big]) big = b;
                    sig[i] = big;
                    for (int b = 0; b < 3; b++)
                        ok = ok && ph[b] == (b == big ? phi[(size_t)off[q]*dpe+sig[0]] : phi[(size_t)off[q]*dpe+(sig[0]+1)%3]);
                }
                ok = ok && sig[0] != sig[1] && sig[0] != sig[2] && sig[1] != sig[2];
                if (ok) {
                    for (int i = 0; i < 3; i++) ord[sig[i]] = i;
                    ctx->uni_struct[q] = true;
                }
            }
            // six points in two orbits of three (the symmetric 6-point rules of degree 3 / 4): per orbit equal weights and shape values
            // A_o + B_o delta(b, sigma(i)) -> the points in the order orbit 0 (positions 0, 1, 2), orbit 1 (positions 0, 1, 2)
            if (n == 6 && dpe == 3) {
                int big[6], orb[6], norb = 0;
                double ow[2] = {0., 0.}, ohi[2] = {0., 0.}, olo[2] = {0., 0.};
                bool ok = true;
                for (int i = 0; i < 6 && ok; i++) {
                    const double *ph = &phi[((size_t)off[q]+i)*dpe];
                    // the distinguished coordinate of a point (x, x, 1 - 2 x): the one that differs from the other two
                    int d = -1;
                    if (ph[0] == ph[1] && ph[0] != ph[2]) d = 2;
                    else if (ph[0] == ph[2] && ph[0] != ph[1]) d = 1;
                    else if (ph[1] == ph[2] && ph[0] != ph[1]) d = 0;
                    if (d < 0) { ok = false; break; }
                    big[i] = d;
                    const double hi = ph[d], lo = ph[(d+1)%3], wi = w[off[q]+i];
                    int o = -1;
                    for (int t = 0; t < norb; t++) if (ow[t] == wi && ohi[t] == hi && olo[t] == lo) o = t;
                    if (o < 0) { if (norb == 2) { ok = false; break; } o = norb++; ow[o] = wi; ohi[o] = hi; olo[o] = lo; }
                    orb[i] = o;
                }
                if (ok && norb == 2) {
                    int seen[2][3] = {{0, 0, 0}, {0, 0, 0}};
                    for (int i = 0; i < 6; i++) seen[orb[i]][big[i]]++;
                    for (int o = 0; o < 2; o++) for (int d = 0; d < 3; d++) ok = ok && seen[o][d] == 1;
                    if (ok) {
                        for (int i = 0; i < 6; i++) ord[3*orb[i]+big[i]] = i;
                        ctx->uni_struct[q] = true;
                    }
                }
            }
            auto pt = [&](int i) { return (size_t)off[q]+ord[i]; };
            for (int i = 0; i < n; i++) for (int k2 = 0; k2 < 3; k2++) uni.push_back(bary[3*pt(i)+k2]);
            for (int i = 0; i < n; i++) uni.push_back(w[pt(i)]);
            for (int i = 0; i < n; i++) for (int a = 0; a < dpe; a++) uni.push_back(w[pt(i)]*phi[pt(i)*dpe+a]);
            for (int a = 0; a < dpe; a++)
                for (int b = a; b < dpe; b++)
                    for (int i = 0; i < n; i++) uni.push_back(w[pt(i)]*phi[pt(i)*dpe+a]*phi[pt(i)*dpe+b]);
        }
        if ((rc = upload(ctx, ctx->b_uni, uni.data(), uni.size()))) return rc;
        ctx->tiles_cached.clear(); ctx->tiles_forms.clear();
    }
    ctx->qmax = qmax;
    ctx->have_rules = true;
    return PNL_OK;
}

int pnl_upload_singular_rule(pnl_context *ctx, int which, int panel, int M, int rows, const double *nodes, const double *w,
                             const double *psi, double facv) {
    if (!ctx) return PNL_ERR_INVALID;
    if (!ctx->have_dofs) return fail(ctx, PNL_ERR_STATE, "upload the DoF map first");
    const int slot = -panel-1;
    const int nslots = which == PNL_INTERIOR ? ctx->dim+1 : ctx->dim;
    if (which < 0 || which > 1 || slot < 0 || slot >= nslots || M <= 0 || rows <= 0 || !nodes || !w || !psi)
        return fail(ctx, PNL_ERR_INVALID, "bad singular-rule arguments");
    const int dim = ctx->dim, dpe = ctx->dpe, dpv = ctx->dpv, dped = ctx->dped, nV = dim+1;
    int rc;
    if (which == PNL_INTERIOR) {
        const int common = slot+1;
        const int expect = common == nV ? dpe : (common == 1 ? 2*dpe-dpv : 2*dpe-2*dpv-dped);
        if (rows != expect) return fail(ctx, PNL_ERR_INVALID, "singular rule has %d rows, expected %d", rows, expect);
        if ((rc = upload(ctx, ctx->C().b_sn[slot], nodes, (size_t)2*nV*M))) return rc;
        if ((rc = upload(ctx, ctx->C().b_sw[slot], w, (size_t)M))) return rc;
        if ((rc = upload(ctx, ctx->C().b_sp[slot], psi, (size_t)rows*M))) return rc;
        ctx->C().sM[slot] = M; ctx->C().sRows[slot] = rows; ctx->C().sFac = facv;
    } else {
        if (rows != dpe) return fail(ctx, PNL_ERR_INVALID, "boundary singular rule has %d rows, expected %d", rows, dpe);
        if ((rc = upload(ctx, ctx->C().b_bn[slot], nodes, (size_t)(nV+dim)*M))) return rc;
        if ((rc = upload(ctx, ctx->C().b_bw[slot], w, (size_t)M))) return rc;
        if ((rc = upload(ctx, ctx->C().b_bp[slot], psi, (size_t)rows*M))) return rc;
        ctx->C().bM[slot] = M; ctx->C().bFac = facv;
    }
    ctx->C().have_sing[which][slot] = true;
    return PNL_OK;
}

int pnl_upload_boundary(pnl_context *ctx, int nb, const int32_t *bcells) {
    if (!ctx) return PNL_ERR_INVALID;
    if (!ctx->have_mesh) return fail(ctx, PNL_ERR_STATE, "upload the mesh first");
    if (nb < 0 || (nb > 0 && !bcells)) return fail(ctx, PNL_ERR_INVALID, "bad boundary arguments");
    ctx->nb = nb;
    ctx->bcells.assign(bcells, bcells+(size_t)nb*ctx->dim);
    ctx->have_boundary = true;
    ctx->dirty = true;
    return PNL_OK;
}

int pnl_tile_cells(pnl_context *ctx) {
    if (!ctx) return PNL_ERR_INVALID;
    int rc = finalize(ctx);
    return rc ? rc : ctx->tile;
}

static int upload_tiles(pnl_context *ctx, std::vector<int2> &tiles, int cell_begin, int cell_end) {
    // repeated assemblies of the same work list keep it resident
    const int ncls = (int)ctx->cls.size();
    std::vector<pnl_order_formula> forms(ncls);
    for (int k = 0; k < ncls; k++) forms[k] = ctx->cls[k]->form[0];
    if (tiles.size() == ctx->tiles_cached.size() && ctx->b_tiles.p && ctx->tiles_cb == cell_begin && ctx->tiles_ce == cell_end &&
        forms.size() == ctx->tiles_forms.size() && std::memcmp(forms.data(), ctx->tiles_forms.data(), sizeof(pnl_order_formula)*ncls) == 0 &&
        ctx->tiles_filter == ctx->tile_cell_filter &&
        (tiles.empty() || std::memcmp(tiles.data(), ctx->tiles_cached.data(), tiles.size()*sizeof(int2)) == 0))
        return PNL_OK;
    const int T = ctx->tile;
    const bool filter = ctx->tile_cell_filter;
    // uniform tiles: order 2 through k_tile_pure (P1 in 1D and 2D), orders 2-4 through k_tile_uniform (2D: P1 orders 3 and 4, P2)
    const bool p1 = T == 64 && (ctx->dpe == 3 || ctx->dpe == 2), p2 = ctx->dim == 2 && ctx->dpe == 6;
    const bool allow = ctx->use_pure && (p1 || p2) && ctx->qmax >= 2 && !ctx->nonsym;
    int qlimit = 2;
    if (ctx->dim == 2) for (int q = 3; q <= 4 && ctx->uni_off[q] >= 0 && ctx->uni_np[q] == 6 && q <= ctx->qmax; q++) qlimit = q;
    const bool q2ok = p1 ? true : (ctx->uni_off[2] >= 0 && ctx->uni_np[2] == 3);
    if (pnl_tune("PNL_UNI_QMAX")) qlimit = std::min(qlimit, std::max(2, atoi(pnl_tune("PNL_UNI_QMAX"))));
    // variable order: a class only visits the tiles whose blocks hold a label pair of that class (most blocks carry one
    // label, so the K passes together classify every tile about once instead of K times)
    const int L = ctx->nlab;
    std::vector<std::vector<int>> blk_labels;
    if (L > 0) {
        blk_labels.resize(ctx->nblocks);
        for (int b = 0; b < ctx->nblocks; b++) {
            auto &v = blk_labels[b];
            for (int c = b*T; c < std::min((b+1)*T, ctx->nc); c++) v.push_back(ctx->cell_labels[c]);
            std::sort(v.begin(), v.end());
            v.erase(std::unique(v.begin(), v.end()), v.end());
        }
    }
    ctx->cls_tile_off.assign(ncls, 0); ctx->cls_n_mixed.assign(ncls, 0); ctx->cls_n_pure.assign(ncls, 0);
    for (int u = 0; u < 3; u++) ctx->cls_n_uni[u].assign(ncls, 0);
    std::vector<std::vector<int2>> mixed(ncls), uni[3];
    for (int u = 0; u < 3; u++) uni[u].resize(ncls);
    // the order bounds of the tiles on a few host threads (4.7 million tiles at 97,537 DoFs), the lists in tile order afterwards
    std::vector<signed char> qof(tiles.size());
    for (int k = 0; k < ncls; k++) {
        const int nthr = (int)std::max(1u, std::min(8u, std::thread::hardware_concurrency()));
        std::vector<std::thread> pool;
        for (int th = 0; th < nthr; th++) pool.emplace_back([&, th]() {
            const size_t i0 = tiles.size()*th/nthr, i1 = tiles.size()*(th+1)/nthr;
            for (size_t i = i0; i < i1; i++) {
                const int2 &t = tiles[i];
                bool single = true;
                if (L > 0) {
                    bool has = false;
                    for (int la : blk_labels[t.x]) for (int lb : blk_labels[t.y])
                        has = has || ctx->cls_of[(size_t)la*L+lb] == k || ctx->cls_of[(size_t)lb*L+la] == k;
                    if (!has) { qof[i] = -1; continue; }
                    single = blk_labels[t.x].size() == 1 && blk_labels[t.y].size() == 1;
                }
                int q = (allow && single) ? tile_uniform_order(ctx, forms[k], t.x, t.y, qlimit) : 0;
                if (q == 2 && !q2ok) q = 0;
                // the cell range of the MPI-style split applies to the a-cells: only blocks entirely inside qualify
                if (q && filter && !(t.x*T >= cell_begin && (t.x+1)*T <= cell_end)) q = 0;
                qof[i] = (signed char)q;
            }
        });
        for (auto &th : pool) th.join();
        // tiles the bounds left open: the exact order range of their cell pairs, on the device (2D; variable order: tiles whose
        // two blocks carry one label each -- their pairs all belong to this class and see its order formula)
        if (allow && ctx->dim == 2 && (T == 64 || T == 32) && !pnl_tune("PNL_NO_EXACT_TILES")) {
            std::vector<int2> cand;
            std::vector<size_t> cand_idx;
            for (size_t i = 0; i < tiles.size(); i++) {
                const int2 &t = tiles[i];
                if (qof[i] != 0 || t.x == t.y || !ctx->blocks[t.x].full || !ctx->blocks[t.y].full) continue;
                if (L > 0 && !(blk_labels[t.x].size() == 1 && blk_labels[t.y].size() == 1)) continue;
                if (filter && !(t.x*T >= cell_begin && (t.x+1)*T <= cell_end)) continue;
                cand.push_back(t); cand_idx.push_back(i);
            }
            if (!cand.empty()) {
                int rc;
                if ((rc = upload(ctx, ctx->b_candtiles, cand.data(), cand.size()))) return rc;
                if ((rc = ensure(ctx, ctx->b_candq, cand.size()))) return rc;
                const int grid = (int)std::min<size_t>(cand.size(), 256*8);
                const DevFormula qo = to_dev(forms[k]);
                if (T == 64)
                    hipLaunchKernelGGL(k_tile_order_range<64>, dim3(grid), dim3(256), 0, ctx->stream, ctx->P, qo, (const int2*)ctx->b_candtiles.p,
                                       (int)cand.size(), (signed char*)ctx->b_candq.p);
                else
                    hipLaunchKernelGGL(k_tile_order_range<32>, dim3(grid), dim3(256), 0, ctx->stream, ctx->P, qo, (const int2*)ctx->b_candtiles.p,
                                       (int)cand.size(), (signed char*)ctx->b_candq.p);
                HIPCHK(ctx, hipGetLastError());
                std::vector<signed char> cq(cand.size());
                HIPCHK(ctx, hipMemcpyAsync(cq.data(), ctx->b_candq.p, cand.size(), hipMemcpyDeviceToHost, ctx->stream));
                HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
                size_t moved = 0;
                for (size_t c = 0; c < cand.size(); c++) {
                    const int q = cq[c];
                    if (q < 2 || q > qlimit || (q == 2 && !q2ok)) continue;
                    qof[cand_idx[c]] = (signed char)q; moved++;
                }
                if (pnl_tune("PNL_VERBOSE")) fprintf(stderr, "[pnl] exact order range: %zu of %zu open tiles are uniform\n", moved, cand.size());
            }
        }
        for (size_t i = 0; i < tiles.size(); i++) {
            const int q = qof[i];
            if (q < 0) continue;
            (q ? uni[q-2][k] : mixed[k]).push_back(tiles[i]);
        }
    }
    std::vector<int2> all;
    std::vector<int32_t> allcls;
    ctx->single_launch = p2;
    if (!p2) {
        // per class: [mixed][order 2][order 3][order 4]
        for (int k = 0; k < ncls; k++) {
            ctx->cls_tile_off[k] = (int)all.size(); ctx->cls_n_mixed[k] = (int)mixed[k].size();
            all.insert(all.end(), mixed[k].begin(), mixed[k].end());
            for (int u = 0; u < 3; u++) { ctx->cls_n_uni[u][k] = (int)uni[u][k].size(); all.insert(all.end(), uni[u][k].begin(), uni[u][k].end()); }
            ctx->cls_n_pure[k] = ctx->cls_n_uni[0][k];
        }
    } else {
        // one launch over all classes: [mixed tiles of all classes][order 2][order 3][order 4]; class word = 2 class + orientation
        // (a non-symmetric order table visits every mixed tile once per orientation)
        const int norient = ctx->nonsym ? 2 : 1;
        ctx->sl_off[0] = 0;
        // symmetric order tables: a tile that holds pairs of several classes gets ONE entry that names the set of them (bit 29
        // + class bits); k_tile_p2 works through the classes inside one visit and flushes once with plain stores
        const bool one_visit = norient == 1 && ncls > 1 && ncls <= 28 && !pnl_tune("PNL_P2_VISIT_PER_CLASS");
        std::vector<unsigned> tile_mask;
        if (one_visit) {
            tile_mask.assign((size_t)ctx->nblocks*ctx->nblocks, 0u);
            for (int k = 0; k < ncls; k++)
                for (const int2 &t : mixed[k]) tile_mask[(size_t)t.x*ctx->nblocks+t.y] |= 1u << k;
            // the multi-class tiles first (several classifications each: the heavy ones), in tile-list order
            for (const int2 &t : tiles) {
                const unsigned m = tile_mask[(size_t)t.x*ctx->nblocks+t.y];
                if (m & (m-1u)) { all.push_back(t); allcls.push_back((int)((1u << 29) | m)); }
            }
        }
        for (int k = 0; k < ncls; k++) {
            ctx->cls_n_mixed[k] = (int)mixed[k].size();
            for (int o = 0; o < norient; o++)
                for (const int2 &t : mixed[k]) {
                    if (one_visit) { const unsigned m = tile_mask[(size_t)t.x*ctx->nblocks+t.y]; if (m & (m-1u)) continue; }
                    all.push_back(t); allcls.push_back(2*k+o);
                }
        }
        ctx->sl_n[0] = (int)all.size();
        for (int u = 0; u < 3; u++) {
            ctx->sl_off[u+1] = (int)all.size();
            for (int k = 0; k < ncls; k++) {
                ctx->cls_n_uni[u][k] = (int)uni[u][k].size();
                for (const int2 &t : uni[u][k]) { all.push_back(t); allcls.push_back(2*k); }
            }
            ctx->sl_n[u+1] = (int)all.size()-ctx->sl_off[u+1];
        }
        for (int k = 0; k < ncls; k++) ctx->cls_n_pure[k] = ctx->cls_n_uni[0][k];
        // tiles that are visited more than once (several classes, both orientations) add into the block-slot storage
        // (bit 30 of the class word) and are zeroed before
        {
            std::vector<long long> keys(all.size());
            for (size_t i = 0; i < all.size(); i++) keys[i] = (long long)all[i].x*ctx->nblocks+all[i].y;
            std::vector<long long> sorted(keys);
            std::sort(sorted.begin(), sorted.end());
            std::vector<int2> multi;
            for (size_t i = 0; i+1 < sorted.size(); i++)
                if (sorted[i] == sorted[i+1] && (i == 0 || sorted[i-1] != sorted[i]))
                    multi.push_back(make_int2((int)(sorted[i]/ctx->nblocks), (int)(sorted[i]%ctx->nblocks)));
            std::vector<long long> mk(multi.size());
            for (size_t i = 0; i < multi.size(); i++) mk[i] = (long long)multi[i].x*ctx->nblocks+multi[i].y;
            for (size_t i = 0; i < all.size(); i++)
                if (std::binary_search(mk.begin(), mk.end(), keys[i])) allcls[i] |= (1 << 30);
            ctx->n_multitiles = (int)multi.size();
            int rc3 = upload(ctx, ctx->b_multitiles, multi.data(), multi.size());
            if (rc3) return rc3;
        }
        int rc2 = upload(ctx, ctx->b_tilecls, allcls.data(), allcls.size());
        if (rc2) return rc2;
    }
    if (pnl_tune("PNL_VERBOSE")) {
        size_t nm = 0, nu[3] = {0, 0, 0};
        for (int k = 0; k < ncls; k++) { nm += mixed[k].size(); for (int u = 0; u < 3; u++) nu[u] += uni[u][k].size(); }
        fprintf(stderr, "[pnl] tiles: %zu mixed, uniform order 2/3/4: %zu / %zu / %zu (qlimit %d)\n", nm, nu[0], nu[1], nu[2], qlimit);
    }
    int rc = upload(ctx, ctx->b_tiles, all.data(), all.size());
    if (rc) return rc;
    ctx->tile_off = 0; ctx->n_mixed = ctx->cls_n_mixed[0]; ctx->n_pure = ctx->cls_n_pure[0];
    ctx->tiles_cached = tiles; ctx->tiles_cb = cell_begin; ctx->tiles_ce = cell_end; ctx->tiles_forms = forms;
    ctx->tiles_filter = ctx->tile_cell_filter;
    return PNL_OK;
}

static int make_tiles(pnl_context *ctx, std::vector<int2> &tiles, int cell_begin, int cell_end) {
    const int T = ctx->tile, nbk = ctx->nblocks;
    const int a0 = cell_begin/T, a1 = (cell_end+T-1)/T;
    // heavy (near-diagonal) tiles first
    for (int d = 0; d < nbk; d++)
        for (int a = a0; a < a1 && a+d < nbk; a++) tiles.push_back(make_int2(a, a+d));
    return PNL_OK;
}

// A row slab (pnl_set_row_slab) is written one-sided: no mirror pass, no scatter of the per-cell diagonal blocks (they stay
// in the per-cell buffer, pnl_get_diag_blocks), columns are counted from col0.  The rows of every cell of the caller's cell
// range must be in the slab.
static int slab_prepare(pnl_context *ctx, double *&A, int &flags, int cell_begin, int cell_end) {
    if (ctx->slab_rows <= 0) return PNL_OK;
    if (flags & PNL_FLAG_SYMMETRIC_FLUSH) return fail(ctx, PNL_ERR_INVALID, "a row slab is one-sided: PNL_FLAG_SYMMETRIC_FLUSH does not apply");
    if (ctx->have_pw) return fail(ctx, PNL_ERR_UNSUPPORTED, "row slabs are not implemented for kernels with an order per quadrature point");
    const auto &rd = ctx->slab_rowdofs, &cd = ctx->slab_coldofs;
    for (int c = cell_begin; c < ctx->nc; c++)
        for (int k = 0; k < ctx->dpe; k++) {
            const int g = ctx->dofs[(size_t)c*ctx->dpe+k];
            if (g < 0) continue;
            if (c < cell_end && !std::binary_search(rd.begin(), rd.end(), g))
                return fail(ctx, PNL_ERR_INVALID, "DoF %d of cell %d is not a row of the slab", g, c);
            if (!std::binary_search(cd.begin(), cd.end(), g))
                return fail(ctx, PNL_ERR_INVALID, "DoF %d of cell %d is not a column of the slab", g, c);
        }
    flags |= PNL_FLAG_NO_MIRROR;
    (void)A;
    return PNL_OK;
}

int pnl_assemble_dense(pnl_context *ctx, double *A, int64_t ldA, int zero_exterior, int cell_begin, int cell_end, int flags) {
    if (!ctx) return PNL_ERR_INVALID;
    int rc;
    if ((rc = check_ready(ctx))) return rc;
    if ((rc = finalize(ctx))) return rc;
    if ((rc = tile_order_ready(ctx))) return rc;
    if (!A || ldA < (ctx->slab_rows ? ctx->slab_cols : ctx->N)) return fail(ctx, PNL_ERR_INVALID, "bad output matrix (ldA=%lld, num_dofs=%d)", (long long)ldA, ctx->N);
    if (cell_begin < 0 || cell_end > ctx->nc || cell_begin > cell_end) return fail(ctx, PNL_ERR_INVALID, "bad cell range");
    if ((rc = slab_prepare(ctx, A, flags, cell_begin, cell_end))) return rc;
    std::vector<int2> tiles;
    if (cell_end > cell_begin) make_tiles(ctx, tiles, cell_begin, cell_end);
    if ((rc = upload_tiles(ctx, tiles, cell_begin, cell_end))) return rc;
    // pairs visited by the reference loop: c1 in [begin,end), c2 in [c1, nc)
    unsigned long long visited = 0;
    for (long long c = cell_begin; c < cell_end; c++)
        if (ctx->real_from[c] != ctx->real_from[c+1]) visited += (unsigned long long)ctx->real_from[c];      // zero-volume padding cells do not count
    ctx->visited_pairs = visited; ctx->visited_is_assembled = false;
    if (pnl_tune("PNL_FORCE_SYMFLUSH")) flags |= PNL_FLAG_SYMMETRIC_FLUSH;     // debug: both sides written by the flush, no mirror pass
    ctx->slot_full_list = true;
    return dispatch(ctx, A, ldA, zero_exterior, (int)tiles.size(), cell_begin, cell_end, flags);
}

// Estimated cost of every block row of the upper block triangle, in units of one uniform order-2 tile: what a rank that owns
// the row spends on its tiles (classified like upload_tiles does, weights from the measured time per tile of the kernels:
// profiles/r02b_*) plus the per-cell work of its cells (touching pairs, boundary term).
int pnl_block_row_costs(pnl_context *ctx, double *out, int n) {
    if (!ctx || !out) return PNL_ERR_INVALID;
    int rc;
    if ((rc = check_ready(ctx))) return rc;
    if ((rc = finalize(ctx))) return rc;
    if (n != ctx->nblocks) return fail(ctx, PNL_ERR_INVALID, "pnl_block_row_costs: %d blocks expected", ctx->nblocks);
    const int T = ctx->tile, nb = ctx->nblocks;
    const bool p1 = T == 64 && (ctx->dpe == 3 || ctx->dpe == 2), p2 = ctx->dim == 2 && ctx->dpe == 6;
    const bool allow = ctx->use_pure && (p1 || p2) && ctx->qmax >= 2 && !ctx->nonsym && ctx->cls.size() == 1;
    int qlimit = 2;
    if (ctx->dim == 2) for (int q = 3; q <= 4 && ctx->uni_off[q] >= 0 && ctx->uni_np[q] == 6 && q <= ctx->qmax; q++) qlimit = q;
    const pnl_order_formula F = ctx->cls[0]->form[0];
    // ns per tile at 98,304 cells (P1: 51 / 138 / 204 incl. its work-list pairs) and 24,576 cells (P2: 50 / 87 / 125)
    const double w_uni3 = p2 ? 1.75 : 2.7, w_mixed = p2 ? 2.5 : 4.0, w_cells = p2 ? 30. : 57.;
    const int nthreads = (int)std::max(1u, std::min(8u, std::thread::hardware_concurrency()));
    std::vector<std::thread> pool;
    for (int t = 0; t < nthreads; t++)
        pool.emplace_back([&, t]() {
            for (int a = t; a < nb; a += nthreads) {
                double c = w_cells;
                for (int b = a; b < nb; b++) {
                    const int q = allow ? tile_uniform_order(ctx, F, a, b, qlimit) : 0;
                    c += q == 2 ? 1. : (q ? w_uni3 : w_mixed);
                }
                out[a] = c;
            }
        });
    for (auto &th : pool) th.join();
    return PNL_OK;
}

int pnl_dense_overwrites(pnl_context *ctx, int cell_begin, int cell_end, int flags) {
    if (!ctx) return PNL_ERR_INVALID;
    int rc;
    if ((rc = check_ready(ctx))) return rc;
    if ((rc = finalize(ctx))) return rc;
    if (pnl_tune("PNL_FORCE_SYMFLUSH")) return 0;
    const bool full = ctx->slot_full_list;
    ctx->slot_full_list = true;                          // what pnl_assemble_dense sets
    const bool ok = slot_eligible(ctx, cell_begin, cell_end, flags) && slot_storage_ready(ctx);
    ctx->slot_full_list = full;
    return ok ? 1 : 0;
}

int pnl_assemble_dense_tiles(pnl_context *ctx, double *A, int64_t ldA, int zero_exterior, int ntiles, const int32_t *tiles_host,
                             int cell_begin, int cell_end, int flags) {
    if (!ctx) return PNL_ERR_INVALID;
    int rc;
    if ((rc = check_ready(ctx))) return rc;
    if ((rc = finalize(ctx))) return rc;
    if ((rc = tile_order_ready(ctx))) return rc;
    if (!A || ldA < (ctx->slab_rows ? ctx->slab_cols : ctx->N) || ntiles < 0 || (ntiles && !tiles_host)) return fail(ctx, PNL_ERR_INVALID, "bad arguments");
    if (cell_begin < 0 || cell_end > ctx->nc || cell_begin > cell_end) return fail(ctx, PNL_ERR_INVALID, "bad cell range");
    if ((rc = slab_prepare(ctx, A, flags, cell_begin, cell_end))) return rc;
    std::vector<int2> tiles(ntiles);
    for (int i = 0; i < ntiles; i++) {
        tiles[i] = make_int2(tiles_host[2*i], tiles_host[2*i+1]);
        if (tiles[i].x < 0 || tiles[i].y >= ctx->nblocks || tiles[i].x > tiles[i].y) return fail(ctx, PNL_ERR_INVALID, "bad tile %d", i);
    }
    ctx->tile_cell_filter = false;
    ctx->slot_full_list = false;
    if ((rc = upload_tiles(ctx, tiles, cell_begin, cell_end))) { ctx->tile_cell_filter = true; return rc; }
    ctx->visited_pairs = 0; ctx->visited_is_assembled = false;
    rc = dispatch(ctx, A, ldA, zero_exterior, ntiles, cell_begin, cell_end, flags);
    ctx->tile_cell_filter = true;
    return rc;
}

int pnl_upload_sparsity(pnl_context *ctx, int nnz, const int32_t *indptr, const int32_t *indices) {
    if (!ctx) return PNL_ERR_INVALID;
    if (!ctx->have_dofs) return fail(ctx, PNL_ERR_STATE, "upload the DoF map first");
    if (nnz < 0 || !indptr || (nnz && !indices) || indptr[0] != 0 || indptr[ctx->N] != nnz)
        return fail(ctx, PNL_ERR_INVALID, "bad sparsity pattern (nnz=%d)", nnz);
    for (int i = 0; i < ctx->N; i++) {
        if (indptr[i+1] < indptr[i]) return fail(ctx, PNL_ERR_INVALID, "indptr is not monotone at row %d", i);
        for (int t = indptr[i]; t < indptr[i+1]; t++)
            if (indices[t] < 0 || indices[t] >= ctx->N || (t > indptr[i] && indices[t] <= indices[t-1]))
                return fail(ctx, PNL_ERR_INVALID, "row %d of the pattern is not sorted / in range", i);
    }
    int rc;
    if ((rc = upload(ctx, ctx->b_sp_indptr, indptr, (size_t)ctx->N+1))) return rc;
    if ((rc = upload(ctx, ctx->b_sp_indices, indices, (size_t)nnz))) return rc;
    ctx->sp_nnz = nnz;
    return PNL_OK;
}

int pnl_upload_sparsity_device(pnl_context *ctx, int nnz, const int32_t *indptr_dev, const int32_t *indices_dev) {
    if (!ctx) return PNL_ERR_INVALID;
    if (!ctx->have_dofs) return fail(ctx, PNL_ERR_STATE, "upload the DoF map first");
    if (nnz < 0 || !indptr_dev || (nnz && !indices_dev)) return fail(ctx, PNL_ERR_INVALID, "bad sparsity pattern (nnz=%d)", nnz);
    int rc;
    if ((rc = ensure(ctx, ctx->b_sp_indptr, sizeof(int32_t)*((size_t)ctx->N+1)))) return rc;
    if ((rc = ensure(ctx, ctx->b_sp_indices, sizeof(int32_t)*(size_t)std::max(nnz, 1)))) return rc;
    HIPCHK(ctx, hipMemcpyAsync(ctx->b_sp_indptr.p, indptr_dev, sizeof(int32_t)*((size_t)ctx->N+1), hipMemcpyDeviceToDevice, ctx->stream));
    if (nnz) HIPCHK(ctx, hipMemcpyAsync(ctx->b_sp_indices.p, indices_dev, sizeof(int32_t)*(size_t)nnz, hipMemcpyDeviceToDevice, ctx->stream));
    int32_t ends[2] = {-1, -1};
    HIPCHK(ctx, hipMemcpyAsync(&ends[0], ctx->b_sp_indptr.p, sizeof(int32_t), hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(ctx, hipMemcpyAsync(&ends[1], (const int32_t*)ctx->b_sp_indptr.p+ctx->N, sizeof(int32_t), hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    if (ends[0] != 0 || ends[1] != nnz) { ctx->sp_nnz = -1; return fail(ctx, PNL_ERR_INVALID, "bad sparsity pattern: indptr runs from %d to %d, nnz=%d", ends[0], ends[1], nnz); }
    ctx->sp_nnz = nnz;
    return PNL_OK;
}

static int sparse_ready(pnl_context *ctx, double *data, double *diag, SparseOut &S) {
    int rc;
    if ((rc = check_ready(ctx))) return rc;
    if ((rc = finalize(ctx))) return rc;
    if (ctx->sp_nnz < 0) return fail(ctx, PNL_ERR_STATE, "upload the sparsity pattern first");
    if (!data && ctx->sp_nnz > 0) return fail(ctx, PNL_ERR_INVALID, "null output");
    refresh_tables(ctx);
    S.indptr = (const int*)ctx->b_sp_indptr.p; S.indices = (const int*)ctx->b_sp_indices.p;
    S.data = data; S.diag = diag;
    S.pairs = (const int*)ctx->b_mp_pairs.p; S.masks = (const unsigned long long*)ctx->b_mp_masks.p;
    return PNL_OK;
}

int pnl_assemble_pairs_masked(pnl_context *ctx, int np, const int32_t *pairs, const uint64_t *masks, double *data, double *diag) {
    if (!ctx) return PNL_ERR_INVALID;
    if (np < 0 || (np && !pairs)) return fail(ctx, PNL_ERR_INVALID, "bad pair list");
    for (int i = 0; i < np; i++)
        if (pairs[2*i] < 0 || pairs[2*i] > pairs[2*i+1] || pairs[2*i+1] >= ctx->nc)
            return fail(ctx, PNL_ERR_INVALID, "pair %d = (%d, %d) is not an ordered pair of cells", i, pairs[2*i], pairs[2*i+1]);
    int rc;
    if ((rc = upload(ctx, ctx->b_mp_pairs, pairs, (size_t)2*np))) return rc;
    if (masks && (rc = upload(ctx, ctx->b_mp_masks, masks, (size_t)4*np))) return rc;
    if (!std::isinf(ctx->C().kern[0].horizon2) && ctx->qmax > PNL_CUT_SHIFT)
        return fail(ctx, PNL_ERR_UNSUPPORTED, "finite horizon: upload distant rules up to order %d at most", PNL_CUT_SHIFT);
    SparseOut S;
    if ((rc = sparse_ready(ctx, data, diag, S))) return rc;
    if (!masks) S.masks = nullptr;               // every entry of every pair is requested
    HIPCHK(ctx, hipMemsetAsync(ctx->b_counters.p, 0, sizeof(unsigned long long)*PNL_NCOUNTERS, ctx->stream));
    ctx->visited_pairs = (unsigned long long)np; ctx->visited_is_assembled = false;
    if (np == 0) return PNL_OK;
    // variable order (piecewise constant, symmetric table): the pair list once per class, k_mp_classify keeps the pairs of the
    // class (the interface terms of NA:1966-2156 are boundary items, pnl_assemble_boundary_masked after pnl_select_class)
    // non-symmetric class table (NA:1776-1840 with symmetricCells == False): the listed pairs (c1 <= c2) once per orientation, each
    // with the class of its orientation and half the kernel (the machinery applies the factor 2 of the symmetric case); the masks
    // of (c1, c2) and (c2, c1) request the same DoF pairs, so the list of the symmetric case serves both
    const int ncls = ctx->nlab > 0 ? (int)ctx->cls.size() : 1, cur0 = ctx->cur;
    const int norient = (ctx->nlab > 0 && ctx->nonsym) ? 2 : 1;
    for (int ko = 0; ko < ncls*norient; ko++) {
        const int k = ko/norient;
        ctx->orient = ko%norient;
        if (ctx->nlab > 0) { ctx->cur = k; refresh_tables(ctx); }
        const int kt = ctx->P.k.fast ? 1 : 0;
        const bool first = ko == 0;
        if (ctx->dim == 2 && ctx->dpe == 3)
            rc = kt ? pairs_masked_impl<2, 3, 1>(ctx, np, S, true, first) : pairs_masked_impl<2, 3, 0>(ctx, np, S, true, first);
        else if (ctx->dim == 2 && ctx->dpe == 6)
            rc = kt ? pairs_masked_impl<2, 6, 1>(ctx, np, S, true, first) : pairs_masked_impl<2, 6, 0>(ctx, np, S, true, first);
        else if (ctx->dim == 2 && ctx->dpe == 1)
            rc = kt ? pairs_masked_impl<2, 1, 1>(ctx, np, S, true, first) : pairs_masked_impl<2, 1, 0>(ctx, np, S, true, first);
        else if (ctx->dim == 1 && ctx->dpe == 2) rc = pairs_masked_impl<1, 2, 0>(ctx, np, S, true, first);
        else if (ctx->dim == 1 && ctx->dpe == 1) rc = pairs_masked_impl<1, 1, 0>(ctx, np, S, true, first);
        else if (ctx->dim == 1 && ctx->dpe == 3) rc = pairs_masked_impl<1, 3, 0>(ctx, np, S, true, first);
        else if (ctx->dim == 1 && ctx->dpe == 4) rc = pairs_masked_impl<1, 4, 0>(ctx, np, S, true, first);
        else rc = fail(ctx, PNL_ERR_UNSUPPORTED, "unsupported (dim=%d, dofs_per_element=%d)", ctx->dim, ctx->dpe);
        if (rc) break;
    }
    ctx->cur = cur0; ctx->orient = 0;
    if (norient > 1) refresh_tables(ctx);
    return rc;
}

int pnl_assemble_pairs_in_horizon(pnl_context *ctx, double *data, double *diag) {
    return pnl_assemble_pairs_in_horizon_range(ctx, data, diag, 0, ctx ? ctx->nc : 0);
}

int pnl_assemble_pairs_in_horizon_range(pnl_context *ctx, double *data, double *diag, int cell_begin, int cell_end) {
    if (!ctx) return PNL_ERR_INVALID;
    if (cell_begin < 0 || cell_end > ctx->nc || cell_begin > cell_end) return fail(ctx, PNL_ERR_INVALID, "bad cell range");
    int rc;
    if ((rc = check_ready(ctx))) return rc;
    if ((rc = finalize(ctx))) return rc;
    // (a non-symmetric order table would need the pairs the horizon cuts re-triangulated with the roles of the two cells swapped in the
    // second orientation, NA:1418 swapCells: the cut evaluation of the sorted pipeline takes the listed order)
    if (ctx->nlab > 0 && ctx->nonsym) return fail(ctx, PNL_ERR_UNSUPPORTED, "finite horizon with a non-symmetric order table");
    if (std::isinf(ctx->C().kern[0].horizon2)) return fail(ctx, PNL_ERR_STATE, "pnl_assemble_pairs_in_horizon needs a finite horizon");
    if (ctx->qmax > PNL_CUT_SHIFT) return fail(ctx, PNL_ERR_UNSUPPORTED, "finite horizon: upload distant rules up to order %d at most", PNL_CUT_SHIFT);
    SparseOut S;
    if ((rc = sparse_ready(ctx, data, diag, S))) return rc;
    HIPCHK(ctx, hipMemsetAsync(ctx->b_counters.p, 0, sizeof(unsigned long long)*PNL_NCOUNTERS, ctx->stream));
    const int kt = ctx->P.k.fast ? 1 : 0;
    if (ctx->dim == 2 && ctx->dpe == 3) return kt ? horizon_impl<2, 3, 1>(ctx, S, cell_begin, cell_end) : horizon_impl<2, 3, 0>(ctx, S, cell_begin, cell_end);
    if (ctx->dim == 2 && ctx->dpe == 6) return kt ? horizon_impl<2, 6, 1>(ctx, S, cell_begin, cell_end) : horizon_impl<2, 6, 0>(ctx, S, cell_begin, cell_end);
    if (ctx->dim == 2 && ctx->dpe == 1) return kt ? horizon_impl<2, 1, 1>(ctx, S, cell_begin, cell_end) : horizon_impl<2, 1, 0>(ctx, S, cell_begin, cell_end);
    if (ctx->dim == 1 && ctx->dpe == 2) return horizon_impl<1, 2, 0>(ctx, S, cell_begin, cell_end);
    if (ctx->dim == 1 && ctx->dpe == 1) return horizon_impl<1, 1, 0>(ctx, S, cell_begin, cell_end);
    if (ctx->dim == 1 && ctx->dpe == 3) return horizon_impl<1, 3, 0>(ctx, S, cell_begin, cell_end);
    if (ctx->dim == 1 && ctx->dpe == 4) return horizon_impl<1, 4, 0>(ctx, S, cell_begin, cell_end);
    return fail(ctx, PNL_ERR_UNSUPPORTED, "unsupported (dim=%d, dofs_per_element=%d)", ctx->dim, ctx->dpe);
}

int pnl_assemble_boundary_masked(pnl_context *ctx, int ni, const int32_t *cells, const int32_t *facets, const uint32_t *masks,
                                 double fac, double *data, double *diag) {
    if (!ctx) return PNL_ERR_INVALID;
    if (ni < 0 || (ni && (!cells || !facets || !masks))) return fail(ctx, PNL_ERR_INVALID, "bad item list");
    if (!ctx->C().have_kernel[1] || !ctx->C().have_form[1]) return fail(ctx, PNL_ERR_STATE, "boundary kernel and order formula must be set");
    for (int i = 0; i < ni; i++) {
        if (cells[i] < 0 || cells[i] >= ctx->nc) return fail(ctx, PNL_ERR_INVALID, "item %d: bad cell %d", i, cells[i]);
        for (int k = 0; k < ctx->dim; k++)
            if (facets[(size_t)i*ctx->dim+k] < 0 || facets[(size_t)i*ctx->dim+k] >= ctx->nv)
                return fail(ctx, PNL_ERR_INVALID, "item %d: bad facet vertex", i);
    }
    for (int s = 0; s < ctx->dim; s++)
        if (!ctx->C().have_sing[1][s]) return fail(ctx, PNL_ERR_STATE, "boundary singular rule for %d common vertices not uploaded", s+1);
    int rc;
    if ((rc = upload(ctx, ctx->b_bi_cells, cells, (size_t)ni))) return rc;
    if ((rc = upload(ctx, ctx->b_bi_facets, facets, (size_t)ni*ctx->dim))) return rc;
    if ((rc = upload(ctx, ctx->b_bi_masks, masks, (size_t)ni))) return rc;
    SparseOut S;
    if ((rc = sparse_ready(ctx, data, diag, S))) return rc;
    if (ni == 0) return PNL_OK;
    if (ctx->dim == 2 && ctx->dpe == 3) return boundary_masked_impl<2, 3>(ctx, ni, fac, S);
    if (ctx->dim == 2 && ctx->dpe == 6) return boundary_masked_impl<2, 6>(ctx, ni, fac, S);
    if (ctx->dim == 2 && ctx->dpe == 1) return boundary_masked_impl<2, 1>(ctx, ni, fac, S);
    if (ctx->dim == 1 && ctx->dpe == 2) return boundary_masked_impl<1, 2>(ctx, ni, fac, S);
    if (ctx->dim == 1 && ctx->dpe == 1) return boundary_masked_impl<1, 1>(ctx, ni, fac, S);
    if (ctx->dim == 1 && ctx->dpe == 3) return boundary_masked_impl<1, 3>(ctx, ni, fac, S);
    if (ctx->dim == 1 && ctx->dpe == 4) return boundary_masked_impl<1, 4>(ctx, ni, fac, S);
    return fail(ctx, PNL_ERR_UNSUPPORTED, "unsupported (dim=%d, dofs_per_element=%d)", ctx->dim, ctx->dpe);
}

int pnl_assemble_clusters_tiled(pnl_context *ctx, const pnl_cluster_plan *pl, int cluster_boundary, double *data, double *diag) {
    if (!ctx || !pl) return PNL_ERR_INVALID;
    if (ctx->nlab > 0) return fail(ctx, PNL_ERR_UNSUPPORTED, "cluster assembly with a variable order needs the jump terms (NA:1966-2156)");
    if (!std::isinf(ctx->C().kern[0].horizon2)) return fail(ctx, PNL_ERR_UNSUPPORTED, "tiled cluster assembly: infinite horizon only");
    int rc;
    SparseOut S;
    if ((rc = sparse_ready(ctx, data, diag, S))) return rc;
    if (pl->tile != ctx->tile) return fail(ctx, PNL_ERR_INVALID, "plan built for tiles of %d cells, the kernels use %d", pl->tile, ctx->tile);
    if (pl->npairs < 0 || pl->ntiles < 0 || pl->num_dslots < 0 || pl->chunk_stride < 1) return fail(ctx, PNL_ERR_INVALID, "bad plan sizes");
    const int T = pl->tile, dim = ctx->dim, dpe = ctx->dpe;
    // light validation of the index arrays (a wrong index here would be an out-of-bounds access on the device)
    for (int t = 0; t < pl->ntiles; t++) {
        if (pl->tile_chunkA[t] < 0 || pl->tile_chunkA[t] >= pl->nchunks || pl->tile_chunkB[t] < 0 || pl->tile_chunkB[t] >= pl->nchunks ||
            pl->tile_pair[t] < 0 || pl->tile_pair[t] >= pl->npairs)
            return fail(ctx, PNL_ERR_INVALID, "tile %d out of range", t);
        for (int l = 0; l < T; l++)
            if (pl->tile_dslotA[(size_t)t*T+l] >= pl->num_dslots || pl->tile_dslotB[(size_t)t*T+l] >= pl->num_dslots)
                return fail(ctx, PNL_ERR_INVALID, "tile %d: diagonal-block slot out of range", t);
    }
    for (size_t i = 0; i < (size_t)pl->nchunks*T; i++)
        if (pl->chunk_cells[i] >= ctx->nc) return fail(ctx, PNL_ERR_INVALID, "chunk cell out of range");
    for (int c = 0; c < pl->nchunks; c++)
        if (pl->chunk_ndof[c] < 0 || pl->chunk_ndof[c] > pl->chunk_stride) return fail(ctx, PNL_ERR_INVALID, "chunk %d: bad DoF count", c);
    for (size_t i = 0; i < (size_t)pl->nchunks*dpe*T; i++)
        if (pl->chunk_slot[i] >= pl->chunk_stride) return fail(ctx, PNL_ERR_INVALID, "chunk slot out of range");
    for (int k = 0; k < 2*pl->npairs; k++)
        if (pl->pair_nodes[k] < 0 || pl->pair_nodes[k] >= pl->nnodes) return fail(ctx, PNL_ERR_INVALID, "pair node out of range");
    for (int d = 0; d < pl->num_dslots; d++)
        if (pl->d_cell[d] < 0 || pl->d_cell[d] >= ctx->nc || pl->d_pair[d] < 0 || pl->d_pair[d] >= pl->npairs)
            return fail(ctx, PNL_ERR_INVALID, "diagonal-block slot %d out of range", d);
    for (size_t i = 0; i < (size_t)pl->nfacets*dim; i++)
        if (pl->fvid[i] < 0 || pl->fvid[i] >= ctx->nv) return fail(ctx, PNL_ERR_INVALID, "facet vertex out of range");
    DevBuf *B = ctx->b_cp;
    int nb = 0;
#define UP(ptr, n) ((rc = upload(ctx, B[nb], ptr, (size_t)(n))) ? nullptr : B[nb++].p)
    ClusterTiles CT;
    std::memset(&CT, 0, sizeof(CT));
    CT.npairs = pl->npairs; CT.chunk_stride = pl->chunk_stride; CT.S = S;
    if (!(CT.pair_nodes = (const int*)UP(pl->pair_nodes, 2*pl->npairs))) return rc;
    if (!(CT.node_off = (const int*)UP(pl->node_off, pl->nnodes+1))) return rc;
    if (!(CT.node_dofs = (const int*)UP(pl->node_dofs, pl->node_off[pl->nnodes]))) return rc;
    if (!(CT.chunk_cells = (const int*)UP(pl->chunk_cells, (size_t)pl->nchunks*T))) return rc;
    if (!(CT.chunk_ndof = (const int*)UP(pl->chunk_ndof, pl->nchunks))) return rc;
    if (!(CT.chunk_dofs = (const int*)UP(pl->chunk_dofs, (size_t)pl->nchunks*pl->chunk_stride))) return rc;
    if (!(CT.chunk_slot = (const short*)UP(pl->chunk_slot, (size_t)pl->nchunks*dpe*T))) return rc;
    if (!(CT.chunkA = (const int*)UP(pl->tile_chunkA, pl->ntiles))) return rc;
    if (!(CT.chunkB = (const int*)UP(pl->tile_chunkB, pl->ntiles))) return rc;
    if (!(CT.pair = (const int*)UP(pl->tile_pair, pl->ntiles))) return rc;
    if (!(CT.flags = (const int*)UP(pl->tile_flags, pl->ntiles))) return rc;
    if (!(CT.dslotA = (const int*)UP(pl->tile_dslotA, (size_t)pl->ntiles*T))) return rc;
    if (!(CT.dslotB = (const int*)UP(pl->tile_dslotB, (size_t)pl->ntiles*T))) return rc;
    const int *d_cell, *d_pair, *pair_foff, *fvid, *bt_cell, *bt_facet;
    const unsigned *bt_slot;
    if (!(d_cell = (const int*)UP(pl->d_cell, pl->num_dslots))) return rc;
    if (!(d_pair = (const int*)UP(pl->d_pair, pl->num_dslots))) return rc;
    if (!(pair_foff = (const int*)UP(pl->pair_foff, pl->npairs+1))) return rc;
    if (!(fvid = (const int*)UP(pl->fvid, (size_t)pl->nfacets*dim))) return rc;
    if (!(bt_cell = (const int*)UP(pl->bt_cell, pl->n_btouch))) return rc;
    if (!(bt_facet = (const int*)UP(pl->bt_facet, (size_t)pl->n_btouch*dim))) return rc;
    if (!(bt_slot = (const unsigned*)UP((const unsigned*)pl->bt_slot, pl->n_btouch))) return rc;
    const int2 *sing_dev[3] = {nullptr, nullptr, nullptr};
    const int *sing_pair_dev[3] = {nullptr, nullptr, nullptr};
    for (int s = 0; s < 3; s++) {
        const int n = pl->n_sing[s];
        std::vector<int2> pr(n);
        std::vector<int> pk(n);
        for (int i = 0; i < n; i++) {
            const int32_t *it = pl->sing_items[s]+3*(size_t)i;
            if (it[0] < 0 || it[0] >= pl->npairs || it[1] < 0 || it[1] > it[2] || it[2] >= ctx->nc)
                return fail(ctx, PNL_ERR_INVALID, "touching item %d of slot %d out of range", i, s);
            pk[i] = it[0]; pr[i] = make_int2(it[1], it[2]);
        }
        if (!(sing_dev[s] = (const int2*)UP(pr.data(), n))) return rc;
        if (!(sing_pair_dev[s] = (const int*)UP(pk.data(), n))) return rc;
    }
    // facet geometry (centre, unit normal, length, |ln(len/H0)|, ln(len)) like DevProblem::bgeo
    int maxf = 0;
    std::vector<double> geo((size_t)(2*dim+3)*std::max(pl->nfacets, 1), 0.);
    for (int k = 0; k < pl->npairs; k++) maxf = std::max(maxf, pl->pair_foff[k+1]-pl->pair_foff[k]);
    for (int f = 0; f < pl->nfacets; f++) {
        const size_t nf = pl->nfacets;
        double len = 1.;
        const double *v0 = &ctx->vertices[(size_t)pl->fvid[(size_t)f*dim]*dim];
        const double *v1 = &ctx->vertices[(size_t)pl->fvid[(size_t)f*dim+(dim-1)]*dim];
        for (int d = 0; d < dim; d++) geo[(size_t)d*nf+f] = dim == 2 ? 0.5*(v0[d]+v1[d]) : v0[d];
        if (dim == 2) {
            const double n0 = v1[1]-v0[1], n1 = v0[0]-v1[0];
            const double inv = 1./std::sqrt(n0*n0+n1*n1);
            geo[(size_t)(dim+0)*nf+f] = n0*inv; geo[(size_t)(dim+1)*nf+f] = n1*inv;
            const double dx = v1[0]-v0[0], dy = v1[1]-v0[1];
            len = std::sqrt(dx*dx+dy*dy);
        }
        geo[(size_t)(2*dim)*nf+f] = len;
        geo[(size_t)(2*dim+1)*nf+f] = std::fabs(std::log(len/ctx->H0));
        geo[(size_t)(2*dim+2)*nf+f] = std::log(len);
    }
    const double *fgeo;
    if (!(fgeo = (const double*)UP(geo.data(), geo.size()))) return rc;
#undef UP
    if ((rc = ensure(ctx, ctx->b_cpD, sizeof(double)*(size_t)std::max(pl->num_dslots, 1)*(dpe*(dpe+1)/2)))) return rc;
    CT.D = (double*)ctx->b_cpD.p;
    ctx->visited_pairs = 0; ctx->visited_is_assembled = false;
    const bool kt = ctx->P.k.fast;
    if (dim == 2 && dpe == 3)
        return kt ? clusters_tiled_impl<2, 3, TILE_P1, 1>(ctx, pl, CT, cluster_boundary, d_cell, d_pair, sing_dev, sing_pair_dev, pair_foff, fvid, fgeo, maxf, bt_cell, bt_facet, bt_slot)
                  : clusters_tiled_impl<2, 3, TILE_P1, 0>(ctx, pl, CT, cluster_boundary, d_cell, d_pair, sing_dev, sing_pair_dev, pair_foff, fvid, fgeo, maxf, bt_cell, bt_facet, bt_slot);
    if (dim == 2 && dpe == 6)
        return kt ? clusters_tiled_impl<2, 6, TILE_P2, 1>(ctx, pl, CT, cluster_boundary, d_cell, d_pair, sing_dev, sing_pair_dev, pair_foff, fvid, fgeo, maxf, bt_cell, bt_facet, bt_slot)
                  : clusters_tiled_impl<2, 6, TILE_P2, 0>(ctx, pl, CT, cluster_boundary, d_cell, d_pair, sing_dev, sing_pair_dev, pair_foff, fvid, fgeo, maxf, bt_cell, bt_facet, bt_slot);
    if (dim == 1 && dpe == 2)
        return kt ? clusters_tiled_impl<1, 2, TILE_P1, 1>(ctx, pl, CT, cluster_boundary, d_cell, d_pair, sing_dev, sing_pair_dev, pair_foff, fvid, fgeo, maxf, bt_cell, bt_facet, bt_slot)
                  : clusters_tiled_impl<1, 2, TILE_P1, 0>(ctx, pl, CT, cluster_boundary, d_cell, d_pair, sing_dev, sing_pair_dev, pair_foff, fvid, fgeo, maxf, bt_cell, bt_facet, bt_slot);
    // P0 (intervals and triangles), P2 and P3 on intervals
#define PNL_CT(D_, E_, T_) \
    if (dim == D_ && dpe == E_) \
        return kt ? clusters_tiled_impl<D_, E_, T_, 1>(ctx, pl, CT, cluster_boundary, d_cell, d_pair, sing_dev, sing_pair_dev, pair_foff, fvid, fgeo, maxf, bt_cell, bt_facet, bt_slot) \
                  : clusters_tiled_impl<D_, E_, T_, 0>(ctx, pl, CT, cluster_boundary, d_cell, d_pair, sing_dev, sing_pair_dev, pair_foff, fvid, fgeo, maxf, bt_cell, bt_facet, bt_slot);
    PNL_CT(2, 1, TILE_P1)
    PNL_CT(1, 1, TILE_P1)
    PNL_CT(1, 3, TILE_P2)
    PNL_CT(1, 4, TILE_P2)
#undef PNL_CT
    return fail(ctx, PNL_ERR_UNSUPPORTED, "unsupported (dim=%d, dofs_per_element=%d)", dim, dpe);
}

int pnl_h2_setup(pnl_context *ctx, const pnl_h2_plan *pl) {
    if (!ctx || !pl) return PNL_ERR_INVALID;
    int rc;
    if (ctx->have_pw) {
        // order per quadrature point: no kernel block of a class, the order function evaluates s(x) (pnl_pwnear.hip)
        if (ctx->pw.type == 5) return fail(ctx, PNL_ERR_UNSUPPORTED, "H2 far field of an order given as a finite element function");
        if ((rc = pnl_pw_prepare(ctx, 0))) return rc;
    } else {
    if ((rc = check_ready(ctx))) return rc;
    if ((rc = finalize(ctx))) return rc;
    refresh_tables(ctx);
    // finite horizon: every admissible pair must lie inside it (pnl_tree_build_horizon drops the pairs beyond the horizon and keeps the
    // ones it may cut in the near field, clusterMethodCy.pyx:4069-4090); the interpolants are those of the kernel itself
    if (!std::isinf(ctx->C().kern[0].horizon2)) {
        if (ctx->have_xform) return fail(ctx, PNL_ERR_UNSUPPORTED, "H2 far field of a finite horizon: l2 ball only");
        if (!pl->box || (pl->nfar > 0 && !pl->far)) return PNL_ERR_INVALID;
        const double h2 = ctx->C().kern[0].horizon2;
        for (int p = 0; p < pl->nfar; p++) {
            const double *a = pl->box+(size_t)pl->far[2*p]*ctx->dim*2, *b = pl->box+(size_t)pl->far[2*p+1]*ctx->dim*2;
            // maxDistBoxes as the reference writes it (interactionDomains.pyx:325-337): what its admissibility test compares with the
            // horizon; interpolation nodes that do lie beyond the horizon get the kernel value 0 there and here (kern_eval)
            double d2 = 0.;
            for (int d = 0; d < ctx->dim; d++) {
                const bool first = a[2*d] > b[2*d];
                const double e = std::max((first ? a[2*d+1] : b[2*d+1])-(first ? b[2*d] : a[2*d]), 0.);
                d2 += e*e;
            }
            if (d2 > h2*(1.+1e-12))
                return fail(ctx, PNL_ERR_INVALID, "H2 far field: the clusters of admissible pair %d reach beyond the horizon", p);
        }
    }
    }
    // (the admissible pairs are ORDERED -- (n1, n2) and (n2, n1) are two entries, each with the class of its orientation -- so a
    // non-symmetric order table needs nothing beyond its far_class)
    if (ctx->nlab > 0 && pl->nfar > 0 && !pl->far_class)
        return fail(ctx, PNL_ERR_UNSUPPORTED, "H2 far field of a variable order: a kernel class per admissible pair is needed");
    if (pl->far_class)
        for (int i = 0; i < pl->nfar; i++)
            if (pl->far_class[i] < 0 || pl->far_class[i] >= (int)ctx->cls.size()) return fail(ctx, PNL_ERR_INVALID, "far pair %d: bad kernel class", i);
    const int dim = ctx->dim, m = pl->m;
    if (pl->nnodes <= 0 || pl->nleaves <= 0 || pl->nfar < 0 || m < 1 || m > 16 || pl->nq <= 0) return fail(ctx, PNL_ERR_INVALID, "bad H2 plan sizes");
    int M = 1;
    for (int d = 0; d < dim; d++) M *= m;
    int nroot = 0;
    for (int n = 0; n < pl->nnodes; n++) {
        if (pl->parent[n] < -1 || pl->parent[n] >= pl->nnodes || pl->level[n] < 0 || pl->level[n] >= pl->nlevels)
            return fail(ctx, PNL_ERR_INVALID, "node %d: bad parent / level", n);
        if (pl->parent[n] < 0) nroot++;
        else if (pl->level[pl->parent[n]] != pl->level[n]-1) return fail(ctx, PNL_ERR_INVALID, "node %d: level is not its parent's + 1", n);
    }
    if (nroot != 1) return fail(ctx, PNL_ERR_INVALID, "the tree needs exactly one root");
    for (int i = 0; i < 2*pl->nfar; i++)
        if (pl->far[i] < 0 || pl->far[i] >= pl->nnodes) return fail(ctx, PNL_ERR_INVALID, "far pair out of range");
    std::vector<long long> voff(pl->nleaves);
    long long vtot = 0;
    std::vector<char> covered(ctx->N, 0);
    for (int l = 0; l < pl->nleaves; l++) {
        if (pl->leaf_node[l] < 0 || pl->leaf_node[l] >= pl->nnodes) return fail(ctx, PNL_ERR_INVALID, "leaf %d: bad node", l);
        voff[l] = vtot;
        vtot += (long long)(pl->leaf_dof_off[l+1]-pl->leaf_dof_off[l])*M;
        for (int t = pl->leaf_dof_off[l]; t < pl->leaf_dof_off[l+1]; t++) {
            const int I = pl->leaf_dofs[t];
            if (I < 0 || I >= ctx->N || covered[I] || (t > pl->leaf_dof_off[l] && pl->leaf_dofs[t-1] >= I))
                return fail(ctx, PNL_ERR_INVALID, "leaf %d: DoFs must be sorted and the leaves must partition the DoFs", l);
            covered[I] = 1;
        }
        for (int t = pl->leaf_cell_off[l]; t < pl->leaf_cell_off[l+1]; t++)
            if (pl->leaf_cells[t] < 0 || pl->leaf_cells[t] >= ctx->nc) return fail(ctx, PNL_ERR_INVALID, "leaf %d: bad cell", l);
    }
    if (!pl->partial_leaves)
        for (int I = 0; I < ctx->N; I++)
            if (!covered[I]) return fail(ctx, PNL_ERR_INVALID, "DoF %d belongs to no leaf (set partial_leaves for a rank-local plan)", I);
    DevBuf *B = ctx->b_h2;
    H2Dev &H = ctx->h2;
    std::memset(&H, 0, sizeof(H));
    H.dim = dim; H.m = m; H.M = M; H.nnodes = pl->nnodes; H.nleaves = pl->nleaves; H.nfar = pl->nfar;
    if ((rc = upload(ctx, B[0], pl->box, (size_t)pl->nnodes*dim*2))) return rc;
    if ((rc = upload(ctx, B[1], pl->parent, (size_t)pl->nnodes))) return rc;
    if ((rc = upload(ctx, B[2], pl->leaf_node, (size_t)pl->nleaves))) return rc;
    if ((rc = upload(ctx, B[3], pl->leaf_dof_off, (size_t)pl->nleaves+1))) return rc;
    if ((rc = upload(ctx, B[4], pl->leaf_dofs, (size_t)pl->leaf_dof_off[pl->nleaves]))) return rc;
    if ((rc = upload(ctx, B[5], pl->leaf_cell_off, (size_t)pl->nleaves+1))) return rc;
    if ((rc = upload(ctx, B[6], pl->leaf_cells, (size_t)pl->leaf_cell_off[pl->nleaves]))) return rc;
    if ((rc = upload(ctx, B[7], voff.data(), voff.size()))) return rc;
    ctx->h2_vtot = vtot;
    if ((rc = upload(ctx, B[8], pl->far, (size_t)2*pl->nfar))) return rc;
    if ((rc = upload(ctx, B[9], pl->transfer, (size_t)pl->nnodes*M*M))) return rc;
    if ((rc = ensure(ctx, B[10], sizeof(double)*(size_t)std::max<long long>(vtot, 1)))) return rc;
    if ((rc = ensure(ctx, B[11], sizeof(double)*(size_t)std::max(pl->nfar, 1)*M*M))) return rc;
    if ((rc = ensure(ctx, B[12], sizeof(double)*(size_t)pl->nnodes*M))) return rc;
    if ((rc = ensure(ctx, B[13], sizeof(double)*(size_t)pl->nnodes*M))) return rc;
    if ((rc = upload(ctx, B[14], pl->qbary, (size_t)3*pl->nq))) return rc;
    if ((rc = upload(ctx, B[15], pl->qw, (size_t)pl->nq))) return rc;
    if ((rc = upload(ctx, B[16], pl->qphi, (size_t)pl->nq*ctx->dpe))) return rc;
    H.box = (const double*)B[0].p; H.parent = (const int*)B[1].p; H.leaf_node = (const int*)B[2].p;
    H.leaf_dof_off = (const int*)B[3].p; H.leaf_dofs = (const int*)B[4].p; H.leaf_cell_off = (const int*)B[5].p;
    H.leaf_cells = (const int*)B[6].p; H.leaf_val_off = (const long long*)B[7].p; H.far = (const int*)B[8].p;
    H.T = (const double*)B[9].p; H.V = (double*)B[10].p; H.K = (double*)B[11].p; H.cup = (double*)B[12].p; H.cdown = (double*)B[13].p;
    // nodes per level (children lists), concatenated on the device
    ctx->h2_levels.assign(pl->nlevels, std::vector<int>());
    for (int n = 0; n < pl->nnodes; n++)
        if (pl->parent[n] >= 0) ctx->h2_levels[pl->level[n]].push_back(n);
    std::vector<int> cat;
    ctx->h2_level_off.assign(pl->nlevels+1, 0);
    for (int l = 0; l < pl->nlevels; l++) {
        ctx->h2_level_off[l] = cat.size();
        cat.insert(cat.end(), ctx->h2_levels[l].begin(), ctx->h2_levels[l].end());
    }
    ctx->h2_level_off[pl->nlevels] = cat.size();
    if ((rc = upload(ctx, B[17], cat.data(), cat.size()))) return rc;
    HIPCHK(ctx, hipMemsetAsync(H.V, 0, sizeof(double)*(size_t)std::max<long long>(vtot, 1), ctx->stream));
    const double *qb = (const double*)B[14].p, *qw = (const double*)B[15].p, *qp = (const double*)B[16].p;
    if (dim == 2 && ctx->dpe == 3) hipLaunchKernelGGL((k_h2_leaf_values<2, 3>), dim3(pl->nleaves), dim3(PNL_NTHREADS), 0, ctx->stream, ctx->P, H, pl->nq, qb, qw, qp);
    else if (dim == 2 && ctx->dpe == 6) hipLaunchKernelGGL((k_h2_leaf_values<2, 6>), dim3(pl->nleaves), dim3(PNL_NTHREADS), 0, ctx->stream, ctx->P, H, pl->nq, qb, qw, qp);
    else if (dim == 1 && ctx->dpe == 2) hipLaunchKernelGGL((k_h2_leaf_values<1, 2>), dim3(pl->nleaves), dim3(PNL_NTHREADS), 0, ctx->stream, ctx->P, H, pl->nq, qb, qw, qp);
    else if (dim == 2 && ctx->dpe == 1) hipLaunchKernelGGL((k_h2_leaf_values<2, 1>), dim3(pl->nleaves), dim3(PNL_NTHREADS), 0, ctx->stream, ctx->P, H, pl->nq, qb, qw, qp);
    else if (dim == 1 && ctx->dpe == 1) hipLaunchKernelGGL((k_h2_leaf_values<1, 1>), dim3(pl->nleaves), dim3(PNL_NTHREADS), 0, ctx->stream, ctx->P, H, pl->nq, qb, qw, qp);
    else if (dim == 1 && ctx->dpe == 3) hipLaunchKernelGGL((k_h2_leaf_values<1, 3>), dim3(pl->nleaves), dim3(PNL_NTHREADS), 0, ctx->stream, ctx->P, H, pl->nq, qb, qw, qp);
    else if (dim == 1 && ctx->dpe == 4) hipLaunchKernelGGL((k_h2_leaf_values<1, 4>), dim3(pl->nleaves), dim3(PNL_NTHREADS), 0, ctx->stream, ctx->P, H, pl->nq, qb, qw, qp);
    else return fail(ctx, PNL_ERR_UNSUPPORTED, "unsupported (dim=%d, dofs_per_element=%d)", dim, ctx->dpe);
    if (pl->nfar > 0) {
        const DevKernel *kcls = nullptr;
        const int *fcls = nullptr;
        if (pl->far_class) {
            std::vector<DevKernel> kc;
            for (auto *c : ctx->cls) kc.push_back(to_dev(c->kern[0], dim));
            if ((rc = upload(ctx, B[18], kc.data(), kc.size()))) return rc;
            if ((rc = upload(ctx, B[19], pl->far_class, (size_t)pl->nfar))) return rc;
            kcls = (const DevKernel*)B[18].p; fcls = (const int*)B[19].p;
        }
        if (ctx->have_pw) {
            // order per quadrature point: the kernel with the order at the nodes of the row cluster (pnl_pwnear.hip)
            if ((rc = pnl_pw_h2_interp(ctx))) return rc;
        } else
        if (dim == 2) hipLaunchKernelGGL((k_h2_kernel_interp<2>), dim3(pl->nfar), dim3(PNL_NTHREADS), 0, ctx->stream, ctx->P, H, kcls, fcls);
        else hipLaunchKernelGGL((k_h2_kernel_interp<1>), dim3(pl->nfar), dim3(PNL_NTHREADS), 0, ctx->stream, ctx->P, H, kcls, fcls);
    }
    HIPCHK(ctx, hipGetLastError());
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    ctx->have_h2 = true;
    return PNL_OK;
}

int pnl_h2_matvec(pnl_context *ctx, const double *x, double *y) {
    if (!ctx || !x || !y) return PNL_ERR_INVALID;
    if (!ctx->have_h2) return fail(ctx, PNL_ERR_STATE, "pnl_h2_setup first");
    const H2Dev &H = ctx->h2;
    const int nlev = (int)ctx->h2_levels.size();
    const int *lev = (const int*)ctx->b_h2[17].p;
    HIPCHK(ctx, hipMemsetAsync(H.cup, 0, sizeof(double)*(size_t)H.nnodes*H.M, ctx->stream));
    HIPCHK(ctx, hipMemsetAsync(H.cdown, 0, sizeof(double)*(size_t)H.nnodes*H.M, ctx->stream));
    hipLaunchKernelGGL(k_h2_up_leaves, dim3(H.nleaves), dim3(64), 0, ctx->stream, H, x);
    for (int l = nlev-1; l >= 1; l--) {
        const int n = (int)ctx->h2_levels[l].size();
        if (n) hipLaunchKernelGGL(k_h2_up_level, dim3(n), dim3(64), 0, ctx->stream, H, lev+ctx->h2_level_off[l], n);
    }
    if (H.nfar) hipLaunchKernelGGL(k_h2_far, dim3(H.nfar), dim3(64), 0, ctx->stream, H);
    for (int l = 1; l < nlev; l++) {
        const int n = (int)ctx->h2_levels[l].size();
        if (n) hipLaunchKernelGGL(k_h2_down_level, dim3(n), dim3(64), 0, ctx->stream, H, lev+ctx->h2_level_off[l], n);
    }
    hipLaunchKernelGGL(k_h2_down_leaves, dim3(H.nleaves), dim3(64), 0, ctx->stream, H, y);
    HIPCHK(ctx, hipGetLastError());
    return PNL_OK;
}

// kernel interpolants K[nfar][M][M] (which = 0) and leaf values V (which = 1: the blocks V_leaf[ndofs][M] of the plan's leaves, one
// after the other) between the device and the host: the H2 operator file (clusterMethodCy.pyx:2449-2550) stores them
static int h2_copy(pnl_context *ctx, int which, double *host, bool to_host) {
    if (!ctx || !host) return PNL_ERR_INVALID;
    if (!ctx->have_h2) return fail(ctx, PNL_ERR_STATE, "pnl_h2_setup first");
    if (which != 0 && which != 1) return fail(ctx, PNL_ERR_INVALID, "pnl_h2_get / _set: which = 0 (interpolants) or 1 (leaf values)");
    const H2Dev &H = ctx->h2;
    const size_t n = which == 0 ? (size_t)H.nfar*H.M*H.M : (size_t)ctx->h2_vtot;
    double *dev = which == 0 ? H.K : H.V;
    if (n) HIPCHK(ctx, hipMemcpyAsync(to_host ? (void*)host : (void*)dev, to_host ? (const void*)dev : (const void*)host, n*sizeof(double),
                                      to_host ? hipMemcpyDeviceToHost : hipMemcpyHostToDevice, ctx->stream));
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    return PNL_OK;
}
int pnl_h2_get(pnl_context *ctx, int which, double *dst_host) { return h2_copy(ctx, which, dst_host, true); }
int pnl_h2_set(pnl_context *ctx, int which, const double *src_host) { return h2_copy(ctx, which, const_cast<double*>(src_host), false); }

int pnl_h2_sizes(pnl_context *ctx, int32_t *out2) {
    if (!ctx || !out2) return PNL_ERR_INVALID;
    if (!ctx->have_h2) return fail(ctx, PNL_ERR_STATE, "pnl_h2_setup first");
    out2[0] = ctx->h2.nnodes; out2[1] = ctx->h2.M;
    return PNL_OK;
}

int pnl_h2_upward(pnl_context *ctx, const double *x, double *cup) {
    if (!ctx || !x || !cup) return PNL_ERR_INVALID;
    if (!ctx->have_h2) return fail(ctx, PNL_ERR_STATE, "pnl_h2_setup first");
    H2Dev H = ctx->h2;
    H.cup = cup;
    const int nlev = (int)ctx->h2_levels.size();
    const int *lev = (const int*)ctx->b_h2[17].p;
    HIPCHK(ctx, hipMemsetAsync(cup, 0, sizeof(double)*(size_t)H.nnodes*H.M, ctx->stream));
    hipLaunchKernelGGL(k_h2_up_leaves, dim3(H.nleaves), dim3(64), 0, ctx->stream, H, x);
    for (int l = nlev-1; l >= 1; l--) {
        const int n = (int)ctx->h2_levels[l].size();
        if (n) hipLaunchKernelGGL(k_h2_up_level, dim3(n), dim3(64), 0, ctx->stream, H, lev+ctx->h2_level_off[l], n);
    }
    HIPCHK(ctx, hipGetLastError());
    return PNL_OK;
}

int pnl_h2_interact(pnl_context *ctx, const double *cup, double *cdown) {
    if (!ctx || !cup || !cdown) return PNL_ERR_INVALID;
    if (!ctx->have_h2) return fail(ctx, PNL_ERR_STATE, "pnl_h2_setup first");
    H2Dev H = ctx->h2;
    H.cup = const_cast<double*>(cup); H.cdown = cdown;
    HIPCHK(ctx, hipMemsetAsync(cdown, 0, sizeof(double)*(size_t)H.nnodes*H.M, ctx->stream));
    if (H.nfar) hipLaunchKernelGGL(k_h2_far, dim3(H.nfar), dim3(64), 0, ctx->stream, H);
    HIPCHK(ctx, hipGetLastError());
    return PNL_OK;
}

int pnl_h2_downward(pnl_context *ctx, double *cdown, double *y) {
    if (!ctx || !cdown || !y) return PNL_ERR_INVALID;
    if (!ctx->have_h2) return fail(ctx, PNL_ERR_STATE, "pnl_h2_setup first");
    H2Dev H = ctx->h2;
    H.cdown = cdown;
    const int nlev = (int)ctx->h2_levels.size();
    const int *lev = (const int*)ctx->b_h2[17].p;
    for (int l = 1; l < nlev; l++) {
        const int n = (int)ctx->h2_levels[l].size();
        if (n) hipLaunchKernelGGL(k_h2_down_level, dim3(n), dim3(64), 0, ctx->stream, H, lev+ctx->h2_level_off[l], n);
    }
    hipLaunchKernelGGL(k_h2_down_leaves, dim3(H.nleaves), dim3(64), 0, ctx->stream, H, y);
    HIPCHK(ctx, hipGetLastError());
    return PNL_OK;
}

int pnl_spmv(pnl_context *ctx, const double *data, const double *diag, const double *x, double *y) {
    if (!ctx || !x || !y) return PNL_ERR_INVALID;
    if (ctx->sp_nnz < 0) return fail(ctx, PNL_ERR_STATE, "upload the sparsity pattern first");
    if (!data && ctx->sp_nnz > 0) return fail(ctx, PNL_ERR_INVALID, "null matrix data");
    const int n = ctx->N;
    if (diag) HIPCHK(ctx, hipMemsetAsync(y, 0, sizeof(double)*n, ctx->stream));
    hipLaunchKernelGGL(k_spmv, dim3((n+3)/4), dim3(PNL_NTHREADS), 0, ctx->stream, (const int*)ctx->b_sp_indptr.p,
                       (const int*)ctx->b_sp_indices.p, data, diag, n, x, y);
    HIPCHK(ctx, hipGetLastError());
    return PNL_OK;
}

int pnl_inv_diagonal(pnl_context *ctx, const double *A, int64_t ldA, int n, double *dinv) {
    if (!ctx || !A || !dinv || n <= 0 || ldA < n) return fail(ctx, PNL_ERR_INVALID, "bad arguments");
    hipLaunchKernelGGL(k_diag_inv, dim3((n+PNL_NTHREADS-1)/PNL_NTHREADS), dim3(PNL_NTHREADS), 0, ctx->stream, A, (long long)ldA, n, dinv);
    HIPCHK(ctx, hipGetLastError());
    return PNL_OK;
}

// ---- non-symmetric kernels with an order s(x) per quadrature point ------------------------------------------------------
int pnl_set_order_function(pnl_context *ctx, const pnl_order_function *f, const double *cell_smax, const double *facet_smax,
                           double c0, double bc0, double sing_fac, double bsing_fac) {
    if (!ctx || !f || !cell_smax) return fail(ctx, PNL_ERR_INVALID, "bad order-function arguments");
    if (!ctx->have_mesh) return fail(ctx, PNL_ERR_STATE, "upload the mesh first");
    if (f->type < 1 || f->type > 5) return fail(ctx, PNL_ERR_UNSUPPORTED, "order function type %d is not implemented", f->type);
    if (ctx->have_dofs && !(ctx->dpe == ctx->dim+1 || (ctx->dim == 2 && ctx->dpe == 6) || (ctx->dim == 1 && ctx->dpe == 3)))
        return fail(ctx, PNL_ERR_UNSUPPORTED, "pointwise variable orders: P1 and P2 elements");
    std::memset(&ctx->pw, 0, sizeof(ctx->pw));
    ctx->pw.type = f->type; ctx->pw.normalized = f->normalized;
    for (int i = 0; i < 6; i++) ctx->pw.p[i] = f->p[i];
    if (f->scal_n < 0 || f->scal_n > 32 || (f->scal_n > 0 && !(f->scal_half > 0.))) return fail(ctx, PNL_ERR_INVALID, "bad Chebyshev series of the scaling");
    ctx->pw.scal_n = f->scal_n; ctx->pw.scal_mid = f->scal_mid; ctx->pw.scal_inv_half = f->scal_n > 0 ? 1./f->scal_half : 0.;
    for (int i = 0; i < 32; i++) ctx->pw.scal_cheb[i] = i < f->scal_n ? f->scal_cheb[i] : 0.;
    ctx->pw.c0 = c0; ctx->pw.bc0 = bc0; ctx->pw.sfac = sing_fac; ctx->pw.bfac = bsing_fac;
    ctx->pw_cell_smax.assign(cell_smax, cell_smax+ctx->nc);
    ctx->pw_facet_smax.clear();
    if (facet_smax && ctx->have_boundary) ctx->pw_facet_smax.assign(facet_smax, facet_smax+ctx->nb);
    for (int w = 0; w < 2; w++) for (int s = 0; s < 3; s++) ctx->have_pw_rules[w][s] = false;
    ctx->pw_vertex_s.clear();
    ctx->have_pw = true;
    return PNL_OK;
}

int pnl_set_order_vertex_values(pnl_context *ctx, int nv, const double *values) {
    if (!ctx || !values) return PNL_ERR_INVALID;
    if (!ctx->have_pw || ctx->pw.type != 5) return fail(ctx, PNL_ERR_STATE, "set an order function of type 5 first");
    if (nv != ctx->nv) return fail(ctx, PNL_ERR_INVALID, "pnl_set_order_vertex_values: %d vertices expected", ctx->nv);
    ctx->pw_vertex_s.assign(values, values+nv);
    return PNL_OK;
}

int pnl_upload_pointwise_rules(pnl_context *ctx, int which, int panel, int nkeys, int M, int rows, const double *nodes,
                               const double *w, const double *phi0, const double *phi1) {
    if (!ctx) return PNL_ERR_INVALID;
    if (!ctx->have_pw || !ctx->have_dofs) return fail(ctx, PNL_ERR_STATE, "set the order function and the DoF map first");
    const int slot = -panel-1, dim = ctx->dim, nV = dim+1, dpe = ctx->dpe;
    const int nslots = which == PNL_INTERIOR ? nV : dim;
    if (which < 0 || which > 1 || slot < 0 || slot >= nslots || nkeys <= 0 || M <= 0 || !nodes || !w || !phi0 ||
        (which == PNL_INTERIOR && !phi1))
        return fail(ctx, PNL_ERR_INVALID, "bad pointwise-rule arguments");
    int rc;
    if (which == PNL_INTERIOR) {
        const int common = slot+1;
        const int expect = common == nV ? dpe : (common == 1 ? 2*dpe-ctx->dpv : 2*dpe-2*ctx->dpv-ctx->dped);
        if (rows != expect) return fail(ctx, PNL_ERR_INVALID, "pointwise rule has %d rows, expected %d", rows, expect);
        if ((rc = upload(ctx, ctx->b_pw_rule[0][slot][0], nodes, (size_t)nkeys*2*nV*M))) return rc;
        if ((rc = upload(ctx, ctx->b_pw_rule[0][slot][1], w, (size_t)nkeys*M))) return rc;
        if ((rc = upload(ctx, ctx->b_pw_rule[0][slot][2], phi0, (size_t)nkeys*rows*M))) return rc;
        if ((rc = upload(ctx, ctx->b_pw_rule[0][slot][3], phi1, (size_t)nkeys*rows*M))) return rc;
        ctx->pw.M[slot] = M; ctx->pw.rows[slot] = rows;
        ctx->pw.nodes[slot] = (const double*)ctx->b_pw_rule[0][slot][0].p; ctx->pw.w[slot] = (const double*)ctx->b_pw_rule[0][slot][1].p;
        ctx->pw.phi0[slot] = (const double*)ctx->b_pw_rule[0][slot][2].p; ctx->pw.phi1[slot] = (const double*)ctx->b_pw_rule[0][slot][3].p;
    } else {
        if (rows != dpe) return fail(ctx, PNL_ERR_INVALID, "pointwise boundary rule has %d rows, expected %d", rows, dpe);
        if ((rc = upload(ctx, ctx->b_pw_rule[1][slot][0], nodes, (size_t)nkeys*(nV+dim)*M))) return rc;
        if ((rc = upload(ctx, ctx->b_pw_rule[1][slot][1], w, (size_t)nkeys*M))) return rc;
        if ((rc = upload(ctx, ctx->b_pw_rule[1][slot][2], phi0, (size_t)nkeys*rows*M))) return rc;
        ctx->pw.bM[slot] = M;
        ctx->pw.bnodes[slot] = (const double*)ctx->b_pw_rule[1][slot][0].p; ctx->pw.bw[slot] = (const double*)ctx->b_pw_rule[1][slot][1].p;
        ctx->pw.bphi[slot] = (const double*)ctx->b_pw_rule[1][slot][2].p;
    }
    ctx->pw_nkeys[which] = nkeys;
    ctx->have_pw_rules[which][slot] = true;
    return PNL_OK;
}


}  // extern "C"

// what every assembly with an order per quadrature point needs before its first launch (pnl_assemble_dense_pointwise here, the
// near-field entry points in pnl_pwnear.hip): padded cell tables, the per-cell / per-facet largest orders and the vertex values of
// a P1 order function on the device, the distant rules in the problem description
int pnl_pw_prepare(pnl_context *ctx, int need_boundary) {
    int rc;
    if (!ctx->have_pw || !ctx->have_rules) return fail(ctx, PNL_ERR_STATE, "order function and distant rules must be set before assembling");
    if ((rc = finalize(ctx))) return rc;
    if (ctx->pw.type == 5) {
        if ((int)ctx->pw_vertex_s.size() != ctx->nv) return fail(ctx, PNL_ERR_STATE, "order function of type 5 without vertex values");
        const int nV = ctx->dim+1;
        std::vector<double> sv((size_t)nV*ctx->ncp, 0.);
        for (int c = 0; c < ctx->nc; c++)
            for (int k = 0; k < nV; k++) sv[(size_t)k*ctx->ncp+c] = ctx->pw_vertex_s[ctx->cells[(size_t)c*nV+k]];
        if ((rc = upload(ctx, ctx->b_pw_cellsv, sv.data(), sv.size()))) return rc;
        ctx->pw.cell_sv = (const double*)ctx->b_pw_cellsv.p; ctx->pw.sv_stride = ctx->ncp;
    }
    if (!(ctx->dpe == ctx->dim+1 || (ctx->dim == 2 && ctx->dpe == 6) || (ctx->dim == 1 && ctx->dpe == 3)))
        return fail(ctx, PNL_ERR_UNSUPPORTED, "pointwise variable orders: P1 and P2 elements");
    for (int s = 0; s <= ctx->dim; s++)
        if (!ctx->have_pw_rules[0][s]) return fail(ctx, PNL_ERR_STATE, "pointwise rule for %d common vertices not uploaded", s+1);
    if (need_boundary) {
        if (!ctx->have_boundary || (int)ctx->pw_facet_smax.size() != ctx->nb)
            return fail(ctx, PNL_ERR_STATE, "the boundary term needs boundary facets and their orders");
        for (int s = 0; s < ctx->dim; s++)
            if (!ctx->have_pw_rules[1][s]) return fail(ctx, PNL_ERR_STATE, "pointwise boundary rule for %d common vertices not uploaded", s+1);
    }
    std::vector<double> sm(ctx->ncp, 0.);
    std::copy(ctx->pw_cell_smax.begin(), ctx->pw_cell_smax.end(), sm.begin());
    if ((rc = upload(ctx, ctx->b_pw_csm, sm.data(), sm.size()))) return rc;
    if ((rc = upload(ctx, ctx->b_pw_fsm, ctx->pw_facet_smax.data(), ctx->pw_facet_smax.size()))) return rc;
    ctx->pw.cell_smax = (const double*)ctx->b_pw_csm.p; ctx->pw.facet_smax = (const double*)ctx->b_pw_fsm.p;
    DevProblem &P = ctx->P;
    P.qmax = ctx->qmax;
    P.off = (const int*)ctx->b_off.p; P.bary = (const double*)ctx->b_bary.p; P.w = (const double*)ctx->b_w.p;
    P.phi = (const double*)ctx->b_phi.p; P.foff = (const int*)ctx->b_foff.p; P.fbary = (const double*)ctx->b_fbary.p;
    P.fw = (const double*)ctx->b_fw.p;
    P.cur_class = -1;
    return PNL_OK;
}

extern "C" {

int pnl_assemble_dense_pointwise(pnl_context *ctx, double *A, int64_t ldA, int zero_exterior, int cell_begin, int cell_end,
                                 int npairs, const int32_t *pairs, int nbpairs, const int32_t *bpairs) {
    if (!ctx) return PNL_ERR_INVALID;
    int rc;
    if ((rc = pnl_pw_prepare(ctx, zero_exterior))) return rc;
    if (!A || ldA < ctx->N) return fail(ctx, PNL_ERR_INVALID, "bad output matrix (ldA=%lld, num_dofs=%d)", (long long)ldA, ctx->N);
    if (cell_begin < 0 || cell_end > ctx->nc || cell_begin > cell_end) return fail(ctx, PNL_ERR_INVALID, "bad cell range");
    if (npairs < 0 || nbpairs < 0 || (npairs && !pairs) || (nbpairs && !bpairs)) return fail(ctx, PNL_ERR_INVALID, "bad pair lists");
    for (int t = 0; t < npairs; t++) {
        const int32_t *q = pairs+4*(size_t)t;
        if (q[0] < 0 || q[1] < q[0] || q[1] >= ctx->nc || q[2] < 1 || q[2] > ctx->dim+1 || q[3] < 0 || q[3] >= ctx->pw_nkeys[0])
            return fail(ctx, PNL_ERR_INVALID, "bad touching pair %d", t);
    }
    for (int t = 0; t < nbpairs; t++) {
        const int32_t *q = bpairs+4*(size_t)t;
        if (q[0] < 0 || q[0] >= ctx->nc || q[1] < 0 || q[1] >= ctx->nb || q[2] < 1 || q[2] > ctx->dim || q[3] < 0 || q[3] >= ctx->pw_nkeys[1])
            return fail(ctx, PNL_ERR_INVALID, "bad touching cell/facet pair %d", t);
    }
    if ((rc = upload(ctx, ctx->b_pw_pairs, pairs, (size_t)4*npairs))) return rc;
    if ((rc = upload(ctx, ctx->b_pw_bpairs, bpairs, (size_t)4*nbpairs))) return rc;
    unsigned long long visited = 0;
    for (long long c = cell_begin; c < cell_end; c++) visited += (unsigned long long)(ctx->nc-c);
    ctx->visited_pairs = visited; ctx->visited_is_assembled = false;
    if (ctx->dim == 2)
        return ctx->dpe == 6 ? pointwise_impl<2, 6>(ctx, A, ldA, zero_exterior, cell_begin, cell_end, npairs, nbpairs)
                             : pointwise_impl<2, 3>(ctx, A, ldA, zero_exterior, cell_begin, cell_end, npairs, nbpairs);
    return ctx->dpe == 3 ? pointwise_impl<1, 3>(ctx, A, ldA, zero_exterior, cell_begin, cell_end, npairs, nbpairs)
                         : pointwise_impl<1, 2>(ctx, A, ldA, zero_exterior, cell_begin, cell_end, npairs, nbpairs);
}

int pnl_get_counters(pnl_context *ctx, int64_t *out, int n) {
    if (!ctx || !out || n <= 0) return PNL_ERR_INVALID;
    if (!ctx->b_counters.p) return fail(ctx, PNL_ERR_STATE, "nothing assembled yet");
    unsigned long long tmp[PNL_NCOUNTERS];
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    HIPCHK(ctx, hipMemcpy(tmp, ctx->b_counters.p, sizeof(tmp), hipMemcpyDeviceToHost));
    tmp[0] = ctx->visited_is_assembled ? tmp[1] : ctx->visited_pairs;
    for (int i = 0; i < n && i < PNL_NCOUNTERS; i++) out[i] = (int64_t)tmp[i];
    return check_overflow(ctx);
}

int pnl_get_phase_ms(pnl_context *ctx, float *out, int n) {
    if (!ctx || !out || n <= 0) return PNL_ERR_INVALID;
    if (!ctx->ev_valid) return fail(ctx, PNL_ERR_STATE, "nothing assembled yet");
    HIPCHK(ctx, hipEventSynchronize(ctx->ev[5]));
    float t[7] = {0, 0, 0, 0, 0, 0, 0}, tmp;
    if (ctx->tiles_launched) {
        HIPCHK(ctx, hipEventElapsedTime(&t[0], ctx->ev[0], ctx->ev[6]));  // tile kernels (uniform + general; last class)
        HIPCHK(ctx, hipEventElapsedTime(&t[1], ctx->ev[6], ctx->ev[1]));  // work-list kernels
        if (ctx->pure_launched && ctx->cls.size() == 1) {
            HIPCHK(ctx, hipEventElapsedTime(&t[6], ctx->ev[0], ctx->ev[7]));   // uniform-tile kernel alone
            t[0] -= t[6];
        }
    }
    HIPCHK(ctx, hipEventElapsedTime(&tmp, ctx->ev[1], ctx->ev[2]));       // mirror
    HIPCHK(ctx, hipEventElapsedTime(&t[2], ctx->ev[2], ctx->ev[3]));      // singular
    HIPCHK(ctx, hipEventElapsedTime(&t[3], ctx->ev[3], ctx->ev[4]));      // boundary
    HIPCHK(ctx, hipEventElapsedTime(&t[4], ctx->ev[4], ctx->ev[5]));      // diagonal scatter
    t[4] += tmp;
    HIPCHK(ctx, hipEventElapsedTime(&t[5], ctx->ev[0], ctx->ev[5]));
    for (int i = 0; i < n && i < 7; i++) out[i] = t[i];
    return PNL_OK;
}

int pnl_get_kernel_ms(pnl_context *ctx, float *out, int n) {
    if (!ctx || !out || n <= 0) return PNL_ERR_INVALID;
    if (!ctx->ev_valid) return fail(ctx, PNL_ERR_STATE, "nothing assembled yet");
    HIPCHK(ctx, hipEventSynchronize(ctx->ev[5]));
    for (int s = 0; s < n && s < PNL_NUM_KERNEL_SLOTS; s++) {
        out[s] = 0.f;
        if (ctx->kev_set[s]) HIPCHK(ctx, hipEventElapsedTime(&out[s], ctx->kev[s][0], ctx->kev[s][1]));
    }
    return PNL_OK;
}

int pnl_gemv(pnl_context *ctx, const double *A, int64_t ldA, int n, const double *x, double *y, int symmetric_half) {
    if (!ctx || !A || !x || !y || n <= 0 || ldA < n) return fail(ctx, PNL_ERR_INVALID, "bad gemv arguments");
    // 2: A is stored in full and is symmetric -- its upper triangle is read once for both A x and A^T x (4 n^2 bytes, pnl_gemv2.hip)
    if (symmetric_half == 2) return pnl_launch_gemv_symmetric(ctx, A, (long long)ldA, n, x, 1., 0., nullptr, y);
    hipLaunchKernelGGL(k_gemv, dim3((n+3)/4), dim3(PNL_NTHREADS), 0, ctx->stream, A, (long long)ldA, n, x, y);
    HIPCHK(ctx, hipGetLastError());
    if (symmetric_half) {
        const int rows = 128;
        hipLaunchKernelGGL(k_gemv_t_add, dim3((n+PNL_NTHREADS-1)/PNL_NTHREADS, (n+rows-1)/rows), dim3(PNL_NTHREADS), 0, ctx->stream,
                           A, (long long)ldA, n, x, y, rows);
        HIPCHK(ctx, hipGetLastError());
    }
    return PNL_OK;
}

int pnl_cg_jacobi(pnl_context *ctx, const double *A, int64_t ldA, int n, const double *b, double *x, double tol, int maxiter,
                  int *iters, double *residual) {
    if (!ctx || !A || !b || !x || n <= 0 || ldA < n || maxiter < 0) return fail(ctx, PNL_ERR_INVALID, "bad cg arguments");
    int rc;
    for (int i = 0; i < 5; i++)
        if ((rc = ensure(ctx, ctx->b_vec[i], sizeof(double)*n))) return rc;
    if ((rc = ensure(ctx, ctx->b_scal, sizeof(double)*4))) return rc;
    double *r = (double*)ctx->b_vec[0].p, *p = (double*)ctx->b_vec[1].p, *Ap = (double*)ctx->b_vec[2].p,
           *z = (double*)ctx->b_vec[3].p, *dinv = (double*)ctx->b_vec[4].p, *scal = (double*)ctx->b_scal.p;
    const int gv = (n+PNL_NTHREADS-1)/PNL_NTHREADS, gd = std::min(gv, 1024);
    hipStream_t st = ctx->stream;
    auto dot = [&](const double *u, const double *v, int slot) {
        (void)hipMemsetAsync(scal+slot, 0, sizeof(double), st);
        hipLaunchKernelGGL(k_dot, dim3(gd), dim3(PNL_NTHREADS), 0, st, u, v, n, scal+slot);
    };
    double hs[4];
    // solvers.pyx:363-444 with the Jacobi preconditioner (:229-245); convergence in the preconditioner norm
    // CG needs a symmetric operator: every product reads the upper triangle only (pnl_gemv2.hip; 4 n^2 bytes instead of 8 n^2)
    auto gemv = [&](const double *v, double *out) { return pnl_launch_gemv_symmetric(ctx, A, (long long)ldA, n, v, 1., 0., nullptr, out); };
    hipLaunchKernelGGL(k_diag_inv, dim3(gv), dim3(PNL_NTHREADS), 0, st, A, (long long)ldA, n, dinv);
    if ((rc = gemv(x, Ap))) return rc;
    hipLaunchKernelGGL(k_cg_init, dim3(gv), dim3(PNL_NTHREADS), 0, st, b, (const double*)Ap, (const double*)dinv, n, r, p);
    dot(r, p, 0);                                                // betaOld = r . Br
    HIPCHK(ctx, hipMemcpyAsync(hs, scal, sizeof(double), hipMemcpyDeviceToHost, st));
    HIPCHK(ctx, hipStreamSynchronize(st));
    double conv = std::sqrt(hs[0]);
    int it = 0, k = 0;
    if (conv > tol) {
        for (it = 0; it < maxiter; it++) {
            if ((rc = gemv(p, Ap))) return rc;
            dot(p, Ap, 1);
            hipLaunchKernelGGL(k_cg_update, dim3(gv), dim3(PNL_NTHREADS), 0, st, (const double*)scal, (const double*)p,
                               (const double*)Ap, (const double*)dinv, n, x, r, z);
            if (k == 50) {
                // recalculate the residual to limit rounding drift (solvers.pyx:412-415)
                if ((rc = gemv(x, Ap))) return rc;
                hipLaunchKernelGGL(k_cg_init, dim3(gv), dim3(PNL_NTHREADS), 0, st, b, (const double*)Ap, (const double*)dinv, n, r, z);
                k = 0;
            }
            dot(r, z, 2);                                        // beta = r . Br
            HIPCHK(ctx, hipMemcpyAsync(hs, scal, 3*sizeof(double), hipMemcpyDeviceToHost, st));
            HIPCHK(ctx, hipStreamSynchronize(st));
            conv = std::sqrt(hs[2]);
            if (conv <= tol) break;
            hipLaunchKernelGGL(k_cg_dir, dim3(gv), dim3(PNL_NTHREADS), 0, st, (const double*)scal, (const double*)z, n, p);
            // betaOld = beta
            HIPCHK(ctx, hipMemcpyAsync(scal, scal+2, sizeof(double), hipMemcpyDeviceToDevice, st));
            k++;
        }
    }
    HIPCHK(ctx, hipGetLastError());
    if (iters) *iters = it;
    if (residual) *residual = conv;
    return PNL_OK;
}

}  // extern "C"
