// Host-side planning of the H2 / near-field assembly in C++ (no device code): cluster tree, admissibility recursion, the cells
// of the cluster nodes, the tile / chunk work lists of the tiled near-field assembly and the transfer matrices of the far field.
//
// Reference (clusterMethodCy.pyx = CM, nonlocalAssembly_{SCALAR}.pxi = NA, nonlocalAssembly.pyx):
//   tree_node.refine                 CM:354-663     (here: the MEDIAN split along the longest box edge, like clusters.py)
//   queryAdmissibility / getAdmissibleClusters  CM:4008-4136  (eta criterion, near / far recursion, merge of near-field children)
//   tree_node.cells                  NA:2887-2898   (cells touching the DoFs of a cluster)
//   nearFieldClusterPair.set_cells   nonlocalAssembly.pyx:374-392 (cellsUnion, cellsInter)
//   boundaryEdges                    nonlocalAssembly.pyx:540-578
//   transferMatrixBuilder            CM:2004-2073
// In the reference all of this is Python-object code in Cython; round 1 of this repo had it in numpy (clusters.py, h2.py:
// 0.8 s + 1.5 s + 0.2 s for getH2 at 49k DoFs).  The arrays that come out are the ones pnl_assemble_clusters_tiled and
// pnl_h2_setup take; clusters.py wraps them in the tree_node / nearFieldClusterPair objects the callers know.
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstring>
#include <numeric>
#include <vector>
#include <chrono>
#include <thread>
#include <sched.h>
#include <cstdio>
#include <cstdlib>
#include "pnl_hip.h"

extern "C++" const char *pnl_tune(const char *name);     // pnl_hip.hip: options set through pnl_set_option

#include "pnl_plan.h"

namespace {

int block_of_range(const pnl_tree *T, int beg, int end) {
    if (T->dof_block.empty()) return 0;
    const int b = T->dof_block[T->perm[beg]];
    for (int t = beg+1; t < end; t++) if (T->dof_block[T->perm[t]] != b) return -1;
    return b;
}

double dist_boxes(const PNode &a, const PNode &b, int dim) { return pnl_dist_boxes(a.box, b.box, dim); }

double diam_box(const PNode &a, int dim) { return pnl_diam_box(a.box, dim); }

void set_box(pnl_tree *T, PNode &n) {
    for (int d = 0; d < T->dim; d++) { n.box[d][0] = INFINITY; n.box[d][1] = -INFINITY; }
    for (int t = n.beg; t < n.end; t++) {
        const double *b = &T->boxes[(size_t)T->perm[t]*T->dim*2];
        for (int d = 0; d < T->dim; d++) { n.box[d][0] = std::min(n.box[d][0], b[2*d]); n.box[d][1] = std::max(n.box[d][1], b[2*d+1]); }
    }
}

// median split of node k (clusters.tree_node.refine): x < median left, x >= median right, both keep the ascending DoF order
void refine(pnl_tree *T, int k, int minSize, int maxLevels, std::vector<double> &xs, std::vector<int32_t> &tmp) {
    const PNode nd = T->nodes[k];
    const int n = nd.end-nd.beg;
    if (nd.block < 0) {
        // DoFs of several kernel blocks: split off the lowest block (the reference hangs the blocks below the node, NA:2619-2640;
        // here a chain of binary nodes), whatever the size and the depth
        int bmin = T->dof_block[T->perm[nd.beg]];
        for (int t = nd.beg; t < nd.end; t++) bmin = std::min(bmin, T->dof_block[T->perm[t]]);
        tmp.resize(n);
        int nl = 0;
        for (int t = 0; t < n; t++) if (T->dof_block[T->perm[nd.beg+t]] == bmin) tmp[nl++] = T->perm[nd.beg+t];
        int nr = nl;
        for (int t = 0; t < n; t++) if (T->dof_block[T->perm[nd.beg+t]] != bmin) tmp[nr++] = T->perm[nd.beg+t];
        std::copy(tmp.begin(), tmp.begin()+n, T->perm.begin()+nd.beg);
        for (int c = 0; c < 2; c++) {
            PNode ch;
            ch.beg = c ? nd.beg+nl : nd.beg; ch.end = c ? nd.end : nd.beg+nl;
            ch.parent = k; ch.child[0] = ch.child[1] = -1; ch.level = nd.level+1;
            ch.block = block_of_range(T, ch.beg, ch.end);
            set_box(T, ch);
            T->nodes[k].child[c] = (int)T->nodes.size();
            T->nodes.push_back(ch);
        }
        return;
    }
    if (nd.level+1 >= maxLevels || n <= minSize) return;
    int ax = 0;
    double best = -1.;
    for (int d = 0; d < T->dim; d++) { const double e = nd.box[d][1]-nd.box[d][0]; if (e > best) { best = e; ax = d; } }
    xs.resize(n);
    for (int t = 0; t < n; t++) xs[t] = T->coords[(size_t)T->perm[nd.beg+t]*T->dim+ax];
    double med;
    if (T->ref_type == 1) med = 0.5*(nd.box[ax][0]+nd.box[ax][1]);          // GEOMETRIC: the box is halved (CM:388-390, 606-607)
    else if (T->ref_type == 2) {                                              // BARYCENTER: the mean of the DoF coordinates (CM:391-398)
        double sum = 0.;
        for (int t = 0; t < n; t++) sum += xs[t];
        med = sum/n;
    } else {
        std::vector<double> srt(xs);
        if (n & 1) { std::nth_element(srt.begin(), srt.begin()+n/2, srt.end()); med = srt[n/2]; }
        else {
            std::nth_element(srt.begin(), srt.begin()+n/2, srt.end());
            const double hi = srt[n/2];
            const double lo = *std::max_element(srt.begin(), srt.begin()+n/2);
            med = (lo+hi)/2.;                    // numpy.median of an even count: mean of the two middle values
        }
    }
    tmp.resize(n);
    int nl = 0;
    for (int t = 0; t < n; t++) if (xs[t] < med) tmp[nl++] = T->perm[nd.beg+t];
    int nr = nl;
    for (int t = 0; t < n; t++) if (!(xs[t] < med)) tmp[nr++] = T->perm[nd.beg+t];
    const int nright = n-nl;
    if (nl < minSize || nright < minSize || nl == n || nright == n) return;
    std::copy(tmp.begin(), tmp.begin()+n, T->perm.begin()+nd.beg);
    for (int c = 0; c < 2; c++) {
        PNode ch;
        ch.beg = c ? nd.beg+nl : nd.beg; ch.end = c ? nd.end : nd.beg+nl;
        ch.parent = k; ch.child[0] = ch.child[1] = -1; ch.level = nd.level+1;
        ch.block = nd.block;
        set_box(T, ch);
        T->nodes[k].child[c] = (int)T->nodes.size();
        T->nodes.push_back(ch);
    }
}

// interactionDomain.maxDistBoxes (interactionDomains.pyx:325-337), as written there
double max_dist_boxes(const PNode &a, const PNode &b, int dim) {
    double s = 0.;
    for (int d = 0; d < dim; d++) {
        const bool first = a.box[d][0] > b.box[d][0];
        const double b1 = first ? b.box[d][0] : a.box[d][0], a2 = first ? a.box[d][1] : b.box[d][1];
        const double e = std::max(a2-b1, 0.);
        s += e*e;
    }
    return std::sqrt(s);
}

// horizon < inf (CM:4074-4090, 4115, 4131-4135): pairs of clusters farther apart than the horizon do not interact (reported as
// "far field added" so that they are not merged into a near-field block), pairs the horizon may cut stay in the near field, and
// near-field children are merged into one block only if the block fits into the horizon
bool admissible_rec(pnl_tree *T, int n1, int n2, double eta, int maxLevels, int level, double horizon) {
    const PNode &a = T->nodes[n1], &b = T->nodes[n2];
    const double dist = dist_boxes(a, b, T->dim);
    // clusters of one kernel block each; the interface block stays in the near field (mixed_node, CM:4038)
    const bool pure = a.block >= 0 && b.block >= 0 && a.block != T->mixed_block && b.block != T->mixed_block;
    bool seems = pure && eta*dist >= std::max(diam_box(a, T->dim), diam_box(b, T->dim));
    const bool finite = horizon < INFINITY;
    double diamUnion = 0.;
    if (finite) {
        if (dist > horizon) return true;
        if (horizon <= max_dist_boxes(a, b, T->dim)) seems = false;
        double s = 0.;
        for (int d = 0; d < T->dim; d++) {
            const double e = std::max(a.box[d][1], b.box[d][1])-std::min(a.box[d][0], b.box[d][0]);
            s += e*e;
        }
        diamUnion = std::sqrt(s);
    }
    if (seems) {
        T->far.push_back(n1); T->far.push_back(n2); T->far.push_back(level);
        return true;
    }
    const size_t lenNear = T->near.size();
    const bool leaf1 = a.child[0] < 0, leaf2 = b.child[0] < 0;
    if ((leaf1 && leaf2) || level == maxLevels) {
        T->near.push_back(n1); T->near.push_back(n2);
        return false;
    }
    bool added = false;
    const int c1[2] = {a.child[0], a.child[1]}, c2[2] = {b.child[0], b.child[1]};
    if (leaf1) { for (int j = 0; j < 2; j++) added |= admissible_rec(T, n1, c2[j], eta, maxLevels, level+1, horizon); }
    else if (leaf2) { for (int i = 0; i < 2; i++) added |= admissible_rec(T, c1[i], n2, eta, maxLevels, level+1, horizon); }
    else
        for (int i = 0; i < 2; i++) for (int j = 0; j < 2; j++) added |= admissible_rec(T, c1[i], c2[j], eta, maxLevels, level+1, horizon);
    if (!added && (!finite || diamUnion < horizon)) {
        // no far-field pair below: keep the whole block as one near-field pair (CM:4131-4135)
        T->near.resize(lenNear);
        T->near.push_back(n1); T->near.push_back(n2);
    }
    return added;
}

}  // namespace

extern "C" {

int pnl_tree_build(int N, int dim, const double *boxes, const int64_t *d2c_ptr, const int32_t *d2c_idx, int nc, double eta,
                   int min_size, int max_levels, int do_admissibility, pnl_tree **out) {
    return pnl_tree_build_blocks(N, dim, boxes, d2c_ptr, d2c_idx, nc, eta, min_size, max_levels, do_admissibility, nullptr, -1, out);
}

int pnl_tree_build_blocks(int N, int dim, const double *boxes, const int64_t *d2c_ptr, const int32_t *d2c_idx, int nc, double eta,
                          int min_size, int max_levels, int do_admissibility, const int32_t *dof_block, int mixed_block,
                          pnl_tree **out) {
    return pnl_tree_build_refined(N, dim, boxes, d2c_ptr, d2c_idx, nc, eta, min_size, max_levels, do_admissibility, dof_block, mixed_block, 0, out);
}

int pnl_tree_build_refined(int N, int dim, const double *boxes, const int64_t *d2c_ptr, const int32_t *d2c_idx, int nc, double eta,
                           int min_size, int max_levels, int do_admissibility, const int32_t *dof_block, int mixed_block, int ref_type,
                           pnl_tree **out) {
    return pnl_tree_build_horizon(N, dim, boxes, d2c_ptr, d2c_idx, nc, eta, min_size, max_levels, do_admissibility, dof_block, mixed_block,
                                  ref_type, INFINITY, out);
}

int pnl_tree_build_horizon(int N, int dim, const double *boxes, const int64_t *d2c_ptr, const int32_t *d2c_idx, int nc, double eta,
                           int min_size, int max_levels, int do_admissibility, const int32_t *dof_block, int mixed_block, int ref_type,
                           double horizon, pnl_tree **out) {
    if (!(horizon > 0.)) return PNL_ERR_INVALID;
    if (!out || N <= 0 || dim < 1 || dim > 3 || !boxes || !d2c_ptr || !d2c_idx || ref_type < 0 || ref_type > 2) return PNL_ERR_INVALID;
    if (dof_block) for (int i = 0; i < N; i++) if (dof_block[i] < 0) return PNL_ERR_INVALID;
    pnl_tree *T = new pnl_tree();
    T->N = N; T->dim = dim; T->nc = nc; T->ref_type = ref_type;
    if (dof_block) { T->dof_block.assign(dof_block, dof_block+N); T->mixed_block = mixed_block; }
    T->boxes.assign(boxes, boxes+(size_t)N*dim*2);
    T->coords.resize((size_t)N*dim);
    for (int i = 0; i < N; i++) for (int d = 0; d < dim; d++) T->coords[(size_t)i*dim+d] = (boxes[((size_t)i*dim+d)*2]+boxes[((size_t)i*dim+d)*2+1])/2.;
    T->d2c_ptr.assign(d2c_ptr, d2c_ptr+N+1);
    T->d2c_idx.assign(d2c_idx, d2c_idx+d2c_ptr[N]);
    T->perm.resize(N);
    std::iota(T->perm.begin(), T->perm.end(), 0);
    PNode root;
    root.beg = 0; root.end = N; root.parent = -1; root.child[0] = root.child[1] = -1; root.level = 0;
    root.block = block_of_range(T, 0, N);
    set_box(T, root);
    T->nodes.push_back(root);
    // the recursion from (root, root) reaches every node through its diagonal pair, which is never admissible: the tree
    // is refined completely (breadth first here; the node order differs from the lazy Python version, the tree does not)
    std::vector<double> xs;
    std::vector<int32_t> tmp;
    if (do_admissibility >= 0)
        for (size_t k = 0; k < T->nodes.size(); k++) refine(T, (int)k, min_size, max_levels, xs, tmp);
    if (do_admissibility > 0) admissible_rec(T, 0, 0, eta, max_levels, 0, horizon);
    *out = T;
    return PNL_OK;
}

// the same tree and lists, refined and classified on the device (pnl_plan_dev.hip): MEDIAN / GEOMETRIC refinement of a constant-order
// kernel's DoFs (kernel blocks and the BARYCENTER split stay with the host loops above: PNL_ERR_UNSUPPORTED)
int pnl_tree_build_device(int N, int dim, const double *boxes, const int64_t *d2c_ptr, const int32_t *d2c_idx, int nc, double eta,
                          int min_size, int max_levels, int do_admissibility, int ref_type, pnl_tree **out) {
    if (!out || N <= 0 || dim < 1 || dim > 3 || !boxes || !d2c_ptr || !d2c_idx || ref_type < 0 || ref_type > 2) return PNL_ERR_INVALID;
    if (ref_type == 2) return PNL_ERR_UNSUPPORTED;
    pnl_tree *T = new pnl_tree();
    T->N = N; T->dim = dim; T->nc = nc; T->ref_type = ref_type;
    T->boxes.assign(boxes, boxes+(size_t)N*dim*2);
    T->coords.resize((size_t)N*dim);
    for (int i = 0; i < N; i++) for (int d = 0; d < dim; d++) T->coords[(size_t)i*dim+d] = (boxes[((size_t)i*dim+d)*2]+boxes[((size_t)i*dim+d)*2+1])/2.;
    T->d2c_ptr.assign(d2c_ptr, d2c_ptr+N+1);
    T->d2c_idx.assign(d2c_idx, d2c_idx+d2c_ptr[N]);
    T->perm.resize(N);
    std::iota(T->perm.begin(), T->perm.end(), 0);
    PNode root;
    root.beg = 0; root.end = N; root.parent = -1; root.child[0] = root.child[1] = -1; root.level = 0;
    root.block = 0;
    set_box(T, root);
    T->nodes.push_back(root);
    const int rc = pnl_tree_fill_device(T, eta, min_size, max_levels, do_admissibility);
    if (rc) { delete T; return rc; }
    *out = T;
    return PNL_OK;
}

void pnl_tree_destroy(pnl_tree *T) { delete T; }

// sizes: [0] nodes, [1] near pairs, [2] far pairs
int pnl_tree_sizes(const pnl_tree *T, int64_t *out) {
    if (!T || !out) return PNL_ERR_INVALID;
    out[0] = (int64_t)T->nodes.size(); out[1] = (int64_t)T->near.size()/2; out[2] = (int64_t)T->far.size()/3;
    return PNL_OK;
}

// node table: range[nn][2], parent[nn], children[nn][2], level[nn], box[nn][dim][2]; perm[N]; near[nnear][2]; far[nfar][3]
int pnl_tree_get(const pnl_tree *T, int32_t *range, int32_t *parent, int32_t *children, int32_t *level, double *box, int32_t *perm,
                 int32_t *near, int32_t *far) {
    if (!T) return PNL_ERR_INVALID;
    const size_t nn = T->nodes.size();
    for (size_t k = 0; k < nn; k++) {
        const PNode &n = T->nodes[k];
        if (range) { range[2*k] = n.beg; range[2*k+1] = n.end; }
        if (parent) parent[k] = n.parent;
        if (children) { children[2*k] = n.child[0]; children[2*k+1] = n.child[1]; }
        if (level) level[k] = n.level;
        if (box) for (int d = 0; d < T->dim; d++) { box[(k*T->dim+d)*2] = n.box[d][0]; box[(k*T->dim+d)*2+1] = n.box[d][1]; }
    }
    if (perm) std::copy(T->perm.begin(), T->perm.end(), perm);
    if (near) std::copy(T->near.begin(), T->near.end(), near);
    if (far) std::copy(T->far.begin(), T->far.end(), far);
    return PNL_OK;
}

// cells touching the DoFs of the given nodes, sorted ascending, as CSR (off[n+1], then cells); two-call protocol: cells == NULL
// returns the total in off[n]
int pnl_tree_node_cells(const pnl_tree *T, int n, const int32_t *node_ids, int64_t *off, int32_t *cells) {
    if (!T || n < 0 || (n && !node_ids) || !off) return PNL_ERR_INVALID;
    std::vector<char> mark(T->nc, 0);
    std::vector<int32_t> buf;
    off[0] = 0;
    for (int i = 0; i < n; i++) {
        const int k = node_ids[i];
        if (k < 0 || k >= (int)T->nodes.size()) return PNL_ERR_INVALID;
        const PNode &nd = T->nodes[k];
        buf.clear();
        for (int t = nd.beg; t < nd.end; t++) {
            const int I = T->perm[t];
            for (int64_t p = T->d2c_ptr[I]; p < T->d2c_ptr[I+1]; p++) {
                const int c = T->d2c_idx[p];
                if (!mark[c]) { mark[c] = 1; buf.push_back(c); }
            }
        }
        for (int c : buf) mark[c] = 0;
        if (cells) {
            std::sort(buf.begin(), buf.end());
            std::copy(buf.begin(), buf.end(), cells+off[i]);
        }
        off[i+1] = off[i]+(int64_t)buf.size();
    }
    return PNL_OK;
}

// transfer matrices T[k][I][J] = L^parent_I(xi^child_J) of every non-root node (CM:2010-2073), tensor index i_0 + m i_1
int pnl_h2_transfer_matrices(int nnodes, int dim, int m, const double *box, const int32_t *parent, double *out) {
    if (nnodes <= 0 || dim < 1 || dim > 2 || m < 1 || m > 32 || !box || !parent || !out) return PNL_ERR_INVALID;
    const int M = dim == 1 ? m : m*m;
    std::vector<double> eta(m), L((size_t)2*m*m);
    for (int j = 0; j < m; j++) eta[j] = std::cos((2.0*(m-j)-1.0)/(2.0*m)*3.14159265358979323846);
    for (int k = 0; k < nnodes; k++) {
        double *Tk = out+(size_t)k*M*M;
        if (parent[k] < 0) { std::fill(Tk, Tk+(size_t)M*M, 0.); continue; }
        for (int d = 0; d < dim; d++) {
            const double pa = box[((size_t)parent[k]*dim+d)*2], pb = box[((size_t)parent[k]*dim+d)*2+1];
            const double ca = box[((size_t)k*dim+d)*2], cb = box[((size_t)k*dim+d)*2+1];
            // Ld[l][j] = l-th Lagrange polynomial on the parent's nodes at the child's j-th node
            for (int l = 0; l < m; l++)
                for (int j = 0; j < m; j++) {
                    const double x = (cb-ca)*0.5*(eta[j]+1.0)+ca;
                    double v = 1.;
                    const double xl = (pb-pa)*0.5*(eta[l]+1.0)+pa;
                    for (int q = 0; q < m; q++)
                        if (q != l) { const double xq = (pb-pa)*0.5*(eta[q]+1.0)+pa; v *= (x-xq)/(xl-xq); }
                    L[(size_t)d*m*m+l*m+j] = v;
                }
        }
        if (dim == 1) std::copy(L.begin(), L.begin()+(size_t)m*m, Tk);
        else
            for (int i1 = 0; i1 < m; i1++) for (int i0 = 0; i0 < m; i0++)
                for (int j1 = 0; j1 < m; j1++) for (int j0 = 0; j0 < m; j0++)
                    Tk[(size_t)(i0+m*i1)*M+(j0+m*j1)] = L[(size_t)i0*m+j0]*L[(size_t)m*m+i1*m+j1];
    }
    return PNL_OK;
}

}  // extern "C"

// threads of the planning loops: the host cores this process may use, at most 16 (pnl_set_option("PNL_PLAN_THREADS", n) overrides)
static int plan_threads() {
    if (const char *e = pnl_tune("PNL_PLAN_THREADS")) return std::max(1, atoi(e));
    int n = (int)std::thread::hardware_concurrency();
    cpu_set_t set;
    if (sched_getaffinity(0, sizeof(set), &set) == 0) n = std::min(n > 0 ? n : 1, CPU_COUNT(&set));
    return std::max(1, std::min(n, 16));
}

// f(thread, begin, end) over contiguous index ranges of about equal weight; the results of thread t precede those of t + 1
template <class F>
static void par_ranges(int n, const std::vector<double> &w, int nthreads, F f) {
    std::vector<int> cut(nthreads+1, n);
    cut[0] = 0;
    double total = 0.;
    for (int i = 0; i < n; i++) total += w[i];
    double run = 0.;
    int t = 1;
    for (int i = 0; i < n && t < nthreads; i++) {
        run += w[i];
        while (t < nthreads && run >= total*t/nthreads) cut[t++] = i+1;
    }
    if (nthreads == 1 || n < 2*nthreads) { for (int k = 0; k < nthreads; k++) f(k, cut[k], cut[k+1]); return; }
    std::vector<std::thread> th;
    for (int k = 1; k < nthreads; k++) th.emplace_back([&, k] { f(k, cut[k], cut[k+1]); });
    f(0, cut[0], cut[1]);
    for (auto &x : th) x.join();
}

// =====================================================================================================================
// Work lists of the tiled near-field assembly (clusters.nearFieldPlan; the GPU's decomposition of assembleClusters,
// NA:1663-1964): chunks of <= tile Morton-ordered cells per node whose DoFs inside the node fit the LDS budget, tiles =
// chunk pairs of every unordered cluster pair, slots of the per-(pair, cell) diagonal-block buffer, touching element pairs
// per cluster pair, boundary facets of cellsUnion and the touching (cell, facet) pairs.
struct pnl_nfplan {
    int tile = 64, dpe = 0, dim = 0, nU = 1;
    std::vector<int32_t> node_chunk_off, chunk_cells, chunk_ndof, chunk_dofs;     // chunk_dofs: [nchunks][nU]
    std::vector<int16_t> chunk_slot;                                                  // [nchunks][dpe][tile]
    std::vector<int32_t> tile_chunkA, tile_chunkB, tile_pair, tile_flags, tile_dslotA, tile_dslotB;
    std::vector<int32_t> sing[3];                                                     // [n][3] (pair, c1, c2)
    std::vector<int32_t> d_cell, d_pair, pair_foff, fvid, bt_slot, bt_cell, bt_facet;
    int64_t num_dslots = 0;
};

extern "C" {

int pnl_nfplan_build(int dim, int nv, const double *vertices, int nc, const int32_t *cells, int dpe, int N, const int32_t *dofs,
                     int nnodes, const int64_t *node_off, const int32_t *node_dofs, const int64_t *node_cell_off,
                     const int32_t *node_cells, int npairs, const int32_t *pair_nodes, int tile, int max_chunk_dofs,
                     pnl_nfplan **out) {
    if (!out || dim < 1 || dim > 2 || !vertices || !cells || !dofs || nnodes < 0 || npairs < 0 || tile <= 0 || tile > 64)
        return PNL_ERR_INVALID;
    const int nV = dim+1;
    pnl_nfplan *P = new pnl_nfplan();
    P->tile = tile; P->dpe = dpe; P->dim = dim;
    const bool timing = pnl_tune("PNL_PLAN_TIMING") != nullptr;
    auto tnow = [] { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
    double tt[8] = {0}, t_last = tnow();
    auto lap = [&](int k) { const double t = tnow(); tt[k] += t-t_last; t_last = t; };
    // Morton code of the cell centres (16 bits per coordinate)
    std::vector<uint64_t> morton(nc, 0);
    {
        double lo[2] = {INFINITY, INFINITY}, hi[2] = {-INFINITY, -INFINITY};
        std::vector<double> cen((size_t)nc*dim);
        for (int c = 0; c < nc; c++)
            for (int d = 0; d < dim; d++) {
                double s = 0.;
                for (int k = 0; k < nV; k++) s += vertices[(size_t)cells[(size_t)c*nV+k]*dim+d];
                // numpy: mean over the vertices = sum / count
                s /= nV;
                cen[(size_t)c*dim+d] = s;
                lo[d] = std::min(lo[d], s); hi[d] = std::max(hi[d], s);
            }
        for (int c = 0; c < nc; c++) {
            uint64_t g[2] = {0, 0};
            for (int d = 0; d < dim; d++) {
                const double ext = std::max(hi[d]-lo[d], 1e-300);
                const double v = (cen[(size_t)c*dim+d]-lo[d])/ext*65535.;
                g[d] = std::min<uint64_t>((uint64_t)v, 65535ull);
            }
            uint64_t m = 0;
            for (int bit = 0; bit < 16; bit++)
                for (int d = 0; d < dim; d++) m |= ((g[d] >> bit) & 1ull) << (dim*bit+d);
            morton[c] = m;
        }
    }
    lap(0);
    // ---- chunks of every node's cell list: the nodes are independent, threads take contiguous node ranges of equal work ----
    struct ChunkOut {
        std::vector<int32_t> chunk_cells, chunk_ndof, per_node;
        std::vector<int16_t> chunk_slot;
        std::vector<std::vector<int32_t>> dofl;
    };
    const int nthreads = plan_threads();
    std::vector<ChunkOut> cout_(nthreads);
    {
        std::vector<double> w(nnodes);
        for (int i = 0; i < nnodes; i++) w[i] = 1.+(double)(node_cell_off[i+1]-node_cell_off[i]);
        par_ranges(nnodes, w, nthreads, [&](int tid, int i0, int i1) {
            ChunkOut &O = cout_[tid];
            std::vector<char> member((size_t)N+1, 0);
            std::vector<int32_t> order, uniq;
            for (int i = i0; i < i1; i++) {
                const int32_t *nc_cells = node_cells+node_cell_off[i];
                const int ncell = (int)(node_cell_off[i+1]-node_cell_off[i]);
                order.resize(ncell);
                std::iota(order.begin(), order.end(), 0);
                std::stable_sort(order.begin(), order.end(), [&](int x, int y) { return morton[nc_cells[x]] < morton[nc_cells[y]]; });
                for (int64_t t = node_off[i]; t < node_off[i+1]; t++) member[node_dofs[t]] = 1;
                int s0 = 0, made = 0;
                while (s0 < ncell) {
                    int s1 = std::min(s0+tile, ncell);
                    while (true) {
                        uniq.clear();
                        for (int t = s0; t < s1; t++) {
                            const int c = nc_cells[order[t]];
                            for (int k = 0; k < dpe; k++) { const int g = dofs[(size_t)c*dpe+k]; if (g >= 0 && member[g]) uniq.push_back(g); }
                        }
                        std::sort(uniq.begin(), uniq.end());
                        uniq.erase(std::unique(uniq.begin(), uniq.end()), uniq.end());
                        if ((int)uniq.size() <= max_chunk_dofs || s1-s0 <= 1) break;
                        s1 = s0+std::max(1, (int)((long long)(s1-s0)*max_chunk_dofs/(long long)uniq.size()));
                    }
                    const size_t ch = O.chunk_ndof.size();
                    O.chunk_cells.resize((ch+1)*tile, -1);
                    O.chunk_slot.resize((ch+1)*(size_t)dpe*tile, (int16_t)-1);
                    for (int t = s0; t < s1; t++) {
                        const int c = nc_cells[order[t]];
                        O.chunk_cells[ch*tile+(t-s0)] = c;
                        for (int k = 0; k < dpe; k++) {
                            const int g = dofs[(size_t)c*dpe+k];
                            if (g >= 0 && member[g])
                                O.chunk_slot[(ch*dpe+k)*tile+(t-s0)] = (int16_t)(std::lower_bound(uniq.begin(), uniq.end(), g)-uniq.begin());
                        }
                    }
                    O.chunk_ndof.push_back((int32_t)uniq.size());
                    O.dofl.push_back(uniq);
                    s0 = s1; made++;
                }
                for (int64_t t = node_off[i]; t < node_off[i+1]; t++) member[node_dofs[t]] = 0;
                O.per_node.push_back(made);
            }
        });
    }
    P->node_chunk_off.assign(nnodes+1, 0);
    std::vector<std::vector<int32_t>> chunk_dofl;
    {
        int node = 0;
        for (ChunkOut &O : cout_) {
            P->chunk_cells.insert(P->chunk_cells.end(), O.chunk_cells.begin(), O.chunk_cells.end());
            P->chunk_slot.insert(P->chunk_slot.end(), O.chunk_slot.begin(), O.chunk_slot.end());
            P->chunk_ndof.insert(P->chunk_ndof.end(), O.chunk_ndof.begin(), O.chunk_ndof.end());
            for (auto &u : O.dofl) chunk_dofl.push_back(std::move(u));
            for (int made : O.per_node) { P->node_chunk_off[node+1] = P->node_chunk_off[node]+made; node++; }
        }
    }
    const int nchunks = (int)P->chunk_ndof.size();
    P->nU = 1;
    for (int v : P->chunk_ndof) P->nU = std::max(P->nU, v);
    P->chunk_dofs.assign((size_t)nchunks*P->nU, 0);
    for (int k = 0; k < nchunks; k++) std::copy(chunk_dofl[k].begin(), chunk_dofl[k].end(), P->chunk_dofs.begin()+(size_t)k*P->nU);
    lap(1);
    // ---- cell adjacency through shared vertices (the cell itself included) ----
    std::vector<int64_t> vptr(nv+1, 0);
    for (int c = 0; c < nc; c++) for (int k = 0; k < nV; k++) vptr[cells[(size_t)c*nV+k]+1]++;
    for (int v = 0; v < nv; v++) vptr[v+1] += vptr[v];
    std::vector<int32_t> vcells(vptr[nv]);
    {
        std::vector<int64_t> fill(vptr.begin(), vptr.end()-1);
        for (int c = 0; c < nc; c++) for (int k = 0; k < nV; k++) vcells[fill[cells[(size_t)c*nV+k]]++] = c;
    }
    // neighbour across every edge of every triangle (-1: none), through one sort of all edges
    std::vector<int32_t> enbr;
    if (dim == 2) {
        enbr.assign((size_t)nc*3, -1);
        struct E { int64_t key; int32_t slot; };
        std::vector<E> es((size_t)nc*3);
        for (int c = 0; c < nc; c++)
            for (int p = 0; p < 3; p++) {
                const int va = cells[(size_t)c*3+p], vb = cells[(size_t)c*3+(p+1)%3];
                es[(size_t)c*3+p] = {(int64_t)std::min(va, vb)*nv+std::max(va, vb), (int32_t)(c*3+p)};
            }
        std::sort(es.begin(), es.end(), [](const E &x, const E &y) { return x.key < y.key || (x.key == y.key && x.slot < y.slot); });
        for (size_t t = 0; t+1 < es.size(); t++)
            if (es[t].key == es[t+1].key) { enbr[es[t].slot] = es[t+1].slot/3; enbr[es[t+1].slot] = es[t].slot/3; }
    }
    lap(2);
    // ---- per pair: independent too once the first slot of every pair in the diagonal-block buffer is known ----
    for (int k = 0; k < npairs; k++)
        if (pair_nodes[2*k] < 0 || pair_nodes[2*k] >= nnodes || pair_nodes[2*k+1] < 0 || pair_nodes[2*k+1] >= nnodes) { delete P; return PNL_ERR_INVALID; }
    std::vector<int64_t> dbase_of(npairs+1, 0);
    {
        std::vector<double> w(npairs, 1.);
        std::vector<int64_t> ni(npairs, 0);
        for (int k = 0; k < npairs; k++) w[k] = 1.+(double)(node_cell_off[pair_nodes[2*k]+1]-node_cell_off[pair_nodes[2*k]])
                                                  +(double)(node_cell_off[pair_nodes[2*k+1]+1]-node_cell_off[pair_nodes[2*k+1]]);
        par_ranges(npairs, w, nthreads, [&](int, int k0, int k1) {
            for (int k = k0; k < k1; k++) {
                const int a = pair_nodes[2*k], b = pair_nodes[2*k+1];
                const int32_t *c1 = node_cells+node_cell_off[a], *c2 = node_cells+node_cell_off[b];
                const int32_t *e1 = node_cells+node_cell_off[a+1], *e2 = node_cells+node_cell_off[b+1];
                int64_t cnt = 0;
                while (c1 < e1 && c2 < e2) { if (*c1 < *c2) c1++; else if (*c2 < *c1) c2++; else { cnt++; c1++; c2++; } }
                ni[k] = cnt;
            }
        });
        for (int k = 0; k < npairs; k++) dbase_of[k+1] = dbase_of[k]+ni[k];
    }
    struct PairOut {
        std::vector<int32_t> tile_chunkA, tile_chunkB, tile_pair, tile_flags, tile_dslotA, tile_dslotB, sing[3], d_cell, d_pair, nfacets,
            fvid, bt_slot, bt_cell, bt_facet;
    };
    std::vector<PairOut> pout(nthreads);
    {
        std::vector<double> w(npairs, 1.);
        for (int k = 0; k < npairs; k++) {
            const double n1 = (double)(node_cell_off[pair_nodes[2*k]+1]-node_cell_off[pair_nodes[2*k]]);
            const double n2 = (double)(node_cell_off[pair_nodes[2*k+1]+1]-node_cell_off[pair_nodes[2*k+1]]);
            w[k] = 1.+20.*n1+n2+n1*n2/(double)(tile*tile)*2.*tile;
        }
        par_ranges(npairs, w, nthreads, [&](int tid, int k0, int k1) {
            PairOut &O = pout[tid];
            std::vector<int64_t> pos(nc, -1);
            std::vector<char> in2(nc, 0), inU(nc, 0), vmark(nv, 0);
            std::vector<int32_t> inter, nb, facets, stamp(nc, -1);
            std::vector<int64_t> keys;
            for (int k = k0; k < k1; k++) {
                const int a = pair_nodes[2*k], b = pair_nodes[2*k+1];
                const bool sym = a == b;
                const int32_t *c1 = node_cells+node_cell_off[a], *c2 = node_cells+node_cell_off[b];
                const int n1 = (int)(node_cell_off[a+1]-node_cell_off[a]), n2 = (int)(node_cell_off[b+1]-node_cell_off[b]);
                const int64_t dbase = dbase_of[k];
                inter.clear();
                std::set_intersection(c1, c1+n1, c2, c2+n2, std::back_inserter(inter));
                for (size_t t = 0; t < inter.size(); t++) { pos[inter[t]] = dbase+(int64_t)t; O.d_cell.push_back(inter[t]); O.d_pair.push_back(k); }
                // tiles: chunk pairs (a <= b for n1 == n2)
                for (int ca = P->node_chunk_off[a]; ca < P->node_chunk_off[a+1]; ca++)
                    for (int cb = P->node_chunk_off[b]; cb < P->node_chunk_off[b+1]; cb++) {
                        if (sym && ca > cb) continue;
                        O.tile_chunkA.push_back(ca); O.tile_chunkB.push_back(cb); O.tile_pair.push_back(k); O.tile_flags.push_back(sym ? 1 : 0);
                        for (int l = 0; l < tile; l++) { const int c = P->chunk_cells[(size_t)ca*tile+l]; O.tile_dslotA.push_back(c >= 0 ? (int32_t)pos[c] : -1); }
                        for (int l = 0; l < tile; l++) { const int c = P->chunk_cells[(size_t)cb*tile+l]; O.tile_dslotB.push_back(c >= 0 ? (int32_t)pos[c] : -1); }
                    }
                // touching element pairs {X, Y}, X in n1.cells, Y in n2.cells (folded, unique, ascending key); a neighbour is
                // met through up to nV shared vertices: the stamp keeps one of them
                for (int t = 0; t < n2; t++) in2[c2[t]] = 1;
                keys.clear();
                for (int t = 0; t < n1; t++) {
                    const int X = c1[t];
                    for (int kk = 0; kk < nV; kk++) {
                        const int v = cells[(size_t)X*nV+kk];
                        for (int64_t p = vptr[v]; p < vptr[v+1]; p++) {
                            const int Y = vcells[p];
                            if (in2[Y] && stamp[Y] != X) { stamp[Y] = X; keys.push_back((int64_t)std::min(X, Y)*nc+std::max(X, Y)); }
                        }
                    }
                }
                for (int t = 0; t < n1; t++) {
                    const int X = c1[t];
                    for (int kk = 0; kk < nV; kk++) { const int v = cells[(size_t)X*nV+kk]; for (int64_t p = vptr[v]; p < vptr[v+1]; p++) stamp[vcells[p]] = -1; }
                }
                for (int t = 0; t < n2; t++) in2[c2[t]] = 0;
                std::sort(keys.begin(), keys.end());
                keys.erase(std::unique(keys.begin(), keys.end()), keys.end());
                for (int64_t key : keys) {
                    const int lo = (int)(key/nc), hi = (int)(key%nc);
                    int common = 0;
                    if (lo == hi) common = nV;
                    else for (int p = 0; p < nV; p++) for (int q = 0; q < nV; q++) common += cells[(size_t)lo*nV+p] == cells[(size_t)hi*nV+q];
                    if (common >= 1 && common <= nV) { auto &S = O.sing[common-1]; S.push_back(k); S.push_back(lo); S.push_back(hi); }
                }
                // cluster-local Gauss-theorem term: boundary facets of cellsUnion (orientation of the owning cell), touching (cell, facet)
                int nf = 0;
                if (!inter.empty()) {
                    for (int t = 0; t < n1; t++) inU[c1[t]] = 1;
                    for (int t = 0; t < n2; t++) inU[c2[t]] = 1;
                    facets.clear();
                    nb.clear();
                    std::set_union(c1, c1+n1, c2, c2+n2, std::back_inserter(nb));
                    if (dim == 1) {
                        // vertices that belong to exactly one cell of the set (ascending vertex id)
                        for (int c : nb)
                            for (int p = 0; p < 2; p++) {
                                const int v = cells[(size_t)c*2+p];
                                int cnt = 0;
                                for (int64_t q = vptr[v]; q < vptr[v+1]; q++) cnt += inU[vcells[q]];
                                if (cnt == 1) facets.push_back(v);
                            }
                        std::sort(facets.begin(), facets.end());
                    } else {
                        // edges (oriented like in their cell: (v0,v1), (v1,v2), (v2,v0)) whose neighbour across the edge is not in
                        // the set, in the order they are met walking the cells (clusters.boundaryFacetsOfCells)
                        for (int c : nb)
                            for (int p = 0; p < 3; p++) {
                                const int o = enbr[(size_t)c*3+p];
                                if (o < 0 || !inU[o]) { facets.push_back(cells[(size_t)c*3+p]); facets.push_back(cells[(size_t)c*3+(p+1)%3]); }
                            }
                    }
                    for (int c : nb) inU[c] = 0;
                    nf = (int)facets.size()/dim;
                    O.fvid.insert(O.fvid.end(), facets.begin(), facets.end());
                    // touching (cell of cellsInter, facet) pairs: only cells with a vertex on the boundary of cellsUnion qualify
                    for (int32_t v : facets) vmark[v] = 1;
                    for (int X : inter) {
                        bool any = false;
                        for (int p = 0; p < nV; p++) any = any || vmark[cells[(size_t)X*nV+p]];
                        if (!any) continue;
                        for (int f = 0; f < nf; f++) {
                            bool touch = false;
                            for (int p = 0; p < nV && !touch; p++) for (int q = 0; q < dim; q++) touch = touch || cells[(size_t)X*nV+p] == facets[(size_t)f*dim+q];
                            if (touch) {
                                O.bt_slot.push_back((int32_t)pos[X]); O.bt_cell.push_back(X);
                                for (int q = 0; q < dim; q++) O.bt_facet.push_back(facets[(size_t)f*dim+q]);
                            }
                        }
                    }
                    for (int32_t v : facets) vmark[v] = 0;
                }
                O.nfacets.push_back(nf);
                for (int X : inter) pos[X] = -1;
            }
        });
    }
    lap(3);
    P->pair_foff.assign(npairs+1, 0);
    {
        int pair = 0;
        auto app = [](std::vector<int32_t> &dst, const std::vector<int32_t> &src) { dst.insert(dst.end(), src.begin(), src.end()); };
        for (PairOut &O : pout) {
            app(P->tile_chunkA, O.tile_chunkA); app(P->tile_chunkB, O.tile_chunkB); app(P->tile_pair, O.tile_pair); app(P->tile_flags, O.tile_flags);
            app(P->tile_dslotA, O.tile_dslotA); app(P->tile_dslotB, O.tile_dslotB);
            for (int s3 = 0; s3 < 3; s3++) app(P->sing[s3], O.sing[s3]);
            app(P->d_cell, O.d_cell); app(P->d_pair, O.d_pair); app(P->fvid, O.fvid);
            app(P->bt_slot, O.bt_slot); app(P->bt_cell, O.bt_cell); app(P->bt_facet, O.bt_facet);
            for (int nf : O.nfacets) { P->pair_foff[pair+1] = P->pair_foff[pair]+nf; pair++; }
        }
    }
    P->num_dslots = dbase_of[npairs];
    lap(4);
    // heavy tiles (more real cells) first, stable
    {
        const size_t nt = P->tile_pair.size();
        std::vector<int32_t> real(nchunks, 0);
        for (int c = 0; c < nchunks; c++) for (int l = 0; l < tile; l++) real[c] += P->chunk_cells[(size_t)c*tile+l] >= 0;
        std::vector<size_t> ord(nt);
        std::iota(ord.begin(), ord.end(), 0);
        std::stable_sort(ord.begin(), ord.end(), [&](size_t x, size_t y) {
            return (long long)real[P->tile_chunkA[x]]*real[P->tile_chunkB[x]] > (long long)real[P->tile_chunkA[y]]*real[P->tile_chunkB[y]];
        });
        auto apply = [&](std::vector<int32_t> &v, int w) {
            std::vector<int32_t> t(v.size());
            for (size_t i = 0; i < nt; i++) std::copy(v.begin()+ord[i]*w, v.begin()+(ord[i]+1)*w, t.begin()+i*w);
            v.swap(t);
        };
        apply(P->tile_chunkA, 1); apply(P->tile_chunkB, 1); apply(P->tile_pair, 1); apply(P->tile_flags, 1);
        apply(P->tile_dslotA, tile); apply(P->tile_dslotB, tile);
    }
    lap(6);
    if (timing) fprintf(stderr, "[pnl_nfplan] morton %.3f chunks %.3f adjacency %.3f pairs %.3f merge %.3f sort %.3f s (%d threads)\n", tt[0], tt[1], tt[2], tt[3], tt[4], tt[6], nthreads);
    *out = P;
    return PNL_OK;
}

void pnl_nfplan_destroy(pnl_nfplan *P) { delete P; }

// sizes: [0] nchunks, [1] nU, [2] ntiles, [3..5] touching pairs with 1 / 2 / 3 shared vertices, [6] diagonal-block slots,
// [7] facets, [8] touching (cell, facet) pairs
int pnl_nfplan_sizes(const pnl_nfplan *P, int64_t *out) {
    if (!P || !out) return PNL_ERR_INVALID;
    out[0] = (int64_t)P->chunk_ndof.size(); out[1] = P->nU; out[2] = (int64_t)P->tile_pair.size();
    for (int s = 0; s < 3; s++) out[3+s] = (int64_t)P->sing[s].size()/3;
    out[6] = P->num_dslots; out[7] = (int64_t)P->fvid.size()/P->dim; out[8] = (int64_t)P->bt_cell.size();
    return PNL_OK;
}

// copy-out of one array by name index (see clusters.nearFieldPlan for the shapes)
int pnl_nfplan_get(const pnl_nfplan *P, int which, void *dst) {
    if (!P || !dst) return PNL_ERR_INVALID;
    auto cp32 = [&](const std::vector<int32_t> &v) { if (!v.empty()) std::memcpy(dst, v.data(), v.size()*sizeof(int32_t)); return PNL_OK; };
    switch (which) {
        case 0: return cp32(P->node_chunk_off);
        case 1: return cp32(P->chunk_cells);
        case 2: return cp32(P->chunk_ndof);
        case 3: return cp32(P->chunk_dofs);
        case 4: if (!P->chunk_slot.empty()) std::memcpy(dst, P->chunk_slot.data(), P->chunk_slot.size()*sizeof(int16_t)); return PNL_OK;
        case 5: return cp32(P->tile_chunkA);
        case 6: return cp32(P->tile_chunkB);
        case 7: return cp32(P->tile_pair);
        case 8: return cp32(P->tile_flags);
        case 9: return cp32(P->tile_dslotA);
        case 10: return cp32(P->tile_dslotB);
        case 11: return cp32(P->sing[0]);
        case 12: return cp32(P->sing[1]);
        case 13: return cp32(P->sing[2]);
        case 14: return cp32(P->d_cell);
        case 15: return cp32(P->d_pair);
        case 16: return cp32(P->pair_foff);
        case 17: return cp32(P->fvid);
        case 18: return cp32(P->bt_slot);
        case 19: return cp32(P->bt_cell);
        case 20: return cp32(P->bt_facet);
    }
    return PNL_ERR_INVALID;
}

}  // extern "C"

// =====================================================================================================================
// Sparsity pattern of a finite-horizon operator (getSparse, NA:1062-1260): two cells interact unless all their vertex
// distances are >= delta (getRelativePosition, interactionDomains.pyx:875-898), so the pattern is the set of DoF pairs (I, J)
// for which a cell holding I and a cell holding J have a vertex pair closer than delta -- G = M Q M^T with Q the vertex pairs
// within delta and M = DoF -> vertices of its patch (builder.getSparse formed it with scipy sparse products: 0.45 s at
// 129^2 vertices).  Here: vertices within delta through a uniform grid, then per DoF a bitmap over the DoFs, rows in
// parallel.
// largest number of stored entries a pattern may have: the row pointer is int32 like the reference's INDEX_t
// (pnl_pattern_set_max_nnz lowers it; tests force the guard with it)
static long long g_pattern_max_nnz = 2147483647ll;

struct pnl_pattern {
    std::vector<int32_t> indptr, indices;
    // near-field pattern: the rows of a leaf share one sorted column list; pnl_pattern_get writes the rows straight into the
    // caller's array (no 200 MB intermediate copy)
    std::vector<std::vector<int32_t>> leaf_cols;
    std::vector<int32_t> leaf_of;
    long long nnz_gen = -1;
};

extern "C" {

// Sparsity pattern of the near field (getSparseNearField NA:3226-3289): the union of the blocks n1.dofs x n2.dofs of the cluster
// pairs, CSR with sorted rows; strict_lower keeps I > J (SSS).  The rows of a leaf share one column set: the DoFs of the column
// clusters paired with the leaf or one of its ancestors -- marked in a bitmap per leaf, emitted in order.
int pnl_near_pattern(const pnl_tree *T, int npairs, const int32_t *pairs, int strict_lower, pnl_pattern **out) {
    if (!T || !out || npairs < 0 || (npairs && !pairs)) return PNL_ERR_INVALID;
    const int N = T->N, nn = (int)T->nodes.size();
    for (int k = 0; k < 2*npairs; k++) if (pairs[k] < 0 || pairs[k] >= nn) return PNL_ERR_INVALID;
    // column clusters per row cluster
    std::vector<std::vector<int32_t>> cols_of(nn);
    for (int k = 0; k < npairs; k++) cols_of[pairs[2*k]].push_back(pairs[2*k+1]);
    std::vector<int32_t> leaves;
    for (int n = 0; n < nn; n++) if (T->nodes[n].child[0] < 0) leaves.push_back(n);
    const int nl = (int)leaves.size();
    std::vector<std::vector<int32_t>> leaf_cols(nl);
    std::vector<double> w(nl);
    for (int l = 0; l < nl; l++) w[l] = 1.+(T->nodes[leaves[l]].end-T->nodes[leaves[l]].beg);
    const int nthreads = plan_threads();
    par_ranges(nl, w, nthreads, [&](int, int l0, int l1) {
        std::vector<uint64_t> bits(((size_t)N+63)/64);
        for (int l = l0; l < l1; l++) {
            std::fill(bits.begin(), bits.end(), 0ull);
            bool any = false;
            for (int a = leaves[l]; a >= 0; a = T->nodes[a].parent)
                for (int32_t n2 : cols_of[a]) {
                    const PNode &B = T->nodes[n2];
                    for (int t = B.beg; t < B.end; t++) { const int J = T->perm[t]; bits[J >> 6] |= 1ull << (J & 63); }
                    any = true;
                }
            if (!any) continue;
            auto &v = leaf_cols[l];
            for (size_t wd = 0; wd < bits.size(); wd++) {
                uint64_t m = bits[wd];
                while (m) { const int b = __builtin_ctzll(m); v.push_back((int32_t)(wd*64+b)); m &= m-1; }
            }
        }
    });
    pnl_pattern *P = new pnl_pattern();
    P->indptr.assign((size_t)N+1, 0);
    std::vector<int32_t> leaf_of(N, -1);
    for (int l = 0; l < nl; l++) {
        const PNode &L = T->nodes[leaves[l]];
        for (int t = L.beg; t < L.end; t++) leaf_of[T->perm[t]] = l;
    }
    long long total = 0;
    for (int I = 0; I < N; I++) {
        const auto &v = leaf_cols[leaf_of[I]];
        const long long cnt = strict_lower ? (std::lower_bound(v.begin(), v.end(), I)-v.begin()) : (long long)v.size();
        total += cnt;
        if (total > g_pattern_max_nnz) { delete P; return PNL_ERR_UNSUPPORTED; }
        P->indptr[I+1] = (int32_t)total;
    }
    P->nnz_gen = total;
    P->leaf_cols.swap(leaf_cols);
    P->leaf_of.swap(leaf_of);
    *out = P;
    return PNL_OK;
}

int pnl_horizon_pattern(int dim, int nv, const double *vertices, int nc, const int32_t *cells, int dpe, int N, const int32_t *dofs,
                        double delta, int strict_lower, pnl_pattern **out) {
    if (!out || dim < 1 || dim > 2 || nv <= 0 || nc <= 0 || !vertices || !cells || !dofs || !(delta > 0.) || N <= 0) return PNL_ERR_INVALID;
    const int nV = dim+1;
    const double r = delta*(1.+1e-9), r2 = r*r;
    // vertex -> cells, DoF -> cells
    std::vector<int64_t> vptr(nv+1, 0), dptr(N+1, 0);
    for (int c = 0; c < nc; c++) {
        for (int k = 0; k < nV; k++) vptr[cells[(size_t)c*nV+k]+1]++;
        for (int k = 0; k < dpe; k++) { const int g = dofs[(size_t)c*dpe+k]; if (g >= 0) dptr[g+1]++; }
    }
    for (int v = 0; v < nv; v++) vptr[v+1] += vptr[v];
    for (int g = 0; g < N; g++) dptr[g+1] += dptr[g];
    std::vector<int32_t> vcells(vptr[nv]), dcells(dptr[N]);
    {
        std::vector<int64_t> fv(vptr.begin(), vptr.end()-1), fd(dptr.begin(), dptr.end()-1);
        for (int c = 0; c < nc; c++) {
            for (int k = 0; k < nV; k++) vcells[fv[cells[(size_t)c*nV+k]]++] = c;
            for (int k = 0; k < dpe; k++) { const int g = dofs[(size_t)c*dpe+k]; if (g >= 0) dcells[fd[g]++] = c; }
        }
    }
    // uniform grid of width r over the vertices
    double lo[2] = {INFINITY, INFINITY}, hi[2] = {-INFINITY, -INFINITY};
    for (int v = 0; v < nv; v++) for (int d = 0; d < dim; d++) { lo[d] = std::min(lo[d], vertices[(size_t)v*dim+d]); hi[d] = std::max(hi[d], vertices[(size_t)v*dim+d]); }
    int ng[2] = {1, 1};
    for (int d = 0; d < dim; d++) ng[d] = std::max(1, std::min(4096, (int)std::floor((hi[d]-lo[d])/r)+1));
    auto gcell = [&](int v, int d) { return std::min(ng[d]-1, std::max(0, (int)std::floor((vertices[(size_t)v*dim+d]-lo[d])/r))); };
    const int ncellg = ng[0]*ng[1];
    std::vector<int32_t> gptr(ncellg+1, 0), gverts(nv);
    for (int v = 0; v < nv; v++) gptr[gcell(v, 0)+(dim == 2 ? ng[0]*gcell(v, 1) : 0)+1]++;
    for (int g = 0; g < ncellg; g++) gptr[g+1] += gptr[g];
    {
        std::vector<int32_t> fill(gptr.begin(), gptr.end()-1);
        for (int v = 0; v < nv; v++) gverts[fill[gcell(v, 0)+(dim == 2 ? ng[0]*gcell(v, 1) : 0)]++] = v;
    }
    auto tnow = [] { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
    const double tp0 = tnow();
    // vertices within r of every vertex (Q, the vertex itself included): one grid query per vertex, in parallel
    const int nthreads = plan_threads();
    std::vector<int64_t> qptr(nv+1, 0);
    std::vector<int32_t> qidx;
    {
        struct QOut { std::vector<int32_t> len, idx; };
        std::vector<QOut> qo(nthreads);
        std::vector<double> wv(nv, 1.);
        par_ranges(nv, wv, nthreads, [&](int tid, int v0, int v1) {
            QOut &O = qo[tid];
            for (int v = v0; v < v1; v++) {
                const int gx = gcell(v, 0), gy = dim == 2 ? gcell(v, 1) : 0;
                int cnt = 0;
                for (int yy = std::max(0, gy-1); yy <= std::min(ng[1]-1, gy+1); yy++)
                    for (int xx = std::max(0, gx-1); xx <= std::min(ng[0]-1, gx+1); xx++) {
                        const int g = xx+ng[0]*yy;
                        for (int q = gptr[g]; q < gptr[g+1]; q++) {
                            const int u = gverts[q];
                            double d2 = 0.;
                            for (int d = 0; d < dim; d++) { const double t = vertices[(size_t)v*dim+d]-vertices[(size_t)u*dim+d]; d2 += t*t; }
                            if (d2 <= r2) { O.idx.push_back(u); cnt++; }
                        }
                    }
                O.len.push_back(cnt);
            }
        });
        size_t total = 0;
        for (QOut &O : qo) total += O.idx.size();
        qidx.reserve(total);
        int v = 0;
        for (QOut &O : qo) {
            for (int l : O.len) { qptr[v+1] = qptr[v]+l; v++; }
            qidx.insert(qidx.end(), O.idx.begin(), O.idx.end());
        }
    }
    // the distinct DoFs of the cells holding a vertex
    std::vector<int64_t> udptr(nv+1, 0);
    std::vector<int32_t> udofs;
    {
        std::vector<int32_t> tmp;
        for (int u = 0; u < nv; u++) {
            tmp.clear();
            for (int64_t p = vptr[u]; p < vptr[u+1]; p++)
                for (int k = 0; k < dpe; k++) { const int J = dofs[(size_t)vcells[p]*dpe+k]; if (J >= 0) tmp.push_back(J); }
            std::sort(tmp.begin(), tmp.end());
            tmp.erase(std::unique(tmp.begin(), tmp.end()), tmp.end());
            udofs.insert(udofs.end(), tmp.begin(), tmp.end());
            udptr[u+1] = (int64_t)udofs.size();
        }
    }
    const double tp1 = tnow();
    // rows in parallel: cells of I -> their vertices -> vertices within r -> cells holding those -> their DoFs
    struct Rows { std::vector<int32_t> len, idx; };
    std::vector<Rows> rows(nthreads);
    std::vector<double> w(N, 1.);
    par_ranges(N, w, nthreads, [&](int tid, int i0, int i1) {
        Rows &R = rows[tid];
        const int nwords = (N+63)/64;
        std::vector<uint64_t> bits(nwords, 0);
        std::vector<int32_t> vstamp(nv, -1), ustamp(nv, -1), vlist;
        for (int I = i0; I < i1; I++) {
            // vertices within r of a vertex of a cell of I
            vlist.clear();
            for (int64_t p = dptr[I]; p < dptr[I+1]; p++) {
                const int c1 = dcells[p];
                for (int k = 0; k < nV; k++) {
                    const int v = cells[(size_t)c1*nV+k];
                    if (vstamp[v] == I) continue;                 // this source vertex has been expanded for I
                    vstamp[v] = I;
                    for (int64_t q = qptr[v]; q < qptr[v+1]; q++) {
                        const int u = qidx[q];
                        if (ustamp[u] != I) { ustamp[u] = I; vlist.push_back(u); }
                    }
                }
            }
            // DoFs of the cells holding those vertices
            int wlo = nwords, whi = -1;
            for (int u : vlist)
                for (int64_t p = udptr[u]; p < udptr[u+1]; p++) {
                    const int J = udofs[p];
                    if (strict_lower && J >= I) continue;
                    bits[J >> 6] |= 1ull << (J & 63);
                    wlo = std::min(wlo, J >> 6); whi = std::max(whi, J >> 6);
                }
            int cnt = 0;
            for (int wq = wlo; wq <= whi; wq++) {
                uint64_t b = bits[wq];
                bits[wq] = 0;
                while (b) { const int t = __builtin_ctzll(b); R.idx.push_back(wq*64+t); b &= b-1; cnt++; }
            }
            R.len.push_back(cnt);
        }
    });
    const double tp2 = tnow();
    pnl_pattern *P = new pnl_pattern();
    P->indptr.assign(N+1, 0);
    {
        int I = 0;
        size_t total = 0;
        for (Rows &R : rows) total += R.idx.size();
        // INDEX_t is 32 bits (the CSR / SSS operators and pnl_upload_sparsity index with int32): refuse instead of wrapping
        if ((long long)total > g_pattern_max_nnz) { delete P; return PNL_ERR_UNSUPPORTED; }
        P->indices.reserve(total);
        for (Rows &R : rows) {
            for (int l : R.len) { P->indptr[I+1] = (int32_t)((long long)P->indptr[I]+l); I++; }
            P->indices.insert(P->indices.end(), R.idx.begin(), R.idx.end());
        }
    }
    if (pnl_tune("PNL_PLAN_TIMING")) fprintf(stderr, "[pnl_pattern] vertex neighbours %.3f rows %.3f merge %.3f s (%d threads)\n", tp1-tp0, tp2-tp1, tnow()-tp2, nthreads);
    *out = P;
    return PNL_OK;
}

int64_t pnl_pattern_set_max_nnz(int64_t max_nnz) {
    const long long old = g_pattern_max_nnz;
    if (max_nnz > 0 && max_nnz <= 2147483647ll) g_pattern_max_nnz = max_nnz;
    return old;
}

int64_t pnl_pattern_nnz(const pnl_pattern *P) { return P ? (P->nnz_gen >= 0 ? (int64_t)P->nnz_gen : (int64_t)P->indices.size()) : -1; }

int pnl_pattern_get(const pnl_pattern *P, int32_t *indptr, int32_t *indices) {
    if (!P || !indptr) return PNL_ERR_INVALID;
    std::copy(P->indptr.begin(), P->indptr.end(), indptr);
    if (!indices) return PNL_OK;
    if (P->nnz_gen < 0) { std::copy(P->indices.begin(), P->indices.end(), indices); return PNL_OK; }
    const int N = (int)P->indptr.size()-1;
    std::vector<double> wr(N);
    for (int I = 0; I < N; I++) wr[I] = 1.+(P->indptr[I+1]-P->indptr[I]);
    par_ranges(N, wr, plan_threads(), [&](int, int i0, int i1) {
        for (int I = i0; I < i1; I++) {
            const auto &v = P->leaf_cols[P->leaf_of[I]];
            std::copy(v.begin(), v.begin()+(P->indptr[I+1]-P->indptr[I]), indices+P->indptr[I]);
        }
    });
    return PNL_OK;
}

void pnl_pattern_destroy(pnl_pattern *P) { delete P; }

}  // extern "C"
