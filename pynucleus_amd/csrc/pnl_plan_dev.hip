// Cluster tree and admissible pairs on the device (gfx950 only): SURVEY 8(f) row 2, stage 1 of the planner.
//
// Reference: tree_node.refine (clusterMethodCy.pyx:354-663: MEDIAN / GEOMETRIC bisection along the longest box axis until a cluster
// holds minSize DoFs) and getAdmissibleClusters (:4046-4136: the recursion from (root, root): eta dist >= max diam -> far field,
// two leaves -> near field, otherwise the children; a sub-tree without a far-field pair collapses into one near-field pair).
//
// Both are recursions over python objects in the reference and loops over std::vectors in pnl_plan.hip.  Here they are
// level-synchronous sweeps over flat arrays in HBM:
//   * refinement, one level per sweep: boxes of the level's nodes (ordered-integer atomics per DoF), the split axis, the median --
//     two stable radix sorts (coordinate, then node) put every node's coordinates in order, the median is the middle element(s) --,
//     a stable partition of every node's DoF range through ONE exclusive scan of the "left" flags, child numbers through a scan of
//     the "did split" flags: the same breadth-first node numbering, the same ascending DoF order inside every range as the host loop;
//   * admissibility: the frontier of cluster pairs of one depth is classified by one thread per pair (far / near / expand), the
//     children are written behind a scan of their counts; a bottom-up sweep marks the pairs with a far-field pair below them, a
//     top-down sweep the collapsed sub-trees; the surviving pairs are sorted by their path (two bits per level), which IS the depth
//     first order of the reference's recursion.
// The result is the host planner's tree and lists entry for entry (tests/test_plan_native.py): box metrics without FMA contraction
// on both sides (pnl_plan.h), medians as (lower middle + upper middle) / 2.
#include <hipcub/hipcub.hpp>
#include <algorithm>
#include <cstdio>
#include <cstring>
#include <numeric>
#include "pnl_hip.h"
#include "pnl_plan.h"

namespace {

#define PD_CHK(call) do { hipError_t e_ = (call); if (e_ != hipSuccess) { fprintf(stderr, "[pnl_plan_dev] %s: %s\n", #call, hipGetErrorString(e_)); return PNL_ERR_HIP; } } while (0)

struct DBuf {
    void *p = nullptr;
    size_t bytes = 0;
    ~DBuf() { if (p) (void)hipFree(p); }
    int need(size_t b) {
        b = std::max<size_t>(b, 16);
        if (bytes >= b) return PNL_OK;
        if (p) (void)hipFree(p);
        p = nullptr; bytes = 0;
        if (hipMalloc(&p, b) != hipSuccess) return PNL_ERR_HIP;
        bytes = b;
        return PNL_OK;
    }
    template <class T> T *as() { return (T*)p; }
};

// order-preserving map double -> uint64 (and back): the radix sort and the integer atomics see the order of the doubles
__host__ __device__ inline unsigned long long d2key(double x) {
    unsigned long long u;
    __builtin_memcpy(&u, &x, 8);
    return (u >> 63) ? ~u : (u | 0x8000000000000000ull);
}
__host__ __device__ inline double key2d(unsigned long long k) {
    const unsigned long long u = (k >> 63) ? (k & 0x7fffffffffffffffull) : ~k;
    double x;
    __builtin_memcpy(&x, &u, 8);
    return x;
}

struct DNode { int beg, end, parent, child0, child1, level; };

// ---- refinement ---------------------------------------------------------------------------------------------------------------
// box of the nodes [n0, n1) of one level: every DoF position adds its box with ordered-integer atomics
__global__ void k_level_boxes(const DNode *__restrict__ nodes, const int *__restrict__ seg, const int *__restrict__ perm,
                              const double *__restrict__ boxes, int dim, int N, int n0, unsigned long long *__restrict__ bx) {
    const int t = blockIdx.x*blockDim.x+threadIdx.x;
    if (t >= N) return;
    const int k = seg[t];
    if (k < n0) return;                                        // the position belongs to a node that is final
    const double *b = boxes+(size_t)perm[t]*dim*2;
    for (int d = 0; d < dim; d++) {
        atomicMin(&bx[((size_t)k*3+d)*2], d2key(b[2*d]));
        atomicMax(&bx[((size_t)k*3+d)*2+1], d2key(b[2*d+1]));
    }
}

// split axis of the level's nodes that may still be refined, coordinate keys of their DoFs
__global__ void k_level_axis(const DNode *__restrict__ nodes, const unsigned long long *__restrict__ bx, int dim, int n0, int n1, int min_size,
                             int max_levels, int *__restrict__ axis) {
    const int k = n0+blockIdx.x*blockDim.x+threadIdx.x;
    if (k >= n1) return;
    const DNode nd = nodes[k];
    int ax = -1;
    if (nd.level+1 < max_levels && nd.end-nd.beg > min_size) {
        double best = -1.;
        ax = 0;
        for (int d = 0; d < dim; d++) {
            const double e = key2d(bx[((size_t)k*3+d)*2+1])-key2d(bx[((size_t)k*3+d)*2]);
            if (e > best) { best = e; ax = d; }
        }
    }
    axis[k] = ax;
}

__global__ void k_level_keys(const int *__restrict__ seg, const int *__restrict__ perm, const double *__restrict__ coords, const int *__restrict__ axis,
                             int dim, int N, int n0, unsigned long long *__restrict__ ckey, int *__restrict__ pos, unsigned *__restrict__ nkey) {
    const int t = blockIdx.x*blockDim.x+threadIdx.x;
    if (t >= N) return;
    const int k = seg[t];
    const int ax = k >= n0 ? axis[k] : -1;
    // positions of nodes that are not split sort to the end of everything (node key = all ones)
    ckey[t] = ax >= 0 ? d2key(coords[(size_t)perm[t]*dim+ax]) : 0ull;
    nkey[t] = ax >= 0 ? (unsigned)k : 0xffffffffu;
    pos[t] = t;
}

// split value: the median of the sorted coordinates of the node (numpy.median: the mean of the two middle values of an even count),
// or the middle of the box (GEOMETRIC)
__global__ void k_level_median(const DNode *__restrict__ nodes, const int *__restrict__ axis, const unsigned long long *__restrict__ bx,
                               const unsigned long long *__restrict__ sorted_ckey, const int *__restrict__ node_first, int n0, int n1, int ref_type,
                               double *__restrict__ med) {
    const int k = n0+blockIdx.x*blockDim.x+threadIdx.x;
    if (k >= n1) return;
    const int ax = axis[k];
    if (ax < 0) return;
    const DNode nd = nodes[k];
    const int n = nd.end-nd.beg;
    if (ref_type == 1) { med[k] = 0.5*(key2d(bx[((size_t)k*3+ax)*2])+key2d(bx[((size_t)k*3+ax)*2+1])); return; }
    const unsigned long long *s = sorted_ckey+node_first[k];
    if (n & 1) med[k] = key2d(s[n/2]);
    else med[k] = (key2d(s[n/2-1])+key2d(s[n/2]))/2.;
}

__global__ void k_gather_u32(const unsigned *__restrict__ src, const int *__restrict__ pos, int N, unsigned *__restrict__ out) {
    const int t = blockIdx.x*blockDim.x+threadIdx.x;
    if (t < N) out[t] = src[pos[t]];
}

// first position of every split node in the (coordinate, node)-sorted order: the sorted node keys are non-decreasing
__global__ void k_level_first(const unsigned *__restrict__ sorted_nkey, int N, int *__restrict__ node_first) {
    const int t = blockIdx.x*blockDim.x+threadIdx.x;
    if (t >= N) return;
    const unsigned k = sorted_nkey[t];
    if (k == 0xffffffffu) return;
    if (t == 0 || sorted_nkey[t-1] != k) node_first[k] = t;
}

__global__ void k_level_flags(const int *__restrict__ seg, const int *__restrict__ perm, const double *__restrict__ coords, const int *__restrict__ axis,
                              const double *__restrict__ med, int dim, int N, int n0, int *__restrict__ left) {
    const int t = blockIdx.x*blockDim.x+threadIdx.x;
    if (t >= N) return;
    const int k = seg[t];
    const int ax = k >= n0 ? axis[k] : -1;
    left[t] = (ax >= 0 && coords[(size_t)perm[t]*dim+ax] < med[k]) ? 1 : 0;
}

// does the node split?  (both halves need minSize DoFs; clusters.tree_node.refine)
__global__ void k_level_split(const DNode *__restrict__ nodes, const int *__restrict__ axis, const int *__restrict__ left_scan, int N, int n0, int n1,
                              int min_size, int *__restrict__ nleft, int *__restrict__ does) {
    const int k = n0+blockIdx.x*blockDim.x+threadIdx.x;
    if (k >= n1) return;
    const DNode nd = nodes[k];
    int nl = 0, ok = 0;
    if (axis[k] >= 0) {
        // left_scan is the EXCLUSIVE scan of the flags over all positions, with the total at [N]
        nl = left_scan[nd.end]-left_scan[nd.beg];
        const int n = nd.end-nd.beg, nr = n-nl;
        ok = !(nl < min_size || nr < min_size || nl == n || nr == n);
    }
    nleft[k] = nl;
    does[k-n0] = ok;
}

// children of the split nodes (numbered breadth first: behind all existing nodes, in the order of their parents)
__global__ void k_level_children(DNode *__restrict__ nodes, const int *__restrict__ nleft, const int *__restrict__ does, const int *__restrict__ does_scan,
                                 int n0, int n1, int nn) {
    const int k = n0+blockIdx.x*blockDim.x+threadIdx.x;
    if (k >= n1 || !does[k-n0]) return;
    const DNode nd = nodes[k];
    const int c0 = nn+2*does_scan[k-n0];
    nodes[k].child0 = c0; nodes[k].child1 = c0+1;
    DNode a, b;
    a.beg = nd.beg; a.end = nd.beg+nleft[k]; a.parent = k; a.child0 = a.child1 = -1; a.level = nd.level+1;
    b.beg = a.end; b.end = nd.end; b.parent = k; b.child0 = b.child1 = -1; b.level = nd.level+1;
    nodes[c0] = a; nodes[c0+1] = b;
}

// stable partition of the DoF ranges of the split nodes; every position learns its new node
__global__ void k_level_partition(const DNode *__restrict__ nodes, const int *__restrict__ seg, const int *__restrict__ perm, const int *__restrict__ left,
                                  const int *__restrict__ left_scan, const int *__restrict__ does, int N, int n0, int *__restrict__ perm_out,
                                  int *__restrict__ seg_out) {
    const int t = blockIdx.x*blockDim.x+threadIdx.x;
    if (t >= N) return;
    const int k = seg[t];
    if (k < n0 || !does[k-n0]) { perm_out[t] = perm[t]; seg_out[t] = k; return; }
    const DNode nd = nodes[k];
    const int before = left_scan[t]-left_scan[nd.beg];           // "left" positions of the node in front of t
    const int nl = left_scan[nd.end]-left_scan[nd.beg];
    const int dst = left[t] ? nd.beg+before : nd.beg+nl+((t-nd.beg)-before);
    perm_out[dst] = perm[t];
    seg_out[dst] = left[t] ? nd.child0 : nd.child1;
}

// ---- admissibility ------------------------------------------------------------------------------------------------------------
struct DPair { int n1, n2, parent, first_child; unsigned long long path; int kind, nchild; };      // kind 0 far, 1 near, 2 expanded
enum { PD_FAR = 0, PD_NEAR = 1, PD_EXPAND = 2 };

__global__ void k_pairs_classify(DPair *__restrict__ pairs, int p0, int p1, const DNode *__restrict__ nodes, const double *__restrict__ nbox, int dim,
                                 double eta, int level, int max_levels, int *__restrict__ count) {
    const int p = p0+blockIdx.x*blockDim.x+threadIdx.x;
    if (p >= p1) return;
    DPair pr = pairs[p];
    const double (*a)[2] = (const double (*)[2])(nbox+(size_t)pr.n1*6), (*b)[2] = (const double (*)[2])(nbox+(size_t)pr.n2*6);
    const double dist = pnl_dist_boxes(a, b, dim);
    const double dm = fmax(pnl_diam_box(a, dim), pnl_diam_box(b, dim));
    int kind, nch = 0;
    if (eta*dist >= dm) kind = PD_FAR;
    else {
        const bool leaf1 = nodes[pr.n1].child0 < 0, leaf2 = nodes[pr.n2].child0 < 0;
        if ((leaf1 && leaf2) || level == max_levels) kind = PD_NEAR;
        else { kind = PD_EXPAND; nch = (leaf1 ? 1 : 2)*(leaf2 ? 1 : 2); }
    }
    pairs[p].kind = kind; pairs[p].nchild = nch;
    count[p-p0] = nch;
}

__global__ void k_pairs_expand(DPair *__restrict__ pairs, int p0, int p1, const DNode *__restrict__ nodes, const int *__restrict__ count_scan, int level) {
    const int p = p0+blockIdx.x*blockDim.x+threadIdx.x;
    if (p >= p1) return;
    const DPair pr = pairs[p];
    if (pr.kind != PD_EXPAND) return;
    const int base = p1+count_scan[p-p0];
    pairs[p].first_child = base;
    const DNode a = nodes[pr.n1], b = nodes[pr.n2];
    const bool leaf1 = a.child0 < 0, leaf2 = b.child0 < 0;
    const int c1[2] = {leaf1 ? pr.n1 : a.child0, a.child1}, c2[2] = {leaf2 ? pr.n2 : b.child0, b.child1};
    int r = 0;
    // the order of the reference's loops: leaf1 -> over the children of n2; leaf2 -> over the children of n1; else i outer, j inner
    for (int i = 0; i < (leaf1 ? 1 : 2); i++)
        for (int j = 0; j < (leaf2 ? 1 : 2); j++) {
            DPair ch;
            ch.n1 = c1[i]; ch.n2 = c2[j]; ch.parent = p; ch.first_child = -1; ch.kind = -1; ch.nchild = 0;
            ch.path = pr.path | ((unsigned long long)r << (62-2*level));        // level of the PARENT pair: child rank in bits 62-2l, 63-2l
            pairs[base+r] = ch;
            r++;
        }
}

// added[p] = a far-field pair lies in the sub-tree of p (bottom-up, one depth per launch)
__global__ void k_pairs_added(const DPair *__restrict__ pairs, int p0, int p1, char *__restrict__ added) {
    const int p = p0+blockIdx.x*blockDim.x+threadIdx.x;
    if (p >= p1) return;
    const DPair pr = pairs[p];
    char a = pr.kind == PD_FAR;
    if (pr.kind == PD_EXPAND) for (int r = 0; r < pr.nchild; r++) a = a || added[pr.first_child+r];
    added[p] = a;
}

// top-down: a pair below a collapsed pair is dead; emit far pairs, terminal near pairs and collapsed sub-trees
__global__ void k_pairs_emit(const DPair *__restrict__ pairs, int p0, int p1, const char *__restrict__ added, char *__restrict__ dead, int level,
                             unsigned long long *__restrict__ okey, int4 *__restrict__ oval, unsigned *__restrict__ ocount) {
    const int p = p0+blockIdx.x*blockDim.x+threadIdx.x;
    if (p >= p1) return;
    const DPair pr = pairs[p];
    const bool isdead = pr.parent >= 0 && dead[pr.parent];
    const bool collapse = pr.kind == PD_EXPAND && !added[p];
    dead[p] = isdead || collapse;
    if (isdead) return;
    if (pr.kind == PD_FAR || pr.kind == PD_NEAR || collapse) {
        const unsigned o = atomicAdd(ocount, 1u);
        okey[o] = pr.path;
        oval[o] = make_int4(pr.n1, pr.n2, level, pr.kind == PD_FAR ? 1 : 0);
    }
}

}  // namespace

int pnl_tree_fill_device(pnl_tree *T, double eta, int min_size, int max_levels, int do_admissibility) {
    if (!T || T->nodes.size() != 1 || T->ref_type > 1 || !T->dof_block.empty()) return PNL_ERR_UNSUPPORTED;
    const int N = T->N, dim = T->dim;
    const int NT = 256;
    auto grid = [&](long long n) { return dim3((unsigned)((n+NT-1)/NT)); };
    // a binary tree over N DoFs with at least one DoF per leaf
    const int max_nodes = 2*N+1;
    DBuf b_boxes, b_coords, b_perm[2], b_seg[2], b_nodes, b_bx, b_axis, b_ckey[2], b_nkey[2], b_pos[2], b_first, b_med, b_left, b_lscan, b_nleft, b_does, b_dscan, b_tmp;
    int rc;
    if ((rc = b_boxes.need(sizeof(double)*(size_t)N*dim*2)) || (rc = b_coords.need(sizeof(double)*(size_t)N*dim))) return rc;
    for (int i = 0; i < 2; i++)
        if ((rc = b_perm[i].need(sizeof(int)*(size_t)N)) || (rc = b_seg[i].need(sizeof(int)*(size_t)N)) || (rc = b_ckey[i].need(8*(size_t)N)) ||
            (rc = b_nkey[i].need(4*(size_t)N)) || (rc = b_pos[i].need(4*(size_t)N))) return rc;
    if ((rc = b_nodes.need(sizeof(DNode)*(size_t)max_nodes)) || (rc = b_bx.need(8*(size_t)max_nodes*6)) || (rc = b_axis.need(4*(size_t)max_nodes)) ||
        (rc = b_first.need(4*(size_t)max_nodes)) || (rc = b_med.need(8*(size_t)max_nodes)) || (rc = b_left.need(4*(size_t)(N+1))) ||
        (rc = b_lscan.need(4*(size_t)(N+1))) || (rc = b_nleft.need(4*(size_t)max_nodes)) || (rc = b_does.need(4*(size_t)max_nodes)) ||
        (rc = b_dscan.need(4*(size_t)max_nodes))) return rc;
    hipStream_t st = nullptr;
    PD_CHK(hipMemcpyAsync(b_boxes.p, T->boxes.data(), sizeof(double)*(size_t)N*dim*2, hipMemcpyHostToDevice, st));
    PD_CHK(hipMemcpyAsync(b_coords.p, T->coords.data(), sizeof(double)*(size_t)N*dim, hipMemcpyHostToDevice, st));
    PD_CHK(hipMemcpyAsync(b_perm[0].p, T->perm.data(), sizeof(int)*(size_t)N, hipMemcpyHostToDevice, st));
    PD_CHK(hipMemsetAsync(b_seg[0].p, 0, sizeof(int)*(size_t)N, st));
    {
        DNode root = {0, N, -1, -1, -1, 0};
        PD_CHK(hipMemcpyAsync(b_nodes.p, &root, sizeof(DNode), hipMemcpyHostToDevice, st));
    }
    // box table: min keys start at all ones, max keys at zero
    {
        std::vector<unsigned long long> init((size_t)max_nodes*6);
        for (size_t i = 0; i < init.size(); i++) init[i] = (i & 1) ? 0ull : ~0ull;
        PD_CHK(hipMemcpyAsync(b_bx.p, init.data(), 8*init.size(), hipMemcpyHostToDevice, st));
        PD_CHK(hipStreamSynchronize(st));
    }
    size_t tmp_bytes = 0, need = 0;
    (void)hipcub::DeviceRadixSort::SortPairs(nullptr, need, b_ckey[0].as<unsigned long long>(), b_ckey[1].as<unsigned long long>(), b_pos[0].as<int>(), b_pos[1].as<int>(), N, 0, 64, st);
    tmp_bytes = std::max(tmp_bytes, need);
    (void)hipcub::DeviceRadixSort::SortPairs(nullptr, need, b_nkey[1].as<unsigned>(), b_nkey[0].as<unsigned>(), b_ckey[1].as<unsigned long long>(),
                                             b_ckey[0].as<unsigned long long>(), N, 0, 32, st);
    tmp_bytes = std::max(tmp_bytes, need);
    (void)hipcub::DeviceScan::ExclusiveSum(nullptr, need, b_left.as<int>(), b_lscan.as<int>(), N+1, st);
    tmp_bytes = std::max(tmp_bytes, need);
    if ((rc = b_tmp.need(tmp_bytes))) return rc;
    int cur = 0, n0 = 0, nn = 1;
    if (do_admissibility >= 0)
    for (int level = 0; level < max_levels && n0 < nn; level++) {
        const int n1 = nn, nl = n1-n0;
        int *perm = b_perm[cur].as<int>(), *seg = b_seg[cur].as<int>();
        DNode *nodes = b_nodes.as<DNode>();
        hipLaunchKernelGGL(k_level_boxes, grid(N), dim3(NT), 0, st, nodes, seg, perm, b_boxes.as<double>(), dim, N, n0, b_bx.as<unsigned long long>());
        hipLaunchKernelGGL(k_level_axis, grid(nl), dim3(NT), 0, st, nodes, b_bx.as<unsigned long long>(), dim, n0, n1, min_size, max_levels, b_axis.as<int>());
        hipLaunchKernelGGL(k_level_keys, grid(N), dim3(NT), 0, st, seg, perm, b_coords.as<double>(), b_axis.as<int>(), dim, N, n0,
                           b_ckey[0].as<unsigned long long>(), b_pos[0].as<int>(), b_nkey[0].as<unsigned>());
        if (T->ref_type == 0) {
            // sort by coordinate, gather the node keys, stable sort by node: every node's coordinates in order
            size_t tb = b_tmp.bytes;
            PD_CHK(hipcub::DeviceRadixSort::SortPairs(b_tmp.p, tb, b_ckey[0].as<unsigned long long>(), b_ckey[1].as<unsigned long long>(), b_pos[0].as<int>(),
                                                      b_pos[1].as<int>(), N, 0, 64, st));
            // node key of the sorted positions, and the coordinate keys as values of the second sort
            hipLaunchKernelGGL(k_gather_u32, grid(N), dim3(NT), 0, st, b_nkey[0].as<unsigned>(), b_pos[1].as<int>(), N, b_nkey[1].as<unsigned>());
            tb = b_tmp.bytes;
            // values of this sort: the coordinate keys themselves (8 bytes), reusing b_ckey[0] for the output
            PD_CHK(hipcub::DeviceRadixSort::SortPairs(b_tmp.p, tb, b_nkey[1].as<unsigned>(), b_nkey[0].as<unsigned>(), b_ckey[1].as<unsigned long long>(),
                                                      b_ckey[0].as<unsigned long long>(), N, 0, 32, st));
            hipLaunchKernelGGL(k_level_first, grid(N), dim3(NT), 0, st, b_nkey[0].as<unsigned>(), N, b_first.as<int>());
        }
        hipLaunchKernelGGL(k_level_median, grid(nl), dim3(NT), 0, st, nodes, b_axis.as<int>(), b_bx.as<unsigned long long>(), b_ckey[0].as<unsigned long long>(),
                           b_first.as<int>(), n0, n1, T->ref_type, b_med.as<double>());
        hipLaunchKernelGGL(k_level_flags, grid(N), dim3(NT), 0, st, seg, perm, b_coords.as<double>(), b_axis.as<int>(), b_med.as<double>(), dim, N, n0, b_left.as<int>());
        PD_CHK(hipMemsetAsync(b_left.as<int>()+N, 0, sizeof(int), st));
        size_t tb = b_tmp.bytes;
        PD_CHK(hipcub::DeviceScan::ExclusiveSum(b_tmp.p, tb, b_left.as<int>(), b_lscan.as<int>(), N+1, st));
        hipLaunchKernelGGL(k_level_split, grid(nl), dim3(NT), 0, st, nodes, b_axis.as<int>(), b_lscan.as<int>(), N, n0, n1, min_size, b_nleft.as<int>(), b_does.as<int>());
        tb = b_tmp.bytes;
        PD_CHK(hipMemsetAsync(b_does.as<int>()+nl, 0, sizeof(int), st));
        PD_CHK(hipcub::DeviceScan::ExclusiveSum(b_tmp.p, tb, b_does.as<int>(), b_dscan.as<int>(), nl+1, st));
        hipLaunchKernelGGL(k_level_children, grid(nl), dim3(NT), 0, st, nodes, b_nleft.as<int>(), b_does.as<int>(), b_dscan.as<int>(), n0, n1, nn);
        hipLaunchKernelGGL(k_level_partition, grid(N), dim3(NT), 0, st, nodes, seg, perm, b_left.as<int>(), b_lscan.as<int>(), b_does.as<int>(), N, n0,
                           b_perm[cur^1].as<int>(), b_seg[cur^1].as<int>());
        int nsplit = 0;
        PD_CHK(hipMemcpyAsync(&nsplit, b_dscan.as<int>()+nl, sizeof(int), hipMemcpyDeviceToHost, st));
        PD_CHK(hipStreamSynchronize(st));
        PD_CHK(hipGetLastError());
        cur ^= 1;
        n0 = n1;
        nn += 2*nsplit;
        if (nn > max_nodes) return PNL_ERR_STATE;
    }
    // boxes of the last level's nodes (never split, but pairs are classified with them)
    if (n0 < nn)
        hipLaunchKernelGGL(k_level_boxes, grid(N), dim3(NT), 0, st, b_nodes.as<DNode>(), b_seg[cur].as<int>(), b_perm[cur].as<int>(), b_boxes.as<double>(), dim, N, n0,
                           b_bx.as<unsigned long long>());
    // ---- back to the host structure ----
    std::vector<DNode> hn(nn);
    std::vector<unsigned long long> hbx((size_t)nn*6);
    PD_CHK(hipMemcpyAsync(hn.data(), b_nodes.p, sizeof(DNode)*(size_t)nn, hipMemcpyDeviceToHost, st));
    PD_CHK(hipMemcpyAsync(hbx.data(), b_bx.p, 8*(size_t)nn*6, hipMemcpyDeviceToHost, st));
    PD_CHK(hipMemcpyAsync(T->perm.data(), b_perm[cur].p, sizeof(int)*(size_t)N, hipMemcpyDeviceToHost, st));
    PD_CHK(hipStreamSynchronize(st));
    T->nodes.resize(nn);
    std::vector<double> nbox((size_t)nn*6, 0.);
    for (int k = 0; k < nn; k++) {
        PNode &n = T->nodes[k];
        n.beg = hn[k].beg; n.end = hn[k].end; n.parent = hn[k].parent; n.child[0] = hn[k].child0; n.child[1] = hn[k].child1; n.level = hn[k].level;
        n.block = 0;
        for (int d = 0; d < 3; d++) { n.box[d][0] = 0.; n.box[d][1] = 0.; }
        for (int d = 0; d < dim; d++) {
            n.box[d][0] = key2d(hbx[((size_t)k*3+d)*2]); n.box[d][1] = key2d(hbx[((size_t)k*3+d)*2+1]);
            nbox[(size_t)k*6+2*d] = n.box[d][0]; nbox[(size_t)k*6+2*d+1] = n.box[d][1];
        }
    }
    if (do_admissibility <= 0) return PNL_OK;
    // ---- admissible pairs: level-synchronous expansion from (root, root) ----
    if (max_levels > 31) {
        // the path key holds two bits per depth
        int depth = 0;
        for (const PNode &n : T->nodes) depth = std::max(depth, n.level);
        if (2*depth+1 > 31) return PNL_ERR_UNSUPPORTED;
    }
    DBuf b_nbox, b_pairs, b_count, b_cscan, b_added, b_dead, b_okey[2], b_oval[2], b_ocount;
    if ((rc = b_nbox.need(8*(size_t)nn*6))) return rc;
    PD_CHK(hipMemcpyAsync(b_nbox.p, nbox.data(), 8*(size_t)nn*6, hipMemcpyHostToDevice, st));
    size_t cap = std::max<size_t>(1 << 16, (size_t)nn*64);
    if ((rc = b_pairs.need(sizeof(DPair)*cap)) || (rc = b_count.need(4*(cap+1))) || (rc = b_cscan.need(4*(cap+1)))) return rc;
    {
        DPair root;
        root.n1 = 0; root.n2 = 0; root.parent = -1; root.first_child = -1; root.path = 0ull; root.kind = -1; root.nchild = 0;
        PD_CHK(hipMemcpyAsync(b_pairs.p, &root, sizeof(DPair), hipMemcpyHostToDevice, st));
    }
    std::vector<int> lvl_off(1, 0);
    int p0 = 0, p1 = 1;
    size_t scan_bytes = 0;
    (void)hipcub::DeviceScan::ExclusiveSum(nullptr, scan_bytes, b_count.as<int>(), b_cscan.as<int>(), (int)cap+1, st);
    if ((rc = b_tmp.need(scan_bytes))) return rc;
    for (int level = 0; p0 < p1; level++) {
        lvl_off.push_back(p1);
        const int np = p1-p0;
        hipLaunchKernelGGL(k_pairs_classify, grid(np), dim3(NT), 0, st, b_pairs.as<DPair>(), p0, p1, b_nodes.as<DNode>(), b_nbox.as<double>(), dim, eta, level,
                           max_levels, b_count.as<int>());
        PD_CHK(hipMemsetAsync(b_count.as<int>()+np, 0, sizeof(int), st));
        size_t tb = b_tmp.bytes;
        PD_CHK(hipcub::DeviceScan::ExclusiveSum(b_tmp.p, tb, b_count.as<int>(), b_cscan.as<int>(), np+1, st));
        int nchild = 0;
        PD_CHK(hipMemcpyAsync(&nchild, b_cscan.as<int>()+np, sizeof(int), hipMemcpyDeviceToHost, st));
        PD_CHK(hipStreamSynchronize(st));
        if ((size_t)p1+nchild > cap) {
            // grow the pair table (copy what exists)
            const size_t ncap = std::max(cap*2, (size_t)p1+nchild+1024);
            DBuf nb;
            if ((rc = nb.need(sizeof(DPair)*ncap))) return rc;
            PD_CHK(hipMemcpyAsync(nb.p, b_pairs.p, sizeof(DPair)*(size_t)p1, hipMemcpyDeviceToDevice, st));
            PD_CHK(hipStreamSynchronize(st));
            std::swap(nb.p, b_pairs.p); std::swap(nb.bytes, b_pairs.bytes);
            cap = ncap;
            if ((rc = b_count.need(4*(cap+1))) || (rc = b_cscan.need(4*(cap+1)))) return rc;
            // (count / scan are scratch per level: their old contents are not needed)
            hipLaunchKernelGGL(k_pairs_classify, grid(np), dim3(NT), 0, st, b_pairs.as<DPair>(), p0, p1, b_nodes.as<DNode>(), b_nbox.as<double>(), dim, eta, level,
                               max_levels, b_count.as<int>());
            PD_CHK(hipMemsetAsync(b_count.as<int>()+np, 0, sizeof(int), st));
            (void)hipcub::DeviceScan::ExclusiveSum(nullptr, scan_bytes, b_count.as<int>(), b_cscan.as<int>(), (int)cap+1, st);
            if ((rc = b_tmp.need(scan_bytes))) return rc;
            tb = b_tmp.bytes;
            PD_CHK(hipcub::DeviceScan::ExclusiveSum(b_tmp.p, tb, b_count.as<int>(), b_cscan.as<int>(), np+1, st));
        }
        if (nchild) hipLaunchKernelGGL(k_pairs_expand, grid(np), dim3(NT), 0, st, b_pairs.as<DPair>(), p0, p1, b_nodes.as<DNode>(), b_cscan.as<int>(), level);
        PD_CHK(hipGetLastError());
        p0 = p1; p1 += nchild;
        if (level > 31) return PNL_ERR_UNSUPPORTED;
    }
    const int ntot = p1, nlev = (int)lvl_off.size()-1;
    if ((rc = b_added.need((size_t)ntot)) || (rc = b_dead.need((size_t)ntot)) || (rc = b_ocount.need(4))) return rc;
    for (int i = 0; i < 2; i++) if ((rc = b_okey[i].need(8*(size_t)ntot)) || (rc = b_oval[i].need(sizeof(int4)*(size_t)ntot))) return rc;
    for (int l = nlev-1; l >= 0; l--) {
        const int a = lvl_off[l], b = lvl_off[l+1];
        if (b > a) hipLaunchKernelGGL(k_pairs_added, grid(b-a), dim3(NT), 0, st, b_pairs.as<DPair>(), a, b, b_added.as<char>());
    }
    PD_CHK(hipMemsetAsync(b_ocount.p, 0, 4, st));
    for (int l = 0; l < nlev; l++) {
        const int a = lvl_off[l], b = lvl_off[l+1];
        if (b > a) hipLaunchKernelGGL(k_pairs_emit, grid(b-a), dim3(NT), 0, st, b_pairs.as<DPair>(), a, b, b_added.as<char>(), b_dead.as<char>(), l,
                                      b_okey[0].as<unsigned long long>(), b_oval[0].as<int4>(), b_ocount.as<unsigned>());
    }
    unsigned nout = 0;
    PD_CHK(hipMemcpyAsync(&nout, b_ocount.p, 4, hipMemcpyDeviceToHost, st));
    PD_CHK(hipStreamSynchronize(st));
    PD_CHK(hipGetLastError());
    // depth-first order of the reference's recursion = ascending path
    size_t sb = 0;
    (void)hipcub::DeviceRadixSort::SortPairs(nullptr, sb, b_okey[0].as<unsigned long long>(), b_okey[1].as<unsigned long long>(), b_oval[0].as<int4>(), b_oval[1].as<int4>(),
                                             (int)nout, 0, 64, st);
    if ((rc = b_tmp.need(sb))) return rc;
    sb = b_tmp.bytes;
    PD_CHK(hipcub::DeviceRadixSort::SortPairs(b_tmp.p, sb, b_okey[0].as<unsigned long long>(), b_okey[1].as<unsigned long long>(), b_oval[0].as<int4>(), b_oval[1].as<int4>(),
                                              (int)nout, 0, 64, st));
    std::vector<int4> out(nout);
    PD_CHK(hipMemcpyAsync(out.data(), b_oval[1].p, sizeof(int4)*(size_t)nout, hipMemcpyDeviceToHost, st));
    PD_CHK(hipStreamSynchronize(st));
    T->near.clear(); T->far.clear();
    for (const int4 &e : out) {
        if (e.w) { T->far.push_back(e.x); T->far.push_back(e.y); T->far.push_back(e.z); }
        else { T->near.push_back(e.x); T->near.push_back(e.y); }
    }
    return PNL_OK;
}
