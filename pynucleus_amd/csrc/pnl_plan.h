// Internal: the cluster tree the planners fill (pnl_plan.hip on host threads, pnl_plan_dev.hip on the device) and the box metrics
// both use -- one definition, evaluated without FMA contraction, so that the two planners take the same admissibility decisions bit
// for bit (structured meshes put pairs exactly ON the threshold eta dist = diam).
#pragma once
#include <cmath>
#include <cstdint>
#include <vector>
#include <hip/hip_runtime.h>

struct PNode {
    int beg, end, parent, child[2], level;
    int block;                               // kernel block of all DoFs of the node, -1: several blocks (never admissible)
    double box[3][2];
};

struct pnl_tree {
    int N = 0, dim = 0, nc = 0;
    std::vector<int32_t> perm;               // DoFs: node k owns perm[beg:end), ascending inside a LEAF; children are sub-ranges
    std::vector<PNode> nodes;
    std::vector<int32_t> near, far;          // [n][2], [n][3] (n1, n2, level)
    // cells of the nodes that occur in near-field pairs or are leaves (CSR over `cell_nodes`)
    std::vector<int32_t> cell_nodes, cell_off, cells;
    std::vector<int64_t> d2c_ptr;
    std::vector<int32_t> d2c_idx;
    std::vector<double> boxes, coords;
    // variable order: kernel block of every DoF (getKernelBlocksAndJumps NA:2312-2352), mixed_block = the interface DoFs
    std::vector<int32_t> dof_block;
    int mixed_block = -1;
    int ref_type = 0;                        // refinementType: 0 MEDIAN, 1 GEOMETRIC, 2 BARYCENTER (CM:354-663)
};

// distBoxes / diamBox (clusterMethodCy.pyx: the Euclidean gap of two boxes, the diagonal of one) on [dim][2] arrays with stride 2
#pragma clang fp contract(off)
__host__ __device__ inline double pnl_dist_boxes(const double (*a)[2], const double (*b)[2], int dim) {
    double s = 0.;
    for (int d = 0; d < dim; d++) {
        const double g1 = a[d][0]-b[d][1], g2 = b[d][0]-a[d][1];
        const double gap = fmax(0., fmax(g1, g2));
        const double sq = gap*gap;
        s = s+sq;
    }
    return sqrt(s);
}
__host__ __device__ inline double pnl_diam_box(const double (*a)[2], int dim) {
    double s = 0.;
    for (int d = 0; d < dim; d++) { const double e = a[d][1]-a[d][0]; const double sq = e*e; s = s+sq; }
    return sqrt(s);
}
#pragma clang fp contract(fast)

// pnl_plan_dev.hip: refinement and admissibility on the device into the same structure (same nodes, same lists, same order)
int pnl_tree_fill_device(pnl_tree *T, double eta, int min_size, int max_levels, int do_admissibility);
