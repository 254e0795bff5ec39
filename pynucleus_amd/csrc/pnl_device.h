// Device-side problem description shared by all kernels of libpnl_hip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#define PNL_NTHREADS 256
#define PNL_MAXQ 120            // nonlocalOperator.pyx:107 MAX_PANEL
#define PNL_NCOUNTERS 134
#define PNL_WL_SLOTS 256          // work-list fill counters: one per order class / class pass of an assembly

struct DevKernel {
    int ktype;                  // 0 fractional, 1 indicator, 2 peridynamic
    int fast;                   // 1: fractional, exponent == -qm/4 with an integer qm, no horizon (s = 1/4, 1/2, 3/4 in 1D / 2D)
    double exponent, scale, horizon2;
    int interaction, qm;        // finite horizon: 1 ball2_retriangulation, 2 ball2_barycenter; qm: see fast
    // general exponent (fractional, not fast): binomial coefficients C(exponent, 1..6) and the tables of pnl_pow_tab
    // (pnl_common.h) in device memory: [128] 1/c_j, [128] scale c_j^exponent, [128] 2^(exponent (i - 96)); nullptr: none
    double pb[6];
    const double *ptab;
};

struct DevFormula {
    double c0, a, b, e, den0;
    int clip;
    int pad;
};

struct DevProblem {
    int dim, dpe, nc, ncp, N, dpv, dped, nb;
    double H0;
    // SoA cell data, padded to ncp (multiple of the tile size)
    const double *cellv;        // [(dim+1)*dim][ncp] vertex coordinates of each cell
    const double *ccen;         // [dim][ncp] cell centres
    const double *cvol, *ch;    // [ncp]
    const double *clog;         // [3][ncp]: ln h, |ln(h/H0)| (inputs of the order formula, FL2:622-642), radius centre -- vertex
    const int *cvid;            // [dim+1][ncp] vertex ids (-1-l for padding cells)
    const int *cdof;            // [dpe][ncp] global DoF ids (negative = boundary / padding)
    const short *cslot;         // [dpe][ncp] slot of the DoF in its block's unique-DoF list (-1 if none)
    const int *blk_ndof;        // [nblocks]
    const int *blk_dofs;        // [nblocks][blk_stride]
    int blk_stride, nblocks;
    const int *perm_table;      // [(dim+1)!][dpe]
    DevKernel k, bk;
    DevKernel bkn;              // boundary kernel with the 1/|x-y| of the normal factor folded in (2D)
    DevFormula qo, bqo;
    // distant rules
    int qmax, pad0;
    const int *off;             // [qmax+2]
    const double *bary;         // [total][3]
    const double *w;            // [total]
    const double *phi;          // [total][dpe]
    const int *foff;            // facet rules
    const double *fbary;        // [ftotal][2]
    const double *fw;
    // rules the tile kernel integrates one pair per lane, packed for one coalesced copy into LDS
    const int *tt_n, *tt_off;   // [PNL_MAXQ+2]
    const double *tt_tab;       // [tt_npts][4+dpe]: bary[3], w, phi[dpe]
    const double *tt_wphi;      // [tt_npts][dpe]: w, w*phi[0..dpe-2] (wave-uniform scalar loads; the last product follows from sum_b phi_b = 1)
    const double *tt_wphif;     // [tt_npts][dpe+1]: w, w*phi[0..dpe-1] (second-generation tile kernels, pnl_tile2.h)
    int tt_npts, pad1;
    // singular rules (slot 0 vertex, 1 edge, 2 face)
    int sM[3], sRows[3];
    const double *sNodes[3], *sW[3], *sPsi[3];
    double sFac;
    int bM[2];
    const double *bNodes[2], *bW[2], *bPhi[2];
    double bFac;
    // boundary facets
    const int *bvid;            // [dim][nb]
    const double *bv;           // [dim*dim][nb] facet vertex coordinates
    const double *bgeo;         // [2 dim + 3][nb]: centre, unit normal (2D), length, |ln(len/H0)|, ln(len)
    // variable order, piecewise constant per element pair: labels of cells / boundary facets, class of a label pair and
    // the class this launch assembles (-1: constant order, no filter)
    const int *clabel, *blabel, *cls_of;
    int nlab, cur_class;
    // non-symmetric order table: every pair is assembled once per orientation with half the kernel (the tile machinery
    // applies the factor 2 of the symmetric case); orient = 1 looks up cls_of[label2][label1]; idfac = weight of an
    // identical pair relative to the halved kernel (1 symmetric, 2 non-symmetric: identical pairs are visited once)
    int orient, pad2;
    double idfac;
    unsigned long long *counters;
    // row slab of a rank (pnl_set_row_slab): row / column of the output that holds global DoF I, or -1 (nullptr: I); onesided:
    // symmetric contributions are written once, at (min, max), the operator is A' + A'^T - diag(A')
    const int *rowmap, *colmap;
    int onesided, pad3;
};

// row of the dense output that global DoF I is stored in (identity without a row slab); -1: not stored by this rank
__device__ __forceinline__ long long pnl_row(const DevProblem &P, int I) { return P.rowmap ? (long long)P.rowmap[I] : (long long)I; }
// ... and the column of global DoF J (the columns of a rank's slab are the DoFs of its cells and of all later cells)
__device__ __forceinline__ int pnl_col(const DevProblem &P, int J) { return P.colmap ? P.colmap[J] : J; }

// H2 far field (clusterMethodCy.pyx; kernels in pnl_kernels.h)
// copy table of k_fold_mirror (pnl_tile2.h): storage offset of the copy's row, block << 5 | local DoF, slot column
#define PNL_FOLD_TAB 63
struct FoldEntry { long long off; int ar, cy; };

struct H2Dev {
    int dim, m, M, nnodes, nleaves, nfar;
    const double *box;          // [nnodes][dim][2]
    const int *parent;          // [nnodes] (-1: root)
    const int *leaf_node;       // [nleaves]
    const int *leaf_dof_off, *leaf_dofs;        // sorted DoFs of the leaves
    const int *leaf_cell_off, *leaf_cells;      // cells touching them
    const long long *leaf_val_off;              // [nleaves] offset of V_leaf[ndofs][M]
    const int *far;             // [nfar][2] (n1, n2)
    double *V, *K;              // leaf values, kernel interpolants [nfar][M][M]
    const double *T;            // [nnodes][M_parent][M_child] transfer operator of every non-root node
    double *cup, *cdown;        // [nnodes][M]
};

// order per quadrature point (kernels in pnl_pointwise.h)
struct PwDev {
    int type, normalized;       // 1 constant, 2 smoothStep(x0), 3 linearStep(x0), 4 smoothStepRadial (fractionalOrders.pyx:338-540)
    double p[6];                // sl, sr, r, interface | radius, slope
    int scal_n, pad0;           // scaling C(s) as a Chebyshev series over the range of the order (0: Gamma functions)
    double scal_mid, scal_inv_half, scal_cheb[32];
    double c0, bc0;             // constant term of the interior / boundary order formula
    const double *cell_smax, *facet_smax;
    int M[3], rows[3];
    const double *nodes[3], *w[3], *phi0[3], *phi1[3];      // [nkeys][...] per slot
    int bM[2], pad;
    const double *bnodes[2], *bw[2], *bphi[2];
    double sfac, bfac;
    const double *cell_sv;      // type 5: values of the P1 order function at the vertices of every cell, [(dim+1)][ncp]
    long long sv_stride;        // ncp
};

// block-slot storage of the one-sided operator (pnl_tile2.h)
struct SlotOut {
    double *A2;                 // block-slot storage (nullptr: flush with atomics into A)
    const long long *rowoff;    // [nblocks]
    const int *colbase;         // [nblocks+1], multiples of 8
    int S, nU;                  // nU: the largest number of DoFs of a block (rows of the LDS sub-block of the tile kernels)
};
