/* pnl_hip.h -- C ABI of the MI355X (gfx950) nonlocal-assembly library libpnl_hip.so.
 *
 * This is the drop-in boundary for the hot path of PyNucleus_nl's nonlocalBuilder:
 * the double loop over element pairs that classifies the pair, integrates the
 * kernel over it and scatters the local matrix (reference, Cython, CPU only):
 *
 *   nl/PyNucleus_nl/nonlocalAssembly_{SCALAR}.pxi:1262-1473   nonlocalBuilder.getDense
 *   nl/PyNucleus_nl/nonlocalOperator_decl_{SCALAR}.pxi:8-91   the cdef seam it sits behind
 *        (setMesh1/2, setCell1/2, getPanelType(), eval(contrib, panel, mask))
 *   nl/PyNucleus_nl/kernelsCy.pxd:17                          kernel_fun_t(x, y, c_params)
 *
 * The reference has no C ABI of its own; every entry point below names the
 * reference routine(s) whose work it takes over.  Plain pointers and sizes only --
 * no Python, numpy or torch types.  Pointers named *_host are read on the CPU and
 * copied into HBM by the library; pointers named *_dev must be device memory
 * (hipMalloc / a torch.cuda tensor's data_ptr).  All indices are int32, all reals
 * fp64, arrays are C-contiguous (the reference's INDEX_t / REAL_t, myTypes64.pxd).
 *
 * Error convention: every function returns 0 on success or a negative pnl_status;
 * pnl_error_string() returns a description of the last failure of that context.
 * The host layer maps PNL_ERR_UNSUPPORTED to NotImplementedError and
 * PNL_ERR_INVALID to AssertionError like the reference (NA:915, 941, 1022, 1055).
 * A context is single-caller like the reference's builder objects; internally
 * every call is asynchronous on the context's HIP stream unless stated otherwise.
 */
#ifndef PNL_HIP_H
#define PNL_HIP_H
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct pnl_context pnl_context;

enum pnl_status {
    PNL_OK = 0,
    PNL_ERR_INVALID = -1,       /* bad argument / inconsistent sizes           */
    PNL_ERR_UNSUPPORTED = -2,   /* (dim, element, kernel) combination not built */
    PNL_ERR_HIP = -3,           /* a HIP runtime call failed                    */
    PNL_ERR_STATE = -4,         /* called before the required uploads           */
    PNL_ERR_ORDER = -5          /* a pair needs a quadrature order beyond the uploaded tables */
};

/* PNL_GAUSSIAN: scale exp(exponent |x-y|^2), exponent = -1/(horizon/3)^2 or -1/(2 variance^dim) (kernelsCy.pyx:388-416, 687-692);
 * PNL_EXPONENTIAL: scale exp(exponent |x-y|), exponent = -rate (kernelsCy.pyx:448-462);
 * PNL_GAUSSIAN_BOUNDARY / PNL_EXPONENTIAL_BOUNDARY: their Gauss-theorem twins on the full space, to be set as the boundary kernel
 * (kernelsCy.pyx:418-477): 1D scale sqrt(pi / -exponent) erfc(sqrt(-exponent) |x-y|), 2D scale exp(exponent |x-y|^2) / (-exponent |x-y|);
 * 2 scale exp(exponent |x-y|) / -exponent */
enum pnl_kernel_type { PNL_FRACTIONAL = 0, PNL_INDICATOR = 1, PNL_PERIDYNAMIC = 2, PNL_GAUSSIAN = 3, PNL_EXPONENTIAL = 4,
                       PNL_GAUSSIAN_BOUNDARY = 5, PNL_EXPONENTIAL_BOUNDARY = 6 };

/* gamma(x,y) = scale * (|x-y|^2)^exponent inside |x-y|^2 <= horizon2 (inf: everywhere).
 * Replaces the opaque c_kernel_params block + kernelFun pointer (kernel_params.pxi:8-30,
 * kernelsCy.pyx:75-294).  interaction (finite horizon only): how element pairs cut by the horizon are integrated,
 * 1 = ball2_retriangulation (interactionDomains.pyx:395-822, 866-980), 2 = ball2_barycenter (:340-392, 982-1067).
 * Finite-horizon kernels are assembled from an explicit pair list (pnl_assemble_pairs_masked = getSparse NA:1062-1260). */
typedef struct {
    int32_t ktype;
    int32_t interaction;
    double exponent;
    double scale;
    double horizon2;
} pnl_kernel;

/* Distant-pair quadrature order (fractionalLaplacian2D.pyx:622-642, :1226-1243,
 * fractionalLaplacian1D.pyx:234-253, :646-660):
 * order = max(ceil((c0 + a*L_other + b*Lmax - e*logdh_other)/(max(logdh_self,0)+den0)), 2) */
typedef struct {
    double c0, a, b, e, den0;
    int32_t clip_num;
    int32_t pad;
} pnl_order_formula;

#define PNL_INTERIOR 0
#define PNL_BOUNDARY 1

/* counters (pnl_get_counters), names after the reference's PLogger values NA:1834-1838 */
enum pnl_counter {
    PNL_C_NUM_CELL_PAIRS = 0,          /* pairs visited (c1 <= c2)                       */
    PNL_C_NUM_ASSEMBLED_CELL_PAIRS,    /* pairs with panel != IGNORED                    */
    PNL_C_NUM_INTEGRATIONS,            /* kernel evaluations, interior                   */
    PNL_C_NUM_BOUNDARY_PAIRS,
    PNL_C_NUM_BOUNDARY_INTEGRATIONS,
    PNL_C_ORDER_OVERFLOW,              /* pairs whose order exceeded the tables (error)  */
    PNL_C_UNIFORM_TILE_PAIRS,          /* pairs integrated by the uniform-tile kernel (subset of assembled) */
    PNL_C_RESERVED7,
    PNL_C_HIST0 = 8,                   /* [8+q]: distant pairs of order q, q < 120; [128..130]: vertex/edge/face */
    PNL_C_UNIFORM_Q2 = 131             /* [131..133]: pairs integrated by the uniform-tile kernels of order 2, 3, 4 */
};
#define PNL_NUM_COUNTERS 134

/* Options (process-wide).  The library reads no environment variable on its assembly path; an option exists once it has been
 * set here (value NULL removes it).  A product build accepts
 *   "PNL_WL_FRAC"     capacity of the device work lists as a fraction of the candidate pairs (tests force the overflow report),
 *   "PNL_FH_NOTILES"  finite horizon: every pair through the per-pair pipeline instead of the block tiles,
 *   "PNL_NO_POWTAB"   general exponent: exp(e ln x) instead of the table-driven power,
 *   "PNL_BND_OLD"     Omega x Omega^c term: the per-pair kernel instead of the tiled one (2D),
 *   "PNL_PLAN_THREADS" host threads of the planner,
 *   "PNL_TILE_WGS"    at most this many workgroups per persistent tile kernel (tests: every workgroup then walks many tiles),
 *   "PNL_UNI_GENERIC" uniform-order tiles: the evaluator for arbitrary rules also where the rule has the orbit structure of the
 *                     symmetric 3- and 6-point rules (tests: both evaluators against each other),
 *   "PNL_NO_OVERLAP", "PNL_NO_FORK"   profiling: no side streams, every phase on the caller's stream one after the other,
 *   "PNL_VERBOSE", "PNL_PLAN_TIMING"   diagnostics on stderr,
 * and returns PNL_ERR_UNSUPPORTED for any other name.  A tuning build (make EXTRA=-DPNL_TUNING; pnl_version says so) accepts every
 * name and falls back to the environment: the A/B switches named in DESIGN.md live there. */
int pnl_set_option(const char *name, const char *value);

/* ---- lifetime ---------------------------------------------------------------- */
int pnl_create(int device_id, pnl_context **ctx);
void pnl_destroy(pnl_context *ctx);
const char *pnl_error_string(pnl_context *ctx);
const char *pnl_version(void);
/* use a caller-owned HIP stream (hipStream_t) for all later work; NULL = the context's own stream */
int pnl_set_stream(pnl_context *ctx, void *hip_stream);
int pnl_synchronize(pnl_context *ctx);

/* ---- problem description (replaces setMesh1/2 + precomputeSimplices, NO:111-191) ---------- */
/* vertices[nv][dim], cells[nc][dim+1], vol[nc], h[nc] (mesh.volVector / hVector), H0 = diam/sqrt(8) */
int pnl_upload_mesh(pnl_context *ctx, int dim, int nv, const double *vertices_host, int nc, const int32_t *cells_host,
                    const double *vol_host, const double *h_host, double H0);
/* Cells renumbered by the host (spatial locality, label-following blocks): orig[nc] = the caller's number of every uploaded cell.
 * Touching pairs then keep the orientation the reference's loop gives them -- cellNo1 <= cellNo2 in the CALLER's numbering decides
 * which cell is the first argument of the singular rule (NA:1386-1396) -- so the operator does not depend on the renumbering
 * beyond summation order.  A cell of volume 0 without DoFs is padding inside the mesh: it forms no pair and is not counted.
 * orig = NULL: the numbering of the upload. */
int pnl_set_cell_order(pnl_context *ctx, int nc, const int32_t *orig_host);
/* Interaction set = a linear image of the l2 ball: the kernel sees |T (x - y)| instead of |x - y| (ellipse_retriangulation /
 * ellipse_barycenter = linearTransformInteraction over ball2, interactionDomains.pyx:1393-1630; T = [[cos t / a, -sin t / a],
 * [sin t / b, cos t / b]] row-major, 2D).  Quadrature points and the cut-element geometry are formed in the transformed
 * coordinates; cell volumes, h and the centre distance of the order formula stay those of the mesh.  NULL: identity. */
int pnl_set_interaction_transform(pnl_context *ctx, int dim, const double *T_host);
/* dofs[nc][dpe] (negative = boundary DoF, dm.dofs), dof_perm_table[(dim+1)!][dpe]
 * (precomputedDoFPermutations, NO:66-109) */
int pnl_upload_dofmap(pnl_context *ctx, int dpe, int dofs_per_vertex, int dofs_per_edge, int num_dofs,
                      const int32_t *dofs_host, const int32_t *dof_perm_table_host);
/* Variable fractional order that is piecewise constant per element pair (Kernel.evalParams at the two cell centres,
 * NO:509-513, kernelsCy.pyx:1852-1867; order functions fractionalOrders.pyx:203-336, 826-882): the order takes nclasses
 * distinct values.  cell_labels[nc] / facet_labels[nb] (boundary facets, may be NULL without zeroExterior) are the labels
 * of the centres, cls_of[num_labels][num_labels] the class of a label pair.  After this call pnl_select_class(k) makes
 * pnl_set_kernel / pnl_set_order_formula / pnl_upload_singular_rule act on class k (kernel parameters, order formula and
 * the near rules keyed by the class's singularity, FL2:664/688).  nclasses = 1, num_labels = 0 restores a constant order. */
int pnl_set_classes(pnl_context *ctx, int nclasses, int num_labels, const int32_t *cell_labels_host,
                    const int32_t *facet_labels_host, const int32_t *cls_of_host);
/* The order table of pnl_set_classes is not symmetric, s(label1, label2) != s(label2, label1) (a piecewise-constant,
 * non-symmetric fractional order): every element pair is assembled in both orientations, each with the parameters
 * Kernel.evalParams finds for that orientation -- the symmetricCells == False branch of getDense
 * (nonlocalAssembly_{SCALAR}.pxi:1411-1428) with the fractionalLaplacian{1,2}D_nonsym local matrices, which for parameters
 * frozen per orientation reduce to the symmetric local matrix (temp == temp2 in nonlocalOperator_{SCALAR}.pxi:899-923). */
int pnl_set_nonsymmetric(pnl_context *ctx, int on);
int pnl_select_class(pnl_context *ctx, int k);
/* which = PNL_INTERIOR (gamma) or PNL_BOUNDARY (Gauss-theorem boundary kernel, KC:1982-2027) */
int pnl_set_kernel(pnl_context *ctx, int which, const pnl_kernel *kernel);
int pnl_set_order_formula(pnl_context *ctx, int which, const pnl_order_formula *formula);
/* Distant rules for orders 0..qmax (addQuadRule NO:549-600, addQuadRule_boundary NO:988-1020):
 * off[qmax+2], bary[total][3], w[total], phi[total][dpe]; facet rule foff[qmax+2], fbary[ftotal][2], fw[ftotal] */
int pnl_upload_distant_rules(pnl_context *ctx, int qmax, const int32_t *off_host, const double *bary_host,
                             const double *w_host, const double *phi_host, const int32_t *foff_host,
                             const double *fbary_host, const double *fw_host);
/* Singular rules (getNearQuadRule FL2:644-813, FL2:1255-1314, FL1:255-330, :672-712):
 * panel = -1 vertex, -2 edge, -3 face; nodes[ncoord][M] (ncoord = 2(dim+1) interior, 2dim+1 boundary),
 * w[M], psi[rows][M], fac = multiplier of vol1*vol2 (4 / -2 in 2D, 1 in 1D) */
int pnl_upload_singular_rule(pnl_context *ctx, int which, int panel, int M, int rows, const double *nodes_host,
                             const double *w_host, const double *psi_host, double fac);
/* boundary facets bcells[nb][dim] (mesh.get_surface_mesh().cells, oriented as in their cell) */
int pnl_upload_boundary(pnl_context *ctx, int nb, const int32_t *bcells_host);

/* Finite-horizon operator without a host pair list (nonlocalBuilder.getSparse, nonlocalAssembly_{SCALAR}.pxi:1062-1260): the
 * candidate cell pairs are those of the block tiles the horizon can reach, REMOTE pairs are dropped
 * (getRelativePosition, interactionDomains.pyx:875-898; nonlocalOperator_{SCALAR}.pxi:515-517), the others are integrated
 * (pairs inside the horizon tile-wise, cut pairs through the sub-simplex loops) and added without masks into the uploaded
 * pattern.  Replaces the loops
 * NA:1150-1200 over the cells of the covering cluster pairs (clusterMethodCy.pyx:4139-4194). */
int pnl_assemble_pairs_in_horizon(pnl_context *ctx, double *data, double *diag);
/* the same for the pairs whose FIRST cell lies in [cell_begin, cell_end): the reference's split of cellNo1 over ranks
 * (nonlocalAssembly_{SCALAR}.pxi:1280-1285); the parts of a partition add up to the operator (atomic adds into data / diag) */
int pnl_assemble_pairs_in_horizon_range(pnl_context *ctx, double *data_dev, double *diag_dev, int cell_begin, int cell_end);

/* ---- the hot path --------------------------------------------------------------------------- */
/* nonlocalBuilder.getDense (NA:1262-1473): accumulates the operator into A_dev[num_dofs][ldA]
 * (caller zeroes it).  Cell pairs (c1,c2), c1<=c2, with c1 in [cell_begin, cell_end) are assembled
 * (the reference's MPI split NA:1280-1285); zero_exterior adds the Omega x Omega^c term for the
 * same cells.  flags: PNL_FLAG_* below. */
#define PNL_FLAG_NO_MIRROR 1   /* leave cross contributions in A'[I,J] only (operator = A' + A'^T), for sharded matvec */
#define PNL_FLAG_SYMMETRIC_FLUSH 2   /* write every cross contribution at (I,J) and (J,I) instead of the N^2 mirror pass:
                                      * the flush work is shared by the ranks, the mirror pass is not (multi-GPU) */
int pnl_assemble_dense(pnl_context *ctx, double *A_dev, int64_t ldA, int zero_exterior, int cell_begin, int cell_end,
                       int flags);
/* Estimated cost of every block row a of the upper block triangle (tiles (a, b), b >= a, weighted by the kernel that will take them,
 * plus the per-cell work of the row's cells), for dealing contiguous block-row ranges of equal WORK over ranks
 * (tree_node.partition clusterMethodCy.pyx:1854-1896 deals by DoF count); n = number of blocks = ceil(num_cells / pnl_tile_cells) */
int pnl_block_row_costs(pnl_context *ctx, double *cost_out, int n);
/* 1 if pnl_assemble_dense with these arguments OVERWRITES every entry of A (the operator is formed in the block-slot storage and
 * folded into A in one sweep: P2 elements in 2D, whole cell range, mirrored, room for the storage), 0 if it ADDS to A and the
 * caller has to zero the matrix first (the reference allocates a zeroed matrix, NA:1262-1290), < 0 on error */
int pnl_dense_overwrites(pnl_context *ctx, int cell_begin, int cell_end, int flags);
/* Like pnl_assemble_dense but with an explicit work list of block-tile pairs for the distant pairs
 * (multi-GPU balancing): tiles[2*i] <= tiles[2*i+1] are cell-block indices, block size pnl_tile_cells().
 * Touching pairs, the Omega x Omega^c term and nothing else are restricted to c1 in [cell_begin, cell_end). */
int pnl_tile_cells(pnl_context *ctx);
int pnl_assemble_dense_tiles(pnl_context *ctx, double *A_dev, int64_t ldA, int zero_exterior, int ntiles,
                             const int32_t *tiles_host, int cell_begin, int cell_end, int flags);
/* ---- H2 near field: nonlocalBuilder.assembleClusters (NA:1663-1964) ------------------------------------------ */
/* Sparsity pattern of the near-field matrix (getSparseNearField NA:3226-3289): CSR indptr[num_dofs+1], indices[nnz],
 * column indices strictly increasing per row.  For an SSS_LinearOperator only the strict lower triangle (I > J) is
 * listed and the diagonal lives in its own vector (SSS_LinearOperator_{SCALAR}.pxi:23-60). */
int pnl_upload_sparsity(pnl_context *ctx, int nnz, const int32_t *indptr_host, const int32_t *indices_host);
/* the same with the pattern already in device memory (built there by the host layer's getSparseNearField): device-to-device
 * copies on the context's stream; only the two ends of indptr are checked -- the caller guarantees sorted, in-range rows (use
 * pnl_upload_sparsity for a validated upload) */
int pnl_upload_sparsity_device(pnl_context *ctx, int nnz, const int32_t *indptr_dev, const int32_t *indices_dev);
/* 'interior' loop over the recorded element pairs (NA:1776-1832): pairs[np][2] with c1 <= c2, masks[np][4] = the
 * 256-bit MASK_t of requested entries of the symmetric local matrix (bit k(p,q), p <= q over the 2*dpe local DoFs,
 * buildMasksForClusters NA:260-391).  Panel + quadrature as in pnl_assemble_dense; scatter = addToMatrixElemElemSymMasked
 * (NA:503-520) with the reference's addToEntry semantics: entries absent from the pattern are dropped
 * (CSR_LinearOperator_{SCALAR}.pxi:150-170), SSS keeps I >= J only (SSS_LinearOperator_{SCALAR}.pxi:104-130).
 * data_dev[nnz] (+ diag_dev[num_dofs] for SSS, NULL for CSR) are accumulated into; the caller zeroes them. */
/* masks == NULL: every entry of every pair is requested */
int pnl_assemble_pairs_masked(pnl_context *ctx, int np, const int32_t *pairs_host, const uint64_t *masks_host,
                              double *data_dev, double *diag_dev);
/* Gauss-theorem boundary term over explicit items: the cluster-local term (NA:1842-1889; facets = boundary of
 * cellsUnion, nonlocalAssembly.pyx:505-578) and the global Omega x Omega^c term with fac = +-1 (NA:1896-1913, 1945-1964).
 * cells[ni], facets[ni][dim] vertex ids oriented as in their owning cell, masks[ni] bit field over the dpe(dpe+1)/2
 * entries (getElemSymMaskCluster NA:463-478), scatter addToMatrixElemSymMasked NA:534-546 times fac. */
int pnl_assemble_boundary_masked(pnl_context *ctx, int ni, const int32_t *cells_host, const int32_t *facets_host,
                                 const uint32_t *masks_host, double fac, double *data_dev, double *diag_dev);
/* Tiled near-field assembly: the same operator as pnl_assemble_pairs_masked + pnl_assemble_boundary_masked over all
 * cluster pairs (assembleClusters NA:1663-1964), organised for the GPU instead of element pair by element pair.  The host
 * (clusters.nearFieldPlan) lists, per UNORDERED near-field cluster pair {n1, n2} (pair_nodes; nodes = sorted DoF lists
 * node_off / node_dofs): tiles (64-cell chunk of n1.cells) x (chunk of n2.cells) -- chunk tables: cells (-1 = padding),
 * the chunk's DoFs that belong to the node and the slot of every local DoF in that list --; a slot in the diagonal-block
 * buffer for every cell of cellsInter (d_cell, d_pair; tile_dslotA/B give the slot of a tile's cells or -1); the
 * touching element pairs (sing_items[common vertices - 1] = (pair, c1 <= c2)); the facets of the boundary of cellsUnion
 * (pair_foff, fvid) and the touching (cell, facet) items of the cluster-local Gauss-theorem term (bt_slot, bt_cell,
 * bt_facet).  tile_flags bit 0: n1 == n2.  No masks cross the boundary: an entry of a local matrix is kept iff its DoF pair
 * {I, J} belongs to the cluster pair ((I in n1, J in n2) or (I in n2, J in n1)), and is written at (I, J) and (J, I) with the
 * addToEntry semantics of the uploaded pattern.  Constant order, infinite horizon.  cluster_boundary = 0 skips the
 * Gauss-theorem term. */
typedef struct {
    int32_t npairs, nnodes, nchunks, chunk_stride, ntiles, num_dslots, nfacets, tile;
    const int32_t *pair_nodes, *node_off, *node_dofs;
    const int32_t *chunk_cells, *chunk_ndof, *chunk_dofs;
    const int16_t *chunk_slot;
    const int32_t *tile_chunkA, *tile_chunkB, *tile_pair, *tile_flags, *tile_dslotA, *tile_dslotB;
    const int32_t *d_cell, *d_pair;
    int32_t n_sing[3];
    int32_t n_btouch;
    const int32_t *sing_items[3];
    const int32_t *pair_foff, *fvid;
    const int32_t *bt_slot, *bt_cell, *bt_facet;
} pnl_cluster_plan;
int pnl_assemble_clusters_tiled(pnl_context *ctx, const pnl_cluster_plan *plan, int cluster_boundary, double *data_dev,
                                double *diag_dev);
/* ---- H2 far field (clusterMethodCy.pyx) -----------------------------------------------------------------------------
 * Cluster tree (nodes with parent, level, box), its leaves (sorted DoFs, cells touching them), the admissible cluster pairs
 * far[nfar][2] = (n1, n2) (getAdmissibleClusters :4046-4136), one interpolation order m for all nodes, the transfer
 * operators transfer[node][M][M] (row: tensor index on the parent's Chebyshev grid; transferMatrixBuilder :2004-2073; unused
 * for the root) and a volume quadrature rule for the leaf values.  pnl_h2_setup evaluates on the device the kernel
 * interpolants -2 gamma(xi_i, eta_j) of every admissible pair (assembleFarFieldInteractions :2153-2238) and the leaf values
 * int phi_I L_alpha (enterLeafValues :1205-1325).  pnl_h2_matvec adds the far field to y: upward pass, interactions,
 * downward pass (H2Matrix.matvec :2269-2295 without the near-field term, which is pnl_spmv).  A kernel with a finite horizon (l2 ball)
 * is taken if every admissible pair lies inside the horizon (pnl_tree_build_horizon).  Tensor index
 * alpha = alpha_0 + m alpha_1 (coordinate 0 fastest), M = m^dim. */
typedef struct {
    int32_t nnodes, nleaves, nfar, m, nlevels, nq;
    const double *box;
    const int32_t *parent, *level;
    const int32_t *leaf_node, *leaf_dof_off, *leaf_dofs, *leaf_cell_off, *leaf_cells;
    const int32_t *far;
    const double *transfer;
    const double *qbary, *qw, *qphi;
    /* variable order (pnl_set_classes): kernel class of every admissible pair -- the order between the kernel blocks of its two
     * clusters (the reference evaluates the variable kernel at the interpolation points, clusterMethodCy.pyx:2213); NULL for a
     * constant order */
    const int32_t *far_class;
    /* 1: the leaves cover only a part of the DoFs (a rank's own subtrees; DistributedH2Matrix_localData clusterMethodCy.pyx:3368-3920):
     * the upward pass reads x, and the downward pass adds to y, at the DoFs of these leaves only; 0: the leaves partition the DoFs */
    int32_t partial_leaves;
} pnl_h2_plan;
int pnl_h2_setup(pnl_context *ctx, const pnl_h2_plan *plan);
int pnl_h2_matvec(pnl_context *ctx, const double *x_dev, double *y_dev);
/* The phases of pnl_h2_matvec on caller-owned coefficient arrays cup / cdown [nnodes][M] (device memory), for operators that exchange
 * cluster coefficients between ranks instead of vector entries (communicateFar, clusterMethodCy.pyx:3610-3647):
 *   pnl_h2_upward    cup = 0; leaves of the plan: cup[leaf] = V^T x; then level by level cup[parent] += T cup[child]
 *   pnl_h2_interact  cdown = 0; cdown[n1] += K cup[n2] over the plan's admissible pairs
 *   pnl_h2_downward  level by level cdown[child] += T^T cdown[parent]; y[dofs of the plan's leaves] += V cdown[leaf]
 * pnl_h2_sizes: out2 = (number of nodes, M) */
int pnl_h2_upward(pnl_context *ctx, const double *x_dev, double *cup_dev);
int pnl_h2_interact(pnl_context *ctx, const double *cup_dev, double *cdown_dev);
int pnl_h2_downward(pnl_context *ctx, double *cdown_dev, double *y_dev);
int pnl_h2_sizes(pnl_context *ctx, int32_t *out2);
/* what an H2 operator file stores besides the tree (H2Matrix.HDF5write / HDF5read, clusterMethodCy.pyx:2449-2550): which = 0 the
 * kernel interpolants K[nfar][M][M] of the plan's admissible pairs, which = 1 the leaf values, V_leaf[ndofs][M] of the plan's leaves
 * one after the other.  _get copies them to the host, _set replaces what pnl_h2_setup computed (an operator read from a file) */
int pnl_h2_get(pnl_context *ctx, int which, double *dst_host);
int pnl_h2_set(pnl_context *ctx, int which, const double *src_host);

/* y = A x for the uploaded pattern (CSR: diag_dev NULL; SSS: lower triangle + diagonal, y = (L + D + L^T) x):
 * CSR_LinearOperator.matvec / SSS_LinearOperator.matvec */
int pnl_spmv(pnl_context *ctx, const double *data_dev, const double *diag_dev, const double *x_dev, double *y_dev);

/* ---- non-symmetric kernels with a fractional order s(x) evaluated per quadrature point --------------------------------
 * Replaces fractionalLaplacian{1,2}D_nonsym (fractionalLaplacian2D.pyx:894-1184, fractionalLaplacian1D.pyx:410-604) with
 * piecewise == False kernels gamma(x, y) = C(s(x)) |x-y|^(-d-2 s(x)) (kernels.py:147-149, kernelsCy.pyx:596-622), the
 * non-symmetric branch of getDense (nonlocalAssembly_{SCALAR}.pxi:1411-1428, scatter :222-253) and the boundary term with
 * the pointwise boundary kernel.  P1 elements, infinite horizon.
 * type: 1 constant (p[0]), 2 smoothStep in x0, 3 linearStep in x0, 4 smoothStepRadial (fractionalOrders.pyx:338-540);
 * p = sl, sr, r, interface (radius for type 4), slope; type 5: pnl_set_order_vertex_values.  normalized: variableFractionalLaplacianScaling
 * (kernelNormalization.pyx:329-364) or 1/2. */
typedef struct pnl_order_function {
    int32_t type, normalized;
    double p[6];
    /* optional: the scaling C(s) as a Chebyshev series sum_k scal_cheb[k] T_k((s - scal_mid)/scal_half) over the range of
     * the order (a polynomial instead of two Gamma functions per quadrature point); scal_n == 0: evaluate the formula */
    int32_t scal_n, pad;
    double scal_mid, scal_half;
    double scal_cheb[32];
} pnl_order_function;

/* cell_smax[nc] / facet_smax[nb]: largest order over the centre and the vertices of a cell / boundary facet (the per-pair
 * order of Kernel.evalParamsOnSimplices, kernelsCy.pyx:1825-1846, is the max of the two); c0 / bc0: constant term of the
 * interior / boundary order formula; sing_fac / bsing_fac as in pnl_upload_singular_rule */
int pnl_set_order_function(pnl_context *ctx, const pnl_order_function *f, const double *cell_smax, const double *facet_smax,
                           double c0, double bc0, double sing_fac, double bsing_fac);
/* type 5 (feFractionalOrder / lookupExtended, fractionalOrders.pyx:541-587, 660-668): the order is a continuous P1 function on the
 * mesh of the assembly, values[nv] at its vertices.  A quadrature point is known by its cell and barycentric coordinates, so
 * s(x) = sum_k lambda_k(x) values[vertex k of the cell] needs no point location. */
int pnl_set_order_vertex_values(pnl_context *ctx, int nv, const double *values_host);
/* near rules for the nkeys distinct orders of the touching pairs (the reference keys its rule dictionary by the singularity
 * value, fractionalLaplacian2D.pyx:957): nodes[nkeys][2(dim+1) | (dim+1)+dim][M], w[nkeys][M], phi0 / phi1[nkeys][rows][M]
 * = the x and y parts of the merged-DoF shape functions (PHI3 of the reference; boundary: phi0 = PHI, phi1 unused) */
int pnl_upload_pointwise_rules(pnl_context *ctx, int which, int panel, int nkeys, int M, int rows, const double *nodes,
                               const double *w, const double *phi0, const double *phi1);
/* dense assembly; pairs[npairs][4] = (c1 <= c2 sharing `common` vertices, common, key into the near rules), bpairs[nbpairs][4]
 * = (cell, boundary facet, common, key).  Distant pairs are found and classified on the device. */
int pnl_assemble_dense_pointwise(pnl_context *ctx, double *A, int64_t ldA, int zero_exterior, int cell_begin, int cell_end,
                                 int npairs, const int32_t *pairs, int nbpairs, const int32_t *bpairs);
/* near field of these kernels: the masked element-pair loop of assembleClusters with symmetricCells == symmetricLocalMatrix == False
 * (nonlocalAssembly_{SCALAR}.pxi:1776-1840: masks over cellsUnion x cellsUnion :322-349, getElemElemMask :425-440, scatter
 * addToMatrixElemElemMasked :520-532).  pairs[np][2]: ORDERED cell pairs, each evaluated in its own orientation (swapCells is the
 * listing of (c2, c1)); masks[np][4]: bit p (2 dpe) + q of the 256-bit MASK_t requests local entry (p, q); rule[np]: -1 for a pair
 * without a common vertex (classified on the device), otherwise the key of the pair's near rule (pnl_upload_pointwise_rules);
 * data: unsymmetric CSR over the uploaded pattern (pnl_upload_sparsity), entries absent from it are dropped (addToEntry). */
int pnl_assemble_pairs_masked_pointwise(pnl_context *ctx, int np, const int32_t *pairs, const uint64_t *masks, const int32_t *rule,
                                        double *data);
/* cluster exterior of these kernels (local_matrix_surface with the pointwise boundary kernel, nonlocalAssembly_{SCALAR}.pxi:1966-2028;
 * the global Omega x Omega^c term :2126-2156 with fac = -1): items (cell, facet vertex ids[dim], mask over the dpe (dpe+1)/2 entries)
 * as pnl_assemble_boundary_masked, plus rule[ni] (-1: no common vertex, else the key of the boundary near rule) and sv[ni], the
 * largest order over the centres and vertices of cell and facet (evalParamsOnSimplices, kernelsCy.pyx:1825-1846). */
int pnl_assemble_boundary_masked_pointwise(pnl_context *ctx, int ni, const int32_t *cells, const int32_t *facets, const uint32_t *masks,
                                           const int32_t *rule, const double *sv, double fac, double *data, double *diag);

/* counters of the last assemble call (synchronises the stream) */
int pnl_get_counters(pnl_context *ctx, int64_t *out, int n);
/* device time of the last assemble call per phase in milliseconds (HIP events on the context's stream):
 * [0] general tile kernel (distant pairs, one per lane), [1] work-list kernels (high orders),
 * [2] singular pairs, [3] boundary term, [4] mirror + diagonal scatter, [5] total, [6] uniform-tile kernel */
int pnl_get_phase_ms(pnl_context *ctx, float *out, int n);
/* device time of the tile kernels of the last assemble call, one HIP-event pair around each launch (milliseconds, 0 if the
 * kernel did not run; with several order classes the last launch of a slot): [0] general tile kernel, [1] uniform tiles of
 * order 2, [2] order 3, [3] order 4, [4] fold + mirror of the block-slot storage, [5] work-list kernels */
enum pnl_kernel_slot { PNL_K_TILE_GENERAL = 0, PNL_K_TILE_UNIFORM2, PNL_K_TILE_UNIFORM3, PNL_K_TILE_UNIFORM4, PNL_K_FOLD_MIRROR,
                       PNL_K_WORKLIST, PNL_NUM_KERNEL_SLOTS };
int pnl_get_kernel_ms(pnl_context *ctx, float *out, int n);

/* ---- host-side planning of the H2 / near-field assembly (no GPU needed): cluster tree + admissibility
 *      (tree_node.refine clusterMethodCy.pyx:354-663 -- median splits here --, getAdmissibleClusters :4046-4136), cells of
 *      the cluster nodes (NA:2887-2898), the tile work lists of pnl_assemble_clusters_tiled (what clusters.nearFieldPlan
 *      built in numpy: chunks, tiles, touching pairs, boundary facets of cellsUnion nonlocalAssembly.pyx:540-578) and the
 *      transfer matrices of the far field (transferMatrixBuilder :2004-2073) ------------------------------------------ */
typedef struct pnl_tree pnl_tree;
typedef struct pnl_nfplan pnl_nfplan;
/* boxes[N][dim][2] support boxes of the DoFs, (d2c_ptr, d2c_idx) DoF -> cells CSR; do_admissibility: 1 tree + recursion from
 * (root, root), 0 tree only, -1 root only */
int pnl_tree_build(int N, int dim, const double *boxes, const int64_t *d2c_ptr, const int32_t *d2c_idx, int nc, double eta,
                   int min_size, int max_levels, int do_admissibility, pnl_tree **out);
/* variable order: dof_block[N] >= 0 names the kernel block of every DoF (getKernelBlocksAndJumps NA:2312-2352, the reference
 * hangs one child per block below the leaves of the partition tree, NA:2619-2640), mixed_block the block of the interface DoFs:
 * nodes with DoFs of several blocks are split by block first, only pairs of single-block nodes other than mixed_block can be
 * admissible (mixed_node, clusterMethodCy.pyx:4038) */
int pnl_tree_build_blocks(int N, int dim, const double *boxes, const int64_t *d2c_ptr, const int32_t *d2c_idx, int nc, double eta,
                          int min_size, int max_levels, int do_admissibility, const int32_t *dof_block, int mixed_block,
                          pnl_tree **out);
/* ... with the refinement type of the reference (refinementType, NA:3033-3040; tree_node.refine CM:354-663): 0 MEDIAN (the default
 * of both), 1 GEOMETRIC (the node's box is halved along its longest edge), 2 BARYCENTER (split at the mean of the DoF coordinates) */
int pnl_tree_build_refined(int N, int dim, const double *boxes, const int64_t *d2c_ptr, const int32_t *d2c_idx, int nc, double eta,
                           int min_size, int max_levels, int do_admissibility, const int32_t *dof_block, int mixed_block, int ref_type,
                           pnl_tree **out);
/* ... for a kernel with a finite horizon (getAdmissibleClusters, clusterMethodCy.pyx:4069-4090, 4115, 4131-4135, with distBoxes /
 * maxDistBoxes of the l2 ball, interactionDomains.pyx:304-337): cluster pairs farther apart than the horizon are dropped, pairs the
 * horizon may cut stay in the near field, near-field children are merged into one block only if the block fits into the horizon.
 * horizon = INFINITY: pnl_tree_build_refined */
int pnl_tree_build_horizon(int N, int dim, const double *boxes, const int64_t *d2c_ptr, const int32_t *d2c_idx, int nc, double eta,
                           int min_size, int max_levels, int do_admissibility, const int32_t *dof_block, int mixed_block, int ref_type,
                           double horizon, pnl_tree **out);
/* the same structure with refinement (MEDIAN / GEOMETRIC) and admissibility run ON THE DEVICE as level-synchronous sweeps
 * (clusterMethodCy.pyx:354-663, 4046-4136; csrc/pnl_plan_dev.hip): same nodes, same lists in the same order as pnl_tree_build_refined;
 * kernel blocks and the BARYCENTER split: PNL_ERR_UNSUPPORTED (use the host planner).  Uses the current HIP device. */
int pnl_tree_build_device(int N, int dim, const double *boxes, const int64_t *d2c_ptr, const int32_t *d2c_idx, int nc, double eta,
                          int min_size, int max_levels, int do_admissibility, int ref_type, pnl_tree **out);
void pnl_tree_destroy(pnl_tree *T);
int pnl_tree_sizes(const pnl_tree *T, int64_t *out3);                /* nodes, near pairs, far pairs */
int pnl_tree_get(const pnl_tree *T, int32_t *range, int32_t *parent, int32_t *children, int32_t *level, double *box, int32_t *perm,
                 int32_t *near, int32_t *far);
int pnl_tree_node_cells(const pnl_tree *T, int n, const int32_t *node_ids, int64_t *off, int32_t *cells);
int pnl_h2_transfer_matrices(int nnodes, int dim, int m, const double *box, const int32_t *parent, double *out);
int pnl_nfplan_build(int dim, int nv, const double *vertices, int nc, const int32_t *cells, int dpe, int N, const int32_t *dofs,
                     int nnodes, const int64_t *node_off, const int32_t *node_dofs, const int64_t *node_cell_off,
                     const int32_t *node_cells, int npairs, const int32_t *pair_nodes, int tile, int max_chunk_dofs,
                     pnl_nfplan **out);
void pnl_nfplan_destroy(pnl_nfplan *P);
int pnl_nfplan_sizes(const pnl_nfplan *P, int64_t *out9);
int pnl_nfplan_get(const pnl_nfplan *P, int which, void *dst);

/* sparsity pattern of a finite-horizon operator (getSparse NA:1062-1260): the DoF pairs (I, J) for which a cell holding I and
 * a cell holding J have a vertex pair closer than delta (the pairs getRelativePosition does not call REMOTE,
 * interactionDomains.pyx:875-898); strict_lower: only J < I (SSS storage).  Rows sorted, CSR through the getters. */
typedef struct pnl_pattern pnl_pattern;
int pnl_horizon_pattern(int dim, int nv, const double *vertices, int nc, const int32_t *cells, int dpe, int N, const int32_t *dofs,
                        double delta, int strict_lower, pnl_pattern **out);
/* pattern of the near field of the cluster pairs pairs[npairs][2] (node ids of the tree): union of the blocks n1.dofs x n2.dofs,
 * getSparseNearField NA:3226-3289; strict_lower keeps I > J (SSS) */
int pnl_near_pattern(const pnl_tree *T, int npairs, const int32_t *pairs, int strict_lower, pnl_pattern **out);
/* The row pointer of a pattern is int32 (the reference's INDEX_t; pnl_upload_sparsity and the CSR / SSS operators index with
 * it): both builders return PNL_ERR_UNSUPPORTED for a pattern with more stored entries than the limit instead of wrapping.
 * pnl_pattern_set_max_nnz lowers the limit (process-wide, 1 ... 2^31 - 1; other values leave it unchanged) and returns the old one. */
int64_t pnl_pattern_set_max_nnz(int64_t max_nnz);
int64_t pnl_pattern_nnz(const pnl_pattern *P);
int pnl_pattern_get(const pnl_pattern *P, int32_t *indptr, int32_t *indices);
void pnl_pattern_destroy(pnl_pattern *P);

/* ---- row slab of a rank: distributed dense operator (the reference's DistributedH2Matrix_globalData.matvec,
 *      clusterMethodCy.pyx:3127-3154, on a row partition, tree_node.partition :1854-1896) ------------------------------
 * After pnl_set_row_slab the dense assemble calls write ONE-SIDED into a slab of nrows x ncols doubles (leading dimension
 * ldA >= ncols): row r holds global DoF rowdofs[r], column j global DoF coldofs[j] (both increasing); cross blocks at (c1's
 * DoF, c2's DoF) for c1 < c2, the symmetric local matrices of touching pairs once at (min, max); no mirror pass; the
 * per-cell diagonal blocks stay in the per-cell buffer (pnl_get_diag_blocks: 2 x ncp x ND doubles, the rank's partial
 * sums over ALL cells).  The rows must contain every DoF of the cells in the caller's cell range and of the cells touching
 * them, the columns every DoF of the cells from cell_begin on.  The rank's part of the operator is
 * A' + A'^T - diag(A') + scatter(D); pnl_slab_matvec applies it (y is overwritten; the caller all-reduces y over the
 * ranks).  nrows = 0 restores the full N x N output. */
int pnl_set_row_slab(pnl_context *ctx, int nrows, const int32_t *rowdofs_host, int ncols, const int32_t *coldofs_host);
int pnl_diag_blocks_size(pnl_context *ctx);
int pnl_get_diag_blocks(pnl_context *ctx, double *dst_dev);
int pnl_slab_matvec(pnl_context *ctx, const double *slab_dev, int64_t ld, const double *diag_blocks_dev, const double *x_dev,
                    double *y_dev);
int pnl_slab_diagonal(pnl_context *ctx, const double *slab_dev, int64_t ld, const double *diag_blocks_dev, double *diag_dev);

/* ---- adjacent solve path: Dense_LinearOperator.matvec (dgemv, DenseLinearOperator_{SCALAR}.pxi:14-18)
 *      and cg_solver + jacobi (base/PyNucleus_base/solvers.pyx:363-444, 229-245) ------------------- */
/* y = A x (n x n, row-major, leading dimension ldA); symmetric_half = 1: y = (A + A^T) x for PNL_FLAG_NO_MIRROR storage;
 * symmetric_half = 2: A is stored in full and is symmetric -- the upper triangle is read once for both A x and A^T x (4 n^2 bytes
 * instead of 8 n^2: a GEMV is bound by HBM).  pnl_cg_jacobi always reads the upper triangle (CG needs a symmetric operator). */
int pnl_gemv(pnl_context *ctx, const double *A_dev, int64_t ldA, int n, const double *x_dev, double *y_dev,
             int symmetric_half);
/* Jacobi-preconditioned CG on A x = b; x_dev holds the initial guess and the result.
 * Stops when ||r||_2 <= tol (absolute, like the reference's default) or after maxiter; returns iterations in
 * *iters and the final residual norm in *residual. */
int pnl_cg_jacobi(pnl_context *ctx, const double *A_dev, int64_t ldA, int n, const double *b_dev, double *x_dev,
                  double tol, int maxiter, int *iters, double *residual);

/* ---- solver side on the device: geometric multigrid over a hierarchy of assembled nonlocal operators
 *      (fractionalLevel, nl/PyNucleus_nl/helpers.py:312-380), multigrid-preconditioned CG, theta time stepping -----------
 * y = alpha A x + beta b for a dense row-major nrows x ncols block (LinearOperator.residual = (alpha, beta) = (-1, 1);
 * b_dev may be y_dev) and y = alpha A x + beta y for a CSR matrix on the device (restriction / prolongation
 * multilevelSolver/PyNucleus_multilevelSolver/restriction_*_P1.pxi, mass matrix). */
int pnl_gemv_axpby(pnl_context *ctx, const double *A_dev, int64_t ldA, int nrows, int ncols, const double *x_dev, double alpha,
                   double beta, const double *b_dev, double *y_dev);
int pnl_csr_matvec(pnl_context *ctx, int nrows, const int32_t *indptr_dev, const int32_t *indices_dev, const double *data_dev,
                   const double *x_dev, double alpha, double beta, double *y_dev);
/* One level of the hierarchy, level 0 = coarsest (multigrid_{SCALAR}.pxi:86-135 levelMemory: A, R, P): the dense operator and
 * its diagonal (Jacobi smoother, smoothers_{SCALAR}.pxi:118-131), the restriction to the next coarser level (n_coarse x n)
 * and the prolongation from it (n x n_coarse) as CSR arrays on the device (unused on level 0, where only n counts). */
typedef struct {
    int32_t n;
    int32_t pad;
    const double *A_dev;
    int64_t ldA;
    const double *diag_dev;
    const int32_t *R_indptr_dev, *R_indices_dev;
    const double *R_data_dev;
    const int32_t *P_indptr_dev, *P_indices_dev;
    const double *P_data_dev;
    /* kind 0: the dense operator A_dev.  kind 1 (the FINEST level only): the H2 operator currently set up in the context
     * (pnl_h2_setup): near field as full CSR (near_*_dev, n rows) + the far field through pnl_h2_matvec; A_dev is unused,
     * diag_dev = the diagonal of the near field.  The hierarchies of the reference's drivers put H2 operators on the fine
     * levels (helpers.py:312-380 with matrixFormat 'H2').  kind 2: the dense operator A_dev is SYMMETRIC and stored in full; its
     * products read the upper triangle only (the two-sided sweep of pnl_gemv with symmetric_half = 2). */
    int32_t kind, pad2;
    const int32_t *near_indptr_dev, *near_indices_dev;
    const double *near_data_dev;
} pnl_mg_level_desc;
typedef struct pnl_mg pnl_mg;
/* multigrid.__init__ / setup (:86-235): V cycle, Jacobi smoother with damping omega and presmooth / postsmooth sweeps
 * (defaults of the reference: 2/3, 1, 1), coarse solver = multiplication with the inverse of the coarsest operator
 * (coarse_inverse_dev, n_0 x n_0 row-major; the reference factorises it with LU).  The level arrays stay owned by the caller
 * and must outlive the object. */
int pnl_mg_create(pnl_context *ctx, int nlevels, const pnl_mg_level_desc *levels, const double *coarse_inverse_dev, double omega,
                  int presmooth, int postsmooth, pnl_mg **out);
int pnl_mg_destroy(pnl_mg *mg);
/* one cycle on the finest level (solveOnLevel :237-292); x_is_zero: the first residual is b (simpleResidual) */
int pnl_mg_cycle(pnl_mg *mg, const double *b_dev, double *x_dev, int x_is_zero);
/* multigrid.solve (:296-390): cycles until ||b - A x||_2 <= tol or maxiter; residuals[0..min(iters+1, cap)) = the norms */
int pnl_mg_solve(pnl_mg *mg, const double *b_dev, double *x_dev, double tol, int maxiter, int x_is_zero, int *iters, double *residuals,
                 int residuals_cap);
/* cg_solver.solve (base/PyNucleus_base/solvers.pyx:363-444) preconditioned by one cycle (multigridPreconditioner :470-497);
 * A_dev NULL: the finest operator of the hierarchy; convergence on sqrt(r.Br) like the reference */
int pnl_mg_cg(pnl_mg *mg, const double *A_dev, int64_t ldA, const double *b_dev, double *x_dev, double tol, int maxiter, int x_is_zero,
              int *iters, double *residuals, int residuals_cap);
/* CrankNicolson.step (base/PyNucleus_base/timestepping.py:93-112) for M u_t + S u = g:
 * (M/dt + theta S) u_new = (M/dt) u - (1-theta) S u + forcing, forcing = (1-theta) g(t) + theta g(t+dt) (setRHS :76-91),
 * solved by pnl_mg_cg on the hierarchy mg of M/dt + theta S from the initial guess u (u_dev is overwritten). */
int pnl_theta_step(pnl_mg *mg, const double *S_dev, int64_t ldS, const int32_t *M_indptr_dev, const int32_t *M_indices_dev,
                   const double *M_data_dev, double dt, double theta, const double *forcing_dev, double *u_dev, double tol, int maxiter,
                   int *iters, double *residual);
/* 1/diag(A) into dinv_dev (jacobi_solver.setup, solvers.pyx:233-237) */
int pnl_inv_diagonal(pnl_context *ctx, const double *A_dev, int64_t ldA, int n, double *dinv_dev);

#ifdef __cplusplus
}
#endif
#endif
