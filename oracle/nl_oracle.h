/* CPU oracle for the nonlocal element-pair assembly path -- TEST INFRASTRUCTURE ONLY.
 *
 * Plain-C, single-threaded restatement of the reference algorithm (PyNucleus_nl,
 * Cython).  Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may
 * load this library; the product path (pynucleus_amd + libpnl_hip.so) never does.
 *
 * Parity status: the reference cannot be built or imported here (mpi4py, modepy,
 * meshpy missing), so this oracle is pinned by the reference's own known-answer
 * tests and stored accuracy numbers (tests/test_oracle_pinning.py), not by running
 * the original binary.  Distant-pair triangle rules come from modepy in the
 * reference (un-vendored, version unpinned): entry-wise parity with the original is
 * "unpinned" for those pairs; everything else follows the cited lines literally.
 */
#ifndef NL_ORACLE_H
#define NL_ORACLE_H
#include <stdint.h>

#define NLO_MAX_ORDER 120   /* nonlocalOperator.pyx:107 MAX_PANEL */
#define NLO_IGNORED (-6)    /* panelTypes.pxi */

typedef struct {
    double c0, a, b, e, den0;   /* order = max(ceil((c0 + a*L_other + b*Lmax - e*logdh_other)/(max(logdh_self,0)+den0)), 2) */
    int32_t clip_num;           /* clip logdh at 0 in the numerator too (2D boundary) */
    int32_t pad;
} nlo_order_formula;

typedef struct {
    int32_t ktype;              /* 0 fractional, 1 indicator, 2 peridynamic */
    int32_t interaction;        /* finite horizon only: 1 ball2_retriangulation, 2 ball2_barycenter (interactionDomains.pyx) */
    double exponent;            /* power of |x-y|^2 */
    double scale;
    double horizon2;            /* inf = no truncation */
} nlo_kernel;

typedef struct nlo_problem_s {
    int32_t dim, dpe, nc, nv, num_dofs, dofs_per_vertex, dofs_per_edge, pad0;
    const double *vertices;       /* [nv][dim] */
    const int32_t *cells;         /* [nc][dim+1] */
    const int32_t *dofs;          /* [nc][dpe], negative = boundary DoF */
    const double *vol, *h;        /* [nc] */
    double H0;
    const int32_t *dof_perm_table;/* [(dim+1)!][dpe], row = Lehmer rank of the vertex permutation */

    nlo_kernel kernel;
    nlo_order_formula qo;

    /* distant rules, orders 0..qmax (absent orders have zero points) */
    int32_t qmax, pad1;
    const int32_t *dist_off;      /* [qmax+2] */
    const double *dist_bary;      /* [total][3] (third column unused in 1D) */
    const double *dist_w;         /* [total] */
    const double *dist_phi;       /* [total][dpe] shape functions at the nodes */

    /* singular rules; slot 0: common vertex, 1: common edge, 2: common face (2D only) */
    int32_t sing_M[3], sing_rows[3];
    const double *sing_nodes[3];  /* [2(dim+1)][M] */
    const double *sing_w[3];      /* [M] */
    const double *sing_psi[3];    /* [rows][M] */
    double sing_fac;              /* 4 in 2D (FL2:851), 1 in 1D (FL1:374) */

    /* zeroExterior: cells x boundary facets */
    int32_t nb, pad2;
    const int32_t *bcells;        /* [nb][dim] vertex ids of boundary edges (2D) / points (1D) */
    nlo_kernel bkernel;
    nlo_order_formula bqo;
    const int32_t *bfacet_off;    /* [qmax+2] facet rule (Gauss on the edge / single point) */
    const double *bfacet_bary;    /* [total][2] */
    const double *bfacet_w;       /* [total] */
    int32_t bsing_M[2], bpad[2];  /* slot 0: common vertex, 1: common edge */
    const double *bsing_nodes[2]; /* [(dim+1)+dim][M] */
    const double *bsing_w[2];
    const double *bsing_phi[2];   /* [dpe][M] */
    double bsing_fac;             /* -2 in 2D (FL2:1375), +1 in 1D */

    /* variable order, piecewise constant per element pair (NO:509-513 evalParams at the two centres, FL2:664/688 near
     * rules keyed by the pair's singularity): nclasses == 0 for a constant order; otherwise classes[k] holds the
     * kernel / order formula / singular rules of the k-th distinct order value and a pair (c1, c2) belongs to class
     * cls_of[cell_labels[c1]*num_labels + cell_labels[c2]] (cell/facet pairs: facet_labels[b] for the second index) */
    int32_t nclasses, num_labels;
    const struct nlo_problem_s *classes;
    const int32_t *cell_labels, *facet_labels, *cls_of;

    /* non-symmetric kernels with an order s(x) evaluated per quadrature point (fractionalLaplacian{1,2}D_nonsym with
     * piecewise == False): pw_type 0 = off, 1 constant, 2 smoothStep(x0), 3 linearStep(x0), 4 smoothStepRadial
     * (fractionalOrders.pyx:338-540; pw_p = sl, sr, r, interface | radius, slope).  Near rules are keyed by the pair's
     * order (the reference keys its dictionary by the singularity value, FL2:957): pw_keys sorted, tables [nkeys][...]
     * with the point count / row count of sing_M / sing_rows (bsing_M for the boundary twin). */
    int32_t pw_type, pw_normalized;
    double pw_p[6];
    double pw_c0, pw_bc0;             /* constant term of the interior / boundary order formula */
    int32_t pw_nkeys, pw_nbkeys;
    const double *pw_keys, *pw_bkeys;
    const double *pw_nodes[3], *pw_w[3], *pw_phi0[3], *pw_phi1[3];
    const double *pw_bnodes[2], *pw_bw[2], *pw_bphi[2];
    double xform[4];            /* interaction set = linear image of the l2 ball (ellipse domains, interactionDomains.pyx:1393-1630): the
                                 * kernel and the cut-element geometry see T (x - y); has_xform = 0: identity */
    int32_t has_xform, pad3;
    const double *pw_vertex_s;  /* pw_type 5 (feFractionalOrder, fractionalOrders.pyx:541-587, 660-668): values of a P1 function at the mesh vertices */
} nlo_problem;

/* counters[0] pairs visited, [1] pairs assembled (panel != IGNORED, not all-boundary),
 * [2] kernel evaluations, [3] boundary pairs, [4] boundary kernel evaluations,
 * [8 .. 8+NLO_MAX_ORDER) histogram of distant orders, [8+NLO_MAX_ORDER .. +3) vertex/edge/face counts */
#define NLO_NUM_COUNTERS (8 + NLO_MAX_ORDER + 3)

int nlo_panel(const nlo_problem *P, int c1, int c2, int *perm1, int *perm2, int *perm);
void nlo_eval(const nlo_problem *P, int c1, int c2, int panel, const int *perm1, const int *perm2, const int *perm,
              double *contrib, int64_t *nevals);
int nlo_panel_boundary(const nlo_problem *P, int c1, int b, int *perm1, int *perm2, int *perm);
void nlo_eval_boundary(const nlo_problem *P, int c1, int b, int panel, const int *perm1, const int *perm2, const int *perm,
                       double *contrib, int64_t *nevals);
/* A[num_dofs*num_dofs] is accumulated into (caller zeroes).  seconds[0] interior, seconds[1] zeroExterior. */
int nlo_get_dense(const nlo_problem *P, double *A, int zero_exterior, int cell_start, int cell_end,
                  int64_t *counters, double *seconds);
/* same loops, only counting (for sampling the cost of a sub-range without storing A): A may be NULL */
int nlo_get_dense_rows(const nlo_problem *P, double *A, int zero_exterior, int cell_start, int cell_end,
                       int64_t *counters, double *seconds, int store);
/* non-symmetric local matrices (symmetricCells == symmetricLocalMatrix == False): both orientations of every pair are
 * evaluated and scattered with addToMatrixElemElem (NA:1411-1428, 222-253); counters as nlo_get_dense */
int nlo_get_dense_nonsym(const nlo_problem *P, double *A, int zero_exterior, int cell_start, int cell_end,
                         int64_t *counters, double *seconds, int store);
double nlo_pw_order(const nlo_problem *P, const double *x);
double nlo_pw_svalue(const nlo_problem *P, int c1, int c2);
/* H2 near field (NA:1663-1964): masked interior pairs (NA:1812-1832 with the scatter NA:503-520) into CSR (diag == NULL)
 * or SSS (strict lower triangle in data + diag); entries absent from the pattern are dropped like the reference's addToEntry.
 * pairs[np][2] with c1 <= c2, masks[np][4] (256-bit MASK_t).  counters: [0] pairs, [1] assembled, [2] kernel evaluations. */
int nlo_assemble_pairs_masked(const nlo_problem *P, int np, const int32_t *pairs, const uint64_t *masks, const int32_t *indptr,
                              const int32_t *indices, double *data, double *diag, int64_t *counters);
/* cluster-local Gauss-theorem term (NA:1842-1889) and the global one with fac = +-1 (NA:1896-1913, 1945-1964):
 * items (cell, facet vertex ids[dim], mask over the dpe(dpe+1)/2 entries), scatter NA:534-546 */
int nlo_assemble_boundary_masked(const nlo_problem *P, int ni, const int32_t *cells, const int32_t *facets, const uint32_t *masks,
                                 double fac, const int32_t *indptr, const int32_t *indices, double *data, double *diag);
/* the same loop for non-symmetric kernels (symmetricCells == symmetricLocalMatrix == False): ORDERED pairs, masks over the
 * (2 dpe)^2 entries (k = p (2 dpe) + q), unsymmetric CSR only; counters as nlo_get_dense_nonsym (NLO_NUM_COUNTERS entries) */
int nlo_assemble_pairs_masked_nonsym(const nlo_problem *P, int np, const int32_t *pairs, const uint64_t *masks, const int32_t *indptr,
                                     const int32_t *indices, double *data, int64_t *counters);
#endif
