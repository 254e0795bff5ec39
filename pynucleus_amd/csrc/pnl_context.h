// Internal: the context behind the C ABI (include/pnl_hip.h), shared by the translation units of libpnl_hip.so.
#pragma once
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <string>
#include <vector>
#include "pnl_hip.h"
#include "pnl_device.h"


constexpr int TILE_P1 = 64;     // cells per block for dpe <= 3
constexpr int TILE_P2 = 32;     // cells per block for dpe == 6 (bigger LDS sub-block per cell)

struct DevBuf {
    void *p = nullptr;
    size_t bytes = 0;
    ~DevBuf() { release(); }
    void release() { if (p) { (void)hipFree(p); p = nullptr; bytes = 0; } }
};


struct pnl_context {
    int device = 0;
    hipStream_t own_stream = nullptr, stream = nullptr;
    // side streams for the per-class passes of a variable order (work list, touching pairs, boundary): the passes only add
    // to A / the diagonal blocks with atomics, so they may overlap; each fills the other's tail (ClassFork in pnl_hip.hip)
    static constexpr int NAUX = 4;      // side streams (the runtime maps streams onto four hardware queues: more streams share them; measured, no gain from six)
    hipStream_t aux[NAUX] = {};
    hipEvent_t ev_fork = nullptr, ev_join[NAUX] = {};
    // recorded behind the fold + mirror pass of a block-slot assembly: from here on A only receives atomic adds, so the touching
    // pairs and the boundary term run on side streams next to the work-list kernels (one order class)
    hipEvent_t ev_fold = nullptr;
    hipEvent_t ev_bnd = nullptr;             // zero fill of the diagonal-block buffer + class tables of the boundary term are in the stream
    bool fold_event_set = false;
    std::string err;
    // host copies
    int dim = 0, nv = 0, nc = 0, dpe = 0, dpv = 0, dped = 0, N = 0, nb = 0, qmax = -1;
    int nreal = 0;                            // cells of positive volume (the others are in-mesh padding, finalize())
    std::vector<int> cell_orig;               // the caller's number of every cell (pnl_set_cell_order), empty = the numbering of the upload
    std::vector<int> real_from;               // [nc + 1] number of real cells with index >= c
    double H0 = 0.;
    std::vector<double> vertices, vol, h;
    std::vector<int32_t> cells, dofs, perm_table, bcells;
    // per order class (one class for a constant order; pnl_set_classes for a piecewise-constant variable order): kernel,
    // order formula, singular rules and the touching pairs that belong to the class
    struct ClassData {
        pnl_kernel kern[2];
        pnl_order_formula form[2];
        bool have_kernel[2] = {false, false}, have_form[2] = {false, false};
        bool have_sing[2][3] = {{false, false, false}, {false, false, false}};
        DevBuf b_sn[3], b_sw[3], b_sp[3], b_bn[2], b_bw[2], b_bp[2], b_spairs[3], b_bpairs[2], b_spairs1[3];
        int sM[3] = {0, 0, 0}, sRows[3] = {0, 0, 0}, bM[2] = {0, 0};
        double sFac = 0., bFac = 0.;
        int n_spairs[3] = {0, 0, 0}, n_bpairs[2] = {0, 0};
        int n_spairs1[3] = {0, 0, 0};   // non-symmetric order: touching pairs (c2, c1) of the second orientation
        ClassData() { std::memset(kern, 0, sizeof(kern)); std::memset(form, 0, sizeof(form)); }
    };
    std::vector<ClassData*> cls;
    int cur = 0;                      // class the setters and launchers currently act on
    ClassData &C() { return *cls[cur]; }
    int nlab = 0;                     // labels of the variable order (0: constant order)
    bool nonsym = false;              // cls_of is not symmetric: both orientations of every pair (pnl_set_nonsymmetric)
    int orient = 0;                   // orientation the launchers currently act on
    std::vector<int32_t> cell_labels, facet_labels, cls_of;
    bool have_mesh = false, have_dofs = false, have_rules = false, have_boundary = false;
    bool dirty = true;
    // device
    DevProblem P;
    DevBuf b_cellv, b_ccen, b_cvol, b_ch, b_clog, b_cvid, b_cdof, b_cslot, b_blk_ndof, b_blk_dofs, b_perm, b_off, b_bary, b_w, b_phi,
        b_foff, b_fbary, b_fw, b_bvid, b_bv, b_bgeo, b_counters, b_D, b_tiles,
        b_vec[6], b_clabel, b_blabel, b_clsof, b_scal, b_wl, b_wlcount, b_tilectr, b_ttn, b_ttoff, b_tttab, b_ttwphi, b_wlsorted, b_wlaux,
        b_vertices, b_sp_indptr, b_sp_indices, b_mp_pairs, b_mp_masks, b_mp_wl, b_mp_sorted, b_mp_aux, b_bi_cells, b_bi_facets,
        b_bi_masks, b_cp[28], b_cpD, b_wlds, b_wlpair, b_h2[20];
    // second-generation tile kernels (pnl_tile2.h): per-class kernel / order-formula tables, rules of the uniform-order tiles,
    // class word of every tile entry (2 class + orientation), w and w phi of the packed rules
    DevBuf b_kcls, b_fcls, b_uni, b_tilecls, b_ttwphif;
    // block-slot storage (pnl_tile2.h): padded column offsets of the blocks, row offsets, copies (block, slot) of every DoF,
    // the tiles that several order classes visit, the storage itself (allocated by the first assembly that uses it)
    DevBuf b_scolbase, b_srowoff, b_cpoff, b_cpslot, b_cprow, b_foldtab, b_bkcls, b_bfcls, b_bdefer, b_multitiles, b_slotA, b_candtiles, b_candq;
    int slot_S = 0, n_multitiles = 0;
    long long slot_total = 0;         // doubles
    // row slab of a rank (pnl_set_row_slab, pnl_slab.hip)
    DevBuf b_rowmap, b_rowdof, b_colmap, b_coldof;
    std::vector<int32_t> slab_rowdofs, slab_coldofs;
    int slab_rows = 0, slab_cols = 0;
    bool slot_full_list = false;      // the current tile list is the whole upper block triangle (pnl_assemble_dense)
    bool slot_used = false;           // the tile kernels of the current assembly wrote the block-slot storage
    std::vector<DevKernel> kcls_host;
    std::vector<DevKernel> bkcls_host;
    std::vector<DevFormula> bfcls_host;
    std::vector<DevFormula> fcls_host;
    int uni_off[5] = {-1, -1, -1, -1, -1}, uni_np[5] = {0, 0, 0, 0, 0};
    // the 3-point rule of P1 has equal weights and shape values w phi_b(y_j) = A + B delta_bj (points reordered to make it so):
    // k_tile_uniform<3, 3, KT, true> forms the cross block from row, column and total sums of the nine kernel values
    bool uni_struct[5] = {false, false, false, false, false};
    H2Dev h2;
    bool have_h2 = false;
    // non-symmetric kernels with an order per quadrature point (pnl_set_order_function)
    bool have_tile_order = false;     // permuted cell tables for the tile kernels (finalize starts the search, tile_order_ready ends it)
    struct TileOrderJob *tile_job = nullptr;
    DevBuf b_cellv_t, b_cdof_t, b_cslot_t, b_Dt;    // b_Dt: diagonal blocks in the tile kernels' local order
    PwDev pw;
    bool have_pw = false, have_pw_rules[2][3] = {{false, false, false}, {false, false, false}};
    int pw_nkeys[2] = {0, 0};
    std::vector<double> pw_cell_smax, pw_facet_smax;
    std::vector<double> pw_vertex_s;           // order function of type 5: values at the mesh vertices
    DevBuf b_pw_cellsv;
    DevBuf b_pw_csm, b_pw_fsm, b_pw_rule[2][3][4], b_pw_pairs, b_pw_bpairs;
    std::vector<std::vector<int>> h2_levels;   // nodes of every level >= 1
    long long h2_vtot = 0;                    // doubles of the leaf values V (pnl_h2_get / _set)
    std::vector<size_t> h2_level_off;
    std::vector<int32_t> rule_off, frule_off;   // host copies of the distant-rule offsets (cell points, facet points) per order
    int sp_nnz = -1;                // near-field sparsity pattern (pnl_upload_sparsity)
    unsigned wl_cap = 0;
    int tile = TILE_P1, nblocks = 0, ncp = 0, nU = 0;
    std::vector<int2> spairs_host[3];
    std::vector<int2> tiles_cached;   // tile list (as given by the caller) currently resident in b_tiles
    size_t tiles_cap = 0;
    // b_tiles holds the mixed tiles first, then the uniform ones (all pairs distant with the lowest order)
    int n_mixed = 0, n_pure = 0, tile_off = 0, tiles_cb = -1, tiles_ce = -1;
    std::vector<int> cls_tile_off, cls_n_mixed, cls_n_pure;     // per order class: its slice of b_tiles (mixed tiles, then uniform)
    std::vector<int> cls_n_uni[3];    // uniform tiles of order 2, 3, 4 per class (cls_n_pure = cls_n_uni[0]); they follow the mixed ones
    // dim 2, dpe 6 (one launch over all classes): b_tiles = [mixed tiles of all classes][order 2][order 3][order 4], b_tilecls alike
    bool single_launch = false;
    int sl_off[4] = {0, 0, 0, 0}, sl_n[4] = {0, 0, 0, 0};
    unsigned wl_cap_each = 0;         // capacity of one work-list region (one region per class in a single-launch assembly)
    int wl_slots = 1;                 // work-list counters in use by the current assembly (one per class / pass)
    std::vector<pnl_order_formula> tiles_forms;
    pnl_order_formula tiles_form;
    bool tiles_filter = true;
    bool use_pure = true;             // debug: PNL_PURE=0 sends every tile through the general kernel
    // tables of the general power (pnl_pow_tab, pnl_common.h), one per (exponent, scale) seen by this context
    struct PowTab { double exponent, scale; DevBuf buf; };
    std::vector<PowTab*> powtabs;
    // tcx / tcy / trad: centre and radius (cell centres + vertex reach) in the coordinates of the interaction transform
    struct BlockAgg { double cx, cy, rad, hmax, hmin, Lmin, Lmax; bool full; double tcx, tcy, trad; };
    // linear transform of the interaction set (ellipse domains, interactionDomains.pyx:1393-1630): the kernel sees |T (x - y)|
    bool have_xform = false;
    double xform[4] = {1., 0., 0., 1.};
    std::vector<BlockAgg> blocks;
    hipEvent_t ev[8] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
    hipEvent_t kev[PNL_NUM_KERNEL_SLOTS][2] = {};   // event pair around every tile-kernel launch (pnl_get_kernel_ms)
    bool kev_set[PNL_NUM_KERNEL_SLOTS] = {};
    bool pure_launched = false;
    bool symflush = false;          // PNL_FLAG_SYMMETRIC_FLUSH of the current assembly
    bool ev_valid = false;
    unsigned long long visited_pairs = 0;
    bool visited_is_assembled = false;      // finite-horizon tiles: every visited (non-REMOTE) pair is an assembled one
    bool tiles_launched = false;
    int ablate = 0;                 // debug: PNL_ABLATE env bits (1 no LDS accumulate, 2 no evaluation)
    bool tile_cell_filter = true;   // apply [cell_begin, cell_end) to the a-cells of the tiles too
    bool wl_lane = true;            // debug: PNL_WL_LANE=0 sends every work-list order to the 16-lanes-per-pair kernel
};


inline int fail(pnl_context *ctx, int code, const char *fmt, ...) {
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof(buf), fmt, ap);
    va_end(ap);
    if (ctx) ctx->err = buf;
    return code;
}

#define HIPCHK(ctx, call)                                                                            \
    do {                                                                                             \
        hipError_t e_ = (call);                                                                      \
        if (e_ != hipSuccess) return fail(ctx, PNL_ERR_HIP, "%s failed: %s", #call, hipGetErrorString(e_)); \
    } while (0)

template <class T>
int upload(pnl_context *ctx, DevBuf &b, const T *src, size_t n) {
    const size_t bytes = std::max<size_t>(n*sizeof(T), 16);
    if (b.bytes < bytes) {
        b.release();
        HIPCHK(ctx, hipMalloc(&b.p, bytes));
        b.bytes = bytes;
    }
    if (n) HIPCHK(ctx, hipMemcpyAsync(b.p, src, n*sizeof(T), hipMemcpyHostToDevice, ctx->stream));
    // host vectors passed here may be temporaries: make the copy synchronous
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    return PNL_OK;
}

inline int ensure(pnl_context *ctx, DevBuf &b, size_t bytes) {
    bytes = std::max<size_t>(bytes, 16);
    if (b.bytes < bytes) {
        b.release();
        HIPCHK(ctx, hipMalloc(&b.p, bytes));
        b.bytes = bytes;
    }
    return PNL_OK;
}

inline DevKernel to_dev(const pnl_kernel &k, int dim) {
    DevKernel d;
    d.ktype = k.ktype;
    d.exponent = k.exponent;
    d.scale = k.scale;
    d.horizon2 = k.horizon2;
    d.interaction = k.interaction;
    // quarter-integer exponents get the rsqrt-based evaluation
    const double m4 = -4.*k.exponent;
    const int qm = (int)std::lround(m4);
    (void)dim;
    d.qm = qm;
    d.fast = (k.ktype == PNL_FRACTIONAL && std::isinf(k.horizon2) && qm >= 1 && qm <= 32 && std::fabs(m4-qm) < 1e-13) ? 1 : 0;
    {
        long double b = 1.L;
        for (int i = 0; i < 6; i++) { b *= ((long double)k.exponent-i)/(long double)(i+1); d.pb[i] = (double)b; }
    }
    d.ptab = nullptr;
    return d;
}

inline DevFormula to_dev(const pnl_order_formula &f) {
    DevFormula d;
    d.c0 = f.c0; d.a = f.a; d.b = f.b; d.e = f.e; d.den0 = f.den0; d.clip = f.clip_num; d.pad = 0;
    return d;
}

// Options of the library.  The product build reads NO environment variable on the assembly path: an option exists only after
// pnl_set_option(name, value) named it (include/pnl_hip.h lists the options a product build accepts -- the hooks the parity
// tests use to reach the alternative code paths).  Tuning builds (make EXTRA=-DPNL_TUNING) accept every name and fall back to the
// environment, which is how the A/B measurements of DESIGN.md were made.  Returns the value string or nullptr.
const char *pnl_tune(const char *name);
// grid of a persistent tile kernel, capped by the option PNL_TILE_WGS (tests)
inline int pnl_grid_cap(int grid) {
    const char *e = pnl_tune("PNL_TILE_WGS");
    const int cap = e ? atoi(e) : 0;
    return cap > 0 ? std::max(1, std::min(grid, cap)) : grid;
}

// row stride of the LDS sub-block: nU + 1 columns (+1: trash column / row for boundary DoFs); PNL_ACC_PAD=m rounds it up
// to 1 mod m so that consecutive rows start in different LDS banks
inline int acc_stride_of(int nU, size_t fixed_bytes = 0) {
    int st = nU+1;
    // rows that start in different LDS banks (stride = 1 mod 32 doubles) see fewer conflicts in the ds_add_f64 of the
    // cross blocks (measured: -0.4 ms at noRef 6), if the bigger sub-block still leaves two workgroups per CU
    const int m = pnl_tune("PNL_ACC_PAD") ? atoi(pnl_tune("PNL_ACC_PAD")) : 32;
    if (m > 1) {
        int padded = st;
        while (padded % m != 1) padded++;
        if (fixed_bytes+sizeof(double)*(size_t)(nU+1)*padded <= 80*1024) st = padded;
        else if (st % 2 == 0 && fixed_bytes+sizeof(double)*(size_t)(nU+1)*(st+1) <= 80*1024) st++;    // at least an odd stride
    }
    return st;
}

// the problem description the dense tile kernels see: cell tables in the conflict-reducing local vertex order
inline DevProblem tile_problem(const pnl_context *ctx) {
    DevProblem Pt = ctx->P;
    if (ctx->have_tile_order) {
        Pt.cellv = (const double*)ctx->b_cellv_t.p; Pt.cdof = (const int*)ctx->b_cdof_t.p; Pt.cslot = (const short*)ctx->b_cslot_t.p;
    }
    return Pt;
}


inline void kt_begin(pnl_context *ctx, int slot) { (void)hipEventRecord(ctx->kev[slot][0], ctx->stream); }
inline void kt_end(pnl_context *ctx, int slot) { (void)hipEventRecord(ctx->kev[slot][1], ctx->stream); ctx->kev_set[slot] = true; }

// pnl_hip.hip: joins the vertex-order search finalize() started and uploads the permuted cell tables (sets have_tile_order)
int pnl_tile_order_ready(pnl_context *ctx);

// pnl_hip.hip / pnl_pwnear.hip: kernels with an order per quadrature point
int pnl_pw_prepare(pnl_context *ctx, int need_boundary);
int pnl_pw_h2_interp(pnl_context *ctx);

// pnl_gemv2.hip: one pass over a row-major block for both A x and A^T x
int pnl_launch_gemv_symmetric(pnl_context *ctx, const double *A, long long ldA, int n, const double *x, double alpha, double beta,
                              const double *b, double *y);
int pnl_launch_slab_two_sided(pnl_context *ctx, const double *slab, long long ld, int nrows, int ncols, const int *rowdof, const int *coldof,
                              const double *x, double *y);

// pnl_tile2.hip
int pnl2_launch_uniform(pnl_context *ctx, int kt, const DevProblem &Pt, const int2 *tiles, const int *tile_cls, int ntiles, int q,
                        double *A, int64_t ldA, double *Dglob, const SlotOut &SO);
int pnl2_launch_p2(pnl_context *ctx, int kt, const int2 *tiles, const int *tile_cls, int ntiles, double *A, int64_t ldA,
                   int cell_begin, int cell_end, unsigned wl_cap_each, const SlotOut &SO);
int pnl2_zero_slot_tiles(pnl_context *ctx, const SlotOut &SO);
int pnl2_fold_mirror(pnl_context *ctx, const SlotOut &SO, double *A, int64_t ldA);
inline SlotOut slot_out(const pnl_context *ctx) {
    SlotOut SO;
    SO.A2 = (double*)ctx->b_slotA.p; SO.rowoff = (const long long*)ctx->b_srowoff.p; SO.colbase = (const int*)ctx->b_scolbase.p;
    SO.S = ctx->slot_S; SO.nU = ctx->nU;
    return SO;
}
