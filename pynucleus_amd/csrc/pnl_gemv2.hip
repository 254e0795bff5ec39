// Two-sided GEMV: ONE pass over a row-major block produces both  y[R(r)] += sum_c S[r][c] x[C(c)]  and  y[C(c)] += sum_r S[r][c] x[R(r)]
// (gfx950 only).
//
// Reference: Dense_LinearOperator.matvec (base/PyNucleus_base/DenseLinearOperator_{SCALAR}.pxi:14-18: dgemv on the full block) inside
// the CG loop of the drivers (solvers.pyx:363-444), and the local products of the distributed operator
// (clusterMethodCy.pyx:3127-3154).  A GEMV is bound by HBM: the symmetric operator needs its upper triangle once, 4 N^2 bytes
// instead of 8 N^2; a rank's one-sided slab A' is applied as A' x and A'^T x in the same sweep instead of
// two.
//
// Workgroup = 64 rows x 4096 columns; its four waves split the COLUMNS (1024 each) and walk the same 64 rows, so a row is read as
// 32 contiguous KB.  A lane owns 16 columns (eight non-temporal 16-byte loads per row, all in flight), keeps their x values and
// their column sums in registers -- they leave as one atomic per column and workgroup, 1 / 64 of an atomic per entry read; the row
// sum is a wave reduction, the four waves' parts meet in LDS.  The grid holds the upper block triangle only.  Measured at
// N = 48,769 (tools/gemv_probe.py): 1.94 ms = 4.9 TB/s on 4 N^2 bytes against 3.15 ms = 6.05 TB/s on 8 N^2 for the one-sided
// k_gemv; ablations (no wave reduction, no atomics: 2.03 of 2.12 ms) and the tile shape (waves over rows: the same) change nothing,
// streaming loads do (2.12 -> 1.94 ms).  No MFMA: four FMAs per 16 bytes is a hundredth of the vector rate.
#include "pnl_context.h"
#include "pnl_common.h"

namespace {

constexpr int G2_RB = 64, G2_WC = 1024, G2_CB = 4*G2_WC;     // rows per workgroup, columns per wave, columns per workgroup

// SLAB: rows / columns of S are the DoFs rowdof[r] / coldof[c] (both increasing); the transposed sweep leaves out the entries with
// rowdof[r] == coldof[c] (the slab holds every symmetric contribution once, its diagonal belongs to the first sweep).
// !SLAB: S = the full symmetric matrix, R = C = identity; only entries with c >= r are read: both sweeps for c > r, the first for c == r.
// dense: the grid holds the blocks of the upper block triangle only, row block after row block: block t -> (row block bi, column block
// ci >= first(bi) = bi RB / CB).  CB / RB row blocks share their first column block, so the prefix sum has a closed form per group.
__device__ __forceinline__ void g2_upper_block(int t, int ncb, int &bi, int &ci) {
    constexpr int GR = G2_CB/G2_RB;                              // row blocks per group
    int g = 0, base = 0;
    while (t >= base+GR*(ncb-g)) { base += GR*(ncb-g); g++; }   // at most ncb steps, uniform over the workgroup
    const int r = (t-base)/(ncb-g);
    bi = g*GR+r;
    ci = g+(t-base)-r*(ncb-g);
}
__host__ __device__ inline long long g2_upper_blocks(int nrb, int ncb) {
    constexpr int GR = G2_CB/G2_RB;
    long long tot = 0;
    for (int b = 0; b < nrb; b++) tot += ncb-b/GR;
    return tot;
}

template <bool SLAB>
__global__ void __launch_bounds__(PNL_NTHREADS, 4)
k_gemv_two_sided(const double *__restrict__ S, long long ld, int nrows, int ncols, const int *__restrict__ rowdof,
                 const int *__restrict__ coldof, const double *__restrict__ x, double alpha, double *__restrict__ y) {
    __shared__ double s_row[4][G2_RB];
    int bi = blockIdx.y, ci = blockIdx.x;
    if (!SLAB) g2_upper_block(blockIdx.x, (ncols+G2_CB-1)/G2_CB, bi, ci);
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int r0 = bi*G2_RB, c0 = ci*G2_CB+w*G2_WC;              // this wave's columns: [c0, c0 + 1024)
    const int rend = min(nrows, r0+G2_RB), cend = min(ncols, c0+G2_WC);
    // masks only where the wave's columns meet the diagonal (dense) / where a row DoF can equal a column DoF (slab), or at the ragged edge
    bool plain = cend == c0+G2_WC && ((((uintptr_t)S) | ((uintptr_t)(ld*sizeof(double)))) & 15) == 0;
    if (SLAB) plain = plain && c0 < ncols && (rowdof[rend-1] < coldof[c0] || rowdof[r0] > coldof[cend-1]);
    else plain = plain && c0 >= r0+G2_RB;
    const bool none = c0 >= ncols || (!SLAB && c0+G2_WC <= r0);   // nothing of this wave's strip counts (it still joins the row reduction)
    double xc[16], cacc[16];
    int J[16];
#pragma unroll
    for (int k = 0; k < 8; k++)
#pragma unroll
        for (int h = 0; h < 2; h++) {
            const int j = c0+2*lane+128*k+h;
            J[2*k+h] = (j < cend && !none) ? (SLAB ? coldof[j] : j) : -1;
            xc[2*k+h] = J[2*k+h] >= 0 ? x[J[2*k+h]] : 0.;
            cacc[2*k+h] = 0.;
        }
    for (int rr = 0; rr < G2_RB; rr++) {
        const int r = r0+rr;
        if (r >= rend) break;
        double rs = 0.;
        if (!none) {
            const int I = SLAB ? rowdof[r] : r;
            const double xr = x[I];
            const double *__restrict__ a = S+(long long)r*ld+c0+2*lane;
            if (plain) {
                double2 v[8];
#pragma unroll
                for (int k = 0; k < 8; k++) {
                    typedef double g2_d2 __attribute__((ext_vector_type(2)));
                    const g2_d2 t = __builtin_nontemporal_load((const g2_d2*)(a+128*k));
                    v[k].x = t.x; v[k].y = t.y;
                }
#pragma unroll
                for (int k = 0; k < 8; k++) {
                    rs = __builtin_fma(v[k].x, xc[2*k], rs);
                    rs = __builtin_fma(v[k].y, xc[2*k+1], rs);
                    cacc[2*k] = __builtin_fma(v[k].x, xr, cacc[2*k]);
                    cacc[2*k+1] = __builtin_fma(v[k].y, xr, cacc[2*k+1]);
                }
            } else {
                // strips on the diagonal (dense) / where a row DoF can be a column DoF (slab), and the ragged right edge: the same 16-byte
                // loads where both entries exist, the triangle / diagonal conditions as selects (the entries below the diagonal of the full
                // matrix are there to be loaded, they just do not count)
                const bool al = ((((uintptr_t)a) & 15) == 0);
#pragma unroll
                for (int k = 0; k < 8; k++) {
                    const int j0 = c0+2*lane+128*k;
                    double2 v = make_double2(0., 0.);
                    if (al && j0+1 < cend) v = *(const double2*)(a+128*k);
                    else { if (j0 < cend) v.x = a[128*k]; if (j0+1 < cend) v.y = a[128*k+1]; }
                    const int ja = J[2*k], jb = J[2*k+1];
                    const double fa = (SLAB ? true : ja >= I) ? v.x : 0., fb = (SLAB ? true : jb >= I) ? v.y : 0.;
                    const double sa = (SLAB ? ja != I : ja > I) ? v.x : 0., sb = (SLAB ? jb != I : jb > I) ? v.y : 0.;
                    rs = __builtin_fma(fa, xc[2*k], rs);
                    rs = __builtin_fma(fb, xc[2*k+1], rs);
                    cacc[2*k] = __builtin_fma(sa, xr, cacc[2*k]);
                    cacc[2*k+1] = __builtin_fma(sb, xr, cacc[2*k+1]);
                }
            }
        }
        rs = wave_sum(rs);
        if (lane == 0) s_row[w][rr] = rs;
    }
    // this wave owns its columns: their sums leave from registers
#pragma unroll
    for (int k = 0; k < 16; k++)
        if (J[k] >= 0 && cacc[k] != 0.) atomic_add_f64(&y[J[k]], alpha*cacc[k]);
    __syncthreads();
    if (tid < rend-r0) {
        const double s = (s_row[0][tid]+s_row[1][tid])+(s_row[2][tid]+s_row[3][tid]);
        if (s != 0.) atomic_add_f64(&y[SLAB ? rowdof[r0+tid] : r0+tid], alpha*s);
    }
}

// y = beta b  (b may be y)
__global__ void __launch_bounds__(PNL_NTHREADS)
k_g2_scaled_copy(int n, double beta, const double *b, double *y) {
    const int i = blockIdx.x*PNL_NTHREADS+threadIdx.x;
    if (i < n) y[i] = beta*b[i];
}

}  // namespace

// y = alpha A x + beta b for a symmetric matrix stored in full: reads the upper triangle only (b may be y or, with beta = 0, null;
// x must not be y)
int pnl_launch_gemv_symmetric(pnl_context *ctx, const double *A, long long ldA, int n, const double *x, double alpha, double beta,
                              const double *b, double *y) {
    if (beta == 0. || !b) HIPCHK(ctx, hipMemsetAsync(y, 0, sizeof(double)*(size_t)n, ctx->stream));
    else if (!(b == y && beta == 1.)) {
        hipLaunchKernelGGL(k_g2_scaled_copy, dim3((n+PNL_NTHREADS-1)/PNL_NTHREADS), dim3(PNL_NTHREADS), 0, ctx->stream, n, beta, b, y);
        HIPCHK(ctx, hipGetLastError());
    }
    const int nrb = (n+G2_RB-1)/G2_RB, ncb = (n+G2_CB-1)/G2_CB;
    hipLaunchKernelGGL((k_gemv_two_sided<false>), dim3((unsigned)g2_upper_blocks(nrb, ncb)), dim3(PNL_NTHREADS), 0, ctx->stream, A, ldA, n, n,
                       (const int*)nullptr, (const int*)nullptr, x, alpha, y);
    HIPCHK(ctx, hipGetLastError());
    return PNL_OK;
}

// y += (A' + A'^T - diag(A')) x for a rank's one-sided slab (y is NOT zeroed: the per-cell diagonal blocks add into it as well)
int pnl_launch_slab_two_sided(pnl_context *ctx, const double *slab, long long ld, int nrows, int ncols, const int *rowdof, const int *coldof,
                              const double *x, double *y) {
    hipLaunchKernelGGL((k_gemv_two_sided<true>), dim3((ncols+G2_CB-1)/G2_CB, (nrows+G2_RB-1)/G2_RB), dim3(PNL_NTHREADS), 0, ctx->stream, slab, ld,
                       nrows, ncols, rowdof, coldof, x, 1., y);
    HIPCHK(ctx, hipGetLastError());
    return PNL_OK;
}
