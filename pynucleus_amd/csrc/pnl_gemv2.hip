// Two-sided GEMV: ONE pass over a row-major block produces both  y[R(r)] += sum_c S[r][c] x[C(c)]  and  y[C(c)] += sum_r S[r][c] x[R(r)]
// (gfx950 only).
//
// Reference: Dense_LinearOperator.matvec (base/PyNucleus_base/DenseLinearOperator_{SCALAR}.pxi:14-18: dgemv on the full block) inside
// the CG loop of the drivers (solvers.pyx:363-444), and the local products of the distributed operator
// (clusterMethodCy.pyx:3127-3154).  A GEMV is bound by HBM: the symmetric operator needs its upper triangle once, 4 N^2 bytes
// instead of 8 N^2; a rank's one-sided slab A' is applied as A' x and A'^T x in the same sweep instead of
// two.
//
// Workgroup = 64 rows x 1024 columns, four waves of 16 rows each.  A lane owns 16 columns (eight 16-byte loads per row, all in
// flight), keeps their x values and their column sums in registers; the row sum is a wave reduction; the column sums of the four
// waves meet in LDS and leave as one atomic per column and workgroup, 1 / 64 of an atomic per entry read.  No MFMA: four FMAs per
// 16 bytes is a hundredth of the vector rate.
#include "pnl_context.h"
#include "pnl_common.h"

namespace {

constexpr int G2_RB = 64, G2_CB = 1024, G2_RW = G2_RB/4;

// SLAB: rows / columns of S are the DoFs rowdof[r] / coldof[c] (both increasing); the transposed sweep leaves out the entries with
// rowdof[r] == coldof[c] (the slab holds every symmetric contribution once, its diagonal belongs to the first sweep).
// !SLAB: S = the full symmetric matrix, R = C = identity; only entries with c >= r are read: both sweeps for c > r, the first for c == r.
template <bool SLAB>
__global__ void __launch_bounds__(PNL_NTHREADS)
k_gemv_two_sided(const double *__restrict__ S, long long ld, int nrows, int ncols, const int *__restrict__ rowdof,
                 const int *__restrict__ coldof, const double *__restrict__ x, double *__restrict__ y) {
    __shared__ double s_col[4][G2_CB];
    const int r0 = blockIdx.y*G2_RB, c0 = blockIdx.x*G2_CB;
    if (!SLAB && c0+G2_CB <= r0) return;                         // below the diagonal: the mirror image is read instead
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int rend = min(nrows, r0+G2_RB), cend = min(ncols, c0+G2_CB);
    // masks only where the block meets the diagonal (dense) / where a row DoF can equal a column DoF (slab), or at the ragged edge
    bool plain = cend == c0+G2_CB && ((((uintptr_t)S) | ((uintptr_t)(ld*sizeof(double)))) & 15) == 0;
    if (SLAB) plain = plain && (rowdof[rend-1] < coldof[c0] || rowdof[r0] > coldof[cend-1]);
    else plain = plain && c0 >= r0+G2_RB;
    double xc[16], cacc[16];
    int J[16];
#pragma unroll
    for (int k = 0; k < 8; k++)
#pragma unroll
        for (int h = 0; h < 2; h++) {
            const int j = c0+2*lane+128*k+h;
            J[2*k+h] = j < cend ? (SLAB ? coldof[j] : j) : -1;
            xc[2*k+h] = j < cend ? x[J[2*k+h]] : 0.;
            cacc[2*k+h] = 0.;
        }
    for (int rr = 0; rr < G2_RW; rr++) {
        const int r = r0+w*G2_RW+rr;
        if (r >= rend) break;
        const int I = SLAB ? rowdof[r] : r;
        const double xr = x[I];
        const double *__restrict__ a = S+(long long)r*ld+c0+2*lane;
        double rs = 0.;
        if (plain) {
            double2 v[8];
#pragma unroll
            for (int k = 0; k < 8; k++) v[k] = *(const double2*)(a+128*k);
#pragma unroll
            for (int k = 0; k < 8; k++) {
                rs = __builtin_fma(v[k].x, xc[2*k], rs);
                rs = __builtin_fma(v[k].y, xc[2*k+1], rs);
                cacc[2*k] = __builtin_fma(v[k].x, xr, cacc[2*k]);
                cacc[2*k+1] = __builtin_fma(v[k].y, xr, cacc[2*k+1]);
            }
        } else {
            // blocks on the diagonal (dense) / where a row DoF can be a column DoF (slab), and the ragged right edge: the same 16-byte
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
        rs = wave_sum(rs);
        if (lane == 0 && rs != 0.) atomic_add_f64(&y[I], rs);
    }
#pragma unroll
    for (int k = 0; k < 8; k++) {
        s_col[w][2*lane+128*k] = cacc[2*k];
        s_col[w][2*lane+128*k+1] = cacc[2*k+1];
    }
    __syncthreads();
    for (int t = tid; t < G2_CB; t += PNL_NTHREADS) {
        const int j = c0+t;
        if (j >= cend) break;
        const double s = (s_col[0][t]+s_col[1][t])+(s_col[2][t]+s_col[3][t]);
        if (s != 0.) atomic_add_f64(&y[SLAB ? coldof[j] : j], s);
    }
}

}  // namespace

// y = A x for a symmetric matrix stored in full: reads the upper triangle only (y is zeroed here)
int pnl_launch_gemv_symmetric(pnl_context *ctx, const double *A, long long ldA, int n, const double *x, double *y) {
    HIPCHK(ctx, hipMemsetAsync(y, 0, sizeof(double)*(size_t)n, ctx->stream));
    hipLaunchKernelGGL((k_gemv_two_sided<false>), dim3((n+G2_CB-1)/G2_CB, (n+G2_RB-1)/G2_RB), dim3(PNL_NTHREADS), 0, ctx->stream, A, ldA, n, n,
                       (const int*)nullptr, (const int*)nullptr, x, y);
    HIPCHK(ctx, hipGetLastError());
    return PNL_OK;
}

// y += (A' + A'^T - diag(A')) x for a rank's one-sided slab (y is NOT zeroed: the per-cell diagonal blocks add into it as well)
int pnl_launch_slab_two_sided(pnl_context *ctx, const double *slab, long long ld, int nrows, int ncols, const int *rowdof, const int *coldof,
                              const double *x, double *y) {
    hipLaunchKernelGGL((k_gemv_two_sided<true>), dim3((ncols+G2_CB-1)/G2_CB, (nrows+G2_RB-1)/G2_RB), dim3(PNL_NTHREADS), 0, ctx->stream, slab, ld,
                       nrows, ncols, rowdof, coldof, x, y);
    HIPCHK(ctx, hipGetLastError());
    return PNL_OK;
}
