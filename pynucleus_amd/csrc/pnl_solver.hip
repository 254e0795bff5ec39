// Solver side of the nonlocal operators (gfx950 only): geometric multigrid on a hierarchy of dense nonlocal operators,
// multigrid-preconditioned CG and the theta time stepper of the fractional heat equation.
//
// Reference (Cython / Python, CPU):
//   multilevelSolver/PyNucleus_multilevelSolver/multigrid_{SCALAR}.pxi:237-292   multigrid.solveOnLevel (the cycle)
//                                                                    :296-390   multigrid.solve (stationary iteration)
//                                                                    :470-497   multigridPreconditioner
//   multilevelSolver/PyNucleus_multilevelSolver/smoothers_{SCALAR}.pxi:88-108   separableSmoother.eval
//                                                                    :118-131   jacobiPreconditioner (omega / D)
//   base/PyNucleus_base/solvers.pyx:363-444                                      cg_solver.solve
//   base/PyNucleus_base/timestepping.py:64-112                                   CrankNicolson.step (theta method)
//   nl/PyNucleus_nl/helpers.py:312-380                                           fractionalLevel (one assembled operator per level)
//
// Everything here is HBM-bound: a cycle reads the finest operator three times (pre-smoothing residual, residual, post-
// smoothing residual), 8 n^2 bytes each.  The GEMV keeps four 16-byte loads per lane in flight and fuses the vector
// epilogue (alpha A x + beta b) so that no n-vector makes an extra round trip; the coarse levels are launch-bound and
// run as a fixed sequence of small launches on the context's stream, no host synchronisation inside a cycle.
#include "pnl_context.h"
#include "pnl_common.h"

struct pnl_mg {
    pnl_context *ctx = nullptr;
    int nlevels = 0;
    std::vector<pnl_mg_level_desc> lv;
    const double *coarse_inv = nullptr;
    double omega = 2./3.;
    int pre = 1, post = 1;
    // per level: invD (omega / diag), rhs, sol, temp
    std::vector<DevBuf> invD, rhs, sol, temp;
    DevBuf r, p, Ap, z, scal, work, h2tmp;
    double last_conv = 0.;          // preconditioned residual norm sqrt(r.Br) at the end of the last pnl_mg_cg
};

namespace {

// y[row] = alpha * sum_j A[row][j] x[j] + beta * b[row]   (one wave per row; b may be y)
__global__ void __launch_bounds__(PNL_NTHREADS)
k_gemv_axpby(const double *__restrict__ A, long long ldA, int nrows, int ncols, const double *__restrict__ x, double alpha, double beta,
             const double *b, double *y) {
    const int lane = threadIdx.x & 63;
    const int row = (blockIdx.x*PNL_NTHREADS+threadIdx.x) >> 6;
    if (row >= nrows) return;
    const double *__restrict__ a = A+(long long)row*ldA;
    double s0 = 0., s1 = 0., s2 = 0., s3 = 0.;
    const bool aligned = ((((uintptr_t)a) | ((uintptr_t)x)) & 15) == 0;
    if (aligned) {
        const int n2 = ncols >> 1;
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
            const double2 v = a2[j], w = x2[j];
            s0 = __builtin_fma(v.x, w.x, s0); s1 = __builtin_fma(v.y, w.y, s1);
        }
        if ((ncols & 1) && lane == 0) s0 = __builtin_fma(a[ncols-1], x[ncols-1], s0);
    } else {
        for (int j = lane; j < ncols; j += 64) s0 = __builtin_fma(a[j], x[j], s0);
    }
    const double s = wave_sum((s0+s1)+(s2+s3));
    if (lane == 0) y[row] = beta != 0. ? __builtin_fma(alpha, s, beta*b[row]) : alpha*s;
}

// y[row] = alpha * sum_k data[k] x[indices[k]] + beta * y[row]   (restriction, prolongation, mass matrix: a few entries per row)
__global__ void __launch_bounds__(PNL_NTHREADS)
k_csr_axpby(int nrows, const int *__restrict__ indptr, const int *__restrict__ indices, const double *__restrict__ data,
            const double *__restrict__ x, double alpha, double beta, double *y) {
    const int row = blockIdx.x*PNL_NTHREADS+threadIdx.x;
    if (row >= nrows) return;
    double s = 0.;
    for (int k = indptr[row]; k < indptr[row+1]; k++) s = __builtin_fma(data[k], x[indices[k]], s);
    y[row] = beta != 0. ? __builtin_fma(alpha, s, beta*y[row]) : alpha*s;
}

// x += d * r  (update(y, prec * residual), smoothers_{SCALAR}.pxi:105-108); d == nullptr: x += r
__global__ void __launch_bounds__(PNL_NTHREADS)
k_vec_update(int n, const double *__restrict__ d, const double *__restrict__ r, double *x) {
    const int i = blockIdx.x*PNL_NTHREADS+threadIdx.x;
    if (i < n) x[i] = d ? __builtin_fma(d[i], r[i], x[i]) : x[i]+r[i];
}

// y = a x + b y
__global__ void __launch_bounds__(PNL_NTHREADS)
k_vec_axpby(int n, double a, const double *__restrict__ x, double b, double *y) {
    const int i = blockIdx.x*PNL_NTHREADS+threadIdx.x;
    if (i < n) y[i] = b != 0. ? __builtin_fma(a, x[i], b*y[i]) : a*x[i];
}

__global__ void __launch_bounds__(PNL_NTHREADS)
k_vec_scale_inv(int n, double omega, const double *__restrict__ d, double *out) {
    const int i = blockIdx.x*PNL_NTHREADS+threadIdx.x;
    if (i < n) out[i] = omega/d[i];
}

// out[0] += x . y  (out zeroed before the launch)
__global__ void __launch_bounds__(PNL_NTHREADS)
k_vec_dot(int n, const double *__restrict__ x, const double *__restrict__ y, double *out) {
    double s = 0.;
    for (int i = blockIdx.x*PNL_NTHREADS+threadIdx.x; i < n; i += gridDim.x*PNL_NTHREADS) s = __builtin_fma(x[i], y[i], s);
    s = wave_sum(s);
    if ((threadIdx.x & 63) == 0 && s != 0.) atomic_add_f64(out, s);
}

inline unsigned blocks_for(long long n) { return (unsigned)std::max<long long>(1, (n+PNL_NTHREADS-1)/PNL_NTHREADS); }

int gemv(pnl_context *ctx, const double *A, int64_t ldA, int nrows, int ncols, const double *x, double alpha, double beta, const double *b,
         double *y) {
    if (nrows <= 0) return PNL_OK;
    hipLaunchKernelGGL(k_gemv_axpby, dim3(blocks_for((long long)nrows*64)), dim3(PNL_NTHREADS), 0, ctx->stream, A, (long long)ldA, nrows, ncols,
                       x, alpha, beta, b, y);
    HIPCHK(ctx, hipGetLastError());
    return PNL_OK;
}

int csr(pnl_context *ctx, int nrows, const int *indptr, const int *indices, const double *data, const double *x, double alpha, double beta,
        double *y) {
    if (nrows <= 0) return PNL_OK;
    hipLaunchKernelGGL(k_csr_axpby, dim3(blocks_for(nrows)), dim3(PNL_NTHREADS), 0, ctx->stream, nrows, indptr, indices, data, x, alpha, beta, y);
    HIPCHK(ctx, hipGetLastError());
    return PNL_OK;
}

int update(pnl_context *ctx, int n, const double *d, const double *r, double *x) {
    hipLaunchKernelGGL(k_vec_update, dim3(blocks_for(n)), dim3(PNL_NTHREADS), 0, ctx->stream, n, d, r, x);
    HIPCHK(ctx, hipGetLastError());
    return PNL_OK;
}

int axpby(pnl_context *ctx, int n, double a, const double *x, double b, double *y) {
    hipLaunchKernelGGL(k_vec_axpby, dim3(blocks_for(n)), dim3(PNL_NTHREADS), 0, ctx->stream, n, a, x, b, y);
    HIPCHK(ctx, hipGetLastError());
    return PNL_OK;
}

// y = alpha A_l x + beta b with the operator of level l: dense GEMV, or (kind 1, finest level) near-field CSR product + the far
// field of the H2 operator set up in the context (b may be y; x must not be y)
int apply_level(pnl_mg *mg, int l, const double *x, double alpha, double beta, const double *b, double *y) {
    pnl_context *ctx = mg->ctx;
    const pnl_mg_level_desc &L = mg->lv[l];
    if (L.kind == 0) return gemv(ctx, L.A_dev, L.ldA, L.n, L.n, x, alpha, beta, b, y);
    if (L.kind == 2) return pnl_launch_gemv_symmetric(ctx, L.A_dev, L.ldA, L.n, x, alpha, beta, b, y);
    double *t = (double*)mg->h2tmp.p;
    int rc;
    if ((rc = csr(ctx, L.n, L.near_indptr_dev, L.near_indices_dev, L.near_data_dev, x, 1., 0., t))) return rc;
    if ((rc = pnl_h2_matvec(ctx, x, t))) return rc;
    if (beta != 0. && b != y) HIPCHK(ctx, hipMemcpyAsync(y, b, sizeof(double)*L.n, hipMemcpyDeviceToDevice, ctx->stream));
    return axpby(ctx, L.n, alpha, t, beta, y);
}

// synchronous: the scalar is needed on the host (step lengths and the convergence test of the CG / multigrid iterations)
int dot(pnl_mg *mg, int n, const double *x, const double *y, double *out) {
    pnl_context *ctx = mg->ctx;
    double *d = (double*)mg->scal.p;
    HIPCHK(ctx, hipMemsetAsync(d, 0, sizeof(double), ctx->stream));
    hipLaunchKernelGGL(k_vec_dot, dim3(std::min(blocks_for(n), 1024u)), dim3(PNL_NTHREADS), 0, ctx->stream, n, x, y, d);
    HIPCHK(ctx, hipGetLastError());
    HIPCHK(ctx, hipMemcpyAsync(out, d, sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    return PNL_OK;
}

// separableSmoother.eval with the Jacobi preconditioner (smoothers_{SCALAR}.pxi:88-108, 118-131): steps sweeps of
// x += (omega / D) (b - A x); simple: x is zero, the first residual is b
int smooth(pnl_mg *mg, int l, const double *b, double *x, int steps, bool simple) {
    pnl_context *ctx = mg->ctx;
    const pnl_mg_level_desc &L = mg->lv[l];
    double *res = (double*)mg->temp[l].p;
    const double *invD = (const double*)mg->invD[l].p;
    int rc;
    for (int k = 0; k < steps; k++) {
        if (simple) {
            if ((rc = update(ctx, L.n, invD, b, x))) return rc;
            simple = false;
            continue;
        }
        if ((rc = apply_level(mg, l, x, -1., 1., b, res))) return rc;
        if ((rc = update(ctx, L.n, invD, res, x))) return rc;
    }
    return PNL_OK;
}

// multigrid.solveOnLevel (multigrid_{SCALAR}.pxi:237-292), V cycle
int solve_on_level(pnl_mg *mg, int l, const double *b, double *x, bool simple) {
    pnl_context *ctx = mg->ctx;
    const pnl_mg_level_desc &L = mg->lv[l];
    int rc;
    if (l == 0) return gemv(ctx, mg->coarse_inv, L.n, L.n, L.n, b, 1., 0., nullptr, x);       // coarse solver: x = A_0^{-1} b
    const pnl_mg_level_desc &C = mg->lv[l-1];
    double *res = (double*)mg->temp[l].p, *defect = (double*)mg->rhs[l-1].p, *solcg = (double*)mg->sol[l-1].p;
    if ((rc = smooth(mg, l, b, x, mg->pre, simple))) return rc;
    // residual, restricted to the coarser level
    if (simple && mg->pre == 0) HIPCHK(ctx, hipMemcpyAsync(res, b, sizeof(double)*L.n, hipMemcpyDeviceToDevice, ctx->stream));
    else if ((rc = apply_level(mg, l, x, -1., 1., b, res))) return rc;
    if ((rc = csr(ctx, C.n, L.R_indptr_dev, L.R_indices_dev, L.R_data_dev, res, 1., 0., defect))) return rc;
    HIPCHK(ctx, hipMemsetAsync(solcg, 0, sizeof(double)*C.n, ctx->stream));
    if ((rc = solve_on_level(mg, l-1, defect, solcg, true))) return rc;
    // correction: x += P solcg
    if ((rc = csr(ctx, L.n, L.P_indptr_dev, L.P_indices_dev, L.P_data_dev, solcg, 1., 1., x))) return rc;
    return smooth(mg, l, b, x, mg->post, false);
}

}  // namespace

extern "C" {

int pnl_gemv_axpby(pnl_context *ctx, const double *A_dev, int64_t ldA, int nrows, int ncols, const double *x_dev, double alpha, double beta,
                   const double *b_dev, double *y_dev) {
    if (!ctx || !A_dev || !x_dev || !y_dev || nrows < 0 || ncols < 0 || ldA < ncols || (beta != 0. && !b_dev))
        return ctx ? fail(ctx, PNL_ERR_INVALID, "pnl_gemv_axpby: bad arguments") : PNL_ERR_INVALID;
    return gemv(ctx, A_dev, ldA, nrows, ncols, x_dev, alpha, beta, b_dev, y_dev);
}

int pnl_csr_matvec(pnl_context *ctx, int nrows, const int32_t *indptr_dev, const int32_t *indices_dev, const double *data_dev,
                   const double *x_dev, double alpha, double beta, double *y_dev) {
    if (!ctx || nrows < 0 || !indptr_dev || !x_dev || !y_dev) return ctx ? fail(ctx, PNL_ERR_INVALID, "pnl_csr_matvec: bad arguments") : PNL_ERR_INVALID;
    return csr(ctx, nrows, indptr_dev, indices_dev, data_dev, x_dev, alpha, beta, y_dev);
}

int pnl_mg_create(pnl_context *ctx, int nlevels, const pnl_mg_level_desc *levels, const double *coarse_inverse_dev, double omega, int presmooth,
                  int postsmooth, pnl_mg **out) {
    if (!ctx || !out || nlevels < 1 || !levels || !coarse_inverse_dev || presmooth < 0 || postsmooth < 0 || !(omega > 0.))
        return ctx ? fail(ctx, PNL_ERR_INVALID, "pnl_mg_create: bad arguments") : PNL_ERR_INVALID;
    for (int l = 0; l < nlevels; l++) {
        const pnl_mg_level_desc &L = levels[l];
        if (L.kind != 0 && L.kind != 2 && !(L.kind == 1 && l == nlevels-1 && l > 0 && L.near_indptr_dev && L.near_indices_dev && L.near_data_dev && L.diag_dev))
            return fail(ctx, PNL_ERR_INVALID, "pnl_mg_create: level %d: an H2 operator (kind 1) is taken on the finest level only, with its near field", l);
        if (L.n <= 0 || (l > 0 && L.kind != 1 && (!L.A_dev || !L.diag_dev || L.ldA < L.n)))
            return fail(ctx, PNL_ERR_INVALID, "pnl_mg_create: level %d needs n > 0, the operator and its diagonal", l);
        if (l > 0 && (!L.R_indptr_dev || !L.R_indices_dev || !L.R_data_dev || !L.P_indptr_dev || !L.P_indices_dev || !L.P_data_dev))
            return fail(ctx, PNL_ERR_INVALID, "pnl_mg_create: level %d needs restriction and prolongation", l);
        if (l > 0 && L.n <= levels[l-1].n) return fail(ctx, PNL_ERR_INVALID, "pnl_mg_create: levels must be ordered coarse to fine");
    }
    pnl_mg *mg = new pnl_mg;
    mg->ctx = ctx; mg->nlevels = nlevels; mg->lv.assign(levels, levels+nlevels); mg->coarse_inv = coarse_inverse_dev;
    mg->omega = omega; mg->pre = presmooth; mg->post = postsmooth;
    mg->invD.resize(nlevels); mg->rhs.resize(nlevels); mg->sol.resize(nlevels); mg->temp.resize(nlevels);
    int rc = PNL_OK;
    for (int l = 0; l < nlevels && !rc; l++) {
        const size_t bytes = sizeof(double)*(size_t)levels[l].n;
        if ((rc = ensure(ctx, mg->rhs[l], bytes)) || (rc = ensure(ctx, mg->sol[l], bytes)) || (rc = ensure(ctx, mg->temp[l], bytes))) break;
        if (l > 0) {
            if ((rc = ensure(ctx, mg->invD[l], bytes))) break;
            hipLaunchKernelGGL(k_vec_scale_inv, dim3(blocks_for(levels[l].n)), dim3(PNL_NTHREADS), 0, ctx->stream, levels[l].n, omega,
                               levels[l].diag_dev, (double*)mg->invD[l].p);
        }
    }
    const size_t nb = sizeof(double)*(size_t)levels[nlevels-1].n;
    if (!rc) rc = ensure(ctx, mg->r, nb);
    if (!rc) rc = ensure(ctx, mg->p, nb);
    if (!rc) rc = ensure(ctx, mg->Ap, nb);
    if (!rc) rc = ensure(ctx, mg->z, nb);
    if (!rc) rc = ensure(ctx, mg->work, nb);
    if (!rc) rc = ensure(ctx, mg->h2tmp, nb);
    if (!rc) rc = ensure(ctx, mg->scal, 4*sizeof(double));
    if (!rc && hipGetLastError() != hipSuccess) rc = fail(ctx, PNL_ERR_HIP, "pnl_mg_create: launch failed");
    if (rc) { delete mg; return rc; }
    *out = mg;
    return PNL_OK;
}

int pnl_mg_destroy(pnl_mg *mg) {
    if (!mg) return PNL_ERR_INVALID;
    (void)hipStreamSynchronize(mg->ctx->stream);
    delete mg;
    return PNL_OK;
}

int pnl_mg_cycle(pnl_mg *mg, const double *b_dev, double *x_dev, int x_is_zero) {
    if (!mg || !b_dev || !x_dev) return PNL_ERR_INVALID;
    return solve_on_level(mg, mg->nlevels-1, b_dev, x_dev, x_is_zero != 0);
}

int pnl_mg_solve(pnl_mg *mg, const double *b_dev, double *x_dev, double tol, int maxiter, int x_is_zero, int *iters, double *residuals,
                 int residuals_cap) {
    if (!mg || !b_dev || !x_dev || maxiter < 0) return PNL_ERR_INVALID;
    pnl_context *ctx = mg->ctx;
    const int top = mg->nlevels-1;
    const pnl_mg_level_desc &L = mg->lv[top];
    double *res = (double*)mg->r.p;
    int rc, it = 0;
    bool simple = x_is_zero != 0;
    double n2 = 0.;
    auto residual_norm = [&](bool simple_res) -> int {
        int r2;
        if (simple_res) HIPCHK(ctx, hipMemcpyAsync(res, b_dev, sizeof(double)*L.n, hipMemcpyDeviceToDevice, ctx->stream));
        else if ((r2 = apply_level(mg, top, x_dev, -1., 1., b_dev, res))) return r2;
        return dot(mg, L.n, res, res, &n2);
    };
    if (top == 0) {
        // one level: the coarse solver is the solver
        // residual history like the multi-level case: ||b - A x0|| before, ||b - A x|| after the one (direct) solve
        int nres1 = 0;
        if (residuals && residuals_cap > 0) {
            if ((rc = residual_norm(simple))) return rc;
            residuals[nres1++] = std::sqrt(n2);
        }
        if ((rc = gemv(ctx, mg->coarse_inv, L.n, L.n, L.n, b_dev, 1., 0., nullptr, x_dev))) return rc;
        if (residuals && nres1 < residuals_cap && L.A_dev) {
            if ((rc = residual_norm(false))) return rc;
            residuals[nres1++] = std::sqrt(n2);
        }
        if (iters) *iters = 1;
        return PNL_OK;
    }
    if ((rc = residual_norm(simple))) return rc;
    int nres = 0;
    if (residuals && nres < residuals_cap) residuals[nres++] = std::sqrt(n2);
    while (std::sqrt(n2) > tol && it < maxiter) {
        it++;
        if ((rc = solve_on_level(mg, top, b_dev, x_dev, simple))) return rc;
        simple = false;
        if ((rc = residual_norm(false))) return rc;
        if (residuals && nres < residuals_cap) residuals[nres++] = std::sqrt(n2);
    }
    if (iters) *iters = it;
    return PNL_OK;
}

// cg_solver.solve (solvers.pyx:363-444) with one V cycle from a zero guess as preconditioner (multigridPreconditioner)
int pnl_mg_cg(pnl_mg *mg, const double *A_dev, int64_t ldA, const double *b_dev, double *x_dev, double tol, int maxiter, int x_is_zero,
              int *iters, double *residuals, int residuals_cap) {
    if (!mg || !b_dev || !x_dev || maxiter < 0) return PNL_ERR_INVALID;
    pnl_context *ctx = mg->ctx;
    const int top = mg->nlevels-1;
    const pnl_mg_level_desc &L = mg->lv[top];
    const bool h2top = !A_dev && L.kind == 1;
    const bool own = !A_dev || (A_dev == L.A_dev && ldA == L.ldA);      // the finest level's own operator (it may be symmetric: kind 2)
    if (!A_dev) { A_dev = L.A_dev; ldA = L.ldA; }
    if (!h2top && (!A_dev || ldA < L.n)) return fail(ctx, PNL_ERR_INVALID, "pnl_mg_cg: no operator");
    const int n = L.n;
    auto applyA = [&](const double *x, double alpha, double beta, const double *b, double *y) -> int {
        return (h2top || (own && L.kind == 2)) ? apply_level(mg, top, x, alpha, beta, b, y) : gemv(ctx, A_dev, ldA, n, n, x, alpha, beta, b, y);
    };
    double *r = (double*)mg->r.p, *p = (double*)mg->p.p, *Ap = (double*)mg->Ap.p, *z = (double*)mg->z.p;
    int rc;
    auto precond = [&](const double *in, double *out) -> int {
        HIPCHK(ctx, hipMemsetAsync(out, 0, sizeof(double)*n, ctx->stream));
        return solve_on_level(mg, top, in, out, true);
    };
    if (x_is_zero) HIPCHK(ctx, hipMemcpyAsync(r, b_dev, sizeof(double)*n, hipMemcpyDeviceToDevice, ctx->stream));
    else if ((rc = applyA(x_dev, -1., 1., b_dev, r))) return rc;
    if ((rc = precond(r, p))) return rc;
    double betaOld = 0., beta = 0., pAp = 0.;
    if ((rc = dot(mg, n, r, p, &betaOld))) return rc;
    double conv = std::sqrt(std::fabs(betaOld));
    int nres = 0, its = 0;
    if (residuals && nres < residuals_cap) residuals[nres++] = conv;
    if (conv > tol) {
        int k = 0;
        its = maxiter;
        for (int i = 0; i < maxiter; i++) {
            if ((rc = applyA(p, 1., 0., nullptr, Ap))) return rc;
            if ((rc = dot(mg, n, p, Ap, &pAp))) return rc;
            const double alpha = betaOld/pAp;
            if ((rc = axpby(ctx, n, alpha, p, 1., x_dev))) return rc;
            if ((rc = axpby(ctx, n, -alpha, Ap, 1., r))) return rc;
            if (k == 50) {
                // recalculate the residual to avoid rounding errors (solvers.pyx:412-415)
                if ((rc = applyA(x_dev, -1., 1., b_dev, r))) return rc;
                k = 0;
            }
            if ((rc = precond(r, z))) return rc;
            if ((rc = dot(mg, n, r, z, &beta))) return rc;
            conv = std::sqrt(std::fabs(beta));
            if (residuals && nres < residuals_cap) residuals[nres++] = conv;
            its = i;
            if (conv <= tol) break;
            if ((rc = axpby(ctx, n, 1., z, beta/betaOld, p))) return rc;
            betaOld = beta;
            k++;
            if (i == maxiter-1) its = maxiter;
        }
    }
    if (iters) *iters = its;
    mg->last_conv = conv;
    return PNL_OK;
}

// One step of the theta method for M u_t + S u = g (CrankNicolson.step, timestepping.py:93-112):
//   (M / dt + theta S) u_new = (M / dt) u - (1 - theta) S u + forcing,    forcing = (1 - theta) g(t) + theta g(t + dt),
// solved by multigrid-preconditioned CG on the hierarchy mg of  M / dt + theta S  with u as initial guess.
int pnl_theta_step(pnl_mg *mg, const double *S_dev, int64_t ldS, const int32_t *M_indptr_dev, const int32_t *M_indices_dev,
                   const double *M_data_dev, double dt, double theta, const double *forcing_dev, double *u_dev, double tol, int maxiter,
                   int *iters, double *residual) {
    if (!mg) return PNL_ERR_INVALID;
    pnl_context *ctx = mg->ctx;
    if (!S_dev || !M_indptr_dev || !M_indices_dev || !M_data_dev || !u_dev || !(dt > 0.) || theta < 0. || theta > 1.)
        return fail(ctx, PNL_ERR_INVALID, "pnl_theta_step: null operator / vector, dt <= 0 or theta outside [0, 1]");
    const int n = mg->lv[mg->nlevels-1].n;
    if (ldS < n) return fail(ctx, PNL_ERR_INVALID, "pnl_theta_step: ldS < n");
    double *rhs = (double*)mg->work.p;
    int rc;
    if (forcing_dev) HIPCHK(ctx, hipMemcpyAsync(rhs, forcing_dev, sizeof(double)*n, hipMemcpyDeviceToDevice, ctx->stream));
    else HIPCHK(ctx, hipMemsetAsync(rhs, 0, sizeof(double)*n, ctx->stream));
    if ((rc = csr(ctx, n, M_indptr_dev, M_indices_dev, M_data_dev, u_dev, 1./dt, 1., rhs))) return rc;
    if (theta < 1. && (rc = gemv(ctx, S_dev, ldS, n, n, u_dev, -(1.-theta), 1., rhs, rhs))) return rc;
    int its = 0;
    if ((rc = pnl_mg_cg(mg, nullptr, 0, rhs, u_dev, tol, maxiter, 0, &its, nullptr, 0))) return rc;
    if (iters) *iters = its;
    if (residual) *residual = mg->last_conv;
    return PNL_OK;
}

}  // extern "C"
