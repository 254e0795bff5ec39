/* CPU oracle for the nonlocal element-pair assembly path -- TEST INFRASTRUCTURE ONLY
 * (see nl_oracle.h for scope and parity status).
 *
 * Each function cites the reference lines it restates; paths are relative to
 * /root/reference/nl/PyNucleus_nl/ with
 *   NO  = nonlocalOperator_{SCALAR}.pxi     NA  = nonlocalAssembly_{SCALAR}.pxi
 *   FL2 = fractionalLaplacian2D.pyx         FL1 = fractionalLaplacian1D.pyx
 *   KC  = kernelsCy.pyx                     Q   = ../../fem/PyNucleus_fem/quadrature.pyx
 */
#define _POSIX_C_SOURCE 199309L
#define _DEFAULT_SOURCE
#include "nl_oracle.h"
#include <math.h>
#include <string.h>
#include <stdlib.h>
#include <time.h>

#define MAXV 3     /* vertices per cell (dim+1 <= 3) */
#define MAXDPE 6
#define MAXE (2*MAXDPE*(2*MAXDPE+1)/2)

static double now_s(void) {
    struct timespec ts;
    clock_gettime(CLOCK_MONOTONIC, &ts);
    return ts.tv_sec + 1e-9*ts.tv_nsec;
}

/* KC:159-183 (fractional), KC:273-294 (indicator), KC:321-360 (peridynamic), finite variants KC:75-114:
 * the interaction test comes first. */
static inline double kernel_eval(const nlo_kernel *K, double d2) {
    if (!(d2 <= K->horizon2)) return 0.;
    switch (K->ktype) {
    case 0: return K->scale*pow(d2, K->exponent);
    case 1: return K->scale;
    case 2: return K->scale/sqrt(d2);
    case 3: return K->scale*exp(K->exponent*d2);          /* gaussianKernel1D / 2D, KC:388-416: C exp(-d2 invD), exponent = -invD */
    case 4: return K->scale*exp(K->exponent*sqrt(d2));     /* exponentialKernel, KC:448-462: C exp(-a |x-y|), exponent = -a */
    /* Gauss-theorem twins on the full space; gammainc(a, x) = Gamma(a) gammaincc(a, x) (KC:39-40): Gamma(1/2, z) = sqrt(pi) erfc(sqrt z),
     * Gamma(1, z) = exp(-z).  5: gaussianKernel1Dboundary KC:418-430, 7: gaussianKernel2Dboundary KC:433-445 (oracle.py picks by
     * dimension), 6: exponentialKernelBoundary KC:463-477 */
    case 5: return K->scale*sqrt(1./(d2*-K->exponent))*(sqrt(M_PI)*erfc(sqrt(d2*-K->exponent)))*sqrt(d2);
    case 7: return K->scale*(1./(d2*-K->exponent))*exp(K->exponent*d2)*sqrt(d2);
    default: return 2.0*K->scale*exp(K->exponent*sqrt(d2))/(-K->exponent);
    }
}

/* nonlocalOperator.pyx:64-79 PermutationIndexer.rank (Lehmer code) */
static int perm_rank(const int *perm, int n) {
    static const int fact[4] = {1, 1, 2, 6};
    int index = 0;
    for (int i = 0; i < n; i++) {
        int smaller = 0;
        for (int j = 0; j < i; j++) smaller += (perm[j] < perm[i]);
        index += (perm[i]-smaller)*fact[n-1-i];
    }
    return index;
}

/* Kernel.evalParams for a piecewise-constant variable order (kernelsCy.pyx:1852-1867 with the order functions of
 * fractionalOrders.pyx:285-336, 826-882): the parameters of the pair are those of the class of the two labels */
static const nlo_problem *class_of_cells(const nlo_problem *P, int c1, int c2) {
    if (!P->nclasses) return P;
    return &P->classes[P->cls_of[P->cell_labels[c1]*P->num_labels+P->cell_labels[c2]]];
}
static const nlo_problem *class_of_cell_facet(const nlo_problem *P, int c1, int b) {
    if (!P->nclasses) return P;
    return &P->classes[P->cls_of[P->cell_labels[c1]*P->num_labels+P->facet_labels[b]]];
}

static void simplex_of(const nlo_problem *P, int c, double s[MAXV][2], double center[2]) {
    int nV = P->dim+1;
    center[0] = center[1] = 0.;
    for (int m = 0; m < nV; m++) {
        int v = P->cells[c*nV+m];
        for (int l = 0; l < P->dim; l++) {
            s[m][l] = P->vertices[v*P->dim+l];
            center[l] += s[m][l];
        }
    }
    double fac = 1./nV;                         /* NO:116-126 */
    for (int l = 0; l < P->dim; l++) center[l] *= fac;
    /* linearTransformInteraction (interactionDomains.pyx:1417-1470, 1502-1520): relative position, sub-simplices and the kernel's
     * distance are those of the simplices transformed by T (everything they enter is a difference x - y); the centre -- the
     * order formula -- is the mesh's */
    if (P->has_xform && P->dim == 2)
        for (int m = 0; m < nV; m++) {
            const double x = s[m][0], y = s[m][1];
            s[m][0] = P->xform[0]*x+P->xform[1]*y;
            s[m][1] = P->xform[2]*x+P->xform[3]*y;
        }
}

#define NLO_INTERACT 0
#define NLO_REMOTE 1
#define NLO_CUT 2
static int rel_position(const nlo_problem *P, double s1[MAXV][2], int n1, double s2[MAXV][2], int n2);

/* FL2:622-642, FL1:234-253, boundary FL2:1226-1243, FL1:646-660 */
static int quad_order(const nlo_order_formula *F, double H0, double h1, double h2, double d) {
    double logdh1 = log(d/h1), logdh2 = log(d/h2);
    double L1 = fabs(log(h1/H0)), L2 = fabs(log(h2/H0));
    double Lm = fmax(L1, L2);
    double n1 = logdh1, n2 = logdh2;
    if (F->clip_num) { n1 = fmax(logdh1, 0.); n2 = fmax(logdh2, 0.); }
    double p1 = ceil((F->c0 + F->a*L2 + F->b*Lm - F->e*n2)/(fmax(logdh1, 0.) + F->den0));
    double p2 = ceil((F->c0 + F->a*L1 + F->b*Lm - F->e*n1)/(fmax(logdh2, 0.) + F->den0));
    int q1 = (int)fmax(p1, 2.), q2 = (int)fmax(p2, 2.);
    return q1 > q2 ? q1 : q2;
}

/* NO:280-378 getProtoPanelType + NO:493-540 getPanelType (constant-order kernel, infinite or finite horizon without cut handling) */
int nlo_panel(const nlo_problem *P, int c1, int c2, int *perm1, int *perm2, int *perm) {
    const int nV = P->dim+1, dpe = P->dpe;
    if (c1 > c2) return NLO_IGNORED;
    for (int k = 0; k < nV; k++) { perm1[k] = k; perm2[k] = k; }
    for (int k = 0; k < dpe; k++) perm[k] = k;
    if (c1 == c2) return -nV;                   /* IDENTICAL */
    int mask1 = 0, mask2 = 0, common = 0;
    for (int a = 0; a < nV; a++) {
        int v1 = P->cells[c1*nV+a];
        for (int b = 0; b < nV; b++) {
            if (mask2 & (1 << b)) continue;
            if (v1 == P->cells[c2*nV+b]) {
                perm1[common] = a; perm2[common] = b;
                mask1 += (1 << a); mask2 += (1 << b);
                common++;
                break;
            }
        }
    }
    if (common == 0) {
        double s1[MAXV][2], s2[MAXV][2], ce1[2], ce2[2];
        simplex_of(P, c1, s1, ce1);
        simplex_of(P, c2, s2, ce2);
        if (rel_position(P, s1, nV, s2, nV) == NLO_REMOTE) return NLO_IGNORED;     /* NO:515-517 */
        double d2 = 0.;
        for (int j = 0; j < P->dim; j++) d2 += (ce1[j]-ce2[j])*(ce1[j]-ce2[j]);
        return quad_order(&class_of_cells(P, c1, c2)->qo, P->H0, P->h[c1], P->h[c2], sqrt(d2));
    }
    int i = 0;
    for (int k = common; k < nV; k++) { while (mask1 & (1 << i)) i++; perm1[k] = i; mask1 += (1 << i); }
    i = 0;
    for (int k = common; k < nV; k++) { while (mask2 & (1 << i)) i++; perm2[k] = i; mask2 += (1 << i); }
    const int *t1 = P->dof_perm_table + perm_rank(perm1, nV)*dpe;
    const int *t2 = P->dof_perm_table + perm_rank(perm2, nV)*dpe;
    const int dpv = P->dofs_per_vertex, dped = P->dofs_per_edge;
    for (int k = 0; k < dpe; k++) perm[k] = t1[k];
    if (common == 1) {
        for (int k = dpv; k < dpe; k++) perm[dpe+k-dpv] = dpe+t2[k];
    } else if (common == 2) {
        for (int k = 2*dpv; k < nV*dpv; k++) perm[dpe+k-2*dpv] = dpe+t2[k];
        for (int k = nV*dpv+dped; k < dpe; k++) perm[dpe+k-2*dpv-dped] = dpe+t2[k];
    }
    return -common;
}


/* ---- finite horizon (interactionDomains.pyx) ------------------------------------------------------------------
 * RELATIVE_POSITION of two simplices for the l2 ball: ball2_retriangulation.getRelativePosition :875-898 and
 * ball2_barycenter.getRelativePosition :990-1013 (the same vertex-distance test). */
static int rel_position(const nlo_problem *P, double s1[MAXV][2], int n1, double s2[MAXV][2], int n2) {
    const double h2 = P->kernel.horizon2;
    if (isinf(h2)) return NLO_INTERACT;         /* fullSpace :834 */
    double dmin2 = INFINITY, dmax2 = 0.;
    for (int i = 0; i < n1; i++)
        for (int k = 0; k < n2; k++) {
            double d2 = 0.;
            for (int j = 0; j < P->dim; j++) d2 += (s1[i][j]-s2[k][j])*(s1[i][j]-s2[k][j]);
            if (d2 < dmin2) dmin2 = d2;
            if (d2 > dmax2) dmax2 = d2;
        }
    if (dmin2 >= h2) return NLO_REMOTE;
    if (dmax2 <= h2) return NLO_INTERACT;
    return NLO_CUT;
}

typedef struct { int n; double A[3][3][3], b[3][3], vol[3]; } subs_t;

static int is_inside(const nlo_problem *P, const double *x, const double *y) {   /* :900-909 */
    double d2 = 0.;
    for (int j = 0; j < P->dim; j++) d2 += (x[j]-y[j])*(x[j]-y[j]);
    return d2 <= P->kernel.horizon2;
}

/* ball2_retriangulation.findIntersections :911-938 */
static int find_intersections(const nlo_problem *P, const double *x, double simplex[MAXV][2], int start, int end, double *out) {
    double nn = 0., p = 0., q = 0.;
    for (int k = 0; k < P->dim; k++) {
        double A = simplex[end][k]-simplex[start][k], B = simplex[start][k]-x[k];
        nn += A*A; p += A*B; q += B*B;
    }
    nn = 1./nn;
    p *= 2.*nn;
    q = (q-P->kernel.horizon2)*nn;
    double A = -p*0.5, B = sqrt(A*A-q), c;
    int num = 0;
    c = A-B;
    if (c >= 0 && c <= 1) out[num++] = c;
    c = A+B;
    if (c >= 0 && c <= 1) out[num++] = c;
    return num;
}

static void subs_identity(subs_t *S, int nv) {
    memset(S, 0, sizeof(*S));
    for (int k = 0; k < nv; k++) S->A[0][k][k] = 1.;
    S->vol[0] = 1.;
    S->n = 1;
}

/* startLoopSubSimplices_Simplex: retriangulationDomain :406-567 (2D; the 1D interval branch :425-446 with
 * nextSubSimplex_Simplex :71-87), barycenterDomain :358-374 */
static void subs_simplex(const nlo_problem *P, double s1[MAXV][2], double s2[MAXV][2], subs_t *S) {
    const int dim = P->dim, nV = dim+1;
    memset(S, 0, sizeof(*S));
    if (P->kernel.interaction == 2) {
        double bary[2] = {0., 0.};
        for (int v = 0; v < nV; v++) for (int j = 0; j < dim; j++) bary[j] += s2[v][j];
        for (int j = 0; j < dim; j++) bary[j] /= nV;
        for (int v = 0; v < nV; v++)
            if (is_inside(P, s1[v], bary)) { subs_identity(S, nV); return; }
        return;
    }
    if (dim == 1) {
        const double horizon = sqrt(P->kernel.horizon2);
        const int lr = s1[0][0] < s2[0][0];
        const double vol1 = fabs(s1[0][0]-s1[1][0]), inv = 1./vol1;
        double iv[4];
        int it, itEnd;
        iv[0] = s1[0][0]*inv; iv[3] = s1[1][0]*inv;
        if (lr) {
            iv[1] = fmax(s1[0][0], s2[0][0]-horizon)*inv;
            iv[2] = fmin(s1[1][0], s2[1][0]-horizon)*inv;
            it = 1; itEnd = 3;
        } else {
            iv[1] = fmax(s1[0][0], s2[0][0]+horizon)*inv;
            iv[2] = fmin(s1[1][0], s2[1][0]+horizon)*inv;
            it = 0; itEnd = 2;
        }
        for (; it < itEnd; it++) {
            const double l = iv[it], r = iv[it+1];
            if (r-l <= 0) continue;
            const int m = S->n++;
            S->A[m][0][0] = r-l; S->A[m][1][1] = r-l;
            S->b[m][0] = iv[3]-r; S->b[m][1] = l-iv[0];
            S->vol[m] = r-l;
        }
        return;
    }
    int insideIJ[3][3], insideI[3], numInside = 0;
    for (int i = 0; i < 3; i++) {
        int any = 0;
        for (int k = 0; k < 3; k++) { insideIJ[i][k] = is_inside(P, s1[i], s2[k]); any |= insideIJ[i][k]; }
        insideI[i] = any;
        numInside += any;
    }
    double isec[2];
    if (numInside == 0) return;                 /* the reference raises NotImplementedError; CUT implies numInside >= 1 */
    if (numInside == 1) {
        int inside = 0;
        while (!insideI[inside]) inside++;
        const int o1 = (inside+1)%3, o2 = (inside+2)%3;
        double c1 = 0., c2 = 0.;
        for (int j = 0; j < 3; j++)
            if (insideIJ[inside][j]) {
                find_intersections(P, s2[j], s1, inside, o1, isec);
                c1 = fmax(c1, isec[0]);
                find_intersections(P, s2[j], s1, inside, o2, isec);
                c2 = fmax(c2, isec[0]);
            }
        if (c1*c2 > 0) {
            S->A[0][inside][inside] = c1+c2;
            S->A[0][inside][o1] = c2;
            S->A[0][inside][o2] = c1;
            S->A[0][o1][o1] = c1;
            S->A[0][o2][o2] = c2;
            S->b[0][inside] = 1-c1-c2;
            S->vol[0] = c1*c2;
            S->n = 1;
        }
        return;
    }
    if (numInside == 2) {
        int outside = 0;
        while (insideI[outside]) outside++;
        const int i1 = (outside+1)%3, i2 = (outside+2)%3;
        double c1 = 1., c2 = 1.;
        for (int j = 0; j < 3; j++) {
            if (insideIJ[i1][j]) { find_intersections(P, s2[j], s1, outside, i1, isec); c1 = fmin(c1, isec[0]); }
            if (insideIJ[i2][j]) { find_intersections(P, s2[j], s1, outside, i2, isec); c2 = fmin(c2, isec[0]); }
        }
        /* :509-516, literally (simplex2 coordinates, d2 without the square) */
        double d1 = 0., d2 = 0.;
        for (int k = 0; k < 2; k++) {
            const double t = s2[outside][k]+c1*(s2[i1][k]-s2[outside][k])-s2[i2][k];
            d1 += t*t;
            d2 += s2[outside][k]+c2*(s2[i2][k]-s2[outside][k])-s2[i1][k];
        }
        if (d1 < d2) {
            if (1-c1 > 0) {
                const int m = S->n++;
                S->A[m][outside][outside] = 1-c1;
                S->A[m][i1][i1] = 1-c1;
                S->A[m][i1][i2] = -c1;
                S->A[m][i2][i2] = 1.;
                S->b[m][i1] = c1;
                S->vol[m] = 1-c1;
            }
            if (c1*(1-c2) > 0.) {
                const int m = S->n++;
                S->A[m][outside][outside] = 1-c2;
                S->A[m][i2][i2] = 1;
                S->A[m][i2][outside] = c2;
                S->A[m][outside][i1] = 1-c1;
                S->A[m][i1][i1] = c1;
                S->vol[m] = c1*(1-c2);
            }
        } else {
            if (1-c2 > 0) {
                const int m = S->n++;
                S->A[m][outside][outside] = 1-c2;
                S->A[m][i2][i2] = 1-c2;
                S->A[m][i2][i1] = -c2;
                S->A[m][i1][i1] = 1.;
                S->b[m][i2] = c2;
                S->vol[m] = 1-c2;
            }
            if (c2*(1-c1) > 0.) {
                const int m = S->n++;
                S->A[m][outside][outside] = 1-c1;
                S->A[m][i1][i1] = 1;
                S->A[m][i1][outside] = c1;
                S->A[m][outside][i2] = 1-c2;
                S->A[m][i2][i2] = c2;
                S->vol[m] = c2*(1-c1);
            }
        }
        return;
    }
    subs_identity(S, 3);
}

/* startLoopSubSimplices_Node: retriangulationDomain :570-822 (no special points for the l2 ball: specialOffsets is empty,
 * :43), barycenterDomain :376-392 */
static void subs_node(const nlo_problem *P, const double *x, double s2[MAXV][2], subs_t *S) {
    const int dim = P->dim, nV = dim+1;
    memset(S, 0, sizeof(*S));
    if (P->kernel.interaction == 2) {
        double bary[2] = {0., 0.};
        for (int v = 0; v < nV; v++) for (int j = 0; j < dim; j++) bary[j] += s2[v][j];
        for (int j = 0; j < dim; j++) bary[j] /= nV;
        if (is_inside(P, x, bary)) subs_identity(S, nV);
        return;
    }
    int ind[3] = {0, 0, 0}, numInside = 0;
    for (int j = 0; j < nV; j++) { ind[j] = is_inside(P, x, s2[j]); numInside += ind[j]; }
    double isec[2];
    if (dim == 1) {
        if (numInside == 0) {
            if (find_intersections(P, x, s2, 0, 1, isec) == 2) {
                S->A[0][0][0] = 1-isec[0]; S->A[0][1][0] = isec[0];
                S->A[0][1][1] = isec[1]; S->A[0][0][1] = 1.-isec[1];
                S->vol[0] = isec[1]-isec[0];
                S->n = 1;
            }
        } else if (numInside == 1) {
            int inside = 0;
            while (!ind[inside]) inside++;
            const int outside = (inside+1)%2;
            find_intersections(P, x, s2, inside, outside, isec);
            S->A[0][inside][inside] = 1.;
            S->A[0][outside][outside] = isec[0];
            S->A[0][inside][outside] = 1.-isec[0];
            S->vol[0] = isec[0];
            S->n = 1;
        } else subs_identity(S, 2);
        return;
    }
    if (numInside == 0) return;                 /* "There can be a nonzero intersection, but we ignore it" :664 */
    if (numInside == 1) {
        int inside = 0;
        while (!ind[inside]) inside++;
        const int o1 = (inside+1)%3, o2 = (inside+2)%3;
        find_intersections(P, x, s2, inside, o1, isec);
        const double c1 = isec[0];
        find_intersections(P, x, s2, inside, o2, isec);
        const double c2 = isec[0];
        const int num = find_intersections(P, x, s2, o1, o2, isec);
        if (num == 0) {
            S->A[0][inside][inside] = 1; S->A[0][inside][o1] = 1-c1; S->A[0][o1][o1] = c1;
            S->A[0][o2][o2] = c2; S->A[0][inside][o2] = 1-c2;
            S->vol[0] = c1*c2;
            S->n = 1;
        } else if (num == 2) {
            S->A[0][inside][inside] = 1; S->A[0][o1][o1] = c1; S->A[0][inside][o1] = 1-c1;
            S->A[0][o2][o2] = isec[0]; S->A[0][o1][o2] = 1-isec[0];
            S->vol[0] = c1*isec[0];
            S->A[1][inside][inside] = 1; S->A[1][o1][o1] = 1-isec[0]; S->A[1][o2][o1] = isec[0];
            S->A[1][o1][o2] = 1-isec[1]; S->A[1][o2][o2] = isec[1];
            S->vol[1] = isec[1]-isec[0];
            S->A[2][inside][inside] = 1; S->A[2][o1][o1] = 1-isec[1]; S->A[2][o2][o1] = isec[1];
            S->A[2][o2][o2] = c2; S->A[2][inside][o2] = 1-c2;
            S->vol[2] = c2*(1-isec[1]);
            S->n = 3;
        } else {
            S->A[0][inside][inside] = 1; S->A[0][o1][o1] = c1; S->A[0][inside][o1] = 1-c1;
            S->A[0][o2][o2] = isec[0]; S->A[0][o1][o2] = 1-isec[0];
            S->vol[0] = c1*isec[0];
            S->A[1][inside][inside] = 1; S->A[1][o1][o1] = 1-isec[0]; S->A[1][o2][o1] = isec[0];
            S->A[1][o2][o2] = c2; S->A[1][inside][o2] = 1-c2;
            S->vol[1] = c2*(1-isec[0]);
            S->n = 2;
        }
        return;
    }
    if (numInside == 2) {
        int outside = 0;
        while (ind[outside]) outside++;
        const int i1 = (outside+1)%3, i2 = (outside+2)%3;
        find_intersections(P, x, s2, outside, i1, isec);
        const double c1 = isec[0];
        find_intersections(P, x, s2, outside, i2, isec);
        const double c2 = isec[0];
        double d1 = 0., d2 = 0.;
        for (int k = 0; k < 2; k++) {
            const double t1 = s2[i2][k]-(c1*s2[i1][k]+(1-c1)*s2[outside][k]);
            const double t2 = s2[i1][k]-(c2*s2[i2][k]+(1-c2)*s2[outside][k]);
            d1 += t1*t1; d2 += t2*t2;
        }
        if (d1 < d2) {
            S->A[0][i2][i2] = 1; S->A[0][outside][outside] = 1-c2; S->A[0][i2][outside] = c2;
            S->A[0][i1][i1] = c1; S->A[0][outside][i1] = 1-c1;
            S->vol[0] = c1*(1-c2);
            S->A[1][i1][i1] = 1; S->A[1][i2][i2] = 1; S->A[1][outside][outside] = 1-c1; S->A[1][i1][outside] = c1;
            S->vol[1] = 1-c1;
        } else {
            S->A[0][i1][i1] = 1; S->A[0][i2][i2] = c2; S->A[0][outside][i2] = 1-c2;
            S->A[0][outside][outside] = 1-c1; S->A[0][i1][outside] = c1;
            S->vol[0] = c2*(1-c1);
            S->A[1][i1][i1] = 1; S->A[1][i2][i2] = 1; S->A[1][outside][outside] = 1-c2; S->A[1][i2][outside] = c2;
            S->vol[1] = 1-c2;
        }
        S->n = 2;
        return;
    }
    subs_identity(S, 3);
}

/* local shape functions at barycentric coordinates (fem/PyNucleus_fem/DoFMaps.pyx:1854-2025): P1, P2 on triangles */
static void shape_eval(const nlo_problem *P, const double *lam, double *phi) {
    if (P->dpe == P->dim+1) { for (int k = 0; k < P->dpe; k++) phi[k] = lam[k]; return; }
    phi[0] = lam[0]*(2*lam[0]-1); phi[1] = lam[1]*(2*lam[1]-1); phi[2] = lam[2]*(2*lam[2]-1);
    phi[3] = 4*lam[0]*lam[1]; phi[4] = 4*lam[1]*lam[2]; phi[5] = 4*lam[0]*lam[2];
}

/* NO:790-847 eval_distant, cut branch: sub-simplices of simplex1, per quadrature node sub-simplices of simplex2 */
static void eval_distant_cut(const nlo_problem *P, double s1[MAXV][2], double s2[MAXV][2], double vol, int order, double *contrib,
                             int64_t *nevals) {
    const int dim = P->dim, nV = dim+1, dpe = P->dpe;
    const int off = P->dist_off[order], n = P->dist_off[order+1]-off;
    const double *bary = P->dist_bary+3*off, *w = P->dist_w+off;
    const int E = (2*dpe)*(2*dpe+1)/2;
    for (int k = 0; k < E; k++) contrib[k] = 0.;
    subs_t S1, S2;
    subs_simplex(P, s1, s2, &S1);
    for (int a = 0; a < S1.n; a++)
        for (int i = 0; i < n; i++) {
            double lx[3] = {0., 0., 0.}, x[2] = {0., 0.}, psi[2*MAXDPE];
            for (int k = 0; k < nV; k++) {
                lx[k] = S1.b[a][k];
                for (int j = 0; j < nV; j++) lx[k] += S1.A[a][k][j]*bary[3*i+j];
            }
            for (int k = 0; k < nV; k++) for (int m = 0; m < dim; m++) x[m] += lx[k]*s1[k][m];
            shape_eval(P, lx, psi);
            subs_node(P, x, s2, &S2);
            for (int b = 0; b < S2.n; b++)
                for (int j = 0; j < n; j++) {
                    double ly[3] = {0., 0., 0.}, y[2] = {0., 0.};
                    for (int k = 0; k < nV; k++) for (int jj = 0; jj < nV; jj++) ly[k] += S2.A[b][k][jj]*bary[3*j+jj];
                    for (int k = 0; k < nV; k++) for (int m = 0; m < dim; m++) y[m] += ly[k]*s2[k][m];
                    shape_eval(P, ly, psi+dpe);
                    double d2 = 0.;
                    for (int m = 0; m < dim; m++) d2 += (x[m]-y[m])*(x[m]-y[m]);
                    double val = w[i]*w[j]*kernel_eval(&P->kernel, d2);
                    val *= S1.vol[a]*S2.vol[b]*vol;
                    (*nevals)++;
                    int k = 0;
                    for (int I = 0; I < 2*dpe; I++) {
                        const double pI = I < dpe ? psi[I] : -psi[I];
                        for (int J = I; J < 2*dpe; J++) {
                            const double pJ = J < dpe ? psi[J] : -psi[J];
                            contrib[k++] += val*pI*pJ;
                        }
                    }
                }
        }
}

/* NO:549-600 addQuadRule: PSI[2 dpe][n*n], rows 0..dpe-1 = phi_I(x_i), rows dpe.. = -phi_I(y_j), k = i*n+j */
static double *build_distant_psi(const nlo_problem *P, int order) {
    const int dpe = P->dpe, off = P->dist_off[order], n = P->dist_off[order+1]-off;
    const double *phi = P->dist_phi+(size_t)off*dpe;
    double *PSI = (double*)malloc(sizeof(double)*2*dpe*(size_t)n*n);
    for (int I = 0; I < dpe; I++)
        for (int i = 0; i < n; i++)
            for (int j = 0; j < n; j++) {
                PSI[(size_t)I*n*n+i*n+j] = phi[i*dpe+I];
                PSI[(size_t)(I+dpe)*n*n+i*n+j] = -phi[j*dpe+I];
            }
    return PSI;
}

/* NO:722-789 eval_distant, uncut branch.  scratch holds temp[n*n], x[n][2], y[n][2]. */
static void eval_distant(const nlo_problem *P, double s1[MAXV][2], double s2[MAXV][2], double vol, int order,
                         const double *PSI, double *scratch, double *contrib) {
    const int dim = P->dim, nV = dim+1, dpe = P->dpe;
    const int off = P->dist_off[order], n = P->dist_off[order+1]-off, nn = n*n;
    const double *bary = P->dist_bary+3*off, *w = P->dist_w+off;
    double *temp = scratch, *x = scratch+nn, *y = x+2*n;
    for (int i = 0; i < n; i++)                 /* Q:76-87 nodesInGlobalCoords */
        for (int m = 0; m < dim; m++) {
            double a = 0., b = 0.;
            for (int k = 0; k < nV; k++) { a += bary[3*i+k]*s1[k][m]; b += bary[3*i+k]*s2[k][m]; }
            x[2*i+m] = a; y[2*i+m] = b;
        }
    int k = 0;
    for (int i = 0; i < n; i++)
        for (int j = 0; j < n; j++) {
            double d2 = 0.;
            for (int m = 0; m < dim; m++) d2 += (x[2*i+m]-y[2*j+m])*(x[2*i+m]-y[2*j+m]);
            temp[k++] = (w[i]*w[j])*kernel_eval(&P->kernel, d2);   /* Q:224 weights[k] = w1[i]*w2[j] */
        }
    k = 0;
    for (int I = 0; I < 2*dpe; I++)
        for (int J = I; J < 2*dpe; J++) {
            const double *pI = PSI+(size_t)I*nn, *pJ = PSI+(size_t)J*nn;
            double val = 0.;
            for (int i = 0; i < nn; i++) val += temp[i]*pI[i]*pJ[i];
            contrib[k++] = val*vol;
        }
}

/* NO:722-789 (uncut distant pairs) and FL2:823-891 / FL1:349-407 (singular pairs) */
void nlo_eval(const nlo_problem *P0, int c1, int c2, int panel, const int *perm1, const int *perm2, const int *perm,
              double *contrib, int64_t *nevals) {
    const nlo_problem *P = class_of_cells(P0, c1, c2);
    const int dim = P->dim, nV = dim+1, dpe = P->dpe;
    const int E = (2*dpe)*(2*dpe+1)/2;
    double s1[MAXV][2], s2[MAXV][2], ce[2];
    simplex_of(P, c1, s1, ce);
    simplex_of(P, c2, s2, ce);
    for (int k = 0; k < E; k++) contrib[k] = 0.;
    if (panel >= 1 && rel_position(P, s1, nV, s2, nV) == NLO_CUT) {
        eval_distant_cut(P, s1, s2, P->vol[c1]*P->vol[c2], panel, contrib, nevals);
        return;
    }
    if (panel >= 1) {
        const int off = P->dist_off[panel], n = P->dist_off[panel+1]-off;
        double *PSI = build_distant_psi(P, panel);
        double *scratch = (double*)malloc(sizeof(double)*((size_t)n*n+4*(size_t)n));
        eval_distant(P, s1, s2, P->vol[c1]*P->vol[c2], panel, PSI, scratch, contrib);
        *nevals += (int64_t)n*n;
        free(scratch); free(PSI);
        return;
    }
    const int slot = -panel-1;                  /* -1 vertex, -2 edge, -3 face */
    const int M = P->sing_M[slot], rows = P->sing_rows[slot];
    const double *nodes = P->sing_nodes[slot], *w = P->sing_w[slot], *PSI = P->sing_psi[slot];
    const double vol = P->sing_fac*P->vol[c1]*P->vol[c2];
    double *temp = (double*)malloc(sizeof(double)*M);
    for (int m = 0; m < M; m++) {
        double d2 = 0.;
        for (int j = 0; j < dim; j++) {
            double xx = 0., yy = 0.;
            for (int k = 0; k < nV; k++) {
                xx += s1[perm1[k]][j]*nodes[(size_t)k*M+m];
                yy += s2[perm2[k]][j]*nodes[(size_t)(nV+k)*M+m];
            }
            d2 += (xx-yy)*(xx-yy);
        }
        temp[m] = w[m]*kernel_eval(&P->kernel, d2);
    }
    *nevals += M;
    for (int I = 0; I < rows; I++) {
        int i = perm[I];
        for (int J = I; J < rows; J++) {
            int j = perm[J];
            int k = j < i ? 2*dpe*j-(j*(j+1) >> 1)+i : 2*dpe*i-(i*(i+1) >> 1)+j;
            double val = 0.;
            for (int m = 0; m < M; m++) val += temp[m]*PSI[(size_t)I*M+m]*PSI[(size_t)J*M+m];
            contrib[k] = val*vol;
        }
    }
    free(temp);
}

static void facet_of(const nlo_problem *P, int b, double s[MAXV][2], double center[2], double *vol) {
    int nF = P->dim;
    center[0] = center[1] = 0.;
    for (int m = 0; m < nF; m++) {
        int v = P->bcells[b*nF+m];
        for (int l = 0; l < P->dim; l++) { s[m][l] = P->vertices[v*P->dim+l]; center[l] += s[m][l]; }
    }
    for (int l = 0; l < P->dim; l++) center[l] *= 1./nF;
    if (P->dim == 2) *vol = sqrt((s[1][0]-s[0][0])*(s[1][0]-s[0][0]) + (s[1][1]-s[0][1])*(s[1][1]-s[0][1]));
    else *vol = 1.;
}

/* NO:280-378 with cells2 = boundary facets (symmetricCells False), NO:515-533 */
int nlo_panel_boundary(const nlo_problem *P, int c1, int b, int *perm1, int *perm2, int *perm) {
    const int nV = P->dim+1, nF = P->dim, dpe = P->dpe;
    for (int k = 0; k < nV; k++) perm1[k] = k;
    for (int k = 0; k < nF; k++) perm2[k] = k;
    for (int k = 0; k < dpe; k++) perm[k] = k;
    int mask1 = 0, mask2 = 0, common = 0;
    for (int a = 0; a < nV; a++) {
        int v1 = P->cells[c1*nV+a];
        for (int f = 0; f < nF; f++) {
            if (mask2 & (1 << f)) continue;
            if (v1 == P->bcells[b*nF+f]) {
                perm1[common] = a; perm2[common] = f;
                mask1 += (1 << a); mask2 += (1 << f);
                common++;
                break;
            }
        }
    }
    if (common == 0) {
        double s1[MAXV][2], s2[MAXV][2], ce1[2], ce2[2], vol2;
        simplex_of(P, c1, s1, ce1);
        facet_of(P, b, s2, ce2, &vol2);
        double d2 = 0.;
        for (int j = 0; j < P->dim; j++) d2 += (ce1[j]-ce2[j])*(ce1[j]-ce2[j]);
        /* h2 = get_h_surface_simplex: edge length in 2D, 1 in 1D (nonlocalOperator.pyx:121-122,164-171) */
        return quad_order(&class_of_cell_facet(P, c1, b)->bqo, P->H0, P->h[c1], vol2, sqrt(d2));
    }
    int i = 0;
    for (int k = common; k < nV; k++) { while (mask1 & (1 << i)) i++; perm1[k] = i; mask1 += (1 << i); }
    i = 0;
    for (int k = common; k < nF; k++) { while (mask2 & (1 << i)) i++; perm2[k] = i; mask2 += (1 << i); }
    const int *t1 = P->dof_perm_table + perm_rank(perm1, nV)*dpe;
    for (int k = 0; k < dpe; k++) perm[k] = t1[k];
    return -common;
}

/* NO:1022-1108 eval_distant_boundary, FL2:1324-1407, FL1:726-785 */
void nlo_eval_boundary(const nlo_problem *P0, int c1, int b, int panel, const int *perm1, const int *perm2, const int *perm,
                       double *contrib, int64_t *nevals) {
    const nlo_problem *P = class_of_cell_facet(P0, c1, b);
    const int dim = P->dim, nV = dim+1, nF = dim, dpe = P->dpe;
    const int E = dpe*(dpe+1)/2;
    double s1[MAXV][2], s2[MAXV][2], ce[2], vol2, nrm[2] = {0., 0.};
    simplex_of(P, c1, s1, ce);
    facet_of(P, b, s2, ce, &vol2);
    if (dim == 2) {
        nrm[0] = s2[1][1]-s2[0][1];
        nrm[1] = s2[0][0]-s2[1][0];
        double v = 1./sqrt(nrm[0]*nrm[0]+nrm[1]*nrm[1]);
        nrm[0] *= v; nrm[1] *= v;
    }
    for (int k = 0; k < E; k++) contrib[k] = 0.;
    if (panel >= 1) {
        const int off = P->dist_off[panel], n = P->dist_off[panel+1]-off;
        const int foff = P->bfacet_off[panel], nf = P->bfacet_off[panel+1]-foff;
        const double *bary = P->dist_bary+3*off, *w = P->dist_w+off, *phi = P->dist_phi+(size_t)off*dpe;
        const double *fb = P->bfacet_bary+2*foff, *fw = P->bfacet_w+foff;
        const double vol = P->vol[c1]*vol2;
        double *temp = (double*)malloc(sizeof(double)*(size_t)n*nf);
        for (int k = 0; k < n; k++)
            for (int m = 0; m < nf; m++) {
                double x[2] = {0., 0.}, y[2] = {0., 0.}, wv[2], normW = 0., nw;
                for (int l = 0; l < dim; l++) {
                    for (int q = 0; q < nV; q++) x[l] += bary[3*k+q]*s1[q][l];
                    for (int q = 0; q < nF; q++) y[l] += fb[2*m+q]*s2[q][l];
                }
                double d2 = 0.;
                for (int l = 0; l < dim; l++) d2 += (x[l]-y[l])*(x[l]-y[l]);
                if (dim == 1) nw = 1.;
                else {
                    for (int l = 0; l < dim; l++) { wv[l] = y[l]-x[l]; normW += wv[l]*wv[l]; }
                    normW = 1./sqrt(normW);
                    nw = nrm[0]*wv[0]*normW + nrm[1]*wv[1]*normW;
                }
                temp[k*nf+m] = (w[k]*fw[m])*nw*kernel_eval(&P->bkernel, d2);
            }
        *nevals += (int64_t)n*nf;
        int e = 0;
        for (int I = 0; I < dpe; I++)
            for (int J = I; J < dpe; J++) {
                double val = 0.;
                for (int k = 0; k < n; k++)
                    for (int m = 0; m < nf; m++) val += temp[k*nf+m]*phi[k*dpe+I]*phi[k*dpe+J];
                contrib[e++] = val*vol;
            }
        free(temp);
        return;
    }
    const int slot = -panel-1;
    const int M = P->bsing_M[slot];
    const double *nodes = P->bsing_nodes[slot], *w = P->bsing_w[slot], *PHI = P->bsing_phi[slot];
    const double vol = dim == 2 ? P->bsing_fac*P->vol[c1]*vol2 : P->bsing_fac*P->vol[c1];
    double *temp = (double*)malloc(sizeof(double)*M);
    for (int m = 0; m < M; m++) {
        double wv[2] = {0., 0.}, normW = 0., nw = 1.;
        for (int j = 0; j < dim; j++) {
            double xx = 0., yy = 0.;
            for (int k = 0; k < nV; k++) xx += s1[perm1[k]][j]*nodes[(size_t)k*M+m];
            for (int k = 0; k < nF; k++) yy += s2[perm2[k]][j]*nodes[(size_t)(nV+k)*M+m];
            wv[j] = xx-yy;
            normW += wv[j]*wv[j];
        }
        if (dim == 2) {
            double inv = 1./sqrt(normW);
            nw = nrm[0]*wv[0]*inv + nrm[1]*wv[1]*inv;
        }
        temp[m] = w[m]*nw*kernel_eval(&P->bkernel, normW);
    }
    *nevals += M;
    for (int I = 0; I < dpe; I++) {
        int i = perm[I];
        for (int J = I; J < dpe; J++) {
            int j = perm[J];
            int k = j < i ? dpe*j-(j*(j+1) >> 1)+i : dpe*i-(i*(i+1) >> 1)+j;
            double val = 0.;
            for (int m = 0; m < M; m++) val += temp[m]*PHI[(size_t)I*M+m]*PHI[(size_t)J*M+m];
            contrib[k] = val*vol;
        }
    }
    free(temp);
}

/* NA:204-221 addToMatrixElemElemSym */
static void scatter_elem_elem_sym(double *A, int64_t N, const int *ld, int n2, const double *contrib, double fac) {
    int k = 0;
    for (int p = 0; p < n2; p++) {
        int I = ld[p];
        if (I >= 0) {
            A[(int64_t)I*N+I] += fac*contrib[k];
            k++;
            for (int q = p+1; q < n2; q++) {
                int J = ld[q];
                if (J >= 0) {
                    A[(int64_t)I*N+J] += fac*contrib[k];
                    A[(int64_t)J*N+I] += fac*contrib[k];
                }
                k++;
            }
        } else k += n2-p;
    }
}

/* NA:1262-1473 getDense: 'interior' loop NA:1386-1428 (symmetric cells/local matrix) and
 * 'zeroExterior' loop NA:1430-1448 (scatter NA:152-168 is the same routine on dpe local DoFs). */
int nlo_get_dense_rows(const nlo_problem *P, double *A, int zero_exterior, int cell_start, int cell_end,
                       int64_t *counters, double *seconds, int store) {
    const int dpe = P->dpe, nV = P->dim+1;
    const int64_t N = P->num_dofs;
    if (dpe > MAXDPE || nV > MAXV) return -1;
    double contrib[MAXE];
    int perm1[MAXV], perm2[MAXV], perm[2*MAXDPE], ld[2*MAXDPE];
    double *psi_cache[NLO_MAX_ORDER+1];         /* distantQuadRulesPtr, NO:441-443 */
    memset(psi_cache, 0, sizeof(psi_cache));
    int nmax = 1;
    for (int q = 0; q <= P->qmax; q++) { int n = P->dist_off[q+1]-P->dist_off[q]; if (n > nmax) nmax = n; }
    double *scratch = (double*)malloc(sizeof(double)*((size_t)nmax*nmax+4*(size_t)nmax));
    memset(counters, 0, sizeof(int64_t)*NLO_NUM_COUNTERS);
    double t0 = now_s();
    for (int c1 = cell_start; c1 < cell_end; c1++) {
        for (int c2 = c1; c2 < P->nc; c2++) {
            counters[0]++;
            int skip = 1;                       /* NA:138-150 getDoFsElemElem */
            for (int p = 0; p < dpe; p++) { ld[p] = P->dofs[c1*dpe+p]; skip = skip && ld[p] < 0; }
            for (int p = 0; p < dpe; p++) { ld[dpe+p] = P->dofs[c2*dpe+p]; skip = skip && ld[dpe+p] < 0; }
            if (skip) continue;
            int panel = nlo_panel(P, c1, c2, perm1, perm2, perm);
            if (panel == NLO_IGNORED) continue;
            if (panel >= 1 && (panel > P->qmax || P->dist_off[panel+1] == P->dist_off[panel])) return -(1000+panel);
            counters[1]++;
            if (panel >= 1) counters[8+panel]++; else counters[8+NLO_MAX_ORDER+(-panel-1)]++;
            if (panel >= 1 && !isinf(P->kernel.horizon2)) {
                nlo_eval(P, c1, c2, panel, perm1, perm2, perm, contrib, &counters[2]);
            } else if (panel >= 1) {
                if (!psi_cache[panel]) psi_cache[panel] = build_distant_psi(P, panel);
                double s1[MAXV][2], s2[MAXV][2], ce[2];
                simplex_of(P, c1, s1, ce);
                simplex_of(P, c2, s2, ce);
                eval_distant(class_of_cells(P, c1, c2), s1, s2, P->vol[c1]*P->vol[c2], panel, psi_cache[panel], scratch, contrib);
                int n = P->dist_off[panel+1]-P->dist_off[panel];
                counters[2] += (int64_t)n*n;
            } else
                nlo_eval(P, c1, c2, panel, perm1, perm2, perm, contrib, &counters[2]);
            if (store) scatter_elem_elem_sym(A, N, ld, 2*dpe, contrib, c1 == c2 ? 1. : 2.);
        }
    }
    double t1 = now_s();
    if (zero_exterior) {
        for (int c1 = cell_start; c1 < cell_end; c1++) {
            for (int p = 0; p < dpe; p++) ld[p] = P->dofs[c1*dpe+p];
            for (int b = 0; b < P->nb; b++) {
                int panel = nlo_panel_boundary(P, c1, b, perm1, perm2, perm);
                if (panel >= 1 && (panel > P->qmax || P->dist_off[panel+1] == P->dist_off[panel]
                                   || P->bfacet_off[panel+1] == P->bfacet_off[panel])) return -(2000+panel);
                counters[3]++;
                nlo_eval_boundary(P, c1, b, panel, perm1, perm2, perm, contrib, &counters[4]);
                if (store) scatter_elem_elem_sym(A, N, ld, dpe, contrib, 1.);
            }
        }
    }
    double t2 = now_s();
    if (seconds) { seconds[0] = t1-t0; seconds[1] = t2-t1; }
    for (int q = 0; q <= NLO_MAX_ORDER; q++) free(psi_cache[q]);
    free(scratch);
    return 0;
}

int nlo_get_dense(const nlo_problem *P, double *A, int zero_exterior, int cell_start, int cell_end,
                  int64_t *counters, double *seconds) {
    return nlo_get_dense_rows(P, A, zero_exterior, cell_start, cell_end, counters, seconds, 1);
}

/* ---- non-symmetric kernels, order s(x) per quadrature point ---------------------------------------------------------
 * fractionalOrders.pyx:338-540: constantExtended, smoothStep, linearStep, smoothStepRadial */
/* lookupExtended.evalPtr (fractionalOrders.pyx:573-587): find the cell that holds x (cellFinder2), evaluate the P1 function
 * there.  Point location by barycentric coordinates over all cells, starting at the cell found last. */
static double fe_order_lookup(const nlo_problem *P, const double *x) {
    static int last = 0;
    const int nV = P->dim+1;
    double best_val = 0., best_min = -1e300;
    for (int t = 0; t < P->nc; t++) {
        const int c = (last+t) % P->nc;
        double lam[3];
        const int32_t *cv = P->cells+(size_t)c*nV;
        if (P->dim == 1) {
            const double a = P->vertices[cv[0]], b = P->vertices[cv[1]];
            lam[1] = (x[0]-a)/(b-a); lam[0] = 1.-lam[1]; lam[2] = 0.;
        } else {
            const double *v0 = P->vertices+2*(size_t)cv[0], *v1 = P->vertices+2*(size_t)cv[1], *v2 = P->vertices+2*(size_t)cv[2];
            const double det = (v1[0]-v0[0])*(v2[1]-v0[1])-(v2[0]-v0[0])*(v1[1]-v0[1]);
            lam[1] = ((x[0]-v0[0])*(v2[1]-v0[1])-(v2[0]-v0[0])*(x[1]-v0[1]))/det;
            lam[2] = ((v1[0]-v0[0])*(x[1]-v0[1])-(x[0]-v0[0])*(v1[1]-v0[1]))/det;
            lam[0] = 1.-lam[1]-lam[2];
        }
        double mn = lam[0] < lam[1] ? lam[0] : lam[1];
        if (P->dim == 2 && lam[2] < mn) mn = lam[2];
        double val = 0.;
        for (int k = 0; k < nV; k++) val += lam[k]*P->pw_vertex_s[cv[k]];
        if (mn >= -1e-12) { last = c; return val; }
        if (mn > best_min) { best_min = mn; best_val = val; }
    }
    return best_val;                             /* a point on the curved boundary side of the mesh: the nearest cell */
}

double nlo_pw_order(const nlo_problem *P, const double *x) {
    const double *p = P->pw_p;                  /* sl, sr, r, interface | radius, slope */
    switch (P->pw_type) {
    case 1: return p[0];
    case 5: return fe_order_lookup(P, x);
    case 2:
        if (x[0] < p[3]-p[2]) return p[0];
        else if (x[0] > p[3]+p[2]) return p[1];
        return p[0] + (p[1]-p[0]) * (3.0*pow((x[0]-p[3])*p[4]+0.5, 2.0) - 2.0*pow((x[0]-p[3])*p[4]+0.5, 3.0));
    case 3:
        if (x[0] < p[3]-p[2]) return p[0];
        else if (x[0] > p[3]+p[2]) return p[1];
        return p[0] + p[4]*(x[0]-p[3]+p[2]);
    default: {
        double r = 0.;
        for (int k = 0; k < P->dim; k++) r += x[k]*x[k];
        r = sqrt(r);
        if (r < p[3]-p[2]) return p[0];
        else if (r > p[3]+p[2]) return p[1];
        return p[0] + (p[1]-p[0]) * (3.0*pow((r-p[3])*p[4]+0.5, 2.0) - 2.0*pow((r-p[3])*p[4]+0.5, 3.0));
    }
    }
}

/* variableFractionalLaplacianScaling.evalPtr (kernelNormalization.pyx:416-440, infinite horizon, derivative 0); the
 * boundary twin multiplies with phi = inverseTwoPoint(s) (KC:1990-1994) */
static double pw_scaling(const nlo_problem *P, double s, int boundary) {
    double C = 0.5;
    if (P->pw_normalized) C = pow(2.0, 2.0*s) * s * tgamma(s+0.5*P->dim) * pow(M_PI, -0.5*P->dim) / tgamma(1.0-s) * 0.5;
    return boundary ? (1./s)*C : C;
}

/* updateAndEvalFractional (KC:596-622) + fracKernelInfinite{1,2}D[boundary] (KC:159-174, 216-231): the order is s(x, y) = sFun(x) */
static double pw_kernel(const nlo_problem *P, const double *x, const double *y, int boundary) {
    const double s = nlo_pw_order(P, x);
    const double C = pw_scaling(P, s, boundary);
    double d2 = 0.;
    for (int k = 0; k < P->dim; k++) d2 += (x[k]-y[k])*(x[k]-y[k]);
    const double e = boundary ? (P->dim == 1 ? -s : -0.5-s) : (P->dim == 1 ? -0.5-s : -1.-s);
    return C*pow(d2, e);
}

/* Kernel.evalParamsOnSimplices for a non-symmetric order (KC:1825-1846): the largest order over both centres and all vertices */
static double pw_svalue_simplices(const nlo_problem *P, const double *ce1, const double *ce2, double s1[MAXV][2], int n1,
                                  double s2[MAXV][2], int n2) {
    double sValue = 0.;
    sValue = fmax(sValue, nlo_pw_order(P, ce1));
    sValue = fmax(sValue, nlo_pw_order(P, ce2));
    for (int i = 0; i < n1; i++) sValue = fmax(sValue, nlo_pw_order(P, s1[i]));
    for (int i = 0; i < n2; i++) sValue = fmax(sValue, nlo_pw_order(P, s2[i]));
    return sValue;
}

double nlo_pw_svalue(const nlo_problem *P, int c1, int c2) {
    double s1[MAXV][2], s2[MAXV][2], ce1[2], ce2[2];
    simplex_of(P, c1, s1, ce1);
    simplex_of(P, c2, s2, ce2);
    return pw_svalue_simplices(P, ce1, ce2, s1, P->dim+1, s2, P->dim+1);
}

static int pw_key(const double *keys, int n, double sv) {       /* nearest key (host and oracle evaluate s(x) independently) */
    int best = 0;
    for (int k = 1; k < n; k++) if (fabs(keys[k]-sv) < fabs(keys[best]-sv)) best = k;
    return best;
}

/* FL2:915-935 / FL1:431-450 */
static nlo_order_formula pw_formula(const nlo_problem *P, double sv) {
    nlo_order_formula F;
    memset(&F, 0, sizeof(F));
    F.c0 = P->pw_c0;
    if (P->dim == 2) { F.a = sv-1.; F.b = 1.; F.e = sv; F.den0 = 0.4; }
    else { F.a = 2.*sv-1.; F.b = 0.; F.e = 2.*sv; F.den0 = 0.8; }
    return F;
}

/* FL2:1226-1243 / FL1:644-660 with s = max(0.5(-singularity-1), 0), singularity = 1-d-2 sValue */
static nlo_order_formula pw_formula_boundary(const nlo_problem *P, double sv) {
    nlo_order_formula F;
    memset(&F, 0, sizeof(F));
    F.c0 = P->pw_bc0;
    const double st = fmax(0.5*(-(1.-P->dim-2.*sv)-1.), 0.);
    if (P->dim == 2) { F.a = st-1.; F.b = 1.; F.e = st; F.den0 = 0.35; F.clip_num = 1; }
    else { F.a = 2.*st-1.; F.b = 0.; F.e = 2.*st; F.den0 = 0.8; }
    return F;
}

/* NO:849-930 eval_distant_nonsym (uncut) with the PHI / PSI tables of addQuadRule_nonSym NO:602-662;
 * contrib[(2 dpe)^2], k = I*(2 dpe)+J */
static void eval_distant_nonsym(const nlo_problem *P, double s1[MAXV][2], double s2[MAXV][2], double vol, int order, double *contrib) {
    const int dim = P->dim, nV = dim+1, dpe = P->dpe, n2 = 2*dpe;
    const int off = P->dist_off[order], n = P->dist_off[order+1]-off, nn = n*n;
    const double *bary = P->dist_bary+3*off, *w = P->dist_w+off, *phi = P->dist_phi+(size_t)off*dpe;
    double *temp = (double*)malloc(sizeof(double)*(2*(size_t)nn+4*(size_t)n)), *temp2 = temp+nn, *x = temp2+nn, *y = x+2*n;
    for (int i = 0; i < n; i++)
        for (int m = 0; m < dim; m++) {
            double a = 0., b = 0.;
            for (int k = 0; k < nV; k++) { a += bary[3*i+k]*s1[k][m]; b += bary[3*i+k]*s2[k][m]; }
            x[2*i+m] = a; y[2*i+m] = b;
        }
    int k = 0;
    for (int i = 0; i < n; i++)
        for (int j = 0; j < n; j++) {
            const double ww = w[i]*w[j];
            temp[k] = ww*pw_kernel(P, x+2*i, y+2*j, 0);
            temp2[k] = ww*pw_kernel(P, y+2*j, x+2*i, 0);
            k++;
        }
    k = 0;
    for (int I = 0; I < n2; I++)
        for (int J = 0; J < n2; J++) {
            double val = 0.;
            for (int i = 0; i < n; i++)
                for (int j = 0; j < n; j++) {
                    const double PHI0 = I < dpe ? phi[i*dpe+I] : 0., PHI1 = I < dpe ? 0. : phi[j*dpe+I-dpe];
                    const double PSIJ = J < dpe ? phi[i*dpe+J] : -phi[j*dpe+J-dpe];
                    val += (temp[i*n+j]*PHI0 - temp2[i*n+j]*PHI1)*PSIJ;
                }
            contrib[k++] = val*vol;
        }
    free(temp);
}

/* FL2:1133-1184 / FL1:548-604 eval for touching pairs */
static void eval_singular_nonsym(const nlo_problem *P, double s1[MAXV][2], double s2[MAXV][2], double vol12, int slot, int key,
                                 const int *perm1, const int *perm2, const int *perm, double *contrib, int64_t *nevals) {
    const int dim = P->dim, nV = dim+1, dpe = P->dpe, n2 = 2*dpe;
    const int M = P->sing_M[slot], rows = P->sing_rows[slot];
    const double *nodes = P->pw_nodes[slot]+(size_t)key*2*nV*M, *w = P->pw_w[slot]+(size_t)key*M;
    const double *PHI0 = P->pw_phi0[slot]+(size_t)key*rows*M, *PHI1 = P->pw_phi1[slot]+(size_t)key*rows*M;
    const double vol = P->sing_fac*vol12;
    double *temp = (double*)malloc(sizeof(double)*2*M), *temp2 = temp+M;
    for (int m = 0; m < M; m++) {
        double x[2] = {0., 0.}, y[2] = {0., 0.};
        for (int j = 0; j < dim; j++)
            for (int k = 0; k < nV; k++) {
                x[j] += s1[perm1[k]][j]*nodes[(size_t)k*M+m];
                y[j] += s2[perm2[k]][j]*nodes[(size_t)(nV+k)*M+m];
            }
        temp[m] = w[m]*pw_kernel(P, x, y, 0);
        temp2[m] = w[m]*pw_kernel(P, y, x, 0);
    }
    *nevals += M;
    for (int k = 0; k < n2*n2; k++) contrib[k] = 0.;
    for (int I = 0; I < rows; I++) {
        const int i = perm[I];
        for (int J = 0; J < rows; J++) {
            const int j = perm[J];
            double val = 0.;
            for (int m = 0; m < M; m++)
                val += (temp[m]*PHI0[(size_t)I*M+m] - temp2[m]*PHI1[(size_t)I*M+m]) * (PHI0[(size_t)J*M+m] - PHI1[(size_t)J*M+m]);
            contrib[i*n2+j] = val*vol;
        }
    }
    free(temp);
}

/* getPanelType (NO:493-540) for the non-symmetric local matrices (any order of c1, c2); returns the panel and, for
 * pointwise orders, the pair's order.  Piecewise orders: the formula of the class of (label c1, label c2) (evalParams at
 * the two centres in THIS orientation, NO:509-513) */
static int panel_nonsym(const nlo_problem *P, int c1, int c2, int *perm1, int *perm2, int *perm, double *sv) {
    const int nV = P->dim+1, dpe = P->dpe;
    for (int k = 0; k < nV; k++) { perm1[k] = k; perm2[k] = k; }
    for (int k = 0; k < 2*dpe; k++) perm[k] = k;
    *sv = P->pw_type ? nlo_pw_svalue(P, c1, c2) : 0.;
    if (c1 == c2) return -nV;
    int mask1 = 0, mask2 = 0, common = 0;
    for (int a = 0; a < nV; a++) {
        int v1 = P->cells[c1*nV+a];
        for (int b = 0; b < nV; b++) {
            if (mask2 & (1 << b)) continue;
            if (v1 == P->cells[c2*nV+b]) {
                perm1[common] = a; perm2[common] = b;
                mask1 += (1 << a); mask2 += (1 << b);
                common++;
                break;
            }
        }
    }
    if (common == 0) {
        double s1[MAXV][2], s2[MAXV][2], ce1[2], ce2[2];
        simplex_of(P, c1, s1, ce1);
        simplex_of(P, c2, s2, ce2);
        double d2 = 0.;
        for (int j = 0; j < P->dim; j++) d2 += (ce1[j]-ce2[j])*(ce1[j]-ce2[j]);
        const nlo_order_formula F = P->pw_type ? pw_formula(P, *sv) : class_of_cells(P, c1, c2)->qo;
        /* h = get_h_simplex (nonlocalOperator.pyx:114-118, 152-160): the longest edge, which is what hVector holds */
        return quad_order(&F, P->H0, P->h[c1], P->h[c2], sqrt(d2));
    }
    int i = 0;
    for (int k = common; k < nV; k++) { while (mask1 & (1 << i)) i++; perm1[k] = i; mask1 += (1 << i); }
    i = 0;
    for (int k = common; k < nV; k++) { while (mask2 & (1 << i)) i++; perm2[k] = i; mask2 += (1 << i); }
    const int *t1 = P->dof_perm_table + perm_rank(perm1, nV)*dpe;
    const int *t2 = P->dof_perm_table + perm_rank(perm2, nV)*dpe;
    const int dpv = P->dofs_per_vertex, dped = P->dofs_per_edge;
    for (int k = 0; k < dpe; k++) perm[k] = t1[k];
    if (common == 1) {
        for (int k = dpv; k < dpe; k++) perm[dpe+k-dpv] = dpe+t2[k];
    } else if (common == 2) {
        for (int k = 2*dpv; k < nV*dpv; k++) perm[dpe+k-2*dpv] = dpe+t2[k];
        for (int k = nV*dpv+dped; k < dpe; k++) perm[dpe+k-2*dpv-dped] = dpe+t2[k];
    }
    return -common;
}

/* NA:222-253 addToMatrixElemElem */
static void scatter_elem_elem(double *A, int64_t N, const int *ld, int n2, const double *contrib, double fac) {
    int k = 0;
    for (int p = 0; p < n2; p++) {
        const int I = ld[p];
        if (I >= 0) {
            for (int q = 0; q < n2; q++) {
                const int J = ld[q];
                if (J >= 0) A[(int64_t)I*N+J] += fac*contrib[k];
                k++;
            }
        } else k += n2;
    }
}

/* boundary term with the pointwise kernel: NO:1022-1108, FL2:1324-1407, FL1:726-785 with kernel.evalPtr(x, y), x in the cell */
static int panel_boundary_pw(const nlo_problem *P, int c1, int b, int *perm1, int *perm2, int *perm, double *sv) {
    double s1[MAXV][2], s2[MAXV][2], ce1[2], ce2[2], vol2;
    simplex_of(P, c1, s1, ce1);
    facet_of(P, b, s2, ce2, &vol2);
    *sv = pw_svalue_simplices(P, ce1, ce2, s1, P->dim+1, s2, P->dim);
    nlo_problem Q = *P;
    Q.nclasses = 0;
    Q.bqo = pw_formula_boundary(P, *sv);
    return nlo_panel_boundary(&Q, c1, b, perm1, perm2, perm);
}

static void eval_boundary_pw(const nlo_problem *P, int c1, int b, int panel, double sv, const int *perm1, const int *perm2,
                             const int *perm, double *contrib, int64_t *nevals) {
    const int dim = P->dim, nV = dim+1, nF = dim, dpe = P->dpe;
    const int E = dpe*(dpe+1)/2;
    double s1[MAXV][2], s2[MAXV][2], ce[2], vol2, nrm[2] = {0., 0.};
    simplex_of(P, c1, s1, ce);
    facet_of(P, b, s2, ce, &vol2);
    if (dim == 2) {
        nrm[0] = s2[1][1]-s2[0][1];
        nrm[1] = s2[0][0]-s2[1][0];
        double v = 1./sqrt(nrm[0]*nrm[0]+nrm[1]*nrm[1]);
        nrm[0] *= v; nrm[1] *= v;
    }
    for (int k = 0; k < E; k++) contrib[k] = 0.;
    if (panel >= 1) {
        const int off = P->dist_off[panel], n = P->dist_off[panel+1]-off;
        const int foff = P->bfacet_off[panel], nf = P->bfacet_off[panel+1]-foff;
        const double *bary = P->dist_bary+3*off, *w = P->dist_w+off, *phi = P->dist_phi+(size_t)off*dpe;
        const double *fb = P->bfacet_bary+2*foff, *fw = P->bfacet_w+foff;
        const double vol = P->vol[c1]*vol2;
        double *temp = (double*)malloc(sizeof(double)*(size_t)n*nf);
        for (int k = 0; k < n; k++)
            for (int m = 0; m < nf; m++) {
                double x[2] = {0., 0.}, y[2] = {0., 0.}, wv[2], normW = 0., nw;
                for (int l = 0; l < dim; l++) {
                    for (int q = 0; q < nV; q++) x[l] += bary[3*k+q]*s1[q][l];
                    for (int q = 0; q < nF; q++) y[l] += fb[2*m+q]*s2[q][l];
                }
                if (dim == 1) nw = 1.;
                else {
                    for (int l = 0; l < dim; l++) { wv[l] = y[l]-x[l]; normW += wv[l]*wv[l]; }
                    normW = 1./sqrt(normW);
                    nw = nrm[0]*wv[0]*normW + nrm[1]*wv[1]*normW;
                }
                temp[k*nf+m] = (w[k]*fw[m])*nw*pw_kernel(P, x, y, 1);
            }
        *nevals += (int64_t)n*nf;
        int e = 0;
        for (int I = 0; I < dpe; I++)
            for (int J = I; J < dpe; J++) {
                double val = 0.;
                for (int k = 0; k < n; k++)
                    for (int m = 0; m < nf; m++) val += temp[k*nf+m]*phi[k*dpe+I]*phi[k*dpe+J];
                contrib[e++] = val*vol;
            }
        free(temp);
        return;
    }
    const int slot = -panel-1, key = pw_key(P->pw_bkeys, P->pw_nbkeys, sv);
    const int M = P->bsing_M[slot];
    const double *nodes = P->pw_bnodes[slot]+(size_t)key*(nV+nF)*M, *w = P->pw_bw[slot]+(size_t)key*M;
    const double *PHI = P->pw_bphi[slot]+(size_t)key*dpe*M;
    const double vol = dim == 2 ? P->bsing_fac*P->vol[c1]*vol2 : P->bsing_fac*P->vol[c1];
    double *temp = (double*)malloc(sizeof(double)*M);
    for (int m = 0; m < M; m++) {
        double x[2] = {0., 0.}, y[2] = {0., 0.}, wv[2] = {0., 0.}, normW = 0., nw = 1.;
        for (int j = 0; j < dim; j++) {
            for (int k = 0; k < nV; k++) x[j] += s1[perm1[k]][j]*nodes[(size_t)k*M+m];
            for (int k = 0; k < nF; k++) y[j] += s2[perm2[k]][j]*nodes[(size_t)(nV+k)*M+m];
            wv[j] = x[j]-y[j];
            normW += wv[j]*wv[j];
        }
        if (dim == 2) {
            double inv = 1./sqrt(normW);
            nw = nrm[0]*wv[0]*inv + nrm[1]*wv[1]*inv;
        }
        temp[m] = w[m]*nw*pw_kernel(P, x, y, 1);
    }
    *nevals += M;
    for (int I = 0; I < dpe; I++) {
        int i = perm[I];
        for (int J = I; J < dpe; J++) {
            int j = perm[J];
            int k = j < i ? dpe*j-(j*(j+1) >> 1)+i : dpe*i-(i*(i+1) >> 1)+j;
            double val = 0.;
            for (int m = 0; m < M; m++) val += temp[m]*PHI[(size_t)I*M+m]*PHI[(size_t)J*M+m];
            contrib[k] = val*vol;
        }
    }
    free(temp);
}

/* NA:1386-1448 with symmetricCells == symmetricLocalMatrix == False */
int nlo_get_dense_nonsym(const nlo_problem *P, double *A, int zero_exterior, int cell_start, int cell_end,
                         int64_t *counters, double *seconds, int store) {
    const int dpe = P->dpe, nV = P->dim+1, n2 = 2*dpe;
    const int64_t N = P->num_dofs;
    if (dpe > MAXDPE || nV > MAXV || (!P->pw_type && !P->nclasses)) return -1;
    const int piecewise = !P->pw_type;          /* order frozen per orientation of the pair: temp == temp2, the local matrix is the symmetric one */
    double contrib[4*MAXDPE*MAXDPE];
    int perm1[MAXV], perm2[MAXV], perm[2*MAXDPE], ld[2*MAXDPE];
    memset(counters, 0, sizeof(int64_t)*NLO_NUM_COUNTERS);
    double t0 = now_s();
    for (int c1 = cell_start; c1 < cell_end; c1++)
        for (int c2 = c1; c2 < P->nc; c2++) {
            counters[0]++;
            int skip = 1;
            for (int p = 0; p < dpe; p++) skip = skip && P->dofs[c1*dpe+p] < 0 && P->dofs[c2*dpe+p] < 0;
            if (skip) continue;
            for (int orient = 0; orient < (c1 == c2 ? 1 : 2); orient++) {
                const int a = orient ? c2 : c1, b = orient ? c1 : c2;     /* swapCells NA:1418 */
                for (int p = 0; p < dpe; p++) { ld[p] = P->dofs[a*dpe+p]; ld[dpe+p] = P->dofs[b*dpe+p]; }
                double sv, s1[MAXV][2], s2[MAXV][2], ce[2];
                const int panel = panel_nonsym(P, a, b, perm1, perm2, perm, &sv);
                if (panel >= 1 && (panel > P->qmax || P->dist_off[panel+1] == P->dist_off[panel])) return -(1000+panel);
                simplex_of(P, a, s1, ce);
                simplex_of(P, b, s2, ce);
                if (!orient || piecewise) {     /* piecewise: the two orientations may differ in order and rule, count both */
                    counters[1]++;
                    if (panel >= 1) counters[8+panel]++; else counters[8+NLO_MAX_ORDER+(-panel-1)]++;
                }
                if (piecewise) {
                    nlo_eval(P, a, b, panel, perm1, perm2, perm, contrib, &counters[2]);
                    if (store) scatter_elem_elem_sym(A, N, ld, n2, contrib, 1.);
                    continue;
                }
                if (panel >= 1) {
                    eval_distant_nonsym(P, s1, s2, P->vol[a]*P->vol[b], panel, contrib);
                    const int n = P->dist_off[panel+1]-P->dist_off[panel];
                    counters[2] += (int64_t)n*n;
                } else
                    eval_singular_nonsym(P, s1, s2, P->vol[a]*P->vol[b], -panel-1, pw_key(P->pw_keys, P->pw_nkeys, sv), perm1, perm2,
                                         perm, contrib, &counters[2]);
                if (store) scatter_elem_elem(A, N, ld, n2, contrib, 1.);
            }
        }
    double t1 = now_s();
    if (zero_exterior) {
        for (int c1 = cell_start; c1 < cell_end; c1++) {
            for (int p = 0; p < dpe; p++) ld[p] = P->dofs[c1*dpe+p];
            for (int b = 0; b < P->nb; b++) {
                double sv;
                if (piecewise) {
                    int panel = nlo_panel_boundary(P, c1, b, perm1, perm2, perm);
                    if (panel >= 1 && (panel > P->qmax || P->dist_off[panel+1] == P->dist_off[panel]
                                       || P->bfacet_off[panel+1] == P->bfacet_off[panel])) return -(2000+panel);
                    counters[3]++;
                    nlo_eval_boundary(P, c1, b, panel, perm1, perm2, perm, contrib, &counters[4]);
                    if (store) scatter_elem_elem_sym(A, N, ld, dpe, contrib, 1.);
                    continue;
                }
                int panel = panel_boundary_pw(P, c1, b, perm1, perm2, perm, &sv);
                if (panel >= 1 && (panel > P->qmax || P->dist_off[panel+1] == P->dist_off[panel]
                                   || P->bfacet_off[panel+1] == P->bfacet_off[panel])) return -(2000+panel);
                counters[3]++;
                eval_boundary_pw(P, c1, b, panel, sv, perm1, perm2, perm, contrib, &counters[4]);
                if (store) scatter_elem_elem_sym(A, N, ld, dpe, contrib, 1.);
            }
        }
    }
    double t2 = now_s();
    if (seconds) { seconds[0] = t1-t0; seconds[1] = t2-t1; }
    return 0;
}

/* CSR_LinearOperator.addToEntry / SSS_LinearOperator.addToEntry
 * (base/PyNucleus_base/CSR_LinearOperator_{SCALAR}.pxi:150-170, SSS_LinearOperator_{SCALAR}.pxi:104-130):
 * search the row, silently drop entries that are not in the pattern; SSS keeps I > J plus a diagonal vector. */
static void sparse_add(const int32_t *indptr, const int32_t *indices, double *data, double *diag, int I, int J, double v) {
    if (I < 0 || J < 0) return;                 /* boundary DoFs are skipped by every addToMatrix* (NA:152-253) */
    if (diag) {
        if (I == J) { diag[I] += v; return; }
        if (I < J) return;
    }
    int lo = indptr[I], hi = indptr[I+1];
    while (lo < hi) {
        int mid = (lo+hi) >> 1;
        if (indices[mid] < J) lo = mid+1; else hi = mid;
    }
    if (lo < indptr[I+1] && indices[lo] == J) data[lo] += v;
}

int nlo_assemble_pairs_masked(const nlo_problem *P, int np, const int32_t *pairs, const uint64_t *masks, const int32_t *indptr,
                              const int32_t *indices, double *data, double *diag, int64_t *counters) {
    const int dpe = P->dpe, n2 = 2*dpe;
    double contrib[MAXE];
    int perm1[MAXV], perm2[MAXV], perm[2*MAXDPE], ld[2*MAXDPE];
    /* variable orders: nlo_panel / nlo_eval take the class of the pair (evalParams at the two centres, NO:509-513); the interface
     * terms NA:1966-2156 are boundary items of the class problems (oracle.py assemble_clusters_variable) */
    counters[0] = counters[1] = counters[2] = 0;
    for (int t = 0; t < np; t++) {
        const int c1 = pairs[2*t], c2 = pairs[2*t+1];
        const uint64_t *mask = masks+4*(size_t)t;
        counters[0]++;
        int panel = nlo_panel(P, c1, c2, perm1, perm2, perm);
        if (panel == NLO_IGNORED) continue;
        if (panel >= 1 && (panel > P->qmax || P->dist_off[panel+1] == P->dist_off[panel])) return -(1000+panel);
        counters[1]++;
        int skip = 1;
        for (int p = 0; p < dpe; p++) { ld[p] = P->dofs[c1*dpe+p]; skip = skip && ld[p] < 0; }
        for (int p = 0; p < dpe; p++) { ld[dpe+p] = P->dofs[c2*dpe+p]; skip = skip && ld[dpe+p] < 0; }
        if (skip) continue;
        nlo_eval(P, c1, c2, panel, perm1, perm2, perm, contrib, &counters[2]);
        const double fac = c1 == c2 ? 1. : 2.;
        int k = 0;                                  /* NA:503-520 addToMatrixElemElemSymMasked */
        for (int p = 0; p < n2; p++) {
            const int I = ld[p];
            if ((mask[k >> 6] >> (k & 63)) & 1) sparse_add(indptr, indices, data, diag, I, I, fac*contrib[k]);
            k++;
            for (int q = p+1; q < n2; q++) {
                if ((mask[k >> 6] >> (k & 63)) & 1) {
                    const int J = ld[q];
                    sparse_add(indptr, indices, data, diag, I, J, fac*contrib[k]);
                    sparse_add(indptr, indices, data, diag, J, I, fac*contrib[k]);
                }
                k++;
            }
        }
    }
    return 0;
}

int nlo_assemble_boundary_masked(const nlo_problem *P, int ni, const int32_t *cells, const int32_t *facets, const uint32_t *masks,
                                 double fac, const int32_t *indptr, const int32_t *indices, double *data, double *diag) {
    const int dpe = P->dpe, dim = P->dim;
    double contrib[MAXE];
    int perm1[MAXV], perm2[MAXV], perm[2*MAXDPE];
    int64_t nevals = 0;
    nlo_problem Q = *P;
    Q.nb = 1;
    for (int t = 0; t < ni; t++) {
        const int c1 = cells[t];
        Q.bcells = facets+(size_t)t*dim;
        int panel;
        if (P->pw_type) {
            /* order per quadrature point: local_matrix_surface with the pointwise boundary kernel, s(x) at x in the cell; no
             * shift of the facet centre for orders of one variable (surfaceIntegralNeedsShift, NA:1966) */
            double sv;
            panel = panel_boundary_pw(&Q, c1, 0, perm1, perm2, perm, &sv);
            if (panel >= 1 && (panel > P->qmax || P->dist_off[panel+1] == P->dist_off[panel]
                               || P->bfacet_off[panel+1] == P->bfacet_off[panel])) return -(2000+panel);
            eval_boundary_pw(&Q, c1, 0, panel, sv, perm1, perm2, perm, contrib, &nevals);
        } else {
        panel = nlo_panel_boundary(&Q, c1, 0, perm1, perm2, perm);
        if (panel >= 1 && (panel > P->qmax || P->dist_off[panel+1] == P->dist_off[panel]
                           || P->bfacet_off[panel+1] == P->bfacet_off[panel])) return -(2000+panel);
        nlo_eval_boundary(&Q, c1, 0, panel, perm1, perm2, perm, contrib, &nevals);
        }
        const uint32_t mask = masks[t];
        int k = 0;                                  /* NA:534-546 addToMatrixElemSymMasked */
        for (int p = 0; p < dpe; p++) {
            const int I = P->dofs[c1*dpe+p];
            if ((mask >> k) & 1) sparse_add(indptr, indices, data, diag, I, I, fac*contrib[k]);
            k++;
            for (int q = p+1; q < dpe; q++) {
                if ((mask >> k) & 1) {
                    const int J = P->dofs[c1*dpe+q];
                    sparse_add(indptr, indices, data, diag, I, J, fac*contrib[k]);
                    sparse_add(indptr, indices, data, diag, J, I, fac*contrib[k]);
                }
                k++;
            }
        }
    }
    return 0;
}

/* NA:1812-1832 with symmetricCells == symmetricLocalMatrix == False (non-symmetric kernels): the masks hold ORDERED cell pairs --
 * buildMasksForClusters walks cellsUnion x cellsUnion (NA:322-349) -- with one bit per entry of the (2 dpe)^2 local matrix,
 * k = p (2 dpe) + q (getElemElemMask NA:425-440); every listed pair is evaluated in ITS orientation (setCell1 / setCell2, the
 * kernel parameters of that orientation) and scattered with fac = 1 by addToMatrixElemElemMasked (NA:520-532) into CSR.
 * Orders per quadrature point: the _nonsym local matrices; piecewise-constant non-symmetric orders: the symmetric local matrix
 * of the class of (label c1, label c2), stored in full (as nlo_get_dense_nonsym). */
int nlo_assemble_pairs_masked_nonsym(const nlo_problem *P, int np, const int32_t *pairs, const uint64_t *masks, const int32_t *indptr,
                                     const int32_t *indices, double *data, int64_t *counters) {
    const int dpe = P->dpe, n2 = 2*dpe;
    if (dpe > MAXDPE || (!P->pw_type && !P->nclasses)) return -1;
    const int piecewise = !P->pw_type;
    double contrib[4*MAXDPE*MAXDPE], csym[MAXE];
    int perm1[MAXV], perm2[MAXV], perm[2*MAXDPE], ld[2*MAXDPE];
    memset(counters, 0, sizeof(int64_t)*NLO_NUM_COUNTERS);
    for (int t = 0; t < np; t++) {
        const int a = pairs[2*t], b = pairs[2*t+1];
        const uint64_t *mask = masks+4*(size_t)t;
        counters[0]++;
        double sv, s1[MAXV][2], s2[MAXV][2], ce[2];
        const int panel = panel_nonsym(P, a, b, perm1, perm2, perm, &sv);
        if (panel >= 1 && (panel > P->qmax || P->dist_off[panel+1] == P->dist_off[panel])) return -(1000+panel);
        counters[1]++;
        int skip = 1;
        for (int p = 0; p < dpe; p++) { ld[p] = P->dofs[a*dpe+p]; ld[dpe+p] = P->dofs[b*dpe+p]; skip = skip && ld[p] < 0 && ld[dpe+p] < 0; }
        if (skip) continue;
        if (panel >= 1) counters[8+panel]++; else counters[8+NLO_MAX_ORDER+(-panel-1)]++;
        simplex_of(P, a, s1, ce);
        simplex_of(P, b, s2, ce);
        if (piecewise) {
            nlo_eval(P, a, b, panel, perm1, perm2, perm, csym, &counters[2]);
            int k = 0;
            for (int p = 0; p < n2; p++)
                for (int q = p; q < n2; q++) { contrib[p*n2+q] = csym[k]; contrib[q*n2+p] = csym[k]; k++; }
        } else if (panel >= 1) {
            eval_distant_nonsym(P, s1, s2, P->vol[a]*P->vol[b], panel, contrib);
            const int n = P->dist_off[panel+1]-P->dist_off[panel];
            counters[2] += (int64_t)n*n;
        } else
            eval_singular_nonsym(P, s1, s2, P->vol[a]*P->vol[b], -panel-1, pw_key(P->pw_keys, P->pw_nkeys, sv), perm1, perm2, perm,
                                 contrib, &counters[2]);
        int k = 0;
        for (int p = 0; p < n2; p++)
            for (int q = 0; q < n2; q++) {
                if ((mask[k >> 6] >> (k & 63)) & 1) sparse_add(indptr, indices, data, NULL, ld[p], ld[q], contrib[k]);
                k++;
            }
    }
    return 0;
}
