/*
 * bqp_oracle.c -- CPU ORACLE (test infrastructure, NOT product code) for the GENERIC constrained binary QP solver of the reference:
 *     min x'Ax + b'x   s.t.  Cx = d,  Ex <= f,  x in {0,1}^n
 * Restates, without Eigen, SEGcpp = Segmentation/Segmentation/cython/src/LPboxADMMsolver.cpp:
 *     ADMM_bqp                       SEGcpp:1384-1832  (one loop for the four problem types, SolverInstruction switches)
 *     the four wrappers              SEGcpp:1834-1853 (unconstrained), :1884-1912 (linear_eq), :1955-1978 (linear_ineq), :2023-2053 (both)
 *     hyper-parameter presets        SEGcpp:587-601 (eq), :603-617 (ineq), :620-634 (eq + ineq), :658-672 (unconstrained)
 *     matrix-expression product      SEGcpp:361-411    (result = (2A + (rho1+rho2) I) x ; result += rho3 C'(C x) ; result += rho4 E'(E x))
 *     PCG on the expression          SEGcpp:415-469    (Eigen's CG with Jacobi preconditioner)
 *     helpers                        SEGcpp:482-586    (std_dev, compute_std_obj, projections, compute_cost)
 * SparseMatrix is ROW-major in this flavour (SEGh:17): sparse*dense = per row, sum in ascending column order.  rho3 C' and rho4 E'
 * are SEPARATE scaled copies that are multiplied by learning_fact at every rho update (SEGcpp:1643,1648), so each stored entry
 * carries its own rounding history: this file keeps them as explicit value arrays.
 *
 * Reduction orders: 0 = Eigen 3.3.8 SSE2 redux; 1 = the HIP kernels' two-level tree (see lpbox_oracle.c), rows and columns summed
 * sequentially in ascending index order, sqrt instead of pow(v, 1/2).
 *
 * PARITY: nothing in the reference calls or tests this function from Python and it ships no outputs: parity with the reference
 * binary is UNPINNED.  Cross-checks in tests/: the unconstrained type against the segmentation oracle's legacy loop (same
 * arithmetic, SEGcpp:1200-1380), and the HIP path against this file bit for bit.
 */
#include "bqp_oracle.h"

#include <float.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

typedef struct { int rows, cols, nnz; int *ptr, *idx; double *val; } csr_t;

static void csr_free(csr_t *m) { free(m->ptr); free(m->idx); free(m->val); memset(m, 0, sizeof(*m)); }
static void csr_set(csr_t *m, int rows, int cols, const int *ptr, const int *idx, const double *val) {
    csr_free(m);
    const int nnz = ptr ? ptr[rows] : 0;
    m->rows = rows; m->cols = cols; m->nnz = nnz;
    m->ptr = (int *)calloc((size_t)rows + 1, sizeof(int));
    m->idx = (int *)malloc(sizeof(int) * (size_t)(nnz > 0 ? nnz : 1));
    m->val = (double *)malloc(sizeof(double) * (size_t)(nnz > 0 ? nnz : 1));
    if (ptr) memcpy(m->ptr, ptr, sizeof(int) * ((size_t)rows + 1));
    if (nnz) { memcpy(m->idx, idx, sizeof(int) * (size_t)nnz); memcpy(m->val, val, sizeof(double) * (size_t)nnz); }
}
/* transpose: rows of the result = columns of s, entries in ascending original row order (Eigen's transpose of a compressed matrix) */
static void csr_transpose(csr_t *t, const csr_t *s) {
    csr_free(t);
    t->rows = s->cols; t->cols = s->rows; t->nnz = s->nnz;
    t->ptr = (int *)calloc((size_t)t->rows + 1, sizeof(int));
    t->idx = (int *)malloc(sizeof(int) * (size_t)(s->nnz > 0 ? s->nnz : 1));
    t->val = (double *)malloc(sizeof(double) * (size_t)(s->nnz > 0 ? s->nnz : 1));
    for (int k = 0; k < s->nnz; k++) t->ptr[s->idx[k] + 1]++;
    for (int j = 0; j < t->rows; j++) t->ptr[j + 1] += t->ptr[j];
    int *cur = (int *)malloc(sizeof(int) * (size_t)(t->rows > 0 ? t->rows : 1));
    memcpy(cur, t->ptr, sizeof(int) * (size_t)t->rows);
    for (int i = 0; i < s->rows; i++)
        for (int k = s->ptr[i]; k < s->ptr[i + 1]; k++) { const int p = cur[s->idx[k]]++; t->idx[p] = i; t->val[p] = s->val[k]; }
    free(cur);
}
/* RowMajor sparse * dense (Eigen): res[i] = 0 + 1.0 * (sum_k val*v[col], ascending col) */
static void spmv_row(const csr_t *m, const double *v, double *res) {
    for (int i = 0; i < m->rows; i++) {
        double tmp = 0;
        for (int k = m->ptr[i]; k < m->ptr[i + 1]; k++) tmp += m->val[k] * v[m->idx[k]];
        double r = 0.0;
        r += 1.0 * tmp;
        res[i] = r;
    }
}

static double redux_sum_eigen(const double *a, int size) {      /* Eigen 3.3.8 redux, SSE2 packets of 2 (see lpbox_oracle.c) */
    if (size <= 0) return 0.0;
    const int P = 2, a2 = (size / (2 * P)) * (2 * P), a1 = (size / P) * P;
    double res;
    if (a1) {
        double p0a = a[0], p0b = a[1];
        if (a1 > P) {
            double p1a = a[2], p1b = a[3];
            for (int i = 2 * P; i < a2; i += 2 * P) { p0a = p0a + a[i]; p0b = p0b + a[i + 1]; p1a = p1a + a[i + 2]; p1b = p1b + a[i + 3]; }
            p0a = p0a + p1a; p0b = p0b + p1b;
            if (a1 > a2) { p0a = p0a + a[a2]; p0b = p0b + a[a2 + 1]; }
        }
        res = p0a + p0b;
        for (int i = a1; i < size; ++i) res = res + a[i];
    } else {
        res = a[0];
        for (int i = 1; i < size; ++i) res = res + a[i];
    }
    return res;
}
/* block tree of the HIP kernels: element e -> thread e % T, slots ascending; 64 lanes: l + l^32, then l^16, then inside a row of 16 */
static double block_tree(const double *full, int len, int T) {
    double local[1024];
    for (int t = 0; t < T; t++) local[t] = 0.0;
    for (int e = 0; e < len; e++) { int t = e % T; local[t] = local[t] + full[e]; }
    const int W = T / 64;
    double part[16];
    for (int w = 0; w < W; w++) {
        double *a = local + 64 * w;
        for (int i = 0; i < 32; i++) a[i] = a[i] + a[i + 32];
        for (int i = 0; i < 16; i++) a[i] = a[i] + a[i + 16];
        for (int s = 1; s < 16; s <<= 1) for (int i = 0; i < 16; i += 2 * s) a[i] = a[i] + a[i + s];
        part[w] = a[0];
    }
    for (int s = 1; s < W; s <<= 1) for (int i = 0; i < W; i += 2 * s) part[i] = part[i] + part[i + s];
    return part[0];
}

struct bqpo {
    int order_mode, T, CHUNK;
    int n, m, l, type;                       /* type bit 0: equality, bit 1: inequality (SEGh problem_t) */
    csr_t A, tm, C, Ct, r3Ct, E, Et, r4Et;   /* tm = 2A + (rho1+rho2) I ; r3Ct / r4Et = the scaled transposes */
    double *b, *d, *f, *x0;
    /* hyper-parameters */
    double stop_threshold, std_threshold, gamma_val0, gamma_factor, initial_rho, history_size, learning_fact, pcg_tol;
    int rho_change_step, max_iters, pcg_maxiters, projection_lp;
    /* state */
    double *x, *y1, *y2, *z1, *z2, *y3, *z3, *z4, *best, *tv, *tmm, *tl, *tmv, *pdiag, *invdiag, *Csq, *Esq, *cg, *full;
    double rho1, rho2, rho3, rho4, gamma_val, std_obj, cvg1, cvg2, cur_obj, best_bin_obj, obj_val;
    double *obj_list; int obj_n, obj_cap;
    int iters, stop, total_pcg, last_pcg;
    int *trace; int trace_n, trace_cap;
};

bqpo_t *bqpo_create(void) {
    bqpo_t *o = (bqpo_t *)calloc(1, sizeof(bqpo_t));
    o->T = 256; o->CHUNK = 512;
    bqpo_preset(o, 0);
    return o;
}
static void free_vecs(bqpo_t *o) {
    double **v[] = {&o->x, &o->y1, &o->y2, &o->z1, &o->z2, &o->y3, &o->z3, &o->z4, &o->best, &o->tv, &o->tmm, &o->tl, &o->tmv, &o->pdiag,
                    &o->invdiag, &o->Csq, &o->Esq, &o->cg, &o->full};
    for (size_t i = 0; i < sizeof(v) / sizeof(v[0]); i++) { free(*v[i]); *v[i] = NULL; }
}
void bqpo_destroy(bqpo_t *o) {
    if (!o) return;
    csr_free(&o->A); csr_free(&o->tm); csr_free(&o->C); csr_free(&o->Ct); csr_free(&o->r3Ct); csr_free(&o->E); csr_free(&o->Et); csr_free(&o->r4Et);
    free(o->b); free(o->d); free(o->f); free(o->x0); free_vecs(o); free(o->obj_list); free(o->trace);
    free(o);
}
void bqpo_set_order(bqpo_t *o, int mode, int T, int chunk) { o->order_mode = mode; if (T >= 64) o->T = T; if (chunk >= T) o->CHUNK = chunk; }

int bqpo_preset(bqpo_t *o, int type) {
    switch (type) {
    case 0:   /* ADMM_bqp_unconstrained_init SEGcpp:658-672 */
        o->std_threshold = 1e-6; o->gamma_val0 = 1.0; o->gamma_factor = 0.99; o->initial_rho = 5; o->learning_fact = 1 + 3.0 / 100;
        o->history_size = 5; o->rho_change_step = 5; o->stop_threshold = 1e-3; o->max_iters = (int)1e4; o->projection_lp = 2;
        o->pcg_tol = 1e-3; o->pcg_maxiters = (int)1e3; return 0;
    case 1:   /* ADMM_bqp_linear_eq_init :587-601 */
        o->stop_threshold = 1e-4; o->std_threshold = 1e-6; o->gamma_val0 = 1.6; o->gamma_factor = 0.95; o->rho_change_step = 5;
        o->max_iters = (int)5e3; o->initial_rho = 1; o->history_size = 3; o->learning_fact = 1 + 5.0 / 100; o->pcg_tol = 1e-4;
        o->pcg_maxiters = (int)1e3; o->projection_lp = 2; return 0;
    case 2:   /* ADMM_bqp_linear_ineq_init :603-617 */
    case 3:   /* ADMM_bqp_linear_eq_and_uneq_init :620-634 (same values) */
        o->stop_threshold = 1e-4; o->std_threshold = 1e-6; o->gamma_val0 = 1.6; o->gamma_factor = 0.95; o->rho_change_step = 5;
        o->max_iters = (int)1e4; o->initial_rho = 25; o->history_size = 3; o->learning_fact = 1 + 1.0 / 100; o->pcg_tol = 1e-4;
        o->pcg_maxiters = (int)1e3; o->projection_lp = 2; return 0;
    }
    return -1;
}
int bqpo_set_params(bqpo_t *o, const double *p) {
    o->stop_threshold = p[0]; o->std_threshold = p[1]; o->gamma_val0 = p[2]; o->gamma_factor = p[3]; o->rho_change_step = (int)p[4];
    o->max_iters = (int)p[5]; o->initial_rho = p[6]; o->history_size = p[7]; o->learning_fact = p[8]; o->pcg_tol = p[9];
    o->pcg_maxiters = (int)p[10];
    return 0;
}

static int check_csr(int rows, int cols, const int *ptr, const int *idx) {
    if (ptr[0] != 0) return -1;
    for (int i = 0; i < rows; i++) {
        if (ptr[i + 1] < ptr[i]) return -1;
        for (int k = ptr[i]; k < ptr[i + 1]; k++) {
            if (idx[k] < 0 || idx[k] >= cols) return -1;
            if (k > ptr[i] && idx[k] <= idx[k - 1]) return -1;
        }
    }
    return 0;
}

int bqpo_set_problem(bqpo_t *o, int n, const int *Ap, const int *Ai, const double *Av, const double *b, const double *x0,
                     int m, const int *Cp, const int *Ci, const double *Cv, const double *d,
                     int l, const int *Ep, const int *Ei, const double *Ev, const double *f) {
    if (n <= 0 || !Ap || !b || !x0 || check_csr(n, n, Ap, Ai)) return -1;
    for (int i = 0; i < n; i++) {           /* `.diagonal() +=` on a compressed sparse matrix needs every diagonal entry to exist (SEGcpp:1478) */
        int found = 0;
        for (int k = Ap[i]; k < Ap[i + 1]; k++) if (Ai[k] == i) found = 1;
        if (!found) return -2;
    }
    if (m > 0 && (!Cp || !d || check_csr(m, n, Cp, Ci))) return -1;
    if (l > 0 && (!Ep || !f || check_csr(l, n, Ep, Ei))) return -1;
    o->n = n; o->m = m > 0 ? m : 0; o->l = l > 0 ? l : 0;
    o->type = (o->m ? 1 : 0) | (o->l ? 2 : 0);
    csr_set(&o->A, n, n, Ap, Ai, Av);
    csr_set(&o->C, o->m, n, o->m ? Cp : NULL, Ci, Cv);
    csr_set(&o->E, o->l, n, o->l ? Ep : NULL, Ei, Ev);
    free(o->b); free(o->d); free(o->f); free(o->x0);
    o->b = (double *)malloc(sizeof(double) * (size_t)n); memcpy(o->b, b, sizeof(double) * (size_t)n);
    o->x0 = (double *)malloc(sizeof(double) * (size_t)n); memcpy(o->x0, x0, sizeof(double) * (size_t)n);
    o->d = (double *)malloc(sizeof(double) * (size_t)(o->m + 1)); if (o->m) memcpy(o->d, d, sizeof(double) * (size_t)o->m);
    o->f = (double *)malloc(sizeof(double) * (size_t)(o->l + 1)); if (o->l) memcpy(o->f, f, sizeof(double) * (size_t)o->l);
    return 0;
}

static double reduce(bqpo_t *o, const double *a) {
    if (o->order_mode == 0) return redux_sum_eigen(a, o->n);
    const int N = o->n, CH = o->CHUNK, G = (N + CH - 1) / CH;
    double *part = o->full;
    for (int g = 0; g < G; g++) {
        int len = N - g * CH; if (len > CH) len = CH;
        part[g] = block_tree(a + (size_t)g * CH, len, o->T);
    }
    return block_tree(part, G, o->T);
}
static double dot(bqpo_t *o, const double *a, const double *b) {
    double *p = o->cg + (size_t)4 * o->n;
    for (int i = 0; i < o->n; i++) p[i] = a[i] * b[i];
    return reduce(o, p);
}
static double pow_half(bqpo_t *o, double v) { return o->order_mode ? sqrt(v) : pow(v, 1.0 / 2); }

static double compute_cost(bqpo_t *o, const double *x) {      /* SEGcpp:560-572: x.dot(A x) + b.dot(x) */
    spmv_row(&o->A, x, o->tmm);
    const double v1 = dot(o, x, o->tmm), v2 = dot(o, o->b, x);
    return v1 + v2;
}

/* calculate_mat_expr_multiplication SEGcpp:361-411 */
static void expr_mul(bqpo_t *o, const double *x, double *res) {
    spmv_row(&o->tm, x, res);
    if (o->type & 1) {
        spmv_row(&o->C, x, o->tl);                 /* temp = C x */
        spmv_row(&o->r3Ct, o->tl, o->tmv);         /* temp = (rho3 C') temp */
        for (int i = 0; i < o->n; i++) res[i] += o->tmv[i];
    }
    if (o->type & 2) {
        spmv_row(&o->E, x, o->tl);
        spmv_row(&o->r4Et, o->tl, o->tmv);
        for (int i = 0; i < o->n; i++) res[i] += o->tmv[i];
    }
}

static int pcg(bqpo_t *o, const double *rhs, double *x) {     /* SEGcpp:415-469 */
    const int n = o->n;
    double *residual = o->cg, *p = o->cg + n, *z = o->cg + 2 * (size_t)n, *tmp = o->cg + 3 * (size_t)n;
    expr_mul(o, x, tmp);
    for (int i = 0; i < n; i++) residual[i] = rhs[i] - tmp[i];
    const double rhsNorm2 = dot(o, rhs, rhs);
    if (rhsNorm2 == 0) { for (int i = 0; i < n; i++) x[i] = 0; return 0; }
    double threshold = o->pcg_tol * o->pcg_tol * rhsNorm2;
    if (threshold < DBL_MIN) threshold = DBL_MIN;
    double residualNorm2 = dot(o, residual, residual);
    if (residualNorm2 < threshold) return 0;
    for (int i = 0; i < n; i++) p[i] = o->invdiag[i] * residual[i];
    double absNew = dot(o, residual, p);
    int i = 0;
    while (i < o->pcg_maxiters) {
        expr_mul(o, p, tmp);
        const double alpha = absNew / dot(o, p, tmp);
        for (int k = 0; k < n; k++) x[k] += alpha * p[k];
        for (int k = 0; k < n; k++) residual[k] -= alpha * tmp[k];
        residualNorm2 = dot(o, residual, residual);
        if (residualNorm2 < threshold) { i++; break; }
        for (int k = 0; k < n; k++) z[k] = o->invdiag[k] * residual[k];
        const double absOld = absNew;
        absNew = dot(o, residual, z);
        const double beta = absNew / absOld;
        for (int k = 0; k < n; k++) p[k] = z[k] + beta * p[k];
        i++;
    }
    return i;
}

static double std_obj_of(bqpo_t *o) {          /* compute_std_obj + std_dev SEGcpp:482-507, :574-585 */
    size_t s = (size_t)o->obj_n, hs = (size_t)o->history_size;
    size_t begin = s <= hs ? 0 : s - hs, size = s - begin;
    double mean = 0, sd = 0;
    for (size_t i = begin; i < s; i++) mean += o->obj_list[i];
    mean /= size;
    for (size_t i = 0; i < size; i++) sd += (o->obj_list[begin + i] - mean) * (o->obj_list[begin + i] - mean);
    sd /= size - 1;
    const double r = sd == 0 ? 0 : pow_half(o, sd);
    return r / fabs(o->obj_list[s - 1]);
}

static double *vec(size_t n) { return (double *)calloc(n > 0 ? n : 1, sizeof(double)); }

int bqpo_solve(bqpo_t *o) {                    /* ADMM_bqp SEGcpp:1384-1832 */
    const int n = o->n, m = o->m, l = o->l;
    if (n <= 0) return -1;
    const int eq = o->type & 1, ineq = o->type & 2;
    free_vecs(o);
    o->x = vec(n); o->y1 = vec(n); o->y2 = vec(n); o->z1 = vec(n); o->z2 = vec(n); o->best = vec(n); o->tv = vec(n); o->tmm = vec(n);
    o->tmv = vec(n); o->pdiag = vec(n); o->invdiag = vec(n); o->Csq = vec(n); o->Esq = vec(n); o->cg = vec((size_t)5 * n);
    o->y3 = vec(l); o->z3 = vec(m); o->z4 = vec(l); o->tl = vec((size_t)(m > l ? m : l));
    o->full = vec((size_t)n / 64 + 1024);
    o->obj_n = 0; o->trace_n = 0; o->total_pcg = 0; o->stop = 0;
    memcpy(o->x, o->x0, sizeof(double) * (size_t)n);                                   /* :1432 */
    double rho1 = o->initial_rho, rho2 = rho1, rho3 = rho1, rho4 = rho1;
    double prev_rho1 = rho1, prev_rho2 = rho2, prev_rho3 = rho3, prev_rho4 = rho4, rcr = 0;
    o->gamma_val = o->gamma_val0; o->std_obj = 1;
    int rhoUpdated = 1;
    if (eq) {                                                                            /* :1464-1470 */
        csr_transpose(&o->Ct, &o->C);
        csr_transpose(&o->r3Ct, &o->C);
        for (int k = 0; k < o->r3Ct.nnz; k++) o->r3Ct.val[k] = rho3 * o->r3Ct.val[k];
    }
    if (ineq) {                                                                          /* :1473-1479 */
        csr_transpose(&o->Et, &o->E);
        csr_transpose(&o->r4Et, &o->E);
        for (int k = 0; k < o->r4Et.nnz; k++) o->r4Et.val[k] = rho4 * o->r4Et.val[k];
    }
    csr_set(&o->tm, n, n, o->A.ptr, o->A.idx, o->A.val);                                 /* :1482-1483 */
    for (int k = 0; k < o->tm.nnz; k++) o->tm.val[k] = 2 * o->tm.val[k];
    for (int i = 0; i < n; i++)
        for (int k = o->tm.ptr[i]; k < o->tm.ptr[i + 1]; k++) if (o->tm.idx[k] == i) { o->tm.val[k] += rho1 + rho2; o->pdiag[i] = o->tm.val[k]; }
    if (eq) {                                                                            /* :1513-1526 */
        for (int j = 0; j < n; j++) {
            double s = 0;
            for (int k = o->Ct.ptr[j]; k < o->Ct.ptr[j + 1]; k++) if (o->Ct.val[k] != 0.0) s += o->Ct.val[k] * o->Ct.val[k];
            o->Csq[j] = s;
        }
        for (int j = 0; j < n; j++) o->pdiag[j] += rho3 * o->Csq[j];
    }
    if (ineq) {                                                                          /* :1535-1548 */
        for (int j = 0; j < n; j++) {
            double s = 0;
            for (int k = o->Et.ptr[j]; k < o->Et.ptr[j + 1]; k++) if (o->Et.val[k] != 0.0) s += o->Et.val[k] * o->Et.val[k];
            o->Esq[j] = s;
        }
        for (int j = 0; j < n; j++) o->pdiag[j] += rho4 * o->Esq[j];
    }
    memcpy(o->y1, o->x, sizeof(double) * (size_t)n); memcpy(o->y2, o->x, sizeof(double) * (size_t)n);   /* :1551-1552 */
    if (ineq) { spmv_row(&o->E, o->x, o->tl); for (int i = 0; i < l; i++) o->y3[i] = o->f[i] - o->tl[i]; }   /* :1554 */
    memcpy(o->best, o->x, sizeof(double) * (size_t)n);
    o->best_bin_obj = compute_cost(o, o->x);                                             /* :1560 */
    int iter;
    for (iter = 0; iter < o->max_iters; iter++) {
        double *tv = o->tv;
        for (int i = 0; i < n; i++) tv[i] = o->x[i] + o->z1[i] / rho1;                   /* :1598-1601 */
        for (int i = 0; i < n; i++) o->y1[i] = tv[i] > 1 ? 1 : (tv[i] < 0 ? 0 : tv[i]);
        for (int i = 0; i < n; i++) tv[i] = o->x[i] + o->z2[i] / rho2;                   /* :1603-1606 */
        for (int i = 0; i < n; i++) o->y2[i] = tv[i] - 0.5;
        {
            const double nrm = sqrt(dot(o, o->y2, o->y2));
            const double c1 = pow((double)n, 1.0 / o->projection_lp), c2 = 2 * nrm;
            for (int i = 0; i < n; i++) o->y2[i] = o->y2[i] * c1 / c2 + 0.5;
        }
        if (ineq) {                                                                      /* :1609-1613 */
            spmv_row(&o->E, o->x, o->tl);
            for (int i = 0; i < l; i++) { const double v = o->f[i] - o->tl[i] - o->z4[i] / rho4; o->y3[i] = v < 0 ? 0 : v; }
        }
        if (iter != 0 && rhoUpdated) {                                                   /* :1619-1650 */
            const double inc = rcr * (prev_rho1 + prev_rho2);
            for (int i = 0; i < n; i++)
                for (int k = o->tm.ptr[i]; k < o->tm.ptr[i + 1]; k++) if (o->tm.idx[k] == i) o->tm.val[k] += inc;
            if (o->type != 0) for (int i = 0; i < n; i++) o->pdiag[i] += inc;
            if (eq) {
                for (int i = 0; i < n; i++) o->pdiag[i] += rcr * prev_rho3 * o->Csq[i];
                for (int k = 0; k < o->r3Ct.nnz; k++) o->r3Ct.val[k] = o->learning_fact * o->r3Ct.val[k];
            }
            if (ineq) {
                for (int i = 0; i < n; i++) o->pdiag[i] += rcr * prev_rho4 * o->Esq[i];
                for (int k = 0; k < o->r4Et.nnz; k++) o->r4Et.val[k] = o->learning_fact * o->r4Et.val[k];
            }
        }
        for (int i = 0; i < n; i++) tv[i] = (rho1 * o->y1[i] + rho2 * o->y2[i]) - ((o->b[i] + o->z1[i]) + o->z2[i]);   /* :1656 ff */
        if (eq) {
            spmv_row(&o->r3Ct, o->d, o->tmv); for (int i = 0; i < n; i++) tv[i] += o->tmv[i];
            spmv_row(&o->Ct, o->z3, o->tmv);  for (int i = 0; i < n; i++) tv[i] -= o->tmv[i];
        }
        if (ineq) {
            for (int i = 0; i < l; i++) o->tl[i] = o->f[i] - o->y3[i];
            spmv_row(&o->r4Et, o->tl, o->tmv); for (int i = 0; i < n; i++) tv[i] += o->tmv[i];
            spmv_row(&o->Et, o->z4, o->tmv);   for (int i = 0; i < n; i++) tv[i] -= o->tmv[i];
        }
        if (rhoUpdated) {                                                                /* :1711-1718 */
            for (int i = 0; i < n; i++) {
                double dg = o->pdiag[i];
                if (o->type == 0) for (int k = o->tm.ptr[i]; k < o->tm.ptr[i + 1]; k++) if (o->tm.idx[k] == i) dg = o->tm.val[k];
                o->invdiag[i] = dg != 0.0 ? 1.0 / dg : 1.0;
            }
            rhoUpdated = 0;
        }
        memcpy(o->x, o->y1, sizeof(double) * (size_t)n);                                 /* :1721 */
        const int k_it = pcg(o, tv, o->x);
        o->last_pcg = k_it; o->total_pcg += k_it;
        if (o->trace_n == o->trace_cap) { o->trace_cap = o->trace_cap ? 2 * o->trace_cap : 1024; o->trace = (int *)realloc(o->trace, sizeof(int) * (size_t)o->trace_cap); }
        o->trace[o->trace_n++] = k_it;
        {
            const double g1 = o->gamma_val * rho1, g2 = o->gamma_val * rho2;             /* :1733-1734 */
            for (int i = 0; i < n; i++) o->z1[i] = o->z1[i] + g1 * (o->x[i] - o->y1[i]);
            for (int i = 0; i < n; i++) o->z2[i] = o->z2[i] + g2 * (o->x[i] - o->y2[i]);
        }
        if (eq) {                                                                        /* :1736 */
            const double g3 = o->gamma_val * rho3;
            spmv_row(&o->C, o->x, o->tl);
            for (int i = 0; i < m; i++) o->z3[i] = o->z3[i] + g3 * (o->tl[i] - o->d[i]);
        }
        if (ineq) {                                                                      /* :1739 */
            const double g4 = o->gamma_val * rho4;
            spmv_row(&o->E, o->x, o->tl);
            for (int i = 0; i < l; i++) o->z4[i] = o->z4[i] + g4 * ((o->tl[i] + o->y3[i]) - o->f[i]);
        }
        {
            const double xn = sqrt(dot(o, o->x, o->x));
            const double t0 = xn < 2.2204e-16 ? 2.2204e-16 : xn;
            for (int i = 0; i < n; i++) tv[i] = o->x[i] - o->y1[i];
            o->cvg1 = sqrt(dot(o, tv, tv)) / t0;
            for (int i = 0; i < n; i++) tv[i] = o->x[i] - o->y2[i];
            o->cvg2 = sqrt(dot(o, tv, tv)) / t0;
            if (o->cvg1 <= o->stop_threshold && o->cvg2 <= o->stop_threshold) { o->stop = 1; break; }   /* :1745 */
        }
        if ((iter + 1) % o->rho_change_step == 0) {                                      /* :1753-1770 */
            prev_rho1 = rho1; prev_rho2 = rho2;
            rho1 = o->learning_fact * rho1; rho2 = o->learning_fact * rho2;
            if (eq) { prev_rho3 = rho3; rho3 = o->learning_fact * rho3; }
            if (ineq) { prev_rho4 = rho4; rho4 = o->learning_fact * rho4; }
            const double g = o->gamma_val * o->gamma_factor;
            o->gamma_val = g < 1.0 ? 1.0 : g;
            rhoUpdated = 1; rcr = o->learning_fact - 1.0;
        }
        o->obj_val = compute_cost(o, o->x);                                              /* :1772 */
        if (o->obj_n == o->obj_cap) { o->obj_cap = o->obj_cap ? 2 * o->obj_cap : 1024; o->obj_list = (double *)realloc(o->obj_list, sizeof(double) * (size_t)o->obj_cap); }
        o->obj_list[o->obj_n++] = o->obj_val;
        if ((double)o->obj_n >= o->history_size) o->std_obj = std_obj_of(o);
        if (o->std_obj <= o->std_threshold) { o->stop = 2; break; }                      /* :1777 */
        for (int i = 0; i < n; i++) tv[i] = o->x[i] >= 0.5 ? 1.0 : 0.0;                  /* :1786-1793 */
        o->cur_obj = compute_cost(o, tv);
        if (o->best_bin_obj >= o->cur_obj) { o->best_bin_obj = o->cur_obj; memcpy(o->best, o->x, sizeof(double) * (size_t)n); }
    }
    o->iters = iter;
    o->rho1 = rho1; o->rho2 = rho2; o->rho3 = rho3; o->rho4 = rho4;
    return iter;
}

int bqpo_get_vec(const bqpo_t *o, const char *name, double *out, int cap) {
    const double *s = NULL; int len = o->n;
    if (!strcmp(name, "x")) s = o->x; else if (!strcmp(name, "y1")) s = o->y1; else if (!strcmp(name, "y2")) s = o->y2;
    else if (!strcmp(name, "z1")) s = o->z1; else if (!strcmp(name, "z2")) s = o->z2; else if (!strcmp(name, "best_sol")) s = o->best;
    else if (!strcmp(name, "y3")) { s = o->y3; len = o->l; } else if (!strcmp(name, "z4")) { s = o->z4; len = o->l; }
    else if (!strcmp(name, "z3")) { s = o->z3; len = o->m; }
    if (!s) return -1;
    if (len > cap) return -len;
    memcpy(out, s, sizeof(double) * (size_t)len);
    return len;
}
double bqpo_get_scalar(const bqpo_t *o, const char *name) {
    const struct { const char *n; double v; } tab[] = {
        {"rho1", o->rho1}, {"rho3", o->rho3}, {"rho4", o->rho4}, {"gamma", o->gamma_val}, {"std_obj", o->std_obj}, {"cvg1", o->cvg1},
        {"cvg2", o->cvg2}, {"cur_obj", o->cur_obj}, {"best_bin_obj", o->best_bin_obj}, {"obj_val", o->obj_val},
        {"iters", (double)o->iters}, {"stop", (double)o->stop}, {"total_pcg", (double)o->total_pcg}, {"last_pcg", (double)o->last_pcg},
    };
    for (size_t k = 0; k < sizeof(tab) / sizeof(tab[0]); k++) if (!strcmp(name, tab[k].n)) return tab[k].v;
    return NAN;
}
int bqpo_get_trace(const bqpo_t *o, int *out, int cap) { int c = o->trace_n < cap ? o->trace_n : cap; for (int i = 0; i < c; i++) out[i] = o->trace[i]; return c; }
