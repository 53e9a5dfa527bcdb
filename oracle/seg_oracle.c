/*
 * seg_oracle.c -- CPU ORACLE (test infrastructure, NOT product code) for the SEGMENTATION flavour of the solver.
 *
 * Restates, without Eigen / OpenCV, the unconstrained-BQP path of the reference:
 *   SEGcpp = Segmentation/Segmentation/cython/src/LPboxADMMsolver.cpp      (SEGh = the .h next to it)
 *     cost builders      SEGcpp:46-248   (vectorize, get_unary_cost, generate_pixel_pairs, get_binary_cost, get_A_b_from_cost)
 *     PCG                SEGcpp:272-342  (explicit-matrix overload; verbatim Eigen CG + Jacobi)
 *     init               SEGcpp:658-810  (ADMM_bqp_unconstrained_init, minus image decode/resize = OpenCV, see seg_set_image)
 *     l2f loop           SEGcpp:917-1195 (ADMM_bqp_unconstrained_l2f)
 *     legacy loop        SEGcpp:1200-1380 (ADMM_bqp_unconstrained_legacy)
 *     getters            SEGcpp:839-914
 * In this flavour SparseMatrix is ROW-major (SEGh:17): sparse*dense = per row, tmp = sum_k val_k * v[col_k] in ascending
 * column order, res[i] = 0 + 1.0 * tmp (Eigen SparseDenseProduct.h, RowMajor branch).  Reductions: see lpbox_oracle.c.
 *
 * PARITY: the reference ships no outputs for this path either; OpenCV's JPEG decode + resize are replaced by an explicit
 * grayscale array handed in by the caller (the image decode is outside the solver).  Parity with the binary is UNPINNED.
 */
#include "seg_oracle.h"

#include <float.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#ifndef M_PI
#define M_PI 3.14159265358979323846
#endif

typedef struct { int rows, cols, nnz; int *ptr, *idx; double *val; } csr_t;   /* row-major, columns ascending in a row */

static void csr_free(csr_t *m) { free(m->ptr); free(m->idx); free(m->val); memset(m, 0, sizeof(*m)); }
static void csr_alloc(csr_t *m, int rows, int cols, int nnz) {
    m->rows = rows; m->cols = cols; m->nnz = nnz;
    m->ptr = (int *)calloc((size_t)rows + 1, sizeof(int));
    m->idx = (int *)malloc(sizeof(int) * (size_t)(nnz > 0 ? nnz : 1));
    m->val = (double *)malloc(sizeof(double) * (size_t)(nnz > 0 ? nnz : 1));
}
static void csr_copy(csr_t *d, const csr_t *s) {
    csr_free(d); csr_alloc(d, s->rows, s->cols, s->nnz);
    memcpy(d->ptr, s->ptr, sizeof(int) * ((size_t)s->rows + 1));
    if (s->nnz) { memcpy(d->idx, s->idx, sizeof(int) * (size_t)s->nnz); memcpy(d->val, s->val, sizeof(double) * (size_t)s->nnz); }
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

/* Eigen 3.3.8 redux (SSE2) -- same restatement as lpbox_oracle.c */
static double redux_sum_eigen(const double *a, int size) {
    if (size <= 0) return 0.0;
    const int P = 2;
    const int alignedEnd2 = (size / (2 * P)) * (2 * P), alignedEnd = (size / P) * P;
    double res;
    if (alignedEnd) {
        double p0a = a[0], p0b = a[1];
        if (alignedEnd > P) {
            double p1a = a[2], p1b = a[3];
            for (int i = 2 * P; i < alignedEnd2; i += 2 * P) { p0a += a[i]; p0b += a[i + 1]; p1a += a[i + 2]; p1b += a[i + 3]; }
            p0a = p0a + p1a; p0b = p0b + p1b;
            if (alignedEnd > alignedEnd2) { p0a = p0a + a[alignedEnd2]; p0b = p0b + a[alignedEnd2 + 1]; }
        }
        res = p0a + p0b;
        for (int i = alignedEnd; i < size; ++i) res = res + a[i];
    } else {
        res = a[0];
        for (int i = 1; i < size; ++i) res = res + a[i];
    }
    return res;
}

/* block tree of the HIP kernels (T threads, element e -> thread e % T, slots ascending, 64-lane butterfly, wave tree) */
static double block_tree(const double *full, int len, int T) {
    double local[1024];
    for (int t = 0; t < T; t++) local[t] = 0.0;
    for (int e = 0; e < len; e++) { int t = e % T; local[t] = local[t] + full[e]; }
    int W = T / 64;
    double part[16];
    for (int w = 0; w < W; w++) {
        double *a = local + 64 * w;
        for (int i = 0; i < 32; i++) a[i] = a[i] + a[i + 32];              /* halves first, then row pairs, then inside a row of 16 lanes */
        for (int i = 0; i < 16; i++) a[i] = a[i] + a[i + 16];
        for (int s = 1; s < 16; s <<= 1) for (int i = 0; i < 16; i += 2 * s) a[i] = a[i] + a[i + s];
        part[w] = a[0];
    }
    for (int s = 1; s < W; s <<= 1) for (int i = 0; i < W; i += 2 * s) part[i] = part[i] + part[i + s];
    return part[0];
}

struct sego {
    int print_info, node, problem, verbose;
    int order_mode, T, CHUNK;         /* GPU order: workgroups of T threads own CHUNK consecutive ORIGINAL indices */
    /* image -> problem */
    int scaled_row, scaled_col;
    csr_t A, orgA, tm;                /* *A_ptr, *org_A, temp_mat */
    double *b, *orgb, c;
    int n, org_n, x_len;
    /* hyper-parameters SEGcpp:659-672 */
    double std_threshold, gamma_val, gamma_factor, initial_rho, learning_fact, history_size, rel_tol, stop_threshold,
        pcg_tol, projection_lp;
    int rho_change_step, max_iters, pcg_maxiters;
    double *x, *y1, *y2, *z1, *z2, *tv, *tmm, *cg, *invdiag; int invdiag_len;
    double rho1, rho2, prev_rho1, prev_rho2, rho_change_ratio, cur_obj, std_obj, cvg1, cvg2, best_bin_obj, obj_val;
    int rhoUpdated;
    double *obj_list; int obj_n, obj_cap;
    double *x_iters; int xi_rows, xi_cols;
    int *left_idx, *ret_idx_prev; double *ret_val_prev; int ret_prev_len;
    int fix_sum, iter_legacy_p1, last_stop, last_pcg;
    long total_pcg, total_outer;
    int *trace; int trace_n, trace_cap;
    double *full; int inited, has_problem;
};

sego_t *sego_create(int print_info, int numNodes, int problem) {   /* SEGcpp:650-655 */
    sego_t *o = (sego_t *)calloc(1, sizeof(sego_t));
    o->print_info = print_info; o->node = numNodes; o->problem = problem;
    o->rhoUpdated = 1; o->std_obj = 1; o->T = 256; o->CHUNK = 512;
    return o;
}

static void free_state(sego_t *o) {
    free(o->x); free(o->y1); free(o->y2); free(o->z1); free(o->z2); free(o->tv); free(o->tmm); free(o->cg); free(o->invdiag);
    free(o->obj_list); free(o->x_iters); free(o->left_idx); free(o->ret_idx_prev); free(o->ret_val_prev); free(o->trace); free(o->full);
    o->x = o->y1 = o->y2 = o->z1 = o->z2 = o->tv = o->tmm = o->cg = o->invdiag = o->obj_list = o->x_iters = o->full = NULL;
    o->left_idx = o->ret_idx_prev = o->trace = NULL; o->ret_val_prev = NULL;
}

void sego_destroy(sego_t *o) {
    if (!o) return;
    free_state(o);
    csr_free(&o->A); csr_free(&o->orgA); csr_free(&o->tm);
    free(o->b); free(o->orgb); free(o);
}

void sego_set_order(sego_t *o, int mode, int T, int chunk) { o->order_mode = mode; if (T >= 64) o->T = T; if (chunk >= T) o->CHUNK = chunk; }
void sego_set_verbose(sego_t *o, int v) { o->verbose = v; }

/* sum over the live variables; compact element i sits at original index left_idx[i] */
static double reduce(sego_t *o, const double *a) {
    if (o->order_mode == 0) return redux_sum_eigen(a, o->n);
    const int N = o->org_n, CH = o->CHUNK;
    for (int i = 0; i < N; i++) o->full[i] = 0.0;
    for (int i = 0; i < o->n; i++) o->full[o->left_idx[i]] = a[i];
    const int G = (N + CH - 1) / CH;
    double *part = o->full + N;                      /* G <= 8192 scratch behind the vector */
    for (int g = 0; g < G; g++) {
        int len = N - g * CH; if (len > CH) len = CH;
        part[g] = block_tree(o->full + (size_t)g * CH, len, o->T);
    }
    return block_tree(part, G, o->T);                /* second level: the G workgroup partials through the same tree */
}

static double dot(sego_t *o, const double *a, const double *b) {
    double *p = o->cg + (size_t)5 * o->org_n;
    for (int i = 0; i < o->n; i++) p[i] = a[i] * b[i];
    return reduce(o, p);
}

static double pow_half(sego_t *o, double v) { return o->order_mode ? sqrt(v) : pow(v, 1.0 / 2); }

/* ------------------------------------------------------------------------------------------ */
/* cost builders SEGcpp:46-248.  img = grayscale pixel values (0..255), ROW-major rows x cols (what cv2eigen yields).  */
/* ------------------------------------------------------------------------------------------ */
int sego_build_costs(int rows, int cols, const double *img, int *n_out, int **rowptr_out, int **colidx_out,
                     double **val_out, double **b_out, double *c_out) {
    const int n = rows * cols;
    double *nodes = (double *)malloc(sizeof(double) * (size_t)n);      /* vectorize (:46-53): column-major flatten of I/263 */
    for (int j = 0; j < cols; j++) for (int i = 0; i < rows; i++) nodes[j * rows + i] = img[i * cols + j] / 263.0;   /* :727 */
    /* get_unary_cost :55-81 with sigma=0.1, b=0.6, f1=f2=0.2 (:736-739), then round (:747) */
    const double sigma = 0.1, bb = 0.6, f1 = 0.2, f2 = 0.2;
    const double cc = log(2.0 * M_PI) / 2.0 + log(sigma);
    double *U1 = (double *)malloc(sizeof(double) * (size_t)n), *U2 = (double *)malloc(sizeof(double) * (size_t)n);
    for (int p = 0; p < n; p++) {
        double alpha_b = pow(nodes[p] - bb, 2.0) / (2 * sigma * sigma) + cc;
        double aa = exp(-pow(nodes[p] - f1, 2.0) / (2 * sigma * sigma)) + exp(-pow(nodes[p] - f2, 2) / (2 * sigma * sigma));
        double alpha_f = -log(aa + DBL_EPSILON) + cc + log(2.0);
        U1[p] = round(alpha_b); U2[p] = round(alpha_f);
    }
    /* get_binary_cost :173-224: sigma = sample std of the image vector (NOT variance) */
    double mean = 0;
    {
        /* Eigen .mean() = sum()/size with the vectorised redux order; .square().sum() likewise */
        mean = redux_sum_eigen(nodes, n) / n;
    }
    double *sq = (double *)malloc(sizeof(double) * (size_t)n);
    for (int p = 0; p < n; p++) sq[p] = (nodes[p] - mean) * (nodes[p] - mean);
    const double sig = sqrt(redux_sum_eigen(sq, n) / (n - 1));
    free(sq);
    /* generate_pixel_pairs :144-171 (k = 1): pairs (i*ncols+j, (i+a)*ncols+(j+b)), a != b, visited i, j, a, b ascending.
     * Inside one row of the matrix that is ascending column order already except for nothing: (a,b) lexicographic gives
     * increasing (i+a)*ncols + (j+b). */
    int *rowptr = (int *)calloc((size_t)n + 1, sizeof(int));
    int cap = 7 * n;
    int *colidx = (int *)malloc(sizeof(int) * (size_t)cap);
    double *val = (double *)malloc(sizeof(double) * (size_t)cap);
    int k = 0;
    for (int i = 0; i < rows; i++) for (int j = 0; j < cols; j++) {
        const int r = i * cols + j;
        rowptr[r] = k;
        int diag_done = 0;
        double We = 0;                                   /* row sum of W in the order Eigen's A*ones visits the row */
        const int first = k;
        for (int a = -1; a <= 1; a++) for (int b2 = -1; b2 <= 1; b2++) {
            if (a == 0 && b2 == 0) {                     /* the explicit zero diagonal (:213-219) sits between the neighbours */
                colidx[k] = r; val[k] = 0.0; k++; diag_done = 1;
                continue;
            }
            if (a != b2 && i + a >= 0 && i + a < rows && j + b2 >= 0 && j + b2 < cols) {
                const int q = (i + a) * cols + (j + b2);
                /* intensities are fetched as if the pair index were column-major (:192-193) = the vectorised image */
                const double d = pow(nodes[r] - nodes[q], 2.0) / sig;
                const double w = round(3 * exp(-d));     /* :196-206, :209 */
                colidx[k] = q; val[k] = w; k++;
            }
        }
        (void)diag_done;
        /* get_A_b_from_cost :226-248: A = -W; We = -A*ones (row-major product: tmp += val*1 over the row, ascending col);
         * A.diagonal() += We; A = 2A; then A_ptr = A/2 (:755-756) */
        for (int e = first; e < k; e++) { double a_e = -val[e]; We += a_e * 1.0; }
        We = -(0.0 + 1.0 * We);
        for (int e = first; e < k; e++) {
            double a_e = -val[e];
            if (colidx[e] == r) a_e += We;
            a_e = 2 * a_e;
            val[e] = a_e / 2;
        }
    }
    rowptr[n] = k;
    double *b = (double *)malloc(sizeof(double) * (size_t)n);
    for (int p = 0; p < n; p++) b[p] = U2[p] - U1[p];                   /* :232 */
    *c_out = redux_sum_eigen(U1, n);                                    /* :245 c = U1.array().sum() */
    free(nodes); free(U1); free(U2);
    *n_out = n; *rowptr_out = rowptr; *colidx_out = colidx; *val_out = val; *b_out = b;
    return k;
}

void sego_free_arrays(int *a, int *b, double *c, double *d) { free(a); free(b); free(c); free(d); }

/* cv::resize(src, dst, Size(), scale, scale) with INTER_LINEAR on 8-bit data, restated from OpenCV 4.4's published
 * algorithm (imgproc/resize.cpp): dsize = round(src*scale); source coordinate fx = (dx+0.5)/scale - 0.5 clamped, weights
 * quantised to 11 bits, two-pass fixed-point interpolation with the (v + 2) >> 2 / (.. + 2^15) >> 16 rounding of
 * VResizeLinear<uchar,int,short>.  Unpinned against a real OpenCV build. */
int sego_resize_linear_u8(int rows, int cols, const unsigned char *src, double scale, int *orows, int *ocols, unsigned char *dst) {
    const int dr = (int)lround(rows * scale), dc = (int)lround(cols * scale);
    *orows = dr; *ocols = dc;
    if (!dst) return 0;
    const double inv_x = 1.0 / scale, inv_y = 1.0 / scale;
    const int ONE = 2048;
    int *xofs = (int *)malloc(sizeof(int) * (size_t)dc), *yofs = (int *)malloc(sizeof(int) * (size_t)dr);
    short *ax = (short *)malloc(sizeof(short) * 2 * (size_t)dc), *ay = (short *)malloc(sizeof(short) * 2 * (size_t)dr);
    for (int dx = 0; dx < dc; dx++) {
        float fx = (float)((dx + 0.5) * inv_x - 0.5);
        int sx = (int)floorf(fx); fx -= sx;
        if (sx < 0) { fx = 0; sx = 0; }
        if (sx >= cols - 1) { fx = 0; sx = cols - 1; }
        xofs[dx] = sx;
        ax[2 * dx] = (short)lrintf((1.f - fx) * ONE); ax[2 * dx + 1] = (short)lrintf(fx * ONE);
    }
    for (int dy = 0; dy < dr; dy++) {
        float fy = (float)((dy + 0.5) * inv_y - 0.5);
        int sy = (int)floorf(fy); fy -= sy;
        if (sy < 0) { fy = 0; sy = 0; }
        if (sy >= rows - 1) { fy = 0; sy = rows - 1; }
        yofs[dy] = sy;
        ay[2 * dy] = (short)lrintf((1.f - fy) * ONE); ay[2 * dy + 1] = (short)lrintf(fy * ONE);
    }
    for (int dy = 0; dy < dr; dy++) {
        const int sy0 = yofs[dy], sy1 = sy0 + 1 < rows ? sy0 + 1 : sy0;
        for (int dx = 0; dx < dc; dx++) {
            const int sx0 = xofs[dx], sx1 = sx0 + 1 < cols ? sx0 + 1 : sx0;
            const int r0 = src[sy0 * cols + sx0] * ax[2 * dx] + src[sy0 * cols + sx1] * ax[2 * dx + 1];
            const int r1 = src[sy1 * cols + sx0] * ax[2 * dx] + src[sy1 * cols + sx1] * ax[2 * dx + 1];
            const int v = (((ay[2 * dy] * (r0 >> 4)) >> 16) + ((ay[2 * dy + 1] * (r1 >> 4)) >> 16) + 2) >> 2;
            dst[dy * dc + dx] = (unsigned char)(v < 0 ? 0 : (v > 255 ? 255 : v));
        }
    }
    free(xofs); free(yofs); free(ax); free(ay);
    return 0;
}

/* problem input: what ADMM_bqp_unconstrained_init holds after get_A_b_from_cost: A_ptr = _A/2 (row-major), b, c */
int sego_set_problem(sego_t *o, int n, int nnz, const int *rowptr, const int *colidx, const double *vals, const double *b, double c,
                     int scaled_row, int scaled_col) {
    if (n <= 0 || nnz < 0) return -1;
    csr_free(&o->A); csr_alloc(&o->A, n, n, nnz);
    memcpy(o->A.ptr, rowptr, sizeof(int) * ((size_t)n + 1));
    memcpy(o->A.idx, colidx, sizeof(int) * (size_t)nnz);
    memcpy(o->A.val, vals, sizeof(double) * (size_t)nnz);
    csr_copy(&o->orgA, &o->A);
    free(o->b); free(o->orgb);
    o->b = (double *)malloc(sizeof(double) * (size_t)n); o->orgb = (double *)malloc(sizeof(double) * (size_t)n);
    memcpy(o->b, b, sizeof(double) * (size_t)n); memcpy(o->orgb, b, sizeof(double) * (size_t)n);
    o->c = c; o->n = n; o->scaled_row = scaled_row; o->scaled_col = scaled_col;
    o->has_problem = 1; o->inited = 0;
    return 0;
}

/* temp_mat = 2*A; diagonal += rho1+rho2 (:784-786 / :1054-1057) */
static void build_temp_mat(sego_t *o) {
    csr_copy(&o->tm, &o->A);
    for (int k = 0; k < o->tm.nnz; k++) o->tm.val[k] = 2 * o->A.val[k];
    for (int i = 0; i < o->tm.rows; i++)
        for (int k = o->tm.ptr[i]; k < o->tm.ptr[i + 1]; k++) if (o->tm.idx[k] == i) o->tm.val[k] += o->rho1 + o->rho2;
}

/* compute_cost (:568-572): x'(A x) + b'x */
static double compute_cost(sego_t *o, const double *x) {
    spmv_row(&o->A, x, o->tmm);
    double val = dot(o, x, o->tmm);
    double val2 = dot(o, o->b, x);
    return val + val2;
}

int sego_init(sego_t *o) {   /* ADMM_bqp_unconstrained_init SEGcpp:658-810 (image part: sego_build_costs + sego_set_problem) */
    if (!o->has_problem) return -1;
    o->std_threshold = 1e-6; o->gamma_val = 1.0; o->gamma_factor = 0.99; o->initial_rho = 5;
    o->learning_fact = 1 + 3.0 / 100; o->history_size = 5; o->rho_change_step = 5; o->rel_tol = 1e-5;
    o->stop_threshold = 1e-3; o->max_iters = (int)1e4; o->projection_lp = 2; o->pcg_tol = 1e-3; o->pcg_maxiters = (int)1e3;
    const int n = o->A.cols;
    o->n = n; o->org_n = n; o->x_len = n;
    free_state(o);
    size_t nn = (size_t)n;
    o->x = (double *)calloc(nn, sizeof(double)); o->y1 = (double *)calloc(nn, sizeof(double)); o->y2 = (double *)calloc(nn, sizeof(double));
    o->z1 = (double *)calloc(nn, sizeof(double)); o->z2 = (double *)calloc(nn, sizeof(double)); o->tv = (double *)calloc(nn, sizeof(double));
    o->tmm = (double *)calloc(nn, sizeof(double)); o->cg = (double *)calloc(6 * nn, sizeof(double)); o->invdiag = (double *)calloc(nn, sizeof(double));
    o->full = (double *)calloc(nn + 16384, sizeof(double));
    o->left_idx = (int *)malloc(sizeof(int) * nn);
    for (int i = 0; i < n; i++) o->left_idx[i] = i;
    o->rho1 = o->rho2 = o->initial_rho; o->prev_rho1 = o->rho1; o->prev_rho2 = o->rho2;
    build_temp_mat(o);
    o->best_bin_obj = compute_cost(o, o->x);                                   /* :792 */
    o->fix_sum = 0; o->ret_prev_len = 0;
    o->rhoUpdated = 1; o->std_obj = 1; o->cur_obj = 0; o->obj_n = 0;
    o->inited = 1;
    return 1;
}

static int pcg(sego_t *o, const double *rhs, double *x, int *iters) {          /* SEGcpp:272-342 */
    const int n = o->n;
    double *residual = o->cg, *p = o->cg + n, *z = o->cg + 2 * (size_t)n, *tmp = o->cg + 3 * (size_t)n;
    spmv_row(&o->tm, x, tmp);
    for (int i = 0; i < n; i++) residual[i] = rhs[i] - tmp[i];
    double rhsNorm2 = dot(o, rhs, rhs);
    if (rhsNorm2 == 0) { for (int i = 0; i < n; i++) x[i] = 0; *iters = 0; return 1; }
    double threshold = o->pcg_tol * o->pcg_tol * rhsNorm2;
    if (threshold < DBL_MIN) threshold = DBL_MIN;
    double residualNorm2 = dot(o, residual, residual);
    if (residualNorm2 < threshold) { *iters = 0; return 1; }
    for (int i = 0; i < n; i++) p[i] = o->invdiag[i] * residual[i];
    double absNew = dot(o, residual, p);
    int i = 0;
    while (i < o->pcg_maxiters) {
        spmv_row(&o->tm, p, tmp);
        double alpha = absNew / dot(o, p, tmp);
        for (int k = 0; k < n; k++) x[k] += alpha * p[k];
        for (int k = 0; k < n; k++) residual[k] -= alpha * tmp[k];
        residualNorm2 = dot(o, residual, residual);
        if (residualNorm2 < threshold) { i++; break; }
        for (int k = 0; k < n; k++) z[k] = o->invdiag[k] * residual[k];
        double absOld = absNew;
        absNew = dot(o, residual, z);
        double beta = absNew / absOld;
        for (int k = 0; k < n; k++) p[k] = z[k] + beta * p[k];
        i++;
    }
    *iters = i;
    return 1;
}

static void obj_push(sego_t *o, double v) {
    if (o->obj_n == o->obj_cap) { o->obj_cap = o->obj_cap ? 2 * o->obj_cap : 1024; o->obj_list = (double *)realloc(o->obj_list, sizeof(double) * (size_t)o->obj_cap); }
    o->obj_list[o->obj_n++] = v;
}
static void trace_push(sego_t *o, int v) {
    if (o->trace_n == o->trace_cap) { o->trace_cap = o->trace_cap ? 2 * o->trace_cap : 1024; o->trace = (int *)realloc(o->trace, sizeof(int) * (size_t)o->trace_cap); }
    o->trace[o->trace_n++] = v;
}

static double std_obj_of(sego_t *o) {        /* compute_std_obj + std_dev (same code as the LP flavour, SEGcpp:482-533) */
    size_t s = (size_t)o->obj_n, hs = (size_t)o->history_size;
    size_t begin = s <= hs ? 0 : s - hs, end = s, size = end - begin;
    double mean = 0, sd = 0;
    for (size_t i = begin; i < end; i++) mean += o->obj_list[i];
    mean /= size;
    for (size_t i = 0; i < size; i++) sd += (o->obj_list[begin + i] - mean) * (o->obj_list[begin + i] - mean);
    sd /= size - 1;
    double r = sd == 0 ? 0 : pow_half(o, sd);
    return r / fabs(o->obj_list[s - 1]);
}

/* one iteration shared by legacy (:1221-1356) and l2f (:1067-1177); returns 1 on break */
static int iteration(sego_t *o, int iter, int l2f, int *ret, int *cc) {
    const int n = o->n;
    double *tv = o->tv;
    for (int i = 0; i < n; i++) tv[i] = o->x[i] + o->z1[i] / o->rho1;
    for (int i = 0; i < n; i++) o->y1[i] = tv[i] > 1 ? 1 : (tv[i] < 0 ? 0 : tv[i]);
    for (int i = 0; i < n; i++) tv[i] = o->x[i] + o->z2[i] / o->rho2;
    for (int i = 0; i < n; i++) o->y2[i] = tv[i] - 0.5;
    {
        double nrm = sqrt(dot(o, o->y2, o->y2));
        double c1 = pow((double)n, 1.0 / (int)o->projection_lp), c2 = 2 * nrm;
        for (int i = 0; i < n; i++) o->y2[i] = o->y2[i] * c1 / c2 + 0.5;
    }
    if (iter != 0 && o->rhoUpdated) {                                       /* :1085-1088 / :1240-1243 */
        double inc = (o->prev_rho1 + o->prev_rho2) * o->rho_change_ratio;
        for (int i = 0; i < o->tm.rows; i++)
            for (int k = o->tm.ptr[i]; k < o->tm.ptr[i + 1]; k++) if (o->tm.idx[k] == i) o->tm.val[k] += inc;
    }
    for (int i = 0; i < n; i++) tv[i] = (o->rho1 * o->y1[i] + o->rho2 * o->y2[i]) - ((o->b[i] + o->z1[i]) + o->z2[i]);   /* :1091 */
    if (o->rhoUpdated || o->invdiag_len != n) {                             /* :1098-1101 (stale length = UB in the reference -> recompute) */
        for (int i = 0; i < n; i++) {
            double d = 0; int found = 0;
            for (int k = o->tm.ptr[i]; k < o->tm.ptr[i + 1]; k++) if (o->tm.idx[k] == i) { d = o->tm.val[k]; found = 1; }
            o->invdiag[i] = (found && d != 0.0) ? 1.0 / d : 1.0;
        }
        o->invdiag_len = n; o->rhoUpdated = 0;
    }
    for (int i = 0; i < n; i++) o->x[i] = o->y1[i];                         /* :1104 */
    int k_it = o->pcg_maxiters;
    pcg(o, tv, o->x, &k_it);
    o->last_pcg = k_it; o->total_pcg += k_it; o->total_outer++; trace_push(o, k_it);
    if (l2f) { for (int i = 0; i < n; i++) o->x_iters[(size_t)(*cc) * (size_t)o->xi_rows + i] = o->x[i]; (*cc)++; }   /* :1113-1116 */
    {
        double g1 = o->gamma_val * o->rho1, g2 = o->gamma_val * o->rho2;
        for (int i = 0; i < n; i++) o->z1[i] = o->z1[i] + g1 * (o->x[i] - o->y1[i]);
        for (int i = 0; i < n; i++) o->z2[i] = o->z2[i] + g2 * (o->x[i] - o->y2[i]);
    }
    {
        double xn = sqrt(dot(o, o->x, o->x));
        double t0 = xn < 2.2204e-16 ? 2.2204e-16 : xn;
        for (int i = 0; i < n; i++) tv[i] = o->x[i] - o->y1[i];
        o->cvg1 = sqrt(dot(o, tv, tv)) / t0;
        for (int i = 0; i < n; i++) tv[i] = o->x[i] - o->y2[i];
        o->cvg2 = sqrt(dot(o, tv, tv)) / t0;
        if (o->cvg1 <= o->stop_threshold && o->cvg2 <= o->stop_threshold) {  /* :1127 / :1282 */
            if (l2f) *ret = 1;
            o->last_stop = 1;
            if (o->verbose) printf("Terminate by condition xyy. iter: %d, stop_threshold: %.6f\n", iter, o->cvg1 > o->cvg2 ? o->cvg1 : o->cvg2);
            return 1;
        }
    }
    if ((iter + 1) % o->rho_change_step == 0) {                             /* :1137-1145 */
        o->prev_rho1 = o->rho1; o->prev_rho2 = o->rho2;
        o->rho1 = o->learning_fact * o->rho1; o->rho2 = o->learning_fact * o->rho2;
        double g = o->gamma_val * o->gamma_factor;
        o->gamma_val = g < 1.0 ? 1.0 : g;
        o->rhoUpdated = 1; o->rho_change_ratio = o->learning_fact - 1.0;
    }
    o->obj_val = compute_cost(o, o->x);                                     /* :1148 */
    obj_push(o, o->obj_val);
    if ((double)o->obj_n >= o->history_size) o->std_obj = std_obj_of(o);
    if (o->std_obj <= o->std_threshold) {
        if (l2f) *ret = 1;
        o->last_stop = 2;
        if (o->verbose) printf("Terminate by condition obj_std. iter: %d, std_threshold: %.6f\n", iter, o->std_obj);
        return 1;
    }
    for (int i = 0; i < n; i++) tv[i] = o->x[i] >= 0.5 ? 1.0 : 0.0;         /* :1164-1173 */
    o->cur_obj = compute_cost(o, tv);
    if (o->best_bin_obj >= o->cur_obj) o->best_bin_obj = o->cur_obj;
    return 0;
}

int sego_legacy(sego_t *o) {                                                /* SEGcpp:1200-1380 */
    if (!o->inited) return -1;
    int ret = 0, cc = 0, iter;
    o->trace_n = 0; o->last_stop = 0;
    for (iter = 0; iter < o->max_iters; iter++) if (iteration(o, iter, 0, &ret, &cc)) break;
    o->iter_legacy_p1 = iter + 1;
    for (int i = 0; i < o->n; i++) o->tv[i] = o->x[i] >= 0.5 ? 1.0 : 0.0;   /* :1371-1372 */
    o->cur_obj = compute_cost(o, o->tv);
    return (int)(o->cur_obj + o->c);                                        /* :1379 */
}

int sego_l2f(sego_t *o, int iter_start, int iter_end, const double *vec, int fix_num) {   /* SEGcpp:917-1195 */
    if (!o->inited) return -1;
    int ret = 0, cc = 0, n = o->n;
    o->trace_n = 0; o->last_stop = 0;
    if (fix_num < 0 || fix_num > n) return -2;
    if (iter_end - iter_start > 10) return -3;                              /* x_iters has 10 columns (:924) */
    if (fix_num) { int cnt = 0; for (int i = 0; i < n; i++) if (vec[i] == 1 || vec[i] == 0) cnt++; if (cnt != fix_num) return -4; }
    free(o->x_iters);
    o->xi_rows = n - fix_num; o->xi_cols = 10;
    o->x_iters = (double *)calloc((size_t)(o->xi_rows > 0 ? o->xi_rows : 1) * 10, sizeof(double));
    if (fix_num != 0) {                                                     /* :927-1062 */
        o->fix_sum += fix_num;
        const int nk = n - fix_num;
        int *det_idx = (int *)malloc(sizeof(int) * (size_t)n);
        int *fix_idx = (int *)malloc(sizeof(int) * (size_t)fix_num), *non_fix_idx = (int *)malloc(sizeof(int) * (size_t)(nk > 0 ? nk : 1));
        double *x2 = (double *)malloc(sizeof(double) * (size_t)fix_num);
        int j = 0, k = 0;
        for (int i = 0; i < n; i++) {
            if (vec[i] == 1) { fix_idx[j] = i; x2[j] = 1; det_idx[i] = j; j++; }
            else if (vec[i] == 0) { fix_idx[j] = i; x2[j] = 0; det_idx[i] = j; j++; }
            else { non_fix_idx[k] = i; det_idx[i] = k; k++; }
        }
        /* Ma (kept x kept), Mb (kept x fixed): rows of kept variables, columns relabelled (:944-1015); setFromTriplets keeps
         * ascending column order inside a row because det_idx is monotone on each class */
        csr_t Ma, Mb; memset(&Ma, 0, sizeof(Ma)); memset(&Mb, 0, sizeof(Mb));
        int na = 0, nb = 0;
        for (int q = 0; q < nk; q++) { int i = non_fix_idx[q]; for (int e = o->A.ptr[i]; e < o->A.ptr[i + 1]; e++) { if (vec[o->A.idx[e]] == -1) na++; else nb++; } }
        csr_alloc(&Ma, nk, nk, na); csr_alloc(&Mb, nk, fix_num, nb);
        na = nb = 0;
        for (int q = 0; q < nk; q++) {
            int i = non_fix_idx[q];
            Ma.ptr[q] = na; Mb.ptr[q] = nb;
            for (int e = o->A.ptr[i]; e < o->A.ptr[i + 1]; e++) {
                int col = o->A.idx[e];
                if (vec[col] == -1) { Ma.idx[na] = det_idx[col]; Ma.val[na] = o->A.val[e]; na++; }
                else { Mb.idx[nb] = det_idx[col]; Mb.val[nb] = o->A.val[e]; nb++; }
            }
        }
        Ma.ptr[nk] = na; Mb.ptr[nk] = nb;
        int new_len = o->ret_prev_len + fix_num;                            /* :1017-1026 */
        o->ret_idx_prev = (int *)realloc(o->ret_idx_prev, sizeof(int) * (size_t)new_len);
        o->ret_val_prev = (double *)realloc(o->ret_val_prev, sizeof(double) * (size_t)new_len);
        for (int q = 0; q < fix_num; q++) { o->ret_idx_prev[o->ret_prev_len + q] = o->left_idx[fix_idx[q]]; o->ret_val_prev[o->ret_prev_len + q] = x2[q]; }
        o->ret_prev_len = new_len;
        for (int q = 0; q < nk; q++) o->left_idx[q] = o->left_idx[non_fix_idx[q]];
        if (nk == 0) { ret = 1; o->n = 0; iter_end = iter_start; o->last_stop = 4; }   /* :1028-1032 */
        else {
            for (int q = 0; q < nk; q++) {                                  /* :1034-1041 */
                int s = non_fix_idx[q];
                o->x[q] = o->x[s]; o->y1[q] = o->y1[s]; o->y2[q] = o->y2[s]; o->z1[q] = o->z1[s]; o->z2[q] = o->z2[s];
                o->tv[q] = o->b[s];
            }
            o->n = nk; o->x_len = nk;
            double *t = (double *)malloc(sizeof(double) * (size_t)nk);
            spmv_row(&Mb, x2, t);                                           /* :1051 */
            for (int q = 0; q < nk; q++) o->b[q] = 2 * t[q] + o->tv[q];     /* :1052 b = 2*Mb*x2 + b1 */
            free(t);
            csr_copy(&o->A, &Ma);                                           /* :1048-1049 */
            build_temp_mat(o);                                              /* :1054-1057 */
        }
        csr_free(&Ma); csr_free(&Mb);
        free(det_idx); free(fix_idx); free(non_fix_idx); free(x2);
    }
    for (int iter = iter_start; iter < iter_end; iter++) if (iteration(o, iter, 1, &ret, &cc)) break;
    return ret;
}

int sego_get_n(const sego_t *o) { return o->n; }
int sego_get_org_n(const sego_t *o) { return o->org_n; }
int sego_get_x_iters_rows(const sego_t *o) { return o->x_iters ? o->xi_rows : 0; }
int sego_get_x_iters(const sego_t *o, int ws, double *out) {
    if (!o->x_iters || ws < 0 || ws > o->xi_cols) return -1;
    for (int i = 0; i < o->xi_rows; i++) for (int j = 0; j < ws; j++) out[(size_t)i * ws + j] = o->x_iters[(size_t)j * o->xi_rows + i];
    return o->xi_rows;
}
int sego_get_x_sol(const sego_t *o, double *out) {                          /* SEGcpp:895-914 */
    for (int q = 0; q < o->ret_prev_len; q++) out[o->ret_idx_prev[q]] = o->ret_val_prev[q];
    if (o->n != 0) for (int q = 0; q < o->n; q++) out[o->left_idx[q]] = o->x[q] >= 0.5 ? 1.0 : 0.0;
    return o->org_n;
}
double sego_get_final_obj(sego_t *o) {                                      /* SEGcpp:868-893: energy on the ORIGINAL A, b plus c */
    double *xx = (double *)calloc((size_t)o->org_n, sizeof(double)), *t = (double *)calloc((size_t)o->org_n, sizeof(double));
    sego_get_x_sol(o, xx);
    spmv_row(&o->orgA, xx, t);
    /* x.dot(A*x) and b.dot(x) over the full-length vectors */
    int n_save = o->n; int *li = o->left_idx;
    int *ident = (int *)malloc(sizeof(int) * (size_t)o->org_n);
    for (int i = 0; i < o->org_n; i++) ident[i] = i;
    o->n = o->org_n; o->left_idx = ident;
    double *save_cg = o->cg; double *tmp_cg = (double *)calloc(6 * (size_t)o->org_n, sizeof(double)); o->cg = tmp_cg;
    double val = dot(o, xx, t), val2 = dot(o, o->orgb, xx);
    o->cg = save_cg; free(tmp_cg);
    o->n = n_save; o->left_idx = li; free(ident);
    free(xx); free(t);
    return (val + val2) + o->c;
}
double sego_get_c(const sego_t *o) { return o->c; }
int sego_last_stop(const sego_t *o) { return o->last_stop; }
int sego_legacy_iter_plus1(const sego_t *o) { return o->iter_legacy_p1; }
long sego_total_pcg(const sego_t *o) { return o->total_pcg; }
long sego_total_outer(const sego_t *o) { return o->total_outer; }
int sego_get_trace(const sego_t *o, int *out, int cap) { int c = o->trace_n < cap ? o->trace_n : cap; for (int i = 0; i < c; i++) out[i] = o->trace[i]; return c; }
int sego_get_vec(const sego_t *o, const char *name, double *out, int cap) {
    const double *s = NULL; int len = o->n;
    if (!strcmp(name, "x")) { s = o->x; len = o->x_len; } else if (!strcmp(name, "z1")) s = o->z1; else if (!strcmp(name, "z2")) s = o->z2;
    else if (!strcmp(name, "b")) s = o->b; else if (!strcmp(name, "y1")) s = o->y1; else if (!strcmp(name, "y2")) s = o->y2;
    else if (!strcmp(name, "left_idx")) { if (len > cap) return -len; for (int i = 0; i < len; i++) out[i] = o->left_idx[i]; return len; }
    else return -1;
    if (len > cap) return -len;
    memcpy(out, s, sizeof(double) * (size_t)len);
    return len;
}
double sego_get_scalar(const sego_t *o, const char *name) {
    const struct { const char *n; double v; } tab[] = {
        {"rho1", o->rho1}, {"gamma", o->gamma_val}, {"cur_obj", o->cur_obj}, {"std_obj", o->std_obj}, {"cvg1", o->cvg1},
        {"cvg2", o->cvg2}, {"obj_val", o->obj_val}, {"best_bin_obj", o->best_bin_obj},
    };
    for (size_t k = 0; k < sizeof(tab) / sizeof(tab[0]); k++) if (!strcmp(name, tab[k].n)) return tab[k].v;
    return NAN;
}
