/*
 * lpbox_oracle.h -- CPU ORACLE (test infrastructure, NOT product code).
 *
 * A from-scratch, single-threaded, plain-C restatement of the Lp-Box ADMM inner
 * solver of SCLBD/Accelerated-Lpbox-ADMM, LP flavour:
 *   LinerProgramming/LinearProgramming/cython_solver/LPboxADMMsolver.cpp  ("LPcpp")
 *   LinerProgramming/LinearProgramming/cython_solver/LPboxADMMsolver.h    ("LPh")
 * Every function cites the LPcpp lines it follows.  The third-party arithmetic
 * the reference takes from Eigen 3.3.8 (un-vendored; README.md:24) -- sparse x dense
 * products, dot / squaredNorm / norm reductions, DiagonalPreconditioner -- is
 * restated from Eigen's published algorithm (see lpbox_oracle.c).
 *
 * PARITY PINNING: the reference ships no tests, golden vectors or result files for
 * this path and cannot be built here (Eigen absent), so
 * this oracle is pinned by (i) an independent numpy restatement (oracle/lpbox_numpy.py),
 * (ii) analytic known-answer tests, (iii) an exact MILP solve bound.  With respect to
 * the reference *binary* parity is UNPINNED -- see DESIGN.md.
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may use
 * anything in oracle/.  The product path never links or calls it.
 */
#ifndef LPBOX_ORACLE_H
#define LPBOX_ORACLE_H

#ifdef __cplusplus
extern "C" {
#endif

typedef struct lpo lpo_t;

/* Reduction association used for dot / norm reductions. */
enum {
    LPO_ORDER_EIGEN = 0, /* Eigen 3.3.8 SSE2 linear-vectorised redux (what the reference binary would do) */
    LPO_ORDER_GPU   = 1  /* the fixed tree the HIP kernels use (threads T, original index positions)   */
};

/* LPcpp:477-480 LPboxADMMsolver(int print_fix_info) */
lpo_t *lpo_create(int print_info);
void   lpo_destroy(lpo_t *o);

/* Select the reduction order (default LPO_ORDER_EIGEN). T = workgroup threads of the GPU order. */
void lpo_set_order(lpo_t *o, int mode, int T);
/* GPU order only: the kernels store variable j at position pos_of_var[j] (lpbox_get_layout) and the reduction tree is
 * defined over positions.  NULL / never called = identity. */
void lpo_set_positions(lpo_t *o, const int *pos_of_var, int n, int npos);
/* GPU order only: lanes_of_row[i] in {1,2,4,8} lanes share the sum of row i of E (lpbox_get_row_split). */
void lpo_set_row_split(lpo_t *o, const int *lanes_of_row, int l);
/* GPU order of the column sums (E^T w): own[n] leading entries by the own lane, help4[4n] chunk sizes by quad lane (include/lpbox_hip.h lpbox_get_col_split) */
void lpo_set_col_split(lpo_t *o, const int *own, const int *help4, int n);
/* GPU order of the LARGE-instance kernels: reductions are two-level (a block tree over every `chunk` consecutive positions, then
 * the same tree over the chunk partials).  0 = single workgroup (default). */
void lpo_set_chunk(lpo_t *o, int chunk);
/* GPU order of the variable-sharded run: `ranks` contiguous blocks of variables; per-rank sums added in rank order (needs lpo_set_chunk) */
void lpo_set_ranks(lpo_t *o, int ranks);
/* mirror of the large-instance kernels' opt-in comm-lean PCG (no reference counterpart): p.Mp = dI (p.p) + r4Et (q.q), q = E p */
void lpo_set_pcg_lean(lpo_t *o, int on, int row_chunk);
/* 0 = the reference's Jacobi-PCG x-update (default); 1 = the HIP kernels' opt-in DIRECT x-update (Woodbury with a dense l x l inverse;
 * NOT the reference's algorithm -- mirrored here only so that the kernel mode has a bit-exact checker) */
void lpo_set_x_update(lpo_t *o, int mode);
/* direct mode: dense index of every row of E among the G rows, -1 for a D row (include/lpbox_hip.h lpbox_get_direct_rows); never called = all G */
void lpo_set_direct_rows(lpo_t *o, const int *gidx_of_row, int l);
/* the reference's default per-iteration text log (does_log, LPh:148; LPcpp:1013-1067) appended to `path`; NULL / "" turns it off (the default here) */
int lpo_set_log(lpo_t *o, const char *path);
/* 1 = print the reference's stop messages to stdout (default 0 = quiet). */
void lpo_set_verbose(lpo_t *o, int verbose);

/* Problem as the reference holds it after readFile (LPcpp:2446-2545): E (l x n) column-major
 * (CSC: colptr[n+1], rowidx[nnz] ascending inside a column), b already NEGATED, f. */
int lpo_set_problem(lpo_t *o, int n, int l, int nnz, const int *colptr, const int *rowidx,
                    const double *vals, const double *b, const double *f);
/* readFile restated with explicit paths (LPcpp:2407-2545): "row,col,val" 1-based triplets, b := -b, f := 1.
 * k is the reference's item-count argument (k==2 negates the matrix values, LPcpp:2436-2439). */
int lpo_read_files(lpo_t *o, const char *path_C, const char *path_b, int k);

int lpo_init(lpo_t *o);                                  /* ADMM_lp_iters_init  LPcpp:489-763  */
int lpo_iters(lpo_t *o, int iter_start, int iter_end);   /* ADMM_lp_iters       LPcpp:766-1095 */
int lpo_iters_l2f(lpo_t *o, int iter_start, int iter_end, const double *vec, int fix_num); /* LPcpp:1098-1574 */

int    lpo_get_n(const lpo_t *o);                        /* LPh:394-396 */
int    lpo_get_org_n(const lpo_t *o);
int    lpo_get_l(const lpo_t *o);
int    lpo_get_iter(const lpo_t *o);                     /* LPh:347-349 */
int    lpo_get_x_iters_rows(const lpo_t *o);
int    lpo_get_x_iters(const lpo_t *o, int ws, double *out);   /* get_x_iters_d LPcpp:1616-1627; out[rows*ws] */
int    lpo_get_x_sol(lpo_t *o, double *out);             /* LPcpp:1648-1665; out[org_n] */
int    lpo_get_final_x_sol(const lpo_t *o, double *out); /* LPcpp:1668-1685; out[x_len]; returns x_len */
double lpo_cal_obj(const lpo_t *o);                      /* LPcpp:1630-1642 */
double lpo_cur_bin_obj(const lpo_t *o);                  /* LPcpp:1644-1646 */
int    lpo_check_infeasible_lpbox(lpo_t *o);             /* LPcpp:1577-1591 */
int    lpo_check_infeasible_l2f(lpo_t *o);               /* LPcpp:1593-1612 */

/* ---- inspection for tests (no reference counterpart) ---- */
/* plain-loop epilogue values (LPcpp:1081: file_idx,-cur_obj,iter+1,secs) */
int    lpo_last_plain_iter_plus1(const lpo_t *o);
long   lpo_total_pcg_iters(const lpo_t *o);
long   lpo_total_outer_iters(const lpo_t *o);
int    lpo_last_pcg_iters(const lpo_t *o);
/* which stop fired last: 0 none, 1 y1_y2 (LPcpp:934/1504), 2 obj_std (:977/:1537), 3 pcg alpha<0 (:1450), 4 all fixed (:1212) */
int    lpo_last_stop_reason(const lpo_t *o);
/* copy a named state vector: "x","y1","y2","y3","z1","z2","z4","b","f","pd","left_idx"; returns length */
int    lpo_get_vec(const lpo_t *o, const char *name, double *out, int cap);
/* named scalars: "rho1","rho2","rho4","prev_rho1","prev_rho4","gamma","dI","rho4Et","std_obj","cur_obj",
 * "sum_fix_obj","best_bin_obj","cvg1","cvg2","obj_val","pow_sqrt_mismatch" */
double lpo_get_scalar(const lpo_t *o, const char *name);
/* per-outer-iteration PCG iteration counts of the LAST lpo_iters/_l2f call; returns count copied */
int    lpo_get_pcg_trace(const lpo_t *o, int *out, int cap);

/* stand-alone pieces for known-answer tests: project_box LPcpp:409-421, project_shifted_Lp_ball LPcpp:423-428 (p = 2) */
void lpo_project_box(int n, const double *x, double *y);
void lpo_project_shifted_lp_ball(int n, const double *x, double *y);

#ifdef __cplusplus
}
#endif
#endif
