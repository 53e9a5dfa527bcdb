/* bqp_oracle.h -- CPU ORACLE (test infrastructure only) of the reference's generic constrained BQP solver ADMM_bqp
 * (Segmentation/Segmentation/cython/src/LPboxADMMsolver.cpp:1384-1832).  See bqp_oracle.c. */
#ifndef BQP_ORACLE_H
#define BQP_ORACLE_H
#ifdef __cplusplus
extern "C" {
#endif
typedef struct bqpo bqpo_t;
bqpo_t *bqpo_create(void);
void bqpo_destroy(bqpo_t *o);
void bqpo_set_order(bqpo_t *o, int mode, int T, int chunk);     /* 0 Eigen order, 1 the kernels' two-level tree */
int bqpo_preset(bqpo_t *o, int type);                           /* 0 unconstrained, 1 equality, 2 inequality, 3 both (the *_init() presets) */
int bqpo_set_params(bqpo_t *o, const double *p11);              /* stop, std, gamma, gamma_factor, rho_step, max_iters, rho0, history, learning_fact, pcg_tol, pcg_maxiters */
/* A: n x n CSR with every diagonal entry stored; C: m x n (m = 0: none); E: l x n (l = 0: none); columns ascending inside a row */
int bqpo_set_problem(bqpo_t *o, int n, const int *Ap, const int *Ai, const double *Av, const double *b, const double *x0,
                     int m, const int *Cp, const int *Ci, const double *Cv, const double *d,
                     int l, const int *Ep, const int *Ei, const double *Ev, const double *f);
int bqpo_solve(bqpo_t *o);                                      /* returns the iteration index the loop ended on */
int bqpo_get_vec(const bqpo_t *o, const char *name, double *out, int cap);
double bqpo_get_scalar(const bqpo_t *o, const char *name);
int bqpo_get_trace(const bqpo_t *o, int *out, int cap);
#ifdef __cplusplus
}
#endif
#endif
