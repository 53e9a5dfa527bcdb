/* seg_oracle.h -- CPU ORACLE (test infrastructure) for the segmentation flavour; see seg_oracle.c. */
#ifndef SEG_ORACLE_H
#define SEG_ORACLE_H
#ifdef __cplusplus
extern "C" {
#endif
typedef struct sego sego_t;

sego_t *sego_create(int print_info, int numNodes, int problem);           /* SEGcpp:650-655 */
void    sego_destroy(sego_t *o);
void    sego_set_order(sego_t *o, int mode, int T, int chunk);            /* 0 = Eigen order, 1 = GPU order (T threads own `chunk` consecutive indices) */
void    sego_set_verbose(sego_t *o, int v);

/* SEGcpp:46-248 + :727,:747,:755: grayscale image (rows x cols, row-major, values 0..255) -> A_ptr (row-major CSR), b, c.
 * Returns nnz; arrays are malloc'ed, release with sego_free_arrays. */
int  sego_build_costs(int rows, int cols, const double *img, int *n_out, int **rowptr_out, int **colidx_out,
                      double **val_out, double **b_out, double *c_out);
void sego_free_arrays(int *a, int *b, double *c, double *d);
/* cv::resize(.., Size(), scale, scale, INTER_LINEAR) on 8-bit data, restated (SEGcpp:705-714); dst may be NULL to query the size */
int  sego_resize_linear_u8(int rows, int cols, const unsigned char *src, double scale, int *orows, int *ocols, unsigned char *dst);

int  sego_set_problem(sego_t *o, int n, int nnz, const int *rowptr, const int *colidx, const double *vals, const double *b,
                      double c, int scaled_row, int scaled_col);
int  sego_init(sego_t *o);                                                /* ADMM_bqp_unconstrained_init   SEGcpp:658-810  */
int  sego_legacy(sego_t *o);                                              /* ADMM_bqp_unconstrained_legacy SEGcpp:1200-1380 */
int  sego_l2f(sego_t *o, int iter_start, int iter_end, const double *vec, int fix_num);   /* SEGcpp:917-1195 */

int    sego_get_n(const sego_t *o);
int    sego_get_org_n(const sego_t *o);
int    sego_get_x_iters_rows(const sego_t *o);
int    sego_get_x_iters(const sego_t *o, int ws, double *out);            /* SEGcpp:839-851 */
int    sego_get_x_sol(const sego_t *o, double *out);                      /* SEGcpp:895-914 */
double sego_get_final_obj(sego_t *o);                                     /* SEGcpp:868-893 */
double sego_get_c(const sego_t *o);
int    sego_last_stop(const sego_t *o);                                   /* 1 xyy, 2 obj_std, 4 all fixed */
int    sego_legacy_iter_plus1(const sego_t *o);
long   sego_total_pcg(const sego_t *o);
long   sego_total_outer(const sego_t *o);
int    sego_get_trace(const sego_t *o, int *out, int cap);
int    sego_get_vec(const sego_t *o, const char *name, double *out, int cap);
double sego_get_scalar(const sego_t *o, const char *name);
#ifdef __cplusplus
}
#endif
#endif
