// lpbox_capi_internal.h -- glue between the generic C-ABI entry points (lpbox_capi.hip) and the segmentation host code
// (lpbox_seg_capi.hip).  Not part of the public boundary.
#pragma once
#include <cstddef>

int lpbox_fail(int code, const char *fmt, ...);       // records the thread-local error text, returns code

struct SegSolver;
SegSolver *segc_create(int print_info, int device);
void segc_destroy(SegSolver *s);
int segc_set_problem(SegSolver *s, int n, int nnz, const int *rowptr, const int *colidx, const double *vals, const double *b,
                     double c, int rows, int cols);
int segc_set_image(SegSolver *s, const unsigned char *gray, int rows, int cols, int num_nodes);
int segc_init(SegSolver *s);
int segc_legacy(SegSolver *s, int *energy);
int segc_legacy_batch(SegSolver **ss, int count, int *energies);
int segc_l2f(SegSolver *s, int iter_start, int iter_end, const double *vec, int num, int *ret);
int segc_get_n(SegSolver *s);
int segc_get_org_n(SegSolver *s);
int segc_get_iter(SegSolver *s);
int segc_get_x_iters(SegSolver *s, int ws, double *out);
int segc_get_x_iters_device(SegSolver *s, int ws, void **dev_ptr, long *stride);
int segc_set_record(SegSolver *s, int on);
int segc_get_x_history(SegSolver *s, int first, int count, double *out);
int segc_get_x_sol(SegSolver *s, double *out);
int segc_get_obj(SegSolver *s, double *out);
int segc_get_shape(SegSolver *s, int *rows, int *cols);
int segc_get_config(SegSolver *s, int *threads, int *ept, int *groups);
int segc_get_counters(SegSolver *s, long long *outer, long long *pcg);
int segc_get_stop(SegSolver *s, int *reason, int *legacy_iter_p1);
int segc_kernel_time(SegSolver *s, double *ms, long long *launches, int reset);
int segc_debug_vec(SegSolver *s, const char *name, double *out, int cap);
int segc_debug_scalar(SegSolver *s, const char *name, double *out);
int segc_get_problem(SegSolver *s, int *n, int *nnz, int *rowptr, int *colidx, double *vals, double *b, double *c);
