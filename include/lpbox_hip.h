/*
 * lpbox_hip.h -- C-ABI of liblpbox_hip.so, the MI355X (gfx950) Lp-Box ADMM inner solver.
 *
 * This is the drop-in boundary for the reference's Cython solver API.  Each entry point names the
 * reference interface it replaces ("LP pxd" = LinerProgramming/LinearProgramming/cython_solver/LPboxADMMsolver.pxd,
 * "LP pyx" = .../cython_solver/lpbox.pyx, "LPcpp" = .../cython_solver/LPboxADMMsolver.cpp,
 * "SEG pxd/pyx/cpp" = Segmentation/Segmentation/cython/src/{LPboxADMMsolver.pxd,lpbox.pyx,LPboxADMMsolver.cpp}).
 *
 * Conventions
 *  - plain C types only: pointers + sizes, caller-owned buffers, no ownership ever returned
 *    (the reference returns leaked `new double[]`, LPcpp:1620,1657,1677 -- not reproduced);
 *  - every function returns an int status unless documented otherwise: 0 (or a non-negative
 *    payload) = ok, < 0 = LPBOX_E_* error; lpbox_last_error() gives the text.  Nothing calls exit();
 *  - one handle owns one HIP stream; calls on a handle are synchronous and must be serialised by the
 *    caller; different handles may be used from different host threads;
 *  - all arithmetic is fp64 (LPh:16), indices int32 on the API, uint16 inside the kernels;
 *  - there is NO CPU fallback: without a HIP device every compute call fails with LPBOX_E_NODEVICE.
 */
#ifndef LPBOX_HIP_H
#define LPBOX_HIP_H

#ifdef __cplusplus
extern "C" {
#endif

#define LPBOX_OK            0
#define LPBOX_E_BADHANDLE  -1
#define LPBOX_E_BADARG     -2
#define LPBOX_E_STATE      -3   /* call order violated (e.g. iterate before init) */
#define LPBOX_E_IO         -4   /* instance file missing / unparsable (LPcpp:2409-2412 would exit(-1)) */
#define LPBOX_E_HIP        -5   /* HIP runtime error, text in lpbox_last_error() */
#define LPBOX_E_NODEVICE   -6
#define LPBOX_E_UNSUPPORTED -7  /* instance outside what the persistent kernels hold on-chip */
#define LPBOX_E_NOMEM      -8
#define LPBOX_E_TOOLARGE   -9   /* lpbox_init: the instance fits neither the register slots nor the LDS of one CU -- same algorithm,
                                   other entry points: lpbox_big_* (the drop-in classes route there by this code, not by the text) */

#define LPBOX_FLAVOUR_LP   0    /* min b'x  s.t. Ex<=f, x in {0,1}^n      (LPcpp) */
#define LPBOX_FLAVOUR_SEG  1    /* min x'Ax + b'x, x in {0,1}^n           (SEGcpp) */

typedef struct lpbox_solver lpbox_t;      /* one or a batch of independent instances on one GPU */

/* ---- library / device ---------------------------------------------------------------------- */
const char *lpbox_version(void);
const char *lpbox_last_error(void);               /* thread-local text of the last failure */
int  lpbox_device_count(void);                    /* number of HIP devices visible (0 if none) */
int  lpbox_set_device(int device);                /* device used by handles created afterwards on this thread */

/* ---- object life-cycle ---------------------------------------------------------------------- */
/* LP pxd:6-8 / LP pyx:13-14  `LPboxADMMsolver(int print_info)`  (SEG pyx:14-15 for flavour SEG).
 * batch = number of independent instances held by the handle (1 = the reference's single object). */
lpbox_t *lpbox_create(int flavour, int batch, int print_info);
void     lpbox_destroy(lpbox_t *h);

/* ---- problem input (instance index idx in [0,batch)) ------------------------------------------ */
/* What readFile leaves in the object (LPcpp:2446-2545): E (l x n) column-major, row indices ascending in a
 * column, all stored values must be 1.0 (vals == NULL means all ones; anything else -> LPBOX_E_UNSUPPORTED),
 * b ALREADY NEGATED (LPcpp:2520), f (NULL means all ones, LPcpp:2522). */
int lpbox_set_problem_lp(lpbox_t *h, int idx, int n, int l, int nnz, const int *colptr, const int *rowidx,
                         const double *vals, const double *b, const double *f);
/* LP pxd:9 `void readFile(int,int,int)` (LPcpp:2446-2545) with the two paths explicit; k as in the reference. */
int lpbox_read_files_lp(lpbox_t *h, int idx, const char *path_C, const char *path_b, int k);
/* LP pxd:9 verbatim: instance i, k items, j bids under `<root>/instance/<k>_<j>/instance_<i>_{C,b}.txt`;
 * root NULL = the reference's "../cython_solver/data" relative to the CWD (LPcpp:2451). */
int lpbox_read_file(lpbox_t *h, int idx, const char *root, int i, int k, int j);

/* The instance as set_problem / readFile left it in the handle (what LPcpp:2446-2545 leaves in the solver's members n, l, E, b, f): sizes
 * always, arrays where the pointer is not NULL (colptr n+1, rowidx nnz, b n, f l).  Used by the Python class to hand an instance
 * that exceeds the on-chip kernel (max(n, l) > 2048) over to the large-instance path. */
int lpbox_get_problem_lp(lpbox_t *h, int idx, int *n, int *l, int *nnz, int *colptr, int *rowidx, double *b, double *f);

/* ---- solver (whole batch per call) ------------------------------------------------------------ */
/* LP pxd:10 `int ADMM_lp_iters_init()` (LPcpp:489-763).  Returns 1 like the reference. */
int lpbox_init(lpbox_t *h);
/* LP pxd:11 `int ADMM_lp_iters(int,int)` (LPcpp:766-1095).  rets[batch] receives each instance's return value
 * (1 iff stopped by the objective-std test, LPcpp:977-979); may be NULL.  Function returns rets[0]. */
int lpbox_iterate(lpbox_t *h, int iter_start, int iter_end, int *rets);
/* LP pxd:13 `int ADMM_lp_iters_l2f(int,int,double*,int)` (LPcpp:1098-1574).
 * vec: for instance idx the fix vector starts at vec + idx*vec_stride and holds >= n_live entries in {1,0,-1}
 * over the CURRENT live variables; nums[idx] = count of non -1 entries (0 = ignore vec, as LPcpp:1124).
 * rets[batch] as above (1 = converged / all fixed / PCG broke down).  Function returns rets[0]. */
int lpbox_iterate_l2f(lpbox_t *h, int iter_start, int iter_end, const double *vec, long vec_stride,
                      const int *nums, int *rets);

/* ---- results (instance idx) ------------------------------------------------------------------- */
int lpbox_get_n(lpbox_t *h, int idx);                         /* LP pxd:15 get_n(): live variables          */
int lpbox_get_org_n(lpbox_t *h, int idx);                     /* SEG pxd get_org_n(); original n            */
int lpbox_get_l(lpbox_t *h, int idx);
int lpbox_get_iter(lpbox_t *h, int idx);                      /* LP pxd:16 get_iter() (LPh:347-349)         */
/* LP pxd:14 `double* get_x_iters_d(int)` (LPcpp:1616-1627): out[rows*ws] row-major, rows = n_live of the last
 * l2f call; returns rows.  out may be NULL to query rows. */
int lpbox_get_x_iters(lpbox_t *h, int idx, int ws, double *out);
/* Plain loop with print_fix_info 2/3 (LPcpp:776-779, 903-909, 940-946, 986-992): the reference streams x_sol of every iteration to
 * <root>/xiter/<k>_<j>_xiters_<i>.csv.  With record on, lpbox_iterate keeps those iterates on the device (one column per iteration of the
 * call, exactly like the x_iters window of the l2f loop) and lpbox_get_x_iters / _device hand them out; the CSV itself is written by the
 * host wrapper.  Off by default. */
int lpbox_set_record(lpbox_t *h, int on);
/* LP flavour, OPT-IN and without a reference counterpart: how the x-update solves ((rho1+rho2) I + rho4 E^T E) x = rhs (the system of
 * LPcpp:872-894).  LPBOX_XUPDATE_PCG (default) = the reference's Jacobi-PCG to 1e-3 (LPcpp:251-335), bit-exact against the oracle of the
 * reference's algorithm.  LPBOX_XUPDATE_DIRECT = an exact solve through the Woodbury identity with a dense l x l inverse kept in LDS
 * (DESIGN.md section 17): a different (more accurate) x-update, hence different iterates and iteration counts than the reference's;
 * available for n <= 512 with at most 128 rows of E that share columns with other rows (the XOR rows of an auction do not).  May be switched between calls; returns LPBOX_E_UNSUPPORTED when the batch does not fit. */
#define LPBOX_XUPDATE_PCG 0
#define LPBOX_XUPDATE_DIRECT 1
int lpbox_set_x_update(lpbox_t *h, int mode);
/* The row split of the direct mode for instance idx (the order of its arithmetic; tests hand it to the oracle's mirror): gidx_of_row[l] =
 * dense index of the row among the rows solved through the on-chip inverse, -1 for a row handled in closed form (its columns meet no
 * other such row).  Returns the number of dense rows. */
int lpbox_get_direct_rows(lpbox_t *h, int idx, int *gidx_of_row);
/* Segmentation flavour: with record on (on == 1: up to 2000 iterations, on > 1: that many), lpbox_seg_legacy keeps x_sol of every iteration
 * (print_info 1 -> ../xiter/<problem>.csv, SEGcpp:1209-1213, 1270-1277).  out == NULL: number of iterations recorded; otherwise copies
 * iterations [first, first+count) as count rows of org_n doubles. */
int lpbox_seg_get_x_history(lpbox_t *h, int first, int count, double *out);
/* ---- the early-fixing policy's encoder, fused on the device (the step either side of the solver in LP/trainer.py:523-535) ----
 * GraphAttentionEncoder up to its flatten (LP/mha.py:202-243; SEG/mha.py same with 5 tokens), eval mode, fp16 MFMA with fp32
 * accumulation.  x_dev: fp64 iterates (typically the buffer lpbox_get_x_iters_device returns); variable r's token t is
 * x_dev[row_off_dev[r] + t*tok_stride + 0..4] (LP/trainer.py:526-527 -> tok_stride 5, tokens 20; SEG/trainer.py:721-725 ->
 * tok_stride 1, tokens 5).  weights_dev / consts_dev: packed as csrc/lpbox_policy.h describes (lpbox_hip/policy.py packs a
 * reference state_dict); lpbox_policy_layout gives their sizes.  out_dev: fp16 [rows][tokens*128].  Asynchronous on hip_stream. */
int lpbox_policy_layout(int tokens, long *weight_halves, long *const_floats);
int lpbox_policy_encode_f16(const double *x_dev, const long long *row_off_dev, long rows, int tokens, int tok_stride,
                            const void *weights_dev, const float *consts_dev, void *out_dev, void *hip_stream);
/* The same encoder in FLOAT32 on the matrix cores (v_mfma_f32_16x16x4_f32: f32 in, f32 accumulate -- the reference's arithmetic, LP/mha.py
 * evaluates in float32) at usable speed.  weights_dev: f32 fragments packed by lpbox_hip/policy.py (lpbox_policy_f32frag_layout gives the
 * count; csrc/lpbox_policy_f32_kernels.hip documents the order), consts_dev: the SAME constant block as lpbox_policy_encode_f16.
 * out_dev: float [rows][tokens*128].  Asynchronous on hip_stream. */
int lpbox_policy_f32frag_layout(int tokens, long *weight_floats, long *const_floats);
int lpbox_policy_encode_f32(const double *x_dev, const long long *row_off_dev, long rows, int tokens, int tok_stride,
                            const float *weights_dev, const float *consts_dev, float *out_dev, void *hip_stream);
/* The whole network -- encoder and MLP head -- in fp32 on the device, one workgroup per variable: the reference's arithmetic
 * (LP/mha.py evaluates in float32), for fixing decisions near a threshold and as the fp32 check of the fused kernel; not a fast
 * path.  weights_dev: floats in the order csrc/lpbox_policy_kernels.hip documents (lpbox_policy_f32_layout gives the count);
 * sigmoid_dev[rows] (and logit_dev[rows] if not NULL) receive the scores.  Asynchronous on hip_stream. */
int lpbox_policy_f32_layout(int tokens, long *weight_floats);
int lpbox_policy_score_f32(const double *x_dev, const long long *row_off_dev, long rows, int tokens, int tok_stride,
                           const float *weights_dev, float *sigmoid_dev, float *logit_dev, void *hip_stream);
/* The same kernel as a filter: sigmoid_dev[rows] already holds scores (the fused fp16 path's); every variable whose score lies within
 * `band` of thr_hi or thr_lo is re-evaluated in fp32 and overwritten, the others are left alone; *count_dev (optional, device) is
 * incremented per re-scored variable.  No host round trip: the selection happens on the device. */
int lpbox_policy_rescore_f32(const double *x_dev, const long long *row_off_dev, long rows, int tokens, int tok_stride,
                             const float *weights_dev, float *sigmoid_dev, float band, float thr_hi, float thr_lo,
                             unsigned long long *count_dev, void *hip_stream);
/* Batched use only: park (active[i] == 0) or resume instances.  A parked instance is skipped by lpbox_iterate / _l2f and keeps its
 * state and return code; the reference has no counterpart because its loop simply stops calling a finished solver
 * (LP/trainer.py:511-512).  active == NULL resumes all. */
int lpbox_set_active(lpbox_t *h, const int *active);
/* The same (rows x ws) row-major windows for the WHOLE batch, left on the device: instance idx starts at dev_ptr + idx*stride
 * doubles and holds lpbox_get_x_iters(h, idx, ws, NULL) rows.  Lets a policy network read the iterates without a host
 * round trip (valid until the next solver call). */
int lpbox_get_x_iters_device(lpbox_t *h, int ws, void **dev_ptr, long *stride_doubles);
int lpbox_get_x_sol(lpbox_t *h, int idx, double *out);        /* LP pxd:17 (LPcpp:1648-1665): out[org_n] in {0,1} */
int lpbox_get_final_x_sol(lpbox_t *h, int idx, double *out);  /* LP pxd:18 (LPcpp:1668-1685): raw live x, returns its length */
int lpbox_cal_obj(lpbox_t *h, int idx, double *out);          /* LP pxd:12 cal_obj() (LPcpp:1630-1642)      */
int lpbox_cur_bin_obj(lpbox_t *h, int idx, double *out);      /* LP pxd:19 get_curBinObj() (LPcpp:1644-1646) */
int lpbox_check_infeasible_lpbox(lpbox_t *h, int idx);        /* LP pxd:20 (LPcpp:1577-1591): count, >= 0   */
int lpbox_check_infeasible_l2f(lpbox_t *h, int idx);          /* LP pxd:21 (LPcpp:1593-1612): count, >= 0   */

/* ---- segmentation flavour (handle created with LPBOX_FLAVOUR_SEG, batch 1) ---------------------
 * The generic entry points above serve it too: lpbox_init = ADMM_bqp_unconstrained_init (SEG pxd:9, SEGcpp:658-810, after the
 * problem has been set), lpbox_iterate_l2f = ADMM_bqp_unconstrained_l2f (SEG pxd:11, SEGcpp:917-1195, windows of <= 10
 * iterations), lpbox_get_x_iters / _get_n / _get_org_n / _get_x_sol (SEG pxd:12-16). */
/* What the reference's init holds after get_A_b_from_cost (SEGcpp:226-248, :755-756): A_ptr = A/2 row-major CSR (columns
 * ascending, every row stores its diagonal), b, the constant c, and the image shape (for save_img). */
int lpbox_set_problem_bqp(lpbox_t *h, int n, int nnz, const int *rowptr, const int *colidx, const double *vals,
                          const double *b, double c, int rows, int cols);
/* The image part of ADMM_bqp_unconstrained_init (SEGcpp:702-756): grayscale pixels (rows x cols, row-major, what
 * cv::imread(path, 0) yields), scaled to ~num_nodes pixels (cv::resize INTER_LINEAR), costs per SEGcpp:46-248. */
int lpbox_seg_set_image(lpbox_t *h, const unsigned char *gray, int rows, int cols, int num_nodes);
/* `cv::imread(imagePath, 0)` (SEGcpp:705) without an image library: the luminance plane of a baseline / extended-sequential 8-bit
 * Huffman JPEG, reconstructed with libjpeg's default ISLOW inverse DCT -- bit for bit what libjpeg yields for JCS_GRAYSCALE output,
 * which is what OpenCV's grayscale imread and PIL's draft("L") hand out (csrc/lpbox_jpeg_host.cpp; pinned against PIL in the tests).
 * out == NULL: only *rows / *cols.  Progressive or arithmetic-coded files: LPBOX_E_BADARG with the reason in lpbox_last_error(). */
int lpbox_read_jpeg_gray(const char *path, unsigned char *out, long cap, int *rows, int *cols);
/* SEG pxd:10 `int ADMM_bqp_unconstrained_legacy()` (SEGcpp:1200-1380): *energy = int(cur_obj + c). */
int lpbox_seg_legacy(lpbox_t *h, int *energy);
/* solve_init + solve_iter for `count` segmentation handles (each after lpbox_seg_set_image / lpbox_set_problem_bqp) advanced in lockstep by
 * ONE launch chain -- the workload of image_segmentation.cpp:24-29 (images 0..99 at 1e4 nodes), where a single solve is launch-bound.
 * Per problem the arithmetic is that of lpbox_init + lpbox_seg_legacy on its own (own control state, sums, iteration counts); afterwards
 * every handle answers its getters as usual.  energies[count]. */
int lpbox_seg_legacy_batch(lpbox_t **handles, int count, int *energies);
/* SEG pxd:15 `double get_final_obj()` (SEGcpp:868-893): energy of the assembled rounded solution on the ORIGINAL A, b, plus c. */
int lpbox_seg_get_obj(lpbox_t *h, double *out);
int lpbox_seg_get_shape(lpbox_t *h, int *rows, int *cols);   /* scaled_row, scaled_col (SEGcpp:716-717), for save_img */
/* inspection: the problem the handle holds (pass NULL arrays to query n / nnz first) */
int lpbox_seg_get_problem(lpbox_t *h, int *n, int *nnz, int *rowptr, int *colidx, double *vals, double *b, double *c);

/* ---- extensions beyond the pxd (batching, measurement, inspection) ---------------------------- */
/* Workgroup geometry picked for the batch: threads per instance and slots per thread (threads*slots storage positions;
 * the reduction tree depends on both). */
int lpbox_get_config(lpbox_t *h, int *threads, int *elems_per_thread, int *lds_bytes);
/* Storage position of every variable of instance idx (pos_of_var[org_n]): the kernels keep variables sorted by column
 * length, and the reduction tree is defined over positions (element at position p -> thread p % threads). */
int lpbox_get_layout(lpbox_t *h, int idx, int *pos_of_var);
/* Number of lanes (1,2,4,8) that share the sum of each row of E in instance idx (lanes_of_row[l]): lane g adds the entries
 * g, g+G, ... of the row in ascending column order, the G partial sums are combined by a butterfly. */
int lpbox_get_row_split(lpbox_t *h, int idx, int *lanes_of_row);
/* How the kernels add up column j of E (E^T w) in instance idx: own[j] leading entries (ascending row id) by the variable's
 * own lane; for a split column the rest in consecutive chunks of help4[4*j + q] entries by lane q = 0..3 of its quad of lanes
 * (0 for the own lane), the four chunk sums h_q combined as (h0 + h1) + (h2 + h3) and added to the own part.  own[org_n],
 * help4[4*org_n]; an unsplit column has own[j] = its length and help4 = 0. */
int lpbox_get_col_split(lpbox_t *h, int idx, int *own, int *help4);
/* Counters accumulated since init: outer ADMM iterations and PCG iterations of instance idx (LPcpp:894 maxiter). */
/* The reference's per-iteration text log (does_log, LPh:148, written by ADMM_lp_iters, LPcpp:789, :898-901, :1013-1067), opt-in: while on,
 * every lpbox_iterate call keeps LPBOX_LOG_VALS doubles per iteration it completed -- PCG iterations, |x_sol|, |y1|, |y2|, |y3|, |z1|, |z2|,
 * |z4|, b.x ("dou_obj"), b.round(x) ("bin_obj"), seconds since the call started (device clock), iteration -- and lpbox_get_log returns the
 * records of instance idx in iteration order (rows written; an iteration that stopped the loop is not logged, as in the reference).
 * Default PCG kernels only (LPBOX_E_UNSUPPORTED otherwise); the Python wrapper formats the text file. */
#define LPBOX_LOG_VALS 12
int lpbox_set_log(lpbox_t *h, int on);
int lpbox_get_log(lpbox_t *h, int idx, double *out, int cap_rows);
int lpbox_get_counters(lpbox_t *h, int idx, long long *outer_iters, long long *pcg_iters);
/* Which stop fired in the last call: 0 none, 1 y1_y2, 2 obj_std, 3 PCG alpha<0, 4 all fixed; and iter+1 of the last plain call (LPcpp:1081). */
int lpbox_get_stop(lpbox_t *h, int idx, int *reason, int *plain_iter_plus1);
/* Device time (ms, HIP events on the handle's stream) and launch count of the ADMM window kernel since the last reset. */
int lpbox_kernel_time(lpbox_t *h, double *ms_total, long long *launches, int reset);
/* Copy a named device state vector of instance idx ("x","z1","z2","z4","f","pd","b"); returns its length. */
int lpbox_debug_get_vec(lpbox_t *h, int idx, const char *name, double *out, int cap);
int lpbox_debug_get_scalar(lpbox_t *h, int idx, const char *name, double *out);

/* ---- LARGE single LP, variable-sharded over ranks (BASELINE config 5; no reference counterpart: the reference is one
 * process).  Rank r holds the columns [c0, c0+n_loc) of E (all l rows), its slice of b, and the full f; the l-vectors are
 * replicated.  Where the algorithm sums over all variables (E*v: an l-vector once per PCG iteration and twice per outer
 * iteration; <= 5 scalars per reduction) the ranks exchange their contributions and EVERY rank adds them in rank order
 * (c0 + c1) + c2 ..., so a W-rank run is reproducible and has an exact CPU model (oracle lpo_set_ranks).  Transport, chosen
 * before lpbox_big_init:
 *   - RCCL driven by the library itself on the handle's stream (lpbox_big_rccl_unique_id on one rank, the 128 bytes handed to
 *     every rank by the caller's control plane, lpbox_big_rccl_init on every rank): per E*v one grouped send/recv of row blocks
 *     (a reduce-scatter whose adds are ours) + one all-gather; per scalar group one all-gather.  No Python in the loop.
 *   - or a caller-supplied all-gather `fn(send_dev, count, recv_dev, user)`: recv_dev[r*count .. (r+1)*count) := rank r's
 *     send_dev[0..count), stream-ordered after the work already queued on the handle's stream (tests: gloo through torch).
 * world == 1 without a transport issues no collective.  Semantics: ADMM_lp_iters (LPcpp:766-1095). */
typedef struct lpbox_big lpbox_big_t;
typedef int (*lpbox_allgather_fn)(const void *send_dev, long count, void *recv_dev, void *user);
lpbox_big_t *lpbox_big_create(int rank, int world, int device);
void lpbox_big_destroy(lpbox_big_t *h);
int lpbox_big_set_stream(lpbox_big_t *h, void *hip_stream);
int lpbox_big_set_allgather(lpbox_big_t *h, lpbox_allgather_fn fn, void *user);
int lpbox_big_rccl_unique_id(void *out128);                    /* ncclGetUniqueId: 128 bytes, returns 128 */
int lpbox_big_rccl_init(lpbox_big_t *h, const void *unique_id128);   /* ncclCommInitRank(world, id, rank) on the handle's device */
int lpbox_big_set_problem(lpbox_big_t *h, long n_glob, int c0, int n_loc, int l, const int *colptr, const int *rowidx,
                          const double *b, const double *f);
/* Opt-in, NOT the reference's arithmetic and outside the parity claim (like lpbox_set_x_update): the PCG's step length from
 * p.Mp = dI (p.p) + r4Et (q.q), q = E p, instead of the dot product p.(M p) of LPcpp:300 -- the p.p partials ride with the q exchange, the column
 * product and the vector updates become one kernel: 3 instead of 4 RCCL operations and 2 instead of 3 launches per PCG iteration.  Call before
 * lpbox_big_init.  Bit-exact against its own oracle mirror (lpo_set_pcg_lean). */
#define LPBOX_PCG_REFERENCE 0
#define LPBOX_PCG_COMM_LEAN 1
int lpbox_big_set_pcg_mode(lpbox_big_t *h, int mode);
int lpbox_big_init(lpbox_big_t *h);                                             /* ADMM_lp_iters_init LPcpp:489-763 */
int lpbox_big_iterate(lpbox_big_t *h, int iter_start, int iter_end, int *ret);  /* ADMM_lp_iters      LPcpp:766-1095 */
/* print_fix_info 2 behind the size hand-over (LPcpp:777-780, :903-909): the following lpbox_big_iterate calls keep x after every
 * iteration on the device; lpbox_big_get_x_iters(ws = iterations of the call) reads the (rows x ws) block back. */
int lpbox_big_set_record(lpbox_big_t *h, int on);
/* ADMM_lp_iters_l2f (LPcpp:1098-1574) on the sharded instance.  vec_local: this rank's slice of the fix vector, one entry per LOCAL
 * live variable in ascending order (1.0 / 0.0 fix, anything else leave); num_global: number of fixes over ALL ranks (callers sum their
 * local counts with one tiny all-reduce; it sets the shrunken n of the sphere projection, LPcpp:427).  x_iters of the window stay on
 * the device per rank: a policy scores its own shard, no gather. */
int lpbox_big_iterate_l2f(lpbox_big_t *h, int iter_start, int iter_end, const double *vec_local, long num_global, int *ret);
int lpbox_big_get_n(lpbox_big_t *h);                                            /* live variables of this rank */
int lpbox_big_get_x_iters(lpbox_big_t *h, int ws, double *out);                 /* (rows x ws) row-major; out == NULL: rows */
int lpbox_big_get_x_iters_device(lpbox_big_t *h, int ws, void **dev_ptr, int *rows);
int lpbox_big_get_x_sol(lpbox_big_t *h, double *out_local);                     /* binary: fixed value / rounded x (LPcpp:1648-1666) */
int lpbox_big_cal_obj(lpbox_big_t *h, double *out);                             /* sum_fix_obj + cur_obj (LPcpp:1630-1642) */
int lpbox_big_get_x(lpbox_big_t *h, double *out_local);                         /* this rank's slice of x_sol */
int lpbox_big_get_vec(lpbox_big_t *h, const char *name, double *out, long cap); /* "x","z1","z2","pd","live" (local), "z4","Ex" (rows) */
/* LP pxd:20 / :21 on the large path, one rank only: which = 0 check_infeasible_lpbox (LPcpp:1577-1591: rows of the CURRENT E, live
 * columns, raw iterate), 1 check_infeasible_l2f (LPcpp:1593-1612: original E times the binary full-length solution); count >= 0. */
int lpbox_big_check_infeasible(lpbox_big_t *h, int which);
int lpbox_big_get_scalar(lpbox_big_t *h, const char *name, double *out);        /* "cur_obj","iter","stop","outer_total","pcg_total",... */

/* ---- generic constrained binary QP: min x'Ax + b'x  s.t.  Cx = d, Ex <= f, x in {0,1}^n -----------------------------------------
 * The reference's ADMM_bqp (Segmentation/.../LPboxADMMsolver.cpp:1384-1832) behind ADMM_bqp_unconstrained / _linear_eq / _linear_ineq /
 * _linear_eq_and_uneq (:1834-2109; C++ only, not in the pyx).  Matrices are CSR with ascending columns; A must store every diagonal
 * entry (the reference adds rho to A.diagonal() in place, :1483).  m = 0: no equality constraints, l = 0: no inequality constraints. */
typedef struct lpbox_bqp lpbox_bqp_t;
lpbox_bqp_t *lpbox_bqp_create(int device);
void lpbox_bqp_destroy(lpbox_bqp_t *h);
int lpbox_bqp_preset(lpbox_bqp_t *h, int type);      /* hyper-parameters of ADMM_bqp_{unconstrained,linear_eq,linear_ineq,linear_eq_and_uneq}_init (:587-672): 0,1,2,3 */
int lpbox_bqp_set_params(lpbox_bqp_t *h, const double *p11);   /* stop_threshold, std_threshold, gamma_val, gamma_factor, rho_change_step, max_iters, initial_rho, history_size, learning_fact, pcg_tol, pcg_maxiters */
int lpbox_bqp_set_problem(lpbox_bqp_t *h, int n, const int *Ap, const int *Ai, const double *Av, const double *b, const double *x0,
                          int m, const int *Cp, const int *Ci, const double *Cv, const double *d,
                          int l, const int *Ep, const int *Ei, const double *Ev, const double *f);
int lpbox_bqp_solve(lpbox_bqp_t *h, int *iterations);          /* the whole ADMM_bqp loop; *iterations = the `iter` it ended on */
int lpbox_bqp_get_vec(lpbox_bqp_t *h, const char *name, double *out, long cap);   /* Solution (LPh): "x" (x_sol), "y1", "y2", "best_sol"; also "z1","z2","z3","z4","y3" */
int lpbox_bqp_get_scalar(lpbox_bqp_t *h, const char *name, double *out);          /* "iters","stop" (1 xyy, 2 obj_std, 0 max_iters),"cur_obj","best_bin_obj","total_pcg",... */

#ifdef __cplusplus
}
#endif
#endif
