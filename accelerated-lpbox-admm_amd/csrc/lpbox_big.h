// lpbox_big.h -- internal layout of the LARGE-instance LP path (BASELINE config 5: one LP with up to ~1e6 variables,
// variable-sharded over the GPUs of a node).  Not part of the C-ABI.
//
// A rank owns the contiguous column block [c0, c1) of E (all rows): its n-vectors x, y1, y2, z1, z2, b, ... are local,
// the l-vectors y3, z4, f, E*x, E*p are REPLICATED (every rank computes all l entries).  One ADMM iteration is a chain of
// kernels cut at the grid-wide dependencies; where the algorithm needs a sum over all variables the chain has a collective:
//   * E*v  = sum over ranks of E[:, shard] * v_shard      -> all-reduce of an l-vector (once per PCG iteration + 2 per outer)
//   * dot products / norms                                 -> all-reduce of <= 5 scalars
// (north_star's "one all-reduce per iteration" under-counts: SURVEY section 8e.)  With one rank the collectives vanish and
// the path is bit-comparable with the oracle (two-level reduction order, rows and columns summed in ascending order).
// Control state lives on the device (BigState, ping-ponged), kernels fall through once the PCG has converged / the solver
// has stopped, so the host enqueues a static sequence.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "lpbox_lp.h"   // hyper-parameters LP_* (LPcpp:491-507) and stop reasons

#define BIG_T 256
#define BIG_NPART 8
#define BIG_MAXW 16        // ranks whose workgroup counts fit the descriptor (folded reductions; more ranks use the reduction launch)
#define BIG_FOLD_U 4       // folded reductions: at most this many workgroup partials per thread (G <= BIG_FOLD_U * BIG_T)
// Workgroup partials are kept per PHASE of the chain (a kernel that consumes the partials of one phase while it produces those of the
// next must not overwrite what a slower workgroup still has to read): A prep -> y, B resid -> rows(first), C pcg_cols -> pcg_upd,
// D pcg_upd -> rows / post, E post -> prep; X = the rare passes (init, fix), always reduced by big_k_fin.
enum { BIG_PH_A = 0, BIG_PH_B, BIG_PH_C, BIG_PH_D, BIG_PH_E, BIG_PH_X, BIG_PH_COUNT };

enum { BIG_HALT_NONE = 0, BIG_HALT_STOP = 1, BIG_HALT_WINDOW = 2, BIG_HALT_PCG_MORE = 3 };

struct BigState {
    double rho1, rho2, rho4, prev_rho1, prev_rho2, prev_rho4, gamma_val, dI, r4Et, rcr;
    double std_obj, cur_obj, best_bin_obj, cvg1, cvg2, obj_val, c1;
    double hist[LP_HIST];
    double threshold, absNew, rhsNorm2, beta;
    int rhoUpdated, hist_n, iter, iter_start, iter_end, have_prev, halt, stop, ret;
    int pcg_k, pcg_done, pcg_first, phase;
    int pcg_total, outer_total, last_pcg, plain_iter_p1, pcg_max, expr_ready;
    int l2f, cc, n_live_lo, n_live_hi;      // l2f window semantics (LPcpp:1098-1574); x_iters column; live variables (all ranks), 2 x 31 bits
    double sum_fix_obj, fix_obj, prev_sum, prev_obj;
    int rec;                                // plain loop: keep x of every iteration in xhist too (print_fix_info 2, LPcpp:903-909)
};

struct BigDev {
    int n_loc, l, G, Gl, EPT, EPTl;       // G workgroups over the local variables, Gl over the rows
    int P, Glr;                           // column slices of the row-side storage (1 = plain CSR); workgroups of big_k_rows at one row per thread
    long n_glob;
    // E restricted to the local columns: CSR (rows -> local column index; slice-major when P > 1: rptr has P * l + 1 entries, see
    // row_sum_sliced) and CSC (local column -> rows), values all 1
    const int *rptr, *rcol, *cptr, *crow;
    double *x, *y1, *y2, *z1, *z2, *b, *pd, *dinv, *rhs, *r, *z, *tmp, *p0, *p1, *gsrc;   // local n-vectors
    double *y3, *z4, *f, *Ex, *q;         // replicated l-vectors (q doubles as the all-reduce buffer of E*v)
    double2 *fz;                          // (f - y3, z4) per row: the two l-vectors the rhs assembly gathers, as ONE 16-byte element
    double2 *zp;                          // (z, p) of the last PCG update packed per variable: ONE 16-byte gather per entry of E*p
    double *xt;                           // PCG iterate (committed to x for the live variables after the PCG)
    uint8_t *live;                        // 1 live, 0 fixed (x holds the fixed value)
    const uint8_t *newfix;                // this call's fix request: 0 none, 1 -> 0.0, 2 -> 1.0
    double *xhist; int ws_cap;            // x_iters staging [ws_cap][n_loc]
    double *part;                         // [BIG_PH_COUNT][BIG_NPART][Gs] workgroup partials (Gs = partial stride, the same on every rank)
    double *red;                          // [BIG_PH_COUNT][BIG_NPART] reduced scalars of each phase (all-reduced over the ranks) -- the route
                                          // with a reduction launch.  Per phase: a chain that halts inside the PCG still runs the launches
                                          // enqueued behind it, and the resumed PCG needs the totals it halted on
    // FOLDED reductions (G <= BIG_FOLD_U * BIG_T): no reduction launch -- every consumer workgroup adds up the workgroup partials itself
    // (same two-level tree, same bits); with W > 1 ranks the partials of all ranks are all-gathered into gpart[phase][rank][nv][Gs]
    // and every consumer runs rank r's tree over its Gr[r] partials, then adds the W totals in rank order (what big_k_rank_sum did).
    int fold, W, gathered, Gs, Gr[BIG_MAXW];      // gathered: the consumers read gpart (W > 1, or an RCCL communicator of one rank)
    const double *gpart;
    // opt-in COMM-LEAN PCG (lpbox_big_set_pcg_mode; NOT the reference's arithmetic): the step length comes from
    // p.Mp = dI (p.p) + r4Et (q.q), q = E p.  p.p partials (one per column workgroup) and q.q partials (one per row workgroup: of the
    // row gather on one rank, of the rank's row block otherwise) sit in lsmall[0 .. Gs) and lsmall[Gs .. Gs + Gqs); with W > 1 they ride
    // with the q exchange into gsmall[rank][Gs + Gqs].  pcg_cols and pcg_upd become ONE kernel, the p.Mp exchange disappears.
    int lean, Gq, Gqs, Gqr[BIG_MAXW];
    double *lsmall; const double *gsmall;
    BigState *st;                         // st[0], st[1]
};

// launch helpers (lpbox_big_kernels.hip); every state-carrying launch reads st[*parity], writes st[*parity^1], flips *parity
hipError_t big_launch_init(const BigDev &d, double c1, hipStream_t s);      // state, x = 1, partial b.x0 -> red[0]
hipError_t big_launch_init2(const BigDev &d, hipStream_t s);               // best_bin_obj = red[0] (after the all-reduce)
hipError_t big_launch_set_window(const BigDev &d, int iter_start, int iter_end, int l2f, int *parity, hipStream_t s);
hipError_t big_launch_fix1(const BigDev &d, hipStream_t s);                          // gsrc = newly fixed values, partial b.x2 (:1237)
hipError_t big_launch_fix2(const BigDev &d, int *parity, hipStream_t s);              // f -= E2 x2 (:1278), mask, partial |x_live|^2
hipError_t big_launch_fix3(const BigDev &d, long n_live_new, double c1_new, int *parity, hipStream_t s);   // state + update_expression (:1329)
hipError_t big_launch_pack_xiters(const BigDev &d, const int *live_idx, int rows, int ws, double *out, hipStream_t s);
hipError_t big_launch_prep(const BigDev &d, int do_prep, int *parity, hipStream_t s);
hipError_t big_launch_fin(const BigDev &d, int nv, int phase, hipStream_t s);         // partials of a phase -> red[0..nv)
hipError_t big_launch_y(const BigDev &d, int *parity, hipStream_t s);                 // y1, y2, refresh, rhs base, y3 (all rows)
hipError_t big_launch_rhs_cols(const BigDev &d, int *parity, hipStream_t s);          // rhs += r4Et E^T(f-y3) - E^T z4
hipError_t big_launch_rows(const BigDev &d, int mode, int *parity, hipStream_t s);    // q = E[:,shard] * v  (mode 0: gsrc, 1: PCG p)
hipError_t big_launch_resid(const BigDev &d, int *parity, hipStream_t s);
hipError_t big_launch_pcg_cols(const BigDev &d, int *parity, hipStream_t s);
hipError_t big_launch_pcg_upd(const BigDev &d, int *parity, hipStream_t s);
hipError_t big_launch_pcg_lean(const BigDev &d, int *parity, hipStream_t s);          // comm-lean mode: tmp = M p, alpha from p.p and q.q, x / r / z updates, partials (r.r, r.z)
hipError_t big_launch_rank_sum_qq(const double *g, int W, long count, long stride, double *out, double *qq_part, hipStream_t s);   // rank-ordered sum of a row block + q.q partials per 256 rows
hipError_t big_launch_post(const BigDev &d, int *parity, hipStream_t s);              // duals z1,z2, partials(5), gsrc = x
hipError_t big_launch_z4(const BigDev &d, int init_only, int *parity, hipStream_t s); // Ex = q [, z4 update]
hipError_t big_launch_resume(const BigDev &d, int reset_pcg_max, int *parity, hipStream_t s);
hipError_t big_launch_rank_sum(const double *g, int W, long count, long stride, double *out, hipStream_t s);   // rank-ordered sum of W gathered contributions
