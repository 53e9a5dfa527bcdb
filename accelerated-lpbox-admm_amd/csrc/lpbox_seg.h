// lpbox_seg.h -- internal device-side layout of the SEGMENTATION flavour (unconstrained BQP min x'Ax + b'x, x in {0,1}^n;
// SEGcpp = Segmentation/Segmentation/cython/src/LPboxADMMsolver.cpp).  Not part of the C-ABI.
//
// One instance has n ~ 1e4 .. 1e6 variables: far too large for one workgroup, so every vector lives in HBM (L2 /
// Infinity-Cache resident at these sizes) and one ADMM iteration is a short chain of kernels separated by the grid-wide
// dependencies of the algorithm (sparse product -> dot product -> update).  Workgroup g owns the CHUNK = T*EPT consecutive
// variables [g*CHUNK, (g+1)*CHUNK); reductions are two-level (block tree per workgroup -> G partials -> the same block tree
// over the partials, recomputed by every consumer workgroup), which the CPU oracle mirrors exactly.
// All control state is on the device (SegState, ping-ponged between kernels), so a window of iterations is a static
// sequence of launches that a hipGraph replays.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

// hyper-parameters, hard-coded as in ADMM_bqp_unconstrained_init (SEGcpp:659-672)
#define SEG_STD_THRESHOLD 1e-6                 // :659
#define SEG_GAMMA0        1.0                  // :660
#define SEG_GAMMA_FACTOR  0.99                 // :661
#define SEG_RHO0          5.0                  // :662
#define SEG_LEARNING_FACT (1 + 3.0 / 100)      // :663
#define SEG_HIST          5                    // :665 history_size
#define SEG_RHO_STEP      5                    // :666
#define SEG_STOP_THRESHOLD 1e-3                // :668
#define SEG_MAX_ITERS     10000                // :669
#define SEG_PCG_TOL       1e-3                 // :671
#define SEG_PCG_MAXITERS  1000                 // :672
#define SEG_XITERS_COLS   10                   // :924
#define SEG_REC_COLS      2000                 // default capacity (iterations) of a recorded legacy solve

#define SEG_T 256
#define SEG_NPART 8        // partial-sum slots per phase

enum { SEG_HALT_NONE = 0, SEG_HALT_STOP = 1, SEG_HALT_WINDOW = 2, SEG_HALT_PCG_MORE = 3, SEG_HALT_ALLFIXED = 4 };
enum { SEG_STOP_NONE = 0, SEG_STOP_XYY = 1, SEG_STOP_OBJSTD = 2, SEG_STOP_ALLFIXED = 4 };

struct SegState {
    // scalars of the reference object (SEGh)
    double rho1, rho2, prev_rho1, prev_rho2, gamma_val, rcr, std_obj, cur_obj, best_bin_obj, cvg1, cvg2, obj_val, c1;
    double hist[SEG_HIST];
    // PCG scalars carried between kernels
    double threshold, absNew, rhsNorm2;
    int rhoUpdated, hist_n, n_live;
    int iter;            // next outer iteration to run
    int iter_end;        // window end (exclusive)
    int phase;           // 0 = between iterations, 1 = inside an iteration (after prep), 2 = PCG running, 3 = PCG done
    int pcg_k;           // PCG iterations finished in the current outer iteration
    int pcg_done;
    int have_prev;       // an iteration's partE is waiting to be finalised
    int halt, stop, ret, l2f, cc;
    int rec;             // keep x of every iteration in xhist column cc (x_iters of the l2f loop; print_info 1 of the legacy loop)
    int pcg_total, outer_total, last_pcg, legacy_iter_p1;
    int pcg_max;         // largest PCG iteration count since the host last reset it (drives the adaptive launch count)
    int dinv_stale;      // a fix rebuilt the diagonal while no rho update was pending (stale preconditioner = UB in the reference)
};

struct SegDev {
    int n, nnz, G, EPT;                    // G workgroups of SEG_T threads, EPT slots each
    // A_ptr (row-major, ascending columns; SEGh:17) in ELL form: slot k of row i at [k*n + i], rowlen[i] slots used
    const int *ecol; const double *eval; const uint8_t *rowlen; int ell_w;
    // ... or, when the matrix has at most three diagonals either side of the main one and every off-diagonal value is -w with an
    // integer w in 0..255 (the image problems: offsets -ncols, -(ncols-1), -1, +1, ncols-1, ncols and w = round(3 exp(.)) in 0..3,
    // SEGcpp:144-224), as DIAGONALS: slot k of row i is column i + doff[k] (slot 3 = the main diagonal, value adiag[i]); the six
    // off-diagonal weights of a row are the low six bytes of dpack[i].  A slot whose entry does not exist has w = 0 and contributes
    // (-0.0) * v = +-0.0 to the row sum in the same ascending-column position -- the sum's bits are those of the stored entries alone.
    // 16 B of matrix per row instead of 84, and the gathers need no index loads.
    int dia; int doff[7]; const unsigned long long *dpack; const double *adiag;
    double *x, *y1, *y2, *z1, *z2, *b, *rhs, *r, *z, *tmp, *dinv, *td, *p0, *p1;
    uint8_t *live;          // 1 live, 0 fixed (x = 0 there; the fixed value is kept in fixval)
    uint8_t *fixval;
    const uint8_t *newfix;  // consumed by the fix kernel: 0 none, 1 fix to 0, 2 fix to 1
    double *part;           // [phase(5)][SEG_NPART][Gmax]
    double *xhist; int ws_cap;
    SegState *st;           // st[0], st[1] ping-pong
    double c1_init;         // pow(n, 1/2) for the batched init (SEGcpp:557,670)
};

hipError_t seg_launch_init(const SegDev &d, double c1, hipStream_t s);
// Every launch below reads st[*parity], writes st[*parity ^ 1] and flips *parity.
hipError_t seg_launch_set_window(const SegDev &d, int iter_start, int iter_end, int l2f, int *parity, hipStream_t s);
hipError_t seg_launch_fix(const SegDev &d, int n_live_new, double c1_new, int *parity, hipStream_t s);
// `iters` outer iterations, each = prep, yrhs, resid, kmax x (matvec, update), post  (an even number of launches)
hipError_t seg_enqueue_prep(const SegDev &d, int *parity, hipStream_t s);                  // head of a batch of iterations (1 launch)
hipError_t seg_enqueue_iterations(const SegDev &d, int iters, int kmax, int *parity, hipStream_t s);   // 3 + 2 kmax launches each
hipError_t seg_enqueue_pcg_more(const SegDev &d, int pairs, int *parity, hipStream_t s);   // resume a stalled PCG, then post
hipError_t seg_enqueue_finalize(const SegDev &d, int *parity, hipStream_t s);              // finalise the last iteration only
hipError_t seg_launch_copy(const SegDev &d, int reset_pcg_max, int *parity, hipStream_t s);   // state copy (parity flip), optionally pcg_max = 0
hipError_t seg_launch_pack_xiters(const SegDev &d, const int *live_idx, int rows, int ws, double *out, hipStream_t s);

// Batches of problems advanced in lockstep by ONE launch chain (devs = device array of B descriptors, grid (Gmax, B)): the kernels are
// the same bodies; every problem keeps its own control state, partial sums and iteration counts, so a problem whose PCG has converged
// or that has stopped falls through while the others go on.
hipError_t segb_launch_init(const SegDev *devs, int B, int Gmax, hipStream_t s);
hipError_t segb_launch_set_window(const SegDev *devs, int B, int iter_start, int iter_end, int mode, int *parity, hipStream_t s);
hipError_t segb_launch_copy(const SegDev *devs, int B, int reset_pcg_max, int *parity, hipStream_t s);
hipError_t segb_enqueue_iterations(const SegDev *devs, int B, int Gmax, int iters, int kmax, int *parity, hipStream_t s);
hipError_t segb_enqueue_pcg_more(const SegDev *devs, int B, int Gmax, int pairs, int *parity, hipStream_t s);
hipError_t segb_enqueue_finalize(const SegDev *devs, int B, int Gmax, int *parity, hipStream_t s);
hipError_t segb_collect_states(const SegDev *devs, int B, int parity, SegState *out, hipStream_t s);
