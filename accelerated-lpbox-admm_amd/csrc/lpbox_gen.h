// lpbox_gen.h -- internal layout of the GENERIC constrained binary-QP path (the reference's ADMM_bqp, SEGcpp:1384-1832):
//     min x'Ax + b'x  s.t.  Cx = d,  Ex <= f,  x in {0,1}^n      (C and / or E optional)
// One problem per handle, one GPU.  Same construction as the large-instance LP path (lpbox_big.h): a chain of kernels cut at the
// grid-wide dependencies, control state on the device (ping-ponged GenState), fall-through launches, fixed two-level reduction
// tree.  The operator of the PCG is the reference's matrix expression (SEGcpp:361-411)
//     M v = (2A + (rho1+rho2) I) v  +  (rho3 C') (C v)  +  (rho4 E') (E v)
// with the scaled transposes kept as explicit value arrays (they are multiplied by learning_fact at every rho update, SEGcpp:1643,1648,
// so each entry carries its own rounding history).  Not part of the C-ABI.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#define GEN_T 256
#define GEN_NPART 8
#define GEN_HIST_MAX 8

enum { GEN_HALT_NONE = 0, GEN_HALT_STOP = 1, GEN_HALT_END = 2, GEN_HALT_PCG_MORE = 3 };
enum { GEN_STOP_NONE = 0, GEN_STOP_XYY = 1, GEN_STOP_OBJSTD = 2 };

struct GenParams {            // the hyper-parameters the *_init() presets set (SEGcpp:587-672)
    double stop_threshold, std_threshold, gamma_val, gamma_factor, initial_rho, learning_fact, pcg_tol;
    int rho_change_step, max_iters, history_size, pcg_maxiters;
};

struct GenState {
    double rho1, rho2, rho3, rho4, prev_rho1, prev_rho2, prev_rho3, prev_rho4, gamma_val, rcr;
    double std_obj, cur_obj, best_bin_obj, cvg1, cvg2, obj_val, c1;
    double hist[GEN_HIST_MAX];
    double threshold, absNew, rhsNorm2, beta;
    int rhoUpdated, hist_n, iter, have_prev, halt, stop, copy_best;
    int pcg_k, pcg_done, pcg_first, phase;
    int pcg_total, outer_total, last_pcg, pcg_max;
};

struct GenCsr { const int *ptr, *idx; const double *val; };

struct GenDev {
    int n, m, l, G, Gm, Gl, EPT, EPTm, EPTl;
    int eq, ineq;
    GenParams prm;
    // A (rows, ascending columns, every diagonal entry present): tmval = 2A + (rho1+rho2) I, adiag[i] = position of the diagonal entry
    const int *aptr, *aidx; const double *aval; double *tmval; const int *adiag;
    // C (m x n) by rows and by columns; E (l x n) likewise.  *_sv = the scaled transposes rho3 C' / rho4 E'
    GenCsr Cr, Cc, Er, Ec; double *Cc_sv, *Ec_sv; int Cnnz, Ennz;
    double *x, *xt, *y1, *y2, *z1, *z2, *rhs, *r, *z, *tmp, *p0, *p1, *gsrc, *pdiag, *dinv, *Csq, *Esq, *best;
    const double *b;
    double2 *zp;
    double *z3, *qC; const double *d;       // m-vectors
    double *y3, *z4, *fy, *qE, *Ex; const double *f;   // l-vectors
    double *part, *red;
    GenState *st;
};

size_t gen_state_bytes();
hipError_t gen_launch_init(const GenDev &d, double c1, const double *x0, hipStream_t s);                       // state, y1 = y2 = x = x0, partial cost(x0)
hipError_t gen_launch_init2(const GenDev &d, hipStream_t s);                                  // best_bin_obj from red[]
hipError_t gen_launch_fin(const GenDev &d, int nv, hipStream_t s);
hipError_t gen_launch_prep(const GenDev &d, int do_prep, int *parity, hipStream_t s);
hipError_t gen_launch_y(const GenDev &d, int *parity, hipStream_t s);
hipError_t gen_launch_rhs_cols(const GenDev &d, int *parity, hipStream_t s);
hipError_t gen_launch_rows(const GenDev &d, int mode, int *parity, hipStream_t s);            // qC = C v, qE = E v (mode 0: gsrc, 1: PCG p)
hipError_t gen_launch_resid(const GenDev &d, int *parity, hipStream_t s);
hipError_t gen_launch_pcg_cols(const GenDev &d, int *parity, hipStream_t s);
hipError_t gen_launch_pcg_upd(const GenDev &d, int *parity, hipStream_t s);
hipError_t gen_launch_post(const GenDev &d, int *parity, hipStream_t s);
hipError_t gen_launch_dual(const GenDev &d, int init_only, int *parity, hipStream_t s);
hipError_t gen_launch_resume(const GenDev &d, int reset_pcg_max, int *parity, hipStream_t s);
