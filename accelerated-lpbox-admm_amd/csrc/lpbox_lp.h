// lpbox_lp.h -- internal device-side layout of a batch of LP instances (not part of the C-ABI).
//
// One 'instance' = one combinatorial-auction LP of the reference (LPcpp = LinerProgramming/LinearProgramming/
// cython_solver/LPboxADMMsolver.cpp).  A batch keeps every per-instance array in one pool with a common
// stride so that instance i, element j lives at pool[i*stride + j]:
//   n-vectors (stride NS): x, z1, z2, b, pd, live, newfix          (storage = position order, never compacted:
//                                                                   early fixing is a mask)
//   l-vectors (stride LS): z4, f
//   indices:   rs_ptr (LS+1) / rs_col (ZS)  rows of E  -> E*v   gathers
//              cs_ptr (NS+1) / cs_row (ZS)  cols of E  -> E^T*w gathers
//   scalars:   dsc[ND_*], isc[NI_*], hist[LP_HIST]
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

// ---- hyper-parameters, hard-coded as in ADMM_lp_iters_init (LPcpp:491-507) ----
#define LP_STOP_THRESHOLD 1e-4                  // :491
#define LP_STD_THRESHOLD  1e-12                 // :506
#define LP_GAMMA0         1.6                   // :493
#define LP_GAMMA_FACTOR   0.95                  // :494
#define LP_RHO_STEP       25                    // :495
#define LP_RHO0           25.0                  // :497
#define LP_LEARNING_FACT  (1 + 1.0 / 100)       // :499
#define LP_PCG_TOL        1e-3                  // :500
#define LP_PCG_MAXITERS   1000                  // :501
#define LP_HIST           10                    // :507 history_size
#define LP_XITERS_COLS    500                   // :1113

// double scalars per instance
enum {
    ND_RHO1 = 0, ND_RHO2, ND_RHO4, ND_PREV_RHO1, ND_PREV_RHO2, ND_PREV_RHO4,
    ND_GAMMA, ND_DI, ND_R4ET, ND_RCR, ND_STD_OBJ, ND_CUR_OBJ, ND_BEST_BIN_OBJ,
    ND_SUM_FIX_OBJ, ND_FIX_OBJ, ND_C1, ND_CVG1, ND_CVG2, ND_OBJ_VAL, ND_PREV_SUM, ND_PREV_OBJ,
    ND_COUNT = 24
};
// int scalars per instance
enum {
    NI_N = 0, NI_L, NI_NNZ, NI_NLIVE, NI_RHO_UPDATED, NI_ITER, NI_HIST_N, NI_RET, NI_STOP,
    NI_PCG_TOTAL, NI_OUTER_TOTAL, NI_LAST_PCG, NI_PLAIN_ITER_P1, NI_APPLY_FIX, NI_FIX_NUM, NI_ACTIVE,
    NI_EXPR_READY, NI_H_VALID,
    NI_COUNT = 24
};
// stop reasons
enum { LP_STOP_NONE = 0, LP_STOP_Y1Y2 = 1, LP_STOP_OBJSTD = 2, LP_STOP_PCG = 3, LP_STOP_ALLFIXED = 4 };

struct LpBatchDev {
    int B, NS, LS, ZS;
    // structure, in STORAGE order: variables sit at positions (columns of E sorted by decreasing length), row slots
    // likewise hold rows by decreasing length (rid[slot] = original row id; l-vectors stay indexed by original row id)
    const int *rs_ptr; const uint16_t *rs_col;   // row slot q: positions of its columns, ascending ORIGINAL column index
    const int *cs_ptr; const uint16_t *cs_row;   // position p: storage indices of the rows of its column's OWN part, ascending row id
    const int *hs_ptr;                           // position p: its helper chunk of the long column of its quad of lanes (entries in cs_row too)
    const uint16_t *cmeta;                       // position p: column length | 0x8000 if p owns its quad's split column
    const uint16_t *rid;      // row-task slot -> original row id (0xFFFF = no task)
    const uint16_t *rgl;      // row-task slot -> storage index of the row inside the gathered LDS l-vectors (bank-conflict aware)
    const uint16_t *rmeta;    // row-task slot -> (G << 4) | g: lane g of the G lanes sharing the row
    // state
    double *x, *z1, *z2, *b, *pd;
    uint8_t *live;          // 1 = live, 0 = fixed (x then holds the fixed value)
    const uint8_t *newfix;  // 0 = nothing, 1 = fix to 0, 2 = fix to 1 (consumed when NI_APPLY_FIX)
    double *z4, *f;
    double *dsc; int *isc; double *hist;
    // per-launch control written by the host: ctl[inst*4 + {0: apply fix, 1: fix_num, 2: n_live after the fix, 3: -}],
    // dctl[inst] = pow(n_live_after, 1/2) (the host's libm pow, as the reference computes it, LPcpp:427)
    const int *ctl; const double *dctl;
    // l2f iterate window: xhist[(inst*ws_cap + c)*NS + pos]
    double *xhist; int ws_cap;
    // opt-in DIRECT x-update (lpbox_set_x_update): H = (c I + E E^T)^-1 per instance, HL rows of pitch HLD, saved
    // here between launches (the kernel works on an LDS copy); nullptr / 0 in the default PCG mode
    double *H; int HL, HLD;       // per instance: HL x HLD doubles of H, then LS doubles of W (by row storage index)
    const int16_t *rdir;          // row-task slot -> dense index of its row among the G rows, -1 = D row
    const int *dng;               // G rows per instance
    // opt-in per-iteration log of the plain loop (lpbox_set_log; what the reference's does_log writes, LPcpp:1013-1067): LP_LOG_VALS doubles
    // per completed iteration of the call at logbuf[(inst * log_cap + (it - iter_start)) * LP_LOG_VALS]; nullptr = off
    double *logbuf; int log_cap;
    unsigned long long *stamps;   // diagnostic build only (LPBOX_STAMPS): 16 phase counters per instance, else nullptr
    int stamp_wave;               // ... of this wavefront of the workgroup (LPBOX_STAMP_WAVE, default 0)
};

// one log record: PCG iterations, |x|, |y1|, |y2|, |y3|, |z1|, |z2|, |z4|, b.x, b.round(x), device wall-clock ticks since the launch started, iteration
#define LP_LOG_VALS 12

// launchers (lpbox_lp_kernels.hip)
size_t lp_window_lds_bytes(int T, int NS, int LS, int ZS, int HL = 0, int HLD = 0);   // HL > 0: with the direct mode's dense inverse
hipError_t lp_launch_init(const LpBatchDev &bd, int T, int EPT, const double *f_org, const double *c1_init,
                          const uint8_t *live_init, hipStream_t s);
hipError_t lp_launch_window(const LpBatchDev &bd, int T, int EPT, size_t lds, int iter_start, int iter_end, int l2f,
                            hipStream_t s, bool direct = false, bool log = false);
bool lp_log_supported(int T, int EPT);         // geometries the logging variant is compiled for (the default ones: 512 threads)
bool lp_direct_supported(int T, int EPT);      // geometries the DIRECT variant is compiled for
hipError_t lp_launch_pack_xiters(const LpBatchDev &bd, const int *live_pos, const int *rows, int ws, double *out,
                                 long out_stride, hipStream_t s);
