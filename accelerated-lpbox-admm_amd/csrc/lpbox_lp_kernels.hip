// lpbox_lp_kernels.hip -- hand-written gfx950 kernels for the LP flavour of the Lp-Box ADMM inner solver.
//
// Design (MI355X-first):
//   * one workgroup == one LP instance, persistent over a whole window of ADMM iterations (one launch runs
//     iterations [iter_start, iter_end) of every instance of the batch; 256 instances fill the 256 CUs);
//   * every n- and l-vector of the algorithm lives in REGISTERS of the thread that owns the element; only the
//     vectors other threads must gather (p / x for E*v; E*v, f-y3, z4 for E^T*w) are staged in LDS.
//     HBM is touched once per launch (state in / state out) plus the x_iters window of the l2f path;
//   * variables are stored in POSITION order (columns of E sorted by decreasing length, rows likewise) so that
//     the 64 lanes of a wavefront walk gather lists of equal length; each thread keeps the LDS byte offsets
//     of its own row / column in registers (loaded once per launch), so a sparse product is a burst of
//     independent ds_read_b64 gathers followed by the additions in ascending index order -- the order Eigen's
//     column-major product uses (LPcpp:102-108).  Lists are padded with the offset of a zero slot: adding +0.0
//     is exact, so padding never changes a sum;
//   * all dot products / norms use one fixed reduction tree: per-thread slot sums -> 64-lane xor butterfly on
//     DPP + v_permlane{16,32}_swap -> wave partials through LDS, added in wave order.  The CPU oracle reproduces
//     exactly this association (oracle/lpbox_oracle.c, LPO_ORDER_GPU) which makes the kernel bit-comparable to it;
//   * early fixing is a mask: a fixed variable keeps its register slot, contributes +0.0 to every gather and
//     reduction by predication, and the x-update commits `live ? x_new : x_fixed` (branchless).
//
// Reference citations: LPcpp = LinerProgramming/LinearProgramming/cython_solver/LPboxADMMsolver.cpp.
// Built with -ffp-contract=off: the reference is compiled without FMA (plain g++ -O3 on x86-64).
#include "lpbox_lp.h"
#include "lpbox_dev_common.h"

#include <float.h>

#include <utility>

namespace {

// Diagnostic build only (-DLPBOX_STAMPS, liblpbox_hip_stamps.so): per-phase cycle shares of wave 0, written to
// bd.stamps[inst*16 + phase].  The shipped library contains no stamp.
#ifdef LPBOX_STAMPS
#define STAMP_DECL unsigned long long st_acc[16] = {0}; unsigned long long st_t = __builtin_amdgcn_s_memtime();
#define STAMP(k) { __builtin_amdgcn_sched_barrier(0); unsigned long long st_n = __builtin_amdgcn_s_memtime(); \
                   __builtin_amdgcn_s_waitcnt(0xC07F); st_acc[k] += st_n - st_t; st_t = st_n; __builtin_amdgcn_sched_barrier(0); }
#define STAMP_STORE if (tid == 64 * bd.stamp_wave && bd.stamps) { for (int k = 0; k < 16; k++) bd.stamps[(size_t)inst * 16 + k] = st_acc[k]; }
#else
#define STAMP_DECL
#define STAMP(k)
#define STAMP_STORE
#endif
// -DLPBOX_STAMPS_PREBAR (with LPBOX_STAMPS): the two LDS hand-overs of a PCG iteration are stamped on BOTH sides of their barrier --
// slot 12 = "p -> LDS" up to the barrier, 3 = waiting at it; 13 = "q -> LDS", 5 = waiting -- and the post-PCG phases F, G, H share slot 14
#if defined(LPBOX_STAMPS) && defined(LPBOX_STAMPS_PREBAR)
#define STAMP_PRE(k) STAMP(k)
#define STAMP_POST(k) STAMP(14)
#else
#define STAMP_PRE(k)
#define STAMP_POST(k) STAMP(k)
#endif

// ------------------------------------------------------------------------------------------------
// LDS carve-up shared by the launcher (size) and the kernel (pointers)
// ------------------------------------------------------------------------------------------------
struct LdsLayout {
    size_t gx, gl, red, hist, rowf, rowy3, rs_ptr, cs_ptr, hs_ptr, rs_col, cs_row, H, dcol, drow, dmap, total;
    __host__ __device__ LdsLayout(int NS, int LS, int ZS, bool lean, int HL = 0, int HLD = 0) {
        size_t o = 0;
        auto take = [&](size_t bytes) { size_t at = o; o += (bytes + 15) & ~size_t(15); return at; };
        gx = take(sizeof(double) * (2 * (size_t)NS + 3)); // n-vector being gathered + zero slot at [NS]; a second one (the PCG start vector) at [NS + 2 ..], zero slot at [2 NS + 2]
        gl = take(sizeof(double) * 3 * ((size_t)LS + 1)); // three l-vectors interleaved: gl[3*i + c]; zero slot at i = LS
        red = take(sizeof(double) * 2 * RED_MAXV * RED_MAXW);
        hist = take(sizeof(double) * 2 * LP_HIST);        // objective history, double-buffered (read by all, rewritten by thread 0)
        rowf = take(lean ? sizeof(double) * ((size_t)LS + 1) : 0);     // multi-slot variants: f and y3 by row live here, not in registers
        rowy3 = take(lean ? sizeof(double) * ((size_t)LS + 1) : 0);
        rs_ptr = take(sizeof(int) * ((size_t)NS + 1));
        cs_ptr = take(sizeof(int) * ((size_t)NS + 1));
        hs_ptr = take(sizeof(int) * ((size_t)NS + 1));
        rs_col = take(sizeof(uint16_t) * (size_t)ZS);
        cs_row = take(sizeof(uint16_t) * (size_t)ZS);
        // direct x-update only (HL > 0): the dense inverse and the pivot column / row of its Gauss-Jordan build
        H = take(sizeof(double) * (size_t)HL * (size_t)HLD);
        dcol = take(HL ? sizeof(double) * ((size_t)HL + 1) : 0);
        drow = take(HL ? sizeof(double) * ((size_t)HL + 1) : 0);
        dmap = take(HL ? sizeof(int) * ((size_t)HL + 1) : 0);     // dense index of a G row -> its storage index
        total = o;
    }
};

// ------------------------------------------------------------------------------------------------
// gather lists: CAP absolute LDS addresses in registers (+ an LDS index tail for longer lists)
// ------------------------------------------------------------------------------------------------
// register-lean variants (cold vectors in memory / LDS): the multi-slot ones and every 16-wave geometry (128 registers per lane)
__host__ __device__ constexpr bool lp_is_lean(int T, int EPT) { return EPT >= 4 || T == 1024; }

typedef __attribute__((address_space(3))) double lds_double;

__device__ __forceinline__ unsigned lds_addr(const void *p) {
    return (unsigned)(size_t)(__attribute__((address_space(3))) const void *)p;
}
// component COMP (an immediate ds_read offset) of the element at LDS address a
template <int COMP>
__device__ __forceinline__ double lds_ld(unsigned a) {
    return ((const lds_double *)(size_t)a)[COMP];
}

// per-slot register capacities of the gather lists of one thread (slot s keeps its first at(s) entries in registers)
template <int... V>
struct Caps {
    static constexpr int N = sizeof...(V);
    static constexpr int total = (V + ... + 0);
    static constexpr int at(int i) { constexpr int a[N] = {V...}; return a[i]; }
    static constexpr int off(int i) { constexpr int a[N] = {V...}; int o = 0; for (int k = 0; k < i; k++) o += a[k]; return o; }
};

template <typename F, int... I>
__device__ __forceinline__ void static_for_impl(F &&f, std::integer_sequence<int, I...>) {
    (f(std::integral_constant<int, I>{}), ...);
}
template <int N, typename F>
__device__ __forceinline__ void static_for(F &&f) {
    static_for_impl(f, std::make_integer_sequence<int, N>{});
}

template <typename C>
struct Lists {
    unsigned addr[C::total > 0 ? C::total : 1];   // LDS address of each gathered element, padded with the zero slot's address
    const int *ptr;                               // LDS copy of the list pointers (entries beyond the register capacity are re-read from there)
    int pos0, pstride;                            // this thread's list of slot s is ptr[pos0 + s * pstride] .. ptr[pos0 + s * pstride + 1]
    int tail_any[C::N];                           // wave-uniform: some lane of the wave has entries beyond the register capacity
    // single/double-slot variants keep the tail bounds in registers (re-reading them costs the 512 x 1 kernel 3 %); the multi-slot
    // variants, short of registers, re-read them from the LDS pointer arrays in the rare wave that has a tail
    static constexpr bool TAILREG = C::N < 4;
    int tail_begin[TAILREG ? C::N : 1], tail_end[TAILREG ? C::N : 1];
    int wlen[C::N];                               // wave-uniform number of register entries to walk
    unsigned base, zero;                          // address of element 0 / of the zero slot
};

// STRIDE = bytes between consecutive elements of the gathered vector (8 for gx, 24 for the interleaved gl)
template <typename C, int S, int STRIDE>
__device__ __forceinline__ void build_list(Lists<C> &g, int begin, int end, const uint16_t *idx) {
    constexpr int CAP = C::at(S), OFF = C::off(S);
#pragma unroll
    for (int k = 0; k < CAP; k++) g.addr[OFF + k] = (begin + k < end) ? g.base + STRIDE * (unsigned)idx[begin + k] : g.zero;
    const int len = end - begin;
    g.tail_any[S] = wave_max_int(len > CAP ? 1 : 0);
    if constexpr (Lists<C>::TAILREG) { g.tail_begin[S] = begin + CAP < end ? begin + CAP : end; g.tail_end[S] = end; }
    g.wlen[S] = wave_max_int(len < CAP ? len : CAP);
}

// Make every register address of a list live at one point.  In the register-lean variants the allocator parks a few list addresses in
// scratch during the outer phases and, left alone, fetches them back one at a time right in front of their LDS gather -- a scratch round
// trip per entry inside a dependent chain.  Touching them together turns that into one batch of loads ahead of the gather.
template <typename C>
__device__ __forceinline__ void touch_list(const Lists<C> &g) {
#ifndef LPBOX_NO_TOUCH
#pragma unroll
    for (int k = 0; k < C::total; k++) { const unsigned v = g.addr[k]; asm volatile("" : : "v"(v)); }
#endif
}

constexpr int GCH = 4;   // gathers of one slot issued together before their (ordered) additions

template <typename C>
constexpr int caps_max() { int m = 0; for (int i = 0; i < C::N; i++) m = C::at(i) > m ? C::at(i) : m; return m; }

// dispatch a wave-uniform chunk count nch in [0, MAXCH] to a body instantiated for that count (a straight-line path per
// count: every LDS gather of the list is issued before the first addition, so the list costs ONE LDS latency, not one per chunk)
template <int MAXCH, typename F>
__device__ __forceinline__ void dispatch_chunks(int nch, F &&f) {
    static_for<MAXCH + 1>([&](auto K) {
        constexpr int k = decltype(K)::value;
        if (nch == k) f(K);
    });
}

// out[s] = sum_k src[list_s[k]] (k ascending) for every slot s of the thread, through OP(acc, v) which keeps the exact
// expression of the reference: `acc + v` for E (stored values 1.0) or `acc + s * v` for the scaled transpose.
// One slot per thread: all gathers of the (wave-uniform) list length first, then the additions in order.
// Several slots: chunk-major -- round c issues the c-th GCH gathers of EVERY slot before any addition, so EPT*GCH independent
// LDS reads are in flight (the slots are independent sums; inside a slot the additions stay in ascending order).
template <typename C, int STRIDE, int COMP, typename OP>
__device__ __forceinline__ void gather_all(const Lists<C> &g, const uint16_t *idx, OP op, double (&out)[C::N]) {
    constexpr int N = C::N, MAXCAP = caps_max<C>();
    double acc[N];
#pragma unroll
    for (int s = 0; s < N; s++) acc[s] = 0.0;
    if constexpr (N == 1) {
#ifndef LPBOX_LP_G1
#define LPBOX_LP_G1 2
#endif
        constexpr int G1 = LPBOX_LP_G1;                                // granularity of the wave-uniform list length (a straight-line path per multiple)
        int nch_ = (g.wlen[0] + G1 - 1) / G1;
#ifdef LPBOX_KO_COLCAP
        if (STRIDE == 24 && nch_ > LPBOX_KO_COLCAP) nch_ = LPBOX_KO_COLCAP;
#endif
#ifdef LPBOX_KO_ROWCAP
        if (STRIDE == 8 && nch_ > LPBOX_KO_ROWCAP) nch_ = LPBOX_KO_ROWCAP;
#endif
        dispatch_chunks<MAXCAP / G1>(nch_, [&](auto K) {
            constexpr int n = decltype(K)::value * G1;
            double v[n > 0 ? n : 1];
#pragma unroll
            for (int q = 0; q < n; q++) v[q] = lds_ld<COMP>(g.addr[q]);
#pragma unroll
            for (int q = 0; q < n; q++) acc[0] = op(acc[0], v[q]);
        });
    } else {
    static_for<MAXCAP / GCH>([&](auto CI) {
        constexpr int c = decltype(CI)::value * GCH;
        double v[N][GCH];
        static_for<N>([&](auto S) {
            constexpr int s = decltype(S)::value;
            if constexpr (c < C::at(s)) {
                if (c < g.wlen[s]) {                                    // wave-uniform
#pragma unroll
                    for (int q = 0; q < GCH; q++) v[s][q] = lds_ld<COMP>(g.addr[C::off(s) + c + q]);
                }
            }
        });
        static_for<N>([&](auto S) {
            constexpr int s = decltype(S)::value;
            if constexpr (c < C::at(s)) {
                if (c < g.wlen[s]) {
#pragma unroll
                    for (int q = 0; q < GCH; q++) acc[s] = op(acc[s], v[s][q]);
                }
            }
        });
    });
    }
    static_for<N>([&](auto S) {                                         // long lists: indices from LDS
        constexpr int s = decltype(S)::value;
        int tb = 0, te = 0;
        if constexpr (Lists<C>::TAILREG) { tb = g.tail_begin[s]; te = g.tail_end[s]; }
        else if (g.tail_any[s]) {                                       // rare: the list pointers come back from LDS, not from registers
            const int lb = g.ptr[g.pos0 + s * g.pstride];
            te = g.ptr[g.pos0 + s * g.pstride + 1];
            tb = lb + C::at(s) < te ? lb + C::at(s) : te;
        }
        for (int k = tb; k < te; k += GCH) {
            unsigned o[GCH];
#pragma unroll
            for (int q = 0; q < GCH; q++) o[q] = (k + q < te) ? g.base + STRIDE * (unsigned)idx[k + q] : g.zero;
            double v[GCH];
#pragma unroll
            for (int q = 0; q < GCH; q++) v[q] = lds_ld<COMP>(o[q]);
#pragma unroll
            for (int q = 0; q < GCH; q++) acc[s] = op(acc[s], v[q]);
        }
        out[s] = acc[s];
    });
}

// two sums per slot from components CA and CB of the gathered elements (rhs assembly)
template <typename C, int STRIDE, int CA, int CB, typename OPA, typename OPB>
__device__ __forceinline__ void gather_all2(const Lists<C> &g, const uint16_t *idx, OPA opa, OPB opb, double (&outA)[C::N],
                                            double (&outB)[C::N]) {
    constexpr int N = C::N, MAXCAP = caps_max<C>();
    double a[N], b[N];
#pragma unroll
    for (int s = 0; s < N; s++) { a[s] = 0.0; b[s] = 0.0; }
    static_for<MAXCAP / GCH>([&](auto CI) {
        constexpr int c = decltype(CI)::value * GCH;
        double va[N][GCH], vb[N][GCH];
        static_for<N>([&](auto S) {
            constexpr int s = decltype(S)::value;
            if constexpr (c < C::at(s)) {
                if (c < g.wlen[s]) {
#pragma unroll
                    for (int q = 0; q < GCH; q++) {
                        va[s][q] = lds_ld<CA>(g.addr[C::off(s) + c + q]);
                        vb[s][q] = lds_ld<CB>(g.addr[C::off(s) + c + q]);
                    }
                }
            }
        });
        static_for<N>([&](auto S) {
            constexpr int s = decltype(S)::value;
            if constexpr (c < C::at(s)) {
                if (c < g.wlen[s]) {
#pragma unroll
                    for (int q = 0; q < GCH; q++) { a[s] = opa(a[s], va[s][q]); b[s] = opb(b[s], vb[s][q]); }
                }
            }
        });
    });
    static_for<N>([&](auto S) {
        constexpr int s = decltype(S)::value;
        int tb = 0, te = 0;
        if constexpr (Lists<C>::TAILREG) { tb = g.tail_begin[s]; te = g.tail_end[s]; }
        else if (g.tail_any[s]) {                                       // rare: the list pointers come back from LDS, not from registers
            const int lb = g.ptr[g.pos0 + s * g.pstride];
            te = g.ptr[g.pos0 + s * g.pstride + 1];
            tb = lb + C::at(s) < te ? lb + C::at(s) : te;
        }
        for (int k = tb; k < te; k += GCH) {
            unsigned o[GCH];
#pragma unroll
            for (int q = 0; q < GCH; q++) o[q] = (k + q < te) ? g.base + STRIDE * (unsigned)idx[k + q] : g.zero;
            double va[GCH], vb[GCH];
#pragma unroll
            for (int q = 0; q < GCH; q++) { va[q] = lds_ld<CA>(o[q]); vb[q] = lds_ld<CB>(o[q]); }
#pragma unroll
            for (int q = 0; q < GCH; q++) { a[s] = opa(a[s], va[q]); b[s] = opb(b[s], vb[q]); }
        }
        outA[s] = a[s]; outB[s] = b[s];
    });
}

// ------------------------------------------------------------------------------------------------
// ADMM_lp_iters_init (LPcpp:489-763) for every instance: x=1, z=0, rho=25, ...
// ------------------------------------------------------------------------------------------------
template <int T, int EPT>
__global__ void __launch_bounds__(T) lp_init_kernel(LpBatchDev bd, const double *f_org, const double *c1_init,
                                                    const uint8_t *live_init) {
    __shared__ double red[2 * RED_MAXV * RED_MAXW];
    const int inst = blockIdx.x, tid = threadIdx.x;
    int *isc = bd.isc + (size_t)inst * NI_COUNT;
    double *dsc = bd.dsc + (size_t)inst * ND_COUNT;
    const int l = isc[NI_L];
    const size_t on = (size_t)inst * bd.NS, ol = (size_t)inst * bd.LS;
    int parity = 0;
    double part[1] = {0.0};
#pragma unroll
    for (int s = 0; s < EPT; s++) {
        const int pos = s * T + tid;
        const bool isvar = live_init[on + pos] != 0;   // storage positions without a variable ("holes") behave like fixed zeros
        bd.x[on + pos] = isvar ? 1.0 : 0.0;            // :583-586
        bd.z1[on + pos] = 0.0; bd.z2[on + pos] = 0.0;  // :616-617
        bd.pd[on + pos] = 0.0;
        bd.live[on + pos] = isvar ? 1 : 0;
        part[0] = part[0] + (isvar ? bd.b[on + pos] * 1.0 : 0.0);   // best_bin_obj = b.dot(x0), :727
    }
    for (int i = tid; i < l; i += T) { bd.z4[ol + i] = 0.0; bd.f[ol + i] = f_org[ol + i]; }   // :650
    block_sum<T, 1>(part, red, parity);
    if (tid == 0) {
        dsc[ND_RHO1] = LP_RHO0; dsc[ND_RHO2] = LP_RHO0; dsc[ND_RHO4] = LP_RHO0;             // :623-630
        dsc[ND_PREV_RHO1] = LP_RHO0; dsc[ND_PREV_RHO2] = LP_RHO0; dsc[ND_PREV_RHO4] = LP_RHO0;
        dsc[ND_GAMMA] = LP_GAMMA0;
        dsc[ND_DI] = 0.0; dsc[ND_R4ET] = 0.0; dsc[ND_RCR] = 0.0;
        dsc[ND_STD_OBJ] = 1.0;                       // LPh:219
        dsc[ND_CUR_OBJ] = 0.0;                       // LPh:213
        dsc[ND_BEST_BIN_OBJ] = part[0];
        dsc[ND_SUM_FIX_OBJ] = 0.0; dsc[ND_FIX_OBJ] = 0.0;   // :593-594
        dsc[ND_C1] = c1_init[inst];                  // pow(n, 1/p), p = 2 (:427,:503)
        dsc[ND_CVG1] = 0.0; dsc[ND_CVG2] = 0.0; dsc[ND_OBJ_VAL] = 0.0;
        dsc[ND_PREV_SUM] = 0.0; dsc[ND_PREV_OBJ] = 0.0;
        isc[NI_NLIVE] = isc[NI_N];
        isc[NI_RHO_UPDATED] = 1;                     // LPh:214
        isc[NI_ITER] = 0; isc[NI_HIST_N] = 0; isc[NI_RET] = 0; isc[NI_STOP] = 0;
        isc[NI_PCG_TOTAL] = 0; isc[NI_OUTER_TOTAL] = 0; isc[NI_LAST_PCG] = 0; isc[NI_PLAIN_ITER_P1] = 0;
        isc[NI_EXPR_READY] = 0; isc[NI_H_VALID] = 0;
        for (int k = 0; k < LP_HIST; k++) bd.hist[(size_t)inst * LP_HIST + k] = 0.0;
    }
}

// ------------------------------------------------------------------------------------------------
// The ADMM window: iterations [iter_start, iter_end) of ADMM_lp_iters (LPcpp:766-1095, l2f == 0) or
// ADMM_lp_iters_l2f (LPcpp:1098-1574, l2f == 1) for one instance per workgroup.
//   T    threads per instance, EPT variable (and row) slots per thread,
//   RCAPS / CCAPS  per-slot register capacities (Caps<...>) of the row-task / column gather lists.
// ------------------------------------------------------------------------------------------------
// a wave-uniform double moved to scalar registers (the compiler cannot prove uniformity of values that came out of an LDS reduction)
__device__ __forceinline__ double uniform_f64(double v) {
    return __hiloint2double(__builtin_amdgcn_readfirstlane(__double2hiint(v)), __builtin_amdgcn_readfirstlane(__double2loint(v)));
}

// One value per slot of the thread.  MODE 0: in registers for the whole launch.  The register-lean variants keep the vectors the PCG
// loop never reads out of the register file WHILE THAT LOOP RUNS: MODE 1 leaves the vector in its HBM/L2 pool and touches it at its
// (rare) use sites (pd: read and written only when rho changes); MODE 2 holds it in registers outside the PCG loop -- park() stores it
// right before the loop, unpark() brings it back right after, all slots in one batch of loads -- so the outer phases see registers
// (z1, z2, b: a load issued behind a store to a possibly aliasing address has to wait for the store on this target, which made
// every get-after-set of the memory-backed form a full memory round trip: 8 of them per outer iteration in the dual update alone).
// Every thread only ever touches its own elements, so no synchronisation is involved.
template <int MODE, int N, int STRIDE>
struct SlotVec {
    double r[MODE == 1 ? 1 : N];
    // element of slot s = *(ub + boff + s * STRIDE * 8 bytes): ONE wave-uniform base (scalar registers) and ONE 32-bit BYTE offset per
    // thread -- the `global_load v, v_off, s[base]` form.  (A per-thread 64-bit pointer per vector and slot -- what `base + tid`
    // compiles to, the slot stride being beyond the 12-bit immediate offset of a global load -- cost the four-slot variant 32 registers,
    // which it kept in scratch: every access of a cold vector was a scratch reload of its address followed by the dependent global
    // access, 12 such pairs one after the other when the parked vectors came back after the PCG.)
    char *ub;
    unsigned boff;
    __device__ __forceinline__ double *elem(int s) const {
        unsigned o = boff;
        asm volatile("" : "+v"(o));      // opaque: keeps the compiler from hoisting sixteen loop-invariant 64-bit addresses out of the ADMM loop (and spilling them)
        return (double *)(ub + (o + (unsigned)(s * STRIDE * 8)));
    }
    __device__ __forceinline__ void load(double *uniform_base, unsigned thread_idx) {
        ub = (char *)uniform_base; boff = thread_idx * 8u;
        if constexpr (MODE != 1) {
#pragma unroll
            for (int s = 0; s < N; s++) r[s] = *elem(s);
        }
    }
    __device__ __forceinline__ double get(int s) const { if constexpr (MODE == 1) return *elem(s); else return r[s]; }
    __device__ __forceinline__ void set(int s, double v) { if constexpr (MODE == 1) *elem(s) = v; else r[s] = v; }
    __device__ __forceinline__ void park(bool dirty = true) {          // MODE 2: registers -> memory (nothing to write for a vector that never changes)
        if constexpr (MODE == 2) {
            if (dirty) {
#pragma unroll
                for (int s = 0; s < N; s++) *elem(s) = r[s];
            }
        }
    }
    __device__ __forceinline__ void unpark() {                         // MODE 2: memory -> registers
        if constexpr (MODE == 2) {
#pragma unroll
            for (int s = 0; s < N; s++) r[s] = *elem(s);
        }
    }
    __device__ __forceinline__ void store() const {
        if constexpr (MODE != 1) {
#pragma unroll
            for (int s = 0; s < N; s++) *elem(s) = r[s];
        }
    }
};

// One value per row task of the thread (owned by the task's leader lane), in registers or -- multi-slot variants -- in an LDS vector
// indexed like the gathered l-vectors (element idx at lds[idx * STRIDE]); only the leader lane ever touches its element.
template <bool MEM, int N, int STRIDE>
struct RowVec {
    double r[MEM ? 1 : N];
    double *lds;
    // (MEM: the element's address is formed at the access.  Left to itself the compiler hoists the loop-invariant addresses of all slots
    //  out of the iteration loop, runs out of registers, parks them in scratch and fetches each one back right in front of its LDS access:
    //  a scratch round trip inside every dependent y3 / z4 chain of the outer phases -- an and + shift is cheaper.)
    static __device__ __forceinline__ int at(int idx) {
#ifndef LPBOX_NO_OPAQUE_ROWVEC
        asm volatile("" : "+v"(idx));
#endif
        return idx * STRIDE;
    }
    __device__ __forceinline__ double get(int s, int idx, bool leader) const {
        if constexpr (MEM) return leader ? lds[at(idx)] : 0.0; else return r[s];
    }
    __device__ __forceinline__ void set(int s, int idx, bool leader, double v) {
        if constexpr (MEM) { if (leader) lds[at(idx)] = v; } else r[s] = v;
    }
};

// LOG: the plain loop also leaves the reference's per-iteration log values (does_log, LPcpp:1013-1067: six vector norms that nothing else
// needs) in bd.logbuf -- an instantiation of its own, so the default kernel's code is untouched.
template <int T, int EPT, typename RCAPS, typename CCAPS, typename HCAPS, bool DIRECT = false, bool LOG = false>
__global__ void __launch_bounds__(T) lp_window_kernel(LpBatchDev bd, int iter_start, int iter_end, int mode) {
    const int l2f = mode & 1, rec = mode & 2;     // rec: keep x after every iteration in xhist (x_iters of the l2f loop; print_fix_info 2/3 of the plain loop)
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int inst = blockIdx.x, tid = threadIdx.x;
    int *isc = bd.isc + (size_t)inst * NI_COUNT;
    double *dsc = bd.dsc + (size_t)inst * ND_COUNT;
    if (!isc[NI_ACTIVE]) return;

    const int nnz = isc[NI_NNZ];
    const size_t on = (size_t)inst * bd.NS, ol = (size_t)inst * bd.LS, oz = (size_t)inst * bd.ZS;

    static_assert(!DIRECT || EPT == 1, "the direct x-update is built for the one-slot variants");
    const LdsLayout L(bd.NS, bd.LS, bd.ZS, lp_is_lean(T, EPT), DIRECT ? bd.HL : 0, DIRECT ? bd.HLD : 0);
    double *gx = (double *)(smem + L.gx);
    double *gl = (double *)(smem + L.gl);       // gl[3*i + c]: c = 0: q = E*p / f - y3, 1: z4, 2: E*y1
    double *red = (double *)(smem + L.red);
    double *s_hist = (double *)(smem + L.hist);
    int *s_rs_ptr = (int *)(smem + L.rs_ptr);
    int *s_cs_ptr = (int *)(smem + L.cs_ptr);
    int *s_hs_ptr = (int *)(smem + L.hs_ptr);
    uint16_t *s_rs_col = (uint16_t *)(smem + L.rs_col);
    uint16_t *s_cs_row = (uint16_t *)(smem + L.cs_row);

    // ---- stage the index sets of E into LDS ----
    {
        const int *gp = bd.rs_ptr + (size_t)inst * (bd.NS + 1);
        const int *gc = bd.cs_ptr + (size_t)inst * (bd.NS + 1);
        const int *gh = bd.hs_ptr + (size_t)inst * (bd.NS + 1);
        for (int i = tid; i <= bd.NS; i += T) { s_rs_ptr[i] = gp[i]; s_cs_ptr[i] = gc[i]; s_hs_ptr[i] = gh[i]; }
        const uint16_t *c0 = bd.rs_col + oz, *r0 = bd.cs_row + oz;
        for (int k = tid; k < nnz; k += T) { s_rs_col[k] = c0[k]; s_cs_row[k] = r0[k]; }
        if (tid == 0) { gx[bd.NS] = 0.0; gx[2 * bd.NS + 2] = 0.0; gl[3 * bd.LS] = 0.0; gl[3 * bd.LS + 1] = 0.0; gl[3 * bd.LS + 2] = 0.0; }   // zero slots
    }
    // ---- direct x-update: the dense inverse H, kept in LDS for the launch (section 17 of DESIGN.md) ----
    double *Hm = (double *)(smem + L.H), *dcol = (double *)(smem + L.dcol), *drow = (double *)(smem + L.drow);
    int *dmap = (int *)(smem + L.dmap);
    int h_valid = 0;
    if constexpr (DIRECT) {
        h_valid = isc[NI_H_VALID];
        const size_t hsz = (size_t)bd.HL * bd.HLD;
        if (h_valid) { const double *Hg = bd.H + (size_t)inst * (hsz + bd.LS); for (size_t e = tid; e < hsz; e += T) Hm[e] = Hg[e]; }
    }

    // ---- per-thread state ----
    constexpr bool LEAN = lp_is_lean(T, EPT);    // multi-slot variants: z1, z2, b, pd stay in memory, y1 / y2 are recomputed after the PCG
    double x[EPT], dinv[EPT];
    SlotVec<LEAN ? 2 : 0, EPT, T> z1, z2, b;
    SlotVec<LEAN ? 1 : 0, EPT, T> pd;
    z1.load(bd.z1 + on, (unsigned)tid); z2.load(bd.z2 + on, (unsigned)tid); b.load(bd.b + on, (unsigned)tid); pd.load(bd.pd + on, (unsigned)tid);
    // Esq_diag_j = sum of squared entries of column j (LPcpp:2378-2390) = its length (entries are 1.0): needed only when the
    // diagonal is rebuilt (iteration 0, a fix, a rho update), so it is re-read from the layout instead of living in a register
    double Esq_reg[LEAN ? 1 : EPT];
    if constexpr (!LEAN) {
#pragma unroll
        for (int s = 0; s < EPT; s++) Esq_reg[s] = (double)(bd.cmeta[on + s * T + tid] & 0x7FFF);
    }
    auto Esq = [&](int s) { if constexpr (LEAN) return (double)(bd.cmeta[on + s * T + tid] & 0x7FFF); else return Esq_reg[s]; };
    bool live[EPT];
    double Ex[EPT];
    RowVec<LEAN, EPT, 3> z4;   // multi-slot variants: component 1 of the gathered l-vectors IS z4
    RowVec<LEAN, EPT, 1> f, y3;
    z4.lds = gl + 1; f.lds = (double *)(smem + L.rowf); y3.lds = (double *)(smem + L.rowy3);
    // row task of slot s, packed: bits 0-15 storage index of the row in the gathered LDS l-vectors, 16-19 lanes sharing the row (1,2,4,8),
    // bit 20 this lane is the LEADER (lane 0) of the task: it owns the row's l-vector entries
    // (single-slot variants keep the three fields in registers of their own: the unpacking would sit in the PCG loop)
    int rinfo[EPT], rGreg[LEAN ? 1 : EPT];
    bool rvreg[LEAN ? 1 : EPT];
    auto rgl = [&](int s) { return rinfo[s] & 0xFFFF; };
    auto rG = [&](int s) { if constexpr (LEAN) return (rinfo[s] >> 16) & 15; else return rGreg[s]; };
    auto rvalid = [&](int s) { if constexpr (LEAN) return (rinfo[s] & (1 << 20)) != 0; else return rvreg[s]; };
#pragma unroll
    for (int s = 0; s < EPT; s++) {
        const int pos = s * T + tid;
        x[s] = bd.x[on + pos];
        live[s] = bd.live[on + pos] != 0;
        const int rid = bd.rid[on + pos], meta = bd.rmeta[on + pos];
        const bool rtask = rid != 0xFFFF, leader = rtask && (meta & 15) == 0;
        rinfo[s] = rtask ? ((int)bd.rgl[on + pos] | ((meta >> 4) << 16) | (leader ? 1 << 20 : 0)) : (1 << 16);
        if constexpr (!LEAN) { rinfo[s] &= 0xFFFF; rGreg[s] = rtask ? (meta >> 4) : 1; rvreg[s] = leader; }
        z4.set(s, rgl(s), leader, leader ? bd.z4[ol + rid] : 0.0);
        f.set(s, rgl(s), leader, leader ? bd.f[ol + rid] : 0.0);
        y3.set(s, rgl(s), leader, 0.0);
        Ex[s] = 0.0;
    }
    // direct x-update: class of this lane's row (dense index among the G rows, -1 = a D row: its columns meet no other D row) and the
    // D row's weight 1 / (c + live variables in the row)
    int gdir = -1;
    double wrow = 0.0;
    if constexpr (DIRECT) {
        if (rvalid(0)) {
            gdir = bd.rdir[on + tid];
            if (gdir >= 0) dmap[gdir] = rgl(0);
            if (h_valid) wrow = bd.H[(size_t)inst * ((size_t)bd.HL * bd.HLD + bd.LS) + (size_t)bd.HL * bd.HLD + rgl(0)];
            gl[3 * rgl(0) + 2] = 0.0;
        }
    }
    double rho1 = dsc[ND_RHO1], rho2 = dsc[ND_RHO2], rho4 = dsc[ND_RHO4];
    double prev_rho1 = dsc[ND_PREV_RHO1], prev_rho2 = dsc[ND_PREV_RHO2], prev_rho4 = dsc[ND_PREV_RHO4];
    double gamma_val = dsc[ND_GAMMA], dI = dsc[ND_DI], r4Et = dsc[ND_R4ET], rcr = dsc[ND_RCR];
    double std_obj = dsc[ND_STD_OBJ], cur_obj = dsc[ND_CUR_OBJ], best_bin_obj = dsc[ND_BEST_BIN_OBJ];
    double sum_fix_obj = dsc[ND_SUM_FIX_OBJ], fix_obj = dsc[ND_FIX_OBJ], c1 = dsc[ND_C1];
    double cvg1 = dsc[ND_CVG1], cvg2 = dsc[ND_CVG2], obj_val = dsc[ND_OBJ_VAL];
    double prev_sum = dsc[ND_PREV_SUM], prev_obj = dsc[ND_PREV_OBJ];
    int n_live = isc[NI_NLIVE], rhoUpdated = isc[NI_RHO_UPDATED], hist_n = isc[NI_HIST_N];
    int pcg_total = isc[NI_PCG_TOTAL], outer_total = isc[NI_OUTER_TOTAL], last_pcg = isc[NI_LAST_PCG];
    int expr_ready = isc[NI_EXPR_READY];
    // the last LP_HIST objective values (compute_std_obj, LPcpp:459-469) live in LDS: every thread reads the window of the previous
    // iteration from one buffer, thread 0 writes the new window into the other (no thread can still be reading that one: barriers of a
    // whole iteration lie in between)
    int hbuf = 0;
    double h_reg[LEAN ? 1 : LP_HIST];            // single-slot variants have the registers to keep the window where it is used
    if constexpr (LEAN) { if (tid < LP_HIST) s_hist[tid] = bd.hist[(size_t)inst * LP_HIST + tid]; }
    else {
#pragma unroll
        for (int k = 0; k < LP_HIST; k++) h_reg[k] = bd.hist[(size_t)inst * LP_HIST + k];
    }
    const double learning_fact = LP_LEARNING_FACT;
    int ret = 0, stop = LP_STOP_NONE, parity = 0;

    __syncthreads();   // index sets staged

    // ---- gather lists of this thread's rows and columns, kept in registers for the whole launch ----
    static_assert(RCAPS::N == EPT && CCAPS::N == EPT && HCAPS::N == EPT, "one capacity per slot");
    Lists<RCAPS> rl;
    Lists<CCAPS> cl;             // the leading entries of this position's own column
    Lists<HCAPS> hl;             // helper share: a chunk of the tail of the long column of this lane's quad (section 5 of DESIGN.md)
    bool clong[EPT];             // this lane owns the quad's long column: its sum = own part + the quad's helper partials
    int anyhelp[EPT];            // wave-uniform: some quad of this wave splits a column
    rl.base = lds_addr(gx); rl.zero = lds_addr(gx + bd.NS);
    rl.ptr = s_rs_ptr; cl.ptr = s_cs_ptr; hl.ptr = s_hs_ptr;
    rl.pos0 = cl.pos0 = hl.pos0 = tid; rl.pstride = cl.pstride = hl.pstride = T;
    cl.base = lds_addr(gl); cl.zero = lds_addr(gl + 3 * bd.LS);
    hl.base = cl.base; hl.zero = cl.zero;
    int rGmax[EPT];              // wave-uniform largest lane group in the slot
    static_for<EPT>([&](auto S) {
        constexpr int s = decltype(S)::value;
        const int pos = s * T + tid;
        build_list<RCAPS, s, 8>(rl, s_rs_ptr[pos], s_rs_ptr[pos + 1], s_rs_col);
        rGmax[s] = wave_max_int(rG(s));
        build_list<CCAPS, s, 24>(cl, s_cs_ptr[pos], s_cs_ptr[pos + 1], s_cs_row);
        build_list<HCAPS, s, 24>(hl, s_hs_ptr[pos], s_hs_ptr[pos + 1], s_cs_row);
        const int cm = bd.cmeta[on + pos];
        clong[s] = (cm & 0x8000) != 0;
        anyhelp[s] = wave_max_int(s_hs_ptr[pos + 1] - s_hs_ptr[pos]) > 0;
    });
    auto op_add = [](double acc, double v) { return acc + v; };                 // res += 1.0 * v
    // out = (E * gx)_row for this thread's row tasks: the G lanes of a task add their interleaved share of the row in
    // ascending column order, then combine by an xor butterfly inside the (aligned) lane group; every lane of the group
    // ends up with the row sum, the leader uses it.
    constexpr int GX2 = T * EPT + 2;             // the second n-vector of gx, as an immediate ds_read offset in doubles
    auto rows_gather_at = [&](auto COMPC, double (&out)[EPT]) {
        double part[EPT];
        if constexpr (LEAN) touch_list(rl);
        gather_all<RCAPS, 8, decltype(COMPC)::value>(rl, s_rs_col, op_add, part);
        static_for<EPT>([&](auto S) {
            constexpr int s = decltype(S)::value;
            double v = part[s];
            if (rGmax[s] >= 2) { const double u = v + dpp_mov<0xB1>(v); v = rG(s) >= 2 ? u : v; }
            if (rGmax[s] >= 4) { const double u = v + dpp_mov<0x4E>(v); v = rG(s) >= 4 ? u : v; }
            if (rGmax[s] >= 8) { const double u = v + dpp_mov<0x141>(v); v = rG(s) >= 8 ? u : v; }
            out[s] = v;
        });
    };
    auto rows_gather = [&](double (&out)[EPT]) { rows_gather_at(std::integral_constant<int, 0>{}, out); };

    // out = (E^T * gl[.][COMP])_col for this thread's variables through OP.  A long column is shared inside its quad of lanes:
    // its owner adds the leading entries, the other lanes of the quad add consecutive chunks of the rest into a second
    // accumulator after their own (short) columns; the helper partials are combined by an xor butterfly over the quad
    // ((h0 + h1) + (h2 + h3), the owner's own helper partial is +0.0) and added to the owner's part.  Every partial sum runs
    // in ascending row order from +0.0.
    auto quad_sum = [](double v) { v = v + dpp_mov<0xB1>(v); v = v + dpp_mov<0x4E>(v); return v; };
    auto cols_gather = [&](auto COMPC, auto op, double (&out)[EPT]) {
        constexpr int COMP = decltype(COMPC)::value;
        double own[EPT];
        if constexpr (LEAN) { touch_list(cl); touch_list(hl); }
        gather_all<CCAPS, 24, COMP>(cl, s_cs_row, op, own);
        if constexpr (HCAPS::total > 0) {
            double hp[EPT];
            gather_all<HCAPS, 24, COMP>(hl, s_cs_row, op, hp);
            static_for<EPT>([&](auto S) {
                constexpr int s = decltype(S)::value;
                if (anyhelp[s]) { const double ht = quad_sum(hp[s]); out[s] = clong[s] ? own[s] + ht : own[s]; }
                else out[s] = own[s];
            });
        } else {
#pragma unroll
            for (int s = 0; s < EPT; s++) out[s] = own[s];
        }
    };
    auto cols_gather2 = [&](auto opa, auto opb, double (&outA)[EPT], double (&outB)[EPT]) {      // components 0 and 1 (rhs assembly)
        double ownA[EPT], ownB[EPT];
        if constexpr (LEAN) { touch_list(cl); touch_list(hl); }
        gather_all2<CCAPS, 24, 0, 1>(cl, s_cs_row, opa, opb, ownA, ownB);
        if constexpr (HCAPS::total > 0) {
            double hA[EPT], hB[EPT];
            gather_all2<HCAPS, 24, 0, 1>(hl, s_cs_row, opa, opb, hA, hB);
            static_for<EPT>([&](auto S) {
                constexpr int s = decltype(S)::value;
                if (anyhelp[s]) {
                    const double ta = quad_sum(hA[s]), tb = quad_sum(hB[s]);
                    outA[s] = clong[s] ? ownA[s] + ta : ownA[s];
                    outB[s] = clong[s] ? ownB[s] + tb : ownB[s];
                } else { outA[s] = ownA[s]; outB[s] = ownB[s]; }
            });
        } else {
#pragma unroll
            for (int s = 0; s < EPT; s++) { outA[s] = ownA[s]; outB[s] = ownB[s]; }
        }
    };

    // ---- early fixing: apply this call's fix vector (LPcpp:1124-1335) as a mask ----
    bool finished = false;
    if (bd.ctl[(size_t)inst * 4 + 0]) {
        const int n_live_new = bd.ctl[(size_t)inst * 4 + 2];
        double part[1] = {0.0};
        uint8_t nf[EPT];
#pragma unroll
        for (int s = 0; s < EPT; s++) {
            const int pos = s * T + tid;
            nf[s] = bd.newfix[on + pos];
            const double val = nf[s] == 2 ? 1.0 : 0.0;
            part[0] = part[0] + (nf[s] ? b.get(s) * val : 0.0);     // fix_obj = b2.dot(x2), :1237
            gx[pos] = nf[s] ? val : 0.0;
        }
        block_sum<T, 1>(part, red, parity);                      // (barrier inside also publishes gx)
        fix_obj = part[0];
        if (n_live_new == 0) {                                   // :1212-1217 (nothing else is updated)
            ret = 1; stop = LP_STOP_ALLFIXED; n_live = 0; finished = true;
#pragma unroll
            for (int s = 0; s < EPT; s++) if (nf[s]) { live[s] = false; x[s] = nf[s] == 2 ? 1.0 : 0.0; }
        } else {
            double cnt[EPT];
            rows_gather(cnt);                                         // E2*x2, :1276
#pragma unroll
            for (int s = 0; s < EPT; s++) {
                f.set(s, rgl(s), rvalid(s), f.get(s, rgl(s), rvalid(s)) - cnt[s]);      // f1 = f - E2*x2, :1278
                if (nf[s]) { live[s] = false; x[s] = nf[s] == 2 ? 1.0 : 0.0; }
            }
            double px[1] = {0.0};
#pragma unroll
            for (int s = 0; s < EPT; s++) px[0] = px[0] + (live[s] ? x[s] * x[s] : 0.0);
            block_sum<T, 1>(px, red, parity);
            if (sqrt(px[0]) < 1e-3) ret = 1;                          // :1223
            prev_sum = sum_fix_obj; sum_fix_obj += fix_obj; prev_obj = cur_obj;   // :1247-1250
            n_live = n_live_new;
            c1 = bd.dctl[inst];
            // update_expression (:1329 -> :2289-2404)
            dI = 0.0; dI += rho1 + rho2;
#pragma unroll
            for (int s = 0; s < EPT; s++) { double v = dI; v += rho4 * Esq(s); pd.set(s, v); }
            r4Et = rho4;
            expr_ready = 1;
            h_valid = 0;                                             // direct mode: E lost columns, the inverse is rebuilt
        }
    }

    int it = iter_start;
    if (!finished) {
        // DiagonalPreconditioner state (LPcpp:883-890): always 1/pd as of the last compute; recomputed here so that a fix
        // applied while no rho update is pending (stale preconditioner = UB in the reference) is well defined.
#pragma unroll
        for (int s = 0; s < EPT; s++) { const double v = pd.get(s); dinv[s] = (v != 0.0) ? 1.0 / v : 1.0; }
        // E*x for the first iteration's y3
        __syncthreads();
#pragma unroll
        for (int s = 0; s < EPT; s++) gx[s * T + tid] = live[s] ? x[s] : 0.0;
        __syncthreads();
        rows_gather(Ex);

        // The first half of an iteration (LPcpp:806-828) needs only what the END of the previous one holds -- x, z1, z2, z4, E x and the
        // rho of the next iteration: y3 with the published l-vectors, the PCG start vector y1 and this thread's share of the sphere
        // norm are prepared there, so that the norm rides in the residual reduction and the l-vectors are published by its barrier
        // (two barriers fewer per iteration than computing them where the reference does; same expressions, same bits).
        // (one-/two-slot variants carry y1 and the unscaled y2 into the next iteration in registers; the register-lean ones recompute them)
        double y1c[LEAN ? 1 : EPT], uc[LEAN ? 1 : EPT];
        auto prepare = [&](double r1, double r2, double r4) {
            double part = 0.0;
#pragma unroll
            for (int s = 0; s < EPT; s++) {
                const double u = (x[s] + z2.get(s) / r2) - 0.5;       // project_shifted_Lp_ball :423-428
                part = part + (live[s] ? u * u : 0.0);
                if constexpr (!DIRECT || !LEAN) {
                    const double t = x[s] + z1.get(s) / r1;
                    const double y1n = t > 1 ? 1 : (t < 0 ? 0 : t);   // project_box :409-421
                    if constexpr (!DIRECT) gx[GX2 + s * T + tid] = live[s] ? y1n : 0.0;      // PCG start x0 = y1 (:892)
                    if constexpr (!LEAN) { y1c[s] = y1n; uc[s] = u; }
                }
                // y3 = max(0, f - E x - z4/rho4), LPcpp:824-828
                if constexpr (LEAN) {
                    // LDS-backed row vectors: the leader lane's whole read-compute-write in ONE predicated region with one index (four
                    // separate predicated accesses were four branches, each fetching the spilled index back from scratch first)
                    if (rvalid(s)) {
                        const int i = RowVec<true, EPT, 1>::at(rgl(s));
                        const double fs = f.lds[i], z4s = z4.lds[3 * i];
                        const double v = fs - Ex[s] - z4s / r4;
                        const double y3s = v < 0 ? 0 : v;
                        y3.lds[i] = y3s;
                        gl[3 * i] = fs - y3s;
                    }
                } else {
                const double fs = f.get(s, rgl(s), rvalid(s)), z4s = z4.get(s, rgl(s), rvalid(s));
                const double v = fs - Ex[s] - z4s / r4;
                const double y3s = v < 0 ? 0 : v;
                y3.set(s, rgl(s), rvalid(s), y3s);
                if (rvalid(s)) { gl[3 * rgl(s)] = fs - y3s; gl[3 * rgl(s) + 1] = z4s; }
                }
            }
            return part;
        };
        double pnorm;
        {
            double pn[1] = {prepare(rho1, rho2, rho4)};
            block_sum<T, 1>(pn, red, parity);                         // (its barrier publishes the l-vectors and the start vector)
            pnorm = pn[0];
        }

        int cc = 0;
        unsigned long long log_t0 = 0;
        if constexpr (LOG) log_t0 = wall_clock64();
        STAMP_DECL
        for (; it < iter_end; ++it) {
            STAMP(15)
            // ---------------- y1 (box) and y2 (shifted L2 sphere), LPcpp:806-818 ----------------
            double y1[EPT], y2[EPT];
            const double c2 = 2 * sqrt(pnorm);
#pragma unroll
            for (int s = 0; s < EPT; s++) {
                if constexpr (LEAN) {
                    const double t = x[s] + z1.get(s) / rho1;
                    y1[s] = t > 1 ? 1 : (t < 0 ? 0 : t);
                    const double u = (x[s] + z2.get(s) / rho2) - 0.5;
                    y2[s] = u * c1 / c2 + 0.5;
                } else {
                    y1[s] = y1c[s];
                    y2[s] = uc[s] * c1 / c2 + 0.5;
                }
            }
            STAMP(0)
            // ---------------- matrix-expression refresh, LPcpp:831-866 ----------------
            if (it == 0) {                                            // update_expression(0)
                dI = 0.0; dI += rho1 + rho2;
#pragma unroll
                for (int s = 0; s < EPT; s++) { double v = dI; v += rho4 * Esq(s); pd.set(s, v); }
                r4Et = rho4;
                expr_ready = 1;
            }
            if (it != 0 && rhoUpdated) {
                const double inc = rcr * (prev_rho1 + prev_rho2);
                const double inc4 = rcr * prev_rho4;
                dI += inc;
#pragma unroll
                for (int s = 0; s < EPT; s++) { double v = pd.get(s); v += inc; v += inc4 * Esq(s); pd.set(s, v); }
                r4Et = learning_fact * r4Et;                          // rho4_E_transpose *= learning_fact (:864)
            }
            const double r4 = r4Et;
#ifdef LPBOX_KO_NOMUL
            auto op_scaled = [r4](double acc, double v) { return acc + v; };
#else
            auto op_scaled = [r4](double acc, double v) { return acc + r4 * v; };   // res += (rho4*1.0) * v
#endif
            // ---------------- rhs (:872-878) and q = E*y1 ----------------
            double rhs[EPT], tAs[EPT], tBs[EPT];
            cols_gather2(op_scaled, op_add, tAs, tBs);                                  // (rho4 E^T)(f - y3) and E^T z4 in one pass
#pragma unroll
            for (int s = 0; s < EPT; s++) {
                const double tA = tAs[s], tB = tBs[s];
                double r_ = (rho1 * y1[s] + rho2 * y2[s]) - ((b.get(s) + z1.get(s)) + z2.get(s));
                r_ += tA;
                r_ -= tB;
                rhs[s] = r_;
            }
            if constexpr (!DIRECT) {
                double q[EPT];
                rows_gather_at(std::integral_constant<int, GX2>{}, q);    // q = E y1 (the start vector sits in the second n-vector)
#pragma unroll
                for (int s = 0; s < EPT; s++) if (rvalid(s)) gl[3 * rgl(s) + 2] = r4 * q[s];   // the factor of the column product, once per row (see the PCG loop)
                __syncthreads();
            }
            if (rhoUpdated) {                                         // DiagonalPreconditioner::compute (:883-890)
#pragma unroll
                for (int s = 0; s < EPT; s++) { const double v = pd.get(s); dinv[s] = (v != 0.0) ? 1.0 / v : 1.0; }
                rhoUpdated = 0;
            }
            z1.park(); z2.park(); b.park(false);                      // register-lean variants: out of the register file while the PCG runs
            STAMP(1)
            // ---------------- PCG (LPcpp:251-335) on (dI*I + rho4 E^T E) x = rhs ----------------
            double xt[EPT];
            int k_it = 0;
            bool pcg_fail = false;
            if constexpr (DIRECT) {
                // ---------------- DIRECT x-update (opt-in, NOT the reference's PCG) ----------------
                // (a I + r E^T E) x = rhs, a = dI, r = r4Et, solved exactly.  Rows of E: D rows (pairwise disjoint columns, chosen by the
                // host) and G rows (the rest, dense index gdir).  a I + r E_D^T E_D is block diagonal with blocks a I + r 1 1^T: its
                // inverse is Q / a, Q v = v - E_D^T W E_D v, W = diag(1 / (c + m_i)), c = a / r.  Woodbury over the G rows:
                //     x = Q (rhs - E_G^T H E_G Q rhs) / a,   H = (c I + E_G Q E_G^T)^-1  (|G| x |G|, in LDS).
                // c is constant while rho1, rho2, rho4 are scaled together (LPcpp:951-970): H and W are built once per instance and fix.
                const int nG = bd.dng[inst], HLD = bd.HLD;
                // Q v for this thread's variable: sigma = E v by rows, D rows publish w * sigma (G rows 0), the column sum picks it up
                auto apply_Q = [&](double vin) {
                    gx[tid] = live[0] ? vin : 0.0;
                    __syncthreads();
                    double sg[EPT], cs[EPT];
                    rows_gather(sg);
                    if (rvalid(0)) gl[3 * rgl(0)] = gdir < 0 ? wrow * sg[0] : 0.0;
                    __syncthreads();
                    cols_gather(std::integral_constant<int, 0>{}, op_add, cs);
                    return vin - cs[0];
                };
                if (!h_valid) {
                    const double cdiag = dI / r4Et;
                    {
                        gx[tid] = live[0] ? 1.0 : 0.0;
                        __syncthreads();
                        double cnt[EPT];
                        rows_gather(cnt);                                     // m_i
                        wrow = 1.0 / (cdiag + cnt[0]);
                    }
                    for (int e = tid; e < bd.HL * HLD; e += T) Hm[e] = 0.0;
                    for (int bcol = 0; bcol < nG; bcol++) {                   // column bcol of c I + E_G Q E_G^T
                        if (rvalid(0) && gdir >= 0) gl[3 * rgl(0) + 2] = (gdir == bcol) ? 1.0 : 0.0;
                        __syncthreads();
                        double w[EPT], g[EPT];
                        cols_gather(std::integral_constant<int, 2>{}, op_add, w);
                        const double qw = apply_Q(w[0]);
                        gx[tid] = live[0] ? qw : 0.0;
                        __syncthreads();
                        rows_gather(g);
                        if (rvalid(0) && gdir >= 0) Hm[gdir * HLD + bcol] = (gdir == bcol) ? g[0] + cdiag : g[0];
                    }
                    __syncthreads();
                    // in-place Gauss-Jordan without pivoting (SPD): step k scales row k by the reciprocal pivot, every other element
                    // loses (its column-k entry) x (scaled row k), column k becomes -(entry x reciprocal pivot)
                    for (int k = 0; k < nG; k++) {
                        const double piv = 1.0 / Hm[k * HLD + k];
                        if (tid < nG) { dcol[tid] = Hm[tid * HLD + k]; drow[tid] = tid == k ? piv : Hm[k * HLD + tid] * piv; }
                        __syncthreads();
                        {
                            const int j = tid & 127;
                            if (j < nG) {
                                const double rk = drow[j];
                                for (int i = tid >> 7; i < nG; i += T / 128) {
                                    const double ck = dcol[i];
                                    const double cur = Hm[i * HLD + j];
                                    Hm[i * HLD + j] = i == k ? rk : (j == k ? -(ck * piv) : cur - ck * rk);
                                }
                            }
                        }
                        __syncthreads();
                    }
                    {
                        double *Hg = bd.H + (size_t)inst * ((size_t)bd.HL * HLD + bd.LS);
                        for (int e = tid; e < bd.HL * HLD; e += T) Hg[e] = Hm[e];
                        if (rvalid(0)) Hg[(size_t)bd.HL * HLD + rgl(0)] = wrow;
                    }
                    h_valid = 1;
                }
                const double q1 = apply_Q(rhs[0]);
                gx[tid] = live[0] ? q1 : 0.0;
                __syncthreads();
                {
                    double t[EPT];
                    rows_gather(t);                                           // t = E_G Q rhs
                    if (rvalid(0) && gdir >= 0) dcol[gdir] = t[0];
                }
                __syncthreads();
                // u = H t: the four lanes of a quad share a G row, lane q adds the columns j = q, q + 4, ... in ascending order,
                // the partials combine as (p0 + p1) + (p2 + p3)
                for (int i = tid >> 2; i < nG; i += T / 4) {
                    const double *Hr = Hm + i * HLD;
                    double acc = 0.0;
#pragma unroll 4
                    for (int j = tid & 3; j < nG; j += 4) acc = acc + Hr[j] * dcol[j];
                    const double u = quad_sum(acc);
                    if ((tid & 3) == 0) gl[3 * dmap[i] + 2] = u;
                }
                __syncthreads();
                double v[EPT];
                cols_gather(std::integral_constant<int, 2>{}, op_add, v);    // v = E_G^T u (the D rows hold 0)
                const double yv = rhs[0] - v[0];
                xt[0] = apply_Q(yv) / dI;
            } else {
            double r[EPT], p[EPT];
            double p3[3] = {0.0, 0.0, 0.0};
            double tcol[EPT];
            cols_gather(std::integral_constant<int, 2>{}, op_add, tcol);
#pragma unroll
            for (int s = 0; s < EPT; s++) {
                const double t = tcol[s];
                xt[s] = y1[s];
                double Mx = 0.0;
                Mx += dI * (1.0 * xt[s]);
                Mx += t;
                r[s] = rhs[s] - Mx;                                   // :268
                p[s] = dinv[s] * r[s];                                // :291
                p3[0] = p3[0] + (live[s] ? rhs[s] * rhs[s] : 0.0);    // rhsNorm2 :271
                p3[1] = p3[1] + (live[s] ? r[s] * r[s] : 0.0);        // residualNorm2 :282
                p3[2] = p3[2] + (live[s] ? r[s] * p[s] : 0.0);        // absNew :294
            }
            block_sum<T, 3>(p3, red, parity);
            STAMP(2)
            const double rhsNorm2 = p3[0];
            double residualNorm2 = p3[1], absNew = p3[2];
            if (rhsNorm2 == 0) {                                      // :273-278
#pragma unroll
                for (int s = 0; s < EPT; s++) xt[s] = 0.0;
            } else {
                double threshold = LP_PCG_TOL * LP_PCG_TOL * rhsNorm2;    // :281
                if (threshold < DBL_MIN) threshold = DBL_MIN;
                if (!(residualNorm2 < threshold)) {                   // :284
                    while (k_it < LP_PCG_MAXITERS) {                  // :296
#pragma unroll
                        for (int s = 0; s < EPT; s++) gx[s * T + tid] = live[s] ? p[s] : 0.0;
                        STAMP_PRE(12)
                        __syncthreads();
                        STAMP(3)
                        {
                            double q[EPT];
#ifdef LPBOX_KO_ROWS
                            for (int s = 0; s < EPT; s++) q[s] = p[s];
#else
                            rows_gather(q);
#endif
                            STAMP(4)
#pragma unroll
                            // the column product adds (rho4 * 1.0) * q_i per ENTRY (LPcpp:115-162): that product is the same value for every
                            // column that meets row i, so the row's owner forms it once and publishes r4 * q_i -- one multiplication per row
                            // instead of one per entry of E in the gather below; fl(acc + fl(r4 * q_i)) is unchanged bit for bit
#ifdef LPBOX_KO_NOMUL
                            for (int s = 0; s < EPT; s++) if (rvalid(s)) gl[3 * rgl(s)] = q[s];
#else
                            for (int s = 0; s < EPT; s++) if (rvalid(s)) gl[3 * rgl(s)] = r4 * q[s];
#endif
                        }
                        STAMP_PRE(13)
                        __syncthreads();
                        STAMP(5)
                        double tmp[EPT];
                        double p1[1] = {0.0};
#ifdef LPBOX_KO_COLS
                        for (int s = 0; s < EPT; s++) tcol[s] = gl[3 * (tid % 128)];
#else
                        cols_gather(std::integral_constant<int, 0>{}, op_add, tcol);          // entries already carry the factor r4
#endif
#pragma unroll
                        for (int s = 0; s < EPT; s++) {               // tmp = M p (:298), fused p.tmp
                            const double t = tcol[s];
                            double Mp = 0.0;
                            Mp += dI * (1.0 * p[s]);
                            Mp += t;
                            tmp[s] = Mp;
                            p1[0] = p1[0] + (live[s] ? p[s] * tmp[s] : 0.0);
                        }
                        STAMP(6)
#ifndef LPBOX_KO_RED1
                        block_sum<T, 1>(p1, red, parity);
#endif
                        STAMP(7)
#ifdef LPBOX_KO_DIV
                        const double alpha = absNew * p1[0];
#else
                        const double alpha = absNew / p1[0];          // :300
                        if (alpha < 0) { pcg_fail = true; break; }    // :301
#endif
                        double p2[2] = {0.0, 0.0};
                        double z[EPT];
#pragma unroll
                        for (int s = 0; s < EPT; s++) {
                            xt[s] += alpha * p[s];                    // :302
                            r[s] -= alpha * tmp[s];                   // :304
                            z[s] = dinv[s] * r[s];                    // :314
                            p2[0] = p2[0] + (live[s] ? r[s] * r[s] : 0.0);   // :305
                            p2[1] = p2[1] + (live[s] ? r[s] * z[s] : 0.0);   // :317
                        }
                        STAMP(8)
#ifndef LPBOX_KO_RED2
                        block_sum<T, 2>(p2, red, parity);
#endif
                        STAMP(9)
                        residualNorm2 = p2[0];
#ifdef LPBOX_KO_FIXED_PCG
                        if (k_it + 1 >= LPBOX_KO_FIXED_PCG) { k_it++; break; }
#else
                        if (residualNorm2 < threshold) { k_it++; break; }     // :309-312
#endif
                        const double absOld = absNew;
                        absNew = p2[1];
#ifdef LPBOX_KO_DIV
                        const double beta = absNew * absOld;
#else
                        const double beta = absNew / absOld;          // :318
#endif
#pragma unroll
                        for (int s = 0; s < EPT; s++) p[s] = z[s] + beta * p[s];   // :319
                        k_it++;
                        STAMP(10)
                    }
                }
            }
            }   // PCG
            STAMP(11)
            z1.unpark(); z2.unpark(); b.unpark();                     // ... and back, one batch of loads
            last_pcg = k_it;
            pcg_total += k_it;
            if (pcg_fail) stop = LP_STOP_PCG;
            if (pcg_fail && l2f) { ret = 1; break; }                  // :1450-1454: return 1, x_sol untouched
            if constexpr (LEAN) {       // y1, y2 were not kept across the PCG: the same expressions on the same operands give the same bits
#pragma unroll
                for (int s = 0; s < EPT; s++) {
                    const double t = x[s] + z1.get(s) / rho1;
                    y1[s] = t > 1 ? 1 : (t < 0 ? 0 : t);
                    const double u = (x[s] + z2.get(s) / rho2) - 0.5;
                    y2[s] = u * c1 / c2 + 0.5;
                }
            }
            // ---------------- commit x: branchless predicated write (fixed variables keep their value) ----------------
#pragma unroll
            for (int s = 0; s < EPT; s++) x[s] = live[s] ? xt[s] : x[s];
            outer_total++;
            if (rec) {                                                // x_iters column cc (:1472-1475); plain loop: the xiter dump (:903-909)
                double *xh = bd.xhist + ((size_t)inst * bd.ws_cap + cc) * bd.NS;
#pragma unroll
                for (int s = 0; s < EPT; s++) xh[s * T + tid] = x[s];
                cc++;
            }
            // ---------------- duals (:917-924 / :1487-1491) ----------------
            const double g1 = gamma_val * rho1, g2 = gamma_val * rho2, g4 = gamma_val * rho4;
#pragma unroll
            for (int s = 0; s < EPT; s++) {
                z1.set(s, z1.get(s) + g1 * (x[s] - y1[s]));
                z2.set(s, z2.get(s) + g2 * (x[s] - y2[s]));
                gx[s * T + tid] = live[s] ? x[s] : 0.0;
            }
            __syncthreads();
            rows_gather(Ex);                                          // E*x: feeds z4 now and y3 of the next iteration
#pragma unroll
            for (int s = 0; s < EPT; s++) {
                if constexpr (LEAN) {
                    if (rvalid(s)) {                                  // (one predicated region per slot: see prepare())
                        const int i = RowVec<true, EPT, 1>::at(rgl(s));
                        const double d = g4 * ((Ex[s] + y3.lds[i]) - f.lds[i]);
                        const double z4o = z4.lds[3 * i];
                        z4.lds[3 * i] = (!l2f && it == iter_start) ? d : z4o + d;
                    }
                } else {
                const double d = g4 * ((Ex[s] + y3.get(s, rgl(s), rvalid(s))) - f.get(s, rgl(s), rvalid(s)));
                z4.set(s, rgl(s), rvalid(s), (!l2f && it == iter_start) ? d : z4.get(s, rgl(s), rvalid(s)) + d);   // :920-923 (plain loop overwrites on its first iteration)
                }
            }
            // ---------------- residual norms, objective (:931-1011); the next iteration's first half rides along ----------------
            double lg[6] = {0.0, 0.0, 0.0, 0.0, 0.0, 0.0};            // LOG: |y1|^2 |y2|^2 |y3|^2 |z1|^2 |z2|^2 |z4|^2 of THIS iteration (before prepare() moves y3 on)
            if constexpr (LOG) {
#pragma unroll
                for (int s = 0; s < EPT; s++) {
                    const double vy3 = y3.get(s, rgl(s), rvalid(s)), vz4 = z4.get(s, rgl(s), rvalid(s));
                    lg[0] = lg[0] + (live[s] ? y1[s] * y1[s] : 0.0);
                    lg[1] = lg[1] + (live[s] ? y2[s] * y2[s] : 0.0);
                    lg[2] = lg[2] + (rvalid(s) ? vy3 * vy3 : 0.0);
                    lg[3] = lg[3] + (live[s] ? z1.get(s) * z1.get(s) : 0.0);
                    lg[4] = lg[4] + (live[s] ? z2.get(s) * z2.get(s) : 0.0);
                    lg[5] = lg[5] + (rvalid(s) ? vz4 * vz4 : 0.0);
                }
            }
            const bool rho_step = (it + 1) % LP_RHO_STEP == 0;        // the rho the next iteration will see (:951-970)
            double p5[6] = {0.0, 0.0, 0.0, 0.0, 0.0, 0.0};
            p5[5] = prepare(rho_step ? learning_fact * rho1 : rho1, rho_step ? learning_fact * rho2 : rho2, rho_step ? learning_fact * rho4 : rho4);
#pragma unroll
            for (int s = 0; s < EPT; s++) {
                const double d1 = x[s] - y1[s], d2 = x[s] - y2[s];
                const double xb = x[s] >= 0.5 ? 1.0 : 0.0;
                p5[0] = p5[0] + (live[s] ? x[s] * x[s] : 0.0);
                p5[1] = p5[1] + (live[s] ? d1 * d1 : 0.0);
                p5[2] = p5[2] + (live[s] ? d2 * d2 : 0.0);
                p5[3] = p5[3] + (live[s] ? b.get(s) * x[s] : 0.0);
                p5[4] = p5[4] + (live[s] ? b.get(s) * xb : 0.0);
            }
            STAMP_POST(12)
            block_sum<T, 6>(p5, red, parity);
            if constexpr (LOG) block_sum<T, 6>(lg, red, parity);
            pnorm = p5[5];
            STAMP_POST(13)
            {
                const double xn = sqrt(p5[0]);
                const double temp0 = (xn < 2.2204e-16) ? 2.2204e-16 : xn;
                cvg1 = sqrt(p5[1]) / temp0;
                cvg2 = sqrt(p5[2]) / temp0;
            }
            if (cvg1 <= LP_STOP_THRESHOLD && cvg2 <= LP_STOP_THRESHOLD && (l2f || it != iter_start)) {
                if (l2f) ret = 1;                                     // :1505 (plain loop: ret stays 0, :934-949)
                stop = LP_STOP_Y1Y2;
                break;
            }
            if (rho_step) {                                           // :951-970
                prev_rho1 = rho1; prev_rho2 = rho2;
                rho1 = learning_fact * rho1;
                rho2 = learning_fact * rho2;
                prev_rho4 = rho4;
                rho4 = learning_fact * rho4;
                const double g = gamma_val * LP_GAMMA_FACTOR;
                gamma_val = g < 1.0 ? 1.0 : g;
                rhoUpdated = 1;
                rcr = learning_fact - 1.0;
            }
            obj_val = p5[3];                                          // :972
            double h[LP_HIST];
            if constexpr (LEAN) {
                const double *hs = s_hist + hbuf * LP_HIST;
#pragma unroll
                for (int k = 0; k < LP_HIST; k++) h[k] = hs[k];
            } else {
#pragma unroll
                for (int k = 0; k < LP_HIST; k++) h[k] = h_reg[k];
            }
            if (hist_n < LP_HIST) {
#pragma unroll
                for (int k = 0; k < LP_HIST; k++) if (k == hist_n) h[k] = obj_val;
            } else {
#pragma unroll
                for (int k = 0; k < LP_HIST - 1; k++) h[k] = h[k + 1];
                h[LP_HIST - 1] = obj_val;
            }
            if constexpr (LEAN) {
                hbuf ^= 1;
                if (tid == 0) {
#pragma unroll
                    for (int k = 0; k < LP_HIST; k++) s_hist[hbuf * LP_HIST + k] = h[k];
                }
            } else {
#pragma unroll
                for (int k = 0; k < LP_HIST; k++) h_reg[k] = h[k];
            }
            if (hist_n < 0x3fffffff) hist_n++;
            if (hist_n >= LP_HIST) {                                  // compute_std_obj :459-469, std_dev :358-377
                double mean = 0;
#pragma unroll
                for (int k = 0; k < LP_HIST; k++) mean += h[k];
                mean /= (double)LP_HIST;
                double dev = 0;
#pragma unroll
                for (int k = 0; k < LP_HIST; k++) dev += (h[k] - mean) * (h[k] - mean);
                dev /= (double)(LP_HIST - 1);
                const double sd = (dev == 0) ? 0.0 : sqrt(dev);
                std_obj = sd / fabs(h[LP_HIST - 1]);
            }
            if (std_obj <= LP_STD_THRESHOLD) { ret = 1; stop = LP_STOP_OBJSTD; break; }   // :977
            cur_obj = p5[4];                                          // :1001-1003
            if (best_bin_obj >= cur_obj) best_bin_obj = cur_obj;
            if constexpr (LOG) {                                      // the reference logs the iterations that did not break (:1013-1067)
                if (tid == 0 && bd.logbuf && it - iter_start < bd.log_cap) {
                    double *lr = bd.logbuf + ((size_t)inst * bd.log_cap + (it - iter_start)) * LP_LOG_VALS;
                    lr[0] = (double)k_it; lr[1] = sqrt(p5[0]);
#pragma unroll
                    for (int k = 0; k < 6; k++) lr[2 + k] = sqrt(lg[k]);
                    lr[8] = obj_val; lr[9] = cur_obj; lr[10] = (double)(wall_clock64() - log_t0); lr[11] = (double)it;
                }
            }
            if constexpr (LEAN) {       // the scalar state is uniform: keep it out of the vector registers across the next iteration
                rho1 = uniform_f64(rho1); rho2 = uniform_f64(rho2); rho4 = uniform_f64(rho4);
                prev_rho1 = uniform_f64(prev_rho1); prev_rho2 = uniform_f64(prev_rho2); prev_rho4 = uniform_f64(prev_rho4);
                gamma_val = uniform_f64(gamma_val); rcr = uniform_f64(rcr); dI = uniform_f64(dI); r4Et = uniform_f64(r4Et);
                std_obj = uniform_f64(std_obj); cur_obj = uniform_f64(cur_obj); best_bin_obj = uniform_f64(best_bin_obj);
                cvg1 = uniform_f64(cvg1); cvg2 = uniform_f64(cvg2); obj_val = uniform_f64(obj_val); pnorm = uniform_f64(pnorm);
            }
            STAMP(14)
        }
        STAMP_STORE
    }

    // ---- write the state back ----
    z1.store(); z2.store(); pd.store();
#pragma unroll
    for (int s = 0; s < EPT; s++) {
        const int pos = s * T + tid;
        {
            bd.x[on + pos] = x[s];
            bd.live[on + pos] = live[s] ? 1 : 0;
        }
        if (rvalid(s)) { const int rid = bd.rid[on + pos]; bd.z4[ol + rid] = z4.get(s, rgl(s), true); bd.f[ol + rid] = f.get(s, rgl(s), true); }
    }
    if (tid == 0) {
        dsc[ND_RHO1] = rho1; dsc[ND_RHO2] = rho2; dsc[ND_RHO4] = rho4;
        dsc[ND_PREV_RHO1] = prev_rho1; dsc[ND_PREV_RHO2] = prev_rho2; dsc[ND_PREV_RHO4] = prev_rho4;
        dsc[ND_GAMMA] = gamma_val; dsc[ND_DI] = dI; dsc[ND_R4ET] = r4Et; dsc[ND_RCR] = rcr;
        dsc[ND_STD_OBJ] = std_obj; dsc[ND_CUR_OBJ] = cur_obj; dsc[ND_BEST_BIN_OBJ] = best_bin_obj;
        dsc[ND_SUM_FIX_OBJ] = sum_fix_obj; dsc[ND_FIX_OBJ] = fix_obj; dsc[ND_C1] = c1;
        dsc[ND_CVG1] = cvg1; dsc[ND_CVG2] = cvg2; dsc[ND_OBJ_VAL] = obj_val;
        dsc[ND_PREV_SUM] = prev_sum; dsc[ND_PREV_OBJ] = prev_obj;
        isc[NI_NLIVE] = n_live; isc[NI_RHO_UPDATED] = rhoUpdated; isc[NI_HIST_N] = hist_n;
        isc[NI_RET] = ret; isc[NI_STOP] = stop;
        isc[NI_PCG_TOTAL] = pcg_total; isc[NI_OUTER_TOTAL] = outer_total; isc[NI_LAST_PCG] = last_pcg;
        isc[NI_EXPR_READY] = expr_ready;
        if constexpr (DIRECT) isc[NI_H_VALID] = h_valid;
        if (l2f) isc[NI_ITER] = it;                                   // member `iter` (LPh:279), advanced by l2f only
        else isc[NI_PLAIN_ITER_P1] = it + 1;                          // LPcpp:1081
        for (int k = 0; k < LP_HIST; k++) bd.hist[(size_t)inst * LP_HIST + k] = LEAN ? s_hist[hbuf * LP_HIST + k] : h_reg[LEAN ? 0 : k];
    }
}

// x_iters export (get_x_iters_d, LPcpp:1616-1627): out[inst][r*ws + c] = x after iteration c of live variable r
// (live_pos[r] = storage position of the r-th live variable in original order).
__global__ void lp_pack_xiters_kernel(LpBatchDev bd, const int *live_pos, const int *rows, int ws, double *out,
                                      long out_stride) {
    const int inst = blockIdx.y;
    const int nr = rows[inst];
    const long total = (long)nr * ws;
    const double *xh = bd.xhist + (size_t)inst * bd.ws_cap * bd.NS;
    const int *li = live_pos + (size_t)inst * bd.NS;
    double *o = out + (size_t)inst * out_stride;
    for (long e = (long)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (long)gridDim.x * blockDim.x) {
        const int r = (int)(e / ws), c = (int)(e % ws);
        o[e] = xh[(size_t)c * bd.NS + li[r]];
    }
}

}  // namespace

// ------------------------------------------------------------------------------------------------
// launchers
// ------------------------------------------------------------------------------------------------
size_t lp_window_lds_bytes(int T, int NS, int LS, int ZS, int HL, int HLD) {
    return LdsLayout(NS, LS, ZS, lp_is_lean(T, NS / T), HL, HLD).total;
}

bool lp_direct_supported(int T, int EPT) { return T == 512 && EPT == 1; }     // the default geometry of every n <= 512 batch
bool lp_log_supported(int T, int EPT) { return T == 512 && (EPT == 1 || EPT == 2 || EPT == 4); }

// (threads, slots per thread) -> per-slot register capacities of the row-task / column gather lists.
// (512 x 4: the load-balanced deal of lpbox_capi.hip puts each wave's longest block of row tasks / columns into slot 0, so slot 0 alone
//  gets the longer register lists -- 12 / 12 / 8 -- and the tail path of a longer list through the LDS index arrays, two dependent LDS
//  round trips per chunk, stays out of the PCG loop: 88.8 -> 85.1 us per iteration.  12 entries in two slots spill: 96.7 us.)
#define LP_DISPATCH(KERNEL_CALL)                                                                                                          \
    if (T == 256 && EPT == 1) { KERNEL_CALL(256, 1, (Caps<12>), (Caps<12>), (Caps<8>)) }                                                     \
    else if (T == 256 && EPT == 2) { KERNEL_CALL(256, 2, (Caps<12, 12>), (Caps<12, 8>), (Caps<8, 4>)) }                                       \
    else if (T == 256 && EPT == 4) { KERNEL_CALL(256, 4, (Caps<12, 12, 12, 12>), (Caps<24, 8, 8, 8>), (Caps<0, 0, 0, 0>)) }                  \
    else if (T == 256 && EPT == 8) { KERNEL_CALL(256, 8, (Caps<8, 8, 8, 8, 8, 8, 8, 8>), (Caps<8, 8, 8, 8, 4, 4, 4, 4>), (Caps<0, 0, 0, 0, 0, 0, 0, 0>)) } \
    else if (T == 512 && EPT == 1) { KERNEL_CALL(512, 1, (Caps<12>), (Caps<12>), (Caps<8>)) }                                                \
    else if (T == 512 && EPT == 2) { KERNEL_CALL(512, 2, (Caps<12, 12>), (Caps<12, 8>), (Caps<8, 4>)) }                                      \
    else if (T == 512 && EPT == 4) { KERNEL_CALL(512, 4, (Caps<12, 8, 8, 8>), (Caps<12, 8, 8, 8>), (Caps<8, 4, 4, 4>)) }                        \
    else if (T == 1024 && EPT == 1) { KERNEL_CALL(1024, 1, (Caps<8>), (Caps<8>), (Caps<8>)) }                                                \
    else if (T == 1024 && EPT == 2) { KERNEL_CALL(1024, 2, (Caps<8, 8>), (Caps<8, 8>), (Caps<4, 4>)) }                                     \
    else return hipErrorInvalidConfiguration;

#define LP_UNPAREN(...) __VA_ARGS__

hipError_t lp_launch_init(const LpBatchDev &bd, int T, int EPT, const double *f_org, const double *c1_init,
                          const uint8_t *live_init, hipStream_t s) {
#define CALL_INIT(TT, EE, RR, CCC, HHH) hipLaunchKernelGGL((lp_init_kernel<TT, EE>), dim3(bd.B), dim3(TT), 0, s, bd, f_org, c1_init, live_init);
    LP_DISPATCH(CALL_INIT)
#undef CALL_INIT
    return hipGetLastError();
}

hipError_t lp_launch_window(const LpBatchDev &bd, int T, int EPT, size_t lds, int iter_start, int iter_end, int l2f,
                            hipStream_t s, bool direct, bool log) {
    if (log) {
        if (direct || !lp_log_supported(T, EPT) || bd.logbuf == nullptr) return hipErrorInvalidConfiguration;
#define CALL_LOG(EE, RR, CCC, HHH)                                                                             \
    {                                                                                                          \
        auto kfn = lp_window_kernel<512, EE, LP_UNPAREN RR, LP_UNPAREN CCC, LP_UNPAREN HHH, false, true>;      \
        hipError_t e = hipFuncSetAttribute((const void *)kfn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); \
        if (e != hipSuccess) return e;                                                                         \
        hipLaunchKernelGGL(kfn, dim3(bd.B), dim3(512), lds, s, bd, iter_start, iter_end, l2f);                 \
    }
        if (EPT == 1) CALL_LOG(1, (Caps<12>), (Caps<12>), (Caps<8>))
        else if (EPT == 2) CALL_LOG(2, (Caps<12, 12>), (Caps<12, 8>), (Caps<8, 4>))
        else CALL_LOG(4, (Caps<12, 8, 8, 8>), (Caps<12, 8, 8, 8>), (Caps<8, 4, 4, 4>))
#undef CALL_LOG
        return hipGetLastError();
    }
    if (direct) {
        if (!lp_direct_supported(T, EPT) || bd.H == nullptr || bd.HL <= 0 || bd.HL > 128) return hipErrorInvalidConfiguration;
#define CALL_DIRECT(TT)                                                                                        \
    {                                                                                                          \
        auto kfn = lp_window_kernel<TT, 1, Caps<12>, Caps<12>, Caps<8>, true>;                                 \
        hipError_t e = hipFuncSetAttribute((const void *)kfn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); \
        if (e != hipSuccess) return e;                                                                         \
        hipLaunchKernelGGL(kfn, dim3(bd.B), dim3(TT), lds, s, bd, iter_start, iter_end, l2f);                  \
    }
        CALL_DIRECT(512)
#undef CALL_DIRECT
        return hipGetLastError();
    }
#define CALL_WIN(TT, EE, RR, CCC, HHH)                                                                             \
    {                                                                                                          \
        auto kfn = lp_window_kernel<TT, EE, LP_UNPAREN RR, LP_UNPAREN CCC, LP_UNPAREN HHH>;                                    \
        hipError_t e = hipFuncSetAttribute((const void *)kfn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); \
        if (e != hipSuccess) return e;                                                                         \
        hipLaunchKernelGGL(kfn, dim3(bd.B), dim3(TT), lds, s, bd, iter_start, iter_end, l2f);                  \
    }
    LP_DISPATCH(CALL_WIN)
#undef CALL_WIN
    return hipGetLastError();
}

hipError_t lp_launch_pack_xiters(const LpBatchDev &bd, const int *live_pos, const int *rows, int ws, double *out,
                                 long out_stride, hipStream_t s) {
    dim3 grid(8, bd.B);
    hipLaunchKernelGGL(lp_pack_xiters_kernel, grid, dim3(256), 0, s, bd, live_pos, rows, ws, out, out_stride);
    return hipGetLastError();
}
