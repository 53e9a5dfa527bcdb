// lpbox_big_kernels.hip -- gfx950 kernels of the LARGE-instance LP path (one LP, n up to ~1e6, variable-sharded).
// Same arithmetic as lp_window_kernel (LPcpp:766-1095, ADMM_lp_iters), cut into kernels at the grid-wide dependencies:
//   prep -> [fin] -> y -> rhs_cols -> rows -> [AR q] -> resid -> [fin] ->
//   K x { rows(p) -> [AR q] -> pcg_cols -> [fin] -> pcg_upd -> [fin] } -> post -> [fin] -> rows(x) -> [AR q] -> z4
// ([fin] = reduce the workgroup partials to red[]; on more than one rank the host all-reduces red[] / q after it.)
// The PCG search direction p = z + beta p is recomputed on the fly for the gathered columns inside rows(p) (bit-identical
// expression), so a PCG iteration needs no separate p-update launch.  Rows and columns are summed by one lane in ascending
// index order; no FMA contraction; IEEE divide / sqrt.
#include "lpbox_big.h"
#include "lpbox_dev_common.h"

#include <float.h>
#include <algorithm>

namespace {

constexpr int T = BIG_T;
#define LEADER (blockIdx.x == 0 && threadIdx.x == 0)

// Streamed once per launch (index lists, pointers, the n-vectors a column owns).  Measured: the non-temporal hint on these makes
// the chain SLOWER (1.52 -> 1.83 ms per iteration at n = 1e6: a lane's index run shares its 128-byte line with its neighbours'
// and with its own next loads, which the hint gives up), so it is a diagnostic knob only (LPBOX_BIG_NT).
#ifndef LPBOX_BIG_NT
template <typename V> __device__ __forceinline__ V ld_stream(const V *p) { return *p; }
template <typename V> __device__ __forceinline__ void st_stream(V *p, V v) { *p = v; }
#else
template <typename V> __device__ __forceinline__ V ld_stream(const V *p) { return __builtin_nontemporal_load(p); }
template <typename V> __device__ __forceinline__ void st_stream(V *p, V v) { __builtin_nontemporal_store(v, p); }
#endif

__device__ __forceinline__ void forward_state(const BigDev &d, int in, int out) {
    if (LEADER) d.st[out] = d.st[in];
}

// Workgroup partials of NV values -> d.part; big_k_fin reduces them to d.red in a launch of its own.  Measured alternative: the
// workgroup that arrives last (ticket counter, __threadfence before and after) reduces them inside the producing kernel -- 41
// launches per iteration fewer, but 3.05 instead of 1.52 ms per iteration at n = 1e6: a device-scope fence per workgroup writes the
// XCD's L2 back each time.  Dropped.
__device__ __forceinline__ double *part_ptr(const BigDev &d, int phase) { return d.part + (size_t)phase * BIG_NPART * d.Gs; }

template <int NV>
__device__ __forceinline__ void store_partials(const BigDev &d, int phase, double (&v)[NV], double *red, int &parity) {
    block_sum<T, NV>(v, red, parity);
    if (threadIdx.x == 0) {
        double *pp = part_ptr(d, phase);
#pragma unroll
        for (int k = 0; k < NV; k++) pp[(size_t)k * d.Gs + blockIdx.x] = v[k];
    }
}

// red[v] = tree over the G workgroup partials of value v (second level of the fixed reduction order)
__global__ void __launch_bounds__(T) big_k_fin(BigDev d, int nv, int phase) {
    __shared__ double red[2 * RED_MAXV * RED_MAXW];
    int parity = 0;
    fin_reduce<T>(part_ptr(d, phase), d.G, nv, d.red + phase * BIG_NPART, red, parity, d.Gs);
}

// The totals of the first NV of the NVP values a phase holds, for every thread of the calling workgroup (a uniform call: it contains
// barriers).  Route with a reduction launch: big_k_fin (+ the cross-rank sum) has left them in d.red.  Folded route: this workgroup
// adds the partials up itself -- per rank the order of big_k_fin (thread t: partials t, t + T, ... ascending, then the block tree),
// the rank totals in rank order like big_k_rank_sum -- so both routes give the same bits.  All loads of a rank are issued before
// the first addition.
template <int NV, int NVP>
__device__ __forceinline__ void get_red(const BigDev &d, int phase, double (&out)[NV], double *red, int &parity) {
    static_assert(NV <= NVP, "a phase holds NVP values");
    if (!d.fold) {
#pragma unroll
        for (int k = 0; k < NV; k++) out[k] = d.red[phase * BIG_NPART + k];
        return;
    }
    const int W = d.gathered ? d.W : 1;
    for (int r = 0; r < W; r++) {
        const double *base = d.gathered ? d.gpart + ((size_t)phase * W * BIG_NPART + (size_t)r * NVP) * d.Gs : part_ptr(d, phase);
        const int G = d.gathered ? d.Gr[r] : d.G;
        double t[NV][BIG_FOLD_U];
#pragma unroll
        for (int k = 0; k < NV; k++)
#pragma unroll
            for (int u = 0; u < BIG_FOLD_U; u++) { const int e = threadIdx.x + u * T; t[k][u] = e < G ? base[(size_t)k * d.Gs + e] : 0.0; }
        double a[NV];
#pragma unroll
        for (int k = 0; k < NV; k++) {
            double acc = 0.0;
#pragma unroll
            for (int u = 0; u < BIG_FOLD_U; u++) acc = (int)threadIdx.x + u * T < G ? acc + t[k][u] : acc;
            a[k] = acc;
        }
        block_sum<T, NV>(a, red, parity);
#pragma unroll
        for (int k = 0; k < NV; k++) out[k] = r == 0 ? a[k] : out[k] + a[k];
    }
}

__global__ void __launch_bounds__(T) big_k_init(BigDev d, double c1) {          // ADMM_lp_iters_init LPcpp:489-763
    __shared__ double red[2 * RED_MAXV * RED_MAXW];
    int parity = 0;
    double pb[1] = {0.0};
    for (int s = 0; s < d.EPT; s++) {
        const int j = blockIdx.x * (T * d.EPT) + s * T + threadIdx.x;
        double c = 0.0;
        if (j < d.n_loc) {
            d.x[j] = 1.0; d.z1[j] = 0.0; d.z2[j] = 0.0; d.pd[j] = 0.0; d.dinv[j] = 1.0;   // :583-586, :616-617
            d.y1[j] = 1.0; d.y2[j] = 1.0; d.gsrc[j] = 1.0; d.xt[j] = 1.0; d.live[j] = 1;
            d.r[j] = 0.0; d.z[j] = 0.0; d.tmp[j] = 0.0; d.p0[j] = 0.0; d.p1[j] = 0.0; d.rhs[j] = 0.0;
            c = d.b[j] * 1.0;                                                    // best_bin_obj = b.dot(x0) (:727)
        }
        pb[0] = pb[0] + c;
    }
    if (blockIdx.x < d.G) store_partials<1>(d, BIG_PH_X, pb, red, parity);
    for (int s = 0; s < d.EPTl; s++) {
        const int i = blockIdx.x * (T * d.EPTl) + s * T + threadIdx.x;
        if (i < d.l) { d.z4[i] = 0.0; d.y3[i] = 0.0; d.fz[i] = make_double2(0.0, 0.0); d.Ex[i] = 0.0; }   // :650
    }
    if (LEADER) {
        BigState *s = d.st;
        memset(s, 0, sizeof(BigState));
        s->rho1 = s->rho2 = s->rho4 = s->prev_rho1 = s->prev_rho2 = s->prev_rho4 = LP_RHO0;   // :623-630
        s->gamma_val = LP_GAMMA0; s->std_obj = 1.0; s->rhoUpdated = 1; s->c1 = c1;
        s->n_live_lo = (int)(d.n_glob & 0x7fffffff); s->n_live_hi = (int)(d.n_glob >> 31);
        d.st[1] = d.st[0];
    }
}

__global__ void big_k_init2(BigDev d) {       // after the partial of b.x0 has been reduced: best_bin_obj (:727)
    d.st[0].best_bin_obj = d.red[BIG_PH_X * BIG_NPART];
    d.st[1].best_bin_obj = d.red[BIG_PH_X * BIG_NPART];
}

__global__ void big_k_set_window(BigDev d, int in, int out, int iter_start, int iter_end, int l2f) {
    d.st[out] = d.st[in];
    BigState *s = d.st + out;
    s->iter = iter_start; s->iter_start = iter_start; s->iter_end = iter_end; s->ret = 0; s->stop = LP_STOP_NONE; s->halt = BIG_HALT_NONE;
    s->l2f = l2f & 1; s->rec = (l2f >> 1) & 1; s->cc = 0;
}

__global__ void big_k_resume(BigDev d, int in, int out, int reset_pcg_max) {
    d.st[out] = d.st[in];
    if (d.st[out].halt == BIG_HALT_PCG_MORE) d.st[out].halt = BIG_HALT_NONE;
    if (reset_pcg_max) d.st[out].pcg_max = 0;
}

// finalise the previous iteration from red[0..5) (x.x, |x-y1|^2, |x-y2|^2, b.x, b.round(x)), then partial ||x+z2/rho2-1/2||^2
__global__ void __launch_bounds__(T) big_k_prep(BigDev d, int in, int out, int do_prep) {
    __shared__ double red[2 * RED_MAXV * RED_MAXW];
    int parity = 0;
    const BigState *si = d.st + in;
    const int halt0 = si->halt, have_prev = si->have_prev, it = si->iter, iter_end = si->iter_end;
    double rho2 = si->rho2;
    const bool fin = !halt0 && have_prev;
    if (fin && (it + 1) % LP_RHO_STEP == 0) rho2 = LP_LEARNING_FACT * rho2;
    const int next_iter = fin ? it + 1 : it;
    const bool will_prep = do_prep && !halt0 && next_iter < iter_end;
    double e5[5] = {0.0, 0.0, 0.0, 0.0, 0.0};
    if (blockIdx.x == 0 && fin) get_red<5, 5>(d, BIG_PH_E, e5, red, parity);    // uniform over workgroup 0
    if (LEADER) {
        d.st[out] = *si;
        BigState *s = d.st + out;
        if (fin) {
            s->have_prev = 0;
            const double xn = sqrt(e5[0]);
            const double t0 = (xn < 2.2204e-16) ? 2.2204e-16 : xn;
            s->cvg1 = sqrt(e5[1]) / t0; s->cvg2 = sqrt(e5[2]) / t0;            // :931-933
            bool stopped = false;
            if (s->cvg1 <= LP_STOP_THRESHOLD && s->cvg2 <= LP_STOP_THRESHOLD && (s->l2f || it != s->iter_start)) {   // :934 / :1503
                if (s->l2f) s->ret = 1;                                          // :1505 (plain loop: ret stays 0)
                s->stop = LP_STOP_Y1Y2; stopped = true;
            } else {
                if ((it + 1) % LP_RHO_STEP == 0) {                                // :951-970
                    s->prev_rho1 = s->rho1; s->prev_rho2 = s->rho2; s->prev_rho4 = s->rho4;
                    s->rho1 = LP_LEARNING_FACT * s->rho1; s->rho2 = LP_LEARNING_FACT * s->rho2; s->rho4 = LP_LEARNING_FACT * s->rho4;
                    const double g = s->gamma_val * LP_GAMMA_FACTOR;
                    s->gamma_val = g < 1.0 ? 1.0 : g;
                    s->rhoUpdated = 1; s->rcr = LP_LEARNING_FACT - 1.0;
                }
                s->obj_val = e5[3];                                               // :972
                int hn = s->hist_n;
                if (hn < LP_HIST) s->hist[hn] = s->obj_val;
                else { for (int k = 0; k < LP_HIST - 1; k++) s->hist[k] = s->hist[k + 1]; s->hist[LP_HIST - 1] = s->obj_val; }
                if (hn < 0x3fffffff) hn++;
                s->hist_n = hn;
                if (hn >= LP_HIST) {                                              // :459-469, :358-377
                    double mean = 0;
                    for (int k = 0; k < LP_HIST; k++) mean += s->hist[k];
                    mean /= (double)LP_HIST;
                    double dev = 0;
                    for (int k = 0; k < LP_HIST; k++) dev += (s->hist[k] - mean) * (s->hist[k] - mean);
                    dev /= (double)(LP_HIST - 1);
                    const double sd = (dev == 0) ? 0.0 : sqrt(dev);
                    s->std_obj = sd / fabs(s->hist[LP_HIST - 1]);
                }
                if (s->std_obj <= LP_STD_THRESHOLD) { s->ret = 1; s->stop = LP_STOP_OBJSTD; stopped = true; }   // :977
                else {
                    s->cur_obj = e5[4];                                           // :1001-1003
                    if (s->best_bin_obj >= s->cur_obj) s->best_bin_obj = s->cur_obj;
                }
            }
            if (stopped) { s->halt = BIG_HALT_STOP; s->plain_iter_p1 = it + 1; }
            else s->iter = it + 1;
        }
        if (!s->halt && s->iter >= s->iter_end) { s->halt = BIG_HALT_WINDOW; s->plain_iter_p1 = s->iter + 1; }
        if (!s->halt && do_prep) s->phase = 1;
    }
    if (!will_prep || blockIdx.x >= d.G) return;
    double pa[1] = {0.0};
    for (int q = 0; q < d.EPT; q++) {
        const int j = blockIdx.x * (T * d.EPT) + q * T + threadIdx.x;
        double c = 0.0;
        if (j < d.n_loc && d.live[j]) { const double u = (d.x[j] + d.z2[j] / rho2) - 0.5; c = u * u; }
        pa[0] = pa[0] + c;
    }
    store_partials<1>(d, BIG_PH_A, pa, red, parity);
}

// y1, y2, expression refresh (:831-866), rhs base, PCG start x0 = y1; for the rows: y3 = max(0, f - Ex - z4/rho4), fz = (f - y3, z4)
__global__ void __launch_bounds__(T) big_k_y(BigDev d, int in, int out) {
    __shared__ double red[2 * RED_MAXV * RED_MAXW];
    int parity = 0;
    const BigState *si = d.st + in;
    if (si->halt || si->phase != 1) { forward_state(d, in, out); return; }
    const double rho1 = si->rho1, rho2 = si->rho2, rho4 = si->rho4, c1 = si->c1;
    const int it = si->iter, rhoUpdated = si->rhoUpdated;
    double dI = si->dI, r4Et = si->r4Et;
    const bool first = it == 0, refresh = it != 0 && rhoUpdated;
    const double inc = si->rcr * (si->prev_rho1 + si->prev_rho2), inc4 = si->rcr * si->prev_rho4;
    if (first) { dI = 0.0; dI += rho1 + rho2; r4Et = rho4; }                     // update_expression(0) :2289-2404
    if (refresh) { dI += inc; r4Et = LP_LEARNING_FACT * r4Et; }                  // :851-866
    double a1[1];
    get_red<1, 1>(d, BIG_PH_A, a1, red, parity);
    const double c2 = 2 * sqrt(a1[0]);
    if (blockIdx.x < d.G)
        for (int q = 0; q < d.EPT; q++) {
            const int j = blockIdx.x * (T * d.EPT) + q * T + threadIdx.x;
            if (j >= d.n_loc) continue;
            const double x = d.x[j], z1 = d.z1[j], z2 = d.z2[j];
            const double t = x + z1 / rho1;
            const double y1 = t > 1 ? 1 : (t < 0 ? 0 : t);                       // :806-809
            double y2 = (x + z2 / rho2) - 0.5;                                   // :815-818
            y2 = y2 * c1 / c2 + 0.5;
            d.y1[j] = y1; d.y2[j] = y2;
            double pd = d.pd[j];
            const double Esq = (double)(d.cptr[j + 1] - d.cptr[j]);
            if (first) { pd = dI; pd += rho4 * Esq; }
            if (refresh) { pd += inc; pd += inc4 * Esq; }
            d.pd[j] = pd;
            if (rhoUpdated) d.dinv[j] = (pd != 0.0) ? 1.0 / pd : 1.0;            // :883-890
            d.rhs[j] = (rho1 * y1 + rho2 * y2) - ((d.b[j] + z1) + z2);            // :872
            d.gsrc[j] = d.live[j] ? y1 : 0.0;                                     // x_sol = y1 (:892); fixed columns are gone from E
        }
    if (blockIdx.x < d.Gl)
        for (int q = 0; q < d.EPTl; q++) {
            const int i = blockIdx.x * (T * d.EPTl) + q * T + threadIdx.x;
            if (i >= d.l) continue;
            const double f = d.f[i];
            const double z4 = d.z4[i];
            const double v = f - d.Ex[i] - z4 / rho4;                             // :824-828
            const double y3 = v < 0 ? 0 : v;
            d.y3[i] = y3; d.fz[i] = make_double2(f - y3, z4);                     // what rhs_cols gathers per entry: ONE 16-byte element
        }
    if (LEADER) {
        d.st[out] = *si;
        BigState *s = d.st + out;
        s->dI = dI; s->r4Et = r4Et; s->rhoUpdated = 0; s->expr_ready = 1;
    }
}

__global__ void __launch_bounds__(T) big_k_rhs_cols(BigDev d, int in, int out) {   // :874-877
    const BigState *si = d.st + in;
    if (si->halt || si->phase != 1) { forward_state(d, in, out); return; }
    const double r4Et = si->r4Et;
    for (int q = 0; q < d.EPT; q++) {
        const int j = blockIdx.x * (T * d.EPT) + q * T + threadIdx.x;
        if (j >= d.n_loc) continue;
        double tA = 0.0, tB = 0.0;
        const int k1 = d.cptr[j + 1];
        int k = d.cptr[j];
        for (; k + 4 <= k1; k += 4) {                      // 4 gathers in flight, additions in column order
            const int i0 = d.crow[k], i1 = d.crow[k + 1], i2 = d.crow[k + 2], i3 = d.crow[k + 3];
            const double2 e0 = d.fz[i0], e1 = d.fz[i1], e2 = d.fz[i2], e3 = d.fz[i3];     // (f - y3, z4) of the row
            tA += r4Et * e0.x; tB += e0.y; tA += r4Et * e1.x; tB += e1.y; tA += r4Et * e2.x; tB += e2.y; tA += r4Et * e3.x; tB += e3.y;
        }
        for (; k < k1; k++) { const double2 e = d.fz[d.crow[k]]; tA += r4Et * e.x; tB += e.y; }
        double r_ = d.rhs[j];
        r_ += tA;
        r_ -= tB;
        d.rhs[j] = r_;
    }
    forward_state(d, in, out);
}

// q_part = E[:, shard] * v over this rank's columns.  mode 0: v = gsrc.  mode 1: v = the PCG search direction, with the exit
// test / beta of the previous PCG iteration evaluated here (LPcpp:296-319).
// Row i of E[:, shard] * v when the table being gathered does not fit an XCD's L2 (n_loc * 16 B > the slice size, i.e. the
// single-GPU run of a 1e6-variable LP): the entries are stored SLICE-major -- slice ph holds the columns [ph * SW, (ph + 1) * SW),
// and inside a slice the rows follow each other, each with its entries in ascending column order (d.rptr[ph * l + i] is the start
// of the run of (slice ph, row i); the runs are consecutive, so the next pointer is its end).  Every thread walks the slices in
// order, which is still the ascending column order of its row: the sum is bit for bit the one of the unsliced loop.  What
// changes is WHEN a column is touched: all workgroups of the launch are co-resident and progress at the same pace, so at any
// time the whole chip gathers from ONE slice of (z, p) -- 1-2 MB, resident in every XCD's 4 MB L2 -- instead of drawing random
// 128-byte lines of a 16 MB table from the Infinity Cache at 12 % utilisation.  Runs are short (~1.5 entries), so the loop is
// software-pipelined over the slices: pointers two slices ahead, indices one slice ahead, gathers of the current slice.
constexpr int SLU = 6;   // entries of a run handled without a loop (longer runs: remainder loop)
template <bool ZP>
__device__ __forceinline__ double row_sum_sliced(const BigDev &d, int i, const double *src, const double2 *zp, double beta) {
    const int P = d.P;
    const size_t l = (size_t)d.l;
    const int *sp = d.rptr + i;
    double acc = 0.0;
    int ka = sp[0], kae = sp[1];                       // current slice
    int kb = 0, kbe = 0;                               // next slice
    if (P > 1) { kb = sp[l]; kbe = sp[l + 1]; }
    int c[SLU];
#pragma unroll
    for (int u = 0; u < SLU; u++) c[u] = ka + u < kae ? ld_stream(d.rcol + ka + u) : -1;
    for (int ph = 0; ph < P; ph++) {
        double vx[SLU], vy[SLU];
#pragma unroll
        for (int u = 0; u < SLU; u++) {
            vx[u] = 0.0; vy[u] = 0.0;
            if (c[u] >= 0) {
                if constexpr (ZP) { const double2 t = zp[c[u]]; vx[u] = t.x; vy[u] = t.y; }
                else vx[u] = src[c[u]];
            }
        }
        int cn[SLU];
#pragma unroll
        for (int u = 0; u < SLU; u++) cn[u] = kb + u < kbe ? ld_stream(d.rcol + kb + u) : -1;      // ph + 1 < P, else the run is empty
        int kc = 0, kce = 0;
        if (ph + 2 < P) { kc = sp[(size_t)(ph + 2) * l]; kce = sp[(size_t)(ph + 2) * l + 1]; }
#pragma unroll
        for (int u = 0; u < SLU; u++) {
            if constexpr (ZP) { const double t = vx[u] + beta * vy[u]; acc = c[u] >= 0 ? acc + t : acc; }
            else acc = c[u] >= 0 ? acc + vx[u] : acc;
        }
        for (int k = ka + SLU; k < kae; k++) {         // rare: a run longer than SLU entries
            const int cc = d.rcol[k];
            if constexpr (ZP) { const double2 t = zp[cc]; acc += t.x + beta * t.y; }
            else acc += src[cc];
        }
#pragma unroll
        for (int u = 0; u < SLU; u++) c[u] = cn[u];
        ka = kb; kae = kbe; kb = kc; kbe = kce;
    }
    return acc;
}

__global__ void __launch_bounds__(T) big_k_rows(BigDev d, int in, int out, int mode) {
    __shared__ double red[2 * RED_MAXV * RED_MAXW];
    int parity = 0;
    const BigState *si = d.st + in;
    if (si->halt) { forward_state(d, in, out); return; }
    double beta = 0.0;
    bool first = false;
    const double *pold = nullptr;
    if (mode == 1) {
        if (si->phase != 2 || si->pcg_done) { forward_state(d, in, out); return; }
        const int k = si->pcg_k;
        double threshold = si->threshold, absNew = si->absNew, rhsNorm2 = si->rhsNorm2;
        bool done = false; int zero_x = 0;
        first = k == 0;
        if (first) {
            double b3[3];
            get_red<3, 3>(d, BIG_PH_B, b3, red, parity);
            rhsNorm2 = b3[0];
            if (rhsNorm2 == 0) { done = true; zero_x = 1; }                      // :273-278
            else {
                double thr = LP_PCG_TOL * LP_PCG_TOL * rhsNorm2;                 // :281
                if (thr < DBL_MIN) thr = DBL_MIN;
                threshold = thr;
                if (b3[1] < thr) done = true;                                    // :284
                absNew = b3[2];
            }
        } else {
            double d2[2];
            get_red<2, 2>(d, BIG_PH_D, d2, red, parity);
            if (d2[0] < threshold || k >= LP_PCG_MAXITERS) done = true;          // :309-312, :296
            else { const double absOld = absNew; absNew = d2[1]; beta = absNew / absOld; }   // :316-318
        }
        if (LEADER) {
            d.st[out] = *si;
            BigState *s = d.st + out;
            s->threshold = threshold; s->absNew = absNew; s->rhsNorm2 = rhsNorm2; s->beta = beta;
            s->pcg_done = done ? 1 : 0; s->pcg_first = zero_x;
        }
        if (done) return;
        pold = ((k - 1) & 1) ? d.p1 : d.p0;
    } else forward_state(d, in, out);
    const double *gs = d.gsrc, *p0 = d.p0;
    (void)pold;
    const bool lean = d.lean && mode == 1;
    const int rowgrid = d.P > 1 ? d.Glr : d.Gl;
    if (lean && (int)blockIdx.x < d.G) {
        // comm-lean PCG: the search direction p = z + beta p of this iteration for the workgroup's own variables (what pcg_cols does in the
        // default mode) and the workgroup partial of p.p, which rides with the q exchange
        const int k = si->pcg_k;
        double *pnew = (k & 1) ? d.p1 : d.p0;
        double pp[1] = {0.0};
        for (int q = 0; q < d.EPT; q++) {
            const int j = blockIdx.x * (T * d.EPT) + q * T + threadIdx.x;
            double c = 0.0;
            if (j < d.n_loc) {
                double pj;
                if (first) pj = p0[j];
                else { pj = d.z[j] + beta * pold[j]; pnew[j] = pj; }                // p = z + beta p (:319)
                c = d.live[j] ? pj * pj : 0.0;
            }
            pp[0] = pp[0] + c;
        }
        block_sum<T, 1>(pp, red, parity);
        if (threadIdx.x == 0) d.lsmall[blockIdx.x] = pp[0];
    }
    if ((int)blockIdx.x >= rowgrid) return;
    const bool lean_qq = lean && !d.gathered;        // no exchange: q is final here, so its squares are summed here too
    double qq[1] = {0.0};
    if (d.P > 1) {                                   // column-sliced rows (see row_sum_sliced): one row per thread
        const int i = blockIdx.x * T + threadIdx.x;
        if (i < d.l) {
            double qi;
            if (mode == 0 || first) qi = row_sum_sliced<false>(d, i, mode == 0 ? gs : p0, nullptr, 0.0);
            else qi = row_sum_sliced<true>(d, i, nullptr, d.zp, beta);
            d.q[i] = qi;
            qq[0] = qq[0] + qi * qi;
        }
        if (lean_qq) { block_sum<T, 1>(qq, red, parity); if (threadIdx.x == 0) d.lsmall[d.Gs + blockIdx.x] = qq[0]; }
        return;
    }
    for (int s = 0; s < d.EPTl; s++) {
        const int i = blockIdx.x * (T * d.EPTl) + s * T + threadIdx.x;
        if (i >= d.l) continue;
        double acc = 0.0;
        const int k1 = d.rptr[i + 1];
        int k = d.rptr[i];
        // the row is summed in ascending column order by this one lane; the gathers of 8 entries are issued together so that
        // their latencies overlap (the additions stay sequential: same rounding as the one-at-a-time loop)
        if (mode == 0 || first) {
            const double *src = mode == 0 ? gs : p0;
            for (; k + 8 <= k1; k += 8) {
                int c[8]; double v[8];
#pragma unroll
                for (int u = 0; u < 8; u++) c[u] = d.rcol[k + u];
#pragma unroll
                for (int u = 0; u < 8; u++) v[u] = src[c[u]];
#pragma unroll
                for (int u = 0; u < 8; u++) acc += v[u];
            }
            for (; k < k1; k++) acc += src[d.rcol[k]];
        } else {
            for (; k + 8 <= k1; k += 8) {
                int c[8]; double2 v[8];
#pragma unroll
                for (int u = 0; u < 8; u++) c[u] = d.rcol[k + u];
#pragma unroll
                for (int u = 0; u < 8; u++) v[u] = d.zp[c[u]];
#pragma unroll
                for (int u = 0; u < 8; u++) acc += v[u].x + beta * v[u].y;
            }
            for (; k < k1; k++) { const double2 v = d.zp[d.rcol[k]]; acc += v.x + beta * v.y; }
        }
        d.q[i] = acc;
        qq[0] = qq[0] + acc * acc;
    }
    if (lean_qq) { block_sum<T, 1>(qq, red, parity); if (threadIdx.x == 0) d.lsmall[d.Gs + blockIdx.x] = qq[0]; }
}

__global__ void __launch_bounds__(T) big_k_resid(BigDev d, int in, int out) {       // :267-294
    __shared__ double red[2 * RED_MAXV * RED_MAXW];
    int parity = 0;
    const BigState *si = d.st + in;
    if (si->halt || si->phase != 1) { forward_state(d, in, out); return; }
    const double dI = si->dI, r4Et = si->r4Et;
    double pb[3] = {0.0, 0.0, 0.0};
    for (int q = 0; q < d.EPT; q++) {
        const int j = blockIdx.x * (T * d.EPT) + q * T + threadIdx.x;
        double c0 = 0.0, c1 = 0.0, c2 = 0.0;
        if (j < d.n_loc) {
            double t = 0.0;
            const int k1 = d.cptr[j + 1];
            int k = d.cptr[j];
            for (; k + 4 <= k1; k += 4) {
                const double v0 = d.q[d.crow[k]], v1 = d.q[d.crow[k + 1]], v2 = d.q[d.crow[k + 2]], v3 = d.q[d.crow[k + 3]];
                t += r4Et * v0; t += r4Et * v1; t += r4Et * v2; t += r4Et * v3;
            }
            for (; k < k1; k++) t += r4Et * d.q[d.crow[k]];
            const double y1 = d.y1[j];
            double Mx = 0.0;
            Mx += dI * (1.0 * y1);
            Mx += t;
            const double rhs = d.rhs[j];
            const double r = rhs - Mx;
            const double p = d.dinv[j] * r;
            const bool lv = d.live[j];
            d.xt[j] = y1; d.r[j] = r; d.p0[j] = lv ? p : 0.0;
            if (lv) { c0 = rhs * rhs; c1 = r * r; c2 = r * p; }
        }
        pb[0] = pb[0] + c0; pb[1] = pb[1] + c1; pb[2] = pb[2] + c2;
    }
    store_partials<3>(d, BIG_PH_B, pb, red, parity);
    if (LEADER) { d.st[out] = *si; d.st[out].pcg_k = 0; d.st[out].pcg_done = 0; d.st[out].pcg_first = 0; d.st[out].phase = 2; }
}

__global__ void __launch_bounds__(T) big_k_pcg_cols(BigDev d, int in, int out) {    // tmp = M p, partial p.tmp (:298-300)
    __shared__ double red[2 * RED_MAXV * RED_MAXW];
    int parity = 0;
    const BigState *si = d.st + in;
    if (si->halt || si->phase != 2) { forward_state(d, in, out); return; }
    if (si->pcg_done) {
        if (si->pcg_first)                                                       // rhs == 0: x := 0 (:273-278)
            for (int q = 0; q < d.EPT; q++) { const int j = blockIdx.x * (T * d.EPT) + q * T + threadIdx.x; if (j < d.n_loc) d.xt[j] = 0.0; }
        if (LEADER) { d.st[out] = *si; d.st[out].pcg_first = 0; }
        return;
    }
    const int k = si->pcg_k;
    const double dI = si->dI, r4Et = si->r4Et, beta = si->beta;
    const bool first = k == 0;
    const double *pold = ((k - 1) & 1) ? d.p1 : d.p0;
    double *pnew = (k & 1) ? d.p1 : d.p0;
    double pc[1] = {0.0};
    for (int q = 0; q < d.EPT; q++) {
        const int j = blockIdx.x * (T * d.EPT) + q * T + threadIdx.x;
        double c = 0.0;
        if (j < d.n_loc) {
            double pj;
            if (first) pj = ld_stream(d.p0 + j);
            else { pj = ld_stream(d.z + j) + beta * ld_stream(pold + j); st_stream(pnew + j, pj); }   // p = z + beta p (:319)
            double t = 0.0;
            const int k1 = d.cptr[j + 1];
            int kk = d.cptr[j];
            for (; kk + 4 <= k1; kk += 4) {
                const int r0 = ld_stream(d.crow + kk), r1 = ld_stream(d.crow + kk + 1), r2 = ld_stream(d.crow + kk + 2), r3 = ld_stream(d.crow + kk + 3);
                const double v0 = d.q[r0], v1 = d.q[r1], v2 = d.q[r2], v3 = d.q[r3];
                t += r4Et * v0; t += r4Et * v1; t += r4Et * v2; t += r4Et * v3;
            }
            for (; kk < k1; kk++) t += r4Et * d.q[ld_stream(d.crow + kk)];
            double Mp = 0.0;
            Mp += dI * (1.0 * pj);
            Mp += t;
            st_stream(d.tmp + j, Mp);
            c = d.live[j] ? pj * Mp : 0.0;
        }
        pc[0] = pc[0] + c;
    }
    store_partials<1>(d, BIG_PH_C, pc, red, parity);
    forward_state(d, in, out);
}

__global__ void __launch_bounds__(T) big_k_pcg_upd(BigDev d, int in, int out) {     // :300-317
    __shared__ double red[2 * RED_MAXV * RED_MAXW];
    int parity = 0;
    const BigState *si = d.st + in;
    if (si->halt || si->phase != 2 || si->pcg_done) { forward_state(d, in, out); return; }
    const int k = si->pcg_k;
    double c1v[1];
    get_red<1, 1>(d, BIG_PH_C, c1v, red, parity);
    const double alpha = si->absNew / c1v[0];
    // (alpha < 0 -> the plain loop ignores the PCG's -1 return, LPcpp:894; x keeps the updates made so far, which is what
    //  happens here too because the remaining pairs fall through once pcg_done is set)
    const bool fail = alpha < 0;
    const double *p = (k & 1) ? d.p1 : d.p0;
    double pd2[2] = {0.0, 0.0};
    if (!fail)
        for (int q = 0; q < d.EPT; q++) {
            const int j = blockIdx.x * (T * d.EPT) + q * T + threadIdx.x;
            double a = 0.0, b2 = 0.0;
            if (j < d.n_loc) {
                double x = d.xt[j], r = d.r[j];
                x += alpha * p[j];
                r -= alpha * d.tmp[j];
                const double z = d.dinv[j] * r;
                const bool lv = d.live[j];
                d.xt[j] = x; d.r[j] = r; d.z[j] = lv ? z : 0.0;
                d.zp[j] = make_double2(lv ? z : 0.0, p[j]);          // what rows() gathers for the next search direction z + beta p
                if (lv) { a = r * r; b2 = r * z; }
            }
            pd2[0] = pd2[0] + a; pd2[1] = pd2[1] + b2;
        }
    if (!fail) store_partials<2>(d, BIG_PH_D, pd2, red, parity);
    if (LEADER) {
        d.st[out] = *si;
        if (fail) { d.st[out].pcg_done = 1; d.st[out].stop = LP_STOP_PCG; }
        else d.st[out].pcg_k = k + 1;
    }
}

// COMM-LEAN PCG (opt-in, not the reference's arithmetic): pcg_cols and pcg_upd in one launch.  p.Mp is not summed over the variables but
// taken from p.p and q.q -- both complete once the q exchange is: alpha = absNew / (dI (p.p) + r4Et (q.q)) -- so no exchange separates the
// column product from the vector updates.  Totals: per rank the tree over its workgroup partials, the rank totals in rank order.
constexpr int FOLD_UQ = 8;     // q.q partials per thread (one rank: up to 2048 row workgroups)
__device__ __forceinline__ double lean_total(const BigDev &d, bool qq, double *red, int &parity) {
    const int W = d.W > 1 ? d.W : 1;
    double out = 0.0;
    for (int r = 0; r < W; r++) {
        const double *base = (d.gathered ? d.gsmall + (size_t)r * (d.Gs + d.Gqs) : d.lsmall) + (qq ? d.Gs : 0);
        const int G = qq ? (d.gathered ? d.Gqr[r] : d.Gq) : (d.gathered ? d.Gr[r] : d.G);
        double t[FOLD_UQ];
#pragma unroll
        for (int u = 0; u < FOLD_UQ; u++) { const int e = threadIdx.x + u * T; t[u] = e < G ? base[e] : 0.0; }
        double a[1] = {0.0};
#pragma unroll
        for (int u = 0; u < FOLD_UQ; u++) a[0] = (int)threadIdx.x + u * T < G ? a[0] + t[u] : a[0];
        block_sum<T, 1>(a, red, parity);
        out = r == 0 ? a[0] : out + a[0];
    }
    return out;
}

__global__ void __launch_bounds__(T) big_k_pcg_lean(BigDev d, int in, int out) {
    __shared__ double red[2 * RED_MAXV * RED_MAXW];
    int parity = 0;
    const BigState *si = d.st + in;
    if (si->halt || si->phase != 2) { forward_state(d, in, out); return; }
    if (si->pcg_done) {
        if (si->pcg_first)                                                       // rhs == 0: x := 0 (:273-278)
            for (int q = 0; q < d.EPT; q++) { const int j = blockIdx.x * (T * d.EPT) + q * T + threadIdx.x; if (j < d.n_loc) d.xt[j] = 0.0; }
        if (LEADER) { d.st[out] = *si; d.st[out].pcg_first = 0; }
        return;
    }
    const int k = si->pcg_k;
    const double dI = si->dI, r4Et = si->r4Et;
    const double pp = lean_total(d, false, red, parity), qq = lean_total(d, true, red, parity);
    const double pMp = dI * pp + r4Et * qq;
    const double alpha = si->absNew / pMp;
    const bool fail = alpha < 0;
    const double *p = (k & 1) ? d.p1 : d.p0;
    double pd2[2] = {0.0, 0.0};
    if (!fail)
        for (int q = 0; q < d.EPT; q++) {
            const int j = blockIdx.x * (T * d.EPT) + q * T + threadIdx.x;
            double a = 0.0, b2 = 0.0;
            if (j < d.n_loc) {
                const double pj = p[j];
                double t = 0.0;
                const int k1 = d.cptr[j + 1];
                int kk = d.cptr[j];
                for (; kk + 4 <= k1; kk += 4) {
                    const int r0 = d.crow[kk], r1 = d.crow[kk + 1], r2 = d.crow[kk + 2], r3 = d.crow[kk + 3];
                    const double v0 = d.q[r0], v1 = d.q[r1], v2 = d.q[r2], v3 = d.q[r3];
                    t += r4Et * v0; t += r4Et * v1; t += r4Et * v2; t += r4Et * v3;
                }
                for (; kk < k1; kk++) t += r4Et * d.q[d.crow[kk]];
                double Mp = 0.0;
                Mp += dI * (1.0 * pj);
                Mp += t;                                                         // tmp = M p (:298)
                double x = d.xt[j], r = d.r[j];
                x += alpha * pj;                                                 // :302
                r -= alpha * Mp;                                                 // :304
                const double z = d.dinv[j] * r;                                  // :314
                const bool lv = d.live[j];
                d.xt[j] = x; d.r[j] = r; d.z[j] = lv ? z : 0.0;
                d.zp[j] = make_double2(lv ? z : 0.0, pj);
                if (lv) { a = r * r; b2 = r * z; }
            }
            pd2[0] = pd2[0] + a; pd2[1] = pd2[1] + b2;
        }
    if (!fail) store_partials<2>(d, BIG_PH_D, pd2, red, parity);
    if (LEADER) {
        d.st[out] = *si;
        if (fail) { d.st[out].pcg_done = 1; d.st[out].stop = LP_STOP_PCG; }
        else d.st[out].pcg_k = k + 1;
    }
}

// rank-ordered sum of this rank's block of E*v (as big_k_rank_sum) and, for the comm-lean PCG, the workgroup partials of q.q over the block:
// one row per thread, 256 rows per workgroup
__global__ void __launch_bounds__(T) big_k_rank_sum_qq(const double *g, int W, long count, long stride, double *out, double *qq_part) {
    __shared__ double red[2 * RED_MAXV * RED_MAXW];
    int parity = 0;
    const long i = (long)blockIdx.x * T + threadIdx.x;
    double c[1] = {0.0};
    if (i < count) {
        double acc = g[i];
        for (int r = 1; r < W; r++) acc = acc + g[(long)r * stride + i];
        out[i] = acc;
        c[0] = c[0] + acc * acc;
    }
    block_sum<T, 1>(c, red, parity);
    if (threadIdx.x == 0) qq_part[blockIdx.x] = c[0];
}

// after the PCG: duals z1, z2 (:917-918), the five partials (:931-1003), gsrc = x for the E*x that feeds z4 and the next y3
__global__ void __launch_bounds__(T) big_k_post(BigDev d, int in, int out) {
    __shared__ double red[2 * RED_MAXV * RED_MAXW];
    int parity = 0;
    const BigState *si = d.st + in;
    if (si->halt || si->phase != 2) { forward_state(d, in, out); return; }
    const int k = si->pcg_k;
    if (!si->pcg_done) {                      // the exit test of the last update is still pending
        double d1[1] = {0.0};
        if (k >= 1) get_red<1, 2>(d, BIG_PH_D, d1, red, parity);
        if (!(k >= 1 && (d1[0] < si->threshold || k >= LP_PCG_MAXITERS))) {
            if (LEADER) { d.st[out] = *si; d.st[out].halt = BIG_HALT_PCG_MORE; }
            return;
        }
    }
    if (si->l2f && si->stop == LP_STOP_PCG) {   // alpha < 0 inside the l2f loop: return 1, x_sol untouched (:1450-1454)
        if (LEADER) { d.st[out] = *si; BigState *s = d.st + out; s->ret = 1; s->halt = BIG_HALT_STOP; s->pcg_done = 1; s->last_pcg = k; s->pcg_total += k; }
        return;
    }
    const double g1 = si->gamma_val * si->rho1, g2 = si->gamma_val * si->rho2;
    double *xh = ((si->l2f || si->rec) && d.xhist && si->cc < d.ws_cap) ? d.xhist + (size_t)si->cc * d.n_loc : nullptr;
    double e5[5] = {0.0, 0.0, 0.0, 0.0, 0.0};
    for (int q = 0; q < d.EPT; q++) {
        const int j = blockIdx.x * (T * d.EPT) + q * T + threadIdx.x;
        double v0 = 0.0, v1 = 0.0, v2 = 0.0, v3 = 0.0, v4 = 0.0;
        if (j < d.n_loc) {
            const bool lv = d.live[j];
            const double x = lv ? d.xt[j] : d.x[j], y1 = d.y1[j], y2 = d.y2[j], b = d.b[j];   // commit: fixed variables keep their value
            d.x[j] = x;
            if (xh) xh[j] = x;                                                // x_iters column (:1472-1475)
            d.z1[j] = d.z1[j] + g1 * (x - y1);
            d.z2[j] = d.z2[j] + g2 * (x - y2);
            d.gsrc[j] = lv ? x : 0.0;
            const double d1 = x - y1, d2 = x - y2, xb = x >= 0.5 ? 1.0 : 0.0;
            if (lv) { v0 = x * x; v1 = d1 * d1; v2 = d2 * d2; v3 = b * x; v4 = b * xb; }
        }
        e5[0] = e5[0] + v0; e5[1] = e5[1] + v1; e5[2] = e5[2] + v2; e5[3] = e5[3] + v3; e5[4] = e5[4] + v4;
    }
    store_partials<5>(d, BIG_PH_E, e5, red, parity);
    if (LEADER) {
        d.st[out] = *si;
        BigState *s = d.st + out;
        s->pcg_done = 1; s->last_pcg = k; s->pcg_total += k; s->outer_total++;
        if (si->l2f || si->rec) s->cc = si->cc + 1;
        if (k > s->pcg_max) s->pcg_max = k;
        s->phase = 3;
    }
}

// Ex = q (= E*x, already all-reduced), z4 dual update (:919-924; the plain loop OVERWRITES z4 on the first iteration of a call)
__global__ void __launch_bounds__(T) big_k_z4(BigDev d, int in, int out, int init_only) {
    const BigState *si = d.st + in;
    if (init_only) {
        for (int s = 0; s < d.EPTl; s++) { const int i = blockIdx.x * (T * d.EPTl) + s * T + threadIdx.x; if (i < d.l) d.Ex[i] = d.q[i]; }
        forward_state(d, in, out);
        return;
    }
    if (si->halt || si->phase != 3) { forward_state(d, in, out); return; }
    const double g4 = si->gamma_val * si->rho4;
    const bool overwrite = !si->l2f && si->iter == si->iter_start;
    for (int s = 0; s < d.EPTl; s++) {
        const int i = blockIdx.x * (T * d.EPTl) + s * T + threadIdx.x;
        if (i >= d.l) continue;
        const double Ex = d.q[i];
        d.Ex[i] = Ex;
        const double dd = g4 * ((Ex + d.y3[i]) - d.f[i]);
        d.z4[i] = overwrite ? dd : d.z4[i] + dd;
    }
    if (LEADER) { d.st[out] = *si; d.st[out].have_prev = 1; d.st[out].phase = 0; }
}

// ---- early fixing (LPcpp:1124-1335) as a mask; three passes cut at the two reductions it needs ----
__global__ void __launch_bounds__(T) big_k_fix1(BigDev d) {          // x2 = the newly fixed values; partial fix_obj = b2.x2 (:1237)
    __shared__ double red[2 * RED_MAXV * RED_MAXW];
    int parity = 0;
    double pf[1] = {0.0};
    for (int q = 0; q < d.EPT; q++) {
        const int j = blockIdx.x * (T * d.EPT) + q * T + threadIdx.x;
        double c = 0.0;
        if (j < d.n_loc) {
            const int nf = d.newfix[j];
            const double val = nf == 2 ? 1.0 : 0.0;
            d.gsrc[j] = nf ? val : 0.0;
            if (nf) c = d.b[j] * val;
        }
        pf[0] = pf[0] + c;
    }
    store_partials<1>(d, BIG_PH_X, pf, red, parity);
}

__global__ void __launch_bounds__(T) big_k_fix2(BigDev d, int in, int out) {   // red[0] = fix_obj, q = E2*x2 (all ranks)
    __shared__ double red[2 * RED_MAXV * RED_MAXW];
    int parity = 0;
    if (blockIdx.x < d.Gl)
        for (int s = 0; s < d.EPTl; s++) {
            const int i = blockIdx.x * (T * d.EPTl) + s * T + threadIdx.x;
            if (i < d.l) d.f[i] = d.f[i] - d.q[i];                                  // f1 = f - E2*x2 (:1278)
        }
    if (LEADER) { d.st[out] = d.st[in]; d.st[out].fix_obj = d.red[BIG_PH_X * BIG_NPART]; }
    if (blockIdx.x >= d.G) return;
    double px[1] = {0.0};
    for (int q = 0; q < d.EPT; q++) {
        const int j = blockIdx.x * (T * d.EPT) + q * T + threadIdx.x;
        double c = 0.0;
        if (j < d.n_loc) {
            const int nf = d.newfix[j];
            if (nf) { d.live[j] = 0; d.x[j] = nf == 2 ? 1.0 : 0.0; }
            else if (d.live[j]) { const double x = d.x[j]; c = x * x; }
        }
        px[0] = px[0] + c;
    }
    store_partials<1>(d, BIG_PH_X, px, red, parity);
}

// red[0] = |x_live|^2.  n_live_new == 0: everything is fixed (:1212-1217, nothing else is updated).
__global__ void __launch_bounds__(T) big_k_fix3(BigDev d, int in, int out, long n_live_new, double c1_new) {
    const BigState *si = d.st + in;
    const double rho1 = si->rho1, rho2 = si->rho2, rho4 = si->rho4;
    if (LEADER) {
        d.st[out] = *si;
        BigState *s = d.st + out;
        s->n_live_lo = (int)(n_live_new & 0x7fffffff); s->n_live_hi = (int)(n_live_new >> 31);
        if (n_live_new == 0) { s->ret = 1; s->stop = LP_STOP_ALLFIXED; s->halt = BIG_HALT_STOP; }
        else {
            if (sqrt(d.red[BIG_PH_X * BIG_NPART]) < 1e-3) s->ret = 1;                                 // :1223
            s->prev_sum = s->sum_fix_obj; s->sum_fix_obj += s->fix_obj; s->prev_obj = s->cur_obj;   // :1247-1250
            s->c1 = c1_new;
            double dI = 0.0; dI += rho1 + rho2;                                      // update_expression (:1329 -> :2289-2404)
            s->dI = dI; s->r4Et = rho4; s->expr_ready = 1;
        }
    }
    if (n_live_new == 0) return;
    double dI = 0.0; dI += rho1 + rho2;
    for (int q = 0; q < d.EPT; q++) {
        const int j = blockIdx.x * (T * d.EPT) + q * T + threadIdx.x;
        if (j >= d.n_loc) continue;
        double pd = dI;
        pd += rho4 * (double)(d.cptr[j + 1] - d.cptr[j]);
        d.pd[j] = pd;
        d.dinv[j] = (pd != 0.0) ? 1.0 / pd : 1.0;        // the preconditioner always mirrors pd (a stale one is UB in the reference)
        d.gsrc[j] = d.live[j] ? d.x[j] : 0.0;            // for E*x of the first iteration's y3
    }
}

// out[r*ws + c] = x after iteration c of the r-th LOCAL live variable (get_x_iters_d, LPcpp:1616-1627)
// out[i] = g[0][i] + g[1][i] + ... + g[W-1][i], added in RANK ORDER: the cross-rank association of every sum over variables of the
// variable-sharded run (each g[r] is the contribution of rank r; every rank runs this on the same gathered data and gets the same bits).
__global__ void big_k_rank_sum(const double *g, int W, long count, long stride, double *out) {
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < count; i += (long)gridDim.x * blockDim.x) {
        double acc = g[i];
        for (int r = 1; r < W; r++) acc = acc + g[(long)r * stride + i];
        out[i] = acc;
    }
}

__global__ void big_k_pack_xiters(BigDev d, const int *live_idx, int rows, int ws, double *out) {
    const long total = (long)rows * ws;
    for (long e = (long)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (long)gridDim.x * blockDim.x) {
        const int r = (int)(e / ws), c = (int)(e % ws);
        out[e] = c < d.ws_cap ? d.xhist[(size_t)c * d.n_loc + live_idx[r]] : 0.0;
    }
}

}  // namespace

// ---------------------------------------------------------------------------------------------------------------------
#define BIG_LAUNCH(kernel, grid, ...)                                                                     \
    do {                                                                                                  \
        hipLaunchKernelGGL(kernel, dim3(grid), dim3(T), 0, s, d, *parity, *parity ^ 1, ##__VA_ARGS__);    \
        *parity ^= 1;                                                                                     \
    } while (0)

hipError_t big_launch_init(const BigDev &d, double c1, hipStream_t s) {
    hipLaunchKernelGGL(big_k_init, dim3(d.G > d.Gl ? d.G : d.Gl), dim3(T), 0, s, d, c1);
    hipLaunchKernelGGL(big_k_fin, dim3(1), dim3(T), 0, s, d, 1, BIG_PH_X);
    return hipGetLastError();
}
hipError_t big_launch_init2(const BigDev &d, hipStream_t s) {
    hipLaunchKernelGGL(big_k_init2, dim3(1), dim3(1), 0, s, d);
    return hipGetLastError();
}
hipError_t big_launch_fix1(const BigDev &d, hipStream_t s) {
    hipLaunchKernelGGL(big_k_fix1, dim3(d.G), dim3(T), 0, s, d);
    return hipGetLastError();
}
hipError_t big_launch_fix2(const BigDev &d, int *parity, hipStream_t s) { BIG_LAUNCH(big_k_fix2, (d.G > d.Gl ? d.G : d.Gl)); return hipGetLastError(); }
hipError_t big_launch_fix3(const BigDev &d, long n_live_new, double c1_new, int *parity, hipStream_t s) { BIG_LAUNCH(big_k_fix3, d.G, n_live_new, c1_new); return hipGetLastError(); }
hipError_t big_launch_pack_xiters(const BigDev &d, const int *live_idx, int rows, int ws, double *out, hipStream_t s) {
    if (rows <= 0 || ws <= 0) return hipSuccess;
    const long total = (long)rows * ws;
    const int grid = (int)std::min<long>((total + 255) / 256, 65535);
    hipLaunchKernelGGL(big_k_pack_xiters, dim3(grid), dim3(256), 0, s, d, live_idx, rows, ws, out);
    return hipGetLastError();
}
hipError_t big_launch_rank_sum(const double *g, int W, long count, long stride, double *out, hipStream_t s) {
    if (count <= 0) return hipSuccess;
    const int grid = (int)std::min<long>((count + 255) / 256, 4096);
    hipLaunchKernelGGL(big_k_rank_sum, dim3(grid), dim3(256), 0, s, g, W, count, stride, out);
    return hipGetLastError();
}
hipError_t big_launch_set_window(const BigDev &d, int iter_start, int iter_end, int l2f, int *parity, hipStream_t s) {
    hipLaunchKernelGGL(big_k_set_window, dim3(1), dim3(1), 0, s, d, *parity, *parity ^ 1, iter_start, iter_end, l2f);
    *parity ^= 1;
    return hipGetLastError();
}
hipError_t big_launch_resume(const BigDev &d, int reset_pcg_max, int *parity, hipStream_t s) {
    hipLaunchKernelGGL(big_k_resume, dim3(1), dim3(1), 0, s, d, *parity, *parity ^ 1, reset_pcg_max);
    *parity ^= 1;
    return hipGetLastError();
}
hipError_t big_launch_prep(const BigDev &d, int do_prep, int *parity, hipStream_t s) { BIG_LAUNCH(big_k_prep, d.G, do_prep); return hipGetLastError(); }
hipError_t big_launch_fin(const BigDev &d, int nv, int phase, hipStream_t s) {
    hipLaunchKernelGGL(big_k_fin, dim3(1), dim3(T), 0, s, d, nv, phase);
    return hipGetLastError();
}
hipError_t big_launch_y(const BigDev &d, int *parity, hipStream_t s) { BIG_LAUNCH(big_k_y, (d.G > d.Gl ? d.G : d.Gl)); return hipGetLastError(); }
hipError_t big_launch_rhs_cols(const BigDev &d, int *parity, hipStream_t s) { BIG_LAUNCH(big_k_rhs_cols, d.G); return hipGetLastError(); }
hipError_t big_launch_rows(const BigDev &d, int mode, int *parity, hipStream_t s) {
    const int rowgrid = d.P > 1 ? d.Glr : d.Gl;
    BIG_LAUNCH(big_k_rows, (d.lean && mode == 1 && d.G > rowgrid ? d.G : rowgrid), mode);      // comm-lean: the column workgroups' p.p partials ride along
    return hipGetLastError();
}
hipError_t big_launch_pcg_lean(const BigDev &d, int *parity, hipStream_t s) { BIG_LAUNCH(big_k_pcg_lean, d.G); return hipGetLastError(); }
hipError_t big_launch_rank_sum_qq(const double *g, int W, long count, long stride, double *out, double *qq_part, hipStream_t s) {
    if (count <= 0) return hipSuccess;
    hipLaunchKernelGGL(big_k_rank_sum_qq, dim3((unsigned)((count + T - 1) / T)), dim3(T), 0, s, g, W, count, stride, out, qq_part);
    return hipGetLastError();
}
hipError_t big_launch_resid(const BigDev &d, int *parity, hipStream_t s) { BIG_LAUNCH(big_k_resid, d.G); return hipGetLastError(); }
hipError_t big_launch_pcg_cols(const BigDev &d, int *parity, hipStream_t s) { BIG_LAUNCH(big_k_pcg_cols, d.G); return hipGetLastError(); }
hipError_t big_launch_pcg_upd(const BigDev &d, int *parity, hipStream_t s) { BIG_LAUNCH(big_k_pcg_upd, d.G); return hipGetLastError(); }
hipError_t big_launch_post(const BigDev &d, int *parity, hipStream_t s) { BIG_LAUNCH(big_k_post, d.G); return hipGetLastError(); }
hipError_t big_launch_z4(const BigDev &d, int init_only, int *parity, hipStream_t s) { BIG_LAUNCH(big_k_z4, d.Gl, init_only); return hipGetLastError(); }
