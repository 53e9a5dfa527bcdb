// lpbox_seg_kernels.hip -- gfx950 kernels of the SEGMENTATION flavour (unconstrained BQP, SEGcpp:658-1380).
//
// n ~ 1e4..1e6, so the vectors live in HBM / L2 and one ADMM iteration is a chain of kernels cut at the algorithm's
// grid-wide dependencies:
//   prep    : (head / tail of a batch of iterations, and every iteration of the multi-problem chain) finalise the previous iteration
//             (stop tests, rho schedule, objective history), partial ||x+z2/rho2-1/2||^2; inside a single-problem batch `post` and
//             `yrhs` do this work and the launch is dropped
//   yrhs    : y1, y2, diagonal/preconditioner refresh, rhs, PCG start x0 = y1
//   resid   : r = rhs - (2A+(rho1+rho2)I) x0,  p = r/diag, partials rhs.rhs, r.r, r.p
//   matvec  : [beta, p = z + beta p]  tmp = M p, partial p.tmp          \  one PCG iteration = 2 launches: the p update of
//   update  : alpha, x += alpha p, r -= alpha tmp, z = r/diag, partials /  iteration k is recomputed on the fly for the
//                                                                          gathered neighbours inside matvec k
//   post    : duals z1,z2, A x and A round(x) (objective), the five norms / dot partials, x_iters column
// Each kernel reads the control state st[in], every workgroup re-derives the same scalar decisions from the same partial
// sums (two-level fixed tree, bit-reproducible), and thread 0 of workgroup 0 writes st[out]; a halted / finished state makes
// every later kernel of the replayed graph fall through.  Sparse rows are summed by one lane in ascending column order
// (Eigen's row-major product, SEGh:17), no FMA contraction, IEEE divide/sqrt -- as in the LP kernels.
#include "lpbox_seg.h"
#include "lpbox_dev_common.h"

#include <float.h>

namespace {

constexpr int T = SEG_T;
enum { PH_A = 0, PH_B = 1, PH_C = 2, PH_D = 3, PH_E = 4, PH_COUNT = 5 };

__device__ __forceinline__ double *part_ptr(const SegDev &d, int phase, int v) {
    return d.part + ((size_t)(phase * SEG_NPART + v)) * (size_t)d.G;
}

// every workgroup: total of the G workgroup partials of NV values (second level of the fixed tree)
// This sits at the head of every consumer kernel, i.e. on the critical path of the launch chain: the host keeps G <= 2 T, so a thread
// has at most two partials per value, and all of them (NV x 2 loads) are requested before the first addition -- one memory latency
// instead of up to 2 NV dependent ones.  Same per-thread order (partial t, then t + T), same tree.
template <int NV>
__device__ __forceinline__ void final_sums(const SegDev &d, int phase, double (&out)[NV], double *red, int &parity) {
    const int e0 = threadIdx.x, e1 = threadIdx.x + T;
    double t0[NV], t1[NV];
#pragma unroll
    for (int k = 0; k < NV; k++) {
        const double *p = part_ptr(d, phase, k);
        t0[k] = e0 < d.G ? p[e0] : 0.0;
        t1[k] = e1 < d.G ? p[e1] : 0.0;
    }
#pragma unroll
    for (int k = 0; k < NV; k++) {
        double a = 0.0;
        if (e0 < d.G) a = a + t0[k];
        if (e1 < d.G) a = a + t1[k];
        out[k] = a;
    }
    block_sum<T, NV>(out, red, parity);
}

template <int NV>
__device__ __forceinline__ void store_partials(const SegDev &d, int phase, double (&v)[NV], double *red, int &parity) {
    block_sum<T, NV>(v, red, parity);
    if (threadIdx.x == 0) {
#pragma unroll
        for (int k = 0; k < NV; k++) part_ptr(d, phase, k)[blockIdx.x] = v[k];
    }
}

#define LEADER (blockIdx.x == 0 && threadIdx.x == 0)

// The control state is read field by field (uniform scalar loads); only thread 0 of workgroup 0 copies it forward and
// edits it.  (A per-thread copy of the whole struct would live in scratch memory and cost ~15 us of dispatch per launch.)
__device__ __forceinline__ void forward_state(const SegDev &d, int in, int out) {
    if (LEADER) d.st[out] = d.st[in];
}

// The same row product in three separable steps, for the kernels that handle the two rows of a thread TOGETHER (image problems:
// ell_w <= 8, EPT == 2): these launches are a few microseconds long and bound by the chain of dependent memory round trips of one
// thread, not by bandwidth -- a row at a time costs row length -> indices/values -> gathered elements -> result store, and the second
// row's loads queue behind the first row's stores (one counter for loads and stores on this target).  ell_load can be issued before
// the kernel's reduction of the previous launch's partials, ell_gather for both rows at once; ell_sum keeps tm_row's expression and
// order.  Padded slots hold column i and value 0 (lpbox_seg_capi.hip), so all ell_w slots are read and the length only masks the sum.
struct EllRow { int len; double tdi; int c[8]; double v[8]; };
__device__ __forceinline__ void ell_load(const SegDev &d, int i, EllRow &R) {
    R.tdi = d.td[i];
    if (d.dia) {                                                             // diagonal storage (uniform branch): see SegDev
        const unsigned long long pk = d.dpack[i];
        const double aii = d.adiag[i];
        const int other = i ? i - 1 : 1;                                     // an in-range column != i for the slots that do not exist
        R.len = 7;
#pragma unroll
        for (int k = 0; k < 7; k++) {
            if (k == 3) { R.c[k] = i; R.v[k] = aii; continue; }
            const int s = k < 3 ? k : k - 1;
            const int ci = i + d.doff[k];
            R.c[k] = (unsigned)ci < (unsigned)d.n ? ci : other;
            R.v[k] = -(double)(unsigned)((pk >> (8 * s)) & 255ull);
        }
        R.c[7] = i; R.v[7] = 0.0;
        return;
    }
    R.len = d.rowlen[i];
#pragma unroll
    for (int k = 0; k < 8; k++) {
        const bool in = k < d.ell_w;
        R.c[k] = in ? d.ecol[(size_t)k * d.n + i] : i;
        R.v[k] = in ? d.eval[(size_t)k * d.n + i] : 0.0;
    }
}
__device__ __forceinline__ double ell_sum(const EllRow &R, int i, const double (&g)[8]) {
    double tmp = 0;
#pragma unroll
    for (int k = 0; k < 8; k++) if (k < R.len) tmp += ((R.c[k] == i) ? R.tdi : 2 * R.v[k]) * g[k];
    double res = 0.0;
    res += 1.0 * tmp;
    return res;
}

// (2A + (rho1+rho2) I) row i times a gathered vector: diagonal entry = td[i], off-diagonal = 2*A_ik (SEGcpp:784-786).
// The matrix is held in ELL form (slot k of row i at [k*n + i], ascending columns): consecutive lanes = consecutive rows
// read consecutive addresses, and the 7-diagonal structure makes the gathers of v coalesced as well.
template <typename GET>
__device__ __forceinline__ double tm_row(const SegDev &d, int i, GET get) {
    double tmp = 0;
    if (d.ell_w <= 8) {
        // rows of the image problems have <= 7 entries: fetch all indices, then all values, then all gathered elements before the
        // (ordered) additions, so that the loads of one row are in flight together
        EllRow R;
        ell_load(d, i, R);
        double g[8];
#pragma unroll
        for (int k = 0; k < 8; k++) g[k] = get(R.c[k]);
        return ell_sum(R, i, g);
    } else {
        const int len = d.rowlen[i];
        const double tdi = d.td[i];
        for (int k = 0; k < len; k++) {
            const int c = d.ecol[(size_t)k * d.n + i];
            const double val = (c == i) ? tdi : 2 * d.eval[(size_t)k * d.n + i];
            tmp += val * get(c);
        }
    }
    double res = 0.0;
    res += 1.0 * tmp;
    return res;
}

// ---------------------------------------------------------------------------------------------------------------------
__device__ __forceinline__ void seg_b_init(const SegDev &d, double c1) {      // ADMM_bqp_unconstrained_init SEGcpp:658-810
    for (int s = 0; s < d.EPT; s++) {
        const int i = blockIdx.x * (T * d.EPT) + s * T + threadIdx.x;
        if (i >= d.n) continue;
        d.x[i] = 0.0; d.y1[i] = 0.0; d.y2[i] = 0.0; d.z1[i] = 0.0; d.z2[i] = 0.0;   // :762-777
        d.live[i] = 1; d.fixval[i] = 0;
        double aii = 0.0;
        if (d.dia) aii = d.adiag[i];
        else for (int k = 0; k < d.rowlen[i]; k++) if (d.ecol[(size_t)k * d.n + i] == i) aii = d.eval[(size_t)k * d.n + i];
        double t = 2 * aii;
        t += SEG_RHO0 + SEG_RHO0;                                           // temp_mat.diagonal() += rho1 + rho2 (:785)
        d.td[i] = t;
        d.dinv[i] = 1.0; d.r[i] = 0.0; d.z[i] = 0.0; d.tmp[i] = 0.0; d.p0[i] = 0.0; d.p1[i] = 0.0; d.rhs[i] = 0.0;
    }
    if (LEADER) {
        SegState *s = d.st;
        memset(s, 0, sizeof(SegState));
        s->rho1 = s->rho2 = s->prev_rho1 = s->prev_rho2 = SEG_RHO0;
        s->gamma_val = SEG_GAMMA0; s->std_obj = 1.0; s->rhoUpdated = 1;
        s->best_bin_obj = 0.0 + 0.0;                                        // compute_cost(x = 0) (:792)
        s->n_live = d.n; s->c1 = c1;
        d.st[1] = d.st[0];
    }
}

__device__ __forceinline__ void seg_b_set_window(const SegDev &d, int in, int out, int iter_start, int iter_end, int mode) {
    const int l2f = mode & 1;
    d.st[out] = d.st[in];
    SegState *s = d.st + out;
    s->iter = iter_start; s->iter_end = iter_end; s->l2f = l2f; s->rec = (mode >> 1) & 1; s->cc = 0; s->ret = 0; s->stop = SEG_STOP_NONE;
    if (s->halt != SEG_HALT_ALLFIXED) s->halt = SEG_HALT_NONE;
}

__device__ __forceinline__ void seg_b_resume(const SegDev &d, int in, int out, int reset_pcg_max) {
    d.st[out] = d.st[in];
    if (d.st[out].halt == SEG_HALT_PCG_MORE) d.st[out].halt = SEG_HALT_NONE;
    if (reset_pcg_max) d.st[out].pcg_max = 0;
}

// early fixing as a mask (SEGcpp:927-1062): fixed variables leave the problem, b := 2*Mb*x2 + b1, temp_mat rebuilt
__global__ void __launch_bounds__(T) seg_k_fix(SegDev d, int in, int out, int n_live_new, double c1_new) {
    const double rho12 = d.st[in].rho1 + d.st[in].rho2;
    for (int q = 0; q < d.EPT; q++) {
        const int i = blockIdx.x * (T * d.EPT) + q * T + threadIdx.x;
        if (i >= d.n) continue;
        const int nf = d.newfix[i];
        if (nf) { d.live[i] = 0; d.fixval[i] = nf == 2 ? 1 : 0; d.x[i] = 0.0; d.p0[i] = 0.0; d.p1[i] = 0.0; d.z[i] = 0.0; d.r[i] = 0.0; continue; }
        if (!d.live[i]) continue;
        if (n_live_new == 0) continue;
        double tmp = 0, aii = 0.0;
        if (d.dia) {
            EllRow R;
            ell_load(d, i, R);
            aii = R.v[3];
#pragma unroll
            for (int k = 0; k < 7; k++) {
                const int nfc = d.newfix[R.c[k]];
                if (nfc) tmp += R.v[k] * (nfc == 2 ? 1.0 : 0.0);             // a slot without an entry adds -0.0: no change
            }
        } else
        for (int k = 0; k < d.rowlen[i]; k++) {
            const int c = d.ecol[(size_t)k * d.n + i];
            const double a = d.eval[(size_t)k * d.n + i];
            if (c == i) aii = a;
            const int nfc = d.newfix[c];
            if (nfc) tmp += a * (nfc == 2 ? 1.0 : 0.0);                      // Mb * x2, ascending column (:1051)
        }
        double res = 0.0;
        res += 1.0 * tmp;
        d.b[i] = 2 * res + d.b[i];                                          // :1052
        double t = 2 * aii;
        t += rho12;                                                         // :1054-1057
        d.td[i] = t;
    }
    if (LEADER) {
        d.st[out] = d.st[in];
        SegState *s = d.st + out;
        if (n_live_new == 0) { s->ret = 1; s->stop = SEG_STOP_ALLFIXED; s->halt = SEG_HALT_ALLFIXED; s->n_live = 0; }   // :1028-1032
        else { s->n_live = n_live_new; s->c1 = c1_new; s->dinv_stale = 1; }
    }
}

// ---- finalisation of an iteration (SEGcpp:1122-1180 / :1277-1376): the totals of what `post` left as workgroup partials ...
__device__ __forceinline__ void finalise_sums(const SegDev &d, double (&e)[5], double (&e2)[2], double *red, int &parity) {
    final_sums<5>(d, PH_E, e, red, parity);                                  // x.x, |x-y1|^2, |x-y2|^2, x.Ax, b.x
    const double *p5 = part_ptr(d, PH_E, 5), *p6 = part_ptr(d, PH_E, 6);
    double a = 0.0, b2 = 0.0;
    for (int q = threadIdx.x; q < d.G; q += T) { a = a + p5[q]; b2 = b2 + p6[q]; }
    e2[0] = a; e2[1] = b2;
    block_sum<T, 2>(e2, red, parity);                                        // xb.A xb, b.xb
}
// ... and the scalar decisions on them, applied to a copy *s of the state (one thread): stop tests, rho schedule, objective history
__device__ __forceinline__ void finalise_state(SegState *s, const double (&e)[5], const double (&e2)[2]) {
    const int it = s->iter;
    s->have_prev = 0;
    const double xn = sqrt(e[0]);
    const double t0 = xn < 2.2204e-16 ? 2.2204e-16 : xn;
    s->cvg1 = sqrt(e[1]) / t0; s->cvg2 = sqrt(e[2]) / t0;
    const double bin_cost = e2[0] + e2[1];                       // compute_cost(round(x)) (:1167 / :1372)
    bool stopped = false;
    if (s->cvg1 <= SEG_STOP_THRESHOLD && s->cvg2 <= SEG_STOP_THRESHOLD) {   // :1127 / :1282
        if (s->l2f) s->ret = 1;
        s->stop = SEG_STOP_XYY; stopped = true;
    } else {
        if ((it + 1) % SEG_RHO_STEP == 0) {                      // :1137-1145
            s->prev_rho1 = s->rho1; s->prev_rho2 = s->rho2;
            s->rho1 = SEG_LEARNING_FACT * s->rho1; s->rho2 = SEG_LEARNING_FACT * s->rho2;
            const double g = s->gamma_val * SEG_GAMMA_FACTOR;
            s->gamma_val = g < 1.0 ? 1.0 : g;
            s->rhoUpdated = 1; s->rcr = SEG_LEARNING_FACT - 1.0;
        }
        s->obj_val = e[3] + e[4];                                // compute_cost(x) (:1148)
        int hn = s->hist_n;
        if (hn < SEG_HIST) s->hist[hn] = s->obj_val;
        else { for (int k = 0; k < SEG_HIST - 1; k++) s->hist[k] = s->hist[k + 1]; s->hist[SEG_HIST - 1] = s->obj_val; }
        if (hn < 0x3fffffff) hn++;
        s->hist_n = hn;
        if (hn >= SEG_HIST) {
            double mean = 0;
            for (int k = 0; k < SEG_HIST; k++) mean += s->hist[k];
            mean /= (double)SEG_HIST;
            double dev = 0;
            for (int k = 0; k < SEG_HIST; k++) dev += (s->hist[k] - mean) * (s->hist[k] - mean);
            dev /= (double)(SEG_HIST - 1);
            const double sd = dev == 0 ? 0.0 : sqrt(dev);
            s->std_obj = sd / fabs(s->hist[SEG_HIST - 1]);
        }
        if (s->std_obj <= SEG_STD_THRESHOLD) { if (s->l2f) s->ret = 1; s->stop = SEG_STOP_OBJSTD; stopped = true; }
        else {
            s->cur_obj = bin_cost;
            if (s->best_bin_obj >= s->cur_obj) s->best_bin_obj = s->cur_obj;
        }
    }
    if (stopped) {
        s->halt = SEG_HALT_STOP;
        if (!s->l2f) { s->cur_obj = bin_cost; s->legacy_iter_p1 = it + 1; }   // legacy epilogue (:1371-1376)
    } else s->iter = it + 1;
}
// the window ends here / another iteration follows
__device__ __forceinline__ void close_or_continue(SegState *s, int do_prep) {
    if (!s->halt && s->iter >= s->iter_end) { s->halt = SEG_HALT_WINDOW; if (!s->l2f) s->legacy_iter_p1 = s->iter + 1; }
    if (!s->halt && do_prep) s->phase = 1;
}

// prep: finalise the previous iteration if that is still pending (workgroup 0 only), then the sphere-norm partials of the next one.
// Launched at the head of a batch of iterations and (do_prep = 0) at its end; INSIDE a batch `post` leaves those partials and `yrhs`
// finalises (every workgroup for itself), so an iteration is 3 + 2 kmax launches.
__device__ __forceinline__ void seg_b_prep(const SegDev &d, int in, int out, int do_prep) {
    __shared__ double red[2 * RED_MAXV * RED_MAXW];
    int parity = 0;
    const SegState *si = d.st + in;
    const int halt0 = si->halt, have_prev = si->have_prev, it = si->iter, iter_end = si->iter_end;
    double rho2 = si->rho2;
    // what every workgroup can decide without the sums: the rho the next iteration will use, and whether a next one exists
    const bool fin = !halt0 && have_prev;
    if (fin && (it + 1) % SEG_RHO_STEP == 0) rho2 = SEG_LEARNING_FACT * rho2;
    const int next_iter = fin ? it + 1 : it;
    const bool will_prep = do_prep && !halt0 && next_iter < iter_end;       // (a stop detected below only wastes this launch's partials)
    if (blockIdx.x == 0) {
        double e[5] = {0, 0, 0, 0, 0}, e2[2] = {0, 0};
        if (fin) finalise_sums(d, e, e2, red, parity);                       // uniform over the workgroup
        if (threadIdx.x == 0) {
            d.st[out] = *si;
            SegState *s = d.st + out;
            if (fin) finalise_state(s, e, e2);
            close_or_continue(s, do_prep);
        }
    }
    if (!will_prep) return;
    double pa[1] = {0.0};
    for (int q = 0; q < d.EPT; q++) {
        const int i = blockIdx.x * (T * d.EPT) + q * T + threadIdx.x;
        double c = 0.0;
        if (i < d.n && d.live[i]) { const double u = (d.x[i] + d.z2[i] / rho2) - 0.5; c = u * u; }
        pa[0] = pa[0] + c;
    }
    store_partials<1>(d, PH_A, pa, red, parity);
}

template <bool PAIR>
__device__ __forceinline__ void seg_b_yrhs(const SegDev &d, int in, int out) {
    __shared__ double red[2 * RED_MAXV * RED_MAXW];
    int parity = 0;
    const SegState *si = d.st + in;
    if (si->halt) { forward_state(d, in, out); return; }
    // inside a batch the previous iteration is still to be finalised (its `post` left the partials): every workgroup does that for
    // itself -- same sums, same scalar code, same result -- on a copy of the state in LDS; workgroup 0 publishes it
    __shared__ SegState fst;
    if (si->have_prev) {                                                     // uniform over the grid
        double e[5] = {0, 0, 0, 0, 0}, e2[2] = {0, 0};
        finalise_sums(d, e, e2, red, parity);
        if (threadIdx.x == 0) { fst = *si; finalise_state(&fst, e, e2); close_or_continue(&fst, 1); }
        __syncthreads();
        si = &fst;
        if (si->halt) { if (LEADER) d.st[out] = fst; return; }
    }
    const double rho1 = si->rho1, rho2 = si->rho2, c1 = si->c1;
    const int rhoUpdated = si->rhoUpdated, stale = si->dinv_stale;
    const bool refresh = si->iter != 0 && rhoUpdated;
    const double inc = (si->prev_rho1 + si->prev_rho2) * si->rcr;            // :1086
    // two rows per thread together, their operands requested before the reduction (see ell_load)
    const bool pair = PAIR && d.EPT == 2;
    int ri[2] = {0, 0}; bool rok[2] = {false, false};
    double vx[2], vz1[2], vz2[2], vtd[2], vb[2];
    if (pair) {
#pragma unroll
        for (int q = 0; q < 2; q++) {
            const int i = blockIdx.x * (T * 2) + q * T + threadIdx.x;
            const bool inr = i < d.n;
            ri[q] = inr ? i : d.n - 1;
            rok[q] = inr && d.live[ri[q]];
            vx[q] = d.x[ri[q]]; vz1[q] = d.z1[ri[q]]; vz2[q] = d.z2[ri[q]]; vtd[q] = d.td[ri[q]]; vb[q] = d.b[ri[q]];
        }
    }
    double a[1];
    final_sums<1>(d, PH_A, a, red, parity);
    const double c2 = 2 * sqrt(a[0]);
    if (pair) {
#pragma unroll
        for (int q = 0; q < 2; q++) {
            if (!rok[q]) continue;
            const int i = ri[q];
            const double x = vx[q], z1 = vz1[q], z2 = vz2[q];
            const double t = x + z1 / rho1;
            const double y1 = t > 1 ? 1 : (t < 0 ? 0 : t);
            double y2 = (x + z2 / rho2) - 0.5;
            y2 = y2 * c1 / c2 + 0.5;
            d.y1[i] = y1; d.y2[i] = y2;
            double td = vtd[q];
            if (refresh) { td += inc; d.td[i] = td; }
            if (rhoUpdated || stale) d.dinv[i] = td != 0.0 ? 1.0 / td : 1.0;    // DiagonalPreconditioner::compute (:1098-1101)
            d.rhs[i] = (rho1 * y1 + rho2 * y2) - ((vb[q] + z1) + z2);           // :1091
            d.x[i] = y1;                                                         // x_sol = y1 (:1104)
        }
        if (LEADER) { d.st[out] = *si; d.st[out].rhoUpdated = 0; d.st[out].dinv_stale = 0; }
        return;
    }
    for (int q = 0; q < d.EPT; q++) {
        const int i = blockIdx.x * (T * d.EPT) + q * T + threadIdx.x;
        if (i >= d.n || !d.live[i]) continue;
        const double x = d.x[i], z1 = d.z1[i], z2 = d.z2[i];
        const double t = x + z1 / rho1;
        const double y1 = t > 1 ? 1 : (t < 0 ? 0 : t);
        double y2 = (x + z2 / rho2) - 0.5;
        y2 = y2 * c1 / c2 + 0.5;
        d.y1[i] = y1; d.y2[i] = y2;
        double td = d.td[i];
        if (refresh) { td += inc; d.td[i] = td; }
        if (rhoUpdated || stale) d.dinv[i] = td != 0.0 ? 1.0 / td : 1.0;    // DiagonalPreconditioner::compute (:1098-1101)
        d.rhs[i] = (rho1 * y1 + rho2 * y2) - ((d.b[i] + z1) + z2);           // :1091
        d.x[i] = y1;                                                         // x_sol = y1 (:1104)
    }
    if (LEADER) { d.st[out] = *si; d.st[out].rhoUpdated = 0; d.st[out].dinv_stale = 0; }
}

template <bool PAIR>
__device__ __forceinline__ void seg_b_resid(const SegDev &d, int in, int out) {
    __shared__ double red[2 * RED_MAXV * RED_MAXW];
    int parity = 0;
    const SegState *si = d.st + in;
    if (si->halt) { forward_state(d, in, out); return; }
    double pb[3] = {0.0, 0.0, 0.0};
    const double *x = d.x;
    if (PAIR && d.EPT == 2 && d.ell_w <= 8) {            // two rows per thread together (see ell_load)
        EllRow R[2];
        int ri[2]; bool rok[2]; double vrhs[2], vdi[2], g[2][8];
#pragma unroll
        for (int q = 0; q < 2; q++) {
            const int i = blockIdx.x * (T * 2) + q * T + threadIdx.x;
            const bool inr = i < d.n;
            ri[q] = inr ? i : d.n - 1;
            ell_load(d, ri[q], R[q]);
            rok[q] = inr && d.live[ri[q]];
            vrhs[q] = d.rhs[ri[q]]; vdi[q] = d.dinv[ri[q]];
        }
#pragma unroll
        for (int q = 0; q < 2; q++)
#pragma unroll
            for (int k = 0; k < 8; k++) g[q][k] = x[R[q].c[k]];
#pragma unroll
        for (int q = 0; q < 2; q++) {
            double c0 = 0.0, c1 = 0.0, c2 = 0.0;
            const double Mx = ell_sum(R[q], ri[q], g[q]);
            const double rhs = vrhs[q];
            const double r = rhs - Mx;                                        // :263
            const double p = vdi[q] * r;                                      // :286
            if (rok[q]) {
                d.r[ri[q]] = r; d.p0[ri[q]] = p;
                c0 = rhs * rhs; c1 = r * r; c2 = r * p;
            }
            pb[0] = pb[0] + c0; pb[1] = pb[1] + c1; pb[2] = pb[2] + c2;
        }
        store_partials<3>(d, PH_B, pb, red, parity);
        if (LEADER) { d.st[out] = *si; d.st[out].pcg_k = 0; d.st[out].pcg_done = 0; d.st[out].phase = 2; }
        return;
    }
    for (int q = 0; q < d.EPT; q++) {
        const int i = blockIdx.x * (T * d.EPT) + q * T + threadIdx.x;
        double c0 = 0.0, c1 = 0.0, c2 = 0.0;
        if (i < d.n && d.live[i]) {
            const double Mx = tm_row(d, i, [x](int c) { return x[c]; });
            const double rhs = d.rhs[i];
            const double r = rhs - Mx;                                        // :263
            const double p = d.dinv[i] * r;                                   // :286
            d.r[i] = r; d.p0[i] = p;
            c0 = rhs * rhs; c1 = r * r; c2 = r * p;
        }
        pb[0] = pb[0] + c0; pb[1] = pb[1] + c1; pb[2] = pb[2] + c2;
    }
    store_partials<3>(d, PH_B, pb, red, parity);
    if (LEADER) { d.st[out] = *si; d.st[out].pcg_k = 0; d.st[out].pcg_done = 0; d.st[out].phase = 2; }
}

// tmp = M p with the search-direction update of the previous PCG iteration folded in
template <bool PAIR>
__device__ __forceinline__ void seg_b_matvec(const SegDev &d, int in, int out) {
    __shared__ double red[2 * RED_MAXV * RED_MAXW];
    int parity = 0;
    const SegState *si = d.st + in;
    if (si->halt || si->pcg_done || si->phase != 2) { forward_state(d, in, out); return; }
    const int pcg_k = si->pcg_k;
    double threshold = si->threshold, absNew = si->absNew, rhsNorm2 = si->rhsNorm2;
    double beta = 0.0;
    const bool first = pcg_k == 0;
    bool done = false, zero_x = false;
    // two rows per thread, handled together (see ell_load): everything that does not depend on beta is requested before the
    // reduction below, i.e. its latency hides behind the reduction's
    // (single-problem launches only: they are latency-bound; a batch of problems fills the chip and wants the registers for occupancy)
    const bool pair = PAIR && d.EPT == 2 && d.ell_w <= 8;
    const double *pold = ((pcg_k - 1) & 1) ? d.p1 : d.p0;     // iteration k reads p_old = buffer (k-1)&1 (k >= 1), writes buffer k&1; k = 0: p0 as resid left it
    const double *zz = d.z;
    EllRow R[2];
    int ri[2] = {0, 0}; bool rin[2] = {false, false}; uint8_t rlv[2] = {0, 0};
    double rz[2] = {0.0, 0.0}, rp[2] = {0.0, 0.0};
    if (pair) {
#pragma unroll
        for (int q = 0; q < 2; q++) {
            const int i = blockIdx.x * (T * 2) + q * T + threadIdx.x;
            rin[q] = i < d.n; ri[q] = rin[q] ? i : d.n - 1;
            ell_load(d, ri[q], R[q]);
            rlv[q] = d.live[ri[q]];
            if (first) rz[q] = d.p0[ri[q]]; else { rz[q] = zz[ri[q]]; rp[q] = pold[ri[q]]; }
        }
    }
    if (first) {
        double b3[3];
        final_sums<3>(d, PH_B, b3, red, parity);
        rhsNorm2 = b3[0];
        if (rhsNorm2 == 0) { done = true; zero_x = true; }                   // :265-271
        else {
            double thr = SEG_PCG_TOL * SEG_PCG_TOL * rhsNorm2;               // :274
            if (thr < DBL_MIN) thr = DBL_MIN;
            threshold = thr;
            if (b3[1] < thr) done = true;                                     // :277
            absNew = b3[2];
        }
    } else {
        double d2[2];
        final_sums<2>(d, PH_D, d2, red, parity);
        if (d2[0] < threshold || pcg_k >= SEG_PCG_MAXITERS) done = true;     // :304-307
        else { const double absOld = absNew; absNew = d2[1]; beta = absNew / absOld; }   // :311-313
    }
    if (LEADER) {
        d.st[out] = *si;
        SegState *s = d.st + out;
        s->threshold = threshold; s->absNew = absNew; s->rhsNorm2 = rhsNorm2; s->pcg_done = done ? 1 : 0;
    }
    if (done) {
        if (zero_x)
            for (int q = 0; q < d.EPT; q++) { const int i = blockIdx.x * (T * d.EPT) + q * T + threadIdx.x; if (i < d.n) d.x[i] = 0.0; }
        return;
    }
    double *pnew = (pcg_k & 1) ? d.p1 : d.p0;
    double pc[1] = {0.0};
    if (pair) {
        double gz[2][8], gp[2][8];
#pragma unroll
        for (int q = 0; q < 2; q++)
#pragma unroll
            for (int k = 0; k < 8; k++) {
                if (first) { gz[q][k] = d.p0[R[q].c[k]]; gp[q][k] = 0.0; }
                else { gz[q][k] = zz[R[q].c[k]]; gp[q][k] = pold[R[q].c[k]]; }
            }
#pragma unroll
        for (int q = 0; q < 2; q++) {
            double g[8];
#pragma unroll
            for (int k = 0; k < 8; k++) g[k] = first ? gz[q][k] : gz[q][k] + beta * gp[q][k];
            const double Mp = ell_sum(R[q], ri[q], g);
            const double pi = first ? rz[q] : rz[q] + beta * rp[q];            // p = z + beta p (:314)
            double c = 0.0;
            if (rin[q] && rlv[q]) {
                const int i = ri[q];
                if (!first) pnew[i] = pi;
                d.tmp[i] = Mp;
                c = pi * Mp;
            }
            pc[0] = pc[0] + c;
        }
        store_partials<1>(d, PH_C, pc, red, parity);
        return;
    }
    for (int q = 0; q < d.EPT; q++) {
        const int i = blockIdx.x * (T * d.EPT) + q * T + threadIdx.x;
        double c = 0.0;
        if (i < d.n && d.live[i]) {
            double Mp, pi;
            if (first) {
                pi = d.p0[i];
                const double *p0 = d.p0;
                Mp = tm_row(d, i, [p0](int c2) { return p0[c2]; });
            } else {
                pi = zz[i] + beta * pold[i];                                   // p = z + beta p (:314)
                Mp = tm_row(d, i, [zz, pold, beta](int c2) { return zz[c2] + beta * pold[c2]; });
                pnew[i] = pi;
            }
            d.tmp[i] = Mp;
            c = pi * Mp;
        }
        pc[0] = pc[0] + c;
    }
    store_partials<1>(d, PH_C, pc, red, parity);
}

template <bool PAIR>
__device__ __forceinline__ void seg_b_update(const SegDev &d, int in, int out) {
    __shared__ double red[2 * RED_MAXV * RED_MAXW];
    int parity = 0;
    const SegState *si = d.st + in;
    if (si->halt || si->pcg_done || si->phase != 2) { forward_state(d, in, out); return; }
    const int pcg_k = si->pcg_k;
    const double absNew = si->absNew;
    const double *p = (pcg_k & 1) ? d.p1 : d.p0;
    // two rows per thread together: their operands are requested before the reduction that yields alpha (see ell_load)
    const bool pair = PAIR && d.EPT == 2;
    int ri[2] = {0, 0}; bool rok[2] = {false, false};
    double vx[2], vr[2], vp[2], vt[2], vd[2];
    if (pair) {
#pragma unroll
        for (int q = 0; q < 2; q++) {
            const int i = blockIdx.x * (T * 2) + q * T + threadIdx.x;
            const bool in = i < d.n;
            ri[q] = in ? i : d.n - 1;
            rok[q] = in && d.live[ri[q]];
            vx[q] = d.x[ri[q]]; vr[q] = d.r[ri[q]]; vp[q] = p[ri[q]]; vt[q] = d.tmp[ri[q]]; vd[q] = d.dinv[ri[q]];
        }
    }
    double c1[1];
    final_sums<1>(d, PH_C, c1, red, parity);
    const double alpha = absNew / c1[0];                                      // :295
    double pd2[2] = {0.0, 0.0};
    if (pair) {
#pragma unroll
        for (int q = 0; q < 2; q++) {
            double a = 0.0, b2 = 0.0;
            double x = vx[q], r = vr[q];
            x += alpha * vp[q];                                               // :297
            r -= alpha * vt[q];                                               // :299
            const double z = vd[q] * r;                                       // :309
            if (rok[q]) {
                const int i = ri[q];
                d.x[i] = x; d.r[i] = r; d.z[i] = z;
                a = r * r; b2 = r * z;
            }
            pd2[0] = pd2[0] + a; pd2[1] = pd2[1] + b2;
        }
        store_partials<2>(d, PH_D, pd2, red, parity);
        if (LEADER) { d.st[out] = *si; d.st[out].pcg_k = pcg_k + 1; }
        return;
    }
    for (int q = 0; q < d.EPT; q++) {
        const int i = blockIdx.x * (T * d.EPT) + q * T + threadIdx.x;
        double a = 0.0, b2 = 0.0;
        if (i < d.n && d.live[i]) {
            double x = d.x[i], r = d.r[i];
            x += alpha * p[i];                                                // :297
            r -= alpha * d.tmp[i];                                            // :299
            const double z = d.dinv[i] * r;                                   // :309
            d.x[i] = x; d.r[i] = r; d.z[i] = z;
            a = r * r; b2 = r * z;
        }
        pd2[0] = pd2[0] + a; pd2[1] = pd2[1] + b2;
    }
    store_partials<2>(d, PH_D, pd2, red, parity);
    if (LEADER) { d.st[out] = *si; d.st[out].pcg_k = pcg_k + 1; }
}

template <bool PAIR>
__device__ __forceinline__ void seg_b_post(const SegDev &d, int in, int out) {
    __shared__ double red[2 * RED_MAXV * RED_MAXW];
    int parity = 0;
    const SegState *si = d.st + in;
    if (si->halt || si->phase != 2) { forward_state(d, in, out); return; }
    const int pcg_k = si->pcg_k, rec = si->rec, cc = si->cc;
    int done = si->pcg_done;
    // two rows per thread together, operands requested before the pending reduction (see ell_load)
    const bool pair = PAIR && d.EPT == 2 && d.ell_w <= 8;
    EllRow R[2];
    int ri[2] = {0, 0}; bool rok[2] = {false, false};
    double vx[2], vy1[2], vy2[2], vb[2], vz1[2], vz2[2];
    if (pair) {
#pragma unroll
        for (int q = 0; q < 2; q++) {
            const int i = blockIdx.x * (T * 2) + q * T + threadIdx.x;
            const bool inr = i < d.n;
            ri[q] = inr ? i : d.n - 1;
            ell_load(d, ri[q], R[q]);
            rok[q] = inr && d.live[ri[q]];
            vx[q] = d.x[ri[q]]; vy1[q] = d.y1[ri[q]]; vy2[q] = d.y2[ri[q]]; vb[q] = d.b[ri[q]]; vz1[q] = d.z1[ri[q]]; vz2[q] = d.z2[ri[q]];
        }
    }
    if (!done) {                           // the exit test of the last update is still pending
        double d2[2];
        final_sums<2>(d, PH_D, d2, red, parity);
        if (pcg_k >= 1 && (d2[0] < si->threshold || pcg_k >= SEG_PCG_MAXITERS)) done = 1;
        else { if (LEADER) { d.st[out] = *si; d.st[out].halt = SEG_HALT_PCG_MORE; } return; }
    }
    const double g1 = si->gamma_val * si->rho1, g2 = si->gamma_val * si->rho2;
    // the rho2 of the NEXT iteration (rho schedule :1137-1145; if this iteration turns out to stop, the partial below is simply not used)
    const double rho2n = (si->iter + 1) % SEG_RHO_STEP == 0 ? SEG_LEARNING_FACT * si->rho2 : si->rho2;
    const double *x = d.x;
    double e5[5] = {0.0, 0.0, 0.0, 0.0, 0.0}, e2[3] = {0.0, 0.0, 0.0};
    double *xh = (rec && cc < d.ws_cap) ? d.xhist + (size_t)cc * d.n : nullptr;
    if (pair) {
        double xc[2][8];
#pragma unroll
        for (int q = 0; q < 2; q++)
#pragma unroll
            for (int k = 0; k < 8; k++) xc[q][k] = x[R[q].c[k]];
#pragma unroll
        for (int q = 0; q < 2; q++) {
            double v0 = 0.0, v1 = 0.0, v2 = 0.0, v3 = 0.0, v4 = 0.0, v5 = 0.0, v6 = 0.0, v7 = 0.0;
            if (rok[q]) {
                const int i = ri[q];
                const double xi = vx[q], y1 = vy1[q], y2 = vy2[q], bi = vb[q];
                d.z1[i] = vz1[q] + g1 * (xi - y1);                                // :1119-1120
                const double z2n = vz2[q] + g2 * (xi - y2);
                d.z2[i] = z2n;
                { const double u = (xi + z2n / rho2n) - 0.5; v7 = u * u; }
                if (xh) xh[i] = xi;
                double t1 = 0, t2 = 0;
#pragma unroll
                for (int k = 0; k < 8; k++) if (k < R[q].len) { t1 += R[q].v[k] * xc[q][k]; t2 += R[q].v[k] * (xc[q][k] >= 0.5 ? 1.0 : 0.0); }
                double Ax = 0.0; Ax += 1.0 * t1;
                double Axb = 0.0; Axb += 1.0 * t2;
                const double xb = xi >= 0.5 ? 1.0 : 0.0;
                const double d1 = xi - y1, d2 = xi - y2;
                v0 = xi * xi; v1 = d1 * d1; v2 = d2 * d2; v3 = xi * Ax; v4 = bi * xi; v5 = xb * Axb; v6 = bi * xb;
            }
            e5[0] = e5[0] + v0; e5[1] = e5[1] + v1; e5[2] = e5[2] + v2; e5[3] = e5[3] + v3; e5[4] = e5[4] + v4;
            e2[0] = e2[0] + v5; e2[1] = e2[1] + v6; e2[2] = e2[2] + v7;
        }
    } else
    for (int q = 0; q < d.EPT; q++) {
        const int i = blockIdx.x * (T * d.EPT) + q * T + threadIdx.x;
        double v0 = 0.0, v1 = 0.0, v2 = 0.0, v3 = 0.0, v4 = 0.0, v5 = 0.0, v6 = 0.0, v7 = 0.0;
        if (i < d.n && d.live[i]) {
            const double xi = x[i], y1 = d.y1[i], y2 = d.y2[i], bi = d.b[i];
            d.z1[i] = d.z1[i] + g1 * (xi - y1);                               // :1119-1120
            const double z2n = d.z2[i] + g2 * (xi - y2);
            d.z2[i] = z2n;
            { const double u = (xi + z2n / rho2n) - 0.5; v7 = u * u; }           // next iteration's ||x + z2/rho2 - 1/2||^2 (:1075-1080)
            if (xh) xh[i] = xi;                                               // x_iters column (:1113-1116)
            // A x and A round(x) in one pass over the row (compute_cost :568-572, A_ptr restricted to the live variables)
            double t1 = 0, t2 = 0;
            if (d.ell_w <= 8) {                                               // all loads of the row in flight together (cf. tm_row)
                EllRow Rr;
                ell_load(d, i, Rr);
                double xc[8];
#pragma unroll
                for (int k = 0; k < 8; k++) xc[k] = x[Rr.c[k]];
#pragma unroll
                for (int k = 0; k < 8; k++) if (k < Rr.len) { t1 += Rr.v[k] * xc[k]; t2 += Rr.v[k] * (xc[k] >= 0.5 ? 1.0 : 0.0); }
            } else {
                const int len = d.rowlen[i];
                for (int k = 0; k < len; k++) {
                    const int c = d.ecol[(size_t)k * d.n + i];
                    const double xc = x[c], a = d.eval[(size_t)k * d.n + i];
                    t1 += a * xc;
                    t2 += a * (xc >= 0.5 ? 1.0 : 0.0);                       // fixed variables hold x = 0
                }
            }
            double Ax = 0.0; Ax += 1.0 * t1;
            double Axb = 0.0; Axb += 1.0 * t2;
            const double xb = xi >= 0.5 ? 1.0 : 0.0;
            const double d1 = xi - y1, d2 = xi - y2;
            v0 = xi * xi; v1 = d1 * d1; v2 = d2 * d2; v3 = xi * Ax; v4 = bi * xi; v5 = xb * Axb; v6 = bi * xb;
        }
        e5[0] = e5[0] + v0; e5[1] = e5[1] + v1; e5[2] = e5[2] + v2; e5[3] = e5[3] + v3; e5[4] = e5[4] + v4;
        e2[0] = e2[0] + v5; e2[1] = e2[1] + v6; e2[2] = e2[2] + v7;
    }
    block_sum<T, 5>(e5, red, parity);
    block_sum<T, 3>(e2, red, parity);
    if (threadIdx.x == 0) {
        for (int k = 0; k < 5; k++) part_ptr(d, PH_E, k)[blockIdx.x] = e5[k];
        part_ptr(d, PH_E, 5)[blockIdx.x] = e2[0];
        part_ptr(d, PH_E, 6)[blockIdx.x] = e2[1];
        part_ptr(d, PH_A, 0)[blockIdx.x] = e2[2];
    }
    if (LEADER) {
        d.st[out] = *si;
        SegState *s = d.st + out;
        s->pcg_done = 1; s->last_pcg = pcg_k; s->pcg_total += pcg_k; s->outer_total++;
        if (pcg_k > s->pcg_max) s->pcg_max = pcg_k;
        if (rec) s->cc = cc + 1;
        s->have_prev = 1; s->phase = 0;
    }
}

__global__ void seg_k_pack(SegDev d, const int *live_idx, int rows, int ws, double *out) {
    const long total = (long)rows * ws;
    for (long e = (long)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (long)gridDim.x * blockDim.x) {
        const int r = (int)(e / ws), c = (int)(e % ws);
        out[e] = c < d.ws_cap ? d.xhist[(size_t)c * d.n + live_idx[r]] : 0.0;    // columns beyond the staged window: zeros, like the reference's matrix
    }
}


__global__ void __launch_bounds__(T) seg_kb_init_c1(const SegDev *devs) { const SegDev &d = devs[blockIdx.y]; if ((int)blockIdx.x >= d.G) return; seg_b_init(d, d.c1_init); }
__global__ void seg_kb_collect(const SegDev *devs, int parity, SegState *out) { out[blockIdx.x] = devs[blockIdx.x].st[parity]; }

// entry points: one problem (descriptor by value) / a batch of problems in lockstep (descriptor array, blockIdx.y = problem;
// workgroups beyond a problem's own count leave at once).  The bodies above use blockIdx.x only.
__global__ void __launch_bounds__(T) seg_k_init(SegDev d, double c1) { seg_b_init(d, c1); }
__global__ void __launch_bounds__(T) seg_kb_init(const SegDev *devs, double c1) { const SegDev &d = devs[blockIdx.y]; if ((int)blockIdx.x >= d.G) return; seg_b_init(d, c1); }
__global__ void seg_k_set_window(SegDev d, int in, int out, int iter_start, int iter_end, int mode) { seg_b_set_window(d, in, out, iter_start, iter_end, mode); }
__global__ void seg_kb_set_window(const SegDev *devs, int in, int out, int iter_start, int iter_end, int mode) { const SegDev &d = devs[blockIdx.y]; if ((int)blockIdx.x >= d.G) return; seg_b_set_window(d, in, out, iter_start, iter_end, mode); }
__global__ void seg_k_resume(SegDev d, int in, int out, int reset_pcg_max) { seg_b_resume(d, in, out, reset_pcg_max); }
__global__ void seg_kb_resume(const SegDev *devs, int in, int out, int reset_pcg_max) { const SegDev &d = devs[blockIdx.y]; if ((int)blockIdx.x >= d.G) return; seg_b_resume(d, in, out, reset_pcg_max); }
__global__ void __launch_bounds__(T) seg_k_prep(SegDev d, int in, int out, int do_prep) { seg_b_prep(d, in, out, do_prep); }
__global__ void __launch_bounds__(T) seg_kb_prep(const SegDev *devs, int in, int out, int do_prep) { const SegDev &d = devs[blockIdx.y]; if ((int)blockIdx.x >= d.G) return; seg_b_prep(d, in, out, do_prep); }
__global__ void __launch_bounds__(T) seg_k_yrhs(SegDev d, int in, int out) { seg_b_yrhs<true>(d, in, out); }
__global__ void __launch_bounds__(T) seg_kb_yrhs(const SegDev *devs, int in, int out) { const SegDev &d = devs[blockIdx.y]; if ((int)blockIdx.x >= d.G) return; seg_b_yrhs<false>(d, in, out); }
__global__ void __launch_bounds__(T) seg_k_resid(SegDev d, int in, int out) { seg_b_resid<true>(d, in, out); }
__global__ void __launch_bounds__(T) seg_kb_resid(const SegDev *devs, int in, int out) { const SegDev &d = devs[blockIdx.y]; if ((int)blockIdx.x >= d.G) return; seg_b_resid<false>(d, in, out); }
__global__ void __launch_bounds__(T) seg_k_matvec(SegDev d, int in, int out) { seg_b_matvec<true>(d, in, out); }
__global__ void __launch_bounds__(T) seg_kb_matvec(const SegDev *devs, int in, int out) { const SegDev &d = devs[blockIdx.y]; if ((int)blockIdx.x >= d.G) return; seg_b_matvec<false>(d, in, out); }
__global__ void __launch_bounds__(T) seg_k_update(SegDev d, int in, int out) { seg_b_update<true>(d, in, out); }
__global__ void __launch_bounds__(T) seg_kb_update(const SegDev *devs, int in, int out) { const SegDev &d = devs[blockIdx.y]; if ((int)blockIdx.x >= d.G) return; seg_b_update<false>(d, in, out); }
__global__ void __launch_bounds__(T) seg_k_post(SegDev d, int in, int out) { seg_b_post<true>(d, in, out); }
__global__ void __launch_bounds__(T) seg_kb_post(const SegDev *devs, int in, int out) { const SegDev &d = devs[blockIdx.y]; if ((int)blockIdx.x >= d.G) return; seg_b_post<false>(d, in, out); }
}  // namespace

// ---------------------------------------------------------------------------------------------------------------------
hipError_t seg_launch_init(const SegDev &d, double c1, hipStream_t s) {
    hipLaunchKernelGGL(seg_k_init, dim3(d.G), dim3(T), 0, s, d, c1);
    return hipGetLastError();
}

// every launch flips the ping-pong index *parity (state read from st[*parity], written to st[*parity ^ 1])
#define SEG_LAUNCH(kernel, grid, ...)                                                        \
    do {                                                                                     \
        hipLaunchKernelGGL(kernel, dim3(grid), dim3(T), 0, s, d, *parity, *parity ^ 1, ##__VA_ARGS__); \
        *parity ^= 1;                                                                        \
    } while (0)

hipError_t seg_launch_set_window(const SegDev &d, int iter_start, int iter_end, int mode, int *parity, hipStream_t s) {
    hipLaunchKernelGGL(seg_k_set_window, dim3(1), dim3(1), 0, s, d, *parity, *parity ^ 1, iter_start, iter_end, mode);
    *parity ^= 1;
    return hipGetLastError();
}

hipError_t seg_launch_fix(const SegDev &d, int n_live_new, double c1_new, int *parity, hipStream_t s) {
    SEG_LAUNCH(seg_k_fix, d.G, n_live_new, c1_new);
    return hipGetLastError();
}

hipError_t seg_enqueue_prep(const SegDev &d, int *parity, hipStream_t s) {      // head of a batch of iterations
    SEG_LAUNCH(seg_k_prep, d.G, 1);
    return hipGetLastError();
}

hipError_t seg_enqueue_iterations(const SegDev &d, int iters, int kmax, int *parity, hipStream_t s) {
    for (int it = 0; it < iters; it++) {
        SEG_LAUNCH(seg_k_yrhs, d.G);
        SEG_LAUNCH(seg_k_resid, d.G);
        for (int k = 0; k < kmax; k++) { SEG_LAUNCH(seg_k_matvec, d.G); SEG_LAUNCH(seg_k_update, d.G); }
        SEG_LAUNCH(seg_k_post, d.G);
    }
    return hipGetLastError();
}

hipError_t seg_enqueue_pcg_more(const SegDev &d, int pairs, int *parity, hipStream_t s) {
    hipLaunchKernelGGL(seg_k_resume, dim3(1), dim3(1), 0, s, d, *parity, *parity ^ 1, 0);
    *parity ^= 1;
    for (int k = 0; k < pairs; k++) { SEG_LAUNCH(seg_k_matvec, d.G); SEG_LAUNCH(seg_k_update, d.G); }
    SEG_LAUNCH(seg_k_post, d.G);
    return hipGetLastError();
}

hipError_t seg_launch_copy(const SegDev &d, int reset_pcg_max, int *parity, hipStream_t s) {   // state copy: flips the ping-pong parity
    hipLaunchKernelGGL(seg_k_resume, dim3(1), dim3(1), 0, s, d, *parity, *parity ^ 1, reset_pcg_max);
    *parity ^= 1;
    return hipGetLastError();
}

hipError_t seg_enqueue_finalize(const SegDev &d, int *parity, hipStream_t s) {
    SEG_LAUNCH(seg_k_prep, d.G, 0);
    return hipGetLastError();
}

hipError_t seg_launch_pack_xiters(const SegDev &d, const int *live_idx, int rows, int ws, double *out, hipStream_t s) {
    hipLaunchKernelGGL(seg_k_pack, dim3(256), dim3(256), 0, s, d, live_idx, rows, ws, out);
    return hipGetLastError();
}

// ---- the same launch sequences for a batch of problems (devs: device array of descriptors; grid = (largest workgroup count, problems)) ----
#define SEG_LAUNCH_B(kernel, gx, ...)                                                                          \
    do {                                                                                                       \
        hipLaunchKernelGGL(kernel, dim3(gx, B), dim3(T), 0, s, devs, *parity, *parity ^ 1, ##__VA_ARGS__);      \
        *parity ^= 1;                                                                                          \
    } while (0)

hipError_t segb_launch_init(const SegDev *devs, int B, int Gmax, hipStream_t s) {
    hipLaunchKernelGGL(seg_kb_init_c1, dim3(Gmax, B), dim3(T), 0, s, devs);
    return hipGetLastError();
}
hipError_t segb_launch_set_window(const SegDev *devs, int B, int iter_start, int iter_end, int mode, int *parity, hipStream_t s) {
    hipLaunchKernelGGL(seg_kb_set_window, dim3(1, B), dim3(1), 0, s, devs, *parity, *parity ^ 1, iter_start, iter_end, mode);
    *parity ^= 1;
    return hipGetLastError();
}
hipError_t segb_launch_copy(const SegDev *devs, int B, int reset_pcg_max, int *parity, hipStream_t s) {
    hipLaunchKernelGGL(seg_kb_resume, dim3(1, B), dim3(1), 0, s, devs, *parity, *parity ^ 1, reset_pcg_max);
    *parity ^= 1;
    return hipGetLastError();
}
hipError_t segb_enqueue_iterations(const SegDev *devs, int B, int Gmax, int iters, int kmax, int *parity, hipStream_t s) {
    // (a batch keeps the prep launch per iteration: with B x G workgroups the redundant finalisation inside yrhs costs more than the
    //  launch it saves -- 150 vs 137 ms for 100 images at 10^4 nodes; the single-problem chain drops it)
    for (int it = 0; it < iters; it++) {
        SEG_LAUNCH_B(seg_kb_prep, Gmax, 1);
        SEG_LAUNCH_B(seg_kb_yrhs, Gmax);
        SEG_LAUNCH_B(seg_kb_resid, Gmax);
        for (int k = 0; k < kmax; k++) { SEG_LAUNCH_B(seg_kb_matvec, Gmax); SEG_LAUNCH_B(seg_kb_update, Gmax); }
        SEG_LAUNCH_B(seg_kb_post, Gmax);
    }
    return hipGetLastError();
}
hipError_t segb_enqueue_pcg_more(const SegDev *devs, int B, int Gmax, int pairs, int *parity, hipStream_t s) {
    hipLaunchKernelGGL(seg_kb_resume, dim3(1, B), dim3(1), 0, s, devs, *parity, *parity ^ 1, 0);
    *parity ^= 1;
    for (int k = 0; k < pairs; k++) { SEG_LAUNCH_B(seg_kb_matvec, Gmax); SEG_LAUNCH_B(seg_kb_update, Gmax); }
    SEG_LAUNCH_B(seg_kb_post, Gmax);
    return hipGetLastError();
}
hipError_t segb_enqueue_finalize(const SegDev *devs, int B, int Gmax, int *parity, hipStream_t s) {
    SEG_LAUNCH_B(seg_kb_prep, Gmax, 0);
    return hipGetLastError();
}
hipError_t segb_collect_states(const SegDev *devs, int B, int parity, SegState *out, hipStream_t s) {
    hipLaunchKernelGGL(seg_kb_collect, dim3(B), dim3(1), 0, s, devs, parity, out);
    return hipGetLastError();
}
