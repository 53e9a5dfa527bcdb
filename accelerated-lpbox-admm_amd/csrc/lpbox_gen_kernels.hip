// lpbox_gen_kernels.hip -- gfx950 kernels of the GENERIC constrained binary-QP path (ADMM_bqp, SEGcpp:1384-1832; see lpbox_gen.h).
//   prep -> [fin] -> y -> rhs_cols -> rows(y1) -> resid -> [fin] -> K x { rows(p) -> pcg_cols -> [fin] -> pcg_upd -> [fin] } -> post -> [fin]
//   -> rows(x) -> dual
// Sparse products follow Eigen's RowMajor sparse * dense (SEGh:17): per row tmp = sum val * v[col] in ascending column order,
// res = 0 + 1.0 * tmp; the matrix expression adds its three terms in the order A, C, E (SEGcpp:361-411).  As in the large-instance LP
// path the PCG search direction p = z + beta p is recomputed on the fly for gathered entries (one packed (z, p) read each), rows and
// columns are summed by one lane in ascending index order, no FMA contraction, IEEE divide / sqrt.
#include "lpbox_gen.h"
#include "lpbox_dev_common.h"

#include <float.h>

namespace {

constexpr int T = GEN_T;
#define LEADER (blockIdx.x == 0 && threadIdx.x == 0)

__device__ __forceinline__ void forward_state(const GenDev &d, int in, int out) {
    if (LEADER) d.st[out] = d.st[in];
}

template <int NV>
__device__ __forceinline__ void store_partials(const GenDev &d, int slot0, double (&v)[NV], double *red, int &parity) {
    block_sum<T, NV>(v, red, parity);
    if (threadIdx.x == 0) {
#pragma unroll
        for (int k = 0; k < NV; k++) d.part[(size_t)(slot0 + k) * d.G + blockIdx.x] = v[k];
    }
}

__device__ __forceinline__ double eigen_res(double tmp) { double r = 0.0; r += 1.0 * tmp; return r; }   // res[i] = 0 + alpha * tmp

__global__ void __launch_bounds__(T) gen_k_fin(GenDev d, int nv) {
    __shared__ double red[2 * RED_MAXV * RED_MAXW];
    int parity = 0;
    fin_reduce<T>(d.part, d.G, nv, d.red, red, parity);
}

// A x for row j with x read through `ld` (ascending columns, diagonal included)
template <typename LD>
__device__ __forceinline__ double a_row(const GenDev &d, int j, const double *vals, LD ld) {
    double tmp = 0;
    const int k1 = d.aptr[j + 1];
    int k = d.aptr[j];
    for (; k + 4 <= k1; k += 4) {
        const double v0 = ld(d.aidx[k]), v1 = ld(d.aidx[k + 1]), v2 = ld(d.aidx[k + 2]), v3 = ld(d.aidx[k + 3]);
        tmp += vals[k] * v0; tmp += vals[k + 1] * v1; tmp += vals[k + 2] * v2; tmp += vals[k + 3] * v3;
    }
    for (; k < k1; k++) tmp += vals[k] * ld(d.aidx[k]);
    return eigen_res(tmp);
}
// (scaled transpose row j) . q = sum over the column's entries in ascending row order
__device__ __forceinline__ double col_dot(const GenCsr &c, const double *vals, int j, const double *q) {
    double tmp = 0;
    const int k1 = c.ptr[j + 1];
    int k = c.ptr[j];
    for (; k + 4 <= k1; k += 4) {
        const double v0 = q[c.idx[k]], v1 = q[c.idx[k + 1]], v2 = q[c.idx[k + 2]], v3 = q[c.idx[k + 3]];
        tmp += vals[k] * v0; tmp += vals[k + 1] * v1; tmp += vals[k + 2] * v2; tmp += vals[k + 3] * v3;
    }
    for (; k < k1; k++) tmp += vals[k] * q[c.idx[k]];
    return eigen_res(tmp);
}

__global__ void __launch_bounds__(T) gen_k_init(GenDev d, double c1, const double *x0) {     // SEGcpp:1430-1560
    __shared__ double red[2 * RED_MAXV * RED_MAXW];
    int parity = 0;
    const GenParams &P = d.prm;
    const double rho = P.initial_rho;
    double pc[2] = {0.0, 0.0};
    if (blockIdx.x < d.G)
        for (int s = 0; s < d.EPT; s++) {
            const int j = blockIdx.x * (T * d.EPT) + s * T + threadIdx.x;
            double c0 = 0.0, c1v = 0.0;
            if (j < d.n) {
                const double xj = x0[j];
                d.x[j] = xj; d.xt[j] = xj; d.y1[j] = xj; d.y2[j] = xj; d.best[j] = xj; d.gsrc[j] = xj;
                d.z1[j] = 0.0; d.z2[j] = 0.0; d.r[j] = 0.0; d.z[j] = 0.0; d.tmp[j] = 0.0; d.p0[j] = 0.0; d.p1[j] = 0.0; d.rhs[j] = 0.0;
                d.zp[j] = make_double2(0.0, 0.0);
                for (int k = d.aptr[j]; k < d.aptr[j + 1]; k++) d.tmval[k] = 2 * d.aval[k];            // 2 * A (:1482)
                d.tmval[d.adiag[j]] += rho + rho;                                                          // diagonal += rho1 + rho2 (:1483)
                double pd = d.tmval[d.adiag[j]];
                if (d.eq) {                                                                                // Csq_diag (:1513-1526)
                    double sq = 0;
                    for (int k = d.Cc.ptr[j]; k < d.Cc.ptr[j + 1]; k++) { const double v = d.Cc.val[k]; if (v != 0.0) sq += v * v; d.Cc_sv[k] = rho * v; }
                    d.Csq[j] = sq; pd += rho * sq;
                }
                if (d.ineq) {                                                                              // Esq_diag (:1535-1548)
                    double sq = 0;
                    for (int k = d.Ec.ptr[j]; k < d.Ec.ptr[j + 1]; k++) { const double v = d.Ec.val[k]; if (v != 0.0) sq += v * v; d.Ec_sv[k] = rho * v; }
                    d.Esq[j] = sq; pd += rho * sq;
                }
                d.pdiag[j] = pd; d.dinv[j] = 1.0;
                const double ax = a_row(d, j, d.aval, [x0](int c) { return x0[c]; });                       // best_bin_obj = cost(x0) (:1560)
                c0 = xj * ax; c1v = d.b[j] * xj;
            }
            pc[0] = pc[0] + c0; pc[1] = pc[1] + c1v;
        }
    if (blockIdx.x < d.G) store_partials<2>(d, 0, pc, red, parity);
    if (blockIdx.x < d.Gm) for (int s = 0; s < d.EPTm; s++) { const int i = blockIdx.x * (T * d.EPTm) + s * T + threadIdx.x; if (i < d.m) { d.z3[i] = 0.0; d.qC[i] = 0.0; } }
    if (blockIdx.x < d.Gl) for (int s = 0; s < d.EPTl; s++) { const int i = blockIdx.x * (T * d.EPTl) + s * T + threadIdx.x; if (i < d.l) { d.z4[i] = 0.0; d.y3[i] = 0.0; d.fy[i] = 0.0; d.Ex[i] = 0.0; d.qE[i] = 0.0; } }
    if (LEADER) {
        GenState *s = d.st;
        memset(s, 0, sizeof(GenState));
        s->rho1 = s->rho2 = s->rho3 = s->rho4 = s->prev_rho1 = s->prev_rho2 = s->prev_rho3 = s->prev_rho4 = rho;
        s->gamma_val = P.gamma_val; s->std_obj = 1.0; s->rhoUpdated = 1; s->c1 = c1;
        d.st[1] = d.st[0];
    }
}

__global__ void gen_k_init2(GenDev d) {
    const double v = d.red[0] + d.red[1];
    d.st[0].best_bin_obj = v; d.st[1].best_bin_obj = v;
}

__global__ void gen_k_resume(GenDev d, int in, int out, int reset_pcg_max) {
    d.st[out] = d.st[in];
    if (d.st[out].halt == GEN_HALT_PCG_MORE) d.st[out].halt = GEN_HALT_NONE;
    if (reset_pcg_max) d.st[out].pcg_max = 0;
}

// finalise the previous iteration from red[0..7) = x.x, |x-y1|^2, |x-y2|^2, x.Ax, b.x, xb.A xb, b.xb (SEGcpp:1742-1794), then the
// partial of ||x + z2/rho2 - 1/2||^2 for the next one
__global__ void __launch_bounds__(T) gen_k_prep(GenDev d, int in, int out, int do_prep) {
    __shared__ double red[2 * RED_MAXV * RED_MAXW];
    int parity = 0;
    const GenState *si = d.st + in;
    const GenParams &P = d.prm;
    const int halt0 = si->halt, have_prev = si->have_prev, it = si->iter;
    double rho2 = si->rho2;
    const bool fin = !halt0 && have_prev;
    bool will_stop = false;
    if (fin) {       // every thread needs to know whether the loop goes on (and the rho2 of the next iteration)
        const double xn = sqrt(d.red[0]);
        const double t0 = (xn < 2.2204e-16) ? 2.2204e-16 : xn;
        if (sqrt(d.red[1]) / t0 <= P.stop_threshold && sqrt(d.red[2]) / t0 <= P.stop_threshold) will_stop = true;
        else if ((it + 1) % P.rho_change_step == 0) rho2 = P.learning_fact * rho2;
    }
    if (LEADER) {
        d.st[out] = *si;
        GenState *s = d.st + out;
        if (fin) {
            s->have_prev = 0;
            const double xn = sqrt(d.red[0]);
            const double t0 = (xn < 2.2204e-16) ? 2.2204e-16 : xn;
            s->cvg1 = sqrt(d.red[1]) / t0; s->cvg2 = sqrt(d.red[2]) / t0;                                   // :1742-1744
            bool stopped = false;
            if (s->cvg1 <= P.stop_threshold && s->cvg2 <= P.stop_threshold) { s->stop = GEN_STOP_XYY; stopped = true; }   // :1745
            else {
                if ((it + 1) % P.rho_change_step == 0) {                                                     // :1753-1770
                    s->prev_rho1 = s->rho1; s->prev_rho2 = s->rho2;
                    s->rho1 = P.learning_fact * s->rho1; s->rho2 = P.learning_fact * s->rho2;
                    if (d.eq) { s->prev_rho3 = s->rho3; s->rho3 = P.learning_fact * s->rho3; }
                    if (d.ineq) { s->prev_rho4 = s->rho4; s->rho4 = P.learning_fact * s->rho4; }
                    const double g = s->gamma_val * P.gamma_factor;
                    s->gamma_val = g < 1.0 ? 1.0 : g;
                    s->rhoUpdated = 1; s->rcr = P.learning_fact - 1.0;
                }
                s->obj_val = d.red[3] + d.red[4];                                                            // :1772
                const int H = P.history_size;
                int hn = s->hist_n;
                if (hn < H) s->hist[hn] = s->obj_val;
                else { for (int k = 0; k < H - 1; k++) s->hist[k] = s->hist[k + 1]; s->hist[H - 1] = s->obj_val; }
                if (hn < 0x3fffffff) hn++;
                s->hist_n = hn;
                if (hn >= H) {                                                                               // :482-507, :574-585
                    double mean = 0;
                    for (int k = 0; k < H; k++) mean += s->hist[k];
                    mean /= (double)H;
                    double dev = 0;
                    for (int k = 0; k < H; k++) dev += (s->hist[k] - mean) * (s->hist[k] - mean);
                    dev /= (double)(H - 1);
                    const double sd = (dev == 0) ? 0.0 : sqrt(dev);
                    s->std_obj = sd / fabs(s->hist[H - 1]);
                }
                if (s->std_obj <= P.std_threshold) { s->stop = GEN_STOP_OBJSTD; stopped = true; }             // :1777
                else {
                    s->cur_obj = d.red[5] + d.red[6];                                                        // :1786-1793
                    if (s->best_bin_obj >= s->cur_obj) { s->best_bin_obj = s->cur_obj; s->copy_best = 1; }
                }
            }
            if (stopped) s->halt = GEN_HALT_STOP;
            else s->iter = it + 1;
        }
        if (!s->halt && s->iter >= P.max_iters) s->halt = GEN_HALT_END;
        if (!s->halt && do_prep) s->phase = 1;
    }
    // (the std stop is only known to the leader; a workgroup that computes an unneeded partial does no harm)
    const int next_iter = (fin && !will_stop) ? it + 1 : it;
    if (!do_prep || halt0 || will_stop || next_iter >= P.max_iters || blockIdx.x >= d.G) return;
    double pa[1] = {0.0};
    for (int q = 0; q < d.EPT; q++) {
        const int j = blockIdx.x * (T * d.EPT) + q * T + threadIdx.x;
        double c = 0.0;
        if (j < d.n) { const double u = (d.x[j] + d.z2[j] / rho2) - 0.5; c = u * u; }
        pa[0] = pa[0] + c;
    }
    store_partials<1>(d, 0, pa, red, parity);
}

// y1, y2, y3, matrix / preconditioner refresh (:1619-1650), rhs base (:1656), preconditioner (:1711-1718), best_sol copy (:1792)
__global__ void __launch_bounds__(T) gen_k_y(GenDev d, int in, int out) {
    const GenState *si = d.st + in;
    if (si->halt || si->phase != 1) { forward_state(d, in, out); return; }
    const GenParams &P = d.prm;
    const double rho1 = si->rho1, rho2 = si->rho2, rho4 = si->rho4, c1 = si->c1;
    const int it = si->iter, rhoUpdated = si->rhoUpdated, copy_best = si->copy_best;
    const bool refresh = it != 0 && rhoUpdated;
    const double inc = si->rcr * (si->prev_rho1 + si->prev_rho2), s3 = si->rcr * si->prev_rho3, s4 = si->rcr * si->prev_rho4;
    const double c2 = 2 * sqrt(d.red[0]);
    const int type = (d.eq ? 1 : 0) | (d.ineq ? 2 : 0);
    if (blockIdx.x < d.G)
        for (int q = 0; q < d.EPT; q++) {
            const int j = blockIdx.x * (T * d.EPT) + q * T + threadIdx.x;
            if (j >= d.n) continue;
            const double x = d.x[j], z1 = d.z1[j], z2 = d.z2[j];
            if (copy_best) d.best[j] = x;
            const double t = x + z1 / rho1;
            const double y1 = t > 1 ? 1 : (t < 0 ? 0 : t);                       // :1598-1601
            double y2 = (x + z2 / rho2) - 0.5;                                   // :1603-1606, :553-558
            y2 = y2 * c1 / c2 + 0.5;
            d.y1[j] = y1; d.y2[j] = y2;
            double pd = d.pdiag[j];
            if (refresh) {
                d.tmval[d.adiag[j]] += inc;                                      // :1623
                if (type != 0) pd += inc;                                        // :1626
                if (d.eq) pd += s3 * d.Csq[j];                                   // :1641
                if (d.ineq) pd += s4 * d.Esq[j];                                 // :1646
                d.pdiag[j] = pd;
            }
            if (rhoUpdated) {                                                    // :1711-1718
                const double dg = type == 0 ? d.tmval[d.adiag[j]] : pd;
                d.dinv[j] = (dg != 0.0) ? 1.0 / dg : 1.0;
            }
            d.rhs[j] = (rho1 * y1 + rho2 * y2) - ((d.b[j] + z1) + z2);            // :1656
            d.gsrc[j] = y1;                                                       // x_sol = y1 (:1721)
        }
    if (refresh) {                                                                // the scaled transposes (:1643, :1648)
        const long stride = (long)gridDim.x * T;
        if (d.eq) for (long k = (long)blockIdx.x * T + threadIdx.x; k < d.Cnnz; k += stride) d.Cc_sv[k] = P.learning_fact * d.Cc_sv[k];
        if (d.ineq) for (long k = (long)blockIdx.x * T + threadIdx.x; k < d.Ennz; k += stride) d.Ec_sv[k] = P.learning_fact * d.Ec_sv[k];
    }
    if (d.ineq && blockIdx.x < d.Gl)
        for (int q = 0; q < d.EPTl; q++) {
            const int i = blockIdx.x * (T * d.EPTl) + q * T + threadIdx.x;
            if (i >= d.l) continue;
            const double f = d.f[i];
            const double v = f - d.Ex[i] - d.z4[i] / rho4;                        // :1609-1613
            const double y3 = v < 0 ? 0 : v;
            d.y3[i] = y3; d.fy[i] = f - y3;
        }
    if (LEADER) { d.st[out] = *si; d.st[out].rhoUpdated = 0; d.st[out].copy_best = 0; }
}

__global__ void __launch_bounds__(T) gen_k_rhs_cols(GenDev d, int in, int out) {   // :1663-1706
    const GenState *si = d.st + in;
    if (si->halt || si->phase != 1) { forward_state(d, in, out); return; }
    for (int q = 0; q < d.EPT; q++) {
        const int j = blockIdx.x * (T * d.EPT) + q * T + threadIdx.x;
        if (j >= d.n) continue;
        double r_ = d.rhs[j];
        if (d.eq) { r_ += col_dot(d.Cc, d.Cc_sv, j, d.d); r_ -= col_dot(d.Cc, d.Cc.val, j, d.z3); }
        if (d.ineq) { r_ += col_dot(d.Ec, d.Ec_sv, j, d.fy); r_ -= col_dot(d.Ec, d.Ec.val, j, d.z4); }
        d.rhs[j] = r_;
    }
    forward_state(d, in, out);
}

// qC = C v, qE = E v.  mode 0: v = gsrc.  mode 1: v = the PCG search direction; the exit test / beta of the previous PCG iteration is
// evaluated here (SEGcpp:445-467) -- this launch exists (one workgroup) even without constraints.
__global__ void __launch_bounds__(T) gen_k_rows(GenDev d, int in, int out, int mode) {
    const GenState *si = d.st + in;
    if (si->halt) { forward_state(d, in, out); return; }
    const GenParams &P = d.prm;
    double beta = 0.0;
    bool first = false;
    if (mode == 1) {
        if (si->phase != 2 || si->pcg_done) { forward_state(d, in, out); return; }
        const int k = si->pcg_k;
        double threshold = si->threshold, absNew = si->absNew, rhsNorm2 = si->rhsNorm2;
        bool done = false; int zero_x = 0;
        first = k == 0;
        if (first) {
            rhsNorm2 = d.red[0];
            if (rhsNorm2 == 0) { done = true; zero_x = 1; }                      // :424-430
            else {
                double thr = P.pcg_tol * P.pcg_tol * rhsNorm2;                   // :433
                if (thr < DBL_MIN) thr = DBL_MIN;
                threshold = thr;
                if (d.red[1] < thr) done = true;                                 // :435
                absNew = d.red[2];
            }
        } else {
            if (d.red[0] < threshold || k >= P.pcg_maxiters) done = true;        // :453-456, :445
            else { const double absOld = absNew; absNew = d.red[1]; beta = absNew / absOld; }   // :460-463
        }
        if (LEADER) {
            d.st[out] = *si;
            GenState *s = d.st + out;
            s->threshold = threshold; s->absNew = absNew; s->rhsNorm2 = rhsNorm2; s->beta = beta;
            s->pcg_done = done ? 1 : 0; s->pcg_first = zero_x;
        }
        if (done) return;
    } else forward_state(d, in, out);
    const double *src = mode == 0 ? d.gsrc : d.p0;
    const bool plain = mode == 0 || first;
    auto row = [&](const GenCsr &M, int i) {
        double tmp = 0;
        const int k1 = M.ptr[i + 1];
        int k = M.ptr[i];
        if (plain) {
            for (; k + 4 <= k1; k += 4) {
                const double v0 = src[M.idx[k]], v1 = src[M.idx[k + 1]], v2 = src[M.idx[k + 2]], v3 = src[M.idx[k + 3]];
                tmp += M.val[k] * v0; tmp += M.val[k + 1] * v1; tmp += M.val[k + 2] * v2; tmp += M.val[k + 3] * v3;
            }
            for (; k < k1; k++) tmp += M.val[k] * src[M.idx[k]];
        } else {
            for (; k + 4 <= k1; k += 4) {
                const double2 v0 = d.zp[M.idx[k]], v1 = d.zp[M.idx[k + 1]], v2 = d.zp[M.idx[k + 2]], v3 = d.zp[M.idx[k + 3]];
                tmp += M.val[k] * (v0.x + beta * v0.y); tmp += M.val[k + 1] * (v1.x + beta * v1.y);
                tmp += M.val[k + 2] * (v2.x + beta * v2.y); tmp += M.val[k + 3] * (v3.x + beta * v3.y);
            }
            for (; k < k1; k++) { const double2 v = d.zp[M.idx[k]]; tmp += M.val[k] * (v.x + beta * v.y); }
        }
        return eigen_res(tmp);
    };
    if (d.eq && blockIdx.x < d.Gm)
        for (int s = 0; s < d.EPTm; s++) { const int i = blockIdx.x * (T * d.EPTm) + s * T + threadIdx.x; if (i < d.m) d.qC[i] = row(d.Cr, i); }
    if (d.ineq && blockIdx.x < d.Gl)
        for (int s = 0; s < d.EPTl; s++) { const int i = blockIdx.x * (T * d.EPTl) + s * T + threadIdx.x; if (i < d.l) d.qE[i] = row(d.Er, i); }
}

__global__ void __launch_bounds__(T) gen_k_resid(GenDev d, int in, int out) {       // :419-441
    __shared__ double red[2 * RED_MAXV * RED_MAXW];
    int parity = 0;
    const GenState *si = d.st + in;
    if (si->halt || si->phase != 1) { forward_state(d, in, out); return; }
    double pb[3] = {0.0, 0.0, 0.0};
    const double *gs = d.gsrc;
    for (int q = 0; q < d.EPT; q++) {
        const int j = blockIdx.x * (T * d.EPT) + q * T + threadIdx.x;
        double c0 = 0.0, c1 = 0.0, c2 = 0.0;
        if (j < d.n) {
            double Mx = a_row(d, j, d.tmval, [gs](int c) { return gs[c]; });
            if (d.eq) Mx += col_dot(d.Cc, d.Cc_sv, j, d.qC);
            if (d.ineq) Mx += col_dot(d.Ec, d.Ec_sv, j, d.qE);
            const double rhs = d.rhs[j];
            const double r = rhs - Mx;
            const double p = d.dinv[j] * r;
            d.xt[j] = d.y1[j]; d.r[j] = r; d.p0[j] = p;
            c0 = rhs * rhs; c1 = r * r; c2 = r * p;
        }
        pb[0] = pb[0] + c0; pb[1] = pb[1] + c1; pb[2] = pb[2] + c2;
    }
    store_partials<3>(d, 0, pb, red, parity);
    if (LEADER) { d.st[out] = *si; d.st[out].pcg_k = 0; d.st[out].pcg_done = 0; d.st[out].pcg_first = 0; d.st[out].phase = 2; }
}

__global__ void __launch_bounds__(T) gen_k_pcg_cols(GenDev d, int in, int out) {    // tmp = M p, partial p.tmp (:447-448)
    __shared__ double red[2 * RED_MAXV * RED_MAXW];
    int parity = 0;
    const GenState *si = d.st + in;
    if (si->halt || si->phase != 2) { forward_state(d, in, out); return; }
    if (si->pcg_done) {
        if (si->pcg_first)                                                       // rhs == 0: x := 0 (:424-430)
            for (int q = 0; q < d.EPT; q++) { const int j = blockIdx.x * (T * d.EPT) + q * T + threadIdx.x; if (j < d.n) d.xt[j] = 0.0; }
        if (LEADER) { d.st[out] = *si; d.st[out].pcg_first = 0; }
        return;
    }
    const int k = si->pcg_k;
    const double beta = si->beta;
    const bool first = k == 0;
    const double *pold = ((k - 1) & 1) ? d.p1 : d.p0;
    double *pnew = (k & 1) ? d.p1 : d.p0;
    const double *p0 = d.p0;
    const double2 *zp = d.zp;
    double pc[1] = {0.0};
    for (int q = 0; q < d.EPT; q++) {
        const int j = blockIdx.x * (T * d.EPT) + q * T + threadIdx.x;
        double c = 0.0;
        if (j < d.n) {
            double pj, Mp;
            if (first) { pj = p0[j]; Mp = a_row(d, j, d.tmval, [p0](int c2) { return p0[c2]; }); }
            else {
                pj = d.z[j] + beta * pold[j];                                     // p = z + beta p (:464)
                pnew[j] = pj;
                Mp = a_row(d, j, d.tmval, [zp, beta](int c2) { const double2 v = zp[c2]; return v.x + beta * v.y; });
            }
            if (d.eq) Mp += col_dot(d.Cc, d.Cc_sv, j, d.qC);
            if (d.ineq) Mp += col_dot(d.Ec, d.Ec_sv, j, d.qE);
            d.tmp[j] = Mp;
            c = pj * Mp;
        }
        pc[0] = pc[0] + c;
    }
    store_partials<1>(d, 0, pc, red, parity);
    forward_state(d, in, out);
}

__global__ void __launch_bounds__(T) gen_k_pcg_upd(GenDev d, int in, int out) {     // :448-462
    __shared__ double red[2 * RED_MAXV * RED_MAXW];
    int parity = 0;
    const GenState *si = d.st + in;
    if (si->halt || si->phase != 2 || si->pcg_done) { forward_state(d, in, out); return; }
    const int k = si->pcg_k;
    const double alpha = si->absNew / d.red[0];
    const double *p = (k & 1) ? d.p1 : d.p0;
    double pd2[2] = {0.0, 0.0};
    for (int q = 0; q < d.EPT; q++) {
        const int j = blockIdx.x * (T * d.EPT) + q * T + threadIdx.x;
        double a = 0.0, b2 = 0.0;
        if (j < d.n) {
            double x = d.xt[j], r = d.r[j];
            const double pj = p[j];
            x += alpha * pj;
            r -= alpha * d.tmp[j];
            const double z = d.dinv[j] * r;
            d.xt[j] = x; d.r[j] = r; d.z[j] = z;
            d.zp[j] = make_double2(z, pj);
            a = r * r; b2 = r * z;
        }
        pd2[0] = pd2[0] + a; pd2[1] = pd2[1] + b2;
    }
    store_partials<2>(d, 0, pd2, red, parity);
    if (LEADER) { d.st[out] = *si; d.st[out].pcg_k = k + 1; }
}

// after the PCG: commit x, duals z1 z2 (:1733-1734), the seven partials (:1742-1793; A x and A round(x) gathered from the final iterate)
__global__ void __launch_bounds__(T) gen_k_post(GenDev d, int in, int out) {
    __shared__ double red[2 * RED_MAXV * RED_MAXW];
    int parity = 0;
    const GenState *si = d.st + in;
    if (si->halt || si->phase != 2) { forward_state(d, in, out); return; }
    const GenParams &P = d.prm;
    const int k = si->pcg_k;
    if (!si->pcg_done) {                      // the exit test of the last update is still pending
        if (!(k >= 1 && (d.red[0] < si->threshold || k >= P.pcg_maxiters))) {
            if (LEADER) { d.st[out] = *si; d.st[out].halt = GEN_HALT_PCG_MORE; }
            return;
        }
    }
    const double g1 = si->gamma_val * si->rho1, g2 = si->gamma_val * si->rho2;
    const double *xt = d.xt;
    double e5[5] = {0.0, 0.0, 0.0, 0.0, 0.0}, e2[2] = {0.0, 0.0};
    for (int q = 0; q < d.EPT; q++) {
        const int j = blockIdx.x * (T * d.EPT) + q * T + threadIdx.x;
        double v0 = 0.0, v1 = 0.0, v2 = 0.0, v3 = 0.0, v4 = 0.0, v5 = 0.0, v6 = 0.0;
        if (j < d.n) {
            const double x = xt[j], y1 = d.y1[j], y2 = d.y2[j], b = d.b[j];
            d.x[j] = x;
            d.z1[j] = d.z1[j] + g1 * (x - y1);
            d.z2[j] = d.z2[j] + g2 * (x - y2);
            d.gsrc[j] = x;
            const double ax = a_row(d, j, d.aval, [xt](int c) { return xt[c]; });                       // compute_cost (:560-572)
            const double axb = a_row(d, j, d.aval, [xt](int c) { return xt[c] >= 0.5 ? 1.0 : 0.0; });
            const double d1 = x - y1, d2 = x - y2, xb = x >= 0.5 ? 1.0 : 0.0;
            v0 = x * x; v1 = d1 * d1; v2 = d2 * d2; v3 = x * ax; v4 = b * x; v5 = xb * axb; v6 = b * xb;
        }
        e5[0] = e5[0] + v0; e5[1] = e5[1] + v1; e5[2] = e5[2] + v2; e5[3] = e5[3] + v3; e5[4] = e5[4] + v4;
        e2[0] = e2[0] + v5; e2[1] = e2[1] + v6;
    }
    store_partials<5>(d, 0, e5, red, parity);
    store_partials<2>(d, 5, e2, red, parity);
    if (LEADER) {
        d.st[out] = *si;
        GenState *s = d.st + out;
        s->pcg_done = 1; s->last_pcg = k; s->pcg_total += k; s->outer_total++;
        if (k > s->pcg_max) s->pcg_max = k;
        s->phase = 3;
    }
}

// z3 += gamma rho3 (C x - d) (:1736), z4 += gamma rho4 (E x + y3 - f) (:1739); Ex = E x for the next y3
__global__ void __launch_bounds__(T) gen_k_dual(GenDev d, int in, int out, int init_only) {
    const GenState *si = d.st + in;
    if (init_only) {
        if (d.ineq && blockIdx.x < d.Gl)
            for (int s = 0; s < d.EPTl; s++) { const int i = blockIdx.x * (T * d.EPTl) + s * T + threadIdx.x; if (i < d.l) d.Ex[i] = d.qE[i]; }
        forward_state(d, in, out);
        return;
    }
    if (si->halt || si->phase != 3) { forward_state(d, in, out); return; }
    const double g3 = si->gamma_val * si->rho3, g4 = si->gamma_val * si->rho4;
    if (d.eq && blockIdx.x < d.Gm)
        for (int s = 0; s < d.EPTm; s++) {
            const int i = blockIdx.x * (T * d.EPTm) + s * T + threadIdx.x;
            if (i < d.m) d.z3[i] = d.z3[i] + g3 * (d.qC[i] - d.d[i]);
        }
    if (d.ineq && blockIdx.x < d.Gl)
        for (int s = 0; s < d.EPTl; s++) {
            const int i = blockIdx.x * (T * d.EPTl) + s * T + threadIdx.x;
            if (i >= d.l) continue;
            const double Ex = d.qE[i];
            d.Ex[i] = Ex;
            d.z4[i] = d.z4[i] + g4 * ((Ex + d.y3[i]) - d.f[i]);
        }
    if (LEADER) { d.st[out] = *si; d.st[out].have_prev = 1; d.st[out].phase = 0; }
}

}  // namespace

// ---------------------------------------------------------------------------------------------------------------------
size_t gen_state_bytes() { return sizeof(GenState); }

static inline int gmax3(int a, int b, int c) { int m = a > b ? a : b; return m > c ? m : c; }

#define GEN_LAUNCH(kernel, grid, ...)                                                                     \
    do {                                                                                                  \
        hipLaunchKernelGGL(kernel, dim3(grid), dim3(T), 0, s, d, *parity, *parity ^ 1, ##__VA_ARGS__);    \
        *parity ^= 1;                                                                                     \
    } while (0)

hipError_t gen_launch_init(const GenDev &d, double c1, const double *x0, hipStream_t s) {
    hipLaunchKernelGGL(gen_k_init, dim3(gmax3(d.G, d.Gm, d.Gl)), dim3(T), 0, s, d, c1, x0);
    hipLaunchKernelGGL(gen_k_fin, dim3(1), dim3(T), 0, s, d, 2);
    return hipGetLastError();
}
hipError_t gen_launch_init2(const GenDev &d, hipStream_t s) {
    hipLaunchKernelGGL(gen_k_init2, dim3(1), dim3(1), 0, s, d);
    return hipGetLastError();
}
hipError_t gen_launch_fin(const GenDev &d, int nv, hipStream_t s) {
    hipLaunchKernelGGL(gen_k_fin, dim3(1), dim3(T), 0, s, d, nv);
    return hipGetLastError();
}
hipError_t gen_launch_resume(const GenDev &d, int reset_pcg_max, int *parity, hipStream_t s) {
    hipLaunchKernelGGL(gen_k_resume, dim3(1), dim3(1), 0, s, d, *parity, *parity ^ 1, reset_pcg_max);
    *parity ^= 1;
    return hipGetLastError();
}
hipError_t gen_launch_prep(const GenDev &d, int do_prep, int *parity, hipStream_t s) { GEN_LAUNCH(gen_k_prep, d.G, do_prep); return hipGetLastError(); }
hipError_t gen_launch_y(const GenDev &d, int *parity, hipStream_t s) { GEN_LAUNCH(gen_k_y, gmax3(d.G, d.Gl, 1)); return hipGetLastError(); }
hipError_t gen_launch_rhs_cols(const GenDev &d, int *parity, hipStream_t s) { GEN_LAUNCH(gen_k_rhs_cols, d.G); return hipGetLastError(); }
hipError_t gen_launch_rows(const GenDev &d, int mode, int *parity, hipStream_t s) { GEN_LAUNCH(gen_k_rows, gmax3(d.Gm, d.Gl, 1), mode); return hipGetLastError(); }
hipError_t gen_launch_resid(const GenDev &d, int *parity, hipStream_t s) { GEN_LAUNCH(gen_k_resid, d.G); return hipGetLastError(); }
hipError_t gen_launch_pcg_cols(const GenDev &d, int *parity, hipStream_t s) { GEN_LAUNCH(gen_k_pcg_cols, d.G); return hipGetLastError(); }
hipError_t gen_launch_pcg_upd(const GenDev &d, int *parity, hipStream_t s) { GEN_LAUNCH(gen_k_pcg_upd, d.G); return hipGetLastError(); }
hipError_t gen_launch_post(const GenDev &d, int *parity, hipStream_t s) { GEN_LAUNCH(gen_k_post, d.G); return hipGetLastError(); }
hipError_t gen_launch_dual(const GenDev &d, int init_only, int *parity, hipStream_t s) { GEN_LAUNCH(gen_k_dual, gmax3(d.Gm, d.Gl, 1), init_only); return hipGetLastError(); }
