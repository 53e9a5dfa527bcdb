// lpbox_lp_kernels.hip -- hand-written gfx950 kernels for the LP flavour of the Lp-Box ADMM inner solver.
//
// Design (MI355X-first):
//   * one workgroup == one LP instance, persistent over a whole window of ADMM iterations (one launch runs
//     iterations [iter_start, iter_end) of every instance of the batch; 256 instances fill the 256 CUs);
//   * every n- and l-vector of the algorithm lives in REGISTERS of the thread that owns the element
//     (element pos -> thread pos % T, slot pos / T); only the three vectors that other threads must gather
//     (p / x for E*v, E*v and f-y3 / z4 for E^T*w) are staged in LDS, together with the uint16 CSR+CSC
//     index sets of E.  HBM is touched once per launch (state in / state out) plus the x_iters window;
//   * all dot products / norms use one fixed reduction tree: per-thread slot sums -> 64-lane xor butterfly
//     on DPP + v_permlane{16,32}_swap -> wave partials through LDS, added in wave order.  The CPU oracle
//     reproduces exactly this association (oracle/lpbox_oracle.c, LPO_ORDER_GPU), which makes the kernel
//     bit-comparable with it; sparse row sums run in ascending index order like Eigen's CSC product;
//   * early fixing is a mask: a fixed variable keeps its register slot, is excluded from every gather and
//     reduction by predication (it contributes +0.0) and the x-update commits `live ? x_new : x_fixed`.
//
// Reference citations: LPcpp = LinerProgramming/LinearProgramming/cython_solver/LPboxADMMsolver.cpp.
// Built with -ffp-contract=off: the reference is compiled without FMA (plain g++ -O3 on x86-64).
#include "lpbox_lp.h"

#include <float.h>

namespace {

// ------------------------------------------------------------------------------------------------
// wave / block reductions
// ------------------------------------------------------------------------------------------------
template <int CTRL>
__device__ __forceinline__ double dpp_mov(double v) {
    int lo = __double2loint(v), hi = __double2hiint(v);
    lo = __builtin_amdgcn_mov_dpp(lo, CTRL, 0xf, 0xf, true);
    hi = __builtin_amdgcn_mov_dpp(hi, CTRL, 0xf, 0xf, true);
    return __hiloint2double(hi, lo);
}

// 64-lane all-reduce, association = balanced binary tree over the lane index (pairs, quads, ..., halves).
// Steps 3/4 use row_half_mirror / row_mirror: after the quad steps every lane of a quad (8-group) holds the same
// partial, so the mirrored partner carries exactly the xor-4 (xor-8) partner's value.
__device__ __forceinline__ double wave_allreduce_sum(double v) {
#ifdef LPBOX_REDUCE_SHFL
    for (int off = 1; off < 64; off <<= 1) v = v + __shfl_xor(v, off, 64);
    return v;
#else
    v = v + dpp_mov<0xB1>(v);    // quad_perm [1,0,3,2]
    v = v + dpp_mov<0x4E>(v);    // quad_perm [2,3,0,1]
    v = v + dpp_mov<0x141>(v);   // row_half_mirror
    v = v + dpp_mov<0x140>(v);   // row_mirror
    {
        int lo = __double2loint(v), hi = __double2hiint(v);
        auto a = __builtin_amdgcn_permlane16_swap(lo, lo, false, false);
        auto b = __builtin_amdgcn_permlane16_swap(hi, hi, false, false);
        // a[0]/b[0]: rows {0,0,2,2}; a[1]/b[1]: rows {1,1,3,3} of the input -> even-row + odd-row partial
        v = __hiloint2double(b[0], a[0]) + __hiloint2double(b[1], a[1]);
    }
    {
        int lo = __double2loint(v), hi = __double2hiint(v);
        auto a = __builtin_amdgcn_permlane32_swap(lo, lo, false, false);
        auto b = __builtin_amdgcn_permlane32_swap(hi, hi, false, false);
        v = __hiloint2double(b[0], a[0]) + __hiloint2double(b[1], a[1]);   // lower half + upper half
    }
    return v;
#endif
}

constexpr int RED_MAXV = 5;    // values reduced together
constexpr int RED_MAXW = 16;   // waves per workgroup

// Sum NV per-thread partials over the workgroup; every thread receives the totals.  `red` is a ping-pong LDS
// scratch (2 * RED_MAXV * RED_MAXW doubles): a thread can run at most one block_sum ahead of the slowest one.
template <int T, int NV>
__device__ __forceinline__ void block_sum(double (&v)[NV], double *red, int &parity) {
    constexpr int W = T / 64;
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    double *buf = red + parity * (RED_MAXV * RED_MAXW);
#pragma unroll
    for (int k = 0; k < NV; k++) v[k] = wave_allreduce_sum(v[k]);
    if (lane == 0) {
#pragma unroll
        for (int k = 0; k < NV; k++) buf[k * RED_MAXW + w] = v[k];
    }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < NV; k++) {
        double t = buf[k * RED_MAXW];
#pragma unroll
        for (int w2 = 1; w2 < W; w2++) t = t + buf[k * RED_MAXW + w2];
        v[k] = t;
    }
    parity ^= 1;
}

// ------------------------------------------------------------------------------------------------
// LDS carve-up shared by the launcher (size) and the kernel (pointers)
// ------------------------------------------------------------------------------------------------
struct LdsLayout {
    size_t gx, gl0, gl1, gl2, red, csr_ptr, csc_ptr, csr_col, csc_row, total;
    __host__ __device__ LdsLayout(int NS, int LS, int ZS) {
        size_t o = 0;
        auto take = [&](size_t bytes) { size_t at = o; o += (bytes + 15) & ~size_t(15); return at; };
        gx = take(sizeof(double) * (size_t)NS);
        gl0 = take(sizeof(double) * (size_t)LS);
        gl1 = take(sizeof(double) * (size_t)LS);
        gl2 = take(sizeof(double) * (size_t)LS);
        red = take(sizeof(double) * 2 * RED_MAXV * RED_MAXW);
        csr_ptr = take(sizeof(int) * ((size_t)LS + 1));
        csc_ptr = take(sizeof(int) * ((size_t)NS + 1));
        csr_col = take(sizeof(uint16_t) * (size_t)ZS);
        csc_row = take(sizeof(uint16_t) * (size_t)ZS);
        total = o;
    }
};

// y = E*v restricted to this thread's rows: res_i = sum_j E_ij * v_j with j ascending (Eigen CSC product order,
// LPcpp:102-108), all stored values are 1.0 so the product term is v_j itself.
template <int T, int EPT>
__device__ __forceinline__ void rows_gather(const int *s_ptr, const uint16_t *s_col, const double *gx, int l,
                                            double (&out)[EPT]) {
#pragma unroll
    for (int s = 0; s < EPT; s++) {
        const int i = s * T + (int)threadIdx.x;
        double acc = 0.0;
        if (i < l) {
            const int k1 = s_ptr[i + 1];
            for (int k = s_ptr[i]; k < k1; k++) acc += gx[s_col[k]];
        }
        out[s] = acc;
    }
}

// ------------------------------------------------------------------------------------------------
// ADMM_lp_iters_init (LPcpp:489-763) for every instance: x=1, z=0, rho=25, ...
// ------------------------------------------------------------------------------------------------
template <int T, int EPT>
__global__ void __launch_bounds__(T) lp_init_kernel(LpBatchDev bd, const double *f_org, const double *c1_init) {
    __shared__ double red[2 * RED_MAXV * RED_MAXW];
    const int inst = blockIdx.x, tid = threadIdx.x;
    int *isc = bd.isc + (size_t)inst * NI_COUNT;
    double *dsc = bd.dsc + (size_t)inst * ND_COUNT;
    const int n = isc[NI_N], l = isc[NI_L];
    const size_t on = (size_t)inst * bd.NS, ol = (size_t)inst * bd.LS;
    int parity = 0;
    double part[1] = {0.0};
#pragma unroll
    for (int s = 0; s < EPT; s++) {
        const int pos = s * T + tid;
        double prod = 0.0;
        if (pos < n) {
            bd.x[on + pos] = 1.0;                    // :583-586
            bd.z1[on + pos] = 0.0; bd.z2[on + pos] = 0.0;   // :616-617
            bd.pd[on + pos] = 0.0;
            bd.live[on + pos] = 1;
            prod = bd.b[on + pos] * 1.0;             // best_bin_obj = b.dot(x0), :727
        }
        part[0] = part[0] + prod;
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
        isc[NI_NLIVE] = n;
        isc[NI_RHO_UPDATED] = 1;                     // LPh:214
        isc[NI_ITER] = 0; isc[NI_HIST_N] = 0; isc[NI_RET] = 0; isc[NI_STOP] = 0;
        isc[NI_PCG_TOTAL] = 0; isc[NI_OUTER_TOTAL] = 0; isc[NI_LAST_PCG] = 0; isc[NI_PLAIN_ITER_P1] = 0;
        isc[NI_EXPR_READY] = 0;
        for (int k = 0; k < LP_HIST; k++) bd.hist[(size_t)inst * LP_HIST + k] = 0.0;
    }
}

// ------------------------------------------------------------------------------------------------
// The ADMM window: iterations [iter_start, iter_end) of ADMM_lp_iters (LPcpp:766-1095, l2f == 0) or
// ADMM_lp_iters_l2f (LPcpp:1098-1574, l2f == 1) for one instance per workgroup.
// ------------------------------------------------------------------------------------------------
template <int T, int EPT>
__global__ void __launch_bounds__(T) lp_window_kernel(LpBatchDev bd, int iter_start, int iter_end, int l2f) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int inst = blockIdx.x, tid = threadIdx.x;
    int *isc = bd.isc + (size_t)inst * NI_COUNT;
    double *dsc = bd.dsc + (size_t)inst * ND_COUNT;
    if (!isc[NI_ACTIVE]) return;

    const int n = isc[NI_N], l = isc[NI_L], nnz = isc[NI_NNZ];
    const size_t on = (size_t)inst * bd.NS, ol = (size_t)inst * bd.LS, oz = (size_t)inst * bd.ZS;

    const LdsLayout L(bd.NS, bd.LS, bd.ZS);
    double *gx = (double *)(smem + L.gx);
    double *gl0 = (double *)(smem + L.gl0);
    double *gl1 = (double *)(smem + L.gl1);
    double *gl2 = (double *)(smem + L.gl2);
    double *red = (double *)(smem + L.red);
    int *s_csr_ptr = (int *)(smem + L.csr_ptr);
    int *s_csc_ptr = (int *)(smem + L.csc_ptr);
    uint16_t *s_csr_col = (uint16_t *)(smem + L.csr_col);
    uint16_t *s_csc_row = (uint16_t *)(smem + L.csc_row);

    // ---- stage the index sets of E into LDS ----
    {
        const int *gp = bd.csr_ptr + (size_t)inst * (bd.LS + 1);
        for (int i = tid; i <= l; i += T) s_csr_ptr[i] = gp[i];
        const int *gc = bd.csc_ptr + (size_t)inst * (bd.NS + 1);
        for (int i = tid; i <= n; i += T) s_csc_ptr[i] = gc[i];
        const uint16_t *c0 = bd.csr_col + oz, *r0 = bd.csc_row + oz;
        for (int k = tid; k < nnz; k += T) { s_csr_col[k] = c0[k]; s_csc_row[k] = r0[k]; }
    }

    // ---- per-thread state ----
    double x[EPT], z1[EPT], z2[EPT], b[EPT], pd[EPT], dinv[EPT], Esq[EPT];
    bool live[EPT], valid[EPT];
    double z4[EPT], f[EPT], Ex[EPT], y3[EPT];
    bool rvalid[EPT];
#pragma unroll
    for (int s = 0; s < EPT; s++) {
        const int pos = s * T + tid;
        valid[s] = pos < n;
        x[s] = z1[s] = z2[s] = b[s] = pd[s] = 0.0;
        live[s] = false;
        if (valid[s]) {
            x[s] = bd.x[on + pos]; z1[s] = bd.z1[on + pos]; z2[s] = bd.z2[on + pos];
            b[s] = bd.b[on + pos]; pd[s] = bd.pd[on + pos];
            live[s] = bd.live[on + pos] != 0;
        }
        rvalid[s] = pos < l;
        z4[s] = f[s] = 0.0;
        if (rvalid[s]) { z4[s] = bd.z4[ol + pos]; f[s] = bd.f[ol + pos]; }
        Ex[s] = y3[s] = 0.0;
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
    double h[LP_HIST];
#pragma unroll
    for (int k = 0; k < LP_HIST; k++) h[k] = bd.hist[(size_t)inst * LP_HIST + k];
    const double learning_fact = LP_LEARNING_FACT;
    int ret = 0, stop = LP_STOP_NONE, parity = 0;

    __syncthreads();   // index sets staged
#pragma unroll
    for (int s = 0; s < EPT; s++) {   // Esq_diag_j = sum of squared entries of column j (LPcpp:2378-2390); entries are 1.0
        double e = 0.0;
        if (valid[s]) { const int pos = s * T + tid; e = (double)(s_csc_ptr[pos + 1] - s_csc_ptr[pos]); }   // = 1.0*1.0 added c times
        Esq[s] = e;
    }

    // ---- early fixing: apply this call's fix vector (LPcpp:1124-1335) as a mask ----
    bool finished = false;
    if (bd.ctl[(size_t)inst * 4 + 0]) {
        const int n_live_new = bd.ctl[(size_t)inst * 4 + 2];
        double part[1] = {0.0};
        uint8_t nf[EPT];
#pragma unroll
        for (int s = 0; s < EPT; s++) {
            const int pos = s * T + tid;
            nf[s] = valid[s] ? bd.newfix[on + pos] : 0;
            const double val = nf[s] == 2 ? 1.0 : 0.0;
            part[0] = part[0] + (nf[s] ? b[s] * val : 0.0);     // fix_obj = b2.dot(x2), :1237
            if (valid[s]) gx[pos] = nf[s] ? val : 0.0;
        }
        block_sum<T, 1>(part, red, parity);                      // (barrier inside also publishes gx)
        fix_obj = part[0];
        if (n_live_new == 0) {                                   // :1212-1217 (nothing else is updated)
            ret = 1; stop = LP_STOP_ALLFIXED; n_live = 0; finished = true;
#pragma unroll
            for (int s = 0; s < EPT; s++) if (nf[s]) { live[s] = false; x[s] = nf[s] == 2 ? 1.0 : 0.0; }
        } else {
            double cnt[EPT];
            rows_gather<T, EPT>(s_csr_ptr, s_csr_col, gx, l, cnt);   // E2*x2, :1276
#pragma unroll
            for (int s = 0; s < EPT; s++) {
                f[s] = f[s] - cnt[s];                                 // f1 = f - E2*x2, :1278
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
            for (int s = 0; s < EPT; s++) { pd[s] = dI; pd[s] += rho4 * Esq[s]; }
            r4Et = rho4;
            expr_ready = 1;
        }
    }

    int it = iter_start;
    if (!finished) {
        // DiagonalPreconditioner state (LPcpp:883-890): always 1/pd as of the last compute; recomputed here so that a fix
        // applied while no rho update is pending (stale preconditioner = UB in the reference) is well defined.
#pragma unroll
        for (int s = 0; s < EPT; s++) dinv[s] = (pd[s] != 0.0) ? 1.0 / pd[s] : 1.0;
        // E*x for the first iteration's y3
        __syncthreads();
#pragma unroll
        for (int s = 0; s < EPT; s++) if (valid[s]) gx[s * T + tid] = live[s] ? x[s] : 0.0;
        __syncthreads();
        rows_gather<T, EPT>(s_csr_ptr, s_csr_col, gx, l, Ex);

        int cc = 0;
        for (; it < iter_end; ++it) {
            // ---------------- y1 (box) and y2 (shifted L2 sphere), LPcpp:806-818 ----------------
            double y1[EPT], y2[EPT];
            double pn[1] = {0.0};
#pragma unroll
            for (int s = 0; s < EPT; s++) {
                const double t = x[s] + z1[s] / rho1;
                y1[s] = t > 1 ? 1 : (t < 0 ? 0 : t);                  // project_box :409-421
                const double u = (x[s] + z2[s] / rho2) - 0.5;         // project_shifted_Lp_ball :423-428
                y2[s] = u;
                pn[0] = pn[0] + (live[s] ? u * u : 0.0);
            }
            block_sum<T, 1>(pn, red, parity);
            {
                const double c2 = 2 * sqrt(pn[0]);
#pragma unroll
                for (int s = 0; s < EPT; s++) y2[s] = y2[s] * c1 / c2 + 0.5;
            }
            // ---------------- y3 = max(0, f - E x - z4/rho4), LPcpp:824-828 ----------------
#pragma unroll
            for (int s = 0; s < EPT; s++) {
                const double v = f[s] - Ex[s] - z4[s] / rho4;
                y3[s] = v < 0 ? 0 : v;
                if (rvalid[s]) { gl0[s * T + tid] = f[s] - y3[s]; gl1[s * T + tid] = z4[s]; }
                if (valid[s]) gx[s * T + tid] = live[s] ? y1[s] : 0.0;      // PCG start x0 = y1 (:892)
            }
            __syncthreads();
            // ---------------- matrix-expression refresh, LPcpp:831-866 ----------------
            if (it == 0) {                                            // update_expression(0)
                dI = 0.0; dI += rho1 + rho2;
#pragma unroll
                for (int s = 0; s < EPT; s++) { pd[s] = dI; pd[s] += rho4 * Esq[s]; }
                r4Et = rho4;
                expr_ready = 1;
            }
            if (it != 0 && rhoUpdated) {
                const double inc = rcr * (prev_rho1 + prev_rho2);
                const double inc4 = rcr * prev_rho4;
                dI += inc;
#pragma unroll
                for (int s = 0; s < EPT; s++) { pd[s] += inc; pd[s] += inc4 * Esq[s]; }
                r4Et = learning_fact * r4Et;                          // rho4_E_transpose *= learning_fact (:864)
            }
            // ---------------- rhs (:872-878) and q = E*y1 ----------------
            double rhs[EPT];
#pragma unroll
            for (int s = 0; s < EPT; s++) {
                double tA = 0.0, tB = 0.0;
                if (valid[s]) {
                    const int pos = s * T + tid;
                    const int k1 = s_csc_ptr[pos + 1];
                    for (int k = s_csc_ptr[pos]; k < k1; k++) {
                        const int i = s_csc_row[k];
                        tA += r4Et * gl0[i];                          // (rho4 E^T)(f - y3)
                        tB += gl1[i];                                 // E^T z4
                    }
                }
                double r_ = (rho1 * y1[s] + rho2 * y2[s]) - ((b[s] + z1[s]) + z2[s]);
                r_ += tA;
                r_ -= tB;
                rhs[s] = r_;
            }
            {
                double q[EPT];
                rows_gather<T, EPT>(s_csr_ptr, s_csr_col, gx, l, q);
#pragma unroll
                for (int s = 0; s < EPT; s++) if (rvalid[s]) gl2[s * T + tid] = q[s];
            }
            __syncthreads();
            if (rhoUpdated) {                                         // DiagonalPreconditioner::compute (:883-890)
#pragma unroll
                for (int s = 0; s < EPT; s++) dinv[s] = (pd[s] != 0.0) ? 1.0 / pd[s] : 1.0;
                rhoUpdated = 0;
            }
            // ---------------- PCG (LPcpp:251-335) on (dI*I + rho4 E^T E) x = rhs ----------------
            double xt[EPT], r[EPT], p[EPT];
            double p3[3] = {0.0, 0.0, 0.0};
#pragma unroll
            for (int s = 0; s < EPT; s++) {
                double t = 0.0;
                if (valid[s]) {
                    const int pos = s * T + tid;
                    const int k1 = s_csc_ptr[pos + 1];
                    for (int k = s_csc_ptr[pos]; k < k1; k++) t += r4Et * gl2[s_csc_row[k]];
                }
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
            const double rhsNorm2 = p3[0];
            double residualNorm2 = p3[1], absNew = p3[2];
            int k_it = 0;
            bool pcg_fail = false;
            if (rhsNorm2 == 0) {                                      // :273-278
#pragma unroll
                for (int s = 0; s < EPT; s++) xt[s] = 0.0;
            } else {
                double threshold = LP_PCG_TOL * LP_PCG_TOL * rhsNorm2;    // :281
                if (threshold < DBL_MIN) threshold = DBL_MIN;
                if (!(residualNorm2 < threshold)) {                   // :284
                    while (k_it < LP_PCG_MAXITERS) {                  // :296
#pragma unroll
                        for (int s = 0; s < EPT; s++) if (valid[s]) gx[s * T + tid] = live[s] ? p[s] : 0.0;
                        __syncthreads();
                        {
                            double q[EPT];
                            rows_gather<T, EPT>(s_csr_ptr, s_csr_col, gx, l, q);
#pragma unroll
                            for (int s = 0; s < EPT; s++) if (rvalid[s]) gl0[s * T + tid] = q[s];
                        }
                        __syncthreads();
                        double tmp[EPT];
                        double p1[1] = {0.0};
#pragma unroll
                        for (int s = 0; s < EPT; s++) {               // tmp = M p (:298), fused p.tmp
                            double t = 0.0;
                            if (valid[s]) {
                                const int pos = s * T + tid;
                                const int k1 = s_csc_ptr[pos + 1];
                                for (int k = s_csc_ptr[pos]; k < k1; k++) t += r4Et * gl0[s_csc_row[k]];
                            }
                            double Mp = 0.0;
                            Mp += dI * (1.0 * p[s]);
                            Mp += t;
                            tmp[s] = Mp;
                            p1[0] = p1[0] + (live[s] ? p[s] * tmp[s] : 0.0);
                        }
                        block_sum<T, 1>(p1, red, parity);
                        const double alpha = absNew / p1[0];          // :300
                        if (alpha < 0) { pcg_fail = true; break; }    // :301
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
                        block_sum<T, 2>(p2, red, parity);
                        residualNorm2 = p2[0];
                        if (residualNorm2 < threshold) { k_it++; break; }     // :309-312
                        const double absOld = absNew;
                        absNew = p2[1];
                        const double beta = absNew / absOld;          // :318
#pragma unroll
                        for (int s = 0; s < EPT; s++) p[s] = z[s] + beta * p[s];   // :319
                        k_it++;
                    }
                }
            }
            last_pcg = k_it;
            pcg_total += k_it;
            if (pcg_fail) stop = LP_STOP_PCG;
            if (pcg_fail && l2f) { ret = 1; break; }                  // :1450-1454: return 1, x_sol untouched
            // ---------------- commit x: branchless predicated write (fixed variables keep their value) ----------------
#pragma unroll
            for (int s = 0; s < EPT; s++) x[s] = live[s] ? xt[s] : x[s];
            outer_total++;
            if (l2f) {                                                // x_iters column cc (:1472-1475)
                double *xh = bd.xhist + ((size_t)inst * bd.ws_cap + cc) * bd.NS;
#pragma unroll
                for (int s = 0; s < EPT; s++) if (valid[s]) xh[s * T + tid] = x[s];
                cc++;
            }
            // ---------------- duals (:917-924 / :1487-1491) ----------------
            const double g1 = gamma_val * rho1, g2 = gamma_val * rho2, g4 = gamma_val * rho4;
#pragma unroll
            for (int s = 0; s < EPT; s++) {
                z1[s] = z1[s] + g1 * (x[s] - y1[s]);
                z2[s] = z2[s] + g2 * (x[s] - y2[s]);
                if (valid[s]) gx[s * T + tid] = live[s] ? x[s] : 0.0;
            }
            __syncthreads();
            rows_gather<T, EPT>(s_csr_ptr, s_csr_col, gx, l, Ex);     // E*x: feeds z4 now and y3 of the next iteration
#pragma unroll
            for (int s = 0; s < EPT; s++) {
                const double d = g4 * ((Ex[s] + y3[s]) - f[s]);
                z4[s] = (!l2f && it == iter_start) ? d : z4[s] + d;   // :920-923 (plain loop overwrites on its first iteration)
            }
            // ---------------- residual norms, objective (:931-1011) ----------------
            double p5[5] = {0.0, 0.0, 0.0, 0.0, 0.0};
#pragma unroll
            for (int s = 0; s < EPT; s++) {
                const double d1 = x[s] - y1[s], d2 = x[s] - y2[s];
                const double xb = x[s] >= 0.5 ? 1.0 : 0.0;
                p5[0] = p5[0] + (live[s] ? x[s] * x[s] : 0.0);
                p5[1] = p5[1] + (live[s] ? d1 * d1 : 0.0);
                p5[2] = p5[2] + (live[s] ? d2 * d2 : 0.0);
                p5[3] = p5[3] + (live[s] ? b[s] * x[s] : 0.0);
                p5[4] = p5[4] + (live[s] ? b[s] * xb : 0.0);
            }
            block_sum<T, 5>(p5, red, parity);
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
            if ((it + 1) % LP_RHO_STEP == 0) {                        // :951-970
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
            if (hist_n < LP_HIST) {
#pragma unroll
                for (int k = 0; k < LP_HIST; k++) if (k == hist_n) h[k] = obj_val;
            } else {
#pragma unroll
                for (int k = 0; k < LP_HIST - 1; k++) h[k] = h[k + 1];
                h[LP_HIST - 1] = obj_val;
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
        }
    }

    // ---- write the state back ----
#pragma unroll
    for (int s = 0; s < EPT; s++) {
        const int pos = s * T + tid;
        if (valid[s]) {
            bd.x[on + pos] = x[s]; bd.z1[on + pos] = z1[s]; bd.z2[on + pos] = z2[s]; bd.pd[on + pos] = pd[s];
            bd.live[on + pos] = live[s] ? 1 : 0;
        }
        if (rvalid[s]) { bd.z4[ol + pos] = z4[s]; bd.f[ol + pos] = f[s]; }
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
        if (l2f) isc[NI_ITER] = it;                                   // member `iter` (LPh:279), advanced by l2f only
        else isc[NI_PLAIN_ITER_P1] = it + 1;                          // LPcpp:1081
#pragma unroll
        for (int k = 0; k < LP_HIST; k++) bd.hist[(size_t)inst * LP_HIST + k] = h[k];
    }
}

// x_iters export (get_x_iters_d, LPcpp:1616-1627): out[inst][r*ws + c] = x after iteration c of live variable r.
__global__ void lp_pack_xiters_kernel(LpBatchDev bd, const int *left_idx, const int *rows, int ws, double *out,
                                      long out_stride) {
    const int inst = blockIdx.y;
    const int nr = rows[inst];
    const long total = (long)nr * ws;
    const double *xh = bd.xhist + (size_t)inst * bd.ws_cap * bd.NS;
    const int *li = left_idx + (size_t)inst * bd.NS;
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
size_t lp_window_lds_bytes(int T, int NS, int LS, int ZS) {
    (void)T;
    return LdsLayout(NS, LS, ZS).total;
}

#define LP_DISPATCH(KERNEL_CALL)                                                                   \
    if (T == 256 && EPT == 1) { KERNEL_CALL(256, 1) }                                              \
    else if (T == 256 && EPT == 2) { KERNEL_CALL(256, 2) }                                         \
    else if (T == 256 && EPT == 4) { KERNEL_CALL(256, 4) }                                         \
    else if (T == 512 && EPT == 1) { KERNEL_CALL(512, 1) }                                         \
    else if (T == 512 && EPT == 2) { KERNEL_CALL(512, 2) }                                         \
    else if (T == 512 && EPT == 4) { KERNEL_CALL(512, 4) }                                         \
    else if (T == 1024 && EPT == 1) { KERNEL_CALL(1024, 1) }                                       \
    else if (T == 1024 && EPT == 2) { KERNEL_CALL(1024, 2) }                                       \
    else if (T == 1024 && EPT == 4) { KERNEL_CALL(1024, 4) }                                       \
    else return hipErrorInvalidConfiguration;

hipError_t lp_launch_init(const LpBatchDev &bd, int T, int EPT, const double *f_org, const double *c1_init, hipStream_t s) {
#define CALL_INIT(TT, EE) hipLaunchKernelGGL((lp_init_kernel<TT, EE>), dim3(bd.B), dim3(TT), 0, s, bd, f_org, c1_init);
    LP_DISPATCH(CALL_INIT)
#undef CALL_INIT
    return hipGetLastError();
}

hipError_t lp_launch_window(const LpBatchDev &bd, int T, int EPT, size_t lds, int iter_start, int iter_end, int l2f,
                            hipStream_t s) {
#define CALL_WIN(TT, EE)                                                                                       \
    {                                                                                                          \
        hipError_t e = hipFuncSetAttribute((const void *)lp_window_kernel<TT, EE>,                             \
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);              \
        if (e != hipSuccess) return e;                                                                         \
        hipLaunchKernelGGL((lp_window_kernel<TT, EE>), dim3(bd.B), dim3(TT), lds, s, bd, iter_start, iter_end, l2f); \
    }
    LP_DISPATCH(CALL_WIN)
#undef CALL_WIN
    return hipGetLastError();
}

hipError_t lp_launch_pack_xiters(const LpBatchDev &bd, const int *left_idx, const int *rows, int ws, double *out,
                                 long out_stride, hipStream_t s) {
    dim3 grid(8, bd.B);
    hipLaunchKernelGGL(lp_pack_xiters_kernel, grid, dim3(256), 0, s, bd, left_idx, rows, ws, out, out_stride);
    return hipGetLastError();
}
