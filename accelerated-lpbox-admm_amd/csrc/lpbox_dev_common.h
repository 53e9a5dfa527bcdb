// lpbox_dev_common.h -- device helpers shared by the LP and SEG kernels: the fixed reduction tree.
#pragma once
#include <hip/hip_runtime.h>

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

// The fixed reduction tree of a 64-lane wavefront, per value: halves first (lane l + lane l^32), then the two row pairs (l^16),
// then inside a row of 16 lanes pairs, quads, 8, 16.  Putting the two wide steps FIRST lets several values share them: a
// v_permlane32_swap / v_permlane16_swap of two DIFFERENT values is a reduce-scatter step (one half / row keeps value A, the other
// value B), so NV values cost one narrow butterfly instead of NV (wave_reduce_scatter below).  Floating-point addition is
// commutative, so "own + partner" and "partner + own" are the same bits and every lane of a group holds the identical partial.
struct HalfSwap {           // r0 = [a.lanes 0-31 , b.lanes 0-31], r1 = [a.lanes 32-63, b.lanes 32-63]
    static __device__ __forceinline__ void run(double a, double b, double &r0, double &r1) {
        auto lo = __builtin_amdgcn_permlane32_swap(__double2loint(a), __double2loint(b), false, false);
        auto hi = __builtin_amdgcn_permlane32_swap(__double2hiint(a), __double2hiint(b), false, false);
        r0 = __hiloint2double(hi[0], lo[0]); r1 = __hiloint2double(hi[1], lo[1]);
    }
};
struct RowSwap {            // r0 = rows [a0, b0, a2, b2], r1 = rows [a1, b1, a3, b3] (rows of 16 lanes)
    static __device__ __forceinline__ void run(double a, double b, double &r0, double &r1) {
        auto lo = __builtin_amdgcn_permlane16_swap(__double2loint(a), __double2loint(b), false, false);
        auto hi = __builtin_amdgcn_permlane16_swap(__double2hiint(a), __double2hiint(b), false, false);
        r0 = __hiloint2double(hi[0], lo[0]); r1 = __hiloint2double(hi[1], lo[1]);
    }
};
// one wide step for two values: the lower half / even rows end up with a's pair sums, the upper half / odd rows with b's
template <typename SWAP>
__device__ __forceinline__ double pair_step(double a, double b) { double r0, r1; SWAP::run(a, b, r0, r1); return r0 + r1; }

__device__ __forceinline__ double row_allreduce(double v) {      // inside each row of 16 lanes: pairs, quads, 8, 16
    v = v + dpp_mov<0xB1>(v);    // quad_perm [1,0,3,2]
    v = v + dpp_mov<0x4E>(v);    // quad_perm [2,3,0,1]
    v = v + dpp_mov<0x141>(v);   // row_half_mirror: after the quad steps the mirrored partner holds the xor-4 partner's value
    v = v + dpp_mov<0x140>(v);   // row_mirror: likewise xor-8
    return v;
}

// 64-lane all-reduce of one value (every lane receives the sum).
__device__ __forceinline__ double wave_allreduce_sum(double v) {
#ifdef LPBOX_REDUCE_SHFL
    v = v + __shfl_xor(v, 32, 64); v = v + __shfl_xor(v, 16, 64);
    for (int off = 1; off < 16; off <<= 1) v = v + __shfl_xor(v, off, 64);
    return v;
#else
    v = pair_step<HalfSwap>(v, v);
    v = pair_step<RowSwap>(v, v);
    return row_allreduce(v);
#endif
}

// NV <= 4 values through ONE narrow butterfly: on return the total of value k sits in every lane of row ROW_OF[k]
// (rows of 16 lanes: value 0 -> row 0, 1 -> row 2, 2 -> row 1, 3 -> row 3); the result is returned in `out`, to be read
// only in those rows.  Same per-value tree as wave_allreduce_sum.
template <int NV>
__device__ __forceinline__ double wave_reduce_scatter(const double *v) {
    static_assert(NV >= 2 && NV <= 4, "2..4 values");
    const double x = pair_step<HalfSwap>(v[0], v[1]);                                   // lower half: value 0, upper half: value 1
    double s;
    if constexpr (NV == 2) s = pair_step<RowSwap>(x, x);
    else {
        const double y = NV == 4 ? pair_step<HalfSwap>(v[2], v[3]) : pair_step<HalfSwap>(v[2], v[2]);
        s = pair_step<RowSwap>(x, y);                                                   // rows: [v0, v2, v1, v3 (or v2 again)]
    }
    return row_allreduce(s);
}
// lane that owns value k after wave_reduce_scatter<NV>
__device__ __forceinline__ constexpr int scatter_lane(int nv, int k) { return nv == 2 ? 32 * k : (k == 0 ? 0 : k == 1 ? 32 : k == 2 ? 16 : 48); }

__device__ __forceinline__ int wave_max_int(int v) {
    for (int off = 1; off < 64; off <<= 1) { int o = __shfl_xor(v, off, 64); v = o > v ? o : v; }
    return __builtin_amdgcn_readfirstlane(v);
}

constexpr int RED_MAXV = 6;    // values reduced together
constexpr int RED_MAXW = 16;   // waves per workgroup

// Sum NV per-thread partials over the workgroup; every thread receives the totals.  Wave partials go through LDS and
// are combined by a second butterfly (lane i reads partial i mod W; balanced tree over the wave index), i.e. ONE LDS
// round trip instead of W dependent reads.  `red` is a ping-pong scratch (2 * RED_MAXV * RED_MAXW doubles): a thread
// can run at most one block_sum ahead of the slowest one.
template <int T, int NV>
__device__ __forceinline__ void block_sum(double (&v)[NV], double *red, int &parity) {
    constexpr int W = T / 64;
    static_assert(W == 1 || W == 2 || W == 4 || W == 8 || W == 16, "waves per workgroup");
    if constexpr (W == 1) {              // one wavefront per instance: no LDS round trip, no barrier
#pragma unroll
        for (int k = 0; k < NV; k++) v[k] = wave_allreduce_sum(v[k]);
        return;
    }
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    double *buf = red + parity * (RED_MAXV * RED_MAXW);
    if constexpr (NV == 1) {
        const double t0 = wave_allreduce_sum(v[0]);
        if (lane == 0) buf[w] = t0;
    } else {
        // values 0..3 share one butterfly (reduce-scatter); a fifth goes alone, a fifth and a sixth share a second one
        constexpr int NS4 = NV > 4 ? 4 : NV;
        const double sc = wave_reduce_scatter<NS4>(v);
#pragma unroll
        for (int k = 0; k < NS4; k++) if (lane == scatter_lane(NS4, k)) buf[k * RED_MAXW + w] = sc;
        if constexpr (NV == 5) {
            const double t4 = wave_allreduce_sum(v[4]);
            if (lane == 0) buf[4 * RED_MAXW + w] = t4;
        } else if constexpr (NV == 6) {          // values 4 and 5 share a second butterfly
            const double sc2 = wave_reduce_scatter<2>(v + 4);
#pragma unroll
            for (int k = 0; k < 2; k++) if (lane == scatter_lane(2, k)) buf[(4 + k) * RED_MAXW + w] = sc2;
        } else static_assert(NV <= 4, "at most six values");
    }
    __syncthreads();
    if constexpr (W >= 16) {
        // 16 waves (128-register budget): lane i takes the partial of wave i mod 16 and the row of 16 lanes runs the balanced tree
        // over the wave index on DPP -- the same association as the register version below, NV instead of NV * W live values
        double t[NV];
#pragma unroll
        for (int k = 0; k < NV; k++) t[k] = buf[k * RED_MAXW + (lane & (W - 1))];
#pragma unroll
        for (int k = 0; k < NV; k++) v[k] = row_allreduce(t[k]);
        parity ^= 1;
        return;
    }
#ifdef LPBOX_STAGE2_DPP
    double t[NV];
#pragma unroll
    for (int k = 0; k < NV; k++) t[k] = buf[k * RED_MAXW + (lane & (W - 1))];
#pragma unroll
    for (int k = 0; k < NV; k++) {
        double u = t[k];
        u = u + dpp_mov<0xB1>(u);                    // waves (0,1) (2,3) ...
        if (W >= 4) u = u + dpp_mov<0x4E>(u);        // quads of waves
        if (W >= 8) u = u + dpp_mov<0x141>(u);       // 8 waves
        if (W >= 16) u = u + dpp_mov<0x140>(u);      // 16 waves
        v[k] = u;
    }
#else
    // every lane reads all W wave partials (same address in all lanes: an LDS broadcast) and adds them in registers in the
    // balanced tree over the wave index -- pairs (0,1) (2,3) ..., quads, ... -- i.e. the association of the xor butterfly the
    // first version ran on DPP (floating-point addition is commutative, so the bits are identical), without its dependent
    // cross-lane steps.
    double t[NV][W];
#pragma unroll
    for (int k = 0; k < NV; k++)
#pragma unroll
        for (int w2 = 0; w2 < W; w2++) t[k][w2] = buf[k * RED_MAXW + w2];
#pragma unroll
    for (int k = 0; k < NV; k++) {
#pragma unroll
        for (int span = 1; span < W; span <<= 1)
#pragma unroll
            for (int w2 = 0; w2 < W; w2 += 2 * span) t[k][w2] = t[k][w2] + t[k][w2 + span];
        v[k] = t[k][0];
    }
#endif
    parity ^= 1;
}

// Second level of the fixed reduction order, shared by the multi-kernel paths (large LP, generic BQP): out[v] = block tree over
// (thread t: partials t, t + T, ... of value v in ascending order), partials of value v at part[v * G ..].  Runs as ONE workgroup on
// the critical path of every reduction, so for 2 T < G <= T * FIN_U all loads of a pair of values are issued before the first (ordered)
// addition -- one memory latency instead of one per partial (5.1 -> about 3 us per launch at G = 1954).
constexpr int FIN_U = 16;
template <int T>
__device__ __forceinline__ void fin_reduce(const double *part, int G, int nv, double *out, double *red, int &parity, int stride = 0) {
    if (stride <= 0) stride = G;                 // partials of value v at part[v * stride ..]
    if (G > 2 * T && G <= T * FIN_U) {          // (with one or two partials per thread the plain loop below is the shorter program: A/B on the generic path)
        for (int v0 = 0; v0 < nv; v0 += 2) {
            const bool two = v0 + 1 < nv;
            const double *pa = part + (size_t)v0 * stride, *pb = part + (size_t)(two ? v0 + 1 : v0) * stride;
            double ta[FIN_U], tb[FIN_U];
#pragma unroll
            for (int u = 0; u < FIN_U; u++) {
                const int e = threadIdx.x + u * T;
                ta[u] = e < G ? pa[e] : 0.0;
                tb[u] = (two && e < G) ? pb[e] : 0.0;
            }
            double a[1] = {0.0}, b[1] = {0.0};
#pragma unroll
            for (int u = 0; u < FIN_U; u++) {
                const bool in = (int)threadIdx.x + u * T < G;
                a[0] = in ? a[0] + ta[u] : a[0];
                b[0] = in ? b[0] + tb[u] : b[0];
            }
            block_sum<T, 1>(a, red, parity);
            if (threadIdx.x == 0) out[v0] = a[0];
            if (two) {
                block_sum<T, 1>(b, red, parity);
                if (threadIdx.x == 0) out[v0 + 1] = b[0];
            }
        }
        return;
    }
    for (int v = 0; v < nv; v++) {
        const double *p = part + (size_t)v * stride;
        double a[1] = {0.0};
        for (int e = threadIdx.x; e < G; e += T) a[0] = a[0] + p[e];
        block_sum<T, 1>(a, red, parity);
        if (threadIdx.x == 0) out[v] = a[0];
    }
}

}  // namespace
