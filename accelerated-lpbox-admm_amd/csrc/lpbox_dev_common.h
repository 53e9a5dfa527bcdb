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

__device__ __forceinline__ int wave_max_int(int v) {
    for (int off = 1; off < 64; off <<= 1) { int o = __shfl_xor(v, off, 64); v = o > v ? o : v; }
    return __builtin_amdgcn_readfirstlane(v);
}

constexpr int RED_MAXV = 5;    // values reduced together
constexpr int RED_MAXW = 16;   // waves per workgroup

// Sum NV per-thread partials over the workgroup; every thread receives the totals.  Wave partials go through LDS and
// are combined by a second butterfly (lane i reads partial i mod W; balanced tree over the wave index), i.e. ONE LDS
// round trip instead of W dependent reads.  `red` is a ping-pong scratch (2 * RED_MAXV * RED_MAXW doubles): a thread
// can run at most one block_sum ahead of the slowest one.
template <int T, int NV>
__device__ __forceinline__ void block_sum(double (&v)[NV], double *red, int &parity) {
    constexpr int W = T / 64;
    static_assert(W == 1 || W == 2 || W == 4 || W == 8 || W == 16, "waves per workgroup");
#pragma unroll
    for (int k = 0; k < NV; k++) v[k] = wave_allreduce_sum(v[k]);
    if constexpr (W == 1) return;        // one wavefront per instance: no LDS round trip, no barrier
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    double *buf = red + parity * (RED_MAXV * RED_MAXW);
    if (lane == 0) {
#pragma unroll
        for (int k = 0; k < NV; k++) buf[k * RED_MAXW + w] = v[k];
    }
    __syncthreads();
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
    parity ^= 1;
}

}  // namespace
