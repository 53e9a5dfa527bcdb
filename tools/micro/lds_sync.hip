// Micro-benchmark of the building blocks on the critical path of lp_window_kernel: cycles per
//   (a) s_barrier alone, (b) ds_write_b64 + barrier + ds_read_b64 (one LDS hand-over), (c) the 64-lane f64 butterfly (6 DPP steps),
//   (d) one IEEE f64 division, (e) a dependent chain of f64 additions, (f) a random 64-lane ds_read_b64 gather + add chain of L entries.
// One 512-thread workgroup per CU (256 workgroups), 2 waves per SIMD as in the solver.  Build: hipcc -O3 --offload-arch=gfx950 -ffp-contract=off
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>

template <int CTRL> __device__ __forceinline__ double dpp_mov(double v) {
    int lo = __double2loint(v), hi = __double2hiint(v);
    lo = __builtin_amdgcn_mov_dpp(lo, CTRL, 0xf, 0xf, true);
    hi = __builtin_amdgcn_mov_dpp(hi, CTRL, 0xf, 0xf, true);
    return __hiloint2double(hi, lo);
}
__device__ __forceinline__ double butterfly(double v) {
    v = v + __shfl_xor(v, 32, 64);
    v = v + __shfl_xor(v, 16, 64);
    v = v + dpp_mov<0xB1>(v); v = v + dpp_mov<0x4E>(v); v = v + dpp_mov<0x141>(v); v = v + dpp_mov<0x140>(v);
    return v;
}

template <int MODE, int L>
__global__ void __launch_bounds__(512) k(double *out, unsigned long long *cyc, const unsigned short *idx, int iters, double seed) {
    __shared__ double buf[1024];
    const int tid = threadIdx.x;
    double v = seed + tid * 1e-3, w = seed * 0.5 + 1.0;
    buf[tid] = v; buf[tid + 512] = w;
    unsigned a[L > 0 ? L : 1];
    for (int q = 0; q < L; q++) a[q] = idx[(blockIdx.x % 8) * 512 * 32 + q * 512 + tid];
    __syncthreads();
    unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; it++) {
        if (MODE == 0) { __builtin_amdgcn_s_barrier(); }
        if (MODE == 1) { buf[tid] = v; __syncthreads(); v = buf[(tid + 64) & 511] + 1.0; __syncthreads(); }   // two hand-overs
        if (MODE == 2) { v = butterfly(v) * 0.015625; }
        if (MODE == 3) { v = w / v + 1.5; }
        if (MODE == 4) {
#pragma unroll
            for (int q = 0; q < 16; q++) v = v + w;
        }
        if (MODE == 5) {
            double g[L > 0 ? L : 1];
#pragma unroll
            for (int q = 0; q < L; q++) g[q] = buf[a[q]];
            double acc = 0.0;
#pragma unroll
            for (int q = 0; q < L; q++) acc = acc + g[q];
            v = acc * 0.1;
            __syncthreads(); buf[tid] = v; __syncthreads();
        }
        if (MODE == 6) { __syncthreads(); buf[tid] = v; __syncthreads(); v = v * 0.5; }       // the hand-over part of MODE 5 alone
    }
    unsigned long long t1 = __builtin_amdgcn_s_memtime();
    out[blockIdx.x * 512 + tid] = v;
    if (tid == 0) cyc[blockIdx.x] = t1 - t0;
}

#define RUN(MODE, L, name, per)                                                                                         \
    {                                                                                                                   \
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);                                                    \
        k<MODE, L><<<256, 512>>>(out, cyc, idx, 100, 1.25);                                                             \
        hipEventRecord(e0); k<MODE, L><<<256, 512>>>(out, cyc, idx, iters, 1.25); hipEventRecord(e1);                   \
        hipDeviceSynchronize(); float ms; hipEventElapsedTime(&ms, e0, e1);                                             \
        unsigned long long h[256]; hipMemcpy(h, cyc, sizeof(h), hipMemcpyDeviceToHost);                                 \
        double s = 0; for (int i = 0; i < 256; i++) s += h[i];                                                          \
        printf("%-46s %8.1f cycles  %7.1f ns  (clock %.2f GHz)\n", name, s / 256 / iters / per, 1e6 * ms / iters / per, \
               s / 256 / (ms * 1e6));                                                                                   \
    }

int main() {
    const int iters = 20000;
    double *out; unsigned long long *cyc; unsigned short *idx;
    hipMalloc(&out, 256 * 512 * 8); hipMalloc(&cyc, 256 * 8); hipMalloc(&idx, 8 * 512 * 32 * 2);
    unsigned short *h = (unsigned short *)malloc(8 * 512 * 32 * 2);
    srand(1); for (int i = 0; i < 8 * 512 * 32; i++) h[i] = rand() % 1000;
    hipMemcpy(idx, h, 8 * 512 * 32 * 2, hipMemcpyHostToDevice);
    RUN(0, 0, "s_barrier (8 waves)", 1)
    RUN(1, 0, "ds_write+barrier+ds_read hand-over", 2)
    RUN(2, 0, "64-lane f64 butterfly (6 steps) + mul", 1)
    RUN(3, 0, "f64 division + add", 1)
    RUN(4, 0, "dependent f64 add", 16)
    RUN(6, 0, "barrier+ds_write+barrier (gather frame)", 1)
    RUN(5, 4, "frame + random gather L=4 + add chain", 1)
    RUN(5, 8, "frame + random gather L=8 + add chain", 1)
    RUN(5, 12, "frame + random gather L=12 + add chain", 1)
    RUN(5, 16, "frame + random gather L=16 + add chain", 1)
    RUN(5, 24, "frame + random gather L=24 + add chain", 1)
    return 0;
}
