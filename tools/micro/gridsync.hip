// Microbenchmark: cost of a grid-wide barrier on MI355X (cooperative groups vs a hand-rolled atomic barrier), G workgroups x 256 threads.
#include <hip/hip_runtime.h>
#include <hip/hip_cooperative_groups.h>
#include <cstdio>
#include <cstdlib>
namespace cg = cooperative_groups;

__global__ void k_cg(double *buf, int iters) {
    cg::grid_group g = cg::this_grid();
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    double v = buf[i];
    for (int it = 0; it < iters; it++) { buf[i] = v + 1.0; g.sync(); v = buf[(i + 256) % (gridDim.x * blockDim.x)]; }
    buf[i] = v;
}

__device__ __forceinline__ void gbar(unsigned *count, unsigned *gen, unsigned nb, unsigned &local_gen) {
    __threadfence();
    __syncthreads();
    if (threadIdx.x == 0) {
        local_gen++;
        if (atomicAdd(count, 1u) == nb - 1) { atomicExch(count, 0u); __threadfence(); atomicExch(gen, local_gen); }
        else { long spins = 0; while (__hip_atomic_load(gen, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) != local_gen && ++spins < 200000000L) __builtin_amdgcn_s_sleep(1); }
    }
    __syncthreads();
    __threadfence();
}

__global__ void k_own(double *buf, int iters, unsigned *count, unsigned *gen) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    unsigned lg = 0;
    double v = buf[i];
    for (int it = 0; it < iters; it++) { buf[i] = v + 1.0; gbar(count, gen, gridDim.x, lg); v = buf[(i + 256) % (gridDim.x * blockDim.x)]; }
    buf[i] = v;
}

int main(int argc, char **argv) {
    int G = argc > 1 ? atoi(argv[1]) : 366, iters = argc > 2 ? atoi(argv[2]) : 2000;
    double *buf; unsigned *ctr;
    hipMalloc(&buf, sizeof(double) * G * 256); hipMemset(buf, 0, sizeof(double) * G * 256);
    hipMalloc(&ctr, 8); hipMemset(ctr, 0, 8);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int rep = 0; rep < 2; rep++) {
        void *args[] = {&buf, &iters};
        hipEventRecord(e0);
        hipError_t e = hipLaunchCooperativeKernel((void *)k_cg, dim3(G), dim3(256), args, 0, 0);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        printf("cooperative groups grid.sync: G=%d, %d syncs in %.2f ms -> %.2f us each (%s)\n", G, iters, ms, 1e3 * ms / iters, hipGetErrorString(e));
        unsigned *count = ctr, *gen = ctr + 1; hipMemset(ctr, 0, 8);
        void *args2[] = {&buf, &iters, &count, &gen};
        hipEventRecord(e0);
        e = hipLaunchCooperativeKernel((void *)k_own, dim3(G), dim3(256), args2, 0, 0);
        hipEventRecord(e1); hipEventSynchronize(e1);
        hipEventElapsedTime(&ms, e0, e1);
        printf("atomic barrier:               G=%d, %d syncs in %.2f ms -> %.2f us each (%s)\n", G, iters, ms, 1e3 * ms / iters, hipGetErrorString(e));
    }
    double h[4]; hipMemcpy(h, buf, sizeof(h), hipMemcpyDeviceToHost); printf("check %g (expect %d)\n", h[0], 2 * 2 * iters);
    return 0;
}
