// Do two workgroups that share a CU keep their LDS allocations apart when each holds MORE than 64 KB?  Every workgroup fills `lds_kb` of
// dynamic LDS with a pattern unique to it (also through ds_write_b64 / b128 at offsets beyond 64 KB), spins so that neighbours overlap
// in time, and verifies.  usage: lds_coresidency [lds_kb=78] [threads=256] [groups=8192]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
__global__ void k(unsigned *bad, int words, int spin) {
    extern __shared__ __attribute__((aligned(16))) unsigned lds[];
    const unsigned tag = blockIdx.x * 2654435761u;
    for (int i = threadIdx.x * 4; i < words; i += blockDim.x * 4) {
        uint4 v = {tag + i, tag + i + 1, tag + i + 2, tag + i + 3};
        *(uint4 *)(lds + i) = v;
    }
    __syncthreads();
    unsigned wrong = 0;
    for (int s = 0; s < spin; s++) {
        for (int i = threadIdx.x * 2; i < words; i += blockDim.x * 2) {
            const uint2 v = *(const uint2 *)(lds + i);
            wrong += (v.x != tag + i) + (v.y != tag + i + 1);
        }
        __syncthreads();
        for (int i = threadIdx.x * 4; i < words; i += blockDim.x * 4) {       // rewrite (a neighbour's stray write would be caught next round)
            uint4 v = {tag + i, tag + i + 1, tag + i + 2, tag + i + 3};
            *(uint4 *)(lds + i) = v;
        }
        __syncthreads();
    }
    if (wrong) atomicAdd(bad, wrong);
}
int main(int argc, char **argv) {
    const int lds_kb = argc > 1 ? atoi(argv[1]) : 78, threads = argc > 2 ? atoi(argv[2]) : 256, groups = argc > 3 ? atoi(argv[3]) : 8192;
    unsigned *bad; (void)hipMalloc(&bad, 8); (void)hipMemset(bad, 0, 8);
    (void)hipFuncSetAttribute((const void *)k, hipFuncAttributeMaxDynamicSharedMemorySize, lds_kb * 1024);
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    hipLaunchKernelGGL(k, dim3(groups), dim3(threads), lds_kb * 1024, 0, bad, lds_kb * 256, 6);      // warm
    (void)hipEventRecord(e0, 0);
    hipLaunchKernelGGL(k, dim3(groups), dim3(threads), lds_kb * 1024, 0, bad, lds_kb * 256, 6);
    (void)hipEventRecord(e1, 0);
    unsigned h[2]; (void)hipMemcpy(h, bad, 8, hipMemcpyDeviceToHost);
    float ms = 0; (void)hipEventElapsedTime(&ms, e0, e1);
    // (two workgroups per CU show as about half the time per KB of the one-per-CU sizes)
    printf("lds %d KB, %d threads, %d groups: %u wrong words, %.2f ms = %.3f us per workgroup-KB (%s)\n", lds_kb, threads, groups, h[0], ms,
           1e3 * ms / groups / lds_kb, hipGetErrorString(hipGetLastError()));
    return h[0] != 0;
}
