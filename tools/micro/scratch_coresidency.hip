// Does per-wave scratch survive several workgroups sharing a CU?  Every lane keeps a private array with lane-unique values (forced into
// scratch by dynamic indexing), the workgroup also holds `lds_kb` of LDS (sets how many workgroups share a CU), spins, then checks the array.
// usage: scratch_coresidency [lds_kb=78] [threads=256] [groups=4096]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#ifndef NA
#define NA 600          // large enough that the array cannot be promoted to registers: real scratch (check .private_segment_fixed_size)
#endif
__global__ void k(unsigned *bad, int spin, int salt) {
    extern __shared__ unsigned lds[];
    unsigned a[NA];
    const unsigned id = blockIdx.x * blockDim.x + threadIdx.x;
    for (int i = 0; i < NA; i++) a[(i * 7 + salt) % NA] = id * 31u + i;
    lds[threadIdx.x] = id;
    __syncthreads();
    unsigned acc = 0;
    for (int s = 0; s < spin; s++) { acc += lds[(threadIdx.x + s) % blockDim.x]; a[(s + salt) % NA] += 0u; __builtin_amdgcn_s_sleep(1); }
    unsigned wrong = 0;
    for (int i = 0; i < NA; i++) wrong += a[(i * 7 + salt) % NA] != id * 31u + i;
    if (wrong) atomicAdd(bad, wrong);
    if (acc == 0xFFFFFFFFu) bad[1] = acc;
}
int main(int argc, char **argv) {
    const int lds_kb = argc > 1 ? atoi(argv[1]) : 78, threads = argc > 2 ? atoi(argv[2]) : 256, groups = argc > 3 ? atoi(argv[3]) : 4096;
    unsigned *bad; hipMalloc(&bad, 8); hipMemset(bad, 0, 8);
    hipFuncSetAttribute((const void *)k, hipFuncAttributeMaxDynamicSharedMemorySize, lds_kb * 1024);
    for (int rep = 0; rep < 5; rep++) hipLaunchKernelGGL(k, dim3(groups), dim3(threads), lds_kb * 1024, 0, bad, 200, rep);
    unsigned h[2]; hipMemcpy(h, bad, 8, hipMemcpyDeviceToHost);
    printf("lds %d KB, %d threads, %d groups: %u corrupted scratch words (%s)\n", lds_kb, threads, groups, h[0], hipGetErrorString(hipGetLastError()));
    return h[0] != 0;
}
