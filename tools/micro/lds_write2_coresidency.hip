// Do PAIRED LDS accesses (ds_write2_b64 / ds_read2_b64: one instruction, two addresses) survive when a co-resident workgroup's LDS
// allocation does not start at 0, i.e. when the two addresses can lie on either side of the absolute 64 KB / 128 KB marks?
// Every thread writes pairs 32 bytes apart over the whole region (the compiler fuses them into ds_write2_b64), verification reads are single.
// usage: lds_write2_coresidency [lds_kb=78] [threads=256] [groups=8192] [read2=0]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
__global__ void k(unsigned *bad, int words, int read2) {
    extern __shared__ __attribute__((aligned(16))) unsigned lds[];
    const unsigned tag = blockIdx.x * 2654435761u;
    uint2 *p = (uint2 *)lds;                       // 8-byte units
    const int units = words / 2;
    // unit u and unit u + 4 (32 bytes apart) by one thread: u = 8 * j + (t & 3) + ... cover all units with pairs
    for (int base = 0; base + 8 <= units; base += 8 * (int)blockDim.x) {
        const int u = base + 8 * (threadIdx.x) + 0;
        if (u + 8 <= units) {
#pragma unroll
            for (int q = 0; q < 4; q++) {
                uint2 a = {tag + 2 * (u + q), tag + 2 * (u + q) + 1}, b = {tag + 2 * (u + q + 4), tag + 2 * (u + q + 4) + 1};
                p[u + q] = a; p[u + q + 4] = b;                       // -> ds_write2_b64 offset0:q offset1:q+4
            }
        }
    }
    __syncthreads();
    unsigned wrong = 0;
    if (!read2) {
        for (int u = threadIdx.x; u < (units / 8) * 8; u += blockDim.x) {
            const uint2 v = p[u];
            wrong += (v.x != tag + 2 * u) + (v.y != tag + 2 * u + 1);
        }
    } else {
        for (int base = 0; base + 8 <= units; base += 8 * (int)blockDim.x) {
            const int u = base + 8 * threadIdx.x;
            if (u + 8 <= units) {
#pragma unroll
                for (int q = 0; q < 4; q++) {
                    const uint2 a = p[u + q], b = p[u + q + 4];            // -> ds_read2_b64
                    wrong += (a.x != tag + 2 * (u + q)) + (a.y != tag + 2 * (u + q) + 1) + (b.x != tag + 2 * (u + q + 4)) + (b.y != tag + 2 * (u + q + 4) + 1);
                }
            }
        }
    }
    if (wrong) atomicAdd(bad, wrong);
}
int main(int argc, char **argv) {
    const int lds_kb = argc > 1 ? atoi(argv[1]) : 78, threads = argc > 2 ? atoi(argv[2]) : 256, groups = argc > 3 ? atoi(argv[3]) : 8192;
    const int read2 = argc > 4 ? atoi(argv[4]) : 0;
    unsigned *bad; (void)hipMalloc(&bad, 8); (void)hipMemset(bad, 0, 8);
    (void)hipFuncSetAttribute((const void *)k, hipFuncAttributeMaxDynamicSharedMemorySize, lds_kb * 1024);
    for (int rep = 0; rep < 4; rep++) hipLaunchKernelGGL(k, dim3(groups), dim3(threads), lds_kb * 1024, 0, bad, lds_kb * 256, read2);
    unsigned h[2]; (void)hipMemcpy(h, bad, 8, hipMemcpyDeviceToHost);
    printf("lds %d KB, %d threads, %d groups, read2 %d: %u wrong words (%s)\n", lds_kb, threads, groups, read2, h[0], hipGetErrorString(hipGetLastError()));
    return h[0] != 0;
}
