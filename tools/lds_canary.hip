// Diagnostic (tools/lds_canary.py): a kernel that fills its LDS and 32 registers per thread with a pattern, keeps checking
// them for a while and reports anything that changed under it -- run beside another stream's kernels it shows whether those
// write outside their own LDS allocation, or whether wave state is damaged when queues share the CUs.
#include <hip/hip_runtime.h>
#include <stdint.h>

extern "C" __global__ void __launch_bounds__(256) canary_kernel(unsigned* report, int words, int rounds) {
    extern __shared__ unsigned lds[];
    const unsigned tag = 0xC0DE0000u ^ (blockIdx.x << 20);
    for (int i = threadIdx.x; i < words; i += 256) lds[i] = tag ^ (unsigned)i;
    unsigned r[32];
#pragma unroll
    for (int k = 0; k < 32; ++k) r[k] = tag + 977u * k + threadIdx.x;
    __syncthreads();
    for (int it = 0; it < rounds; ++it) {
        for (int i = threadIdx.x; i < words; i += 256) {
            const unsigned v = lds[i];
            if (v != (tag ^ (unsigned)i)) {
                const unsigned slot = atomicAdd(report, 1u);
                if (slot < 64) {
                    report[4 + 4 * slot] = blockIdx.x; report[5 + 4 * slot] = (unsigned)i; report[6 + 4 * slot] = v; report[7 + 4 * slot] = (unsigned)it;
                }
                lds[i] = tag ^ (unsigned)i;
            }
        }
#pragma unroll
        for (int k = 0; k < 32; ++k) {
            asm volatile("" : "+v"(r[k]));                 // keep the value in a VGPR across the loop
            if (r[k] != tag + 977u * k + threadIdx.x) {
                const unsigned slot = atomicAdd(report + 1, 1u);
                if (slot < 16) { report[300 + 4 * slot] = blockIdx.x; report[301 + 4 * slot] = threadIdx.x; report[302 + 4 * slot] = (unsigned)k; report[303 + 4 * slot] = r[k]; }
                r[k] = tag + 977u * k + threadIdx.x;
            }
        }
        __builtin_amdgcn_s_sleep(8);
    }
}

extern "C" int canary_launch(unsigned* report, int blocks, int lds_bytes, int rounds, void* stream) {
    hipFuncSetAttribute((const void*)canary_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes);
    hipLaunchKernelGGL(canary_kernel, dim3(blocks), dim3(256), lds_bytes, (hipStream_t)stream, report, lds_bytes / 4, rounds);
    return (int)hipGetLastError();
}
