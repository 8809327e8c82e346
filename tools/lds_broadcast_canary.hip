// Diagnostic (tools/lds_broadcast_canary.py): the access pattern of the TRP projection kernel that dropped FMA terms beside
// another stream's kernels (DESIGN 9b'), reduced to its bones: a 32x32 fp32 table staged in LDS, read back through
// uniform-address ds_reads that feed a fully unrolled FMA chain; every thread checks its 32 results against the same sums
// taken with the table read through the scalar cache.  MODE 1 replaces the LDS reads by the scalar ones (the fix's form).
#include <hip/hip_runtime.h>
#include <stdint.h>

template <int MODE>
__global__ void __launch_bounds__(256) bcast_kernel(const float* __restrict__ w, const float* __restrict__ x, unsigned* report, int rounds) {
    constexpr int C = 32;
    __shared__ float ws[C * C + C];
    for (int i = threadIdx.x; i < C * C + C; i += 256) ws[i] = w[i];
    __syncthreads();
    const size_t row = ((size_t)blockIdx.x * 256 + threadIdx.x) * C;
    float xv[C];
#pragma unroll
    for (int k = 0; k < C / 4; ++k) {
        const float4 v = *reinterpret_cast<const float4*>(x + row + 4 * k);
        xv[4 * k] = v.x; xv[4 * k + 1] = v.y; xv[4 * k + 2] = v.z; xv[4 * k + 3] = v.w;
    }
    for (int it = 0; it < rounds; ++it) {
        float yv[C];
#pragma unroll
        for (int co = 0; co < C; ++co) {
            float acc = MODE ? w[C * C + co] : ws[C * C + co];
#pragma unroll
            for (int ci = 0; ci < C; ++ci) acc = fmaf(MODE ? w[co * C + ci] : ws[co * C + ci], xv[ci], acc);
            yv[co] = acc;
        }
        // reference: the same sums, the table through the scalar cache, a different instruction stream
#pragma unroll 1
        for (int co = 0; co < C; ++co) {
            float acc = w[C * C + co];
#pragma unroll 4
            for (int ci = 0; ci < C; ++ci) acc = fmaf(w[co * C + ci], xv[ci], acc);
            float got = yv[0];
#pragma unroll
            for (int k = 1; k < C; ++k) got = co == k ? yv[k] : got;
            if (got != acc) {
                const unsigned slot = atomicAdd(report, 1u);
                if (slot < 32) { report[4 + 4 * slot] = blockIdx.x; report[5 + 4 * slot] = threadIdx.x; report[6 + 4 * slot] = (unsigned)co; report[7 + 4 * slot] = (unsigned)it; }
            }
        }
    }
}

extern "C" int bcast_launch(const float* w, const float* x, unsigned* report, int blocks, int rounds, int mode, void* stream) {
    if (mode) hipLaunchKernelGGL(bcast_kernel<1>, dim3(blocks), dim3(256), 0, (hipStream_t)stream, w, x, report, rounds);
    else hipLaunchKernelGGL(bcast_kernel<0>, dim3(blocks), dim3(256), 0, (hipStream_t)stream, w, x, report, rounds);
    return (int)hipGetLastError();
}
