// Diagnostic (tools/trp_old_projection.py): trp_w_project_kernel AS IT STOOD before commit e01bca0 (W staged in LDS, uniform-address
// ds_reads feeding the FMA chain, then 16 block reductions through a second LDS array), built on its own so that the failure beside
// another stream's kernels (DESIGN 9b') can be studied under different compiler settings.
#include <hip/hip_runtime.h>
#include <stdint.h>
struct F16 { typedef _Float16 elem; typedef _Float16 vec8 __attribute__((ext_vector_type(8))); };
__device__ __forceinline__ float block_sum256(float v, float* red) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
    __syncthreads();
    return red[0] + red[1] + red[2] + red[3];
}

// IN = float: e is TRP's fp32 E [B,S,C].  IN = a 16-bit storage type: e is a 16-bit NHWC map with channel stride e_ld
// (RELATION_SUB_SAMPLE: the output of the up-sampling deconv, association.py:254-268).
template <int C, class IN>
__global__ void __launch_bounds__(256) trp_w_project_kernel(const void* __restrict__ e_, int e_ld, const float* __restrict__ w,
                                                           const float* __restrict__ bias, float* __restrict__ y,
                                                           float* __restrict__ part, int S) {
    constexpr int G = 8, CG = C / G;
#ifdef TWO_COPIES          // the table twice; odd lanes read the second copy: no instruction reads ONE address in all lanes
    __shared__ float ws2[2 * (C * C + C)];
    __shared__ float red[4];
    for (int i = threadIdx.x; i < C * C + C; i += 256) ws2[i] = ws2[C * C + C + i] = i < C * C ? w[i] : bias[i - C * C];
    __syncthreads();
    const float* ws = ws2 + (threadIdx.x & 1) * (C * C + C);
#else
    __shared__ float ws[C * C + C];
    __shared__ float red[4];
    for (int i = threadIdx.x; i < C * C + C; i += 256) ws[i] = i < C * C ? w[i] : bias[i - C * C];
    __syncthreads();
#endif
    const int b = blockIdx.y, s = blockIdx.x * 256 + threadIdx.x;
    const size_t row = ((size_t)b * S + s) * C;
    float xv[C], yv[C];
    if constexpr (sizeof(IN) == 4) {
        const float* e = reinterpret_cast<const float*>(e_);
#pragma unroll
        for (int k = 0; k < C / 4; ++k) {
            const float4 v = *reinterpret_cast<const float4*>(e + row + 4 * k);
            xv[4 * k] = v.x; xv[4 * k + 1] = v.y; xv[4 * k + 2] = v.z; xv[4 * k + 3] = v.w;
        }
    } else {
        const char* e = reinterpret_cast<const char*>(e_) + ((size_t)b * S + s) * e_ld * 2;
#pragma unroll
        for (int k = 0; k < C / 8; ++k) {
            const typename IN::vec8 v = __builtin_bit_cast(typename IN::vec8, *reinterpret_cast<const uint4*>(e + 16 * k));
#pragma unroll
            for (int j = 0; j < 8; ++j) xv[8 * k + j] = (float)v[j];
        }
    }
#pragma unroll
    for (int co = 0; co < C; ++co) {
        float acc = ws[C * C + co];
#pragma unroll
        for (int ci = 0; ci < C; ++ci) acc = fmaf(ws[co * C + ci], xv[ci], acc);
        yv[co] = acc;
    }
#pragma unroll
    for (int k = 0; k < C / 4; ++k)
        *reinterpret_cast<float4*>(y + row + 4 * k) = make_float4(yv[4 * k], yv[4 * k + 1], yv[4 * k + 2], yv[4 * k + 3]);
    float* po = part + (((size_t)b * gridDim.x + blockIdx.x) * G) * 2;
#pragma unroll
    for (int g = 0; g < G; ++g) {
        float sum = 0.f;
#pragma unroll
        for (int c = 0; c < CG; ++c) sum += yv[CG * g + c];
        const float mean = block_sum256(sum, red) * (1.f / (256 * CG));
        float m2 = 0.f;
#pragma unroll
        for (int c = 0; c < CG; ++c) m2 += (yv[CG * g + c] - mean) * (yv[CG * g + c] - mean);
        m2 = block_sum256(m2, red);
        if (threadIdx.x == 0) { po[2 * g] = mean; po[2 * g + 1] = m2; }
    }
}


extern "C" int old_project_launch(const float* e, const float* w, const float* bias, float* y, float* part, int S, int batch, void* stream) {
    hipLaunchKernelGGL((trp_w_project_kernel<32, float>), dim3(S / 256, batch), dim3(256), 0, (hipStream_t)stream, (const void*)e, 0, w, bias, y, part, S);
    return (int)hipGetLastError();
}
