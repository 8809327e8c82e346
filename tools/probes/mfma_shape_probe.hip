// Probe (GPU box only, not part of the library): does the k-loop of a C = 32 chain layer run faster on the 16x16x32 MFMA
// than on the 32x32x16 one?  Same FLOPs, same LDS bytes per FLOP (one ds_read_b128 per 32 MFMA cycles), same weight
// fragments from L2, random fp16 data, 8 waves per workgroup (two per SIMD) or 4 (one per SIMD), one workgroup per CU
// (150 KB of LDS).  MI355X_MICROARCH.md "DVFS give-back" item 7 measured 1.12-1.15x for the bare loops.
// Prints nothing itself: tools/probes/mfma_shape_probe.py launches and times it.
#include <hip/hip_runtime.h>
#include <stdint.h>
typedef _Float16 h8_t __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

#define PS 80
#define RPITCH 4096          // 50 pixels * 80 B + 96 B pad: == 48 * 80 (mod 256)
#define LDS_BYTES (38 * RPITCH)

__device__ __forceinline__ void fill_lds(char* smem, const uint4* src, int nthr) {
    for (int o = threadIdx.x * 16; o < LDS_BYTES; o += nthr * 16) *reinterpret_cast<uint4*>(smem + o) = src[(o / 16) & 4095];
    __syncthreads();
}

// ---- 32x32x16, NT pixel groups of 32 per wave, one A fragment per k-step (the current chain_body<32> loop) -------------
template <int NT, int NWAVES>
__global__ void __launch_bounds__(64 * NWAVES) probe32(const uint4* __restrict__ w, const uint4* __restrict__ rnd, float* out,
                                                        int layers, unsigned long long* clk) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    fill_lds(smem, rnd, 64 * NWAVES);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, n = lane & 31, hh = lane >> 5;
    int pixoff[NT];
#pragma unroll
    for (int g = 0; g < NT; ++g) {
        const int q = ((wave * NT + g) * 32 + n) % (36 * 48);
        pixoff[g] = (q / 48 + 1) * RPITCH + (q % 48 + 1) * PS + hh * 16;
    }
    constexpr int KS = 18, NPF = 6;
    uint4 aq[NPF];
    const uint4* ap = w + lane;
#pragma unroll
    for (int k = 0; k < NPF; ++k) aq[k] = ap[k * 64];
    f32x16 acc[NT];
#pragma unroll
    for (int g = 0; g < NT; ++g)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[g][r] = 0.f;
    unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    auto boff = [&](int ks) { const int t = ks / 2, c = ks % 2; return (t / 3 - 1) * RPITCH + (t % 3 - 1) * PS + c * 32; };
    for (int l = 0; l < layers; ++l) {
        uint4 bq[NT];
#pragma unroll
        for (int g = 0; g < NT; ++g) bq[g] = *reinterpret_cast<const uint4*>(smem + pixoff[g] + boff(0));
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
            const h8_t a = __builtin_bit_cast(h8_t, aq[ks % NPF]);
            aq[ks % NPF] = ap[(((l * KS + ks + NPF) % (8 * KS))) * 64];
#pragma unroll
            for (int g = 0; g < NT; ++g) {
                acc[g] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, __builtin_bit_cast(h8_t, bq[g]), acc[g], 0, 0, 0);
                if (ks + 1 < KS) bq[g] = *reinterpret_cast<const uint4*>(smem + pixoff[g] + boff(ks + 1));
            }
            __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);
#pragma unroll
            for (int g = 0; g < NT; ++g) {
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
            }
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    float s = 0.f;
#pragma unroll
    for (int g = 0; g < NT; ++g)
#pragma unroll
        for (int r = 0; r < 16; ++r) s += acc[g][r];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (threadIdx.x == 0) { clk[blockIdx.x * 2] = t1 - t0; clk[blockIdx.x * 2 + 1] = r1 - r0; }
}

// ---- 16x16x32, NG pixel groups of 16 per wave x 2 output-channel tiles, two A fragments per tap ---------------------------
// lane (col, q): pixel pi(col) of the group, input channels 8 sigma(q) .. + 7 (pi: columns {0-3, 12-15} take the even
// pixels, {4-11} the odd ones; sigma = [0, 2, 1, 3]: with an 80-byte pixel stride every 16-lane phase of a ds_read_b128
// then covers all 64 banks once)
template <int NG, int NWAVES, int APRE, int NB = NG>
__global__ void __launch_bounds__(64 * NWAVES) probe16(const uint4* __restrict__ w, const uint4* __restrict__ rnd, float* out,
                                                        int layers, unsigned long long* clk) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    fill_lds(smem, rnd, 64 * NWAVES);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, col = lane & 15, q = lane >> 4;
    const int pi = (col < 4) ? 2 * col : (col >= 12 ? 2 * (col - 8) : 2 * (col - 4) + 1);
    const int sg = ((q & 1) << 1) | (q >> 1);
    int pixoff[NG];
#pragma unroll
    for (int g = 0; g < NG; ++g) {
        const int p = ((wave * NG + g) * 16 + pi) % (36 * 48);
        pixoff[g] = (p / 48 + 1) * RPITCH + (p % 48 + 1) * PS + sg * 16;
    }
    constexpr int TAPS = 9, NPF = APRE;          // ring depth in taps
    uint4 aq[NPF][2];
    const uint4* ap = w + lane;
#pragma unroll
    for (int k = 0; k < NPF; ++k) { aq[k][0] = ap[(2 * k) * 64]; aq[k][1] = ap[(2 * k + 1) * 64]; }
    f32x4 acc[NG][2];
#pragma unroll
    for (int g = 0; g < NG; ++g)
#pragma unroll
        for (int m = 0; m < 2; ++m) acc[g][m] = f32x4{0.f, 0.f, 0.f, 0.f};
    unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    auto boff = [&](int t) { return (t / 3 - 1) * RPITCH + (t % 3 - 1) * PS; };
    constexpr int NJ = TAPS * NG;              // B ring of NB fragments over the layer's (tap, group) sequence
    for (int l = 0; l < layers; ++l) {
        uint4 bq[NB];
#pragma unroll
        for (int j = 0; j < NB; ++j) bq[j] = *reinterpret_cast<const uint4*>(smem + pixoff[j % NG] + boff(j / NG));
#pragma unroll
        for (int t = 0; t < TAPS; ++t) {
            const h8_t a0 = __builtin_bit_cast(h8_t, aq[t % NPF][0]), a1 = __builtin_bit_cast(h8_t, aq[t % NPF][1]);
            aq[t % NPF][0] = ap[((2 * (l * TAPS + t + NPF)) % (16 * TAPS)) * 64];
            aq[t % NPF][1] = ap[((2 * (l * TAPS + t + NPF) + 1) % (16 * TAPS)) * 64];
#pragma unroll
            for (int g = 0; g < NG; ++g) {
                const int j = t * NG + g;
                const h8_t b = __builtin_bit_cast(h8_t, bq[j % NB]);
                acc[g][0] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a0, b, acc[g][0], 0, 0, 0);
                acc[g][1] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a1, b, acc[g][1], 0, 0, 0);
                if (j + NB < NJ) bq[j % NB] = *reinterpret_cast<const uint4*>(smem + pixoff[(j + NB) % NG] + boff((j + NB) / NG));
            }
            __builtin_amdgcn_sched_group_barrier(0x020, 2, 0);
#pragma unroll
            for (int g = 0; g < NG; ++g) {
                __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);
                __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
            }
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    float s = 0.f;
#pragma unroll
    for (int g = 0; g < NG; ++g)
#pragma unroll
        for (int m = 0; m < 2; ++m)
#pragma unroll
            for (int r = 0; r < 4; ++r) s += acc[g][m][r];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (threadIdx.x == 0) { clk[blockIdx.x * 2] = t1 - t0; clk[blockIdx.x * 2 + 1] = r1 - r0; }
}

extern "C" int probe_run(int variant, const void* w, const void* rnd, void* out, int layers, void* clk, int blocks) {
    const uint4* W = (const uint4*)w; const uint4* R = (const uint4*)rnd; float* O = (float*)out; unsigned long long* K = (unsigned long long*)clk;
#define GO(fn, nthr)                                                                                       \
    do {                                                                                                   \
        (void)hipFuncSetAttribute((const void*)fn, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES);       \
        hipLaunchKernelGGL(fn, dim3(blocks), dim3(nthr), LDS_BYTES, 0, W, R, O, layers, K);                 \
    } while (0)
    switch (variant) {
        case 0: GO((probe32<7, 8>), 512); break;
        case 1: GO((probe16<14, 8, 3>), 512); break;
        case 2: GO((probe32<7, 4>), 256); break;
        case 3: GO((probe16<14, 4, 3>), 256); break;
        case 4: GO((probe16<14, 8, 2>), 512); break;
        case 5: GO((probe32<6, 8>), 512); break;
        case 6: GO((probe16<12, 8, 3>), 512); break;
        case 7: GO((probe16<14, 8, 3, 7>), 512); break;
        case 8: GO((probe16<14, 8, 3, 10>), 512); break;
        default: return 1;
    }
    return hipGetLastError() == hipSuccess ? 0 : 2;
}
