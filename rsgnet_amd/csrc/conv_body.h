// Device body of the implicit-GEMM convolution (see conv_igemm.hip for the design notes); a function
// template so that both the stand-alone kernel and the grouped launch (group.hip) instantiate it.
#pragma once
#include "conv_args.h"

// Diagnostic build (-DRSG_CONV_STAMPS): thread 0 of every workgroup records shader-clock stamps at the phase
// boundaries (tools/conv_group_stamps.py); the production build contains none of this.
#ifdef RSG_CONV_STAMPS
static __device__ unsigned long long* g_conv_stamps = nullptr;      // [blocks][8]; one copy per translation unit
#define RSG_CSTAMP(slot)                                                                      \
    do {                                                                                      \
        if (g_conv_stamps && threadIdx.x == 0) {                                              \
            unsigned long long t__;                                                           \
            asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t__)::"memory");      \
            g_conv_stamps[(size_t)blockIdx.x * 8 + (slot)] = t__;                             \
        }                                                                                     \
    } while (0)
#else
#define RSG_CSTAMP(slot) do {} while (0)
#endif

template <class DT, int CINB, int TAPH, int TAPW, int WM, int WN, int NT, bool TAIL = false>
__device__ __forceinline__ void conv_body(const ConvArgs& p, const int block_x, const int block_y, char* smem) {
    constexpr int NTHR = 64 * WM * WN;
    constexpr int KPT = CINB / 16;       // k-steps per filter tap
    constexpr int NTAPS = TAPH * TAPW;
    constexpr int PS = 2 * CINB + 16;    // LDS bytes per pixel (odd number of 16-B slots)
    constexpr int CPP = CINB / 8;        // 16-byte chunks per pixel
    constexpr int NLD = 8;               // staging loads in flight per thread
    constexpr int KSB = NTAPS * KPT;     // k-steps per channel block
    constexpr int NPF = KSB < 6 ? KSB : 6;   // A-operand prefetch depth

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave / WN, wn = wave % WN;
    const int n = lane & 31, hh = lane >> 5;
    RSG_CSTAMP(1);

    int bx = block_x;
    const int tix = bx % p.tiles_x;
    bx /= p.tiles_x;
    const int tiy = bx % p.tiles_y;
    const int b = bx / p.tiles_y;
    const int y0 = tiy * p.TH, x0 = tix * p.TW;
    const int ct = block_y * WM + wm;

    int pixoff[NT];
    int opix[NT];
#pragma unroll
    for (int g = 0; g < NT; ++g) {
        const int q = (wn * NT + g) * 32 + n;
        const int py = q / p.TW, px = q - py * p.TW;
        const bool in_tile = q < p.TH * p.TW;
        const bool valid = in_tile && (y0 + py) < p.Hout && (x0 + px) < p.Wout;
        pixoff[g] = (in_tile ? (py * p.SI) * p.RP + (px * p.SI) * PS : 0) + hh * 16;
        opix[g] = valid ? ((b * p.Hp + (y0 + py) * p.up + p.upy) * p.Wp + (x0 + px) * p.up + p.upx) : -1;
    }

    // accumulators start at the folded-BN bias of this lane's 16 output channels
    const int cbase = ct * 32 + hh * 16;
    f32x16 acc[NT];
    {
        float bias[16];
#pragma unroll
        for (int r4 = 0; r4 < 4; ++r4) {
            const float4 bv = *reinterpret_cast<const float4*>(p.bias + cbase + r4 * 4);
            bias[r4 * 4 + 0] = bv.x;
            bias[r4 * 4 + 1] = bv.y;
            bias[r4 * 4 + 2] = bv.z;
            bias[r4 * 4 + 3] = bv.w;
        }
#pragma unroll
        for (int g = 0; g < NT; ++g)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[g][r] = bias[r];
    }

    const uint4* ap = p.wpk + (size_t)ct * p.KS * 64 + lane;
    const int total = p.TIH * p.TIW * CPP;
    const int iy0 = y0 * p.SI + p.oy0, ix0 = x0 * p.SI + p.ox0;
    const size_t img = (size_t)b * p.Hin * p.Win;

    for (int ch = 0; ch < p.nchunks; ++ch) {
        // first weight fragments of this channel block: issued before the staging so they land under it
        const uint4* apc = ap + (size_t)ch * KSB * 64;
        uint4 aq[NPF];
#pragma unroll
        for (int k = 0; k < NPF; ++k) aq[k] = apc[k * 64];
        // ---- stage CINB channels of the input tile (+halo) into LDS, zero outside the image ------
        for (int base = 0; base < total; base += NTHR * NLD) {
            uint4 r[NLD];
            int lo[NLD];
#pragma unroll
            for (int j = 0; j < NLD; ++j) {
                const int id = base + j * NTHR + tid;
                r[j] = make_uint4(0, 0, 0, 0);
                lo[j] = -1;
                if (id < total) {
                    const int pix = id / CPP, cg = id - pix * CPP;
                    const int ty = pix / p.TIW, tx = pix - ty * p.TIW;
                    const int iy = iy0 + ty, ix = ix0 + tx;
                    lo[j] = ty * p.RP + tx * PS + cg * 16;
                    if (iy >= 0 && iy < p.Hin && ix >= 0 && ix < p.Win) {
                        const size_t e = (img + (size_t)iy * p.Win + ix) * p.ldi + ch * CINB + cg * 8;
                        r[j] = *reinterpret_cast<const uint4*>(p.in + e * 2);
                    }
                }
            }
#pragma unroll
            for (int j = 0; j < NLD; ++j)
                if (lo[j] >= 0) *reinterpret_cast<uint4*>(smem + lo[j]) = r[j];
        }
        __syncthreads();
        if (ch == 0) RSG_CSTAMP(2);

        // ---- MFMA over the taps of this channel block: one scheduling region per k-step (A loads NPF
        // steps ahead, LDS reads one step ahead), else the compiler sinks the loads next to their uses
        auto boff = [&](int ks) {
            const int t = ks / KPT, c = ks % KPT;
            return (t / TAPW) * p.RP + (t % TAPW) * PS + c * 32;
        };
        uint4 bq[2][NT];
#pragma unroll
        for (int g = 0; g < NT; ++g) bq[0][g] = *reinterpret_cast<const uint4*>(smem + pixoff[g] + boff(0));
#pragma unroll
        for (int ks = 0; ks < KSB; ++ks) {
            const typename DT::vec8 a = __builtin_bit_cast(typename DT::vec8, aq[ks % NPF]);
            if (ks + 1 < KSB) {
#pragma unroll
                for (int g = 0; g < NT; ++g)
                    bq[(ks + 1) & 1][g] = *reinterpret_cast<const uint4*>(smem + pixoff[g] + boff(ks + 1));
            }
            if (ks + NPF < KSB) aq[ks % NPF] = apc[(ks + NPF) * 64];
            // pin: reads issue before the MFMAs (see chain.hip)
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int g = 0; g < NT; ++g)
                acc[g] = DT::mfma(a, __builtin_bit_cast(typename DT::vec8, bq[ks & 1][g]), acc[g]);
            __builtin_amdgcn_sched_barrier(0);
        }
        if (ch + 1 < p.nchunks) __syncthreads();
    }

    RSG_CSTAMP(3);
    if constexpr (TAIL) {
        // ---- fused 1x1 tail (Bottleneck conv2 -> conv3, pose_hrnet.py:93-101): h = ReLU(acc) goes to LDS as 16-bit
        // [tile pixel][32*WM + 8 pad] -- every channel of a pixel comes from this workgroup (grid.y == 1) -- and is the
        // B operand of a second GEMM with the tail's weights; the 64-channel map never goes to HBM.
        constexpr int C1 = 32 * WM, PS2 = 2 * C1 + 16, KS2 = C1 / 16, NP2 = WN * NT;
        __syncthreads();                                 // every wave has finished reading the staged input
#pragma unroll
        for (int g = 0; g < NT; ++g) {
            const int q = (wn * NT + g) * 32 + n;
            if (q < p.TH * p.TW) {
#pragma unroll
                for (int k = 0; k < 2; ++k) {
                    f32x8 f;
#pragma unroll
                    for (int e = 0; e < 8; ++e) f[e] = acc[g][k * 8 + e];
                    *reinterpret_cast<uint4*>(smem + q * PS2 + (wm * 32 + hh * 16 + k * 8) * 2) =
                        __builtin_bit_cast(uint4, DT::relu_pack(f));
                }
            }
        }
        __syncthreads();
        const int ntile2 = p.C2 / 32;
        for (int ct2 = wave; ct2 < ntile2; ct2 += WM * WN) {
            const int cb2 = ct2 * 32 + hh * 16;
            uint4 a2[KS2];
#pragma unroll
            for (int k = 0; k < KS2; ++k) a2[k] = p.wpk2[((size_t)ct2 * KS2 + k) * 64 + lane];
            f32x16 acc2[NP2];
#pragma unroll
            for (int r4 = 0; r4 < 4; ++r4) {
                const float4 bv = *reinterpret_cast<const float4*>(p.bias2 + cb2 + r4 * 4);
#pragma unroll
                for (int j = 0; j < NP2; ++j) {
                    acc2[j][r4 * 4 + 0] = bv.x; acc2[j][r4 * 4 + 1] = bv.y;
                    acc2[j][r4 * 4 + 2] = bv.z; acc2[j][r4 * 4 + 3] = bv.w;
                }
            }
#pragma unroll
            for (int k = 0; k < KS2; ++k) {
                const typename DT::vec8 a = __builtin_bit_cast(typename DT::vec8, a2[k]);
#pragma unroll
                for (int j = 0; j < NP2; ++j) {
                    const int q = j * 32 + n;
                    const uint4 bv = *reinterpret_cast<const uint4*>(smem + (q < p.TH * p.TW ? q : 0) * PS2 + k * 32 + hh * 16);
                    acc2[j] = DT::mfma(a, __builtin_bit_cast(typename DT::vec8, bv), acc2[j]);
                }
            }
            const bool relu2 = (p.relu >> (ct2 & 31)) & 1;
#pragma unroll
            for (int j = 0; j < NP2; ++j) {
                const int q = j * 32 + n;
                const int py = q / p.TW, px = q - py * p.TW;
                if (q >= p.TH * p.TW || y0 + py >= p.Hout || x0 + px >= p.Wout) continue;
                const size_t op = (size_t)(b * p.Hout + y0 + py) * p.Wout + x0 + px;
                float v[16];
#pragma unroll
                for (int r = 0; r < 16; ++r) v[r] = acc2[j][r];
                if (p.res) {
                    const char* rp = p.res + (op * p.ldr + cb2) * 2;
#pragma unroll
                    for (int k = 0; k < 2; ++k) {
                        const uint4 rv = *reinterpret_cast<const uint4*>(rp + k * 16);
                        const f32x8 rf = __builtin_convertvector(__builtin_bit_cast(typename DT::vec8, rv), f32x8);
#pragma unroll
                        for (int e = 0; e < 8; ++e) v[k * 8 + e] += rf[e];
                    }
                }
                char* dst = p.out + (op * p.ldo + cb2) * 2;
#pragma unroll
                for (int k = 0; k < 2; ++k) {
                    f32x8 f;
#pragma unroll
                    for (int e = 0; e < 8; ++e) f[e] = relu2 ? fmaxf(v[k * 8 + e], 0.f) : v[k * 8 + e];
                    *reinterpret_cast<uint4*>(dst + k * 16) = __builtin_bit_cast(uint4, __builtin_convertvector(f, typename DT::vec8));
                }
            }
        }
        RSG_CSTAMP(4);
        return;
    }
    // ---- epilogue: (+ residual) (ReLU) -> 16-bit NHWC and/or fp32 NCHW; the branches are wave-uniform ----
    const bool relu = (p.relu >> (ct & 31)) & 1;
#pragma unroll
    for (int g = 0; g < NT; ++g) {
        if (opix[g] < 0) continue;
        float v[16];
#pragma unroll
        for (int r = 0; r < 16; ++r) v[r] = acc[g][r];
        if (p.res) {
            auto add = [&](const char* base, int ld) {
                const char* rp = base + ((size_t)opix[g] * ld + cbase) * 2;
#pragma unroll
                for (int k = 0; k < 2; ++k) {
                    const uint4 rv = *reinterpret_cast<const uint4*>(rp + k * 16);
                    const f32x8 rf = __builtin_convertvector(__builtin_bit_cast(typename DT::vec8, rv), f32x8);
#pragma unroll
                    for (int e = 0; e < 8; ++e) v[k * 8 + e] += rf[e];
                }
            };
            add(p.res, p.ldr);
            if (p.res2) add(p.res2, p.ldr2);
            if (p.res3) add(p.res3, p.ldr3);
        }
        if (relu) {
#pragma unroll
            for (int r = 0; r < 16; ++r) v[r] = fmaxf(v[r], 0.f);
        }
        if (p.out) {
            char* op = p.out + ((size_t)opix[g] * p.ldo + cbase) * 2;
#pragma unroll
            for (int k = 0; k < 2; ++k) {
                f32x8 f;
#pragma unroll
                for (int e = 0; e < 8; ++e) f[e] = v[k * 8 + e];
                const typename DT::vec8 o = __builtin_convertvector(f, typename DT::vec8);
                if (cbase + k * 8 < p.CoutStore) *reinterpret_cast<uint4*>(op + k * 16) = __builtin_bit_cast(uint4, o);
            }
        }
        if (p.out32) {
            const int plane = p.Hp * p.Wp;
            const int rem = opix[g] - b * plane;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int c = cbase + r;
                if (c < p.Cout) p.out32[((size_t)b * p.Cout + c) * plane + rem] = v[r];
            }
        }
    }
    RSG_CSTAMP(4);
}
