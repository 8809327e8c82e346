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

template <class DT, int CINB, int TAPH, int TAPW, int WM, int WN, int NT, bool TAIL = false, int HEAD = 0, bool NEXT = false, bool KCAT = false,
          class ARGS = ConvArgs, bool RTAIL = false>
__device__ __forceinline__ void conv_body(const ARGS& p, const int block_x, const int block_y, char* smem) {
    constexpr int NTHR = 64 * WM * WN;
    constexpr int KPT = CINB / 16;       // k-steps per filter tap
    constexpr int NTAPS = TAPH * TAPW;
    constexpr int PS = 2 * CINB + 16;    // LDS bytes per pixel (odd number of 16-B slots)
    constexpr int CPP = CINB / 8;        // 16-byte chunks per pixel
#ifndef RSG_CONV_NLD
#define RSG_CONV_NLD 8
#endif
    constexpr int NLD = RSG_CONV_NLD;    // staging loads in flight per thread
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
        // A lane past the end of the tile reads the pixel TH*TW positions back (its product is discarded).  With every idle
        // lane on ONE address, that address shared a bank with one of the live lanes of its 16-lane read group in half of
        // the groups (a 48-pixel tile in 64 lanes: the 8x6 maps); a tile of a multiple of 16 pixels is a whole number of
        // 256-byte bank rows, so the wrapped lanes continue the live lanes' conflict-free slot progression instead.
        const int qw = in_tile ? q : q % (p.TH * p.TW);
        const int pyw = qw / p.TW, pxw = qw - pyw * p.TW;
        pixoff[g] = (pyw * p.SI) * p.RP + pxw * PS + hh * 16;     // columns: see `deint` below
        opix[g] = valid ? ((b * p.Hp + (y0 + py) * p.up + p.upy) * p.Wp + (x0 + px) * p.up + p.upx) : -1;
    }

    if constexpr (HEAD) {
        // ---- fused 1x1 head (Bottleneck conv1, pose_hrnet.py:85-87): the 64-channel input tile (+halo) of the 3x3 conv
        // is computed here from the 256-channel block input and written to LDS where the staging loop would have put
        // it.  The B operand needs no LDS: with the input channels permuted at pack time (logical k = 16j + 8h + e <->
        // channel 128h + 8j + e) a lane's 16 k-steps are 256 contiguous bytes of its pixel.  Positions outside the
        // picture are the 3x3 conv's zero padding, not conv1(0).
        static_assert(CINB == 32 * WM, "the head produces every input channel of the inner conv");
        constexpr int KS0 = 16;
        const int total_px = p.TIH * p.TIW;
        const int ngroups = (total_px + 31) >> 5;
        const int iy00 = y0 * p.SI + p.oy0, ix00 = x0 * p.SI + p.ox0;
        uint4 a0[KS0];
#pragma unroll
        for (int k = 0; k < KS0; ++k) a0[k] = p.wpk0[((size_t)wm * KS0 + k) * 64 + lane];
        float bias0[16];
#pragma unroll
        for (int r4 = 0; r4 < 4; ++r4) {
            const float4 bv = *reinterpret_cast<const float4*>(p.bias0 + wm * 32 + hh * 16 + r4 * 4);
            bias0[r4 * 4 + 0] = bv.x; bias0[r4 * 4 + 1] = bv.y; bias0[r4 * 4 + 2] = bv.z; bias0[r4 * 4 + 3] = bv.w;
        }
        // one pixel group: addresses, the 16 loads of its 256 B per lane, then 16 MFMAs and the LDS write
        struct Grp { int lo; bool live, inside; };
        auto issue = [&](int g, uint4 (&bq0)[KS0]) -> Grp {
            const int id = g * 32 + n;
            const int ty = id / p.TIW, tx = id - ty * p.TIW;
            const int iy = iy00 + ty, ix = ix00 + tx;
            Grp r;
            r.live = id < total_px;
            r.inside = r.live && iy >= 0 && iy < p.Hin && ix >= 0 && ix < p.Win;
            r.lo = ty * p.RP + tx * PS + (wm * 32 + hh * 16) * 2;
            const char* src = p.in0 + (((size_t)b * p.Hin * p.Win + (r.inside ? iy * p.Win + ix : 0)) * p.ldi0 + hh * 128) * 2;
#pragma unroll
            for (int k = 0; k < KS0; ++k) bq0[k] = *reinterpret_cast<const uint4*>(src + k * 16);
            return r;
        };
        auto finish = [&](const Grp& r, const uint4 (&bq0)[KS0]) {
            f32x16 h;
#pragma unroll
            for (int q = 0; q < 16; ++q) h[q] = bias0[q];
#pragma unroll
            for (int k = 0; k < KS0; ++k)
                h = DT::mfma(__builtin_bit_cast(typename DT::vec8, a0[k]), __builtin_bit_cast(typename DT::vec8, bq0[k]), h);
            if (r.live) {
#pragma unroll
                for (int k = 0; k < 2; ++k) {
                    f32x8 f;
#pragma unroll
                    for (int e = 0; e < 8; ++e) f[e] = h[k * 8 + e];
                    uint4 v = __builtin_bit_cast(uint4, DT::relu_pack(f));
                    if (!r.inside) v = make_uint4(0, 0, 0, 0);
                    *reinterpret_cast<uint4*>(smem + r.lo + k * 16) = v;
                }
            }
        };
        if constexpr (HEAD == 1) {
            for (int g = wn; g < ngroups; g += WN) {
                uint4 bq0[KS0];
                const Grp r = issue(g, bq0);
                finish(r, bq0);
            }
        } else {
            // two groups in flight: the loads of the next group are issued before the MFMAs of the current one
            uint4 bA[KS0], bB[KS0];
            Grp rA = issue(wn, bA), rB = rA;
            for (int g = wn; g < ngroups; g += 2 * WN) {
                const bool hasB = g + WN < ngroups;          // wave-uniform
                if (hasB) rB = issue(g + WN, bB);
                __builtin_amdgcn_sched_barrier(0);
                finish(rA, bA);
                __builtin_amdgcn_sched_barrier(0);
                if (g + 2 * WN < ngroups) rA = issue(g + 2 * WN, bA);
                __builtin_amdgcn_sched_barrier(0);
                if (hasB) finish(rB, bB);
                __builtin_amdgcn_sched_barrier(0);
            }
        }
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

    // Stride-2 convs keep the even and the odd input columns of a tile row in two halves (slot = (x & 1) * HW + x / 2):
    // consecutive output pixels then read consecutive slots for every tap, as at stride 1.  With the columns left
    // interleaved the lanes of a read are two pixel pitches apart, a 2-way bank conflict on every B-operand read
    // (SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE = 0.40-0.46 on exactly these kernels, profiles/r01i_sq_summary.json).
    const bool deint = p.SI == 2;
    const int HW = (p.TIW + 1) >> 1;
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
        for (int base = 0; base < (HEAD ? 0 : total); base += NTHR * NLD) {
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
                    lo[j] = ty * p.RP + (deint ? (tx & 1) * HW + (tx >> 1) : tx) * PS + cg * 16;
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
            const int t = ks / KPT, c = ks % KPT, kx = t % TAPW;
            return (t / TAPW) * p.RP + (deint ? (kx & 1) * HW + (kx >> 1) : kx) * PS + c * 32;
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
        if constexpr (NEXT) {
            // ---- pair + the NEXT Bottleneck's conv1 (pose_hrnet.py:85-87 of the following block): the block output is
            // the 1x1 conv1's input pixel for pixel, so it never has to be read back.  The tail runs pixel-major -- a wave
            // owns two pixel groups and walks all output-channel tiles of conv3 -- and each finished tile (after residual
            // and ReLU, packed to 16 bits for the store) is at once the B operand of two k-steps of conv1: an accumulator
            // tile has its column (pixel) on the lane and 16 consecutive channels in the registers, which is an MFMA B
            // fragment when the weights are packed with the matching channel order (plan.py: within every 32 channels the
            // k-step bit and the lane-half bit swap places).  One launch and 100 MB of reads less per Bottleneck.
            static_assert(WM * WN == 4 && NP2 == 8, "four waves, two pixel groups each");
            constexpr int KSN = 16;                            // k-steps of the next conv1: 256 input channels
            // KCAT (layer1.0, pose_hrnet.py:93-101 with its downsample branch): the tail is a 1x1 conv over the channel
            // concat [conv2 output | block input x] with weights [s3*W3 | sd*Wd] -- conv3 and the projected residual in
            // one contraction (K = 128).  The second half of K comes from the tile's own pixels of x, staged behind the
            // conv2 image; there is no residual operand.
            constexpr int KSE = KCAT ? 2 * KS2 : KS2;
            char* const sx2 = smem + NP2 * 32 * PS2;
            if constexpr (KCAT) {
                constexpr int CH = C1 / 8;                     // 16-byte chunks per pixel of x
                uint4 v[CH];
                const int q = tid;                             // 256 threads, 256 pixels: one pixel each
                const int py = q / p.TW, px = q - py * p.TW;
                const bool ok = q < p.TH * p.TW && y0 + py < p.Hout && x0 + px < p.Wout;
                const char* src = p.in2 + ((size_t)((b * p.Hout + y0 + py) * p.Wout + x0 + px) * p.ldi2) * 2;
#pragma unroll
                for (int c = 0; c < CH; ++c) v[c] = ok ? *reinterpret_cast<const uint4*>(src + c * 16) : make_uint4(0, 0, 0, 0);
#pragma unroll
                for (int c = 0; c < CH; ++c) *reinterpret_cast<uint4*>(sx2 + q * PS2 + c * 16) = v[c];
                __syncthreads();
            }
            int op[2];
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const int q = (wave + 4 * j) * 32 + n;
                const int py = q / p.TW, px = q - py * p.TW;
                const bool ok = q < p.TH * p.TW && y0 + py < p.Hout && x0 + px < p.Wout;
                op[j] = ok ? (b * p.Hout + y0 + py) * p.Wout + x0 + px : -1;
            }
            f32x16 t1[2][2];                                    // [conv1 output tile][pixel group]
#pragma unroll
            for (int c = 0; c < 2; ++c) {
                float bn[16];
#pragma unroll
                for (int r4 = 0; r4 < 4; ++r4) {
                    const float4 bv = *reinterpret_cast<const float4*>(p.bias3 + c * 32 + hh * 16 + r4 * 4);
                    bn[r4 * 4 + 0] = bv.x; bn[r4 * 4 + 1] = bv.y; bn[r4 * 4 + 2] = bv.z; bn[r4 * 4 + 3] = bv.w;
                }
#pragma unroll
                for (int j = 0; j < 2; ++j)
#pragma unroll
                    for (int r = 0; r < 16; ++r) t1[c][j][r] = bn[r];
            }
            RSG_CSTAMP(5);
            for (int ct2 = 0; ct2 < ntile2; ++ct2) {
                const int cb2 = ct2 * 32 + hh * 16;
                uint4 a2[KSE], an[2][2];
#pragma unroll
                for (int k = 0; k < KSE; ++k) a2[k] = p.wpk2[((size_t)ct2 * KSE + k) * 64 + lane];
#pragma unroll
                for (int c = 0; c < 2; ++c)
#pragma unroll
                    for (int k = 0; k < 2; ++k) an[c][k] = p.wpk3[((size_t)c * KSN + 2 * ct2 + k) * 64 + lane];
                uint4 rv[2][2];
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    rv[j][0] = rv[j][1] = make_uint4(0, 0, 0, 0);
                    if (p.res && op[j] >= 0) {
                        const char* rp = p.res + ((size_t)op[j] * p.ldr + (size_t)ct2 * p.tsr + hh * 16) * 2;
                        rv[j][0] = *reinterpret_cast<const uint4*>(rp);
                        rv[j][1] = *reinterpret_cast<const uint4*>(rp + 16);
                    }
                }
                float bias2[16];
#pragma unroll
                for (int r4 = 0; r4 < 4; ++r4) {
                    const float4 bv = *reinterpret_cast<const float4*>(p.bias2 + cb2 + r4 * 4);
                    bias2[r4 * 4 + 0] = bv.x; bias2[r4 * 4 + 1] = bv.y; bias2[r4 * 4 + 2] = bv.z; bias2[r4 * 4 + 3] = bv.w;
                }
                f32x16 acc2[2];
#pragma unroll
                for (int j = 0; j < 2; ++j)
#pragma unroll
                    for (int r = 0; r < 16; ++r) acc2[j][r] = bias2[r];
#pragma unroll
                for (int k = 0; k < KSE; ++k) {
                    const typename DT::vec8 a = __builtin_bit_cast(typename DT::vec8, a2[k]);
#pragma unroll
                    for (int j = 0; j < 2; ++j) {
                        const int q = (wave + 4 * j) * 32 + n;
                        const char* img = k < KS2 ? smem : sx2;
                        const uint4 bv = *reinterpret_cast<const uint4*>(img + (q < p.TH * p.TW ? q : 0) * PS2 + (k % KS2) * 32 + hh * 16);
                        acc2[j] = DT::mfma(a, __builtin_bit_cast(typename DT::vec8, bv), acc2[j]);
                    }
                }
                const bool relu2 = (p.relu >> (ct2 & 31)) & 1;
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    uint4 o[2];
#pragma unroll
                    for (int k = 0; k < 2; ++k) {
                        const f32x8 rf = __builtin_convertvector(__builtin_bit_cast(typename DT::vec8, rv[j][k]), f32x8);
                        f32x8 f;
#pragma unroll
                        for (int e = 0; e < 8; ++e) {
                            const float v = acc2[j][k * 8 + e] + rf[e];
                            f[e] = relu2 ? fmaxf(v, 0.f) : v;
                        }
                        o[k] = __builtin_bit_cast(uint4, __builtin_convertvector(f, typename DT::vec8));
                    }
                    if (op[j] >= 0) {
                        char* dst = p.out + ((size_t)op[j] * p.ldo + (size_t)ct2 * p.tso + hh * 16) * 2;
                        *reinterpret_cast<uint4*>(dst) = o[0];
                        *reinterpret_cast<uint4*>(dst + 16) = o[1];
                    }
                    // the stored 16-bit values ARE conv1's input: registers 8k..8k+7 of this lane = channels
                    // cb2 + 8k .. + 7 = the B fragment of k-step 2*ct2 + k under the permuted pack
#pragma unroll
                    for (int c = 0; c < 2; ++c)
#pragma unroll
                        for (int k = 0; k < 2; ++k)
                            t1[c][j] = DT::mfma(__builtin_bit_cast(typename DT::vec8, an[c][k]), __builtin_bit_cast(typename DT::vec8, o[k]), t1[c][j]);
                }
            }
            RSG_CSTAMP(6);
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                if (op[j] < 0) continue;
#pragma unroll
                for (int c = 0; c < 2; ++c) {
                    char* dst = p.out3 + ((size_t)op[j] * p.ldo3 + c * 32 + hh * 16) * 2;
#pragma unroll
                    for (int k = 0; k < 2; ++k) {
                        f32x8 f;
#pragma unroll
                        for (int e = 0; e < 8; ++e) f[e] = t1[c][j][k * 8 + e];
                        *reinterpret_cast<uint4*>(dst + k * 16) = __builtin_bit_cast(uint4, DT::relu_pack(f));
                    }
                }
            }
            RSG_CSTAMP(4);
            return;
        }
        // the pixel groups of the tile in two passes: half the accumulators, and the residual of a pass is requested
        // before its MFMAs so that it lands under them (3 workgroups per CU need <= 168 registers)
        constexpr int NPH = (NP2 + 1) / 2;
        for (int ct2 = wave; ct2 < ntile2; ct2 += WM * WN) {
            const int cb2 = ct2 * 32 + hh * 16;
            uint4 a2[KS2];
#pragma unroll
            for (int k = 0; k < KS2; ++k) a2[k] = p.wpk2[((size_t)ct2 * KS2 + k) * 64 + lane];
            float bias2[16];
#pragma unroll
            for (int r4 = 0; r4 < 4; ++r4) {
                const float4 bv = *reinterpret_cast<const float4*>(p.bias2 + cb2 + r4 * 4);
                bias2[r4 * 4 + 0] = bv.x; bias2[r4 * 4 + 1] = bv.y; bias2[r4 * 4 + 2] = bv.z; bias2[r4 * 4 + 3] = bv.w;
            }
            const bool relu2 = (p.relu >> (ct2 & 31)) & 1;
#pragma unroll
            for (int half = 0; half < 2; ++half) {
                int op[NPH];
                uint4 rv[NPH][2];
#pragma unroll
                for (int j = 0; j < NPH; ++j) {
                    const int q = (half * NPH + j) * 32 + n;
                    const int py = q / p.TW, px = q - py * p.TW;
                    const bool ok = half * NPH + j < NP2 && q < p.TH * p.TW && y0 + py < p.Hout && x0 + px < p.Wout;
                    op[j] = ok ? (b * p.Hout + y0 + py) * p.Wout + x0 + px : -1;
                    rv[j][0] = rv[j][1] = make_uint4(0, 0, 0, 0);
                    if (p.res && ok) {
                        const char* rp = p.res + ((size_t)op[j] * p.ldr + (size_t)ct2 * p.tsr + hh * 16) * 2;
                        rv[j][0] = *reinterpret_cast<const uint4*>(rp);
                        rv[j][1] = *reinterpret_cast<const uint4*>(rp + 16);
                    }
                }
                f32x16 acc2[NPH];
#pragma unroll
                for (int j = 0; j < NPH; ++j)
#pragma unroll
                    for (int r = 0; r < 16; ++r) acc2[j][r] = bias2[r];
#pragma unroll
                for (int k = 0; k < KS2; ++k) {
                    const typename DT::vec8 a = __builtin_bit_cast(typename DT::vec8, a2[k]);
#pragma unroll
                    for (int j = 0; j < NPH; ++j) {
                        const int q = (half * NPH + j) * 32 + n;
                        const uint4 bv = *reinterpret_cast<const uint4*>(smem + (q < p.TH * p.TW ? q : 0) * PS2 + k * 32 + hh * 16);
                        acc2[j] = DT::mfma(a, __builtin_bit_cast(typename DT::vec8, bv), acc2[j]);
                    }
                }
#pragma unroll
                for (int j = 0; j < NPH; ++j) {
                    if (op[j] < 0) continue;
                    char* dst = p.out + ((size_t)op[j] * p.ldo + (size_t)ct2 * p.tso + hh * 16) * 2;
#pragma unroll
                    for (int k = 0; k < 2; ++k) {
                        const f32x8 rf = __builtin_convertvector(__builtin_bit_cast(typename DT::vec8, rv[j][k]), f32x8);
                        f32x8 f;
#pragma unroll
                        for (int e = 0; e < 8; ++e) {
                            const float v = acc2[j][k * 8 + e] + rf[e];
                            f[e] = relu2 ? fmaxf(v, 0.f) : v;
                        }
                        *reinterpret_cast<uint4*>(dst + k * 16) = __builtin_bit_cast(uint4, __builtin_convertvector(f, typename DT::vec8));
                    }
                }
            }
        }
        RSG_CSTAMP(4);
        return;
    }
    if constexpr (RTAIL) {
        // ---- register tail (rsg_conv_tail1x1): this conv has 32 output channels (WM == 1, one tile: a lane pair holds ALL
        // channels of its pixel), and a 1x1 conv over exactly those channels follows (RSGNet: predict_net -> final_layer,
        // pose_rsgnet.py:999-1000).  The ReLU'd accumulators, rounded to 16 bits as the map the two-launch form stores, ARE the
        // B operand of the 1x1's two k-steps (after one half-wave swap of a chunk, below): no LDS, no barrier, and the 32-channel map at the output's resolution (50 MB per 64 crops
        // at 128x96) is neither written nor read back.  p.out32 / p.C2 / p.bias2 / p.wpk2 describe the 1x1 (fp32 NCHW output).
        static_assert(WM == 1 && !TAIL && !HEAD, "one 32-channel output tile per workgroup");
        const bool relu1 = p.relu & 1;
        uint4 a2[2];
        a2[0] = p.wpk2[lane];
        a2[1] = p.wpk2[64 + lane];
        float b2[16];
#pragma unroll
        for (int r4 = 0; r4 < 4; ++r4) {
            const float4 bv = *reinterpret_cast<const float4*>(p.bias2 + hh * 16 + r4 * 4);
            b2[r4 * 4 + 0] = bv.x; b2[r4 * 4 + 1] = bv.y; b2[r4 * 4 + 2] = bv.z; b2[r4 * 4 + 3] = bv.w;
        }
        const int plane = p.Hp * p.Wp;
#pragma unroll
        for (int g = 0; g < NT; ++g) {
            typename DT::vec8 o[2];
#pragma unroll
            for (int k = 0; k < 2; ++k) {
                f32x8 f;
#pragma unroll
                for (int e = 0; e < 8; ++e) f[e] = relu1 ? fmaxf(acc[g][k * 8 + e], 0.f) : acc[g][k * 8 + e];
                o[k] = __builtin_convertvector(f, typename DT::vec8);       // the generic epilogue's rounding
            }
            // lane (n, hh) holds channels 16 hh .. 16 hh + 15 of its pixel; k-step s of the 1x1 wants channels 16 s + 8 hh .. + 7
            // from it: the second chunk of the lower half-wave and the first chunk of the upper one trade places
            // (v_permlane32_swap: lanes 32-63 of the first chunk's registers <-> lanes 0-31 of the second's), then chunk 0 / chunk 1 are the two
            // k-steps' fragments in the 1x1 conv's OWN channel order (plain weight pack, the two launches' summation order)
            u32x4 c0 = __builtin_bit_cast(u32x4, o[0]), c1 = __builtin_bit_cast(u32x4, o[1]);
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                // (the builtin trades the UPPER 32 lanes of its first operand with the LOWER 32 of its second)
                const auto sw2 = __builtin_amdgcn_permlane32_swap(c0[e], c1[e], false, false);
                c0[e] = sw2[0];
                c1[e] = sw2[1];
            }
            f32x16 t;
#pragma unroll
            for (int r = 0; r < 16; ++r) t[r] = b2[r];
            t = DT::mfma(__builtin_bit_cast(typename DT::vec8, a2[0]), __builtin_bit_cast(typename DT::vec8, c0), t);
            t = DT::mfma(__builtin_bit_cast(typename DT::vec8, a2[1]), __builtin_bit_cast(typename DT::vec8, c1), t);
            if (opix[g] < 0) continue;
            const int rem = opix[g] - b * plane;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int c = hh * 16 + r;
                if (c < p.C2) p.out32[((size_t)b * p.C2 + c) * plane + rem] = t[r];
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
