// transition1 of HRNet (pose_hrnet.py:336-375, forward :437-443): the two 3x3 convs that start branches 0 and 1 of stage 2
// read the SAME 256-channel map (100 MB at batch 64) -- conv A at stride 1 (256 -> 32), conv B at stride 2 (256 -> 64).
// As two rsg_conv_igemm launches that map is staged twice (52 + 34 us, both bound by the read).  Here one workgroup stages
// a 16x16 tile (+ halo) per 64-channel block once and runs both contractions on it: conv A's 256 output pixels are two
// pixel groups per wave, conv B's 8x8 outputs x 64 channels are four (pixel group, channel tile) jobs, one per wave, so
// every wave issues the same number of MFMAs.  Conv B reads its B fragments at a two-pixel stride from the plain image
// (a 2-way bank conflict on a third of the LDS reads; conv_igemm's stride-2 kernel de-interleaves the columns instead).
// Same k order, same accumulation, same epilogue as rsg_conv_igemm: the results are bit-identical to the two launches.
#include "conv_args.h"

struct DualArgs {
    const char* in;
    const uint4* wA; const float* bA; char* outA;
    const uint4* wB; const float* bB; char* outB;
    int B, H, W, ldi;
    int tsi;        // elements between the 32-channel tiles of `in` (32 = dense NHWC; batch*H*W*32 with ldi = 32: channel-tile planes)
    int ldoA, storeA, ldoB;
    int nchunks, tiles_x, tiles_y, RP;
    int reluA, reluB;
};

// CTA / CTB: 32-channel output tiles of conv A / conv B -- (1, 2) for HRNet-W32 (256 -> 32 and 256 -> 64), (2, 3) for
// HRNet-W48 (256 -> 48 padded to 64, and 256 -> 96).  Conv A: a wave owns two pixel groups and all CTA tiles of them (one B
// fragment feeds CTA MFMAs); conv B: 2 x CTB (pixel group, tile) jobs dealt to the four waves, job j = wave + 4 k.
// NW = 2 (round 5, HRNet-W32's transition): TWO waves per workgroup, four conv-A pixel groups each and both pixel groups of
// ONE conv-B tile, so that a weight fragment feeds four / two MFMAs instead of two / one.  The launch is bound by the weight
// fragments its waves fetch (885 MB per launch through the CUs' vector-memory paths at ~25 B/clk per CU = its 59 us; the map it
// reads is 127 MB): half the fragments per pixel, the same three workgroups per CU.  Same per-pixel arithmetic: bit-identical.
template <class DT, int CTA, int CTB, int NW = 4>
__global__ void __launch_bounds__(64 * NW) conv_dual_kernel(const DualArgs p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int TH = 16, TW = 16, TIH = 18, TIW = 18, CINB = 64, PS = 2 * CINB + 16, CPP = CINB / 8, KPT = 4, KSB = 36, NT = 8 / NW;
    constexpr int NLD = 8, NPF = 6, NJB = (2 * CTB + NW - 1) / NW, NTHR = 64 * NW;
    constexpr bool SHARE_B = NW == 2 && CTB == 2;         // a wave's conv-B jobs are the two pixel groups of one tile
    static_assert(NW == 4 || SHARE_B, "two waves: the conv-B jobs of a wave must share their tile");
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int n = lane & 31, hh = lane >> 5;
    int bx = rsg_xcd_tile(blockIdx.x, gridDim.x);
    const int tix = bx % p.tiles_x;
    bx /= p.tiles_x;
    const int tiy = bx % p.tiles_y, b = bx / p.tiles_y;
    const int y0 = tiy * TH, x0 = tix * TW;
    const int H2 = p.H >> 1, W2 = p.W >> 1;

    int pixA[NT], opixA[NT];
#pragma unroll
    for (int g = 0; g < NT; ++g) {
        const int q = (wave * NT + g) * 32 + n, py = q / TW, px = q - py * TW;
        pixA[g] = py * p.RP + px * PS + hh * 16;
        opixA[g] = (y0 + py < p.H && x0 + px < p.W) ? (b * p.H + y0 + py) * p.W + x0 + px : -1;
    }
    int pixB[NJB], opixB[NJB], ctB[NJB];
#pragma unroll
    for (int k = 0; k < NJB; ++k) {
        const int j = wave + NW * k;                          // wave-uniform
        ctB[k] = j < 2 * CTB ? j % CTB : -1;
        const int q = (j < 2 * CTB ? j / CTB : 0) * 32 + n, py = q / (TW / 2), px = q - py * (TW / 2);
        pixB[k] = (2 * py) * p.RP + (2 * px) * PS + hh * 16;
        opixB[k] = (j < 2 * CTB && y0 / 2 + py < H2 && x0 / 2 + px < W2) ? (b * H2 + y0 / 2 + py) * W2 + x0 / 2 + px : -1;
    }
    auto bias16 = [&](const float* bp, f32x16& acc) {
#pragma unroll
        for (int r4 = 0; r4 < 4; ++r4) {
            const float4 bv = *reinterpret_cast<const float4*>(bp + r4 * 4);
            acc[r4 * 4 + 0] = bv.x; acc[r4 * 4 + 1] = bv.y; acc[r4 * 4 + 2] = bv.z; acc[r4 * 4 + 3] = bv.w;
        }
    };
    f32x16 accA[NT][CTA], accB[NJB];
#pragma unroll
    for (int c = 0; c < CTA; ++c) {
        bias16(p.bA + c * 32 + hh * 16, accA[0][c]);
#pragma unroll
        for (int g = 1; g < NT; ++g) accA[g][c] = accA[0][c];
    }
#pragma unroll
    for (int k = 0; k < NJB; ++k) bias16(p.bB + (ctB[k] < 0 ? 0 : ctB[k]) * 32 + hh * 16, accB[k]);

    const int KS = p.nchunks * KSB;
    const int total = TIH * TIW * CPP;
    const int iy0 = y0 - 1, ix0 = x0 - 1;
    const size_t img = (size_t)b * p.H * p.W;
    auto boff = [&](int ks) {
        const int t = ks / KPT, c = ks % KPT;
        return (t / 3) * p.RP + (t % 3) * PS + c * 32;
    };
    for (int ch = 0; ch < p.nchunks; ++ch) {
        // conv A's fragments in (k-step, tile) order: flattened step f = ks * CTA + c
        auto fragA = [&](int f) { return p.wA[((size_t)(f % CTA) * KS + (size_t)ch * KSB + f / CTA) * 64 + lane]; };
        uint4 aq[NPF];
#pragma unroll
        for (int f = 0; f < NPF; ++f) aq[f] = fragA(f);
        // ---- stage 64 channels of the input tile (+halo), zero outside the picture
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
                    const int ty = pix / TIW, tx = pix - ty * TIW;
                    const int iy = iy0 + ty, ix = ix0 + tx;
                    lo[j] = ty * p.RP + tx * PS + cg * 16;
                    if (iy >= 0 && iy < p.H && ix >= 0 && ix < p.W)
                        r[j] = *reinterpret_cast<const uint4*>(p.in + ((img + (size_t)iy * p.W + ix) * p.ldi + (size_t)(ch * (CINB / 32) + (cg >> 2)) * p.tsi + (cg & 3) * 8) * 2);
                }
            }
#pragma unroll
            for (int j = 0; j < NLD; ++j)
                if (lo[j] >= 0) *reinterpret_cast<uint4*>(smem + lo[j]) = r[j];
        }
        __syncthreads();
        // ---- conv A: one scheduling region per k-step, A loads NPF steps ahead, LDS reads one step ahead
        {
            uint4 bq[2][NT];
#pragma unroll
            for (int g = 0; g < NT; ++g) bq[0][g] = *reinterpret_cast<const uint4*>(smem + pixA[g] + boff(0));
#pragma unroll
            for (int ks = 0; ks < KSB; ++ks) {
                typename DT::vec8 a[CTA];
#pragma unroll
                for (int c = 0; c < CTA; ++c) a[c] = __builtin_bit_cast(typename DT::vec8, aq[(ks * CTA + c) % NPF]);
                if (ks + 1 < KSB) {
#pragma unroll
                    for (int g = 0; g < NT; ++g) bq[(ks + 1) & 1][g] = *reinterpret_cast<const uint4*>(smem + pixA[g] + boff(ks + 1));
                }
#pragma unroll
                for (int c = 0; c < CTA; ++c)
                    if (ks * CTA + c + NPF < KSB * CTA) aq[(ks * CTA + c) % NPF] = fragA(ks * CTA + c + NPF);
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int g = 0; g < NT; ++g)
#pragma unroll
                    for (int c = 0; c < CTA; ++c)
                        accA[g][c] = DT::mfma(a[c], __builtin_bit_cast(typename DT::vec8, bq[ks & 1][g]), accA[g][c]);
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        // ---- conv B on the same staged tile: both pixel groups of the wave's tile per fragment (two waves), or job by job
        if constexpr (SHARE_B) {
            const uint4* aB = p.wB + ((size_t)ctB[0] * KS + (size_t)ch * KSB) * 64 + lane;
#pragma unroll
            for (int f = 0; f < NPF; ++f) aq[f] = aB[f * 64];
            uint4 bq[2][2];
#pragma unroll
            for (int k = 0; k < 2; ++k) bq[0][k] = *reinterpret_cast<const uint4*>(smem + pixB[k] + boff(0));
#pragma unroll
            for (int ks = 0; ks < KSB; ++ks) {
                const typename DT::vec8 a = __builtin_bit_cast(typename DT::vec8, aq[ks % NPF]);
                if (ks + 1 < KSB) {
#pragma unroll
                    for (int k = 0; k < 2; ++k) bq[(ks + 1) & 1][k] = *reinterpret_cast<const uint4*>(smem + pixB[k] + boff(ks + 1));
                }
                if (ks + NPF < KSB) aq[ks % NPF] = aB[(ks + NPF) * 64];
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int k = 0; k < 2; ++k) accB[k] = DT::mfma(a, __builtin_bit_cast(typename DT::vec8, bq[ks & 1][k]), accB[k]);
                __builtin_amdgcn_sched_barrier(0);
            }
        } else
#pragma unroll
        for (int k = 0; k < NJB; ++k) {
            if (ctB[k] < 0) continue;                         // wave-uniform
            const uint4* aB = p.wB + ((size_t)ctB[k] * KS + (size_t)ch * KSB) * 64 + lane;
#pragma unroll
            for (int f = 0; f < NPF; ++f) aq[f] = aB[f * 64];
            uint4 bq[2];
            bq[0] = *reinterpret_cast<const uint4*>(smem + pixB[k] + boff(0));
#pragma unroll
            for (int ks = 0; ks < KSB; ++ks) {
                const typename DT::vec8 a = __builtin_bit_cast(typename DT::vec8, aq[ks % NPF]);
                if (ks + 1 < KSB) bq[(ks + 1) & 1] = *reinterpret_cast<const uint4*>(smem + pixB[k] + boff(ks + 1));
                if (ks + NPF < KSB) aq[ks % NPF] = aB[(ks + NPF) * 64];
                __builtin_amdgcn_sched_barrier(0);
                accB[k] = DT::mfma(a, __builtin_bit_cast(typename DT::vec8, bq[ks & 1]), accB[k]);
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        if (ch + 1 < p.nchunks) __syncthreads();
    }
    // ---- epilogues: ReLU -> 16-bit NHWC
    auto store = [&](const f32x16& acc, char* out, int opix, int ldo, int cbase, int nstore, int relu) {
        if (opix < 0) return;
        char* op = out + ((size_t)opix * ldo + cbase) * 2;
#pragma unroll
        for (int k = 0; k < 2; ++k) {
            f32x8 f;
#pragma unroll
            for (int e = 0; e < 8; ++e) f[e] = relu ? fmaxf(acc[k * 8 + e], 0.f) : acc[k * 8 + e];
            const typename DT::vec8 o = __builtin_convertvector(f, typename DT::vec8);
            if (cbase + k * 8 < nstore) *reinterpret_cast<uint4*>(op + k * 16) = __builtin_bit_cast(uint4, o);
        }
    };
#pragma unroll
    for (int g = 0; g < NT; ++g)
#pragma unroll
        for (int c = 0; c < CTA; ++c) store(accA[g][c], p.outA, opixA[g], p.ldoA, c * 32 + hh * 16, p.storeA, p.reluA);
#pragma unroll
    for (int k = 0; k < NJB; ++k)
        if (ctB[k] >= 0) store(accB[k], p.outB, opixB[k], p.ldoB, ctB[k] * 32 + hh * 16, 32 * CTB, p.reluB);
}

// a: the stride-1 conv (3x3, pad 1, cin a multiple of 64, 16-bit NHWC output, no residual); the stride-2 conv reads the same
// input: wB/biasB packed for (cin, cout_b, 3x3), outB [batch, h/2, w/2, out_ld_b] (h, w even).  (cout_a <= 32, cout_b = 64)
// or (cout_a <= 64, cout_b = 96).
int g_dual_waves = 2;                  // rsg_tune("dual_waves", 2|4): waves per workgroup of the 256 -> 32 / 64 form

extern "C" int rsg_conv_dual(const rsg_conv_desc* a, const void* wB_packed, const float* biasB, void* outB16, int32_t out_ld_b,
                             int32_t cout_b, int32_t relu_b, void* stream) {
    if (!a || !a->in || !a->w_packed || !a->bias || !a->out16 || !wB_packed || !biasB || !outB16) RSG_FAIL("conv_dual: null pointer");
    if (a->kh != 3 || a->kw != 3 || a->stride != 1 || a->pad_y != 1 || a->pad_x != 1 || a->in_c < 64 || a->in_c % 64 || a->out_c < 1 ||
        a->residual || a->out_nchw_f32 || a->up != 1 || a->out_h != a->in_h || a->out_w != a->in_w)
        RSG_FAIL("conv_dual: conv A must be 3x3 stride 1 pad 1 with cin a multiple of 64");
    const int cta = ceil_div(a->out_c, 32);
    if (!((cta == 1 && cout_b == 64) || (cta == 2 && cout_b == 96)))
        RSG_FAIL("conv_dual: output channels %d / %d (built: <= 32 with 64, <= 64 with 96)", a->out_c, cout_b);
    if ((a->in_h | a->in_w) & 1) RSG_FAIL("conv_dual: the map must have even height and width");
    const int store = a->out_c_store ? a->out_c_store : 32 * cta;
    const bool planes_in = (a->layout & RSG_LAYOUT_IN_PLANES) != 0;
    if ((planes_in ? a->in_ld != 32 : a->in_ld < a->in_c) || (a->in_ld & 7) || a->out_ld < store || (a->out_ld & 7) || store % 8 || store > 32 * cta || out_ld_b < cout_b || (out_ld_b & 7))
        RSG_FAIL("conv_dual: strides");
    if (((uintptr_t)a->in | (uintptr_t)a->w_packed | (uintptr_t)a->bias | (uintptr_t)a->out16 | (uintptr_t)wB_packed | (uintptr_t)biasB | (uintptr_t)outB16) & 15)
        RSG_FAIL("conv_dual: pointers must be 16-byte aligned");
    if ((long long)a->batch * a->in_h * a->in_w * (planes_in ? a->in_c : a->in_ld) * 2 >= (1LL << 31)) RSG_FAIL("conv_dual: map too large for 32-bit offsets");
    DualArgs p;
    p.in = (const char*)a->in; p.wA = (const uint4*)a->w_packed; p.bA = a->bias; p.outA = (char*)a->out16;
    p.wB = (const uint4*)wB_packed; p.bB = biasB; p.outB = (char*)outB16;
    p.B = a->batch; p.H = a->in_h; p.W = a->in_w; p.ldi = a->in_ld;
    p.tsi = planes_in ? a->batch * a->in_h * a->in_w * 32 : 32;
    p.ldoA = a->out_ld; p.storeA = store; p.ldoB = out_ld_b;
    p.nchunks = a->in_c / 64;
    p.tiles_x = ceil_div(a->in_w, 16); p.tiles_y = ceil_div(a->in_h, 16);
    const int PS = 144, base = 18 * PS;
    p.RP = base + ((0 - base) % 256 + 256) % 256;          // == 16 * PS (mod 256): see conv_igemm.hip, tile_geometry
    p.reluA = a->relu ? 1 : 0; p.reluB = relu_b ? 1 : 0;
    const int lds = 18 * p.RP;
    const dim3 grid(a->batch * p.tiles_x * p.tiles_y);
    hipStream_t s = (hipStream_t)stream;
    if (cta == 1 && g_dual_waves == 2) {
        if (a->dtype == RSG_BF16) RSG_LAUNCH((conv_dual_kernel<BF16, 1, 2, 2>), grid, dim3(128), lds, s, p);
        else RSG_LAUNCH((conv_dual_kernel<F16, 1, 2, 2>), grid, dim3(128), lds, s, p);
    } else if (cta == 1) {
        if (a->dtype == RSG_BF16) RSG_LAUNCH((conv_dual_kernel<BF16, 1, 2>), grid, dim3(256), lds, s, p);
        else RSG_LAUNCH((conv_dual_kernel<F16, 1, 2>), grid, dim3(256), lds, s, p);
    } else {
        if (a->dtype == RSG_BF16) RSG_LAUNCH((conv_dual_kernel<BF16, 2, 3>), grid, dim3(256), lds, s, p);
        else RSG_LAUNCH((conv_dual_kernel<F16, 2, 3>), grid, dim3(256), lds, s, p);
    }
    RSG_CHECK_LAUNCH("conv_dual");
    return 0;
}
