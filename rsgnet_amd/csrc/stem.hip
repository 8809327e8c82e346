// Fused HRNet stem for gfx950 (MI355X):  conv1 (3x3 s2, 3 -> 64) + BN + ReLU -> conv2 (3x3 s2, 64 -> 64) + BN + ReLU
// (pose_hrnet.py:285-291,429-434; pose_rsgnet.py:612-618) in ONE launch.  As two launches the 64-channel half-resolution
// map between them (100 MB at batch 64) is written and read back: 78 us for 0.27 GFLOP/img.
//
// conv1 runs as a 2x2 stride-1 conv over the space-to-depth input (rsg_nchw_f32_to_s2d_nhwc16: 16 channels per
// half-resolution pixel, 12 used; weights rearranged by the caller exactly as for the unfused path).  A workgroup owns a
// TH2 x TW2 tile of conv2's output: it stages the (2*TH2+2) x (2*TW2+2) window of the s2d input, computes conv1 on the
// (2*TH2+1) x (2*TW2+1) positions conv2 needs into LDS (16-bit, zero where conv2's padding lies), then conv2 from there.
// conv1 positions on tile borders are recomputed by the neighbour (9/8 x 25/24 of the conv1 work, 1 % of the stem).
#include "common.h"

struct StemArgs {
    const char* xs;            // s2d input NHWC16 [B, H1, W1, 16], or null:
    const float* x32;          // the picture itself, fp32 NCHW [B, 3, 2*H1, 2*W1] (space-to-depth done while staging)
    const uint4* w1;           // conv1 as 2x2 conv, packed (CINB 16: KS = 4), 64 outputs
    const float* b1;
    const uint4* w2;           // conv2 3x3, packed (CINB 64: KS = 36), 64 outputs
    const float* b2;
    char* out;                 // NHWC16 [B, H2, W2, ldo] (+ channel offset folded into the pointer)
    int B, H1, W1, H2, W2, ldo;
    int tiles_x, tiles_y;
    // optional tail: a 1x1 conv 64 -> 64 (+BN, ReLU) on the stem's output in the same launch (layer1.0.conv1,
    // pose_hrnet.py:85-87): its input pixel is complete in this workgroup, so the 25 MB map is not read back
    const uint4* w3;           // packed (CINB 64: KS = 4), or null
    const float* b3;
    char* out3;                // NHWC16 [B, H2, W2, ldo3]
    int ldo3;
};

constexpr int stem_pitch_s(int CS, int C1) { return CS * 48 + ((C1 * 48 - CS * 48) % 256 + 256) % 256; }
constexpr int stem_pitch_1(int C1, int TW2) { return C1 * 144 + ((TW2 * 72 - C1 * 144) % 128 + 128) % 128; }

template <class DT, int TH2, int TW2>
__global__ void __launch_bounds__(256, 3) stem_kernel(const StemArgs p) {
    constexpr int R1 = 2 * TH2 + 1, C1 = 2 * TW2 + 1;      // conv1 positions held
    constexpr int RS = R1 + 1, CS = C1 + 1;                // s2d pixels staged
    constexpr int PSS = 48;                                // s2d pixel: 16 ch * 2 B + 16 pad
    constexpr int PS1 = 144;                               // conv1 pixel: 64 ch * 2 B + 16 pad
    // Row pitches padded so that a 32-position group which wraps to the next row continues the same bank progression (as the
    // chain and conv kernels do): conv1 walks the s2d window C1 positions per row (pitch == C1 * PSS mod 256), conv2 walks the
    // conv1 window TW2 outputs per row, two conv1 rows apart (2 * pitch == TW2 * PS1 mod 256).  Dense pitches cost 42 % of the
    // LDS cycles in bank conflicts (SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE, profiles/r03p_sq_summary.json).
    constexpr int RPS = stem_pitch_s(CS, C1), RP1 = stem_pitch_1(C1, TW2);
    // conv2 has stride 2: the conv1 window keeps its even and odd columns in two halves of a row (slot = (c & 1) * HW1 +
    // c / 2), so that consecutive output pixels read consecutive slots (no 2-way LDS bank conflict; cf. conv_body `deint`)
    constexpr int HW1 = (C1 + 1) / 2;
    constexpr int NP1 = (R1 * C1 + 31) / 32;               // 32-pixel groups of the conv1 window
    constexpr int NT1 = (NP1 + 1) / 2;                     // per wave column (WN = 2)
    constexpr int NP2 = (TH2 * TW2 + 31) / 32, NT2 = (NP2 + 1) / 2;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* const sS = smem;                                 // s2d window
    char* const sH = smem + RS * RPS;                      // conv1 window

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 1, wn = wave & 1;               // 2 output-channel tiles x 2 pixel columns
    const int n = lane & 31, hh = lane >> 5;
    int bx = rsg_xcd_tile(blockIdx.x, gridDim.x);          // the tiles of a picture on one XCD (shared halo in its L2)
    const int tix = bx % p.tiles_x;
    bx /= p.tiles_x;
    const int tiy = bx % p.tiles_y;
    const int b = bx / p.tiles_y;
    const int y2 = tiy * TH2, x2 = tix * TW2;              // first conv2 output of the tile
    const int y1 = 2 * y2 - 1, x1 = 2 * x2 - 1;            // first conv1 position held (may be -1: conv2's padding)
    const int ys = y1 - 1, xs0 = x1 - 1;                   // first s2d pixel staged (conv1's own padding at -1)

    // ---- stage the s2d window: 2 chunks of 16 B per pixel ------------------------------------------------------------
    if (p.x32) {
        // straight from the fp32 NCHW picture: s2d pixel (Y, X) = in[:, 2Y..2Y+1, 2X..2X+1], channel (py*2+px)*3 + k
        const int H0 = 2 * p.H1, W0 = 2 * p.W1;
        for (int pix = tid; pix < RS * CS; pix += 256) {
            const int r = pix / CS, c = pix - r * CS;
            const int y = ys + r, x = xs0 + c;
            typename DT::elem v[16];
#pragma unroll
            for (int k = 0; k < 16; ++k) v[k] = (typename DT::elem)0.f;
            if (y >= 0 && y < p.H1 && x >= 0 && x < p.W1) {
#pragma unroll
                for (int k = 0; k < 3; ++k) {
                    const float* pl = p.x32 + (((size_t)b * 3 + k) * H0 + 2 * y) * W0 + 2 * x;
                    const float2 r0 = *reinterpret_cast<const float2*>(pl);
                    const float2 r1 = *reinterpret_cast<const float2*>(pl + W0);
                    v[0 * 3 + k] = (typename DT::elem)r0.x;
                    v[1 * 3 + k] = (typename DT::elem)r0.y;
                    v[2 * 3 + k] = (typename DT::elem)r1.x;
                    v[3 * 3 + k] = (typename DT::elem)r1.y;
                }
            }
            uint4 o[2];
            __builtin_memcpy(o, v, 32);
            *reinterpret_cast<uint4*>(sS + r * RPS + c * PSS) = o[0];
            *reinterpret_cast<uint4*>(sS + r * RPS + c * PSS + 16) = o[1];
        }
    } else {
        for (int id = tid; id < RS * CS * 2; id += 256) {
            const int pix = id >> 1, cg = id & 1;
            const int r = pix / CS, c = pix - r * CS;
            const int y = ys + r, x = xs0 + c;
            uint4 v = make_uint4(0, 0, 0, 0);
            if (y >= 0 && y < p.H1 && x >= 0 && x < p.W1)
                v = *reinterpret_cast<const uint4*>(p.xs + ((((size_t)b * p.H1 + y) * p.W1 + x) * 16 + cg * 8) * 2);
            *reinterpret_cast<uint4*>(sS + r * RPS + c * PSS + cg * 16) = v;
        }
    }
    const int cbase = wm * 32 + hh * 16;
    float4 bv1[4], bv2[4];
#pragma unroll
    for (int r4 = 0; r4 < 4; ++r4) {
        bv1[r4] = *reinterpret_cast<const float4*>(p.b1 + cbase + r4 * 4);
        bv2[r4] = *reinterpret_cast<const float4*>(p.b2 + cbase + r4 * 4);
    }
    uint4 a1[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) a1[k] = p.w1[((size_t)wm * 4 + k) * 64 + lane];
    __syncthreads();

    // ---- conv1 on the R1 x C1 window: tap (a, b) of the 2x2 conv reads s2d pixel (r + a, c + b) --------------------------
    {
        f32x16 acc[NT1];
        int so[NT1];
#pragma unroll
        for (int g = 0; g < NT1; ++g) {
            const int q = (wn * NT1 + g) * 32 + n;
            const int qq = q < R1 * C1 ? q : 0;
            const int r = qq / C1, c = qq - r * C1;
            so[g] = r * RPS + c * PSS + hh * 16;
#pragma unroll
            for (int r4 = 0; r4 < 4; ++r4) {
                acc[g][r4 * 4 + 0] = bv1[r4].x; acc[g][r4 * 4 + 1] = bv1[r4].y;
                acc[g][r4 * 4 + 2] = bv1[r4].z; acc[g][r4 * 4 + 3] = bv1[r4].w;
            }
        }
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const typename DT::vec8 a = __builtin_bit_cast(typename DT::vec8, a1[k]);
#pragma unroll
            for (int g = 0; g < NT1; ++g) {
                const uint4 bq = *reinterpret_cast<const uint4*>(sS + so[g] + (k >> 1) * RPS + (k & 1) * PSS);
                acc[g] = DT::mfma(a, __builtin_bit_cast(typename DT::vec8, bq), acc[g]);
            }
        }
#pragma unroll
        for (int g = 0; g < NT1; ++g) {
            const int q = (wn * NT1 + g) * 32 + n;
            if (q >= R1 * C1) continue;
            const int r = q / C1, c = q - r * C1;
            const bool real = (y1 + r) >= 0 && (y1 + r) < p.H1 && (x1 + c) >= 0 && (x1 + c) < p.W1;   // else conv2's zero padding
#pragma unroll
            for (int k = 0; k < 2; ++k) {
                f32x8 f;
#pragma unroll
                for (int e = 0; e < 8; ++e) f[e] = real ? acc[g][k * 8 + e] : 0.f;
                *reinterpret_cast<uint4*>(sH + r * RP1 + ((c & 1) * HW1 + (c >> 1)) * PS1 + (cbase + k * 8) * 2) =
                    __builtin_bit_cast(uint4, DT::relu_pack(f));
            }
        }
    }
    __syncthreads();

    // ---- conv2 (3x3 stride 2) from the conv1 window: output (py, px) reads window pixels (2py + ky, 2px + kx) -----------
    f32x16 acc2[NT2];
    int ho[NT2];
#pragma unroll
    for (int g = 0; g < NT2; ++g) {
        const int q = (wn * NT2 + g) * 32 + n;
        const int qq = q < TH2 * TW2 ? q : 0;
        const int py = qq / TW2, px = qq - py * TW2;
        ho[g] = (2 * py) * RP1 + px * PS1 + hh * 16;
#pragma unroll
        for (int r4 = 0; r4 < 4; ++r4) {
            acc2[g][r4 * 4 + 0] = bv2[r4].x; acc2[g][r4 * 4 + 1] = bv2[r4].y;
            acc2[g][r4 * 4 + 2] = bv2[r4].z; acc2[g][r4 * 4 + 3] = bv2[r4].w;
        }
    }
    constexpr int NPF = 6;
    const uint4* ap = p.w2 + (size_t)wm * 36 * 64 + lane;
    uint4 aq[NPF];
#pragma unroll
    for (int k = 0; k < NPF; ++k) aq[k] = ap[k * 64];
    // B fragments NB - 1 k-steps ahead: one MFMA per step cannot cover an LDS read issued one step earlier
    constexpr int NB = 4;
    auto hoff = [&](int ks) {
        const int t = ks >> 2, c = ks & 3, kx = t % 3;
        return (t / 3) * RP1 + ((kx & 1) * HW1 + (kx >> 1)) * PS1 + c * 32;
    };
    uint4 bq[NB][NT2];
#pragma unroll
    for (int s0 = 0; s0 < NB - 1; ++s0)
#pragma unroll
        for (int g = 0; g < NT2; ++g) bq[s0][g] = *reinterpret_cast<const uint4*>(sH + ho[g] + hoff(s0));
#pragma unroll
    for (int ks = 0; ks < 36; ++ks) {
        const typename DT::vec8 a = __builtin_bit_cast(typename DT::vec8, aq[ks % NPF]);
        if (ks + NB - 1 < 36) {
#pragma unroll
            for (int g = 0; g < NT2; ++g)
                bq[(ks + NB - 1) % NB][g] = *reinterpret_cast<const uint4*>(sH + ho[g] + hoff(ks + NB - 1));
        }
        if (ks + NPF < 36) aq[ks % NPF] = ap[(ks + NPF) * 64];
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int g = 0; g < NT2; ++g) acc2[g] = DT::mfma(a, __builtin_bit_cast(typename DT::vec8, bq[ks % NB][g]), acc2[g]);
        __builtin_amdgcn_sched_barrier(0);
    }
    // the s2d window is dead since the barrier behind conv1: its space takes the output tile for the tail, [pixel slot][64 ch]
    char* const sT = sS;
    static_assert(2 * NT2 * 32 * PS1 <= RS * RPS, "the tail's tile fits the s2d window");
#pragma unroll
    for (int g = 0; g < NT2; ++g) {
        const int q = (wn * NT2 + g) * 32 + n;
        const int py = q / TW2, px = q - py * TW2;
        const bool ok = q < TH2 * TW2 && y2 + py < p.H2 && x2 + px < p.W2;
        char* dst = p.out + ((((size_t)b * p.H2 + y2 + py) * p.W2 + x2 + px) * p.ldo + cbase) * 2;
#pragma unroll
        for (int k = 0; k < 2; ++k) {
            f32x8 f;
#pragma unroll
            for (int e = 0; e < 8; ++e) f[e] = acc2[g][k * 8 + e];
            const uint4 v = __builtin_bit_cast(uint4, DT::relu_pack(f));
            if (ok) *reinterpret_cast<uint4*>(dst + k * 16) = v;
            if (p.w3) *reinterpret_cast<uint4*>(sT + q * PS1 + (cbase + k * 8) * 2) = v;
        }
    }
    if (!p.w3) return;
    // ---- tail: 1x1 conv over the 64 channels of the tile's pixels (K = 64: 4 k-steps), BN folded, ReLU -------------------
    uint4 a3[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) a3[k] = p.w3[((size_t)wm * 4 + k) * 64 + lane];
    float4 bv3[4];
#pragma unroll
    for (int r4 = 0; r4 < 4; ++r4) bv3[r4] = *reinterpret_cast<const float4*>(p.b3 + cbase + r4 * 4);
    __syncthreads();
#pragma unroll
    for (int g = 0; g < NT2; ++g) {
        const int q = (wn * NT2 + g) * 32 + n;
        f32x16 acc3;
#pragma unroll
        for (int r4 = 0; r4 < 4; ++r4) {
            acc3[r4 * 4 + 0] = bv3[r4].x; acc3[r4 * 4 + 1] = bv3[r4].y; acc3[r4 * 4 + 2] = bv3[r4].z; acc3[r4 * 4 + 3] = bv3[r4].w;
        }
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const uint4 bq3 = *reinterpret_cast<const uint4*>(sT + q * PS1 + k * 32 + hh * 16);
            acc3 = DT::mfma(__builtin_bit_cast(typename DT::vec8, a3[k]), __builtin_bit_cast(typename DT::vec8, bq3), acc3);
        }
        const int py = q / TW2, px = q - py * TW2;
        if (q >= TH2 * TW2 || y2 + py >= p.H2 || x2 + px >= p.W2) continue;
        char* dst = p.out3 + ((((size_t)b * p.H2 + y2 + py) * p.W2 + x2 + px) * p.ldo3 + cbase) * 2;
#pragma unroll
        for (int k = 0; k < 2; ++k) {
            f32x8 f;
#pragma unroll
            for (int e = 0; e < 8; ++e) f[e] = acc3[k * 8 + e];
            *reinterpret_cast<uint4*>(dst + k * 16) = __builtin_bit_cast(uint4, DT::relu_pack(f));
        }
    }
}

// ---------------------------------------------------------------------------------------------------------------------------
// Persistent form (round 5).  What bounds stem_kernel is not its arithmetic (mfma_busy 20 %) but the weight fragments every
// workgroup fetches again: 180 KB per 48-pixel tile x 16 tiles per CU = 2.9 MB per CU through the vector-memory path at the
// ~25 B/clk a CU gets out of the L2 (profiles/r05_small_experiments.txt).  Here a workgroup (four waves, as before: 2
// output-channel tiles x 2 pixel columns) stays on its CU and walks tiles of 4 x 16 conv2 outputs; the 36 fragments of conv2's
// weights that its wave multiplies with stay in 144 REGISTERS for the whole launch (two workgroups per CU = two waves per
// SIMD = 256 registers each), conv1's and the tail's 4 + 4 fragments are re-read per tile (L2 hits, requested a phase early).
// The next tile's window of the picture is requested in front of conv1 and written to LDS behind conv2 (the window buffer is
// free from the barrier that ends conv1), so that its memory latency runs under both convs: two barriers per tile.  Per-pixel arithmetic, k order and
// rounding are stem_kernel's: bit-identical (the stem tests compare through the model).
template <class DT, bool X32>
__global__ void __launch_bounds__(256, 2) stem_persist_kernel(const StemArgs p, const int ntiles) {
    constexpr int TH2 = 4, TW2 = 16;
    constexpr int R1 = 2 * TH2 + 1, C1 = 2 * TW2 + 1, RS = R1 + 1, CS = C1 + 1;
    constexpr int PSS = 48, PS1 = 144;
    constexpr int RPS = stem_pitch_s(CS, C1), RP1 = stem_pitch_1(C1, TW2);
    constexpr int HW1 = (C1 + 1) / 2;
    constexpr int NP1 = (R1 * C1 + 31) / 32, NT1 = (NP1 + 1) / 2;
    static_assert(TH2 * TW2 == 64, "one 32-pixel group of conv2 outputs per wave column");
    constexpr int NPX = (RS * CS + 255) / 256;             // s2d pixels per thread and tile (2)
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* const sS = smem;                                 // s2d window of the current tile
    char* const sH = sS + RS * RPS;                        // conv1 window
    char* const sT = sH + R1 * RP1;                        // conv2 output tile for the tail [64 pixels][64 ch + pad]
    // the three folded-BN biases wait in LDS: read per tile as global loads they would queue BEHIND the prefetch of the next
    // tile's window (vmcnt retires in order) and expose its latency at the first bias a conv needs
    float* const sB = reinterpret_cast<float*>(sT + 64 * PS1);        // [3][64]

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 1, wn = wave & 1;
    const int n = lane & 31, hh = lane >> 5;
    const int cbase = wm * 32 + hh * 16;
    const int per_img = p.tiles_x * p.tiles_y;

    // conv2's weights of this wave's output-channel tile: resident
    uint4 a2[36];
    {
        const uint4* ap = p.w2 + (size_t)wm * 36 * 64 + lane;
#pragma unroll
        for (int k = 0; k < 36; ++k) a2[k] = ap[k * 64];
    }
    // ---- the picture window of a tile: request (global -> registers) and deposit (registers -> LDS) ----------------------
    uint4 pf[X32 ? 1 : NPX][2];                            // xs path: the pixel's two chunks
    float2 pq[X32 ? NPX : 1][6];                           // x32 path (X32): 3 channels x 2 rows of the 2x2 block
    auto tile_origin = [&](int t, int& b, int& y2, int& x2) {
        const int tix = t % p.tiles_x;
        t /= p.tiles_x;
        b = t / p.tiles_y;
        y2 = (t - b * p.tiles_y) * TH2;
        x2 = tix * TW2;
    };
    auto request = [&](int t) {
        int b, y2, x2;
        tile_origin(t, b, y2, x2);
        const int ys = 2 * y2 - 2, xs0 = 2 * x2 - 2;
#pragma unroll
        for (int i = 0; i < NPX; ++i) {
            const int pix = tid + i * 256;
            const int r = pix / CS, c = pix - r * CS;
            const int y = ys + r, x = xs0 + c;
            const bool in = pix < RS * CS && y >= 0 && y < p.H1 && x >= 0 && x < p.W1;
            if constexpr (X32) {
                const int H0 = 2 * p.H1, W0 = 2 * p.W1;
#pragma unroll
                for (int k = 0; k < 3; ++k) {
                    const float* pl = p.x32 + (((size_t)b * 3 + k) * H0 + 2 * (in ? y : 0)) * W0 + 2 * (in ? x : 0);
                    pq[i][2 * k] = in ? *reinterpret_cast<const float2*>(pl) : make_float2(0.f, 0.f);
                    pq[i][2 * k + 1] = in ? *reinterpret_cast<const float2*>(pl + W0) : make_float2(0.f, 0.f);
                }
            } else {
                const char* src = p.xs + ((((size_t)b * p.H1 + (in ? y : 0)) * p.W1 + (in ? x : 0)) * 16) * 2;
                pf[i][0] = in ? *reinterpret_cast<const uint4*>(src) : make_uint4(0, 0, 0, 0);
                pf[i][1] = in ? *reinterpret_cast<const uint4*>(src + 16) : make_uint4(0, 0, 0, 0);
            }
        }
    };
    auto deposit = [&]() {
#pragma unroll
        for (int i = 0; i < NPX; ++i) {
            const int pix = tid + i * 256;
            if (pix >= RS * CS) continue;
            const int r = pix / CS, c = pix - r * CS;
            uint4 o[2];
            if constexpr (X32) {
                typename DT::elem v[16];
#pragma unroll
                for (int k = 0; k < 16; ++k) v[k] = (typename DT::elem)0.f;
#pragma unroll
                for (int k = 0; k < 3; ++k) {
                    v[0 * 3 + k] = (typename DT::elem)pq[i][2 * k].x;
                    v[1 * 3 + k] = (typename DT::elem)pq[i][2 * k].y;
                    v[2 * 3 + k] = (typename DT::elem)pq[i][2 * k + 1].x;
                    v[3 * 3 + k] = (typename DT::elem)pq[i][2 * k + 1].y;
                }
                __builtin_memcpy(o, v, 32);
            } else {
                o[0] = pf[i][0]; o[1] = pf[i][1];
            }
            *reinterpret_cast<uint4*>(sS + r * RPS + c * PSS) = o[0];
            *reinterpret_cast<uint4*>(sS + r * RPS + c * PSS + 16) = o[1];
        }
    };

    int t = blockIdx.x;
    if (t >= ntiles) return;
    if (tid < 64) {
        sB[tid] = p.b1[tid];
        sB[64 + tid] = p.b2[tid];
        sB[128 + tid] = p.w3 ? p.b3[tid] : 0.f;
    }
    request(t);
    deposit();
    __syncthreads();
    for (; t < ntiles; t += gridDim.x) {
        int b, y2, x2;
        tile_origin(t, b, y2, x2);
        const int y1 = 2 * y2 - 1, x1 = 2 * x2 - 1;
        const int tn = t + gridDim.x;
        // ---- conv1 on the R1 x C1 window, one 32-position group at a time ------------------------------------------------
        {
            uint4 a1[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) a1[k] = p.w1[((size_t)wm * 4 + k) * 64 + lane];
            // the next tile's picture window: requested BEHIND conv1's own fragments (conv1 then waits for those only) and in
            // flight under conv1 and conv2
            if (tn < ntiles) request(tn);
            float4 bv1[4];
#pragma unroll
            for (int r4 = 0; r4 < 4; ++r4) bv1[r4] = *reinterpret_cast<const float4*>(sB + cbase + r4 * 4);
#pragma unroll 1
            for (int g = 0; g < NT1; ++g) {
                const int q = (wn * NT1 + g) * 32 + n;
                const int qq = q < R1 * C1 ? q : 0;
                const int r = qq / C1, c = qq - r * C1;
                const int so = r * RPS + c * PSS + hh * 16;
                f32x16 acc;
#pragma unroll
                for (int r4 = 0; r4 < 4; ++r4) {
                    acc[r4 * 4 + 0] = bv1[r4].x; acc[r4 * 4 + 1] = bv1[r4].y; acc[r4 * 4 + 2] = bv1[r4].z; acc[r4 * 4 + 3] = bv1[r4].w;
                }
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const uint4 bq = *reinterpret_cast<const uint4*>(sS + so + (k >> 1) * RPS + (k & 1) * PSS);
                    acc = DT::mfma(__builtin_bit_cast(typename DT::vec8, a1[k]), __builtin_bit_cast(typename DT::vec8, bq), acc);
                }
                if (q < R1 * C1) {
                    const bool real = (y1 + r) >= 0 && (y1 + r) < p.H1 && (x1 + c) >= 0 && (x1 + c) < p.W1;   // else conv2's zero padding
#pragma unroll
                    for (int k = 0; k < 2; ++k) {
                        f32x8 f;
#pragma unroll
                        for (int e = 0; e < 8; ++e) f[e] = real ? acc[k * 8 + e] : 0.f;
                        *reinterpret_cast<uint4*>(sH + r * RP1 + ((c & 1) * HW1 + (c >> 1)) * PS1 + (cbase + k * 8) * 2) =
                            __builtin_bit_cast(uint4, DT::relu_pack(f));
                    }
                }
            }
        }
        __syncthreads();                                   // the conv1 window is complete; the s2d window is free
        // ---- conv2 (3x3 stride 2) from the conv1 window: this wave's 32 outputs x 32 channels ----------------------------
        const int q2 = wn * 32 + n;
        const int py = q2 / TW2, px = q2 - py * TW2;
        const bool ok = y2 + py < p.H2 && x2 + px < p.W2;
        {
            const int ho = (2 * py) * RP1 + px * PS1 + hh * 16;
            f32x16 acc2;
#pragma unroll
            for (int r4 = 0; r4 < 4; ++r4) {
                const float4 bv = *reinterpret_cast<const float4*>(sB + 64 + cbase + r4 * 4);
                acc2[r4 * 4 + 0] = bv.x; acc2[r4 * 4 + 1] = bv.y; acc2[r4 * 4 + 2] = bv.z; acc2[r4 * 4 + 3] = bv.w;
            }
            constexpr int NB = 4;
            auto hoff = [&](int ks) {
                const int tp = ks >> 2, c = ks & 3, kx = tp % 3;
                return (tp / 3) * RP1 + ((kx & 1) * HW1 + (kx >> 1)) * PS1 + c * 32;
            };
            uint4 bq[NB];
#pragma unroll
            for (int s0 = 0; s0 < NB - 1; ++s0) bq[s0] = *reinterpret_cast<const uint4*>(sH + ho + hoff(s0));
#pragma unroll
            for (int ks = 0; ks < 36; ++ks) {
                if (ks + NB - 1 < 36) bq[(ks + NB - 1) % NB] = *reinterpret_cast<const uint4*>(sH + ho + hoff(ks + NB - 1));
                acc2 = DT::mfma(__builtin_bit_cast(typename DT::vec8, a2[ks]), __builtin_bit_cast(typename DT::vec8, bq[ks % NB]), acc2);
            }
            char* dst = p.out + ((((size_t)b * p.H2 + y2 + py) * p.W2 + x2 + px) * p.ldo + cbase) * 2;
#pragma unroll
            for (int k = 0; k < 2; ++k) {
                f32x8 f;
#pragma unroll
                for (int e = 0; e < 8; ++e) f[e] = acc2[k * 8 + e];
                const uint4 v = __builtin_bit_cast(uint4, DT::relu_pack(f));
                if (ok) *reinterpret_cast<uint4*>(dst + k * 16) = v;
                if (p.w3) *reinterpret_cast<uint4*>(sT + q2 * PS1 + (cbase + k * 8) * 2) = v;
            }
        }
        if (tn < ntiles) deposit();                        // the next tile's picture window (the loads have had conv2 to land)
        __syncthreads();                                   // conv2's tile and the next s2d window are complete; the conv1 window is free
        // ---- tail: 1x1 conv over the 64 channels of the tile's pixels (K = 64), BN folded, ReLU --------------------------
        if (p.w3) {
            uint4 a3[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) a3[k] = p.w3[((size_t)wm * 4 + k) * 64 + lane];
            f32x16 acc3;
#pragma unroll
            for (int r4 = 0; r4 < 4; ++r4) {
                const float4 bv = *reinterpret_cast<const float4*>(sB + 128 + cbase + r4 * 4);
                acc3[r4 * 4 + 0] = bv.x; acc3[r4 * 4 + 1] = bv.y; acc3[r4 * 4 + 2] = bv.z; acc3[r4 * 4 + 3] = bv.w;
            }
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const uint4 bq3 = *reinterpret_cast<const uint4*>(sT + q2 * PS1 + k * 32 + hh * 16);
                acc3 = DT::mfma(__builtin_bit_cast(typename DT::vec8, a3[k]), __builtin_bit_cast(typename DT::vec8, bq3), acc3);
            }
            if (ok) {
                char* dst = p.out3 + ((((size_t)b * p.H2 + y2 + py) * p.W2 + x2 + px) * p.ldo3 + cbase) * 2;
#pragma unroll
                for (int k = 0; k < 2; ++k) {
                    f32x8 f;
#pragma unroll
                    for (int e = 0; e < 8; ++e) f[e] = acc3[k * 8 + e];
                    *reinterpret_cast<uint4*>(dst + k * 16) = __builtin_bit_cast(uint4, DT::relu_pack(f));
                }
            }
        }
        // (the tail's reads of sT end before the NEXT tile's conv2 writes it: the barrier behind its conv1 lies between)
    }
}

int g_stem_persist = 1;                // rsg_tune("stem_persist", 0|1)

// xs: output of rsg_nchw_f32_to_s2d_nhwc16 [batch, h/2, w/2, 16], or NULL with x_nchw_f32 = the fp32 NCHW picture
// [batch, 3, h, w] itself (then the layout conversion happens while staging and needs no launch); w1/b1: the 2x2 s2d form of conv1 packed with
// rsg_pack_conv_weights_host(cin 16, cout 64, 2x2); w2/b2: conv2 packed (cin 64, cout 64, 3x3); out16: [batch, h/4,
// w/4, out_ld] 16-bit NHWC (64 channels written).  h, w: the INPUT picture size (multiples of 4).
static int stem_launch(const void* xs, const float* x_nchw_f32, const void* w1_packed, const float* bias1, const void* w2_packed,
                       const float* bias2, void* out16, int32_t out_ld, const void* w3_packed, const float* bias3, void* out3,
                       int32_t out3_ld, int32_t batch, int32_t h, int32_t w, int32_t dtype, void* stream) {
    if ((!xs && !x_nchw_f32) || !w1_packed || !bias1 || !w2_packed || !bias2 || !out16) RSG_FAIL("stem_fused: null pointer");
    if (w3_packed && (!bias3 || !out3 || out3_ld < 64 || (out3_ld & 7) || (((uintptr_t)w3_packed | (uintptr_t)bias3 | (uintptr_t)out3) & 15)))
        RSG_FAIL("stem_fused: the tail needs bias, a 16-byte aligned output and out3_ld >= 64");
    if ((uintptr_t)x_nchw_f32 & 7) RSG_FAIL("stem_fused: the picture must be 8-byte aligned");
    if (batch < 1 || h < 4 || w < 4 || (h & 3) || (w & 3) || out_ld < 64 || (out_ld & 7)) RSG_FAIL("stem_fused: bad shape");
    if (((uintptr_t)xs | (uintptr_t)w1_packed | (uintptr_t)w2_packed | (uintptr_t)out16) & 15) RSG_FAIL("stem_fused: alignment");
    // tile of conv2 outputs per workgroup (-DRSG_STEM_TH2 / _TW2: same-visit A/B builds; every tiling gives the same bits)
#ifndef RSG_STEM_TH2
#define RSG_STEM_TH2 4
#endif
#ifndef RSG_STEM_TW2
#define RSG_STEM_TW2 12
#endif
    constexpr int TH2 = RSG_STEM_TH2, TW2 = RSG_STEM_TW2;
    StemArgs a;
    a.xs = (const char*)xs; a.x32 = xs ? nullptr : x_nchw_f32; a.w1 = (const uint4*)w1_packed; a.b1 = bias1; a.w2 = (const uint4*)w2_packed; a.b2 = bias2;
    a.out = (char*)out16; a.B = batch; a.H1 = h / 2; a.W1 = w / 2; a.H2 = h / 4; a.W2 = w / 4; a.ldo = out_ld;
    a.w3 = (const uint4*)w3_packed; a.b3 = bias3; a.out3 = (char*)out3; a.ldo3 = out3_ld;
    a.tiles_x = ceil_div(a.W2, TW2); a.tiles_y = ceil_div(a.H2, TH2);
    if (g_stem_persist) {
        // two workgroups per CU, each walking tiles of 4 x 16 outputs (grid-stride: consecutive tiles = one picture's rows)
        constexpr int PH = 4, PW = 16;
        a.tiles_x = ceil_div(a.W2, PW); a.tiles_y = ceil_div(a.H2, PH);
        const int ntiles = batch * a.tiles_x * a.tiles_y;
        const int plds = (2 * PH + 2) * stem_pitch_s(2 * PW + 2, 2 * PW + 1) + (2 * PH + 1) * stem_pitch_1(2 * PW + 1, PW) + 64 * 144 + 3 * 64 * 4;
        static int cus[RSG_MAX_DEVICES], lds_set[RSG_MAX_DEVICES][4];
        int dev = 0;
        if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= RSG_MAX_DEVICES) RSG_FAIL("stem_fused: no current device");
        if (!cus[dev] && (hipDeviceGetAttribute(&cus[dev], hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus[dev] < 1))
            RSG_FAIL("stem_fused: device properties");
        const int v = (dtype == RSG_BF16 ? 2 : 0) + (a.x32 ? 1 : 0);
        const void* fns[4] = {(const void*)stem_persist_kernel<F16, false>, (const void*)stem_persist_kernel<F16, true>,
                              (const void*)stem_persist_kernel<BF16, false>, (const void*)stem_persist_kernel<BF16, true>};
        if (lds_set[dev][v] < plds) {
            if (hipFuncSetAttribute(fns[v], hipFuncAttributeMaxDynamicSharedMemorySize, plds) != hipSuccess) RSG_FAIL("stem_fused: cannot reserve %d B of LDS", plds);
            lds_set[dev][v] = plds;
        }
        const dim3 pgrid(ntiles < 2 * cus[dev] ? ntiles : 2 * cus[dev]);
        switch (v) {
            case 0: RSG_LAUNCH((stem_persist_kernel<F16, false>), pgrid, dim3(256), plds, (hipStream_t)stream, a, ntiles); break;
            case 1: RSG_LAUNCH((stem_persist_kernel<F16, true>), pgrid, dim3(256), plds, (hipStream_t)stream, a, ntiles); break;
            case 2: RSG_LAUNCH((stem_persist_kernel<BF16, false>), pgrid, dim3(256), plds, (hipStream_t)stream, a, ntiles); break;
            default: RSG_LAUNCH((stem_persist_kernel<BF16, true>), pgrid, dim3(256), plds, (hipStream_t)stream, a, ntiles); break;
        }
        RSG_CHECK_LAUNCH("stem_fused");
        return 0;
    }
    const int lds = (2 * TH2 + 2) * stem_pitch_s(2 * TW2 + 2, 2 * TW2 + 1) + (2 * TH2 + 1) * stem_pitch_1(2 * TW2 + 1, TW2);
    const dim3 grid(batch * a.tiles_x * a.tiles_y);
    if (dtype == RSG_BF16) RSG_LAUNCH((stem_kernel<BF16, TH2, TW2>), grid, dim3(256), lds, (hipStream_t)stream, a);
    else RSG_LAUNCH((stem_kernel<F16, TH2, TW2>), grid, dim3(256), lds, (hipStream_t)stream, a);
    RSG_CHECK_LAUNCH("stem_fused");
    return 0;
}

extern "C" int rsg_stem_fused(const void* xs, const float* x_nchw_f32, const void* w1_packed, const float* bias1, const void* w2_packed,
                              const float* bias2, void* out16, int32_t out_ld, int32_t batch, int32_t h, int32_t w,
                              int32_t dtype, void* stream) {
    return stem_launch(xs, x_nchw_f32, w1_packed, bias1, w2_packed, bias2, out16, out_ld, nullptr, nullptr, nullptr, 0, batch, h, w,
                       dtype, stream);
}

// rsg_stem_fused + the first Bottleneck's conv1 (pose_hrnet.py:85-87 of layer1.0: 1x1, 64 -> 64, BatchNorm folded, ReLU) on the
// stem's output in the same launch; w3/b3 from rsg_pack_conv_weights_host(cin 64, cout 64, 1x1); out3: [batch, h/4, w/4, out3_ld].
// Bit-identical to rsg_stem_fused followed by rsg_conv_igemm.
extern "C" int rsg_stem_fused_conv1(const void* xs, const float* x_nchw_f32, const void* w1_packed, const float* bias1,
                                    const void* w2_packed, const float* bias2, void* out16, int32_t out_ld, const void* w3_packed,
                                    const float* bias3, void* out3, int32_t out3_ld, int32_t batch, int32_t h, int32_t w,
                                    int32_t dtype, void* stream) {
    if (!w3_packed) RSG_FAIL("stem_fused_conv1: null pointer");
    return stem_launch(xs, x_nchw_f32, w1_packed, bias1, w2_packed, bias2, out16, out_ld, w3_packed, bias3, out3, out3_ld, batch, h, w,
                       dtype, stream);
}
