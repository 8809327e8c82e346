// The last HighResolutionModule's fuse sum for branch 0 (pose_hrnet.py:259-266) and the 1x1 classifier on it
// (pose_hrnet.py:326-332,461) in ONE launch: y0 = ReLU(x0 + up(a1) + up(a2) + up(a3)) is summed while a workgroup stages
// its 128 pixels (the operand order and the 16-bit rounding of rsg_hr_fuse_sum / fuse_body), goes to LDS and is the B
// operand of the classifier's two k-steps; the heat maps leave as fp32 NCHW.  The 12.6 MB map y0 is neither written nor
// read back, and one launch goes.  Bit-identical to rsg_conv_fuse_group_run (the sum) followed by rsg_conv_igemm.
#include <type_traits>
#include "fuse_body.h"

struct FuseHeadArgs {
    FuseArgs f;                // operands of the sum; f.out unused
    const uint4* wpk;          // classifier, packed (cin 32: KS = 2), cout <= 32
    const float* bias;         // [32]
    float* out32;              // [batch, cout, h, w]
    int cout, npix;            // npix = batch * h * w
};

template <class DT>
__global__ void __launch_bounds__(256) fuse_head_kernel(const FuseHeadArgs p) {
    constexpr int PS = 80;                                 // 32 ch * 2 B + 16 pad (odd number of 16-byte slots)
    __shared__ __attribute__((aligned(16))) char tile[128 * PS];
    const FuseArgs& a = p.f;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int n = lane & 31, hh = lane >> 5;
    const unsigned w = (unsigned)a.w, h = (unsigned)a.h;
    const int pix_base = blockIdx.x * 128;
    // ---- the sum, 8 channels of a pixel per step (fuse_body's arithmetic), two steps per thread --------------------------
#pragma unroll
    for (int rep = 0; rep < 2; ++rep) {
        const int id = rep * 256 + tid;                    // 128 pixels x 4 channel groups
        const int pl = id >> 2, cg = id & 3;
        const unsigned pix0 = (unsigned)(pix_base + pl);
        uint4 out = make_uint4(0, 0, 0, 0);
        if ((int)pix0 < p.npix) {
            uint4 vs[4], vl[3];
#pragma unroll
            for (int k = 0; k < 4; ++k)
                if (k < a.n_same) vs[k] = reinterpret_cast<const uint4*>(a.same[k])[(size_t)pix0 * a.ld_same[k] + cg];
            if (a.n_low) {
                const unsigned t = pix0 / w, x = pix0 - t * w;
                const unsigned b = t / h, y = t - b * h;
#pragma unroll
                for (int k = 0; k < 3; ++k)
                    if (k < a.n_low) {
                        const int s = a.shift[k];
                        const size_t src = (((size_t)b * (h >> s) + (y >> s)) * (w >> s) + (x >> s)) * a.ld_low[k] + cg;
                        vl[k] = reinterpret_cast<const uint4*>(a.low[k])[src];
                    }
            }
            float acc[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) acc[e] = 0.f;
#pragma unroll
            for (int k = 0; k < 4; ++k)
                if (k < a.n_same) {
                    const typename DT::vec8 v = __builtin_bit_cast(typename DT::vec8, vs[k]);
#pragma unroll
                    for (int e = 0; e < 8; ++e) acc[e] += (float)v[e];
                }
#pragma unroll
            for (int k = 0; k < 3; ++k)
                if (k < a.n_low) {
                    const typename DT::vec8 v = __builtin_bit_cast(typename DT::vec8, vl[k]);
#pragma unroll
                    for (int e = 0; e < 8; ++e) acc[e] += (float)v[e];
                }
            f32x8 f;
#pragma unroll
            for (int e = 0; e < 8; ++e) f[e] = acc[e];
            out = __builtin_bit_cast(uint4, DT::relu_pack(f));
        }
        *reinterpret_cast<uint4*>(tile + pl * PS + cg * 16) = out;
    }
    // ---- classifier: a wave owns 32 pixels; lane (n, hh) ends with 16 consecutive output channels of pixel n ------------
    const uint4 a0 = p.wpk[lane], a1 = p.wpk[64 + lane];
    f32x16 acc;
#pragma unroll
    for (int r4 = 0; r4 < 4; ++r4) {
        const float4 bv = *reinterpret_cast<const float4*>(p.bias + hh * 16 + r4 * 4);
        acc[r4 * 4 + 0] = bv.x; acc[r4 * 4 + 1] = bv.y; acc[r4 * 4 + 2] = bv.z; acc[r4 * 4 + 3] = bv.w;
    }
    __syncthreads();
    const char* px = tile + (wave * 32 + n) * PS + hh * 16;
    acc = DT::mfma(__builtin_bit_cast(typename DT::vec8, a0), __builtin_bit_cast(typename DT::vec8, *reinterpret_cast<const uint4*>(px)), acc);
    acc = DT::mfma(__builtin_bit_cast(typename DT::vec8, a1), __builtin_bit_cast(typename DT::vec8, *reinterpret_cast<const uint4*>(px + 32)), acc);
    const int pix = pix_base + wave * 32 + n;
    if (pix >= p.npix) return;
    const int plane = a.h * a.w;
    const int b = pix / plane, rem = pix - b * plane;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int c = hh * 16 + r;
        if (c < p.cout) p.out32[((size_t)b * p.cout + c) * plane + rem] = acc[r];
    }
}

// same / low operands as rsg_hr_fuse_sum (32 channels: the highest-resolution branch of HRNet-W32); w_packed / bias from
// rsg_pack_conv_weights_host(cin 32, cout, 1x1) with the classifier's bias as conv_bias; out: fp32 [batch, cout, h, w].
extern "C" int rsg_fuse_classifier(const void* const* same, const int32_t* same_ld, int32_t n_same, const void* const* low,
                                   const int32_t* low_shift, const int32_t* low_ld, int32_t n_low, const void* w_packed,
                                   const float* bias, float* out_nchw_f32, int32_t batch, int32_t h, int32_t w, int32_t cout,
                                   int32_t dtype, void* stream) {
    if (!w_packed || !bias || !out_nchw_f32 || cout < 1 || cout > 32) RSG_FAIL("fuse_classifier: bad arguments (cout <= 32)");
    if (((uintptr_t)w_packed | (uintptr_t)bias) & 15) RSG_FAIL("fuse_classifier: alignment");
    FuseHeadArgs p;
    int blocks = 0;
    char dummy;
    if (rsg_fuse_prepare(same, n_same, low, low_shift, n_low, &dummy, batch, h, w, 32, p.f, blocks, same_ld, low_ld)) return 1;
    p.f.out = nullptr;
    p.wpk = (const uint4*)w_packed; p.bias = bias; p.out32 = out_nchw_f32; p.cout = cout; p.npix = batch * h * w;
    const int grid = (p.npix + 127) / 128;
    if (dtype == RSG_BF16) RSG_LAUNCH(fuse_head_kernel<BF16>, dim3(grid), dim3(256), 0, (hipStream_t)stream, p);
    else RSG_LAUNCH(fuse_head_kernel<F16>, dim3(grid), dim3(256), 0, (hipStream_t)stream, p);
    RSG_CHECK_LAUNCH("fuse_classifier");
    return 0;
}

// ---- the whole fuse of the LAST HighResolutionModule (multi_scale_output = False: only output 0) + the classifier ----------
// y0 = ReLU(x0 + up2(bn(conv1x1(x1))) + up4(bn(conv1x1(x2))) + up8(bn(conv1x1(x3)))) (pose_hrnet.py:190-213,259-266), then the
// 1x1 classifier (:326-332,461).  A workgroup owns 8 rows of the highest-resolution map: exactly 4 / 2 / 1 rows of the
// three lower ones, so nothing is computed twice.  It stages those low-resolution rows, runs the three 1x1 convs on them
// (K = 64 / 128 / 256, 32 outputs, BatchNorm folded, no ReLU, rounded to 16 bits as the grouped launch stores them), sums
// (fuse_body's order and rounding) into an LDS tile and runs the classifier on it.  Replaces two launches (the module's
// 1x1 convs; sum + classifier) and the traffic of their outputs.  Bit-identical to rsg_conv_igemm x3 + rsg_fuse_classifier.
struct FuseHead4Args {
    const char* x0; int ld0;                  // [B, H, W, ld0], 32 channels
    const char* xl[3]; int ldl[3];            // [B, H >> (j+1), W >> (j+1), ldl[j]], 64 << j channels
    const uint4* wl[3]; const float* bl[3];   // packed 1x1 convs (cin 64 << j, cout 32), bias [32]
    const uint4* wc; const float* bc;         // classifier (cin 32), bias [32]
    float* out32;
    int B, H, W, cout, n_low;
};

template <class DT>
__global__ void __launch_bounds__(256) fuse_head4_kernel(const FuseHead4Args p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int PSA = 80;                                     // 32-channel pixel in LDS: 64 B + 16 pad
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int n = lane & 31, hh = lane >> 5;
    const int tiles = p.H >> 3;
    const int b = blockIdx.x / tiles, ty = blockIdx.x - b * tiles;
    const int W = p.W;
    // LDS map: [low-resolution input tiles j = 0..2][their conv outputs a_j][y0 tile]
    int off_x[3], off_a[3], npx[3], psx[3];
    int off = 0;
#pragma unroll
    for (int j = 0; j < 3; ++j) {
        npx[j] = (8 >> (j + 1)) * (W >> (j + 1));
        psx[j] = (64 << j) * 2 + 16;
        off_x[j] = off;
        if (j < p.n_low) off += npx[j] * psx[j];
    }
#pragma unroll
    for (int j = 0; j < 3; ++j) { off_a[j] = off; if (j < p.n_low) off += npx[j] * PSA; }
    char* const sy = smem + off;
    // ---- 1. stage the low-resolution rows (each tile is one contiguous run of rows of its map) ---------------------------
#pragma unroll
    for (int j = 0; j < 3; ++j) {
        if (j >= p.n_low) break;
        const int cpp = (64 << j) / 8;                          // 16-byte chunks per pixel
        const int rows = 8 >> (j + 1), wl = W >> (j + 1), hl = p.H >> (j + 1);
        const char* src = p.xl[j] + ((size_t)(b * hl + ty * rows) * wl) * p.ldl[j] * 2;
        for (int id = tid; id < npx[j] * cpp; id += 256) {
            const int px = id / cpp, cg = id - px * cpp;
            *reinterpret_cast<uint4*>(smem + off_x[j] + px * psx[j] + cg * 16) =
                *reinterpret_cast<const uint4*>(src + ((size_t)px * p.ldl[j] + cg * 8) * 2);
        }
    }
    __syncthreads();
    // ---- 2. the three 1x1 convs on the staged pixels: one job = (conv, 32-pixel group); the jobs go round the four waves and a
    // job's weight fragments are all requested before its first MFMA ------------------------------------------------------
    auto job = [&](auto ksn, int j, int g) {
        constexpr int KSN = decltype(ksn)::value;
        const int q = g * 32 + n;
        const char* px = smem + off_x[j] + (q < npx[j] ? q : 0) * psx[j] + hh * 16;
        uint4 a[KSN];
#pragma unroll
        for (int ks = 0; ks < KSN; ++ks) a[ks] = p.wl[j][ks * 64 + lane];
        f32x16 acc;
#pragma unroll
        for (int r4 = 0; r4 < 4; ++r4) {
            const float4 bv = *reinterpret_cast<const float4*>(p.bl[j] + hh * 16 + r4 * 4);
            acc[r4 * 4 + 0] = bv.x; acc[r4 * 4 + 1] = bv.y; acc[r4 * 4 + 2] = bv.z; acc[r4 * 4 + 3] = bv.w;
        }
#pragma unroll
        for (int ks = 0; ks < KSN; ++ks)
            acc = DT::mfma(__builtin_bit_cast(typename DT::vec8, a[ks]),
                           __builtin_bit_cast(typename DT::vec8, *reinterpret_cast<const uint4*>(px + ks * 32)), acc);
        if (q < npx[j]) {
#pragma unroll
            for (int k = 0; k < 2; ++k) {
                f32x8 f;
#pragma unroll
                for (int e = 0; e < 8; ++e) f[e] = acc[k * 8 + e];
                *reinterpret_cast<uint4*>(smem + off_a[j] + q * PSA + (hh * 16 + k * 8) * 2) =
                    __builtin_bit_cast(uint4, __builtin_convertvector(f, typename DT::vec8));
            }
        }
    };
    {
        int jn = 0;
#pragma unroll
        for (int j = 0; j < 3; ++j) {
            if (j >= p.n_low) break;
            for (int g = 0; g * 32 < npx[j]; ++g, ++jn) {
                if ((jn & 3) != wave) continue;
                if (j == 0) job(std::integral_constant<int, 4>{}, 0, g);
                else if (j == 1) job(std::integral_constant<int, 8>{}, 1, g);
                else job(std::integral_constant<int, 16>{}, 2, g);
            }
        }
    }
    __syncthreads();
    // ---- 3. the sum (fuse_body: x0, then the up-sampled addends in order; fp32; ReLU; 16 bits) into the y0 tile ----------
    const int npix = 8 * W;
    const char* x0 = p.x0 + ((size_t)(b * p.H + ty * 8) * W) * p.ld0 * 2;
    for (int id = tid; id < npix * 4; id += 256) {
        const int pl = id >> 2, cg = id & 3;
        const int y = pl / W, x = pl - y * W;
        float acc[8];
        {
            const typename DT::vec8 v = __builtin_bit_cast(typename DT::vec8, *reinterpret_cast<const uint4*>(x0 + ((size_t)pl * p.ld0 + cg * 8) * 2));
#pragma unroll
            for (int e = 0; e < 8; ++e) acc[e] = 0.f + (float)v[e];
        }
#pragma unroll
        for (int j = 0; j < 3; ++j) {
            if (j >= p.n_low) break;
            const int s = j + 1;
            const typename DT::vec8 v = __builtin_bit_cast(typename DT::vec8, *reinterpret_cast<const uint4*>(
                smem + off_a[j] + ((y >> s) * (W >> s) + (x >> s)) * PSA + cg * 16));
#pragma unroll
            for (int e = 0; e < 8; ++e) acc[e] += (float)v[e];
        }
        f32x8 f;
#pragma unroll
        for (int e = 0; e < 8; ++e) f[e] = acc[e];
        *reinterpret_cast<uint4*>(sy + pl * PSA + cg * 16) = __builtin_bit_cast(uint4, DT::relu_pack(f));
    }
    // ---- 4. classifier -------------------------------------------------------------------------------------------------
    const uint4 a0 = p.wc[lane], a1 = p.wc[64 + lane];
    float cb[16];
#pragma unroll
    for (int r4 = 0; r4 < 4; ++r4) {
        const float4 bv = *reinterpret_cast<const float4*>(p.bc + hh * 16 + r4 * 4);
        cb[r4 * 4 + 0] = bv.x; cb[r4 * 4 + 1] = bv.y; cb[r4 * 4 + 2] = bv.z; cb[r4 * 4 + 3] = bv.w;
    }
    __syncthreads();
    const int plane = p.H * W;
    for (int g = wave; g * 32 < npix; g += 4) {
        const int q = g * 32 + n;
        const char* px = sy + (q < npix ? q : 0) * PSA + hh * 16;
        f32x16 acc;
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[r] = cb[r];
        acc = DT::mfma(__builtin_bit_cast(typename DT::vec8, a0), __builtin_bit_cast(typename DT::vec8, *reinterpret_cast<const uint4*>(px)), acc);
        acc = DT::mfma(__builtin_bit_cast(typename DT::vec8, a1), __builtin_bit_cast(typename DT::vec8, *reinterpret_cast<const uint4*>(px + 32)), acc);
        if (q >= npix) continue;
        const int rem = ty * 8 * W + q;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int c = hh * 16 + r;
            if (c < p.cout) p.out32[((size_t)b * p.cout + c) * plane + rem] = acc[r];
        }
    }
}

// x0: [batch,h,w,ld0] 32 channels; lows[j]: [batch, h >> (j+1), w >> (j+1), low_ld[j]] with 64 << j channels, w_low[j] / b_low[j]
// from rsg_pack_conv_weights_host(cin 64 << j, cout 32, 1x1) with the BatchNorm folded; w_cls / b_cls: the classifier
// (cin 32, cout <= 32, its bias folded into b_cls); out: fp32 [batch, cout, h, w].  h % 8 == 0, w % 8 == 0.
extern "C" int rsg_fuse_head4(const void* x0, int32_t ld0, const void* const* lows, const int32_t* low_ld, int32_t n_low,
                              const void* const* w_low, const float* const* b_low, const void* w_cls, const float* b_cls,
                              float* out_nchw_f32, int32_t batch, int32_t h, int32_t w, int32_t cout, int32_t dtype, void* stream) {
    if (!x0 || !w_cls || !b_cls || !out_nchw_f32 || n_low < 1 || n_low > 3 || cout < 1 || cout > 32 || batch < 1)
        RSG_FAIL("fuse_head4: bad arguments");
    if ((h & 7) || (w & 7) || ld0 < 32 || (ld0 & 7) || (((uintptr_t)x0 | (uintptr_t)w_cls | (uintptr_t)b_cls) & 15)) RSG_FAIL("fuse_head4: shape / alignment");
    FuseHead4Args p;
    memset(&p, 0, sizeof(p));
    p.x0 = (const char*)x0; p.ld0 = ld0; p.wc = (const uint4*)w_cls; p.bc = b_cls; p.out32 = out_nchw_f32;
    p.B = batch; p.H = h; p.W = w; p.cout = cout; p.n_low = n_low;
    int lds = 8 * w * 80;
    for (int j = 0; j < n_low; ++j) {
        const int c = 64 << j;
        if (!lows[j] || !w_low[j] || !b_low[j] || low_ld[j] < c || (low_ld[j] & 7) ||
            (((uintptr_t)lows[j] | (uintptr_t)w_low[j] | (uintptr_t)b_low[j]) & 15))
            RSG_FAIL("fuse_head4: low operand %d", j);
        p.xl[j] = (const char*)lows[j]; p.ldl[j] = low_ld[j]; p.wl[j] = (const uint4*)w_low[j]; p.bl[j] = b_low[j];
        const int npx = (8 >> (j + 1)) * (w >> (j + 1));
        lds += npx * (c * 2 + 16) + npx * 80;
    }
    if (lds > 150 * 1024) RSG_FAIL("fuse_head4: a %d-pixel-wide map needs %d B of LDS", w, lds);
    const void* fn = dtype == RSG_BF16 ? (const void*)fuse_head4_kernel<BF16> : (const void*)fuse_head4_kernel<F16>;
    static int lds_set[2] = {0, 0};
    if (lds_set[dtype == RSG_BF16] < lds) {
        if (hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, lds) != hipSuccess) RSG_FAIL("fuse_head4: cannot reserve %d B of LDS", lds);
        lds_set[dtype == RSG_BF16] = lds;
    }
    const dim3 grid(batch * (h / 8));
    if (dtype == RSG_BF16) RSG_LAUNCH(fuse_head4_kernel<BF16>, grid, dim3(256), lds, (hipStream_t)stream, p);
    else RSG_LAUNCH(fuse_head4_kernel<F16>, grid, dim3(256), lds, (hipStream_t)stream, p);
    RSG_CHECK_LAUNCH("fuse_head4");
    return 0;
}

// ---- the two highest-resolution outputs of ANY HighResolutionModule in one launch (pose_hrnet.py:190-268) -----------------
//   y0 = ReLU(x0 + up2(a01) + up4(a02) + up8(a03)),            a0j = bn(conv1x1(x_j)), 32 channels
//   y1 = ReLU(d10 + x1 + up2(a12) + up4(a13)),                 d10 = bn(conv3x3 stride 2 (x0)), a1j = bn(conv1x1(x_j)), 64 channels
// These paths hold 90 % of a module's fuse bytes and 6 of its up to 12 convs (half of the grouped launches' workgroups),
// but only 57 k weights, so a row band can afford to re-read them: a workgroup owns 8 rows of y0 = 4 rows of y1 and
// stages 9 rows of x0 (one halo row above, a zero column left), 4 / 2 / 1 rows of x1 / x2 / x3.  The grouped launches keep
// the deep paths (whose weights want whole batches per workgroup).  Arithmetic, operand order and the 16-bit rounding
// points are those of the grouped convs and of fuse_body: bit-identical.
// Diagnostic build (-DRSG_BAND_STAMPS, tools/band_stamps.py): thread 0 of every band workgroup records the shader clock at
// the phase boundaries; the production build contains none of this.
#ifdef RSG_BAND_STAMPS
static __device__ unsigned long long* g_band_stamps = nullptr;       // [blocks][8]
extern "C" int rsg_band_set_stamp_buffer(void* dev_ptr) {
    unsigned long long* q = (unsigned long long*)dev_ptr;
    if (hipMemcpyToSymbol(HIP_SYMBOL(g_band_stamps), &q, sizeof(q)) != hipSuccess) RSG_FAIL("band stamps: memcpy failed");
    return 0;
}
#define RSG_BSTAMP(slot)                                                                                   \
    do {                                                                                                   \
        if (g_band_stamps && threadIdx.x == 0) {                                                           \
            unsigned long long t__;                                                                        \
            asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t__)::"memory");                 \
            g_band_stamps[(size_t)blockIdx.x * 8 + (slot)] = t__;                                          \
        }                                                                                                  \
    } while (0)
#else
#define RSG_BSTAMP(slot) do {} while (0)
#endif

struct FuseBandArgs {
    const char* x[4]; int ld[4];              // x0 (32 ch) .. x3 (256 ch)
    const uint4* w0[3]; const float* b0[3];   // a0j, j = 1..3: cin 64 << (j-1) -> 32
    const uint4* wd; const float* bd;         // d10: 3x3 stride 2, 32 -> 64
    const uint4* w1[2]; const float* b1[2];   // a1j, j = 2..3: cin 64 << (j-1) -> 64
    char* y0; int ldy0;
    char* y1; int ldy1;
    int B, H, W, n_low;
};

template <class DT>
__global__ void __launch_bounds__(256, 2) fuse_band_kernel(const FuseBandArgs p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int PS0 = 80, PS1 = 144;                          // LDS pixel pitch of 32- / 64-channel maps
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int n = lane & 31, hh = lane >> 5;
    RSG_BSTAMP(0);
    const int tiles = p.H >> 3;
    const int b = blockIdx.x / tiles, ty = blockIdx.x - b * tiles;
    const int W = p.W, W1 = W >> 1, H1 = p.H >> 1;
    // A workgroup is a chain of latencies (rows arriving, a weight round trip per conv job, dependent MFMAs): the weights of
    // the strided conv are requested at entry, every job's successor's weights under its own MFMAs, and a wave runs its
    // two pixel groups of the strided conv interleaved.
    const int d_ct = wave & 1;                                  // strided conv: waves 0/2 take channel tile 0, 1/3 tile 1
    // LDS map
    const int RP0 = (W + 1) * PS0;                              // x0 rows: one zero pixel in front
    char* const sx0 = smem;                                     // 9 rows: image rows 8 ty - 1 .. 8 ty + 7
    int off = 9 * RP0;
    int off_x[3], npx[3], psx[3], off_a0[3], off_a1[2];
#pragma unroll
    for (int j = 0; j < 3; ++j) {
        npx[j] = (8 >> (j + 1)) * (W >> (j + 1));
        psx[j] = (64 << j) * 2 + 16;
        off_x[j] = off;
        if (j < p.n_low) off += npx[j] * psx[j];
    }
#pragma unroll
    for (int j = 0; j < 3; ++j) { off_a0[j] = off; if (j < p.n_low) off += npx[j] * PS0; }
#pragma unroll
    for (int j = 1; j < 3; ++j) { off_a1[j - 1] = off; if (j < p.n_low) off += npx[j] * PS1; }
    // ---- the 1x1 conv jobs of this wave: job k = (conv, 32-pixel group, channel tile), k = wave, wave + 4, ...; order: for every
    // lower map j its conv j -> 0 (groups), then its conv j -> 1 (groups x 2 tiles, j >= 1) ---------------------------------
    int ng[3], jfirst[4];
    jfirst[0] = 0;
#pragma unroll
    for (int j = 0; j < 3; ++j) {
        ng[j] = j < p.n_low ? (npx[j] + 31) >> 5 : 0;
        jfirst[j + 1] = jfirst[j] + ng[j] * (j >= 1 ? 3 : 1);
    }
    const int njobs = jfirst[3];
    struct Job { const uint4* w; const float* bias; int src, ps, np, ksn, g, ct, dst, psd; };
    // (offsets and sizes as arithmetic in j: a select chain over the small per-map arrays is turned into a table in scratch)
    const int npx0 = 4 * (W >> 1), npx1 = 2 * (W >> 2), npx2 = W >> 3;
    const int base_x = 9 * RP0;
    const int base_a0 = off_a0[0], base_a1 = off_a1[0];
    auto job_of = [&](int k) -> Job {
        Job r;
        const int j = (k >= jfirst[1] ? 1 : 0) + (k >= jfirst[2] ? 1 : 0);
        const int t = k - (j >= 1 ? jfirst[1] : 0) - (j >= 2 ? jfirst[2] - jfirst[1] : 0);
        r.np = (8 >> (j + 1)) * (W >> (j + 1));
        const int ngj = (r.np + 31) >> 5;
        const bool to1 = t >= ngj;
        r.g = to1 ? (t - ngj) >> 1 : t;
        r.ct = to1 ? (t - ngj) & 1 : 0;
        r.ksn = 4 << j;
        r.ps = (128 << j) + 16;
        r.src = base_x + (j >= 1 ? npx0 * 144 : 0) + (j >= 2 ? npx1 * 272 : 0);
        r.dst = to1 ? base_a1 + (j >= 2 ? npx1 * PS1 : 0) : base_a0 + (j >= 1 ? npx0 * PS0 : 0) + (j >= 2 ? npx1 * PS0 : 0);
        r.psd = to1 ? PS1 : PS0;
        const uint4* w = to1 ? p.w1[0] : p.w0[0];
        const float* bs = to1 ? p.b1[0] : p.b0[0];
        if (to1 && j == 2) { w = p.w1[1]; bs = p.b1[1]; }
        if (!to1 && j == 1) { w = p.w0[1]; bs = p.b0[1]; }
        if (!to1 && j == 2) { w = p.w0[2]; bs = p.b0[2]; }
        r.w = w + (size_t)r.ct * r.ksn * 64 + lane;
        r.bias = bs;
        return r;
    };
    // always 16 loads (a short job re-reads its last fragment): a fixed count lets the compiler wait for exactly one job's
    // weights (s_waitcnt vmcnt(16)) while the next job's are still in flight; predicated loads made it wait for everything
    auto fetch_w = [&](const Job jb, bool, uint4 (&a)[16]) {
#pragma unroll
        for (int ks = 0; ks < 16; ++ks) a[ks] = jb.w[(ks < jb.ksn ? ks : jb.ksn - 1) * 64];
    };
    Job j0 = job_of(wave < njobs ? wave : 0);
    uint4 wa[16], wb[16];
    fetch_w(j0, wave < njobs, wa);                              // the first job's weights travel with the staged rows
    // ---- 1. stage: ONE round trip -- the lower maps' rows (up to 4 loads per thread and map) and the first 8 x0 loads per thread
    // are all requested before the first LDS write (wider maps: further x0 rounds of 8) ---------------------------------------
    {
        uint4 v2[3][4];
        int lo2[3][4];
#pragma unroll
        for (int j = 0; j < 3; ++j) {
            const int cpp = (64 << j) / 8;
            const int rows = 8 >> (j + 1), wl = W >> (j + 1), hl = p.H >> (j + 1);
            const char* src = j < p.n_low ? p.x[j + 1] + ((size_t)(b * hl + ty * rows) * wl) * p.ld[j + 1] * 2 : nullptr;
            const int total = j < p.n_low ? npx[j] * cpp : 0;
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int id = u * 256 + tid;
                lo2[j][u] = -1;
                v2[j][u] = make_uint4(0, 0, 0, 0);
                if (id < total) {
                    const int px = id / cpp, cg = id - px * cpp;
                    lo2[j][u] = off_x[j] + px * psx[j] + cg * 16;
                    v2[j][u] = *reinterpret_cast<const uint4*>(src + ((size_t)px * p.ld[j + 1] + cg * 8) * 2);
                }
            }
        }
        const int y_lo = ty * 8 - 1;
        const int total0 = 9 * (W + 1) * 4;
        for (int base = 0; base < total0; base += 256 * 8) {
            uint4 v[8];
            int lo[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int id = base + u * 256 + tid;
                v[u] = make_uint4(0, 0, 0, 0);
                lo[u] = -1;
                if (id < total0) {
                    const int cg = id & 3, pc = id >> 2;
                    const int r = pc / (W + 1), c = pc - r * (W + 1);   // c = 0: the zero pixel at x = -1
                    const int y = y_lo + r, x = c - 1;
                    lo[u] = r * RP0 + c * PS0 + cg * 16;
                    if (y >= 0 && x >= 0) v[u] = *reinterpret_cast<const uint4*>(p.x[0] + (((size_t)(b * p.H + y) * W + x) * p.ld[0] + cg * 8) * 2);
                }
            }
            if (base == 0) {
#pragma unroll
                for (int j = 0; j < 3; ++j)
#pragma unroll
                    for (int u = 0; u < 4; ++u)
                        if (lo2[j][u] >= 0) *reinterpret_cast<uint4*>(smem + lo2[j][u]) = v2[j][u];
            }
#pragma unroll
            for (int u = 0; u < 8; ++u)
                if (lo[u] >= 0) *reinterpret_cast<uint4*>(sx0 + lo[u]) = v[u];
        }
    }
    __syncthreads();
    RSG_BSTAMP(1);
    // ---- 2. the 1x1 convs: this wave's jobs, two per round so that the weight buffers alternate with static indices ---------
    auto run_job = [&](const Job jb, const uint4 (&a)[16]) {
        const int q = jb.g * 32 + n;
        const char* px = smem + jb.src + (q < jb.np ? q : 0) * jb.ps + hh * 16;
        f32x16 acc;
#pragma unroll
        for (int r4 = 0; r4 < 4; ++r4) {
            const float4 bv = *reinterpret_cast<const float4*>(jb.bias + jb.ct * 32 + hh * 16 + r4 * 4);
            acc[r4 * 4 + 0] = bv.x; acc[r4 * 4 + 1] = bv.y; acc[r4 * 4 + 2] = bv.z; acc[r4 * 4 + 3] = bv.w;
        }
#pragma unroll
        for (int ks = 0; ks < 16; ++ks)
            if (ks < jb.ksn)                                    // wave-uniform
                acc = DT::mfma(__builtin_bit_cast(typename DT::vec8, a[ks]),
                               __builtin_bit_cast(typename DT::vec8, *reinterpret_cast<const uint4*>(px + ks * 32)), acc);
        if (q < jb.np) {
#pragma unroll
            for (int k = 0; k < 2; ++k) {
                f32x8 f;
#pragma unroll
                for (int e = 0; e < 8; ++e) f[e] = acc[k * 8 + e];
                *reinterpret_cast<uint4*>(smem + jb.dst + q * jb.psd + (jb.ct * 32 + hh * 16 + k * 8) * 2) =
                    __builtin_bit_cast(uint4, __builtin_convertvector(f, typename DT::vec8));
            }
        }
    };
    uint4 ad[18];                                               // the strided conv's weights travel under the 1x1 jobs
    {
        const uint4* wp = p.wd + (size_t)d_ct * 18 * 64 + lane;
#pragma unroll
        for (int ks = 0; ks < 18; ++ks) ad[ks] = wp[ks * 64];
    }
    for (int k = wave; k < njobs; k += 8) {
        const bool has1 = k + 4 < njobs, has2 = k + 8 < njobs;
        const Job j1 = job_of(has1 ? k + 4 : 0);
        fetch_w(j1, has1, wb);
        run_job(j0, wa);
        j0 = job_of(has2 ? k + 8 : 0);
        fetch_w(j0, has2, wa);
        if (has1) run_job(j1, wb);
    }
    __syncthreads();
    RSG_BSTAMP(2);
    // ---- 3. d10 (3x3, stride 2, pad 1, 32 -> 64) on the x0 band: waves 0/1 own pixel groups 0 and 2 (interleaved), waves 2/3
    // group 1; epilogue = the fuse sum of output 1 ---------------------------------------------------------------------------
    {
        const int np1 = 4 * W1;                                 // y1 pixels of the band
        const int ngd = (np1 + 31) >> 5;
        for (int g0 = wave >> 1; g0 < ngd; g0 += 4) {           // groups g0 and g0 + 2 (when it exists) together
            const bool two = g0 + 2 < ngd;
            int qv[2], pyv[2], pxv[2];
            const char* src[2];
            f32x16 acc[2];
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                qv[u] = (g0 + 2 * u) * 32 + n;
                const int qq = qv[u] < np1 ? qv[u] : 0;
                pyv[u] = qq / W1; pxv[u] = qq - pyv[u] * W1;
                // input pixel of tap (ky, kx): image (8 ty + 2 py + ky - 1, 2 px + kx - 1) = tile (2 py + ky, 2 px + kx)
                src[u] = sx0 + (2 * pyv[u]) * RP0 + (2 * pxv[u]) * PS0 + hh * 16;
#pragma unroll
                for (int r4 = 0; r4 < 4; ++r4) {
                    const float4 bv = *reinterpret_cast<const float4*>(p.bd + d_ct * 32 + hh * 16 + r4 * 4);
                    acc[u][r4 * 4 + 0] = bv.x; acc[u][r4 * 4 + 1] = bv.y; acc[u][r4 * 4 + 2] = bv.z; acc[u][r4 * 4 + 3] = bv.w;
                }
            }
#pragma unroll
            for (int ks = 0; ks < 18; ++ks) {                   // k order of the packed weights: tap-major, 16 channels per step
                const int t = ks >> 1, c = ks & 1;
                const int o = (t / 3) * RP0 + (t % 3) * PS0 + c * 32;
                acc[0] = DT::mfma(__builtin_bit_cast(typename DT::vec8, ad[ks]),
                                  __builtin_bit_cast(typename DT::vec8, *reinterpret_cast<const uint4*>(src[0] + o)), acc[0]);
                if (two)
                    acc[1] = DT::mfma(__builtin_bit_cast(typename DT::vec8, ad[ks]),
                                      __builtin_bit_cast(typename DT::vec8, *reinterpret_cast<const uint4*>(src[1] + o)), acc[1]);
            }
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                if ((u == 1 && !two) || qv[u] >= np1) continue;
                const int cb = d_ct * 32 + hh * 16, py = pyv[u], px = pxv[u], qq = qv[u];
                const size_t opix = (size_t)(b * H1 + ty * 4 + py) * W1 + px;
#pragma unroll
                for (int k = 0; k < 2; ++k) {
                    f32x8 f;
#pragma unroll
                    for (int e = 0; e < 8; ++e) f[e] = acc[u][k * 8 + e];
                    // the conv's own 16-bit result (what the grouped launch stores), then fuse_body's sum: same operands first
                    const typename DT::vec8 d16 = __builtin_convertvector(f, typename DT::vec8);
                    const typename DT::vec8 x1v = __builtin_bit_cast(typename DT::vec8, *reinterpret_cast<const uint4*>(smem + off_x[0] + qq * psx[0] + (cb + k * 8) * 2));
                    // a module with two branches: output 1 is the LAST output, whose sum the grouped launch folds into the conv's
                    // epilogue (acc + x1 in fp32, no 16-bit stop in between: plan.py fold_into) -- the same here
                    float s[8];
#pragma unroll
                    for (int e = 0; e < 8; ++e) s[e] = p.n_low == 1 ? f[e] + (float)x1v[e] : (0.f + (float)d16[e]) + (float)x1v[e];
#pragma unroll
                    for (int j = 1; j < 3; ++j) {
                        if (j >= p.n_low) break;
                        const typename DT::vec8 av = __builtin_bit_cast(typename DT::vec8, *reinterpret_cast<const uint4*>(
                            smem + off_a1[j - 1] + ((py >> j) * (W1 >> j) + (px >> j)) * PS1 + (cb + k * 8) * 2));
#pragma unroll
                        for (int e = 0; e < 8; ++e) s[e] += (float)av[e];
                    }
#pragma unroll
                    for (int e = 0; e < 8; ++e) f[e] = s[e];
                    *reinterpret_cast<uint4*>(p.y1 + (opix * p.ldy1 + cb + k * 8) * 2) = __builtin_bit_cast(uint4, DT::relu_pack(f));
                }
            }
        }
    }
    RSG_BSTAMP(3);
    // ---- 4. output 0: x0 + the up-sampled a0j (fuse_body) ----------------------------------------------------------------
    for (int id = tid; id < 8 * W * 4; id += 256) {
        const int pl = id >> 2, cg = id & 3;
        const int y = pl / W, x = pl - y * W;
        float acc[8];
        {
            const typename DT::vec8 v = __builtin_bit_cast(typename DT::vec8, *reinterpret_cast<const uint4*>(sx0 + (y + 1) * RP0 + (x + 1) * PS0 + cg * 16));
#pragma unroll
            for (int e = 0; e < 8; ++e) acc[e] = 0.f + (float)v[e];
        }
#pragma unroll
        for (int j = 0; j < 3; ++j) {
            if (j >= p.n_low) break;
            const int s = j + 1;
            const typename DT::vec8 v = __builtin_bit_cast(typename DT::vec8, *reinterpret_cast<const uint4*>(
                smem + off_a0[j] + ((y >> s) * (W >> s) + (x >> s)) * PS0 + cg * 16));
#pragma unroll
            for (int e = 0; e < 8; ++e) acc[e] += (float)v[e];
        }
        f32x8 f;
#pragma unroll
        for (int e = 0; e < 8; ++e) f[e] = acc[e];
        *reinterpret_cast<uint4*>(p.y0 + (((size_t)(b * p.H + ty * 8) * W + pl) * p.ldy0 + cg * 8) * 2) = __builtin_bit_cast(uint4, DT::relu_pack(f));
    }
    RSG_BSTAMP(4);
}

// xs[j] / ld[j]: branch j's map, [batch, h >> j, w >> j, ld[j]], 32 << j channels, j = 0 .. n_low.  w_a0 / b_a0 [n_low]: the 1x1
// convs j -> 0 (cout 32); w_d10 / b_d10: the 3x3 stride-2 conv 0 -> 1 (cout 64); w_a1 / b_a1 [n_low - 1]: the 1x1 convs
// j -> 1, j >= 2 (cout 64); all from rsg_pack_conv_weights_host with the BatchNorm folded.  y0: [batch,h,w,ld_y0] 32 channels,
// y1: [batch,h/2,w/2,ld_y1] 64 channels.  h, w multiples of 8.
extern "C" int rsg_fuse_band(const void* const* xs, const int32_t* ld, int32_t n_low, const void* const* w_a0, const float* const* b_a0,
                             const void* w_d10, const float* b_d10, const void* const* w_a1, const float* const* b_a1, void* y0,
                             int32_t ld_y0, void* y1, int32_t ld_y1, int32_t batch, int32_t h, int32_t w, int32_t dtype, void* stream) {
    if (!xs || !ld || !w_a0 || !b_a0 || !w_d10 || !b_d10 || !y0 || !y1 || n_low < 1 || n_low > 3 || batch < 1) RSG_FAIL("fuse_band: bad arguments");
    if ((h & 7) || (w & 7) || ld_y0 < 32 || (ld_y0 & 7) || ld_y1 < 64 || (ld_y1 & 7) || (((uintptr_t)y0 | (uintptr_t)y1 | (uintptr_t)w_d10 | (uintptr_t)b_d10) & 15))
        RSG_FAIL("fuse_band: shape / alignment");
    if ((long long)batch * h * w * ld[0] * 2 >= (1LL << 31)) RSG_FAIL("fuse_band: map too large");
    FuseBandArgs p;
    memset(&p, 0, sizeof(p));
    p.B = batch; p.H = h; p.W = w; p.n_low = n_low; p.y0 = (char*)y0; p.ldy0 = ld_y0; p.y1 = (char*)y1; p.ldy1 = ld_y1;
    p.wd = (const uint4*)w_d10; p.bd = b_d10;
    int lds = 9 * (w + 1) * 80;
    for (int j = 0; j <= n_low; ++j) {
        const int c = 32 << j;
        if (!xs[j] || ld[j] < c || (ld[j] & 7) || ((uintptr_t)xs[j] & 15)) RSG_FAIL("fuse_band: operand %d", j);
        p.x[j] = (const char*)xs[j]; p.ld[j] = ld[j];
    }
    for (int j = 0; j < n_low; ++j) {
        if (!w_a0[j] || !b_a0[j] || (((uintptr_t)w_a0[j] | (uintptr_t)b_a0[j]) & 15)) RSG_FAIL("fuse_band: weights of the 1x1 conv %d -> 0", j + 1);
        p.w0[j] = (const uint4*)w_a0[j]; p.b0[j] = b_a0[j];
        const int npx = (8 >> (j + 1)) * (w >> (j + 1));
        lds += npx * ((64 << j) * 2 + 16) + npx * 80;
        if (j >= 1) {
            if (!w_a1 || !b_a1 || !w_a1[j - 1] || !b_a1[j - 1] || (((uintptr_t)w_a1[j - 1] | (uintptr_t)b_a1[j - 1]) & 15))
                RSG_FAIL("fuse_band: weights of the 1x1 conv %d -> 1", j + 1);
            p.w1[j - 1] = (const uint4*)w_a1[j - 1]; p.b1[j - 1] = b_a1[j - 1];
            lds += npx * 144;
        }
    }
    if (lds > 150 * 1024) RSG_FAIL("fuse_band: a %d-pixel-wide map needs %d B of LDS", w, lds);
    const void* fn = dtype == RSG_BF16 ? (const void*)fuse_band_kernel<BF16> : (const void*)fuse_band_kernel<F16>;
    static int lds_set[2] = {0, 0};                            // per dtype: the attribute call is host work on every launch otherwise
    if (lds_set[dtype == RSG_BF16] < lds) {
        if (hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, lds) != hipSuccess) RSG_FAIL("fuse_band: cannot reserve %d B of LDS", lds);
        lds_set[dtype == RSG_BF16] = lds;
    }
    const dim3 grid(batch * (h / 8));
    if (dtype == RSG_BF16) RSG_LAUNCH(fuse_band_kernel<BF16>, grid, dim3(256), lds, (hipStream_t)stream, p);
    else RSG_LAUNCH(fuse_band_kernel<F16>, grid, dim3(256), lds, (hipStream_t)stream, p);
    RSG_CHECK_LAUNCH("fuse_band");
    return 0;
}
