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
    // the attribute is per device (a process may drive several: nn.DataParallel replicas); racing threads set the same value
    static int lds_set[RSG_MAX_DEVICES][2];
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= RSG_MAX_DEVICES) RSG_FAIL("fuse_head4: no current device");
    if (lds_set[dev][dtype == RSG_BF16] < lds) {
        if (hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, lds) != hipSuccess) RSG_FAIL("fuse_head4: cannot reserve %d B of LDS", lds);
        lds_set[dev][dtype == RSG_BF16] = lds;
    }
    const dim3 grid(batch * (h / 8));
    if (dtype == RSG_BF16) RSG_LAUNCH(fuse_head4_kernel<BF16>, grid, dim3(256), lds, (hipStream_t)stream, p);
    else RSG_LAUNCH(fuse_head4_kernel<F16>, grid, dim3(256), lds, (hipStream_t)stream, p);
    RSG_CHECK_LAUNCH("fuse_head4");
    return 0;
}
