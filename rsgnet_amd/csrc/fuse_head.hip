// The last HighResolutionModule's fuse sum for branch 0 (pose_hrnet.py:259-266) and the 1x1 classifier on it
// (pose_hrnet.py:326-332,461) in ONE launch: y0 = ReLU(x0 + up(a1) + up(a2) + up(a3)) is summed while a workgroup stages
// its 128 pixels (the operand order and the 16-bit rounding of rsg_hr_fuse_sum / fuse_body), goes to LDS and is the B
// operand of the classifier's two k-steps; the heat maps leave as fp32 NCHW.  The 12.6 MB map y0 is neither written nor
// read back, and one launch goes.  Bit-identical to rsg_conv_fuse_group_run (the sum) followed by rsg_conv_igemm.
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
