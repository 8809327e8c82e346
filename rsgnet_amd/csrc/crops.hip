// Crop producer: the datasets' `cv2.warpAffine(img, trans, image_size, INTER_LINEAR)` followed by
// ToTensor + Normalize (lib/dataset/CPJointsDataset.py:1281-1290, tools/test.py:105-112), for a whole batch of
// person boxes in one launch, written straight as the fp32 NCHW tensor the stem kernel reads.
//
// The arithmetic is OpenCV's fixed-point bilinear warp (imgwarp.cpp, the classic WarpAffineInvoker +
// remapBilinear<FixedPtCast<int, uchar, 15>> path), restated:
//   * the 2x3 map is inverted in double;
//   * source coordinates are integers in 1/1024 px: rint(M0*x*1024) + rint((M1*y+M2)*1024) + 16, cut to 1/32 px;
//   * the four taps are weighted by (32-fx)(32-fy)*32 ... (sum 2^15, the unit weight saturating to 32767) and the
//     sum is rounded with +2^14 >> 15; taps outside the picture count as 0 (BORDER_CONSTANT, value 0).
// HBM-bound: 3 bytes read x 4 taps (L2 hits) and 12 bytes written per pixel; one thread per output pixel.
#include "common.h"

// OpenCV computes the map with separately rounded multiplies and adds: this file is built with -ffp-contract=off
// (rsgnet_amd/build.py); a fused multiply-add moves a coordinate across a rounding tie once in ~10^5 pixels.

__device__ __forceinline__ int sat_rint(double v) {
    // cv::saturate_cast<int>(double): round half to even, saturating
    if (v >= 2147483647.0) return 2147483647;
    if (v <= -2147483648.0) return (int)0x80000000;
    return __double2int_rn(v);
}

__global__ void __launch_bounds__(256) crop_warp_kernel(const rsg_crop_desc* __restrict__ crops,
                                                        float* __restrict__ out, float* __restrict__ out_flip, int out_h, int out_w,
                                                        int swap_rb,
                                                        float mean0, float mean1, float mean2, float std0,
                                                        float std1, float std2) {
    const int x = blockIdx.x * 64 + threadIdx.x;            // a wave covers 64 pixels of one row
    const int y = blockIdx.y * 4 + threadIdx.y, b = blockIdx.z;
    if (x >= out_w || y >= out_h) return;
    const rsg_crop_desc d = crops[b];
    // inverse map, the operation order of cv::warpAffine; no fused multiply-adds (see above)
    const double xd = (double)x, yd = (double)y;
    double D = d.m[0] * d.m[4] - d.m[1] * d.m[3];
    D = D != 0.0 ? 1.0 / D : 0.0;
    const double i0 = d.m[4] * D, i4 = d.m[0] * D, i1 = d.m[1] * -D, i3 = d.m[3] * -D;
    const double i2 = -i0 * d.m[2] - i1 * d.m[5];
    const double i5 = -i3 * d.m[2] - i4 * d.m[5];
    const int adelta = sat_rint(i0 * xd * 1024.0);
    const int bdelta = sat_rint(i3 * xd * 1024.0);
    const int X0 = sat_rint((i1 * yd + i2) * 1024.0) + 16;
    const int Y0 = sat_rint((i4 * yd + i5) * 1024.0) + 16;
    const int X = (int)((unsigned)X0 + (unsigned)adelta) >> 5;      // 1/32 px
    const int Y = (int)((unsigned)Y0 + (unsigned)bdelta) >> 5;
    int sx = X >> 5, sy = Y >> 5;
    sx = min(max(sx, -32768), 32767);
    sy = min(max(sy, -32768), 32767);
    const int fx = X & 31, fy = Y & 31;
    const int w00 = min((32 - fx) * (32 - fy) * 32, 32767), w01 = fx * (32 - fy) * 32;
    const int w10 = (32 - fx) * fy * 32, w11 = fx * fy * 32;
    int acc[3] = {0, 0, 0};
    const bool x0 = (unsigned)sx < (unsigned)d.width, x1 = (unsigned)(sx + 1) < (unsigned)d.width;
    const bool y0 = (unsigned)sy < (unsigned)d.height, y1 = (unsigned)(sy + 1) < (unsigned)d.height;
    const uint8_t* r0 = (const uint8_t*)d.image + (long)sy * d.stride + (long)sx * 3;
    const uint8_t* r1 = r0 + d.stride;
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        const int v00 = (x0 && y0) ? r0[k] : 0, v01 = (x1 && y0) ? r0[3 + k] : 0;
        const int v10 = (x0 && y1) ? r1[k] : 0, v11 = (x1 && y1) ? r1[3 + k] : 0;
        acc[k] = (v00 * w00 + v01 * w01 + v10 * w10 + v11 * w11 + (1 << 14)) >> 15;
    }
    const float mean[3] = {mean0, mean1, mean2}, sd[3] = {std0, std1, std2};
    const long plane = (long)out_h * out_w;
    float* o = out + ((long)b * 3) * plane + (long)y * out_w + x;
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        const int v = min(swap_rb ? acc[2 - k] : acc[k], 255);
        // ToTensor: uint8 -> float / 255; Normalize: (t - mean) / std, each correctly rounded in fp32
        const float t = ((float)v / 255.f - mean[k]) / sd[k];
        o[k * plane] = t;
        if (out_flip) out_flip[((long)b * 3 + k) * plane + (long)y * out_w + (out_w - 1 - x)] = t;   // x.flip(3)
    }
}

extern "C" int rsg_crop_warp_affine(const struct rsg_crop_desc* crops_dev, float* out_nchw, float* out_flipped_nchw,
                                    int32_t batch, int32_t out_h, int32_t out_w, const float* mean3_host,
                                    const float* std3_host, int32_t swap_rb, void* stream) {
    if (!crops_dev || !out_nchw || !mean3_host || !std3_host) RSG_FAIL("crop_warp_affine: null pointer");
    if (batch < 1 || batch > 65535 || out_h < 1 || out_h > 262140 || out_w < 1 || out_w > 32767)
        RSG_FAIL("crop_warp_affine: batch=%d size=%dx%d out of range", batch, out_h, out_w);
    for (int k = 0; k < 3; ++k)
        if (!(std3_host[k] != 0.f)) RSG_FAIL("crop_warp_affine: std[%d] is zero", k);
    static_assert(sizeof(rsg_crop_desc) == 72, "rsg_crop_desc layout");
    RSG_LAUNCH(crop_warp_kernel, dim3((out_w + 63) / 64, (out_h + 3) / 4, batch), dim3(64, 4), 0, (hipStream_t)stream,
               crops_dev, out_nchw, out_flipped_nchw, out_h, out_w, swap_rb ? 1 : 0, mean3_host[0],
               mean3_host[1], mean3_host[2], std3_host[0], std3_host[1], std3_host[2]);
    RSG_CHECK_LAUNCH("crop_warp_affine");
    return 0;
}
