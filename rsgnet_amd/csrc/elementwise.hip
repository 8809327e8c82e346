// Bandwidth-bound helpers around the convolution kernels (gfx950): 16 bytes per lane everywhere.
#include "common.h"
#include "fuse_body.h"

// ---- NCHW fp32 -> NHWC 16-bit, channels zero-padded (stem input; pose_hrnet.py:429) ---------------
// One thread per pixel: reads c planes (coalesced across the wave along W), writes c_pad*2 bytes.
template <class DT, int CPAD>
__global__ void __launch_bounds__(256) nchw_to_nhwc16_kernel(const float* __restrict__ in, char* __restrict__ out,
                                                            int c, int hw, long total) {
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i >= total) return;
    const long b = i / hw;
    const int s = (int)(i - b * hw);
    typename DT::elem v[CPAD];
#pragma unroll
    for (int k = 0; k < CPAD; ++k) v[k] = (typename DT::elem)0.f;
#pragma unroll
    for (int k = 0; k < CPAD; ++k)
        if (k < c) v[k] = (typename DT::elem)in[(b * c + k) * hw + s];
    uint4* o = reinterpret_cast<uint4*>(out + i * CPAD * 2);
#pragma unroll
    for (int k = 0; k < CPAD / 8; ++k) {
        typename DT::vec8 t;
#pragma unroll
        for (int e = 0; e < 8; ++e) t[e] = v[k * 8 + e];
        o[k] = __builtin_bit_cast(uint4, t);
    }
}

extern "C" int rsg_nchw_f32_to_nhwc16(const float* in, void* out, int32_t batch, int32_t c, int32_t h,
                                      int32_t w, int32_t c_pad, int32_t dtype, void* stream) {
    if (!in || !out) RSG_FAIL("nchw_to_nhwc16: null pointer");
    if ((c_pad != 16 && c_pad != 32) || c > c_pad || c < 1) RSG_FAIL("nchw_to_nhwc16: c=%d c_pad=%d (c_pad 16 or 32)", c, c_pad);
    const long total = (long)batch * h * w;
    const int grid = (int)((total + 255) / 256);
    if (c_pad == 32) {       // the heat-map gradient (17 / 14 joints) on its way into the backward convs
        if (dtype == RSG_BF16)
            RSG_LAUNCH((nchw_to_nhwc16_kernel<BF16, 32>), dim3(grid), dim3(256), 0, (hipStream_t)stream, in, (char*)out, c, h * w, total);
        else
            RSG_LAUNCH((nchw_to_nhwc16_kernel<F16, 32>), dim3(grid), dim3(256), 0, (hipStream_t)stream, in, (char*)out, c, h * w, total);
        RSG_CHECK_LAUNCH("nchw_to_nhwc16");
        return 0;
    }
    if (dtype == RSG_BF16)
        RSG_LAUNCH((nchw_to_nhwc16_kernel<BF16, 16>), dim3(grid), dim3(256), 0, (hipStream_t)stream,
                           in, (char*)out, c, h * w, total);
    else
        RSG_LAUNCH((nchw_to_nhwc16_kernel<F16, 16>), dim3(grid), dim3(256), 0, (hipStream_t)stream,
                           in, (char*)out, c, h * w, total);
    RSG_CHECK_LAUNCH("nchw_to_nhwc16");
    return 0;
}

// ---- NCHW fp32 -> space-to-depth(2) NHWC 16-bit: out[b, Y, X, (py*2+px)*c + k] = in[b, k, 2Y+py, 2X+px] -------
// With 3 input channels that is 12 of 16 channels per half-resolution pixel: the stem's 3x3 stride-2
// conv1 (pose_hrnet.py:285-286) becomes a 2x2 stride-1 conv over this map (4 k-steps instead of 9, a
// quarter of the bytes of the channel-padded full-resolution layout).
template <class DT>
__global__ void __launch_bounds__(256) nchw_to_s2d16_kernel(const float* __restrict__ in, char* __restrict__ out,
                                                           int c, int h, int w, long total) {
    // 32-bit index arithmetic throughout: a 64-bit division costs ~100 instructions on this hardware
    const unsigned i = blockIdx.x * 256u + threadIdx.x;       // one thread per output (half-res) pixel
    if (i >= total) return;
    const int w2 = w / 2, h2 = h / 2;
    const unsigned t = i / (unsigned)w2;
    const int X = (int)(i - t * w2);
    const unsigned b = t / (unsigned)h2;
    const int Y = (int)(t - b * h2);
    typename DT::elem v[16];
#pragma unroll
    for (int k = 0; k < 16; ++k) v[k] = (typename DT::elem)0.f;
    for (int k = 0; k < c; ++k) {
        const float* pl = in + ((b * c + k) * h + 2 * Y) * (long)w + 2 * X;
        const float2 r0 = *reinterpret_cast<const float2*>(pl);
        const float2 r1 = *reinterpret_cast<const float2*>(pl + w);
        v[0 * c + k] = (typename DT::elem)r0.x;
        v[1 * c + k] = (typename DT::elem)r0.y;
        v[2 * c + k] = (typename DT::elem)r1.x;
        v[3 * c + k] = (typename DT::elem)r1.y;
    }
    uint4* o = reinterpret_cast<uint4*>(out + i * 32);
#pragma unroll
    for (int q = 0; q < 2; ++q) {
        typename DT::vec8 p8;
#pragma unroll
        for (int e = 0; e < 8; ++e) p8[e] = v[q * 8 + e];
        o[q] = __builtin_bit_cast(uint4, p8);
    }
}

extern "C" int rsg_nchw_f32_to_s2d_nhwc16(const float* in, void* out, int32_t batch, int32_t c, int32_t h,
                                          int32_t w, int32_t dtype, void* stream) {
    if (!in || !out) RSG_FAIL("nchw_to_s2d: null pointer");
    if (c < 1 || c > 4 || (h & 1) || (w & 1) || ((uintptr_t)in & 7)) RSG_FAIL("nchw_to_s2d: c=%d (<=4), even h,w required", c);
    const long total = (long)batch * (h / 2) * (w / 2);
    const int grid = (int)((total + 255) / 256);
    if (dtype == RSG_BF16)
        RSG_LAUNCH(nchw_to_s2d16_kernel<BF16>, dim3(grid), dim3(256), 0, (hipStream_t)stream, in, (char*)out, c, h, w, total);
    else
        RSG_LAUNCH(nchw_to_s2d16_kernel<F16>, dim3(grid), dim3(256), 0, (hipStream_t)stream, in, (char*)out, c, h, w, total);
    RSG_CHECK_LAUNCH("nchw_to_s2d");
    return 0;
}

// ---- HighResolutionModule fuse (pose_hrnet.py:259-266) ------------------------------------------------
template <class DT>
__global__ void __launch_bounds__(256) hr_fuse_kernel(const FuseArgs a) {
    fuse_body<DT>(a, blockIdx.x, gridDim.x);
}

int rsg_fuse_prepare(const void* const* same, int n_same, const void* const* low, const int32_t* low_shift,
                     int n_low, void* out, int batch, int h, int w, int c, FuseArgs& a, int& blocks,
                     const int32_t* same_ld, const int32_t* low_ld) {
    if (n_same < 1 || n_same > 4 || n_low < 0 || n_low > 3 || !out) RSG_FAIL("hr_fuse: bad operand count");
    if (c % 8) RSG_FAIL("hr_fuse: channels %d not a multiple of 8", c);
    memset(&a, 0, sizeof(a));
    for (int k = 0; k < n_same; ++k) {
        a.same[k] = (const char*)same[k];
        const int ld = same_ld && same_ld[k] ? same_ld[k] : c;
        if (ld < c || ld % 8 || ((uintptr_t)same[k] & 15)) RSG_FAIL("hr_fuse: operand %d stride %d / alignment", k, ld);
        a.ld_same[k] = ld / 8;
    }
    for (int k = 0; k < n_low; ++k) {
        a.low[k] = (const char*)low[k];
        a.shift[k] = low_shift[k];
        const int ld = low_ld && low_ld[k] ? low_ld[k] : c;
        if (ld < c || ld % 8 || ((uintptr_t)low[k] & 15)) RSG_FAIL("hr_fuse: low operand %d stride %d / alignment", k, ld);
        a.ld_low[k] = ld / 8;
        if ((h & ((1 << a.shift[k]) - 1)) || (w & ((1 << a.shift[k]) - 1)) || a.shift[k] < 1)
            RSG_FAIL("hr_fuse: %dx%d not divisible by 2^%d", h, w, a.shift[k]);
    }
    a.n_same = n_same; a.n_low = n_low; a.out = (char*)out;
    a.h = h; a.w = w; a.c8 = c / 8;
    a.total = (long)batch * h * w * a.c8;
    if (a.total >= (1L << 31)) RSG_FAIL("hr_fuse: too many elements for 32-bit indexing");
    long b = (a.total + 255) / 256;
    if (b > 256 * 16) b = 256 * 16;
    blocks = (int)b;
    return 0;
}

extern "C" int rsg_hr_fuse_sum(const void* const* same, int32_t n_same, const void* const* low,
                               const int32_t* low_shift, int32_t n_low, void* out, int32_t batch,
                               int32_t h, int32_t w, int32_t c, int32_t dtype, void* stream) {
    FuseArgs a;
    int blocks = 0;
    if (rsg_fuse_prepare(same, n_same, low, low_shift, n_low, out, batch, h, w, c, a, blocks)) return 1;
    if (dtype == RSG_BF16)
        RSG_LAUNCH(hr_fuse_kernel<BF16>, dim3(blocks), dim3(256), 0, (hipStream_t)stream, a);
    else
        RSG_LAUNCH(hr_fuse_kernel<F16>, dim3(blocks), dim3(256), 0, (hipStream_t)stream, a);
    RSG_CHECK_LAUNCH("hr_fuse_sum");
    return 0;
}

// ---- bilinear x2, align_corners=True, optional sigmoid, fp32 NCHW (pose_rsgnet.py:1009-1013) -----------
// src = dst * (in-1)/(out-1); out = 2*in.  A workgroup owns BL_ROWS output rows of one plane: it stages the source rows they
// read (about half as many) in LDS with coalesced loads, then every thread produces float4s of output -- four LDS reads and
// three lerps per output.  (Round 5: the one-thread-per-float4 form with its operands in registers selected by compare chains
// was 370 instructions per float4, branches included, and bound by instruction issue at 2.7 TB/s of stores; this form is a
// quarter of that.)  Up to two maps per launch (RSGNet's multi-joint scores and its limb maps, the latter through a sigmoid):
// the planes of map b follow those of map a in the grid.  Expressions and their association are ATen's upsample_bilinear2d.
#ifndef BL_ROWS
#define BL_ROWS 64
#endif
struct BilinearArgs {
    const float* in[2];
    float* out[2];
    int planes[2], sigmoid[2];
    int h, w, bands;
};

__global__ void __launch_bounds__(256) bilinear2x_kernel(const BilinearArgs a) {
    extern __shared__ float bl_rows[];                   // [(BL_ROWS / 2 + 2) source rows][w]
    const int h = a.h, w = a.w, ow = 2 * w, oh = 2 * h;
    const float ry = oh > 1 ? (float)(h - 1) / (float)(oh - 1) : 0.f;
    const float rx = ow > 1 ? (float)(w - 1) / (float)(ow - 1) : 0.f;
    int pl = blockIdx.x / a.bands;
    const int band = blockIdx.x - pl * a.bands;
    const int m = pl >= a.planes[0] ? 1 : 0;             // which map (block-uniform)
    pl -= m ? a.planes[0] : 0;
    const float* in = (m ? a.in[1] : a.in[0]) + (size_t)pl * h * w;
    float* out = (m ? a.out[1] : a.out[0]) + (size_t)pl * oh * ow;
    const int sig = m ? a.sigmoid[1] : a.sigmoid[0];
    const int oy0 = band * BL_ROWS, oy1 = min(oy0 + BL_ROWS, oh);
    const int ys0 = (int)(ry * (float)oy0);              // first source row any output row of the band reads
    int ys1 = (int)(ry * (float)(oy1 - 1)) + 1;          // last one
    if (ys1 > h - 1) ys1 = h - 1;
    const int nsrc = (ys1 - ys0 + 1) * w;
    for (int i = threadIdx.x; i < nsrc; i += 256) bl_rows[i] = in[ys0 * w + i];
    __syncthreads();
    // thread (row lane 0..7, x lane 0..31): a fixed float4 column, rows rl, rl + 8, ... of the band -- the column's source
    // indices and weights are computed once, the row loop has no division (the per-item form, (row, x4) = divmod(item, ow / 4)
    // for every float4, left the kernel at 2.7 TB/s of stores where a fill reaches 6.5: tools/probes/store_bandwidth.py)
    const int ow4 = ow >> 2, xl = threadIdx.x & 31, rl = threadIdx.x >> 5;
    for (int x4 = xl; x4 < ow4; x4 += 32) {
        int ia[4], ib[4];
        float wx[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int ox = x4 * 4 + k;
            const float sx = rx * ox;
            int xa = (int)sx;
            if (xa > w - 1) xa = w - 1;
            ia[k] = xa;
            ib[k] = xa < w - 1 ? xa + 1 : xa;
            wx[k] = sx - (float)xa;
        }
#pragma unroll 2
        for (int oy = oy0 + rl; oy < oy1; oy += 8) {
            const float sy = ry * oy;
            int y0 = (int)sy;
            if (y0 > h - 1) y0 = h - 1;
            const int y1 = y0 < h - 1 ? y0 + 1 : y0;
            const float wy = sy - (float)y0;
            const float* r0 = bl_rows + (y0 - ys0) * w;
            const float* r1 = bl_rows + (y1 - ys0) * w;
            float o[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                // same association as ATen's upsample_bilinear2d: lerp in x on both rows, then in y
                const float top = r0[ia[k]] * (1.f - wx[k]) + r0[ib[k]] * wx[k];
                const float bot = r1[ia[k]] * (1.f - wx[k]) + r1[ib[k]] * wx[k];
                float v = top * (1.f - wy) + bot * wy;
                if (sig) v = __builtin_amdgcn_rcpf(1.f + __builtin_amdgcn_exp2f(v * -1.4426950408889634f));
                o[k] = v;
            }
            *reinterpret_cast<float4*>(out + (size_t)oy * ow + x4 * 4) = make_float4(o[0], o[1], o[2], o[3]);
        }
    }
}

static int bilinear_launch(const float* in_a, float* out_a, int planes_a, int sig_a, const float* in_b, float* out_b, int planes_b,
                           int sig_b, int h, int w, void* stream, const char* what) {
    if (!in_a || !out_a || planes_a < 1 || h < 1 || w < 2 || (w % 2) || planes_b < 0 || (planes_b && (!in_b || !out_b)))
        RSG_FAIL("%s: bad arguments", what);
    if ((((uintptr_t)out_a | (uintptr_t)out_b) & 15)) RSG_FAIL("%s: outputs must be 16-byte aligned", what);
    BilinearArgs a;
    memset(&a, 0, sizeof(a));
    a.in[0] = in_a; a.out[0] = out_a; a.planes[0] = planes_a; a.sigmoid[0] = sig_a ? 1 : 0;
    a.in[1] = in_b; a.out[1] = out_b; a.planes[1] = planes_b; a.sigmoid[1] = sig_b ? 1 : 0;
    a.h = h; a.w = w; a.bands = ceil_div(2 * h, BL_ROWS);
    const long blocks = (long)(planes_a + planes_b) * a.bands;
    const int lds = (BL_ROWS / 2 + 2) * w * (int)sizeof(float);
    if (blocks >= (1L << 31) || lds > 64 * 1024) RSG_FAIL("%s: map too large (%ld workgroups, %d B of LDS)", what, blocks, lds);
    RSG_LAUNCH(bilinear2x_kernel, dim3((unsigned)blocks), dim3(256), lds, (hipStream_t)stream, a);
    RSG_CHECK_LAUNCH(what);
    return 0;
}

extern "C" int rsg_bilinear2x_nchw_f32(const float* in, float* out, int32_t planes, int32_t h, int32_t w,
                                       int32_t sigmoid, void* stream) {
    return bilinear_launch(in, out, planes, sigmoid, nullptr, nullptr, 0, 0, h, w, stream, "bilinear2x");
}

extern "C" int rsg_bilinear2x_pair_nchw_f32(const float* in_a, float* out_a, int32_t planes_a, int32_t sigmoid_a, const float* in_b,
                                            float* out_b, int32_t planes_b, int32_t sigmoid_b, int32_t h, int32_t w, void* stream) {
    if (planes_b < 1) RSG_FAIL("bilinear2x_pair: the second map has no planes");
    return bilinear_launch(in_a, out_a, planes_a, sigmoid_a, in_b, out_b, planes_b, sigmoid_b, h, w, stream, "bilinear2x_pair");
}

// ---- backward of the above (training: pose_rsgnet.py:1009-1013 under autograd) -------------------------------------------
// din[pl,y,x] = sum over the output pixels that read (y,x) of their bilinear weight * g, g = dout (* p (1 - p) behind the
// sigmoid, p = the forward's output).  Gather form, fixed order: one thread per INPUT pixel walks the <= 6 x 6 output pixels
// whose source window can touch it and recomputes their weights with the forward's own float expressions.
__global__ void __launch_bounds__(256) bilinear2x_backward_kernel(const float* __restrict__ dout, const float* __restrict__ prob,
                                                                 float* __restrict__ din, int h, int w, long total) {
    const int ow = 2 * w, oh = 2 * h;
    const float ry = oh > 1 ? (float)(h - 1) / (float)(oh - 1) : 0.f;
    const float rx = ow > 1 ? (float)(w - 1) / (float)(ow - 1) : 0.f;
    for (unsigned i = blockIdx.x * 256u + threadIdx.x; i < (unsigned)total; i += gridDim.x * 256u) {
        const unsigned t = i / (unsigned)w;
        const int x = (int)(i - t * w);
        const unsigned pl = t / (unsigned)h;
        const int y = (int)(t - pl * h);
        // output rows / columns whose floor(source) is y-1 or y (x-1 or x): source = r * o, r < 1/2
        const int oy_lo = ry > 0.f ? max((int)floorf((float)(y - 1) / ry), 0) : 0;
        const int oy_hi = ry > 0.f ? min((int)ceilf((float)(y + 1) / ry), oh - 1) : oh - 1;
        const int ox_lo = rx > 0.f ? max((int)floorf((float)(x - 1) / rx), 0) : 0;
        const int ox_hi = rx > 0.f ? min((int)ceilf((float)(x + 1) / rx), ow - 1) : ow - 1;
        const float* gp = dout + (size_t)pl * oh * ow;
        const float* pp = prob ? prob + (size_t)pl * oh * ow : nullptr;
        float acc = 0.f;
        for (int oy = oy_lo; oy <= oy_hi; ++oy) {
            const float sy = ry * oy;
            int y0 = (int)sy;
            if (y0 > h - 1) y0 = h - 1;
            const int y1 = y0 < h - 1 ? y0 + 1 : y0;
            const float wy = sy - (float)y0;
            const float cy = (y0 == y ? 1.f - wy : 0.f) + (y1 == y ? wy : 0.f);
            if (cy == 0.f) continue;
            float row = 0.f;
            for (int ox = ox_lo; ox <= ox_hi; ++ox) {
                const float sx = rx * ox;
                int xa = (int)sx;
                if (xa > w - 1) xa = w - 1;
                const int xb = xa < w - 1 ? xa + 1 : xa;
                const float wx = sx - (float)xa;
                const float cx = (xa == x ? 1.f - wx : 0.f) + (xb == x ? wx : 0.f);
                if (cx == 0.f) continue;
                float g = gp[(size_t)oy * ow + ox];
                if (pp) {
                    const float p = pp[(size_t)oy * ow + ox];
                    g *= p * (1.f - p);
                }
                row += cx * g;
            }
            acc += cy * row;
        }
        din[i] = acc;
    }
}

extern "C" int rsg_bilinear2x_backward_nchw_f32(const float* dout, const float* prob, float* din, int32_t planes, int32_t h,
                                                int32_t w, void* stream) {
    if (!dout || !din || planes < 1 || h < 1 || w < 2 || (w % 2)) RSG_FAIL("bilinear2x_backward: bad arguments");
    const long total = (long)planes * h * w;
    if (total * 4 >= (1L << 31)) RSG_FAIL("bilinear2x_backward: too many elements for 32-bit indexing");
    long blocks = (total + 255) / 256;
    if (blocks > 256 * 32) blocks = 256 * 32;
    RSG_LAUNCH(bilinear2x_backward_kernel, dim3((int)blocks), dim3(256), 0, (hipStream_t)stream, dout, prob, din, h, w, total);
    RSG_CHECK_LAUNCH("bilinear2x_backward");
    return 0;
}

// ---- flip-test merge (lib/core/function.py:400-427, lib/utils/transforms.py:23-37) ---------------------
// out = 0.5 * (h + shift(flip_back(hf))): flip_back reverses W and swaps the left/right joint channels,
// SHIFT_HEATMAP moves the flipped map one pixel to the right (column 0 keeps its value).  The reference
// does this on the host through numpy; here it is one fp32 NCHW pass.
struct FlipArgs {
    const float* h;
    const float* hf;
    float* out;
    int J, H, W, shift;
    long total;
    int perm[64];
};

__global__ void __launch_bounds__(256) flip_merge_kernel(const FlipArgs a) {
    for (unsigned i = blockIdx.x * 256u + threadIdx.x; i < (unsigned)a.total; i += gridDim.x * 256u) {
        unsigned t = i / (unsigned)a.W;
        const int x = (int)(i - t * a.W);
        unsigned u = t / (unsigned)a.H;
        const int y = (int)(t - u * a.H);
        const unsigned b = u / (unsigned)a.J;
        const int j = (int)(u - b * a.J);
        const int xs = a.shift ? (x > 0 ? x - 1 : 0) : x;          // column of the un-shifted flipped-back map
        const size_t src = (((size_t)b * a.J + a.perm[j]) * a.H + y) * a.W + (a.W - 1 - xs);
        a.out[i] = 0.5f * (a.h[i] + a.hf[src]);
    }
}

extern "C" int rsg_flip_merge_nchw_f32(const float* h, const float* h_flipped, const int32_t* perm_host,
                                       float* out, int32_t batch, int32_t joints, int32_t height, int32_t width,
                                       int32_t shift_heatmap, void* stream) {
    if (!h || !h_flipped || !perm_host || !out) RSG_FAIL("flip_merge: null pointer");
    if (joints < 1 || joints > 64 || batch < 1 || height < 1 || width < 1) RSG_FAIL("flip_merge: bad shape");
    FlipArgs a;
    memset(&a, 0, sizeof(a));
    a.h = h; a.hf = h_flipped; a.out = out; a.J = joints; a.H = height; a.W = width; a.shift = shift_heatmap;
    a.total = (long)batch * joints * height * width;
    if (a.total >= (1L << 31)) RSG_FAIL("flip_merge: too many elements for 32-bit indexing");
    for (int j = 0; j < joints; ++j) {
        if (perm_host[j] < 0 || perm_host[j] >= joints) RSG_FAIL("flip_merge: perm[%d]=%d", j, perm_host[j]);
        a.perm[j] = perm_host[j];
    }
    long blocks = (a.total + 255) / 256;
    if (blocks > 256 * 16) blocks = 256 * 16;
    RSG_LAUNCH(flip_merge_kernel, dim3((int)blocks), dim3(256), 0, (hipStream_t)stream, a);
    RSG_CHECK_LAUNCH("flip_merge");
    return 0;
}

// ---- heatmap decode (lib/core/inference.py:21-82, lib/utils/transforms.py:57-101) ---------------------
// One wave per (crop, joint): argmax with numpy's first-index tie-break, coordinates zeroed when the maximum
// is <= 0, quarter-pixel refinement towards the larger neighbour (POST_PROCESS), then the inverse crop
// affine.  With rot = 0 that affine is x' = cx + (x - W/2) * s, y' = cy + (y - H/2) * s, s = 200*scale_x/W
// (get_affine_transform builds it from the width only).  The reference does this on the host in a Python
// double loop after copying the full heatmaps off the GPU.
__global__ void __launch_bounds__(64) decode_kernel(const float* __restrict__ heat, const float* __restrict__ center,
                                                   const float* __restrict__ scale, float* __restrict__ preds,
                                                   float* __restrict__ maxvals, int J, int H, int W, int post) {
    const int bj = blockIdx.x, b = bj / J, lane = threadIdx.x;
    const float* hm = heat + (size_t)bj * H * W;
    float best = -INFINITY;
    int bi = 0x7fffffff;
    for (int i = lane; i < H * W; i += 64) {
        const float v = hm[i];
        if (v > best) { best = v; bi = i; }          // strictly greater: keeps the first index per lane
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        const float ov = __shfl_xor(best, o);
        const int oi = __shfl_xor(bi, o);
        if (ov > best || (ov == best && oi < bi)) { best = ov; bi = oi; }
    }
    if (lane == 0) {
        float x = (float)(bi % W), y = (float)(bi / W);
        if (!(best > 0.f)) { x = 0.f; y = 0.f; }
        if (post) {
            const int px = (int)floorf(x + 0.5f), py = (int)floorf(y + 0.5f);
            if (1 < px && px < W - 1 && 1 < py && py < H - 1) {
                const float dx = hm[py * W + px + 1] - hm[py * W + px - 1];
                const float dy = hm[(py + 1) * W + px] - hm[(py - 1) * W + px];
                x += (dx > 0.f ? 0.25f : (dx < 0.f ? -0.25f : 0.f));
                y += (dy > 0.f ? 0.25f : (dy < 0.f ? -0.25f : 0.f));
            }
        }
        const float s = scale[2 * b] * 200.0f / (float)W;
        preds[2 * bj] = center[2 * b] + (x - 0.5f * (float)W) * s;
        preds[2 * bj + 1] = center[2 * b + 1] + (y - 0.5f * (float)H) * s;
        maxvals[bj] = best;
    }
}

extern "C" int rsg_decode_heatmaps(const float* heat, const float* center, const float* scale, float* preds,
                                   float* maxvals, int32_t batch, int32_t joints, int32_t height, int32_t width,
                                   int32_t post_process, void* stream) {
    if (!heat || !center || !scale || !preds || !maxvals) RSG_FAIL("decode: null pointer");
    if (batch < 1 || joints < 1 || height < 1 || width < 1) RSG_FAIL("decode: bad shape");
    RSG_LAUNCH(decode_kernel, dim3(batch * joints), dim3(64), 0, (hipStream_t)stream, heat, center, scale, preds,
               maxvals, joints, height, width, post_process);
    RSG_CHECK_LAUNCH("decode");
    return 0;
}


// ---- TRP with RELATION_SUB_SAMPLE (association.py:254-268,286-287) --------------------------------------------
// nn.MaxPool2d(kernel_size=(2, 2)) on a 16-bit NHWC map (a channel slice of a wider buffer: stride in_ld): one thread
// per 8 channels of an output pixel, four 16-byte loads, one 16-byte store.  Max of 16-bit values is exact.
template <class DT>
__global__ void __launch_bounds__(256) maxpool2x2_kernel(const char* __restrict__ in, int in_ld, char* __restrict__ out,
                                                        int out_ld, int ho, int wo, int c8, unsigned total) {
    const unsigned i = blockIdx.x * 256u + threadIdx.x;
    if (i >= total) return;
    const unsigned pix = i / (unsigned)c8, cg = i - pix * (unsigned)c8;
    const unsigned t = pix / (unsigned)wo, x = pix - t * (unsigned)wo;
    const unsigned b = t / (unsigned)ho, y = t - b * (unsigned)ho;
    const int wi = 2 * wo;
    const size_t p00 = ((size_t)b * (2 * ho) + 2 * y) * wi + 2 * x;
    float m[8];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const size_t src = p00 + (size_t)(q >> 1) * wi + (q & 1);
        const uint4 raw = *reinterpret_cast<const uint4*>(in + (src * in_ld + cg * 8) * 2);
        const typename DT::vec8 v = __builtin_bit_cast(typename DT::vec8, raw);
#pragma unroll
        for (int e = 0; e < 8; ++e) m[e] = q == 0 ? (float)v[e] : fmaxf(m[e], (float)v[e]);
    }
    typename DT::vec8 o;
#pragma unroll
    for (int e = 0; e < 8; ++e) o[e] = (typename DT::elem)m[e];
    *reinterpret_cast<uint4*>(out + ((size_t)pix * out_ld + cg * 8) * 2) = __builtin_bit_cast(uint4, o);
}

extern "C" int rsg_maxpool2x2_nhwc16(const void* in, int32_t in_ld, void* out, int32_t out_ld, int32_t batch, int32_t h,
                                     int32_t w, int32_t c, int32_t dtype, void* stream) {
    if (!in || !out) RSG_FAIL("maxpool2x2: null pointer");
    if (batch < 1 || h < 2 || w < 2 || (h & 1) || (w & 1)) RSG_FAIL("maxpool2x2: map %dx%d (even sizes only)", h, w);
    if (c < 8 || c % 8 || in_ld < c || out_ld < c || in_ld % 8 || out_ld % 8 || ((uintptr_t)in & 15) || ((uintptr_t)out & 15))
        RSG_FAIL("maxpool2x2: c=%d in_ld=%d out_ld=%d (multiples of 8, 16-byte aligned)", c, in_ld, out_ld);
    const long total = (long)batch * (h / 2) * (w / 2) * (c / 8);
    if (total >= (1L << 31)) RSG_FAIL("maxpool2x2: map too large");
    const int grid = (int)((total + 255) / 256);
    if (dtype == RSG_BF16)
        RSG_LAUNCH(maxpool2x2_kernel<BF16>, dim3(grid), dim3(256), 0, (hipStream_t)stream, (const char*)in, in_ld, (char*)out, out_ld, h / 2, w / 2, c / 8, (unsigned)total);
    else
        RSG_LAUNCH(maxpool2x2_kernel<F16>, dim3(grid), dim3(256), 0, (hipStream_t)stream, (const char*)in, in_ld, (char*)out, out_ld, h / 2, w / 2, c / 8, (unsigned)total);
    RSG_CHECK_LAUNCH("maxpool2x2");
    return 0;
}

// fp32 rows [rows, c] -> 16-bit rows with stride out_ld (TRP's relation embeddings E on their way into the
// up-sampling deconv, which is an MFMA conv over 16-bit maps like every other conv of the plan)
template <class DT>
__global__ void __launch_bounds__(256) f32_rows_to_16_kernel(const float* __restrict__ in, char* __restrict__ out, int c8,
                                                            int out_ld, unsigned total) {
    const unsigned i = blockIdx.x * 256u + threadIdx.x;
    if (i >= total) return;
    const unsigned row = i / (unsigned)c8, cg = i - row * (unsigned)c8;
    const float4 a = *reinterpret_cast<const float4*>(in + (size_t)i * 8);
    const float4 b = *reinterpret_cast<const float4*>(in + (size_t)i * 8 + 4);
    const f32x8 f = {a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w};
    *reinterpret_cast<uint4*>(out + ((size_t)row * out_ld + cg * 8) * 2) =
        __builtin_bit_cast(uint4, __builtin_convertvector(f, typename DT::vec8));
}

extern "C" int rsg_f32_rows_to_nhwc16(const float* in, void* out, int32_t out_ld, int64_t rows, int32_t c, int32_t dtype,
                                      void* stream) {
    if (!in || !out) RSG_FAIL("f32_rows_to_nhwc16: null pointer");
    if (rows < 1 || c < 8 || c % 8 || out_ld < c || out_ld % 8 || ((uintptr_t)in & 15) || ((uintptr_t)out & 15))
        RSG_FAIL("f32_rows_to_nhwc16: c=%d out_ld=%d (multiples of 8, 16-byte aligned)", c, out_ld);
    const long total = (long)rows * (c / 8);
    if (total >= (1L << 31)) RSG_FAIL("f32_rows_to_nhwc16: too many rows");
    const int grid = (int)((total + 255) / 256);
    if (dtype == RSG_BF16)
        RSG_LAUNCH(f32_rows_to_16_kernel<BF16>, dim3(grid), dim3(256), 0, (hipStream_t)stream, in, (char*)out, c / 8, out_ld, (unsigned)total);
    else
        RSG_LAUNCH(f32_rows_to_16_kernel<F16>, dim3(grid), dim3(256), 0, (hipStream_t)stream, in, (char*)out, c / 8, out_ld, (unsigned)total);
    RSG_CHECK_LAUNCH("f32_rows_to_nhwc16");
    return 0;
}
