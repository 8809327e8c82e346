// Training-mode kernels around the convolutions (gfx950): BatchNorm with batch statistics (forward and backward),
// the ReLU / fuse-sum / nearest-upsample backward passes and gradient plumbing.
//
// Replaces, for `model.train()` (lib/core/function.py:27-80 drives it):
//   nn.BatchNorm2d in training mode            pose_hrnet.py:37,41,70,73,76 ... (momentum 0.1, eps 1e-5)
//   autograd of BatchNorm2d / ReLU / residual add / nn.Upsample(nearest) / the fuse sum   pose_hrnet.py:259-266
// All maps are dense NHWC 16-bit [pixels][C]; statistics, affine parameters and their gradients are fp32.
// Every reduction runs in a fixed order (per-block partials, then one pass over the partials in index order, in
// fp64): the same inputs give the same bits, there are no float atomics.
#include "common.h"

// 16 bytes per lane: a thread owns 8 consecutive channels of a pixel; with C/8 a power of two dividing 256 the
// thread keeps the SAME channels for every pixel it visits (stride 256 threads), so per-channel sums stay private.
#define BN_BLOCKS_MAX 512

template <class DT>
__device__ __forceinline__ f32x8 load8(const char* p) {
    return __builtin_convertvector(__builtin_bit_cast(typename DT::vec8, *reinterpret_cast<const uint4*>(p)), f32x8);
}
template <class DT>
__device__ __forceinline__ void store8(char* p, f32x8 v) {
    *reinterpret_cast<uint4*>(p) = __builtin_bit_cast(uint4, __builtin_convertvector(v, typename DT::vec8));
}

// Per-block partial sums of up to two per-channel quantities.  MODE 0: (sum y, sum y^2).
// MODE 1: (sum g, sum g*yhat) with g = dz * (z > 0) [z may be null: no ReLU], yhat = (y - mean) * invstd.
template <class DT, int MODE>
__global__ void __launch_bounds__(256) bn_partial_kernel(const char* __restrict__ y, const char* __restrict__ dz,
                                                         const char* __restrict__ z, const float* __restrict__ mean,
                                                         const float* __restrict__ invstd, long pixels, int C,
                                                         float* __restrict__ part) {
    __shared__ float red[2][256][8];
    const int CP = C / 8, tid = threadIdx.x;
    const int cc = tid % CP, pl = tid / CP, PPB = 256 / CP;
    const long per = (pixels + gridDim.x - 1) / gridDim.x;
    const long p0 = (long)blockIdx.x * per, p1 = min(pixels, p0 + per);
    f32x8 s0 = {0, 0, 0, 0, 0, 0, 0, 0}, s1 = s0, mu = s0, is = s0;
    if (MODE == 1) {
#pragma unroll
        for (int e = 0; e < 8; ++e) { mu[e] = mean[cc * 8 + e]; is[e] = invstd[cc * 8 + e]; }
    }
    for (long p = p0 + pl; p < p1; p += PPB) {
        const long off = (p * C + cc * 8) * 2;
        const f32x8 v = load8<DT>(y + off);
        if (MODE == 0) {
#pragma unroll
            for (int e = 0; e < 8; ++e) { s0[e] += v[e]; s1[e] += v[e] * v[e]; }
        } else {
            f32x8 g = load8<DT>(dz + off);
            if (z) {
                const f32x8 zz = load8<DT>(z + off);
#pragma unroll
                for (int e = 0; e < 8; ++e) g[e] = zz[e] > 0.f ? g[e] : 0.f;
            }
#pragma unroll
            for (int e = 0; e < 8; ++e) { s0[e] += g[e]; s1[e] += g[e] * ((v[e] - mu[e]) * is[e]); }
        }
    }
#pragma unroll
    for (int e = 0; e < 8; ++e) { red[0][tid][e] = s0[e]; red[1][tid][e] = s1[e]; }
    __syncthreads();
    if (tid < C) {                          // channel tid: sum over the pixel lanes in index order
        const int c8 = tid / 8, e = tid % 8;
        float a = 0.f, b = 0.f;
        for (int k = 0; k < PPB; ++k) { a += red[0][k * CP + c8][e]; b += red[1][k * CP + c8][e]; }
        part[((long)blockIdx.x * 2 + 0) * C + tid] = a;
        part[((long)blockIdx.x * 2 + 1) * C + tid] = b;
    }
}

// MODE 0: mean / invstd of the batch + running statistics (momentum update; the running variance takes the unbiased
// estimate, the normalisation the biased one: SURVEY App. D).  MODE 1: dgamma = sum g*yhat, dbeta = sum g, and the two
// sums (kept in out0/out1 as fp32) for the apply pass.
template <int MODE>
__global__ void __launch_bounds__(64) bn_final_kernel(const float* __restrict__ part, int nblocks, int C, long pixels,
                                                      float* __restrict__ out0, float* __restrict__ out1,
                                                      float* __restrict__ run_mean, float* __restrict__ run_var,
                                                      float momentum, float eps) {
    // one wave per channel: lane t sums the partials t, t+64, ... (fp64), then a fixed butterfly over the lanes --
    // the first version walked the partials with one thread per channel, a chain of up to 512 dependent L2 reads
    // (~100 us per BatchNorm, more than the passes over the map)
    __shared__ double ra[64], rb[64];
    const int c = blockIdx.x, t = threadIdx.x;
    double a = 0.0, b = 0.0;
    for (int k = t; k < nblocks; k += 64) { a += (double)part[((long)k * 2 + 0) * C + c]; b += (double)part[((long)k * 2 + 1) * C + c]; }
    ra[t] = a; rb[t] = b;
    __syncthreads();
    for (int w = 32; w > 0; w >>= 1) {
        if (t < w) { ra[t] += ra[t + w]; rb[t] += rb[t + w]; }
        __syncthreads();
    }
    if (t != 0) return;
    a = ra[0]; b = rb[0];
    if (MODE == 0) {
        const double m = a / (double)pixels;
        double var = b / (double)pixels - m * m;
        if (var < 0.0) var = 0.0;
        out0[c] = (float)m;
        out1[c] = (float)(1.0 / sqrt(var + (double)eps));
        if (run_mean) run_mean[c] = (float)((1.0 - momentum) * (double)run_mean[c] + momentum * m);
        if (run_var) {
            const double unb = pixels > 1 ? var * (double)pixels / (double)(pixels - 1) : var;
            run_var[c] = (float)((1.0 - momentum) * (double)run_var[c] + momentum * unb);
        }
    } else {
        out0[c] = (float)a;     // sum g       = dbeta
        out1[c] = (float)b;     // sum g*yhat  = dgamma
    }
}

// z = [ReLU]( gamma * (y - mean) * invstd + beta [+ residual] )
template <class DT>
__global__ void __launch_bounds__(256) bn_apply_kernel(const char* __restrict__ y, const float* __restrict__ mean,
                                                       const float* __restrict__ invstd, const float* __restrict__ gamma,
                                                       const float* __restrict__ beta, const char* __restrict__ res,
                                                       int relu, char* __restrict__ out, long pixels, int C) {
    const int CP = C / 8, tid = threadIdx.x, cc = tid % CP, pl = tid / CP, PPB = 256 / CP;
    f32x8 a, b;
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        const int c = cc * 8 + e;
        a[e] = gamma[c] * invstd[c];
        b[e] = beta[c] - mean[c] * a[e];
    }
    for (long p = (long)blockIdx.x * PPB + pl; p < pixels; p += (long)gridDim.x * PPB) {
        const long off = (p * C + cc * 8) * 2;
        f32x8 v = load8<DT>(y + off);
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] = v[e] * a[e] + b[e];
        if (res) {
            const f32x8 r = load8<DT>(res + off);
#pragma unroll
            for (int e = 0; e < 8; ++e) v[e] += r[e];
        }
        if (relu) {
#pragma unroll
            for (int e = 0; e < 8; ++e) v[e] = fmaxf(v[e], 0.f);
        }
        store8<DT>(out + off, v);
    }
}

// dy = gamma * invstd * (g - sum_g / M - yhat * sum_gy / M);  dres = g (the masked upstream gradient) when asked for
template <class DT>
__global__ void __launch_bounds__(256) bn_backward_apply_kernel(const char* __restrict__ y, const char* __restrict__ dz,
                                                                const char* __restrict__ z, const float* __restrict__ mean,
                                                                const float* __restrict__ invstd, const float* __restrict__ gamma,
                                                                const float* __restrict__ sum_g, const float* __restrict__ sum_gy,
                                                                char* __restrict__ dy, char* __restrict__ dres, long pixels, int C) {
    const int CP = C / 8, tid = threadIdx.x, cc = tid % CP, pl = tid / CP, PPB = 256 / CP;
    const float invM = 1.f / (float)pixels;
    f32x8 mu, is, a, m1, m2;
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        const int c = cc * 8 + e;
        mu[e] = mean[c]; is[e] = invstd[c]; a[e] = gamma[c] * invstd[c];
        m1[e] = sum_g[c] * invM; m2[e] = sum_gy[c] * invM;
    }
    for (long p = (long)blockIdx.x * PPB + pl; p < pixels; p += (long)gridDim.x * PPB) {
        const long off = (p * C + cc * 8) * 2;
        const f32x8 v = load8<DT>(y + off);
        f32x8 g = load8<DT>(dz + off);
        if (z) {
            const f32x8 zz = load8<DT>(z + off);
#pragma unroll
            for (int e = 0; e < 8; ++e) g[e] = zz[e] > 0.f ? g[e] : 0.f;
        }
        if (dres) store8<DT>(dres + off, g);
        f32x8 o;
#pragma unroll
        for (int e = 0; e < 8; ++e) o[e] = a[e] * (g[e] - m1[e] - (v[e] - mu[e]) * is[e] * m2[e]);
        store8<DT>(dy + off, o);
    }
}

static int bn_check(int C, long pixels) { return C >= 8 && C <= 256 && (C & (C - 1)) == 0 && pixels >= 1; }
static int bn_blocks(long pixels, int C) {
    const long ppb = 256 / (C / 8);
    long b = (pixels + ppb * 8 - 1) / (ppb * 8);       // >= 8 pixels per pixel lane
    return (int)(b < 1 ? 1 : (b > BN_BLOCKS_MAX ? BN_BLOCKS_MAX : b));
}

extern "C" size_t rsg_bn_workspace_bytes(int32_t c) { return (size_t)BN_BLOCKS_MAX * 2 * c * sizeof(float); }

extern "C" int rsg_bn_stats(const void* y, int64_t pixels, int32_t c, int32_t dtype, float* workspace, float* mean,
                            float* invstd, float* running_mean, float* running_var, float momentum, float eps, void* stream) {
    if (!y || !workspace || !mean || !invstd) RSG_FAIL("bn_stats: null pointer");
    if (!bn_check(c, pixels)) RSG_FAIL("bn_stats: channels %d must be a power of two in [8, 256]", c);
    const int nb = bn_blocks(pixels, c);
    hipStream_t s = (hipStream_t)stream;
    if (dtype == RSG_BF16)
        RSG_LAUNCH((bn_partial_kernel<BF16, 0>), dim3(nb), dim3(256), 0, s, (const char*)y, nullptr, nullptr, nullptr, nullptr, (long)pixels, c, workspace);
    else
        RSG_LAUNCH((bn_partial_kernel<F16, 0>), dim3(nb), dim3(256), 0, s, (const char*)y, nullptr, nullptr, nullptr, nullptr, (long)pixels, c, workspace);
    RSG_LAUNCH(bn_final_kernel<0>, dim3(c), dim3(64), 0, s, (const float*)workspace, nb, c, (long)pixels, mean, invstd,
               running_mean, running_var, momentum, eps);
    RSG_CHECK_LAUNCH("bn_stats");
    return 0;
}

extern "C" int rsg_bn_apply(const void* y, const float* mean, const float* invstd, const float* gamma, const float* beta,
                            const void* residual, int32_t relu, void* out, int64_t pixels, int32_t c, int32_t dtype, void* stream) {
    if (!y || !mean || !invstd || !gamma || !beta || !out) RSG_FAIL("bn_apply: null pointer");
    if (!bn_check(c, pixels)) RSG_FAIL("bn_apply: channels %d must be a power of two in [8, 256]", c);
    const long ppb = 256 / (c / 8);
    const int nb = (int)min((pixels + ppb - 1) / ppb, (long)2048);
    if (dtype == RSG_BF16)
        RSG_LAUNCH(bn_apply_kernel<BF16>, dim3(nb), dim3(256), 0, (hipStream_t)stream, (const char*)y, mean, invstd, gamma, beta,
                   (const char*)residual, relu, (char*)out, (long)pixels, c);
    else
        RSG_LAUNCH(bn_apply_kernel<F16>, dim3(nb), dim3(256), 0, (hipStream_t)stream, (const char*)y, mean, invstd, gamma, beta,
                   (const char*)residual, relu, (char*)out, (long)pixels, c);
    RSG_CHECK_LAUNCH("bn_apply");
    return 0;
}

extern "C" int rsg_bn_backward(const void* dz, const void* z, const void* y, const float* mean, const float* invstd,
                               const float* gamma, void* dy, void* dres, float* dgamma, float* dbeta, float* workspace,
                               int64_t pixels, int32_t c, int32_t dtype, void* stream) {
    if (!dz || !y || !mean || !invstd || !gamma || !dy || !dgamma || !dbeta || !workspace) RSG_FAIL("bn_backward: null pointer");
    if (!bn_check(c, pixels)) RSG_FAIL("bn_backward: channels %d must be a power of two in [8, 256]", c);
    const int nb = bn_blocks(pixels, c);
    hipStream_t s = (hipStream_t)stream;
    const long ppb = 256 / (c / 8);
    const int na = (int)min((pixels + ppb - 1) / ppb, (long)2048);
    if (dtype == RSG_BF16) {
        RSG_LAUNCH((bn_partial_kernel<BF16, 1>), dim3(nb), dim3(256), 0, s, (const char*)y, (const char*)dz, (const char*)z, mean, invstd, (long)pixels, c, workspace);
        RSG_LAUNCH(bn_final_kernel<1>, dim3(c), dim3(64), 0, s, (const float*)workspace, nb, c, (long)pixels, dbeta, dgamma,
                   nullptr, nullptr, 0.f, 0.f);
        RSG_LAUNCH(bn_backward_apply_kernel<BF16>, dim3(na), dim3(256), 0, s, (const char*)y, (const char*)dz, (const char*)z, mean, invstd,
                   gamma, (const float*)dbeta, (const float*)dgamma, (char*)dy, (char*)dres, (long)pixels, c);
    } else {
        RSG_LAUNCH((bn_partial_kernel<F16, 1>), dim3(nb), dim3(256), 0, s, (const char*)y, (const char*)dz, (const char*)z, mean, invstd, (long)pixels, c, workspace);
        RSG_LAUNCH(bn_final_kernel<1>, dim3(c), dim3(64), 0, s, (const float*)workspace, nb, c, (long)pixels, dbeta, dgamma,
                   nullptr, nullptr, 0.f, 0.f);
        RSG_LAUNCH(bn_backward_apply_kernel<F16>, dim3(na), dim3(256), 0, s, (const char*)y, (const char*)dz, (const char*)z, mean, invstd,
                   gamma, (const float*)dbeta, (const float*)dgamma, (char*)dy, (char*)dres, (long)pixels, c);
    }
    RSG_CHECK_LAUNCH("bn_backward");
    return 0;
}

// ---- elementwise gradient plumbing ------------------------------------------------------------------------------
// MODE 0: g = dout * (out > 0)   (backward of the ReLU that ends a fuse sum, pose_hrnet.py:266)
// MODE 1: out = a + b            (a map consumed twice receives two gradients)
template <class DT, int MODE>
__global__ void __launch_bounds__(256) ew_kernel(const char* __restrict__ a, const char* __restrict__ b, char* __restrict__ o, long n8) {
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n8; i += (long)gridDim.x * 256) {
        f32x8 x = load8<DT>(a + i * 16);
        const f32x8 y = load8<DT>(b + i * 16);
#pragma unroll
        for (int e = 0; e < 8; ++e) x[e] = MODE == 0 ? (y[e] > 0.f ? x[e] : 0.f) : x[e] + y[e];
        store8<DT>(o + i * 16, x);
    }
}

static int ew_launch(int mode, const void* a, const void* b, void* o, int64_t n, int dtype, void* stream, const char* what) {
    if (!a || !b || !o || n < 8 || (n & 7)) RSG_FAIL("%s: bad arguments (n must be a positive multiple of 8)", what);
    const long n8 = n / 8;
    const int nb = (int)min((n8 + 255) / 256, (long)4096);
    hipStream_t s = (hipStream_t)stream;
    if (mode == 0) {
        if (dtype == RSG_BF16) RSG_LAUNCH((ew_kernel<BF16, 0>), dim3(nb), dim3(256), 0, s, (const char*)a, (const char*)b, (char*)o, n8);
        else RSG_LAUNCH((ew_kernel<F16, 0>), dim3(nb), dim3(256), 0, s, (const char*)a, (const char*)b, (char*)o, n8);
    } else {
        if (dtype == RSG_BF16) RSG_LAUNCH((ew_kernel<BF16, 1>), dim3(nb), dim3(256), 0, s, (const char*)a, (const char*)b, (char*)o, n8);
        else RSG_LAUNCH((ew_kernel<F16, 1>), dim3(nb), dim3(256), 0, s, (const char*)a, (const char*)b, (char*)o, n8);
    }
    RSG_CHECK_LAUNCH(what);
    return 0;
}

extern "C" int rsg_relu_backward(const void* dout, const void* out, void* g, int64_t n, int32_t dtype, void* stream) {
    return ew_launch(0, dout, out, g, n, dtype, stream, "relu_backward");
}
extern "C" int rsg_add16(const void* a, const void* b, void* out, int64_t n, int32_t dtype, void* stream) {
    return ew_launch(1, a, b, out, n, dtype, stream, "add16");
}

// Backward of nn.Upsample(scale_factor = 2^shift, mode='nearest') (pose_hrnet.py:213): every low-resolution pixel
// receives the sum of the 2^shift x 2^shift gradient pixels it was copied to.  out: [batch, h, w, c]; in: [batch, h<<s, w<<s, c].
template <class DT>
__global__ void __launch_bounds__(256) sumpool_kernel(const char* __restrict__ in, char* __restrict__ out, int batch, int h, int w,
                                                      int C, int shift) {
    const int CP = C / 8;
    const long total = (long)batch * h * w * CP;
    const int f = 1 << shift, W2 = w << shift;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
        const int cc = (int)(i % CP);
        const long p = i / CP;
        const int x = (int)(p % w), yy = (int)((p / w) % h);
        const long b = p / ((long)w * h);
        f32x8 s = {0, 0, 0, 0, 0, 0, 0, 0};
        for (int dy = 0; dy < f; ++dy)
            for (int dx = 0; dx < f; ++dx) {
                const long q = (b * (h << shift) + ((yy << shift) + dy)) * W2 + ((x << shift) + dx);
                const f32x8 v = load8<DT>(in + (q * C + cc * 8) * 2);
#pragma unroll
                for (int e = 0; e < 8; ++e) s[e] += v[e];
            }
        store8<DT>(out + (p * C + cc * 8) * 2, s);
    }
}

extern "C" int rsg_upsample_nearest_backward(const void* grad_hi, void* grad_lo, int32_t batch, int32_t h_lo, int32_t w_lo,
                                             int32_t c, int32_t shift, int32_t dtype, void* stream) {
    if (!grad_hi || !grad_lo || c % 8 || shift < 1 || shift > 3 || batch < 1) RSG_FAIL("upsample_nearest_backward: bad arguments");
    const long total = (long)batch * h_lo * w_lo * (c / 8);
    const int nb = (int)min((total + 255) / 256, (long)4096);
    if (dtype == RSG_BF16)
        RSG_LAUNCH(sumpool_kernel<BF16>, dim3(nb), dim3(256), 0, (hipStream_t)stream, (const char*)grad_hi, (char*)grad_lo, batch, h_lo, w_lo, c, shift);
    else
        RSG_LAUNCH(sumpool_kernel<F16>, dim3(nb), dim3(256), 0, (hipStream_t)stream, (const char*)grad_hi, (char*)grad_lo, batch, h_lo, w_lo, c, shift);
    RSG_CHECK_LAUNCH("upsample_nearest_backward");
    return 0;
}

// Data gradient of a stride-2 conv, first half: the output gradient spread onto the input grid with zeros between
// (out[b, 2y, 2x] = in[b, y, x]); a stride-1 conv with the flipped, transposed weights over it is conv_transpose.
__global__ void __launch_bounds__(256) zero_insert_kernel(const uint4* __restrict__ in, uint4* __restrict__ out, int batch, int h, int w, int CP) {
    const long total = (long)batch * (2 * h) * (2 * w) * CP;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
        const int cc = (int)(i % CP);
        const long p = i / CP;
        const int X = (int)(p % (2 * w)), Y = (int)((p / (2 * w)) % (2 * h));
        const long b = p / ((long)4 * w * h);
        uint4 v = make_uint4(0, 0, 0, 0);
        if (!(X & 1) && !(Y & 1)) v = in[((b * h + (Y >> 1)) * w + (X >> 1)) * CP + cc];
        out[i] = v;
    }
}

extern "C" int rsg_zero_insert2x(const void* in, void* out, int32_t batch, int32_t h, int32_t w, int32_t c, void* stream) {
    if (!in || !out || c % 8 || batch < 1 || h < 1 || w < 1) RSG_FAIL("zero_insert2x: bad arguments");
    const long total = (long)batch * 4 * h * w * (c / 8);
    const int nb = (int)min((total + 255) / 256, (long)4096);
    RSG_LAUNCH(zero_insert_kernel, dim3(nb), dim3(256), 0, (hipStream_t)stream, (const uint4*)in, (uint4*)out, batch, h, w, c / 8);
    RSG_CHECK_LAUNCH("zero_insert2x");
    return 0;
}

// Bias gradient of the classifier conv (pose_hrnet.py:326-332): out[c] = sum over batch and pixels of the fp32 NCHW
// heat-map gradient.  Workgroup (c, s) sums the planes b = s, s + S, ... of channel c (fixed order), the last-level sum
// over the S segments runs in the second kernel, in segment order, in fp64.
#define CSUM_SEGS 32
__global__ void __launch_bounds__(256) channel_sum_kernel(const float* __restrict__ x, double* __restrict__ part, int batch, int C, int hw) {
    __shared__ double red[256];
    const int c = blockIdx.x, seg = blockIdx.y;
    double s = 0.0;
    for (int b = seg; b < batch; b += CSUM_SEGS) {
        const float* p = x + ((long)b * C + c) * hw;
        float t = 0.f;
        for (int i = threadIdx.x; i < hw; i += 256) t += p[i];
        s += (double)t;
    }
    red[threadIdx.x] = s;
    __syncthreads();
    for (int k = 128; k > 0; k >>= 1) {
        if (threadIdx.x < k) red[threadIdx.x] += red[threadIdx.x + k];
        __syncthreads();
    }
    if (threadIdx.x == 0) part[c * CSUM_SEGS + seg] = red[0];
}
__global__ void __launch_bounds__(64) channel_sum_final_kernel(const double* __restrict__ part, float* __restrict__ out, int C) {
    const int c = blockIdx.x * 64 + threadIdx.x;
    if (c >= C) return;
    double s = 0.0;
    for (int k = 0; k < CSUM_SEGS; ++k) s += part[c * CSUM_SEGS + k];
    out[c] = (float)s;
}

extern "C" size_t rsg_channel_sum_workspace_bytes(int32_t c) { return (size_t)c * CSUM_SEGS * sizeof(double); }

extern "C" int rsg_channel_sum_nchw_f32(const float* x, float* out, void* workspace, int32_t batch, int32_t c, int32_t hw, void* stream) {
    if (!x || !out || !workspace || batch < 1 || c < 1 || hw < 1) RSG_FAIL("channel_sum: bad arguments");
    RSG_LAUNCH(channel_sum_kernel, dim3(c, CSUM_SEGS), dim3(256), 0, (hipStream_t)stream, x, (double*)workspace, batch, c, hw);
    RSG_LAUNCH(channel_sum_final_kernel, dim3(ceil_div(c, 64)), dim3(64), 0, (hipStream_t)stream, (const double*)workspace, out, c);
    RSG_CHECK_LAUNCH("channel_sum");
    return 0;
}

// ---- grouped BatchNorm launches ------------------------------------------------------------------------------------
// The four branches of a HighResolutionModule are independent, and on most of their maps a BatchNorm pass is a
// launch-latency-sized kernel (3.2 us per kernel in a replayed graph against ~1 us of work): up to 8 independent
// BatchNorms run as ONE partial, ONE final and ONE apply launch.  A workgroup finds its item by a scan of the block
// offsets (static indices: the item arrives with one batch of scalar loads) and then runs the single-map code.
#define RSG_BN_GROUP_MAX 8
struct BnGroupItem {
    const char* y; const char* dz; const char* z; const char* res;
    char* out; char* dres;
    const float* gamma; const float* beta;
    float* mean; float* invstd; float* run_mean; float* run_var; float* dgamma; float* dbeta; float* part;
    long pixels;
    int C, relu, nb, na;
};
struct BnGroupArgs {
    int n;
    float momentum, eps;
    int first_p[RSG_BN_GROUP_MAX + 1], first_f[RSG_BN_GROUP_MAX + 1], first_a[RSG_BN_GROUP_MAX + 1];
    BnGroupItem it[RSG_BN_GROUP_MAX];
};

__device__ __forceinline__ int bn_group_pick(const int (&first)[RSG_BN_GROUP_MAX + 1], int n, BnGroupItem& it, const BnGroupArgs& g, int& local) {
    int idx = 0;
#pragma unroll
    for (int k = 1; k < RSG_BN_GROUP_MAX; ++k) idx += (k < n && (int)blockIdx.x >= first[k]) ? 1 : 0;
#pragma unroll
    for (int k = 0; k < RSG_BN_GROUP_MAX; ++k)
        if (k == idx) { it = g.it[k]; local = (int)blockIdx.x - first[k]; }
    return idx;
}

template <class DT, int MODE>
__global__ void __launch_bounds__(256) bn_partial_group_kernel(const BnGroupArgs g) {
    __shared__ float red[2][256][8];
    BnGroupItem it; int blk = 0;
    bn_group_pick(g.first_p, g.n, it, g, blk);
    const int C = it.C, CP = C / 8, tid = threadIdx.x;
    const int cc = tid % CP, pl = tid / CP, PPB = 256 / CP;
    const long per = (it.pixels + it.nb - 1) / it.nb;
    const long p0 = (long)blk * per, p1 = min(it.pixels, p0 + per);
    f32x8 s0 = {0, 0, 0, 0, 0, 0, 0, 0}, s1 = s0, mu = s0, is = s0;
    if (MODE == 1) {
#pragma unroll
        for (int e = 0; e < 8; ++e) { mu[e] = it.mean[cc * 8 + e]; is[e] = it.invstd[cc * 8 + e]; }
    }
    for (long p = p0 + pl; p < p1; p += PPB) {
        const long off = (p * C + cc * 8) * 2;
        const f32x8 v = load8<DT>(it.y + off);
        if (MODE == 0) {
#pragma unroll
            for (int e = 0; e < 8; ++e) { s0[e] += v[e]; s1[e] += v[e] * v[e]; }
        } else {
            f32x8 gq = load8<DT>(it.dz + off);
            if (it.z) {
                const f32x8 zz = load8<DT>(it.z + off);
#pragma unroll
                for (int e = 0; e < 8; ++e) gq[e] = zz[e] > 0.f ? gq[e] : 0.f;
            }
#pragma unroll
            for (int e = 0; e < 8; ++e) { s0[e] += gq[e]; s1[e] += gq[e] * ((v[e] - mu[e]) * is[e]); }
        }
    }
#pragma unroll
    for (int e = 0; e < 8; ++e) { red[0][tid][e] = s0[e]; red[1][tid][e] = s1[e]; }
    __syncthreads();
    if (tid < C) {
        const int c8 = tid / 8, e = tid % 8;
        float a = 0.f, b = 0.f;
        for (int k = 0; k < PPB; ++k) { a += red[0][k * CP + c8][e]; b += red[1][k * CP + c8][e]; }
        it.part[((long)blk * 2 + 0) * C + tid] = a;
        it.part[((long)blk * 2 + 1) * C + tid] = b;
    }
}

template <int MODE>
__global__ void __launch_bounds__(64) bn_final_group_kernel(const BnGroupArgs g) {
    __shared__ double ra[64], rb[64];
    BnGroupItem it; int c = 0;
    bn_group_pick(g.first_f, g.n, it, g, c);
    const int t = threadIdx.x, C = it.C;
    double a = 0.0, b = 0.0;
    for (int k = t; k < it.nb; k += 64) { a += (double)it.part[((long)k * 2 + 0) * C + c]; b += (double)it.part[((long)k * 2 + 1) * C + c]; }
    ra[t] = a; rb[t] = b;
    __syncthreads();
    for (int w = 32; w > 0; w >>= 1) {
        if (t < w) { ra[t] += ra[t + w]; rb[t] += rb[t + w]; }
        __syncthreads();
    }
    if (t != 0) return;
    a = ra[0]; b = rb[0];
    if (MODE == 0) {
        const double m = a / (double)it.pixels;
        double var = b / (double)it.pixels - m * m;
        if (var < 0.0) var = 0.0;
        it.mean[c] = (float)m;
        it.invstd[c] = (float)(1.0 / sqrt(var + (double)g.eps));
        if (it.run_mean) it.run_mean[c] = (float)((1.0 - g.momentum) * (double)it.run_mean[c] + g.momentum * m);
        if (it.run_var) {
            const double unb = it.pixels > 1 ? var * (double)it.pixels / (double)(it.pixels - 1) : var;
            it.run_var[c] = (float)((1.0 - g.momentum) * (double)it.run_var[c] + g.momentum * unb);
        }
    } else {
        it.dbeta[c] = (float)a;
        it.dgamma[c] = (float)b;
    }
}

// MODE 0: forward apply; MODE 1: backward apply
template <class DT, int MODE>
__global__ void __launch_bounds__(256) bn_apply_group_kernel(const BnGroupArgs g) {
    BnGroupItem it; int blk = 0;
    bn_group_pick(g.first_a, g.n, it, g, blk);
    const int C = it.C, CP = C / 8, tid = threadIdx.x, cc = tid % CP, pl = tid / CP, PPB = 256 / CP;
    if (MODE == 0) {
        f32x8 a, b;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const int c = cc * 8 + e;
            a[e] = it.gamma[c] * it.invstd[c];
            b[e] = it.beta[c] - it.mean[c] * a[e];
        }
        for (long p = (long)blk * PPB + pl; p < it.pixels; p += (long)it.na * PPB) {
            const long off = (p * C + cc * 8) * 2;
            f32x8 v = load8<DT>(it.y + off);
#pragma unroll
            for (int e = 0; e < 8; ++e) v[e] = v[e] * a[e] + b[e];
            if (it.res) {
                const f32x8 r = load8<DT>(it.res + off);
#pragma unroll
                for (int e = 0; e < 8; ++e) v[e] += r[e];
            }
            if (it.relu) {
#pragma unroll
                for (int e = 0; e < 8; ++e) v[e] = fmaxf(v[e], 0.f);
            }
            store8<DT>(it.out + off, v);
        }
    } else {
        const float invM = 1.f / (float)it.pixels;
        f32x8 mu, is, a, m1, m2;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const int c = cc * 8 + e;
            mu[e] = it.mean[c]; is[e] = it.invstd[c]; a[e] = it.gamma[c] * it.invstd[c];
            m1[e] = it.dbeta[c] * invM; m2[e] = it.dgamma[c] * invM;
        }
        for (long p = (long)blk * PPB + pl; p < it.pixels; p += (long)it.na * PPB) {
            const long off = (p * C + cc * 8) * 2;
            const f32x8 v = load8<DT>(it.y + off);
            f32x8 gq = load8<DT>(it.dz + off);
            if (it.z) {
                const f32x8 zz = load8<DT>(it.z + off);
#pragma unroll
                for (int e = 0; e < 8; ++e) gq[e] = zz[e] > 0.f ? gq[e] : 0.f;
            }
            if (it.dres) store8<DT>(it.dres + off, gq);
            f32x8 o;
#pragma unroll
            for (int e = 0; e < 8; ++e) o[e] = a[e] * (gq[e] - m1[e] - (v[e] - mu[e]) * is[e] * m2[e]);
            store8<DT>(it.out + off, o);
        }
    }
}

static int bn_group_fill(const rsg_bn_item* items, int n, int backward, BnGroupArgs& g, const char* what) {
    if (!items || n < 1 || n > RSG_BN_GROUP_MAX) RSG_FAIL("%s: 1..%d items", what, RSG_BN_GROUP_MAX);
    memset(&g, 0, sizeof(g));
    g.n = n;
    int fp = 0, ff = 0, fa = 0;
    for (int k = 0; k < n; ++k) {
        const rsg_bn_item& s = items[k];
        if (!s.y || !s.mean || !s.invstd || !s.gamma || !s.workspace || !s.out) RSG_FAIL("%s: item %d: null pointer", what, k);
        if (backward ? (!s.dz || !s.dgamma || !s.dbeta) : !s.beta) RSG_FAIL("%s: item %d: null pointer", what, k);
        if (!bn_check(s.c, s.pixels)) RSG_FAIL("%s: item %d: channels %d must be a power of two in [8, 256]", what, k, s.c);
        BnGroupItem& d = g.it[k];
        d.y = (const char*)s.y; d.dz = (const char*)s.dz; d.z = (const char*)s.z; d.res = (const char*)s.residual;
        d.out = (char*)s.out; d.dres = (char*)s.dres; d.gamma = s.gamma; d.beta = s.beta; d.mean = s.mean; d.invstd = s.invstd;
        d.run_mean = s.running_mean; d.run_var = s.running_var; d.dgamma = s.dgamma; d.dbeta = s.dbeta; d.part = s.workspace;
        d.pixels = s.pixels; d.C = s.c; d.relu = s.relu;
        d.nb = bn_blocks(s.pixels, s.c);
        const long ppb = 256 / (s.c / 8);
        d.na = (int)min((s.pixels + ppb - 1) / ppb, (long)1024);
        g.first_p[k] = fp; g.first_f[k] = ff; g.first_a[k] = fa;
        fp += d.nb; ff += s.c; fa += d.na;
    }
    g.first_p[n] = fp; g.first_f[n] = ff; g.first_a[n] = fa;
    return 0;
}

extern "C" int rsg_bn_forward_group(const rsg_bn_item* items, int32_t n, float momentum, float eps, int32_t dtype, void* stream) {
    BnGroupArgs g;
    if (bn_group_fill(items, n, 0, g, "bn_forward_group")) return 1;
    g.momentum = momentum; g.eps = eps;
    hipStream_t s = (hipStream_t)stream;
    if (dtype == RSG_BF16) RSG_LAUNCH((bn_partial_group_kernel<BF16, 0>), dim3(g.first_p[n]), dim3(256), 0, s, g);
    else RSG_LAUNCH((bn_partial_group_kernel<F16, 0>), dim3(g.first_p[n]), dim3(256), 0, s, g);
    RSG_LAUNCH(bn_final_group_kernel<0>, dim3(g.first_f[n]), dim3(64), 0, s, g);
    if (dtype == RSG_BF16) RSG_LAUNCH((bn_apply_group_kernel<BF16, 0>), dim3(g.first_a[n]), dim3(256), 0, s, g);
    else RSG_LAUNCH((bn_apply_group_kernel<F16, 0>), dim3(g.first_a[n]), dim3(256), 0, s, g);
    RSG_CHECK_LAUNCH("bn_forward_group");
    return 0;
}

extern "C" int rsg_bn_backward_group(const rsg_bn_item* items, int32_t n, int32_t dtype, void* stream) {
    BnGroupArgs g;
    if (bn_group_fill(items, n, 1, g, "bn_backward_group")) return 1;
    hipStream_t s = (hipStream_t)stream;
    if (dtype == RSG_BF16) RSG_LAUNCH((bn_partial_group_kernel<BF16, 1>), dim3(g.first_p[n]), dim3(256), 0, s, g);
    else RSG_LAUNCH((bn_partial_group_kernel<F16, 1>), dim3(g.first_p[n]), dim3(256), 0, s, g);
    RSG_LAUNCH(bn_final_group_kernel<1>, dim3(g.first_f[n]), dim3(64), 0, s, g);
    if (dtype == RSG_BF16) RSG_LAUNCH((bn_apply_group_kernel<BF16, 1>), dim3(g.first_a[n]), dim3(256), 0, s, g);
    else RSG_LAUNCH((bn_apply_group_kernel<F16, 1>), dim3(g.first_a[n]), dim3(256), 0, s, g);
    RSG_CHECK_LAUNCH("bn_backward_group");
    return 0;
}
