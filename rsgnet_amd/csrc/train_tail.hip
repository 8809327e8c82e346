// Training tail of RSGNet for gfx950 (MI355X): fused loss value + gradient kernels, the relation-target vector and
// a flat Adam step.  HBM-bound elementwise/reduction work: one pass over the operands, 16-byte accesses, fp64
// partial sums reduced in a fixed order (bit-stable from run to run).
//
// Replaces (reference file:line):
//   rsg_joints_mse_loss      JointsMSELoss.forward + its autograd backward          lib/core/loss.py:14-38
//   rsg_bce_loss             torch.nn.BCELoss() (mean) forward + backward            lib/core/function.py:240,293
//   rsg_relation_target_vec  max over joints, bilinear x0.5 (align_corners=True)     lib/core/function.py:261-268
//   rsg_adam_step            torch.optim.Adam(lr) update, no weight decay            lib/utils/utils.py:70-74
#include "common.h"

// ---- 0.5 * mean_j mean_{b,p} ((pred - gt) * w)^2 and its gradient -------------------------------------------
// one workgroup per (b, j) row of P values: partial[b*J+j] = sum d^2 (fp64); grad = d * w * gscale
__global__ void __launch_bounds__(256) joints_mse_kernel(const float* __restrict__ pred, const float* __restrict__ gt,
                                                         const float* __restrict__ w, int P, float gscale,
                                                         double* __restrict__ partial, float* __restrict__ grad) {
    const int row = blockIdx.x;
    const float wv = w ? w[row] : 1.f;
    const float4* p4 = reinterpret_cast<const float4*>(pred + (size_t)row * P);
    const float4* g4 = reinterpret_cast<const float4*>(gt + (size_t)row * P);
    float4* o4 = grad ? reinterpret_cast<float4*>(grad + (size_t)row * P) : nullptr;
    double acc = 0.0;
    for (int i = threadIdx.x; i < P / 4; i += 256) {
        const float4 a = p4[i], b = g4[i];
        const float d0 = (a.x - b.x) * wv, d1 = (a.y - b.y) * wv, d2 = (a.z - b.z) * wv, d3 = (a.w - b.w) * wv;
        acc += (double)d0 * d0 + (double)d1 * d1 + (double)d2 * d2 + (double)d3 * d3;
        if (o4) o4[i] = make_float4(d0 * wv * gscale, d1 * wv * gscale, d2 * wv * gscale, d3 * wv * gscale);
    }
    for (int i = (P / 4) * 4 + threadIdx.x; i < P; i += 256) {      // tail (P not a multiple of 4)
        const float d = (pred[(size_t)row * P + i] - gt[(size_t)row * P + i]) * wv;
        acc += (double)d * d;
        if (grad) grad[(size_t)row * P + i] = d * wv * gscale;
    }
    __shared__ double red[256];
    red[threadIdx.x] = acc;
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) {
        if (threadIdx.x < s) red[threadIdx.x] += red[threadIdx.x + s];
        __syncthreads();
    }
    if (threadIdx.x == 0) partial[row] = red[0];
}

// out[0] (+)= scale * sum(partial[0..n)) in a fixed order
__global__ void __launch_bounds__(256) final_sum_kernel(const double* __restrict__ partial, int n, double scale,
                                                        float* __restrict__ out, int accumulate) {
    __shared__ double red[256];
    double acc = 0.0;
    for (int i = threadIdx.x; i < n; i += 256) acc += partial[i];
    red[threadIdx.x] = acc;
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) {
        if (threadIdx.x < s) red[threadIdx.x] += red[threadIdx.x + s];
        __syncthreads();
    }
    if (threadIdx.x == 0) out[0] = (accumulate ? out[0] : 0.f) + (float)(red[0] * scale);
}

extern "C" size_t rsg_loss_workspace_bytes(int32_t rows) { return (size_t)(rows > 1024 ? rows : 1024) * sizeof(double); }

extern "C" int rsg_joints_mse_loss(const float* pred, const float* gt, const float* weight, int32_t batch,
                                   int32_t joints, int32_t pixels, float grad_scale, float* loss, int32_t accumulate,
                                   float* grad, void* workspace, void* stream) {
    if (!pred || !gt || !loss || !workspace) RSG_FAIL("joints_mse_loss: null pointer");
    if (batch < 1 || joints < 1 || pixels < 1) RSG_FAIL("joints_mse_loss: empty input");
    if (((uintptr_t)pred | (uintptr_t)gt | (uintptr_t)grad) & 15 || (pixels & 3))
        RSG_FAIL("joints_mse_loss: operands must be 16-byte aligned and pixels a multiple of 4");
    const int rows = batch * joints;
    const double norm = 1.0 / ((double)batch * pixels * joints);
    RSG_LAUNCH(joints_mse_kernel, dim3(rows), dim3(256), 0, (hipStream_t)stream, pred, gt, weight, pixels,
               (float)(norm * grad_scale), (double*)workspace, grad);
    RSG_LAUNCH(final_sum_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, (const double*)workspace, rows, 0.5 * norm,
               loss, accumulate);
    RSG_CHECK_LAUNCH("joints_mse_loss");
    return 0;
}

// ---- BCELoss (mean) on probabilities, torch's clamps: log >= -100, gradient denominator >= 1e-12 -----------
__global__ void __launch_bounds__(256) bce_kernel(const float* __restrict__ x, const float* __restrict__ t, long n,
                                                  float gscale, double* __restrict__ partial, float* __restrict__ grad) {
    double acc = 0.0;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) {
        const float xv = x[i], tv = t[i];
        const float lx = fmaxf(logf(xv), -100.f), l1 = fmaxf(log1pf(-xv), -100.f);
        acc -= (double)(tv * lx + (1.f - tv) * l1);
        if (grad) grad[i] = (xv - tv) / fmaxf((1.f - xv) * xv, 1e-12f) * gscale;
    }
    __shared__ double red[256];
    red[threadIdx.x] = acc;
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) {
        if (threadIdx.x < s) red[threadIdx.x] += red[threadIdx.x + s];
        __syncthreads();
    }
    if (threadIdx.x == 0) partial[blockIdx.x] = red[0];
}

extern "C" int rsg_bce_loss(const float* prob, const float* target, int64_t n, float loss_scale, float grad_scale,
                            float* loss, int32_t accumulate, float* grad, void* workspace, void* stream) {
    if (!prob || !target || !loss || !workspace || n < 1) RSG_FAIL("bce_loss: bad arguments");
    const int blocks = (int)(n / 256 < 1 ? 1 : (n / 256 > 1024 ? 1024 : n / 256));
    RSG_LAUNCH(bce_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, prob, target, (long)n,
               (float)((double)loss_scale * grad_scale / (double)n), (double*)workspace, grad);
    RSG_LAUNCH(final_sum_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, (const double*)workspace, blocks,
               (double)loss_scale / (double)n, loss, accumulate);
    RSG_CHECK_LAUNCH("bce_loss");
    return 0;
}

// ---- relation target vector: p[b, y, x] = bilinear x0.5 (align_corners) of max_j target[b, j] -----------------
__global__ void __launch_bounds__(256) relation_vec_kernel(const float* __restrict__ t, float* __restrict__ out, int B,
                                                           int J, int H, int W) {
    const int Ho = H / 2, Wo = W / 2;
    const int id = blockIdx.x * 256 + threadIdx.x;
    if (id >= B * Ho * Wo) return;
    const int b = id / (Ho * Wo), r = id - b * Ho * Wo, yo = r / Wo, xo = r - yo * Wo;
    const float sy = Ho > 1 ? (float)(H - 1) / (float)(Ho - 1) : 0.f, sx = Wo > 1 ? (float)(W - 1) / (float)(Wo - 1) : 0.f;
    const float fy = yo * sy, fx = xo * sx;
    const int y0 = min((int)fy, H - 1), x0 = min((int)fx, W - 1);
    const int y1 = min(y0 + 1, H - 1), x1 = min(x0 + 1, W - 1);
    const float ly = fy - y0, lx = fx - x0;
    float m00 = -3.0e38f, m01 = -3.0e38f, m10 = -3.0e38f, m11 = -3.0e38f;
    for (int j = 0; j < J; ++j) {
        const float* p = t + ((size_t)b * J + j) * H * W;
        m00 = fmaxf(m00, p[y0 * W + x0]); m01 = fmaxf(m01, p[y0 * W + x1]);
        m10 = fmaxf(m10, p[y1 * W + x0]); m11 = fmaxf(m11, p[y1 * W + x1]);
    }
    out[id] = (m00 * (1.f - lx) + m01 * lx) * (1.f - ly) + (m10 * (1.f - lx) + m11 * lx) * ly;
}

extern "C" int rsg_relation_target_vec(const float* target, float* out, int32_t batch, int32_t joints, int32_t h,
                                       int32_t w, void* stream) {
    if (!target || !out || batch < 1 || joints < 1 || h < 2 || w < 2) RSG_FAIL("relation_target_vec: bad arguments");
    const int n = batch * (h / 2) * (w / 2);
    RSG_LAUNCH(relation_vec_kernel, dim3(ceil_div(n, 256)), dim3(256), 0, (hipStream_t)stream, target, out, batch, joints, h, w);
    RSG_CHECK_LAUNCH("relation_target_vec");
    return 0;
}

// ---- Adam on a flat fp32 buffer (parameters, gradients and both moments contiguous) --------------------------
__global__ void __launch_bounds__(256) adam_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m,
                                                   float* __restrict__ v, long n, float lr_c, float b1, float b2, float eps,
                                                   float rsqrt_bc2, float gscale) {
    const long i4 = ((long)blockIdx.x * 256 + threadIdx.x) * 4;
    if (i4 + 3 < n) {
        float4 pv = *reinterpret_cast<float4*>(p + i4), mv = *reinterpret_cast<float4*>(m + i4), vv = *reinterpret_cast<float4*>(v + i4);
        const float4 gv = *reinterpret_cast<const float4*>(g + i4);
#define RSG_ADAM1(c)                                                          \
        {                                                                     \
            const float gg = gv.c * gscale;                                   \
            mv.c = b1 * mv.c + (1.f - b1) * gg;                               \
            vv.c = b2 * vv.c + (1.f - b2) * gg * gg;                          \
            pv.c -= lr_c * mv.c / (sqrtf(vv.c) * rsqrt_bc2 + eps);            \
        }
        RSG_ADAM1(x) RSG_ADAM1(y) RSG_ADAM1(z) RSG_ADAM1(w)
#undef RSG_ADAM1
        *reinterpret_cast<float4*>(p + i4) = pv; *reinterpret_cast<float4*>(m + i4) = mv; *reinterpret_cast<float4*>(v + i4) = vv;
    } else {
        for (long i = i4; i < n; ++i) {
            const float gg = g[i] * gscale;
            m[i] = b1 * m[i] + (1.f - b1) * gg;
            v[i] = b2 * v[i] + (1.f - b2) * gg * gg;
            p[i] -= lr_c * m[i] / (sqrtf(v[i]) * rsqrt_bc2 + eps);
        }
    }
}

extern "C" int rsg_adam_step(float* param, const float* grad, float* exp_avg, float* exp_avg_sq, int64_t n, float lr,
                             float beta1, float beta2, float eps, int32_t step, float grad_scale, void* stream) {
    if (!param || !grad || !exp_avg || !exp_avg_sq || n < 1 || step < 1) RSG_FAIL("adam_step: bad arguments");
    if (((uintptr_t)param | (uintptr_t)grad | (uintptr_t)exp_avg | (uintptr_t)exp_avg_sq) & 15)
        RSG_FAIL("adam_step: buffers must be 16-byte aligned");
    const double bc1 = 1.0 - pow((double)beta1, step), bc2 = 1.0 - pow((double)beta2, step);
    RSG_LAUNCH(adam_kernel, dim3((unsigned)ceil_div((int)((n + 3) / 4), 256)), dim3(256), 0, (hipStream_t)stream, param, grad,
               exp_avg, exp_avg_sq, (long)n, (float)(lr / bc1), beta1, beta2, eps, (float)(1.0 / sqrt(bc2)), grad_scale);
    RSG_CHECK_LAUNCH("adam_step");
    return 0;
}
