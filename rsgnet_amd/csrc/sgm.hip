// SGM (Skeleton Graph Machine) graph convolution: KTMachine.forward of /root/reference/lib/models/pose_rsgnet.py:576-600 --
//
//   K  = matrix_limb (.) real_matrix_limb            [L,J]   learned limb x joint weights masked by the 0/1 skeleton prior (:576-578)
//   T  = K . Wk                                      [L,D]   Wk = final_layer.weight [J,D,1,1]: every limb gathers its joints' classifiers (:595-596)
//   Wl = Linear2(LeakyReLU_0.02(Linear1(T)))         [L,D]   kpt_transformer (:517-521, :597); the limb classifier of :1005
//
// and its backward (training: the reference's autograd through the same four statements).  L <= 32 limbs, J <= 32 joints,
// D <= 64 channels (18 x 17 x 32 on COCO, 14 x 14 x 32 / 48 on CrowdPose): a few thousand multiply-adds, so ONE workgroup of
// 256 threads does all of it, fp32, operands in LDS, every sum in index order (the same bits on every run).  The point of
// running it here rather than as four framework operators is the training step's launch count and that the plan's limb
// classifier comes from the library's own arithmetic (rsgnet_amd/plan_rsgnet.py, train_rsgnet.py).
#include "common.h"

#define SGM_MAX_L 32
#define SGM_MAX_J 32
#define SGM_MAX_D 64
#define SGM_THREADS 256

struct SgmArgs {
    const float *limb, *prior, *wk;            // [L,J], [L,J], [J,D]
    const float *w1, *b1, *w2, *b2;            // Linear: y = x W^T + b, W [D,D] (torch layout: [out,in])
    float *wl, *t, *h;                         // out [L,D]; saved for the backward: T [L,D], Linear1's output before the LeakyReLU [L,D]
    // backward only
    const float* d_wl;                         // [L,D]
    float *d_limb, *d_wk, *d_w1, *d_b1, *d_w2, *d_b2;
    int L, J, D;
    float slope;
};

// stage `n` floats into LDS (zero-padded to `pad`)
__device__ __forceinline__ void sgm_load(float* dst, const float* src, int n) {
    for (int i = threadIdx.x; i < n; i += SGM_THREADS) dst[i] = src[i];
}

__global__ void __launch_bounds__(SGM_THREADS) sgm_forward_kernel(const SgmArgs a) {
    __shared__ float sK[SGM_MAX_L * SGM_MAX_J], sWk[SGM_MAX_J * SGM_MAX_D], sT[SGM_MAX_L * SGM_MAX_D], sA[SGM_MAX_L * SGM_MAX_D];
    __shared__ float sW[SGM_MAX_D * (SGM_MAX_D + 1)];      // a Linear's weight, rows padded by one float (a thread walks a row)
    const int L = a.L, J = a.J, D = a.D, LD = L * D;
    for (int i = threadIdx.x; i < L * J; i += SGM_THREADS) sK[i] = a.limb[i] * a.prior[i];
    sgm_load(sWk, a.wk, J * D);
    for (int i = threadIdx.x; i < D * D; i += SGM_THREADS) sW[(i / D) * (D + 1) + i % D] = a.w1[i];
    __syncthreads();
    for (int i = threadIdx.x; i < LD; i += SGM_THREADS) {          // T = K . Wk
        const int l = i / D, d = i - l * D;
        float acc = 0.f;
        for (int j = 0; j < J; ++j) acc = fmaf(sK[l * J + j], sWk[j * D + d], acc);
        sT[i] = acc;
        a.t[i] = acc;
    }
    __syncthreads();
    for (int i = threadIdx.x; i < LD; i += SGM_THREADS) {          // H = T W1^T + b1; A = LeakyReLU(H)
        const int l = i / D, o = i - l * D;
        float acc = a.b1[o];
        for (int k = 0; k < D; ++k) acc = fmaf(sT[l * D + k], sW[o * (D + 1) + k], acc);
        a.h[i] = acc;
        sA[i] = acc > 0.f ? acc : acc * a.slope;
    }
    __syncthreads();
    for (int i = threadIdx.x; i < D * D; i += SGM_THREADS) sW[(i / D) * (D + 1) + i % D] = a.w2[i];
    __syncthreads();
    for (int i = threadIdx.x; i < LD; i += SGM_THREADS) {          // Wl = A W2^T + b2
        const int l = i / D, o = i - l * D;
        float acc = a.b2[o];
        for (int k = 0; k < D; ++k) acc = fmaf(sA[l * D + k], sW[o * (D + 1) + k], acc);
        a.wl[i] = acc;
    }
}

// Backward.  With dWl given:  dB2 = sum_l dWl;  dW2 = dWl^T A;  dA = dWl W2;  dH = dA (.) LeakyReLU'(H);  dB1 = sum_l dH;
// dW1 = dH^T T;  dT = dH W1;  dK = dT Wk^T;  d matrix_limb = dK (.) real_matrix_limb;  dWk = K^T dT.
__global__ void __launch_bounds__(SGM_THREADS) sgm_backward_kernel(const SgmArgs a) {
    __shared__ float sK[SGM_MAX_L * SGM_MAX_J], sWk[SGM_MAX_J * SGM_MAX_D], sT[SGM_MAX_L * SGM_MAX_D], sA[SGM_MAX_L * SGM_MAX_D];
    __shared__ float sG[SGM_MAX_L * SGM_MAX_D], sDH[SGM_MAX_L * SGM_MAX_D];
    __shared__ float sW[SGM_MAX_D * (SGM_MAX_D + 1)];
    float* const sDT = sG;                  // dWl is dead once dH exists (60.6 KB of the 64 KB a kernel may declare)
    const int L = a.L, J = a.J, D = a.D, LD = L * D;
    for (int i = threadIdx.x; i < L * J; i += SGM_THREADS) sK[i] = a.limb[i] * a.prior[i];
    sgm_load(sWk, a.wk, J * D);
    sgm_load(sT, a.t, LD);
    sgm_load(sG, a.d_wl, LD);
    for (int i = threadIdx.x; i < LD; i += SGM_THREADS) { const float h = a.h[i]; sA[i] = h > 0.f ? h : h * a.slope; }
    for (int i = threadIdx.x; i < D * D; i += SGM_THREADS) sW[(i / D) * (D + 1) + i % D] = a.w2[i];
    __syncthreads();
    for (int o = threadIdx.x; o < D; o += SGM_THREADS) {            // dB2
        float acc = 0.f;
        for (int l = 0; l < L; ++l) acc += sG[l * D + o];
        a.d_b2[o] = acc;
    }
    for (int i = threadIdx.x; i < D * D; i += SGM_THREADS) {        // dW2[o,k] = sum_l dWl[l,o] A[l,k]
        const int o = i / D, k = i - o * D;
        float acc = 0.f;
        for (int l = 0; l < L; ++l) acc = fmaf(sG[l * D + o], sA[l * D + k], acc);
        a.d_w2[i] = acc;
    }
    for (int i = threadIdx.x; i < LD; i += SGM_THREADS) {           // dH[l,k] = LeakyReLU'(H[l,k]) sum_o dWl[l,o] W2[o,k]
        const int l = i / D, k = i - l * D;
        float acc = 0.f;
        for (int o = 0; o < D; ++o) acc = fmaf(sG[l * D + o], sW[o * (D + 1) + k], acc);
        sDH[i] = a.h[i] > 0.f ? acc : acc * a.slope;
    }
    __syncthreads();
    for (int i = threadIdx.x; i < D * D; i += SGM_THREADS) sW[(i / D) * (D + 1) + i % D] = a.w1[i];
    for (int o = threadIdx.x; o < D; o += SGM_THREADS) {            // dB1
        float acc = 0.f;
        for (int l = 0; l < L; ++l) acc += sDH[l * D + o];
        a.d_b1[o] = acc;
    }
    for (int i = threadIdx.x; i < D * D; i += SGM_THREADS) {        // dW1[o,k] = sum_l dH[l,o] T[l,k]
        const int o = i / D, k = i - o * D;
        float acc = 0.f;
        for (int l = 0; l < L; ++l) acc = fmaf(sDH[l * D + o], sT[l * D + k], acc);
        a.d_w1[i] = acc;
    }
    __syncthreads();
    for (int i = threadIdx.x; i < LD; i += SGM_THREADS) {           // dT[l,k] = sum_o dH[l,o] W1[o,k]
        const int l = i / D, k = i - l * D;
        float acc = 0.f;
        for (int o = 0; o < D; ++o) acc = fmaf(sDH[l * D + o], sW[o * (D + 1) + k], acc);
        sDT[i] = acc;
    }
    __syncthreads();
    for (int i = threadIdx.x; i < L * J; i += SGM_THREADS) {        // d matrix_limb[l,j] = prior[l,j] sum_d dT[l,d] Wk[j,d]
        const int l = i / J, j = i - l * J;
        float acc = 0.f;
        for (int d = 0; d < D; ++d) acc = fmaf(sDT[l * D + d], sWk[j * D + d], acc);
        a.d_limb[i] = acc * a.prior[i];
    }
    for (int i = threadIdx.x; i < J * D; i += SGM_THREADS) {        // dWk[j,d] = sum_l K[l,j] dT[l,d]
        const int j = i / D, d = i - j * D;
        float acc = 0.f;
        for (int l = 0; l < L; ++l) acc = fmaf(sK[l * J + j], sDT[l * D + d], acc);
        a.d_wk[i] = acc;
    }
}

static int sgm_check(const SgmArgs& a) {
    if (a.L < 1 || a.L > SGM_MAX_L || a.J < 1 || a.J > SGM_MAX_J || a.D < 1 || a.D > SGM_MAX_D)
        RSG_FAIL("sgm_graph_conv: %d limbs x %d joints x %d channels (at most %d x %d x %d)", a.L, a.J, a.D, SGM_MAX_L, SGM_MAX_J, SGM_MAX_D);
    if (!a.limb || !a.prior || !a.wk || !a.w1 || !a.b1 || !a.w2 || !a.b2 || !a.t || !a.h) RSG_FAIL("sgm_graph_conv: null pointer");
    return 0;
}

extern "C" int rsg_sgm_graph_conv(const float* matrix_limb, const float* prior, const float* wk, const float* w1, const float* b1,
                                  const float* w2, const float* b2, float* wl, float* t_saved, float* h_saved, int32_t limbs,
                                  int32_t joints, int32_t channels, float slope, void* stream) {
    SgmArgs a;
    memset(&a, 0, sizeof(a));
    a.limb = matrix_limb; a.prior = prior; a.wk = wk; a.w1 = w1; a.b1 = b1; a.w2 = w2; a.b2 = b2;
    a.wl = wl; a.t = t_saved; a.h = h_saved; a.L = limbs; a.J = joints; a.D = channels; a.slope = slope;
    if (sgm_check(a)) return 1;
    if (!wl) RSG_FAIL("sgm_graph_conv: null output");
    RSG_LAUNCH(sgm_forward_kernel, dim3(1), dim3(SGM_THREADS), 0, (hipStream_t)stream, a);
    RSG_CHECK_LAUNCH("sgm_graph_conv");
    return 0;
}

extern "C" int rsg_sgm_graph_conv_backward(const float* matrix_limb, const float* prior, const float* wk, const float* w1,
                                           const float* w2, const float* t_saved, const float* h_saved, const float* d_wl,
                                           float* d_matrix_limb, float* d_wk, float* d_w1, float* d_b1, float* d_w2, float* d_b2,
                                           int32_t limbs, int32_t joints, int32_t channels, float slope, void* stream) {
    SgmArgs a;
    memset(&a, 0, sizeof(a));
    a.limb = matrix_limb; a.prior = prior; a.wk = wk; a.w1 = w1; a.w2 = w2; a.b1 = w1; a.b2 = w2;     // (biases are not read)
    a.t = const_cast<float*>(t_saved); a.h = const_cast<float*>(h_saved); a.d_wl = d_wl;
    a.d_limb = d_matrix_limb; a.d_wk = d_wk; a.d_w1 = d_w1; a.d_b1 = d_b1; a.d_w2 = d_w2; a.d_b2 = d_b2;
    a.L = limbs; a.J = joints; a.D = channels; a.slope = slope;
    if (sgm_check(a)) return 1;
    if (!d_wl || !d_matrix_limb || !d_wk || !d_w1 || !d_b1 || !d_w2 || !d_b2) RSG_FAIL("sgm_graph_conv_backward: null pointer");
    RSG_LAUNCH(sgm_backward_kernel, dim3(1), dim3(SGM_THREADS), 0, (hipStream_t)stream, a);
    RSG_CHECK_LAUNCH("sgm_graph_conv_backward");
    return 0;
}
