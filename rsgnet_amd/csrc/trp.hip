// TRP = SpatialRelationHead (/root/reference/lib/models/association.py:280-301) for gfx950.
//
//   g   = conv1x1(x) + b                         [B,S,C]
//   R   = sigmoid(x x^T)                          [B,S,S]   (theta = phi = x, un-normalised)
//   E   = R g                                     [B,S,C]
//   out = GroupNorm8(conv1x1(E) + b)
//
// The reference materialises R (37.7 MB fp32 per 256x192 crop).  Here one wave owns 32 query
// positions and streams over the keys in tiles of 32:
//   X  = K_tile . Q^T     two v_mfma_f32_32x32x16 (C = 32 channels)  -> X[key][query] in 16 accum regs,
//                          the query on the lane
//   P  = sigmoid(X)        v_exp_f32 + v_rcp_f32 per element, fp32
//   Y += G^T_tile . P      P's accumulator registers ARE the B operand of the next MFMA (it sums over
//                          P's row index = key), converted pairwise to 16 bit; no LDS, no shuffles.
// The k order of that second product is permuted (element j of lane half h is key
// 16s + 8(j>>2) + 4h + (j&3)); rsg_trp_project_g stores g transposed with the keys of every 16-block
// in the order [0-3, 8-11, 4-7, 12-15] so each lane's 8 A-operand values are one 16-byte load.
// E is kept in fp32 (it is ~S times larger than its variation across positions, which is what the
// GroupNorm after it keeps).
#include "common.h"

enum { TRP_ATTN = 0, TRP_SCORES = 1, TRP_MSE = 2 };

// 1/(1 + 2^(-a*log2 e)) on the hardware transcendentals (v_exp_f32, v_rcp_f32: 1 ulp each); the libm
// forms (__expf / __frcp_rn) expand to ~20 instructions per element and made the kernel 3x VALU-bound
__device__ __forceinline__ float sigmoid_f32(float a) {
    return __builtin_amdgcn_rcpf(1.f + __builtin_amdgcn_exp2f(a * -1.4426950408889634f));
}

struct TrpArgs {
    const char* x;       // [B,S,ldx] 16-bit
    const char* gT;      // [B,C,S] 16-bit, key-permuted
    float* e;            // [B,S,C] fp32            (ATTN)
    float* scores;       // [B,S,S] fp32            (SCORES)
    const float* p;      // [B,S] fp32              (MSE)
    float* mse;          // [B] fp32, pre-zeroed    (MSE)
    int S, ldx;
    float inv_ss;
};

template <class DT, int MODE>
__global__ void __launch_bounds__(256) trp_kernel(const TrpArgs a) {
    constexpr int C = 32;
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int n = lane & 31, hh = lane >> 5;
    const int b = blockIdx.y;
    const int q0 = (blockIdx.x * 4 + wave) * 32;          // this wave's 32 queries
    const char* xb = a.x + (size_t)b * a.S * a.ldx * 2;

    // query fragments (B operand of X = K.Q^T): lane (i=n, hh) holds x[q0+i][16s+8hh .. +8]
    typename DT::vec8 qf[2];
#pragma unroll
    for (int s = 0; s < 2; ++s)
        qf[s] = __builtin_bit_cast(typename DT::vec8,
                                   *reinterpret_cast<const uint4*>(xb + ((size_t)(q0 + n) * a.ldx + 16 * s + 8 * hh) * 2));

    f32x16 y;
#pragma unroll
    for (int r = 0; r < 16; ++r) y[r] = 0.f;
    float pq = 0.f, part = 0.f;
    if (MODE == TRP_MSE) pq = a.p[(size_t)b * a.S + q0 + n];

    // K / G^T tiles of 32 keys are shared by the 4 waves of the workgroup: each thread fetches ONE 16-byte
    // piece (threads 0-127: keys x 64 B of x, threads 128-255: 32 channel rows x 64 B of g^T), coalesced,
    // into a double-buffered LDS tile with 80-byte rows (5 slots: the 16-lane ds_read_b128 groups hit 16
    // different banks).  Per-wave scattered global loads of the same data saturated the address path.
    __shared__ uint4 tiles[2][2][32 * 5];
    const int trow = (threadIdx.x & 127) >> 2, tpiece = threadIdx.x & 3;
    const bool is_g = threadIdx.x >= 128;
    const bool loader = !is_g || MODE == TRP_ATTN;
    auto fetch = [&](int j0) -> uint4 {
        if (!loader) return make_uint4(0, 0, 0, 0);
        const char* src = is_g ? a.gT + (((size_t)b * C + trow) * a.S + j0 + tpiece * 8) * 2
                               : xb + ((size_t)(j0 + trow) * a.ldx + tpiece * 8) * 2;
        return *reinterpret_cast<const uint4*>(src);
    };
    auto stash = [&](int buf, uint4 v) {
        if (loader) tiles[buf][is_g ? 1 : 0][trow * 5 + tpiece] = v;
    };
    auto load_tile = [&](int buf, typename DT::vec8 (&kf)[2], typename DT::vec8 (&gf)[2]) {
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            kf[s] = __builtin_bit_cast(typename DT::vec8, tiles[buf][0][n * 5 + 2 * s + hh]);
            if (MODE == TRP_ATTN) gf[s] = __builtin_bit_cast(typename DT::vec8, tiles[buf][1][n * 5 + 2 * s + hh]);
        }
    };
    auto do_tile = [&](int j0, const typename DT::vec8 (&kf)[2], const typename DT::vec8 (&gf)[2]) {
        f32x16 xacc;
#pragma unroll
        for (int r = 0; r < 16; ++r) xacc[r] = 0.f;
        xacc = DT::mfma(kf[0], qf[0], xacc);
        xacc = DT::mfma(kf[1], qf[1], xacc);
        // xacc[r] = <x_key, x_query>, key = j0 + (r&3) + 8*(r>>2) + 4*hh, query = q0 + n
        float pr[16];
#pragma unroll
        for (int r = 0; r < 16; ++r) pr[r] = sigmoid_f32(xacc[r]);
        if (MODE == TRP_ATTN) {
#pragma unroll
            for (int s = 0; s < 2; ++s) {
                typename DT::vec8 pf;
#pragma unroll
                for (int e = 0; e < 8; ++e) pf[e] = (typename DT::elem)pr[8 * s + e];
                y = DT::mfma(gf[s], pf, y);
            }
        } else if (MODE == TRP_SCORES) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int key = j0 + (r & 3) + 8 * (r >> 2) + 4 * hh;
                a.scores[((size_t)b * a.S + key) * a.S + q0 + n] = pr[r];   // R is symmetric
            }
        } else {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int key = j0 + (r & 3) + 8 * (r >> 2) + 4 * hh;
                const float d = pq * a.p[(size_t)b * a.S + key] - pr[r];
                part += d * d;
            }
        }
    };
    // one barrier per tile: fetch tile t+1 into registers, compute tile t from LDS, stash, barrier
    const int ntiles = a.S / 32;
    stash(0, fetch(0));
    __syncthreads();
    for (int t = 0; t < ntiles; ++t) {
        const bool more = t + 1 < ntiles;
        uint4 nxt = make_uint4(0, 0, 0, 0);
        if (more) nxt = fetch((t + 1) * 32);
        typename DT::vec8 kf[2], gf[2];
        load_tile(t & 1, kf, gf);
        do_tile(t * 32, kf, gf);
        if (more) stash((t + 1) & 1, nxt);
        __syncthreads();
    }

    if (MODE == TRP_ATTN) {
        // y[r] = E[query q0+n][channel (r&3) + 8*(r>>2) + 4*hh]
        float* eo = a.e + ((size_t)b * a.S + q0 + n) * C + 4 * hh;
#pragma unroll
        for (int k = 0; k < 4; ++k)
            *reinterpret_cast<float4*>(eo + 8 * k) = make_float4(y[4 * k], y[4 * k + 1], y[4 * k + 2], y[4 * k + 3]);
    } else if (MODE == TRP_MSE) {
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) part += __shfl_xor(part, o);
        if (lane == 0) atomicAdd(a.mse + b, part * a.inv_ss);
    }
}

static int trp_check(const void* x, int ld_x, int batch, int S, int C) {
    if (!x || batch < 1) RSG_FAIL("trp: null input");
    if (C != 32) RSG_FAIL("trp: C=%d (only 32 is built)", C);
    if (S % 128 || S < 128) RSG_FAIL("trp: S=%d must be a multiple of 128", S);
    if (ld_x < C || ld_x % 8) RSG_FAIL("trp: ld_x=%d", ld_x);
    return 0;
}

template <int MODE>
static int trp_launch(const TrpArgs& a, int batch, int dtype, void* stream) {
    dim3 grid(ceil_div(a.S, 128), batch);
    if (dtype == RSG_BF16)
        RSG_LAUNCH((trp_kernel<BF16, MODE>), grid, dim3(256), 0, (hipStream_t)stream, a);
    else
        RSG_LAUNCH((trp_kernel<F16, MODE>), grid, dim3(256), 0, (hipStream_t)stream, a);
    RSG_CHECK_LAUNCH("trp");
    return 0;
}

extern "C" int rsg_trp_attention(const void* x, int32_t ld_x, const void* gT_hi, const void* gT_lo,
                                 float* e_out, int32_t batch, int32_t S, int32_t C, int32_t dtype,
                                 void* stream) {
    if (trp_check(x, ld_x, batch, S, C)) return 1;
    if (!gT_hi || !e_out) RSG_FAIL("trp_attention: null pointer");
    if (gT_lo) RSG_FAIL("trp_attention: split (hi+lo) values are not built; pass gT_lo = NULL");
    TrpArgs a;
    memset(&a, 0, sizeof(a));
    a.x = (const char*)x; a.gT = (const char*)gT_hi; a.e = e_out; a.S = S; a.ldx = ld_x;
    return trp_launch<TRP_ATTN>(a, batch, dtype, stream);
}

extern "C" int rsg_trp_scores(const void* x, int32_t ld_x, float* scores, int32_t batch, int32_t S,
                              int32_t C, int32_t dtype, void* stream) {
    if (trp_check(x, ld_x, batch, S, C)) return 1;
    if (!scores) RSG_FAIL("trp_scores: null pointer");
    TrpArgs a;
    memset(&a, 0, sizeof(a));
    a.x = (const char*)x; a.scores = scores; a.S = S; a.ldx = ld_x;
    return trp_launch<TRP_SCORES>(a, batch, dtype, stream);
}

extern "C" int rsg_trp_relation_mse(const void* x, int32_t ld_x, const float* p, float* mse, int32_t batch,
                                    int32_t S, int32_t C, int32_t dtype, void* stream) {
    if (trp_check(x, ld_x, batch, S, C)) return 1;
    if (!p || !mse) RSG_FAIL("trp_relation_mse: null pointer");
    if (hipMemsetAsync(mse, 0, sizeof(float) * batch, (hipStream_t)stream) != hipSuccess)
        RSG_FAIL("trp_relation_mse: memset failed");
    TrpArgs a;
    memset(&a, 0, sizeof(a));
    a.x = (const char*)x; a.p = p; a.mse = mse; a.S = S; a.ldx = ld_x;
    a.inv_ss = 1.0f / ((float)S * (float)S);
    return trp_launch<TRP_MSE>(a, batch, dtype, stream);
}

// ---- g projection, transposed + key-permuted (association.py:288-289) -------------------------------
template <class DT>
__global__ void __launch_bounds__(256) trp_project_g_kernel(const char* __restrict__ x, int ldx,
                                                           const float* __restrict__ w,
                                                           const float* __restrict__ bias,
                                                           char* __restrict__ gT, int S, long total) {
    constexpr int C = 32;
    __shared__ float ws[C * C + C];
    for (int i = threadIdx.x; i < C * C + C; i += 256) ws[i] = i < C * C ? w[i] : bias[i - C * C];
    __syncthreads();
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i >= total) return;
    const long b = i / S;
    const int s = (int)(i - b * S);
    float xv[C];
#pragma unroll
    for (int k = 0; k < C / 8; ++k) {
        const typename DT::vec8 v =
            __builtin_bit_cast(typename DT::vec8, *reinterpret_cast<const uint4*>(x + (i * ldx + 8 * k) * 2));
#pragma unroll
        for (int e = 0; e < 8; ++e) xv[8 * k + e] = (float)v[e];
    }
    // key position inside its 16-block: [0-3, 8-11, 4-7, 12-15]
    const int k16 = s & 15;
    const int pos = (k16 & 3) | ((k16 & 8) >> 1) | ((k16 & 4) << 1);
    typename DT::elem* go = reinterpret_cast<typename DT::elem*>(gT) + (size_t)b * C * S + (s & ~15) + pos;
#pragma unroll 4
    for (int co = 0; co < C; ++co) {
        float acc = ws[C * C + co];
#pragma unroll
        for (int ci = 0; ci < C; ++ci) acc = fmaf(ws[co * C + ci], xv[ci], acc);
        go[(size_t)co * S] = (typename DT::elem)acc;
    }
}

extern "C" int rsg_trp_project_g(const void* x, int32_t ld_x, const float* w, const float* bias, void* gT_hi,
                                 void* gT_lo, int32_t batch, int32_t S, int32_t C, int32_t dtype,
                                 void* stream) {
    if (trp_check(x, ld_x, batch, S, C)) return 1;
    if (!w || !bias || !gT_hi) RSG_FAIL("trp_project_g: null pointer");
    if (gT_lo) RSG_FAIL("trp_project_g: split values are not built; pass gT_lo = NULL");
    const long total = (long)batch * S;
    const int grid = (int)((total + 255) / 256);
    if (dtype == RSG_BF16)
        RSG_LAUNCH(trp_project_g_kernel<BF16>, dim3(grid), dim3(256), 0, (hipStream_t)stream,
                           (const char*)x, ld_x, w, bias, (char*)gT_hi, S, total);
    else
        RSG_LAUNCH(trp_project_g_kernel<F16>, dim3(grid), dim3(256), 0, (hipStream_t)stream,
                           (const char*)x, ld_x, w, bias, (char*)gT_hi, S, total);
    RSG_CHECK_LAUNCH("trp_project_g");
    return 0;
}

// ---- W: conv1x1 + bias, GroupNorm(8, C) (association.py:298-300; eps 1e-5, biased variance) -----------
// Two launches.  (1) y = W.E + b per position (fp32, written to the workspace) and, per workgroup of 256
// positions, the (mean, M2) of each of the 8 channel groups.  (2) every workgroup merges its crop's
// partials with Chan's formula (no E[x^2]-E[x]^2 cancellation), normalises its positions and writes 16 bit.
__device__ __forceinline__ float block_sum256(float v, float* red) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
    __syncthreads();
    return red[0] + red[1] + red[2] + red[3];
}

__global__ void __launch_bounds__(256) trp_w_project_kernel(const float* __restrict__ e, const float* __restrict__ w,
                                                           const float* __restrict__ bias, float* __restrict__ y,
                                                           float* __restrict__ part, int S) {
    constexpr int C = 32, G = 8, CG = 4;
    __shared__ float ws[C * C + C];
    __shared__ float red[4];
    for (int i = threadIdx.x; i < C * C + C; i += 256) ws[i] = i < C * C ? w[i] : bias[i - C * C];
    __syncthreads();
    const int b = blockIdx.y, s = blockIdx.x * 256 + threadIdx.x;
    const size_t row = ((size_t)b * S + s) * C;
    float xv[C], yv[C];
#pragma unroll
    for (int k = 0; k < C / 4; ++k) {
        const float4 v = *reinterpret_cast<const float4*>(e + row + 4 * k);
        xv[4 * k] = v.x; xv[4 * k + 1] = v.y; xv[4 * k + 2] = v.z; xv[4 * k + 3] = v.w;
    }
#pragma unroll
    for (int co = 0; co < C; ++co) {
        float acc = ws[C * C + co];
#pragma unroll
        for (int ci = 0; ci < C; ++ci) acc = fmaf(ws[co * C + ci], xv[ci], acc);
        yv[co] = acc;
    }
#pragma unroll
    for (int k = 0; k < C / 4; ++k)
        *reinterpret_cast<float4*>(y + row + 4 * k) = make_float4(yv[4 * k], yv[4 * k + 1], yv[4 * k + 2], yv[4 * k + 3]);
    float* po = part + (((size_t)b * gridDim.x + blockIdx.x) * G) * 2;
#pragma unroll
    for (int g = 0; g < G; ++g) {
        const float sum = (yv[4 * g] + yv[4 * g + 1]) + (yv[4 * g + 2] + yv[4 * g + 3]);
        const float mean = block_sum256(sum, red) * (1.f / (256 * CG));
        float m2 = 0.f;
#pragma unroll
        for (int c = 0; c < CG; ++c) m2 += (yv[4 * g + c] - mean) * (yv[4 * g + c] - mean);
        m2 = block_sum256(m2, red);
        if (threadIdx.x == 0) { po[2 * g] = mean; po[2 * g + 1] = m2; }
    }
}

template <class DT>
__global__ void __launch_bounds__(256) trp_gn_apply_kernel(const float* __restrict__ y, const float* __restrict__ part,
                                                          const float* __restrict__ gamma,
                                                          const float* __restrict__ beta, char* __restrict__ out,
                                                          int out_ld, int S) {
    constexpr int C = 32, G = 8, CG = 4;
    __shared__ float scale[C], shift[C];
    const int b = blockIdx.y, nb = gridDim.x;
    if (threadIdx.x < G) {
        const int g = threadIdx.x;
        const float* pi = part + ((size_t)b * nb * G + g) * 2;
        const float nk = 256.f * CG;
        float mean = 0.f;
        for (int k = 0; k < nb; ++k) mean += pi[(size_t)k * G * 2];
        mean /= (float)nb;                               // equal-sized parts
        float m2 = 0.f;
        for (int k = 0; k < nb; ++k) {
            const float d = pi[(size_t)k * G * 2] - mean;
            m2 += pi[(size_t)k * G * 2 + 1] + nk * d * d;
        }
        const float rstd = rsqrtf(m2 / (nk * (float)nb) + 1e-5f);
#pragma unroll
        for (int c = 0; c < CG; ++c) {
            const float ga = gamma[g * CG + c] * rstd;
            scale[g * CG + c] = ga;
            shift[g * CG + c] = beta[g * CG + c] - mean * ga;
        }
    }
    __syncthreads();
    const int s = blockIdx.x * 256 + threadIdx.x;
    const float* yr = y + ((size_t)b * S + s) * C;
    typename DT::elem* orow = reinterpret_cast<typename DT::elem*>(out) + ((size_t)b * S + s) * out_ld;
#pragma unroll
    for (int k = 0; k < C / 8; ++k) {
        const float4 v0 = *reinterpret_cast<const float4*>(yr + 8 * k);
        const float4 v1 = *reinterpret_cast<const float4*>(yr + 8 * k + 4);
        const float vv[8] = {v0.x, v0.y, v0.z, v0.w, v1.x, v1.y, v1.z, v1.w};
        typename DT::vec8 o;
#pragma unroll
        for (int j = 0; j < 8; ++j) o[j] = (typename DT::elem)(vv[j] * scale[8 * k + j] + shift[8 * k + j]);
        *reinterpret_cast<uint4*>(orow + 8 * k) = __builtin_bit_cast(uint4, o);
    }
}

extern "C" size_t rsg_trp_tail_workspace_bytes(int32_t batch, int32_t S, int32_t C) {
    return ((size_t)batch * S * C + (size_t)batch * (S / 256 + 1) * 8 * 2) * sizeof(float);
}

extern "C" int rsg_trp_tail(const float* e, const float* w, const float* bias, const float* gn_gamma,
                            const float* gn_beta, void* out, int32_t out_ld, int32_t batch, int32_t S,
                            int32_t C, int32_t dtype, void* workspace, void* stream) {
    if (!e || !w || !bias || !gn_gamma || !gn_beta || !out || !workspace) RSG_FAIL("trp_tail: null pointer");
    if (C != 32) RSG_FAIL("trp_tail: C=%d (only 32 is built)", C);
    if (S % 256) RSG_FAIL("trp_tail: S=%d must be a multiple of 256", S);
    if (out_ld < C || out_ld % 8 || ((uintptr_t)out & 15)) RSG_FAIL("trp_tail: out_ld=%d / alignment", out_ld);
    float* y = (float*)workspace;
    float* part = y + (size_t)batch * S * C;
    dim3 grid(S / 256, batch);
    RSG_LAUNCH(trp_w_project_kernel, grid, dim3(256), 0, (hipStream_t)stream, e, w, bias, y, part, S);
    RSG_CHECK_LAUNCH("trp_tail(project)");
    if (dtype == RSG_BF16)
        RSG_LAUNCH(trp_gn_apply_kernel<BF16>, grid, dim3(256), 0, (hipStream_t)stream, (const float*)y,
                   (const float*)part, gn_gamma, gn_beta, (char*)out, out_ld, S);
    else
        RSG_LAUNCH(trp_gn_apply_kernel<F16>, grid, dim3(256), 0, (hipStream_t)stream, (const float*)y,
                   (const float*)part, gn_gamma, gn_beta, (char*)out, out_ld, S);
    RSG_CHECK_LAUNCH("trp_tail(groupnorm)");
    return 0;
}
