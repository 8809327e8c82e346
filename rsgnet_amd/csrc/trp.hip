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
#include <algorithm>

enum { TRP_ATTN = 0, TRP_SCORES = 1, TRP_MSE = 2 };

// 1/(1 + 2^(-a*log2 e)) on the hardware transcendentals (v_exp_f32, v_rcp_f32: 1 ulp each); the libm
// forms (__expf / __frcp_rn) expand to ~20 instructions per element and made the kernel 3x VALU-bound
__device__ __forceinline__ float sigmoid_f32(float a) {
    return __builtin_amdgcn_rcpf(1.f + __builtin_amdgcn_exp2f(a * -1.4426950408889634f));
}

// Which 16-byte piece of a K / G^T tile thread p stages.  Rows of four pieces (C = 32; every G^T tile) are padded to five slots
// for the ds_read_b128s of the MFMA operands (5 n mod 16 is a permutation on every 16-lane read group); a ds_write_b128 is
// served in groups of eight lanes whose slots must differ mod 8, and with p -> (row p / 4, piece p % 4) a group spans two rows
// whose slots 5 r and 5 r + 8 collide (a quarter of the LDS cycles of the attention kernel in bank conflicts,
// profiles/r04g_sq_summary_rsgnet_crowdpose_fliptest.json).  Eight consecutive lanes take the SAME piece of eight consecutive
// rows instead: slots 5 r + piece, r = r0 .. r0 + 7 -- 5 r mod 8 is a permutation.  The global side is unchanged in what it
// touches (a wave-instruction still reads 16 whole rows).  Six-piece rows (C = 48) keep the plain order.
template <int PK>
__device__ __forceinline__ int trp_row(int p) { return PK == 4 ? ((p & 7) | ((p >> 5) << 3)) : p / PK; }
template <int PK>
__device__ __forceinline__ int trp_piece(int p) { return PK == 4 ? ((p >> 3) & 3) : p % PK; }

struct TrpArgs {
    const char* x;       // [B,S,ldx] 16-bit
    const char* gT;      // [B,C,S] 16-bit, key-permuted
    float* e;            // [B,S,C] fp32            (ATTN)
    float* scores;       // [B,S,S] fp32            (SCORES)
    const float* p;      // [B,S] fp32              (MSE: factor of the target, T = p p^T)
    const float* t;      // [B,S,S] fp32            (MSE: the target itself, read tile by tile; p is null then)
    float* mse_part;     // [B][S/32] fp32: one partial sum of squared differences per wave (MSE)
    int S, ldx;
    float inv_ss;
};

// C = channels of x and g (32: HRNet-W32 head, 48: W48 head).  The value product pads its output channels
// to CP = 32 or 64 (MFMA rows); the rows of the G^T tile beyond C stay zero.
template <class DT, int MODE, int C>
__global__ void __launch_bounds__(256) trp_kernel(const TrpArgs a) {
    constexpr int KQ = C / 16;                 // k-steps of X = K.Q^T
    constexpr int CP = (C + 31) / 32 * 32, MY = CP / 32;
    constexpr int PK = C / 8;                  // 16-byte pieces per key row
    constexpr int RSK = PK + 1;                // K tile row in 16-byte slots (odd: conflict-free ds_read_b128)
    constexpr int NKP = 32 * PK, NGP = C * 4;  // pieces of the K tile / of the G^T tile (real rows only)
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int n = lane & 31, hh = lane >> 5;
    const int b = blockIdx.y;
    const int q0 = (blockIdx.x * 4 + wave) * 32;          // this wave's 32 queries
    const char* xb = a.x + (size_t)b * a.S * a.ldx * 2;

    // query fragments (B operand of X = K.Q^T): lane (i=n, hh) holds x[q0+i][16s+8hh .. +8]
    typename DT::vec8 qf[KQ];
#pragma unroll
    for (int s = 0; s < KQ; ++s)
        qf[s] = __builtin_bit_cast(typename DT::vec8,
                                   *reinterpret_cast<const uint4*>(xb + ((size_t)(q0 + n) * a.ldx + 16 * s + 8 * hh) * 2));

    f32x16 y[MY];
#pragma unroll
    for (int m = 0; m < MY; ++m)
#pragma unroll
        for (int r = 0; r < 16; ++r) y[m][r] = 0.f;
    float pq = 0.f, part = 0.f;
    if (MODE == TRP_MSE && a.p) pq = a.p[(size_t)b * a.S + q0 + n];

    // K / G^T tiles of 32 keys are shared by the 4 waves of the workgroup: every thread fetches 16-byte
    // pieces (coalesced) into a double-buffered LDS tile; per-wave scattered global loads of the same
    // data saturated the address path (295 -> 145 us at S = 3072).  Rows are padded to an odd number of
    // 16-byte slots so the 16-lane ds_read_b128 groups hit 16 different banks.
    // A stage is KT tiles of 32 keys: one barrier per stage.  A barrier per 32-key tile (~450 issue cycles of work per
    // wave) cost as much as a quarter of the loop.
    constexpr int KT = 4;
    __shared__ uint4 tk[2][KT][32 * RSK];
    __shared__ uint4 tg[2][KT][CP * 5];
    constexpr int NPIECE = NKP + (MODE == TRP_ATTN ? NGP : 0);
    constexpr int PPT = (NPIECE + 255) / 256;             // pieces per thread and tile
    if (MODE == TRP_ATTN)
        for (int i = threadIdx.x; i < 2 * KT * CP * 5; i += 256) (&tg[0][0][0])[i] = make_uint4(0, 0, 0, 0);
    auto fetch = [&](int jstage, uint4 (&v)[KT][PPT]) {
#pragma unroll
        for (int u = 0; u < KT; ++u) {
            const int j0 = jstage + u * 32;
#pragma unroll
            for (int k = 0; k < PPT; ++k) {
                const int p = threadIdx.x + k * 256;
                v[u][k] = make_uint4(0, 0, 0, 0);
                if (p < NKP) {
                    const int row = trp_row<PK>(p), piece = trp_piece<PK>(p);
                    v[u][k] = *reinterpret_cast<const uint4*>(xb + ((size_t)(j0 + row) * a.ldx + piece * 8) * 2);
                } else if (p < NPIECE) {
                    const int q = p - NKP, row = trp_row<4>(q), piece = trp_piece<4>(q);
                    v[u][k] = *reinterpret_cast<const uint4*>(a.gT + (((size_t)b * C + row) * a.S + j0 + piece * 8) * 2);
                }
            }
        }
    };
    auto stash = [&](int buf, const uint4 (&v)[KT][PPT]) {
#pragma unroll
        for (int u = 0; u < KT; ++u)
#pragma unroll
            for (int k = 0; k < PPT; ++k) {
                const int p = threadIdx.x + k * 256;
                if (p < NKP) tk[buf][u][trp_row<PK>(p) * RSK + trp_piece<PK>(p)] = v[u][k];
                else if (p < NPIECE) tg[buf][u][trp_row<4>(p - NKP) * 5 + trp_piece<4>(p - NKP)] = v[u][k];
            }
    };
    auto do_tile = [&](int j0, const uint4* tkb, const uint4* tgb) {
        f32x16 xacc;
#pragma unroll
        for (int r = 0; r < 16; ++r) xacc[r] = 0.f;
#pragma unroll
        for (int s = 0; s < KQ; ++s)
            xacc = DT::mfma(__builtin_bit_cast(typename DT::vec8, tkb[n * RSK + 2 * s + hh]), qf[s], xacc);
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
#pragma unroll
                for (int m = 0; m < MY; ++m)
                    y[m] = DT::mfma(__builtin_bit_cast(typename DT::vec8, tgb[(32 * m + n) * 5 + 2 * s + hh]), pf, y[m]);
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
                // the sum over all (i, j) takes every element of T once whichever of T_ij / T_ji a lane reads: row `key`,
                // columns q0 .. q0 + 31 are 128 contiguous bytes per (r, hh)
                const float tv = a.t ? a.t[((size_t)b * a.S + key) * a.S + q0 + n] : pq * a.p[(size_t)b * a.S + key];
                const float d = tv - pr[r];
                part += d * d;
            }
        }
    };
    // one barrier per stage: fetch stage t+1 into registers, compute the KT tiles of stage t from LDS, stash, barrier
    const int nstages = a.S / (32 * KT);
    uint4 nxt[KT][PPT];
    fetch(0, nxt);
    __syncthreads();                                       // the zero fill of tg is complete
    stash(0, nxt);
    __syncthreads();
    for (int t = 0; t < nstages; ++t) {
        const bool more = t + 1 < nstages;
        if (more) fetch((t + 1) * 32 * KT, nxt);
#pragma unroll
        for (int u = 0; u < KT; ++u) do_tile((t * KT + u) * 32, tk[t & 1][u], tg[t & 1][u]);
        if (more) stash((t + 1) & 1, nxt);
        __syncthreads();
    }

    if (MODE == TRP_ATTN) {
        // y[m][r] = E[query q0+n][channel 32m + (r&3) + 8*(r>>2) + 4*hh]
        float* eo = a.e + ((size_t)b * a.S + q0 + n) * C + 4 * hh;
#pragma unroll
        for (int m = 0; m < MY; ++m)
#pragma unroll
            for (int k = 0; k < 4; ++k)
                if (32 * m + 8 * k < C)
                    *reinterpret_cast<float4*>(eo + 32 * m + 8 * k) =
                        make_float4(y[m][4 * k], y[m][4 * k + 1], y[m][4 * k + 2], y[m][4 * k + 3]);
    } else if (MODE == TRP_MSE) {
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) part += __shfl_xor(part, o);
        // one partial per wave, summed in a fixed order by trp_mse_sum_kernel: no atomics, no pre-zeroed output, the
        // same bits on every run (and inside a replayed graph)
        if (lane == 0) a.mse_part[(size_t)b * (a.S / 32) + blockIdx.x * 4 + wave] = part;
    }
}

static int trp_check(const void* x, int ld_x, int batch, int S, int C) {
    if (!x || batch < 1) RSG_FAIL("trp: null input");
    if (C != 32 && C != 48) RSG_FAIL("trp: C=%d (32 and 48 are built)", C);
    if (S % 128 || S < 128) RSG_FAIL("trp: S=%d must be a multiple of 128", S);
    if (ld_x < C || ld_x % 8 || ((uintptr_t)x & 15)) RSG_FAIL("trp: ld_x=%d / alignment", ld_x);
    return 0;
}

template <int MODE>
static int trp_launch(const TrpArgs& a, int batch, int C, int dtype, void* stream) {
    dim3 grid(ceil_div(a.S, 128), batch);
#define RSG_TRP(DT_, C_) RSG_LAUNCH((trp_kernel<DT_, MODE, C_>), grid, dim3(256), 0, (hipStream_t)stream, a)
    if (dtype == RSG_BF16) { if (C == 32) RSG_TRP(BF16, 32); else RSG_TRP(BF16, 48); }
    else { if (C == 32) RSG_TRP(F16, 32); else RSG_TRP(F16, 48); }
#undef RSG_TRP
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
    return trp_launch<TRP_ATTN>(a, batch, C, dtype, stream);
}

extern "C" int rsg_trp_scores(const void* x, int32_t ld_x, float* scores, int32_t batch, int32_t S,
                              int32_t C, int32_t dtype, void* stream) {
    if (trp_check(x, ld_x, batch, S, C)) return 1;
    if (!scores) RSG_FAIL("trp_scores: null pointer");
    TrpArgs a;
    memset(&a, 0, sizeof(a));
    a.x = (const char*)x; a.scores = scores; a.S = S; a.ldx = ld_x;
    return trp_launch<TRP_SCORES>(a, batch, C, dtype, stream);
}

// mse[b] = inv_ss * sum of the image's S/32 per-wave partials, in index order (double accumulator)
__global__ void __launch_bounds__(64) trp_mse_sum_kernel(const float* __restrict__ part, float* __restrict__ mse, int nper,
                                                         int batch, float inv_ss) {
    const int b = blockIdx.x * 64 + threadIdx.x;
    if (b >= batch) return;
    double acc = 0.0;
    for (int i = 0; i < nper; ++i) acc += (double)part[(size_t)b * nper + i];
    mse[b] = (float)(acc * (double)inv_ss);
}

extern "C" size_t rsg_trp_mse_workspace_bytes(int32_t batch, int32_t S) {
    if (batch < 1 || S < 128 || S % 128) return 0;
    return (size_t)batch * (S / 32) * sizeof(float);
}

static int trp_mse(const void* x, int32_t ld_x, const float* p, const float* t, float* mse, void* workspace, int32_t batch,
                   int32_t S, int32_t C, int32_t dtype, void* stream) {
    if (trp_check(x, ld_x, batch, S, C)) return 1;
    if ((!p && !t) || !mse) RSG_FAIL("trp_relation_mse: null pointer");
    if (!workspace || ((uintptr_t)workspace & 3)) RSG_FAIL("trp_relation_mse: workspace (rsg_trp_mse_workspace_bytes) missing");
    TrpArgs a;
    memset(&a, 0, sizeof(a));
    a.x = (const char*)x; a.p = p; a.t = t; a.mse_part = (float*)workspace; a.S = S; a.ldx = ld_x;
    a.inv_ss = 1.0f / ((float)S * (float)S);
    if (trp_launch<TRP_MSE>(a, batch, C, dtype, stream)) return 1;
    RSG_LAUNCH(trp_mse_sum_kernel, dim3(ceil_div(batch, 64)), dim3(64), 0, (hipStream_t)stream, (const float*)workspace, mse,
               S / 32, batch, a.inv_ss);
    RSG_CHECK_LAUNCH("trp_relation_mse(sum)");
    return 0;
}

extern "C" int rsg_trp_relation_mse(const void* x, int32_t ld_x, const float* p, float* mse, void* workspace, int32_t batch,
                                    int32_t S, int32_t C, int32_t dtype, void* stream) {
    return trp_mse(x, ld_x, p, nullptr, mse, workspace, batch, S, C, dtype, stream);
}

extern "C" int rsg_trp_relation_mse_dense(const void* x, int32_t ld_x, const float* target, float* mse, void* workspace,
                                          int32_t batch, int32_t S, int32_t C, int32_t dtype, void* stream) {
    if (!target || ((uintptr_t)target & 3)) RSG_FAIL("trp_relation_mse_dense: the target [batch,S,S] fp32");
    return trp_mse(x, ld_x, nullptr, target, mse, workspace, batch, S, C, dtype, stream);
}

// ---- TRP backward (training; pose_rsgnet.py:987,1014-1018 under autograd) ---------------------------------------
// Forward, per crop:  G = X Wg^T + bg,  R = sigmoid(X X^T),  E = R G,  mse = mean_ij (p_i p_j - R_ij)^2.
// Given dE [S,C] and the upstream scalar of the relation loss (coef = dL/dmse * 2 / S^2), with R' = R (1 - R):
//     dG_i = sum_j R_ij dE_j                                                    (R is symmetric: the forward's kernel with dE as values)
//     dX_i = sum_j M_ij X_j,   M_ij = R'_ij [ dE_i.G_j + dE_j.G_i + 2 coef (R_ij - p_i p_j) ]     (A = X X^T: dA + dA^T)
// The S x S map is never stored: one wave owns 32 query rows i and streams over the keys j in tiles of 32, recomputing
// sigmoid(X X^T) per tile as the forward does.  Per tile three score products (X_j.X_i, G_j.dE_i, dE_j.G_i: K = C each)
// and two value products (dE_j^T.P -> dG_i, X_j^T.M -> dX_i) run on the matrix cores; P and M go from accumulator
// registers straight into the next MFMA's B operand (the forward's trick).  Operands are 16-bit (bf16 in training),
// accumulation and outputs fp32.  Inputs: x, g, de as rows [B,S,C]; xT, deT transposed [B,C,S] with the keys of every
// 16-block in the order [0-3, 8-11, 4-7, 12-15] (rsg_trp_project_g's layout).
struct TrpBwdArgs {
    const char *x, *g, *de;        // [B,S,ld*] 16-bit rows
    const char *xT, *deT;          // [B,C,S] 16-bit, key-permuted
    const float* p;                // [B,S] factor of the relation target, or null
    const float* t;                // [B,S,S] the relation target itself (symmetric), or null; neither: no relation loss
    const float* coef;             // [B]: dL/dmse_b * 2 / S^2
    float *dx, *dg;                // [B,S,C] fp32
    int S, ldx, ldg, ldd;
};

template <class DT, int C>
__global__ void __launch_bounds__(256) trp_backward_kernel(const TrpBwdArgs a) {
    static_assert(C == 32, "the training head is built for 32 channels (HRNet-W32)");
    constexpr int KQ = C / 16, PK = C / 8, RSK = PK + 1, NKP = 32 * PK, NGP = C * 4;
    constexpr int KT = 2;                              // tiles of 32 keys per stage (one barrier per stage)
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int n = lane & 31, hh = lane >> 5;
    const int b = blockIdx.y;
    const int q0 = (blockIdx.x * 4 + wave) * 32;
    const size_t boff = (size_t)b * a.S;
    const char* xb = a.x + boff * a.ldx * 2;
    const char* gb = a.g + boff * a.ldg * 2;
    const char* db = a.de + boff * a.ldd * 2;
    typename DT::vec8 qx[KQ], qg[KQ], qd[KQ];          // query-side fragments: lane (n, hh) holds row q0+n, channels 16s+8hh..+8
#pragma unroll
    for (int s = 0; s < KQ; ++s) {
        qx[s] = __builtin_bit_cast(typename DT::vec8, *reinterpret_cast<const uint4*>(xb + ((size_t)(q0 + n) * a.ldx + 16 * s + 8 * hh) * 2));
        qg[s] = __builtin_bit_cast(typename DT::vec8, *reinterpret_cast<const uint4*>(gb + ((size_t)(q0 + n) * a.ldg + 16 * s + 8 * hh) * 2));
        qd[s] = __builtin_bit_cast(typename DT::vec8, *reinterpret_cast<const uint4*>(db + ((size_t)(q0 + n) * a.ldd + 16 * s + 8 * hh) * 2));
    }
    f32x16 ydx, ydg;
#pragma unroll
    for (int r = 0; r < 16; ++r) ydx[r] = ydg[r] = 0.f;
    const float pq = a.p ? a.p[boff + q0 + n] : 0.f;
    const float c2 = (a.p || a.t) ? 2.f * a.coef[b] : 0.f;
    // dense target T [B,S,S]: d/dX of sum_ij (R_ij - T_ij)^2 with R symmetric takes (T_ij + T_ji) / 2 -- both entries are read
    // (the reference's p p^T is symmetric, then this is T_ij exactly; any other target gets its true gradient: advisor, round 4)
    const float* const trow = a.t ? a.t + boff * a.S + q0 + n : nullptr;      // + key * S: T[key][query]
    const float* const tcol = a.t ? a.t + (boff + q0 + n) * a.S : nullptr;    // + key:     T[query][key]

    __shared__ uint4 tk[2][KT][3][32 * RSK];          // X, G, dE rows of the key tile
    __shared__ uint4 tg[2][KT][2][C * 5];             // X^T, dE^T of the key tile
    __shared__ float tp[2][KT][32];                   // p of the key tile
    // staging: 256 threads = KT tiles x 128 pieces; every thread moves one 16-byte piece of each of the five operands
    static_assert(NKP == 128 && NGP == 128 && KT == 2, "one piece per thread, operand and stage");
    const int su = threadIdx.x >> 7, sq = threadIdx.x & 127;
    const int krow = sq / PK, kpiece = sq % PK, grow = sq >> 2, gpiece = sq & 3;
    uint4 v0, v1, v2, v3, v4;
    float pv;
    auto fetch = [&](int jstage) {
        const int j0 = jstage + su * 32;
        v0 = *reinterpret_cast<const uint4*>(xb + ((size_t)(j0 + krow) * a.ldx + kpiece * 8) * 2);
        v1 = *reinterpret_cast<const uint4*>(gb + ((size_t)(j0 + krow) * a.ldg + kpiece * 8) * 2);
        v2 = *reinterpret_cast<const uint4*>(db + ((size_t)(j0 + krow) * a.ldd + kpiece * 8) * 2);
        v3 = *reinterpret_cast<const uint4*>(a.xT + (((size_t)b * C + grow) * a.S + j0 + gpiece * 8) * 2);
        v4 = *reinterpret_cast<const uint4*>(a.deT + (((size_t)b * C + grow) * a.S + j0 + gpiece * 8) * 2);
        pv = (a.p && threadIdx.x < 32 * KT) ? a.p[boff + jstage + threadIdx.x] : 0.f;
    };
    auto stash = [&](int buf) {
        tk[buf][su][0][krow * RSK + kpiece] = v0;
        tk[buf][su][1][krow * RSK + kpiece] = v1;
        tk[buf][su][2][krow * RSK + kpiece] = v2;
        tg[buf][su][0][grow * 5 + gpiece] = v3;
        tg[buf][su][1][grow * 5 + gpiece] = v4;
        if (threadIdx.x < 32 * KT) tp[buf][threadIdx.x >> 5][threadIdx.x & 31] = pv;
    };
    auto do_tile = [&](const int j0, const uint4 (&tkb)[3][32 * RSK], const uint4 (&tgb)[2][C * 5], const float (&pk)[32]) {
        f32x16 xa, x1, x2;
#pragma unroll
        for (int r = 0; r < 16; ++r) xa[r] = x1[r] = x2[r] = 0.f;
#pragma unroll
        for (int s = 0; s < KQ; ++s) {
            xa = DT::mfma(__builtin_bit_cast(typename DT::vec8, tkb[0][n * RSK + 2 * s + hh]), qx[s], xa);   // X_key . X_query
            x1 = DT::mfma(__builtin_bit_cast(typename DT::vec8, tkb[1][n * RSK + 2 * s + hh]), qd[s], x1);   // G_key . dE_query
            x2 = DT::mfma(__builtin_bit_cast(typename DT::vec8, tkb[2][n * RSK + 2 * s + hh]), qg[s], x2);   // dE_key . G_query
        }
        // element r of this lane: key = (r&3) + 8*(r>>2) + 4*hh of the tile, query = q0 + n
        float pr[16], mr[16];
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const float R = sigmoid_f32(xa[r]);
            const int kk = (r & 3) + 8 * (r >> 2) + 4 * hh;
            const float tv = trow ? 0.5f * (trow[(size_t)(j0 + kk) * a.S] + tcol[j0 + kk]) : pq * pk[kk];
            const float t = x1[r] + x2[r] + c2 * (R - tv);
            pr[r] = R;
            mr[r] = R * (1.f - R) * t;
        }
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            typename DT::vec8 pf, mf;
#pragma unroll
            for (int e = 0; e < 8; ++e) { pf[e] = (typename DT::elem)pr[8 * s + e]; mf[e] = (typename DT::elem)mr[8 * s + e]; }
            ydg = DT::mfma(__builtin_bit_cast(typename DT::vec8, tgb[1][n * 5 + 2 * s + hh]), pf, ydg);      // dE^T . P
            ydx = DT::mfma(__builtin_bit_cast(typename DT::vec8, tgb[0][n * 5 + 2 * s + hh]), mf, ydx);      // X^T . M
        }
    };
    const int nstages = a.S / (32 * KT);
    fetch(0);
    stash(0);
    __syncthreads();
    for (int t = 0; t < nstages; ++t) {
        const bool more = t + 1 < nstages;
        if (more) fetch((t + 1) * 32 * KT);
#pragma unroll
        for (int u = 0; u < KT; ++u) do_tile((t * KT + u) * 32, tk[t & 1][u], tg[t & 1][u], tp[t & 1][u]);
        if (more) stash((t + 1) & 1);
        __syncthreads();
    }
    // y[r] = value for query q0+n, channel (r&3) + 8*(r>>2) + 4*hh
    float* ox = a.dx + (boff + q0 + n) * C + 4 * hh;
    float* og = a.dg + (boff + q0 + n) * C + 4 * hh;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        *reinterpret_cast<float4*>(ox + 8 * k) = make_float4(ydx[4 * k], ydx[4 * k + 1], ydx[4 * k + 2], ydx[4 * k + 3]);
        *reinterpret_cast<float4*>(og + 8 * k) = make_float4(ydg[4 * k], ydg[4 * k + 1], ydg[4 * k + 2], ydg[4 * k + 3]);
    }
}

static int trp_backward(const void* x, int32_t ld_x, const void* g, int32_t ld_g, const void* de, int32_t ld_de,
                        const void* xT, const void* deT, const float* p, const float* t, const float* coef, float* dx, float* dg,
                        int32_t batch, int32_t S, int32_t C, int32_t dtype, void* stream);

extern "C" int rsg_trp_backward(const void* x, int32_t ld_x, const void* g, int32_t ld_g, const void* de, int32_t ld_de,
                                const void* xT, const void* deT, const float* p, const float* coef, float* dx, float* dg,
                                int32_t batch, int32_t S, int32_t C, int32_t dtype, void* stream) {
    return trp_backward(x, ld_x, g, ld_g, de, ld_de, xT, deT, p, nullptr, coef, dx, dg, batch, S, C, dtype, stream);
}

extern "C" int rsg_trp_backward_dense(const void* x, int32_t ld_x, const void* g, int32_t ld_g, const void* de, int32_t ld_de,
                                      const void* xT, const void* deT, const float* target, const float* coef, float* dx,
                                      float* dg, int32_t batch, int32_t S, int32_t C, int32_t dtype, void* stream) {
    if (!target || ((uintptr_t)target & 3)) RSG_FAIL("trp_backward_dense: the target [batch,S,S] fp32");
    return trp_backward(x, ld_x, g, ld_g, de, ld_de, xT, deT, nullptr, target, coef, dx, dg, batch, S, C, dtype, stream);
}

static int trp_backward(const void* x, int32_t ld_x, const void* g, int32_t ld_g, const void* de, int32_t ld_de,
                        const void* xT, const void* deT, const float* p, const float* t, const float* coef, float* dx, float* dg,
                        int32_t batch, int32_t S, int32_t C, int32_t dtype, void* stream) {
    if (trp_check(x, ld_x, batch, S, C) || trp_check(g, ld_g, batch, S, C) || trp_check(de, ld_de, batch, S, C)) return 1;
    if (C != 32) RSG_FAIL("trp_backward: C=%d (the training head is built for 32 channels)", C);
    if (!xT || !deT || !dx || !dg || ((p || t) && !coef)) RSG_FAIL("trp_backward: null pointer");
    if ((((uintptr_t)xT | (uintptr_t)deT | (uintptr_t)dx | (uintptr_t)dg) & 15)) RSG_FAIL("trp_backward: pointers must be 16-byte aligned");
    TrpBwdArgs a;
    a.x = (const char*)x; a.g = (const char*)g; a.de = (const char*)de; a.xT = (const char*)xT; a.deT = (const char*)deT;
    a.p = p; a.t = t; a.coef = coef; a.dx = dx; a.dg = dg; a.S = S; a.ldx = ld_x; a.ldg = ld_g; a.ldd = ld_de;
    dim3 grid(S / 128, batch);
    if (dtype == RSG_BF16) RSG_LAUNCH((trp_backward_kernel<BF16, 32>), grid, dim3(256), 0, (hipStream_t)stream, a);
    else RSG_LAUNCH((trp_backward_kernel<F16, 32>), grid, dim3(256), 0, (hipStream_t)stream, a);
    RSG_CHECK_LAUNCH("trp_backward");
    return 0;
}

// ---- g projection, transposed + key-permuted (association.py:288-289) -------------------------------
template <class DT, int C>
__global__ void __launch_bounds__(256) trp_project_g_kernel(const char* __restrict__ x, int ldx,
                                                           const float* __restrict__ w,
                                                           const float* __restrict__ bias,
                                                           char* __restrict__ gT, char* __restrict__ grows, int S, long total) {
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
        float acc = bias[co];                              // uniform addresses: SGPR operands (see trp_w_project_kernel)
#pragma unroll
        for (int ci = 0; ci < C; ++ci) acc = fmaf(w[co * C + ci], xv[ci], acc);
        go[(size_t)co * S] = (typename DT::elem)acc;
        if (grows) reinterpret_cast<typename DT::elem*>(grows)[i * C + co] = (typename DT::elem)acc;    // rows [B,S,C] (training)
    }
}

extern "C" int rsg_trp_project_g(const void* x, int32_t ld_x, const float* w, const float* bias, void* gT_hi,
                                 void* g_rows, int32_t batch, int32_t S, int32_t C, int32_t dtype,
                                 void* stream) {
    if (trp_check(x, ld_x, batch, S, C)) return 1;
    if (!w || !bias || !gT_hi) RSG_FAIL("trp_project_g: null pointer");
    if ((uintptr_t)g_rows & 15) RSG_FAIL("trp_project_g: g_rows alignment");
    const long total = (long)batch * S;
    const int grid = (int)((total + 255) / 256);
#define RSG_PG(DT_, C_) RSG_LAUNCH((trp_project_g_kernel<DT_, C_>), dim3(grid), dim3(256), 0, (hipStream_t)stream, \
                                   (const char*)x, ld_x, w, bias, (char*)gT_hi, (char*)g_rows, S, total)
    if (dtype == RSG_BF16) { if (C == 32) RSG_PG(BF16, 32); else RSG_PG(BF16, 48); }
    else { if (C == 32) RSG_PG(F16, 32); else RSG_PG(F16, 48); }
#undef RSG_PG
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

// IN = float: e is TRP's fp32 E [B,S,C].  IN = a 16-bit storage type: e is a 16-bit NHWC map with channel stride e_ld
// (RELATION_SUB_SAMPLE: the output of the up-sampling deconv, association.py:254-268).
// One position's row of E as fp32 (shared by the two passes of the tail)
template <int C, class IN>
__device__ __forceinline__ void trp_load_row(const void* e_, int e_ld, size_t pos, float (&xv)[C]) {
    if constexpr (sizeof(IN) == 4) {
        const float* e = reinterpret_cast<const float*>(e_) + pos * C;
#pragma unroll
        for (int k = 0; k < C / 4; ++k) {
            const float4 v = *reinterpret_cast<const float4*>(e + 4 * k);
            xv[4 * k] = v.x; xv[4 * k + 1] = v.y; xv[4 * k + 2] = v.z; xv[4 * k + 3] = v.w;
        }
    } else {
        const char* e = reinterpret_cast<const char*>(e_) + pos * e_ld * 2;
#pragma unroll
        for (int k = 0; k < C / 8; ++k) {
            const typename IN::vec8 v = __builtin_bit_cast(typename IN::vec8, *reinterpret_cast<const uint4*>(e + 16 * k));
#pragma unroll
            for (int j = 0; j < 8; ++j) xv[8 * k + j] = (float)v[j];
        }
    }
}

// y == nullptr (inference, rsg_trp_tail_ex keep_y = 0): only the GroupNorm partials leave the kernel; the second pass
// recomputes W e + b (1 024 - 2 304 multiply-adds per position) instead of reading 4 C bytes per position back
template <int C, class IN>
__global__ void __launch_bounds__(256) trp_w_project_kernel(const void* __restrict__ e_, int e_ld, const float* __restrict__ w,
                                                           const float* __restrict__ bias, float* __restrict__ y,
                                                           float* __restrict__ part, int S) {
    constexpr int G = 8, CG = C / G;
    __shared__ float red[4];
    const int b = blockIdx.y, s = blockIdx.x * 256 + threadIdx.x;
    const size_t row = ((size_t)b * S + s) * C;
    float xv[C], yv[C];
    trp_load_row<C, IN>(e_, e_ld, (size_t)b * S + s, xv);
#pragma unroll
    for (int co = 0; co < C; ++co) {
        // W and its bias are read through the scalar cache into SGPRs (uniform addresses).  They used to be staged in LDS and
        // read back as broadcast ds_reads feeding v_pk_fma_f32: beside another stream's kernels that form dropped single FMA
        // terms for 16 lanes at a time (profiles/r03n_trp_projection_under_contention.txt)
        float acc = bias[co];
#pragma unroll
        for (int ci = 0; ci < C; ++ci) acc = fmaf(w[co * C + ci], xv[ci], acc);
        yv[co] = acc;
    }
    if (y) {
#pragma unroll
        for (int k = 0; k < C / 4; ++k)
            *reinterpret_cast<float4*>(y + row + 4 * k) = make_float4(yv[4 * k], yv[4 * k + 1], yv[4 * k + 2], yv[4 * k + 3]);
    }
    float* po = part + (((size_t)b * gridDim.x + blockIdx.x) * G) * 2;
#pragma unroll
    for (int g = 0; g < G; ++g) {
        float sum = 0.f;
#pragma unroll
        for (int c = 0; c < CG; ++c) sum += yv[CG * g + c];
        const float mean = block_sum256(sum, red) * (1.f / (256 * CG));
        float m2 = 0.f;
#pragma unroll
        for (int c = 0; c < CG; ++c) m2 += (yv[CG * g + c] - mean) * (yv[CG * g + c] - mean);
        m2 = block_sum256(m2, red);
        if (threadIdx.x == 0) { po[2 * g] = mean; po[2 * g + 1] = m2; }
    }
}

// IN = void: y is read from the workspace (training keeps it for the backward).  Otherwise (inference): y is recomputed from
// E (IN = float or the 16-bit storage type, as in trp_w_project_kernel) with the project kernel's own expression.
template <class DT, int C, class IN = void>
__global__ void __launch_bounds__(256) trp_gn_apply_kernel(const float* __restrict__ y, const float* __restrict__ part,
                                                          const float* __restrict__ gamma,
                                                          const float* __restrict__ beta, char* __restrict__ out,
                                                          int out_ld, int S, const void* __restrict__ e_ = nullptr, int e_ld = 0,
                                                          const float* __restrict__ w = nullptr, const float* __restrict__ bias = nullptr) {
    constexpr int G = 8, CG = C / G;
    // every thread merges the crop's partials itself (uniform addresses: scalar loads, the same operations in the same order
    // in every thread); no LDS (see trp_w_project_kernel)
    float scale[C], shift[C];
    const int b = blockIdx.y, nb = gridDim.x;
#pragma unroll
    for (int g = 0; g < G; ++g) {
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
    const int s = blockIdx.x * 256 + threadIdx.x;
    typename DT::elem* orow = reinterpret_cast<typename DT::elem*>(out) + ((size_t)b * S + s) * out_ld;
    float xv[std::is_void<IN>::value ? 1 : C];
    if constexpr (!std::is_void<IN>::value) trp_load_row<C, IN>(e_, e_ld, (size_t)b * S + s, xv);
#pragma unroll
    for (int k = 0; k < C / 8; ++k) {
        float vv[8];
        if constexpr (std::is_void<IN>::value) {
            const float* yr = y + ((size_t)b * S + s) * C;
            const float4 v0 = *reinterpret_cast<const float4*>(yr + 8 * k);
            const float4 v1 = *reinterpret_cast<const float4*>(yr + 8 * k + 4);
            vv[0] = v0.x; vv[1] = v0.y; vv[2] = v0.z; vv[3] = v0.w; vv[4] = v1.x; vv[5] = v1.y; vv[6] = v1.z; vv[7] = v1.w;
        } else {
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const int co = 8 * k + j;
                float acc = bias[co];                    // trp_w_project_kernel's expression: the same bits
#pragma unroll
                for (int ci = 0; ci < C; ++ci) acc = fmaf(w[co * C + ci], xv[ci], acc);
                vv[j] = acc;
            }
        }
        typename DT::vec8 o;
#pragma unroll
        for (int j = 0; j < 8; ++j) o[j] = (typename DT::elem)(vv[j] * scale[8 * k + j] + shift[8 * k + j]);
        *reinterpret_cast<uint4*>(orow + 8 * k) = __builtin_bit_cast(uint4, o);
    }
}

extern "C" size_t rsg_trp_tail_workspace_bytes(int32_t batch, int32_t S, int32_t C) {
    return ((size_t)batch * S * C + (size_t)batch * (S / 256 + 1) * 8 * 2) * sizeof(float);
}

static int trp_tail_impl(const void* e, int e16_ld, const float* w, const float* bias, const float* gn_gamma,
                         const float* gn_beta, void* out, int32_t out_ld, int32_t batch, int32_t S,
                         int32_t C, int32_t dtype, void* workspace, void* stream, bool keep_y = true) {
    if (!e || !w || !bias || !gn_gamma || !gn_beta || !out || !workspace) RSG_FAIL("trp_tail: null pointer");
    if (C != 32 && C != 48) RSG_FAIL("trp_tail: C=%d (32 and 48 are built)", C);
    if (S % 256) RSG_FAIL("trp_tail: S=%d must be a multiple of 256", S);
    if (out_ld < C || out_ld % 8 || ((uintptr_t)out & 15)) RSG_FAIL("trp_tail: out_ld=%d / alignment", out_ld);
    if (e16_ld && (e16_ld < C || e16_ld % 8 || ((uintptr_t)e & 15))) RSG_FAIL("trp_tail: e_ld=%d / alignment", e16_ld);
    float* y = (float*)workspace;
    float* part = y + (size_t)batch * S * C;
    dim3 grid(S / 256, batch);
    float* const yw = keep_y ? y : nullptr;
#define RSG_WP(C_, IN_) RSG_LAUNCH((trp_w_project_kernel<C_, IN_>), grid, dim3(256), 0, (hipStream_t)stream, e, e16_ld, w, bias, yw, part, S)
    if (!e16_ld) { if (C == 32) RSG_WP(32, float); else RSG_WP(48, float); }
    else if (dtype == RSG_BF16) { if (C == 32) RSG_WP(32, BF16); else RSG_WP(48, BF16); }
    else { if (C == 32) RSG_WP(32, F16); else RSG_WP(48, F16); }
#undef RSG_WP
    RSG_CHECK_LAUNCH("trp_tail(project)");
#define RSG_GN(DT_, C_) RSG_LAUNCH((trp_gn_apply_kernel<DT_, C_>), grid, dim3(256), 0, (hipStream_t)stream, \
                                   (const float*)y, (const float*)part, gn_gamma, gn_beta, (char*)out, out_ld, S, (const void*)nullptr, 0, \
                                   (const float*)nullptr, (const float*)nullptr)
#define RSG_GNR(DT_, C_, IN_) RSG_LAUNCH((trp_gn_apply_kernel<DT_, C_, IN_>), grid, dim3(256), 0, (hipStream_t)stream, \
                                        (const float*)nullptr, (const float*)part, gn_gamma, gn_beta, (char*)out, out_ld, S, e, e16_ld, w, bias)
    if (keep_y) {
        if (dtype == RSG_BF16) { if (C == 32) RSG_GN(BF16, 32); else RSG_GN(BF16, 48); }
        else { if (C == 32) RSG_GN(F16, 32); else RSG_GN(F16, 48); }
    } else if (!e16_ld) {
        if (dtype == RSG_BF16) { if (C == 32) RSG_GNR(BF16, 32, float); else RSG_GNR(BF16, 48, float); }
        else { if (C == 32) RSG_GNR(F16, 32, float); else RSG_GNR(F16, 48, float); }
    } else {
        if (dtype == RSG_BF16) { if (C == 32) RSG_GNR(BF16, 32, BF16); else RSG_GNR(BF16, 48, BF16); }
        else { if (C == 32) RSG_GNR(F16, 32, F16); else RSG_GNR(F16, 48, F16); }
    }
#undef RSG_GNR
#undef RSG_GN
    RSG_CHECK_LAUNCH("trp_tail(groupnorm)");
    return 0;
}

// keep_y = 0 (inference): the fp32 projection is not written to the workspace (nothing reads it: rsg_trp_tail_backward is
// training's); the GroupNorm pass recomputes it.  e16_ld = 0: e is fp32 [batch,S,C]; else the 16-bit map of rsg_trp_tail_in16.
extern "C" int rsg_trp_tail_ex(const void* e, int32_t e16_ld, const float* w, const float* bias, const float* gn_gamma,
                               const float* gn_beta, void* out, int32_t out_ld, int32_t batch, int32_t S, int32_t C,
                               int32_t dtype, int32_t keep_y, void* workspace, void* stream) {
    if (e16_ld && e16_ld < 8) RSG_FAIL("trp_tail_ex: e_ld=%d", e16_ld);
    return trp_tail_impl(e, e16_ld, w, bias, gn_gamma, gn_beta, out, out_ld, batch, S, C, dtype, workspace, stream, keep_y != 0);
}

extern "C" int rsg_trp_tail(const float* e, const float* w, const float* bias, const float* gn_gamma,
                            const float* gn_beta, void* out, int32_t out_ld, int32_t batch, int32_t S,
                            int32_t C, int32_t dtype, void* workspace, void* stream) {
    return trp_tail_impl(e, 0, w, bias, gn_gamma, gn_beta, out, out_ld, batch, S, C, dtype, workspace, stream);
}

extern "C" int rsg_trp_tail_in16(const void* e16, int32_t e_ld, const float* w, const float* bias, const float* gn_gamma,
                                 const float* gn_beta, void* out, int32_t out_ld, int32_t batch, int32_t S,
                                 int32_t C, int32_t dtype, void* workspace, void* stream) {
    if (e_ld < 8) RSG_FAIL("trp_tail_in16: e_ld=%d", e_ld);
    return trp_tail_impl(e16, e_ld, w, bias, gn_gamma, gn_beta, out, out_ld, batch, S, C, dtype, workspace, stream);
}

// ---- backward of W (training; autograd of association.py:298-300) ------------------------------------------------------
// out = gamma * yhat + beta, yhat = (y - mean_g) * rstd_g over the S x C/8 values of a crop's channel group, y = W e + b.
//   dgamma_c = sum dout * yhat, dbeta_c = sum dout;  dyhat = dout * gamma;
//   dy = rstd_g * (dyhat - mean_g(dyhat) - yhat * mean_g(dyhat * yhat));  de = W^T dy  (dW = dy^T e and db = sum dy are two
//   small GEMM-shaped sums over the B*S rows, left to the caller's BLAS).
// Launch 1: per workgroup of 256 positions the column sums of [dout*yhat | dout | per-group dyhat, dyhat*yhat] through an LDS
// tile (fixed order).  Launch 2: merges a crop's group sums over its workgroups (fixed order), writes dy and de.
// y and the forward's (mean, M2) partials are read from the forward's workspace (rsg_trp_tail keeps them there).
template <int C>
__device__ __forceinline__ void gn_group_stats(const float* part, int b, int nb, float* mean_s, float* rstd_s) {
    constexpr int G = 8, CG = C / G;
    if (threadIdx.x < G) {
        const int g = threadIdx.x;
        const float* pi = part + ((size_t)b * nb * G + g) * 2;
        const float nk = 256.f * CG;
        float mean = 0.f;
        for (int k = 0; k < nb; ++k) mean += pi[(size_t)k * G * 2];
        mean /= (float)nb;
        float m2 = 0.f;
        for (int k = 0; k < nb; ++k) {
            const float d = pi[(size_t)k * G * 2] - mean;
            m2 += pi[(size_t)k * G * 2 + 1] + nk * d * d;
        }
        mean_s[g] = mean;
        rstd_s[g] = rsqrtf(m2 / (nk * (float)nb) + 1e-5f);           // the expressions of trp_gn_apply_kernel
    }
}

template <class DT, int C>
__global__ void __launch_bounds__(256) trp_tail_bwd_sums_kernel(const char* __restrict__ dout, int ld, const float* __restrict__ y,
                                                               const float* __restrict__ part, const float* __restrict__ gamma,
                                                               float* __restrict__ pc, float* __restrict__ pg, int S) {
    constexpr int G = 8, CG = C / G, NC = 2 * C + 2 * G, PITCH = NC + 1;
    extern __shared__ float tile[];                                   // [256][PITCH]
    __shared__ float mean_s[G], rstd_s[G];
    const int b = blockIdx.y, nb = gridDim.x;
    gn_group_stats<C>(part, b, nb, mean_s, rstd_s);
    __syncthreads();
    const int s = blockIdx.x * 256 + threadIdx.x;
    const float* yr = y + ((size_t)b * S + s) * C;
    const char* dr = dout + ((size_t)b * S + s) * ld * 2;
    float* tr = tile + threadIdx.x * PITCH;
#pragma unroll
    for (int g = 0; g < G; ++g) {
        float a1 = 0.f, a2 = 0.f;
#pragma unroll
        for (int c = 0; c < CG; ++c) {
            const int ch = g * CG + c;
            const float yh = (yr[ch] - mean_s[g]) * rstd_s[g];
            const float d = (float)reinterpret_cast<const typename DT::elem*>(dr)[ch];
            tr[ch] = d * yh;
            tr[C + ch] = d;
            const float dyh = d * gamma[ch];
            a1 += dyh;
            a2 += dyh * yh;
        }
        tr[2 * C + 2 * g] = a1;
        tr[2 * C + 2 * g + 1] = a2;
    }
    __syncthreads();
    if (threadIdx.x < NC) {
        float acc = 0.f;
        for (int r = 0; r < 256; ++r) acc += tile[r * PITCH + threadIdx.x];
        const size_t blk = (size_t)b * nb + blockIdx.x;
        if (threadIdx.x < 2 * C) pc[blk * 2 * C + threadIdx.x] = acc;
        else pg[blk * 2 * G + (threadIdx.x - 2 * C)] = acc;
    }
}

template <class DT, int C>
__global__ void __launch_bounds__(256) trp_tail_bwd_apply_kernel(const char* __restrict__ dout, int ld, const float* __restrict__ y,
                                                                const float* __restrict__ part, const float* __restrict__ gamma,
                                                                const float* __restrict__ w, const float* __restrict__ pg,
                                                                float* __restrict__ dy, float* __restrict__ de, int S) {
    constexpr int G = 8, CG = C / G;
    __shared__ float mean_s[G], rstd_s[G], m1_s[G], m2_s[G];
    const int b = blockIdx.y, nb = gridDim.x;
    gn_group_stats<C>(part, b, nb, mean_s, rstd_s);
    if (threadIdx.x < G) {
        float a1 = 0.f, a2 = 0.f;
        for (int k = 0; k < nb; ++k) {
            a1 += pg[((size_t)b * nb + k) * 2 * G + 2 * threadIdx.x];
            a2 += pg[((size_t)b * nb + k) * 2 * G + 2 * threadIdx.x + 1];
        }
        const float inv_n = 1.f / ((float)S * CG);
        m1_s[threadIdx.x] = a1 * inv_n;
        m2_s[threadIdx.x] = a2 * inv_n;
    }
    __syncthreads();
    const int s = blockIdx.x * 256 + threadIdx.x;
    const size_t row = ((size_t)b * S + s) * C;
    const char* dr = dout + ((size_t)b * S + s) * ld * 2;
    float dyv[C], dev[C];
#pragma unroll
    for (int ch = 0; ch < C; ++ch) {
        const int g = ch / CG;
        const float yh = (y[row + ch] - mean_s[g]) * rstd_s[g];
        const float dyh = (float)reinterpret_cast<const typename DT::elem*>(dr)[ch] * gamma[ch];
        dyv[ch] = rstd_s[g] * (dyh - m1_s[g] - yh * m2_s[g]);
        dev[ch] = 0.f;
    }
#pragma unroll
    for (int co = 0; co < C; ++co)
#pragma unroll
        for (int ci = 0; ci < C; ++ci) dev[ci] = fmaf(w[co * C + ci], dyv[co], dev[ci]);     // SGPR operands (see trp_w_project_kernel)
#pragma unroll
    for (int k = 0; k < C / 4; ++k) {
        *reinterpret_cast<float4*>(dy + row + 4 * k) = make_float4(dyv[4 * k], dyv[4 * k + 1], dyv[4 * k + 2], dyv[4 * k + 3]);
        *reinterpret_cast<float4*>(de + row + 4 * k) = make_float4(dev[4 * k], dev[4 * k + 1], dev[4 * k + 2], dev[4 * k + 3]);
    }
}

extern "C" size_t rsg_trp_tail_backward_partial_floats(int32_t batch, int32_t S, int32_t C) {
    return (size_t)batch * (S / 256) * (2 * C + 16);
}

extern "C" int rsg_trp_tail_backward(const void* dout, int32_t ld, const void* fwd_workspace, const float* w, const float* gn_gamma,
                                     float* dy, float* de, float* partials, int32_t batch, int32_t S, int32_t C, int32_t dtype,
                                     void* stream) {
    if (!dout || !fwd_workspace || !w || !gn_gamma || !dy || !de || !partials) RSG_FAIL("trp_tail_backward: null pointer");
    if (C != 32) RSG_FAIL("trp_tail_backward: C=%d (the training head is built for 32 channels)", C);
    if (S % 256 || S < 256 || batch < 1) RSG_FAIL("trp_tail_backward: S=%d must be a multiple of 256", S);
    if (ld < C || ld % 8 || ((uintptr_t)dout & 15) || ((uintptr_t)dy & 15) || ((uintptr_t)de & 15)) RSG_FAIL("trp_tail_backward: ld=%d / alignment", ld);
    const float* y = (const float*)fwd_workspace;
    const float* part = y + (size_t)batch * S * C;
    float* pc = partials;                                               // [batch][S/256][2C]: sums of dout*yhat, then of dout
    float* pg = partials + (size_t)batch * (S / 256) * 2 * C;           // [batch][S/256][8][2]
    dim3 grid(S / 256, batch);
    const int lds = 256 * (2 * C + 16 + 1) * (int)sizeof(float);
    const void* sums_fn = dtype == RSG_BF16 ? (const void*)trp_tail_bwd_sums_kernel<BF16, 32> : (const void*)trp_tail_bwd_sums_kernel<F16, 32>;
    if (hipFuncSetAttribute(sums_fn, hipFuncAttributeMaxDynamicSharedMemorySize, lds) != hipSuccess)
        RSG_FAIL("trp_tail_backward: cannot reserve %d B of LDS", lds);
    if (dtype == RSG_BF16) {
        RSG_LAUNCH((trp_tail_bwd_sums_kernel<BF16, 32>), grid, dim3(256), lds, (hipStream_t)stream, (const char*)dout, ld, y, part, gn_gamma, pc, pg, S);
        RSG_CHECK_LAUNCH("trp_tail_backward(sums)");
        RSG_LAUNCH((trp_tail_bwd_apply_kernel<BF16, 32>), grid, dim3(256), 0, (hipStream_t)stream, (const char*)dout, ld, y, part, gn_gamma, w, (const float*)pg, dy, de, S);
    } else {
        RSG_LAUNCH((trp_tail_bwd_sums_kernel<F16, 32>), grid, dim3(256), lds, (hipStream_t)stream, (const char*)dout, ld, y, part, gn_gamma, pc, pg, S);
        RSG_CHECK_LAUNCH("trp_tail_backward(sums)");
        RSG_LAUNCH((trp_tail_bwd_apply_kernel<F16, 32>), grid, dim3(256), 0, (hipStream_t)stream, (const char*)dout, ld, y, part, gn_gamma, w, (const float*)pg, dy, de, S);
    }
    RSG_CHECK_LAUNCH("trp_tail_backward(apply)");
    return 0;
}

// ---- small fp32 row products of the training head (autograd of the TRP's 1x1 projections, association.py:288-289,298) ----
// rows_project:  out[n, j] (+)= sum_k a[n, k] m[k, j]      (data gradient of a 1x1 projection: dx += dg Wg)
// rows_gram:     gram[i, j] = sum_n a[n, i] b[n, j],  colsum[i] = sum_n a[n, i]   (its weight and bias gradients)
// One thread per row / fixed-order two-stage sums: no atomics, identical bits from run to run.  C = 32 or 48.
template <int C>
__global__ void __launch_bounds__(256) rows_project_kernel(const float* __restrict__ a, const float* __restrict__ m,
                                                          float* __restrict__ out, long rows, int accumulate) {
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i >= rows) return;
    float av[C];
#pragma unroll
    for (int k = 0; k < C / 4; ++k) {
        const float4 v = *reinterpret_cast<const float4*>(a + i * C + 4 * k);
        av[4 * k] = v.x; av[4 * k + 1] = v.y; av[4 * k + 2] = v.z; av[4 * k + 3] = v.w;
    }
#pragma unroll 4
    for (int j4 = 0; j4 < C / 4; ++j4) {
        float4 o = accumulate ? *reinterpret_cast<const float4*>(out + i * C + 4 * j4) : make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
        for (int k = 0; k < C; ++k) {                      // m[k][j]: uniform addresses (scalar loads)
            o.x = fmaf(av[k], m[k * C + 4 * j4], o.x);
            o.y = fmaf(av[k], m[k * C + 4 * j4 + 1], o.y);
            o.z = fmaf(av[k], m[k * C + 4 * j4 + 2], o.z);
            o.w = fmaf(av[k], m[k * C + 4 * j4 + 3], o.w);
        }
        *reinterpret_cast<float4*>(out + i * C + 4 * j4) = o;
    }
}

extern "C" int rsg_rows_project_f32(const float* a, const float* m, float* out, int64_t rows, int32_t C, int32_t accumulate,
                                    void* stream) {
    if (!a || !m || !out || rows < 1) RSG_FAIL("rows_project: bad arguments");
    if (C != 32 && C != 48) RSG_FAIL("rows_project: C=%d (32 and 48 are built)", C);
    if (((uintptr_t)a | (uintptr_t)out) & 15) RSG_FAIL("rows_project: alignment");
    const dim3 grid((unsigned)((rows + 255) / 256));
    if (C == 32) RSG_LAUNCH((rows_project_kernel<32>), grid, dim3(256), 0, (hipStream_t)stream, a, m, out, (long)rows, accumulate);
    else RSG_LAUNCH((rows_project_kernel<48>), grid, dim3(256), 0, (hipStream_t)stream, a, m, out, (long)rows, accumulate);
    RSG_CHECK_LAUNCH("rows_project");
    return 0;
}

#define RSG_GRAM_BLOCKS 256
// block k sums its contiguous share of the rows: thread (i = tid / 8, jg = tid % 8) keeps gram[i][jg * C/8 .. + C/8)
template <int C>
__global__ void __launch_bounds__(256) rows_gram_partial_kernel(const float* __restrict__ a, const float* __restrict__ b,
                                                               float* __restrict__ part, long rows) {
    constexpr int JW = C / 8, TR = 64;                    // columns per thread; rows staged per round
    __shared__ float sa[TR][C + 1], sb[TR][C + 1];
    const int tid = threadIdx.x, i = tid >> 3, jg = tid & 7;
    const long per = (rows + gridDim.x - 1) / gridDim.x;
    const long r0 = blockIdx.x * per, r1 = min(rows, r0 + per);
    float acc[JW], cs = 0.f;
#pragma unroll
    for (int j = 0; j < JW; ++j) acc[j] = 0.f;
    for (long base = r0; base < r1; base += TR) {
        const int nr = (int)min((long)TR, r1 - base);
        for (int e = tid; e < TR * C; e += 256) {
            const int r = e / C, c = e - r * C;
            sa[r][c] = r < nr ? a[(base + r) * C + c] : 0.f;
            sb[r][c] = r < nr ? b[(base + r) * C + c] : 0.f;
        }
        __syncthreads();
        if (i < C) {
#pragma unroll 8
            for (int r = 0; r < TR; ++r) {
                const float av = sa[r][i];
#pragma unroll
                for (int j = 0; j < JW; ++j) acc[j] = fmaf(av, sb[r][jg * JW + j], acc[j]);
                if (jg == 0) cs += av;
            }
        }
        __syncthreads();
    }
    if (i < C) {
        float* o = part + (size_t)blockIdx.x * (C * C + C);
#pragma unroll
        for (int j = 0; j < JW; ++j) o[i * C + jg * JW + j] = acc[j];
        if (jg == 0) o[C * C + i] = cs;
    }
}

__global__ void __launch_bounds__(256) rows_gram_sum_kernel(const float* __restrict__ part, float* __restrict__ gram,
                                                           float* __restrict__ colsum, int C, int nblocks) {
    const int e = blockIdx.x * 256 + threadIdx.x, n = C * C + C;
    if (e >= n) return;
    double s = 0.0;
    for (int k = 0; k < nblocks; ++k) s += (double)part[(size_t)k * n + e];     // fixed order
    if (e < C * C) gram[e] = (float)s;
    else if (colsum) colsum[e - C * C] = (float)s;
}

extern "C" size_t rsg_rows_gram_workspace_bytes(int32_t C) { return (size_t)RSG_GRAM_BLOCKS * (C * C + C) * sizeof(float); }

extern "C" int rsg_rows_gram_f32(const float* a, const float* b, float* gram, float* colsum, int64_t rows, int32_t C,
                                 void* workspace, void* stream) {
    if (!a || !b || !gram || !workspace || rows < 1) RSG_FAIL("rows_gram: bad arguments");
    if (C != 32) RSG_FAIL("rows_gram: C=%d (32 is built: 256 threads = 32 rows x 8 column groups)", C);
    const int nb = (int)std::min<int64_t>(RSG_GRAM_BLOCKS, (rows + 63) / 64);
    RSG_LAUNCH((rows_gram_partial_kernel<32>), dim3(nb), dim3(256), 0, (hipStream_t)stream, a, b, (float*)workspace, (long)rows);
    RSG_CHECK_LAUNCH("rows_gram partial");
    RSG_LAUNCH(rows_gram_sum_kernel, dim3((C * C + C + 255) / 256), dim3(256), 0, (hipStream_t)stream, (const float*)workspace, gram,
               colsum, (int)C, nb);
    RSG_CHECK_LAUNCH("rows_gram sum");
    return 0;
}
