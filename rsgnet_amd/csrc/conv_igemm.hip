// Implicit-GEMM convolution for gfx950 (MI355X): NHWC 16-bit activations, fp32 accumulate on
// v_mfma_f32_32x32x16_{f16,bf16}, folded-BatchNorm bias + residual + ReLU fused in the epilogue.
//
// Replaces the reference's nn.Conv2d/BatchNorm2d/ReLU/ConvTranspose2d chains
// (/root/reference/lib/models/pose_hrnet.py:23-101,203-243,336-375; pose_rsgnet.py:666-750).
//
// Mapping onto the MFMA (D[32x32] += A[32x16] * B[16x32], one wave):
//   A rows   = 32 output channels (weights, BN scale folded in), read straight from global memory in
//              a fragment-packed order (one coalesced 1 KiB load per wave and k-step, L2 resident)
//   B cols   = 32 output pixels, gathered with ds_read_b128 from an LDS image of the input tile+halo
//   k        = 16 input channels of one filter tap
//   D        = lane holds ONE pixel and 16 output channels; the packer permutes the weight rows so
//              those 16 channels are contiguous -> the epilogue does two 16-byte stores per pixel.
// LDS image : [tile row][tile col][CINB channels + 8 pad] 16-bit; the odd 16-byte-slot pixel stride
//             makes the 16-lane ds_read_b128 groups conflict-free, the row pitch is a multiple of 256 B.
// Work split: a workgroup = WM x WN waves; WM splits output channels (32 each), WN splits the
//             tile's pixels (NT groups of 32 per wave).  Channels are staged CINB at a time.
#include "common.h"

thread_local char g_rsg_err[512] = "";
extern "C" const char* rsg_last_error(void) { return g_rsg_err; }
extern "C" int rsg_version(void) { return 1; }

struct ConvArgs {
    const char* in;
    const uint4* wpk;
    const float* bias;
    const char* res;
    char* out;
    float* out32;
    int B, Hin, Win, ldi;
    int Hout, Wout, Cout, ldo, ldr;
    int SI, oy0, ox0;
    int up, upy, upx, Hp, Wp;
    int relu;
    int TH, TW, TIH, TIW, RP;
    int tiles_x, tiles_y;
    int nchunks, KS;
};

template <class DT, int CINB, int TAPH, int TAPW, int WM, int WN, int NT>
__global__ void __launch_bounds__(64 * WM * WN) conv_igemm_kernel(const ConvArgs p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int NTHR = 64 * WM * WN;
    constexpr int KPT = CINB / 16;       // k-steps per filter tap
    constexpr int NTAPS = TAPH * TAPW;
    constexpr int PS = 2 * CINB + 16;    // LDS bytes per pixel (odd number of 16-B slots)
    constexpr int CPP = CINB / 8;        // 16-byte chunks per pixel
    constexpr int NLD = 8;               // staging loads in flight per thread
    constexpr int KSB = NTAPS * KPT;     // k-steps per channel block
    constexpr int NPF = KSB < 6 ? KSB : 6;   // A-operand prefetch depth

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave / WN, wn = wave % WN;
    const int n = lane & 31, hh = lane >> 5;

    int bx = blockIdx.x;
    const int tix = bx % p.tiles_x;
    bx /= p.tiles_x;
    const int tiy = bx % p.tiles_y;
    const int b = bx / p.tiles_y;
    const int y0 = tiy * p.TH, x0 = tix * p.TW;
    const int ct = blockIdx.y * WM + wm;

    int pixoff[NT];
    int opix[NT];
#pragma unroll
    for (int g = 0; g < NT; ++g) {
        const int q = (wn * NT + g) * 32 + n;
        const int py = q / p.TW, px = q - py * p.TW;
        const bool in_tile = q < p.TH * p.TW;
        const bool valid = in_tile && (y0 + py) < p.Hout && (x0 + px) < p.Wout;
        pixoff[g] = (in_tile ? (py * p.SI) * p.RP + (px * p.SI) * PS : 0) + hh * 16;
        opix[g] = valid ? ((b * p.Hp + (y0 + py) * p.up + p.upy) * p.Wp + (x0 + px) * p.up + p.upx) : -1;
    }

    f32x16 acc[NT];
#pragma unroll
    for (int g = 0; g < NT; ++g)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[g][r] = 0.f;

    const uint4* ap = p.wpk + (size_t)ct * p.KS * 64 + lane;
    const int total = p.TIH * p.TIW * CPP;
    const int iy0 = y0 * p.SI + p.oy0, ix0 = x0 * p.SI + p.ox0;
    const size_t img = (size_t)b * p.Hin * p.Win;

    for (int ch = 0; ch < p.nchunks; ++ch) {
        // first weight fragments of this channel block: issued before the staging so they land under it
        const uint4* apc = ap + (size_t)ch * KSB * 64;
        uint4 aq[NPF];
#pragma unroll
        for (int k = 0; k < NPF; ++k) aq[k] = apc[k * 64];
        // ---- stage CINB channels of the input tile (+halo) into LDS, zero outside the image ------
        for (int base = 0; base < total; base += NTHR * NLD) {
            uint4 r[NLD];
            int lo[NLD];
#pragma unroll
            for (int j = 0; j < NLD; ++j) {
                const int id = base + j * NTHR + tid;
                r[j] = make_uint4(0, 0, 0, 0);
                lo[j] = -1;
                if (id < total) {
                    const int pix = id / CPP, cg = id - pix * CPP;
                    const int ty = pix / p.TIW, tx = pix - ty * p.TIW;
                    const int iy = iy0 + ty, ix = ix0 + tx;
                    lo[j] = ty * p.RP + tx * PS + cg * 16;
                    if (iy >= 0 && iy < p.Hin && ix >= 0 && ix < p.Win) {
                        const size_t e = (img + (size_t)iy * p.Win + ix) * p.ldi + ch * CINB + cg * 8;
                        r[j] = *reinterpret_cast<const uint4*>(p.in + e * 2);
                    }
                }
            }
#pragma unroll
            for (int j = 0; j < NLD; ++j)
                if (lo[j] >= 0) *reinterpret_cast<uint4*>(smem + lo[j]) = r[j];
        }
        __syncthreads();

        // ---- MFMA over the taps of this channel block: one scheduling region per k-step (A loads NPF
        // steps ahead, LDS reads one step ahead), else the compiler sinks the loads next to their uses
        auto boff = [&](int ks) {
            const int t = ks / KPT, c = ks % KPT;
            return (t / TAPW) * p.RP + (t % TAPW) * PS + c * 32;
        };
        uint4 bq[2][NT];
#pragma unroll
        for (int g = 0; g < NT; ++g) bq[0][g] = *reinterpret_cast<const uint4*>(smem + pixoff[g] + boff(0));
#pragma unroll
        for (int ks = 0; ks < KSB; ++ks) {
            const typename DT::vec8 a = __builtin_bit_cast(typename DT::vec8, aq[ks % NPF]);
            if (ks + 1 < KSB) {
#pragma unroll
                for (int g = 0; g < NT; ++g)
                    bq[(ks + 1) & 1][g] = *reinterpret_cast<const uint4*>(smem + pixoff[g] + boff(ks + 1));
            }
            if (ks + NPF < KSB) aq[ks % NPF] = apc[(ks + NPF) * 64];
#pragma unroll
            for (int g = 0; g < NT; ++g)
                acc[g] = DT::mfma(a, __builtin_bit_cast(typename DT::vec8, bq[ks & 1][g]), acc[g]);
            __builtin_amdgcn_sched_barrier(0);
        }
        if (ch + 1 < p.nchunks) __syncthreads();
    }

    // ---- epilogue: + bias (+ residual) (ReLU) -> 16-bit NHWC and/or fp32 NCHW ---------------------------
    const int cbase = ct * 32 + hh * 16;
    float bias[16];
#pragma unroll
    for (int r4 = 0; r4 < 4; ++r4) {
        const float4 bv = *reinterpret_cast<const float4*>(p.bias + cbase + r4 * 4);
        bias[r4 * 4 + 0] = bv.x;
        bias[r4 * 4 + 1] = bv.y;
        bias[r4 * 4 + 2] = bv.z;
        bias[r4 * 4 + 3] = bv.w;
    }
#pragma unroll
    for (int g = 0; g < NT; ++g) {
        if (opix[g] < 0) continue;
        float v[16];
#pragma unroll
        for (int r = 0; r < 16; ++r) v[r] = acc[g][r] + bias[r];
        if (p.res) {
            const char* rp = p.res + ((size_t)opix[g] * p.ldr + cbase) * 2;
#pragma unroll
            for (int k = 0; k < 2; ++k) {
                const uint4 rv = *reinterpret_cast<const uint4*>(rp + k * 16);
                const typename DT::vec8 rh = __builtin_bit_cast(typename DT::vec8, rv);
#pragma unroll
                for (int e = 0; e < 8; ++e) v[k * 8 + e] += (float)rh[e];
            }
        }
        if (p.relu) {
#pragma unroll
            for (int r = 0; r < 16; ++r) v[r] = fmaxf(v[r], 0.f);
        }
        if (p.out) {
            char* op = p.out + ((size_t)opix[g] * p.ldo + cbase) * 2;
#pragma unroll
            for (int k = 0; k < 2; ++k) {
                typename DT::vec8 o;
#pragma unroll
                for (int e = 0; e < 8; ++e) o[e] = (typename DT::elem)v[k * 8 + e];
                *reinterpret_cast<uint4*>(op + k * 16) = __builtin_bit_cast(uint4, o);
            }
        }
        if (p.out32) {
            const int plane = p.Hp * p.Wp;
            const int rem = opix[g] - b * plane;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int c = cbase + r;
                if (c < p.Cout) p.out32[((size_t)b * p.Cout + c) * plane + rem] = v[r];
            }
        }
    }
}

// ------------------------------------------------------------------------------------------------
// host side: tiling choice, dispatch, weight packing
// ------------------------------------------------------------------------------------------------
struct TilePlan {
    int CINB, WM, WN, NT, TH, TW, TIH, TIW, RP, lds, tiles_x, tiles_y;
};

static int pick_cinb(int cin) {
    if (cin % 64 == 0) return 64;
    if (cin % 32 == 0) return 32;
    if (cin % 16 == 0) return 16;
    return 0;
}

extern "C" int32_t rsg_conv_padded_cin(int32_t cin) {
    if (cin <= 16) return 16;
    if (cin % 16 == 0) return cin;
    return round_up(cin, 32);
}

static void tile_geometry(const rsg_conv_desc* d, TilePlan& t) {
    const int PS = 2 * t.CINB + 16;
    t.TIH = (t.TH - 1) * d->stride + d->kh;
    t.TIW = (t.TW - 1) * d->stride + d->kw;
    t.RP = t.TW >= 16 ? round_up(t.TIW * PS, 256) : t.TIW * PS;
    t.lds = t.TIH * t.RP;
    t.tiles_y = ceil_div(d->out_h, t.TH);
    t.tiles_x = ceil_div(d->out_w, t.TW);
}

static const int LDS_LIMIT = 64 * 1024;

static int choose_tile(const rsg_conv_desc* d, TilePlan& best) {
    const int cinb = pick_cinb(d->in_c);
    if (!cinb) RSG_FAIL("conv: in_c=%d is not a multiple of 16 (pad with rsg_conv_padded_cin)", d->in_c);
    const int cout_pad = round_up(d->out_c, 32);
    if (d->tile_hint) {
        best.CINB = cinb;
        best.WM = d->tile_hint & 0xf;
        best.NT = (d->tile_hint >> 4) & 0xf;
        best.TH = (d->tile_hint >> 8) & 0xff;
        best.TW = (d->tile_hint >> 16) & 0xff;
        best.WN = 4 / best.WM;
        if (best.WM < 1 || best.WM > 4 || best.WM == 3 || cout_pad % (32 * best.WM) || best.TH < 1 ||
            best.TW < 1 || best.TH * best.TW > 32 * best.NT * best.WN)
            RSG_FAIL("conv: bad tile_hint 0x%x", d->tile_hint);
        tile_geometry(d, best);
        if (best.lds > LDS_LIMIT) RSG_FAIL("conv: tile_hint needs %d B of LDS", best.lds);
        return 0;
    }
    int WM = 1;
    if (cout_pad % 128 == 0 && d->out_h * d->out_w <= 64) WM = 4;
    else if (cout_pad % 64 == 0) WM = 2;
    const int WN = 4 / WM;
    static const int nts[3][3] = {{3, 1, 0}, {3, 1, 0}, {2, 0, 0}};   // per WM = 1, 2, 4
    const int* ntl = nts[WM == 1 ? 0 : (WM == 2 ? 1 : 2)];
    static const int tws[] = {0, 96, 64, 48, 32, 24, 16, 12, 8, 6, 4};
    double best_score = -1e30;
    bool found = false;
    for (int i = 0; i < 3 && ntl[i]; ++i) {
        const int NT = ntl[i], cap = 32 * NT * WN;
        for (int tw : tws) {
            TilePlan t;
            t.CINB = cinb; t.WM = WM; t.WN = WN; t.NT = NT;
            t.TW = tw ? tw : d->out_w;
            if (t.TW > d->out_w || t.TW > cap) continue;
            t.TH = cap / t.TW;
            if (t.TH > d->out_h) t.TH = d->out_h;
            if (t.TH < 1) continue;
            tile_geometry(d, t);
            if (t.lds > LDS_LIMIT) {     // shrink the tile height until it fits
                while (t.TH > 1 && t.lds > LDS_LIMIT) { --t.TH; tile_geometry(d, t); }
                if (t.lds > LDS_LIMIT) continue;
            }
            const double useful = (double)d->out_h * d->out_w;
            const double spent = (double)t.tiles_y * t.tiles_x * cap;
            // MFMA utilisation, with a mild preference for less halo re-read and conflict-free widths
            const double halo = (double)(t.TIH * t.TIW) / (t.TH * t.TW * d->stride * d->stride);
            double score = useful / spent - 0.02 * halo + (t.TW >= 16 ? 0.01 : 0.0);
            if (score > best_score) { best_score = score; best = t; found = true; }
        }
    }
    if (!found) RSG_FAIL("conv: no tile fits (cin=%d cout=%d %dx%d k%d s%d)", d->in_c, d->out_c,
                                 d->out_h, d->out_w, d->kh, d->stride);
    return 0;
}

extern "C" int rsg_conv_plan(const rsg_conv_desc* d, int32_t* o) {
    TilePlan t;
    if (choose_tile(d, t)) return 1;
    o[0] = t.CINB; o[1] = t.WM; o[2] = t.WN; o[3] = t.NT; o[4] = t.TH; o[5] = t.TW; o[6] = t.lds;
    o[7] = d->batch * t.tiles_y * t.tiles_x;
    return 0;
}

typedef void (*conv_fn)(const ConvArgs);

template <class DT>
static conv_fn lookup(int cinb, int taps, int wm, int nt) {
#define RSG_CASE(C, TH_, TW_, WM_, WN_, NT_) \
    if (cinb == C && taps == TH_ * TW_ && wm == WM_ && nt == NT_) \
        return conv_igemm_kernel<DT, C, TH_, TW_, WM_, WN_, NT_>;
#define RSG_CFGS(C, TH_, TW_)      \
    RSG_CASE(C, TH_, TW_, 1, 4, 3) \
    RSG_CASE(C, TH_, TW_, 1, 4, 1) \
    RSG_CASE(C, TH_, TW_, 2, 2, 3) \
    RSG_CASE(C, TH_, TW_, 2, 2, 1) \
    RSG_CASE(C, TH_, TW_, 4, 1, 2)
    RSG_CFGS(16, 3, 3)
    RSG_CFGS(32, 3, 3)
    RSG_CFGS(64, 3, 3)
    RSG_CFGS(16, 1, 1)
    RSG_CFGS(32, 1, 1)
    RSG_CFGS(64, 1, 1)
    RSG_CFGS(16, 2, 2)
    RSG_CFGS(32, 2, 2)
#undef RSG_CFGS
#undef RSG_CASE
    return nullptr;
}

extern "C" int rsg_conv_igemm(const rsg_conv_desc* d, void* stream) {
    if (!d || !d->in || !d->w_packed || !d->bias || (!d->out16 && !d->out_nchw_f32))
        RSG_FAIL("conv: null pointer in descriptor");
    if (!((d->kh == 3 && d->kw == 3) || (d->kh == 1 && d->kw == 1) || (d->kh == 2 && d->kw == 2)))
        RSG_FAIL("conv: kernel %dx%d not supported (3x3, 1x1, 2x2)", d->kh, d->kw);
    if (d->stride < 1 || d->stride > 2) RSG_FAIL("conv: stride %d", d->stride);
    if (d->batch < 1 || d->out_h < 1 || d->out_w < 1 || d->out_c < 1) RSG_FAIL("conv: empty output");
    if (d->in_ld < d->in_c || (d->in_ld % 8) || d->in_c != rsg_conv_padded_cin(d->in_c))
        RSG_FAIL("conv: in_c=%d in_ld=%d must be padded/aligned (multiple of 8, in_c padded)", d->in_c, d->in_ld);
    const int cout_pad = round_up(d->out_c, 32);
    if (d->out16 && (d->out_ld < cout_pad || d->out_ld % 8)) RSG_FAIL("conv: out_ld=%d < padded cout %d", d->out_ld, cout_pad);
    if (d->residual && (d->res_ld < cout_pad || d->res_ld % 8)) RSG_FAIL("conv: res_ld=%d", d->res_ld);
    if (d->up < 1 || d->up_y < 0 || d->up_x < 0 || (d->out_h - 1) * d->up + d->up_y >= d->phys_h ||
        (d->out_w - 1) * d->up + d->up_x >= d->phys_w)
        RSG_FAIL("conv: output scatter (up=%d,%d,%d) exceeds %dx%d", d->up, d->up_y, d->up_x, d->phys_h, d->phys_w);
    if (((uintptr_t)d->in | (uintptr_t)d->w_packed | (uintptr_t)d->bias | (uintptr_t)d->residual |
         (uintptr_t)d->out16) & 15)
        RSG_FAIL("conv: pointers must be 16-byte aligned");
    TilePlan t;
    if (choose_tile(d, t)) return 1;
    conv_fn fn = d->dtype == RSG_BF16 ? lookup<BF16>(t.CINB, d->kh * d->kw, t.WM, t.NT)
                                      : lookup<F16>(t.CINB, d->kh * d->kw, t.WM, t.NT);
    if (!fn) RSG_FAIL("conv: no kernel for CINB=%d k=%d WM=%d NT=%d", t.CINB, d->kh, t.WM, t.NT);
    ConvArgs a;
    a.in = (const char*)d->in; a.wpk = (const uint4*)d->w_packed; a.bias = d->bias;
    a.res = (const char*)d->residual; a.out = (char*)d->out16; a.out32 = d->out_nchw_f32;
    a.B = d->batch; a.Hin = d->in_h; a.Win = d->in_w; a.ldi = d->in_ld;
    a.Hout = d->out_h; a.Wout = d->out_w; a.Cout = d->out_c; a.ldo = d->out_ld; a.ldr = d->res_ld;
    a.SI = d->stride; a.oy0 = -d->pad_y; a.ox0 = -d->pad_x;
    a.up = d->up; a.upy = d->up_y; a.upx = d->up_x; a.Hp = d->phys_h; a.Wp = d->phys_w;
    a.relu = d->relu;
    a.TH = t.TH; a.TW = t.TW; a.TIH = t.TIH; a.TIW = t.TIW; a.RP = t.RP;
    a.tiles_x = t.tiles_x; a.tiles_y = t.tiles_y;
    a.nchunks = d->in_c / t.CINB;
    a.KS = a.nchunks * d->kh * d->kw * (t.CINB / 16);
    dim3 grid(d->batch * t.tiles_y * t.tiles_x, cout_pad / (32 * t.WM));
    RSG_LAUNCH(fn, grid, dim3(256), t.lds, (hipStream_t)stream, a);
    RSG_CHECK_LAUNCH("conv_igemm");
    return 0;
}

extern "C" size_t rsg_conv_packed_bytes(int32_t cin, int32_t cout, int32_t kh, int32_t kw) {
    const int cin_pad = rsg_conv_padded_cin(cin);
    return (size_t)round_up(cout, 32) * cin_pad * kh * kw * 2;
}

static inline uint16_t f32_to_16(float v, int dtype) {
    if (dtype == RSG_F16) {
        _Float16 h = (_Float16)v;
        uint16_t u;
        memcpy(&u, &h, 2);
        return u;
    }
    uint32_t u;
    memcpy(&u, &v, 4);
    if ((u & 0x7fffffffu) > 0x7f800000u) return (uint16_t)((u >> 16) | 0x40);   // NaN stays NaN
    return (uint16_t)((u + 0x7fffu + ((u >> 16) & 1u)) >> 16);
}

// Fragment order: [cout tile of 32][k-step][lane 0..63][8 elements]; k-steps run
// (channel block of CINB) -> (tap) -> (16 channels); within a k-step lane (i = lane&31, h = lane>>5)
// holds row i, k = 8h..8h+7.  Row i is output channel 16*((i>>2)&1) + (i&3) + 4*(i>>3) of the tile,
// which makes a lane's 16 accumulator registers 16 consecutive channels.
extern "C" int rsg_pack_conv_weights_host(const float* w, const float* scale, const float* shift,
                                          const float* cbias, int32_t cin, int32_t cout, int32_t kh,
                                          int32_t kw, int32_t dtype, void* wp, float* bias_out) {
    if (!w || !wp || !bias_out) RSG_FAIL("pack: null pointer");
    const int cin_pad = rsg_conv_padded_cin(cin), cout_pad = round_up(cout, 32);
    const int cinb = pick_cinb(cin_pad);
    if (!cinb) RSG_FAIL("pack: cin %d", cin);
    const int ntaps = kh * kw, kpt = cinb / 16, nchunks = cin_pad / cinb;
    const int KS = nchunks * ntaps * kpt;
    uint16_t* o = (uint16_t*)wp;
    for (int ct = 0; ct < cout_pad / 32; ++ct)
        for (int ch = 0; ch < nchunks; ++ch)
            for (int t = 0; t < ntaps; ++t)
                for (int c = 0; c < kpt; ++c) {
                    const int ks = (ch * ntaps + t) * kpt + c;
                    for (int lane = 0; lane < 64; ++lane) {
                        const int i = lane & 31, h = lane >> 5;
                        const int co = ct * 32 + 16 * ((i >> 2) & 1) + (i & 3) + 4 * (i >> 3);
                        for (int j = 0; j < 8; ++j) {
                            const int ci = ch * cinb + c * 16 + h * 8 + j;
                            float v = 0.f;
                            if (co < cout && ci < cin) {
                                v = w[((size_t)co * cin + ci) * ntaps + t];
                                if (scale) v *= scale[co];
                            }
                            o[(((size_t)ct * KS + ks) * 64 + lane) * 8 + j] = f32_to_16(v, dtype);
                        }
                    }
                }
    for (int co = 0; co < cout_pad; ++co) {
        float bsum = 0.f;
        if (co < cout) {
            const float s = scale ? scale[co] : 1.f;
            bsum = (cbias ? cbias[co] * s : 0.f) + (shift ? shift[co] : 0.f);
        }
        bias_out[co] = bsum;
    }
    return 0;
}
