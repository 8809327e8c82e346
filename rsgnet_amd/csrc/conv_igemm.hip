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
#include "conv_body.h"

thread_local char g_rsg_err[512] = "";
extern "C" const char* rsg_last_error(void) { return g_rsg_err; }
extern "C" int rsg_version(void) { return RSG_ABI_VERSION; }


#ifdef RSG_CONV_STAMPS
extern "C" int rsg_conv_set_stamp_buffer_single(void* dev_ptr) {       // this translation unit's copy
    unsigned long long* p = (unsigned long long*)dev_ptr;
    if (hipMemcpyToSymbol(HIP_SYMBOL(g_conv_stamps), &p, sizeof(p)) != hipSuccess) RSG_FAIL("stamps: memcpy failed");
    return 0;
}
#endif

template <class DT, int CINB, int TAPH, int TAPW, int WM, int WN, int NT>
__global__ void __launch_bounds__(64 * WM * WN) conv_igemm_kernel(const ConvArgs p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    RSG_CSTAMP(0);
    // the XCD of a workgroup follows blockIdx.x + blockIdx.y * gridDim.x: with gridDim.x a multiple of 8 (or one row)
    // that is blockIdx.x mod 8, and every output-channel tile of an input tile stays on the input tile's XCD
    const bool swz = gridDim.y == 1 || (gridDim.x & 7) == 0;
    conv_body<DT, CINB, TAPH, TAPW, WM, WN, NT>(p, swz ? rsg_xcd_tile(blockIdx.x, gridDim.x) : (int)blockIdx.x, blockIdx.y, smem);
}

// ------------------------------------------------------------------------------------------------
// host side: tiling choice, dispatch, weight packing
// ------------------------------------------------------------------------------------------------

static int pick_cinb(int cin) {
    if (cin % 64 == 0) return 64;
    if (cin % 32 == 0) return 32;
    if (cin == 48) return 48;       // HRNet-W48's first branch: one staging round and 27 k-steps instead of three rounds of nine
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
    // row pitch == TW*PS (mod 256 B): a 32-pixel group wrapping to the next tile row keeps the odd-slot
    // bank progression (see chain.hip); >= the dense pitch
    const int base = t.TIW * PS, want = (t.TW * PS) % 256;
    t.RP = base + ((want - base) % 256 + 256) % 256;
    if (d->stride == 2 && t.TW % 2 == 0) {
        // de-interleaved columns (conv_body): an output row ends TW slots on, the next one starts 2 rows down
        const int want2 = (t.TW * PS / 2) % 128;
        t.RP = base + ((want2 - base) % 128 + 128) % 128;
    }
    t.lds = t.TIH * t.RP;
    t.tiles_y = ceil_div(d->out_h, t.TH);
    t.tiles_x = ceil_div(d->out_w, t.TW);
}

int g_conv_full_k = 1;                 // rsg_tune("conv_full_k", 0|1)
int g_conv_lds_limit = 52 * 1024;      // rsg_tune("conv_lds_limit", bytes): LDS cap of the tile chooser
int g_conv_spread = 128;               // rsg_tune("conv_spread", n): re-tile a conv whose grid has fewer than n workgroups (0 = off)
#define LDS_LIMIT g_conv_lds_limit

static int choose_tile(const rsg_conv_desc* d, TilePlan& best) {
    int cinb = pick_cinb(d->in_c);
    // 1x1 convs over 256 channels on the large maps (Bottleneck conv1, 100 MB of input at batch 64): the whole K of a
    // 64-pixel tile in ONE staging round (34 KB, 4 workgroups per CU) instead of four 64-channel rounds in sequence:
    // 128 KB in flight per CU instead of 74 KB, which is what an HBM-bound kernel lives on
    if (g_conv_full_k && d->kh == 1 && d->kw == 1 && d->in_c == 256 && d->out_h * d->out_w >= 1024) cinb = 256;
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
    // 128 output channels per workgroup for tiny maps, and for wide outputs (>= 256 channels: measured on
    // layer1's 64->256 1x1 convs, 53 -> 43 us: the input tile is staged half as often)
    int WM0 = 1;
    if (cout_pad % 128 == 0 && (d->out_h * d->out_w <= 64 || cout_pad >= 256)) WM0 = 4;
    else if (cout_pad % 64 == 0) WM0 = 2;
    // the best tile for a given split of the four waves (WM output-channel tiles x WN pixel columns) and the pixel
    // groups per wave allowed; returns false when nothing fits
    auto search = [&](int WM, int max_nt, TilePlan& out) -> bool {
        const int WN = 4 / WM;
        static const int nts[3][3] = {{3, 1, 0}, {3, 1, 0}, {2, 0, 0}};   // per WM = 1, 2, 4
        const int* ntl = nts[WM == 1 ? 0 : (WM == 2 ? 1 : 2)];
        static const int tws[] = {0, 96, 64, 48, 32, 24, 16, 12, 8, 6, 4};
        double best_score = -1e30;
        bool found = false;
        for (int i = 0; i < 3 && ntl[i]; ++i) {
            const int NT = ntl[i], cap = 32 * NT * WN;
            if (NT > max_nt) continue;
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
                if (score > best_score) { best_score = score; out = t; found = true; }
            }
        }
        return found;
    };
    if (!search(WM0, 3, best))
        RSG_FAIL("conv: no tile fits (cin=%d cout=%d %dx%d k%d s%d)", d->in_c, d->out_c, d->out_h, d->out_w, d->kh, d->stride);
    // A grid that leaves most CUs without a workgroup (the low-resolution, wide convs of a training batch: 32 crops of
    // 8x6 pixels x 256 channels are 64 workgroups of 12 us each) is latency, not throughput: its workgroups run their
    // whole K alone on a CU, so halving what one of them owns halves the launch.  Spread such a conv over more, smaller
    // workgroups -- fewer output channels, then fewer pixel groups per wave -- until it has a workgroup for half the CUs
    // (same-visit A/B on the training step, profiles/r04_ab_train_conv_spread.txt: 128 is -1.7 %, 256 and 512 are slower
    // than off; results are bit-identical in every tiling).
    if (g_conv_spread) {
        auto nwg = [&](const TilePlan& t) { return (long)d->batch * t.tiles_y * t.tiles_x * (cout_pad / (32 * t.WM)); };
        const long want = g_conv_spread;
        if (nwg(best) < want) {
            // every split of the waves and both pixel-group limits: the coarsest tiling that reaches `want` workgroups,
            // else the finest there is
            TilePlan pick = best;
            bool reached = false;
            for (int WM = 4; WM >= 1; WM >>= 1) {
                if (cout_pad % (32 * WM)) continue;
                for (int max_nt = 3; max_nt >= 1; max_nt -= 2) {
                    TilePlan t;
                    if (!search(WM, max_nt, t)) continue;
                    const long n = nwg(t);
                    if (n >= want ? (!reached || n < nwg(pick)) : (!reached && n > nwg(pick))) {
                        pick = t;
                        reached = reached || n >= want;
                    }
                }
            }
            best = pick;
        }
    }
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
    RSG_CFGS(48, 3, 3)
    RSG_CFGS(32, 3, 3)
    RSG_CFGS(64, 3, 3)
    RSG_CFGS(16, 1, 1)
    RSG_CFGS(48, 1, 1)
    RSG_CFGS(32, 1, 1)
    RSG_CFGS(64, 1, 1)
    RSG_CFGS(256, 1, 1)
    RSG_CFGS(16, 2, 2)
    RSG_CFGS(48, 2, 2)
    RSG_CFGS(32, 2, 2)
#undef RSG_CFGS
#undef RSG_CASE
    return nullptr;
}

// Validates a descriptor, picks the tiling and fills the kernel arguments.
int rsg_conv_prepare(const rsg_conv_desc* d, ConvArgs& a, TilePlan& t, dim3& grid) {
    a.tso = a.tsr = 32;                 // dense NHWC maps (the pair tails may be told otherwise: RSG_LAYOUT_*)
    if (!d || !d->in || !d->w_packed || !d->bias || (!d->out16 && !d->out_nchw_f32))
        RSG_FAIL("conv: null pointer in descriptor");
    if (d->layout) RSG_FAIL("conv: layout bits 0x%x: channel-tile planes are read / written by rsg_conv_pair* and rsg_conv_dual only", d->layout);
    if (!((d->kh == 3 && d->kw == 3) || (d->kh == 1 && d->kw == 1) || (d->kh == 2 && d->kw == 2)))
        RSG_FAIL("conv: kernel %dx%d not supported (3x3, 1x1, 2x2)", d->kh, d->kw);
    if (d->stride < 1 || d->stride > 2) RSG_FAIL("conv: stride %d", d->stride);
    if (d->batch < 1 || d->out_h < 1 || d->out_w < 1 || d->out_c < 1) RSG_FAIL("conv: empty output");
    if (d->in_ld < d->in_c || (d->in_ld % 8) || d->in_c != rsg_conv_padded_cin(d->in_c))
        RSG_FAIL("conv: in_c=%d in_ld=%d must be padded/aligned (multiple of 8, in_c padded)", d->in_c, d->in_ld);
    const int cout_pad = round_up(d->out_c, 32);
    const int store = d->out_c_store ? d->out_c_store : cout_pad;
    if (store % 8 || store > cout_pad || store < 8) RSG_FAIL("conv: out_c_store=%d", d->out_c_store);
    if (d->out16 && (d->out_ld < store || d->out_ld % 8)) RSG_FAIL("conv: out_ld=%d < stored channels %d", d->out_ld, store);
    if (d->residual && (d->res_ld < cout_pad || d->res_ld % 8)) RSG_FAIL("conv: res_ld=%d", d->res_ld);
    if ((d->residual2 && (d->res2_ld < cout_pad || d->res2_ld % 8 || ((uintptr_t)d->residual2 & 15))) ||
        (d->residual3 && (d->res3_ld < cout_pad || d->res3_ld % 8 || ((uintptr_t)d->residual3 & 15))))
        RSG_FAIL("conv: residual2/3 stride or alignment");
    if ((d->residual2 || d->residual3) && !d->residual) RSG_FAIL("conv: residual2/3 need residual");
    if (d->up < 1 || d->up_y < 0 || d->up_x < 0 || (d->out_h - 1) * d->up + d->up_y >= d->phys_h ||
        (d->out_w - 1) * d->up + d->up_x >= d->phys_w)
        RSG_FAIL("conv: output scatter (up=%d,%d,%d) exceeds %dx%d", d->up, d->up_y, d->up_x, d->phys_h, d->phys_w);
    if (((uintptr_t)d->in | (uintptr_t)d->w_packed | (uintptr_t)d->bias | (uintptr_t)d->residual |
         (uintptr_t)d->out16) & 15)
        RSG_FAIL("conv: pointers must be 16-byte aligned");
    if (choose_tile(d, t)) return 1;
    a.in = (const char*)d->in; a.wpk = (const uint4*)d->w_packed; a.bias = d->bias;
    a.res = (const char*)d->residual; a.out = (char*)d->out16; a.out32 = d->out_nchw_f32;
    a.res2 = (const char*)d->residual2; a.res3 = (const char*)d->residual3; a.ldr2 = d->res2_ld; a.ldr3 = d->res3_ld;
    a.B = d->batch; a.Hin = d->in_h; a.Win = d->in_w; a.ldi = d->in_ld;
    a.Hout = d->out_h; a.Wout = d->out_w; a.Cout = d->out_c; a.ldo = d->out_ld; a.ldr = d->res_ld;
    a.CoutStore = store;
    a.SI = d->stride; a.oy0 = -d->pad_y; a.ox0 = -d->pad_x;
    a.up = d->up; a.upy = d->up_y; a.upx = d->up_x; a.Hp = d->phys_h; a.Wp = d->phys_w;
    // bit t = ReLU on the t-th 32-channel output tile; relu_tile_mask lets one launch mix convs with and
    // without ReLU (fan-out fusion of the HRNet fuse layers)
    a.relu = d->relu_tile_mask ? d->relu_tile_mask : (d->relu ? -1 : 0);
    a.TH = t.TH; a.TW = t.TW; a.TIH = t.TIH; a.TIW = t.TIW; a.RP = t.RP;
    a.tiles_x = t.tiles_x; a.tiles_y = t.tiles_y;
    a.nchunks = d->in_c / t.CINB;
    a.KS = a.nchunks * d->kh * d->kw * (t.CINB / 16);
    a.wpk2 = nullptr; a.bias2 = nullptr; a.C2 = 0;

    a.in0 = nullptr; a.wpk0 = nullptr; a.bias0 = nullptr; a.ldi0 = 0;
    grid = dim3(d->batch * t.tiles_y * t.tiles_x, cout_pad / (32 * t.WM));
    return 0;
}

template <class DT>
__global__ void __launch_bounds__(256) conv_pair_kernel(const ConvArgs p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    RSG_CSTAMP(0);
    conv_body<DT, 64, 3, 3, 2, 2, 3, true>(p, rsg_xcd_tile(blockIdx.x, gridDim.x), 0, smem);
}

// conv2 -> conv3 (+x, ReLU) -> the NEXT Bottleneck's conv1 (1x1 256 -> 64, ReLU), 256-pixel tiles (conv_body NEXT)
template <class DT>
__global__ void __launch_bounds__(256) conv_pair_next_kernel(const ConvNextArgs p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    RSG_CSTAMP(0);
    conv_body<DT, 64, 3, 3, 2, 2, 4, true, 0, true>(p, rsg_xcd_tile(blockIdx.x, gridDim.x), 0, smem);
}

// ... with the tail's K = 128: [conv2 output | block input] (the first Bottleneck's conv3 + projected residual in one conv)
template <class DT>
__global__ void __launch_bounds__(256) conv_pair_next_cat_kernel(const ConvNextArgs p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    RSG_CSTAMP(0);
    conv_body<DT, 64, 3, 3, 2, 2, 4, true, 0, true, true>(p, rsg_xcd_tile(blockIdx.x, gridDim.x), 0, smem);
}

// Whole Bottleneck (pose_hrnet.py:85-101) in one launch: conv1 1x1 (head, into LDS) -> conv2 3x3 -> conv3 1x1 + x.
template <class DT, int HEAD>
__global__ void __launch_bounds__(256) bottleneck_kernel(const ConvArgs p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    conv_body<DT, 64, 3, 3, 2, 2, 3, true, HEAD>(p, rsg_xcd_tile(blockIdx.x, gridDim.x), 0, smem);
}
int g_bottleneck_head = 2;     // rsg_tune("bottleneck_head", 1|2): 2 = two pixel groups in flight
#ifdef RSG_DEBUG
int g_bottleneck_debug = 0;    // rsg_tune("bottleneck_debug", 1): timing experiment, no residual read (WRONG results; diagnostic build only)
#endif

// 3x3 stride-1 conv (C1 = 64 outputs, BatchNorm folded, ReLU) followed by a 1x1 conv (+BN, +residual, ReLU) in one
// launch: `inner` describes the 3x3 conv (its out16/residual are ignored), the other arguments the 1x1 tail.
static int pair_launch(const rsg_conv_desc* inner, const void* head_x, int32_t head_ld, const void* head_w,
                       const float* head_bias, const void* tail_w_packed, const float* tail_bias,
                       int32_t tail_out_c, const void* residual, int32_t res_ld, void* out16, int32_t out_ld,
                       int32_t relu, void* stream, const void* next_w = nullptr, const float* next_bias = nullptr,
                       void* next_out = nullptr, int32_t next_ld = 0, const void* cat_x = nullptr, int32_t cat_ld = 0) {
    const int tile_px = next_w ? 256 : 192;             // pixels per workgroup: 2 x 2 waves x NT (3 or 4) groups of 32
    if (!inner || !tail_w_packed || !tail_bias || !out16) RSG_FAIL("conv_pair: null pointer");
    if (inner->kh != 3 || inner->kw != 3 || inner->stride != 1 || inner->out_c != 64 || inner->in_c % 64 ||
        inner->up != 1 || inner->out_nchw_f32)
        RSG_FAIL("conv_pair: the inner conv must be 3x3 stride 1 with 64 output channels and cin a multiple of 64");
    const bool planes_out = (inner->layout & RSG_LAYOUT_OUT_PLANES) != 0, planes_res = residual && (inner->layout & RSG_LAYOUT_RES_PLANES);
    if (tail_out_c < 32 || tail_out_c % 32 || (planes_out ? out_ld != 32 : (out_ld < tail_out_c || out_ld % 8)) ||
        (residual && (planes_res ? res_ld != 32 : (res_ld < tail_out_c || res_ld % 8))))
        RSG_FAIL("conv_pair: tail_out_c=%d out_ld=%d res_ld=%d (a map in channel-tile planes has a pixel stride of 32)", tail_out_c, out_ld, res_ld);
    if ((planes_out || planes_res) && head_x) RSG_FAIL("conv_pair: the fused-head form takes dense maps only");
    if ((long long)inner->batch * inner->out_h * inner->out_w * tail_out_c * 2 >= (1LL << 31)) RSG_FAIL("conv_pair: map too large");
    if (((uintptr_t)tail_w_packed | (uintptr_t)tail_bias | (uintptr_t)residual | (uintptr_t)out16) & 15)
        RSG_FAIL("conv_pair: pointers must be 16-byte aligned");
    rsg_conv_desc d = *inner;
    d.out16 = out16;                   // placeholders for the shared validation; the tail's values are set below
    d.out_ld = 64; d.residual = nullptr; d.relu = 1; d.relu_tile_mask = 0; d.out_c_store = 0;
    d.layout = 0;                      // the plane bits were consumed above (they describe the tail's output / residual)
    // all 64 inner channels of a pixel in one workgroup: 2 x 2 waves, 3 pixel groups each
    int best_h = 0, best_w = 0;
    {
        double best = -1;
        static const int tws[] = {96, 64, 48, 32, 24, 16, 12, 8};
        for (int tw : tws) {
            if (tw > d.out_w) tw = d.out_w;
            int th = tile_px / tw;
            if (th > d.out_h) th = d.out_h;
            if (th < 1) continue;
            const int tih = th + 2, tiw = tw + 2;
            const int rp = tiw * 144 + ((((tw * 144) % 256) - (tiw * 144)) % 256 + 256) % 256;
            if (tih * rp > g_conv_lds_limit) continue;
            const double use = (double)d.out_h * d.out_w / ((double)ceil_div(d.out_h, th) * ceil_div(d.out_w, tw) * tile_px);
            if (use > best) { best = use; best_h = th; best_w = tw; }
        }
        if (!best_h) RSG_FAIL("conv_pair: no tile fits");
    }
    d.tile_hint = 2 | ((next_w ? 4 : 3) << 4) | (best_h << 8) | (best_w << 16);
    ConvArgs a;
    TilePlan t;
    dim3 grid;
    if (rsg_conv_prepare(&d, a, t, grid)) return 1;
    if (grid.y != 1 || t.CINB != 64) RSG_FAIL("conv_pair: unexpected tiling");
    a.out = (char*)out16; a.ldo = out_ld; a.res = (const char*)residual; a.ldr = res_ld;
    const int plane = inner->batch * inner->out_h * inner->out_w * 32;      // elements of one 32-channel plane
    a.tso = planes_out ? plane : 32;
    a.tsr = planes_res ? plane : 32;
    a.relu = relu ? -1 : 0; a.Cout = tail_out_c; a.CoutStore = tail_out_c;
    a.wpk2 = (const uint4*)tail_w_packed; a.bias2 = tail_bias; a.C2 = tail_out_c;
    if (t.lds < t.TH * t.TW * 144) t.lds = t.TH * t.TW * 144;
    if (next_w) {
        if (tail_out_c != 256 || !next_bias || !next_out || next_ld < 64 || next_ld % 8 ||
            (((uintptr_t)next_w | (uintptr_t)next_bias | (uintptr_t)next_out) & 15))
            RSG_FAIL("conv_pair_next: needs a 256-channel tail, 16-byte aligned next_* pointers and next_ld >= 64");
        ConvNextArgs na;
        static_cast<ConvArgs&>(na) = a;
        na.wpk3 = (const uint4*)next_w; na.bias3 = next_bias; na.out3 = (char*)next_out; na.ldo3 = next_ld;
        na.in2 = nullptr; na.ldi2 = 0;
        if (cat_x) {
            if (residual || cat_ld < 64 || cat_ld % 8 || ((uintptr_t)cat_x & 15) || t.TH * t.TW > 256)
                RSG_FAIL("conv_pair_next: the K-concat form takes no residual; cat_x 16-byte aligned, cat_ld >= 64");
            na.in2 = (const char*)cat_x; na.ldi2 = cat_ld;
            const int lds = 2 * 256 * 144;                  // conv2 image + x image
            if (t.lds < lds) t.lds = lds;
            void (*fn)(const ConvNextArgs) = inner->dtype == RSG_BF16 ? conv_pair_next_cat_kernel<BF16> : conv_pair_next_cat_kernel<F16>;
            if (hipFuncSetAttribute((const void*)fn, hipFuncAttributeMaxDynamicSharedMemorySize, t.lds) != hipSuccess)
                RSG_FAIL("conv_pair_next: cannot reserve %d B of LDS", t.lds);
            RSG_LAUNCH(fn, dim3(grid.x), dim3(256), t.lds, (hipStream_t)stream, na);
        } else if (inner->dtype == RSG_BF16) RSG_LAUNCH(conv_pair_next_kernel<BF16>, dim3(grid.x), dim3(256), t.lds, (hipStream_t)stream, na);
        else RSG_LAUNCH(conv_pair_next_kernel<F16>, dim3(grid.x), dim3(256), t.lds, (hipStream_t)stream, na);
        RSG_CHECK_LAUNCH("conv_pair_next");
        return 0;
    }
    if (head_x) {
        a.in0 = (const char*)head_x; a.ldi0 = head_ld; a.wpk0 = (const uint4*)head_w; a.bias0 = head_bias;
#ifdef RSG_DEBUG
        if (g_bottleneck_debug & 1) a.res = nullptr;
#endif
        if (g_bottleneck_head == 1) {
            if (inner->dtype == RSG_BF16) RSG_LAUNCH((bottleneck_kernel<BF16, 1>), dim3(grid.x), dim3(256), t.lds, (hipStream_t)stream, a);
            else RSG_LAUNCH((bottleneck_kernel<F16, 1>), dim3(grid.x), dim3(256), t.lds, (hipStream_t)stream, a);
        } else {
            if (inner->dtype == RSG_BF16) RSG_LAUNCH((bottleneck_kernel<BF16, 2>), dim3(grid.x), dim3(256), t.lds, (hipStream_t)stream, a);
            else RSG_LAUNCH((bottleneck_kernel<F16, 2>), dim3(grid.x), dim3(256), t.lds, (hipStream_t)stream, a);
        }
    } else {
        if (inner->dtype == RSG_BF16) RSG_LAUNCH(conv_pair_kernel<BF16>, dim3(grid.x), dim3(256), t.lds, (hipStream_t)stream, a);
        else RSG_LAUNCH(conv_pair_kernel<F16>, dim3(grid.x), dim3(256), t.lds, (hipStream_t)stream, a);
    }
    RSG_CHECK_LAUNCH("conv_pair");
    return 0;
}


extern "C" int rsg_conv_pair(const rsg_conv_desc* inner, const void* tail_w_packed, const float* tail_bias,
                             int32_t tail_out_c, const void* residual, int32_t res_ld, void* out16, int32_t out_ld,
                             int32_t relu, void* stream) {
    return pair_launch(inner, nullptr, 0, nullptr, nullptr, tail_w_packed, tail_bias, tail_out_c, residual, res_ld,
                       out16, out_ld, relu, stream);
}

// rsg_conv_pair + the following Bottleneck's conv1 (1x1, 256 -> 64, BatchNorm folded, ReLU) in the same launch; next_w_packed
// must be packed from conv1's weight with its INPUT channels permuted: packed channel l = source channel
// rsg_conv_pair_next_perm(l) (the order in which an accumulator tile serves as the next MFMA's B operand).
extern "C" int32_t rsg_conv_pair_next_perm(int32_t l) { return (l & ~0x18) | (((l >> 3) & 1) << 4) | (((l >> 4) & 1) << 3); }
extern "C" int rsg_conv_pair_next(const rsg_conv_desc* inner, const void* tail_w_packed, const float* tail_bias,
                                  const void* residual, int32_t res_ld, void* out16, int32_t out_ld, const void* next_w_packed,
                                  const float* next_bias, void* next_out16, int32_t next_out_ld, const void* cat_x, int32_t cat_ld,
                                  void* stream) {
    if (!next_w_packed) RSG_FAIL("conv_pair_next: null pointer");
    return pair_launch(inner, nullptr, 0, nullptr, nullptr, tail_w_packed, tail_bias, 256, residual, res_ld, out16, out_ld, 1,
                       stream, next_w_packed, next_bias, next_out16, next_out_ld, cat_x, cat_ld);
}

// A 3x3 (or 1x1 / 2x2) conv with 32 input-block channels and exactly 32 output channels (+BN folded, +ReLU) followed by a 1x1
// conv over those 32 channels (+bias) whose result leaves as fp32 NCHW -- RSGNet's predict_net -> final_layer
// (/root/reference/lib/models/pose_rsgnet.py:999-1000) -- in ONE launch: conv_body's register tail.  `inner` describes the first
// conv (its out16 / out_nchw_f32 are ignored: the 32-channel map is not stored); tail_w_packed: the 1x1's weight [tail_cout, 32]
// packed by rsg_pack_conv_weights_host as for rsg_conv_igemm; tail_bias: round_up(tail_cout, 32) floats; out_nchw_f32: [batch, tail_cout, phys_h, phys_w].  Bit-identical to rsg_conv_igemm (16-bit output) + rsg_conv_igemm
// (1x1, fp32 NCHW output).
template <class DT>
__global__ void __launch_bounds__(256) conv_rtail_kernel(const ConvArgs p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    conv_body<DT, 32, 3, 3, 1, 4, 3, false, 0, false, false, ConvArgs, true>(p, rsg_xcd_tile(blockIdx.x, gridDim.x), 0, smem);
}

extern "C" int rsg_conv_tail1x1(const rsg_conv_desc* inner, const void* tail_w_packed, const float* tail_bias, int32_t tail_cout,
                                float* out_nchw_f32, void* stream) {
    if (!inner || !tail_w_packed || !tail_bias || !out_nchw_f32) RSG_FAIL("conv_tail1x1: null pointer");
    if (inner->kh != 3 || inner->kw != 3 || inner->stride != 1 || inner->in_c != 32 || inner->out_c != 32 || inner->residual ||
        inner->up != 1 || inner->relu_tile_mask)
        RSG_FAIL("conv_tail1x1: the first conv must be 3x3 stride 1, 32 -> 32 channels, no residual (got %dx%d s%d %d -> %d)",
                 inner->kh, inner->kw, inner->stride, inner->in_c, inner->out_c);
    if (tail_cout < 1 || tail_cout > 32) RSG_FAIL("conv_tail1x1: %d output channels (1..32: one MFMA tile)", tail_cout);
    if ((((uintptr_t)tail_w_packed | (uintptr_t)tail_bias) & 15) || ((uintptr_t)out_nchw_f32 & 3)) RSG_FAIL("conv_tail1x1: alignment");
    rsg_conv_desc d = *inner;
    d.out16 = nullptr; d.out_nchw_f32 = out_nchw_f32;      // (validated as the conv's own fp32 output; the kernel writes the tail's)
    d.out_c_store = 0; d.tile_hint = 0;
    ConvArgs a;
    TilePlan t;
    dim3 grid;
    if (rsg_conv_prepare(&d, a, t, grid)) return 1;
    if (t.CINB != 32 || t.WM != 1 || t.NT != 3 || grid.y != 1) {
        // the chooser spread a small grid over finer tiles: force the (1 x 4 waves, 3 pixel groups) form this kernel is built for
        d.tile_hint = 1 | (3 << 4) | (std::min(inner->out_h, 8) << 8) | (std::min(inner->out_w, 48) << 16);
        if (rsg_conv_prepare(&d, a, t, grid)) return 1;
        if (t.CINB != 32 || t.WM != 1 || t.NT != 3 || grid.y != 1) RSG_FAIL("conv_tail1x1: no (1, 4, 3) tiling for a %dx%d map", inner->out_h, inner->out_w);
    }
    a.out = nullptr;
    a.wpk2 = (const uint4*)tail_w_packed; a.bias2 = tail_bias; a.C2 = tail_cout;
    if (inner->dtype == RSG_BF16) RSG_LAUNCH(conv_rtail_kernel<BF16>, grid, dim3(256), t.lds, (hipStream_t)stream, a);
    else RSG_LAUNCH(conv_rtail_kernel<F16>, grid, dim3(256), t.lds, (hipStream_t)stream, a);
    RSG_CHECK_LAUNCH("conv_tail1x1");
    return 0;
}

// logical input channel l of the head's packed weights <-> channel of x (see conv_body HEAD)
extern "C" int32_t rsg_bottleneck_head_perm(int32_t l) { return 128 * ((l >> 3) & 1) + 8 * (l >> 4) + (l & 7); }

// Bottleneck without a downsample branch (pose_hrnet.py:85-101) in one launch.  x: [batch,h,w] pixels of 256
// channels (stride x_ld); `inner` describes the 3x3 conv on the 64-channel map conv1 would have produced (its inp is
// ignored); head_w_packed: conv1's weights, input channels permuted by rsg_bottleneck_head_perm before packing.
extern "C" int rsg_bottleneck_fused(const rsg_conv_desc* inner, const void* x, int32_t x_ld, const void* head_w_packed,
                                    const float* head_bias, const void* tail_w_packed, const float* tail_bias,
                                    void* out16, int32_t out_ld, void* stream) {
    if (!inner || !x || !head_w_packed || !head_bias) RSG_FAIL("bottleneck_fused: null pointer");
    if (inner->in_c != 64 || inner->in_h != inner->out_h || inner->in_w != inner->out_w || inner->pad_y != 1 ||
        inner->pad_x != 1)
        RSG_FAIL("bottleneck_fused: the inner conv must be 3x3, stride 1, pad 1, 64 -> 64 channels");
    if (x_ld < 256 || x_ld % 8 || (((uintptr_t)x | (uintptr_t)head_w_packed | (uintptr_t)head_bias) & 15))
        RSG_FAIL("bottleneck_fused: x_ld=%d / alignment", x_ld);
    if (x == out16) RSG_FAIL("bottleneck_fused: in-place operation is not supported (tiles read their neighbours' halo)");
    rsg_conv_desc d = *inner;
    d.in = x;                          // placeholder for the shared validation
    d.in_ld = x_ld;
    return pair_launch(&d, x, x_ld, head_w_packed, head_bias, tail_w_packed, tail_bias, 256, x, x_ld, out16, out_ld, 1,
                       stream);
}

extern "C" int rsg_conv_igemm(const rsg_conv_desc* d, void* stream) {
    ConvArgs a;
    TilePlan t;
    dim3 grid;
    if (rsg_conv_prepare(d, a, t, grid)) return 1;
    conv_fn fn = d->dtype == RSG_BF16 ? lookup<BF16>(t.CINB, d->kh * d->kw, t.WM, t.NT)
                                      : lookup<F16>(t.CINB, d->kh * d->kw, t.WM, t.NT);
    if (!fn) RSG_FAIL("conv: no kernel for CINB=%d k=%d WM=%d NT=%d", t.CINB, d->kh, t.WM, t.NT);
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

// ---- device-side packing (training: the fp32 master weights change every step) ---------------------------------
// The same fragment order as rsg_pack_conv_weights_host, no BatchNorm folding (training normalises with batch
// statistics in its own kernels).  One launch packs EVERY conv of the model: items[] (device) names the tensors,
// item k owns blocks [first_block[k], first_block[k+1]); a thread writes one lane's 8 elements (16 bytes).
// transpose = 1 packs the data-gradient form of the same weight: W'[i][o][ky][kx] = W[o][i][kh-1-ky][kw-1-kx]
// (conv_transpose of a stride-1 conv = a conv with these weights; pose_hrnet.py's convs, backward).
template <class DT>
__global__ void __launch_bounds__(256) pack_weights_kernel(const rsg_pack_item* __restrict__ items, int n_items) {
    // One workgroup per (item, 32-channel output tile, input-channel block): the fp32 weights of the tile are whole
    // contiguous runs of the tensor ([cout][cin][taps]: cinb x taps floats per output channel; transposed: 32 x taps per
    // original output channel), read coalesced into a 16-bit LDS tile [32][cinb][taps]; every thread then gathers its
    // lane's 8 elements from LDS.  (The first version gathered them straight from global memory, 4 bytes at a 36-byte
    // stride: 512 us for the 29 M parameters of HRNet-W32 in both forms.)
    __shared__ typename DT::elem tile[32 * 64 * 9];
    // binary search: last item whose first_block <= blockIdx.x
    int lo = 0, hi = n_items - 1;
    while (lo < hi) {
        const int mid = (lo + hi + 1) >> 1;
        if (items[mid].first_block <= (int)blockIdx.x) lo = mid; else hi = mid - 1;
    }
    const rsg_pack_item it = items[lo];
    const int wcin = it.cin, wcout = it.cout, ntaps = it.kh * it.kw;   // the weight tensor is [wcout][wcin][kh][kw]
    const int cin = it.transpose ? wcout : wcin, cout = it.transpose ? wcin : wcout;   // the packed conv's own cin/cout
    const int cin_pad = cin <= 16 ? 16 : (cin % 16 == 0 ? cin : (cin + 31) / 32 * 32);
    const int cinb = cin_pad % 64 == 0 ? 64 : (cin_pad % 32 == 0 ? 32 : (cin_pad == 48 ? 48 : 16));
    const int kpt = cinb / 16, nchunks = cin_pad / cinb, KSB = ntaps * kpt, KS = nchunks * KSB;
    const int local = (int)blockIdx.x - it.first_block;
    const int ct = local / nchunks, ch = local - ct * nchunks;
    if (ct * 32 >= (cout + 31) / 32 * 32) return;
    const int tid = threadIdx.x;
    const int run = cinb * ntaps;                                       // tile row: [ci_l][t]
    if (!it.transpose) {
        for (int idx = tid; idx < 32 * run; idx += 256) {
            const int col = idx / run, r = idx - col * run;             // r = ci_l * ntaps + t
            const int co = ct * 32 + col, ci = ch * cinb + r / ntaps;
            float f = 0.f;
            if (co < cout && ci < cin) f = it.w[((long)co * wcin + ch * cinb) * ntaps + r];
            tile[idx] = DT::from_f32(f);
        }
    } else {
        // packed (co', ci', t') = original (i, o, taps - 1 - t'): for a fixed o the 32 i of the tile are one run
        const int orun = 32 * ntaps;
        for (int idx = tid; idx < cinb * orun; idx += 256) {
            const int ol = idx / orun, r = idx - ol * orun;             // r = i_l * ntaps + t
            const int il = r / ntaps, t = r - il * ntaps;
            const int o = ch * cinb + ol, i = ct * 32 + il;
            float f = 0.f;
            if (o < wcout && i < wcin) f = it.w[((long)o * wcin + ct * 32) * ntaps + r];
            tile[il * run + ol * ntaps + (ntaps - 1 - t)] = DT::from_f32(f);
        }
    }
    __syncthreads();
    uint4* dst = reinterpret_cast<uint4*>(it.dst) + ((long)ct * KS + (long)ch * KSB) * 64;
    for (int e = tid; e < KSB * 64; e += 256) {
        const int lane = e & 63, ks = e >> 6;
        const int c = ks % kpt, t = ks / kpt;
        const int i = lane & 31, h = lane >> 5;
        const int col = 16 * ((i >> 2) & 1) + (i & 3) + 4 * (i >> 3);
        typename DT::vec8 v;
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] = tile[col * run + (c * 16 + h * 8 + j) * ntaps + t];
        dst[e] = __builtin_bit_cast(uint4, v);
    }
}

extern "C" int32_t rsg_pack_item_blocks(int32_t cin, int32_t cout, int32_t kh, int32_t kw) {
    // cin/cout of the PACKED conv (swap them for a transposed item): one block per (output tile, input-channel block)
    (void)kh; (void)kw;
    const int cin_pad = rsg_conv_padded_cin(cin);
    return ceil_div(cout, 32) * (cin_pad / pick_cinb(cin_pad));
}

extern "C" int rsg_pack_conv_weights_device(const rsg_pack_item* items_device, int32_t n_items, int32_t total_blocks,
                                            int32_t dtype, void* stream) {
    if (!items_device || n_items < 1 || total_blocks < 1) RSG_FAIL("pack_device: bad arguments");
    if (dtype == RSG_BF16)
        RSG_LAUNCH(pack_weights_kernel<BF16>, dim3(total_blocks), dim3(256), 0, (hipStream_t)stream, items_device, n_items);
    else
        RSG_LAUNCH(pack_weights_kernel<F16>, dim3(total_blocks), dim3(256), 0, (hipStream_t)stream, items_device, n_items);
    RSG_CHECK_LAUNCH("pack_device");
    return 0;
}
