/*
 * librsgnet_hip.so — C ABI of the MI355X (gfx950) kernels behind RSGNet's per-crop forward hot path.
 *
 * The reference has no FFI: its hot path is torch operators called from
 * lib/models/pose_hrnet.py, pose_rsgnet.py and association.py.  Each entry point below
 * replaces the torch-operator group named in its comment; a maintainer binds them with ctypes (see
 * INTEGRATION.md).  Conventions:
 *   - every pointer is a DEVICE pointer unless the name ends in _host; nothing is retained past the call
 *   - activations are NHWC, 16-bit (dtype RSG_F16 or RSG_BF16), channel counts padded as documented
 *   - `stream` is a hipStream_t passed as void*; kernels are only enqueued, never synchronised
 *   - return value: 0 on success, non-zero on error; rsg_last_error() gives the message (thread-local)
 */
#ifndef RSGNET_HIP_H
#define RSGNET_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

enum { RSG_F16 = 0, RSG_BF16 = 1 };

const char* rsg_last_error(void);
/* The version of THIS header's entry points and structs; rsg_version() returns the one the library was built with, and a
 * caller compares the two before its first call (rsgnet_amd/hip.py does).  Bumped with every incompatible change:
 *   1  rounds 1-3
 *   2  round 5 (covers round 4's changes, which shipped under 1: rsg_trp_relation_mse[_dense] gained a workspace pointer,
 *      rsg_chain_item lost its tail fields, rsg_fuse_band and the chain tails were removed; round 5 adds rsg_fuse_module) */
#define RSG_ABI_VERSION 2
int rsg_version(void);
/* tuning knobs for experiments (process-wide): "chain_max_ni" = images per workgroup cap of the chain kernel;
 * "conv_spread" = n: a conv whose grid would have fewer than n workgroups is re-tiled into more, smaller ones (default 128,
 * 0 = off; the tiling never changes a result bit); the others are listed at rsg_tune in csrc/chain.hip */
int rsg_tune(const char* key, int32_t value);

/* ---- convolution family ---------------------------------------------------------------------
 * One implicit-GEMM kernel family covers
 *   nn.Conv2d 3x3 s1/s2 p1 + BatchNorm2d(eval) [+ residual] [+ ReLU]   pose_hrnet.py:23-59,62-101,
 *                                                                       336-375 (transitions), 215-243 (fuse down)
 *   nn.Conv2d 1x1 (+BN | +bias)                                         pose_hrnet.py:68-76,203-208,326-332
 *   nn.ConvTranspose2d k4 s2 p1 + BN + ReLU (as 4 sub-pixel 2x2 convs)  pose_rsgnet.py:727-736
 *   functional conv2d with SGM-generated weights                        pose_rsgnet.py:1005
 * BatchNorm is folded into the packed weights/bias by rsg_pack_conv_weights_host().
 *
 * Input  : [batch, in_h, in_w, in_ld] 16-bit, the conv reads channels [0, in_c) of each pixel
 *          (pass a pointer already offset to the first channel when reading a slice of a concat buffer);
 *          in_c must equal rsg_conv_padded_cin(real cin) and the padding channels must hold zeros.
 * Output : out16  [batch, phys_h, phys_w, out_ld] 16-bit, channels [0, round_up(out_c,32)) written (or out_c_store)
 *          (padding channels get bias-only values = 0 when packed by the packer), and/or
 *          out_nchw_f32 [batch, out_c, phys_h, phys_w] fp32 (classifier heads).
 * Geometry: logical output pixel (y,x), y<out_h, x<out_w reads input (y*stride + ky - pad_y, ...)
 *          and lands on physical pixel (y*up + up_y, x*up + up_x)  (up=1, up_y=up_x=0 for plain convs;
 *          up=2 for the deconv phases).
 */
/* A wide map (HRNet layer1's 256 channels, /root/reference/lib/models/pose_hrnet.py:63-101) stored as channel-tile PLANES
 * [C/32][batch*h*w][32] instead of dense NHWC [batch*h*w][C]: the MFMA accumulator layout of the conv kernels has a lane's
 * pixel on the lane axis and 16 channels in registers, so a wave's store (or residual load) of one 32-channel tile is
 * 32 pixels x 64 B -- one contiguous 2 KB run in planes, 32 pieces at a 512-byte stride in NHWC.  The pixel stride (ld) of
 * such a map is 32. */
#define RSG_LAYOUT_OUT_PLANES 1
#define RSG_LAYOUT_RES_PLANES 2
#define RSG_LAYOUT_IN_PLANES 4
typedef struct rsg_conv_desc {
    const void* in;
    const void* w_packed;      /* from rsg_pack_conv_weights_host, copied to the device */
    const float* bias;         /* [round_up(out_c,32)] */
    const void* residual;      /* optional, [batch, phys_h, phys_w, res_ld] 16-bit, added before ReLU */
    void* out16;               /* optional */
    float* out_nchw_f32;       /* optional */
    int32_t batch, in_h, in_w, in_c, in_ld;
    int32_t out_h, out_w, out_c, out_ld, res_ld;
    int32_t kh, kw, stride, pad_y, pad_x;
    int32_t up, up_y, up_x, phys_h, phys_w;
    int32_t relu;
    int32_t dtype;             /* RSG_F16 | RSG_BF16 */
    int32_t tile_hint;         /* 0 = auto; else WM | NT<<4 | TH<<8 | TW<<16 (tuning/tests) */
    int32_t relu_tile_mask;    /* 0 = use `relu` for all channels; else bit t = ReLU on output channels [32t, 32t+32) */
    int32_t out_c_store;       /* 0 = write round_up(out_c,32) channels of out16; else only [0, out_c_store) (multiple
                                  of 8): for outputs that are a channel slice narrower than a 32-channel tile */
    int32_t layout;            /* RSG_LAYOUT_* bits; 0 = every map dense NHWC.  Honoured by rsg_conv_pair / rsg_conv_pair_next
                                  (OUT, RES: the tail's output / residual) and rsg_conv_dual (IN) only */
    /* two more addends of the epilogue (same layout rules as `residual`; NULL = none): the HRNet fuse sum of the
       lowest-resolution output rides on the last conv that feeds it (pose_hrnet.py:259-266) */
    const void* residual2;
    const void* residual3;
    int32_t res2_ld, res3_ld;
} rsg_conv_desc;

int rsg_conv_igemm(const rsg_conv_desc* d, void* stream);

/* what the auto-tiler picked for a descriptor (for logging / tuning): fills 8 ints
 * {CINB, WM, WN, NT, TH, TW, lds_bytes, grid_x} */
/* HRNet stem in one launch (pose_hrnet.py:285-291,429-434): conv1 3x3 s2 3->64 (+BN, ReLU) as a 2x2 conv over the
 * space-to-depth input xs = rsg_nchw_f32_to_s2d_nhwc16(x) (or xs = NULL and x_nchw_f32 = the fp32 NCHW picture: the
 * conversion then happens while staging), then conv2 3x3 s2 64->64 (+BN, ReLU); the half-resolution
 * 64-channel map between them stays in LDS.  w1: packed 2x2 weights (cin 16, cout 64), w2: packed 3x3 (cin 64, cout 64);
 * out16: [batch, h/4, w/4, out_ld] 16-bit NHWC, 64 channels written; h, w = input picture size (multiples of 4). */
int rsg_stem_fused(const void* xs, const float* x_nchw_f32, const void* w1_packed, const float* bias1, const void* w2_packed, const float* bias2,
                   void* out16, int32_t out_ld, int32_t batch, int32_t h, int32_t w, int32_t dtype, void* stream);
/* rsg_stem_fused + conv1 of the first Bottleneck (pose_hrnet.py:85-87 of layer1.0: 1x1, 64 -> 64, BatchNorm folded, ReLU) on the
 * stem's output in the same launch (its input pixel is complete in the workgroup that produced it: the 64-channel map is
 * not read back and one launch goes).  w3/bias3 from rsg_pack_conv_weights_host(cin 64, cout 64, 1x1); out3: [batch, h/4,
 * w/4, out3_ld] 16-bit.  Bit-identical to rsg_stem_fused followed by rsg_conv_igemm. */
int rsg_stem_fused_conv1(const void* xs, const float* x_nchw_f32, const void* w1_packed, const float* bias1, const void* w2_packed,
                         const float* bias2, void* out16, int32_t out_ld, const void* w3_packed, const float* bias3, void* out3,
                         int32_t out3_ld, int32_t batch, int32_t h, int32_t w, int32_t dtype, void* stream);
/* Bottleneck conv2 -> conv3 (pose_hrnet.py:93-101) in one launch: `inner` = the 3x3 stride-1 conv with 64 output
 * channels (BatchNorm folded, ReLU; its out16 / residual / relu fields are ignored), then a 1x1 conv to tail_out_c
 * channels (weights from rsg_pack_conv_weights_host with kh = kw = 1, cin = 64) + residual + optional ReLU.  The
 * 64-channel intermediate stays in LDS. */
int rsg_conv_pair(const rsg_conv_desc* inner, const void* tail_w_packed, const float* tail_bias, int32_t tail_out_c,
                  const void* residual, int32_t res_ld, void* out16, int32_t out_ld, int32_t relu, void* stream);
/* rsg_conv_pair with a 256-channel tail, followed IN THE SAME LAUNCH by the next Bottleneck's conv1 (pose_hrnet.py:85-87
 * of the following block: 1x1 256 -> 64, BatchNorm folded, ReLU) on the block output it has just produced: the finished
 * accumulator tiles are the B operand of conv1's MFMAs, so the 256-channel map is not read back.  next_w_packed: conv1's
 * weights packed by rsg_pack_conv_weights_host AFTER permuting its input channels, packed channel l = source channel
 * rsg_conv_pair_next_perm(l); next_out16: [batch,h,w] pixels of 64 channels (stride next_out_ld).
 * cat_x != NULL: the FIRST Bottleneck, whose residual is projected (pose_hrnet.py:93-101 with `downsample`): the tail is a
 * 1x1 conv over the channel concat [conv2 output | block input x] (tail_w_packed packed with cin = 128, weights
 * [s3*W3 | sd*Wd]); cat_x: the 64-channel block input (pixel stride cat_ld), residual must be NULL. */
int32_t rsg_conv_pair_next_perm(int32_t l);
int rsg_conv_pair_next(const rsg_conv_desc* inner, const void* tail_w_packed, const float* tail_bias, const void* residual,
                       int32_t res_ld, void* out16, int32_t out_ld, const void* next_w_packed, const float* next_bias,
                       void* next_out16, int32_t next_out_ld, const void* cat_x, int32_t cat_ld, void* stream);
/* A 3x3 stride-1 conv 32 -> 32 (+BN folded, +ReLU) and the 1x1 conv (+bias) that follows it, fp32 NCHW result, in ONE launch --
 * RSGNet's predict_net -> final_layer (/root/reference/lib/models/pose_rsgnet.py:999-1000): the first conv's accumulators,
 * rounded to 16 bits, are the second's MFMA operand in registers; the 32-channel map between them is never stored.
 * tail_w_packed: the 1x1's weight packed by rsg_pack_conv_weights_host exactly as for rsg_conv_igemm; tail_bias: 32 floats; tail_cout <= 32; out_nchw_f32: [batch, tail_cout, phys_h, phys_w].
 * Bit-identical to the two rsg_conv_igemm launches. */
int rsg_conv_tail1x1(const rsg_conv_desc* inner, const void* tail_w_packed, const float* tail_bias, int32_t tail_cout,
                     float* out_nchw_f32, void* stream);
/* A whole Bottleneck without downsample branch (pose_hrnet.py:85-101: conv1 1x1 256->64, conv2 3x3, conv3 1x1
 * 64->256, each + BatchNorm, + x, ReLU) in ONE launch: conv1 is evaluated per tile (with its one-pixel halo) straight
 * into LDS, so neither 64-channel map ever reaches HBM.  x: 256-channel NHWC map (pixel stride x_ld elements), also
 * the residual; inner: the 3x3 conv (64 -> 64; its `in` is ignored); head_w_packed: conv1 packed by
 * rsg_pack_conv_weights_host AFTER permuting its input channels, packed channel l = x channel
 * rsg_bottleneck_head_perm(l); out16: [batch,h,w,256] with pixel stride out_ld, must not alias x. */
int32_t rsg_bottleneck_head_perm(int32_t l);
int rsg_bottleneck_fused(const struct rsg_conv_desc* inner, const void* x, int32_t x_ld, const void* head_w_packed,
                         const float* head_bias, const void* tail_w_packed, const float* tail_bias, void* out16,
                         int32_t out_ld, void* stream);
int rsg_conv_plan(const rsg_conv_desc* d, int32_t* out8);

int32_t rsg_conv_padded_cin(int32_t cin);              /* 3->16, 17->32, multiples of 16 kept ... */
size_t rsg_conv_packed_bytes(int32_t cin, int32_t cout, int32_t kh, int32_t kw);
/* HOST function.  w_oihw_host: fp32 [cout][cin][kh][kw] (nn.Conv2d layout).  scale/shift: per-cout
 * folded BatchNorm (gamma/sqrt(var+eps), beta - mean*scale) or NULL; conv_bias: per-cout or NULL.
 * Writes the fragment-ordered 16-bit weights (rsg_conv_packed_bytes) and fp32 bias [round_up(cout,32)]:
 *   w' = w*scale (rounded once to 16 bits), bias' = conv_bias*scale + shift. */
int rsg_pack_conv_weights_host(const float* w_oihw_host, const float* scale_host, const float* shift_host,
                               const float* conv_bias_host, int32_t cin, int32_t cout, int32_t kh,
                               int32_t kw, int32_t dtype, void* w_packed_host, float* bias_host);

/* ---- HIP graph of a launch list -----------------------------------------------------------------
 * Between rsg_graph_begin and rsg_graph_end every rsg_* launch made by the calling thread becomes a
 * kernel node of `*graph` instead of being enqueued (the stream argument is ignored); node ids count
 * the launches from 0.  rsg_graph_next_deps names the nodes the NEXT launch depends on (none = root).
 * rsg_graph_end instantiates; rsg_graph_launch replays the whole forward on a stream with one call.
 * This replaces the reference's per-operator dispatch from Python (lib/core/function.py:121,389). */
int rsg_graph_begin(void** graph);
int rsg_graph_next_deps(void* graph, const int32_t* node_ids, int32_t n);
int rsg_graph_node_count(void* graph);
int rsg_graph_end(void* graph);
int rsg_graph_abort(void* graph);
int rsg_graph_launch(void* graph, void* stream);
int rsg_graph_destroy(void* graph);

/* transition1 of HRNet (pose_hrnet.py:336-375, :437-443) in one launch: `a` describes the stride-1 3x3 conv (pad 1, cin a
 * multiple of 64, 16-bit NHWC output, BatchNorm folded, ReLU per a->relu, no residual); the stride-2 3x3 conv of the second
 * new branch reads the SAME input: wB_packed / biasB from rsg_pack_conv_weights_host (cin, cout_b, 3x3), outB16
 * [batch, h/2, w/2, out_ld_b] (h, w even).  Built for (a->out_c <= 32, cout_b = 64: HRNet-W32) and (a->out_c <= 64,
 * cout_b = 96: HRNet-W48).  The input tile is staged once for both; results are bit-identical to two rsg_conv_igemm launches. */
int rsg_conv_dual(const rsg_conv_desc* a, const void* wB_packed, const float* biasB, void* outB16, int32_t out_ld_b,
                  int32_t cout_b, int32_t relu_b, void* stream);

/* ---- fused BasicBlock chain -----------------------------------------------------------------
 * x = branches[i](x) of a HighResolutionModule (pose_hrnet.py:254-255): nlayers = 2*nblocks 3x3 convs
 * C->C (BatchNorm folded, ReLU after each, block input added before every second ReLU), with the
 * activation tile resident in LDS for the whole chain.  in/out: dense NHWC 16-bit [batch,h,w,c],
 * in != out.  w_packed[l], bias[l]: per-layer device pointers from rsg_pack_conv_weights_host
 * (HOST arrays of nlayers device pointers).  c in {32,64,128,256} (HRNet-W32; 8 waves per workgroup) or
 * {48,96,192,384} (HRNet-W48; 12 waves, one wave row per 32-channel output tile, c = 48 padded to two tiles).
 * rsg_basic_chain_supported returns 1 when (c,h,w,nlayers) can run and fills
 * info4 = {tile rows, images per workgroup, LDS bytes, workgroups}.
 * workspace: device scratch of rsg_basic_chain_workspace_bytes(...) bytes; it carries each block's input
 * to the residual add of that block and holds nothing between calls, so chains that never run
 * concurrently may share one. */
int rsg_basic_chain(const void* in, void* out, const void* const* w_packed, const float* const* bias,
                    int32_t nlayers, int32_t batch, int32_t h, int32_t w, int32_t c, int32_t dtype,
                    void* workspace, void* stream);
size_t rsg_basic_chain_workspace_bytes(int32_t c, int32_t h, int32_t w, int32_t nlayers, int32_t batch);
/* c = 48 runs on the 16x16x32 MFMA (exactly 48 channels, csrc/chain48.h): ONE BasicBlock per launch (nlayers == 2), no
 * hand-off flags, and its layers are packed by rsg_pack_chain48_weights_host (w fp32 [48][48][3][3], BatchNorm
 * scale / shift or NULL; rsg_chain48_packed_bytes() bytes of weights, 48 floats of bias) instead of
 * rsg_pack_conv_weights_host. */
size_t rsg_chain48_packed_bytes(void);
int rsg_pack_chain48_weights_host(const float* w, const float* scale, const float* shift, int32_t dtype, void* wp_out, float* bias_out);
/* Weight pack of the chains by width.  rsg_chain_pack_kind(c): 16 = the widths of HRNet-W32 (32 / 64 / 128 / 256 channels:
 * chains on the 16x16x32 MFMA, csrc/chain16.h) take rsg_pack_chain16_weights_host (w fp32 [c][c][3][3], BatchNorm scale / shift
 * or NULL; 18 c^2 bytes of weights -- the size of the conv pack -- and c floats of bias); 48 = rsg_pack_chain48_weights_host;
 * 0 = rsg_pack_conv_weights_host.  These chains need no block-input workspace (rsg_basic_chain_workspace_bytes returns 16). */
int rsg_chain_pack_kind(int32_t c);
int rsg_pack_chain16_weights_host(const float* w, const float* scale, const float* shift, int32_t c, int32_t dtype,
                                  void* wp_out, float* bias_out);
int rsg_basic_chain_supported(int32_t c, int32_t h, int32_t w, int32_t nlayers, int32_t batch,
                              int32_t* info4);
/* Up to 6 chains (the branches of one HighResolutionModule, pose_hrnet.py:254-255) in ONE launch; the
 * workgroups of items[0] are dispatched first, so put the longest chain there.
 * A branch whose map needs row tiles (halo grows with the chain) can be cut in two items that hand over inside
 * the launch: the first sets done_flags[workgroup] after its stores, the second (later in the array, same
 * shape, in == the first's out) names those flags in wait_flags, polls the tiles it reads (bounded) and clears
 * what it saw.  done_flags: int32[3 * workgroups of the item], zeroed by the caller before the first launch (a
 * launch leaves every word that was seen at zero again) and again whenever rsg_chain_handoff_timeouts reports a
 * poll that gave up. */
typedef struct rsg_chain_item {
    const void* in;
    void* out;
    const void* const* w_packed;   /* HOST array of nlayers device pointers */
    const float* const* bias;      /* HOST array of nlayers device pointers */
    int32_t nlayers, batch, h, w, c;
    int32_t images_per_workgroup;  /* whole-image maps: images one workgroup takes (0 = default 1); fewer, longer workgroups */
    void* workspace;               /* rsg_basic_chain_workspace_bytes(c,h,w,nlayers,batch) bytes, own per item */
    void* done_flags;              /* NULL, or 3 x int32 per workgroup (see above) */
    const void* wait_flags;        /* NULL, or the done_flags of an earlier item */
    int32_t split_rows;            /* 1: a map whose images fit LDS whole is cut into TWO row tiles all the same (halo = nlayers
                                      rows, recomputed): 1.5x the work of a 32-row map's 8-layer chain on twice the CUs, for
                                      the chain that is a launch's critical path; rsg_basic_chain_supported_split says whether
                                      the shape allows it.  0: the library's own choice */
    int32_t exchange;              /* 1 (HRNet-W32's widths, nlayers a multiple of 4): the row tiles hold a halo of nlayers / 2 rows and
                                      trade it half way through the chain INSIDE the launch -- every tile writes the rows its
                                      neighbours lack to `workspace` (rsg_basic_chain_exchange_bytes bytes), raises a word of
                                      `done_flags` (3 zeroed words per workgroup, self-clearing) for each, waits for theirs --
                                      instead of two chains of nlayers / 2 layers handing the whole map over through memory.
                                      One staging and one store per tile; every tile counts as a waiter of the launch */
    int32_t in_ld, out_ld;         /* pixel stride (elements) of in / out; 0 = dense (= c).  Only c = 48 takes other strides
                                      (HRNet-W48 keeps its 48-channel maps in 64-channel buffers) */
} rsg_chain_item;
int rsg_basic_chain_group(const rsg_chain_item* items, int32_t n, int32_t dtype, void* stream);
/* rsg_chain_item.exchange: bytes of the item's workspace, 0 when the shape does not run in that form; info4 (or NULL) =
 * {rows per tile, 1, LDS bytes, workgroups} */
size_t rsg_basic_chain_exchange_bytes(int32_t c, int32_t h, int32_t w, int32_t nlayers, int32_t batch, int32_t* info4);
/* rsg_basic_chain_supported for an item with split_rows = 1: 1 and info4 = {rows per tile, 1, LDS bytes, workgroups} when
 * the map can run as two row tiles, else 0 */
int rsg_basic_chain_supported_split(int32_t c, int32_t h, int32_t w, int32_t nlayers, int32_t batch, int32_t* info4);
/* Sticky error word of the hand-off: the number of polls that gave up since the library was loaded on the current
 * device.  An increase means a consumer chain staged without its producer's data: that launch's output is wrong and
 * the flag words it was waiting on must be re-zeroed by the caller before the next launch.
 * rsg_chain_handoff_timeouts synchronises the device; the _async form enqueues a 4-byte copy into PINNED host
 * memory on `stream` (the caller reads it once an event behind it has completed), which is how the production path
 * watches the word without stalling (rsgnet_amd/plan.py).
 * rsg_basic_chain_handoff_max_waiters: the launch is deadlock-free whatever the dispatch order as long as fewer
 * workgroups than this + 1 wait on flags (one chain workgroup is resident per compute unit, so at least one unit
 * always runs workgroups that never block); rsg_basic_chain_group refuses a launch with more. */
int rsg_chain_handoff_timeouts(uint32_t* count);
int rsg_chain_handoff_timeouts_async(uint32_t* pinned_host, void* stream);
int rsg_basic_chain_handoff_max_waiters(void);

/* ---- layout / elementwise -------------------------------------------------------------------- */
/* x.float NCHW -> NHWC 16-bit with channels zero-padded to c_pad (input of the stem conv,
 * pose_hrnet.py:429; replaces the implicit layout the reference's conv1 reads) */
int rsg_nchw_f32_to_nhwc16(const float* in, void* out, int32_t batch, int32_t c, int32_t h, int32_t w,
                           int32_t c_pad, int32_t dtype, void* stream);

/* x.float NCHW -> space-to-depth(2) NHWC 16-bit [batch, h/2, w/2, 16]: channel (py*2+px)*c + k holds
 * in[b,k,2Y+py,2X+px], channels 4c..15 are zero (c <= 4).  Lets a 3x3 stride-2 conv over few channels (the
 * stem's conv1, pose_hrnet.py:285-286) run as a 2x2 stride-1 conv with pad (1,1). */
int rsg_nchw_f32_to_s2d_nhwc16(const float* in, void* out, int32_t batch, int32_t c, int32_t h, int32_t w,
                               int32_t dtype, void* stream);

/* HighResolutionModule fuse for ONE output branch (pose_hrnet.py:259-266):
 *   out = ReLU( sum_k same[k] + sum_k nearest_up(low[k], low_shift[k]) )
 * same[k]: [batch,h,w,c] 16-bit (identity branch and the strided-conv results),
 * low[k]:  [batch,h>>s,w>>s,c] 16-bit (the 1x1+BN results, upsampled by 2^s = nn.Upsample nearest). */
int rsg_hr_fuse_sum(const void* const* same, int32_t n_same, const void* const* low,
                    const int32_t* low_shift, int32_t n_low, void* out, int32_t batch, int32_t h,
                    int32_t w, int32_t c, int32_t dtype, void* stream);
/* The fuse sum of the highest-resolution output (as rsg_hr_fuse_sum: 32 channels, operand strides same_ld / low_ld in
 * elements, 0 = dense) and the 1x1 classifier on it (pose_hrnet.py:326-332,461) in one launch: the summed map goes through
 * LDS into the classifier's MFMAs and is never stored.  w_packed / bias: rsg_pack_conv_weights_host(cin 32, cout <= 32, 1x1)
 * with the classifier's bias; out: fp32 [batch, cout, h, w].  Bit-identical to the sum followed by rsg_conv_igemm. */
int rsg_fuse_classifier(const void* const* same, const int32_t* same_ld, int32_t n_same, const void* const* low,
                        const int32_t* low_shift, const int32_t* low_ld, int32_t n_low, const void* w_packed, const float* bias,
                        float* out_nchw_f32, int32_t batch, int32_t h, int32_t w, int32_t cout, int32_t dtype, void* stream);
/* The whole fuse of the LAST HighResolutionModule (pose_hrnet.py:190-213,259-266 with multi_scale_output = False) and the
 * classifier (:326-332,461) in one launch: the 1x1 convs (+BatchNorm) of the three lower branches on the rows a workgroup
 * needs, the sum with x0, ReLU, the classifier.  x0: [batch,h,w,ld0], 32 channels; lows[j]: [batch, h>>(j+1), w>>(j+1),
 * low_ld[j]], 64<<j channels; w_low / b_low: rsg_pack_conv_weights_host(cin 64<<j, cout 32, 1x1) with the BatchNorm folded;
 * w_cls / b_cls: the classifier (cin 32, cout <= 32); out: fp32 [batch,cout,h,w].  h, w multiples of 8.  Bit-identical to
 * rsg_conv_igemm x n_low + rsg_fuse_classifier. */
int rsg_fuse_head4(const void* x0, int32_t ld0, const void* const* lows, const int32_t* low_ld, int32_t n_low,
                   const void* const* w_low, const float* const* b_low, const void* w_cls, const float* b_cls,
                   float* out_nchw_f32, int32_t batch, int32_t h, int32_t w, int32_t cout, int32_t dtype, void* stream);

/* One launch for several INDEPENDENT ops of a module's fuse layers (pose_hrnet.py:190-268): up to 12
 * convolutions (descriptors as for rsg_conv_igemm; rsg_conv_groupable tells whether one can join a
 * group) and up to 4 fuse-sums (arguments as for rsg_hr_fuse_sum). */
typedef struct rsg_fuse_item {
    const void* same[4];
    const void* low[3];
    int32_t low_shift[3];
    int32_t n_same, n_low;
    void* out;
    int32_t batch, h, w, c;
    int32_t same_ld[4];        /* channel stride of each operand (0 = dense, i.e. c): operands may be */
    int32_t low_ld[3];         /* channel slices of wider buffers                                       */
    int32_t reserved;
} rsg_fuse_item;
int rsg_conv_groupable(const rsg_conv_desc* d);
/* two steps: _plan (HOST) writes the launch table (rsg_group_table_bytes bytes, HOST memory the caller keeps) and
 * launch2 = {grid blocks, LDS bytes}; _run enqueues the launch with the table as the kernel argument (read from host
 * memory at the call, like any argument). */
size_t rsg_group_table_bytes(void);
int rsg_conv_fuse_group_plan(const rsg_conv_desc* convs, int32_t n_convs, const rsg_fuse_item* fuses,
                             int32_t n_fuses, int32_t dtype, void* table_host, int32_t* launch2);
int rsg_conv_fuse_group_run(const void* table_host, int32_t blocks, int32_t lds, int32_t dtype, void* stream);

/* ---- level 0 of a HighResolutionModule's fuse layers in ONE launch of row bands (csrc/fuse_module.hip) ---------------
 * Replaces, of x_fuse[i] = relu(sum_j fuse_layers[i][j](x[j])) (/root/reference/lib/models/pose_hrnet.py:190-268, forward
 * :250-268): every conv that reads a branch map x[j] directly -- the 1x1 "up" convs j -> i (:199-213) and the first 3x3
 * stride-2 conv of every "down" chain (:217-243) -- and the sums of outputs 0 and 1 (:259-266).  Branch i is
 * [batch, h >> i, w >> i, x_ld[i]] 16-bit NHWC with 32 << i channels (HRNet-W32's widths); h, w multiples of 8.
 * A conv: w_packed / bias from rsg_pack_conv_weights_host(cin = 32 << src, cout, k x k) with the BatchNorm folded;
 *   k = 1: 1x1 at the resolution of branch src; k = 3: 3x3 stride 2 pad 1, output at the resolution of branch src + 1
 *   (src 0 or 1);  role says where the result goes:
 *     RSG_FM_TO_GLOBAL  out16 [batch, h >> l, w >> l, out_ld] (l = the output's level), ReLU if `relu` (the module's later
 *                       fuse levels read it: grouped launches as before)
 *     RSG_FM_UP_Y0      1x1, cout c0, one per branch 1..n-1: the up-sampled addends of y0 (stay in LDS, 16-bit as the
 *                       grouped path stored them)
 *     RSG_FM_UP_Y1      1x1, cout 2 c0, one per branch 2..n-1: the up-sampled addends of y1
 *     RSG_FM_DOWN_Y1    the 0 -> 1 conv (k = 3, cout 2 c0), exactly one: its epilogue is the y1 sum
 * y[0] = relu(x0 + up(t10) + up(t20) + up(t30)), y[1] = relu(round16(conv01(x0)) + x1 + up(t21) + up(t31)) in the
 * operand order and rounding of rsg_hr_fuse_sum; y1_fold = 1 (two branches only): y[1] = relu(conv01(x0) + x1) in fp32, the
 * residual epilogue of rsg_conv_igemm (the module's LAST output, whose sum the grouped path folds into the conv).
 * Results are bit-identical to the grouped launches.  _plan (HOST) validates, chooses the band height (16 or 8 rows of
 * branch 0 by what fits the 160 KB of LDS), deals the conv jobs to the 16 waves and writes the launch table
 * (rsg_fuse_module_table_bytes bytes of HOST memory the caller keeps) and launch2 = {grid blocks, LDS bytes}; it fails
 * (status 1, rsg_last_error) for shapes the kernel does not cover: the caller keeps the grouped path for those. */
#define RSG_FM_MAX_CONVS 12
#define RSG_FM_TO_GLOBAL 0
#define RSG_FM_UP_Y0 1
#define RSG_FM_UP_Y1 2
#define RSG_FM_DOWN_Y1 3
typedef struct rsg_fm_conv {
    const void* w_packed;
    const float* bias;
    void* out16;
    int32_t out_ld;
    int32_t src, k, cout, relu, role;
    int32_t reserved;
} rsg_fm_conv;
typedef struct rsg_fuse_module_desc {
    const void* x[4];
    void* y[2];
    int32_t x_ld[4], y_ld[2];
    int32_t batch, h, w, n_branches, dtype;
    int32_t y1_fold;
    int32_t n_convs;
    int32_t c0;                /* channels of branch 0: 32 (HRNet-W32; 0 means 32) or 48 (HRNet-W48: 48 / 96 / 192 / 384, the
                                  48-channel maps in 64-channel buffers, x_ld[0] = y_ld[0] = 64) */
    rsg_fm_conv convs[RSG_FM_MAX_CONVS];
} rsg_fuse_module_desc;
size_t rsg_fuse_module_table_bytes(void);
int rsg_fuse_module_plan(const rsg_fuse_module_desc* d, void* table_host, int32_t* launch2);
int rsg_fuse_module_run(const void* table_host, int32_t blocks, int32_t lds, int32_t dtype, void* stream);
/* what _plan decided (tests, logs): info4 = {band rows of branch 0, jobs, LDS offset of the last image, 0}; first17[w] ..
 * first17[w + 1]: wave w's jobs in jobs4 (96 x {conv, 32-channel tile, first 32-pixel group, groups}; conv 255 = the
 * barrier between the 1x1 convs whose results stay in LDS and everything that reads them) */
int rsg_fuse_module_table_jobs(const void* table_host, int32_t* info4, int32_t* first17, uint8_t* jobs4);

/* F.interpolate(scale_factor=2, mode='bilinear', align_corners=True) [+ sigmoid] on fp32 NCHW
 * (pose_rsgnet.py:1009-1013) */
int rsg_bilinear2x_nchw_f32(const float* in, float* out, int32_t planes, int32_t h, int32_t w,
                            int32_t sigmoid, void* stream);
/* Two maps of the same size in ONE launch (RSGNet's multi_kpt_scores and limbs_scores, pose_rsgnet.py:1009-1013: the second
 * through torch.sigmoid): map a's planes, then map b's. */
int rsg_bilinear2x_pair_nchw_f32(const float* in_a, float* out_a, int32_t planes_a, int32_t sigmoid_a, const float* in_b,
                                 float* out_b, int32_t planes_b, int32_t sigmoid_b, int32_t h, int32_t w, void* stream);
/* Backward of rsg_bilinear2x_nchw_f32 (training; autograd of F.interpolate(scale_factor=2, 'bilinear', align_corners=True)
 * [+ torch.sigmoid], /root/reference/lib/models/pose_rsgnet.py:1009-1013): din [planes,h,w] from dout [planes,2h,2w];
 * prob = the forward's sigmoid output (the gradient then passes through p (1 - p) first) or NULL.  Gather form, fixed order. */
int rsg_bilinear2x_backward_nchw_f32(const float* dout, const float* prob, float* din, int32_t planes, int32_t h,
                                     int32_t w, void* stream);

/* Flip-test merge on the device (lib/core/function.py:400-427 with lib/utils/transforms.py:23-37):
 * out = 0.5 * (h + shift(flip_back(h_flipped))), fp32 NCHW [batch,joints,height,width].  perm_host[j] is the
 * channel of h_flipped that lands on joint j (identity with the dataset's flip_pairs swapped); HOST
 * array.  shift_heatmap = cfg.TEST.SHIFT_HEATMAP. */
int rsg_flip_merge_nchw_f32(const float* h, const float* h_flipped, const int32_t* perm_host, float* out,
                            int32_t batch, int32_t joints, int32_t height, int32_t width,
                            int32_t shift_heatmap, void* stream);

/* get_final_preds on the device (lib/core/inference.py:52-82): per (crop, joint) argmax (first index on
 * ties), zero when max <= 0, +-0.25 px towards the larger neighbour when post_process, then the inverse
 * crop affine for rot = 0.  heat fp32 [batch,joints,height,width]; center, scale fp32 [batch,2] (as
 * produced by the datasets: scale in units of 200 px); preds fp32 [batch,joints,2] in image pixels;
 * maxvals fp32 [batch,joints]. */
int rsg_decode_heatmaps(const float* heat, const float* center, const float* scale, float* preds, float* maxvals,
                        int32_t batch, int32_t joints, int32_t height, int32_t width, int32_t post_process,
                        void* stream);

/* ---- TRP: SpatialRelationHead (association.py:280-301) ---------------------------------------- */
/* E[b,i,:] = sum_j sigmoid(<x_i,x_j>) * g_j   without materialising the S x S map.
 * x: [batch,S,ld_x] 16-bit (channels [0,C)), gT: [batch,C,S] 16-bit pair (hi, lo) of the projected
 * values g = conv1x1(x)+b split as g = hi + lo, E out: fp32 [batch,S,C].  C must be 32 or 48. */
int rsg_trp_attention(const void* x, int32_t ld_x, const void* gT_hi, const void* gT_lo, float* e_out,
                      int32_t batch, int32_t S, int32_t C, int32_t dtype, void* stream);
/* g projection (association.py:288-289) from x [batch,S,ld_x] with fp32 weights w[C][C], bias[C]: gT_hi[b,c,s] transposed,
 * the keys of every 16-block in the order [0-3, 8-11, 4-7, 12-15]; g_rows (or NULL): the same values as rows [batch,S,C]
 * (what rsg_trp_backward reads) */
int rsg_trp_project_g(const void* x, int32_t ld_x, const float* w, const float* bias, void* gT_hi,
                      void* g_rows, int32_t batch, int32_t S, int32_t C, int32_t dtype, void* stream);
/* W: conv1x1(+bias) then GroupNorm(8,C) on E, written 16-bit into out[b,s, 0:C] of a buffer with
 * channel stride out_ld (association.py:298-300).  workspace: rsg_trp_tail_workspace_bytes() of device
 * memory (the fp32 projection and the per-workgroup GroupNorm partials).  S % 256 == 0. */
size_t rsg_trp_tail_workspace_bytes(int32_t batch, int32_t S, int32_t C);
int rsg_trp_tail(const float* e, const float* w, const float* bias, const float* gn_gamma,
                 const float* gn_beta, void* out, int32_t out_ld, int32_t batch, int32_t S, int32_t C,
                 int32_t dtype, void* workspace, void* stream);
/* The same with the input form and the workspace's use as arguments: e16_ld = 0: e is fp32 [batch,S,C] (rsg_trp_tail), else
 * the 16-bit map of rsg_trp_tail_in16 with that channel stride; keep_y = 0 (inference): the fp32 projection y = W e + b is NOT
 * written to the workspace -- only rsg_trp_tail_backward reads it -- and the GroupNorm pass recomputes it from e (the same
 * expression, the same bits): 4 C bytes per position less written and read back. */
int rsg_trp_tail_ex(const void* e, int32_t e16_ld, const float* w, const float* bias, const float* gn_gamma,
                    const float* gn_beta, void* out, int32_t out_ld, int32_t batch, int32_t S, int32_t C,
                    int32_t dtype, int32_t keep_y, void* workspace, void* stream);
/* RELATION_SUB_SAMPLE variant of the TRP (/root/reference/lib/models/association.py:254-268,286-287; set by
 * experiments/crowdpose/hrnet/"rsgnet_w48_256x192_adam_lr1e-3 .yaml":42): the relation runs on the 2x2 max-pooled map
 * and an up-sampling deconv + BatchNorm + ReLU precedes W.
 *  rsg_maxpool2x2_nhwc16   nn.MaxPool2d((2,2)) on a 16-bit NHWC map in[b,y,x,0:c] (channel strides in_ld / out_ld)
 *  rsg_f32_rows_to_nhwc16  E fp32 [rows,c] -> 16-bit rows of stride out_ld (the deconv is an MFMA conv on 16-bit maps)
 *  rsg_trp_tail_in16       rsg_trp_tail whose input is the deconv's 16-bit NHWC output (stride e_ld) instead of fp32 E */
int rsg_maxpool2x2_nhwc16(const void* in, int32_t in_ld, void* out, int32_t out_ld, int32_t batch, int32_t h, int32_t w,
                          int32_t c, int32_t dtype, void* stream);
int rsg_f32_rows_to_nhwc16(const float* in, void* out, int32_t out_ld, int64_t rows, int32_t c, int32_t dtype, void* stream);
int rsg_trp_tail_in16(const void* e16, int32_t e_ld, const float* w, const float* bias, const float* gn_gamma,
                      const float* gn_beta, void* out, int32_t out_ld, int32_t batch, int32_t S, int32_t C,
                      int32_t dtype, void* workspace, void* stream);
/* Backward of rsg_trp_tail (training; autograd of W = conv1x1 + bias -> GroupNorm(8,C), association.py:298-300).
 * dout: gradient of the tail's 16-bit output, rows [batch,S,ld]; fwd_workspace: the workspace rsg_trp_tail filled in the
 * forward (y = W e + b in fp32 and the group partials).  Writes dy (gradient of y) and de = W^T dy, both fp32 [batch,S,C], and
 * `partials` (rsg_trp_tail_backward_partial_floats floats): [batch][S/256][2C] per-workgroup sums of dout*yhat (first C) and
 * dout (next C) -- dgamma / dbeta are their sums over the first two axes -- followed by the group sums the second launch
 * merges.  dW = dy^T e and dbias = sum dy are left to the caller's BLAS.  C = 32, S % 256 == 0, fixed summation order. */
size_t rsg_trp_tail_backward_partial_floats(int32_t batch, int32_t S, int32_t C);
int rsg_trp_tail_backward(const void* dout, int32_t ld, const void* fwd_workspace, const float* w, const float* gn_gamma,
                          float* dy, float* de, float* partials, int32_t batch, int32_t S, int32_t C, int32_t dtype,
                          void* stream);
/* TRP backward for training (autograd of /root/reference/lib/models/association.py:288-297 and of the relation MSE,
 * pose_rsgnet.py:1014-1018, which the reference leaves to torch on the materialised [B,S,S] map).  Inputs, all 16-bit of
 * `dtype`: x, g = conv1x1(x) + b and de = dL/dE as rows [batch,S,ld]; xT, deT transposed [batch,C,S] with the keys of
 * every 16-block in the order [0-3, 8-11, 4-7, 12-15] (rsg_trp_project_g's layout).  p [batch,S] fp32: factor of the
 * relation target (T = p p^T) with coef[b] = dL/dmse_b * 2 / S^2, or null without relation loss.  Outputs fp32
 * [batch,S,C]: dx = the gradient through the Gram product (add dg . Wg for the g path), dg = R . de.  C = 32, S % 128 == 0. */
int rsg_trp_backward(const void* x, int32_t ld_x, const void* g, int32_t ld_g, const void* de, int32_t ld_de,
                     const void* xT, const void* deT, const float* p, const float* coef, float* dx, float* dg,
                     int32_t batch, int32_t S, int32_t C, int32_t dtype, void* stream);
/* The same with the relation target given as the reference's training loop builds it (/root/reference/lib/core/function.py:
 * 261-272: a [batch,S,S] fp32 tensor, which is p p^T there): read tile by tile, 128 contiguous bytes per lane group and
 * accumulator register; a lane reads T_ji for T_ij, i.e. the target is taken as SYMMETRIC (for an asymmetric one
 * dA + dA^T would need (T_ij + T_ji) / 2).  rsg_trp_relation_mse_dense: the forward value, exact for any target. */
int rsg_trp_backward_dense(const void* x, int32_t ld_x, const void* g, int32_t ld_g, const void* de, int32_t ld_de,
                           const void* xT, const void* deT, const float* target, const float* coef, float* dx, float* dg,
                           int32_t batch, int32_t S, int32_t C, int32_t dtype, void* stream);
int rsg_trp_relation_mse_dense(const void* x, int32_t ld_x, const float* target, float* mse, void* workspace, int32_t batch,
                               int32_t S, int32_t C, int32_t dtype, void* stream);
/* fp32 row products of the training head (autograd of the TRP's 1x1 projections, association.py:288-289, 298), fixed
 * summation order.  rsg_rows_project_f32: out[n,j] (+)= sum_k a[n,k] m[k,j] (C = 32 or 48; m is [C,C]).
 * rsg_rows_gram_f32: gram[i,j] = sum_n a[n,i] b[n,j] and, when colsum is not NULL, colsum[i] = sum_n a[n,i] (C = 32);
 * workspace: rsg_rows_gram_workspace_bytes(C) bytes of device memory. */
int rsg_rows_project_f32(const float* a, const float* m, float* out, int64_t rows, int32_t C, int32_t accumulate, void* stream);
size_t rsg_rows_gram_workspace_bytes(int32_t C);
int rsg_rows_gram_f32(const float* a, const float* b, float* gram, float* colsum, int64_t rows, int32_t C, void* workspace,
                      void* stream);
/* SGM graph convolution = KTMachine.forward (/root/reference/lib/models/pose_rsgnet.py:576-600; constructor :499-521):
 *   K = matrix_limb (.) real_matrix_limb [limbs,joints];  T = K . wk  (wk = final_layer.weight as [joints,channels]);
 *   wl = Linear2(LeakyReLU_slope(Linear1(T)))  [limbs,channels]  -- the limb classifier's weight of :1003-1005.
 * Everything fp32, row-major, on the device; w1 / w2 in torch's Linear layout [out,in]; limbs, joints <= 32, channels <= 64.
 * t_saved / h_saved [limbs,channels]: T and Linear1's output before the LeakyReLU, written by the forward and read by the
 * backward (training).  One workgroup; every sum in index order (bit-reproducible).  The backward writes the gradients of
 * matrix_limb (masked by the prior, as autograd's), wk, and the two Linears from d_wl [limbs,channels]. */
int rsg_sgm_graph_conv(const float* matrix_limb, const float* prior, const float* wk, const float* w1, const float* b1,
                       const float* w2, const float* b2, float* wl, float* t_saved, float* h_saved, int32_t limbs,
                       int32_t joints, int32_t channels, float slope, void* stream);
int rsg_sgm_graph_conv_backward(const float* matrix_limb, const float* prior, const float* wk, const float* w1,
                                const float* w2, const float* t_saved, const float* h_saved, const float* d_wl,
                                float* d_matrix_limb, float* d_wk, float* d_w1, float* d_b1, float* d_w2, float* d_b2,
                                int32_t limbs, int32_t joints, int32_t channels, float slope, void* stream);
/* optional materialisation of relation_score [batch,S,S] fp32 (association.py:294-295) */
int rsg_trp_scores(const void* x, int32_t ld_x, float* scores, int32_t batch, int32_t S, int32_t C,
                   int32_t dtype, void* stream);
/* training tail (pose_rsgnet.py:1014-1018) with the target kept factored, T = p p^T:
 * mse[b] = mean_ij (p_i p_j - sigmoid(<x_i,x_j>))^2 ; p: fp32 [batch,S].  mse is overwritten (no pre-zeroing): every wave
 * writes one partial sum into `workspace` (rsg_trp_mse_workspace_bytes(batch, S) bytes of device memory) and a second
 * launch adds an image's partials in index order -- the same bits on every run, no atomics. */
size_t rsg_trp_mse_workspace_bytes(int32_t batch, int32_t S);
int rsg_trp_relation_mse(const void* x, int32_t ld_x, const float* p, float* mse, void* workspace, int32_t batch,
                         int32_t S, int32_t C, int32_t dtype, void* stream);

/* ---- training tail ------------------------------------------------------------
 * Fused loss value + gradient kernels for the reference's training step (lib/core/function.py:246-305).
 * All tensors fp32, contiguous, 16-byte aligned; loss: one device float, overwritten or (accumulate != 0)
 * added to; grad may be NULL (value only).  workspace: rsg_loss_workspace_bytes(batch*joints) bytes. */
size_t rsg_loss_workspace_bytes(int32_t rows);
/* JointsMSELoss (lib/core/loss.py:14-38): loss = mean_j 0.5 * mean_{b,p} ((pred - gt) * w[b,j])^2 ; weight NULL =
 * use_target_weight False; grad = grad_scale * dloss/dpred. */
int rsg_joints_mse_loss(const float* pred, const float* gt, const float* weight, int32_t batch, int32_t joints,
                        int32_t pixels, float grad_scale, float* loss, int32_t accumulate, float* grad,
                        void* workspace, void* stream);
/* loss_scale * torch.nn.BCELoss()(prob, target) (function.py:240,293 uses 0.01) and its gradient. */
int rsg_bce_loss(const float* prob, const float* target, int64_t n, float loss_scale, float grad_scale, float* loss,
                 int32_t accumulate, float* grad, void* workspace, void* stream);
/* p[b, :] of the relation target p p^T (function.py:261-268): max over joints, bilinear x0.5, align_corners=True;
 * feeds rsg_trp_relation_mse. target [batch,joints,h,w] -> out [batch,(h/2)*(w/2)]. */
int rsg_relation_target_vec(const float* target, float* out, int32_t batch, int32_t joints, int32_t h, int32_t w,
                            void* stream);
/* torch.optim.Adam(lr) (lib/utils/utils.py:70-74): one step on flat fp32 buffers; step counts from 1;
 * grad_scale folds the data-parallel 1/world average into the update. */
int rsg_adam_step(float* param, const float* grad, float* exp_avg, float* exp_avg_sq, int64_t n, float lr,
                  float beta1, float beta2, float eps, int32_t step, float grad_scale, void* stream);

/* ---- training step: BatchNorm with batch statistics, backward kernels -------------------
 * `model.train()` forward + backward of pose_hrnet (lib/core/function.py:27-80 drives the reference; its arithmetic is
 * torch autograd over pose_hrnet.py).  Maps are dense NHWC 16-bit; statistics, affine parameters, gradients fp32.
 * Convolutions themselves run through rsg_conv_igemm (forward: weights packed WITHOUT BatchNorm folding; data
 * gradient: the same kernel with the transposed, flipped weights, stride-2 convs behind rsg_zero_insert2x).
 * Every reduction has a fixed order: no float atomics, bit-reproducible gradients. */
/* One launch packs every conv weight of the model from its fp32 master copy (device) into the MFMA fragment order of
 * rsg_pack_conv_weights_host (no scale/shift): item k owns blocks [first_block, next item's first_block) of
 * rsg_pack_item_blocks(packed cin, packed cout, kh, kw) 256-thread blocks.  transpose = 1: the data-gradient form
 * W'[i][o][ky][kx] = W[o][i][kh-1-ky][kw-1-kx] (packed cin = cout of W, packed cout = cin of W). */
typedef struct rsg_pack_item {
    const float* w;            /* [cout][cin][kh][kw] fp32, device */
    void* dst;                 /* rsg_conv_packed_bytes(packed cin, packed cout, kh, kw) bytes, device */
    int32_t cin, cout, kh, kw; /* of the weight tensor as stored */
    int32_t transpose;
    int32_t first_block;
} rsg_pack_item;
int32_t rsg_pack_item_blocks(int32_t cin, int32_t cout, int32_t kh, int32_t kw);
int rsg_pack_conv_weights_device(const rsg_pack_item* items_device, int32_t n_items, int32_t total_blocks,
                                 int32_t dtype, void* stream);
/* nn.BatchNorm2d in training mode (pose_hrnet.py:37 etc.; momentum 0.1, eps 1e-5): per-channel mean and
 * 1/sqrt(biased var + eps) of y [pixels][c], running statistics updated in place (unbiased variance; NULL = skip).
 * c: a power of two in [8, 256].  workspace: rsg_bn_workspace_bytes(c). */
size_t rsg_bn_workspace_bytes(int32_t c);
int rsg_bn_stats(const void* y, int64_t pixels, int32_t c, int32_t dtype, float* workspace, float* mean, float* invstd,
                 float* running_mean, float* running_var, float momentum, float eps, void* stream);
/* out = [ReLU]( gamma * (y - mean) * invstd + beta [+ residual] ) */
int rsg_bn_apply(const void* y, const float* mean, const float* invstd, const float* gamma, const float* beta,
                 const void* residual, int32_t relu, void* out, int64_t pixels, int32_t c, int32_t dtype, void* stream);
/* Backward of the unit above.  dz: gradient of the unit's output; z: that output (ReLU mask; NULL = the unit had no
 * ReLU); y: the conv output the BatchNorm normalised.  Writes dy (gradient of y), dgamma, dbeta and, when dres is
 * not NULL, the masked gradient g = dz * (z > 0), which is the gradient of the residual operand. */
int rsg_bn_backward(const void* dz, const void* z, const void* y, const float* mean, const float* invstd,
                    const float* gamma, void* dy, void* dres, float* dgamma, float* dbeta, float* workspace,
                    int64_t pixels, int32_t c, int32_t dtype, void* stream);
/* Up to 8 INDEPENDENT BatchNorms (the branches of a HighResolutionModule, pose_hrnet.py:254-255) as three launches
 * instead of three each.  Forward: fields y, gamma, beta, residual (or NULL), relu, out, mean, invstd, running_* (or
 * NULL), workspace are read as by rsg_bn_stats + rsg_bn_apply.  Backward: y, dz, z (or NULL), mean, invstd, gamma,
 * out (= dy), dres (or NULL), dgamma, dbeta, workspace as by rsg_bn_backward.  Every item needs a workspace of its
 * own (rsg_bn_workspace_bytes).  Same arithmetic, same fixed reduction order as the single-map entry points. */
typedef struct rsg_bn_item {
    const void* y; const void* dz; const void* z; const void* residual;
    void* out; void* dres;
    const float* gamma; const float* beta;
    float* mean; float* invstd; float* running_mean; float* running_var; float* dgamma; float* dbeta;
    float* workspace;
    int64_t pixels;
    int32_t c, relu;
} rsg_bn_item;
int rsg_bn_forward_group(const rsg_bn_item* items, int32_t n, float momentum, float eps, int32_t dtype, void* stream);
int rsg_bn_backward_group(const rsg_bn_item* items, int32_t n, int32_t dtype, void* stream);
/* g = dout * (out > 0): the ReLU that ends a fuse sum (pose_hrnet.py:266);  out = a + b: a map with two consumers */
int rsg_relu_backward(const void* dout, const void* out, void* g, int64_t n, int32_t dtype, void* stream);
int rsg_add16(const void* a, const void* b, void* out, int64_t n, int32_t dtype, void* stream);
/* backward of nn.Upsample(scale_factor = 2^shift, 'nearest') (pose_hrnet.py:213): grad_lo [batch,h_lo,w_lo,c] =
 * sum of the 2^shift x 2^shift block of grad_hi [batch, h_lo << shift, w_lo << shift, c] */
int rsg_upsample_nearest_backward(const void* grad_hi, void* grad_lo, int32_t batch, int32_t h_lo, int32_t w_lo,
                                  int32_t c, int32_t shift, int32_t dtype, void* stream);
/* out [batch, 2h, 2w, c] = in [batch, h, w, c] on the even positions, zeros elsewhere (data gradient of stride-2 convs) */
int rsg_zero_insert2x(const void* in, void* out, int32_t batch, int32_t h, int32_t w, int32_t c, void* stream);
/* out[c] = sum over batch and pixels of x [batch, c, hw] fp32 (bias gradient of the classifier conv); workspace:
 * rsg_channel_sum_workspace_bytes(c) bytes, 8-byte aligned */
size_t rsg_channel_sum_workspace_bytes(int32_t c);
int rsg_channel_sum_nchw_f32(const float* x, float* out, void* workspace, int32_t batch, int32_t c, int32_t hw, void* stream);
/* Weight gradient of nn.Conv2d (3x3 pad 1 or 1x1 pad 0, stride 1 or 2) on the matrix cores:
 *   dw[o][i][ky][kx] (fp32, nn.Conv2d layout, cout x cin real channels) = sum dy[b,r,c,o] * x[b, r*s+ky-p, c*s+kx-p, i]
 * x: [batch,in_h,in_w,cin_pad], dy: [batch,out_h,out_w,cout_pad] dense NHWC 16-bit (padding channels are ignored).
 * workspace: rsg_conv_wgrad_workspace_bytes(...) bytes (0 = unsupported geometry). */
size_t rsg_conv_wgrad_workspace_bytes(int32_t batch, int32_t in_h, int32_t in_w, int32_t cin_pad, int32_t out_h,
                                      int32_t out_w, int32_t cout_pad, int32_t kh, int32_t kw, int32_t stride, int32_t pad);
int rsg_conv_wgrad(const void* x, const void* dy, float* dw, void* workspace, int32_t batch, int32_t in_h, int32_t in_w,
                   int32_t cin, int32_t cin_pad, int32_t out_h, int32_t out_w, int32_t cout, int32_t cout_pad, int32_t kh,
                   int32_t kw, int32_t stride, int32_t pad, int32_t dtype, void* stream);

/* Up to 4 independent weight gradients of the same kernel size in one launch pair (the same conv position of a
 * HighResolutionModule's branches); fields as the arguments of rsg_conv_wgrad, a workspace per item. */
typedef struct rsg_wgrad_item {
    const void* x; const void* dy; float* dw; void* workspace;
    int32_t batch, in_h, in_w, cin, cin_pad, out_h, out_w, cout, cout_pad, kh, kw, stride, pad, reserved;
} rsg_wgrad_item;
int rsg_conv_wgrad_group(const rsg_wgrad_item* items, int32_t n, int32_t dtype, void* stream);
/* Pipelined form: one launch = the main kernel of `items` + (in its first workgroups) the partial-tile reduction of `prev`,
 * a group whose main kernel ran in an earlier call on this stream with defer_reduce = 1.  defer_reduce = 0 appends this
 * group's own reduction as a second launch (= rsg_conv_wgrad_group); n = 0 runs only prev's reduction.  The workspaces of
 * `items` and `prev` must be disjoint (they are in use at the same time). */
int rsg_conv_wgrad_group_ex(const rsg_wgrad_item* items, int32_t n, const rsg_wgrad_item* prev, int32_t n_prev,
                            int32_t defer_reduce, int32_t dtype, void* stream);
/* experiment switch (process-wide): 0 = every weight gradient in the one-tile-pair-per-workgroup form */
void rsg_wgrad_set_quad(int32_t on);
/* slice policy of the weight gradient: at least `min_px_per_slice` output pixels per slice (0 = no bound), and the 2x2
 * form only where it still launches `quad_min_wgs` workgroups (else one tile pair per workgroup); lds_budget_bytes: LDS per
 * workgroup (16..144 KB; sets the chunk size and how many workgroups share a compute unit; other values keep the current one); chunk_px_cap: output pixels
 * of a chunk (64..4096, else unchanged); waves: 4 or 8 waves per workgroup of the 2x2 form (else unchanged) */
void rsg_wgrad_tune(int32_t min_px_per_slice, int32_t quad_min_wgs, int32_t lds_budget_bytes, int32_t chunk_px_cap, int32_t waves);

/* ---- crop producer ---------------------------------------------------------------
 * One person box per entry: `cv2.warpAffine(img, trans, (out_w, out_h), flags=INTER_LINEAR)` (constant border 0)
 * followed by ToTensor + Normalize (lib/dataset/CPJointsDataset.py:1281-1290; JointsDataset the same;
 * tools/test.py:105-112), OpenCV's fixed-point bilinear arithmetic restated bit for bit.  The descriptor array
 * lives in DEVICE memory; `image` points at a device uint8 picture, HWC with 3 interleaved channels. */
struct rsg_crop_desc {
    const void* image;
    int32_t height, width;   /* source picture */
    int32_t stride;          /* bytes per source row */
    int32_t reserved;
    double m[6];             /* forward 2x3 map source -> crop, row-major (get_affine_transform, transforms.py:64) */
};
/* out: fp32 [batch,3,out_h,out_w]; out_flipped (may be NULL): the same crops mirrored left-right, i.e. `input.flip(3)`
 * of the flip test (lib/core/function.py:400-402), written in the same pass; mean3/std3: HOST arrays of 3 (per output
 * channel); swap_rb != 0 reverses the channel order on the way (cv2.COLOR_BGR2RGB, DATASET.COLOR_RGB). */
int rsg_crop_warp_affine(const struct rsg_crop_desc* crops_dev, float* out_nchw, float* out_flipped_nchw, int32_t batch,
                         int32_t out_h, int32_t out_w, const float* mean3_host, const float* std3_host, int32_t swap_rb,
                         void* stream);

#ifdef __cplusplus
}
#endif
#endif /* RSGNET_HIP_H */
