// LDS-resident chain of BasicBlocks for gfx950 (MI355X).
//
// Replaces  x = branches[i](x)  of HighResolutionModule.forward
// (/root/reference/lib/models/pose_hrnet.py:254-255; BasicBlock :30-59): L = 2*nblocks 3x3 convs C->C
// with BatchNorm folded, ReLU after every conv and the block input added before the second ReLU.
//
// The per-conv kernel (conv_igemm.hip) pays an HBM round trip per conv: ~38 MB moved for 3.6 GFLOP.
// Here a workgroup keeps its activation tile in ONE LDS image for the whole chain:
//   * whole-image mode (rows == H): `ni` complete images per workgroup, no halo, any chain length
//     (64x24x32 ... 8x6x256 maps of HRNet branches 1-3 fit: 127 / 69 / 84 KB);
//   * row-tile mode: `rows` output rows + L halo rows each side (branch 0: 64x48x32 does not fit), the
//     outer rows are recomputed by the neighbouring tile.
// Each layer runs IN PLACE: all waves accumulate their output tiles from the image (k-loop), barrier,
// write the 16-bit results back over it, barrier.  A wave owns the same (32 couts x 32 pixels) tiles in
// every layer, so the block input it must add back is private to each lane.  It is NOT parked in registers
// across the two k-loops (48-56 VGPRs: the compiler spilled them to scratch and reloaded them at the point of
// use, +3..7k cycles on every second layer): it waits in a per-lane slot of a global workspace, written from the
// staged image (first block) or by the previous block's epilogue, and is fetched one k-step before the k-loop
// ends.  Rows outside the image are re-zeroed after every layer (zero padding).
// MFMA mapping, weight packing and LDS pixel layout are those of conv_igemm.hip.
#include "common.h"
#include <algorithm>
#include <type_traits>

#define RSG_CHAIN_MAX_LAYERS 8
// pixel groups of the grouped launch's chains whose block input waits in registers (chain_body NHELD) instead of the
// global workspace; -DRSG_CHAIN_HELD64=0 -DRSG_CHAIN_HELD32=0 builds the all-workspace form for a same-visit A/B
#ifndef RSG_CHAIN_HELD64
#define RSG_CHAIN_HELD64 6
#endif
#ifndef RSG_CHAIN_HELD32_NT6
#define RSG_CHAIN_HELD32_NT6 6     // the 32-row tiles (six pixel groups per wave): all of them
#endif
#ifndef RSG_CHAIN_HELD32
#define RSG_CHAIN_HELD32 4
#endif
#ifndef RSG_CHAIN_NPF32
#define RSG_CHAIN_NPF32 6      // weight-fragment ring depth of the C = 32 chains (3 leaves room for a 5th held group)
#endif

// Diagnostic build (-DRSG_CHAIN_STAMPS): wave 0 of every workgroup records shader-clock stamps at the phase
// boundaries into a side buffer that nothing else reads (tools/chain_stamps.py prints the shares).  The
// production build contains none of this.
#ifdef RSG_CHAIN_STAMPS
__device__ unsigned long long* g_chain_stamps = nullptr;   // [blocks][16 waves][64]
#define RSG_STAMP(slot)                                                                              \
    do {                                                                                             \
        if (g_chain_stamps && (threadIdx.x & 63) == 0) {                                             \
            unsigned long long t__;                                                                  \
            asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t__)::"memory");              \
            g_chain_stamps[((size_t)blockIdx.x * 16 + (threadIdx.x >> 6)) * 64 + (slot)] = t__;       \
        }                                                                                            \
    } while (0)
#else
#define RSG_STAMP(slot) do {} while (0)
#endif

// Sticky error word: polls that gave up (a producer tile never signalled within the bound).  The results of that
// launch are then wrong.  The production path reads it back (rsg_chain_handoff_timeouts / _async: rsgnet_amd/plan.py
// polls it without stalling the stream, re-zeroes the plan's flags and raises).  Forward progress does NOT rest on
// the dispatch order: rsg_basic_chain_group refuses a launch whose waiting workgroups could fill the machine (see
// there), so a producer always finds a CU; the bound only turns a bug into an error instead of a hung GPU.
__device__ unsigned g_chain_handoff_timeouts = 0;

struct ChainArgs {
    const char* in;
    char* out;
    const uint4* wpk[RSG_CHAIN_MAX_LAYERS];
    const float* bias[RSG_CHAIN_MAX_LAYERS];
    int nlayers;
    int B, H, W;
    int TH;       // output rows per tile
    int R;        // rows held per image slot: TH + 2*halo, TH + halo with two tiles, H with one
    int halo;     // = nlayers in row-tile mode, 0 in whole-image mode
    int NI;       // images per workgroup (whole-image mode), else 1
    int tiles_y;
    int RP;       // LDS row pitch in bytes (host: chain_row_pitch)
    uint4* ws;    // block inputs: [workgroup][NT*2][threads] x 16 B
    // hand-off inside one grouped launch (row-tiled branch cut in two chains): a workgroup sets its three words
    // done[3*index + d] after its stores, one for each chain tile that reads it (the tile above, the same tile,
    // the tile below); a dependent chain polls the words addressed to it before it stages and clears them.
    // Both null for an ordinary chain.
    int* done;
    const int* wait;
    int dual;     // 1: two LDS images (ping-pong): a layer reads one and writes the other, the block input stays in LDS
    int ldi, ldo; // pixel stride (elements) of in / out; == C except for the 48-channel maps of HRNet-W48 (kept in 64-channel buffers)
    // halo exchange inside the launch (rsg_chain_item.exchange, chain16 bodies): a row-tiled chain of 2 k layers holds a k-row
    // halo; after layer k every tile publishes the rows its neighbours lack (xbuf, written through), raises a word of xflag for
    // each of them, waits for theirs and replaces its stale halo rows -- ONE staging and ONE final store for all 2 k layers
    // where two k-layer chains handing over through memory stage and store twice.  All null / 0 for an ordinary chain.
    char* xbuf;   // [workgroup][2 sides][halo rows][W][C] 16-bit
    int* xflag;   // [workgroup][3]: word 0 = "my rows for the tile above are in xbuf", word 1 = "... for the tile below"
    int xchg;     // the layer in front of which the exchange happens (k), 0 = none
    int debug;    // rsg_tune("chain_debug"): bit 4 = fault injection, producers never signal (the error word's test); -DRSG_DEBUG builds only: bit 2 = no block-input fetch (timing experiment, WRONG results)
};

// NHELD: the block input of the first NHELD pixel groups waits in REGISTERS instead of the global workspace (in-place
// mode only): each lane reads its own pixels of the block input from the LDS image just before it overwrites them with
// the first conv's output and keeps them (8 vector registers per group) across the second conv's k-loop.  As many
// groups as the NT*16 accumulators leave room for: all 6 at C = 64 (96 + 48 of 256 registers), some of the 7 at C = 32.
template <class DT, int C, int WM, int WN, int NT, bool DUAL = false, int NHELD = 0>
__device__ __forceinline__ void chain_body(const ChainArgs& p, const int block_id, char* smem, const unsigned pf = 0) {
    static_assert(!(DUAL && NHELD) && NHELD <= NT, "ping-pong images keep the block input in LDS already");
    constexpr int NTHR = 64 * WM * WN;
    constexpr int CINB = (C % 64 == 0) ? 64 : (C % 32 == 0 ? 32 : 16);   // same k order as rsg_pack_conv_weights_host
    // C = 48 (HRNet-W48's first branch): the output channels are padded to two 32-channel tiles (zero weight rows);
    // the lanes of the padding own no channels of the LDS pixel and write nothing.  Its maps live in buffers with a
    // pixel stride of their own (ldi / ldo); every other width is dense.
    constexpr bool PADDED = (C % 32) != 0;
    static_assert(C % 16 == 0 && 32 * WM >= C && 32 * WM - C < 32, "one wave row per 32 output channels");
    constexpr int NCH = C / CINB, KPT = CINB / 16;
    constexpr int KS = NCH * 9 * KPT;
    constexpr int PS = 2 * C + 16;                       // LDS bytes per pixel
    constexpr int CPP = C / 8;                           // 16-byte chunks per pixel

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave / WN, wn = wave % WN;
    const int n = lane & 31, hh = lane >> 5;
    const int W = p.W, R = p.R;
    // LDS row pitch: one zero pixel left and right, padded so that RP == W*PS (mod 256 B).  A 32-pixel
    // group that wraps to the next row then continues the same odd-slot bank progression (conflict-free
    // ds_read_b128); the dense pitch cost 43 % of the LDS cycles in bank conflicts (SQ_LDS_BANK_CONFLICT)
    const int RP = p.RP;
    const int slot_rows = R + 2;                         // + one zero row above and below

    const int tile = block_id % p.tiles_y;
    const int b0 = (block_id / p.tiles_y) * p.NI;      // first image of this workgroup
    // image row held in slot row 0 (+1 border).  Tiles at the top/bottom of the picture need no halo on
    // that side: the window is clamped into the picture, so two 32-row tiles of a 64-row map hold 36 rows
    // each instead of 40 and every held row is a real one.
    const int y_first = min(max(tile * p.TH - p.halo, 0), p.H - R);

    RSG_STAMP(0);
#ifdef RSG_CHAIN_STAMPS
    if (g_chain_stamps && (threadIdx.x & 63) == 0) {         // which XCD (clock domain) and CU this workgroup runs on
        unsigned xcc, hwid;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)\n\ts_getreg_b32 %1, hwreg(HW_REG_HW_ID)" : "=s"(xcc), "=s"(hwid));
        g_chain_stamps[((size_t)blockIdx.x * 16 + (threadIdx.x >> 6)) * 64 + 63] = ((unsigned long long)xcc << 32) | hwid;
    }
#endif
    // ---- zero what staging does not write: the border row above and below every image slot and the border pixel
    // left and right of every row (the zero padding of each conv; the epilogues never touch them).  Every held row
    // lies inside the picture (clamped window), so the interior is overwritten by the staging below; only an image
    // slot past the end of the batch is zeroed as a whole.
    const int lds_bytes = p.NI * slot_rows * RP;
    // ping-pong mode is a separate instantiation: as a run-time switch inside one body it made the in-place path of the
    // same kernel give different results from run to run (two barriers, one of them conditional, in one unswitched loop)
    constexpr bool dual = DUAL;
    {
        const int row_chunks = RP / 16;                  // RP is a multiple of 16
        const int per_slot = 2 * row_chunks + R * 2 * CPP;
        for (int id = tid; id < p.NI * per_slot; id += NTHR) {
            const int i = id / per_slot, e = id - i * per_slot;
            int off;
            if (e < 2 * row_chunks) off = (e < row_chunks ? 0 : (R + 1) * RP) + (e % row_chunks) * 16;
            else {
                const int q = e - 2 * row_chunks, r = q / (2 * CPP), c = q - r * (2 * CPP);
                off = (r + 1) * RP + (c < CPP ? 0 : (W + 1) * PS) + (c % CPP) * 16;
            }
            *reinterpret_cast<uint4*>(smem + i * slot_rows * RP + off) = make_uint4(0, 0, 0, 0);
            if (dual) *reinterpret_cast<uint4*>(smem + lds_bytes + i * slot_rows * RP + off) = make_uint4(0, 0, 0, 0);
        }
        for (int i = 0; i < p.NI; ++i)
            if (b0 + i >= p.B)
                for (int o = tid * 16; o < slot_rows * RP; o += NTHR * 16) {
                    *reinterpret_cast<uint4*>(smem + i * slot_rows * RP + o) = make_uint4(0, 0, 0, 0);
                    if (dual) *reinterpret_cast<uint4*>(smem + lds_bytes + i * slot_rows * RP + o) = make_uint4(0, 0, 0, 0);
                }
    }
    RSG_STAMP(1);
    if (p.wait) {
        // producer tiles this window overlaps: tile-1 .. tile+1 of the same image.  One lane per flag polls with
        // sc1 loads (L2, never this CU's L1); the barrier below releases the other waves (MI355X_MICROARCH.md,
        // inter-workgroup visibility: sc1 stores + sc1 flag on the producer, sc1 poll + sc1 loads here, no fences).
        // The poll is bounded: a producer that never signals costs a wrong answer, not a hung GPU.
        // Every (producer tile, consumer tile) pair has a flag word of its own, so the consumer can clear it once
        // it has seen it: the flags are back at zero when the launch ends and need no reset between launches.
        const int t_lo = max(tile - 1, 0), t_hi = min(tile + 1, p.tiles_y - 1);
        if (tid <= t_hi - t_lo) {
            const int tp = t_lo + tid;                                  // producer tile
            int* f = const_cast<int*>(p.wait) + (((block_id / p.tiles_y) * p.tiles_y + tp) * 3 + (tile - tp + 1));
            int spin = 0;
            for (; spin < (1 << 20) && __hip_atomic_load(f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0; ++spin)
                __builtin_amdgcn_s_sleep(8);
            // a flag that was never seen is NOT cleared (a late producer would set it behind the clear and the stale 1
            // would let every later replay through without waiting): the host re-zeroes all flags when it finds the
            // error word raised
            if (spin == (1 << 20)) atomicAdd(&g_chain_handoff_timeouts, 1u);
            else __hip_atomic_store(f, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        __syncthreads();
    }
    const int chunks_per_img = R * W * CPP;
    constexpr int NLD = 14;                              // staging loads in flight per thread (one round for every HRNet-W32 map)
    // sc1 loads (aux 16): served by L2, so a tile written by another workgroup of this launch is read fresh
    const int ldi2 = PADDED ? p.ldi * 2 : C * 2, ldo2 = PADDED ? p.ldo * 2 : C * 2;     // pixel strides in bytes
    const __amdgpu_buffer_rsrc_t in_rsrc = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<char*>(p.in), 0, p.B * p.H * W * ldi2, 0x00020000);
    // The held rows of an image are one contiguous byte range of the input, so chunk k of the window is at
    // start + 16k; only its LDS address needs (row, position in the row).  The row is walked incrementally: a
    // division by the run-time row length per chunk cost more cycles than the loads themselves.
    const int rowc = W * CPP;                            // 16-byte chunks per image row
    const int step_q = NTHR / rowc, step_r = NTHR - step_q * rowc;
    for (int i = 0; i < p.NI; ++i) {
        if (b0 + i >= p.B) break;
        const int gbase = ((b0 + i) * p.H + y_first) * W * ldi2;
        char* const slot = smem + i * slot_rows * RP;
        int r = tid / rowc, c = tid - r * rowc;
        for (int base = 0; base < chunks_per_img; base += NTHR * NLD) {
            uint4 v[NLD];
            int lo[NLD];
#pragma unroll
            for (int j = 0; j < NLD; ++j) {
                lo[j] = -1;
                if (r < R) {
                    v[j] = __builtin_bit_cast(uint4, __builtin_amdgcn_raw_buffer_load_b128(
                        in_rsrc, gbase + (PADDED ? (r * W + c / CPP) * ldi2 + (c % CPP) * 16 : (base + j * NTHR + tid) * 16), 0, 16));
                    lo[j] = (r + 1) * RP + PS + (c / CPP) * PS + (c % CPP) * 16;
                }
                r += step_q; c += step_r;
                if (c >= rowc) { c -= rowc; ++r; }
            }
#pragma unroll
            for (int j = 0; j < NLD; ++j)
                if (lo[j] >= 0) *reinterpret_cast<uint4*>(slot + lo[j]) = v[j];
        }
    }

    // folded-BN biases of every layer, behind the image (read back at each layer top: LDS, not L2, latency)
    float* sbias = reinterpret_cast<float*>(smem + (dual ? 2 : 1) * lds_bytes);
    for (int l = 0; l < p.nlayers; ++l)
        for (int c = tid; c < C; c += NTHR) sbias[l * C + c] = p.bias[l][c];

    // ---- this lane's pixels ---------------------------------------------------------------------------
    const int cbase = wm * 32 + hh * 16;                 // this lane's 16 output channels
    int pixoff[NT];     // LDS byte address of the pixel (centre tap), + lane-half offset
    bool live[NT];      // pixel exists in this workgroup's tile
    const int npix = p.NI * R * W;
#pragma unroll
    for (int g = 0; g < NT; ++g) {
        const int q = (wn * NT + g) * 32 + n;
        live[g] = q < npix;
        const int qq = live[g] ? q : 0;
        const int i = qq / (R * W), rem = qq - i * (R * W);
        const int r = rem / W, x = rem - r * W;
        const int y = y_first + r;
        pixoff[g] = (i * slot_rows + r + 1) * RP + (x + 1) * PS + hh * 16;
    }
    __syncthreads();
    asm volatile("" ::"v"(pf));                           // the grouped launch's weight prefetch loads end here
    RSG_STAMP(2);

    // ring depth; must divide KS (18, 36, 72, 144).  C = 128 with six pixel groups per wave has 6 x 128 = 768 cycles of
    // MFMA per k-step: a 6-deep ring covers an L2 miss as well as 9 did with three groups, and frees 12 registers
    constexpr int NPF = (C == 256) ? 9 : (C == 128 ? (NT >= 6 ? 6 : 9) : (C == 64 ? 4 : (C == 32 ? RSG_CHAIN_NPF32 : (C == 48 || C == 96 ? 3 : 6))));
    static_assert(KS % NPF == 0, "prefetch ring must wrap at the layer boundary");
    uint4 aq[NPF];
    // block-input slots of this lane, through a buffer descriptor: uniform base in SGPRs, ONE address VGPR (the
    // lane offset) and a constant per (pixel group, half).  Plain pointers made hipcc hoist 2*NT 64-bit
    // addresses out of the layer loop and spill them.
    const __amdgpu_buffer_rsrc_t ws_rsrc = __builtin_amdgcn_make_buffer_rsrc(
        reinterpret_cast<char*>(p.ws) + (size_t)block_id * (NT * 2 * NTHR * 16), 0, NT * 2 * NTHR * 16, 0x00020000);
    const int ws_lane = tid * 16;
    auto ws_load = [&](int g, int k) {
        return __builtin_bit_cast(uint4, __builtin_amdgcn_raw_buffer_load_b128(ws_rsrc, ws_lane, (g * 2 + k) * NTHR * 16, 0));
    };
    auto ws_store = [&](int g, int k, const uint4& v) {
        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v), ws_rsrc, ws_lane, (g * 2 + k) * NTHR * 16, 0);
    };
    const int own_delta = cbase * 2 - hh * 16;           // pixoff -> this lane's 16 channels of the pixel
    uint4 held[NHELD ? NHELD : 1][2];                    // the block input of the block in flight, groups [0, NHELD)
    if (!dual) {
#pragma unroll
        for (int g = NHELD; g < NT; ++g) {               // the first block's input, from the staged image
            ws_store(g, 0, *reinterpret_cast<const uint4*>(smem + pixoff[g] + own_delta));
            ws_store(g, 1, *reinterpret_cast<const uint4*>(smem + pixoff[g] + own_delta + 16));
        }
    }
    for (int l = 0; l < p.nlayers; ++l) {
        // ping-pong mode: layer l reads image (l & 1) and writes the other one -- which still holds this block's input
        // at an odd l, read back by its owner lane just before it is overwritten.  No workspace, one barrier per layer.
        const char* const rd = smem + (dual ? (l & 1) * lds_bytes : 0);
        char* const wr = smem + (dual ? ((l + 1) & 1) * lds_bytes : 0);
        uint4 res[NT][2];                               // block input at this lane's pixels/channels (odd layers)
        // fetched in two parts: the first NRE pixel groups one k-step before the k-loop ends (all the registers
        // the widest variant has left there), the others right after it; the epilogue consumes them in order
#ifdef RSG_DBG_LATE_RES
        constexpr int NRE = 0;
#else
        // 768-thread variants (HRNet-W48 widths, 170 registers per lane): everything after the k-loop
        constexpr int NRE = NTHR > 512 ? 0 : (NT >= 7 ? 3 : NT);
#endif
        auto fetch_res = [&](int g0, int g1) {
#ifdef RSG_DEBUG
            if (p.debug & 2) {                           // timing experiment only (diagnostic build)
#pragma unroll
                for (int g = g0; g < g1; ++g) res[g][0] = res[g][1] = make_uint4(0, 0, 0, 0);
                return;
            }
#endif
            if (dual) {
#pragma unroll
                for (int g = g0; g < g1; ++g) {
                    res[g][0] = *reinterpret_cast<const uint4*>(wr + pixoff[g] + own_delta);
                    res[g][1] = *reinterpret_cast<const uint4*>(wr + pixoff[g] + own_delta + 16);
                }
                return;
            }
#pragma unroll
            for (int g = g0; g < g1; ++g) {
                if (g < NHELD) continue;                 // the epilogue reads `held` itself
                res[g][0] = ws_load(g, 0);
                res[g][1] = ws_load(g, 1);
            }
        };
        float bias[16];
#pragma unroll
        for (int r4 = 0; r4 < 4; ++r4) {
            const float4 bv = *reinterpret_cast<const float4*>(sbias + l * C + cbase + r4 * 4);
            bias[r4 * 4 + 0] = bv.x; bias[r4 * 4 + 1] = bv.y; bias[r4 * 4 + 2] = bv.z; bias[r4 * 4 + 3] = bv.w;
        }
        f32x16 acc[NT];                                  // accumulators start at the folded-BN bias
#pragma unroll
        for (int g = 0; g < NT; ++g)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[g][r] = bias[r];

        // A operands: an NPF-deep prefetch ring that keeps running across the layer boundary
        const uint4* ap = p.wpk[l] + (size_t)wm * KS * 64 + lane;
        const uint4* apn = p.wpk[l + 1 < p.nlayers ? l + 1 : l] + (size_t)wm * KS * 64 + lane;
        if (l == 0) {
#pragma unroll
            for (int k = 0; k < NPF; ++k) aq[k] = ap[k * 64];
        }
        RSG_STAMP(3 + 4 * l);
        // k-loop, one scheduling region per k-step (the compiler otherwise sinks the prefetches next to
        // their uses): issue the A load NPF steps ahead and the B reads one step ahead, then the MFMAs.
        auto boff = [&](int ks) {
            const int ch = ks / (9 * KPT), t = (ks / KPT) % 9, c = ks % KPT;
            return (t / 3 - 1) * RP + (t % 3 - 1) * PS + ch * CINB * 2 + c * 32;
        };
        // B fragments.  The order "reads of step ks+1 issue before/between the MFMAs of step ks" must be pinned:
        // left alone, hipcc sinks the reads under the MFMAs (fewer live registers) and every k-step then waits
        // ~130 cycles on LDS reads it has only just issued (measured with the stamp build: 295 instead of 160
        // cycles per k-step, the two waves of a SIMD serialise).
        constexpr bool DOUBLE_BUF = NT <= 4;
        constexpr int RES_LEAD = 1;                      // k-steps between the block-input fetch and its use
        if constexpr (DOUBLE_BUF && KS > 144) {
            // C = 384 (216 k-steps): the same loop in two halves of whole channel blocks -- hipcc gives up unrolling 216
            // steps and would index the fragment rings at run time (scratch)
            constexpr int KSO = 2, KSI = KS / KSO;
            static_assert(NCH % KSO == 0 && KSI % NPF == 0 && KSI % 2 == 0 && NRE == 0, "halves of whole channel blocks");
            uint4 bq[2][NT];
#pragma unroll
            for (int g = 0; g < NT; ++g) bq[0][g] = *reinterpret_cast<const uint4*>(rd + pixoff[g] + boff(0));
#pragma unroll 1
            for (int ko = 0; ko < KSO; ++ko) {
                const char* const rdk = rd + ko * (NCH / KSO) * CINB * 2;
                const uint4* const apk = ap + ko * KSI * 64;
                const uint4* const nxt = (ko + 1 < KSO) ? apk + KSI * 64 : apn;
                const bool more = ko + 1 < KSO;
#pragma unroll
                for (int ki = 0; ki < KSI; ++ki) {
                    const typename DT::vec8 a = __builtin_bit_cast(typename DT::vec8, aq[ki % NPF]);
                    if (ki + 1 < KSI || more) {
#pragma unroll
                        for (int g = 0; g < NT; ++g)
                            bq[(ki + 1) & 1][g] = *reinterpret_cast<const uint4*>(rdk + pixoff[g] + boff(ki + 1));
                    }
                    aq[ki % NPF] = (ki + NPF < KSI) ? apk[(ki + NPF) * 64] : nxt[(ki + NPF - KSI) * 64];
                    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                    for (int g = 0; g < NT; ++g)
                        acc[g] = DT::mfma(a, __builtin_bit_cast(typename DT::vec8, bq[ki & 1][g]), acc[g]);
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
        } else if constexpr (DOUBLE_BUF) {
            // narrow tiles: two register sets, reads one whole k-step ahead, two scheduling regions per step
            uint4 bq[2][NT];
#pragma unroll
            for (int g = 0; g < NT; ++g) bq[0][g] = *reinterpret_cast<const uint4*>(rd + pixoff[g] + boff(0));
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) {
                if (NRE > 0 && ks == KS - RES_LEAD && (l & 1)) fetch_res(0, NRE);
                const typename DT::vec8 a = __builtin_bit_cast(typename DT::vec8, aq[ks % NPF]);
                if (ks + 1 < KS) {
#pragma unroll
                    for (int g = 0; g < NT; ++g)
                        bq[(ks + 1) & 1][g] = *reinterpret_cast<const uint4*>(rd + pixoff[g] + boff(ks + 1));
                }
                aq[ks % NPF] = (ks + NPF < KS) ? ap[(ks + NPF) * 64] : apn[(ks + NPF - KS) * 64];
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int g = 0; g < NT; ++g)
                    acc[g] = DT::mfma(a, __builtin_bit_cast(typename DT::vec8, bq[ks & 1][g]), acc[g]);
                __builtin_amdgcn_sched_barrier(0);
            }
        } else {
            // wide tiles: ONE register set; the read that refills fragment g for step ks+1 issues right after
            // the MFMA that consumed it, so it has (NT-1) MFMAs = 128+ cycles to land (1 MFMA : 1 DS_READ)
            uint4 bq[NT];
#pragma unroll
            for (int g = 0; g < NT; ++g) bq[g] = *reinterpret_cast<const uint4*>(rd + pixoff[g] + boff(0));
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) {
                if (NRE > 0 && ks == KS - RES_LEAD && (l & 1)) fetch_res(0, NRE);
                const typename DT::vec8 a = __builtin_bit_cast(typename DT::vec8, aq[ks % NPF]);
                aq[ks % NPF] = (ks + NPF < KS) ? ap[(ks + NPF) * 64] : apn[(ks + NPF - KS) * 64];
#pragma unroll
                for (int g = 0; g < NT; ++g) {
                    acc[g] = DT::mfma(a, __builtin_bit_cast(typename DT::vec8, bq[g]), acc[g]);
                    if (ks + 1 < KS) bq[g] = *reinterpret_cast<const uint4*>(rd + pixoff[g] + boff(ks + 1));
                }
                __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);          // the A prefetch (VMEM read)
#pragma unroll
                for (int g = 0; g < NT; ++g) {
                    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);      // 1 MFMA
                    __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);      // 1 DS read
                }
                __builtin_amdgcn_sched_barrier(0);
            }
        }

        RSG_STAMP(4 + 4 * l);
        // ---- (+ block input on odd layers) -> 16 bit -> ReLU.  One packed convert and one packed max per two
        // elements (the residual add is a packed fp32 add); a wave-uniform branch for the residual instead of a
        // per-element select.  No in-picture mask: every held row lies inside the picture (clamped window), and an
        // image slot past the end of the batch only ever feeds itself.  The first version issued 8.5 instructions
        // per element and cost as much as the k-loop.
        uint4 o[NT][2];
        if (l & 1) {
            if constexpr (NRE < NT) fetch_res(NRE, NT);
#pragma unroll
            for (int g = 0; g < NT; ++g) {
#pragma unroll
                for (int k = 0; k < 2; ++k) {
                    const f32x8 rf = __builtin_convertvector(__builtin_bit_cast(typename DT::vec8, g < NHELD ? held[g < NHELD ? g : 0][k] : res[g][k]), f32x8);
                    f32x8 v;
#pragma unroll
                    for (int e = 0; e < 8; ++e) v[e] = acc[g][k * 8 + e] + rf[e];
                    o[g][k] = __builtin_bit_cast(uint4, DT::relu_pack(v));
                    if (l + 1 < p.nlayers && !dual && g >= NHELD) ws_store(g, k, o[g][k]); // the next block's input
                }
            }
        } else {
#pragma unroll
            for (int g = 0; g < NT; ++g) {
#pragma unroll
                for (int k = 0; k < 2; ++k) {
                    f32x8 v;
#pragma unroll
                    for (int e = 0; e < 8; ++e) v[e] = acc[g][k * 8 + e];
                    o[g][k] = __builtin_bit_cast(uint4, DT::relu_pack(v));
                }
            }
        }
        RSG_STAMP(5 + 4 * l);
        if (!dual) __syncthreads();                    // in place: every wave has finished reading the image
        int wd = own_delta;
        asm volatile("" : "+v"(wd));                     // keep the 2*NT write addresses out of the loop preheader
        if constexpr (NHELD > 0) {
            if (!(l & 1)) {                              // first conv of a block: pick up the block input before it is overwritten
#pragma unroll
                for (int g = 0; g < NHELD; ++g) {
                    held[g][0] = *reinterpret_cast<const uint4*>(wr + pixoff[g] + wd);
                    held[g][1] = *reinterpret_cast<const uint4*>(wr + pixoff[g] + wd + 16);
                }
            }
        }
#pragma unroll
        for (int g = 0; g < NT; ++g) {
            if (live[g] && (!PADDED || cbase < C)) {
                *reinterpret_cast<uint4*>(wr + pixoff[g] + wd) = o[g][0];
                *reinterpret_cast<uint4*>(wr + pixoff[g] + wd + 16) = o[g][1];
            }
        }
        __syncthreads();                                 // the new image is complete
        RSG_STAMP(6 + 4 * l);
    }

    // ---- store the TH output rows of every image slot (sc1: written through, see the hand-off above) ----------
    const __amdgpu_buffer_rsrc_t out_rsrc = __builtin_amdgcn_make_buffer_rsrc(p.out, 0, p.B * p.H * W * ldo2, 0x00020000);
    const int out_rows = min(p.TH, p.H - tile * p.TH);
    for (int i = 0; i < p.NI; ++i) {
        if (b0 + i >= p.B) break;
        const int gbase = ((b0 + i) * p.H + tile * p.TH) * W * ldo2;
        const char* const slot = smem + (i * slot_rows + tile * p.TH - y_first + 1) * RP + PS;
        int r = tid / rowc, c = tid - r * rowc;
        for (int k = tid; k < out_rows * rowc; k += NTHR) {
            const uint4 v = *reinterpret_cast<const uint4*>(slot + r * RP + (c / CPP) * PS + (c % CPP) * 16);
            __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v), out_rsrc,
                                                   gbase + (PADDED ? (r * W + c / CPP) * ldo2 + (c % CPP) * 16 : k * 16), 0, 16);
            r += step_q; c += step_r;
            if (c >= rowc) { c -= rowc; ++r; }
        }
    }
    if (p.done) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");     // every wave: its stores have left the CU
        __syncthreads();
        if (tid < 3 && !(p.debug & 4))     // debug bit 4 (tests): the producer never signals, every consumer poll gives up
            __hip_atomic_store(p.done + block_id * 3 + tid, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    RSG_STAMP(62);
}

#include "chain48.h"
#include "chain16.h"

// HRNet-W32's widths run chain16_body (16x16x32 MFMA, chain16.h); -DRSG_CHAIN16=0 builds the 32x32x16 bodies for them
// (same-visit A/B).  rsg_chain_pack_kind tells a caller which weight pack the chains of a width expect.
#ifndef RSG_CHAIN16
#define RSG_CHAIN16 1
#endif
// -DRSG_CHAIN16_128=0 keeps the 128-channel chains on chain_body (same-visit A/B): with TWO output-channel tiles per wave its
// four waves (one per SIMD) could not keep the 16-cycle MFMA fed (k-loop 69 % against 88 % with the 32-cycle form); with four
// tiles per wave (chain16_mt) a B fragment feeds four MFMAs
#ifndef RSG_CHAIN16_128
#define RSG_CHAIN16_128 1
#endif
static bool chain_is16(int C) { return RSG_CHAIN16 && (C == 32 || C == 64 || (RSG_CHAIN16_128 && C == 128) || C == 256); }

template <class DT, int C, int WM, int WN, int NG, bool DUAL = false, int WC = 0>
__global__ void __launch_bounds__(C == 128 ? 512 : 64 * WM * WN) chain16_kernel(const ChainArgs p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    chain16_body<DT, C, WM, WN, NG, DUAL, WC>(p, rsg_xcd_tile(blockIdx.x, gridDim.x), smem);
}
#ifdef RSG_CHAIN16_PROBE    // register budgets of the bodies the grouped kernel inlines (diagnostic compile only)
template __global__ void chain16_kernel<F16, 64, 1, 8, 6, false, 24>(const ChainArgs);
template __global__ void chain16_kernel<F16, 128, 2, 2, 6, false, 12>(const ChainArgs);
template __global__ void chain16_kernel<F16, 128, 2, 2, 6, true, 12>(const ChainArgs);
template __global__ void chain16_kernel<F16, 256, 8, 1, 3, true, 6>(const ChainArgs);
template __global__ void chain16_kernel<F16, 256, 8, 1, 6, false, 6>(const ChainArgs);
#endif

template <class DT>
__global__ void __launch_bounds__(768) chain48_kernel(const ChainArgs p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    chain48_body<DT>(p, rsg_xcd_tile(blockIdx.x, gridDim.x), smem);
}

template <class DT, int C, int WM, int WN, int NT, bool DUAL = false>
__global__ void __launch_bounds__(64 * WM * WN) chain_kernel(const ChainArgs p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    chain_body<DT, C, WM, WN, NT, DUAL>(p, rsg_xcd_tile(blockIdx.x, gridDim.x), smem);      // both row bands of an image on one XCD
}

// Grouped launch: up to 4 chains (different branches of one HighResolutionModule) in ONE grid; the
// workgroups of item 0 come first (put the longest chain there).  Every variant runs 512 threads.
#define RSG_CHAIN_GROUP_MAX 6
struct ChainGroupArgs {
    int n;
    // weight prefetch: one contiguous range of 128-byte lines per item (all its layers), 0 lines = none
    const char* pre_ptr[RSG_CHAIN_GROUP_MAX];
    int pre_lines[RSG_CHAIN_GROUP_MAX];
    int first[RSG_CHAIN_GROUP_MAX + 1];   // first[i] = first block of item i, first[n] = grid size
    int c[RSG_CHAIN_GROUP_MAX];
    ChainArgs a[RSG_CHAIN_GROUP_MAX];
};

// Weight prefetch of a grouped launch: before it stages, each of the first 256 workgroups (one per CU) touches
// its share of the launch's weight lines once, first layers first, which brings them into the memory-side cache
// while every workgroup is still staging (the loads are waited for behind the staging barrier).  In the running
// forward nothing of a module's 3-12 MB of weights is cached any more when its turn comes (a stage-4 launch:
// 188 us, 131 us repeated back to back, 141 us with only the weights warmed; tools/op_in_context.py) and the
// A-operand ring does not cover an HBM miss.  Dedicated prefetch workgroups did the same for stage 4 but cost
// stage 3 a CU each (they hold the launch's LDS).
template <int NTHR = 512>
static __device__ __forceinline__ unsigned chain_prefetch_weights(const ChainGroupArgs& g) {
    const int nw = min((int)gridDim.x, 256);
    unsigned acc = 0;
    if ((int)blockIdx.x >= nw) return 0;
#pragma unroll
    for (int i = 0; i < RSG_CHAIN_GROUP_MAX; ++i) {          // static indices: the scalar loads of the table batch up
        const char* w = g.pre_ptr[i];
        const int lines = g.pre_lines[i];                    // 0 for an unused entry
        for (int k = blockIdx.x * NTHR + threadIdx.x; k < lines; k += nw * NTHR)
            acc ^= *reinterpret_cast<const unsigned*>(w + (size_t)k * 128);
    }
    return acc;
}

template <class DT>
__global__ void __launch_bounds__(512) chain_group_kernel(const ChainGroupArgs g) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const unsigned pf = chain_prefetch_weights(g);
    int i = 0;       // constant indices: first[] arrives with one batch of scalar loads (see group.hip)
#pragma unroll
    for (int j = 1; j < RSG_CHAIN_GROUP_MAX; ++j) i += (j < g.n && (int)blockIdx.x >= g.first[j]) ? 1 : 0;
    const int bid = rsg_xcd_tile(blockIdx.x - g.first[i], g.first[i + 1] - g.first[i]);   // an item's row bands of one image on one XCD
#if RSG_CHAIN16
    // the maps of a 256x192 crop (48 / 24 / 12 / 6 pixels wide) with the row pitch as a compile-time constant (chain16.h, WC)
    const ChainArgs& a = g.a[i];
    const int npx = a.NI * a.R * a.W;
    switch (g.c[i]) {
        case 32:   // 36-row tiles of 4-layer chains: 108 pixel groups over 8 waves; 32-row tiles: 96
            // (14 groups per wave fit the registers only with the pitch as a constant: chain_geometry keeps other maps' tiles <= 1536 pixels)
            if (npx <= 1536) chain16_body<DT, 32, 1, 8, 12>(a, bid, smem, pf);
            else chain16_body<DT, 32, 1, 8, 14, false, 48>(a, bid, smem, pf);
            break;
        case 64:
            if (a.W == 24 && a.RP == chain16_row_pitch(64, 24)) chain16_body<DT, 64, 1, 8, 6, false, 24>(a, bid, smem, pf);
            else chain16_body<DT, 64, 1, 8, 6>(a, bid, smem, pf);
            break;
        case 128:  // 16x12 maps on FOUR waves (one per SIMD; waves 4-7 leave at once): half the weight-fragment loads of eight
            // waves (profiles/r02m_ab_w4_c128.txt); 2 x 2 waves of 64 output channels x 6 pixel groups
            if (threadIdx.x >= 256) return;
#if RSG_CHAIN16_128
            if (a.W == 12 && a.RP == chain16_row_pitch(128, 12)) {
                if (a.dual) chain16_body<DT, 128, 2, 2, 6, true, 12>(a, bid, smem, pf);
                else chain16_body<DT, 128, 2, 2, 6, false, 12>(a, bid, smem, pf);
            } else if (a.dual) chain16_body<DT, 128, 2, 2, 6, true>(a, bid, smem, pf);
            else chain16_body<DT, 128, 2, 2, 6>(a, bid, smem, pf);
#else
            if (a.dual) chain_body<DT, 128, 4, 1, 6, true>(a, bid, smem, pf);
            else chain_body<DT, 128, 4, 1, 6>(a, bid, smem, pf);
#endif
            break;
        case 256:   // 8x6 maps: one image = exactly 3 pixel groups; two images per workgroup = 6
            if (npx <= 48) {
                if (a.dual && a.W == 6 && a.RP == chain16_row_pitch(256, 6)) chain16_body<DT, 256, 8, 1, 3, true, 6>(a, bid, smem, pf);
                else if (a.dual) chain16_body<DT, 256, 8, 1, 3, true>(a, bid, smem, pf);
                else chain16_body<DT, 256, 8, 1, 3>(a, bid, smem, pf);
            } else if (a.W == 6 && a.RP == chain16_row_pitch(256, 6)) chain16_body<DT, 256, 8, 1, 6, false, 6>(a, bid, smem, pf);
            else chain16_body<DT, 256, 8, 1, 6>(a, bid, smem, pf);
            break;
    }
#else
    switch (g.c[i]) {
        case 32:   // 36-row tiles of 4-layer chains need 7 pixel groups per wave, 32-row tiles 6
            if (g.a[i].R * g.a[i].W <= 1536) chain_body<DT, 32, 1, 8, 6, false, RSG_CHAIN_HELD32_NT6>(g.a[i], bid, smem, pf);
            else chain_body<DT, 32, 1, 8, 7, false, RSG_CHAIN_HELD32>(g.a[i], bid, smem, pf);
            break;
        case 64:
            chain_body<DT, 64, 2, 4, 6, false, RSG_CHAIN_HELD64>(g.a[i], bid, smem, pf);
            break;
        case 128:  // 16x12 maps: one image = 6 pixel groups over 2 wave columns; two images per workgroup = 12
            // FOUR waves (one per SIMD, six pixel groups each; waves 4-7 leave at once) -- half the weight-fragment
            // loads, one wave per matrix pipe; same-visit A/B 1.774 -> 1.743 ms (profiles/r02m_ab_w4_c128.txt).  bf16 too
            // since its ReLU epilogue is packed (BF16::relu_pack; with the fp32 ReLU six groups did not fit 256 registers).
            if (threadIdx.x >= 256) return;
            if (g.a[i].dual) chain_body<DT, 128, 4, 1, 6, true>(g.a[i], bid, smem, pf);
            else chain_body<DT, 128, 4, 1, 6>(g.a[i], bid, smem, pf);
            break;
        case 256:   // NT follows the pixels a workgroup owns (1 image = 48 px = 2 groups, 2 images = 3 groups)
            if (g.a[i].NI * g.a[i].R * g.a[i].W <= 64) {
                if (g.a[i].dual) chain_body<DT, 256, 8, 1, 2, true>(g.a[i], bid, smem, pf);
                else chain_body<DT, 256, 8, 1, 2>(g.a[i], bid, smem, pf);
            } else chain_body<DT, 256, 8, 1, 3>(g.a[i], bid, smem, pf);
            break;
    }
#endif
}

// The widths of HRNet-W48 (48 / 96 / 192 / 384 channels: 1.5, 3, 6, 12 output-channel tiles) do not divide eight waves:
// their grouped launch runs TWELVE waves per workgroup (three per SIMD, 170 registers per lane), one wave row per
// output-channel tile.  At 384x288 only the 12x9x384 map fits LDS whole; the others run in row tiles, so a launch takes
// one BasicBlock (two layers, halo 2) of every branch.
#define RSG_CHAIN_WIDE_THREADS 768
template <class DT>
__global__ void __launch_bounds__(RSG_CHAIN_WIDE_THREADS) chain_group_wide_kernel(const ChainGroupArgs g) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const unsigned pf = chain_prefetch_weights<RSG_CHAIN_WIDE_THREADS>(g);
    int i = 0;
#pragma unroll
    for (int j = 1; j < RSG_CHAIN_GROUP_MAX; ++j) i += (j < g.n && (int)blockIdx.x >= g.first[j]) ? 1 : 0;
    const int bid = rsg_xcd_tile(blockIdx.x - g.first[i], g.first[i + 1] - g.first[i]);
    switch (g.c[i]) {
        case 48: chain48_body<DT>(g.a[i], bid, smem, pf); break;     // 16x16x32 MFMA, exactly 48 channels (chain48.h)
        case 96: chain_body<DT, 96, 3, 4, 5>(g.a[i], bid, smem, pf); break;
        case 192: chain_body<DT, 192, 6, 2, 4>(g.a[i], bid, smem, pf); break;
        case 384:   // NT follows the pixels a workgroup owns (the whole 12x9 map: 4 groups; one of its two row tiles: 3)
            if (g.a[i].NI * g.a[i].R * g.a[i].W <= 96) chain_body<DT, 384, 12, 1, 3>(g.a[i], bid, smem, pf);
            else chain_body<DT, 384, 12, 1, 4>(g.a[i], bid, smem, pf);
            break;
    }
}

// ------------------------------------------------------------------------------------------------
struct ChainCfg { int C, WM, WN, NT; };
static const ChainCfg kChainCfgs[] = {{32, 1, 8, 7}, {64, 2, 4, 6}, {128, 4, 1, 6}, {256, 8, 1, 3},
                                      {48, 2, 6, 6}, {96, 3, 4, 5}, {192, 6, 2, 4}, {384, 12, 1, 4}};
// widest variants of the grouped launch (512 threads each; 768 for the HRNet-W48 widths)
static const ChainCfg kChainGroupCfgs[] = {{32, 1, 8, 7}, {64, 2, 4, 6}, {128, 4, 2, 3}, {256, 8, 1, 3},
                                           {48, 2, 6, 6}, {96, 3, 4, 5}, {192, 6, 2, 4}, {384, 12, 1, 4}};
static bool chain_is_wide(int C) { return C == 48 || C == 96 || C == 192 || C == 384; }
static const int CHAIN_LDS_LIMIT = 160 * 1024;
static const int kChainMaxNT = 8;      // workspace slots are sized for the widest variant (standalone or grouped)
// measured (8x6x256 branch, batch 64): 1 image per workgroup (64 workgroups, shorter critical path) beats 2
static int g_chain_max_ni = 1;
static int g_chain_pitch_pad = 1;   // rsg_tune("chain_pitch_pad", 0|1)
// rsg_tune("chain_debug", bits).  Production library: only bit 4, a fault injection whose effect is DETECTED (hand-off
// producers never signal: every consumer poll gives up, the error word counts it, the plan raises HandoffTimeout).
// The timing experiments that silently give WRONG results exist only in a -DRSG_DEBUG build (RSGNET_HIP_DEBUG_BUILD=1
// for rsgnet_amd.build): 1 = every layer reads layer 0's weights, 2 = no block-input fetch.
static int g_chain_debug = 0;
#ifdef RSG_DEBUG
#define RSG_CHAIN_DEBUG_MASK 7
#else
#define RSG_CHAIN_DEBUG_MASK 4
#endif
static int g_chain_dual = 1;       // rsg_tune("chain_dual", 0|1): ping-pong LDS images for the whole-image chains that fit twice
static int g_chain_prefetch = 1;   // rsg_tune("chain_prefetch", 0|1): weight prefetch at the start of a grouped launch
static int g_chain_split_rows = 1; // rsg_tune("chain_split_rows", 0|1): two row tiles for the 384-channel map at small batch (chain_geometry)
static int chain_row_pitch(int C, int W) {
    const int PS = 2 * C + 16;
    return (W + 2) * PS + (g_chain_pitch_pad ? (256 - (2 * PS) % 256) % 256 : 0);
}   // rsg_tune("chain_max_ni", n): images per workgroup cap in whole-image mode
extern int g_conv_lds_limit;
extern int g_conv_full_k;
extern int g_conv_spread;
extern int g_bottleneck_head;
#ifdef RSG_DEBUG
extern int g_bottleneck_debug;
#endif
extern int g_fm_prefetch;              // fuse_module.hip
extern int g_stem_persist;             // stem.hip
extern int g_dual_waves;               // dual.hip
extern "C" int rsg_tune(const char* key, int32_t value) {
    if (key && !strcmp(key, "fm_prefetch")) { g_fm_prefetch = value ? 1 : 0; return 0; }
    if (key && !strcmp(key, "stem_persist")) { g_stem_persist = value ? 1 : 0; return 0; }
    if (key && !strcmp(key, "dual_waves") && (value == 2 || value == 4)) { g_dual_waves = value; return 0; }
    if (key && !strcmp(key, "bottleneck_head") && (value == 1 || value == 2)) { g_bottleneck_head = value; return 0; }
#ifdef RSG_DEBUG
    if (key && !strcmp(key, "bottleneck_debug")) { g_bottleneck_debug = value; return 0; }
#endif
    if (key && !strcmp(key, "conv_full_k")) { g_conv_full_k = value ? 1 : 0; return 0; }
    if (key && !strcmp(key, "conv_spread") && value >= 0 && value <= 4096) { g_conv_spread = value; return 0; }
    if (key && !strcmp(key, "conv_lds_limit") && value >= 8192 && value <= 64 * 1024) { g_conv_lds_limit = value; return 0; }
    if (key && !strcmp(key, "chain_max_ni") && value >= 1) { g_chain_max_ni = value; return 0; }
    if (key && !strcmp(key, "chain_pitch_pad")) { g_chain_pitch_pad = value ? 1 : 0; return 0; }
    if (key && !strcmp(key, "chain_prefetch")) { g_chain_prefetch = value ? 1 : 0; return 0; }
    if (key && !strcmp(key, "chain_debug")) {
        if (value & ~RSG_CHAIN_DEBUG_MASK) RSG_FAIL("tune: chain_debug bits 0x%x exist only in a -DRSG_DEBUG build of the library", value & ~RSG_CHAIN_DEBUG_MASK);
        g_chain_debug = value;
        return 0;
    }
    if (key && !strcmp(key, "chain_dual")) { g_chain_dual = value ? 1 : 0; return 0; }
    if (key && !strcmp(key, "chain_split_rows")) { g_chain_split_rows = value ? 1 : 0; return 0; }
    RSG_FAIL("tune: unknown key or bad value");
}

typedef void (*chain_fn)(const ChainArgs);
template <class DT>
static chain_fn chain_lookup(int C, int npix, int dual) {
#if RSG_CHAIN16
    switch (C) {
        case 32: return npix <= 1536 ? chain16_kernel<DT, 32, 1, 8, 12> : chain16_kernel<DT, 32, 1, 8, 14, false, 48>;
        case 64: return chain16_kernel<DT, 64, 1, 8, 6>;
#if RSG_CHAIN16_128
        case 128: return dual ? chain16_kernel<DT, 128, 2, 2, 6, true> : chain16_kernel<DT, 128, 2, 2, 6>;
#endif
        case 256:
            if (npix <= 48) return dual ? chain16_kernel<DT, 256, 8, 1, 3, true> : chain16_kernel<DT, 256, 8, 1, 3>;
            return chain16_kernel<DT, 256, 8, 1, 6>;
    }
#endif
    if (dual) {
        if (C == 256 && npix <= 64) return chain_kernel<DT, 256, 8, 1, 2, true>;
        if (C == 128) return chain_kernel<DT, 128, 4, 1, 6, true>;
        return nullptr;
    }
    if (C == 256 && npix <= 64) return chain_kernel<DT, 256, 8, 1, 2>;
    if (C == 32 && npix <= 1536) return chain_kernel<DT, 32, 1, 8, 6>;
    switch (C) {
        case 32: return chain_kernel<DT, 32, 1, 8, 7>;
        case 64: return chain_kernel<DT, 64, 2, 4, 6>;
        case 128: return chain_kernel<DT, 128, 4, 1, 6>;
        case 256: return chain_kernel<DT, 256, 8, 1, 3>;
        case 48: return chain48_kernel<DT>;
        case 96: return chain_kernel<DT, 96, 3, 4, 5>;
        case 192: return chain_kernel<DT, 192, 6, 2, 4>;
        case 384: return npix <= 96 ? chain_kernel<DT, 384, 12, 1, 3> : chain_kernel<DT, 384, 12, 1, 4>;
    }
    return nullptr;
}

// Picks (tile rows, images per workgroup) for a chain; returns 0 when the shape is supported.
// ni_want: images per workgroup asked for in whole-image mode (0 = the process-wide default, rsg_tune)
static int chain_geometry(int C, int H, int W, int nlayers, int batch, ChainArgs& a, int& lds, int& nthr,
                          bool grouped = false, int ni_want = 0, bool force_split = false, bool exchange = false) {
    const ChainCfg* cfg = nullptr;
    for (const ChainCfg& c : (grouped ? kChainGroupCfgs : kChainCfgs)) if (c.C == C) cfg = &c;
    if (ni_want < 1) ni_want = g_chain_max_ni;
    if (!cfg) return 1;
    if (C == 48 && nlayers != 2) return 1;               // chain48_body: one BasicBlock, the residual is the launch's own input
    int cap = 32 * cfg->NT * cfg->WN;                    // pixels a workgroup can own
    const int PS = 2 * C + 16;
    nthr = 64 * cfg->WM * cfg->WN;
    const int RP = chain_row_pitch(C, W);
    // chain16_body<32>: 14 pixel groups per wave (1792 pixels) exist only for the 48-pixel-wide map with its pitch as a constant
    if (chain_is16(C) && C == 32 && !(W == 48 && RP == chain16_row_pitch(32, 48))) cap = 1536;
    a.RP = RP;
    auto bytes = [&](int ni, int R) { return ni * (R + 2) * RP + RSG_CHAIN_MAX_LAYERS * C * 4; };   // image + biases
    // 384 channels, one BasicBlock, a small batch: whole images would occupy `batch` compute units for the longest
    // workgroups of the launch (12x9x384, batch 32: 112 us on 32 of 256 CUs while the other branches need ~90 us of the
    // rest).  Two row tiles with a 2-row halo do 1.5x the work on twice the CUs and shorten that critical path by a quarter.
    // Same-visit A/B at batch 32: neutral (5.152 vs 5.153 ms per step) while the 48-channel chain made a stage-4 launch
    // work-bound; with that branch on the 16x16x32 MFMA (chain48.h) the 384-channel workgroups are the critical path again
    // and the split pays: 4.545 -> 4.449 ms.
    // (rsg_chain_item.split_rows, round 5: the same for the whole-image 32x24x64 chain of the two-branch module -- 8 layers,
    // 72 us on 64 of 256 CUs while the launch's other branch is done after 60; two row tiles with an 8-row halo each hold
    // 24 of the 32 rows)
    const bool split_rows = (g_chain_split_rows && C == 384 && nlayers == 2 && H >= 8 && batch * 4 <= 256 &&
                             (H / 2 + nlayers) * W <= 96) ||
                            (force_split && H >= 16 && H / 2 + nlayers < H);
    if (exchange && (!chain_is16(C) || nlayers < 4 || (nlayers & 3))) return 1;   // halves of whole BasicBlocks, chain16 bodies only
    if (!exchange && !split_rows && H * W <= cap && bytes(1, H) <= CHAIN_LDS_LIMIT) {            // whole images
        int ni = cap / (H * W);
        while (ni > 1 && (bytes(ni, H) > CHAIN_LDS_LIMIT || ni > batch || ni > ni_want)) --ni;
        a.TH = H; a.R = H; a.halo = 0; a.NI = ni; a.tiles_y = 1;
        lds = bytes(ni, H);
        // two images (ping-pong) where they fit: 16x12x128 and 8x6x256 (73 / 45 KB each)
        const int image = ni * (H + 2) * RP;
        a.dual = g_chain_dual && (C == 128 || (C == 256 && ni * H * W <= 64)) && lds + image <= CHAIN_LDS_LIMIT ? 1 : 0;
        if (a.dual) lds += image;
        return 0;
    }
    // row tiles with halo recompute: the fewest tiles whose window (TH rows + an L-row halo on every side
    // that is not the picture's edge) fits the pixel capacity and the LDS
    const int halo = exchange ? nlayers / 2 : nlayers;   // with the exchange the halo is renewed half way
    for (int tiles = 2; tiles <= H; ++tiles) {
        const int TH = ceil_div(H, tiles);
        const int R = TH + (tiles == 2 ? 1 : 2) * halo;
        if (TH < 4) break;
        if (R > H || R * W > cap || bytes(1, R) > CHAIN_LDS_LIMIT) continue;
        if (exchange && (TH < halo || H % TH)) return 1;  // a tile gives `halo` of its own rows to each neighbour
        a.TH = TH; a.R = R; a.halo = halo; a.NI = 1; a.tiles_y = ceil_div(H, TH); a.dual = 0;
        a.xchg = exchange ? halo : 0;
        lds = bytes(1, R);
        return 0;
    }
    return 1;
}

extern "C" int rsg_chain_handoff_timeouts(uint32_t* count) {
    if (!count) RSG_FAIL("chain_handoff_timeouts: null pointer");
    unsigned v = 0;
    if (hipMemcpyFromSymbol(&v, HIP_SYMBOL(g_chain_handoff_timeouts), sizeof(v)) != hipSuccess)
        RSG_FAIL("chain_handoff_timeouts: hipMemcpyFromSymbol failed");
    *count = v;
    return 0;
}

extern "C" int rsg_chain_handoff_timeouts_async(uint32_t* pinned_host, void* stream) {
    if (!pinned_host) RSG_FAIL("chain_handoff_timeouts_async: null pointer");
    if (hipMemcpyFromSymbolAsync(pinned_host, HIP_SYMBOL(g_chain_handoff_timeouts), sizeof(unsigned), 0,
                                 hipMemcpyDeviceToHost, (hipStream_t)stream) != hipSuccess)
        RSG_FAIL("chain_handoff_timeouts_async: hipMemcpyFromSymbolAsync failed");
    return 0;
}

// Compute units of the current device (cached per device id); 0 when the query fails.
static int chain_device_cus() {
    static int cus[64] = {0};
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return 0;
    if (!cus[dev]) {
        int v = 0;
        if (hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess) cus[dev] = v;
    }
    return cus[dev];
}

extern "C" int rsg_basic_chain_handoff_max_waiters(void) {
    // every chain workgroup holds more than half a CU's LDS, so one is resident per CU; as long as FEWER workgroups
    // than CUs can wait on flags, at least one CU always runs workgroups that never block, every producer gets its
    // turn whatever the dispatch order, and the launch cannot deadlock
    const int cus = chain_device_cus();
    return cus > 0 ? cus - 1 : 0;
}

extern "C" int rsg_basic_chain_supported(int32_t c, int32_t h, int32_t w, int32_t nlayers, int32_t batch,
                                         int32_t* info4) {
    ChainArgs a;
    int lds = 0, nthr = 0;
    if (nlayers < 2 || nlayers > RSG_CHAIN_MAX_LAYERS || (nlayers & 1)) return 0;
    if (chain_geometry(c, h, w, nlayers, batch, a, lds, nthr)) return 0;
    if (info4) { info4[0] = a.TH; info4[1] = a.NI; info4[2] = lds; info4[3] = ceil_div(batch, a.NI) * a.tiles_y; }
    return 1;
}

// bytes of the block-input workspace of one chain: a 32-byte slot per lane and pixel group, per workgroup
static size_t chain_ws_bytes(int C, const ChainArgs& a, int batch, int nlayers, int nthr) {
    if (chain_is16(C)) return 16;       // chain16_body parks nothing in memory (the pointer must still be valid and aligned)
    const int blocks = ceil_div(batch, a.NI) * a.tiles_y;
    return (size_t)blocks * nthr * kChainMaxNT * 32;
}

extern "C" int rsg_basic_chain_supported_split(int32_t c, int32_t h, int32_t w, int32_t nlayers, int32_t batch, int32_t* info4) {
    ChainArgs a;
    int lds = 0, nthr = 0;
    if (nlayers < 2 || nlayers > RSG_CHAIN_MAX_LAYERS || (nlayers & 1)) return 0;
    if (chain_geometry(c, h, w, nlayers, batch, a, lds, nthr, true, 0, true) || a.tiles_y != 2) return 0;
    if (info4) { info4[0] = a.TH; info4[1] = a.NI; info4[2] = lds; info4[3] = ceil_div(batch, a.NI) * a.tiles_y; }
    return 1;
}

extern "C" size_t rsg_basic_chain_workspace_bytes(int32_t c, int32_t h, int32_t w, int32_t nlayers, int32_t batch) {
    ChainArgs a;
    int lds = 0, nthr = 0;
    if (nlayers < 2 || nlayers > RSG_CHAIN_MAX_LAYERS || (nlayers & 1)) return 0;
    if (chain_geometry(c, h, w, nlayers, batch, a, lds, nthr)) return 0;
    return chain_ws_bytes(c, a, batch, nlayers, chain_is_wide(c) ? RSG_CHAIN_WIDE_THREADS : 512);
}

// rsg_chain_item.exchange: bytes of the item's workspace (the exchange buffer) and, through info4 = {rows per tile, 1, LDS bytes,
// workgroups}, whether the shape runs in that form at all (0: it does not)
extern "C" size_t rsg_basic_chain_exchange_bytes(int32_t c, int32_t h, int32_t w, int32_t nlayers, int32_t batch, int32_t* info4) {
    ChainArgs a;
    memset(&a, 0, sizeof(a));
    int lds = 0, nthr = 0;
    if (nlayers < 4 || nlayers > RSG_CHAIN_MAX_LAYERS || batch < 1) return 0;
    if (chain_geometry(c, h, w, nlayers, batch, a, lds, nthr, true, 0, false, true)) return 0;
    if (info4) { info4[0] = a.TH; info4[1] = 1; info4[2] = lds; info4[3] = batch * a.tiles_y; }
    return (size_t)batch * a.tiles_y * 2 * a.halo * w * c * 2;
}

extern "C" int rsg_basic_chain(const void* in, void* out, const void* const* w_packed, const float* const* bias,
                               int32_t nlayers, int32_t batch, int32_t h, int32_t w, int32_t c, int32_t dtype,
                               void* workspace, void* stream) {
    if (!in || !out || !w_packed || !bias) RSG_FAIL("basic_chain: null pointer");
    if (!workspace || ((uintptr_t)workspace & 15))
        RSG_FAIL("basic_chain: needs a 16-byte aligned workspace (rsg_basic_chain_workspace_bytes)");
    if (in == out) RSG_FAIL("basic_chain: in-place operation is not supported (row tiles read neighbours' halos)");
    if (nlayers < 2 || nlayers > RSG_CHAIN_MAX_LAYERS || (nlayers & 1))
        RSG_FAIL("basic_chain: nlayers=%d must be even and <= %d", nlayers, RSG_CHAIN_MAX_LAYERS);
    ChainArgs a;
    memset(&a, 0, sizeof(a));
    int lds = 0, nthr = 0;
    if (chain_geometry(c, h, w, nlayers, batch, a, lds, nthr))
        RSG_FAIL("basic_chain: unsupported shape C=%d %dx%d L=%d", c, h, w, nlayers);
    if ((long long)batch * h * w * c * 2 >= (1LL << 31))
        RSG_FAIL("basic_chain: map of %lld bytes exceeds the 32-bit buffer offsets", (long long)batch * h * w * c * 2);
    a.in = (const char*)in; a.out = (char*)out; a.nlayers = nlayers; a.B = batch; a.H = h; a.W = w;
    a.ldi = a.ldo = c;
    a.ws = (uint4*)workspace;
    a.done = nullptr; a.wait = nullptr; a.debug = g_chain_debug;
    for (int l = 0; l < nlayers; ++l) {
        if (!w_packed[l] || !bias[l]) RSG_FAIL("basic_chain: null weights for layer %d", l);
        a.wpk[l] = (const uint4*)w_packed[g_chain_debug & 1 ? 0 : l];
        a.bias[l] = bias[l];
    }
    const int npix = a.NI * a.R * w;
    chain_fn fn = dtype == RSG_BF16 ? chain_lookup<BF16>(c, npix, a.dual) : chain_lookup<F16>(c, npix, a.dual);
    if (!fn) RSG_FAIL("basic_chain: no kernel for C=%d", c);
    if (hipFuncSetAttribute((const void*)fn, hipFuncAttributeMaxDynamicSharedMemorySize, lds) != hipSuccess)
        RSG_FAIL("basic_chain: cannot reserve %d B of LDS", lds);
    dim3 grid(ceil_div(batch, a.NI) * a.tiles_y);
    RSG_LAUNCH(fn, grid, dim3(nthr), lds, (hipStream_t)stream, a);
    RSG_CHECK_LAUNCH("basic_chain");
    return 0;
}

extern "C" int rsg_basic_chain_group(const rsg_chain_item* items, int32_t n, int32_t dtype, void* stream) {
    if (!items || n < 1 || n > RSG_CHAIN_GROUP_MAX) RSG_FAIL("chain_group: 1..%d items", RSG_CHAIN_GROUP_MAX);
    ChainGroupArgs g;
    memset(&g, 0, sizeof(g));
    g.n = n;
    int lds_max = 0, blocks = 0, waiters = 0;
    for (int i = 0; i < n; ++i) {
        const rsg_chain_item& it = items[i];
        if (!it.in || !it.out || !it.w_packed || !it.bias || it.in == it.out) RSG_FAIL("chain_group: bad pointers in item %d", i);
        if (it.nlayers < 2 || it.nlayers > RSG_CHAIN_MAX_LAYERS || (it.nlayers & 1)) RSG_FAIL("chain_group: nlayers=%d", it.nlayers);
        ChainArgs& a = g.a[i];
        int lds = 0, nthr = 0;
        if (it.split_rows & ~1) RSG_FAIL("chain_group: item %d: split_rows = %d (0 or 1)", i, it.split_rows);
        if (it.exchange & ~1) RSG_FAIL("chain_group: item %d: exchange = %d (0 or 1)", i, it.exchange);
        if (it.exchange && (it.split_rows || it.wait_flags || !it.done_flags || ((uintptr_t)it.done_flags & 3)))
            RSG_FAIL("chain_group: item %d: a chain with a halo exchange takes zeroed done_flags (3 words per workgroup) and neither wait_flags nor split_rows", i);
        if (chain_geometry(it.c, it.h, it.w, it.nlayers, it.batch, a, lds, nthr, true, it.images_per_workgroup, it.split_rows != 0, it.exchange != 0))
            RSG_FAIL("chain_group: unsupported shape C=%d %dx%d L=%d%s", it.c, it.h, it.w, it.nlayers, it.exchange ? " with a halo exchange" : "");
        if ((!it.workspace || ((uintptr_t)it.workspace & 15))) RSG_FAIL("chain_group: item %d needs a workspace", i);
        if ((long long)it.batch * it.h * it.w * it.c * 2 >= (1LL << 31))
            RSG_FAIL("chain_group: item %d: map of %lld bytes exceeds the 32-bit buffer offsets", i,
                     (long long)it.batch * it.h * it.w * it.c * 2);
        a.in = (const char*)it.in; a.out = (char*)it.out; a.nlayers = it.nlayers; a.B = it.batch; a.H = it.h; a.W = it.w;
        a.ldi = it.in_ld ? it.in_ld : it.c; a.ldo = it.out_ld ? it.out_ld : it.c;
        if (a.ldi < it.c || a.ldo < it.c || (a.ldi & 7) || (a.ldo & 7) || ((a.ldi != it.c || a.ldo != it.c) && it.c % 32 == 0))
            RSG_FAIL("chain_group: item %d: pixel strides %d / %d (only the 48-channel chain takes padded buffers)", i, a.ldi, a.ldo);
        if ((long long)it.batch * it.h * it.w * std::max(a.ldi, a.ldo) * 2 >= (1LL << 31))
            RSG_FAIL("chain_group: item %d: padded map exceeds the 32-bit buffer offsets", i);
        if (chain_is_wide(it.c) != chain_is_wide(items[0].c))
            RSG_FAIL("chain_group: item %d: the 48/96/192/384-channel chains (768 threads) cannot share a launch with the others", i);
        a.ws = (uint4*)it.workspace;
        a.done = it.exchange ? nullptr : (int*)it.done_flags;
        a.wait = (const int*)it.wait_flags;
        if (it.exchange) {                     // (workspace: rsg_basic_chain_exchange_bytes; every tile both signals and waits)
            a.xbuf = (char*)it.workspace;
            a.xflag = (int*)it.done_flags;
            waiters += it.batch * a.tiles_y;
        }
        a.debug = g_chain_debug;
        if (it.wait_flags) {
            // the producer comes EARLIER in the grid (in-order dispatch then makes the wait short; correctness does
            // not need it, see the waiter bound below), writes what this item reads and cuts the map into the same tiles
            int j = 0;
            while (j < i && items[j].done_flags != it.wait_flags) ++j;
            if (j == i) RSG_FAIL("chain_group: item %d waits on flags no earlier item sets", i);
            const rsg_chain_item& pr = items[j];
            if (pr.out != it.in || pr.c != it.c || pr.h != it.h || pr.w != it.w || pr.batch != it.batch ||
                pr.nlayers != it.nlayers || g.a[j].NI != 1 || a.NI != 1 || g.a[j].tiles_y != a.tiles_y)
                RSG_FAIL("chain_group: item %d must read item %d's output with the same tiling", i, j);
        }
        for (int l = 0; l < it.nlayers; ++l) {
            if (!it.w_packed[l] || !it.bias[l]) RSG_FAIL("chain_group: null weights (item %d layer %d)", i, l);
            a.wpk[l] = (const uint4*)it.w_packed[g_chain_debug & 1 ? 0 : l];
            a.bias[l] = it.bias[l];
        }
        // prefetch range: the item's layers when the caller packed them back to back (the plan does)
        const size_t layer_bytes = it.c == 48 ? (size_t)RSG_CHAIN48_KS * 3 * 64 * 16 : (size_t)it.c * it.c * 18;
        if (it.c == 48 && (it.wait_flags || it.done_flags)) RSG_FAIL("chain_group: item %d: the 48-channel chain takes no hand-off flags", i);
        bool contiguous = g_chain_prefetch != 0;
        for (int l = 1; l < it.nlayers; ++l)
            contiguous = contiguous && (const char*)it.w_packed[l] == (const char*)it.w_packed[0] + l * layer_bytes;
        if (contiguous) {
            g.pre_ptr[i] = (const char*)it.w_packed[0];
            g.pre_lines[i] = (int)(layer_bytes * it.nlayers / 128);
        }
        g.c[i] = it.c;
        g.first[i] = blocks;
        if (it.wait_flags) waiters += ceil_div(it.batch, a.NI) * a.tiles_y;
        blocks += ceil_div(it.batch, a.NI) * a.tiles_y;
        if (lds > lds_max) lds_max = lds;
    }
    g.first[n] = blocks;
    if (waiters > 0 && waiters > rsg_basic_chain_handoff_max_waiters())
        RSG_FAIL("chain_group: %d workgroups wait on hand-off flags but the device guarantees progress for at most %d "
                 "(compute units - 1): run the dependent chains as a second launch", waiters,
                 rsg_basic_chain_handoff_max_waiters());
    const bool wide = chain_is_wide(items[0].c);
    void (*fn)(const ChainGroupArgs) = wide ? (dtype == RSG_BF16 ? chain_group_wide_kernel<BF16> : chain_group_wide_kernel<F16>)
                                            : (dtype == RSG_BF16 ? chain_group_kernel<BF16> : chain_group_kernel<F16>);
    if (hipFuncSetAttribute((const void*)fn, hipFuncAttributeMaxDynamicSharedMemorySize, lds_max) != hipSuccess)
        RSG_FAIL("chain_group: cannot reserve %d B of LDS", lds_max);
    RSG_LAUNCH(fn, dim3(blocks), dim3(wide ? RSG_CHAIN_WIDE_THREADS : 512), lds_max, (hipStream_t)stream, g);
    RSG_CHECK_LAUNCH("basic_chain_group");
    return 0;
}

#ifdef RSG_CHAIN_STAMPS
extern "C" int rsg_chain_set_stamp_buffer(void* dev_ptr) {
    unsigned long long* p = (unsigned long long*)dev_ptr;
    if (hipMemcpyToSymbol(HIP_SYMBOL(g_chain_stamps), &p, sizeof(p)) != hipSuccess) RSG_FAIL("stamps: memcpy failed");
    return 0;
}
#endif

// Weights of one layer of the 48-channel chain (chain48.h): w fp32 [48][48][3][3], BatchNorm folded like
// rsg_pack_conv_weights_host (scale into the weights, shift into the bias; either may be NULL).
// wp_out: 14 * 3 * 64 * 16 = 43008 bytes; bias_out: 48 floats.
extern "C" size_t rsg_chain48_packed_bytes(void) { return (size_t)RSG_CHAIN48_KS * 3 * 64 * 16; }
extern "C" int rsg_pack_chain48_weights_host(const float* w, const float* scale, const float* shift, int32_t dtype, void* wp_out, float* bias_out) {
    if (!w || !wp_out || !bias_out) RSG_FAIL("pack_chain48: null pointer");
    unsigned short* dst = (unsigned short*)wp_out;
    for (int s = 0; s < RSG_CHAIN48_KS; ++s)
        for (int m = 0; m < 3; ++m)
            for (int lane = 0; lane < 64; ++lane)
                for (int j = 0; j < 8; ++j) {
                    const int co = 16 * m + (lane & 15), f = 32 * s + 8 * (lane >> 4) + j;
                    float v = 0.f;
                    if (f < 432) {
                        const int tap = f / 48, ci = f % 48;
                        v = w[((size_t)co * 48 + ci) * 9 + tap] * (scale ? scale[co] : 1.f);
                    }
                    unsigned short bits;
                    if (dtype == RSG_BF16) { const BF16::elem e = BF16::from_f32(v); memcpy(&bits, &e, 2); }
                    else { const F16::elem e = F16::from_f32(v); memcpy(&bits, &e, 2); }
                    dst[(((size_t)s * 3 + m) * 64 + lane) * 8 + j] = bits;
                }
    for (int co = 0; co < 48; ++co) bias_out[co] = shift ? shift[co] : 0.f;
    return 0;
}

// Which weight pack the chains of width c expect: 16 = rsg_pack_chain16_weights_host, 48 = rsg_pack_chain48_weights_host,
// 0 = the conv pack (rsg_pack_conv_weights_host).  rsg_pack_chain_weights_host packs one layer in that format.
extern "C" int rsg_chain_pack_kind(int32_t c) { return chain_is16(c) ? 16 : (c == 48 ? 48 : 0); }

// Weights of one layer of a chain16_body chain (chain16.h): w fp32 [C][C][3][3], BatchNorm folded like
// rsg_pack_conv_weights_host (scale into the weights, shift into the bias; either may be NULL).
// wp_out: 18 C^2 bytes = [wave row C / (16 MT)][k-step (32-channel block, tap)][tile MT][lane 64] x 16 B, MT = chain16_mt(C);
// bias_out: C floats.
extern "C" int rsg_pack_chain16_weights_host(const float* w, const float* scale, const float* shift, int32_t c, int32_t dtype,
                                             void* wp_out, float* bias_out) {
    if (!w || !wp_out || !bias_out) RSG_FAIL("pack_chain16: null pointer");
    if (c < 32 || c % 32) RSG_FAIL("pack_chain16: %d channels (multiples of 32)", c);
    unsigned short* dst = (unsigned short*)wp_out;
    const int KS = c / 32 * 9, MT = chain16_mt(c), CW = 16 * MT;
    for (int wm = 0; wm < c / CW; ++wm)
        for (int s = 0; s < KS; ++s)
            for (int m = 0; m < MT; ++m)
                for (int lane = 0; lane < 64; ++lane) {
                    const int i = lane & 15, q = lane >> 4, sig = ((q & 1) << 1) | (q >> 1);
                    const int co = CW * wm + 4 * MT * (i >> 2) + 4 * m + (i & 3);
                    const int cb = s / 9, tap = s % 9;
                    for (int e = 0; e < 8; ++e) {
                        const int ci = 32 * cb + 8 * sig + e;
                        const float v = w[((size_t)co * c + ci) * 9 + tap] * (scale ? scale[co] : 1.f);
                        unsigned short bits;
                        if (dtype == RSG_BF16) { const BF16::elem h = BF16::from_f32(v); memcpy(&bits, &h, 2); }
                        else { const F16::elem h = F16::from_f32(v); memcpy(&bits, &h, 2); }
                        dst[((((size_t)wm * KS + s) * MT + m) * 64 + lane) * 8 + e] = bits;
                    }
                }
    for (int co = 0; co < c; ++co) bias_out[co] = shift ? shift[co] : 0.f;
    return 0;
}
