// LDS-resident BasicBlock chains of HRNet-W32's widths (32 / 64 / 128 / 256 channels) on the 16x16x32 MFMA -- included by
// chain.hip (ChainArgs, RSG_STAMP, Mfma16 of chain48.h).
//
// Replaces  x = branches[i](x)  of HighResolutionModule.forward (/root/reference/lib/models/pose_hrnet.py:254-255;
// BasicBlock :30-59), like chain_body, which it supersedes for these widths.  Why a second body:
//   * POWER, not issue slots, bounds an MFMA-dense loop on this chip: the C = 32 k-loop on random data holds 1.49 GHz with
//     v_mfma_f32_32x32x16_f16 and 1.74 GHz with v_mfma_f32_16x16x32_f16 at the same LDS and L2 bytes per FLOP
//     (tools/probes/mfma_shape_probe.*, profiles/r04_mfma_shape_probe.txt: 1 440 -> 1 619 TFLOP/s, +12 %);
//   * a 16-pixel group fits the 8x6 map of the 256-channel branch exactly (48 pixels = 3 groups; the 32-pixel form
//     multiplied a quarter of its matrix work by padding);
//   * in this accumulator layout a lane owns 8 CONSECUTIVE channels of one pixel (two output-channel tiles x 4 rows, the
//     weight rows permuted at pack time) = one 16-byte chunk = exactly the chunk it supplies as a B fragment of that pixel.
//     The block input a lane must add back before the second ReLU is therefore ONE 16-byte LDS read of its own pixel
//     just before it overwrites it (in-place mode; ping-pong mode reads it from the image it is about to overwrite), and
//     it does not wait anywhere: it becomes the START VALUE of the second conv's accumulators (bias + x).  No held
//     registers across a k-loop, no global workspace (chain_body parked 3 of 7 pixel groups of the C = 32 chain there:
//     155 MB moved per grouped launch for 76 MB of activations), and the odd layers' epilogue is the even layers'.
//
// Mapping.  A wave owns 16 MT output channels (wave row wm: MT = 2 or 4 tiles of 16, chain16_mt) of NG groups of 16 pixels
// (wave column wn).  One k-step = one tap of one 32-channel block of the input: MT A fragments (weights, from L2 in packed
// order through a register ring that keeps running across layers), one ds_read_b128 per pixel group (B), MT MFMAs per group.
//   A lane (i = lane & 15, q = lane >> 4): output channel 16 MT wm + 4 MT (i >> 2) + 4 m + (i & 3) of tile m,
//                                         input channels 32 cb + 8 sigma(q) .. + 7           (rsg_pack_chain16_weights_host)
//   B lane (col, q): pixel pi(col) of the group, the same 8 input channels = 16 bytes at chunk sigma(q) of the block
//   D lane (col, q): rows 4q .. 4q + 3 of tile m  ->  channels 16 MT wm + 4 MT q + 4 m + r: the lane's 4 MT accumulators of
//                    a pixel group are channels [16 MT wm + 4 MT q, + 4 MT) of pixel pi(col): MT / 2 chunks of 16 bytes
// sigma = [0, 2, 1, 3] and pi (columns {0-3, 12-15} take the group's even pixels, {4-11} the odd ones) make every
// 16-lane phase of a ds_read_b128 ({0-3, 12-15, 20-27}, ...: MI355X_MICROARCH.md, LDS) cover all 64 banks once with the
// image's odd pixel stride of (2 C + 16) / 16 chunks; the 8-lane phases of the ds_write_b128 of the epilogue cover the 32
// store banks once.  LDS image, borders, staging, hand-off and final store are chain_body's.
#pragma once
#ifndef RSG_CHAIN16_NB128
#define RSG_CHAIN16_NB128 3          // B fragments in flight of the ping-pong 128-channel chain (3 x 64 matrix-pipe cycles)
#endif

__device__ __forceinline__ int chain16_pi(int col) {
    return col < 4 ? 2 * col : (col >= 12 ? 2 * (col - 8) : 2 * (col - 4) + 1);
}

// Barrier between the phases of a layer: the LDS operations of this wave have completed (lgkmcnt), nothing else is waited
// for -- __syncthreads() also drains vmcnt, i.e. the weight fragments the ring has just requested for the NEXT layer
// (an L2 round trip in front of every first barrier of a layer).
__device__ __forceinline__ void chain16_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

// WC: the map's width as a compile-time constant (0 = run-time p.W / p.RP).  With it the row pitch is a constant and every
// tap of a B fragment is an immediate offset of ONE address register per pixel group; at run-time pitch each read costs
// a vector add, and a 16-cycle MFMA leaves half of its issue slots to everything else.
constexpr int chain16_row_pitch(int C, int W) { return (W + 2) * (2 * C + 16) + (256 - (2 * (2 * C + 16)) % 256) % 256; }

// Output-channel tiles (of 16) per wave, by width -- shared by the kernels and rsg_pack_chain16_weights_host.  Four where the
// width allows: a B fragment then feeds four MFMAs (one LDS read + its wait per 64 matrix-pipe cycles instead of per 32; a
// 16-cycle MFMA holds the issue port for 8), and the 128-channel map's four waves stay fed with one wave per SIMD.
constexpr int chain16_mt(int C) { return (C == 64 || C == 128) ? 4 : 2; }

template <class DT, int C, int WM, int WN, int NG, bool DUAL = false, int WC = 0>
__device__ __forceinline__ void chain16_body(const ChainArgs& p, const int block_id, char* smem, const unsigned pf = 0) {
    constexpr int NTHR = 64 * WM * WN;
    constexpr int MT = chain16_mt(C);                    // 16-channel output tiles per wave
    constexpr int CW = 16 * MT, NX = MT / 2;             // output channels per wave; 16-byte chunks of a pixel per lane
    static_assert(C % 32 == 0 && WM * CW == C, "one wave row per 16 MT output channels");
    constexpr int NCB = C / 32, KS = NCB * 9;           // k-steps per layer: (32-channel block, tap)
    constexpr int PS = 2 * C + 16;                       // LDS bytes per pixel
    constexpr int CPP = C / 8;                           // 16-byte chunks per pixel

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave / WN, wn = wave % WN;
    const int col = lane & 15, q = lane >> 4;
    const int W = WC ? WC : p.W, R = p.R;
    const int RP = WC ? chain16_row_pitch(C, WC) : p.RP; // row pitch == W * PS (mod 256 B): chain_row_pitch
    const int slot_rows = R + 2;
    const int tile = block_id % p.tiles_y;
    const int b0 = (block_id / p.tiles_y) * p.NI;
    const int y_first = min(max(tile * p.TH - p.halo, 0), p.H - R);

    RSG_STAMP(0);
#ifdef RSG_CHAIN_STAMPS
    if (g_chain_stamps && (threadIdx.x & 63) == 0) {
        unsigned xcc, hwid;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)\n\ts_getreg_b32 %1, hwreg(HW_REG_HW_ID)" : "=s"(xcc), "=s"(hwid));
        g_chain_stamps[((size_t)blockIdx.x * 16 + (threadIdx.x >> 6)) * 64 + 63] = ((unsigned long long)xcc << 32) | hwid;
    }
#endif
    // ---- zero what staging does not write (border rows and pixels of every image slot; whole slots past the batch) ----
    const int lds_bytes = p.NI * slot_rows * RP;
    {
        const int row_chunks = RP / 16;
        const int per_slot = 2 * row_chunks + R * 2 * CPP;
        for (int id = tid; id < p.NI * per_slot; id += NTHR) {
            const int i = id / per_slot, e = id - i * per_slot;
            int off;
            if (e < 2 * row_chunks) off = (e < row_chunks ? 0 : (R + 1) * RP) + (e % row_chunks) * 16;
            else {
                const int qq = e - 2 * row_chunks, r = qq / (2 * CPP), c = qq - r * (2 * CPP);
                off = (r + 1) * RP + (c < CPP ? 0 : (W + 1) * PS) + (c % CPP) * 16;
            }
            *reinterpret_cast<uint4*>(smem + i * slot_rows * RP + off) = make_uint4(0, 0, 0, 0);
            if (DUAL) *reinterpret_cast<uint4*>(smem + lds_bytes + i * slot_rows * RP + off) = make_uint4(0, 0, 0, 0);
        }
        for (int i = 0; i < p.NI; ++i)
            if (b0 + i >= p.B)
                for (int o = tid * 16; o < slot_rows * RP; o += NTHR * 16) {
                    *reinterpret_cast<uint4*>(smem + i * slot_rows * RP + o) = make_uint4(0, 0, 0, 0);
                    if (DUAL) *reinterpret_cast<uint4*>(smem + lds_bytes + i * slot_rows * RP + o) = make_uint4(0, 0, 0, 0);
                }
    }
    RSG_STAMP(1);
    if (p.wait) {       // hand-off inside a grouped launch: see chain_body
        const int t_lo = max(tile - 1, 0), t_hi = min(tile + 1, p.tiles_y - 1);
        if (tid <= t_hi - t_lo) {
            const int tp = t_lo + tid;
            int* f = const_cast<int*>(p.wait) + (((block_id / p.tiles_y) * p.tiles_y + tp) * 3 + (tile - tp + 1));
            int spin = 0;
            for (; spin < (1 << 20) && __hip_atomic_load(f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0; ++spin)
                __builtin_amdgcn_s_sleep(8);
            if (spin == (1 << 20)) atomicAdd(&g_chain_handoff_timeouts, 1u);
            else __hip_atomic_store(f, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        __syncthreads();
    }
    // ---- stage the window: the held rows of an image are one contiguous byte range (sc1 loads: see chain_body) -------------
    const int chunks_per_img = R * W * CPP;
    constexpr int NLD = 14;
    const __amdgpu_buffer_rsrc_t in_rsrc = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<char*>(p.in), 0, p.B * p.H * W * C * 2, 0x00020000);
    const int rowc = W * CPP;
    const int step_q = NTHR / rowc, step_r = NTHR - step_q * rowc;
    for (int i = 0; i < p.NI; ++i) {
        if (b0 + i >= p.B) break;
        const int gbase = ((b0 + i) * p.H + y_first) * W * C * 2;
        char* const slot = smem + i * slot_rows * RP;
        int r = tid / rowc, c = tid - r * rowc;
        for (int base = 0; base < chunks_per_img; base += NTHR * NLD) {
            uint4 v[NLD];
            int lo[NLD];
#pragma unroll
            for (int j = 0; j < NLD; ++j) {
                lo[j] = -1;
                if (r < R) {
                    v[j] = __builtin_bit_cast(uint4, __builtin_amdgcn_raw_buffer_load_b128(in_rsrc, gbase + (base + j * NTHR + tid) * 16, 0, 16));
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
    float* sbias = reinterpret_cast<float*>(smem + (DUAL ? 2 : 1) * lds_bytes);
    for (int l = 0; l < p.nlayers; ++l)
        for (int c = tid; c < C; c += NTHR) sbias[l * C + c] = p.bias[l][c];

    // ---- this lane's pixels: group g of the wave = pixels [16 (wn NG + g), + 16) of the tile in (image, row, x) order,
    // this lane's column of the MFMA = pixel pi(col) of them -----------------------------------------------------------
    const int npix = p.NI * R * W;
    const int pi = chain16_pi(col);
    const int sig = ((q & 1) << 1) | (q >> 1);
    int pix[NG];            // LDS byte address of the pixel's upper left tap + this lane's B chunk
#pragma unroll
    for (int g = 0; g < NG; ++g) {
        const int qi = (wn * NG + g) * 16 + pi;
        const int qq = qi < npix ? qi : 0;
        const int i = qq / (R * W), rem = qq - i * (R * W);
        const int r = rem / W, x = rem - r * W;
        pix[g] = (i * slot_rows + r) * RP + x * PS + sig * 16;      // the pixel's (-1, -1) tap: every tap offset is >= 0
    }
    // groups [0, glim) of this lane are pixels of the tile (the others multiply pixel 0 and write nothing)
    const int glim = (npix - pi + 15) / 16 - wn * NG;
    const int cb8 = CW * wm + 4 * MT * q;                // this lane's 4 MT consecutive output channels
    const int own = RP + PS + cb8 * 2 - sig * 16;        // pix[g] -> those channels of the pixel
    __syncthreads();
    asm volatile("" ::"v"(pf));                           // the grouped launch's weight prefetch loads end here
    RSG_STAMP(2);

    // A ring: NPF k-steps ahead (a step is NG x 32 matrix-pipe cycles per wave); B ring: NB fragments in flight.  Both
    // wrap at the end of a 32-channel block (9 taps x NG groups), so the body of one block can be a loop iteration.
    constexpr int NPF = (C == 256 && NG <= 3) ? 9 : 3;
    constexpr int NB = (C == 128 && DUAL) ? RSG_CHAIN16_NB128 : (NG % 7 == 0 ? 7 : (NG % 6 == 0 && (NG > 6 || MT == 4) ? 6 : 9));   // (ping-pong C = 128: 3 x 64 pipe cycles in flight fit 256 registers)
    constexpr int NJ = 9 * NG;                           // (tap, pixel group) pairs of a channel block, in issue order
    static_assert(9 % NPF == 0 && NJ % NB == 0, "the fragment rings wrap at the end of a channel block");
    uint4 aq[NPF][MT];
    {
        const uint4* ap = p.wpk[0] + (size_t)wm * KS * MT * 64 + lane;
#pragma unroll
        for (int k = 0; k < NPF; ++k)
#pragma unroll
            for (int m = 0; m < MT; ++m) aq[k][m] = ap[(MT * k + m) * 64];
    }
    // One layer; ODD = second conv of a BasicBlock.  xr: this lane's chunk of its pixels of the block input -- picked up by
    // the first conv's write-back just before it overwrites it (in place) and start value of the second conv's accumulators.
    auto layer = [&](const int l, auto odd_tag, uint4 (&xr)[NG][NX]) {
        constexpr bool ODD = decltype(odd_tag)::value;
        const char* const rd = smem + (DUAL ? (ODD ? lds_bytes : 0) : 0);
        char* const wr = smem + (DUAL ? (ODD ? 0 : lds_bytes) : 0);
        // opaque per layer: left loop-invariant, hipcc hoists pixel address + row offset for every (group, tap row) out of
        // the layer loop (3 NG registers)
        int rpl = RP;
        if (!WC) asm volatile("" : "+s"(rpl));
        auto toff = [&](int t) { return (t / 3) * rpl + (t % 3) * PS; };
        // ---- accumulators start at the folded-BN bias; second conv of a block: + the block input -------------------------
        f32x4 acc[NG][MT];
        {
            float bv[4 * MT];
#pragma unroll
            for (int m = 0; m < MT; ++m) {
                const float4 b4 = *reinterpret_cast<const float4*>(sbias + l * C + cb8 + 4 * m);
                bv[4 * m] = b4.x; bv[4 * m + 1] = b4.y; bv[4 * m + 2] = b4.z; bv[4 * m + 3] = b4.w;
            }
            if constexpr (ODD) {
                // ping-pong mode: the block input is still in the image this layer overwrites at its end; in batches of GB
                // pixel groups (all reads first would hold NG x NX x 4 registers beside the accumulators)
                constexpr int GB = DUAL ? 3 : NG;
#pragma unroll
                for (int g0 = 0; g0 < NG; g0 += GB) {
                    if constexpr (DUAL) {
#pragma unroll
                        for (int g = g0; g < g0 + GB && g < NG; ++g)
#pragma unroll
                            for (int c = 0; c < NX; ++c) xr[g][c] = *reinterpret_cast<const uint4*>(wr + pix[g] + own + 16 * c);
                    }
#pragma unroll
                    for (int g = g0; g < g0 + GB && g < NG; ++g)
#pragma unroll
                        for (int c = 0; c < NX; ++c) {
                            const f32x8 xf = __builtin_convertvector(__builtin_bit_cast(typename DT::vec8, xr[g][c]), f32x8);
#pragma unroll
                            for (int r = 0; r < 4; ++r) {
                                acc[g][2 * c][r] = bv[8 * c + r] + xf[r];
                                acc[g][2 * c + 1][r] = bv[8 * c + 4 + r] + xf[4 + r];
                            }
                        }
                    if constexpr (DUAL) __builtin_amdgcn_sched_barrier(0);
                }
            } else {
#pragma unroll
                for (int g = 0; g < NG; ++g)
#pragma unroll
                    for (int m = 0; m < MT; ++m)
#pragma unroll
                        for (int r = 0; r < 4; ++r) acc[g][m][r] = bv[4 * m + r];
            }
        }
        const uint4* const ap = p.wpk[l] + (size_t)wm * KS * MT * 64 + lane;
        const uint4* const apn = p.wpk[l + 1 < p.nlayers ? l + 1 : l] + (size_t)wm * KS * MT * 64 + lane;
        RSG_STAMP(3 + 4 * l);
        // ---- k-loop.  Issue order pinned per k-step (left alone, hipcc sinks the loads next to their uses): the two A
        // loads of step s + NPF, then per pixel group its two MFMAs and the B read that refills the ring slot they freed
        // (fragment j + NB of the block's (tap, group) sequence; the last NB of a block fetch the next block's first).
        uint4 bq[NB];
#pragma unroll
        for (int j = 0; j < NB; ++j) bq[j] = *reinterpret_cast<const uint4*>(rd + pix[j % NG] + toff(j / NG));
        auto block = [&](const int cb, auto more_tag) {
            constexpr bool MORE = decltype(more_tag)::value;    // another channel block of this layer follows
            const char* const rdk = rd + cb * 64;
            const uint4* const apk = ap + cb * (9 * MT * 64);
            const uint4* const nxt = MORE ? apk + 9 * MT * 64 : apn;
#pragma unroll
            for (int t = 0; t < 9; ++t) {
                // the row offset of the fragments issued in this step stays a scalar operand of one add per read: as
                // loop invariants hipcc keeps pixel address + row offset for every (group, tap row) in registers (3 NG)
                int ro[3] = {0, rpl, 2 * rpl};
                if (!WC) asm volatile("" : "+s"(ro[1]), "+s"(ro[2]));
                auto roff = [&](int tt) { return ro[tt / 3] + (tt % 3) * PS; };
                typename DT::vec8 a[MT];
#pragma unroll
                for (int m = 0; m < MT; ++m) a[m] = __builtin_bit_cast(typename DT::vec8, aq[t % NPF][m]);
#pragma unroll
                for (int g = 0; g < NG; ++g) {
                    const int j = t * NG + g;
                    const typename DT::vec8 b = __builtin_bit_cast(typename DT::vec8, bq[j % NB]);
#pragma unroll
                    for (int m = 0; m < MT; ++m) acc[g][m] = Mfma16<DT>::run(a[m], b, acc[g][m]);
                    if (j + NB < NJ) bq[j % NB] = *reinterpret_cast<const uint4*>(rdk + pix[(j + NB) % NG] + roff((j + NB) / NG));
                    else if (MORE) bq[j % NB] = *reinterpret_cast<const uint4*>(rdk + 64 + pix[(j + NB - NJ) % NG] + roff((j + NB - NJ) / NG));
                }
                // the A fragments of step t + NPF go into the slot this step's MFMAs have just read (issued behind them: no copy)
#pragma unroll
                for (int m = 0; m < MT; ++m)
                    aq[t % NPF][m] = (t + NPF < 9) ? apk[(MT * (t + NPF) + m) * 64] : nxt[(MT * (t + NPF - 9) + m) * 64];
#pragma unroll
                for (int g = 0; g < NG; ++g) {
                    __builtin_amdgcn_sched_group_barrier(0x008, MT, 0);         // MT MFMAs
                    __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);          // 1 DS read
                }
                __builtin_amdgcn_sched_group_barrier(0x020, MT, 0);             // the A prefetches (VMEM reads)
                __builtin_amdgcn_sched_barrier(0);
            }
        };
        if constexpr (NCB * NJ <= 256) {
#pragma unroll
            for (int cb = 0; cb + 1 < NCB; ++cb) block(cb, std::true_type{});
        } else {
#pragma unroll 1
            for (int cb = 0; cb + 1 < NCB; ++cb) block(cb, std::true_type{});
        }
        block(NCB - 1, std::false_type{});
        RSG_STAMP(4 + 4 * l);
        int wd = own;
        if (!WC) asm volatile("" : "+v"(wd));            // keep the NG write addresses out of the loop preheader
        // ---- ReLU -> 16 bit: one packed convert and one packed max per two elements -----------------------------------------
        uint4 o[NG][NX];
#pragma unroll
        for (int g = 0; g < NG; ++g)
#pragma unroll
            for (int c = 0; c < NX; ++c) {
                f32x8 v;
#pragma unroll
                for (int r = 0; r < 4; ++r) { v[r] = acc[g][2 * c][r]; v[4 + r] = acc[g][2 * c + 1][r]; }
                o[g][c] = __builtin_bit_cast(uint4, DT::relu_pack(v));
                // pinned in front of the barrier: left alone, hipcc sinks the conversions into the predicated stores BEHIND it,
                // where every wave then converts while the image waits (they belong under the wait for the slowest wave)
                asm volatile("" ::"v"(o[g][c].x), "v"(o[g][c].y), "v"(o[g][c].z), "v"(o[g][c].w));
            }
        if constexpr (!DUAL && !ODD) {
            // first conv of a block, in place: pick up the block input (this lane's chunk of its own pixels) before the
            // write-back overwrites it.  Nobody writes the image before the barrier below, so the reads go out in front of
            // it (behind the conversions: both sets of values and the accumulators do not fit the registers together).
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int g = 0; g < NG; ++g)
#pragma unroll
                for (int c = 0; c < NX; ++c) xr[g][c] = *reinterpret_cast<const uint4*>(wr + pix[g] + wd + 16 * c);
        }
        RSG_STAMP(5 + 4 * l);
        if (ODD && l + 1 == p.nlayers) {
            // the chain's last layer goes straight from the accumulators to memory (the TH output rows of the tile; sc1:
            // written through, see the hand-off): no write-back into the image, no barrier, no second pass over LDS.  A
            // lane's 16 bytes are a chunk of its pixel; a wave's store covers 16 pixels x 64 B.
            const __amdgpu_buffer_rsrc_t out_rsrc = __builtin_amdgcn_make_buffer_rsrc(p.out, 0, p.B * p.H * W * C * 2, 0x00020000);
            const int r_lo = tile * p.TH - y_first, r_hi = r_lo + min(p.TH, p.H - tile * p.TH);
            // (opaque: as loop invariants hipcc computes the 14 addresses and their predicates in front of the layer loop and
            // spills them across it)
            int tl = threadIdx.x, RWq = R * W, Wq = W;
            asm volatile("" : "+v"(tl), "+s"(RWq), "+s"(Wq));                  // (and the division constants)
            const int pi2 = chain16_pi(tl & 15);
            const int q0 = wn * NG * 16 + pi2, gl = (npix - pi2 + 15) / 16 - wn * NG;
            int i = q0 / RWq, rem = q0 - i * RWq;
            int r = rem / Wq, x = rem - r * Wq;
#pragma unroll
            for (int g = 0; g < NG; ++g) {
                if (g < gl && b0 + i < p.B && r >= r_lo && r < r_hi) {
#pragma unroll
                    for (int c = 0; c < NX; ++c)
                        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, o[g][c]), out_rsrc,
                                                               (((b0 + i) * p.H + y_first + r) * W + x) * (C * 2) + cb8 * 2, 16 * c, 16);
                }
                x += 16;
#pragma unroll
                for (int k = 0; k < 3; ++k)               // a 16-pixel step crosses at most three rows of the narrowest map (6 pixels)
                    if (x >= W) { x -= W; if (++r == R) { r = 0; ++i; } }
            }
            RSG_STAMP(6 + 4 * l);
            return;
        }
        if (!DUAL) chain16_barrier();                    // in place: every wave has finished reading the image
#pragma unroll
        for (int g = 0; g < NG; ++g)
            if (g < glim) {
#pragma unroll
                for (int c = 0; c < NX; ++c) *reinterpret_cast<uint4*>(wr + pix[g] + wd + 16 * c) = o[g][c];
            }
        chain16_barrier();                               // the new image is complete
        RSG_STAMP(6 + 4 * l);
    };
    // ---- halo exchange (ChainArgs::xchg, in-place images only): in front of layer `xchg` the tile's halo rows are as stale as a
    // halo of xchg rows gets (one row per layer).  Every tile copies the rows its neighbours lack -- its own first / last
    // `halo` output rows, which ARE valid -- from the LDS image to xbuf (written through), raises the neighbour's word, waits
    // for the neighbours' words and copies their rows over its halo.  The poll is bounded like the hand-off's (a tile whose
    // neighbour never shows up goes on with stale rows and the error word counts it); the words clear themselves.
    auto exchange = [&]() {
        const int XR = p.halo, rowc = W * CPP, chunks = XR * rowc;
        const __amdgpu_buffer_rsrc_t xrs = __builtin_amdgcn_make_buffer_rsrc(p.xbuf, 0, p.B * p.tiles_y * 2 * chunks * 16, 0x00020000);
        const int first = block_id - tile;                // workgroup of the picture's tile 0
#pragma unroll 1
        for (int side = 0; side < 2; ++side) {            // 0: for the tile above, 1: for the tile below
            const int nb = tile + (side ? 1 : -1);
            if (nb < 0 || nb >= p.tiles_y) continue;      // (uniform)
            const int r0 = (side ? min((tile + 1) * p.TH, p.H) - XR : tile * p.TH) - y_first;
            for (int i = tid; i < chunks; i += NTHR) {
                const int r = i / rowc, c = i - r * rowc;
                const uint4 v = *reinterpret_cast<const uint4*>(smem + (r0 + r + 1) * RP + PS + (c / CPP) * PS + (c % CPP) * 16);
                __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v), xrs, ((block_id * 2 + side) * chunks + i) * 16, 0, 16);
            }
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // every wave: its stores have left the CU
        __syncthreads();
        if (tid < 2) {
            const int nb = tile + (tid ? 1 : -1);
            if (nb >= 0 && nb < p.tiles_y) {
                if (!(p.debug & 4)) __hip_atomic_store(p.xflag + block_id * 3 + tid, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                // the tile above raises ITS word 1 (rows for the tile below it), the tile below its word 0
                int* f = p.xflag + (first + nb) * 3 + (tid ? 0 : 1);
                int spin = 0;
                for (; spin < (1 << 20) && __hip_atomic_load(f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0; ++spin)
                    __builtin_amdgcn_s_sleep(8);
                if (spin == (1 << 20)) atomicAdd(&g_chain_handoff_timeouts, 1u);
                else __hip_atomic_store(f, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
        }
        __syncthreads();
#pragma unroll 1
        for (int side = 0; side < 2; ++side) {            // rows FROM the tile above (0) / below (1): that tile's other side
            const int nb = tile + (side ? 1 : -1);
            if (nb < 0 || nb >= p.tiles_y) continue;
            const int r0 = (side ? (tile + 1) * p.TH : tile * p.TH - XR) - y_first;
            for (int i = tid; i < chunks; i += NTHR) {
                const int r = i / rowc, c = i - r * rowc;
                const uint4 v = __builtin_bit_cast(uint4, __builtin_amdgcn_raw_buffer_load_b128(
                    xrs, (((first + nb) * 2 + (side ? 0 : 1)) * chunks + i) * 16, 0, 16));
                *reinterpret_cast<uint4*>(smem + (r0 + r + 1) * RP + PS + (c / CPP) * PS + (c % CPP) * 16) = v;
            }
        }
        __syncthreads();
    };
#pragma unroll 1
    for (int l = 0; l < p.nlayers; l += 2) {             // one BasicBlock per iteration
        if (!DUAL && p.xchg && l == p.xchg) exchange();
        uint4 xr[NG][NX];
        layer(l, std::false_type{}, xr);
        layer(l + 1, std::true_type{}, xr);
    }
    if (p.done) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");     // every wave: its stores have left the CU
        __syncthreads();
        const int t3 = threadIdx.x;
        if (t3 < 3 && !(p.debug & 4))      // debug bit 4 (tests): the producer never signals, every consumer poll gives up
            __hip_atomic_store(p.done + block_id * 3 + t3, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    RSG_STAMP(62);
}
