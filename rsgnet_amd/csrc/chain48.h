// One BasicBlock of HRNet-W48's 48-channel branch (pose_hrnet.py:30-59; HighResolutionModule.forward :254-255) on the
// 16x16x32 MFMA -- included by chain.hip (ChainArgs, RSG_STAMP).
//
// 48 channels are 1.5 tiles of the 32x32x16 MFMA every other conv kernel here uses: chain_body<48> pads the OUTPUT
// channels to 64 (a quarter of its matrix work multiplies zero weight rows) and, at 96 accumulator registers for six pixel
// groups, cannot hold more than 13 rows of a 96x72 map per workgroup (45 % halo rows at a 2-row halo).  The 16x16x32 form
// fits exactly: three 16-channel output tiles, and the contraction index runs over the FLATTENED (tap, input channel)
// pairs, k = 48 * tap + cin -- 432 = 13.5 steps of 32, so 14 steps with 3.7 % padding.  A lane's 8 consecutive k of a step
// stay inside one tap (48 % 8 == 0), so its B fragment is still one 16-byte LDS read of its pixel; which tap it belongs to
// depends on the lane group (k = 32 s + 8 (lane >> 4) + j), a few integer instructions per step.  A wave owns all three
// output tiles of its pixels (the B fragment is read once for three MFMAs) and 6 groups of 16 pixels: 72 accumulator
// registers, 16 rows of a 72-pixel-wide map per workgroup (12 output rows + a 2-row halo on each side).
//
// Layout: the LDS image, the zero borders, staging and the final store are chain_body's (pixel stride 112 B, maps in
// buffers with their own pixel strides).  Exactly TWO layers (one BasicBlock): the block input is this launch's own
// input, so the second layer's epilogue reads it back from global memory (L2-hot) -- no workspace.
// Weights: rsg_pack_chain48_weights_host -- [k-step 14][output tile 3][lane 64] x 16 B, lane l holds
// W[16 m + (l & 15)][k = 32 s + 8 (l >> 4) + j], zero for k >= 432.
#pragma once

typedef _Float16 h4_t __attribute__((ext_vector_type(4)));
typedef __bf16 b4_t __attribute__((ext_vector_type(4)));
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));

template <class DT> struct Mfma16;
template <> struct Mfma16<F16> {
    static __device__ __forceinline__ f32x4 run(h8_t a, h8_t b, f32x4 c) { return __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c, 0, 0, 0); }
    static __device__ __forceinline__ uint2 relu_pack(f32x4 v) {
        const h4_t z = {0, 0, 0, 0};
        return __builtin_bit_cast(uint2, __builtin_elementwise_max(__builtin_convertvector(v, h4_t), z));
    }
    static __device__ __forceinline__ f32x4 widen(uint2 v) { return __builtin_convertvector(__builtin_bit_cast(h4_t, v), f32x4); }
};
template <> struct Mfma16<BF16> {
    static __device__ __forceinline__ f32x4 run(b8_t a, b8_t b, f32x4 c) { return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0); }
    static __device__ __forceinline__ uint2 relu_pack(f32x4 v) {       // see BF16::relu_pack (common.h)
        typedef short s4_t __attribute__((ext_vector_type(4)));
        const s4_t z = {0, 0, 0, 0};
        return __builtin_bit_cast(uint2, __builtin_elementwise_max(__builtin_bit_cast(s4_t, __builtin_convertvector(v, b4_t)), z));
    }
    static __device__ __forceinline__ f32x4 widen(uint2 v) { return __builtin_convertvector(__builtin_bit_cast(b4_t, v), f32x4); }
};

#define RSG_CHAIN48_KS 14
#define RSG_CHAIN48_NTP 6

template <class DT>
__device__ __forceinline__ void chain48_body(const ChainArgs& p, const int block_id, char* smem, const unsigned pf = 0) {
    constexpr int C = 48, PS = 2 * C + 16, CPP = C / 8, NTHR = 768, NTP = RSG_CHAIN48_NTP, KS = RSG_CHAIN48_KS, MT = 3;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int col = lane & 15, q = lane >> 4;
    const int W = p.W, R = p.R, RP = p.RP, slot_rows = R + 2;
    const int tile = block_id % p.tiles_y;
    const int b0 = (block_id / p.tiles_y) * p.NI;
    const int y_first = min(max(tile * p.TH - p.halo, 0), p.H - R);

    // ---- zero borders (see chain_body) ------------------------------------------------------------------------------
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
        }
        for (int i = 0; i < p.NI; ++i)
            if (b0 + i >= p.B)
                for (int o = tid * 16; o < slot_rows * RP; o += NTHR * 16) *reinterpret_cast<uint4*>(smem + i * slot_rows * RP + o) = make_uint4(0, 0, 0, 0);
    }
    // ---- stage the window --------------------------------------------------------------------------------------------
    const int ldi2 = p.ldi * 2, ldo2 = p.ldo * 2;
    const __amdgpu_buffer_rsrc_t in_rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(p.in), 0, p.B * p.H * W * ldi2, 0x00020000);
    const int rowc = W * CPP;
    const int step_q = NTHR / rowc, step_r = NTHR - step_q * rowc;
    constexpr int NLD = 10;
    for (int i = 0; i < p.NI; ++i) {
        if (b0 + i >= p.B) break;
        const int gbase = ((b0 + i) * p.H + y_first) * W * ldi2;
        char* const slot = smem + i * slot_rows * RP;
        int r = tid / rowc, c = tid - r * rowc;
        for (int base = 0; base < R * rowc; base += NTHR * NLD) {
            uint4 v[NLD];
            int lo[NLD];
#pragma unroll
            for (int j = 0; j < NLD; ++j) {
                lo[j] = -1;
                if (r < R) {
                    v[j] = __builtin_bit_cast(uint4, __builtin_amdgcn_raw_buffer_load_b128(in_rsrc, gbase + (r * W + c / CPP) * ldi2 + (c % CPP) * 16, 0, 16));
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
    // ---- this lane's pixels ------------------------------------------------------------------------------------------
    int pix[NTP];          // LDS byte address of the pixel (centre tap)
    const int npix = p.NI * R * W;
    // byte offset of pixel group g's pixel in the input map (the block input, for the residual); out of the buffer's range
    // for pixels this workgroup does not own: a raw buffer load returns zeros there, no branch.  Recomputed where it is
    // needed (second epilogue) -- six more registers across both k-loops do not fit.
#pragma unroll
    for (int g = 0; g < NTP; ++g) {
        const int qi = (wave * NTP + g) * 16 + col;
        const int qq = qi < npix ? qi : 0;
        const int i = qq / (R * W), rem = qq - i * (R * W);
        const int r = rem / W, x = rem - r * W;
        pix[g] = (i * slot_rows + r + 1) * RP + (x + 1) * PS;
    }
    __syncthreads();
    asm volatile("" ::"v"(pf));

    // k-step s of this lane: flattened k0 = 32 s + 8 q -> (tap, cin), as the byte offset of its 16-byte B fragment from the
    // centre tap of the pixel.  With s = 3 t + u: k0 = 96 t + 32 u + 8 q, i.e. tap 2t for u = 0 (cin 8q), tap 2t + 1 for
    // u = 2 (cin 16 + 8q), and for u = 1 tap 2t (cin 32 + 8q) in the lower two lane groups, tap 2t + 1 (cin 8q - 16) in the
    // upper two: one select per step, the tap offsets themselves are wave-uniform.  k >= 432 (s = 13, upper lane groups)
    // multiplies zero weights: the centre tap stands in.
    const bool hi = q >= 2;
    const int c0 = 16 * q, c1 = hi ? 16 * (q - 2) : 64 + 16 * q, c2 = 32 + 16 * q;     // byte offsets of cin
    constexpr int NPF = 1;                                   // weight-fragment ring: k-steps ahead (2 does not fit the registers)
    uint4 aq[NPF][MT];
#pragma unroll
    for (int l = 0; l < 2; ++l) {
        // opaque per layer: left loop-invariant, hipcc hoists pixel address + offset for all 14 steps x 6 groups out of the
        // layer loop (84 registers, spilled)
        int rpl = RP, ql = lane >> 4;
        asm volatile("" : "+s"(rpl), "+v"(ql));
        const int c0l = 16 * ql, c1l = ql >= 2 ? 16 * ql - 32 : 64 + 16 * ql, c2l = 32 + 16 * ql;
        auto tapoff = [&](int tap) { return (tap / 3 - 1) * rpl + (tap % 3 - 1) * PS; };
        auto koff = [&](int s) {
            const int t = s / 3, u = s % 3;
            if (u == 0) return tapoff(2 * t) + c0l;
            if (u == 2) return tapoff(2 * t + 1) + c2l;
            const int ta = tapoff(2 * t + 1 < 9 ? 2 * t + 1 : 4), tb = tapoff(2 * t);      // wave-uniform
            return (ql >= 2 ? ta : tb) + c1l;
        };
        f32x4 acc[NTP][MT];
        {
            f32x4 bias[MT];
#pragma unroll
            for (int m = 0; m < MT; ++m) bias[m] = *reinterpret_cast<const f32x4*>(p.bias[l] + 16 * m + 4 * q);
#pragma unroll
            for (int g = 0; g < NTP; ++g)
#pragma unroll
                for (int m = 0; m < MT; ++m) acc[g][m] = bias[m];
        }
        int ln = lane;
        asm volatile("" : "+v"(ln));                         // (pointers rebuilt per layer instead of carried in registers)
        const uint4* ap = p.wpk[l] + ln;
        const uint4* apn = p.wpk[1] + ln;
        if (l == 0) {
#pragma unroll
            for (int k = 0; k < NPF; ++k)
#pragma unroll
                for (int m = 0; m < MT; ++m) aq[k][m] = ap[(k * MT + m) * 64];
        }
        uint4 bq[NTP];
        {
            const int o0 = koff(0);
#pragma unroll
            for (int g = 0; g < NTP; ++g) bq[g] = *reinterpret_cast<const uint4*>(smem + pix[g] + o0);
        }
#pragma unroll
        for (int s = 0; s < KS; ++s) {
            typename DT::vec8 a[MT];
#pragma unroll
            for (int m = 0; m < MT; ++m) a[m] = __builtin_bit_cast(typename DT::vec8, aq[s % NPF][m]);
#pragma unroll
            for (int m = 0; m < MT; ++m) {
                // the ring keeps running across the layer boundary (layer 1's first steps behind layer 0's last)
                // (layer 1 re-reads its own first fragments at its end: valid memory, never used)
                aq[s % NPF][m] = (s + NPF < KS) ? ap[((s + NPF) * MT + m) * 64] : apn[((s + NPF - KS) * MT + m) * 64];
            }
            const int on = s + 1 < KS ? koff(s + 1) : 0;
#pragma unroll
            for (int g = 0; g < NTP; ++g) {
#pragma unroll
                for (int m = 0; m < MT; ++m)
                    acc[g][m] = Mfma16<DT>::run(a[m], __builtin_bit_cast(typename DT::vec8, bq[g]), acc[g][m]);
                if (s + 1 < KS) bq[g] = *reinterpret_cast<const uint4*>(smem + pix[g] + on);
            }
            __builtin_amdgcn_sched_group_barrier(0x020, MT, 0);             // the A prefetches (VMEM reads)
#pragma unroll
            for (int g = 0; g < NTP; ++g) {
                __builtin_amdgcn_sched_group_barrier(0x008, MT, 0);         // 3 MFMAs
                __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);          // 1 DS read
            }
            __builtin_amdgcn_sched_barrier(0);
        }
        // ---- epilogue: (+ block input) -> ReLU -> 16 bit, 4 channels (8 bytes) per output tile -------------------------
        uint2 o[NTP][MT];
        if (l == 1) {
            // the block input arrives one or two pixel groups ahead of its use (all 18 loads at once do not fit the registers)
            uint2 rv[NTP][MT];
            int qo = 8 * q;                                   // opaque: keeps the 18 addresses out of the loop preheader
            asm volatile("" : "+v"(qo));
            int wv = wave, Wq = W, RWq = R * W;
            asm volatile("" : "+s"(wv), "+s"(Wq), "+s"(RWq));   // (and the pixel decomposition itself, with its division constants)
            auto fetch = [&](int g) {
                const int qi = (wv * NTP + g) * 16 + col;
                const bool live = qi < npix;
                const int qq = live ? qi : 0;
                const int i = qq / RWq, rem = qq - i * RWq;
                const int r = rem / Wq, x = rem - r * Wq;
                const int gin = (live && b0 + i < p.B) ? (((b0 + i) * p.H + y_first + r) * Wq + x) * ldi2 : p.B * p.H * Wq * ldi2;
#pragma unroll
                for (int m = 0; m < MT; ++m) {
                    const u32x2 t = __builtin_bit_cast(u32x2, __builtin_amdgcn_raw_buffer_load_b64(in_rsrc, gin + qo, 32 * m, 0));
                    rv[g][m] = make_uint2(t[0], t[1]);
                }
            };
            constexpr int LEAD = 2;
#pragma unroll
            for (int g = 0; g < LEAD; ++g) fetch(g);
#pragma unroll
            for (int g = 0; g < NTP; ++g) {
                if (g + LEAD < NTP) fetch(g + LEAD);
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int m = 0; m < MT; ++m) {
                    const f32x4 rf = Mfma16<DT>::widen(rv[g][m]);
                    f32x4 v = acc[g][m];
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[e] += rf[e];
                    o[g][m] = Mfma16<DT>::relu_pack(v);
                }
                __builtin_amdgcn_sched_barrier(0);
            }
        } else {
#pragma unroll
            for (int g = 0; g < NTP; ++g)
#pragma unroll
                for (int m = 0; m < MT; ++m) o[g][m] = Mfma16<DT>::relu_pack(acc[g][m]);
        }
        __syncthreads();                                     // every wave has finished reading the image
        int wo = 8 * q;
        asm volatile("" : "+v"(wo));                         // (the same for the 18 write addresses)
#pragma unroll
        for (int g = 0; g < NTP; ++g) {
            if ((wave * NTP + g) * 16 + col < npix) {
#pragma unroll
                for (int m = 0; m < MT; ++m) *reinterpret_cast<uint2*>(smem + pix[g] + wo + 32 * m) = o[g][m];
            }
        }
        __syncthreads();                                     // the new image is complete
    }
    // ---- store the TH output rows --------------------------------------------------------------------------------------
    const __amdgpu_buffer_rsrc_t out_rsrc = __builtin_amdgcn_make_buffer_rsrc(p.out, 0, p.B * p.H * W * ldo2, 0x00020000);
    const int out_rows = min(p.TH, p.H - tile * p.TH);
    int ts = threadIdx.x, Ws = W;
    asm volatile("" : "+v"(ts), "+s"(Ws));                  // the store loop's division constants are built here, not before the k-loops
    const int rowcs = Ws * CPP, sq = NTHR / rowcs, sr = NTHR - sq * rowcs;
    for (int i = 0; i < p.NI; ++i) {
        if (b0 + i >= p.B) break;
        const int gbase = ((b0 + i) * p.H + tile * p.TH) * Ws * ldo2;
        const char* const slot = smem + (i * slot_rows + tile * p.TH - y_first + 1) * RP + PS;
        int r = ts / rowcs, c = ts - r * rowcs;
        for (int k = ts; k < out_rows * rowcs; k += NTHR) {
            const uint4 v = *reinterpret_cast<const uint4*>(slot + r * RP + (c / CPP) * PS + (c % CPP) * 16);
            __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v), out_rsrc, gbase + (r * Ws + c / CPP) * ldo2 + (c % CPP) * 16, 0, 16);
            r += sq; c += sr;
            if (c >= rowcs) { c -= rowcs; ++r; }
        }
    }
}
