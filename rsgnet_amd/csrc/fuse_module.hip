// Level 0 of a HighResolutionModule's fuse layers as ONE launch of row bands: every conv that reads a branch map directly
// and the sums of the two highest-resolution outputs.
//
// Replaces, of  x_fuse[i] = relu(sum_j fuse_layers[i][j](x[j]))  (/root/reference/lib/models/pose_hrnet.py:190-268):
//   * the 1x1 "up" convs j -> i (j > i; :199-213; BatchNorm folded, no ReLU, nearest up-sampling applied by the reader),
//   * the FIRST 3x3 stride-2 conv of every "down" chain j -> i (j < i; :217-243),
//   * y0 = relu(x0 + up(t10) + up(t20) + up(t30)) and y1 = relu(conv01(x0) + x1 + up(t21) + up(t31))  (:259-266).
// The grouped launches (group.hip) ran these as up to 12 independent convs + 2 sums: the 64x48x32 map was staged by four
// workgroups' worth of convs and read again by the sum, every workgroup a chain of four dependent memory round trips at
// 2.5 workgroups per CU (3.0x the algorithmic bytes, 47 % of the achievable HBM rate).
//
// Here a workgroup of 16 waves owns a BAND of R0 (16) rows of branch 0 = R0 >> i rows of branch i -- the bands of the four
// maps nest exactly, so only the 3x3 convs need one halo row above -- and
//   1. requests every byte it will read in ONE round (<= 8 x 16 B per thread, ~100 KB per workgroup in flight), into LDS
//      images [row][slot][C + 8 pad] (a map that feeds a stride-2 conv: one zero column on the left, even and odd columns
//      in separate halves of a row so that consecutive output pixels read consecutive slots);
//   2. phase A: the 1x1 convs whose results the sums need, 16-bit into LDS (what the grouped path stored to memory);
//   3. phase B: the y0 sum straight from LDS to memory, then the strided convs, each wave a static list of jobs
//      (conv, 32-channel output tile, <= 3 groups of 32 pixels) dealt by the host (longest first); the 0 -> 1 conv ends in the
//      y1 sum (operands from LDS), the others store 16-bit maps for the module's later levels.
// Arithmetic, k order and rounding points are the grouped path's (conv_body / fuse_body): results are bit-identical
// (tests/test_gpu_kernels.py::test_fuse_module_equals_the_grouped_launches).
#include <algorithm>
#include <type_traits>
#include <vector>
#include "common.h"

#define FM_MAX_CONVS RSG_FM_MAX_CONVS
#define FM_MAX_JOBS 48
#define FM_WAVES 16
#define FM_THREADS (64 * FM_WAVES)
#define FM_SLOTS 10                // staging loads in flight per thread
#define FM_SYNC 255                // pseudo-job: barrier (phase A's results are in LDS), then this thread's share of the y0 sum

struct FmConv {
    const uint4* w;
    const float* bias;
    char* out;             // role TO_GLOBAL: 16-bit map at the output's level
    int ldo;
    int src, taps, cout, relu;
    int lvl;               // level (branch index) whose resolution the output has
    int role;
    int t_off;             // LDS byte offset of the 16-bit result [pixel][cout] (roles UP_Y0 / UP_Y1)
    int ks;                // k-steps of 16
    int cout_pad;          // cout rounded up to the 32-channel output tiles (48 -> 64: the upper half of tile 1 is padding)
};
struct FmJob { unsigned char conv, ct, g0, ng; };
struct FmArgs {
    const char* x[4];
    char* y[2];
    int ldx[4], ldy[2];
    int B, H, W, nbr, R0;
    int C0;                          // channels of branch 0 (32: HRNet-W32, 48: HRNet-W48); branch i has C0 << i
    int y1_fold, conv_y1;
    int t0_off[3], t1_off[3];        // LDS offsets of the 1x1 results j -> 0 / j -> 1 by j - 1
    int n_conv;
    int off_x[4], rp[4], halo[4];
    // staging slot s of a thread: chunk (slot_q[s] rows, slot_m[s] chunks) + its own (tid / rowc, tid % rowc) of branch slot_br[s]
    int slot_br[FM_SLOTS], slot_q[FM_SLOTS], slot_m[FM_SLOTS];
    int pre_total, pre_first[FM_MAX_CONVS];   // weight prefetch: 128-byte lines of all convs / first line of conv c (0 lines = off)
    int first[FM_WAVES + 1];         // wave w runs job[first[w] .. first[w + 1]): phase A jobs, FM_SYNC, phase B jobs
    FmConv conv[FM_MAX_CONVS];
    FmJob job[2 * FM_MAX_JOBS];
};

#ifdef RSG_FM_STAMPS
static __device__ unsigned long long* g_fm_stamps = nullptr;      // [blocks][16 waves][8]
#define FM_STAMP(slot)                                                                               \
    do {                                                                                             \
        if (g_fm_stamps && (threadIdx.x & 63) == 0) {                                                \
            unsigned long long t__;                                                                  \
            asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t__)::"memory");             \
            g_fm_stamps[((size_t)blockIdx.x * FM_WAVES + (threadIdx.x >> 6)) * 8 + (slot)] = t__;   \
        }                                                                                            \
    } while (0)
extern "C" int rsg_fuse_module_set_stamp_buffer(void* dev_ptr) {
    unsigned long long* q = (unsigned long long*)dev_ptr;
    if (hipMemcpyToSymbol(HIP_SYMBOL(g_fm_stamps), &q, sizeof(q)) != hipSuccess) RSG_FAIL("fuse_module stamps: memcpy failed");
    return 0;
}
#else
#define FM_STAMP(slot) do {} while (0)
#endif

// slot of input column x in a row of a de-interleaved image (tile column x + 1: column 0 is the zero padding)
__device__ __forceinline__ int fm_slot(int x, int hws) { return ((x + 1) & 1) * hws + ((x + 1) >> 1); }

// One job: NG groups of 32 output pixels x one 32-channel output tile of one conv, operands from the LDS images.
// k order of rsg_pack_conv_weights_host: input-channel block (CINB = 16 KPT channels: 64, 48 or 32 by the conv's cin), then
// tap, then the KPT k-steps of the block.  NCH blocks; KS = NCH * TAPS * KPT.
template <class DT, int TAPS, int NCH, int KPT, int NG>
__device__ __forceinline__ void fm_job(const FmArgs& p, const FmConv& cv, const int ct, const int g0, char* smem,
                                       const int b, const int band, const int lane) {
    constexpr int KS = TAPS * NCH * KPT;
    constexpr int NPF = KS < 6 ? KS : (NG == 3 ? 4 : 6);  // weight fragments in flight
    const int n = lane & 31, hh = lane >> 5;
    const int lvl = cv.lvl, s = cv.src;
    const int OW = p.W >> lvl, OR = p.R0 >> lvl, npx = OW * OR;
    const int PSs = ((2 * p.C0) << s) + 16, RPs = p.rp[s], HWs = ((p.W >> s) + 2) >> 1, hs = p.halo[s];
    const char* const src = smem + p.off_x[s];
    const uint4* const ap = cv.w + (size_t)ct * KS * 64 + lane;
    uint4 aq[NPF];
#pragma unroll
    for (int k = 0; k < NPF; ++k) aq[k] = ap[k * 64];

    int pixoff[NG];
#pragma unroll
    for (int g = 0; g < NG; ++g) {
        const int q = (g0 + g) * 32 + n;
        const int qw = q < npx ? q : q % npx;            // idle lanes: a live pixel's address (product discarded)
        const int y = qw / OW, x = qw - y * OW;
        if constexpr (TAPS == 9) {
            pixoff[g] = (2 * y) * RPs + x * PSs + hh * 16;               // tap (ky, kx) adds ky rows / the column half
        } else {
            const int slot = hs ? fm_slot(x, HWs) : x;
            pixoff[g] = (y + hs) * RPs + slot * PSs + hh * 16;
        }
    }
    const int cbase = ct * 32 + hh * 16;
    f32x16 acc[NG];
    {
        float bias[16];
#pragma unroll
        for (int r4 = 0; r4 < 4; ++r4) {
            const float4 bv = *reinterpret_cast<const float4*>(cv.bias + cbase + r4 * 4);
            bias[r4 * 4 + 0] = bv.x; bias[r4 * 4 + 1] = bv.y; bias[r4 * 4 + 2] = bv.z; bias[r4 * 4 + 3] = bv.w;
        }
#pragma unroll
        for (int g = 0; g < NG; ++g)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[g][r] = bias[r];
    }
    auto boff = [&](int ks) {
        const int ch = ks / (TAPS * KPT), rem = ks % (TAPS * KPT), t = rem / KPT, c = rem % KPT;
        if constexpr (TAPS == 1) return (ch * KPT + c) * 32;
        else {
            const int ky = t / 3, kx = t % 3;
            return ky * RPs + ((kx & 1) * HWs + (kx >> 1)) * PSs + (ch * KPT + c) * 32;
        }
    };
    uint4 bq[2][NG];
#pragma unroll
    for (int g = 0; g < NG; ++g) bq[0][g] = *reinterpret_cast<const uint4*>(src + pixoff[g] + boff(0));
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
        const typename DT::vec8 a = __builtin_bit_cast(typename DT::vec8, aq[ks % NPF]);
        if (ks + 1 < KS) {
#pragma unroll
            for (int g = 0; g < NG; ++g) bq[(ks + 1) & 1][g] = *reinterpret_cast<const uint4*>(src + pixoff[g] + boff(ks + 1));
        }
        if (ks + NPF < KS) aq[ks % NPF] = ap[(ks + NPF) * 64];
        __builtin_amdgcn_sched_barrier(0);               // reads issue before the MFMAs (conv_body)
#pragma unroll
        for (int g = 0; g < NG; ++g) acc[g] = DT::mfma(a, __builtin_bit_cast(typename DT::vec8, bq[ks & 1][g]), acc[g]);
        __builtin_amdgcn_sched_barrier(0);
    }

    // ---- epilogue by role (wave-uniform) ----------------------------------------------------------------------------------
    // (the pixel coordinates are divided out again here: held across the k-loop they cost three registers per group)
    const int role = cv.role;
    int qq[NG], py[NG], px[NG];
    int ow2 = OW;
    asm volatile("" : "+s"(ow2));                        // opaque: keeps the divisions behind the k-loop
#pragma unroll
    for (int g = 0; g < NG; ++g) {
        qq[g] = (g0 + g) * 32 + n;
        py[g] = qq[g] / ow2;
        px[g] = qq[g] - py[g] * ow2;
    }
#pragma unroll
    for (int g = 0; g < NG; ++g) {
        if (qq[g] >= npx) continue;
        if (role == RSG_FM_UP_Y0 || role == RSG_FM_UP_Y1) {
            char* dst = smem + cv.t_off + (qq[g] * cv.cout + cbase) * 2;
#pragma unroll
            for (int k = 0; k < 2; ++k) {
                f32x8 f;
#pragma unroll
                for (int e = 0; e < 8; ++e) f[e] = acc[g][k * 8 + e];
                if (cbase + k * 8 < cv.cout)
                    *reinterpret_cast<uint4*>(dst + k * 16) = __builtin_bit_cast(uint4, __builtin_convertvector(f, typename DT::vec8));
            }
            continue;
        }
        const int gpix = (b * (p.H >> lvl) + band * OR + py[g]) * OW + px[g];
        if (role == RSG_FM_TO_GLOBAL) {
            char* dst = cv.out + ((size_t)gpix * cv.ldo + cbase) * 2;
#pragma unroll
            for (int k = 0; k < 2; ++k) {
                f32x8 f;
#pragma unroll
                for (int e = 0; e < 8; ++e) f[e] = acc[g][k * 8 + e];
                if (cv.relu) {                         // conv_body's epilogue: fmaxf in fp32, then one rounding
#pragma unroll
                    for (int e = 0; e < 8; ++e) f[e] = fmaxf(f[e], 0.f);
                }
                // (the padding channels of a 48-channel result -- zero weights, zero bias -- are written like rsg_conv_igemm
                // writes them: the map's buffer has round_up(cout, 32) channels)
                *reinterpret_cast<uint4*>(dst + k * 16) = __builtin_bit_cast(uint4, __builtin_convertvector(f, typename DT::vec8));
            }
            continue;
        }
        // role DOWN_Y1: the sum of output 1 (lvl == 1).  Folded form (the module's LAST output, rsg_conv_igemm's residual
        // epilogue): relu(acc + x1) in fp32; item form (fuse_body): 0 + round16(conv) + x1 + up(t21) + up(t31)
        const int h1 = p.halo[1], C1 = 2 * p.C0;
        const char* x1p = smem + p.off_x[1] + (py[g] + h1) * p.rp[1] + (h1 ? fm_slot(px[g], ((p.W >> 1) + 2) >> 1) : px[g]) * (2 * C1 + 16) + cbase * 2;
        char* dst = p.y[1] + ((size_t)gpix * p.ldy[1] + cbase) * 2;
#pragma unroll
        for (int k = 0; k < 2; ++k) {
            f32x8 f;
#pragma unroll
            for (int e = 0; e < 8; ++e) f[e] = acc[g][k * 8 + e];
            const f32x8 xf = __builtin_convertvector(__builtin_bit_cast(typename DT::vec8, *reinterpret_cast<const uint4*>(x1p + k * 16)), f32x8);
            if (p.y1_fold) {                           // conv_body's residual epilogue: +x, fmaxf, one rounding
#pragma unroll
                for (int e = 0; e < 8; ++e) f[e] = fmaxf(f[e] + xf[e], 0.f);
                *reinterpret_cast<uint4*>(dst + k * 16) = __builtin_bit_cast(uint4, __builtin_convertvector(f, typename DT::vec8));
                continue;
            } else {
                const f32x8 cf = __builtin_convertvector(__builtin_convertvector(f, typename DT::vec8), f32x8);
#pragma unroll
                for (int e = 0; e < 8; ++e) f[e] = (0.f + cf[e]) + xf[e];
#pragma unroll
                for (int j = 2; j < 4; ++j) {
                    if (j >= p.nbr) break;
                    const char* tp = smem + p.t1_off[j - 1] + (((py[g] >> (j - 1)) * (p.W >> j) + (px[g] >> (j - 1))) * C1 + cbase) * 2;
                    const f32x8 tf = __builtin_convertvector(__builtin_bit_cast(typename DT::vec8, *reinterpret_cast<const uint4*>(tp + k * 16)), f32x8);
#pragma unroll
                    for (int e = 0; e < 8; ++e) f[e] += tf[e];
                }
            }
            *reinterpret_cast<uint4*>(dst + k * 16) = __builtin_bit_cast(uint4, DT::relu_pack(f));
        }
    }
}

template <class DT>
__device__ __forceinline__ void fm_run(const FmArgs& p, const FmJob jb, char* smem, const int b, const int band, const int lane) {
    const FmConv cv = p.conv[jb.conv];      // a batch of scalar loads from the kernel-argument segment
    const int key = (cv.taps == 9 ? 1000 : 0) + cv.ks * 4 + jb.ng;
#define FM_CASE(T, NCH, KPT, N) case ((T) == 9 ? 1000 : 0) + (T) * (NCH) * (KPT) * 4 + (N): fm_job<DT, T, NCH, KPT, N>(p, cv, jb.ct, jb.g0, smem, b, band, lane); break;
#define FM_CASE3(T, NCH, KPT) FM_CASE(T, NCH, KPT, 1) FM_CASE(T, NCH, KPT, 2) FM_CASE(T, NCH, KPT, 3)
    switch (key) {
        // HRNet-W32: 1x1 over 64 / 128 / 256 channels (blocks of 64), 3x3 stride 2 over 32 / 64
        FM_CASE3(1, 1, 4)
        FM_CASE3(1, 2, 4)
        FM_CASE3(1, 4, 4)
        FM_CASE3(9, 1, 2)
        FM_CASE(9, 1, 4, 1) FM_CASE(9, 1, 4, 2)
        // HRNet-W48: 1x1 over 96 (3 blocks of 32) / 192 / 384 (blocks of 64), 3x3 stride 2 over 48 (one block) / 96 (3 x 32)
        FM_CASE3(1, 3, 2)
        FM_CASE3(1, 3, 4)
        FM_CASE3(1, 6, 4)
        FM_CASE3(9, 1, 3)
        FM_CASE(9, 3, 2, 1) FM_CASE(9, 3, 2, 2)
    }
#undef FM_CASE3
#undef FM_CASE
}

template <class DT>
__global__ void __launch_bounds__(FM_THREADS) fuse_module_kernel(const FmArgs p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    FM_STAMP(0);
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int bands = p.H / p.R0;
    const int blk = rsg_xcd_tile(blockIdx.x, gridDim.x);        // the bands of a picture (shared halo rows) on one XCD
    const int b = blk / bands, band = blk - b * bands;

    // ---- 0. weight prefetch: nothing of the module's fuse weights (0.3 - 0.6 MB) is cached when its turn comes, every
    // workgroup streams all of them, and the workgroups run in lockstep -- each fragment's first request would come from all
    // 256 CUs at once and wait for HBM behind the staging traffic (stamps: both conv phases ran at the ~27 B/clk per CU that
    // a 4-6 deep fragment ring gets at an HBM miss's latency).  The first workgroups touch one dword per 128-byte line before
    // anything else is requested (chain.hip, chain_prefetch_weights); the loads are waited for behind the staging barrier.
    unsigned pf = 0;
    if (p.pre_total) {
        // line f of the concatenated weight tensors belongs to thread f / grid of workgroup f % grid: ~20 loads per workgroup
        const int f = blockIdx.x + gridDim.x * tid;
        if (f < p.pre_total) {
            int c = 0;
#pragma unroll
            for (int k = 1; k < FM_MAX_CONVS; ++k) c += (k < p.n_conv && f >= p.pre_first[k]) ? 1 : 0;
            const char* w = nullptr;
            int base = 0;
#pragma unroll
            for (int k = 0; k < FM_MAX_CONVS; ++k)
                if (k == c) { w = reinterpret_cast<const char*>(p.conv[k].w); base = p.pre_first[k]; }
            pf = *reinterpret_cast<const unsigned*>(w + (size_t)(f - base) * 128);
        }
    }
    // ---- 1. every byte of the band in one round of requests -----------------------------------------------------------
    // A row of EVERY branch is W * C0 / 8 chunks of 16 B (C doubles as W halves).  Slot s of a thread is chunk
    // (slot_q[s] rows + slot_m[s] chunks) + (tid / rowc, tid % rowc) of branch slot_br[s]: the host's table, so the slots
    // are static registers and a thread divides once.
    {
        const int cpp0 = p.C0 >> 3;                     // 16-byte chunks per pixel of branch 0
        const int rowc = p.W * cpp0;
        const int r_t = tid / rowc, c_t = tid - r_t * rowc;
        uint4 v[FM_SLOTS];
        int lo[FM_SLOTS];
#pragma unroll
        for (int s = 0; s < FM_SLOTS; ++s) {
            const int i = p.slot_br[s];
            lo[s] = -1;
            v[s] = make_uint4(0, 0, 0, 0);
            if (i < 0) continue;                        // wave-uniform
            const int hi = p.halo[i], rows = (p.R0 >> i) + hi, Wi = p.W >> i, Hi = p.H >> i;
            const int cpp = cpp0 << i, PSi = cpp * 16 + 16, HWi = (Wi + 2) >> 1, ld2 = p.ldx[i] * 2;
            const int row0 = band * (p.R0 >> i) - hi;
            int r = r_t + p.slot_q[s], c = c_t + p.slot_m[s];
            if (c >= rowc) { c -= rowc; ++r; }
            if (r < rows) {
                const int x = c / cpp, cg = c - x * cpp;
                lo[s] = p.off_x[i] + r * p.rp[i] + (hi ? fm_slot(x, HWi) : x) * PSi + cg * 16;
                if (row0 + r >= 0)
                    v[s] = *reinterpret_cast<const uint4*>(p.x[i] + ((size_t)(b * Hi + row0 + r) * Wi + x) * ld2 + cg * 16);
            }
        }
        // the zero column of the de-interleaved images while the loads are in flight
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            if (!p.halo[i]) continue;
            const int rows = (p.R0 >> i) + 1, cpp = (p.C0 >> 3) << i;
            for (int id = tid; id < rows * cpp; id += FM_THREADS) {
                const int r = id / cpp, cg = id - r * cpp;
                *reinterpret_cast<uint4*>(smem + p.off_x[i] + r * p.rp[i] + cg * 16) = make_uint4(0, 0, 0, 0);
            }
        }
#pragma unroll
        for (int s = 0; s < FM_SLOTS; ++s)
            if (lo[s] >= 0) *reinterpret_cast<uint4*>(smem + lo[s]) = v[s];
    }
    __syncthreads();
    asm volatile("" ::"v"(pf));                          // the prefetch loads end here
    FM_STAMP(1);

    // ---- 2. this wave's jobs: phase A (1x1 convs into LDS), FM_SYNC, phase B ------------------------------------------------
    for (int j = p.first[wave]; j < p.first[wave + 1]; ++j) {
        const FmJob jb = p.job[j];
        if (jb.conv != FM_SYNC) {
            fm_run<DT>(p, jb, smem, b, band, lane);
            continue;
        }
        __syncthreads();
        FM_STAMP(2);
        // ---- y0 = relu(x0 + up(t10) + up(t20) + up(t30)): fuse_body's order and rounding, LDS -> memory ------------------
        const int W = p.W, h0 = p.halo[0], HW0 = (W + 2) >> 1, cpp0 = p.C0 >> 3, PS0 = 2 * p.C0 + 16;
        const int cppo = (p.C0 + 31) >> 5 << 2;         // chunks of the output pixel incl. the padding of a 48-in-64 buffer
        const int total = p.R0 * W * cppo;
        for (int id = tid; id < total; id += FM_THREADS) {
            const int pl = id / cppo, cg = id - pl * cppo;
            const int y = pl / W, x = pl - y * W;
            if (cg >= cpp0) {                            // channels 48..63: zeros, as the grouped path's sum of zero paddings
                *reinterpret_cast<uint4*>(p.y[0] + ((size_t)((b * p.H + band * p.R0 + y) * W + x) * p.ldy[0]) * 2 + cg * 16) = make_uint4(0, 0, 0, 0);
                continue;
            }
            float acc[8];
            {
                const typename DT::vec8 xv = __builtin_bit_cast(typename DT::vec8, *reinterpret_cast<const uint4*>(
                    smem + p.off_x[0] + (y + h0) * p.rp[0] + (h0 ? fm_slot(x, HW0) : x) * PS0 + cg * 16));
#pragma unroll
                for (int e = 0; e < 8; ++e) acc[e] = 0.f + (float)xv[e];
            }
#pragma unroll
            for (int jj = 1; jj < 4; ++jj) {
                if (jj >= p.nbr) break;
                const typename DT::vec8 tv = __builtin_bit_cast(typename DT::vec8, *reinterpret_cast<const uint4*>(
                    smem + p.t0_off[jj - 1] + ((y >> jj) * (W >> jj) + (x >> jj)) * (2 * p.C0) + cg * 16));
#pragma unroll
                for (int e = 0; e < 8; ++e) acc[e] += (float)tv[e];
            }
            f32x8 f;
#pragma unroll
            for (int e = 0; e < 8; ++e) f[e] = acc[e];
            *reinterpret_cast<uint4*>(p.y[0] + ((size_t)((b * p.H + band * p.R0 + y) * W + x) * p.ldy[0]) * 2 + cg * 16) =
                __builtin_bit_cast(uint4, DT::relu_pack(f));
        }
        FM_STAMP(3);
    }
    FM_STAMP(4);
}

// ---------------------------------------------------------------------------------------------------------------------------
// host
// ---------------------------------------------------------------------------------------------------------------------------
extern "C" size_t rsg_fuse_module_table_bytes(void) { return sizeof(FmArgs); }
int g_fm_prefetch = 1;                 // rsg_tune("fm_prefetch", 0|1)

static int fm_row_pitch(int slots, int ps, int out_w) {
    // row pitch == (out_w * ps / 2) (mod 128 B): an output row of a stride-2 conv ends out_w slots on and the next one starts
    // two rows down -- a 32-pixel group that wraps keeps the odd-slot bank progression (conv_igemm.hip, tile_geometry)
    const int base = slots * ps, want = (out_w * ps / 2) % 128;
    return base + ((want - base) % 128 + 128) % 128;
}

extern "C" int rsg_fuse_module_plan(const rsg_fuse_module_desc* d, void* table_host, int32_t* launch2) {
    if (!d || !table_host || !launch2) RSG_FAIL("fuse_module: null pointer");
    const int nbr = d->n_branches;
    if (nbr < 2 || nbr > 4 || d->batch < 1 || d->n_convs < 2 || d->n_convs > FM_MAX_CONVS) RSG_FAIL("fuse_module: %d branches, %d convs", nbr, d->n_convs);
    const int C0 = d->c0 ? d->c0 : 32;
    if (C0 != 32 && C0 != 48) RSG_FAIL("fuse_module: %d channels in branch 0 (32: HRNet-W32, 48: HRNet-W48)", C0);
    if (d->w < 8 || (d->w & 7) || (d->h & 7) || d->w * (C0 / 8) > FM_THREADS) RSG_FAIL("fuse_module: %dx%d map (multiples of 8, at most %d channels per row)", d->h, d->w, FM_THREADS * 8);
    for (int i = 0; i < nbr; ++i)
        if (!d->x[i] || ((uintptr_t)d->x[i] & 15) || d->x_ld[i] < (C0 << i) || (d->x_ld[i] & 7)) RSG_FAIL("fuse_module: branch %d operand", i);
    for (int i = 0; i < 2; ++i)
        if (!d->y[i] || ((uintptr_t)d->y[i] & 15) || d->y_ld[i] < round_up(C0 << i, 32) || (d->y_ld[i] & 7)) RSG_FAIL("fuse_module: output %d", i);
    if ((long long)d->batch * d->h * d->w * std::max(d->y_ld[0], d->x_ld[0]) * 2 >= (1LL << 31)) RSG_FAIL("fuse_module: map too large");
    FmArgs& a = *(FmArgs*)table_host;
    memset(&a, 0, sizeof(a));
    for (int i = 0; i < nbr; ++i) { a.x[i] = (const char*)d->x[i]; a.ldx[i] = d->x_ld[i]; }
    for (int i = 0; i < 2; ++i) { a.y[i] = (char*)d->y[i]; a.ldy[i] = d->y_ld[i]; }
    a.B = d->batch; a.H = d->h; a.W = d->w; a.nbr = nbr; a.y1_fold = d->y1_fold ? 1 : 0; a.n_conv = d->n_convs; a.C0 = C0;
    a.conv_y1 = -1;
    int up0[3] = {-1, -1, -1}, up1[3] = {-1, -1, -1};
    for (int c = 0; c < d->n_convs; ++c) {
        const rsg_fm_conv& s = d->convs[c];
        FmConv& o = a.conv[c];
        if (!s.w_packed || !s.bias || (((uintptr_t)s.w_packed | (uintptr_t)s.bias) & 15)) RSG_FAIL("fuse_module: conv %d weights", c);
        if (s.src < 0 || s.src >= nbr || (s.k != 1 && s.k != 3) || s.cout < C0 || s.cout > 8 * C0 || (s.cout % C0)) RSG_FAIL("fuse_module: conv %d shape", c);
        o.w = (const uint4*)s.w_packed; o.bias = s.bias; o.out = (char*)s.out16; o.ldo = s.out_ld;
        o.src = s.src; o.taps = s.k == 3 ? 9 : 1; o.cout = s.cout; o.relu = s.relu ? 1 : 0; o.role = s.role;
        o.lvl = s.k == 3 ? s.src + 1 : s.src;
        o.ks = o.taps * ((C0 << s.src) / 16);
        o.cout_pad = round_up(s.cout, 32);
        if (o.lvl >= nbr) RSG_FAIL("fuse_module: conv %d leaves the module", c);
        if (s.k == 3) {
            if (s.src > 1) RSG_FAIL("fuse_module: conv %d: strided convs of branches 0 and 1 only", c);
            a.halo[s.src] = 1;
        }
        switch (s.role) {
        case RSG_FM_TO_GLOBAL:
            if (!s.out16 || ((uintptr_t)s.out16 & 15) || s.out_ld < round_up(s.cout, 32) || (s.out_ld & 7)) RSG_FAIL("fuse_module: conv %d output", c);
            break;
        case RSG_FM_UP_Y0:
            if (s.k != 1 || s.src < 1 || s.cout != C0 || s.relu || up0[s.src - 1] >= 0) RSG_FAIL("fuse_module: conv %d as an addend of y0", c);
            up0[s.src - 1] = c;
            break;
        case RSG_FM_UP_Y1:
            if (s.k != 1 || s.src < 2 || s.cout != 2 * C0 || s.relu || up1[s.src - 1] >= 0) RSG_FAIL("fuse_module: conv %d as an addend of y1", c);
            up1[s.src - 1] = c;
            break;
        case RSG_FM_DOWN_Y1:
            if (s.k != 3 || s.src != 0 || s.cout != 2 * C0 || s.relu || a.conv_y1 >= 0) RSG_FAIL("fuse_module: conv %d as the 0 -> 1 conv", c);
            a.conv_y1 = c;
            break;
        default: RSG_FAIL("fuse_module: conv %d role %d", c, s.role);
        }
    }
    if (a.conv_y1 < 0) RSG_FAIL("fuse_module: no 0 -> 1 conv");
    for (int j = 1; j < nbr; ++j)
        if (up0[j - 1] < 0 || (j >= 2 && !a.y1_fold && up1[j - 1] < 0)) RSG_FAIL("fuse_module: missing 1x1 conv of branch %d", j);
    if (a.y1_fold && nbr != 2) RSG_FAIL("fuse_module: the folded y1 sum is the two-branch module's");
    a.halo[0] = 1;
    // band height: the largest of 16 / 8 rows whose images fit
    int lds = 0;
    for (int R0 = 16; R0 >= 8; R0 >>= 1) {
        if ((d->h % R0) || (R0 >> (nbr - 1)) < 1) continue;
        int off = 0, rounds = 0;
        const int rowc = d->w * (C0 / 8);
        for (int s = 0; s < FM_SLOTS; ++s) a.slot_br[s] = -1;
        for (int i = 0; i < nbr; ++i) {
            const int rows = (R0 >> i) + a.halo[i], Wi = d->w >> i, ps = ((2 * C0) << i) + 16;
            a.off_x[i] = off;
            a.rp[i] = a.halo[i] ? fm_row_pitch(Wi + 1, ps, Wi / 2) : Wi * ps;
            off += rows * a.rp[i];
            for (int e = 0; e < rows * rowc; e += FM_THREADS, ++rounds)
                if (rounds < FM_SLOTS) { a.slot_br[rounds] = i; a.slot_q[rounds] = e / rowc; a.slot_m[rounds] = e % rowc; }
        }
        for (int c = 0; c < d->n_convs; ++c) {
            FmConv& o = a.conv[c];
            if (o.role != RSG_FM_UP_Y0 && o.role != RSG_FM_UP_Y1) continue;
            o.t_off = off;
            (o.role == RSG_FM_UP_Y0 ? a.t0_off : a.t1_off)[o.src - 1] = off;
            off += (R0 >> o.lvl) * (d->w >> o.lvl) * o.cout * 2;
        }
        if (off <= 160 * 1024 && rounds <= FM_SLOTS) { a.R0 = R0; lds = off; break; }
    }
    if (!a.R0) RSG_FAIL("fuse_module: the band of a %dx%d map with %d branches does not fit LDS", d->h, d->w, nbr);
    // jobs: (conv, output tile, <= 3 pixel groups); phase A = the LDS-resident 1x1 results, phase B = the rest
    struct J { FmJob j; int cost; };
    std::vector<J> ja, jb;
    for (int c = 0; c < d->n_convs; ++c) {
        const FmConv& o = a.conv[c];
        const int npx = (a.R0 >> o.lvl) * (a.W >> o.lvl), groups = (npx + 31) / 32;
        const int ngmax = o.ks >= 36 ? 2 : 3;           // (registers: 16 accumulators per group beside the fragment rings)
        if (groups > 255) RSG_FAIL("fuse_module: band too large");
        std::vector<J>& dst = (o.role == RSG_FM_UP_Y0 || o.role == RSG_FM_UP_Y1) ? ja : jb;
        for (int ct = 0; ct < o.cout_pad / 32; ++ct) {
            const int parts = (groups + ngmax - 1) / ngmax;
            for (int q = 0, g = 0; q < parts; ++q) {
                const int ng = (groups - g + (parts - q) - 1) / (parts - q);
                dst.push_back({{(unsigned char)c, (unsigned char)ct, (unsigned char)g, (unsigned char)ng}, ng * o.ks});
                g += ng;
            }
        }
    }
    // fewer jobs than waves: halve the longest multi-group job until every wave has one or nothing splits
    auto split = [&](std::vector<J>& v) {
        while ((int)v.size() < FM_WAVES) {
            int best = -1;
            for (int i = 0; i < (int)v.size(); ++i)
                if (v[i].j.ng > 1 && (best < 0 || v[i].cost > v[best].cost)) best = i;
            if (best < 0) break;
            J x = v[best];
            const int n1 = x.j.ng / 2, per = x.cost / x.j.ng;
            v[best].j.ng = (unsigned char)n1; v[best].cost = per * n1;
            J y = x;
            y.j.g0 = (unsigned char)(x.j.g0 + n1); y.j.ng = (unsigned char)(x.j.ng - n1); y.cost = per * y.j.ng;
            v.push_back(y);
        }
    };
    split(ja);
    split(jb);
    // longest first onto the least-loaded wave; ties go to the wave whose SIMD (wave mod 4) carries least
    auto deal = [&](std::vector<J>& v, std::vector<std::vector<FmJob>>& per) {
        std::stable_sort(v.begin(), v.end(), [](const J& l, const J& r) { return l.cost > r.cost; });
        int load[FM_WAVES] = {0};
        for (const J& x : v) {
            int simd[4] = {0, 0, 0, 0};
            for (int w = 0; w < FM_WAVES; ++w) simd[w & 3] += load[w];
            int bw = 0;
            for (int w = 1; w < FM_WAVES; ++w)
                if (load[w] < load[bw] || (load[w] == load[bw] && simd[w & 3] < simd[bw & 3])) bw = w;
            per[bw].push_back(x.j);
            load[bw] += x.cost;
        }
    };
    if ((int)(ja.size() + jb.size()) + FM_WAVES > 2 * FM_MAX_JOBS) RSG_FAIL("fuse_module: %d jobs", (int)(ja.size() + jb.size()));
    std::vector<std::vector<FmJob>> pa(FM_WAVES), pb(FM_WAVES);
    deal(ja, pa);
    deal(jb, pb);
    {
        int k = 0;
        for (int w = 0; w < FM_WAVES; ++w) {
            a.first[w] = k;
            for (const FmJob& j : pa[w]) a.job[k++] = j;
            a.job[k++] = FmJob{FM_SYNC, 0, 0, 0};
            for (const FmJob& j : pb[w]) a.job[k++] = j;
        }
        a.first[FM_WAVES] = k;
    }
    if (g_fm_prefetch) {
        int n = 0;
        for (int c = 0; c < d->n_convs; ++c) { a.pre_first[c] = n; n += (a.conv[c].cout_pad >> 5) * a.conv[c].ks * 8; }
        a.pre_total = n;
    }
    launch2[0] = d->batch * (d->h / a.R0);
    launch2[1] = lds;
    return 0;
}

// tests / logs: what _plan decided.  info4 = {band rows of branch 0, jobs, LDS bytes of the images, 0}; first17 / jobs4
// (96 x {conv, tile, first group, groups}; conv 255 = the sync job): the waves' job lists
extern "C" int rsg_fuse_module_table_jobs(const void* table_host, int32_t* info4, int32_t* first17, uint8_t* jobs4) {
    if (!table_host || !info4 || !first17 || !jobs4) RSG_FAIL("fuse_module_table_jobs: null pointer");
    const FmArgs& a = *(const FmArgs*)table_host;
    info4[0] = a.R0; info4[1] = a.first[FM_WAVES]; info4[2] = a.off_x[a.nbr - 1]; info4[3] = 0;
    for (int w = 0; w <= FM_WAVES; ++w) first17[w] = a.first[w];
    memcpy(jobs4, a.job, sizeof(a.job));
    return 0;
}

extern "C" int rsg_fuse_module_run(const void* table_host, int32_t blocks, int32_t lds, int32_t dtype, void* stream) {
    if (!table_host || blocks < 1 || lds < 1 || lds > 160 * 1024) RSG_FAIL("fuse_module_run: bad arguments");
    static_assert(sizeof(FmArgs) <= 4096, "the table travels as the kernel argument");
    const FmArgs& a = *(const FmArgs*)table_host;
    if (a.R0 < 8 || a.B < 1) RSG_FAIL("fuse_module_run: the table is not one rsg_fuse_module_plan completed");
    if (blocks != a.B * (a.H / a.R0)) RSG_FAIL("fuse_module_run: %d blocks for a table of %d", blocks, a.B * (a.H / a.R0));
    const void* fn = dtype == RSG_BF16 ? (const void*)fuse_module_kernel<BF16> : (const void*)fuse_module_kernel<F16>;
    static int lds_set[RSG_MAX_DEVICES][2];
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= RSG_MAX_DEVICES) RSG_FAIL("fuse_module: no current device");
    if (lds_set[dev][dtype == RSG_BF16] < lds) {
        if (hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, lds) != hipSuccess) RSG_FAIL("fuse_module: cannot reserve %d B of LDS", lds);
        lds_set[dev][dtype == RSG_BF16] = lds;
    }
    if (dtype == RSG_BF16) RSG_LAUNCH(fuse_module_kernel<BF16>, dim3(blocks), dim3(FM_THREADS), lds, (hipStream_t)stream, a);
    else RSG_LAUNCH(fuse_module_kernel<F16>, dim3(blocks), dim3(FM_THREADS), lds, (hipStream_t)stream, a);
    RSG_CHECK_LAUNCH("fuse_module");
    return 0;
}
