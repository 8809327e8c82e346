// Weight gradient of a convolution on the matrix cores (gfx950).
//
// Replaces autograd's conv weight gradient for the nn.Conv2d layers of pose_hrnet.py (3x3 s1/s2 p1, 1x1) in the
// training step (lib/core/function.py:27-80):
//     dW[o][i][ky][kx] = sum over (b, r, c) of  dy[b, r, c, o] * x[b, r*s + ky - p, c*s + kx - p, i]
// i.e. per tap a GEMM  D[o][i] = A[o][pixel] * B[pixel][i]  whose contraction index is the PIXEL.  Both operands are
// NHWC (channel-fast), so a lane's MFMA fragment (8 consecutive k for one row/column) is strided in memory; the tiles
// are staged in LDS as [pixel][32 channels] (64 dense bytes per pixel) and read with ds_read_b64_tr_b16, the hardware
// transpose read: 4 pixels x 16 channels per 16-lane group, every lane supplying the address of its own pixel row, so
// the shifted (and for stride 2: strided) window of a tap is just a per-lane address.
//
// Decomposition: one workgroup (4 waves) owns ONE 32x32 (cout tile, cin tile) pair for ALL taps and a slice of the
// pixels (chunks = row bands of single images, dealt round-robin to the slices); inside a chunk the waves take
// alternate 16-pixel k-steps, each holding taps x 16 accumulator registers.  The four waves' tiles are added in LDS
// in a fixed order, the slices' partial tiles go to a workspace and a second kernel sums them in slice order into
// dW (fp32, nn.Conv2d layout): no atomics, bit-reproducible.
#include "common.h"
#include <algorithm>

#define NLD 8            // staging loads in flight per thread
typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef short s16x8 __attribute__((ext_vector_type(8)));

struct WgradArgs {
    const char* x;     // [B, Hin, Win, CinP] 16-bit
    const char* dy;    // [B, Hout, Wout, CoutP] 16-bit
    float* part;       // [pairs][nslices][taps][32][32] fp32
    int B, Hin, Win, CinP, Hout, Wout, CoutP;
    int kh, kw, stride, pad;
    int n_ot, n_it, nslices;
    int RB, nbands;    // output rows per chunk, chunks per image
    int nimg;          // > 1: a chunk is `nimg` whole images (small maps: 8x6 is 3 k-steps, too little between two barriers)
    int xrows, xcols;  // x tile: (RB-1)*stride + kh rows, Win + 2*pad columns
    int P16;           // pixels of a chunk, padded to a multiple of 16
    int quad;          // 1: a workgroup owns a 2x2 block of tile pairs, one per wave (wgrad_quad_body)
};

template <class DT>
__device__ __forceinline__ f32x16 wg_mfma(s16x8 a, s16x8 b, f32x16 c) {
    return DT::mfma(__builtin_bit_cast(typename DT::vec8, a), __builtin_bit_cast(typename DT::vec8, b), c);
}

__device__ __forceinline__ s16x4 tr_read(const char* lds, int off) {
    return __builtin_amdgcn_ds_read_tr16_b64_v4i16(
        (__attribute__((address_space(3))) s16x4*)(__attribute__((address_space(3))) char*)(lds + off));
}

template <class DT, int NT>
__device__ __forceinline__ void wgrad_body(const WgradArgs& a, const int block, char* smem) {
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int pair = block / a.nslices, slice = block - pair * a.nslices;
    const int ot = pair / a.n_it, it = pair - ot * a.n_it;
    char* const sdy = smem;                                   // [P16][64 B]
    char* const sx = smem + a.P16 * 64;                       // [xrows * xcols][64 B]
    int* const soff = reinterpret_cast<int*>(sx + a.nimg * a.xrows * a.xcols * 64);   // [P16]: byte offset of a pixel's tap (0,0) in sx

    // fragment geometry of this lane (see the header): 16-lane group g reads channels cb..cb+15, k rows kb..kb+7
    const int grp = lane >> 4, idx = lane & 15, q = idx >> 2, pp = idx & 3;
    const int cb = 16 * (grp & 1), kb = 8 * (grp >> 1);
    const int lane_col = (cb + 4 * pp) * 2;                   // byte offset of this lane's 4 channels inside a pixel

    f32x16 acc[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;

    const int nchunks = a.nimg > 1 ? (a.B + a.nimg - 1) / a.nimg : a.B * a.nbands;
    const int xtile = a.xrows * a.xcols, opix = a.Hout * a.Wout;
    for (int chunk = slice; chunk < nchunks; chunk += a.nslices) {
        const int b = a.nimg > 1 ? chunk * a.nimg : chunk / a.nbands, band = a.nimg > 1 ? 0 : chunk - b * a.nbands;
        const int nim = a.nimg > 1 ? min(a.nimg, a.B - b) : 1;
        const int r0 = band * a.RB;                           // first output row of the chunk
        const int rows = min(a.RB, a.Hout - r0);
        const int P = nim * rows * a.Wout;                    // valid pixels (whole images are contiguous in memory)
        __syncthreads();                                      // the previous chunk's reads are done
        // ---- dy tile: the band is one contiguous pixel range of the image
        {
            // NLD loads in flight per thread, then the LDS writes (one load per iteration waited for its own latency)
            const char* src = a.dy + ((size_t)(b * a.Hout + r0) * a.Wout) * a.CoutP * 2 + ot * 64;
            for (int base = 0; base < a.P16 * 4; base += 256 * NLD) {
                uint4 v[NLD];
#pragma unroll
                for (int j = 0; j < NLD; ++j) {
                    const int i = base + j * 256 + tid, p = i >> 2, c4 = i & 3;
                    v[j] = make_uint4(0, 0, 0, 0);
                    if (i < a.P16 * 4 && p < P && ot * 32 + c4 * 8 < a.CoutP) v[j] = *reinterpret_cast<const uint4*>(src + (size_t)p * a.CoutP * 2 + c4 * 16);
                }
#pragma unroll
                for (int j = 0; j < NLD; ++j) {
                    const int i = base + j * 256 + tid;
                    if (i < a.P16 * 4) *reinterpret_cast<uint4*>(sdy + i * 16) = v[j];
                }
            }
        }
        // ---- x tile with its zero border
        {
            const int iy0 = r0 * a.stride - a.pad;
            const int npix = nim * xtile;
            for (int base = 0; base < npix * 4; base += 256 * NLD) {
                uint4 v[NLD];
#pragma unroll
                for (int j = 0; j < NLD; ++j) {
                    const int i = base + j * 256 + tid, p = i >> 2, c4 = i & 3;
                    const int im = p / xtile, pr = p - im * xtile;
                    const int ry = pr / a.xcols, cx = pr - ry * a.xcols;
                    const int iy = iy0 + ry, ix = cx - a.pad;
                    v[j] = make_uint4(0, 0, 0, 0);
                    if (i < npix * 4 && iy >= 0 && iy < a.Hin && ix >= 0 && ix < a.Win && it * 32 + c4 * 8 < a.CinP)
                        v[j] = *reinterpret_cast<const uint4*>(a.x + ((size_t)((b + im) * a.Hin + iy) * a.Win + ix) * a.CinP * 2 + it * 64 + c4 * 16);
                }
#pragma unroll
                for (int j = 0; j < NLD; ++j) {
                    const int i = base + j * 256 + tid;
                    if (i < npix * 4) *reinterpret_cast<uint4*>(sx + i * 16) = v[j];
                }
            }
        }
        for (int p = tid; p < a.P16; p += 256) {
            int o = 0;
            if (p < P) {
                const int im = a.nimg > 1 ? p / opix : 0, pr = p - im * opix;
                const int r = pr / a.Wout, c = pr - r * a.Wout;
                o = (im * xtile + (r * a.stride) * a.xcols + c * a.stride) * 64;
            }
            soff[p] = o;                                      // padded pixels: dy is zero there, any finite x will do
        }
        __syncthreads();
        const int ksteps = (P + 15) >> 4;
        for (int ks = wave; ks < ksteps; ks += 4) {
            const int pix0 = ks * 16 + kb + q, pix1 = pix0 + 4;
            const s16x4 a0 = tr_read(sdy, pix0 * 64 + lane_col), a1 = tr_read(sdy, pix1 * 64 + lane_col);
            const s16x8 af = __builtin_shufflevector(a0, a1, 0, 1, 2, 3, 4, 5, 6, 7);
            const int x0 = soff[pix0] + lane_col, x1 = soff[pix1] + lane_col;
#pragma unroll
            for (int t = 0; t < NT; ++t) {
                constexpr int KW = NT == 9 ? 3 : 1;
                const int toff = ((t / KW) * a.xcols + (t % KW)) * 64;
                const s16x4 b0 = tr_read(sx, x0 + toff), b1 = tr_read(sx, x1 + toff);
                const s16x8 bf = __builtin_shufflevector(b0, b1, 0, 1, 2, 3, 4, 5, 6, 7);
                acc[t] = wg_mfma<DT>(af, bf, acc[t]);
            }
        }
    }
    // ---- add the four waves' tiles in a fixed order: (0 += 2, 1 += 3), then 0 += 1
    __syncthreads();
    float* red = reinterpret_cast<float*>(smem);               // [2][NT][16][64]
    if (wave >= 2) {
#pragma unroll
        for (int t = 0; t < NT; ++t)
#pragma unroll
            for (int r = 0; r < 16; ++r) red[(((wave - 2) * NT + t) * 16 + r) * 64 + lane] = acc[t][r];
    }
    __syncthreads();
    if (wave < 2) {
#pragma unroll
        for (int t = 0; t < NT; ++t)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[t][r] += red[((wave * NT + t) * 16 + r) * 64 + lane];
    }
    __syncthreads();
    if (wave == 1) {
#pragma unroll
        for (int t = 0; t < NT; ++t)
#pragma unroll
            for (int r = 0; r < 16; ++r) red[(t * 16 + r) * 64 + lane] = acc[t][r];
    }
    __syncthreads();
    if (wave == 0) {
        float* dst = a.part + ((size_t)pair * a.nslices + slice) * NT * 1024;
        const int col = lane & 31, hh = lane >> 5;
#pragma unroll
        for (int t = 0; t < NT; ++t)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = (r & 3) + 8 * (r >> 2) + 4 * hh;             // C/D map of the 32x32 MFMA: row = cout, col = cin
                dst[t * 1024 + row * 32 + col] = acc[t][r] + red[(t * 16 + r) * 64 + lane];
            }
    }
}

// Maps with at least 64 channels on both sides: a workgroup owns a 2x2 block of (cout tile, cin tile) pairs, ONE PAIR PER
// WAVE.  The dy and x tiles are staged 64 channels wide once for all four pairs (the one-pair form stages the same 32-channel
// slices in two workgroups each), every wave walks ALL k-steps of the chunk for its own pair, so there is no cross-wave add
// through LDS and twice the MFMA work per staged byte.  Pixel pitch 144 B (64 channels + 16): the four rows of a transposed
// read then start 36 banks apart.
// NW = 8: two waves per tile pair (two per SIMD), taking alternate k-steps; their tiles are added through LDS in a fixed order
// (wave w += wave w + 4) before the partial tile is written.  Twice the loads in flight while staging, and a second wave to
// run MFMAs while the first waits for its transposed LDS reads.
template <class DT, int NT, int NW = 4>
__device__ __forceinline__ void wgrad_quad_body(const WgradArgs& a, const int block, char* smem) {
    constexpr int PSQ = 144, NTHR = 64 * NW, KPAR = NW / 4;
    constexpr int NLQ = NW == 8 ? 4 : NLD;                  // staging loads in flight per thread (512 threads: the same bytes in flight)
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int pw = wave & 3, kpar = wave >> 2;                // tile pair of this wave, its k-step parity
    const int nq_i = (a.n_it + 1) >> 1;
    const int quad = block / a.nslices, slice = block - quad * a.nslices;
    const int oq = quad / nq_i, iq = quad - oq * nq_i;
    const int ot = oq * 2 + (pw >> 1), it = iq * 2 + (pw & 1);
    const bool active = ot < a.n_ot && it < a.n_it;          // wave-uniform
    char* const sdy = smem;                                   // [P16][144 B]
    char* const sx = smem + a.P16 * PSQ;                      // [xrows * xcols][144 B]
    int* const soff = reinterpret_cast<int*>(sx + a.nimg * a.xrows * a.xcols * PSQ);
    const int grp = lane >> 4, idx = lane & 15, q = idx >> 2, pp = idx & 3;
    const int cb = 16 * (grp & 1), kb = 8 * (grp >> 1);
    const int col_a = (cb + 4 * pp) * 2 + (pw >> 1) * 64, col_b = (cb + 4 * pp) * 2 + (pw & 1) * 64;

    f32x16 acc[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;

    const int nchunks = a.nimg > 1 ? (a.B + a.nimg - 1) / a.nimg : a.B * a.nbands;
    const int xtile = a.xrows * a.xcols, opix = a.Hout * a.Wout;
    for (int chunk = slice; chunk < nchunks; chunk += a.nslices) {
        const int b = a.nimg > 1 ? chunk * a.nimg : chunk / a.nbands, band = a.nimg > 1 ? 0 : chunk - b * a.nbands;
        const int nim = a.nimg > 1 ? min(a.nimg, a.B - b) : 1;
        const int r0 = band * a.RB;
        const int rows = min(a.RB, a.Hout - r0);
        const int P = nim * rows * a.Wout;
        __syncthreads();
        {
            const char* src = a.dy + ((size_t)(b * a.Hout + r0) * a.Wout) * a.CoutP * 2 + oq * 128;
            for (int base = 0; base < a.P16 * 8; base += NTHR * NLQ) {
                uint4 v[NLQ];
#pragma unroll
                for (int j = 0; j < NLQ; ++j) {
                    const int i = base + j * NTHR + tid, p = i >> 3, c8 = i & 7;
                    v[j] = make_uint4(0, 0, 0, 0);
                    if (i < a.P16 * 8 && p < P && oq * 64 + c8 * 8 < a.CoutP) v[j] = *reinterpret_cast<const uint4*>(src + (size_t)p * a.CoutP * 2 + c8 * 16);
                }
#pragma unroll
                for (int j = 0; j < NLQ; ++j) {
                    const int i = base + j * NTHR + tid, p = i >> 3, c8 = i & 7;
                    if (i < a.P16 * 8) *reinterpret_cast<uint4*>(sdy + p * PSQ + c8 * 16) = v[j];
                }
            }
        }
        {
            const int iy0 = r0 * a.stride - a.pad;
            const int npix = nim * xtile;
            for (int base = 0; base < npix * 8; base += NTHR * NLQ) {
                uint4 v[NLQ];
#pragma unroll
                for (int j = 0; j < NLQ; ++j) {
                    const int i = base + j * NTHR + tid, p = i >> 3, c8 = i & 7;
                    const int im = p / xtile, pr = p - im * xtile;
                    const int ry = pr / a.xcols, cx = pr - ry * a.xcols;
                    const int iy = iy0 + ry, ix = cx - a.pad;
                    v[j] = make_uint4(0, 0, 0, 0);
                    if (i < npix * 8 && iy >= 0 && iy < a.Hin && ix >= 0 && ix < a.Win && iq * 64 + c8 * 8 < a.CinP)
                        v[j] = *reinterpret_cast<const uint4*>(a.x + ((size_t)((b + im) * a.Hin + iy) * a.Win + ix) * a.CinP * 2 + iq * 128 + c8 * 16);
                }
#pragma unroll
                for (int j = 0; j < NLQ; ++j) {
                    const int i = base + j * NTHR + tid, p = i >> 3, c8 = i & 7;
                    if (i < npix * 8) *reinterpret_cast<uint4*>(sx + p * PSQ + c8 * 16) = v[j];
                }
            }
        }
        for (int p = tid; p < a.P16; p += NTHR) {
            int o = 0;
            if (p < P) {
                const int im = a.nimg > 1 ? p / opix : 0, pr = p - im * opix;
                const int r = pr / a.Wout, c = pr - r * a.Wout;
                o = (im * xtile + (r * a.stride) * a.xcols + c * a.stride) * PSQ;
            }
            soff[p] = o;
        }
        __syncthreads();
        const int ksteps = (P + 15) >> 4;
        if (active) {
            for (int ks = kpar; ks < ksteps; ks += KPAR) {
                const int pix0 = ks * 16 + kb + q, pix1 = pix0 + 4;
                const s16x4 a0 = tr_read(sdy, pix0 * PSQ + col_a), a1 = tr_read(sdy, pix1 * PSQ + col_a);
                const s16x8 af = __builtin_shufflevector(a0, a1, 0, 1, 2, 3, 4, 5, 6, 7);
                const int x0 = soff[pix0] + col_b, x1 = soff[pix1] + col_b;
#pragma unroll
                for (int t = 0; t < NT; ++t) {
                    constexpr int KW = NT == 9 ? 3 : 1;
                    const int toff = ((t / KW) * a.xcols + (t % KW)) * PSQ;
                    const s16x4 b0 = tr_read(sx, x0 + toff), b1 = tr_read(sx, x1 + toff);
                    const s16x8 bf = __builtin_shufflevector(b0, b1, 0, 1, 2, 3, 4, 5, 6, 7);
                    acc[t] = wg_mfma<DT>(af, bf, acc[t]);
                }
            }
        }
    }
    if constexpr (NW == 8) {
        // wave w += wave w + 4, a few taps at a time (4 waves x TB taps x 4 KB of LDS)
        constexpr int TB = NT >= 5 ? 5 : NT;
        float* red = reinterpret_cast<float*>(smem);
#pragma unroll
        for (int t0 = 0; t0 < NT; t0 += TB) {
            __syncthreads();                                  // staging tiles (first round) / the previous round are done with
            if (kpar == 1) {
#pragma unroll
                for (int t = 0; t < TB; ++t)
                    if (t0 + t < NT)
#pragma unroll
                        for (int r = 0; r < 16; ++r) red[((pw * TB + t) * 16 + r) * 64 + lane] = acc[t0 + t < NT ? t0 + t : 0][r];
            }
            __syncthreads();
            if (kpar == 0) {
#pragma unroll
                for (int t = 0; t < TB; ++t)
                    if (t0 + t < NT)
#pragma unroll
                        for (int r = 0; r < 16; ++r) acc[t0 + t < NT ? t0 + t : 0][r] += red[((pw * TB + t) * 16 + r) * 64 + lane];
            }
        }
        if (kpar == 1) return;
    }
    if (!active) return;
    float* dst = a.part + ((size_t)(ot * a.n_it + it) * a.nslices + slice) * NT * 1024;
    const int col = lane & 31, hh = lane >> 5;
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) dst[t * 1024 + ((r & 3) + 8 * (r >> 2) + 4 * hh) * 32 + col] = acc[t][r];
}

template <class DT, int NT, int NW = 4>
__global__ void __launch_bounds__(64 * NW) wgrad_kernel(const WgradArgs a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    if (a.quad) wgrad_quad_body<DT, NT, NW>(a, blockIdx.x, smem);
    else {
        if (NW == 8 && threadIdx.x >= 256) return;       // the one-pair form runs four waves
        wgrad_body<DT, NT>(a, blockIdx.x, smem);
    }
}

// Up to 4 independent weight gradients (the same conv position of a HighResolutionModule's branches) in one launch
#define RSG_WGRAD_GROUP_MAX 4
struct WgradGroupArgs {
    int n;
    int first[RSG_WGRAD_GROUP_MAX + 1];
    WgradArgs a[RSG_WGRAD_GROUP_MAX];
};
// dW[o][i][t] = sum over slices of part[(ot, it)][slice][t][o % 32][i % 32], in a fixed order: a workgroup owns a
// quarter (256 elements) of one (pair, tap) tile; lane l of slice group sg (4 groups) adds the float4 l of slices
// sg, sg+4, ... and the four group sums are added in group order through LDS.  Every read is a coalesced 1 KB row.
// (The first version gave each thread one output element and walked the slices with a 36 KB stride: uncoalesced, a
// quarter of the training step.)
__device__ __forceinline__ void wgrad_reduce_body(const float* __restrict__ part, float* __restrict__ dw, int cout, int cin,
                                                  int ntaps, int n_it, int nslices, const int block) {
    __shared__ float4 red[3][64];
    const int tile = block >> 2, quarter = block & 3;                    // tile = pair * ntaps + t
    const int pair = tile / ntaps, t = tile - pair * ntaps;
    const int l = threadIdx.x & 63, sg = threadIdx.x >> 6;
    const float4* p = reinterpret_cast<const float4*>(part + ((size_t)pair * nslices * ntaps + t) * 1024) + quarter * 64 + l;
    const size_t stride = (size_t)ntaps * 256;                            // float4s between two slices
    float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int k = sg; k < nslices; k += 4) {
        const float4 v = p[(size_t)k * stride];
        s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w;
    }
    if (sg) red[sg - 1][l] = s;
    __syncthreads();
    if (sg) return;
    for (int g = 0; g < 3; ++g) { const float4 v = red[g][l]; s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w; }
    const int ot = pair / n_it, it = pair - ot * n_it;
    const int e0 = quarter * 256 + l * 4;                                 // element of the 32x32 tile: row = cout, col = cin
    const int o = ot * 32 + (e0 >> 5), i0 = it * 32 + (e0 & 31);
    if (o >= cout) return;
    const float v[4] = {s.x, s.y, s.z, s.w};
#pragma unroll
    for (int j = 0; j < 4; ++j)
        if (i0 + j < cin) dw[((size_t)o * cin + i0 + j) * ntaps + t] = v[j];
}

__global__ void __launch_bounds__(256) wgrad_reduce_kernel(const float* __restrict__ part, float* __restrict__ dw, int cout, int cin,
                                                           int ntaps, int n_it, int nslices) {
    wgrad_reduce_body(part, dw, cout, cin, ntaps, n_it, nslices, blockIdx.x);
}
struct WgradReduceGroupArgs {
    int n, ntaps;
    int first[RSG_WGRAD_GROUP_MAX + 1];
    const float* part[RSG_WGRAD_GROUP_MAX];
    float* dw[RSG_WGRAD_GROUP_MAX];
    int cout[RSG_WGRAD_GROUP_MAX], cin[RSG_WGRAD_GROUP_MAX], n_it[RSG_WGRAD_GROUP_MAX], nslices[RSG_WGRAD_GROUP_MAX];
};
__device__ __forceinline__ void wgrad_reduce_group_body(const WgradReduceGroupArgs& g, const int blk) {
    int idx = 0;
#pragma unroll
    for (int k = 1; k < RSG_WGRAD_GROUP_MAX; ++k) idx += (k < g.n && blk >= g.first[k]) ? 1 : 0;
    const float* part = nullptr; float* dw = nullptr;
    int cout = 0, cin = 0, n_it = 0, nslices = 0, block = 0;
#pragma unroll
    for (int k = 0; k < RSG_WGRAD_GROUP_MAX; ++k)
        if (k == idx) { part = g.part[k]; dw = g.dw[k]; cout = g.cout[k]; cin = g.cin[k]; n_it = g.n_it[k]; nslices = g.nslices[k]; block = blk - g.first[k]; }
    wgrad_reduce_body(part, dw, cout, cin, g.ntaps, n_it, nslices, block);
}
__global__ void __launch_bounds__(256) wgrad_reduce_group_kernel(const WgradReduceGroupArgs g) {
    wgrad_reduce_group_body(g, (int)blockIdx.x);
}

// The main kernel of one group may carry the REDUCTION of the previous group (pipelined form, rsg_conv_wgrad_group_ex): the
// first r.first[r.n] workgroups sum the previous launch's partial tiles (bandwidth work, no LDS) while the others stage
// and multiply (latency-bound, one per CU for its LDS) -- the reduction rides along instead of holding the stream.
template <class DT, int NT, int NW = 4>
__global__ void __launch_bounds__(64 * NW) wgrad_group_kernel(const WgradGroupArgs g, const WgradReduceGroupArgs r) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    // grid order: the first `lead` main workgroups (one per CU: they take the LDS), then the reduction blocks (no LDS: they
    // fit beside them), then the other main workgroups.  The dispatcher works in order: with the reduction in front the main
    // workgroups would wait for it to be dispatched, behind them it would wait for LDS to come free.
    const int nred = r.n ? r.first[r.n] : 0, lead = min(g.first[g.n], 256);
    int bid = (int)blockIdx.x;
    if (bid >= lead && bid < lead + nred) {
        if (threadIdx.x >= 256) return;
        wgrad_reduce_group_body(r, bid - lead);
        return;
    }
    if (bid >= lead) bid -= nred;
    int idx = 0;
#pragma unroll
    for (int k = 1; k < RSG_WGRAD_GROUP_MAX; ++k) idx += (k < g.n && bid >= g.first[k]) ? 1 : 0;
    WgradArgs a;
    int block = 0;
#pragma unroll
    for (int k = 0; k < RSG_WGRAD_GROUP_MAX; ++k)
        if (k == idx) { a = g.a[k]; block = bid - g.first[k]; }
    if (a.quad) wgrad_quad_body<DT, NT, NW>(a, block, smem);
    else {
        if (NW == 8 && threadIdx.x >= 256) return;
        wgrad_body<DT, NT>(a, block, smem);
    }
}

struct WgradPlan { WgradArgs a; int lds, grid, ntaps; };
static int g_wgrad_quad = 1;       // RSGNET_HIP_WGRAD_QUAD=0: the one-pair form everywhere (A/B)

// Slice policy (rsg_wgrad_tune): every (tile pair, slice) writes a taps x 4 KB partial tile that the reduce kernel
// reads back, so a slice must own enough pixels to be worth it.  g_wgrad_min_px: lower bound on the pixels per slice
// (0 = none); g_wgrad_quad_min_wgs: the 2x2 form only where it still gives this many workgroups, else one pair per
// workgroup (four times the columns, a quarter of the partial bytes per workgroup).
static int g_wgrad_min_px = 0;
static int g_wgrad_quad_min_wgs = 0;
static int g_wgrad_lds_budget = 128 * 1024;  // LDS per workgroup = chunk size.  Same-visit sweep of the training step: 44 KB 16.66 ms,
                                             // 72 KB (two workgroups per CU) 15.77, 96 KB 15.73, 128 KB 15.41: larger chunks beat more workgroups
static int g_wgrad_group_wgs = 0;            // workgroups a grouped launch aims at in total; 0 = as many as fit one round (rsg_wgrad_tune: min_px slot < 0 sets -value)
static int g_wgrad_waves = 8;                // waves per workgroup of the 2x2 form (rsg_wgrad_tune): 4, or 8 = two per tile pair
static int g_wgrad_px_cap = 768;             // output pixels of a chunk (128 KB: 512 -> 15.39 ms, 640 15.20, 768 14.66, 896 14.72, 1024 14.79)

static int wgrad_plan_form(int batch, int in_h, int in_w, int cin_pad, int out_h, int out_w, int cout_pad, int kh, int kw, int stride,
                           int pad, WgradPlan& p, int target_wgs, int quad, bool sizing = false) {
    memset(&p, 0, sizeof(p));
    if (!((kh == 3 && kw == 3 && pad == 1) || (kh == 1 && kw == 1 && pad == 0)) || (stride != 1 && stride != 2)) return 1;
    if (cin_pad % 8 || cout_pad % 8 || batch < 1) return 1;
    if ((in_h + 2 * pad - kh) / stride + 1 != out_h || (in_w + 2 * pad - kw) / stride + 1 != out_w) return 1;
    WgradArgs& a = p.a;
    a.B = batch; a.Hin = in_h; a.Win = in_w; a.CinP = cin_pad; a.Hout = out_h; a.Wout = out_w; a.CoutP = cout_pad;
    a.kh = kh; a.kw = kw; a.stride = stride; a.pad = pad;
    a.n_ot = ceil_div(cout_pad, 32); a.n_it = ceil_div(cin_pad, 32);
    a.xcols = in_w + 2 * pad;
    p.ntaps = kh * kw;
    a.quad = quad;
    const int pitch = a.quad ? 144 : 64;
    const int red_bytes = a.quad ? 0 : 2 * p.ntaps * 16 * 64 * 4;
    const int budget = g_wgrad_lds_budget;
    int rb = 0;
    for (int r = max(1, min(out_h, g_wgrad_px_cap / out_w)); r >= 1; --r) {
        const int p16 = round_up(r * out_w, 16), xr = (r - 1) * stride + kh;
        if (p16 * pitch + xr * a.xcols * pitch + p16 * 4 <= budget) { rb = r; break; }
    }
    if (!rb) return 1;
    a.RB = rb; a.nbands = ceil_div(out_h, rb);
    a.xrows = (rb - 1) * stride + kh;
    a.nimg = 1;
    if (a.nbands == 1) {                                    // small maps: several whole images per chunk
        for (int n = min(batch, g_wgrad_px_cap / max(1, out_h * out_w)); n >= 2; --n) {
            const int p16 = round_up(n * out_h * out_w, 16);
            if (p16 * pitch + n * a.xrows * a.xcols * pitch + p16 * 4 <= budget) { a.nimg = n; break; }
        }
    }
    a.P16 = round_up(a.nimg * rb * out_w, 16);
    const int red8 = (a.quad && g_wgrad_waves == 8) ? 4 * (p.ntaps >= 5 ? 5 : p.ntaps) * 4096 : 0;   // wgrad_quad_body<.., 8>
    p.lds = max(max(a.P16 * pitch + a.nimg * a.xrows * a.xcols * pitch + a.P16 * 4, red_bytes), red8);
    // pixel slices per workgroup column (a tile pair, or a 2x2 block of pairs): enough workgroups to fill the chip (2 per
    // CU), but every slice pays a fixed price (a taps x 4 KB partial tile per pair written and read back; the one-pair form
    // also its cross-wave add through LDS), so a slice should own several chunks when the columns alone already give
    // the parallelism (256-channel maps: 64 pairs of 8x6 chunks)
    const int cols = a.quad ? ceil_div(a.n_ot, 2) * ceil_div(a.n_it, 2) : a.n_ot * a.n_it;
    const int chunks = a.nimg > 1 ? ceil_div(batch, a.nimg) : batch * a.nbands;
    a.nslices = max(1, min(chunks, ceil_div(target_wgs, cols)));
    if (g_wgrad_min_px > 0 && !sizing) a.nslices = max(1, min(a.nslices, (int)((long long)batch * out_h * out_w / g_wgrad_min_px)));
    p.grid = cols * a.nslices;
    return 0;
}

static int wgrad_plan(int batch, int in_h, int in_w, int cin_pad, int out_h, int out_w, int cout_pad, int kh, int kw, int stride,
                      int pad, WgradPlan& p, int target_wgs = 512) {
    const bool can_quad = ceil_div(cout_pad, 32) >= 2 && ceil_div(cin_pad, 32) >= 2 && g_wgrad_quad;
    if (can_quad) {
        if (!wgrad_plan_form(batch, in_h, in_w, cin_pad, out_h, out_w, cout_pad, kh, kw, stride, pad, p, target_wgs, 1) &&
            p.grid >= g_wgrad_quad_min_wgs)
            return 0;
    }
    return wgrad_plan_form(batch, in_h, in_w, cin_pad, out_h, out_w, cout_pad, kh, kw, stride, pad, p, target_wgs, 0);
}

extern "C" void rsg_wgrad_tune(int32_t min_px_per_slice, int32_t quad_min_wgs, int32_t lds_budget, int32_t px_cap, int32_t waves) {
    if (waves == 4 || waves == 8) g_wgrad_waves = waves;
    if (px_cap >= 64 && px_cap <= 4096) g_wgrad_px_cap = px_cap;
    if (min_px_per_slice < 0) { g_wgrad_group_wgs = -min_px_per_slice; min_px_per_slice = 0; }      // experiment: total workgroups of a grouped launch
    g_wgrad_min_px = min_px_per_slice > 0 ? min_px_per_slice : 0;
    g_wgrad_quad_min_wgs = quad_min_wgs > 0 ? quad_min_wgs : 0;
    if (lds_budget >= 16 * 1024 && lds_budget <= 144 * 1024) g_wgrad_lds_budget = lds_budget;
}

extern "C" void rsg_wgrad_set_quad(int32_t on) { g_wgrad_quad = on ? 1 : 0; }

extern "C" size_t rsg_conv_wgrad_workspace_bytes(int32_t batch, int32_t in_h, int32_t in_w, int32_t cin_pad, int32_t out_h,
                                                 int32_t out_w, int32_t cout_pad, int32_t kh, int32_t kw, int32_t stride, int32_t pad) {
    // the larger of the two forms at the single launch's slice count: whatever the policy (rsg_wgrad_tune, the grouped
    // launch's smaller target) picks later needs no more
    WgradPlan p;
    size_t need = 0;
    for (int quad = 0; quad < 2; ++quad) {
        if (quad && !(ceil_div(cout_pad, 32) >= 2 && ceil_div(cin_pad, 32) >= 2)) continue;
        if (wgrad_plan_form(batch, in_h, in_w, cin_pad, out_h, out_w, cout_pad, kh, kw, stride, pad, p, 512, quad, true)) continue;
        need = std::max(need, (size_t)p.a.n_ot * p.a.n_it * p.a.nslices * p.ntaps * 1024 * sizeof(float));
    }
    return need;
}

extern "C" int rsg_conv_wgrad(const void* x, const void* dy, float* dw, void* workspace, int32_t batch, int32_t in_h, int32_t in_w,
                              int32_t cin, int32_t cin_pad, int32_t out_h, int32_t out_w, int32_t cout, int32_t cout_pad, int32_t kh,
                              int32_t kw, int32_t stride, int32_t pad, int32_t dtype, void* stream) {
    if (!x || !dy || !dw || !workspace) RSG_FAIL("conv_wgrad: null pointer");
    if (((uintptr_t)x | (uintptr_t)dy | (uintptr_t)workspace) & 15) RSG_FAIL("conv_wgrad: operands must be 16-byte aligned");
    if (cin < 1 || cin > cin_pad || cout < 1 || cout > cout_pad) RSG_FAIL("conv_wgrad: cin %d/%d cout %d/%d", cin, cin_pad, cout, cout_pad);
    WgradPlan p;
    if (wgrad_plan(batch, in_h, in_w, cin_pad, out_h, out_w, cout_pad, kh, kw, stride, pad, p))
        RSG_FAIL("conv_wgrad: unsupported geometry %dx%d k%d s%d p%d (%dx%d -> %dx%d)", cin_pad, cout_pad, kh, stride, pad, in_h, in_w, out_h, out_w);
    if ((long long)batch * in_h * in_w * cin_pad * 2 >= (1LL << 31) || (long long)batch * out_h * out_w * cout_pad * 2 >= (1LL << 31))
        RSG_FAIL("conv_wgrad: map too large for 32-bit offsets");
    p.a.x = (const char*)x; p.a.dy = (const char*)dy; p.a.part = (float*)workspace;
    hipStream_t s = (hipStream_t)stream;
    void (*fn)(const WgradArgs) = nullptr;
    const bool w8 = g_wgrad_waves == 8;
    if (dtype == RSG_BF16) fn = p.ntaps == 9 ? (w8 ? wgrad_kernel<BF16, 9, 8> : wgrad_kernel<BF16, 9>) : (w8 ? wgrad_kernel<BF16, 1, 8> : wgrad_kernel<BF16, 1>);
    else fn = p.ntaps == 9 ? (w8 ? wgrad_kernel<F16, 9, 8> : wgrad_kernel<F16, 9>) : (w8 ? wgrad_kernel<F16, 1, 8> : wgrad_kernel<F16, 1>);
    if (hipFuncSetAttribute((const void*)fn, hipFuncAttributeMaxDynamicSharedMemorySize, p.lds) != hipSuccess)
        RSG_FAIL("conv_wgrad: cannot reserve %d B of LDS", p.lds);
    RSG_LAUNCH(fn, dim3(p.grid), dim3(w8 ? 512 : 256), p.lds, s, p.a);
    RSG_LAUNCH(wgrad_reduce_kernel, dim3(p.a.n_ot * p.a.n_it * p.ntaps * 4), dim3(256), 0, s, (const float*)workspace, dw, cout, cin,
               p.ntaps, p.a.n_it, p.a.nslices);
    RSG_CHECK_LAUNCH("conv_wgrad");
    return 0;
}


// Compute units of the current device (cached per device id); 256 when the query fails.
static int wgrad_device_cus() {
    static int cus[64] = {0};
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return 256;
    if (!cus[dev]) {
        int v = 0;
        cus[dev] = (hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && v > 0) ? v : 256;
    }
    return cus[dev];
}

// Fills the main-kernel table (g) and the reduction table (r) of one group of weight gradients.
static int wgrad_group_tables(const rsg_wgrad_item* items, int n, const char* what, WgradGroupArgs* g, WgradReduceGroupArgs& r,
                              int& lds, int& blocks, int& rblocks, int& ntaps, bool pipelined) {
    if (!items || n < 1 || n > RSG_WGRAD_GROUP_MAX) RSG_FAIL("%s: 1..%d items", what, RSG_WGRAD_GROUP_MAX);
    for (int k = 0; k < n; ++k) {
        const rsg_wgrad_item& it = items[k];
        if (!it.x || !it.dy || !it.dw || !it.workspace) RSG_FAIL("%s: item %d: null pointer", what, k);
        if (((uintptr_t)it.x | (uintptr_t)it.dy | (uintptr_t)it.workspace) & 15) RSG_FAIL("%s: item %d: alignment", what, k);
        if (it.cin < 1 || it.cin > it.cin_pad || it.cout < 1 || it.cout > it.cout_pad) RSG_FAIL("%s: item %d: channels", what, k);
        if ((long long)it.batch * it.in_h * it.in_w * it.cin_pad * 2 >= (1LL << 31) || (long long)it.batch * it.out_h * it.out_w * it.cout_pad * 2 >= (1LL << 31))
            RSG_FAIL("%s: item %d: map too large for 32-bit offsets", what, k);
    }
    // Slices per item.  A workgroup holds a whole CU (128 KB of LDS), so a launch runs in rounds of one workgroup per CU
    // and a round that is not full costs as much as a full one: the largest per-item target (never above the 512 the
    // workspaces are sized for) whose workgroups still fit ONE round -- three quarters of a round when the launch also
    // carries the previous group's reduction (pipelined form), whose blocks then find free CUs from the start.  Same-visit
    // sweeps of the training step by total workgroups of a 4-item launch, pipelined: 768 (the first policy) 13.72 ms,
    // 320 13.63, 288 13.75, 256 13.25-13.37, 224 13.19, 192 13.03-13.07, 160 13.50; not pipelined: 256 13.40, 192 13.75.
    auto total_for = [&](int target) {
        int tot = 0;
        for (int k = 0; k < n; ++k) {
            const rsg_wgrad_item& it = items[k];
            WgradPlan p;
            if (wgrad_plan(it.batch, it.in_h, it.in_w, it.cin_pad, it.out_h, it.out_w, it.cout_pad, it.kh, it.kw, it.stride, it.pad, p, target)) return -1;
            tot += p.grid;
        }
        return tot;
    };
    int target = min(512, max(16, g_wgrad_group_wgs / n));
    if (g_wgrad_group_wgs == 0) {                            // default: fit one round
        const int cus = pipelined ? wgrad_device_cus() * 3 / 4 : wgrad_device_cus();
        int lo = 1, hi = 512;                                // total_for is monotonic in the target
        if (total_for(lo) < 0) RSG_FAIL("%s: unsupported geometry", what);
        while (lo < hi) {
            const int mid = (lo + hi + 1) >> 1;
            if (total_for(mid) <= cus) lo = mid; else hi = mid - 1;
        }
        target = lo;
    }
    if (g) memset(g, 0, sizeof(*g));
    memset(&r, 0, sizeof(r));
    if (g) g->n = n;
    r.n = n;
    lds = 0; blocks = 0; rblocks = 0; ntaps = 0;
    for (int k = 0; k < n; ++k) {
        const rsg_wgrad_item& it = items[k];
        WgradPlan p;
        if (wgrad_plan(it.batch, it.in_h, it.in_w, it.cin_pad, it.out_h, it.out_w, it.cout_pad, it.kh, it.kw, it.stride, it.pad, p, target))
            RSG_FAIL("%s: item %d: unsupported geometry", what, k);
        if (k && p.ntaps != ntaps) RSG_FAIL("%s: every item must have the same kernel size", what);
        ntaps = p.ntaps;
        p.a.x = (const char*)it.x; p.a.dy = (const char*)it.dy; p.a.part = (float*)it.workspace;
        if (g) { g->a[k] = p.a; g->first[k] = blocks; }
        blocks += p.grid;
        lds = max(lds, p.lds);
        r.part[k] = (const float*)it.workspace; r.dw[k] = it.dw; r.cout[k] = it.cout; r.cin[k] = it.cin;
        r.n_it[k] = p.a.n_it; r.nslices[k] = p.a.nslices;
        r.first[k] = rblocks; rblocks += p.a.n_ot * p.a.n_it * p.ntaps * 4;
    }
    if (g) g->first[n] = blocks;
    r.first[n] = rblocks; r.ntaps = ntaps;
    return 0;
}

// Pipelined form: ONE launch runs the main kernel of `items` and, in its first workgroups, the reduction of `prev`
// (a group whose main kernel ran in an EARLIER launch on this stream with defer_reduce set).  defer_reduce = 1 leaves
// this group's partial tiles in its workspaces for the next call; 0 appends its reduction as a second launch, as
// rsg_conv_wgrad_group does.  items == NULL (n = 0): only prev's reduction.  prev's workspaces must not be workspaces
// of `items` (the two run concurrently).
extern "C" int rsg_conv_wgrad_group_ex(const rsg_wgrad_item* items, int32_t n, const rsg_wgrad_item* prev, int32_t n_prev,
                                       int32_t defer_reduce, int32_t dtype, void* stream) {
    hipStream_t s = (hipStream_t)stream;
    WgradReduceGroupArgs rp;
    memset(&rp, 0, sizeof(rp));
    int plds = 0, pblocks = 0, prb = 0, ptaps = 0;
    // a group that deferred its reduction was planned as pipelined (below): the same plan again
    if (n_prev > 0 && wgrad_group_tables(prev, n_prev, "conv_wgrad_group_ex(prev)", nullptr, rp, plds, pblocks, prb, ptaps, true)) return 1;
    if (n <= 0 || !items) {
        if (n_prev > 0) RSG_LAUNCH(wgrad_reduce_group_kernel, dim3(prb), dim3(256), 0, s, rp);
        RSG_CHECK_LAUNCH("conv_wgrad_group_ex");
        return 0;
    }
    WgradGroupArgs g;
    WgradReduceGroupArgs r;
    int lds = 0, blocks = 0, rblocks = 0, ntaps = 0;
    if (wgrad_group_tables(items, n, "conv_wgrad_group", &g, r, lds, blocks, rblocks, ntaps, defer_reduce || n_prev > 0)) return 1;
    for (int k = 0; k < n; ++k)
        for (int j = 0; j < n_prev; ++j)
            if (items[k].workspace == prev[j].workspace) RSG_FAIL("conv_wgrad_group_ex: item %d shares its workspace with the group being reduced", k);
    void (*fn)(const WgradGroupArgs, const WgradReduceGroupArgs) = nullptr;
    const bool w8 = g_wgrad_waves == 8;
    if (dtype == RSG_BF16) fn = ntaps == 9 ? (w8 ? wgrad_group_kernel<BF16, 9, 8> : wgrad_group_kernel<BF16, 9>) : (w8 ? wgrad_group_kernel<BF16, 1, 8> : wgrad_group_kernel<BF16, 1>);
    else fn = ntaps == 9 ? (w8 ? wgrad_group_kernel<F16, 9, 8> : wgrad_group_kernel<F16, 9>) : (w8 ? wgrad_group_kernel<F16, 1, 8> : wgrad_group_kernel<F16, 1>);
    if (hipFuncSetAttribute((const void*)fn, hipFuncAttributeMaxDynamicSharedMemorySize, lds) != hipSuccess)
        RSG_FAIL("conv_wgrad_group: cannot reserve %d B of LDS", lds);
    RSG_LAUNCH(fn, dim3(blocks + prb), dim3(w8 ? 512 : 256), lds, s, g, rp);
    if (!defer_reduce) RSG_LAUNCH(wgrad_reduce_group_kernel, dim3(rblocks), dim3(256), 0, s, r);
    RSG_CHECK_LAUNCH("conv_wgrad_group");
    return 0;
}

extern "C" int rsg_conv_wgrad_group(const rsg_wgrad_item* items, int32_t n, int32_t dtype, void* stream) {
    if (!items || n < 1) RSG_FAIL("conv_wgrad_group: 1..%d items", RSG_WGRAD_GROUP_MAX);
    return rsg_conv_wgrad_group_ex(items, n, nullptr, 0, 0, dtype, stream);
}
