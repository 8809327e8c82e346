// HighResolutionModule fuse-sum body (pose_hrnet.py:259-266), shared by elementwise.hip and group.hip.
#pragma once
#include "common.h"

struct FuseArgs {
    const char* same[4];
    const char* low[3];
    int shift[3];
    int n_same, n_low;
    char* out;
    int h, w, c8;   // c8 = channels / 8
    int ld_same[4]; // 16-byte vectors per pixel of each operand (c8 when dense)
    int ld_low[3];
    long total;     // batch*h*w*c8 16-byte vectors
};

// One thread per 8 channels of a pixel.  32-bit index arithmetic (the host checks total < 2^31) and operand loops
// unrolled over the fixed maxima with constant indices: the operand pointers and strides then come from the kernel
// arguments once, and all loads of an element are in flight together.  (The first version divided 64-bit indices five
// times per element and walked the operand lists with run-time indices: 19 us for a sum that moves 26 MB.)
template <class DT>
__device__ __forceinline__ void fuse_body(const FuseArgs& a, const int block, const int nblocks) {
    const unsigned total = (unsigned)a.total, c8 = (unsigned)a.c8, w = (unsigned)a.w, h = (unsigned)a.h;
    const int n_same = a.n_same, n_low = a.n_low;
    for (unsigned i = (unsigned)block * 256u + threadIdx.x; i < total; i += (unsigned)nblocks * 256u) {
        const unsigned pix0 = i / c8, cg = i - pix0 * c8;
        uint4 vs[4], vl[3];
#pragma unroll
        for (int k = 0; k < 4; ++k)
            if (k < n_same) vs[k] = reinterpret_cast<const uint4*>(a.same[k])[(size_t)pix0 * a.ld_same[k] + cg];
        if (n_low) {
            const unsigned t = pix0 / w, x = pix0 - t * w;
            const unsigned b = t / h, y = t - b * h;
#pragma unroll
            for (int k = 0; k < 3; ++k)
                if (k < n_low) {
                    const int s = a.shift[k];
                    const size_t src = (((size_t)b * (h >> s) + (y >> s)) * (w >> s) + (x >> s)) * a.ld_low[k] + cg;
                    vl[k] = reinterpret_cast<const uint4*>(a.low[k])[src];
                }
        }
        float acc[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) acc[e] = 0.f;
#pragma unroll
        for (int k = 0; k < 4; ++k)
            if (k < n_same) {
                const typename DT::vec8 v = __builtin_bit_cast(typename DT::vec8, vs[k]);
#pragma unroll
                for (int e = 0; e < 8; ++e) acc[e] += (float)v[e];
            }
#pragma unroll
        for (int k = 0; k < 3; ++k)
            if (k < n_low) {
                const typename DT::vec8 v = __builtin_bit_cast(typename DT::vec8, vl[k]);
#pragma unroll
                for (int e = 0; e < 8; ++e) acc[e] += (float)v[e];
            }
        f32x8 f;
#pragma unroll
        for (int e = 0; e < 8; ++e) f[e] = acc[e];
        reinterpret_cast<uint4*>(a.out)[i] = __builtin_bit_cast(uint4, DT::relu_pack(f));
    }
}


int rsg_fuse_prepare(const void* const* same, int n_same, const void* const* low, const int32_t* low_shift,
                     int n_low, void* out, int batch, int h, int w, int c, FuseArgs& a, int& blocks,
                     const int32_t* same_ld = nullptr, const int32_t* low_ld = nullptr);
