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

template <class DT>
__device__ __forceinline__ void fuse_body(const FuseArgs& a, const int block, const int nblocks) {
    for (long i = (long)block * 256 + threadIdx.x; i < a.total; i += (long)nblocks * 256) {
        float acc[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) acc[e] = 0.f;
        const int cg0 = (int)(i % a.c8);
        const long pix0 = i / a.c8;
        for (int k = 0; k < a.n_same; ++k) {
            const typename DT::vec8 v = __builtin_bit_cast(
                typename DT::vec8, reinterpret_cast<const uint4*>(a.same[k])[pix0 * a.ld_same[k] + cg0]);
#pragma unroll
            for (int e = 0; e < 8; ++e) acc[e] += (float)v[e];
        }
        if (a.n_low) {
            const int cg = (int)(i % a.c8);
            long pix = i / a.c8;
            const int x = (int)(pix % a.w);
            pix /= a.w;
            const int y = (int)(pix % a.h);
            const long b = pix / a.h;
            for (int k = 0; k < a.n_low; ++k) {
                const int s = a.shift[k];
                const long src = ((b * (a.h >> s) + (y >> s)) * (a.w >> s) + (x >> s)) * a.ld_low[k] + cg;
                const typename DT::vec8 v =
                    __builtin_bit_cast(typename DT::vec8, reinterpret_cast<const uint4*>(a.low[k])[src]);
#pragma unroll
                for (int e = 0; e < 8; ++e) acc[e] += (float)v[e];
            }
        }
        typename DT::vec8 o;
#pragma unroll
        for (int e = 0; e < 8; ++e) o[e] = (typename DT::elem)fmaxf(acc[e], 0.f);
        reinterpret_cast<uint4*>(a.out)[i] = __builtin_bit_cast(uint4, o);
    }
}


int rsg_fuse_prepare(const void* const* same, int n_same, const void* const* low, const int32_t* low_shift,
                     int n_low, void* out, int batch, int h, int w, int c, FuseArgs& a, int& blocks,
                     const int32_t* same_ld = nullptr, const int32_t* low_ld = nullptr);
