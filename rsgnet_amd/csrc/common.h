// Shared helpers for librsgnet_hip (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>
#include <tuple>
#include <utility>
#include "../../include/rsgnet_hip.h"

typedef _Float16 h8_t __attribute__((ext_vector_type(8)));
typedef __bf16 b8_t __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x8 __attribute__((ext_vector_type(8)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

#define RSG_MAX_DEVICES 64       // per-device caches of launch attributes (hipFuncSetAttribute is per device)

extern thread_local char g_rsg_err[512];

#define RSG_FAIL(...)                                        \
    do {                                                     \
        snprintf(g_rsg_err, sizeof(g_rsg_err), __VA_ARGS__); \
        return 1;                                            \
    } while (0)

// Every kernel launch goes through rsg_launch_raw (graph.hip): a plain hipLaunchKernel, or a kernel node
// of the HIP graph that is open on this thread.
extern thread_local int g_rsg_launch_status;
int rsg_launch_raw(const void* fn, dim3 grid, dim3 block, unsigned lds, hipStream_t stream, void** params);

template <class Tuple, size_t... I>
static inline void rsg_fill_params(void** p, Tuple& t, std::index_sequence<I...>) {
    ((p[I] = (void*)&std::get<I>(t)), ...);
}
template <class... KArgs, class... Args>
static inline int rsg_launch(void (*fn)(KArgs...), dim3 grid, dim3 block, unsigned lds, hipStream_t stream,
                             Args... args) {
    static_assert(sizeof...(KArgs) == sizeof...(Args), "kernel argument count");
    std::tuple<KArgs...> t{static_cast<KArgs>(args)...};
    void* params[sizeof...(KArgs)];
    rsg_fill_params(params, t, std::index_sequence_for<KArgs...>{});
    return rsg_launch_raw((const void*)fn, grid, block, lds, stream, params);
}
#define RSG_LAUNCH(fn, grid, block, lds, stream, ...) rsg_launch(fn, grid, block, lds, stream, __VA_ARGS__)

#define RSG_CHECK_LAUNCH(what)                                                     \
    do {                                                                           \
        if (g_rsg_launch_status) {                                                 \
            char m__[400];                                                         \
            snprintf(m__, sizeof(m__), "%s", g_rsg_err);                           \
            RSG_FAIL("%s: launch failed: %s", what, m__);                          \
        }                                                                          \
    } while (0)

// 16-bit storage types: one struct per dtype so kernels can be templated on it.
struct F16 {
    typedef _Float16 elem;
    typedef h8_t vec8;
    static __device__ __forceinline__ f32x16 mfma(vec8 a, vec8 b, f32x16 c) {
        return __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0);
    }
    static __device__ __host__ __forceinline__ float to_f32(elem v) { return (float)v; }
    static __device__ __host__ __forceinline__ elem from_f32(float v) { return (elem)v; }
    // ReLU of 8 fp32 values as 16-bit: round first (v_cvt_pk_f16_f32), then v_pk_max_f16 with 0 -- one instruction
    // per element for both, and max(round(x), 0) == round(max(x, 0))
    static __device__ __forceinline__ vec8 relu_pack(f32x8 v) {
        const vec8 z = {0, 0, 0, 0, 0, 0, 0, 0};
        return __builtin_elementwise_max(__builtin_convertvector(v, vec8), z);
    }
};
struct BF16 {
    typedef __bf16 elem;
    typedef b8_t vec8;
    static __device__ __forceinline__ f32x16 mfma(vec8 a, vec8 b, f32x16 c) {
        return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0);
    }
    static __device__ __host__ __forceinline__ float to_f32(elem v) { return (float)v; }
    static __device__ __host__ __forceinline__ elem from_f32(float v) { return (elem)v; }
    // ReLU of 8 fp32 values as bf16: round first (v_cvt_pk_bf16_f32), then v_pk_max_i16 with 0 on the bit patterns -- a
    // negative bf16 is a negative int16 (sign bit), a non-negative one keeps its pattern; gfx950 has no packed bf16 max, but
    // this costs what F16's v_pk_max_f16 costs.  max(round(x), 0) == round(max(x, 0)); -0 becomes +0 either way.
    static __device__ __forceinline__ vec8 relu_pack(f32x8 v) {
        typedef short s8_t __attribute__((ext_vector_type(8)));
        const s8_t z = {0, 0, 0, 0, 0, 0, 0, 0};
        return __builtin_bit_cast(vec8, __builtin_elementwise_max(__builtin_bit_cast(s8_t, __builtin_convertvector(v, vec8)), z));
    }
};

static inline int round_up(int a, int b) { return (a + b - 1) / b * b; }
static inline int ceil_div(int a, int b) { return (a + b - 1) / b; }

// XCD-aware tile order.  The dispatcher deals consecutive workgroup ids round-robin to the 8 XCDs, each with its own
// 4 MB L2, so neighbouring tiles of one picture (which share halo rows) and the two row bands of a chain image would
// land in 8 different L2s.  This permutation gives every XCD one contiguous range of the n logical tiles instead:
// residue class c = l % 8 (one XCD, whatever the phase of the first id) takes tiles [start(c), start(c) + size(c)).
#ifndef RSG_XCD_SWIZZLE
#define RSG_XCD_SWIZZLE 1
#endif
__device__ __forceinline__ int rsg_xcd_tile(int l, int n) {
#if RSG_XCD_SWIZZLE
    const int c = l & 7, per = n >> 3, rem = n & 7;
    return c * per + (c < rem ? c : rem) + (l >> 3);
#else
    return l;
#endif
}
