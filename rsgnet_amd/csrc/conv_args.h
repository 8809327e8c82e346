// Kernel-argument structs shared by conv_igemm.hip and group.hip.
#pragma once
#include "common.h"

struct ConvArgs {
    const char* in;
    const uint4* wpk;
    const float* bias;
    const char* res;
    const char* res2;          // further addends of the epilogue (HRNet fuse sum folded into the conv), or null
    const char* res3;
    char* out;
    float* out32;
    int B, Hin, Win, ldi;
    int Hout, Wout, Cout, ldo, ldr, ldr2, ldr3;
    int CoutStore;   // channels [0, CoutStore) are written to the 16-bit output (multiple of 8)
    // conv_pair / conv_pair_next tails only: elements between the 32-channel tiles of `out` and of `res` (32 = the dense
    // NHWC map; batch*H*W*32 with ldo / ldr = 32: the map as channel-tile PLANES [C/32][pixel][32], in which a wave's 32
    // pixels x 64 B of an accumulator tile are one contiguous 2 KB run instead of 32 pieces at a 512-byte stride)
    int tso, tsr;
    int SI, oy0, ox0;
    int up, upy, upx, Hp, Wp;
    int relu;
    int TH, TW, TIH, TIW, RP;
    int tiles_x, tiles_y;
    int nchunks, KS;
    // fused 1x1 tail (conv_body<..., TAIL=true>, rsg_conv_pair): the conv above is an inner stage whose ReLU'd
    // 64-channel result stays in LDS; out/res/ldo/ldr/relu/Cout/CoutStore describe the TAIL's output
    const uint4* wpk2;
    const float* bias2;
    int C2;          // output channels of the tail (multiple of 32)

    // fused 1x1 head (conv_body<..., HEAD=true>, rsg_bottleneck_fused): the conv's 64-channel input tile is not read
    // from `in` but computed in place, ReLU(conv1x1(in0) + bias0), from a 256-channel map (Bottleneck conv1)
    const char* in0;
    const uint4* wpk0;      // packed with the head's channel permutation (rsg_bottleneck_head_perm)
    const float* bias0;
    int ldi0;
};

// rsg_conv_pair_next only (kept out of ConvArgs: twelve of those travel as ONE kernel argument in group.hip)
struct ConvNextArgs : ConvArgs {
    // the NEXT block's conv1 on the tail's output (conv_body<..., NEXT>): 256 -> 64, ReLU
    const uint4* wpk3;      // packed with the accumulator-as-operand channel order (rsg_conv_pair_next_perm)
    const float* bias3;
    char* out3;
    int ldo3;
    const char* in2;        // KCAT: the block input x (64 channels per pixel, stride ldi2), second half of the tail's K
    int ldi2;
};

struct TilePlan {
    int CINB, WM, WN, NT, TH, TW, TIH, TIW, RP, lds, tiles_x, tiles_y;
};

int rsg_conv_prepare(const rsg_conv_desc* d, ConvArgs& a, TilePlan& t, dim3& grid);
