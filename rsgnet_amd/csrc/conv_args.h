// Kernel-argument structs shared by conv_igemm.hip and group.hip.
#pragma once
#include "common.h"

struct ConvArgs {
    const char* in;
    const uint4* wpk;
    const float* bias;
    const char* res;
    char* out;
    float* out32;
    int B, Hin, Win, ldi;
    int Hout, Wout, Cout, ldo, ldr;
    int CoutStore;   // channels [0, CoutStore) are written to the 16-bit output (multiple of 8)
    int SI, oy0, ox0;
    int up, upy, upx, Hp, Wp;
    int relu;
    int TH, TW, TIH, TIW, RP;
    int tiles_x, tiles_y;
    int nchunks, KS;
};

struct TilePlan {
    int CINB, WM, WN, NT, TH, TW, TIH, TIW, RP, lds, tiles_x, tiles_y;
};

int rsg_conv_prepare(const rsg_conv_desc* d, ConvArgs& a, TilePlan& t, dim3& grid);
