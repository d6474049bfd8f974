#pragma once
#include "common.h"

struct PrepEntry {
    const float* src;
    void* dst;
    int rows, cols, ld;   // destination geometry
    int mode, bf16;
    int cin, cout, k, py, px, T, s;
    int KT, ksplit, kpad0, n_off;   // mode 4/5: MFMA-fragment packing (chain.hip)
};
#define SP_MAX_PREP 32
struct PrepTable {
    int n;
    PrepEntry e[SP_MAX_PREP];
};

int misc_pad_input(const float* x, float* xp, int B, int C, int I, int pre, int Ip, hipStream_t s);
int misc_prep(const PrepTable& T, hipStream_t s);
int misc_export(const float* src, int ld, int col0, int ch, const int* cell_h, const int* cell_w, int B, int G, float* out, hipStream_t s);
int misc_conv0_fwd(const float* xp, const float* w, const float* bias, float* out, int B, int Hin, int C, int k, int s, int Hout, int Cout, int out_bf16, hipStream_t st);
int misc_conv0_wgrad(const float* xp, const float* dout, float* dw, int B, int Hin, int C, int k, int s, int Hout, int Cout, hipStream_t st);
