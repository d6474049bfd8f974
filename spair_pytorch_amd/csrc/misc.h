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
int misc_export16(const void* src, int ld, int col0, int ch, const int* cell_h, const int* cell_w, int B, int G, float* out, hipStream_t s);
// x: the unpadded image [B,1,I,I] (read directly when misc_conv0_reads_unpadded), xp: its zero-padded copy [B,Hin,Hin,C] (the general path)
bool misc_conv0_reads_unpadded(int B, int Hin, int C, int k, int Cout);
bool misc_conv0_writes_mask(int B, int Hin, int C, int k, int s, int Cout, int out_bf16);      // the stem kernel that can leave the sign-bit mask
int misc_conv0_fwd(const float* x, const float* xp, const float* w, const float* bias, float* out, int B, int I, int pre, int Hin, int C, int k, int s,
                   int Hout, int Cout, int out_bf16, hipStream_t st, unsigned char* mask = nullptr);
int misc_conv0_wgrad(const float* xp, const float* dout, float* dw, int B, int Hin, int C, int k, int s, int Hout, int Cout, hipStream_t st);
