#pragma once
#include "common.h"
// dec_fused.hip: the object decoder's forward (z_attr -> 128 -> 256 -> P*P*2 logits -> sprites) as ONE activation-stationary kernel
bool dec_fused_supported(int A, int n_out, int ld_za, long long N, int ld_s);
size_t dec_fused_stream_bytes(int n_out);
// packs the three weight matrices (fp32 parameters, row-major [out][in]) into the kernel's fragment stream
// (decoder.out's rows pre-multiplied by -scale * log2 e: the kernel's accumulator is the exp2 argument of the sprite sigmoid)
int dec_fused_pack(const float* W0, const float* W1, const float* W2, int A, int n_out, float obj_scale, float alpha_scale, void* stream_buf,
                   hipStream_t s);
int dec_fused_fwd(const void* Za16, int ld_za, const void* stream_buf, const float* b0, const float* b1, const float* b2, void* H1, void* H2,
                  void* S, int ld_s, long long N, int A, int n_out, float obj_scale, float alpha_scale, float alpha_bias, hipStream_t s);
// dec_fused_bwd.hip: the decoder's data-gradient chain (d-logits -> dH2 -> dH1 -> d z_attr) as ONE kernel; transposed bf16 weights (k_prep mode 1)
bool dec_fused_bwd_supported(int A, int n_out, long long N, int ld_s, int ld2, int ld_dza);
int dec_fused_bwd(const void* dL, int ld_s, const void* W2t, int ld2, const void* W1t, const void* W0t, const void* H2, const void* H1, void* dH2,
                  void* dH1, float* dza, int ld_dza, long long N, int A, int n_out, hipStream_t s);
