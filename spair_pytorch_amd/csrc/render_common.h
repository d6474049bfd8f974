// Device helpers shared by the renderer kernels (render.hip: first-generation kernels, kept as the generic fallback and the fp32
// backward; render2.hip: LDS-staged forward and the one-wave-per-object backward of the bf16 step).
#pragma once
#include "cells.h"
#include "stn_math.h"

#define RT 16            // output tile side
#define RCH 256          // objects culled per pass (= threads per block)

// 16-bit sprites are FP16 (grey, alpha) pairs: sigmoid outputs in (0, 1), 11 significant bits
typedef _Float16 sprite_h2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ float2 sprite_unpack(unsigned u) {
    const sprite_h2 h = __builtin_bit_cast(sprite_h2, u);
    return make_float2((float)h.x, (float)h.y);
}
// (grey, alpha) of texel idx of a sprite row: fp32 pairs, or fp16 pairs in the bf16 training step (the decoder GEMM writes them so)
template <bool S16>
__device__ __forceinline__ float2 ld_texel(const float* __restrict__ S, size_t idx) {
    if constexpr (S16) {
        return sprite_unpack(reinterpret_cast<const unsigned*>(S)[idx]);
    } else {
        return reinterpret_cast<const float2*>(S)[idx];
    }
}

struct Cand {
    float ax, bx, ay, by, pres, depth;
    int row;
};

// Source coordinate from the base (normalised) output coordinate -- the reference's own sequence (affine_grid, then
// grid_sample's unnormalise).  Forward and backward MUST round identically: the compositing adjoint contains
// (a*g - pre), which cancels to rounding level where one object dominates a pixel.
__device__ __forceinline__ float src_from_base(float a, float b, float base, int nsrc, int ac, float& g) {
    g = fmaf(a, base, b);       // explicit: the forward and the backward kernels must agree on this coordinate to the bit (see stn_math.h)
    return ac ? (g + 1.f) * 0.5f * (float)(nsrc - 1) : fmaf(g + 1.f, (float)nsrc, -1.f) * 0.5f;
}
__device__ __forceinline__ float src_of(float a, float b, int j, int nout, int nsrc, int ac) {
    float g;
    return src_from_base(a, b, stn_base(j, nout, ac), nsrc, ac, g);
}
// src_of() is affine in the output index: src_of(j) = c0 + j*slope.  The slope is formed analytically
// (a difference of two src_of values would cancel ~5 digits).
__device__ __forceinline__ void src_affine(float a, float b, int nout, int nsrc, int ac, float& c0, float& slope) {
    const float bstep = ac ? (nout > 1 ? 2.f / (float)(nout - 1) : 0.f) : 2.f / (float)nout;
    const float cm = ac ? 0.5f * (float)(nsrc - 1) : 0.5f * (float)nsrc;
    slope = a * bstep * cm;
    c0 = src_of(a, b, 0, nout, nsrc, ac);
}

// stn_base() for the renderer kernels: with an image side that is a power of two (IP2) the division by n is a multiplication by the
// exactly representable 1/n -- bit-identical to stn_base(), ~10 instructions cheaper per call.
template <int AC, int IP2>
__device__ __forceinline__ float rf_base(int j, int n, float inv_n) {
    if constexpr (IP2 && !AC) return (2.f * (float)j + 1.f) * inv_n - 1.f;
    else return stn_base(j, n, AC);
}

// first / last index in [0, I-1] whose source coordinate lies in (-1, P); exact w.r.t. the forward's own coordinate formula
template <int AC, int IP2>
__device__ __forceinline__ void rb2_range(float a, float b, float c0, float inv, int I, float inv_I, int P, int& lo, int& hi) {
    float g;
    // src(j) ~ c0 + j / inv: the boundary index is within one of the estimate, two exact tests settle it
    lo = max((int)floorf((-1.f - c0) * inv), 0);
#pragma unroll
    for (int t = 0; t < 2; ++t)
        if (lo < I && src_from_base(a, b, rf_base<AC, IP2>(lo, I, inv_I), P, AC, g) <= -1.f) ++lo;
    hi = min((int)ceilf(((float)P - c0) * inv), I - 1);
#pragma unroll
    for (int t = 0; t < 2; ++t)
        if (hi >= 0 && src_from_base(a, b, rf_base<AC, IP2>(hi, I, inv_I), P, AC, g) >= (float)P) --hi;
}

