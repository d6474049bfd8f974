// Shared pieces of the matrix-core renderer kernels (render3.hip: records + forward; tools/exp/render3b.hip: the all-matrix-core backward that was measured and not shipped).
#pragma once
#include "render_common.h"

typedef _Float16 r3_h8 __attribute__((ext_vector_type(8)));
typedef _Float16 r3_h2 __attribute__((ext_vector_type(2)));
typedef float r3_f2 __attribute__((ext_vector_type(2)));
typedef unsigned short r3_u2 __attribute__((ext_vector_type(2)));
typedef short r3_i2 __attribute__((ext_vector_type(2)));

#define R3_P 28                     // sprite side this kernel is built for (row = 112 B = 7 x 16 B)
#define R3_ROWB (R3_P * 4)
#define R3_SPRB (R3_P * R3_P * 4)
#define R3_MAXHW 1024               // objects per sample (list capacity: HW / 4 entries per culling wave)
#define R3_EMPTY 0x7fffu            // first index of an empty footprint
// Hat weights: w = max(0, 1 - |s - u|) is formed in fp32 (s - u and, for the far tap, 1 - |s - u| are exact there) and rounded to fp16
// ONCE, to nearest even.  The small weight of a tap pair therefore keeps 11 significant bits however small it is -- at the rim of a
// footprint the reconstruction is proportional to that weight and the loss gradient to its reciprocal, so an ABSOLUTE rounding of the
// weight (a coordinate grid) would put O(1) relative errors on exactly the pixels with the largest gradients -- and the pair sums to
// 1 +- 2^-12.  The forward and the backward kernel build the fragments with the same instructions: what the backward differentiates
// is bit for bit what the forward composited.
// Per object, sample-major [b][k]: B * HW object records, then B * HW cull records, then B * HW backward records.
// Source coordinate of output index j on either axis: s = A * base(j) + Bc with A = a * P / 2, Bc = (b + 1) * P / 2 - 1 / 2 (the reference's
// affine_grid + unnormalise sequence (g + 1) * P / 2 - 1 / 2, g = a * base + b, re-associated: <= 1e-5 texel from that sequence).
struct __attribute__((aligned(16))) RenderObjRec {
    float Ax, Bx, Ay, By;           // s = fma(A, base, B)
    float pres, mscale;             // importance = mscale * max(alpha, mfloor)
    unsigned mfloor;                // fp16 pair
    float depth;
};
struct __attribute__((aligned(16))) RenderCullRec {
    float Ay, By;                   // as in the object record: the sprite-row window of a tile comes from floor(s)
    unsigned xr, yr;                // pixel footprint: first | last << 16 (first = R3_EMPTY: nothing to draw)
};
// the raw inverse-affine parameters (a_x, b_x, a_y, b_y), as the backward kernel forms them itself when it has no records
struct __attribute__((aligned(16))) RenderBwdRec { float ax, bx, ay, by; };
#define R3_REC_BYTES 64

// 8 hat weights max(0, 1 - |s - u_j|) as an fp16 MFMA fragment; c[jp] = -(u_2jp, u_2jp+1) (a slot that must stay empty carries a large
// value).  Four instructions per pair: v_pk_add_f32, two v_sub_f32 1 - |d| with the clamp modifier, v_cvt_pk_f16_f32 (nearest even).
__device__ __forceinline__ float r3_hat1(float d) { return __builtin_fminf(__builtin_fmaxf(1.f - __builtin_fabsf(d), 0.f), 1.f); }
__device__ __forceinline__ r3_h8 r3_hat8(float s, const r3_f2 (&c)[4]) {
    const r3_f2 s2 = {s, s};
    r3_h2 w[4];
#pragma unroll
    for (int jp = 0; jp < 4; ++jp) {
        const r3_f2 d = s2 + c[jp];
        w[jp] = r3_h2{(_Float16)r3_hat1(d.x), (_Float16)r3_hat1(d.y)};
    }
    return r3_h8{w[0].x, w[0].y, w[1].x, w[1].y, w[2].x, w[2].y, w[3].x, w[3].y};
}

// The same fragment together with the derivative of each weight with respect to the coordinate: -sign(s - u) where the weight is not
// zero, 0 elsewhere -- i.e. -1 on the lower tap, +1 on the upper one (at s - u = 0 exactly, a 2^-24 event, the upper tap's +1 is lost).
// The indicator of a non-zero weight is min(w * 2^14, 1): 1 for every normal fp16 weight (a subnormal one, < 6.1e-5, scales its
// derivative down instead of switching it off).
__device__ __forceinline__ void r3_hat8d(float s, const r3_f2 (&c)[4], r3_h8& wv, r3_h8& dv) {
    const r3_h2 one = {(_Float16)1.f, (_Float16)1.f};
    const r3_h2 big = {(_Float16)16384.f, (_Float16)16384.f};
    const r3_f2 s2 = {s, s};
    unsigned w[4], dd[4];
#pragma unroll
    for (int jp = 0; jp < 4; ++jp) {
        const r3_f2 d = s2 + c[jp];
        const r3_h2 wh = {(_Float16)r3_hat1(d.x), (_Float16)r3_hat1(d.y)};
        w[jp] = __builtin_bit_cast(unsigned, wh);
        // the two sign bits (the upper halves of the two floats), turned into -sign(d) as +-1.0: (~sign) | 0x3c00
        const unsigned sg2 = __builtin_amdgcn_perm(__float_as_uint(d.y), __float_as_uint(d.x), 0x07060302u);
        const r3_h2 sg = __builtin_bit_cast(r3_h2, (~sg2 & 0x80008000u) | 0x3c003c00u);
        dd[jp] = __builtin_bit_cast(unsigned, sg * __builtin_elementwise_min(wh * big, one));
    }
    wv = __builtin_bit_cast(r3_h8, u32x4_t{w[0], w[1], w[2], w[3]});
    dv = __builtin_bit_cast(r3_h8, u32x4_t{dd[0], dd[1], dd[2], dd[3]});
}

struct R3Frag { u32x4_t lo, hi; };      // 8 texels (grey, alpha) of one sprite row

// fl: the importance floor as an fp16 pair -- 0 for a sprite row beyond the sprite, whose texels read as zeros: the PADDING's importance
// is 0, not the floor (the reference clamps the importance sprite, then grid_sample pads it with zeros)
__device__ __forceinline__ void r3_split(const R3Frag& f, unsigned fl, r3_h8& g, r3_h8& a, r3_h8& m) {
    const unsigned d[8] = {f.lo.x, f.lo.y, f.lo.z, f.lo.w, f.hi.x, f.hi.y, f.hi.z, f.hi.w};
    const r3_u2 floor2 = __builtin_bit_cast(r3_u2, fl);
    unsigned gg[4], aa[4], mm[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        gg[i] = __builtin_amdgcn_perm(d[2 * i + 1], d[2 * i], 0x05040100u);
        aa[i] = __builtin_amdgcn_perm(d[2 * i + 1], d[2 * i], 0x07060302u);
        // importance / mscale, models.py:497-499: max(alpha, floor) on the BIT patterns (both >= 0: fp16 order = unsigned order; the
        // float form costs a second v_pk_max_f16 per pair, the compiler's canonicalisation of a loaded value)
        mm[i] = __builtin_bit_cast(unsigned, __builtin_elementwise_max(__builtin_bit_cast(r3_u2, aa[i]), floor2));
    }
    g = __builtin_bit_cast(r3_h8, u32x4_t{gg[0], gg[1], gg[2], gg[3]});
    a = __builtin_bit_cast(r3_h8, u32x4_t{aa[0], aa[1], aa[2], aa[3]});
    m = __builtin_bit_cast(r3_h8, u32x4_t{mm[0], mm[1], mm[2], mm[3]});
}

__device__ __forceinline__ unsigned r3_pk(float a, float b) {
    const r3_h2 h = {(_Float16)a, (_Float16)b};            // v_cvt_pk_f16_f32: round to nearest even
    return __builtin_bit_cast(unsigned, h);
}

// 1 / d: v_rcp_f32 + one Newton step (<= 1 ulp for normal d; the IEEE division sequence is ~10 instructions)
__device__ __forceinline__ float r3_rcp(float d) {
    const float r = __builtin_amdgcn_rcpf(d);
    return fmaf(fmaf(-d, r, 1.f), r, r);
}

// host side (render3.hip)
int render_prep_supported(int HW, int I, int P, int ac);
const void* render_rec_cull(const void* rec, int B, int HW);
const void* render_rec_bwd(const void* rec, int B, int HW);
