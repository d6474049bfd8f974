// Direct fp32 convolutions of the convolutional object encoder / decoder variant (objconv.hip): tiny per-object images (28 x 28 down to
// 2 x 2, <= 32 channels), so there is no GEMM to tile -- one thread per output element, the layer's weights in LDS.
#pragma once
#include "common.h"

// element (r, y, x, c) of a per-object tensor: p[r * rs + y * ys + x * xs + c * cs]; H x H pixels, C channels
struct OcTensor {
    float* p;
    long long rs;
    int ys, xs, cs;
    int H, C;
};
static inline OcTensor oc_hwc(float* p, long long rs, int H, int C) { return OcTensor{p, rs, H * C, C, 1, H, C}; }
static inline OcTensor oc_chw(float* p, long long rs, int H, int C) { return OcTensor{p, rs, H, 1, H * H, H, C}; }
static inline OcTensor oc_rows(const OcTensor& t, long long r0) { OcTensor o = t; o.p = t.p ? t.p + r0 * t.rs : nullptr; return o; }

// W is [X][Y][k][k] in both calls (Conv2d: X = out channels, Y = in channels; ConvTranspose2d: X = in channels, Y = out channels).
//   strided gather:    out(r, y, x, X) = bias + sum_{ky,kx,Y} in(r, y*s + ky, x*s + kx, Y) * W      Conv2d forward, ConvTranspose2d data gradient
//   transposed gather: out(r, y, x, Y) = bias + sum_{ky,kx,X} in(r, (y-ky)/s, (x-kx)/s, X) * W      ConvTranspose2d forward, Conv2d data gradient
// then out = 0 where gate <= 0 (gate: a tensor of out's shape or p == nullptr), then relu if asked.
int oc_gather(bool transposed, const OcTensor& in, const float* W, const float* bias, const OcTensor& out, const OcTensor& gate, int k, int s,
              int relu, long long R, hipStream_t st);
// G[cs][cb][ky][kx] += sum_{r,y,x} small(r, y, x, cs) * big(r, y*s + ky, x*s + kx, cb); bias_small[cs] += sum small (nullptr: skipped).
// Conv2d: small = d out, big = in; ConvTranspose2d: small = in, big = d out.  fp32 atomics (as the fp32 step's other weight gradients).
int oc_wgrad(const OcTensor& small, const OcTensor& big, float* G, float* bias_small, int k, int s, long long R, hipStream_t st);
