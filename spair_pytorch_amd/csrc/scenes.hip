// Synthetic scattered-digit scenes generated ON THE DEVICE: the reference reads scattered-MNIST scenes
// from an HDF5 file (spair/dataloader.py:10-36, item = (image [1,I,I], bbox [K,4] = (x, y, w, h) px zero padded, digit_count)); that
// file and MNIST are not available here, and a host-side generator cannot feed a 6 ms training step.  Scenes are a pure function of
// (seed, sample index): every random number is Philox4x32-10 keyed by the seed with counter (sample, object, draw), so any batch of any
// epoch can be regenerated anywhere -- oracle/scenes_oracle.py restates this file in numpy.
//
// A scene has k in [0, K] glyphs; a glyph is a size x size patch (size in [smin, smax]) placed uniformly inside the image, made of 2-3
// anti-aliased strokes (ring sectors and bars), composited with max.  Only +, -, *, /, sqrt, min/max on fp32 are used.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "common.h"

namespace {

__device__ __host__ inline void philox(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3, uint32_t k0, uint32_t k1, uint32_t out[4]) {
    for (int i = 0; i < 10; ++i) {
        const uint64_t p0 = (uint64_t)0xD2511F53u * c0, p1 = (uint64_t)0xCD9E8D57u * c2;
        const uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0, n1 = (uint32_t)p1;
        const uint32_t n2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1, n3 = (uint32_t)p0;
        c0 = n0; c1 = n1; c2 = n2; c3 = n3;
        k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
    }
    out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
}
__device__ __forceinline__ float uni(uint32_t x) { return (float)(x >> 8) * (1.0f / 16777216.0f); }   // [0,1), exact in fp32

constexpr int SC_MAXOBJ = 32;
constexpr int SC_STROKE = 8;                 // floats per stroke: kind, cy, cx, p0, ax, ay, p1, -
constexpr int SC_OBJ = 4 + 3 * SC_STROKE;    // y0, x0, size, nstrokes | 3 strokes

// thread per (sample, object slot): object geometry and stroke parameters
__global__ __launch_bounds__(64) void k_scene_params(uint64_t seed, long long first, int B, int I, int K, int smin, int smax, float* __restrict__ par,
                                                     float* __restrict__ bbox, long long* __restrict__ count) {
    const int b = blockIdx.x, j = threadIdx.x;
    if (j >= K) return;
    const uint64_t gi = (uint64_t)(first + b);
    const uint32_t k0 = (uint32_t)seed, k1 = (uint32_t)(seed >> 32);
    uint32_t r[4];
    philox((uint32_t)gi, (uint32_t)(gi >> 32), 0u, 0u, k0, k1, r);
    const int k = (int)(r[0] % (uint32_t)(K + 1));
    if (j == 0) count[b] = k;
    float* o = par + ((size_t)b * K + j) * SC_OBJ;
    float* bb = bbox + ((size_t)b * K + j) * 4;
    if (j >= k) {
        o[3] = 0.f;
        bb[0] = bb[1] = bb[2] = bb[3] = 0.f;
        return;
    }
    philox((uint32_t)gi, (uint32_t)(gi >> 32), 1u + (uint32_t)j, 0u, k0, k1, r);
    const int size = min(smin + (int)(r[0] % (uint32_t)(smax - smin + 1)), I);
    const int y0 = (int)(r[1] % (uint32_t)(I - size + 1)), x0 = (int)(r[2] % (uint32_t)(I - size + 1));
    const int ns = 2 + (int)(r[3] & 1u);
    o[0] = (float)y0; o[1] = (float)x0; o[2] = (float)size; o[3] = (float)ns;
    bb[0] = (float)x0; bb[1] = (float)y0; bb[2] = (float)size; bb[3] = (float)size;       // (x, y, w, h) as in the reference's file
    const float n = (float)size, c = (n - 1.f) * 0.5f;
    for (int s = 0; s < 3; ++s) {
        uint32_t a[4], q[4];
        philox((uint32_t)gi, (uint32_t)(gi >> 32), 1u + (uint32_t)j, 1u + 2u * (uint32_t)s, k0, k1, a);
        philox((uint32_t)gi, (uint32_t)(gi >> 32), 1u + (uint32_t)j, 2u + 2u * (uint32_t)s, k0, k1, q);
        float* st = o + 4 + s * SC_STROKE;
        const bool arc = uni(a[0]) < 0.45f;
        float vx = 2.f * uni(q[0]) - 1.f, vy = 2.f * uni(q[1]) - 1.f;                      // direction: normalised random vector
        const float vn = sqrtf(vx * vx + vy * vy);
        if (vn < 1e-3f) { vx = 1.f; vy = 0.f; } else { vx = vx / vn; vy = vy / vn; }
        st[0] = arc ? 1.f : 0.f;
        st[4] = vx; st[5] = vy; st[7] = 0.f;
        if (arc) {      // ring of radius p0 around (cy, cx), kept where the direction from the centre has cosine >= p1 with (ax, ay)
            st[1] = c + (0.3f * uni(a[1]) - 0.15f) * n;
            st[2] = c + (0.3f * uni(a[2]) - 0.15f) * n;
            st[3] = (0.2f + 0.22f * uni(a[3])) * n;
            st[6] = 1.3f * uni(q[2]) - 1.f;
        } else {        // bar through (cy, cx) along (ax, ay), half length p0
            st[1] = c + (0.4f * uni(a[1]) - 0.2f) * n;
            st[2] = c + (0.4f * uni(a[2]) - 0.2f) * n;
            st[3] = (0.25f + 0.2f * uni(a[3])) * n;
            st[6] = 0.f;
        }
    }
}

__device__ __forceinline__ float stroke_value(const float* st, float y, float x) {
    const float dy = y - st[1], dx = x - st[2];
    if (st[0] != 0.f) {
        const float r = sqrtf(dy * dy + dx * dx);
        const float v = fminf(fmaxf(1.4f - fabsf(r - st[3]) / 1.2f, 0.f), 1.f);
        const float cosang = dx * st[4] + dy * st[5];                                      // r * cos(angle to the arc's axis)
        return cosang >= st[6] * r ? v : 0.f;
    }
    const float across = fabsf(dy * st[4] - dx * st[5]), along = fabsf(dy * st[5] + dx * st[4]);
    const float v = fminf(fmaxf(1.4f - across / 1.2f, 0.f), 1.f);
    return along < st[3] ? v : 0.f;
}

// thread per pixel
__global__ __launch_bounds__(256) void k_scene_render(const float* __restrict__ par, const long long* __restrict__ count, int B, int I, int K,
                                                      float* __restrict__ image) {
    const long long idx = (long long)blockIdx.x * 256 + threadIdx.x;
    if (idx >= (long long)B * I * I) return;
    const int x = (int)(idx % I), y = (int)((idx / I) % I), b = (int)(idx / ((long long)I * I));
    const int k = (int)count[b];
    float v = 0.f;
    for (int j = 0; j < k; ++j) {
        const float* o = par + ((size_t)b * K + j) * SC_OBJ;
        const int y0 = (int)o[0], x0 = (int)o[1], size = (int)o[2], ns = (int)o[3];
        if (y < y0 || y >= y0 + size || x < x0 || x >= x0 + size) continue;
        const float ly = (float)(y - y0), lx = (float)(x - x0);
        float g = 0.f;
        for (int s = 0; s < ns; ++s) g = fmaxf(g, stroke_value(o + 4 + s * SC_STROKE, ly, lx));
        v = fmaxf(v, g);
    }
    image[idx] = v;
}

}  // namespace

// image [B,1,I,I] fp32, bbox [B,K,4] fp32 (x, y, w, h), count [B] int64; scratch: B*K*28 floats.  Samples first .. first+B-1 of the
// stream defined by `seed`.
extern "C" int spair_scenes_generate(uint64_t seed, long long first, int B, int I, int K, int size_min, int size_max, float* image, float* bbox,
                                     long long* count, float* scratch, void* stream) {
    if (B <= 0 || I <= 0 || K <= 0 || K > SC_MAXOBJ || size_min < 4 || size_max < size_min || !image || !bbox || !count || !scratch) return SPAIR_ERR_SHAPE;
    hipStream_t s = (hipStream_t)stream;
    hipLaunchKernelGGL(k_scene_params, dim3(B), dim3(64), 0, s, seed, first, B, I, K, size_min, size_max, scratch, bbox, count);
    SPAIR_CHECK_LAUNCH();
    const long long total = (long long)B * I * I;
    hipLaunchKernelGGL(k_scene_render, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, scratch, count, B, I, K, image);
    SPAIR_CHECK_LAUNCH();
    return SPAIR_OK;
}
