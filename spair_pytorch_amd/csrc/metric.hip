// Evaluation metrics on the device (reference spair/metric.py:5-99): COCO-style mean average
// precision of the predicted boxes against the ground-truth boxes, object-count accuracy, batched Jaccard overlap.
// The reference mutates its arguments in place and reads the batch size from the config; these kernels do neither.  The reference's
// box convention is kept as it is (z_where channels taken as top-left x, y and width, height in image fractions, metric.py:15-21).
#include <hip/hip_runtime.h>

#include "common.h"

namespace {

__device__ __forceinline__ float iou_corner(float ax1, float ay1, float ax2, float ay2, float bx1, float by1, float bx2, float by2) {
    const float iw = fmaxf(fminf(ax2, bx2) - fmaxf(ax1, bx1), 0.f), ih = fmaxf(fminf(ay2, by2) - fmaxf(ay1, by1), 0.f);   // metric.py:72-77
    const float inter = iw * ih;
    const float area_a = (ax2 - ax1) * (ay2 - ay1), area_b = (bx2 - bx1) * (by2 - by1);                                     // :93-97
    return inter / (area_a + area_b - inter);
}

// one workgroup per sample: per_sample[b] = (sum_j AP_j / count_b, count_b - #round(z_pres))
__global__ __launch_bounds__(256) void k_metrics(const float* __restrict__ z_where, const float* __restrict__ z_pres,
                                                 const float* __restrict__ bbox, const float* __restrict__ count, int HW, float I, int K,
                                                 float* __restrict__ per_sample) {
    __shared__ float red[4];
    __shared__ float best[64];
    const int b = blockIdx.x, tid = threadIdx.x;
    const float* zw = z_where + (size_t)b * 4 * HW;
    // object count: torch.round = round half to even
    float np_ = 0.f;
    for (int k = tid; k < HW; k += 256) np_ += rintf(z_pres[(size_t)b * HW + k]);
    np_ = block_reduce_sum_256(np_, red);
    float ap_sum = 0.f;
    for (int j0 = 0; j0 < K; j0 += 64) {
        const int nj = min(64, K - j0);
        for (int j = 0; j < nj; ++j) {
            const float* g = bbox + ((size_t)b * K + j0 + j) * 4;
            const float gx1 = g[0], gy1 = g[1], gx2 = g[2] + g[0], gy2 = g[3] + g[1];          // metric.py:22
            float m = -INFINITY;
            for (int k = tid; k < HW; k += 256) {
                const float x1 = zw[k] * I, y1 = zw[HW + k] * I, x2 = zw[2 * HW + k] * I + x1, y2 = zw[3 * HW + k] * I + y1;   // :15-21
                const float v = iou_corner(x1, y1, x2, y2, gx1, gy1, gx2, gy2);
                m = (v > m || v != v) ? v : m;                                                   // torch.max propagates NaN
            }
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) { const float t = __shfl_xor(m, o, 64); m = (t > m || t != t) ? t : m; }
            __syncthreads();
            if ((tid & 63) == 0) red[tid >> 6] = m;
            __syncthreads();
            if (tid == 0) {
                float q = red[0];
                for (int w = 1; w < 4; ++w) q = (red[w] > q || red[w] != red[w]) ? red[w] : q;
                best[j] = q;
            }
        }
        __syncthreads();
        if (tid == 0) {
            for (int j = 0; j < nj; ++j) {
                float ap = 0.f;
                for (int i = 0; i < 9; ++i) {                                                   // AP @ [0.1:0.1:0.9], metric.py:40-41
                    const float s = (float)(0.1 + 0.1 * (double)i);                               // torch.arange(0.1, 1.0, 0.1), fp32
                    ap += fminf(fmaxf((best[j] - s) / (1.f - s), 0.f), 1.f);
                }
                ap_sum += ap / 9.f;
            }
        }
        __syncthreads();
    }
    if (tid == 0) {
        // metric.py:45 divides by the object count; an EMPTY scene (count 0: 0/0 = NaN in the reference, whose dataset has none) is
        // marked with -1 here and left out of the batch mean instead of poisoning it
        per_sample[2 * b] = count[b] > 0.f ? ap_sum / count[b] : -1.f;
        per_sample[2 * b + 1] = count[b] - np_;                                                  // metric.py:55
    }
}

__global__ void k_metrics_final(const float* __restrict__ per_sample, int B, float* __restrict__ out) {
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    float a = 0.f, c = 0.f;
    int nv = 0;
    for (int b = 0; b < B; ++b) {                                                              // fixed order: deterministic
        if (per_sample[2 * b] >= 0.f) { a += per_sample[2 * b]; ++nv; }
        c += per_sample[2 * b + 1];
    }
    // == a / B when no scene is empty; a batch of ONLY empty scenes has no mAP: NaN, as the reference's 0/0 gives (a caller can
    // tell "no data" from an mAP of 0)
    out[0] = nv > 0 ? a / (float)nv : __builtin_nanf("");
    out[1] = c / (float)B;
}

// iou[b][i][j] of corner-format boxes a[b][i], bx[b][j]
__global__ __launch_bounds__(256) void k_batch_jaccard(const float* __restrict__ a, const float* __restrict__ bx, int A, int Bn, long long total,
                                                       float* __restrict__ iou) {
    const long long idx = (long long)blockIdx.x * 256 + threadIdx.x;
    if (idx >= total) return;
    const int j = (int)(idx % Bn);
    const long long t = idx / Bn;
    const int i = (int)(t % A);
    const long long b = t / A;
    const float* pa = a + (b * A + i) * 4;
    const float* pb = bx + (b * Bn + j) * 4;
    iou[idx] = iou_corner(pa[0], pa[1], pa[2], pa[3], pb[0], pb[1], pb[2], pb[3]);
}

}  // namespace

// z_where [B,4,G,G] (x, y, w, h as image fractions), z_pres [B,1,G,G], bbox [B,K,4] (x, y, w, h px, zero padded), count [B] (float);
// scratch: 2*B floats; out[0] = mAP, out[1] = object-count accuracy (mean of count - predicted count, as the reference defines it)
extern "C" int spair_metrics(const float* z_where, const float* z_pres, const float* bbox, const float* count, int B, int G, int image_side,
                             int K, float* scratch, float* out, void* stream) {
    if (B <= 0 || G <= 0 || K <= 0 || !z_where || !z_pres || !bbox || !count || !scratch || !out) return SPAIR_ERR_SHAPE;
    hipStream_t s = (hipStream_t)stream;
    hipLaunchKernelGGL(k_metrics, dim3(B), dim3(256), 0, s, z_where, z_pres, bbox, count, G * G, (float)image_side, K, scratch);
    SPAIR_CHECK_LAUNCH();
    hipLaunchKernelGGL(k_metrics_final, dim3(1), dim3(64), 0, s, scratch, B, out);
    SPAIR_CHECK_LAUNCH();
    return SPAIR_OK;
}

extern "C" int spair_batch_jaccard(const float* box_a, const float* box_b, int B, int A, int Bn, float* iou, void* stream) {
    if (B <= 0 || A <= 0 || Bn <= 0) return SPAIR_ERR_SHAPE;
    const long long total = (long long)B * A * Bn;
    hipLaunchKernelGGL(k_batch_jaccard, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, box_a, box_b, A, Bn, total, iou);
    SPAIR_CHECK_LAUNCH();
    return SPAIR_OK;
}
