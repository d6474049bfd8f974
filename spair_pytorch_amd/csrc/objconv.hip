// Convolutional object encoder / decoder variant (reference sketch: models.py:606-665, config.py:15-20): the
// per-object convolutions of the fp32 per-wavefront step.  Images are 28 x 28 down to 2 x 2 with <= 32 channels: a layer is at most
// 9216 weights and 288 multiply-adds per output element, so the kernels are direct -- one thread per output element with the output
// channel fastest (the 32 lanes of a pixel read the same input element: one broadcast load; their weights are consecutive LDS words),
// the layer's weights staged in LDS once per workgroup, everything in fp32.
#include "objconv.h"

#define OC_MAX_W 16384          // weights of one layer that fit the LDS copy (64 KB)

// CB: output channels per thread (4 when out.C % 4 == 0: one input load and one 16-byte LDS read feed four multiply-adds; 1 otherwise)
template <bool T, int CB>
__global__ __launch_bounds__(256) void k_oc_gather(const OcTensor in, const float* __restrict__ W, const float* __restrict__ bias,
                                                    const OcTensor out, const OcTensor gate, int k, int s, int relu, long long total) {
    extern __shared__ __attribute__((aligned(16))) float wl[];               // [tap][reduced channel][output channel]
    const int Co = out.C, Cr = in.C, kk = k * k;
    for (int i = threadIdx.x; i < Co * Cr * kk; i += 256) {
        const int co = i % Co, t2 = i / Co, cr = t2 % Cr, tap = t2 / Cr;
        wl[i] = W[T ? (cr * Co + co) * kk + tap : (co * Cr + cr) * kk + tap];
    }
    __syncthreads();
    const unsigned idx = blockIdx.x * 256u + threadIdx.x;          // (the launcher keeps total < 2^31)
    if (idx >= total) return;
    const unsigned Cg = (unsigned)Co / CB;
    const int co = (int)(idx % Cg) * CB;
    unsigned rest = idx / Cg;
    const int ox = (int)(rest % (unsigned)out.H);
    rest /= (unsigned)out.H;
    const int oy = (int)(rest % (unsigned)out.H);
    const long long r = rest / (unsigned)out.H;
    float acc[CB];
#pragma unroll
    for (int c = 0; c < CB; ++c) acc[c] = bias ? bias[co + c] : 0.f;
    const float* ip = in.p + r * in.rs;
    // the input rows / columns that reach this output: strided gather iy = oy*s + ky; transposed gather iy*s + ky = oy, 0 <= ky < k
    int y0, y1, x0, x1;
    if (T) {
        y0 = oy - k + s > 0 ? (oy - k + s) / s : 0; y1 = min(oy / s, in.H - 1);
        x0 = ox - k + s > 0 ? (ox - k + s) / s : 0; x1 = min(ox / s, in.H - 1);
    } else {
        y0 = oy * s; y1 = min(y0 + k, in.H) - 1;
        x0 = ox * s; x1 = min(x0 + k, in.H) - 1;
    }
    for (int iy = y0; iy <= y1; ++iy) {
        const int ky = T ? oy - iy * s : iy - y0;
        for (int ix = x0; ix <= x1; ++ix) {
            const int kx = T ? ox - ix * s : ix - x0;
            const float* q = ip + iy * in.ys + ix * in.xs;
            const float* w = wl + (ky * k + kx) * Cr * Co + co;
            // eight channels per round, the 16 loads issued before the first multiply-add (one dependent load per multiply-add otherwise)
            int cr = 0;
            for (; cr + 8 <= Cr; cr += 8) {
                float a[8];
                float b[8][CB];
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    a[u] = q[(cr + u) * in.cs];
                    if (CB == 4) {
                        const float4 v = *reinterpret_cast<const float4*>(w + (cr + u) * Co);
                        b[u][0] = v.x; b[u][1 % CB] = v.y; b[u][2 % CB] = v.z; b[u][3 % CB] = v.w;
                    } else b[u][0] = w[(cr + u) * Co];
                }
#pragma unroll
                for (int u = 0; u < 8; ++u)
#pragma unroll
                    for (int c = 0; c < CB; ++c) acc[c] = fmaf(a[u], b[u][c], acc[c]);
            }
            for (; cr < Cr; ++cr) {
                const float a = q[cr * in.cs];
#pragma unroll
                for (int c = 0; c < CB; ++c) acc[c] = fmaf(a, w[cr * Co + c], acc[c]);
            }
        }
    }
    const long long oo = r * out.rs + oy * out.ys + ox * out.xs + co * out.cs;
#pragma unroll
    for (int c = 0; c < CB; ++c) {
        float v = acc[c];
        if (gate.p && !(gate.p[r * gate.rs + oy * gate.ys + ox * gate.xs + (co + c) * gate.cs] > 0.f)) v = 0.f;
        if (relu) v = fmaxf(v, 0.f);
        acc[c] = v;
    }
    if (CB == 4 && out.cs == 1) *reinterpret_cast<float4*>(out.p + oo) = make_float4(acc[0], acc[1 % CB], acc[2 % CB], acc[3 % CB]);
    else
#pragma unroll
        for (int c = 0; c < CB; ++c) out.p[oo + c * out.cs] = acc[c];
}

template <bool T, int CB>
static int oc_gather_launch(const OcTensor& in, const float* W, const float* bias, const OcTensor& out, const OcTensor& gate, int k, int s,
                            int relu, long long total, size_t lds, hipStream_t st) {
    static std::atomic<unsigned long long> done{0};
    if (lds > 48 * 1024) { const int rc = spair_dyn_lds_once((const void*)k_oc_gather<T, CB>, OC_MAX_W * 4, done); if (rc) return rc; }
    hipLaunchKernelGGL((k_oc_gather<T, CB>), dim3((unsigned)((total + 255) / 256)), dim3(256), lds, st, in, W, bias, out, gate, k, s, relu, total);
    SPAIR_CHECK_LAUNCH();
    return SPAIR_OK;
}

int oc_gather(bool transposed, const OcTensor& in, const float* W, const float* bias, const OcTensor& out, const OcTensor& gate, int k, int s,
              int relu, long long R, hipStream_t st) {
    if (!in.p || !out.p || !W || k < 1 || s < 1 || R <= 0 || in.C < 1 || out.C < 1) return SPAIR_ERR_SHAPE;
    const long long nw = (long long)in.C * out.C * k * k;
    if (nw > OC_MAX_W) return SPAIR_ERR_UNSUPPORTED;
    if (gate.p && (gate.H != out.H || gate.C != out.C)) return SPAIR_ERR_SHAPE;
    if (!transposed && (out.H - 1) * s + k > in.H) return SPAIR_ERR_SHAPE;      // every tap of every output lies inside the input
    if (transposed && (in.H - 1) * s + k > out.H) return SPAIR_ERR_SHAPE;       // every input element lands inside the output
    // four channels per thread: 16-byte LDS reads always (Co % 4 == 0), 16-byte stores when the output is channel-contiguous and aligned
    const bool cb4 = out.C % 4 == 0 && (out.cs != 1 || (((out.rs | out.ys | out.xs) & 3) == 0 && (reinterpret_cast<uintptr_t>(out.p) & 15) == 0));
    const long long total = R * out.H * out.H * (cb4 ? out.C / 4 : out.C);
    if (total >= 0x7fffffffll) return SPAIR_ERR_SHAPE;
    const size_t lds = (size_t)nw * sizeof(float);
    if (transposed) return cb4 ? oc_gather_launch<true, 4>(in, W, bias, out, gate, k, s, relu, total, lds, st)
                               : oc_gather_launch<true, 1>(in, W, bias, out, gate, k, s, relu, total, lds, st);
    return cb4 ? oc_gather_launch<false, 4>(in, W, bias, out, gate, k, s, relu, total, lds, st)
               : oc_gather_launch<false, 1>(in, W, bias, out, gate, k, s, relu, total, lds, st);
}

// Weight gradient: a workgroup walks a slice of the (object, y, x) positions of the SMALL tensor; thread t owns the weights t + 256 j
// (weight index = (tap * Cb + cb) * Cs + cs: consecutive threads take consecutive small channels -- one coalesced load of the small
// tensor's pixel -- and a 32-lane group shares its big-tensor element), keeps them in registers and adds them to G once at the end.
template <int NACC, bool SAME>
__global__ __launch_bounds__(256) void k_oc_wgrad(const OcTensor sm, const OcTensor bg, float* __restrict__ G, float* __restrict__ gbias, int k,
                                                   int s, long long items, int per_block) {
    const int Cs = sm.C, Cb = bg.C, kk = k * k, Wn = Cs * Cb * kk;
    int offs[NACC], offb[NACC];
    float acc[NACC];
#pragma unroll
    for (int j = 0; j < NACC; ++j) {
        const int w = threadIdx.x + 256 * j;
        const bool ok = w < Wn;
        const int cs = w % Cs, rest = w / Cs, cb = rest % Cb, tap = rest / Cb;
        offs[j] = ok ? cs * sm.cs : 0;
        offb[j] = ok ? (tap / k) * bg.ys + (tap % k) * bg.xs + cb * bg.cs : 0;
        acc[j] = 0.f;
    }
    float bacc = 0.f;
    const bool bias_lane = gbias && (int)threadIdx.x < Cs;
    const long long i0 = (long long)blockIdx.x * per_block;
    const long long i1 = i0 + per_block < items ? i0 + per_block : items;
    int x = (int)(i0 % sm.H), y = (int)((i0 / sm.H) % sm.H);
    long long r = i0 / sm.H / sm.H;
#pragma unroll 4
    for (long long it = i0; it < i1; ++it, ++x) {
        if (x == sm.H) { x = 0; if (++y == sm.H) { y = 0; ++r; } }
        const float* sp = sm.p + r * sm.rs + y * sm.ys + x * sm.xs;
        const float* bp = bg.p + r * bg.rs + (y * s) * bg.ys + (x * s) * bg.xs;
        if (SAME) {
            const float sv = sp[offs[0]];
#pragma unroll
            for (int j = 0; j < NACC; ++j) acc[j] = fmaf(sv, bp[offb[j]], acc[j]);
        } else {
#pragma unroll
            for (int j = 0; j < NACC; ++j) acc[j] = fmaf(sp[offs[j]], bp[offb[j]], acc[j]);
        }
        if (bias_lane) bacc += sp[threadIdx.x * sm.cs];
    }
#pragma unroll
    for (int j = 0; j < NACC; ++j) {
        const int w = threadIdx.x + 256 * j;
        if (w < Wn) {
            const int cs = w % Cs, rest = w / Cs, cb = rest % Cb, tap = rest / Cb;
            atomicAdd(&G[(cs * Cb + cb) * kk + tap], acc[j]);
        }
    }
    if (bias_lane) atomicAdd(&gbias[threadIdx.x], bacc);
}

int oc_wgrad(const OcTensor& small, const OcTensor& big, float* G, float* bias_small, int k, int s, long long R, hipStream_t st) {
    if (!small.p || !big.p || !G || k < 1 || s < 1 || R <= 0) return SPAIR_ERR_SHAPE;
    if ((small.H - 1) * s + k > big.H) return SPAIR_ERR_SHAPE;
    const int Wn = small.C * big.C * k * k;
    if (Wn > 256 * 36 || small.C > 256) return SPAIR_ERR_UNSUPPORTED;
    const long long items = R * small.H * small.H;
    // the walk is a chain of dependent loads per wave: many resident waves (8 workgroups per CU when the registers allow) hide it;
    // at least 32 positions per workgroup (the final atomics are Wn per workgroup)
    long long blocks = (Wn <= 256 * 12 ? 8ll : 2ll) * spair_num_cus();
    if (items / blocks < 32) blocks = items / 32 > 0 ? items / 32 : 1;
    const int per_block = (int)((items + blocks - 1) / blocks);
    blocks = (items + per_block - 1) / per_block;
    const bool same = 256 % small.C == 0;       // every weight of a thread multiplies the same small-tensor channel
#define OC_WG(NACC, SAME) hipLaunchKernelGGL((k_oc_wgrad<NACC, SAME>), dim3((unsigned)blocks), dim3(256), 0, st, small, big, G, bias_small, k, s, items, per_block)
    if (Wn <= 256 * 4) { if (same) OC_WG(4, true); else OC_WG(4, false); }
    else if (Wn <= 256 * 12) { if (same) OC_WG(12, true); else OC_WG(12, false); }
    else { if (same) OC_WG(36, true); else OC_WG(36, false); }
#undef OC_WG
    SPAIR_CHECK_LAUNCH();
    return SPAIR_OK;
}

// ---- C ABI (unit-level entry points; include/spair_hip.h) -------------------------------------------------------------------------------
// t7 = {rs, ys, xs, cs, H, C} of a tensor (rs as a 64-bit value)
static OcTensor oc_from(const float* p, const long long* t6) {
    return OcTensor{const_cast<float*>(p), t6[0], (int)t6[1], (int)t6[2], (int)t6[3], (int)t6[4], (int)t6[5]};
}
extern "C" int spair_objconv_gather(int transposed, const float* in, const long long* in6, const float* W, const float* bias, float* out,
                                    const long long* out6, const float* gate, int k, int s, int relu, long long R, void* stream) {
    if (!in6 || !out6) return SPAIR_ERR_SHAPE;
    const OcTensor o = oc_from(out, out6);
    OcTensor g = o;
    g.p = const_cast<float*>(gate);
    return oc_gather(transposed != 0, oc_from(in, in6), W, bias, o, g, k, s, relu, R, (hipStream_t)stream);
}
extern "C" int spair_objconv_wgrad(const float* small, const long long* small6, const float* big, const long long* big6, float* G,
                                   float* bias_small, int k, int s, long long R, void* stream) {
    if (!small6 || !big6) return SPAIR_ERR_SHAPE;
    return oc_wgrad(oc_from(small, small6), oc_from(big, big6), G, bias_small, k, s, R, (hipStream_t)stream);
}
