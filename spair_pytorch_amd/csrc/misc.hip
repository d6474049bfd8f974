// Small utility kernels: fused Adam (K9), counter-based noise, input padding, weight preparation,
// per-row -> NCHW map export, and the direct first backbone layer (Cin = image channels).
#include <cstdlib>
#include <cstring>
#include "cells.h"
#include "misc.h"

// ---- K9: torch.optim.Adam defaults (train.py:44) on the flat parameter buffer ------------------
__global__ __launch_bounds__(256) void k_adam(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m,
                                              float* __restrict__ v, long long n, float lr, float b1, float b2, float eps,
                                              float bc1, float sqrt_bc2) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float gi = g[i];
    const float mi = b1 * m[i] + (1.f - b1) * gi;
    const float vi = b2 * v[i] + (1.f - b2) * gi * gi;
    m[i] = mi;
    v[i] = vi;
    const float denom = sqrtf(vi) / sqrt_bc2 + eps;
    p[i] -= (lr / bc1) * (mi / denom);
}

extern "C" int spair_adam(float* params, const float* grads, float* exp_avg, float* exp_avg_sq, int64_t n, float lr,
                          float beta1, float beta2, float eps, int step, void* stream) {
    if (n <= 0 || step < 1) return SPAIR_ERR_SHAPE;
    const float bc1 = 1.f - powf(beta1, (float)step);
    const float sqrt_bc2 = sqrtf(1.f - powf(beta2, (float)step));
    hipLaunchKernelGGL(k_adam, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, params, grads, exp_avg,
                       exp_avg_sq, (long long)n, lr, beta1, beta2, eps, bc1, sqrt_bc2);
    SPAIR_CHECK_LAUNCH();
    return SPAIR_OK;
}

// The guarded form (include/spair_hip.h): a step whose forward flagged a non-finite loss is left out whole, an element with a non-finite
// gradient on its own -- lr * NaN never reaches a parameter (the reference raises before its optimizer step: debug_tools.py:245-271).
__global__ __launch_bounds__(256) void k_adam_guarded(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m,
                                                      float* __restrict__ v, long long n, float lr, float b1, float b2, float eps,
                                                      float bc1, float sqrt_bc2, const int* __restrict__ skip, int* __restrict__ counters) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (skip && *skip) {
        if (i == 0) counters[0] += 1;
        return;
    }
    if (i >= n) return;
    const float gi = g[i];
    if (!(fabsf(gi) <= 3.402823466e38f)) { counters[1] = 1; return; }      // NaN or inf
    const float mi = b1 * m[i] + (1.f - b1) * gi;
    const float vi = b2 * v[i] + (1.f - b2) * gi * gi;
    m[i] = mi;
    v[i] = vi;
    const float denom = sqrtf(vi) / sqrt_bc2 + eps;
    p[i] -= (lr / bc1) * (mi / denom);
}

extern "C" int spair_adam_guarded(float* params, const float* grads, float* exp_avg, float* exp_avg_sq, int64_t n, float lr,
                                  float beta1, float beta2, float eps, int step, const int* skip, int* counters, void* stream) {
    if (n <= 0 || step < 1 || !counters) return SPAIR_ERR_SHAPE;
    const float bc1 = 1.f - powf(beta1, (float)step);
    const float sqrt_bc2 = sqrtf(1.f - powf(beta2, (float)step));
    hipLaunchKernelGGL(k_adam_guarded, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, params, grads, exp_avg,
                       exp_avg_sq, (long long)n, lr, beta1, beta2, eps, bc1, sqrt_bc2, skip, counters);
    SPAIR_CHECK_LAUNCH();
    return SPAIR_OK;
}

extern "C" int spair_host_word_alloc(int** out) {
    if (!out) return SPAIR_ERR_SHAPE;
    void* p = nullptr;
    if (hipHostMalloc(&p, 64, hipHostMallocMapped | hipHostMallocCoherent) != hipSuccess) return SPAIR_ERR_LAUNCH;
    memset(p, 0, 64);
    *out = reinterpret_cast<int*>(p);
    return SPAIR_OK;
}
extern "C" int spair_host_word_free(int* word) {
    if (!word) return SPAIR_OK;
    return hipHostFree(word) == hipSuccess ? SPAIR_OK : SPAIR_ERR_LAUNCH;
}

// ---- noise: Philox4x32-10, Box-Muller; one counter per output element ---------------------------
__device__ __forceinline__ void philox4x32_10(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3, uint32_t k0, uint32_t k1,
                                              uint32_t out[4]) {
#pragma unroll
    for (int i = 0; i < 10; ++i) {
        const uint64_t p0 = (uint64_t)0xD2511F53u * c0, p1 = (uint64_t)0xCD9E8D57u * c2;
        const uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0, n1 = (uint32_t)p1;
        const uint32_t n2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1, n3 = (uint32_t)p0;
        c0 = n0; c1 = n1; c2 = n2; c3 = n3;
        k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
    }
    out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
}
__device__ __forceinline__ float u01(uint32_t x) { return ((float)(x >> 8) + 0.5f) * (1.0f / 16777216.0f); }  // (0,1)

__global__ __launch_bounds__(256) void k_noise(uint64_t seed, float* eps_box, long long n_box, float* eps_attr, long long n_attr,
                                               float* eps_depth, long long n_depth, float* u_pres, long long n_pres) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    const long long total = n_box + n_attr + n_depth + n_pres;
    if (i >= total) return;
    uint32_t o[4];
    philox4x32_10((uint32_t)i, (uint32_t)(i >> 32), 0x5bd1e995u, 0u, (uint32_t)seed, (uint32_t)(seed >> 32), o);
    const float n = sqrtf(-2.f * logf(u01(o[0]))) * cosf(6.28318530718f * u01(o[1]));
    if (i < n_box) eps_box[i] = n;
    else if (i < n_box + n_attr) eps_attr[i - n_box] = n;
    else if (i < n_box + n_attr + n_depth) eps_depth[i - n_box - n_attr] = n;
    else u_pres[i - n_box - n_attr - n_depth] = u01(o[2]);
}

extern "C" int spair_noise_fill(const SpairDims* d, uint64_t seed, float* eps_box, float* eps_attr, float* eps_depth,
                                float* u_pres, void* stream) {
    const long long cells = (long long)d->B * d->G * d->G;
    const long long total = cells * (4 + d->A + 2);
    hipLaunchKernelGGL(k_noise, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, seed, eps_box, cells * 4,
                       eps_attr, cells * d->A, eps_depth, cells, u_pres, cells);
    SPAIR_CHECK_LAUNCH();
    return SPAIR_OK;
}

// ---- input: NCHW [B,C,I,I] -> zero-padded NHWC [B,Ip,Ip,C] (modules.py:105,108) ----------------
__global__ __launch_bounds__(256) void k_pad_input(const float* __restrict__ x, float* __restrict__ xp, int B, int C, int I, int pre,
                                                   int Ip) {
    const long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    const long long total = (long long)B * Ip * Ip * C;
    if (idx >= total) return;
    const int c = (int)(idx % C);
    long long t = idx / C;
    const int px = (int)(t % Ip); t /= Ip;
    const int py = (int)(t % Ip);
    const int b = (int)(t / Ip);
    const int sx = px - pre, sy = py - pre;
    xp[idx] = (sx >= 0 && sx < I && sy >= 0 && sy < I) ? x[(((size_t)b * C + c) * I + sy) * I + sx] : 0.f;
}
int misc_pad_input(const float* x, float* xp, int B, int C, int I, int pre, int Ip, hipStream_t s) {
    const long long total = (long long)B * Ip * Ip * C;
    hipLaunchKernelGGL(k_pad_input, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, x, xp, B, C, I, pre, Ip);
    SPAIR_CHECK_LAUNCH();
    return SPAIR_OK;
}

// ---- weight preparation: dst[r*ld + c] = src[map(r,c)] as fp32 or bf16, zero padded --------------
// mode 0: dst[r][c] = src[r*cols + c]                      (copy / pad / convert)
// mode 1: dst[r][c] = src[c*rows + r]                      (transpose: dst rows = src cols)
// mode 2: conv OIHW -> [O][ky][kx][ci]: src[((r*Cin+ci)*k+ky)*k+kx], c=(ky*k+kx)*Cin+ci
// mode 4: MFMA-fragment packing of a Linear weight src[rows][cols] for the fused chain (forward, B[k][n] = W[n][k]):
//         fragment (nt,kt) = 64 lanes x 8 bf16, lane l holds W[nt*16 + (l&15)][kk], kk = kt*32 + 8*(l>>4) + j; packed k -> source
//         column: kk < kpad0 ? (kk < ksplit ? kk : none) : ksplit + (kk - kpad0); rows are offset by n_off (concatenated heads)
// mode 3: conv dgrad class (py,px), stride s, T=k/s taps: dst[ci][(ty*T+tx)*O + co] = src[((co*Cin+ci)*k + py+s*ty)*k + px+s*tx]
__global__ __launch_bounds__(256) void k_prep(PrepTable T) {
    const PrepEntry e = T.e[blockIdx.y];
    if (e.mode == 4) {
        const int KP = e.KT * 32;
        const long long n4 = (long long)e.rows * KP;
        for (long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x; idx < n4; idx += (long long)gridDim.x * blockDim.x) {
            const int r = (int)(idx / KP), kk = (int)(idx - (long long)r * KP);
            int col = (kk < e.kpad0) ? (kk < e.ksplit ? kk : -1) : e.ksplit + (kk - e.kpad0);
            if (col >= e.cols) col = -1;
            float v = col >= 0 ? e.src[(size_t)r * e.cols + col] : 0.f;
            if (e.bf16 == 2) v -= (float)(__bf16)v;              // the LOW part of a split-bf16 operand (chain.hip, box network)
            const int nn = e.n_off + r, nt = nn >> 4, li = nn & 15, kt = kk >> 5, g = (kk & 31) >> 3, j = kk & 7;
            reinterpret_cast<__bf16*>(e.dst)[(((size_t)nt * e.KT + kt) * 64 + g * 16 + li) * 8 + j] = (__bf16)v;
        }
        return;
    }
    if (e.mode == 5) {   // data-gradient pack: B[k = n_off + o][n = i] = W[o][i]; padding relies on the zero-initialised workspace
        const long long n5 = (long long)e.rows * e.cols;
        for (long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x; idx < n5; idx += (long long)gridDim.x * blockDim.x) {
            const int o = (int)(idx / e.cols), i = (int)(idx - (long long)o * e.cols);
            const int kk = e.n_off + o, nt = i >> 4, li = i & 15, kt = kk >> 5, g = (kk & 31) >> 3, j = kk & 7;
            reinterpret_cast<__bf16*>(e.dst)[(((size_t)nt * e.KT + kt) * 64 + g * 16 + li) * 8 + j] = (__bf16)e.src[idx];
        }
        return;
    }
    const long long n = (long long)e.rows * e.cols;   // pad columns are never written (workspace is zero-initialised)
    for (long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x; idx < n; idx += (long long)gridDim.x * blockDim.x) {
        const int r = (int)(idx / e.cols), c = (int)(idx - (long long)r * e.cols);
        float v;
        if (e.mode == 0) v = e.src[(size_t)r * e.cols + c];
        else if (e.mode == 1) v = e.src[(size_t)c * e.rows + r];
        else if (e.mode == 2) {
            int ky, kx, ci;
            if (e.s > 0) {   // tap-parity K order of gemm16.hip (GemmNT::ktab): 64-column blocks = (parity class, channel block, tap of the class)
                const int blk = c >> 6, TT = e.T * e.T, nh = e.cin >> 6;
                const int tq = blk % TT, rr = blk / TT, h = rr % nh, cls = rr / nh;
                ky = cls / e.s + e.s * (tq / e.T); kx = cls % e.s + e.s * (tq % e.T); ci = h * 64 + (c & 63);
            } else {
                const int tap = c / e.cin;
                ci = c - tap * e.cin; ky = tap / e.k; kx = tap - ky * e.k;
            }
            v = e.src[(((size_t)r * e.cin + ci) * e.k + ky) * e.k + kx];
        } else {
            const int t = c / e.cout, co = c - t * e.cout, ty = t / e.T, tx = t - ty * e.T;
            v = e.src[(((size_t)co * e.cin + r) * e.k + (e.py + e.s * ty)) * e.k + (e.px + e.s * tx)];
        }
        const size_t o = (size_t)r * e.ld + c;
        if (e.bf16) reinterpret_cast<__bf16*>(e.dst)[o] = (__bf16)v;
        else reinterpret_cast<float*>(e.dst)[o] = v;
    }
}
int misc_prep(const PrepTable& T, hipStream_t s) {
    if (T.n <= 0) return SPAIR_OK;
    hipLaunchKernelGGL(k_prep, dim3(64, T.n), dim3(256), 0, s, T);
    SPAIR_CHECK_LAUNCH();
    return SPAIR_OK;
}

// ---- per-row quantity -> NCHW map -----------------------------------------------------------------
__global__ __launch_bounds__(256) void k_export(const float* __restrict__ src, int ld, int col0, int ch, const int* __restrict__ cell_h,
                                                const int* __restrict__ cell_w, int B, int G, float* __restrict__ out) {
    const long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    const long long total = (long long)B * G * G * ch;
    if (idx >= total) return;
    const int c = (int)(idx % ch);
    const long long r = idx / ch;
    const int cp = (int)(r / B), b = (int)(r - (long long)cp * B);
    out[(((size_t)b * ch + c) * G + cell_h[cp]) * G + cell_w[cp]] = src[r * ld + col0 + c];
}
// the same from a row buffer stored as bf16 (the fused chain's gradient rows: same leading dimension in ELEMENTS)
__global__ __launch_bounds__(256) void k_export16(const __bf16* __restrict__ src, int ld, int col0, int ch, const int* __restrict__ cell_h,
                                                  const int* __restrict__ cell_w, int B, int G, float* __restrict__ out) {
    const long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    const long long total = (long long)B * G * G * ch;
    if (idx >= total) return;
    const int c = (int)(idx % ch);
    const long long r = idx / ch;
    const int cp = (int)(r / B), b = (int)(r - (long long)cp * B);
    out[(((size_t)b * ch + c) * G + cell_h[cp]) * G + cell_w[cp]] = (float)src[r * ld + col0 + c];
}
int misc_export16(const void* src, int ld, int col0, int ch, const int* cell_h, const int* cell_w, int B, int G, float* out, hipStream_t s) {
    const long long total = (long long)B * G * G * ch;
    hipLaunchKernelGGL(k_export16, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, reinterpret_cast<const __bf16*>(src), ld, col0, ch,
                       cell_h, cell_w, B, G, out);
    SPAIR_CHECK_LAUNCH();
    return SPAIR_OK;
}
int misc_export(const float* src, int ld, int col0, int ch, const int* cell_h, const int* cell_w, int B, int G, float* out,
                hipStream_t s) {
    const long long total = (long long)B * G * G * ch;
    hipLaunchKernelGGL(k_export, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, src, ld, col0, ch, cell_h, cell_w, B, G, out);
    SPAIR_CHECK_LAUNCH();
    return SPAIR_OK;
}

// ---- backbone layer 0 (Cin = C, tiny K = k*k*C): direct kernels -------------------------------------
// forward: out[m][co] = relu(bias[co] + sum_k w[co][k] * patch(m)[k]); weights in LDS; thread = (pixel, 4 channels)
__global__ __launch_bounds__(256) void k_conv0_fwd(const float* __restrict__ xp, const float* __restrict__ w, const float* __restrict__ bias,
                                                   float* __restrict__ out, int B, int Hin, int C, int k, int s, int Hout, int Cout, int out_bf16) {
    extern __shared__ float wsh[];   // [K0][Cout] transposed for conflict-free float4 reads
    const int K0 = k * k * C;
    for (int i = threadIdx.x; i < K0 * Cout; i += blockDim.x) {
        const int co = i / K0, kk = i - co * K0;          // src OIHW: k index = (ci*k + ky)*k + kx
        const int ci = kk / (k * k), rem = kk - ci * k * k;
        wsh[((rem * C) + ci) * Cout + co] = w[i];            // dst k index = (ky*k+kx)*C + ci
    }
    __syncthreads();
    const int q = Cout / 4;
    const long long total = (long long)B * Hout * Hout * q;
    for (long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long long)gridDim.x * blockDim.x) {
        const int cq = (int)(idx % q);
        const long long m = idx / q;
        const int ox = (int)(m % Hout), oy = (int)((m / Hout) % Hout), b = (int)(m / ((long long)Hout * Hout));
        float4 acc = *reinterpret_cast<const float4*>(bias + cq * 4);
        for (int ky = 0; ky < k; ++ky) {
            const float* row = xp + (((size_t)b * Hin + oy * s + ky) * Hin + ox * s) * C;
            for (int t = 0; t < k * C; ++t) {
                const float xv = row[t];
                const float4 wv = *reinterpret_cast<const float4*>(&wsh[((ky * k * C) + t) * Cout + cq * 4]);
                acc.x += xv * wv.x; acc.y += xv * wv.y; acc.z += xv * wv.z; acc.w += xv * wv.w;
            }
        }
        acc.x = fmaxf(acc.x, 0.f); acc.y = fmaxf(acc.y, 0.f); acc.z = fmaxf(acc.z, 0.f); acc.w = fmaxf(acc.w, 0.f);
        if (out_bf16) {
            bf16x4 o;
            o[0] = (__bf16)acc.x; o[1] = (__bf16)acc.y; o[2] = (__bf16)acc.z; o[3] = (__bf16)acc.w;
            *reinterpret_cast<bf16x4*>(reinterpret_cast<__bf16*>(out) + (size_t)m * Cout + cq * 4) = o;
        } else {
            *reinterpret_cast<float4*>(out + (size_t)m * Cout + cq * 4) = acc;
        }
    }
}
constexpr int C0_ROWS = 5;
// Single-channel 4x4 specialisation (the reference's grey-scale stem): workgroup = C0_ROWS output rows of one sample, their
// input rows staged in LDS, the 16x4 weights of a thread's channel quad held in registers, thread = (pixel group,
// channel quad) so a pixel's 128 channels leave as one contiguous 256 B (bf16) / 512 B (fp32) store.  HBM-write bound.
// The input is read straight from the unpadded image (zero outside [pre, pre + I)): the padded copy the weight gradient
// reads later is made beside this kernel, not before it.
__global__ __launch_bounds__(256) void k_conv0_fwd_c1k4(const float* __restrict__ x, const float* __restrict__ w,
                                                        const float* __restrict__ bias, float* __restrict__ out, int Hin, int s,
                                                        int Hout, int Cout, int out_bf16, int I, int pre) {
    extern __shared__ float xs[];    // [s * (C0_ROWS - 1) + 4][Hin]: the input rows of this workgroup's C0_ROWS output rows
    const int oy0 = blockIdx.x * C0_ROWS, b = blockIdx.y;
    const int nrow = min(C0_ROWS, Hout - oy0), nin = s * (nrow - 1) + 4;
    const int q = Cout >> 2, groups = 256 / q;
    const int cq = threadIdx.x % q, pg = threadIdx.x / q;
    {   // input rows: eight loads per thread in flight at once, through a buffer descriptor (a pixel of the zero padding goes out of
        // range and reads 0; written `in ? x[..] : 0` each load was conditional and sat behind its own vmcnt(0): 7 round trips per WG)
        const __amdgpu_buffer_rsrc_t rsx = buf_rsrc(x + (size_t)b * I * I);
        const int total = nin * Hin;
        for (int i0 = threadIdx.x; i0 < total; i0 += 8 * 256) {
            unsigned v[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const int i = i0 + e * 256;
                const int ky = i / Hin, xx = i - ky * Hin;
                const int sy = oy0 * s + ky - pre, sx = xx - pre;
                const bool in = i < total && (unsigned)sy < (unsigned)I && (unsigned)sx < (unsigned)I;
                v[e] = __builtin_amdgcn_raw_buffer_load_b32(rsx, in ? (sy * I + sx) * 4 : (int)BUF_OOB, 0, 0);
            }
#pragma unroll
            for (int e = 0; e < 8; ++e)
                if (i0 + e * 256 < total) xs[i0 + e * 256] = __uint_as_float(v[e]);
        }
    }
    typedef float f2 __attribute__((ext_vector_type(2)));
    f2 wlo[16], whi[16];             // per tap: channels (0,1) and (2,3) of the quad -- the 64 FMAs of a pixel issue as 32 packed ones
    {
        const float4* w4 = reinterpret_cast<const float4*>(w + (size_t)cq * 4 * 16);     // 4 channels x 16 taps, contiguous
        float4 t[16];
#pragma unroll
        for (int i = 0; i < 16; ++i) t[i] = w4[i];
#pragma unroll
        for (int tq = 0; tq < 4; ++tq) {      // t[c*4 + tq] holds taps 4tq..4tq+3 of channel c
            wlo[tq * 4 + 0] = f2{t[tq].x, t[4 + tq].x}; whi[tq * 4 + 0] = f2{t[8 + tq].x, t[12 + tq].x};
            wlo[tq * 4 + 1] = f2{t[tq].y, t[4 + tq].y}; whi[tq * 4 + 1] = f2{t[8 + tq].y, t[12 + tq].y};
            wlo[tq * 4 + 2] = f2{t[tq].z, t[4 + tq].z}; whi[tq * 4 + 2] = f2{t[8 + tq].z, t[12 + tq].z};
            wlo[tq * 4 + 3] = f2{t[tq].w, t[4 + tq].w}; whi[tq * 4 + 3] = f2{t[8 + tq].w, t[12 + tq].w};
        }
    }
    const float4 bv = *reinterpret_cast<const float4*>(bias + cq * 4);
    __syncthreads();
    if (pg >= groups) return;
    const size_t mrow = ((size_t)b * Hout + oy0) * Hout;
    const int npix = nrow * Hout;
#pragma unroll 2
    for (int p = pg; p < npix; p += groups) {
        const int r = p / Hout, ox = p - r * Hout;
        const float* xr = xs + r * s * Hin + ox * s;
        f2 a0 = f2{bv.x, bv.y}, a1 = f2{bv.z, bv.w};
#pragma unroll
        for (int ky = 0; ky < 4; ++ky)
#pragma unroll
            for (int kx = 0; kx < 4; ++kx) {
                const float xv = xr[ky * Hin + kx];
                const f2 x2 = f2{xv, xv};
                a0 = __builtin_elementwise_fma(x2, wlo[ky * 4 + kx], a0);
                a1 = __builtin_elementwise_fma(x2, whi[ky * 4 + kx], a1);
            }
        float4 acc;
        acc.x = fmaxf(a0.x, 0.f); acc.y = fmaxf(a0.y, 0.f); acc.z = fmaxf(a1.x, 0.f); acc.w = fmaxf(a1.y, 0.f);
        if (out_bf16) {
            bf16x4 o;
            o[0] = (__bf16)acc.x; o[1] = (__bf16)acc.y; o[2] = (__bf16)acc.z; o[3] = (__bf16)acc.w;
            *reinterpret_cast<bf16x4*>(reinterpret_cast<__bf16*>(out) + (mrow + p) * Cout + cq * 4) = o;
        } else {
            *reinterpret_cast<float4*>(out + (mrow + p) * Cout + cq * 4) = acc;
        }
    }
}

// The same stem on the matrix cores (bf16 mode, 128 output channels, stride 2).  As a GEMM the stem is [channels x 16 taps] . [16 taps x
// pixels]: v_mfma_f32_16x16x16_bf16 with M = 16 channels, N = 16 consecutive output pixels of one row, K = the 16 taps (lane group q holds
// the 4 taps of kernel row q: 4 consecutive input pixels).  fp32 operands are split x = xh + xl, w = wh + wl into bf16 pairs and three
// products (xh.wh + xh.wl + xl.wh) accumulate in fp32 -- what is dropped is xl.wl, below 2^-16 of |x||w| -- so the result agrees with the
// fp32 FMA kernel above to well inside the bf16 rounding of the stored activation.  24 MFMAs per 16 pixels x 128 channels replace 2,048
// lane-FMAs per wave-instruction slot.
// Channel order: row 4q + r of channel group g is channel 32q + 4g + r, so after the 8 groups lane (q, pixel) holds the 32 CONSECUTIVE
// channels 32q .. 32q+31 of its pixel.  Stored straight from there every wave-instruction scatters 64 x 16 bytes at a 64-byte pitch
// (measured: 122 us = 2.6 TB/s for the 321 MB of config 2, hardly better than the FMA kernel's 134 us with its 8-byte stores); through a
// per-wave LDS tile (pixel pitch 272 B: conflict-free both ways) each store instruction writes 4 whole pixel rows = 1 KiB contiguous:
// 74 us = 4.4 TB/s, 66 us with non-temporal stores.
typedef short c0_s16x4 __attribute__((ext_vector_type(4)));
constexpr int C0_TP = 272;           // output staging tile: bytes per pixel
__device__ __forceinline__ void c0_split4(const float (&v)[4], c0_s16x4& hi, c0_s16x4& lo) {
    bf16x4 h, l;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        h[e] = (__bf16)v[e];
        l[e] = (__bf16)(v[e] - (float)h[e]);
    }
    hi = __builtin_bit_cast(c0_s16x4, h);
    lo = __builtin_bit_cast(c0_s16x4, l);
}
__host__ __device__ constexpr size_t c0_mfma_xs_bytes(int Hin) { return (((size_t)(2 * (C0_ROWS - 1) + 4) * (Hin + 2) * sizeof(float)) + 15) & ~(size_t)15; }
// `mask` (optional): one byte per (pixel, 8 channels), [B][Hout][Hout][16], bit e = channel 8 g + e of the stored activation > 0 -- the ReLU
// gate conv_1's data gradient needs, 16 x smaller than the activation rows it otherwise re-reads (268 MB at the benchmark shape).
__global__ __launch_bounds__(256) void k_conv0_fwd_c1k4_mfma(const float* __restrict__ x, const float* __restrict__ w,
                                                             const float* __restrict__ bias, __bf16* __restrict__ out, int Hin, int Hout,
                                                             int I, int pre, unsigned char* __restrict__ mask) {
    extern __shared__ float xs[];    // [2 * (C0_ROWS - 1) + 4][Hin + 2]: the input rows of this workgroup's output rows; then 4 per-wave output tiles
    char* tile = reinterpret_cast<char*>(xs) + c0_mfma_xs_bytes(Hin);
    const int oy0 = blockIdx.x * C0_ROWS, b = blockIdx.y;
    const int nrow = min(C0_ROWS, Hout - oy0), nin = 2 * (nrow - 1) + 4;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int LDX = Hin + 2;
    {
        const __amdgpu_buffer_rsrc_t rsx = buf_rsrc(x + (size_t)b * I * I);
        const int total = nin * LDX;
        for (int i0 = tid; i0 < total; i0 += 8 * 256) {
            unsigned v[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const int i = i0 + e * 256;
                const int ky = i / LDX, xx = i - ky * LDX;
                const int sy = oy0 * 2 + ky - pre, sx = xx - pre;
                const bool in = i < total && (unsigned)sy < (unsigned)I && (unsigned)sx < (unsigned)I;
                v[e] = __builtin_amdgcn_raw_buffer_load_b32(rsx, in ? (sy * I + sx) * 4 : (int)BUF_OOB, 0, 0);
            }
#pragma unroll
            for (int e = 0; e < 8; ++e)
                if (i0 + e * 256 < total) xs[i0 + e * 256] = __uint_as_float(v[e]);
        }
    }
    // weights: lane (m = lane & 15, q = lane >> 4) holds taps (ky = q, kx = 0..3) of channel 32*(m >> 2) + 4g + (m & 3), g = 0..7
    const int q = lane >> 4, m = lane & 15;
    c0_s16x4 wh[8], wl[8];
    f32x4 bv[8];                      // accumulator start = bias of the lane's OUTPUT channels 32q + 4g + r
#pragma unroll
    for (int g = 0; g < 8; ++g) {
        const int ch = 32 * (m >> 2) + 4 * g + (m & 3);
        const float4 t = *reinterpret_cast<const float4*>(w + (size_t)ch * 16 + q * 4);
        const float tv[4] = {t.x, t.y, t.z, t.w};
        c0_split4(tv, wh[g], wl[g]);
        const float4 bq = *reinterpret_cast<const float4*>(bias + 32 * q + 4 * g);
        bv[g] = (f32x4){bq.x, bq.y, bq.z, bq.w};
    }
    __syncthreads();
    const int nbx = (Hout + 15) >> 4, nblk = nrow * nbx;
    char* tw = tile + wave * (16 * C0_TP);
    for (int blk = wave; blk < nblk; blk += 4) {
        const int r = blk / nbx, ox0 = (blk - r * nbx) * 16;
        const int oxc = min(ox0 + m, Hout - 1);
        const float* xr = xs + (r * 2 + q) * LDX + oxc * 2;            // 8-byte aligned (LDX even)
        const float2 p0 = *reinterpret_cast<const float2*>(xr), p1 = *reinterpret_cast<const float2*>(xr + 2);
        const float pv[4] = {p0.x, p0.y, p1.x, p1.y};
        c0_s16x4 xh, xl;
        c0_split4(pv, xh, xl);
        f32x4 acc[8];
#pragma unroll
        for (int g = 0; g < 8; ++g) {
            acc[g] = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(wh[g], xh, bv[g], 0, 0, 0);
            acc[g] = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(wl[g], xh, acc[g], 0, 0, 0);
            acc[g] = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(wh[g], xl, acc[g], 0, 0, 0);
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            bf16x8 o;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                o[e] = (__bf16)fmaxf(acc[2 * j][e], 0.f);
                o[4 + e] = (__bf16)fmaxf(acc[2 * j + 1][e], 0.f);
            }
            *reinterpret_cast<bf16x8*>(tw + m * C0_TP + q * 64 + j * 16) = o;
        }
        wave_lds_fence();       // the tile is wave-private, but every lane reads what OTHER lanes wrote
        __bf16* drow = out + (((size_t)b * Hout + oy0 + r) * Hout + ox0) * 128;
        const int npx = min(16, Hout - ox0);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int px = 4 * j + q;
            const bf16x8 o = *reinterpret_cast<const bf16x8*>(tw + px * C0_TP + m * 16);
            // non-temporal: the 321 MB stream past the L2 instead of evicting it (74 -> 66 us alone; in the step conv_1's forward, which
            // reads this tensor next, went 0.231 -> 0.210 ms as well)
            if (px < npx) __builtin_nontemporal_store(o, reinterpret_cast<bf16x8*>(drow + px * 128 + m * 8));
            if (mask) {          // (wave-uniform)
                const u32x4_t wv = __builtin_bit_cast(u32x4_t, o);      // post-ReLU: a channel is on iff its 16 bits are not zero
                auto nz2 = [](unsigned v) { return (unsigned)((int)(v << 16) > 0) | ((unsigned)((int)v >> 16 > 0) << 1); };      // as a signed 16-bit value > 0: the gate `activation > 0` exactly (-0.0 and NaN stay off)
                const unsigned byte = nz2(wv.x) | (nz2(wv.y) << 2) | (nz2(wv.z) << 4) | (nz2(wv.w) << 6);
                if (px < npx) mask[(((size_t)b * Hout + oy0 + r) * Hout + ox0 + px) * 16 + m] = (unsigned char)byte;
            }
        }
        wave_lds_fence();       // ... and the next block's writes stay behind these reads
    }
}

bool misc_conv0_reads_unpadded(int B, int Hin, int C, int k, int Cout) {
    return Cout % 4 == 0 && C == 1 && k == 4 && Cout / 4 <= 256 && 256 % (Cout / 4) == 0 && B <= 65535 &&
           (size_t)(4 * (C0_ROWS - 1) + 4) * Hin * sizeof(float) <= 48 * 1024;     // stride <= 4
}
bool misc_conv0_writes_mask(int B, int Hin, int C, int k, int s, int Cout, int out_bf16) {
    return misc_conv0_reads_unpadded(B, Hin, C, k, Cout) && out_bf16 && Cout == 128 && s == 2 && (Hin & 1) == 0;
}
int misc_conv0_fwd(const float* x, const float* xp, const float* w, const float* bias, float* out, int B, int I, int pre, int Hin, int C, int k,
                   int s, int Hout, int Cout, int out_bf16, hipStream_t st, unsigned char* mask) {
    if (Cout % 4) return SPAIR_ERR_ALIGN;
    if (misc_conv0_reads_unpadded(B, Hin, C, k, Cout)) {
        if (s < 1 || s > 4) return SPAIR_ERR_UNSUPPORTED;
        if (out_bf16 && Cout == 128 && s == 2 && (Hin & 1) == 0) {
            hipLaunchKernelGGL(k_conv0_fwd_c1k4_mfma, dim3((Hout + C0_ROWS - 1) / C0_ROWS, B), dim3(256), c0_mfma_xs_bytes(Hin) + 4 * 16 * C0_TP, st, x,
                               w, bias, reinterpret_cast<__bf16*>(out), Hin, Hout, I, pre, mask);
            SPAIR_CHECK_LAUNCH();
            return SPAIR_OK;
        }
        hipLaunchKernelGGL(k_conv0_fwd_c1k4, dim3((Hout + C0_ROWS - 1) / C0_ROWS, B), dim3(256),
                           (size_t)(s * (C0_ROWS - 1) + 4) * Hin * sizeof(float), st, x, w, bias, out, Hin, s, Hout, Cout, out_bf16, I, pre);
        SPAIR_CHECK_LAUNCH();
        return SPAIR_OK;
    }
    const size_t lds = (size_t)k * k * C * Cout * sizeof(float);
    if (lds > 64 * 1024) return SPAIR_ERR_UNSUPPORTED;
    const long long total = (long long)B * Hout * Hout * (Cout / 4);
    const unsigned grid = (unsigned)min((long long)4096, (total + 255) / 256);
    hipLaunchKernelGGL(k_conv0_fwd, dim3(grid), dim3(256), lds, st, xp, w, bias, out, B, Hin, C, k, s, Hout, Cout, out_bf16);
    SPAIR_CHECK_LAUNCH();
    return SPAIR_OK;
}

// the grey-scale 4x4 stem alone (parity tests): x [B][I][I] fp32 unpadded, zero outside [pre, pre + I) of the Hin x Hin padded frame
extern "C" int spair_stem_conv_fwd(const float* x, const float* w, const float* bias, void* out, int B, int I, int pad_pre, int Hin, int Hout,
                                   int Cout, int stride, int out_bf16, void* stream) {
    if (!x || !w || !bias || !out) return SPAIR_ERR_SHAPE;
    if (B <= 0 || I <= 0 || stride < 1 || Hin < I + pad_pre || Hout != (Hin - 4) / stride + 1) return SPAIR_ERR_SHAPE;
    if (!misc_conv0_reads_unpadded(B, Hin, 1, 4, Cout)) return SPAIR_ERR_UNSUPPORTED;
    return misc_conv0_fwd(x, nullptr, w, bias, reinterpret_cast<float*>(out), B, I, pad_pre, Hin, 1, 4, stride, Hout, Cout, out_bf16, (hipStream_t)stream);
}
// the same with the sign-bit mask of the output (bf16, 128 channels, stride 2): mask8 [B][Hout][Hout][16] bytes
extern "C" int spair_stem_conv_fwd_mask(const float* x, const float* w, const float* bias, void* out, void* mask8, int B, int I, int pad_pre, int Hin,
                                        int Hout, void* stream) {
    if (!x || !w || !bias || !out || !mask8) return SPAIR_ERR_SHAPE;
    if (B <= 0 || I <= 0 || Hin < I + pad_pre || Hout != (Hin - 4) / 2 + 1) return SPAIR_ERR_SHAPE;
    if (!misc_conv0_writes_mask(B, Hin, 1, 4, 2, 128, 1)) return SPAIR_ERR_UNSUPPORTED;
    return misc_conv0_fwd(x, nullptr, w, bias, reinterpret_cast<float*>(out), B, I, pad_pre, Hin, 1, 4, 2, Hout, 128, 1, (hipStream_t)stream,
                          reinterpret_cast<unsigned char*>(mask8));
}

// weight gradient: dW[co][ci][ky][kx] += sum_m dOut[m][co] * patch(m)[(ky,kx,ci)]; thread = (co, k) pairs
__global__ __launch_bounds__(256) void k_conv0_wgrad(const float* __restrict__ xp, const float* __restrict__ dout, float* __restrict__ dw,
                                                     int B, int Hin, int C, int k, int s, int Hout, int Cout, int rows_per_block) {
    extern __shared__ float sh[];    // dout tile [64][Cout] then patches [64][K0]
    const int K0 = k * k * C;
    float* dsh = sh;
    float* psh = sh + 64 * Cout;
    const long long M = (long long)B * Hout * Hout;
    const long long m_beg = (long long)blockIdx.x * rows_per_block, m_end = min(M, m_beg + rows_per_block);
    const int npair = Cout * K0;
    // each thread owns pairs p = threadIdx.x + i*256 (co = p % Cout, kk = p / Cout), up to 8 of them
    float acc[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    for (long long m0 = m_beg; m0 < m_end; m0 += 64) {
        const int nr = (int)min((long long)64, m_end - m0);
        __syncthreads();
        for (int i = threadIdx.x; i < nr * Cout; i += blockDim.x) dsh[i] = dout[(size_t)m0 * Cout + i];
        for (int i = threadIdx.x; i < nr * K0; i += blockDim.x) {
            const int rr = i / K0, kk = i - rr * K0;
            const long long m = m0 + rr;
            const int ox = (int)(m % Hout), oy = (int)((m / Hout) % Hout), b = (int)(m / ((long long)Hout * Hout));
            const int tap = kk / C, ci = kk - tap * C, ky = tap / k, kx = tap - ky * k;
            psh[i] = xp[(((size_t)b * Hin + oy * s + ky) * Hin + ox * s + kx) * C + ci];
        }
        __syncthreads();
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int p = threadIdx.x + i * 256;
            if (p >= npair) break;
            const int co = p % Cout, kk = p / Cout;
            float a = 0.f;
            for (int rr = 0; rr < nr; ++rr) a += dsh[rr * Cout + co] * psh[rr * K0 + kk];
            acc[i] += a;
        }
    }
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const int p = threadIdx.x + i * 256;
        if (p >= npair) break;
        const int co = p % Cout, kk = p / Cout;
        const int tap = kk / C, ci = kk - tap * C, ky = tap / k, kx = tap - ky * k;
        atomicAdd(&dw[(((size_t)co * C + ci) * k + ky) * k + kx], acc[i]);
    }
}
int misc_conv0_wgrad(const float* xp, const float* dout, float* dw, int B, int Hin, int C, int k, int s, int Hout, int Cout,
                     hipStream_t st) {
    const int K0 = k * k * C;
    if (Cout * K0 > 8 * 256) return SPAIR_ERR_UNSUPPORTED;
    const long long M = (long long)B * Hout * Hout;
    const int rpb = 2048;
    const size_t lds = (size_t)64 * (Cout + K0) * sizeof(float);
    hipLaunchKernelGGL(k_conv0_wgrad, dim3((unsigned)((M + rpb - 1) / rpb)), dim3(256), lds, st, xp, dout, dw, B, Hin, C, k, s, Hout,
                       Cout, rpb);
    SPAIR_CHECK_LAUNCH();
    return SPAIR_OK;
}
