// Fused stack of 1x1 convolutions on bf16-stored NHWC activations (Backbone layers 3..n and conv_out, reference
// spair/modules.py:59-64,107-111): a 1x1 conv is a per-pixel Linear, so the whole stack is a row-wise MLP.
//
// One workgroup owns 128 pixels (rows) and runs every layer back to back: the 128x128 activation tile stays in LDS, the layer's
// 128x128 weight matrix is staged in LDS (the next layer's is prefetched into registers during the MFMAs), each layer's result is
// written back into the activation tile (bf16) and leaves for HBM through a coalesced 16-byte-per-lane pass -- every layer output
// is still stored, the weight-gradient GEMMs and (backward) the relu masks need them.  Per-layer launches of the generic GEMM took
// 25 us (forward) / 60 us (data gradient) each for 2 GFLOP: pure per-launch and per-tile latency.
//
//   forward : Y_l = relu(Y_{l-1} W_l^T + b_l) (bf16 out), last layer: fp32 out, no relu, N_last <= 128 columns
//   backward: dX_{l-1} = (dX_l Wd_l^T) * [X_{l-1} > 0], first input dY has K_first <= 128 columns (zero padded in LDS)
#include <hip/hip_runtime.h>

#include <cstring>

#include "common.h"
#include "gemm.h"

namespace {

typedef unsigned short u16;
constexpr int PW_BM = 128, PW_N = 128, PW_LD = PW_N + 16, PW_MAXL = 4;      // pitch 288 B = 32 mod 64: conflict-free ds_read_b128 fragments (K + 8: 2-way)

struct PwArgs {
    const void* X; int ldx, kx;            // first input: bf16 [M][ldx], kx valid columns (<= 128)
    const void* W[PW_MAXL]; int ldw[PW_MAXL], wrows[PW_MAXL], wcols[PW_MAXL];   // bf16 [wrows][ldw], wcols valid columns
    const float* bias[PW_MAXL];            // forward only
    const void* mask[PW_MAXL];             // backward only: bf16 [M][128] activations whose sign gates the layer's output
    void* Y[PW_MAXL];                      // bf16 [M][128] outputs (forward: all but the last layer; backward: all)
    float* Ylast; int ldlast, nlast;       // forward: fp32 output of the last layer
    int M, L;
};

// 16 bytes at element offset `off` of a bf16 tensor, zeros when !ok: a buffer load whose masked lanes go out of range (common.h) --
// written as `ok ? *p : 0` hipcc made every one of these a conditional load behind its own vmcnt(0) (16 serialised round trips
// for the first tile of a workgroup)
__device__ __forceinline__ uint4 ld16_or_zero(__amdgpu_buffer_rsrc_t r, size_t off, bool ok) {
    return buf_load16(r, ok ? (unsigned)(off * 2) : BUF_OOB);
}

// 8 bf16 of `v` zeroed where the matching bf16 of `m` is not > 0
__device__ __forceinline__ uint4 relu_gate8(uint4 v, uint4 m) {
    unsigned* pv = reinterpret_cast<unsigned*>(&v);
    const unsigned* pm = reinterpret_cast<const unsigned*>(&m);
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const unsigned lo = pm[i] & 0xffffu, hi = pm[i] >> 16;
        const unsigned keep = (((lo & 0x7fffu) != 0u && !(lo & 0x8000u)) ? 0xffffu : 0u) | (((hi & 0x7fffu) != 0u && !(hi & 0x8000u)) ? 0xffff0000u : 0u);
        pv[i] &= keep;
    }
    return v;
}

// one layer of the stack; `l` is a template parameter so that every a.X[l] is a fixed slot of the kernel-argument struct (a
// runtime-indexed pointer table would be fetched with FLAT loads)
template <bool BWD, int l>
__device__ __forceinline__ void pw_layer(const PwArgs& a, __bf16* As, __bf16* Ws, int m0, int tid, int lane, int wave, int s_kc, int s_r0,
                                         int wrow0, int fr, int fk) {
    const bool more = (l + 1) < a.L;
    // prefetch: next layer's weights, this layer's relu-gate activations (backward)
    uint4 wq[8], mq[8];
    {
        constexpr int l1 = (l + 1 < PW_MAXL) ? l + 1 : l;      // (never dereferenced past the last layer: `more` is false there)
        const u16* Wn = reinterpret_cast<const u16*>(more ? a.W[l1] : a.W[l]);
        const int wr = more ? a.wrows[l1] : a.wrows[l], wc = more ? a.wcols[l1] : a.wcols[l], lw = more ? a.ldw[l1] : a.ldw[l];
        const __amdgpu_buffer_rsrc_t rsW = buf_rsrc(Wn);
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int row = s_r0 + i * 16;
            wq[i] = ld16_or_zero(rsW, (size_t)min(row, wr - 1) * lw + min(s_kc, lw - 8), row < wr && s_kc < wc);
        }
        if (BWD) {
            const u16* Mk = reinterpret_cast<const u16*>(a.mask[l]);
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const int mr = min(m0 + s_r0 + i * 16, a.M - 1);
                mq[i] = *reinterpret_cast<const uint4*>(Mk + (size_t)mr * PW_N + s_kc);
            }
        }
    }
    __builtin_amdgcn_sched_barrier(0);        // the prefetch is ISSUED here: left to itself hipcc sinks these loads to their first use,
                                              // behind the MFMAs and the second barrier, and the layer waits for them there
    __syncthreads();                          // As / Ws of this layer complete
    f32x4 acc[2][8];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 8; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int kt = 0; kt < 4; ++kt) {
        bf16x8 af[2], bfr[8];
#pragma unroll
        for (int i = 0; i < 2; ++i) af[i] = *reinterpret_cast<const bf16x8*>(&As[(wrow0 + i * 16 + fr) * PW_LD + kt * 32 + fk]);
#pragma unroll
        for (int j = 0; j < 8; ++j) bfr[j] = *reinterpret_cast<const bf16x8*>(&Ws[(j * 16 + fr) * PW_LD + kt * 32 + fk]);
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 8; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[i], bfr[j], acc[i][j], 0, 0, 0);
    }
    __syncthreads();                          // every wave is done reading As / Ws
    // (unconditional on purpose: with the write under `if (more)` LLVM sinks the eight prefetch loads into that block -- behind the
    //  MFMAs and this barrier -- and the layer then waits a full memory round trip for them; on the last layer the copy is unused)
#pragma unroll
    for (int i = 0; i < 8; ++i) *reinterpret_cast<uint4*>(&Ws[(s_r0 + i * 16) * PW_LD + s_kc]) = wq[i];
    // epilogue: C layout -> lane holds column j*16 + (lane&15), rows wrow0 + i*16 + (lane>>4)*4 + r
    const int ccol = lane & 15, crow = (lane >> 4) * 4;
    if (!BWD && !more) {                      // last forward layer: fp32, no relu, straight from the registers
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int n = j * 16 + ccol;
            if (n >= a.nlast) continue;
            const float bv = a.bias[l][n];
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int m = m0 + wrow0 + i * 16 + crow + r;
                    if (m < a.M) a.Ylast[(size_t)m * a.ldlast + n] = acc[i][j][r] + bv;
                }
        }
        return;
    }
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const int n = j * 16 + ccol;
        const float bv = BWD ? 0.f : a.bias[l][n];
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                float v = acc[i][j][r] + bv;
                if (!BWD) v = fmaxf(v, 0.f);
                As[(wrow0 + i * 16 + crow + r) * PW_LD + n] = (__bf16)v;
            }
    }
    __syncthreads();
    // coalesced pass: (backward) gate by the saved activation's sign, store the layer output, keep it as the next input
    {
        const __amdgpu_buffer_rsrc_t rsY = buf_rsrc(a.Y[l]);      // rows past M store out of range: no branch around the store
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int row = s_r0 + i * 16;
            uint4 v = *reinterpret_cast<const uint4*>(&As[row * PW_LD + s_kc]);
            if (BWD) {
                v = relu_gate8(v, mq[i]);
                *reinterpret_cast<uint4*>(&As[row * PW_LD + s_kc]) = v;
            }
            buf_store16(rsY, (m0 + row < a.M) ? (unsigned)(((size_t)(m0 + row) * PW_N + s_kc) * 2) : BUF_OOB, v);
        }
    }
    // (the next iteration's first barrier orders these LDS writes before its MFMA reads)
}

template <bool BWD>
__global__ __launch_bounds__(256, 2) void k_pw_stack(PwArgs a) {
    __shared__ __attribute__((aligned(16))) __bf16 As[PW_BM * PW_LD];
    __shared__ __attribute__((aligned(16))) __bf16 Ws[PW_N * PW_LD];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int m0 = blockIdx.x * PW_BM;
    // staging map: 128 rows x 16 chunks of 8 bf16 = 2048 chunks, 8 per thread: chunk f = tid + i*256 -> row f>>4, k-chunk f&15
    const int s_kc = (tid & 15) * 8, s_r0 = tid >> 4;
    {
        const __amdgpu_buffer_rsrc_t rsX = buf_rsrc(a.X), rsW0 = buf_rsrc(a.W[0]);
        uint4 xq[8], wq0[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) {       // all sixteen loads in flight before the first LDS write
            const int row = s_r0 + i * 16;
            const int mr = min(m0 + row, a.M - 1);
            const bool okx = (m0 + row) < a.M && s_kc < a.kx;
            xq[i] = ld16_or_zero(rsX, (size_t)mr * a.ldx + min(s_kc, a.ldx - 8), okx);
            const bool okw = row < a.wrows[0] && s_kc < a.wcols[0];
            wq0[i] = ld16_or_zero(rsW0, (size_t)min(row, a.wrows[0] - 1) * a.ldw[0] + min(s_kc, a.ldw[0] - 8), okw);
        }
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int row = s_r0 + i * 16;
            *reinterpret_cast<uint4*>(&As[row * PW_LD + s_kc]) = xq[i];
            *reinterpret_cast<uint4*>(&Ws[row * PW_LD + s_kc]) = wq0[i];
        }
    }
    const int wrow0 = wave * 32;                 // this wave's 32 rows; all 128 columns
    const int fr = lane & 15, fk = (lane >> 4) * 8;
    if (a.L > 0) pw_layer<BWD, 0>(a, As, Ws, m0, tid, lane, wave, s_kc, s_r0, wrow0, fr, fk);
    if (a.L > 1) pw_layer<BWD, 1>(a, As, Ws, m0, tid, lane, wave, s_kc, s_r0, wrow0, fr, fk);
    if (a.L > 2) pw_layer<BWD, 2>(a, As, Ws, m0, tid, lane, wave, s_kc, s_r0, wrow0, fr, fk);
    if (a.L > 3) pw_layer<BWD, 3>(a, As, Ws, m0, tid, lane, wave, s_kc, s_r0, wrow0, fr, fk);
}

}  // namespace

// Forward stack.  W[l]: bf16 [cout_l][ldw_l] (cin = 128 columns), bias[l] fp32, Y[l]: bf16 [M][128] for l < L-1, Ylast fp32 [M][ldlast].
int spair_pw_stack_fwd16(const void* X, const void* const* W, const int* ldw, const int* cout, const float* const* bias, void* const* Y,
                         float* Ylast, int ldlast, int M, int L, hipStream_t s) {
    if (L < 1 || L > PW_MAXL || M <= 0) return SPAIR_ERR_SHAPE;
    if ((long long)M * PW_N * 2 >= 0xffffff00ll) return SPAIR_ERR_UNSUPPORTED;      // 32-bit byte offsets of the buffer loads
    PwArgs a;
    memset(&a, 0, sizeof(a));
    a.X = X; a.ldx = PW_N; a.kx = PW_N; a.M = M; a.L = L; a.Ylast = Ylast; a.ldlast = ldlast; a.nlast = cout[L - 1];
    for (int l = 0; l < L; ++l) {
        if (cout[l] > PW_N || ldw[l] < PW_N || (ldw[l] & 7) || (l < L - 1 && cout[l] != PW_N)) return SPAIR_ERR_UNSUPPORTED;
        a.W[l] = W[l]; a.ldw[l] = ldw[l]; a.wrows[l] = cout[l]; a.wcols[l] = PW_N; a.bias[l] = bias[l]; a.Y[l] = (l < L - 1) ? Y[l] : nullptr;
    }
    hipLaunchKernelGGL(k_pw_stack<false>, dim3((M + PW_BM - 1) / PW_BM), dim3(256), 0, s, a);
    SPAIR_CHECK_LAUNCH();
    return SPAIR_OK;
}

// Data-gradient stack, layers listed in BACKWARD order (l = 0 is the top layer).  dY: bf16 [M][ldd] with kd valid columns;
// Wd[l]: bf16 [128][ldw_l] = W_l transposed (cin rows, cout_l valid columns); gate[l]: the bf16 activation that was that layer's
// INPUT in the forward pass; dX[l]: bf16 [M][128].
int spair_pw_stack_bwd16(const void* dY, int ldd, int kd, const void* const* Wd, const int* ldw, const int* cout, const void* const* gate,
                         void* const* dX, int M, int L, hipStream_t s) {
    if (L < 1 || L > PW_MAXL || M <= 0 || kd > PW_N || (ldd & 7)) return SPAIR_ERR_SHAPE;
    if ((long long)M * ldd * 2 >= 0xffffff00ll) return SPAIR_ERR_UNSUPPORTED;         // 32-bit byte offsets of the buffer loads
    PwArgs a;
    memset(&a, 0, sizeof(a));
    a.X = dY; a.ldx = ldd; a.kx = kd; a.M = M; a.L = L;
    for (int l = 0; l < L; ++l) {
        if (cout[l] > PW_N || (ldw[l] & 7) || ldw[l] < 8 || (l > 0 && cout[l] != PW_N)) return SPAIR_ERR_UNSUPPORTED;
        a.W[l] = Wd[l]; a.ldw[l] = ldw[l]; a.wrows[l] = PW_N; a.wcols[l] = cout[l]; a.mask[l] = gate[l]; a.Y[l] = dX[l];
    }
    hipLaunchKernelGGL(k_pw_stack<true>, dim3((M + PW_BM - 1) / PW_BM), dim3(256), 0, s, a);
    SPAIR_CHECK_LAUNCH();
    return SPAIR_OK;
}

// ---- C ABI (include/spair_hip.h) -------------------------------------------------------------------
extern "C" int spair_conv1x1_stack_fwd16(const void* X, const void* const* W, const int* ldw, const int* cout, const float* const* bias,
                                         void* const* Y, float* Ylast, int ldlast, int M, int L, void* stream) {
    return spair_pw_stack_fwd16(X, W, ldw, cout, bias, Y, Ylast, ldlast, M, L, (hipStream_t)stream);
}
extern "C" int spair_conv1x1_stack_bwd16(const void* dY, int ldd, int kd, const void* const* Wd, const int* ldw, const int* cout,
                                         const void* const* gate, void* const* dX, int M, int L, void* stream) {
    return spair_pw_stack_bwd16(dY, ldd, kd, Wd, ldw, cout, gate, dX, M, L, (hipStream_t)stream);
}
