// Persistent fused kernels for the sequential per-cell encoder (reference: models.py:68-117).
//
// MI355X-first design of the one truly sequential part of SPAIR: samples are independent, so ONE
// WORKGROUP OWNS ONE SAMPLE and walks all 3G-2 dependency wavefronts by itself -- no inter-workgroup
// synchronisation, no kernel boundary per layer (the per-wavefront path needs ~18 launches x 46 steps).
// Per wavefront the <=16 independent cells of the sample form one 16-row MFMA tile; activations,
// context records and latents stay in LDS; the ~0.9 MB of bf16 weights are streamed from L2 every
// step in MFMA-FRAGMENT-PACKED order (one contiguous 1 KiB wave-load per 16x32 operand, prepared by
// k_prep mode 4) straight into registers -- they are used once per step and not shared between
// waves, so LDS staging would be pure overhead (cdna_hip_programming.md §5, "GEMV / M <= 16" row).
// Everything the backward pass and the weight-gradient GEMMs need is written to the same HBM row
// buffers the per-wavefront path produces, so the two paths are interchangeable (and are compared
// against each other in tests/test_chain_gpu.py).
#include "cell_math.h"
#include "chain.h"
#include "stn_math.h"

namespace {

constexpr int MT = 16;                    // rows per wavefront tile
constexpr int F = 100, REC = 56, CTX = 224, A_ = 50, NP = 100, GLN = 784;
constexpr int KC = 352, LD_XC = KC + 8;   // [feat | ctx] padded to 11 k-steps
constexpr int KX = 160, LD_XT = KX + 8;   // [pass | box | attr | depth] padded to 5 k-steps
constexpr int KG = 800, LD_GL = KG + 8;   // glimpse padded to 25 k-steps
constexpr int LD_H = 256 + 8;
constexpr int LD_O = 112;

__device__ __forceinline__ bf16x8 as_frag(const uint4& v) {
    union { uint4 u; bf16x8 b; } c;
    c.u = v;
    return c.b;
}

// Weight-fragment pipeline: PD k-steps (x up to 4 column tiles) of 16-byte-per-lane loads in flight per wave.  The ring lives in
// registers across layer boundaries and barriers: pipe_fill() for layer l+1 is issued right after the MFMAs of layer l, so the L2
// latency of a layer's first fragments hides under the previous layer's epilogue and the workgroup barrier.
constexpr int NW = 8;                     // waves per workgroup
constexpr int NTH = NW * 64;
constexpr int PD = 4;
struct WPipe { uint4 q[PD][2]; };

template <int KT, int NT>
__device__ __forceinline__ void pipe_fill(const uint4* __restrict__ Wp, WPipe& p, int wave, int lane) {
    constexpr int MY = (NT + NW - 1) / NW;
#pragma unroll
    for (int d = 0; d < PD; ++d) {
        if (d >= KT) break;
#pragma unroll
        for (int j = 0; j < MY; ++j) {
            const int nt = wave + NW * j;
            if (nt < NT) p.q[d][j] = Wp[(size_t)(nt * KT + d) * 64 + lane];
        }
    }
}

// acc[j] (j-th column tile of this wave: nt = wave + 4j) = in[16, K] . Wp^T ; K = 32*(KT0+KT1), the first KT0 k-steps read inA,
// the rest inB.  Wp is fragment-packed: fragment (nt, kt) = 64 lanes x 16 B at Wp[(nt*KT + kt)*64 + lane].  Expects pipe_fill<KT,NT>.
template <int KT0, int KT1, int NT>
__device__ __forceinline__ void wg_gemm(const __bf16* inA, int ldA, const __bf16* inB, int ldB, const uint4* __restrict__ Wp, WPipe& p,
                                        f32x4 (&acc)[(NT + NW - 1) / NW], int wave, int lane) {
    constexpr int KT = KT0 + KT1, MY = (NT + NW - 1) / NW;
#pragma unroll
    for (int j = 0; j < MY; ++j) acc[j] = (f32x4){0.f, 0.f, 0.f, 0.f};
    const int arow = lane & 15, kg = (lane >> 4) * 8;
#pragma unroll 1
    for (int kt0 = 0; kt0 < KT; kt0 += PD) {
#pragma unroll
        for (int d = 0; d < PD; ++d) {
            const int kt = kt0 + d;
            if (kt < KT) {
                const __bf16* src = (kt < KT0) ? inA + arow * ldA + kt * 32 + kg : inB + arow * ldB + (kt - KT0) * 32 + kg;
                const bf16x8 af = *reinterpret_cast<const bf16x8*>(src);
#pragma unroll
                for (int j = 0; j < MY; ++j) {
                    const int nt = wave + NW * j;
                    if (nt < NT) {
                        acc[j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af, as_frag(p.q[d][j]), acc[j], 0, 0, 0);
                        if (kt + PD < KT) p.q[d][j] = Wp[(size_t)(nt * KT + kt + PD) * 64 + lane];
                    }
                }
            }
        }
    }
}

// epilogue: v = acc + bias (+relu); optional bf16 copy to LDS (next layer's input), fp32 copy to LDS (head outputs) and to the
// HBM row buffer (what backward / weight-gradient GEMMs read).  Lane holds col = nt*16 + (lane&15), rows (lane>>4)*4 + r.
template <int NT, bool RELU>
__device__ __forceinline__ void wg_store(const f32x4 (&acc)[(NT + NW - 1) / NW], const float* __restrict__ bias, int nout, __bf16* lds_bf, int ld_bf,
                                         float* lds_f, int ld_f, float* __restrict__ hbm, int ld_hbm, const int* row_r, int nc, int wave,
                                         int lane) {
    constexpr int MY = (NT + NW - 1) / NW;
#pragma unroll
    for (int j = 0; j < MY; ++j) {
        const int nt = wave + NW * j;
        if (nt >= NT) continue;
        const int n = nt * 16 + (lane & 15);
        if (n >= nout) continue;
        const float bv = bias[n];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int row = (lane >> 4) * 4 + r;
            float v = acc[j][r] + bv;
            if (RELU) v = fmaxf(v, 0.f);
            if (lds_bf) lds_bf[row * ld_bf + n] = (__bf16)v;
            if (lds_f) lds_f[row * ld_f + n] = v;
            if (row < nc) hbm[(size_t)row_r[row] * ld_hbm + n] = v;
        }
    }
}

}  // namespace

__global__ __launch_bounds__(NTH) void k_chain_fwd(ChainArgs a) {
    __shared__ __attribute__((aligned(16))) __bf16 Xc[MT * LD_XC];
    __shared__ __attribute__((aligned(16))) __bf16 XtZ[MT * LD_XT];
    __shared__ __attribute__((aligned(16))) __bf16 XtO[MT * LD_XT];
    __shared__ __attribute__((aligned(16))) __bf16 Gl[MT * LD_GL];
    __shared__ __attribute__((aligned(16))) __bf16 Ha[MT * LD_H];
    __shared__ __attribute__((aligned(16))) __bf16 Hb[MT * LD_H];
    __shared__ __attribute__((aligned(16))) float Ost[MT * LD_O];
    __shared__ __attribute__((aligned(16))) float recs[4][MT][REC];
    __shared__ float nb_sh[MT][4];
    __shared__ int row_r[MT], row_h[MT], row_w[MT], row_cp[MT];
    __shared__ int dstart_sh[3 * 32 + 2];
    __shared__ int nbr_sh[32 * 32 * 4];

    const CellLayout& L = a.L;
    const CellBufs& P = a.P;
    const CellHyper& H = a.H;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int b = blockIdx.x;
    const int G = L.G, T = 3 * G - 2;

    for (int i = tid; i < MT * LD_XC; i += NTH) Xc[i] = (__bf16)0.f;
    for (int i = tid; i < MT * LD_XT; i += NTH) { XtZ[i] = (__bf16)0.f; XtO[i] = (__bf16)0.f; }
    for (int i = tid; i < MT * LD_GL; i += NTH) Gl[i] = (__bf16)0.f;
    for (int i = tid; i < MT * LD_H; i += NTH) { Ha[i] = (__bf16)0.f; Hb[i] = (__bf16)0.f; }
    for (int i = tid; i < MT * LD_O; i += NTH) Ost[i] = 0.f;
    for (int i = tid; i <= T; i += NTH) dstart_sh[i] = P.diag_start[i];
    for (int i = tid; i < L.HW * 4; i += NTH) nbr_sh[i] = P.nbr[i];
    __syncthreads();

    WPipe pipe;
    pipe_fill<11, 7>(a.w[CW_BOX0], pipe, wave, lane);
    for (int t = 0; t < T; ++t) {
        const int c0 = dstart_sh[t];
        const int nc = dstart_sh[t + 1] - c0;
        float (*rec_cur)[REC] = recs[t & 3];
        if (tid < MT) {
            const int cp = c0 + min(tid, nc - 1);
            row_cp[tid] = cp;
            row_r[tid] = cp * L.B + b;
            row_h[tid] = P.cell_h[cp];
            row_w[tid] = P.cell_w[cp];
        }
        __syncthreads();
        // ---- S0: [feat | context] (models.py:71-76,292-320), 4 floats per thread
        for (int idx = tid; idx < MT * (KC / 4); idx += NTH) {
            const int row = idx / (KC / 4), c4 = (idx - row * (KC / 4)) * 4;
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (row < nc && c4 < F + CTX) {
                if (c4 < F) {
                    v = *reinterpret_cast<const float4*>(P.feat + ((size_t)(b * G + row_h[row]) * G + row_w[row]) * P.ld_feat + c4);
                } else {
                    const int s = (c4 - F) / REC, j = (c4 - F) - s * REC;
                    const int nbc = nbr_sh[row_cp[row] * 4 + s];
                    if (nbc >= 0) {
                        const int dt = (s == 0) ? 3 : (s == 1 ? 2 : 1);      // UL: t-3, U: t-2, UR and L: t-1
                        v = *reinterpret_cast<const float4*>(&recs[(t - dt) & 3][nbc - dstart_sh[t - dt]][j]);
                    } else {
                        v = *reinterpret_cast<const float4*>(P.edge + j);
                    }
                }
                const size_t r = row_r[row];
                *reinterpret_cast<float4*>(P.Xb + r * L.ld_xb + c4) = v;
                *reinterpret_cast<float4*>(P.Xz + r * L.ld_x + c4) = v;
                *reinterpret_cast<float4*>(P.Xo + r * L.ld_x + c4) = v;
            }
            bf16x4 o;
            o[0] = (__bf16)v.x; o[1] = (__bf16)v.y; o[2] = (__bf16)v.z; o[3] = (__bf16)v.w;
            *reinterpret_cast<bf16x4*>(&Xc[row * LD_XC + c4]) = o;
        }
        __syncthreads();
        // ---- z_where: box MLP (models.py:76-77)
        {
            f32x4 acc[1];
            wg_gemm<11, 0, 7>(Xc, LD_XC, nullptr, 0, a.w[CW_BOX0], pipe, acc, wave, lane);
            pipe_fill<4, 7>(a.w[CW_BOX1], pipe, wave, lane);
            wg_store<7, true>(acc, a.bias[CW_BOX0], 100, Ha, LD_H, nullptr, 0, P.Hb1, SP_LDH, row_r, nc, wave, lane);
        }
        __syncthreads();
        {
            f32x4 acc[1];
            wg_gemm<4, 0, 7>(Ha, LD_H, nullptr, 0, a.w[CW_BOX1], pipe, acc, wave, lane);
            pipe_fill<4, 7>(a.w[CW_BOXH], pipe, wave, lane);
            wg_store<7, true>(acc, a.bias[CW_BOX1], 100, Hb, LD_H, nullptr, 0, P.Hb2, SP_LDH, row_r, nc, wave, lane);
        }
        __syncthreads();
        {
            f32x4 acc[1];
            wg_gemm<4, 0, 7>(Hb, LD_H, nullptr, 0, a.w[CW_BOXH], pipe, acc, wave, lane);
            pipe_fill<25, 16>(a.w[CW_ENC0], pipe, wave, lane);
            wg_store<7, false>(acc, a.bias[CW_BOXH], NP + 8, nullptr, 0, Ost, LD_O, P.Ob, L.ld_ob, row_r, nc, wave, lane);
        }
        __syncthreads();
        // ---- box latents (models.py:322-381); passthrough -> z-net input
        for (int idx = tid; idx < MT * NP; idx += NTH) {
            const int row = idx / NP, i = idx - row * NP;
            const float v = Ost[row * LD_O + i];
            XtZ[row * LD_XT + i] = (__bf16)v;
            if (row < nc) P.Xz[(size_t)row_r[row] * L.ld_x + L.x_pass + i] = v;
        }
        if (tid < nc) {
            const int h = row_h[tid], w = row_w[tid];
            const size_t r = row_r[tid];
            float eps[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) eps[k] = P.eps_box[(((size_t)b * 4 + k) * G + h) * G + w];
            const BoxFwd o = box_forward(&Ost[tid * LD_O + NP], eps, H, h, w);
            float* st = P.stat + r * SP_LDSTAT;
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                st[ST_MU_BOX + k] = o.mu[k];
                st[ST_SD_BOX + k] = o.sd[k];
                rec_cur[tid][k] = o.box[k];
                nb_sh[tid][k] = o.nbox[k];
                XtZ[tid * LD_XT + NP + k] = (__bf16)o.box[k];
                XtO[tid * LD_XT + NP + k] = (__bf16)o.box[k];
                P.rec[r * L.ld_rec + k] = o.box[k];
                P.Xz[r * L.ld_x + L.x_box + k] = o.box[k];
                P.Xo[r * L.ld_x + L.x_box + k] = o.box[k];
                P.nbox[r * 4 + k] = o.nbox[k];
                P.z_where[(((size_t)b * 4 + k) * G + h) * G + w] = o.nbox[k];
            }
        }
        __syncthreads();
        // ---- z_what: glimpse (modules.py:216-273, border padding) + encoder MLP (models.py:383-391)
        for (int idx = tid; idx < nc * (GLN / 4); idx += NTH) {
            const int row = idx / (GLN / 4), e = (idx - row * (GLN / 4)) * 4;
            const int i = e / a.Pp, j0 = e - i * a.Pp;          // P % 4 == 0: the 4 elements share the row i
            float iy, my;
            stn_src_coord(nb_sh[row][3], 2.f * nb_sh[row][1] - 1.f, i, a.Pp, a.I, a.ac, true, iy, my);
            const float* img = a.x + (size_t)b * a.I * a.I;
            const int y0 = (int)floorf(iy);
            const float wy1 = iy - (float)y0, wy0 = 1.f - wy1;
            const bool yin = (y0 + 1) < a.I;
            const float* r0p = img + y0 * a.I;
            const float* r1p = img + (yin ? y0 + 1 : y0) * a.I;
            float out[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                float ix, mx;
                stn_src_coord(nb_sh[row][2], 2.f * nb_sh[row][0] - 1.f, j0 + q, a.Pp, a.I, a.ac, true, ix, mx);
                const int x0 = (int)floorf(ix);
                const float wx1 = ix - (float)x0, wx0 = 1.f - wx1;
                const int x1 = ((x0 + 1) < a.I) ? x0 + 1 : x0;
                const float m1 = ((x0 + 1) < a.I) ? 1.f : 0.f, n1 = yin ? 1.f : 0.f;
                out[q] = r0p[x0] * (wy0 * wx0) + m1 * r0p[x1] * (wy0 * wx1) + n1 * r1p[x0] * (wy1 * wx0) + m1 * n1 * r1p[x1] * (wy1 * wx1);
            }
            bf16x4 o;
            o[0] = (__bf16)out[0]; o[1] = (__bf16)out[1]; o[2] = (__bf16)out[2]; o[3] = (__bf16)out[3];
            *reinterpret_cast<bf16x4*>(&Gl[row * LD_GL + e]) = o;
            *reinterpret_cast<float4*>(P.glimpse + (size_t)row_r[row] * L.ld_gl + e) = make_float4(out[0], out[1], out[2], out[3]);
        }
        __syncthreads();
        {
            f32x4 acc[2];
            wg_gemm<25, 0, 16>(Gl, LD_GL, nullptr, 0, a.w[CW_ENC0], pipe, acc, wave, lane);
            pipe_fill<8, 8>(a.w[CW_ENC1], pipe, wave, lane);
            wg_store<16, true>(acc, a.bias[CW_ENC0], 256, Ha, LD_H, nullptr, 0, P.He1, SP_ENC_H1, row_r, nc, wave, lane);
        }
        __syncthreads();
        {
            f32x4 acc[1];
            wg_gemm<8, 0, 8>(Ha, LD_H, nullptr, 0, a.w[CW_ENC1], pipe, acc, wave, lane);
            pipe_fill<4, 7>(a.w[CW_ENC2], pipe, wave, lane);
            wg_store<8, true>(acc, a.bias[CW_ENC1], 128, Hb, LD_H, nullptr, 0, P.He2, SP_ENC_H2, row_r, nc, wave, lane);
        }
        __syncthreads();
        {
            f32x4 acc[1];
            wg_gemm<4, 0, 7>(Hb, LD_H, nullptr, 0, a.w[CW_ENC2], pipe, acc, wave, lane);
            pipe_fill<16, 7>(a.w[CW_Z0], pipe, wave, lane);
            wg_store<7, false>(acc, a.bias[CW_ENC2], 2 * A_, nullptr, 0, Ost, LD_O, P.Oe, L.ld_oe, row_r, nc, wave, lane);
        }
        __syncthreads();
        // ---- attributes (models.py:83-85)
        for (int idx = tid; idx < nc * A_; idx += NTH) {
            const int row = idx / A_, j = idx - row * A_;
            const size_t r = row_r[row];
            const float eps = P.eps_attr[(((size_t)b * A_ + j) * G + row_h[row]) * G + row_w[row]];
            float sd, attr;
            attr_forward(Ost[row * LD_O + j], Ost[row * LD_O + A_ + j], eps, sd, attr);
            rec_cur[row][4 + j] = attr;
            XtZ[row * LD_XT + NP + 4 + j] = (__bf16)attr;
            XtO[row * LD_XT + NP + 4 + j] = (__bf16)attr;
            P.sd_attr[r * L.ld_rec + j] = sd;
            P.rec[r * L.ld_rec + 4 + j] = attr;
            P.Za[r * L.ld_rec + j] = attr;
            P.Xz[r * L.ld_x + L.x_attr + j] = attr;
            P.Xo[r * L.ld_x + L.x_attr + j] = attr;
        }
        __syncthreads();
        // ---- z_depth (models.py:88-97)
        {
            f32x4 acc[1];
            wg_gemm<11, 5, 7>(Xc, LD_XC, XtZ, LD_XT, a.w[CW_Z0], pipe, acc, wave, lane);
            pipe_fill<4, 7>(a.w[CW_Z1], pipe, wave, lane);
            wg_store<7, true>(acc, a.bias[CW_Z0], 100, Ha, LD_H, nullptr, 0, P.Hz1, SP_LDH, row_r, nc, wave, lane);
        }
        __syncthreads();
        {
            f32x4 acc[1];
            wg_gemm<4, 0, 7>(Ha, LD_H, nullptr, 0, a.w[CW_Z1], pipe, acc, wave, lane);
            pipe_fill<4, 7>(a.w[CW_ZH], pipe, wave, lane);
            wg_store<7, true>(acc, a.bias[CW_Z1], 100, Hb, LD_H, nullptr, 0, P.Hz2, SP_LDH, row_r, nc, wave, lane);
        }
        __syncthreads();
        {
            f32x4 acc[1];
            wg_gemm<4, 0, 7>(Hb, LD_H, nullptr, 0, a.w[CW_ZH], pipe, acc, wave, lane);
            pipe_fill<16, 7>(a.w[CW_OBJ0], pipe, wave, lane);
            wg_store<7, false>(acc, a.bias[CW_ZH], NP + 2, nullptr, 0, Ost, LD_O, P.Oz, L.ld_oz, row_r, nc, wave, lane);
        }
        __syncthreads();
        for (int idx = tid; idx < MT * NP; idx += NTH) {
            const int row = idx / NP, i = idx - row * NP;
            const float v = Ost[row * LD_O + i];
            XtO[row * LD_XT + i] = (__bf16)v;
            if (row < nc) P.Xo[(size_t)row_r[row] * L.ld_x + L.x_pass + i] = v;
        }
        if (tid < nc) {
            const int h = row_h[tid], w = row_w[tid];
            const size_t r = row_r[tid];
            const float eps = P.eps_depth[((size_t)b * G + h) * G + w];
            float mu, sd, depth;
            depth_forward(Ost[tid * LD_O + NP], Ost[tid * LD_O + NP + 1], eps, H, mu, sd, depth);
            float* st = P.stat + r * SP_LDSTAT;
            st[ST_MU_DEPTH] = mu;
            st[ST_SD_DEPTH] = sd;
            rec_cur[tid][4 + A_] = depth;
            XtO[tid * LD_XT + NP + 4 + A_] = (__bf16)depth;
            P.rec[r * L.ld_rec + 4 + A_] = depth;
            P.Xo[r * L.ld_x + L.x_depth] = depth;
        }
        __syncthreads();
        // ---- z_pres (models.py:100-102,393-411)
        {
            f32x4 acc[1];
            wg_gemm<11, 5, 7>(Xc, LD_XC, XtO, LD_XT, a.w[CW_OBJ0], pipe, acc, wave, lane);
            pipe_fill<4, 7>(a.w[CW_OBJ1], pipe, wave, lane);
            wg_store<7, true>(acc, a.bias[CW_OBJ0], 100, Ha, LD_H, nullptr, 0, P.Ho1, SP_LDH, row_r, nc, wave, lane);
        }
        __syncthreads();
        {
            f32x4 acc[1];
            wg_gemm<4, 0, 7>(Ha, LD_H, nullptr, 0, a.w[CW_OBJ1], pipe, acc, wave, lane);
            pipe_fill<4, 1>(a.w[CW_OBJ2], pipe, wave, lane);
            wg_store<7, true>(acc, a.bias[CW_OBJ1], 100, Hb, LD_H, nullptr, 0, P.Ho2, SP_LDH, row_r, nc, wave, lane);
        }
        __syncthreads();
        {
            f32x4 acc[1];
            wg_gemm<4, 0, 1>(Hb, LD_H, nullptr, 0, a.w[CW_OBJ2], pipe, acc, wave, lane);
            pipe_fill<11, 7>(a.w[CW_BOX0], pipe, wave, lane);
            wg_store<1, false>(acc, a.bias[CW_OBJ2], 1, nullptr, 0, Ost, LD_O, P.Oo, L.ld_oo, row_r, nc, wave, lane);
        }
        __syncthreads();
        if (tid < nc) {
            const int h = row_h[tid], w = row_w[tid];
            const size_t r = row_r[tid];
            const float u = P.u_pres[((size_t)b * G + h) * G + w];
            const float pres = pres_forward(Ost[tid * LD_O], u, H);
            rec_cur[tid][REC - 1] = pres;
            P.rec[r * L.ld_rec + REC - 1] = pres;
            P.z_pres[((size_t)b * G + h) * G + w] = pres;
        }
        __syncthreads();
    }
}

int chain_fwd_supported(const SpairDims& d) {
    return d.dtype == SPAIR_BF16 && d.F == F && d.A == A_ && d.NP == NP && d.P == 28 && d.C == 1 && d.G <= 32 &&
           (d.G + 1) / 2 <= MT && d.G >= 2;
}

int chain_fwd(const ChainArgs& a, hipStream_t s) {
    hipLaunchKernelGGL(k_chain_fwd, dim3(a.L.B), dim3(NTH), 0, s, a);
    SPAIR_CHECK_LAUNCH();
    return SPAIR_OK;
}
