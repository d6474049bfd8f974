// MFMA GEMM core for the SPAIR hot path (gfx950 / CDNA4, wave64).
//
// Everything matmul-shaped in the step goes through the two kernels here:
//   gemm_nt : C[M,N] = epi(A[M,K] . B[N,K]^T)   forward Linear / conv (implicit GEMM), data-grad
//             (with a pre-transposed weight copy).  A may be a conv gather (ConvDesc).
//   gemm_tn : C[M,N] += sum_r A[r,M] . B[r,N]      weight-grad, split over the long row dimension,
//             fp32 atomics into a zeroed staging buffer.  B may be a conv gather.
// Operands live in HBM as fp32 (weights optionally as bf16 copies); the MFMA type is chosen per
// call: exact fp32 (v_mfma_f32_16x16x4_f32) or bf16 inputs / fp32 accumulate
// (v_mfma_f32_16x16x32_bf16), conversion happening in registers on the way into LDS.
// Fragment maps follow /opt/skills/guides/cdna_hip_programming.md §3:
//   A: lane l holds A[row l&15][k-group l>>4], B: B[k-group l>>4][col l&15],
//   C/D: lane l, reg r -> row (l>>4)*4 + r, col l&15.
#include "common.h"
#include "spair_hip.h"
#include "gemm.h"

template <int MMA> struct MmaTraits;
template <> struct MmaTraits<SPAIR_F32> {
    static constexpr int BK = 16, PAD = 4;
    using lds_t = float;
};
template <> struct MmaTraits<SPAIR_BF16> {
    static constexpr int BK = 32, PAD = 8;
    using lds_t = __bf16;
};


// Implicit-GEMM addressing without divisions in the K loop: a GEMM row is a position (b,y,x) of the logical output grid and a
// GEMM column a tap (ky,kx,ci); both are decoded once and then ADVANCED incrementally as the tiles march on.
struct ConvRow { int b, y, x; };
struct ConvTap { int ky, kx, ci; };

__device__ __forceinline__ void conv_row_init(const ConvDesc& c, int m, ConvRow& r) {
    const int hw = c.Hout * c.Wout;
    r.b = m / hw;
    const int rem = m - r.b * hw;
    r.y = rem / c.Wout;
    r.x = rem - r.y * c.Wout;
}
__device__ __forceinline__ void conv_row_advance(const ConvDesc& c, ConvRow& r, int step) {
    r.x += step;
    while (r.x >= c.Wout) { r.x -= c.Wout; ++r.y; }
    while (r.y >= c.Hout) { r.y -= c.Hout; ++r.b; }
}
__device__ __forceinline__ void conv_tap_init(const ConvDesc& c, int k, ConvTap& t) {
    const int tap = k / c.Cin;
    t.ci = k - tap * c.Cin;
    t.ky = tap / c.kw;
    t.kx = tap - t.ky * c.kw;
}
__device__ __forceinline__ void conv_tap_advance(const ConvDesc& c, ConvTap& t, int step) {
    t.ci += step;
    while (t.ci >= c.Cin) {
        t.ci -= c.Cin;
        if (++t.kx == c.kw) { t.kx = 0; ++t.ky; }
    }
}
// 4 consecutive k-elements of one tap (Cin % 4 == 0), bounds-checked (zero outside the tensor)
__device__ __forceinline__ float4 conv_load4(const float* __restrict__ In, const ConvDesc& c, const ConvRow& r, const ConvTap& t) {
    const int sy = r.y * c.sy + c.oy + t.ky * c.dky;
    const int sx = r.x * c.sx + c.ox + t.kx * c.dkx;
    if (sy < 0 || sy >= c.Hin || sx < 0 || sx >= c.Win) return make_float4(0.f, 0.f, 0.f, 0.f);
    const size_t off = (((size_t)r.b * c.Hin + sy) * c.Win + sx) * c.Cin + t.ci;
    return *reinterpret_cast<const float4*>(In + off);
}
// forward-geometry gather for the weight-gradient B operand (always inside the tensor); handles Cin % 4 != 0 by 4 scalar loads
__device__ __forceinline__ float4 conv_gather4(const float* __restrict__ In, const ConvDesc& c, const ConvRow& r, const int (&tapoff)[4], bool vec) {
    const size_t base = (((size_t)r.b * c.Hin + r.y * c.sy + c.oy) * c.Win + r.x * c.sx + c.ox) * c.Cin;
    if (vec) return *reinterpret_cast<const float4*>(In + base + tapoff[0]);
    return make_float4(In[base + tapoff[0]], In[base + tapoff[1]], In[base + tapoff[2]], In[base + tapoff[3]]);
}

template <int MMA, int BM, int BN, bool ACONV>
__global__ __launch_bounds__(256) void gemm_nt_kernel(GemmNT g) {
    using T = MmaTraits<MMA>;
    using lds_t = typename T::lds_t;
    constexpr int BK = T::BK, LD = BK + T::PAD;
    constexpr int WM = BM / 2, WN = BN / 2, TM = WM / 16, TN = WN / 16;
    constexpr int KQ = BK / 4;                    // float4 chunks per A row
    constexpr int NA = BM * KQ / 256;             // float4 loads per thread for A
    constexpr bool BF = (MMA == SPAIR_BF16);
    constexpr int BQ = BF ? BK / 8 : BK / 4;      // 16-byte chunks per B row
    constexpr int NB = BN * BQ / 256;
    static_assert(NA >= 1 && NB >= 1, "tile too small for 256 threads");

    __shared__ __attribute__((aligned(16))) lds_t As[BM * LD];
    __shared__ __attribute__((aligned(16))) lds_t Bs[BN * LD];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int m0 = blockIdx.x * BM, n0 = blockIdx.y * BN;

    // per-thread staging coordinates
    int a_row[NA], a_kq[NA];
    ConvRow a_cr[NA];
    ConvTap a_ct[NA];
    bool a_ok[NA];
#pragma unroll
    for (int i = 0; i < NA; ++i) {
        const int f = tid + i * 256;
        a_row[i] = f / KQ;
        a_kq[i] = f - a_row[i] * KQ;
        const int gm = m0 + a_row[i];
        a_ok[i] = gm < g.M;
        a_cr[i].b = a_cr[i].y = a_cr[i].x = 0;
        a_ct[i].ky = a_ct[i].kx = a_ct[i].ci = 0;
        if (ACONV && a_ok[i]) {
            conv_row_init(g.conv, gm, a_cr[i]);
            conv_tap_init(g.conv, a_kq[i] * 4, a_ct[i]);
        }
    }
    int b_row[NB], b_q[NB];
    bool b_ok[NB];
#pragma unroll
    for (int i = 0; i < NB; ++i) {
        const int f = tid + i * 256;
        b_row[i] = f / BQ;
        b_q[i] = f - b_row[i] * BQ;
        b_ok[i] = (n0 + b_row[i]) < g.N;
    }

    float4 ra[NA];
    uint4 rb[NB];
    auto load_tiles = [&](int k0) {
#pragma unroll
        for (int i = 0; i < NA; ++i) {
            const int k = k0 + a_kq[i] * 4;
            if (a_ok[i] && k < g.K) {
                if (ACONV) ra[i] = conv_load4(g.A, g.conv, a_cr[i], a_ct[i]);
                else ra[i] = *reinterpret_cast<const float4*>(g.A + (size_t)(m0 + a_row[i]) * g.lda + k);
            } else {
                ra[i] = make_float4(0.f, 0.f, 0.f, 0.f);
            }
            if (ACONV) conv_tap_advance(g.conv, a_ct[i], BK);   // load_tiles is called once per consecutive K tile
        }
#pragma unroll
        for (int i = 0; i < NB; ++i) {
            const int k = k0 + b_q[i] * (BF ? 8 : 4);
            if (b_ok[i] && k < g.K) {
                if (BF) rb[i] = *reinterpret_cast<const uint4*>(reinterpret_cast<const u16*>(g.B) + (size_t)(n0 + b_row[i]) * g.ldb + k);
                else rb[i] = *reinterpret_cast<const uint4*>(reinterpret_cast<const float*>(g.B) + (size_t)(n0 + b_row[i]) * g.ldb + k);
            } else {
                rb[i] = make_uint4(0u, 0u, 0u, 0u);
            }
        }
    };
    auto store_tiles = [&]() {
#pragma unroll
        for (int i = 0; i < NA; ++i) {
            lds_t* dst = &As[a_row[i] * LD + a_kq[i] * 4];
            if constexpr (BF) {
                bf16x4 v;
                v[0] = (__bf16)ra[i].x; v[1] = (__bf16)ra[i].y; v[2] = (__bf16)ra[i].z; v[3] = (__bf16)ra[i].w;
                *reinterpret_cast<bf16x4*>(dst) = v;
            } else {
                *reinterpret_cast<float4*>(dst) = ra[i];
            }
        }
#pragma unroll
        for (int i = 0; i < NB; ++i) {
            lds_t* dst = &Bs[b_row[i] * LD + b_q[i] * (BF ? 8 : 4)];
            *reinterpret_cast<uint4*>(dst) = rb[i];
        }
    };

    f32x4 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

    const int nk = (g.K + BK - 1) / BK;
    load_tiles(0);
    store_tiles();
    __syncthreads();
    for (int kt = 0; kt < nk; ++kt) {
        if (kt + 1 < nk) load_tiles((kt + 1) * BK);
        const int arow = wm * WM + (lane & 15), brow = wn * WN + (lane & 15), kg = lane >> 4;
        if constexpr (BF) {
            bf16x8 af[TM], bfr[TN];
#pragma unroll
            for (int i = 0; i < TM; ++i) af[i] = *reinterpret_cast<const bf16x8*>(&As[(arow + i * 16) * LD + kg * 8]);
#pragma unroll
            for (int j = 0; j < TN; ++j) bfr[j] = *reinterpret_cast<const bf16x8*>(&Bs[(brow + j * 16) * LD + kg * 8]);
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[i], bfr[j], acc[i][j], 0, 0, 0);
        } else {
#pragma unroll
            for (int kk = 0; kk < BK / 4; ++kk) {
                float af[TM], bfr[TN];
#pragma unroll
                for (int i = 0; i < TM; ++i) af[i] = As[(arow + i * 16) * LD + kk * 4 + kg];
#pragma unroll
                for (int j = 0; j < TN; ++j) bfr[j] = Bs[(brow + j * 16) * LD + kk * 4 + kg];
#pragma unroll
                for (int i = 0; i < TM; ++i)
#pragma unroll
                    for (int j = 0; j < TN; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(af[i], bfr[j], acc[i][j], 0, 0, 0);
            }
        }
        __syncthreads();
        if (kt + 1 < nk) {
            store_tiles();
            __syncthreads();
        }
    }

    // epilogue
    const int col_l = lane & 15, rgrp = (lane >> 4) * 4;
#pragma unroll
    for (int i = 0; i < TM; ++i) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int m = m0 + wm * WM + i * 16 + rgrp + r;
            if (m >= g.M) continue;
            size_t crow;
            if (g.use_cmap) {
                const int hw = g.cmap.Hout * g.cmap.Wout;
                const int b = m / hw, rem = m - b * hw, y = rem / g.cmap.Wout, x = rem - y * g.cmap.Wout;
                crow = ((size_t)b * g.cmap.Hc + (y * g.cmap.osy + g.cmap.ooy)) * g.cmap.Wc + (x * g.cmap.osx + g.cmap.oox);
            } else {
                crow = (size_t)m;
            }
            float* cptr = g.C + crow * g.ldc;
#pragma unroll
            for (int j = 0; j < TN; ++j) {
                const int n = n0 + wn * WN + j * 16 + col_l;
                if (n >= g.N) continue;
                float v = acc[i][j][r];
                if (g.bias) v += g.bias[n];
                if (g.accumulate) v += cptr[n];
                if (g.relu) v = fmaxf(v, 0.f);
                if (g.mask) v = (g.mask[crow * g.ldmask + n] > 0.f) ? v : 0.f;
                if (g.sprite_ch > 0) {
                    const float t = ((n % g.sprite_ch) == g.sprite_ch - 1) ? v * g.alpha_scale + g.alpha_bias : v * g.obj_scale;
                    v = 1.f / (expf(-t) + 1.f);
                }
                cptr[n] = v;
            }
        }
    }
}

template <int MMA, int BM, int BN>
static int launch_nt(const GemmNT& g, bool conv, hipStream_t s) {
    dim3 grid(ceil_div(g.M, BM), ceil_div(g.N, BN));
    if (conv) hipLaunchKernelGGL((gemm_nt_kernel<MMA, BM, BN, true>), grid, dim3(256), 0, s, g);
    else hipLaunchKernelGGL((gemm_nt_kernel<MMA, BM, BN, false>), grid, dim3(256), 0, s, g);
    SPAIR_CHECK_LAUNCH();
    return SPAIR_OK;
}

int spair_gemm_nt_impl(const GemmNT& g, bool conv, int dtype, hipStream_t s) {
    if (g.M <= 0 || g.N <= 0 || g.K <= 0) return SPAIR_ERR_SHAPE;
    if ((g.K & 3) || (!conv && (g.lda & 3))) return SPAIR_ERR_ALIGN;
    if (dtype == SPAIR_BF16 && ((g.K & 7) || (g.ldb & 7))) return SPAIR_ERR_ALIGN;
    if (dtype == SPAIR_F32 && (g.ldb & 3)) return SPAIR_ERR_ALIGN;
    if (conv && (g.conv.Cin & 3)) return SPAIR_ERR_ALIGN;
    const bool big = (g.M >= 8192 && g.N >= 96);
    if (dtype == SPAIR_BF16) return big ? launch_nt<SPAIR_BF16, 128, 128>(g, conv, s) : launch_nt<SPAIR_BF16, 64, 64>(g, conv, s);
    if (dtype == SPAIR_F32) return big ? launch_nt<SPAIR_F32, 128, 128>(g, conv, s) : launch_nt<SPAIR_F32, 64, 64>(g, conv, s);
    return SPAIR_ERR_DTYPE;
}

// ---------------------------------------------------------------------------------------------
// TN: C[M,N] += sum_r A[r, m] * B[r, n]   (weight gradients; fp32 MFMA, split over r, atomics)
// ---------------------------------------------------------------------------------------------

template <int BM, int BN, bool BCONV>
__global__ __launch_bounds__(256) void gemm_tn_kernel(GemmTN g) {
    constexpr int BK = 16, LDA = BM + 16, LDB = BN + 16;
    constexpr int WM = BM / 2, WN = BN / 2, TM = WM / 16, TN = WN / 16;
    constexpr int NA = BK * BM / 4 / 256, NB = BK * BN / 4 / 256;
    __shared__ __attribute__((aligned(16))) float As[BK * LDA];
    __shared__ __attribute__((aligned(16))) float Bs[BK * LDB];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    // XCD-aware tile order (1-D grid): all (m,n) tiles of one row split run back to back on ONE XCD (blocks id, id+8, ... share an
    // XCD under round-robin dispatch), so the split's rows of A and B -- for a conv weight-gradient the same input pixels seen
    // through all kh*kw taps -- are fetched from HBM once and then hit that XCD's L2.  Speed only; any mapping is correct.
    int mt_, nt_, sp_;
    {
        const int id = blockIdx.x, ntm = g.tiles_m, ntn = g.tiles_n;
        if ((g.nsplit & 7) == 0) {
            const int xcd = id & 7, j = id >> 3;
            mt_ = j % ntm; nt_ = (j / ntm) % ntn; sp_ = (j / (ntm * ntn)) * 8 + xcd;
        } else {
            mt_ = id % ntm; nt_ = (id / ntm) % ntn; sp_ = id / (ntm * ntn);
        }
    }
    const int m0 = mt_ * BM, n0 = nt_ * BN;
    const int r_begin = sp_ * g.rows_per_split;
    const int r_end = min(g.R, r_begin + g.rows_per_split);

    f32x4 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

    float4 ra[NA], rb[NB];
    // weight-gradient B operand as a conv gather: the column (tap) of each of this thread's chunks is fixed for the whole kernel,
    // its row (b,y,x) advances by BK per tile -- no division in the loop
    ConvRow b_cr[NB];
    int b_tapoff[NB][4];
    bool b_vec = true;
    if (BCONV) {
        b_vec = (g.conv.Cin & 3) == 0;
#pragma unroll
        for (int i = 0; i < NB; ++i) {
            const int f = tid + i * 256, kr = f / (BN / 4), nq = f - kr * (BN / 4);
            conv_row_init(g.conv, min(r_begin + kr, g.R - 1), b_cr[i]);
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                ConvTap t;
                conv_tap_init(g.conv, min(n0 + nq * 4 + e, g.N - 1), t);
                b_tapoff[i][e] = (t.ky * g.conv.dky * g.conv.Win + t.kx * g.conv.dkx) * g.conv.Cin + t.ci;
            }
        }
    }
    float csum[NA][4];
#pragma unroll
    for (int i = 0; i < NA; ++i) csum[i][0] = csum[i][1] = csum[i][2] = csum[i][3] = 0.f;
    const bool do_colsum = g.colsum_out != nullptr && nt_ == 0;
    auto load_tiles = [&](int r0) {
#pragma unroll
        for (int i = 0; i < NA; ++i) {
            const int f = tid + i * 256, kr = f / (BM / 4), mq = f - kr * (BM / 4);
            const int r = r0 + kr, m = m0 + mq * 4;
            ra[i] = (r < r_end && m < g.M) ? *reinterpret_cast<const float4*>(g.A + (size_t)r * g.lda + m)
                                           : make_float4(0.f, 0.f, 0.f, 0.f);
            if (do_colsum) { csum[i][0] += ra[i].x; csum[i][1] += ra[i].y; csum[i][2] += ra[i].z; csum[i][3] += ra[i].w; }
        }
#pragma unroll
        for (int i = 0; i < NB; ++i) {
            const int f = tid + i * 256, kr = f / (BN / 4), nq = f - kr * (BN / 4);
            const int r = r0 + kr, n = n0 + nq * 4;
            if (r < r_end && n < g.N) {
                if (BCONV) {
                    rb[i] = conv_gather4(g.B, g.conv, b_cr[i], b_tapoff[i], b_vec);
                } else {
                    rb[i] = *reinterpret_cast<const float4*>(g.B + (size_t)r * g.ldb + n);
                }
            } else {
                rb[i] = make_float4(0.f, 0.f, 0.f, 0.f);
            }
            if (BCONV) conv_row_advance(g.conv, b_cr[i], BK);
        }
    };
    auto store_tiles = [&]() {
#pragma unroll
        for (int i = 0; i < NA; ++i) {
            const int f = tid + i * 256, kr = f / (BM / 4), mq = f - kr * (BM / 4);
            *reinterpret_cast<float4*>(&As[kr * LDA + mq * 4]) = ra[i];
        }
#pragma unroll
        for (int i = 0; i < NB; ++i) {
            const int f = tid + i * 256, kr = f / (BN / 4), nq = f - kr * (BN / 4);
            *reinterpret_cast<float4*>(&Bs[kr * LDB + nq * 4]) = rb[i];
        }
    };

    if (r_begin < r_end) {
        load_tiles(r_begin);
        store_tiles();
        __syncthreads();
        for (int r0 = r_begin; r0 < r_end; r0 += BK) {
            const bool more = (r0 + BK) < r_end;
            if (more) load_tiles(r0 + BK);
            const int arow = wm * WM + (lane & 15), brow = wn * WN + (lane & 15), kg = lane >> 4;
#pragma unroll
            for (int kk = 0; kk < BK / 4; ++kk) {
                float af[TM], bfr[TN];
#pragma unroll
                for (int i = 0; i < TM; ++i) af[i] = As[(kk * 4 + kg) * LDA + arow + i * 16];
#pragma unroll
                for (int j = 0; j < TN; ++j) bfr[j] = Bs[(kk * 4 + kg) * LDB + brow + j * 16];
#pragma unroll
                for (int i = 0; i < TM; ++i)
#pragma unroll
                    for (int j = 0; j < TN; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(af[i], bfr[j], acc[i][j], 0, 0, 0);
            }
            __syncthreads();
            if (more) {
                store_tiles();
                __syncthreads();
            }
        }
    }
    if (do_colsum) {   // bias gradient: column sums of the A operand, reduced over this block's rows in LDS, one atomic per column
        constexpr int GR = 256 / (BM / 4);
        __syncthreads();
        float* scr = reinterpret_cast<float*>(As);
        const int mq = tid % (BM / 4), grp = tid / (BM / 4);
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            float t = 0.f;
#pragma unroll
            for (int i = 0; i < NA; ++i) t += csum[i][e];
            scr[grp * BM + mq * 4 + e] = t;
        }
        __syncthreads();
        for (int m = tid; m < BM; m += 256) {
            float t = 0.f;
#pragma unroll
            for (int q = 0; q < GR; ++q) t += scr[q * BM + m];
            if (m0 + m < g.Mstore) atomicAdd(&g.colsum_out[m0 + m], t);
        }
    }
    const int col_l = lane & 15, rgrp = (lane >> 4) * 4;
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int m = m0 + wm * WM + i * 16 + rgrp + r;
            if (m >= g.Mstore) continue;
#pragma unroll
            for (int j = 0; j < TN; ++j) {
                const int n = n0 + wn * WN + j * 16 + col_l;
                if (n >= g.Nstore) continue;
                int nc = n;
                if (g.cw_cin > 0) { const int tap = n / g.cw_cin, ci = n - tap * g.cw_cin; nc = ci * g.cw_taps + tap; }
                atomicAdd(&g.C[(size_t)m * g.ldc + nc], acc[i][j][r]);
            }
        }
}


// ---- bf16 TN: operands are staged row-major ([r][col], exactly as they sit in HBM) and the k-contiguous
// MFMA fragments come out of LDS through gfx950's transposing read ds_read_b64_tr_b16: per 16-lane group,
// lane 4q+p supplies the address of row q / columns 4p..4p+3 of a 4x16 block and receives column (lane&15)
// of the 4 rows (cdna_hip_programming.md T10).  Two such reads give the 8 k-values a 16x16x32 operand needs.
typedef short v4s_t __attribute__((ext_vector_type(4)));
__device__ __forceinline__ bf16x8 lds_tr_frag(const __bf16* tile, int ld, int k0, int c0, int lane) {
    const int g = lane >> 4, i = lane & 15, q = i >> 2, p = i & 3;
    const __bf16* a0 = tile + (k0 + 8 * g + q) * ld + c0 + 4 * p;
    const __bf16* a1 = a0 + 4 * ld;
    const v4s_t lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) v4s_t*)(a0));
    const v4s_t hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) v4s_t*)(a1));
    union { struct { v4s_t lo, hi; } s; bf16x8 v; } u;
    u.s.lo = lo; u.s.hi = hi;
    return u.v;
}

template <int BM, int BN, bool BCONV>
__global__ __launch_bounds__(256) void gemm_tn_bf16_kernel(GemmTN g) {
    constexpr int BK = 32, LDA = BM + 8, LDB = BN + 8;
    constexpr int WM = BM / 2, WN = BN / 2, TM = WM / 16, TN = WN / 16;
    constexpr int NA = BK * BM / 4 / 256, NB = BK * BN / 4 / 256;
    __shared__ __attribute__((aligned(16))) __bf16 As[BK * LDA];
    __shared__ __attribute__((aligned(16))) __bf16 Bs[BK * LDB];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    // XCD-aware tile order (1-D grid): all (m,n) tiles of one row split run back to back on ONE XCD (blocks id, id+8, ... share an
    // XCD under round-robin dispatch), so the split's rows of A and B -- for a conv weight-gradient the same input pixels seen
    // through all kh*kw taps -- are fetched from HBM once and then hit that XCD's L2.  Speed only; any mapping is correct.
    int mt_, nt_, sp_;
    {
        const int id = blockIdx.x, ntm = g.tiles_m, ntn = g.tiles_n;
        if ((g.nsplit & 7) == 0) {
            const int xcd = id & 7, j = id >> 3;
            mt_ = j % ntm; nt_ = (j / ntm) % ntn; sp_ = (j / (ntm * ntn)) * 8 + xcd;
        } else {
            mt_ = id % ntm; nt_ = (id / ntm) % ntn; sp_ = id / (ntm * ntn);
        }
    }
    const int m0 = mt_ * BM, n0 = nt_ * BN;
    const int r_begin = sp_ * g.rows_per_split;
    const int r_end = min(g.R, r_begin + g.rows_per_split);

    f32x4 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

    float4 ra[NA], rb[NB];
    // weight-gradient B operand as a conv gather: the column (tap) of each of this thread's chunks is fixed for the whole kernel,
    // its row (b,y,x) advances by BK per tile -- no division in the loop
    ConvRow b_cr[NB];
    int b_tapoff[NB][4];
    bool b_vec = true;
    if (BCONV) {
        b_vec = (g.conv.Cin & 3) == 0;
#pragma unroll
        for (int i = 0; i < NB; ++i) {
            const int f = tid + i * 256, kr = f / (BN / 4), nq = f - kr * (BN / 4);
            conv_row_init(g.conv, min(r_begin + kr, g.R - 1), b_cr[i]);
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                ConvTap t;
                conv_tap_init(g.conv, min(n0 + nq * 4 + e, g.N - 1), t);
                b_tapoff[i][e] = (t.ky * g.conv.dky * g.conv.Win + t.kx * g.conv.dkx) * g.conv.Cin + t.ci;
            }
        }
    }
    float csum[NA][4];
#pragma unroll
    for (int i = 0; i < NA; ++i) csum[i][0] = csum[i][1] = csum[i][2] = csum[i][3] = 0.f;
    const bool do_colsum = g.colsum_out != nullptr && nt_ == 0;
    auto load_tiles = [&](int r0) {
#pragma unroll
        for (int i = 0; i < NA; ++i) {
            const int f = tid + i * 256, kr = f / (BM / 4), mq = f - kr * (BM / 4);
            const int r = r0 + kr, m = m0 + mq * 4;
            ra[i] = (r < r_end && m < g.M) ? *reinterpret_cast<const float4*>(g.A + (size_t)r * g.lda + m)
                                           : make_float4(0.f, 0.f, 0.f, 0.f);
            if (do_colsum) { csum[i][0] += ra[i].x; csum[i][1] += ra[i].y; csum[i][2] += ra[i].z; csum[i][3] += ra[i].w; }
        }
#pragma unroll
        for (int i = 0; i < NB; ++i) {
            const int f = tid + i * 256, kr = f / (BN / 4), nq = f - kr * (BN / 4);
            const int r = r0 + kr, n = n0 + nq * 4;
            if (r < r_end && n < g.N) {
                if (BCONV) {
                    rb[i] = conv_gather4(g.B, g.conv, b_cr[i], b_tapoff[i], b_vec);
                } else {
                    rb[i] = *reinterpret_cast<const float4*>(g.B + (size_t)r * g.ldb + n);
                }
            } else {
                rb[i] = make_float4(0.f, 0.f, 0.f, 0.f);
            }
            if (BCONV) conv_row_advance(g.conv, b_cr[i], BK);
        }
    };
    auto store_tiles = [&]() {
#pragma unroll
        for (int i = 0; i < NA; ++i) {
            const int f = tid + i * 256, kr = f / (BM / 4), mq = f - kr * (BM / 4);
            bf16x4 v;
            v[0] = (__bf16)ra[i].x; v[1] = (__bf16)ra[i].y; v[2] = (__bf16)ra[i].z; v[3] = (__bf16)ra[i].w;
            *reinterpret_cast<bf16x4*>(&As[kr * LDA + mq * 4]) = v;
        }
#pragma unroll
        for (int i = 0; i < NB; ++i) {
            const int f = tid + i * 256, kr = f / (BN / 4), nq = f - kr * (BN / 4);
            bf16x4 v;
            v[0] = (__bf16)rb[i].x; v[1] = (__bf16)rb[i].y; v[2] = (__bf16)rb[i].z; v[3] = (__bf16)rb[i].w;
            *reinterpret_cast<bf16x4*>(&Bs[kr * LDB + nq * 4]) = v;
        }
    };

    if (r_begin < r_end) {
        load_tiles(r_begin);
        store_tiles();
        __syncthreads();
        for (int r0 = r_begin; r0 < r_end; r0 += BK) {
            const bool more = (r0 + BK) < r_end;
            if (more) load_tiles(r0 + BK);
            bf16x8 af[TM], bfr[TN];
#pragma unroll
            for (int i = 0; i < TM; ++i) af[i] = lds_tr_frag(As, LDA, 0, wm * WM + i * 16, lane);
#pragma unroll
            for (int j = 0; j < TN; ++j) bfr[j] = lds_tr_frag(Bs, LDB, 0, wn * WN + j * 16, lane);
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[i], bfr[j], acc[i][j], 0, 0, 0);
            __syncthreads();
            if (more) {
                store_tiles();
                __syncthreads();
            }
        }
    }
    if (do_colsum) {   // bias gradient: column sums of the A operand, reduced over this block's rows in LDS, one atomic per column
        constexpr int GR = 256 / (BM / 4);
        __syncthreads();
        float* scr = reinterpret_cast<float*>(As);
        const int mq = tid % (BM / 4), grp = tid / (BM / 4);
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            float t = 0.f;
#pragma unroll
            for (int i = 0; i < NA; ++i) t += csum[i][e];
            scr[grp * BM + mq * 4 + e] = t;
        }
        __syncthreads();
        for (int m = tid; m < BM; m += 256) {
            float t = 0.f;
#pragma unroll
            for (int q = 0; q < GR; ++q) t += scr[q * BM + m];
            if (m0 + m < g.Mstore) atomicAdd(&g.colsum_out[m0 + m], t);
        }
    }
    const int col_l = lane & 15, rgrp = (lane >> 4) * 4;
    if (g.part) {
        // split-K partial: plain stores, summed by k_tn_reduce
        float* pt = g.part + ((size_t)sp_ * (g.tiles_m * g.tiles_n) + (size_t)nt_ * g.tiles_m + mt_) * (BM * BN);
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int ml = wm * WM + i * 16 + rgrp + r;
#pragma unroll
                for (int j = 0; j < TN; ++j) pt[ml * BN + wn * WN + j * 16 + col_l] = acc[i][j][r];
            }
        return;
    }
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int m = m0 + wm * WM + i * 16 + rgrp + r;
            if (m >= g.Mstore) continue;
#pragma unroll
            for (int j = 0; j < TN; ++j) {
                const int n = n0 + wn * WN + j * 16 + col_l;
                if (n >= g.Nstore) continue;
                int nc = n;
                if (g.cw_cin > 0) { const int tap = n / g.cw_cin, ci = n - tap * g.cw_cin; nc = ci * g.cw_taps + tap; }
                atomicAdd(&g.C[(size_t)m * g.ldc + nc], acc[i][j][r]);
            }
        }
}

// Second stage of the split-K weight gradient: C[m][col(n)] += sum over splits of part[split][tile][m][n].
// A workgroup owns 32 float4 elements; its 8 thread groups each sum every 8th split (coalesced 512 B rows), the
// groups are combined through LDS and group 0 does the one read-modify-write of C.
__global__ __launch_bounds__(256) void k_tn_reduce(GemmTN g, int bm, int bn, int n_main) {
    __shared__ float4 red[8][32];
    const int tiles = g.tiles_m * g.tiles_n, per = bm * bn / 4, bn4 = bn / 4;
    if ((int)blockIdx.x >= n_main) {
        // bias gradients: one workgroup per tile sums that tile's per-split column sums (g.colpart) in a fixed order
        const int tile = blockIdx.x - n_main;
        const int mt = tile % g.tiles_m, nt = tile / g.tiles_m;
        const bool grouped = g.ngroup > 1;
        const GemmTN::Tile& gt = g.tile[grouped ? nt : 0];
        float* cs = grouped ? gt.colsum : (nt == 0 ? g.colsum_out : nullptr);
        if (!cs) return;                                      // workgroup-uniform
        const int Mst = grouped ? gt.Mstore : g.Mstore, mskip = grouped ? gt.m_skip : 0;
        float* redc = reinterpret_cast<float*>(red);
        const int col = threadIdx.x % bm, part = threadIdx.x / bm, np = 256 / bm;      // bm = 128: two partial sums per column
        float t = 0.f;
        for (int sp = part; sp < g.nsplit; sp += np) t += g.colpart[((size_t)sp * tiles + tile) * bm + col];
        redc[part * bm + col] = t;
        __syncthreads();
        if (part != 0) return;
        for (int q = 1; q < np; ++q) t += redc[q * bm + col];
        const int m = mt * bm + col;
        if (m >= mskip && m < mskip + Mst) cs[m - mskip] += t;
        return;
    }
    const int el = threadIdx.x & 31, sl = threadIdx.x >> 5;
    const int idx = blockIdx.x * 32 + el;
    const int tile = idx / per, e = idx - tile * per;
    const int ml = e / bn4, nl = (e - ml * bn4) * 4;
    const int mt = tile % g.tiles_m, nt = tile / g.tiles_m;
    const bool grouped = g.ngroup > 1;
    const GemmTN::Tile& gt = g.tile[grouped ? nt : 0];
    float* Cp = grouped ? gt.C : g.C;
    const int ldc = grouped ? gt.ldc : g.ldc;
    const int Mst = grouped ? gt.Mstore : g.Mstore;
    const int Nst = grouped ? gt.Nstore : g.Nstore;
    const int mskip = grouped ? gt.m_skip : 0, nskip = grouped ? gt.n_skip : 0;
    const int m = mt * bm + ml, n = (grouped ? 0 : nt * bn) + nl;
    const bool live = tile < tiles && m >= mskip && m < mskip + Mst && n < nskip + Nst;
    float4 s0 = make_float4(0.f, 0.f, 0.f, 0.f), s1 = s0;
    if (live) {
        const float* pt = g.part + (size_t)tile * (bm * bn) + ml * bn + nl;
        const size_t sstride = (size_t)tiles * (bm * bn);
        int sp = sl;
        for (; sp + 24 < g.nsplit; sp += 32) {          // four independent 16-byte loads in flight per thread
            const float4 a = *reinterpret_cast<const float4*>(pt + (size_t)sp * sstride);
            const float4 b = *reinterpret_cast<const float4*>(pt + (size_t)(sp + 8) * sstride);
            const float4 c = *reinterpret_cast<const float4*>(pt + (size_t)(sp + 16) * sstride);
            const float4 d = *reinterpret_cast<const float4*>(pt + (size_t)(sp + 24) * sstride);
            s0.x += a.x; s0.y += a.y; s0.z += a.z; s0.w += a.w;
            s1.x += b.x; s1.y += b.y; s1.z += b.z; s1.w += b.w;
            s0.x += c.x; s0.y += c.y; s0.z += c.z; s0.w += c.w;
            s1.x += d.x; s1.y += d.y; s1.z += d.z; s1.w += d.w;
        }
        for (; sp + 8 < g.nsplit; sp += 16) {
            const float4 a = *reinterpret_cast<const float4*>(pt + (size_t)sp * sstride);
            const float4 b = *reinterpret_cast<const float4*>(pt + (size_t)(sp + 8) * sstride);
            s0.x += a.x; s0.y += a.y; s0.z += a.z; s0.w += a.w;
            s1.x += b.x; s1.y += b.y; s1.z += b.z; s1.w += b.w;
        }
        if (sp < g.nsplit) {
            const float4 a = *reinterpret_cast<const float4*>(pt + (size_t)sp * sstride);
            s0.x += a.x; s0.y += a.y; s0.z += a.z; s0.w += a.w;
        }
    }
    red[sl][el] = make_float4(s0.x + s1.x, s0.y + s1.y, s0.z + s1.z, s0.w + s1.w);
    __syncthreads();
    if (sl != 0 || !live) return;
    float v[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int q = 0; q < 8; ++q) { const float4 t = red[q][el]; v[0] += t.x; v[1] += t.y; v[2] += t.z; v[3] += t.w; }
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const int nn = n + q;
        if (nn >= nskip + Nst) break;
        if (nn < nskip) continue;
        int nc = nn - nskip;
        if (g.cw_cin > 0) { const int tap = nn / g.cw_cin, ci = nn - tap * g.cw_cin; nc = ci * g.cw_taps + tap; }
        Cp[(size_t)(m - mskip) * ldc + nc] += v[q];
    }
}
int spair_tn_reduce(const GemmTN& g, int bm, int bn, hipStream_t s) {
    const long long n4 = (long long)g.tiles_m * g.tiles_n * bm * bn / 4;
    const int n_main = (int)((n4 + 31) / 32);
    bool any_colsum = false;
    if (g.colpart && bm <= 256 && 256 % bm == 0) {
        if (g.ngroup > 1) { for (int q = 0; q < g.ngroup; ++q) any_colsum = any_colsum || g.tile[q].colsum != nullptr; }
        else any_colsum = g.colsum_out != nullptr;
    }
    hipLaunchKernelGGL(k_tn_reduce, dim3((unsigned)(n_main + (any_colsum ? g.tiles_m * g.tiles_n : 0))), dim3(256), 0, s, g, bm, bn, n_main);
    SPAIR_CHECK_LAUNCH();
    return SPAIR_OK;
}

template <int BM, int BN>
static int launch_tn_bf16(GemmTN g, bool conv, hipStream_t s) {
    const int tiles = ceil_div(g.M, BM) * ceil_div(g.N, BN);
    int nsplit = max(1, min(ceil_div(g.R, 512), ceil_div(1024, tiles)));
    if (nsplit >= 8) nsplit = nsplit / 8 * 8;
    int rps = round_up(ceil_div(g.R, nsplit), 32);
    if (ceil_div(g.R, rps) != nsplit) nsplit = ceil_div(g.R, rps);
    g.rows_per_split = rps; g.nsplit = nsplit; g.tiles_m = ceil_div(g.M, BM); g.tiles_n = ceil_div(g.N, BN);
    dim3 grid(g.tiles_m * g.tiles_n * nsplit);
    if (g.part && (long long)grid.x * BM * BN > g.part_cap) g.part = nullptr;     // scratch too small: atomics
    if (conv) hipLaunchKernelGGL((gemm_tn_bf16_kernel<BM, BN, true>), grid, dim3(256), 0, s, g);
    else hipLaunchKernelGGL((gemm_tn_bf16_kernel<BM, BN, false>), grid, dim3(256), 0, s, g);
    SPAIR_CHECK_LAUNCH();
    if (g.part) return spair_tn_reduce(g, BM, BN, s);
    return SPAIR_OK;
}

int spair_gemm_tn_impl(GemmTN g, bool conv, int dtype, hipStream_t s) {
    if (g.M <= 0 || g.N <= 0 || g.R <= 0) return SPAIR_ERR_SHAPE;
    if (g.Mstore <= 0) g.Mstore = g.M;
    if (g.Nstore <= 0) g.Nstore = g.N;
    if ((g.M & 3) || (g.N & 3) || (g.lda & 3) || (!conv && (g.ldb & 3))) return SPAIR_ERR_ALIGN;
    if (dtype == SPAIR_BF16) {
        if (g.M > 64 && g.N > 64) return launch_tn_bf16<128, 128>(g, conv, s);
        return launch_tn_bf16<64, 64>(g, conv, s);
    }
    if (dtype != SPAIR_F32) return SPAIR_ERR_DTYPE;
    constexpr int BM = 64, BN = 64;
    const int tiles = ceil_div(g.M, BM) * ceil_div(g.N, BN);
    // aim for ~4 waves of blocks over 256 CUs, at least 256 rows per split
    int nsplit = max(1, min(ceil_div(g.R, 256), ceil_div(2048, tiles)));
    if (nsplit >= 8) nsplit = nsplit / 8 * 8;
    int rps = round_up(ceil_div(g.R, nsplit), 16);
    if (ceil_div(g.R, rps) != nsplit) nsplit = ceil_div(g.R, rps);
    g.rows_per_split = rps; g.nsplit = nsplit; g.tiles_m = ceil_div(g.M, BM); g.tiles_n = ceil_div(g.N, BN);
    dim3 grid(g.tiles_m * g.tiles_n * nsplit);
    if (conv) hipLaunchKernelGGL((gemm_tn_kernel<BM, BN, true>), grid, dim3(256), 0, s, g);
    else hipLaunchKernelGGL((gemm_tn_kernel<BM, BN, false>), grid, dim3(256), 0, s, g);
    SPAIR_CHECK_LAUNCH();
    return SPAIR_OK;
}

// column sums: out[n] += sum_r A[r*lda + n]   (bias gradients)
__global__ __launch_bounds__(256) void colsum_kernel(const float* __restrict__ A, int lda, int R, int N, int rows_per_block,
                                                     float* __restrict__ out) {
    const int n = blockIdx.x * 64 + (threadIdx.x & 63);
    const int sub = threadIdx.x >> 6;
    const int r0 = blockIdx.y * rows_per_block, r1 = min(R, r0 + rows_per_block);
    float s = 0.f;
    if (n < N)
        for (int r = r0 + sub; r < r1; r += 4) s += A[(size_t)r * lda + n];
    __shared__ float red[4][64];
    red[sub][threadIdx.x & 63] = s;
    __syncthreads();
    if (sub == 0 && n < N) atomicAdd(&out[n], red[0][threadIdx.x] + red[1][threadIdx.x] + red[2][threadIdx.x] + red[3][threadIdx.x]);
}

int spair_colsum_impl(const float* A, int lda, int R, int N, float* out, hipStream_t s) {
    if (R <= 0 || N <= 0) return SPAIR_ERR_SHAPE;
    const int rpb = max(64, ceil_div(R, 512));
    dim3 grid(ceil_div(N, 64), ceil_div(R, rpb));
    hipLaunchKernelGGL(colsum_kernel, grid, dim3(256), 0, s, A, lda, R, N, rpb, out);
    SPAIR_CHECK_LAUNCH();
    return SPAIR_OK;
}

// ---- C ABI (unit-level entry points; see include/spair_hip.h) ------------------------------
extern "C" int spair_gemm_nt(const float* A, int lda, const void* B, int ldb, float* C, int ldc, int M, int N, int K,
                             const float* bias, const float* relu_mask, int ldmask, int relu, int accumulate, int dtype,
                             void* stream) {
    GemmNT g{};
    g.A = A; g.lda = lda; g.B = B; g.ldb = ldb; g.C = C; g.ldc = ldc; g.M = M; g.N = N; g.K = K;
    g.bias = bias; g.mask = relu_mask; g.ldmask = ldmask; g.relu = relu; g.accumulate = accumulate;
    return spair_gemm_nt_impl(g, false, dtype, (hipStream_t)stream);
}

extern "C" int spair_gemm_tn(const float* A, int lda, const float* B, int ldb, float* C, int ldc, int M, int N, int R,
                             int dtype, void* stream) {
    GemmTN g{};
    g.A = A; g.lda = lda; g.B = B; g.ldb = ldb; g.C = C; g.ldc = ldc; g.M = M; g.N = N; g.R = R;
    return spair_gemm_tn_impl(g, false, dtype, (hipStream_t)stream);
}

extern "C" int spair_colsum(const float* A, int lda, int R, int N, float* out, void* stream) {
    return spair_colsum_impl(A, lda, R, N, out, (hipStream_t)stream);
}

static ConvDesc conv_from_ints(const int* p) {
    ConvDesc c;
    c.Hin = p[0]; c.Win = p[1]; c.Cin = p[2]; c.Hout = p[3]; c.Wout = p[4]; c.kh = p[5]; c.kw = p[6];
    c.sy = p[7]; c.sx = p[8]; c.dky = p[9]; c.dkx = p[10]; c.oy = p[11]; c.ox = p[12];
    return c;
}

// conv13 = {Hin,Win,Cin,Hout,Wout,kh,kw,sy,sx,dky,dkx,oy,ox}; cmap8 = {Hout,Wout,Hc,Wc,osy,osx,ooy,oox} or NULL
extern "C" int spair_gemm_nt_conv(const float* In, const int* conv13, const void* B, int ldb, float* C, int ldc, int M,
                                  int N, int K, const float* bias, const float* relu_mask, int ldmask, int relu,
                                  int accumulate, const int* cmap8, int dtype, void* stream) {
    GemmNT g{};
    g.A = In; g.lda = 0; g.B = B; g.ldb = ldb; g.C = C; g.ldc = ldc; g.M = M; g.N = N; g.K = K;
    g.bias = bias; g.mask = relu_mask; g.ldmask = ldmask; g.relu = relu; g.accumulate = accumulate;
    g.conv = conv_from_ints(conv13);
    if (cmap8) {
        g.use_cmap = 1;
        g.cmap.Hout = cmap8[0]; g.cmap.Wout = cmap8[1]; g.cmap.Hc = cmap8[2]; g.cmap.Wc = cmap8[3];
        g.cmap.osy = cmap8[4]; g.cmap.osx = cmap8[5]; g.cmap.ooy = cmap8[6]; g.cmap.oox = cmap8[7];
    }
    return spair_gemm_nt_impl(g, true, dtype, (hipStream_t)stream);
}

extern "C" int spair_gemm_tn_conv(const float* A, int lda, const float* In, const int* conv13, float* C, int ldc, int M,
                                  int N, int R, int dtype, void* stream) {
    GemmTN g{};
    g.A = A; g.lda = lda; g.B = In; g.ldb = 0; g.C = C; g.ldc = ldc; g.M = M; g.N = N; g.R = R;
    g.conv = conv_from_ints(conv13);
    return spair_gemm_tn_impl(g, true, dtype, (hipStream_t)stream);
}
