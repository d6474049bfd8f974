// KL terms and the loss (reference: _compute_KL models.py:169-262, _build_loss :544-563).
//   K8  k_count_kl : the sequential count-prior Bernoulli KL -- one wave per sample walks the HW
//                    cells in row-major order with the (HW+1)-bin count distribution in LDS.
//   K7  k_gauss_kl : the six presence-masked Gaussian KL sums (deterministic two-stage reduce).
//   k_loss_finalize: BCE partials + KL sums -> loss_out[0..8].
#include "cells.h"

#define KL_MAXBINS 1025   // HW+1 for G <= 32

// (wave_lds_fence(), common.h: orders a wave's own LDS writes before its later reads of other lanes' values -- one wave per sample here)
// One wave per sample; the (HW+1)-bin count distribution lives in REGISTERS (bin e = k*64 + lane, NBR bins per lane, held as PAIRS for
// the packed fp32 pipe), so a step is register math + two DPP wave reductions (with the bins in LDS each step paid two LDS round trips per bin: 2.2 us per step at
// G = 32, where this kernel, not the decoder beside it, set the forward's length).
// 1 / x: v_rcp_f32 + one Newton step (<= 1 ulp for normal x; the IEEE division sequence is ~12 instructions of the ~130 a cell costs)
__device__ __forceinline__ float kl_rcp(float x) {
    const float r = __builtin_amdgcn_rcpf(x);
    return fmaf(fmaf(-x, r, 1.f), r, r);
}
// Wave sum through the gfx9 row-broadcast DPP forms: four adds give every lane its 16-lane row's sum, row_bcast:15 / row_bcast:31 carry the
// sums up the rows, lane 63 holds the total -- 6 DPP adds + 1 v_readlane instead of 4 + 4 readlanes + 3 adds (wave_reduce_sum, common.h)
__device__ __forceinline__ float kl_wave_sum(float v) {
    v = dpp_add_<0xB1>(v);
    v = dpp_add_<0x4E>(v);
    v = dpp_add_<0x141>(v);
    v = dpp_add_<0x140>(v);
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x142, 0xa, 0xf, false));      // row_bcast:15 -> rows 1, 3
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x143, 0xc, 0xf, false));      // row_bcast:31 -> rows 2, 3
    return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 63));
}

// ---- K8: the sequential count-prior KL (models.py:186-257) ----------------------------------------------------------------------------------
// One wave per sample walks the HW cells in row-major order.  The count distribution lives in REGISTERS in RELATIVE form: bin j = (total
// object count) - (objects seen so far), j = k * 64 + lane in register k.  In that form
//   * the factor of a bin is q_j = j / rem (rem = cells left), no subtraction of the running count and no clamp: the bins outside [0, rem]
//     are exactly the ones that hold no mass (a cell that is ON multiplies bin 0 by q_0 = 0 and then every bin moves down by one -- one
//     wavefront-shift DPP move per register, on the few steps where a cell is on; a cell that is OFF multiplies bin `rem` by 1 - 1 = 0);
//   * the number of registers that can hold mass, floor(rem / 64) + 1, depends on the step only, not on the data: the loop is a sequence of
//     phases with NA = NBR, NBR - 1, ..., 1 active registers, each compiled for its own NA -- half the bin work on average, no branches.
// (Absolute bins with clamps cost 9 instructions per pair of bins on all NBR registers of every step: 0.46 ms at 32 x 32 cells, where this
//  kernel is exposed behind the chain)  The bins run as PAIRS on the packed fp32 pipe; only the last active register can hold
// bins past `rem`: its factor is clamped to 1, and 1 / rem is rounded so that the bin AT `rem` gets exactly 1 (kl_step) -- bins past `rem` stay
// exactly empty.  (Round 5's form left residues of 1 - rem * (1 / rem) there and, in the phases with an odd register count, ran the dropped
// register through the packed arithmetic unclamped: the residues grew geometrically through runs of present cells -- NaN loss at mean
// z_pres ~ 0.7.  kl_phase also empties the dropped register at every hand-over.)
// WPB waves (= samples) per workgroup.  The kernel runs on the helper stream beside the decoder and the forward renderer; its waves are single
// dependent chains that take issue slots from whatever shares their SIMD.  Four per workgroup (one per SIMD of a quarter of the CUs) is the
// measured optimum at 16 x 16 cells, B = 256: 16 / 8 / 4 / 2 / 1 waves per workgroup = step 3.289 / 3.246 / 3.221 / 3.274 / 3.336 ms -- packed, the
// kernel itself becomes the longest thing beside the decoder (0.24 ms); spread, every CU's renderer waves share a SIMD with it.
typedef float kl_f2 __attribute__((ext_vector_type(2)));
struct KlState { float znext, count_on; };

template <int NA, int NP>
__device__ __forceinline__ void kl_step(kl_f2 (&c2)[NP], const kl_f2 (&e2)[NP], const float* zs, float* pzs, int i, int HW, int lane, float& znext) {
    constexpr int PA = (NA + 1) / 2;                 // active pairs
    const float z = znext;
    znext = zs[min(i + 1, HW - 1)];
    // q_j = j * (1 / rem) is within 1-2 ulp of j / rem.  The one bin where that matters is j = rem: the reference's rem / rem is EXACTLY 1, so
    // a cell that is off empties that bin (factor 1 - 1) and a cell that is on keeps it whole.  1 / rem is therefore rounded UP when
    // rem * (1 / rem) < 1 (exact test through the fma): the product is then >= 1 and the clamp below makes it 1 -- no residue is ever
    // left in a bin past `rem` (before: ~1e-7 of the largest bin per absent cell, re-normalised upwards to ~1e-5 of p_z in dense grids).
    const float rem = (float)(HW - i), r0 = kl_rcp(rem);
    const float inv_rem = fmaf(rem, r0, -1.f) < 0.f ? __builtin_bit_cast(float, __builtin_bit_cast(int, r0) + 1) : r0;
    const bool on = rintf(z) != 0.f;                 // torch.round: half to even
    // the factor of c: q where the cell is on, 1 - q where it is off = sa * q + sb with wave-uniform (sa, sb) -- exact either way
    const float sa = on ? 1.f : -1.f, sb = on ? 0.f : 1.f;
    const kl_f2 ir2 = {inv_rem, inv_rem}, sa2 = {sa, sa}, sb2 = {sb, sb};
    kl_f2 pz2 = {0.f, 0.f}, np2 = {0.f, 0.f};
#pragma unroll
    for (int k = 0; k < PA; ++k) {
        kl_f2 q = e2[k] * ir2;
        if (k == (NA - 1) / 2) {                      // the last active register: bins past `rem`
            if ((NA - 1) & 1) q.y = fminf(q.y, 1.f); else q.x = fminf(q.x, 1.f);
        }
        pz2 += c2[k] * q;
        const kl_f2 v = (sa2 * q + sb2) * c2[k];
        c2[k] = v;
        np2 += v;
    }
    const float pz = kl_wave_sum(pz2.x + pz2.y);
    const float np = fmaxf(kl_wave_sum(np2.x + np2.y), 1e-6f);
    const float inv_np = kl_rcp(np);
    const kl_f2 in2 = {inv_np, inv_np};
#pragma unroll
    for (int k = 0; k < PA; ++k) c2[k] = c2[k] * in2;
    if (lane == 0) pzs[i] = pz;
    if (on) {                                         // wave-uniform: every bin moves down by one (bin 0, now empty, drops out)
        auto shl1 = [](float cur, float from_next) {      // lane l <- lane l + 1; lane 63 <- lane 0 of the next register
            return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(__builtin_bit_cast(int, from_next), __builtin_bit_cast(int, cur), 0x130, 0xf, 0xf, false));
        };
        auto lane0 = [](float v) { return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 0)); };
#pragma unroll
        for (int r = 0; r < NA; ++r) {
            const float nxt = r + 1 < NA ? lane0((r + 1) & 1 ? c2[(r + 1) / 2].y : c2[(r + 1) / 2].x) : 0.f;
            if (r & 1) c2[r / 2].y = shl1(c2[r / 2].y, nxt); else c2[r / 2].x = shl1(c2[r / 2].x, nxt);
        }
    }
}
template <int NA, int NP>
__device__ __forceinline__ void kl_phase(kl_f2 (&c2)[NP], const kl_f2 (&e2)[NP], const float* zs, float* pzs, int HW, int lane, float& znext) {
    // steps with floor((HW - i) / 64) + 1 == NA
    // Register NA dropped out at this phase change (its bins are all past `rem` now: no mass, but a residue of 1 - rem * (1 / rem) may sit
    // in it).  In an odd phase it is the upper half of the last PAIR and still runs through the packed arithmetic with q = j / rem > 1:
    // left alone, the residue is multiplied by q on every present cell, is never shifted out, and is summed into p_z and the normaliser
    // -- it grows geometrically through runs of present cells until p_z > 1 (NaN loss at mean z_pres ~ 0.7).  Empty it here: 0 * q stays 0.
    if constexpr (NA & 1) c2[NA / 2].y = 0.f;
    const int lo = max(0, HW - 64 * NA + 1), hi = min(HW - 1, HW - 64 * (NA - 1));
    for (int i = lo; i <= hi; ++i) kl_step<NA, NP>(c2, e2, zs, pzs, i, HW, lane, znext);
    if constexpr (NA > 1) kl_phase<NA - 1, NP>(c2, e2, zs, pzs, HW, lane, znext);
}

template <int NBR, int WPB>
__global__ __launch_bounds__(WPB * 64) void k_count_kl(CellLayout L, CellBufs P, float prior_prob, float* __restrict__ klp) {
    extern __shared__ float kl_sh[];          // [WPB][2][HW]: z_pres in cell order, p_z
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int b = blockIdx.x * WPB + wave;
    if (b >= L.B) return;
    const int HW = L.HW, NB = HW + 1;
    float* zs = kl_sh + (size_t)wave * 2 * HW;
    float* pzs = zs + HW;
    // all of this sample's z_pres in row-major cell order: no global access inside the sequential loop
    for (int i = lane; i < HW; i += 64) zs[i] = P.rec[((size_t)P.cidx[i] * L.B + b) * L.ld_rec + L.REC - 1];
    // geometric count distribution (1-p) p^k, normalised (models.py:190-193)
    float c[NBR], ef[NBR];
    float part = 0.f;
#pragma unroll
    for (int k = 0; k < NBR; ++k) {
        const int e = k * 64 + lane;
        ef[k] = (float)e;
        c[k] = e < NB ? (1.f - prior_prob) * powf(prior_prob, (float)e) : 0.f;     // bins past HW stay 0
        part += c[k];
    }
    const float norm0 = wave_reduce_sum(part);
#pragma unroll
    for (int k = 0; k < NBR; ++k) c[k] = c[k] / norm0;
    wave_lds_fence();
    constexpr int NP = (NBR + 1) / 2;
    kl_f2 c2[NP], e2[NP];
#pragma unroll
    for (int k = 0; k < NP; ++k) {
        c2[k] = kl_f2{c[2 * k], 2 * k + 1 < NBR ? c[2 * k + 1] : 0.f};
        e2[k] = kl_f2{ef[2 * k], 2 * k + 1 < NBR ? ef[2 * k + 1] : 0.f};
    }
    float znext = zs[0];
    kl_phase<NBR, NP>(c2, e2, zs, pzs, HW, lane, znext);
    wave_lds_fence();
    // Bernoulli KL per cell (models.py:223-226) and the p_z map the backward pass needs
    float klsum = 0.f;
    for (int i = lane; i < HW; i += 64) {
        const float z = zs[i], pz = pzs[i], e9 = 1e-9f;
        P.stat[((size_t)P.cidx[i] * L.B + b) * SP_LDSTAT + ST_PZ] = pz;
        klsum += z * (logf(z + e9) - logf(pz + e9)) + (1.f - z) * (logf(1.f - z + e9) - logf(1.f - pz + e9));
    }
    klsum = wave_reduce_sum(klsum);
    if (lane == 0) klp[b] = klsum;
}

__device__ __forceinline__ float kl_gauss_l(float mu, float sd, float m, float s) {
    const float vr = (sd / s) * (sd / s);
    const float t1 = ((mu - m) / s) * ((mu - m) / s);
    return 0.5f * (vr + t1 - 1.f - logf(vr));
}

// partial[blockIdx][6]: sum over this block's rows of z_pres * KL for (cy,cx,height,width,attr,depth).
// A wave takes whole rows: lane j < A the attribute element j, lanes A .. A+3 the four box latents, lane A+4 the depth -- coalesced row
// reads, no index divisions, one set of reductions per block (the element-parallel form with an integer division per element and six block
// reductions took 43 us on the forward's critical path behind the count-prior KL).
__global__ __launch_bounds__(256) void k_gauss_kl(CellLayout L, CellBufs P, CellHyper H, int rows_per_block, float* __restrict__ partial) {
    __shared__ float red[4][6];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int rbeg = blockIdx.x * rows_per_block, rend = min(L.N, rbeg + rows_per_block);
    const int A = L.A;
    const int kind = lane < A ? 4 : lane < A + 4 ? lane - A : lane == A + 4 ? 5 : -1;      // which of the six sums this lane feeds
    const float pm = kind >= 0 ? H.prior_mean[kind] : 0.f, ps = kind >= 0 ? H.prior_std[kind] : 1.f;
    float acc = 0.f;
#pragma unroll 4
    for (int r = rbeg + wave; r < rend; r += 4) {
        const float zp = P.rec[(size_t)r * L.ld_rec + L.REC - 1];
        const float* st = P.stat + (size_t)r * SP_LDSTAT;
        float mu = pm, sd = ps;                                                              // (idle lanes: KL = 0)
        if (kind == 4) { mu = P.Oe[(size_t)r * L.ld_oe + lane]; sd = P.sd_attr[(size_t)r * L.ld_rec + lane]; }
        else if (kind == 5) { mu = st[ST_MU_DEPTH]; sd = st[ST_SD_DEPTH]; }
        else if (kind >= 0) { mu = st[ST_MU_BOX + kind]; sd = st[ST_SD_BOX + kind]; }
        acc += zp * kl_gauss_l(mu, sd, pm, ps);
    }
    // lanes of one kind are contiguous: the attribute lanes by a masked wave sum, the other five by their own lane
    const float attr = wave_reduce_sum(kind == 4 ? acc : 0.f);
    if (lane == 0) red[wave][4] = attr;
    if (kind >= 0 && kind != 4) red[wave][kind] = acc;
    __syncthreads();
    if (threadIdx.x < 6) partial[blockIdx.x * 6 + threadIdx.x] = (red[0][threadIdx.x] + red[1][threadIdx.x]) + (red[2][threadIdx.x] + red[3][threadIdx.x]);
}

// loss_out[0]=total, [1]=BCE, [2..7]=Gaussian KLs * kl_scale, [8]=presence KL * kl_scale
__global__ __launch_bounds__(256) void k_loss_finalize(const float* __restrict__ bce_partial, int n_bce, const float* __restrict__ kl_partial,
                                                       int n_kl, const float* __restrict__ klp, int B, float kl_scale, float beta,
                                                       float* __restrict__ loss_out, const int* __restrict__ failed,
                                                       int* __restrict__ status, int* __restrict__ status_host) {
    // all eight sums in one pass: every load of a thread is issued before the first add, one LDS reduction for the lot
    // (eight block reductions in sequence, each behind its own dependent loads, took 23 us between the renderer's two passes)
    __shared__ float red[4][8];
    float acc[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    for (int i0 = threadIdx.x; i0 < n_bce; i0 += 8 * 256) {
        float v[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) { const int i = i0 + e * 256; v[e] = i < n_bce ? bce_partial[i] : 0.f; }
#pragma unroll
        for (int e = 0; e < 8; ++e) acc[0] += v[e];
    }
    for (int i0 = threadIdx.x; i0 < n_kl; i0 += 4 * 256) {
        float v[4][6];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const int i = i0 + e * 256;
#pragma unroll
            for (int k = 0; k < 6; ++k) v[e][k] = i < n_kl ? kl_partial[i * 6 + k] : 0.f;
        }
#pragma unroll
        for (int e = 0; e < 4; ++e)
#pragma unroll
            for (int k = 0; k < 6; ++k) acc[1 + k] += v[e][k];
    }
    for (int i = threadIdx.x; i < B; i += 256) acc[7] += klp[i];
#pragma unroll
    for (int k = 0; k < 8; ++k) acc[k] = wave_reduce_sum(acc[k]);
    if ((threadIdx.x & 63) == 0) {
#pragma unroll
        for (int k = 0; k < 8; ++k) red[threadIdx.x >> 6][k] = acc[k];
    }
    __syncthreads();
    float bce = 0.f, kls[7];
    if (threadIdx.x == 0) {
        bce = (red[0][0] + red[1][0]) + (red[2][0] + red[3][0]);
        for (int k = 0; k < 7; ++k) kls[k] = ((red[0][1 + k] + red[1][1 + k]) + (red[2][1 + k] + red[3][1 + k])) * kl_scale;
    }
    if (threadIdx.x == 0) {
        float kl_total = 0.f;
        for (int k = 0; k < 7; ++k) { loss_out[2 + k] = kls[k]; kl_total += kls[k]; }
        loss_out[1] = bce;
        // `failed`: the band-split chain's sticky time-out word (chain.h) -- a step whose hand-off timed out announces itself as a NaN loss
        const int timed_out = (failed && *failed) ? 1 : 0;
        const float total = bce + beta * kl_total;
        loss_out[0] = loss_out[9] = timed_out ? __builtin_nanf("") : total;
        // SpairStep.status / status_host (include/spair_hip.h): a failed or non-finite step announces itself without a host synchronisation
        // (every loss term feeds `total`, so one test covers the nine)
        const int bits = timed_out | (fabsf(total) <= 3.402823466e38f ? 0 : 2);
        if (status) { status[1] = bits; if (bits) status[0] |= bits; }
        if (status_host && bits) *(volatile int*)status_host = bits;
    }
}

int loss_count_kl(const CellLayout& L, const CellBufs& P, float prior_prob, float* klp, hipStream_t s) {
    if (L.HW + 1 > KL_MAXBINS) return SPAIR_ERR_UNSUPPORTED;
#ifndef KL_WPB_SMALL
#define KL_WPB_SMALL 4
#endif
    if (L.HW + 1 <= 5 * 64) {
        constexpr int W = KL_WPB_SMALL;
        hipLaunchKernelGGL((k_count_kl<5, W>), dim3(ceil_div(L.B, W)), dim3(W * 64), (size_t)W * 2 * L.HW * sizeof(float), s, L, P, prior_prob, klp);
    } else if (L.HW + 1 <= 9 * 64) {
        hipLaunchKernelGGL((k_count_kl<9, 4>), dim3(ceil_div(L.B, 4)), dim3(256), (size_t)4 * 2 * L.HW * sizeof(float), s, L, P, prior_prob, klp);
    } else {
        hipLaunchKernelGGL((k_count_kl<17, 4>), dim3(ceil_div(L.B, 4)), dim3(256), (size_t)4 * 2 * L.HW * sizeof(float), s, L, P, prior_prob, klp);
    }
    SPAIR_CHECK_LAUNCH();
    return SPAIR_OK;
}
#define KL_ROWS_PER_BLOCK 16      // four rows per wave, all their loads in flight together
int loss_gauss_kl_blocks(const CellLayout& L) { return ceil_div(L.N, KL_ROWS_PER_BLOCK); }
int loss_gauss_kl(const CellLayout& L, const CellBufs& P, const CellHyper& H, float* partial, hipStream_t s) {
    if (L.A + 5 > 64) return SPAIR_ERR_UNSUPPORTED;        // one lane per latent element of a row
    hipLaunchKernelGGL(k_gauss_kl, dim3(loss_gauss_kl_blocks(L)), dim3(256), 0, s, L, P, H, KL_ROWS_PER_BLOCK, partial);
    SPAIR_CHECK_LAUNCH();
    return SPAIR_OK;
}
int loss_finalize(const float* bce_partial, int n_bce, const float* kl_partial, int n_kl, const float* klp, int B, float kl_scale,
                  float beta, float* loss_out, const int* failed, int* status, int* status_host, hipStream_t s) {
    hipLaunchKernelGGL(k_loss_finalize, dim3(1), dim3(256), 0, s, bce_partial, n_bce, kl_partial, n_kl, klp, B, kl_scale, beta, loss_out, failed,
                       status, status_host);
    SPAIR_CHECK_LAUNCH();
    return SPAIR_OK;
}
