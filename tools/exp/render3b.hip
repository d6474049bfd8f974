// K6 backward on the MATRIX CORES (reference: autograd of models.py:485-547 / stn(inverse=True) modules.py:256-269; the forward is
// render3.hip).  One wave per object, no LDS, no atomics, nothing synchronised across waves.
//
// The forward sampled  V_c = Wy . S_c . Wx^T  per (object, 16 x 16 pixel tile) with fp16 hat weights (render3.h).  The backward needs, per pixel of the object's footprint, the three sampled channels again (to form the adjoints of the
// composite), their derivatives along x and y (for d z_where), and the transpose of the sampling applied to the adjoints (d sprite):
//     T_c  = S_c . Wx^T          TD_c = S_c . Dx^T                      (first products, once per 16-pixel COLUMN of tiles; Dx = d Wx / d s:
//     V_c  = Wy . T_c            DX_c = Wy . TD_c      DY_c = Dy . T_c   -1 on the lower tap, +1 on the upper one)
//     adjoints (VALU, 4 pixels per lane in the accumulator layout) -> d z_where sums in fp32 registers
//     TB_c += adj_c^T . Wy       (K = the tile's 16 pixel rows: the accumulator layout of V_c IS the A operand of v_mfma_f32_16x16x16_bf16)
//     O_c  += Wx^T . TB_c        (once per column of tiles; K = its 16 pixel columns, TB's accumulator layout is the B operand as it stands)
// O_c[u][v] is the gradient of sprite texel (row v, column u) with a lane holding ONE sprite row v and four consecutive columns: the
// d-logits leave as whole 16-byte pieces straight from registers, and the sprite values the sigmoid' needs are one 16-byte load each.
// The sprite itself is read once, as the forward reads it: row pieces straight into the A-operand layout, resident for the whole object.
// V_c is bit-identical to what the forward composited (same weight fragments, same fp16 rounding of T), so the (a g - pre) term of the
// importance adjoint cancels exactly as it must; the adjoints and the hat weights of the two transposed products are bf16 (as in
// k_render_bwd2: the d-logits are stored as bf16 anyway), d z_where is fp32 sums of fp32 products of the sampled derivatives.
//
// NOT on the training step's path (round 5): against k_render_bwd2 (one wave per object too, sampling on VALU from an LDS copy of the
// sprite: 2,190 VALU + 204 LDS instructions per object at configs[1], 11.4 waves per CU, VALU pipe 93 % busy, 0.240 ms) this kernel issues
// 1,420 VALU instructions + ~170 MFMAs per object but holds 248 registers -- 2 waves per SIMD, VALU pipe 75 % busy -- and takes 0.226-0.235
// ms; and although its d z_where is closer to the oracle's than k_render_bwd2's bound (5e-4 of the largest element), the step's conv_0
// weight gradient on the c1_b8_step1001 fixture falls from cosine 0.998 to 0.988 with it (every other tensor and fixture unchanged).
// Kept as a unit-level entry (spair_render_bwd16m, tests/test_kernels_gpu.py::test_render16m_bwd_vs_oracle); DESIGN.md has the account.
#include <stdlib.h>
#include "render3.h"

typedef short r3b_s4 __attribute__((ext_vector_type(4)));
typedef __bf16 r3b_b2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ r3b_s4 r3b_bf4(const f32x4& v) {
    const r3b_b2 lo = {(__bf16)v[0], (__bf16)v[1]}, hi = {(__bf16)v[2], (__bf16)v[3]};
    return __builtin_bit_cast(r3b_s4, u32x2_t{__builtin_bit_cast(unsigned, lo), __builtin_bit_cast(unsigned, hi)});
}

template <int IP2>
__global__ __launch_bounds__(64, 2) void k_render_bwd_mma(const void* __restrict__ S, unsigned s_bytes, const RenderObjRec* __restrict__ orec,
                                                          const RenderCullRec* __restrict__ crec, const RenderBwdRec* __restrict__ brec,
                                                          const float2* __restrict__ aux, const float* __restrict__ gloss,
                                                          __bf16* __restrict__ dlogits, float* __restrict__ dnbox, float* __restrict__ dpres,
                                                          float* __restrict__ ddepth, int ld_g, int B, int HW, int I, float obj_scale,
                                                          float alpha_scale) {
    constexpr int P = R3_P;
    const int lane = threadIdx.x, l15 = lane & 15, q = lane >> 4;
    // (sample, object): consecutive workgroup ids walk the samples, so with B % 8 == 0 every object of sample b lands on XCD b % 8
    const int b = blockIdx.x, k = blockIdx.y;
    const int r = k * B + b;
    const size_t idx = (size_t)b * HW + k;
    // the sprite: 16-row tile t, lane (row 16t + l15, k-group q) = bytes 32q .. 32q+31 of that row (rows >= P read zeros)
    const __amdgpu_buffer_rsrc_t srs = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(S), (short)0, (int)s_bytes, 0x00020000);
    const unsigned ob = (unsigned)r * (unsigned)R3_SPRB;
    const unsigned voff = ob + (unsigned)(l15 * R3_ROWB + 32 * q), voffh = voff + (q == 3 ? 0u : 16u);
    const bool ok1 = l15 + 16 < P;
    R3Frag f0, f1;
    f0.lo = __builtin_amdgcn_raw_buffer_load_b128(srs, (int)voff, 0, 0);
    f0.hi = __builtin_amdgcn_raw_buffer_load_b128(srs, (int)voffh, 0, 0);
    f1.lo = __builtin_amdgcn_raw_buffer_load_b128(srs, (int)(ok1 ? voff + 16u * R3_ROWB : BUF_OOB), 0, 0);
    f1.hi = __builtin_amdgcn_raw_buffer_load_b128(srs, (int)(ok1 ? voffh + 16u * R3_ROWB : BUF_OOB), 0, 0);
    // the records (wave-uniform)
    const RenderObjRec o = orec[idx];
    const uint4 pc = *reinterpret_cast<const uint4*>(crec + idx);
    const RenderBwdRec pb = brec[idx];
    const int PX0 = __builtin_amdgcn_readfirstlane((int)(pc.z & 0xffff)), PX1 = __builtin_amdgcn_readfirstlane((int)(pc.z >> 16));
    const int PY0 = __builtin_amdgcn_readfirstlane((int)(pc.w & 0xffff)), PY1 = __builtin_amdgcn_readfirstlane((int)(pc.w >> 16));
    const float pr = o.pres, mscale = o.mscale, gl = *gloss;
    const float inv_I = 1.f / (float)I;
    // the per-pixel record (dBCE/dpre / D, pre) through a descriptor: 32-bit offsets, and a load that cannot turn conditional
    const __amdgpu_buffer_rsrc_t ars = __builtin_amdgcn_make_buffer_rsrc(const_cast<float2*>(aux + (size_t)b * I * I), (short)0,
                                                                         (int)((unsigned)I * (unsigned)I * 8u), 0x00020000);
    const float dgy_ = pb.ay * 2.f * inv_I;          // the normalised grid coordinate's step per pixel row
    const unsigned arow = (unsigned)I * 8u;           // bytes per pixel row of the record

    // operand constants (render3.hip): first product B operand, lane (column, k-group q): texel columns u = 8q + j;
    // second product A operand, lane (row, k-group q): sprite rows {4q .. 4q+3, 16+4q .. 16+4q+3}
    r3_f2 cx[4], cy[4];
#pragma unroll
    for (int jp = 0; jp < 4; ++jp) {
        const int u = 8 * q + 2 * jp;
        cx[jp] = r3_f2{u < P ? -(float)u : -1.0e4f, u + 1 < P ? -(float)(u + 1) : -1.0e4f};
        const int v = (jp < 2 ? 4 * q + 2 * jp : 16 + 4 * q + 2 * (jp - 2));
        cy[jp] = r3_f2{-(float)v, -(float)(v + 1)};
    }
    // The transposed products need the hat weights with the SPRITE index on the lanes (A operand of O += Wx^T . TB: lane u, four pixel
    // columns; B operand of TB += adj^T . Wy: lane v, four pixel rows).  Those are the fragments above transposed, and the matrix core
    // transposes: W . Sel with a 0/1 selection matrix Sel[k-slot][n] = (slot's sprite index == 16 t + n) comes out in the accumulator
    // layout -- lane n, rows 4q .. 4q+3 -- which is that operand up to the bf16 conversion (one MFMA instead of ~17 VALU per fragment).
    r3_h8 selx[2], sely[2];
#pragma unroll
    for (int t = 0; t < 2; ++t) {
        // x fragments: slot j of k-group q is texel column 8q + j -> the one non-zero slot of lane (n = l15, q) is j = (16t + l15) - 8q;
        // y fragments: slot j is sprite row 4q + j (j < 4) / 16 + 4q + (j - 4) -> non-zero where q == l15 / 4, at j = 4t + l15 % 4
        const int jx = 16 * t + l15 - 8 * q, jy = 4 * t + (l15 & 3);
        const bool okx = jx >= 0 && jx < 8, oky = q == (l15 >> 2);
        const unsigned px_ = okx ? 0x3c00u << (16 * (jx & 1)) : 0u, py_ = oky ? 0x3c00u << (16 * (jy & 1)) : 0u;
        u32x4_t dxv, dyv;
#pragma unroll
        for (int i = 0; i < 4; ++i) { dxv[i] = (jx >> 1) == i ? px_ : 0u; dyv[i] = (jy >> 1) == i ? py_ : 0u; }
        selx[t] = __builtin_bit_cast(r3_h8, dxv);
        sely[t] = __builtin_bit_cast(r3_h8, dyv);
    }

    r3_h8 sg[2], sa[2], sm[2];
    r3_split(f0, o.mfloor, sg[0], sa[0], sm[0]);
    r3_split(f1, ok1 ? o.mfloor : 0u, sg[1], sa[1], sm[1]);

    const f32x4 z = {0.f, 0.f, 0.f, 0.f};
    f32x4 O[3][2][2];
#pragma unroll
    for (int c = 0; c < 3; ++c)
#pragma unroll
        for (int ut = 0; ut < 2; ++ut)
#pragma unroll
            for (int vt = 0; vt < 2; ++vt) O[c][ut][vt] = z;
    float g_tx = 0.f, g_ty = 0.f, g_xs = 0.f, g_ys = 0.f;

    for (int tx = PX0; tx <= PX1; tx += 16) {
        // ---- this column of tiles: Wx / Dx, the first products, the second transposed product's A operand
        const float basex = rf_base<0, IP2>(min(tx + l15, I - 1), I, inv_I);
        const float sxq = fmaf(o.Ax, basex, o.Bx);
        const float gnx = fmaf(pb.ax, basex, pb.bx);                    // the normalised grid coordinate: d s / d (scale) = g * cgx

        r3_h8 wx, dx;
        r3_hat8d(sxq, cx, wx, dx);
        u32x4_t th[3], tdh[3];
        {
            const r3_h8* ch[3] = {sg, sa, sm};
#pragma unroll
            for (int c = 0; c < 3; ++c) {
                const f32x4 t0 = __builtin_amdgcn_mfma_f32_16x16x32_f16(ch[c][0], wx, z, 0, 0, 0);
                const f32x4 t1 = __builtin_amdgcn_mfma_f32_16x16x32_f16(ch[c][1], wx, z, 0, 0, 0);
                const f32x4 d0 = __builtin_amdgcn_mfma_f32_16x16x32_f16(ch[c][0], dx, z, 0, 0, 0);
                const f32x4 d1 = __builtin_amdgcn_mfma_f32_16x16x32_f16(ch[c][1], dx, z, 0, 0, 0);
                th[c] = u32x4_t{r3_pk(t0[0], t0[1]), r3_pk(t0[2], t0[3]), r3_pk(t1[0], t1[1]), r3_pk(t1[2], t1[3])};
                tdh[c] = u32x4_t{r3_pk(d0[0], d0[1]), r3_pk(d0[2], d0[3]), r3_pk(d1[0], d1[1]), r3_pk(d1[2], d1[3])};
            }
        }
        // Wx with the texel column on the lanes (see selx)
        const r3b_s4 wxb0 = r3b_bf4(__builtin_amdgcn_mfma_f32_16x16x32_f16(wx, selx[0], z, 0, 0, 0));
        const r3b_s4 wxb1 = r3b_bf4(__builtin_amdgcn_mfma_f32_16x16x32_f16(wx, selx[1], z, 0, 0, 0));
        // pixels beyond the image must not contribute (their coordinate is the last column's again); pixels inside the image but outside
        // the footprint need no mask: every sampled value and derivative is 0 there and the transposed products' weights too
        const float xokf = tx + l15 < I ? gl : 0.f;                    // (the loss scale rides on the column mask)
        const unsigned xoff = (unsigned)min(tx + l15, I - 1) * 8u;

        f32x4 TB[3][2];
#pragma unroll
        for (int c = 0; c < 3; ++c) { TB[c][0] = z; TB[c][1] = z; }
        // the tiles' per-pixel loss gradients -- pixel (tx + l15, ty + 4q + i) -- are fetched one tile ahead (rows beyond the image are
        // beyond the descriptor's range: they read zeros)
        auto ld_aux = [&](int ty_, u32x2_t (&a)[4]) {
            const unsigned aoff = (unsigned)(ty_ + 4 * q) * arow + xoff;
            a[0] = __builtin_amdgcn_raw_buffer_load_b64(ars, (int)aoff, 0, 0);
            a[1] = __builtin_amdgcn_raw_buffer_load_b64(ars, (int)aoff, (int)arow, 0);
            a[2] = __builtin_amdgcn_raw_buffer_load_b64(ars, (int)aoff, (int)(2u * arow), 0);
            a[3] = __builtin_amdgcn_raw_buffer_load_b64(ars, (int)aoff, (int)(3u * arow), 0);
        };
        u32x2_t avn[4];
        ld_aux(PY0, avn);
        for (int ty = PY0; ty <= PY1; ty += 16) {
            u32x2_t av[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) av[i] = avn[i];
            ld_aux(ty + 16, avn);
            // ---- Wy / Dy of pixel row ty + l15, second products
            const float basey = rf_base<0, IP2>(min(ty + l15, I - 1), I, inv_I);
            const float syq = fmaf(o.Ay, basey, o.By);
            r3_h8 wy, dy;
            r3_hat8d(syq, cy, wy, dy);
            f32x4 V[3], DX[3], DY[3];
#pragma unroll
            for (int c = 0; c < 3; ++c) {
                V[c] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wy, __builtin_bit_cast(r3_h8, th[c]), z, 0, 0, 0);
                DX[c] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wy, __builtin_bit_cast(r3_h8, tdh[c]), z, 0, 0, 0);
                DY[c] = __builtin_amdgcn_mfma_f32_16x16x32_f16(dy, __builtin_bit_cast(r3_h8, th[c]), z, 0, 0, 0);
            }
            // the accumulator layout's pixel rows ty + 4q + j: coordinates for d z_where and for the transposed product's B operand
            float gny[4];
            gny[0] = fmaf(pb.ay, rf_base<0, IP2>(ty + 4 * q, I, inv_I), pb.by);
#pragma unroll
            for (int j = 1; j < 4; ++j) gny[j] = fmaf((float)j, dgy_, gny[0]);
            // Wy with the sprite row on the lanes (see sely)
            const r3b_s4 wyb0 = r3b_bf4(__builtin_amdgcn_mfma_f32_16x16x32_f16(wy, sely[0], z, 0, 0, 0));
            const r3b_s4 wyb1 = r3b_bf4(__builtin_amdgcn_mfma_f32_16x16x32_f16(wy, sely[1], z, 0, 0, 0));
            // ---- adjoints of the composite (models.py:511-540): num = sum g a (m + 1e-9), den = sum m
            f32x4 adg, ada, adm;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const float g = V[0][i], a = V[1][i] * pr, m = V[2][i] * mscale;
                const float go = __uint_as_float(av[i].x) * xokf;           // dBCE/dpre / D (x the incoming loss gradient)
                const float tt = go * (m + 1e-9f);
                const float d_g = tt * a, d_a = tt * g;                     // wrt grey, wrt (alpha * pres)
                const float d_m = go * (a * g - __uint_as_float(av[i].y));
                const float dap = d_a * pr, dms = d_m * mscale;
                const float g_sx = d_g * DX[0][i] + dap * DX[1][i] + dms * DX[2][i];       // d / d(source x), texel units
                const float g_sy = d_g * DY[0][i] + dap * DY[1][i] + dms * DY[2][i];
                g_tx += g_sx; g_xs = fmaf(g_sx, gnx, g_xs);
                g_ty += g_sy; g_ys = fmaf(g_sy, gny[i], g_ys);
                adg[i] = d_g; ada[i] = d_a; adm[i] = d_m;
            }
            // ---- TB_c[px][v] += sum_py adj_c[py][px] . Wy[py][v]
            const r3b_s4 ab[3] = {r3b_bf4(adg), r3b_bf4(ada), r3b_bf4(adm)};
#pragma unroll
            for (int c = 0; c < 3; ++c) {
                TB[c][0] = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(ab[c], wyb0, TB[c][0], 0, 0, 0);
                TB[c][1] = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(ab[c], wyb1, TB[c][1], 0, 0, 0);
            }
        }
        // ---- O_c[u][v] += sum_px Wx[px][u] . TB_c[px][v]
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            const r3b_s4 tb0 = r3b_bf4(TB[c][0]), tb1 = r3b_bf4(TB[c][1]);
            O[c][0][0] = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(wxb0, tb0, O[c][0][0], 0, 0, 0);
            O[c][0][1] = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(wxb0, tb1, O[c][0][1], 0, 0, 0);
            O[c][1][0] = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(wxb1, tb0, O[c][1][0], 0, 0, 0);
            O[c][1][1] = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(wxb1, tb1, O[c][1][1], 0, 0, 0);
        }
    }

    // ---- epilogue: per texel sigmoid' and the logit scales (models.py:485-492), d pres / d depth; this lane: sprite row 16 vt + l15,
    // columns 16 ut + 4q .. + 3 = one 16-byte piece of the sprite and of the d-logits
    float g_pr = 0.f, g_s2a = 0.f;
    char* dst = reinterpret_cast<char*>(dlogits + (size_t)r * ld_g);
    const unsigned short flo = (unsigned short)(o.mfloor & 0xffffu);
    // the four sprite pieces first (the sprite's last use in the step), the arithmetic behind them
    u32x4_t svv[2][2];
#pragma unroll
    for (int vt = 0; vt < 2; ++vt)
#pragma unroll
        for (int ut = 0; ut < 2; ++ut) {
            const int v = 16 * vt + l15, u0 = 16 * ut + 4 * q;
            svv[vt][ut] = __builtin_amdgcn_raw_buffer_load_b128(srs, (int)((v < P && u0 < P) ? ob + (unsigned)(v * P + u0) * 4u : BUF_OOB), 0, 2);
        }
#pragma unroll
    for (int vt = 0; vt < 2; ++vt)
#pragma unroll
        for (int ut = 0; ut < 2; ++ut) {
            const int v = 16 * vt + l15, u0 = 16 * ut + 4 * q;
            const unsigned svi[4] = {svv[vt][ut][0], svv[vt][ut][1], svv[vt][ut][2], svv[vt][ut][3]};
            u32x4_t outv;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                // sigmoid'(logit) = s (1 - s) of both channels in one packed fp16 fma (the d-logits are bf16: 11 bits are plenty); the
                // products with the fp32 sums mix the fp16 factor in (v_fma_mix_f32)
                const r3_h2 sh = __builtin_bit_cast(r3_h2, svi[i]);
                const r3_h2 fac = sh - sh * sh;
                const float s0 = O[0][ut][vt][i], s1 = O[1][ut][vt][i], s2 = O[2][ut][vt][i];
                const bool act = (unsigned short)(svi[i] >> 16) >= flo;         // the importance of this texel is alpha * pd, not the floor
                const float s2s = act ? s2 : 0.f;
                g_pr = fmaf(s1, (float)sh.y, g_pr);                             // (texels beyond the sprite read as zeros: no contribution)
                g_s2a = fmaf(s2s, (float)sh.y, g_s2a);
                const float ox = (s0 * obj_scale) * (float)fac.x;
                const float oy = (fmaf(s1, pr, s2s * mscale) * alpha_scale) * (float)fac.y;
                const r3b_b2 ob2 = {(__bf16)ox, (__bf16)oy};
                outv[i] = __builtin_bit_cast(unsigned, ob2);
            }
            if (v < P && u0 < P) __builtin_nontemporal_store(outv, reinterpret_cast<u32x4_t*>(dst + (unsigned)(v * P + u0) * 4u));
        }
    const float mult = 0.5f * (float)P;
    const float cgx = -mult * pb.ax, cgy = -mult * pb.ay;            // d(source coord)/d(t) incl. the unnormalisation
    g_tx = wave_reduce_sum(g_tx) * cgx; g_ty = wave_reduce_sum(g_ty) * cgy;
    g_xs = wave_reduce_sum(g_xs) * cgx; g_ys = wave_reduce_sum(g_ys) * cgy;
    g_pr = wave_reduce_sum(g_pr); g_s2a = wave_reduce_sum(g_s2a);
    if (lane == 0) {
        *reinterpret_cast<float4*>(dnbox + (size_t)r * 4) = make_float4(2.f * g_tx, 2.f * g_ty, g_xs, g_ys);
        dpres[r] = g_pr + g_s2a * o.depth;
        ddepth[r] = g_s2a * pr;
    }
}

// SPAIR_ERR_UNSUPPORTED: the caller keeps k_render_bwd2
int render_bwd_mma(const void* S16, int ld_s, const void* rec, const float* aux, const float* gloss, void* dlogits16, float* dnbox,
                   float* dpres, float* ddepth, int ld_g, int B, int HW, int I, int P, int ac, float obj_scale, float alpha_scale,
                   hipStream_t s) {
    if (!rec || !render_prep_supported(HW, I, P, ac) || ld_s != R3_P * R3_P * 2 || (ld_g & 7) || HW > 65535) return SPAIR_ERR_UNSUPPORTED;
    if ((unsigned long long)B * HW * R3_SPRB >= 0xfffffff0ull - 64) return SPAIR_ERR_UNSUPPORTED;
    if ((reinterpret_cast<uintptr_t>(S16) & 15) || (reinterpret_cast<uintptr_t>(dlogits16) & 15) || (reinterpret_cast<uintptr_t>(rec) & 15))
        return SPAIR_ERR_UNSUPPORTED;
    const RenderObjRec* orec = reinterpret_cast<const RenderObjRec*>(rec);
    const RenderCullRec* crec = reinterpret_cast<const RenderCullRec*>(render_rec_cull(rec, B, HW));
    const RenderBwdRec* brec = reinterpret_cast<const RenderBwdRec*>(render_rec_bwd(rec, B, HW));
    const unsigned s_bytes = (unsigned)((size_t)B * HW * R3_SPRB);
    const dim3 grid(B, HW), block(64);
    if ((I & (I - 1)) == 0)
        hipLaunchKernelGGL((k_render_bwd_mma<1>), grid, block, 0, s, S16, s_bytes, orec, crec, brec, reinterpret_cast<const float2*>(aux), gloss,
                           reinterpret_cast<__bf16*>(dlogits16), dnbox, dpres, ddepth, ld_g, B, HW, I, obj_scale, alpha_scale);
    else
        hipLaunchKernelGGL((k_render_bwd_mma<0>), grid, block, 0, s, S16, s_bytes, orec, crec, brec, reinterpret_cast<const float2*>(aux), gloss,
                           reinterpret_cast<__bf16*>(dlogits16), dnbox, dpres, ddepth, ld_g, B, HW, I, obj_scale, alpha_scale);
    SPAIR_CHECK_LAUNCH();
    return SPAIR_OK;
}
