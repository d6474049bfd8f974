// Element-wise stages of the sequential per-cell encoder (reference: models.py:68-117 and the
// helpers :292-450), batched over one dependency wavefront (rows r0 .. r0+R-1) per launch.
// The matmuls between these stages are gemm.hip calls issued by engine.hip.
#include "cells.h"

__device__ __forceinline__ float clamp10(float x) { return fminf(fmaxf(x, -10.f), 10.f); }
__device__ __forceinline__ float in10(float x) { return (x >= -10.f && x <= 10.f) ? 1.f : 0.f; }
// value-preserving freeze (models.py:425): f*x + (1-f)*x
__device__ __forceinline__ float freeze_val(float f, float x) { return f * x + (1.f - f) * x; }

// ---------------------------------------------------------------------------------------------
// F1: assemble [feat | context] for the three per-cell nets (models.py:71-76,292-320)
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(128) void k_ctx_gather(CellLayout L, CellBufs P, int r0) {
    const int r = r0 + blockIdx.x;
    const int cp = r / L.B, b = r - cp * L.B;
    const int h = P.cell_h[cp], w = P.cell_w[cp];
    const float* frow = P.feat + ((size_t)(b * L.G + h) * L.G + w) * P.ld_feat;
    for (int col = threadIdx.x; col < L.F + L.CTX; col += blockDim.x) {
        float v;
        if (col < L.F) {
            v = frow[col];
        } else {
            const int s = (col - L.F) / L.REC, j = (col - L.F) - s * L.REC;
            const int nb = P.nbr[cp * 4 + s];
            v = nb >= 0 ? P.rec[((size_t)nb * L.B + b) * L.ld_rec + j] : P.edge[j];
        }
        P.Xb[(size_t)r * L.ld_xb + col] = v;
        P.Xz[(size_t)r * L.ld_x + col] = v;
        P.Xo[(size_t)r * L.ld_x + col] = v;
    }
}

// ---------------------------------------------------------------------------------------------
// F5: box latents -> box, normalised box (models.py:322-381); passthrough -> Xz
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(128) void k_box_sample(CellLayout L, CellBufs P, CellHyper H, int r0) {
    const int r = r0 + blockIdx.x;
    const int cp = r / L.B, b = r - cp * L.B;
    const float* ob = P.Ob + (size_t)r * L.ld_ob;
    for (int i = threadIdx.x; i < L.NP; i += blockDim.x) P.Xz[(size_t)r * L.ld_x + L.x_pass + i] = ob[i];
    if (threadIdx.x != 0) return;
    const int h = P.cell_h[cp], w = P.cell_w[cp];
    float z[4];
    float* st = P.stat + (size_t)r * SP_LDSTAT;
#pragma unroll
    for (int k = 0; k < 4; ++k) {  // chunk order cy, cx, height, width (models.py:330-331)
        const float mean = freeze_val(H.wheel, ob[L.ob_lat + k]);
        const float std_ = freeze_val(H.wheel, 2.f * sigmoidf_(clamp10(ob[L.ob_lat + 4 + k])));
        const float eps = P.eps_box[(((size_t)b * 4 + k) * L.G + h) * L.G + w];
        z[k] = mean + std_ * eps;
        st[ST_MU_BOX + k] = mean;
        st[ST_SD_BOX + k] = std_;
    }
    const float ryx = H.max_yx - H.min_yx, rhw = H.max_hw - H.min_hw;
    const float cell_y = ryx * sigmoidf_(clamp10(z[0])) + H.min_yx;
    const float cell_x = ryx * sigmoidf_(clamp10(z[1])) + H.min_yx;
    const float height = rhw * sigmoidf_(clamp10(z[2])) + H.min_hw;
    const float width = rhw * sigmoidf_(clamp10(z[3])) + H.min_hw;
    const float ys = height * H.anchor / H.img;
    const float xs = width * H.anchor / H.img;
    const float yt = H.cell_over_img * (cell_y + (float)h);
    const float xt = H.cell_over_img * (cell_x + (float)w);
    const float box[4] = {cell_x, cell_y, width, height};
    const float nb[4] = {xt, yt, xs, ys};
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        P.rec[(size_t)r * L.ld_rec + k] = box[k];
        P.Xz[(size_t)r * L.ld_x + L.x_box + k] = box[k];
        P.Xo[(size_t)r * L.ld_x + L.x_box + k] = box[k];
        P.nbox[(size_t)r * 4 + k] = nb[k];
        P.z_where[(((size_t)b * 4 + k) * L.G + h) * L.G + w] = nb[k];
    }
}

// ---------------------------------------------------------------------------------------------
// F10: attribute latents (models.py:83-85): no freeze on attr
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(64) void k_attr_sample(CellLayout L, CellBufs P, int r0) {
    const int r = r0 + blockIdx.x;
    const int cp = r / L.B, b = r - cp * L.B;
    const int h = P.cell_h[cp], w = P.cell_w[cp];
    const float* oe = P.Oe + (size_t)r * L.ld_oe;
    for (int j = threadIdx.x; j < L.A; j += blockDim.x) {
        const float mean = oe[j];
        const float std_ = 2.f * sigmoidf_(clamp10(oe[L.A + j]));
        const float eps = P.eps_attr[(((size_t)b * L.A + j) * L.G + h) * L.G + w];
        const float attr = mean + std_ * eps;
        P.sd_attr[(size_t)r * L.ld_rec + j] = std_;
        P.rec[(size_t)r * L.ld_rec + 4 + j] = attr;
        P.Za[(size_t)r * L.ld_rec + j] = attr;
        P.Xz[(size_t)r * L.ld_x + L.x_attr + j] = attr;
        P.Xo[(size_t)r * L.ld_x + L.x_attr + j] = attr;
    }
}

// ---------------------------------------------------------------------------------------------
// F14: depth (models.py:90-97); passthrough -> Xo
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(128) void k_depth_sample(CellLayout L, CellBufs P, CellHyper H, int r0) {
    const int r = r0 + blockIdx.x;
    const int cp = r / L.B, b = r - cp * L.B;
    const float* oz = P.Oz + (size_t)r * L.ld_oz;
    for (int i = threadIdx.x; i < L.NP; i += blockDim.x) P.Xo[(size_t)r * L.ld_x + L.x_pass + i] = oz[i];
    if (threadIdx.x != 0) return;
    const int h = P.cell_h[cp], w = P.cell_w[cp];
    const float mean = freeze_val(H.wheel, oz[L.oz_lat]);
    const float std_ = freeze_val(H.wheel, 2.f * sigmoidf_(clamp10(oz[L.oz_lat + 1])));
    const float eps = P.eps_depth[((size_t)b * L.G + h) * L.G + w];
    const float dl = mean + std_ * eps;
    const float depth = 4.f * sigmoidf_(clamp10(dl));
    float* st = P.stat + (size_t)r * SP_LDSTAT;
    st[ST_MU_DEPTH] = mean;
    st[ST_SD_DEPTH] = std_;
    P.rec[(size_t)r * L.ld_rec + 4 + L.A] = depth;
    P.Xo[(size_t)r * L.ld_x + L.x_depth] = depth;
}

// ---------------------------------------------------------------------------------------------
// F18: presence (models.py:393-411): logistic-noise relaxed Bernoulli, temperature 1
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_pres_sample(CellLayout L, CellBufs P, CellHyper H, int r0, int R) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= R) return;
    const int r = r0 + i;
    const int cp = r / L.B, b = r - cp * L.B;
    const int h = P.cell_h[cp], w = P.cell_w[cp];
    const float logit = freeze_val(H.wheel, P.Oo[(size_t)r * L.ld_oo]);
    const float lo = clamp10(logit);
    const float u = P.u_pres[((size_t)b * L.G + h) * L.G + w];
    const float noise = logf(u + 1e-9f) - logf(1.0f - u + 1e-9f);
    const float pres = sigmoidf_(lo + noise);
    P.rec[(size_t)r * L.ld_rec + L.REC - 1] = pres;
    P.z_pres[((size_t)b * L.G + h) * L.G + w] = pres;
}

// =============================================================================================
// Backward stages (reverse wavefront order)
// =============================================================================================
__device__ __forceinline__ float kl_gauss(float mu, float sd, float m, float s) {
    const float vr = (sd / s) * (sd / s);
    const float t1 = ((mu - m) / s) * ((mu - m) / s);
    return 0.5f * (vr + t1 - 1.f - logf(vr));
}

// B1: gather the gradient of this cell's record from its (up to 4) consumers, then presence.
// One wave per row.
__global__ __launch_bounds__(256) void k_bwd_pres(CellLayout L, CellBufs P, CellHyper H, int r0, int R) {
    const int lane = threadIdx.x & 63;
    const int i = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (i >= R) return;
    const int r = r0 + i;
    const int cp = r / L.B, b = r - cp * L.B;
    const float gl = *P.gloss;
    const float ks = H.kl_scale * gl;
    // --- record gradient from consumers' context columns
    float gsum_pres = 0.f;
    for (int j = lane; j < L.REC; j += 64) {
        float g = 0.f;
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            const int q = P.cons[cp * 4 + s];
            if (q < 0) continue;
            const size_t qr = (size_t)q * L.B + b;
            const int col = L.x_ctx + s * L.REC + j;
            g += P.dXb[qr * L.ld_xb + col] + P.dXz[qr * L.ld_x + col] + P.dXo[qr * L.ld_x + col];
        }
        P.grec[(size_t)r * L.ld_rec + j] = g;
        if (j == L.REC - 1) gsum_pres = g;
    }
    // --- sum of Gaussian KL elements (they are masked by z_pres: models.py:175-176)
    const float* st = P.stat + (size_t)r * SP_LDSTAT;
    float kl = 0.f;
    for (int j = lane; j < L.A; j += 64)
        kl += kl_gauss(P.Oe[(size_t)r * L.ld_oe + j], P.sd_attr[(size_t)r * L.ld_rec + j], H.prior_mean[4], H.prior_std[4]);
    if (lane < 4) kl += kl_gauss(st[ST_MU_BOX + lane], st[ST_SD_BOX + lane], H.prior_mean[lane], H.prior_std[lane]);
    if (lane == 4) kl += kl_gauss(st[ST_MU_DEPTH], st[ST_SD_DEPTH], H.prior_mean[5], H.prior_std[5]);
    kl = wave_reduce_sum(kl);
    gsum_pres = wave_reduce_sum(gsum_pres);
    if (lane == 0) {
        const float z = P.rec[(size_t)r * L.ld_rec + L.REC - 1];
        const float pz = st[ST_PZ];
        // d/dz of z*(log(z+e)-log(pz+e)) + (1-z)*(log(1-z+e)-log(1-pz+e))   (models.py:223-226)
        const float e = 1e-9f;
        const float dkl = logf(z + e) - logf(pz + e) + z / (z + e) - logf(1.f - z + e) + logf(1.f - pz + e) -
                          (1.f - z) / (1.f - z + e);
        const float g = gsum_pres + P.g_pres_r[r] + ks * (kl + dkl);
        const float logit = P.Oo[(size_t)r * L.ld_oo];  // freeze_val(logit) == logit for wheel in {0,1}
        const float lf = freeze_val(H.wheel, logit);
        P.dOo[(size_t)r * L.ld_oo] = g * z * (1.f - z) * in10(lf) * (1.f - H.wheel);
    }
}

// B5: depth; also route the passthrough gradient dXo[pass] -> dOz[pass]
__global__ __launch_bounds__(128) void k_bwd_depth(CellLayout L, CellBufs P, CellHyper H, int r0) {
    const int r = r0 + blockIdx.x;
    const int cp = r / L.B, b = r - cp * L.B;
    float* doz = P.dOz + (size_t)r * L.ld_oz;
    for (int i = threadIdx.x; i < L.NP; i += blockDim.x) doz[i] = P.dXo[(size_t)r * L.ld_x + L.x_pass + i];
    if (threadIdx.x != 0) return;
    const int h = P.cell_h[cp], w = P.cell_w[cp];
    const float ks = H.kl_scale * (*P.gloss);
    const float* st = P.stat + (size_t)r * SP_LDSTAT;
    const float mu = st[ST_MU_DEPTH], sd = st[ST_SD_DEPTH];
    const float eps = P.eps_depth[((size_t)b * L.G + h) * L.G + w];
    const float dl = mu + sd * eps;
    const float s = sigmoidf_(clamp10(dl));
    const float zp = P.rec[(size_t)r * L.ld_rec + L.REC - 1];
    const float g_depth = P.grec[(size_t)r * L.ld_rec + 4 + L.A] + P.dXo[(size_t)r * L.ld_x + L.x_depth] + P.g_depth_r[r];
    const float g_dl = g_depth * 4.f * s * (1.f - s) * in10(dl);
    const float m = H.prior_mean[5], ps = H.prior_std[5];
    const float g_mu = (g_dl + ks * zp * (mu - m) / (ps * ps)) * (1.f - H.wheel);
    const float g_sd = (g_dl * eps + ks * zp * (sd / (ps * ps) - 1.f / sd)) * (1.f - H.wheel);
    const float ls = P.Oz[(size_t)r * L.ld_oz + L.oz_lat + 1];
    const float sl = sigmoidf_(clamp10(ls));
    doz[L.oz_lat] = g_mu;
    doz[L.oz_lat + 1] = g_sd * 2.f * sl * (1.f - sl) * in10(ls);
}

// B9: attributes -> gradient of the encoder output
__global__ __launch_bounds__(64) void k_bwd_attr(CellLayout L, CellBufs P, CellHyper H, int r0) {
    const int r = r0 + blockIdx.x;
    const int cp = r / L.B, b = r - cp * L.B;
    const int h = P.cell_h[cp], w = P.cell_w[cp];
    const float ks = H.kl_scale * (*P.gloss);
    const float zp = P.rec[(size_t)r * L.ld_rec + L.REC - 1];
    const float m = H.prior_mean[4], ps = H.prior_std[4];
    for (int j = threadIdx.x; j < L.A; j += blockDim.x) {
        const float g = P.grec[(size_t)r * L.ld_rec + 4 + j] + P.dXz[(size_t)r * L.ld_x + L.x_attr + j] +
                        P.dXo[(size_t)r * L.ld_x + L.x_attr + j] + P.g_attr_r[(size_t)r * L.ld_rec + j];
        const float mu = P.Oe[(size_t)r * L.ld_oe + j];
        const float sd = P.sd_attr[(size_t)r * L.ld_rec + j];
        const float eps = P.eps_attr[(((size_t)b * L.A + j) * L.G + h) * L.G + w];
        const float g_mu = g + ks * zp * (mu - m) / (ps * ps);
        const float g_sd = g * eps + ks * zp * (sd / (ps * ps) - 1.f / sd);
        const float ls = P.Oe[(size_t)r * L.ld_oe + L.A + j];
        const float sl = sigmoidf_(clamp10(ls));
        P.dOe[(size_t)r * L.ld_oe + j] = g_mu;
        P.dOe[(size_t)r * L.ld_oe + L.A + j] = g_sd * 2.f * sl * (1.f - sl) * in10(ls);
    }
}

// B14: box; also route dXz[pass] -> dOb[pass]
__global__ __launch_bounds__(128) void k_bwd_box(CellLayout L, CellBufs P, CellHyper H, int r0) {
    const int r = r0 + blockIdx.x;
    const int cp = r / L.B, b = r - cp * L.B;
    float* dob = P.dOb + (size_t)r * L.ld_ob;
    for (int i = threadIdx.x; i < L.NP; i += blockDim.x) dob[i] = P.dXz[(size_t)r * L.ld_x + L.x_pass + i];
    if (threadIdx.x != 0) return;
    const int h = P.cell_h[cp], w = P.cell_w[cp];
    const float ks = H.kl_scale * (*P.gloss);
    const float zp = P.rec[(size_t)r * L.ld_rec + L.REC - 1];
    const float* st = P.stat + (size_t)r * SP_LDSTAT;
    // total gradient of (xt, yt, xs, ys): glimpse STN + renderer
    float gn[4], gb[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        gn[k] = P.g_nbox_stn[(size_t)r * 4 + k] + P.g_nbox_r[(size_t)r * 4 + k];
        gb[k] = P.grec[(size_t)r * L.ld_rec + k] + P.dXz[(size_t)r * L.ld_x + L.x_box + k] +
                P.dXo[(size_t)r * L.ld_x + L.x_box + k];  // (cell_x, cell_y, width, height)
    }
    const float ryx = H.max_yx - H.min_yx, rhw = H.max_hw - H.min_hw;
    // gradient wrt the 4 squashed quantities in latent order (cy, cx, height, width)
    const float gq[4] = {
        (gb[1] + gn[1] * H.cell_over_img) * ryx,   // cell_y
        (gb[0] + gn[0] * H.cell_over_img) * ryx,   // cell_x
        (gb[3] + gn[3] * H.anchor / H.img) * rhw,  // height
        (gb[2] + gn[2] * H.anchor / H.img) * rhw,  // width
    };
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const float mu = st[ST_MU_BOX + k], sd = st[ST_SD_BOX + k];
        const float eps = P.eps_box[(((size_t)b * 4 + k) * L.G + h) * L.G + w];
        const float z = mu + sd * eps;
        const float s = sigmoidf_(clamp10(z));
        const float g_z = gq[k] * s * (1.f - s) * in10(z);
        const float m = H.prior_mean[k], ps = H.prior_std[k];
        const float g_mu = (g_z + ks * zp * (mu - m) / (ps * ps)) * (1.f - H.wheel);
        const float g_sd = (g_z * eps + ks * zp * (sd / (ps * ps) - 1.f / sd)) * (1.f - H.wheel);
        const float ls = P.Ob[(size_t)r * L.ld_ob + L.ob_lat + 4 + k];
        const float sl = sigmoidf_(clamp10(ls));
        dob[L.ob_lat + k] = g_mu;
        dob[L.ob_lat + 4 + k] = g_sd * 2.f * sl * (1.f - sl) * in10(ls);
    }
}

// After the loop: d feat = sum of the three nets' feature-column gradients (row order (b,h,w)),
// and the learned edge element's gradient = sum over every out-of-grid context slot.
__global__ __launch_bounds__(256) void k_dfeat_edge(CellLayout L, CellBufs P, int rows_per_block, float* __restrict__ gedge) {
    const int rbeg = blockIdx.x * rows_per_block, rend = min(L.N, rbeg + rows_per_block);
    const int col = threadIdx.x;  // 0 .. F+CTX-1 handled in strides
    for (int c = col; c < L.F + L.CTX; c += blockDim.x) {
        float acc = 0.f;
        const int s = c >= L.F ? (c - L.F) / L.REC : -1;
        for (int r = rbeg; r < rend; ++r) {
            const int cp = r / L.B, b = r - cp * L.B;
            const float v = P.dXb[(size_t)r * L.ld_xb + c] + P.dXz[(size_t)r * L.ld_x + c] + P.dXo[(size_t)r * L.ld_x + c];
            if (s < 0) {
                const int h = P.cell_h[cp], w = P.cell_w[cp];
                P.dfeat[((size_t)(b * L.G + h) * L.G + w) * P.ld_feat + c] = v;
            } else if (P.nbr[cp * 4 + s] < 0) {
                acc += v;
            }
        }
        if (s >= 0 && acc != 0.f) atomicAdd(&gedge[(c - L.F) - s * L.REC], acc);
    }
}

// ---- tables: wavefront order, neighbours, consumers -----------------------------------------
__global__ __launch_bounds__(1024) void k_init_tables(int G, int* cell_h, int* cell_w, int* cidx, int* nbr, int* cons, int* diag_start) {
    __shared__ int dstart[3 * 32 + 2];
    const int T = 3 * G - 2, HW = G * G;
    if (threadIdx.x == 0) {
        int c = 0;
        for (int t = 0; t < T; ++t) {
            dstart[t] = c;
            const int hlo = max(0, (t - (G - 1) + 1) / 2), hhi = min(G - 1, t / 2);   // 0 <= t-2h < G
            c += max(0, hhi - hlo + 1);
        }
        dstart[T] = c;
    }
    __syncthreads();
    for (int t = threadIdx.x; t <= T; t += blockDim.x) diag_start[t] = dstart[t];
    // wavefront index of cell (h,w): cells of diagonal t = 2h+w are ordered by ascending h
    auto cp_of = [&](int h, int w) {
        const int t = 2 * h + w;
        const int hlo = max(0, (t - (G - 1) + 1) / 2);
        return dstart[t] + (h - hlo);
    };
    const int dh[4] = {-1, -1, -1, 0}, dw[4] = {-1, 0, 1, -1};   // UL, U, UR, L (models.py:297-304)
    for (int k = threadIdx.x; k < HW; k += blockDim.x) {
        const int h = k / G, w = k - h * G;
        const int c = cp_of(h, w);
        cell_h[c] = h; cell_w[c] = w; cidx[k] = c;
        for (int s = 0; s < 4; ++s) {
            const int nh = h + dh[s], nw = w + dw[s];
            nbr[c * 4 + s] = (nh >= 0 && nh < G && nw >= 0 && nw < G) ? cp_of(nh, nw) : -1;
            const int qh = h - dh[s], qw = w - dw[s];   // the cell that sees (h,w) in slot s
            cons[c * 4 + s] = (qh >= 0 && qh < G && qw >= 0 && qw < G) ? cp_of(qh, qw) : -1;
        }
    }
}

// ---- host launchers -------------------------------------------------------------------------
int cells_init_tables(int G, int* cell_h, int* cell_w, int* cidx, int* nbr, int* cons, int* diag_start, hipStream_t s) {
    hipLaunchKernelGGL(k_init_tables, dim3(1), dim3(1024), 0, s, G, cell_h, cell_w, cidx, nbr, cons, diag_start);
    SPAIR_CHECK_LAUNCH();
    return SPAIR_OK;
}
#define LAUNCH(k, grid, block, ...)                                        \
    do {                                                                   \
        hipLaunchKernelGGL(k, dim3(grid), dim3(block), 0, s, __VA_ARGS__); \
        SPAIR_CHECK_LAUNCH();                                              \
        return SPAIR_OK;                                                   \
    } while (0)

int cells_ctx_gather(const CellLayout& L, const CellBufs& P, int r0, int R, hipStream_t s) { LAUNCH(k_ctx_gather, R, 128, L, P, r0); }
int cells_box_sample(const CellLayout& L, const CellBufs& P, const CellHyper& H, int r0, int R, hipStream_t s) { LAUNCH(k_box_sample, R, 128, L, P, H, r0); }
int cells_attr_sample(const CellLayout& L, const CellBufs& P, int r0, int R, hipStream_t s) { LAUNCH(k_attr_sample, R, 64, L, P, r0); }
int cells_depth_sample(const CellLayout& L, const CellBufs& P, const CellHyper& H, int r0, int R, hipStream_t s) { LAUNCH(k_depth_sample, R, 128, L, P, H, r0); }
int cells_pres_sample(const CellLayout& L, const CellBufs& P, const CellHyper& H, int r0, int R, hipStream_t s) { LAUNCH(k_pres_sample, ceil_div(R, 256), 256, L, P, H, r0, R); }
int cells_bwd_pres(const CellLayout& L, const CellBufs& P, const CellHyper& H, int r0, int R, hipStream_t s) { LAUNCH(k_bwd_pres, ceil_div(R, 4), 256, L, P, H, r0, R); }
int cells_bwd_depth(const CellLayout& L, const CellBufs& P, const CellHyper& H, int r0, int R, hipStream_t s) { LAUNCH(k_bwd_depth, R, 128, L, P, H, r0); }
int cells_bwd_attr(const CellLayout& L, const CellBufs& P, const CellHyper& H, int r0, int R, hipStream_t s) { LAUNCH(k_bwd_attr, R, 64, L, P, H, r0); }
int cells_bwd_box(const CellLayout& L, const CellBufs& P, const CellHyper& H, int r0, int R, hipStream_t s) { LAUNCH(k_bwd_box, R, 128, L, P, H, r0); }
int cells_dfeat_edge(const CellLayout& L, const CellBufs& P, float* gedge, hipStream_t s) {
    const int rpb = 64;
    LAUNCH(k_dfeat_edge, ceil_div(L.N, rpb), 256, L, P, rpb, gedge);
}
