// Element-wise stages of the sequential per-cell encoder (reference: models.py:68-117 and the
// helpers :292-450), batched over one dependency wavefront (rows r0 .. r0+R-1) per launch.
// The matmuls between these stages are gemm.hip calls issued by engine.hip.
#include "cells.h"
#include "cell_math.h"

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
            const int nb = P.nbr[cp * L.NB + s];
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
    float eps[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) eps[k] = P.eps_box[(((size_t)b * 4 + k) * L.G + h) * L.G + w];
    const BoxFwd o = box_forward(ob + L.ob_lat, eps, H, h, w);
    float* st = P.stat + (size_t)r * SP_LDSTAT;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        st[ST_MU_BOX + k] = o.mu[k];
        st[ST_SD_BOX + k] = o.sd[k];
        P.rec[(size_t)r * L.ld_rec + k] = o.box[k];
        P.Xz[(size_t)r * L.ld_x + L.x_box + k] = o.box[k];
        P.Xo[(size_t)r * L.ld_x + L.x_box + k] = o.box[k];
        P.nbox[(size_t)r * 4 + k] = o.nbox[k];
        P.z_where[(((size_t)b * 4 + k) * L.G + h) * L.G + w] = o.nbox[k];
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
        const float eps = P.eps_attr[(((size_t)b * L.A + j) * L.G + h) * L.G + w];
        float std_, attr;
        attr_forward(mean, oe[L.A + j], eps, std_, attr);
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
    const float eps = P.eps_depth[((size_t)b * L.G + h) * L.G + w];
    float mean, std_, depth;
    depth_forward(oz[L.oz_lat], oz[L.oz_lat + 1], eps, H, mean, std_, depth);
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
    const float u = P.u_pres[((size_t)b * L.G + h) * L.G + w];
    const float pres = pres_forward(P.Oo[(size_t)r * L.ld_oo], u, H);
    P.rec[(size_t)r * L.ld_rec + L.REC - 1] = pres;
    P.z_pres[((size_t)b * L.G + h) * L.G + w] = pres;
}

// =============================================================================================
// Backward stages (reverse wavefront order)
// =============================================================================================
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
        for (int s = 0; s < L.NB; ++s) {
            const int q = P.cons[cp * L.NB + s];
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
        P.dOo[(size_t)r * L.ld_oo] = pres_backward(gsum_pres + P.g_pres_r[r], z, st[ST_PZ], kl, P.Oo[(size_t)r * L.ld_oo], ks, H);
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
    const float eps = P.eps_depth[((size_t)b * L.G + h) * L.G + w];
    const float zp = P.rec[(size_t)r * L.ld_rec + L.REC - 1];
    const float g_depth = P.grec[(size_t)r * L.ld_rec + 4 + L.A] + P.dXo[(size_t)r * L.ld_x + L.x_depth] + P.g_depth_r[r];
    depth_backward(g_depth, st[ST_MU_DEPTH], st[ST_SD_DEPTH], P.Oz[(size_t)r * L.ld_oz + L.oz_lat + 1], eps, zp, ks, H, doz[L.oz_lat],
                   doz[L.oz_lat + 1]);
}

// B9: attributes -> gradient of the encoder output
__global__ __launch_bounds__(64) void k_bwd_attr(CellLayout L, CellBufs P, CellHyper H, int r0) {
    const int r = r0 + blockIdx.x;
    const int cp = r / L.B, b = r - cp * L.B;
    const int h = P.cell_h[cp], w = P.cell_w[cp];
    const float ks = H.kl_scale * (*P.gloss);
    const float zp = P.rec[(size_t)r * L.ld_rec + L.REC - 1];
    for (int j = threadIdx.x; j < L.A; j += blockDim.x) {
        const float g = P.grec[(size_t)r * L.ld_rec + 4 + j] + P.dXz[(size_t)r * L.ld_x + L.x_attr + j] +
                        P.dXo[(size_t)r * L.ld_x + L.x_attr + j] + P.g_attr_r[(size_t)r * L.ld_rec + j];
        const float eps = P.eps_attr[(((size_t)b * L.A + j) * L.G + h) * L.G + w];
        attr_backward(g, P.Oe[(size_t)r * L.ld_oe + j], P.sd_attr[(size_t)r * L.ld_rec + j], P.Oe[(size_t)r * L.ld_oe + L.A + j], eps, zp, ks,
                      H, P.dOe[(size_t)r * L.ld_oe + j], P.dOe[(size_t)r * L.ld_oe + L.A + j]);
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
    float eps[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) eps[k] = P.eps_box[(((size_t)b * 4 + k) * L.G + h) * L.G + w];
    box_backward(gn, gb, st + ST_MU_BOX, st + ST_SD_BOX, eps, P.Ob + (size_t)r * L.ld_ob + L.ob_lat + 4, zp, ks, H, dob + L.ob_lat);
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
            } else if (P.nbr[cp * L.NB + s] < 0) {
                acc += v;
            }
        }
        if (s >= 0 && acc != 0.f) atomicAdd(&gedge[(c - L.F) - s * L.REC], acc);
    }
}

// ---- tables: wavefront order, neighbours, consumers -----------------------------------------
// LB = N_LOOKBACK: cell (h, w) reads rows h-LB..h, columns w-LB..w+LB of the cells before it, so t = (LB+1) h + w is a dependency order
// (its latest input, (h-1, w+LB), has t - 1); LB = 1 gives the 3G-2 anti-diagonals of UL, U, UR, L.
__global__ __launch_bounds__(1024) void k_init_tables(int G, int LB, int* cell_h, int* cell_w, int* cidx, int* nbr, int* cons, int* diag_start) {
    __shared__ int dstart[5 * 32 + 2];
    const int S = LB + 1, T = S * (G - 1) + G, HW = G * G, NB = 2 * LB * S;
    if (threadIdx.x == 0) {
        int c = 0;
        for (int t = 0; t < T; ++t) {
            dstart[t] = c;
            const int hlo = max(0, (t - (G - 1) + S - 1) / S), hhi = min(G - 1, t / S);   // 0 <= t - S h < G
            c += max(0, hhi - hlo + 1);
        }
        dstart[T] = c;
    }
    __syncthreads();
    for (int t = threadIdx.x; t <= T; t += blockDim.x) diag_start[t] = dstart[t];
    // wavefront index of cell (h,w): cells of wavefront t = S h + w are ordered by ascending h
    auto cp_of = [&](int h, int w) {
        const int t = S * h + w;
        const int hlo = max(0, (t - (G - 1) + S - 1) / S);
        return dstart[t] + (h - hlo);
    };
    for (int k = threadIdx.x; k < HW; k += blockDim.x) {
        const int h = k / G, w = k - h * G;
        const int c = cp_of(h, w);
        cell_h[c] = h; cell_w[c] = w; cidx[k] = c;
        // slots in the reference's order (models.py:297-304): rows -LB..0, columns -LB..LB, without the current cell and its right side
        // (LB = 1: UL, U, UR, L)
        int s = 0;
        for (int dh = -LB; dh <= 0; ++dh)
            for (int dw = -LB; dw <= (dh ? LB : -1); ++dw, ++s) {
                const int nh = h + dh, nw = w + dw;
                nbr[c * NB + s] = (nh >= 0 && nh < G && nw >= 0 && nw < G) ? cp_of(nh, nw) : -1;
                const int qh = h - dh, qw = w - dw;   // the cell that sees (h,w) in slot s
                cons[c * NB + s] = (qh >= 0 && qh < G && qw >= 0 && qw < G) ? cp_of(qh, qw) : -1;
            }
    }
}

// ---- host launchers -------------------------------------------------------------------------
int cells_init_tables(int G, int LB, int* cell_h, int* cell_w, int* cidx, int* nbr, int* cons, int* diag_start, hipStream_t s) {
    if (G > 32 || LB < 1 || LB > 3) return SPAIR_ERR_UNSUPPORTED;
    hipLaunchKernelGGL(k_init_tables, dim3(1), dim3(1024), 0, s, G, LB, cell_h, cell_w, cidx, nbr, cons, diag_start);
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
