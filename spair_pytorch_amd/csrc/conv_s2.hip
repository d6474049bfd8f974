#include <numeric>
// Patch-resident forward of the backbone's strided convolutions in the bf16 step (reference: modules.py:59-64 -- Conv2d(128, 128, 4, stride 2)
// + ReLU): conv_1 and conv_2 at 128 -> 128 channels, 4x4 kernel, stride 2, NHWC bf16 in / out.
//
// Why another conv kernel.  As an implicit GEMM (gemm16.hip) conv_1 moves, per 128-row tile and K step, 16 KB of gathered input and 16 KB of
// weights from L2 into LDS: 2.4 GB per launch, and the launch takes exactly what 2.4 GB take at the ~11.5 TB/s the chip's L2 -> LDS paths
// deliver to this access pattern (0.21 ms) -- the kernel is bound by operand bytes, not by the matrix cores.  Two changes cut the bytes 2.3x:
//   * PATCH-RESIDENT input.  The 16 taps of a 4x4 / stride-2 kernel fall into 4 parity classes (ky % 2, kx % 2); the 4 taps of a class read
//     the same input SUB-LATTICE S[yy][xx] = in[2 yy + py][2 xx + px] at (y + dy, x + dx), dy, dx in {0, 1}.  A workgroup stages, per (class,
//     64-channel half), the sub-lattice patch its 256 output pixels touch ONCE (<= 11 rows x 35 pixels x 128 B at conv_1) and reads the A
//     fragments of all 4 taps out of it: 45 KB per 4 K steps instead of 4 x 32 KB.
//   * 256-row tiles (8 computing waves, 4 x 2, 64 x 64 each, + 4 loader waves): the weight tile of a K step is shared by twice the rows.
// K runs in the tap-parity order the prepared weights already have (gemm.h, GemmNT::ktab: class, half, tap).
// Pipeline: weight tiles through a ring of 3 x 16 KB, three K steps ahead; the next stage's patch into the other of two patch buffers; everything
// by LDS-DMA (buffer_load ... lds, XOR swizzle applied on the source side) behind counted s_waitcnt vmcnt; the two wave groups of a workgroup
// alternate between a fragment-load phase and an MFMA phase (ping-pong, see the kernel).
#include "common.h"
#include "gemm.h"

namespace {

constexpr int CP_BM = 256, CP_C = 128;           // rows per workgroup, channels (in = out)
constexpr int CP_PPX = 392;                      // patch capacity in pixels (49 DMA pieces of 8 pixels x 128 B)
constexpr int CP_PATCH_B = CP_PPX * 128;         // 50,176 B per patch buffer
constexpr int CP_BT_B = 128 * 128;               // weight tile [128 n][64 k] bf16
constexpr int CP_NPIECE_W = 5;                   // patch pieces per GB wave per C phase (3 phases x 4 waves x 5 >= 49)
constexpr int CP_LDS = 3 * CP_BT_B + 2 * CP_PATCH_B + 1024;      // + 1 KiB that the surplus (out-of-range) patch pieces are pointed at
constexpr int CP_LDC = CP_C + 16;                // epilogue staging row (bf16 elements): 288 B = 32 mod 64, conflict-free 16-byte row reads (see chain.hip)

struct ConvPatchArgs {
    const u16* in; const u16* wf; const float* bias; u16* out;
    unsigned char* mask;                         // optional: sign bits of the output, one byte per (pixel, 8 channels) [M][16] (the next layer's dgrad gate)
    int B, Hin, Hout, M, K;                      // M = B * Hout * Hout, K = 16 * 128
    int tpi;                                     // 0: tiles of 256 consecutive rows of the whole batch; > 0: tiles per image (a tile never crosses an image)
};

template <int W>
__device__ __forceinline__ void cp_wait() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(W) : "memory"); }

__global__ __launch_bounds__(768, 3) void k_conv_s2k4_patch(ConvPatchArgs a) {
    extern __shared__ __attribute__((aligned(16))) char cp_sm[];
    char* bt = cp_sm;                            // [3][128][64] bf16, chunk ^ (row & 7)
    char* patch = cp_sm + 3 * CP_BT_B;           // [2][CP_PPX][64] bf16, chunk ^ (pixel & 7)
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int q = lane >> 4, r16 = lane & 15;
    const int Hout = a.Hout, Ws = Hout + 1, Hs = Hout + 1;
    int m0, mend;
    if (a.tpi) {
        const int img = blockIdx.x / a.tpi, HH = Hout * Hout;
        m0 = img * HH + (blockIdx.x - img * a.tpi) * CP_BM;
        mend = min(m0 + CP_BM, (img + 1) * HH);
    } else {
        m0 = blockIdx.x * CP_BM;
        mend = min(m0 + CP_BM, a.M);
    }
    const int mlast = mend - 1;
    // extended sub-lattice row of output row g = (b, y): E = g + b (every image owns Hout + 1 sub-lattice rows).  The patch is the LINEAR
    // window of sub-lattice pixels from the first tile pixel's tap (0, 0) to the last tile pixel's tap (1, 1) -- not whole rows E0 .. E1:
    // 256 + Ws + 1 pixels plus one row per image boundary crossed, whatever the image side (whole rows refused conv_1 at 256 x 256)
    const int g0 = m0 / Hout, g1 = mlast / Hout;
    const int L0 = (g0 + g0 / Hout) * Ws + (m0 - g0 * Hout);
    const int npx = (g1 + g1 / Hout + 1) * Ws + (mlast - g1 * Hout) + 1 - L0 + 1;          // <= CP_PPX (checked by the launcher)

    // ---- roles.  Waves 0..7 are CONSUMERS (4 x 2, a 64 x 64 output tile each; two per SIMD, so that one's LDS round trip at the top of a K step
    // meets the other's MFMAs), waves 8..11 are LOADERS (one per SIMD): they issue every LDS-DMA of the workgroup and nothing else.  A DMA instruction costs its issuing wave ~100-200 cycles of queue back-pressure (the CU takes in
    // ~30 B/clk when every CU streams); on a computing wave that time is lost to the matrix pipe (measured: all 8 waves loading + computing
    // 0.196 ms, a ping-pong schedule with the DMA inside the MFMA phases of half the waves 0.235 ms).  One workgroup barrier per K step:
    //     consumers:  barrier(kt) | fragments + 32 MFMAs of step kt
    //     loaders:    barrier(kt) | issue [patch part][weight tile kt + 2] | wait until tile kt + 1 (and everything older) has landed
    const bool loader = wave >= 8;
    const int lw = wave & 3;
    const int wm = (wave >> 1) & 3, wn = wave & 1;
    const __amdgpu_buffer_rsrc_t rin = buf_rsrc(a.in), rwf = buf_rsrc(a.wf);
    auto glds = [&](__amdgpu_buffer_rsrc_t rs, unsigned byte_off, char* dst) {
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (__attribute__((address_space(3))) void*)dst, 16, (int)byte_off, 0, 0, 0);
    };
    auto issue_b = [&](int kt) {                 // weight tile of K step kt -> ring slot kt % 3; four 1-KiB pieces (8 rows each) per loader wave
        const bool live = kt < (a.K >> 6);
        char* dst = bt + (kt % 3) * CP_BT_B;
#pragma unroll
        for (int p = 0; p < 4; ++p) {
            const int piece = lw * 4 + p, n = piece * 8 + (lane >> 3), pos = lane & 7, c = pos ^ (n & 7);
            glds(rwf, live ? ((unsigned)n * (unsigned)a.K + (unsigned)kt * 64u + (unsigned)c * 8u) * 2u : BUF_OOB, dst + piece * 1024);
        }
    };
    auto issue_patch = [&](int st, int part) {   // pieces (part * 4 + lw) * 5 .. + 4 of stage st's sub-lattice patch -> patch buffer st & 1
        const int cls = st >> 1, half = st & 1, py = cls >> 1, px = cls & 1;
        char* dst = patch + (st & 1) * CP_PATCH_B;
#pragma unroll
        for (int e = 0; e < CP_NPIECE_W; ++e) {
            const int piece = (part * 4 + lw) * CP_NPIECE_W + e;       // 0 .. 59, the patch has <= 49
            const int pp = piece * 8 + (lane >> 3), pos = lane & 7, c = pos ^ (pp & 7);
            const int pg = L0 + pp, E = pg / Ws, xx = pg - E * Ws;
            const int b = E / Hs, yy = E - b * Hs;
            const bool ok = st < 8 && pp < npx && b < a.B;
            const unsigned off = (((unsigned)(b * a.Hin + 2 * yy + py) * (unsigned)a.Hin + (unsigned)(2 * xx + px)) * CP_C + half * 64 + c * 8) * 2u;
            // surplus pieces are issued too (same instruction count in every wave and step), always out of range, into a dump KiB
            glds(rin, ok ? off : BUF_OOB, piece < CP_PPX / 8 ? dst + piece * 1024 : cp_sm + 3 * CP_BT_B + 2 * CP_PATCH_B);
        }
    };

    f32x4 acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

    if (loader) {
#pragma unroll
        for (int part = 0; part < 3; ++part) issue_patch(0, part);
        issue_b(0);
        issue_b(1);
        cp_wait<4>();                              // everything but tile 1
        for (int st = 0; st < 8; ++st) {
#pragma unroll
            for (int j4 = 0; j4 < 4; ++j4) {
                const int kt = st * 4 + j4;
                __builtin_amdgcn_s_barrier();          // barrier(kt): the consumers are done with step kt - 1 (slot (kt + 2) % 3, and at j4 = 0 the other patch buffer)
                asm volatile("" ::: "memory");
                if (j4 != 3) issue_patch(st + 1, j4);
                issue_b(kt + 2);
                // tile kt + 1 = the last 4 operations of the previous batch; younger: this batch
                if (j4 == 3) cp_wait<4>(); else cp_wait<4 + CP_NPIECE_W>();
            }
        }
        __builtin_amdgcn_s_barrier();              // barrier(32): pairs with the consumers' final one
    } else {
        // ---- per-lane geometry of the A fragments: patch pixel of (row tile i, row r16) at tap (0, 0)
        int pbase[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int m = min(m0 + wm * 64 + i * 16 + r16, mlast);
            const int g = m / Hout, x = m - g * Hout;
            pbase[i] = (g + g / Hout) * Ws + x - L0;
        }
        for (int st = 0; st < 8; ++st) {
            const char* pb = patch + (st & 1) * CP_PATCH_B;
#pragma unroll
            for (int j4 = 0; j4 < 4; ++j4) {
                const int kt = st * 4 + j4;
                __builtin_amdgcn_s_barrier();          // barrier(kt): tile kt (and at j4 = 0 this stage's patch) has landed for every loader
                asm volatile("" ::: "memory");
                const char* bs = bt + (kt % 3) * CP_BT_B;
                const int tap = (j4 >> 1) * Ws + (j4 & 1);
#pragma unroll
                for (int ks = 0; ks < 2; ++ks) {
                    bf16x8 af[4], bfr[4];
                    const int c = ks * 4 + q;
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        const int pp = pbase[i] + tap;
                        af[i] = *reinterpret_cast<const bf16x8*>(pb + pp * 128 + ((c ^ (pp & 7)) << 4));
                    }
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        const int n = wn * 64 + j * 16 + r16;
                        bfr[j] = *reinterpret_cast<const bf16x8*>(bs + n * 128 + ((c ^ (n & 7)) << 4));
                    }
#pragma unroll
                    for (int i = 0; i < 4; ++i)
#pragma unroll
                        for (int j = 0; j < 4; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[i], bfr[j], acc[i][j], 0, 0, 0);
                }
            }
        }
        __builtin_amdgcn_s_barrier();              // barrier(32)
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");         // the dummy tail DMAs target this workgroup's LDS
    __syncthreads();

    // ---- epilogue: bias + relu -> bf16 rows staged in LDS (over the patch buffers), then whole 256-byte output rows, 16 bytes per lane
    __bf16* cs = reinterpret_cast<__bf16*>(patch);
    if (!loader) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int n = wn * 64 + j * 16 + r16;
            const float bv = a.bias[n];
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int r = 0; r < 4; ++r) cs[(wm * 64 + i * 16 + q * 4 + r) * CP_LDC + n] = (__bf16)fmaxf(acc[i][j][r] + bv, 0.f);
        }
    }
    __syncthreads();
    const __amdgpu_buffer_rsrc_t rout = buf_rsrc(a.out);
    if (tid < 512) {
#pragma unroll
        for (int it = 0; it < CP_BM / 32; ++it) {
            const int row = it * 32 + (tid >> 4), ch = (tid & 15) * 8;
            const uint4 v = *reinterpret_cast<const uint4*>(cs + row * CP_LDC + ch);
            buf_store16(rout, (m0 + row) < mend ? ((unsigned)(m0 + row) * CP_C + ch) * 2u : BUF_OOB, v);
            if (a.mask) {        // (uniform) post-ReLU values: a channel is on iff its 16 bits are not zero
                auto nz2 = [](unsigned w) { return (unsigned)((int)(w << 16) > 0) | ((unsigned)((int)w >> 16 > 0) << 1); };      // as a signed 16-bit value > 0: the gate `activation > 0` exactly (-0.0 and NaN stay off)
                const unsigned byte = nz2(v.x) | (nz2(v.y) << 2) | (nz2(v.z) << 4) | (nz2(v.w) << 6);
                if ((m0 + row) < mend) a.mask[(size_t)(m0 + row) * 16 + (tid & 15)] = (unsigned char)byte;
            }
        }
    }
}

}  // namespace

// in: bf16 NHWC [B][Hin][Hin][128]; wf: bf16 [128][2048] in tap-parity K order (engine.hip fill_ktab / k_prep mode 2 with T, s set);
// out: bf16 [B*Hout*Hout][128] = relu(conv + bias).  SPAIR_ERR_UNSUPPORTED: the caller keeps the implicit-GEMM kernel.
// geometry check shared by the launcher and by conv_s2k4_patch_fwd16_fits(): SPAIR_OK with the tiling, or SPAIR_ERR_UNSUPPORTED
static int cp_plan(int B, int Hin, int Hout, int cin, int cout, int k, int s_, int& tiles_out, int& tpi_out) {
    if (cin != CP_C || cout != CP_C || k != 4 || s_ != 2 || Hin != 2 * Hout + 2 || B <= 0 || Hout <= 0) return SPAIR_ERR_UNSUPPORTED;
    const long long M = (long long)B * Hout * Hout;
    if (M * CP_C >= (1ll << 31) || (long long)B * Hin * Hin * CP_C >= (1ll << 31)) return SPAIR_ERR_UNSUPPORTED;
    // the largest patch window of any tile must fit the buffer: tiles of 256 consecutive rows of the whole batch if they do (the pattern repeats
    // with lcm(256, Hout * Hout)), else tiles that restart at every image (the last tile of an image is partial), else not this kernel
    auto window = [&](long long m0, long long ml) {
        const long long g0 = m0 / Hout, g1 = ml / Hout;
        return (int)((g1 + g1 / Hout + 1) * (Hout + 1) + (ml - g1 * Hout) + 1 - ((g0 + g0 / Hout) * (Hout + 1) + (m0 - g0 * Hout)) + 1);
    };
    int tiles = (int)((M + CP_BM - 1) / CP_BM), tpi = 0, worst = 0;
    // the tile pattern repeats every Hout^2 / gcd(CP_BM, Hout^2) tiles: every distinct tile is checked (a period too long to walk -- an odd
    // Hout: 16,129 tiles at 127 -- takes the per-image tiling, which is checked exactly below)
    const long long period = (long long)Hout * Hout / std::gcd((long long)CP_BM, (long long)Hout * Hout);
    const bool walk = std::min<long long>(tiles, period) <= 65536;
    for (int t = 0; walk && t < tiles && t < period; ++t)
        worst = std::max(worst, window((long long)t * CP_BM, std::min<long long>((long long)(t + 1) * CP_BM, M) - 1));
    if (walk && tiles > period) worst = std::max(worst, window((long long)(tiles - 1) * CP_BM, M - 1));       // the batch's partial last tile
    if (!walk || worst > CP_PPX) {
        const int HH = Hout * Hout;
        tpi = (HH + CP_BM - 1) / CP_BM;
        worst = 0;
        for (int t = 0; t < tpi; ++t) worst = std::max(worst, window((long long)t * CP_BM, std::min<long long>((long long)(t + 1) * CP_BM, HH) - 1));
        if (worst > CP_PPX) return SPAIR_ERR_UNSUPPORTED;
        tiles = B * tpi;
    }
    tiles_out = tiles; tpi_out = tpi;
    return SPAIR_OK;
}
// does the patch-resident kernel take this layer (then it also leaves the sign-bit mask the next layer's data gradient reads)?
bool conv_s2k4_patch_fwd16_fits(int B, int Hin, int Hout, int cin, int cout, int k, int s_) {
    int t, p;
    return cp_plan(B, Hin, Hout, cin, cout, k, s_, t, p) == SPAIR_OK;
}
int conv_s2k4_patch_fwd16(const void* in, const void* wf, const float* bias, void* out, int B, int Hin, int Hout, int cin, int cout, int k, int s_,
                          hipStream_t s, void* mask) {
    int tiles = 0, tpi = 0;
    { const int rc = cp_plan(B, Hin, Hout, cin, cout, k, s_, tiles, tpi); if (rc != SPAIR_OK) return rc; }
    const long long M = (long long)B * Hout * Hout;
    static std::atomic<unsigned long long> attr_done{0};
    if (spair_dyn_lds_once(reinterpret_cast<const void*>(&k_conv_s2k4_patch), CP_LDS, attr_done) != SPAIR_OK) return SPAIR_ERR_LAUNCH;
    ConvPatchArgs a;
    a.in = reinterpret_cast<const u16*>(in); a.wf = reinterpret_cast<const u16*>(wf); a.bias = bias; a.out = reinterpret_cast<u16*>(out);
    a.mask = reinterpret_cast<unsigned char*>(mask);
    a.B = B; a.Hin = Hin; a.Hout = Hout; a.M = (int)M; a.K = 16 * CP_C; a.tpi = tpi;
    hipLaunchKernelGGL(k_conv_s2k4_patch, dim3(tiles), dim3(768), CP_LDS, s, a);
    SPAIR_CHECK_LAUNCH();
    return SPAIR_OK;
}

// unit-level C ABI: in16 NHWC bf16 [B][Hin][Hin][128], wf16 bf16 [128][2048] with K = ((class * 2 + half) * 4 + tap) * 64 + c for
// (ky, kx, ci) = (py + 2 dy, px + 2 dx, half * 64 + c), class = py * 2 + px, tap = dy * 2 + dx; out16 bf16 [B*Hout*Hout][128]
extern "C" int spair_conv_s2k4_fwd16(const void* in16, const void* wf16, const float* bias, void* out16, int B, int Hin, int Hout, void* stream) {
    return conv_s2k4_patch_fwd16(in16, wf16, bias, out16, B, Hin, Hout, 128, 128, 4, 2, (hipStream_t)stream);
}
