// Shared device/host helpers for the SPAIR gfx950 kernels.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <atomic>

#define SPAIR_OK 0
#define SPAIR_ERR_SHAPE (-1)
#define SPAIR_ERR_DTYPE (-2)
#define SPAIR_ERR_LAUNCH (-3)
#define SPAIR_ERR_UNSUPPORTED (-4)
#define SPAIR_ERR_ALIGN (-5)

#define SPAIR_F32 0
#define SPAIR_BF16 1

#define SPAIR_CHECK_LAUNCH()                                   \
    do {                                                       \
        hipError_t e__ = hipGetLastError();                    \
        if (e__ != hipSuccess) return SPAIR_ERR_LAUNCH;        \
    } while (0)


// Per-DEVICE launch state.  A kernel that needs more than 64 KB of dynamic LDS must be given the attribute on every device it is launched
// on, and a persistent grid is sized from the CURRENT device's CU count: both are keyed by hipGetDevice() here (one process may drive several
// devices from several host threads: include/spair_hip.h).  Two threads racing on the same device both set the attribute -- idempotent.
static inline int spair_dyn_lds_once(const void* fn, int bytes, std::atomic<unsigned long long>& done) {
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) return SPAIR_ERR_LAUNCH;
    const unsigned long long bit = 1ull << (dev & 63);
    if (!(done.load(std::memory_order_acquire) & bit)) {
        if (hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, bytes) != hipSuccess) return SPAIR_ERR_LAUNCH;
        done.fetch_or(bit, std::memory_order_release);
    }
    return SPAIR_OK;
}
static inline int spair_num_cus() {
    static std::atomic<int> cus[64];
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) return 256;
    int n = cus[dev & 63].load(std::memory_order_relaxed);
    if (n <= 0) {
        if (hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n <= 0) n = 256;
        cus[dev & 63].store(n, std::memory_order_relaxed);
    }
    return n;
}

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef unsigned short u16;

static inline int ceil_div(int a, int b) { return (a + b - 1) / b; }
static inline int round_up(int a, int b) { return ceil_div(a, b) * b; }

// Implicit-GEMM view of an NHWC tensor.  Row m = (b, y, x) over a logical
// Hout x Wout grid; column k = (ky*kw + kx)*Cin + ci.  Element (m,k) is
//   In[((b*Hin + y*sy + oy + ky*dky)*Win + x*sx + ox + kx*dkx)*Cin + ci]   (0 outside the tensor).
// Forward conv: sy=sx=stride, dky=dkx=+1, oy=ox=0 (the input is pre-padded).
// Data-gradient of a stride-2 conv, one output-parity class: sy=sx=1, dky=dkx=-1.
struct ConvDesc {
    int Hin, Win, Cin;
    int Hout, Wout;
    int kh, kw;
    int sy, sx;
    int dky, dkx;
    int oy, ox;
};

// Where row m=(b,y,x) of the GEMM result lands: C + (((b*Hc + y*osy+ooy)*Wc + x*osx+oox) * ldc).
struct RowMap {
    int Hout, Wout;
    int Hc, Wc;
    int osy, osx, ooy, oox;
};

__device__ __forceinline__ float sigmoidf_(float x) { return 1.0f / (1.0f + expf(-x)); }

template <int CTRL>
__device__ __forceinline__ float dpp_add_(float v) {
    return v + __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xf, 0xf, true));
}
// Sum over the 64 lanes, result in every lane.  Four DPP steps give each 16-lane row its sum (no LDS traffic, unlike
// __shfl_xor = ds_bpermute); the four row sums are then combined through scalar registers.
__device__ __forceinline__ float wave_reduce_sum(float v) {
    v = dpp_add_<0xB1>(v);      // quad_perm [1,0,3,2]
    v = dpp_add_<0x4E>(v);      // quad_perm [2,3,0,1]
    v = dpp_add_<0x141>(v);     // row_half_mirror
    v = dpp_add_<0x140>(v);     // row_mirror
    const int i = __builtin_bit_cast(int, v);
    return (__builtin_bit_cast(float, __builtin_amdgcn_readlane(i, 0)) + __builtin_bit_cast(float, __builtin_amdgcn_readlane(i, 16))) +
           (__builtin_bit_cast(float, __builtin_amdgcn_readlane(i, 32)) + __builtin_bit_cast(float, __builtin_amdgcn_readlane(i, 48)));
}

#if defined(__HIPCC__)
// Orders a wave's LDS writes before the LDS reads its OTHER lanes make afterwards, for data that never leaves the wave (wave-private
// tiles / tables): the hardware already executes one wave's DS instructions in order; this states the ordering for the compiler, which
// may otherwise move a load above a store it can prove distinct for the executing lane.  No instruction is emitted.
__device__ __forceinline__ void wave_lds_fence() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}
// 16-byte load through a buffer descriptor: an offset past num_records returns zeros, so a masked lane needs no select on the result
// and -- what matters -- no branch around the load (hipcc turns `ok ? *p : 0` into a conditional load, and a conditional VM op makes
// every later wait a vmcnt(0)).  The descriptor spans the whole 32-bit offset range; masked lanes pass BUF_OOB.
typedef unsigned int u32x4_t __attribute__((ext_vector_type(4)));
#define BUF_OOB 0xfffffff0u
__device__ __forceinline__ __amdgpu_buffer_rsrc_t buf_rsrc(const void* p) {
    const unsigned long long a = reinterpret_cast<unsigned long long>(p);
    const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)a), hi = __builtin_amdgcn_readfirstlane((unsigned)(a >> 32));
    void* q = reinterpret_cast<void*>(((unsigned long long)hi << 32) | lo);
    return __builtin_amdgcn_make_buffer_rsrc(q, (short)0, (int)0xffffff00u, 0x00020000);
}
__device__ __forceinline__ uint4 buf_load16(__amdgpu_buffer_rsrc_t r, unsigned byte_off) {
    const u32x4_t v = __builtin_amdgcn_raw_buffer_load_b128(r, (int)byte_off, 0, 0);
    return make_uint4(v.x, v.y, v.z, v.w);
}
// Stores through a descriptor: a masked lane passes BUF_OOB and the range check drops its write -- the store is unconditional for the
// compiler, so the pending-operation count stays known and a later wait for a LOAD is vmcnt(n), not vmcnt(0) (which on gfx9, one
// in-order counter for loads and stores, also waits for the acknowledgement of every store before it).
typedef unsigned int u32x2_t __attribute__((ext_vector_type(2)));
__device__ __forceinline__ void buf_store2(__amdgpu_buffer_rsrc_t r, unsigned byte_off, unsigned short bits) {
    __builtin_amdgcn_raw_buffer_store_b16((short)bits, r, (int)byte_off, 0, 0);
}
__device__ __forceinline__ void buf_store4(__amdgpu_buffer_rsrc_t r, unsigned byte_off, unsigned bits) {
    __builtin_amdgcn_raw_buffer_store_b32((int)bits, r, (int)byte_off, 0, 0);
}
__device__ __forceinline__ void buf_store8(__amdgpu_buffer_rsrc_t r, unsigned byte_off, unsigned lo, unsigned hi) {
    __builtin_amdgcn_raw_buffer_store_b64(u32x2_t{lo, hi}, r, (int)byte_off, 0, 0);
}
__device__ __forceinline__ void buf_store16(__amdgpu_buffer_rsrc_t r, unsigned byte_off, uint4 v) {
    __builtin_amdgcn_raw_buffer_store_b128(u32x4_t{v.x, v.y, v.z, v.w}, r, (int)byte_off, 0, 0);
}
#endif

// Sum over a 256-thread block (4 waves); result valid in every thread.
__device__ __forceinline__ float block_reduce_sum_256(float v, float* smem4) {
    v = wave_reduce_sum(v);
    const int w = threadIdx.x >> 6;
    __syncthreads();
    if ((threadIdx.x & 63) == 0) smem4[w] = v;
    __syncthreads();
    return smem4[0] + smem4[1] + smem4[2] + smem4[3];
}
