// Shared device/host helpers for the SPAIR gfx950 kernels.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

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

__device__ __forceinline__ float wave_reduce_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

// Sum over a 256-thread block (4 waves); result valid in every thread.
__device__ __forceinline__ float block_reduce_sum_256(float v, float* smem4) {
    v = wave_reduce_sum(v);
    const int w = threadIdx.x >> 6;
    __syncthreads();
    if ((threadIdx.x & 63) == 0) smem4[w] = v;
    __syncthreads();
    return smem4[0] + smem4[1] + smem4[2] + smem4[3];
}
