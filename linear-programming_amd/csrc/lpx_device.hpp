// lpx_device.hpp -- shared device helpers for liblpx (gfx950 / CDNA4, wave64).
//
// Reductions: every selection rule of the reference is "extreme value, lowest index on ties"
// (np.argmin / np.argmax / np.argmax(mask)); they are implemented as 64-lane shuffle reductions on
// (value, index) pairs, then across the 4 waves of a 256-thread workgroup through LDS, then across
// workgroups through a per-workgroup partial array that the NEXT kernel in the stream combines (a kernel
// boundary is the cheapest grid-wide barrier on MI355X: ~1.5 us vs ~4-5 us for an in-kernel grid barrier).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "lpx.h"

#define LPX_BLOCK 256
#define LPX_WAVES (LPX_BLOCK / 64)
#define LPX_MAXWAVES 16 /* reductions work for any workgroup size up to 1024 */
#define LPX_MAXP 2048 /* capacity of one partial array */
#define LPX_ATOL 1e-8 /* numpy.isclose(x, 0) default atol, primal_solvers.py:142 */
#define LPX_RTOL 1e-5

typedef double d2_t __attribute__((ext_vector_type(2)));
typedef float f4_t __attribute__((ext_vector_type(4)));
typedef float f2_t __attribute__((ext_vector_type(2)));

#define LPX_INF (__builtin_huge_val())
#define LPX_NOIDX ((int64_t)0x4000000000000000LL)

// ---------------------------------------------------------------------------------------------------------
// error plumbing (host)
// ---------------------------------------------------------------------------------------------------------
void lpx_set_error(const char *fmt, ...);
#define LPX_HIP(call)                                                                          \
    do {                                                                                       \
        hipError_t _e = (call);                                                                \
        if (_e != hipSuccess) {                                                                \
            lpx_set_error("%s failed: %s (%s:%d)", #call, hipGetErrorString(_e), __FILE__, __LINE__); \
            return LPX_E_HIP;                                                                  \
        }                                                                                      \
    } while (0)
#define LPX_REQUIRE(cond, msg)                                          \
    do {                                                                \
        if (!(cond)) {                                                  \
            lpx_set_error("%s: %s", __func__, msg);                     \
            return LPX_E_ARG;                                           \
        }                                                               \
    } while (0)
#define LPX_LAUNCH_CHECK() LPX_HIP(hipGetLastError())

static inline bool lpx_aligned16(const void *p) { return (((uintptr_t)p) & 15u) == 0; }
static inline int64_t lpx_cdiv(int64_t a, int64_t b) { return (a + b - 1) / b; }
static inline int64_t lpx_roundup(int64_t a, int64_t b) { return lpx_cdiv(a, b) * b; }
__device__ __forceinline__ int64_t lpx_roundup_dev(int64_t a, int64_t b) { return (a + b - 1) / b * b; }

// ---------------------------------------------------------------------------------------------------------
// (value, index) reductions
// ---------------------------------------------------------------------------------------------------------
struct VI {
    double v;
    int64_t i;
};

// lexicographic "smaller value, then smaller index"
__device__ __forceinline__ bool vi_less(double v, int64_t i, double bv, int64_t bi) { return v < bv || (v == bv && i < bi); }

__device__ __forceinline__ VI wave_argmin(VI a)
{
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        const double ov = __shfl_xor(a.v, off);
        const int64_t oi = __shfl_xor(a.i, off);
        if (vi_less(ov, oi, a.v, a.i)) { a.v = ov; a.i = oi; }
    }
    return a;
}
__device__ __forceinline__ double wave_min(double v)
{
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v = fmin(v, __shfl_xor(v, off));
    return v;
}
__device__ __forceinline__ double wave_max(double v)
{
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v = fmax(v, __shfl_xor(v, off));
    return v;
}
__device__ __forceinline__ int64_t wave_imin(int64_t v)
{
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        const int64_t o = __shfl_xor(v, off);
        v = o < v ? o : v;
    }
    return v;
}
template <typename T>
__device__ __forceinline__ T wave_sum(T v)
{
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off);
    return v;
}

// LDS scratch for block-level reductions (one per kernel)
struct BlockRed {
    double v[LPX_MAXWAVES];
    int64_t i[LPX_MAXWAVES];
};

// all threads of the block receive the result; red must not be in use
__device__ __forceinline__ VI block_argmin(VI a, BlockRed &red)
{
    a = wave_argmin(a);
    const int w = threadIdx.x >> 6;
    if ((threadIdx.x & 63) == 0) { red.v[w] = a.v; red.i[w] = a.i; }
    __syncthreads();
    VI r{red.v[0], red.i[0]};
    const int nw = (blockDim.x + 63) >> 6;
    for (int k = 1; k < nw; ++k)
        if (vi_less(red.v[k], red.i[k], r.v, r.i)) { r.v = red.v[k]; r.i = red.i[k]; }
    __syncthreads();
    return r;
}
__device__ __forceinline__ double block_min(double v, BlockRed &red)
{
    v = wave_min(v);
    const int w = threadIdx.x >> 6;
    if ((threadIdx.x & 63) == 0) red.v[w] = v;
    __syncthreads();
    double r = red.v[0];
    const int nw = (blockDim.x + 63) >> 6;
    for (int k = 1; k < nw; ++k) r = fmin(r, red.v[k]);
    __syncthreads();
    return r;
}
__device__ __forceinline__ double block_max(double v, BlockRed &red)
{
    v = wave_max(v);
    const int w = threadIdx.x >> 6;
    if ((threadIdx.x & 63) == 0) red.v[w] = v;
    __syncthreads();
    double r = red.v[0];
    const int nw = (blockDim.x + 63) >> 6;
    for (int k = 1; k < nw; ++k) r = fmax(r, red.v[k]);
    __syncthreads();
    return r;
}
__device__ __forceinline__ int64_t block_imin(int64_t v, BlockRed &red)
{
    v = wave_imin(v);
    const int w = threadIdx.x >> 6;
    if ((threadIdx.x & 63) == 0) red.i[w] = v;
    __syncthreads();
    int64_t r = red.i[0];
    const int nw = (blockDim.x + 63) >> 6;
    for (int k = 1; k < nw; ++k) r = red.i[k] < r ? red.i[k] : r;
    __syncthreads();
    return r;
}

// One partial record per workgroup of a reduction stage (structure of arrays in the workspace).
struct Partials {
    double *v0;  // stage-dependent extreme value
    int64_t *i0; // its index
    double *v1;  // second value (min rc / max d / min row / second argmin value)
    int64_t *i1; // second index
};

// combine per-workgroup partials (executed redundantly by every workgroup of the consuming kernel)
__device__ __forceinline__ VI combine_argmin(const double *v, const int64_t *i, int64_t np, BlockRed &red)
{
    VI b{LPX_INF, LPX_NOIDX};
    for (int64_t k = threadIdx.x; k < np; k += blockDim.x)
        if (vi_less(v[k], i[k], b.v, b.i)) { b.v = v[k]; b.i = i[k]; }
    return block_argmin(b, red);
}
__device__ __forceinline__ double combine_min(const double *v, int64_t np, BlockRed &red)
{
    double b = LPX_INF;
    for (int64_t k = threadIdx.x; k < np; k += blockDim.x) b = fmin(b, v[k]);
    return block_min(b, red);
}
__device__ __forceinline__ double combine_max(const double *v, int64_t np, BlockRed &red)
{
    double b = -LPX_INF;
    for (int64_t k = threadIdx.x; k < np; k += blockDim.x) b = fmax(b, v[k]);
    return block_max(b, red);
}
__device__ __forceinline__ int64_t combine_imin(const int64_t *i, int64_t np, BlockRed &red)
{
    int64_t b = LPX_NOIDX;
    for (int64_t k = threadIdx.x; k < np; k += blockDim.x) b = i[k] < b ? i[k] : b;
    return block_imin(b, red);
}

// deterministic dot(c, x) by one workgroup of 1024 lanes: lane-strided fma partial sums, fixed reduction tree
// (primal_solvers.py:77-79, np.dot(c, x))
static __global__ __launch_bounds__(1024) void k_cost(const float *c, const double *x, int64_t n, double *cost)
{
    __shared__ double part[16];
    double acc = 0.0;
    for (int64_t j = threadIdx.x; j < n; j += 1024) acc = fma((double)c[j], x[j], acc);
    acc = wave_sum<double>(acc);
    if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) {
        double t = 0.0;
        for (int w = 0; w < 16; ++w) t += part[w];
        *cost = t;
    }
}

// ---------------------------------------------------------------------------------------------------------
// small math
// ---------------------------------------------------------------------------------------------------------
__device__ __forceinline__ bool lpx_is_optimal_min(double mn) { return (mn >= 0.0) || (fabs(mn) <= LPX_ATOL); }
__device__ __forceinline__ bool lpx_is_optimal_max(double mx) { return (mx <= 0.0) || (fabs(mx) <= LPX_ATOL); }

// utils.py:5-10 primal_simplex_div; f32: float32 quotient
__device__ __forceinline__ double lpx_psd(double n, double d, bool f32)
{
    if (!(d > 0.0)) return LPX_INF;
    return f32 ? (double)((float)n / (float)d) : n / d;
}
// utils.py:12-17 dual_simplex_div
__device__ __forceinline__ double lpx_dsd(double n, double d, bool f32)
{
    if (!(d < 0.0)) return LPX_INF;
    return f32 ? (double)((-(float)n) / (float)d) : (-n) / d;
}
