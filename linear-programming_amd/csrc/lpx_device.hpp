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

// Wave-level all-reduce on the VALU: DPP lane moves inside a row of 16 lanes (quad_perm xor 1, xor 2, row_ror 4,
// row_ror 8) and gfx950's v_permlane16_swap / v_permlane32_swap across rows, instead of six rounds of ds_bpermute
// shuffles through the LDS crossbar (~6x the latency; these reductions sit on the critical path of every small
// kernel of the loop).  Every lane ends with the result; each input is combined exactly once.
#define LPX_DPP_QUAD_XOR1 0xB1 /* quad_perm:[1,0,3,2] */
#define LPX_DPP_QUAD_XOR2 0x4E /* quad_perm:[2,3,0,1] */
#define LPX_DPP_ROW_ROR4 0x124
#define LPX_DPP_ROW_ROR8 0x128

template <typename T>
struct LpxWords {
    static_assert(sizeof(T) % 4 == 0, "whole dwords");
    int w[sizeof(T) / 4];
};
// (every destination lane of these controls has a source and the masks are full, so the move's "old" operand is dead:
// naming none saves the copy, and its wait states, that tying it to the source costs -- two VALU instructions a dword)
template <typename T, int CTRL>
__device__ __forceinline__ T lpx_dpp_move(T v)
{
    LpxWords<T> u;
    __builtin_memcpy(&u, &v, sizeof(T));
#pragma unroll
    for (unsigned k = 0; k < sizeof(T) / 4; ++k) u.w[k] = __builtin_amdgcn_mov_dpp(u.w[k], CTRL, 0xf, 0xf, false);
    T r;
    __builtin_memcpy(&r, &u, sizeof(T));
    return r;
}
// a = rows (v0, v0, v2, v2), b = rows (v1, v1, v3, v3) for SWAP32 == false (16-lane rows);
// a = (lo half, lo half), b = (hi half, hi half) for SWAP32 == true
template <typename T, bool SWAP32>
__device__ __forceinline__ void lpx_lane_swap(T v, T &a, T &b)
{
    LpxWords<T> u, ua, ub;
    __builtin_memcpy(&u, &v, sizeof(T));
#pragma unroll
    for (unsigned k = 0; k < sizeof(T) / 4; ++k) {
        if (SWAP32) {
            const auto r = __builtin_amdgcn_permlane32_swap((unsigned)u.w[k], (unsigned)u.w[k], false, false);
            ua.w[k] = (int)r[0];
            ub.w[k] = (int)r[1];
        } else {
            const auto r = __builtin_amdgcn_permlane16_swap((unsigned)u.w[k], (unsigned)u.w[k], false, false);
            ua.w[k] = (int)r[0];
            ub.w[k] = (int)r[1];
        }
    }
    __builtin_memcpy(&a, &ua, sizeof(T));
    __builtin_memcpy(&b, &ub, sizeof(T));
}
// every lane ends with the combination of its own row of 16 lanes
template <typename T, typename Op>
__device__ __forceinline__ T row16_allreduce(T v, Op op)
{
    v = op(v, lpx_dpp_move<T, LPX_DPP_QUAD_XOR1>(v));
    v = op(v, lpx_dpp_move<T, LPX_DPP_QUAD_XOR2>(v));
    v = op(v, lpx_dpp_move<T, LPX_DPP_ROW_ROR4>(v));
    return op(v, lpx_dpp_move<T, LPX_DPP_ROW_ROR8>(v));
}
template <typename T, typename Op>
__device__ __forceinline__ T wave_allreduce(T v, Op op)
{
    v = row16_allreduce(v, op);
    T a, b;
    lpx_lane_swap<T, false>(v, a, b);
    v = op(a, b);
    lpx_lane_swap<T, true>(v, a, b);
    return op(a, b);
}

// x += v[lane N of the lane's own row of 16 lanes] * p: the one DPP control the 64-bit VALU ops of gfx90a+ accept
// (row_newbcast).  One VALU instruction where a scalar broadcast costs two v_readlane_b32 and the fma; same issue rate
// as a plain v_fmac_f64 (tools/fma_rate.hip).  Every lane of the wave must be active, and `v` must have been written at
// least two wait states earlier (lpx_rows_bcast ends with them).
template <int N>
__device__ __forceinline__ void lpx_fmac_row_bcast(double &x, double v, double p)
{
    asm("v_fmac_f64_dpp %0, %1, %2 row_newbcast:%3 row_mask:0xf bank_mask:0xf" : "+v"(x) : "v"(v), "v"(p), "n"(N));
}
// the same for a lane index that is an unrolled loop's variable (a template argument must be a constant expression at
// parse time, an "n" operand only once the loop is unrolled)
#define LPX_FMAC_ROW_BCAST(x, v, p, N) \
    asm("v_fmac_f64_dpp %0, %1, %2 row_newbcast:%3 row_mask:0xf bank_mask:0xf" : "+v"(x) : "v"(v), "v"(p), "n"(N))
// E[g][lane] = v[16 g + lane % 16]: row g of the wave's 16-lane rows copied to all four rows -- three lane swaps per
// dword (v_permlane32_swap, then v_permlane16_swap on each half), the operand layout lpx_fmac_row_bcast broadcasts from
__device__ __forceinline__ void lpx_rows_bcast(double v, double (&E)[4])
{
    double lo, hi;
    lpx_lane_swap<double, true>(v, lo, hi);      // lo = rows (0, 1, 0, 1), hi = rows (2, 3, 2, 3)
    lpx_lane_swap<double, false>(lo, E[0], E[1]); // rows (0, 0, 0, 0), (1, 1, 1, 1)
    lpx_lane_swap<double, false>(hi, E[2], E[3]);
    asm volatile("s_nop 1" : "+v"(E[0]), "+v"(E[1]), "+v"(E[2]), "+v"(E[3])); // VALU write -> DPP read of the same VGPR
}

__device__ __forceinline__ VI wave_argmin(VI a)
{
    return wave_allreduce(a, [](VI x, VI y) { return vi_less(y.v, y.i, x.v, x.i) ? y : x; });
}
// min / max of doubles inside a reduction (numbers or infinities, +-inf the identity; a quiet NaN loses to a number as in
// fmin / fmax): the bare instruction, without the two quieting v_max_f64 that fmin() / fmax() prepend to operands whose
// origin the compiler does not know (lane moves)
__device__ __forceinline__ double lpx_min_f64(double x, double y)
{
    double r;
    asm("v_min_f64 %0, %1, %2" : "=v"(r) : "v"(x), "v"(y));
    return r;
}
__device__ __forceinline__ double lpx_max_f64(double x, double y)
{
    double r;
    asm("v_max_f64 %0, %1, %2" : "=v"(r) : "v"(x), "v"(y));
    return r;
}
__device__ __forceinline__ double wave_min(double v)
{
    return wave_allreduce(v, [](double x, double y) { return lpx_min_f64(x, y); });
}
__device__ __forceinline__ double wave_max(double v)
{
    return wave_allreduce(v, [](double x, double y) { return lpx_max_f64(x, y); });
}
__device__ __forceinline__ int wave_imin32(int v)
{
    return wave_allreduce(v, [](int x, int y) { return y < x ? y : x; });
}
__device__ __forceinline__ int64_t wave_imin(int64_t v)
{
    return wave_allreduce(v, [](int64_t x, int64_t y) { return y < x ? y : x; });
}
template <typename T>
__device__ __forceinline__ T wave_sum(T v)
{
    return wave_allreduce(v, [](T x, T y) { return x + y; });
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
