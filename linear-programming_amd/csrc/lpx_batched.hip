// lpx_batched.hip -- batches of independent small LPs: one workgroup per LP, whole solver state in LDS.
//
// Semantics: PrimalRevisedSimplexSolver(c, A, b, basis).solve(maxiters) per LP (primal_solvers.py:176-263),
// including the float32 first iteration.  HBM traffic is one load of (A, b, c, B^-1_0, basis) and one store of
// (x, basis, cost, status) per LP; every pivot runs out of the CU's 160 KB LDS.
//
// Arithmetic: every dot product is accumulated by ONE lane in index order with fma, exactly the oracle's
// (oracle/lpx_oracle.c) stated order, so this kernel is bit-identical to the CPU restatement -- not just
// sequence-identical.  Parallelism comes from the independent outputs (m or n of them per stage) and from
// the batch (two workgroups per CU, 512 LPs in flight on the chip).
//
// LDS image per LP:  B^-1 as m x (m+1) fp64 (row pad: column reads then hit distinct bank pairs),
// A as m x n fp32, and the small vectors.
#include "lpx_device.hpp"
#include <stdlib.h>
#include <string.h>

// Diagnostic build only (tools/batch_stamps.hip defines LPX_BATCH_STAMPS): per-phase cycle sums of lane 0, written to
// a buffer of their own; the product build compiles the hooks away.
#ifdef LPX_BATCH_STAMPS
__device__ unsigned long long g_batch_stamps[16];
#define LPX_STAMP_DECL unsigned long long st_last = clock64(), st_acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
#define LPX_STAMP(k)                                   \
    {                                                  \
        const unsigned long long now = clock64();      \
        st_acc[k] += now - st_last;                    \
        st_last = now;                                 \
    }
#define LPX_STAMP_FLUSH                                                                      \
    if (threadIdx.x == 0) {                                                                  \
        for (int q = 0; q < 8; ++q) atomicAdd(&g_batch_stamps[q], st_acc[q]);                \
        atomicAdd(&g_batch_stamps[8 + (__builtin_amdgcn_s_getreg(2308) & 3)], 1ULL);         \
        atomicAdd(&g_batch_stamps[12 + (__builtin_amdgcn_s_getreg(6148) & 3)], 1ULL);        \
    }
#else
#define LPX_STAMP_DECL
#define LPX_STAMP(k)
#define LPX_STAMP_FLUSH
#endif

struct BatchArgs {
    int64_t batch, m, n;
    const float *A, *b, *c, *Binv0;
    int64_t *basis;
    double *x, *cost;
    int32_t *status;
    int64_t *iters, *pivots;
    int64_t maxiters;
    int64_t *log;
    int64_t log_cap;
    int64_t kstruct; // 0: A is m x n.  > 0: inequality form G x <= h -- A holds only G (m x kstruct), c only its kstruct
                     // costs; columns kstruct..n-1 are the slack identity with zero cost and form the starting basis
};

struct BatchLds {
    double *Binv; // m x ldb
    float *A;     // m x n
    double *xB, *y, *rc, *d, *e, *rowr;
    float *c, *cb; // cb[i] = c[basis[i]]
    int *basis;
    int *vstate;
    int64_t ldb;
    int64_t bytes;
};
__host__ __device__ static inline BatchLds batch_lds(int64_t m, int64_t n, char *base)
{
    BatchLds L;
    char *p = base;
    auto take = [&](int64_t bytes) {
        char *q = p;
        p += (bytes + 15) / 16 * 16;
        return q;
    };
    L.ldb = m + 1;
    L.Binv = (double *)take(8 * m * L.ldb);
    L.xB = (double *)take(8 * m);
    L.y = (double *)take(8 * m);
    L.rc = (double *)take(8 * n);
    L.d = (double *)take(8 * m);
    L.e = (double *)take(8 * m);
    L.rowr = (double *)take(8 * m);
    L.A = (float *)take(4 * m * n);
    L.c = (float *)take(4 * n);
    L.cb = (float *)take(4 * m);
    L.basis = (int *)take(4 * m);
    L.vstate = (int *)take(4 * n);
    L.bytes = (int64_t)(p - base);
    return L;
}

// Workgroup reductions for the batched kernel: (value, 32-bit index) pairs, the independent shuffles of one round
// issued together, one barrier pair per reduction.  numpy semantics: first index of the extreme value.
struct BatchRed {
    double v[LPX_WAVES], w[LPX_WAVES];
    int i[LPX_WAVES];
};
struct BatchVI {
    double v;
    int i;
    int pad;
};
struct BatchVIW {
    double v, w;
    int i;
    int pad;
};
// min of v and (independently) min of idx over the workgroup
__device__ __forceinline__ void batch_min_imin(double &v, int &idx, BatchRed &red)
{
    const BatchVI r = wave_allreduce(BatchVI{v, idx, 0}, [](BatchVI x, BatchVI y) {
        return BatchVI{fmin(x.v, y.v), y.i < x.i ? y.i : x.i, 0};
    });
    v = r.v; idx = r.i;
    const int wv = threadIdx.x >> 6;
    if ((threadIdx.x & 63) == 0) { red.v[wv] = v; red.i[wv] = idx; }
    __syncthreads();
    v = red.v[0]; idx = red.i[0];
#pragma unroll
    for (int k = 1; k < LPX_WAVES; ++k) {
        v = fmin(v, red.v[k]);
        idx = red.i[k] < idx ? red.i[k] : idx;
    }
    __syncthreads();
}
// lexicographic argmin of (v, idx) and (independently) max of w over the workgroup
__device__ __forceinline__ void batch_argmin_max(double &v, int &idx, double &w, BatchRed &red)
{
    const BatchVIW r = wave_allreduce(BatchVIW{v, w, idx, 0}, [](BatchVIW x, BatchVIW y) {
        const bool take = y.v < x.v || (y.v == x.v && y.i < x.i);
        return BatchVIW{take ? y.v : x.v, fmax(x.w, y.w), take ? y.i : x.i, 0};
    });
    v = r.v; idx = r.i; w = r.w;
    const int wv = threadIdx.x >> 6;
    if ((threadIdx.x & 63) == 0) { red.v[wv] = v; red.i[wv] = idx; red.w[wv] = w; }
    __syncthreads();
    v = red.v[0]; idx = red.i[0]; w = red.w[0];
#pragma unroll
    for (int k = 1; k < LPX_WAVES; ++k) {
        if (red.v[k] < v || (red.v[k] == v && red.i[k] < idx)) { v = red.v[k]; idx = red.i[k]; }
        w = fmax(w, red.w[k]);
    }
    __syncthreads();
}

// The same for the primal ratio test, where lane t holds row t (idx == threadIdx.x on the lanes that have a row, rows
// ascend with the lane): only the value goes through DPP (2 dwords instead of 6), the first lane at the wave's minimum is
// the first argmin (np.argmin).  Lanes without a row pass v = +inf.
__device__ __forceinline__ void batch_row_argmin_max(double &v, int &idx, double &w, BatchRed &red)
{
    const double mn = wave_min(v);
    w = wave_max(w);
    const unsigned long long who = __ballot(v == mn);
    const int wv = threadIdx.x >> 6;
    const int first = who ? (wv << 6) + __ffsll((long long)who) - 1 : 0x7FFFFFFF; // empty only with NaNs
    if ((threadIdx.x & 63) == 0) { red.v[wv] = mn; red.i[wv] = first; red.w[wv] = w; }
    __syncthreads();
    v = red.v[0]; idx = red.i[0]; w = red.w[0];
#pragma unroll
    for (int k = 1; k < LPX_WAVES; ++k) {
        if (red.v[k] < v) { v = red.v[k]; idx = red.i[k]; } // strict: the earlier wave (lower rows) keeps ties
        w = fmax(w, red.w[k]);
    }
    __syncthreads();
}

// Sequential dot product acc = fma(a_k, b_k, acc), k ascending (the oracle's order), with the LDS loads of 8
// steps issued ahead of the 8 dependent fmas so that LDS latency overlaps the fma chain.
template <typename AccT, typename TA, typename TB>
__device__ __forceinline__ AccT batch_dot(const TA *pa, int sa, const TB *pb, int sb, int len)
{
    AccT acc = (AccT)0;
    int k = 0;
    for (; k + 8 <= len; k += 8) {
        AccT x[8], y[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            x[u] = (AccT)pa[(k + u) * sa];
            y[u] = (AccT)pb[(k + u) * sb];
        }
#pragma unroll
        for (int u = 0; u < 8; ++u) acc = fma(x[u], y[u], acc);
    }
    for (; k < len; ++k) acc = fma((AccT)pa[k * sa], (AccT)pb[k * sb], acc);
    return acc;
}

// The same sequential dot product with operand b in GLOBAL memory (stride sb): a lone workgroup has nothing to hide an L2
// round trip behind, so DEPTH loads are in flight before the first dependent fma (8 at a time costs one round trip per 8
// steps: 6 us per pass at m = 128).  Same order of operations, same bits.
template <typename AccT, int DEPTH, typename TA, typename TB>
__device__ __forceinline__ AccT block_dot_g(const TA *pa, int sa, const TB *pb, int64_t sb, int len)
{
    AccT acc = (AccT)0;
    int k = 0;
    for (; k + DEPTH <= len; k += DEPTH) {
        TB y[DEPTH];
#pragma unroll
        for (int u = 0; u < DEPTH; ++u) y[u] = pb[(int64_t)(k + u) * sb];
#pragma unroll
        for (int u = 0; u < DEPTH; ++u) acc = fma((AccT)pa[(k + u) * sa], (AccT)y[u], acc);
    }
    if (k < len) {
        TB y[DEPTH];
#pragma unroll
        for (int u = 0; u < DEPTH; ++u) y[u] = (k + u < len) ? pb[(int64_t)(k + u) * sb] : (TB)0;
#pragma unroll
        for (int u = 0; u < DEPTH; ++u)
            if (k + u < len) acc = fma((AccT)pa[(k + u) * sa], (AccT)y[u], acc);
    }
    return acc;
}

// The same sequential dot product when operand a is CONTIGUOUS and 16-byte aligned in LDS: a is read as 16-byte vectors
// (4 floats / 2 doubles per LDS instruction), 16 steps of loads ahead of their 16 dependent fmas.  The chains of the
// one-workgroup kernel are bound by LDS instruction issue (~13 cycles each), not by the fma chain.  Same order of
// operations, same bits.  (Issuing block b+1's loads before block b's fmas, as staged_blocks does, gained 5 % on the chain
// over B^-1 and lost 25 % on the one over A.)
template <typename TA> struct ChainVec;
template <> struct ChainVec<float> {
    typedef float vec __attribute__((ext_vector_type(4)));
    static constexpr int W = 4;
};
template <> struct ChainVec<double> {
    typedef d2_t vec;
    static constexpr int W = 2;
};
template <typename AccT, typename TA, typename TB>
__device__ __forceinline__ AccT chain_dot_va(const TA *pa, const TB *pb, int64_t sb, int len)
{
    typedef typename ChainVec<TA>::vec vec;
    constexpr int W = ChainVec<TA>::W, NV = 16 / W;
    AccT acc = (AccT)0;
    int k = 0;
    for (; k + 16 <= len; k += 16) {
        vec x[NV];
        TB y[16];
#pragma unroll
        for (int u = 0; u < NV; ++u) x[u] = *(const vec *)(pa + k + W * u);
#pragma unroll
        for (int u = 0; u < 16; ++u) y[u] = pb[(int64_t)(k + u) * sb];
#pragma unroll
        for (int u = 0; u < 16; ++u) acc = fma((AccT)x[u / W][u % W], (AccT)y[u], acc);
    }
    for (; k < len; ++k) acc = fma((AccT)pa[k], (AccT)pb[(int64_t)k * sb], acc);
    return acc;
}

// The same sequential dot product when BOTH operands are contiguous: 16-byte vector loads on both sides (2 doubles / 4
// floats per instruction) when both are 16-byte aligned -- a chain over a column of A kept TRANSPOSED (LDS copy or the
// state's At) against y or a row of B^-1 costs 48 load instructions per 64 steps instead of 96-128.  Same order, same bits.
// DEPTH steps of loads ahead of their fmas: 16 for operands in LDS, 32 when b comes from global memory (one round trip
// per block).
template <typename AccT, int DEPTH, typename TA, typename TB>
__device__ __forceinline__ AccT chain_dot_vv(const TA *pa, const TB *pb, int len)
{
    typedef typename ChainVec<TA>::vec veca;
    typedef typename ChainVec<TB>::vec vecb;
    constexpr int WA = ChainVec<TA>::W, WB = ChainVec<TB>::W;
    AccT acc = (AccT)0;
    int k = 0;
    if ((((uintptr_t)pa | (uintptr_t)pb) & 15) == 0) {
        for (; k + DEPTH <= len; k += DEPTH) {
            veca x[DEPTH / WA];
            vecb y[DEPTH / WB];
#pragma unroll
            for (int u = 0; u < DEPTH / WB; ++u) y[u] = *(const vecb *)(pb + k + WB * u);
#pragma unroll
            for (int u = 0; u < DEPTH / WA; ++u) x[u] = *(const veca *)(pa + k + WA * u);
#pragma unroll
            for (int u = 0; u < DEPTH; ++u) acc = fma((AccT)x[u / WA][u % WA], (AccT)y[u / WB][u % WB], acc);
        }
    }
    for (; k + 8 <= len; k += 8) {
        TA x[8];
        TB y[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            x[u] = pa[k + u];
            y[u] = pb[k + u];
        }
#pragma unroll
        for (int u = 0; u < 8; ++u) acc = fma((AccT)x[u], (AccT)y[u], acc);
    }
    for (; k < len; ++k) acc = fma((AccT)pa[k], (AccT)pb[k], acc);
    return acc;
}

// two sequential dot products against the same contiguous operand b, everything 16-byte aligned and contiguous (the bounded
// solver's sums along a row of the LDS copy of A): 2 + 2 + 1 load instructions per 4 steps instead of 12
template <typename AccT, typename TA, typename TB>
__device__ __forceinline__ void chain_dot2_vv(const TA *pa1, const TA *pa2, const TB *pb, int len, AccT &o1, AccT &o2)
{
    typedef typename ChainVec<TA>::vec veca;
    typedef typename ChainVec<TB>::vec vecb;
    constexpr int WA = ChainVec<TA>::W, WB = ChainVec<TB>::W;
    AccT a1 = (AccT)0, a2 = (AccT)0;
    int k = 0;
    if ((((uintptr_t)pa1 | (uintptr_t)pa2 | (uintptr_t)pb) & 15) == 0) {
        for (; k + 16 <= len; k += 16) {
            veca x1[16 / WA], x2[16 / WA];
            vecb y[16 / WB];
#pragma unroll
            for (int u = 0; u < 16 / WB; ++u) y[u] = *(const vecb *)(pb + k + WB * u);
#pragma unroll
            for (int u = 0; u < 16 / WA; ++u) {
                x1[u] = *(const veca *)(pa1 + k + WA * u);
                x2[u] = *(const veca *)(pa2 + k + WA * u);
            }
#pragma unroll
            for (int u = 0; u < 16; ++u) {
                a1 = fma((AccT)x1[u / WA][u % WA], (AccT)y[u / WB][u % WB], a1);
                a2 = fma((AccT)x2[u / WA][u % WA], (AccT)y[u / WB][u % WB], a2);
            }
        }
    }
    for (; k < len; ++k) {
        a1 = fma((AccT)pa1[k], (AccT)pb[k], a1);
        a2 = fma((AccT)pa2[k], (AccT)pb[k], a2);
    }
    o1 = a1;
    o2 = a2;
}

// two such dot products against the same global operand in one pass over its loads
template <typename AccT, int DEPTH, typename TA, typename TB>
__device__ __forceinline__ void block_dot2_g(const TA *pa1, const TA *pa2, const TB *pb, int64_t sb, int len, AccT &o1, AccT &o2)
{
    AccT a1 = (AccT)0, a2 = (AccT)0;
    for (int k = 0; k < len; k += DEPTH) {
        TB y[DEPTH];
#pragma unroll
        for (int u = 0; u < DEPTH; ++u) y[u] = (k + u < len) ? pb[(int64_t)(k + u) * sb] : (TB)0;
#pragma unroll
        for (int u = 0; u < DEPTH; ++u) {
            if (k + u < len) {
                a1 = fma((AccT)pa1[k + u], (AccT)y[u], a1);
                a2 = fma((AccT)pa2[k + u], (AccT)y[u], a2);
            }
        }
    }
    o1 = a1;
    o2 = a2;
}

__global__ __launch_bounds__(LPX_BLOCK) void k_batched_primal(BatchArgs a)
{
    extern __shared__ __attribute__((aligned(16))) char smem[];
    __shared__ BatchRed red;
    const int m = (int)a.m, n = (int)a.n, t = threadIdx.x;
    const int64_t lp = blockIdx.x;
    const BatchLds L = batch_lds(m, n, smem);
    const int ldb = (int)L.ldb;

    // ---- load the LP into LDS
    const int ks = (int)a.kstruct;
    if (ks == 0) {
        const float *gA = a.A + lp * (int64_t)m * n;
        for (int idx = t; idx < m * n; idx += LPX_BLOCK) L.A[idx] = gA[idx];
    } else { // [G | I] assembled here: the identity never exists in HBM
        const float *gG = a.A + lp * (int64_t)m * ks;
        for (int idx = t; idx < m * n; idx += LPX_BLOCK) {
            const int row = idx / n, col = idx - row * n;
            L.A[idx] = col < ks ? gG[row * ks + col] : (col - ks == row ? 1.f : 0.f);
        }
    }
    for (int idx = t; idx < n; idx += LPX_BLOCK) {
        L.c[idx] = ks == 0 ? a.c[lp * n + idx] : (idx < ks ? a.c[lp * ks + idx] : 0.f);
        L.vstate[idx] = LPX_VS_LOWER;
    }
    {
        int i = t / m, k = t - i * m; // incremental (row, col) walk: no division in the loop
        const int di = LPX_BLOCK / m, dk = LPX_BLOCK - di * m;
        for (int idx = t; idx < m * m; idx += LPX_BLOCK) {
            L.Binv[i * ldb + k] = a.Binv0 ? (double)a.Binv0[lp * (int64_t)m * m + idx] : (i == k ? 1.0 : 0.0);
            i += di; k += dk;
            if (k >= m) { k -= m; i += 1; }
        }
    }
    if (t < m) L.basis[t] = ks == 0 ? (int)a.basis[lp * m + t] : ks + t;
    __syncthreads();
    if (t < m) {
        L.vstate[L.basis[t]] = LPX_VS_BASIC;
        L.cb[t] = L.c[L.basis[t]];
        // bfs = inv_basis_matrix @ b in float32 (primal_solvers.py:38)
        float s0 = 0.f;
        for (int k = 0; k < m; ++k) s0 = fmaf((float)L.Binv[t * ldb + k], a.b[lp * m + k], s0);
        L.xB[t] = (double)s0;
    }
    __syncthreads();

    bool f32 = true;
    int status = LPX_RUNNING;
    int64_t counter = 0, pivots = 0;
    LPX_STAMP_DECL
    while (counter < a.maxiters) { // primal_solvers.py:191
        counter += 1;
        LPX_STAMP(7)
        // ---- y_t = (c_B @ B^-1)_t (:64, first product): one lane per column, i ascending
        if (t < m)
            L.y[t] = f32 ? (double)batch_dot<float>(L.cb, 1, L.Binv + t, ldb, m)
                         : (double)batch_dot<double>(L.cb, 1, L.Binv + t, ldb, m);
        LPX_STAMP(0)
        __syncthreads();
        LPX_STAMP(1)
        // ---- rc_t = c_t - (y @ A)_t ; rc[basis] = 0 (:64-65); min and first negative (:135-142)
        double mn = LPX_INF;
        int j = 0x7FFFFFFF;
        if (t < n) {
            double rc = f32 ? (double)(L.c[t] - batch_dot<float>(L.y, 1, L.A + t, n, m))
                            : (double)L.c[t] - batch_dot<double>(L.y, 1, L.A + t, n, m);
            if (L.vstate[t] == LPX_VS_BASIC) rc = 0.0;
            mn = rc;
            if (rc < 0.0) j = t;
        }
        LPX_STAMP(2)
        batch_min_imin(mn, j, red);
        LPX_STAMP(3)
        if (lpx_is_optimal_min(mn)) { // :140-142
            status = LPX_OPTIMAL;
            break;
        }
        // ---- d_t = (B^-1 @ A[:, j])_t (:129): one lane per row, k ascending; ratios (:165), max(d) (:146)
        double di = -LPX_INF, th = LPX_INF, dmax = -LPX_INF;
        int r = 0x7FFFFFFF;
        if (t < m) {
            di = f32 ? (double)batch_dot<float>(L.Binv + t * ldb, 1, L.A + j, n, m)
                     : (double)batch_dot<double>(L.Binv + t * ldb, 1, L.A + j, n, m);
            L.d[t] = di;
            th = lpx_psd(L.xB[t], di, f32);
            r = t;
            dmax = di;
        }
        LPX_STAMP(4)
        batch_argmin_max(th, r, dmax, red); // np.argmin: first minimum (:166)
        LPX_STAMP(5)
        if (dmax <= 0.0) {                  // :144-147
            status = LPX_PRIMAL_UNBOUNDED;
            break;
        }
        const double dr = L.d[r];
        const double xr = L.xB[r];
        // ---- eta column (:236-244) and copy of the old pivot row
        if (t < m) {
            L.e[t] = (t == r) ? 1.0 / dr : (-di) / dr;
            L.rowr[t] = L.Binv[r * ldb + t];
        }
        __syncthreads();
        LPX_STAMP(6)
        // ---- B^-1 <- E @ B^-1 ; bfs <- E @ bfs (:247-253) as a rank-1 update
        if (LPX_BLOCK % m == 0) {
            // lane owns column k and every (256/m)-th row: pivot-row entry in a register, 8 independent
            // read-modify-writes in flight
            const int rpp = LPX_BLOCK / m, k = t % m;
            const double pr = L.rowr[k];
#pragma unroll 8
            for (int i = t / m; i < m; i += rpp) {
                const double ei = L.e[i];
                double *p = &L.Binv[i * ldb + k];
                *p = (i == r) ? ei * pr : fma(ei, pr, *p);
            }
        } else {
            int i = t / m, k = t - i * m;
            const int dstep = LPX_BLOCK / m, kstep = LPX_BLOCK - dstep * m;
            for (int idx = t; idx < m * m; idx += LPX_BLOCK) {
                const double ei = L.e[i];
                double *p = &L.Binv[i * ldb + k];
                *p = (i == r) ? ei * L.rowr[k] : fma(ei, L.rowr[k], *p);
                i += dstep; k += kstep;
                if (k >= m) { k -= m; i += 1; }
            }
        }
        if (t < m) L.xB[t] = (t == r) ? L.e[t] * xr : fma(L.e[t], xr, L.xB[t]);
        if (t == 0) {
            const int leaving = L.basis[r];
            if (a.log && pivots < a.log_cap) {
                int64_t *lg = a.log + (lp * a.log_cap + pivots) * 3;
                lg[0] = r; lg[1] = j; lg[2] = leaving;
            }
            L.basis[r] = j; // :261
            L.cb[r] = L.c[j];
            L.vstate[leaving] = LPX_VS_LOWER;
            L.vstate[j] = LPX_VS_BASIC;
        }
        pivots += 1;
        f32 = false;
        __syncthreads();
    }

    LPX_STAMP_FLUSH
    // ---- result: x = scatter(x_B), cost = dot(c, x) in index order (primal_solvers.py:72-79)
    __syncthreads();
    for (int idx = t; idx < n; idx += LPX_BLOCK) L.rc[idx] = 0.0;
    __syncthreads();
    if (t < m) L.rc[L.basis[t]] = L.xB[t];
    __syncthreads();
    for (int idx = t; idx < n; idx += LPX_BLOCK) a.x[lp * n + idx] = L.rc[idx];
    if (t < m) a.basis[lp * m + t] = L.basis[t];
    if (t == 0) {
        double s1 = 0.0;
        for (int jx = 0; jx < n; ++jx) s1 = fma((double)L.c[jx], L.rc[jx], s1);
        a.cost[lp] = s1;
        a.status[lp] = status;
        a.iters[lp] = counter;
        a.pivots[lp] = pivots;
    }
}

// ---------------------------------------------------------------------------------------------------------
// One wave per LP: m <= 64 (a multiple of 8), n <= 256.
//
// Same arithmetic in the same order as k_batched_primal (bit-identical results).  What bounds a pivot at these
// sizes is not bandwidth but the latency of three chains of m dependent fmas (y, rc, d) and the instruction
// issue of a lone wave (tools/ubench.hip: ~6.5 cycles per fp64 fma, ~13.5 per LDS instruction of any width, 60
// cycles LDS latency).  With a 256-lane workgroup per LP three of four waves idle through every chain, so here an
// LP owns ONE wave and the four SIMDs of a CU run four LPs; nothing ever waits on another wave:
//  * lane t is row t of B^-1 and of x_B, and column t (t+64, ...) of A;
//  * A lives in registers for the whole solve (read from HBM once, coalesced): the rc chains read only y, as one
//    broadcast ds_read_b128 per two steps shared by all columns of the lane; the entering column reaches the d chain
//    through v_readlane, not through memory;
//  * B^-1 is the only matrix in LDS (rows of 66 doubles: 16-byte aligned rows for b128 row access, conflict-free
//    column access): 4 LPs per CU at m = 64, more for smaller m;
//  * min(rc) and max(d) are only ever compared with a threshold and the entering/leaving indices are "first lane
//    that ...": ballots, plus one 64-bit min on DPP for the ratio test;
//  * the pivot row of E @ B^-1 is e_r * row_r, not row_r + e_r * row_r: lane r takes its addend from a row of
//    -0.0 (fma(a, b, -0.0) == a * b bit for bit), so the update is one select-free fma stream over b128 accesses.
#define LPX_WAVE_LDB 66

struct WaveLds {
    double *Binv; // m x 66
    double *y, *cbd, *negz, *xs;
    float *cf;
    int *vst;
    int64_t bytes;
};
__host__ __device__ static inline WaveLds wave_lds(int64_t m, int64_t cpl, char *base)
{
    WaveLds L;
    char *p = base;
    auto take = [&](int64_t bytes) {
        char *q = p;
        p += (bytes + 15) / 16 * 16;
        return q;
    };
    L.Binv = (double *)take(8 * m * LPX_WAVE_LDB);
    L.y = (double *)take(8 * 64);
    L.cbd = (double *)take(8 * 64);
    L.negz = (double *)take(8 * 64);
    L.xs = (double *)take(8 * 64 * cpl);
    L.cf = (float *)take(4 * 64 * cpl);
    L.vst = (int *)take(4 * 64 * cpl);
    L.bytes = (int64_t)(p - base);
    return L;
}

// v of lane `lane` (wave-uniform) through the scalar unit
template <typename T>
__device__ __forceinline__ T lane_bcast(T v, int lane)
{
    LpxWords<T> u;
    __builtin_memcpy(&u, &v, sizeof(T));
#pragma unroll
    for (unsigned k = 0; k < sizeof(T) / 4; ++k) u.w[k] = __builtin_amdgcn_readlane(u.w[k], lane);
    T r;
    __builtin_memcpy(&r, &u, sizeof(T));
    return r;
}

// Chains and the update run over blocks of 8 steps.  Left alone the compiler issues each LDS load right before its
// use and a step waits out the full LDS latency (~100 cycles with four waves on the LDS); here block b+1's operands
// are loaded (ld) while block b is computed (cp), and fences keep the two stages apart in the schedule.
#define LPX_STAGE_FENCE            \
    asm volatile("" ::: "memory"); \
    __builtin_amdgcn_sched_barrier(0);
// ld / cp get (block, buffer, steps in the block): 8, or with PARTIAL the runtime count of the last block of an m
// that is not a multiple of 8 (uniform guards; the PARTIAL = false instantiation carries none).
template <typename Buf, bool PARTIAL, typename Load, typename Compute>
__device__ __forceinline__ void staged_blocks(int m, Load ld, Compute cp)
{
    Buf b0, b1;
    auto cnt = [&](int kb) { return PARTIAL ? (m - kb * 8 < 8 ? m - kb * 8 : 8) : 8; };
    ld(0, b0, cnt(0));
#pragma unroll
    for (int kb = 0; kb < 8; kb += 2) {
        if (kb * 8 < m) {
            if ((kb + 1) * 8 < m) ld(kb + 1, b1, cnt(kb + 1));
            LPX_STAGE_FENCE
            cp(kb, b0, cnt(kb));
            LPX_STAGE_FENCE
            if ((kb + 1) * 8 < m) {
                if (kb + 2 < 8 && (kb + 2) * 8 < m) ld(kb + 2, b0, cnt(kb + 2));
                LPX_STAGE_FENCE
                cp(kb + 1, b1, cnt(kb + 1));
                LPX_STAGE_FENCE
            }
        }
    }
}
struct WaveBuf8x2 {
    double a[8], b[8];
};
struct WaveBufRow {
    d2_t own[4], rr[4];
};

// CPL columns of A per lane, held as ColT (double: no conversion inside the rc chain; float when 4 columns per lane
// would not fit the register file as doubles)
// DUAL: DualRevisedSimplexSolver's loop (dual_solvers.py:82-120) on the same state -- leaving row = first x_B < 0, the dual
// row B^-1[r,:] A accumulated beside the reduced costs in the same pass over A's registers, ratio test over the columns.
template <int CPL, typename ColT, bool PARTIAL, bool DUAL>
__global__ __launch_bounds__(64) void k_batched_wave(BatchArgs a)
{
    constexpr int LDB = LPX_WAVE_LDB;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int m = (int)a.m, n = (int)a.n, t = threadIdx.x;
    const int64_t lp = blockIdx.x;
    const WaveLds S = wave_lds(m, CPL, smem);
    const bool has_row = t < m;
    const int trow = has_row ? t : 0; // idle lanes shadow row 0 so that every address stays in range

    // ---- columns t, t+64, ... of A and their costs into registers
    ColT Ac[CPL][64];
    float ct[CPL];
    {
        const int ks = (int)a.kstruct, na = ks == 0 ? n : ks; // stored columns of A (inequality form: G only)
        const float *gA = a.A + lp * (int64_t)m * na;
#pragma unroll
        for (int q = 0; q < CPL; ++q) {
            const int col = t + 64 * q, cc = col < na ? col : na - 1;
            const bool slack = ks != 0 && col >= ks; // unit column, never read from HBM
#pragma unroll
            for (int k = 0; k < 64; ++k) {
                const float v = gA[(int64_t)(k < m ? k : m - 1) * na + cc];
                Ac[q][k] = (ColT)(slack ? (k == col - ks ? 1.f : 0.f) : v);
            }
            ct[q] = slack ? 0.f : a.c[lp * na + cc];
            S.cf[col] = col < n ? ct[q] : 0.f;
            S.vst[col] = 0;
        }
    }
    // ---- B^-1, basis, c_B, x_B = B^-1 b in float32 (primal_solvers.py:38)
    for (int i = 0; i < m; ++i)
        if (has_row) S.Binv[i * LDB + t] = a.Binv0 ? (double)a.Binv0[lp * (int64_t)m * m + (int64_t)i * m + t] : (i == t ? 1.0 : 0.0);
    S.negz[t] = -0.0;
    int bas = 0;
    if (has_row) {
        bas = a.kstruct == 0 ? (int)a.basis[lp * m + t] : (int)a.kstruct + t;
        S.y[t] = (double)a.b[lp * m + t];
    }
    __syncthreads();
    double xB = 0.0;
    if (has_row) {
        S.vst[bas] = 1;
        S.cbd[t] = (double)S.cf[bas];
        float s0 = 0.f;
        for (int k = 0; k < m; ++k) s0 = fmaf((float)S.Binv[t * LDB + k], (float)S.y[k], s0);
        xB = (double)s0;
    }
    __syncthreads();
    bool basic[CPL];
#pragma unroll
    for (int q = 0; q < CPL; ++q) basic[q] = S.vst[t + 64 * q] != 0;

    int status = LPX_RUNNING;
    int64_t counter = 0, pivots = 0;
    LPX_STAMP_DECL

    // one iteration of the loop at primal_solvers.py:191; false when the loop ends
    auto iterate = [&](auto tag) -> bool {
        using AccT = decltype(tag);
        constexpr bool F32 = sizeof(AccT) == 4;
        if constexpr (sizeof(ColT) == 4) { // keep the float -> double conversions inside the chain (registers)
#pragma unroll
            for (int q = 0; q < CPL; ++q)
#pragma unroll
                for (int k = 0; k < 64; ++k) asm volatile("" : "+v"(Ac[q][k]));
        }
        LPX_STAMP(7)
        int r_dual = 0;
        if constexpr (DUAL) { // dual_solvers.py:47-54: optimal iff min(x_B) >= 0 or |min| <= atol; leaving = first x_B < 0
            if (__ballot(has_row && xB < -LPX_ATOL) == 0) {
                status = LPX_OPTIMAL;
                return false;
            }
            r_dual = __builtin_ctzll(__ballot(has_row && xB < 0.0));
        }
        // ---- y_t = (c_B @ B^-1)_t (:64, first product): column t, i ascending
        {
            AccT acc = (AccT)0;
            const double *col = S.Binv + trow;
            staged_blocks<WaveBuf8x2, PARTIAL>(
                m,
                [&](int kb, WaveBuf8x2 &B, int cnt) {
#pragma unroll
                    for (int u = 0; u < 8; ++u) {
                        B.a[u] = S.cbd[kb * 8 + u];
                        B.b[u] = col[(kb * 8 + (u < cnt ? u : 0)) * LDB]; // rows past m do not exist
                    }
                },
                [&](int, WaveBuf8x2 &B, int cnt) {
#pragma unroll
                    for (int u = 0; u < 8; ++u)
                        if (u < cnt) acc = fma((AccT)B.a[u], (AccT)B.b[u], acc);
                });
            if (has_row) S.y[t] = (double)acc;
        }
        LPX_STAMP(0)
        __syncthreads();
        LPX_STAMP(1)
        // ---- rc = c - y @ A ; rc[basis] = 0 (:64-65).  min(rc) is only used as "min >= 0 or |min| <= atol"
        //      (:140-142), i.e. no rc below -atol, and the entering index is the first rc < 0 (:135-139): ballots.
        int j = 0x7FFFFFFF;
        bool open = false;
        {
            AccT acc[CPL], acr[CPL]; // reduced-cost sums; dual row sums (DUAL)
#pragma unroll
            for (int q = 0; q < CPL; ++q) acc[q] = acr[q] = (AccT)0;
            const double *brow = S.Binv + r_dual * LDB; // row r of B^-1: broadcast operand of the dual row (:38)
            staged_blocks<WaveBuf8x2, PARTIAL>(
                m,
                [&](int kb, WaveBuf8x2 &B, int) {
#pragma unroll
                    for (int u = 0; u < 8; ++u) {
                        B.a[u] = S.y[kb * 8 + u];
                        if constexpr (DUAL) B.b[u] = brow[kb * 8 + u];
                    }
                },
                [&](int kb, WaveBuf8x2 &B, int cnt) {
#pragma unroll
                    for (int u = 0; u < 8; ++u) {
                        if (u < cnt) {
                            const AccT yk = (AccT)B.a[u];
#pragma unroll
                            for (int q = 0; q < CPL; ++q) acc[q] = fma(yk, (AccT)Ac[q][kb * 8 + u], acc[q]);
                            if constexpr (DUAL) {
                                const AccT bk = (AccT)B.b[u];
#pragma unroll
                                for (int q = 0; q < CPL; ++q) acr[q] = fma(bk, (AccT)Ac[q][kb * 8 + u], acr[q]);
                            }
                        }
                    }
                });
            LPX_STAMP(2)
            if constexpr (!DUAL) {
#pragma unroll
                for (int q = CPL - 1; q >= 0; --q) {
                    double rc;
                    if constexpr (F32) rc = (double)(ct[q] - acc[q]);
                    else rc = (double)ct[q] - acc[q];
                    if (basic[q] || t + 64 * q >= n) rc = 0.0;
                    const unsigned long long neg = __ballot(rc < 0.0);
                    open = open || __ballot(rc < -LPX_ATOL) != 0;
                    if (neg) j = 64 * q + __builtin_ctzll(neg);
                }
            } else {
                // dual ratio test (dual_solvers.py:61-80): theta_j = (-rc_j)/row_j over row_j < 0, first argmin;
                // min(row) >= 0, i.e. no candidate at all, is dual unboundedness (:41-44)
                double th[CPL], best = LPX_INF;
                bool any = false;
#pragma unroll
                for (int q = 0; q < CPL; ++q) {
                    double rc;
                    if constexpr (F32) rc = (double)(ct[q] - acc[q]);
                    else rc = (double)ct[q] - acc[q];
                    double row = (double)acr[q];
                    if (basic[q] || t + 64 * q >= n) { rc = 0.0; row = 0.0; }
                    th[q] = lpx_dsd(rc, row, F32);
                    best = fmin(best, th[q]);
                    any = any || __ballot(row < 0.0) != 0;
                }
                if (!any) {
                    status = LPX_DUAL_UNBOUNDED;
                    return false;
                }
                const double thmin = wave_min(best);
#pragma unroll
                for (int q = CPL - 1; q >= 0; --q) {
                    const unsigned long long at = __ballot(th[q] == thmin);
                    if (at) j = 64 * q + __builtin_ctzll(at);
                }
                open = true;
            }
        }
        LPX_STAMP(3)
        if (!open) { // :140-142
            status = LPX_OPTIMAL;
            return false;
        }
        j = j < n ? j : 0; // only reachable with NaNs in the data; keeps every index in range
        // ---- d_t = (B^-1 @ A[:, j])_t (:129): row t, k ascending; A[k][j] comes from its owner lane by v_readlane
        double di;
        float cj = 0.f;
        {
            const int lj = j & 63, qj = j >> 6;
            const double *row = S.Binv + trow * LDB;
            AccT acc = (AccT)0;
#pragma unroll
            for (int q = 0; q < CPL; ++q) {
                if (q == qj) {
                    staged_blocks<WaveBuf8x2, PARTIAL>(
                        m,
                        [&](int kb, WaveBuf8x2 &B, int) {
#pragma unroll
                            for (int u = 0; u < 8; ++u) B.a[u] = row[kb * 8 + u]; // rows are 66 doubles long
                        },
                        [&](int kb, WaveBuf8x2 &B, int cnt) {
#pragma unroll
                            for (int u = 0; u < 8; ++u)
                                if (u < cnt) acc = fma((AccT)B.a[u], (AccT)lane_bcast(Ac[q][kb * 8 + u], lj), acc);
                        });
                    cj = lane_bcast(ct[q], lj);
                }
            }
            di = has_row ? (double)acc : -LPX_INF;
        }
        // ---- ratios (:165), max(d) (:146), np.argmin (:166): the first lane holding the minimum ratio
        int r = r_dual;
        LPX_STAMP(4)
        if constexpr (!DUAL) {
            const double th = has_row ? lpx_psd(xB, di, F32) : LPX_INF;
            if (__ballot(di > 0.0) == 0) { // max(d) <= 0 (:144-147)
                status = LPX_PRIMAL_UNBOUNDED;
                return false;
            }
            const double thmin = wave_min(th);
            const unsigned long long at = __ballot(th == thmin);
            r = at ? __builtin_ctzll(at) : 0;
        }
        LPX_STAMP(5)
        // ---- eta column (:236-244), bfs <- E @ bfs, bookkeeping (:261)
        const double dr = lane_bcast(di, r), xr = lane_bcast(xB, r);
        const int leaving = lane_bcast(bas, r);
        const double e = ((t == r) ? 1.0 : -di) / dr;
        xB = (t == r) ? e * xr : fma(e, xr, xB);
        if (t == r) {
            if (a.log && pivots < a.log_cap) {
                int64_t *lg = a.log + (lp * a.log_cap + pivots) * 3;
                lg[0] = r; lg[1] = j; lg[2] = leaving;
            }
            bas = j;
            S.cbd[r] = (double)cj;
        }
#pragma unroll
        for (int q = 0; q < CPL; ++q) {
            if (t + 64 * q == leaving) basic[q] = false;
            if (t + 64 * q == j) basic[q] = true;
        }
        LPX_STAMP(6)
        // ---- B^-1 <- E @ B^-1 (:247-253): row t += e_t * row r; row r = e_r * row r (addend -0.0)
        if (has_row) {
            const d2_t *src = (const d2_t *)((t == r) ? S.negz : S.Binv + t * LDB);
            d2_t *dst = (d2_t *)(S.Binv + t * LDB);
            const d2_t *rowr = (const d2_t *)(S.Binv + r * LDB);
            staged_blocks<WaveBufRow, PARTIAL>( // whole blocks of 8 columns: the row pad absorbs the excess
                m,
                [&](int kb, WaveBufRow &B, int) {
#pragma unroll
                    for (int u = 0; u < 4; ++u) {
                        B.own[u] = src[kb * 4 + u];
                        B.rr[u] = rowr[kb * 4 + u];
                    }
                },
                [&](int kb, WaveBufRow &B, int) {
#pragma unroll
                    for (int u = 0; u < 4; ++u) {
                        B.own[u].x = fma(e, B.rr[u].x, B.own[u].x);
                        B.own[u].y = fma(e, B.rr[u].y, B.own[u].y);
                    }
#pragma unroll
                    for (int u = 0; u < 4; ++u) dst[kb * 4 + u] = B.own[u];
                });
        }
        pivots += 1;
        __syncthreads();
        return true;
    };

    if (counter < a.maxiters) { // float32 until the first eta update
        counter += 1;
        if (iterate(0.f))
            while (counter < a.maxiters) {
                counter += 1;
                if (!iterate(0.0)) break;
            }
    }

    LPX_STAMP_FLUSH
    // ---- result: x = scatter(x_B), cost = dot(c, x) in index order (primal_solvers.py:72-79)
    __syncthreads();
#pragma unroll
    for (int q = 0; q < CPL; ++q) S.xs[t + 64 * q] = 0.0;
    __syncthreads();
    if (has_row) S.xs[bas] = xB;
    __syncthreads();
#pragma unroll
    for (int q = 0; q < CPL; ++q)
        if (t + 64 * q < n) a.x[lp * n + t + 64 * q] = S.xs[t + 64 * q];
    if (has_row) a.basis[lp * m + t] = bas;
    if (t == 0) {
        double s1 = 0.0;
        for (int jx = 0; jx < n; ++jx) s1 = fma((double)S.cf[jx], S.xs[jx], s1);
        a.cost[lp] = s1;
        a.status[lp] = status;
        a.iters[lp] = counter;
        a.pivots[lp] = pivots;
    }
}

template <int CPL, typename ColT, bool PARTIAL, bool DUAL>
static int batched_wave_launch2(const BatchArgs &a, hipStream_t stream)
{
    const int64_t bytes = wave_lds(a.m, CPL, nullptr).bytes;
    // per launch, not cached: the attribute belongs to the current device and a process may drive several
    LPX_HIP(hipFuncSetAttribute((const void *)k_batched_wave<CPL, ColT, PARTIAL, DUAL>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                (int)bytes));
    hipLaunchKernelGGL((k_batched_wave<CPL, ColT, PARTIAL, DUAL>), dim3((unsigned)a.batch), dim3(64), (size_t)bytes, stream, a);
    LPX_LAUNCH_CHECK();
    return LPX_OK;
}
template <int CPL, typename ColT, bool DUAL = false>
static int batched_wave_launch(const BatchArgs &a, hipStream_t stream)
{
    return a.m % 8 ? batched_wave_launch2<CPL, ColT, true, DUAL>(a, stream) : batched_wave_launch2<CPL, ColT, false, DUAL>(a, stream);
}

static int batched_solve(int64_t batch, int64_t m, int64_t n, int64_t kstruct, const float *A, const float *b, const float *c,
                         const float *Binv0, int64_t *basis, double *x, double *cost, int32_t *status, int64_t *iters,
                         int64_t *pivots, int64_t maxiters, int64_t *log, int64_t log_cap, void *stream)
{
    LPX_REQUIRE(batch > 0 && m > 0 && n > 0 && maxiters >= 0, "bad sizes");
    LPX_REQUIRE(A && b && c && basis && x && cost && status && iters && pivots, "null pointer");
    LPX_REQUIRE(log == nullptr || log_cap > 0, "log without capacity");
    if (m > LPX_BLOCK || n > LPX_BLOCK) {
        lpx_set_error("lpx_batched_primal_solve: m and n must be <= %d (one lane per row/column)", LPX_BLOCK);
        return LPX_E_SIZE;
    }
    BatchArgs a{batch, m, n, A, b, c, Binv0, basis, x, cost, status, iters, pivots, maxiters, log, log_cap, kstruct};
    // LPX_BATCH_KERNEL = wave | generic pins one of the two kernels (tests, diagnostics); both produce the same bits.
    // Default: one wave per LP whenever the rows fit a wave.
    const char *pin = getenv("LPX_BATCH_KERNEL");
    const bool want_wave = pin && !strcmp(pin, "wave"), want_generic = pin && !strcmp(pin, "generic");
    if (pin && pin[0] && !want_wave && !want_generic) {
        lpx_set_error("lpx_batched_primal_solve: LPX_BATCH_KERNEL must be wave or generic, not '%s'", pin);
        return LPX_E_ARG;
    }
    if (want_wave && m > 64) {
        lpx_set_error("lpx_batched_primal_solve: the wave kernel takes m <= 64, not %lld", (long long)m);
        return LPX_E_SIZE;
    }
    if (m <= 64 && !want_generic) {
        if (n <= 64) return batched_wave_launch<1, double>(a, (hipStream_t)stream);
        if (n <= 128) return batched_wave_launch<2, double>(a, (hipStream_t)stream);
        return batched_wave_launch<4, float>(a, (hipStream_t)stream);
    }
    const int64_t bytes = batch_lds(m, n, nullptr).bytes;
    if (bytes > 158 * 1024) {
        lpx_set_error("lpx_batched_primal_solve: LP needs %lld B of LDS, more than one CU has", (long long)bytes);
        return LPX_E_SIZE;
    }
    LPX_HIP(hipFuncSetAttribute((const void *)k_batched_primal, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes));
    hipLaunchKernelGGL(k_batched_primal, dim3((unsigned)batch), dim3(LPX_BLOCK), (size_t)bytes, (hipStream_t)stream, a);
    LPX_LAUNCH_CHECK();
    return LPX_OK;
}

extern "C" int lpx_batched_primal_solve(int64_t batch, int64_t m, int64_t n, const float *A, const float *b, const float *c,
                                        const float *Binv0, int64_t *basis, double *x, double *cost, int32_t *status,
                                        int64_t *iters, int64_t *pivots, int64_t maxiters, int64_t *log, int64_t log_cap,
                                        void *stream)
{
    return batched_solve(batch, m, n, 0, A, b, c, Binv0, basis, x, cost, status, iters, pivots, maxiters, log, log_cap, stream);
}

extern "C" int lpx_batched_primal_solve_ineq(int64_t batch, int64_t m, int64_t k, const float *G, const float *h, const float *c,
                                             int64_t *basis, double *x, double *cost, int32_t *status, int64_t *iters,
                                             int64_t *pivots, int64_t maxiters, int64_t *log, int64_t log_cap, void *stream)
{
    LPX_REQUIRE(k > 0, "no structural columns");
    return batched_solve(batch, m, k + m, k, G, h, c, nullptr, basis, x, cost, status, iters, pivots, maxiters, log, log_cap, stream);
}

// DualRevisedSimplexSolver(c, A, b, basis).solve(maxiters) per LP (dual_solvers.py:82-120): same arguments as the primal
// entry point; the starting basis must be dual feasible (caller's check, dual_solvers.py:14-16).  One wave per LP only.
extern "C" int lpx_batched_dual_solve(int64_t batch, int64_t m, int64_t n, const float *A, const float *b, const float *c,
                                      const float *Binv0, int64_t *basis, double *x, double *cost, int32_t *status,
                                      int64_t *iters, int64_t *pivots, int64_t maxiters, int64_t *log, int64_t log_cap,
                                      void *stream)
{
    LPX_REQUIRE(batch > 0 && m > 0 && n > 0 && maxiters >= 0, "bad sizes");
    LPX_REQUIRE(A && b && c && basis && x && cost && status && iters && pivots, "null pointer");
    LPX_REQUIRE(log == nullptr || log_cap > 0, "log without capacity");
    if (m > 64 || n > 256) {
        lpx_set_error("lpx_batched_dual_solve: m <= 64 and n <= 256 (one wave per LP), not %lld x %lld", (long long)m, (long long)n);
        return LPX_E_SIZE;
    }
    BatchArgs a{batch, m, n, A, b, c, Binv0, basis, x, cost, status, iters, pivots, maxiters, log, log_cap, 0};
    if (n <= 64) return batched_wave_launch<1, double, true>(a, (hipStream_t)stream);
    if (n <= 128) return batched_wave_launch<2, double, true>(a, (hipStream_t)stream);
    return batched_wave_launch<4, float, true>(a, (hipStream_t)stream);
}

// ---------------------------------------------------------------------------------------------------------
// Whole-solve kernel for ONE small LP in revised form: a solve() call of PrimalRevisedSimplexSolver (primal_solvers.py:
// 176-263), DualRevisedSimplexSolver (dual_solvers.py:82-120) or BoundedVariablePrimalSimplexSolver (special_solvers.py:
// 221-343) as a single launch.  The multi-kernel chain of lpx_revised_iterate costs 5-9 launches (~40 us) per iteration
// whatever the size; a lone LP of a few hundred rows is latency-bound, so here one 256-lane workgroup loads the solver
// state (lpx_revised: B^-1, x_B, basis, var_state, control block) into LDS, runs the loop to the end and writes the
// state back.  B^-1 lives in LDS (m x (m+1) fp64: m <= 128 at n = 2m), A stays in global memory (read-only, cached).
// A batch of such states (an array of lpx_revised on the device) runs as one workgroup each.
// Arithmetic = the oracle's: every dot product accumulated by one lane in index order with fma, so pivot logs, B^-1, x_B
// and the per-variable states are bit-identical to lpo_primal_solve / lpo_dual_solve / lpo_bounded_solve.
// ---------------------------------------------------------------------------------------------------------
struct BlockArgs {
    lpx_revised single;          // used when states == nullptr
    const lpx_revised *states;   // device array, one per workgroup, all of the same m and n
    int64_t maxiters;
    double *x, *cost;            // optional per-LP results (batch-major: x is batch x n), or NULL
    // primal-dual algorithm (PD = true): the state is the restricted primal [A | I], n = n_orig + m
    double *pd_y;                // m, dual vector, in/out
    const double *pd_c64;        // n_orig float64 costs of the unrestricted problem
    int64_t pd_n_orig, pd_maxiters1;
    int a_in_lds;                // A (m x n float32) is copied into LDS behind the solver state (the launch made room)
};
struct BlockLds {
    double *Binv, *xB, *y, *d, *e, *rowr, *w, *rc, *row, *lb, *ub, *bu, *yd, *v;
    float *c, *cb;
    int *basis, *vstate;
    int64_t ldb, bytes;
};
#define LPX_KIND_PD 100 /* internal: LPX_KIND_PRIMAL inside the primal-dual outer loop */
__host__ __device__ static inline BlockLds block_lds(int64_t m, int64_t n, int kind, char *base)
{
    BlockLds L;
    char *p = base;
    auto take = [&](int64_t bytes) {
        char *q = p;
        p += (bytes + 15) / 16 * 16;
        return q;
    };
    // rows of B^-1 start 16-byte aligned (vector loads along a row) an odd number of 16-byte units apart (rows t, t + 1, ...
    // read by consecutive lanes fall into different banks); down a column consecutive lanes read consecutive words anyway
    const bool bnd = kind == LPX_KIND_BOUNDED || kind == LPX_KIND_DUAL_BOUNDED;
    L.ldb = m + 2 - (m & 1);
    if (((L.ldb / 2) & 1) == 0) L.ldb += 2;
    L.Binv = (double *)take(8 * m * L.ldb);
    L.xB = (double *)take(8 * m);
    L.y = (double *)take(8 * m);
    L.d = (double *)take(8 * m);
    L.e = (double *)take(8 * m);
    L.rowr = (double *)take(8 * m);
    L.w = (double *)take(bnd ? 8 * m : 0);
    L.rc = (double *)take(8 * n);
    L.row = (double *)take(kind == LPX_KIND_DUAL ? 8 * n : 0);
    L.lb = (double *)take(bnd ? 8 * n : 0);
    L.ub = (double *)take(bnd ? 8 * n : 0);
    L.bu = (double *)take(bnd ? 8 * n : 0);
    L.yd = (double *)take(kind == LPX_KIND_PD ? 8 * m : 0);
    L.v = (double *)take(kind == LPX_KIND_PD ? 8 * n : 0);
    L.c = (float *)take(4 * n);
    L.cb = (float *)take(4 * m);
    L.basis = (int *)take(4 * m);
    L.vstate = (int *)take(4 * n);
    L.bytes = (int64_t)(p - base);
    return L;
}
#define LPX_BLOCK_LDS_MAX (158 * 1024)
// A beside the state in LDS when both fit: a lone workgroup has nothing to hide an L2 round trip behind, and every chain
// over A (reduced costs, direction, the bounded solver's b - A_lb lb - A_ub ub) waits for one per 32 steps.  The copy is
// TRANSPOSED (column j contiguous, as the state's At): the chains walk DOWN a column, so a lane reads 4 floats per LDS
// instruction (chain_dot_vv).  Column stride = an odd number of 16-byte units: conflict-free for lane = column, and a
// row (the bounded solver's sums: lane = row, stride ldt between the steps) has consecutive lanes on consecutive words.
// The bounded loops keep the copy ROW-major (rows 16-byte aligned, an odd number of 16-byte units apart): their longest
// chain is the n-long sum along a ROW of A (x_B from scratch every iteration, special_solvers.py:169-175), which then reads 4
// floats per instruction (chain_dot2_vv); their column chains stay strided, with the other operand in 16-byte vectors.
__host__ __device__ static inline int64_t block_lds_A_ld(int64_t m)
{
    int64_t l = (m + 3) / 4 * 4;
    return ((l / 4) & 1) ? l : l + 4;
}
__host__ __device__ static inline bool block_kind_bounded(int kind) { return kind == LPX_KIND_BOUNDED || kind == LPX_KIND_DUAL_BOUNDED; }
__host__ __device__ static inline int64_t block_lds_A_bytes(int64_t m, int64_t n, int kind)
{
    return block_kind_bounded(kind) ? 4 * m * block_lds_A_ld(n) : 4 * n * block_lds_A_ld(m);
}
static inline bool block_A_fits(int64_t state_bytes, int64_t m, int64_t n, int kind)
{
    static const bool on = [] { const char *e = getenv("LPX_SMALL_A_LDS"); return !(e && e[0] == '0'); }();
    return on && state_bytes + block_lds_A_bytes(m, n, kind) <= LPX_BLOCK_LDS_MAX;
}

// PD: PrimalDualAlgorithm.solve (special_solvers.py:402-447) around the primal loop -- per outer iteration the admissible
// set J = isclose(y A, c) (:405-406), a FRESH restricted primal on [A_J | I] from the artificial basis (:408-425, the
// reference's cold restart, float32 first iteration included), then while its cost is positive the restricted dual
// y_rp = c_B B^-1, the unboundedness test (:435-436), theta (:438-443) and y += theta y_rp (:444).  One launch for the
// whole algorithm instead of ~20 launches and two host round trips per outer iteration.
// -DLPX_BLOCK_STAMPS: diagnostic build, thread 0 prints where the cycles of a solve went (never shipped)
#ifdef LPX_BLOCK_STAMPS
#define BS_DECL unsigned long long bs_last = clock64(), bs_acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
#define BS(k) { const unsigned long long bs_now = clock64(); bs_acc[k] += bs_now - bs_last; bs_last = bs_now; }
#define BS_PRINT if (t == 0 && blockIdx.x == 0) printf("block stamps: pivots %lld | y %llu rc %llu d %llu ratio %llu update %llu books %llu outer %llu (cycles per pivot)\n", (long long)pivots, bs_acc[0] / (pivots + 1), bs_acc[1] / (pivots + 1), bs_acc[2] / (pivots + 1), bs_acc[3] / (pivots + 1), bs_acc[4] / (pivots + 1), bs_acc[5] / (pivots + 1), bs_acc[6] / (pivots + 1));
#else
#define BS_DECL
#define BS(k)
#define BS_PRINT
#endif
template <int KIND, bool PD = false>
__global__ __launch_bounds__(LPX_BLOCK) void k_block_solve(BlockArgs a)
{
    extern __shared__ __attribute__((aligned(16))) char smem[];
    __shared__ BatchRed red;
    __shared__ double sh_cost;
    const lpx_revised s = a.states ? a.states[blockIdx.x] : a.single;
    lpx_ctrl *ctrl = s.ctrl;
    if (ctrl->status != LPX_RUNNING) return;
    const int m = (int)s.m, n = (int)s.n, t = threadIdx.x;
    const BlockLds L = block_lds(m, n, PD ? LPX_KIND_PD : KIND, smem);
    const int ldb = (int)L.ldb;
    // both bounded-variable loops: the primal one (special_solvers.py:94-343) and the opt-in dual one (lpx.h, LPX_KIND_DUAL_BOUNDED)
    constexpr bool BND = KIND == LPX_KIND_BOUNDED || KIND == LPX_KIND_DUAL_BOUNDED;
    // BND: A row-major (entry (i, col) at A[i * ldA + col]; At: the state's transposed copy while A is in global memory);
    // else Ac + col * ldc: column col of A, contiguous (the LDS copy, or the state's transposed copy)
    const float *A = s.A, *At = s.At, *Ac = s.At;
    int64_t ldA = s.ldA, ldc = s.ldAt;
    if (a.a_in_lds) {
        float *Al = (float *)(smem + L.bytes);
        if (BND) {
            const int ldl = (int)block_lds_A_ld(n);
            for (int idx = t; idx < m * n; idx += LPX_BLOCK) {
                const int i = idx / n, k = idx - i * n;
                Al[i * ldl + k] = s.A[(int64_t)i * s.ldA + k];
            }
            A = Al;
            ldA = ldl;
            At = nullptr; // a row of the LDS copy is read as it lies
        } else {
            const int ldl = (int)block_lds_A_ld(m);
            for (int idx = t; idx < m * n; idx += LPX_BLOCK) {
                const int k = idx / m, i = idx - k * m;
                Al[k * ldl + i] = s.At[(int64_t)k * s.ldAt + i];
            }
            Ac = Al;
            ldc = ldl;
        }
    }
    const bool lds_a = a.a_in_lds != 0;
// chain over column col of A against the contiguous vector pa (y, a row of B^-1): k ascending, one fma per step.
// COLDOT: lane = column.  From global memory the row-major matrix is the coalesced one (64 lanes read 64 consecutive
// floats per step; the transposed copy would touch 64 cache lines per instruction).  COLDOT_U: the same column in every
// lane (the direction B^-1 A_j): contiguous from the transposed copy wherever it lies.
#define COLDOT(AccT, pa, col)                                                                                           \
    (!lds_a ? block_dot_g<AccT, 32>((pa), 1, A + (col), ldA, m)                                                          \
            : (BND ? chain_dot_va<AccT>((pa), A + (col), ldA, m) : chain_dot_vv<AccT, 16>((pa), Ac + (int64_t)(col) * ldc, m)))
#define COLDOT_U(AccT, pa, col)                                                                                         \
    (BND ? (lds_a ? chain_dot_va<AccT>((pa), A + (col), ldA, m) : block_dot_g<AccT, 32>((pa), 1, A + (col), ldA, m))      \
         : (lds_a ? chain_dot_vv<AccT, 16>((pa), Ac + (int64_t)(col) * ldc, m) : chain_dot_vv<AccT, 32>((pa), Ac + (int64_t)(col) * ldc, m)))
    const int n_orig = PD ? (int)a.pd_n_orig : 0;
    int64_t outer = 0;
    int pd_status = LPX_RUNNING;

    // ---- solver state into LDS
    for (int idx = t; idx < m * m; idx += LPX_BLOCK) {
        const int i = idx / m, k = idx - i * m;
        L.Binv[i * ldb + k] = s.Binv[(int64_t)i * s.ldB + k];
    }
    for (int j = t; j < n; j += LPX_BLOCK) {
        L.c[j] = s.c[j];
        L.vstate[j] = s.var_state[j];
        if (BND) {
            L.lb[j] = s.lb[j];
            L.ub[j] = s.ub[j];
        }
    }
    if (t < m) {
        L.basis[t] = (int)s.basis[t];
        L.xB[t] = s.xB[t];
    }
    __syncthreads();
    if (t < m) L.cb[t] = L.c[L.basis[t]];
    if (PD && t < m) L.yd[t] = a.pd_y[t];
    bool f32 = ctrl->f32_phase != 0 && KIND != LPX_KIND_DUAL_BOUNDED; // the opt-in bounded dual loop is float64 throughout
    int status = LPX_RUNNING;
    int64_t counter = ctrl->counter, pivots = ctrl->pivots;
    int last_j = -1, last_r = -1, last_leaving = -1, last_flip = 0, last_lstate = 0;
    double last_price = 0.0;
    __syncthreads();

    // x_B = B^-1 (b - A_lb lb - A_ub ub) from scratch, special_solvers.py:169-175 (bounded solver: after every iteration, and
    // once at the start -- a function of the state alone, so recomputing it here equals the value the constructor or the
    // previous call left, in this kernel's own summation order)
    auto bounded_update_bfs = [&]() {
        // The two sums run over ALL columns with the bound of a column that is not at that bound replaced by 0:
        // fma(a, 0, acc) == acc bit for bit (acc is never -0.0: it starts at +0.0 and an exact cancellation rounds
        // to +0.0), so the selection costs no branch inside the chain.  Row t of A is read as column t of the
        // transposed copy (coalesced across lanes) when the state has one.
        for (int col = t; col < n; col += LPX_BLOCK) {
            const int vs = L.vstate[col];
            L.rc[col] = (vs == LPX_VS_LOWER) ? L.lb[col] : 0.0;
            L.bu[col] = (vs == LPX_VS_UPPER) ? L.ub[col] : 0.0;
        }
        __syncthreads();
        if (t < m) {
            double accl, accu;
            if (At != nullptr) block_dot2_g<double, 32>(L.rc, L.bu, At + t, s.ldAt, n, accl, accu);
            else chain_dot2_vv<double>(L.rc, L.bu, A + (int64_t)t * ldA, n, accl, accu);
            L.w[t] = ((double)s.b[t] - accl) - accu;
        }
        __syncthreads();
        if (t < m) L.xB[t] = chain_dot_vv<double, 16>(L.Binv + t * ldb, L.w, m);
        __syncthreads();
    };
    if (BND) bounded_update_bfs();

    BS_DECL
    do { // one pass, or the outer loop of the primal-dual algorithm
    BS(6)
    if (PD) {
        outer += 1;
        for (int col = t; col < n_orig; col += LPX_BLOCK) { // :405-406
            const double v = COLDOT(double, L.yd, col);
            const double cj = a.pd_c64[col];
            L.v[col] = v;
            L.vstate[col] = (fabs(v - cj) <= LPX_ATOL + LPX_RTOL * fabs(cj)) ? LPX_VS_LOWER : LPX_VS_EXCLUDED; // np.isclose
        }
        for (int idx = t; idx < m * m; idx += LPX_BLOCK) { // :408-424: artificial basis, B^-1 = I, bfs = b
            const int i = idx / m, k = idx - i * m;
            L.Binv[i * ldb + k] = (i == k) ? 1.0 : 0.0;
        }
        if (t < m) {
            L.basis[t] = n_orig + t;
            L.xB[t] = (double)s.b[t];
            L.cb[t] = L.c[n_orig + t];
            L.vstate[n_orig + t] = LPX_VS_BASIC;
        }
        f32 = true;
        status = LPX_RUNNING;
        counter = 0;
        __syncthreads();
    }
    BS(6)
    for (int64_t it = 0; it < a.maxiters; ++it) {
        counter += 1;
        int r = 0, j = 0;
        BS(5)
        if (KIND == LPX_KIND_DUAL) { // dual_solvers.py:47-54: min(x_B), first row with x_B < 0
            double mn = (t < m) ? L.xB[t] : LPX_INF;
            int first = (t < m && mn < 0.0) ? t : 0x7FFFFFFF;
            batch_min_imin(mn, first, red);
            last_price = mn;
            if (lpx_is_optimal_min(mn)) {
                status = LPX_OPTIMAL;
                break;
            }
            r = first;
        }
        bool below = false;
        if (KIND == LPX_KIND_DUAL_BOUNDED) {
            // leaving row: the FIRST basic variable outside its bounds; optimal when the worst violation is <= 1e-8
            // (dual_solvers.py:47-54 with "x_i < 0" read as "outside [lb, ub]"; oracle: lpo_dual_bounded_solve)
            double neg = LPX_INF; // -(violation): min over the rows = -(worst violation)
            int first = 0x7FFFFFFF;
            if (t < m) {
                const int bv = L.basis[t];
                const double x = L.xB[t], lo = L.lb[bv] - x, hi = x - L.ub[bv];
                const double inf = lo > hi ? lo : hi;
                neg = -inf;
                if (inf > 0.0) first = t;
            }
            batch_min_imin(neg, first, red);
            last_price = neg;
            if (lpx_is_optimal_min(neg)) {
                status = LPX_OPTIMAL;
                break;
            }
            r = first;
            below = L.xB[r] < L.lb[L.basis[r]];
        }
        // ---- y = c_B @ B^-1 (primal_solvers.py:64, first product): one lane per column, i ascending
        if (t < m)
            L.y[t] = f32 ? (double)chain_dot_va<float>(L.cb, L.Binv + t, ldb, m) : (double)chain_dot_va<double>(L.cb, L.Binv + t, ldb, m);
        __syncthreads();
        BS(0)
        // ---- reduced costs, one lane per column (columns t, t + 256, ...), and the rule's reduction
        if (KIND == LPX_KIND_PRIMAL) { // :64-65, :135-142
            double mn = LPX_INF;
            int jn = 0x7FFFFFFF;
            for (int col = t; col < n; col += LPX_BLOCK) {
                const int vs = L.vstate[col];
                if (vs == LPX_VS_EXCLUDED) continue; // restricted primal of the primal-dual algorithm
                double rc = f32 ? (double)(L.c[col] - COLDOT(float, L.y, col))
                                : (double)L.c[col] - COLDOT(double, L.y, col);
                if (vs == LPX_VS_BASIC) rc = 0.0;
                mn = fmin(mn, rc);
                if (rc < 0.0 && col < jn) jn = col;
            }
            batch_min_imin(mn, jn, red);
            last_price = mn;
            if (lpx_is_optimal_min(mn)) { // :140-142
                status = LPX_OPTIMAL;
                break;
            }
            j = jn;
        } else if (KIND == LPX_KIND_DUAL) { // dual_solvers.py:37-40, :61-80
            double th = LPX_INF, rowmin = LPX_INF;
            int jn = 0x7FFFFFFF;
            for (int col = t; col < n; col += LPX_BLOCK) {
                const int vs = L.vstate[col];
                if (vs == LPX_VS_EXCLUDED) continue;
                double rc, row;
                if (lds_a) { // both chains in one pass over the column's 16-byte loads, their fmas interleaved (same order each)
                    if (f32) {
                        float z, rw;
                        chain_dot2_vv<float>(L.y, L.Binv + r * ldb, Ac + (int64_t)col * ldc, m, z, rw);
                        rc = (double)(L.c[col] - z);
                        row = (double)rw;
                    } else {
                        double z;
                        chain_dot2_vv<double>(L.y, L.Binv + r * ldb, Ac + (int64_t)col * ldc, m, z, row);
                        rc = (double)L.c[col] - z;
                    }
                } else if (f32) { // from global memory two passes: sharing the loads there measured slower (16 vs 10 us)
                    rc = (double)(L.c[col] - COLDOT(float, L.y, col));
                    row = (double)COLDOT(float, L.Binv + r * ldb, col);
                } else {
                    rc = (double)L.c[col] - COLDOT(double, L.y, col);
                    row = COLDOT(double, L.Binv + r * ldb, col);
                }
                if (vs == LPX_VS_BASIC) rc = row = 0.0;
                const double tj = lpx_dsd(rc, row, f32);
                if (tj < th || (tj == th && col < jn)) { th = tj; jn = col; }
                rowmin = fmin(rowmin, row);
            }
            double negmin = -rowmin;
            batch_argmin_max(th, jn, negmin, red); // first argmin of theta; max(-row) = -min(row)
            if (-negmin >= 0.0) { // dual_solvers.py:41-44
                status = LPX_DUAL_UNBOUNDED;
                break;
            }
            j = jn;
        } else if (KIND == LPX_KIND_DUAL_BOUNDED) {
            // ratio test over the nonbasic columns: alpha = row when x_r sits below its bound, -row above; candidates are
            // "at lower with alpha < 0" and "at upper with alpha > 0"; theta = (-rc) / alpha, first argmin (utils.py:12-17)
            double th = LPX_INF, dummy = 0.0;
            int jn = 0x7FFFFFFF;
            for (int col = t; col < n; col += LPX_BLOCK) {
                const int vs = L.vstate[col];
                if (vs != LPX_VS_LOWER && vs != LPX_VS_UPPER) continue;
                const double rc = (double)L.c[col] - COLDOT(double, L.y, col);
                const double row = COLDOT(double, L.Binv + r * ldb, col);
                const double alpha = below ? row : -row;
                if ((vs == LPX_VS_LOWER && alpha < 0.0) || (vs == LPX_VS_UPPER && alpha > 0.0)) {
                    const double tj = (-rc) / alpha;
                    if (jn == 0x7FFFFFFF || tj < th) { th = tj; jn = col; } // columns ascend per lane: first minimum
                }
            }
            if (jn == 0x7FFFFFFF) th = LPX_INF;
            batch_argmin_max(th, jn, dummy, red);
            if (jn == 0x7FFFFFFF) { // no column can repair row r: the primal is infeasible (dual_solvers.py:41-44)
                status = LPX_DUAL_UNBOUNDED;
                break;
            }
            j = jn;
        } else { // special_solvers.py:177-214: float32 reduced-cost array, first argmax
            double best = LPX_INF; // argmin of (-rc, col) == first argmax of rc
            int jn = 0x7FFFFFFF;
            for (int col = t; col < n; col += LPX_BLOCK) {
                const int vs = L.vstate[col];
                double rc = 0.0;
                if (vs != LPX_VS_BASIC) {
                    float tt;
                    if (f32) tt = (-L.c[col]) + COLDOT(float, L.y, col);
                    else tt = (float)((double)(-L.c[col]) + COLDOT(double, L.y, col));
                    rc = (vs == LPX_VS_UPPER) ? (double)(-tt) : (double)tt;
                }
                if (-rc < best || (-rc == best && col < jn)) { best = -rc; jn = col; }
            }
            double dummy = 0.0;
            batch_argmin_max(best, jn, dummy, red);
            const double mx = -best;
            last_price = mx;
            if (lpx_is_optimal_max(mx)) { // :213-214
                status = LPX_OPTIMAL;
                break;
            }
            j = jn;
        }
        BS(1)
        double sign = 1.0;
        if (KIND == LPX_KIND_BOUNDED) {
            const int vj = L.vstate[j];
            if (vj == LPX_VS_UPPER) sign = -1.0;
            else if (vj != LPX_VS_LOWER) { // special_solvers.py:331
                status = LPX_NOT_NONBASIC;
                last_j = j;
                break;
            }
        }
        // ---- d = B^-1 @ A[:, j] (primal_solvers.py:129): one lane per row, k ascending
        double di = 0.0;
        if (t < m) {
            di = f32 ? (double)COLDOT_U(float, L.Binv + t * ldb, j)
                     : COLDOT_U(double, L.Binv + t * ldb, j);
            L.d[t] = di;
        }
        BS(2)
        bool flip = false;
        int lstate = LPX_VS_LOWER;
        if (KIND == LPX_KIND_PRIMAL) { // ratio test :149-167, unboundedness :144-147
            double th = (t < m) ? lpx_psd(L.xB[t], di, f32) : LPX_INF, dmax = (t < m) ? di : -LPX_INF;
            int rr = 0;
            batch_row_argmin_max(th, rr, dmax, red);
            if (dmax <= 0.0) {
                status = LPX_PRIMAL_UNBOUNDED;
                last_j = j;
                break;
            }
            r = rr;
        } else if (KIND == LPX_KIND_BOUNDED) { // three-way ratio test, special_solvers.py:239-329
            double g1 = LPX_INF, g2 = LPX_INF, dummy = 0.0;
            int r1 = (t < m) ? t : 0x7FFFFFFF, r2 = r1;
            if (t < m) {
                const int bv = L.basis[t];
                g1 = lpx_psd(L.xB[t] - L.lb[bv], sign * di, false);
                g2 = lpx_psd(L.ub[bv] - L.xB[t], -sign * di, false);
            }
            batch_argmin_max(g1, r1, dummy, red);
            batch_argmin_max(g2, r2, dummy, red);
            const double g3 = L.ub[j] - L.lb[j];
            const double delta = g1 < g2 ? g1 : g2;
            if (delta == LPX_INF) { // :256-257 / :302-303
                status = LPX_PRIMAL_UNBOUNDED;
                last_j = j;
                break;
            }
            if (g3 <= delta) flip = true; // :264-268 / :309-313
            else if (g1 < g2) { r = r1; lstate = LPX_VS_LOWER; } // :271-277
            else { r = r2; lstate = LPX_VS_UPPER; }              // :278-284
        } else {
            if (KIND == LPX_KIND_DUAL_BOUNDED) lstate = below ? LPX_VS_LOWER : LPX_VS_UPPER; // the bound it violated
            __syncthreads(); // L.d complete
        }
        BS(3)
        int leaving = -1;
        if (!flip) {
            // ---- eta column (:236-244), copy of the old pivot row, rank-1 update of B^-1 (:247-249)
            const double dr = L.d[r], xr = L.xB[r];
            leaving = L.basis[r];
            if (t < m) {
                L.e[t] = (t == r) ? 1.0 / dr : (-di) / dr;
                L.rowr[t] = L.Binv[r * ldb + t];
            }
            __syncthreads();
            {
                // rows of B^-1 are 16-byte aligned here: a lane updates PAIRS of neighbouring columns (one 16-byte load and
                // store per two elements, the pivot row's pair loaded once when a lane keeps its columns); 8 pairs per lane
                // at a time, every load before the first store.  A last odd column pairs with the row's padding word.
                const int P = (m + 1) >> 1, total = m * P;
                int i = t / P, pp = t - i * P;
                const int di2 = LPX_BLOCK / P, dp2 = LPX_BLOCK - di2 * P;
                const bool fixp = dp2 == 0;
                const d2_t rk0 = *(const d2_t *)(L.rowr + 2 * (pp < P ? pp : 0));
                for (int idx = t; idx < total; idx += 8 * LPX_BLOCK) {
                    double ei[8];
                    d2_t rk[8], bv[8];
                    int off[8], row[8];
#pragma unroll
                    for (int u = 0; u < 8; ++u) {
                        const bool ok = idx + u * LPX_BLOCK < total;
                        row[u] = ok ? i : -1;
                        off[u] = ok ? i * ldb + 2 * pp : 0;
                        ei[u] = L.e[ok ? i : 0];
                        rk[u] = fixp ? rk0 : *(const d2_t *)(L.rowr + (ok ? 2 * pp : 0));
                        bv[u] = *(const d2_t *)(L.Binv + off[u]);
                        i += di2; pp += dp2;
                        if (pp >= P) { pp -= P; i += 1; }
                    }
#pragma unroll
                    for (int u = 0; u < 8; ++u) {
                        if (row[u] >= 0) {
                            d2_t o;
                            o[0] = (row[u] == r) ? ei[u] * rk[u][0] : fma(ei[u], rk[u][0], bv[u][0]);
                            o[1] = (row[u] == r) ? ei[u] * rk[u][1] : fma(ei[u], rk[u][1], bv[u][1]);
                            *(d2_t *)(L.Binv + off[u]) = o;
                        }
                    }
                }
            }
            if (!BND && t < m) L.xB[t] = (t == r) ? L.e[t] * xr : fma(L.e[t], xr, L.xB[t]); // :251-253
        }
        BS(4)
        if (t == 0) {
            if (s.log && pivots < s.log_cap) {
                int64_t *lg = s.log + pivots * 3;
                lg[0] = flip ? -1 : r; lg[1] = j; lg[2] = leaving;
            }
            if (flip) {
                L.vstate[j] = (sign > 0.0) ? LPX_VS_UPPER : LPX_VS_LOWER;
            } else {
                L.basis[r] = j; // :261
                L.cb[r] = L.c[j];
                L.vstate[leaving] = lstate;
                L.vstate[j] = LPX_VS_BASIC;
            }
        }
        pivots += 1;
        last_j = j; last_r = flip ? -1 : r; last_leaving = leaving; last_flip = flip ? 1 : 0; last_lstate = lstate;
        if (!flip) f32 = false;
        __syncthreads();
        if (BND) bounded_update_bfs(); // special_solvers.py:341
    }
    if (PD) {
        __syncthreads();
        if (status == LPX_PRIMAL_UNBOUNDED) {
            pd_status = LPX_PRIMAL_UNBOUNDED;
            break;
        }
        // cost of the restricted primal, dot(c_rp, x) in index order (primal_solvers.py:72-79)
        for (int col = t; col < n; col += LPX_BLOCK) L.rc[col] = 0.0;
        __syncthreads();
        if (t < m) L.rc[L.basis[t]] = L.xB[t];
        __syncthreads();
        if (t == 0) {
            double s1 = 0.0;
            for (int col = 0; col < n; ++col) s1 = fma((double)L.c[col], L.rc[col], s1);
            sh_cost = s1;
        }
        __syncthreads();
        if (!(sh_cost > 0.0)) { // :445-447: complementary slackness attained
            pd_status = LPX_OPTIMAL;
            break;
        }
        if (t < m) L.y[t] = batch_dot<double>(L.cb, 1, L.Binv + t, ldb, m); // y_rp = c_B B^-1 (:430-433, float64 inverse)
        __syncthreads();
        double th = LPX_INF, um = -LPX_INF;
        for (int col = t; col < n_orig; col += LPX_BLOCK) {
            const double u = COLDOT(double, L.y, col);
            if (L.vstate[col] == LPX_VS_EXCLUDED) th = fmin(th, lpx_psd(a.pd_c64[col] - L.v[col], u, false)); // :438-443
            else um = fmax(um, u);
        }
        if (t < m) um = fmax(um, L.y[t]); // the artificial block of A_rp: y_rp @ I
        int dummy = 0;
        batch_argmin_max(th, dummy, um, red);
        if (um <= 0.0) { // :435-436
            pd_status = LPX_DUAL_UNBOUNDED;
            break;
        }
        if (t < m) L.yd[t] = L.yd[t] + th * L.y[t]; // :444 (theta * y_rp, then +=: two roundings)
        __syncthreads();
    }
    } while (PD && outer < a.pd_maxiters1);
    BS(6)
    BS_PRINT
    if (PD) {
        status = pd_status;
        counter = outer;
        if (t < m) a.pd_y[t] = L.yd[t];
    }

    // ---- state back to HBM
    __syncthreads();
    for (int idx = t; idx < m * m; idx += LPX_BLOCK) {
        const int i = idx / m, k = idx - i * m;
        s.Binv[(int64_t)i * s.ldB + k] = L.Binv[i * ldb + k];
    }
    for (int col = t; col < n; col += LPX_BLOCK) s.var_state[col] = L.vstate[col];
    if (t < m) {
        s.basis[t] = L.basis[t];
        s.xB[t] = L.xB[t];
    }
    if (a.x != nullptr) { // result assembly, primal_solvers.py:72-79 / special_solvers.py:201-206; cost in index order
        for (int col = t; col < n; col += LPX_BLOCK) {
            const int vs = L.vstate[col];
            L.rc[col] = (BND && vs == LPX_VS_LOWER) ? L.lb[col] : (BND && vs == LPX_VS_UPPER) ? L.ub[col] : 0.0;
        }
        __syncthreads();
        if (t < m) L.rc[L.basis[t]] = L.xB[t];
        __syncthreads();
        for (int col = t; col < n; col += LPX_BLOCK) a.x[(int64_t)blockIdx.x * n + col] = L.rc[col];
    }
    if (t == 0) {
        if (a.cost != nullptr) {
            double s1 = 0.0;
            for (int col = 0; col < n; ++col) s1 = fma((double)L.c[col], L.rc[col], s1);
            a.cost[blockIdx.x] = s1;
        }
        ctrl->status = status;
        ctrl->f32_phase = f32 ? 1 : 0;
        ctrl->counter = counter;
        ctrl->pivots = pivots;
        ctrl->enter_j = last_j;
        ctrl->leave_r = last_r;
        ctrl->leave_var = last_leaving;
        ctrl->price_val = last_price;
        ctrl->flip = last_flip;
        ctrl->leave_state = last_lstate;
    }
}

#undef COLDOT
#undef COLDOT_U

extern "C" int64_t lpx_revised_small_lds_bytes(int64_t m, int64_t n, int kind)
{
    if (m <= 0 || n <= 0) return 0;
    return block_lds(m, n, kind, nullptr).bytes;
}

extern "C" int lpx_revised_small_a_in_lds(int64_t m, int64_t n, int kind)
{
    if (m <= 0 || n <= 0) return 0;
    const int64_t bytes = block_lds(m, n, kind, nullptr).bytes;
    return (bytes <= LPX_BLOCK_LDS_MAX && block_A_fits(bytes, m, n, kind)) ? 1 : 0;
}

static int block_launch(const BlockArgs &a0, int64_t batch, int64_t m, int64_t n, int kind, hipStream_t st)
{
    LPX_REQUIRE(kind == LPX_KIND_PRIMAL || kind == LPX_KIND_DUAL || kind == LPX_KIND_BOUNDED || kind == LPX_KIND_DUAL_BOUNDED,
                "kind must be primal, dual, bounded or dual-bounded");
    int64_t bytes = block_lds(m, n, kind, nullptr).bytes;
    if (bytes > LPX_BLOCK_LDS_MAX || m > 4096 || n > 65536) {
        lpx_set_error("lpx_revised_solve_small: the state of a %lld x %lld LP needs %lld B of LDS, one CU has %d", (long long)m,
                      (long long)n, (long long)bytes, LPX_BLOCK_LDS_MAX);
        return LPX_E_SIZE;
    }
    BlockArgs a = a0;
    a.a_in_lds = block_A_fits(bytes, m, n, kind) ? 1 : 0;
    if (a.a_in_lds) bytes += block_lds_A_bytes(m, n, kind);
#define LPX_BLOCK_GO(K)                                                                                                      \
    {                                                                                                                        \
        LPX_HIP(hipFuncSetAttribute((const void *)k_block_solve<K>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes)); \
        hipLaunchKernelGGL((k_block_solve<K>), dim3((unsigned)batch), dim3(LPX_BLOCK), (size_t)bytes, st, a);                \
    }
    if (kind == LPX_KIND_PRIMAL) LPX_BLOCK_GO(LPX_KIND_PRIMAL)
    else if (kind == LPX_KIND_DUAL) LPX_BLOCK_GO(LPX_KIND_DUAL)
    else if (kind == LPX_KIND_DUAL_BOUNDED) LPX_BLOCK_GO(LPX_KIND_DUAL_BOUNDED)
    else LPX_BLOCK_GO(LPX_KIND_BOUNDED)
#undef LPX_BLOCK_GO
    LPX_LAUNCH_CHECK();
    return LPX_OK;
}

extern "C" int lpx_revised_solve_small(const lpx_revised *s, int kind, int64_t maxiters, void *stream)
{
    LPX_REQUIRE(s != nullptr && s->m > 0 && s->n > 0 && maxiters >= 0, "bad state / sizes");
    LPX_REQUIRE(s->A && s->b && s->c && s->Binv && s->xB && s->basis && s->var_state && s->ctrl, "null device pointer");
    LPX_REQUIRE(s->ldA >= s->n && s->ldB >= s->m, "leading dimensions too small");
    LPX_REQUIRE((kind != LPX_KIND_BOUNDED && kind != LPX_KIND_DUAL_BOUNDED) || (s->lb && s->ub), "the bounded solvers need lb and ub");
    BlockArgs a{};
    a.single = *s;
    a.states = nullptr;
    a.maxiters = maxiters;
    return block_launch(a, 1, s->m, s->n, kind, (hipStream_t)stream);
}

extern "C" int lpx_pd_solve_small(const lpx_revised *s, double *y, const double *c64, int64_t n_orig, int64_t maxiters1,
                                  int64_t maxiters2, double *x, double *cost, void *stream)
{
    LPX_REQUIRE(s != nullptr && s->m > 0 && n_orig > 0 && s->n == n_orig + s->m, "the state must be the restricted primal [A | I]");
    LPX_REQUIRE(s->A && s->At && s->b && s->c && s->Binv && s->xB && s->basis && s->var_state && s->ctrl, "null device pointer");
    LPX_REQUIRE(y && c64 && x && cost && maxiters1 >= 0 && maxiters2 >= 0, "bad arguments");
    int64_t bytes = block_lds(s->m, s->n, LPX_KIND_PD, nullptr).bytes;
    if (bytes > LPX_BLOCK_LDS_MAX) {
        lpx_set_error("lpx_pd_solve_small: the state of a %lld x %lld LP needs %lld B of LDS, one CU has %d", (long long)s->m,
                      (long long)n_orig, (long long)bytes, LPX_BLOCK_LDS_MAX);
        return LPX_E_SIZE;
    }
    if (maxiters1 == 0) return LPX_OK;
    BlockArgs a{};
    a.single = *s;
    a.maxiters = maxiters2;
    a.x = x;
    a.cost = cost;
    a.pd_y = y;
    a.pd_c64 = c64;
    a.pd_n_orig = n_orig;
    a.pd_maxiters1 = maxiters1;
    a.a_in_lds = block_A_fits(bytes, s->m, s->n, LPX_KIND_PRIMAL) ? 1 : 0;
    if (a.a_in_lds) bytes += block_lds_A_bytes(s->m, s->n, LPX_KIND_PRIMAL);
    LPX_HIP(hipFuncSetAttribute((const void *)k_block_solve<LPX_KIND_PRIMAL, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes));
    hipLaunchKernelGGL((k_block_solve<LPX_KIND_PRIMAL, true>), dim3(1), dim3(LPX_BLOCK), (size_t)bytes, (hipStream_t)stream, a);
    LPX_LAUNCH_CHECK();
    return LPX_OK;
}
extern "C" int64_t lpx_pd_small_lds_bytes(int64_t m, int64_t n_orig)
{
    if (m <= 0 || n_orig <= 0) return 0;
    return block_lds(m, n_orig + m, LPX_KIND_PD, nullptr).bytes;
}

extern "C" int lpx_revised_solve_small_batch(const lpx_revised *dev_states, int64_t batch, int64_t m, int64_t n, int kind,
                                             int64_t maxiters, double *x, double *cost, void *stream)
{
    LPX_REQUIRE(dev_states != nullptr && batch > 0 && m > 0 && n > 0 && maxiters >= 0, "bad arguments");
    LPX_REQUIRE((x == nullptr) == (cost == nullptr), "x and cost come together");
    BlockArgs a{};
    a.states = dev_states;
    a.maxiters = maxiters;
    a.x = x;
    a.cost = cost;
    return block_launch(a, batch, m, n, kind, (hipStream_t)stream);
}
