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
#define LPX_STAMP_FLUSH                                                             \
    if (threadIdx.x == 0)                                                           \
        for (int q = 0; q < 8; ++q) atomicAdd(&g_batch_stamps[q], st_acc[q]);
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

__global__ __launch_bounds__(LPX_BLOCK) void k_batched_primal(BatchArgs a)
{
    extern __shared__ __attribute__((aligned(16))) char smem[];
    __shared__ BatchRed red;
    const int m = (int)a.m, n = (int)a.n, t = threadIdx.x;
    const int64_t lp = blockIdx.x;
    const BatchLds L = batch_lds(m, n, smem);
    const int ldb = (int)L.ldb;

    // ---- load the LP into LDS
    const float *gA = a.A + lp * (int64_t)m * n;
    for (int idx = t; idx < m * n; idx += LPX_BLOCK) L.A[idx] = gA[idx];
    for (int idx = t; idx < n; idx += LPX_BLOCK) {
        L.c[idx] = a.c[lp * n + idx];
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
    if (t < m) L.basis[t] = (int)a.basis[lp * m + t];
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
// Shape-specialised kernel for m in {8, 16, 32, 64}, n <= 256.
//
// Same arithmetic in the same order as k_batched_primal (bit-identical results), laid out for the critical path,
// which is three chains of m dependent fmas per pivot (y, rc, d):
//  * m is a template parameter: the chains are straight-line code, every LDS address an immediate, so a step
//    costs one fma plus ~one load instead of a dozen address/convert/move instructions;
//  * lane t keeps column t of A in registers for the whole solve (A is read from HBM once, coalesced, and never
//    again; the rc chain reads only y from LDS); LDS holds B^-1 and a few vectors, ~37 KB at m = 64, so four
//    workgroups share a CU and their chains interleave;
//  * each wave publishes the column of its own first negative reduced cost next to its partial minimum, which
//    folds "broadcast the entering column" into the barrier of the pricing reduction;
//  * direction, ratio test, eta column and x_B live in wave 0 (m <= 64 rows): that reduction is pure DPP, no barrier;
//  * four barriers per pivot.
template <int M>
struct __attribute__((aligned(16))) FixedLds {
    static constexpr int LDB = M + 1; // row pad: column reads hit distinct banks
    double Binv[M * LDB];
    double y[M], e[M], rowr[M], cbd[M]; // cbd[i] = c[basis[i]]
    double xs[LPX_BLOCK];
    float cf[LPX_BLOCK];
    float acol[LPX_WAVES][M]; // candidate entering columns, one slot per wave
    double red_mn[LPX_WAVES];
    int red_j[LPX_WAVES];
    int vst[LPX_BLOCK];
    int r, leave, stat, pad;
};

template <int M>
__global__ __launch_bounds__(LPX_BLOCK) void k_batched_fixed(BatchArgs a)
{
    constexpr int LDB = M + 1;
    constexpr int ACTIVE = M * M < LPX_BLOCK ? M * M : LPX_BLOCK; // lanes of the rank-1 update
    constexpr int RPP = ACTIVE / M;                                // rows covered per pass
    constexpr int CNT = M / RPP;                                   // B^-1 entries per lane
    static_assert(M % 4 == 0 && M <= 64 && LPX_BLOCK % M == 0, "m must divide the workgroup and fit one wave");
    __shared__ FixedLds<M> S;
    const int n = (int)a.n, t = threadIdx.x, wv = t >> 6;
    const int64_t lp = blockIdx.x;
    const bool has_col = t < n, has_row = t < M;

    // ---- column t of A and c_t into registers; B^-1, basis, c_B into LDS
    float Ac[M];
    float ct = 0.f;
    {
        const float *gA = a.A + lp * (int64_t)M * n;
#pragma unroll
        for (int k = 0; k < M; ++k) Ac[k] = has_col ? gA[(int64_t)k * n + t] : 0.f;
        if (has_col) ct = a.c[lp * n + t];
    }
    S.cf[t] = ct;
    S.vst[t] = 0;
    for (int idx = t; idx < M * M; idx += LPX_BLOCK) {
        const int i = idx / M, k = idx % M;
        S.Binv[i * LDB + k] = a.Binv0 ? (double)a.Binv0[lp * (int64_t)M * M + idx] : (i == k ? 1.0 : 0.0);
    }
    int bas = 0;
    if (has_row) bas = (int)a.basis[lp * M + t];
    if (t == 0) S.stat = LPX_RUNNING;
    __syncthreads();
    double xB = 0.0;
    if (has_row) {
        S.vst[bas] = 1;
        S.cbd[t] = (double)S.cf[bas];
        // bfs = inv_basis_matrix @ b in float32 (primal_solvers.py:38)
        float s0 = 0.f;
#pragma unroll
        for (int k = 0; k < M; ++k) s0 = fmaf((float)S.Binv[t * LDB + k], a.b[lp * M + k], s0);
        xB = (double)s0;
    }
    __syncthreads();
    bool basic = S.vst[t] != 0;

    int status = LPX_RUNNING;
    int64_t counter = 0, pivots = 0;
    LPX_STAMP_DECL

    // one iteration of the loop at primal_solvers.py:191; false when the loop ends
    auto iterate = [&](auto tag) -> bool {
        using AccT = decltype(tag);
        constexpr bool F32 = sizeof(AccT) == 4;
        LPX_STAMP(7)
        // ---- y_t = (c_B @ B^-1)_t (:64, first product): lane t of wave 0, i ascending
        if (has_row) {
            AccT acc = (AccT)0;
#pragma unroll
            for (int i = 0; i < M; ++i) acc = fma((AccT)S.cbd[i], (AccT)S.Binv[i * LDB + t], acc);
            S.y[t] = (double)acc;
        }
        LPX_STAMP(0)
        __syncthreads();
        LPX_STAMP(1)
        // ---- rc_t = c_t - (y @ A)_t ; rc[basis] = 0 (:64-65); min and first negative (:135-142)
        double mn = LPX_INF;
        int j = 0x7FFFFFFF;
        if (has_col) {
            AccT acc = (AccT)0;
#pragma unroll
            for (int k = 0; k < M; ++k) acc = fma((AccT)S.y[k], (AccT)Ac[k], acc);
            double rc;
            if constexpr (F32) rc = (double)(ct - acc);
            else rc = (double)ct - acc;
            if (basic) rc = 0.0;
            mn = rc;
            if (rc < 0.0) j = t;
        }
        LPX_STAMP(2)
        {
            const BatchVI w = wave_allreduce(BatchVI{mn, j, 0}, [](BatchVI x, BatchVI y) {
                return BatchVI{fmin(x.v, y.v), y.i < x.i ? y.i : x.i, 0};
            });
            if (w.i == t) { // this wave's candidate for the entering column
#pragma unroll
                for (int k = 0; k < M; k += 4) *(f4_t *)&S.acol[wv][k] = f4_t{Ac[k], Ac[k + 1], Ac[k + 2], Ac[k + 3]};
            }
            if ((t & 63) == 0) { S.red_mn[wv] = w.v; S.red_j[wv] = w.i; }
        }
        __syncthreads();
        mn = S.red_mn[0]; j = S.red_j[0];
#pragma unroll
        for (int k = 1; k < LPX_WAVES; ++k) {
            mn = fmin(mn, S.red_mn[k]);
            j = S.red_j[k] < j ? S.red_j[k] : j;
        }
        LPX_STAMP(3)
        if (lpx_is_optimal_min(mn)) { // :140-142
            status = LPX_OPTIMAL;
            return false;
        }
        // ---- wave 0: d_t = (B^-1 @ A[:, j])_t (:129), ratios (:165), max(d) (:146), eta column (:236-244),
        //      copy of the old pivot row, bfs <- E @ bfs
        if (wv == 0) {
            double di = -LPX_INF, th = LPX_INF;
            int r = 0x7FFFFFFF;
            if (has_row) {
                const float *col = S.acol[(j >> 6) & (LPX_WAVES - 1)];
                AccT acc = (AccT)0;
#pragma unroll
                for (int k = 0; k < M; ++k) acc = fma((AccT)S.Binv[t * LDB + k], (AccT)col[k], acc);
                di = (double)acc;
                th = lpx_psd(xB, di, F32);
                r = t;
            }
            LPX_STAMP(4)
            const BatchVIW w = wave_allreduce(BatchVIW{th, di, r, 0}, [](BatchVIW x, BatchVIW y) {
                const bool take = y.v < x.v || (y.v == x.v && y.i < x.i); // np.argmin: first minimum (:166)
                return BatchVIW{take ? y.v : x.v, fmax(x.w, y.w), take ? y.i : x.i, 0};
            });
            LPX_STAMP(5)
            if (w.w <= 0.0) { // :144-147
                if (t == 0) S.stat = LPX_PRIMAL_UNBOUNDED;
            } else {
                r = __builtin_amdgcn_readfirstlane(w.i);
                const double dr = __shfl(di, r), xr = __shfl(xB, r);
                if (has_row) {
                    const double e = (t == r) ? 1.0 / dr : (-di) / dr;
                    S.e[t] = e;
                    S.rowr[t] = S.Binv[r * LDB + t];
                    xB = (t == r) ? e * xr : fma(e, xr, xB);
                    if (t == r) {
                        if (a.log && pivots < a.log_cap) {
                            int64_t *lg = a.log + (lp * a.log_cap + pivots) * 3;
                            lg[0] = r; lg[1] = j; lg[2] = bas;
                        }
                        S.r = r;
                        S.leave = bas;
                        bas = j; // :261
                    }
                }
            }
        }
        __syncthreads();
        LPX_STAMP(6)
        if (S.stat != LPX_RUNNING) {
            status = S.stat;
            return false;
        }
        // ---- B^-1 <- E @ B^-1 (:247-253) as a rank-1 update: lane owns column k and every RPP-th row
        const int r = S.r;
        if (t == S.leave) basic = false;
        if (t == j) {
            basic = true;
            S.cbd[r] = (double)ct;
        }
        if (t < ACTIVE) {
            const int k = t % M, i0 = t / M;
            const double pr = S.rowr[k];
            double ei[CNT], bv[CNT];
#pragma unroll
            for (int q = 0; q < CNT; ++q) {
                ei[q] = S.e[i0 + q * RPP];
                bv[q] = S.Binv[(i0 + q * RPP) * LDB + k];
            }
#pragma unroll
            for (int q = 0; q < CNT; ++q) bv[q] = (i0 + q * RPP == r) ? ei[q] * pr : fma(ei[q], pr, bv[q]);
#pragma unroll
            for (int q = 0; q < CNT; ++q) S.Binv[(i0 + q * RPP) * LDB + k] = bv[q];
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
    S.xs[t] = 0.0;
    __syncthreads();
    if (has_row) S.xs[bas] = xB;
    __syncthreads();
    if (has_col) a.x[lp * n + t] = S.xs[t];
    if (has_row) a.basis[lp * M + t] = bas;
    if (t == 0) {
        double s1 = 0.0;
        for (int jx = 0; jx < n; ++jx) s1 = fma((double)S.cf[jx], S.xs[jx], s1);
        a.cost[lp] = s1;
        a.status[lp] = status;
        a.iters[lp] = counter;
        a.pivots[lp] = pivots;
    }
}

template <int M>
static void batched_fixed_launch(const BatchArgs &a, hipStream_t stream)
{
    hipLaunchKernelGGL(k_batched_fixed<M>, dim3((unsigned)a.batch), dim3(LPX_BLOCK), 0, stream, a);
}

extern "C" int lpx_batched_primal_solve(int64_t batch, int64_t m, int64_t n, const float *A, const float *b, const float *c,
                                        const float *Binv0, int64_t *basis, double *x, double *cost, int32_t *status,
                                        int64_t *iters, int64_t *pivots, int64_t maxiters, int64_t *log, int64_t log_cap,
                                        void *stream)
{
    LPX_REQUIRE(batch > 0 && m > 0 && n > 0 && maxiters >= 0, "bad sizes");
    LPX_REQUIRE(A && b && c && basis && x && cost && status && iters && pivots, "null pointer");
    LPX_REQUIRE(log == nullptr || log_cap > 0, "log without capacity");
    if (m > LPX_BLOCK || n > LPX_BLOCK) {
        lpx_set_error("lpx_batched_primal_solve: m and n must be <= %d (one lane per row/column)", LPX_BLOCK);
        return LPX_E_SIZE;
    }
    BatchArgs a{batch, m, n, A, b, c, Binv0, basis, x, cost, status, iters, pivots, maxiters, log, log_cap};
    const char *generic = getenv("LPX_BATCH_GENERIC"); // diagnostics: force the shape-generic kernel
    if (!(generic && generic[0] == '1')) {
        bool done = true;
        switch (m) {
        case 8: batched_fixed_launch<8>(a, (hipStream_t)stream); break;
        case 16: batched_fixed_launch<16>(a, (hipStream_t)stream); break;
        case 32: batched_fixed_launch<32>(a, (hipStream_t)stream); break;
        case 64: batched_fixed_launch<64>(a, (hipStream_t)stream); break;
        default: done = false;
        }
        if (done) {
            LPX_LAUNCH_CHECK();
            return LPX_OK;
        }
    }
    const int64_t bytes = batch_lds(m, n, nullptr).bytes;
    if (bytes > 158 * 1024) {
        lpx_set_error("lpx_batched_primal_solve: LP needs %lld B of LDS, more than one CU has", (long long)bytes);
        return LPX_E_SIZE;
    }
    static int64_t configured = 0;
    if (bytes > configured) {
        LPX_HIP(hipFuncSetAttribute((const void *)k_batched_primal, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes));
        configured = bytes;
    }
    hipLaunchKernelGGL(k_batched_primal, dim3((unsigned)batch), dim3(LPX_BLOCK), (size_t)bytes, (hipStream_t)stream, a);
    LPX_LAUNCH_CHECK();
    return LPX_OK;
}
