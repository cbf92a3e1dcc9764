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
    float *c;
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
    L.basis = (int *)take(4 * m);
    L.vstate = (int *)take(4 * n);
    L.bytes = (int64_t)(p - base);
    return L;
}

template <typename AccT>
__device__ __forceinline__ void batch_price(const BatchLds &L, int m, int n, int t)
{
    // y = c_B @ B^-1 (primal_solvers.py:64, first product), one lane per column, i ascending
    if (t < m) {
        AccT acc = (AccT)0;
        for (int i = 0; i < m; ++i) acc = fma((AccT)L.c[L.basis[i]], (AccT)L.Binv[i * L.ldb + t], acc);
        L.y[t] = (double)acc;
    }
    __syncthreads();
    // rc = c - y @ A ; rc[basis] = 0 (:64-65)
    if (t < n) {
        AccT acc = (AccT)0;
        for (int i = 0; i < m; ++i) acc = fma((AccT)L.y[i], (AccT)L.A[i * n + t], acc);
        double rc = (double)((AccT)L.c[t] - acc);
        if (L.vstate[t] == LPX_VS_BASIC) rc = 0.0;
        L.rc[t] = rc;
    }
    __syncthreads();
}

template <typename AccT>
__device__ __forceinline__ void batch_direction(const BatchLds &L, int m, int n, int t, int j)
{
    // d = B^-1 @ A[:, j] (:129), one lane per row, k ascending
    if (t < m) {
        AccT acc = (AccT)0;
        for (int k = 0; k < m; ++k) acc = fma((AccT)L.Binv[t * L.ldb + k], (AccT)L.A[k * n + j], acc);
        L.d[t] = (double)acc;
    }
    __syncthreads();
}

__global__ __launch_bounds__(LPX_BLOCK) void k_batched_primal(BatchArgs a)
{
    extern __shared__ __attribute__((aligned(16))) char smem[];
    __shared__ BlockRed red;
    const int m = (int)a.m, n = (int)a.n, t = threadIdx.x;
    const int64_t lp = blockIdx.x;
    const BatchLds L = batch_lds(m, n, smem);

    // ---- load the LP into LDS
    const float *gA = a.A + lp * (int64_t)m * n;
    for (int idx = t; idx < m * n; idx += LPX_BLOCK) L.A[idx] = gA[idx];
    for (int idx = t; idx < n; idx += LPX_BLOCK) {
        L.c[idx] = a.c[lp * n + idx];
        L.vstate[idx] = LPX_VS_LOWER;
    }
    for (int idx = t; idx < m * m; idx += LPX_BLOCK) {
        const int i = idx / m, k = idx - i * m;
        L.Binv[i * L.ldb + k] = a.Binv0 ? (double)a.Binv0[lp * (int64_t)m * m + idx] : (i == k ? 1.0 : 0.0);
    }
    if (t < m) L.basis[t] = (int)a.basis[lp * m + t];
    __syncthreads();
    if (t < m) L.vstate[L.basis[t]] = LPX_VS_BASIC;
    // bfs = inv_basis_matrix @ b in float32 (primal_solvers.py:38)
    if (t < m) {
        float acc = 0.f;
        for (int k = 0; k < m; ++k) acc = fmaf((float)L.Binv[t * L.ldb + k], a.b[lp * m + k], acc);
        L.xB[t] = (double)acc;
    }
    __syncthreads();

    bool f32 = true;
    int status = LPX_RUNNING;
    int64_t counter = 0, pivots = 0;
    while (counter < a.maxiters) { // primal_solvers.py:191
        counter += 1;
        if (f32) batch_price<float>(L, m, n, t);
        else batch_price<double>(L, m, n, t);
        const double rcv = (t < n) ? L.rc[t] : LPX_INF;
        const double mn = block_min(rcv, red);
        const int j = (int)block_imin((t < n && rcv < 0.0) ? t : LPX_NOIDX, red);
        if (lpx_is_optimal_min(mn)) { // :140-142
            status = LPX_OPTIMAL;
            break;
        }
        if (f32) batch_direction<float>(L, m, n, t, j);
        else batch_direction<double>(L, m, n, t, j);
        const double di = (t < m) ? L.d[t] : -LPX_INF;
        VI th{(t < m) ? lpx_psd(L.xB[t], di, f32) : LPX_INF, (t < m) ? t : LPX_NOIDX};
        th = block_argmin(th, red); // :165-166
        const double dmax = block_max(di, red);
        if (dmax <= 0.0) { // :144-147
            status = LPX_PRIMAL_UNBOUNDED;
            break;
        }
        const int r = (int)th.i;
        const double dr = L.d[r];
        if (t < m) {
            L.e[t] = (t == r) ? 1.0 / dr : (-L.d[t]) / dr; // :236-244
            L.rowr[t] = L.Binv[r * L.ldb + t];
        }
        const double xr = L.xB[r];
        __syncthreads();
        // B^-1 <- E @ B^-1 ; bfs <- E @ bfs (:247-253) as a rank-1 update
        for (int idx = t; idx < m * m; idx += LPX_BLOCK) {
            const int i = idx / m, k = idx - i * m;
            const double ei = L.e[i];
            double *p = &L.Binv[i * L.ldb + k];
            *p = (i == r) ? ei * L.rowr[k] : fma(ei, L.rowr[k], *p);
        }
        if (t < m) L.xB[t] = (t == r) ? L.e[t] * xr : fma(L.e[t], xr, L.xB[t]);
        if (t == 0) {
            const int leaving = L.basis[r];
            if (a.log && pivots < a.log_cap) {
                int64_t *lg = a.log + (lp * a.log_cap + pivots) * 3;
                lg[0] = r; lg[1] = j; lg[2] = leaving;
            }
            L.basis[r] = j; // :261
            L.vstate[leaving] = LPX_VS_LOWER;
            L.vstate[j] = LPX_VS_BASIC;
        }
        pivots += 1;
        f32 = false;
        __syncthreads();
    }

    // ---- result: x = scatter(x_B), cost = dot(c, x) in index order (primal_solvers.py:72-79)
    __syncthreads();
    for (int idx = t; idx < n; idx += LPX_BLOCK) L.rc[idx] = 0.0;
    __syncthreads();
    if (t < m) L.rc[L.basis[t]] = L.xB[t];
    __syncthreads();
    for (int idx = t; idx < n; idx += LPX_BLOCK) a.x[lp * n + idx] = L.rc[idx];
    if (t < m) a.basis[lp * m + t] = L.basis[t];
    if (t == 0) {
        double acc = 0.0;
        for (int jx = 0; jx < n; ++jx) acc = fma((double)L.c[jx], L.rc[jx], acc);
        a.cost[lp] = acc;
        a.status[lp] = status;
        a.iters[lp] = counter;
        a.pivots[lp] = pivots;
    }
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
    BatchArgs a{batch, m, n, A, b, c, Binv0, basis, x, cost, status, iters, pivots, maxiters, log, log_cap};
    hipLaunchKernelGGL(k_batched_primal, dim3((unsigned)batch), dim3(LPX_BLOCK), (size_t)bytes, (hipStream_t)stream, a);
    LPX_LAUNCH_CHECK();
    return LPX_OK;
}
