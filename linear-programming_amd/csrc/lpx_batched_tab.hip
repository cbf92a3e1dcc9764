// lpx_batched_tab.hip -- batches of small LPs in CONDENSED TABLEAU form, one wave per LP, the tableau in registers.
//
// Same rules as lpx_batched_primal_solve (primal_solvers.py:135-167 per LP), different carried state: instead of B^-1
// (three chains of m dependent fmas per pivot: y, rc, d, plus the update) the wave carries the nonbasic columns of
// T = B^-1 A, the reduced-cost row and x_B -- lpx_tableau's condensed layout -- and a pivot is ONE rank-1 update with one
// fma per element.  Lane t owns column slot t: its m entries and its reduced cost live in registers for the whole solve
// (130 VGPRs at m = 64); lane t also owns x_B[t] and basis[t].  Per pivot: two ballots price the row (any rc < 0?  any
// rc < -1e-8?), a wave min over variable ids gives the first negative index, the entering lane publishes its column
// through LDS, lane t forms its ratio and its eta entry e_t, and every lane updates its own column with the e_i read
// back from LDS.  No dependent chains, ~1 KB of LDS per LP (so occupancy is set by registers, not LDS).
// Arithmetic = the tableau loop's (lpx_tableau.hip, lpo_tableau_solve): bit-identical to that oracle, the reference's pivot
// sequences.  Starting basis: the slack block (identity columns n-m .. n-1) with zero basic costs, checked by the caller.
#include "lpx_device.hpp"

#define TB_MAX 64
// one case per row: r is wave-uniform, so "register r of this lane" is a scalar branch, not 64 selects
#define TB_ROWS8(F, b) F(b + 0) F(b + 1) F(b + 2) F(b + 3) F(b + 4) F(b + 5) F(b + 6) F(b + 7)
#define TB_ROWS(F) TB_ROWS8(F, 0) TB_ROWS8(F, 8) TB_ROWS8(F, 16) TB_ROWS8(F, 24) TB_ROWS8(F, 32) TB_ROWS8(F, 40) TB_ROWS8(F, 48) TB_ROWS8(F, 56)

// value of v in lane `lane` (wave-uniform, compile-time or scalar) as a scalar: two v_readlane_b32, no LDS round trip
__device__ __forceinline__ double tb_readlane(double v, int lane)
{
    const long long b = __double_as_longlong(v);
    const unsigned lo = (unsigned)__builtin_amdgcn_readlane((int)(b & 0xffffffffLL), lane);
    const unsigned hi = (unsigned)__builtin_amdgcn_readlane((int)(b >> 32), lane);
    return __longlong_as_double((long long)(((unsigned long long)hi << 32) | lo));
}

struct TabBatchArgs {
    int64_t batch, m, n;
    const float *A, *b, *c;
    int64_t *basis;
    double *x, *cost;
    int32_t *status;
    int64_t *iters, *pivots;
    int64_t maxiters;
    int64_t *log;
    int64_t log_cap;
    int64_t na; // stored columns of A and entries of c per LP: n, or n - m for the inequality form (G and its costs only)
    // general starting basis (else NULL): the starting tableau itself, formed by the caller with the reference's float32
    // expressions (primal_solvers.py:37-38, :64) -- T0 = (inv(A_B) A)[:, nonbasic] (m x k per LP), its reduced costs, x_B and
    // the variable of every column slot; basis is then an input as well
    const float *T0, *rc0, *xb0;
    const int64_t *var0;
};

// MR = rows carried per lane (m rounded up to 16): smaller LPs do proportionally less work and fit more waves per SIMD
// DUAL: the dual simplex rules (dual_solvers.py:47-108) on the same carried state; the starting basis is then the slack
// block with +-1 on its diagonal (-1 where preprocessing flipped a row with b < 0), x_B = B^-1 b may be negative.
template <int MR, bool DUAL>
__global__ __launch_bounds__(64) void k_batched_tab(TabBatchArgs a)
{
    __shared__ double dcol[TB_MAX + 2];
    __shared__ double xs[2 * TB_MAX];
    __shared__ float cf[2 * TB_MAX];
    const int t = threadIdx.x;
    const int64_t lp = blockIdx.x;
    const int m = (int)a.m, n = (int)a.n, k = n - m;
    const bool col = t < k, row = t < m;
    const int na = (int)a.na;
    const float *A = a.A + lp * a.m * a.na;

    const bool from_T0 = !DUAL && a.T0 != nullptr; // wave-uniform: the caller supplies the starting tableau (A, b unused)
    double x[MR];
#pragma unroll
    for (int i = 0; i < MR; ++i) {
        // B^-1 of a +-identity basis is itself: row i of B^-1 A is row i of A times B_ii (exact)
        const double sg = (DUAL && i < m) ? (double)A[(int64_t)i * na + k + i] : 1.0;
        x[i] = (col && i < m && !from_T0) ? sg * (double)A[(int64_t)i * na + t] : 0.0;
    }
    double rc = (col && !from_T0) ? (double)a.c[lp * na + t] : 0.0; // basic costs are 0: rc = c - c_B B^-1 A = c
    double xb = (row && !from_T0) ? (double)a.b[lp * m + t] : 0.0;
    if (DUAL && row) xb *= (double)A[(int64_t)t * na + k + t];
    int64_t var = col ? (int64_t)t : LPX_NOIDX; // variable held by this lane's column slot
    int64_t bas = row ? (int64_t)(k + t) : -1;  // basis[t]
    if (from_T0) {
        const float *T0 = a.T0 + lp * a.m * k;
#pragma unroll
        for (int i = 0; i < MR; ++i) x[i] = (col && i < m) ? (double)T0[(int64_t)i * k + t] : 0.0;
        rc = col ? (double)a.rc0[lp * k + t] : 0.0;
        xb = row ? (double)a.xb0[lp * m + t] : 0.0;
        var = col ? a.var0[lp * k + t] : LPX_NOIDX;
        bas = row ? a.basis[lp * m + t] : -1;
    }
    for (int j = t; j < n; j += 64) cf[j] = j < na ? a.c[lp * na + j] : 0.f;

    bool f32 = true; // the first ratio test divides in float32
    int32_t status = LPX_RUNNING;
    int64_t counter = 0, pivots = 0;
    while (counter < a.maxiters) {
        counter += 1;
        int r, jl;
        int64_t jv;
        double d, dr, pr = 0.0;
        if (!DUAL) {
            const bool neg = col && rc < 0.0;
            // optimal <=> min rc >= 0 or |min rc| <= 1e-8 (primal_solvers.py:140-142) <=> no rc below -1e-8
            if (__ballot(neg) == 0 || __ballot(col && rc < -LPX_ATOL) == 0) {
                status = LPX_OPTIMAL;
                break;
            }
            jv = wave_imin(neg ? var : LPX_NOIDX); // first negative index, by variable id (:135-138)
            jl = __builtin_amdgcn_readfirstlane(__ffsll((long long)__ballot(var == jv)) - 1);
        } else {
            const bool negx = row && xb < 0.0;
            // optimal <=> min x_B >= 0 or |min x_B| <= 1e-8 (dual_solvers.py:52-54); leaving row = first x_B < 0 (:47-50)
            const unsigned long long nb = __ballot(negx);
            if (nb == 0 || __ballot(row && xb < -LPX_ATOL) == 0) {
                status = LPX_OPTIMAL;
                break;
            }
            r = __builtin_amdgcn_readfirstlane(__ffsll((long long)nb) - 1);
            switch (r) { // this column's entry of the dual row B^-1[r,:] A (:37-38)
#define TB_GET(i) case (i): if constexpr ((i) < MR) pr = x[(i) < MR ? (i) : 0]; break;
                TB_ROWS(TB_GET)
#undef TB_GET
            }
            if (__ballot(col && pr < 0.0) == 0) { // min(row) >= 0 (:56-59)
                status = LPX_DUAL_UNBOUNDED;
                break;
            }
            VI th{col ? lpx_dsd(rc, pr, f32) : LPX_INF, col ? var : LPX_NOIDX};
            th = wave_argmin(th); // first argmin of (-rc_j) / row_j over row_j < 0, by variable id (:61-80)
            jv = th.i;
            jl = __builtin_amdgcn_readfirstlane(__ffsll((long long)__ballot(var == jv)) - 1);
        }
        // the entering column goes from registers of one lane to one entry per lane: through LDS
        if (t == jl) {
#pragma unroll
            for (int i = 0; i < MR; ++i) dcol[i] = x[i];
        }
        const double rcj = tb_readlane(rc, jl);
        __syncthreads();
        d = row ? dcol[t] : 0.0;
        if (!DUAL) {
            if (__ballot(row && d > 0.0) == 0) { // max(d) <= 0 (:144-147)
                status = LPX_PRIMAL_UNBOUNDED;
                break;
            }
            VI th{row ? lpx_psd(xb, d, f32) : LPX_INF, row ? (int64_t)t : LPX_NOIDX};
            th = wave_argmin(th); // first argmin (:165-166)
            r = __builtin_amdgcn_readfirstlane((int)th.i);
        }
        dr = tb_readlane(d, r);
        const double e = ((t == r) ? 1.0 : -d) / dr; // :236-244: 1/d_r or (-d_i)/d_r (lanes >= m: d = 0, e = -0, rows stay 0)
        const double em = (-rcj) / dr;               // eliminates the entering column's reduced cost
        const double xr = tb_readlane(xb, r);
        if (row) xb = (t == r) ? e * xr : fma(e, xr, xb); // :251-253
        const int64_t leaving = (int64_t)__builtin_amdgcn_readlane((int)bas, r); // variable ids < 2^31 here (n <= 128)
        if (!DUAL) { // this column's pivot-row entry
            switch (r) {
#define TB_GET(i) case (i): if constexpr ((i) < MR) pr = x[(i) < MR ? (i) : 0]; break;
                TB_ROWS(TB_GET)
#undef TB_GET
            }
        }
        // One select per element covers both special cases: the pivot row (every lane) and the entering lane (every row)
        // take their addend from -0.0 -- fma(e_r, row_r, -0.0) is the plain product, and with the entering lane's
        // multiplicand set to 1 fma(e_i, 1, -0.0) is e_i itself, bit for bit.
        const bool mine = t == jl;
        const double prs = mine ? 1.0 : pr;
#pragma unroll
        for (int i = 0; i < MR; ++i) {
            const double ei = tb_readlane(e, i);
            x[i] = fma(ei, prs, (mine || i == r) ? -0.0 : x[i]);
        }
        rc = fma(em, prs, mine ? -0.0 : rc);
        if (mine) var = leaving;
        if (t == r) bas = jv; // :261
        if (a.log && t == 0 && pivots < a.log_cap) {
            int64_t *lg = a.log + (lp * a.log_cap + pivots) * 3;
            lg[0] = r;
            lg[1] = jv;
            lg[2] = leaving;
        }
        pivots += 1;
        f32 = false;
    }

    // ---- result: x = scatter(x_B), cost = dot(c, x) in index order (primal_solvers.py:72-79)
    __syncthreads();
    for (int j = t; j < n; j += 64) xs[j] = 0.0;
    __syncthreads();
    if (row) xs[bas] = xb;
    __syncthreads();
    for (int j = t; j < n; j += 64) a.x[lp * n + j] = xs[j];
    if (row) a.basis[lp * m + t] = bas;
    if (t == 0) {
        double s1 = 0.0;
        for (int jx = 0; jx < n; ++jx) s1 = fma((double)cf[jx], xs[jx], s1);
        a.cost[lp] = s1;
        a.status[lp] = status;
        a.iters[lp] = counter;
        a.pivots[lp] = pivots;
    }
}

template <bool DUAL>
static int tab_launch(const TabBatchArgs &a, void *stream)
{
    const dim3 grid((unsigned)a.batch), block(64);
    hipStream_t st = (hipStream_t)stream;
    if (a.m <= 16) hipLaunchKernelGGL((k_batched_tab<16, DUAL>), grid, block, 0, st, a);
    else if (a.m <= 32) hipLaunchKernelGGL((k_batched_tab<32, DUAL>), grid, block, 0, st, a);
    else if (a.m <= 48) hipLaunchKernelGGL((k_batched_tab<48, DUAL>), grid, block, 0, st, a);
    else hipLaunchKernelGGL((k_batched_tab<64, DUAL>), grid, block, 0, st, a);
    LPX_LAUNCH_CHECK();
    return LPX_OK;
}

// PrimalRevisedSimplexSolver(c, A, b, basis).solve(maxiters) per LP (primal_solvers.py:176-263) in condensed tableau form.
// m <= 64, n - m <= 64; starting basis = the slack block (columns n-m .. n-1 are the identity, b >= 0) with zero costs on
// it -- the caller's check.  Arrays batch-major as in lpx_batched_primal_solve; basis is written (m per LP).
extern "C" int lpx_batched_tableau_solve(int64_t batch, int64_t m, int64_t n, const float *A, const float *b, const float *c,
                                         int64_t *basis, double *x, double *cost, int32_t *status, int64_t *iters,
                                         int64_t *pivots, int64_t maxiters, int64_t *log, int64_t log_cap, void *stream)
{
    LPX_REQUIRE(batch > 0 && m > 0 && n > m && maxiters >= 0, "bad sizes");
    LPX_REQUIRE(A && b && c && basis && x && cost && status && iters && pivots, "null pointer");
    LPX_REQUIRE(log == nullptr || log_cap > 0, "log without capacity");
    LPX_REQUIRE(m <= TB_MAX && n - m <= TB_MAX, "m <= 64 and n - m <= 64 (one lane per row and per nonbasic column)");
    return tab_launch<false>(TabBatchArgs{batch, m, n, A, b, c, basis, x, cost, status, iters, pivots, maxiters, log, log_cap, n, nullptr, nullptr, nullptr, nullptr}, stream);
}

// min c.x s.t. G x <= h, x >= 0 per LP (the [G | I] standard form of preprocessing.py:17-69 with its slack basis): G is
// m x k, c has k entries, h >= 0; the identity block is never stored or read.  Results carry all k + m variables.
extern "C" int lpx_batched_tableau_solve_ineq(int64_t batch, int64_t m, int64_t k, const float *G, const float *h, const float *c,
                                              int64_t *basis, double *x, double *cost, int32_t *status, int64_t *iters,
                                              int64_t *pivots, int64_t maxiters, int64_t *log, int64_t log_cap, void *stream)
{
    LPX_REQUIRE(batch > 0 && m > 0 && k > 0 && maxiters >= 0, "bad sizes");
    LPX_REQUIRE(G && h && c && basis && x && cost && status && iters && pivots, "null pointer");
    LPX_REQUIRE(log == nullptr || log_cap > 0, "log without capacity");
    LPX_REQUIRE(m <= TB_MAX && k <= TB_MAX, "m <= 64 and k <= 64 (one lane per row and per structural column)");
    return tab_launch<false>(TabBatchArgs{batch, m, k + m, G, h, c, basis, x, cost, status, iters, pivots, maxiters, log, log_cap, k, nullptr, nullptr, nullptr, nullptr}, stream);
}

// DualRevisedSimplexSolver(c, A, b, basis).solve(maxiters) per LP (dual_solvers.py:82-120) in condensed tableau form.
// A and b as preprocessing.py:6-15 leaves them (rows with b < 0 flipped); the starting basis is the slack block, whose
// diagonal is then +-1, with zero costs on it and c >= 0 elsewhere (dual feasible, :14-16) -- the caller's checks.
extern "C" int lpx_batched_tableau_dual_solve(int64_t batch, int64_t m, int64_t n, const float *A, const float *b, const float *c,
                                              int64_t *basis, double *x, double *cost, int32_t *status, int64_t *iters,
                                              int64_t *pivots, int64_t maxiters, int64_t *log, int64_t log_cap, void *stream)
{
    LPX_REQUIRE(batch > 0 && m > 0 && n > m && maxiters >= 0, "bad sizes");
    LPX_REQUIRE(A && b && c && basis && x && cost && status && iters && pivots, "null pointer");
    LPX_REQUIRE(log == nullptr || log_cap > 0, "log without capacity");
    LPX_REQUIRE(m <= TB_MAX && n - m <= TB_MAX, "m <= 64 and n - m <= 64 (one lane per row and per nonbasic column)");
    return tab_launch<true>(TabBatchArgs{batch, m, n, A, b, c, basis, x, cost, status, iters, pivots, maxiters, log, log_cap, n, nullptr, nullptr, nullptr, nullptr}, stream);
}

// lpx_batched_tableau_solve from ANY primal feasible starting basis: the caller passes the starting tableau (see
// TabBatchArgs: T0 m x k, rc0 k, xb0 m, var0 k per LP, all float32-valued as the reference's constructor state) and the
// basis (input and output).  c (n per LP) is only needed for the cost of the result.
extern "C" int lpx_batched_tableau_solve_from(int64_t batch, int64_t m, int64_t n, const float *T0, const float *rc0,
                                              const float *xb0, const int64_t *var0, const float *c, int64_t *basis, double *x,
                                              double *cost, int32_t *status, int64_t *iters, int64_t *pivots, int64_t maxiters,
                                              int64_t *log, int64_t log_cap, void *stream)
{
    LPX_REQUIRE(batch > 0 && m > 0 && n > m && maxiters >= 0, "bad sizes");
    LPX_REQUIRE(T0 && rc0 && xb0 && var0 && c && basis && x && cost && status && iters && pivots, "null pointer");
    LPX_REQUIRE(log == nullptr || log_cap > 0, "log without capacity");
    LPX_REQUIRE(m <= TB_MAX && n - m <= TB_MAX, "m <= 64 and n - m <= 64 (one lane per row and per nonbasic column)");
    // A and b are not read in this mode: T0 / xb0 stand in so that the unused pointers stay valid
    return tab_launch<false>(TabBatchArgs{batch, m, n, T0, xb0, c, basis, x, cost, status, iters, pivots, maxiters, log, log_cap, n, T0, rc0,
                                          xb0, var0},
                             stream);
}
