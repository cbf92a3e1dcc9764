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
// diagnostic build only (tools/tab_cycles.hip): resident cycles and pivots summed over the workgroups of a launch
#ifdef LPX_TAB_STAMPS
__device__ unsigned long long g_tab_stamps[2];
#define TB_STAMP_BEGIN const unsigned long long tb_t0 = clock64();
#define TB_STAMP_END                                                      \
    if (threadIdx.x == 0) {                                               \
        atomicAdd(&g_tab_stamps[0], clock64() - tb_t0);                   \
        atomicAdd(&g_tab_stamps[1], (unsigned long long)pivots);          \
    }
#else
#define TB_STAMP_BEGIN
#define TB_STAMP_END
#endif
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
__global__ __launch_bounds__(64, MR <= 32 ? 4 : 3) void k_batched_tab(TabBatchArgs a)
{
    __shared__ double dcol[TB_MAX + 2];
    __shared__ double xs[2 * TB_MAX];
    __shared__ float cf[2 * TB_MAX];
    TB_STAMP_BEGIN
    const int t = threadIdx.x;
    const int64_t lp = blockIdx.x;
    const int m = (int)a.m, n = (int)a.n, k = n - m;
    const bool col = t < k, row = t < m;
    const int na = (int)a.na;
    const float *A = a.A + lp * a.m * a.na;

    const bool from_T0 = !DUAL && a.T0 != nullptr; // wave-uniform: the caller supplies the starting tableau (A, b unused)
    // Loads are unconditional, from clamped (always valid) addresses, and selected afterwards: a load under a lane
    // condition compiles to a branch with its own wait, i.e. MR global-memory round trips one after the other
    const int tc = col ? t : 0, tr = row ? t : 0;
    double x[MR];
    double rc, xb;
    int64_t var = col ? (int64_t)t : LPX_NOIDX; // variable held by this lane's column slot
    int64_t bas = row ? (int64_t)(k + t) : -1;  // basis[t]
    if (!from_T0) {
        float av[MR], sv[MR];
#pragma unroll
        for (int i = 0; i < MR; ++i) {
            const int ri = i < m ? i : m - 1;
            av[i] = A[(int64_t)ri * na + tc];
            // B^-1 of a +-identity basis is itself: row i of B^-1 A is row i of A times B_ii (exact)
            sv[i] = DUAL ? A[(int64_t)ri * na + k + ri] : 1.f;
        }
        const float cv = a.c[lp * na + tc], bv = a.b[lp * m + tr];
        const float sb = DUAL ? A[(int64_t)tr * na + k + tr] : 1.f;
#pragma unroll
        for (int i = 0; i < MR; ++i) x[i] = (col && i < m) ? (double)sv[i] * (double)av[i] : 0.0;
        rc = col ? (double)cv : 0.0; // basic costs are 0: rc = c - c_B B^-1 A = c
        xb = row ? (double)bv : 0.0;
        if (DUAL && row) xb *= (double)sb;
    } else {
        const float *T0 = a.T0 + lp * a.m * k;
        float av[MR];
#pragma unroll
        for (int i = 0; i < MR; ++i) av[i] = T0[(int64_t)(i < m ? i : m - 1) * k + tc];
        const float cv = a.rc0[lp * k + tc], bv = a.xb0[lp * m + tr];
        const int64_t v0 = a.var0[lp * k + tc], b0 = a.basis[lp * m + tr];
#pragma unroll
        for (int i = 0; i < MR; ++i) x[i] = (col && i < m) ? (double)av[i] : 0.0;
        rc = col ? (double)cv : 0.0;
        xb = row ? (double)bv : 0.0;
        var = col ? v0 : LPX_NOIDX;
        bas = row ? b0 : -1;
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
            // first negative index, by variable id (:135-138); ids are below 2^31 (n <= 256 here): a 32-bit reduction
            jv = (int64_t)wave_imin32(neg ? (int)var : 0x7FFFFFFF);
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
            const double th = row ? lpx_psd(xb, d, f32) : LPX_INF;
            const double thmin = wave_min(th);
            // rows ascend with the lane: the first lane at the minimum is np.argmin's first argmin (:165-166)
            r = __builtin_amdgcn_readfirstlane(__ffsll((long long)__ballot(th == thmin)) - 1);
        }
        dr = tb_readlane(d, r);
        const double e = ((t == r) ? 1.0 : -d) / dr; // :236-244: 1/d_r or (-d_i)/d_r (lanes >= m: d = 0, e = -0, rows stay 0)
        const double em = (-rcj) / dr;               // eliminates the entering column's reduced cost
        const double xr = tb_readlane(xb, r);
        if (row) xb = fma(e, xr, (t == r) ? -0.0 : xb); // :251-253 (fma on -0.0 = the plain product)
        const int64_t leaving = (int64_t)__builtin_amdgcn_readlane((int)bas, r); // variable ids < 2^31 here (n <= 128)
        if (!DUAL) { // this column's pivot-row entry
            switch (r) {
#define TB_GET(i) case (i): if constexpr ((i) < MR) pr = x[(i) < MR ? (i) : 0]; break;
                TB_ROWS(TB_GET)
#undef TB_GET
            }
        }
        // Two special cases take their addend from -0.0: the pivot row in every lane -- fma(e_r, row_r, -0.0) is the plain
        // product -- and every row of the entering lane, whose multiplicand is 1: fma(e_i, 1, -0.0) is e_i itself, bit for
        // bit.  Neither costs a select per element: the entering lane clears its column under its own exec mask, and only
        // the 8 rows around the pivot row (a scalar branch) run selects.  The update is then 2 readlanes + 1 fma per row
        // (e as broadcast ds_read_b128, one per two rows, is slower: 718 against 790 M pivots/s at 64 x 128 -- every LDS
        // instruction costs a lone wave ~13 cycles of issue and the four SIMDs share one LDS pipe).
        const bool mine = t == jl;
        const double prs = mine ? 1.0 : pr;
        if (mine) {
#pragma unroll
            for (int i = 0; i < MR; ++i) x[i] = -0.0;
        }
        {
            // e_i lives in lane i; E[g] holds row g of the wave's four 16-lane rows in every row, so that row i's
            // multiplier reaches all 64 lanes as a DPP row broadcast inside the fma itself: ONE VALU instruction per element
            // (two v_readlane_b32 + v_fmac_f64 before: 3.1 -> see DESIGN.md section 5)
            double E[4];
            lpx_rows_bcast(e, E);
            const int rchunk = r >> 3;
#pragma unroll
            for (int c = 0; c < MR / 8; ++c) {
                if (rchunk == c) { // (clearing the row inside the switch that reads it costs 3x: 64 register merges per case)
#pragma unroll
                    for (int j = 0; j < 8; ++j) x[8 * c + j] = (8 * c + j == r) ? -0.0 : x[8 * c + j];
                }
#pragma unroll
                for (int j = 0; j < 8; ++j) LPX_FMAC_ROW_BCAST(x[8 * c + j], E[(8 * c + j) / 16], prs, (8 * c + j) % 16);
            }
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
    TB_STAMP_END
}

extern "C" int lpx_batched_tableau_waves(int64_t m, int64_t k);
template <bool DUAL>
static int tab_launch_multi(const TabBatchArgs &a, void *stream);
template <bool DUAL>
static int tab_launch(const TabBatchArgs &a, void *stream)
{
    if (a.m > TB_MAX || a.n - a.m > TB_MAX) return tab_launch_multi<DUAL>(a, stream);
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
// Shapes: lpx_batched_tableau_waves(m, n - m) != 0; starting basis = the slack block (columns n-m .. n-1 are the identity, b >= 0) with zero costs on
// it -- the caller's check.  Arrays batch-major as in lpx_batched_primal_solve; basis is written (m per LP).
extern "C" int lpx_batched_tableau_solve(int64_t batch, int64_t m, int64_t n, const float *A, const float *b, const float *c,
                                         int64_t *basis, double *x, double *cost, int32_t *status, int64_t *iters,
                                         int64_t *pivots, int64_t maxiters, int64_t *log, int64_t log_cap, void *stream)
{
    LPX_REQUIRE(batch > 0 && m > 0 && n > m && maxiters >= 0, "bad sizes");
    LPX_REQUIRE(A && b && c && basis && x && cost && status && iters && pivots, "null pointer");
    LPX_REQUIRE(log == nullptr || log_cap > 0, "log without capacity");
    if (!lpx_batched_tableau_waves(m, n - m)) {
        lpx_set_error("lpx_batched_tableau_solve: m <= 128 with n - m <= 256, or m <= 256 with n - m <= 128 (up to eight waves per LP)");
        return LPX_E_SIZE;
    }
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
    if (!lpx_batched_tableau_waves(m, k)) {
        lpx_set_error("lpx_batched_tableau_solve_ineq: m <= 128 with k <= 256, or m <= 256 with k <= 128 (up to eight waves per LP)");
        return LPX_E_SIZE;
    }
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
    LPX_REQUIRE(lpx_batched_tableau_waves(m, n - m) != 0, "m <= 128 with n - m <= 256, or m <= 256 with n - m <= 128 (up to eight waves per LP)");
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
    LPX_REQUIRE(lpx_batched_tableau_waves(m, n - m) != 0, "m <= 128 with n - m <= 256, or m <= 256 with n - m <= 128 (up to eight waves per LP)");
    // A and b are not read in this mode: T0 / xb0 stand in so that the unused pointers stay valid
    return tab_launch<false>(TabBatchArgs{batch, m, n, T0, xb0, c, basis, x, cost, status, iters, pivots, maxiters, log, log_cap, n, T0, rc0,
                                          xb0, var0},
                             stream);
}

// ---------------------------------------------------------------------------------------------------------
// The same kernel for LPs beyond one wave: a GRID of RW x CW waves per LP (one workgroup).  Wave (rg, cg) carries rows
// rg*MRL .. rg*MRL+MRL-1 of column slots cg*64 .. cg*64+63: at most 64 rows per lane whatever m is, so the register count
// stays that of the one-wave kernel (three waves per SIMD) where a single column of up to 128 rows per lane would leave
// one wave per SIMD and nothing to hide a barrier behind.  Thread t of the workgroup also owns row t (x_B[t], basis[t]).
// What the one-wave kernel moves between lanes with v_readlane and ballots goes through a few words of LDS here, four
// exchanges per pivot, one barrier each:
//   (A) pricing: per column group "any rc < 0", "any rc < -1e-8" and the smallest negative variable id packed with its
//       slot (ids are distinct, so the minimum of id * 256 + slot names the entering slot as well); every row group
//       keeps its own copy of the reduced costs (same fma, same bits), row group 0 reports;
//   (B) the entering column: its owner in every row group writes its MRL rows;
//   (C) the ratio test's per-wave candidates (theta, row, x_B, basic variable): rows ascend with the thread, first
//       minimum wins;
//   (D) the eta column e and the pivot row (written by the row group that holds row r).
// The update itself is one fma per element with no selects: the pivot row's addend is set to -0.0 by the scalar switch
// that fetches it, the entering slot's lane sets its column to -0.0 under its own exec mask (one wave in CW pays for it).
// Same arithmetic, same order: bit-identical to lpo_tableau_solve and to the one-wave kernel on shapes both take.
// Shapes: RW * CW <= 8 waves, i.e. m <= 128 with n - m <= 256, or m <= 256 with n - m <= 128.  Primal rules, slack start
// (or inequality form).
// ---------------------------------------------------------------------------------------------------------
#define TBG_MAX_N 512
struct Tab2Price {
    int64_t key; // smallest negative variable id * 256 + its column slot (INT64_MAX: none)
    int anyneg, anyopen;
};
struct Tab2Ratio {
    double th, xb;
    int row, bas, anypos, pad;
};

struct Tab2Row { // dual rules: a wave's first row with x_B < 0
    double xb;
    int row, bas, anyneg, anyopen;
};
struct Tab2Theta { // dual rules: a column group's first argmin of theta, by variable id
    double th;
    int var, slot, anyrow, pad;
};

// DUAL: the dual simplex rules (dual_solvers.py:47-108) on the same carried state, as in k_batched_tab.
template <int MRL, int RW, int CW, bool DUAL>
__global__ __launch_bounds__(64 * RW * CW, MRL <= 32 ? 4 : 3) void k_batched_tab_grid(TabBatchArgs a)
{
    constexpr int NW = RW * CW, NT = 64 * NW, M = RW * MRL, NRW = (M + 63) / 64;
    static_assert(MRL % 8 == 0 && MRL <= 64 && NW <= 16 && 64 * CW <= 256, "shape");
    __shared__ __attribute__((aligned(16))) double dcol[M + 2];
    __shared__ __attribute__((aligned(16))) double es[M + 64];
    __shared__ double prow[64 * CW];
    __shared__ double xs[TBG_MAX_N];
    __shared__ float cf[TBG_MAX_N];
    __shared__ Tab2Price shp[CW];
    __shared__ Tab2Ratio shr[NRW];
    __shared__ Tab2Row shd[NRW];
    __shared__ Tab2Theta shq[CW];
    TB_STAMP_BEGIN
    const int t = threadIdx.x, lane = t & 63;
    const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
    const int rg = wave / CW, cg = wave - rg * CW;
    const int cs = cg * 64 + lane; // this lane's column slot
    const int r0 = rg * MRL;       // first row of this wave's row group
    const int64_t lp = blockIdx.x;
    const int m = (int)a.m, n = (int)a.n, k = n - m;
    const bool col = cs < k, row = t < m;
    const int na = (int)a.na;
    const float *A = a.A + lp * a.m * a.na;
    const bool from_T0 = !DUAL && a.T0 != nullptr; // uniform: the caller supplies the starting tableau (A, b unused)

    double x[MRL];
    double rc, xb;
    int var = col ? cs : 0x7FFFFFFF; // variable held by this lane's column slot
    int bas = row ? k + t : -1;      // basis[t]
    { // unconditional loads from clamped addresses, selected afterwards (see k_batched_tab)
        const int tc = col ? cs : 0, tr = row ? t : 0;
        float av[MRL], sv[MRL];
        if (!from_T0) {
#pragma unroll
            for (int i = 0; i < MRL; ++i) {
                const int ri = r0 + i < m ? r0 + i : m - 1;
                av[i] = A[(int64_t)ri * na + tc];
                // B^-1 of a +-identity basis is itself: row i of B^-1 A is row i of A times B_ii (exact)
                sv[i] = DUAL ? A[(int64_t)ri * na + k + ri] : 1.f;
            }
            const float cv = a.c[lp * na + tc], bv = a.b[lp * m + tr];
            const float sb = DUAL ? A[(int64_t)tr * na + k + tr] : 1.f;
            rc = col ? (double)cv : 0.0; // basic costs are 0: rc = c - c_B B^-1 A = c
            xb = row ? (double)bv : 0.0;
            if (DUAL && row) xb *= (double)sb;
        } else {
            const float *T0 = a.T0 + lp * a.m * k;
#pragma unroll
            for (int i = 0; i < MRL; ++i) {
                av[i] = T0[(int64_t)(r0 + i < m ? r0 + i : m - 1) * k + tc];
                sv[i] = 1.f;
            }
            const float cv = a.rc0[lp * k + tc], bv = a.xb0[lp * m + tr];
            const int64_t v0 = a.var0[lp * k + tc], b0 = a.basis[lp * m + tr];
            rc = col ? (double)cv : 0.0;
            xb = row ? (double)bv : 0.0;
            var = col ? (int)v0 : 0x7FFFFFFF;
            bas = row ? (int)b0 : -1;
        }
#pragma unroll
        for (int i = 0; i < MRL; ++i) x[i] = (col && r0 + i < m) ? (double)sv[i] * (double)av[i] : 0.0;
    }
    for (int j = t; j < n; j += NT) cf[j] = j < na ? a.c[lp * na + j] : 0.f;
    if (t < 64) es[M + t] = 0.0; // the tail a whole-wave read of a short row group touches

    bool f32 = true; // the first ratio test divides in float32
    int32_t status = LPX_RUNNING;
    int64_t counter = 0, pivots = 0;
    while (counter < a.maxiters) {
        counter += 1;
        int r, jv, jl, leaving;
        double d, rcj, dr, xr, pr = 0.0;
        bool mine;
        if (!DUAL) {
            // ---- (A) pricing (primal_solvers.py:135-142)
            if (rg == 0) {
                const bool neg = col && rc < 0.0;
                const int64_t mine_key = neg ? (((int64_t)var << 8) | cs) : INT64_MAX; // ids < 2^31, slots < 256
                const int64_t kmin = wave_imin(mine_key);
                const int anyn = __ballot(neg) != 0, anyo = __ballot(col && rc < -LPX_ATOL) != 0;
                if (lane == 0) shp[cg] = Tab2Price{kmin, anyn, anyo};
            }
            __syncthreads();
            int64_t key = INT64_MAX;
            bool anyneg = false, anyopen = false;
#pragma unroll
            for (int w = 0; w < CW; ++w) {
                const Tab2Price p = shp[w];
                key = p.key < key ? p.key : key;
                anyneg = anyneg || p.anyneg;
                anyopen = anyopen || p.anyopen;
            }
            if (!anyneg || !anyopen) { // optimal <=> min rc >= 0 or |min rc| <= 1e-8 (:140-142)
                status = LPX_OPTIMAL;
                break;
            }
            jv = (int)(key >> 8), jl = (int)(key & 0xFF); // first negative index by variable id, and its slot
            mine = cs == jl;                               // one lane in every row group
            // ---- (B) entering column (:149-152)
            if (mine) {
#pragma unroll
                for (int i = 0; i < MRL; ++i) dcol[r0 + i] = x[i];
                if (rg == 0) dcol[M] = rc;
            }
            __syncthreads();
            d = row ? dcol[t] : 0.0;
            rcj = dcol[M];
            // ---- (C) ratio test (:154-167)
            if (wave < NRW) {
                const double th = row ? lpx_psd(xb, d, f32) : LPX_INF;
                const double thmin = wave_min(th);
                const unsigned long long who = __ballot(th == thmin); // rows ascend with the lane: first lane = np.argmin
                const int src = __builtin_amdgcn_readfirstlane(__ffsll((long long)who) - 1);
                const double xbs = tb_readlane(xb, src);
                const int bs = __builtin_amdgcn_readlane(bas, src);
                const int anypos = __ballot(row && d > 0.0) != 0;
                if (lane == 0) shr[wave] = Tab2Ratio{thmin, xbs, wave * 64 + src, bs, anypos, 0};
            }
            __syncthreads();
            Tab2Ratio best = shr[0];
#pragma unroll
            for (int w = 1; w < NRW; ++w) {
                const Tab2Ratio q = shr[w];
                const bool take = q.th < best.th; // strict: the earlier wave (lower rows) keeps ties
                const int ap = best.anypos | q.anypos;
                if (take) best = q;
                best.anypos = ap;
            }
            if (!best.anypos) { // max(d) <= 0 (:144-147)
                status = LPX_PRIMAL_UNBOUNDED;
                break;
            }
            r = __builtin_amdgcn_readfirstlane(best.row); // the same in every lane: make it a scalar
            dr = dcol[r], xr = best.xb;
            leaving = best.bas;
        } else {
            // ---- (A) leaving row = first x_B < 0; optimal iff min x_B >= 0 or |min| <= 1e-8 (dual_solvers.py:47-54)
            if (wave < NRW) {
                const unsigned long long nb = __ballot(row && xb < 0.0);
                const int anyo = __ballot(row && xb < -LPX_ATOL) != 0;
                const int src = nb ? __builtin_amdgcn_readfirstlane(__ffsll((long long)nb) - 1) : 0;
                const double xbs = tb_readlane(xb, src);
                const int bs = __builtin_amdgcn_readlane(bas, src);
                if (lane == 0) shd[wave] = Tab2Row{xbs, wave * 64 + src, bs, nb != 0, anyo};
            }
            __syncthreads();
            Tab2Row lead = shd[0];
            bool anyopen = lead.anyopen != 0;
#pragma unroll
            for (int w = 1; w < NRW; ++w) {
                const Tab2Row q = shd[w];
                anyopen = anyopen || q.anyopen;
                if (!lead.anyneg && q.anyneg) lead = q; // rows ascend with the wave: the first wave that has one
            }
            if (!lead.anyneg || !anyopen) {
                status = LPX_OPTIMAL;
                break;
            }
            r = __builtin_amdgcn_readfirstlane(lead.row);
            xr = lead.xb;
            leaving = lead.bas;
            // ---- (B) the dual row B^-1[r,:] A (:37-38): each column's entry sits in the row group that holds row r
            {
                const int rr = r / MRL, rl = r - rr * MRL;
                if (rg == rr) {
                    switch (rl) {
#define TB_GET(i) case (i): if constexpr ((i) < MRL) pr = x[(i) < MRL ? (i) : 0]; break;
                        TB_ROWS(TB_GET)
#undef TB_GET
                    }
                    prow[cs] = pr;
                }
                __syncthreads();
                if (rg != rr) pr = prow[cs];
            }
            // ---- (C) ratio test over the columns: first argmin of (-rc_j) / row_j over row_j < 0, by variable id (:61-80)
            if (rg == 0) {
                const double th = col ? lpx_dsd(rc, pr, f32) : LPX_INF;
                const double thmin = wave_min(th);
                const int vmin = (int)wave_imin((th == thmin && col) ? (int64_t)var : (int64_t)0x7FFFFFFF);
                const unsigned long long who = __ballot(col && var == vmin);
                const int src = who ? __builtin_amdgcn_readfirstlane(__ffsll((long long)who) - 1) : 0;
                const int anyrow = __ballot(col && pr < 0.0) != 0;
                if (lane == 0) shq[cg] = Tab2Theta{thmin, vmin, cg * 64 + src, anyrow, 0};
            }
            __syncthreads();
            Tab2Theta best = shq[0];
#pragma unroll
            for (int w = 1; w < CW; ++w) {
                const Tab2Theta q = shq[w];
                const int ar = best.anyrow | q.anyrow;
                if (q.th < best.th || (q.th == best.th && q.var < best.var)) best = q;
                best.anyrow = ar;
            }
            if (!best.anyrow) { // min(row) >= 0 (:56-59)
                status = LPX_DUAL_UNBOUNDED;
                break;
            }
            jv = best.var;
            jl = __builtin_amdgcn_readfirstlane(best.slot);
            mine = cs == jl;
            // ---- (D) entering column
            if (mine) {
#pragma unroll
                for (int i = 0; i < MRL; ++i) dcol[r0 + i] = x[i];
                if (rg == 0) dcol[M] = rc;
            }
            __syncthreads();
            d = row ? dcol[t] : 0.0;
            rcj = dcol[M];
            dr = dcol[r];
        }
        // ---- (D) eta column (:236-244), x_B (:251-253), pivot row
        const double em = (-rcj) / dr; // eliminates the entering column's reduced cost
        if (wave < NRW) {
            const double e = ((t == r) ? 1.0 : -d) / dr; // rows >= m: d = 0, e = -0: they stay 0
            if (t < M) es[t] = e;
            if (row) xb = (t == r) ? e * xr : fma(e, xr, xb);
        }
        const int rr = r / MRL, rl = r - rr * MRL;
        if (!DUAL && rg == rr) { // this column's pivot-row entry (the dual rules fetched it for their ratio test)
            switch (rl) {
#define TB_GET(i) case (i): if constexpr ((i) < MRL) pr = x[(i) < MRL ? (i) : 0]; break;
                TB_ROWS(TB_GET)
#undef TB_GET
            }
            prow[cs] = pr;
        }
        __syncthreads();
        if (!DUAL && rg != rr) pr = prow[cs];
        // the entering slot receives the eta column itself: multiplicand 1, addends -0.0 (fma(e_i, 1, -0.0) = e_i bit for bit)
        const double prs = mine ? 1.0 : pr;
        if (mine) {
#pragma unroll
            for (int i = 0; i < MRL; ++i) x[i] = -0.0;
        }
        // the pivot row's addend is -0.0 as well (fma(e_r, row_r, -0.0) is the plain product): a select on a scalar condition
        const int rlm = (rg == rr) ? rl : -1;
        {
            const double ew = es[r0 + lane]; // lane i of every wave of the row group holds e of its row i
            double E[4]; // ... and E[g] row g of those 64 in all four 16-lane rows: the fma broadcasts from there (k_batched_tab)
            lpx_rows_bcast(ew, E);
            const int rchunk = rlm >> 3; // only the 8 rows around the pivot row pay for the selects: a scalar branch
#pragma unroll
            for (int c = 0; c < MRL / 8; ++c) {
                if (rchunk == c) {
#pragma unroll
                    for (int j = 0; j < 8; ++j) x[8 * c + j] = (8 * c + j == rlm) ? -0.0 : x[8 * c + j];
                }
#pragma unroll
                for (int j = 0; j < 8; ++j) LPX_FMAC_ROW_BCAST(x[8 * c + j], E[(8 * c + j) / 16], prs, (8 * c + j) % 16);
            }
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
    for (int j = t; j < n; j += NT) xs[j] = 0.0;
    __syncthreads();
    if (row) xs[bas] = xb;
    __syncthreads();
    for (int j = t; j < n; j += NT) a.x[lp * n + j] = xs[j];
    if (row) a.basis[lp * m + t] = bas;
    if (t == 0) {
        double s1 = 0.0;
        for (int jx = 0; jx < n; ++jx) s1 = fma((double)cf[jx], xs[jx], s1);
        a.cost[lp] = s1;
        a.status[lp] = status;
        a.iters[lp] = counter;
        a.pivots[lp] = pivots;
    }
    TB_STAMP_END
}

// waves per LP for a shape, 0 if it does not fit a workgroup of 8 waves
extern "C" int lpx_batched_tableau_waves(int64_t m, int64_t k)
{
    if (m <= 0 || k <= 0 || m > 256 || k > 256) return 0;
    const int rw = m <= 64 ? 1 : m <= 128 ? 2 : 4, cw = (int)((k + 63) / 64);
    return rw * cw <= 8 ? rw * cw : 0;
}

template <bool DUAL>
static int tab_launch_multi(const TabBatchArgs &a, void *stream)
{
    const int m = (int)a.m, k = (int)(a.n - a.m);
    const int rw = m <= 64 ? 1 : m <= 128 ? 2 : 4, cw = (k + 63) / 64;
    const int per = (m + rw - 1) / rw;
    // rows per lane: in steps of 8 for the primal rules (every shape of a batch sweep gets a close fit), of 16 for the dual
    const int mrl = per <= 16 ? 16 : DUAL ? (per + 15) / 16 * 16 : (per + 7) / 8 * 8;
    const dim3 grid((unsigned)a.batch), block((unsigned)(64 * rw * cw));
    hipStream_t st = (hipStream_t)stream;
    bool done = false;
#define TBG_GO(MRLv, RWv, CWv)                                                                                          \
    if (!done && mrl == MRLv && rw == RWv && cw == CWv) {                                                              \
        hipLaunchKernelGGL((k_batched_tab_grid<MRLv, RWv, CWv, DUAL>), grid, block, 0, st, a);                         \
        done = true;                                                                                                   \
    }
#define TBG_ODD(MRLv, RWv, CWv) if constexpr (!DUAL) { TBG_GO(MRLv, RWv, CWv) }
#define TBG_ROWSN(RWv, CWv) TBG_ODD(40, RWv, CWv) TBG_GO(48, RWv, CWv) TBG_ODD(56, RWv, CWv) TBG_GO(64, RWv, CWv)
#define TBG_ROWS1(CWv) TBG_GO(16, 1, CWv) TBG_ODD(24, 1, CWv) TBG_GO(32, 1, CWv) TBG_ROWSN(1, CWv)
    TBG_ROWS1(2) TBG_ROWS1(3) TBG_ROWS1(4)
    TBG_ROWSN(2, 1) TBG_ROWSN(2, 2) TBG_ROWSN(2, 3) TBG_ROWSN(2, 4)
    TBG_ROWSN(4, 1) TBG_ROWSN(4, 2)
#undef TBG_ROWSN
#undef TBG_ROWS1
#undef TBG_ODD
#undef TBG_GO
    if (!done) {
        lpx_set_error("lpx_batched_tableau_solve: no kernel for m = %d, n - m = %d", m, k);
        return LPX_E_SIZE;
    }
    LPX_LAUNCH_CHECK();
    return LPX_OK;
}
