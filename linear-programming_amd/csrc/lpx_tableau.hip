// lpx_tableau.hip -- tableau-form primal simplex loop (unpartitioned and column-partitioned).
//
// T = [ B^-1 A | x_B ; rc | -z ] is (m+1) x ld fp64 row-major, resident in HBM and updated in place by the
// streaming rank-1 kernel K1 (lpx_rank1.hpp).  Pricing reads the carried reduced-cost row (8 n B), the
// direction is a strided read of one column (8 m B), so an iteration moves 16 (m+1)(n+1) B + O(m+n).
// Rules (first negative reduced cost enters, first minimum ratio leaves, float32 first ratio test) are those
// of primal_solvers.py:135-167; the reference itself forms B^-1 A only at special_solvers.py:69.
//
// Column-partitioned form (one rank per GPU): rank g carries global columns [col0, col0+n) plus a replica of
// the x_B column.  Per pivot each rank proposes (local min rc, first owned negative column, that column of T);
// ONE all-gather of the proposals (RCCL, issued by the Python host on the same stream) makes every rank able
// to pick the global entering column, run the ratio test redundantly on its x_B replica and update its slice.
#include "lpx_device.hpp"
#include "lpx_rank1.hpp"

struct TabWs {
    double *d, *e, *rowr, *scal;
    Partials price, ratio;
    int64_t bytes;
};
static inline char *tcarve(char *&p, int64_t bytes)
{
    char *q = p;
    p += lpx_roundup(bytes, 256);
    return q;
}
static inline void tcarve_partials(char *&p, Partials &P)
{
    P.v0 = (double *)tcarve(p, 8 * LPX_MAXP);
    P.i0 = (int64_t *)tcarve(p, 8 * LPX_MAXP);
    P.v1 = (double *)tcarve(p, 8 * LPX_MAXP);
    P.i1 = (int64_t *)tcarve(p, 8 * LPX_MAXP);
}
static TabWs tab_ws(int64_t m, int64_t n, void *base)
{
    TabWs w;
    char *p = (char *)base;
    w.d = (double *)tcarve(p, 8 * (m + 2));
    w.e = (double *)tcarve(p, 8 * (m + 2));
    w.rowr = (double *)tcarve(p, 8 * (lpx_roundup(n + 1, 2) + 2));
    w.scal = (double *)tcarve(p, 8 * 16);
    tcarve_partials(p, w.price);
    tcarve_partials(p, w.ratio);
    w.bytes = (int64_t)(p - (char *)base);
    return w;
}
extern "C" int64_t lpx_tableau_workspace_bytes(int64_t m, int64_t n)
{
    if (m <= 0 || n <= 0) return 0;
    return tab_ws(m, n, nullptr).bytes;
}
extern "C" int64_t lpx_tableau_msg_len(int64_t m) { return lpx_roundup(m + 3, 2); }

static int check_tableau(const lpx_tableau *s)
{
    LPX_REQUIRE(s != nullptr, "null state");
    LPX_REQUIRE(s->m > 0 && s->n > 0, "empty problem");
    LPX_REQUIRE(s->ld >= s->n + 1 && (s->ld % 2) == 0, "ld must be >= n+1 and even");
    LPX_REQUIRE(s->T && s->basis && s->var_state && s->ctrl && s->workspace, "null device pointer");
    LPX_REQUIRE(lpx_aligned16(s->T) && lpx_aligned16(s->workspace), "T and workspace must be 16-byte aligned");
    LPX_REQUIRE(s->workspace_bytes >= lpx_tableau_workspace_bytes(s->m, s->n), "workspace too small");
    LPX_REQUIRE(lpx_cdiv(s->n, LPX_BLOCK) <= LPX_MAXP && lpx_cdiv(s->m + 1, LPX_BLOCK) <= LPX_MAXP,
                "problem too large for the partial arrays");
    LPX_REQUIRE(s->col0 >= 0 && s->n_global >= s->col0 + s->n, "bad column partition");
    return LPX_OK;
}

// K3 (tableau): scan the carried reduced-cost row: per-workgroup (min rc, first global column with rc<0)
__global__ __launch_bounds__(LPX_BLOCK) void k_t_price(lpx_tableau s, Partials P)
{
    __shared__ BlockRed red;
    if (s.ctrl->status != LPX_RUNNING) return;
    const int64_t j = (int64_t)blockIdx.x * LPX_BLOCK + threadIdx.x;
    const bool in = j < s.n;
    const double rc = in ? s.T[s.m * s.ld + j] : LPX_INF;
    const double mn = block_min(rc, red);
    const int64_t jn = block_imin((in && rc < 0.0) ? s.col0 + j : LPX_NOIDX, red);
    if (threadIdx.x == 0) {
        P.v0[blockIdx.x] = mn;
        P.i0[blockIdx.x] = jn;
    }
}

__device__ __forceinline__ VI t_combine_argmin(const double *v, const int64_t *i, int64_t np, BlockRed &red)
{
    VI b{LPX_INF, LPX_NOIDX};
    for (int64_t k = threadIdx.x; k < np; k += LPX_BLOCK)
        if (vi_less(v[k], i[k], b.v, b.i)) { b.v = v[k]; b.i = i[k]; }
    return block_argmin(b, red);
}
__device__ __forceinline__ double t_combine_min(const double *v, int64_t np, BlockRed &red)
{
    double b = LPX_INF;
    for (int64_t k = threadIdx.x; k < np; k += LPX_BLOCK) b = fmin(b, v[k]);
    return block_min(b, red);
}
__device__ __forceinline__ double t_combine_max(const double *v, int64_t np, BlockRed &red)
{
    double b = -LPX_INF;
    for (int64_t k = threadIdx.x; k < np; k += LPX_BLOCK) b = fmax(b, v[k]);
    return block_max(b, red);
}
__device__ __forceinline__ int64_t t_combine_imin(const int64_t *i, int64_t np, BlockRed &red)
{
    int64_t b = LPX_NOIDX;
    for (int64_t k = threadIdx.x; k < np; k += LPX_BLOCK) b = i[k] < b ? i[k] : b;
    return block_imin(b, red);
}

// ratio-test partial of one workgroup over rows [i] given the direction value di (all threads participate)
__device__ __forceinline__ void t_ratio_partial(const lpx_tableau &s, int64_t i, double di, bool f32, Partials R,
                                                BlockRed &red)
{
    const bool in = i < s.m;
    VI t{in ? lpx_psd(s.T[i * s.ld + s.n], di, f32) : LPX_INF, in ? i : LPX_NOIDX};
    t = block_argmin(t, red);
    const double dmax = block_max(in ? di : -LPX_INF, red);
    if (threadIdx.x == 0) {
        R.v0[blockIdx.x] = t.v;
        R.i0[blockIdx.x] = t.i;
        R.v1[blockIdx.x] = dmax;
    }
}

// K4 (tableau, unpartitioned): entering decision + gather of column j + ratio partials
__global__ __launch_bounds__(LPX_BLOCK) void k_t_column(lpx_tableau s, Partials P, int64_t nprice, Partials R, double *d)
{
    __shared__ BlockRed red;
    lpx_ctrl *ctrl = s.ctrl;
    if (ctrl->status != LPX_RUNNING) return;
    const bool f32 = ctrl->f32_phase != 0;
    const double mn = t_combine_min(P.v0, nprice, red);
    const int64_t j = t_combine_imin(P.i0, nprice, red);
    const bool opt = lpx_is_optimal_min(mn);
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        ctrl->counter += 1;
        ctrl->price_val = mn;
        if (opt) ctrl->status = LPX_OPTIMAL;
        else ctrl->enter_j = j;
    }
    if (opt) return;
    const int64_t i = (int64_t)blockIdx.x * LPX_BLOCK + threadIdx.x;
    double di = 0.0;
    if (i <= s.m) {
        di = s.T[i * s.ld + (j - s.col0)];
        d[i] = di;
    }
    t_ratio_partial(s, i, di, f32, R, red);
}

// column-partitioned, part 1: local proposal
__global__ __launch_bounds__(LPX_BLOCK) void k_t_propose(lpx_tableau s, Partials P, int64_t nprice, double *sendbuf)
{
    __shared__ BlockRed red;
    if (s.ctrl->status != LPX_RUNNING) return;
    const double mn = t_combine_min(P.v0, nprice, red);
    const int64_t j = t_combine_imin(P.i0, nprice, red);
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        sendbuf[0] = mn;
        sendbuf[1] = (double)j; // exact: column indices < 2^53, LPX_NOIDX = 2^62
    }
    const int64_t i = (int64_t)blockIdx.x * LPX_BLOCK + threadIdx.x;
    if (i <= s.m) sendbuf[2 + i] = (j == LPX_NOIDX) ? 0.0 : s.T[i * s.ld + (j - s.col0)];
}

// column-partitioned, part 2a: global entering decision from the gathered proposals + ratio partials
__global__ __launch_bounds__(LPX_BLOCK) void k_t_select(lpx_tableau s, const double *recv, int64_t nranks, int64_t msg,
                                                       Partials R, double *d)
{
    __shared__ BlockRed red;
    lpx_ctrl *ctrl = s.ctrl;
    if (ctrl->status != LPX_RUNNING) return;
    const bool f32 = ctrl->f32_phase != 0;
    double mn = LPX_INF;
    int64_t j = LPX_NOIDX, win = 0;
    for (int64_t g = 0; g < nranks; ++g) { // a handful of ranks: every thread scans them
        const double v = recv[g * msg];
        const int64_t idx = (int64_t)recv[g * msg + 1];
        mn = fmin(mn, v);
        if (idx < j) { j = idx; win = g; }
    }
    const bool opt = lpx_is_optimal_min(mn);
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        ctrl->counter += 1;
        ctrl->price_val = mn;
        if (opt) ctrl->status = LPX_OPTIMAL;
        else ctrl->enter_j = j;
    }
    if (opt) return;
    const int64_t i = (int64_t)blockIdx.x * LPX_BLOCK + threadIdx.x;
    double di = 0.0;
    if (i <= s.m) {
        di = recv[win * msg + 2 + i];
        d[i] = di;
    }
    t_ratio_partial(s, i, di, f32, R, red);
}

// K1 prologue (tableau): leaving row, eta column (incl. the reduced-cost row's multiplier), old pivot row copy
__global__ __launch_bounds__(LPX_BLOCK) void k_t_prepare(lpx_tableau s, Partials R, int64_t nratio, const double *d,
                                                        double *e, double *rowr)
{
    __shared__ BlockRed red;
    lpx_ctrl *ctrl = s.ctrl;
    if (ctrl->status != LPX_RUNNING) return;
    const bool lead = (blockIdx.x == 0 && threadIdx.x == 0);
    const VI t = t_combine_argmin(R.v0, R.i0, nratio, red);
    const double dmax = t_combine_max(R.v1, nratio, red);
    if (dmax <= 0.0) { // primal_solvers.py:144-147
        if (lead) {
            ctrl->aux_val = dmax;
            ctrl->status = LPX_PRIMAL_UNBOUNDED;
        }
        return;
    }
    const int64_t r = t.i;
    const int64_t j = ctrl->enter_j;
    const double dr = d[r];
    const int64_t gi = (int64_t)blockIdx.x * LPX_BLOCK + threadIdx.x;
    if (gi <= s.m) e[gi] = (gi == r) ? 1.0 / dr : (-d[gi]) / dr; // e[m] eliminates rc_j
    if (gi < lpx_roundup_dev(s.n + 1, 2)) rowr[gi] = s.T[r * s.ld + gi]; // K1 reads 2-wide: even length <= ld
    if (lead) {
        const int64_t leaving = s.basis[r];
        s.basis[r] = j;
        const int64_t ll = leaving - s.col0, jl = j - s.col0;
        if (ll >= 0 && ll < s.n) s.var_state[ll] = LPX_VS_LOWER;
        if (jl >= 0 && jl < s.n) s.var_state[jl] = LPX_VS_BASIC;
        if (s.log && ctrl->pivots < s.log_cap) {
            s.log[3 * ctrl->pivots + 0] = r;
            s.log[3 * ctrl->pivots + 1] = j;
            s.log[3 * ctrl->pivots + 2] = leaving;
        }
        ctrl->pivots += 1;
        ctrl->leave_r = r;
        ctrl->leave_var = leaving;
        ctrl->ratio_val = t.v;
        ctrl->aux_val = dmax;
        ctrl->flip = 0;
        ctrl->f32_phase = 0;
    }
}

static inline int64_t t_prep_len(const lpx_tableau *s)
{
    const int64_t a = lpx_roundup(s->n + 1, 2), b = s->m + 1;
    return a > b ? a : b;
}
static void t_launch_update(const lpx_tableau *s, const TabWs &w, hipStream_t st)
{
    Rank1Args a{};
    a.M = s->T; a.rows = s->m + 1; a.cols = s->n + 1; a.ld = s->ld; a.e = w.e; a.rowr = w.rowr; a.r = -1;
    a.ctrl = s->ctrl; a.xB = nullptr; a.xr_old = nullptr; a.col0 = s->col0;
    const Rank1Plan p = lpx_rank1_plan(a.rows, a.cols);
    lpx_k1_begin(st);
    lpx_rank1_launch<true>(p, a, st);
    lpx_k1_end(st);
}

extern "C" int lpx_tableau_iterate(const lpx_tableau *s, int64_t iters, void *stream)
{
    int rc = check_tableau(s);
    if (rc) return rc;
    LPX_REQUIRE(s->col0 == 0 && s->n_global == s->n, "lpx_tableau_iterate needs an unpartitioned tableau");
    const TabWs w = tab_ws(s->m, s->n, s->workspace);
    hipStream_t st = (hipStream_t)stream;
    const int64_t nprice = lpx_cdiv(s->n, LPX_BLOCK), nrow = lpx_cdiv(s->m + 1, LPX_BLOCK);
    for (int64_t it = 0; it < iters; ++it) {
        hipLaunchKernelGGL(k_t_price, dim3((unsigned)nprice), dim3(LPX_BLOCK), 0, st, *s, w.price);
        hipLaunchKernelGGL(k_t_column, dim3((unsigned)nrow), dim3(LPX_BLOCK), 0, st, *s, w.price, nprice, w.ratio, w.d);
        hipLaunchKernelGGL(k_t_prepare, dim3((unsigned)lpx_cdiv(t_prep_len(s), LPX_BLOCK)), dim3(LPX_BLOCK), 0, st, *s, w.ratio,
                           nrow, (const double *)w.d, w.e, w.rowr);
        t_launch_update(s, w, st);
    }
    LPX_LAUNCH_CHECK();
    return LPX_OK;
}

extern "C" int lpx_tableau_propose(const lpx_tableau *s, double *sendbuf, void *stream)
{
    int rc = check_tableau(s);
    if (rc) return rc;
    LPX_REQUIRE(sendbuf != nullptr, "null send buffer");
    const TabWs w = tab_ws(s->m, s->n, s->workspace);
    hipStream_t st = (hipStream_t)stream;
    const int64_t nprice = lpx_cdiv(s->n, LPX_BLOCK), nrow = lpx_cdiv(s->m + 1, LPX_BLOCK);
    hipLaunchKernelGGL(k_t_price, dim3((unsigned)nprice), dim3(LPX_BLOCK), 0, st, *s, w.price);
    hipLaunchKernelGGL(k_t_propose, dim3((unsigned)nrow), dim3(LPX_BLOCK), 0, st, *s, w.price, nprice, sendbuf);
    LPX_LAUNCH_CHECK();
    return LPX_OK;
}

extern "C" int lpx_tableau_commit(const lpx_tableau *s, const double *recvbuf, int64_t nranks, void *stream)
{
    int rc = check_tableau(s);
    if (rc) return rc;
    LPX_REQUIRE(recvbuf != nullptr && nranks >= 1 && nranks <= 64, "bad receive buffer / rank count");
    const TabWs w = tab_ws(s->m, s->n, s->workspace);
    hipStream_t st = (hipStream_t)stream;
    const int64_t nrow = lpx_cdiv(s->m + 1, LPX_BLOCK);
    hipLaunchKernelGGL(k_t_select, dim3((unsigned)nrow), dim3(LPX_BLOCK), 0, st, *s, recvbuf, nranks,
                       lpx_tableau_msg_len(s->m), w.ratio, w.d);
    hipLaunchKernelGGL(k_t_prepare, dim3((unsigned)lpx_cdiv(t_prep_len(s), LPX_BLOCK)), dim3(LPX_BLOCK), 0, st, *s, w.ratio,
                       nrow, (const double *)w.d, w.e, w.rowr);
    t_launch_update(s, w, st);
    LPX_LAUNCH_CHECK();
    return LPX_OK;
}

// primal_solvers.py:72-79 on the tableau's replicated x_B column
__global__ __launch_bounds__(LPX_BLOCK) void k_t_fill(double *x, int64_t n)
{
    const int64_t j = (int64_t)blockIdx.x * LPX_BLOCK + threadIdx.x;
    if (j < n) x[j] = 0.0;
}
__global__ __launch_bounds__(LPX_BLOCK) void k_t_scatter(lpx_tableau s, double *x)
{
    const int64_t i = (int64_t)blockIdx.x * LPX_BLOCK + threadIdx.x;
    if (i < s.m) x[s.basis[i]] = s.T[i * s.ld + s.n];
}
__global__ __launch_bounds__(1024) void k_t_cost(const float *c, const double *x, int64_t n, double *cost)
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
extern "C" int lpx_tableau_expand(const lpx_tableau *s, const float *c_global, double *x, double *cost, void *stream)
{
    int rc = check_tableau(s);
    if (rc) return rc;
    LPX_REQUIRE(c_global && x && cost, "null pointer");
    hipStream_t st = (hipStream_t)stream;
    hipLaunchKernelGGL(k_t_fill, dim3((unsigned)lpx_cdiv(s->n_global, LPX_BLOCK)), dim3(LPX_BLOCK), 0, st, x, s->n_global);
    hipLaunchKernelGGL(k_t_scatter, dim3((unsigned)lpx_cdiv(s->m, LPX_BLOCK)), dim3(LPX_BLOCK), 0, st, *s, x);
    hipLaunchKernelGGL(k_t_cost, dim3(1), dim3(1024), 0, st, c_global, (const double *)x, s->n_global, cost);
    LPX_LAUNCH_CHECK();
    return LPX_OK;
}
