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
#include "lpx_graph.hpp"

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
    LPX_REQUIRE((s->slot_var == nullptr) == (s->var_slot == nullptr), "slot_var and var_slot come together");
    LPX_REQUIRE(s->slot_var != nullptr || (s->col0 >= 0 && s->n_global >= s->col0 + s->n), "bad column partition");
    LPX_REQUIRE(s->slot_var == nullptr || s->n_global >= s->n, "more slots than variables");
    LPX_REQUIRE(s->pricing == LPX_PRICE_FIRST_NEGATIVE || s->pricing == LPX_PRICE_DANTZIG, "bad pricing rule");
    return LPX_OK;
}

// Condensed form (slot_var != NULL): T holds only the NONBASIC columns -- a basic column is a unit vector that the
// rank-1 update leaves untouched (its pivot-row entry is 0), so carrying it costs memory and 16 bytes of HBM traffic per
// element and pivot for nothing.  Column slot k holds variable slot_var[k]; var_slot[] is the inverse map (-1: not in a
// slot of this rank).  On a pivot the entering variable's slot takes over the leaving variable, whose column is the eta
// column e itself (unit vector e_r updated: 0 + e_i * 1), written by K1 where the full form writes the new unit vector.
__device__ __forceinline__ int64_t t_var(const lpx_tableau &s, int64_t slot) { return s.slot_var ? s.slot_var[slot] : s.col0 + slot; }
// local column of variable j, -1 if this rank does not hold it
__device__ __forceinline__ int64_t t_slot(const lpx_tableau &s, int64_t j)
{
    if (s.var_slot) return s.var_slot[j];
    const int64_t k = j - s.col0;
    return (k < 0 || k >= s.n) ? -1 : k;
}
// bookkeeping of a basis change (single thread): the full form flips var_state, the condensed form hands the slot over
// It leaves ctrl->enter_slot = the local column that the update rewrites as a whole (full form: the entering variable's,
// which becomes a unit vector; condensed: the slot handed to the leaving variable, which receives the eta column), -1 if
// that column is not on this rank.  K1 reads it with the rest of the control block (no lookup through var_slot in
// front of its 10^5..10^6 short workgroups).
__device__ __forceinline__ void t_swap_columns(const lpx_tableau &s, int64_t j, int64_t leaving)
{
    if (s.slot_var) {
        const int64_t k = s.var_slot[j];
        if (k >= 0) {
            s.slot_var[k] = leaving;
            s.var_slot[leaving] = k;
        }
        s.var_slot[j] = -1;
        s.ctrl->enter_slot = k;
        return;
    }
    const int64_t ll = leaving - s.col0, jl = j - s.col0;
    if (ll >= 0 && ll < s.n) s.var_state[ll] = LPX_VS_LOWER;
    if (jl >= 0 && jl < s.n) s.var_state[jl] = LPX_VS_BASIC;
    s.ctrl->enter_slot = (jl >= 0 && jl < s.n) ? jl : -1;
}
__device__ __forceinline__ int64_t t_unit_col(const lpx_tableau &s) { return s.ctrl->enter_slot; }

// K3 (tableau): scan the carried reduced-cost row: one partial (min rc, first global column with rc<0) per
// column strip of K1's launch plan, i.e. in the same layout K1's fused pricing epilogue writes afterwards.
__global__ __launch_bounds__(LPX_BLOCK) void k_t_price(lpx_tableau s, Partials P, int64_t strip)
{
    __shared__ BlockRed red;
    if (s.ctrl->status != LPX_RUNNING) return;
    const int64_t c0 = (int64_t)blockIdx.x * strip;
    const int64_t c1 = (c0 + strip < s.n) ? c0 + strip : s.n;
    double mn = LPX_INF;
    int64_t jn = LPX_NOIDX;
    for (int64_t j = c0 + threadIdx.x; j < c1; j += LPX_BLOCK) {
        const double rc = s.T[s.m * s.ld + j];
        if (s.pricing == LPX_PRICE_DANTZIG) {
            if (vi_less(rc, t_var(s, j), mn, jn)) { mn = rc; jn = t_var(s, j); }
        } else {
            mn = fmin(mn, rc);
            if (rc < 0.0 && t_var(s, j) < jn) jn = t_var(s, j);
        }
    }
    if (s.pricing == LPX_PRICE_DANTZIG) {
        const VI t = block_argmin(VI{mn, jn}, red);
        mn = t.v;
        jn = t.i;
    } else {
        mn = block_min(mn, red);
        jn = block_imin(jn, red);
    }
    if (threadIdx.x == 0) {
        P.v0[blockIdx.x] = mn;
        P.i0[blockIdx.x] = jn;
    }
}

// entering decision from the per-strip pricing partials: the reference's first-negative rule or opt-in Dantzig
__device__ __forceinline__ void combine_price(int64_t pricing, const Partials &P, int64_t np, BlockRed &red, double &mn,
                                              int64_t &j)
{
    if (pricing == LPX_PRICE_DANTZIG) {
        const VI t = combine_argmin(P.v0, P.i0, np, red);
        mn = t.v;
        j = t.i;
    } else {
        mn = combine_min(P.v0, np, red);
        j = combine_imin(P.i0, np, red);
    }
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
    double mn;
    int64_t j;
    combine_price(s.pricing, P, nprice, red, mn, j);
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
        di = s.T[i * s.ld + t_slot(s, j)];
        d[i] = di;
    }
    t_ratio_partial(s, i, di, f32, R, red);
}

// column-partitioned, part 1: local proposal
__global__ __launch_bounds__(LPX_BLOCK) void k_t_propose(lpx_tableau s, Partials P, int64_t nprice, double *sendbuf)
{
    __shared__ BlockRed red;
    if (s.ctrl->status != LPX_RUNNING) return;
    double mn;
    int64_t j;
    combine_price(s.pricing, P, nprice, red, mn, j);
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        sendbuf[0] = mn;
        sendbuf[1] = (double)j; // exact: column indices < 2^53, LPX_NOIDX = 2^62
    }
    const int64_t i = (int64_t)blockIdx.x * LPX_BLOCK + threadIdx.x;
    if (i <= s.m) sendbuf[2 + i] = (j == LPX_NOIDX) ? 0.0 : s.T[i * s.ld + t_slot(s, j)];
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
        if (s.pricing == LPX_PRICE_DANTZIG) {
            if (vi_less(v, idx, mn, j)) { mn = v; j = idx; win = g; }
        } else {
            mn = fmin(mn, v);
            if (idx < j) { j = idx; win = g; }
        }
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
    const VI t = combine_argmin(R.v0, R.i0, nratio, red);
    const double dmax = combine_max(R.v1, nratio, red);
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
        t_swap_columns(s, j, leaving);
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

// Small tableaus: entering decision, column gather, ratio test, eta column, pivot-row copy and bookkeeping in ONE
// single-workgroup kernel (replaces k_t_column + k_t_prepare, or k_t_select + k_t_prepare when partitioned).
// It is latency-bound, so it is organised as three dependent global-memory round trips (x_B + pricing partials;
// the entering column; the pivot row) with everything else in registers / LDS: each lane keeps its <= 4 rows'
// x_B and d values in registers, and the two block reductions each cost one barrier pair.
#define T_SMALL_RPT 4
#define T_SMALL_THREADS 1024
struct SmallRed {
    double a[LPX_MAXWAVES], b[LPX_MAXWAVES], c[LPX_MAXWAVES];
    int64_t i[LPX_MAXWAVES];
};
template <bool PART>
__global__ __launch_bounds__(T_SMALL_THREADS) void k_t_pre_small(lpx_tableau s, Partials P, int64_t nprice,
                                                                const double *recv, int64_t nranks, int64_t msg,
                                                                double *d_out, double *e, double *rowr)
{
    __shared__ SmallRed red;
    lpx_ctrl *ctrl = s.ctrl;
    if (ctrl->status != LPX_RUNNING) return;
    const bool f32 = ctrl->f32_phase != 0;
    const int tid = threadIdx.x, nthr = blockDim.x, wave = tid >> 6, lane = tid & 63, nw = nthr >> 6;
    const bool lead = tid == 0;
    const int64_t m = s.m;

    // ---- round trip 1: x_B column (independent of the entering decision) + pricing partials / proposals
    double xb[T_SMALL_RPT];
#pragma unroll
    for (int q = 0; q < T_SMALL_RPT; ++q) {
        const int64_t i = tid + (int64_t)q * nthr;
        xb[q] = (i < m) ? s.T[i * s.ld + s.n] : 0.0;
    }
    double mn = LPX_INF;
    int64_t j = LPX_NOIDX, win = 0;
    if (!PART) {
        const bool dz = s.pricing == LPX_PRICE_DANTZIG;
        for (int64_t k = tid; k < nprice; k += nthr) {
            if (dz) {
                if (vi_less(P.v0[k], P.i0[k], mn, j)) { mn = P.v0[k]; j = P.i0[k]; }
            } else {
                mn = fmin(mn, P.v0[k]);
                j = P.i0[k] < j ? P.i0[k] : j;
            }
        }
        if (dz) {
            const VI t = wave_argmin(VI{mn, j});
            mn = t.v;
            j = t.i;
        } else {
            mn = wave_min(mn);
            j = wave_imin(j);
        }
        if (lane == 0) { red.a[wave] = mn; red.i[wave] = j; }
        __syncthreads();
        mn = red.a[0];
        j = red.i[0];
        for (int k = 1; k < nw; ++k) {
            if (dz) {
                if (vi_less(red.a[k], red.i[k], mn, j)) { mn = red.a[k]; j = red.i[k]; }
            } else {
                mn = fmin(mn, red.a[k]);
                j = red.i[k] < j ? red.i[k] : j;
            }
        }
        __syncthreads();
    } else {
        for (int64_t g = 0; g < nranks; ++g) {
            const double v = recv[g * msg];
            const int64_t idx = (int64_t)recv[g * msg + 1];
            if (s.pricing == LPX_PRICE_DANTZIG) {
                if (vi_less(v, idx, mn, j)) { mn = v; j = idx; win = g; }
            } else {
                mn = fmin(mn, v);
                if (idx < j) { j = idx; win = g; }
            }
        }
    }
    const bool opt = lpx_is_optimal_min(mn);
    if (lead) {
        ctrl->counter += 1;
        ctrl->price_val = mn;
        if (opt) ctrl->status = LPX_OPTIMAL;
        else ctrl->enter_j = j;
    }
    if (opt) return;

    // ---- round trip 2: the entering column (d[m] = its reduced cost)
    const double *src = PART ? recv + win * msg + 2 : s.T + t_slot(s, j);
    const int64_t stride = PART ? 1 : s.ld;
    double dv[T_SMALL_RPT];
    VI best{LPX_INF, LPX_NOIDX};
    double dbest = 0.0, dmax = -LPX_INF;
#pragma unroll
    for (int q = 0; q < T_SMALL_RPT; ++q) {
        const int64_t i = tid + (int64_t)q * nthr;
        dv[q] = (i <= m) ? src[i * stride] : 0.0;
    }
#pragma unroll
    for (int q = 0; q < T_SMALL_RPT; ++q) {
        const int64_t i = tid + (int64_t)q * nthr;
        if (i < m) {
            const double th = lpx_psd(xb[q], dv[q], f32);
            if (vi_less(th, i, best.v, best.i)) { best.v = th; best.i = i; dbest = dv[q]; }
            dmax = fmax(dmax, dv[q]);
        }
        if (i <= m) d_out[i] = dv[q];
    }
    // argmin (theta, row) carrying d_row, and max(d): one barrier pair
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        const double ov = __shfl_xor(best.v, off), od = __shfl_xor(dbest, off);
        const int64_t oi = __shfl_xor(best.i, off);
        if (vi_less(ov, oi, best.v, best.i)) { best.v = ov; best.i = oi; dbest = od; }
    }
    dmax = wave_max(dmax);
    if (lane == 0) { red.a[wave] = best.v; red.i[wave] = best.i; red.b[wave] = dbest; red.c[wave] = dmax; }
    __syncthreads();
    best.v = red.a[0]; best.i = red.i[0]; dbest = red.b[0]; dmax = red.c[0];
    for (int k = 1; k < nw; ++k) {
        if (vi_less(red.a[k], red.i[k], best.v, best.i)) { best.v = red.a[k]; best.i = red.i[k]; dbest = red.b[k]; }
        dmax = fmax(dmax, red.c[k]);
    }
    if (dmax <= 0.0) { // primal_solvers.py:144-147
        if (lead) {
            ctrl->aux_val = dmax;
            ctrl->status = LPX_PRIMAL_UNBOUNDED;
        }
        return;
    }
    const int64_t r = best.i;
    const double dr = dbest;

    // ---- round trip 3: old pivot row copy (+ basis[r] for the bookkeeping); eta column from registers
    const int64_t ncopy = lpx_roundup_dev(s.n + 1, 2);
    for (int64_t k = tid; k < ncopy; k += nthr) rowr[k] = s.T[r * s.ld + k];
#pragma unroll
    for (int q = 0; q < T_SMALL_RPT; ++q) {
        const int64_t i = tid + (int64_t)q * nthr;
        if (i <= m) e[i] = (i == r) ? 1.0 / dr : (-dv[q]) / dr; // primal_solvers.py:236-244; e[m] eliminates rc_j
    }
    if (lead) {
        const int64_t leaving = s.basis[r];
        s.basis[r] = j;
        t_swap_columns(s, j, leaving);
        if (s.log && ctrl->pivots < s.log_cap) {
            s.log[3 * ctrl->pivots + 0] = r;
            s.log[3 * ctrl->pivots + 1] = j;
            s.log[3 * ctrl->pivots + 2] = leaving;
        }
        ctrl->pivots += 1;
        ctrl->leave_r = r;
        ctrl->leave_var = leaving;
        ctrl->ratio_val = best.v;
        ctrl->aux_val = dmax;
        ctrl->flip = 0;
        ctrl->f32_phase = 0;
    }
}
static inline bool t_is_small(const lpx_tableau *s)
{
    if (lpx_env_int("LPX_TABLEAU_SMALL", 1) == 0) return false;
    // measured on MI355X: one workgroup gathers a strided column slower than many once m is beyond a few hundred
    // rows (m=1000: 11.4 us vs 3+3 us), so the single-workgroup form is kept for genuinely small tableaus
    return s->m + 1 <= lpx_env_int("LPX_TABLEAU_SMALL_M", 512) && s->n + 2 <= 4096;
}

static inline int64_t t_prep_len(const lpx_tableau *s)
{
    const int64_t a = lpx_roundup(s->n + 1, 2), b = s->m + 1;
    return a > b ? a : b;
}
static void t_launch_update(const lpx_tableau *s, const TabWs &w, const Rank1Plan &p, hipStream_t st, bool price = true)
{
    Rank1Args a{};
    a.M = s->T; a.rows = s->m + 1; a.cols = s->n + 1; a.ld = s->ld; a.e = w.e; a.rowr = w.rowr; a.r = -1;
    a.ctrl = s->ctrl; a.xB = nullptr; a.xr_old = nullptr; a.col0 = s->col0;
    a.slot_var = s->slot_var; a.var_slot = s->var_slot;
    a.price_v0 = price ? w.price.v0 : nullptr; a.price_i0 = price ? w.price.i0 : nullptr; a.pricing = (int)s->pricing;
    lpx_k1_begin(st);
    lpx_rank1_launch<true>(p, a, st);
    lpx_k1_end(st);
}
static void t_launch_price(const lpx_tableau *s, const TabWs &w, const Rank1Plan &p, hipStream_t st)
{
    const int64_t strip = 2 * (int64_t)p.threads * p.vpt;
    hipLaunchKernelGGL(k_t_price, dim3(p.grid.x), dim3(LPX_BLOCK), 0, st, *s, w.price, strip);
}
struct TabKey {
    lpx_tableau s;
    int vpt, unroll, nt, threads;
    int64_t rpb;
    int what;
};

extern "C" int lpx_tableau_iterate(const lpx_tableau *s, int64_t iters, void *stream)
{
    int rc = check_tableau(s);
    if (rc) return rc;
    LPX_REQUIRE(s->slot_var != nullptr || (s->col0 == 0 && s->n_global == s->n), "lpx_tableau_iterate needs an unpartitioned tableau");
    const TabWs w = tab_ws(s->m, s->n, s->workspace);
    hipStream_t st = (hipStream_t)stream;
    const Rank1Plan plan = lpx_rank1_plan(s->m + 1, s->n + 1);
    LPX_REQUIRE(plan.grid.x <= LPX_MAXP, "too many column strips for the partial arrays");
    const int64_t nprice = plan.grid.x, nrow = lpx_cdiv(s->m + 1, LPX_BLOCK);
    if (iters <= 0) return LPX_OK;
    // K1 refreshes the pricing partials of the NEXT iteration; the first iteration of a call scans the row itself
    t_launch_price(s, w, plan, st);
    const lpx_tableau sv = *s;
    const bool small = t_is_small(s);
    auto one = [&](hipStream_t q) {
        if (small) {
            hipLaunchKernelGGL((k_t_pre_small<false>), dim3(1), dim3(T_SMALL_THREADS), 0, q, sv, w.price, nprice,
                               (const double *)nullptr, (int64_t)0, (int64_t)0, w.d, w.e, w.rowr);
        } else {
            hipLaunchKernelGGL(k_t_column, dim3((unsigned)nrow), dim3(LPX_BLOCK), 0, q, sv, w.price, nprice, w.ratio, w.d);
            hipLaunchKernelGGL(k_t_prepare, dim3((unsigned)lpx_cdiv(t_prep_len(&sv), LPX_BLOCK)), dim3(LPX_BLOCK), 0, q, sv,
                               w.ratio, nrow, (const double *)w.d, w.e, w.rowr);
        }
        t_launch_update(&sv, w, plan, q);
    };
    TabKey key;
    memset(&key, 0, sizeof key);
    key.s = sv; key.vpt = plan.vpt; key.unroll = plan.unroll; key.nt = plan.nt; key.threads = plan.threads;
    key.rpb = plan.rows_per_block; key.what = 1;
    return lpx_run_iterations(&key, sizeof key, iters, st, one);
}

extern "C" int lpx_tableau_propose(const lpx_tableau *s, double *sendbuf, int rescan, void *stream)
{
    int rc = check_tableau(s);
    if (rc) return rc;
    LPX_REQUIRE(sendbuf != nullptr, "null send buffer");
    const TabWs w = tab_ws(s->m, s->n, s->workspace);
    hipStream_t st = (hipStream_t)stream;
    const Rank1Plan plan = lpx_rank1_plan(s->m + 1, s->n + 1);
    LPX_REQUIRE(plan.grid.x <= LPX_MAXP, "too many column strips for the partial arrays");
    const int64_t nrow = lpx_cdiv(s->m + 1, LPX_BLOCK);
    if (rescan) t_launch_price(s, w, plan, st);
    hipLaunchKernelGGL(k_t_propose, dim3((unsigned)nrow), dim3(LPX_BLOCK), 0, st, *s, w.price, (int64_t)plan.grid.x, sendbuf);
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
    const Rank1Plan plan = lpx_rank1_plan(s->m + 1, s->n + 1);
    const int64_t nrow = lpx_cdiv(s->m + 1, LPX_BLOCK), msg = lpx_tableau_msg_len(s->m);
    if (t_is_small(s)) {
        hipLaunchKernelGGL((k_t_pre_small<true>), dim3(1), dim3(T_SMALL_THREADS), 0, st, *s, w.price, (int64_t)0, recvbuf, nranks, msg,
                           w.d, w.e, w.rowr);
    } else {
        hipLaunchKernelGGL(k_t_select, dim3((unsigned)nrow), dim3(LPX_BLOCK), 0, st, *s, recvbuf, nranks, msg, w.ratio, w.d);
        hipLaunchKernelGGL(k_t_prepare, dim3((unsigned)lpx_cdiv(t_prep_len(s), LPX_BLOCK)), dim3(LPX_BLOCK), 0, st, *s, w.ratio,
                           nrow, (const double *)w.d, w.e, w.rowr);
    }
    t_launch_update(s, w, plan, st);
    LPX_LAUNCH_CHECK();
    return LPX_OK;
}

// ---------------------------------------------------------------------------------------------------------
// Column-partitioned loop with the exchange of pivot t+1 overlapped with the update of pivot t ("lookahead").
// Once pivot t's leaving row is known (e, rowr are set), the NEXT proposal of a rank needs only two thin pieces of the
// updated tableau: its reduced-cost row (to find the next local entering candidate) and that candidate's column.  Both
// are formed from the old T, e and rowr with K1's own expressions -- bit for bit what K1 is about to write -- so the
// all-gather of pivot t+1 can travel over xGMI while K1 streams the slice through HBM:
//     commit_begin(recv_t, send_t+1): select, prepare, price_next, propose_next      (stream s)
//     all-gather(send_t+1 -> recv_t+1), asynchronous                                  (RCCL's stream, after propose_next)
//     commit_update: K1                                                               (stream s, concurrent with it)
// ---------------------------------------------------------------------------------------------------------
// per-strip pricing partials of the reduced-cost row as K1 will leave it (layout of k_t_price / K1's epilogue)
__global__ __launch_bounds__(LPX_BLOCK) void k_t_price_next(lpx_tableau s, Partials P, int64_t strip, const double *e,
                                                           const double *rowr)
{
    __shared__ BlockRed red;
    if (s.ctrl->status != LPX_RUNNING) return;
    const int64_t unit = t_unit_col(s);
    const double em = e[s.m];
    const int64_t c0 = (int64_t)blockIdx.x * strip;
    const int64_t c1 = (c0 + strip < s.n) ? c0 + strip : s.n;
    double mn = LPX_INF;
    int64_t jn = LPX_NOIDX;
    for (int64_t j = c0 + threadIdx.x; j < c1; j += LPX_BLOCK) {
        // the rewritten column: unit vector (full form, rc 0) or the eta column (condensed, rc e_m)
        const double rc = (j == unit) ? (s.slot_var ? em : 0.0) : fma(em, rowr[j], s.T[s.m * s.ld + j]);
        if (s.pricing == LPX_PRICE_DANTZIG) {
            if (vi_less(rc, t_var(s, j), mn, jn)) { mn = rc; jn = t_var(s, j); }
        } else {
            mn = fmin(mn, rc);
            if (rc < 0.0 && t_var(s, j) < jn) jn = t_var(s, j);
        }
    }
    if (s.pricing == LPX_PRICE_DANTZIG) {
        const VI t = block_argmin(VI{mn, jn}, red);
        mn = t.v;
        jn = t.i;
    } else {
        mn = block_min(mn, red);
        jn = block_imin(jn, red);
    }
    if (threadIdx.x == 0) {
        P.v0[blockIdx.x] = mn;
        P.i0[blockIdx.x] = jn;
    }
}
// the proposal for the next pivot: (min rc, first owned negative column, that column) of the tableau after K1
__global__ __launch_bounds__(LPX_BLOCK) void k_t_propose_next(lpx_tableau s, Partials P, int64_t nprice, const double *e,
                                                             const double *rowr, double *sendbuf)
{
    __shared__ BlockRed red;
    if (s.ctrl->status != LPX_RUNNING) return;
    double mn;
    int64_t j;
    combine_price(s.pricing, P, nprice, red, mn, j);
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        sendbuf[0] = mn;
        sendbuf[1] = (double)j;
    }
    const int64_t i = (int64_t)blockIdx.x * LPX_BLOCK + threadIdx.x;
    if (i > s.m) return;
    double v = 0.0;
    if (j != LPX_NOIDX) {
        const int64_t jl = t_slot(s, j), r = s.ctrl->leave_r;
        if (jl == t_unit_col(s)) v = s.slot_var ? e[i] : ((i == r) ? 1.0 : 0.0);
        else v = (i == r) ? e[i] * rowr[jl] : fma(e[i], rowr[jl], s.T[i * s.ld + jl]);
    }
    sendbuf[2 + i] = v;
}

// select + prepare from the gathered proposals of this pivot; sendbuf_next != NULL: also this rank's proposal for the
// next pivot.  The update itself is lpx_tableau_commit_update.
extern "C" int lpx_tableau_commit_begin(const lpx_tableau *s, const double *recvbuf, int64_t nranks, double *sendbuf_next,
                                        void *stream)
{
    int rc = check_tableau(s);
    if (rc) return rc;
    LPX_REQUIRE(recvbuf != nullptr && nranks >= 1 && nranks <= 64, "bad receive buffer / rank count");
    const TabWs w = tab_ws(s->m, s->n, s->workspace);
    hipStream_t st = (hipStream_t)stream;
    const Rank1Plan plan = lpx_rank1_plan(s->m + 1, s->n + 1);
    LPX_REQUIRE(plan.grid.x <= LPX_MAXP, "too many column strips for the partial arrays");
    const int64_t nrow = lpx_cdiv(s->m + 1, LPX_BLOCK), msg = lpx_tableau_msg_len(s->m);
    if (t_is_small(s)) {
        hipLaunchKernelGGL((k_t_pre_small<true>), dim3(1), dim3(T_SMALL_THREADS), 0, st, *s, w.price, (int64_t)0, recvbuf, nranks, msg,
                           w.d, w.e, w.rowr);
    } else {
        hipLaunchKernelGGL(k_t_select, dim3((unsigned)nrow), dim3(LPX_BLOCK), 0, st, *s, recvbuf, nranks, msg, w.ratio, w.d);
        hipLaunchKernelGGL(k_t_prepare, dim3((unsigned)lpx_cdiv(t_prep_len(s), LPX_BLOCK)), dim3(LPX_BLOCK), 0, st, *s, w.ratio,
                           nrow, (const double *)w.d, w.e, w.rowr);
    }
    if (sendbuf_next != nullptr) {
        const int64_t strip = 2 * (int64_t)plan.threads * plan.vpt;
        hipLaunchKernelGGL(k_t_price_next, dim3(plan.grid.x), dim3(LPX_BLOCK), 0, st, *s, w.price, strip, (const double *)w.e,
                           (const double *)w.rowr);
        hipLaunchKernelGGL(k_t_propose_next, dim3((unsigned)nrow), dim3(LPX_BLOCK), 0, st, *s, w.price, (int64_t)plan.grid.x,
                           (const double *)w.e, (const double *)w.rowr, sendbuf_next);
    }
    LPX_LAUNCH_CHECK();
    return LPX_OK;
}
extern "C" int lpx_tableau_commit_update(const lpx_tableau *s, void *stream)
{
    int rc = check_tableau(s);
    if (rc) return rc;
    const TabWs w = tab_ws(s->m, s->n, s->workspace);
    const Rank1Plan plan = lpx_rank1_plan(s->m + 1, s->n + 1);
    // the pricing partials of the next pivot exist already (k_t_price_next) or are rescanned by lpx_tableau_propose
    t_launch_update(s, w, plan, (hipStream_t)stream, false);
    LPX_LAUNCH_CHECK();
    return LPX_OK;
}

// ---------------------------------------------------------------------------------------------------------
// Dual simplex on the same tableau (DualNaiveSimplexSolver.solve, dual_solvers.py:47-108): the leaving row is the
// first x_B < 0 (x_B = column n, replicated when partitioned), the dual row is row r of T with basic columns
// forced to 0 (:38-39), the reduced costs are row m, the ratio test (-rc_j)/row_j over row_j < 0 with first argmin
// (:61-80).  The pivot itself is K1, unchanged.  Chain: k_td_row -> k_td_ratio -> k_td_prepare -> K1.
// ---------------------------------------------------------------------------------------------------------

// K5 (tableau): one partial (min x_B, first row with x_B < 0) per workgroup of rows
__global__ __launch_bounds__(LPX_BLOCK) void k_td_row(lpx_tableau s, Partials R)
{
    __shared__ BlockRed red;
    if (s.ctrl->status != LPX_RUNNING) return;
    const int64_t i = (int64_t)blockIdx.x * LPX_BLOCK + threadIdx.x;
    const double v = (i < s.m) ? s.T[i * s.ld + s.n] : LPX_INF;
    const double mn = block_min(v, red);
    const int64_t first = block_imin((i < s.m && v < 0.0) ? i : LPX_NOIDX, red);
    if (threadIdx.x == 0) {
        R.v0[blockIdx.x] = mn;
        R.i0[blockIdx.x] = first;
    }
}

// K6 (tableau): leaving-row decision (every workgroup re-derives it from the row partials) + one partial
// (min theta with first global column, min of the dual row) per column strip
__global__ __launch_bounds__(LPX_BLOCK) void k_td_ratio(lpx_tableau s, Partials R, int64_t nrow, Partials P, int64_t strip)
{
    __shared__ BlockRed red;
    lpx_ctrl *ctrl = s.ctrl;
    if (ctrl->status != LPX_RUNNING) return;
    const bool f32 = ctrl->f32_phase != 0;
    const double mn = combine_min(R.v0, nrow, red);
    const int64_t r = combine_imin(R.i0, nrow, red);
    const bool opt = lpx_is_optimal_min(mn); // dual_solvers.py:52-54
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        ctrl->counter += 1;
        ctrl->price_val = mn;
        if (opt) ctrl->status = LPX_OPTIMAL;
        else ctrl->leave_r = r;
    }
    if (opt) return;
    const int64_t c0 = (int64_t)blockIdx.x * strip;
    const int64_t c1 = (c0 + strip < s.n) ? c0 + strip : s.n;
    VI best{LPX_INF, LPX_NOIDX};
    double rowmin = LPX_INF;
    for (int64_t j = c0 + threadIdx.x; j < c1; j += LPX_BLOCK) {
        // basic columns count as 0 (dual_solvers.py:38-39); the condensed form carries none
        const double row = (!s.slot_var && s.var_state[j] == LPX_VS_BASIC) ? 0.0 : s.T[r * s.ld + j];
        const double th = lpx_dsd(s.T[s.m * s.ld + j], row, f32);
        if (vi_less(th, t_var(s, j), best.v, best.i)) { best.v = th; best.i = t_var(s, j); }
        rowmin = fmin(rowmin, row);
    }
    best = block_argmin(best, red);
    rowmin = block_min(rowmin, red);
    if (threadIdx.x == 0) {
        P.v0[blockIdx.x] = best.v;
        P.i0[blockIdx.x] = best.i;
        P.v1[blockIdx.x] = rowmin;
    }
}

// shared tail of the dual prepare kernels: eta column from the entering column d (d[m] = its reduced cost), copy of
// the old pivot row, bookkeeping
__device__ __forceinline__ void td_finish(const lpx_tableau &s, int64_t r, int64_t j, double dr, double di, double theta,
                                          double rowmin, double *d, double *e, double *rowr)
{
    lpx_ctrl *ctrl = s.ctrl;
    const int64_t gi = (int64_t)blockIdx.x * LPX_BLOCK + threadIdx.x;
    if (gi <= s.m) {
        d[gi] = di;
        e[gi] = (gi == r) ? 1.0 / dr : (-di) / dr; // primal_solvers.py:236-244; e[m] eliminates rc_j
    }
    if (gi < lpx_roundup_dev(s.n + 1, 2)) rowr[gi] = s.T[r * s.ld + gi];
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        const int64_t leaving = s.basis[r];
        s.basis[r] = j;
        t_swap_columns(s, j, leaving);
        if (s.log && ctrl->pivots < s.log_cap) {
            s.log[3 * ctrl->pivots + 0] = r;
            s.log[3 * ctrl->pivots + 1] = j;
            s.log[3 * ctrl->pivots + 2] = leaving;
        }
        ctrl->pivots += 1;
        ctrl->enter_j = j;
        ctrl->leave_var = leaving;
        ctrl->ratio_val = theta;
        ctrl->aux_val = rowmin;
        ctrl->flip = 0;
        ctrl->f32_phase = 0;
    }
}

// unpartitioned: entering decision from the strip partials, dual unboundedness (:41-44, :56-59), column gather
__global__ __launch_bounds__(LPX_BLOCK) void k_td_prepare(lpx_tableau s, Partials P, int64_t nprice, double *d, double *e,
                                                         double *rowr)
{
    __shared__ BlockRed red;
    lpx_ctrl *ctrl = s.ctrl;
    if (ctrl->status != LPX_RUNNING) return;
    const VI t = combine_argmin(P.v0, P.i0, nprice, red);
    const double rowmin = combine_min(P.v1, nprice, red);
    if (rowmin >= 0.0) {
        if (blockIdx.x == 0 && threadIdx.x == 0) {
            ctrl->aux_val = rowmin;
            ctrl->status = LPX_DUAL_UNBOUNDED;
        }
        return;
    }
    const int64_t r = ctrl->leave_r, j = t.i; // leave_r: written by k_td_ratio, only read here
    const int64_t jl = t_slot(s, j);
    const double dr = s.T[r * s.ld + jl];
    const int64_t gi = (int64_t)blockIdx.x * LPX_BLOCK + threadIdx.x;
    const double di = (gi <= s.m) ? s.T[gi * s.ld + jl] : 0.0;
    td_finish(s, r, j, dr, di, t.v, rowmin, d, e, rowr);
}

// column-partitioned, part 1: local proposal (min theta, its global column, that column of T)
__global__ __launch_bounds__(LPX_BLOCK) void k_td_propose(lpx_tableau s, Partials P, int64_t nprice, double *sendbuf)
{
    __shared__ BlockRed red;
    if (s.ctrl->status != LPX_RUNNING) return;
    const VI t = combine_argmin(P.v0, P.i0, nprice, red);
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        sendbuf[0] = t.v;
        sendbuf[1] = (double)t.i;
    }
    const int64_t i = (int64_t)blockIdx.x * LPX_BLOCK + threadIdx.x;
    if (i <= s.m) sendbuf[2 + i] = s.T[i * s.ld + t_slot(s, t.i)];
}

// column-partitioned, part 2: global entering column = first argmin of theta over the ranks' proposals ("lowest variable
// index on ties" is np.argmin's first minimum, whichever rank holds the variable).  No candidate anywhere
// (every theta = +inf) is exactly min(row) >= 0: dual unbounded.
__global__ __launch_bounds__(LPX_BLOCK) void k_td_commit(lpx_tableau s, const double *recv, int64_t nranks, int64_t msg, double *d,
                                                        double *e, double *rowr)
{
    lpx_ctrl *ctrl = s.ctrl;
    if (ctrl->status != LPX_RUNNING) return;
    VI best{LPX_INF, LPX_NOIDX};
    int64_t win = 0;
    for (int64_t g = 0; g < nranks; ++g) {
        const double v = recv[g * msg];
        const int64_t idx = (int64_t)recv[g * msg + 1];
        if (vi_less(v, idx, best.v, best.i)) { best.v = v; best.i = idx; win = g; }
    }
    if (best.v == LPX_INF) {
        if (blockIdx.x == 0 && threadIdx.x == 0) {
            ctrl->aux_val = 0.0;
            ctrl->status = LPX_DUAL_UNBOUNDED;
        }
        return;
    }
    const int64_t r = ctrl->leave_r;
    const double *col = recv + win * msg + 2;
    const int64_t gi = (int64_t)blockIdx.x * LPX_BLOCK + threadIdx.x;
    td_finish(s, r, best.i, col[r], (gi <= s.m) ? col[gi] : 0.0, best.v, -1.0, d, e, rowr);
}

static void td_launch_select(const lpx_tableau *s, const TabWs &w, const Rank1Plan &p, hipStream_t st)
{
    const int64_t nrow = lpx_cdiv(s->m, LPX_BLOCK), strip = 2 * (int64_t)p.threads * p.vpt;
    hipLaunchKernelGGL(k_td_row, dim3((unsigned)nrow), dim3(LPX_BLOCK), 0, st, *s, w.ratio);
    hipLaunchKernelGGL(k_td_ratio, dim3(p.grid.x), dim3(LPX_BLOCK), 0, st, *s, w.ratio, nrow, w.price, strip);
}

extern "C" int lpx_tableau_dual_iterate(const lpx_tableau *s, int64_t iters, void *stream)
{
    int rc = check_tableau(s);
    if (rc) return rc;
    LPX_REQUIRE(s->slot_var != nullptr || (s->col0 == 0 && s->n_global == s->n), "lpx_tableau_dual_iterate needs an unpartitioned tableau");
    const TabWs w = tab_ws(s->m, s->n, s->workspace);
    hipStream_t st = (hipStream_t)stream;
    const Rank1Plan plan = lpx_rank1_plan(s->m + 1, s->n + 1);
    LPX_REQUIRE(plan.grid.x <= LPX_MAXP, "too many column strips for the partial arrays");
    if (iters <= 0) return LPX_OK;
    const lpx_tableau sv = *s;
    auto one = [&](hipStream_t q) {
        td_launch_select(&sv, w, plan, q);
        hipLaunchKernelGGL(k_td_prepare, dim3((unsigned)lpx_cdiv(t_prep_len(&sv), LPX_BLOCK)), dim3(LPX_BLOCK), 0, q, sv, w.price,
                           (int64_t)plan.grid.x, w.d, w.e, w.rowr);
        t_launch_update(&sv, w, plan, q);
    };
    TabKey key;
    memset(&key, 0, sizeof key);
    key.s = sv; key.vpt = plan.vpt; key.unroll = plan.unroll; key.nt = plan.nt; key.threads = plan.threads;
    key.rpb = plan.rows_per_block; key.what = 2;
    return lpx_run_iterations(&key, sizeof key, iters, st, one);
}

extern "C" int lpx_tableau_dual_propose(const lpx_tableau *s, double *sendbuf, void *stream)
{
    int rc = check_tableau(s);
    if (rc) return rc;
    LPX_REQUIRE(sendbuf != nullptr, "null send buffer");
    const TabWs w = tab_ws(s->m, s->n, s->workspace);
    hipStream_t st = (hipStream_t)stream;
    const Rank1Plan plan = lpx_rank1_plan(s->m + 1, s->n + 1);
    LPX_REQUIRE(plan.grid.x <= LPX_MAXP, "too many column strips for the partial arrays");
    td_launch_select(s, w, plan, st);
    hipLaunchKernelGGL(k_td_propose, dim3((unsigned)lpx_cdiv(s->m + 1, LPX_BLOCK)), dim3(LPX_BLOCK), 0, st, *s, w.price,
                       (int64_t)plan.grid.x, sendbuf);
    LPX_LAUNCH_CHECK();
    return LPX_OK;
}

extern "C" int lpx_tableau_dual_commit(const lpx_tableau *s, const double *recvbuf, int64_t nranks, void *stream)
{
    int rc = check_tableau(s);
    if (rc) return rc;
    LPX_REQUIRE(recvbuf != nullptr && nranks >= 1 && nranks <= 64, "bad receive buffer / rank count");
    const TabWs w = tab_ws(s->m, s->n, s->workspace);
    hipStream_t st = (hipStream_t)stream;
    const Rank1Plan plan = lpx_rank1_plan(s->m + 1, s->n + 1);
    hipLaunchKernelGGL(k_td_commit, dim3((unsigned)lpx_cdiv(t_prep_len(s), LPX_BLOCK)), dim3(LPX_BLOCK), 0, st, *s, recvbuf, nranks,
                       lpx_tableau_msg_len(s->m), w.d, w.e, w.rowr);
    t_launch_update(s, w, plan, st);
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
extern "C" int lpx_tableau_expand(const lpx_tableau *s, const float *c_global, double *x, double *cost, void *stream)
{
    int rc = check_tableau(s);
    if (rc) return rc;
    LPX_REQUIRE(c_global && x && cost, "null pointer");
    hipStream_t st = (hipStream_t)stream;
    hipLaunchKernelGGL(k_t_fill, dim3((unsigned)lpx_cdiv(s->n_global, LPX_BLOCK)), dim3(LPX_BLOCK), 0, st, x, s->n_global);
    hipLaunchKernelGGL(k_t_scatter, dim3((unsigned)lpx_cdiv(s->m, LPX_BLOCK)), dim3(LPX_BLOCK), 0, st, *s, x);
    hipLaunchKernelGGL(k_cost, dim3(1), dim3(1024), 0, st, c_global, (const double *)x, s->n_global, cost);
    LPX_LAUNCH_CHECK();
    return LPX_OK;
}
