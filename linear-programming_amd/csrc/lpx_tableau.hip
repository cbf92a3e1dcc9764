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

#include "lpx_tableau_blocked_decl.hpp"
#include "lpx_tableau_dev.hpp"
#include "lpx_tableau_small_decl.hpp"

#include <rccl/rccl.h>

// the one collective of a column-partitioned pivot, issued from the library's own loop (lpx_tableau_dist_iterate): an
// all-gather of the ranks' proposals, RCCL over xGMI, stream-ordered
struct TbDist {
    ncclComm_t comm;
    int64_t nranks, msg;
    double *send, *recv;
};
#define LPX_NCCL(call)                                                                            \
    do {                                                                                          \
        ncclResult_t _r = (call);                                                                 \
        if (_r != ncclSuccess) {                                                                  \
            lpx_set_error("%s failed: %s (%s:%d)", #call, ncclGetErrorString(_r), __FILE__, __LINE__); \
            return LPX_E_HIP;                                                                     \
        }                                                                                         \
    } while (0)

// ratio-test partial of the single-launch chain: one record per lookahead workgroup (<= TF_MAXLA)
#define TF_MAXLA 64
struct FusedPart {
    double *th;    // min ratio over the workgroup's rows
    int64_t *row;  // first row attaining it (LPX_NOIDX: none)
    double *dmax;  // max of the entering column over the rows
    double *dbest; // entering-column entry of that row (the pivot element if it wins)
    int64_t *bvar; // basic variable of that row
};
struct TabWs {
    double *d, *e, *rowr, *scal;
    Partials price, ratio;
    // single-launch chain (k_t_fused): everything a launch reads was written by the launch before it, so the entering
    // column, the ratio partials and the entering decision exist twice and alternate with the parity of the pivot
    double *d2[2];
    FusedPart fp[2];
    double *next[2]; // primal: {min rc, entering variable, its column slot, optimal?}; dual: the TD_* record
    int64_t *base;   // pivots at the start of the current lpx_tableau_iterate call
    double *pend;    // partitioned dual loop with lookahead: {leaving row, optimal?, min x_B} decided one pivot ahead
    TbWs tb;         // deferred-update chain (lpx_tableau_blocked.hpp)
    TbWs tb2;        // pipelined chain: the pending pairs of every other block (E, Et, R, prow, pslot, rowflag, cnt of its own)
    Partials ratio2; // Harris-style ratio test (opt-in): second-pass partials (largest pivot among the admissible rows)
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
    for (int q = 0; q < 2; ++q) {
        w.d2[q] = (double *)tcarve(p, 8 * (m + 2));
        w.fp[q].th = (double *)tcarve(p, 8 * TF_MAXLA);
        w.fp[q].row = (int64_t *)tcarve(p, 8 * TF_MAXLA);
        w.fp[q].dmax = (double *)tcarve(p, 8 * TF_MAXLA);
        w.fp[q].dbest = (double *)tcarve(p, 8 * TF_MAXLA);
        w.fp[q].bvar = (int64_t *)tcarve(p, 8 * TF_MAXLA);
        w.next[q] = (double *)tcarve(p, 8 * 16);
    }
    w.base = (int64_t *)tcarve(p, 8 * 2);
    w.pend = (double *)tcarve(p, 8 * 4);
    w.tb.ldr = lpx_roundup(n + 1, 2) + 2;
    w.tb.E = (double *)tcarve(p, 8 * (m + 2) * TB_KMAX);
    w.tb.lde = lpx_roundup(m + 2, 2);
    w.tb.Et = (double *)tcarve(p, 8 * w.tb.lde * TB_KMAX);
    w.tb.R = (double *)tcarve(p, 8 * w.tb.ldr * TB_KMAX);
    w.tb.rc = (double *)tcarve(p, 8 * (n + 2));
    w.tb.xb = (double *)tcarve(p, 8 * (m + 2));
    w.tb.prow = (int64_t *)tcarve(p, 8 * TB_KMAX);
    w.tb.pslot = (int64_t *)tcarve(p, 8 * TB_KMAX);
    w.tb.rowflag = (uint32_t *)tcarve(p, 4 * (m + 2));
    w.tb.cnt = (int64_t *)tcarve(p, 8 * 2);
    w.tb.scal = (double *)tcarve(p, 8 * 16);
    w.tb.dbest = (double *)tcarve(p, 8 * LPX_MAXP);
    w.tb.xbest = (double *)tcarve(p, 8 * LPX_MAXP);
    w.tb.tsel = (int64_t *)tcarve(p, 8 * 2);
    w.tb2 = w.tb;
    w.tb2.E = (double *)tcarve(p, 8 * (m + 2) * TB_KMAX);
    w.tb2.Et = (double *)tcarve(p, 8 * w.tb.lde * TB_KMAX);
    w.tb2.R = (double *)tcarve(p, 8 * w.tb.ldr * TB_KMAX);
    w.tb2.prow = (int64_t *)tcarve(p, 8 * TB_KMAX);
    w.tb2.pslot = (int64_t *)tcarve(p, 8 * TB_KMAX);
    w.tb2.rowflag = (uint32_t *)tcarve(p, 4 * (m + 2));
    w.tb2.cnt = (int64_t *)tcarve(p, 8 * 2);
    tcarve_partials(p, w.ratio2);
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
    LPX_REQUIRE(s->ratio_rule == LPX_RATIO_TEXTBOOK || (s->ratio_rule == LPX_RATIO_HARRIS && s->ratio_tol >= 0.0 && s->ratio_tol < 1.0),
                "bad ratio rule / tolerance");
    return LPX_OK;
}

// Condensed form (slot_var != NULL): T holds only the NONBASIC columns -- a basic column is a unit vector that the
// rank-1 update leaves untouched (its pivot-row entry is 0), so carrying it costs memory and 16 bytes of HBM traffic per
// element and pivot for nothing.  Column slot k holds variable slot_var[k]; var_slot[] is the inverse map (-1: not in a
// slot of this rank).  On a pivot the entering variable's slot takes over the leaving variable, whose column is the eta
// column e itself (unit vector e_r updated: 0 + e_i * 1), written by K1 where the full form writes the new unit vector.
__device__ __forceinline__ int64_t t_unit_col(const lpx_tableau &s) { return s.ctrl->enter_slot; }

// K3 (tableau): scan the carried reduced-cost row: one partial (min rc, first global column with rc<0) per
// column strip of K1's launch plan, i.e. in the same layout K1's fused pricing epilogue writes afterwards.
__global__ __launch_bounds__(LPX_BLOCK) void k_t_price(lpx_tableau s, Partials P, int64_t strip)
{
    __shared__ BlockRed red;
    if (s.ctrl->status != LPX_RUNNING) return;
    const int64_t c0 = (int64_t)blockIdx.x * strip;
    const int64_t c1 = (blockIdx.x == gridDim.x - 1) ? s.n : c0 + strip; // the last strip may be folded (lpx_rank1.hpp)
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
    // opt-in Harris-style test, pass 1: the largest step a tol-relaxed feasibility allows (float64 whatever the phase)
    if (s.ratio_rule == LPX_RATIO_HARRIS) t.v = (in && di > 0.0) ? (s.T[i * s.ld + s.n] + s.ratio_tol) / di : LPX_INF;
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

// Opt-in Harris-style ratio test, pass 2 (no counterpart in the reference, whose only test is primal_solvers.py:149-167;
// oracle: lpo_ratio_harris): theta_max = min of the pass-1 partials; among the rows with d_i > 0 and x_i / d_i <= theta_max
// the LARGEST pivot d_i wins, first row on ties.  Partials in k_t_prepare's format: (-d_i, row) for its argmin, max d.
__global__ __launch_bounds__(LPX_BLOCK) void k_t_harris(lpx_tableau s, Partials R, int64_t nratio, const double *d, Partials R2)
{
    __shared__ BlockRed red;
    if (s.ctrl->status != LPX_RUNNING) return;
    const double tmax = combine_min(R.v0, nratio, red);
    const int64_t i = (int64_t)blockIdx.x * LPX_BLOCK + threadIdx.x;
    const bool in = i < s.m;
    const double di = in ? d[i] : 0.0;
    const bool ok = in && di > 0.0 && s.T[i * s.ld + s.n] / di <= tmax;
    const VI t = block_argmin(VI{ok ? -di : LPX_INF, ok ? i : LPX_NOIDX}, red);
    if (threadIdx.x == 0) {
        R2.v0[blockIdx.x] = t.v;
        R2.i0[blockIdx.x] = t.i;
        R2.v1[blockIdx.x] = R.v1[blockIdx.x]; // max d of the same rows
    }
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

// ---------------------------------------------------------------------------------------------------------
// Single-launch pivot chain for tableaus that are launch-bound rather than HBM-bound (BASELINE config 2: 1000 x 2000).
// The three-kernel chain above costs three launches of ~5 us for 16 MB of traffic.  Here ONE launch does a whole pivot:
//   * the tableau ping-pongs between two buffers (T and T_alt): a launch reads Tin and writes Tout, so the old pivot row
//     and every old column stay readable for the whole launch -- no pivot-row copy, no eta array (e_i = -d_i / d_r is
//     formed per row on the fly), no in-place race;
//   * a few extra "lookahead" workgroups, placed first in the grid, work on the NEXT pivot while the others stream the
//     update: they form the reduced-cost row the update is about to write (same expressions, bit for bit), price it
//     (primal_solvers.py:135-142), form that entering column and x_B as the update will leave them, and run the ratio
//     test (:149-167) over their rows.  Their records (entering decision, column, per-workgroup ratio partials) are what
//     the next launch starts from: its workgroups combine the <= 64 partials in their prologue (uniform loads).
// So the kernel boundary is the only grid-wide synchronisation of a pivot.  A launch never reads what it writes: the
// records alternate between two copies with the pivot's parity `par`, a kernel ARGUMENT (the host enqueues iteration t
// of a call with par = t & 1; launches after a terminal status are no-ops, so host and device parity cannot drift; the
// hipGraph unroll is even).  A bootstrap launch (BOOT) prices the carried row itself at the start of every call, and
// a final launch copies T_alt back to T when the call ends on odd parity, so every other entry point finds the state in T.
// ---------------------------------------------------------------------------------------------------------
#define TF_THREADS 256
#define TF_UN 4
struct FusedArgs {
    lpx_tableau s;
    const double *Tin;
    double *Tout;
    FusedPart Rc, Rn;     // ratio partials of this pivot (read) / of the next one (written)
    const double *dc;     // entering column of this pivot, m+1 entries (dc[m] = its reduced cost)
    double *dn;
    const double *nc;     // entering decision of this pivot
    double *nn;
    int64_t *base;
    int64_t nla;          // lookahead workgroups = cdiv(m + 1, TF_THREADS)
    int64_t nstrips;      // column strips of 2 * TF_THREADS columns
    int64_t rpb;          // rows per update workgroup
};

// lookahead workgroup q: pricing of the next reduced-cost row (whole row, every lookahead workgroup redundantly), then the
// next entering column and the ratio partial over rows [q * TF_THREADS, (q+1) * TF_THREADS).  BOOT: the state in Tin is
// priced as it is (no pending update); r, dr, j, k, leaving are then unused.
// The kernels below are latency chains: what they cost is the number of DEPENDENT global-memory round trips on the
// longest path (tools/launch_floor.hip: ~2.1 us for an empty launch in a replayed graph, +0.3-0.4 us per dependent
// trip, +1.2 us for the 8 MB stream beside it).  So every load whose address does not depend on this pivot's leaving
// row is issued in the first instructions of the kernel, before the status test (all those addresses are valid
// whatever the status), reductions are one DPP wave reduction + one LDS exchange, and loops over memory are unrolled
// in batches whose loads all go out before the first use.
// -DLPX_TF_STAMPS: diagnostic build, lookahead workgroup 0 prints where its time goes (100 MHz ticks since kernel start)
#ifdef LPX_TF_STAMPS
__device__ long long g_tf_t0;
__device__ long long g_tf_stamp[16];
#define LPX_TF_STAMP(k) do { if (blockIdx.x == 0 && threadIdx.x == 0) g_tf_stamp[k] = (long long)wall_clock64() - g_tf_t0; } while (0)
#else
#define LPX_TF_STAMP(k) do { } while (0)
#endif
struct TfPrice {
    double v;     // min reduced cost (first-negative rule) / the candidate's reduced cost (Dantzig)
    int64_t var;  // entering candidate: variable id
    int64_t slot; // its column in T
};
struct TfRatio {
    double th;    // min ratio
    int64_t i;    // first row attaining it
    double dv;    // entering-column entry of that row
    int64_t bv;   // its basic variable
    double dmax;  // max of the entering column
};
__device__ __forceinline__ TfRatio tf_ratio_op(TfRatio x, TfRatio y);
// Wave reduction of ratio records whose row indices ASCEND with the lane (lanes without a row hold th = +inf, i = NOIDX
// and come last): the first lane attaining the wave's minimum ratio is the first argmin (np.argmin, primal_solvers.py:166),
// so only the ratio itself (2 dwords) and max(d) go through DPP; the rest is read from the winning lane.
__device__ __forceinline__ TfRatio tf_wave_ratio(TfRatio t)
{
    const double mn = wave_min(t.th);
    const double dmax = wave_max(t.dmax);
    const unsigned long long who = __ballot(t.th == mn); // never empty: +inf == +inf (no NaN: d > 0 is checked first)
    const int src = __builtin_amdgcn_readfirstlane(__ffsll((long long)who) - 1);
    return TfRatio{mn, tf_readlane_i64(t.i, src), tf_readlane_f64(t.dv, src), tf_readlane_i64(t.bv, src), dmax};
}
__device__ __forceinline__ TfRatio tf_ratio_op(TfRatio x, TfRatio y)
{
    TfRatio u = vi_less(y.th, y.i, x.th, x.i) ? y : x;
    u.dmax = fmax(x.dmax, y.dmax);
    return u;
}
// round trip 1 of a lookahead workgroup: its rows' x_B, basic variables and current entering-column entries, and the
// first batch of the reduced-cost row with the variable ids of its slots
struct TfLaPre {
    double xi, dci, dcm;
    int64_t bsi;
    double rcm[TF_UN];
    int64_t var[TF_UN];
};
template <bool BOOT>
__device__ __forceinline__ TfLaPre tf_la_preload(const FusedArgs &a, int64_t q)
{
    const lpx_tableau &s = a.s;
    const int64_t m = s.m, n = s.n, ld = s.ld;
    const int64_t i = q * TF_THREADS + threadIdx.x;
    TfLaPre p;
    p.xi = (i <= m) ? a.Tin[i * ld + n] : 0.0;
    p.dci = (!BOOT && i <= m) ? a.dc[i] : 0.0;
    p.dcm = BOOT ? 0.0 : a.dc[m];
    p.bsi = (i < m) ? s.basis[i] : -1;
#pragma unroll
    for (int u = 0; u < TF_UN; ++u) {
        const int64_t c = threadIdx.x + (int64_t)u * TF_THREADS;
        p.rcm[u] = (c < n) ? a.Tin[m * ld + c] : LPX_INF;
        p.var[u] = (c >= n) ? LPX_NOIDX : (s.slot_var ? s.slot_var[c] : s.col0 + c);
    }
    return p;
}

// lookahead workgroup q: pricing of the next reduced-cost row (whole row, every lookahead workgroup redundantly), then the
// next entering column and the ratio partial over rows [q * TF_THREADS, (q+1) * TF_THREADS).  BOOT: the state in Tin is
// priced as it is (no pending update); r, dr, j, k, leaving are then unused.
template <bool BOOT>
__device__ __forceinline__ void tf_lookahead(const FusedArgs &a, int64_t q, const TfLaPre &pre, int64_t r, double dr,
                                             int64_t j, int64_t k, int64_t leaving, bool f32)
{
    __shared__ TfPrice shp[TF_THREADS / 64];
    __shared__ TfRatio shr[TF_THREADS / 64];
    const lpx_tableau &s = a.s;
    const int64_t m = s.m, n = s.n, ld = s.ld;
    const bool cond = s.slot_var != nullptr;
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const double *Tin = a.Tin;
    const bool dz = s.pricing == LPX_PRICE_DANTZIG;
    const int64_t i = q * TF_THREADS + tid;
    const bool in = i < m;
    LPX_TF_STAMP(1);
    // round trip 2 (not BOOT): the old pivot row under this thread's columns, and x_B[r]
    const double xr = (!BOOT && i <= m) ? Tin[r * ld + n] : 0.0;
    const int64_t bsi = (!BOOT && i == r) ? j : pre.bsi; // the basis as this pivot leaves it
    const double em = BOOT ? 0.0 : (-pre.dcm) / dr;      // multiplier of the reduced-cost row (k_t_prepare's e[m])
    TfPrice p{LPX_INF, LPX_NOIDX, -1};
    for (int64_t c0 = tid; c0 < n; c0 += TF_THREADS * TF_UN) {
        double pr[TF_UN], rcm[TF_UN];
        int64_t var[TF_UN];
#pragma unroll
        for (int u = 0; u < TF_UN; ++u) {
            const int64_t c = c0 + (int64_t)u * TF_THREADS;
            const bool ok = c < n;
            pr[u] = (ok && !BOOT) ? Tin[r * ld + c] : 0.0;
            if (c0 == tid) { // first batch: loaded before the leaving row was known
                rcm[u] = pre.rcm[u];
                var[u] = pre.var[u];
            } else {
                rcm[u] = ok ? Tin[m * ld + c] : LPX_INF;
                var[u] = !ok ? LPX_NOIDX : (cond ? s.slot_var[c] : s.col0 + c);
            }
        }
#pragma unroll
        for (int u = 0; u < TF_UN; ++u) {
            const int64_t c = c0 + (int64_t)u * TF_THREADS;
            if (c >= n) continue;
            double rc = rcm[u];
            int64_t v = var[u];
            if (!BOOT) {
                if (c == k) { // the column the update rewrites as a whole: eta column (condensed) / unit vector (full)
                    rc = cond ? em : 0.0;
                    v = cond ? leaving : s.col0 + c;
                } else {
                    rc = fma(em, pr[u], rc);
                }
            }
            if (dz) {
                if (vi_less(rc, v, p.v, p.var)) p = TfPrice{rc, v, c};
            } else {
                p.v = fmin(p.v, rc);
                if (rc < 0.0 && v < p.var) { p.var = v; p.slot = c; }
            }
        }
    }
    // wave reduction on the narrowest payloads: min rc (2 dwords), smallest negative variable id (2 dwords), then the
    // slot from the lane that holds that id -- a struct reduction moves 6 dwords through every DPP step
    const auto pop = [dz](TfPrice x, TfPrice y) {
        if (dz) return vi_less(y.v, y.var, x.v, x.var) ? y : x;
        TfPrice t = (y.var < x.var) ? y : x; // first negative index: smallest variable id
        t.v = fmin(x.v, y.v);
        return t;
    };
    LPX_TF_STAMP(2);
    if (dz) {
        p = wave_allreduce(p, pop);
    } else {
        const int64_t myvar = p.var;
        p.v = wave_min(p.v);
        p.var = wave_imin(myvar);
        const unsigned long long who = __ballot(myvar == p.var); // variable ids are distinct; all lanes when none is negative
        p.slot = tf_readlane_i64(p.slot, __builtin_amdgcn_readfirstlane(__ffsll((long long)who) - 1));
    }
    LPX_TF_STAMP(3);
    if (lane == 0) shp[wave] = p;
    __syncthreads();
    p = shp[0];
#pragma unroll
    for (int w = 1; w < TF_THREADS / 64; ++w) p = pop(p, shp[w]);
    const int64_t kp = p.slot;
    const bool opt = lpx_is_optimal_min(p.v);
    if (q == 0 && tid == 0) {
        a.nn[0] = p.v;
        a.nn[1] = (double)p.var; // exact: ids < 2^53, LPX_NOIDX = 2^62
        a.nn[2] = (double)kp;
        a.nn[3] = opt ? 1.0 : 0.0;
    }
    LPX_TF_STAMP(4);
    if (opt || kp < 0) return;
    // round trip 3: the next entering column as the update will leave it
    double dv = 0.0, xb = pre.xi;
    if (i <= m) {
        if (BOOT) {
            dv = Tin[i * ld + kp];
        } else {
            const double ei = (i == r) ? 1.0 / dr : (-pre.dci) / dr; // primal_solvers.py:236-244
            if (kp == k) dv = cond ? ei : ((i == r) ? 1.0 : 0.0);
            else {
                const double prk = Tin[r * ld + kp];
                dv = (i == r) ? ei * prk : fma(ei, prk, Tin[i * ld + kp]);
            }
            xb = (i == r) ? ei * xr : fma(ei, xr, pre.xi); // :251-253
        }
        a.dn[i] = dv;
    }
    TfRatio t{in ? lpx_psd(xb, dv, f32) : LPX_INF, in ? i : LPX_NOIDX, dv, bsi, in ? dv : -LPX_INF};
    LPX_TF_STAMP(5);
    t = tf_wave_ratio(t);
    LPX_TF_STAMP(6);
    if (lane == 0) shr[wave] = t;
    __syncthreads();
    LPX_TF_STAMP(7);
    if (tid == 0) {
        t = shr[0];
#pragma unroll
        for (int w = 1; w < TF_THREADS / 64; ++w) t = tf_ratio_op(t, shr[w]);
        a.Rn.th[q] = t.th;
        a.Rn.row[q] = t.i;
        a.Rn.dmax[q] = t.dmax;
        a.Rn.dbest[q] = (t.i == LPX_NOIDX) ? 0.0 : t.dv;
        a.Rn.bvar[q] = (t.i == LPX_NOIDX) ? -1 : t.bv;
    }
}

__global__ __launch_bounds__(TF_THREADS) void k_t_boot(FusedArgs a)
{
    lpx_ctrl *ctrl = a.s.ctrl;
    const TfLaPre pre = tf_la_preload<true>(a, blockIdx.x);
    const int32_t status = ctrl->status, f32 = ctrl->f32_phase;
    if (blockIdx.x == 0 && threadIdx.x == 0) *a.base = ctrl->pivots; // also when the loop has ended: k_t_syncback reads it
    if (status != LPX_RUNNING) return;
    tf_lookahead<true>(a, blockIdx.x, pre, 0, 1.0, 0, -1, 0, f32 != 0);
}

__global__ __launch_bounds__(TF_THREADS) void k_t_fused(FusedArgs a)
{
    const lpx_tableau &s = a.s;
    lpx_ctrl *ctrl = s.ctrl;
    const int64_t m = s.m, ld = s.ld;
    const bool is_la = (int64_t)blockIdx.x < a.nla;
    const bool is_book = blockIdx.x == gridDim.x - 1; // the last workgroup keeps the books
#ifdef LPX_TF_STAMPS
    if (blockIdx.x == 0 && threadIdx.x == 0) g_tf_t0 = (long long)wall_clock64();
#endif
    // ---- round trip 1: everything whose address does not depend on this pivot's leaving row
    const int32_t status = ctrl->status;
    const int64_t counter0 = ctrl->counter, pivots0 = ctrl->pivots; // used by the bookkeeper only
    const double mn = a.nc[0], nj = a.nc[1], nk = a.nc[2], nopt = a.nc[3]; // entering decision of this pivot
    TfRatio t{LPX_INF, LPX_NOIDX, 0.0, -1, -LPX_INF};
    {   // ratio partials of this pivot: lane q of every wave loads partial q
        const int q = threadIdx.x & 63;
        if (q < a.nla) t = TfRatio{a.Rc.th[q], a.Rc.row[q], a.Rc.dbest[q], a.Rc.bvar[q], a.Rc.dmax[q]};
    }
    TfLaPre pre;
    d2_t x[TF_UN];
    double di[TF_UN];
    int64_t row0 = 0, row1 = 0, col = 0;
    bool live = false;
    if (is_la) {
        pre = tf_la_preload<false>(a, blockIdx.x);
    } else if (!is_book) {
        const uint32_t w = blockIdx.x - (uint32_t)a.nla; // 32-bit division: the grid has < 2^31 workgroups
        const int64_t band = w / (uint32_t)a.nstrips, sx = w - (uint32_t)band * (uint32_t)a.nstrips;
        row0 = band * a.rpb;
        row1 = (row0 + a.rpb < m + 1) ? row0 + a.rpb : m + 1;
        col = (sx * TF_THREADS + threadIdx.x) * 2;
        live = col <= s.n; // columns 0 .. n (column n is x_B); ld is even, so the pair stays inside the row
#pragma unroll
        for (int u = 0; u < TF_UN; ++u) {
            const int64_t i = row0 + u;
            if (live && i < row1) {
                x[u] = *(const d2_t *)(a.Tin + i * ld + col);
                di[u] = a.dc[i];
            }
        }
    }
#ifdef LPX_TF_STAMPS
    LPX_TF_STAMP(9);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    LPX_TF_STAMP(10);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    LPX_TF_STAMP(11);
#endif
    if (status != LPX_RUNNING) return;
    const bool lead = is_book && threadIdx.x == 0;
    const int64_t j = (int64_t)nj, k = (int64_t)nk;
    if (nopt != 0.0) { // primal_solvers.py:195-198
        if (lead) {
            ctrl->counter = counter0 + 1;
            ctrl->price_val = mn;
            ctrl->status = LPX_OPTIMAL;
        }
        return;
    }
    // leaving row: first minimum ratio over the lookahead workgroups' partials (ascending row ranges)
    t = tf_wave_ratio(t); // partial q covers rows [256 q, 256 q + 256): ascending with the lane
    const double th = t.th, dmax = t.dmax, dr = t.dv;
    const int64_t r = __builtin_amdgcn_readfirstlane((int)t.i), leaving = t.bv; // rows < 2^31 (m + 1 <= 16384)
    if (dmax <= 0.0) { // primal_solvers.py:144-147
        if (lead) {
            ctrl->counter = counter0 + 1;
            ctrl->price_val = mn;
            ctrl->enter_j = j;
            ctrl->aux_val = dmax;
            ctrl->status = LPX_PRIMAL_UNBOUNDED;
        }
        return;
    }
    if (is_book) {
        // bookkeeping of this pivot, by a workgroup of its own (in a lookahead workgroup its dependent loads and stores
        // would sit on the critical path of the launch): nothing below is read by any workgroup of this launch
        if (threadIdx.x == 0) {
            ctrl->counter = counter0 + 1;
            ctrl->price_val = mn;
            ctrl->enter_j = j;
            s.basis[r] = j; // primal_solvers.py:261
            if (s.slot_var) {
                s.slot_var[k] = leaving;
                s.var_slot[leaving] = k;
                s.var_slot[j] = -1;
            } else {
                const int64_t ll = leaving - s.col0;
                if (ll >= 0 && ll < s.n) s.var_state[ll] = LPX_VS_LOWER;
                s.var_state[k] = LPX_VS_BASIC;
            }
            ctrl->enter_slot = k;
            if (s.log && pivots0 < s.log_cap) {
                s.log[3 * pivots0 + 0] = r;
                s.log[3 * pivots0 + 1] = j;
                s.log[3 * pivots0 + 2] = leaving;
            }
            ctrl->pivots = pivots0 + 1;
            ctrl->leave_r = r;
            ctrl->leave_var = leaving;
            ctrl->ratio_val = th;
            ctrl->aux_val = dmax;
            ctrl->flip = 0;
            ctrl->f32_phase = 0;
        }
        return;
    }
    if (is_la) {
        tf_lookahead<false>(a, blockIdx.x, pre, r, dr, j, k, leaving, false);
#ifdef LPX_TF_STAMPS
        LPX_TF_STAMP(8);
        if (blockIdx.x == 0 && threadIdx.x == 0 && ctrl->pivots >= 40 && ctrl->pivots < 46)
            printf("tf stamps (10 ns): issued %lld scalar-done %lld vector-done %lld pre-r %lld | row-r loads+scan %lld | price dpp %lld | lds+decide %lld | col loads %lld | ratio dpp %lld | lds %lld | end %lld\n",
                   g_tf_stamp[9], g_tf_stamp[10], g_tf_stamp[11], g_tf_stamp[1], g_tf_stamp[2], g_tf_stamp[3], g_tf_stamp[4], g_tf_stamp[5], g_tf_stamp[6], g_tf_stamp[7], g_tf_stamp[8]);
#endif
        return;
    }
    // ---- update workgroup: rows [row0, row1) x one strip of 2 * TF_THREADS columns, Tin -> Tout
    if (!live) return;
    const bool cond = s.slot_var != nullptr;
    const d2_t pr = *(const d2_t *)(a.Tin + r * ld + col); // round trip 2: the old pivot row
    const bool fix0 = col == k, fix1 = col + 1 == k;
    for (int64_t i0 = row0; i0 < row1; i0 += TF_UN) { // TF_UN rows in flight
        if (i0 != row0) {
#pragma unroll
            for (int u = 0; u < TF_UN; ++u) {
                const int64_t i = i0 + u;
                if (i < row1) {
                    x[u] = *(const d2_t *)(a.Tin + i * ld + col);
                    di[u] = a.dc[i];
                }
            }
        }
#pragma unroll
        for (int u = 0; u < TF_UN; ++u) {
            const int64_t i = i0 + u;
            if (i >= row1) break;
            const bool isr = i == r;
            const double ei = isr ? 1.0 / dr : (-di[u]) / dr; // :236-244, row m: eliminates rc_j
            if (isr) {
                x[u][0] = ei * pr[0];
                x[u][1] = ei * pr[1];
            } else {
                x[u][0] = fma(ei, pr[0], x[u][0]);
                x[u][1] = fma(ei, pr[1], x[u][1]);
            }
            if (fix0) x[u][0] = cond ? ei : (isr ? 1.0 : 0.0);
            if (fix1) x[u][1] = cond ? ei : (isr ? 1.0 : 0.0);
            *(d2_t *)(a.Tout + i * ld + col) = x[u];
        }
    }
}

// end of a call: the state belongs in T
__global__ __launch_bounds__(LPX_BLOCK) void k_t_syncback(lpx_tableau s, const int64_t *base, const double *Talt, int64_t count)
{
    if (((s.ctrl->pivots - *base) & 1) == 0) return;
    for (int64_t t = ((int64_t)blockIdx.x * LPX_BLOCK + threadIdx.x) * 2; t < count; t += (int64_t)gridDim.x * LPX_BLOCK * 2)
        *(d2_t *)(s.T + t) = *(const d2_t *)(Talt + t);
}

// ---------------------------------------------------------------------------------------------------------
// The single-launch chain for the DUAL rules (dual_solvers.py:47-108): the same launch structure -- lookahead
// workgroups, update workgroups, one bookkeeper, two buffers, records alternating with the parity of the pivot -- with
// the dual's order of decisions.  While pivot t's update streams Tin -> Tout, every lookahead workgroup forms, from Tin,
// the pending eta column and the old pivot row (K1's expressions, bit for bit what the update writes):
//   1. the x_B column after the update -> leaving row r' = first x_B < 0, optimality test (whole column, redundantly);
//   2. row r' and the reduced-cost row after the update -> theta_j = (-rc_j) / row_j over row_j < 0, first argmin by
//      variable id -> entering j' (whole rows, redundantly), min(row) for the unboundedness test;
//   3. its own 256 rows of column j' after the update -> the next entering column (dn), the pivot element d_r'.
// Four dependent round trips (decision record -> x_B[r] -> row r' -> column j').  The decision record of a pivot:
enum { TD_MIN = 0, TD_R, TD_OPT, TD_TH, TD_J, TD_K, TD_ROWMIN, TD_DR, TD_LEAVING, TD_LEN };
// ---------------------------------------------------------------------------------------------------------
struct DualFusedArgs {
    lpx_tableau s;
    const double *Tin;
    double *Tout;
    const double *dc; // entering column of this pivot, m + 1 entries (dc[m] = its reduced cost)
    double *dn;
    const double *nc; // decision record of this pivot (TD_*)
    double *nn;
    int64_t *base;
    int64_t nla, nstrips, rpb;
};
// first batch of the loads whose addresses do not depend on this pivot's record
struct TdLaPre {
    double xi[TF_UN], dci[TF_UN]; // x_B column and pending entering column, rows tid + u * TF_THREADS
    double rcm[TF_UN];            // reduced-cost row, columns tid + u * TF_THREADS
    int64_t var[TF_UN];
    double dcm, dc_own;           // dc[m]; dc of this thread's own row (column gather)
};
template <bool BOOT>
__device__ __forceinline__ TdLaPre td_la_preload(const DualFusedArgs &a, int64_t q)
{
    const lpx_tableau &s = a.s;
    const int64_t m = s.m, n = s.n, ld = s.ld;
    TdLaPre p;
#pragma unroll
    for (int u = 0; u < TF_UN; ++u) {
        const int64_t i = threadIdx.x + (int64_t)u * TF_THREADS;
        p.xi[u] = (i < m) ? a.Tin[i * ld + n] : 0.0;
        p.dci[u] = (!BOOT && i < m) ? a.dc[i] : 0.0;
        p.rcm[u] = (i < n) ? a.Tin[m * ld + i] : 0.0;
        p.var[u] = (i >= n) ? LPX_NOIDX : (s.slot_var ? s.slot_var[i] : s.col0 + i);
    }
    const int64_t own = q * TF_THREADS + threadIdx.x;
    p.dcm = BOOT ? 0.0 : a.dc[m];
    p.dc_own = (!BOOT && own <= m) ? a.dc[own] : 0.0;
    return p;
}
template <bool BOOT>
__device__ __forceinline__ void td_lookahead(const DualFusedArgs &a, int64_t q, const TdLaPre &pre, int64_t r, double dr,
                                             int64_t j, int64_t k, int64_t leaving, bool f32)
{
    __shared__ BlockRed red;
    __shared__ int64_t sslot;
    const lpx_tableau &s = a.s;
    const int64_t m = s.m, n = s.n, ld = s.ld;
    const bool cond = s.slot_var != nullptr;
    const int tid = threadIdx.x;
    const double *Tin = a.Tin;
    // ---- 1. the x_B column as the update leaves it (primal_solvers.py:251-253), leaving row (dual_solvers.py:47-54)
    const double xr = BOOT ? 0.0 : Tin[r * ld + n];
    double mn = LPX_INF;
    int64_t first = LPX_NOIDX;
    for (int64_t i0 = tid; i0 < m; i0 += TF_THREADS * TF_UN) {
        double xi[TF_UN], dci[TF_UN];
#pragma unroll
        for (int u = 0; u < TF_UN; ++u) {
            const int64_t i = i0 + (int64_t)u * TF_THREADS;
            if (i0 == tid) {
                xi[u] = pre.xi[u];
                dci[u] = pre.dci[u];
            } else {
                xi[u] = (i < m) ? Tin[i * ld + n] : 0.0;
                dci[u] = (!BOOT && i < m) ? a.dc[i] : 0.0;
            }
        }
#pragma unroll
        for (int u = 0; u < TF_UN; ++u) {
            const int64_t i = i0 + (int64_t)u * TF_THREADS;
            if (i >= m) continue;
            double xb = xi[u];
            if (!BOOT) {
                const double ei = (i == r) ? 1.0 / dr : (-dci[u]) / dr;
                xb = (i == r) ? ei * xr : fma(ei, xr, xi[u]);
            }
            mn = fmin(mn, xb);
            if (xb < 0.0 && i < first) first = i;
        }
    }
    mn = block_min(mn, red);
    const int64_t r2 = block_imin(first, red);
    const bool opt = lpx_is_optimal_min(mn);
    if (opt) {
        if (q == 0 && tid == 0) {
            a.nn[TD_MIN] = mn;
            a.nn[TD_OPT] = 1.0;
            a.nn[TD_R] = 0.0;
            a.nn[TD_ROWMIN] = -1.0;
        }
        return;
    }
    // ---- 2. row r2 and the reduced-cost row as the update leaves them; ratio test over the columns (:61-80)
    const double em = BOOT ? 0.0 : (-pre.dcm) / dr;
    const double er2 = BOOT ? 0.0 : ((r2 == r) ? 1.0 / dr : (-a.dc[r2]) / dr);
    const int64_t bs2 = (!BOOT && r2 == r) ? j : s.basis[r2]; // the basic variable of row r2 as this pivot leaves it
    VI best{LPX_INF, LPX_NOIDX};
    int64_t bslot = -1;
    double rowmin = LPX_INF;
    for (int64_t c0 = tid; c0 < n; c0 += TF_THREADS * TF_UN) {
        double pr[TF_UN], rcm[TF_UN], rw[TF_UN];
        int64_t var[TF_UN];
        int32_t vst[TF_UN];
#pragma unroll
        for (int u = 0; u < TF_UN; ++u) {
            const int64_t c = c0 + (int64_t)u * TF_THREADS;
            const bool ok = c < n;
            pr[u] = (ok && !BOOT) ? Tin[r * ld + c] : 0.0;
            rw[u] = ok ? Tin[r2 * ld + c] : 0.0;
            vst[u] = (ok && !cond) ? s.var_state[c] : LPX_VS_LOWER;
            if (c0 == tid) {
                rcm[u] = pre.rcm[u];
                var[u] = pre.var[u];
            } else {
                rcm[u] = ok ? Tin[m * ld + c] : 0.0;
                var[u] = !ok ? LPX_NOIDX : (cond ? s.slot_var[c] : s.col0 + c);
            }
        }
#pragma unroll
        for (int u = 0; u < TF_UN; ++u) {
            const int64_t c = c0 + (int64_t)u * TF_THREADS;
            if (c >= n) continue;
            double row = rw[u], rc = rcm[u];
            int64_t v = var[u];
            bool basic = !cond && vst[u] == LPX_VS_BASIC;
            if (!BOOT) {
                if (c == k) { // the column the update rewrites as a whole: eta column (condensed) / unit vector (full)
                    row = cond ? er2 : ((r2 == r) ? 1.0 : 0.0);
                    rc = cond ? em : 0.0;
                    v = cond ? leaving : v;
                    basic = !cond;
                } else {
                    row = (r2 == r) ? er2 * pr[u] : fma(er2, pr[u], rw[u]);
                    rc = fma(em, pr[u], rcm[u]);
                    if (!cond && v == leaving) basic = false; // leaves the basis with this pivot
                }
            }
            if (basic) row = 0.0; // dual_solvers.py:38-39
            const double th = lpx_dsd(rc, row, f32);
            if (vi_less(th, v, best.v, best.i)) { best.v = th; best.i = v; bslot = c; }
            rowmin = fmin(rowmin, row);
        }
    }
    const VI win = block_argmin(best, red);
    rowmin = block_min(rowmin, red);
    if (best.i == win.i && best.v == win.v && bslot >= 0) sslot = bslot; // variable ids are distinct: one thread
    __syncthreads();
    const int64_t kp = sslot;
    if (q == 0 && tid == 0) {
        a.nn[TD_MIN] = mn;
        a.nn[TD_R] = (double)r2;
        a.nn[TD_OPT] = 0.0;
        a.nn[TD_TH] = win.v;
        a.nn[TD_J] = (double)win.i; // exact: ids < 2^53
        a.nn[TD_K] = (double)kp;
        a.nn[TD_ROWMIN] = rowmin;
        a.nn[TD_LEAVING] = (double)bs2;
    }
    if (rowmin >= 0.0) return; // dual unbounded (:41-44, :56-59): the next launch stops before it needs a column
    // ---- 3. this workgroup's rows of the next entering column as the update leaves it
    const int64_t i = q * TF_THREADS + tid;
    if (i <= m) {
        double dv;
        if (BOOT) {
            dv = Tin[i * ld + kp];
        } else {
            const double ei = (i == r) ? 1.0 / dr : (-pre.dc_own) / dr; // primal_solvers.py:236-244
            if (kp == k) dv = cond ? ei : ((i == r) ? 1.0 : 0.0);
            else {
                const double prk = Tin[r * ld + kp];
                dv = (i == r) ? ei * prk : fma(ei, prk, Tin[i * ld + kp]);
            }
        }
        a.dn[i] = dv;
        if (i == r2) a.nn[TD_DR] = dv;
    }
}

__global__ __launch_bounds__(TF_THREADS) void k_td_boot(DualFusedArgs a)
{
    lpx_ctrl *ctrl = a.s.ctrl;
    const TdLaPre pre = td_la_preload<true>(a, blockIdx.x);
    const int32_t status = ctrl->status, f32 = ctrl->f32_phase;
    if (blockIdx.x == 0 && threadIdx.x == 0) *a.base = ctrl->pivots; // also when the loop has ended: k_t_syncback reads it
    if (status != LPX_RUNNING) return;
    td_lookahead<true>(a, blockIdx.x, pre, 0, 1.0, 0, -1, 0, f32 != 0);
}

__global__ __launch_bounds__(TF_THREADS) void k_td_fused(DualFusedArgs a)
{
    const lpx_tableau &s = a.s;
    lpx_ctrl *ctrl = s.ctrl;
    const int64_t m = s.m, ld = s.ld;
    const bool is_la = (int64_t)blockIdx.x < a.nla;
    const bool is_book = blockIdx.x == gridDim.x - 1; // the last workgroup keeps the books
    // ---- round trip 1: everything whose address does not depend on this pivot's decisions
    const int32_t status = ctrl->status;
    const int64_t counter0 = ctrl->counter, pivots0 = ctrl->pivots; // used by the bookkeeper only
    double rec[TD_LEN];
#pragma unroll
    for (int u = 0; u < TD_LEN; ++u) rec[u] = a.nc[u];
    TdLaPre pre;
    d2_t x[TF_UN];
    double di[TF_UN];
    int64_t row0 = 0, row1 = 0, col = 0;
    bool live = false;
    if (is_la) {
        pre = td_la_preload<false>(a, blockIdx.x);
    } else if (!is_book) {
        const uint32_t w = blockIdx.x - (uint32_t)a.nla; // 32-bit division: the grid has < 2^31 workgroups
        const int64_t band = w / (uint32_t)a.nstrips, sx = w - (uint32_t)band * (uint32_t)a.nstrips;
        row0 = band * a.rpb;
        row1 = (row0 + a.rpb < m + 1) ? row0 + a.rpb : m + 1;
        col = (sx * TF_THREADS + threadIdx.x) * 2;
        live = col <= s.n; // columns 0 .. n (column n is x_B); ld is even, so the pair stays inside the row
#pragma unroll
        for (int u = 0; u < TF_UN; ++u) {
            const int64_t i = row0 + u;
            if (live && i < row1) {
                x[u] = *(const d2_t *)(a.Tin + i * ld + col);
                di[u] = a.dc[i];
            }
        }
    }
    if (status != LPX_RUNNING) return;
    const bool lead = is_book && threadIdx.x == 0;
    const double mn = rec[TD_MIN];
    if (rec[TD_OPT] != 0.0) { // dual_solvers.py:52-54
        if (lead) {
            ctrl->counter = counter0 + 1;
            ctrl->price_val = mn;
            ctrl->status = LPX_OPTIMAL;
        }
        return;
    }
    const int64_t r = (int64_t)rec[TD_R];
    if (rec[TD_ROWMIN] >= 0.0) { // :41-44, :56-59
        if (lead) {
            ctrl->counter = counter0 + 1;
            ctrl->price_val = mn;
            ctrl->leave_r = r;
            ctrl->aux_val = rec[TD_ROWMIN];
            ctrl->status = LPX_DUAL_UNBOUNDED;
        }
        return;
    }
    const int64_t j = (int64_t)rec[TD_J], k = (int64_t)rec[TD_K], leaving = (int64_t)rec[TD_LEAVING];
    const double dr = rec[TD_DR];
    if (is_book) {
        if (threadIdx.x == 0) { // k_td_ratio's and td_finish's bookkeeping; nothing below is read by this launch
            ctrl->counter = counter0 + 1;
            ctrl->price_val = mn;
            s.basis[r] = j;
            if (s.slot_var) {
                s.slot_var[k] = leaving;
                s.var_slot[leaving] = k;
                s.var_slot[j] = -1;
            } else {
                const int64_t ll = leaving - s.col0;
                if (ll >= 0 && ll < s.n) s.var_state[ll] = LPX_VS_LOWER;
                s.var_state[k] = LPX_VS_BASIC;
            }
            ctrl->enter_slot = k;
            if (s.log && pivots0 < s.log_cap) {
                s.log[3 * pivots0 + 0] = r;
                s.log[3 * pivots0 + 1] = j;
                s.log[3 * pivots0 + 2] = leaving;
            }
            ctrl->pivots = pivots0 + 1;
            ctrl->leave_r = r;
            ctrl->enter_j = j;
            ctrl->leave_var = leaving;
            ctrl->ratio_val = rec[TD_TH];
            ctrl->aux_val = rec[TD_ROWMIN];
            ctrl->flip = 0;
            ctrl->f32_phase = 0;
        }
        return;
    }
    if (is_la) {
        td_lookahead<false>(a, blockIdx.x, pre, r, dr, j, k, leaving, false);
        return;
    }
    // ---- update workgroup: rows [row0, row1) x one strip of 2 * TF_THREADS columns, Tin -> Tout (as k_t_fused)
    if (!live) return;
    const bool cond = s.slot_var != nullptr;
    const d2_t pr = *(const d2_t *)(a.Tin + r * ld + col); // the old pivot row
    const bool fix0 = col == k, fix1 = col + 1 == k;
    for (int64_t i0 = row0; i0 < row1; i0 += TF_UN) { // TF_UN rows in flight
        if (i0 != row0) {
#pragma unroll
            for (int u = 0; u < TF_UN; ++u) {
                const int64_t i = i0 + u;
                if (i < row1) {
                    x[u] = *(const d2_t *)(a.Tin + i * ld + col);
                    di[u] = a.dc[i];
                }
            }
        }
#pragma unroll
        for (int u = 0; u < TF_UN; ++u) {
            const int64_t i = i0 + u;
            if (i >= row1) break;
            const bool isr = i == r;
            const double ei = isr ? 1.0 / dr : (-di[u]) / dr; // :236-244, row m: eliminates rc_j
            if (isr) {
                x[u][0] = ei * pr[0];
                x[u][1] = ei * pr[1];
            } else {
                x[u][0] = fma(ei, pr[0], x[u][0]);
                x[u][1] = fma(ei, pr[1], x[u][1]);
            }
            if (fix0) x[u][0] = cond ? ei : (isr ? 1.0 : 0.0);
            if (fix1) x[u][1] = cond ? ei : (isr ? 1.0 : 0.0);
            *(d2_t *)(a.Tout + i * ld + col) = x[u];
        }
    }
}

static inline bool t_use_fused(const lpx_tableau *s)
{
    if (s->T_alt == nullptr || lpx_env_int("LPX_TABLEAU_FUSED", 1) == 0) return false;
    return lpx_cdiv(s->m + 1, TF_THREADS) <= TF_MAXLA && s->n + 1 <= 32768;
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


static int tableau_iterate_fused(const lpx_tableau *s, const TabWs &w, int64_t iters, hipStream_t st)
{
    LPX_REQUIRE(lpx_aligned16(s->T_alt), "T_alt must be 16-byte aligned");
    FusedArgs a{};
    a.s = *s;
    a.base = w.base;
    a.nla = lpx_cdiv(s->m + 1, TF_THREADS);
    a.nstrips = lpx_cdiv(s->n + 1, 2 * TF_THREADS);
    // tall update tiles amortise the per-workgroup prologue (kernel arguments, partials, pivot-row strip) as long as
    // about one workgroup per CU remains: 1000 x 1000 on MI355X, rows per workgroup 1 / 4 / 8 / 16 / 32 -> 144 / 147 /
    // 148.5 / 145 / 102 k pivots/s (tools/fused_sweep.sh)
    int64_t rpb = lpx_env_int("LPX_FUSED_ROWS", 0);
    if (rpb <= 0) {
        rpb = 1;
        while (rpb < 8 && a.nstrips * lpx_cdiv(s->m + 1, 2 * rpb) >= 240) rpb *= 2;
    }
    a.rpb = rpb;
    const int64_t nupd = a.nstrips * lpx_cdiv(s->m + 1, rpb);
    LPX_REQUIRE(a.nla + nupd + 1 < ((int64_t)1 << 31), "grid too large");
    auto args_for = [&](int par) {
        FusedArgs b = a;
        b.Tin = par ? s->T_alt : s->T;
        b.Tout = par ? s->T : s->T_alt;
        b.Rc = w.fp[par]; b.Rn = w.fp[par ^ 1];
        b.dc = w.d2[par]; b.dn = w.d2[par ^ 1];
        b.nc = w.next[par]; b.nn = w.next[par ^ 1];
        return b;
    };
    { // bootstrap: price the carried row, first entering column and ratio partials -> the records of parity 0
        FusedArgs b = args_for(1); // writes the "next" side of parity 1 = parity 0's records ...
        b.Tin = s->T;              // ... from the state in T
        b.Tout = nullptr;
        hipLaunchKernelGGL(k_t_boot, dim3((unsigned)a.nla), dim3(TF_THREADS), 0, st, b);
    }
    int64_t it = 0; // iteration index inside this call: parity of the launch
    auto one = [&](hipStream_t q) {
        const FusedArgs b = args_for((int)(it & 1));
        lpx_k1_begin(q);
        hipLaunchKernelGGL(k_t_fused, dim3((unsigned)(a.nla + nupd + 1)), dim3(TF_THREADS), 0, q, b);
        lpx_k1_end(q);
        it += 1;
    };
    struct {
        lpx_tableau s;
        int64_t rpb;
        int what;
    } key;
    memset(&key, 0, sizeof key);
    key.s = *s; key.rpb = rpb; key.what = 3;
    const int rc = lpx_run_iterations(&key, sizeof key, iters, st, one);
    const int64_t count = (s->m + 1) * s->ld;
    int64_t nb = lpx_cdiv(count / 2, LPX_BLOCK);
    if (nb > 4096) nb = 4096;
    hipLaunchKernelGGL(k_t_syncback, dim3((unsigned)nb), dim3(LPX_BLOCK), 0, st, *s, (const int64_t *)w.base,
                       (const double *)s->T_alt, count);
    if (rc) return rc;
    LPX_LAUNCH_CHECK();
    return LPX_OK;
}

#include "lpx_tableau_blocked.hpp"

// block size of the deferred-update chain for this tableau, 0 = not used.  The chain pays two small launches per pivot
// and saves (KB - 1) / KB of the update's traffic: it wins once the tableau is too large to stream in a few microseconds.
static inline int t_block_size(const lpx_tableau *s)
{
    // measured on MI355X (tools/block_crossover.sh, k pivots/s, one update per pivot / blocks of 16 / of 32): 2000 x 4000
    // 80 / 62 / 60; 3000 x 6000 41 / 57 / 57; 5000 x 10000 13 / 50 / 52; 8000 x 16000 5.7 / 38 / 43; 20000 x 40000
    // 0.96 / - / 17.8: the chain pays ~15 us of decision kernels per pivot whatever the size
    const int64_t bytes = 8 * (s->m + 1) * (s->n + 1);
    if (bytes < ((int64_t)lpx_env_int("LPX_TABLEAU_BLOCK_MIN_MB", 48) << 20)) return 0;
    // one partial record per workgroup of TB_THREADS rows / columns
    if (lpx_cdiv(s->n, TB_THREADS) > LPX_MAXP || lpx_cdiv(s->m + 2, TB_THREADS) > LPX_MAXP) return 0;
    const int kb = lpx_env_int("LPX_TABLEAU_BLOCK", bytes >= ((int64_t)192 << 20) ? 32 : 16);
    return (kb == 8 || kb == 16 || kb == 32) ? kb : 0;
}

extern "C" int64_t lpx_tableau_block_size(const lpx_tableau *s)
{
    if (s == nullptr || s->m <= 0 || s->n <= 0) return 0;
    return t_block_size(s);
}

static TbArgs tb_args(const lpx_tableau *s, const TabWs &w)
{
    TbArgs a{};
    a.s = *s;
    a.b = w.tb;
    a.price = w.price;
    a.ratio = w.ratio;
    a.d = w.d;
    a.nprice = lpx_cdiv(s->n, TB_THREADS);
    a.nratio = lpx_cdiv(s->m + 1, TB_THREADS);
    a.u = 0;
    return a;
}
static void tb_boot(const lpx_tableau *s, const TbArgs &a, hipStream_t st)
{
    const int64_t nboot = lpx_cdiv(s->n > s->m ? s->n : s->m, TB_THREADS);
    hipLaunchKernelGGL(k_tb_boot, dim3((unsigned)nboot), dim3(TB_THREADS), 0, st, a);
}
// the pass over the tableau: strips of 128 columns x bands of rows; a workgroup (4 waves) walks down its band 32 rows at a
// time, so bands are whole multiples of 32 rows, and there are enough of them for ~5000 workgroups (measured on the
// 20001 x 20001 tableau, tools/tb_flush_bench.hip: bands of 2501 / 1251 / 640 / 320 rows -> 1.21 / 1.18 / 1.17 / 1.17 ms)
static void tb_flush_from(const lpx_tableau *s, const TbWs &h, const double *src, double *dst, int KB, hipStream_t q)
{
    TbFlushArgs f{};
    f.T = src; f.Tdst = dst; f.rows = s->m + 1; f.cols = s->n + 1; f.ld = s->ld; f.cond = s->slot_var != nullptr;
    f.E = h.E; f.R = h.R; f.ldr = h.ldr; f.pslot = h.pslot; f.rowflag = h.rowflag; f.cnt = h.cnt;
    const int64_t nstrips = lpx_cdiv(f.cols, TB2_COLS);
    const int nt = lpx_env_int("LPX_RANK1_NT", (8 * f.rows * f.cols > ((int64_t)192 << 20)) ? 1 : 0) ? 1 : 0;
    int64_t rpb = lpx_env_int("LPX_TB_ROWS", 0);
    if (rpb <= 0) {
        // workgroups to aim for: the condensed layout's blocks of 32 run 4 workgroups per CU (lpx_tb_flush32.hip), the rest 2.
        // 10240 (bands of ~320 rows at 20001 x 20001): as fast as 20480 (profiles/r03_pass_wgs_sweep.log) with half the
        // refills of the pivot-row strips; fewer, taller workgroups lose
        int64_t nbands = lpx_cdiv(lpx_env_int("LPX_TB_WGS", (KB == 32 && nt && f.cond) ? 10240 : 5120), nstrips);
        if (nbands < 1) nbands = 1;
        rpb = lpx_roundup(lpx_cdiv(f.rows, nbands), 32);
    }
    f.rpb = rpb;
    lpx_k1_begin(q);
    lpx_tb_flush_launch(f, KB, nt, q);
    lpx_k1_end(q);
}
static void tb_flush(const lpx_tableau *s, const TabWs &w, int KB, hipStream_t q) { tb_flush_from(s, w.tb, s->T, s->T, KB, q); }

// one column-partitioned pivot of the deferred-update chains: propose -> all-gather -> select + prepare (a.b / a.o as in
// the unpartitioned chains; the exchange is the only step that leaves the rank)
template <bool OLD>
static int tb_dist_pivot(const TbArgs &a, int u, const TbDist &d, int64_t nprep, hipStream_t q)
{
    TbArgs b = a;
    b.u = u;
    b.sendbuf = d.send;
    hipLaunchKernelGGL((k_tb_column<TB_PROPOSE, OLD>), dim3((unsigned)a.nratio), dim3(TB_THREADS), 0, q, b);
    LPX_NCCL(ncclAllGather(d.send, d.recv, (size_t)d.msg, ncclDouble, d.comm, q));
    b.recv = d.recv;
    b.nranks = d.nranks;
    b.msg = d.msg;
    hipLaunchKernelGGL((k_tb_column<TB_SELECT>), dim3((unsigned)a.nratio), dim3(TB_THREADS), 0, q, b);
    hipLaunchKernelGGL((k_tb_prepare<OLD>), dim3((unsigned)nprep), dim3(TB_THREADS), 0, q, b);
    return LPX_OK;
}

static int tableau_iterate_blocked(const lpx_tableau *s, const TabWs &w, int64_t iters, hipStream_t st, int KB, const TbDist *dist = nullptr)
{
    const TbArgs a = tb_args(s, w);
    const int64_t nprep = lpx_cdiv(t_prep_len(s), TB_THREADS);
    auto flush = [&](hipStream_t q) { tb_flush(s, w, KB, q); };
    tb_boot(s, a, st);
    if (dist != nullptr) { // column-partitioned: eager launches around the collective
        for (int64_t it = 0; it < iters; ++it) {
            const int rc = tb_dist_pivot<false>(a, (int)(it % KB), *dist, nprep, st);
            if (rc) return rc;
            if (it % KB == KB - 1 || it + 1 == iters) flush(st);
        }
        LPX_LAUNCH_CHECK();
        return LPX_OK;
    }
    int64_t it = 0;
    auto one = [&](hipStream_t q) {
        TbArgs b = a;
        b.u = (int)(it % KB);
        hipLaunchKernelGGL((k_tb_column<TB_LOCAL>), dim3((unsigned)a.nratio), dim3(TB_THREADS), 0, q, b);
        hipLaunchKernelGGL((k_tb_prepare<false>), dim3((unsigned)nprep), dim3(TB_THREADS), 0, q, b);
        it += 1;
    };
    struct {
        lpx_tableau s;
        int kb, what;
    } key;
    memset(&key, 0, sizeof key);
    key.s = *s; key.kb = KB; key.what = 5;
    // a block = KB pivots' decisions (replayed as one graph) + one pass over the tableau
    while (iters > 0) {
        const int64_t k = iters < KB ? iters : KB;
        it = 0;
        const int rc = lpx_run_iterations(&key, sizeof key, k, st, one, KB, true);
        if (rc) return rc;
        flush(st);
        iters -= k;
    }
    LPX_LAUNCH_CHECK();
    return LPX_OK;
}

// ---- pipelined chain: the pass of block k streams T_old -> T_new on the caller's stream while block k+1 is decided on a
// high-priority side stream from T_old, block k's pairs and its own (lpx_tableau_blocked.hpp).  Dependencies, all by
// events, none through the host:  D(k+1) after D(k) [same stream] and F(k-1) [T_old complete, its pair arrays free];
// F(k) after D(k) and F(k-1) [same stream].  Buffers alternate T, T_alt, T, ...; the LAST pass of a call writes T whatever
// it reads (nothing is decided beside it, so it may run in place), and a closing kernel copies T_alt back in the one case
// the host cannot see from here: a loop that ended inside an earlier block (TbWs::tsel).  T is current when the call's
// work has drained, as for every other chain.
struct TbPipe {
    int dev;
    hipStream_t side;  // decisions
    hipStream_t pass;  // the pass, when the two run on disjoint sets of CUs (else the caller's stream carries it)
    hipEvent_t fork, join, evD[4], evF[4];
};
static std::mutex g_tb_pipe_mu;
static TbPipe g_tb_pipes[64];
static int g_tb_npipes = 0;
// The pass keeps every wave slot of the chip busy for a millisecond; a decision workgroup needs four free slots and
// 128+ free registers on each SIMD of ONE CU at the same moment, which a retiring pass workgroup never leaves (its
// successor takes them first): beside the pass the decision kernels starve until it drains, whatever the stream
// priorities (rocprofv3 trace, profiles/r03_pipe_overlap_nomask.txt).  So the two get disjoint CUs: LPX_TB_PIPE_CUS (default
// 32 of 256) for the decisions -- the low bits of the mask, which the driver deals round-robin over the XCDs and their
// shader engines -- and the rest for the pass.  0: no masks, one high-priority side stream.
static int tb_pipe_get(TbPipe **out)
{
    int dev = 0;
    LPX_HIP(hipGetDevice(&dev));
    for (int k = 0; k < g_tb_npipes; ++k)
        if (g_tb_pipes[k].dev == dev) { *out = &g_tb_pipes[k]; return LPX_OK; }
    LPX_REQUIRE(g_tb_npipes < 64, "more than 64 devices");
    TbPipe p{};
    p.dev = dev;
    hipDeviceProp_t prop;
    LPX_HIP(hipGetDeviceProperties(&prop, dev));
    const int ncu = prop.multiProcessorCount;
    int res = lpx_env_int("LPX_TB_PIPE_CUS", 32);
    if (res < 0 || 2 * res > ncu || ncu > 1024) res = 0;
    if (res > 0) {
        uint32_t md[32] = {0}, mp[32] = {0};
        for (int c = 0; c < ncu; ++c) (c < res ? md : mp)[c / 32] |= 1u << (c % 32);
        const uint32_t words = (uint32_t)((ncu + 31) / 32);
        if (hipExtStreamCreateWithCUMask(&p.side, words, md) != hipSuccess || hipExtStreamCreateWithCUMask(&p.pass, words, mp) != hipSuccess) {
            (void)hipGetLastError();
            if (p.side) (void)hipStreamDestroy(p.side);
            p.side = p.pass = nullptr;
            res = 0;
        }
    }
    if (res == 0) {
        int lo = 0, hi = 0;
        LPX_HIP(hipDeviceGetStreamPriorityRange(&lo, &hi)); // hi = greatest priority (numerically lowest)
        LPX_HIP(hipStreamCreateWithPriority(&p.side, hipStreamNonBlocking, hi));
    }
    LPX_HIP(hipEventCreateWithFlags(&p.fork, hipEventDisableTiming));
    LPX_HIP(hipEventCreateWithFlags(&p.join, hipEventDisableTiming));
    for (int k = 0; k < 4; ++k) {
        LPX_HIP(hipEventCreateWithFlags(&p.evD[k], hipEventDisableTiming));
        LPX_HIP(hipEventCreateWithFlags(&p.evF[k], hipEventDisableTiming));
    }
    g_tb_pipes[g_tb_npipes] = p;
    *out = &g_tb_pipes[g_tb_npipes++];
    return LPX_OK;
}
__global__ __launch_bounds__(256) void k_tb_syncback(const int64_t *tsel, const double *src, double *dst, int64_t npairs)
{
    if (tsel[0] == 0) return;
    for (int64_t k = (int64_t)blockIdx.x * 256 + threadIdx.x; k < npairs; k += (int64_t)gridDim.x * 256)
        lpx_st2<1>(dst + 2 * k, lpx_ld2<1>(src + 2 * k));
}
// tableaus whose pass is long enough to hide a block's decisions behind it (they pay chains of up to 2 KB for it)
static inline bool t_use_pipeline(const lpx_tableau *s)
{
    if (s->T_alt == nullptr || s->T_alt == s->T) return false;
    if (!(s->slot_var != nullptr || (s->col0 == 0 && s->n_global == s->n))) return false;
    const int64_t bytes = 8 * (s->m + 1) * (s->n + 1);
    const int min_mb = lpx_env_int("LPX_TABLEAU_PIPELINE_MIN_MB", 1024); // negative: never
    return min_mb >= 0 && bytes >= ((int64_t)min_mb << 20);
}
extern "C" int lpx_tableau_pipelined(const lpx_tableau *s)
{
    if (s == nullptr || s->m <= 0 || s->n <= 0) return 0;
    return (t_block_size(s) > 0 && t_use_pipeline(s)) ? 1 : 0;
}
static int tableau_iterate_pipelined(const lpx_tableau *s, const TabWs &w, int64_t iters, hipStream_t st, int KB, const TbDist *dist = nullptr)
{
    LPX_REQUIRE(lpx_aligned16(s->T_alt), "T_alt must be 16-byte aligned");
    std::lock_guard<std::mutex> lock(g_tb_pipe_mu);
    TbPipe *pp = nullptr;
    int rc = tb_pipe_get(&pp);
    if (rc) return rc;
    hipStream_t sd = pp->side;
    hipStream_t sf = pp->pass ? pp->pass : st; // the stream of the passes
    const int64_t nprep = lpx_cdiv(t_prep_len(s), TB_THREADS);
    const int64_t nb = lpx_cdiv(iters, KB);
    double *buf[2] = {s->T, s->T_alt};
    const bool serial = lpx_env_int("LPX_TB_PIPE_SERIAL", 0) != 0; // diagnostic: same kernels and streams, nothing side by side
    LPX_HIP(hipEventRecord(pp->fork, st));
    LPX_HIP(hipStreamWaitEvent(sd, pp->fork, 0));
    if (sf != st) LPX_HIP(hipStreamWaitEvent(sf, pp->fork, 0));
    tb_boot(s, tb_args(s, w), sd);
    struct {
        lpx_tableau s;
        int kb, what;
    } key;
    for (int64_t k = 0; k < nb; ++k) {
        const int64_t cnt = (k + 1 < nb) ? KB : iters - k * KB;
        const int srcid = (int)(k & 1), dstid = (k + 1 == nb) ? 0 : (srcid ^ 1);
        TbArgs a = tb_args(s, w);
        a.b = (k & 1) ? w.tb2 : w.tb;
        a.o = (k & 1) ? w.tb : w.tb2;
        a.uo = k > 0 ? KB : 0;
        a.dstid = dstid;
        if (k > 0) a.s.T = buf[(k - 1) & 1]; // the source of the pass in flight
        if (k >= 2) LPX_HIP(hipStreamWaitEvent(sd, pp->evF[(k - 2) & 3], 0));
        if (k >= 1 && serial) LPX_HIP(hipStreamWaitEvent(sd, pp->evF[(k - 1) & 3], 0));
        int64_t it = 0;
        const bool old = k > 0;
        auto one = [&](hipStream_t q) {
            TbArgs b = a;
            b.u = (int)(it % KB);
            if (old) {
                hipLaunchKernelGGL((k_tb_column<TB_LOCAL, true>), dim3((unsigned)a.nratio), dim3(TB_THREADS), 0, q, b);
                hipLaunchKernelGGL((k_tb_prepare<true>), dim3((unsigned)nprep), dim3(TB_THREADS), 0, q, b);
            } else {
                hipLaunchKernelGGL((k_tb_column<TB_LOCAL>), dim3((unsigned)a.nratio), dim3(TB_THREADS), 0, q, b);
                hipLaunchKernelGGL((k_tb_prepare<false>), dim3((unsigned)nprep), dim3(TB_THREADS), 0, q, b);
            }
            it += 1;
        };
        memset(&key, 0, sizeof key);
        key.s = a.s; key.kb = KB;
        key.what = 16 + (old ? 1 : 0) + 2 * (int)(k & 1) + 4 * dstid;
        if (dist != nullptr) { // column-partitioned: the same block, each pivot around its all-gather (eager launches)
            for (int64_t u = 0; u < cnt && rc == LPX_OK; ++u)
                rc = old ? tb_dist_pivot<true>(a, (int)u, *dist, nprep, sd) : tb_dist_pivot<false>(a, (int)u, *dist, nprep, sd);
        } else {
            rc = lpx_run_iterations(&key, sizeof key, cnt, sd, one, KB, true);
        }
        if (rc) return rc;
        LPX_HIP(hipEventRecord(pp->evD[k & 3], sd));
        LPX_HIP(hipStreamWaitEvent(sf, pp->evD[k & 3], 0));
        tb_flush_from(s, a.b, buf[srcid], buf[dstid], KB, sf);
        LPX_HIP(hipEventRecord(pp->evF[k & 3], sf));
    }
    const int64_t npairs = (s->m + 1) * s->ld / 2;
    hipLaunchKernelGGL(k_tb_syncback, dim3(4096), dim3(256), 0, sf, (const int64_t *)w.tb.tsel, (const double *)s->T_alt, s->T, npairs);
    if (sf != st) {
        LPX_HIP(hipEventRecord(pp->join, sf));
        LPX_HIP(hipStreamWaitEvent(st, pp->join, 0));
    }
    LPX_LAUNCH_CHECK();
    return LPX_OK;
}

// ---- launch-bound tableaus (lpx_tableau_small.hpp): one launch per block of TS_KB pivots -- workgroup 0 decides block k
// while the other workgroups apply block k-1 (T_old -> T_new).  Buffers and the closing copy-back as in the pipelined chain.
static inline bool t_use_small(const lpx_tableau *s)
{
    if (s->T_alt == nullptr || s->T_alt == s->T || lpx_env_int("LPX_TABLEAU_SMALL_CHAIN", 1) == 0) return false;
    if (!(s->slot_var != nullptr || (s->col0 == 0 && s->n_global == s->n))) return false;
    return s->m + 1 <= TS_THREADS && lpx_roundup(s->n + 1, 2) <= TS_THREADS && s->ratio_rule == LPX_RATIO_TEXTBOOK;
}
static int tableau_iterate_small(const lpx_tableau *s, const TabWs &w, int64_t iters, hipStream_t st)
{
    LPX_REQUIRE(lpx_aligned16(s->T_alt), "T_alt must be 16-byte aligned");
    constexpr int KB = TS_KB;
    const int64_t nb = lpx_cdiv(iters, KB);
    double *buf[2] = {s->T, s->T_alt};
    tb_boot(s, tb_args(s, w), st);
    const int64_t rows = s->m + 1, cols = s->n + 1;
    const int64_t nstrips = lpx_cdiv(cols, TB2_COLS);
    const int64_t rpb = lpx_env_int("LPX_TS_ROWS", 128), nbands = lpx_cdiv(rows, rpb);
    for (int64_t L = 0; L <= nb; ++L) {
        TsArgs a{};
        a.nstrips = nstrips;
        a.decide = L < nb;
        a.pass = L >= 1;
        if (a.decide) {
            const int64_t k = L;
            a.s = *s;
            a.b = (k & 1) ? w.tb2 : w.tb;
            a.o = (k & 1) ? w.tb : w.tb2;
            a.uo = k > 0 ? KB : 0;
            a.u = (int)((k + 1 < nb) ? KB : iters - k * KB);
            a.dstid = (k + 1 == nb) ? 0 : (int)((k & 1) ^ 1);
            if (k > 0) a.s.T = buf[(k - 1) & 1]; // the source of the pass beside it
        }
        if (a.pass) {
            const int64_t k = L - 1;
            const TbWs &h = (k & 1) ? w.tb2 : w.tb;
            const int srcid = (int)(k & 1), dstid = (k + 1 == nb) ? 0 : (srcid ^ 1);
            TbFlushArgs &f = a.f;
            f.T = buf[srcid]; f.Tdst = buf[dstid]; f.rows = rows; f.cols = cols; f.ld = s->ld; f.cond = s->slot_var != nullptr;
            f.E = h.E; f.R = h.R; f.ldr = h.ldr; f.pslot = h.pslot; f.rowflag = h.rowflag; f.cnt = h.cnt; f.rpb = rpb;
            a.fEt = h.Et;
            a.flde = h.lde;
        }
        const unsigned grid = a.pass ? (unsigned)(1 + nstrips * nbands) : 1u;
        lpx_k1_begin(st);
        lpx_tb_small_launch(a, grid, st);
        lpx_k1_end(st);
    }
    const int64_t npairs = (s->m + 1) * s->ld / 2;
    hipLaunchKernelGGL(k_tb_syncback, dim3(256), dim3(256), 0, st, (const int64_t *)w.tb.tsel, (const double *)s->T_alt, s->T, npairs);
    LPX_LAUNCH_CHECK();
    return LPX_OK;
}

// ---- column-partitioned tableau with deferred updates: every rank defers the updates of its own slice.  Replicated:
// x_B (xb[]), the eta columns (formed from the gathered winner's column) and all decisions; local: the slices of the
// pivot rows (R), of the reduced-cost row (rc[]) and the pricing partials.  Per pivot: propose -> all-gather -> commit;
// per block (or at the end of the caller's loop): flush.
extern "C" int lpx_tableau_block_begin(const lpx_tableau *s, void *stream)
{
    int rc = check_tableau(s);
    if (rc) return rc;
    LPX_REQUIRE(s->ratio_rule == LPX_RATIO_TEXTBOOK, "the Harris-style ratio test is an option of lpx_tableau_iterate only");
    const TabWs w = tab_ws(s->m, s->n, s->workspace);
    tb_boot(s, tb_args(s, w), (hipStream_t)stream);
    LPX_LAUNCH_CHECK();
    return LPX_OK;
}
extern "C" int lpx_tableau_block_propose(const lpx_tableau *s, int64_t pending, double *sendbuf, void *stream)
{
    int rc = check_tableau(s);
    if (rc) return rc;
    LPX_REQUIRE(s->ratio_rule == LPX_RATIO_TEXTBOOK, "the Harris-style ratio test is an option of lpx_tableau_iterate only");
    LPX_REQUIRE(sendbuf != nullptr && pending >= 0 && pending < TB_KMAX, "bad send buffer / pending count");
    const TabWs w = tab_ws(s->m, s->n, s->workspace);
    TbArgs a = tb_args(s, w);
    a.u = (int)pending;
    a.sendbuf = sendbuf;
    hipLaunchKernelGGL((k_tb_column<TB_PROPOSE>), dim3((unsigned)a.nratio), dim3(TB_THREADS), 0, (hipStream_t)stream, a);
    LPX_LAUNCH_CHECK();
    return LPX_OK;
}
extern "C" int lpx_tableau_block_commit(const lpx_tableau *s, int64_t pending, const double *recvbuf, int64_t nranks, void *stream)
{
    int rc = check_tableau(s);
    if (rc) return rc;
    LPX_REQUIRE(s->ratio_rule == LPX_RATIO_TEXTBOOK, "the Harris-style ratio test is an option of lpx_tableau_iterate only");
    LPX_REQUIRE(recvbuf != nullptr && nranks >= 1 && nranks <= 64, "bad receive buffer / rank count");
    LPX_REQUIRE(pending >= 0 && pending < TB_KMAX, "bad pending count");
    const TabWs w = tab_ws(s->m, s->n, s->workspace);
    TbArgs a = tb_args(s, w);
    a.u = (int)pending;
    a.recv = recvbuf;
    a.nranks = nranks;
    a.msg = lpx_tableau_msg_len(s->m);
    hipStream_t st = (hipStream_t)stream;
    hipLaunchKernelGGL((k_tb_column<TB_SELECT>), dim3((unsigned)a.nratio), dim3(TB_THREADS), 0, st, a);
    hipLaunchKernelGGL((k_tb_prepare<false>), dim3((unsigned)lpx_cdiv(t_prep_len(s), TB_THREADS)), dim3(TB_THREADS), 0, st, a);
    LPX_LAUNCH_CHECK();
    return LPX_OK;
}
extern "C" int lpx_tableau_block_flush(const lpx_tableau *s, int64_t block, void *stream)
{
    int rc = check_tableau(s);
    if (rc) return rc;
    LPX_REQUIRE(s->ratio_rule == LPX_RATIO_TEXTBOOK, "the Harris-style ratio test is an option of lpx_tableau_iterate only");
    LPX_REQUIRE(block == 8 || block == 16 || block == 32, "block size must be 8, 16 or 32");
    const TabWs w = tab_ws(s->m, s->n, s->workspace);
    tb_flush(s, w, (int)block, (hipStream_t)stream);
    LPX_LAUNCH_CHECK();
    return LPX_OK;
}

// ---- the column-partitioned loop as ONE call: the library enqueues propose -> ncclAllGather -> commit for every pivot and
// the passes over the slice itself, nothing returns to the host in between (the Python loop around lpx_tableau_block_* pays
// three ctypes calls and a c10d collective per pivot: 10.6 k pivots/s through one rank against 19 k unpartitioned).
extern "C" int lpx_dist_unique_id(void *id128)
{
    LPX_REQUIRE(id128 != nullptr, "null id buffer");
    static_assert(sizeof(ncclUniqueId) == 128, "ncclUniqueId is 128 bytes");
    LPX_NCCL(ncclGetUniqueId((ncclUniqueId *)id128));
    return LPX_OK;
}
extern "C" int lpx_dist_comm_create(const void *id128, int64_t nranks, int64_t rank, void **comm)
{
    LPX_REQUIRE(id128 != nullptr && comm != nullptr && nranks >= 1 && nranks <= 64 && rank >= 0 && rank < nranks, "bad communicator arguments");
    ncclUniqueId id;
    memcpy(&id, id128, sizeof id);
    ncclComm_t c = nullptr;
    LPX_NCCL(ncclCommInitRank(&c, (int)nranks, id, (int)rank));
    *comm = (void *)c;
    return LPX_OK;
}
extern "C" int lpx_dist_comm_destroy(void *comm)
{
    if (comm == nullptr) return LPX_OK;
    LPX_NCCL(ncclCommDestroy((ncclComm_t)comm));
    return LPX_OK;
}
extern "C" int lpx_tableau_dist_iterate(const lpx_tableau *s, void *comm, int64_t nranks, int64_t iters, int64_t block, double *sendbuf,
                                        double *recvbuf, void *stream)
{
    int rc = check_tableau(s);
    if (rc) return rc;
    LPX_REQUIRE(s->ratio_rule == LPX_RATIO_TEXTBOOK, "the Harris-style ratio test is an option of lpx_tableau_iterate only");
    LPX_REQUIRE(comm != nullptr && sendbuf != nullptr && recvbuf != nullptr && nranks >= 1 && nranks <= 64, "bad communicator / buffers / rank count");
    LPX_REQUIRE(block == 8 || block == 16 || block == 32, "block size must be 8, 16 or 32 (the same on every rank)");
    LPX_REQUIRE(lpx_cdiv(s->n, TB_THREADS) <= LPX_MAXP && lpx_cdiv(s->m + 2, TB_THREADS) <= LPX_MAXP, "slice too large for the partial arrays");
    if (iters <= 0) return LPX_OK;
    const TabWs w = tab_ws(s->m, s->n, s->workspace);
    const TbDist d{(ncclComm_t)comm, nranks, lpx_tableau_msg_len(s->m), sendbuf, recvbuf};
    hipStream_t st = (hipStream_t)stream;
    if (t_use_pipeline(s) && iters > block) return tableau_iterate_pipelined(s, w, iters, st, (int)block, &d);
    return tableau_iterate_blocked(s, w, iters, st, (int)block, &d);
}

extern "C" int lpx_tableau_chain_kind(const lpx_tableau *s)
{
    if (s == nullptr || s->m <= 0 || s->n <= 0) return -1;
    if (s->ratio_rule == LPX_RATIO_HARRIS) return LPX_CHAIN_EAGER;
    if (t_block_size(s) > 0) return t_use_pipeline(s) ? LPX_CHAIN_PIPELINED : LPX_CHAIN_DEFERRED;
    if (t_use_small(s)) return LPX_CHAIN_SMALL;
    if (t_use_fused(s)) return LPX_CHAIN_FUSED;
    return LPX_CHAIN_EAGER;
}

extern "C" int lpx_tableau_iterate(const lpx_tableau *s, int64_t iters, void *stream)
{
    int rc = check_tableau(s);
    if (rc) return rc;
    LPX_REQUIRE(s->slot_var != nullptr || (s->col0 == 0 && s->n_global == s->n), "lpx_tableau_iterate needs an unpartitioned tableau");
    const TabWs w = tab_ws(s->m, s->n, s->workspace);
    hipStream_t st = (hipStream_t)stream;
    if (iters <= 0) return LPX_OK;
    // the opt-in Harris-style ratio test lives in the eager three-kernel chain only (one more small kernel per pivot)
    const bool harris = s->ratio_rule == LPX_RATIO_HARRIS;
    if (const int kb = harris ? 0 : t_block_size(s)) {
        // one block has nothing to overlap with: the in-place chain (shorter chains, no second buffer touched)
        if (t_use_pipeline(s) && iters > kb) return tableau_iterate_pipelined(s, w, iters, st, kb);
        return tableau_iterate_blocked(s, w, iters, st, kb);
    }
    if (t_use_small(s)) return tableau_iterate_small(s, w, iters, st);
    if (!harris && t_use_fused(s)) return tableau_iterate_fused(s, w, iters, st);
    const Rank1Plan plan = lpx_rank1_plan(s->m + 1, s->n + 1);
    LPX_REQUIRE(plan.grid.x <= LPX_MAXP, "too many column strips for the partial arrays");
    const int64_t nprice = plan.grid.x, nrow = lpx_cdiv(s->m + 1, LPX_BLOCK);
    // K1 refreshes the pricing partials of the NEXT iteration; the first iteration of a call scans the row itself
    t_launch_price(s, w, plan, st);
    const lpx_tableau sv = *s;
    const bool small = t_is_small(s) && !harris;
    auto one = [&](hipStream_t q) {
        if (small) {
            hipLaunchKernelGGL((k_t_pre_small<false>), dim3(1), dim3(T_SMALL_THREADS), 0, q, sv, w.price, nprice,
                               (const double *)nullptr, (int64_t)0, (int64_t)0, w.d, w.e, w.rowr);
        } else {
            hipLaunchKernelGGL(k_t_column, dim3((unsigned)nrow), dim3(LPX_BLOCK), 0, q, sv, w.price, nprice, w.ratio, w.d);
            if (harris)
                hipLaunchKernelGGL(k_t_harris, dim3((unsigned)nrow), dim3(LPX_BLOCK), 0, q, sv, w.ratio, nrow, (const double *)w.d, w.ratio2);
            hipLaunchKernelGGL(k_t_prepare, dim3((unsigned)lpx_cdiv(t_prep_len(&sv), LPX_BLOCK)), dim3(LPX_BLOCK), 0, q, sv,
                               harris ? w.ratio2 : w.ratio, nrow, (const double *)w.d, w.e, w.rowr);
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
    LPX_REQUIRE(s->ratio_rule == LPX_RATIO_TEXTBOOK, "the Harris-style ratio test is an option of lpx_tableau_iterate only");
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
    LPX_REQUIRE(s->ratio_rule == LPX_RATIO_TEXTBOOK, "the Harris-style ratio test is an option of lpx_tableau_iterate only");
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
    const int64_t c1 = (blockIdx.x == gridDim.x - 1) ? s.n : c0 + strip; // the last strip may be folded (lpx_rank1.hpp)
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
    LPX_REQUIRE(s->ratio_rule == LPX_RATIO_TEXTBOOK, "the Harris-style ratio test is an option of lpx_tableau_iterate only");
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
    LPX_REQUIRE(s->ratio_rule == LPX_RATIO_TEXTBOOK, "the Harris-style ratio test is an option of lpx_tableau_iterate only");
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
    const int64_t c1 = (blockIdx.x == gridDim.x - 1) ? s.n : c0 + strip; // the last strip may be folded (lpx_rank1.hpp)
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
                                                        double *e, double *rowr, const double *pend)
{
    lpx_ctrl *ctrl = s.ctrl;
    if (ctrl->status != LPX_RUNNING) return;
    int64_t r;
    if (pend != nullptr) {
        // lookahead: this pivot's leaving row was decided by the previous pivot's k_td_ratio_next, which may not touch
        // the control block (that pivot's K1 still had to read it): adopt the decision here, as k_td_ratio would have
        r = (int64_t)pend[0];
        const bool opt = pend[1] != 0.0;
        if (blockIdx.x == 0 && threadIdx.x == 0) {
            ctrl->counter += 1;
            ctrl->price_val = pend[2];
            if (opt) ctrl->status = LPX_OPTIMAL;
            else ctrl->leave_r = r;
        }
        if (opt) return;
    } else {
        r = ctrl->leave_r;
    }
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
    const double *col = recv + win * msg + 2;
    const int64_t gi = (int64_t)blockIdx.x * LPX_BLOCK + threadIdx.x;
    td_finish(s, r, best.i, col[r], (gi <= s.m) ? col[gi] : 0.0, best.v, -1.0, d, e, rowr);
}

// ---- lookahead for the partitioned dual loop: the proposal of pivot t+1 from the tableau as pivot t's K1 is about to
// leave it.  Everything it needs is thin: the x_B column (leaving row r'), row r' and the reduced-cost row (ratio test)
// and one column (the proposal), each formed from the old T, e and rowr with K1's own expressions, bit for bit.
// entry (i, j) of the tableau after the pending update (K1's expressions, lpx_rank1.hpp)
__device__ __forceinline__ double td_next_entry(const lpx_tableau &s, int64_t i, int64_t j, int64_t r, int64_t unit, const double *e,
                                                const double *rowr)
{
    if (j == unit) return s.slot_var ? e[i] : ((i == r) ? 1.0 : 0.0);
    return (i == r) ? e[i] * rowr[j] : fma(e[i], rowr[j], s.T[i * s.ld + j]);
}
__global__ __launch_bounds__(LPX_BLOCK) void k_td_row_next(lpx_tableau s, Partials R, const double *e, const double *rowr)
{
    __shared__ BlockRed red;
    if (s.ctrl->status != LPX_RUNNING) return;
    const int64_t i = (int64_t)blockIdx.x * LPX_BLOCK + threadIdx.x;
    const double v = (i < s.m) ? td_next_entry(s, i, s.n, s.ctrl->leave_r, -1, e, rowr) : LPX_INF;
    const double mn = block_min(v, red);
    const int64_t first = block_imin((i < s.m && v < 0.0) ? i : LPX_NOIDX, red);
    if (threadIdx.x == 0) {
        R.v0[blockIdx.x] = mn;
        R.i0[blockIdx.x] = first;
    }
}
__global__ __launch_bounds__(LPX_BLOCK) void k_td_ratio_next(lpx_tableau s, Partials R, int64_t nrow, Partials P, int64_t strip,
                                                            const double *e, const double *rowr, double *pend)
{
    __shared__ BlockRed red;
    if (s.ctrl->status != LPX_RUNNING) return;
    const double mn = combine_min(R.v0, nrow, red);
    const int64_t r2 = combine_imin(R.i0, nrow, red);
    const bool opt = lpx_is_optimal_min(mn); // dual_solvers.py:52-54
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        pend[0] = (double)(opt ? 0 : r2);
        pend[1] = opt ? 1.0 : 0.0;
        pend[2] = mn;
    }
    if (opt) return;
    const int64_t r = s.ctrl->leave_r, unit = t_unit_col(s);
    const int64_t c0 = (int64_t)blockIdx.x * strip;
    const int64_t c1 = (blockIdx.x == gridDim.x - 1) ? s.n : c0 + strip; // the last strip may be folded (lpx_rank1.hpp)
    VI best{LPX_INF, LPX_NOIDX};
    for (int64_t j = c0 + threadIdx.x; j < c1; j += LPX_BLOCK) {
        // basic columns count as 0 (dual_solvers.py:38-39; var_state already reflects the pending pivot)
        const double row = (!s.slot_var && s.var_state[j] == LPX_VS_BASIC) ? 0.0 : td_next_entry(s, r2, j, r, unit, e, rowr);
        const double rc = td_next_entry(s, s.m, j, r, unit, e, rowr);
        const double th = lpx_dsd(rc, row, false); // never the first iteration
        if (vi_less(th, t_var(s, j), best.v, best.i)) { best.v = th; best.i = t_var(s, j); }
    }
    best = block_argmin(best, red);
    if (threadIdx.x == 0) {
        P.v0[blockIdx.x] = best.v;
        P.i0[blockIdx.x] = best.i;
    }
}
__global__ __launch_bounds__(LPX_BLOCK) void k_td_propose_next(lpx_tableau s, Partials P, int64_t nprice, const double *e,
                                                              const double *rowr, const double *pend, double *sendbuf)
{
    __shared__ BlockRed red;
    if (s.ctrl->status != LPX_RUNNING) return;
    const int64_t i = (int64_t)blockIdx.x * LPX_BLOCK + threadIdx.x;
    if (pend[1] != 0.0) { // optimal after the pending pivot: the next commit stops before it looks at the proposals
        if (i == 0) {
            sendbuf[0] = LPX_INF;
            sendbuf[1] = (double)LPX_NOIDX;
        }
        if (i <= s.m) sendbuf[2 + i] = 0.0;
        return;
    }
    const VI t = combine_argmin(P.v0, P.i0, nprice, red);
    if (i == 0) {
        sendbuf[0] = t.v;
        sendbuf[1] = (double)t.i;
    }
    if (i <= s.m) sendbuf[2 + i] = td_next_entry(s, i, t_slot(s, t.i), s.ctrl->leave_r, t_unit_col(s), e, rowr);
}

static void td_launch_select(const lpx_tableau *s, const TabWs &w, const Rank1Plan &p, hipStream_t st)
{
    const int64_t nrow = lpx_cdiv(s->m, LPX_BLOCK), strip = 2 * (int64_t)p.threads * p.vpt;
    hipLaunchKernelGGL(k_td_row, dim3((unsigned)nrow), dim3(LPX_BLOCK), 0, st, *s, w.ratio);
    hipLaunchKernelGGL(k_td_ratio, dim3(p.grid.x), dim3(LPX_BLOCK), 0, st, *s, w.ratio, nrow, w.price, strip);
}

static int tableau_dual_iterate_fused(const lpx_tableau *s, const TabWs &w, int64_t iters, hipStream_t st)
{
    LPX_REQUIRE(lpx_aligned16(s->T_alt), "T_alt must be 16-byte aligned");
    DualFusedArgs a{};
    a.s = *s;
    a.base = w.base;
    a.nla = lpx_cdiv(s->m + 1, TF_THREADS);
    a.nstrips = lpx_cdiv(s->n + 1, 2 * TF_THREADS);
    int64_t rpb = lpx_env_int("LPX_FUSED_ROWS", 0); // as tableau_iterate_fused
    if (rpb <= 0) {
        rpb = 1;
        while (rpb < 8 && a.nstrips * lpx_cdiv(s->m + 1, 2 * rpb) >= 240) rpb *= 2;
    }
    a.rpb = rpb;
    const int64_t nupd = a.nstrips * lpx_cdiv(s->m + 1, rpb);
    LPX_REQUIRE(a.nla + nupd + 1 < ((int64_t)1 << 31), "grid too large");
    auto args_for = [&](int par) {
        DualFusedArgs b = a;
        b.Tin = par ? s->T_alt : s->T;
        b.Tout = par ? s->T : s->T_alt;
        b.dc = w.d2[par]; b.dn = w.d2[par ^ 1];
        b.nc = w.next[par]; b.nn = w.next[par ^ 1];
        return b;
    };
    { // bootstrap: leaving row, ratio test and entering column of the state in T -> the records of parity 0
        DualFusedArgs b = args_for(1);
        b.Tin = s->T;
        b.Tout = nullptr;
        hipLaunchKernelGGL(k_td_boot, dim3((unsigned)a.nla), dim3(TF_THREADS), 0, st, b);
    }
    int64_t it = 0;
    auto one = [&](hipStream_t q) {
        const DualFusedArgs b = args_for((int)(it & 1));
        lpx_k1_begin(q);
        hipLaunchKernelGGL(k_td_fused, dim3((unsigned)(a.nla + nupd + 1)), dim3(TF_THREADS), 0, q, b);
        lpx_k1_end(q);
        it += 1;
    };
    struct {
        lpx_tableau s;
        int64_t rpb;
        int what;
    } key;
    memset(&key, 0, sizeof key);
    key.s = *s; key.rpb = rpb; key.what = 4;
    const int rc = lpx_run_iterations(&key, sizeof key, iters, st, one);
    const int64_t count = (s->m + 1) * s->ld;
    int64_t nb = lpx_cdiv(count / 2, LPX_BLOCK);
    if (nb > 4096) nb = 4096;
    hipLaunchKernelGGL(k_t_syncback, dim3((unsigned)nb), dim3(LPX_BLOCK), 0, st, *s, (const int64_t *)w.base,
                       (const double *)s->T_alt, count);
    if (rc) return rc;
    LPX_LAUNCH_CHECK();
    return LPX_OK;
}

static int tableau_dual_iterate_blocked(const lpx_tableau *s, const TabWs &w, int64_t iters, hipStream_t st, int KB)
{
    const TbArgs a = tb_args(s, w);
    const int64_t ncopy = lpx_roundup(s->n + 1, 2);
    const int64_t nrow = lpx_cdiv(ncopy > s->m + 1 ? ncopy : s->m + 1, TB_THREADS);
    const int64_t ncol = lpx_cdiv(s->n > s->m + 1 ? s->n : s->m + 1, TB_THREADS);
    hipLaunchKernelGGL(k_tbd_boot, dim3((unsigned)ncol), dim3(TB_THREADS), 0, st, a);
    int64_t it = 0;
    auto one = [&](hipStream_t q) {
        TbArgs b = a;
        b.u = (int)(it % KB);
        hipLaunchKernelGGL(k_tbd_row, dim3((unsigned)nrow), dim3(TB_THREADS), 0, q, b);
        hipLaunchKernelGGL((k_tbd_column<TB_LOCAL>), dim3((unsigned)ncol), dim3(TB_THREADS), 0, q, b);
        it += 1;
    };
    struct {
        lpx_tableau s;
        int kb, what;
    } key;
    memset(&key, 0, sizeof key);
    key.s = *s; key.kb = KB; key.what = 6;
    while (iters > 0) { // a block = KB pivots' decisions (replayed as one graph) + one pass over the tableau
        const int64_t k = iters < KB ? iters : KB;
        it = 0;
        const int rc = lpx_run_iterations(&key, sizeof key, k, st, one, KB, true);
        if (rc) return rc;
        tb_flush(s, w, KB, st);
        iters -= k;
    }
    LPX_LAUNCH_CHECK();
    return LPX_OK;
}

// ---- column-partitioned dual loop with deferred updates (see lpx_tableau_block_* for the primal twin): the leaving row
// comes from the replicated x_B, so every rank picks it without communication; each rank proposes its best column
extern "C" int lpx_tableau_dual_block_begin(const lpx_tableau *s, void *stream)
{
    int rc = check_tableau(s);
    if (rc) return rc;
    LPX_REQUIRE(s->ratio_rule == LPX_RATIO_TEXTBOOK, "the Harris-style ratio test is an option of lpx_tableau_iterate only");
    const TabWs w = tab_ws(s->m, s->n, s->workspace);
    const TbArgs a = tb_args(s, w);
    hipLaunchKernelGGL(k_tbd_boot, dim3((unsigned)lpx_cdiv(s->n > s->m + 1 ? s->n : s->m + 1, TB_THREADS)), dim3(TB_THREADS), 0,
                       (hipStream_t)stream, a);
    LPX_LAUNCH_CHECK();
    return LPX_OK;
}
extern "C" int lpx_tableau_dual_block_propose(const lpx_tableau *s, int64_t pending, double *sendbuf, void *stream)
{
    int rc = check_tableau(s);
    if (rc) return rc;
    LPX_REQUIRE(s->ratio_rule == LPX_RATIO_TEXTBOOK, "the Harris-style ratio test is an option of lpx_tableau_iterate only");
    LPX_REQUIRE(sendbuf != nullptr && pending >= 0 && pending < TB_KMAX, "bad send buffer / pending count");
    const TabWs w = tab_ws(s->m, s->n, s->workspace);
    TbArgs a = tb_args(s, w);
    a.u = (int)pending;
    a.sendbuf = sendbuf;
    const int64_t ncopy = lpx_roundup(s->n + 1, 2);
    hipStream_t st = (hipStream_t)stream;
    hipLaunchKernelGGL(k_tbd_row, dim3((unsigned)lpx_cdiv(ncopy > s->m + 1 ? ncopy : s->m + 1, TB_THREADS)), dim3(TB_THREADS), 0, st, a);
    hipLaunchKernelGGL((k_tbd_column<TB_PROPOSE>), dim3((unsigned)lpx_cdiv(s->m + 1, TB_THREADS)), dim3(TB_THREADS), 0, st, a);
    LPX_LAUNCH_CHECK();
    return LPX_OK;
}
extern "C" int lpx_tableau_dual_block_commit(const lpx_tableau *s, int64_t pending, const double *recvbuf, int64_t nranks, void *stream)
{
    int rc = check_tableau(s);
    if (rc) return rc;
    LPX_REQUIRE(s->ratio_rule == LPX_RATIO_TEXTBOOK, "the Harris-style ratio test is an option of lpx_tableau_iterate only");
    LPX_REQUIRE(recvbuf != nullptr && nranks >= 1 && nranks <= 64, "bad receive buffer / rank count");
    LPX_REQUIRE(pending >= 0 && pending < TB_KMAX, "bad pending count");
    const TabWs w = tab_ws(s->m, s->n, s->workspace);
    TbArgs a = tb_args(s, w);
    a.u = (int)pending;
    a.recv = recvbuf;
    a.nranks = nranks;
    a.msg = lpx_tableau_msg_len(s->m);
    hipLaunchKernelGGL((k_tbd_column<TB_SELECT>), dim3((unsigned)lpx_cdiv(s->n > s->m + 1 ? s->n : s->m + 1, TB_THREADS)),
                       dim3(TB_THREADS), 0, (hipStream_t)stream, a);
    LPX_LAUNCH_CHECK();
    return LPX_OK;
}

// the dual twin of lpx_tableau_dist_iterate: leaving row from the replicated x_B, each rank's best column, one all-gather,
// first argmin of theta over the ranks; one pass over the slice per block (in place)
extern "C" int lpx_tableau_dual_dist_iterate(const lpx_tableau *s, void *comm, int64_t nranks, int64_t iters, int64_t block,
                                             double *sendbuf, double *recvbuf, void *stream)
{
    int rc = check_tableau(s);
    if (rc) return rc;
    LPX_REQUIRE(s->ratio_rule == LPX_RATIO_TEXTBOOK, "the Harris-style ratio test is an option of lpx_tableau_iterate only");
    LPX_REQUIRE(comm != nullptr && sendbuf != nullptr && recvbuf != nullptr && nranks >= 1 && nranks <= 64, "bad communicator / buffers / rank count");
    LPX_REQUIRE(block == 8 || block == 16 || block == 32, "block size must be 8, 16 or 32 (the same on every rank)");
    LPX_REQUIRE(lpx_cdiv(s->n, TB_THREADS) <= LPX_MAXP && lpx_cdiv(s->m + 2, TB_THREADS) <= LPX_MAXP, "slice too large for the partial arrays");
    if (iters <= 0) return LPX_OK;
    const TabWs w = tab_ws(s->m, s->n, s->workspace);
    hipStream_t st = (hipStream_t)stream;
    TbArgs a = tb_args(s, w);
    a.sendbuf = sendbuf;
    a.recv = recvbuf;
    a.nranks = nranks;
    a.msg = lpx_tableau_msg_len(s->m);
    const int64_t ncopy = lpx_roundup(s->n + 1, 2);
    const int64_t nrow = lpx_cdiv(ncopy > s->m + 1 ? ncopy : s->m + 1, TB_THREADS);
    const int64_t ncol = lpx_cdiv(s->n > s->m + 1 ? s->n : s->m + 1, TB_THREADS);
    hipLaunchKernelGGL(k_tbd_boot, dim3((unsigned)ncol), dim3(TB_THREADS), 0, st, a);
    for (int64_t it = 0; it < iters; ++it) {
        TbArgs b = a;
        b.u = (int)(it % block);
        hipLaunchKernelGGL(k_tbd_row, dim3((unsigned)nrow), dim3(TB_THREADS), 0, st, b);
        hipLaunchKernelGGL((k_tbd_column<TB_PROPOSE>), dim3((unsigned)lpx_cdiv(s->m + 1, TB_THREADS)), dim3(TB_THREADS), 0, st, b);
        LPX_NCCL(ncclAllGather(sendbuf, recvbuf, (size_t)a.msg, ncclDouble, (ncclComm_t)comm, st));
        hipLaunchKernelGGL((k_tbd_column<TB_SELECT>), dim3((unsigned)ncol), dim3(TB_THREADS), 0, st, b);
        if (it % block == block - 1 || it + 1 == iters) tb_flush(s, w, (int)block, st);
    }
    LPX_LAUNCH_CHECK();
    return LPX_OK;
}

extern "C" int lpx_tableau_dual_iterate(const lpx_tableau *s, int64_t iters, void *stream)
{
    int rc = check_tableau(s);
    if (rc) return rc;
    LPX_REQUIRE(s->ratio_rule == LPX_RATIO_TEXTBOOK, "the Harris-style ratio test is an option of lpx_tableau_iterate only");
    LPX_REQUIRE(s->slot_var != nullptr || (s->col0 == 0 && s->n_global == s->n), "lpx_tableau_dual_iterate needs an unpartitioned tableau");
    const TabWs w = tab_ws(s->m, s->n, s->workspace);
    hipStream_t st = (hipStream_t)stream;
    const Rank1Plan plan = lpx_rank1_plan(s->m + 1, s->n + 1);
    LPX_REQUIRE(plan.grid.x <= LPX_MAXP, "too many column strips for the partial arrays");
    if (iters <= 0) return LPX_OK;
    if (const int kb = t_block_size(s)) return tableau_dual_iterate_blocked(s, w, iters, st, kb);
    if (t_use_fused(s)) return tableau_dual_iterate_fused(s, w, iters, st);
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
    LPX_REQUIRE(s->ratio_rule == LPX_RATIO_TEXTBOOK, "the Harris-style ratio test is an option of lpx_tableau_iterate only");
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
    LPX_REQUIRE(s->ratio_rule == LPX_RATIO_TEXTBOOK, "the Harris-style ratio test is an option of lpx_tableau_iterate only");
    LPX_REQUIRE(recvbuf != nullptr && nranks >= 1 && nranks <= 64, "bad receive buffer / rank count");
    const TabWs w = tab_ws(s->m, s->n, s->workspace);
    hipStream_t st = (hipStream_t)stream;
    const Rank1Plan plan = lpx_rank1_plan(s->m + 1, s->n + 1);
    hipLaunchKernelGGL(k_td_commit, dim3((unsigned)lpx_cdiv(t_prep_len(s), LPX_BLOCK)), dim3(LPX_BLOCK), 0, st, *s, recvbuf, nranks,
                       lpx_tableau_msg_len(s->m), w.d, w.e, w.rowr, (const double *)nullptr);
    t_launch_update(s, w, plan, st);
    LPX_LAUNCH_CHECK();
    return LPX_OK;
}

// The same pivot in two halves, for the loop that overlaps the exchange of pivot t+1 with the update of pivot t (see
// lpx_tableau_commit_begin).  adopt != 0: the leaving row of this pivot was decided by the previous call's lookahead
// (sendbuf_next != NULL there) instead of lpx_tableau_dual_propose.  sendbuf_next != NULL: also this rank's proposal for
// the next pivot.  The update itself is lpx_tableau_commit_update.
extern "C" int lpx_tableau_dual_commit_begin(const lpx_tableau *s, const double *recvbuf, int64_t nranks, int adopt,
                                             double *sendbuf_next, void *stream)
{
    int rc = check_tableau(s);
    if (rc) return rc;
    LPX_REQUIRE(s->ratio_rule == LPX_RATIO_TEXTBOOK, "the Harris-style ratio test is an option of lpx_tableau_iterate only");
    LPX_REQUIRE(recvbuf != nullptr && nranks >= 1 && nranks <= 64, "bad receive buffer / rank count");
    const TabWs w = tab_ws(s->m, s->n, s->workspace);
    hipStream_t st = (hipStream_t)stream;
    const Rank1Plan plan = lpx_rank1_plan(s->m + 1, s->n + 1);
    LPX_REQUIRE(plan.grid.x <= LPX_MAXP, "too many column strips for the partial arrays");
    hipLaunchKernelGGL(k_td_commit, dim3((unsigned)lpx_cdiv(t_prep_len(s), LPX_BLOCK)), dim3(LPX_BLOCK), 0, st, *s, recvbuf, nranks,
                       lpx_tableau_msg_len(s->m), w.d, w.e, w.rowr, adopt ? (const double *)w.pend : (const double *)nullptr);
    if (sendbuf_next != nullptr) {
        const int64_t nrow = lpx_cdiv(s->m, LPX_BLOCK), strip = 2 * (int64_t)plan.threads * plan.vpt;
        hipLaunchKernelGGL(k_td_row_next, dim3((unsigned)nrow), dim3(LPX_BLOCK), 0, st, *s, w.ratio, (const double *)w.e,
                           (const double *)w.rowr);
        hipLaunchKernelGGL(k_td_ratio_next, dim3(plan.grid.x), dim3(LPX_BLOCK), 0, st, *s, w.ratio, nrow, w.price, strip,
                           (const double *)w.e, (const double *)w.rowr, w.pend);
        hipLaunchKernelGGL(k_td_propose_next, dim3((unsigned)lpx_cdiv(s->m + 1, LPX_BLOCK)), dim3(LPX_BLOCK), 0, st, *s, w.price,
                           (int64_t)plan.grid.x, (const double *)w.e, (const double *)w.rowr, (const double *)w.pend, sendbuf_next);
    }
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
