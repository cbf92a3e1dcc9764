// lpx_tableau_blocked.hpp -- the tableau loop with DEFERRED updates: one pass over the tableau per KB pivots.
//
// The rank-1 update of a pivot (primal_solvers.py:247-253 as a tableau update, lpx_rank1.hpp) moves all 16 (m+1)(n+1)
// bytes of T through HBM and does one fma per element: at 20000 x 20000 that is 1 ms of streaming per pivot, and nothing
// else in the pivot costs more than a few microseconds.  But the DECISIONS of a pivot read only thin pieces of the
// tableau: the reduced-cost row (entering variable, primal_solvers.py:135-142), one column and x_B (ratio test,
// :149-167), and its update needs one row (the old pivot row).  A thin piece of the tableau "as the pending updates
// would leave it" can be formed from the stale tableau T0 in HBM and the pending (eta column, pivot row) pairs with the
// update's own expressions, element by element in the update's own order:
//
//     x <- T0[i][j];   for u = 0 .. P-1:   x <- (i == r_u) ? e_u[i] * R_u[j] : fma(e_u[i], R_u[j], x)
//
// (with the rewritten column of each pivot handled as K1 does) -- bit for bit the value the P eager updates would have
// produced, because every element goes through the same chain of roundings.  So KB pivots are decided from T0, P <= KB
// stored pairs, and the two thin pieces that ARE kept current (the reduced-cost row rc[] and the x_B column xb[], updated
// eagerly with the same expressions), and then ONE kernel applies the KB rank-1 updates to every element in order: each
// element is read once and written once per KB pivots instead of once per pivot.  HBM traffic per pivot drops by KB;
// the fma count is unchanged (KB per element and pass: 0.33 ms of fp64 VALU at 20000 x 20000 and KB = 32, beside 1 ms
// of streaming).  Same arithmetic, same bits, same pivots as the eager loop.
//
// Chain per pivot (position u in its block = number of pending updates, a kernel ARGUMENT like the parity of the
// single-launch chain: launches after a terminal status are no-ops, so host and device cannot drift):
//     k_tb_column(u):  entering decision from the pricing partials; gathers column s of T0, applies the u pending
//                      updates; ratio-test partials against xb[]
//     k_tb_prepare(u): leaving row from the ratio partials; eta column -> E[.][u]; xb[] updated; row r of T0 with the
//                      u pending updates applied -> R[u][.]; rc[] updated and priced (partials for the next pivot);
//                      bookkeeping (basis, slot maps, log, control block)
// and after KB pivots (or at the end of a call)
//     k_tb_flush:      T <- T with the cnt pending updates applied, in order; R strips in registers, e_u[i] wave-uniform
// No kernel reads a word that the same launch writes: cnt / rowflag are written by k_tb_prepare (and cleared by
// k_tb_column(0), also when the loop has ended, so a later flush applies nothing), read by k_tb_flush only.
#pragma once

#include "lpx_tableau_blocked_decl.hpp"

#ifndef LPX_TB_APPLY_DEFINED
#define LPX_TB_APPLY_DEFINED
// entry (i, j) of the tableau after pending update u, from its value x before it (K1's expressions, lpx_rank1.hpp)
__device__ __forceinline__ double tb_apply(double x, bool is_slot, bool is_row, double e, double ru, bool cond)
{
    if (is_slot) return cond ? e : (is_row ? 1.0 : 0.0); // eta column (condensed) / unit vector (full form)
    return is_row ? e * ru : fma(e, ru, x);
}
#endif

// -DLPX_TB_STAMPS: diagnostic build, workgroup 0 prints where the two per-pivot kernels spend their time (10 ns ticks)
#ifdef LPX_TB_STAMPS
__device__ long long g_tb_stamp[2][16];
#define LPX_TB_STAMP(k, n) do { if (blockIdx.x == 0 && threadIdx.x == 0) g_tb_stamp[k][n] = (long long)wall_clock64(); } while (0)
#else
#define LPX_TB_STAMP(k, n) do { } while (0)
#endif

struct TbArgs {
    lpx_tableau s;
    TbWs b;
    Partials price, ratio; // price: v0 = min rc, i0 = entering candidate (variable id), i1 = its slot
                           // ratio: v0 = min ratio, i0 = row, v1 = max d, i1 = basic variable of that row
    double *d;
    int64_t nprice, nratio;
    int u;
    // column-partitioned tableau: this rank's proposal (written) / the gathered proposals of all ranks (read)
    double *sendbuf;
    const double *recv;
    int64_t nranks, msg;
};
// k_tb_column's three roles
#define TB_LOCAL 0   /* unpartitioned: entering decision from the pricing partials, column, ratio partials */
#define TB_PROPOSE 1 /* partitioned, before the exchange: local candidate and its column -> sendbuf */
#define TB_SELECT 2  /* partitioned, after the exchange: global entering decision, the winner's column, ratio partials */

// pricing candidate of one column (first-negative rule or opt-in Dantzig), as k_t_price
struct TbCand {
    double mn;
    int64_t var, slot;
};
__device__ __forceinline__ void tb_price_one(int64_t pricing, double rc, int64_t v, int64_t slot, TbCand &c)
{
    if (pricing == LPX_PRICE_DANTZIG) {
        if (vi_less(rc, v, c.mn, c.var)) c = TbCand{rc, v, slot};
    } else {
        c.mn = fmin(c.mn, rc);
        if (rc < 0.0 && v < c.var) { c.var = v; c.slot = slot; }
    }
}
// workgroup reduction of pricing candidates; every thread returns the winner (variable ids are distinct)
__device__ __forceinline__ TbCand tb_price_reduce(int64_t pricing, TbCand c, BlockRed &red, int64_t *sh_slot)
{
    TbCand w;
    if (pricing == LPX_PRICE_DANTZIG) {
        const VI t = block_argmin(VI{c.mn, c.var}, red);
        w.mn = t.v;
        w.var = t.i;
    } else {
        w.mn = block_min(c.mn, red);
        w.var = block_imin(c.var, red);
    }
    if (threadIdx.x == 0) *sh_slot = -1;
    __syncthreads();
    if (w.var != LPX_NOIDX && c.var == w.var) *sh_slot = c.slot;
    __syncthreads();
    w.slot = *sh_slot;
    __syncthreads();
    return w;
}

// start of a call: rc[] and xb[] from the tableau, pricing partials of the carried row
__global__ __launch_bounds__(TB_THREADS) void k_tb_boot(TbArgs a)
{
    __shared__ BlockRed red;
    __shared__ int64_t sh_slot;
    const lpx_tableau &s = a.s;
    const int64_t g = (int64_t)blockIdx.x * TB_THREADS + threadIdx.x;
    if (g == 0) *a.b.cnt = 0;
    if (s.ctrl->status != LPX_RUNNING) return;
    if (g < s.m) a.b.xb[g] = s.T[g * s.ld + s.n];
    if ((int64_t)blockIdx.x >= a.nprice) return;
    TbCand c{LPX_INF, LPX_NOIDX, -1};
    if (g < s.n) {
        const double rc = s.T[s.m * s.ld + g];
        a.b.rc[g] = rc;
        tb_price_one(s.pricing, rc, t_var(s, g), g, c);
    }
    c = tb_price_reduce(s.pricing, c, red, &sh_slot);
    if (threadIdx.x == 0) {
        a.price.v0[blockIdx.x] = c.mn;
        a.price.i0[blockIdx.x] = c.var;
        a.price.i1[blockIdx.x] = c.slot;
    }
}

// entering decision, column of the tableau as the u pending updates leave it, ratio-test partials
template <int MODE>
__global__ __launch_bounds__(TB_THREADS) void k_tb_column(TbArgs a)
{
    __shared__ BlockRed red;
    __shared__ int64_t sh_slot;
    const lpx_tableau &s = a.s;
    lpx_ctrl *ctrl = s.ctrl;
    const int tid = threadIdx.x;
    const int64_t i = (int64_t)blockIdx.x * TB_THREADS + tid;
    const int P = a.u;
    LPX_TB_STAMP(0, 0);
    if (P == 0 && MODE != TB_SELECT) { // a new block: nothing pending (also after the loop has ended: the flush then applies nothing)
        if (i <= s.m) a.b.rowflag[i] = 0u;
        if (i == 0) *a.b.cnt = 0;
    }
    // loads whose addresses depend on nothing decided in this launch
    const int32_t status = ctrl->status;
    const bool f32 = ctrl->f32_phase != 0;
    const bool in = i < s.m, inm = i <= s.m;
    const double xi = (in && MODE != TB_PROPOSE) ? a.b.xb[i] : 0.0;
    const int64_t bvi = (in && MODE != TB_PROPOSE) ? s.basis[i] : -1;
    // the pending updates' pivot rows and rewritten columns: ONE vector load each (lane l holds update l), broadcast with
    // v_readlane in the chain -- as 2 x 32 scalar loads they missed the scalar cache one after the other (5 us at P = 31)
    const int l32 = tid & 31;
    const int pv = (MODE != TB_SELECT && l32 < P) ? ((const int *)a.b.prow)[2 * l32] : -2; // rows and slots < 2^31 (or -1)
    const int sv = (MODE != TB_SELECT && l32 < P) ? ((const int *)a.b.pslot)[2 * l32] : -2;
    double e[TB_KMAX];
    if (MODE != TB_SELECT) {
        const double *Eti = a.b.Et + (inm ? i : 0); // coalesced over the rows of the workgroup
#pragma unroll
        for (int u = 0; u < TB_KMAX; ++u) e[u] = (u < P) ? Eti[u * a.b.lde] : 0.0;
    }
    if (status != LPX_RUNNING) return;
    LPX_TB_STAMP(0, 1);
    // entering variable (primal_solvers.py:135-142) and its column slot
    TbCand c{LPX_INF, LPX_NOIDX, -1};
    int64_t win = 0;
    if (MODE == TB_SELECT) { // lowest global index over the ranks' candidates (a handful of ranks: every thread scans them)
        for (int64_t g = 0; g < a.nranks; ++g) {
            const double v = a.recv[g * a.msg];
            const int64_t idx = (int64_t)a.recv[g * a.msg + 1]; // exact: ids < 2^53, LPX_NOIDX = 2^62
            if (s.pricing == LPX_PRICE_DANTZIG) {
                if (vi_less(v, idx, c.mn, c.var)) { c.mn = v; c.var = idx; win = g; }
            } else {
                c.mn = fmin(c.mn, v);
                if (idx < c.var) { c.var = idx; win = g; }
            }
        }
        c.slot = (c.var == LPX_NOIDX) ? -1 : t_slot(s, c.var); // -1: the column lives on another rank
    } else {
        for (int64_t k = tid; k < a.nprice; k += TB_THREADS) {
            const double v = a.price.v0[k];
            const int64_t id = a.price.i0[k], sl = a.price.i1[k];
            if (s.pricing == LPX_PRICE_DANTZIG) {
                if (vi_less(v, id, c.mn, c.var)) c = TbCand{v, id, sl};
            } else {
                c.mn = fmin(c.mn, v);
                if (id < c.var) { c.var = id; c.slot = sl; }
            }
        }
        LPX_TB_STAMP(0, 2);
        c = tb_price_reduce(s.pricing, c, red, &sh_slot);
    }
    LPX_TB_STAMP(0, 3);
    const int64_t slot = c.slot;
    const bool cond = s.slot_var != nullptr;
    if (MODE == TB_PROPOSE) { // (local min rc, first owned candidate, its column as the pending updates leave it)
        if (i == 0) {
            a.sendbuf[0] = c.mn;
            a.sendbuf[1] = (double)c.var;
        }
        double di = 0.0;
        if (c.var != LPX_NOIDX) { // workgroup-uniform
            const double rv = (l32 < P) ? a.b.R[l32 * a.b.ldr + slot] : 0.0;
            di = inm ? s.T[i * s.ld + slot] : 0.0;
#pragma unroll
            for (int u = 0; u < TB_KMAX; ++u)
                if (u < P)
                    di = tb_apply(di, (int)slot == __builtin_amdgcn_readlane(sv, u), (int)i == __builtin_amdgcn_readlane(pv, u), e[u],
                                  tf_readlane_f64(rv, u), cond);
        }
        if (inm) a.sendbuf[2 + i] = di;
        return;
    }
    const bool opt = lpx_is_optimal_min(c.mn);
    if (blockIdx.x == 0 && tid == 0) {
        ctrl->counter += 1;
        ctrl->price_val = c.mn;
        if (opt) ctrl->status = LPX_OPTIMAL;
        else {
            ctrl->enter_j = c.var;
            a.b.scal[0] = (double)slot;
        }
    }
    if (opt) return;
    double di = 0.0;
    if (MODE == TB_SELECT) {
        if (inm) {
            di = a.recv[win * a.msg + 2 + i];
            a.d[i] = di;
        }
    } else {
        const double rv = (l32 < P) ? a.b.R[l32 * a.b.ldr + slot] : 0.0; // lane l: entry `slot` of pivot row l
        di = inm ? s.T[i * s.ld + slot] : 0.0;
#pragma unroll
        for (int u = 0; u < TB_KMAX; ++u)
            if (u < P)
                di = tb_apply(di, (int)slot == __builtin_amdgcn_readlane(sv, u), (int)i == __builtin_amdgcn_readlane(pv, u), e[u],
                              tf_readlane_f64(rv, u), cond);
        if (inm) a.d[i] = di; // d[m] = the entering variable's reduced cost
    }
    // ratio test (primal_solvers.py:149-167): first minimum over this workgroup's rows, with what the winner needs
    const VI t{in ? lpx_psd(xi, di, f32) : LPX_INF, in ? i : LPX_NOIDX};
    LPX_TB_STAMP(0, 4);
    const VI w = block_argmin(t, red);
    const double dmax = block_max(in ? di : -LPX_INF, red);
    LPX_TB_STAMP(0, 5);
    if (tid == 0) {
        a.ratio.v1[blockIdx.x] = dmax;
        if (w.i == LPX_NOIDX) {
            a.ratio.v0[blockIdx.x] = LPX_INF;
            a.ratio.i0[blockIdx.x] = LPX_NOIDX;
        }
    }
    if (in && i == w.i) {
        a.ratio.v0[blockIdx.x] = w.v;
        a.ratio.i0[blockIdx.x] = w.i;
        a.ratio.i1[blockIdx.x] = bvi;
        a.b.dbest[blockIdx.x] = di;
        a.b.xbest[blockIdx.x] = xi;
    }
    LPX_TB_STAMP(0, 6);
}

// leaving row, eta column, x_B, the pivot row as the pending updates leave it, the reduced-cost row and its pricing
// partials, bookkeeping
__global__ __launch_bounds__(TB_THREADS) void k_tb_prepare(TbArgs a)
{
    __shared__ BlockRed red;
    __shared__ int64_t sh_slot;
    __shared__ double sh_pay[2];
    __shared__ int64_t sh_bv;
    const lpx_tableau &s = a.s;
    lpx_ctrl *ctrl = s.ctrl;
    const int tid = threadIdx.x;
    const int64_t g = (int64_t)blockIdx.x * TB_THREADS + tid;
    const int P = a.u;
    LPX_TB_STAMP(1, 0);
    const int64_t m = s.m, n = s.n, ncopy = lpx_roundup_dev(n + 1, 2);
    // loads whose addresses depend on nothing decided in this launch
    const int32_t status = ctrl->status;
    const int64_t j = ctrl->enter_j;
    const int64_t slot = (int64_t)a.b.scal[0];
    const bool incol = g < ncopy;
    double ru[TB_KMAX];
#pragma unroll
    for (int u = 0; u < TB_KMAX; ++u) ru[u] = (u < P && incol) ? a.b.R[u * a.b.ldr + g] : 0.0;
    const double rc_old = (g < n) ? a.b.rc[g] : 0.0;
    const double dg = (g <= m) ? a.d[g] : 0.0, dm = a.d[m];
    const double xg = (g < m) ? a.b.xb[g] : 0.0;
    const int64_t varg = (g < n) ? t_var(s, g) : LPX_NOIDX;
    const int l32 = tid & 31; // lane l holds pending update l's pivot row / rewritten column / multiplier of row r
    const int pv = (l32 < P) ? ((const int *)a.b.prow)[2 * l32] : -2;
    const int sv = (l32 < P) ? ((const int *)a.b.pslot)[2 * l32] : -2;
    if (status != LPX_RUNNING) return;
    LPX_TB_STAMP(1, 1);
    // leaving row: first minimum ratio over the workgroups' partials (primal_solvers.py:166, np.argmin)
    VI best{LPX_INF, LPX_NOIDX};
    int64_t bk = -1;
    double dmax = -LPX_INF;
    for (int64_t k = tid; k < a.nratio; k += TB_THREADS) {
        const double v = a.ratio.v0[k];
        const int64_t r = a.ratio.i0[k];
        if (vi_less(v, r, best.v, best.i)) { best = VI{v, r}; bk = k; }
        dmax = fmax(dmax, a.ratio.v1[k]);
    }
    LPX_TB_STAMP(1, 2);
    const VI w = block_argmin(best, red);
    dmax = block_max(dmax, red);
    LPX_TB_STAMP(1, 3);
    if (dmax <= 0.0) { // primal_solvers.py:144-147
        if (g == 0) {
            ctrl->aux_val = dmax;
            ctrl->status = LPX_PRIMAL_UNBOUNDED;
        }
        return;
    }
    if (bk >= 0 && best.i == w.i) { // rows are distinct: one thread
        sh_pay[0] = a.b.dbest[bk];
        sh_pay[1] = a.b.xbest[bk];
        sh_bv = a.ratio.i1[bk];
    }
    __syncthreads();
    const int64_t r = w.i, leaving = sh_bv;
    const double dr = sh_pay[0], xr = sh_pay[1];
    const bool cond = s.slot_var != nullptr;
    LPX_TB_STAMP(1, 4);
    // eta column (primal_solvers.py:217-245), x_B (:251-253)
    if (g <= m) {
        const double eg = (g == r) ? 1.0 / dr : (-dg) / dr; // e[m] eliminates rc_j
        a.b.E[g * TB_KMAX + P] = eg;
        a.b.Et[P * a.b.lde + g] = eg;
        if (g < m) a.b.xb[g] = (g == r) ? eg * xr : fma(eg, xr, xg);
    }
    LPX_TB_STAMP(1, 5);
    // row r of the tableau as the P pending updates leave it: the old pivot row of this update
    double rowp = 0.0;
    {
        const double erv = (l32 < P) ? a.b.E[r * TB_KMAX + l32] : 0.0;
        rowp = incol ? s.T[r * s.ld + g] : 0.0;
#pragma unroll
        for (int u = 0; u < TB_KMAX; ++u)
            if (u < P)
                rowp = tb_apply(rowp, (int)g == __builtin_amdgcn_readlane(sv, u), (int)r == __builtin_amdgcn_readlane(pv, u),
                                tf_readlane_f64(erv, u), ru[u], cond);
        if (incol) a.b.R[P * a.b.ldr + g] = rowp;
    }
    LPX_TB_STAMP(1, 6);
    // the reduced-cost row after this pivot, and its pricing partial (the next pivot's entering decision)
    if ((int64_t)blockIdx.x < a.nprice) {
        TbCand c{LPX_INF, LPX_NOIDX, -1};
        if (g < n) {
            const double em = (-dm) / dr;
            double rc;
            int64_t v = varg;
            if (g == slot) { // the rewritten column: eta column (condensed; its variable is now the leaving one) / unit vector
                rc = cond ? em : 0.0;
                if (cond) v = leaving;
            } else {
                rc = fma(em, rowp, rc_old);
            }
            a.b.rc[g] = rc;
            tb_price_one(s.pricing, rc, v, g, c);
        }
        c = tb_price_reduce(s.pricing, c, red, &sh_slot);
        if (tid == 0) {
            a.price.v0[blockIdx.x] = c.mn;
            a.price.i0[blockIdx.x] = c.var;
            a.price.i1[blockIdx.x] = c.slot;
        }
    }
    LPX_TB_STAMP(1, 7);
    if (g == 0) { // bookkeeping of the basis change (primal_solvers.py:96-108, :261)
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
        ctrl->ratio_val = w.v;
        ctrl->aux_val = dmax;
        ctrl->flip = 0;
        ctrl->f32_phase = 0;
        a.b.prow[P] = r;
        a.b.pslot[P] = slot;
        a.b.rowflag[r] |= 1u << P;
        *a.b.cnt = P + 1;
#ifdef LPX_TB_STAMPS
        g_tb_stamp[1][8] = (long long)wall_clock64();
        if (ctrl->pivots >= 96 && ctrl->pivots < 160 && (P == 0 || P == 15 || P == 31)) {
            const long long *c0 = g_tb_stamp[0], *p0 = g_tb_stamp[1];
            printf("P=%d column: status %lld | partials %lld | price reduce %lld | column+chain %lld | ratio reduce %lld | end %lld || "
                   "gap %lld || prepare: status %lld | partials %lld | ratio reduce %lld | payload %lld | eta %lld | row chain %lld | "
                   "rc+price reduce %lld | books %lld\n", P, c0[1] - c0[0], c0[2] - c0[1], c0[3] - c0[2], c0[4] - c0[3], c0[5] - c0[4],
                   c0[6] - c0[5], p0[0] - c0[6], p0[1] - p0[0], p0[2] - p0[1], p0[3] - p0[2], p0[4] - p0[3], p0[5] - p0[4],
                   p0[6] - p0[5], p0[7] - p0[6], p0[8] - p0[7]);
        }
#endif
    }
}

