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
//
// PIPELINED chain (unpartitioned tableaus with a second buffer, tableau_iterate_pipelined in lpx_tableau.hip).  The pass
// and the next block's decisions have nothing to wait for in each other if the pass writes ANOTHER buffer: while pass k
// streams T_old -> T_new, block k+1 is decided from T_old, the KB pairs of block k (complete, read-only by now) and its
// own u pairs -- the same chain, KB + u long.  Every workspace array of a block's pairs exists twice (TbArgs::b = this
// block's, TbArgs::o = the previous block's); rc[], xb[] and the partials belong to the decisions alone.  A step then
// costs max(pass, decisions) instead of their sum.
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
__device__ long long g_tb_wgend[2][4096]; // every workgroup's last instruction: how long is a kernel's tail beyond workgroup 0?
#define LPX_TB_STAMP(k, n) do { if (blockIdx.x == 0 && threadIdx.x == 0) g_tb_stamp[k][n] = (long long)wall_clock64(); } while (0)
#define LPX_TB_WGEND(k) do { if (threadIdx.x == 0 && blockIdx.x < 4096) g_tb_wgend[k][blockIdx.x] = (long long)wall_clock64(); } while (0)
#else
#define LPX_TB_WGEND(k) do { } while (0)
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
    // pipelined chain (OLD instances of the two per-pivot kernels): s.T is the tableau as it stood BEFORE the block whose
    // pass is still in flight; that block's uo (eta column, pivot row) pairs come first in every chain, then the u of
    // this block.  dstid: the buffer this block's pass writes (TbWs::tsel)
    TbWs o;
    int uo, dstid;
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
// Workgroup reductions with ONE LDS exchange and ONE barrier: the whole record goes through the DPP wave reduction
// (every lane ends with the wave's result), lane 0 of each wave publishes it, every thread combines the four.  `sh` is
// private to the call site (never reused inside a kernel), so no barrier follows.  Ties are decided inside `op`
// (lowest variable id / lowest row), so the result does not depend on the order of combination.
template <typename T, typename Op>
__device__ __forceinline__ T tb_block_reduce(T v, Op op, T *sh)
{
    v = wave_allreduce(v, op);
    if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = v;
    __syncthreads();
    T r = sh[0];
#pragma unroll
    for (int w = 1; w < TB_THREADS / 64; ++w) r = op(r, sh[w]);
    return r;
}
__device__ __forceinline__ TbCand tb_price_reduce(int64_t pricing, TbCand c, TbCand *sh)
{
    // two capture-free operators (a capturing one was not inlined into the DPP reduction: 3-5 us per call)
    if (pricing == LPX_PRICE_DANTZIG)
        return tb_block_reduce(c, [](TbCand x, TbCand y) { return vi_less(y.mn, y.var, x.mn, x.var) ? y : x; }, sh);
    return tb_block_reduce(c, [](TbCand x, TbCand y) {
        TbCand t = (y.var < x.var) ? y : x; // first negative index: smallest variable id (ids are distinct)
        t.mn = fmin(x.mn, y.mn);
        return t;
    }, sh);
}
// ratio test record: first minimum ratio with what the winner needs later (its row, entering-column entry, x_B, basic variable)
struct TbRatio {
    double th;
    int64_t row;
    double dmax, dv, xv;
    int64_t bv;
};
__device__ __forceinline__ TbRatio tb_ratio_reduce(TbRatio t, TbRatio *sh)
{
    return tb_block_reduce(t, [](TbRatio x, TbRatio y) {
        TbRatio u = vi_less(y.th, y.row, x.th, x.row) ? y : x;
        u.dmax = fmax(x.dmax, y.dmax);
        return u;
    }, sh);
}

// start of a call: rc[] and xb[] from the tableau, pricing partials of the carried row
__global__ __launch_bounds__(TB_THREADS) void k_tb_boot(TbArgs a)
{
    __shared__ TbCand sh_price[TB_THREADS / 64];
    const lpx_tableau &s = a.s;
    const int64_t g = (int64_t)blockIdx.x * TB_THREADS + threadIdx.x;
    if (g == 0) {
        *a.b.cnt = 0;
        a.b.tsel[0] = 0; // a call starts (and ends) with the tableau in T
    }
    if (s.ctrl->status != LPX_RUNNING) return;
    if (g < s.m) a.b.xb[g] = s.T[g * s.ld + s.n];
    if ((int64_t)blockIdx.x >= a.nprice) return;
    TbCand c{LPX_INF, LPX_NOIDX, -1};
    if (g < s.n) {
        const double rc = s.T[s.m * s.ld + g];
        a.b.rc[g] = rc;
        tb_price_one(s.pricing, rc, t_var(s, g), g, c);
    }
    c = tb_price_reduce(s.pricing, c, sh_price);
    if (threadIdx.x == 0) {
        a.price.v0[blockIdx.x] = c.mn;
        a.price.i0[blockIdx.x] = c.var;
        a.price.i1[blockIdx.x] = c.slot;
    }
}

// entry i of column `slot` after the P pending updates: the column an update rewrote is a wave-uniform case, the pivot
// row of an update is the same fma on an accumulator of -0.0 (fma(e, R, -0.0) == e * R bit for bit)
// e[k] belongs to update U0 + k (the pipelined chain walks its 2 KB pairs in chunks of TB_CH: registers for two chunks)
template <int NE, int U0 = 0>
__device__ __forceinline__ double tb_column_chain(double di, int P, int slot, int i, int sv, int pv, const double (&e)[NE], double rv,
                                                  bool cond)
{
#pragma unroll
    for (int k = 0; k < NE; ++k) {
        const int u = U0 + k;
        if (u < P) {
            const int pu = __builtin_amdgcn_readlane(pv, u);
            if (slot == __builtin_amdgcn_readlane(sv, u)) di = cond ? e[k] : ((i == pu) ? 1.0 : 0.0);
            else di = fma(e[k], tf_readlane_f64(rv, u), (i == pu) ? -0.0 : di);
        }
    }
    return di;
}
#define TB_CH 16
// buf[k] = base[(U0 + k) * stride] for the updates below P; `on`: this lane has an element
template <int U0>
__device__ __forceinline__ void tb_ld_chunk(double (&buf)[TB_CH], const double *base, int64_t stride, int P, bool on)
{
#pragma unroll
    for (int k = 0; k < TB_CH; ++k) buf[k] = (U0 + k < P && on) ? base[(U0 + k) * stride] : 0.0;
}
// loads issued after this point stay after it (the compiler would hoist a chunk's loads above the chain before it and
// pay for the registers: two waves per SIMD instead of four)
#define TB_LOADS_STAY_BELOW() asm volatile("" ::: "memory")

// entering decision, column of the tableau as the u pending updates leave it, ratio-test partials
template <int MODE, bool OLD = false>
__global__ __launch_bounds__(TB_THREADS) void k_tb_column(TbArgs a)
{
    __shared__ TbCand sh_price[TB_THREADS / 64];
    __shared__ TbRatio sh_ratio[TB_THREADS / 64];
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
    // the block whose pass is in flight (pipelined chain): its pairs are complete
    const int PO = OLD ? a.uo : 0;
    double e[OLD ? 1 : TB_KMAX];
    double ea[OLD ? TB_CH : 1], eb[OLD ? TB_CH : 1]; // OLD: the 2 KB multipliers of a row pass through two chunks of registers
    const double *Eti = a.b.Et + (inm ? i : 0); // coalesced over the rows of the workgroup
    const double *Eto = OLD ? a.o.Et + (inm ? i : 0) : nullptr;
    if constexpr (OLD) {
        tb_ld_chunk<0>(ea, Eto, a.o.lde, PO, true);
        tb_ld_chunk<TB_CH>(eb, Eto, a.o.lde, PO, true);
    } else if (MODE != TB_SELECT) {
#pragma unroll
        for (int u = 0; u < TB_KMAX; ++u) e[u] = (u < P) ? Eti[u * a.b.lde] : 0.0;
    }
    const int pvo = (OLD && l32 < PO) ? ((const int *)a.o.prow)[2 * l32] : -2;
    const int svo = (OLD && l32 < PO) ? ((const int *)a.o.pslot)[2 * l32] : -2;
    // entering variable (primal_solvers.py:135-142) and its column slot; the scan of the pricing partials goes out with the
    // other loads (valid addresses whatever the status)
    TbCand c{LPX_INF, LPX_NOIDX, -1};
    int64_t win = 0;
    if (MODE != TB_SELECT) {
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
    }
    if (status != LPX_RUNNING) return;
    LPX_TB_STAMP(0, 1);
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
        LPX_TB_STAMP(0, 2);
        c = tb_price_reduce(s.pricing, c, sh_price);
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
            const double rvo = (OLD && l32 < PO) ? a.o.R[l32 * a.o.ldr + slot] : 0.0;
            di = inm ? s.T[i * s.ld + slot] : 0.0;
            if constexpr (OLD) { // the pipelined chain's 2 KB multipliers pass through two chunks of registers (see below)
                di = tb_column_chain<TB_CH, 0>(di, PO, (int)slot, (int)i, svo, pvo, ea, rvo, cond);
                TB_LOADS_STAY_BELOW();
                tb_ld_chunk<0>(ea, Eti, a.b.lde, P, true);
                di = tb_column_chain<TB_CH, TB_CH>(di, PO, (int)slot, (int)i, svo, pvo, eb, rvo, cond);
                TB_LOADS_STAY_BELOW();
                tb_ld_chunk<TB_CH>(eb, Eti, a.b.lde, P, true);
                di = tb_column_chain<TB_CH, 0>(di, P, (int)slot, (int)i, sv, pv, ea, rv, cond);
                di = tb_column_chain<TB_CH, TB_CH>(di, P, (int)slot, (int)i, sv, pv, eb, rv, cond);
            } else {
                di = tb_column_chain(di, P, (int)slot, (int)i, sv, pv, e, rv, cond);
            }
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
        const double rvo = (OLD && l32 < PO) ? a.o.R[l32 * a.o.ldr + slot] : 0.0;
        di = inm ? s.T[i * s.ld + slot] : 0.0;
        if constexpr (OLD) {
            di = tb_column_chain<TB_CH, 0>(di, PO, (int)slot, (int)i, svo, pvo, ea, rvo, cond);
            TB_LOADS_STAY_BELOW();
            tb_ld_chunk<0>(ea, Eti, a.b.lde, P, true);
            di = tb_column_chain<TB_CH, TB_CH>(di, PO, (int)slot, (int)i, svo, pvo, eb, rvo, cond);
            TB_LOADS_STAY_BELOW();
            tb_ld_chunk<TB_CH>(eb, Eti, a.b.lde, P, true);
            di = tb_column_chain<TB_CH, 0>(di, P, (int)slot, (int)i, sv, pv, ea, rv, cond);
            di = tb_column_chain<TB_CH, TB_CH>(di, P, (int)slot, (int)i, sv, pv, eb, rv, cond);
        } else {
            di = tb_column_chain(di, P, (int)slot, (int)i, sv, pv, e, rv, cond);
        }
        if (inm) a.d[i] = di; // d[m] = the entering variable's reduced cost
    }
    // ratio test (primal_solvers.py:149-167): first minimum over this workgroup's rows, with what the winner needs
    TbRatio t{in ? lpx_psd(xi, di, f32) : LPX_INF, in ? i : LPX_NOIDX, in ? di : -LPX_INF, di, xi, bvi};
    LPX_TB_STAMP(0, 4);
    t = tb_ratio_reduce(t, sh_ratio);
    LPX_TB_STAMP(0, 5);
    if (tid == 0) {
        const bool none = t.row == LPX_NOIDX;
        a.ratio.v0[blockIdx.x] = t.th;
        a.ratio.i0[blockIdx.x] = t.row;
        a.ratio.v1[blockIdx.x] = t.dmax;
        a.ratio.i1[blockIdx.x] = none ? -1 : t.bv;
        a.b.dbest[blockIdx.x] = none ? 0.0 : t.dv;
        a.b.xbest[blockIdx.x] = none ? 0.0 : t.xv;
    }
    LPX_TB_STAMP(0, 6);
    LPX_TB_WGEND(0);
}

// leaving row, eta column, x_B, the pivot row as the pending updates leave it, the reduced-cost row and its pricing
// partials, bookkeeping
// row r of the tableau through the P pending updates of one block: lane l of every wave holds update l's pivot row (pv),
// rewritten column (sv) and multiplier of row r (erv); ru[u] = R[u][g]
template <int NE, int U0 = 0>
__device__ __forceinline__ double tb_row_chain(double rowp, int P, int r, int g, int pv, int sv, double erv, const double (&ru)[NE], bool cond)
{
#pragma unroll
    for (int k = 0; k < NE; ++k) {
        const int u = U0 + k;
        if (u < P) {
            const double er = tf_readlane_f64(erv, u);
            const bool isr = r == __builtin_amdgcn_readlane(pv, u); // wave-uniform
            const double v = isr ? er * ru[k] : fma(er, ru[k], rowp);
            rowp = (g == __builtin_amdgcn_readlane(sv, u)) ? (cond ? er : (isr ? 1.0 : 0.0)) : v;
        }
    }
    return rowp;
}

template <bool OLD = false>
__global__ __launch_bounds__(TB_THREADS) void k_tb_prepare(TbArgs a)
{
    __shared__ TbCand sh_price[TB_THREADS / 64];
    __shared__ TbRatio sh_ratio[TB_THREADS / 64];
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
    const int PO = OLD ? a.uo : 0; // the block whose pass is in flight (pipelined chain)
    double ru[OLD ? 1 : TB_KMAX];
    double ra[OLD ? TB_CH : 1], rb[OLD ? TB_CH : 1]; // OLD: the 2 KB pivot-row entries of a column pass through two chunks
    const double *Rg = a.b.R + (incol ? g : 0), *Rog = OLD ? a.o.R + (incol ? g : 0) : nullptr;
    if constexpr (OLD) {
        tb_ld_chunk<0>(ra, Rog, a.o.ldr, PO, incol);
        tb_ld_chunk<TB_CH>(rb, Rog, a.o.ldr, PO, incol);
    } else {
#pragma unroll
        for (int u = 0; u < TB_KMAX; ++u) ru[u] = (u < P && incol) ? a.b.R[u * a.b.ldr + g] : 0.0;
    }
    const double rc_old = (g < n) ? a.b.rc[g] : 0.0;
    const double dg = (g <= m) ? a.d[g] : 0.0, dm = a.d[m];
    const double xg = (g < m) ? a.b.xb[g] : 0.0;
    const int64_t varg = (g < n) ? t_var(s, g) : LPX_NOIDX;
    const int l32 = tid & 31; // lane l holds pending update l's pivot row / rewritten column / multiplier of row r
    const int pv = (l32 < P) ? ((const int *)a.b.prow)[2 * l32] : -2;
    const int sv = (l32 < P) ? ((const int *)a.b.pslot)[2 * l32] : -2;
    const int pvo = (OLD && l32 < PO) ? ((const int *)a.o.prow)[2 * l32] : -2;
    const int svo = (OLD && l32 < PO) ? ((const int *)a.o.pslot)[2 * l32] : -2;
    const int64_t pivots0 = ctrl->pivots; // used by the bookkeeping thread only
    // the ratio partials too (their addresses are valid whatever the status): leaving row = first minimum ratio over the
    // workgroups' partials (primal_solvers.py:166, np.argmin), each with its payload
    TbRatio best{LPX_INF, LPX_NOIDX, -LPX_INF, 0.0, 0.0, -1};
    for (int64_t k = tid; k < a.nratio; k += TB_THREADS) {
        const TbRatio t{a.ratio.v0[k], a.ratio.i0[k], a.ratio.v1[k], a.b.dbest[k], a.b.xbest[k], a.ratio.i1[k]};
        const double dm = fmax(best.dmax, t.dmax);
        if (vi_less(t.th, t.row, best.th, best.row)) best = t;
        best.dmax = dm;
    }
    if (status != LPX_RUNNING) return;
    LPX_TB_STAMP(1, 1);
    LPX_TB_STAMP(1, 2);
    const TbRatio w = tb_ratio_reduce(best, sh_ratio);
    LPX_TB_STAMP(1, 3);
    const double dmax = w.dmax;
    if (dmax <= 0.0) { // primal_solvers.py:144-147
        if (g == 0) {
            ctrl->aux_val = dmax;
            ctrl->status = LPX_PRIMAL_UNBOUNDED;
        }
        return;
    }
    const int64_t r = w.row, leaving = w.bv;
    const double dr = w.dv, xr = w.xv;
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
        const double ervo = (OLD && l32 < PO) ? a.o.E[r * TB_KMAX + l32] : 0.0;
        rowp = incol ? s.T[r * s.ld + g] : 0.0;
        if constexpr (OLD) {
            rowp = tb_row_chain<TB_CH, 0>(rowp, PO, (int)r, (int)g, pvo, svo, ervo, ra, cond);
            TB_LOADS_STAY_BELOW();
            tb_ld_chunk<0>(ra, Rg, a.b.ldr, P, incol);
            rowp = tb_row_chain<TB_CH, TB_CH>(rowp, PO, (int)r, (int)g, pvo, svo, ervo, rb, cond);
            TB_LOADS_STAY_BELOW();
            tb_ld_chunk<TB_CH>(rb, Rg, a.b.ldr, P, incol);
            rowp = tb_row_chain<TB_CH, 0>(rowp, P, (int)r, (int)g, pv, sv, erv, ra, cond);
            rowp = tb_row_chain<TB_CH, TB_CH>(rowp, P, (int)r, (int)g, pv, sv, erv, rb, cond);
        } else {
            rowp = tb_row_chain(rowp, P, (int)r, (int)g, pv, sv, erv, ru, cond);
        }
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
        c = tb_price_reduce(s.pricing, c, sh_price);
        if (tid == 0) {
            a.price.v0[blockIdx.x] = c.mn;
            a.price.i0[blockIdx.x] = c.var;
            a.price.i1[blockIdx.x] = c.slot;
        }
    }
    LPX_TB_STAMP(1, 7);
    if (g == 0) { // bookkeeping of the basis change (primal_solvers.py:96-108, :261)
        s.basis[r] = j;
        if (s.slot_var) { // t_swap_columns without its look-up: the entering variable's slot is `slot` (-1: another rank's)
            if (slot >= 0) {
                s.slot_var[slot] = leaving;
                s.var_slot[leaving] = slot;
            }
            s.var_slot[j] = -1;
            ctrl->enter_slot = slot;
        } else {
            t_swap_columns(s, j, leaving);
        }
        if (s.log && pivots0 < s.log_cap) {
            s.log[3 * pivots0 + 0] = r;
            s.log[3 * pivots0 + 1] = j;
            s.log[3 * pivots0 + 2] = leaving;
        }
        ctrl->pivots = pivots0 + 1;
        ctrl->leave_r = r;
        ctrl->leave_var = leaving;
        ctrl->ratio_val = w.th;
        ctrl->aux_val = dmax;
        ctrl->flip = 0;
        ctrl->f32_phase = 0;
        a.b.prow[P] = r;
        a.b.pslot[P] = slot;
        a.b.rowflag[r] |= 1u << P;
        *a.b.cnt = P + 1;
        a.b.tsel[0] = a.dstid; // where the tableau lives once this block's pass has run (0 unless the chain is pipelined)
#ifdef LPX_TB_STAMPS
        g_tb_stamp[1][8] = (long long)wall_clock64();
        if (ctrl->pivots >= 96 && ctrl->pivots < 160 && (P == 0 || P == 15 || P == 31)) {
            const long long *c0 = g_tb_stamp[0], *p0 = g_tb_stamp[1];
            long long cmax = 0;
            for (int64_t k = 0; k < a.nratio && k < 4096; ++k) cmax = g_tb_wgend[0][k] > cmax ? g_tb_wgend[0][k] : cmax;
            printf("P=%d column tail beyond workgroup 0: %lld, boundary after the last workgroup: %lld | ", P, cmax - c0[6], p0[0] - cmax);
            printf("P=%d column: status %lld | partials %lld | price reduce %lld | column+chain %lld | ratio reduce %lld | end %lld || "
                   "gap %lld || prepare: status %lld | partials %lld | ratio reduce %lld | payload %lld | eta %lld | row chain %lld | "
                   "rc+price reduce %lld | books %lld\n", P, c0[1] - c0[0], c0[2] - c0[1], c0[3] - c0[2], c0[4] - c0[3], c0[5] - c0[4],
                   c0[6] - c0[5], p0[0] - c0[6], p0[1] - p0[0], p0[2] - p0[1], p0[3] - p0[2], p0[4] - p0[3], p0[5] - p0[4],
                   p0[6] - p0[5], p0[7] - p0[6], p0[8] - p0[7]);
        }
#endif
    }
}


// ---------------------------------------------------------------------------------------------------------
// The same chain for the DUAL rules (dual_solvers.py:47-108) -- the decisions come in the dual's order:
//     k_tbd_row(P):    leaving row r = first x_B < 0 from the partials over xb[] (optimality: min x_B); row r of T0 with
//                      the P pending updates applied -> R[P][.] (the dual row AND this update's old pivot row); ratio test
//                      theta_j = (-rc_j) / row_j over row_j < 0 against rc[]: per-workgroup partials (first argmin by
//                      variable id) that carry the winner's slot, reduced cost and row entry (the pivot element)
//     k_tbd_column(P): entering variable from the partials, dual unboundedness (min(row) >= 0); column of T0 with the P
//                      pending updates applied; eta column -> E[.][P]; xb[] and rc[] updated eagerly; partials over the
//                      new xb[] for the next pivot; bookkeeping
// and the same pass (k_tb_flush2) after a block.  As in the primal chain no launch reads a word it writes: x_B[r] comes
// from R[P][n] (the row kernel formed it), the entering reduced cost and the pivot element ride in the partials.
// ---------------------------------------------------------------------------------------------------------
struct TbXb {
    double mn;
    int64_t first;
};
__device__ __forceinline__ TbXb tb_xb_reduce(TbXb t, TbXb *sh)
{
    return tb_block_reduce(t, [](TbXb x, TbXb y) { return TbXb{fmin(x.mn, y.mn), y.first < x.first ? y.first : x.first}; }, sh);
}
struct TbDual {
    double th;
    int64_t var, slot;
    double rowmin, rcv, drv;
};
__device__ __forceinline__ TbDual tb_dual_reduce(TbDual t, TbDual *sh)
{
    return tb_block_reduce(t, [](TbDual x, TbDual y) {
        TbDual u = vi_less(y.th, y.var, x.th, x.var) ? y : x;
        u.rowmin = fmin(x.rowmin, y.rowmin);
        return u;
    }, sh);
}

// start of a call: rc[] and xb[] from the tableau, partials over x_B
__global__ __launch_bounds__(TB_THREADS) void k_tbd_boot(TbArgs a)
{
    __shared__ TbXb sh_xb[TB_THREADS / 64];
    const lpx_tableau &s = a.s;
    const int64_t g = (int64_t)blockIdx.x * TB_THREADS + threadIdx.x;
    if (g == 0) *a.b.cnt = 0;
    if (s.ctrl->status != LPX_RUNNING) return;
    if (g < s.n) a.b.rc[g] = s.T[s.m * s.ld + g];
    if ((int64_t)blockIdx.x >= a.nratio) return;
    TbXb t{LPX_INF, LPX_NOIDX};
    if (g < s.m) {
        const double v = s.T[g * s.ld + s.n];
        a.b.xb[g] = v;
        t = TbXb{v, v < 0.0 ? g : LPX_NOIDX};
    }
    t = tb_xb_reduce(t, sh_xb);
    if (threadIdx.x == 0) {
        a.ratio.v0[blockIdx.x] = t.mn;
        a.ratio.i0[blockIdx.x] = t.first;
    }
}

__global__ __launch_bounds__(TB_THREADS) void k_tbd_row(TbArgs a)
{
    __shared__ TbXb sh_xb[TB_THREADS / 64];
    __shared__ TbDual sh_du[TB_THREADS / 64];
    const lpx_tableau &s = a.s;
    lpx_ctrl *ctrl = s.ctrl;
    const int tid = threadIdx.x;
    const int64_t g = (int64_t)blockIdx.x * TB_THREADS + tid;
    const int P = a.u;
    const int64_t m = s.m, n = s.n, ncopy = lpx_roundup_dev(n + 1, 2);
    if (P == 0) { // a new block: nothing pending (also after the loop has ended: the flush then applies nothing)
        if (g <= m) a.b.rowflag[g] = 0u;
        if (g == 0) *a.b.cnt = 0;
    }
    // loads whose addresses depend on nothing decided in this launch
    const int32_t status = ctrl->status;
    const bool f32 = ctrl->f32_phase != 0;
    const bool incol = g < ncopy, cond = s.slot_var != nullptr;
    double ru[TB_KMAX];
#pragma unroll
    for (int u = 0; u < TB_KMAX; ++u) ru[u] = (u < P && incol) ? a.b.R[u * a.b.ldr + g] : 0.0;
    const double rcg = (g < n) ? a.b.rc[g] : 0.0;
    const int64_t varg = (g < n) ? t_var(s, g) : LPX_NOIDX;
    const bool basic = (g < n && !cond) ? s.var_state[g] == LPX_VS_BASIC : false; // dual_solvers.py:38-39
    const int l32 = tid & 31;
    const int pv = (l32 < P) ? ((const int *)a.b.prow)[2 * l32] : -2;
    const int sv = (l32 < P) ? ((const int *)a.b.pslot)[2 * l32] : -2;
    if (status != LPX_RUNNING) return;
    // leaving row: first x_B < 0 (dual_solvers.py:47-54)
    TbXb xp{LPX_INF, LPX_NOIDX};
    for (int64_t k = tid; k < a.nratio; k += TB_THREADS) {
        xp.mn = fmin(xp.mn, a.ratio.v0[k]);
        const int64_t f = a.ratio.i0[k];
        xp.first = f < xp.first ? f : xp.first;
    }
    xp = tb_xb_reduce(xp, sh_xb);
    const bool opt = lpx_is_optimal_min(xp.mn);
    const int64_t r = xp.first;
    if (g == 0) {
        ctrl->counter += 1;
        ctrl->price_val = xp.mn;
        if (opt) ctrl->status = LPX_OPTIMAL;
        else ctrl->leave_r = r;
    }
    if (opt) return;
    // row r of the tableau as the P pending updates leave it
    double rowp;
    {
        const double erv = (l32 < P) ? a.b.E[r * TB_KMAX + l32] : 0.0;
        rowp = incol ? s.T[r * s.ld + g] : 0.0;
#pragma unroll
        for (int u = 0; u < TB_KMAX; ++u) {
            if (u < P) {
                const double er = tf_readlane_f64(erv, u);
                const bool isr = (int)r == __builtin_amdgcn_readlane(pv, u); // wave-uniform
                const double v = isr ? er * ru[u] : fma(er, ru[u], rowp);
                rowp = ((int)g == __builtin_amdgcn_readlane(sv, u)) ? (cond ? er : (isr ? 1.0 : 0.0)) : v;
            }
        }
        if (incol) a.b.R[P * a.b.ldr + g] = rowp;
    }
    // ratio test over the columns (dual_solvers.py:61-80, utils.py:12-17)
    if ((int64_t)blockIdx.x < a.nprice) {
        TbDual c{LPX_INF, LPX_NOIDX, -1, LPX_INF, 0.0, 0.0};
        if (g < n) {
            const double row = basic ? 0.0 : rowp;
            c = TbDual{lpx_dsd(rcg, row, f32), varg, g, row, rcg, rowp};
        }
        c = tb_dual_reduce(c, sh_du);
        if (tid == 0) {
            a.price.v0[blockIdx.x] = c.th;
            a.price.i0[blockIdx.x] = c.var;
            a.price.i1[blockIdx.x] = c.slot;
            a.price.v1[blockIdx.x] = c.rowmin;
            a.b.dbest[blockIdx.x] = c.drv;
            a.b.xbest[blockIdx.x] = c.rcv;
        }
    }
}

// MODE as for k_tb_column: TB_LOCAL (unpartitioned), TB_PROPOSE (this rank's best column -> sendbuf: [theta, variable,
// column]), TB_SELECT (first argmin of theta over the ranks' proposals -- lowest variable id on ties, np.argmin's first
// minimum whichever rank holds it; no candidate anywhere is exactly min(row) >= 0 -- then the column from the winner's message)
template <int MODE>
__global__ __launch_bounds__(TB_THREADS) void k_tbd_column(TbArgs a)
{
    __shared__ TbXb sh_xb[TB_THREADS / 64];
    __shared__ TbDual sh_du[TB_THREADS / 64];
    const lpx_tableau &s = a.s;
    lpx_ctrl *ctrl = s.ctrl;
    const int tid = threadIdx.x;
    const int64_t g = (int64_t)blockIdx.x * TB_THREADS + tid;
    const int P = a.u;
    const int64_t m = s.m, n = s.n;
    // loads whose addresses depend on nothing decided in this launch
    const int32_t status = ctrl->status;
    const int64_t r = ctrl->leave_r;
    const bool inm = g <= m, cond = s.slot_var != nullptr;
    double e[TB_KMAX];
    {
        const double *Eti = a.b.Et + (inm ? g : 0);
#pragma unroll
        for (int u = 0; u < TB_KMAX; ++u) e[u] = (u < P) ? Eti[u * a.b.lde] : 0.0;
    }
    const double xg = (g < m) ? a.b.xb[g] : 0.0;
    const double rcg = (g < n) ? a.b.rc[g] : 0.0;
    const double rpg = (g < n) ? a.b.R[P * a.b.ldr + g] : 0.0; // row r as the pending updates leave it (k_tbd_row)
    const double xr = a.b.R[P * a.b.ldr + n];                  // its x_B entry: x_B[r] before this pivot
    const int l32 = tid & 31;
    const int pv = (l32 < P) ? ((const int *)a.b.prow)[2 * l32] : -2;
    const int sv = (l32 < P) ? ((const int *)a.b.pslot)[2 * l32] : -2;
    if (status != LPX_RUNNING) return;
    // entering variable: first argmin of theta by variable id; dual unboundedness (dual_solvers.py:41-44, :56-59)
    TbDual w{LPX_INF, LPX_NOIDX, -1, LPX_INF, 0.0, 0.0};
    int64_t win = 0;
    if (MODE == TB_SELECT) {
        for (int64_t q = 0; q < a.nranks; ++q) { // a handful of ranks: every thread scans them
            const double v = a.recv[q * a.msg];
            const int64_t idx = (int64_t)a.recv[q * a.msg + 1];
            if (vi_less(v, idx, w.th, w.var)) { w.th = v; w.var = idx; win = q; }
        }
        if (w.th == LPX_INF) {
            if (g == 0) {
                ctrl->aux_val = 0.0;
                ctrl->status = LPX_DUAL_UNBOUNDED;
            }
            return;
        }
        const double *col = a.recv + win * a.msg + 2;
        w.slot = t_slot(s, w.var); // -1: the column lives on another rank
        w.drv = col[r];
        w.rcv = col[m];
        w.rowmin = -1.0;
    } else {
        TbDual best{LPX_INF, LPX_NOIDX, -1, LPX_INF, 0.0, 0.0};
        for (int64_t k = tid; k < a.nprice; k += TB_THREADS) {
            const TbDual t{a.price.v0[k], a.price.i0[k], a.price.i1[k], a.price.v1[k], a.b.xbest[k], a.b.dbest[k]};
            const double rm = fmin(best.rowmin, t.rowmin);
            if (vi_less(t.th, t.var, best.th, best.var)) best = t;
            best.rowmin = rm;
        }
        w = tb_dual_reduce(best, sh_du);
        if (MODE == TB_LOCAL && w.rowmin >= 0.0) {
            if (g == 0) {
                ctrl->aux_val = w.rowmin;
                ctrl->status = LPX_DUAL_UNBOUNDED;
            }
            return;
        }
    }
    const int64_t j = w.var, slot = w.slot;
    const double dr = w.drv, rcv = w.rcv;
    // the entering column as the pending updates leave it; eta column (primal_solvers.py:217-245); x_B (:251-253)
    double di;
    if (MODE == TB_SELECT) {
        di = inm ? a.recv[win * a.msg + 2 + g] : 0.0;
    } else {
        const double rv = (l32 < P) ? a.b.R[l32 * a.b.ldr + slot] : 0.0;
        di = inm ? s.T[g * s.ld + slot] : 0.0;
        di = tb_column_chain(di, P, (int)slot, (int)g, sv, pv, e, rv, cond);
    }
    if (MODE == TB_PROPOSE) {
        if (g == 0) {
            a.sendbuf[0] = w.th;
            a.sendbuf[1] = (double)j; // exact: ids < 2^53
        }
        if (inm) a.sendbuf[2 + g] = di;
        return;
    }
    double nx = LPX_INF;
    if (inm) {
        a.d[g] = di;
        const double eg = (g == r) ? 1.0 / dr : (-di) / dr; // e[m] eliminates rc_j
        a.b.E[g * TB_KMAX + P] = eg;
        a.b.Et[P * a.b.lde + g] = eg;
        if (g < m) {
            nx = (g == r) ? eg * xr : fma(eg, xr, xg);
            a.b.xb[g] = nx;
        }
    }
    if ((int64_t)blockIdx.x < a.nratio) { // partials over the new x_B: the next pivot's leaving row
        TbXb t{nx, (g < m && nx < 0.0) ? g : LPX_NOIDX};
        t = tb_xb_reduce(t, sh_xb);
        if (tid == 0) {
            a.ratio.v0[blockIdx.x] = t.mn;
            a.ratio.i0[blockIdx.x] = t.first;
        }
    }
    if (g < n) { // the reduced-cost row after this pivot
        const double em = (-rcv) / dr;
        a.b.rc[g] = (g == slot) ? (cond ? em : 0.0) : fma(em, rpg, rcg);
    }
    if (g == 0) { // bookkeeping (td_finish)
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
        ctrl->ratio_val = w.th;
        ctrl->aux_val = w.rowmin;
        ctrl->flip = 0;
        ctrl->f32_phase = 0;
        a.b.prow[P] = r;
        a.b.pslot[P] = slot;
        a.b.rowflag[r] |= 1u << P;
        *a.b.cnt = P + 1;
    }
}
