// lpx_tableau_small.hip -- the deferred-update chain for LAUNCH-BOUND tableaus (BASELINE config 2: 1000 x 2000).
//
// At 1001 x 1001 a pass over the tableau costs 2-3 us and every kernel boundary as much again: the single-launch chain
// (k_t_fused) pays one boundary per pivot (6.7 us per pivot, 148 k pivots/s; this chain: 4.8 us, 207 k).  Here ONE launch carries a whole block of
// TS_KB pivots, in two roles (the pipelined chain of lpx_tableau_blocked.hpp folded into a kernel):
//   * workgroup 0 DECIDES block k+1 -- all of its TS_KB pivots, one after the other, with no kernel boundary and no other
//     workgroup to wait for: thread t owns row t and column t of the tableau "as the pending updates leave it", keeps its
//     row's multipliers e_u[t] and its column's pivot-row entries R_u[t] of both blocks IN REGISTERS, and everything a pivot
//     exchanges (pricing, ratio test, the pivot row's multipliers, the entering column's row entries) goes through LDS:
//     two gathers from the stale tableau (the entering column, the pivot row) are the only memory round trips of a pivot;
//   * the other workgroups are the PASS of block k: T_old -> T_new with block k's pairs (ts_pass below: tb_apply's
//     expressions, lpx_tb_flush.hpp).
// Both read the same stale buffer, neither reads what the other writes, and the next launch starts when both are done:
// the launch boundary is the only grid-wide synchronisation of TS_KB pivots.  Same expressions in the same order as
// k_tb_column / k_tb_prepare and hence as one update per pivot (primal_solvers.py:135-167, :217-253): bit-identical.
// Shapes: m + 1 <= 1024 rows and n + 1 <= 1024 carried columns, x_B included (one workgroup holds a row and a column per thread).
#include "lpx_tb_flush.hpp"
#include "lpx_tableau_dev.hpp"
#include "lpx_tableau_small_decl.hpp"

#define TS_INTERNAL (-9) /* control-block status of an index that failed its range check (never seen; not an LPX_* status) */
#define TS_WAVES (TS_THREADS / 64)
// -DLPX_TS_STAMPS: diagnostic build, thread 0 of the deciding workgroup prints where a block's time goes (10 ns ticks)
#ifdef LPX_TS_STAMPS
__device__ long long g_ts_entry, g_ts_exit;
#define TS_STAMP(n) do { if (threadIdx.x == 0) { const long long now_ = (long long)wall_clock64(); ts_acc[n] += now_ - ts_last; ts_last = now_; } } while (0)
#else
#define TS_STAMP(n) do { } while (0)
#endif

// One CU issues every instruction of the deciding workgroup 16 times (16 waves, four to a SIMD): a pivot's time is its
// two gathers plus 16 x 4 cycles per instruction of the thread program, so the program is written for its instruction count:
//   * what is the same for every thread (the pivot's own row / slot, which pending pairs concern this wave) lives in
//     scalar registers, and a test on it is a scalar branch; the pending pairs' rows and slots themselves sit in ONE
//     vector register (lane u: pair u), read by ballot (does any pair rewrite this column?) or v_readlane (rare paths);
//   * the one thread of 1024 a select singles out (the pivot row in the column chain, the rewritten slot in the row chain)
//     costs its WAVE a select and the other fifteen waves a scalar compare;
//   * a reduction carries what the DECISION needs and nothing else: Bland's rule the smallest negative-cost variable id
//     (32 bits) and whether any cost is below -atol; the ratio test the ratio, and whether any entry is positive.  The
//     values only the control block reports (the smallest reduced cost, the column's maximum) are reduced once, after the
//     block's last pivot, from the operands kept for it;
//   * records cross the workgroup as wave partials that every wave combines in one 16-lane DPP reduction;
//   * nothing is stored to memory inside the block: the pairs, the basis and the log leave from registers / LDS after it,
//     to addresses the first lines parked in LDS (held in scalar registers across the block they would crowd out the pairs'
//     rows and slots, and a spilled scalar is a VALU lane move per use).
#define TS_KEEP_BRANCH() asm volatile("" ::: "memory") /* keeps a wave-uniform branch a branch (no if-conversion) */

struct TsCand { // pricing, Dantzig's rule: the smallest reduced cost, its variable, its slot
    double mn;
    int var, slot;
};
struct TsBland { // pricing, Bland's rule: the smallest variable id of negative reduced cost, its slot; any cost below -atol
    int var, slot, live, pad;
};
struct TsRatio { // a wave's ratio-test winner: ratio, entering-column entry, x_B, row, basic variable
    double th, dv, xv;
    int row, bv, pos, pad;
};
struct TsMinMax { // what the control block reports: the smallest reduced cost priced, the largest entry of the column tested
    double mn, mx;
};
#define TS_NOVAR 0x7fffffff
__device__ __forceinline__ bool ts_less(double v, int i, double bv, int bi) { return v < bv || (v == bv && i < bi); }
// a pending pair's pivot row and rewritten slot in one scalar register (both below 1024)
__device__ __forceinline__ int ts_pack(int row, int slot) { return (row << 16) | slot; }
#define TS_ROW(pk) ((pk) >> 16)
#define TS_SLOT(pk) ((pk) & 0xffff)
#define TS_NOPAIR ts_pack(0x7000, 0x7000)

// The two update chains of a pivot.  `pkv`: lane u of every wave holds pending pair u (0 .. KB-1: the old block's, KB ..
// 2 KB - 1: this block's); `wrow` / `wslot`: the pairs whose pivot row / rewritten slot belongs to a thread of THIS wave.
// Almost every pivot enters a column no pending pair has rewritten and leaves by a row no pending pair pivoted on: then
// (GENERAL == false) a step is one fma, plus a select in the one wave the pair's row / slot lives in.
// (`nsteps` is a constant wherever this is inlined: the pivot loop is unrolled)
template <int KB, int BASE, bool GENERAL>
__device__ __forceinline__ double ts_col_steps(int nsteps, double di, const double (&e)[KB], const double (&rk)[KB], int pkv,
                                               unsigned wrow, int slot, int t, int wave, bool cond)
{
#pragma unroll
    for (int u = 0; u < KB; ++u) {
        if (u >= nsteps) break;
        if (GENERAL) {
            const int pk = __builtin_amdgcn_readlane(pkv, BASE + u), pu = TS_ROW(pk);
            if (slot == TS_SLOT(pk)) {
                di = cond ? e[u] : ((t == pu) ? 1.0 : 0.0);
            } else {
                double acc = di;
                if ((pu >> 6) == wave) {
                    TS_KEEP_BRANCH();
                    acc = (t == pu) ? -0.0 : di;
                }
                di = fma(e[u], rk[u], acc);
            }
        } else {
            double acc = di;
            if (__builtin_expect((wrow >> (BASE + u)) & 1u, 0)) {
                TS_KEEP_BRANCH();
                acc = (t == TS_ROW(__builtin_amdgcn_readlane(pkv, BASE + u))) ? -0.0 : di;
            }
            di = fma(e[u], rk[u], acc);
        }
    }
    return di;
}
template <int KB, int BASE, bool GENERAL>
__device__ __forceinline__ double ts_row_steps(int nsteps, double rowp, const double (&er)[KB], const double (&rr)[KB], int pkv,
                                               unsigned wslot, int r, int t, int wave, bool cond)
{
#pragma unroll
    for (int u = 0; u < KB; ++u) {
        if (u >= nsteps) break;
        if (GENERAL) {
            const int pk = __builtin_amdgcn_readlane(pkv, BASE + u), su = TS_SLOT(pk);
            const bool isr = r == TS_ROW(pk);
            double acc = rowp;
            if (isr) acc = -0.0;
            rowp = fma(er[u], rr[u], acc);
            if ((su >> 6) == wave) {
                TS_KEEP_BRANCH();
                if (t == su) rowp = cond ? er[u] : (isr ? 1.0 : 0.0);
            }
        } else {
            rowp = fma(er[u], rr[u], rowp);
            if (__builtin_expect((wslot >> (BASE + u)) & 1u, 0)) {
                TS_KEEP_BRANCH();
                if (t == TS_SLOT(__builtin_amdgcn_readlane(pkv, BASE + u))) rowp = cond ? er[u] : 0.0;
            }
        }
    }
    return rowp;
}

// PO: pairs of the block whose pass runs beside this workgroup -- none (the first block of a call) or a whole block (a block
// that ends early ends the loop), so the old block's chain is compiled for its length
template <int KB, int PO>
__device__ __forceinline__ void ts_decide(const TsArgs &a)
{
    static_assert(TS_WAVES == 16 && KB % 2 == 0, "sixteen wave partials combine in one 16-lane row; pairs move two at a time");
    __shared__ __align__(16) TsCand sh_price[2][TS_WAVES];
    __shared__ __align__(16) TsBland sh_bland[2][TS_WAVES];
    __shared__ __align__(16) TsRatio sh_ratio[2][TS_WAVES];
    __shared__ __align__(16) double sh_rk[2 * KB]; // R_u[slot]: the entering column's entries of the pending pivot rows (old block, then new)
    __shared__ __align__(16) double sh_er[2 * KB]; // e_u[r]: the leaving row's multipliers of the pending updates
    __shared__ double sh_em[2];                    // the reduced-cost row's multiplier of this pivot (row m's eta entry)
    __shared__ __align__(16) TsMinMax sh_red[TS_WAVES]; // the closing reduction
    __shared__ int sh_piv[KB][4];                  // (row, slot, entering, leaving) of the block's pivots, for the closing bookkeeping
    __shared__ __align__(16) TsArgs sh_args;       // the launch record, for the closing stores
#ifdef LPX_TS_STAMPS
    const long long ts_entry = (long long)wall_clock64();
#endif
    const lpx_tableau &s = a.s;
    const int t = threadIdx.x;
    const int wave = __builtin_amdgcn_readfirstlane(t >> 6), lane = t & 63;
    const int m = (int)s.m, n = (int)s.n, ncopy = (int)lpx_roundup_dev(s.n + 1, 2), n_global = (int)s.n_global;
    const int64_t ld = s.ld;
    const bool cond = s.slot_var != nullptr;
    const bool dantzig = s.pricing == LPX_PRICE_DANTZIG;
    const bool inrow = t <= m, inx = t < m, incol = t < ncopy, inrc = t < n;
    // ---- everything whose address depends on nothing decided here, in one round trip
    if (t < (int)(sizeof(TsArgs) / 4)) ((int *)&sh_args)[t] = ((const int *)&a)[t];
    const int32_t status0 = s.ctrl->status;
    const bool f32 = s.ctrl->f32_phase != 0;
    double eo[KB], en[KB], ro[KB], rn[KB];
    int pkv = (PO > 0 && lane < PO) ? ts_pack((int)a.o.prow[lane], (int)a.o.pslot[lane]) : TS_NOPAIR;
    {
#pragma unroll
        for (int u = 0; u < KB; ++u) {
            eo[u] = (u < PO && inrow) ? a.o.Et[u * a.o.lde + t] : 0.0; // the pairs cross launches transposed: a row of 8 KB a pair
            ro[u] = (u < PO && incol) ? a.o.R[u * a.o.ldr + t] : 0.0;
            en[u] = 0.0;
            rn[u] = 0.0;
        }
    }
    double rc = inrc ? a.b.rc[t] : LPX_INF, xb = inx ? a.b.xb[t] : 0.0;
    int var = inrc ? (int)t_var(s, t) : TS_NOVAR;
    int bas = inx ? (int)s.basis[t] : -1;
    const double *Trow = s.T + (int64_t)t * ld, *Tcol = s.T + t;
    // a new block: nothing pending (also when the loop has ended: the pass of this block then applies nothing)
    if (inrow) a.b.rowflag[t] = 0u;
    if (t == 0) *a.b.cnt = 0;
    if (status0 != LPX_RUNNING) return;
    unsigned wrow = (unsigned)__ballot((TS_ROW(pkv) >> 6) == wave), wslot = (unsigned)__ballot((TS_SLOT(pkv) >> 6) == wave);
    int status = LPX_RUNNING;
    int last_j = -1;
    double last_ratio = 0.0;
    double rc_priced = LPX_INF, d_tested = -LPX_INF; // operands of the last pricing / ratio test, for the values the control block reports
    int done = 0, priced = 0;
    uint32_t myflag = 0u;
    bool mycol = false;
#ifdef LPX_TS_STAMPS
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    long long ts_acc[8] = {0, 0, 0, 0, 0, 0, 0, 0}, ts_last = (long long)wall_clock64();
    const long long ts_first = ts_last;
#endif
    // (every exit of the loop is taken by all threads or none; the tests are made scalar so that the compiler knows it too:
    // after a branch it takes for divergent it would keep the pending pairs' rows and slots in vector registers)
#pragma unroll
    for (int P = 0; P < KB; ++P) {
        if (P >= a.u) break;
        const int ph = P & 1;
        // ---- entering variable (primal_solvers.py:135-142): wave partials by DPP, one LDS exchange, 16 partials by DPP again
        int slot, j;
        bool optimal;
        rc_priced = rc;
        if (dantzig) {
            TsCand c{LPX_INF, TS_NOVAR, -1};
            if (inrc && ts_less(rc, var, c.mn, c.var)) c = TsCand{rc, var, t};
            c = wave_allreduce(c, [](TsCand x, TsCand y) { return ts_less(y.mn, y.var, x.mn, x.var) ? y : x; });
            if (lane == 0) sh_price[ph][wave] = c;
            __syncthreads();
            // the sixteen partials: smallest cost, smallest id among the waves that have it, and that wave's slot
            const TsCand *wc = &sh_price[ph][lane & (TS_WAVES - 1)];
            const double wmn = wc->mn;
            const int wvar = wc->var;
            const double mn = row16_allreduce(wmn, [](double x, double y) { return lpx_min_f64(x, y); });
            const int minvar = row16_allreduce(wmn == mn ? wvar : TS_NOVAR, [](int x, int y) { return y < x ? y : x; });
            const unsigned who = (unsigned)__ballot(wmn == mn && wvar == minvar) & 0xffffu;
            optimal = lpx_is_optimal_min(mn);
            slot = sh_price[ph][who ? __ffs(who) - 1 : 0].slot;
            j = minvar;
        } else {
            // the loop ends when min rc >= -atol (lpx_is_optimal_min), i.e. when no reduced cost lies below -atol
            const int myvar = (inrc && rc < 0.0) ? var : TS_NOVAR; // first negative index: smallest variable id (ids are distinct)
            TsBland c;
            c.var = wave_allreduce(myvar, [](int x, int y) { return y < x ? y : x; });
            c.slot = (wave << 6) + __builtin_amdgcn_readfirstlane(__ffsll((long long)__ballot(myvar == c.var)) - 1);
            c.live = __ballot(inrc && rc < -LPX_ATOL) != 0ull;
            c.pad = 0;
            if (lane == 0) sh_bland[ph][wave] = c;
            __syncthreads();
            // the sixteen partials: smallest id, the wave that has it (ids are distinct), its slot
            const TsBland *wb = &sh_bland[ph][lane & (TS_WAVES - 1)];
            const int wvar = wb->var;
            const int minvar = row16_allreduce(wvar, [](int x, int y) { return y < x ? y : x; });
            const unsigned who = (unsigned)__ballot(wvar == minvar) & 0xffffu;
            optimal = ((unsigned)__ballot(wb->live != 0) & 0xffffu) == 0u;
            slot = sh_bland[ph][who ? __ffs(who) - 1 : 0].slot;
            j = minvar;
        }
        TS_STAMP(0);
        priced = P + 1;
        if (__builtin_amdgcn_readfirstlane((int)optimal) != 0) {
            status = LPX_OPTIMAL;
            break;
        }
        slot = __builtin_amdgcn_readfirstlane(slot);
        j = __builtin_amdgcn_readfirstlane(j);
        if (slot < 0 || slot >= n || j < 0 || j >= n_global) { // cannot happen; an index that would fault is a status instead
            status = TS_INTERNAL;
            break;
        }
        // ---- the entering column as the pending updates leave it (tb_column_chain's expressions, old block first)
        double di = inrow ? Trow[slot] : 0.0;
        if (t == slot) {
#pragma unroll
            for (int u = 0; u < KB; ++u) sh_rk[u] = ro[u];
#pragma unroll
            for (int u = 0; u < P; ++u) sh_rk[KB + u] = rn[u];
        }
        __syncthreads();
#ifdef LPX_TS_STAMPS
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#endif
        TS_STAMP(1);
        {
            const bool general = __ballot(TS_SLOT(pkv) == slot) != 0ull; // the entering column is one a pending update rewrote
            if (PO > 0) {
                double rk[KB];
#pragma unroll
                for (int u = 0; u < KB; ++u) rk[u] = sh_rk[u];
                if (general) di = ts_col_steps<KB, 0, true>(PO, di, eo, rk, pkv, wrow, slot, t, wave, cond);
                else di = ts_col_steps<KB, 0, false>(PO, di, eo, rk, pkv, wrow, slot, t, wave, cond);
            }
            if (P > 0) {
                double rk[KB];
#pragma unroll
                for (int u = 0; u < KB; ++u) rk[u] = (u < P) ? sh_rk[KB + u] : 0.0;
                if (general) di = ts_col_steps<KB, KB, true>(P, di, en, rk, pkv, wrow, slot, t, wave, cond);
                else di = ts_col_steps<KB, KB, false>(P, di, en, rk, pkv, wrow, slot, t, wave, cond);
            }
        }
        TS_STAMP(2);
        // ---- ratio test (primal_solvers.py:149-167): first minimum, with what the winner needs.  Rows ascend with the lane
        // and with the wave: the first lane at the wave's minimum is the wave's np.argmin row (:166).  The loop ends when
        // max d <= 0 (:144-147), i.e. when no entry is positive
        d_tested = inx ? di : -LPX_INF;
        const double th = inx ? lpx_psd(xb, di, P == 0 && f32) : LPX_INF;
        const double thw = wave_allreduce(th, [](double x, double y) { return lpx_min_f64(x, y); });
        const int src = __builtin_amdgcn_readfirstlane(__ffsll((long long)__ballot(th == thw)) - 1);
        // the cross-lane reads stay OUTSIDE the lane-0 branch: under it the source lane is inactive and the read undefined
        const TsRatio wrec{thw, tf_readlane_f64(di, src), tf_readlane_f64(xb, src), (wave << 6) + src,
                           __builtin_amdgcn_readlane(bas, src), __ballot(inx && di > 0.0) != 0ull, 0};
        if (lane == 0) sh_ratio[ph][wave] = wrec;
        __syncthreads();
        // the sixteen partials: the smallest ratio, the FIRST wave that has it (rows ascend with the wave), its record
        const TsRatio *wr = &sh_ratio[ph][lane & (TS_WAVES - 1)];
        const double wth = wr->th;
        const double thmin = row16_allreduce(wth, [](double x, double y) { return lpx_min_f64(x, y); });
        const unsigned who = (unsigned)__ballot(wth == thmin) & 0xffffu;
        const bool anypos = ((unsigned)__ballot(wr->pos != 0) & 0xffffu) != 0u;
        TS_STAMP(3);
        if (!anypos) {
            status = LPX_PRIMAL_UNBOUNDED;
            last_j = j;
            break;
        }
        const TsRatio win = sh_ratio[ph][who ? __ffs(who) - 1 : 0];
        const int r = __builtin_amdgcn_readfirstlane(win.row), leaving = __builtin_amdgcn_readfirstlane(win.bv);
        if (r < 0 || r >= m || leaving < 0 || leaving >= n_global) { // as above
            status = TS_INTERNAL;
            break;
        }
        const double dr = win.dv, xr = win.xv;
        // ---- eta column (:217-245), x_B (:251-253); row m's entry of it is the reduced-cost row's multiplier
        const double eg = ((t == r) ? 1.0 : -di) / dr;
        double rowp = incol ? Tcol[(int64_t)r * ld] : 0.0;
        if (inx) xb = fma(eg, xr, (t == r) ? -0.0 : xb);
        if (t == m) sh_em[ph] = eg;
        if (t == r) {
#pragma unroll
            for (int u = 0; u < KB; ++u) sh_er[u] = eo[u];
#pragma unroll
            for (int u = 0; u < P; ++u) sh_er[KB + u] = en[u];
            bas = j;
            myflag |= 1u << P;
        }
        __syncthreads();
#ifdef LPX_TS_STAMPS
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#endif
        TS_STAMP(4);
        // ---- row r as the pending updates leave it (tb_row_chain's expressions): this update's old pivot row
        {
            const bool general = __ballot(TS_ROW(pkv) == r) != 0ull; // the leaving row is one a pending update pivoted on
            if (PO > 0) {
                double er[KB];
#pragma unroll
                for (int u = 0; u < KB; ++u) er[u] = sh_er[u];
                if (general) rowp = ts_row_steps<KB, 0, true>(PO, rowp, er, ro, pkv, wslot, r, t, wave, cond);
                else rowp = ts_row_steps<KB, 0, false>(PO, rowp, er, ro, pkv, wslot, r, t, wave, cond);
            }
            if (P > 0) {
                double er[KB];
#pragma unroll
                for (int u = 0; u < KB; ++u) er[u] = (u < P) ? sh_er[KB + u] : 0.0;
                if (general) rowp = ts_row_steps<KB, KB, true>(P, rowp, er, rn, pkv, wslot, r, t, wave, cond);
                else rowp = ts_row_steps<KB, KB, false>(P, rowp, er, rn, pkv, wslot, r, t, wave, cond);
            }
        }
        TS_STAMP(5);
        // ---- the reduced-cost row after this pivot
        if (inrc) {
            const double em = sh_em[ph];
            if (t == slot) {
                rc = cond ? em : 0.0;
                if (cond) {
                    var = leaving;
                    mycol = true;
                }
            } else {
                rc = fma(em, rowp, rc);
            }
        }
        // this pivot joins the pending ones
        en[P] = eg;
        rn[P] = rowp;
        if (lane == KB + P) pkv = ts_pack(r, slot);
        wrow |= (unsigned)((r >> 6) == wave) << (KB + P);
        wslot |= (unsigned)((slot >> 6) == wave) << (KB + P);
        if (t == 0) {
            sh_piv[P][0] = r;
            sh_piv[P][1] = slot;
            sh_piv[P][2] = j;
            sh_piv[P][3] = leaving;
        }
        done = P + 1;
        last_j = j;
        last_ratio = thmin;
        TS_STAMP(6);
        // no barrier closes a pivot: the next one's first LDS write that anybody here still reads (sh_rk) comes after its
        // pricing barrier, sh_er after its ratio barrier, and the records alternate two copies
    }
    // ---- what the control block reports of the last pricing and the last ratio test: the smallest reduced cost, the
    // entering column's maximum (exact whatever the order: the values the per-pivot kernels leave there).  Every launch
    // leaves them: the loop may end on the first pricing of the next one, which then has no ratio test of its own to report
    const TsArgs &z = sh_args;
    TsMinMax mm{rc_priced, d_tested};
    mm = wave_allreduce(mm, [](TsMinMax x, TsMinMax y) { return TsMinMax{lpx_min_f64(x.mn, y.mn), lpx_max_f64(x.mx, y.mx)}; });
    if (lane == 0) sh_red[wave] = mm;
    __syncthreads();
    mm = row16_allreduce(sh_red[lane & (TS_WAVES - 1)], [](TsMinMax x, TsMinMax y) { return TsMinMax{lpx_min_f64(x.mn, y.mn), lpx_max_f64(x.mx, y.mx)}; });
    // ---- the block's pairs and the state the next launch (or the caller) starts from
    if (inrow) {
        double *dst = z.b.Et + t;
        const int64_t lde = z.b.lde;
#pragma unroll
        for (int u = 0; u < KB; ++u)
            if (u < done) dst[u * lde] = en[u];
    }
    if (incol) {
        double *dst = z.b.R + t;
        const int64_t ldr = z.b.ldr;
#pragma unroll
        for (int u = 0; u < KB; ++u)
            if (u < done) dst[u * ldr] = rn[u];
    }
    if (inrc) z.b.rc[t] = rc;
    if (inx) z.b.xb[t] = xb;
    if (myflag) {
        z.b.rowflag[t] = myflag;
        z.s.basis[t] = bas; // primal_solvers.py:96-108, :261
    }
    if (mycol) z.s.slot_var[t] = var;
    if (t == 0) {
        // one thread, in pivot order (a variable may leave and enter again inside a block); everything it needs is read
        // first, so that the stores follow one another instead of each waiting for its own LDS reads
        lpx_ctrl *ctrl = z.s.ctrl;
        int64_t *var_slot = z.s.var_slot, *log = z.s.log, *prow = z.b.prow, *pslot = z.b.pslot;
        const int64_t log_cap = z.s.log_cap, pivots0 = ctrl->pivots;
        int pr[KB], ps[KB], pj[KB], pl[KB];
#pragma unroll
        for (int P = 0; P < KB; ++P) {
            pr[P] = sh_piv[P][0];
            ps[P] = sh_piv[P][1];
            pj[P] = sh_piv[P][2];
            pl[P] = sh_piv[P][3];
        }
#pragma unroll
        for (int P = 0; P < KB; ++P) {
            if (P < done) {
                if (cond) {
                    var_slot[pl[P]] = ps[P];
                    var_slot[pj[P]] = -1;
                } else {
                    t_swap_columns(z.s, pj[P], pl[P]);
                }
                const int64_t k = pivots0 + P;
                if (log && k < log_cap) {
                    log[3 * k + 0] = pr[P];
                    log[3 * k + 1] = pj[P];
                    log[3 * k + 2] = pl[P];
                }
                prow[P] = pr[P];
                pslot[P] = ps[P];
            }
        }
        *z.b.cnt = done;
        if (done > 0) z.b.tsel[0] = z.dstid;
        ctrl->counter += priced;
        ctrl->pivots = pivots0 + done;
        ctrl->price_val = mm.mn;
        if (status != LPX_RUNNING) ctrl->status = status;
        if (last_j >= 0) ctrl->enter_j = last_j;
        if (status == LPX_PRIMAL_UNBOUNDED) ctrl->aux_val = mm.mx;
        if (done > 0) {
            ctrl->enter_slot = sh_piv[done - 1][1];
            ctrl->leave_r = sh_piv[done - 1][0];
            ctrl->leave_var = sh_piv[done - 1][3];
            ctrl->ratio_val = last_ratio;
            if (status != LPX_PRIMAL_UNBOUNDED) ctrl->aux_val = mm.mx;
            ctrl->flip = 0;
            ctrl->f32_phase = 0;
        }
    }
#ifdef LPX_TS_STAMPS
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (t == 0) {
        const long long now = (long long)wall_clock64();
        if (((z.s.ctrl->pivots) & 1023) < KB)
            printf("ts_decide ticks/8 pivots: price %lld colgather %lld colchain %lld ratio %lld rowgather %lld rowchain %lld tail %lld loop %lld | prologue %lld epilogue %lld period %lld prev-exit-to-entry %lld\n",
                   ts_acc[0], ts_acc[1], ts_acc[2], ts_acc[3], ts_acc[4], ts_acc[5], ts_acc[6], ts_last - ts_first,
                   ts_first - ts_entry, now - ts_last, ts_entry - g_ts_entry, ts_entry - g_ts_exit);
        g_ts_entry = ts_entry;
        g_ts_exit = now;
    }
#endif
}

// the pass of one workgroup of the small chain: strip bx (128 columns), band by (rpb rows), 16 waves each on TS_RP rows at a
// time; the strip's entries of the pending pivot rows in LDS; tb_apply's expressions per element and update (the general
// form: a tableau of 8 MB needs no register-tuned fast path, and this workgroup shares its register budget with the
// decision role)
#define TS_RP 2
template <int KB>
__device__ __forceinline__ void ts_pass(const TbFlushArgs &a, const double *Et, int64_t lde, int64_t bx, int64_t by, d2_t *prs)
{
    const int cnt = (int)*a.cnt;
    if (cnt <= 0) return;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const bool cond = a.cond != 0;
    const int64_t c0 = bx * TB2_COLS, col = c0 + 2 * lane;
    const bool live = col < a.cols;
    for (int idx = threadIdx.x; idx < KB * 64; idx += TS_THREADS) {
        const int u = idx >> 6;
        const int64_t c = c0 + 2 * (idx & 63);
        prs[idx] = (u < cnt && c < a.cols) ? *(const d2_t *)(a.R + u * a.ldr + c) : d2_t{0.0, 0.0};
    }
    uint32_t fm0 = 0u, fm1 = 0u;
    for (int u = 0; u < cnt; ++u) {
        const int64_t su = a.pslot[u];
        fm0 |= (uint32_t)(col == su) << u;
        fm1 |= (uint32_t)(col + 1 == su) << u;
    }
    __syncthreads();
    if (!live) return;
    const int64_t row0 = by * a.rpb, row1 = (row0 + a.rpb < a.rows) ? row0 + a.rpb : a.rows;
    for (int64_t i0 = row0 + wave; i0 < row1; i0 += (int64_t)TS_WAVES * TS_RP) {
        d2_t x[TS_RP];
        double e[TS_RP][KB];
        uint32_t rf[TS_RP];
#pragma unroll
        for (int q = 0; q < TS_RP; ++q) {
            const int64_t i = (i0 + q * TS_WAVES < row1) ? i0 + q * TS_WAVES : row1 - 1; // ragged tail: re-read, not stored
            x[q] = *(const d2_t *)(a.T + i * a.ld + col);
            rf[q] = a.rowflag[i];
#pragma unroll
            for (int u = 0; u < KB; ++u) e[q][u] = Et[u * lde + i];
        }
#pragma unroll
        for (int u = 0; u < KB; ++u) {
            if (u < cnt) {
                const d2_t pr = prs[u * 64 + lane];
#pragma unroll
                for (int q = 0; q < TS_RP; ++q) {
                    const bool isr = (rf[q] >> u) & 1u;
                    x[q][0] = tb_apply(x[q][0], (fm0 >> u) & 1u, isr, e[q][u], pr[0], cond);
                    x[q][1] = tb_apply(x[q][1], (fm1 >> u) & 1u, isr, e[q][u], pr[1], cond);
                }
            }
        }
#pragma unroll
        for (int q = 0; q < TS_RP; ++q)
            if (i0 + q * TS_WAVES < row1) *(d2_t *)(a.Tdst + (i0 + q * TS_WAVES) * a.ld + col) = x[q];
    }
}

// grid: workgroup 0 decides, workgroups 1 .. nstrips * nbands are the pass (strip-major as k_tb_flush2's grid)
template <int KB, int PO>
__global__ __launch_bounds__(TS_THREADS) void k_tb_small(TsArgs a)
{
    if (blockIdx.x == 0) {
        if (a.decide) ts_decide<KB, PO>(a);
        return;
    }
    if (!a.pass) return;
    __shared__ d2_t prs_all[KB * 64];
    const int64_t w = (int64_t)blockIdx.x - 1;
    const int64_t by = w / a.nstrips, bx = w - by * a.nstrips;
    ts_pass<KB>(a.f, a.fEt, a.flde, bx, by, prs_all);
}

void lpx_tb_small_launch(const TsArgs &a, unsigned grid, void *stream)
{
    if (a.decide && a.uo == TS_KB) hipLaunchKernelGGL((k_tb_small<TS_KB, TS_KB>), dim3(grid), dim3(TS_THREADS), 0, (hipStream_t)stream, a);
    else hipLaunchKernelGGL((k_tb_small<TS_KB, 0>), dim3(grid), dim3(TS_THREADS), 0, (hipStream_t)stream, a); // a.uo == 0, or no decisions
}
