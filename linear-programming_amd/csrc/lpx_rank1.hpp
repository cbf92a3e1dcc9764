// lpx_rank1.hpp -- K1, the streaming rank-1 pivot update (the dominant kernel of the path).
//
// Replaces PrimalRevisedSimplexSolver._update_inv_basis_matrix / _update_bfs (primal_solvers.py:247-253):
// the reference multiplies by a dense eta matrix E (m x m dgemm, 2 m^3 flop); E differs from I in one column,
// so  E @ M  ==  M[i,:] += e_i * M_old[r,:]  (i != r),  M[r,:] = e_r * M_old[r,:].
//
// HBM-bound: every element of M is read once and written once (16 B per element, 2 flop).
// Layout / mapping: M row-major fp64; a workgroup owns a tile of `rows_per_block` rows x 512 columns; each
// lane owns two adjacent columns (one 16-byte access per row), so a wave touches 1 KiB contiguous per row.
// The pivot-row strip (old row r, copied to `rowr` by the prepare kernel so that the in-place update cannot
// race with it) lives in registers for the whole tile; the eta multipliers e_i are workgroup-uniform scalar
// loads.  Rows are unrolled by 4 to keep >= 4 x 16 B loads per lane in flight.
#pragma once
#include "lpx_device.hpp"

struct Rank1Args {
    double *M;
    int64_t rows, cols, ld;
    const double *e;       // rows
    const double *rowr;    // >= ld (old pivot row)
    int64_t r;             // used when ctrl == nullptr
    const lpx_ctrl *ctrl;  // non-null: run only if status == RUNNING && !flip; r = ctrl->leave_r
    int64_t rows_per_block;
    double *xB;            // revised form: x_B updated by the first column strip (else nullptr)
    const double *xr_old;  // revised form: old x_B[r]
    int64_t col0;          // tableau form: global index of local column 0
    // tableau form: pricing fused into the update -- the workgroups that produce the new reduced-cost row
    // (last row of M) also reduce it: price_v0[strip] = min rc, price_i0[strip] = first global column with rc<0
    double *price_v0;
    int64_t *price_i0;
    int pricing;           // LPX_PRICE_FIRST_NEGATIVE: i0 = first column with rc<0 ; LPX_PRICE_DANTZIG: i0 = argmin rc
    // condensed tableau (nonbasic columns only, lpx_tableau.hip): variable of each column slot and its inverse map; the
    // slot handed to the leaving variable is rewritten with the eta column instead of a unit vector
    const int64_t *slot_var;
    const int64_t *var_slot;
    // ragged last strip folded into its neighbour: the workgroups of the last strip make a second pass over the few
    // columns left beyond it (20001 = 39 x 512 + 33: a 40th workgroup per row with 17 live lanes costs as much as a full one)
    int fold;
};

// NT: 0 plain; 1 non-temporal loads and stores; 2 / 3 non-temporal loads / stores only.  (Write-through `sc1` stores
// for the pass of the pipelined chain, so that kernels ending beside it find no dirty lines of its to write back, were
// measured in round 3 and lost: 21.5 k against 22.6 k pivots/s)
template <int NT>
__device__ __forceinline__ d2_t lpx_ld2(const double *p)
{
    if (NT == 1 || NT == 2) return __builtin_nontemporal_load((const d2_t *)p);
    return *(const d2_t *)p;
}
template <int NT>
__device__ __forceinline__ void lpx_st2(double *p, d2_t v)
{
    if (NT == 1 || NT == 3) __builtin_nontemporal_store(v, (d2_t *)p);
    else *(d2_t *)p = v;
}

// VPT = 16-byte vectors per lane per row (a workgroup covers 512*VPT columns), UNROLL = rows in flight.
template <bool TABLEAU, int NT, int VPT, int UNROLL>
__global__ __launch_bounds__(1024) void k_rank1(Rank1Args a)
{
    int64_t r = a.r;
    int64_t unit_col = -1;
    if (a.ctrl) {
        if (a.ctrl->status != LPX_RUNNING || a.ctrl->flip) return;
        r = a.ctrl->leave_r;
        if (TABLEAU) {
            // local index of the entering column; another rank's column (column partition) must not alias the
            // x_B column at local index cols-1
            // local column rewritten as a whole (unit vector / eta column), left by the prepare kernel; another rank's
            // column (column partition) gives -1 and cannot alias the x_B column at local index cols-1
            unit_col = a.ctrl->enter_slot;
            if (unit_col < 0 || unit_col >= a.cols - 1) unit_col = -1;
        }
    }
    const int nthr = blockDim.x;
    const bool cond = TABLEAU && a.var_slot != nullptr;
    const int64_t row0 = (int64_t)blockIdx.y * a.rows_per_block;
    const int64_t row1 = (row0 + a.rows_per_block < a.rows) ? row0 + a.rows_per_block : a.rows;
    const double *__restrict__ e = a.e;

    if (!TABLEAU && a.xB != nullptr && blockIdx.x == 0) {
        // primal_solvers.py:251-253  bfs = E @ bfs
        const double xr = *a.xr_old;
        for (int64_t i = row0 + threadIdx.x; i < row1; i += nthr) {
            const double ei = e[i];
            a.xB[i] = (i == r) ? ei * xr : fma(ei, xr, a.xB[i]);
        }
    }

    const int npass = (a.fold && blockIdx.x == gridDim.x - 1) ? 2 : 1;
    const bool pricing_wg = TABLEAU && a.price_v0 != nullptr && row1 == a.rows; // workgroup-uniform
    double mn = LPX_INF;
    int64_t jn = LPX_NOIDX;
    for (int pass = 0; pass < npass; ++pass) {
    const int64_t colbase = (((int64_t)blockIdx.x + pass) * (nthr * VPT) + threadIdx.x) * 2;
    d2_t pr[VPT];
    bool live[VPT], fix0[VPT], fix1[VPT];
#pragma unroll
    for (int v = 0; v < VPT; ++v) {
        const int64_t col = colbase + (int64_t)v * (2 * nthr);
        live[v] = col < a.cols;
        pr[v] = live[v] ? *(const d2_t *)(a.rowr + col) : d2_t{0.0, 0.0};
        // tableau form: the entering column becomes the unit vector exactly (primal_solvers.py:65 keeps basic
        // reduced costs at exactly 0; a carried tableau must do the same for the whole column)
        fix0[v] = TABLEAU && (col == unit_col);
        fix1[v] = TABLEAU && (col + 1 == unit_col);
    }
    const int64_t ld = a.ld;
    const int64_t lastrow = a.rows - 1;
    d2_t lastv[VPT];
#pragma unroll
    for (int v = 0; v < VPT; ++v) lastv[v] = d2_t{0.0, 0.0};
    double *p = a.M + row0 * ld + colbase;
    int64_t i = row0;
    if (live[0]) {
    for (; i + UNROLL <= row1; i += UNROLL) {
        d2_t x[UNROLL][VPT];
#pragma unroll
        for (int u = 0; u < UNROLL; ++u)
#pragma unroll
            for (int v = 0; v < VPT; ++v)
                if (live[v]) x[u][v] = lpx_ld2<NT>(p + u * ld + v * (2 * nthr));
#pragma unroll
        for (int u = 0; u < UNROLL; ++u) {
            const double ei = e[i + u];
            const bool isr = (i + u) == r;
#pragma unroll
            for (int v = 0; v < VPT; ++v) {
                if (isr) {
                    x[u][v][0] = ei * pr[v][0];
                    x[u][v][1] = ei * pr[v][1];
                } else {
                    x[u][v][0] = fma(ei, pr[v][0], x[u][v][0]);
                    x[u][v][1] = fma(ei, pr[v][1], x[u][v][1]);
                }
                if (fix0[v]) x[u][v][0] = cond ? ei : (isr ? 1.0 : 0.0);
                if (fix1[v]) x[u][v][1] = cond ? ei : (isr ? 1.0 : 0.0);
                if (TABLEAU && (i + u) == lastrow) lastv[v] = x[u][v];
            }
        }
#pragma unroll
        for (int u = 0; u < UNROLL; ++u)
#pragma unroll
            for (int v = 0; v < VPT; ++v)
                if (live[v]) lpx_st2<NT>(p + u * ld + v * (2 * nthr), x[u][v]);
        p += UNROLL * ld;
    }
    for (; i < row1; ++i) {
        const double ei = e[i];
        const bool isr = i == r;
#pragma unroll
        for (int v = 0; v < VPT; ++v) {
            if (!live[v]) continue;
            d2_t x0 = lpx_ld2<NT>(p + v * (2 * nthr));
            if (isr) {
                x0[0] = ei * pr[v][0];
                x0[1] = ei * pr[v][1];
            } else {
                x0[0] = fma(ei, pr[v][0], x0[0]);
                x0[1] = fma(ei, pr[v][1], x0[1]);
            }
            if (fix0[v]) x0[0] = cond ? ei : (isr ? 1.0 : 0.0);
            if (fix1[v]) x0[1] = cond ? ei : (isr ? 1.0 : 0.0);
            if (TABLEAU && i == lastrow) lastv[v] = x0;
            lpx_st2<NT>(p + v * (2 * nthr), x0);
        }
        p += ld;
    }
    } // live[0]

    if (pricing_wg) { // this pass's part of the new reduced-cost row
        const int64_t nstruct = a.cols - 1; // column cols-1 is x_B / -objective
#pragma unroll
        for (int v = 0; v < VPT; ++v) {
            const int64_t col = colbase + (int64_t)v * (2 * nthr);
#pragma unroll
            for (int q = 0; q < 2; ++q) {
                if (col + q < nstruct) {
                    const double rc = lastv[v][q];
                    const int64_t gj = a.slot_var ? a.slot_var[col + q] : a.col0 + col + q;
                    if (a.pricing == LPX_PRICE_DANTZIG) {
                        if (vi_less(rc, gj, mn, jn)) { mn = rc; jn = gj; }
                    } else {
                        mn = fmin(mn, rc);
                        if (rc < 0.0 && gj < jn) jn = gj;
                    }
                }
            }
        }
    }
    } // pass

    if (pricing_wg) {
        __shared__ BlockRed red;
        if (a.pricing == LPX_PRICE_DANTZIG) {
            const VI t = block_argmin(VI{mn, jn}, red);
            mn = t.v;
            jn = t.i;
        } else {
            mn = block_min(mn, red);
            jn = block_imin(jn, red);
        }
        if (threadIdx.x == 0) {
            a.price_v0[blockIdx.x] = mn;
            a.price_i0[blockIdx.x] = jn;
        }
    }
}

// Revised form: K1 fused with the NEXT iteration's y pass.  y = c_B B^-1 (primal_solvers.py:64) is a weighted column
// sum of the matrix K1 has just produced, so the workgroup that writes a row chunk of the new B^-1 also leaves that
// chunk's partial column sums -- 8 m^2 bytes of the next iteration (its k_colsum<double> launch) never leave HBM.
// The price is K1's flat mapping: the grid is the column-sum grid (strips x row chunks, ~2 workgroups per CU, each
// walking down its chunk), 5.4 instead of 6.1 TB/s for the update itself at m = 10000, 34 us lost against 122 us saved.
// The partials are bit-identical to k_colsum's: same chunks, same sequential fma chain per column, same weights.
struct Rank1YArgs {
    const float *c;       // weights c[basis[i]] (basis already holds the entering variable: k_prepare ran)
    const int64_t *basis;
    double *ypart;        // [chunk][ldyp]
    int64_t ldyp;
};
template <int NT, int UNROLL>
__global__ __launch_bounds__(LPX_BLOCK) void k_rank1_y(Rank1Args a, Rank1YArgs f)
{
    extern __shared__ __attribute__((aligned(16))) char rank1y_lds[];
    double *wl = (double *)rank1y_lds;
    if (a.ctrl->status != LPX_RUNNING || a.ctrl->flip) return;
    const int64_t r = a.ctrl->leave_r;
    const int nthr = blockDim.x;
    const int64_t row0 = (int64_t)blockIdx.y * a.rows_per_block;
    const int64_t row1 = (row0 + a.rows_per_block < a.rows) ? row0 + a.rows_per_block : a.rows;
    const double *__restrict__ e = a.e;
    for (int64_t i = row0 + threadIdx.x; i < row1; i += nthr) wl[i - row0] = (double)f.c[f.basis[i]];
    if (a.xB != nullptr && blockIdx.x == 0) { // primal_solvers.py:251-253  bfs = E @ bfs
        const double xr = *a.xr_old;
        for (int64_t i = row0 + threadIdx.x; i < row1; i += nthr) {
            const double ei = e[i];
            a.xB[i] = (i == r) ? ei * xr : fma(ei, xr, a.xB[i]);
        }
    }
    __syncthreads();
    const int64_t col = ((int64_t)blockIdx.x * nthr + threadIdx.x) * 2;
    if (col >= a.cols) return;
    const d2_t pr = *(const d2_t *)(a.rowr + col);
    const int64_t ld = a.ld;
    d2_t acc{0.0, 0.0};
    double *p = a.M + row0 * ld + col;
    int64_t i = row0;
    for (; i + UNROLL <= row1; i += UNROLL) {
        d2_t x[UNROLL];
#pragma unroll
        for (int u = 0; u < UNROLL; ++u) x[u] = lpx_ld2<NT>(p + u * ld);
#pragma unroll
        for (int u = 0; u < UNROLL; ++u) {
            const double ei = e[i + u];
            if ((i + u) == r) {
                x[u][0] = ei * pr[0];
                x[u][1] = ei * pr[1];
            } else {
                x[u][0] = fma(ei, pr[0], x[u][0]);
                x[u][1] = fma(ei, pr[1], x[u][1]);
            }
        }
#pragma unroll
        for (int u = 0; u < UNROLL; ++u) lpx_st2<NT>(p + u * ld, x[u]);
#pragma unroll
        for (int u = 0; u < UNROLL; ++u) {
            const double w = wl[i + u - row0];
            acc[0] = fma(w, x[u][0], acc[0]);
            acc[1] = fma(w, x[u][1], acc[1]);
        }
        p += UNROLL * ld;
    }
    for (; i < row1; ++i) {
        const double ei = e[i];
        d2_t x0 = lpx_ld2<NT>(p);
        if (i == r) {
            x0[0] = ei * pr[0];
            x0[1] = ei * pr[1];
        } else {
            x0[0] = fma(ei, pr[0], x0[0]);
            x0[1] = fma(ei, pr[1], x0[1]);
        }
        lpx_st2<NT>(p, x0);
        const double w = wl[i - row0];
        acc[0] = fma(w, x0[0], acc[0]);
        acc[1] = fma(w, x0[1], acc[1]);
        p += ld;
    }
    double *out = f.ypart + (int64_t)blockIdx.y * f.ldyp + col;
    out[0] = acc[0];
    if (col + 1 < a.cols) out[1] = acc[1];
}

// Host-side launch plan.  rows_per_block is chosen so that the grid is many times the ~2048 resident
// workgroups (256 CUs x 8): a grid that only just exceeds residency has a 2x tail.
struct Rank1Plan {
    dim3 grid;
    int64_t rows_per_block;
    int vpt, unroll, nt, threads;
    int fold; // the last strip's workgroups also cover the ragged remainder (Rank1Args.fold)
};
static inline int lpx_env_int(const char *name, int dflt)
{
    const char *env = getenv(name);
    return (env && *env) ? atoi(env) : dflt;
}
static inline Rank1Plan lpx_rank1_plan(int64_t rows, int64_t cols)
{
    Rank1Plan p;
    // defaults from the sweep on MI355X (tools/k1_sweep.py, profiles/): one row x 512 columns per workgroup,
    // non-temporal loads and stores -- workgroups are dispatched x-fastest, so the chip-wide access front
    // walks the matrix in address order.
    p.vpt = lpx_env_int("LPX_RANK1_VPT", 1);
    p.unroll = lpx_env_int("LPX_RANK1_UNROLL", 1);
    // non-temporal only when the matrix cannot stay in the 256 MiB Infinity Cache between pivots
    p.nt = lpx_env_int("LPX_RANK1_NT", (rows * cols * 8 > ((int64_t)192 << 20)) ? 1 : 0);
    p.threads = lpx_env_int("LPX_RANK1_THREADS", LPX_BLOCK);
    if (p.threads != 64 && p.threads != 128 && p.threads != 512 && p.threads != 1024) p.threads = LPX_BLOCK;
    int64_t rpb = lpx_env_int("LPX_RANK1_ROWS", 1);
    if (p.vpt != 2 && p.vpt != 4) p.vpt = 1;
    if (p.unroll != 1 && p.unroll != 2 && p.unroll != 8 && p.unroll != 16) p.unroll = 4;
    if (p.nt < 0 || p.nt > 3) p.nt = 0;
    int64_t nstrips = lpx_cdiv(cols, 2 * (int64_t)p.threads * p.vpt);
    // a remainder of at most a quarter strip is folded into the strip before it
    const int64_t strip = 2 * (int64_t)p.threads * p.vpt, rem = cols - (nstrips - 1) * strip;
    p.fold = (lpx_env_int("LPX_RANK1_FOLD", 1) != 0 && nstrips > 1 && 4 * rem <= strip) ? 1 : 0;
    nstrips -= p.fold;
    // small problems: shrink the tile until there are enough workgroups to occupy the chip
    const int64_t minwg = lpx_env_int("LPX_RANK1_MINWG", 1024);
    while (rpb > 4 && nstrips * lpx_cdiv(rows, rpb) < minwg) rpb /= 2;
    if (rpb < p.unroll) p.unroll = rpb >= 8 ? 8 : rpb >= 4 ? 4 : rpb >= 2 ? 2 : 1;
    p.rows_per_block = rpb;
    p.grid = dim3((unsigned)nstrips, (unsigned)lpx_cdiv(rows, rpb), 1);
    return p;
}
template <bool TABLEAU>
static inline void lpx_rank1_launch(const Rank1Plan &p, Rank1Args a, hipStream_t st)
{
    a.rows_per_block = p.rows_per_block;
    a.fold = p.fold;
#define LPX_R1_GO(NT, VPT, UN) hipLaunchKernelGGL((k_rank1<TABLEAU, NT, VPT, UN>), p.grid, dim3(p.threads), 0, st, a)
#define LPX_R1_UN(NT, VPT)                          \
    switch (p.unroll) {                             \
    case 16: LPX_R1_GO(NT, VPT, 16); break;         \
    case 8: LPX_R1_GO(NT, VPT, 8); break;           \
    case 2: LPX_R1_GO(NT, VPT, 2); break;           \
    case 1: LPX_R1_GO(NT, VPT, 1); break;           \
    default: LPX_R1_GO(NT, VPT, 4); break;          \
    }
#define LPX_R1_VPT(NT)                              \
    if (p.vpt == 4) { LPX_R1_UN(NT, 4) } else if (p.vpt == 2) { LPX_R1_UN(NT, 2) } else { LPX_R1_UN(NT, 1) }
    switch (p.nt) {
    case 1: LPX_R1_VPT(1) break;
    case 2: LPX_R1_VPT(2) break;
    case 3: LPX_R1_VPT(3) break;
    default: LPX_R1_VPT(0) break;
    }
#undef LPX_R1_VPT
#undef LPX_R1_UN
#undef LPX_R1_GO
}

// ---------------------------------------------------------------------------------------------------------
// K1 timing hooks (bench.py roofline): HIP events recorded on the launch stream around every K1 launch while
// enabled.  Events are created once by lpx_profile_k1(1); recording them allocates nothing.
// ---------------------------------------------------------------------------------------------------------
#define LPX_PROF_CAP 4096
struct LpxK1Profile {
    bool enabled = false;
    bool created = false;
    int64_t count = 0;
    hipEvent_t start[LPX_PROF_CAP];
    hipEvent_t stop[LPX_PROF_CAP];
};
LpxK1Profile &lpx_k1_profile();
static inline void lpx_k1_begin(hipStream_t st)
{
    LpxK1Profile &p = lpx_k1_profile();
    if (p.enabled && p.count < LPX_PROF_CAP) (void)hipEventRecord(p.start[p.count], st);
}
static inline void lpx_k1_end(hipStream_t st)
{
    LpxK1Profile &p = lpx_k1_profile();
    if (p.enabled && p.count < LPX_PROF_CAP) {
        (void)hipEventRecord(p.stop[p.count], st);
        p.count += 1;
    }
}
