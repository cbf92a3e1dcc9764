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
};

template <bool TABLEAU, bool NT>
__device__ __forceinline__ d2_t lpx_ld2(const double *p)
{
    if (NT) return __builtin_nontemporal_load((const d2_t *)p);
    return *(const d2_t *)p;
}
template <bool NT>
__device__ __forceinline__ void lpx_st2(double *p, d2_t v)
{
    if (NT) __builtin_nontemporal_store(v, (d2_t *)p);
    else *(d2_t *)p = v;
}

template <bool TABLEAU, bool NT>
__global__ __launch_bounds__(LPX_BLOCK) void k_rank1(Rank1Args a)
{
    int64_t r = a.r;
    int64_t unit_col = -1;
    if (a.ctrl) {
        if (a.ctrl->status != LPX_RUNNING || a.ctrl->flip) return;
        r = a.ctrl->leave_r;
        if (TABLEAU) unit_col = a.ctrl->enter_j - a.col0;
    }
    const int64_t row0 = (int64_t)blockIdx.y * a.rows_per_block;
    const int64_t row1 = (row0 + a.rows_per_block < a.rows) ? row0 + a.rows_per_block : a.rows;
    const double *__restrict__ e = a.e;

    if (!TABLEAU && a.xB != nullptr && blockIdx.x == 0) {
        // primal_solvers.py:251-253  bfs = E @ bfs
        const double xr = *a.xr_old;
        for (int64_t i = row0 + threadIdx.x; i < row1; i += LPX_BLOCK) {
            const double ei = e[i];
            a.xB[i] = (i == r) ? ei * xr : fma(ei, xr, a.xB[i]);
        }
    }

    const int64_t col = ((int64_t)blockIdx.x * LPX_BLOCK + threadIdx.x) * 2;
    if (col >= a.cols) return;
    const d2_t pr = *(const d2_t *)(a.rowr + col);
    // tableau form: the entering column becomes the unit vector exactly (primal_solvers.py:65 keeps basic
    // reduced costs at exactly 0; a carried tableau must do the same for the whole column)
    const bool fix0 = TABLEAU && (col == unit_col);
    const bool fix1 = TABLEAU && (col + 1 == unit_col);

    double *p = a.M + row0 * a.ld + col;
    const int64_t ld = a.ld;
    int64_t i = row0;
#define LPX_R1_ROW(vv, ii)                                                \
    {                                                                     \
        const double ei = e[ii];                                          \
        if ((ii) == r) {                                                  \
            vv[0] = ei * pr[0];                                           \
            vv[1] = ei * pr[1];                                           \
        } else {                                                          \
            vv[0] = fma(ei, pr[0], vv[0]);                                \
            vv[1] = fma(ei, pr[1], vv[1]);                                \
        }                                                                 \
        if (fix0) vv[0] = ((ii) == r) ? 1.0 : 0.0;                        \
        if (fix1) vv[1] = ((ii) == r) ? 1.0 : 0.0;                        \
    }
    for (; i + 4 <= row1; i += 4) {
        d2_t x0 = lpx_ld2<TABLEAU, NT>(p);
        d2_t x1 = lpx_ld2<TABLEAU, NT>(p + ld);
        d2_t x2 = lpx_ld2<TABLEAU, NT>(p + 2 * ld);
        d2_t x3 = lpx_ld2<TABLEAU, NT>(p + 3 * ld);
        LPX_R1_ROW(x0, i)
        LPX_R1_ROW(x1, i + 1)
        LPX_R1_ROW(x2, i + 2)
        LPX_R1_ROW(x3, i + 3)
        lpx_st2<NT>(p, x0);
        lpx_st2<NT>(p + ld, x1);
        lpx_st2<NT>(p + 2 * ld, x2);
        lpx_st2<NT>(p + 3 * ld, x3);
        p += 4 * ld;
    }
    for (; i < row1; ++i) {
        d2_t x0 = lpx_ld2<TABLEAU, NT>(p);
        LPX_R1_ROW(x0, i)
        lpx_st2<NT>(p, x0);
        p += ld;
    }
#undef LPX_R1_ROW
}

// Host-side launch plan: 512 columns per workgroup; rows_per_block chosen so that the grid is many times the
// 2048 resident workgroups (256 CUs x 8) -- a grid that only just exceeds residency has a 2x tail.
struct Rank1Plan {
    dim3 grid;
    int64_t rows_per_block;
};
static inline Rank1Plan lpx_rank1_plan(int64_t rows, int64_t cols)
{
    int64_t rpb = 64;
    const char *env = getenv("LPX_RANK1_ROWS");
    if (env && atoi(env) > 0) rpb = atoi(env);
    const int64_t nstrips = lpx_cdiv(cols, 2 * LPX_BLOCK);
    // small problems: shrink the tile until there are enough workgroups to occupy the chip
    while (rpb > 4 && nstrips * lpx_cdiv(rows, rpb) < 1024) rpb /= 2;
    Rank1Plan p;
    p.rows_per_block = rpb;
    p.grid = dim3((unsigned)nstrips, (unsigned)lpx_cdiv(rows, rpb), 1);
    return p;
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
