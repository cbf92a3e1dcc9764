// lpx_core.hip -- revised-form simplex loop on MI355X (gfx950): pricing, direction, ratio tests,
// eta/prepare, rank-1 update, expansion; plus the individually exported kernels and control-block helpers.
//
// One iteration is a short chain of kernels on one stream; scalars that one stage produces for the next
// (entering column, leaving row, status) travel through the device-resident control block and through
// per-workgroup partial arrays that the next kernel combines in its prologue, so there is NO host round trip
// inside the loop.  A kernel never reads a control field that the same kernel writes, except `status`,
// whose only transitions are RUNNING -> terminal and are recomputed identically by every workgroup.
#include <stdarg.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "lpx_device.hpp"
#include "lpx_rank1.hpp"
#include "lpx_graph.hpp"

// ---------------------------------------------------------------------------------------------------------
// error string (thread local)
// ---------------------------------------------------------------------------------------------------------
static thread_local char g_err[512] = "";
void lpx_set_error(const char *fmt, ...)
{
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof g_err, fmt, ap);
    va_end(ap);
}
extern "C" const char *lpx_last_error(void) { return g_err; }
extern "C" int lpx_version(void) { return LPX_VERSION; }

// ---------------------------------------------------------------------------------------------------------
// K1 timing hooks
// ---------------------------------------------------------------------------------------------------------
LpxK1Profile &lpx_k1_profile()
{
    static LpxK1Profile p;
    return p;
}
LpxGraphCache &lpx_graph_cache()
{
    static LpxGraphCache c;
    return c;
}
extern "C" int lpx_profile_k1(int enable)
{
    LpxK1Profile &p = lpx_k1_profile();
    if (enable && !p.created) {
        for (int i = 0; i < LPX_PROF_CAP; ++i) {
            LPX_HIP(hipEventCreate(&p.start[i]));
            LPX_HIP(hipEventCreate(&p.stop[i]));
        }
        p.created = true;
    }
    p.enabled = enable != 0;
    p.count = 0;
    return LPX_OK;
}
extern "C" int lpx_profile_k1_read(double *total_ms, int64_t *launches)
{
    LPX_REQUIRE(total_ms && launches, "null output");
    LpxK1Profile &p = lpx_k1_profile();
    double tot = 0.0;
    for (int64_t i = 0; i < p.count; ++i) {
        LPX_HIP(hipEventSynchronize(p.stop[i]));
        float ms = 0.f;
        LPX_HIP(hipEventElapsedTime(&ms, p.start[i], p.stop[i]));
        tot += ms;
    }
    *total_ms = tot;
    *launches = p.count;
    return LPX_OK;
}

// ---------------------------------------------------------------------------------------------------------
// workspace layout (pure function of m, n)
// ---------------------------------------------------------------------------------------------------------
struct ColsumPlan {
    int64_t nstrips, nch, rpc;
    int threads;
};
// V = columns per lane (16-byte vector): 2 for fp64 matrices, 4 for fp32.
// Row chunks are capped at 32 so that the finishing pass (one lane sums a column's partials) stays short; for
// small matrices the workgroup is narrowed (64/128 lanes) until there are enough workgroups to occupy the chip.
// About two workgroups per CU (strips x chunks ~ 480) streams best: measured on the 10000 x 10000 B^-1, 20 strips,
// chunks 8 / 16 / 20 / 24 / 32 / 64 -> 1234 / 1527 / 1558 / 1545 / 1504 / 1478 pivots/s for the whole iteration.
static inline ColsumPlan colsum_plan(int64_t rows, int64_t cols, int V, int64_t maxch_dflt = 32)
{
    ColsumPlan p;
    p.threads = LPX_BLOCK;
    while (p.threads > 64 && lpx_cdiv(cols, (int64_t)p.threads * V) * 32 < 512) p.threads /= 2;
    p.nstrips = lpx_cdiv(cols, (int64_t)p.threads * V);
    int64_t nch = lpx_cdiv(480 * (LPX_BLOCK / p.threads), p.nstrips); // ~2 x 4 waves per CU whatever the width
    const int64_t maxch = lpx_env_int("LPX_COLSUM_MAXCH", (int)maxch_dflt);
    if (nch > maxch) nch = maxch;
    if (nch < 1) nch = 1;
    p.rpc = lpx_cdiv(rows, nch);
    if (p.rpc < 4) p.rpc = 4;
    p.nch = lpx_cdiv(rows, p.rpc);
    return p;
}

struct RevisedWs {
    ColsumPlan py, pa;
    int64_t ldyp, ldzp;
    double *ypart, *y, *zpart, *zpart2, *rc, *row, *d, *e, *rowr, *w, *scal;
    // staged pricing of the primal loop (see revised_staged_pricing): columns [0, n1) first, the rest only if needed
    int64_t n1;
    ColsumPlan p1, p2;
    double *zpart1;
    lpx_ctrl *scr_ctrl; // scratch control block for the primal-dual steps
    Partials price, ratio;
    int64_t bytes;
};
static inline char *carve(char *&p, int64_t bytes)
{
    char *q = p;
    p += lpx_roundup(bytes, 256);
    return q;
}
static inline void carve_partials(char *&p, Partials &P)
{
    P.v0 = (double *)carve(p, 8 * LPX_MAXP);
    P.i0 = (int64_t *)carve(p, 8 * LPX_MAXP);
    P.v1 = (double *)carve(p, 8 * LPX_MAXP);
    P.i1 = (int64_t *)carve(p, 8 * LPX_MAXP);
}
static RevisedWs revised_ws(int64_t m, int64_t n, void *base)
{
    RevisedWs w;
    w.py = colsum_plan(m, m, 2);
    w.pa = colsum_plan(m, n, 4);
    w.ldyp = lpx_roundup(m, 2);
    w.ldzp = lpx_roundup(n, 4);
    char *p = (char *)base;
    w.ypart = (double *)carve(p, 8 * w.py.nch * w.ldyp);
    w.y = (double *)carve(p, 8 * w.ldyp);
    w.zpart = (double *)carve(p, 8 * w.pa.nch * w.ldzp);
    w.zpart2 = (double *)carve(p, 8 * w.pa.nch * w.ldzp);
    w.rc = (double *)carve(p, 8 * w.ldzp);
    w.row = (double *)carve(p, 8 * w.ldzp);
    w.d = (double *)carve(p, 8 * (m + 2));
    w.e = (double *)carve(p, 8 * (m + 2));
    w.rowr = (double *)carve(p, 8 * (lpx_roundup(m, 2) + 2));
    w.w = (double *)carve(p, 8 * (m + 2));
    w.scal = (double *)carve(p, 8 * 16);
    w.n1 = 0;
    w.zpart1 = nullptr;
    // worth it once a pass over A costs more than the two extra launches (LPX_STAGED_MIN_MB: tests force small cases)
    if (4 * m * n >= ((int64_t)lpx_env_int("LPX_STAGED_MIN_MB", 64) << 20) && w.pa.nstrips >= 2) {
        const int64_t strip = (int64_t)w.pa.threads * 4;
        w.n1 = lpx_cdiv(w.pa.nstrips, 8) * strip;
        w.p1 = colsum_plan(m, w.n1, 4, 128);
        w.p2 = colsum_plan(m, n - w.n1, 4);
        w.zpart1 = (double *)carve(p, 8 * w.p1.nch * w.ldzp);
    }
    w.scr_ctrl = (lpx_ctrl *)carve(p, sizeof(lpx_ctrl));
    carve_partials(p, w.price);
    carve_partials(p, w.ratio);
    w.bytes = (int64_t)(p - (char *)base);
    return w;
}
extern "C" int64_t lpx_revised_workspace_bytes(int64_t m, int64_t n)
{
    if (m <= 0 || n <= 0) return 0;
    return revised_ws(m, n, nullptr).bytes;
}

// ---------------------------------------------------------------------------------------------------------
// control block helpers
// ---------------------------------------------------------------------------------------------------------
__global__ void k_ctrl_init(lpx_ctrl *c, int32_t f32_phase)
{
    lpx_ctrl z;
    memset(&z, 0, sizeof z);
    z.f32_phase = f32_phase;
    z.enter_j = -1;
    z.leave_r = -1;
    z.leave_var = -1;
    *c = z;
}
__global__ void k_ctrl_begin(lpx_ctrl *c)
{
    c->counter = 0;
    c->status = LPX_RUNNING;
    c->flip = 0;
}
extern "C" int lpx_ctrl_init(lpx_ctrl *dev_ctrl, int32_t f32_phase, void *stream)
{
    LPX_REQUIRE(dev_ctrl != nullptr, "null control block");
    hipLaunchKernelGGL(k_ctrl_init, dim3(1), dim3(1), 0, (hipStream_t)stream, dev_ctrl, f32_phase);
    LPX_LAUNCH_CHECK();
    return LPX_OK;
}
extern "C" int lpx_ctrl_begin_solve(lpx_ctrl *dev_ctrl, void *stream)
{
    LPX_REQUIRE(dev_ctrl != nullptr, "null control block");
    hipLaunchKernelGGL(k_ctrl_begin, dim3(1), dim3(1), 0, (hipStream_t)stream, dev_ctrl);
    LPX_LAUNCH_CHECK();
    return LPX_OK;
}
extern "C" int lpx_read_ctrl(const lpx_ctrl *dev_ctrl, lpx_ctrl *host_ctrl, void *stream)
{
    LPX_REQUIRE(dev_ctrl != nullptr && host_ctrl != nullptr, "null control block");
    LPX_HIP(hipMemcpyAsync(host_ctrl, dev_ctrl, sizeof(lpx_ctrl), hipMemcpyDeviceToHost, (hipStream_t)stream));
    LPX_HIP(hipStreamSynchronize((hipStream_t)stream));
    return LPX_OK;
}

// ---------------------------------------------------------------------------------------------------------
// K2a: column sums  part[chunk][k] = sum_{i in chunk} w_i * M[i][k]      (y = c_B B^-1 ; z = y A ; row = B^-1[r,:] A)
// primal_solvers.py:64 evaluates  c[basis] @ inv_basis_matrix @ A  left to right; dual_solvers.py:37-39.
// HBM-bound: M is read once (8 m^2 B for B^-1, 4 m n B for the float32 A).
// ---------------------------------------------------------------------------------------------------------
enum { W_CB = 0, W_VEC = 1, W_BINV_ROW = 2, W_YPART = 3 };
struct ColsumArgs {
    const void *M;
    int64_t ld, rows, cols, rpc;
    int wmode;
    const float *c;
    const int64_t *basis;
    const double *wvec;
    const double *Binv;
    int64_t ldB;
    double *part;
    int64_t ldp;
    const lpx_ctrl *ctrl;
    // W_YPART: the weights are y = c_B B^-1, still in the form of the row-chunk partials of the preceding
    // column-sum launch over B^-1; each workgroup finishes the y_i of its own rows into LDS first
    const double *ypart;
    int64_t ynch, yld;
    int lds_weights; // weights staged in dynamic LDS (rpc * 8 bytes)
    int nt;          // matrix larger than the Infinity Cache: non-temporal loads (it is read once per launch)
    // staged pricing: pricing partials of the columns scanned so far; the launch is skipped when they already decide
    // the iteration (pricing_resolved)
    const double *gate_v0;
    const int64_t *gate_i0;
    int64_t gate_n;
    // dual loop: a second weighted column sum in the same pass over M, part2[chunk][k] = sum_i B^-1[r][i] * M[i][k]
    // (the dual row, dual_solvers.py:37-38), r = ctrl->leave_r; needs lds_weights (2 * rpc * 8 bytes)
    double *part2;
};

// primal_solvers.py:135-142 needs two facts about rc: the first index with rc_j < 0 and whether min(rc) passes the
// optimality test.  A prefix of the columns that holds a negative entry below -1e-8 settles both -- the entering column
// is its first negative, "not optimal" is certain -- whatever the rest of the row looks like.
__device__ __forceinline__ bool pricing_resolved(const double *v0, const int64_t *i0, int64_t np, BlockRed &red)
{
    const double mn = combine_min(v0, np, red);
    const int64_t j = combine_imin(i0, np, red);
    return j != LPX_NOIDX && !lpx_is_optimal_min(mn);
}

template <typename AccT>
__device__ __forceinline__ AccT colsum_weight(const ColsumArgs &a, int64_t i, int64_t r)
{
    if (a.wmode == W_CB) return (AccT)a.c[a.basis[i]];
    if (a.wmode == W_VEC) return (AccT)a.wvec[i];
    return (AccT)a.Binv[r * a.ldB + i];
}

// sum of a column's partials in chunk order (fixed => deterministic); loads issued 8 ahead of the adds
template <typename AccT>
__device__ __forceinline__ AccT fin_sum(const double *part, int64_t nch, int64_t ldp, int64_t col)
{
    AccT s = (AccT)0;
    const double *p = part + col;
    int64_t ch = 0;
    for (; ch + 8 <= nch; ch += 8) {
        double v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) v[u] = p[(ch + u) * ldp];
#pragma unroll
        for (int u = 0; u < 8; ++u) s += (AccT)v[u];
    }
    for (; ch < nch; ++ch) s += (AccT)p[ch * ldp];
    return s;
}

// a matrix that does not fit the 256 MiB Infinity Cache next to the rest of an iteration is streamed past it
// (LPX_STREAM_NT = 0 / 1 forces the choice)
static inline int lpx_stream_nt(int64_t bytes)
{
    const int f = lpx_env_int("LPX_STREAM_NT", -1);
    return f >= 0 ? (f != 0) : (bytes > ((int64_t)192 << 20));
}

template <bool NT, typename VecT>
__device__ __forceinline__ VecT lpx_stream_load(const VecT *p)
{
    return NT ? __builtin_nontemporal_load(p) : *p;
}

template <typename MatT, typename AccT, int V, bool NT, bool TWO = false>
__device__ __forceinline__ void colsum_body(const ColsumArgs &a)
{
    typedef MatT vec_t __attribute__((ext_vector_type(V)));
    extern __shared__ __attribute__((aligned(16))) char colsum_lds[];
    AccT *wl = (AccT *)colsum_lds;
    const int64_t col = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) * V;
    const int64_t row0 = (int64_t)blockIdx.y * a.rpc;
    const int64_t row1 = (row0 + a.rpc < a.rows) ? row0 + a.rpc : a.rows;
    // weights of this workgroup's rows go to LDS first: fetching them inside the streaming loop would put one or
    // two dependent global loads (basis[i] -> c[...]) on every row's critical path
    const bool lds_w = a.lds_weights != 0;
    if (lds_w) {
        const int64_t rr = (a.wmode == W_BINV_ROW) ? a.ctrl->leave_r : 0;
        for (int64_t i = row0 + threadIdx.x; i < row1; i += blockDim.x)
            wl[i - row0] = (a.wmode == W_YPART) ? fin_sum<AccT>(a.ypart, a.ynch, a.yld, i) : colsum_weight<AccT>(a, i, rr);
        if (TWO) {
            const int64_t r2 = a.ctrl->leave_r;
            for (int64_t i = row0 + threadIdx.x; i < row1; i += blockDim.x) wl[a.rpc + i - row0] = (AccT)a.Binv[r2 * a.ldB + i];
        }
        __syncthreads();
    }
    if (col >= a.cols) return;
    const int64_t r = (a.wmode == W_BINV_ROW) ? a.ctrl->leave_r : 0;
    AccT acc[V], acc2[V];
#pragma unroll
    for (int v = 0; v < V; ++v) acc[v] = acc2[v] = (AccT)0;
    const MatT *p = (const MatT *)a.M + row0 * a.ld + col;
    int64_t i = row0;
    for (; i + 4 <= row1; i += 4) {
        const vec_t x0 = lpx_stream_load<NT>((const vec_t *)p);
        const vec_t x1 = lpx_stream_load<NT>((const vec_t *)(p + a.ld));
        const vec_t x2 = lpx_stream_load<NT>((const vec_t *)(p + 2 * a.ld));
        const vec_t x3 = lpx_stream_load<NT>((const vec_t *)(p + 3 * a.ld));
        AccT w0, w1, w2, w3;
        if (lds_w) {
            w0 = wl[i - row0]; w1 = wl[i + 1 - row0]; w2 = wl[i + 2 - row0]; w3 = wl[i + 3 - row0];
        } else {
            w0 = colsum_weight<AccT>(a, i, r); w1 = colsum_weight<AccT>(a, i + 1, r);
            w2 = colsum_weight<AccT>(a, i + 2, r); w3 = colsum_weight<AccT>(a, i + 3, r);
        }
#pragma unroll
        for (int v = 0; v < V; ++v) {
            acc[v] = fma(w0, (AccT)x0[v], acc[v]);
            acc[v] = fma(w1, (AccT)x1[v], acc[v]);
            acc[v] = fma(w2, (AccT)x2[v], acc[v]);
            acc[v] = fma(w3, (AccT)x3[v], acc[v]);
        }
        if (TWO) {
            const AccT u0 = wl[a.rpc + i - row0], u1 = wl[a.rpc + i + 1 - row0], u2 = wl[a.rpc + i + 2 - row0],
                       u3 = wl[a.rpc + i + 3 - row0];
#pragma unroll
            for (int v = 0; v < V; ++v) {
                acc2[v] = fma(u0, (AccT)x0[v], acc2[v]);
                acc2[v] = fma(u1, (AccT)x1[v], acc2[v]);
                acc2[v] = fma(u2, (AccT)x2[v], acc2[v]);
                acc2[v] = fma(u3, (AccT)x3[v], acc2[v]);
            }
        }
        p += 4 * a.ld;
    }
    for (; i < row1; ++i) {
        const vec_t x0 = lpx_stream_load<NT>((const vec_t *)p);
        const AccT w0 = lds_w ? wl[i - row0] : colsum_weight<AccT>(a, i, r);
#pragma unroll
        for (int v = 0; v < V; ++v) acc[v] = fma(w0, (AccT)x0[v], acc[v]);
        if (TWO) {
            const AccT u0 = wl[a.rpc + i - row0];
#pragma unroll
            for (int v = 0; v < V; ++v) acc2[v] = fma(u0, (AccT)x0[v], acc2[v]);
        }
        p += a.ld;
    }
    double *out = a.part + (int64_t)blockIdx.y * a.ldp + col;
#pragma unroll
    for (int v = 0; v < V; ++v)
        if (col + v < a.cols) out[v] = (double)acc[v];
    if (TWO) {
        double *out2 = a.part2 + (int64_t)blockIdx.y * a.ldp + col;
#pragma unroll
        for (int v = 0; v < V; ++v)
            if (col + v < a.cols) out2[v] = (double)acc2[v];
    }
}

template <typename MatT, int V>
__global__ __launch_bounds__(LPX_BLOCK) void k_colsum(ColsumArgs a)
{
    if (a.ctrl->status != LPX_RUNNING) return;
    if (a.gate_n > 0) {
        __shared__ BlockRed red;
        if (pricing_resolved(a.gate_v0, a.gate_i0, a.gate_n, red)) return;
    }
    if (a.part2 != nullptr) { // dual loop: z and the dual row in one pass
        if (a.ctrl->f32_phase) colsum_body<MatT, float, V, false, true>(a);
        else if (a.nt) colsum_body<MatT, double, V, true, true>(a);
        else colsum_body<MatT, double, V, false, true>(a);
        return;
    }
    if (a.ctrl->f32_phase) colsum_body<MatT, float, V, false>(a); // first iteration only
    else if (a.nt) colsum_body<MatT, double, V, true>(a);
    else colsum_body<MatT, double, V, false>(a);
}

// ---------------------------------------------------------------------------------------------------------
// K2b/K3: finish the column sums (fixed chunk order) and apply the stage's epilogue + workgroup partial reduce
//   FIN_Y       y_k                                                        primal_solvers.py:64 (first product)
//   FIN_PRIMAL  rc_j = c_j - z_j ; rc[basis] = 0 ; (min rc, first j: rc_j<0)   :64-66, :135-142
//   FIN_BOUNDED rc_j = +-fl32(-c_j + z_j) ; (max rc, first argmax)             special_solvers.py:177-214
//   FIN_DUAL    rc_j as FIN_PRIMAL, row_j = sum2 ; theta_j = (-rc_j)/row_j ; (min theta, first j), min row
//                                                                              dual_solvers.py:37-40, :61-80
// ---------------------------------------------------------------------------------------------------------
enum { FIN_Y = 0, FIN_PRIMAL = 1, FIN_BOUNDED = 2, FIN_DUAL = 3 };
struct FinArgs {
    int mode;
    int64_t cols, nch, ldp;
    const double *part;
    const double *part2;
    const float *c;
    const int32_t *var_state;
    double *out;  // y or rc
    double *out2; // dual row
    Partials P;
    const lpx_ctrl *ctrl;
    // staged pricing (FIN_PRIMAL): this launch covers columns [col0, col0 + cols) -- part points at column col0 --
    // and writes the pricing partials from index pbase on; gate_n > 0: neutral partials if the earlier stage decided
    int64_t col0, pbase, gate_n;
};

__global__ __launch_bounds__(LPX_BLOCK) void k_finalize(FinArgs a)
{
    __shared__ BlockRed red;
    if (a.ctrl->status != LPX_RUNNING) return;
    if (a.gate_n > 0 && pricing_resolved(a.P.v0, a.P.i0, a.gate_n, red)) {
        if (threadIdx.x == 0) {
            a.P.v0[a.pbase + blockIdx.x] = LPX_INF;
            a.P.i0[a.pbase + blockIdx.x] = LPX_NOIDX;
        }
        return;
    }
    const bool f32 = a.ctrl->f32_phase != 0;
    const int64_t lcol = (int64_t)blockIdx.x * LPX_BLOCK + threadIdx.x;
    const int64_t col = a.col0 + lcol;
    const bool in = lcol < a.cols;
    double s = 0.0, s2 = 0.0;
    if (in) {
        s = f32 ? (double)fin_sum<float>(a.part, a.nch, a.ldp, lcol) : fin_sum<double>(a.part, a.nch, a.ldp, lcol);
        if (a.mode == FIN_DUAL)
            s2 = f32 ? (double)fin_sum<float>(a.part2, a.nch, a.ldp, lcol) : fin_sum<double>(a.part2, a.nch, a.ldp, lcol);
    }
    if (a.mode == FIN_Y) {
        if (in) a.out[col] = s;
        return;
    }
    const int32_t vs = in ? a.var_state[col] : LPX_VS_EXCLUDED;
    const bool live = in && vs != LPX_VS_EXCLUDED;
    if (a.mode == FIN_PRIMAL || a.mode == FIN_DUAL) {
        double rc = 0.0;
        if (live && vs != LPX_VS_BASIC) rc = f32 ? (double)(a.c[col] - (float)s) : (double)a.c[col] - s;
        if (in) a.out[col] = rc;
        if (a.mode == FIN_PRIMAL) {
            const double mn = block_min(live ? rc : LPX_INF, red);
            const int64_t jn = block_imin((live && rc < 0.0) ? col : LPX_NOIDX, red);
            if (threadIdx.x == 0) {
                a.P.v0[a.pbase + blockIdx.x] = mn;
                a.P.i0[a.pbase + blockIdx.x] = jn;
            }
        } else {
            double row = 0.0;
            if (live && vs != LPX_VS_BASIC) row = s2;
            if (in) a.out2[col] = row;
            VI t{live ? lpx_dsd(rc, row, f32) : LPX_INF, live ? col : LPX_NOIDX};
            t = block_argmin(t, red);
            const double rmin = block_min(live ? row : LPX_INF, red);
            if (threadIdx.x == 0) {
                a.P.v0[blockIdx.x] = t.v;
                a.P.i0[blockIdx.x] = t.i;
                a.P.v1[blockIdx.x] = rmin;
            }
        }
    } else { // FIN_BOUNDED: the reference's reduced-cost array is float32 (special_solvers.py:182-196)
        double rc = 0.0;
        if (in && vs != LPX_VS_BASIC) {
            const float t = f32 ? ((-a.c[col]) + (float)s) : (float)((double)(-a.c[col]) + s);
            rc = (vs == LPX_VS_UPPER) ? (double)(-t) : (double)t;
        }
        if (in) a.out[col] = rc;
        // first argmax == argmin of (-rc, index)
        VI t{in ? -rc : LPX_INF, in ? col : LPX_NOIDX};
        t = block_argmin(t, red);
        if (threadIdx.x == 0) {
            a.P.v0[blockIdx.x] = -t.v;
            a.P.i0[blockIdx.x] = t.i;
        }
    }
}

// ---------------------------------------------------------------------------------------------------------
// K5: dual row selection (single workgroup): min(x_B), first i with x_i < 0     dual_solvers.py:47-54, :95-102
// ---------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(LPX_BLOCK) void k_dual_select(const double *xB, int64_t m, lpx_ctrl *ctrl)
{
    __shared__ BlockRed red;
    if (ctrl->status != LPX_RUNNING) return;
    double mn = LPX_INF;
    int64_t r = LPX_NOIDX;
    for (int64_t i = threadIdx.x; i < m; i += LPX_BLOCK) {
        const double x = xB[i];
        mn = fmin(mn, x);
        if (x < 0.0 && i < r) r = i;
    }
    mn = block_min(mn, red);
    r = block_imin(r, red);
    if (threadIdx.x == 0) {
        ctrl->counter += 1;
        ctrl->price_val = mn;
        if (lpx_is_optimal_min(mn)) ctrl->status = LPX_OPTIMAL;
        else ctrl->leave_r = r;
    }
}

// ---------------------------------------------------------------------------------------------------------
// K4: entering decision (prologue) + direction d = B^-1 A[:,j] + ratio-test partials
//   primal_solvers.py:110-167, dual: :234 via pivot, bounded: special_solvers.py:239-329
// One wave per row of B^-1 (coalesced 16-byte loads along the row); the column a_j is read from the
// transposed copy At (contiguous).  8 m^2 B of HBM traffic.
// ---------------------------------------------------------------------------------------------------------
enum { DIR_PRIMAL = 0, DIR_DUAL = 1, DIR_BOUNDED = 2, DIR_FORCED = 3 };
// rows per wave: 1 while the per-workgroup ratio partials fit their arrays (more, smaller workgroups fill the chip
// at m ~ 1000), more for very tall matrices
static inline int dir_rows_per_wave(int64_t m)
{
    int rpw = 1;
    while (lpx_cdiv(m, (int64_t)rpw * LPX_WAVES) > LPX_MAXP) rpw *= 2;
    return rpw;
}

struct DirArgs {
    lpx_revised s;
    int kind;
    int64_t forced_j;
    int64_t nprice; // number of pricing partials
    int rpw;        // rows per wave
    int nt;         // B^-1 larger than the Infinity Cache: non-temporal loads
    Partials price, ratio;
    double *d;
};

template <typename AccT, bool NT = false>
__device__ __forceinline__ double dir_row(const double *Bi, const float *aj, int64_t m)
{
    const int lane = threadIdx.x & 63;
    AccT acc0 = (AccT)0, acc1 = (AccT)0;
    const int64_t m2 = m & ~(int64_t)1;
    for (int64_t k = lane * 2; k < m2; k += 128) {
        const d2_t bv = lpx_stream_load<NT>((const d2_t *)(Bi + k));
        const f2_t av = *(const f2_t *)(aj + k);
        acc0 = fma((AccT)bv.x, (AccT)av.x, acc0);
        acc1 = fma((AccT)bv.y, (AccT)av.y, acc1);
    }
    if ((m & 1) && lane == 0) acc0 = fma((AccT)Bi[m - 1], (AccT)aj[m - 1], acc0);
    return (double)wave_sum<AccT>(acc0 + acc1);
}

__global__ __launch_bounds__(LPX_BLOCK) void k_direction(DirArgs a)
{
    __shared__ BlockRed red;
    __shared__ double sv0[LPX_WAVES], sv1[LPX_WAVES], sv2[LPX_WAVES];
    __shared__ int64_t si0[LPX_WAVES], si1[LPX_WAVES];
    lpx_ctrl *ctrl = a.s.ctrl;
    if (ctrl->status != LPX_RUNNING) return;
    const bool f32 = ctrl->f32_phase != 0;
    const bool lead = (blockIdx.x == 0 && threadIdx.x == 0);
    int64_t j = a.forced_j;
    double sign = 1.0;

    if (a.kind == DIR_PRIMAL) {
        const double mn = combine_min(a.price.v0, a.nprice, red);
        j = combine_imin(a.price.i0, a.nprice, red);
        const bool opt = lpx_is_optimal_min(mn);
        if (lead) {
            ctrl->counter += 1;
            ctrl->price_val = mn;
            if (opt) ctrl->status = LPX_OPTIMAL;
            else ctrl->enter_j = j;
        }
        if (opt) return;
    } else if (a.kind == DIR_BOUNDED) {
        // first argmax over the partials: value descending, index ascending
        VI b{LPX_INF, LPX_NOIDX};
        for (int64_t k = threadIdx.x; k < a.nprice; k += blockDim.x)
            if (vi_less(-a.price.v0[k], a.price.i0[k], b.v, b.i)) { b.v = -a.price.v0[k]; b.i = a.price.i0[k]; }
        b = block_argmin(b, red);
        const double mx = -b.v;
        j = b.i;
        const bool opt = lpx_is_optimal_max(mx);
        int32_t vs = LPX_VS_BASIC;
        if (!opt) vs = a.s.var_state[j];
        const bool bad = !opt && vs != LPX_VS_LOWER && vs != LPX_VS_UPPER;
        if (lead) {
            ctrl->counter += 1;
            ctrl->price_val = mx;
            if (opt) ctrl->status = LPX_OPTIMAL;
            else if (bad) ctrl->status = LPX_NOT_NONBASIC;
            else ctrl->enter_j = j;
        }
        if (opt || bad) return;
        sign = (vs == LPX_VS_LOWER) ? 1.0 : -1.0;
    } else if (a.kind == DIR_DUAL) {
        const VI t = combine_argmin(a.price.v0, a.price.i0, a.nprice, red);
        const double rmin = combine_min(a.price.v1, a.nprice, red);
        j = t.i;
        const bool unb = rmin >= 0.0; // dual_solvers.py:56-59
        if (lead) {
            ctrl->ratio_val = t.v;
            ctrl->aux_val = rmin;
            if (unb) ctrl->status = LPX_DUAL_UNBOUNDED;
            else ctrl->enter_j = j;
        }
        if (unb) return;
    } else {
        if (lead) ctrl->enter_j = j;
    }

    const int64_t m = a.s.m;
    const float *aj = a.s.At + j * a.s.ldAt;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int64_t rbase = ((int64_t)blockIdx.x * LPX_WAVES + wave) * a.rpw;
    VI t1{LPX_INF, LPX_NOIDX}, t2{LPX_INF, LPX_NOIDX};
    double dmax = -LPX_INF;
    for (int q = 0; q < a.rpw; ++q) {
        const int64_t i = rbase + q;
        if (i >= m) break; // wave-uniform
        const double *Bi = a.s.Binv + i * a.s.ldB;
        const double di = f32 ? dir_row<float>(Bi, aj, m) : (a.nt ? dir_row<double, true>(Bi, aj, m) : dir_row<double>(Bi, aj, m));
        if (lane == 0) a.d[i] = di;
        if (a.kind == DIR_PRIMAL) {
            const double th = lpx_psd(a.s.xB[i], di, f32);
            if (vi_less(th, i, t1.v, t1.i)) { t1.v = th; t1.i = i; }
            dmax = fmax(dmax, di);
        } else if (a.kind == DIR_BOUNDED) {
            const int64_t bi = a.s.basis[i];
            const double x = a.s.xB[i];
            const double g1 = lpx_psd(x - a.s.lb[bi], sign * di, false);
            const double g2 = lpx_psd(a.s.ub[bi] - x, -sign * di, false);
            if (vi_less(g1, i, t1.v, t1.i)) { t1.v = g1; t1.i = i; }
            if (vi_less(g2, i, t2.v, t2.i)) { t2.v = g2; t2.i = i; }
        }
    }
    if (a.kind == DIR_PRIMAL || a.kind == DIR_BOUNDED) {
        // every lane of a wave holds the same values: combine the 4 waves through LDS
        if (lane == 0) {
            sv0[wave] = t1.v; si0[wave] = t1.i;
            sv1[wave] = t2.v; si1[wave] = t2.i;
            sv2[wave] = dmax;
        }
        __syncthreads();
        if (threadIdx.x == 0) {
            for (int w = 1; w < LPX_WAVES; ++w) {
                if (vi_less(sv0[w], si0[w], sv0[0], si0[0])) { sv0[0] = sv0[w]; si0[0] = si0[w]; }
                if (vi_less(sv1[w], si1[w], sv1[0], si1[0])) { sv1[0] = sv1[w]; si1[0] = si1[w]; }
                sv2[0] = fmax(sv2[0], sv2[w]);
            }
            a.ratio.v0[blockIdx.x] = sv0[0];
            a.ratio.i0[blockIdx.x] = si0[0];
            if (a.kind == DIR_PRIMAL) {
                a.ratio.v1[blockIdx.x] = sv2[0];
            } else {
                a.ratio.v1[blockIdx.x] = sv1[0];
                a.ratio.i1[blockIdx.x] = si1[0];
            }
        }
    }
}

// ---------------------------------------------------------------------------------------------------------
// K1 prologue ("prepare"): leaving decision, eta column, copy of the old pivot row, bookkeeping
//   primal_solvers.py:130-132, :165-166, :235-244, :261 ; bounded: special_solvers.py:252-339
// ---------------------------------------------------------------------------------------------------------
enum { PREP_PRIMAL = 0, PREP_DUAL = 1, PREP_BOUNDED = 2, PREP_FORCED = 3 };
struct PrepArgs {
    lpx_revised s;
    int kind;
    int naive; // naive solvers: basis change only (no eta / pivot-row copy), the state stays float32
    int64_t forced_r;
    int64_t nratio;
    Partials ratio;
    const double *d;
    double *e, *rowr, *scal;
};

__global__ __launch_bounds__(LPX_BLOCK) void k_prepare(PrepArgs a)
{
    __shared__ BlockRed red;
    lpx_ctrl *ctrl = a.s.ctrl;
    if (ctrl->status != LPX_RUNNING) return;
    const bool lead = (blockIdx.x == 0 && threadIdx.x == 0);
    const int64_t m = a.s.m;
    const int64_t j = ctrl->enter_j;
    int64_t r = a.forced_r;
    double theta = 0.0;
    int32_t leave_state = LPX_VS_LOWER;
    if (a.kind == PREP_PRIMAL) {
        const VI t = combine_argmin(a.ratio.v0, a.ratio.i0, a.nratio, red);
        const double dmax = combine_max(a.ratio.v1, a.nratio, red);
        r = t.i;
        theta = t.v;
        if (dmax <= 0.0) { // primal_solvers.py:144-147
            if (lead) {
                ctrl->aux_val = dmax;
                ctrl->status = LPX_PRIMAL_UNBOUNDED;
            }
            return;
        }
        if (lead) ctrl->aux_val = dmax;
    } else if (a.kind == PREP_DUAL) {
        r = ctrl->leave_r;
    } else if (a.kind == PREP_BOUNDED) {
        const VI t1 = combine_argmin(a.ratio.v0, a.ratio.i0, a.nratio, red);
        const VI t2 = combine_argmin(a.ratio.v1, a.ratio.i1, a.nratio, red);
        const double g3 = a.s.ub[j] - a.s.lb[j];
        const double delta = fmin(t1.v, t2.v);
        if (delta == LPX_INF) { // special_solvers.py:256-257 / :302-303
            if (lead) ctrl->status = LPX_PRIMAL_UNBOUNDED;
            return;
        }
        if (g3 <= delta) { // bound flip, special_solvers.py:264-268 / :309-313
            if (lead) {
                const int32_t vs = a.s.var_state[j];
                a.s.var_state[j] = (vs == LPX_VS_LOWER) ? LPX_VS_UPPER : LPX_VS_LOWER;
                if (a.s.log && ctrl->pivots < a.s.log_cap) {
                    a.s.log[3 * ctrl->pivots + 0] = -1;
                    a.s.log[3 * ctrl->pivots + 1] = j;
                    a.s.log[3 * ctrl->pivots + 2] = -1;
                }
                ctrl->pivots += 1;
                ctrl->flip = 1;
                ctrl->leave_r = -1;
                ctrl->leave_var = -1;
                ctrl->ratio_val = g3;
            }
            return;
        }
        if (t1.v < t2.v) { r = t1.i; leave_state = LPX_VS_LOWER; theta = t1.v; } // :271-277
        else { r = t2.i; leave_state = LPX_VS_UPPER; theta = t2.v; }              // :278-284
    }

    const double dr = a.d[r];
    const int64_t gi = (int64_t)blockIdx.x * LPX_BLOCK + threadIdx.x;
    if (!a.naive && gi < m) a.e[gi] = (gi == r) ? 1.0 / dr : (-a.d[gi]) / dr; // primal_solvers.py:236-244
    if (!a.naive && gi < lpx_roundup_dev(m, 2)) a.rowr[gi] = a.s.Binv[r * a.s.ldB + gi]; // K1 reads 2-wide: even length <= ldB
    if (lead) {
        const int64_t leaving = a.s.basis[r];
        a.scal[0] = a.s.xB[r];
        a.s.basis[r] = j; // primal_solvers.py:261
        a.s.var_state[leaving] = leave_state;
        a.s.var_state[j] = LPX_VS_BASIC;
        if (a.s.log && ctrl->pivots < a.s.log_cap) {
            a.s.log[3 * ctrl->pivots + 0] = r;
            a.s.log[3 * ctrl->pivots + 1] = j;
            a.s.log[3 * ctrl->pivots + 2] = leaving;
        }
        ctrl->pivots += 1;
        ctrl->leave_r = r;
        ctrl->leave_var = leaving;
        ctrl->leave_state = leave_state;
        ctrl->ratio_val = theta;
        ctrl->flip = 0;
        if (!a.naive) ctrl->f32_phase = 0; // E is float64 (np.eye): everything is float64 from here on
    }
}

// bounded: clear the flip flag at the start of an iteration (single thread; the previous iteration's update
// kernel has consumed it)
__global__ void k_clear_flip(lpx_ctrl *ctrl)
{
    if (ctrl->status == LPX_RUNNING) ctrl->flip = 0;
}

// ---------------------------------------------------------------------------------------------------------
// bounded x_B recomputation  special_solvers.py:169-175:  w = b - A_lb lb - A_ub ub ; x_B = B^-1 w
// ---------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(LPX_BLOCK) void k_bounded_w(lpx_revised s, double *w)
{
    if (s.ctrl->status != LPX_RUNNING && s.ctrl->status != LPX_OPTIMAL) return;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int64_t i = (int64_t)blockIdx.x * LPX_WAVES + wave;
    if (i >= s.m) return;
    const float *Ai = s.A + i * s.ldA;
    double accl = 0.0, accu = 0.0;
    for (int64_t j = lane; j < s.n; j += 64) {
        const int32_t vs = s.var_state[j];
        if (vs == LPX_VS_LOWER) accl = fma((double)Ai[j], s.lb[j], accl);
        else if (vs == LPX_VS_UPPER) accu = fma((double)Ai[j], s.ub[j], accu);
    }
    accl = wave_sum<double>(accl);
    accu = wave_sum<double>(accu);
    if (lane == 0) w[i] = ((double)s.b[i] - accl) - accu;
}
__global__ __launch_bounds__(LPX_BLOCK) void k_rowdot(const double *M, int64_t ld, int64_t rows, int64_t cols,
                                                      const double *v, double *out, const lpx_ctrl *ctrl)
{
    if (ctrl->status != LPX_RUNNING && ctrl->status != LPX_OPTIMAL) return;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int64_t i = (int64_t)blockIdx.x * LPX_WAVES + wave;
    if (i >= rows) return;
    const double *Mi = M + i * ld;
    double acc = 0.0;
    for (int64_t k = lane; k < cols; k += 64) acc = fma(Mi[k], v[k], acc);
    acc = wave_sum<double>(acc);
    if (lane == 0) out[i] = acc;
}

// ---------------------------------------------------------------------------------------------------------
// result expansion  primal_solvers.py:72-79 / special_solvers.py:201-206
// ---------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(LPX_BLOCK) void k_expand_fill(lpx_revised s, double *x)
{
    const int64_t j = (int64_t)blockIdx.x * LPX_BLOCK + threadIdx.x;
    if (j >= s.n) return;
    double v = 0.0;
    if (s.lb != nullptr) {
        const int32_t vs = s.var_state[j];
        if (vs == LPX_VS_LOWER) v = s.lb[j];
        else if (vs == LPX_VS_UPPER) v = s.ub[j];
    }
    x[j] = v;
}
__global__ __launch_bounds__(LPX_BLOCK) void k_expand_scatter(const double *xB, const int64_t *basis, int64_t m,
                                                              double *x)
{
    const int64_t i = (int64_t)blockIdx.x * LPX_BLOCK + threadIdx.x;
    if (i < m) x[basis[i]] = xB[i];
}
// ---------------------------------------------------------------------------------------------------------
// Naive solvers: float32 re-inversion of the basis after every pivot (primal_solvers.py:92-94) and
// bfs = inv_basis_matrix @ b (:68-70).  Gauss-Jordan with partial pivoting on W = [B | I] (m x 2m float32);
// per elimination step one single-workgroup kernel (pivot search, row swap, pivot-row scaling, multiplier column)
// and one multi-workgroup rank-1 elimination.  The reference re-inverts with LAPACK; this is its O(m^3) slow
// variant and not a performance path -- correctness (sequence parity) is the bar.
// ---------------------------------------------------------------------------------------------------------
struct NaiveWs {
    float *W, *prow, *colk;
    int64_t ldw, bytes;
};
static NaiveWs naive_ws(int64_t m, void *base)
{
    NaiveWs w;
    char *p = (char *)base;
    w.ldw = lpx_roundup(2 * m, 4);
    w.W = (float *)carve(p, 4 * m * w.ldw);
    w.prow = (float *)carve(p, 4 * w.ldw);
    w.colk = (float *)carve(p, 4 * (m + 4));
    w.bytes = (int64_t)(p - (char *)base);
    return w;
}
extern "C" int64_t lpx_naive_workspace_bytes(int64_t m) { return m > 0 ? naive_ws(m, nullptr).bytes : 0; }

__global__ __launch_bounds__(LPX_BLOCK) void k_inv_init(lpx_revised s, NaiveWs w)
{
    if (s.ctrl->status != LPX_RUNNING) return;
    const int64_t idx = (int64_t)blockIdx.x * LPX_BLOCK + threadIdx.x;
    const int64_t m = s.m;
    if (idx >= m * 2 * m) return;
    const int64_t i = idx / (2 * m), k = idx - i * 2 * m;
    w.W[i * w.ldw + k] = (k < m) ? s.At[s.basis[k] * s.ldAt + i] : ((k - m) == i ? 1.f : 0.f);
}
__global__ __launch_bounds__(1024) void k_inv_pivot(lpx_revised s, NaiveWs w, int64_t k)
{
    __shared__ BlockRed red;
    if (s.ctrl->status != LPX_RUNNING) return;
    const int64_t m = s.m;
    VI best{LPX_INF, LPX_NOIDX}; // argmax |W[i][k]|, i >= k, first on ties
    for (int64_t i = k + threadIdx.x; i < m; i += blockDim.x) {
        const double v = -(double)fabsf(w.W[i * w.ldw + k]);
        if (vi_less(v, i, best.v, best.i)) { best.v = v; best.i = i; }
    }
    best = block_argmin(best, red);
    const int64_t p = best.i;
    const float piv = w.W[p * w.ldw + k];
    // swap rows k and p, scale the pivot row; keep it in prow
    for (int64_t c = threadIdx.x; c < 2 * m; c += blockDim.x) {
        const float a = w.W[p * w.ldw + c], b = w.W[k * w.ldw + c];
        const float pr = a / piv;
        w.prow[c] = pr;
        w.W[k * w.ldw + c] = pr;
        if (p != k) w.W[p * w.ldw + c] = b;
    }
    __syncthreads();
    // multiplier column (after the swap), read before the elimination overwrites it
    for (int64_t i = threadIdx.x; i < m; i += blockDim.x) w.colk[i] = (i == k) ? 0.f : w.W[i * w.ldw + k];
}
__global__ __launch_bounds__(LPX_BLOCK) void k_inv_elim(lpx_revised s, NaiveWs w, int64_t k)
{
    if (s.ctrl->status != LPX_RUNNING) return;
    const int64_t c = (int64_t)blockIdx.x * LPX_BLOCK + threadIdx.x;
    const int64_t i = blockIdx.y;
    if (c >= 2 * s.m || i == k) return;
    const float f = w.colk[i];
    w.W[i * w.ldw + c] = fmaf(-f, w.prow[c], w.W[i * w.ldw + c]);
}
__global__ __launch_bounds__(LPX_BLOCK) void k_inv_finish(lpx_revised s, NaiveWs w)
{
    if (s.ctrl->status != LPX_RUNNING) return;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int64_t i = (int64_t)blockIdx.x * LPX_WAVES + wave;
    const int64_t m = s.m;
    if (i >= m) return;
    float acc = 0.f; // bfs = inv_basis_matrix @ b in float32
    for (int64_t k = lane; k < m; k += 64) {
        const float v = w.W[i * w.ldw + m + k];
        s.Binv[i * s.ldB + k] = (double)v;
        acc = fmaf(v, s.b[k], acc);
    }
    acc = wave_sum<float>(acc);
    if (lane == 0) s.xB[i] = (double)acc;
}
static int launch_reinvert(const lpx_revised *s, hipStream_t st)
{
    LPX_REQUIRE(s->naive_workspace != nullptr && s->naive_workspace_bytes >= lpx_naive_workspace_bytes(s->m),
                "naive solvers need naive_workspace (lpx_naive_workspace_bytes)");
    LPX_REQUIRE(s->m <= 65535, "m too large for the re-inversion grid");
    const NaiveWs w = naive_ws(s->m, s->naive_workspace);
    const int64_t m = s->m;
    hipLaunchKernelGGL(k_inv_init, dim3((unsigned)lpx_cdiv(2 * m * m, LPX_BLOCK)), dim3(LPX_BLOCK), 0, st, *s, w);
    for (int64_t k = 0; k < m; ++k) {
        hipLaunchKernelGGL(k_inv_pivot, dim3(1), dim3(1024), 0, st, *s, w, k);
        hipLaunchKernelGGL(k_inv_elim, dim3((unsigned)lpx_cdiv(2 * m, LPX_BLOCK), (unsigned)m), dim3(LPX_BLOCK), 0, st, *s, w, k);
    }
    hipLaunchKernelGGL(k_inv_finish, dim3((unsigned)lpx_cdiv(m, LPX_WAVES)), dim3(LPX_BLOCK), 0, st, *s, w);
    return LPX_OK;
}

// ---------------------------------------------------------------------------------------------------------
// host-side drivers
// ---------------------------------------------------------------------------------------------------------
static int check_revised(const lpx_revised *s)
{
    LPX_REQUIRE(s != nullptr, "null state");
    LPX_REQUIRE(s->m > 0 && s->n > 0, "empty problem");
    LPX_REQUIRE(s->ldA >= s->n && (s->ldA % 4) == 0, "ldA must be >= n and a multiple of 4");
    LPX_REQUIRE(s->ldAt >= s->m && (s->ldAt % 4) == 0, "ldAt must be >= m and a multiple of 4");
    LPX_REQUIRE(s->ldB >= s->m && (s->ldB % 2) == 0, "ldB must be >= m and a multiple of 2");
    LPX_REQUIRE(s->A && s->At && s->b && s->c && s->Binv && s->xB && s->basis && s->var_state && s->ctrl,
                "null device pointer");
    LPX_REQUIRE(lpx_aligned16(s->A) && lpx_aligned16(s->At) && lpx_aligned16(s->Binv) && lpx_aligned16(s->workspace),
                "A, At, Binv and workspace must be 16-byte aligned");
    LPX_REQUIRE(s->workspace != nullptr && s->workspace_bytes >= lpx_revised_workspace_bytes(s->m, s->n),
                "workspace too small (see lpx_revised_workspace_bytes)");
    LPX_REQUIRE(lpx_cdiv(s->n, LPX_BLOCK) <= LPX_MAXP, "problem too large for the partial arrays (n <= 524288)");
    return LPX_OK;
}

// dynamic LDS for the row-chunk weights, 0 if the chunk is too tall (then weights are read in the loop)
static inline size_t colsum_lds_bytes(int64_t rpc) { return (rpc * 8 <= 48 * 1024) ? (size_t)rpc * 8 : 0; }

static void launch_colsum_B(const lpx_revised *s, const RevisedWs &w, hipStream_t st)
{
    ColsumArgs a{};
    a.M = s->Binv; a.ld = s->ldB; a.rows = s->m; a.cols = s->m; a.rpc = w.py.rpc;
    a.wmode = W_CB; a.c = s->c; a.basis = s->basis;
    a.part = w.ypart; a.ldp = w.ldyp; a.ctrl = s->ctrl;
    const size_t lds = colsum_lds_bytes(w.py.rpc);
    a.lds_weights = lds != 0;
    a.nt = lpx_stream_nt(8 * s->m * s->ldB);
    hipLaunchKernelGGL((k_colsum<double, 2>), dim3((unsigned)w.py.nstrips, (unsigned)w.py.nch), dim3(w.py.threads), lds, st, a);
}
// columns [col0, col0 + cols) of A with plan p; gate_n > 0: skipped when the first gate_n pricing partials decide
static void launch_colsum_A_range(const lpx_revised *s, const RevisedWs &w, const ColsumPlan &p, int wmode, double *part,
                                  int64_t col0, int64_t cols, int64_t gate_n, hipStream_t st, double *part2 = nullptr)
{
    ColsumArgs a{};
    a.M = s->A + col0; a.ld = s->ldA; a.rows = s->m; a.cols = cols; a.rpc = p.rpc;
    a.wmode = wmode; a.wvec = w.y; a.Binv = s->Binv; a.ldB = s->ldB;
    a.part = part; a.ldp = w.ldzp; a.ctrl = s->ctrl;
    a.ypart = w.ypart; a.ynch = w.py.nch; a.yld = w.ldyp;
    a.gate_v0 = w.price.v0; a.gate_i0 = w.price.i0; a.gate_n = gate_n;
    size_t lds = colsum_lds_bytes(p.rpc);
    a.lds_weights = lds != 0;
    a.part2 = part2;
    if (part2 != nullptr) lds *= 2; // callers check revised_dual_one_pass
    a.nt = lpx_stream_nt(4 * s->m * s->ldA);
    hipLaunchKernelGGL((k_colsum<float, 4>), dim3((unsigned)p.nstrips, (unsigned)p.nch), dim3(p.threads), lds, st, a);
}
// dual loop: z = y A and the dual row B^-1[r,:] A from one pass over A (weights of both in LDS)
static inline bool revised_dual_one_pass(const RevisedWs &w)
{
    return lpx_env_int("LPX_DUAL_ONE_PASS", 1) != 0 && colsum_lds_bytes(w.pa.rpc) != 0 && 2 * w.pa.rpc * 8 <= 48 * 1024;
}
static void launch_colsum_A(const lpx_revised *s, const RevisedWs &w, int wmode, double *part, hipStream_t st)
{
    launch_colsum_A_range(s, w, w.pa, wmode, part, 0, s->n, 0, st);
}
static void launch_fin(const lpx_revised *s, const RevisedWs &w, int mode, double *rc_out, hipStream_t st)
{
    FinArgs a{};
    a.mode = mode; a.ctrl = s->ctrl; a.c = s->c; a.var_state = s->var_state; a.P = w.price;
    if (mode == FIN_Y) {
        a.cols = s->m; a.nch = w.py.nch; a.ldp = w.ldyp; a.part = w.ypart; a.out = w.y;
    } else {
        a.cols = s->n; a.nch = w.pa.nch; a.ldp = w.ldzp; a.part = w.zpart; a.part2 = w.zpart2;
        a.out = rc_out ? rc_out : w.rc; a.out2 = w.row;
    }
    hipLaunchKernelGGL(k_finalize, dim3((unsigned)lpx_cdiv(a.cols, LPX_BLOCK)), dim3(LPX_BLOCK), 0, st, a);
}
// Staged pricing of the primal loop (large A only): z and rc for the first eighth of the columns, then -- gated on the
// device by pricing_resolved -- for the rest.  Decisions are the reference's exactly; rc of skipped columns is not
// produced (nothing in the loop reads it).  4 launches, the last two usually empty.
static inline bool revised_staged_pricing(const RevisedWs &w)
{
    return w.n1 > 0 && lpx_env_int("LPX_STAGED_PRICING", 1) != 0 && colsum_lds_bytes(w.p1.rpc) != 0 &&
           colsum_lds_bytes(w.p2.rpc) != 0;
}
static void launch_fin_primal_range(const lpx_revised *s, const RevisedWs &w, const ColsumPlan &p, const double *part,
                                    int64_t col0, int64_t cols, int64_t gate_n, hipStream_t st)
{
    FinArgs a{};
    a.mode = FIN_PRIMAL; a.ctrl = s->ctrl; a.c = s->c; a.var_state = s->var_state; a.P = w.price;
    a.cols = cols; a.nch = p.nch; a.ldp = w.ldzp; a.part = part; a.out = w.rc;
    a.col0 = col0; a.pbase = col0 / LPX_BLOCK; a.gate_n = gate_n;
    hipLaunchKernelGGL(k_finalize, dim3((unsigned)lpx_cdiv(cols, LPX_BLOCK)), dim3(LPX_BLOCK), 0, st, a);
}
static void launch_staged_pricing(const lpx_revised *s, const RevisedWs &w, hipStream_t st)
{
    const int64_t nb1 = w.n1 / LPX_BLOCK; // n1 is a whole number of 1024-column strips
    launch_colsum_A_range(s, w, w.p1, W_YPART, w.zpart1, 0, w.n1, 0, st);
    launch_fin_primal_range(s, w, w.p1, w.zpart1, 0, w.n1, 0, st);
    launch_colsum_A_range(s, w, w.p2, W_YPART, w.zpart + w.n1, w.n1, s->n - w.n1, nb1, st);
    launch_fin_primal_range(s, w, w.p2, w.zpart + w.n1, w.n1, s->n - w.n1, nb1, st);
}
// y = c_B B^-1 (row-chunk partials), z = y A with y finished inside the second launch (2 launches)
static void launch_pricing_sums(const lpx_revised *s, const RevisedWs &w, hipStream_t st)
{
    launch_colsum_B(s, w, st);
    if (colsum_lds_bytes(w.pa.rpc) != 0) {
        launch_colsum_A(s, w, W_YPART, w.zpart, st);
    } else { // very tall matrices: finish y in its own launch
        launch_fin(s, w, FIN_Y, nullptr, st);
        launch_colsum_A(s, w, W_VEC, w.zpart, st);
    }
}
static void launch_direction(const lpx_revised *s, const RevisedWs &w, int kind, int64_t forced_j, hipStream_t st)
{
    DirArgs a{};
    a.s = *s; a.kind = kind; a.forced_j = forced_j; a.nprice = lpx_cdiv(s->n, LPX_BLOCK);
    a.rpw = dir_rows_per_wave(s->m);
    a.nt = lpx_stream_nt(8 * s->m * s->ldB);
    a.price = w.price; a.ratio = w.ratio; a.d = w.d;
    hipLaunchKernelGGL(k_direction, dim3((unsigned)lpx_cdiv(s->m, (int64_t)a.rpw * LPX_WAVES)), dim3(LPX_BLOCK), 0, st, a);
}
static void launch_prepare(const lpx_revised *s, const RevisedWs &w, int kind, int64_t forced_r, hipStream_t st,
                           int naive = 0)
{
    PrepArgs a{};
    a.s = *s; a.kind = kind; a.naive = naive; a.forced_r = forced_r; a.nratio = lpx_cdiv(s->m, (int64_t)dir_rows_per_wave(s->m) * LPX_WAVES);
    a.ratio = w.ratio; a.d = w.d; a.e = w.e; a.rowr = w.rowr; a.scal = w.scal;
    hipLaunchKernelGGL(k_prepare, dim3((unsigned)lpx_cdiv(lpx_roundup(s->m, 2), LPX_BLOCK)), dim3(LPX_BLOCK), 0, st, a);
}
static void launch_update(const lpx_revised *s, const RevisedWs &w, bool update_xB, hipStream_t st)
{
    Rank1Args a{};
    a.M = s->Binv; a.rows = s->m; a.cols = s->m; a.ld = s->ldB; a.e = w.e; a.rowr = w.rowr; a.r = -1;
    a.ctrl = s->ctrl; a.xB = update_xB ? s->xB : nullptr; a.xr_old = w.scal; a.col0 = 0;
    const Rank1Plan p = lpx_rank1_plan(s->m, s->m);
    lpx_k1_begin(st);
    lpx_rank1_launch<false>(p, a, st);
    lpx_k1_end(st);
}
// K1 that also leaves the next iteration's y partials (k_rank1_y): grid and chunking are the y pass's
static inline bool revised_fuse_y(const RevisedWs &w)
{
    return lpx_env_int("LPX_FUSE_Y", 1) != 0 && colsum_lds_bytes(w.py.rpc) != 0 && colsum_lds_bytes(w.pa.rpc) != 0;
}
static void launch_update_y(const lpx_revised *s, const RevisedWs &w, bool update_xB, hipStream_t st)
{
    Rank1Args a{};
    a.M = s->Binv; a.rows = s->m; a.cols = s->m; a.ld = s->ldB; a.e = w.e; a.rowr = w.rowr; a.r = -1;
    a.ctrl = s->ctrl; a.xB = update_xB ? s->xB : nullptr; a.xr_old = w.scal; a.col0 = 0;
    a.rows_per_block = w.py.rpc;
    const Rank1YArgs f{s->c, s->basis, w.ypart, w.ldyp};
    const dim3 grid((unsigned)w.py.nstrips, (unsigned)w.py.nch);
    const size_t lds = colsum_lds_bytes(w.py.rpc);
    lpx_k1_begin(st);
    // unroll 2 / 4 / 8 and non-temporal loads / stores / both are within 3 % of each other at m = 10000
    if (lpx_stream_nt(8 * s->m * s->ldB)) hipLaunchKernelGGL((k_rank1_y<1, 4>), grid, dim3(w.py.threads), lds, st, a, f);
    else hipLaunchKernelGGL((k_rank1_y<0, 4>), grid, dim3(w.py.threads), lds, st, a, f);
    lpx_k1_end(st);
}
static void launch_bounded_bfs(const lpx_revised *s, const RevisedWs &w, hipStream_t st)
{
    hipLaunchKernelGGL(k_bounded_w, dim3((unsigned)lpx_cdiv(s->m, LPX_WAVES)), dim3(LPX_BLOCK), 0, st, *s, w.w);
    hipLaunchKernelGGL(k_rowdot, dim3((unsigned)lpx_cdiv(s->m, LPX_WAVES)), dim3(LPX_BLOCK), 0, st,
                       (const double *)s->Binv, s->ldB, s->m, s->m, (const double *)w.w, s->xB,
                       (const lpx_ctrl *)s->ctrl);
}

extern "C" int lpx_revised_iterate(const lpx_revised *s, int kind, int64_t iters, void *stream)
{
    int rc = check_revised(s);
    if (rc) return rc;
    LPX_REQUIRE(kind >= LPX_KIND_PRIMAL && kind <= LPX_KIND_DUAL_NAIVE, "bad kind");
    if (kind == LPX_KIND_PRIMAL_NAIVE || kind == LPX_KIND_DUAL_NAIVE) {
        LPX_REQUIRE(s->naive_workspace != nullptr && s->naive_workspace_bytes >= lpx_naive_workspace_bytes(s->m),
                    "naive solvers need naive_workspace (lpx_naive_workspace_bytes)");
        const RevisedWs wn = revised_ws(s->m, s->n, s->workspace);
        hipStream_t st = (hipStream_t)stream;
        for (int64_t it = 0; it < iters; ++it) {
            if (kind == LPX_KIND_PRIMAL_NAIVE) {
                launch_pricing_sums(s, wn, st);
                launch_fin(s, wn, FIN_PRIMAL, nullptr, st);
                launch_direction(s, wn, DIR_PRIMAL, -1, st);
                launch_prepare(s, wn, PREP_PRIMAL, -1, st, 1);
            } else {
                hipLaunchKernelGGL(k_dual_select, dim3(1), dim3(LPX_BLOCK), 0, st, (const double *)s->xB, s->m, s->ctrl);
                launch_colsum_A(s, wn, W_BINV_ROW, wn.zpart2, st);
                launch_pricing_sums(s, wn, st);
                launch_fin(s, wn, FIN_DUAL, nullptr, st);
                launch_direction(s, wn, DIR_DUAL, -1, st);
                launch_prepare(s, wn, PREP_DUAL, -1, st, 1);
            }
            int rr = launch_reinvert(s, st); // primal_solvers.py:169-174
            if (rr) return rr;
        }
        LPX_LAUNCH_CHECK();
        return LPX_OK;
    }
    LPX_REQUIRE(kind != LPX_KIND_BOUNDED || (s->lb && s->ub), "bounded solver needs lb/ub");
    const RevisedWs w = revised_ws(s->m, s->n, s->workspace);
    hipStream_t st0 = (hipStream_t)stream;
    const lpx_revised sv = *s;
    const lpx_revised *sp = &sv;
    // fuse: the y partials of an iteration come from the previous iteration's K1 (k_rank1_y); only the first iteration
    // of a call computes them itself (the caller may have changed B^-1 or the basis in between).  A bound flip or a
    // terminal status skips K1 and leaves B^-1, the basis and therefore the partials as they are.
    const bool fuse = revised_fuse_y(w);
    auto sums = [&](hipStream_t st) {
        if (fuse) launch_colsum_A(sp, w, W_YPART, w.zpart, st);
        else launch_pricing_sums(sp, w, st);
    };
    auto update = [&](bool update_xB, hipStream_t st) {
        if (fuse) launch_update_y(sp, w, update_xB, st);
        else launch_update(sp, w, update_xB, st);
    };
    const bool staged = fuse && kind == LPX_KIND_PRIMAL && revised_staged_pricing(w);
    auto one = [&](hipStream_t st) {
        if (kind == LPX_KIND_PRIMAL) {
            if (staged) {
                launch_staged_pricing(sp, w, st);
            } else {
                sums(st);
                launch_fin(sp, w, FIN_PRIMAL, nullptr, st);
            }
            launch_direction(sp, w, DIR_PRIMAL, -1, st);
            launch_prepare(sp, w, PREP_PRIMAL, -1, st);
            update(true, st);
        } else if (kind == LPX_KIND_DUAL) {
            hipLaunchKernelGGL(k_dual_select, dim3(1), dim3(LPX_BLOCK), 0, st, (const double *)sp->xB, sp->m, sp->ctrl);
            if (fuse && revised_dual_one_pass(w)) {
                launch_colsum_A_range(sp, w, w.pa, W_YPART, w.zpart, 0, sp->n, 0, st, w.zpart2);
            } else {
                launch_colsum_A(sp, w, W_BINV_ROW, w.zpart2, st);
                sums(st);
            }
            launch_fin(sp, w, FIN_DUAL, nullptr, st);
            launch_direction(sp, w, DIR_DUAL, -1, st);
            launch_prepare(sp, w, PREP_DUAL, -1, st);
            update(true, st);
        } else {
            hipLaunchKernelGGL(k_clear_flip, dim3(1), dim3(1), 0, st, sp->ctrl);
            sums(st);
            launch_fin(sp, w, FIN_BOUNDED, nullptr, st);
            launch_direction(sp, w, DIR_BOUNDED, -1, st);
            launch_prepare(sp, w, PREP_BOUNDED, -1, st);
            update(false, st);
            launch_bounded_bfs(sp, w, st);
        }
    };
    struct {
        lpx_revised s;
        int kind, vpt, unroll, nt, threads, fuse;
        int64_t rpb, n1, ych, ach; // everything the captured launches depend on besides the state itself
    } key;
    memset(&key, 0, sizeof key);
    const Rank1Plan plan = lpx_rank1_plan(s->m, s->m);
    key.s = sv; key.kind = kind; key.vpt = plan.vpt; key.unroll = plan.unroll; key.nt = plan.nt;
    key.threads = plan.threads; key.rpb = plan.rows_per_block; key.fuse = fuse + 2 * staged + 4 * (int)revised_dual_one_pass(w);
    key.n1 = w.n1; key.ych = w.py.nch; key.ach = w.pa.nch;
    if (iters <= 0) return LPX_OK;
    if (fuse) launch_colsum_B(sp, w, st0);
    return lpx_run_iterations(&key, sizeof key, iters, st0, one);
}

extern "C" int lpx_revised_pivot(const lpx_revised *s, int64_t row, int64_t col, void *stream)
{
    int rc = check_revised(s);
    if (rc) return rc;
    LPX_REQUIRE(row >= 0 && row < s->m && col >= 0 && col < s->n, "pivot position out of range");
    const RevisedWs w = revised_ws(s->m, s->n, s->workspace);
    hipStream_t st = (hipStream_t)stream;
    launch_direction(s, w, DIR_FORCED, col, st);
    launch_prepare(s, w, PREP_FORCED, row, st);
    launch_update(s, w, true, st);
    LPX_LAUNCH_CHECK();
    return LPX_OK;
}

extern "C" int lpx_revised_reinvert(const lpx_revised *s, void *stream)
{
    int rc = check_revised(s);
    if (rc) return rc;
    rc = launch_reinvert(s, (hipStream_t)stream);
    if (rc) return rc;
    LPX_LAUNCH_CHECK();
    return LPX_OK;
}

extern "C" int lpx_naive_pivot(const lpx_revised *s, int64_t row, int64_t col, void *stream)
{
    int rc = check_revised(s);
    if (rc) return rc;
    LPX_REQUIRE(row >= 0 && row < s->m && col >= 0 && col < s->n, "pivot position out of range");
    const RevisedWs w = revised_ws(s->m, s->n, s->workspace);
    hipStream_t st = (hipStream_t)stream;
    launch_direction(s, w, DIR_FORCED, col, st);
    launch_prepare(s, w, PREP_FORCED, row, st, 1);
    rc = launch_reinvert(s, st);
    if (rc) return rc;
    LPX_LAUNCH_CHECK();
    return LPX_OK;
}

extern "C" int lpx_bounded_update_bfs(const lpx_revised *s, void *stream)
{
    int rc = check_revised(s);
    if (rc) return rc;
    LPX_REQUIRE(s->lb && s->ub, "bounded solver needs lb/ub");
    const RevisedWs w = revised_ws(s->m, s->n, s->workspace);
    launch_bounded_bfs(s, w, (hipStream_t)stream);
    LPX_LAUNCH_CHECK();
    return LPX_OK;
}

extern "C" int lpx_revised_reduced_costs(const lpx_revised *s, int bounded, double *rc_out, void *stream)
{
    int rc = check_revised(s);
    if (rc) return rc;
    LPX_REQUIRE(rc_out != nullptr, "null output");
    const RevisedWs w = revised_ws(s->m, s->n, s->workspace);
    hipStream_t st = (hipStream_t)stream;
    launch_pricing_sums(s, w, st);
    launch_fin(s, w, bounded ? FIN_BOUNDED : FIN_PRIMAL, rc_out, st);
    LPX_LAUNCH_CHECK();
    return LPX_OK;
}

// ---------------------------------------------------------------------------------------------------------
// Problem construction on the device (preprocessing.py:6-69, special_solvers.py:48-52): the standard-form matrix of
//   min c x  s.t.  A x = b (me rows),  G x <= h (mi rows),  bounds on nb variables
// assembled directly in HBM instead of by vstack / hstack chains on the host:
//        nx cols   mi cols   nb cols   na cols
//      [   A         0         0        I    ]   me rows
//      [   G         I         0        I    ]   mi rows   canonical_form_to_standard_form, :17-28
//      [   E         0       diag(s)    I    ]   nb rows   add_variables_bounds_to_coefficient_matrix, :30-69:
//                                                          x_i - s = lb_i (s = -1) or x_i + s = ub_i (s = +1), one row and
//                                                          one column per finite bound, lower bounds first
// with every row whose right-hand side is negative negated afterwards (preprocess_problem, :6-15); the last block
// (na = rows or 0) is the artificial identity of the phase-one problem, appended AFTER the sign normalisation as the
// reference does (special_solvers.py:48-52), so it stays +I.  One thread per element of the output.
// ---------------------------------------------------------------------------------------------------------
struct StdFormArgs {
    int64_t me, mi, nx, nb, na, ld;
    const float *A, *b, *G, *h;
    const int64_t *bound_var;
    const float *bound_sign, *bound_val;
    float *out, *out_b;
};
__global__ __launch_bounds__(LPX_BLOCK) void k_build_standard_form(StdFormArgs a)
{
    const int64_t rows = a.me + a.mi + a.nb, core = a.nx + a.mi + a.nb, cols = core + a.na;
    const int64_t idx = (int64_t)blockIdx.x * LPX_BLOCK + threadIdx.x;
    if (idx >= rows * cols) return;
    const int64_t i = idx / cols, j = idx - i * cols;
    float rhs, v = 0.f;
    if (i < a.me) {
        rhs = a.b[i];
        if (j < a.nx) v = a.A[i * a.nx + j];
    } else if (i < a.me + a.mi) {
        const int64_t g = i - a.me;
        rhs = a.h[g];
        if (j < a.nx) v = a.G[g * a.nx + j];
        else if (j - a.nx == g) v = 1.f;
    } else {
        const int64_t t = i - a.me - a.mi;
        rhs = a.bound_val[t];
        if (j < a.nx) v = (j == a.bound_var[t]) ? 1.f : 0.f;
        else if (j - a.nx - a.mi == t) v = a.bound_sign[t];
    }
    const bool flip = rhs < 0.f; // preprocessing.py:11-13
    if (j < core) v = flip ? -v : v;
    else v = (j - core == i) ? 1.f : 0.f; // artificial block, added after the flips
    a.out[i * a.ld + j] = v;
    if (j == 0) a.out_b[i] = flip ? -rhs : rhs;
}
extern "C" int lpx_build_standard_form(int64_t me, int64_t mi, int64_t nx, int64_t nb, int64_t na, const float *A, const float *b,
                                       const float *G, const float *h, const int64_t *bound_var, const float *bound_sign,
                                       const float *bound_val, float *out, int64_t ld, float *out_b, void *stream)
{
    const int64_t rows = me + mi + nb, cols = nx + mi + nb + na;
    LPX_REQUIRE(me >= 0 && mi >= 0 && nb >= 0 && nx > 0 && rows > 0, "bad sizes");
    LPX_REQUIRE(na == 0 || na == rows, "the artificial block has one column per row or none");
    LPX_REQUIRE((me == 0 || (A && b)) && (mi == 0 || (G && h)) && (nb == 0 || (bound_var && bound_sign && bound_val)),
                "null input for a non-empty block");
    LPX_REQUIRE(out && out_b && ld >= cols, "null output / ld < columns");
    StdFormArgs a{me, mi, nx, nb, na, ld, A, b, G, h, bound_var, bound_sign, bound_val, out, out_b};
    hipLaunchKernelGGL(k_build_standard_form, dim3((unsigned)lpx_cdiv(rows * cols, LPX_BLOCK)), dim3(LPX_BLOCK), 0,
                       (hipStream_t)stream, a);
    LPX_LAUNCH_CHECK();
    return LPX_OK;
}

// One row of the tableau B^-1 A (special_solvers.py:69-71: PhaseOneSimplexSolver looks for a positive entry in the row of
// a basic artificial variable to pivot it out; the reference forms the whole m x n product for it).  A thread per column,
// the weights B^-1[row, :] staged in LDS in chunks, fma in row order.  Independent of the loop's status word.
#define TROW_CHUNK 1024
__global__ __launch_bounds__(LPX_BLOCK) void k_tableau_row(lpx_revised s, int64_t row, double *out)
{
    __shared__ double w[TROW_CHUNK];
    const int64_t j = (int64_t)blockIdx.x * LPX_BLOCK + threadIdx.x;
    double acc = 0.0;
    for (int64_t i0 = 0; i0 < s.m; i0 += TROW_CHUNK) {
        const int64_t cnt = (s.m - i0 < TROW_CHUNK) ? s.m - i0 : TROW_CHUNK;
        __syncthreads();
        for (int64_t i = threadIdx.x; i < cnt; i += LPX_BLOCK) w[i] = s.Binv[row * s.ldB + i0 + i];
        __syncthreads();
        if (j < s.n)
            for (int64_t i = 0; i < cnt; ++i) acc = fma(w[i], (double)s.A[(i0 + i) * s.ldA + j], acc);
    }
    if (j < s.n) out[j] = acc;
}
extern "C" int lpx_revised_tableau_row(const lpx_revised *s, int64_t row, double *out, void *stream)
{
    int rc = check_revised(s);
    if (rc) return rc;
    LPX_REQUIRE(out != nullptr && row >= 0 && row < s->m, "bad row / null output");
    hipLaunchKernelGGL(k_tableau_row, dim3((unsigned)lpx_cdiv(s->n, LPX_BLOCK)), dim3(LPX_BLOCK), 0, (hipStream_t)stream, *s,
                       row, out);
    LPX_LAUNCH_CHECK();
    return LPX_OK;
}

extern "C" int lpx_revised_expand(const lpx_revised *s, double *x, double *cost, void *stream)
{
    int rc = check_revised(s);
    if (rc) return rc;
    LPX_REQUIRE(x != nullptr && cost != nullptr, "null output");
    hipStream_t st = (hipStream_t)stream;
    hipLaunchKernelGGL(k_expand_fill, dim3((unsigned)lpx_cdiv(s->n, LPX_BLOCK)), dim3(LPX_BLOCK), 0, st, *s, x);
    hipLaunchKernelGGL(k_expand_scatter, dim3((unsigned)lpx_cdiv(s->m, LPX_BLOCK)), dim3(LPX_BLOCK), 0, st,
                       (const double *)s->xB, (const int64_t *)s->basis, s->m, x);
    hipLaunchKernelGGL(k_cost, dim3(1), dim3(1024), 0, st, s->c, (const double *)x, s->n, cost);
    LPX_LAUNCH_CHECK();
    return LPX_OK;
}

// ---------------------------------------------------------------------------------------------------------
// primal-dual restricted-problem updates  special_solvers.py:405-444
// The restricted primal is the full matrix [A | I] (n = n_orig + m columns) with inadmissible columns
// EXCLUDED; the artificial block stays in.  v = y A and u = y_rp A reuse the column-sum kernels.
// ---------------------------------------------------------------------------------------------------------
struct PdArgs {
    int64_t n_orig, nch, ldp, m;
    const double *part;
    const double *c64;
    double *v;
    int32_t *var_state;
    const double *vprev;
    const double *yrp;
    Partials P;
};
__global__ __launch_bounds__(LPX_BLOCK) void k_pd_admissible(PdArgs a)
{
    const int64_t col = (int64_t)blockIdx.x * LPX_BLOCK + threadIdx.x;
    if (col >= a.n_orig) return;
    const double v = fin_sum<double>(a.part, a.nch, a.ldp, col);
    a.v[col] = v;
    const double c = a.c64[col];
    const bool adm = fabs(v - c) <= LPX_ATOL + LPX_RTOL * fabs(c); // np.isclose(y @ A, c)
    a.var_state[col] = adm ? LPX_VS_LOWER : LPX_VS_EXCLUDED;
}
__global__ __launch_bounds__(LPX_BLOCK) void k_pd_theta(PdArgs a)
{
    __shared__ BlockRed red;
    const int64_t col = (int64_t)blockIdx.x * LPX_BLOCK + threadIdx.x;
    double th = LPX_INF, um = -LPX_INF;
    if (col < a.n_orig) {
        const double u = fin_sum<double>(a.part, a.nch, a.ldp, col);
        const int32_t vs = a.var_state[col];
        if (vs == LPX_VS_EXCLUDED) th = lpx_psd(a.c64[col] - a.vprev[col], u, false);
        else um = u;
    }
    if (col < a.m) um = fmax(um, a.yrp[col]); // artificial block of A_rp: y_rp @ I
    th = block_min(th, red);
    um = block_max(um, red);
    if (threadIdx.x == 0) {
        a.P.v0[blockIdx.x] = th;
        a.P.v1[blockIdx.x] = um;
    }
}
__global__ __launch_bounds__(LPX_BLOCK) void k_pd_axpy(Partials P, int64_t np, const double *yrp, double *y, int64_t m,
                                                      double *out, const double *gate)
{
    __shared__ BlockRed red;
    const double th = combine_min(P.v0, np, red);
    const double um = combine_max(P.v1, np, red);
    const double g = gate ? *gate : 1.0; // cost of the restricted primal: the dual step exists only while it is > 0 (:427)
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        out[0] = th;
        out[1] = um;
        if (gate) out[2] = g;
    }
    if (!(g > 0.0)) return; // complementary slackness attained (:445-447): y stays
    if (um <= 0.0) return; // DualIsUnboundedError, special_solvers.py:435-436
    const int64_t i = (int64_t)blockIdx.x * LPX_BLOCK + threadIdx.x;
    if (i < m) y[i] = y[i] + th * yrp[i]; // :444  (numpy: theta * y_rp, then +=; two roundings)
}
// forces status RUNNING-independent execution of the column-sum kernels: they gate on ctrl->status, which is
// OPTIMAL after the inner solve, so the primal-dual steps run them with a scratch control block.
__global__ void k_ctrl_scratch(lpx_ctrl *c)
{
    c->status = LPX_RUNNING;
    c->f32_phase = 0;
    c->leave_r = 0;
}

extern "C" int lpx_pd_admissible(const lpx_revised *s, const double *y, const double *c64, int64_t n_orig, double *v,
                                 void *stream)
{
    int rc = check_revised(s);
    if (rc) return rc;
    LPX_REQUIRE(y && c64 && v && n_orig > 0 && n_orig <= s->n, "bad arguments");
    const RevisedWs w = revised_ws(s->m, s->n, s->workspace);
    hipStream_t st = (hipStream_t)stream;
    lpx_ctrl *scr = w.scr_ctrl;
    hipLaunchKernelGGL(k_ctrl_scratch, dim3(1), dim3(1), 0, st, scr);
    ColsumArgs a{};
    a.M = s->A; a.ld = s->ldA; a.rows = s->m; a.cols = n_orig; a.rpc = w.pa.rpc;
    a.wmode = W_VEC; a.wvec = y; a.part = w.zpart; a.ldp = w.ldzp; a.ctrl = scr;
    hipLaunchKernelGGL((k_colsum<float, 4>), dim3((unsigned)w.pa.nstrips, (unsigned)w.pa.nch), dim3(w.pa.threads), 0, st, a);
    PdArgs p{};
    p.n_orig = n_orig; p.nch = w.pa.nch; p.ldp = w.ldzp; p.m = s->m; p.part = w.zpart; p.c64 = c64; p.v = v;
    p.var_state = s->var_state;
    hipLaunchKernelGGL(k_pd_admissible, dim3((unsigned)lpx_cdiv(n_orig, LPX_BLOCK)), dim3(LPX_BLOCK), 0, st, p);
    LPX_LAUNCH_CHECK();
    return LPX_OK;
}

extern "C" int lpx_pd_dual_step(const lpx_revised *s, double *y, const double *c64, const double *v, int64_t n_orig,
                                double *out, const double *cost_gate, void *stream)
{
    int rc = check_revised(s);
    if (rc) return rc;
    LPX_REQUIRE(y && c64 && v && out && n_orig > 0 && n_orig <= s->n, "bad arguments");
    const RevisedWs w = revised_ws(s->m, s->n, s->workspace);
    hipStream_t st = (hipStream_t)stream;
    lpx_ctrl *scr = w.scr_ctrl;
    hipLaunchKernelGGL(k_ctrl_scratch, dim3(1), dim3(1), 0, st, scr);
    // y_rp = c_B B^-1 of the solved restricted primal (float64 inverse; the reference re-inverts in float32)
    ColsumArgs b{};
    b.M = s->Binv; b.ld = s->ldB; b.rows = s->m; b.cols = s->m; b.rpc = w.py.rpc;
    b.wmode = W_CB; b.c = s->c; b.basis = s->basis; b.part = w.ypart; b.ldp = w.ldyp; b.ctrl = scr;
    hipLaunchKernelGGL((k_colsum<double, 2>), dim3((unsigned)w.py.nstrips, (unsigned)w.py.nch), dim3(w.py.threads), 0, st, b);
    FinArgs f{};
    f.mode = FIN_Y; f.ctrl = scr; f.cols = s->m; f.nch = w.py.nch; f.ldp = w.ldyp; f.part = w.ypart; f.out = w.y;
    hipLaunchKernelGGL(k_finalize, dim3((unsigned)lpx_cdiv(s->m, LPX_BLOCK)), dim3(LPX_BLOCK), 0, st, f);
    ColsumArgs a{};
    a.M = s->A; a.ld = s->ldA; a.rows = s->m; a.cols = n_orig; a.rpc = w.pa.rpc;
    a.wmode = W_VEC; a.wvec = w.y; a.part = w.zpart; a.ldp = w.ldzp; a.ctrl = scr;
    hipLaunchKernelGGL((k_colsum<float, 4>), dim3((unsigned)w.pa.nstrips, (unsigned)w.pa.nch), dim3(w.pa.threads), 0, st, a);
    PdArgs p{};
    p.n_orig = n_orig; p.nch = w.pa.nch; p.ldp = w.ldzp; p.m = s->m; p.part = w.zpart; p.c64 = c64;
    p.var_state = s->var_state; p.vprev = v; p.yrp = w.y; p.P = w.price;
    const int64_t nb = lpx_cdiv(n_orig > s->m ? n_orig : s->m, LPX_BLOCK);
    hipLaunchKernelGGL(k_pd_theta, dim3((unsigned)nb), dim3(LPX_BLOCK), 0, st, p);
    hipLaunchKernelGGL(k_pd_axpy, dim3((unsigned)lpx_cdiv(s->m, LPX_BLOCK)), dim3(LPX_BLOCK), 0, st, w.price, nb,
                       (const double *)w.y, y, s->m, out, cost_gate);
    LPX_LAUNCH_CHECK();
    return LPX_OK;
}

// ---------------------------------------------------------------------------------------------------------
// individually exported kernels (unit tests against the oracle; bench of K1)
// ---------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(LPX_BLOCK) void k_copy_row(const double *src, double *dst, int64_t n)
{
    const int64_t i = (int64_t)blockIdx.x * LPX_BLOCK + threadIdx.x;
    if (i < n) dst[i] = src[i];
}

extern "C" int lpx_pivot_update(double *M, int64_t rows, int64_t cols, int64_t ld, int64_t r, const double *e,
                                double *rowr_scratch, void *stream)
{
    LPX_REQUIRE(M && e && rowr_scratch, "null pointer");
    LPX_REQUIRE(rows > 0 && cols > 0 && ld >= cols && (ld % 2) == 0, "ld must be >= cols and even");
    LPX_REQUIRE(r >= 0 && r < rows, "pivot row out of range");
    LPX_REQUIRE(lpx_aligned16(M) && lpx_aligned16(rowr_scratch), "M and rowr_scratch must be 16-byte aligned");
    hipStream_t st = (hipStream_t)stream;
    hipLaunchKernelGGL(k_copy_row, dim3((unsigned)lpx_cdiv(ld, LPX_BLOCK)), dim3(LPX_BLOCK), 0, st,
                       (const double *)(M + r * ld), rowr_scratch, ld);
    Rank1Args a{};
    a.M = M; a.rows = rows; a.cols = cols; a.ld = ld; a.e = e; a.rowr = rowr_scratch; a.r = r; a.ctrl = nullptr;
    const Rank1Plan p = lpx_rank1_plan(rows, cols);
    lpx_k1_begin(st);
    lpx_rank1_launch<false>(p, a, st);
    lpx_k1_end(st);
    LPX_LAUNCH_CHECK();
    return LPX_OK;
}

// The same update for callers that cannot allocate (no counterpart in the reference): the copy of the old pivot row, which
// lpx_pivot_update keeps in the caller's rowr_scratch, lives in a file-scope device array.  Each GPU holds its own copy of
// the array at its own address; the host-side name `scratch` is only a handle for hipGetSymbolAddress, so the device
// address is looked up once per device (with that device current) and only that pointer is ever handed to a kernel.
// Every call on a GPU shares the array: calls that overlap on one GPU (two streams, two threads) race on it.
// LPX_STATIC_ROW_MAX (lpx.h) doubles: 512 KiB per GPU, held until the process ends
#define LPX_STATIC_MAX_DEVICES 64
static __device__ __align__(256) double scratch[LPX_STATIC_ROW_MAX];
static std::mutex g_static_mu;
static double *g_static_row[LPX_STATIC_MAX_DEVICES];

// The array exists once per GPU, each copy at its own address: the copy that belongs to a call is the one on the device
// that owns M (hipPointerGetAttributes), not whatever device happens to be current.  The first call per device looks the
// address up (hipGetSymbolAddress, with that device current) and is therefore NOT capture-safe: make one call per device
// outside stream capture first (lpx.h).
static int static_row_scratch(const void *M, double **out)
{
    int cur = -1;
    LPX_HIP(hipGetDevice(&cur));
    int dev = cur;
    hipPointerAttribute_t attr;
    if (hipPointerGetAttributes(&attr, M) == hipSuccess && attr.type == hipMemoryTypeDevice) dev = attr.device;
    else (void)hipGetLastError();
    LPX_REQUIRE(dev >= 0 && dev < LPX_STATIC_MAX_DEVICES, "device index beyond the per-device table");
    std::lock_guard<std::mutex> lock(g_static_mu);
    if (g_static_row[dev] == nullptr) {
        void *p = nullptr;
        if (dev != cur) LPX_HIP(hipSetDevice(dev));
        const hipError_t e = hipGetSymbolAddress(&p, HIP_SYMBOL(scratch));
        if (dev != cur) (void)hipSetDevice(cur);
        if (e != hipSuccess) {
            lpx_set_error("hipGetSymbolAddress(scratch) failed on device %d: %s", dev, hipGetErrorString(e));
            return LPX_E_HIP;
        }
        LPX_REQUIRE(p != nullptr && lpx_aligned16(p), "static scratch has no usable device address");
        g_static_row[dev] = (double *)p;
    }
    *out = g_static_row[dev];
    return LPX_OK;
}

extern "C" int lpx_pivot_update_static(double *M, int64_t rows, int64_t cols, int64_t ld, int64_t r, const double *e,
                                       void *stream)
{
    LPX_REQUIRE(M && e, "null pointer");
    LPX_REQUIRE(rows > 0 && cols > 0 && ld >= cols && (ld % 2) == 0, "ld must be >= cols and even");
    LPX_REQUIRE(r >= 0 && r < rows, "pivot row out of range");
    LPX_REQUIRE(lpx_aligned16(M), "M must be 16-byte aligned");
    if (ld > LPX_STATIC_ROW_MAX) {
        lpx_set_error("lpx_pivot_update_static: ld = %lld exceeds the static pivot-row buffer of %d doubles; use "
                      "lpx_pivot_update with a caller-owned rowr_scratch",
                      (long long)ld, LPX_STATIC_ROW_MAX);
        return LPX_E_SIZE;
    }
    double *rowr = nullptr;
    int rc = static_row_scratch(M, &rowr);
    if (rc) return rc;
    hipStream_t st = (hipStream_t)stream;
    // the whole row of the last call's contents that this call reads is overwritten first: nothing to clear
    hipLaunchKernelGGL(k_copy_row, dim3((unsigned)lpx_cdiv(ld, LPX_BLOCK)), dim3(LPX_BLOCK), 0, st,
                       (const double *)(M + r * ld), rowr, ld);
    Rank1Args a{};
    a.M = M; a.rows = rows; a.cols = cols; a.ld = ld; a.e = e; a.rowr = rowr; a.r = r; a.ctrl = nullptr;
    const Rank1Plan p = lpx_rank1_plan(rows, cols);
    lpx_rank1_launch<false>(p, a, st);
    LPX_LAUNCH_CHECK();
    return LPX_OK;
}

__global__ __launch_bounds__(LPX_BLOCK) void k_eta(const double *d, int64_t m, int64_t r, double *e)
{
    const int64_t i = (int64_t)blockIdx.x * LPX_BLOCK + threadIdx.x;
    if (i >= m) return;
    const double dr = d[r];
    e[i] = (i == r) ? 1.0 / dr : (-d[i]) / dr;
}
extern "C" int lpx_eta(const double *d, int64_t m, int64_t r, double *e, void *stream)
{
    LPX_REQUIRE(d && e && m > 0 && r >= 0 && r < m, "bad arguments");
    hipLaunchKernelGGL(k_eta, dim3((unsigned)lpx_cdiv(m, LPX_BLOCK)), dim3(LPX_BLOCK), 0, (hipStream_t)stream, d, m, r, e);
    LPX_LAUNCH_CHECK();
    return LPX_OK;
}

// Generic two-stage reductions over a vector: stage 1 = per-workgroup partials, stage 2 = one workgroup.
enum { RED_FIRST_NEG = 0, RED_ARGMAX = 1, RED_RATIO_PRIMAL = 2, RED_RATIO_DUAL = 3, RED_RATIO_BOUNDED = 4 };
struct RedArgs {
    int op;
    int64_t n;
    const double *a, *b, *c, *d;
    const int32_t *var_state;
    int f32;
    double sign;
    Partials P;
    double *out_val;
    int64_t *out_idx;
};
__global__ __launch_bounds__(LPX_BLOCK) void k_red_stage1(RedArgs a)
{
    __shared__ BlockRed red;
    const int64_t k = (int64_t)blockIdx.x * LPX_BLOCK + threadIdx.x;
    const bool in = k < a.n && !(a.var_state && a.var_state[k] == LPX_VS_EXCLUDED);
    const bool f32 = a.f32 != 0;
    if (a.op == RED_FIRST_NEG) {
        const double v = in ? a.a[k] : LPX_INF;
        const double mn = block_min(v, red);
        const int64_t j = block_imin((in && v < 0.0) ? k : LPX_NOIDX, red);
        if (threadIdx.x == 0) { a.P.v0[blockIdx.x] = mn; a.P.i0[blockIdx.x] = j; }
    } else if (a.op == RED_ARGMAX) {
        VI t{in ? -a.a[k] : LPX_INF, in ? k : LPX_NOIDX};
        t = block_argmin(t, red);
        if (threadIdx.x == 0) { a.P.v0[blockIdx.x] = -t.v; a.P.i0[blockIdx.x] = t.i; }
    } else if (a.op == RED_RATIO_PRIMAL) {
        VI t{in ? lpx_psd(a.a[k], a.b[k], f32) : LPX_INF, in ? k : LPX_NOIDX};
        t = block_argmin(t, red);
        const double dm = block_max(in ? a.b[k] : -LPX_INF, red);
        if (threadIdx.x == 0) { a.P.v0[blockIdx.x] = t.v; a.P.i0[blockIdx.x] = t.i; a.P.v1[blockIdx.x] = dm; }
    } else if (a.op == RED_RATIO_DUAL) {
        VI t{in ? lpx_dsd(a.a[k], a.b[k], f32) : LPX_INF, in ? k : LPX_NOIDX};
        t = block_argmin(t, red);
        const double rm = block_min(in ? a.b[k] : LPX_INF, red);
        if (threadIdx.x == 0) { a.P.v0[blockIdx.x] = t.v; a.P.i0[blockIdx.x] = t.i; a.P.v1[blockIdx.x] = rm; }
    } else {
        VI t1{LPX_INF, LPX_NOIDX}, t2{LPX_INF, LPX_NOIDX};
        if (in) {
            t1.v = lpx_psd(a.a[k] - a.c[k], a.sign * a.b[k], false);
            t2.v = lpx_psd(a.d[k] - a.a[k], -a.sign * a.b[k], false);
            t1.i = t2.i = k;
        }
        t1 = block_argmin(t1, red);
        t2 = block_argmin(t2, red);
        if (threadIdx.x == 0) {
            a.P.v0[blockIdx.x] = t1.v; a.P.i0[blockIdx.x] = t1.i;
            a.P.v1[blockIdx.x] = t2.v; a.P.i1[blockIdx.x] = t2.i;
        }
    }
}
__global__ __launch_bounds__(LPX_BLOCK) void k_red_stage2(RedArgs a, int64_t np)
{
    __shared__ BlockRed red;
    if (a.op == RED_FIRST_NEG) {
        const double mn = combine_min(a.P.v0, np, red);
        const int64_t j = combine_imin(a.P.i0, np, red);
        if (threadIdx.x == 0) { a.out_val[0] = mn; a.out_idx[0] = (j == LPX_NOIDX) ? -1 : j; }
    } else if (a.op == RED_ARGMAX) {
        VI b{LPX_INF, LPX_NOIDX};
        for (int64_t k = threadIdx.x; k < np; k += blockDim.x)
            if (vi_less(-a.P.v0[k], a.P.i0[k], b.v, b.i)) { b.v = -a.P.v0[k]; b.i = a.P.i0[k]; }
        b = block_argmin(b, red);
        if (threadIdx.x == 0) { a.out_val[0] = -b.v; a.out_idx[0] = b.i; }
    } else if (a.op == RED_RATIO_PRIMAL) {
        const VI t = combine_argmin(a.P.v0, a.P.i0, np, red);
        const double dm = combine_max(a.P.v1, np, red);
        if (threadIdx.x == 0) { a.out_val[0] = t.v; a.out_val[1] = dm; a.out_idx[0] = t.i; }
    } else if (a.op == RED_RATIO_DUAL) {
        const VI t = combine_argmin(a.P.v0, a.P.i0, np, red);
        const double rm = combine_min(a.P.v1, np, red);
        if (threadIdx.x == 0) { a.out_val[0] = t.v; a.out_val[1] = rm; a.out_idx[0] = t.i; }
    } else {
        const VI t1 = combine_argmin(a.P.v0, a.P.i0, np, red);
        const VI t2 = combine_argmin(a.P.v1, a.P.i1, np, red);
        if (threadIdx.x == 0) { a.out_val[0] = t1.v; a.out_val[1] = t2.v; a.out_idx[0] = t1.i; a.out_idx[1] = t2.i; }
    }
}
extern "C" int64_t lpx_reduce_scratch_bytes(void) { return 4 * lpx_roundup(8 * LPX_MAXP, 256); }
static int run_reduction(RedArgs a, void *scratch, void *stream)
{
    LPX_REQUIRE(scratch && a.out_val && a.out_idx && a.n > 0, "bad arguments");
    const int64_t np = lpx_cdiv(a.n, LPX_BLOCK);
    LPX_REQUIRE(np <= LPX_MAXP, "vector too long for the partial arrays");
    char *p = (char *)scratch;
    carve_partials(p, a.P);
    hipStream_t st = (hipStream_t)stream;
    hipLaunchKernelGGL(k_red_stage1, dim3((unsigned)np), dim3(LPX_BLOCK), 0, st, a);
    hipLaunchKernelGGL(k_red_stage2, dim3(1), dim3(LPX_BLOCK), 0, st, a, np);
    LPX_LAUNCH_CHECK();
    return LPX_OK;
}
extern "C" int lpx_price_first_negative(const double *rc, const int32_t *var_state, int64_t n, double *out_val,
                                        int64_t *out_idx, void *scratch, void *stream)
{
    RedArgs a{};
    a.op = RED_FIRST_NEG; a.n = n; a.a = rc; a.var_state = var_state; a.out_val = out_val; a.out_idx = out_idx;
    LPX_REQUIRE(rc != nullptr, "null input");
    return run_reduction(a, scratch, stream);
}
extern "C" int lpx_price_argmax(const double *rc, int64_t n, double *out_val, int64_t *out_idx, void *scratch, void *stream)
{
    RedArgs a{};
    a.op = RED_ARGMAX; a.n = n; a.a = rc; a.out_val = out_val; a.out_idx = out_idx;
    LPX_REQUIRE(rc != nullptr, "null input");
    return run_reduction(a, scratch, stream);
}
extern "C" int lpx_ratio_primal(const double *x, const double *d, int64_t m, int f32, double *out_val, int64_t *out_idx,
                                void *scratch, void *stream)
{
    RedArgs a{};
    a.op = RED_RATIO_PRIMAL; a.n = m; a.a = x; a.b = d; a.f32 = f32; a.out_val = out_val; a.out_idx = out_idx;
    LPX_REQUIRE(x && d, "null input");
    return run_reduction(a, scratch, stream);
}
extern "C" int lpx_ratio_dual(const double *rc, const double *row, const int32_t *var_state, int64_t n, int f32,
                              double *out_val, int64_t *out_idx, void *scratch, void *stream)
{
    RedArgs a{};
    a.op = RED_RATIO_DUAL; a.n = n; a.a = rc; a.b = row; a.var_state = var_state; a.f32 = f32;
    a.out_val = out_val; a.out_idx = out_idx;
    LPX_REQUIRE(rc && row, "null input");
    return run_reduction(a, scratch, stream);
}
extern "C" int lpx_ratio_bounded(const double *x, const double *d, const double *lbB, const double *ubB, int64_t m,
                                 double sign, double *out_val, int64_t *out_idx, void *scratch, void *stream)
{
    RedArgs a{};
    a.op = RED_RATIO_BOUNDED; a.n = m; a.a = x; a.b = d; a.c = lbB; a.d = ubB; a.sign = sign;
    a.out_val = out_val; a.out_idx = out_idx;
    LPX_REQUIRE(x && d && lbB && ubB, "null input");
    return run_reduction(a, scratch, stream);
}
extern "C" int lpx_dual_row_select(const double *xB, int64_t m, double *out_val, int64_t *out_idx, void *scratch,
                                   void *stream)
{
    return lpx_price_first_negative(xB, nullptr, m, out_val, out_idx, scratch, stream);
}
