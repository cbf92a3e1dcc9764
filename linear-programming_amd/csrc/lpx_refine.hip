// lpx_refine.hip -- opt-in numerical hygiene for the revised form: refine the product-form inverse in fp64
// (Newton-Schulz, first half of this file) or rebuild it from the basis columns (blocked Gauss-Jordan, second half).
//
// The reference never re-inverts in its revised solvers (primal_solvers.py:247-249 multiplies eta matrices for the
// whole solve); its only reinversion is the naive solvers' np.linalg.inv per pivot (primal_solvers.py:92-94).  On
// runs of 10^4..10^5 pivots (Bland's rule at m = 1000) the product form drifts.  Because the drift is
// small, a fresh factorisation is not needed: one Newton-Schulz step
//       R = I - B X ,   X <- X + X R            (B = A[:, basis], X = current B^-1)
// squares the residual (|R| 1e-9 -> 1e-18, i.e. down to fp64 rounding), needs no pivoting and is two GEMMs -- the one
// GEMM-shaped piece of work on this path, so it runs on the matrix cores: v_mfma_f64_16x16x4_f64, LDS-tiled.
// Never part of the parity path: callers ask for it (RevisedEngine.refine); the default solve() is the reference's.
//
// Roofline: MFMA-bound, 4 m^3 flop per step against the 78.6 TFLOP/s dense fp64 matrix peak of MI355X.
#include "lpx_device.hpp"

typedef double d4_t __attribute__((ext_vector_type(4)));

struct GemmArgs {
    // D = alpha * op(A) B + beta * C + gamma * I ; op(A) = A (M x K, row-major, lda) or, with TRANSA, A given as K x M
    const double *A, *B, *C;
    double *D;
    int64_t M, N, K, lda, ldb, ldc, ldd;
    double alpha, beta, gamma;
};

#define GEMM_KT 8 /* K-depth of one LDS tile (two buffers of A and B tiles stay under 64 KB) */

// One workgroup = 4 waves = (WT*32) x (WT*32) tile of D; one wave = (WT*16) x (WT*16): WT x WT MFMA tiles of 16x16.
// LDS tiles are k-major (As[k][row], Bs[k][col]) so that the MFMA operand of lane l -- row/col l%16 at k = l/16 -- is a
// conflict-free read of consecutive doubles.  The next K-tile is fetched into registers while the current one is
// multiplied (an fp64 MFMA occupies a SIMD for 64 cycles: a K-tile is >= 4 k cycles of matrix work per wave).
template <int WT, bool TRANSA>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(2, 2))) void k_dgemm(GemmArgs g)
{
    constexpr int TM = WT * 32, TN = WT * 32, LDT = TM + 4;
    constexpr int PER = TM * GEMM_KT / 256; // doubles per lane per operand tile: 4 (WT=4) or 2 (WT=2)
    __shared__ double As[2][GEMM_KT][LDT];
    __shared__ double Bs[2][GEMM_KT][LDT];
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const int64_t row0 = (int64_t)blockIdx.y * TM, col0 = (int64_t)blockIdx.x * TN;
    const int wm = (wave >> 1) * (WT * 16), wn = (wave & 1) * (WT * 16);

    d4_t acc[WT][WT];
#pragma unroll
    for (int i = 0; i < WT; ++i)
#pragma unroll
        for (int j = 0; j < WT; ++j) acc[i][j] = d4_t{0.0, 0.0, 0.0, 0.0};

    double ra[PER], rb[PER];
    // global -> registers for the K-tile starting at k0
    auto fetch = [&](int64_t k0) {
        if (TRANSA) { // A stored K x M: a K-tile row is contiguous in the row index
            const int k = t / (TM / PER), seg = (t % (TM / PER)) * PER;
#pragma unroll
            for (int u = 0; u < PER; ++u) {
                const int64_t kk = k0 + k, rr = row0 + seg + u;
                ra[u] = (kk < g.K && rr < g.M) ? g.A[kk * g.lda + rr] : 0.0;
            }
        } else { // A stored M x K: lane reads PER consecutive k of one row
            const int r = t / (GEMM_KT / PER), seg = (t % (GEMM_KT / PER)) * PER;
#pragma unroll
            for (int u = 0; u < PER; ++u) {
                const int64_t kk = k0 + seg + u, rr = row0 + r;
                ra[u] = (kk < g.K && rr < g.M) ? g.A[rr * g.lda + kk] : 0.0;
            }
        }
        {
            const int k = t / (TN / PER), seg = (t % (TN / PER)) * PER;
#pragma unroll
            for (int u = 0; u < PER; ++u) {
                const int64_t kk = k0 + k, cc = col0 + seg + u;
                rb[u] = (kk < g.K && cc < g.N) ? g.B[kk * g.ldb + cc] : 0.0;
            }
        }
    };
    auto stage = [&](int buf) {
        if (TRANSA) {
            const int k = t / (TM / PER), seg = (t % (TM / PER)) * PER;
#pragma unroll
            for (int u = 0; u < PER; ++u) As[buf][k][seg + u] = ra[u];
        } else {
            const int r = t / (GEMM_KT / PER), seg = (t % (GEMM_KT / PER)) * PER;
#pragma unroll
            for (int u = 0; u < PER; ++u) As[buf][seg + u][r] = ra[u];
        }
        const int k = t / (TN / PER), seg = (t % (TN / PER)) * PER;
#pragma unroll
        for (int u = 0; u < PER; ++u) Bs[buf][k][seg + u] = rb[u];
    };

    const int64_t ntile = (g.K + GEMM_KT - 1) / GEMM_KT;
    fetch(0);
    stage(0);
    __syncthreads();
    for (int64_t kt = 0; kt < ntile; ++kt) {
        const int buf = (int)(kt & 1);
        if (kt + 1 < ntile) fetch((kt + 1) * GEMM_KT);
        // operands of step k4+1 are read from LDS while the 16 MFMAs of step k4 run
        double a[2][WT], b[2][WT];
        auto operands = [&](int k4, int slot) {
            const int kk = k4 * 4 + (lane >> 4);
#pragma unroll
            for (int i = 0; i < WT; ++i) a[slot][i] = As[buf][kk][wm + i * 16 + (lane & 15)];
#pragma unroll
            for (int j = 0; j < WT; ++j) b[slot][j] = Bs[buf][kk][wn + j * 16 + (lane & 15)];
        };
        operands(0, 0);
#pragma unroll
        for (int k4 = 0; k4 < GEMM_KT / 4; ++k4) {
            if (k4 + 1 < GEMM_KT / 4) operands(k4 + 1, (k4 + 1) & 1);
#pragma unroll
            for (int i = 0; i < WT; ++i)
#pragma unroll
                for (int j = 0; j < WT; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[k4 & 1][i], b[k4 & 1][j], acc[i][j], 0, 0, 0);
        }
        if (kt + 1 < ntile) stage(buf ^ 1); // the other buffer: last read one barrier ago
        __syncthreads();
    }

    // D tile element of lane l, register q of an MFMA tile: row l/16 + 4*q, column l%16 (measured: tools/dbg_gemm.py)
    // C is read in batches of 4*WT values per lane ahead of their use: one exposed round trip per batch instead of one
    // per element (at K = 64, the reinversion's trailing update, the epilogue is most of the kernel)
#pragma unroll
    for (int i = 0; i < WT; ++i) {
        double cv[WT][4];
#pragma unroll
        for (int j = 0; j < WT; ++j)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int64_t r = row0 + wm + i * 16 + (lane >> 4) + 4 * q, c = col0 + wn + j * 16 + (lane & 15);
                cv[j][q] = (g.C && r < g.M && c < g.N) ? g.C[r * g.ldc + c] : 0.0;
            }
#pragma unroll
        for (int j = 0; j < WT; ++j)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int64_t r = row0 + wm + i * 16 + (lane >> 4) + 4 * q, c = col0 + wn + j * 16 + (lane & 15);
                if (r < g.M && c < g.N) {
                    double v = g.alpha * acc[i][j][q];
                    if (g.C) v = fma(g.beta, cv[j][q], v);
                    if (r == c) v += g.gamma;
                    g.D[r * g.ldd + c] = v;
                }
            }
    }
}

static int gemm_check(const GemmArgs &g)
{
    LPX_REQUIRE(g.A && g.B && g.D, "null matrix");
    LPX_REQUIRE(g.M > 0 && g.N > 0 && g.K > 0, "empty GEMM");
    LPX_REQUIRE(g.ldb >= g.N && g.ldd >= g.N && (!g.C || g.ldc >= g.N), "leading dimension too small");
    return LPX_OK;
}
template <bool TRANSA>
static void gemm_launch(const GemmArgs &g, hipStream_t st)
{
    // big tiles once they fill the chip; small problems get 4x as many workgroups
    if (lpx_cdiv(g.M, 128) * lpx_cdiv(g.N, 128) >= 512)
        hipLaunchKernelGGL((k_dgemm<4, TRANSA>), dim3((unsigned)lpx_cdiv(g.N, 128), (unsigned)lpx_cdiv(g.M, 128)), dim3(256), 0, st, g);
    else
        hipLaunchKernelGGL((k_dgemm<2, TRANSA>), dim3((unsigned)lpx_cdiv(g.N, 64), (unsigned)lpx_cdiv(g.M, 64)), dim3(256), 0, st, g);
}

// D = alpha * op(A) B + beta * C + gamma * I  (fp64, row-major; op(A) = A^T when transa, A then being K x M)
extern "C" int lpx_dgemm(int transa, int64_t M, int64_t N, int64_t K, double alpha, const double *A, int64_t lda,
                         const double *B, int64_t ldb, double beta, const double *C, int64_t ldc, double gamma, double *D,
                         int64_t ldd, void *stream)
{
    GemmArgs g{A, B, C, D, M, N, K, lda, ldb, ldc, ldd, alpha, beta, gamma};
    int rc = gemm_check(g);
    if (rc) return rc;
    LPX_REQUIRE(lda >= (transa ? M : K), "lda too small");
    if (transa) gemm_launch<true>(g, (hipStream_t)stream);
    else gemm_launch<false>(g, (hipStream_t)stream);
    LPX_LAUNCH_CHECK();
    return LPX_OK;
}

// ---------------------------------------------------------------------------------------------------------
// refinement driver
// ---------------------------------------------------------------------------------------------------------
// BT[k][i] = A[i][basis[k]] as fp64 (rows of the transposed copy At are contiguous)
__global__ __launch_bounds__(LPX_BLOCK) void k_gather_basis_T(lpx_revised s, double *BT, int64_t ld)
{
    const int64_t k = blockIdx.y;
    const int64_t i = (int64_t)blockIdx.x * LPX_BLOCK + threadIdx.x;
    if (i < s.m) BT[k * ld + i] = (double)s.At[s.basis[k] * s.ldAt + i];
}
__global__ __launch_bounds__(LPX_BLOCK) void k_absmax(const double *M, int64_t rows, int64_t cols, int64_t ld, double *out)
{
    __shared__ BlockRed red;
    double mx = 0.0;
    for (int64_t idx = (int64_t)blockIdx.x * LPX_BLOCK + threadIdx.x; idx < rows * cols; idx += (int64_t)gridDim.x * LPX_BLOCK) {
        const int64_t r = idx / cols, c = idx - r * cols;
        mx = fmax(mx, fabs(M[r * ld + c]));
    }
    mx = block_max(mx, red);
    // non-negative doubles order like their bit patterns: one integer atomic max per workgroup
    if (threadIdx.x == 0) atomicMax((unsigned long long *)out, (unsigned long long)__double_as_longlong(mx));
}
__global__ void k_refine_begin(double *resid, lpx_ctrl *ctrl)
{
    resid[0] = 0.0;
    resid[1] = 0.0;
    ctrl->f32_phase = 0; // the refined inverse is a float64 object (the phase ends with it)
}
__global__ __launch_bounds__(LPX_BLOCK) void k_copy2d(const double *src, int64_t lds_, double *dst, int64_t ldd, int64_t rows,
                                                     int64_t cols)
{
    const int64_t r = blockIdx.y;
    const int64_t c = (int64_t)blockIdx.x * LPX_BLOCK + threadIdx.x;
    if (c < cols) dst[r * ldd + c] = src[r * lds_ + c];
}
// x_B = B^-1 b in fp64 (primal_solvers.py:38 recomputed with the refined inverse)
__global__ __launch_bounds__(LPX_BLOCK) void k_refine_xb(lpx_revised s)
{
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int64_t i = (int64_t)blockIdx.x * LPX_WAVES + wave;
    if (i >= s.m) return;
    double acc = 0.0;
    for (int64_t k = lane; k < s.m; k += 64) acc = fma(s.Binv[i * s.ldB + k], (double)s.b[k], acc);
    acc = wave_sum<double>(acc);
    if (lane == 0) s.xB[i] = acc;
}

extern "C" int64_t lpx_refine_workspace_bytes(int64_t m)
{
    if (m <= 0) return 0;
    const int64_t ld = lpx_roundup(m, 2);
    return 3 * 8 * m * ld + 256;
}

// `steps` Newton-Schulz steps on s->Binv; resid (device, 2 doubles): max|I - B X| before the first step and after the
// last.  update_xB != 0: x_B = B^-1 b afterwards (primal / dual revised solvers; the bounded solver recomputes its own).
extern "C" int lpx_revised_refine(const lpx_revised *s, int64_t steps, int update_xB, void *work, int64_t work_bytes,
                                  double *resid, void *stream)
{
    LPX_REQUIRE(s && s->m > 0 && s->A && s->At && s->Binv && s->basis && s->ctrl && s->b && s->xB, "null state");
    LPX_REQUIRE(steps >= 1 && steps <= 8, "1..8 refinement steps");
    LPX_REQUIRE(work && work_bytes >= lpx_refine_workspace_bytes(s->m) && lpx_aligned16(work), "workspace too small");
    LPX_REQUIRE(resid != nullptr, "null residual buffer");
    const int64_t m = s->m, ld = lpx_roundup(m, 2);
    double *BT = (double *)work, *R = BT + m * ld, *Xn = R + m * ld;
    hipStream_t st = (hipStream_t)stream;
    const dim3 g2((unsigned)lpx_cdiv(m, LPX_BLOCK), (unsigned)m);
    hipLaunchKernelGGL(k_refine_begin, dim3(1), dim3(1), 0, st, resid, s->ctrl);
    hipLaunchKernelGGL(k_gather_basis_T, g2, dim3(LPX_BLOCK), 0, st, *s, BT, ld);
    const GemmArgs gr{BT, s->Binv, nullptr, R, m, m, m, ld, s->ldB, 0, ld, -1.0, 0.0, 1.0};   // R = I - B X
    const GemmArgs gx{s->Binv, R, s->Binv, Xn, m, m, m, s->ldB, ld, s->ldB, ld, 1.0, 1.0, 0.0}; // Xn = X + X R
    for (int64_t it = 0; it < steps; ++it) {
        gemm_launch<true>(gr, st);
        if (it == 0) hipLaunchKernelGGL(k_absmax, dim3(256), dim3(LPX_BLOCK), 0, st, (const double *)R, m, m, ld, resid);
        gemm_launch<false>(gx, st);
        hipLaunchKernelGGL(k_copy2d, g2, dim3(LPX_BLOCK), 0, st, (const double *)Xn, ld, s->Binv, s->ldB, m, m);
    }
    gemm_launch<true>(gr, st);
    hipLaunchKernelGGL(k_absmax, dim3(256), dim3(LPX_BLOCK), 0, st, (const double *)R, m, m, ld, resid + 1);
    if (update_xB) hipLaunchKernelGGL(k_refine_xb, dim3((unsigned)lpx_cdiv(m, LPX_WAVES)), dim3(LPX_BLOCK), 0, st, *s);
    LPX_LAUNCH_CHECK();
    return LPX_OK;
}

// ---------------------------------------------------------------------------------------------------------
// reinversion: B^-1 from scratch in fp64
// ---------------------------------------------------------------------------------------------------------
// The reference's only reinversion is np.linalg.inv(A[:, basis]) (primal_solvers.py:92-94, LAPACK getrf + getri).
// Here: in-place blocked Gauss-Jordan with partial pivoting, 64 columns per block.
//   1. pivot search: LU with partial pivoting on a scratch copy of the block's columns (rows >= k0), one launch per
//      column, ping-ponging between two copies so that no kernel reads what it writes; only the pivot rows are kept
//   2. the 64 row swaps are applied to the whole matrix (and to the row-identity vector `id`)
//   3. P = W[K, K] is inverted in LDS by one workgroup (no pivoting needed: its pivots are those of step 1)
//   4. V = W[:, K] with rows K zeroed; W[:, K] = unit columns; R = P^-1 W[K, :]; W[K, :] = R; W -= V R
//      -- the last is an m x m x 64 GEMM on the matrix cores and carries all 2 m^3 flop of the inversion
//   5. the row swaps come back as a column permutation: out[:, id[r]] = W[:, r]
#define INV_NB 64
#define INV_ROWS 64 /* panel rows per workgroup: 16 per wave, lane = panel column */

struct InvWs {
    double *W, *V, *R, *S[2], *Pinv, *pv[2], *anorm;
    int64_t *pi[2], *piv, *id, *inv;
    int64_t ldw;
};
static inline int64_t inv_ldw(int64_t m) { return lpx_roundup(m, 16); }
extern "C" int64_t lpx_invert_workspace_bytes(int64_t m)
{
    if (m <= 0) return 0;
    const int64_t np = lpx_cdiv(m, INV_ROWS);
    return 8 * (m * inv_ldw(m) + 3 * m * INV_NB + INV_NB * inv_ldw(m) + INV_NB * INV_NB + 4 * np + 3 * m + 2) + 512;
}
static InvWs inv_carve(int64_t m, void *work)
{
    InvWs w;
    const int64_t np = lpx_cdiv(m, INV_ROWS);
    w.ldw = inv_ldw(m);
    double *p = (double *)work;
    w.W = p; p += m * w.ldw;
    w.R = p; p += INV_NB * w.ldw;
    w.V = p; p += m * INV_NB;
    w.S[0] = p; p += m * INV_NB;
    w.S[1] = p; p += m * INV_NB;
    w.Pinv = p; p += INV_NB * INV_NB;
    w.pv[0] = p; p += np;
    w.pv[1] = p; p += np;
    w.anorm = p; p += 2;
    int64_t *q = (int64_t *)p;
    w.pi[0] = q; q += np;
    w.pi[1] = q; q += np;
    w.piv = q; q += m;
    w.id = q; q += m;
    w.inv = q;
    return w;
}

// partial of one workgroup for the next column's pivot search: (-|x|, row), so that "smallest value, then lowest
// index" is the largest magnitude with the first row on ties (LAPACK's idamax order)
__device__ __forceinline__ void inv_publish(double best, int64_t brow, int col_lane, double *pv, int64_t *pi)
{
    __shared__ double bv[LPX_WAVES];
    __shared__ int64_t bi[LPX_WAVES];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (lane == col_lane) { bv[wave] = best; bi[wave] = brow; }
    __syncthreads();
    if (threadIdx.x == 0) {
        double b = -1.0;
        int64_t r = LPX_NOIDX;
        for (int k = 0; k < LPX_WAVES; ++k) // waves hold ascending rows: strict > keeps the first
            if (bv[k] > b) { b = bv[k]; r = bi[k]; }
        pv[blockIdx.x] = r == LPX_NOIDX ? LPX_INF : -b;
        pi[blockIdx.x] = r;
    }
}

__global__ __launch_bounds__(LPX_BLOCK) void k_inv_init(int64_t *id, int64_t *info, double *anorm, int64_t m)
{
    const int64_t r = (int64_t)blockIdx.x * LPX_BLOCK + threadIdx.x;
    if (r < m) id[r] = r;
    if (r == 0) { info[0] = 0; anorm[0] = 0.0; }
}

__global__ __launch_bounds__(LPX_BLOCK) void k_inv_panel_load(const double *W, int64_t ldw, int64_t m, int64_t k0, int w,
                                                              double *S, double *pv, int64_t *pi)
{
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int64_t r0 = k0 + (int64_t)blockIdx.x * INV_ROWS + wave * (INV_ROWS / LPX_WAVES);
    double best = -1.0;
    int64_t brow = LPX_NOIDX;
    for (int t = 0; t < INV_ROWS / LPX_WAVES; ++t) {
        const int64_t r = r0 + t;
        if (r >= m) break;
        const double x = lane < w ? W[r * ldw + k0 + lane] : 0.0;
        S[r * INV_NB + lane] = x;
        const double a = fabs(x);
        if (a > best) { best = a; brow = r; }
    }
    inv_publish(best, brow, 0, pv, pi);
}

// column i of the block: pick its pivot row from the partials of the previous launch, eliminate it from the rows
// below (reading Sin, writing Sout; the row swap happens on the way) and leave the partials for column i+1
__global__ __launch_bounds__(LPX_BLOCK) void k_inv_panel_step(const double *Sin, double *Sout, int64_t m, int64_t k0, int w, int i,
                                                              const double *pvin, const int64_t *piin, int64_t np, double *pvout,
                                                              int64_t *piout, int64_t *piv, int64_t *info, const double *anorm)
{
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int64_t c = k0 + i;
    // the rows' loads do not depend on the pivot choice: issue them first, the partials' round trip runs beside them
    constexpr int RPW = INV_ROWS / LPX_WAVES;
    const int64_t r0 = k0 + (int64_t)blockIdx.x * INV_ROWS + wave * RPW;
    double src[RPW];
    if (i + 1 < w) {
#pragma unroll
        for (int t = 0; t < RPW; ++t) src[t] = (r0 + t < m && r0 + t > c) ? Sin[(r0 + t) * INV_NB + lane] : 0.0;
    }
    const double crow = Sin[c * INV_NB + lane];
    // every wave combines the partials for itself: no LDS round, no barrier on the way to the pivot row's address
    VI top{LPX_INF, LPX_NOIDX};
    for (int64_t k = lane; k < np; k += 64)
        if (vi_less(pvin[k], piin[k], top.v, top.i)) { top.v = pvin[k]; top.i = piin[k]; }
    top = wave_argmin(top);
    int64_t p = top.i;
    // no candidate above m * eps * max|A| (or only NaNs): singular to working precision.  LAPACK flags exact zeros only;
    // a simplex basis with a repeated column leaves rounding dust there, hence the threshold.
    const bool bad = !(-top.v > (double)m * 2.220446049250313e-16 * anorm[0]) || p == LPX_NOIDX;
    if (bad) p = c;
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        piv[c] = p;
        if (bad && info[0] == 0) info[0] = c + 1;
    }
    if (i + 1 >= w) return; // last column of the block: only its pivot row was wanted
    const double prow = Sin[p * INV_NB + lane];
    const double pivot = __shfl(prow, i);
    double best = -1.0;
    int64_t brow = LPX_NOIDX;
#pragma unroll
    for (int t = 0; t < RPW; ++t) {
        const int64_t r = r0 + t;
        if (r >= m || r <= c) continue;
        const double x = r == p ? crow : src[t];
        const double l = __shfl(x, i) / pivot;
        const double out = fma(-l, prow, x);
        Sout[r * INV_NB + lane] = out;
        const double a = fabs(out);
        if (a > best) { best = a; brow = r; }
    }
    inv_publish(best, brow, i + 1, pvout, piout);
}

// the block's row swaps, in order, on every column of W; thread m carries the row-identity vector
__global__ __launch_bounds__(LPX_BLOCK) void k_inv_swap(double *W, int64_t ldw, int64_t m, int64_t k0, int w, const int64_t *piv,
                                                        int64_t *id)
{
    const int64_t j = (int64_t)blockIdx.x * LPX_BLOCK + threadIdx.x;
    if (j > m) return;
    for (int i = 0; i < w; ++i) {
        const int64_t c = k0 + i, p = piv[c];
        if (p == c) continue;
        if (j < m) {
            const double a = W[c * ldw + j], b = W[p * ldw + j];
            W[c * ldw + j] = b;
            W[p * ldw + j] = a;
        } else {
            const int64_t a = id[c], b = id[p];
            id[c] = b;
            id[p] = a;
        }
    }
}

// P^-1 of the w x w diagonal block, Gauss-Jordan in LDS by one workgroup
__global__ __launch_bounds__(LPX_BLOCK) void k_inv_block(const double *W, int64_t ldw, int64_t k0, int w, double *Pinv)
{
    __shared__ double P[INV_NB][INV_NB + 1];
    __shared__ double colk[INV_NB], rowk[INV_NB];
    const int j = threadIdx.x & 63, i0 = threadIdx.x >> 6;
    for (int q = 0; q < INV_NB / LPX_WAVES; ++q) {
        const int i = i0 + LPX_WAVES * q;
        P[i][j] = (i < w && j < w) ? W[(k0 + i) * ldw + k0 + j] : (i == j ? 1.0 : 0.0);
    }
    __syncthreads();
    for (int k = 0; k < w; ++k) {
        if (threadIdx.x < INV_NB) {
            colk[threadIdx.x] = P[threadIdx.x][k];
            rowk[threadIdx.x] = (threadIdx.x == k ? 1.0 : P[k][threadIdx.x]) / P[k][k];
        }
        __syncthreads();
        for (int q = 0; q < INV_NB / LPX_WAVES; ++q) {
            const int i = i0 + LPX_WAVES * q;
            P[i][j] = i == k ? rowk[j] : fma(-colk[i], rowk[j], j == k ? 0.0 : P[i][j]);
        }
        __syncthreads();
    }
    for (int q = 0; q < INV_NB / LPX_WAVES; ++q) {
        const int i = i0 + LPX_WAVES * q;
        Pinv[i * INV_NB + j] = P[i][j];
    }
}

// V = W[:, K] with rows K zeroed; W[:, K] = unit columns
__global__ __launch_bounds__(LPX_BLOCK) void k_inv_setup(double *W, int64_t ldw, int64_t m, int64_t k0, int w, double *V)
{
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int64_t r0 = (int64_t)blockIdx.x * INV_ROWS + wave * (INV_ROWS / LPX_WAVES);
    for (int t = 0; t < INV_ROWS / LPX_WAVES; ++t) {
        const int64_t r = r0 + t;
        if (r >= m) break;
        double v = 0.0;
        if (lane < w) {
            const bool inK = r >= k0 && r < k0 + w;
            if (!inK) v = W[r * ldw + k0 + lane];
            W[r * ldw + k0 + lane] = r == k0 + lane ? 1.0 : 0.0;
        }
        V[r * INV_NB + lane] = v;
    }
}

__global__ __launch_bounds__(LPX_BLOCK) void k_inv_invperm(const int64_t *id, int64_t *inv, int64_t m)
{
    const int64_t r = (int64_t)blockIdx.x * LPX_BLOCK + threadIdx.x;
    if (r < m) inv[id[r]] = r;
}
__global__ __launch_bounds__(LPX_BLOCK) void k_inv_gather(const double *W, int64_t ldw, const int64_t *inv, double *out, int64_t ldo,
                                                          int64_t m)
{
    const int64_t i = blockIdx.y;
    const int64_t j = (int64_t)blockIdx.x * LPX_BLOCK + threadIdx.x;
    if (j < m) out[i * ldo + j] = W[i * ldw + inv[j]];
}

// inverts ws.W in place (up to the column permutation kept in ws.id) and writes the inverse to out
static void inv_run(const InvWs &ws, int64_t m, double *out, int64_t ldo, int64_t *info, hipStream_t st)
{
    hipLaunchKernelGGL(k_inv_init, dim3((unsigned)lpx_cdiv(m, LPX_BLOCK)), dim3(LPX_BLOCK), 0, st, ws.id, info, ws.anorm, m);
    hipLaunchKernelGGL(k_absmax, dim3(256), dim3(LPX_BLOCK), 0, st, (const double *)ws.W, m, m, ws.ldw, ws.anorm);
    for (int64_t k0 = 0; k0 < m; k0 += INV_NB) {
        const int w = (int)(m - k0 < INV_NB ? m - k0 : INV_NB);
        const int64_t np = lpx_cdiv(m - k0, INV_ROWS);
        hipLaunchKernelGGL(k_inv_panel_load, dim3((unsigned)np), dim3(LPX_BLOCK), 0, st, (const double *)ws.W, ws.ldw, m, k0, w, ws.S[0],
                           ws.pv[0], ws.pi[0]);
        for (int i = 0; i < w; ++i)
            hipLaunchKernelGGL(k_inv_panel_step, dim3((unsigned)np), dim3(LPX_BLOCK), 0, st, (const double *)ws.S[i & 1], ws.S[(i + 1) & 1],
                               m, k0, w, i, (const double *)ws.pv[i & 1], (const int64_t *)ws.pi[i & 1], np, ws.pv[(i + 1) & 1],
                               ws.pi[(i + 1) & 1], ws.piv, info, (const double *)ws.anorm);
        hipLaunchKernelGGL(k_inv_swap, dim3((unsigned)lpx_cdiv(m + 1, LPX_BLOCK)), dim3(LPX_BLOCK), 0, st, ws.W, ws.ldw, m, k0, w,
                           (const int64_t *)ws.piv, ws.id);
        hipLaunchKernelGGL(k_inv_block, dim3(1), dim3(LPX_BLOCK), 0, st, (const double *)ws.W, ws.ldw, k0, w, ws.Pinv);
        hipLaunchKernelGGL(k_inv_setup, dim3((unsigned)lpx_cdiv(m, INV_ROWS)), dim3(LPX_BLOCK), 0, st, ws.W, ws.ldw, m, k0, w, ws.V);
        double *rows = ws.W + k0 * ws.ldw;
        const GemmArgs gr{ws.Pinv, rows, nullptr, ws.R, w, m, w, INV_NB, ws.ldw, 0, ws.ldw, 1.0, 0.0, 0.0}; // R = P^-1 W[K, :]
        gemm_launch<false>(gr, st);
        hipLaunchKernelGGL(k_copy2d, dim3((unsigned)lpx_cdiv(m, LPX_BLOCK), (unsigned)w), dim3(LPX_BLOCK), 0, st, (const double *)ws.R,
                           ws.ldw, rows, ws.ldw, (int64_t)w, m);
        const GemmArgs gu{ws.V, ws.R, ws.W, ws.W, m, m, w, INV_NB, ws.ldw, ws.ldw, ws.ldw, -1.0, 1.0, 0.0}; // W -= V R
        gemm_launch<false>(gu, st);
    }
    hipLaunchKernelGGL(k_inv_invperm, dim3((unsigned)lpx_cdiv(m, LPX_BLOCK)), dim3(LPX_BLOCK), 0, st, (const int64_t *)ws.id, ws.inv, m);
    hipLaunchKernelGGL(k_inv_gather, dim3((unsigned)lpx_cdiv(m, LPX_BLOCK), (unsigned)m), dim3(LPX_BLOCK), 0, st, (const double *)ws.W,
                       ws.ldw, (const int64_t *)ws.inv, out, ldo, m);
}

// Ainv = A^-1 (fp64, row-major, m x m; Ainv may be A).  info (device, one int64): 0, or 1 + the first column whose
// pivot candidates were all below m * eps * max|A| or NaN (then Ainv is not an inverse) -- numpy's
// LinAlgError("Singular matrix").
extern "C" int lpx_dinverse(int64_t m, const double *A, int64_t lda, double *Ainv, int64_t ldi, void *work, int64_t work_bytes,
                            int64_t *info, void *stream)
{
    LPX_REQUIRE(m > 0 && A && Ainv && info, "null / empty matrix");
    LPX_REQUIRE(lda >= m && ldi >= m, "leading dimension too small");
    LPX_REQUIRE(m <= 65535, "m beyond the grid's y range");
    LPX_REQUIRE(work && work_bytes >= lpx_invert_workspace_bytes(m) && lpx_aligned16(work), "workspace too small");
    const InvWs ws = inv_carve(m, work);
    hipStream_t st = (hipStream_t)stream;
    hipLaunchKernelGGL(k_copy2d, dim3((unsigned)lpx_cdiv(m, LPX_BLOCK), (unsigned)m), dim3(LPX_BLOCK), 0, st, A, lda, ws.W, ws.ldw, m, m);
    inv_run(ws, m, Ainv, ldi, info, st);
    LPX_LAUNCH_CHECK();
    return LPX_OK;
}

// W[i][k] = A[i][basis[k]] as fp64
__global__ __launch_bounds__(LPX_BLOCK) void k_gather_basis(lpx_revised s, double *W, int64_t ld)
{
    const int64_t i = blockIdx.y;
    const int64_t k = (int64_t)blockIdx.x * LPX_BLOCK + threadIdx.x;
    if (k < s.m) W[i * ld + k] = (double)s.A[i * s.ldA + s.basis[k]];
}
__global__ void k_inv_end(lpx_ctrl *ctrl) { ctrl->f32_phase = 0; } // a fresh fp64 inverse ends the float32 phase

// s->Binv = inv(A[:, basis]) from scratch in fp64; update_xB != 0: x_B = B^-1 b afterwards.  info as in lpx_dinverse.
// Opt-in (RevisedEngine.reinvert): the reference's revised solvers never re-invert.
extern "C" int lpx_revised_reinvert_f64(const lpx_revised *s, int update_xB, void *work, int64_t work_bytes, int64_t *info, void *stream)
{
    LPX_REQUIRE(s && s->m > 0 && s->A && s->Binv && s->basis && s->ctrl && s->b && s->xB, "null state");
    LPX_REQUIRE(s->m <= 65535, "m beyond the grid's y range");
    LPX_REQUIRE(work && work_bytes >= lpx_invert_workspace_bytes(s->m) && lpx_aligned16(work), "workspace too small");
    LPX_REQUIRE(info != nullptr, "null info");
    const int64_t m = s->m;
    const InvWs ws = inv_carve(m, work);
    hipStream_t st = (hipStream_t)stream;
    hipLaunchKernelGGL(k_gather_basis, dim3((unsigned)lpx_cdiv(m, LPX_BLOCK), (unsigned)m), dim3(LPX_BLOCK), 0, st, *s, ws.W, ws.ldw);
    inv_run(ws, m, s->Binv, s->ldB, info, st);
    hipLaunchKernelGGL(k_inv_end, dim3(1), dim3(1), 0, st, s->ctrl);
    if (update_xB) hipLaunchKernelGGL(k_refine_xb, dim3((unsigned)lpx_cdiv(m, LPX_WAVES)), dim3(LPX_BLOCK), 0, st, *s);
    LPX_LAUNCH_CHECK();
    return LPX_OK;
}
