// lpx_refine.hip -- opt-in numerical hygiene for the revised form: refine the product-form inverse in fp64.
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
#pragma unroll
    for (int i = 0; i < WT; ++i)
#pragma unroll
        for (int j = 0; j < WT; ++j)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int64_t r = row0 + wm + i * 16 + (lane >> 4) + 4 * q, c = col0 + wn + j * 16 + (lane & 15);
                if (r < g.M && c < g.N) {
                    double v = g.alpha * acc[i][j][q];
                    if (g.C) v = fma(g.beta, g.C[r * g.ldc + c], v);
                    if (r == c) v += g.gamma;
                    g.D[r * g.ldd + c] = v;
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
