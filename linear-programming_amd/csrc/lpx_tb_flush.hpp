// lpx_tb_flush.hpp -- the one HBM-bound kernel of the deferred-update chain (lpx_tableau_blocked.hpp):
//     T <- T with the cnt pending rank-1 updates applied to every element, in order.
// Algorithmic bytes per launch: 16 x rows x cols (every element read once, written once) + the pending pairs
// (8 cnt (rows + cols), re-read from L2 by every workgroup that needs them); fma per element: cnt.
//
// Mapping.  A workgroup owns ONE narrow strip (128 columns: one wave-width of column pairs) of a band of rows and walks
// down it; its 4 waves take turns on groups of RG rows.  The strip's entries of the cnt pivot rows live in LDS
// (cnt x 1 KiB), read once per group (one ds_read_b128 per update feeds 2 RG fma), so a wave's registers hold only data
// in flight: RG rows of the tableau and their multipliers.  The multipliers e_u[i] of a row are the same for every lane:
// each lane loads the one with its index modulo 16 (per group of 16 updates) and the fma takes it from there through a
// DPP row broadcast.  (Tried first: pivot-row strips in registers -- 2 waves per SIMD and too little in flight; the
// multipliers as scalar loads -- four dependent scalar-cache misses per row with no SGPRs left to hide them behind.)
// The two special cases of an update -- the pivot row itself (x <- e * R) and the column it rewrites (x <- e) -- are
// the same fma on an accumulator reset to -0.0 (fma(e, R, -0.0) == e * R and fma(e, 1, -0.0) == e, bit for bit): one
// wave-uniform test per update, so no row and no column leaves the fma stream (condensed layout; the full layout's
// unit vectors take a general path).
#pragma once
#include "lpx_device.hpp"
#include "lpx_rank1.hpp"
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

// x += e_N * pr with e_N taken from lane N of the lane's own row of 16 lanes (lpx_device.hpp, lpx_fmac_row_bcast): the 16
// multipliers of a group sit one per lane, no scalar loads, no readlanes
template <int N>
__device__ __forceinline__ void tb_fmac_bcast(double &x, double e16, double pr)
{
    lpx_fmac_row_bcast<N>(x, e16, pr);
}
template <int N>
__device__ __forceinline__ double tb_bcast(double e16)
{
    const long long b = __double_as_longlong(e16);
    const unsigned lo = (unsigned)__builtin_amdgcn_readlane((int)(b & 0xffffffffLL), N);
    const unsigned hi = (unsigned)__builtin_amdgcn_readlane((int)(b >> 32), N);
    return __longlong_as_double((long long)(((unsigned long long)hi << 32) | lo));
}

// the chain of updates U .. KB-1 on a wave's group of RG rows; eg[q][g] = E[row q][16 g + lane % 16]; a lane owns CV
// column pairs, 128 columns apart (x[q][2 v], x[q][2 v + 1]; fm[c]: the updates that rewrite its column c)
template <int KB, int RG, int CV, bool FULL, int U>
struct Tb2Chain {
    // special: bit U set = this wave resets something before update U (a pivot row in the group, a rewritten column)
    static __device__ __forceinline__ void fast(double (&x)[RG][2 * CV], const double (&eg)[RG][(KB + 15) / 16], const d2_t *prs, int cnt,
                                                uint32_t special, const uint32_t (&rf)[RG], const uint32_t (&fm)[2 * CV])
    {
        if (FULL || U < cnt) {
            d2_t pr[CV];
#pragma unroll
            for (int v = 0; v < CV; ++v) pr[v] = prs[(U * CV + v) * 64];
            if ((special >> U) & 1u) {
#pragma unroll
                for (int q = 0; q < RG; ++q) {
                    const bool row = (rf[q] >> U) & 1u;
#pragma unroll
                    for (int c = 0; c < 2 * CV; ++c) x[q][c] = (row || ((fm[c] >> U) & 1u)) ? -0.0 : x[q][c];
                }
            }
#pragma unroll
            for (int q = 0; q < RG; ++q) {
#pragma unroll
                for (int v = 0; v < CV; ++v) {
                    tb_fmac_bcast<U % 16>(x[q][2 * v], eg[q][U / 16], pr[v][0]);
                    tb_fmac_bcast<U % 16>(x[q][2 * v + 1], eg[q][U / 16], pr[v][1]);
                }
            }
        }
        Tb2Chain<KB, RG, CV, FULL, U + 1>::fast(x, eg, prs, cnt, special, rf, fm);
    }
    static __device__ __forceinline__ void slow(double (&x)[RG][2 * CV], const double (&eg)[RG][(KB + 15) / 16], const d2_t *prs, int cnt,
                                                const uint32_t (&rf)[RG], const uint32_t (&fm)[2 * CV], bool cond)
    {
        if (FULL || U < cnt) {
#pragma unroll
            for (int v = 0; v < CV; ++v) {
                const d2_t pr = prs[(U * CV + v) * 64];
#pragma unroll
                for (int q = 0; q < RG; ++q) {
                    const double e = tb_bcast<U % 16>(eg[q][U / 16]);
                    const bool isr = (rf[q] >> U) & 1u;
                    x[q][2 * v] = tb_apply(x[q][2 * v], (fm[2 * v] >> U) & 1u, isr, e, pr[0], cond);
                    x[q][2 * v + 1] = tb_apply(x[q][2 * v + 1], (fm[2 * v + 1] >> U) & 1u, isr, e, pr[1], cond);
                }
            }
        }
        Tb2Chain<KB, RG, CV, FULL, U + 1>::slow(x, eg, prs, cnt, rf, fm, cond);
    }
};
template <int KB, int RG, int CV, bool FULL>
struct Tb2Chain<KB, RG, CV, FULL, KB> {
    static __device__ __forceinline__ void fast(double (&)[RG][2 * CV], const double (&)[RG][(KB + 15) / 16], const d2_t *, int, uint32_t,
                                                const uint32_t (&)[RG], const uint32_t (&)[2 * CV]) {}
    static __device__ __forceinline__ void slow(double (&)[RG][2 * CV], const double (&)[RG][(KB + 15) / 16], const d2_t *, int,
                                                const uint32_t (&)[RG], const uint32_t (&)[2 * CV], bool) {}
};

template <int KB, int NT, int RG, bool FULL, int THREADS, int CV, int ONLY = 0>
__device__ __forceinline__ void tb2_walk(const TbFlushArgs &a, int cnt, const d2_t *prs, const uint32_t (&fm)[2 * CV], uint32_t fxmask,
                                         const int64_t (&col)[CV], const bool (&live)[CV], int64_t row0, int64_t row1)
{
    constexpr int NG = (KB + 15) / 16;
    const bool cond = a.cond != 0;
    const int64_t ld = a.ld;
    const int wave = threadIdx.x >> 6;
    const double *__restrict__ El = a.E + (threadIdx.x & 15);
    const uint32_t *__restrict__ rowflag = a.rowflag;
    for (int64_t i0 = row0 + wave * RG; i0 < row1; i0 += (THREADS / 64) * RG) {
        double x[RG][2 * CV], eg[RG][NG];
        uint32_t rf[RG];
#pragma unroll
        for (int q = 0; q < RG; ++q) {
            // rows beyond the band (ragged last group) re-read the band's last row and are not stored
            const int64_t i = (i0 + q < row1) ? i0 + q : row1 - 1;
#pragma unroll
            for (int v = 0; v < CV; ++v) {
                const d2_t t = lpx_ld2<NT>(a.T + i * ld + col[v]);
                x[q][2 * v] = t[0];
                x[q][2 * v + 1] = t[1];
            }
#pragma unroll
            for (int g = 0; g < NG; ++g) eg[q][g] = El[i * TB_KMAX + 16 * g];
            rf[q] = rowflag[i];
        }
        uint32_t rfany = 0u;
#pragma unroll
        for (int q = 0; q < RG; ++q) {
            rf[q] = __builtin_amdgcn_readfirstlane(rf[q]);
            rfany |= rf[q];
        }
        // condensed layout: every case is an fma (the strip in LDS holds 1.0 in the rewritten columns); full layout:
        // unit vectors, general path for the groups and strips that meet one
        if (ONLY == 1 || cond) Tb2Chain<KB, RG, CV, FULL, 0>::fast(x, eg, prs, cnt, rfany | fxmask, rf, fm);
        else if (rfany == 0u && fxmask == 0u) Tb2Chain<KB, RG, CV, FULL, 0>::fast(x, eg, prs, cnt, 0u, rf, fm);
        else Tb2Chain<KB, RG, CV, FULL, 0>::slow(x, eg, prs, cnt, rf, fm, cond);
#pragma unroll
        for (int v = 0; v < CV; ++v) {
            if (live[v]) {
#pragma unroll
                for (int q = 0; q < RG; ++q)
                    if (i0 + q < row1) lpx_st2<NT>(a.Tdst + (i0 + q) * ld + col[v], d2_t{x[q][2 * v], x[q][2 * v + 1]});
            }
        }
    }
}

// the pass of one workgroup: strip bx of CV * TB2_COLS columns, band by of rpb rows; prs_all = KB * 64 * CV pairs of LDS
template <int KB, int NT, int RG, int THREADS, int CV, int ONLY>
__device__ __forceinline__ void tb2_flush_body(const TbFlushArgs &a, int64_t bx, int64_t by, d2_t *prs_all)
{
    const int cnt = (int)*a.cnt;
    if (cnt <= 0) return;
    if (ONLY == 1 && (cnt != KB || !a.cond)) return;
    if (ONLY == 2 && cnt == KB && a.cond) return;
    const int lane = threadIdx.x & 63;
    const int64_t col0 = bx * (TB2_COLS * CV);
    // every lane stays active (a DPP broadcast from a lane that has left writes nothing): lanes beyond the last column
    // work on column 0's data and store nothing
    int64_t col[CV];
    bool live[CV];
#pragma unroll
    for (int v = 0; v < CV; ++v) {
        const int64_t c = col0 + v * TB2_COLS + 2 * lane;
        live[v] = c < a.cols;
        col[v] = live[v] ? c : 0;
    }
    for (int idx = threadIdx.x; idx < KB * 64 * CV; idx += THREADS) {
        const int u = idx / (64 * CV), v = (idx >> 6) % CV;
        const int64_t c = col0 + v * TB2_COLS + 2 * (idx & 63);
        d2_t val = (u < cnt && c < a.cols) ? *(const d2_t *)(a.R + u * a.ldr + c) : d2_t{0.0, 0.0};
        if (a.cond && u < cnt) { // the column this update rewrites receives e itself: fma(e, 1, -0.0)
            const int64_t su = a.pslot[u];
            if (c == su) val[0] = 1.0;
            if (c + 1 == su) val[1] = 1.0;
        }
        prs_all[idx] = val;
    }
    uint32_t fm[2 * CV];
#pragma unroll
    for (int c = 0; c < 2 * CV; ++c) fm[c] = 0u;
    for (int u = 0; u < cnt; ++u) {
        const int64_t su = a.pslot[u];
#pragma unroll
        for (int v = 0; v < CV; ++v) {
            const int64_t c = col0 + v * TB2_COLS + 2 * lane;
            fm[2 * v] |= (uint32_t)(live[v] && c == su) << u;
            fm[2 * v + 1] |= (uint32_t)(live[v] && c + 1 == su) << u;
        }
    }
    uint32_t fmany = 0u;
#pragma unroll
    for (int c = 0; c < 2 * CV; ++c) fmany |= fm[c];
    uint32_t fxmask = 0u; // wave-uniform: updates that rewrite a column of this strip
    for (int u = 0; u < cnt; ++u)
        if (__ballot((fmany >> u) & 1u) != 0ull) fxmask |= 1u << u;
    __syncthreads();
    const int64_t row0 = by * a.rpb;
    const int64_t row1 = (row0 + a.rpb < a.rows) ? row0 + a.rpb : a.rows;
    const d2_t *prs = prs_all + lane;
    if (ONLY == 1) tb2_walk<KB, NT, RG, true, THREADS, CV, 1>(a, cnt, prs, fm, fxmask, col, live, row0, row1);
    else if (cnt == KB) tb2_walk<KB, NT, RG, true, THREADS, CV>(a, cnt, prs, fm, fxmask, col, live, row0, row1);
    else tb2_walk<KB, NT, RG, false, THREADS, CV>(a, cnt, prs, fm, fxmask, col, live, row0, row1);
}

// grid (strips of CV * TB2_COLS columns, bands of rpb rows)
// ONLY = 1: the condensed layout's full blocks alone (every case an fma: the instance whose register count decides the
// occupancy of the large-tableau loop); ONLY = 2: everything else; 0: both in one kernel
template <int KB, int NT, int RG, int THREADS = TB2_THREADS, int CV = 1, int ONLY = 0>
__global__ __launch_bounds__(THREADS) void k_tb_flush2(TbFlushArgs a)
{
    __shared__ d2_t prs_all[KB * 64 * CV];
    tb2_flush_body<KB, NT, RG, THREADS, CV, ONLY>(a, (int64_t)blockIdx.x, (int64_t)blockIdx.y, prs_all);
}
