// lpx_tableau_blocked_decl.hpp -- workspace record of the deferred-update chain (lpx_tableau_blocked.hpp)
#pragma once
#include <stdint.h>

#define TB_KMAX 32 /* largest block; E rows are TB_KMAX doubles apart whatever the block size in use */
#define TB_THREADS 128 /* per-pivot kernels: 128-thread workgroups spread evenly over the CUs the pipelined chain leaves them */

struct TbWs {
    double *E;         // (m + 2) x TB_KMAX: E[i][u] = eta column of pending update u, row i (the pass over T reads rows)
    double *Et;        // TB_KMAX x lde: the same, transposed (the column kernel reads E[.][u] for its rows, coalesced)
    int64_t lde;
    double *R;         // TB_KMAX x ldr: R[u][j] = row r_u of the tableau as pending updates 0..u-1 leave it
    int64_t ldr;
    double *rc;        // n: reduced-cost row, current
    double *xb;        // m: x_B column, current
    int64_t *prow;     // TB_KMAX: r_u
    int64_t *pslot;    // TB_KMAX: column of T that update u rewrites as a whole
    uint32_t *rowflag; // m + 1: bit u set = row is r_u (the few rows that are not a plain fma chain)
    int64_t *cnt;      // pending updates
    double *scal;      // [0] column slot of the entering variable of the pivot in flight
    double *dbest, *xbest; // ratio partials' payload: entering-column entry and x_B of the partial's argmin row
    int64_t *tsel;     // pipelined chain: [0] = buffer that holds the tableau once the passes of the pivots done so far have run
                       // (0: T, 1: T_alt); written with every pivot, read by the call's closing copy-back
};

// the pass over the tableau (lpx_tb_flush.hpp, compiled in lpx_tb_flush.hip)
struct TbFlushArgs {
    const double *T;        // source
    double *Tdst;           // destination: T itself (in place) or the other buffer of the pipelined chain
    int64_t rows, cols, ld; // cols = n + 1 (column n is x_B); ld even
    int cond;               // condensed layout: a rewritten column receives the eta column, else a unit vector
    const double *E;        // rows x TB_KMAX
    const double *R;        // TB_KMAX x ldr
    int64_t ldr;
    const int64_t *pslot;
    const uint32_t *rowflag;
    const int64_t *cnt;
    int64_t rpb;            // rows per workgroup (band height)
};
#define TB2_THREADS 256 /* 4 waves: 3 workgroups per CU at 165 VGPRs (512: one) -- 1.17 against 1.21 ms per pass */
#define TB2_WAVES (TB2_THREADS / 64)
#define TB2_COLS 128
// kb in {8, 16, 32}; nt: non-temporal loads / stores; grid = (strips of TB2_COLS columns, bands of rpb rows)
void lpx_tb_flush_launch(const TbFlushArgs &f, int kb, int nt, void *stream);
