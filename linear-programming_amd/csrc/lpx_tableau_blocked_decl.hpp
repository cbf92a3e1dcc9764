// lpx_tableau_blocked_decl.hpp -- workspace record of the deferred-update chain (lpx_tableau_blocked.hpp)
#pragma once
#include <stdint.h>

#define TB_KMAX 32 /* largest block; E rows are TB_KMAX doubles apart whatever the block size in use */
#define TB_THREADS 256

struct TbWs {
    double *E;         // (m + 2) x TB_KMAX: E[i][u] = eta column of pending update u, row i
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
};
