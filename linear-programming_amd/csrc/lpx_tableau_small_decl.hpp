// lpx_tableau_small_decl.hpp -- launch record of the chain for launch-bound tableaus (lpx_tableau_small.hip)
#pragma once
#include "lpx.h"
#include "lpx_tableau_blocked_decl.hpp"

#define TS_KB 8 /* pivots per launch */
#define TS_THREADS 1024

struct TsArgs {
    // decisions: s.T = the stale buffer, b / o = this block's / the previous block's pairs, u = pivots to decide, uo = pairs
    // of the previous block, dstid = buffer that holds the tableau once this block's pass has run
    lpx_tableau s;
    TbWs b, o;
    int u, uo, dstid;
    TbFlushArgs f; // the pass of the previous block (f.cnt: its pair count; no pass when decide-only)
    const double *fEt; // its eta columns, transposed (TbWs::Et: this chain moves them as rows; f.E is not used)
    int64_t flde;
    int decide, pass;
    int64_t nstrips;
};
// grid = 1 (decide only) or 1 + strips x bands of the pass
void lpx_tb_small_launch(const TsArgs &a, unsigned grid, void *stream);
