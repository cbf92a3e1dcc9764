/*
 * lpx.h -- C ABI of liblpx.so: the MI355X (gfx950) dense simplex pivot path.
 *
 * The reference (agmuth/linear-programming, pure Python + NumPy) has no FFI; its boundary is the Python
 * solver-class surface.  liblpx is what a maintainer would bind from those classes with ctypes: one entry
 * point per NumPy call site of the pivot loop.  Every entry point cites the reference function it replaces
 * (file:line relative to the reference checkout).  INTEGRATION.md shows the ctypes stubs.
 *
 * Conventions
 *   - plain C: raw device pointers, int64 sizes, `void *stream` is a hipStream_t (NULL = default stream);
 *   - every function returns 0 on success, a negative LPX_E_* code otherwise (never throws, never exits);
 *     lpx_last_error() returns a message for the calling thread's last failure;
 *   - the library allocates NO device memory: all buffers, including scratch (`workspace`), are owned by the
 *     caller (the Python host uses torch tensors); *_workspace_bytes() says how much scratch a problem needs;
 *   - all calls are asynchronous on `stream` except lpx_read_ctrl (which synchronises the stream);
 *   - matrices are row-major; leading dimensions are in elements; all base pointers 16-byte aligned and
 *     leading dimensions multiples of 4 (float) / 2 (double) so that rows stay 16-byte aligned.
 *   - the solver state lives in HBM across iterations; the only per-iteration host traffic is the optional
 *     read of the 128-byte control block (lpx_ctrl).
 */
#ifndef LPX_H
#define LPX_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define LPX_VERSION 1

/* error codes */
#define LPX_OK 0
#define LPX_E_ARG (-1)  /* bad argument (null pointer, misaligned, size mismatch, workspace too small) */
#define LPX_E_HIP (-2)  /* a HIP runtime call failed */
#define LPX_E_SIZE (-3) /* problem exceeds a compiled-in limit of the entry point */

/* lpx_ctrl.status: maps 1:1 onto the reference's loop exits / exceptions (linprog/exceptions.py:1-22) */
#define LPX_RUNNING 0          /* loop not finished (maxiters reached => optimum False/None) */
#define LPX_OPTIMAL 1          /* primal_solvers.py:195-198, dual_solvers.py:99-102, special_solvers.py:230-233 */
#define LPX_PRIMAL_UNBOUNDED 2 /* PrimalIsUnboundedError: primal_solvers.py:130-132, special_solvers.py:256-257 */
#define LPX_DUAL_UNBOUNDED 3   /* DualIsUnboundedError: dual_solvers.py:41-44 */
#define LPX_NOT_NONBASIC 4     /* ValueError at special_solvers.py:331 */

/* var_state[j] */
#define LPX_VS_BASIC 0
#define LPX_VS_LOWER 1    /* nonbasic at its lower bound (every nonbasic variable of the standard-form solvers) */
#define LPX_VS_UPPER 2    /* nonbasic at its upper bound (bounded-variable solver) */
#define LPX_VS_EXCLUDED 3 /* column outside the restricted primal (primal-dual algorithm) */

/* lpx_tableau.pricing.  The reference has exactly one entering rule in its standard-form solvers: the first
 * column with a negative reduced cost (primal_solvers.py:135-138).  LPX_PRICE_DANTZIG (most negative reduced cost,
 * lowest index on ties) is the build's opt-in extension for large solves; it has no counterpart in the reference
 * ("parity unpinned"): it is checked against the oracle's restatement of the same rule only. */
#define LPX_PRICE_FIRST_NEGATIVE 0
#define LPX_PRICE_DANTZIG 1
/* ratio test of the tableau loop (lpx_tableau.ratio_rule) */
#define LPX_RATIO_TEXTBOOK 0
#define LPX_RATIO_HARRIS 1

/* lpx_revised_iterate `kind` */
#define LPX_KIND_PRIMAL 0  /* PrimalRevisedSimplexSolver.solve   primal_solvers.py:176-209 */
#define LPX_KIND_DUAL 1    /* DualRevisedSimplexSolver.solve     dual_solvers.py:82-108 */
#define LPX_KIND_BOUNDED 2 /* BoundedVariablePrimalSimplexSolver.solve  special_solvers.py:221-343 */
#define LPX_KIND_PRIMAL_NAIVE 3 /* PrimalNaiveSimplexSolver.solve: float32 re-inversion per pivot, :92-94, :169-174 */
#define LPX_KIND_DUAL_NAIVE 4   /* DualNaiveSimplexSolver.solve */
/* Opt-in extension, lpx_revised_solve_small / _batch only: the DUAL simplex for lb <= x <= ub (BASELINE config 5's "dual
 * simplex with bounded variables").  No counterpart in the reference -- its bounded solver is the primal one
 * (special_solvers.py:94-343), its dual solver knows x >= 0 only (dual_solvers.py:47-108) -- so parity is unpinned against
 * it; the rules are the reference's dual rules generalised: leaving row = first basic variable outside its bounds
 * (optimal when the worst violation is <= 1e-8), ratio test theta_j = (-rc_j) / alpha_j over the nonbasic columns that
 * can repair it (at lower with alpha_j < 0, at upper with alpha_j > 0; alpha = the dual row, negated when x_r is above
 * its bound), first argmin; the leaving variable goes to the bound it violated; x_B recomputed as :169-175.  float64
 * throughout.  Oracle: lpo_dual_bounded_solve. */
#define LPX_KIND_DUAL_BOUNDED 5

/* Device-resident control block (128 bytes).  Written only by kernels (and by lpx_ctrl_reset); the host
 * reads it with lpx_read_ctrl.  `counter` is the reference's self.counter. */
typedef struct lpx_ctrl {
    int32_t status;
    int32_t f32_phase;    /* 1 while B^-1 / x_B still hold float32-computed values */
    int64_t counter;      /* iterations started in the current solve() call (primal_solvers.py:189-192) */
    int64_t pivots;       /* state changes since construction; also the write index of the pivot log */
    int64_t enter_j;      /* entering column of the last iteration */
    int64_t leave_r;      /* leaving row of the last iteration (-1: bound flip) */
    int64_t leave_var;    /* variable that left */
    double price_val;     /* min reduced cost / min x_B / max bounded reduced cost of the last iteration */
    double ratio_val;     /* min ratio of the last ratio test */
    double aux_val;       /* max(d) (primal), min(row) (dual) */
    int32_t flip;         /* bounded: 1 if the last iteration was a bound flip (no basis change) */
    int32_t leave_state;  /* bounded: LPX_VS_LOWER / LPX_VS_UPPER the leaving variable went to */
    int64_t enter_slot;   /* tableau form: local column of T that the update of the last pivot rewrites as a whole, or -1 */
    int64_t reserved[5];
} lpx_ctrl;

/* State of one LP in revised form (explicit B^-1, primal_solvers.py:13-41).  A host-side struct of DEVICE
 * pointers, passed by pointer to every call and copied by value into the kernels' arguments. */
typedef struct lpx_revised {
    int64_t m, n;       /* rows, columns of A */
    int64_t ldA;        /* >= n, multiple of 4 */
    int64_t ldAt;       /* >= m, multiple of 4 */
    int64_t ldB;        /* >= m, multiple of 2 */
    const float *A;     /* m x ldA   float32 values (preprocessing.py:8) */
    const float *At;    /* n x ldAt  transposed copy of A (column j of A contiguous) */
    const float *b;     /* m */
    const float *c;     /* n */
    double *Binv;       /* m x ldB   inv_basis_matrix */
    double *xB;         /* m         bfs */
    int64_t *basis;     /* m */
    int32_t *var_state; /* n */
    const double *lb;   /* n, bounded solver only (else NULL) */
    const double *ub;   /* n, bounded solver only (else NULL) */
    lpx_ctrl *ctrl;
    int64_t *log;       /* 3*log_cap (row, entering, leaving var) per state change, or NULL */
    int64_t log_cap;
    void *workspace;
    int64_t workspace_bytes;
    void *naive_workspace;  /* naive solvers only (else NULL): lpx_naive_workspace_bytes(m) bytes */
    int64_t naive_workspace_bytes;
} lpx_revised;

/* State of one LP in tableau form: T = [B^-1 A | x_B ; rc | -z], (m+1) x ld, carried and updated in place.
 * The build's own data layout for the large single-GPU / column-partitioned configs (SURVEY.md Appendix A,
 * "Tableau equivalent"); the pivot rules are those of primal_solvers.py:135-167.
 * Column-partitioned: this rank holds global columns [col0, col0+n) and a replica of the x_B column. */
typedef struct lpx_tableau {
    int64_t m, n, ld;   /* n = local structural columns; column n of T is x_B; ld >= n+1, multiple of 2 */
    int64_t col0;       /* global index of local column 0 (0 when unpartitioned) */
    int64_t n_global;   /* total structural columns over all ranks */
    int64_t pricing;    /* LPX_PRICE_FIRST_NEGATIVE (the reference's rule, default) or LPX_PRICE_DANTZIG (opt-in) */
    double *T;
    int64_t *basis;     /* m, global column indices (replicated) */
    int32_t *var_state; /* n (local columns) */
    lpx_ctrl *ctrl;
    int64_t *log;
    int64_t log_cap;
    void *workspace;
    int64_t workspace_bytes;
    /* Condensed form (both non-NULL): T holds the NONBASIC columns only -- basic columns are
     * unit vectors that no pivot changes -- so n counts column slots, slot_var[k] is the variable in slot k and
     * var_slot[j] (n_global entries) the slot of variable j on this rank or -1.  col0 is unused.  Same arithmetic, about
     * half the memory and HBM traffic per pivot at n = 2m.  Both NULL: every column is carried (full form). */
    int64_t *slot_var;
    int64_t *var_slot;
    /* Optional second buffer of the size of T ((m+1) x ld doubles, 16-byte aligned), or NULL.  With it, lpx_tableau_iterate
     * and lpx_tableau_dual_iterate run tableaus of up to 16384 rows and 32767 columns as ONE launch per pivot: the
     * tableau ping-pongs between T and T_alt while lookahead workgroups of the same launch prepare the next pivot's
     * decisions (lpx_tableau.hip); primal tableaus of up to 1023 rows and 1023 carried columns as one launch per 8
     * pivots (lpx_tableau_chain_kind, LPX_CHAIN_SMALL).  Large tableaus (lpx_tableau_pipelined() != 0) use it for the PIPELINED deferred-update
     * chain instead: the pass over the tableau streams T -> T_alt (then back) while the next block's pivots are decided
     * beside it.  The state is in T again whenever a call's work has drained; T_alt's contents are scratch between
     * calls.  Same arithmetic, same bits. */
    double *T_alt;
    /* Ratio test of lpx_tableau_iterate: LPX_RATIO_TEXTBOOK (0, the reference's: primal_solvers.py:149-167, utils.py:5-10)
     * or LPX_RATIO_HARRIS (opt-in, never the default; no counterpart in the reference): two passes with tolerance
     * ratio_tol -- theta_max = min over d_i > 0 of (x_i + ratio_tol) / d_i, then among the rows with d_i > 0 and
     * x_i / d_i <= theta_max the largest pivot d_i, first row on ties (oracle: lpo_ratio_harris).  Every other entry
     * point takes LPX_RATIO_TEXTBOOK only. */
    int64_t ratio_rule;
    double ratio_tol;
} lpx_tableau;

/* ---------------------------------------------------------------------------------------------- misc */
int lpx_version(void);
const char *lpx_last_error(void);
/* Resolve every kernel instance the library's launchers can reach (hipFuncGetAttributes on each kernel handle; the list
 * is generated at build time from the host stubs of the library's objects) and check that the current device is gfx950.
 * LPX_OK, or LPX_E_HIP with lpx_last_error() naming the kernel -- instead of an abort inside its first launch.  No
 * counterpart in the reference; run it once per process before trusting a new build (smoke(), bench.py and
 * `LPX_SELFCHECK=1` imports do).  *kernels_checked (may be NULL) receives the number resolved. */
int lpx_selfcheck(int64_t *kernels_checked);
/* Zero a control block and set f32_phase (solver construction: primal_solvers.py:37-40). */
int lpx_ctrl_init(lpx_ctrl *dev_ctrl, int32_t f32_phase, void *stream);
/* Start of a solve() call: counter = 0, status = RUNNING (primal_solvers.py:189-190, dual_solvers.py:95). */
int lpx_ctrl_begin_solve(lpx_ctrl *dev_ctrl, void *stream);
/* Copy the control block to the host and synchronise the stream: the "scalar status flag" round trip. */
int lpx_read_ctrl(const lpx_ctrl *dev_ctrl, lpx_ctrl *host_ctrl, void *stream);

/* Measurement hooks (no counterpart in the reference): while enabled, HIP events are recorded on the launch
 * stream around every launch of the rank-1 pivot-update kernel (at most 4096 launches are kept).
 * lpx_profile_k1_read synchronises on them and returns the summed kernel time and the launch count. */
int lpx_profile_k1(int enable);
int lpx_profile_k1_read(double *total_ms, int64_t *launches);

/* ------------------------------------------------------------------------------ revised-form solver loop */
int64_t lpx_revised_workspace_bytes(int64_t m, int64_t n);

/* Enqueue `iters` iterations of the solver loop `kind` (no host synchronisation; kernels become no-ops
 * once ctrl->status != RUNNING).  Replaces the bodies of
 *   PrimalNaiveSimplexSolver.solve              primal_solvers.py:191-207 (with the revised pivot :255-263),
 *   DualNaiveSimplexSolver.solve                dual_solvers.py:96-106,
 *   BoundedVariablePrimalSimplexSolver.solve    special_solvers.py:225-341. */
int lpx_revised_iterate(const lpx_revised *s, int kind, int64_t iters, void *stream);

/* The same loops as ONE launch for a small LP (kind = LPX_KIND_PRIMAL, LPX_KIND_DUAL, LPX_KIND_BOUNDED or the opt-in
 * LPX_KIND_DUAL_BOUNDED): a whole solve()
 * call -- up to `maxiters` iterations of primal_solvers.py:191-207 / dual_solvers.py:96-106 / special_solvers.py:225-341 --
 * by one workgroup with B^-1 in LDS, state read from and written back to *s (control block included; call
 * lpx_ctrl_begin_solve first, as for lpx_revised_iterate).  A single LP of a few hundred rows is launch-bound on the
 * multi-kernel chain (~40 us per iteration whatever its size); here an iteration costs a few microseconds.
 * lpx_revised_small_lds_bytes(m, n, kind) <= 158 KiB says whether a shape fits (m <= 128 at n = 2m), else LPX_E_SIZE.
 * lpx_revised_small_a_in_lds(m, n, kind) != 0: a copy of A fits beside the state and the kernel makes one (column-major
 * for the primal / dual loops, row-major for the bounded ones: every chain then reads 16 bytes per LDS instruction); without
 * it the chains over A wait for global memory once per 32 steps -- the bounded loops, whose x_B is an n-long chain along a
 * row of A every iteration, are then slower than the multi-kernel chain (43 against 37 us per iteration at 96 x 192).
 * Dot products are accumulated in index order with fma: bit-identical to the oracle's loops.
 * lpx_revised_solve_small_batch: `batch` such states (a DEVICE array of lpx_revised, all m x n), one workgroup each; x
 * (batch x n) and cost (batch) receive the expanded solutions when non-NULL (special_solvers.py:201-206 for bounded).
 * Every state's control block must be RUNNING (lpx_ctrl_begin_solve per LP): a workgroup whose LP is in any other state
 * returns at once and leaves that LP's rows of x and cost UNTOUCHED (stale, not zeroed). */
int64_t lpx_revised_small_lds_bytes(int64_t m, int64_t n, int kind);
int lpx_revised_small_a_in_lds(int64_t m, int64_t n, int kind);
int lpx_revised_solve_small(const lpx_revised *s, int kind, int64_t maxiters, void *stream);
int lpx_revised_solve_small_batch(const lpx_revised *dev_states, int64_t batch, int64_t m, int64_t n, int kind,
                                  int64_t maxiters, double *x, double *cost, void *stream);

/* PrimalRevisedSimplexSolver.pivot(row, col), primal_solvers.py:255-263 (called from outside the loop by
 * PhaseOneSimplexSolver, special_solvers.py:77-79). */
int lpx_revised_pivot(const lpx_revised *s, int64_t row, int64_t col, void *stream);

/* Naive solvers: inv_basis_matrix = inv(A[:, basis]) in float32 (Gauss-Jordan, partial pivoting) and
 * bfs = inv_basis_matrix @ b, primal_solvers.py:92-94 and :68-70 (the reference calls LAPACK through
 * np.linalg.inv; rounding differs in the last float32 bits). */
int64_t lpx_naive_workspace_bytes(int64_t m);
int lpx_revised_reinvert(const lpx_revised *s, void *stream);
/* PrimalNaiveSimplexSolver.pivot(row, col), primal_solvers.py:169-174: basis[row] = col, re-invert, bfs. */
int lpx_naive_pivot(const lpx_revised *s, int64_t row, int64_t col, void *stream);

/* _get_bfs_expanded + _calc_current_cost, primal_solvers.py:72-79 (bounded: special_solvers.py:201-206):
 * x[n] = scatter of x_B (plus nonbasic bound values when lb/ub are set); cost[0] = dot(c, x). */
int lpx_revised_expand(const lpx_revised *s, double *x, double *cost, void *stream);

/* Row `row` of the tableau B^-1 A into out[n] (float64): what PhaseOneSimplexSolver reads of
 * `(solver.inv_basis_matrix @ solver.A)[row, :]` to pivot a basic artificial variable out, special_solvers.py:69-71 --
 * one row instead of the reference's whole m x n product, and no download of B^-1. */
int lpx_revised_tableau_row(const lpx_revised *s, int64_t row, double *out, void *stream);

/* BoundedVariablePrimalSimplexSolver._update_bfs, special_solvers.py:169-175. */
int lpx_bounded_update_bfs(const lpx_revised *s, void *stream);

/* Reduced costs into rc[n] (primal_solvers.py:59-66; bounded=1: special_solvers.py:177-199). */
int lpx_revised_reduced_costs(const lpx_revised *s, int bounded, double *rc, void *stream);

/* ----------------------------------------------------------------- primal-dual restricted-problem updates */
/* special_solvers.py:405-406: v = y @ A ; var_state[j] = LOWER if isclose(v_j, c_j) else EXCLUDED, for the
 * first n_orig columns (the artificial block behind them stays LOWER).  c64 = float64 costs of the
 * unrestricted problem. */
int lpx_pd_admissible(const lpx_revised *s, const double *y, const double *c64, int64_t n_orig, double *v,
                      void *stream);
/* special_solvers.py:430-444: y_rp = c_B B^-1 of the solved restricted primal; u = y_rp @ A;
 * out[0] = theta = min over excluded j with u_j > 0 of (c64_j - v_j)/u_j ; out[1] = max(u_j over admissible j,
 * y_rp_i) (dual unbounded iff <= 0, :435-436); then y += theta * y_rp unless out[1] <= 0.
 * cost_gate (device pointer or NULL): the restricted primal's cost as lpx_revised_expand left it -- the reference takes
 * this step only while it is > 0 (:427) and stops otherwise (:445-447); with the gate the host need not fetch the cost
 * before enqueueing the step: y is left alone when *cost_gate <= 0, and out[2] = *cost_gate (out has 3 doubles then). */
int lpx_pd_dual_step(const lpx_revised *s, double *y, const double *c64, const double *v, int64_t n_orig,
                     double *out, const double *cost_gate, void *stream);

/* ------------------------------------------------------------------------ problem construction on the device */
/* ProblemPreprocessingUtils.canonical_form_to_standard_form + add_variables_bounds_to_coefficient_matrix +
 * preprocess_problem (preprocessing.py:6-69), and the artificial block of PhaseOneSimplexSolver (special_solvers.py:48-52),
 * as ONE kernel that writes the standard-form matrix into HBM instead of vstack / hstack chains on the host:
 * rows = me equalities (A x = b), mi inequalities (G x <= h, one slack column each), nb bound rows (variable bound_var[t],
 * slack sign bound_sign[t] = -1 for x - s = lb / +1 for x + s = ub, right-hand side bound_val[t]; lower bounds first);
 * columns = nx structural + mi + nb slack + na artificial (na = 0 or rows).  Rows with a negative right-hand side are
 * negated (:11-13) except in the artificial block, which the reference appends afterwards.  A is me x nx, G is mi x nx,
 * out is rows x ld (float32), out_b the normalised right-hand side. */
int lpx_build_standard_form(int64_t me, int64_t mi, int64_t nx, int64_t nb, int64_t na, const float *A, const float *b,
                            const float *G, const float *h, const int64_t *bound_var, const float *bound_sign,
                            const float *bound_val, float *out, int64_t ld, float *out_b, void *stream);

/* PrimalDualAlgorithm.solve (special_solvers.py:402-447) as ONE launch for a small LP: s is the restricted-primal state
 * on [A | I] (n = n_orig + m columns, costs 0 / 1); per outer iteration the admissible set (:405-406), a fresh restricted
 * primal from the artificial basis (:408-425), and while its cost is positive y_rp, the unboundedness test (:435-436),
 * theta (:438-443) and y += theta y_rp (:444).  ctrl->status: LPX_OPTIMAL, LPX_RUNNING (maxiters1 outer iterations done),
 * LPX_DUAL_UNBOUNDED or LPX_PRIMAL_UNBOUNDED; ctrl->counter = outer iterations of this call, ctrl->pivots += inner pivots;
 * x (n) / cost[0] = the last restricted primal's solution.  Call lpx_ctrl_begin_solve first.
 * lpx_pd_small_lds_bytes(m, n_orig) <= 158 KiB says whether it fits, else LPX_E_SIZE. */
int64_t lpx_pd_small_lds_bytes(int64_t m, int64_t n_orig);
int lpx_pd_solve_small(const lpx_revised *s, double *y, const double *c64, int64_t n_orig, int64_t maxiters1,
                       int64_t maxiters2, double *x, double *cost, void *stream);

/* ----------------------------------------------------------------------------- tableau-form solver loop */
int64_t lpx_tableau_workspace_bytes(int64_t m, int64_t n);
/* `iters` primal iterations on an unpartitioned tableau (same rules as LPX_KIND_PRIMAL: primal_solvers.py:135-167,
 * :176-263).  Large tableaus run with DEFERRED updates (lpx_tableau_blocked.hpp): the decisions of up to
 * lpx_tableau_block_size() pivots are taken from the stale tableau plus the pending (eta column, pivot row) pairs, then
 * one pass over T applies those rank-1 updates to every element in order -- the arithmetic of :247-253 element by
 * element, bit for bit, with 1/block of the HBM traffic.  T holds the current state whenever a call returns. */
int lpx_tableau_iterate(const lpx_tableau *s, int64_t iters, void *stream);
/* Pivots per pass over the tableau that lpx_tableau_iterate uses for this state: 0 (one update per pivot) when
 * 8 (m+1)(n+1) bytes < LPX_TABLEAU_BLOCK_MIN_MB MiB (default 48); else LPX_TABLEAU_BLOCK if it is set to 8, 16 or 32
 * (any other value: 0); else 32 from 192 MiB on and 16 below.  No counterpart in the reference (it recomputes from B^-1
 * every iteration); a scheduling parameter, not a numerical one. */
int64_t lpx_tableau_block_size(const lpx_tableau *s);
/* 1 when lpx_tableau_iterate runs this state's deferred-update chain PIPELINED: blocks are in use, T_alt is set, the
 * tableau is unpartitioned and 8 (m+1)(n+1) bytes >= LPX_TABLEAU_PIPELINE_MIN_MB MiB (default 1024; negative: never).  The pass of block k
 * then writes the other buffer on the caller's stream while block k+1 is decided, on a stream of the library's own, from
 * the buffer the pass reads plus the pending pairs of both blocks: a block costs max(pass, decisions) instead of their
 * sum.  Same arithmetic element by element (primal_solvers.py:247-253), same bits; a scheduling parameter. */
int lpx_tableau_pipelined(const lpx_tableau *s);
/* Which launch chain lpx_tableau_iterate runs for this state (all take the same pivots and leave the same bits; a
 * scheduling matter, no counterpart in the reference):
 *   LPX_CHAIN_EAGER      column -> prepare -> K1, one rank-1 update of T per pivot, in place
 *   LPX_CHAIN_FUSED      one launch per pivot on two alternating buffers (T_alt set, up to 16384 rows)
 *   LPX_CHAIN_SMALL      one launch per block of 8 pivots: one workgroup decides the block while the others apply the
 *                        previous one (T_alt set, m + 1 <= 1024 and n + 1 <= 1024: launch-bound tableaus, BASELINE config 2)
 *   LPX_CHAIN_DEFERRED   deferred updates, decisions then one pass per block, in place (see lpx_tableau_block_size)
 *   LPX_CHAIN_PIPELINED  deferred updates, the pass beside the next block's decisions (see lpx_tableau_pipelined) */
#define LPX_CHAIN_EAGER 0
#define LPX_CHAIN_FUSED 1
#define LPX_CHAIN_SMALL 2
#define LPX_CHAIN_DEFERRED 3
#define LPX_CHAIN_PIPELINED 4
int lpx_tableau_chain_kind(const lpx_tableau *s);
/* Column-partitioned step, part 1: local pricing over owned columns and candidate pivot column.
 * sendbuf[0] = local min rc, sendbuf[1] = global index of the first owned column with rc<0 (or 2^62),
 * sendbuf[2 .. 2+m] = that column of T (m entries + its reduced cost).  Length lpx_tableau_msg_len(m). */
int64_t lpx_tableau_msg_len(int64_t m);
/* rescan != 0: scan the reduced-cost row first (first pivot after construction / after the caller touched T);
 * rescan == 0: use the pricing partials that the previous pivot's update kernel left behind. */
int lpx_tableau_propose(const lpx_tableau *s, double *sendbuf, int rescan, void *stream);
/* Part 2, after an all-gather of the proposals (recvbuf = nranks consecutive messages): pick the global
 * entering column (lowest index), run the ratio test on the replicated x_B column, update the local slice. */
int lpx_tableau_commit(const lpx_tableau *s, const double *recvbuf, int64_t nranks, void *stream);
/* The same step split in two so that the all-gather of pivot t+1 can overlap the update of pivot t (same reference
 * lines as lpx_tableau_commit: primal_solvers.py:135-167, :247-253).  commit_begin = entering/leaving decision and eta
 * column of this pivot and, when sendbuf_next != NULL, this rank's proposal for the NEXT pivot, formed from the old
 * tableau, e and the old pivot row exactly as the update will write them; commit_update = the rank-1 update.  The host
 * starts an asynchronous all-gather of sendbuf_next between the two calls. */
int lpx_tableau_commit_begin(const lpx_tableau *s, const double *recvbuf, int64_t nranks, double *sendbuf_next,
                             void *stream);
int lpx_tableau_commit_update(const lpx_tableau *s, void *stream);
/* The column-partitioned pivot with DEFERRED updates (see lpx_tableau_iterate): each rank defers the rank-1 updates of
 * its own slice.  Same exchange as lpx_tableau_propose / _commit (same message layout, primal_solvers.py:135-167), but
 * the proposal's column and the commit's pivot row are formed from the stale slice plus the `pending` (eta column,
 * pivot row) pairs of the current block, and no update kernel runs until lpx_tableau_block_flush applies the block
 * (`block` = 8, 16 or 32: at most that many pivots between two flushes; flush before anything else reads T.  Every rank
 * must run with the SAME block -- ranks that flush at different pivots stop exchanging at different pivots -- so the
 * caller agrees on one value first: the largest lpx_tableau_block_size() over the ranks' ragged slices, TableauEngine.kb).
 *   lpx_tableau_block_begin                    once before the first propose of a run of pivots (after the caller touched T)
 *   lpx_tableau_block_propose(pending, send)   pending = pivots since the last flush (0 .. block-1; 32 and more: LPX_E_ARG)
 *   all-gather of the proposals                (caller: RCCL / torch.distributed)
 *   lpx_tableau_block_commit(pending, recv)    entering / leaving decision, eta column, bookkeeping: pending + 1 are now pending
 *   lpx_tableau_block_flush(block)             T <- T with the pending updates applied, in order */
int lpx_tableau_block_begin(const lpx_tableau *s, void *stream);
int lpx_tableau_block_propose(const lpx_tableau *s, int64_t pending, double *sendbuf, void *stream);
int lpx_tableau_block_commit(const lpx_tableau *s, int64_t pending, const double *recvbuf, int64_t nranks, void *stream);
int lpx_tableau_block_flush(const lpx_tableau *s, int64_t block, void *stream);
/* The whole column-partitioned loop as ONE call per chunk of pivots, the collective included: liblpx links RCCL and
 * enqueues, for each of `iters` pivots, propose -> ncclAllGather of the ranks' proposals (lpx_tableau_msg_len(m) doubles
 * each, over xGMI) -> commit, plus the passes over this rank's slice, all on `stream` (slices for which
 * lpx_tableau_pipelined() holds: the passes and the next block's pivots side by side on streams of the library's own, the
 * collective with the pivots) -- nothing returns to the host between two pivots, the host reads the control block once per
 * call as for the unpartitioned loop.  Same decisions and bits as the lpx_tableau_block_* sequence above
 * (primal_solvers.py:135-167, :247-253); no counterpart in the reference, which is single-process.
 *   comm     an RCCL communicator over the ranks that hold the slices: lpx_dist_unique_id on one rank, its 128 bytes
 *            handed to the others by any means, lpx_dist_comm_create on every rank (or a caller's own ncclComm_t)
 *   block    8, 16 or 32, THE SAME ON EVERY RANK (see lpx_tableau_block_*)
 *   sendbuf  lpx_tableau_msg_len(m) doubles, recvbuf nranks times that (device memory, the caller's)
 * lpx_tableau_dual_dist_iterate: the dual rules (dual_solvers.py:47-108), passes in place. */
int lpx_dist_unique_id(void *id128);
int lpx_dist_comm_create(const void *id128, int64_t nranks, int64_t rank, void **comm);
int lpx_dist_comm_destroy(void *comm);
int lpx_tableau_dist_iterate(const lpx_tableau *s, void *comm, int64_t nranks, int64_t iters, int64_t block, double *sendbuf,
                             double *recvbuf, void *stream);
int lpx_tableau_dual_dist_iterate(const lpx_tableau *s, void *comm, int64_t nranks, int64_t iters, int64_t block,
                                  double *sendbuf, double *recvbuf, void *stream);
/* The same for the dual rules (dual_solvers.py:47-108; message layout of lpx_tableau_dual_propose): every rank picks the
 * leaving row from its replica of x_B, proposes (theta, variable, column as the pending updates leave it); the commit takes
 * the first argmin of theta over the ranks.  Flush with lpx_tableau_block_flush. */
int lpx_tableau_dual_block_begin(const lpx_tableau *s, void *stream);
int lpx_tableau_dual_block_propose(const lpx_tableau *s, int64_t pending, double *sendbuf, void *stream);
int lpx_tableau_dual_block_commit(const lpx_tableau *s, int64_t pending, const double *recvbuf, int64_t nranks, void *stream);
/* x[n_global-sized on rank owning..] : x_B and basis are replicated, so every rank can expand: x, cost. */
int lpx_tableau_expand(const lpx_tableau *s, const float *c_global, double *x, double *cost, void *stream);

/* Dual simplex on the same tableau: DualNaiveSimplexSolver.solve with DualRevisedSimplexSolver's pivot
 * (dual_solvers.py:47-120) -- leaving row = first x_B < 0 (:47-54), dual row = row r of T with basic columns forced
 * to 0 (:38-39), ratio test (-rc_j)/row_j over row_j < 0, first argmin (:61-80), LPX_DUAL_UNBOUNDED when
 * min(row) >= 0 (:41-44).  Same state struct, same K1; `pricing` is ignored. */
int lpx_tableau_dual_iterate(const lpx_tableau *s, int64_t iters, void *stream);
/* Column-partitioned dual pivot.  Part 1: every rank picks the leaving row from its x_B replica and proposes its
 * best owned column: sendbuf = [theta, global column, T[0..m, column]] (lpx_tableau_msg_len doubles). */
int lpx_tableau_dual_propose(const lpx_tableau *s, double *sendbuf, void *stream);
/* Part 2, after an all-gather of the proposals: global first argmin of theta, update of the local slice. */
int lpx_tableau_dual_commit(const lpx_tableau *s, const double *recvbuf, int64_t nranks, void *stream);
/* The same pivot split in two so that the all-gather of pivot t+1 overlaps the update of pivot t (same reference lines:
 * dual_solvers.py:47-80, primal_solvers.py:247-253).  dual_commit_begin = entering decision and eta column of this pivot
 * and, when sendbuf_next != NULL, the NEXT pivot's leaving row (first negative entry of the x_B column as the update will
 * leave it) and this rank's proposal for it, formed from the old tableau, e and the old pivot row exactly as the update
 * will write them.  adopt != 0: this pivot's leaving row is the one the previous call decided ahead (it could not go
 * into the control block while that pivot's update was still to read it); adopt == 0: lpx_tableau_dual_propose decided
 * it.  The update itself is lpx_tableau_commit_update. */
int lpx_tableau_dual_commit_begin(const lpx_tableau *s, const double *recvbuf, int64_t nranks, int adopt,
                                  double *sendbuf_next, void *stream);

/* ------------------------------------------------------- opt-in: fp64 refinement of the product-form inverse */
/* No counterpart in the reference's revised solvers, which multiply eta matrices for the whole solve
 * (primal_solvers.py:247-249); its only reinversion is the naive solvers' np.linalg.inv per pivot (:92-94).
 * `steps` Newton-Schulz steps X <- X + X (I - B X), B = A[:, basis], on the matrix cores (fp64 MFMA GEMM).
 * resid: 2 device doubles, max|I - B X| before and after.  update_xB != 0 also recomputes x_B = B^-1 b in fp64.
 * Clears the float32 phase.  Never called by solve(): callers opt in. */
int64_t lpx_refine_workspace_bytes(int64_t m);
int lpx_revised_refine(const lpx_revised *s, int64_t steps, int update_xB, void *work, int64_t work_bytes,
                       double *resid, void *stream);
/* The GEMM behind it: D = alpha * op(A) B + beta * C + gamma * I, fp64 row-major, op(A) = A^T when transa (A is then
 * stored K x M).  C may be NULL.  v_mfma_f64_16x16x4_f64, LDS-tiled. */
int lpx_dgemm(int transa, int64_t M, int64_t N, int64_t K, double alpha, const double *A, int64_t lda, const double *B,
              int64_t ldb, double beta, const double *C, int64_t ldc, double gamma, double *D, int64_t ldd, void *stream);

/* Opt-in reinversion in fp64.  Reference analogue: np.linalg.inv(A[:, basis]),
 * primal_solvers.py:92-94 (the naive solvers' per-pivot reinversion; the revised solvers never re-invert).
 * Blocked Gauss-Jordan with partial pivoting, 64 columns per block; the m x m x 64 trailing updates are fp64 MFMA
 * GEMMs and carry all 2 m^3 flop.  info: one device int64 -- 0, or 1 + the first column whose pivot candidates are
 * all below m * eps * max|A| (numpy raises LinAlgError("Singular matrix") for exact zeros; the output is then not an
 * inverse).
 *   lpx_dinverse:             Ainv = A^-1, row-major m x m, Ainv may be A
 *   lpx_revised_reinvert_f64: s->Binv = inv(A[:, basis]) from scratch, x_B = B^-1 b when update_xB != 0; clears the
 *                             float32 phase.  Never called by solve(). */
int64_t lpx_invert_workspace_bytes(int64_t m);
int lpx_dinverse(int64_t m, const double *A, int64_t lda, double *Ainv, int64_t ldi, void *work, int64_t work_bytes,
                 int64_t *info, void *stream);
int lpx_revised_reinvert_f64(const lpx_revised *s, int update_xB, void *work, int64_t work_bytes, int64_t *info,
                             void *stream);

/* ------------------------------------------------------------------------------------ batched small LPs */
/* PrimalRevisedSimplexSolver(c, A, b, basis).solve(maxiters) (primal_solvers.py:176-263) for `batch` independent
 * LPs of one shape, slack or general start basis (Binv0 = B^-1 per LP, or NULL for the identity).  Arrays are
 * batch-major.  m <= 64, n <= 256: one wave per LP (A in registers, B^-1 in LDS, four LPs per CU at m = 64);
 * larger shapes up to m, n <= 256 and 158 KB of LDS: one 256-lane workgroup per LP; beyond that LPX_E_SIZE.
 * Both kernels accumulate every dot product in index order with fma and return identical bits.
 * Environment: LPX_BATCH_KERNEL = wave | generic pins the kernel (tests). */
int lpx_batched_primal_solve(int64_t batch, int64_t m, int64_t n, const float *A, const float *b, const float *c,
                             const float *Binv0, int64_t *basis, double *x, double *cost, int32_t *status,
                             int64_t *iters, int64_t *pivots, int64_t maxiters, int64_t *log, int64_t log_cap,
                             void *stream);

/* DualRevisedSimplexSolver(c, A, b, basis).solve(maxiters) (dual_solvers.py:82-120) per LP: same arguments; the
 * starting bases must be dual feasible (dual_solvers.py:14-16, caller's check).  status: LPX_OPTIMAL, LPX_RUNNING
 * (maxiters hit) or LPX_DUAL_UNBOUNDED.  One wave per LP: m <= 64, n <= 256 (LPX_E_SIZE otherwise). */
int lpx_batched_dual_solve(int64_t batch, int64_t m, int64_t n, const float *A, const float *b, const float *c,
                           const float *Binv0, int64_t *basis, double *x, double *cost, int32_t *status,
                           int64_t *iters, int64_t *pivots, int64_t maxiters, int64_t *log, int64_t log_cap,
                           void *stream);
/* Inequality form: min c.x s.t. G x <= h, x >= 0 with h >= 0 (caller's check), i.e. the standard form [G | I] that
 * preprocessing.py:17-69 builds on the host, started from the slack basis.  The identity block and the zero slack costs
 * never exist in HBM: the kernels generate them.  G is batch x m x k, c batch x k, x batch x (k+m) (slacks included),
 * basis batch x m (output only).  Same kernels, same bits as lpx_batched_primal_solve on the materialised problem. */
int lpx_batched_primal_solve_ineq(int64_t batch, int64_t m, int64_t k, const float *G, const float *h, const float *c,
                                  int64_t *basis, double *x, double *cost, int32_t *status, int64_t *iters, int64_t *pivots,
                                  int64_t maxiters, int64_t *log, int64_t log_cap, void *stream);
/* The same solver semantics per LP with the CONDENSED TABLEAU as carried state (nonbasic columns of B^-1 A, reduced-cost
 * row, x_B; lpx_tableau's layout) instead of B^-1: one rank-1 update per pivot, the tableau in registers, one wave per LP.
 * Arithmetic of the tableau loop (bit-identical to the oracle's lpo_tableau_solve), the reference's pivot sequences.
 * One wave per LP for m <= 64, n - m <= 64; beyond, a grid of up to eight waves per LP (row groups x column groups, at
 * most 64 rows per lane): m <= 128 with n - m <= 256, or m <= 256 with n - m <= 128 (lpx_batched_tableau_waves != 0).
 * The starting basis is the slack block: columns n-m .. n-1 are the identity, their costs are 0, b >= 0 (caller's
 * check; no Binv0).  basis is output only. */
int lpx_batched_tableau_solve(int64_t batch, int64_t m, int64_t n, const float *A, const float *b, const float *c,
                              int64_t *basis, double *x, double *cost, int32_t *status, int64_t *iters, int64_t *pivots,
                              int64_t maxiters, int64_t *log, int64_t log_cap, void *stream);
/* The same for min c.x s.t. G x <= h, x >= 0 (preprocessing.py:17-69's [G | I] with its slack basis): G is m x k, c has k
 * entries, h >= 0, same shape limits with k for n - m; the identity block is never stored.  Results carry all k + m
 * variables, slacks last. */
int lpx_batched_tableau_solve_ineq(int64_t batch, int64_t m, int64_t k, const float *G, const float *h, const float *c,
                                   int64_t *basis, double *x, double *cost, int32_t *status, int64_t *iters, int64_t *pivots,
                                   int64_t maxiters, int64_t *log, int64_t log_cap, void *stream);
/* Waves per LP that lpx_batched_tableau_solve / _ineq use for m rows and k = n - m nonbasic columns; 0: the shape does not
 * fit (the caller takes lpx_batched_primal_solve).  Host-only, no device needed.  No counterpart in the reference. */
int lpx_batched_tableau_waves(int64_t m, int64_t k);
/* lpx_batched_tableau_solve from ANY primal feasible starting basis (primal_solvers.py:37-53).  The caller forms the
 * starting tableau with the reference's float32 expressions and passes it: T0 = (inv(A_B) A)[:, nonbasic] (m x k per LP,
 * k = n - m), rc0 = its reduced costs (k), xb0 = inv(A_B) b (m), var0 = the variable of every column (k), basis (m, input
 * and output); c (n per LP) is used for the cost of the result only.  Shapes as lpx_batched_tableau_solve. */
int lpx_batched_tableau_solve_from(int64_t batch, int64_t m, int64_t n, const float *T0, const float *rc0, const float *xb0,
                                   const int64_t *var0, const float *c, int64_t *basis, double *x, double *cost,
                                   int32_t *status, int64_t *iters, int64_t *pivots, int64_t maxiters, int64_t *log,
                                   int64_t log_cap, void *stream);
/* DualRevisedSimplexSolver per LP (dual_solvers.py:82-120) with the same carried state.  A, b as preprocessing.py:6-15 leaves
 * them (rows with b < 0 flipped): the slack block then has +-1 on its diagonal and is the starting basis (B^-1 = B), its
 * costs are 0 and c >= 0 elsewhere (dual feasible, :14-16) -- the caller's checks.  Shapes as lpx_batched_tableau_solve. */
int lpx_batched_tableau_dual_solve(int64_t batch, int64_t m, int64_t n, const float *A, const float *b, const float *c,
                                   int64_t *basis, double *x, double *cost, int32_t *status, int64_t *iters, int64_t *pivots,
                                   int64_t maxiters, int64_t *log, int64_t log_cap, void *stream);

/* ------------------------------------------------- individual kernels (unit-tested against the oracle) */
/* Rank-1 pivot update M[i,:] += e_i * M_old[r,:] (i != r), M[r,:] = e_r * M_old[r,:]:
 * _update_inv_basis_matrix / _update_bfs, primal_solvers.py:247-253.  rowr_scratch: >= ld doubles. */
int lpx_pivot_update(double *M, int64_t rows, int64_t cols, int64_t ld, int64_t r, const double *e,
                     double *rowr_scratch, void *stream);
/* lpx_pivot_update for callers that cannot allocate.  NO COUNTERPART IN THE REFERENCE: it exists for code that may not
 * call hipMalloc -- code under HIP graph capture, tight serving loops, harnesses that forbid allocation -- and so cannot
 * bring a rowr_scratch.  The copy of the old pivot row lives in a file-scope device array inside liblpx
 * (LPX_STATIC_ROW_MAX = 65536 doubles = 512 KiB per GPU, held until the process ends); the call allocates nothing, on
 * the device or through any library.  Same kernels and the same bits as lpx_pivot_update.  ld > 65536 returns
 * LPX_E_SIZE (use lpx_pivot_update then).  All calls on one GPU share the array: calls that overlap on one GPU (two
 * streams, two host threads) race on it -- serialise them on one stream.  The array used is the copy on the GPU that owns
 * M (not the current device); the FIRST call per GPU looks its address up (hipGetSymbolAddress) and is not safe inside
 * stream capture: make one call per GPU outside capture first. */
#define LPX_STATIC_ROW_MAX 65536
int lpx_pivot_update_static(double *M, int64_t rows, int64_t cols, int64_t ld, int64_t r, const double *e, void *stream);
/* Eta column, primal_solvers.py:235-244: e_i = (-d_i)/d_r, e_r = 1/d_r. */
int lpx_eta(const double *d, int64_t m, int64_t r, double *e, void *stream);
/* primal_solvers.py:135-142: out_val[0] = min(rc), out_idx[0] = first j with rc_j<0 (or -1).
 * var_state may be NULL; EXCLUDED columns are skipped.  scratch: lpx_reduce_scratch_bytes() bytes. */
int64_t lpx_reduce_scratch_bytes(void);
int lpx_price_first_negative(const double *rc, const int32_t *var_state, int64_t n, double *out_val,
                             int64_t *out_idx, void *scratch, void *stream);
/* special_solvers.py:208-214: max(rc) and first argmax. */
int lpx_price_argmax(const double *rc, int64_t n, double *out_val, int64_t *out_idx, void *scratch, void *stream);
/* primal_solvers.py:149-167 + utils.py:5-10; out_val = {theta_min, max(d)}, out_idx = first argmin. */
int lpx_ratio_primal(const double *x, const double *d, int64_t m, int f32, double *out_val, int64_t *out_idx,
                     void *scratch, void *stream);
/* dual_solvers.py:61-80 + utils.py:12-17; out_val = {theta_min, min(row)}, out_idx = first argmin. */
int lpx_ratio_dual(const double *rc, const double *row, const int32_t *var_state, int64_t n, int f32,
                   double *out_val, int64_t *out_idx, void *scratch, void *stream);
/* special_solvers.py:239-329; out_val = {min g1, min g2}, out_idx = {argmin g1, argmin g2}. */
int lpx_ratio_bounded(const double *x, const double *d, const double *lbB, const double *ubB, int64_t m,
                      double sign, double *out_val, int64_t *out_idx, void *scratch, void *stream);
/* dual_solvers.py:47-54: min(x_B), first i with x_i<0. */
int lpx_dual_row_select(const double *xB, int64_t m, double *out_val, int64_t *out_idx, void *scratch,
                        void *stream);

#ifdef __cplusplus
}
#endif
#endif /* LPX_H */
