/*
 * lpx_oracle.c -- CPU restatement of the dense simplex pivot path of agmuth/linear-programming.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing in the product package (linear-programming_amd/) may import,
 * link or execute this file; only tests/, __graft_entry__.smoke() and bench.py's `cpu_baseline`
 * leg use it, and there only as the checker / the CPU baseline.
 *
 * Parity pin: this restatement is checked against golden vectors generated in the build container by
 * importing the reference itself (tests/golden/make_golden.py -> tests/golden/ JSON files), which include
 * every problem of the reference's own tests/problems.py plus seeded random LPs.  See
 * tests/test_oracle_vs_golden.py.
 *
 * Each function cites the reference file:line (relative to /root/reference) it restates.
 *
 * Arithmetic conventions (fixed, shared with the HIP kernels where a kernel is elementwise):
 *   - dot products are accumulated sequentially in index order with fused multiply-add;
 *   - the rank-1 pivot update is one fma per element, row r is a plain multiply;
 *   - `f32_phase`: the reference keeps inv_basis_matrix / bfs in float32 until the first eta update
 *     (SURVEY.md Appendix B); while it is set, pricing, direction and the primal/dual ratio are
 *     evaluated in float32.
 * Build with -ffp-contract=off so that only the explicit fma()/fmaf() calls fuse.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#define LPO_RUNNING 0
#define LPO_OPTIMAL 1
#define LPO_PRIMAL_UNBOUNDED 2
#define LPO_DUAL_UNBOUNDED 3
#define LPO_NOT_NONBASIC 4

#define VS_BASIC 0
#define VS_LOWER 1 /* nonbasic (at lower bound) */
#define VS_UPPER 2 /* nonbasic at upper bound (bounded solver only) */
#define VS_EXCLUDED 3 /* column not part of the (restricted) problem */

#define ISCLOSE_ATOL 1e-8 /* numpy.isclose defaults used at primal_solvers.py:142 etc. */
#define ISCLOSE_RTOL 1e-5

/* ------------------------------------------------------------------------------------------------
 * Elementwise / reduction building blocks (each is also the checker of one HIP kernel)
 * ---------------------------------------------------------------------------------------------- */

/* primal_solvers.py:247-253 (E @ B^-1, E @ bfs) as a rank-1 update.
 * M is rows x cols, row-major with leading dimension ld.  rowr = old row r (copied by the caller or
 * here), e = eta column (e[r] = 1/d_r, e[i] = -d_i/d_r). */
void lpo_rank1_update(double *M, int64_t rows, int64_t cols, int64_t ld, int64_t r, const double *e,
                      double *rowr_scratch)
{
    memcpy(rowr_scratch, M + r * ld, (size_t)cols * sizeof(double));
    for (int64_t i = 0; i < rows; ++i) {
        double *Mi = M + i * ld;
        const double ei = e[i];
        if (i == r) {
            for (int64_t k = 0; k < cols; ++k) Mi[k] = ei * rowr_scratch[k];
        } else {
            for (int64_t k = 0; k < cols; ++k) Mi[k] = fma(ei, rowr_scratch[k], Mi[k]);
        }
    }
}

/* OpenMP variant of the same update (identical results: rows are independent). */
void lpo_rank1_update_mt(double *M, int64_t rows, int64_t cols, int64_t ld, int64_t r, const double *e,
                         double *rowr_scratch)
{
    memcpy(rowr_scratch, M + r * ld, (size_t)cols * sizeof(double));
#pragma omp parallel for schedule(static)
    for (int64_t i = 0; i < rows; ++i) {
        double *Mi = M + i * ld;
        const double ei = e[i];
        if (i == r) {
            for (int64_t k = 0; k < cols; ++k) Mi[k] = ei * rowr_scratch[k];
        } else {
            for (int64_t k = 0; k < cols; ++k) Mi[k] = fma(ei, rowr_scratch[k], Mi[k]);
        }
    }
}

/* primal_solvers.py:217-245: eta column from the direction d and the leaving row r.
 * E[:,r] = -d ; E[r,r] = 1 ; E[:,r] /= d[r]  =>  e_i = (-d_i)/d_r, e_r = 1/d_r (both divisions). */
void lpo_eta(const double *d, int64_t m, int64_t r, double *e)
{
    const double dr = d[r];
    for (int64_t i = 0; i < m; ++i) e[i] = (-d[i]) / dr;
    e[r] = 1.0 / dr;
}

/* primal_solvers.py:135-142: min(rc) and FIRST index with rc<0 (np.argmax(rc<0)); excluded columns
 * (restricted primal of the primal-dual algorithm) are skipped.  Returns 1 if optimal. */
int lpo_first_negative(const double *rc, const int32_t *var_state, int64_t n, double *min_out, int64_t *j_out)
{
    double mn = INFINITY;
    int64_t j = -1;
    for (int64_t k = 0; k < n; ++k) {
        if (var_state && var_state[k] == VS_EXCLUDED) continue;
        if (rc[k] < mn) mn = rc[k];
        if (j < 0 && rc[k] < 0) j = k;
    }
    *min_out = mn;
    *j_out = j;
    return (mn >= 0) || (fabs(mn) <= ISCLOSE_ATOL);
}

/* special_solvers.py:208-214: max(rc) and FIRST argmax (largest-coefficient rule). Returns 1 if optimal. */
int lpo_first_argmax(const double *rc, int64_t n, double *max_out, int64_t *j_out)
{
    double mx = -INFINITY;
    int64_t j = 0;
    for (int64_t k = 0; k < n; ++k)
        if (rc[k] > mx) { mx = rc[k]; j = k; }
    *max_out = mx;
    *j_out = j;
    return (mx <= 0) || (fabs(mx) <= ISCLOSE_ATOL);
}

/* primal_solvers.py:149-167 + utils.py:5-10: theta_i = x_i/d_i if d_i>0 else inf; FIRST argmin.
 * f32 != 0: the quotient is a float32 division (Appendix B, first iteration of a fresh solver).
 * Also returns max(d) for primal_solvers.py:144-147. */
void lpo_ratio_primal(const double *x, const double *d, int64_t m, int f32, int64_t *r_out, double *theta_out,
                      double *dmax_out)
{
    double best = INFINITY, dmax = -INFINITY;
    int64_t r = 0;
    for (int64_t i = 0; i < m; ++i) {
        double t = INFINITY;
        if (d[i] > 0) t = f32 ? (double)((float)x[i] / (float)d[i]) : x[i] / d[i];
        if (i == 0 || t < best) { best = t; r = i; }
        if (d[i] > dmax) dmax = d[i];
    }
    *r_out = r;
    *theta_out = best;
    *dmax_out = dmax;
}

/* Harris-style two-pass ratio test (the build's opt-in extension; the reference has only the textbook test above, so
 * this restates the BUILD's stated rule, parity unpinned against the reference):
 *   pass 1: theta_max = min over d_i > 0 of (x_i + tol) / d_i          (the step a tol-relaxed feasibility allows)
 *   pass 2: among rows with d_i > 0 and x_i / d_i <= theta_max, the one with the LARGEST pivot d_i, first on ties.
 * Quotients in float64 whatever the float32 phase.  theta_out = x_r / d_r. */
void lpo_ratio_harris(const double *x, const double *d, int64_t m, double tol, int64_t *r_out, double *theta_out,
                      double *dmax_out)
{
    double tmax = INFINITY, dmax = -INFINITY;
    for (int64_t i = 0; i < m; ++i) {
        if (d[i] > 0) {
            const double t = (x[i] + tol) / d[i];
            if (t < tmax) tmax = t;
        }
        if (d[i] > dmax) dmax = d[i];
    }
    int64_t r = 0;
    double best = -INFINITY;
    for (int64_t i = 0; i < m; ++i)
        if (d[i] > 0 && x[i] / d[i] <= tmax && d[i] > best) { best = d[i]; r = i; }
    *r_out = r;
    *theta_out = (dmax > 0) ? x[r] / d[r] : INFINITY;
    *dmax_out = dmax;
}

/* dual_solvers.py:61-80 + utils.py:12-17: theta_j = (-rc_j)/row_j if row_j<0 else inf; FIRST argmin.
 * Also returns min(row) for dual_solvers.py:56-59. */
void lpo_ratio_dual(const double *rc, const double *row, const int32_t *var_state, int64_t n, int f32,
                    int64_t *j_out, double *theta_out, double *rowmin_out)
{
    double best = INFINITY, rmin = INFINITY;
    int64_t j = 0;
    int first = 1;
    for (int64_t k = 0; k < n; ++k) {
        if (var_state && var_state[k] == VS_EXCLUDED) continue;
        double t = INFINITY;
        if (row[k] < 0) t = f32 ? (double)((-(float)rc[k]) / (float)row[k]) : (-rc[k]) / row[k];
        if (first || t < best) { best = t; j = k; first = 0; }
        if (row[k] < rmin) rmin = row[k];
    }
    *j_out = j;
    *theta_out = best;
    *rowmin_out = rmin;
}

/* dual_solvers.py:47-54: min(bfs) and FIRST i with bfs_i<0. Returns 1 if optimal. */
int lpo_dual_row_select(const double *x, int64_t m, double *min_out, int64_t *r_out)
{
    return lpo_first_negative(x, NULL, m, min_out, r_out);
}

/* ------------------------------------------------------------------------------------------------
 * Revised-simplex state (reference: explicit B^-1, primal_solvers.py:13-41)
 * ---------------------------------------------------------------------------------------------- */
typedef struct lpo_revised {
    int64_t m, n;
    const float *A;     /* m x n row-major, float32 (preprocessing.py:8) */
    const float *b;     /* m */
    const float *c;     /* n */
    double *Binv;       /* m x m row-major; float32-valued while f32_phase */
    double *xB;         /* m */
    int64_t *basis;     /* m */
    int32_t *var_state; /* n */
    int32_t f32_phase;
    int32_t status;
    int64_t counter;    /* reference self.counter (per solve() call) */
    int64_t pivots;     /* basis changes since construction */
    int64_t *log;       /* optional (row, entering, leaving var) triples */
    int64_t log_cap;
    /* scratch, caller allocated */
    double *y;          /* m */
    double *rc;         /* n */
    double *d;          /* m */
    double *e;          /* m */
    double *rowr;       /* max(m, n) */
    double *row;        /* n (dual row) */
    /* bounded-variable solver only */
    const double *lb;   /* n */
    const double *ub;   /* n */
    double *w;          /* m */
} lpo_revised;

/* primal_solvers.py:59-66: rc = c - (c_B @ B^-1) @ A, left to right, then rc[basis] = 0. */
void lpo_price(lpo_revised *s)
{
    const int64_t m = s->m, n = s->n;
    if (s->f32_phase) {
        float *yf = (float *)malloc(sizeof(float) * (size_t)(m + n));
        float *zf = yf + m;
        for (int64_t k = 0; k < m; ++k) yf[k] = 0.f;
        for (int64_t i = 0; i < m; ++i) {
            const float cb = s->c[s->basis[i]];
            const double *Bi = s->Binv + i * m;
            for (int64_t k = 0; k < m; ++k) yf[k] = fmaf(cb, (float)Bi[k], yf[k]);
        }
        for (int64_t j = 0; j < n; ++j) zf[j] = 0.f;
        for (int64_t i = 0; i < m; ++i) {
            const float yi = yf[i];
            const float *Ai = s->A + i * n;
            for (int64_t j = 0; j < n; ++j) zf[j] = fmaf(yi, Ai[j], zf[j]);
        }
        for (int64_t k = 0; k < m; ++k) s->y[k] = yf[k];
        for (int64_t j = 0; j < n; ++j) s->rc[j] = (double)(s->c[j] - zf[j]);
        free(yf);
    } else {
        double *z = s->rc;
        for (int64_t k = 0; k < m; ++k) s->y[k] = 0.0;
        for (int64_t i = 0; i < m; ++i) {
            const double cb = s->c[s->basis[i]];
            const double *Bi = s->Binv + i * m;
            for (int64_t k = 0; k < m; ++k) s->y[k] = fma(cb, Bi[k], s->y[k]);
        }
        for (int64_t j = 0; j < n; ++j) z[j] = 0.0;
        for (int64_t i = 0; i < m; ++i) {
            const double yi = s->y[i];
            const float *Ai = s->A + i * n;
            for (int64_t j = 0; j < n; ++j) z[j] = fma(yi, (double)Ai[j], z[j]);
        }
        for (int64_t j = 0; j < n; ++j) s->rc[j] = (double)s->c[j] - z[j];
    }
    for (int64_t i = 0; i < m; ++i) s->rc[s->basis[i]] = 0.0;
}

/* primal_solvers.py:129 / :234: d = B^-1 @ A[:, j]. */
void lpo_direction(lpo_revised *s, int64_t j)
{
    const int64_t m = s->m, n = s->n;
    if (s->f32_phase) {
        for (int64_t i = 0; i < m; ++i) {
            const double *Bi = s->Binv + i * m;
            float acc = 0.f;
            for (int64_t k = 0; k < m; ++k) acc = fmaf((float)Bi[k], s->A[k * n + j], acc);
            s->d[i] = acc;
        }
    } else {
        for (int64_t i = 0; i < m; ++i) {
            const double *Bi = s->Binv + i * m;
            double acc = 0.0;
            for (int64_t k = 0; k < m; ++k) acc = fma(Bi[k], (double)s->A[k * n + j], acc);
            s->d[i] = acc;
        }
    }
}

/* dual_solvers.py:37-40: row = B^-1[r, :] @ A ; row[basis] = 0. */
void lpo_dual_row(lpo_revised *s, int64_t r)
{
    const int64_t m = s->m, n = s->n;
    const double *Br = s->Binv + r * m;
    if (s->f32_phase) {
        float *zf = (float *)malloc(sizeof(float) * (size_t)n);
        for (int64_t j = 0; j < n; ++j) zf[j] = 0.f;
        for (int64_t i = 0; i < m; ++i) {
            const float yi = (float)Br[i];
            const float *Ai = s->A + i * n;
            for (int64_t j = 0; j < n; ++j) zf[j] = fmaf(yi, Ai[j], zf[j]);
        }
        for (int64_t j = 0; j < n; ++j) s->row[j] = zf[j];
        free(zf);
    } else {
        for (int64_t j = 0; j < n; ++j) s->row[j] = 0.0;
        for (int64_t i = 0; i < m; ++i) {
            const double yi = Br[i];
            const float *Ai = s->A + i * n;
            for (int64_t j = 0; j < n; ++j) s->row[j] = fma(yi, (double)Ai[j], s->row[j]);
        }
    }
    for (int64_t i = 0; i < m; ++i) s->row[s->basis[i]] = 0.0;
}

static void lpo_log(lpo_revised *s, int64_t r, int64_t j, int64_t leaving)
{
    if (s->log && s->pivots < s->log_cap) {
        s->log[3 * s->pivots + 0] = r;
        s->log[3 * s->pivots + 1] = j;
        s->log[3 * s->pivots + 2] = leaving;
    }
    s->pivots += 1;
}

/* primal_solvers.py:255-263: PrimalRevisedSimplexSolver.pivot(row, col). */
void lpo_revised_pivot(lpo_revised *s, int64_t r, int64_t j)
{
    const int64_t m = s->m;
    lpo_direction(s, j); /* :234 (recomputed; same value as :129) */
    lpo_eta(s->d, m, r, s->e);
    const int64_t leaving = s->basis[r];
    lpo_log(s, r, j, leaving);
    s->basis[r] = j; /* :261 */
    if (s->var_state) {
        s->var_state[leaving] = VS_LOWER;
        s->var_state[j] = VS_BASIC;
    }
    lpo_rank1_update(s->Binv, m, m, m, r, s->e, s->rowr); /* :262 */
    const double xr = s->xB[r];                           /* :263 */
    for (int64_t i = 0; i < m; ++i) s->xB[i] = (i == r) ? s->e[i] * xr : fma(s->e[i], xr, s->xB[i]);
    s->f32_phase = 0; /* E is float64 (np.eye), so everything is float64 from here on */
}

/* primal_solvers.py:176-209: PrimalNaiveSimplexSolver.solve (inherited by the revised solver).
 * Returns the status; s->counter is the reference's self.counter. */
int lpo_primal_solve(lpo_revised *s, int64_t maxiters)
{
    s->counter = 0;
    s->status = LPO_RUNNING;
    while (s->counter < maxiters) {
        s->counter += 1;
        lpo_price(s);
        double mn;
        int64_t j;
        if (lpo_first_negative(s->rc, s->var_state, s->n, &mn, &j)) {
            s->status = LPO_OPTIMAL;
            break;
        }
        lpo_direction(s, j);
        int64_t r;
        double theta, dmax;
        lpo_ratio_primal(s->xB, s->d, s->m, s->f32_phase, &r, &theta, &dmax);
        if (dmax <= 0) { /* :130-132, :144-147 */
            s->status = LPO_PRIMAL_UNBOUNDED;
            break;
        }
        lpo_revised_pivot(s, r, j);
    }
    return s->status;
}

/* dual_solvers.py:82-108: DualNaiveSimplexSolver.solve with the revised pivot (dual_solvers.py:116-120). */
int lpo_dual_solve(lpo_revised *s, int64_t maxiters)
{
    s->counter = 0;
    s->status = LPO_RUNNING;
    while (s->counter < maxiters) {
        s->counter += 1;
        double mn;
        int64_t r;
        if (lpo_dual_row_select(s->xB, s->m, &mn, &r)) {
            s->status = LPO_OPTIMAL;
            break;
        }
        lpo_dual_row(s, r);
        lpo_price(s); /* dual_solvers.py:77 */
        int64_t j;
        double theta, rowmin;
        lpo_ratio_dual(s->rc, s->row, s->var_state, s->n, s->f32_phase, &j, &theta, &rowmin);
        if (rowmin >= 0) { /* dual_solvers.py:41-44 (raised before the ratio test) */
            s->status = LPO_DUAL_UNBOUNDED;
            break;
        }
        lpo_revised_pivot(s, r, j);
    }
    return s->status;
}

/* ------------------------------------------------------------------------------------------------
 * Bounded-variable primal simplex (special_solvers.py:94-343)
 * ---------------------------------------------------------------------------------------------- */

/* special_solvers.py:169-175: x_B = B^-1 (b - A_lb lb - A_ub ub), recomputed from scratch. */
void lpo_bounded_update_bfs(lpo_revised *s)
{
    const int64_t m = s->m, n = s->n;
    for (int64_t i = 0; i < m; ++i) {
        const float *Ai = s->A + i * n;
        double accl = 0.0, accu = 0.0;
        for (int64_t j = 0; j < n; ++j) {
            if (s->var_state[j] == VS_LOWER) accl = fma((double)Ai[j], s->lb[j], accl);
            else if (s->var_state[j] == VS_UPPER) accu = fma((double)Ai[j], s->ub[j], accu);
        }
        s->w[i] = ((double)s->b[i] - accl) - accu;
    }
    for (int64_t i = 0; i < m; ++i) {
        const double *Bi = s->Binv + i * m;
        double acc = 0.0;
        for (int64_t k = 0; k < m; ++k) acc = fma(Bi[k], s->w[k], acc);
        s->xB[i] = acc;
    }
}

/* special_solvers.py:177-199: bounded reduced costs; the result array is float32. */
void lpo_bounded_price(lpo_revised *s)
{
    const int64_t m = s->m, n = s->n;
    if (s->f32_phase) {
        float *yf = (float *)malloc(sizeof(float) * (size_t)(m + n));
        float *zf = yf + m;
        for (int64_t k = 0; k < m; ++k) yf[k] = 0.f;
        for (int64_t i = 0; i < m; ++i) {
            const float cb = s->c[s->basis[i]];
            const double *Bi = s->Binv + i * m;
            for (int64_t k = 0; k < m; ++k) yf[k] = fmaf(cb, (float)Bi[k], yf[k]);
        }
        for (int64_t j = 0; j < n; ++j) zf[j] = 0.f;
        for (int64_t i = 0; i < m; ++i) {
            const float yi = yf[i];
            const float *Ai = s->A + i * n;
            for (int64_t j = 0; j < n; ++j) zf[j] = fmaf(yi, Ai[j], zf[j]);
        }
        for (int64_t j = 0; j < n; ++j) {
            const float t = (-s->c[j]) + zf[j];
            s->rc[j] = (s->var_state[j] == VS_UPPER) ? (double)(-t) : (double)t;
        }
        free(yf);
    } else {
        double *z = s->rc;
        for (int64_t k = 0; k < m; ++k) s->y[k] = 0.0;
        for (int64_t i = 0; i < m; ++i) {
            const double cb = s->c[s->basis[i]];
            const double *Bi = s->Binv + i * m;
            for (int64_t k = 0; k < m; ++k) s->y[k] = fma(cb, Bi[k], s->y[k]);
        }
        for (int64_t j = 0; j < n; ++j) z[j] = 0.0;
        for (int64_t i = 0; i < m; ++i) {
            const double yi = s->y[i];
            const float *Ai = s->A + i * n;
            for (int64_t j = 0; j < n; ++j) z[j] = fma(yi, (double)Ai[j], z[j]);
        }
        for (int64_t j = 0; j < n; ++j) {
            const float t = (float)((double)(-s->c[j]) + z[j]); /* float32 array += float64 */
            s->rc[j] = (s->var_state[j] == VS_UPPER) ? (double)(-t) : (double)t;
        }
    }
    for (int64_t i = 0; i < m; ++i) s->rc[s->basis[i]] = 0.0;
}

/* special_solvers.py:239-329: three-way ratio test.
 * sign=+1 if the entering variable sits at its lower bound, -1 if at its upper bound.
 * Outputs min/argmin of gammas_1 and gammas_2. */
void lpo_ratio_bounded(const double *x, const double *d, const double *lbB, const double *ubB, int64_t m,
                       double sign, double *g1min, int64_t *r1, double *g2min, int64_t *r2)
{
    double b1 = INFINITY, b2 = INFINITY;
    int64_t i1 = 0, i2 = 0;
    for (int64_t i = 0; i < m; ++i) {
        const double d1 = sign * d[i], d2 = -sign * d[i];
        const double t1 = (d1 > 0) ? (x[i] - lbB[i]) / d1 : INFINITY;
        const double t2 = (d2 > 0) ? (ubB[i] - x[i]) / d2 : INFINITY;
        if (i == 0 || t1 < b1) { b1 = t1; i1 = i; }
        if (i == 0 || t2 < b2) { b2 = t2; i2 = i; }
    }
    *g1min = b1; *r1 = i1; *g2min = b2; *r2 = i2;
}

/* special_solvers.py:221-343: BoundedVariablePrimalSimplexSolver.solve.
 * The log receives one triple per iteration that changes the state:
 *   basis change: (row, entering, leaving var); bound flip: (-1, entering, -1).
 * After each leaving variable the triple's 3rd entry is followed in var_state by its bound. */
int lpo_bounded_solve(lpo_revised *s, int64_t maxiters)
{
    const int64_t m = s->m;
    double *lbB = (double *)malloc(sizeof(double) * 2 * (size_t)m);
    double *ubB = lbB + m;
    s->counter = 0;
    s->status = LPO_RUNNING;
    while (s->counter < maxiters) {
        s->counter += 1;
        lpo_bounded_price(s);
        double mx;
        int64_t j;
        if (lpo_first_argmax(s->rc, s->n, &mx, &j)) {
            s->status = LPO_OPTIMAL;
            break;
        }
        double sign;
        if (s->var_state[j] == VS_LOWER) sign = 1.0;
        else if (s->var_state[j] == VS_UPPER) sign = -1.0;
        else { s->status = LPO_NOT_NONBASIC; break; } /* :331 */
        lpo_direction(s, j);
        for (int64_t i = 0; i < m; ++i) { lbB[i] = s->lb[s->basis[i]]; ubB[i] = s->ub[s->basis[i]]; }
        double g1, g2;
        int64_t r1, r2;
        lpo_ratio_bounded(s->xB, s->d, lbB, ubB, m, sign, &g1, &r1, &g2, &r2);
        const double g3 = s->ub[j] - s->lb[j];
        const double delta = g1 < g2 ? g1 : g2;
        if (delta == INFINITY) { s->status = LPO_PRIMAL_UNBOUNDED; break; } /* :256-257 / :302-303 */
        if (g3 <= delta) {
            s->var_state[j] = (sign > 0) ? VS_UPPER : VS_LOWER; /* bound flip :264-268 / :309-313 */
            lpo_log(s, -1, j, -1);
        } else {
            int64_t r;
            int32_t leave_state;
            if (g1 < g2) { r = r1; leave_state = VS_LOWER; } /* :271-277 */
            else { r = r2; leave_state = VS_UPPER; }         /* :278-284 */
            const int64_t leaving = s->basis[r];
            lpo_eta(s->d, m, r, s->e); /* :335-339, E from the OLD inverse */
            lpo_log(s, r, j, leaving);
            s->basis[r] = j;
            s->var_state[leaving] = leave_state;
            s->var_state[j] = VS_BASIC;
            lpo_rank1_update(s->Binv, m, m, m, r, s->e, s->rowr); /* :340 */
            s->f32_phase = 0;
        }
        lpo_bounded_update_bfs(s); /* :341 */
    }
    free(lbB);
    return s->status;
}

/* ------------------------------------------------------------------------------------------------
 * Bounded-variable DUAL simplex -- the build's opt-in extension for BASELINE config 5's wording ("dual simplex with
 * bounded variables").  The reference has no such solver (its bounded solver is the primal one above, its dual solver
 * knows x >= 0 only: special_solvers.py:94-343, dual_solvers.py:47-108), so this restates the BUILD's stated rules and is
 * parity unpinned against the reference.  Rules, the reference's dual rules generalised to lb <= x <= ub, all float64:
 *   leaving row   = FIRST row whose basic variable violates a bound (x_i < lb or x_i > ub); optimal when the worst
 *                   violation is <= 1e-8 (dual_solvers.py:47-54 with "x_i < 0" read as "outside its bounds")
 *   dual row      = B^-1[r, :] @ A, reduced costs c - c_B B^-1 A (dual_solvers.py:37-40, :77), basic columns 0
 *   ratio test    = alpha_j = row_j when x_r is BELOW its bound, -row_j when above; candidates: nonbasic at lower with
 *                   alpha_j < 0, nonbasic at upper with alpha_j > 0; theta_j = (-rc_j) / alpha_j; FIRST argmin
 *                   (dual_solvers.py:61-80, utils.py:12-17); no candidate = the primal is infeasible (:41-44)
 *   pivot         = eta column from d = B^-1 A_j (primal_solvers.py:217-249); the leaving variable goes to the bound it
 *                   violated; x_B = B^-1 (b - A_lb lb - A_ub ub) from scratch as the bounded primal does (:169-175)
 * ---------------------------------------------------------------------------------------------- */
int lpo_dual_bounded_solve(lpo_revised *s, int64_t maxiters)
{
    const int64_t m = s->m, n = s->n;
    s->counter = 0;
    s->status = LPO_RUNNING;
    s->f32_phase = 0;
    lpo_bounded_update_bfs(s);
    while (s->counter < maxiters) {
        s->counter += 1;
        double worst = -INFINITY;
        int64_t r = -1;
        int below = 0;
        for (int64_t i = 0; i < m; ++i) {
            const double lo = s->lb[s->basis[i]], hi = s->ub[s->basis[i]], x = s->xB[i];
            const double inf = (lo - x > x - hi) ? lo - x : x - hi;
            if (inf > worst) worst = inf;
            if (r < 0 && inf > 0) { r = i; below = x < lo; }
        }
        if (worst <= ISCLOSE_ATOL) {
            s->status = LPO_OPTIMAL;
            break;
        }
        lpo_dual_row(s, r);
        lpo_price(s);
        double best = INFINITY;
        int64_t j = -1;
        for (int64_t k = 0; k < n; ++k) {
            const int32_t vs = s->var_state[k];
            if (vs != VS_LOWER && vs != VS_UPPER) continue;
            const double alpha = below ? s->row[k] : -s->row[k];
            if ((vs == VS_LOWER && alpha < 0) || (vs == VS_UPPER && alpha > 0)) {
                const double t = (-s->rc[k]) / alpha;
                if (j < 0 || t < best) { best = t; j = k; }
            }
        }
        if (j < 0) {
            s->status = LPO_DUAL_UNBOUNDED;
            break;
        }
        lpo_direction(s, j);
        lpo_eta(s->d, m, r, s->e);
        const int64_t leaving = s->basis[r];
        lpo_log(s, r, j, leaving);
        s->basis[r] = j;
        s->var_state[leaving] = below ? VS_LOWER : VS_UPPER;
        s->var_state[j] = VS_BASIC;
        lpo_rank1_update(s->Binv, m, m, m, r, s->e, s->rowr);
        lpo_bounded_update_bfs(s);
    }
    return s->status;
}

/* ------------------------------------------------------------------------------------------------
 * Primal-dual algorithm pieces (special_solvers.py:402-447); the outer loop is driven from Python
 * (oracle/oracle.py) because it rebuilds the restricted primal every outer iteration.
 * ---------------------------------------------------------------------------------------------- */

/* special_solvers.py:405: v = y @ A (float64); admissible_j = isclose(v_j, c_j). A has ldA columns. */
void lpo_pd_admissible(const double *y, const float *A, const double *c, int64_t m, int64_t n, int64_t ldA,
                       double *v, int32_t *adm)
{
    for (int64_t j = 0; j < n; ++j) v[j] = 0.0;
    for (int64_t i = 0; i < m; ++i) {
        const float *Ai = A + i * ldA;
        for (int64_t j = 0; j < n; ++j) v[j] = fma(y[i], (double)Ai[j], v[j]);
    }
    for (int64_t j = 0; j < n; ++j)
        adm[j] = fabs(v[j] - c[j]) <= ISCLOSE_ATOL + ISCLOSE_RTOL * fabs(c[j]);
}

/* special_solvers.py:438-444: theta = min over inadmissible j of psd(c_j - v_j, u_j), u = y_rp @ A;
 * then y += theta * y_rp.  Also reports max(u) over ALL columns of A (for :435-436 the artificial block
 * contributes y_rp itself; handled by the caller). */
void lpo_pd_theta(const double *y_rp, const float *A, const double *c, const double *v, const int32_t *adm,
                  int64_t m, int64_t n, int64_t ldA, double *u, double *theta_out, double *umax_out)
{
    for (int64_t j = 0; j < n; ++j) u[j] = 0.0;
    for (int64_t i = 0; i < m; ++i) {
        const float *Ai = A + i * ldA;
        for (int64_t j = 0; j < n; ++j) u[j] = fma(y_rp[i], (double)Ai[j], u[j]);
    }
    double theta = INFINITY, umax = -INFINITY;
    for (int64_t j = 0; j < n; ++j) {
        if (u[j] > umax) umax = u[j];
        if (adm[j]) continue;
        const double t = (u[j] > 0) ? (c[j] - v[j]) / u[j] : INFINITY;
        if (t < theta) theta = t;
    }
    *theta_out = theta;
    *umax_out = umax;
}

/* ------------------------------------------------------------------------------------------------
 * Tableau form (build's own carrying of T = B^-1 A, x_B and the reduced-cost row; SURVEY.md Appendix A
 * "Tableau equivalent").  The reference forms B^-1 A only at special_solvers.py:69; the pivot rules are
 * those of primal_solvers.py:135-167 applied to the carried row/column.
 *
 * Layout: T is (m+1) x ld row-major, ld >= n+1; column n holds x_B (and -objective in row m), row m holds
 * the reduced costs.
 * ---------------------------------------------------------------------------------------------- */
typedef struct lpo_tableau {
    int64_t m, n, ld;
    double *T;
    int64_t *basis;
    int32_t *var_state;
    int32_t f32_phase;
    int32_t status;
    int64_t counter;
    int64_t pivots;
    int64_t *log;
    int64_t log_cap;
    double *d;    /* m+1 */
    double *e;    /* m+1 */
    double *rowr; /* ld */
    int32_t threads; /* >1: OpenMP update */
    int32_t pricing; /* 0: first negative reduced cost (the reference's rule); 1: Dantzig, most negative, first index
                        on ties -- the build's opt-in extension, no counterpart in the reference */
    int32_t ratio_rule; /* 0: the reference's textbook ratio test (primal_solvers.py:149-167); 1: Harris-style two-pass
                           test with tolerance ratio_tol -- opt-in extension, no counterpart in the reference */
    int32_t pad_;
    double ratio_tol;
} lpo_tableau;

int lpo_tableau_solve(lpo_tableau *s, int64_t maxiters)
{
    const int64_t m = s->m, n = s->n, ld = s->ld;
    double *xcol = (double *)malloc(sizeof(double) * (size_t)m);
    s->counter = 0;
    s->status = LPO_RUNNING;
    while (s->counter < maxiters) {
        s->counter += 1;
        double mn;
        int64_t j;
        const int opt = lpo_first_negative(s->T + m * ld, s->var_state, n, &mn, &j); /* primal_solvers.py:135-142 */
        if (opt) {
            s->status = LPO_OPTIMAL;
            break;
        }
        if (s->pricing == 1) { /* opt-in Dantzig rule: same optimality test, entering = first argmin */
            const double *rc = s->T + m * ld;
            j = 0;
            for (int64_t k = 1; k < n; ++k)
                if (rc[k] < rc[j]) j = k;
        }
        for (int64_t i = 0; i <= m; ++i) s->d[i] = s->T[i * ld + j]; /* d = (B^-1 A)[:, j], d[m] = rc_j */
        for (int64_t i = 0; i < m; ++i) xcol[i] = s->T[i * ld + n];
        int64_t r;
        double theta, dmax;
        if (s->ratio_rule == 1) lpo_ratio_harris(xcol, s->d, m, s->ratio_tol, &r, &theta, &dmax);
        else lpo_ratio_primal(xcol, s->d, m, s->f32_phase, &r, &theta, &dmax); /* :149-167 */
        if (dmax <= 0) { s->status = LPO_PRIMAL_UNBOUNDED; break; }
        const double dr = s->d[r];
        for (int64_t i = 0; i <= m; ++i) s->e[i] = (-s->d[i]) / dr; /* :236-244 */
        s->e[r] = 1.0 / dr;
        const int64_t leaving = s->basis[r];
        if (s->log && s->pivots < s->log_cap) {
            s->log[3 * s->pivots + 0] = r;
            s->log[3 * s->pivots + 1] = j;
            s->log[3 * s->pivots + 2] = leaving;
        }
        s->pivots += 1;
        s->basis[r] = j;
        s->var_state[leaving] = VS_LOWER;
        s->var_state[j] = VS_BASIC;
        if (s->threads > 1) lpo_rank1_update_mt(s->T, m + 1, n + 1, ld, r, s->e, s->rowr);
        else lpo_rank1_update(s->T, m + 1, n + 1, ld, r, s->e, s->rowr);
        /* the entering column becomes the unit vector exactly; its reduced cost exactly 0
         * (primal_solvers.py:65 forces basic reduced costs to 0) */
        for (int64_t i = 0; i <= m; ++i) s->T[i * ld + j] = (i == r) ? 1.0 : 0.0;
        s->f32_phase = 0;
    }
    free(xcol);
    return s->status;
}

/* The same loop on the CONDENSED layout (the build's default for the carried tableau): T holds only the
 * NONBASIC columns -- a basic column of a tableau is a unit vector whose pivot-row entry is 0, so the rank-1 update never
 * changes it.  s->n counts column slots (column n is still x_B), slot_var[k] is the variable in slot k, var_state is
 * unused.  Rules as in lpo_tableau_solve (primal_solvers.py:135-167): "first index with rc < 0" is the smallest
 * VARIABLE id over the slots with a negative reduced cost; the optimality test reads the same values, since the reference
 * forces basic reduced costs to exactly 0 (:65).  On a pivot the entering variable's slot is handed to the leaving
 * variable, whose column after the update is the eta column itself (unit vector e_r updated: 0 + e_i * 1). */
int lpo_tableau_solve_condensed(lpo_tableau *s, int64_t *slot_var, int64_t maxiters)
{
    const int64_t m = s->m, n = s->n, ld = s->ld;
    double *xcol = (double *)malloc(sizeof(double) * (size_t)m);
    s->counter = 0;
    s->status = LPO_RUNNING;
    while (s->counter < maxiters) {
        s->counter += 1;
        const double *rc = s->T + m * ld;
        double mn = INFINITY;
        int64_t k = -1; /* slot of the entering variable */
        for (int64_t q = 0; q < n; ++q) {
            if (rc[q] < mn) mn = rc[q];
            if (s->pricing == 1) { /* opt-in Dantzig rule: most negative, smallest variable id on ties */
                if (k < 0 || rc[q] < rc[k] || (rc[q] == rc[k] && slot_var[q] < slot_var[k])) k = q;
            } else if (rc[q] < 0.0 && (k < 0 || slot_var[q] < slot_var[k])) k = q; /* :135-138 */
        }
        if (mn >= 0.0 || fabs(mn) <= ISCLOSE_ATOL) { /* :140-142 */
            s->status = LPO_OPTIMAL;
            break;
        }
        const int64_t j = slot_var[k];
        for (int64_t i = 0; i <= m; ++i) s->d[i] = s->T[i * ld + k]; /* d = (B^-1 A)[:, j], d[m] = rc_j */
        for (int64_t i = 0; i < m; ++i) xcol[i] = s->T[i * ld + n];
        int64_t r;
        double theta, dmax;
        if (s->ratio_rule == 1) lpo_ratio_harris(xcol, s->d, m, s->ratio_tol, &r, &theta, &dmax);
        else lpo_ratio_primal(xcol, s->d, m, s->f32_phase, &r, &theta, &dmax); /* :149-167 */
        if (dmax <= 0) { s->status = LPO_PRIMAL_UNBOUNDED; break; }
        const double dr = s->d[r];
        for (int64_t i = 0; i <= m; ++i) s->e[i] = (-s->d[i]) / dr; /* :236-244 */
        s->e[r] = 1.0 / dr;
        const int64_t leaving = s->basis[r];
        if (s->log && s->pivots < s->log_cap) {
            s->log[3 * s->pivots + 0] = r;
            s->log[3 * s->pivots + 1] = j;
            s->log[3 * s->pivots + 2] = leaving;
        }
        s->pivots += 1;
        s->basis[r] = j;
        if (s->threads > 1) lpo_rank1_update_mt(s->T, m + 1, n + 1, ld, r, s->e, s->rowr);
        else lpo_rank1_update(s->T, m + 1, n + 1, ld, r, s->e, s->rowr);
        for (int64_t i = 0; i <= m; ++i) s->T[i * ld + k] = s->e[i]; /* the leaving variable's column */
        slot_var[k] = leaving;
        s->f32_phase = 0;
    }
    free(xcol);
    return s->status;
}

/* Dual simplex on the carried tableau: DualNaiveSimplexSolver.solve (dual_solvers.py:82-108) with the rules of
 * :47-80 read off T -- leaving row = first x_B < 0 (column n), dual row = row r of T with basic columns forced to 0
 * (:39), reduced costs = row m, ratio test (-rc_j)/row_j over row_j < 0, first argmin; then the same pivot as
 * lpo_tableau_solve. */
int lpo_tableau_dual_solve(lpo_tableau *s, int64_t maxiters)
{
    const int64_t m = s->m, n = s->n, ld = s->ld;
    double *xcol = (double *)malloc(sizeof(double) * (size_t)m);
    double *row = (double *)malloc(sizeof(double) * (size_t)n);
    s->counter = 0;
    s->status = LPO_RUNNING;
    while (s->counter < maxiters) {
        s->counter += 1;
        for (int64_t i = 0; i < m; ++i) xcol[i] = s->T[i * ld + n];
        double mn;
        int64_t r;
        if (lpo_dual_row_select(xcol, m, &mn, &r)) { /* dual_solvers.py:47-54 */
            s->status = LPO_OPTIMAL;
            break;
        }
        for (int64_t k = 0; k < n; ++k) row[k] = (s->var_state[k] == VS_BASIC) ? 0.0 : s->T[r * ld + k]; /* :38-39 */
        int64_t j;
        double theta, rowmin;
        lpo_ratio_dual(s->T + m * ld, row, s->var_state, n, s->f32_phase, &j, &theta, &rowmin); /* :61-80 */
        if (rowmin >= 0) { /* :41-44 */
            s->status = LPO_DUAL_UNBOUNDED;
            break;
        }
        for (int64_t i = 0; i <= m; ++i) s->d[i] = s->T[i * ld + j];
        const double dr = s->d[r];
        for (int64_t i = 0; i <= m; ++i) s->e[i] = (-s->d[i]) / dr; /* primal_solvers.py:236-244 */
        s->e[r] = 1.0 / dr;
        const int64_t leaving = s->basis[r];
        if (s->log && s->pivots < s->log_cap) {
            s->log[3 * s->pivots + 0] = r;
            s->log[3 * s->pivots + 1] = j;
            s->log[3 * s->pivots + 2] = leaving;
        }
        s->pivots += 1;
        s->basis[r] = j;
        s->var_state[leaving] = VS_LOWER;
        s->var_state[j] = VS_BASIC;
        if (s->threads > 1) lpo_rank1_update_mt(s->T, m + 1, n + 1, ld, r, s->e, s->rowr);
        else lpo_rank1_update(s->T, m + 1, n + 1, ld, r, s->e, s->rowr);
        for (int64_t i = 0; i <= m; ++i) s->T[i * ld + j] = (i == r) ? 1.0 : 0.0;
        s->f32_phase = 0;
    }
    free(xcol);
    free(row);
    return s->status;
}

/* primal_solvers.py:72-79: x = scatter(x_B at basis); cost = dot(c, x) (float64 accumulate, index order). */
double lpo_expand_and_cost(const double *xB, const int64_t *basis, const float *c, int64_t m, int64_t n,
                           double *x)
{
    for (int64_t j = 0; j < n; ++j) x[j] = 0.0;
    for (int64_t i = 0; i < m; ++i) x[basis[i]] = xB[i];
    double acc = 0.0;
    for (int64_t j = 0; j < n; ++j) acc = fma((double)c[j], x[j], acc);
    return acc;
}

/* A batch of independent LPs of one shape through lpo_primal_solve (PrimalRevisedSimplexSolver.solve,
 * primal_solvers.py:176-263), one LP per OpenMP thread: the CPU side of SURVEY.md 8(d) config 3.  Slack or general
 * start: Binv0 = float32 B^-1 per LP or NULL for the identity; x_B = B^-1 b in float32 (:38), accumulated in index
 * order.  Outputs per LP: basis (in place), x (n), cost, status, iters (reference counter), pivots. */
static void lpo_solve_batch(int dual, int64_t batch, int64_t m, int64_t n, const float *A, const float *b, const float *c,
                            const float *Binv0, int64_t *basis, double *x, double *cost, int32_t *status,
                            int64_t *iters, int64_t *pivots, int64_t maxiters, int threads)
{
#pragma omp parallel num_threads(threads > 0 ? threads : 1)
    {
        const int64_t mx = m > n ? m : n;
        double *buf = (double *)malloc(sizeof(double) * (size_t)(m * m + 5 * m + 2 * n + mx));
        int32_t *vs = (int32_t *)malloc(sizeof(int32_t) * (size_t)n);
#pragma omp for schedule(dynamic, 1)
        for (int64_t l = 0; l < batch; ++l) {
            lpo_revised s;
            memset(&s, 0, sizeof s);
            double *p = buf;
            s.m = m; s.n = n;
            s.A = A + l * m * n; s.b = b + l * m; s.c = c + l * n;
            s.Binv = p; p += m * m;
            s.xB = p; p += m;
            s.y = p; p += m;
            s.d = p; p += m;
            s.e = p; p += m;
            s.w = p; p += m;
            s.rc = p; p += n;
            s.row = p; p += n;
            s.rowr = p;
            s.basis = basis + l * m;
            s.var_state = vs;
            for (int64_t i = 0; i < m; ++i)
                for (int64_t k = 0; k < m; ++k)
                    s.Binv[i * m + k] = Binv0 ? (double)Binv0[(l * m + i) * m + k] : (i == k ? 1.0 : 0.0);
            for (int64_t i = 0; i < m; ++i) {
                float acc = 0.f;
                for (int64_t k = 0; k < m; ++k) acc = fmaf((float)s.Binv[i * m + k], s.b[k], acc);
                s.xB[i] = (double)acc;
            }
            for (int64_t j = 0; j < n; ++j) vs[j] = VS_LOWER;
            for (int64_t i = 0; i < m; ++i) vs[s.basis[i]] = VS_BASIC;
            s.f32_phase = 1;
            status[l] = dual ? lpo_dual_solve(&s, maxiters) : lpo_primal_solve(&s, maxiters);
            iters[l] = s.counter;
            pivots[l] = s.pivots;
            cost[l] = lpo_expand_and_cost(s.xB, s.basis, s.c, m, n, x + l * n);
        }
        free(buf);
        free(vs);
    }
}

void lpo_primal_solve_batch(int64_t batch, int64_t m, int64_t n, const float *A, const float *b, const float *c,
                            const float *Binv0, int64_t *basis, double *x, double *cost, int32_t *status,
                            int64_t *iters, int64_t *pivots, int64_t maxiters, int threads)
{
    lpo_solve_batch(0, batch, m, n, A, b, c, Binv0, basis, x, cost, status, iters, pivots, maxiters, threads);
}

/* The same through lpo_dual_solve (DualRevisedSimplexSolver.solve, dual_solvers.py:82-120). */
void lpo_dual_solve_batch(int64_t batch, int64_t m, int64_t n, const float *A, const float *b, const float *c,
                          const float *Binv0, int64_t *basis, double *x, double *cost, int32_t *status, int64_t *iters,
                          int64_t *pivots, int64_t maxiters, int threads)
{
    lpo_solve_batch(1, batch, m, n, A, b, c, Binv0, basis, x, cost, status, iters, pivots, maxiters, threads);
}

int lpo_version(void) { return 1; }
