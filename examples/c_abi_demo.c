/* The C ABI of liblpx.so used from plain C with the HIP runtime only (no Python, no PyTorch): a 2 x 4 LP through
 * PrimalRevisedSimplexSolver's loop (lpx_revised_iterate) and 4 LPs through the batched entry point.
 *
 *   gcc -std=c11 examples/c_abi_demo.c -Iinclude -I/opt/rocm/include -D__HIP_PLATFORM_AMD__ \
 *       -Llinear-programming_amd/csrc -llpx -L/opt/rocm/lib -lamdhip64 -Wl,-rpath,$PWD/linear-programming_amd/csrc \
 *       -Wl,-rpath,/opt/rocm/lib -o examples/c_abi_demo && ./examples/c_abi_demo
 *
 * LP (reference tests/problems.py style):  min -3 x0 - 5 x1  s.t.  x0 + s0 = 4,  2 x1 + s1 = 12  -> x = (4, 6), cost -42. */
#include <hip/hip_runtime_api.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "lpx.h"

#define CHECK(x)                                                                 \
    do {                                                                         \
        if ((x) != 0) {                                                          \
            fprintf(stderr, "%s failed: %s\n", #x, lpx_last_error());            \
            return 1;                                                            \
        }                                                                        \
    } while (0)

static void *dev(const void *host, size_t bytes)
{
    void *p = NULL;
    if (hipMalloc(&p, bytes) != hipSuccess) exit(2);
    if (host) hipMemcpy(p, host, bytes, hipMemcpyHostToDevice);
    else hipMemset(p, 0, bytes);
    return p;
}

int main(void)
{
    enum { M = 2, N = 4 };
    const float A[M][N] = {{1, 0, 1, 0}, {0, 2, 0, 1}}, b[M] = {4, 12}, c[N] = {-3, -5, 0, 0};
    float At[N][M];
    for (int i = 0; i < M; ++i)
        for (int j = 0; j < N; ++j) At[j][i] = A[i][j];
    const double Binv[M][M] = {{1, 0}, {0, 1}}, xB[M] = {4, 12};
    const int64_t basis[M] = {2, 3};
    const int32_t vs[N] = {LPX_VS_LOWER, LPX_VS_LOWER, LPX_VS_BASIC, LPX_VS_BASIC};

    lpx_revised s;
    memset(&s, 0, sizeof s);
    s.m = M; s.n = N; s.ldA = N; s.ldAt = 4; s.ldB = M;
    float Atp[N][4] = {{0}};
    for (int j = 0; j < N; ++j)
        for (int i = 0; i < M; ++i) Atp[j][i] = At[j][i];
    s.A = dev(A, sizeof A); s.At = dev(Atp, sizeof Atp); s.b = dev(b, sizeof b); s.c = dev(c, sizeof c);
    s.Binv = dev(Binv, sizeof Binv); s.xB = dev(xB, sizeof xB); s.basis = dev(basis, sizeof basis);
    s.var_state = dev(vs, sizeof vs);
    s.ctrl = dev(NULL, 128);
    s.workspace_bytes = lpx_revised_workspace_bytes(M, N);
    s.workspace = dev(NULL, (size_t)s.workspace_bytes);
    double *x = dev(NULL, sizeof(double) * N), *cost = dev(NULL, sizeof(double));

    CHECK(lpx_ctrl_init(s.ctrl, 1, NULL));
    CHECK(lpx_ctrl_begin_solve(s.ctrl, NULL));
    CHECK(lpx_revised_iterate(&s, LPX_KIND_PRIMAL, 100, NULL));
    lpx_ctrl ctrl;
    CHECK(lpx_read_ctrl(s.ctrl, &ctrl, NULL));
    CHECK(lpx_revised_expand(&s, x, cost, NULL));
    double hx[N], hcost;
    hipMemcpy(hx, x, sizeof hx, hipMemcpyDeviceToHost);
    hipMemcpy(&hcost, cost, sizeof hcost, hipMemcpyDeviceToHost);
    printf("revised: status %d iters %lld x = (%g, %g, %g, %g) cost %g\n", ctrl.status, (long long)ctrl.counter, hx[0], hx[1],
           hx[2], hx[3], hcost);
    if (ctrl.status != LPX_OPTIMAL || hx[0] != 4.0 || hx[1] != 6.0 || hcost != -42.0) return 3;

    /* the same LP four times through the batched inequality-form entry point: G = A[:, :2], h = b */
    enum { B = 4, K = 2 };
    float G[B][M][K], h[B][M], cg[B][K];
    for (int l = 0; l < B; ++l)
        for (int i = 0; i < M; ++i) {
            h[l][i] = b[i] * (float)(l + 1);
            for (int j = 0; j < K; ++j) { G[l][i][j] = A[i][j]; cg[l][j] = c[j]; }
        }
    float *dG = dev(G, sizeof G), *dh = dev(h, sizeof h), *dc = dev(cg, sizeof cg);
    int64_t *dbasis = dev(NULL, sizeof(int64_t) * B * M), *diters = dev(NULL, 8 * B), *dpiv = dev(NULL, 8 * B);
    double *dx = dev(NULL, sizeof(double) * B * (K + M)), *dcost = dev(NULL, 8 * B);
    int32_t *dstatus = dev(NULL, 4 * B);
    CHECK(lpx_batched_primal_solve_ineq(B, M, K, dG, dh, dc, dbasis, dx, dcost, dstatus, diters, dpiv, 100, NULL, 0, NULL));
    double bc[B];
    int32_t bs[B];
    hipMemcpy(bc, dcost, sizeof bc, hipMemcpyDeviceToHost);
    hipMemcpy(bs, dstatus, sizeof bs, hipMemcpyDeviceToHost);
    for (int l = 0; l < B; ++l) {
        printf("batched LP %d: status %d cost %g\n", l, bs[l], bc[l]);
        if (bs[l] != LPX_OPTIMAL || bc[l] != -42.0 * (l + 1)) return 4;
    }
    printf("c abi demo ok\n");
    return 0;
}
