// Diagnostic build of the batched kernel with per-phase cycle stamps (never shipped; see lpx_batched.hip).
//   hipcc -O3 -std=c++17 --offload-arch=gfx950 -ffp-contract=off -Iinclude -Ilinear-programming_amd/csrc \
//         tools/batch_stamps.hip -o tools/batch_stamps && ./tools/batch_stamps
#define LPX_BATCH_STAMPS 1
#ifdef ONE
#define LPX_WAVE_ONE_PER_SIMD 1
#endif
#include <stdarg.h>
#include <stdio.h>
#include <vector>
void lpx_set_error(const char *fmt, ...) { va_list ap; va_start(ap, fmt); vfprintf(stderr, fmt, ap); va_end(ap); }
#include "../linear-programming_amd/csrc/lpx_batched.hip"

int main()
{
    const int64_t B = 4096, m = 64, n = 128;
    std::vector<float> A(B * m * n, 0.f), b(B * m), c(B * n, 0.f);
    std::vector<int64_t> basis(B * m);
    unsigned long long s = 88172645463325252ULL;
    auto rnd = [&]() { s ^= s << 13; s ^= s >> 7; s ^= s << 17; return (float)((s >> 11) * (1.0 / 9007199254740992.0)); };
    for (int64_t l = 0; l < B; ++l) {
        for (int64_t i = 0; i < m; ++i) {
            for (int64_t j = 0; j < n - m; ++j) A[(l * m + i) * n + j] = rnd();
            A[(l * m + i) * n + (n - m) + i] = 1.f;
            b[l * m + i] = 1.f + rnd();
            basis[l * m + i] = n - m + i;
        }
        for (int64_t j = 0; j < n - m; ++j) c[l * n + j] = -rnd();
    }
    float *dA, *db, *dc; int64_t *dbasis, *diters, *dpiv; double *dx, *dcost; int32_t *dstatus;
    hipMalloc(&dA, A.size() * 4); hipMalloc(&db, b.size() * 4); hipMalloc(&dc, c.size() * 4);
    hipMalloc(&dbasis, basis.size() * 8); hipMalloc(&diters, B * 8); hipMalloc(&dpiv, B * 8);
    hipMalloc(&dx, B * n * 8); hipMalloc(&dcost, B * 8); hipMalloc(&dstatus, B * 4);
    hipMemcpy(dA, A.data(), A.size() * 4, hipMemcpyHostToDevice); hipMemcpy(db, b.data(), b.size() * 4, hipMemcpyHostToDevice);
    hipMemcpy(dc, c.data(), c.size() * 4, hipMemcpyHostToDevice);
    for (int rep = 0; rep < 2; ++rep) {
        hipMemcpy(dbasis, basis.data(), basis.size() * 8, hipMemcpyHostToDevice);
        unsigned long long zero[16] = {0};
        hipMemcpyToSymbol(HIP_SYMBOL(g_batch_stamps), zero, sizeof zero);
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        hipEventRecord(e0, 0);
        int rc = lpx_batched_primal_solve(B, m, n, dA, db, dc, nullptr, dbasis, dx, dcost, dstatus, diters, dpiv, 100000, nullptr, 0, nullptr);
        hipEventRecord(e1, 0); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        unsigned long long st[16]; hipMemcpyFromSymbol(st, HIP_SYMBOL(g_batch_stamps), sizeof st);
        std::vector<int64_t> piv(B); hipMemcpy(piv.data(), dpiv, B * 8, hipMemcpyDeviceToHost);
        long long tp = 0; for (auto v : piv) tp += v;
        double tot = 0; for (int q = 0; q < 8; ++q) tot += (double)st[q];
        const char *names[8] = {"y dot", "barrier", "rc dot", "reduce(min,first-neg)", "d dot + ratio", "reduce(argmin,max)", "eta + barrier", "rank-1 update + bookkeeping"};
        printf("rc=%d  %.3f ms  pivots=%lld  cycles/pivot=%.0f\n", rc, ms, tp, tot / tp);
        for (int q = 0; q < 8; ++q) printf("  %-30s %6.1f %%  %8.0f cycles/pivot\n", names[q], 100.0 * st[q] / tot, (double)st[q] / tp);
        printf("  wave-0 SIMD histogram: %llu %llu %llu %llu ; wave slot (low 2 bits): %llu %llu %llu %llu\n", st[8], st[9], st[10], st[11], st[12], st[13], st[14], st[15]);
    }
    return 0;
}
