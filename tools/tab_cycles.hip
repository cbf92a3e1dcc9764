// Diagnostic build of the batched condensed-tableau kernels with a cycle counter per workgroup (never shipped):
// resident cycles per pivot of one LP's wave(s), at one LP per SIMD (latency of a pivot) and at full residency (what a
// pivot costs the SIMD when other LPs fill its stalls).
//   hipcc -O3 -std=c++17 --offload-arch=gfx950 -ffp-contract=off -Iinclude -Ilinear-programming_amd/csrc \
//         tools/tab_cycles.hip -o tools/tab_cycles && ./tools/tab_cycles
#define LPX_TAB_STAMPS 1
#include <stdarg.h>
#include <stdio.h>
#include <vector>
void lpx_set_error(const char *fmt, ...) { va_list ap; va_start(ap, fmt); vfprintf(stderr, fmt, ap); va_end(ap); }
#include "../linear-programming_amd/csrc/lpx_batched_tab.hip"

static void run(int64_t B, int64_t m, int64_t n)
{
    const int64_t k = n - m;
    std::vector<float> A(B * m * n, 0.f), b(B * m), c(B * n, 0.f);
    unsigned long long s = 88172645463325252ULL;
    auto rnd = [&]() { s ^= s << 13; s ^= s >> 7; s ^= s << 17; return (float)((s >> 11) * (1.0 / 9007199254740992.0)); };
    for (int64_t l = 0; l < B; ++l) {
        for (int64_t i = 0; i < m; ++i) {
            for (int64_t j = 0; j < k; ++j) A[(l * m + i) * n + j] = rnd();
            A[(l * m + i) * n + k + i] = 1.f;
            b[l * m + i] = 1.f + rnd();
        }
        for (int64_t j = 0; j < k; ++j) c[l * n + j] = -rnd();
    }
    float *dA, *db, *dc; int64_t *dbasis, *diters, *dpiv; double *dx, *dcost; int32_t *dstatus;
    hipMalloc(&dA, A.size() * 4); hipMalloc(&db, b.size() * 4); hipMalloc(&dc, c.size() * 4);
    hipMalloc(&dbasis, B * m * 8); hipMalloc(&diters, B * 8); hipMalloc(&dpiv, B * 8);
    hipMalloc(&dx, B * n * 8); hipMalloc(&dcost, B * 8); hipMalloc(&dstatus, B * 4);
    hipMemcpy(dA, A.data(), A.size() * 4, hipMemcpyHostToDevice); hipMemcpy(db, b.data(), b.size() * 4, hipMemcpyHostToDevice);
    hipMemcpy(dc, c.data(), c.size() * 4, hipMemcpyHostToDevice);
    for (int rep = 0; rep < 2; ++rep) {
        unsigned long long zero[2] = {0, 0};
        hipMemcpyToSymbol(HIP_SYMBOL(g_tab_stamps), zero, sizeof zero);
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        hipEventRecord(e0, 0);
        const int rc = lpx_batched_tableau_solve(B, m, n, dA, db, dc, dbasis, dx, dcost, dstatus, diters, dpiv, 100000, nullptr, 0, nullptr);
        hipEventRecord(e1, 0); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        unsigned long long st[2]; hipMemcpyFromSymbol(st, HIP_SYMBOL(g_tab_stamps), sizeof st);
        if (rep == 1)
            printf("%3lldx%-3lld x%-6lld waves/LP %d  rc=%d  %8.3f ms  %9.1f M pivots/s  pivots/LP %6.1f  resident cycles per pivot %7.0f\n",
                   (long long)m, (long long)n, (long long)B, lpx_batched_tableau_waves(m, k), rc, ms, st[1] / (ms * 1e3),
                   (double)st[1] / B, (double)st[0] / (double)st[1]);
    }
    hipFree(dA); hipFree(db); hipFree(dc); hipFree(dbasis); hipFree(diters); hipFree(dpiv); hipFree(dx); hipFree(dcost); hipFree(dstatus);
}

int main(int argc, char **argv)
{
    if (argc > 1) { // one shape at full residency: the run to put under rocprofv3 --pmc
        run(12288, 64, 128);
        return 0;
    }
    for (int64_t B : {1024, 3072, 12288}) run(B, 64, 128);
    for (int64_t B : {512, 1536, 6144}) run(B, 80, 170);
    for (int64_t B : {256, 768, 3072}) run(B, 64, 256);
    for (int64_t B : {128, 256, 1536}) run(B, 128, 256);
    return 0;
}
