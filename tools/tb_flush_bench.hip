// tb_flush_bench.hip -- k_tb_flush2 (deferred-update pass, lpx_tb_flush.hpp) on a synthetic tableau: block sizes, rows per
// wave group and band heights against a one-thread-per-element reference, timed with HIP events.   usage: tb_flush_bench rows cols [cnt]
// build: hipcc -O3 -std=c++17 --offload-arch=gfx950 -ffp-contract=off -I include -I linear-programming_amd/csrc
//        tools/tb_flush_bench.hip -o tools/tb_flush_bench
#include <stdio.h>
#include <stdlib.h>
#include <vector>

#include "lpx_tb_flush.hpp"

void lpx_set_error(const char *, ...) {}
LpxK1Profile &lpx_k1_profile()
{
    static LpxK1Profile p;
    return p;
}

#define CK(x)                                                                        \
    do {                                                                             \
        hipError_t e_ = (x);                                                         \
        if (e_ != hipSuccess) {                                                      \
            printf("%s: %s\n", #x, hipGetErrorString(e_));                           \
            exit(1);                                                                 \
        }                                                                            \
    } while (0)

__global__ void k_ref(TbFlushArgs a, double *out)
{
    const int64_t col = (int64_t)blockIdx.x * 256 + threadIdx.x, i = blockIdx.y;
    if (col >= a.ld) return;
    double x = a.T[i * a.ld + col];
    if (col < a.cols + (a.cols & 1)) {
        const int cnt = (int)*a.cnt;
        const uint32_t rf = a.rowflag[i];
        for (int u = 0; u < cnt; ++u)
            x = tb_apply(x, col == a.pslot[u], (rf >> u) & 1u, a.E[i * TB_KMAX + u], a.R[u * a.ldr + col], a.cond != 0);
    }
    out[i * a.ld + col] = x;
}
__global__ void k_diff(const double *a, const double *b, int64_t n, int64_t ld, int64_t cols, unsigned long long *bad)
{
    int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    unsigned long long c = 0;
    for (; i < n; i += (int64_t)gridDim.x * 256)
        if (i % ld < cols && __double_as_longlong(a[i]) != __double_as_longlong(b[i])) ++c;
    if (c) atomicAdd(bad, c);
}
__global__ void k_fill(double *p, int64_t n, unsigned seed, double scale)
{
    int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    for (; i < n; i += (int64_t)gridDim.x * 256) {
        unsigned long long h = (unsigned long long)i * 6364136223846793005ULL + seed * 1442695040888963407ULL;
        h ^= h >> 29;
        h *= 0xbf58476d1ce4e5b9ULL;
        h ^= h >> 32;
        p[i] = scale * ((double)(h & 0xffffff) / 8388608.0 - 1.0);
    }
}

struct Variant {
    const char *name;
    void (*launch)(TbFlushArgs, hipStream_t);
};
template <int KB, int NT, int RG, int THREADS, int CV>
static void go2(TbFlushArgs a, hipStream_t st)
{
    // a.rpb = rows per workgroup (band height)
    hipLaunchKernelGGL((k_tb_flush2<KB, NT, RG, THREADS, CV>), dim3((unsigned)lpx_cdiv(a.cols, TB2_COLS * CV), (unsigned)lpx_cdiv(a.rows, a.rpb)),
                       dim3(THREADS), 0, st, a);
}

int main(int argc, char **argv)
{
    const int64_t rows = argc > 1 ? atoll(argv[1]) : 20001, cols = argc > 2 ? atoll(argv[2]) : 20001;
    const int cnt = argc > 3 ? atoi(argv[3]) : 32;
    const int64_t ld = lpx_roundup(cols, cols >= 4096 ? 512 : 16), ldr = lpx_roundup(cols, 2) + 2;
    const int64_t total = rows * ld;
    double *T, *T0, *Tref, *E, *R;
    int64_t *pslot, *dcnt;
    uint32_t *rowflag;
    unsigned long long *bad;
    CK(hipMalloc(&T, 8 * total));
    CK(hipMalloc(&T0, 8 * total));
    CK(hipMalloc(&Tref, 8 * total));
    CK(hipMalloc(&E, 8 * (rows + 2) * TB_KMAX));
    CK(hipMalloc(&R, 8 * ldr * TB_KMAX));
    CK(hipMalloc(&pslot, 8 * TB_KMAX));
    CK(hipMalloc(&dcnt, 16));
    CK(hipMalloc(&rowflag, 4 * (rows + 2)));
    CK(hipMalloc(&bad, 8));
    k_fill<<<4096, 256>>>(T0, total, 1, 1.0);
    k_fill<<<1024, 256>>>(E, (rows + 2) * TB_KMAX, 2, 1e-2);
    k_fill<<<1024, 256>>>(R, ldr * TB_KMAX, 3, 1.0);
    std::vector<uint32_t> rf(rows + 2, 0u);
    std::vector<int64_t> ps(TB_KMAX, -1);
    for (int u = 0; u < cnt; ++u) {
        rf[(int64_t)(u * 977 + 13) % rows] |= 1u << u;
        ps[u] = (int64_t)(u * 613 + 7) % (cols - 1);
    }
    if (cnt > 3) { // a row that is the pivot row of two updates, a slot rewritten twice
        rf[13 % rows] |= 1u << 3;
        ps[3] = ps[0];
    }
    const int64_t c64 = cnt;
    CK(hipMemcpy(rowflag, rf.data(), 4 * (rows + 2), hipMemcpyHostToDevice));
    CK(hipMemcpy(pslot, ps.data(), 8 * TB_KMAX, hipMemcpyHostToDevice));
    CK(hipMemcpy(dcnt, &c64, 8, hipMemcpyHostToDevice));
    TbFlushArgs a{};
    a.T = T0; a.rows = rows; a.cols = cols; a.ld = ld; a.cond = 1; a.E = E; a.R = R; a.ldr = ldr;
    a.pslot = pslot; a.rowflag = rowflag; a.cnt = dcnt; a.rpb = 16;
    k_ref<<<dim3((unsigned)lpx_cdiv(ld, 256), (unsigned)rows), 256>>>(a, Tref);
    CK(hipDeviceSynchronize());
    a.T = T;
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    const double bytes = 16.0 * rows * cols;
    printf("rows %lld cols %lld ld %lld cnt %d: %.3f GB per pass\n", (long long)rows, (long long)cols, (long long)ld, cnt, bytes / 1e9);

#define V2(KB, NT, RG, TH, CV) Variant{"kb" #KB " nt" #NT " rg" #RG " threads" #TH " colvecs" #CV, go2<KB, NT, RG, TH, CV>}
    const Variant vs[] = {V2(32, 1, 8, 256, 1), V2(32, 1, 4, 256, 2), V2(32, 1, 2, 256, 4)};
    const int64_t rpbs[] = {1251, 640, 320, 160};
    for (const Variant &v : vs) {
        for (int64_t rpb : rpbs) {
            a.rpb = rpb;
            CK(hipMemcpy(T, T0, 8 * total, hipMemcpyDeviceToDevice));
            v.launch(a, 0);
            CK(hipMemset(bad, 0, 8));
            k_diff<<<4096, 256>>>(T, Tref, total, ld, cols, bad);
            unsigned long long hb = 0;
            CK(hipMemcpy(&hb, bad, 8, hipMemcpyDeviceToHost));
            float best = 1e30f;
            for (int rep = 0; rep < 4; ++rep) {
                CK(hipEventRecord(e0, 0));
                v.launch(a, 0);
                CK(hipEventRecord(e1, 0));
                CK(hipEventSynchronize(e1));
                float ms;
                CK(hipEventElapsedTime(&ms, e0, e1));
                if (ms < best) best = ms;
            }
            printf("%-32s rpb %3lld: %8.3f ms  %7.1f GB/s  (%.3f of 8 TB/s)%s\n", v.name, (long long)rpb, best, bytes / best / 1e6,
                   bytes / best / 1e6 / 8000.0, hb ? "  MISMATCH" : "");
            fflush(stdout);
        }
    }
    return 0;
}
