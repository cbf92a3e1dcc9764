// Experiment: K1's stream (condensed 20001 x 20001 tableau, rows padded to 4 KiB) as PERSISTENT workgroups that walk the
// tiles in address order with a grid stride (tile t, t + G, t + 2G ...: the chip-wide front still advances through memory
// in order) and issue the next tile's loads before the current tile's stores -- against the shipped mapping (one
// workgroup per tile, hardware dispatch).
//   hipcc -O3 --offload-arch=gfx950 tools/k1_persist.hip -o tools/k1_persist && ./tools/k1_persist
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
typedef double d2_t __attribute__((ext_vector_type(2)));

__global__ __launch_bounds__(256) void k_flat(double *T, const double *e, const double *rowr, long cols, long ld)
{
    const long row = blockIdx.y, col = ((long)blockIdx.x * 256 + threadIdx.x) * 2;
    if (col >= cols) return;
    const d2_t pr = *(const d2_t *)(rowr + col);
    const double ei = e[row];
    d2_t x = __builtin_nontemporal_load((const d2_t *)(T + row * ld + col));
    x[0] = fma(ei, pr[0], x[0]);
    x[1] = fma(ei, pr[1], x[1]);
    __builtin_nontemporal_store(x, (d2_t *)(T + row * ld + col));
}

// the shipped kernel reads its control block first (status, leaving row, rewritten column) and only then issues its loads;
// HOIST: the tile's loads go out before the control block is looked at (their addresses do not depend on it)
struct Ctl { int status, flip; long r, unit; };
template <bool HOIST>
__global__ __launch_bounds__(256) void k_ctl(double *T, const double *e, const double *rowr, long cols, long ld, const Ctl *c)
{
    const long row = blockIdx.y, col = ((long)blockIdx.x * 256 + threadIdx.x) * 2;
    d2_t x, pr;
    double ei;
    if (HOIST && col < cols) {
        x = __builtin_nontemporal_load((const d2_t *)(T + row * ld + col));
        pr = *(const d2_t *)(rowr + col);
        ei = e[row];
    }
    if (c->status != 0 || c->flip) return;
    const long r = c->r, unit = c->unit;
    if (col >= cols) return;
    if (!HOIST) {
        x = __builtin_nontemporal_load((const d2_t *)(T + row * ld + col));
        pr = *(const d2_t *)(rowr + col);
        ei = e[row];
    }
    if (row == r) { x[0] = ei * pr[0]; x[1] = ei * pr[1]; }
    else { x[0] = fma(ei, pr[0], x[0]); x[1] = fma(ei, pr[1], x[1]); }
    if (col == unit) x[0] = ei;
    if (col + 1 == unit) x[1] = ei;
    __builtin_nontemporal_store(x, (d2_t *)(T + row * ld + col));
}

// DEPTH tiles in flight per workgroup
template <int DEPTH>
__global__ __launch_bounds__(256) void k_persist(double *T, const double *e, const double *rowr, long cols, long ld, long strips, long total)
{
    const long G = gridDim.x;
    for (long t0 = blockIdx.x; t0 < total; t0 += G * DEPTH) {
        d2_t x[DEPTH], pr[DEPTH];
        double ei[DEPTH];
        long off[DEPTH];
        bool ok[DEPTH];
#pragma unroll
        for (int u = 0; u < DEPTH; ++u) {
            const long t = t0 + u * G;
            const long row = t / strips, col = ((t - row * strips) * 256 + threadIdx.x) * 2;
            ok[u] = t < total && col < cols;
            off[u] = row * ld + col;
            if (ok[u]) {
                x[u] = __builtin_nontemporal_load((const d2_t *)(T + off[u]));
                pr[u] = *(const d2_t *)(rowr + col);
                ei[u] = e[row];
            }
        }
#pragma unroll
        for (int u = 0; u < DEPTH; ++u) {
            if (ok[u]) {
                x[u][0] = fma(ei[u], pr[u][0], x[u][0]);
                x[u][1] = fma(ei[u], pr[u][1], x[u][1]);
                __builtin_nontemporal_store(x[u], (d2_t *)(T + off[u]));
            }
        }
    }
}

int main()
{
    const long rows = 20001, cols = 20001, ld = 20480;
    double *T, *e, *rowr;
    hipMalloc(&T, rows * ld * 8); hipMalloc(&e, rows * 8); hipMalloc(&rowr, ld * 8);
    hipMemset(T, 0, rows * ld * 8); hipMemset(e, 0, rows * 8); hipMemset(rowr, 0, ld * 8);
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    const double bytes = 16.0 * rows * cols;
    const long strips = (cols + 511) / 512, total = strips * rows;
    auto timeit = [&](const char *name, auto launch) {
        float tot = 0, best = 1e9f;
        const int reps = 20;
        for (int it = 0; it < reps + 3; ++it) {
            hipEventRecord(a, 0);
            launch();
            hipEventRecord(b, 0); hipEventSynchronize(b);
            float ms; hipEventElapsedTime(&ms, a, b);
            if (it >= 3) { tot += ms; if (ms < best) best = ms; }
        }
        printf("%-34s avg %.3f ms  %.2f TB/s   best %.3f ms  %.2f TB/s\n", name, tot / reps, bytes / (tot / reps) / 1e9, best, bytes / best / 1e9);
    };
    timeit("flat, one workgroup per tile", [&] { hipLaunchKernelGGL(k_flat, dim3((unsigned)strips, (unsigned)rows), dim3(256), 0, 0, T, e, rowr, cols, ld); });
    Ctl hc{0, 0, 77, 1234}, *dc;
    hipMalloc(&dc, sizeof hc); hipMemcpy(dc, &hc, sizeof hc, hipMemcpyHostToDevice);
    for (int rep = 0; rep < 2; ++rep) {
        timeit("flat, control block first (K1)", [&] { hipLaunchKernelGGL((k_ctl<false>), dim3((unsigned)strips, (unsigned)rows), dim3(256), 0, 0, T, e, rowr, cols, ld, dc); });
        timeit("flat, loads hoisted above it", [&] { hipLaunchKernelGGL((k_ctl<true>), dim3((unsigned)strips, (unsigned)rows), dim3(256), 0, 0, T, e, rowr, cols, ld, dc); });
        timeit("flat, one workgroup per tile", [&] { hipLaunchKernelGGL(k_flat, dim3((unsigned)strips, (unsigned)rows), dim3(256), 0, 0, T, e, rowr, cols, ld); });
    }
    for (int wgs : {4}) {
        char name[64];
        const unsigned G = 256u * wgs;
        snprintf(name, sizeof name, "persistent, %2d WG/CU, depth 1", wgs);
        timeit(name, [&] { hipLaunchKernelGGL((k_persist<1>), dim3(G), dim3(256), 0, 0, T, e, rowr, cols, ld, strips, total); });
        snprintf(name, sizeof name, "persistent, %2d WG/CU, depth 2", wgs);
        timeit(name, [&] { hipLaunchKernelGGL((k_persist<2>), dim3(G), dim3(256), 0, 0, T, e, rowr, cols, ld, strips, total); });
        snprintf(name, sizeof name, "persistent, %2d WG/CU, depth 4", wgs);
        timeit(name, [&] { hipLaunchKernelGGL((k_persist<4>), dim3(G), dim3(256), 0, 0, T, e, rowr, cols, ld, strips, total); });
    }
    return 0;
}
