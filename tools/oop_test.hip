// Experiment: is the rank-1 stream faster out of place (read T0, write T1) than in place?  Same mapping as K1
// (one row x 512 columns per 256-lane workgroup, 16 B per lane, non-temporal accesses, x-fastest dispatch).
//   hipcc -O3 --offload-arch=gfx950 tools/oop_test.hip -o tools/oop_test && ./tools/oop_test
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
typedef double d2_t __attribute__((ext_vector_type(2)));
// MAP: 0 = workgroups in address order (hardware deals consecutive ids round-robin to the 8 XCDs); 1 = every XCD streams
// runs of 8 consecutive 4-KiB strips (32 KiB); 2 = every XCD owns one contiguous eighth of the matrix
template <int NT, int MAP>
__global__ __launch_bounds__(256) void kmap(const double *src, double *dst, const double *e, const double *rowr, long cols, long ld,
                                           long strips, long total)
{
    long id = blockIdx.x;
    if (MAP == 1) {
        const long g = id / 64, w = id % 64;
        id = g * 64 + (w % 8) * 8 + w / 8;
    } else if (MAP == 2) {
        id = (id % 8) * (total / 8) + id / 8;
    }
    if (id >= total) return;
    const long row = id / strips, col = ((id % strips) * 256 + threadIdx.x) * 2;
    if (col >= cols) return;
    const d2_t pr = *(const d2_t *)(rowr + col);
    const double ei = e[row];
    d2_t x = NT ? __builtin_nontemporal_load((const d2_t *)(src + row * ld + col)) : *(const d2_t *)(src + row * ld + col);
    x[0] = fma(ei, pr[0], x[0]);
    x[1] = fma(ei, pr[1], x[1]);
    if (NT) __builtin_nontemporal_store(x, (d2_t *)(dst + row * ld + col));
    else *(d2_t *)(dst + row * ld + col) = x;
}
template <int NT>
__global__ __launch_bounds__(256) void k(const double *src, double *dst, const double *e, const double *rowr, long cols, long ld)
{
    const long row = blockIdx.y, col = ((long)blockIdx.x * 256 + threadIdx.x) * 2;
    if (col >= cols) return;
    const d2_t pr = *(const d2_t *)(rowr + col);
    const double ei = e[row];
    d2_t x = NT ? __builtin_nontemporal_load((const d2_t *)(src + row * ld + col)) : *(const d2_t *)(src + row * ld + col);
    x[0] = fma(ei, pr[0], x[0]);
    x[1] = fma(ei, pr[1], x[1]);
    if (NT) __builtin_nontemporal_store(x, (d2_t *)(dst + row * ld + col));
    else *(d2_t *)(dst + row * ld + col) = x;
}
int main(int argc, char **argv)
{
    const long rows = 20001, cols = 40001, ld = argc > 1 ? atol(argv[1]) : 40002; // row stride experiment: ld as argument
    double *T0, *T1, *e, *rowr;
    hipMalloc(&T0, rows * ld * 8); hipMalloc(&T1, rows * ld * 8); hipMalloc(&e, rows * 8); hipMalloc(&rowr, ld * 8);
    hipMemset(T0, 0, rows * ld * 8); hipMemset(T1, 0, rows * ld * 8); hipMemset(e, 0, rows * 8); hipMemset(rowr, 0, ld * 8);
    dim3 grid((unsigned)((cols + 511) / 512), (unsigned)rows);
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    const double bytes = 16.0 * rows * cols;
    printf("ld = %ld doubles (row stride %ld B)\n", ld, ld * 8);
    for (int mode = 0; mode < 4; ++mode) {
        const bool oop = mode & 1, nt = mode & 2;
        float best = 1e9f, tot = 0;
        const int reps = 20;
        for (int it = 0; it < reps + 3; ++it) {
            const double *s = (oop && (it & 1)) ? T1 : T0;
            double *d = oop ? ((it & 1) ? T0 : T1) : T0;
            hipEventRecord(a, 0);
            if (nt) hipLaunchKernelGGL(k<1>, grid, dim3(256), 0, 0, s, d, e, rowr, cols, ld);
            else hipLaunchKernelGGL(k<0>, grid, dim3(256), 0, 0, s, d, e, rowr, cols, ld);
            hipEventRecord(b, 0); hipEventSynchronize(b);
            float ms; hipEventElapsedTime(&ms, a, b);
            if (it >= 3) { tot += ms; if (ms < best) best = ms; }
        }
        printf("%-12s %-5s avg %.3f ms  %.2f TB/s   best %.3f ms  %.2f TB/s\n", oop ? "out-of-place" : "in-place", nt ? "nt" : "plain",
               tot / reps, bytes / (tot / reps) / 1e9, best, bytes / best / 1e9);
    }
    // workgroup -> XCD mapping variants (in place, non-temporal, 1-D grid)
    const long strips = (cols + 511) / 512, total = ((strips * rows + 63) / 64) * 64;
    for (int map = 0; map < 3; ++map) {
        float tot = 0, best = 1e9f;
        const int reps = 20;
        for (int it = 0; it < reps + 3; ++it) {
            hipEventRecord(a, 0);
            if (map == 0) hipLaunchKernelGGL((kmap<1, 0>), dim3((unsigned)total), dim3(256), 0, 0, T0, T0, e, rowr, cols, ld, strips, strips * rows);
            if (map == 1) hipLaunchKernelGGL((kmap<1, 1>), dim3((unsigned)total), dim3(256), 0, 0, T0, T0, e, rowr, cols, ld, strips, strips * rows);
            if (map == 2) hipLaunchKernelGGL((kmap<1, 2>), dim3((unsigned)total), dim3(256), 0, 0, T0, T0, e, rowr, cols, ld, strips, (strips * rows / 8) * 8);
            hipEventRecord(b, 0); hipEventSynchronize(b);
            float ms; hipEventElapsedTime(&ms, a, b);
            if (it >= 3) { tot += ms; if (ms < best) best = ms; }
        }
        printf("map %d           avg %.3f ms  %.2f TB/s   best %.3f ms  %.2f TB/s\n", map, tot / reps, bytes / (tot / reps) / 1e9, best,
               bytes / best / 1e9);
    }
    return 0;
}
