// What does a launch-bound pivot cost at best on MI355X?  Per-kernel time of chains of small kernels replayed as a
// hipGraph (the way liblpx replays the single-launch tableau chain), for kernels made of 0..3 dependent global-memory
// round trips, with and without an 8 MB -> 8 MB streaming pass beside them.
//   hipcc -O3 --offload-arch=gfx950 tools/launch_floor.hip -o tools/launch_floor && ./tools/launch_floor
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <chrono>
#include <vector>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

// `trips` dependent loads per thread through a table written by the previous kernel in the chain (so it is not in this
// XCD's L2), 4 workgroups; each workgroup ends with one LDS exchange like a block reduction
__global__ __launch_bounds__(256) void k_chase(const int *__restrict__ in, int *__restrict__ out, int n, int trips)
{
    __shared__ int sh[4];
    int p = (blockIdx.x * 256 + threadIdx.x) % n;
    for (int t = 0; t < trips; ++t) p = in[p];
    if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = p;
    __syncthreads();
    p += sh[0] + sh[1] + sh[2] + sh[3];
    out[(blockIdx.x * 256 + threadIdx.x) % n] = (p & 0x7fffffff) % n;
}
// the same chain in the first 4 workgroups, an 8 MB read-modify-write stream in the others
__global__ __launch_bounds__(256) void k_both(const int *__restrict__ in, int *__restrict__ out, int n, int trips,
                                               const double2 *__restrict__ src, double2 *__restrict__ dst, long cnt)
{
    if (blockIdx.x < 4) {
        __shared__ int sh[4];
        int p = (blockIdx.x * 256 + threadIdx.x) % n;
        for (int t = 0; t < trips; ++t) p = in[p];
        if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = p;
        __syncthreads();
        p += sh[0] + sh[1] + sh[2] + sh[3];
        out[(blockIdx.x * 256 + threadIdx.x) % n] = (p & 0x7fffffff) % n;
        return;
    }
    const long i = (long)(blockIdx.x - 4) * 256 + threadIdx.x;
    if (i < cnt) {
        double2 v = src[i];
        v.x = v.x * 1.0000001 + 1e-9;
        v.y = v.y * 1.0000001 + 1e-9;
        dst[i] = v;
    }
}

int main()
{
    const int n = 1 << 20; // 4 MB table
    int *a, *b;
    double2 *s, *d;
    const long cnt = 1001L * 1008 / 2;
    CK(hipMalloc(&a, n * 4));
    CK(hipMalloc(&b, n * 4));
    CK(hipMalloc(&s, cnt * 16));
    CK(hipMalloc(&d, cnt * 16));
    std::vector<int> h(n);
    for (int i = 0; i < n; ++i) h[i] = (int)(((long)i * 7919 + 12345) % n);
    CK(hipMemcpy(a, h.data(), n * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(b, h.data(), n * 4, hipMemcpyHostToDevice));
    CK(hipMemset(s, 0, cnt * 16));
    CK(hipMemset(d, 0, cnt * 16));
    hipStream_t st;
    CK(hipStreamCreate(&st));
    for (int both = 0; both < 2; ++both)
        for (int trips = 0; trips <= 4; ++trips) {
            hipGraph_t g;
            hipGraphExec_t ge;
            CK(hipStreamBeginCapture(st, hipStreamCaptureModeRelaxed));
            for (int u = 0; u < 16; ++u) {
                const int *in = (u & 1) ? b : a;
                int *out = (u & 1) ? a : b;
                if (both)
                    hipLaunchKernelGGL(k_both, dim3(4 + (unsigned)((cnt + 255) / 256)), dim3(256), 0, st, in, out, n, trips,
                                       (u & 1) ? d : s, (u & 1) ? s : d, cnt);
                else
                    hipLaunchKernelGGL(k_chase, dim3(4), dim3(256), 0, st, in, out, n, trips);
            }
            CK(hipStreamEndCapture(st, &g));
            CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
            for (int w = 0; w < 20; ++w) CK(hipGraphLaunch(ge, st));
            CK(hipStreamSynchronize(st));
            const int reps = 200;
            auto t0 = std::chrono::steady_clock::now();
            for (int w = 0; w < reps; ++w) CK(hipGraphLaunch(ge, st));
            CK(hipStreamSynchronize(st));
            const double us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count();
            printf("%s trips=%d: %.2f us per kernel\n", both ? "chain + 8MB stream" : "chain only        ", trips, us / (reps * 16));
            CK(hipGraphExecDestroy(ge));
            CK(hipGraphDestroy(g));
        }
    return 0;
}
