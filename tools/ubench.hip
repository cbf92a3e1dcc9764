// Micro-benchmarks behind the batched kernel's design notes: single-wave latencies on gfx950.
//   hipcc -O3 --offload-arch=gfx950 -ffp-contract=off tools/ubench.hip -o tools/ubench && ./tools/ubench
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <chrono>

__global__ void k_dep_fma64(double *out, double a, double b, int n, unsigned long long *cyc)
{
    double acc = out[threadIdx.x];
    unsigned long long t0 = clock64();
    for (int i = 0; i < n; ++i) {
#pragma unroll
        for (int u = 0; u < 16; ++u) acc = fma(acc, a, b);
    }
    unsigned long long t1 = clock64();
    out[threadIdx.x] = acc;
    if (threadIdx.x == 0) cyc[0] = t1 - t0;
}
__global__ void k_ind_fma64(double *out, double a, double b, int n, unsigned long long *cyc)
{
    double acc[8];
    for (int u = 0; u < 8; ++u) acc[u] = out[threadIdx.x] + u;
    unsigned long long t0 = clock64();
    for (int i = 0; i < n; ++i) {
#pragma unroll
        for (int r = 0; r < 2; ++r)
#pragma unroll
            for (int u = 0; u < 8; ++u) acc[u] = fma(acc[u], a, b);
    }
    unsigned long long t1 = clock64();
    double s = 0;
    for (int u = 0; u < 8; ++u) s += acc[u];
    out[threadIdx.x] = s;
    if (threadIdx.x == 0) cyc[0] = t1 - t0;
}
__global__ void k_dep_fma32(float *out, float a, float b, int n, unsigned long long *cyc)
{
    float acc = out[threadIdx.x];
    unsigned long long t0 = clock64();
    for (int i = 0; i < n; ++i) {
#pragma unroll
        for (int u = 0; u < 16; ++u) acc = fmaf(acc, a, b);
    }
    unsigned long long t1 = clock64();
    out[threadIdx.x] = acc;
    if (threadIdx.x == 0) cyc[0] = t1 - t0;
}
__global__ void k_lds_chase(int *out, int n, unsigned long long *cyc)
{
    __shared__ int next[1024];
    for (int i = threadIdx.x; i < 1024; i += blockDim.x) next[i] = (i + 64) & 1023;
    __syncthreads();
    int p = threadIdx.x;
    unsigned long long t0 = clock64();
    for (int i = 0; i < n; ++i) {
#pragma unroll
        for (int u = 0; u < 16; ++u) p = next[p];
    }
    unsigned long long t1 = clock64();
    out[threadIdx.x] = p;
    if (threadIdx.x == 0) cyc[0] = t1 - t0;
}
// independent ds_read_b64, conflict-free, 16 in flight
__global__ void k_lds_stream(double *out, int n, unsigned long long *cyc)
{
    __shared__ double buf[64 * 65];
    for (int i = threadIdx.x; i < 64 * 65; i += blockDim.x) buf[i] = i;
    __syncthreads();
    double s = 0;
    const int t = threadIdx.x & 63;
    unsigned long long t0 = clock64();
    for (int i = 0; i < n; ++i) {
        double v[16];
#pragma unroll
        for (int u = 0; u < 16; ++u) v[u] = buf[(u + (i & 3) * 16) * 65 + t];
#pragma unroll
        for (int u = 0; u < 16; ++u) s += v[u];
        asm volatile("" ::: "memory");
    }
    unsigned long long t1 = clock64();
    out[threadIdx.x] = s;
    if (threadIdx.x == 0) cyc[0] = t1 - t0;
}
// dependent fma chain with operands streamed from LDS, 16 loads ahead (what a perfect chain kernel would do)
__global__ void k_chain_lds(double *out, int n, unsigned long long *cyc)
{
    __shared__ double buf[64 * 65];
    __shared__ double cb[64];
    for (int i = threadIdx.x; i < 64 * 65; i += blockDim.x) buf[i] = 1e-3 * i;
    if (threadIdx.x < 64) cb[threadIdx.x] = threadIdx.x;
    __syncthreads();
    const int t = threadIdx.x & 63;
    double tot = 0;
    unsigned long long t0 = clock64();
    for (int i = 0; i < n; ++i) {
        double acc = 0;
#pragma unroll
        for (int k = 0; k < 64; ++k) acc = fma(cb[k], buf[k * 65 + t], acc);
        tot += acc;
        asm volatile("" ::: "memory");
    }
    unsigned long long t1 = clock64();
    out[threadIdx.x] = tot;
    if (threadIdx.x == 0) cyc[0] = t1 - t0;
}


// independent LDS reads, 16 in flight, integer xor as the only ALU work.  MODE 0: b64 per-lane stride-1, 1: b128 per-lane
// rows of 66 doubles, 2: b128 broadcast, 3: b64 broadcast
template <int MODE>
__global__ void k_lds_reads(int *out, int n, unsigned long long *cyc)
{
    __shared__ __attribute__((aligned(16))) int buf[64 * 66 * 2 + 64];
    for (int i = threadIdx.x; i < 64 * 66 * 2; i += blockDim.x) buf[i] = i;
    __syncthreads();
    const int t = threadIdx.x & 63;
    int s = 0;
    unsigned long long t0 = clock64();
    for (int i = 0; i < n; ++i) {
        if (MODE == 0 || MODE == 3) {
            typedef int v2 __attribute__((ext_vector_type(2)));
            v2 v[16];
#pragma unroll
            for (int u = 0; u < 16; ++u) v[u] = *(v2 *)&buf[((u + (i & 3) * 16) * 66 + (MODE == 0 ? t : 0)) * 2];
#pragma unroll
            for (int u = 0; u < 16; ++u) s ^= v[u].x ^ v[u].y;
        } else {
            typedef int v4 __attribute__((ext_vector_type(4)));
            v4 v[16];
#pragma unroll
            for (int u = 0; u < 16; ++u) v[u] = *(v4 *)&buf[(MODE == 1 ? t * 66 * 2 : 0) + (u + (i & 1) * 16) * 4];
#pragma unroll
            for (int u = 0; u < 16; ++u) s ^= v[u].x ^ v[u].y ^ v[u].z ^ v[u].w;
        }
        asm volatile("" ::: "memory");
    }
    unsigned long long t1 = clock64();
    out[threadIdx.x] = s;
    if (threadIdx.x == 0) cyc[0] = t1 - t0;
}
// register-resident chain: one operand in registers, the other a broadcast b128 (2 doubles per load)
__global__ __launch_bounds__(256) void k_chain_reg(double *out, int n, unsigned long long *cyc)
{
    __shared__ __attribute__((aligned(16))) double cb[64];
    if (threadIdx.x < 64) cb[threadIdx.x] = threadIdx.x;
    __syncthreads();
    double col[64];
#pragma unroll
    for (int k = 0; k < 64; ++k) col[k] = out[(threadIdx.x + k) & 1023];
    double tot = 0;
    unsigned long long t0 = clock64();
    for (int i = 0; i < n; ++i) {
        double acc = 0;
#pragma unroll
        for (int k = 0; k < 64; ++k) acc = fma(cb[k], col[k], acc);
        tot += acc;
        asm volatile("" ::: "memory");
    }
    unsigned long long t1 = clock64();
    out[threadIdx.x] = tot;
    if (threadIdx.x == 0) cyc[0] = t1 - t0;
}
// rank-1 update of a register column with a uniform special row: v_k = (k == r) ? e_k * p : fma(e_k, p, v_k)
__global__ __launch_bounds__(256) void k_update_reg(double *out, int n, int r, unsigned long long *cyc)
{
    __shared__ __attribute__((aligned(16))) double e[64];
    if (threadIdx.x < 64) e[threadIdx.x] = 1.0 + 1e-9 * threadIdx.x;
    __syncthreads();
    double col[64];
#pragma unroll
    for (int k = 0; k < 64; ++k) col[k] = out[(threadIdx.x + k) & 1023];
    const double p = out[threadIdx.x] + 1e-6;
    unsigned long long t0 = clock64();
    for (int i = 0; i < n; ++i) {
#pragma unroll
        for (int k = 0; k < 64; ++k) col[k] = (k == r) ? e[k] * p : fma(e[k], p, col[k]);
        asm volatile("" ::: "memory");
    }
    unsigned long long t1 = clock64();
    double tot = 0;
#pragma unroll
    for (int k = 0; k < 64; ++k) tot += col[k];
    out[threadIdx.x] = tot;
    if (threadIdx.x == 0) cyc[0] = t1 - t0;
}

// same update without per-element selects: plain fma for every register, then the special row through a dynamic
// (wave-uniform) register index
__global__ __launch_bounds__(256) void k_update_reg_dyn(double *out, int n, int r, unsigned long long *cyc)
{
    __shared__ __attribute__((aligned(16))) double e[64];
    if (threadIdx.x < 64) e[threadIdx.x] = 1.0 + 1e-9 * threadIdx.x;
    __syncthreads();
    double col[64];
#pragma unroll
    for (int k = 0; k < 64; ++k) col[k] = out[(threadIdx.x + k) & 1023];
    const double p = out[threadIdx.x] + 1e-6;
    unsigned long long t0 = clock64();
    for (int i = 0; i < n; ++i) {
#pragma unroll
        for (int k = 0; k < 64; ++k) col[k] = fma(e[k], p, col[k]);
        {
            const double v = e[r] * p;
#define UB_CASE(K) case K: col[K] = v; break;
#define UB_CASE4(K) UB_CASE(K) UB_CASE(K + 1) UB_CASE(K + 2) UB_CASE(K + 3)
#define UB_CASE16(K) UB_CASE4(K) UB_CASE4(K + 4) UB_CASE4(K + 8) UB_CASE4(K + 12)
            switch (__builtin_amdgcn_readfirstlane(r)) { UB_CASE16(0) UB_CASE16(16) UB_CASE16(32) UB_CASE16(48) }
        }
        asm volatile("" ::: "memory");
    }
    unsigned long long t1 = clock64();
    double tot = 0;
#pragma unroll
    for (int k = 0; k < 64; ++k) tot += col[k];
    out[threadIdx.x] = tot;
    if (threadIdx.x == 0) cyc[0] = t1 - t0;
}

template <typename F>
static void run(const char *name, F launch, double ops)
{
    unsigned long long *cyc;
    hipMalloc(&cyc, 8);
    launch(cyc); // warm
    hipDeviceSynchronize();
    auto w0 = std::chrono::steady_clock::now();
    launch(cyc);
    hipDeviceSynchronize();
    auto w1 = std::chrono::steady_clock::now();
    unsigned long long c;
    hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost);
    const double us = std::chrono::duration<double, std::micro>(w1 - w0).count();
    printf("%-28s %10llu ticks  %8.1f us wall  %7.2f ticks/op  %7.2f ns/op  (tick = %.2f ns)\n", name, c, us, c / ops, 1e3 * us / ops, 1e3 * us / c);
    hipFree(cyc);
}

int main()
{
    double *d; float *f; int *ip;
    hipMalloc(&d, 8 * 1024); hipMalloc(&f, 4 * 1024); hipMalloc(&ip, 4 * 1024);
    hipMemset(d, 0, 8 * 1024); hipMemset(f, 0, 4 * 1024);
    const int n = 20000;
    for (int threads : {64, 256}) {
        printf("--- one workgroup of %d threads\n", threads);
        run("dependent fma f64", [&](unsigned long long *c) { hipLaunchKernelGGL(k_dep_fma64, 1, threads, 0, 0, d, 0.999, 0.001, n, c); }, 16.0 * n);
        run("8 independent fma f64", [&](unsigned long long *c) { hipLaunchKernelGGL(k_ind_fma64, 1, threads, 0, 0, d, 0.999, 0.001, n, c); }, 16.0 * n);
        run("dependent fma f32", [&](unsigned long long *c) { hipLaunchKernelGGL(k_dep_fma32, 1, threads, 0, 0, f, 0.999f, 0.001f, n, c); }, 16.0 * n);
        run("LDS pointer chase b32", [&](unsigned long long *c) { hipLaunchKernelGGL(k_lds_chase, 1, threads, 0, 0, ip, n, c); }, 16.0 * n);
        run("LDS stream b64 (16 ahead)", [&](unsigned long long *c) { hipLaunchKernelGGL(k_lds_stream, 1, threads, 0, 0, d, n, c); }, 16.0 * n);
        run("fma chain fed from LDS", [&](unsigned long long *c) { hipLaunchKernelGGL(k_chain_lds, 1, threads, 0, 0, d, n / 4, c); }, 64.0 * (n / 4));
        run("LDS b64 per-lane", [&](unsigned long long *c) { hipLaunchKernelGGL(k_lds_reads<0>, 1, threads, 0, 0, ip, n, c); }, 16.0 * n);
        run("LDS b128 per-lane rows", [&](unsigned long long *c) { hipLaunchKernelGGL(k_lds_reads<1>, 1, threads, 0, 0, ip, n, c); }, 16.0 * n);
        run("LDS b128 broadcast", [&](unsigned long long *c) { hipLaunchKernelGGL(k_lds_reads<2>, 1, threads, 0, 0, ip, n, c); }, 16.0 * n);
        run("LDS b64 broadcast", [&](unsigned long long *c) { hipLaunchKernelGGL(k_lds_reads<3>, 1, threads, 0, 0, ip, n, c); }, 16.0 * n);
        run("chain: regs x b128 bcast", [&](unsigned long long *c) { hipLaunchKernelGGL(k_chain_reg, 1, threads, 0, 0, d, n / 4, c); }, 64.0 * (n / 4));
        run("update, dynamic index fix", [&](unsigned long long *c) { hipLaunchKernelGGL(k_update_reg_dyn, 1, threads, 0, 0, d, n / 4, 17, c); }, 64.0 * (n / 4));
        run("update of register column", [&](unsigned long long *c) { hipLaunchKernelGGL(k_update_reg, 1, threads, 0, 0, d, n / 4, 17, c); }, 64.0 * (n / 4));
    }
    return 0;
}
