// fma_rate.hip -- issue rate of v_fma_f64 on gfx950: plain, with an SGPR operand, and with a DPP row broadcast operand.
// One wave per SIMD (1024 waves), 16 independent chains per lane.  build: hipcc -O3 --offload-arch=gfx950 tools/fma_rate.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
template <int MODE>
__global__ __launch_bounds__(64) void k(double *out, const double *in, int iters)
{
    double x[16], p[16];
    for (int q = 0; q < 16; ++q) { x[q] = in[threadIdx.x + q]; p[q] = in[threadIdx.x + 16 + q]; }
    double e = in[threadIdx.x & 15];
    double es = in[blockIdx.x & 7];
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int q = 0; q < 16; ++q) {
            if (MODE == 0) x[q] = fma(e, p[q], x[q]);
            if (MODE == 1) asm volatile("v_fmac_f64_e32 %0, %1, %2" : "+v"(x[q]) : "s"(es), "v"(p[q]));
            if (MODE == 2) asm volatile("v_fmac_f64_dpp %0, %1, %2 row_newbcast:3 row_mask:0xf bank_mask:0xf" : "+v"(x[q]) : "v"(e), "v"(p[q]));
            if (MODE == 3) asm volatile("v_fmac_f64_e32 %0, %1, %2" : "+v"(x[q]) : "v"(e), "v"(p[q]));
        }
    }
    double s = 0;
    for (int q = 0; q < 16; ++q) s += x[q];
    out[blockIdx.x * 64 + threadIdx.x] = s;
}
int main()
{
    double *in, *out;
    hipMalloc(&in, 8 * 4096);
    hipMalloc(&out, 8 * 64 * 8192);
    {   // random operands in (-1, 1): an all-zero data path draws less power and may clock higher
        double *h = (double *)malloc(8 * 4096);
        unsigned long long z = 88172645463325252ULL;
        for (int i = 0; i < 4096; ++i) { z ^= z << 13; z ^= z >> 7; z ^= z << 17; h[i] = (getenv("FMA_ZERO") ? 0.0 : 1.0) * ((double)(z & 0xffffff) / 8388608.0 - 1.0) * 0.99; }
        hipMemcpy(in, h, 8 * 4096, hipMemcpyHostToDevice);
        free(h);
    }
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    const int iters = 20000;
    const char *names[] = {"fma (compiler)", "v_fmac_f64 sgpr src0", "v_fmac_f64_dpp row_newbcast", "v_fmac_f64 vgpr"};
    for (int waves_per_simd = 1; waves_per_simd <= 4; waves_per_simd *= 2)
        for (int mode = 0; mode < 4; ++mode) {
            float best = 1e30f;
            for (int rep = 0; rep < 3; ++rep) {
                hipEventRecord(e0, 0);
                const int g = 1024 * waves_per_simd;
                if (mode == 0) k<0><<<g, 64>>>(out, in, iters);
                if (mode == 1) k<1><<<g, 64>>>(out, in, iters);
                if (mode == 2) k<2><<<g, 64>>>(out, in, iters);
                if (mode == 3) k<3><<<g, 64>>>(out, in, iters);
                hipEventRecord(e1, 0);
                hipEventSynchronize(e1);
                float ms;
                hipEventElapsedTime(&ms, e0, e1);
                if (ms < best) best = ms;
            }
            const double instr = 16.0 * iters * waves_per_simd; // per SIMD
            printf("%d wave(s)/SIMD %-30s %8.3f ms  %.2f ns per wave-instruction per SIMD (%.1f cycles at 2.4 GHz)  %.1f Tfma/s\n", waves_per_simd,
                   names[mode], best, best * 1e6 / instr, best * 1e6 / instr * 2.4, instr * 1024 * 64 / best / 1e9);
        }
    return 0;
}
