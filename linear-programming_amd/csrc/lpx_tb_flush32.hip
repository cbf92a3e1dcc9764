// lpx_tb_flush32.hip -- the deferred-update pass (lpx_tb_flush.hpp) for blocks of 32 pending updates
#include "lpx_tb_flush.hpp"

void lpx_tb_flush_launch32(const TbFlushArgs &f, int nt, dim3 grid, hipStream_t st)
{
    if (nt) {
        // two kernels, each leaves at once when the block is the other's: the condensed layout's full blocks in an instance
        // of their own (123 VGPRs, 4 waves per SIMD; with the general chains in the same kernel: 218 and 2), the rest
        hipLaunchKernelGGL((k_tb_flush2<32, 1, 8, TB2_THREADS, 1, 1>), grid, dim3(TB2_THREADS), 0, st, f);
        hipLaunchKernelGGL((k_tb_flush2<32, 1, 8, TB2_THREADS, 1, 2>), grid, dim3(TB2_THREADS), 0, st, f);
    }
    else hipLaunchKernelGGL((k_tb_flush2<32, 0, 8>), grid, dim3(TB2_THREADS), 0, st, f);
}
