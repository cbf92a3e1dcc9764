// lpx_tb_flush32.hip -- the deferred-update pass (lpx_tb_flush.hpp) for blocks of 32 pending updates
#include "lpx_tb_flush.hpp"

void lpx_tb_flush_launch32(const TbFlushArgs &f, int nt, dim3 grid, hipStream_t st)
{
    if (nt) hipLaunchKernelGGL((k_tb_flush2<32, 1, 8>), grid, dim3(TB2_THREADS), 0, st, f);
    else hipLaunchKernelGGL((k_tb_flush2<32, 0, 8>), grid, dim3(TB2_THREADS), 0, st, f);
}
