// lpx_tb_flush.hip -- instantiations of the deferred-update pass (lpx_tb_flush.hpp) and their launcher
#include "lpx_tb_flush.hpp"

template <int KB>
static void tb_launch_flush(const TbFlushArgs &f, int nt, dim3 grid, hipStream_t st)
{
    if (nt) hipLaunchKernelGGL((k_tb_flush2<KB, 1, 8>), grid, dim3(TB2_THREADS), 0, st, f);
    else hipLaunchKernelGGL((k_tb_flush2<KB, 0, 8>), grid, dim3(TB2_THREADS), 0, st, f);
}

void lpx_tb_flush_launch(const TbFlushArgs &f, int kb, int nt, void *stream)
{
    const dim3 grid((unsigned)lpx_cdiv(f.cols, TB2_COLS), (unsigned)lpx_cdiv(f.rows, f.rpb), 1);
    hipStream_t st = (hipStream_t)stream;
    if (kb == 8) tb_launch_flush<8>(f, nt, grid, st);
    else if (kb == 16) tb_launch_flush<16>(f, nt, grid, st);
    else tb_launch_flush<32>(f, nt, grid, st);
}
