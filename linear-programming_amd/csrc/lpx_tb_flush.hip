// lpx_tb_flush.hip -- the deferred-update pass (lpx_tb_flush.hpp): launcher and the instantiations for blocks of 8 and 16
// (blocks of 32 are compiled in lpx_tb_flush32.hip: the unrolled chains dominate the build time, two files build in parallel)
#include "lpx_tb_flush.hpp"

void lpx_tb_flush_launch32(const TbFlushArgs &f, int nt, dim3 grid, hipStream_t st);

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
    else lpx_tb_flush_launch32(f, nt, grid, st);
}
