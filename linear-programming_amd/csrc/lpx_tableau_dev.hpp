// lpx_tableau_dev.hpp -- device helpers the tableau kernels of more than one translation unit share
#pragma once
#include "lpx_device.hpp"

// ---- which variable a column slot holds.  The full form carries every column of [col0, col0 + n); the condensed form
// (slot_var != nullptr) carries the nonbasic columns only, slot k holding variable slot_var[k], var_slot[] the inverse map
__device__ __forceinline__ int64_t t_var(const lpx_tableau &s, int64_t slot) { return s.slot_var ? s.slot_var[slot] : s.col0 + slot; }
// local column of variable j, -1 if this rank does not hold it
__device__ __forceinline__ int64_t t_slot(const lpx_tableau &s, int64_t j)
{
    if (s.var_slot) return s.var_slot[j];
    const int64_t k = j - s.col0;
    return (k < 0 || k >= s.n) ? -1 : k;
}
// bookkeeping of a basis change (single thread): the full form flips var_state, the condensed form hands the slot over
// It leaves ctrl->enter_slot = the local column that the update rewrites as a whole (full form: the entering variable's,
// which becomes a unit vector; condensed: the slot handed to the leaving variable, which receives the eta column), -1 if
// that column is not on this rank.  K1 reads it with the rest of the control block (no lookup through var_slot in
// front of its 10^5..10^6 short workgroups).
__device__ __forceinline__ void t_swap_columns(const lpx_tableau &s, int64_t j, int64_t leaving)
{
    if (s.slot_var) {
        const int64_t k = s.var_slot[j];
        if (k >= 0) {
            s.slot_var[k] = leaving;
            s.var_slot[leaving] = k;
        }
        s.var_slot[j] = -1;
        s.ctrl->enter_slot = k;
        return;
    }
    const int64_t ll = leaving - s.col0, jl = j - s.col0;
    if (ll >= 0 && ll < s.n) s.var_state[ll] = LPX_VS_LOWER;
    if (jl >= 0 && jl < s.n) s.var_state[jl] = LPX_VS_BASIC;
    s.ctrl->enter_slot = (jl >= 0 && jl < s.n) ? jl : -1;
}

// ---- a 64-bit value of another lane (two v_readlane_b32); never inside a branch the source lane does not take
__device__ __forceinline__ int64_t tf_readlane_i64(int64_t v, int lane)
{
    const unsigned lo = (unsigned)__builtin_amdgcn_readlane((int)(v & 0xffffffffLL), lane);
    const unsigned hi = (unsigned)__builtin_amdgcn_readlane((int)(v >> 32), lane);
    return (int64_t)(((unsigned long long)hi << 32) | lo);
}
__device__ __forceinline__ double tf_readlane_f64(double v, int lane)
{
    return __longlong_as_double(tf_readlane_i64(__double_as_longlong(v), lane));
}
