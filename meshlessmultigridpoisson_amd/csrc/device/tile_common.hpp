// tile_common.hpp -- device helpers shared by the tile kernels (kernels.hip, kernels_mw.hip).
#pragma once
#include <hip/hip_runtime.h>

#include "kernels.hpp"

namespace mmg {
namespace {

__device__ __forceinline__ double wave_sum(double v)
{
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) v += __shfl_xor(v, m, 64);
    return v;
}

// cross-lane add inside groups of L lanes, DPP (no LDS traffic) up to L = 16
template <int CTRL>
__device__ __forceinline__ double dpp_add(double v)
{
    const unsigned long long u = __double_as_longlong(v);
    const int lo = __builtin_amdgcn_update_dpp(0, (int)(u & 0xffffffffull), CTRL, 0xf, 0xf, true);
    const int hi = __builtin_amdgcn_update_dpp(0, (int)(u >> 32), CTRL, 0xf, 0xf, true);
    return v + __longlong_as_double(((unsigned long long)(unsigned)hi << 32) | (unsigned)lo);
}

template <int L>
__device__ __forceinline__ double row_sum(double v)
{
    if (L >= 2) v = dpp_add<0xB1>(v);    // quad_perm [1,0,3,2]  : lane ^ 1
    if (L >= 4) v = dpp_add<0x4E>(v);    // quad_perm [2,3,0,1]  : lane ^ 2
    if (L >= 8) v = dpp_add<0x141>(v);   // row_half_mirror      : other quad of the 8
    if (L >= 16) v = dpp_add<0x140>(v);  // row_mirror           : other half of the 16
    if (L >= 32) v += __shfl_xor(v, 16, 64);
    if (L >= 64) v += __shfl_xor(v, 32, 64);
    return v;
}

// x accesses of the dependency-driven sweep go through agent-scope relaxed atomics
// (global_load/store ... sc1): L2-served, never stale in another CU's L1
// (MI355X_MICROARCH "Workgroup dispatch ... inter-workgroup visibility").
template <bool SC1>
__device__ __forceinline__ double ld_x(const double *p)
{
    if (SC1) return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    return *p;
}
template <bool SC1>
__device__ __forceinline__ void st_x(double *p, double v)
{
    if (SC1) __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    else *p = v;
}

// Wait until the coupled tiles of `tile` have published what this sweep needs: earlier tiles
// dep_idx[d0, d1) the CURRENT sweep (want_now), later tiles later_idx[l0, l1) the PREVIOUS one
// (want_prev).  One lane per dependency, relaxed agent-scope polls, wave-wide vote.  The wait is
// bounded (a.spin_bound polls): when it runs out -- the co-residency / progress assumption of the
// launch did not hold, e.g. foreign kernels occupy the CUs -- the device error word is set and false is
// returned; every later wait of the launch then returns false at once.  The host notices the word at
// its next synchronisation, restores x and repeats the sweeps with one launch per phase (capi.hip: settle).
template <int SLEEP>
__device__ __forceinline__ bool wait_for_tiles(const TileArgs &a, int tile, int d0, int d1, int l0, int l1,
                                               unsigned want_now, unsigned want_prev)
{
    const int lane = threadIdx.x;
    const int n_wait = (d1 - d0) + (l1 - l0);
    bool good = __all(__hip_atomic_load(a.error, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0u);
    for (int base = 0; good && base < n_wait; base += 64) {
        const int k = base + lane;
        const bool mine = k < n_wait;
        const bool early = k < (d1 - d0);
        const int dep = !mine ? tile : (early ? a.p.dep_idx[d0 + k] : a.p.later_idx[l0 + (k - (d1 - d0))]);
        const unsigned need = early ? want_now : want_prev;
        const unsigned *flag = a.done + dep;
        bool ok = !mine;
        for (int spin = 0; spin < a.spin_bound; ++spin) {
            // flags only grow; unsigned difference handles wrap-around
            if (!ok) ok = (int)(__hip_atomic_load(flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) - need) >= 0;
            if (__all(ok)) break;
            if ((spin & 1023) == 1023 && __hip_atomic_load(a.error, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u) break;
            __builtin_amdgcn_s_sleep(SLEEP);
        }
        if (!__all(ok)) {
            good = false;
            if (lane == 0) __hip_atomic_store(a.error, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
    // compiler-level ordering point (no cache maintenance): the x loads of process_tile stay below the polls
    __atomic_signal_fence(__ATOMIC_SEQ_CST);
    return good;
}


}  // namespace
}  // namespace mmg
