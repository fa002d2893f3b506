// ransac_common.hip.h -- device helpers shared by the F-matrix and PnP RANSAC kernels.
#pragma once
#include <hip/hip_runtime.h>

#include <cfloat>
#include <cstdint>

#include "../../include/svo_math.h"  // sin / cos / acos / cbrt / log shared with the oracle, bit for bit

namespace svo {

// Counter-based generator: draw k of RANSAC iteration i is a pure function of
// (seed, i, k), so every iteration can be evaluated concurrently and the SEQUENTIAL
// algorithm (first-best-wins, adaptive iteration count) is reproduced afterwards.
__host__ __device__ __forceinline__ uint32_t rng_u32(uint64_t seed, uint32_t iter, uint32_t draw)
{
    uint64_t z = seed + 0x9E3779B97F4A7C15ull * ((((uint64_t)iter << 32) | draw) + 1ull);
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    z ^= z >> 31;
    return (uint32_t)(z >> 32);
}

constexpr int kMaxAttempts = 8;
constexpr int kMaxDraws = 64;

// cv RANSACUpdateNumIters
__device__ __forceinline__ int update_num_iters(double p, double ep, int model_points, int max_iters)
{
    p = p < 0 ? 0 : (p > 1 ? 1 : p);
    ep = ep < 0 ? 0 : (ep > 1 ? 1 : ep);
    double num = 1. - p;
    if (num < DBL_MIN)
        num = DBL_MIN;
    double denom = 1. - svo_powi(1. - ep, model_points);  // include/svo_math.h: the oracle runs the same operations
    if (denom < DBL_MIN)
        return 0;
    num = svo_log(num);
    denom = svo_log(denom);
    return denom >= 0 || -num >= max_iters * (-denom) ? max_iters : (int)rint(num / denom);
}

// running state of the sequential RANSAC loop, kept in HBM between the phase kernels
struct RansacState {
    int niters;      // current iteration bound (starts at max_iters)
    int next_iter;   // first iteration the select pass has not looked at yet
    int best_iter;   // -1 = none
    int best_model;
    int best_count;
    int done;        // 1: loop finished (next_iter >= niters or a subset draw failed)
    int iters_run;
    int pad;
};

// exact wave-wide integer sum (values small enough that no carry handling is needed)
__device__ __forceinline__ int wave_sum_small(int v)
{
    v += __builtin_amdgcn_update_dpp(0, v, 0xB1, 0xf, 0xf, false);   // quad_perm [1,0,3,2]
    v += __builtin_amdgcn_update_dpp(0, v, 0x4E, 0xf, 0xf, false);   // quad_perm [2,3,0,1]
    v += __builtin_amdgcn_update_dpp(0, v, 0x141, 0xf, 0xf, false);  // row_half_mirror
    v += __builtin_amdgcn_update_dpp(0, v, 0x140, 0xf, 0xf, false);  // row_mirror
    return __builtin_amdgcn_readlane(v, 0) + __builtin_amdgcn_readlane(v, 16) +
           __builtin_amdgcn_readlane(v, 32) + __builtin_amdgcn_readlane(v, 48);
}

// Replays the SEQUENTIAL RANSAC loop over iterations [st->next_iter, it_end) (called by one
// thread: the last wave of fr_ransac_kernel / of pnp_solve_kernel's first phase, thread 0 of pnp_finish_kernel): first-best-wins,
// adaptive iteration bound, stop at a failed sample -- so that evaluating all hypotheses
// concurrently gives exactly the serial algorithm's answer.  MPI = models per iteration slot
// (nmodels[it] in -1 (sampling failed) .. MPI, counts[it * MPI + k]).
template <int MPI>
__device__ inline RansacState ransac_replay(const RansacState *st_in, int first, int it_end, int max_iters, int n,
                                            double confidence, const int *__restrict__ nmodels,
                                            const int *__restrict__ counts, int model_points)
{
    RansacState s;
    if (first) {
        s.niters = max_iters;
        s.next_iter = 0;
        s.best_iter = -1;
        s.best_model = 0;
        s.best_count = 0;
        s.done = 0;
        s.iters_run = 0;
        s.pad = 0;
    } else {
        s = *st_in;
        if (s.done)
            return s;
    }
    int it = s.next_iter;
    if (n < model_points) {
        s.done = 1;
    } else {
        for (; it < it_end; it++) {
            if (it >= s.niters) {
                s.done = 1;
                break;
            }
            const int nm = nmodels[it];
            if (nm < 0) {
                s.done = 1;
                break;
            }
            for (int k = 0; k < nm; k++) {
                const int c = counts[it * MPI + k];
                const int floor_cnt = s.best_count > model_points - 1 ? s.best_count : model_points - 1;
                if (c > floor_cnt) {
                    s.best_count = c;
                    s.best_iter = it;
                    s.best_model = k;
                    s.niters = update_num_iters(confidence, (double)(n - c) / n, model_points, s.niters);
                }
            }
        }
        if (it >= s.niters || it >= max_iters)
            s.done = 1;
    }
    s.next_iter = it;
    s.iters_run = it;
    return s;
}

// hand-off of LDS data between lanes of ONE wave (LDS ops of a wave execute in order; this
// only pins the compiler)
__device__ __forceinline__ void wave_lds_fence()
{
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

}  // namespace svo
