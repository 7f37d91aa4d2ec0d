// Wave-level reducers of the divide-and-conquer kernels (the `R` concept of dc_logic.h): shared by k_dc.hip (one launch per stage) and
// k_dcblock.hip (whole blocks inside one workgroup).
#pragma once
#include "common.hpp"

namespace mopt {

struct DcWaveRed {
    template <int K, class F> __device__ void sumk(int m, F f, double (&out)[K]) const {
        double acc[K];
#pragma unroll
        for (int q = 0; q < K; ++q) acc[q] = 0.0;
        for (int i = threadIdx.x & 63; i < m; i += WAVE) f(i, acc);
#pragma unroll
        for (int q = 0; q < K; ++q) out[q] = wave_sum(acc[q]);
    }
    template <class F> __device__ void for_each(int m, F f) const { for (int i = threadIdx.x & 63; i < m; i += WAVE) f(i); }
    __device__ void sync() const { __syncthreads(); }       // (64-thread workgroups: a wave-level sync)
};

// sum over the 16 lanes of a DPP row, returned to all of them (quad permutes, half-row mirror, row mirror)
__device__ __forceinline__ double dc_row16_sum(double v) {
    v += dpp_f64z<0xB1>(v);
    v += dpp_f64z<0x4E>(v);
    v += dpp_f64z<0x141>(v);
    v += dpp_f64z<0x140>(v);
    return v;
}
// Reducer over a 16-lane group: four secular roots per wave.  The groups of a wave diverge (different iteration counts); all
// cross-lane traffic stays inside a row, LDS accesses of one wave are served in program order, so sync() only has to stop the
// compiler from moving LDS accesses across it.
struct DcRow16Red {
    template <int K, class F> __device__ void sumk(int m, F f, double (&out)[K]) const {
        double acc[K];
#pragma unroll
        for (int q = 0; q < K; ++q) acc[q] = 0.0;
        for (int i = threadIdx.x & 15; i < m; i += 16) f(i, acc);
#pragma unroll
        for (int q = 0; q < K; ++q) out[q] = dc_row16_sum(acc[q]);
    }
    template <class F> __device__ void for_each(int m, F f) const { for (int i = threadIdx.x & 15; i < m; i += 16) f(i); }
    __device__ void sync() const { __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront"); __builtin_amdgcn_wave_barrier(); }
};


// Reducer over an 8-lane group (half a DPP row): eight secular roots per wave -- for merges with more roots than 16-lane groups
struct DcRow8Red {
    template <int K, class F> __device__ void sumk(int m, F f, double (&out)[K]) const {
        double acc[K];
#pragma unroll
        for (int q = 0; q < K; ++q) acc[q] = 0.0;
        for (int i = threadIdx.x & 7; i < m; i += 8) f(i, acc);
#pragma unroll
        for (int q = 0; q < K; ++q) {
            double v = acc[q];
            v += dpp_f64z<0xB1>(v);                // quad_perm [1,0,3,2]
            v += dpp_f64z<0x4E>(v);                // quad_perm [2,3,0,1]
            v += dpp_f64z<0x141>(v);               // row_half_mirror: the other quad of the 8-lane half row
            out[q] = v;
        }
    }
    template <class F> __device__ void for_each(int m, F f) const { for (int i = threadIdx.x & 7; i < m; i += 8) f(i); }
    __device__ void sync() const { __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront"); __builtin_amdgcn_wave_barrier(); }
};

}  // namespace mopt
