// Shared device helpers for the gfx950 kernels (wave64 reductions, launch bookkeeping).
#pragma once
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdint.h>
#include <map>
#include <string>
#include <vector>

#include "../../include/mopt_hip.h"

namespace mopt {

constexpr int WAVE = 64;

// Cross-lane moves on the DPP path (no LDS traffic, unlike ds_bpermute behind __shfl): quad permutes, row mirrors, then the
// row broadcasts of GFX9; lanes outside row_mask receive `old`.
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ double dpp_f64(double old, double v) {
    const int lo = __builtin_amdgcn_update_dpp(__double2loint(old), __double2loint(v), CTRL, ROW_MASK, 0xF, false);
    const int hi = __builtin_amdgcn_update_dpp(__double2hiint(old), __double2hiint(v), CTRL, ROW_MASK, 0xF, false);
    return __hiloint2double(hi, lo);
}
__device__ __forceinline__ double lane_bcast(double v, int lane) {
    return __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(v), lane), __builtin_amdgcn_readlane(__double2loint(v), lane));
}
// The same move with bound_ctrl and all rows/banks enabled: no `old` operand to materialise (two v_mov_b32_dpp, nothing else);
// a lane without a source lane (row_bcast into row 0) reads 0.
template <int CTRL>
__device__ __forceinline__ double dpp_f64z(double v) {
    const int lo = __builtin_amdgcn_mov_dpp(__double2loint(v), CTRL, 0xF, 0xF, true);
    const int hi = __builtin_amdgcn_mov_dpp(__double2hiint(v), CTRL, 0xF, 0xF, true);
    return __hiloint2double(hi, lo);
}
// Sum over the 64 lanes, valid in LANE 63 ONLY (other lanes hold partial sums).  Order: pairs, quads, half rows, rows (16
// lanes), then row r += row r-1, then rows 2,3 += lane 31: lane 63 ends with ((r0+r1) + (r2+r3)) in the order r3 + r2, + (r1 + r0).
__device__ __forceinline__ double wave_sum63(double v) {
    v += dpp_f64z<0xB1>(v);                // quad_perm [1,0,3,2]
    v += dpp_f64z<0x4E>(v);                // quad_perm [2,3,0,1]
    v += dpp_f64z<0x141>(v);               // row_half_mirror
    v += dpp_f64z<0x140>(v);               // row_mirror
    v += dpp_f64z<0x142>(v);               // row_bcast15: lane 15 of row r to row r+1 (row 0 reads 0)
    v += dpp_f64z<0x143>(v);               // row_bcast31: lane 31 to rows 2 and 3 (rows 0, 1 read 0)
    return v;
}
// Sum over the 64 lanes, returned to every lane.
__device__ __forceinline__ double wave_sum(double v) { return lane_bcast(wave_sum63(v), 63); }
__device__ __forceinline__ double wave_max(double v) {
    v = fmax(v, dpp_f64<0xB1, 0xF>(v, v));
    v = fmax(v, dpp_f64<0x4E, 0xF>(v, v));
    v = fmax(v, dpp_f64<0x141, 0xF>(v, v));
    v = fmax(v, dpp_f64<0x140, 0xF>(v, v));
    v = fmax(v, dpp_f64<0x142, 0xA>(v, v));
    v = fmax(v, dpp_f64<0x143, 0xC>(v, v));
    return lane_bcast(v, 63);
}
__device__ __forceinline__ int wave_min_int(int v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { int t = __shfl_xor(v, o, WAVE); v = t < v ? t : v; }
    return v;
}

// ---- four sums at once (gfx950: v_permlane32_swap, v_permlane16_swap)
__device__ __forceinline__ void swap_halves(double &a, double &b) {        // a[32..63] <-> b[0..31]
    auto r0 = __builtin_amdgcn_permlane32_swap((unsigned)__double2loint(a), (unsigned)__double2loint(b), false, false);
    auto r1 = __builtin_amdgcn_permlane32_swap((unsigned)__double2hiint(a), (unsigned)__double2hiint(b), false, false);
    a = __hiloint2double((int)r1[0], (int)r0[0]);
    b = __hiloint2double((int)r1[1], (int)r0[1]);
}
__device__ __forceinline__ void swap_rows(double &a, double &b) {          // odd 16-lane rows of a <-> even rows of b
    auto r0 = __builtin_amdgcn_permlane16_swap((unsigned)__double2loint(a), (unsigned)__double2loint(b), false, false);
    auto r1 = __builtin_amdgcn_permlane16_swap((unsigned)__double2hiint(a), (unsigned)__double2hiint(b), false, false);
    a = __hiloint2double((int)r1[0], (int)r0[0]);
    b = __hiloint2double((int)r1[1], (int)r0[1]);
}
// Sums over the 64 lanes of four values at once.  Result: every lane of 16-lane row 0 holds sum(d0), row 1 sum(d2), row 2 sum(d1),
// row 3 sum(d3).
__device__ __forceinline__ double wave_sum4(double d0, double d1, double d2, double d3) {
    swap_halves(d0, d1);
    swap_halves(d2, d3);
    double e0 = d0 + d1;                    // lanes 0..31: d0 (lane + lane+32), lanes 32..63: d1
    double e1 = d2 + d3;                    // lanes 0..31: d2, lanes 32..63: d3
    swap_rows(e0, e1);
    double f = e0 + e1;                     // rows: d0, d2, d1, d3 (16 partial sums each)
    f += dpp_f64z<0xB1>(f);                 // quad_perm [1,0,3,2]
    f += dpp_f64z<0x4E>(f);                 // quad_perm [2,3,0,1]
    f += dpp_f64z<0x141>(f);                // row_half_mirror
    f += dpp_f64z<0x140>(f);                // row_mirror
    return f;
}

// Workgroup barrier that orders LDS traffic only: __syncthreads() also waits for the acknowledgement of every global store still
// in flight (1-2 us per step in the persistent tridiagonalisation kernels).
__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

// Block-wide sum of K values per thread; blockDim.x must be a multiple of 64 and <= 1024.
// red must point to K * 16 doubles of LDS.  Result is returned to every thread.
template <int K>
__device__ __forceinline__ void block_sum(double (&v)[K], double *red) {
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6, nw = blockDim.x >> 6;
#pragma unroll
    for (int k = 0; k < K; ++k) v[k] = wave_sum(v[k]);
    __syncthreads();                       // protect red[] from a previous use
    if (lane == 0) {
#pragma unroll
        for (int k = 0; k < K; ++k) red[k * 16 + wid] = v[k];
    }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < K; ++k) {
        double a = 0.0;
        for (int w = 0; w < nw; ++w) a += red[k * 16 + w];
        v[k] = a;
    }
}

// The same with the wave-level sums taken four at a time (wave_sum4: 21 instructions per four values instead of 4 x 20).
// K is rounded up to a multiple of four internally; red: ((K + 3) / 4) * 4 * 16 doubles.
template <int K>
__device__ __forceinline__ void block_sum_x4(double (&v)[K], double *red) {
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6, nw = blockDim.x >> 6;
    constexpr int G = (K + 3) / 4;
    __syncthreads();                       // protect red[] from a previous use
#pragma unroll
    for (int g = 0; g < G; ++g) {
        const double a0 = v[4 * g], a1 = (4 * g + 1 < K) ? v[4 * g + 1] : 0.0, a2 = (4 * g + 2 < K) ? v[4 * g + 2] : 0.0, a3 = (4 * g + 3 < K) ? v[4 * g + 3] : 0.0;
        const double f = wave_sum4(a0, a2, a1, a3);          // 16-lane rows 0 .. 3 hold the sums of a0, a1, a2, a3
        if ((lane & 15) == 0) red[(4 * g + (lane >> 4)) * 16 + wid] = f;
    }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < K; ++k) {
        double a = 0.0;
        for (int w = 0; w < nw; ++w) a += red[k * 16 + w];
        v[k] = a;
    }
}

__device__ __forceinline__ double block_sum1(double x, double *red) {
    double v[1] = {x};
    block_sum<1>(v, red);
    return v[0];
}

// Lane-strided reducer over a single wave: the `R` concept of logic.h.
struct WaveRed {
    template <class F> __device__ double sum(int m, F f) const {
        double a = 0.0;
        for (int i = threadIdx.x & 63; i < m; i += WAVE) a += f(i);
        return wave_sum(a);
    }
    template <class F> __device__ void sum2(int m, F f, double *s0, double *s1) const {      // two sums in one pass: f(i, a0, a1) adds its terms
        double a0 = 0.0, a1 = 0.0;
        for (int i = threadIdx.x & 63; i < m; i += WAVE) f(i, a0, a1);
        *s0 = wave_sum(a0); *s1 = wave_sum(a1);
    }
    template <class P> __device__ int min_index(int m, P p) const {
        int best = m;
        for (int i = threadIdx.x & 63; i < m; i += WAVE) if (p(i)) { best = i; break; }
        return wave_min_int(best);
    }
    template <class F> __device__ void for_each(int m, F f) const {
        for (int i = threadIdx.x & 63; i < m; i += WAVE) f(i);
    }
    __device__ void sync() const { __syncthreads(); }
};

}  // namespace mopt

// ---- host side --------------------------------------------------------------------------
struct mopt_timer { std::vector<std::pair<hipEvent_t, hipEvent_t>> ev; double ms = 0.0; int64_t launches = 0; };

// outcome of the last deflating tridiagonalisation on a workspace, read back without waiting (k_tridiag.hip)
struct DeflProbe { const void *key = nullptr; int *pin = nullptr; int cap = 0; hipEvent_t ev = nullptr; bool pending = false; int batch = 0, dense_streak = 0, skip_left = 0; };

struct mopt_ctx {
    int device = 0;
    hipStream_t stream = nullptr;
    hipStream_t side = nullptr;              // helper stream for work that is independent of the main chain (fork/join by events)
    hipStream_t side2 = nullptr;             // second helper stream (T factors of the back-transformation)
    hipEvent_t ev_fork4 = nullptr, ev_join4 = nullptr;
    hipStream_t side3 = nullptr;             // third helper stream (divide-and-conquer chain of the eigen-solver)
    hipEvent_t ev_fork5 = nullptr, ev_join5 = nullptr;
    hipEvent_t ev_fork6 = nullptr, ev_join6 = nullptr;   // scalar RFO stage next to the back-transformation (pipeline.hip)
    hipEvent_t ev_fork = nullptr, ev_join = nullptr, ev_fork2 = nullptr, ev_join2 = nullptr, ev_fork3 = nullptr, ev_join3 = nullptr;
    std::string err;
    bool timing = false;
    std::map<std::string, mopt_timer> timers;
    int lds_limit = 65536;
    int num_cu = 256;
    DeflProbe defl;
    bool join_by_events = false;             // eigen-solver: join the divide-and-conquer chain with a stream-level event wait instead of the spinning k_dc_wait
                                             // (set for image-sharded bands with more than one rank: their exchange stream is a fifth busy stream of the
                                             // process, more than the default number of hardware queues, and a kernel that spins for a kernel queued behind
                                             // it on a shared queue would stall until its time-out)
    std::vector<int> last_defl;              // mopt_eigh: where the tridiagonalisation of each matrix of the last call stopped early (0: ran to the end); mopt_debug_last_deflation
    long long n_retries = 0;                 // images solved again after a hand-off time-out (mopt_ctx_timing_read family "tri_retries")
    bool tri_no_persist = false;             // eigen-solver: take the launch-per-step tridiagonalisation (retry after a hand-off time-out)
    // early start of the leading block (k_early.hip): staging + flags (grow-only), and the D array known to hold the sentinel
    void *early_buf = nullptr; size_t early_bytes = 0; const void *early_clean = nullptr;
    // generic scratch arena (grow-only)
    void *arena = nullptr;
    size_t arena_bytes = 0;
};

#define MOPT_HIP_TRY(ctx, call)                                                        \
    do {                                                                               \
        hipError_t e__ = (call);                                                       \
        if (e__ != hipSuccess) {                                                       \
            (ctx)->err = std::string(#call) + ": " + hipGetErrorString(e__);           \
            return MOPT_E_DEVICE;                                                      \
        }                                                                              \
    } while (0)

// Scoped timing of a kernel family on the ctx stream (HIP events; only when ctx->timing).
struct mopt_scope {
    mopt_ctx *c; const char *name; hipEvent_t a = nullptr, b = nullptr; int n; hipStream_t st;
    mopt_scope(mopt_ctx *ctx, const char *nm, int nlaunch = 1, hipStream_t on = nullptr) : c(ctx), name(nm), n(nlaunch), st(on ? on : ctx->stream) {
        if (c->timing) { (void)hipEventCreate(&a); (void)hipEventCreate(&b); (void)hipEventRecord(a, st); }
    }
    ~mopt_scope() {
        if (c->timing) {
            (void)hipEventRecord(b, st);
            auto &t = c->timers[name];
            t.ev.emplace_back(a, b);
            t.launches += n;
        }
    }
};
