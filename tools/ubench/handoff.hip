// Developer micro-benchmark: one-way latency of a tagged-word hand-off between two workgroups, same XCD vs different XCDs,
// for the store/load scopes used by k_tri_persist2.   hipcc -O3 --offload-arch=gfx950 tools/ubench/handoff.hip -o tools/ubench/handoff
#include <hip/hip_runtime.h>
#include <cstdio>
typedef unsigned long long u64;

template <int ST, int LD>   // 0 = workgroup scope (sc0), 1 = agent scope (sc1)
__global__ void pingpong(u64 *w, int partner_a, int partner_b, int rounds, long long *out, unsigned *xcc) {
    const int me = blockIdx.x;
    if (me != partner_a && me != partner_b) return;
    if (threadIdx.x != 0) return;
    const int side = (me == partner_b);
    xcc[side] = (unsigned)__builtin_amdgcn_s_getreg((3 << 11) | 20) & 0xF;
    u64 *mine = w + 32 * side, *theirs = w + 32 * (1 - side);
    const long long t0 = wall_clock64();
    for (int r = 1; r <= rounds; ++r) {
        if (side == 0) {
            if (ST) __hip_atomic_store(mine, (u64)r, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); else __hip_atomic_store(mine, (u64)r, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        }
        unsigned spins = 0;
        for (;;) {
            const u64 v = LD ? __hip_atomic_load(theirs, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : __hip_atomic_load(theirs, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            if (v == (u64)r) break;
            if (++spins > 4000000u) { out[2] = 1; return; }
        }
        if (side == 1) {
            if (ST) __hip_atomic_store(mine, (u64)r, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); else __hip_atomic_store(mine, (u64)r, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        }
    }
    if (side == 0) out[0] = wall_clock64() - t0;
}

template <int ST, int LD>
void run(const char *name, int a, int b) {
    u64 *w; long long *out; unsigned *xcc;
    hipMalloc(&w, 4096); hipMalloc(&out, 64); hipMalloc(&xcc, 64);
    hipMemset(w, 0, 4096); hipMemset(out, 0, 64);
    const int rounds = 2000;
    hipLaunchKernelGGL((pingpong<ST, LD>), dim3(256), dim3(64), 0, 0, w, a, b, rounds, out, xcc);
    hipDeviceSynchronize();
    long long h[3]; unsigned hx[2];
    hipMemcpy(h, out, 24, hipMemcpyDeviceToHost); hipMemcpy(hx, xcc, 8, hipMemcpyDeviceToHost);
    printf("%-34s workgroups %3d (XCC %u) <-> %3d (XCC %u): %s one-way %.2f us\n", name, a, hx[0], b, hx[1], h[2] ? "TIMEOUT" : "ok", h[0] * 0.01 / rounds / 2.0);
    hipFree(w); hipFree(out); hipFree(xcc);
}

int main() {
    run<1, 1>("store sc1 / load sc1", 0, 8);
    run<1, 1>("store sc1 / load sc1", 0, 1);
    run<0, 1>("store sc0 / load sc1", 0, 8);
    run<0, 1>("store sc0 / load sc1", 0, 1);
    run<0, 0>("store sc0 / load sc0", 0, 8);
    run<1, 0>("store sc1 / load sc0", 0, 8);
    return 0;
}
