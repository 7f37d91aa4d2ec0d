// Developer micro-benchmark: issue cost of FP64 vector instructions on gfx950 (cycles per wave64 instruction), one wave per SIMD
// and four waves per SIMD.   hipcc -O3 --offload-arch=gfx950 tools/ubench/valu_f64.hip -o tools/ubench/valu_f64 && tools/ubench/valu_f64
#include <hip/hip_runtime.h>
#include <cstdio>

template <int MODE>
__global__ void k(double *out, long long *cyc, int iters) {
    double a[8], x = 1.0000001 + threadIdx.x * 1e-9, y = 0.9999999;
    float fa[8], fx = 1.0000001f, fy = 0.999f;
    for (int i = 0; i < 8; ++i) { a[i] = i + threadIdx.x; fa[i] = i; }
    const long long t0 = clock64();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                if (MODE == 0) a[i] = fma(a[i], x, y);            // 8 independent f64 FMA chains
                if (MODE == 1) fa[i] = fmaf(fa[i], fx, fy);       // f32
                if (MODE == 2) a[i] = a[i] + x;                   // f64 add
                if (MODE == 3) a[i] = a[i] * x;                   // f64 mul
            }
            if (MODE == 4) a[0] = fma(a[0], x, y);                // one dependent f64 FMA chain
        }
    }
    const long long t1 = clock64();
    double s = 0; float fs = 0;
    for (int i = 0; i < 8; ++i) { s += a[i]; fs += fa[i]; }
    out[blockIdx.x * blockDim.x + threadIdx.x] = s + fs;
    if (threadIdx.x == 0 && blockIdx.x == 0) cyc[0] = t1 - t0;
}

int main() {
    double *out; long long *cyc;
    hipMalloc(&out, 1 << 24); hipMalloc(&cyc, 64);
    const int iters = 2000;
    const char *names[5] = {"v_fma_f64 (8 independent chains)", "v_fma_f32", "v_add_f64", "v_mul_f64", "v_fma_f64 dependent chain"};
    for (int threads : {256, 1024}) {
        for (int mode = 0; mode < 5; ++mode) {
            long long h = 0;
            for (int rep = 0; rep < 2; ++rep) {
                if (mode == 0) hipLaunchKernelGGL(k<0>, dim3(256), dim3(threads), 0, 0, out, cyc, iters);
                if (mode == 1) hipLaunchKernelGGL(k<1>, dim3(256), dim3(threads), 0, 0, out, cyc, iters);
                if (mode == 2) hipLaunchKernelGGL(k<2>, dim3(256), dim3(threads), 0, 0, out, cyc, iters);
                if (mode == 3) hipLaunchKernelGGL(k<3>, dim3(256), dim3(threads), 0, 0, out, cyc, iters);
                if (mode == 4) hipLaunchKernelGGL(k<4>, dim3(256), dim3(threads), 0, 0, out, cyc, iters);
                hipDeviceSynchronize();
                hipMemcpy(&h, cyc, 8, hipMemcpyDeviceToHost);
            }
            const double n_instr = (mode == 4) ? iters * 4.0 : iters * 32.0;
            printf("%4d threads/CU (%d wave(s)/SIMD)  %-36s %6.2f cycles per instruction of one wave\n", threads, threads / 256, names[mode], h / n_instr);
        }
    }
    return 0;
}
