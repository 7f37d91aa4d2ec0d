// Developer driver for k_tri_tail (k_tritail.hip): times the single-compute-unit tridiagonalisation on random symmetric matrices
// and, built with -DMOPT_TAIL_PROF, prints where wave 0 spends a step.  Checks D, E against a host Householder reduction.
//   hipcc -O3 -std=c++17 --offload-arch=gfx950 [-DMOPT_TAIL_PROF] -Iinclude tools/ubench/tail_main.hip multioptpy_amd/csrc/k_tritail.hip -o tools/ubench/tail_bench
//   tools/ubench/tail_bench [n=192] [batch=32] [reps=20]
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include "../../multioptpy_amd/csrc/common.hpp"
int mopt_launch_tri_tail(mopt_ctx *ctx, hipStream_t st, int n, int K, int batch, double *d_A, double *d_D, double *d_E, double *d_TAU, long long *d_prof);

static void host_tridiag(int n, std::vector<double> A, std::vector<double> &D, std::vector<double> &E) {      // forward Householder, rows
    D.assign(n, 0.0); E.assign(n, 0.0);
    std::vector<double> v(n), p(n), w(n);
    for (int k = 0; k < n - 1; ++k) {
        D[k] = A[k * n + k];
        double alpha = A[k * n + k + 1], xn2 = 0.0;
        for (int j = k + 2; j < n; ++j) xn2 += A[k * n + j] * A[k * n + j];
        if (k > n - 3 || xn2 == 0.0) { E[k] = alpha; continue; }
        const double bt = -copysign(sqrt(alpha * alpha + xn2), alpha), tau = (bt - alpha) / bt, sc = 1.0 / (alpha - bt);
        E[k] = bt;
        for (int j = 0; j < n; ++j) v[j] = j < k + 1 ? 0.0 : (j == k + 1 ? 1.0 : A[k * n + j] * sc);
        double pv = 0.0;
        for (int i = k + 1; i < n; ++i) { double a = 0.0; for (int j = k + 1; j < n; ++j) a += A[i * n + j] * v[j]; p[i] = tau * a; pv += p[i] * v[i]; }
        for (int i = k + 1; i < n; ++i) w[i] = p[i] - 0.5 * tau * pv * v[i];
        for (int i = k + 1; i < n; ++i) for (int j = k + 1; j < n; ++j) A[i * n + j] -= v[i] * w[j] + w[i] * v[j];
    }
    D[n - 1] = A[(n - 1) * n + n - 1];
}

int main(int argc, char **argv) {
    const int n = argc > 1 ? atoi(argv[1]) : 192, B = argc > 2 ? atoi(argv[2]) : 32, reps = argc > 3 ? atoi(argv[3]) : 20;
    std::vector<double> A((size_t)B * n * n);
    srand(1);
    for (int b = 0; b < B; ++b)
        for (int i = 0; i < n; ++i)
            for (int j = 0; j <= i; ++j) { const double x = rand() / (double)RAND_MAX - 0.5; A[((size_t)b * n + i) * n + j] = x; A[((size_t)b * n + j) * n + i] = x; }
    double *dA, *dA0, *dD, *dE, *dT; long long *dP;
    hipMalloc(&dA, A.size() * 8); hipMalloc(&dA0, A.size() * 8); hipMalloc(&dD, (size_t)B * n * 8); hipMalloc(&dE, (size_t)B * n * 8); hipMalloc(&dT, (size_t)B * n * 8);
    hipMalloc(&dP, (size_t)(B + 16) * 64); hipMemset(dP, 0, (size_t)(B + 16) * 64);
    hipMemcpy(dA0, A.data(), A.size() * 8, hipMemcpyHostToDevice);
    mopt_ctx ctx;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    float best = 1e9f;
    for (int r = 0; r < reps; ++r) {
        hipMemcpy(dA, dA0, A.size() * 8, hipMemcpyDeviceToDevice);
        hipDeviceSynchronize();
        hipEventRecord(e0, 0);
        if (mopt_launch_tri_tail(&ctx, 0, n, -1, B, dA, dD, dE, dT, dP) != 0) { printf("launch failed: %s\n", ctx.err.c_str()); return 1; }
        hipEventRecord(e1, 0); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1); if (ms < best) best = ms;
    }
    std::vector<double> D((size_t)B * n), E((size_t)B * n), Dh, Eh;
    hipMemcpy(D.data(), dD, D.size() * 8, hipMemcpyDeviceToHost); hipMemcpy(E.data(), dE, E.size() * 8, hipMemcpyDeviceToHost);
    double err = 0.0;
    for (int b = 0; b < B; b += (B > 1 ? B - 1 : 1)) {
        host_tridiag(n, std::vector<double>(A.begin() + (size_t)b * n * n, A.begin() + (size_t)(b + 1) * n * n), Dh, Eh);
        for (int i = 0; i < n; ++i) { err = fmax(err, fabs(Dh[i] - D[(size_t)b * n + i])); if (i < n - 1) err = fmax(err, fabs(fabs(Eh[i]) - fabs(E[(size_t)b * n + i]))); }
    }
    printf("n=%d batch=%d: %.4f ms (%.3f us per step), max |D,E - host| = %.2e\n", n, B, best, 1e3 * best / (n - 1), err);
    std::vector<long long> P((size_t)(B + 16) * 8);
    hipMemcpy(P.data(), dP, P.size() * 8, hipMemcpyDeviceToHost);
    if (P[0] | P[1] | P[2]) {
        const char *nm[6] = {"a: w from p", "b: reflector (owner only)", "update sweep", "barrier 1", "product sweep", "barrier 2"};
        double tot = 0; for (int k = 0; k < 6; ++k) tot += (double)P[k];
        for (int k = 0; k < 6; ++k) printf("  %-28s %8.0f clk per step (%.1f %%)\n", nm[k], (double)P[k] / (n - 1), 100.0 * P[k] / tot);
        printf("  total %.0f clk per step (wave 0 of matrix 0; shader clock = %.0f MHz if the step time above is its sum)\n", tot / (n - 1), tot / (n - 1) / (1e3 * best / (n - 1)));
    }
    if (P[(size_t)B * 8]) {
        printf("  step l = 5 of matrix 0, shader clocks relative to the earliest start; columns: start, a done, b done, update done, barrier 1 passed, products done, barrier 2 passed\n");
        long long t0 = P[(size_t)B * 8];
        for (int w = 0; w < 16; ++w) if (P[(size_t)(B + w) * 8] && P[(size_t)(B + w) * 8] < t0) t0 = P[(size_t)(B + w) * 8];
        for (int w = 0; w < 16; ++w) {
            if (!P[(size_t)(B + w) * 8]) continue;
            printf("   wave %2d:", w);
            for (int k = 0; k < 7; ++k) printf(" %6lld", P[(size_t)(B + w) * 8 + k] - t0);
            printf("\n");
        }
    }
    return 0;
}
