// Kernels and C ABI of the callers around the optimizer step:
//   mopt_quadratic_model   Ce = g.s + 1/2 s^T H s                      (trust_radius.py:146-151)
//   mopt_clip_and_move     clip ||s|| <= trust, x_new = (x - s) * bohr2ang   (optimizer.py:789-812)
//   mopt_kabsch            Kabsch alignment with the reference's in-place centring (Utils/calc_tools.py:412-425)
// One workgroup per problem: these run once per optimizer step on O(n^2) / O(N) data.
#include <cstring>

#include "common.hpp"
#include "pipeline.h"

namespace mopt {

// Ce = g.s + 0.5 * s^T H s; wave w takes rows w, w + nw, ...
__global__ void __launch_bounds__(1024) k_quadform(int n, const double *__restrict__ H_, const double *__restrict__ g_,
                                                   const double *__restrict__ s_, double *__restrict__ out) {
    __shared__ double red[2 * 16];
    const int b = blockIdx.x;
    const double *H = H_ + (size_t)b * n * n, *g = g_ + (size_t)b * n, *s = s_ + (size_t)b * n;
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6, nw = blockDim.x >> 6;
    double quad = 0.0, lin = 0.0;
    for (int i = wid; i < n; i += nw) {
        double a = 0.0;
        for (int j = lane; j < n; j += 64) a += H[(size_t)i * n + j] * s[j];
        a = wave_sum(a);
        quad += s[i] * a;                                   // identical on every lane of the wave
    }
    for (int i = threadIdx.x; i < n; i += blockDim.x) lin += g[i] * s[i];
    double v[2] = {lane == 0 ? quad : 0.0, lin};
    block_sum<2>(v, red);
    if (threadIdx.x == 0) out[b] = v[1] + 0.5 * v[0];
}

// step clipped to the trust radius, new geometry in angstrom, step kept in bohr
__global__ void __launch_bounds__(256) k_clip_move(int n, double trust, double bohr2ang, const double *__restrict__ x,
                                                   const double *__restrict__ step, double *__restrict__ new_ang,
                                                   double *__restrict__ move, double *__restrict__ norm_out) {
    __shared__ double red[16];
    double a = 0.0;
    for (int i = threadIdx.x; i < n; i += blockDim.x) a += step[i] * step[i];
    const double nrm = sqrt(block_sum1(a, red));
    const bool clip = nrm > trust;
    for (int i = threadIdx.x; i < n; i += blockDim.x) {
        const double m = clip ? trust * step[i] / nrm : step[i];          // optimizer.py:789-790: trust * v / ||v||
        move[i] = m;
        new_ang[i] = (x[i] - m) * bohr2ang;
    }
    if (threadIdx.x == 0) norm_out[0] = nrm;
}

// 3x3 helpers for the Kabsch rotation (one thread)
__device__ void svd3_rotation(const double Hm[9], double R[9]) {
    // one-sided Jacobi: A = Hm * V with orthogonal columns; sigma_i = ||A[:,i]||, u_i = A[:,i] / sigma_i
    double A[9], V[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1};
    for (int i = 0; i < 9; ++i) A[i] = Hm[i];
    for (int sweep = 0; sweep < 60; ++sweep) {
        int rot = 0;
        for (int p = 0; p < 2; ++p)
            for (int q = p + 1; q < 3; ++q) {
                double al = 0, be = 0, ga = 0;
                for (int k = 0; k < 3; ++k) { al += A[3 * k + p] * A[3 * k + p]; be += A[3 * k + q] * A[3 * k + q]; ga += A[3 * k + p] * A[3 * k + q]; }
                if (ga == 0.0 || fabs(ga) <= 1e-17 * sqrt(al * be)) continue;
                ++rot;
                const double zeta = (be - al) / (2.0 * ga);
                const double t = copysign(1.0, zeta) / (fabs(zeta) + sqrt(1.0 + zeta * zeta));
                const double c = 1.0 / sqrt(1.0 + t * t), s = c * t;
                for (int k = 0; k < 3; ++k) {
                    const double ap = A[3 * k + p], aq = A[3 * k + q];
                    A[3 * k + p] = c * ap - s * aq; A[3 * k + q] = s * ap + c * aq;
                    const double vp = V[3 * k + p], vq = V[3 * k + q];
                    V[3 * k + p] = c * vp - s * vq; V[3 * k + q] = s * vp + c * vq;
                }
            }
        if (!rot) break;
    }
    double sg[3];
    int ord[3] = {0, 1, 2};
    for (int i = 0; i < 3; ++i) sg[i] = sqrt(A[i] * A[i] + A[3 + i] * A[3 + i] + A[6 + i] * A[6 + i]);
    for (int i = 0; i < 2; ++i)
        for (int j = 0; j < 2 - i; ++j)
            if (sg[ord[j]] < sg[ord[j + 1]]) { const int t = ord[j]; ord[j] = ord[j + 1]; ord[j + 1] = t; }
    double U[9], Vs[9];                                       // columns sorted by descending singular value
    const double tiny = 1e-14 * (sg[ord[0]] > 0.0 ? sg[ord[0]] : 1.0);
    int rank = 0;
    for (int c = 0; c < 3; ++c) {
        const int o = ord[c];
        for (int k = 0; k < 3; ++k) Vs[3 * k + c] = V[3 * k + o];
        if (sg[o] > tiny) { for (int k = 0; k < 3; ++k) U[3 * k + c] = A[3 * k + o] / sg[o]; rank = c + 1; }
    }
    if (rank == 0) { for (int i = 0; i < 9; ++i) U[i] = (i % 4 == 0) ? 1.0 : 0.0; rank = 3; }
    if (rank == 1) {                                          // any unit vector orthogonal to u_0
        const double ux = U[0], uy = U[3], uz = U[6];
        double w[3] = {0, 0, 0};
        if (fabs(ux) <= fabs(uy) && fabs(ux) <= fabs(uz)) w[0] = 1.0; else if (fabs(uy) <= fabs(uz)) w[1] = 1.0; else w[2] = 1.0;
        const double d = w[0] * ux + w[1] * uy + w[2] * uz;
        double y0 = w[0] - d * ux, y1 = w[1] - d * uy, y2 = w[2] - d * uz;
        const double nn = sqrt(y0 * y0 + y1 * y1 + y2 * y2);
        U[1] = y0 / nn; U[4] = y1 / nn; U[7] = y2 / nn;
        rank = 2;
    }
    if (rank == 2) {                                          // u_2 = u_0 x u_1
        U[2] = U[3] * U[7] - U[6] * U[4];
        U[5] = U[6] * U[1] - U[0] * U[7];
        U[8] = U[0] * U[4] - U[3] * U[1];
    }
    auto build = [&]() { for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) { double a = 0; for (int c = 0; c < 3; ++c) a += Vs[3 * i + c] * U[3 * j + c]; R[3 * i + j] = a; } };
    build();                                                  // R = V U^T   (calc_tools.py:420)
    const double det = R[0] * (R[4] * R[8] - R[5] * R[7]) - R[1] * (R[3] * R[8] - R[5] * R[6]) + R[2] * (R[3] * R[7] - R[4] * R[6]);
    if (det < 0.0) { for (int k = 0; k < 3; ++k) Vs[3 * k + 2] = -Vs[3 * k + 2]; build(); }      // Vt[-1, :] *= -1  (:421-423)
}

__global__ void __launch_bounds__(256) k_kabsch(int N, const double *__restrict__ P, const double *__restrict__ Q,
                                                double *__restrict__ P_out, double *__restrict__ Q_out) {
    __shared__ double red[16 * 16];
    __shared__ double sR[9];
    // centroids: np.mean of each column (calc_tools.py:414-415)
    double v6[6] = {0, 0, 0, 0, 0, 0};
    for (int a = threadIdx.x; a < N; a += blockDim.x)
        for (int k = 0; k < 3; ++k) { v6[k] += P[3 * a + k]; v6[3 + k] += Q[3 * a + k]; }
    block_sum<6>(v6, red);
    double cP[3], cQ[3];
    for (int k = 0; k < 3; ++k) { cP[k] = v6[k] / N; cQ[k] = v6[3 + k] / N; }
    double h[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
    for (int a = threadIdx.x; a < N; a += blockDim.x) {
        double p[3], q[3];
        for (int k = 0; k < 3; ++k) { p[k] = P[3 * a + k] - cP[k]; q[k] = Q[3 * a + k] - cQ[k]; }
        for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) h[3 * i + j] += p[i] * q[j];     // H = P^T Q
    }
    __syncthreads();
    block_sum<9>(h, red);
    if (threadIdx.x == 0) svd3_rotation(h, sR);
    __syncthreads();
    for (int a = threadIdx.x; a < N; a += blockDim.x) {
        double p[3];
        for (int k = 0; k < 3; ++k) { p[k] = P[3 * a + k] - cP[k]; Q_out[3 * a + k] = Q[3 * a + k] - cQ[k]; }
        for (int i = 0; i < 3; ++i) P_out[3 * a + i] = sR[3 * i] * p[0] + sR[3 * i + 1] * p[1] + sR[3 * i + 2] * p[2];      // (R P^T)^T
    }
}

}  // namespace mopt

using namespace mopt;

namespace {
struct DevBuf {
    void *p = nullptr;
    ~DevBuf() { if (p) hipFree(p); }
    int alloc(mopt_ctx *ctx, size_t bytes) {
        if (hipMalloc(&p, bytes ? bytes : 8) != hipSuccess) { ctx->err = "hipMalloc failed"; p = nullptr; return MOPT_E_NOMEM; }
        return MOPT_OK;
    }
    template <class T> T *as() { return (T *)p; }
};
#define DTRY(x) do { int rc__ = (x); if (rc__ != MOPT_OK) return rc__; } while (0)
#define DH2D(dst, src, bytes) MOPT_HIP_TRY(ctx, hipMemcpyAsync(dst, src, bytes, hipMemcpyHostToDevice, ctx->stream))
#define DD2H(dst, src, bytes) MOPT_HIP_TRY(ctx, hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToHost, ctx->stream))
}  // namespace

extern "C" int mopt_quadratic_model(mopt_ctx *ctx, int n, int batch, const double *H, const double *g, const double *s, double *out) {
    if (!ctx) return MOPT_E_ARG;
    if (n < 1 || n > MOPT_MAX_N || batch < 1 || !H || !g || !s || !out) { ctx->err = "mopt_quadratic_model: bad arguments"; return MOPT_E_ARG; }
    hipSetDevice(ctx->device);
    const size_t nn = (size_t)batch * n * n * 8, nv = (size_t)batch * n * 8;
    DevBuf dH, dg, ds, dout;
    DTRY(dH.alloc(ctx, nn)); DTRY(dg.alloc(ctx, nv)); DTRY(ds.alloc(ctx, nv)); DTRY(dout.alloc(ctx, (size_t)batch * 8));
    DH2D(dH.p, H, nn); DH2D(dg.p, g, nv); DH2D(ds.p, s, nv);
    hipLaunchKernelGGL(k_quadform, dim3(batch), dim3(1024), 0, ctx->stream, n, dH.as<double>(), dg.as<double>(), ds.as<double>(), dout.as<double>());
    MOPT_HIP_TRY(ctx, hipGetLastError());
    DD2H(out, dout.p, (size_t)batch * 8);
    MOPT_HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    return MOPT_OK;
}

extern "C" int mopt_clip_and_move(mopt_ctx *ctx, int n, const double *x_bohr, const double *step, double trust_radius, double bohr2ang,
                                  double *new_geometry_ang, double *move_vector, double *step_norm) {
    if (!ctx) return MOPT_E_ARG;
    if (n < 1 || !x_bohr || !step || !new_geometry_ang || !move_vector) { ctx->err = "mopt_clip_and_move: bad arguments"; return MOPT_E_ARG; }
    hipSetDevice(ctx->device);
    const size_t nv = (size_t)n * 8;
    DevBuf dx, ds, dn, dm, dnorm;
    DTRY(dx.alloc(ctx, nv)); DTRY(ds.alloc(ctx, nv)); DTRY(dn.alloc(ctx, nv)); DTRY(dm.alloc(ctx, nv)); DTRY(dnorm.alloc(ctx, 8));
    DH2D(dx.p, x_bohr, nv); DH2D(ds.p, step, nv);
    hipLaunchKernelGGL(k_clip_move, dim3(1), dim3(256), 0, ctx->stream, n, trust_radius, bohr2ang, dx.as<double>(), ds.as<double>(),
                       dn.as<double>(), dm.as<double>(), dnorm.as<double>());
    MOPT_HIP_TRY(ctx, hipGetLastError());
    DD2H(new_geometry_ang, dn.p, nv); DD2H(move_vector, dm.p, nv);
    double nrm = 0.0;
    DD2H(&nrm, dnorm.p, 8);
    MOPT_HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    if (step_norm) *step_norm = nrm;
    return MOPT_OK;
}

extern "C" int mopt_kabsch(mopt_ctx *ctx, int N, const double *P, const double *Q, double *P_aligned, double *Q_centred) {
    if (!ctx) return MOPT_E_ARG;
    if (N < 1 || !P || !Q || !P_aligned || !Q_centred) { ctx->err = "mopt_kabsch: bad arguments"; return MOPT_E_ARG; }
    hipSetDevice(ctx->device);
    const size_t nv = (size_t)N * 3 * 8;
    DevBuf dP, dQ, dPo, dQo;
    DTRY(dP.alloc(ctx, nv)); DTRY(dQ.alloc(ctx, nv)); DTRY(dPo.alloc(ctx, nv)); DTRY(dQo.alloc(ctx, nv));
    DH2D(dP.p, P, nv); DH2D(dQ.p, Q, nv);
    hipLaunchKernelGGL(k_kabsch, dim3(1), dim3(256), 0, ctx->stream, N, dP.as<double>(), dQ.as<double>(), dPo.as<double>(), dQo.as<double>());
    MOPT_HIP_TRY(ctx, hipGetLastError());
    DD2H(P_aligned, dPo.p, nv); DD2H(Q_centred, dQo.p, nv);
    MOPT_HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    return MOPT_OK;
}
