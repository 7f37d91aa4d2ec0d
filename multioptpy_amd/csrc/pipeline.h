// Internal (non-ABI) interfaces between the translation units of libmopt_hip.so.
#pragma once
#include "common.hpp"
#include "logic.h"

#define MOPT_MAX_N 1500            /* 3N; bounded by the LDS staging of the 6 TR/rot vectors (12 n doubles) */
#define MOPT_DEFAULT_SWEEPS 16     /* Jacobi sweeps launched per solve; converged matrices skip the rest */

namespace mopt {
constexpr int JAC_MAX_SWEEPS = 30;
}

// scratch of the reduction-route eigen-solver (k_tridiag.hip)
// doubles per matrix behind the n^2 of Z2: the 16 x 16 T factors of the blocked back-transformation, ceil((n - 2) / 16) of them
#define MOPT_TRI_T_DOUBLES(n) ((size_t)16 * ((size_t)(n) + 16))
struct TriScratch { double *D, *E, *TAU, *WT, *onenrm, *Z2; int *bstart, *bend, *cstart, *ncl, *fail, *defl; };      // defl: [batch] where the tridiagonalisation stopped early (0: ran to the end)

// Device workspace of one batched RSIRFO.run (all pointers are device pointers).
struct RsirfoWork {
    mopt_ctx *ctx = nullptr;
    int n = 0, batch = 0;
    void *block = nullptr;        // single allocation
    bool owns_H = false;
    double *H = nullptr, *Bias = nullptr, *Hinpl = nullptr;       // [B][n][n]
    double *A = nullptr, *Gt = nullptr, *Vt = nullptr;            // [B][n][n]
    double *x = nullptr, *bg = nullptr, *g = nullptr, *px = nullptr, *pg = nullptr;   // [B][n]
    double *s = nullptr, *y = nullptr, *u = nullptr, *Hy = nullptr, *yd = nullptr, *Av = nullptr;
    double *gp = nullptr, *Q = nullptr, *W6 = nullptr, *eigw = nullptr, *gt_all = nullptr, *coefout = nullptr, *step = nullptr;
    double *be = nullptr, *sigma = nullptr, *info = nullptr;
    int *perm = nullptr, *einfo = nullptr, *rotcount = nullptr, *rules = nullptr, *do_update = nullptr;
    int *sticky = nullptr;        // [1] failure bits (4: solver failure, 8: hand-off time-out) OR-ed by every solve on this workspace, cleared by the host only
    mopt::UpdateCoef *coef = nullptr;
    mopt_rsirfo_params *params = nullptr;
    mopt_rsirfo_state *states = nullptr;
    TriScratch tri = {};
    int defl_hint = -1;           // tridiagonalisation with the deflation test: 0 never (the caller knows the matrices are dense), 1 / -1 yes, unless the
                                  // outcomes of the last solves say otherwise (DeflProbe, k_tridiag.hip)
    // debug mode (mopt_debug_poison): every array of the block is preceded and followed by a guard gap filled with MOPT_CANARY_BYTE and is
    // itself filled with 0xFF bytes (a NaN as a double, -1 as an int) -- an out-of-bounds write damages a gap, a read of memory nobody
    // wrote shows up as NaN in the results.  check_canaries returns MOPT_E_STATE and names the array in front of the first damaged gap.
    size_t total_bytes = 0;
    std::vector<std::pair<size_t, size_t>> gaps;             // (offset, bytes) of the guard gaps inside `block`
    std::vector<const char *> gap_names;                     // array in front of each gap
    int alloc(mopt_ctx *c, int n_, int batch_, bool own_hessian, bool with_bias, bool with_inplace);
    int check_canaries();
    ~RsirfoWork();
};
#define MOPT_CANARY_BYTE 0xA5
#define MOPT_CANARY_GAP 1024     /* bytes; a multiple of the 256-byte array alignment */
bool mopt_debug_poison_on();
// guard-gap check of an arbitrary device block (band.hip's own block uses it too)
int mopt_check_gaps(mopt_ctx *ctx, const void *block, const std::vector<std::pair<size_t, size_t>> &gaps, const std::vector<const char *> &names, const char *what);

// Band path, fused form (k_fused.hip): Ayala term gamma t t^T, quasi-Newton update and TR/rot projection in three launches.
namespace mopt { struct FusedBand { const double *gamma; const double *tau; int first; int upd_all; bool clear_for_eigh; }; }
/* upd_all: every image has a previous step (no per-image flags); clear_for_eigh: k_fused_final also zeroes the eigen-solver's failure flags and
   exchange area (set by the pipeline when the reduction route follows -- on the Jacobi route Vt carries the warm-start vectors) */
int mopt_launch_fused_update_project(mopt_ctx *ctx, RsirfoWork &wk, int n, int batch, const mopt::FusedBand &fb);

// update (optional) -> TR/rot projections -> eigh -> RS-I-RFO step.  Hsym may alias wk.H (in-place).
// fb != null: the caller guarantees bitwise symmetric Hessians in wk.H (= Hsym), no bias, no flowchart rule; the Ayala term is then
// applied here and not by the caller.
int mopt_rsirfo_pipeline(mopt_ctx *ctx, RsirfoWork &wk, int n, int batch, bool any_update, bool any_flowchart,
                         double *d_Hinpl, double *d_Hsym, bool warm_eigh = false, const mopt::FusedBand *fb = nullptr);

// One-workgroup form of the same step for 9 <= n <= 72 (k_small.hip); taken by mopt_rsirfo_pipeline unless MOPT_SMALL=0 or an eigen-solver
// route is forced (MOPT_EIGH_ALGO).
bool mopt_small_eligible(int n);
int mopt_launch_small_rsirfo(mopt_ctx *ctx, RsirfoWork &wk, int n, int batch, bool any_update, bool any_flowchart, double *d_Hinpl, double *d_Hsym,
                             const mopt::FusedBand *fb = nullptr);
// The same with the inputs and outputs named one by one: any of them may be host memory the device can address (hipHostMalloc), which is how
// mopt_rsirfo_run avoids its copies for these sizes.  done: counter incremented once per geometry behind its last store (system scope), or null.
struct SmallStepIO {
    const mopt_rsirfo_params *params; mopt_rsirfo_state *states;
    const double *x, *bg, *g, *px, *pg, *be;
    const int *rules, *do_update;
    const double *Hin, *Bias;
    double *Hsym, *Hinpl, *step;
    unsigned long long *done;
    const double *gamma = nullptr, *tau = nullptr;   // band path: Ayala term gamma_b t t^T added to the stored Hessian first (FusedBand)
    int tau_first = 0;
};
int mopt_launch_small_rsirfo_io(mopt_ctx *ctx, RsirfoWork &wk, int n, int batch, bool any_update, bool any_flowchart, const SmallStepIO &io);
// EnhancedRSPRFO.run for one small geometry in one launch (k_small_rsprfo); flags: [0] Hessian updated, [1] Vt row of the tracked TS mode or -1,
// [2] eigen-solver status word, [3] 0 done / 2 "the candidate's norm bound exceeds 1e6: repeat through the general path".
struct SmallPrfoIO {
    int n, has_prev_step, do_update, have_ts;
    double B_e, pre_move_norm;
    const mopt_rsprfo_params *params; mopt_rsprfo_state *state;
    const double *x, *bg, *px, *pg, *prevg, *prevs, *ts, *Hin, *Bias;
    double *Hout, *step, *tsout;
    int *flags;
    unsigned long long *done;
};
int mopt_launch_small_rsprfo(mopt_ctx *ctx, RsirfoWork &wk, const SmallPrfoIO &io);

// Images whose eigen-solve reported a hand-off time-out of the persistent tridiagonalisation (einfo bit 8; k_rfo_scalar leaves their
// state untouched): A is formed again from the (already updated) Hessian and the image is solved with one launch per Householder
// step, then the RS-I-RFO stage runs for it.  Synchronises the stream.  wk.x / bg / ... must point at the step's inputs.
int mopt_rsirfo_retry_timeouts(mopt_ctx *ctx, RsirfoWork &wk, int n, int batch, const double *d_Hsym, bool have_coef, int *nretried);

// launchers (defined next to their kernels)
int mopt_launch_hess_update(mopt_ctx *ctx, int n, int batch, const int *d_rules, const int *d_do_update, bool any_flowchart,
                            const double *d_x, const double *d_px, const double *d_g, const double *d_pg,
                            double *d_s, double *d_y, double *d_u, double *d_Hy, double *d_yd, double *d_A,
                            mopt::UpdateCoef *d_coef, double *d_info, const double *d_H, double *d_Hinpl, double *d_Hsym,
                            bool have_sy, bool guard_sy = true);
int mopt_launch_trrot_basis_side(mopt_ctx *ctx, int n, int batch, const double *d_X, double *d_Q);
int mopt_launch_project_hess(mopt_ctx *ctx, int n, int batch, const double *d_H, const double *d_Bias, const double *d_X,
                             double *d_Q, double *d_W, double *d_A, bool basis_ready = false);
// variant 0: RSIRFO._project_grad_tr_rot; 1: EnhancedRSPRFO._project_grad_tr_rot (no projection below 3 atoms, QR columns
// with |R_jj| <= 1e-10 dropped)
int mopt_launch_project_grad(mopt_ctx *ctx, int n, int batch, const double *d_g, const double *d_X, double *d_gp, int variant = 0,
                             bool side_stream = false);
int mopt_launch_matvec(mopt_ctx *ctx, int n, int batch, const double *d_H, const double *d_v, double *d_out);
int mopt_launch_sym_add(mopt_ctx *ctx, int n, int batch, const double *d_H, const double *d_Bias, double *d_A);
// AFIR bias potential (k_afir.hip): energy, gradient, Hessian on the device; d_work: 2 + 6 N doubles
int mopt_launch_afir(mopt_ctx *ctx, int N, const double *d_x, const double *d_cov, int n1, const int *d_f1, int n2, const int *d_f2, double alpha,
                     double *d_work, double *d_E, double *d_grad, double *d_H);
// FischerD3ApproxHessianOld.main for a batch of geometries (k_modelhess.hip); d3 is a HOST array {s6, s8, a1, a2}
int mopt_launch_model_hessian(mopt_ctx *ctx, int kind, int N, int batch, const double *d_X, const double *d_par, const double *d3, double *d_H, double *d_Q,
                              double *d_W, double *d_A, bool project);
// Lindh2007D3 model Hessian and the `clip` modifier of ApproxHessian.main (k_lindh.hip)
int mopt_lindh2007d3_full(mopt_ctx *ctx, int N, int batch, const double *d_X, const double *d_par, const double *d3, const double *d_g, RsirfoWork &wk,
                          double *d_out);
int mopt_model_hessian_clip(mopt_ctx *ctx, int n, int batch, RsirfoWork &wk, double *d_H);
int mopt_jacobi_block_width(const mopt_ctx *ctx, int n);
int mopt_launch_eigh(mopt_ctx *ctx, int n, int batch, const double *d_A, double *d_sigma, double *d_Gt, double *d_Vt,
                     int *d_rotcount, double *d_W, int *d_perm, int *d_info, int max_sweeps, bool warm);
int mopt_launch_dgemm(mopt_ctx *ctx, int n, int batch, const double *d_A, const double *d_B, const double *d_addscale, double *d_C);
// Optional hook of the reduction route: the caller's scalar stage runs NEXT TO the back-transformation V = Q Z.  With g~ = V^T g =
// Z^T (Q^T g) the stage needs only the eigenvalues and the eigenvectors Z of the tridiagonal form (k_applyq.hip):
//   helper stream 2:  T factors -> gq = Q^T gp            (gp must be complete when ev_gp has fired)
//   main stream:      ... eigenpairs of T -> sort -> V = Q Z written to `vout` (Z stays in d_Vt) -> [join]
//   helper stream 1:                          [gq ready] -> front(stream, Z, gq) -> scalar(stream)
// `used` tells the caller whether the solver took the hook (it does not for sizes without the blocked back-transformation); the
// eigenvectors are then in `vout`, not in d_Vt.
struct EighRfoHook {
    const double *gp = nullptr; double *gq = nullptr; hipEvent_t ev_gp = nullptr;
    double *vout = nullptr;
    bool pre_cleared = false;      // the caller has zeroed d_fail and the first 8 n + 64 64-bit words of every matrix' slot of d_Vt (exchange area of the persistent kernel)
    void *user = nullptr;
    int (*front)(mopt_ctx *, void *user, hipStream_t st, const double *d_Zt, const double *d_gq) = nullptr;
    int (*scalar)(mopt_ctx *, void *user, hipStream_t st) = nullptr;
    bool used = false;
};
// early start of the leading block's eigen-solve next to the tridiagonalisation (k_early.hip)
#define MOPT_EARLY_STAGE (32 + 32 * 32)   /* doubles per matrix: eigenvalues, eigenvector table (column-major, stride 32) */
int mopt_launch_tri_early(mopt_ctx *ctx, hipStream_t st, int n, int batch, const double *d_D, const double *d_E, const void *d_xch, size_t xch_stride_u64,
                          double *d_stage, int *d_early);
bool mopt_apply_qt_supported(int n);
int mopt_launch_apply_qt(mopt_ctx *ctx, hipStream_t st, int n, int batch, const double *d_A, const double *d_T, const double *d_in, double *d_out);
int mopt_launch_eigh_tridiag(mopt_ctx *ctx, int n, int batch, double *d_A, double *d_Vt, double *d_D, double *d_E, double *d_TAU,
                             double *d_WT, double *d_onenrm, int *d_bstart, int *d_bend, int *d_cstart, int *d_ncl, int *d_fail,
                             double *d_W, int *d_perm, int *d_info, double *d_Z2, EighRfoHook *hook = nullptr, int *d_sticky = nullptr, int *d_defl = nullptr, int defl_hint = -1);
// algorithm switch: 0 = block Jacobi (k_eigh.hip), 1 = tridiagonal reduction route (k_tridiag.hip); env MOPT_EIGH_ALGO overrides
int mopt_eigh_auto(mopt_ctx *ctx, RsirfoWork &wk, int n, int batch, bool warm, bool preserve_A = true, EighRfoHook *hook = nullptr);
int mopt_eigh_algo(int n);   // 1 = reduction route (k_tridiag.hip), 0 = one-sided Jacobi (k_eigh.hip)
int mopt_launch_eig_gather(mopt_ctx *ctx, int n, int batch, const double *d_Vt, const int *d_perm, double *d_V);
// the three kernels of mopt_launch_rfo one by one (pipeline.hip overlaps the middle one with the back-transformation)
int mopt_launch_rfo_scalar(mopt_ctx *ctx, hipStream_t st, int n, int batch, const mopt_rsirfo_params *d_params, mopt_rsirfo_state *d_states,
                           const double *d_W, const int *d_perm, const int *d_einfo, const double *d_gt_all, const double *d_Bg, const double *d_Be,
                           const mopt::UpdateCoef *d_coef, double *d_coefout, const int *d_mf = nullptr, int nmf = 0);
int mopt_launch_vtg(mopt_ctx *ctx, int n, int batch, const double *d_Vt, const double *d_vec, const int *d_einfo, double *d_out,
                    const double *d_fallback = nullptr, hipStream_t st = nullptr);
int mopt_launch_backtransform(mopt_ctx *ctx, int n, int batch, const double *d_Vt, const double *d_coef, const int *d_einfo, double *d_out);
int mopt_launch_rfo(mopt_ctx *ctx, int n, int batch, const mopt_rsirfo_params *d_params, mopt_rsirfo_state *d_states,
                    const double *d_W, const int *d_perm, const int *d_einfo, const double *d_Vt, const double *d_gp,
                    const double *d_Bg, const double *d_Be, const mopt::UpdateCoef *d_coef, double *d_gt_all,
                    double *d_coefout, double *d_step, const int *d_mf = nullptr, int nmf = 0);
int mopt_launch_neb_force_var(mopt_ctx *ctx, int variant, int I, int N, int first, int count, const double *d_X, const double *d_E,
                              const double *d_G, int ci_active, double *d_F);
int mopt_launch_norm_inf(mopt_ctx *ctx, int n, int batch, const double *d_A, double *d_sigma);   // sigma = 1.0625 * ||A||_inf
int mopt_launch_bneb_force(mopt_ctx *ctx, int I, int N, int first, int count, const double *d_X, const double *d_E,
                           const double *d_G, int ci_active, double *d_F, double *d_tau, double *d_gamma = nullptr);
