/*
 * mopt_hip.h -- C ABI of libmopt_hip.so: MI355X (gfx950) kernels for the MultiOptPy
 * optimizer-step hot path.
 *
 * The reference (ss0832/MultiOptPy) is pure Python and has no FFI; its boundary for this
 * path is the duck-typed Optimizer plugin surface.  Each entry point
 * below names the reference interface it stands in for (file:line relative to
 * /root/reference).  multioptpy_amd/ binds these with ctypes and re-exposes the
 * reference's own class/method names on top.
 *
 * Conventions: extern "C"; every function returns 0 on success and a negative MOPT_E_*
 * code otherwise (mopt_last_error gives the text); no exceptions cross the boundary;
 * all floating point data is IEEE double, C-contiguous row-major; n = 3*N; host
 * pointers unless the name says "dev"; one mopt_ctx per device, calls on one ctx are
 * stream-ordered and must not be issued concurrently from several threads.
 */
#ifndef MOPT_HIP_H
#define MOPT_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MOPT_ABI_VERSION 1

enum {
    MOPT_OK = 0,
    MOPT_E_ARG = -1,      /* bad argument (null pointer, size out of range, unknown rule) */
    MOPT_E_DEVICE = -2,   /* HIP runtime error (text in mopt_last_error) */
    MOPT_E_NOMEM = -3,
    MOPT_E_STATE = -4,    /* call sequence error (e.g. Hessian not set) */
    MOPT_E_NUMERIC = -5   /* the eigen-solver reported failure for at least one matrix (outputs are written but must not be used) */
};

/* Hessian update rules = the dispatch keys of multioptpy/Optimizer/rsirfo.py:208-251 in
 * priority order; MOPT_RULE_FLOWCHART is also the fall-back for unknown names. */
enum {
    MOPT_RULE_FLOWCHART = 0,
    MOPT_RULE_BLOCK_CFD_FSB_DD, MOPT_RULE_BLOCK_CFD_FSB_WEIGHTED, MOPT_RULE_BLOCK_CFD_FSB,
    MOPT_RULE_BLOCK_CFD_BOFILL_WEIGHTED, MOPT_RULE_BLOCK_CFD_BOFILL,
    MOPT_RULE_BLOCK_BFGS_DD, MOPT_RULE_BLOCK_BFGS,
    MOPT_RULE_BLOCK_FSB_DD, MOPT_RULE_BLOCK_FSB_WEIGHTED, MOPT_RULE_BLOCK_FSB,
    MOPT_RULE_BLOCK_BOFILL_WEIGHTED, MOPT_RULE_BLOCK_BOFILL,
    MOPT_RULE_BFGS_DD, MOPT_RULE_BFGS, MOPT_RULE_SR1, MOPT_RULE_PCFD_BOFILL,
    MOPT_RULE_CFD_FSB_DD, MOPT_RULE_CFD_FSB, MOPT_RULE_CFD_BOFILL,
    MOPT_RULE_FSB_DD, MOPT_RULE_FSB, MOPT_RULE_BOFILL, MOPT_RULE_PSB, MOPT_RULE_MSP,
    MOPT_RULE_COUNT
};

typedef struct mopt_ctx mopt_ctx;

/* ---- context ------------------------------------------------------------------------ */
int mopt_abi_version(void);
int mopt_ctx_create(int device, mopt_ctx **out);
int mopt_ctx_destroy(mopt_ctx *ctx);
const char *mopt_last_error(const mopt_ctx *ctx);
int mopt_ctx_sync(mopt_ctx *ctx);
/* test hook: `blocks` workgroups holding `lds_bytes` of LDS each stay resident for about `ms` milliseconds on a stream of their
 * own (a co-tenant for the persistent kernels of the eigen-solver); returns immediately. */
int mopt_debug_occupy(mopt_ctx *ctx, double ms, int lds_bytes, int blocks);
/* debug mode for workspaces allocated from now on (also MOPT_DEBUG_POISON=1 in the environment): every array of a workspace is filled
 * with 0xFF bytes (NaN / -1) and separated from its neighbours by guard gaps holding a canary byte; the entry points that end in a
 * synchronisation (mopt_rsirfo_run, mopt_rsprfo_run, mopt_crsirfo_run, mopt_band_download) then check the gaps and return MOPT_E_STATE,
 * naming the array in front of a damaged gap.  A read of memory nobody wrote shows up as NaN in the results. */
int mopt_debug_poison(int enable);
/* test aid: row at which the tridiagonalisation of each matrix of the last mopt_eigh call of this context stopped on a diagonal
 * trailing block (0: it ran to the end, or the route has no such test); returns the number of entries written (<= cap) or a status < 0.
 * No counterpart in the reference (numpy.linalg.eigh, rsirfo.py:606). */
int mopt_debug_last_deflation(mopt_ctx *ctx, int *out, int cap);
/* milliseconds spent in the named kernel family since the last reset, measured with HIP
 * events on the ctx stream when timing is enabled (bench.py roofline leg). */
int mopt_ctx_timing(mopt_ctx *ctx, int enable);
int mopt_ctx_timing_read(mopt_ctx *ctx, const char *family, double *ms, int64_t *launches);

/* ---- building blocks (batched over `batch` independent matrices) ---------------------- */

/* Quasi-Newton update.  Stands in for ModelHessianUpdate.* / BlockHessianUpdate.block_*
 * (multioptpy/Optimizer/hessian_update.py:163-433, block_hessian_update.py:443-709) as
 * driven by RSIRFO.update_hessian (rsirfo.py:1316-1372), including its guards
 * (|s|,|y| < 1e-10 or s.y <= 0 -> no update).
 *   H        in : batch x n x n
 *   s, y     in : batch x n        (displacement, gradient difference)
 *   H_inplace out: H + delta  (what the caller's aliased array holds after `+=`), may be NULL
 *   H_sym     out: 0.5 * (X + X^T) of the above (the optimizer's new self.hessian)
 *   info      out: batch x 8 doubles {applied, s.y, s.s, y.y, s.Hs, phi2/weight, branch, 0}, may be NULL */
int mopt_hess_update(mopt_ctx *ctx, int rule, int n, int batch, const double *H, const double *s,
                     const double *y, double *H_inplace, double *H_sym, double *info);
/* delta-H of one rule without RSIRFO.update_hessian's guards: the return value of ModelHessianUpdate.<rule>_hessian_update
 * (hessian_update.py:163-433) as the IRC integrators use it (IRC/lqa.py:286, IRC/hpc.py:421). */
int mopt_hess_delta(mopt_ctx *ctx, int rule, int n, int batch, const double *H, const double *s, const double *y, double *dH);

/* Calculationtools.project_out_hess_tr_and_rot_for_coord (multioptpy/Utils/calc_tools.py:249-316):
 * Hp = sym(P^T H P), P = I - sum v v^T over the Gram-Schmidt'ed translation/rotation vectors. */
int mopt_project_trrot_hess(mopt_ctx *ctx, int n, int batch, const double *H, const double *X, double *Hp);

/* RSIRFO._project_grad_tr_rot (multioptpy/Optimizer/rsirfo.py:128-190): g - Q Q^T g, Q = reduced
 * Householder QR of the (n,6) translation/rotation matrix. */
int mopt_project_trrot_grad(mopt_ctx *ctx, int n, int batch, const double *g, const double *X, double *gp);

/* numpy.linalg.eigh as called at rsirfo.py:606,626,652: A symmetric (batch x n x n),
 * W ascending (batch x n), V (batch x n x n) with eigenvectors in COLUMNS (V[:, i]).
 * sweeps_out (may be NULL) receives the Jacobi sweep count per matrix. */
int mopt_eigh(mopt_ctx *ctx, int n, int batch, const double *A, double *W, double *V, int *sweeps_out);

/* C = A B for batch x n x n row-major matrices on the FP64 matrix cores (np.dot call sites such as
 * rsirfo.py:423,927,940; used internally to warm-start the eigen-solver from the previous eigenvectors). */
int mopt_dgemm(mopt_ctx *ctx, int n, int batch, const double *A, const double *B, double *C);

/* ---- nebmain's force-only optimizers (no Hessians) ------------------------------------------ */

/* TR_NEB.TR_calc (multioptpy/Optimizer/trust_radius_neb.py:17-98): per image, clip `delta` to half the distance to the
 * neighbour it moves towards, freeze it when force . delta < 0; end images: min(0.5, |delta|) or frozen when fixed.
 * X, F ("total_force_list"), delta, move: I x N x 3 (bohr). */
int mopt_neb_tr_calc(mopt_ctx *ctx, int I, int N, const double *X, const double *F, const double *delta, int fix_init,
                     int fix_end, double *move);

/* FIREOptimizer.optimize (multioptpy/Optimizer/fire_neb.py:38-92).  vel = total_velocity, pre_vel = pre_total_velocity (NULL
 * when the reference's list is empty).  params = {FIRE_N_accelerate, FIRE_f_inc, FIRE_f_decelerate, FIRE_a_start,
 * FIRE_dt_max}; state = {dt, a, n_reset}, in/out (the optimizer object's attributes).  Returns the new geometries in
 * angstrom ((X + TR_calc(...)) * bohr2ang) and, in *dot_out (may be NULL), the band-wide sum of pre_vel . F. */
int mopt_neb_fire(mopt_ctx *ctx, int I, int N, const double *X, const double *F, const double *vel, const double *pre_vel,
                  int optimize_num, int fix_init, int fix_end, const double *params, double *state, double bohr2ang,
                  double *new_geometry_ang, double *dot_out);

/* SteepestDescentOptimizer.optimize (multioptpy/Optimizer/gradientdescent_neb.py:22-51): 0.5 * F clipped to half the
 * neighbour distances; end images free or fixed. */
int mopt_neb_sd(mopt_ctx *ctx, int I, int N, const double *X, const double *F, int fix_init, int fix_end, double bohr2ang,
                double *new_geometry_ang);

/* ---- callers around the step: CalculateMoveVector / TrustRadius / Kabsch ------------------ */

/* TrustRadius.update_trust_radii's predicted energy change (multioptpy/Optimizer/trust_radius.py:146-151):
 * out[b] = g_b . s_b + 1/2 s_b^T H_b s_b for batch x n x n Hessians. */
int mopt_quadratic_model(mopt_ctx *ctx, int n, int batch, const double *H, const double *g, const double *s, double *out);

/* CalculateMoveVector.calc_move_vector, the part after optimizer.run (multioptpy/optimizer.py:789-812):
 * if ||step|| > trust_radius: step = trust_radius * step / ||step||;  new_geometry = (x - step) * bohr2ang (angstrom),
 * move_vector = step (bohr).  n = 3N values each; step_norm (may be NULL) = ||step|| before clipping. */
int mopt_clip_and_move(mopt_ctx *ctx, int n, const double *x_bohr, const double *step, double trust_radius, double bohr2ang,
                       double *new_geometry_ang, double *move_vector, double *step_norm);

/* Calculationtools.kabsch_algorithm (multioptpy/Utils/calc_tools.py:412-425) for N x 3 arrays: both centred on their
 * centroids (the reference does that IN PLACE on its arguments; here the centred Q is returned in Q_centred), proper
 * rotation from the 3 x 3 SVD, P_aligned = (R P_centred^T)^T. */
int mopt_kabsch(mopt_ctx *ctx, int N, const double *P, const double *Q, double *P_aligned, double *Q_centred);

/* ---- RSIRFO.run (multioptpy/Optimizer/rsirfo.py:285-490), batched ---------------------- */

typedef struct mopt_rsirfo_params {
    /* constructor configuration, rsirfo.py:26-91 */
    double alpha0, alpha_max, alpha_step_max;
    double small_eigval_thresh;
    double trust_radius_min, trust_radius_max;
    double good_step_threshold, poor_step_threshold;
    double trust_radius_increase_factor, trust_radius_decrease_factor;
    double energy_change_threshold, gradient_norm_threshold, step_norm_tolerance;
    double adaptive_trust_gradient_norm_threshold;
    double max_curvature_factor, negative_curvature_safety;
    double level_shift_value, condition_number_threshold;
    int32_t max_micro_cycles;
    int32_t saddle_order;
    int32_t rule;                 /* MOPT_RULE_* */
    int32_t use_adaptive_trust_radius;
    int32_t use_level_shift, auto_level_shift;
    int32_t neb_mode;
    int32_t reserved;
} mopt_rsirfo_params;

typedef struct mopt_rsirfo_state {
    /* the mutable attributes RSIRFO keeps between calls */
    double trust_radius;
    double prev_energy;
    double predicted[3];          /* predicted_energy_changes (last <= 3) */
    double actual[3];             /* actual_energy_changes   (last <= 3) */
    int32_t n_predicted, n_actual;
    int32_t has_prev;             /* prev_geometry/prev_gradient/prev_energy are set */
    int32_t iteration;
    int32_t converged;
    int32_t prev_eigvec_size;     /* -1 = None */
    int32_t level_shift_applied;
    /* diagnostics of the last call (not inputs) */
    int32_t branch;               /* 0 within radius, 1 brent, 2 newton loop, 3 steepest-descent fallback */
    int32_t hess_updated;
    int32_t n_kept;               /* eigenpairs after the 1e-6 filter */
    int32_t jacobi_sweeps;
    int32_t secular_iters;
    double alpha, lambda_aug, initial_step_norm, gradient_norm;
} mopt_rsirfo_state;

/* One RSIRFO.run for `batch` independent geometries (images).
 *   hessian      in : batch x n x n   (self.hessian)
 *   bias_hessian in : batch x n x n or NULL (self.bias_hessian is None)
 *   geom         in : batch x n       geom_num_list (bohr)
 *   B_g          in : batch x n       biased gradient (used for the step)
 *   g            in : batch x n       raw gradient (used for the Hessian update)
 *   pre_geom/pre_g in: batch x n or NULL (no history: first call)
 *   B_e          in : batch           biased energy
 *   step        out : batch x n       the value run() returns (to be SUBTRACTED by the caller)
 *   H_inplace   out : batch x n x n or NULL   (caller-aliased array after `+=`, F7)
 *   H_sym       out : batch x n x n   (new self.hessian; equals `hessian` when no update happened)
 * params: one struct per image; state: one struct per image, in/out. */
int mopt_rsirfo_run(mopt_ctx *ctx, int n, int batch, const mopt_rsirfo_params *params,
                    mopt_rsirfo_state *state, const double *hessian, const double *bias_hessian,
                    const double *geom, const double *B_g, const double *g, const double *pre_geom,
                    const double *pre_g, const double *B_e, double *step, double *H_inplace, double *H_sym);

/* ---- MF_RSIRFO.run (multioptpy/Optimizer/mf_rsirfo.py:128-265) in two halves.  The mode follower (Optimizer/mode_following.py: overlaps with
   its reference vectors, Hungarian / greedy matching, EMA update) works on the eigenvectors between the eigen-decomposition and the step, on the
   host; everything else stays on the device between the two calls.
     mopt_mf_rsirfo_begin   Hessian update (rsirfo.py:1316-1372), TR/rot projection, eigh (mf_rsirfo.py:148-171): W_out[n] ascending eigenvalues,
                            V_out[n][n] row k = eigenvector k (the identity when the spectrum is not finite); H_inplace / H_sym as mopt_rsirfo_run.
     mopt_mf_rsirfo_finish  trust radius with the curvature of the first target (:196-205), image function over `targets` (ascending positions, :207-218),
                            RS step with the UNPROJECTED gradient B_g (:163), predicted change, state; step[n] = the value run() returns.
   One geometry per call pair; finish must follow begin on the same context with the same n. */
int mopt_mf_rsirfo_begin(mopt_ctx *ctx, int n, const mopt_rsirfo_params *params, const mopt_rsirfo_state *state, const double *hessian,
                         const double *bias_hessian, const double *geom, const double *B_g, const double *g, const double *pre_geom, const double *pre_g,
                         double B_e, double *W_out, double *V_out, double *H_inplace, double *H_sym, int *hess_updated);
int mopt_mf_rsirfo_finish(mopt_ctx *ctx, int n, const mopt_rsirfo_params *params, mopt_rsirfo_state *state, const int *targets, int ntargets, double *step);

/* ---- EnhancedRSPRFO.run (multioptpy/Optimizer/rsprfo.py:713-886): P-RFO transition-state step ------------------ */

typedef struct mopt_rsprfo_params {
    /* constructor configuration, rsprfo.py:84-190 */
    double alpha0, alpha_max, alpha_min, alpha_step_max;
    double micro_cycle_rtol, micro_cycle_atol;
    double trust_radius_min, trust_radius_max;
    double eta_2, eta_3, gamma_1, gamma_2;
    double rejection_threshold;
    double min_positive_eigval, min_negative_eigval;
    double gradient_scaling_threshold, min_step_scale, gradient_trust_coupling;
    double overlap_threshold, mixing_threshold;
    int32_t max_micro_cycles, saddle_order, rule;
    int32_t step_rejection_enabled, max_consecutive_rejections;
    int32_t hessian_shift_enabled, gradient_scaling_enabled, adaptive_trust_enabled, adapt_trust_radius;
    int32_t mode_following_enabled, eigvec_following;
    int32_t reserved;
} mopt_rsprfo_params;

typedef struct mopt_rsprfo_state {
    double trust_radius;
    double prev_energy;
    double last_predicted;          /* predicted_energy_changes[-1] */
    double last_ratio;              /* out: reduction ratio of the processed previous step (NaN if none) */
    double last_actual;             /* out: actual energy change appended this call (NaN if none) */
    double new_predicted;           /* out: predicted energy change of this step */
    int32_t has_prev;               /* prev_energy is not None and predictions exist */
    int32_t consecutive_rejections;
    int32_t ts_mode_idx;            /* -1 = None */
    int32_t iter;
    /* diagnostics */
    int32_t hess_updated, shifted, step_accepted, micro_converged, n_max, jacobi_sweeps;
    double step_norm, gradient_norm;
} mopt_rsprfo_state;

/* One EnhancedRSPRFO.run for a single geometry (n = 3N).
 *   hessian       in : n x n (self.hessian, already symmetrised by set_hessian), bias_hessian n x n or NULL
 *   geom, B_g     in : n          current geometry / gradient
 *   pre_geom, pre_B_g in: n or NULL   (Hessian update with B_g - pre_B_g, rsprfo.py:763-765)
 *   prev_gradient, prev_move_vector in: n or NULL  (self.prev_gradient / self.prev_move_vector of the previous call)
 *   pre_move_norm in : |pre_move_vector| handed in by the caller, < 0 when the argument was empty (rsprfo.py:924-929)
 *   ts_vec        in : n or NULL  tracked TS eigenvector (self.ts_mode_eigvec)
 *   step         out : n          the returned move vector (to be subtracted by the caller)
 *   H_out        out : n x n      self.hessian after the (possibly reverted) update
 *   ts_vec_out   out : n          new tracked eigenvector (untouched when saddle_order == 0) */
int mopt_rsprfo_run(mopt_ctx *ctx, int n, const mopt_rsprfo_params *params, mopt_rsprfo_state *state,
                    const double *hessian, const double *bias_hessian, const double *geom, const double *B_g,
                    const double *pre_geom, const double *pre_B_g, double B_e, const double *prev_gradient,
                    const double *prev_move_vector, double pre_move_norm, const double *ts_vec, double *step,
                    double *H_out, double *ts_vec_out);

/* ---- NEB band ------------------------------------------------------------------------ */

/* CaluculationBNEB.calc_force (multioptpy/MEP/pathopt_bneb_force.py:33-65) for I images of N atoms.
 *   X, G : I x N x 3, E : I;  F, tau out: I x N x 3 (tau = tau_list, pathopt_bneb_force.py:60)
 *   optimize_num, apply_ci_neb: the CI-NEB switch `optimize_num > APPLY_CI_NEB` (:50-55). */
int mopt_bneb_force(mopt_ctx *ctx, int I, int N, const double *X, const double *E, const double *G,
                    int optimize_num, int apply_ci_neb, double *F, double *tau);

/* The spring / tangent variants of nebmain (neb.py:769-792), same arguments (tau only exists for BNEB and may be NULL):
 *   MOPT_NEB_BNEB   CaluculationBNEB   pathopt_bneb_force.py:21-117    (default; == mopt_bneb_force)
 *   MOPT_NEB_BNEB3  CaluculationBNEB3  pathopt_bneb_force.py:305-356   (-bneb2: + next-nearest-neighbour springs)
 *   MOPT_NEB_EWBNEB CaluculationEWBNEB pathopt_ewbneb_force.py:21-117  (-ewbneb: + energy-weighted springs)
 *   MOPT_NEB_DNEB   CaluculationDNEB   pathopt_dneb_force.py:20-130    (-dneb: per-atom tangents, doubly nudged) */
enum { MOPT_NEB_BNEB = 0, MOPT_NEB_BNEB3 = 1, MOPT_NEB_EWBNEB = 2, MOPT_NEB_DNEB = 3 };
int mopt_neb_force(mopt_ctx *ctx, int variant, int I, int N, const double *X, const double *E, const double *G,
                   int optimize_num, int apply_ci_neb, double *F, double *tau);

/* Device-resident band: the per-image Hessians and RSIRFO instances of RFOOptimizer
 * (multioptpy/Optimizer/rfo_neb.py:86-208) stay in HBM between steps instead of going
 * through tmp_hessian_{i}.npy (rfo_neb.py:18-25,175). */
typedef struct mopt_band mopt_band;

int mopt_band_create(mopt_ctx *ctx, int I, int N, mopt_band **out);
int mopt_band_destroy(mopt_band *band);
/* image-sharded use: this rank owns images [first, first+count) of a band of I images. */
int mopt_band_set_shard(mopt_band *band, int first, int count);
/* H = NULL -> identity (rfo_neb.py:25). */
int mopt_band_set_hessian(mopt_band *band, int image, const double *H);
/* H_image += 0.5 (B + B^T) on the device: the bias Hessian that nebmain's _apply_bias_potential adds to tmp_hessian_{i}.npy (neb.py:806-832) */
int mopt_band_add_hessian(mopt_band *band, int image, const double *B);
int mopt_band_get_hessian(mopt_band *band, int image, double *H);
int mopt_band_get_state(mopt_band *band, int image, mopt_rsirfo_state *out);
/* options: ratio_of_rfo_step (rfo_neb.py:98), fix_init_edge / fix_end_edge (trust_radius_neb.py:7-8),
 * apply_ts_opt (rfo_neb.py:97), APPLY_CI_NEB (pathopt_bneb_force.py:22). */
int mopt_band_configure(mopt_band *band, double ratio_of_rfo_step, int fix_init_edge, int fix_end_edge,
                        int apply_ts_opt, int apply_ci_neb);
/* Copy one step's inputs to HBM: X, G : I x N x 3 (bohr, hartree/bohr), E : I. */
int mopt_band_upload(mopt_band *band, const double *X, const double *E, const double *G);
/* RFOOptimizer.optimize (rfo_neb.py:104-208) on the uploaded inputs: BNEB force, Ayala curvature,
 * per-image RSIRFO.run, step limit, TR_calc, FIRE, blend.  Leaves the new geometry (Angstrom, as
 * the reference returns it) on the device; the previous inputs become pre_geom / pre_gradient. */
int mopt_band_step(mopt_band *band, int optimize_num);
int mopt_band_download(mopt_band *band, double *newX_angstrom, double *F, double *rfo_step);
/* Failure bits of the band steps since the host last looked (download, get_state or this call): 4 = the eigen-solver failed for an
 * image, 8 = the workgroup hand-off of the persistent tridiagonalisation timed out (GPU shared with a long-running kernel).  The bits
 * are sticky on the device -- a later step does not clear them.  A time-out of the LAST step is repaired here (the flagged images are
 * solved again with one launch per Householder step, the tail of the step is repeated) and not reported; anything else returns
 * MOPT_E_NUMERIC.  A caller that enqueues several steps without downloading (bench.py) calls this once behind them.  bits may be NULL. */
int mopt_band_failures(mopt_band *band, int *bits);
/* test hook of the debug mode (mopt_debug_poison): writes one byte into the guard gap behind the band's X array */
int mopt_debug_damage_gap(mopt_band *band);
/* Same as upload / download with DEVICE pointers (inputs already resident in HBM, e.g. produced by a GPU
 * calculator or staged ahead of time); E_host (I doubles, host) is only read when a step re-creates the
 * optimizer instances (optimize_num == 0) and may be NULL otherwise.  Copies are stream-ordered on the ctx. */
int mopt_band_upload_dev(mopt_band *band, const double *dX, const double *dE, const double *dG, const double *E_host);
int mopt_band_download_dev(mopt_band *band, double *d_newX_angstrom);

/* ---- constraints ---------------------------------------------------------------------
 * mopt_constraint_project_hess: ProjectOutConstrain.calc_project_out_hess (multioptpy/Constraint/constraint_condition.py:657-754):
 *   H_out = H - P H - H P + P H P + k_hard P + k_soft P_soft,  P = Q^T Q (Q: c x n orthonormal rows = the basis of the current
 *   constraint rows), P_soft = S^T S (S: cs x n, the orthogonalised residuals of the stored bases; cs may be 0).
 * mopt_crsirfo_run: CRSIRFO.run (multioptpy/Optimizer/crsirfo.py:48-182) behind the SHAKE-like adjustment: arguments as
 *   mopt_rsirfo_run (batch 1) plus U (n x m row-major, orthonormal columns spanning the null space of the constraints): Hessian
 *   update, U^T (H + bias) U, eigen-solve and RS-I-RFO step in the subspace, step = U step_sub.  No TR/rot projection. */
int mopt_constraint_project_hess(mopt_ctx *ctx, int n, const double *H, int c, const double *Q, int cs, const double *S, double k_hard,
                                 double k_soft, double *H_out);
int mopt_crsirfo_run(mopt_ctx *ctx, int n, int m, const mopt_rsirfo_params *params, mopt_rsirfo_state *state, const double *hessian,
                     const double *bias_hessian, const double *geom, const double *B_g, const double *g, const double *pre_geom,
                     const double *pre_g, const double *B_e, const double *U, double *step, double *H_inplace, double *H_sym);
/* The same with the null-space basis formed on the device (crsirfo.py:26-46 returns it from an SVD on the host): Qrows = r orthonormal rows
 * (r x n, r <= 64) spanning the row space of the normalised constraint vectors; U = the last n - r columns of the Householder QR's orthogonal
 * factor of Qrows^T.  The step, the predicted change and the trust-radius logic do not depend on the choice of basis inside the null space. */
int mopt_crsirfo_run_rowspace(mopt_ctx *ctx, int n, int r, const mopt_rsirfo_params *params, mopt_rsirfo_state *state, const double *hessian,
                              const double *bias_hessian, const double *geom, const double *B_g, const double *g, const double *pre_geom,
                              const double *pre_g, const double *B_e, const double *Qrows, double *step, double *H_inplace, double *H_sym);

/* ---- bias potentials ------------------------------------------------------------------------
 * AFIR (multioptpy/Potential/AFIR_potential.py:19-54): E = alpha A / B over the atom pairs of two fragments; the reference gets
 * its gradient and Hessian from torch autograd on the CPU every step (Potential/potential.py:127-152), here they are analytic
 * kernels.  X: N x 3 (bohr); cov: N covalent radii (bohr); frag1 / frag2: 0-based atom indices (disjoint); alpha in hartree/bohr
 * (AFIR_potential.py:34-37 from gamma).  Out: E (1), grad (N x 3), H (3N x 3N). */
int mopt_afir(mopt_ctx *ctx, int N, const double *X, const double *cov, int n1, const int *frag1, int n2, const int *frag2, double alpha,
              double *E, double *grad, double *H);

/* Distance restraints E = sum_p V_p(r_p): harmonic 1/2 k (r - r0)^2 (StructKeepPotential.calc_energy, Potential/keep_potential.py:21-58) or
 * anharmonic D (1 - exp(-sqrt(k / 2D) (r - r0)))^2 (StructAnharmonicKeepPotential.calc_energy, Potential/anharmonic_keep_potential.py:13-27);
 * the reference holds one pair per object and differentiates with torch, Potential/potential.py:127-152.  X: N x 3 (bohr); pairs: npairs x 2
 * 0-based atoms; kr: npairs x 3 = (spring constant hartree / bohr^2, r0 in bohr, well depth D in hartree: D < 0 = harmonic, D >= 0 =
 * anharmonic, D = 0 contributes nothing).  Out: E (1), grad (N x 3), H (3N x 3N). */
int mopt_keep_pot(mopt_ctx *ctx, int N, const double *X, int npairs, const int *pairs, const double *kr, double *E, double *grad, double *H);

/* Angle (kind 0) and dihedral-angle (kind 1) restraints E = sum_t 1/2 k_t (angle_t - angle0_t)^2 with the reference's treatment of the singular
 * geometries: StructKeepAnglePotential.calc_energy (Potential/keep_angle_potential.py:58-224: Taylor branches within 1e-3 rad of 0 / 180 degrees)
 * and StructKeepDihedralAnglePotential.calc_energy (Potential/keep_dihedral_angle_potential.py:60-154: periodic difference, smoothstep switch-off
 * near collinear bonds).  Gradient and Hessian are exact derivatives of those expressions (hyper-dual evaluation on the device) -- what the
 * reference gets from torch.func.jacrev / hessian on the CPU, Potential/potential.py:127-152.  X: N x 3 (bohr); atoms: nterms x 3 (4) 0-based;
 * kp: nterms x 2 = (spring constant hartree / rad^2, equilibrium angle in radians).  Out: E (1), grad (N x 3), H (3N x 3N). */
/* kind 2: out-of-plane-angle restraints, StructKeepOutofPlainAnglePotential.calc_energy (Potential/keep_outofplain_angle_potential.py:33-146): atoms per
 * term = centre, probe, two plane atoms; zero inside the collinearity guard. */
int mopt_keep_angle(mopt_ctx *ctx, int kind, int N, const double *X, int nterms, const int *atoms, const double *kp, double *E, double *grad, double *H);
/* Fragment-centre distance restraint 1/2 k (|c1 - c2| - r0)^2, StructKeepPotentialv2.calc_energy (Potential/keep_potential.py:79-115): frag1 / frag2
 * 0-based atom lists (n1 + n2 <= 64), k in hartree / bohr^2, r0 in bohr. */
int mopt_keep_pot_v2(mopt_ctx *ctx, int N, const double *X, int n1, const int *frag1, int n2, const int *frag2, double k, double r0, double *E, double *grad,
                     double *H);
/* The well potentials of Potential/switching_potential.py -- one radial well (linear walls outside [a, d], quintic switches on [a, b] and [c, d], zero
 * between) on: kind 0 WellPotential (:5-67; list1 / list2 = the two fragments, distance of their centroids), 1 WellPotentialWall (:70-135; |x_i[dir]| of every
 * atom of list1, dir = prm[8] in 0 / 1 / 2), 2 WellPotentialVP (:138-196; |x_i - p|, p = prm[5..7] in bohr), 3 WellPotentialAround (:199-257; list1 = centre
 * atoms, list2 = target atoms, |x_i - centroid|).  prm[0..3] = a, b, c, d in bohr, prm[4] = wall energy in hartree; 0-based atom lists (n2 = 0 for kinds 1, 2),
 * n1 + n2 <= 64.  Energy, gradient and Hessian as the reference takes them from torch autograd (Potential/potential.py:127-152).  Host pointers. */
int mopt_well_pot(mopt_ctx *ctx, int kind, int N, const double *X, int n1, const int *list1, int n2, const int *list2, const double *prm, double *E, double *grad,
                  double *H);

/* ---- model Hessians ----------------------------------------------------------------
 * ApproxHessian.main (multioptpy/ModelHessian/approx_hessian.py:30-112) dispatches by name; on the device:
 *   MOPT_MODEL_FISCHERD3OLD  FischerD3ApproxHessianOld.main (ModelHessian/fischerd3old.py:355-381), the `-modelhess` default
 *                            (interface.py:325-330): Fischer-Almloef stretch / bend / torsion force constants on the covalent
 *                            connectivity + D3(BJ)-style pair terms, symmetrised and TR/rot-projected (Utils/calc_tools.py:249-316).
 *   MOPT_MODEL_FISCHERD3     FischerD3ApproxHessian.main (ModelHessian/fischerd3.py:177-304): the same internal-coordinate terms with
 *                            stricter guards, pair terms whose C6 is scaled by the atoms' fractional coordination numbers
 *                            (fischerd3.py:46-60, 224-229), pair exclusions and torsion bond counts from the 1.1 x table.
 *   MOPT_MODEL_FISCHER       FischerApproxHessian.main (ModelHessian/fischer.py:216-236): internal-coordinate terms only.
 *   MOPT_MODEL_SCHLEGEL      SchlegelApproxHessian.main (ModelHessian/schlegel.py:66-144): H = B^T K B over the all-pairs stretch rows
 *                            (Coordinate/redundant_coordinate.py:15-43), K filled from the bond / angle / dihedral tables.
 *   MOPT_MODEL_MORSE         MorseApproxHessian.create_model_hessian (ModelHessian/morse.py:50-106): all-pairs Morse curvature with
 *                            r_eq from the GNB radii (Parameters/gnb.py:39-55), De = 0.10, a = 0.20; symmetrised, never projected
 *                            (the reference returns it unprojected: `project` is ignored).
 * X: batch x N x 3 (bohr).  par: six arrays of N doubles in a row -- covalent radius, C6, van-der-Waals radius, r4/r2, reference
 * coordination number of every atom in atomic units (the values of multioptpy/Parameters: covalent_radii_lib, D2_C6_coeff_lib,
 * D2_VDW_radii_lib, D3Parameters.get_r4r2; fischerd3.py:29-44 ref_cn_map, 4 for elements it does not list) and one model-specific array:
 * Schlegel's element code (row of his B table as schlegel.py:27-63 assigns it, + 8 for hydrogen) or, for MOPT_MODEL_MORSE, the GNB radius.
 * d3: {s6, s8, a1, a2}.
 * project = 0 skips the TR/rot projection.  H: batch x 3N x 3N. */
enum { MOPT_MODEL_FISCHERD3OLD = 0, MOPT_MODEL_FISCHERD3 = 1, MOPT_MODEL_FISCHER = 2, MOPT_MODEL_SCHLEGEL = 3, MOPT_MODEL_MORSE = 4,
       MOPT_MODEL_LINDH2007D3 = 5 };
int mopt_model_hessian(mopt_ctx *ctx, int kind, int N, int batch, const double *X, const double *par, const double *d3, int project, double *H);
/* The same with the two things ApproxHessian.main hands on besides the geometry (approx_hessian.py:30-112):
 *   MOPT_MODEL_LINDH2007D3   Lindh2007D3ApproxHessian.main (ModelHessian/lindh2007d3.py:777-822), also the dispatcher's fall-back for names it does
 *                            not know (approx_hessian.py:88-91): stretch of every pair, bend of every triple, torsion and out-of-plane term of
 *                            every quadruple (no connectivity), the reference's triangle-wise accumulation and mirroring rule (:803-808), TR/rot
 *                            projection, then H = V |lambda| 0.1 exp(-|g|^2) V^T.  par: the UFF van-der-Waals distance (bohr, UFF_VDW_distance_lib
 *                            / bohr2angstroms) in the van-der-Waals slot, the element group of select_idx (:60-82; 0, 1, 2) in the model-specific slot.
 *   G       batch x N x 3 Cartesian gradients (only the Lindh model reads them; NULL = zero)
 *   flags   MOPT_MODEL_FLAG_CLIP: the `clip` modifier (approx_hessian.py:104-120): eigenvalues |l| >= 1 become sign(l) (2 - |l|^-0.1) */
enum { MOPT_MODEL_FLAG_CLIP = 1 };
int mopt_model_hessian_ex(mopt_ctx *ctx, int kind, int flags, int N, int batch, const double *X, const double *par, const double *d3, const double *G,
                          int project, double *H);
/* The `sr` modifier of ApproxHessian.main (approx_hessian.py:100-102 -> ShortRangeCorrectionHessian.main, ModelHessian/shortrange.py:311-346): the
 * short-range Coulomb pair terms of all non-bonded atom pairs (1.1 x covalent radii, 15 bohr cut-off), TR/rot-projected on their own, added to
 * H, symmetrised, NaN -> 0.  X: batch x N x 3 (bohr); cov: N covalent radii (bohr); chg: N charges (shortrange.py:147-186); prm: omega, cx_sr,
 * scaling factor, cut-off (the reference's defaults 0.2, 0.78, 0.5, 15.0); H: batch x 3N x 3N, in and out.  Host pointers. */
int mopt_model_hessian_sr(mopt_ctx *ctx, int N, int batch, const double *X, const double *cov, const double *chg, const double *prm, double *H);
/* The `clip` modifier on its own (approx_hessian.py:104-120): H <- V smooth(lambda) V^T for batch n x n host matrices, in place -- for the
 * combinations in which another modifier comes between the model and the clipping (the reference's order is ts, sr, clip). */
int mopt_hessian_clip(mopt_ctx *ctx, int n, int batch, double *H);
/* The same for the owned images of a band, from the geometries last uploaded, straight into the band's Hessians in HBM
 * (the reference recomputes the model every -mfc band steps and hands it over in tmp_hessian_{i}.npy, neb.py:515-523). */
int mopt_band_model_hessian(mopt_band *band, int kind, const double *par, const double *d3);
/* (with flags as mopt_model_hessian_ex; the Lindh model takes the gradients last uploaded) */
int mopt_band_model_hessian_ex(mopt_band *band, int kind, int flags, const double *par, const double *d3);

/* ---- image-sharded band and its exchange layer ------------------------------------------
 * One image block per GPU: rank r of R owns images [r I / R, (r + 1) I / R) -- their Hessians, RSIRFO instances
 * and eigen-solves (rfo_neb.py:137-178) never move.  Per step the ranks exchange, inside the library and on a
 * dedicated stream per band: the boundary images' coordinates and gradients with rank +- 1 (BNEB tangent,
 * pathopt_bneb_force.py:67-117; calculate_gamma, rfo_neb.py:63-68; TR_calc, trust_radius_neb.py:36-37), all
 * energies (extremum_list_index / CI-NEB, pathopt_bneb_force.py:6-17,50-55) and one partial sum per rank of
 * FIRE's band-wide sum v_prev . F (fire_neb.py:60-65).
 * Transports: RCCL over xGMI (ncclSend / ncclRecv / ncclAllGather; librccl.so is loaded on first use) for a
 * multi-process job (one communicator per process from a shared id) or for all devices of one process
 * (ncclCommInitAll); "peer": plain copies between the bands of one process (hipMemcpyPeerAsync).  Results do not
 * depend on the transport.
 * Call sequence per step, for the bands of THIS process that share a communicator group (one band in a
 * multi-process job): mopt_band_upload_shard[_dev] on each -> mopt_halo_exchange(all) -> mopt_band_step_multi(all). */
typedef struct mopt_comm mopt_comm;
#define MOPT_COMM_ID_BYTES 128
int mopt_comm_unique_id(void *id, int nbytes);                       /* rank 0 of a multi-process job; the host program broadcasts it */
int mopt_comm_create_rccl_rank(mopt_ctx *ctx, int nranks, int rank, const void *id, mopt_comm **out);
int mopt_comm_create_rccl_all(int nranks, mopt_ctx *const *ctxs, mopt_comm **outs);   /* one process, ctxs[r] = context of rank r's device */
int mopt_comm_create_peer(int nranks, mopt_ctx *const *ctxs, mopt_comm **outs);       /* one process, no library (contexts may share a device) */
int mopt_comm_destroy(mopt_comm *comm);
int mopt_band_attach_comm(mopt_band *band, mopt_comm *comm);         /* after mopt_band_set_shard; the shard must be the rank's block */
int mopt_band_upload_shard(mopt_band *band, const double *X_own, const double *E_own, const double *G_own);          /* host, count images */
int mopt_band_upload_shard_dev(mopt_band *band, const double *dX_own, const double *dE_own, const double *dG_own);   /* device, stream-ordered */
int mopt_halo_exchange(int nbands, mopt_band *const *bands);
int mopt_band_step_multi(int nbands, mopt_band *const *bands, int optimize_num);

#ifdef __cplusplus
}
#endif
#endif /* MOPT_HIP_H */
