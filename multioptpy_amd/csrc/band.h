// Internal layout of the device-resident NEB band (band.hip) shared with the exchange layer (comm.hip).
#pragma once
#include <vector>

#include "pipeline.h"

struct mopt_comm;

struct mopt_band {
    mopt_ctx *ctx = nullptr;
    int I = 0, N = 0, n = 0, first = 0, count = 0;
    double ratio = 0.5;
    int fix_init = 0, fix_end = 0, apply_ts_opt = 1, apply_ci = 99999;
    RsirfoWork wk;
    double *X = nullptr, *G = nullptr, *E = nullptr, *pX = nullptr, *pG = nullptr, *F = nullptr, *pF = nullptr, *tau = nullptr;
    double *gamma = nullptr, *rfo_delta = nullptr, *newX = nullptr, *wx = nullptr, *wg = nullptr;
    void *block = nullptr;
    std::vector<std::pair<size_t, size_t>> gaps;     // debug mode (mopt_debug_poison): guard gaps inside `block`
    std::vector<const char *> gap_names;
    bool have_prev = false, uploaded = false, opts_init = false, alloc_done = false;
    bool stepped = false;            // wk.einfo is valid
    bool have_vecs = false;          // wk.Vt holds the eigenvectors of the previous step (warm start)
    int warm_run = 0;                // consecutive warm-started solves (a cold solve every 64 re-orthogonalises V)
    int warm_enabled = 1;
    int steps_since_check = 0;       // band steps enqueued since the host last looked at the failure words (download / mopt_band_failures)
    int last_optimize_num = 0; bool last_have_prev = false, last_have_coef = false;      // of the last step: what a repeat of its tail needs
    bool dense_hessians = false;     // an explicit (exact / model) Hessian has been handed to an owned image: no low-rank structure to expect
    bool fused_step = false;         // this step takes the fused path (decided in phase a, where the TR/rot basis is forked)
    std::vector<char> H_sym;         // per owned image: the resident Hessian is known to be bitwise symmetric (identity, model Hessian,
                                     // a symmetric upload): the band then takes the fused update + projection path (k_fused.hip)
    std::vector<double> hE;
    // ---- image-sharded band with an exchange layer (comm.hip): this object owns images [first, first + count) and holds
    // the neighbours' boundary images (X, G) and all energies after mopt_halo_exchange
    mopt_comm *comm = nullptr;
    double *dots = nullptr;          // [MOPT_MAX_RANKS] partial sums of FIRE's band-wide  sum v_prev . F  (slot r: rank r)
    hipStream_t cst = nullptr;       // dedicated stream of the exchange
    hipEvent_t ev_in = nullptr, ev_out = nullptr, ev_dot = nullptr, ev_dot_done = nullptr;
    bool halo_done = false;
};

#define MOPT_MAX_RANKS 64

int mopt_band_alloc_internal(mopt_band *b);
int mopt_band_step_phase_a(mopt_band *b, int optimize_num);
int mopt_band_step_phase_b(mopt_band *b, int optimize_num);
int mopt_comm_rank(const mopt_comm *c);
void mopt_comm_detach_band(mopt_comm *c, mopt_band *b);
int mopt_comm_nranks(const mopt_comm *c);
extern "C" int mopt_band_step_multi(int nbands, mopt_band *const *bands, int optimize_num);
