// Exchange layer of the image-sharded NEB band: one image block per GPU, the per-image Hessians and
// optimizer state never move; per band step the ranks exchange
//   * the boundary images' coordinates AND gradients with rank +- 1 (the neighbour terms of the BNEB tangent,
//     pathopt_bneb_force.py:67-117; of calculate_gamma, rfo_neb.py:63-68; of TR_calc, trust_radius_neb.py:36-37),
//   * all energies (extremum_list_index / CI-NEB, pathopt_bneb_force.py:6-17,50-55; is_max at optimizer creation, rfo_neb.py:123),
//   * one partial sum per rank of FIRE's band-wide  sum v_prev . F  (fire_neb.py:60-65),
// on a dedicated stream per band, ordered against the band's compute stream by events only (no host wait).
//
// Two transports behind the same plan:
//   RCCL   (librccl.so over xGMI, loaded on first use): ncclSend / ncclRecv for the boundary rows, ncclAllGather for the
//          energies and the partial sums; one communicator per rank, created either by every process of a multi-process job
//          from a shared unique id (bench.py under torch.distributed.run) or for all devices of ONE process (ncclCommInitAll --
//          the reference hands all images to one Python process).
//   peer   copies between the bands of one process (hipMemcpyPeerAsync, or plain device copies when the bands share a device):
//          no library, the lowest latency for the < 10 KB messages of this path; also what the single-GPU tests run.
// Both push the same bytes into the same slots, and the partial sums are added in rank order on every rank (k_neb_post), so the
// result of a step does not depend on the transport.
#include <dlfcn.h>

#include <cstring>
#include <mutex>
#include <new>
#include <string>
#include <vector>

#include <rccl/rccl.h>

#include "band.h"

namespace {

struct Rccl {
    void *h = nullptr;
    decltype(&ncclGetUniqueId) GetUniqueId = nullptr;
    decltype(&ncclCommInitRank) CommInitRank = nullptr;
    decltype(&ncclCommInitAll) CommInitAll = nullptr;
    decltype(&ncclCommDestroy) CommDestroy = nullptr;
    decltype(&ncclGroupStart) GroupStart = nullptr;
    decltype(&ncclGroupEnd) GroupEnd = nullptr;
    decltype(&ncclSend) Send = nullptr;
    decltype(&ncclRecv) Recv = nullptr;
    decltype(&ncclAllGather) AllGather = nullptr;
    decltype(&ncclGetErrorString) GetErrorString = nullptr;
    std::string err;
};
Rccl g_rccl;
std::mutex g_rccl_mu;

bool rccl_load() {
    std::lock_guard<std::mutex> lk(g_rccl_mu);
    if (g_rccl.h) return true;
    // An RCCL that is already in the process (PyTorch brings its own, built against the HIP runtime it also brings and that this
    // library then shares) is taken first: two RCCL copies on two HIP builds in one process is the configuration to avoid.
    const char *names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so"};
    void *h = nullptr;
    // (RTLD_LOCAL: when this library has to bring RCCL in itself and PyTorch is imported LATER, PyTorch loads its own copy next to it; with
    // a global load the second copy's rocm_smi globals were bound to the first one's and the process aborted at exit -- "double free or
    // corruption" in the static destructors of librocm_smi64, found with rocgdb in round 4 when tests/test_gpu_configs.py was run on its own)
    for (const char *nm : names) { h = dlopen(nm, RTLD_NOW | RTLD_LOCAL | RTLD_NOLOAD); if (h) break; }
    if (!h) for (const char *nm : names) { h = dlopen(nm, RTLD_NOW | RTLD_LOCAL); if (h) break; }
    if (!h) { g_rccl.err = std::string("dlopen(librccl.so): ") + (dlerror() ? dlerror() : "not found"); return false; }
#define SYM(field, name) do { g_rccl.field = (decltype(g_rccl.field))dlsym(h, name); if (!g_rccl.field) { g_rccl.err = std::string("librccl.so lacks ") + name; dlclose(h); return false; } } while (0)
    SYM(GetUniqueId, "ncclGetUniqueId"); SYM(CommInitRank, "ncclCommInitRank"); SYM(CommInitAll, "ncclCommInitAll");
    SYM(CommDestroy, "ncclCommDestroy"); SYM(GroupStart, "ncclGroupStart"); SYM(GroupEnd, "ncclGroupEnd");
    SYM(Send, "ncclSend"); SYM(Recv, "ncclRecv"); SYM(AllGather, "ncclAllGather"); SYM(GetErrorString, "ncclGetErrorString");
#undef SYM
    g_rccl.h = h;
    return true;
}

}  // namespace

enum { MOPT_COMM_RCCL = 0, MOPT_COMM_PEER = 1 };

// the ranks of one band that live in THIS process (all of them for the single-process transports, one for a multi-process job)
struct mopt_comm_group {
    int kind = MOPT_COMM_PEER, nranks = 0, refs = 0;
    std::vector<mopt_band *> bands;          // [nranks], nullptr for ranks of other processes / not yet attached
};

struct mopt_comm {
    mopt_comm_group *grp = nullptr;
    int rank = 0;
    mopt_ctx *ctx = nullptr;
    ncclComm_t nccl = nullptr;
};

int mopt_comm_rank(const mopt_comm *c) { return c->rank; }
int mopt_comm_nranks(const mopt_comm *c) { return c->grp->nranks; }

#define CTRY(ctx, call) do { ncclResult_t r__ = (call); if (r__ != ncclSuccess) { (ctx)->err = std::string(#call) + ": " + g_rccl.GetErrorString(r__); return MOPT_E_DEVICE; } } while (0)

extern "C" int mopt_comm_unique_id(void *id, int nbytes) {
    if (!id || nbytes < (int)sizeof(ncclUniqueId)) return MOPT_E_ARG;
    if (!rccl_load()) return MOPT_E_DEVICE;
    ncclUniqueId u;
    if (g_rccl.GetUniqueId(&u) != ncclSuccess) return MOPT_E_DEVICE;
    std::memcpy(id, &u, sizeof(u));
    return MOPT_OK;
}

extern "C" int mopt_comm_create_rccl_rank(mopt_ctx *ctx, int nranks, int rank, const void *id, mopt_comm **out) {
    if (!ctx || !id || !out || nranks < 1 || nranks > MOPT_MAX_RANKS || rank < 0 || rank >= nranks) return MOPT_E_ARG;
    *out = nullptr;
    if (!rccl_load()) { ctx->err = g_rccl.err; return MOPT_E_DEVICE; }
    hipSetDevice(ctx->device);
    ncclUniqueId u;
    std::memcpy(&u, id, sizeof(u));
    mopt_comm *c = new (std::nothrow) mopt_comm();
    mopt_comm_group *g = new (std::nothrow) mopt_comm_group();
    if (!c || !g) { delete c; delete g; return MOPT_E_NOMEM; }
    g->kind = MOPT_COMM_RCCL; g->nranks = nranks; g->bands.assign(nranks, nullptr); g->refs = 1;
    c->grp = g; c->rank = rank; c->ctx = ctx;
    ncclResult_t r = g_rccl.CommInitRank(&c->nccl, nranks, u, rank);
    if (r != ncclSuccess) { ctx->err = std::string("ncclCommInitRank: ") + g_rccl.GetErrorString(r); delete c; delete g; return MOPT_E_DEVICE; }
    *out = c;
    return MOPT_OK;
}

static int create_group(int kind, int nranks, mopt_ctx *const *ctxs, mopt_comm **outs) {
    if (!ctxs || !outs || nranks < 1 || nranks > MOPT_MAX_RANKS) return MOPT_E_ARG;
    for (int r = 0; r < nranks; ++r) { if (!ctxs[r]) return MOPT_E_ARG; outs[r] = nullptr; }
    mopt_comm_group *g = new (std::nothrow) mopt_comm_group();
    if (!g) return MOPT_E_NOMEM;
    g->kind = kind; g->nranks = nranks; g->bands.assign(nranks, nullptr); g->refs = nranks;
    std::vector<ncclComm_t> comms(nranks, nullptr);
    if (kind == MOPT_COMM_RCCL) {
        if (!rccl_load()) { ctxs[0]->err = g_rccl.err; delete g; return MOPT_E_DEVICE; }
        std::vector<int> devs(nranks);
        for (int r = 0; r < nranks; ++r) devs[r] = ctxs[r]->device;
        ncclResult_t rr = g_rccl.CommInitAll(comms.data(), nranks, devs.data());
        if (rr != ncclSuccess) { ctxs[0]->err = std::string("ncclCommInitAll: ") + g_rccl.GetErrorString(rr); delete g; return MOPT_E_DEVICE; }
    }
    for (int r = 0; r < nranks; ++r) {
        mopt_comm *c = new (std::nothrow) mopt_comm();
        if (!c) {                                            // unwind: the handles created so far, every RCCL communicator, the group
            for (int q = 0; q < r; ++q) { delete outs[q]; outs[q] = nullptr; }
            if (kind == MOPT_COMM_RCCL) for (int q = 0; q < nranks; ++q) if (comms[q]) { hipSetDevice(ctxs[q]->device); g_rccl.CommDestroy(comms[q]); }
            delete g;
            return MOPT_E_NOMEM;
        }
        c->grp = g; c->rank = r; c->ctx = ctxs[r]; c->nccl = comms[r];
        outs[r] = c;
    }
    return MOPT_OK;
}

extern "C" int mopt_comm_create_rccl_all(int nranks, mopt_ctx *const *ctxs, mopt_comm **outs) { return create_group(MOPT_COMM_RCCL, nranks, ctxs, outs); }
extern "C" int mopt_comm_create_peer(int nranks, mopt_ctx *const *ctxs, mopt_comm **outs) { return create_group(MOPT_COMM_PEER, nranks, ctxs, outs); }

// the band leaves its group (mopt_band_destroy): the slot is cleared, nothing of the band is touched
void mopt_comm_detach_band(mopt_comm *c, mopt_band *b) {
    if (!c || !c->grp) return;
    if (c->rank < (int)c->grp->bands.size() && c->grp->bands[c->rank] == b) c->grp->bands[c->rank] = nullptr;
}

extern "C" int mopt_comm_destroy(mopt_comm *c) {
    if (!c) return MOPT_E_ARG;
    if (c->nccl && g_rccl.h) { hipSetDevice(c->ctx->device); g_rccl.CommDestroy(c->nccl); }
    if (c->grp) {
        if (c->rank < (int)c->grp->bands.size() && c->grp->bands[c->rank]) { c->grp->bands[c->rank]->comm = nullptr; c->grp->bands[c->rank] = nullptr; }
        if (--c->grp->refs == 0) delete c->grp;
    }
    delete c;
    return MOPT_OK;
}

extern "C" int mopt_band_attach_comm(mopt_band *b, mopt_comm *c) {
    if (!b || !c) return MOPT_E_ARG;
    mopt_ctx *ctx = b->ctx;
    const int R = c->grp->nranks;
    if (c->ctx != ctx) { ctx->err = "mopt_band_attach_comm: the communicator belongs to another context"; return MOPT_E_ARG; }
    if (b->I % R != 0 || b->count != b->I / R || b->first != c->rank * (b->I / R)) {
        ctx->err = "mopt_band_attach_comm: rank r must own the image block [r I / R, (r + 1) I / R)";
        return MOPT_E_ARG;
    }
    hipSetDevice(ctx->device);
    int rc = mopt_band_alloc_internal(b);
    if (rc != MOPT_OK) return rc;
    b->comm = c;
    c->grp->bands[c->rank] = b;
    if (R > 1) ctx->join_by_events = true;                   // (see common.hpp: no kernel of this context spins for a kernel of another stream any more)
    return MOPT_OK;
}

// ---- the exchange ---------------------------------------------------------------------------------------------------
// Per band (rank r, images [f, f + c) of I, n = 3 N doubles per image), pushed on the band's exchange stream behind ev_in:
//   X[f], G[f]                 -> rank r - 1 (its slots X[f], G[f])            X[f + c - 1], G[f + c - 1] -> rank r + 1
//   E[f .. f + c)              -> every rank
static int halo_rccl(int nb, mopt_band *const *bands) {
    bool any = false;
    for (int k = 0; k < nb; ++k) any = any || bands[k]->comm->grp->nranks > 1;
    mopt_ctx *ctx0 = bands[0]->ctx;
    if (!any) return MOPT_OK;
    // Two groups on the same communicator and stream, one behind the other: (1) the point-to-point boundary rows, (2) the energy
    // all-gather.  RCCL documents grouped send/recv pairs and grouped collectives; a group that MIXES both kinds on one
    // communicator is not a documented use, so they are kept apart (stream order makes (2) follow (1); the cost is one more
    // enqueue of a < 1 KB collective off the critical path).
    CTRY(ctx0, g_rccl.GroupStart());
    for (int k = 0; k < nb; ++k) {
        mopt_band *b = bands[k];
        mopt_comm *c = b->comm;
        const int R = c->grp->nranks, r = c->rank, n = b->n;
        if (R == 1) continue;
        hipSetDevice(b->ctx->device);
        const size_t lo = (size_t)b->first * n, hi = (size_t)(b->first + b->count - 1) * n;
        if (r > 0) {
            CTRY(b->ctx, g_rccl.Send(b->X + lo, n, ncclDouble, r - 1, c->nccl, b->cst));
            CTRY(b->ctx, g_rccl.Send(b->G + lo, n, ncclDouble, r - 1, c->nccl, b->cst));
            CTRY(b->ctx, g_rccl.Recv(b->X + lo - n, n, ncclDouble, r - 1, c->nccl, b->cst));
            CTRY(b->ctx, g_rccl.Recv(b->G + lo - n, n, ncclDouble, r - 1, c->nccl, b->cst));
        }
        if (r < R - 1) {
            CTRY(b->ctx, g_rccl.Send(b->X + hi, n, ncclDouble, r + 1, c->nccl, b->cst));
            CTRY(b->ctx, g_rccl.Send(b->G + hi, n, ncclDouble, r + 1, c->nccl, b->cst));
            CTRY(b->ctx, g_rccl.Recv(b->X + hi + n, n, ncclDouble, r + 1, c->nccl, b->cst));
            CTRY(b->ctx, g_rccl.Recv(b->G + hi + n, n, ncclDouble, r + 1, c->nccl, b->cst));
        }
    }
    CTRY(ctx0, g_rccl.GroupEnd());
    CTRY(ctx0, g_rccl.GroupStart());
    for (int k = 0; k < nb; ++k) {
        mopt_band *b = bands[k];
        mopt_comm *c = b->comm;
        if (c->grp->nranks == 1) continue;
        hipSetDevice(b->ctx->device);
        CTRY(b->ctx, g_rccl.AllGather(b->E + b->first, b->E, (size_t)b->count, ncclDouble, c->nccl, b->cst));      // (in place)
    }
    CTRY(ctx0, g_rccl.GroupEnd());
    return MOPT_OK;
}

static int copy_to_peer(mopt_band *src, mopt_band *dst, const double *from, double *to, size_t count) {
    mopt_ctx *ctx = src->ctx;
    if (src->ctx->device == dst->ctx->device)
        MOPT_HIP_TRY(ctx, hipMemcpyAsync(to, from, count * 8, hipMemcpyDeviceToDevice, src->cst));
    else
        MOPT_HIP_TRY(ctx, hipMemcpyPeerAsync(to, dst->ctx->device, from, src->ctx->device, count * 8, src->cst));
    return MOPT_OK;
}

static int peer_group_check(int nb, mopt_band *const *bands) {
    mopt_comm_group *g = bands[0]->comm->grp;
    if (nb != g->nranks) { bands[0]->ctx->err = "peer transport: all ranks of the group live in one process and must be passed together"; return MOPT_E_ARG; }
    for (int r = 0; r < g->nranks; ++r) if (!g->bands[r]) { bands[0]->ctx->err = "peer transport: a rank has no band attached"; return MOPT_E_STATE; }
    return MOPT_OK;
}

extern "C" int mopt_halo_exchange(int nb, mopt_band *const *bands) {
    if (nb < 1 || !bands) return MOPT_E_ARG;
    for (int k = 0; k < nb; ++k) {
        if (!bands[k] || !bands[k]->comm) return MOPT_E_ARG;
        if (!bands[k]->uploaded) { bands[k]->ctx->err = "mopt_halo_exchange: no inputs uploaded"; return MOPT_E_STATE; }
        if (bands[k]->comm->grp->kind != bands[0]->comm->grp->kind) return MOPT_E_ARG;
    }
    // the exchange streams start behind the uploads of every band
    for (int k = 0; k < nb; ++k) {
        mopt_band *b = bands[k];
        hipSetDevice(b->ctx->device);
        MOPT_HIP_TRY(b->ctx, hipEventRecord(b->ev_in, b->ctx->stream));
    }
    const int kind = bands[0]->comm->grp->kind;
    if (kind == MOPT_COMM_PEER) {
        int rc = peer_group_check(nb, bands);
        if (rc != MOPT_OK) return rc;
        mopt_comm_group *g = bands[0]->comm->grp;
        const int R = g->nranks;
        // a push into a peer's slots must not overtake that peer's own upload (its buffers were swapped there): wait for all
        for (int r = 0; r < R; ++r) {
            mopt_band *b = g->bands[r];
            hipSetDevice(b->ctx->device);
            for (int q = 0; q < R; ++q) MOPT_HIP_TRY(b->ctx, hipStreamWaitEvent(b->cst, g->bands[q]->ev_in, 0));
        }
        for (int r = 0; r < R; ++r) {
            mopt_band *b = g->bands[r];
            hipSetDevice(b->ctx->device);
            mopt_scope sc(b->ctx, "halo", 1, b->cst);
            const int n = b->n;
            const size_t lo = (size_t)b->first * n, hi = (size_t)(b->first + b->count - 1) * n;
            if (r > 0) {
                mopt_band *p = g->bands[r - 1];
                int rc2 = copy_to_peer(b, p, b->X + lo, p->X + lo, n); if (rc2) return rc2;
                rc2 = copy_to_peer(b, p, b->G + lo, p->G + lo, n); if (rc2) return rc2;
            }
            if (r < R - 1) {
                mopt_band *p = g->bands[r + 1];
                int rc2 = copy_to_peer(b, p, b->X + hi, p->X + hi, n); if (rc2) return rc2;
                rc2 = copy_to_peer(b, p, b->G + hi, p->G + hi, n); if (rc2) return rc2;
            }
            for (int q = 0; q < R; ++q)
                if (q != r) { int rc2 = copy_to_peer(b, g->bands[q], b->E + b->first, g->bands[q]->E + b->first, b->count); if (rc2) return rc2; }
        }
        // a band's inputs are complete when every rank's pushes have landed
        std::vector<hipEvent_t> done(R);
        for (int r = 0; r < R; ++r) {
            mopt_band *b = g->bands[r];
            hipSetDevice(b->ctx->device);
            MOPT_HIP_TRY(b->ctx, hipEventRecord(b->ev_dot_done, b->cst));      // (ev_dot_done doubles as "my pushes are out" here; re-recorded by the step)
        }
        for (int r = 0; r < R; ++r) {
            mopt_band *b = g->bands[r];
            hipSetDevice(b->ctx->device);
            for (int q = 0; q < R; ++q) MOPT_HIP_TRY(b->ctx, hipStreamWaitEvent(b->cst, g->bands[q]->ev_dot_done, 0));
            MOPT_HIP_TRY(b->ctx, hipEventRecord(b->ev_out, b->cst));
            b->halo_done = true;
        }
        return MOPT_OK;
    }
    if (!rccl_load()) { bands[0]->ctx->err = g_rccl.err; return MOPT_E_DEVICE; }
    for (int k = 0; k < nb; ++k) {
        mopt_band *b = bands[k];
        hipSetDevice(b->ctx->device);
        MOPT_HIP_TRY(b->ctx, hipStreamWaitEvent(b->cst, b->ev_in, 0));
    }
    std::vector<mopt_scope *> scopes;
    for (int k = 0; k < nb; ++k) { hipSetDevice(bands[k]->ctx->device); scopes.push_back(new mopt_scope(bands[k]->ctx, "halo", 1, bands[k]->cst)); }
    int rc = halo_rccl(nb, bands);
    for (int k = 0; k < nb; ++k) { hipSetDevice(bands[k]->ctx->device); delete scopes[k]; }
    if (rc != MOPT_OK) return rc;
    for (int k = 0; k < nb; ++k) {
        mopt_band *b = bands[k];
        hipSetDevice(b->ctx->device);
        MOPT_HIP_TRY(b->ctx, hipEventRecord(b->ev_out, b->cst));
        b->halo_done = true;
    }
    return MOPT_OK;
}

// the partial sums of FIRE's switch: slot r of every rank's dots[] <- rank r's partial sum
static int exchange_dots(int nb, mopt_band *const *bands) {
    const int kind = bands[0]->comm->grp->kind;
    for (int k = 0; k < nb; ++k) {
        mopt_band *b = bands[k];
        hipSetDevice(b->ctx->device);
        MOPT_HIP_TRY(b->ctx, hipStreamWaitEvent(b->cst, b->ev_dot, 0));
    }
    if (kind == MOPT_COMM_PEER) {
        mopt_comm_group *g = bands[0]->comm->grp;
        const int R = g->nranks;
        for (int r = 0; r < R; ++r) {
            mopt_band *b = g->bands[r];
            hipSetDevice(b->ctx->device);
            for (int q = 0; q < R; ++q)
                if (q != r) { int rc2 = copy_to_peer(b, g->bands[q], b->dots + r, g->bands[q]->dots + r, 1); if (rc2) return rc2; }
            MOPT_HIP_TRY(b->ctx, hipEventRecord(b->ev_in, b->cst));           // (ev_in is free between the halo and the next upload: "my partial sum is out")
        }
        for (int r = 0; r < R; ++r) {
            mopt_band *b = g->bands[r];
            hipSetDevice(b->ctx->device);
            for (int q = 0; q < R; ++q) MOPT_HIP_TRY(b->ctx, hipStreamWaitEvent(b->cst, g->bands[q]->ev_in, 0));
            MOPT_HIP_TRY(b->ctx, hipEventRecord(b->ev_dot_done, b->cst));
        }
        return MOPT_OK;
    }
    bool any = false;
    for (int k = 0; k < nb; ++k) any = any || bands[k]->comm->grp->nranks > 1;
    if (any) CTRY(bands[0]->ctx, g_rccl.GroupStart());
    for (int k = 0; k < nb; ++k) {
        mopt_band *b = bands[k];
        if (b->comm->grp->nranks == 1) continue;
        hipSetDevice(b->ctx->device);
        CTRY(b->ctx, g_rccl.AllGather(b->dots + b->comm->rank, b->dots, 1, ncclDouble, b->comm->nccl, b->cst));
    }
    if (any) CTRY(bands[0]->ctx, g_rccl.GroupEnd());
    for (int k = 0; k < nb; ++k) {
        mopt_band *b = bands[k];
        hipSetDevice(b->ctx->device);
        MOPT_HIP_TRY(b->ctx, hipEventRecord(b->ev_dot_done, b->cst));
    }
    return MOPT_OK;
}

// One band step of all the ranks that live in this process (nbands = 1 for a multi-process job).  Everything is enqueued; the
// partial sums travel while the eigen-solves run.
extern "C" int mopt_band_step_multi(int nb, mopt_band *const *bands, int optimize_num) {
    if (nb < 1 || !bands) return MOPT_E_ARG;
    for (int k = 0; k < nb; ++k) if (!bands[k] || !bands[k]->comm) return MOPT_E_ARG;
    if (bands[0]->comm->grp->kind == MOPT_COMM_PEER) { int rc = peer_group_check(nb, bands); if (rc != MOPT_OK) return rc; }
    for (int k = 0; k < nb; ++k) { int rc = mopt_band_step_phase_a(bands[k], optimize_num); if (rc != MOPT_OK) return rc; }
    { int rc = exchange_dots(nb, bands); if (rc != MOPT_OK) return rc; }
    for (int k = 0; k < nb; ++k) { int rc = mopt_band_step_phase_b(bands[k], optimize_num); if (rc != MOPT_OK) return rc; }
    return MOPT_OK;
}
