// dae_comm.hip -- the data-parallel collective of the training step, in the C ABI and on the STEP'S OWN STREAM.
//
// Reference step being sharded: autoencoder/autoencoder.py:206-246 (one session.run per mini-batch); SURVEY 8(b) proposed `dae_allreduce_grads`,
// SURVEY 8(e) the exchange: all-reduce(sum) of the flat gradient [dW | dbh | dbv] over xGMI, then the identical optimizer step on every rank.
//
// A `dae_comm` is one RCCL communicator (ncclCommInitRank from a 128-byte unique id the host shares over any out-of-band channel) plus the HIP
// plumbing of the bucketed exchange: one "wire" stream for the collectives and a few timing-free events.  Issuing ncclAllReduce from here instead of
// through torch.distributed's process group removes what profiles/r05_dp_step_breakdown.txt measured: ~25 us of host time per collective call and two
// cross-stream hops around each (the process group runs on its own stream).  With ONE bucket the all-reduce runs on the step's stream itself (no hop at
// all); with several, band k's optimizer pass (dae_plan_apply_band) runs on the step's stream while band k + 1 is on the wire.
//
// RCCL is resolved at dae_comm_init / dae_comm_unique_id time with dlopen, not linked: a single-GPU user never loads it, and a process that already
// runs a copy (torch's process group) shares that copy instead of loading a second one.
#include <dlfcn.h>
#include <rccl/rccl.h>      // types and enums only; every entry point is looked up at run time

#include "dae_common.h"
#include "dae_kernels.h"

namespace dae {

struct Rccl {
    void* dl = nullptr;
    ncclResult_t (*GetUniqueId)(ncclUniqueId*) = nullptr;
    ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*AllReduce)(const void*, void*, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
    const char* (*GetErrorString)(ncclResult_t) = nullptr;
    ncclResult_t (*GetVersion)(int*) = nullptr;
    char path[256] = "";
};

static Rccl g_rccl;

static int rccl_load() {
    Rccl& r = g_rccl;
    if (r.dl) return 0;
    // 1. a copy this process already runs (torch links "librccl.so"; ROCm's own has the soname librccl.so.1): share it
    static const char* const names[] = {"librccl.so", "librccl.so.1", "/opt/rocm/lib/librccl.so.1", "/opt/rocm/lib/librccl.so"};
    for (const char* n : names)
        if ((r.dl = dlopen(n, RTLD_NOW | RTLD_NOLOAD))) { snprintf(r.path, sizeof(r.path), "%s (already loaded)", n); break; }
    // 2. else load ROCm's
    if (!r.dl)
        for (const char* n : names)
            if ((r.dl = dlopen(n, RTLD_NOW | RTLD_LOCAL))) { snprintf(r.path, sizeof(r.path), "%s", n); break; }
    if (!r.dl) {
        set_error("dae_comm: RCCL not found (dlopen librccl.so / librccl.so.1 / /opt/rocm/lib/librccl.so.1: %s)", dlerror());
        return 1;
    }
#define DAE_SYM(field, name)                                                                   \
    r.field = reinterpret_cast<decltype(r.field)>(dlsym(r.dl, name));                          \
    if (!r.field) { set_error("dae_comm: %s lacks %s", r.path, name); r.dl = nullptr; return 1; }
    DAE_SYM(GetUniqueId, "ncclGetUniqueId")
    DAE_SYM(CommInitRank, "ncclCommInitRank")
    DAE_SYM(CommDestroy, "ncclCommDestroy")
    DAE_SYM(AllReduce, "ncclAllReduce")
    DAE_SYM(GetErrorString, "ncclGetErrorString")
    DAE_SYM(GetVersion, "ncclGetVersion")
#undef DAE_SYM
    return 0;
}

#define DAE_CHECK_RCCL(expr)                                                                                         \
    do {                                                                                                             \
        ncclResult_t r__ = (expr);                                                                                   \
        if (r__ != ncclSuccess) {                                                                                    \
            set_error("%s failed: %s (%s:%d)", #expr, g_rccl.GetErrorString(r__), __FILE__, __LINE__);               \
            return 3;                                                                                                \
        }                                                                                                            \
    } while (0)

}  // namespace dae

using namespace dae;

struct dae_comm {
    ncclComm_t comm;
    int rank, world, version;
    hipStream_t wire;                              // the collectives of the bucketed exchange
    hipEvent_t ev_ready;                           // step's stream -> wire: the gradient (band) is complete
    hipEvent_t ev_band[DAE_COMM_MAX_BUCKETS];      // wire -> step's stream: band k has been reduced
};

static_assert(sizeof(ncclUniqueId) == DAE_COMM_ID_BYTES, "ncclUniqueId is 128 bytes");

extern "C" int dae_comm_unique_id(void* id_out) {
    DAE_CHECK_ARG(id_out, "comm_unique_id: null buffer");
    if (rccl_load()) return 1;
    ncclUniqueId id;
    DAE_CHECK_RCCL(g_rccl.GetUniqueId(&id));
    memcpy(id_out, &id, sizeof(id));
    return 0;
}

extern "C" int dae_comm_init(const void* id_in, int32_t rank, int32_t world, dae_comm** out) {
    DAE_CHECK_ARG(id_in && out, "comm_init: null argument");
    DAE_CHECK_ARG(world >= 1 && rank >= 0 && rank < world, "comm_init: rank %d outside [0, %d)", rank, world);
    if (rccl_load()) return 1;
    dae_comm* c = new dae_comm();
    memset(c, 0, sizeof(*c));
    c->rank = rank; c->world = world;
    ncclUniqueId id;
    memcpy(&id, id_in, sizeof(id));
    ncclResult_t r = g_rccl.CommInitRank(&c->comm, world, id, rank);       // on the calling thread's current device
    if (r != ncclSuccess) {
        set_error("ncclCommInitRank(rank %d of %d) failed: %s", rank, world, g_rccl.GetErrorString(r));
        delete c;
        return 3;
    }
    (void)g_rccl.GetVersion(&c->version);
    hipError_t e = hipStreamCreateWithFlags(&c->wire, hipStreamNonBlocking);
    if (e == hipSuccess) e = hipEventCreateWithFlags(&c->ev_ready, hipEventDisableTiming);
    for (int k = 0; k < DAE_COMM_MAX_BUCKETS && e == hipSuccess; ++k) e = hipEventCreateWithFlags(&c->ev_band[k], hipEventDisableTiming);
    if (e != hipSuccess) {
        set_error("comm_init: HIP stream / event creation failed: %s", hipGetErrorString(e));
        dae_comm_destroy(c);
        return 2;
    }
    *out = c;
    return 0;
}

extern "C" void dae_comm_destroy(dae_comm* c) {
    if (!c) return;
    if (c->wire) (void)hipStreamSynchronize(c->wire);
    if (c->comm && g_rccl.CommDestroy) (void)g_rccl.CommDestroy(c->comm);
    if (c->ev_ready) (void)hipEventDestroy(c->ev_ready);
    for (int k = 0; k < DAE_COMM_MAX_BUCKETS; ++k)
        if (c->ev_band[k]) (void)hipEventDestroy(c->ev_band[k]);
    if (c->wire) (void)hipStreamDestroy(c->wire);
    delete c;
}

extern "C" int dae_comm_info(const dae_comm* c, int32_t* out4) {
    DAE_CHECK_ARG(c && out4, "comm_info: null argument");
    out4[0] = c->rank; out4[1] = c->world; out4[2] = c->version; out4[3] = DAE_COMM_MAX_BUCKETS;
    return 0;
}

extern "C" const char* dae_comm_library(void) { return g_rccl.path; }

extern "C" int dae_comm_allreduce_f32(dae_comm* c, float* buf, int64_t n, int32_t op, void* stream) {
    DAE_CHECK_ARG(c && buf && n >= 0, "comm_allreduce_f32: bad arguments");
    DAE_CHECK_ARG(op == 0 || op == 1, "comm_allreduce_f32: op %d (0 = sum, 1 = max)", op);
    if (n == 0) return 0;
    DAE_CHECK_RCCL(g_rccl.AllReduce(buf, buf, (size_t)n, ncclFloat32, op == 0 ? ncclSum : ncclMax, c->comm, (hipStream_t)stream));
    return 0;
}

// Row bands of the bucketed exchange: band k = rows [bounds[k], bounds[k + 1]) of W, whole 64-row blocks, bounds[buckets] = Fp.
// Returns the number of bands actually used (buckets clamped to [1, min(DAE_COMM_MAX_BUCKETS, Fp / 64)]).  Pure host arithmetic.
extern "C" int32_t dae_dp_bands(int32_t Fp, int32_t buckets, int32_t* bounds) {
    const int nblk = Fp / 64;
    int nb = buckets < 1 ? 1 : buckets;
    if (nb > DAE_COMM_MAX_BUCKETS) nb = DAE_COMM_MAX_BUCKETS;
    if (nb > nblk) nb = nblk < 1 ? 1 : nblk;
    if (bounds) {
        for (int k = 0; k < nb; ++k) bounds[k] = 64 * (int32_t)(((int64_t)nblk * k) / nb);
        bounds[nb] = Fp;
    }
    return nb;
}

// Internal (dae_api.hip: dae_allreduce_grads / dae_dp_exchange): the pieces a plan-level exchange is assembled from.
namespace dae {
int comm_allreduce_sum(dae_comm* c, float* buf, int64_t n, hipStream_t st) { return dae_comm_allreduce_f32(c, buf, n, 0, st); }
hipStream_t comm_wire(dae_comm* c) { return c->wire; }
hipEvent_t comm_ev_ready(dae_comm* c) { return c->ev_ready; }
hipEvent_t comm_ev_band(dae_comm* c, int k) { return c->ev_band[k]; }
int comm_world(const dae_comm* c) { return c->world; }
}  // namespace dae
