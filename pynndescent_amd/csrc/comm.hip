// comm.hip -- the three transports of comm.h.
#include "comm.h"

#include <dlfcn.h>
#include <rccl/rccl.h>  // types and prototypes only: librccl is opened with dlopen (load_rccl)
#include <string.h>

#include <chrono>
#include <string>

std::recursive_mutex &nnd_lifecycle_mutex();  // capi.hip (state.h): creation / tear-down of handles is serialised

static thread_local char g_cerr[512] = {0};
static void cgerr(const char *fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_cerr, sizeof(g_cerr), fmt, ap);
    va_end(ap);
}
extern "C" const char *nnd_comm_last_error(nnd_comm_t c) { return c ? c->err : g_cerr; }

// ------------------------------------------------------------------------------------------------ RCCL (dlopen)
struct rccl_api {
    void *lib = nullptr;
    decltype(&ncclGetUniqueId) GetUniqueId = nullptr;
    decltype(&ncclCommInitRank) CommInitRank = nullptr;
    decltype(&ncclCommDestroy) CommDestroy = nullptr;
    decltype(&ncclCommAbort) CommAbort = nullptr;
    decltype(&ncclGetVersion) GetVersion = nullptr;
    decltype(&ncclGroupStart) GroupStart = nullptr;
    decltype(&ncclGroupEnd) GroupEnd = nullptr;
    decltype(&ncclSend) Send = nullptr;
    decltype(&ncclRecv) Recv = nullptr;
    decltype(&ncclAllGather) AllGather = nullptr;
    decltype(&ncclGetErrorString) GetErrorString = nullptr;
};
static rccl_api g_rccl;
static std::mutex g_rccl_mu;

static const rccl_api *load_rccl(std::string &why) {
    std::lock_guard<std::mutex> lk(g_rccl_mu);
    if (g_rccl.lib) return &g_rccl;
    // a process that already carries an RCCL (torch ships one) must use THAT copy: one RCCL per process
    const char *names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1", "/opt/rocm/lib/librccl.so"};
    void *lib = nullptr;
    for (const char *nm : names) {
        lib = dlopen(nm, RTLD_NOW | RTLD_GLOBAL | RTLD_NOLOAD);
        if (lib) break;
    }
    for (const char *nm : names) {
        if (lib) break;
        lib = dlopen(nm, RTLD_NOW | RTLD_GLOBAL);
    }
    if (!lib) {
        const char *de = dlerror();  // (a second call would return NULL: the message is consumed)
        why = std::string("librccl.so could not be opened: ") + (de ? de : "not found");
        return nullptr;
    }
#define RCCL_SYM(field, name)                                                  \
    g_rccl.field = (decltype(g_rccl.field))dlsym(lib, name);                   \
    if (!g_rccl.field) { why = std::string("librccl has no symbol ") + name; return nullptr; }
    RCCL_SYM(GetUniqueId, "ncclGetUniqueId")
    RCCL_SYM(CommInitRank, "ncclCommInitRank")
    RCCL_SYM(CommDestroy, "ncclCommDestroy")
    RCCL_SYM(CommAbort, "ncclCommAbort")
    RCCL_SYM(GetVersion, "ncclGetVersion")
    RCCL_SYM(GroupStart, "ncclGroupStart")
    RCCL_SYM(GroupEnd, "ncclGroupEnd")
    RCCL_SYM(Send, "ncclSend")
    RCCL_SYM(Recv, "ncclRecv")
    RCCL_SYM(AllGather, "ncclAllGather")
    RCCL_SYM(GetErrorString, "ncclGetErrorString")
#undef RCCL_SYM
    g_rccl.lib = lib;
    return &g_rccl;
}

#define C_HIP(expr)                                                                                  \
    do {                                                                                             \
        hipError_t _e = (expr);                                                                      \
        if (_e != hipSuccess) {                                                                      \
            c->set_error("%s failed: %s (%s:%d)", #expr, hipGetErrorString(_e), __FILE__, __LINE__); \
            return 1;                                                                                \
        }                                                                                            \
    } while (0)
#define C_NCCL(expr)                                                                                         \
    do {                                                                                                     \
        ncclResult_t _r = (expr);                                                                            \
        if (_r != ncclSuccess) {                                                                             \
            c->set_error("%s failed: %s (%s:%d)", #expr, g_rccl.GetErrorString(_r), __FILE__, __LINE__);     \
            return 1;                                                                                        \
        }                                                                                                    \
    } while (0)

static int comm_common_init(nnd_comm_s *c) {
    C_HIP(hipHostMalloc((void **)&c->h_counts, sizeof(long long) * (size_t)(NND_MAX_RANKS + 1) * (NND_MAX_RANKS + 8), hipHostMallocDefault));
    C_HIP(hipEventCreateWithFlags(&c->ev, hipEventDisableTiming));
    return 0;
}

extern "C" int32_t nnd_comm_unique_id(void *id_out) {
    static_assert(NND_COMM_ID_BYTES == NCCL_UNIQUE_ID_BYTES, "the id travels as an opaque byte block");
    if (!id_out) { cgerr("nnd_comm_unique_id: null argument"); return 1; }
    std::string why;
    const rccl_api *api = load_rccl(why);
    if (!api) { cgerr("nnd_comm_unique_id: %s", why.c_str()); return 1; }
    ncclUniqueId id;
    const ncclResult_t r = api->GetUniqueId(&id);
    if (r != ncclSuccess) { cgerr("ncclGetUniqueId failed: %s", api->GetErrorString(r)); return 1; }
    memcpy(id_out, &id, NCCL_UNIQUE_ID_BYTES);
    return 0;
}

static nnd_comm_s *make_rccl(const void *id_bytes, int32_t world, int32_t rank, int32_t device) {
    nnd_comm_s *c = new nnd_comm_s();
    c->kind = NND_COMM_RCCL;
    c->world = world;
    c->rank = rank;
    c->device = device;
    ncclUniqueId id;
    memcpy(&id, id_bytes, NCCL_UNIQUE_ID_BYTES);
    ncclComm_t comm = nullptr;
    // (not under the lifecycle lock: ncclCommInitRank returns only when every rank of the world has called it)
    const ncclResult_t r = g_rccl.CommInitRank(&comm, world, id, rank);
    if (r != ncclSuccess || comm_common_init(c) ||
        hipMalloc((void **)&c->counts_all_dev, sizeof(long long) * (size_t)world * (NND_MAX_RANKS + 8)) != hipSuccess) {
        cgerr("nnd_comm_create_rccl: %s", r != ncclSuccess ? g_rccl.GetErrorString(r) : (c->err[0] ? c->err : "allocation failed"));
        if (comm) (void)g_rccl.CommAbort(comm);
        if (c->counts_all_dev) (void)hipFree(c->counts_all_dev);
        if (c->h_counts) (void)hipHostFree(c->h_counts);
        if (c->ev) (void)hipEventDestroy(c->ev);
        delete c;
        return nullptr;
    }
    c->nccl = comm;
    return c;
}

// The first exchange of a new RCCL communicator, done at creation (which is collective anyway; nnd_comm_self_test runs the
// same exchange on any transport -- that is how this function is tested on a one-GPU box): every rank sends its
// rank number to every other rank through the same send / recv group every later exchange uses, waits with the bounded
// wait, and checks what arrived.  RCCL sets its peer-to-peer connections up lazily at the first use: doing that HERE keeps
// the set-up (allocations, IPC handles, proxy threads) away from the first build, where the second channel's transfer is
// already in flight on another stream, and a transport that does not work fails at creation with a message that says so.
static int comm_first_exchange(nnd_comm_s *c, hipStream_t st) {
    const int G = c->world, me = c->rank;
    if (G < 2) return 0;
    int32_t *buf = nullptr;
    C_HIP(hipMalloc((void **)&buf, sizeof(int32_t) * 2 * (size_t)G));
    int32_t h[2 * NND_MAX_RANKS];
    for (int i = 0; i < 2 * G; i++) h[i] = -1;
    h[me] = me;  // [0, G): the word this rank sends (slot me); [G, 2 G): what the ranks sent
    int rc = hipMemcpy(buf, h, sizeof(int32_t) * 2 * (size_t)G, hipMemcpyHostToDevice) != hipSuccess;
    size_t soff[NND_MAX_RANKS], scnt[NND_MAX_RANKS], roff[NND_MAX_RANKS], rcnt[NND_MAX_RANKS];
    for (int p = 0; p < G; p++) {
        soff[p] = (size_t)me;
        scnt[p] = 1;
        roff[p] = (size_t)(G + p);
        rcnt[p] = 1;
    }
    void *sb[1] = {buf}, *rb[1] = {buf};
    const int eb[1] = {4};
    if (!rc) rc = comm_alltoallv(c, st, 1, sb, rb, eb, soff, scnt, roff, rcnt);
    if (!rc) rc = comm_wait(c, st, "first exchange of a new communicator");
    if (!rc) rc = hipMemcpy(h, buf, sizeof(int32_t) * 2 * (size_t)G, hipMemcpyDeviceToHost) != hipSuccess;
    if (!rc)
        for (int p = 0; p < G; p++)
            if (h[G + p] != p) {
                c->set_error("first exchange of a new communicator: rank %d received %d from rank %d", me, h[G + p], p);
                rc = 1;
                break;
            }
    if (rc && !c->err[0]) c->set_error("first exchange of a new communicator: a HIP call failed");
    (void)hipFree(buf);
    c->bytes_sent = 0;
    return rc;
}

extern "C" int32_t nnd_comm_create_rccl(nnd_comm_t *out, const void *id_bytes, int32_t world, int32_t rank, int32_t device) {
    if (!out || !id_bytes || world < 1 || world > NND_MAX_RANKS || rank < 0 || rank >= world) { cgerr("nnd_comm_create_rccl: bad arguments"); return 1; }
    *out = nullptr;
    std::string why;
    if (!load_rccl(why)) { cgerr("nnd_comm_create_rccl: %s", why.c_str()); return 1; }
    if (hipSetDevice(device) != hipSuccess) { cgerr("nnd_comm_create_rccl: hipSetDevice(%d) failed", device); return 1; }
    nnd_comm_s *c = make_rccl(id_bytes, world, rank, device);
    if (!c) return 1;
    if (world > 1) {
        hipStream_t st = nullptr;
        int rc = hipStreamCreateWithFlags(&st, hipStreamNonBlocking) != hipSuccess;
        if (!rc) rc = comm_first_exchange(c, st);
        if (st) (void)hipStreamDestroy(st);
        if (rc) {
            cgerr("nnd_comm_create_rccl: %s", c->err[0] ? c->err : "hipStreamCreate failed");
            (void)nnd_comm_destroy(c);
            return 1;
        }
    }
    *out = c;
    return 0;
}

// A second channel for the bulk transfers that overlap the build (the point-set all-gather): its own RCCL communicator
// (a second unique id, created like the first) and its own stream, so that its send / recv kernels neither queue behind
// nor block the collectives of the build's stream.  Collective: every rank of the world calls it, or none.
extern "C" int32_t nnd_comm_add_channel_rccl(nnd_comm_t c, const void *id_bytes) {
    if (!c || !id_bytes || c->kind != NND_COMM_RCCL) { cgerr("nnd_comm_add_channel_rccl: not an RCCL communicator"); return 1; }
    if (c->aux) return 0;
    if (hipSetDevice(c->device) != hipSuccess) { cgerr("nnd_comm_add_channel_rccl: hipSetDevice(%d) failed", c->device); return 1; }
    nnd_comm_s *a = make_rccl(id_bytes, c->world, c->rank, c->device);
    if (!a) return 1;
    a->is_aux = true;
    a->abort_flag = c->abort_flag;
    a->timeout_ms = c->timeout_ms;
    if (hipStreamCreateWithFlags(&c->aux_stream, hipStreamNonBlocking) != hipSuccess) {
        cgerr("nnd_comm_add_channel_rccl: hipStreamCreate failed");
        (void)nnd_comm_destroy(a);
        return 1;
    }
    if (comm_first_exchange(a, c->aux_stream)) {
        cgerr("nnd_comm_add_channel_rccl: %s", a->err);
        (void)hipStreamDestroy(c->aux_stream);
        c->aux_stream = nullptr;
        (void)nnd_comm_destroy(a);
        return 1;
    }
    c->aux = a;
    return 0;
}

// ------------------------------------------------------------------------------------------------ LOCAL
int nnd_local_group::barrier() {
    std::unique_lock<std::mutex> lk(mu);
    if (aborted) return 1;
    const uint64_t gen = generation;
    if (++arrived == world) {
        arrived = 0;
        generation++;
        cv.notify_all();
        return 0;
    }
    const bool ok = cv.wait_for(lk, std::chrono::milliseconds(timeout_ms), [&] { return generation != gen || aborted; });
    if (!ok) {  // a peer never arrived (it returned without telling anyone, or it is stuck): nobody waits for ever
        aborted = true;
        abort_flag.store(1);
        cv.notify_all();
    }
    return aborted ? 1 : 0;
}
void nnd_local_group::abort() {
    std::lock_guard<std::mutex> lk(mu);
    aborted = true;
    abort_flag.store(1);
    cv.notify_all();
}

static nnd_comm_s *make_local(nnd_local_group *g, int world, int r, int device) {
    nnd_comm_s *c = new nnd_comm_s();
    c->kind = NND_COMM_LOCAL;
    c->world = world;
    c->rank = r;
    c->device = device;
    c->grp = g;
    bool bad = hipSetDevice(c->device) != hipSuccess || comm_common_init(c);
    if (!bad) bad = hipEventCreateWithFlags(&g->posts[r].ready, hipEventDisableTiming) != hipSuccess ||
                    hipEventCreateWithFlags(&g->posts[r].done, hipEventDisableTiming) != hipSuccess;
    g->posts[r].device = c->device;
    if (bad) cgerr("nnd_comm_create_local: rank %d on device %d: %s", r, c->device, c->err[0] ? c->err : "event creation failed");
    c->dead = bad;  // (reported by the caller; the object exists so that nnd_comm_destroy can release what was created)
    return c;
}

extern "C" int32_t nnd_comm_create_local(nnd_comm_t *out, int32_t world, const int32_t *devices) {
    if (!out || world < 1 || world > NND_MAX_RANKS) { cgerr("nnd_comm_create_local: need 1 <= world <= %d", NND_MAX_RANKS); return 1; }
    // two groups: the build's channel and the second channel for overlapped bulk transfers (see nnd_comm_add_channel_rccl)
    nnd_local_group *g = new nnd_local_group(), *ga = new nnd_local_group();
    g->world = ga->world = world;
    g->refs = ga->refs = world;
    for (int r = 0; r < world; r++) out[r] = nullptr;
    for (int r = 0; r < world; r++) {
        const int device = devices ? devices[r] : 0;
        nnd_comm_s *c = make_local(g, world, r, device);
        nnd_comm_s *a = make_local(ga, world, r, device);
        a->is_aux = true;
        c->aux = a;
        c->abort_flag = a->abort_flag = &g->abort_flag;  // one flag for both channels
        out[r] = c;
        bool bad = c->dead || a->dead;
        if (!bad) bad = hipStreamCreateWithFlags(&c->aux_stream, hipStreamNonBlocking) != hipSuccess;
        if (bad) {
            g->refs = ga->refs = r + 1;  // the communicators that exist: the last nnd_comm_destroy below releases the groups
            for (int q = 0; q <= r; q++) { (void)nnd_comm_destroy(out[q]); out[q] = nullptr; }
            return 1;
        }
    }
    return 0;
}

extern "C" int32_t nnd_comm_local_set_serial(nnd_comm_t c, int32_t on) {
    if (!c || c->kind != NND_COMM_LOCAL) { cgerr("nnd_comm_local_set_serial: not a local communicator"); return 1; }
    c->grp->serial = on != 0;
    return 0;
}

// ------------------------------------------------------------------------------------------------ HOST
extern "C" int32_t nnd_comm_create_host(nnd_comm_t *out, int32_t world, int32_t rank, int32_t device, nnd_host_exchange_fn fn, void *user) {
    if (!out || !fn || world < 1 || world > NND_MAX_RANKS || rank < 0 || rank >= world) { cgerr("nnd_comm_create_host: bad arguments"); return 1; }
    *out = nullptr;
    if (hipSetDevice(device) != hipSuccess) { cgerr("nnd_comm_create_host: hipSetDevice(%d) failed", device); return 1; }
    nnd_comm_s *c = new nnd_comm_s();
    c->kind = NND_COMM_HOST;
    c->world = world;
    c->rank = rank;
    c->device = device;
    c->fn = fn;
    c->user = user;
    if (comm_common_init(c)) {
        cgerr("nnd_comm_create_host: %s", c->err);
        delete c;
        return 1;
    }
    *out = c;
    return 0;
}

extern "C" int32_t nnd_comm_destroy(nnd_comm_t c) {
    if (!c) return 0;
    if (c->aux) {
        (void)nnd_comm_destroy(c->aux);
        c->aux = nullptr;
    }
    std::lock_guard<std::recursive_mutex> lifecycle(nnd_lifecycle_mutex());
    (void)hipSetDevice(c->device);
    if (c->aux_stream) (void)hipStreamDestroy(c->aux_stream);
    // an aborted communicator was released by ncclCommAbort already
    if (c->kind == NND_COMM_RCCL && c->nccl && !c->dead && g_rccl.CommDestroy) (void)g_rccl.CommDestroy((ncclComm_t)c->nccl);
    if (c->counts_all_dev) (void)hipFree(c->counts_all_dev);
    if (c->h_counts) (void)hipHostFree(c->h_counts);
    if (c->h_send) (void)hipHostFree(c->h_send);
    if (c->h_recv) (void)hipHostFree(c->h_recv);
    if (c->ev) (void)hipEventDestroy(c->ev);
    if (c->kind == NND_COMM_LOCAL && c->grp) {
        nnd_local_group *g = c->grp;
        if (g->posts[c->rank].ready) (void)hipEventDestroy(g->posts[c->rank].ready);
        if (g->posts[c->rank].done) (void)hipEventDestroy(g->posts[c->rank].done);
        g->posts[c->rank].ready = g->posts[c->rank].done = nullptr;
        bool last;
        {
            std::lock_guard<std::mutex> lk(g->mu);
            last = --g->refs == 0;
        }
        // the shared abort flag lives in the MAIN group: it goes last (the aux communicator of a rank is destroyed first)
        if (last) delete g;
    }
    delete c;
    return 0;
}

// This rank gives up (its own failure, a peer's raised flag, or a timeout).  Idempotent.
void comm_fail(nnd_comm_s *c) {
    if (c->abort_flag) c->abort_flag->store(1);
    for (nnd_comm_s *q : {c, c->aux}) {
        if (!q) continue;
        if (q->kind == NND_COMM_LOCAL && q->grp) q->grp->abort();
        if (q->kind == NND_COMM_RCCL && q->nccl && !q->dead && g_rccl.CommAbort) {
            (void)hipSetDevice(q->device);
            (void)g_rccl.CommAbort((ncclComm_t)q->nccl);  // cancels the collectives queued on this rank's streams
        }
        q->dead = true;
    }
}

extern "C" int32_t nnd_comm_abort(nnd_comm_t c) {  // a rank failed: wake the others up instead of leaving them in a collective
    if (c) comm_fail(c);
    return 0;
}

extern "C" int32_t nnd_comm_set_timeout(nnd_comm_t c, int64_t timeout_ms) {
    if (!c || timeout_ms < 1) { cgerr("nnd_comm_set_timeout: bad arguments"); return 1; }
    for (nnd_comm_s *q : {c, c->aux}) {
        if (!q) continue;
        q->timeout_ms = timeout_ms;
        if (q->kind == NND_COMM_LOCAL && q->grp) {
            std::lock_guard<std::mutex> lk(q->grp->mu);
            q->grp->timeout_ms = timeout_ms;
        }
    }
    return 0;
}

// Collective self test: the exchange a new RCCL communicator does at creation, on this communicator's transport (and on
// its second channel when there is one).  0 = every pair of ranks moved its bytes.
extern "C" int32_t nnd_comm_self_test(nnd_comm_t c) {
    if (!c) { cgerr("nnd_comm_self_test: null communicator"); return 1; }
    if (c->kind == NND_COMM_HOST) { c->set_error("nnd_comm_self_test: the HOST transport has no device-side exchange to test"); return 1; }
    if (hipSetDevice(c->device) != hipSuccess) { c->set_error("nnd_comm_self_test: hipSetDevice(%d) failed", c->device); return 1; }
    hipStream_t st = nullptr;
    if (hipStreamCreateWithFlags(&st, hipStreamNonBlocking) != hipSuccess) { c->set_error("nnd_comm_self_test: hipStreamCreate failed"); return 1; }
    int rc = comm_first_exchange(c, st);
    if (!rc && c->aux) {
        rc = comm_first_exchange(c->aux, c->aux_stream ? c->aux_stream : st);
        if (rc) c->set_error("second channel: %s", c->aux->err);
    }
    // LOCAL: a rank's stream may still hold waits on its peers' events: nobody tears its stream down before all are through
    if (!rc && c->kind == NND_COMM_LOCAL && c->world > 1) rc = comm_barrier(c);
    (void)hipStreamSynchronize(st);
    (void)hipStreamDestroy(st);
    return rc;
}

// out[0] transport (NND_COMM_RCCL = 1, LOCAL = 2, HOST = 3), out[1] world, out[2] ncclGetVersion() (0 unless RCCL),
// out[3] 1 when the second channel exists
extern "C" int32_t nnd_comm_info(nnd_comm_t c, int32_t *out) {
    if (!c || !out) { cgerr("nnd_comm_info: null argument"); return 1; }
    out[0] = c->kind;
    out[1] = c->world;
    out[2] = 0;
    if (c->kind == NND_COMM_RCCL && g_rccl.GetVersion) {
        int v = 0;
        if (g_rccl.GetVersion(&v) == ncclSuccess) out[2] = v;
    }
    out[3] = c->aux ? 1 : 0;
    return 0;
}

int comm_wait(nnd_comm_s *c, hipStream_t stream, const char *what) {
    if (c->dead) { c->set_error("%s: the communicator was aborted", what); return 1; }
    C_HIP(hipEventRecord(c->ev, stream));
    const auto t0 = std::chrono::steady_clock::now();
    hipError_t e;
    unsigned spins = 0;
    while ((e = hipEventQuery(c->ev)) == hipErrorNotReady) {
        if (comm_aborted(c)) {
            comm_fail(c);
            c->set_error("%s: another rank failed (rank %d of %d gave up waiting)", what, c->rank, c->world);
            return 1;
        }
        if ((++spins & 0xFFu) == 0) {
            const int64_t ms = std::chrono::duration_cast<std::chrono::milliseconds>(std::chrono::steady_clock::now() - t0).count();
            if (ms > c->timeout_ms) {
                comm_fail(c);
                c->set_error("%s: rank %d of %d timed out after %lld ms waiting for its peers", what, c->rank, c->world, (long long)ms);
                return 1;
            }
        }
    }
    C_HIP(e);
    return 0;
}

static int host_stage_grow(nnd_comm_s *c, unsigned char **buf, size_t *cap, size_t need) {
    if (need <= *cap) return 0;
    if (*buf) C_HIP(hipHostFree(*buf));
    *buf = nullptr;
    *cap = need + need / 4 + 4096;
    C_HIP(hipHostMalloc((void **)buf, *cap, hipHostMallocDefault));
    return 0;
}

// ------------------------------------------------------------------------------------------------ operations
int comm_gather_counts(nnd_comm_s *c, hipStream_t stream, const long long *counts_dev, int nv, long long *matrix_host) {
    const int G = c->world;
    if (nv > NND_MAX_RANKS + 8) { c->set_error("comm_gather_counts: vector too long"); return 1; }
    if (c->kind == NND_COMM_RCCL) {
        if (G > 1) {
            C_NCCL(g_rccl.AllGather(counts_dev, c->counts_all_dev, (size_t)nv, ncclInt64, (ncclComm_t)c->nccl, stream));
            C_HIP(hipMemcpyAsync(c->h_counts, c->counts_all_dev, sizeof(long long) * (size_t)G * nv, hipMemcpyDeviceToHost, stream));
        } else {
            C_HIP(hipMemcpyAsync(c->h_counts, counts_dev, sizeof(long long) * (size_t)nv, hipMemcpyDeviceToHost, stream));
        }
        if (comm_wait(c, stream, "count exchange")) return 1;
        memcpy(matrix_host, c->h_counts, sizeof(long long) * (size_t)G * nv);
        return 0;
    }
    // LOCAL / HOST: every rank brings its own vector to the host first
    C_HIP(hipMemcpyAsync(c->h_counts, counts_dev, sizeof(long long) * (size_t)nv, hipMemcpyDeviceToHost, stream));
    if (comm_wait(c, stream, "count exchange")) return 1;
    if (c->kind == NND_COMM_LOCAL) {
        nnd_local_group *g = c->grp;
        g->posts[c->rank].counts_host = c->h_counts;
        if (g->barrier()) { c->set_error("local group aborted (another rank failed)"); return 1; }
        for (int r = 0; r < G; r++) memcpy(matrix_host + (size_t)r * nv, g->posts[r].counts_host, sizeof(long long) * (size_t)nv);
        if (g->barrier()) { c->set_error("local group aborted (another rank failed)"); return 1; }  // vectors may be overwritten now
        return 0;
    }
    // HOST: an all-gather through the callback (send the same nv words to everyone)
    int64_t soff[NND_MAX_RANKS], sb[NND_MAX_RANKS], roff[NND_MAX_RANKS], rb[NND_MAX_RANKS];
    for (int r = 0; r < G; r++) {
        soff[r] = 0;
        sb[r] = (int64_t)sizeof(long long) * nv;
        roff[r] = (int64_t)sizeof(long long) * nv * r;
        rb[r] = sb[r];
    }
    if (c->fn(c->user, c->h_counts, soff, sb, matrix_host, roff, rb) != 0) { c->set_error("host exchange callback failed (counts)"); return 1; }
    return 0;
}

int comm_alltoallv(nnd_comm_s *c, hipStream_t stream, int narr, void *const *send_bases, void *const *recv_bases, const int *elem_bytes,
                   const size_t *soff, const size_t *scnt, const size_t *roff, const size_t *rcnt) {
    const int G = c->world, me = c->rank;
    if (comm_aborted(c)) { comm_fail(c); c->set_error("exchange: another rank failed"); return 1; }
    for (int d = 0; d < G; d++)
        if (d != me)
            for (int a = 0; a < narr; a++) c->bytes_sent += (int64_t)scnt[d] * elem_bytes[a];
    // own segment: a device copy on this rank's stream (nothing to do for an in-place all-gather)
    auto self_copy = [&]() -> int {
        for (int a = 0; a < narr; a++) {
            const unsigned char *src = (const unsigned char *)send_bases[a] + soff[me] * elem_bytes[a];
            unsigned char *dst = (unsigned char *)recv_bases[a] + roff[me] * elem_bytes[a];
            if (scnt[me] && src != dst) C_HIP(hipMemcpyAsync(dst, src, scnt[me] * elem_bytes[a], hipMemcpyDeviceToDevice, stream));
        }
        return 0;
    };
    if (c->kind == NND_COMM_RCCL) {
        if (self_copy()) return 1;
        if (G == 1) return 0;
        C_NCCL(g_rccl.GroupStart());
        for (int p = 0; p < G; p++) {
            if (p == me) continue;
            for (int a = 0; a < narr; a++) {
                if (scnt[p]) C_NCCL(g_rccl.Send((const unsigned char *)send_bases[a] + soff[p] * elem_bytes[a], scnt[p] * elem_bytes[a], ncclUint8, p, (ncclComm_t)c->nccl, stream));
                if (rcnt[p]) C_NCCL(g_rccl.Recv((unsigned char *)recv_bases[a] + roff[p] * elem_bytes[a], rcnt[p] * elem_bytes[a], ncclUint8, p, (ncclComm_t)c->nccl, stream));
            }
        }
        C_NCCL(g_rccl.GroupEnd());
        return 0;
    }
    if (c->kind == NND_COMM_LOCAL) {
        nnd_local_group *g = c->grp;
        nnd_local_post &mine = g->posts[me];
        mine.send_bases = send_bases;
        mine.soff = soff;
        mine.scnt = scnt;
        C_HIP(hipEventRecord(mine.ready, stream));  // everything this rank sends has been produced before this point
        if (g->barrier()) { c->set_error("local group aborted (another rank failed)"); return 1; }
        for (int s = 0; s < G; s++) {
            const nnd_local_post &ps = g->posts[s];
            if (ps.scnt[me] != rcnt[s]) { c->set_error("comm_alltoallv: rank %d sends %zu elements to rank %d, which expects %zu", s, ps.scnt[me], me, rcnt[s]); g->abort(); return 1; }
            if (!rcnt[s]) continue;
            if (s != me) C_HIP(hipStreamWaitEvent(stream, ps.ready, 0));
            for (int a = 0; a < narr; a++) {
                const unsigned char *src = (const unsigned char *)ps.send_bases[a] + ps.soff[me] * elem_bytes[a];
                unsigned char *dst = (unsigned char *)recv_bases[a] + roff[s] * elem_bytes[a];
                if (src == dst) continue;
                if (ps.device == c->device) C_HIP(hipMemcpyAsync(dst, src, rcnt[s] * elem_bytes[a], hipMemcpyDeviceToDevice, stream));
                else C_HIP(hipMemcpyPeerAsync(dst, c->device, src, ps.device, rcnt[s] * elem_bytes[a], stream));
            }
        }
        C_HIP(hipEventRecord(mine.done, stream));  // this rank has read what it needed from everybody
        if (g->barrier()) { c->set_error("local group aborted (another rank failed)"); return 1; }
        // a sender must not overwrite its buffers before every reader is done with them
        for (int d = 0; d < G; d++)
            if (d != me && scnt[d]) C_HIP(hipStreamWaitEvent(stream, g->posts[d].done, 0));
        // the posted pointers (soff / scnt live on the caller's stack) must stay valid until every rank has read them
        if (g->barrier()) { c->set_error("local group aborted (another rank failed)"); return 1; }
        return 0;
    }
    // HOST: stage through pinned memory, one callback per array
    for (int a = 0; a < narr; a++) {
        int64_t so[NND_MAX_RANKS], sb[NND_MAX_RANKS], ro[NND_MAX_RANKS], rb[NND_MAX_RANKS];
        size_t stot = 0, rtot = 0;
        for (int r = 0; r < G; r++) {
            so[r] = (int64_t)stot;
            sb[r] = (int64_t)(scnt[r] * elem_bytes[a]);
            stot += scnt[r] * elem_bytes[a];
            ro[r] = (int64_t)rtot;
            rb[r] = (int64_t)(rcnt[r] * elem_bytes[a]);
            rtot += rcnt[r] * elem_bytes[a];
        }
        if (host_stage_grow(c, &c->h_send, &c->h_send_cap, stot) || host_stage_grow(c, &c->h_recv, &c->h_recv_cap, rtot)) return 1;
        for (int r = 0; r < G; r++)
            if (sb[r]) C_HIP(hipMemcpyAsync(c->h_send + so[r], (const unsigned char *)send_bases[a] + soff[r] * elem_bytes[a], (size_t)sb[r], hipMemcpyDeviceToHost, stream));
        C_HIP(hipStreamSynchronize(stream));
        if (c->fn(c->user, c->h_send, so, sb, c->h_recv, ro, rb) != 0) { c->set_error("host exchange callback failed"); return 1; }
        for (int r = 0; r < G; r++)
            if (rb[r]) C_HIP(hipMemcpyAsync((unsigned char *)recv_bases[a] + roff[r] * elem_bytes[a], c->h_recv + ro[r], (size_t)rb[r], hipMemcpyHostToDevice, stream));
        C_HIP(hipStreamSynchronize(stream));  // the staging buffer is reused by the next array
    }
    return 0;
}

int comm_barrier(nnd_comm_s *c) {
    if (c->kind == NND_COMM_LOCAL) {
        if (c->grp->barrier()) { c->set_error("local group aborted (another rank failed)"); return 1; }
        return 0;
    }
    if (c->kind == NND_COMM_HOST) {
        int64_t z[NND_MAX_RANKS] = {0};
        // send == recv == NULL is the transport's barrier (a data exchange in which some rank moves no byte is NOT one:
        // with three or more ranks the others would be in point-to-point calls while that rank sits in a barrier)
        if (c->fn(c->user, nullptr, z, z, nullptr, z, z) != 0) { c->set_error("host exchange callback failed (barrier)"); return 1; }
    }
    return 0;  // RCCL: the collectives order the ranks; nothing to do
}

void comm_compute_begin(nnd_comm_s *c) {
    if (c->kind == NND_COMM_LOCAL && c->grp->serial) c->grp->gpu_token.lock();
}
void comm_compute_end(nnd_comm_s *c, hipStream_t stream) {
    if (c->kind == NND_COMM_LOCAL && c->grp->serial) {
        (void)comm_wait(c, stream, "compute section");  // polled: a blocking wait wakes up ~1 ms late and would be charged to the section
        c->grp->gpu_token.unlock();
    }
}
