// comm.h -- transports of the row-sharded multi-GPU build.
//
// One communicator per rank (= per GPU).  Every operation is STREAM-ORDERED on the rank's HIP stream and moves DEVICE
// buffers: kernels and exchanges need no host synchronisation between them; the host waits only inside
// comm_gather_counts, where it needs the record counts to size the next exchange.
//   RCCL  : the product path -- ncclGroupStart / ncclSend / ncclRecv / ncclGroupEnd over xGMI, one process per GPU
//           (bench.py under torch.distributed.run) or one host thread per GPU (nnd_build_multi).  librccl is opened
//           with dlopen the first time a communicator is created, so the library loads on boxes without it.
//   LOCAL : ranks are threads of one process and may SHARE a GPU (tests and the critical-path tool on a one-GPU box):
//           senders post their buffers, receivers copy them with hipMemcpyAsync (peer copies between different GPUs),
//           ordered by events across the ranks' streams.
//   HOST  : device buffers are staged through pinned host memory and a caller-supplied callback moves host bytes
//           (gloo in the two-processes-on-one-GPU test).  Debugging transport, never the measured one.
#pragma once
#include <hip/hip_runtime.h>
#include <stdarg.h>
#include <stdint.h>
#include <stdio.h>

#include <atomic>
#include <condition_variable>
#include <mutex>
#include <vector>

#include "../../include/pynnd_amd.h"

enum { NND_COMM_RCCL = 1, NND_COMM_LOCAL = 2, NND_COMM_HOST = 3 };
#define NND_MAX_RANKS 64

struct nnd_local_post {
    void *const *send_bases = nullptr;   // per array
    const size_t *soff = nullptr, *scnt = nullptr;  // per destination, in elements
    const long long *counts_host = nullptr;  // gather_counts: this rank's pinned vector
    int device = 0;
    hipEvent_t ready = nullptr, done = nullptr;  // data produced on the sender's stream / copies finished on the receiver's
};

// shared by the ranks of one LOCAL group
struct nnd_local_group {
    int world = 0;
    std::mutex mu;
    std::condition_variable cv;
    int arrived = 0;
    uint64_t generation = 0;
    bool aborted = false;
    std::atomic<int> abort_flag{0};  // the same, readable without the lock (polled by the host waits of every rank)
    int64_t timeout_ms = 120000;     // a rank that waits longer than this for its peers gives up (and aborts the group)
    int refs = 0;
    nnd_local_post posts[NND_MAX_RANKS];
    // serial mode (critical-path measurement on one GPU): only one rank's compute section runs at a time
    bool serial = false;
    std::mutex gpu_token;
    int barrier();  // 0 ok, 1 aborted
    void abort();
};

struct nnd_comm_s {
    int kind = 0, world = 1, rank = 0, device = 0;
    char err[512] = {0};
    // failure handling: every host wait of a rank (comm_wait) polls `abort_flag` -- shared by the ranks of one process
    // (LOCAL group, the threads of nnd_build_multi); NULL for one process per GPU -- and gives up after timeout_ms.  A rank
    // that gives up under RCCL calls ncclCommAbort on its communicator(s): the collectives queued on its streams are
    // cancelled, the streams drain, and the build returns an error instead of blocking in a collective for ever.
    std::atomic<int> *abort_flag = nullptr;
    int64_t timeout_ms = 120000;
    bool dead = false;                  // aborted: no further collective may be issued, ncclCommDestroy is skipped
    // second channel (own RCCL communicator / LOCAL group and own stream): bulk transfers that overlap the build's stream
    nnd_comm_s *aux = nullptr;
    hipStream_t aux_stream = nullptr;
    bool is_aux = false;
    // RCCL
    void *nccl = nullptr;               // ncclComm_t
    long long *counts_all_dev = nullptr;  // (world, nv) gathered count vectors
    // LOCAL
    nnd_local_group *grp = nullptr;
    // HOST
    nnd_host_exchange_fn fn = nullptr;
    void *user = nullptr;
    unsigned char *h_send = nullptr, *h_recv = nullptr;  // pinned staging, grow-only
    size_t h_send_cap = 0, h_recv_cap = 0;
    // all kinds
    long long *h_counts = nullptr;      // pinned: this rank's vector, then the gathered matrix
    hipEvent_t ev = nullptr;
    int64_t bytes_sent = 0;             // payload bytes this rank sent to OTHER ranks since the last reset (statistics)
    void set_error(const char *fmt, ...) {
        va_list ap;
        va_start(ap, fmt);
        vsnprintf(err, sizeof(err), fmt, ap);
        va_end(ap);
    }
};

// counts_dev: nv <= world + 8 int64 words on this rank's device (written by kernels on `stream`).  On return
// matrix_host[r * nv + i] holds word i of rank r.  This is the one place where the host waits for the device.
int comm_gather_counts(nnd_comm_s *c, hipStream_t stream, const long long *counts_dev, int nv, long long *matrix_host);
// narr arrays exchanged with ONE segmentation: rank d receives elements [soff[d], soff[d] + scnt[d]) of every send base;
// what rank s sends lands at element roff[s] of every recv base (rcnt[s] elements: the caller knows them from
// comm_gather_counts or from the geometry).  Self segment: copied unless send and recv alias (in-place all-gather).
int comm_alltoallv(nnd_comm_s *c, hipStream_t stream, int narr, void *const *send_bases, void *const *recv_bases, const int *elem_bytes,
                   const size_t *soff, const size_t *scnt, const size_t *roff, const size_t *rcnt);
int comm_barrier(nnd_comm_s *c);
// Wait until everything queued on `stream` has run: an event polled by the host, with the abort flag and the timeout of
// the communicator checked while polling (hipStreamSynchronize would block for ever behind a collective whose peer died).
// Returns 0, or 1 with c->err set (and the communicator aborted).
int comm_wait(nnd_comm_s *c, hipStream_t stream, const char *what);
// a peer (or this rank) failed: mark the shared flag, wake LOCAL waiters, cancel RCCL work (ncclCommAbort)
void comm_fail(nnd_comm_s *c);
// 1 when a rank of this process has raised the shared abort flag
static inline int comm_aborted(const nnd_comm_s *c) { return c->dead || (c->abort_flag && c->abort_flag->load(std::memory_order_relaxed)); }
// LOCAL serial mode: bracket a compute section (no-ops otherwise)
void comm_compute_begin(nnd_comm_s *c);
void comm_compute_end(nnd_comm_s *c, hipStream_t stream);
