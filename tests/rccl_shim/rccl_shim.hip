/*
 * rccl_shim.hip -- TEST INFRASTRUCTURE, not part of the product: a rehearsal stand-in for the eight RCCL entry points the
 * C slab driver's RCCL transport uses (3d_sift_cuda_amd/csrc/zslab_transport.hip loads them by name from whatever library
 * sift3d_zslab_set_transport_library names).  The development box has ONE GPU and real RCCL refuses two ranks on one
 * device, so the RCCL half of zs_xfer -- sends and receives collected inside ncclGroupStart / ncclGroupEnd, two communicator
 * sets, the stream each operation is ordered in -- had never executed.  With this library and
 * SIFT3D_ZSLAB_DUPLICATE_RANKS it runs with 2 .. 8 ranks on one device, and the library checks what real RCCL would only
 * punish with a hang:
 *
 *   - every ncclSend has exactly one ncclRecv posted in the SAME group, on the SAME communicator set, by the peer it names,
 *     naming it back, with the same count and type (pairs are matched first-in first-out per (set, source, destination),
 *     as NCCL matches point-to-point operations);
 *   - no send or receive outside a group (one host thread drives every rank: an ungrouped blocking pair cannot progress);
 *   - a communicator is used on ONE stream within a group (what the driver intends; NCCL would serialise otherwise);
 *   - no rank sends to itself; groups are not left open; communicators are not destroyed inside a group.
 *
 * A matched pair is executed with stream-ordered copies that keep RCCL's rendezvous semantics: the copy runs on the
 * receiver's stream behind an event of the sender's stream (the data is final), and the sender's stream then waits for the
 * copy (a send completes when the peer has the data).  A dependency cycle between streams therefore stalls here exactly as
 * it would stall RCCL's kernels.
 *
 * Counters are read through rccl_shim_stats(): tests assert them against the driver's own byte counts.
 */
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>

#include <map>
#include <mutex>
#include <set>
#include <vector>

namespace {
struct shim_comm {
    int set, rank, n, device;
    bool alive;
};
struct shim_op {
    bool send;
    shim_comm *comm;
    int peer;
    void *buf;
    size_t count;
    ncclDataType_t type;
    hipStream_t stream;
    bool matched;
};
struct shim_state {
    std::mutex mu;
    int depth = 0;
    int next_set = 0;
    std::vector<shim_op> ops;
    std::vector<shim_comm *> comms;
    /* counters */
    int64_t groups = 0, sends = 0, recvs = 0, pairs = 0, bytes = 0, unmatched = 0, ungrouped = 0, count_mismatch = 0, self_sends = 0;
    int64_t multi_stream = 0, max_ops_per_group = 0, comm_sets = 0, destroyed_in_group = 0, hip_errors = 0, dead_comm = 0;
    std::map<int, int64_t> bytes_by_set, pairs_by_set;
    char last[256] = "";
};
shim_state G;

size_t type_bytes(ncclDataType_t t)
{
    switch (t) {
    case ncclInt8: case ncclUint8: return 1;
    case ncclFloat16: case ncclBfloat16: return 2;
    case ncclInt32: case ncclUint32: case ncclFloat32: return 4;
    case ncclInt64: case ncclUint64: case ncclFloat64: return 8;
    default: return 0;
    }
}

void note(const char *msg)
{
    snprintf(G.last, sizeof G.last, "%s", msg);
    fprintf(stderr, "rccl_shim: %s\n", msg);
}

#define SHIM_HIP(call)                         \
    do {                                       \
        if ((call) != hipSuccess) {            \
            G.hip_errors++;                    \
            note("HIP call failed: " #call);   \
            return ncclUnhandledCudaError;     \
        }                                      \
    } while (0)

/* the group's operations: pair them, check them, queue the copies */
ncclResult_t run_group()
{
    G.groups++;
    if ((int64_t)G.ops.size() > G.max_ops_per_group) G.max_ops_per_group = (int64_t)G.ops.size();
    ncclResult_t res = ncclSuccess;
    /* one stream per communicator within a group */
    std::map<shim_comm *, std::set<hipStream_t>> streams;
    for (const shim_op &o : G.ops) streams[o.comm].insert(o.stream);
    for (auto &kv : streams)
        if (kv.second.size() > 1) {
            G.multi_stream++;
            note("a communicator was used on more than one stream inside one group");
        }
    int prev_dev = 0;
    (void)hipGetDevice(&prev_dev);
    for (size_t i = 0; i < G.ops.size(); i++) {
        shim_op &s = G.ops[i];
        if (!s.send) continue;
        shim_op *r = nullptr;
        for (size_t j = 0; j < G.ops.size() && !r; j++) {
            shim_op &c = G.ops[j];
            if (!c.send && !c.matched && c.comm->set == s.comm->set && c.comm->rank == s.peer && c.peer == s.comm->rank) r = &c;
        }
        if (!r) continue; /* counted below */
        s.matched = r->matched = true;
        if (r->count != s.count || r->type != s.type) {
            G.count_mismatch++;
            note("a send and its receive disagree on count or type");
            res = ncclInvalidArgument;
            continue;
        }
        const size_t nbytes = s.count * type_bytes(s.type);
        hipEvent_t sent = nullptr, arrived = nullptr;
        SHIM_HIP(hipSetDevice(s.comm->device));
        SHIM_HIP(hipEventCreateWithFlags(&sent, hipEventDisableTiming));
        SHIM_HIP(hipEventRecord(sent, s.stream));
        SHIM_HIP(hipSetDevice(r->comm->device));
        SHIM_HIP(hipEventCreateWithFlags(&arrived, hipEventDisableTiming));
        SHIM_HIP(hipStreamWaitEvent(r->stream, sent, 0));
        if (nbytes) SHIM_HIP(hipMemcpyPeerAsync(r->buf, r->comm->device, s.buf, s.comm->device, nbytes, r->stream));
        SHIM_HIP(hipEventRecord(arrived, r->stream));
        SHIM_HIP(hipSetDevice(s.comm->device));
        SHIM_HIP(hipStreamWaitEvent(s.stream, arrived, 0));
        SHIM_HIP(hipEventDestroy(sent)); /* released by the runtime once the work queued on them is done */
        SHIM_HIP(hipEventDestroy(arrived));
        G.pairs++;
        G.bytes += (int64_t)nbytes;
        G.bytes_by_set[s.comm->set] += (int64_t)nbytes;
        G.pairs_by_set[s.comm->set]++;
    }
    for (const shim_op &o : G.ops)
        if (!o.matched) {
            G.unmatched++;
            note(o.send ? "a send has no receive in its group" : "a receive has no send in its group");
            res = ncclInvalidUsage;
        }
    G.ops.clear();
    (void)hipSetDevice(prev_dev);
    return res;
}

ncclResult_t post(bool send, void *buf, size_t count, ncclDataType_t type, int peer, ncclComm_t comm, hipStream_t stream)
{
    std::lock_guard<std::mutex> lk(G.mu);
    shim_comm *c = (shim_comm *)comm;
    bool known = false;
    for (shim_comm *k : G.comms) known = known || k == c;
    if (!known || !c->alive) {
        G.dead_comm++;
        note("an operation on a communicator this library did not create (or has destroyed)");
        return ncclInvalidArgument;
    }
    if (peer < 0 || peer >= c->n || type_bytes(type) == 0 || (!buf && count)) return ncclInvalidArgument;
    if (send) G.sends++; else G.recvs++;
    if (send && peer == c->rank) {
        G.self_sends++;
        note("a rank sends to itself");
    }
    if (G.depth == 0) {
        G.ungrouped++;
        note("a send or receive outside ncclGroupStart / ncclGroupEnd");
        return ncclInvalidUsage;
    }
    G.ops.push_back({send, c, peer, buf, count, type, stream, false});
    return ncclSuccess;
}
} // namespace

extern "C" {
ncclResult_t ncclGetVersion(int *v)
{
    if (!v) return ncclInvalidArgument;
    *v = -5; /* negative: no RCCL release; the driver's stats carry it, so a test can tell which library ran */
    return ncclSuccess;
}

const char *ncclGetErrorString(ncclResult_t r)
{
    switch (r) {
    case ncclSuccess: return "no error";
    case ncclUnhandledCudaError: return "rccl_shim: a HIP call failed";
    case ncclInvalidArgument: return "rccl_shim: invalid argument";
    case ncclInvalidUsage: return "rccl_shim: invalid usage (see stderr)";
    default: return "rccl_shim: error";
    }
}

ncclResult_t ncclCommInitAll(ncclComm_t *comms, int n, const int *devlist)
{
    std::lock_guard<std::mutex> lk(G.mu);
    if (!comms || n < 1) return ncclInvalidArgument;
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess) return ncclUnhandledCudaError;
    const int set = G.next_set++;
    G.comm_sets++;
    for (int i = 0; i < n; i++) {
        const int d = devlist ? devlist[i] : i;
        if (d < 0 || d >= ndev) return ncclInvalidArgument;
        shim_comm *c = new shim_comm{set, i, n, d, true};
        G.comms.push_back(c);
        comms[i] = (ncclComm_t)c;
    }
    return ncclSuccess;
}

ncclResult_t ncclCommDestroy(ncclComm_t comm)
{
    std::lock_guard<std::mutex> lk(G.mu);
    shim_comm *c = (shim_comm *)comm;
    for (shim_comm *k : G.comms)
        if (k == c && c->alive) {
            if (G.depth > 0) {
                G.destroyed_in_group++;
                note("a communicator destroyed inside an open group");
            }
            c->alive = false; /* the object stays: a late use is then an error, not a crash */
            return ncclSuccess;
        }
    return ncclInvalidArgument;
}

ncclResult_t ncclGroupStart()
{
    std::lock_guard<std::mutex> lk(G.mu);
    G.depth++;
    return ncclSuccess;
}

ncclResult_t ncclGroupEnd()
{
    std::lock_guard<std::mutex> lk(G.mu);
    if (G.depth == 0) {
        note("ncclGroupEnd without ncclGroupStart");
        return ncclInvalidUsage;
    }
    if (--G.depth > 0) return ncclSuccess;
    return run_group();
}

ncclResult_t ncclSend(const void *buf, size_t count, ncclDataType_t type, int peer, ncclComm_t comm, hipStream_t stream)
{
    return post(true, (void *)buf, count, type, peer, comm, stream);
}

ncclResult_t ncclRecv(void *buf, size_t count, ncclDataType_t type, int peer, ncclComm_t comm, hipStream_t stream)
{
    return post(false, buf, count, type, peer, comm, stream);
}

/* ---- the test's side ---- */
enum { SHIM_N_STATS = 20 };
/* out[0..]: groups, sends, recvs, pairs, bytes, unmatched, ungrouped, count_mismatch, self_sends, multi_stream,
 * max_ops_per_group, comm_sets, destroyed_in_group, hip_errors, open_depth, live_comms, sets_with_traffic, bytes of the
 * first set with traffic, bytes of the second, dead_comm */
int rccl_shim_stats(int64_t *out, int n)
{
    std::lock_guard<std::mutex> lk(G.mu);
    int64_t v[SHIM_N_STATS] = {G.groups, G.sends, G.recvs, G.pairs, G.bytes, G.unmatched, G.ungrouped, G.count_mismatch, G.self_sends,
                               G.multi_stream, G.max_ops_per_group, G.comm_sets, G.destroyed_in_group, G.hip_errors, G.depth, 0,
                               0, 0, 0, G.dead_comm};
    for (shim_comm *c : G.comms) v[15] += c->alive ? 1 : 0;
    int k = 0;
    for (auto &kv : G.bytes_by_set) {
        if (kv.second == 0 && G.pairs_by_set[kv.first] == 0) continue;
        v[16]++;
        if (k < 2) v[17 + k] = kv.second;
        k++;
    }
    for (int i = 0; i < n && i < SHIM_N_STATS; i++) out[i] = v[i];
    return SHIM_N_STATS;
}

void rccl_shim_reset(void)
{
    std::lock_guard<std::mutex> lk(G.mu);
    G.groups = G.sends = G.recvs = G.pairs = G.bytes = G.unmatched = G.ungrouped = G.count_mismatch = G.self_sends = 0;
    G.multi_stream = G.max_ops_per_group = G.comm_sets = G.destroyed_in_group = G.hip_errors = G.dead_comm = 0;
    G.bytes_by_set.clear();
    G.pairs_by_set.clear();
    G.last[0] = 0;
}

const char *rccl_shim_last_message(void) { return G.last; }

/* Fault injection: the communicators of the which-th set that is still alive (in creation order) are marked destroyed behind
 * their owner's back, so that its next operation on them is refused in the middle of a group.  Returns how many. */
int rccl_shim_kill_set(int which)
{
    std::lock_guard<std::mutex> lk(G.mu);
    int seen = -1, last_set = -1, target = -1, n = 0;
    for (shim_comm *c : G.comms) {
        if (!c->alive) continue;
        if (c->set != last_set) {
            last_set = c->set;
            if (++seen == which) target = c->set;
        }
    }
    for (shim_comm *c : G.comms)
        if (c->alive && c->set == target) {
            c->alive = false;
            n++;
        }
    return n;
}
}
