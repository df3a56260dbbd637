/*
 * zslab_transport.hip -- how the C slab driver (sift3d_extract_zslab, zslab_driver.hip) moves a block of slices from one rank's
 * device to another's.  Two transports behind one interface:
 *
 *   peer copies  hipMemcpyPeerAsync on the RECEIVER's stream behind the event the sender recorded (rounds 2 - 3);
 *   RCCL         ncclSend on the sender's stream + ncclRecv on the receiver's, all transfers of one exchange step inside
 *                one ncclGroupStart / ncclGroupEnd (the calling thread queues them for every device, so the group is what lets the
 *                pairs progress together) -- BASELINE.json's north star: "halo exchange over RCCL/xGMI".  One communicator
 *                set per CHANNEL: RCCL orders the operations of a communicator, and the deferred patch halos (channel 1)
 *                must not queue in front of the next level's halo (channel 0).
 *
 * The reference has no multi-GPU code (single device: R/cuda_common/SIFT_cuda_Tools.cu:185).
 *
 * RCCL is loaded at run time (dlopen "librccl.so.1"; sift3d_zslab_set_transport_library names another build): the product
 * library does not link it, a process that never asks for this transport never maps it, and inside a PyTorch process the
 * soname resolves to the copy torch already holds.  A device listed twice (the rehearsal of the slab logic on one GPU)
 * cannot be two RCCL ranks: the transport then falls back to peer copies and says so (zs_transport_fell_back).
 * NEVER RUN BETWEEN TWO GPUS: the development box has one.
 */
#include <dlfcn.h>
#include <rccl/rccl.h> /* types and prototypes only: every call goes through the table below */
#include <stdio.h>
#include <string.h>

#include <mutex>
#include <string>
#include <vector>

#include "zslab_transport.h"

namespace {
struct rccl_api {
    void *lib = nullptr;
    ncclResult_t (*CommInitAll)(ncclComm_t *, int, const int *) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*GroupStart)() = nullptr;
    ncclResult_t (*GroupEnd)() = nullptr;
    ncclResult_t (*Send)(const void *, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*Recv)(void *, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
    const char *(*GetErrorString)(ncclResult_t) = nullptr;
    ncclResult_t (*GetVersion)(int *) = nullptr;
};

std::string g_library; /* empty: the default names */

bool load_rccl(rccl_api &a, char *err, size_t err_len)
{
    const char *names[] = {g_library.empty() ? "librccl.so.1" : g_library.c_str(), g_library.empty() ? "librccl.so" : nullptr};
    for (const char *n : names) {
        if (!n) continue;
        a.lib = dlopen(n, RTLD_NOW | RTLD_LOCAL);
        if (a.lib) break;
    }
    if (!a.lib) {
        snprintf(err, err_len, "slab exchange: cannot load the RCCL library %s: %s", names[0], dlerror());
        return false;
    }
    struct { const char *name; void **fn; } syms[] = {
        {"ncclCommInitAll", (void **)&a.CommInitAll}, {"ncclCommDestroy", (void **)&a.CommDestroy}, {"ncclGroupStart", (void **)&a.GroupStart},
        {"ncclGroupEnd", (void **)&a.GroupEnd},       {"ncclSend", (void **)&a.Send},               {"ncclRecv", (void **)&a.Recv},
        {"ncclGetErrorString", (void **)&a.GetErrorString}, {"ncclGetVersion", (void **)&a.GetVersion}};
    for (auto &s : syms) {
        *s.fn = dlsym(a.lib, s.name);
        if (!*s.fn) {
            snprintf(err, err_len, "slab exchange: %s has no symbol %s", names[0], s.name);
            dlclose(a.lib);
            a.lib = nullptr;
            return false;
        }
    }
    return true;
}
} // namespace

struct zs_transport {
    int kind = ZS_TRANSPORT_PEER;
    int fell_back = 0;
    int version = 0;
    std::vector<int> devices;
    rccl_api api;
    std::vector<ncclComm_t> comm[ZS_CHANNELS]; /* with ZS_FLAG_SERIAL_CHANNELS comm[1] is a copy of comm[0]'s handles */
    int comm_sets = 0;
    bool in_group = false;
    std::mutex err_m; /* peer copies are queued by the receiving rank's host thread: two of them may fail at once */
    char err[384] = "";
};

extern "C" void zs_transport_set_library(const char *path) { g_library = path ? path : ""; }

extern "C" zs_transport *zs_transport_create(int kind, int flags, const int *devices, int n, char *err, size_t err_len)
{
    if (err && err_len) err[0] = 0;
    if (!devices || n < 1 || (kind != ZS_TRANSPORT_PEER && kind != ZS_TRANSPORT_RCCL)) {
        if (err && err_len) snprintf(err, err_len, "slab exchange: bad transport arguments");
        return nullptr;
    }
    zs_transport *t = new zs_transport();
    t->devices.assign(devices, devices + n);
    if (kind == ZS_TRANSPORT_PEER) return t;
    for (int i = 0; i < n && !t->fell_back && !(flags & ZS_FLAG_DUPLICATE_RANKS); i++)
        for (int j = 0; j < i; j++)
            if (devices[i] == devices[j]) t->fell_back = 1; /* one device, several ranks: not something RCCL does */
    if (t->fell_back) return t;
    char e[384];
    if (!load_rccl(t->api, e, sizeof e)) {
        if (err && err_len) snprintf(err, err_len, "%s", e);
        delete t;
        return nullptr;
    }
    t->api.GetVersion(&t->version);
    for (int ch = 0; ch < ZS_CHANNELS; ch++) {
        if (ch > 0 && (flags & ZS_FLAG_SERIAL_CHANNELS)) { /* one set: RCCL orders a communicator's operations, so channel 1 queues behind channel 0 */
            t->comm[ch] = t->comm[0];
            continue;
        }
        t->comm_sets++;
        t->comm[ch].assign((size_t)n, nullptr);
        const ncclResult_t r = t->api.CommInitAll(t->comm[ch].data(), n, devices);
        if (r != ncclSuccess) {
            if (err && err_len) snprintf(err, err_len, "slab exchange: ncclCommInitAll over %d devices failed: %s", n, t->api.GetErrorString(r));
            t->comm[ch].clear();
            zs_transport_destroy(t);
            return nullptr;
        }
    }
    t->kind = ZS_TRANSPORT_RCCL;
    return t;
}

extern "C" void zs_transport_destroy(zs_transport *t)
{
    if (!t) return;
    if (t->in_group) zs_xfer_abort(t); /* a communicator is not destroyed inside an open group */
    for (int ch = 0; ch < ZS_CHANNELS; ch++) {
        if (ch > 0 && !t->comm[ch].empty() && !t->comm[0].empty() && t->comm[ch][0] == t->comm[0][0]) continue; /* the serial form's alias */
        for (ncclComm_t c : t->comm[ch])
            if (c) t->api.CommDestroy(c);
    }
    if (t->api.lib) dlclose(t->api.lib);
    delete t;
}

extern "C" int zs_transport_kind(const zs_transport *t) { return t ? t->kind : ZS_TRANSPORT_PEER; }
extern "C" int zs_transport_fell_back(const zs_transport *t) { return t ? t->fell_back : 0; }
extern "C" int zs_transport_version(const zs_transport *t) { return t ? t->version : 0; }
extern "C" int zs_transport_comm_sets(const zs_transport *t) { return t ? t->comm_sets : 0; }
extern "C" const char *zs_transport_error(const zs_transport *t) { return t ? t->err : ""; }

extern "C" int zs_xfer_begin(zs_transport *t)
{
    if (t->kind != ZS_TRANSPORT_RCCL) return 0;
    const ncclResult_t r = t->api.GroupStart();
    if (r != ncclSuccess) {
        snprintf(t->err, sizeof t->err, "slab exchange: ncclGroupStart failed: %s", t->api.GetErrorString(r));
        return -1;
    }
    t->in_group = true;
    return 0;
}

extern "C" int zs_xfer(zs_transport *t, int channel, int src_rank, const float *src, hipStream_t src_stream, hipEvent_t src_ready, int dst_rank,
                       float *dst, hipStream_t dst_stream, size_t nfloats)
{
    const int n = (int)t->devices.size();
    if (channel < 0 || channel >= ZS_CHANNELS || src_rank < 0 || src_rank >= n || dst_rank < 0 || dst_rank >= n || !src || !dst) {
        std::lock_guard<std::mutex> lk(t->err_m);
        snprintf(t->err, sizeof t->err, "slab exchange: bad transfer %d -> %d on channel %d", src_rank, dst_rank, channel);
        return -1;
    }
    hipError_t e;
    if (t->kind == ZS_TRANSPORT_PEER) {
        /* on the receiver's stream, behind the sender's event; the calling thread is the receiving rank's (zslab_driver.hip: zs_crew) */
        if ((e = hipSetDevice(t->devices[(size_t)dst_rank])) != hipSuccess || (e = hipStreamWaitEvent(dst_stream, src_ready, 0)) != hipSuccess ||
            (e = hipMemcpyPeerAsync(dst, t->devices[(size_t)dst_rank], src, t->devices[(size_t)src_rank], sizeof(float) * nfloats, dst_stream)) != hipSuccess) {
            std::lock_guard<std::mutex> lk(t->err_m);
            snprintf(t->err, sizeof t->err, "slab exchange: peer copy %d -> %d failed: %s", src_rank, dst_rank, hipGetErrorString(e));
            return -1;
        }
        return 0;
    }
    /* RCCL: the send goes out on the sender's stream once its slices are final, the receive is posted on the receiver's.  The
     * event waits are stream operations issued now; the two RCCL calls are deferred to zs_xfer_end (ncclGroupEnd), which is
     * behind them in both streams. */
    if ((e = hipSetDevice(t->devices[(size_t)src_rank])) != hipSuccess || (e = hipStreamWaitEvent(src_stream, src_ready, 0)) != hipSuccess) {
        snprintf(t->err, sizeof t->err, "slab exchange: ordering the send %d -> %d failed: %s", src_rank, dst_rank, hipGetErrorString(e));
        return -1;
    }
    ncclResult_t r = t->api.Send(src, nfloats, ncclFloat, dst_rank, t->comm[channel][(size_t)src_rank], src_stream);
    if (r == ncclSuccess) {
        if ((e = hipSetDevice(t->devices[(size_t)dst_rank])) != hipSuccess) {
            snprintf(t->err, sizeof t->err, "slab exchange: hipSetDevice failed: %s", hipGetErrorString(e));
            return -1;
        }
        r = t->api.Recv(dst, nfloats, ncclFloat, src_rank, t->comm[channel][(size_t)dst_rank], dst_stream);
    }
    if (r != ncclSuccess) {
        snprintf(t->err, sizeof t->err, "slab exchange: ncclSend / ncclRecv %d -> %d failed: %s", src_rank, dst_rank, t->api.GetErrorString(r));
        return -1;
    }
    return 0;
}

extern "C" int zs_xfer_end(zs_transport *t)
{
    if (t->kind != ZS_TRANSPORT_RCCL || !t->in_group) return 0;
    t->in_group = false;
    const ncclResult_t r = t->api.GroupEnd();
    if (r != ncclSuccess) {
        snprintf(t->err, sizeof t->err, "slab exchange: ncclGroupEnd failed: %s", t->api.GetErrorString(r));
        return -1;
    }
    return 0;
}

extern "C" void zs_xfer_abort(zs_transport *t)
{
    if (!t || t->kind != ZS_TRANSPORT_RCCL || !t->in_group) return;
    t->in_group = false;
    (void)t->api.GroupEnd(); /* whatever the group had collected is launched or refused; the caller drops the transport afterwards */
}
