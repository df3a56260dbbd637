/*
 * zslab_transport.h -- rank-to-rank block transfers of the C slab driver: peer copies or RCCL (see zslab_transport.hip).
 * Internal to the library (zslab_driver.hip is the only caller); C linkage so that a test can reach it through dlsym if needed.
 */
#ifndef SIFT3D_ZSLAB_TRANSPORT_H
#define SIFT3D_ZSLAB_TRANSPORT_H
#include <hip/hip_runtime.h>
#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif
#define ZS_TRANSPORT_PEER 0
#define ZS_TRANSPORT_RCCL 1
#define ZS_CHANNELS 2 /* 0: what the next launch waits for (level halos, the gathered octave); 1: the deferred patch halos */

typedef struct zs_transport zs_transport;
#define ZS_FLAG_SERIAL_CHANNELS 1  /* RCCL: ONE communicator set; channel 1's transfers go through channel 0's communicators, behind
                                    * whatever channel 0 has queued (the fallback to try when two communicators per device stall) */
#define ZS_FLAG_DUPLICATE_RANKS 2  /* RCCL: a device listed twice is NOT turned into peer copies: the list goes to ncclCommInitAll as it
                                    * is (real RCCL refuses it; the rehearsal library of tests/rccl_shim takes it) */
/* devices[rank]; NULL + text on failure (RCCL cannot be loaded, ncclCommInitAll fails).  RCCL over a device list with
 * duplicates yields a peer-copy transport with zs_transport_fell_back() == 1 unless ZS_FLAG_DUPLICATE_RANKS is set. */
zs_transport *zs_transport_create(int kind, int flags, const int *devices, int n, char *err, size_t err_len);
void zs_transport_destroy(zs_transport *t);
int zs_transport_kind(const zs_transport *t);
int zs_transport_fell_back(const zs_transport *t);
int zs_transport_version(const zs_transport *t); /* ncclGetVersion, 0 for peer copies */
int zs_transport_comm_sets(const zs_transport *t); /* communicator sets in use: 0 peer copies, 2, or 1 with ZS_FLAG_SERIAL_CHANNELS */
const char *zs_transport_error(const zs_transport *t);
void zs_transport_set_library(const char *path); /* NULL: librccl.so.1, then librccl.so */

/* One exchange step: begin, any number of transfers, end.  A transfer moves nfloats from src (on src_rank's device, final
 * once src_ready has fired; src_stream is a stream of that device the send may occupy) to dst (dst_rank's device), ordered
 * in dst_stream.  Nothing of a step is guaranteed to be queued before zs_xfer_end returns (RCCL defers to the group's
 * end), so events that are to fire behind the arrivals are recorded after it.  Returns 0 or -1 (zs_transport_error). */
int zs_xfer_begin(zs_transport *t);
int zs_xfer(zs_transport *t, int channel, int src_rank, const float *src, hipStream_t src_stream, hipEvent_t src_ready, int dst_rank, float *dst,
            hipStream_t dst_stream, size_t nfloats);
int zs_xfer_end(zs_transport *t);
/* after a failed step: closes a group that zs_xfer_begin opened and the failure left open (result ignored); no-op otherwise */
void zs_xfer_abort(zs_transport *t);
#ifdef __cplusplus
}
#endif
#endif
