/*
 * api_timing.hip -- timing of the launches of a call: event pairs per stage, the launch log, sift3d_enable_timing / sift3d_get_timings / sift3d_get_launch_log
 *
 * One of the five translation units behind include/sift3d.h (round 6: api.hip, 2 300 lines, cut at its seams; no behaviour
 * change): api_context.hip (contexts, buffers, tuning, stream), api_timing.hip (event pairs, the launch log), api_ops.hip
 * (the blur dispatcher, the operator-level entry points, the candidate lists), api_pipeline.hip (volume upload, the
 * per-keypoint stage, run_pipeline, sift3d_extract / sift3d_detect), api_slab.hip (the building blocks a Z-slab driver calls).
 * What they share is pipeline.h.  R/ = /root/reference/3dsift_cleanup-softVote_App_Weight_SoftMax/
 */
#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "sift3d_internal.h"

#include "pipeline.h"

/* ---- timing ------------------------------------------------------------ */
void timing_begin(sift3d_ctx *c)
{
    memset(&c->last, 0, sizeof(c->last));
    c->launches.clear();
    c->pool_used = 0;
    c->resolved = 0;
}

/* Resolves the events of every launch recorded since the last call (idempotent). */
void timing_end(sift3d_ctx *c)
{
    if (!c->timing) return;
    hipStreamSynchronize(c->stream);
    for (size_t i = c->resolved; i < c->launches.size(); i++) {
        timed_launch &t = c->launches[i];
        float ms = 0;
        if (hipEventElapsedTime(&ms, t.e0, t.e1) == hipSuccess) c->last.ms[t.stage] += ms;
        t.ms = ms;
        float since = 0;
        if (hipEventElapsedTime(&since, c->launches.front().e0, t.e0) != hipSuccess) since = 0;
        t.start_ms = since;
    }
    c->resolved = c->launches.size();
    if (!c->launches.empty()) {
        float ms = 0;
        if (hipEventElapsedTime(&ms, c->launches.front().e0, c->launches.back().e1) == hipSuccess) c->last.total_ms = ms;
    }
}

extern "C" int sift3d_enable_timing(sift3d_ctx *c, int on)
{
    if (!c) return SIFT3D_ERR_ARG;
    HIPCHK(c, hipSetDevice(c->device));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    c->timing = on < 0 ? 0 : (on > 3 ? 1 : on);
    timing_begin(c); /* operator-level *_dev calls accumulate from here until the log is read */
    return SIFT3D_OK;
}

extern "C" int sift3d_get_timings(const sift3d_ctx *c, sift3d_timings *t)
{
    if (!c || !t) return SIFT3D_ERR_ARG;
    timing_end(const_cast<sift3d_ctx *>(c));
    *t = c->last;
    return SIFT3D_OK;
}

extern "C" int sift3d_get_launch_log(const sift3d_ctx *c, sift3d_launch_record *out, int64_t cap, int64_t *n)
{
    if (!c || !n) return SIFT3D_ERR_ARG;
    timing_end(const_cast<sift3d_ctx *>(c));
    *n = (int64_t)c->launches.size();
    for (int64_t i = 0; i < *n && i < cap && out; i++) {
        const timed_launch &t = c->launches[(size_t)i];
        out[i].stage = t.stage;
        out[i].ntaps = t.ntaps;
        out[i].nvox = t.nvox;
        out[i].alg_bytes = t.bytes;
        out[i].ms = t.ms;
        out[i].start_ms = t.start_ms;
    }
    return *n > cap ? SIFT3D_ERR_CAPACITY : SIFT3D_OK;
}
