/*
 * pipeline.h -- what the translation units of libsift3d_hip.so share about a context: the sift3d_ctx structure and the
 * building blocks of the api_*.hip translation units (blur, candidate lists, the per-keypoint stage) that the one-process Z-slab driver
 * (zslab_driver.hip) strings together per rank.  Internal: nothing here is part of the C-ABI (include/sift3d.h).
 */
#ifndef SIFT3D_PIPELINE_H
#define SIFT3D_PIPELINE_H
#include <cstdint>
#include <vector>

#include "sift3d_internal.h"

#define SIFT3D_KP_MAX_CHUNKS 16
/* One chunk by default: measured at 512^3 (tools/kp_chunks.py, profiles/r03_kp_chunks.txt) 1: 10.33, 2: 10.31, 3: 10.47, 4: 10.47,
 * 6: 10.73, 8: 10.99, 16: 11.20 ms per extraction -- seven keypoint workgroups fill a CU's LDS (7 x 23 KB), so a descriptor
 * workgroup only becomes resident where a keypoint workgroup has retired, and the two kernels take turns instead of sharing. */
#define SIFT3D_KP_DEFAULT_CHUNKS 1
#define SIFT3D_D4TINY_FLOATS 32768 /* room for the octaves of at most SIFT3D_TINY_VOX voxels of one volume, pitched rows included */

struct timed_launch {
    int stage;
    hipEvent_t e0, e1;
    int ntaps;
    int64_t nvox;
    double bytes;
    float ms;
    float start_ms;
};

/* One detection level: which buffers, which dims, which slices to keep */
struct level_job {
    const float *dp, *dc, *dn;
    int64_t X, Y, Z; /* X is the row pitch of the buffers */
    int z_lo, z_hi;
    int lvl_id;
    int64_t Xl;      /* logical row length (0: same as X) */
    /* neighbour levels that are not stored (sift3d_extrema_lazy): the level below is dp - prev_b; the level above is
     * next_g - blur(next_g, next_taps), evaluated around the candidates only (dn is NULL then) */
    const float *prev_b = nullptr, *next_g = nullptr;
    float next_taps[2 * SIFT3D_FAST_MAX_R + 1] = {};
    int next_ntaps = 0;
};

struct octave_dims {
    int64_t X, Y, Z, off; /* dims and float offset of this octave inside every level buffer */
    int64_t XP;           /* row pitch: X rounded up to whole 16-byte vectors (the pad columns stay zero) */
};

struct sift3d_ctx {
    int device;
    hipStream_t stream;
    hipStream_t ex_stream;     /* extrema detection of an octave, overlapped with the blurs of the coarser octaves */
    hipStream_t cand_stream;   /* where cand_append launches: stream, or ex_stream inside run_pipeline */
    hipStream_t ex_stream2;    /* extrema of the octaves after the first */
    hipEvent_t ev_ex2[2];      /* levels of such an octave complete / its extrema launches complete */
    hipEvent_t ev_reset;       /* the counters of the extrema passes have been cleared (on ex_stream) */
    sift3d_survivor *surv2;    /* own-level list of that stream (the passes of one stream share a list, one after the other) */
    int64_t surv2_cap;
    int surv_sel;              /* which list cand_append uses: 0 = surv, 1 = surv2 */
    hipStream_t kp_stream;     /* descriptor launches of the chunked per-keypoint stage, beside the keypoint kernel of the next chunk */
    hipEvent_t ev_kpc[SIFT3D_KP_MAX_CHUNKS]; /* chunk i's keypoint kernel, scan and record map are complete */
    hipEvent_t ev_desc;        /* the descriptor launches on kp_stream are complete */
    hipEvent_t ev_split[3];    /* split tail (run_pipeline): first-part extrema done on the two extrema streams; its count on the host */
    unsigned long long *h_cnt0; /* pinned, 16 + SIFT3D_KP_MAX_CHUNKS words: [4..7] the small read-backs the host waits for (extrema counts,
                                 * keypoint count), [8..] the record totals of the chunks, [8 + SIFT3D_KP_MAX_CHUNKS ..] the counters of the
                                 * split tail: a copy into pageable memory goes through a staging buffer and costs tens of microseconds more */
    hipEvent_t ev_oct[2];      /* octave's DoG levels complete / extrema launches complete */
    hipEvent_t ev_fence[2];    /* ordering of the *_dev entry points with the legacy default stream (fence_in / fence_out) */
    bool own_stream;
    int64_t capN;   /* voxels of the largest volume */
    int64_t capTot; /* floats per level buffer: all octaves of a capN volume back to back */
    float *vol;   /* input volume */
    float *L[6];  /* Gaussian levels, every octave resident (octave o at offset off_o); L[5] is never stored and stays NULL */
    float *D[5];  /* DoG levels, same layout; D[4] is only allocated when an octave has to store its last DoG level in full
                   * (ensure_level_buffer): by default that level is evaluated around the candidates only */
    float *D4tiny; /* the last DoG level of the octaves that one workgroup builds whole (at most SIFT3D_TINY_VOX voxels each) */
    float *T[2];  /* x- and y-pass intermediates */
    float *d_taps;
    /* extrema as (key, value) pairs, unsorted (a) and sorted (b) */
    unsigned long long *keys_a, *keys_b;
    sift3d_cval *vals_a, *vals_b;
    int64_t cand_cap;
    unsigned long long *d_count; /* [0] validated extrema, [1] own-level survivors of the level in flight, [2] survivor overflow high-water
                                  * mark, [3] keypoints, [4] validated extrema of the second group (split tail) */
    int64_t cand_split_at;       /* 0: one list in keys_a / vals_a; n > 0: entries [0, n) take the first group's extrema (counter [0]), [n, cand_cap)
                                  * the second group's (counter [4]) -- run_pipeline's split tail */
    int cand_group;              /* the group cand_append's launches append to */
    sift3d_survivor *surv;
    sift3d_survivor2 *list2[2];      /* extrema that passed the level below, waiting for the lazily evaluated level above: one list
                                      * per extrema stream (surv_sel) */
    int64_t list2_cap[2];
    unsigned long long *list2_counts; /* one length word per extrema pass (SIFT3D_SURV_SETS), zeroed with surv_counts */
    unsigned long long *surv_counts; /* segment counters of the own-level list: SIFT3D_SURV_SETS sets */
    int surv_set;                    /* next unused set since the last reset */
    int64_t surv_cap;
    int surv_div; /* own-level extrema expected per level: voxels / surv_div (+ slack); 1 after an overflow */
    void *sort_tmp;
    size_t sort_tmp_bytes;
    void *scan_tmp;
    size_t scan_tmp_bytes;
    sift3d_level *d_levels;
    sift3d_dkp *kps;
    float *patch0; /* identity-frame patches of the extrema, kps_cap x 1331 floats */
    int *sampler_tokens; /* per-CU counters of the descriptor kernel's sampling phase (zero whenever no kernel runs) */
    int *d_rec_base;     /* chunked per-keypoint stage: first record of chunk i (SIFT3D_KP_MAX_CHUNKS + 1 ints; [n] = total) */
    int *nrec, *offs; /* per-candidate record count and exclusive prefix */
    int64_t kps_cap;
    int *rec_kp, *rec_frame;
    int64_t recs_cap;       /* record slots of rec_kp / rec_frame: kps_cap * (1 + SIFT3D_MAX_FRAMES), the worst case */
    int64_t capT;           /* floats each of T[0], T[1] holds */
    int64_t hrecs_cap;      /* records the two pinned host buffers below hold: a few per candidate, grown when a run needs more */
    sift3d_feature *h_recs; /* pinned host memory the descriptor kernel stores its records into; reused from call to call */
    int *h_group;           /* per record: level id * 2 + is_max (pinned host) */
    sift3d_feature *d_hrecs; /* the device's addresses of the two */
    int *d_hgroup;
    struct {                /* the per-keypoint stage in flight (describe_queue / _launch / _finish) */
        sift3d_kp_params p;
        float taps5[SIFT3D_MAX_TAPS];
        int64_t ncand, nrec;
        int nchunks, launched;
        int64_t first[SIFT3D_KP_MAX_CHUNKS + 1];
        bool split; /* the chunks were queued on kp_stream by the split tail: the first one's descriptor launch goes to the main stream */
    } kp;
    struct {                /* records placed straight into a list several contexts share (the slab driver; describe_placement) */
        bool counts;            /* describe_queue also counts the records per group (d_counts -> h_counts, behind ev_kpc[0]) */
        int *d_counts, *h_counts, *d_shift; /* SIFT3D_GROUPS ints each: device, pinned host, device */
        sift3d_feature *dst;    /* device-visible address of the shared list (NULL: the context's own pinned buffer) */
    } place;
    int dev_stop;           /* -DSIFT3D_DEV builds: sift3d_dev_set_stop */
    bool count_queued;      /* cand_count_queue ran and nothing was appended since */
    std::vector<struct level_job> jobs; /* extrema launches since the last reset (replayed if the buffer must grow) */
    int64_t nx, ny, nz;
    int64_t pad_nx, pad_ny, pad_nz; /* geometry the pad columns of the level buffers were last cleared for */
    bool has_volume;
    struct {                /* sift3d_set_volume_begin / _planes / _end: a volume arriving in runs of planes */
        bool open;
        int64_t nx, ny, nz, got; /* dims of what arrives; planes received so far */
        int resize;
        std::vector<bool> seen; /* per plane: has it arrived (a plane twice, or runs that overlap, are refused) */
    } up;
    int max_octaves; /* 0: the reference's only stop rule (a dimension <= 2); n > 0: at most n octaves */
    int tune[SIFT3D_TUNE_COUNT]; /* sift3d_set_tuning */
    int64_t host_grows;          /* times describe_launch had to grow the pinned record buffers (tests) */
    int staged = 0;              /* sift3d_describe_dev_counts has run, sift3d_describe_dev_place has not */
    bool lean;       /* a slab context: the caller owns the level buffers, none are allocated here */
    int timing; /* 0 off; 1 every launch bracketed by events; 2 only the blur launches of the finest octave */
    std::vector<timed_launch> launches;
    std::vector<hipEvent_t> pool;
    size_t pool_used;
    size_t resolved; /* launches whose events have been read */
    sift3d_timings last;
    char err[512];
};

/* shared between the translation units of the library, not exported from it */
#pragma GCC visibility push(hidden)
int set_err(sift3d_ctx *c, int code, const char *fmt, ...);

#define HIPCHK(c, call)                                                                                        \
    do {                                                                                                       \
        hipError_t e_ = (call);                                                                                \
        if (e_ != hipSuccess)                                                                                  \
            return set_err((c), SIFT3D_ERR_DEVICE, "%s failed: %s (%s:%d)", #call, hipGetErrorString(e_), __FILE__, \
                           __LINE__);                                                                          \
    } while (0)

static inline int64_t pitch_of(int64_t X) { return (X + 3) / 4 * 4; }

/* entry points that work in the context's own level buffers: not on a slab context, which has none */
#define NEED_LEVELS(c)                                                                                                  \
    do {                                                                                                                \
        if ((c) && (c)->lean) return set_err((c), SIFT3D_ERR_ARG, "%s needs a full context (sift3d_create), not a slab context", __func__); \
    } while (0)

/* runs op between the two fences */
#define FENCED(c, op)                 \
    do {                              \
        int rc_ = fence_in(c);        \
        if (rc_) return rc_;          \
        rc_ = (op);                   \
        if (rc_) return rc_;          \
        return fence_out(c);          \
    } while (0)


/* lean: a slab context (sift3d_create_slab): the caller owns the level buffers */
sift3d_ctx *ctx_create(int device, int64_t nx, int64_t ny, int64_t nz, bool lean);
void timing_begin(sift3d_ctx *c);
/* out = blur(in); dog = in - out when not NULL (api_ops.hip) */
int blur_dev(sift3d_ctx *c, const float *in, float *out, float *dog, int64_t X, int64_t Y, int64_t Z, float sigma, float min_value,
             float *sub = nullptr, bool *sub_done = nullptr);
bool blur_window_supported(int64_t X, int64_t Y, float sigma, float min_value);
int blur_window_dev(sift3d_ctx *c, const float *in, float *out, float *dog, int64_t X, int64_t Y, int64_t Z, int64_t zo0, int64_t zo1, float sigma,
                    float min_value);
/* the candidate lists of a run: reset, one detection level appended, the count requested, awaited (+ sort) */
int cand_reset(sift3d_ctx *c, hipStream_t on = nullptr);
int cand_append(sift3d_ctx *c, const level_job &j, bool record);
int cand_count_queue(sift3d_ctx *c);
int cand_finalize(sift3d_ctx *c, int64_t *count_out);
/* the per-keypoint stage in three phases, so that a driver with several contexts can queue it on all of them before it
 * waits for any */
int describe_queue(sift3d_ctx *c, const std::vector<sift3d_level> &levels, int64_t ncand, int desc_mode, float eig_thres, float size_factor,
                   bool levels_on_device);
/* between describe_queue and describe_launch, for a context whose records go into a list shared with other contexts:
 * describe_group_counts waits for the keypoint side of the stage and returns the records per group (SIFT3D_GROUPS ints, valid
 * until the next run; needs describe_want_group_counts(c, true) before describe_queue); describe_placement names the shared
 * list (its address as this context's device sees it) and the per-group shift of this context's records in it */
void describe_want_group_counts(sift3d_ctx *c, bool on);
int describe_group_counts(sift3d_ctx *c, const int **counts, int64_t *total);
int describe_placement(sift3d_ctx *c, sift3d_feature *shared_list, const int *shift);
int describe_launch(sift3d_ctx *c);
int describe_finish(sift3d_ctx *c, int64_t *n_out);

/* ---- shared by the api_*.hip translation units (round 6: api.hip cut at its seams) ---- */
/* an event of the context's pool (timed launches) */
static inline hipEvent_t get_event(sift3d_ctx *c)
{
    if (c->pool_used == c->pool.size()) {
        hipEvent_t e;
        hipEventCreate(&e);
        c->pool.push_back(e);
    }
    return c->pool[c->pool_used++];
}

struct stage_scope {
    sift3d_ctx *c;
    int stage;
    hipEvent_t e0, e1;
    int ntaps;
    int64_t nvox;
    double bytes;
    hipStream_t st;
    bool timed;
    stage_scope(sift3d_ctx *c_, int stage_, double bytes_, int ntaps_ = 0, int64_t nvox_ = 0, hipStream_t st_ = nullptr)
        : c(c_), stage(stage_), e0(nullptr), e1(nullptr), ntaps(ntaps_), nvox(nvox_), bytes(bytes_), st(st_ ? st_ : c_->stream)
    {
        c->last.launches[stage] += 1;
        c->last.alg_bytes[stage] += bytes;
        /* events cost a few microseconds each (two per launch, ~170 launches: 1 ms of a 13 ms run at 512^3): mode 2
         * keeps them to the dominant kernels, the blur launches on the full-size volume */
        timed = c->timing == 1 || c->timing == 3 ||
                (c->timing == 2 && nvox == pitch_of(c->nx) * c->ny * c->nz &&
                 (stage == SIFT3D_STAGE_BLUR_FUSED || stage == SIFT3D_STAGE_BLUR_X || stage == SIFT3D_STAGE_BLUR_Y ||
                  stage == SIFT3D_STAGE_BLUR_Z_DOG));
        if (timed) {
            e0 = get_event(c);
            e1 = get_event(c);
            hipEventRecord(e0, st);
        }
    }
    void add_bytes(double more) /* the launch turned out to do more (the subsample riding on the level-3 blur) */
    {
        c->last.alg_bytes[stage] += more;
        bytes += more;
    }
    void cancel() /* the launch did not happen */
    {
        c->last.launches[stage] -= 1;
        c->last.alg_bytes[stage] -= bytes;
        if (timed) c->pool_used -= 2;
        stage = -1;
    }
    ~stage_scope()
    {
        if (stage < 0) return;
        if (timed) {
            hipEventRecord(e1, st);
            c->launches.push_back({stage, e0, e1, ntaps, nvox, bytes, 0.0f, 0.0f});
        }
    }
};

/* where the extrema passes of the current group append (the whole list, or its first / second part: SIFT3D_TUNE_SPLIT_TAIL) */
struct cand_target {
    unsigned long long *keys;
    sift3d_cval *vals;
    unsigned long long *count;
    int64_t cap;
};
static inline cand_target cand_target_of(const sift3d_ctx *c)
{
    if (c->cand_split_at <= 0) return {c->keys_a, c->vals_a, c->d_count, c->cand_cap};
    if (c->cand_group == 0) return {c->keys_a, c->vals_a, c->d_count, c->cand_split_at};
    return {c->keys_a + c->cand_split_at, c->vals_a + c->cand_split_at, c->d_count + 4, c->cand_cap - c->cand_split_at};
}

std::vector<octave_dims> octave_list(int64_t X, int64_t Y, int64_t Z);            /* api_context.hip */
int alloc_cands(sift3d_ctx *c, int64_t cap);                                      /* api_context.hip */
int ensure_T(sift3d_ctx *c, int64_t floats);                                      /* api_context.hip */
void timing_end(sift3d_ctx *c);                                                   /* api_timing.hip */
int check_shape(sift3d_ctx *c, int64_t nx, int64_t ny, int64_t nz);               /* api_ops.hip */
int fence_in(sift3d_ctx *c);                                                      /* api_ops.hip */
int fence_out(sift3d_ctx *c);                                                     /* api_ops.hip */
int cand_replay(sift3d_ctx *c);                                                   /* api_ops.hip */
int surv_make_room(sift3d_ctx *c, unsigned long long high_water);                 /* api_ops.hip */
int describe_sorted(sift3d_ctx *c, const std::vector<sift3d_level> &levels, int64_t ncand, int desc_mode, float eig_thres, float size_factor, int64_t *n_out, bool levels_on_device = false);    /* api_pipeline.hip */
int candidates_to_host(sift3d_ctx *c, const std::vector<sift3d_level> &levels, int64_t ncand, sift3d_candidate **cands_out, int64_t *n_out);    /* api_pipeline.hip */
#pragma GCC visibility pop
#endif
