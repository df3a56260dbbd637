/*
 * synth.h -- deterministic synthetic "blob field" volumes (SURVEY.md section
 * 8d): the input used by bench.py, the parity tests and the golden fixtures.
 * White noise is deliberately not offered: its extrema density overflows the
 * reference's fixed candidate capacity (R/src_common/MultiScale.cpp:257-258).
 */
#ifndef SIFT3D_SYNTH_H
#define SIFT3D_SYNTH_H
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif
/* Zero volume plus X*Y*Z/2048 Gaussian blobs drawn from the 32-bit LCG
 * s = s*1664525 + 1013904223 (u = (s>>8)/2^24), per blob in this order:
 * cx = u*X, cy = u*Y, cz = u*Z, sigma = 1.5 + 4u, amp = 200(u - 0.3); each blob
 * adds amp*expf(-d^2/(2 sigma^2)) in float over the box (int)c +- ((int)(3 sigma)+1)
 * clipped to the volume, blobs in generation order. */
void sift3d_synth_blobs(float *vol, int64_t X, int64_t Y, int64_t Z, uint32_t seed);
/* Planes [z0, z1) of the same volume into out ((z1 - z0) * Y * X floats): the same bits as those planes of the whole
 * volume -- a rank of the Z-slab run generates its input slices without the 16 GB of a 2048 x 2048 x 1024 volume. */
void sift3d_synth_blobs_slices(float *out, int64_t X, int64_t Y, int64_t Z, uint32_t seed, int64_t z0, int64_t z1);
#ifdef __cplusplus
}
#endif
#endif
