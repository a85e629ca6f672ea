/*
 * nvqa_rng.h -- counter-based dropout masks shared by the CPU oracle and the
 * HIP kernels (same bits on both sides, so training parity can be checked
 * with dropout active).  Replaces nn.Dropout's MT19937 bernoulli stream
 * (002_train_baseline.lua:143,153; misc/LSTM.lua:37; misc/netdef.lua:10-11),
 * which cannot be reproduced outside Torch7.
 *
 * keep(seed, step, site, idx) = u >= p with u uniform in [0,1) from a
 * splitmix64-style mix; kept elements are scaled by 1/(1-p) (nn.Dropout v2).
 *
 * Sites and element indices (b = sample row in the caller's batch order,
 * t = 0-based column of the question buffer, independent of the length sort):
 *   NVQA_SITE_EMB  arch1 word embedding   idx = (b*T + t)*E + e
 *   NVQA_SITE_LSTM input of layer l >= 2  idx = (((l-2)*B + b)*Tsteps + t)*R + j
 *                  (arch2: Tsteps = T+2 and t is the encoder step)
 *   NVQA_SITE_Q    question vector        idx = b*(2RL) + j   (arch2: b*R + j, head dropout)
 *   NVQA_SITE_V    image feature          idx = b*I + j
 *   NVQA_SITE_Z    fused vector           idx = b*C + j
 */
#ifndef NVQA_RNG_H
#define NVQA_RNG_H

#include <stdint.h>

#if defined(__HIPCC__)
#define NVQA_HD __host__ __device__ __forceinline__
#else
#define NVQA_HD static inline
#endif

#define NVQA_SITE_EMB 0u
#define NVQA_SITE_LSTM 1u
#define NVQA_SITE_Q 2u
#define NVQA_SITE_V 3u
#define NVQA_SITE_Z 4u

NVQA_HD uint32_t nvqa_hash32(uint64_t seed, uint64_t step, uint32_t site, uint64_t idx)
{
    uint64_t x = seed ^ (0x9E3779B97F4A7C15ULL * (step + 1ULL)) ^ ((uint64_t)site << 56);
    x += idx * 0xD1342543DE82EF95ULL;
    x ^= x >> 30;
    x *= 0xBF58476D1CE4E5B9ULL;
    x ^= x >> 27;
    x *= 0x94D049BB133111EBULL;
    x ^= x >> 31;
    return (uint32_t)(x >> 32);
}

/* multiplier applied to the element: 0 or 1/(1-p) */
NVQA_HD float nvqa_dropout_scale(uint64_t seed, uint64_t step, uint32_t site, uint64_t idx, float p,
                                 float inv_keep)
{
    const float u = (float)(nvqa_hash32(seed, step, site, idx) >> 8) * (1.0f / 16777216.0f);
    return u >= p ? inv_keep : 0.0f;
}

#endif /* NVQA_RNG_H */
