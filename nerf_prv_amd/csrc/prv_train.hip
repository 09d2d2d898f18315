// prv_train.hip -- in-process training step of the field (gfx950), replaces the `while testbed.frame()`
// loop that run.py:185-208 drives for --n_steps 2500 (main.cpp:1668).  The algorithm is the one
// oracle/prv_train.c states (published instant-ngp optimiser, parity unpinned); this file is its HIP form.
//
//   train_rays_kernel      one WAVE = one training ray (four per block): counter RNG -> image / pixel / jitter / background, target
//                          colour, dataset camera (lens solve), unit-cube slab test; the lanes test the occupancy of the ray's steps
//                          (PRV_STEP_FIXED_S: <= 128 jittered uniform samples, two per lane; PRV_STEP_NGP: the engine's marcher,
//                          dt = sqrt(3)/1024 from a per-ray random start, <= 1024 steps, sixteen per lane), ballots give the live
//                          masks, one returning atomic per block reserves the list range, every lane appends its own live samples.
//                          From the second step of a call on the batch of step n + 1 is listed by extra blocks of step n's table-Adam
//                          launch.  train_rays_patch_kernel: patches of adjacent pixels, listed depth step by depth step (an option)
//   train_forward_fast_kernel  a lane pair per sample, 32 samples per wave and round: encode from the canonical table, both MLPs as 24
//                          v_mfma_f32_32x32x16_f16 on prepacked A fragments (the render kernel's machinery); the B fragments the lanes
//                          hold between the layers ARE the activations the backward pass needs and are kept (528 B per sample)
//   train_composite_kernel one WAVE = one ray, 64 samples per chunk: transmittance by a prefix product across the lanes, colour by
//                          wave sums, loss, then the per-sample gradient seeds chunk by chunk from the back (suffix sums by a
//                          reverse scan; a chunk's values are recomputed from the logits)
//   train_tile_kernel<F, false, 2>  the backward pass, one 256-thread block = 32 samples at a time, persistent: kept activations
//                          -> [row][sample] LDS arrays (live tiles only: a byte per tile, read 64 candidates at a time), the dX
//                          chain register-resident on TWO waves (bf16-split operands, v_mfma_f32_32x32x16_bf16: a backward layer's
//                          accumulator is the next layer's B operand; wave w owns row tile w = k-steps 2w, 2w + 1 of the layer
//                          behind, partial sums cross through the rows of G they end in), dW as bf16-split MFMAs accumulated in
//                          registers over all tiles of the block (three weight tiles per wave), then the MERGING TABLE SCATTER:
//                          thread (level, corner, feature) walks the tile's 32 samples with one (entry, sum) pair (four for patch
//                          batches) and issues one f32 atomic per run (under PRV_STEP_FIXED_S the memory side's atomic-request rate
//                          bounds the step; under the engine's marcher the tile work does).  The launch's last blocks also sum the
//                          step's loss / used-sample slices.  Two blocks fill a CU's LDS (80 KB less 960 bytes each).  MODE 0 / 1 (recomputed
//                          forward, LDS chain on v_mfma_f32_32x32x2_f32) are the fallbacks for tiles beyond the activation buffer and
//                          the dev switches PRV_TRAIN_FAST_FWD / PRV_TRAIN_REG_CHAIN
//   adam_table_kernel      sparse Adam on the table, one {w[4], m[4], v[4]} record per group of four scalars; extra blocks do the
//                          first stage of the MLP's dW reduction and list the next step's ray batch
//   adam_mlp_kernel        dense Adam (+L2) on the MLP, fp16 working copies; every thread writes its weight into the forward and
//                          backward MFMA fragments; thread 0 closes the step (budget rule) and opens the next
//   density_refresh_fast_kernel  density at every occupancy cell centre (8 MFMAs per 32 cells) -> EMA -> bitfield
// Five kernels per step behind the ray batch, one captured HIP graph per step variant (prv_train_api.inc).
#include <hip/hip_runtime.h>
#include "prv_train.hpp"
#include <type_traits>

#ifndef PRV_TRAIN_SCATTER_WAYS
#define PRV_TRAIN_SCATTER_WAYS 4 // (entry, sum) pairs a scattering thread keeps while it walks a PATCH batch's tile (TrainTileParams::scatter_ways > 1; one pair otherwise)
#endif
#ifndef PRV_TRAIN_ABLATE
#define PRV_TRAIN_ABLATE 0 // dev only: 1 no table scatter, 2 no dW MFMAs, 4 item-parallel scatter (no run merging), 8 no dX chain, 64 no backward tiles at all, 128 dW on the f32 matrix-core form (K = 2), 16 phase time stamps of block 0 (48: summed over its tiles)
#endif
#if PRV_TRAIN_ABLATE & 16
#if PRV_TRAIN_ABLATE & 32 // phase i's time SUMMED over all tiles of block 0 (and over launches): the average under load
#define STAMP(i) do { const unsigned long long now_ = __builtin_amdgcn_s_memtime(); if (blockIdx.x == 0 && threadIdx.x == 0 && P.stamps) P.stamps[(FWD ? 0 : 32) + (i)] += now_ - stamp_prev; stamp_prev = now_; } while (0)
#else
#define STAMP(i) do { if (blockIdx.x == 0 && threadIdx.x == 0 && P.stamps) P.stamps[(FWD ? 0 : 32) + (i)] = __builtin_amdgcn_s_memtime(); } while (0)
#endif
#else
#define STAMP(i) do { } while (0)
#endif

namespace prv {

namespace {

__device__ __forceinline__ uint64_t t_mix64(uint64_t z) {
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
  return z ^ (z >> 31);
}
__device__ __forceinline__ uint32_t rng_u24(uint64_t seed, uint64_t stream, uint64_t i) {
  return (uint32_t)(t_mix64(seed + (stream + 1) * 0xD1B54A32D192ED03ull + i * 0x9E3779B97F4A7C15ull) >> 40);
}

__device__ __forceinline__ float srgb_to_linear(float c) {
  return c <= 0.04045f ? c / 12.92f : powf((c + 0.055f) / 1.055f, 2.4f);
}

__device__ __forceinline__ bool occ_bit(const uint32_t* __restrict__ occ, int R, float px, float py, float pz) {
  const float fr = (float)R;
  int cx = min((int)(clamp01(px) * fr), R - 1), cy = min((int)(clamp01(py) * fr), R - 1),
      cz = min((int)(clamp01(pz) * fr), R - 1);
  const uint32_t bit = (uint32_t)cx + __umul24((uint32_t)R, (uint32_t)cy + __umul24((uint32_t)R, (uint32_t)cz)); // R <= 1024
  return (occ[bit >> 5] >> (bit & 31u)) & 1u;
}

// ------------------------------------------------------------------ rays of one step

// the sample budget: the ray count of the step after one that cast n_active rays and composited `used` samples (integer
// rule, mirrored by oracle/prv_train.c: orc_train_step)
__device__ __forceinline__ uint32_t next_active_rays(int target_samples, uint32_t n_active, unsigned long long used_raw, int n_rays) {
  if (target_samples <= 0) return n_active;
  const unsigned long long used = used_raw ? used_raw : 1ull, act = n_active;
  unsigned long long a = (unsigned long long)target_samples * act / used;
  a = max(a, act / 2ull);
  a = min(a, act * 2ull);
  a = max(a, 1ull);
  a = min(a, (unsigned long long)n_rays);
  return (uint32_t)a;
}
// which step a ray launch prepares, with how many rays, into which counter (TrainRaysParams::next).  false: block bx lies
// beyond that step's rays whatever the budget rule says (it at most doubles the count) -- decided without touching the
// used-sample slices: the launch has n_rays / 4 blocks whatever the step needs, and at upstream's batch under the engine's
// marcher a step casts ~7 K rays of the 2^16 the grid is sized for (the slices' serial walk by every one of the 16 K blocks
// was the tail of the table's Adam launch).  The slices: one per lane, summed across the wave (integers: any order).
__device__ __forceinline__ bool rays_step(const TrainRaysParams& P, uint32_t bx, uint32_t& step, uint32_t& n_active, uint32_t*& counter) {
  step = P.state->step;
  n_active = P.state->n_active;
  if (P.next) {
    if (P.target_samples > 0 && (unsigned long long)bx * 4ull >= 2ull * (unsigned long long)n_active) return false;
    const int n_slices = (P.n_rays + 1023) / 1024, lane = (int)(threadIdx.x & 63u);
    unsigned long long used = 0ull;
    for (int s = lane; s < n_slices; s += 64) used += (unsigned long long)__double_as_longlong(P.loss_part[2 * s + 1]);
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) used += __shfl_xor(used, d);
    n_active = next_active_rays(P.target_samples, n_active, used, P.n_rays);
    step += 1u;
  }
  counter = P.sample_count + (step & 1u);
  return (unsigned long long)bx * 4ull < (unsigned long long)n_active;
}

constexpr float kTrainNgpDt = 1.7320508075688772f / 1024.0f; // = sqrtf(3.0f) / 1024.0f, the oracle's value bit for bit (as prv_kernels.hip)
constexpr int kOrderWord = 144; // scal word of the deterministic append tickets (two, by step parity): sample_count + kOrderWord

// a block's range of the sample list: ONE returning atomic on the step's counter (thread 0 calls).  deterministic: the blocks
// append in index order -- block `ticket` waits until the blocks before it have appended (workgroups are dispatched in index
// order, so the ones it waits for are running or done) -- and the list, its tiles and every sum over them are reproducible.
// A range that does not fit the list sets the sticky overflow flag: every later kernel then sees an empty batch
// (batch_samples below) and prv_train_steps reports it.
__device__ __forceinline__ uint32_t reserve_samples(const TrainRaysParams& P, uint32_t* counter, uint32_t step, uint32_t ticket, uint32_t tot) {
  uint32_t base;
  if (P.deterministic) {
    uint32_t* turn = P.sample_count + kOrderWord + (step & 1u);
    while (__hip_atomic_load(turn, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) != ticket) __builtin_amdgcn_s_sleep(1);
    base = *(volatile uint32_t*)counter;
    *(volatile uint32_t*)counter = base + tot;
    __hip_atomic_store(turn, ticket + 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
  } else {
    base = tot ? atomicAdd(counter, tot) : 0u;
  }
  if (tot && (unsigned long long)base + tot > (unsigned long long)P.sample_cap) {
    P.state->overflow = 1u;
    base = 0xffffffffu;
  }
  return base;
}
// the samples of the step in flight as every kernel behind the ray batch sees them
__device__ __forceinline__ uint32_t batch_samples(const uint32_t* sample_count, const TrainState* state) {
  return state->overflow ? 0u : sample_count[state->step & 1u];
}

// one WAVE = one ray: the lanes test the occupancy of the ray's steps (64 per round: two rounds for S <= 128 uniform
// samples, up to sixteen for the engine's 1024 steps), ballots give the live masks (kept in LDS), each lane appends its
// own live samples at (offset + rank)
template <bool NGP>
__device__ __forceinline__ void train_rays_block(const TrainRaysParams& P, uint32_t bx) {
  const int lane = threadIdx.x & 63;
  const uint32_t j = bx * 4u + (threadIdx.x >> 6);
  uint32_t step, n_active_step;
  uint32_t* counter;
  if (!rays_step(P, bx, step, n_active_step, counter)) return; // whole block beyond this step's ray budget (block-uniform)
  const bool in_budget = j < n_active_step;
  const uint64_t st = (uint64_t)step * 8u;
  const uint32_t img = (uint32_t)(((uint64_t)rng_u24(P.seed, st + 0, j) * (uint64_t)P.n_img) >> 24);
  const uint32_t px = (uint32_t)(((uint64_t)rng_u24(P.seed, st + 1, j) * (uint64_t)P.W) >> 24);
  const uint32_t py = (uint32_t)(((uint64_t)rng_u24(P.seed, st + 2, j) * (uint64_t)P.H) >> 24);
  const float jitter = (float)rng_u24(P.seed, st + 3, j) * (1.0f / 16777216.0f);
  TrainRay r;
  const CamDev cam = P.cams[img];
  raygen(cam, (int)px, (int)py, 0.5f, 0.5f, r.o, r.d);
  float t0, t1;
  r.t0 = 0.f;
  r.dt = 0.f;
  constexpr int kWords = NGP ? kMaxTrainSteps / 64 : 2;
  __shared__ unsigned long long mk[4][kWords];
  __shared__ uint32_t cnt[4], base;
  const int wv = threadIdx.x >> 6;
  uint32_t n_live = 0u;
  int n_words = 0; // wave-uniform: rounds of 64 steps that can hold a live sample
  if (in_budget && ray_aabb(r.o, r.d, t0, t1)) {
    r.t0 = t0;
    r.dt = NGP ? kTrainNgpDt : (t1 - t0) / (float)P.S;
    if (NGP) { // steps beyond the exit are dead: (t1 - t0) / dt + 2 bounds them (the exact `t < t1` test below decides)
      const int inside = (int)fminf((t1 - t0) * (1024.0f / 1.7320508075688772f) + 2.0f, (float)P.S);
      n_words = (min(inside, P.S) + 63) >> 6;
    } else {
      n_words = (P.S + 63) >> 6;
    }
    for (int q = 0; q < n_words; q++) {
      const int i = q * 64 + lane;
      const float t = fmaf((float)i + jitter, r.dt, t0);
      const bool on = i < P.S && (!NGP || t < t1) && occ_bit(P.occ, P.occ_res, fmaf(t, r.d[0], r.o[0]), fmaf(t, r.d[1], r.o[1]), fmaf(t, r.d[2], r.o[2]));
      const unsigned long long m = __ballot(on);
      if (lane == 0) mk[wv][q] = m;
      n_live += (uint32_t)__popcll(m);
    }
  }
  // ONE returning atomic per block (its four rays' counts summed through LDS), not one per ray: 2^16 rays appending
  // to a single counter were a third of this kernel
  if (lane == 0) cnt[wv] = n_live;
  __syncthreads();
  if (threadIdx.x == 0) base = reserve_samples(P, counter, step, bx, cnt[0] + cnt[1] + cnt[2] + cnt[3]);
  __syncthreads();
  const bool fits = base != 0xffffffffu;
  uint32_t offset = fits ? base : 0u;
  for (int w = 0; w < wv; w++) offset += cnt[w];
  if (!fits) n_live = 0u; // the batch overflowed the list: the step is void (TrainState::overflow)
  const unsigned long long below = (1ull << lane) - 1ull;
  uint32_t pre = 0u;
  for (int q = 0; q < n_words && fits; q++) {
    const unsigned long long m = mk[wv][q];
    if ((m >> lane) & 1ull) P.samples[offset + pre + (uint32_t)__popcll(m & below)] = make_uint2(j, (uint32_t)(q * 64 + lane));
    pre += (uint32_t)__popcll(m);
  }
  if (lane == 0 && in_budget) {
    float bg[3] = {0.f, 0.f, 0.f};
    if (P.random_bg)
      for (int k = 0; k < 3; k++) bg[k] = (float)rng_u24(P.seed, st + 4 + k, j) * (1.0f / 16777216.0f);
    const uint8_t* gp = P.images + (((size_t)img * P.H + py) * P.W + px) * 4;
    const float ga = (float)gp[3] * (1.0f / 255.0f);
    for (int k = 0; k < 3; k++) {
      r.target[k] = fmaf(srgb_to_linear((float)gp[k] * (1.0f / 255.0f)), ga, (1.0f - ga) * bg[k]);
      r.bg[k] = bg[k];
    }
    r.jitter = jitter;
    r.n_live = n_live;
    r.offset = offset;
    r.n_used = 0u;
    r.pad[0] = r.pad[1] = 0u;
    P.rays[j] = r;
  }
}
__device__ __forceinline__ void train_rays_block_any(const TrainRaysParams& P, uint32_t bx) {
  if (P.step_mode == PRV_STEP_NGP) train_rays_block<true>(P, bx);
  else train_rays_block<false>(P, bx);
}
__global__ __launch_bounds__(256) void train_rays_kernel(TrainRaysParams P) { train_rays_block_any(P, blockIdx.x); }

// Patch mode (prv_train_opts.patch_w x patch_h = PP > 1): one BLOCK = one patch of PP adjacent pixels of one image, one
// wave per ray as above.  The rays of a patch share image, jitter and (nearly) their depth range, so the samples of ONE
// depth step of the PP rays lie within a few pixel footprints of each other -- less than a cell of the finest level at
// the planner loop's 1280x720.  The patch's live samples are listed depth step by depth step (rays of a step in snake
// order over the patch): the 32 consecutive samples of a backward tile then share their corner entries on every level,
// the hashed ones included, and the tile's scatter merges them into one add per entry (train_tile_kernel) -- the
// memory side's atomic-request rate is what bounds the step (profiles/NOTES.md, round 4).  A ray's samples are no longer
// contiguous in the list: slot_of[ray * S + k] is the list position of its k-th live sample (the compositing kernel's way in).
__global__ __launch_bounds__(1024) void train_rays_patch_kernel(TrainRaysParams P) {
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const uint32_t pw = (uint32_t)P.patch_w, PP = pw * (uint32_t)P.patch_h; // blockDim.x = 64 PP
  const uint32_t q = blockIdx.x, j = q * PP + (uint32_t)wv;
  const uint32_t n_active = P.state->n_active;
  uint32_t* const counter = P.sample_count + (P.state->step & 1u);
  if (q * PP >= n_active) return; // whole patch beyond this step's ray budget (block-uniform)
  const bool in_budget = j < n_active;
  const uint64_t st = (uint64_t)P.state->step * 8u;
  // image, patch origin and jitter are the PATCH's draws (index q), the background is the ray's (index j)
  const uint32_t ry = (uint32_t)wv / pw, rx = (ry & 1u) ? pw - 1u - (uint32_t)wv % pw : (uint32_t)wv % pw;
  const uint32_t img = (uint32_t)(((uint64_t)rng_u24(P.seed, st + 0, q) * (uint64_t)P.n_img) >> 24);
  const uint32_t px = (uint32_t)(((uint64_t)rng_u24(P.seed, st + 1, q) * (uint64_t)(P.W - P.patch_w + 1)) >> 24) + rx;
  const uint32_t py = (uint32_t)(((uint64_t)rng_u24(P.seed, st + 2, q) * (uint64_t)(P.H - P.patch_h + 1)) >> 24) + ry;
  const float jitter = (float)rng_u24(P.seed, st + 3, P.patch_ray_jitter ? j : q) * (1.0f / 16777216.0f);
  TrainRay r;
  const CamDev cam = P.cams[img];
  raygen(cam, (int)px, (int)py, 0.5f, 0.5f, r.o, r.d);
  float t0, t1;
  r.t0 = 0.f;
  r.dt = 0.f;
  unsigned long long m0 = 0ull, m1 = 0ull;
  if (ray_aabb(r.o, r.d, t0, t1)) {
    r.t0 = t0;
    r.dt = (t1 - t0) / (float)P.S;
    bool on[2];
#pragma unroll
    for (int h = 0; h < 2; h++) {
      const int i = h * 64 + lane;
      const float t = fmaf((float)i + jitter, r.dt, t0);
      on[h] = i < P.S && occ_bit(P.occ, P.occ_res, fmaf(t, r.d[0], r.o[0]), fmaf(t, r.d[1], r.o[1]), fmaf(t, r.d[2], r.o[2]));
    }
    m0 = __ballot(on[0]);
    m1 = __ballot(on[1]);
  }
  if (!in_budget) m0 = m1 = 0ull;
  __shared__ unsigned long long mask[16][2];
  __shared__ uint32_t wtot[2], base;
  if (lane == 0) {
    mask[wv][0] = m0;
    mask[wv][1] = m1;
  }
  __syncthreads();
  // thread i < 128 = depth step i of the patch: how many of the PP rays are live there, and where the step's samples start
  const int i = (int)threadIdx.x;
  uint32_t cnt = 0u, excl = 0u;
  if (i < 128) {
    for (uint32_t rr = 0; rr < PP; rr++) cnt += (uint32_t)((mask[rr][i >> 6] >> (i & 63)) & 1ull);
    uint32_t incl = cnt;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
      const uint32_t u = __shfl_up(incl, d);
      if (lane >= d) incl += u;
    }
    excl = incl - cnt;
    if (lane == 63) wtot[wv] = incl;
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    const uint32_t tot = wtot[0] + wtot[1];
    base = reserve_samples(P, counter, P.state->step, q, tot); // one returning atomic per patch
  }
  __syncthreads();
  const bool fits = base != 0xffffffffu;
  if (!fits) m0 = m1 = 0ull; // the batch overflowed the list: the step is void (TrainState::overflow)
  if (i < 128 && cnt && fits) {
    uint32_t pos = base + excl + (wv == 1 ? wtot[0] : 0u);
    const unsigned long long below = (1ull << (i & 63)) - 1ull;
    for (uint32_t rr = 0; rr < PP; rr++) {
      const unsigned long long m = mask[rr][i >> 6];
      if (!((m >> (i & 63)) & 1ull)) continue;
      const uint32_t jr = q * PP + rr;
      const uint32_t k = (uint32_t)__popcll(m & below) + (i >= 64 ? (uint32_t)__popcll(mask[rr][0]) : 0u);
      P.samples[pos] = make_uint2(jr, (uint32_t)i);
      P.slot_of[(size_t)jr * (size_t)P.S + k] = pos;
      pos++;
    }
  }
  if (lane == 0 && in_budget) {
    float bg[3] = {0.f, 0.f, 0.f};
    if (P.random_bg)
      for (int k = 0; k < 3; k++) bg[k] = (float)rng_u24(P.seed, st + 4 + k, j) * (1.0f / 16777216.0f);
    const uint8_t* gp = P.images + (((size_t)img * P.H + py) * P.W + px) * 4;
    const float ga = (float)gp[3] * (1.0f / 255.0f);
    for (int k = 0; k < 3; k++) {
      r.target[k] = fmaf(srgb_to_linear((float)gp[k] * (1.0f / 255.0f)), ga, (1.0f - ga) * bg[k]);
      r.bg[k] = bg[k];
    }
    r.jitter = jitter;
    r.n_live = (uint32_t)__popcll(m0) + (uint32_t)__popcll(m1);
    r.offset = 0u; // unused in patch mode (slot_of)
    r.n_used = 0u;
    r.pad[0] = r.pad[1] = 0u;
    P.rays[j] = r;
  }
}

// ------------------------------------------------------------------ one level of the encoder, canonical table

// the eight corner entries of a level: per-axis terms once (two clamped coordinates per axis, the y and z ones multiplied by
// the hash primes or by the dense strides), then one combine per corner -- the same index, bit for bit, as
// cc0 ^ cc1 * P1 ^ cc2 * P2 masked / cc0 + res * (cc1 + res * cc2), without two multiplies and a branch per corner.  Hashed or
// dense is picked with a bit mask ((a & hm) | (b & ~hm): one v_bitop3), not with v_cndmask: that one costs 10-20 issue cycles
// on this chip (profiles/r04_valu_issue_rate.txt), the three-input bit operation two.
__device__ __forceinline__ uint32_t pick_bits(uint32_t a, uint32_t b, uint32_t hm) { return (a & hm) | (b & ~hm); }
__device__ __forceinline__ void level_corner_indices(const LevelCanon& L, const uint32_t c0[3], uint32_t idx[8]) {
  const uint32_t hm = 0u - L.hashed; // all ones on a hashed level (LevelCanon::hashed is 0 or 1; written so that the compiler does not turn the picks back into selects)
  const uint32_t my = pick_bits(2654435761u, L.res, hm), mz = pick_bits(805459861u, L.res * L.res, hm), mask = L.size - 1u;
  uint32_t x[2], ty[2], tz[2], yz[4];
#pragma unroll
  for (int b = 0; b < 2; b++) {
    x[b] = min(c0[0] + (uint32_t)b, L.res - 1u);
    ty[b] = min(c0[1] + (uint32_t)b, L.res - 1u) * my;
    tz[b] = min(c0[2] + (uint32_t)b, L.res - 1u) * mz;
  }
#pragma unroll
  for (int q = 0; q < 4; q++) yz[q] = pick_bits(ty[q & 1] ^ tz[q >> 1], ty[q & 1] + tz[q >> 1], hm);
#pragma unroll
  for (int c = 0; c < 8; c++) idx[c] = pick_bits((x[c & 1] ^ yz[c >> 1]) & mask, x[c & 1] + yz[c >> 1], hm);
}

// features of one level (binary16 blend, bit for bit encode_level / orc_encode) + the corner entries and
// weights the backward pass scatters into
template <int F>
__device__ __forceinline__ void train_encode_level(const uint16_t* __restrict__ table, const LevelCanon& L, float px,
                                                   float py, float pz, float feat[F], uint32_t cidx[8], float cw[8]) {
  const float p[3] = {clamp01(px), clamp01(py), clamp01(pz)};
  uint32_t c0[3];
  _Float16 wh[3][2];
#pragma unroll
  for (int a = 0; a < 3; a++) {
    const float pos = fmaf(L.scale, p[a], 0.5f);
    const float fl = floorf(pos);
    const float w = pos - fl;
    c0[a] = (uint32_t)(int)fl;
    wh[a][0] = to_half(1.0f - w);
    wh[a][1] = (_Float16)w;
  }
  // the blend: acc_k = fma(w, e_k, acc_k) in fp16, corner after corner -- two features per v_pk_fma_f16 (element by element the
  // same IEEE fma as the scalar instruction; written out because the build switches the SLP vectoriser off)
  typedef _Float16 half2v __attribute__((ext_vector_type(2)));
  half2v acc2[F / 2];
#pragma unroll
  for (int k = 0; k < F / 2; k++) acc2[k] = half2v{(_Float16)0.0f, (_Float16)0.0f};
  uint32_t ix[8];
  level_corner_indices(L, c0, ix);
#pragma unroll
  for (int c = 0; c < 8; c++) {
    const _Float16 wxy = wh[0][c & 1] * wh[1][(c >> 1) & 1];
    const _Float16 w = wxy * wh[2][c >> 2];
    cidx[c] = L.offset + ix[c];
    cw[c] = (float)w;
    typedef _Float16 entry_t __attribute__((ext_vector_type(F)));
    // one 4- or 8-byte load at a 32-bit byte offset from the table's (wave-uniform) base: no 64-bit add per corner (a canonical table is < 4 GiB: prv_train_create)
    const entry_t e = *reinterpret_cast<const entry_t*>(reinterpret_cast<const char*>(table) + (uint32_t)(cidx[c] * (uint32_t)(F * 2)));
    const half2v w2 = {w, w};
#pragma unroll
    for (int k = 0; k < F / 2; k++) acc2[k] = __builtin_elementwise_fma(w2, half2v{e[2 * k], e[2 * k + 1]}, acc2[k]);
  }
#pragma unroll
  for (int k = 0; k < F; k++) feat[k] = (float)acc2[k >> 1][k & 1];
}

// the corner entries and weights alone (no table access): what the backward pass needs of a level when the forward
// pass kept the activations
template <int F>
__device__ __forceinline__ void train_level_corners(const LevelCanon& L, float px, float py, float pz, uint32_t cidx[8], float cw[8]) {
  const float p[3] = {clamp01(px), clamp01(py), clamp01(pz)};
  uint32_t c0[3];
  _Float16 wh[3][2];
#pragma unroll
  for (int a = 0; a < 3; a++) {
    const float pos = fmaf(L.scale, p[a], 0.5f);
    const float fl = floorf(pos);
    const float w = pos - fl;
    c0[a] = (uint32_t)(int)fl;
    wh[a][0] = to_half(1.0f - w);
    wh[a][1] = (_Float16)w;
  }
  uint32_t ix[8];
  level_corner_indices(L, c0, ix);
#pragma unroll
  for (int c = 0; c < 8; c++) {
    const _Float16 wxy = wh[0][c & 1] * wh[1][(c >> 1) & 1];
    cidx[c] = L.offset + ix[c];
    cw[c] = (float)(wxy * wh[2][c >> 2]);
  }
}

// ------------------------------------------------------------------ tile kernel

typedef float f32x16v __attribute__((ext_vector_type(16)));
__device__ __forceinline__ f32x16v mfma32(float a, float b, f32x16v c) {
  return __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c, 0, 0, 0);
}
// v = hi + lo, both bf16 (16 significant bits).  Written on pairs: one v_cvt_pk_bf16_f32 per two values each way, the high parts
// widened again by a shift / a mask of the packed word, the differences as one v_pk_add_f32 -- 20 instructions per eight values.
// The scalar form ((__bf16)v per element) cost 32: this build switches the SLP vectoriser off, and every conversion became an
// instruction of its own.  Same roundings, same bits.
typedef __bf16 bf16x8v __attribute__((ext_vector_type(8)));
__device__ __forceinline__ void split_bf16(const float (&v)[8], bf16x8v& hi, bf16x8v& lo) {
  typedef __bf16 bf16x2v __attribute__((ext_vector_type(2)));
  typedef float f32x2v __attribute__((ext_vector_type(2)));
#pragma unroll
  for (int j = 0; j < 8; j += 2) {
    const f32x2v x = {v[j], v[j + 1]};
    const bf16x2v h = __builtin_convertvector(x, bf16x2v);
    const uint32_t p = __builtin_bit_cast(uint32_t, h);
    const f32x2v hf = {__uint_as_float(p << 16), __uint_as_float(p & 0xffff0000u)};
    const bf16x2v l = __builtin_convertvector(x - hf, bf16x2v);
    hi[j] = h[0];
    hi[j + 1] = h[1];
    lo[j] = l[0];
    lo[j + 1] = l[1];
  }
}
// accumulator register i of lane half h holds row rho(i, h) of the 32x32 tile
__device__ __forceinline__ int rho(int i, int h) { return (i & 3) + 8 * (i >> 2) + 4 * h; }

constexpr int kTS = 33; // sample stride of the [row][sample] LDS arrays (conflict-free in both read directions)
constexpr int kTSA2 = 40, kTSG2 = 36; // the register-chain instance's strides (halfs of an activation row, floats of a gradient row)
// layer l: n_in, n_out, canonical offset, padded LDS row stride and LDS offset of the weights
__device__ constexpr int kLIn[5] = {32, 64, 32, 64, 64}, kLOut[5] = {64, 16, 64, 64, 16};
__device__ constexpr int kLOff[5] = {0, 2048, 3072, 5120, 9216};
__device__ constexpr int kLStr[5] = {65, 17, 65, 65, 17};
__device__ constexpr int kLLds[5] = {0, 2080, 3168, 5248, 9408}; // prefix of n_in * stride
constexpr int kWLds = 10496;
// activation rows: feat 0..31 | h1 32..95 | in2 96..127 | h2 128..191 | h3 192..255
constexpr int kAFeat = 0, kAH1 = 32, kAIn2 = 96, kAH2 = 128, kAH3 = 192, kARows = 256;
// Saved activations (TrainTileParams::act): slot q of lane half h of a forward lane holds 8 halfs; element j is row
// act_row(q, h, j) of the [row][sample] activation array.  Slots: 0,1 the lane's features (levels h, 2+h, 4+h, ...:
// the lane pair splits the levels) | 2..5 h1 | 6 density output | 7 SH | 8..11 h2 | 12..15 h3
// Every slot is {first row, rows per lane half, how j walks the rows}: row = base + per_half * h + act_step(walk, j)
struct ActSlot {
  int base, per_half, walk; // walk 0: 8 (j >> 2) + (j & 3), the accumulator order of a 32x32 tile | 1: j | 2: 4 (j >> 1) + (j & 1) (F = 2 features)
};
template <int F>
__device__ __forceinline__ ActSlot act_slot(int q) {
  if (q < 2) return ActSlot{kAFeat + 16 * q, F, F == 4 ? 0 : F == 8 ? 1 : 2};
  if (q < 6) return ActSlot{kAH1 + 16 * (q - 2), 4, 0};
  if (q == 6) return ActSlot{kAIn2, 4, 0};
  if (q == 7) return ActSlot{kAIn2 + 16, 8, 1};
  return ActSlot{(q < 12 ? kAH2 : kAH3) + 16 * (q & 3), 4, 0};
}
__device__ __forceinline__ constexpr int act_step(int walk, int j) { return walk == 0 ? 8 * (j >> 2) + (j & 3) : walk == 1 ? j : 4 * (j >> 1) + (j & 1); }
template <int F>
__device__ __forceinline__ int act_row(int q, int h, int j) {
  const ActSlot sl = act_slot<F>(q);
  return sl.base + sl.per_half * h + act_step(sl.walk, j);
}
// ... followed by the 32 sample positions of the tile (float4 each: the backward pass computes the corner entries and
// weights from them without walking sample -> ray first)
constexpr int kActSlots = 16, kActPosWord = kActSlots * 64, kActTileWords = kActPosWord + 32; // uint4 per 32-sample tile
static_assert(kActTileBytes == (size_t)kActTileWords * 16, "prv_train.hpp sizes the buffer");

// gradient rows: dOrr 0..15 | dH3 16..79 | dH2 80..143 | dOd 144..175 | dH1 176..239 | dFeat 240..271
constexpr int kGOrr = 0, kGH3 = 16, kGH2 = 80, kGOd = 144, kGH1 = 176, kGFeat = 240, kGRows = 272;

// OUT[32*mt + row][s] = sum_k A(row, k) * IN[k][s]  over k in [0, K): forward (A = W[k][o]) or transposed
// (A = W[row][k]) -- one wave, one 32-row tile, K/2 MFMAs
template <bool TRANSPOSED, typename TIN>
__device__ __forceinline__ f32x16v layer_tile(const _Float16* __restrict__ Wl, int stride, int n_rows_valid, int mt,
                                              const TIN* __restrict__ in, int K, int lane) {
  const int r = lane & 31, h = lane >> 5;
  const int row = 32 * mt + r;
  const bool valid = row < n_rows_valid;
  f32x16v acc = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
  // K is 16, 32 or 64 at every call site: groups of 4 MFMAs whose 8 LDS operands are loaded one group ahead,
  // so the matrix pipe never waits for an LDS round trip
  float a[4], b[4], an[4], bn[4];
  auto load = [&](int k0, float* pa, float* pb) {
#pragma unroll
    for (int j = 0; j < 4; j++) {
      const int k = k0 + 2 * j;
      pa[j] = valid ? (float)(TRANSPOSED ? Wl[row * stride + k] : Wl[k * stride + row]) : 0.0f;
      pb[j] = (float)in[k * kTS + r];
    }
  };
  load(h, an, bn);
  for (int k0 = h; k0 < K; k0 += 8) {
#pragma unroll
    for (int j = 0; j < 4; j++) {
      a[j] = an[j];
      b[j] = bn[j];
    }
    if (k0 + 8 < K) load(k0 + 8, an, bn);
#pragma unroll
    for (int j = 0; j < 4; j++) acc = mfma32(a[j], b[j], acc);
  }
  return acc;
}

// one add into the table gradient: an f32 atomic at the memory side, or -- deterministic (tests) -- a 64-bit integer atomic
// on the value in fixed point (2^-kGradQBits; integer adds commute, so the sum does not depend on the order the tiles arrive in)
__device__ __forceinline__ void table_grad_add(const TrainTileParams& P, size_t idx, float v) {
  if (P.table_grad_q) {
    const long long q = __float2ll_rn(v * (float)(1ull << kGradQBits)); // a power-of-two scale: exact until the conversion rounds
    atomicAdd(reinterpret_cast<unsigned long long*>(P.table_grad_q) + idx, (unsigned long long)q);
  } else {
    atomicAdd(P.table_grad + idx, v);
  }
}

template <int F, bool FWD, int MODE = 0> // MODE 0: recompute the forward pass | 1: kept activations, LDS chain | 2: kept activations, register chain
__global__ __launch_bounds__(256)
__attribute__((amdgpu_waves_per_eu(2))) // two blocks per CU (74 KB of LDS each; 80 KB for the register-chain instance): at most 256 registers per lane
void train_tile_kernel(TrainTileParams P) {
  constexpr bool SAVED = MODE != 0;
  static_assert(!(FWD && SAVED), "kept activations are a backward-pass input");
  // sample strides of the [row][sample] arrays.  The register-chain instance reads the operands of its dW tiles as
  // 16-byte runs of 8 (activations) / 2 x 4 (gradients) consecutive samples of a row: rows 16-byte aligned, and a stride
  // of 4 (mod 8) dwords spreads eight consecutive rows over all 32 banks
  constexpr int TSA = MODE == 2 ? kTSA2 : kTS, TSG = MODE == 2 ? kTSG2 : kTS;
  // weights and activations are fp16 VALUES (working weights, rounded activations): stored as fp16, widened
  // at the operand read; 21 + 17 KB (+ 36 KB of f32 gradients backward; 21 + 20 + 39 KB with the register-chain instance's strides) -> 2 backward / 4 forward blocks per CU
  // 16-byte aligned BY DECLARATION: as `float lds[]` the array sits wherever the static __shared__ data ends, and one 4-byte static word
  // in front of it made every 16- and 8-byte LDS access of this kernel a misaligned one -- 0.335 -> 0.48 ms per step, same results (r06aj, r06au)
  extern __shared__ __attribute__((aligned(16))) float lds[];
#if (PRV_TRAIN_ABLATE & 48) == 48
  unsigned long long stamp_prev = __builtin_amdgcn_s_memtime();
#endif
  float* G = lds;                                                  // kGRows * TSG floats (backward only)
  _Float16* W = reinterpret_cast<_Float16*>(lds + (FWD ? 0 : kGRows * TSG)); // kWLds halfs: [in][out] weights; MODE 2: the 20 backward A fragments
  _Float16* A = W + kWLds;                                         // kARows * TSA halfs
  static_assert(kBwdFrags * kFragHalfs <= kWLds, "the backward fragments take the weight array's place");
  STAMP(0);
  // (the wave index as a scalar: what depends on it -- act_row()'s slot classes, the chain's and dW's roles -- branches on SCC, not on exec masks)
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int r = lane & 31, h = lane >> 5;
  if constexpr (!FWD) {
    // The step's loss and used-sample count ride on this launch: its LAST blocks each sum one 1024-ray slice of the per-ray
    // terms the compositing kernel left (a kernel boundary ago: plain loads) -- thread t the rays t, t + 256, ... of the
    // slice, then a tree over the 256 partial sums: a fixed order -- and the step's last kernel adds the slices in slice
    // order (end_step).  Until round 5 a kernel of its own in every member's chain (6 us alone, 13 us beside the other
    // members' launches); a ticket in the compositing kernel instead costs a device-scope release fence per block: slower.
    // (a block takes every gridDim.x-th slice: n_rays up to 2^22 is 4096 slices, more than any backward grid)
    const int n_slices = P.ray_loss && P.tile_begin == 0 ? (P.n_rays + 1023) / 1024 : 0;
    for (int slice = (int)gridDim.x - 1 - (int)blockIdx.x; slice < n_slices; slice += (int)gridDim.x) {
      double* sl = reinterpret_cast<double*>(lds);
      unsigned long long* su = reinterpret_cast<unsigned long long*>(lds) + 256;
      double a = 0.0;
      unsigned long long u = 0ull;
#pragma unroll
      for (int k = 0; k < 4; k++) {
        const int i = slice * 1024 + k * 256 + (int)threadIdx.x;
        if (i < P.n_rays) {
          a += (double)P.ray_loss[i];
          u += P.ray_used[i];
        }
      }
      sl[threadIdx.x] = a;
      su[threadIdx.x] = u;
      __syncthreads();
      for (int w = 128; w >= 1; w >>= 1) {
        if ((int)threadIdx.x < w) {
          sl[threadIdx.x] += sl[threadIdx.x + w];
          su[threadIdx.x] += su[threadIdx.x + w];
        }
        __syncthreads();
      }
      if (threadIdx.x == 0) {
        P.loss_part[2 * slice] = sl[0];
        P.loss_part[2 * slice + 1] = __longlong_as_double((long long)su[0]);
      }
      __syncthreads(); // the scratch goes back to the tile loop
    }
  }
  const uint32_t n_samples = batch_samples(P.sample_count, P.state);
  const uint32_t n_tiles = min((n_samples + 31u) / 32u, P.tile_limit);
  if (P.tile_begin + blockIdx.x >= n_tiles) { // nothing to do for this block: its slot of the weight-gradient partials is zero
    if (!FWD)
      for (int i = tid; i < PRV_MLP_HALFS; i += 256) P.mlp_grad_partial[(size_t)(P.slot_base + (int)blockIdx.x) * PRV_MLP_HALFS + i] = 0.0f;
    return;
  }
  // level constants: kernel arguments indexed by a per-lane level would be re-fetched from the kernarg
  // segment through the vector memory path before every corner (measured: 37 us per tile) -> LDS copy
  __shared__ LevelCanon lv[16];
  if (tid < 16 * (int)(sizeof(LevelCanon) / 4))
    reinterpret_cast<uint32_t*>(lv)[tid] = reinterpret_cast<const uint32_t*>(P.levels)[tid];
  if constexpr (MODE == 2) { // fp16 fragments as prepack_frags_kernel left them: 20 KB, 16 bytes per thread and step
    uint4* dst = reinterpret_cast<uint4*>(W);
    for (int i = tid; i < kBwdFrags * 64; i += 256) dst[i] = P.bwd_frags[i];
  }
#pragma unroll
  for (int l = 0; l < (MODE == 2 ? 0 : 5); l++) { // n_out is a power of two per layer: shifts, no divisions; loads independent
    constexpr int kSh[5] = {6, 4, 6, 6, 4};
    const int n = kLIn[l] * kLOut[l];
#pragma unroll 4
    for (int i = tid; i < n; i += 256)
      W[kLLds[l] + (i >> kSh[l]) * kLStr[l] + (i & (kLOut[l] - 1))] = (_Float16)P.mlp[kLOff[l] + i];
  }
  __syncthreads(); STAMP(1);

  constexpr int LPT = (32 / F) / 8; // levels per thread (8 threads per sample)
  constexpr int kDw = MODE == 2 ? 6 : 3; // weight-gradient tiles a wave accumulates over all tiles of the block (MODE 2: waves 2 and 3 hold all twelve)
  f32x16v dw[kDw];
  for (int q = 0; q < kDw; q++) dw[q] = f32x16v{0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};

  // what a kept-activation tile reads from memory: this wave's four activation slots, the sample's position and seed
  struct TileIn {
    uint4 av[kActSlots / 4];
    float pos[3];
    float4 seed;
  };
  auto fetch_tile = [&](uint32_t tile, int lane, TileIn& in) {
    const uint4* src = P.act + (size_t)tile * kActTileWords;
    typedef unsigned int u32x4v __attribute__((ext_vector_type(4)));
#pragma unroll
    for (int i = 0; i < kActSlots / 4; i++) // (non-temporal, as the forward pass stored them: read once)
      in.av[i] = __builtin_bit_cast(uint4, __builtin_nontemporal_load(reinterpret_cast<const u32x4v*>(src + (wave + 4 * i) * 64 + lane)));
    const int s = lane & 31;
    const uint32_t sid = tile * 32u + (uint32_t)s;
    const float4 p = __builtin_bit_cast(float4, src[kActPosWord + s]);
    in.pos[0] = p.x;
    in.pos[1] = p.y;
    in.pos[2] = p.z;
    in.seed = make_float4(0.f, 0.f, 0.f, 0.f);
    if (sid < n_samples) in.seed = P.seeds[sid];
  };
  const int lane_outer = lane;
  // Backward: only LIVE tiles are walked -- tile_live[t] != 0: the compositing kernel saw a used sample there.  Samples behind
  // their ray's termination are listed and evaluated (the forward pass cannot know) but carry no gradient, and they form runs
  // (the list is ray by ray in depth order): under the engine's marcher a learnt opaque surface has hundreds of live steps
  // behind it -- a third of the list in the planner loop's steady state, nine tenths in its one-view first round.  A block
  // clears the bytes of the tiles it has read (it is their only reader), so the next step starts from zeros.
  // The block's candidates are the tiles first + c * gridDim.x; their bytes are read 64 candidates at a time (lane j the byte of
  // candidate win_c0 + j, one ballot): a byte per step of the walk was a dependent global load -- a microsecond -- in front of
  // every tile.  (A window's bits are consumed forwards only, so the clearing of the bytes behind the walk does not matter.)
  const uint32_t first = P.tile_begin + blockIdx.x;
  uint32_t win_c0 = 0u;
  unsigned long long win = 0ull;
  auto window = [&](uint32_t c0) {
    const uint32_t t = first + (c0 + (uint32_t)lane) * gridDim.x;
    return __ballot(t < n_tiles && (!P.tile_live || P.tile_live[t] != 0));
  };
  if constexpr (!FWD) win = window(0u);
  uint32_t cand = 0u; // candidate index of `tile`
  auto next_live = [&](uint32_t c) { // the first live candidate >= c (its tile may lie beyond n_tiles: the walk's end)
    if constexpr (!FWD) {
      for (;;) {
        if (first + c * gridDim.x >= n_tiles) break;
        if (c >= win_c0 + 64u) {
          win_c0 = c;
          win = window(c);
        }
        const unsigned long long m = win >> (c - win_c0);
        if (m != 0ull) {
          c += (uint32_t)__builtin_ctzll(m);
          break;
        }
        c = win_c0 + 64u;
      }
    }
    return c;
  };
  TileIn pre; // MODE 2: the NEXT live tile's kept activations, fetched while this one is worked on
  cand = next_live(0u);
  uint32_t tile = first + cand * gridDim.x;
  if ((PRV_TRAIN_ABLATE & 64) && !FWD) tile = n_tiles;
  if constexpr (MODE == 2)
    if (tile < n_tiles) fetch_tile(tile, lane_outer, pre);
  for (; tile < n_tiles;) {
    // The lane id is made opaque once per tile: everything derived from it inside the loop (some 70 LDS addresses of the
    // [row][sample] arrays) is then recomputed per tile instead of being hoisted out of the loop and kept in registers for
    // the whole kernel, which is what pushed the register-chain instance past 256 of them.
    int lane = lane_outer;
    if constexpr (MODE == 2) asm volatile("" : "+v"(lane)); // (the LDS-chain instances are faster with the hoisted form: they have the registers)
    const int tid = wave * 64 + lane, r = lane & 31, h = lane >> 5;
    const uint32_t cand_next = next_live(cand + 1u);
    const uint32_t tile_next = first + cand_next * gridDim.x;
    // ---- phase E: encode (8 threads per sample), SH inputs, gradient seeds
    const int s = tid & 31, g = tid >> 5;
    const uint32_t sid = tile * 32u + (uint32_t)s;
    const bool live = sid < n_samples;
    uint32_t cidx[LPT][8];
    float cw[LPT][8];
    float spos[3] = {0.5f, 0.5f, 0.5f}; // kept-activation tiles: the sample's position (TileIn::pos)
    float4 seed = make_float4(0.f, 0.f, 0.f, 0.f);
    // SAVED: backward pass on tiles whose activations the forward pass kept (launch_train_tiles gives this instance
    // exactly those): no table gather, no forward layers -- the corner entries and weights (arithmetic only), the seeds,
    // and 16 KB of activations copied into the [row][sample] array
    if constexpr (SAVED) {
      // the tile's 16 KB of kept activations, its 32 sample positions and its gradient seeds: independent loads
      TileIn in;
      if constexpr (MODE == 2) {
        in = pre;
        // (opaque here: otherwise the unpacking of these registers is scheduled right behind the loads of the PREVIOUS
        // iteration -- pure arithmetic on their results -- and the wave waits for the prefetch the moment it is issued)
#pragma unroll
        for (int i = 0; i < kActSlots / 4; i++) asm volatile("" : "+v"(in.av[i].x), "+v"(in.av[i].y), "+v"(in.av[i].z), "+v"(in.av[i].w));
        asm volatile("" : "+v"(in.pos[0]), "+v"(in.pos[1]), "+v"(in.pos[2]));
        asm volatile("" : "+v"(in.seed.x), "+v"(in.seed.y), "+v"(in.seed.z), "+v"(in.seed.w));
      } else {
        fetch_tile(tile, lane, in);
      }
      seed = in.seed;
      const uint4* av = in.av;
      // (the corner entries and weights are computed where they are staged for the scatter, behind the chain: three registers of position
      // across the chain and dW instead of sixteen per level)
      spos[0] = in.pos[0];
      spos[1] = in.pos[1];
      spos[2] = in.pos[2];
      if (g == 1) {
        G[(kGOrr + 0) * TSG + s] = seed.y;
        G[(kGOrr + 1) * TSG + s] = seed.z;
        G[(kGOrr + 2) * TSG + s] = seed.w;
#pragma unroll
        for (int k = 3; k < 16; k++) G[(kGOrr + k) * TSG + s] = 0.0f;
      }
#pragma unroll
      for (int i = 0; i < kActSlots / 4; i++) {
        const int q = wave + 4 * i; // slot of this wave's 64 words: lane = (half, sample)
        const uint4 v = av[i];
        const uint32_t vw[4] = {v.x, v.y, v.z, v.w}; // (halves by shifts, not by a pointer cast: the cast makes the compiler carry the
        // prefetched registers from iteration to iteration as sixteen-bit pieces, cut up the moment the load returns)
        // the slot's rows: one base address per slot, the element's row a constant offset -- the way j walks the rows is the same
        // for every slot but the SH one (and the feature slots of F != 4): ONE scalar branch per slot where it can differ.  (With
        // act_row() per element the row was picked by a ladder of scalar branches per 16-bit store: 700 scalar instructions per tile.)
        const ActSlot sl = act_slot<F>(q);
        _Float16* dst = A + (sl.base + sl.per_half * h) * TSA + r;
        auto put = [&](auto walk) {
#pragma unroll
          for (int j = 0; j < 8; j++) dst[act_step(decltype(walk)::value, j) * TSA] = __builtin_bit_cast(_Float16, (uint16_t)(vw[j >> 1] >> (16 * (j & 1))));
        };
        if (i >= 2 || (i == 0 && F == 4)) put(std::integral_constant<int, 0>{}); // hidden layers (and F = 4 features): the accumulator order
        else if (sl.walk == 0) put(std::integral_constant<int, 0>{});
        else if (sl.walk == 1) put(std::integral_constant<int, 1>{});
        else put(std::integral_constant<int, 2>{});
      }
      __syncthreads(); STAMP(2);
      // the next live tile's activations, positions and seeds: in flight while this tile's chain, dW and scatter run
      if constexpr (MODE == 2)
        if (tile_next < n_tiles) fetch_tile(tile_next, lane, pre);
    } else {
    {
      float pos[3] = {0.5f, 0.5f, 0.5f}, dir[3] = {0.f, 0.f, 1.f};
      if (live) {
        const uint2 sr = P.samples[sid];
        const TrainRay* ray = P.rays + sr.x;
        const float t = fmaf((float)sr.y + ray->jitter, ray->dt, ray->t0);
        for (int a = 0; a < 3; a++) {
          dir[a] = ray->d[a];
          pos[a] = fmaf(t, dir[a], ray->o[a]);
        }
        if (!FWD) seed = P.seeds[sid];
      }
#pragma unroll
      for (int q = 0; q < LPT; q++) {
        const int l = g * LPT + q;
        float f[F];
        const LevelCanon L = lv[l];
        train_encode_level<F>(P.table, L, pos[0], pos[1], pos[2], f, cidx[q], cw[q]);
#pragma unroll
        for (int k = 0; k < F; k++) A[(kAFeat + l * F + k) * kTS + s] = (_Float16)(live ? f[k] : 0.0f);
      }
      if (g == 0) {
        float sh[16];
        sh4(dir[0], dir[1], dir[2], sh);
#pragma unroll
        for (int k = 0; k < 16; k++) A[(kAIn2 + 16 + k) * kTS + s] = live ? to_half(sh[k]) : (_Float16)0.0f;
      }
      if (!FWD && g == 1) {
        G[(kGOrr + 0) * kTS + s] = seed.y;
        G[(kGOrr + 1) * kTS + s] = seed.z;
        G[(kGOrr + 2) * kTS + s] = seed.w;
#pragma unroll
        for (int k = 3; k < 16; k++) G[(kGOrr + k) * kTS + s] = 0.0f;
      }
    }
    __syncthreads(); STAMP(2);
    // ---- forward: relu + fp16 rounding of the hidden activations as in inference
    if (wave < 2) { // D1: 32 -> 64
      const f32x16v a = layer_tile<false, _Float16>(W + kLLds[0], kLStr[0], 64, wave, A + kAFeat * kTS, 32, lane);
#pragma unroll
      for (int i = 0; i < 16; i++) A[(kAH1 + 32 * wave + rho(i, h)) * kTS + r] = (_Float16)fmaxf(a[i], 0.0f);
    }
    __syncthreads(); STAMP(3);
    float od0 = 0.0f;
    if (wave == 0) { // D2: 64 -> 16
      const f32x16v a = layer_tile<false, _Float16>(W + kLLds[1], kLStr[1], 16, 0, A + kAH1 * kTS, 64, lane);
#pragma unroll
      for (int i = 0; i < 16; i++) {
        const int row = rho(i, h);
        if (row < 16) A[(kAIn2 + row) * kTS + r] = (_Float16)a[i];
      }
      od0 = a[0]; // row 0 lives in register 0 of lane half 0
    }
    __syncthreads(); STAMP(4);
    if (wave < 2) { // R1: 32 -> 64
      const f32x16v a = layer_tile<false, _Float16>(W + kLLds[2], kLStr[2], 64, wave, A + kAIn2 * kTS, 32, lane);
#pragma unroll
      for (int i = 0; i < 16; i++) A[(kAH2 + 32 * wave + rho(i, h)) * kTS + r] = (_Float16)fmaxf(a[i], 0.0f);
    }
    __syncthreads(); STAMP(5);
    if (wave < 2) { // R2: 64 -> 64
      const f32x16v a = layer_tile<false, _Float16>(W + kLLds[3], kLStr[3], 64, wave, A + kAH2 * kTS, 64, lane);
#pragma unroll
      for (int i = 0; i < 16; i++) A[(kAH3 + 32 * wave + rho(i, h)) * kTS + r] = (_Float16)fmaxf(a[i], 0.0f);
    }
    __syncthreads(); STAMP(6);
    if (FWD) {
      if (wave == 0) { // R3: 64 -> 16, logits out
        const f32x16v a = layer_tile<false, _Float16>(W + kLLds[4], kLStr[4], 16, 0, A + kAH3 * kTS, 64, lane);
        const uint32_t o = tile * 32u + (uint32_t)r;
        if (h == 0 && o < n_samples) P.logits[o] = make_float4(od0, a[0], a[1], a[2]);
      }
      __syncthreads(); STAMP(7);
      tile = tile_next; // (forward tiles: every tile is walked)
      cand = cand_next;
      continue;
    }
    } // !saved
    // ---- backward: dX chain (straight through the fp16 roundings, ReLU masks from the activations)
    if constexpr (MODE == 2) {
      {
        // Register-resident chain on ONE wave, like the forward pass: a backward layer's accumulator rows are the next
        // layer's B operand as they stand (the prepacked A fragments carry the K order, prepack_frags_kernel), no LDS round
        // trip or barrier between the layers.  bf16 MFMAs on split operands: the f32 gradient g = g_hi + g_lo (two bf16,
        // 16 significant bits), the fp16 weight W = W_hi + W_lo (exact), W g ~ W_hi g_hi + W_lo g_hi + W_hi g_lo in f32
        // accumulators (the dropped term is below 2^-16 of the product).  60 MFMAs of K = 16 where the LDS form issues
        // ~350 of K = 2 behind five barriers.  The masked gradients still go to the [row][sample] array: dW and the
        // scatter below read them there.
        // ... on TWO waves since round 6: wave w owns row tile w of the 64-row layers (its masks, its rows of G), which is k-steps
        // 2w, 2w + 1 of the layer behind it -- a K split.  A wave multiplies its own half of K into BOTH row tiles (R2) or into the
        // one 32-row tile (R1, D1) and hands the other wave the partial sums of the rows that wave owns, through the rows of G
        // the final values go to anyway (the lane that reads a value is the lane position that wrote it: one C layout), behind
        // a barrier: three more barriers per tile, half the splits, masks and matrix instructions per wave.  Waves 2 and 3 only
        // keep the barriers company.  (Sums: (k-steps 0, 1) + (k-steps 2, 3) where one wave's accumulator took them in turn.)
        typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
        const half8* wf = reinterpret_cast<const half8*>(W);
        const f32x16v zero = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
        const int cwv = wave; // 0 / 1: the chain's two waves.  (Waves 2, 3 instead in the block that shares the CU -- HW_ID.wave_id parity -- measured: nothing, r06al)
        const bool chain = (cwv == 0 || cwv == 1) && !(PRV_TRAIN_ABLATE & 8);
        auto mm3 = [&](int f, const bf16x8& bh, const bf16x8& bl, f32x16v c) {
          const half8 w = wf[f * 64 + lane]; // one LDS read; the split into bf16 high + low parts is exact (11 bits into 8 + 8)
          bf16x8 ah, al;
          float wv[8];
#pragma unroll
          for (int j = 0; j < 8; j++) wv[j] = (float)w[j];
          split_bf16(wv, ah, al);
          c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bh, c, 0, 0, 0);
          c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al, bh, c, 0, 0, 0);
          c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bl, c, 0, 0, 0);
          return c;
        };
        // registers [base, base + 8) of an accumulator -> the high and low bf16 parts of one B operand
        auto split8 = [&](const f32x16v& acc, int base, bf16x8& hi, bf16x8& lo) {
          float v[8];
#pragma unroll
          for (int j = 0; j < 8; j++) v[j] = acc[base + j];
          split_bf16(v, hi, lo);
        };
        const int ws = cwv;
        // dW[k][o] += sum_s X[k][s] dOut[o][s] rides in the chain's shadow: waves 2 and 3 (u = 0, 1) had nothing to do between the chain's
        // barriers, and every gradient block is final a barrier before the chain ends -- dH3 (and the seeds) behind the first, dH2 behind
        // the second, dOd and dH1 behind the third.  Six 32x32 weight tiles per wave, K = the 32 samples: bf16 matrix cores, 16 samples
        // per instruction; lane (r, h) supplies 8 consecutive samples of activation row r (one 16-byte LDS read) and of gradient row r
        // (two); split operands as in the chain (the fp16 activation = hi + lo exactly, the f32 gradient = hi + lo to 16 bits).
        const int u = wave - 2;
        const bool dww = wave >= 2 && !(PRV_TRAIN_ABLATE & 2);
        auto dw_x = [&](int xa, int kk, bf16x8& xh, bf16x8& xl) {
          const half8 x = *reinterpret_cast<const half8*>(A + (xa + r) * TSA + 16 * kk + 8 * h);
          float v[8];
#pragma unroll
          for (int j = 0; j < 8; j++) v[j] = (float)x[j];
          split_bf16(v, xh, xl);
        };
        auto dw_mac = [&](f32x16v& acc, const bf16x8& xh, const bf16x8& xl, int ga, int g_rows, int kk) {
          const float4* gp = reinterpret_cast<const float4*>(G + (ga + r) * TSG + 16 * kk + 8 * h);
          const float4 g0 = gp[0], g1 = gp[1];
          const bool valid = r < g_rows;
          const float gz[8] = {valid ? g0.x : 0.0f, valid ? g0.y : 0.0f, valid ? g0.z : 0.0f, valid ? g0.w : 0.0f,
                               valid ? g1.x : 0.0f, valid ? g1.y : 0.0f, valid ? g1.z : 0.0f, valid ? g1.w : 0.0f};
          bf16x8 gh, gl;
          split_bf16(gz, gh, gl);
          acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(xh, gh, acc, 0, 0, 0);
          acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(xl, gh, acc, 0, 0, 0);
          acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(xh, gl, acc, 0, 0, 0);
        };
        const int own = 32 * cwv, other = 32 * (1 - cwv); // first row of this wave's / the other wave's tile of a 64-row layer
        bf16x8 bh[2], bl[2];
        f32x16v d = zero, part = zero; // this wave's rows of the layer in hand | its half-K sum of its own rows, until the other half arrives
        if (chain) {
          { // colour logits' seeds: lane half 0 holds outputs 0..7, of which r, g, b carry a gradient
            f32x16v sd = zero;
            if (h == 0) {
              sd[0] = seed.y;
              sd[1] = seed.z;
              sd[2] = seed.w;
            }
            split8(sd, 0, bh[0], bl[0]);
          }
          d = mm3(cwv, bh[0], bl[0], zero); // dH3 = W_r3 dOrr, this wave's rows
#pragma unroll
          for (int i = 0; i < 16; i++) {
            const int row = own + rho(i, h);
            d[i] = A[(kAH3 + row) * TSA + r] > (_Float16)0.0f ? d[i] : 0.0f;
            G[(kGH3 + row) * TSG + r] = d[i];
          }
          split8(d, 0, bh[0], bl[0]);
          split8(d, 8, bh[1], bl[1]);
          // dH2 = W_r2 dH3: k-steps 2 wave, 2 wave + 1 into both row tiles
          part = zero; // fragment 2 + 4 mt + st: its own row tile mt = wave, the other wave's mt = 1 - wave
          f32x16v give = zero;
#pragma unroll
          for (int j = 0; j < 2; j++) {
            part = mm3(2 + 6 * cwv + j, bh[j], bl[j], part);
            give = mm3(6 - 2 * cwv + j, bh[j], bl[j], give);
          }
#pragma unroll
          for (int i = 0; i < 16; i++) G[(kGH2 + other + rho(i, h)) * TSG + r] = give[i];
        }
        __syncthreads();
        if (dww) { // R2 (rows [32 u, 32 u + 32) of h2 x both halves of dH3) and R3 (the same rows of h3 x the colour seeds)
#pragma unroll
          for (int kk = 0; kk < 2; kk++) {
            bf16x8 xh, xl;
            dw_x(kAH2 + 32 * u, kk, xh, xl);
            dw_mac(dw[0], xh, xl, kGH3, 32, kk);
            dw_mac(dw[1], xh, xl, kGH3 + 32, 32, kk);
            dw_x(kAH3 + 32 * u, kk, xh, xl);
            dw_mac(dw[2], xh, xl, kGOrr, 16, kk);
          }
        }
        if (chain) {
#pragma unroll
          for (int i = 0; i < 16; i++) {
            const int row = own + rho(i, h);
            const float got = G[(kGH2 + row) * TSG + r];
            const float v = part[i] + got; // (k-steps 0, 1) + (k-steps 2, 3): the same sum on either wave
            d[i] = A[(kAH2 + row) * TSA + r] > (_Float16)0.0f ? v : 0.0f;
            G[(kGH2 + row) * TSG + r] = d[i];
          }
          split8(d, 0, bh[0], bl[0]);
          split8(d, 8, bh[1], bl[1]);
          // dOd = (W_r1 dH2)[0..15] (+ the density seed on row 0); the SH rows carry no parameters.  Both waves finish the
          // sum (each needs it as its B operand): wave 1's half goes through rows 0..15 of the dOd block, wave 0's through
          // the block's unused rows 16..31; wave 0 then writes the final rows
          part = zero;
#pragma unroll
          for (int j = 0; j < 2; j++) part = mm3(10 + 2 * cwv + j, bh[j], bl[j], part);
#pragma unroll
          for (int i = 0; i < 8; i++) G[(kGOd + 16 * (1 - cwv) + rho(i, h)) * TSG + r] = part[i]; // registers 0..7 = rows < 16
        }
        __syncthreads();
        if (dww) { // R1: the density outputs + SH inputs x half u of dH2
#pragma unroll
          for (int kk = 0; kk < 2; kk++) {
            bf16x8 xh, xl;
            dw_x(kAIn2, kk, xh, xl);
            dw_mac(dw[3], xh, xl, kGH2 + 32 * u, 32, kk);
          }
        }
        if (chain) {
          f32x16v c = zero;
#pragma unroll
          for (int i = 0; i < 8; i++) {
            const float got = G[(kGOd + 16 * cwv + rho(i, h)) * TSG + r];
            c[i] = part[i] + got;
          }
          if (h == 0) c[0] += seed.x; // row 0 = register 0 of lane half 0
          if (cwv == 0) {
#pragma unroll
            for (int i = 0; i < 8; i++) G[(kGOd + rho(i, h)) * TSG + r] = c[i];
          }
          split8(c, 0, bh[0], bl[0]);
          d = mm3(14 + cwv, bh[0], bl[0], zero); // dH1 = W_d2 dOd, this wave's rows
#pragma unroll
          for (int i = 0; i < 16; i++) {
            const int row = own + rho(i, h);
            d[i] = A[(kAH1 + row) * TSA + r] > (_Float16)0.0f ? d[i] : 0.0f;
            G[(kGH1 + row) * TSG + r] = d[i];
          }
          split8(d, 0, bh[0], bl[0]);
          split8(d, 8, bh[1], bl[1]);
          // dFeat = W_d1 dH1: one 32-row tile; wave w finishes accumulator registers [8 w, 8 w + 8) and hands over the others
          part = zero;
#pragma unroll
          for (int j = 0; j < 2; j++) part = mm3(16 + 2 * cwv + j, bh[j], bl[j], part);
          if (ws == 0) {
#pragma unroll
            for (int i = 0; i < 8; i++) G[(kGFeat + rho(8 + i, h)) * TSG + r] = part[8 + i];
          } else {
#pragma unroll
            for (int i = 0; i < 8; i++) G[(kGFeat + rho(i, h)) * TSG + r] = part[i];
          }
        }
        __syncthreads();
        if (dww) { // D1 (the features x half u of dH1) and D2 (rows [32 u, 32 u + 32) of h1 x dOd): the one piece outside the chain's shadow
#pragma unroll
          for (int kk = 0; kk < 2; kk++) {
            bf16x8 xh, xl;
            dw_x(kAFeat, kk, xh, xl);
            dw_mac(dw[4], xh, xl, kGH1 + 32 * u, 32, kk);
            dw_x(kAH1 + 32 * u, kk, xh, xl);
            dw_mac(dw[5], xh, xl, kGOd, 16, kk);
          }
        }
        if (chain) {
          if (ws == 0) {
#pragma unroll
            for (int i = 0; i < 8; i++) G[(kGFeat + rho(i, h)) * TSG + r] += part[i];
          } else {
#pragma unroll
            for (int i = 0; i < 8; i++) G[(kGFeat + rho(8 + i, h)) * TSG + r] += part[8 + i];
          }
        }
        __syncthreads(); STAMP(12);
      }
    } else {
    if (wave < 2) { // dH3 = W_r3 dOrr
      const f32x16v a = layer_tile<true, float>(W + kLLds[4], kLStr[4], 64, wave, G + kGOrr * kTS, 16, lane);
#pragma unroll
      for (int i = 0; i < 16; i++) {
        const int row = 32 * wave + rho(i, h);
        G[(kGH3 + row) * kTS + r] = A[(kAH3 + row) * kTS + r] > (_Float16)0.0f ? a[i] : 0.0f;
      }
    }
    __syncthreads(); STAMP(8);
    if (wave < 2) { // dH2 = W_r2 dH3
      const f32x16v a = layer_tile<true, float>(W + kLLds[3], kLStr[3], 64, wave, G + kGH3 * kTS, 64, lane);
#pragma unroll
      for (int i = 0; i < 16; i++) {
        const int row = 32 * wave + rho(i, h);
        G[(kGH2 + row) * kTS + r] = A[(kAH2 + row) * kTS + r] > (_Float16)0.0f ? a[i] : 0.0f;
      }
    }
    __syncthreads(); STAMP(9);
    if (wave == 0) { // dOd = (W_r1 dH2)[0..15] (+ the density seed on row 0); the SH rows carry no parameters
      const f32x16v a = layer_tile<true, float>(W + kLLds[2], kLStr[2], 32, 0, G + kGH2 * kTS, 64, lane);
      const float sd = seed.x; // every thread of sample s = tid & 31 holds its seed; in wave 0, s == r
#pragma unroll
      for (int i = 0; i < 16; i++) {
        const int row = rho(i, h);
        if (row < 16) G[(kGOd + row) * kTS + r] = a[i] + (row == 0 ? sd : 0.0f);
      }
    }
    __syncthreads(); STAMP(10);
    if (wave < 2) { // dH1 = W_d2 dOd
      const f32x16v a = layer_tile<true, float>(W + kLLds[1], kLStr[1], 64, wave, G + kGOd * kTS, 16, lane);
#pragma unroll
      for (int i = 0; i < 16; i++) {
        const int row = 32 * wave + rho(i, h);
        G[(kGH1 + row) * kTS + r] = A[(kAH1 + row) * kTS + r] > (_Float16)0.0f ? a[i] : 0.0f;
      }
    }
    __syncthreads(); STAMP(11);
    if (wave == 0) { // dFeat = W_d1 dH1
      const f32x16v a = layer_tile<true, float>(W + kLLds[0], kLStr[0], 32, 0, G + kGH1 * kTS, 64, lane);
#pragma unroll
      for (int i = 0; i < 16; i++) G[(kGFeat + rho(i, h)) * kTS + r] = a[i];
    }
    __syncthreads(); STAMP(12);
    }
    // ---- dW[k][o] += sum_s X[k][s] dOut[o][s]: three 32x32 weight tiles per wave, K = the 32 samples
    // (MODE 2: on waves 2 and 3, inside the chain above)
    if constexpr (MODE != 2) {
    if (!(PRV_TRAIN_ABLATE & 2)) {
      // tile q of wave w: {activation row base, gradient row base, valid gradient rows}
      int xa[3], ga[3], gv[3];
      if (wave == 0) { xa[0] = kAH2; ga[0] = kGH3; gv[0] = 32; xa[1] = kAH2; ga[1] = kGH3 + 32; gv[1] = 32; xa[2] = kAH3; ga[2] = kGOrr; gv[2] = 16; }
      else if (wave == 1) { xa[0] = kAH2 + 32; ga[0] = kGH3; gv[0] = 32; xa[1] = kAH2 + 32; ga[1] = kGH3 + 32; gv[1] = 32; xa[2] = kAH3 + 32; ga[2] = kGOrr; gv[2] = 16; }
      else if (wave == 2) { xa[0] = kAIn2; ga[0] = kGH2; gv[0] = 32; xa[1] = kAIn2; ga[1] = kGH2 + 32; gv[1] = 32; xa[2] = kAH1; ga[2] = kGOd; gv[2] = 16; }
      else { xa[0] = kAFeat; ga[0] = kGH1; gv[0] = 32; xa[1] = kAFeat; ga[1] = kGH1 + 32; gv[1] = 32; xa[2] = kAH1 + 32; ga[2] = kGOd; gv[2] = 16; }
      {
      // the three tiles' MFMAs interleaved (independent accumulators), operands loaded a group ahead
      float a[3][4], b[3][4];
      for (int k0 = h; k0 < 32; k0 += 8) {
#pragma unroll
        for (int q = 0; q < 3; q++)
#pragma unroll
          for (int j = 0; j < 4; j++) {
            const int k = k0 + 2 * j;
            a[q][j] = (float)A[(xa[q] + r) * TSA + k];
            b[q][j] = r < gv[q] ? G[(ga[q] + r) * TSG + k] : 0.0f;
          }
#pragma unroll
        for (int j = 0; j < 4; j++)
#pragma unroll
          for (int q = 0; q < 3; q++) dw[q] = mfma32(a[q][j], b[q][j], dw[q]);
      }
      }
    }
    }
    // ---- scatter the feature gradients into the canonical table gradient.  f32 atomics run at the memory
    // side as 64-byte requests (MI355X_MICROARCH.md, global float atomics): one dword per request is the
    // slow shape.  The (entry, weight) pairs are re-dealt through LDS so that F consecutive lanes add the F
    // features of ONE entry (one request carries the whole entry; x-neighbour corners sit on adjacent lane
    // groups and usually share the line too).
    if constexpr (MODE != 2) __syncthreads(); // (MODE 2: the chain's last barrier is this one)
    STAMP(13); // the activation rows are dead: their LDS becomes the staging array
    // (What the adds of the walk below cost beyond their instructions is back-pressure: a wave stalls at its next vector-memory
    // instruction while the CU's queue to the memory side is full -- the unpack phase is 13 us without the adds and 40 with them, and it
    // holds no wait for them: moving every wait for the prefetch in front of the adds changed nothing, r06an / r06ao.  2.6 M requests per
    // launch are 125 us at the memory side's rate; the launch takes 180.)
    {
      // staging = the (dead) activation region: 256 rows x 33 halfs = 2112 uint2 -> 8 levels of 32 samples at a time.
      // A sample's 64 (entry, weight) pairs sit 65 pairs apart: the writers (lanes = samples) then hit 32 different banks
      // (with a stride of 64 all of them hit ONE: half of this kernel's LDS cycles were bank conflicts)
      constexpr int kStageStride = 65;
      static_assert(32 * kStageStride * 8 <= kARows * TSA * 2, "the staging array lives in the activation region");
      uint2* stage = reinterpret_cast<uint2*>(A);
      constexpr int NL = 32 / F, NPASS = NL / 8;
      const bool contributes = live && (seed.x != 0.0f || seed.y != 0.0f || seed.z != 0.0f || seed.w != 0.0f);
#pragma unroll
      for (int pass = 0; pass < NPASS; pass++) {
        if (pass) __syncthreads();
#pragma unroll
        for (int q = 0; q < LPT; q++) {
          const int l = g * LPT + q;
          if (l / 8 != pass) continue;
          if constexpr (SAVED) train_level_corners<F>(lv[l], spos[0], spos[1], spos[2], cidx[q], cw[q]);
#pragma unroll
          for (int c = 0; c < 8; c++)
            stage[s * kStageStride + (l & 7) * 8 + c] = make_uint2(contributes ? cidx[q][c] : 0xffffffffu, __float_as_uint(cw[q][c]));
        }
        __syncthreads();
        if (!(PRV_TRAIN_ABLATE & 1)) {
          // The coarse levels meet the memory-side atomic units as a crowd: every tile of the batch adds into the same few
          // thousand entries of levels 0..3 (4 K ... 150 K entries), and adds to one line serialise there.  A tile's samples
          // are consecutive along their rays, so neighbours in the tile mostly share their corner entries on those levels:
          // thread (level, corner, feature) walks the tile's 32 samples, sums while the entry stays the same and issues ONE
          // add per run (levels 0..3: ~200 adds per tile instead of 1024; the finest levels' runs are one sample long and
          // cost what the item-parallel form cost: 32 steps per thread either way).  F consecutive lanes still add the F
          // features of one entry, x-neighbour corners sit on adjacent lane groups.  -12 % per step at the planner loop's batch.
          if (!(PRV_TRAIN_ABLATE & 4)) {
            if (tid < 8 * 8 * F) {
              const int k = tid % F, c = (tid / F) & 7, l8 = tid / (8 * F), l = pass * 8 + l8;
              // kWays (entry, sum) pairs, the oldest evicted first.  One pair = the run merge of round 4 (samples along ONE
              // ray: an entry comes back only on consecutive samples); patch mode lists a patch depth step by depth step,
              // where a step's rays alternate between the two to four cells the patch straddles (train_rays_patch_kernel)
              // One pair is what i.i.d. rays need, and the walk with four costs 5 % of the step (r06ae): four only for patch batches
              auto walk = [&](auto ways) {
              constexpr int kWays = decltype(ways)::value;
              uint32_t key[kWays];
              float acc[kWays];
#pragma unroll
              for (int w = 0; w < kWays; w++) {
                key[w] = 0xffffffffu;
                acc[w] = 0.0f;
              }
              // the tile's 32 (entry, weighted gradient) pairs first, all of their LDS reads in flight at once: read inside the
              // walk below, every step waited for its own two reads (the walk is a serial chain through the pairs' state)
              uint32_t ek[32];
              float eg[32];
#pragma unroll
              for (int ss = 0; ss < 32; ss++) {
                const uint2 e = stage[ss * kStageStride + l8 * 8 + c];
                ek[ss] = e.x;
                eg[ss] = __uint_as_float(e.y) * G[(kGFeat + l * F + k) * TSG + ss];
              }
#pragma unroll
              for (int ss = 0; ss < 32; ss++) {
                const uint2 e = make_uint2(ek[ss], 0u);
                if (e.x == 0xffffffffu) continue; // a dead sample or one without a gradient
                const float gv = eg[ss];
                bool hit = false;
#pragma unroll
                for (int w = 0; w < kWays; w++)
                  if (e.x == key[w]) { // (keys are distinct: at most one way matches)
                    acc[w] += gv;
                    hit = true;
                  }
                if (!hit) {
                  if (key[kWays - 1] != 0xffffffffu) table_grad_add(P, (size_t)key[kWays - 1] * F + k, acc[kWays - 1]);
#pragma unroll
                  for (int w = kWays - 1; w > 0; w--) {
                    key[w] = key[w - 1];
                    acc[w] = acc[w - 1];
                  }
                  key[0] = e.x;
                  acc[0] = gv;
                }
              }
#pragma unroll
              for (int w = kWays - 1; w >= 0; w--)
                if (key[w] != 0xffffffffu) table_grad_add(P, (size_t)key[w] * F + k, acc[w]);
              };
              // The one-pair walk, flat: a run ends where the entry changes (a sample without a gradient counts as an entry of its
              // own: its sum is never added), ONE predicated region per step -- the add of the finished run -- and a select; which
              // form of add (f32 / fixed point) is decided once per walk, not per add.  The nested form above took ~25 instructions
              // and four branches per step for this case.
              auto walk1 = [&](auto det) {
                constexpr bool kDet = decltype(det)::value;
                uint32_t ek[32];
                float eg[32];
#pragma unroll
                for (int ss = 0; ss < 32; ss++) {
                  const uint2 e = stage[ss * kStageStride + l8 * 8 + c];
                  ek[ss] = e.x;
                  eg[ss] = __uint_as_float(e.y) * G[(kGFeat + l * F + k) * TSG + ss];
                }
                auto add = [&](uint32_t entry, float v) {
                  const size_t idx = (size_t)entry * F + k;
                  if constexpr (kDet) {
                    const long long q = __float2ll_rn(v * (float)(1ull << kGradQBits));
                    atomicAdd(reinterpret_cast<unsigned long long*>(P.table_grad_q) + idx, (unsigned long long)q);
                  } else {
                    atomicAdd(P.table_grad + idx, v);
                  }
                };
                uint32_t key = 0xffffffffu;
                float acc = 0.0f;
#pragma unroll
                for (int ss = 0; ss < 32; ss++) {
                  const uint32_t e = ek[ss];
                  const bool brk = e != key;
                  if (brk && key != 0xffffffffu) add(key, acc);
                  acc = brk ? eg[ss] : acc + eg[ss];
                  key = e;
                }
                if (key != 0xffffffffu) add(key, acc);
              };
              if (P.scatter_ways > 1) walk(std::integral_constant<int, PRV_TRAIN_SCATTER_WAYS>{});
              else if (P.table_grad_q) walk1(std::true_type{});
              else walk1(std::false_type{});
            }
          } else { // dev (timing builds): the item-parallel form of rounds 1-3, one add per (sample, level, corner)
            const int k = tid % F;
            constexpr int kItems = 32 * 8 * 8, kPerPass = 256 / F;
            for (int it = tid / F; it < kItems; it += kPerPass) {
              const uint2 e = stage[(it >> 6) * kStageStride + (it & 63)];
              if (e.x == 0xffffffffu) continue;
              const int ss = it >> 6, l = pass * 8 + ((it >> 3) & 7);
              table_grad_add(P, (size_t)e.x * F + k, __uint_as_float(e.y) * G[(kGFeat + l * F + k) * TSG + ss]);
            }
          }
        }
      }
    }
    __syncthreads(); STAMP(15);
    if (P.tile_live && threadIdx.x == 0) P.tile_live[tile] = 0; // read by this block alone, at the top of the iteration
    tile = tile_next;
    cand = cand_next;
  }
  if (!FWD) { // this block's weight-gradient tiles -> its own slot of the partials (plain stores; a second
    // kernel sums the slots in block order: no same-address atomics, and a reproducible sum)
    // {layer, k base, o base, valid o columns}; accumulator i of lane (r, h) = dW[k = rho(i,h)][o = r]
    int ly[3], kb[3], ob[3], ov[3];
    if (wave == 0) { ly[0] = 3; kb[0] = 0; ob[0] = 0; ov[0] = 32; ly[1] = 3; kb[1] = 0; ob[1] = 32; ov[1] = 32; ly[2] = 4; kb[2] = 0; ob[2] = 0; ov[2] = 16; }
    else if (wave == 1) { ly[0] = 3; kb[0] = 32; ob[0] = 0; ov[0] = 32; ly[1] = 3; kb[1] = 32; ob[1] = 32; ov[1] = 32; ly[2] = 4; kb[2] = 32; ob[2] = 0; ov[2] = 16; }
    else if (wave == 2) { ly[0] = 2; kb[0] = 0; ob[0] = 0; ov[0] = 32; ly[1] = 2; kb[1] = 0; ob[1] = 32; ov[1] = 32; ly[2] = 1; kb[2] = 0; ob[2] = 0; ov[2] = 16; }
    else { ly[0] = 0; kb[0] = 0; ob[0] = 0; ov[0] = 32; ly[1] = 0; kb[1] = 0; ob[1] = 32; ov[1] = 32; ly[2] = 1; kb[2] = 32; ob[2] = 0; ov[2] = 16; }
    float* part = P.mlp_grad_partial + (size_t)(P.slot_base + (int)blockIdx.x) * PRV_MLP_HALFS;
    if constexpr (MODE == 2) { // waves 2 and 3 hold the twelve tiles (six each, the order of the chain's three dW stages)
      if (wave >= 2) {
        const int u = wave - 2;
        const int ly6[6] = {3, 3, 4, 2, 0, 1}, kb6[6] = {32 * u, 32 * u, 32 * u, 0, 0, 32 * u}, ob6[6] = {0, 32, 0, 32 * u, 32 * u, 0}, ov6[6] = {32, 32, 16, 32, 32, 16};
#pragma unroll
        for (int q = 0; q < 6; q++) {
          if (r >= ov6[q]) continue;
#pragma unroll
          for (int i = 0; i < 16; i++) part[kLOff[ly6[q]] + (kb6[q] + rho(i, h)) * kLOut[ly6[q]] + ob6[q] + r] = dw[q][i];
        }
      }
    } else {
#pragma unroll
    for (int q = 0; q < 3; q++) {
      if (r >= ov[q]) continue;
#pragma unroll
      for (int i = 0; i < 16; i++) part[kLOff[ly[q]] + (kb[q] + rho(i, h)) * kLOut[ly[q]] + ob[q] + r] = dw[q][i];
    }
    }
  }
}

// mlp_grad[i] += sum over the blocks' partial slots, in two stages: 16 slot groups summed side by side (grid.y),
// then the 16 group sums of a weight added in group order -- fixed order throughout, 16x the loads in flight
constexpr int kDwGroups = 16;
constexpr int kDwBlocksX = (PRV_MLP_HALFS + 255) / 256; // the first stage is kDwBlocksX x kDwGroups blocks of 256 threads
__device__ __forceinline__ void reduce_dw_block(const float* __restrict__ partial, int n_blocks, float* __restrict__ stage, int bx, int by) {
  const int i = bx * 256 + (int)threadIdx.x;
  if (i >= PRV_MLP_HALFS) return;
  const int per = (n_blocks + kDwGroups - 1) / kDwGroups;
  const int b0 = by * per, b1 = min(n_blocks, b0 + per);
  float acc[4] = {0.f, 0.f, 0.f, 0.f};
  int b = b0;
  for (; b + 3 < b1; b += 4) {
#pragma unroll
    for (int u = 0; u < 4; u++) acc[u] += partial[(size_t)(b + u) * PRV_MLP_HALFS + i];
  }
  for (; b < b1; b++) acc[0] += partial[(size_t)b * PRV_MLP_HALFS + i];
  stage[(size_t)by * PRV_MLP_HALFS + i] = (acc[0] + acc[1]) + (acc[2] + acc[3]);
}
__global__ __launch_bounds__(256) void train_reduce_dw_kernel(const float* __restrict__ partial, int n_blocks,
                                                              float* __restrict__ stage) {
  reduce_dw_block(partial, n_blocks, stage, (int)blockIdx.x, (int)blockIdx.y);
}

__global__ __launch_bounds__(256) void train_reduce_dw2_kernel(const float* __restrict__ stage, float* __restrict__ mlp_grad) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= PRV_MLP_HALFS) return;
  float a = 0.0f;
#pragma unroll
  for (int g = 0; g < kDwGroups; g++) a += stage[(size_t)g * PRV_MLP_HALFS + i];
  mlp_grad[i] += a;
}

// ------------------------------------------------------------------ forward at inference speed

__device__ __forceinline__ void end_step(const AdamParams& P, uint32_t* sample_count, float lr, float beta1, float beta2);

// the 24 MFMA A-fragments of the current fp16 weights (the layout prv_api.cpp: prepack_fragments gives the
// render kernel), rebuilt on the device after every optimiser step: one thread per fragment element
__device__ __forceinline__ int frag_hidden_k(int s, int h, int j) { return 32 * (s >> 1) + 16 * (s & 1) + 8 * (j >> 2) + 4 * h + (j & 3); }

// which canonical weight element i of the fragment array (forward fragments, then the backward ones) holds; -1: padding
__device__ __forceinline__ int prepack_source(int n_features, int i) {
  if (i >= kNumFrags * kFragHalfs) {
    // backward fragments: A[row = input unit][K = output unit] of every layer, fp16 as stored; K order = the order in
    // which a lane holds the previous backward layer's accumulator rows (frag_hidden_k), as in the forward fragments
    const int ib = i - kNumFrags * kFragHalfs;
    const int f = ib / kFragHalfs, ln = (ib % kFragHalfs) >> 3, j = ib & 7;
    const int r = ln & 31, h = ln >> 5;
    int layer, mt, st;
    if (f < 2) { layer = 4; mt = f; st = 0; }                        // dH3 = W_r3 dOrr   (K = 16 colour outputs)
    else if (f < 10) { layer = 3; mt = (f - 2) >> 2; st = (f - 2) & 3; } // dH2 = W_r2 dH3
    else if (f < 14) { layer = 2; mt = 0; st = f - 10; }              // dIn2 = W_r1 dH2
    else if (f < 16) { layer = 1; mt = f - 14; st = 0; }              // dH1 = W_d2 dOd    (K = 16 density outputs)
    else { layer = 0; mt = 0; st = f - 16; }                          // dFeat = W_d1 dH1
    int o; // output unit on the K dimension
    if (layer == 4) o = 8 * h + j;                        // colour logits: lane half h holds outputs 8h..8h+7
    else if (layer == 1) o = (j & 3) + 8 * (j >> 2) + 4 * h; // the density output as its accumulator rows come
    else o = frag_hidden_k(st, h, j);
    const int k_in = 32 * mt + r;
    return k_in < kLIn[layer] && o < kLOut[layer] ? kLOff[layer] + k_in * kLOut[layer] + o : -1;
  }
  const int f = i / kFragHalfs, lane = (i % kFragHalfs) >> 3, j = i & 7;
  const int r = lane & 31, h = lane >> 5;
  int layer, mt, st;
  if (f < 4) { layer = 0; mt = f >> 1; st = f & 1; }
  else if (f < 8) { layer = 1; mt = 0; st = f - 4; }
  else if (f < 12) { layer = 2; mt = (f - 8) >> 1; st = (f - 8) & 1; }
  else if (f < 20) { layer = 3; mt = (f - 12) >> 2; st = (f - 12) & 3; }
  else { layer = 4; mt = 0; st = f - 20; }
  int k;
  if (layer == 0) k = n_features == 4 ? 4 * (2 * (st * 2 + j / 4) + h) + j % 4 : 2 * (2 * (st * 4 + j / 2) + h) + j % 2;
  else if (layer == 2) k = st == 0 ? frag_hidden_k(0, h, j) : 16 + 8 * h + j;
  else k = frag_hidden_k(st, h, j);
  const int out = 32 * mt + r;
  return out < kLOut[layer] ? kLOff[layer] + k * kLOut[layer] + out : -1;
}

// (a launch of its own at trainer creation; inside a step every thread of the MLP's Adam pass writes its own weight's two places)
__global__ __launch_bounds__(256) void prepack_frags_kernel(const uint16_t* __restrict__ mlp, int n_features,
                                                            uint16_t* __restrict__ frags, AdamParams P,
                                                            uint32_t* sample_count, float lr) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i == 0 && P.state) end_step(P, sample_count, lr, P.beta1, P.beta2); // the step's last kernel: it opens the next one too
  if (i >= (kNumFrags + kBwdFrags) * kFragHalfs) return;
  const int src = prepack_source(n_features, i);
  frags[i] = src >= 0 ? mlp[src] : (uint16_t)0;
}

// the inverse of prepack_source: every canonical weight sits at exactly one place of the forward fragments and one of the
// backward ones (checked when this was written: both maps are bijections onto the PRV_MLP_HALFS weights, F = 4 and F = 2)
__global__ __launch_bounds__(256) void frag_positions_kernel(int n_features, int* __restrict__ frag_pos) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= (kNumFrags + kBwdFrags) * kFragHalfs) return;
  const int src = prepack_source(n_features, i);
  if (src >= 0) frag_pos[(i >= kNumFrags * kFragHalfs ? PRV_MLP_HALFS : 0) + src] = i;
}

// logits of every live sample with the render kernel's machinery: a lane pair per sample, each lane encodes
// every other level from the CANONICAL table (what the optimiser updates), both MLPs as 24
// v_mfma_f32_32x32x16_f16 on the prepacked fragments -- 32 samples per wave and round, no LDS round trips
template <int F>
__global__ __launch_bounds__(256) void train_forward_fast_kernel(TrainTileParams P, const half8* __restrict__ frags) {
  __shared__ half8 wl[kNumFrags * 64];
  __shared__ LevelCanon lv[16];
  for (int i = threadIdx.x; i < kNumFrags * 64; i += 256) wl[i] = frags[i];
  if (threadIdx.x < 16 * (int)(sizeof(LevelCanon) / 4))
    reinterpret_cast<uint32_t*>(lv)[threadIdx.x] = reinterpret_cast<const uint32_t*>(P.levels)[threadIdx.x];
  __syncthreads();
  const int lane = threadIdx.x & 63, r = lane & 31, h = lane >> 5;
  const uint32_t n_samples = batch_samples(P.sample_count, P.state);
  const uint32_t n_tiles = (n_samples + 31u) / 32u;
  constexpr int LH = 16 / F;
  for (uint32_t tile = blockIdx.x * 4u + (threadIdx.x >> 6); tile < n_tiles; tile += gridDim.x * 4u) {
    const uint32_t sid = tile * 32u + (uint32_t)r;
    const bool live = sid < n_samples;
    float pos[3] = {0.5f, 0.5f, 0.5f}, dir[3] = {0.f, 0.f, 1.f};
    if (live) {
      const uint2 sr = P.samples[sid];
      const TrainRay* ray = P.rays + sr.x;
      const float t = fmaf((float)sr.y + ray->jitter, ray->dt, ray->t0);
#pragma unroll
      for (int a = 0; a < 3; a++) {
        dir[a] = ray->d[a];
        pos[a] = fmaf(t, dir[a], ray->o[a]);
      }
    }
    half8 f0, f1;
#pragma unroll
    for (int j = 0; j < LH; j++) {
      const LevelCanon L = lv[2 * j + h];
      float f[F], cw[8];
      uint32_t ci[8];
      train_encode_level<F>(P.table, L, pos[0], pos[1], pos[2], f, ci, cw);
#pragma unroll
      for (int k = 0; k < F; k++) {
        const int e = j * F + k;
        if (e < 8) f0[e] = (_Float16)f[k];
        else f1[e - 8] = (_Float16)f[k];
      }
    }
    const half8 shf = sh_fragment(h, dir[0], dir[1], dir[2]);
    // the B fragments the lanes hold between the layers ARE the activations the backward pass needs: kept, 16 bytes
    // per lane and slot, a wave-instruction = 1 KB contiguous (layout: act_row); the backward tiles then skip the table
    // gather and the five forward layers
    uint4* act = P.act != nullptr && tile * 32u + 32u <= P.act_cap ? P.act + (size_t)tile * kActTileWords + lane : nullptr;
    // (non-temporal: 138 MB per step written once here and read once by the backward pass, a kernel later -- they need not push the table's
    // lines out of the L2 on their way; stores and loads both marked: one trainer 0.330 -> 0.316 ms per step, r06av)
    typedef unsigned int u32x4v __attribute__((ext_vector_type(4)));
    auto keep = [&](int slot, const half8& v) { __builtin_nontemporal_store(__builtin_bit_cast(u32x4v, v), reinterpret_cast<u32x4v*>(act + slot * 64)); };
    if (act) {
      keep(0, f0);
      keep(1, f1);
      keep(7, shf);
      if (h == 0) act[kActPosWord] = __builtin_bit_cast(uint4, make_float4(pos[0], pos[1], pos[2], 0.0f)); // act points at this lane's word: + r
    }
    const MlpOut mo = mlp_forward(wl, lane, f0, f1, shf, [&](int stage, const half8* v) {
      if (!act) return;
      if (stage == 1) keep(6, v[0]);
      else
#pragma unroll
        for (int t = 0; t < 4; t++) keep((stage == 0 ? 2 : stage == 2 ? 8 : 12) + t, v[t]);
    });
    if (live && h == 0) P.logits[sid] = make_float4(mo.dens[0], mo.rgb[0], mo.rgb[1], mo.rgb[2]);
  }
}

// ------------------------------------------------------------------ compositing, loss, gradient seeds

// one WAVE = one ray, one lane = one sample (two chunks of 64 for up to 128 samples): transmittance by a
// prefix product across the lanes, colour by wave sums, the suffix sums of the backward pass by a reverse scan
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int d = 32; d >= 1; d >>= 1) v += __shfl_xor(v, d);
  return v;
}
__device__ __forceinline__ float scan_mul_incl(float v, int lane) {
#pragma unroll
  for (int d = 1; d < 64; d <<= 1) {
    const float u = __shfl_up(v, d);
    if (lane >= d) v *= u;
  }
  return v;
}
// suffix sum: v_i + v_{i+1} + ... + v_63
__device__ __forceinline__ float scan_add_rev_incl(float v, int lane) {
#pragma unroll
  for (int d = 1; d < 64; d <<= 1) {
    const float u = __shfl_down(v, d);
    if (lane + d < 64) v += u;
  }
  return v;
}

__global__ __launch_bounds__(256) void train_composite_kernel(TrainCompositeParams P) {
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const uint32_t j = blockIdx.x * 4u + (threadIdx.x >> 6);
  const uint32_t n_active = P.state->n_active;
  if (j >= n_active) {
    if (j < (uint32_t)P.n_rays && (threadIdx.x & 63) == 0) { // rays outside this step's budget count for nothing
      P.ray_loss[j] = 0.0f;
      P.ray_used[j] = 0u;
    }
    return;
  }
  TrainRay* ray = P.rays + j;
  const uint32_t n = P.state->overflow ? 0u : ray->n_live, off = ray->offset; // (a batch that overflowed the list is void)
  const float dt = ray->dt;
  // chunks of 64 samples, front to back: the transmittance at a chunk's start and the samples it used stay in LDS for the
  // way back (<= 16 chunks: the engine's 1024 steps; 2 for the 128 uniform samples)
  constexpr int kChunks = kMaxTrainSteps / 64;
  __shared__ float t_start[4][kChunks];
  __shared__ uint32_t c_used[4][kChunks];
  const int n_chunks = (int)((n + 63u) >> 6);
  auto list_pos = [&](uint32_t k) { return P.slot_of ? P.slot_of[(size_t)j * (size_t)P.S + k] : off + k; };
  // a chunk's forward values from the logits (the same arithmetic on the way out and on the way back)
  struct Chunk {
    float sg, al, rgb[3], incl;
  };
  auto chunk_values = [&](bool have, uint32_t sid, Chunk& c) {
    float4 lg = make_float4(0.f, 0.f, 0.f, 0.f);
    if (have) lg = P.logits[sid];
    c.sg = have ? expf(lg.x + P.density_bias) : 0.0f;
    c.al = have ? 1.0f - expf(-(c.sg * dt)) : 0.0f;
    c.rgb[0] = 1.0f / (1.0f + expf(-lg.y));
    c.rgb[1] = 1.0f / (1.0f + expf(-lg.z));
    c.rgb[2] = 1.0f / (1.0f + expf(-lg.w));
    c.incl = scan_mul_incl(1.0f - c.al, lane);
  };
  float T = 1.0f, C[3] = {0.f, 0.f, 0.f};
  uint32_t used = 0;
  int last_ch = -1; // the last chunk that was composited (the ray terminated there, or it is the ray's last)
  for (int ch = 0; ch < n_chunks; ch++) {
    const uint32_t k = (uint32_t)ch * 64u + (uint32_t)lane;
    const bool have = k < n;
    Chunk c;
    chunk_values(have, have ? list_pos(k) : 0u, c);
    float excl = __shfl_up(c.incl, 1);
    if (lane == 0) excl = 1.0f;
    const float tb = T * excl;
    const float T_after = T * c.incl;
    const unsigned long long term = __ballot(have && T_after < P.min_T);
    uint32_t cnt = (uint32_t)__popcll(__ballot(have)); // samples of this chunk that exist
    if (term) cnt = (uint32_t)__builtin_ctzll(term) + 1u; // the terminating sample is the last one used
    const float wgt = have && (uint32_t)lane < cnt ? c.al * tb : 0.0f;
#pragma unroll
    for (int q = 0; q < 3; q++) C[q] += wave_sum(wgt * c.rgb[q]);
    if (lane == 0) {
      t_start[wv][ch] = T;
      c_used[wv][ch] = cnt;
    }
    if (cnt > 0) T = __shfl(T_after, (int)cnt - 1);
    used += cnt;
    last_ch = ch;
    if (term) break;
  }
  float dC[3], loss = 0.0f, tail[3];
  const float inv = 1.0f / (3.0f * (float)n_active);
#pragma unroll
  for (int c = 0; c < 3; c++) {
    const float e = fmaf(T, ray->bg[c], C[c]) - ray->target[c];
    loss = fmaf(e, e, loss);
    dC[c] = 2.0f * e * inv;
    tail[c] = T * ray->bg[c]; // what lies behind the current chunk
  }
  if (lane == 0) {
    ray->n_used = used;
    P.ray_loss[j] = loss * inv;
    P.ray_used[j] = used;
  }
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); // lane 0's LDS notes -> every lane of the wave
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
  // seeds, last chunk first: suffix_i = sum_{j>i} w_j c_j + T_final bg
  for (int ch = n_chunks - 1; ch >= 0; ch--) {
    const uint32_t k = (uint32_t)ch * 64u + (uint32_t)lane;
    const bool have = k < n;
    const uint32_t sid = have ? list_pos(k) : 0u;
    if (ch > last_ch) { // behind the ray's termination: never evaluated on the way out, no gradient
      if (have) P.seeds[sid] = make_float4(0.f, 0.f, 0.f, 0.f);
      continue;
    }
    Chunk c;
    chunk_values(have, sid, c);
    float excl = __shfl_up(c.incl, 1);
    if (lane == 0) excl = 1.0f;
    const float tb = t_start[wv][ch] * excl;
    const bool usedl = have && (uint32_t)lane < c_used[wv][ch];
    const float wgt = usedl ? c.al * tb : 0.0f;
    const float T_after = tb * (1.0f - c.al);
    float d_sigma = 0.0f, d_orr[3];
#pragma unroll
    for (int q = 0; q < 3; q++) {
      const float wc = wgt * c.rgb[q];
      const float incl = scan_add_rev_incl(wc, lane);
      const float suffix = (incl - wc) + tail[q];
      d_sigma += dC[q] * dt * (T_after * c.rgb[q] - suffix);
      d_orr[q] = dC[q] * wgt * c.rgb[q] * (1.0f - c.rgb[q]);
      tail[q] += __shfl(incl, 0);
    }
    if (have) P.seeds[sid] = usedl ? make_float4(d_sigma * c.sg, d_orr[0], d_orr[1], d_orr[2]) : make_float4(0.f, 0.f, 0.f, 0.f);
    if (usedl && P.tile_live) P.tile_live[sid >> 5] = 1; // the backward pass walks the tiles that hold a used sample (and clears the byte)
  }
}

// first node of a step: the bias-corrected learning rate of step n = state->step + 1, the sample counter reset
__global__ void train_begin_kernel(TrainState* state, uint32_t* sample_count, float lr, float beta1, float beta2) {
  const double n = (double)(state->step + 1u);
  state->lr_t = (float)((double)lr * sqrt(1.0 - pow((double)beta2, n)) / (1.0 - pow((double)beta1, n)));
  sample_count[state->step & 1u] = 0u;
  sample_count[kOrderWord + (state->step & 1u)] = 0u;
}

// ------------------------------------------------------------------ optimiser

__device__ __forceinline__ void adam_update(const AdamParams& P, float lr_t, float g, float& w, float& m, float& v) {
  m = fmaf(P.beta1, m, (1.0f - P.beta1) * g);
  v = fmaf(P.beta2, v, ((1.0f - P.beta2) * g) * g);
  w = w - (lr_t * m) / (sqrtf(v) + P.eps);
}

// four table scalars per thread (one 16-byte load of the gradient; most are zero and cost nothing more)
// end-of-step bookkeeping, ONE thread of the step's last kernel, which must not read any of these itself: the step
// counter, the sample budget (integer rule, mirrored by oracle/prv_train.c) and -- when that kernel also opens the
// next step (sample_count != NULL) -- the next step's bias-corrected learning rate and the sample counter reset
// the slices of the step's loss / used-sample sums (the backward launch left them) in slice order
__device__ __forceinline__ void finish_loss(const double* __restrict__ part, int n_rays, TrainState* state, unsigned long long* used) {
  double t = 0.0;
  unsigned long long u = 0ull;
  for (int s = 0; s < (n_rays + 1023) / 1024; s++) {
    t += part[2 * s];
    u += (unsigned long long)__double_as_longlong(part[2 * s + 1]);
  }
  state->losses[state->step - state->step0] = (float)t;
  *used = u;
}
__global__ void train_loss_finish_kernel(const double* part, int n_rays, TrainState* state, unsigned long long* used) { finish_loss(part, n_rays, state, used); }

__device__ __forceinline__ void end_step(const AdamParams& P, uint32_t* sample_count, float lr, float beta1, float beta2) {
  if (P.loss_part) finish_loss(P.loss_part, P.n_rays, P.state, P.used);
  const uint32_t done = P.state->step + 1u;
  P.state->step = done;
  P.state->n_active = next_active_rays(P.target_samples, P.state->n_active, *P.used, P.n_rays);
  if (sample_count) {
    const double n = (double)(done + 1u);
    P.state->lr_t = (float)((double)lr * sqrt(1.0 - pow((double)beta2, n)) / (1.0 - pow((double)beta1, n)));
    // the finished step's counter (the word of ITS parity) is free again: the step after the next one lists into it (the next
    // step's word was cleared a step ago -- its rays may have been listed already, beside this step's Adam pass)
    uint32_t* mine = sample_count + ((done - 1u) & 1u);
    sample_count[7] = *mine; // dev: the finished step's listed-sample count stays readable (prv_train_api.inc: PRV_TRAIN_TIMING)
    *mine = 0u;
    sample_count[kOrderWord + ((done - 1u) & 1u)] = 0u; // ... and its append ticket (deterministic batches)
  }
}

// The table's optimiser state is ONE record per group of four scalars: {w[4], m[4], v[4]}, 48 contiguous bytes (the f32
// master weights and the moments are the trainer's own: no other code reads them in place).  A touched group then reads and
// writes one or two lines of state where three separate arrays cost three partial lines each way.
__global__ __launch_bounds__(256) void adam_table_kernel(AdamParams P, size_t n, float* __restrict__ grad,
                                                         float* __restrict__ wmv, uint16_t* __restrict__ w16,
                                                         unsigned n_adam_blocks, const float* __restrict__ dw_partial, int dw_slots,
                                                         float* __restrict__ dw_stage, unsigned n_dw_blocks, TrainRaysParams R,
                                                         long long* __restrict__ grad_q) {
  // the blocks behind the table's own do the first stage of the MLP's weight-gradient reduction: both read what the backward
  // launch left and neither needs the other, so the reduction is no node of its own in the step's chain (round 5)
  // ... and the blocks behind THOSE list the next step's ray batch (TrainRaysParams::next): it needs the step's used-sample
  // count and the occupancy grid, neither of which this launch touches, and the step's chain is one kernel shorter
  // Dispatch order (workgroups start in index order): the ray blocks first, then the reduction's, then the table's own -- the
  // first two are few and latency-bound (a chain of dependent loads each; most ray blocks only find that they lie beyond the
  // step's budget) and were this launch's tail when they came last; in front, the table's ~9 K bandwidth-bound blocks fill in
  // behind them.
  const unsigned n_ray_blocks = gridDim.x - n_adam_blocks - n_dw_blocks;
  if (blockIdx.x < n_ray_blocks) {
    train_rays_block_any(R, blockIdx.x);
    return;
  }
  if (blockIdx.x < n_ray_blocks + n_dw_blocks) {
    const int k = (int)(blockIdx.x - n_ray_blocks);
    reduce_dw_block(dw_partial, dw_slots, dw_stage, k % kDwBlocksX, k / kDwBlocksX);
    return;
  }
  const unsigned bx = blockIdx.x - n_ray_blocks - n_dw_blocks;
  if (bx == 0 && threadIdx.x == 0) P.state->lr_cur = P.state->lr_t; // for the step's last kernel (TrainState::lr_cur)
  const size_t i4 = ((size_t)bx * 256 + threadIdx.x) * 4;
  if (i4 >= n) return;
  const float lr_t = P.state->lr_t;
  float4* rec = reinterpret_cast<float4*>(wmv + i4 * 3);
  if (i4 + 4 <= n) {
    float4 g4;
    if (grad_q) { // deterministic (tests): the step's gradient was summed in fixed point (table_grad_add)
      typedef long long ll2 __attribute__((ext_vector_type(2)));
      ll2* q = reinterpret_cast<ll2*>(grad_q + i4);
      const ll2 q0 = q[0], q1 = q[1];
      if ((q0.x | q0.y | q1.x | q1.y) == 0ll) return;
      q[0] = ll2{0ll, 0ll};
      q[1] = ll2{0ll, 0ll};
      constexpr double inv = 1.0 / (double)(1ull << kGradQBits);
      g4 = make_float4((float)((double)q0.x * inv), (float)((double)q0.y * inv), (float)((double)q1.x * inv), (float)((double)q1.y * inv));
    } else {
      g4 = *reinterpret_cast<const float4*>(grad + i4);
    }
    if (g4.x == 0.0f && g4.y == 0.0f && g4.z == 0.0f && g4.w == 0.0f) return; // sparse: untouched entries keep their moments
    if (!grad_q) *reinterpret_cast<float4*>(grad + i4) = make_float4(0.f, 0.f, 0.f, 0.f);
    // the touched scalars of a group are one table entry (F = 4) or two (F = 2): whole-group 16-byte loads and
    // stores of weight and moments instead of four scattered 4-byte ones each (untouched lanes are rewritten as read)
    const float g[4] = {g4.x, g4.y, g4.z, g4.w};
    const float4 w4 = rec[0], m4 = rec[1], v4 = rec[2];
    float ww[4] = {w4.x, w4.y, w4.z, w4.w}, mm[4] = {m4.x, m4.y, m4.z, m4.w}, vv[4] = {v4.x, v4.y, v4.z, v4.w};
#pragma unroll
    for (int k = 0; k < 4; k++)
      if (g[k] != 0.0f) adam_update(P, lr_t, g[k], ww[k], mm[k], vv[k]);
    rec[0] = make_float4(ww[0], ww[1], ww[2], ww[3]);
    rec[1] = make_float4(mm[0], mm[1], mm[2], mm[3]);
    rec[2] = make_float4(vv[0], vv[1], vv[2], vv[3]);
    typedef uint16_t ushort4v __attribute__((ext_vector_type(4)));
    ushort4v h4;
#pragma unroll
    for (int k = 0; k < 4; k++) h4[k] = g[k] != 0.0f ? __builtin_bit_cast(uint16_t, (_Float16)ww[k]) : w16[i4 + k];
    *reinterpret_cast<ushort4v*>(w16 + i4) = h4;
    return;
  }
  float* r = wmv + i4 * 3;
  for (size_t i = i4; i < n; i++) { // the last, partial group
    float g;
    if (grad_q) {
      g = (float)((double)grad_q[i] * (1.0 / (double)(1ull << kGradQBits)));
      grad_q[i] = 0ll;
    } else {
      g = grad[i];
    }
    if (g == 0.0f) continue;
    if (!grad_q) grad[i] = 0.0f;
    const int k = (int)(i - i4);
    float ww = r[k], mm = r[4 + k], vv = r[8 + k];
    adam_update(P, lr_t, g, ww, mm, vv);
    r[k] = ww;
    r[4 + k] = mm;
    r[8 + k] = vv;
    w16[i] = __builtin_bit_cast(uint16_t, to_half(ww));
  }
}

// the f32 master weights of a fresh trainer: the model's fp16 table widened into the records (moments zeroed by the caller)
__global__ __launch_bounds__(256) void widen_table_kernel(const uint16_t* __restrict__ in, size_t n, float* __restrict__ wmv) {
  const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (i < n) wmv[(i >> 2) * 12 + (i & 3)] = (float)__builtin_bit_cast(_Float16, in[i]);
}

// ... and back: the w part of every record, contiguous (prv_train_master: one linear device-to-host copy afterwards)
__global__ __launch_bounds__(256) void narrow_table_kernel(const float* __restrict__ wmv, size_t n, float* __restrict__ out) {
  const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (i < n) out[i] = wmv[(i >> 2) * 12 + (i & 3)];
}

__global__ __launch_bounds__(256) void adam_mlp_kernel(AdamParams P, float l2_reg, float* __restrict__ grad,
                                                       float* __restrict__ w, float* __restrict__ m,
                                                       float* __restrict__ v, uint16_t* __restrict__ w16,
                                                       float* __restrict__ w16_as_f32, const float* __restrict__ stage,
                                                       int end_of_step, uint16_t* __restrict__ frags, const int* __restrict__ frag_pos,
                                                       uint32_t* sample_count, float lr) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= PRV_MLP_HALFS) return;
  float gsum = grad[i];
  if (stage) { // the second stage of the weight-gradient reduction, fused: the group sums in group order
#pragma unroll
    for (int q = 0; q < kDwGroups; q++) gsum += stage[(size_t)q * PRV_MLP_HALFS + i];
  }
  const float g = fmaf(l2_reg, w[i], gsum);
  grad[i] = 0.0f;
  float ww = w[i], mm = m[i], vv = v[i];
  // frags: this is the step's last kernel and thread 0's end_step writes the NEXT step's lr_t while other blocks still run:
  // the rate of the step in flight is read from the copy the table's Adam launch left (TrainState::lr_cur)
  adam_update(P, frags ? P.state->lr_cur : P.state->lr_t, g, ww, mm, vv);
  w[i] = ww;
  m[i] = mm;
  v[i] = vv;
  const _Float16 hh = to_half(ww);
  w16[i] = __builtin_bit_cast(uint16_t, hh);
  w16_as_f32[i] = (float)hh;
  if (frags) {
    // the MFMA fragments of the next forward / backward pass: this weight's two places (the padding elements never change).
    // Until round 5 a kernel of its own behind this one (prepack_frags_kernel), one more node in every member's chain.
    frags[frag_pos[i]] = __builtin_bit_cast(uint16_t, hh);
    frags[frag_pos[PRV_MLP_HALFS + i]] = __builtin_bit_cast(uint16_t, hh);
    if (i == 0) end_step(P, sample_count, lr, P.beta1, P.beta2); // closes the step, opens the next
    return;
  }
  if (i == 0 && end_of_step) end_step(P, nullptr, 0.f, 0.f, 0.f); // last node of a step unless a later kernel takes that role
}

__global__ __launch_bounds__(256) void widen_kernel(const uint16_t* __restrict__ in, size_t n, float* __restrict__ out) {
  const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (i < n) out[i] = (float)__builtin_bit_cast(_Float16, in[i]);
}

// ------------------------------------------------------------------ density grid

// one thread = one occupancy cell: density at the cell centre (encoder + density MLP, fp16 semantics of
// inference, f32 accumulation), EMA, threshold.  32 consecutive cells = one word of the bitfield.
template <int F>
__global__ __launch_bounds__(256) void density_refresh_kernel(DensityParams P) {
  __shared__ float W1[32 * 64], W2[64];
  __shared__ LevelCanon lv[16];
  if (threadIdx.x < 16 * (int)(sizeof(LevelCanon) / 4))
    reinterpret_cast<uint32_t*>(lv)[threadIdx.x] = reinterpret_cast<const uint32_t*>(P.levels)[threadIdx.x];
  for (int i = threadIdx.x; i < 32 * 64; i += 256) W1[i] = P.mlp[i];
  for (int i = threadIdx.x; i < 64; i += 256) W2[i] = P.mlp[2048 + i * 16]; // output 0 of layer D2
  __syncthreads();
  const int R = P.occ_res;
  const uint32_t n_cells = (uint32_t)R * R * R;
  const uint32_t cell = blockIdx.x * 256u + threadIdx.x;
  bool on = false;
  if (cell < n_cells) {
    const uint32_t x = cell % R, y = (cell / R) % R, z = cell / (R * R);
    const float invR = 1.0f / (float)R;
    const float px = ((float)x + 0.5f) * invR, py = ((float)y + 0.5f) * invR, pz = ((float)z + 0.5f) * invR;
    float h1[64];
#pragma unroll
    for (int o = 0; o < 64; o++) h1[o] = 0.0f;
    for (int l = 0; l < 32 / F; l++) {
      float f[F], cw[8];
      uint32_t ci[8];
      const LevelCanon L = lv[l];
      train_encode_level<F>(P.table, L, px, py, pz, f, ci, cw);
#pragma unroll
      for (int k = 0; k < F; k++) {
        const float* wr = W1 + (l * F + k) * 64;
#pragma unroll
        for (int o = 0; o < 64; o++) h1[o] = fmaf(f[k], wr[o], h1[o]);
      }
    }
    float od0 = 0.0f;
#pragma unroll
    for (int o = 0; o < 64; o++) od0 = fmaf((float)(_Float16)fmaxf(h1[o], 0.0f), W2[o], od0);
    const float sigma = expf(od0 + P.density_bias);
    const float e = fmaxf(P.ema[cell] * P.decay, sigma);
    P.ema[cell] = e;
    on = e > P.thresh;
  }
  const unsigned long long b = __ballot(on);
  if ((threadIdx.x & 31) == 0 && cell < n_cells) P.occ[cell >> 5] = (uint32_t)(b >> (threadIdx.x & 32));
}

// the same refresh on the matrix cores: a wave = 32 cells (lane pair per cell, every other level each), the
// density MLP alone (fragments 0..7: 8 v_mfma_f32_32x32x16_f16), EMA + threshold, one ballot = one bitfield word
template <int F>
__global__ __launch_bounds__(256) void density_refresh_fast_kernel(DensityParams P, const half8* __restrict__ frags) {
  __shared__ half8 wl[8 * 64];
  __shared__ LevelCanon lv[16];
  for (int i = threadIdx.x; i < 8 * 64; i += 256) wl[i] = frags[i];
  if (threadIdx.x < 16 * (int)(sizeof(LevelCanon) / 4))
    reinterpret_cast<uint32_t*>(lv)[threadIdx.x] = reinterpret_cast<const uint32_t*>(P.levels)[threadIdx.x];
  __syncthreads();
  const int lane = threadIdx.x & 63, r = lane & 31, h = lane >> 5;
  const int R = P.occ_res;
  const uint32_t n_cells = (uint32_t)R * R * R, n_words = (n_cells + 31u) / 32u;
  const float invR = 1.0f / (float)R;
  constexpr int LH = 16 / F;
  const f32x16 zero = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
  for (uint32_t word = blockIdx.x * 4u + (threadIdx.x >> 6); word < n_words; word += gridDim.x * 4u) {
    const uint32_t cell = word * 32u + (uint32_t)r;
    const bool have = cell < n_cells;
    const uint32_t cc = have ? cell : 0u;
    const uint32_t x = cc % R, y = (cc / R) % R, z = cc / (R * R);
    const float px = ((float)x + 0.5f) * invR, py = ((float)y + 0.5f) * invR, pz = ((float)z + 0.5f) * invR;
    half8 f0, f1;
#pragma unroll
    for (int j = 0; j < LH; j++) {
      const LevelCanon L = lv[2 * j + h];
      float f[F], cw[8];
      uint32_t ci[8];
      train_encode_level<F>(P.table, L, px, py, pz, f, ci, cw);
#pragma unroll
      for (int k = 0; k < F; k++) {
        const int e = j * F + k;
        if (e < 8) f0[e] = (_Float16)f[k];
        else f1[e - 8] = (_Float16)f[k];
      }
    }
    half8 hf[4];
    { // density layer 1: 32 -> 64 (as mlp_forward)
      f32x16 a0 = mfma(wl[0 * 64 + lane], f0, zero);
      f32x16 a1 = mfma(wl[2 * 64 + lane], f0, zero);
      a0 = mfma(wl[1 * 64 + lane], f1, a0);
      a1 = mfma(wl[3 * 64 + lane], f1, a1);
      hf[0] = pack8<true>(a0, 0);
      hf[1] = pack8<true>(a0, 8);
      hf[2] = pack8<true>(a1, 0);
      hf[3] = pack8<true>(a1, 8);
    }
    f32x16 a = mfma(wl[4 * 64 + lane], hf[0], zero); // density layer 2, row 0 = the density logit
    a = mfma(wl[5 * 64 + lane], hf[1], a);
    a = mfma(wl[6 * 64 + lane], hf[2], a);
    a = mfma(wl[7 * 64 + lane], hf[3], a);
    bool on = false;
    if (have && h == 0) {
      const float sigma = expf(a[0] + P.density_bias);
      const float e = fmaxf(P.ema[cell] * P.decay, sigma);
      P.ema[cell] = e;
      on = e > P.thresh;
    }
    const unsigned long long b = __ballot(on);
    if (lane == 0) P.occ[word] = (uint32_t)b; // lanes 0..31 = lane half 0 = the 32 cells of the word
  }
}

} // namespace

// ------------------------------------------------------------------ launchers

size_t train_tile_lds_bytes(bool fwd, int mode) {
  const int tsa = mode == 2 ? kTSA2 : kTS, tsg = mode == 2 ? kTSG2 : kTS;
  return 2u * (size_t)(kWLds + kARows * tsa) + (fwd ? 0u : 4u * (size_t)(kGRows * tsg));
}

hipError_t launch_train_rays(const TrainRaysParams& P, hipStream_t s) {
  const int pp = P.patch_w * P.patch_h;
  if (pp > 1) hipLaunchKernelGGL(train_rays_patch_kernel, dim3((P.n_rays + pp - 1) / pp), dim3(64 * pp), 0, s, P);
  else hipLaunchKernelGGL(train_rays_kernel, dim3((P.n_rays + 3) / 4), dim3(256), 0, s, P);
  return hipGetLastError();
}

template <int F, bool FWD, int MODE = 0>
static hipError_t launch_tile(const TrainTileParams& P, int n_blocks, hipStream_t s) {
  hipLaunchKernelGGL((train_tile_kernel<F, FWD, MODE>), dim3(n_blocks), dim3(256), train_tile_lds_bytes(FWD, MODE), s, P);
  return hipGetLastError();
}

// dynamic LDS above 64 KB has to be allowed per kernel, once, outside any stream capture
hipError_t train_prepare_kernels() {
  hipError_t e;
#define PRV_ALLOW_LDS(F, FWD, MODE)                                                                                  \
  if ((e = hipFuncSetAttribute(reinterpret_cast<const void*>(train_tile_kernel<F, FWD, MODE>),                         \
                               hipFuncAttributeMaxDynamicSharedMemorySize, (int)train_tile_lds_bytes(FWD, MODE))) != hipSuccess) \
    return e;
  PRV_ALLOW_LDS(4, true, 0)
  PRV_ALLOW_LDS(4, false, 0)
  PRV_ALLOW_LDS(4, false, 1)
  PRV_ALLOW_LDS(4, false, 2)
  PRV_ALLOW_LDS(2, true, 0)
  PRV_ALLOW_LDS(2, false, 0)
  PRV_ALLOW_LDS(2, false, 1)
  PRV_ALLOW_LDS(2, false, 2)
#undef PRV_ALLOW_LDS
  return hipSuccess;
}

hipError_t launch_train_tiles(const TrainTileParams& P0, bool forward, int n_blocks, hipStream_t s, bool finish_reduce, int* n_slots_out) {
  TrainTileParams P = P0;
  P.tile_begin = 0;
  P.tile_limit = ~0u;
  P.slot_base = 0;
  const bool f4 = P.n_features == 4;
  if (forward) return f4 ? launch_tile<4, true>(P, n_blocks, s) : launch_tile<2, true>(P, n_blocks, s);
  hipError_t e;
  const int n_lds74 = n_blocks;
  int n_slots = n_blocks;
  const uint32_t kept_tiles = P.act ? P.act_cap / 32u : 0u;
  if (kept_tiles) {
    // the tiles whose activations the forward pass kept, then (a second, nearly always empty launch on slots of its own)
    // the tiles beyond the buffer, which recompute their forward pass
    P.tile_limit = kept_tiles;
    const int n_first = n_blocks, n_tail = n_blocks / 2;
    if (P.bwd_frags) e = f4 ? launch_tile<4, false, 2>(P, n_first, s) : launch_tile<2, false, 2>(P, n_first, s);
    else e = f4 ? launch_tile<4, false, 1>(P, n_first, s) : launch_tile<2, false, 1>(P, n_first, s);
    if (e != hipSuccess) return e;
    n_slots = n_first;
    if (P.sample_cap == 0u || (size_t)kept_tiles * 32u < (size_t)P.sample_cap) {
      // (a buffer that covers every sample a step can list needs no second launch: at the planner loop's 4096-ray cap it
      // always does, and the launch of idle blocks cost 6 us of the step and 20 MB of zeroed slots for the reduction to read)
      P.tile_begin = kept_tiles;
      P.tile_limit = ~0u;
      P.slot_base = n_first; // every slot below n_first + n_tail is written by exactly one block of the two launches
      e = f4 ? launch_tile<4, false>(P, n_tail, s) : launch_tile<2, false>(P, n_tail, s);
      n_slots = n_first + n_tail;
    }
  } else {
    e = f4 ? launch_tile<4, false>(P, n_lds74, s) : launch_tile<2, false>(P, n_lds74, s);
    n_slots = n_lds74;
  }
  if (e != hipSuccess) return e;
  // the stage buffer sits behind the slots (train_dw_slots(n_blocks) slots + kDwGroups group sums)
  float* stage = P.mlp_grad_partial + (size_t)train_dw_slots(n_blocks) * PRV_MLP_HALFS;
  if (n_slots_out) { // the caller's next launch (the table's Adam pass) carries the first stage of the reduction
    *n_slots_out = n_slots;
    return hipGetLastError();
  }
  hipLaunchKernelGGL(train_reduce_dw_kernel, dim3(kDwBlocksX, kDwGroups), dim3(256), 0, s, P.mlp_grad_partial, n_slots, stage);
  if (finish_reduce) hipLaunchKernelGGL(train_reduce_dw2_kernel, dim3((PRV_MLP_HALFS + 255) / 256), dim3(256), 0, s, stage, P.mlp_grad);
  return hipGetLastError();
}

hipError_t launch_prepack_frags(const uint16_t* mlp, int n_features, uint16_t* frags, const AdamParams* end_of_step,
                                uint32_t* sample_count, float lr, hipStream_t s) {
  AdamParams P{};
  if (end_of_step) P = *end_of_step; // state != NULL: this launch closes the step and opens the next
  hipLaunchKernelGGL(prepack_frags_kernel, dim3(((kNumFrags + kBwdFrags) * kFragHalfs + 255) / 256), dim3(256), 0, s, mlp, n_features, frags, P,
                     sample_count, lr);
  return hipGetLastError();
}

hipError_t launch_train_forward_fast(const TrainTileParams& P, const half8* frags, int n_blocks, hipStream_t s) {
  if (P.n_features == 4) hipLaunchKernelGGL(train_forward_fast_kernel<4>, dim3(n_blocks), dim3(256), 0, s, P, frags);
  else hipLaunchKernelGGL(train_forward_fast_kernel<2>, dim3(n_blocks), dim3(256), 0, s, P, frags);
  return hipGetLastError();
}

hipError_t launch_train_composite(const TrainCompositeParams& P, hipStream_t s) {
  hipLaunchKernelGGL(train_composite_kernel, dim3((P.n_rays + 3) / 4), dim3(256), 0, s, P);
  return hipGetLastError();
}

hipError_t launch_train_begin(TrainState* state, uint32_t* sample_count, float lr, float beta1, float beta2, hipStream_t s) {
  hipLaunchKernelGGL(train_begin_kernel, dim3(1), dim3(1), 0, s, state, sample_count, lr, beta1, beta2);
  return hipGetLastError();
}

hipError_t launch_train_loss_finish(const double* loss_part, int n_rays, TrainState* state, unsigned long long* used, hipStream_t s) {
  hipLaunchKernelGGL(train_loss_finish_kernel, dim3(1), dim3(1), 0, s, loss_part, n_rays, state, used);
  return hipGetLastError();
}

// the fixed-point table gradient of a deterministic trainer as f32 (prv_train_gradients' copy-out), cleared
__global__ __launch_bounds__(256) void grad_q_to_f32_kernel(long long* __restrict__ q, size_t n, float* __restrict__ out) {
  const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  out[i] = (float)((double)q[i] * (1.0 / (double)(1ull << kGradQBits)));
  q[i] = 0ll;
}
hipError_t launch_grad_q_to_f32(long long* q, size_t n, float* out, hipStream_t s) {
  hipLaunchKernelGGL(grad_q_to_f32_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, q, n, out);
  return hipGetLastError();
}

hipError_t launch_adam_table(const AdamParams& P, size_t n, float* grad, float* wmv, uint16_t* w16, hipStream_t s, const float* dw_partial,
                             int dw_slots, float* dw_stage, const TrainRaysParams* next_rays, long long* grad_q) {
  const unsigned n_adam = (unsigned)((n + 1023) / 1024), n_dw = dw_partial ? (unsigned)(kDwBlocksX * kDwGroups) : 0u;
  TrainRaysParams R{};
  unsigned n_rays_blocks = 0u;
  if (next_rays) {
    R = *next_rays;
    n_rays_blocks = (unsigned)((R.n_rays + 3) / 4);
  }
  hipLaunchKernelGGL(adam_table_kernel, dim3(n_adam + n_dw + n_rays_blocks), dim3(256), 0, s, P, n, grad, wmv, w16, n_adam, dw_partial, dw_slots,
                     dw_stage, n_dw, R, grad_q);
  return hipGetLastError();
}

hipError_t launch_widen_table(const uint16_t* in, size_t n, float* wmv, hipStream_t s) {
  hipLaunchKernelGGL(widen_table_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, in, n, wmv);
  return hipGetLastError();
}

hipError_t launch_narrow_table(const float* wmv, size_t n, float* out, hipStream_t s) {
  hipLaunchKernelGGL(narrow_table_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, wmv, n, out);
  return hipGetLastError();
}

hipError_t launch_adam_mlp(const AdamParams& P, float l2_reg, float* grad, float* w, float* m, float* v, uint16_t* w16,
                           float* w16_as_f32, const float* stage, int end_of_step, hipStream_t s, uint16_t* frags, const int* frag_pos,
                           uint32_t* sample_count, float lr) {
  hipLaunchKernelGGL(adam_mlp_kernel, dim3((PRV_MLP_HALFS + 255) / 256), dim3(256), 0, s, P, l2_reg, grad, w, m, v, w16,
                     w16_as_f32, stage, end_of_step, frags, frag_pos, sample_count, lr);
  return hipGetLastError();
}

hipError_t launch_frag_positions(int n_features, int* frag_pos, hipStream_t s) {
  hipLaunchKernelGGL(frag_positions_kernel, dim3(((kNumFrags + kBwdFrags) * kFragHalfs + 255) / 256), dim3(256), 0, s, n_features, frag_pos);
  return hipGetLastError();
}

hipError_t launch_widen(const uint16_t* in, size_t n, float* out, hipStream_t s) {
  hipLaunchKernelGGL(widen_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, in, n, out);
  return hipGetLastError();
}

hipError_t launch_density_refresh_fast(const DensityParams& P, int n_features, const half8* frags, int n_blocks, hipStream_t s) {
  if (n_features == 4) hipLaunchKernelGGL(density_refresh_fast_kernel<4>, dim3(n_blocks), dim3(256), 0, s, P, frags);
  else hipLaunchKernelGGL(density_refresh_fast_kernel<2>, dim3(n_blocks), dim3(256), 0, s, P, frags);
  return hipGetLastError();
}

hipError_t launch_density_refresh(const DensityParams& P, int n_features, hipStream_t s) {
  const uint32_t n = (uint32_t)P.occ_res * P.occ_res * P.occ_res;
  if (n_features == 4)
    hipLaunchKernelGGL(density_refresh_kernel<4>, dim3((n + 255) / 256), dim3(256), 0, s, P);
  else
    hipLaunchKernelGGL(density_refresh_kernel<2>, dim3((n + 255) / 256), dim3(256), 0, s, P);
  return hipGetLastError();
}

} // namespace prv
