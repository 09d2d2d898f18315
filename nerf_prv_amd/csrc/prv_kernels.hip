// prv_kernels.hip -- hand-written gfx950 kernels of the render + view-scoring path.
//
//  K_A  march_compact : one lane per primary ray.  Ray generation from the engine-frame
//       camera (replaces pyngp's ray init behind run.py:296,304), unit-cube slab test, S
//       occupancy tests -> 128-bit live-sample mask.  Rays with a non-empty mask are
//       compacted into the ray queue: wave ballot + popcount prefix, one atomic per wave.
//  K_B  render_queue64: persistent waves, 64 ray slots per wave: one lane = one ray = one whole sample per round
//       (all hash-grid levels gathered by the lane itself); v_permlane32_swap turns the wave's 64 feature vectors
//       into the MFMA B operands of two 32-sample column groups.  The two tiny MLPs run on MFMA with the samples on
//       the column, weights staged once per block in LDS as prepacked A fragments, activations never leaving
//       registers.  Front-to-back compositing is sequential per ray inside one lane (deterministic).  A group of 32
//       slots refills as a whole cohort from the queue; thinned-out tails merge / pool (see the kernel).
//  K_S* score kernels : per-view reductions in fp64, fixed reduction order (no float
//       atomics), so rankings are reproducible bit for bit.
//
// Built with -ffp-contract=off: see prv_device.hpp for the arithmetic contract.
#include "prv_kernels.hpp"

#include <algorithm>

#ifndef PRV_ABLATE
#define PRV_ABLATE 0 // dev-only timing ablations: 1 no gather, 2 no MLP, 4 no compositing math (wrong pixels!), 8 march without its rejection test, 16 march: ray set-up but no occupancy walk (every ray dead), 32 march: no set-up either (every ray rejected), 64 march: no dead-pixel writes, 128 march: no mask-extension writes, 256 march: no record copy-out
#endif

namespace prv {

// ------------------------------------------------------------------ K_A march + compact

__device__ __forceinline__ void morton16(uint32_t i, uint32_t& x, uint32_t& y) {
  x = (i & 1u) | ((i >> 1) & 2u) | ((i >> 2) & 4u) | ((i >> 3) & 8u);
  y = ((i >> 1) & 1u) | ((i >> 2) & 2u) | ((i >> 3) & 4u) | ((i >> 4) & 8u);
}

// instant-ngp's stepping rule (PRV_STEP_NGP; SURVEY App. E, what run.py:245-247, 304 renders with): fixed step
// dt = sqrt(3)/1024 from the AABB entry, sample i at t0 + (i + 1/2) dt while inside the box, at most 1024 (the diagonal)
constexpr float kNgpDt = 1.7320508075688772f / 1024.0f; // = sqrtf(3.0f) / 1024.0f, the oracle's value bit for bit

// NGP march of one ray: up to 32 mask words (1024 steps) into the lane's column of the block's LDS scratch `mw`
// ([word][thread]; only words [k_lo, k_hi) are written, the others are empty).  One DILATED coarse lookup clears a whole
// 32-step word: its samples lie within 16 dt = 0.027 of the word's middle, less than one coarse cell (4 / occ_res) for
// occ_res <= 128, so a clear dilated bit there means no sample of the word is in an occupied fine cell.  Masks are
// bit-identical to testing every step.  Returns the number of live samples; first_nz / last_nz = first / last non-empty word.
__device__ __forceinline__ uint32_t march_ngp(const FieldDev& fd, const float o[3], const float d[3], float t0, float t1,
                                              uint32_t (*mw)[256], int& k_lo, int& k_hi, int& first_nz, int& last_nz) {
  const float dt = kNgpDt;
  first_nz = 32;
  last_nz = -1;
  k_lo = k_hi = 0;
  // only samples inside the (one-cell-grown) bounding box of the occupied cells can be live
  float ta = t0, tb = t1;
#pragma unroll
  for (int a = 0; a < 3; a++) {
    const float inv = 1.0f / d[a];
    const float u = (fd.occ_lo[a] - o[a]) * inv, w = (fd.occ_hi[a] - o[a]) * inv;
    ta = fmaxf(ta, fminf(u, w));
    tb = fminf(tb, fmaxf(u, w));
  }
  if (!(tb > ta)) return 0u;
  const float inv_dt = 1024.0f / 1.7320508075688772f;
  const int g_lo = max(0, (int)((ta - t0) * inv_dt) - 2);
  const int g_hi = min(kNgpMaxSteps, (int)((tb - t0) * inv_dt) + 3); // the exact `t < t1` test below ends the ray
  if (g_lo >= g_hi) return 0u;
  k_lo = g_lo >> 5;
  k_hi = (g_hi + 31) >> 5;
  const bool coarse_ok = fd.occ_coarse != nullptr && 16.0f * dt <= 0.99f * 4.0f / (float)fd.occ_res;
  const int Rc_m1 = (fd.occ_res >> 2) - 1;
  const float fRc = (float)(fd.occ_res >> 2);
  const uint32_t Rc = (uint32_t)(Rc_m1 + 1);
  uint32_t n_live = 0u;
  for (int k = k_lo; k < k_hi; k++) {
    uint32_t w = 0u;
    bool pass = true;
    if (coarse_ok) {
      const float tm = fmaf((float)(32 * k + 16), dt, t0);
      const int cx = min(max((int)(fmaf(tm, d[0], o[0]) * fRc), 0), Rc_m1), cy = min(max((int)(fmaf(tm, d[1], o[1]) * fRc), 0), Rc_m1),
                cz = min(max((int)(fmaf(tm, d[2], o[2]) * fRc), 0), Rc_m1);
      const uint32_t bit = (uint32_t)cx + Rc * ((uint32_t)cy + Rc * (uint32_t)cz);
      pass = (fd.occ_coarse[bit >> 5] >> (bit & 31u)) & 1u;
    }
    if (pass) {
#pragma unroll
      for (int q0 = 0; q0 < 32; q0 += 8) { // eight fine tests issued together (eight loads in flight, one wait)
        bool occ[8];
#pragma unroll
        for (int q = 0; q < 8; q++) {
          const int i = 32 * k + q0 + q;
          const float t = fmaf((float)i + 0.5f, dt, t0);
          occ[q] = occupied(fd, fmaf(t, d[0], o[0]), fmaf(t, d[1], o[1]), fmaf(t, d[2], o[2])) && t < t1;
        }
#pragma unroll
        for (int q = 0; q < 8; q++) w |= (uint32_t)occ[q] << (q0 + q);
      }
    }
    mw[k][threadIdx.x] = w;
    if (w != 0u) {
      if (first_nz == 32) first_nz = k;
      last_nz = k;
      n_live += (uint32_t)__popc(w);
    }
  }
  return n_live;
}

// ---- the three parts of a march block: ray set-up, (the occupancy walk, per kernel), emit

struct MarchRay {
  float o[3], d[3], t0, t1;
  uint32_t pix;
  bool valid, hit; // inside the image; the ray enters the unit cube (and passed the rejection tests)
};

// pixel / sub-sample of this thread in tile (tx, ty) of view blockIdx.y, the rejection tests against the occupied box
// [occ_lo, occ_hi], the exact ray, the unit-cube slab test.  PT: MarchParams or MarchMultiParams (same field names).
template <class PT>
__device__ __forceinline__ void march_ray_setup(const PT& P, const float occ_lo[3], const float occ_hi[3], uint32_t tx, uint32_t ty, MarchRay& r) {
  const uint32_t vi = blockIdx.y;
  // Multi-sample launches: with a power-of-two spp the sub-samples of a pixel sit on ADJACENT lanes
  // (a wave = 64/spp pixels x spp sub-samples: rays that share nearly every grid cell); otherwise the
  // sub-sample index is on grid.z.
  uint32_t ix, iy, kk;
  if (P.spp_inner_log2 > 0) {
    kk = threadIdx.x & ((1u << P.spp_inner_log2) - 1u);
    morton16(threadIdx.x >> P.spp_inner_log2, ix, iy);
  } else {
    kk = blockIdx.z;
    morton16(threadIdx.x, ix, iy);
  }
  const int spp_k = P.spp_k + (int)kk;
  const int px = (int)((tx << P.tile_w_log2) + ix), py = (int)((ty << P.tile_h_log2) + iy);
  r.valid = px < P.W && py < P.H;
  r.pix = ((kk * gridDim.y + (uint32_t)vi) * (uint32_t)P.H + (uint32_t)py) * (uint32_t)P.W + (uint32_t)px;
  r.o[0] = r.o[1] = r.o[2] = 0.f;
  r.d[0] = r.d[1] = 0.f;
  r.d[2] = 1.f;
  r.t0 = r.t1 = 0.f;
  r.hit = false;
  bool maybe = r.valid;
  const CamDev& cam = P.cams[P.view_ids[vi]];
  float ox, oy;
  spp_offset(spp_k, ox, oy);
  // whole-tile rejection (block-uniform, before any per-ray work): the tile's pixels against the rectangle outside which no
  // ray of this view can meet the occupied box (CamDev::cull, computed on the host in double per render call)
  if (cam.cull[2] > 0) {
    const int x0 = (int)(tx << P.tile_w_log2), y0 = (int)(ty << P.tile_h_log2);
    if (x0 + (1 << P.tile_w_log2) <= cam.cull[0] || x0 >= cam.cull[2] || y0 + (1 << P.tile_h_log2) <= cam.cull[1] || y0 >= cam.cull[3]) maybe = false;
  }
  if (maybe && !has_lens(cam)) {
    // Cheap rejection before the exact (IEEE divides, normalisation: ~300 instructions) ray set-up: nine rays in ten
    // never come near the object and the pass is VALU bound on that set-up.  A live sample lies in an occupied cell,
    // hence inside [occ_lo, occ_hi]; a slab test of the UNNORMALISED direction, built with reciprocals, against that
    // box grown by `grow` can only err towards "hit": the direction is off by < 1e-6 rad, i.e. < 1e-6 * far at the
    // box, and a ray touching the box crosses the grown one over a chord > 2 * grow.  NaNs (0 * inf) drop a slab.
    const float far = fmaxf(fmaxf(fabsf(cam.c2w[3] - 0.5f), fabsf(cam.c2w[7] - 0.5f)), fabsf(cam.c2w[11] - 0.5f));
    const float grow = fmaf(far, 1e-5f, 1e-3f);
    const float x = (((float)px + ox) - cam.cx) * __builtin_amdgcn_rcpf(cam.fx);
    const float y = (((float)py + oy) - cam.cy) * __builtin_amdgcn_rcpf(cam.fy);
    float ta = 0.0f, tb = __builtin_inff();
#pragma unroll
    for (int a = 0; a < 3; a++) {
      const float* mrow = cam.c2w + a * 4;
      const float inv = __builtin_amdgcn_rcpf(fmaf(mrow[0], x, fmaf(mrow[1], y, mrow[2])));
      const float u = ((occ_lo[a] - grow) - mrow[3]) * inv, w = ((occ_hi[a] + grow) - mrow[3]) * inv;
      ta = fmaxf(ta, fminf(u, w));
      tb = fminf(tb, fmaxf(u, w));
    }
    maybe = !(tb < ta);
  }
#if PRV_ABLATE & 8
  maybe = r.valid; // dev timing only: no rejection test
#endif
  if (maybe) {
    raygen(cam, px, py, ox, oy, r.o, r.d);
    r.hit = ray_aabb(r.o, r.d, r.t0, r.t1);
  }
}

// the queue a march block appends to, and the image its dead rays are written to (one per ensemble member in the
// multi-member launch)
struct MarchSink {
  void* queue;
  uint4* queue_ext;  // chunk-major: chunk j (>= 1) of queue slot s at [(j - 1) * ext_stride + s] -- the lanes of a wave hold
  size_t ext_stride; // consecutive slots, so a chunk's store is one contiguous block (slot-major: 64 lines per instruction)
  uint32_t* queue_count;
  float* out_f32;
  uint32_t* out_u8;
};

// Wave-level compaction of the live rays into the queue + the record + the statistics + the dead ray's pixel.
//   m[4]: the record's own 128-step chunk (FIXED_S: the whole mask; NGP: filled here from word(first_nz ...));
//   word(k): NGP, mask word k of this ray (0 outside what the walk wrote)
template <bool NGP, bool WORDS_READY = false, class PT, class WordFn>
__device__ __forceinline__ void march_write(const PT& P, const MarchSink& Q, uint4 (*stage)[64 * kRecordWords], const MarchRay& r, bool live,
                                            unsigned long long b, uint32_t base, float dt, uint32_t m[4], int first_nz, int last_nz, WordFn word) {
  // b = the wave's ballot of live rays, base = where its records start in the queue (march_reserve)
  const int lane = threadIdx.x & 63;
  if (b != 0ull) {
    // the wave's live records are consecutive in the queue: they are staged in LDS and leave as one contiguous block,
    // consecutive lanes writing consecutive 16-byte words (a lane storing its own 96-byte record word by word makes
    // every store instruction touch 64 different lines, six times over)
    uint4* st = stage[threadIdx.x >> 6];
    if (live) {
      const uint32_t prefix = (uint32_t)__popcll(b & ((1ull << lane) - 1ull));
      uint4* rec = st + prefix * kRecordWords;
      uint32_t chunk_info = 1u << 16; // {first step of chunk 0 (a multiple of 32), number of 128-step mask chunks << 16}
      if constexpr (NGP) {
        // the record carries the 128 steps from the first non-empty word on; the chunks behind it (up to 7 more) go to
        // the record's slot of the extension buffer, which the render kernel reads only when a ray gets that far
        const int n_chunks = ((last_nz - first_nz) >> 2) + 1;
        chunk_info = (uint32_t)(32 * first_nz) | ((uint32_t)n_chunks << 16);
        if constexpr (!WORDS_READY) { // (the multi-member kernel has filled m and written the later chunks itself)
#pragma unroll
          for (int q = 0; q < 4; q++) m[q] = word(first_nz + q);
          uint4* ext = Q.queue_ext + (size_t)(base + prefix);
          for (int j = 1; j < ((PRV_ABLATE & 128) ? 1 : n_chunks); j++)
            ext[(size_t)(j - 1) * Q.ext_stride] = make_uint4(word(first_nz + 4 * j), word(first_nz + 4 * j + 1), word(first_nz + 4 * j + 2), word(first_nz + 4 * j + 3));
        }
      }
      rec[0] = make_uint4(__float_as_uint(r.o[0]), __float_as_uint(r.o[1]), __float_as_uint(r.o[2]), __float_as_uint(r.t0));
      rec[1] = make_uint4(__float_as_uint(r.d[0]), __float_as_uint(r.d[1]), __float_as_uint(r.d[2]), __float_as_uint(dt));
      rec[2] = make_uint4(m[0], m[1], m[2], m[3]);
      rec[3] = make_uint4(r.pix, chunk_info, 0u, 0u);
      // direction encoding once per ray, here, so a slot refill in K_B is loads only
      reinterpret_cast<half8*>(rec)[4] = sh_fragment(0, r.d[0], r.d[1], r.d[2]);
      reinterpret_cast<half8*>(rec)[5] = sh_fragment(1, r.d[0], r.d[1], r.d[2]);
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    uint4* dst = reinterpret_cast<uint4*>(Q.queue) + (size_t)base * kRecordWords;
    const uint32_t n_words = (uint32_t)__popcll(b) * kRecordWords;
    for (uint32_t i = (uint32_t)lane; i < ((PRV_ABLATE & 256) ? 1u : n_words); i += 64u) dst[i] = st[i];
  }
  if (!live && r.valid && !(PRV_ABLATE & 64)) {
    // dead ray: contributes exactly zero to its pixel
    float4* out = reinterpret_cast<float4*>(Q.out_f32) + r.pix;
    float4 v = P.spp_k == 0 ? make_float4(0.f, 0.f, 0.f, 0.f) : *out;
    if (P.last_pass) {
      v.x *= P.inv_spp; v.y *= P.inv_spp; v.z *= P.inv_spp; v.w *= P.inv_spp;
      if (Q.out_u8) Q.out_u8[r.pix] = quantize_rgba8(v.x, v.y, v.z, v.w, P.bg);
    }
    if (P.spp_k == 0 || P.last_pass) *out = v;
  }
}

// Which of the queue's regions a wave's live rays go to (round 6).  A region is drained by ONE XCD (render_queue64: seg = XCC id)
// and every XCD has an L2 of its own: until round 5 the region was the block's index modulo the region count, every region a thin
// slice of EVERY part of the scene, so every L2 held a share of the whole table.  Now it is the OCTANT, about the occupied box's
// centre, of the MIDDLE OF THE LIVE SPAN of the wave's first live ray (a wave is a patch of adjacent pixels / sub-samples: its rays
// cross the object together; the first live sample as the key measured 2-6 % slower): an XCD renders the rays that cross one part
// of the field, and its L2 holds that part.  Pixels do not
// depend on which slot renders a ray (tests/test_gpu_sweep.py).  A region holds 1 / n_seg of the batch's rays (+ 64 slots); a wave
// whose region is full moves on to the next one -- with that slack some region always has room for a wave's <= 64 records.
__device__ __forceinline__ uint32_t ray_octant(const float o[3], const float d[3], float t, const float lo[3], const float hi[3]) {
  uint32_t oct = 0u;
#pragma unroll
  for (int a = 0; a < 3; a++) oct |= (uint32_t)(fmaf(t, d[a], o[a]) > 0.5f * (lo[a] + hi[a])) << a;
  return oct;
}
// called by ONE lane: n records in region `pref` or the first region after it with room; returns the first record's queue slot.
// ONE returning add in the common case (a compare-and-swap loop here made the march pass five times as long: the waves running
// at any one time are neighbours, i.e. all on the same few counters).  The one reservation that straddles a region's end keeps
// its slots empty and says so in the word behind the counter (tail cut: the render kernel drains count - cut records); every
// later add on that counter lands beyond the end and moves on as well.
__device__ __forceinline__ uint32_t region_reserve(uint32_t* queue_count, uint32_t n_seg, uint32_t seg_cap, uint32_t pref, uint32_t n) {
  for (uint32_t k = 0; k < n_seg; k++) {
    const uint32_t shard = (pref + k) % n_seg;
    uint32_t* c = queue_count + 16u * shard;
    if (k != 0u && __hip_atomic_load(c, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) >= seg_cap) continue; // (a full region: no add at all)
    const uint32_t old = atomicAdd(c, n);
    if (old + n <= seg_cap) return shard * seg_cap + old;
    if (old < seg_cap) c[1] = seg_cap - old; // this reservation straddles the end: [old, seg_cap) stays unwritten
  }
  return 0u; // unreachable: the regions together hold every ray of the batch plus 64 slots each (prv_api.cpp: seg_cap)
}

// the statistics: the march count (live samples before any early termination), one atomic per wave, on a counter sharded
// like the queue's (~10^5 atomics on ONE word cost 0.3 ms of a 0.65 ms launch)
template <class PT>
__device__ __forceinline__ void march_count(const PT& P, uint32_t n_live) {
  uint32_t tot = n_live;
#pragma unroll
  for (int sh = 32; sh >= 1; sh >>= 1) tot += (uint32_t)__shfl_xor((int)tot, sh);
  const uint32_t sshard = (blockIdx.x + gridDim.x * (blockIdx.y + gridDim.y * blockIdx.z)) & 7u;
  if ((threadIdx.x & 63) == 0 && tot) atomicAdd(P.stat + 8u * (1u + sshard), (unsigned long long)tot); // 64 bytes apart: stat[8], stat[16], ...
}

// Wave-level compaction of the live rays into the queue + the record + the statistics + the dead ray's pixel.
//   m[4]: the record's own 128-step chunk (FIXED_S: the whole mask; NGP: filled here from word(first_nz ...));
//   word(k): NGP, mask word k of this ray (0 outside what the walk wrote)
template <bool NGP, class PT, class WordFn>
__device__ __forceinline__ void march_emit(const PT& P, const MarchSink& Q, uint4 (*stage)[64 * kRecordWords], const MarchRay& r, bool live,
                                           uint32_t n_live, float dt, uint32_t m[4], int first_nz, int last_nz, WordFn word) {
  // wave-level compaction: ballot + prefix popcount, one atomic per wave.  The queue is cut into n_seg regions of
  // seg_cap records and the counter is sharded with it (a block appends to region `linear block id % n_seg`): one
  // returning atomic word saturates near 90 per microsecond on this chip and ~10^5 waves append per launch -- with a
  // single counter that alone was 0.15 ms of the 0.84 ms pass.  The render kernel drains region by region.
  const unsigned long long b = __ballot(live);
  const int lane = threadIdx.x & 63;
  uint32_t base = 0;
  if (b != 0ull) {
    if (lane == (int)__builtin_ctzll(b)) { // the wave's first live ray: the middle of its live span names the region (ray_octant)
      uint32_t i0 = 0u;
      if constexpr (NGP) {
        i0 = 32u * (uint32_t)first_nz + (uint32_t)__builtin_ctz(word(first_nz));
        if (P.spatial_regions != 2) i0 = (i0 + 32u * (uint32_t)last_nz + 31u - (uint32_t)__builtin_clz(word(last_nz))) >> 1; // the middle of the live span (2, dev: its first sample)
      } else {
        i0 = m[0] ? (uint32_t)__builtin_ctz(m[0]) : m[1] ? 32u + (uint32_t)__builtin_ctz(m[1]) : m[2] ? 64u + (uint32_t)__builtin_ctz(m[2]) : 96u + (uint32_t)__builtin_ctz(m[3]);
        if (P.spatial_regions != 2) {
          const uint32_t i1 = m[3] ? 127u - (uint32_t)__builtin_clz(m[3]) : m[2] ? 95u - (uint32_t)__builtin_clz(m[2]) : m[1] ? 63u - (uint32_t)__builtin_clz(m[1]) : 31u - (uint32_t)__builtin_clz(m[0]);
          i0 = (i0 + i1) >> 1;
        }
      }
      const uint32_t lin = blockIdx.x + gridDim.x * (blockIdx.y + gridDim.y * blockIdx.z);
      const uint32_t pref = P.spatial_regions ? ray_octant(r.o, r.d, fmaf((float)i0 + 0.5f, dt, r.t0), P.field.occ_lo, P.field.occ_hi) * (uint32_t)P.n_sub + lin % (uint32_t)P.n_sub
                                              : lin;
      base = region_reserve(Q.queue_count, (uint32_t)P.n_seg, P.seg_cap, pref % (uint32_t)P.n_seg, (uint32_t)__popcll(b));
    }
    base = __shfl(base, (int)__builtin_ctzll(b));
  }
  march_write<NGP>(P, Q, stage, r, live, b, base, dt, m, first_nz, last_nz, word);
  if (b != 0ull) march_count(P, n_live);
}

template <bool NGP>
__global__ __launch_bounds__(256) void march_compact_kernel(MarchParams P) {
  __shared__ uint4 stage[4][64 * kRecordWords]; // a wave's live records, written out as ONE contiguous block
  __shared__ uint32_t mw[NGP ? 32 : 1][256];    // NGP: the lanes' mask words ([word][thread]: conflict-free columns)
  const uint32_t vi = blockIdx.y;
  uint32_t tx, ty;
  if (P.live_grid) {
    // only the tiles inside the view's cull rectangle are launched (the caller consumes the image through the same
    // rectangles and never reads a pixel outside them: prv_score_views, method 5): blockIdx.x counts the tiles of THIS
    // view's rectangle, the grid is as large as the largest rectangle of the batch
    const CamDev& cv = P.cams[P.view_ids[vi]];
    uint32_t tx0 = 0, ty0 = 0, tx1 = P.tiles_x, ty1 = P.tiles_y;
    if (cv.cull[2] > 0) {
      tx0 = (uint32_t)cv.cull[0] >> P.tile_w_log2;
      ty0 = (uint32_t)cv.cull[1] >> P.tile_h_log2;
      tx1 = min(P.tiles_x, ((uint32_t)cv.cull[2] + (1u << P.tile_w_log2) - 1u) >> P.tile_w_log2);
      ty1 = min(P.tiles_y, ((uint32_t)max(cv.cull[3], 0) + (1u << P.tile_h_log2) - 1u) >> P.tile_h_log2);
    }
    const uint32_t w = tx1 > tx0 ? tx1 - tx0 : 0u, h = ty1 > ty0 ? ty1 - ty0 : 0u;
    if (blockIdx.x >= w * h) return;
    ty = ty0 + blockIdx.x / w;
    tx = tx0 + blockIdx.x - (blockIdx.x / w) * w;
  } else {
    ty = blockIdx.x / P.tiles_x;
    tx = blockIdx.x - ty * P.tiles_x;
  }
  MarchRay r;
  march_ray_setup(P, P.field.occ_lo, P.field.occ_hi, tx, ty, r);
  const float* o = r.o;
  const float* d = r.d;
  const float t0 = r.t0, t1 = r.t1;
  float dt = 0.f;
  uint32_t m[4] = {0, 0, 0, 0};
  uint32_t n_live = 0u;                              // live samples of this ray (statistics: the march count)
  int k_lo = 0, k_hi = 0, first_nz = 32, last_nz = -1; // NGP: written words of mw, first / last non-empty word
  bool live = false;
#if PRV_ABLATE & 32
  r.hit = false;
#endif
  if (r.hit) {
#if PRV_ABLATE & 16
    live = (t0 + t1 + d[0] + o[1]) == 1.0e30f; // never true, but the compiler cannot know: the set-up stays
#else
    if constexpr (NGP) {
      dt = kNgpDt;
      n_live = march_ngp(P.field, o, d, t0, t1, mw, k_lo, k_hi, first_nz, last_nz);
      live = first_nz < 32;
    } else {
      dt = (t1 - t0) / (float)P.S;
      // Two-level test, bit-identical to testing every sample: samples g..g+3 all lie within 1.5*dt of
      // the point at parameter g+2; when that is less than one coarse cell and the DILATED coarse bit
      // there is clear, none of the four can be in an occupied fine cell.
      const bool coarse_ok = P.field.occ_coarse != nullptr && 1.5f * dt <= 0.99f * 4.0f / (float)P.field.occ_res;
      // only samples inside the (one-cell-grown) bounding box of the occupied cells can be live: clip the
      // loop to the box's parameter range, with two samples of slack either side and group alignment
      int g_lo = 0, g_hi = P.S;
      {
        float ta = t0, tb = t1;
#pragma unroll
        for (int a = 0; a < 3; a++) {
          const float inv = 1.0f / d[a];
          const float u = (P.field.occ_lo[a] - o[a]) * inv, w = (P.field.occ_hi[a] - o[a]) * inv;
          ta = fmaxf(ta, fminf(u, w));
          tb = fminf(tb, fmaxf(u, w));
        }
        if (!(tb > ta)) {
          g_hi = 0; // misses the occupied region altogether
        } else if (dt > 0.0f) {
          const float inv_dt = __builtin_amdgcn_rcpf(dt); // bounds only (two samples of slack either side): 1 ulp is plenty
          g_lo = max(0, (int)((ta - t0) * inv_dt) - 2) & ~3;
          g_hi = min(P.S, (int)((tb - t0) * inv_dt) + 3);
        }
      }
      // one mask word (32 samples = 8 groups) at a time, so the word is a plain register and a sample's bit goes in
      // with one shift-or instead of a select chain over m[0..3]; the four fine tests of a group are issued
      // together (four loads in flight, one wait)
      // the coarse test in coarse-cell units: p * Rc = fma(tm, d * Rc, o * Rc).  Rounding differs from occupied_coarse()'s
      // by ~1e-6 of a cell, the dilated grid covers a whole cell and the 0.99 above leaves 1 % of one: the skip stays
      // conservative, the masks bit-identical.
      const int Rc_m1 = (P.field.occ_res >> 2) - 1;
      const float fRc = (float)(P.field.occ_res >> 2);
      const float ocx = o[0] * fRc, ocy = o[1] * fRc, ocz = o[2] * fRc, dcx = d[0] * fRc, dcy = d[1] * fRc, dcz = d[2] * fRc;
      const uint32_t Rc = (uint32_t)(Rc_m1 + 1);
#pragma unroll
      for (int k = 0; k < 4; k++) {
        uint32_t w = 0u;
        const int lo = max(g_lo, 32 * k), hi = min(g_hi, 32 * k + 32);
        if (lo < hi) {
          // the eight coarse tests of this mask word are issued TOGETHER (eight independent loads in flight, one wait): a
          // test-and-wait per group made the ray's 32 groups 32 dependent memory round trips -- the pass was latency bound
          uint32_t pass = 0xffu;
          if (coarse_ok) {
            pass = 0u;
#pragma unroll
            for (int j = 0; j < 8; j++) {
              const int g = 32 * k + 4 * j;
              const float tm = fmaf((float)g + 2.0f, dt, t0);
              const int cx = min(max((int)fmaf(tm, dcx, ocx), 0), Rc_m1), cy = min(max((int)fmaf(tm, dcy, ocy), 0), Rc_m1),
                        cz = min(max((int)fmaf(tm, dcz, ocz), 0), Rc_m1);
              const uint32_t bit = (uint32_t)cx + Rc * ((uint32_t)cy + Rc * (uint32_t)cz);
              pass |= ((P.field.occ_coarse[bit >> 5] >> (bit & 31u)) & 1u) << j;
            }
          }
          // groups inside [lo, hi) only (lo is a multiple of 4)
          pass &= (0xffu << ((lo - 32 * k) >> 2)) & (0xffu >> (7 - ((hi - 1 - 32 * k) >> 2)));
          while (pass) {
            const int j = __builtin_ctz(pass);
            pass &= pass - 1u;
            const int g = 32 * k + 4 * j;
            bool occ[4];
#pragma unroll
            for (int q = 0; q < 4; q++) {
              const int i = g + q;
              const float t = fmaf((float)i + 0.5f, dt, t0);
              occ[q] = occupied(P.field, fmaf(t, d[0], o[0]), fmaf(t, d[1], o[1]), fmaf(t, d[2], o[2])) && i < P.S;
            }
#pragma unroll
            for (int q = 0; q < 4; q++) w |= (uint32_t)occ[q] << ((g + q) & 31);
          }
        }
        m[k] = w;
      }
      live = (m[0] | m[1] | m[2] | m[3]) != 0u;
      n_live = (uint32_t)(__popc(m[0]) + __popc(m[1]) + __popc(m[2]) + __popc(m[3]));
    }
#endif
  }
  const MarchSink Q{P.queue, P.queue_ext, (size_t)P.seg_cap * (size_t)P.n_seg, P.queue_count, P.out_f32, P.out_u8};
  march_emit<NGP>(P, Q, stage, r, live, n_live, dt, m, first_nz, last_nz,
                  [&](int k) { return (k >= k_lo && k < k_hi) ? mw[k][threadIdx.x] : 0u; });
}

// ---- the ensemble's march in ONE launch (PRV_STEP_NGP; the reference trains E members and renders every candidate with
// each of them, main.cpp:2041-2043, 2101-2103; run.py:296-304 -- E march launches until round 4).  The rays of the E
// members are the same rays; what differs is the occupancy grid each walks.  One thread = one ray for ALL members: the
// set-up once, and above all ONE walk -- a step's position, cell and address are computed once (15 of the ~20
// instructions of a fine test), and one byte of the members' interleaved occupancy (bit e = member e, built per round by
// occ_interleave_kernel) answers the step for every member.  The clip range is the union box's, a word is walked when ANY
// member's dilated coarse bit is set: a member's mask bits are its own fine bits wherever they are tested and its grid is
// empty wherever its own launch would not have looked, so every member's masks, records and march count are what its own
// launch writes (the queue order differs, as between any two launches).
// Two passes over the ray's words (all mask words of all members in LDS would be 160 KB per block): the first walks, finds
// every member's first / last non-empty word and its march count and keeps the first kWin words from each member's first
// non-empty one in LDS; the queue slots are then reserved for all members at once; the second pass puts each word where it
// belongs -- the record's own chunk in registers, the later chunks straight into the member's extension buffer -- reading
// the LDS window and walking again only the words some member needs from behind it (live spans over 32 kWin steps).
template <int E, int kWin>
__global__ __launch_bounds__(256) void march_multi_kernel(MarchMultiParams P) {
  __shared__ uint4 stage[4][64 * kRecordWords];
  __shared__ uint32_t mw[E][kWin][256]; // the first kWin words of every member's mask from its first non-empty one (second walk below)
  const uint32_t ty = blockIdx.x / P.tiles_x, tx = blockIdx.x - ty * P.tiles_x;
  MarchRay r;
  march_ray_setup(P, P.occ_lo, P.occ_hi, tx, ty, r);
  const float* o = r.o;
  const float* d = r.d;
  const float t0 = r.t0, t1 = r.t1, dt = kNgpDt;
  int first_nz[E], last_nz[E];
  uint32_t n_live[E];
#pragma unroll
  for (int e = 0; e < E; e++) {
    first_nz[e] = 32;
    last_nz[e] = -1;
    n_live[e] = 0u;
  }
  const int R = P.occ_res;
  const float fR = (float)R;
  const bool coarse_ok = P.occ_coarse_bytes != nullptr && 16.0f * dt <= 0.99f * 4.0f / (float)R;
  const int Rc_m1 = (R >> 2) - 1;
  const float fRc = (float)(R >> 2);
  const uint32_t Rc = (uint32_t)(Rc_m1 + 1);
  // the 32 steps of word k for every member: the bytes of the steps' cells (0 behind the ray's exit), transposed.  All 32
  // loads are issued together: the walk is a chain of memory round trips, one per word this way
  auto fine_word = [&](int k, uint32_t w[E]) {
#pragma unroll
    for (int e = 0; e < E; e++) w[e] = 0u;
    uint32_t by[32];
#pragma unroll
    for (int q = 0; q < 32; q++) {
      const int i = 32 * k + q;
      const float t = fmaf((float)i + 0.5f, dt, t0);
      const float px = fmaf(t, d[0], o[0]), py = fmaf(t, d[1], o[1]), pz = fmaf(t, d[2], o[2]);
      const int cx = min((int)(clamp01(px) * fR), R - 1), cy = min((int)(clamp01(py) * fR), R - 1), cz = min((int)(clamp01(pz) * fR), R - 1);
      const uint32_t cell = (uint32_t)cx + __umul24((uint32_t)R, (uint32_t)cy + __umul24((uint32_t)R, (uint32_t)cz)); // as occupied()
      // (the load is unconditional -- the cell is always inside the grid -- and masked afterwards: a conditional load
      // compiles to a branch and a wait per step)
      by[q] = (uint32_t)P.occ_bytes[cell] & (0u - (uint32_t)(t < t1));
    }
#pragma unroll
    for (int q = 0; q < 32; q++)
#pragma unroll
      for (int e = 0; e < E; e++) w[e] |= ((by[q] >> e) & 1u) << q;
  };
  uint32_t passmask = 0u; // bit k: some member's dilated coarse bit is set at the middle of word k (no live sample of any member otherwise)
  if (r.hit) {
    // only samples inside the (one-cell-grown) bounding box of the members' occupied cells can be live (march_ngp's clip)
    float ta = t0, tb = t1;
#pragma unroll
    for (int a = 0; a < 3; a++) {
      const float inv = 1.0f / d[a];
      const float u = (P.occ_lo[a] - o[a]) * inv, w = (P.occ_hi[a] - o[a]) * inv;
      ta = fmaxf(ta, fminf(u, w));
      tb = fminf(tb, fmaxf(u, w));
    }
    int k_lo = 0, k_hi = 0;
    if (tb > ta) {
      const float inv_dt = 1024.0f / 1.7320508075688772f;
      const int g_lo = max(0, (int)((ta - t0) * inv_dt) - 2);
      const int g_hi = min(kNgpMaxSteps, (int)((tb - t0) * inv_dt) + 3);
      if (g_lo < g_hi) {
        k_lo = g_lo >> 5;
        k_hi = (g_hi + 31) >> 5;
      }
    }
    for (int kb = k_lo; kb < k_hi; kb += 8) { // the coarse tests of eight words together
      uint32_t pass = 0xffu;
      if (coarse_ok) {
        pass = 0u;
#pragma unroll
        for (int j = 0; j < 8; j++) {
          const int k = min(kb + j, k_hi - 1);
          const float tm = fmaf((float)(32 * k + 16), dt, t0);
          const int cx = min(max((int)(fmaf(tm, d[0], o[0]) * fRc), 0), Rc_m1), cy = min(max((int)(fmaf(tm, d[1], o[1]) * fRc), 0), Rc_m1),
                    cz = min(max((int)(fmaf(tm, d[2], o[2]) * fRc), 0), Rc_m1);
          pass |= (uint32_t)(P.occ_coarse_bytes[(uint32_t)cx + Rc * ((uint32_t)cy + Rc * (uint32_t)cz)] != 0) << j;
        }
      }
      for (int j = 0; j < 8 && kb + j < k_hi; j++) {
        if (!((pass >> j) & 1u)) continue;
        const int k = kb + j;
        passmask |= 1u << k;
        uint32_t w[E];
        fine_word(k, w);
#pragma unroll
        for (int e = 0; e < E; e++) {
          if (w[e] != 0u) {
            if (first_nz[e] == 32) first_nz[e] = k;
            last_nz[e] = k;
            n_live[e] += (uint32_t)__popc(w[e]);
          }
          const int j = k - first_nz[e]; // < 0 before the member's first non-empty word
          if (j >= 0 && j < kWin) mw[e][j][threadIdx.x] = w[e];
        }
      }
    }
  }
  // The E members' queue reservations in ONE wave instruction (lane e adds member e's count to member e's counter): one
  // returning-atomic latency per wave instead of E in a row; the march count of all members in one add.
  const int lane = threadIdx.x & 63;
  const size_t ext_stride = (size_t)P.seg_cap * (size_t)P.n_seg;
  unsigned long long b[E];
  uint32_t base[E];
  {
    // the region (ray_octant): the middle of the live span -- of all members together, in whole words -- of the wave's first ray
    // that is live for any member
    int first_any = 32, last_any = 0;
#pragma unroll
    for (int e = 0; e < E; e++) {
      first_any = min(first_any, first_nz[e]);
      last_any = max(last_any, last_nz[e]);
    }
    const unsigned long long b_any = __ballot(first_any < 32);
    uint32_t pref = blockIdx.x + gridDim.x * (blockIdx.y + gridDim.y * blockIdx.z);
    if (P.spatial_regions && b_any != 0ull) {
      const uint32_t oct = ray_octant(o, d, fmaf((float)(16 * (min(first_any, 31) + last_any)) + 16.0f, dt, t0), P.occ_lo, P.occ_hi);
      pref = (uint32_t)__shfl((int)oct, (int)__builtin_ctzll(b_any)) * (uint32_t)P.n_sub + pref % (uint32_t)P.n_sub;
    }
    const uint32_t shard = pref % (uint32_t)P.n_seg;
    uint32_t* qc = nullptr;
    uint32_t mine = 0u, tot = 0u;
#pragma unroll
    for (int e = 0; e < E; e++) {
      b[e] = __ballot(first_nz[e] < 32);
      if (lane == e) {
        qc = P.mem[e].queue_count;
        mine = (uint32_t)__popcll(b[e]);
      }
      tot += n_live[e];
    }
    uint32_t got = 0u;
    if (mine) got = region_reserve(qc, (uint32_t)P.n_seg, P.seg_cap, shard, mine);
#pragma unroll
    for (int e = 0; e < E; e++) base[e] = __shfl(got, e);
    march_count(P, tot);
  }
  // second walk: word k is word j = k - first_nz[e] of member e's mask -- chunk j / 4: the record's own (0) or one of the
  // extension buffer's, written as soon as its four words (or the member's last) are there
  uint32_t m[E][4];
#pragma unroll
  for (int e = 0; e < E; e++)
#pragma unroll
    for (int q = 0; q < 4; q++) m[e][q] = 0u;
  {
    int k2_lo = 32, k2_hi = -1;
#pragma unroll
    for (int e = 0; e < E; e++) {
      k2_lo = min(k2_lo, first_nz[e]);
      k2_hi = max(k2_hi, last_nz[e]);
    }
    uint32_t acc[E][4];
#pragma unroll
    for (int e = 0; e < E; e++)
#pragma unroll
      for (int q = 0; q < 4; q++) acc[e][q] = 0u;
    for (int k = k2_lo; k <= k2_hi; k++) {
      uint32_t w[E];
      // a word is walked again only where some member needs it from behind its LDS window (a live span of more than
      // kWin words: 32 kWin steps); words no member's coarse bit passed were never written and are zero
      bool behind = false;
#pragma unroll
      for (int e = 0; e < E; e++) behind = behind || (k - first_nz[e] >= kWin && k <= last_nz[e]);
      if (!((passmask >> k) & 1u)) {
#pragma unroll
        for (int e = 0; e < E; e++) w[e] = 0u;
      } else if (behind) {
        fine_word(k, w);
      } else {
#pragma unroll
        for (int e = 0; e < E; e++) {
          const int j = k - first_nz[e];
          w[e] = (j >= 0 && k <= last_nz[e]) ? mw[e][j][threadIdx.x] : 0u;
        }
      }
#pragma unroll
      for (int e = 0; e < E; e++) {
        const int j = k - first_nz[e];
        if (j < 0 || k > last_nz[e]) continue;
        const int q = j & 3;
        if (j < 4) {
#pragma unroll
          for (int u = 0; u < 4; u++)
            if (q == u) m[e][u] = w[e];
        } else {
#pragma unroll
          for (int u = 0; u < 4; u++)
            if (q == u) acc[e][u] = w[e];
          if ((q == 3 || k == last_nz[e]) && !(PRV_ABLATE & 128)) {
            const uint32_t slot = base[e] + (uint32_t)__popcll(b[e] & ((1ull << lane) - 1ull));
            P.mem[e].queue_ext[(size_t)((j >> 2) - 1) * ext_stride + slot] = make_uint4(acc[e][0], acc[e][1], acc[e][2], acc[e][3]);
#pragma unroll
            for (int u = 0; u < 4; u++) acc[e][u] = 0u;
          }
        }
      }
    }
  }
#pragma unroll
  for (int e = 0; e < E; e++) {
    const MarchSink Q{P.mem[e].queue, P.mem[e].queue_ext, ext_stride, P.mem[e].queue_count, P.mem[e].out_f32, P.mem[e].out_u8};
    march_write<true, true>(P, Q, stage, r, first_nz[e] < 32, b[e], base[e], dt, m[e], first_nz[e], last_nz[e], [](int) { return 0u; });
  }
}

// byte c of the interleaved grids: bit e = bit c of member e's occupancy bitfield (and of its dilated coarse grid)
__global__ __launch_bounds__(256) void occ_interleave_kernel(OccInterleaveParams P) {
  const uint32_t c = blockIdx.x * 256u + threadIdx.x;
  if (c >= P.n_cells) return;
  uint32_t v = 0u;
  for (int e = 0; e < P.n_members; e++) v |= ((P.bits[e][c >> 5] >> (c & 31u)) & 1u) << e;
  P.out[c] = (uint8_t)v;
}

// ------------------------------------------------------------------ K_B render from the queue

// One wave per launch stamps the shader cycle counter (s_memtime) and the constant-rate reference counter
// (s_memrealtime) into the statistics block: {2, 3} accumulate the deltas, {4, 5} hold the start stamps.  The kernels
// are persistent, so block 0 lives for the whole launch and the ratio is the launch's average shader clock.
__device__ __forceinline__ void clock_stamp_begin(unsigned long long* stat) {
  if (blockIdx.x == 0 && threadIdx.x == 0) {
    stat[4] = __builtin_readcyclecounter();
    stat[5] = __builtin_amdgcn_s_memrealtime();
  }
}
__device__ __forceinline__ void clock_stamp_end(unsigned long long* stat) {
  if (blockIdx.x == 0 && threadIdx.x == 0) {
    atomicAdd(stat + 2, __builtin_readcyclecounter() - stat[4]);
    atomicAdd(stat + 3, __builtin_amdgcn_s_memrealtime() - stat[5]);
  }
}

// ------------------------------------------------------------------ K_B render from the queue, 64 ray slots per wave
// One lane = one ray slot and one whole sample per round: the lane picks its next live sample, gathers ALL levels
// itself (encode_sample: dense levels with no clamps and one-add neighbours, (1-w, w) pairs from one v_cvt_pk each),
// and two v_permlane32_swap per k-step turn the wave's 64 feature vectors into the MFMA B operands of two 32-sample
// column groups (lanes 0..31 = group A, lanes 32..63 = group B).  mlp_forward2 runs both groups off one LDS read of
// every weight fragment.  A group refills when all its 32 slots are idle (whole-group lockstep).
__device__ __forceinline__ int lane_id() { return (int)(threadIdx.x & 63u); }

template <int F, int NDENSE, bool NGP, bool CACHE = false>
__global__ __launch_bounds__(256)
// the cached instances must keep TWO waves per SIMD (<= 256 registers); the plain ones with dense levels are asked for THREE (<= 168): left alone
// they took 172 since the octant regions of round 6 -- two waves -- and three are 1.7-2.5 % faster (r06ay); the all-hashed ones would spill 50-80
__attribute__((amdgpu_waves_per_eu(CACHE ? 2 : NDENSE == 0 ? 1 : 3)))
void render_queue64_kernel(RenderParams P) {
  __shared__ half8 wl[kNumFrags * 64];
  __shared__ uint32_t mv[4][32][6]; // tail merges: {record, next sample, T, r, g, b} of the rays that change slots, per wave
  constexpr uint32_t kPoolCap = 192;
  __shared__ uint32_t pool[kPoolCap][6]; // the block's tail pool: rays a group gave up, waiting for an idle slot of ANY of the block's waves
  __shared__ uint32_t pool_n, pool_lock;
  for (int i = threadIdx.x; i < kNumFrags * 64; i += 256) wl[i] = P.field.frags64[i];
  if (threadIdx.x == 0) {
    pool_n = 0u;
    pool_lock = 0u;
  }
  __syncthreads();
  auto pool_acquire = [&]() {
    if (lane_id() == 0)
      while (atomicCAS(&pool_lock, 0u, 1u) != 0u) __builtin_amdgcn_s_sleep(2);
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
    __builtin_amdgcn_wave_barrier();
  };
  auto pool_release = [&]() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    __builtin_amdgcn_wave_barrier();
    if (lane_id() == 0) atomicExch(&pool_lock, 0u);
  };
  // every lane gathers every level, so the level constants are wave-uniform: they are read straight from the kernel
  // arguments (scalar loads, SGPR operands) instead of LDS -> VGPRs
  const LevelDev* __restrict__ lvl = P.field.levels;
  const HashConsts hc = {P.field.hash_my_b, P.field.hash_mz_b, P.field.hash_m_b, (uint32_t)P.field.wide_offsets};

  const int lane = threadIdx.x & 63, r = lane & 31, g = lane >> 5; // g: the lane's group (A = 0, B = 1) AND its k-row half

  bool active = false;
  uint32_t pix = 0, rec_i = 0; // rec_i: the ray's queue record (a ray that changes slots re-reads its constants from it)
  float o[3] = {0, 0, 0}, d[3] = {0, 0, 1}, t0 = 0.f, dt = 0.f;
  // live-sample mask: the current 32-step word (never 0 while active), the words of its 128-step chunk behind it, and
  // the step index of the current word's bit 0.  PRV_STEP_NGP: further chunks are fetched from the extension buffer
  // when this one is spent (next_chunk below); nothing about them lives in registers.
  uint32_t cur = 0, m1 = 0, m2 = 0, m3 = 0, base = 0;
  float T = 1.f, cr = 0.f, cg = 0.f, cb = 0.f;
  half8 shA = {0, 0, 0, 0, 0, 0, 0, 0}, shB = {0, 0, 0, 0, 0, 0, 0, 0}; // SH rows [8g, 8g+8) of the rays in slots (r, A) and (r, B)
  bool drained = false;
  unsigned long long n_eval = 0ull, n_rounds = 0ull;
  // CACHE instances: the lane's last cell and its corner entries on every hashed level (prv_device.hpp: CornerCache)
  constexpr int kCached = CACHE ? 32 / F - NDENSE : 1;
  CornerCache<F> cc[kCached];
#pragma unroll
  for (int j = 0; j < kCached; j++) cc[j].key = ~0u;
  clock_stamp_begin(P.stat_evaluated);
  uint32_t q_cur = 0, q_end = 0;
  const uint32_t n_seg = (uint32_t)P.n_segments;
  // (an XCD starts at the first sub-region of its own octant region and walks on from there)
  const uint32_t n_sub = (uint32_t)max(P.n_sub, 1);
  uint32_t seg = n_seg > 1u ? (((uint32_t)__builtin_amdgcn_s_getreg((3 << 11) | 20) & 7u) % (n_seg / n_sub)) * n_sub + (blockIdx.x * 4u + (threadIdx.x >> 6)) % n_sub : 0u;
  uint32_t seg_tried = 0;
  const uint4* __restrict__ queue = reinterpret_cast<const uint4*>(P.queue);

  // the lane takes over the ray of queue record ri, its mask walk positioned at sample `next` (a live sample of the
  // ray, or 0 for a fresh ray: then at its first live sample).  Constants come from the record, the dynamic state
  // (T and the colour sums) from the caller.
  auto take_ray = [&](uint32_t ri, uint32_t next) {
    const uint4* rec = queue + (size_t)ri * kRecordWords;
    const uint4 q0 = rec[0], q1 = rec[1], q3 = rec[3];
    uint4 mk = rec[2];
    o[0] = __uint_as_float(q0.x); o[1] = __uint_as_float(q0.y); o[2] = __uint_as_float(q0.z);
    t0 = __uint_as_float(q0.w);
    d[0] = __uint_as_float(q1.x); d[1] = __uint_as_float(q1.y); d[2] = __uint_as_float(q1.z);
    dt = __uint_as_float(q1.w);
    pix = q3.x;
    rec_i = ri;
    base = 0u;
    if constexpr (NGP) {
      const uint32_t base0 = q3.y & 0xffffu; // first step of the record's own chunk
      base = base0;
      if (next >= base0 + 128u) { // the sample lies in a later chunk
        const uint32_t j = (next - base0) >> 7;
        mk = P.queue_ext[(size_t)(j - 1u) * ((size_t)P.seg_cap * (size_t)P.n_segments) + (size_t)ri];
        base = base0 + 128u * j;
      }
    }
    cur = mk.x; m1 = mk.y; m2 = mk.z; m3 = mk.w;
    while (base + 32u <= next) { // the words before the sample are spent
      cur = m1; m1 = m2; m2 = m3; m3 = 0u;
      base += 32u;
    }
    if (next > base) cur &= ~0u << (next & 31u); // chunks start at multiples of 32: next & 31 == next - base
    while (cur == 0u && (m1 | m2 | m3) != 0u) { // fresh FIXED_S rays: the current word must not be empty
      cur = m1; m1 = m2; m2 = m3; m3 = 0u;
      base += 32u;
    }
    active = true;
  };
  // PRV_STEP_NGP, the chunk in the registers is spent: fetch the ray's next non-empty chunk; false = that was the last
  auto next_chunk = [&]() -> bool {
    const uint32_t info = queue[(size_t)rec_i * kRecordWords + 3].y;
    const uint32_t base0 = info & 0xffffu, n_chunks = info >> 16;
    for (uint32_t j = ((base - base0) >> 7) + 1u; j < n_chunks; j++) {
      const uint4 mk = P.queue_ext[(size_t)(j - 1u) * ((size_t)P.seg_cap * (size_t)P.n_segments) + (size_t)rec_i];
      if ((mk.x | mk.y | mk.z | mk.w) == 0u) continue; // a gap between two occupied stretches
      cur = mk.x; m1 = mk.y; m2 = mk.z; m3 = mk.w;
      base = base0 + 128u * j;
      while (cur == 0u) {
        cur = m1; m1 = m2; m2 = m3; m3 = 0u;
        base += 32u;
      }
      return true;
    }
    return false;
  };

  for (;;) {
    // ---- tail merge.  A cohort of 32 rays starts in lockstep (adjacent pixels, same depth: their gathers share cache
    // lines) and thins out as its rays terminate; its last few rays would hold 32 slots' worth of instruction issue.
    // When one group is down to <= merge_max rays and the other group has that many idle slots, the rays move over
    // (dynamic state through LDS, constants re-read from the queue record) and the emptied group takes 32 fresh rays
    // below.  Which lane composites a ray changes, the arithmetic and the sample order do not: pixels are unchanged.
    // (both relocation steps are looked at every fourth iteration: a tail leaves at most three rounds later, the
    // bookkeeping costs a quarter)
    const bool relocate = (n_rounds & 6ull) == 0ull;
    if (relocate && !drained && P.merge_max > 0) {
      const unsigned long long act0 = __ballot(active);
      const uint32_t aA = (uint32_t)act0, aB = (uint32_t)(act0 >> 32);
      const uint32_t nA = (uint32_t)__popc(aA), nB = (uint32_t)__popc(aB);
      int src = -1;
      if (nA != 0u && nB != 0u) {
        if (nA <= (uint32_t)P.merge_max && nA <= 32u - nB) src = 0;
        else if (nB <= (uint32_t)P.merge_max && nB <= 32u - nA) src = 1;
      }
      if (src >= 0) { // wave-uniform
        const uint32_t a_src = src == 0 ? aA : aB, a_dst = src == 0 ? aB : aA, n_src = src == 0 ? nA : nB;
        const uint32_t below = (1u << r) - 1u;
        uint32_t(*slot)[6] = mv[threadIdx.x >> 6];
        if (g == src && active) {
          uint32_t* e = slot[__popc(a_src & below)];
          e[0] = rec_i;
          e[1] = base + (uint32_t)__builtin_ctz(cur); // the next sample to take
          e[2] = __float_as_uint(T);
          e[3] = __float_as_uint(cr);
          e[4] = __float_as_uint(cg);
          e[5] = __float_as_uint(cb);
          active = false;
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        const uint32_t kth = (uint32_t)__popc(~a_dst & below); // rank of slot r among the destination group's idle slots
        if (!((a_dst >> r) & 1u) && kth < n_src) { // BOTH lanes of the slot: the slot's SH rows go to both halves
          const uint32_t* e = slot[kth];
          const uint32_t ri = e[0];
          const half8 sh = reinterpret_cast<const half8*>(queue + (size_t)ri * kRecordWords)[4 + g];
          if (src == 0) shB = sh;
          else shA = sh;
          if (g != src) { // the destination lane itself takes the ray over
            T = __uint_as_float(e[2]); cr = __uint_as_float(e[3]); cg = __uint_as_float(e[4]); cb = __uint_as_float(e[5]);
            take_ray(ri, e[1]);
          }
        }
        __builtin_amdgcn_wave_barrier(); // the scratch is reused by the next merge
      }
    }
    // ---- the block's tail pool.  What the in-wave merge above cannot place (the sibling group has no room) goes to a
    // small LDS pool shared by the block's four waves: a group down to <= merge_max rays DEPOSITS them there (and takes 32
    // fresh rays below); any group with idle slots that is not about to refill ADOPTS from the pool.  Fresh rays still
    // start as whole 32-ray cohorts; only thinned-out tails travel.  A spin lock (one lane per wave) guards the pool.
    if (P.pool_on && (relocate || drained)) {
      const unsigned long long act1 = __ballot(active);
      const uint32_t below = (1u << r) - 1u;
      const uint32_t in_pool = *(volatile uint32_t*)&pool_n; // a stale value only delays a decision by a round
#pragma unroll
      for (int grp = 0; grp < 2; grp++) {
        const uint32_t a_g = grp == 0 ? (uint32_t)act1 : (uint32_t)(act1 >> 32), n_g = (uint32_t)__popc(a_g);
        if (!drained && n_g != 0u && n_g <= (uint32_t)P.merge_max && in_pool + n_g <= kPoolCap) { // deposit
          pool_acquire();
          const uint32_t at = *(volatile uint32_t*)&pool_n;
          const bool fits = at + n_g <= kPoolCap;
          if (fits && g == grp && active) {
            uint32_t* e = pool[at + (uint32_t)__popc(a_g & below)];
            e[0] = rec_i;
            e[1] = base + (uint32_t)__builtin_ctz(cur);
            e[2] = __float_as_uint(T);
            e[3] = __float_as_uint(cr);
            e[4] = __float_as_uint(cg);
            e[5] = __float_as_uint(cb);
            active = false;
          }
          if (fits && lane == 0) *(volatile uint32_t*)&pool_n = at + n_g;
          pool_release();
        }
      }
      const unsigned long long act2 = __ballot(active);
#pragma unroll
      for (int grp = 0; grp < 2; grp++) {
        const uint32_t a_g = grp == 0 ? (uint32_t)act2 : (uint32_t)(act2 >> 32);
        const uint32_t n_idle = 32u - (uint32_t)__popc(a_g);
        // adopt into groups that keep running (some slots busy), or into anything once the queue is drained
        if (n_idle == 0u || (a_g == 0u && !drained) || *(volatile uint32_t*)&pool_n == 0u) continue;
        pool_acquire();
        const uint32_t have = *(volatile uint32_t*)&pool_n;
        const uint32_t take = min(have, n_idle), from = have - take;
        const uint32_t kth = (uint32_t)__popc(~a_g & below);
        uint32_t ent[6] = {0, 0, 0, 0, 0, 0};
        const bool mine = !((a_g >> r) & 1u) && kth < take; // both lanes of the slot
        if (mine) {
#pragma unroll
          for (int q = 0; q < 6; q++) ent[q] = pool[from + kth][q];
        }
        if (lane == 0) *(volatile uint32_t*)&pool_n = from;
        pool_release();
        if (mine) {
          const half8 sh = reinterpret_cast<const half8*>(queue + (size_t)ent[0] * kRecordWords)[4 + g];
          if (grp == 0) shA = sh;
          else shB = sh;
          if (g == grp) {
            T = __uint_as_float(ent[2]); cr = __uint_as_float(ent[3]); cg = __uint_as_float(ent[4]); cb = __uint_as_float(ent[5]);
            take_ray(ent[0], ent[1]);
          }
        }
      }
    }
    // ---- refill: a group whose 32 slots are all idle takes the next records of the wave's claimed range
    const unsigned long long idle = __ballot(!active);
    const bool needA = (uint32_t)idle == 0xffffffffu, needB = (uint32_t)(idle >> 32) == 0xffffffffu;
    if (!drained && (needA || needB)) {
#pragma unroll
      for (int grp = 0; grp < 2; grp++) {
        if (!(grp == 0 ? needA : needB)) continue;
        while (q_cur == q_end && !drained) {
          uint32_t claim = 0;
          if (lane == 0) claim = atomicAdd(P.queue_head + 16u * seg, kClaim);
          claim = __builtin_amdgcn_readfirstlane(claim);
          // region `seg` of the queue: the records added to it, less the slots a reservation left empty at its end (region_reserve)
          const uint32_t s_lo = seg * P.seg_cap, s_cnt = min(P.queue_count[16u * seg], P.seg_cap) - P.queue_count[16u * seg + 1u];
          if (claim < s_cnt) {
            q_cur = s_lo + claim;
            q_end = min(q_cur + kClaim, s_lo + s_cnt);
          } else if (++seg_tried >= n_seg) {
            drained = true;
          } else {
            seg = seg + 1u == n_seg ? 0u : seg + 1u;
          }
        }
        const uint32_t avail = min(32u, q_end - q_cur);
        if ((uint32_t)r < avail) { // both lane halves: the group's SH rows go to every lane, the ray itself to its own lane
          const half8 sh = reinterpret_cast<const half8*>(queue + (size_t)(q_cur + (uint32_t)r) * kRecordWords)[4 + g];
          if (grp == 0) shA = sh;
          else shB = sh;
          if (g == grp) {
            T = 1.f; cr = 0.f; cg = 0.f; cb = 0.f;
            take_ray(q_cur + (uint32_t)r, 0u);
          }
        }
        q_cur += avail;
      }
    }
    const unsigned long long act = __ballot(active);
    if (act == 0ull) {
      // nothing in this wave: leave once the queue is drained AND the pool is empty (whatever is deposited later comes
      // from a wave that is still running and will adopt it itself once it has drained)
      if (drained && (!P.pool_on || *(volatile uint32_t*)&pool_n == 0u)) break;
      continue;
    }
    n_eval += (unsigned long long)__popcll(act);
    n_rounds += 2ull; // counted in 32-slot units: utilisation = evaluated / (32 * rounds)

    // ---- this lane's next live sample, all levels
    // idle lanes feed whatever their registers hold into their own MFMA column: columns are independent and an idle
    // lane's results are never read, so the features are deliberately left unset instead of zeroed (16 v_mov per round)
    half8 f[4];
#pragma unroll
    for (int s = 0; s < 4; s++) asm volatile("" : "=v"(f[s]));
    bool last = false;
    if (active) {
      const uint32_t i = base + (uint32_t)__builtin_ctz(cur);
      cur &= cur - 1u;
      if (cur == 0u) {
        last = (m1 | m2 | m3) == 0u;
        while (cur == 0u && !last) {
          cur = m1; m1 = m2; m2 = m3; m3 = 0u;
          base += 32u;
        }
        if constexpr (NGP) {
          if (last) last = !next_chunk();
        }
      }
      const float t = fmaf((float)i + 0.5f, dt, t0);
      encode_sample<F, NDENSE, CACHE>(P.field.table, lvl, hc, fmaf(t, d[0], o[0]), fmaf(t, d[1], o[1]), fmaf(t, d[2], o[2]), f, cc);
    }
    // f[2s] | f[2s+1] = k rows [16s, 16s+8) | [16s+8, 16s+16) of the lane's own sample -> B operands of the two groups
    swap_halves(f[0], f[1]); // f[0] = group A k-step 0, f[1] = group B k-step 0
    swap_halves(f[2], f[3]);
    const half8 fA[2] = {f[0], f[2]}, fB[2] = {f[1], f[3]};
    const MlpOut2 mo = mlp_forward2(wl, lane, fA, fB, shA, shB);
    // the lane's own sample: group A's results sit in lane half 0 (rows 0..2 = registers 0..2), group B's copies in the
    // padding rows 20..22 = lane half 1, registers 8..10
    const float dens = g ? mo.densB[8] : mo.densA[0];
    const float lr = g ? mo.rgbB[8] : mo.rgbA[0], lg = g ? mo.rgbB[9] : mo.rgbA[1], lb = g ? mo.rgbB[10] : mo.rgbA[2];

    bool done = false;
    if (active) {
      const float sigma = fast_exp(dens + P.field.density_bias);
      const float alpha = 1.0f - fast_exp(-(sigma * dt));
      const float wgt = alpha * T;
      cr = fmaf(wgt, fast_sigmoid(lr), cr);
      cg = fmaf(wgt, fast_sigmoid(lg), cg);
      cb = fmaf(wgt, fast_sigmoid(lb), cb);
      T = T * (1.0f - alpha);
      done = last || T < P.min_T;
    }
    if (done) {
      float4* out = reinterpret_cast<float4*>(P.out_f32) + pix;
      float4 v = make_float4(cr, cg, cb, 1.0f - T);
      if (P.spp_k != 0) {
        const float4 prev = *out;
        v.x = prev.x + v.x; v.y = prev.y + v.y; v.z = prev.z + v.z; v.w = prev.w + v.w;
      }
      if (P.last_pass) {
        v.x *= P.inv_spp; v.y *= P.inv_spp; v.z *= P.inv_spp; v.w *= P.inv_spp;
        if (P.out_u8) P.out_u8[pix] = quantize_rgba8(v.x, v.y, v.z, v.w, P.bg);
      }
      *out = v;
      active = false;
    }
  }
  if (lane == 0 && n_eval) {
    atomicAdd(P.stat_evaluated, n_eval);
    atomicAdd(P.stat_evaluated + 1, n_rounds);
  }
  clock_stamp_end(P.stat_evaluated);
}


// ------------------------------------------------------------------ first-hit ray cast (a13)
// first occupied cell along (o, d) within max_range, or -1: Amanatides-Woo over the occupancy bits
__device__ __forceinline__ int32_t first_hit_dda(const FieldDev& fd, const float o[3], const float d[3], float max_range) {
  float t0, t1;
  int32_t result = -1;
  if (ray_aabb(o, d, t0, t1)) {
    if (t1 > max_range) t1 = max_range;
    if (t1 > t0) {
      const int R = fd.occ_res;
      const float fR = (float)R;
      const float ts = t0 + 1e-6f;
      int c[3], step[3];
      float tmax[3], tdelta[3];
#pragma unroll
      for (int a = 0; a < 3; a++) {
        const float p = clamp01(fmaf(ts, d[a], o[a])) * fR;
        c[a] = min((int)p, R - 1);
        if (d[a] > 0.f) {
          step[a] = 1;
          tmax[a] = (((float)(c[a] + 1)) / fR - o[a]) / d[a];
          tdelta[a] = 1.0f / (fR * d[a]);
        } else if (d[a] < 0.f) {
          step[a] = -1;
          tmax[a] = (((float)c[a]) / fR - o[a]) / d[a];
          tdelta[a] = -1.0f / (fR * d[a]);
        } else {
          step[a] = 0;
          tmax[a] = __builtin_inff();
          tdelta[a] = __builtin_inff();
        }
      }
      for (;;) {
        const uint32_t bit = (uint32_t)c[0] + (uint32_t)R * ((uint32_t)c[1] + (uint32_t)R * (uint32_t)c[2]);
        if ((fd.occ[bit >> 5] >> (bit & 31)) & 1u) {
          result = (int32_t)bit;
          break;
        }
        const int a = tmax[0] < tmax[1] ? (tmax[0] < tmax[2] ? 0 : 2) : (tmax[1] < tmax[2] ? 1 : 2);
        const float tm = a == 0 ? tmax[0] : (a == 1 ? tmax[1] : tmax[2]);
        if (tm > t1) break;
        if (a == 0) { c[0] += step[0]; tmax[0] += tdelta[0]; if (c[0] < 0 || c[0] >= R) break; }
        else if (a == 1) { c[1] += step[1]; tmax[1] += tdelta[1]; if (c[1] < 0 || c[1] >= R) break; }
        else { c[2] += step[2]; tmax[2] += tdelta[2]; if (c[2] < 0 || c[2] >= R) break; }
      }
    }
  }
  return result;
}

// GPU twin of the reference's CPU render path Perception_3D::precept_thread_process
// (main.cpp:238-284): per pixel, the first occupied voxel along the ray (there: OctoMap
// castRay, here: Amanatides-Woo DDA over the occupancy bitfield), max_range like main.cpp:258.
// out = linear cell index x + R*(y + R*z), or -1.  Same float op order as the oracle.
__global__ __launch_bounds__(256) void first_hit_kernel(FieldDev fd, const CamDev* __restrict__ cams, int W, int H,
                                                        float max_range, int32_t* __restrict__ out) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= W * H) return;
  const int v = blockIdx.y;
  const int px = i % W, py = i / W;
  float o[3], d[3], t0, t1;
  raygen(cams[v], px, py, 0.5f, 0.5f, o, d);
  (void)t0;
  (void)t1;
  const int32_t result = first_hit_dda(fd, o, d, max_range);
  out[(size_t)v * W * H + i] = result;
}

// Perception_3D::precept_thread_process per ground-truth voxel (main.cpp:238-284): project through
// the RealSense model, cull, integer pixel, deproject at depth 1, cast from the camera, first hit.
__global__ __launch_bounds__(256) void precept_kernel(FieldDev fd, const float* __restrict__ voxels, int n, PreceptPose pose,
                                                      Rs2Intr in, float max_range, int32_t* __restrict__ out) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  const float origin[3] = {(float)pose.c2w[3], (float)pose.c2w[7], (float)pose.c2w[11]};
  double v[3];
#pragma unroll
  for (int r = 0; r < 3; r++)
    v[r] = pose.w2c[r * 4] * (double)voxels[i * 3] + pose.w2c[r * 4 + 1] * (double)voxels[i * 3 + 1] +
           pose.w2c[r * 4 + 2] * (double)voxels[i * 3 + 2] + pose.w2c[r * 4 + 3];
  const float point_3d[3] = {(float)v[0], (float)v[1], (float)v[2]};
  float pixel[2];
  rs2_project(pixel, in, point_3d);
  int32_t result = -1;
  if (!(pixel[0] < 0 || pixel[0] > (float)in.width || pixel[1] < 0 || pixel[1] > (float)in.height)) { // :248-251
    const float ipx[2] = {(float)(int)pixel[0], (float)(int)pixel[1]}; // :253 int x, int y
    float pt[3], d[3];
    rs2_deproject(pt, in, ipx, 1.0f);
#pragma unroll
    for (int r = 0; r < 3; r++) {
      const float e = (float)(pose.c2w[r * 4] * (double)pt[0] + pose.c2w[r * 4 + 1] * (double)pt[1] +
                              pose.c2w[r * 4 + 2] * (double)pt[2] + pose.c2w[r * 4 + 3]);
      d[r] = e - origin[r];
    }
    const float n2 = fmaf(d[0], d[0], fmaf(d[1], d[1], d[2] * d[2]));
    const float inv = 1.0f / sqrtf(n2);
#pragma unroll
    for (int r = 0; r < 3; r++) d[r] = d[r] * inv;
    result = first_hit_dda(fd, origin, d, max_range);
  }
  out[i] = result;
}

// ------------------------------------------------------------------ multi-sample reduce
// stage[k][pixel] -> out[pixel] = (((s0 + s1) + s2) + ...) * inv_spp : the summation order of the
// per-pass accumulation it replaces (and of the oracle), so results are bit-identical.
__global__ __launch_bounds__(256) void spp_reduce_kernel(const float4* __restrict__ stage, size_t n, int spp, float inv_spp,
                                                         float b0, float b1, float b2, float b3,
                                                         float4* __restrict__ out, uint32_t* __restrict__ out_u8) {
  const float bg[4] = {b0, b1, b2, b3};
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
    float4 a = stage[i];
    for (int k = 1; k < spp; k++) {
      const float4 v = stage[(size_t)k * n + i];
      a.x = a.x + v.x; a.y = a.y + v.y; a.z = a.z + v.z; a.w = a.w + v.w;
    }
    a.x *= inv_spp; a.y *= inv_spp; a.z *= inv_spp; a.w *= inv_spp;
    out[i] = a;
    if (out_u8) out_u8[i] = quantize_rgba8(a.x, a.y, a.z, a.w, bg);
  }
}

// ------------------------------------------------------------------ quantise

__global__ __launch_bounds__(256) void quantize_kernel(const float4* __restrict__ in, size_t n,
                                                       float b0, float b1, float b2, float b3,
                                                       uint32_t* __restrict__ out) {
  const float bg[4] = {b0, b1, b2, b3};
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
    float4 v = in[i];
    out[i] = quantize_rgba8(v.x, v.y, v.z, v.w, bg);
  }
}

// ------------------------------------------------------------------ score reductions

// fixed-order block reduction of a double: lane shuffles, then 4 wave partials via LDS
__device__ __forceinline__ double block_reduce_sum(double v, double* sm) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off);
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  if (lane == 0) sm[w] = v;
  __syncthreads();
  double r = 0.0;
  if (threadIdx.x == 0) r = ((sm[0] + sm[1]) + sm[2]) + sm[3];
  __syncthreads();
  return r;
}

// ensemble scores from uint8 RGBA images (main.cpp:2053-2086 method 2, 2113-2150 method 3), in two passes so that
// the score is the reference's own SEQUENTIAL double sum (rows, columns, then the statements of the loop body in
// order), bit for bit, not a tree of partial sums: views whose scores tie mathematically (ensembles that agree on
// almost every pixel produce a handful of distinct variance values) then tie, or break, exactly as they do in the
// reference's loop.  Pass 1 (parallel): the addends of every pixel, each computed with the reference's operation
// order.  cv::imread hands the reference BGRA pixels, so its "r, g, b" are bytes 2, 1, 0 of the PNG's RGBA order.
template <int METHOD>
__global__ __launch_bounds__(256) void score_ensemble_terms_kernel(EnsembleParams P) {
  constexpr int K = METHOD == PRV_SCORE_ENSEMBLE_RGB ? 3 : 2;
  const size_t npix = P.pixels_per_view;
  const size_t v = blockIdx.y;
  double* terms = P.partial + v * npix * K;
  for (size_t p = (size_t)blockIdx.x * 256 + threadIdx.x; p < npix; p += (size_t)gridDim.x * 256) {
    uint32_t px[PRV_MAX_MODELS];
    for (int e = 0; e < P.E; e++) px[e] = P.imgs[e][((size_t)P.view0 + v) * npix + p];
    double var3[3];
#pragma unroll
    for (int k = 0; k < 3; k++) {
      const int sh = 8 * (2 - k); // reference channel k = byte 2 - k
      double mean = 0.0;
      for (int e = 0; e < P.E; e++) mean += (double)((px[e] >> sh) & 255u);
      mean /= (double)P.E;
      double var = 0.0;
      for (int e = 0; e < P.E; e++) {
        double dlt = (double)((px[e] >> sh) & 255u) - mean;
        var += dlt * dlt;
      }
      var3[k] = var / (double)P.E;
    }
    if (METHOD == PRV_SCORE_ENSEMBLE_RGB) {
#pragma unroll
      for (int k = 0; k < 3; k++) terms[p * 3 + k] = var3[k] > 1e-10 ? log(var3[k]) : 0.0; // x + 0.0 == x: a skipped addend
    } else {
      double md = 0.0;
      for (int e = 0; e < P.E; e++) md += (double)(px[e] >> 24) / 255.0;
      md /= (double)P.E;
      terms[p * 2 + 0] = (var3[0] + var3[1] + var3[2]) / 3.0;
      terms[p * 2 + 1] = (1.0 - md) * (1.0 - md);
    }
  }
}
template __global__ void score_ensemble_terms_kernel<PRV_SCORE_ENSEMBLE_RGB>(EnsembleParams);
template __global__ void score_ensemble_terms_kernel<PRV_SCORE_ENSEMBLE_RGB_DENSITY>(EnsembleParams);

// Pass 2: one wave per view adds the view's addends in order.  The chain of dependent double adds is the cost
// (~n_terms x the add latency: 10,800 addends at the reference's 80x45 -> tens of microseconds, all views in
// parallel); the wave streams the addends through LDS 512 at a time -- coalesced loads of the next chunk are in
// flight while every lane adds the current one from broadcast LDS reads -- so memory latency stays off the chain.
__global__ __launch_bounds__(64) void score_sequential_sum_kernel(const double* __restrict__ terms, size_t n_terms,
                                                                  prv_score_record* __restrict__ rec) {
  __shared__ double buf[2][512];
  const size_t v = blockIdx.x;
  const double* t = terms + v * n_terms;
  const int lane = threadIdx.x;
  const size_t n_chunks = (n_terms + 511) / 512;
  double r[8];
#pragma unroll
  for (int j = 0; j < 8; j++) {
    const size_t i = (size_t)j * 64 + lane;
    r[j] = i < n_terms ? t[i] : 0.0;
  }
  double sum = 0.0;
  for (size_t c = 0; c < n_chunks; c++) {
    double* b = buf[c & 1];
#pragma unroll
    for (int j = 0; j < 8; j++) b[j * 64 + lane] = r[j];
    __syncthreads();
    if (c + 1 < n_chunks) {
#pragma unroll
      for (int j = 0; j < 8; j++) {
        const size_t i = (c + 1) * 512 + (size_t)j * 64 + lane;
        r[j] = i < n_terms ? t[i] : 0.0; // padding adds +0.0: the sum never is -0.0, so its bits do not change
      }
    }
#pragma unroll 16
    for (int i = 0; i < 512; i++) sum += b[i];
  }
  if (lane == 0) {
    prv_score_record out;
    out.score = sum;
    out.psnr = 0.f;
    out.coverage = 0.f;
    rec[v] = out;
  }
}

// PSNR recipe of run.py:257-263 + mean opacity + the density term of main.cpp:2148 ((1 - alpha)^2 per pixel);
// three partial sums per block
__global__ __launch_bounds__(256) void score_psnr_kernel(PsnrParams P) {
  __shared__ double sm[4];
  const int v = blockIdx.y;
  const size_t npix = P.pixels_per_view;
  const float4* img = reinterpret_cast<const float4*>(P.rgba) + (size_t)v * npix;
  const float4* gt = reinterpret_cast<const float4*>(P.gt) + (size_t)v * npix;
  double se = 0.0, cov = 0.0, unc = 0.0;
  // P.cams (optional): the render wrote only the tiles inside the view's cull rectangle (MarchParams::live_grid); every
  // pixel outside it is a dead ray's (0, 0, 0, 0) by construction and is not read.  Same pixels per thread in the same
  // order either way: the sums are bit-identical to scoring a fully written image.
  int c0 = 0, c1 = 0, c2 = 0, c3 = 0;
  if (P.cams && P.cams[v].cull[2] > 0) {
    c0 = P.cams[v].cull[0]; c1 = P.cams[v].cull[1]; c2 = P.cams[v].cull[2]; c3 = P.cams[v].cull[3];
  }
  for (size_t p = (size_t)blockIdx.x * 256 + threadIdx.x; p < npix; p += (size_t)gridDim.x * 256) {
    float4 a4 = make_float4(0.f, 0.f, 0.f, 0.f);
    bool inside = true;
    if (c2 > 0) {
      const int y = (int)((uint32_t)p / (uint32_t)P.W), x = (int)((uint32_t)p - (uint32_t)y * (uint32_t)P.W);
      inside = x >= c0 && x < c2 && y >= c1 && y < c3;
    }
    if (inside) a4 = img[p];
    const float4 r4 = gt[p];
    const float ra = 1.0f - a4.w, rg = 1.0f - r4.w;
    const float av[3] = {a4.x, a4.y, a4.z}, rv[3] = {r4.x, r4.y, r4.z};
#pragma unroll
    for (int k = 0; k < 3; k++) {
      float a = fmaf(ra, P.bg[k], av[k]);
      float rr = fmaf(rg, P.bg[k], rv[k]);
      a = fminf(fmaxf(linear_to_srgb(a), 0.0f), 1.0f);
      rr = fminf(fmaxf(linear_to_srgb(rr), 0.0f), 1.0f);
      double dlt = (double)a - (double)rr;
      se += dlt * dlt;
    }
    cov += (double)a4.w;
    const double u = 1.0 - (double)a4.w;
    unc += u * u;
  }
  double s0 = block_reduce_sum(se, sm);
  double s1 = block_reduce_sum(cov, sm);
  double s2 = block_reduce_sum(unc, sm);
  if (threadIdx.x == 0) {
    P.partial[((size_t)v * gridDim.x + blockIdx.x) * 3 + 0] = s0;
    P.partial[((size_t)v * gridDim.x + blockIdx.x) * 3 + 1] = s1;
    P.partial[((size_t)v * gridDim.x + blockIdx.x) * 3 + 2] = s2;
  }
}

// ---- SSIM of run.py:260 (recipe assumed from upstream common.py; see oracle orc_ssim)
__device__ __forceinline__ float ssim_lum(const float4 p, const float bg[4]) {
  const float rem = 1.0f - p.w;
  const float c[3] = {p.x, p.y, p.z};
  const float kw[3] = {0.2126f, 0.7152f, 0.0722f};
  float l = 0.0f;
#pragma unroll
  for (int k = 0; k < 3; k++) {
    float v = fmaf(rem, bg[k], c[k]);
    v = fminf(fmaxf(linear_to_srgb(v), 0.0f), 1.0f);
    l = fmaf(kw[k], powf(fmaxf(v, 0.0f), 0.4545454545f), l);
  }
  return l;
}

__global__ __launch_bounds__(256) void ssim_lum_kernel(const float4* __restrict__ img, const float4* __restrict__ gt,
                                                       size_t n, float b0, float b1, float b2, float b3,
                                                       float* __restrict__ la, float* __restrict__ lb) {
  const float bg[4] = {b0, b1, b2, b3};
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
    la[i] = ssim_lum(img[i], bg);
    lb[i] = ssim_lum(gt[i], bg);
  }
}

__global__ __launch_bounds__(256) void ssim_map_kernel(const float* __restrict__ la_all, const float* __restrict__ lb_all,
                                                       int W, int H, double* __restrict__ partial) {
  __shared__ double sm[4];
  const int v = blockIdx.y;
  const float* la = la_all + (size_t)v * W * H;
  const float* lb = lb_all + (size_t)v * W * H;
  const int ow = W - 4, oh = H - 4;
  const float tap[5] = {0.120078f, 0.233881f, 0.292082f, 0.233881f, 0.120078f};
  const float c1 = 0.01f * 0.01f, c2 = 0.03f * 0.03f;
  double acc = 0.0;
  for (int p = blockIdx.x * 256 + threadIdx.x; p < ow * oh; p += gridDim.x * 256) {
    const int x = p % ow, y = p / ow;
    float r[5][5]; // per window row: blurred a, b, aa, bb, ab
#pragma unroll
    for (int i = 0; i < 5; i++) {
      float sa = 0.f, sb = 0.f, saa = 0.f, sbb = 0.f, sab = 0.f;
#pragma unroll
      for (int j = 0; j < 5; j++) {
        const float a = la[(size_t)(y + i) * W + (x + j)], b = lb[(size_t)(y + i) * W + (x + j)];
        sa = fmaf(tap[j], a, sa);
        sb = fmaf(tap[j], b, sb);
        saa = fmaf(tap[j], a * a, saa);
        sbb = fmaf(tap[j], b * b, sbb);
        sab = fmaf(tap[j], a * b, sab);
      }
      r[i][0] = sa; r[i][1] = sb; r[i][2] = saa; r[i][3] = sbb; r[i][4] = sab;
    }
    float q[5];
#pragma unroll
    for (int k = 0; k < 5; k++) {
      float s = 0.f;
#pragma unroll
      for (int i = 0; i < 5; i++) s = fmaf(tap[i], r[i][k], s);
      q[k] = s;
    }
    const float mA = q[0], mB = q[1];
    const float sA = q[2] - mA * mA, sB = q[3] - mB * mB, sAB = q[4] - mA * mB;
    const float p1 = (2.0f * mA * mB + c1) / (mA * mA + mB * mB + c1);
    const float p2 = (2.0f * sAB + c2) / (sA + sB + c2);
    acc += (double)(p1 * p2);
  }
  const double s = block_reduce_sum(acc, sm);
  if (threadIdx.x == 0) partial[(size_t)v * gridDim.x + blockIdx.x] = s;
}

__global__ void ssim_finalize_kernel(const double* __restrict__ partial, int n_views, int n_blocks, double count,
                                     double* __restrict__ out) {
  const int v = blockIdx.x * blockDim.x + threadIdx.x;
  if (v >= n_views) return;
  double s = 0.0;
  for (int b = 0; b < n_blocks; b++) s += partial[(size_t)v * n_blocks + b];
  out[v] = s / count;
}

// one thread per view: sum the block partials in block order, emit the record
__global__ void score_finalize_kernel(const double* __restrict__ partial, int n_views, int n_blocks,
                                      int method, size_t pixels_per_view, double coverage_weight,
                                      prv_score_record* __restrict__ rec) {
  const int v = blockIdx.x * blockDim.x + threadIdx.x;
  if (v >= n_views) return;
  prv_score_record out;
  out.psnr = 0.f;
  out.coverage = 0.f;
  if (method == PRV_SCORE_PSNR_COVERAGE) {
    double se = 0.0, cov = 0.0, unc = 0.0;
    for (int b = 0; b < n_blocks; b++) {
      se += partial[((size_t)v * n_blocks + b) * 3 + 0];
      cov += partial[((size_t)v * n_blocks + b) * 3 + 1];
      unc += partial[((size_t)v * n_blocks + b) * 3 + 2];
    }
    const double mse = se / (double)(pixels_per_view * 3);
    const double psnr = -10.0 * log10(mse);
    // worst-reconstructed AND least-covered view first: -PSNR plus the reference's density term (main.cpp:2148,
    // (1 - alpha)^2 per pixel, here its mean) times the documented weight (prv_set_coverage_weight, default 1)
    out.score = -psnr + coverage_weight * (unc / (double)pixels_per_view);
    out.psnr = (float)psnr;
    out.coverage = (float)(cov / (double)pixels_per_view);
  } else {
    double s = 0.0;
    for (int b = 0; b < n_blocks; b++) s += partial[(size_t)v * n_blocks + b];
    out.score = s;
  }
  rec[v] = out;
}

// ------------------------------------------------------------------ synthetic table (counter RNG)

__device__ __forceinline__ unsigned long long mix64(unsigned long long z) {
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
  return z ^ (z >> 31);
}

__global__ __launch_bounds__(256) void synth_table_kernel(uint16_t* __restrict__ table, size_t n,
                                                          unsigned long long seed, float amp) {
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
    unsigned long long hsh = mix64(seed + 1ull * 0xD1B54A32D192ED03ull + (unsigned long long)i * 0x9E3779B97F4A7C15ull);
    uint32_t u = (uint32_t)(hsh >> 40);
    float v = (float)u * (1.0f / 8388608.0f) - 1.0f;
    _Float16 hv = to_half(v * amp);
    table[i] = __builtin_bit_cast(uint16_t, hv);
  }
}

// canonical (ABI) table -> physical layout of one level.
//  dense level  : one thread per vertex, scattered to power-of-two strides (+ the duplicated border)
//  hashed level : straight copy
template <int F>
__global__ __launch_bounds__(256) void repack_level_kernel(const uint16_t* __restrict__ canon,
                                                           uint16_t* __restrict__ phys, RepackLevel L) {
  typedef typename EntryWord<F>::type word_t;
  const word_t* src = reinterpret_cast<const word_t*>(canon) + L.canon_off;
  word_t* dst = reinterpret_cast<word_t*>(phys) + L.phys_off;
  if (L.hashed) { // straight copy
    for (uint32_t i = blockIdx.x * 256 + threadIdx.x; i < L.n; i += gridDim.x * 256) dst[i] = src[i];
    return;
  }
  // dense level: one thread per VERTEX.
  // The last vertex of every row, the last row of every plane and the last plane are duplicated one step further out,
  // so vertex + 1 on any axis reads what the min(c + 1, res - 1) clamp would have read (paired x loads, clamp-free y / z).
  const uint32_t nv = L.res * L.res * L.res;
  for (uint32_t i = blockIdx.x * 256 + threadIdx.x; i < nv; i += gridDim.x * 256) {
    const uint32_t x = i % L.res, y = (i / L.res) % L.res, z = i / (L.res * L.res);
    const word_t v = src[i];
    const uint32_t ex = x == L.res - 1 ? 1u : 0u, ey = y == L.res - 1 ? 1u : 0u, ez = z == L.res - 1 ? 1u : 0u;
    for (uint32_t dz = 0; dz <= ez; dz++)
      for (uint32_t dy = 0; dy <= ey; dy++)
        for (uint32_t dx = 0; dx <= ex; dx++) dst[(x + dx) | ((y + dy) << L.sx) | ((z + dz) << (2 * L.sx))] = v;
  }
}

// ------------------------------------------------------------------ stage hooks (parity tests)

__global__ __launch_bounds__(256) void debug_raygen_kernel(CamDev cam, int W, int H, int spp_k,
                                                           float* __restrict__ o_out,
                                                           float* __restrict__ d_out,
                                                           float* __restrict__ t_out) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= W * H) return;
  const int px = i % W, py = i / W;
  float ox, oy, o[3], d[3], t0, t1;
  spp_offset(spp_k, ox, oy);
  raygen(cam, px, py, ox, oy, o, d);
  ray_aabb(o, d, t0, t1);
  for (int a = 0; a < 3; a++) {
    o_out[i * 3 + a] = o[a];
    d_out[i * 3 + a] = d[a];
  }
  t_out[i * 2] = t0;
  t_out[i * 2 + 1] = t1;
}

// the same hook through the 64-slot kernel's machinery: one wave = 64 points, one lane = one point (encode_sample,
// permlane swaps, mlp_forward2 on the frags64 set)
template <int F, int NDENSE>
__global__ __launch_bounds__(256) void debug_field64_kernel(FieldDev fd, const float* __restrict__ pos, const float* __restrict__ dir,
                                                            int n, uint16_t* __restrict__ feat, float* __restrict__ out36,
                                                            int32_t* __restrict__ occ_out) {
  __shared__ half8 wl[kNumFrags * 64];
  const LevelDev* __restrict__ lvl = fd.levels; // wave-uniform kernel arguments, as in the render kernel
  for (int i = threadIdx.x; i < kNumFrags * 64; i += 256) wl[i] = fd.frags64[i];
  __syncthreads();
  const int lane = threadIdx.x & 63, r = lane & 31, g = lane >> 5;
  const int wave = blockIdx.x * 4 + (threadIdx.x >> 6);
  const int idx = wave * 64 + lane, idxA = wave * 64 + r, idxB = idxA + 32;
  const bool ok = idx < n;
  float p[3] = {0.5f, 0.5f, 0.5f};
  if (ok)
    for (int a = 0; a < 3; a++) p[a] = pos[idx * 3 + a];
  auto dir_of = [&](int i, float dd[3]) {
    dd[0] = 0.f; dd[1] = 0.f; dd[2] = 1.f;
    if (dir && i < n)
      for (int a = 0; a < 3; a++) dd[a] = dir[i * 3 + a];
  };
  float dA[3], dB[3];
  dir_of(idxA, dA);
  dir_of(idxB, dB);
  half8 f[4];
  const HashConsts hc = {fd.hash_my_b, fd.hash_mz_b, fd.hash_m_b, (uint32_t)fd.wide_offsets};
  encode_sample<F, NDENSE>(fd.table, lvl, hc, p[0], p[1], p[2], f);
  if (ok && feat) {
    typedef uint16_t ushort8 __attribute__((ext_vector_type(8)));
    uint16_t* dst = feat + (size_t)idx * 32;
#pragma unroll
    for (int s = 0; s < 4; s++) {
      const ushort8 u = __builtin_bit_cast(ushort8, f[s]);
#pragma unroll
      for (int e = 0; e < 8; e++) dst[8 * s + e] = u[e];
    }
  }
  swap_halves(f[0], f[1]);
  swap_halves(f[2], f[3]);
  const half8 fA[2] = {f[0], f[2]}, fB[2] = {f[1], f[3]};
  const MlpOut2 mo = mlp_forward2(wl, lane, fA, fB, sh_fragment(g, dA[0], dA[1], dA[2]), sh_fragment(g, dB[0], dB[1], dB[2]));
  if (out36) {
    if (ok) { // the lane's own point, read where the render kernel reads it
      float* q = out36 + (size_t)idx * 36;
      q[0] = fast_exp((g ? mo.densB[8] : mo.densA[0]) + fd.density_bias);
      q[1] = fast_sigmoid(g ? mo.rgbB[8] : mo.rgbA[0]);
      q[2] = fast_sigmoid(g ? mo.rgbB[9] : mo.rgbA[1]);
      q[3] = fast_sigmoid(g ? mo.rgbB[10] : mo.rgbA[2]);
    }
#pragma unroll
    for (int i = 0; i < 8; i++) { // raw outputs: register i of lane half g = row (i&3) + 8(i>>2) + 4g of column r, both groups
      const int row = (i & 3) + 8 * (i >> 2) + 4 * g;
      if (idxA < n) {
        out36[(size_t)idxA * 36 + 4 + row] = mo.densA[i];
        out36[(size_t)idxA * 36 + 20 + row] = mo.rgbA[i];
      }
      if (idxB < n) {
        out36[(size_t)idxB * 36 + 4 + row] = mo.densB[i];
        out36[(size_t)idxB * 36 + 20 + row] = mo.rgbB[i];
      }
    }
  }
  if (ok && occ_out) occ_out[idx] = occupied(fd, p[0], p[1], p[2]) ? 1 : 0;
}

// ------------------------------------------------------------------ ground-truth splats
// the PCL screenshot of the coloured cloud + convertToAlpha + 180-degree flip (main.cpp:68-96, 1610-1618;
// Share_Data.hpp:771-784) as a z-buffered square-splat rasteriser: one lane = one point of one view, the
// depth test is a 64-bit atomicMin on (depth bits << 32 | colour), so the result does not depend on order

__device__ __forceinline__ bool splat_project(const CamDev& cam, float scale, const float off[3], const float* p,
                                              float& u, float& v, float& z) {
  const float q[3] = {fmaf(p[0], scale, off[0]), fmaf(p[1], scale, off[1]), fmaf(p[2], scale, off[2])};
  const float e[3] = {q[1], q[2], q[0]};
  const float d[3] = {e[0] - cam.c2w[3], e[1] - cam.c2w[7], e[2] - cam.c2w[11]};
  float c[3];
#pragma unroll
  for (int a = 0; a < 3; a++) c[a] = fmaf(cam.c2w[a], d[0], fmaf(cam.c2w[4 + a], d[1], cam.c2w[8 + a] * d[2]));
  if (!(c[2] > 1e-6f)) return false;
  float x = c[0] / c[2], y = c[1] / c[2];
  if (has_lens(cam)) {
    float xd, yd, J[4];
    lens_eval(cam.lens, x, y, xd, yd, J);
    x = xd;
    y = yd;
  }
  u = fmaf(cam.fx, x, cam.cx);
  v = fmaf(cam.fy, y, cam.cy);
  z = c[2];
  return true;
}

__global__ __launch_bounds__(256) void splat_points_kernel(const float* __restrict__ xyz, const uint8_t* __restrict__ rgb,
                                                           size_t n, float scale, float ox, float oy, float oz,
                                                           const CamDev* __restrict__ cams, int W, int H, int point_size,
                                                           unsigned long long* __restrict__ zbuf) {
  const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  const CamDev cam = cams[blockIdx.y];
  const float off[3] = {ox, oy, oz};
  const float p[3] = {xyz[3 * i], xyz[3 * i + 1], xyz[3 * i + 2]};
  float u, v, z;
  if (!splat_project(cam, scale, off, p, u, v, z)) return;
  const uint32_t col = (uint32_t)rgb[3 * i] | ((uint32_t)rgb[3 * i + 1] << 8) | ((uint32_t)rgb[3 * i + 2] << 16);
  const unsigned long long key = ((unsigned long long)__float_as_uint(z) << 32) | col;
  const int x0 = (int)floorf(u - 0.5f * (float)point_size + 0.5f), y0 = (int)floorf(v - 0.5f * (float)point_size + 0.5f);
  unsigned long long* zb = zbuf + (size_t)blockIdx.y * W * H;
  for (int dy = 0; dy < point_size; dy++)
    for (int dx = 0; dx < point_size; dx++) {
      const int x = x0 + dx, y = y0 + dy;
      if (x < 0 || y < 0 || x >= W || y >= H) continue;
      atomicMin(zb + (size_t)y * W + x, key);
    }
}

__global__ __launch_bounds__(256) void splat_resolve_kernel(const unsigned long long* __restrict__ zbuf, int W, int H,
                                                            int flip180, uint32_t* __restrict__ out) {
  const size_t npix = (size_t)W * H;
  const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= npix) return;
  const unsigned long long k = zbuf[(size_t)blockIdx.y * npix + i];
  uint32_t px = 0x00ffffffu; // white, alpha 0
  if (k != ~0ull) {
    const uint32_t col = (uint32_t)k & 0x00ffffffu;
    px = col | (col == 0x00ffffffu ? 0u : 0xff000000u);
  }
  const int x = (int)(i % W), y = (int)(i / W);
  const size_t o = flip180 ? ((size_t)(H - 1 - y) * W + (size_t)(W - 1 - x)) : i;
  out[(size_t)blockIdx.y * npix + o] = px;
}

// ------------------------------------------------------------------ host-callable launchers

hipError_t launch_splat_points(const float* xyz, const uint8_t* rgb, size_t n, float scale, const float off[3],
                               const CamDev* cams, int n_views, int W, int H, int point_size, int flip180,
                               unsigned long long* zbuf, uint32_t* out, hipStream_t s) {
  hipError_t e = hipMemsetAsync(zbuf, 0xff, (size_t)n_views * W * H * 8, s);
  if (e != hipSuccess) return e;
  if (n > 0)
    hipLaunchKernelGGL(splat_points_kernel, dim3((unsigned)((n + 255) / 256), n_views), dim3(256), 0, s, xyz, rgb, n, scale,
                       off[0], off[1], off[2], cams, W, H, point_size, zbuf);
  hipLaunchKernelGGL(splat_resolve_kernel, dim3((unsigned)(((size_t)W * H + 255) / 256), n_views), dim3(256), 0, s, zbuf, W, H,
                     flip180, out);
  return hipGetLastError();
}

hipError_t launch_spp_reduce(const float* stage, size_t n_pixels, int spp, const float bg[4], float* out, uint32_t* out_u8,
                             hipStream_t s) {
  unsigned blocks = (unsigned)std::min<size_t>(4096, (n_pixels + 255) / 256);
  hipLaunchKernelGGL(spp_reduce_kernel, dim3(blocks), dim3(256), 0, s, reinterpret_cast<const float4*>(stage), n_pixels, spp,
                     1.0f / (float)spp, bg[0], bg[1], bg[2], bg[3], reinterpret_cast<float4*>(out), out_u8);
  return hipGetLastError();
}

hipError_t launch_march(const MarchParams& P, int n_views, int n_spp, hipStream_t s) {
  dim3 grid((unsigned)(P.live_grid ? P.live_tiles_max : P.tiles_x * P.tiles_y), (unsigned)n_views, (unsigned)(P.spp_inner_log2 > 0 ? 1 : n_spp));
  if (grid.x == 0) return hipSuccess; // no view of the batch can see the object
  if (P.step_mode == PRV_STEP_NGP) hipLaunchKernelGGL(march_compact_kernel<true>, grid, dim3(256), 0, s, P);
  else hipLaunchKernelGGL(march_compact_kernel<false>, grid, dim3(256), 0, s, P);
  return hipGetLastError();
}

hipError_t launch_occ_interleave(const OccInterleaveParams& P, hipStream_t s) {
  hipLaunchKernelGGL(occ_interleave_kernel, dim3((P.n_cells + 255u) / 256u), dim3(256), 0, s, P);
  return hipGetLastError();
}

bool march_multi_supported(int n_members) { return n_members == 2 || n_members == 5; }

hipError_t launch_march_multi(const MarchMultiParams& P, int n_views, int n_spp, hipStream_t s) {
  dim3 grid((unsigned)(P.tiles_x * P.tiles_y), (unsigned)n_views, (unsigned)(P.spp_inner_log2 > 0 ? 1 : n_spp));
  if (grid.x == 0) return hipSuccess;
  // the paper's ensembles (Share_Data.hpp:505-510: two members for EnsembleRGB, five for EnsembleRGBDensity)
  if (P.n_members == 5) hipLaunchKernelGGL((march_multi_kernel<5, 8>), grid, dim3(256), 0, s, P); // 24 KB of record staging + 40 KB of mask words
  else if (P.n_members == 2) hipLaunchKernelGGL((march_multi_kernel<2, 12>), grid, dim3(256), 0, s, P);
  else return hipErrorInvalidValue;
  return hipGetLastError();
}

// compiled instances of the render kernel: NDENSE = the field's count of leading dense levels when an instance for
// exactly that count exists (5 / 3 for F = 4, 10 / 6 for F = 2: the BASELINE fields, instant-ngp's base.json, the test
// fields) and its hashed levels share their hash constants; any other field runs the generic instance <F, 0>
int render_instance_dense_levels(const FieldDev& fd) {
  const int n = fd.n_dense_levels;
  if (!fd.hash_shared) return 0;
  if (fd.n_features == 4) return n == 5 ? 5 : n == 3 ? 3 : 0;
  return n == 10 ? 10 : n == 6 ? 6 : 0;
}

template <bool NGP>
static void launch_render_mode(const RenderParams& P, int n_blocks, hipStream_t s) {
  const int nd = render_instance_dense_levels(P.field);
  if (P.cell_cache && nd > 0) { // the caller asked for the per-lane corner cache (prv_api.cpp: render_views says when)
    if (P.field.n_features == 4 && nd == 5) {
      hipLaunchKernelGGL((render_queue64_kernel<4, 5, NGP, true>), dim3(n_blocks), dim3(256), 0, s, P);
      return;
    }
    // (the F = 2 fields have six hashed levels: their cache does not fit 256 registers -- 128 spilled -- so they run without)
  }
  if (P.field.n_features == 4) {
    if (nd == 5) hipLaunchKernelGGL((render_queue64_kernel<4, 5, NGP>), dim3(n_blocks), dim3(256), 0, s, P);
    else if (nd == 3) hipLaunchKernelGGL((render_queue64_kernel<4, 3, NGP>), dim3(n_blocks), dim3(256), 0, s, P);
    else hipLaunchKernelGGL((render_queue64_kernel<4, 0, NGP>), dim3(n_blocks), dim3(256), 0, s, P);
  } else {
    if (nd == 10) hipLaunchKernelGGL((render_queue64_kernel<2, 10, NGP>), dim3(n_blocks), dim3(256), 0, s, P);
    else if (nd == 6) hipLaunchKernelGGL((render_queue64_kernel<2, 6, NGP>), dim3(n_blocks), dim3(256), 0, s, P);
    else hipLaunchKernelGGL((render_queue64_kernel<2, 0, NGP>), dim3(n_blocks), dim3(256), 0, s, P);
  }
}

hipError_t launch_render(const RenderParams& P, int n_blocks, hipStream_t s) {
  if (P.step_mode == PRV_STEP_NGP) launch_render_mode<true>(P, n_blocks, s);
  else launch_render_mode<false>(P, n_blocks, s);
  return hipGetLastError();
}

hipError_t launch_first_hit(const FieldDev& fd, const CamDev* cams, int n_views, int W, int H, float max_range,
                            int32_t* out, hipStream_t s) {
  dim3 grid((unsigned)((W * H + 255) / 256), (unsigned)n_views);
  hipLaunchKernelGGL(first_hit_kernel, grid, dim3(256), 0, s, fd, cams, W, H, max_range, out);
  return hipGetLastError();
}

hipError_t launch_precept(const FieldDev& fd, const float* voxels, int n, const PreceptPose& pose, const Rs2Intr& in,
                          float max_range, int32_t* out, hipStream_t s) {
  hipLaunchKernelGGL(precept_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, fd, voxels, n, pose, in, max_range, out);
  return hipGetLastError();
}

hipError_t launch_quantize(const float* in, size_t n, const float bg[4], uint8_t* out, hipStream_t s) {
  unsigned blocks = (unsigned)((n + 255) / 256);
  if (blocks > 4096) blocks = 4096;
  if (blocks == 0) blocks = 1;
  hipLaunchKernelGGL(quantize_kernel, dim3(blocks), dim3(256), 0, s, reinterpret_cast<const float4*>(in), n,
                     bg[0], bg[1], bg[2], bg[3], reinterpret_cast<uint32_t*>(out));
  return hipGetLastError();
}

hipError_t launch_score_ensemble(const EnsembleParams& P, int method, int n_views, int n_blocks, prv_score_record* rec,
                                 hipStream_t s) {
  dim3 grid((unsigned)n_blocks, (unsigned)n_views);
  const size_t n_terms = P.pixels_per_view * (method == PRV_SCORE_ENSEMBLE_RGB ? 3 : 2);
  if (method == PRV_SCORE_ENSEMBLE_RGB)
    hipLaunchKernelGGL(score_ensemble_terms_kernel<PRV_SCORE_ENSEMBLE_RGB>, grid, dim3(256), 0, s, P);
  else
    hipLaunchKernelGGL(score_ensemble_terms_kernel<PRV_SCORE_ENSEMBLE_RGB_DENSITY>, grid, dim3(256), 0, s, P);
  hipLaunchKernelGGL(score_sequential_sum_kernel, dim3((unsigned)n_views), dim3(64), 0, s, (const double*)P.partial, n_terms,
                     rec + P.view0);
  return hipGetLastError();
}

hipError_t launch_score_psnr(const PsnrParams& P, int n_views, int n_blocks, hipStream_t s) {
  dim3 grid((unsigned)n_blocks, (unsigned)n_views);
  hipLaunchKernelGGL(score_psnr_kernel, grid, dim3(256), 0, s, P);
  return hipGetLastError();
}

hipError_t launch_ssim(const float* img, const float* gt, int n_views, int W, int H, const float bg[4], float* la,
                       float* lb, double* partial, int n_blocks, double* out, hipStream_t s) {
  const size_t n = (size_t)n_views * W * H;
  unsigned blocks = (unsigned)std::min<size_t>(4096, (n + 255) / 256);
  hipLaunchKernelGGL(ssim_lum_kernel, dim3(blocks), dim3(256), 0, s, reinterpret_cast<const float4*>(img),
                     reinterpret_cast<const float4*>(gt), n, bg[0], bg[1], bg[2], bg[3], la, lb);
  hipLaunchKernelGGL(ssim_map_kernel, dim3((unsigned)n_blocks, (unsigned)n_views), dim3(256), 0, s, la, lb, W, H, partial);
  hipLaunchKernelGGL(ssim_finalize_kernel, dim3((unsigned)((n_views + 63) / 64)), dim3(64), 0, s, partial, n_views,
                     n_blocks, (double)(W - 4) * (double)(H - 4), out);
  return hipGetLastError();
}

hipError_t launch_score_finalize(const double* partial, int n_views, int n_blocks, int method,
                                 size_t pixels_per_view, double coverage_weight, prv_score_record* rec, hipStream_t s) {
  hipLaunchKernelGGL(score_finalize_kernel, dim3((unsigned)((n_views + 63) / 64)), dim3(64), 0, s, partial,
                     n_views, n_blocks, method, pixels_per_view, coverage_weight, rec);
  return hipGetLastError();
}

hipError_t launch_synth_table(uint16_t* table, size_t n, uint64_t seed, float amp, hipStream_t s) {
  hipLaunchKernelGGL(synth_table_kernel, dim3(2048), dim3(256), 0, s, table, n, (unsigned long long)seed, amp);
  return hipGetLastError();
}

hipError_t launch_repack_level(const uint16_t* canon, uint16_t* phys, const RepackLevel& L, int F, hipStream_t s) {
  unsigned blocks = ((L.hashed ? L.n : L.res * L.res * L.res) + 255) / 256;
  if (blocks > 8192) blocks = 8192;
  if (blocks == 0) return hipSuccess;
  if (F == 4)
    hipLaunchKernelGGL(repack_level_kernel<4>, dim3(blocks), dim3(256), 0, s, canon, phys, L);
  else
    hipLaunchKernelGGL(repack_level_kernel<2>, dim3(blocks), dim3(256), 0, s, canon, phys, L);
  return hipGetLastError();
}

hipError_t launch_debug_raygen(const CamDev& cam, int W, int H, int spp_k, float* o, float* d, float* t,
                               hipStream_t s) {
  hipLaunchKernelGGL(debug_raygen_kernel, dim3((unsigned)((W * H + 255) / 256)), dim3(256), 0, s, cam, W, H,
                     spp_k, o, d, t);
  return hipGetLastError();
}

hipError_t launch_debug_field(const FieldDev& fd, const float* pos, const float* dir, int n, uint16_t* feat,
                              float* out36, int32_t* occ, hipStream_t s) {
  const unsigned blocks64 = (unsigned)((n + 255) / 256);
  const int nd = render_instance_dense_levels(fd);
  if (fd.n_features == 4) {
    if (nd == 5) hipLaunchKernelGGL((debug_field64_kernel<4, 5>), dim3(blocks64), dim3(256), 0, s, fd, pos, dir, n, feat, out36, occ);
    else if (nd == 3) hipLaunchKernelGGL((debug_field64_kernel<4, 3>), dim3(blocks64), dim3(256), 0, s, fd, pos, dir, n, feat, out36, occ);
    else hipLaunchKernelGGL((debug_field64_kernel<4, 0>), dim3(blocks64), dim3(256), 0, s, fd, pos, dir, n, feat, out36, occ);
  } else {
    if (nd == 10) hipLaunchKernelGGL((debug_field64_kernel<2, 10>), dim3(blocks64), dim3(256), 0, s, fd, pos, dir, n, feat, out36, occ);
    else if (nd == 6) hipLaunchKernelGGL((debug_field64_kernel<2, 6>), dim3(blocks64), dim3(256), 0, s, fd, pos, dir, n, feat, out36, occ);
    else hipLaunchKernelGGL((debug_field64_kernel<2, 0>), dim3(blocks64), dim3(256), 0, s, fd, pos, dir, n, feat, out36, occ);
  }
  return hipGetLastError();
}

} // namespace prv
