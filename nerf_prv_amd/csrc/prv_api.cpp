// prv_api.cpp -- the C ABI of include/prv.h on top of the gfx950 kernels.
// Host-side C++ compiled by hipcc; no CPU fallback anywhere: every compute entry point
// needs a context, and a context needs a GPU.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <memory>
#include <new>
#include <stdexcept>
#include <string>
#include <vector>

#include "prv_json.hpp"
#include "prv_kernels.hpp"
#include "prv_levels.hpp"
#include "prv_ingp.hpp"
#include "prv_train.hpp"

using namespace prv;

// ------------------------------------------------------------------ internals

namespace {

thread_local std::string g_create_error;
std::atomic<int> g_live_contexts{0};        // prv_runtime_shutdown refuses while a context is alive
std::atomic<unsigned long long> g_devices_used{0}; // bit d: a context was created on device d
std::atomic<bool> g_rccl_loaded{false};            // librccl was dlopen'ed by a communicator: it stays loaded for the process

struct Buffer {
  void* p = nullptr;
  size_t bytes = 0;
};

struct Model {
  bool loaded = false;
  bool dirty = false; // a trainer changed the canonical arrays: the derived render state is rebuilt on first use
  uint64_t generation = 0; // bumped whenever NEW parameters are installed (load / synthetic / fresh / file), not by a trainer's publish
  prv_field_desc desc{};
  FieldDev dev{};
  uint64_t table_halfs = 0, occ_words = 0;
  float occ_lo[3] = {0, 0, 0}, occ_hi[3] = {1, 1, 1};
  Buffer table, phys, occ, occ_coarse, frags, frags64, mlp; // table/mlp = canonical (ABI) copies kept for export; phys = kernel layout
};

} // namespace

struct prv_camset {
  std::vector<CamDev> cams;
  CamDev* dev = nullptr;
  int device = 0;
  int width = 0, height = 0;
  bool dataset = false; // intrinsics are the dataset's (own principal point, fl_y, lens), not fov-at-centre
};

static void train_detach_all(struct prv_ctx* c);
static void comm_detach_all(struct prv_ctx* c);

struct prv_ctx {
  int device = 0;
  int n_cu = 256;
  hipStream_t own_stream = nullptr;
  hipStream_t stream = nullptr;
  std::vector<hipStream_t> idle_queues; // streams with a hardware queue of their own, parked by destroyed trainers (prv_train_api.inc)
  hipStream_t capture_stream = nullptr; // the trainers' step graphs are captured on it (non-blocking; nothing ever runs on it)
  std::vector<Buffer> idle_buffers;     // ... and their device buffers, taken again by size (train_buffer)
  size_t idle_bytes = 0;
  std::string err;
  Model models[PRV_MAX_SLOTS];
  // grow-only workspaces
  Buffer queue, queue_ext, stage, counters, view_ids, img_f32, partial, records, dbg[6];
  Buffer counters_multi, occ_multi; // the ensemble's one-launch march: queue heads + counts per member, the interleaved occupancy bytes
  int march_multi = -1;             // PRV_MARCH_MULTI=0/1 (-1: on where an instance exists, render_ensemble_ngp)
  Buffer img_u8[PRV_MAX_MODELS];
  bool profiling = false;
  std::vector<hipEvent_t> ev_render, ev_march; // start/stop pairs of the current profiling window
  std::vector<hipEvent_t> ev_free;             // recycled events: a profiling window creates none once the pool is warm
  std::vector<float> last_render_ms;           // the render launches of the last closed window, one duration each (prv_profile_render_launches)
  void* pin = nullptr;                         // pinned host staging of the per-call camera upload (no pageable copy, no stream sync)
  size_t pin_cap = 0;
  hipEvent_t pin_ev = nullptr; // recorded after the upload: the staging is rewritten only once that copy has run
  int blocks_per_cu = -1; // persistent render blocks per CU (PRV_BLOCKS_PER_CU); -1 = by table and image size, see render_views
  int cell_cache = -1;    // render_queue64 per-lane corner cache (PRV_CELL_CACHE=0/1; -1 = by stepping rule and image size, see render_views)
  std::vector<struct prv_trainer*> trainers; // live trainers of this context (detached by prv_destroy)
  std::vector<struct prv_comm*> comms;       // live communicators of this context (detached by prv_destroy)
  int queue_segments = 8; // ray-queue segments = XCDs (PRV_QUEUE_SEGMENTS: 1 = single shared head)
  int spatial_regions = 1; // a wave's records go to the region of its first live ray's octant (PRV_SPATIAL_REGIONS=0: block id % regions, rounds 1-5)
  int pool_on = -1;       // render_queue64 block-level tail pool (PRV_POOL=0/1; -1 = by table and image size, see render_views)
  int merge_max = -1;     // render_queue64 tail merge threshold (PRV_MERGE_MAX; 0 = off; -1 = by table size, see render_views)
  size_t stage_budget = (size_t)4 << 30; // staging bytes for multi-sample renders (spp x batch x image)
  size_t queue_budget = (size_t)8 << 30; // ray-queue bytes per batch of views (288 GB of HBM): one batch for 64 views at 800x800 and for the
                                         // reference's whole candidate set under the engine's rule (540 x 80x45 x 16 spp x 208 B = 6.5 GB:
                                         // one launch pair per member instead of two, -3 % per round)
  double coverage_weight = PRV_COVERAGE_WEIGHT_DEFAULT; // method 5: score = -PSNR + weight * mean((1 - alpha)^2)
};

namespace {

int fail(prv_ctx* c, int code, const char* fmt, ...) {
  char buf[512];
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(buf, sizeof(buf), fmt, ap);
  va_end(ap);
  if (c) c->err = buf;
  else g_create_error = buf;
  return code;
}

// No exception crosses the C ABI: every entry point that can allocate is a function-try-block ending here.
int caught(prv_ctx* c) noexcept {
  try {
    throw;
  } catch (const std::bad_alloc&) {
    try {
      return fail(c, PRV_E_INTERNAL, "out of host memory");
    } catch (...) {
      return PRV_E_INTERNAL;
    }
  } catch (const std::exception& e) {
    try {
      return fail(c, PRV_E_INTERNAL, "internal error: %s", e.what());
    } catch (...) {
      return PRV_E_INTERNAL;
    }
  } catch (...) {
    return PRV_E_INTERNAL;
  }
}

#define HIPCHK(c, expr)                                                                           \
  do {                                                                                            \
    hipError_t e__ = (expr);                                                                      \
    if (e__ != hipSuccess) return fail((c), PRV_E_HIP, "%s failed: %s", #expr, hipGetErrorString(e__)); \
  } while (0)

void release(Buffer& b);
int ensure(prv_ctx* c, Buffer& b, size_t bytes) {
  if (b.bytes >= bytes && b.p) return PRV_OK;
  if (b.p) {
    HIPCHK(c, hipStreamSynchronize(c->stream));
    HIPCHK(c, hipFree(b.p));
    b.p = nullptr;
    b.bytes = 0;
  }
  hipError_t e = hipMalloc(&b.p, bytes);
  if (e == hipErrorOutOfMemory && !c->idle_buffers.empty()) {
    // the context's own parked memory (destroyed trainers' buffers, up to 16 GiB: prv_train_api.inc) goes before anybody fails
    (void)hipGetLastError();
    HIPCHK(c, hipStreamSynchronize(c->stream));
    for (Buffer& idle : c->idle_buffers) release(idle);
    c->idle_buffers.clear();
    c->idle_bytes = 0;
    e = hipMalloc(&b.p, bytes);
  }
  if (e != hipSuccess) b.p = nullptr;
  HIPCHK(c, e);
  b.bytes = bytes;
  return PRV_OK;
}

void release(Buffer& b) {
  if (b.p) (void)hipFree(b.p);
  b.p = nullptr;
  b.bytes = 0;
}

static_assert(kMaxFieldLevels == kMaxLevels, "level tables of the host and the kernels must agree");

// ---- MFMA A-fragment prepack.  Canonical weights W[layer][in][out] (fp16 bits).
// Fragment (layer, mt, s): lane (r,h), element j = W[kmap(s,h,j)][32*mt + r], zero past n_out.
const int kIn[5] = {32, 64, 32, 64, 64};
const int kOut[5] = {64, 16, 64, 64, 16};
const int kOff[5] = {0, 2048, 3072, 5120, 9216};

// hidden unit held by element j of lane half h in k-step s of a 64-wide activation
inline int hidden_k(int s, int h, int j) { return 32 * (s >> 1) + 16 * (s & 1) + 8 * (j >> 2) + 4 * h + (j & 3); }

// v64 = the fragment set of render_queue64: first-layer k rows in canonical feature order (the lane gathers every
// level of its own sample), and the units the compositing needs repeated in padding rows that land in lane half 1:
// density layer 2's unit 0 at row 20 (register 8), colour layer 3's units 0..2 at rows 20..22 (registers 8..10).
void prepack_fragments(const uint16_t* mlp, int n_features, std::vector<uint16_t>& frags, bool v64 = false) {
  frags.assign((size_t)kNumFrags * kFragHalfs, 0);
  int f = 0;
  auto emit = [&](int layer, int mt, int s, int (*kmap)(int, int, int)) {
    uint16_t* dst = frags.data() + (size_t)f * kFragHalfs;
    for (int lane = 0; lane < 64; lane++) {
      const int r = lane & 31, h = lane >> 5;
      int out = 32 * mt + r;
      if (v64 && (layer == 1 || layer == 4) && out >= 20 && out < 20 + (layer == 1 ? 1 : 3)) out -= 20;
      for (int j = 0; j < 8; j++) {
        const int k = kmap(s, h, j);
        dst[lane * 8 + j] = out < kOut[layer] ? mlp[kOff[layer] + k * kOut[layer] + out] : (uint16_t)0;
      }
    }
    f++;
  };
  // grid features: fragment s, element j of lane half h = feature (j % F) of level 2*(s*LH/2 + j/F) + h
  auto k_feat4 = [](int s, int h, int j) { return 4 * (2 * (s * 2 + j / 4) + h) + j % 4; };
  auto k_feat2 = [](int s, int h, int j) { return 2 * (2 * (s * 4 + j / 2) + h) + j % 2; };
  auto k_rgb_in = [](int s, int h, int j) { return s == 0 ? hidden_k(0, h, j) : 16 + 8 * h + j; }; // [dens | SH]
  auto k_canon = [](int s, int h, int j) { return 16 * s + 8 * h + j; };
  for (int mt = 0; mt < 2; mt++)
    for (int s = 0; s < 2; s++)
      emit(0, mt, s, v64 ? (int (*)(int, int, int))k_canon : n_features == 4 ? (int (*)(int, int, int))k_feat4 : (int (*)(int, int, int))k_feat2);
  for (int s = 0; s < 4; s++) emit(1, 0, s, hidden_k);
  for (int mt = 0; mt < 2; mt++)
    for (int s = 0; s < 2; s++) emit(2, mt, s, k_rgb_in);
  for (int mt = 0; mt < 2; mt++)
    for (int s = 0; s < 4; s++) emit(3, mt, s, hidden_k);
  for (int s = 0; s < 4; s++) emit(4, 0, s, hidden_k);
}

// ---- host fp16 + counter RNG for the (tiny) synthetic MLP weights and occupancy
uint16_t f2h(float f) {
  _Float16 h = (_Float16)f; // host clang: IEEE RNE conversion
  uint16_t u;
  memcpy(&u, &h, 2);
  return u;
}
uint64_t mix64(uint64_t z) {
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
  return z ^ (z >> 31);
}
float rng_sym(uint64_t seed, uint64_t stream, uint64_t i, float amp) {
  const uint64_t hsh = mix64(seed + (stream + 1) * 0xD1B54A32D192ED03ull + i * 0x9E3779B97F4A7C15ull);
  const uint32_t u = (uint32_t)(hsh >> 40);
  const float v = (float)u * (1.0f / 8388608.0f) - 1.0f;
  return v * amp;
}
const float kSynthSpheres[4][4] = {
    {0.50f, 0.50f, 0.50f, 0.35f}, {0.80f, 0.50f, 0.62f, 0.13f}, {0.36f, 0.80f, 0.45f, 0.11f}, {0.40f, 0.24f, 0.78f, 0.10f}};

int install_model(prv_ctx* c, int slot, const prv_field_desc& d, const uint16_t* mlp, const uint32_t* occ_host,
                  const uint16_t* table_host /* may be null: table already on device */) {
  Model& m = c->models[slot];
  HostLevel lv[kMaxLevels];
  uint64_t total = 0;
  if (compute_levels(d, lv, &total) != 0) return fail(c, PRV_E_INVALID, "invalid field descriptor");
  m.loaded = false; // until everything below has succeeded: a failed install leaves an empty slot, not half a model
  m.desc = d;
  m.table_halfs = total * (uint64_t)d.n_features;
  const uint64_t R = (uint64_t)d.occ_res;
  m.occ_words = (R * R * R + 31) / 32;
  int rc;
  if ((rc = ensure(c, m.table, m.table_halfs * 2)) != PRV_OK) return rc;
  if ((rc = ensure(c, m.occ, m.occ_words * 4)) != PRV_OK) return rc;
  if ((rc = ensure(c, m.frags, (size_t)kNumFrags * kFragHalfs * 2)) != PRV_OK) return rc;
  if ((rc = ensure(c, m.frags64, (size_t)kNumFrags * kFragHalfs * 2)) != PRV_OK) return rc;
  if ((rc = ensure(c, m.mlp, PRV_MLP_HALFS * 2)) != PRV_OK) return rc;
  if (table_host) HIPCHK(c, hipMemcpyAsync(m.table.p, table_host, m.table_halfs * 2, hipMemcpyHostToDevice, c->stream));
  HIPCHK(c, hipMemcpyAsync(m.occ.p, occ_host, m.occ_words * 4, hipMemcpyHostToDevice, c->stream));
  // bounding box of the occupied cells (grown by one cell) for the march pass
  {
    const int R = d.occ_res;
    int lo[3] = {R, R, R}, hi[3] = {-1, -1, -1};
    const bool words = R % 32 == 0; // a row is whole words: 65 K words instead of 2 M bits (the planner loop installs five fields per round)
    for (int z = 0; words && z < R; z++)
      for (int y = 0; y < R; y++)
        for (int w = 0; w < R / 32; w++) {
          const uint32_t v = occ_host[((size_t)R * ((size_t)y + (size_t)R * (size_t)z)) / 32 + (size_t)w];
          if (!v) continue;
          lo[0] = std::min(lo[0], 32 * w + __builtin_ctz(v));
          hi[0] = std::max(hi[0], 32 * w + 31 - __builtin_clz(v));
          lo[1] = std::min(lo[1], y);
          hi[1] = std::max(hi[1], y);
          lo[2] = std::min(lo[2], z);
          hi[2] = std::max(hi[2], z);
        }
    for (int z = 0; !words && z < R; z++)
      for (int y = 0; y < R; y++)
        for (int x = 0; x < R; x++) {
          const size_t b = (size_t)x + (size_t)R * ((size_t)y + (size_t)R * (size_t)z);
          if (!((occ_host[b >> 5] >> (b & 31)) & 1u)) continue;
          const int cxyz[3] = {x, y, z};
          for (int a = 0; a < 3; a++) {
            lo[a] = std::min(lo[a], cxyz[a]);
            hi[a] = std::max(hi[a], cxyz[a]);
          }
        }
    for (int a = 0; a < 3; a++) {
      m.occ_lo[a] = hi[a] < 0 ? 2.0f : (float)(lo[a] - 1) / (float)R; // empty grid: an empty box
      m.occ_hi[a] = hi[a] < 0 ? -1.0f : (float)(hi[a] + 2) / (float)R;
    }
  }
  // dilated coarse occupancy for the march pass: coarse cell = 4^3 fine cells; bit set when any fine
  // cell of the block or of its 26 neighbour blocks is occupied
  std::vector<uint32_t> coarse;
  if (d.occ_res % 4 == 0 && d.occ_res >= 8) {
    const int R = d.occ_res, Rc = R / 4;
    std::vector<uint8_t> blk((size_t)Rc * Rc * Rc, 0);
    const bool words = R % 32 == 0; // as above: a word is eight blocks' worth of one row
    for (int z = 0; words && z < R; z++)
      for (int y = 0; y < R; y++)
        for (int w = 0; w < R / 32; w++) {
          const uint32_t v = occ_host[((size_t)R * ((size_t)y + (size_t)R * (size_t)z)) / 32 + (size_t)w];
          if (!v) continue;
          uint8_t* row = &blk[(size_t)(8 * w) + (size_t)Rc * ((size_t)(y / 4) + (size_t)Rc * (size_t)(z / 4))];
          for (int k = 0; k < 8; k++)
            if ((v >> (4 * k)) & 0xfu) row[k] = 1;
        }
    for (int z = 0; !words && z < R; z++)
      for (int y = 0; y < R; y++)
        for (int x = 0; x < R; x += 32) { // 32 fine cells of a row = one word (R is a multiple of 4; handle R < 32 too)
          const size_t bit0 = (size_t)x + (size_t)R * ((size_t)y + (size_t)R * (size_t)z);
          for (int k = 0; k < 32 && x + k < R; k++) {
            const size_t b = bit0 + k;
            if ((occ_host[b >> 5] >> (b & 31)) & 1u) blk[(size_t)((x + k) / 4) + (size_t)Rc * ((size_t)(y / 4) + (size_t)Rc * (size_t)(z / 4))] = 1;
          }
        }
    coarse.assign(((size_t)Rc * Rc * Rc + 31) / 32, 0u);
    for (int z = 0; z < Rc; z++)
      for (int y = 0; y < Rc; y++)
        for (int x = 0; x < Rc; x++) {
          bool any = false;
          for (int dz = -1; dz <= 1 && !any; dz++)
            for (int dy = -1; dy <= 1 && !any; dy++)
              for (int dx = -1; dx <= 1 && !any; dx++) {
                const int xx = x + dx, yy = y + dy, zz = z + dz;
                if (xx < 0 || yy < 0 || zz < 0 || xx >= Rc || yy >= Rc || zz >= Rc) continue;
                any = blk[(size_t)xx + (size_t)Rc * ((size_t)yy + (size_t)Rc * (size_t)zz)] != 0;
              }
          if (any) {
            const size_t b = (size_t)x + (size_t)Rc * ((size_t)y + (size_t)Rc * (size_t)z);
            coarse[b >> 5] |= 1u << (b & 31);
          }
        }
    if ((rc = ensure(c, m.occ_coarse, coarse.size() * 4)) != PRV_OK) return rc;
    HIPCHK(c, hipMemcpyAsync(m.occ_coarse.p, coarse.data(), coarse.size() * 4, hipMemcpyHostToDevice, c->stream));
  }
  std::vector<uint16_t> frags, frags64;
  prepack_fragments(mlp, d.n_features, frags);
  prepack_fragments(mlp, d.n_features, frags64, true);
  HIPCHK(c, hipMemcpyAsync(m.frags.p, frags.data(), frags.size() * 2, hipMemcpyHostToDevice, c->stream));
  HIPCHK(c, hipMemcpyAsync(m.frags64.p, frags64.data(), frags64.size() * 2, hipMemcpyHostToDevice, c->stream));
  HIPCHK(c, hipMemcpyAsync(m.mlp.p, mlp, PRV_MLP_HALFS * 2, hipMemcpyHostToDevice, c->stream));
  HIPCHK(c, hipStreamSynchronize(c->stream)); // host staging vectors go out of scope
  // ---- physical layout: power-of-two strides for dense levels, size-aligned level offsets
  const uint32_t ebytes = (uint32_t)d.n_features * 2u;
  uint32_t psize[kMaxLevels], sx[kMaxLevels], poff[kMaxLevels];
  int order[kMaxLevels];
  auto ceil_log2 = [](uint32_t v) { uint32_t s = 0; while ((1u << s) < v) s++; return s; };
  for (int l = 0; l < d.n_levels; l++) {
    order[l] = l;
    if (lv[l].hashed) {
      sx[l] = 0;
      psize[l] = lv[l].size;
    } else {
      sx[l] = ceil_log2(lv[l].res + 1); // +1: room for the duplicated border vertex / row / plane
      psize[l] = 1u << ceil_log2((lv[l].res + 1u) << (2 * sx[l])); // res + 1 planes: the last one is duplicated
    }
  }
  std::stable_sort(order, order + d.n_levels, [&](int a, int b) { return psize[a] > psize[b]; });
  uint64_t ptotal = 0;
  for (int k = 0; k < d.n_levels; k++) {
    poff[order[k]] = (uint32_t)ptotal; // multiple of every later (smaller or equal) power of two
    ptotal += psize[order[k]];
  }
  if (ptotal * ebytes >= (1ull << 32)) return fail(c, PRV_E_INVALID, "field too large for 32-bit gather offsets");
  if ((rc = ensure(c, m.phys, ptotal * ebytes)) != PRV_OK) return rc;
  HIPCHK(c, hipMemsetAsync(m.phys.p, 0, ptotal * ebytes, c->stream));
  for (int l = 0; l < d.n_levels; l++) {
    RepackLevel R{lv[l].offset, poff[l], lv[l].size, lv[l].res, sx[l], lv[l].hashed};
    HIPCHK(c, launch_repack_level((const uint16_t*)m.table.p, (uint16_t*)m.phys.p, R, d.n_features, c->stream));
  }
  HIPCHK(c, hipStreamSynchronize(c->stream));
  FieldDev& f = m.dev;
  memset(&f, 0, sizeof(f));
  f.table = (const uint16_t*)m.phys.p;
  f.occ = (const uint32_t*)m.occ.p;
  f.occ_coarse = (d.occ_res % 4 == 0 && d.occ_res >= 8) ? (const uint32_t*)m.occ_coarse.p : nullptr;
  f.frags = (const half8*)m.frags.p;
  f.frags64 = (const half8*)m.frags64.p;
  f.n_levels = d.n_levels;
  f.n_features = d.n_features;
  f.occ_res = d.occ_res;
  f.density_bias = d.density_bias;
  for (int a = 0; a < 3; a++) {
    f.occ_lo[a] = m.occ_lo[a];
    f.occ_hi[a] = m.occ_hi[a];
  }
  f.n_dense_levels = 0;
  while (f.n_dense_levels < d.n_levels && !lv[f.n_dense_levels].hashed) f.n_dense_levels++;
  if (getenv("PRV_NO_PAIR")) f.n_dense_levels = 0; // tests: every level through the generic (clamped, 8-load) gather
  // the hashed levels' shared constants (every hashed level has T entries of ebytes): valid for the kernel's shared path
  // when no hashed level is finer than its table, so that (x << esh) needs no mask
  f.hash_my_b = (2654435761u * ebytes) & 0xffffffu;
  f.hash_mz_b = (805459861u * ebytes) & 0xffffffu;
  f.hash_m_b = (uint32_t)(((1ull << d.log2_hashmap) - 1ull) * ebytes);
  f.hash_shared = 1;
  f.wide_offsets = 0;
  for (int l = 0; l < d.n_levels; l++)
    if (lv[l].hashed && (lv[l].res > lv[l].size || l < f.n_dense_levels)) f.hash_shared = 0;
  if (getenv("PRV_NO_PAIR")) f.hash_shared = 0;
  for (int l = 0; l < d.n_levels; l++) {
    LevelDev& L = f.levels[l];
    L.scale = lv[l].scale;
    L.res_m1 = lv[l].res - 1;
    if (lv[l].res > 4096) return fail(c, PRV_E_INVALID, "level %d has %u vertices per axis (limit 4096)", l, lv[l].res);
    if (lv[l].hashed) {
      // the kernels multiply with v_mul_u32_u24 (the low 24 bits of the constants are all a level of <= 16 MiB needs).  A hashed
      // level LARGER than 16 MiB (round 6: tables that really leave the 256 MiB Infinity Cache, e.g. log2_hashmap 24 at F = 2)
      // keeps the full 32-bit constants and runs the generic instance, whose gather then multiplies in 32 bits (wide_offsets)
      if ((uint64_t)(lv[l].size - 1u) * ebytes >= (1ull << 32)) return fail(c, PRV_E_INVALID, "hashed level of 4 GiB or more");
      L.m_b = (lv[l].size - 1u) * ebytes;
      const bool wide = L.m_b >= (1u << 24);
      L.my_b = wide ? 2654435761u * ebytes : (2654435761u * ebytes) & 0xffffffu;
      L.mz_b = wide ? 805459861u * ebytes : (805459861u * ebytes) & 0xffffffu;
      if (wide) {
        f.wide_offsets = 1;
        f.hash_shared = 0;
      }
    } else {
      L.my_b = ebytes << sx[l];
      L.mz_b = ebytes << (2 * sx[l]);
      L.m_b = 0xffffffffu;
      if (L.mz_b >= (1u << 24) || lv[l].res > 4096) return fail(c, PRV_E_INVALID, "dense level too large");
    }
    L.off_b = poff[l] * ebytes;
    L.myz_b = L.my_b + L.mz_b;
    // the render kernel's packed word (prv_device.hpp: LevelDev::pack); the alignment the packing relies on is checked
    // (a field whose offsets leave no room for it -- hashed tables below 4 KiB -- runs the generic instance, which reads the
    // unpacked constants)
    if (lv[l].hashed) {
      if ((L.off_b & 4095u) != 0u || L.res_m1 > 4095u) f.hash_shared = 0;
      L.pack = (L.off_b & ~4095u) | (L.res_m1 & 4095u);
    } else {
      if ((L.off_b & 31u) != 0u || sx[l] > 31u) f.hash_shared = 0;
      L.pack = (L.off_b & ~31u) | (sx[l] & 31u);
    }
  }
  m.loaded = true;
  m.dirty = false;
  return PRV_OK;
}

// derived state from the canonical arrays that already sit in the model's device buffers
int republish_model(prv_ctx* c, int slot) {
  Model& m = c->models[slot];
  HIPCHK(c, hipSetDevice(c->device));
  std::vector<uint16_t> mlp(PRV_MLP_HALFS);
  std::vector<uint32_t> occ(m.occ_words);
  HIPCHK(c, hipMemcpyAsync(mlp.data(), m.mlp.p, PRV_MLP_HALFS * 2, hipMemcpyDeviceToHost, c->stream));
  HIPCHK(c, hipMemcpyAsync(occ.data(), m.occ.p, m.occ_words * 4, hipMemcpyDeviceToHost, c->stream));
  HIPCHK(c, hipStreamSynchronize(c->stream));
  const prv_field_desc d = m.desc;
  return install_model(c, slot, d, mlp.data(), occ.data(), nullptr);
}

int model_present(prv_ctx* c, int slot) {
  if (slot < 0 || slot >= PRV_MAX_SLOTS) return fail(c, PRV_E_INVALID, "model slot %d out of range", slot);
  if (!c->models[slot].loaded) return fail(c, PRV_E_STATE, "model slot %d is empty", slot);
  return PRV_OK;
}

int republish_model(prv_ctx* c, int slot);

// every render / score / export entry point comes through here: a slot a trainer has stepped since its last
// use gets its derived state (physical table, MFMA fragments, coarse occupancy, bounding box) rebuilt first
int check_model(prv_ctx* c, int slot) {
  int rc = model_present(c, slot);
  if (rc != PRV_OK) return rc;
  if (c->models[slot].dirty) return republish_model(c, slot);
  return PRV_OK;
}

int check_opts(prv_ctx* c, const prv_render_opts* o) {
  if (!o) return fail(c, PRV_E_INVALID, "render options are NULL");
  if (o->width < 1 || o->height < 1 || o->width > 16384 || o->height > 16384)
    return fail(c, PRV_E_INVALID, "bad image size %dx%d", o->width, o->height);
  if (o->step_mode != PRV_STEP_FIXED_S && o->step_mode != PRV_STEP_NGP)
    return fail(c, PRV_E_INVALID, "step_mode must be PRV_STEP_FIXED_S (0) or PRV_STEP_NGP (1), got %d", o->step_mode);
  if (o->step_mode == PRV_STEP_FIXED_S && (o->samples_per_ray < 1 || o->samples_per_ray > kMaxSamples))
    return fail(c, PRV_E_INVALID, "samples_per_ray must be in [1,%d], got %d", kMaxSamples, o->samples_per_ray);
  if (o->spp < 1 || o->spp > 1024) return fail(c, PRV_E_INVALID, "spp must be in [1,1024], got %d", o->spp);
  return PRV_OK;
}

// "device pointer" arguments are validated instead of trusted: a host pointer handed to a kernel is a GPU page
// fault that takes the process down; here it is an error code and a message
int check_device_ptr(prv_ctx* c, const void* p, const char* what) {
  if (!p) return PRV_OK; // NULL-ness is each entry point's own business
  hipPointerAttribute_t a;
  if (hipPointerGetAttributes(&a, p) != hipSuccess) {
    (void)hipGetLastError();
    return fail(c, PRV_E_INVALID, "%s is not a device pointer", what);
  }
  if (a.type != hipMemoryTypeDevice && a.type != hipMemoryTypeManaged && a.type != hipMemoryTypeUnified)
    return fail(c, PRV_E_INVALID, "%s is not a device pointer", what);
  return PRV_OK;
}

// events of the per-kernel timing are recycled: creating (and destroying) four per step made the runtime grow its
// signal pool in the middle of a timed region now and then (one 50-90 ms stall in a few hundred steps)
hipError_t take_event(prv_ctx* c, hipEvent_t* e) {
  if (!c->ev_free.empty()) {
    *e = c->ev_free.back();
    c->ev_free.pop_back();
    return hipSuccess;
  }
  return hipEventCreate(e);
}

// cameras at the render resolution: focal from camera_angle_x at the json width (run.py:285-286,
// fov_axis = 0), rescaled to the requested width; principal point at the image centre.
// Dataset sets (prv_cameras_from_dataset_json) keep their principal point and scale per axis.
CamDev cam_at(const prv_camset* cs, int i, int w, int h) {
  const CamDev& c = cs->cams[i];
  CamDev r = c;
  const float s = (float)w / (float)cs->width;
  if (cs->dataset) {
    const float sy = (float)h / (float)cs->height;
    r.fx = c.fx * s;
    r.fy = c.fy * sy;
    r.cx = c.cx * s;
    r.cy = c.cy * sy;
    return r;
  }
  r.fx = c.fx * s;
  r.fy = c.fy * s;
  r.cx = 0.5f * (float)w;
  r.cy = 0.5f * (float)h;
  return r;
}

// The pixel rectangle that contains every ray of a pinhole view that can meet the occupied cells' box (grown like the march
// pass's own rejection test): a ray through pixel p meets a convex box only if p lies in the box's projection, which lies
// in the bounding rectangle of its eight projected corners -- provided all eight are in front of the camera.  The march
// pass skips whole 256-pixel tiles outside it before any per-ray work (most of an 800x800 view of the bench scene).
struct OccBox { // the bounding box of a model's occupied cells (Model::occ_lo / occ_hi), or of several models'
  float occ_lo[3], occ_hi[3];
};
void set_cull_rect(CamDev& cam, const OccBox& m, int W, int H) {
  cam.cull[0] = cam.cull[1] = cam.cull[2] = cam.cull[3] = 0; // not set
  if (cam.lens[0] != 0.f || cam.lens[1] != 0.f || cam.lens[2] != 0.f || cam.lens[3] != 0.f) return;
  if (!(m.occ_hi[0] > m.occ_lo[0])) { // empty occupancy grid: every tile is dead
    cam.cull[2] = 1; // x1 > 0 marks "set"; the rectangle [0, 1) x [0, 0) contains no pixel
    return;
  }
  const double o[3] = {cam.c2w[3], cam.c2w[7], cam.c2w[11]};
  const double far = std::max(std::max(std::fabs(o[0] - 0.5), std::fabs(o[1] - 0.5)), std::fabs(o[2] - 0.5));
  const double grow = 1e-3 + 1e-5 * far + 1e-4; // the march pass's own margin and a little more
  // camera coordinates of a point p: M^-1 (p - o) with M = the 3x3 of c2w (rays are o + t M (dx, dy, 1)); a general
  // inverse, not the transpose: the matrices come from a json and need not be exact rotations
  const double M[3][3] = {{cam.c2w[0], cam.c2w[1], cam.c2w[2]}, {cam.c2w[4], cam.c2w[5], cam.c2w[6]}, {cam.c2w[8], cam.c2w[9], cam.c2w[10]}};
  const double det = M[0][0] * (M[1][1] * M[2][2] - M[1][2] * M[2][1]) - M[0][1] * (M[1][0] * M[2][2] - M[1][2] * M[2][0]) +
                     M[0][2] * (M[1][0] * M[2][1] - M[1][1] * M[2][0]);
  if (!(std::fabs(det) > 1e-12)) return;
  double Mi[3][3];
  for (int r = 0; r < 3; r++)
    for (int q = 0; q < 3; q++) {
      const int r1 = (r + 1) % 3, r2 = (r + 2) % 3, q1 = (q + 1) % 3, q2 = (q + 2) % 3;
      Mi[q][r] = (M[r1][q1] * M[r2][q2] - M[r1][q2] * M[r2][q1]) / det; // adjugate, transposed
    }
  double u0 = 1e300, u1 = -1e300, v0 = 1e300, v1 = -1e300;
  for (int k = 0; k < 8; k++) {
    const double p[3] = {((k & 1) ? m.occ_hi[0] + grow : m.occ_lo[0] - grow) - o[0], ((k & 2) ? m.occ_hi[1] + grow : m.occ_lo[1] - grow) - o[1],
                         ((k & 4) ? m.occ_hi[2] + grow : m.occ_lo[2] - grow) - o[2]};
    double c[3];
    for (int a = 0; a < 3; a++) c[a] = Mi[a][0] * p[0] + Mi[a][1] * p[1] + Mi[a][2] * p[2];
    if (!(c[2] > 1e-3)) return; // a corner beside or behind the camera: no rectangle bounds the projection
    const double u = cam.fx * c[0] / c[2] + cam.cx, v = cam.fy * c[1] / c[2] + cam.cy;
    u0 = std::min(u0, u); u1 = std::max(u1, u);
    v0 = std::min(v0, v); v1 = std::max(v1, v);
  }
  if (!(u1 - u0 < 1e7) || !(v1 - v0 < 1e7)) return;
  // pixel (px, py) casts its rays through [px, px + 1) x [py, py + 1); the rotation's columns are unit to ~1e-7 and the
  // projection is evaluated in double: two pixels of margin cover the float ray set-up many times over
  const long x0 = (long)std::floor(u0) - 2, x1 = (long)std::ceil(u1) + 3, y0 = (long)std::floor(v0) - 2, y1 = (long)std::ceil(v1) + 3;
  cam.cull[0] = (int)std::max<long>(0, std::min<long>(W, x0));
  cam.cull[1] = (int)std::max<long>(0, std::min<long>(H, y0));
  cam.cull[2] = (int)std::max<long>(1, std::min<long>(W, x1)); // > 0: set
  cam.cull[3] = (int)std::max<long>(0, std::min<long>(H, y1));
}

void set_cull_rect(CamDev& cam, const Model& m, int W, int H) {
  OccBox b;
  for (int a = 0; a < 3; a++) {
    b.occ_lo[a] = m.occ_lo[a];
    b.occ_hi[a] = m.occ_hi[a];
  }
  set_cull_rect(cam, b, W, H);
}

constexpr int kSubRegions = 8, kMaxSegments = 8 * kSubRegions; // queue regions (<= 8: one per XCD) x their sub-regions
constexpr size_t kStatOffset = (size_t)kMaxSegments * 128;  // counters buffer: heads (64 B apart), then counts (64 B apart), then the statistics
constexpr size_t kCountersBytes = kStatOffset + 72 * 8; // {evaluated, wave rounds, clock sums and stamps}, then 8 live-sample shards a cache line apart


// which render_queue64 instance and relocation policy a launch gets (results are identical either way)
void render_policy(prv_ctx* c, const Model& m, size_t npix, bool ngp, RenderParams& rp) {
  // Tail merge + the block's tail pool raise slot utilisation (0.77 -> 0.93) at the price of incoherent gathers from the
  // relocated rays.  That pays where the gathers of a fresh cohort are coherent to begin with and the table is cache
  // resident -- large images of the 256^3 field: launch -8 % -- and costs elsewhere: the 512^3 field is bound by random
  // HBM requests (+2...5 %), and at the reference's 80x45 candidates neighbouring rays are five finest cells apart
  // (+3...7 %).  Results are identical either way; the default follows table and image size (PRV_MERGE_MAX / PRV_POOL
  // override).  Measured: profiles/archive/r02_k_tail_merge.txt, profiles/archive/r02_r_tail_pool.txt
  const bool coherent = m.table_halfs * 2 <= ((size_t)32 << 20) && npix >= ((size_t)1 << 17);
  // The per-lane corner cache: under the engine's stepping rule a ray's consecutive samples share their cell on the hashed
  // levels about half of the time, and where the cohort's gathers are incoherent (small images: every corner of every lane
  // its own cache line) the launch is bound by the L2's request rate -- 9.5 L2 requests per sample, 135 G/s, the ceiling
  // scripts/gather_calib.hip finds for this footprint (profiles/r04_reference_round_*) -- so a lane whose cell did not
  // change skips its eight loads of that level.  Costs ~50 VGPRs (two waves per SIMD, which those images run with
  // anyway) and a few VALU per level; off for large images, whose launch is issue-bound, and for the F = 2 fields (six
  // hashed levels do not fit the registers).  Cell keys hold 10 bits per axis.  PRV_CELL_CACHE overrides.
  const bool cached = (c->cell_cache >= 0 ? c->cell_cache != 0 : (ngp && !coherent)) && m.desc.finest_res <= 1023 &&
                      m.desc.n_features == 4 && m.dev.hash_shared && m.dev.n_dense_levels == 5;
  rp.cell_cache = cached ? 1 : 0;
  // With the cache in place the small-image launch is no longer request-bound and fuller slots pay again: relocation on
  // (merge threshold 24, pool), 0.68 -> 0.95 slot utilisation, -13 % launch time (profiles/r04_cell_cache_ab.txt)
  rp.merge_max = c->merge_max >= 0 ? c->merge_max : (coherent ? 16 : cached ? 24 : 0);
  rp.pool_on = (c->pool_on >= 0 ? c->pool_on != 0 : (coherent || cached)) && rp.merge_max > 0;
}

// The render of one batch of views into out_f32 (+ optional out_u8).  Views are dealt to
// the queue in batches so the queue stays within queue_budget bytes.
int render_views(prv_ctx* c, int slot, const prv_camset* cs, const int* view_ids, int n_views,
                 const prv_render_opts* o, float* out_f32, uint8_t* out_u8, bool zero_stats, bool private_output = false) {
  if (o->spp != 1 || out_u8) private_output = false; // sub-sample staging and byte images are written in full
  const Model& m = c->models[slot];
  const int W = o->width, H = o->height;
  const size_t npix = (size_t)W * H;
  int rc;
  // counters: 8 region heads (one 64-byte line each) | 8 region counts (same) | stats {evaluated, wave rounds} | dev histogram
  if ((rc = ensure(c, c->counters, kCountersBytes)) != PRV_OK) return rc;
  uint32_t* q_head = (uint32_t*)c->counters.p;
  uint32_t* q_count = (uint32_t*)c->counters.p + kMaxSegments * 16;
  unsigned long long* stat = (unsigned long long*)((char*)c->counters.p + kStatOffset);
  if (n_views == 0) {
    if (zero_stats) HIPCHK(c, hipMemsetAsync(stat, 0, kCountersBytes - kStatOffset, c->stream));
    return PRV_OK;
  }

  // cameras at this resolution, uploaded per call (tiny): built in pinned memory and sent with one asynchronous copy
  const size_t up_bytes = (size_t)n_views * (sizeof(CamDev) + sizeof(int));
  if (c->pin_cap < up_bytes) {
    if (c->pin_ev) HIPCHK(c, hipEventSynchronize(c->pin_ev));
    if (c->pin) (void)hipHostFree(c->pin);
    c->pin = nullptr;
    c->pin_cap = 0;
    HIPCHK(c, hipHostMalloc(&c->pin, std::max<size_t>(up_bytes, 8192), hipHostMallocDefault));
    c->pin_cap = std::max<size_t>(up_bytes, 8192);
  }
  if (!c->pin_ev) HIPCHK(c, hipEventCreateWithFlags(&c->pin_ev, hipEventDisableTiming));
  else HIPCHK(c, hipEventSynchronize(c->pin_ev));
  CamDev* cams = (CamDev*)c->pin;
  int* ids = (int*)((char*)c->pin + (size_t)n_views * sizeof(CamDev));
  for (int i = 0; i < n_views; i++) {
    const int v = view_ids ? view_ids[i] : i;
    if (v < 0 || v >= (int)cs->cams.size()) return fail(c, PRV_E_INVALID, "view id %d out of range", v);
    cams[i] = cam_at(cs, v, W, H);
    set_cull_rect(cams[i], m, W, H);
    ids[i] = i;
  }
  if ((rc = ensure(c, c->view_ids, up_bytes)) != PRV_OK) return rc;
  CamDev* cams_dev = (CamDev*)c->view_ids.p;
  int* ids_dev = (int*)((char*)c->view_ids.p + (size_t)n_views * sizeof(CamDev));
  HIPCHK(c, hipMemcpyAsync(c->view_ids.p, c->pin, up_bytes, hipMemcpyHostToDevice, c->stream));
  HIPCHK(c, hipEventRecord(c->pin_ev, c->stream));

  // all spp sub-samples of a batch of views go through ONE march + ONE render launch: sub-sample k of
  // view v is image (k*nb + v) of a staging buffer, reduced over k in order afterwards (spp = 1 renders
  // straight into the output).  Batches keep queue and staging within their budgets.
  const int spp = o->spp;
  const bool ngp = o->step_mode == PRV_STEP_NGP;
  const size_t slot_bytes = kRecordBytes + (ngp ? kExtBytes : 0); // NGP: every queue slot has its mask-extension slot
  size_t batch = std::max<size_t>(1, c->queue_budget / (npix * slot_bytes * (size_t)spp));
  batch = std::min<size_t>(batch, (size_t)n_views);
  if (spp > 1) batch = std::min<size_t>(batch, std::max<size_t>(1, c->stage_budget / (npix * 16 * (size_t)spp)));
  if (batch * npix * (size_t)spp >= (1ull << 32)) batch = ((1ull << 32) - 1) / (npix * (size_t)spp); // 32-bit pixel ids
  if (batch == 0) return fail(c, PRV_E_INVALID, "image x spp too large");
  // the queue is n_seg regions: a march block appends to region (linear block id % n_seg), so a region holds at most
  // ceil(blocks / n_seg) * 256 records -- the same total as one flat queue plus less than one block per region
  // spatial regions: every octant's region is cut into kSubRegions sub-regions with a counter each (a march block appends to
  // sub-region `linear block id % kSubRegions` of its octant's region): the waves running at any one time are neighbours in
  // the image, i.e. in ONE octant, and one returning-atomic word serves ~90 of them per microsecond
  const int n_sub = c->spatial_regions ? kSubRegions : 1;
  const int n_seg = c->queue_segments * n_sub;
  auto march_blocks = [&](int nb) {
    const int inner = (spp > 1 && spp <= 64 && (spp & (spp - 1)) == 0) ? spp : 1;
    int pl = 8;
    for (int v = inner; v > 1; v >>= 1) pl--;
    const size_t tw = (size_t)1 << ((pl + 1) / 2), th = (size_t)1 << (pl / 2);
    return ((W + tw - 1) / tw) * ((H + th - 1) / th) * (size_t)nb * (size_t)(inner > 1 ? 1 : spp);
  };
  const size_t seg_cap_max = ((march_blocks((int)batch) + n_seg - 1) / n_seg) * 256 + 64; // (+ 64: a wave moves to the next region when its own is full, region_reserve)
  if (seg_cap_max * (size_t)n_seg >= (1ull << 32)) return fail(c, PRV_E_INVALID, "image x spp too large");
  if ((rc = ensure(c, c->queue, seg_cap_max * (size_t)n_seg * kRecordBytes)) != PRV_OK) return rc;
  if (ngp && (rc = ensure(c, c->queue_ext, seg_cap_max * (size_t)n_seg * kExtBytes)) != PRV_OK) return rc;
  if (spp > 1 && (rc = ensure(c, c->stage, batch * npix * (size_t)spp * 16)) != PRV_OK) return rc;

  // Persistent render blocks per CU.  More resident waves hide more gather latency but also put more random requests in
  // flight and draw more power: measured per workload (64-slot kernel, profiles/archive/r02_bm_blocks_per_cu.txt) -- the
  // cache-resident table at large images is flat from 3 up (3 waves per SIMD is what its registers allow), small images
  // (incoherent gathers: the reference's 80x45 candidates, 320x320) are 6 % faster with 2, and the HBM-bound 512^3 table
  // is 10 % faster with ONE wave per SIMD: fewer requests in flight, a better L2 hit rate.
  int bpc = c->blocks_per_cu;
  if (bpc <= 0) bpc = m.table_halfs * 2 > ((size_t)32 << 20) ? 1 : npix < ((size_t)1 << 17) ? 2 : 4;
  const int n_blocks = c->n_cu * bpc;
  for (size_t b0 = 0; b0 < (size_t)n_views; b0 += batch) {
    const int nb = (int)std::min(batch, (size_t)n_views - b0);
    float* dst_f32 = out_f32 + b0 * npix * 4;
    uint32_t* dst_u8 = out_u8 ? (uint32_t*)out_u8 + b0 * npix : nullptr;
    // one fill per batch: heads and counts, and with them the statistics when this call starts a new window
    HIPCHK(c, hipMemsetAsync(c->counters.p, 0, b0 == 0 && zero_stats ? kCountersBytes : kStatOffset, c->stream));
    MarchParams mp;
    memset(&mp, 0, sizeof(mp));
    mp.field = m.dev;
    mp.cams = cams_dev;
    mp.view_ids = ids_dev + b0;
    mp.W = W;
    mp.H = H;
    mp.S = o->samples_per_ray;
    mp.spp_k = 0;
    // 256 threads = (256 / spp) pixels x spp sub-samples when spp is a power of two <= 64
    int inner = 0;
    if (spp > 1 && spp <= 64 && (spp & (spp - 1)) == 0)
      while ((1 << inner) < spp) inner++;
    const int pix_log2 = 8 - inner; // pixels per block, Morton-ordered: x gets the extra bit
    mp.spp_inner_log2 = inner;
    mp.tile_w_log2 = (pix_log2 + 1) / 2;
    mp.tile_h_log2 = pix_log2 / 2;
    mp.tiles_x = (uint32_t)((W + (1 << mp.tile_w_log2) - 1) >> mp.tile_w_log2);
    mp.tiles_y = (uint32_t)((H + (1 << mp.tile_h_log2) - 1) >> mp.tile_h_log2);
    mp.step_mode = o->step_mode;
    mp.queue = c->queue.p;
    mp.queue_ext = (uint4*)c->queue_ext.p;
    mp.stat = stat;
    mp.queue_count = q_count;
    mp.n_seg = n_seg;
    mp.n_sub = n_sub;
    mp.seg_cap = (uint32_t)(((march_blocks(nb) + n_seg - 1) / n_seg) * 256 + 64);
    mp.spatial_regions = c->spatial_regions;
    mp.out_f32 = spp > 1 ? (float*)c->stage.p : dst_f32;
    mp.out_u8 = spp > 1 ? nullptr : dst_u8;
    mp.inv_spp = 1.0f;
    mp.last_pass = spp == 1; // staged sub-samples are written raw; scaling / bytes happen in the reduce
    memcpy(mp.bg, o->background, sizeof(mp.bg));
    // A caller that consumes the image through the views' cull rectangles (prv_score_views, method 5: the image is a private
    // temporary of the round) gets only the tiles inside them launched and nothing written outside: most of an 800x800
    // view of the bench scene is dead, and 16 B per dead pixel were a 0.4 ms stream of zeros per 64-view step.
    if (private_output) {
      const uint64_t tw = 1ull << mp.tile_w_log2, th = 1ull << mp.tile_h_log2;
      uint64_t live_max = 0;
      for (int i = 0; i < nb; i++) {
        const CamDev& cv = cams[b0 + i];
        uint64_t w = mp.tiles_x, h = mp.tiles_y;
        if (cv.cull[2] > 0) {
          const uint64_t x0 = (uint64_t)cv.cull[0] / tw, y0 = (uint64_t)cv.cull[1] / th;
          const uint64_t x1 = std::min<uint64_t>(mp.tiles_x, ((uint64_t)cv.cull[2] + tw - 1) / tw);
          const uint64_t y1 = std::min<uint64_t>(mp.tiles_y, ((uint64_t)std::max(cv.cull[3], 0) + th - 1) / th);
          w = x1 > x0 ? x1 - x0 : 0;
          h = y1 > y0 ? y1 - y0 : 0;
        }
        live_max = std::max(live_max, w * h);
      }
      mp.live_grid = 1;
      mp.live_tiles_max = (uint32_t)live_max;
    }
    if (c->profiling) {
      hipEvent_t a, b;
      HIPCHK(c, take_event(c, &a));
      HIPCHK(c, take_event(c, &b));
      c->ev_march.push_back(a);
      c->ev_march.push_back(b);
      HIPCHK(c, hipEventRecord(a, c->stream));
    }
    HIPCHK(c, launch_march(mp, nb, spp, c->stream));
    if (c->profiling) HIPCHK(c, hipEventRecord(c->ev_march.back(), c->stream));
    RenderParams rp;
    memset(&rp, 0, sizeof(rp));
    rp.field = m.dev;
    rp.queue = c->queue.p;
    rp.queue_ext = (const uint4*)c->queue_ext.p;
    rp.step_mode = o->step_mode;
    rp.queue_count = q_count;
    rp.queue_head = q_head;
    rp.n_segments = n_seg;
    rp.n_sub = n_sub;
    rp.seg_cap = mp.seg_cap;
    rp.stat_evaluated = stat;
    rp.out_f32 = mp.out_f32;
    rp.out_u8 = mp.out_u8;
    rp.min_T = o->min_transmittance;
    rp.inv_spp = 1.0f;
    rp.spp_k = 0;
    rp.last_pass = mp.last_pass;
    render_policy(c, m, npix, ngp, rp);
    memcpy(rp.bg, o->background, sizeof(rp.bg));
    if (c->profiling) {
      hipEvent_t a, b;
      HIPCHK(c, take_event(c, &a));
      HIPCHK(c, take_event(c, &b));
      c->ev_render.push_back(a);
      c->ev_render.push_back(b);
      HIPCHK(c, hipEventRecord(a, c->stream));
    }
    HIPCHK(c, launch_render(rp, n_blocks, c->stream));
    if (c->profiling) HIPCHK(c, hipEventRecord(c->ev_render.back(), c->stream));
    if (spp > 1) HIPCHK(c, launch_spp_reduce((const float*)c->stage.p, (size_t)nb * npix, spp, o->background, dst_f32, dst_u8, c->stream));
  }
  return PRV_OK;
}

// The ensemble's candidates under the engine's stepping rule: ONE march launch for all members (march_multi_kernel: one
// walk per ray answers every member's occupancy), then one render launch + sub-sample reduce per member from that member's
// own queue.  Every member's masks, records and images are what render_views gives it on its own; only the order of the
// records in the queues differs (as between any two launches).  Returns PRV_OK with *done = false when this path does not
// apply (the caller then renders member by member).
int render_ensemble_ngp(prv_ctx* c, const int* slots, int E, const prv_camset* cs, const int* view_ids, int n_views,
                        const prv_render_opts* o, float* scratch_f32, uint8_t* const* out_u8, bool* done) {
  *done = false;
  if (o->step_mode != PRV_STEP_NGP || !march_multi_supported(E) || n_views == 0) return PRV_OK;
  if (c->march_multi == 0) return PRV_OK;
  // measured (profiles/r05_march_multi.txt): five members 15.6 -> 7.5 ms per round of the reference's size; two members 6.2 ->
  // 7.0 (the launch walks twice, which two members do not win back): on by default from three members up
  if (c->march_multi < 0 && E < 3) return PRV_OK;
  const Model& m0 = c->models[slots[0]];
  for (int e = 1; e < E; e++) {
    const Model& m = c->models[slots[e]];
    if (m.desc.occ_res != m0.desc.occ_res || (m.dev.occ_coarse == nullptr) != (m0.dev.occ_coarse == nullptr)) return PRV_OK;
    for (int f = 0; f < e; f++)
      if (slots[f] == slots[e]) return PRV_OK; // (a slot listed twice is legal for the member-by-member path)
  }
  const int W = o->width, H = o->height, spp = o->spp;
  const size_t npix = (size_t)W * H;
  const size_t slot_bytes = kRecordBytes + kExtBytes;
  // all views in ONE batch, every member with a queue and a staging image of its own: refuse (-> member by member, batched)
  // when that does not fit the budgets
  if ((size_t)n_views * npix * (size_t)spp >= (1ull << 32)) return PRV_OK;
  if ((size_t)n_views * npix * (size_t)spp * slot_bytes > c->queue_budget) return PRV_OK; // per member, as render_views batches
  if (spp > 1 && (size_t)n_views * npix * (size_t)spp * 16 > c->stage_budget) return PRV_OK;
  int rc;
  if ((rc = ensure(c, c->counters, kCountersBytes)) != PRV_OK) return rc;
  if ((rc = ensure(c, c->counters_multi, (size_t)E * kStatOffset)) != PRV_OK) return rc;
  unsigned long long* stat = (unsigned long long*)((char*)c->counters.p + kStatOffset);

  // the members' union box: clip range and cull rectangles (conservative for every member)
  OccBox box;
  for (int a = 0; a < 3; a++) {
    box.occ_lo[a] = 1e30f;
    box.occ_hi[a] = -1e30f;
  }
  bool any = false;
  for (int e = 0; e < E; e++) {
    const Model& m = c->models[slots[e]];
    if (!(m.occ_hi[0] > m.occ_lo[0])) continue; // an empty grid adds nothing
    any = true;
    for (int a = 0; a < 3; a++) {
      box.occ_lo[a] = std::min(box.occ_lo[a], m.occ_lo[a]);
      box.occ_hi[a] = std::max(box.occ_hi[a], m.occ_hi[a]);
    }
  }
  if (!any)
    for (int a = 0; a < 3; a++) box.occ_lo[a] = box.occ_hi[a] = 0.f;

  const size_t up_bytes = (size_t)n_views * (sizeof(CamDev) + sizeof(int));
  if (c->pin_cap < up_bytes) {
    if (c->pin_ev) HIPCHK(c, hipEventSynchronize(c->pin_ev));
    if (c->pin) (void)hipHostFree(c->pin);
    c->pin = nullptr;
    c->pin_cap = 0;
    HIPCHK(c, hipHostMalloc(&c->pin, std::max<size_t>(up_bytes, 8192), hipHostMallocDefault));
    c->pin_cap = std::max<size_t>(up_bytes, 8192);
  }
  if (!c->pin_ev) HIPCHK(c, hipEventCreateWithFlags(&c->pin_ev, hipEventDisableTiming));
  else HIPCHK(c, hipEventSynchronize(c->pin_ev));
  CamDev* cams = (CamDev*)c->pin;
  int* ids = (int*)((char*)c->pin + (size_t)n_views * sizeof(CamDev));
  for (int i = 0; i < n_views; i++) {
    const int v = view_ids ? view_ids[i] : i;
    if (v < 0 || v >= (int)cs->cams.size()) return fail(c, PRV_E_INVALID, "view id %d out of range", v);
    cams[i] = cam_at(cs, v, W, H);
    set_cull_rect(cams[i], box, W, H);
    ids[i] = i;
  }
  if ((rc = ensure(c, c->view_ids, up_bytes)) != PRV_OK) return rc;
  CamDev* cams_dev = (CamDev*)c->view_ids.p;
  int* ids_dev = (int*)((char*)c->view_ids.p + (size_t)n_views * sizeof(CamDev));
  HIPCHK(c, hipMemcpyAsync(c->view_ids.p, c->pin, up_bytes, hipMemcpyHostToDevice, c->stream));
  HIPCHK(c, hipEventRecord(c->pin_ev, c->stream));

  // the members' occupancy, interleaved: byte c = bit c of every member's bitfield (fine grid, then the dilated coarse one)
  const size_t R = (size_t)m0.desc.occ_res, n_fine = R * R * R, Rc = R / 4, n_coarse = Rc * Rc * Rc;
  const bool have_coarse = m0.dev.occ_coarse != nullptr;
  if ((rc = ensure(c, c->occ_multi, n_fine + n_coarse + 64)) != PRV_OK) return rc;
  {
    OccInterleaveParams ip{};
    ip.n_members = E;
    for (int e = 0; e < E; e++) ip.bits[e] = c->models[slots[e]].dev.occ;
    ip.n_cells = (uint32_t)n_fine;
    ip.out = (uint8_t*)c->occ_multi.p;
    HIPCHK(c, launch_occ_interleave(ip, c->stream));
    if (have_coarse) {
      for (int e = 0; e < E; e++) ip.bits[e] = c->models[slots[e]].dev.occ_coarse;
      ip.n_cells = (uint32_t)n_coarse;
      ip.out = (uint8_t*)c->occ_multi.p + n_fine;
      HIPCHK(c, launch_occ_interleave(ip, c->stream));
    }
  }

  // spatial regions: every octant's region is cut into kSubRegions sub-regions with a counter each (a march block appends to
  // sub-region `linear block id % kSubRegions` of its octant's region): the waves running at any one time are neighbours in
  // the image, i.e. in ONE octant, and one returning-atomic word serves ~90 of them per microsecond
  const int n_sub = c->spatial_regions ? kSubRegions : 1;
  const int n_seg = c->queue_segments * n_sub;
  int inner = 0;
  if (spp > 1 && spp <= 64 && (spp & (spp - 1)) == 0)
    while ((1 << inner) < spp) inner++;
  const int pix_log2 = 8 - inner;
  const int tile_w_log2 = (pix_log2 + 1) / 2, tile_h_log2 = pix_log2 / 2;
  const uint32_t tiles_x = (uint32_t)((W + (1 << tile_w_log2) - 1) >> tile_w_log2), tiles_y = (uint32_t)((H + (1 << tile_h_log2) - 1) >> tile_h_log2);
  const size_t blocks = (size_t)tiles_x * tiles_y * (size_t)n_views * (size_t)(inner > 0 ? 1 : spp);
  const size_t seg_cap = ((blocks + n_seg - 1) / n_seg) * 256 + 64;
  if (seg_cap * (size_t)n_seg >= (1ull << 32)) return PRV_OK;
  const size_t q_stride = seg_cap * (size_t)n_seg * kRecordBytes, x_stride = seg_cap * (size_t)n_seg * kExtBytes,
               s_stride = spp > 1 ? (size_t)n_views * npix * (size_t)spp * 16 : 0;
  { // E queues, E extension buffers, E staging images at once: only where the device has the room (else member by member,
    // whose buffers are a fifth of these)
    const size_t grow = (q_stride * (size_t)E > c->queue.bytes ? q_stride * (size_t)E - c->queue.bytes : 0) +
                        (x_stride * (size_t)E > c->queue_ext.bytes ? x_stride * (size_t)E - c->queue_ext.bytes : 0) +
                        (s_stride * (size_t)E > c->stage.bytes ? s_stride * (size_t)E - c->stage.bytes : 0);
    size_t free_b = 0, total_b = 0;
    // (the context's parked trainer buffers count as free: ensure() releases them before an allocation fails)
    if (grow && (hipMemGetInfo(&free_b, &total_b) != hipSuccess || grow + ((size_t)2 << 30) > free_b + c->idle_bytes)) return PRV_OK;
  }
  if ((rc = ensure(c, c->queue, q_stride * (size_t)E)) != PRV_OK) return rc;
  if ((rc = ensure(c, c->queue_ext, x_stride * (size_t)E)) != PRV_OK) return rc;
  if (spp > 1 && (rc = ensure(c, c->stage, s_stride * (size_t)E)) != PRV_OK) return rc;
  HIPCHK(c, hipMemsetAsync(c->counters.p, 0, kCountersBytes, c->stream)); // a new statistics window (this call renders every member)
  HIPCHK(c, hipMemsetAsync(c->counters_multi.p, 0, (size_t)E * kStatOffset, c->stream));

  MarchMultiParams mp;
  memset(&mp, 0, sizeof(mp));
  mp.occ_bytes = (const uint8_t*)c->occ_multi.p;
  mp.occ_coarse_bytes = have_coarse ? (const uint8_t*)c->occ_multi.p + n_fine : nullptr;
  mp.occ_res = (int)R;
  for (int a = 0; a < 3; a++) {
    mp.occ_lo[a] = box.occ_lo[a];
    mp.occ_hi[a] = box.occ_hi[a];
  }
  mp.cams = cams_dev;
  mp.view_ids = ids_dev;
  mp.W = W;
  mp.H = H;
  mp.spp_k = 0;
  mp.tiles_x = tiles_x;
  mp.tiles_y = tiles_y;
  mp.tile_w_log2 = tile_w_log2;
  mp.tile_h_log2 = tile_h_log2;
  mp.spp_inner_log2 = inner;
  mp.stat = stat;
  mp.n_seg = n_seg;
  mp.n_sub = n_sub;
  mp.seg_cap = (uint32_t)seg_cap;
  mp.spatial_regions = c->spatial_regions;
  mp.inv_spp = 1.0f;
  mp.last_pass = spp == 1;
  memcpy(mp.bg, o->background, sizeof(mp.bg));
  mp.n_members = E;
  for (int e = 0; e < E; e++) {
    MarchMember& mm = mp.mem[e];
    mm.queue = (char*)c->queue.p + q_stride * (size_t)e;
    mm.queue_ext = (uint4*)((char*)c->queue_ext.p + x_stride * (size_t)e);
    mm.queue_count = (uint32_t*)((char*)c->counters_multi.p + (size_t)e * kStatOffset) + kMaxSegments * 16;
    // spp > 1: the member's own staging image; spp == 1: the caller's scratch image is shared (nobody reads it), the bytes are the member's
    mm.out_f32 = spp > 1 ? (float*)((char*)c->stage.p + s_stride * (size_t)e) : scratch_f32;
    mm.out_u8 = spp > 1 ? nullptr : (uint32_t*)out_u8[e];
  }
  if (c->profiling) {
    hipEvent_t a, b;
    HIPCHK(c, take_event(c, &a));
    HIPCHK(c, take_event(c, &b));
    c->ev_march.push_back(a);
    c->ev_march.push_back(b);
    HIPCHK(c, hipEventRecord(a, c->stream));
  }
  HIPCHK(c, launch_march_multi(mp, n_views, spp, c->stream));
  if (c->profiling) HIPCHK(c, hipEventRecord(c->ev_march.back(), c->stream));

  for (int e = 0; e < E; e++) {
    const Model& m = c->models[slots[e]];
    int bpc = c->blocks_per_cu;
    if (bpc <= 0) bpc = m.table_halfs * 2 > ((size_t)32 << 20) ? 1 : npix < ((size_t)1 << 17) ? 2 : 4;
    RenderParams rp;
    memset(&rp, 0, sizeof(rp));
    rp.field = m.dev;
    rp.queue = mp.mem[e].queue;
    rp.queue_ext = mp.mem[e].queue_ext;
    rp.step_mode = o->step_mode;
    rp.queue_count = mp.mem[e].queue_count;
    rp.queue_head = (uint32_t*)((char*)c->counters_multi.p + (size_t)e * kStatOffset);
    rp.n_segments = n_seg;
    rp.n_sub = n_sub;
    rp.seg_cap = mp.seg_cap;
    rp.stat_evaluated = stat;
    rp.out_f32 = mp.mem[e].out_f32;
    rp.out_u8 = mp.mem[e].out_u8;
    rp.min_T = o->min_transmittance;
    rp.inv_spp = 1.0f;
    rp.spp_k = 0;
    rp.last_pass = mp.last_pass;
    render_policy(c, m, npix, true, rp);
    memcpy(rp.bg, o->background, sizeof(rp.bg));
    if (c->profiling) {
      hipEvent_t a, b;
      HIPCHK(c, take_event(c, &a));
      HIPCHK(c, take_event(c, &b));
      c->ev_render.push_back(a);
      c->ev_render.push_back(b);
      HIPCHK(c, hipEventRecord(a, c->stream));
    }
    HIPCHK(c, launch_render(rp, c->n_cu * bpc, c->stream));
    if (c->profiling) HIPCHK(c, hipEventRecord(c->ev_render.back(), c->stream));
    if (spp > 1)
      HIPCHK(c, launch_spp_reduce((const float*)mp.mem[e].out_f32, (size_t)n_views * npix, spp, o->background, scratch_f32, (uint32_t*)out_u8[e], c->stream));
  }
  *done = true;
  return PRV_OK;
}

int fetch_stats(prv_ctx* c, const prv_render_opts* o, int n_views, int n_models, prv_stats* st) {
  if (!st) return PRV_OK;
  unsigned long long ev[72] = {0};
  HIPCHK(c, hipMemcpyAsync(ev, (char*)c->counters.p + kStatOffset, sizeof(ev), hipMemcpyDeviceToHost, c->stream));
  HIPCHK(c, hipStreamSynchronize(c->stream));
  st->wave_rounds = ev[1];
  st->rays = (uint64_t)n_views * n_models * o->width * o->height * o->spp;
  st->samples_nominal = st->rays * (uint64_t)(o->step_mode == PRV_STEP_NGP ? kNgpMaxSteps : o->samples_per_ray);
  st->samples_evaluated = ev[0];
  st->samples_live = 0;
  for (int s = 0; s < 8; s++) st->samples_live += ev[8 * (1 + s)]; // the march pass's sharded counter
  return PRV_OK;
}

int score_blocks(size_t npix) { return (int)std::min<size_t>(64, std::max<size_t>(1, (npix + 4095) / 4096)); }

} // namespace

// ------------------------------------------------------------------ context

extern "C" {

int prv_abi_version(void) { return PRV_ABI_VERSION; }

int prv_device_count(void) {
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess) return 0;
  return n;
}

const char* prv_last_error(const prv_ctx* ctx) { return ctx ? ctx->err.c_str() : g_create_error.c_str(); }

int prv_create(prv_ctx** out, int device_id) try {
  if (!out) return fail(nullptr, PRV_E_INVALID, "out is NULL");
  *out = nullptr;
  int n = 0;
  hipError_t e = hipGetDeviceCount(&n);
  if (e != hipSuccess || n <= 0)
    return fail(nullptr, PRV_E_NODEVICE, "no HIP device visible (%s): this library has no CPU path",
                e == hipSuccess ? "count = 0" : hipGetErrorString(e));
  if (device_id < 0 || device_id >= n) return fail(nullptr, PRV_E_INVALID, "device %d out of range [0,%d)", device_id, n);
  hipDeviceProp_t prop;
  if ((e = hipSetDevice(device_id)) != hipSuccess || (e = hipGetDeviceProperties(&prop, device_id)) != hipSuccess)
    return fail(nullptr, PRV_E_HIP, "device %d: %s", device_id, hipGetErrorString(e));
  if (!strstr(prop.gcnArchName, "gfx950"))
    return fail(nullptr, PRV_E_NODEVICE, "device %d is %s; this library is built for gfx950 only", device_id, prop.gcnArchName);
  prv_ctx* c = new prv_ctx();
  c->device = device_id;
  c->n_cu = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
  if ((e = hipStreamCreateWithFlags(&c->own_stream, hipStreamNonBlocking)) != hipSuccess) {
    delete c;
    return fail(nullptr, PRV_E_HIP, "stream create: %s", hipGetErrorString(e));
  }
  c->stream = c->own_stream;
  if (const char* s = getenv("PRV_BLOCKS_PER_CU")) c->blocks_per_cu = std::max(1, atoi(s));
  if (const char* s = getenv("PRV_QUEUE_SEGMENTS")) c->queue_segments = std::min(8, std::max(1, atoi(s)));
  if (const char* s = getenv("PRV_SPATIAL_REGIONS")) c->spatial_regions = std::max(0, std::min(2, atoi(s)));
  if (const char* s = getenv("PRV_MERGE_MAX")) c->merge_max = std::min(31, std::max(0, atoi(s)));
  if (const char* s = getenv("PRV_POOL")) c->pool_on = atoi(s) != 0 ? 1 : 0;
  if (const char* s = getenv("PRV_CELL_CACHE")) c->cell_cache = atoi(s) != 0 ? 1 : 0;
  if (const char* s = getenv("PRV_MARCH_MULTI")) c->march_multi = atoi(s) != 0 ? 1 : 0;
  if (const char* s = getenv("PRV_QUEUE_MB")) c->queue_budget = (size_t)std::max(1, atoi(s)) << 20;
  g_live_contexts.fetch_add(1);
  if (device_id < 64) g_devices_used.fetch_or(1ull << device_id);
  *out = c;
  return PRV_OK;
} catch (...) { return caught(nullptr); }

void prv_destroy(prv_ctx* c) {
  if (!c) return;
  (void)hipSetDevice(c->device);
  (void)hipStreamSynchronize(c->stream);
  train_detach_all(c); // trainers outliving their context become inert handles
  comm_detach_all(c);  // ... and so do communicators
  for (hipEvent_t e : c->ev_render) (void)hipEventDestroy(e);
  for (hipEvent_t e : c->ev_march) (void)hipEventDestroy(e);
  for (hipEvent_t e : c->ev_free) (void)hipEventDestroy(e);
  for (auto& m : c->models) {
    release(m.table);
    release(m.phys);
    release(m.occ_coarse);
    release(m.occ);
    release(m.frags);
    release(m.frags64);
    release(m.mlp);
  }
  release(c->queue);
  release(c->queue_ext);
  release(c->stage);
  release(c->counters);
  if (c->pin) (void)hipHostFree(c->pin);
  if (c->pin_ev) (void)hipEventDestroy(c->pin_ev);
  release(c->view_ids);
  release(c->img_f32);
  release(c->partial);
  release(c->records);
  for (auto& b : c->dbg) release(b);
  for (auto& b : c->img_u8) release(b);
  if (c->own_stream) (void)hipStreamDestroy(c->own_stream);
  delete c;
  g_live_contexts.fetch_sub(1);
}

int prv_runtime_shutdown(void) {
  if (g_live_contexts.load() != 0) {
    g_create_error = "prv_runtime_shutdown: " + std::to_string(g_live_contexts.load()) + " context(s) are still alive";
    return PRV_E_STATE;
  }
  const unsigned long long used = g_devices_used.exchange(0ull);
  for (int d = 0; d < 64; d++) {
    if (!((used >> d) & 1ull)) continue;
    if (hipSetDevice(d) != hipSuccess) continue;
    (void)hipDeviceSynchronize();
    // librccl (when a communicator loaded it) stays loaded and keeps device state of its own that its static destructors
    // release after main: the device is then only synchronised, not reset under it
    if (!g_rccl_loaded.load()) (void)hipDeviceReset();
  }
  (void)hipGetLastError();
  return PRV_OK;
}

int prv_set_stream(prv_ctx* c, void* s) try {
  if (!c) return PRV_E_INVALID;
  HIPCHK(c, hipSetDevice(c->device)); // the calling thread may be a new one (its current device would be 0)
  HIPCHK(c, hipStreamSynchronize(c->stream));
  c->stream = (hipStream_t)s; // NULL is HIP's legacy default stream (what torch uses by default)
  return PRV_OK;
} catch (...) { return caught(c); }

int prv_set_coverage_weight(prv_ctx* c, double weight) try {
  if (!c) return PRV_E_INVALID;
  if (!std::isfinite(weight)) return fail(c, PRV_E_INVALID, "coverage weight must be finite");
  c->coverage_weight = weight;
  return PRV_OK;
} catch (...) { return caught(c); }

int prv_synchronize(prv_ctx* c) try {
  if (!c) return PRV_E_INVALID;
  HIPCHK(c, hipSetDevice(c->device)); // the calling thread may be a new one (its current device would be 0)
  HIPCHK(c, hipStreamSynchronize(c->stream));
  return PRV_OK;
} catch (...) { return caught(c); }

static int drain_events(prv_ctx* c, std::vector<hipEvent_t>& ev, double* ms, int* n, std::vector<float>* each = nullptr) {
  double tot = 0.0;
  if (each) each->clear();
  for (size_t i = 0; i + 1 < ev.size(); i += 2) {
    float t = 0.f;
    HIPCHK(c, hipEventElapsedTime(&t, ev[i], ev[i + 1]));
    tot += t;
    if (each) each->push_back(t);
  }
  if (ms) *ms = tot;
  if (n) *n = (int)(ev.size() / 2);
  for (hipEvent_t e : ev) c->ev_free.push_back(e);
  ev.clear();
  return PRV_OK;
}

int prv_profile_begin(prv_ctx* c) try {
  if (!c) return PRV_E_INVALID;
  HIPCHK(c, hipSetDevice(c->device)); // the calling thread may be a new one (its current device would be 0)
  HIPCHK(c, hipStreamSynchronize(c->stream));
  drain_events(c, c->ev_render, nullptr, nullptr);
  drain_events(c, c->ev_march, nullptr, nullptr);
  c->profiling = true;
  return PRV_OK;
} catch (...) { return caught(c); }

int prv_profile_end(prv_ctx* c, double* render_ms, int* render_n, double* march_ms, int* march_n) try {
  if (!c) return PRV_E_INVALID;
  HIPCHK(c, hipSetDevice(c->device)); // the calling thread may be a new one (its current device would be 0)
  c->profiling = false;
  HIPCHK(c, hipStreamSynchronize(c->stream));
  int rc = drain_events(c, c->ev_render, render_ms, render_n, &c->last_render_ms);
  if (rc != PRV_OK) return rc;
  return drain_events(c, c->ev_march, march_ms, march_n);
} catch (...) { return caught(c); }

/* the render launches of the window prv_profile_end closed last, one duration each, in launch order (a scoring round of an
 * E-member ensemble: E per round, member 0 first); returns how many there were (ms may be NULL or shorter) */
int prv_profile_render_launches(prv_ctx* c, float* ms, int cap) try {
  if (!c) return PRV_E_INVALID;
  for (int i = 0; ms && i < cap && i < (int)c->last_render_ms.size(); i++) ms[i] = c->last_render_ms[(size_t)i];
  return (int)c->last_render_ms.size();
} catch (...) { return caught(c); }

int prv_malloc(prv_ctx* c, void** p, size_t bytes) try {
  if (!c || !p) return PRV_E_INVALID;
  HIPCHK(c, hipSetDevice(c->device));
  HIPCHK(c, hipMalloc(p, bytes ? bytes : 1));
  return PRV_OK;
} catch (...) { return caught(c); }
int prv_free(prv_ctx* c, void* p) try {
  if (!c) return PRV_E_INVALID;
  HIPCHK(c, hipSetDevice(c->device));
  HIPCHK(c, hipStreamSynchronize(c->stream));
  HIPCHK(c, hipFree(p));
  return PRV_OK;
} catch (...) { return caught(c); }
int prv_memcpy_h2d(prv_ctx* c, void* d, const void* s, size_t bytes) try {
  if (!c) return PRV_E_INVALID;
  HIPCHK(c, hipSetDevice(c->device));
  HIPCHK(c, hipMemcpyAsync(d, s, bytes, hipMemcpyHostToDevice, c->stream));
  HIPCHK(c, hipStreamSynchronize(c->stream));
  return PRV_OK;
} catch (...) { return caught(c); }
int prv_memcpy_d2h(prv_ctx* c, void* d, const void* s, size_t bytes) try {
  if (!c) return PRV_E_INVALID;
  HIPCHK(c, hipSetDevice(c->device));
  HIPCHK(c, hipMemcpyAsync(d, s, bytes, hipMemcpyDeviceToHost, c->stream));
  HIPCHK(c, hipStreamSynchronize(c->stream));
  return PRV_OK;
} catch (...) { return caught(c); }

// ------------------------------------------------------------------ model

int prv_model_sizes(const prv_field_desc* d, uint64_t* table_halfs, uint64_t* mlp_halfs, uint64_t* occ_words) try {
  if (!d) return PRV_E_INVALID;
  HostLevel lv[kMaxLevels];
  uint64_t total = 0;
  if (compute_levels(*d, lv, &total) != 0) return PRV_E_INVALID;
  if (table_halfs) *table_halfs = total * (uint64_t)d->n_features;
  if (mlp_halfs) *mlp_halfs = PRV_MLP_HALFS;
  const uint64_t R = (uint64_t)d->occ_res;
  if (occ_words) *occ_words = (R * R * R + 31) / 32;
  return PRV_OK;
} catch (...) { return caught(nullptr); }

int prv_model_load(prv_ctx* c, int slot, const prv_field_desc* d, const uint16_t* table, const uint16_t* mlp,
                   const uint32_t* occ) try {
  if (!c) return PRV_E_INVALID;
  if (slot < 0 || slot >= PRV_MAX_SLOTS) return fail(c, PRV_E_INVALID, "model slot %d out of range", slot);
  if (!d || !table || !mlp || !occ) return fail(c, PRV_E_INVALID, "NULL argument");
  HIPCHK(c, hipSetDevice(c->device));
  c->models[slot].generation++; // a live trainer of this slot holds stale masters from now on (train_check)
  return install_model(c, slot, *d, mlp, occ, table);
} catch (...) { return caught(c); }

static int model_synthetic(prv_ctx* c, int slot, const prv_field_desc* d, uint64_t seed, bool all_occupied);
int prv_model_synthetic(prv_ctx* c, int slot, const prv_field_desc* d, uint64_t seed) try {
  return model_synthetic(c, slot, d, seed, false);
} catch (...) { return caught(c); }
int prv_model_fresh(prv_ctx* c, int slot, const prv_field_desc* d, uint64_t seed) try {
  return model_synthetic(c, slot, d, seed, true);
} catch (...) { return caught(c); }
static int model_synthetic(prv_ctx* c, int slot, const prv_field_desc* d, uint64_t seed, bool all_occupied) {
  if (!c) return PRV_E_INVALID;
  if (slot < 0 || slot >= PRV_MAX_SLOTS) return fail(c, PRV_E_INVALID, "model slot %d out of range", slot);
  if (!d) return fail(c, PRV_E_INVALID, "NULL descriptor");
  HIPCHK(c, hipSetDevice(c->device));
  HostLevel lv[kMaxLevels];
  uint64_t total = 0;
  if (compute_levels(*d, lv, &total) != 0) return fail(c, PRV_E_INVALID, "invalid field descriptor");
  std::vector<uint16_t> mlp(PRV_MLP_HALFS);
  size_t o = 0;
  for (int l = 0; l < 5; l++) {
    const float amp = sqrtf(6.0f / (float)(kIn[l] + kOut[l]));
    const size_t cnt = (size_t)kIn[l] * kOut[l];
    for (size_t i = 0; i < cnt; i++) mlp[o + i] = f2h(rng_sym(seed, (uint64_t)(l + 1), i, amp));
    o += cnt;
  }
  const int R = d->occ_res;
  std::vector<uint32_t> occ(((size_t)R * R * R + 31) / 32, 0u);
  const float invR = 1.0f / (float)R;
  if (all_occupied) { // a fresh field (the planner loop makes five per round): every cell, without the walk over 2 M of them
    const size_t n_cells = (size_t)R * R * R;
    std::fill(occ.begin(), occ.end(), 0xffffffffu);
    if (n_cells & 31) occ.back() = (1u << (n_cells & 31)) - 1u;
  }
  for (int z = 0; z < (all_occupied ? 0 : R); z++)
    for (int y = 0; y < R; y++)
      for (int x = 0; x < R; x++) {
        const float cx = ((float)x + 0.5f) * invR, cy = ((float)y + 0.5f) * invR, cz = ((float)z + 0.5f) * invR;
        bool in = false;
        for (int b = 0; b < 4 && !in; b++) {
          const float dx = cx - kSynthSpheres[b][0], dy = cy - kSynthSpheres[b][1], dz = cz - kSynthSpheres[b][2];
          const float d2 = fmaf(dx, dx, fmaf(dy, dy, dz * dz));
          in = d2 <= kSynthSpheres[b][3] * kSynthSpheres[b][3];
        }
        if (in) {
          const size_t bit = (size_t)x + (size_t)R * ((size_t)y + (size_t)R * (size_t)z);
          occ[bit >> 5] |= 1u << (bit & 31);
        }
      }
  Model& m = c->models[slot];
  int rc = ensure(c, m.table, total * (uint64_t)d->n_features * 2);
  if (rc != PRV_OK) return rc;
  HIPCHK(c, launch_synth_table((uint16_t*)m.table.p, total * (uint64_t)d->n_features, seed, d->table_amp, c->stream));
  m.generation++;
  return install_model(c, slot, *d, mlp.data(), occ.data(), nullptr);
}

int prv_model_export(prv_ctx* c, int slot, uint16_t* table, uint16_t* mlp, uint32_t* occ) try {
  if (!c) return PRV_E_INVALID;
  HIPCHK(c, hipSetDevice(c->device)); // the calling thread may be a new one (its current device would be 0)
  int rc = check_model(c, slot);
  if (rc != PRV_OK) return rc;
  const Model& m = c->models[slot];
  if (table) HIPCHK(c, hipMemcpyAsync(table, m.table.p, m.table_halfs * 2, hipMemcpyDeviceToHost, c->stream));
  if (mlp) HIPCHK(c, hipMemcpyAsync(mlp, m.mlp.p, PRV_MLP_HALFS * 2, hipMemcpyDeviceToHost, c->stream));
  if (occ) HIPCHK(c, hipMemcpyAsync(occ, m.occ.p, m.occ_words * 4, hipMemcpyDeviceToHost, c->stream));
  HIPCHK(c, hipStreamSynchronize(c->stream));
  return PRV_OK;
} catch (...) { return caught(c); }

// the PRVF container's own version (the prv_field_desc of ABI version 3 onwards); not tied to later ABI bumps
static constexpr uint32_t kModelFileVersion = 3u;

int prv_model_save_file(prv_ctx* c, int slot, const char* path) try {
  if (!c) return PRV_E_INVALID;
  int rc = check_model(c, slot);
  if (rc != PRV_OK) return rc;
  if (!path) return fail(c, PRV_E_INVALID, "path is NULL");
  const Model& m = c->models[slot];
  std::vector<uint16_t> table(m.table_halfs), mlp(PRV_MLP_HALFS);
  std::vector<uint32_t> occ(m.occ_words);
  if ((rc = prv_model_export(c, slot, table.data(), mlp.data(), occ.data())) != PRV_OK) return rc;
  FILE* f = fopen(path, "wb");
  if (!f) return fail(c, PRV_E_IO, "cannot write %s", path);
  const uint32_t head[2] = {0x46565250u /* "PRVF" */, kModelFileVersion};
  bool ok = fwrite(head, sizeof(head), 1, f) == 1 && fwrite(&m.desc, sizeof(m.desc), 1, f) == 1 &&
            fwrite(table.data(), 2, table.size(), f) == table.size() && fwrite(mlp.data(), 2, mlp.size(), f) == mlp.size() &&
            fwrite(occ.data(), 4, occ.size(), f) == occ.size();
  ok = (fclose(f) == 0) && ok;
  return ok ? PRV_OK : fail(c, PRV_E_IO, "short write to %s", path);
} catch (...) { return caught(c); }

int prv_model_load_file(prv_ctx* c, int slot, const char* path) try {
  if (!c) return PRV_E_INVALID;
  if (!path) return fail(c, PRV_E_INVALID, "path is NULL");
  FILE* f = fopen(path, "rb");
  if (!f) return fail(c, PRV_E_IO, "cannot read %s", path);
  uint32_t head[2] = {0, 0};
  prv_field_desc d;
  uint64_t th = 0, mh = 0, ow = 0;
  if (fread(head, sizeof(head), 1, f) != 1 || head[0] != 0x46565250u || head[1] != kModelFileVersion ||
      fread(&d, sizeof(d), 1, f) != 1 || prv_model_sizes(&d, &th, &mh, &ow) != PRV_OK) {
    fclose(f);
    return fail(c, PRV_E_IO, "%s is not a PRVF model file of format version %u", path, kModelFileVersion);
  }
  { // the descriptor is input, not truth: the file must have the size it implies BEFORE buffers of that size exist
    const long at = ftell(f);
    const bool seek_ok = at >= 0 && fseek(f, 0, SEEK_END) == 0;
    const long end = seek_ok ? ftell(f) : -1;
    if (!seek_ok || end < at || fseek(f, at, SEEK_SET) != 0 ||
        (uint64_t)(end - at) != th * 2ull + mh * 2ull + ow * 4ull) {
      fclose(f);
      return fail(c, PRV_E_IO, "%s is truncated (or not the field its header describes)", path);
    }
  }
  std::vector<uint16_t> table(th), mlp(mh);
  std::vector<uint32_t> occ(ow);
  const bool ok = fread(table.data(), 2, th, f) == th && fread(mlp.data(), 2, mh, f) == mh && fread(occ.data(), 4, ow, f) == ow;
  fclose(f);
  if (!ok) return fail(c, PRV_E_IO, "%s is truncated", path);
  return prv_model_load(c, slot, &d, table.data(), mlp.data(), occ.data());
} catch (...) { return caught(c); }

int prv_model_load_ingp(prv_ctx* c, int slot, const char* path) try {
  if (!c) return PRV_E_INVALID;
  if (!path) return fail(c, PRV_E_INVALID, "path is NULL");
  prvingp::Field f;
  std::string err;
  const int rc = prvingp::read_snapshot(path, f, err);
  if (rc != 0) return fail(c, rc, "instant-ngp snapshot %s: %s", path, err.c_str());
  return prv_model_load(c, slot, &f.desc, f.table.data(), f.mlp.data(), f.occ.data());
} catch (...) { return caught(c); }

int prv_model_save_ingp(prv_ctx* c, int slot, const char* path) try {
  if (!c) return PRV_E_INVALID;
  int rc = check_model(c, slot);
  if (rc != PRV_OK) return rc;
  if (!path) return fail(c, PRV_E_INVALID, "path is NULL");
  const Model& m = c->models[slot];
  prvingp::Field f;
  f.desc = m.desc;
  f.table.resize(m.table_halfs);
  f.mlp.resize(PRV_MLP_HALFS);
  f.occ.resize(m.occ_words);
  if ((rc = prv_model_export(c, slot, f.table.data(), f.mlp.data(), f.occ.data())) != PRV_OK) return rc;
  std::string err;
  rc = prvingp::write_snapshot(path, f, err);
  return rc == 0 ? PRV_OK : fail(c, rc, "instant-ngp snapshot %s: %s", path, err.c_str());
} catch (...) { return caught(c); }

int prv_model_desc(prv_ctx* c, int slot, prv_field_desc* out) try {
  if (!c) return PRV_E_INVALID;
  int rc = model_present(c, slot);
  if (rc != PRV_OK) return rc;
  if (!out) return fail(c, PRV_E_INVALID, "out is NULL");
  *out = c->models[slot].desc;
  return PRV_OK;
} catch (...) { return caught(c); }

// ------------------------------------------------------------------ cameras

// transform_matrix (NeRF convention, as written at main.cpp:1626-1641) -> engine frame:
// negate columns 1 and 2, t*scale + offset, cycle rows (y,z,x).  ASSUMED from upstream
// instant-ngp (not in the reference tree).
static int build_camset(prv_ctx* c, const double* tm, int n, const prv_intrinsics& in, bool dataset, double scale,
                        const double offset[3], prv_camset** out) {
  if (!out) return fail(c, PRV_E_INVALID, "out is NULL");
  *out = nullptr;
  if (n < 0 || (n > 0 && !tm)) return fail(c, PRV_E_INVALID, "bad matrix array");
  if (in.w < 1 || in.h < 1) return fail(c, PRV_E_INVALID, "bad size %dx%d", in.w, in.h);
  if (!(in.fl_x > 0.0) || !(in.fl_y > 0.0) || !std::isfinite(in.fl_x) || !std::isfinite(in.fl_y))
    return fail(c, PRV_E_INVALID, "bad focal length %g, %g", in.fl_x, in.fl_y);
  if (!std::isfinite(in.cx) || !std::isfinite(in.cy) || !std::isfinite(in.k1) || !std::isfinite(in.k2) ||
      !std::isfinite(in.p1) || !std::isfinite(in.p2))
    return fail(c, PRV_E_INVALID, "non-finite intrinsics");
  std::unique_ptr<prv_camset> owner(new prv_camset());
  prv_camset* cs = owner.get();
  cs->device = c->device;
  cs->width = in.w;
  cs->height = in.h;
  cs->dataset = dataset;
  const double off[3] = {offset ? offset[0] : 0.5, offset ? offset[1] : 0.5, offset ? offset[2] : 0.5};
  static const int src_row[3] = {1, 2, 0};
  cs->cams.resize(n);
  for (int i = 0; i < n; i++) {
    const double* m = tm + (size_t)i * 16;
    double e[3][4];
    for (int r = 0; r < 3; r++) {
      e[r][0] = m[r * 4 + 0];
      e[r][1] = -m[r * 4 + 1];
      e[r][2] = -m[r * 4 + 2];
      e[r][3] = m[r * 4 + 3] * scale + off[r];
    }
    CamDev& cam = cs->cams[i];
    for (int r = 0; r < 3; r++)
      for (int k = 0; k < 4; k++) cam.c2w[r * 4 + k] = (float)e[src_row[r]][k];
    cam.fx = (float)in.fl_x;
    cam.fy = (float)in.fl_y;
    cam.cx = (float)in.cx;
    cam.cy = (float)in.cy;
    cam.lens[0] = (float)in.k1;
    cam.lens[1] = (float)in.k2;
    cam.lens[2] = (float)in.p1;
    cam.lens[3] = (float)in.p2;
  }
  *out = owner.release();
  return PRV_OK;
}

int prv_cameras_from_matrices(prv_ctx* c, const double* tm, int n, double camera_angle_x, int width, int height,
                              double scale, const double offset[3], prv_camset** out) try {
  if (!c) return PRV_E_INVALID;
  if (out) *out = nullptr;
  if (width < 1 || height < 1) return fail(c, PRV_E_INVALID, "bad size %dx%d", width, height);
  if (!(camera_angle_x > 0.0 && camera_angle_x < M_PI)) return fail(c, PRV_E_INVALID, "bad camera_angle_x %g", camera_angle_x);
  prv_intrinsics in{};
  in.fl_x = in.fl_y = (double)(float)(0.5 * (double)width / std::tan(0.5 * camera_angle_x));
  in.cx = 0.5 * width;
  in.cy = 0.5 * height;
  in.w = width;
  in.h = height;
  return build_camset(c, tm, n, in, false, scale, offset, out);
} catch (...) { return caught(c); }

int prv_cameras_from_matrices_intr(prv_ctx* c, const double* tm, int n, const prv_intrinsics* intr, double scale,
                                   const double offset[3], prv_camset** out) try {
  if (!c) return PRV_E_INVALID;
  if (out) *out = nullptr;
  if (!intr) return fail(c, PRV_E_INVALID, "intrinsics are NULL");
  return build_camset(c, tm, n, *intr, true, scale, offset, out);
} catch (...) { return caught(c); }

// header + frames of a transforms.json; dataset = use the file's own intrinsics block
static int cameras_from_json(prv_ctx* c, const char* path, bool dataset, prv_camset** out) {
  if (!c) return PRV_E_INVALID;
  if (!out || !path) return fail(c, PRV_E_INVALID, "NULL argument");
  *out = nullptr;
  std::string text, err;
  if (!prvjson::read_file(path, text)) return fail(c, PRV_E_IO, "cannot read %s", path);
  prvjson::Value root;
  if (!prvjson::Parser(text).parse(root, err)) return fail(c, PRV_E_IO, "%s: %s", path, err.c_str());
  if (!root.has("frames") || !root.has("w") || !root.has("h")) return fail(c, PRV_E_IO, "%s: missing w / h / frames", path);
  if (!dataset && !root.has("camera_angle_x")) return fail(c, PRV_E_IO, "%s: missing camera_angle_x", path);
  const double scale = root.has("scale") ? root.at("scale").number() : 0.33;
  double offset[3] = {0.5, 0.5, 0.5};
  if (root.has("offset") && root.at("offset").arr.size() == 3)
    for (int k = 0; k < 3; k++) offset[k] = root.at("offset").arr[k].number();
  const auto& frames = root.at("frames").arr;
  std::vector<double> tm(frames.size() * 16);
  for (size_t i = 0; i < frames.size(); i++) {
    if (!frames[i].has("transform_matrix")) return fail(c, PRV_E_IO, "%s: frame %zu has no transform_matrix", path, i);
    const auto& M = frames[i].at("transform_matrix");
    if (M.arr.size() != 4) return fail(c, PRV_E_IO, "%s: frame %zu has no 4x4 transform_matrix", path, i);
    for (int r = 0; r < 4; r++) {
      if (M.arr[r].arr.size() != 4) return fail(c, PRV_E_IO, "%s: frame %zu row %d is not 4 wide", path, i, r);
      for (int k = 0; k < 4; k++) tm[i * 16 + r * 4 + k] = M.arr[r].arr[k].number();
    }
  }
  const int w = (int)root.at("w").number(), h = (int)root.at("h").number();
  if (!dataset)
    return prv_cameras_from_matrices(c, tm.data(), (int)frames.size(), root.at("camera_angle_x").number(), w, h, scale,
                                     offset, out);
  auto num = [&](const char* k, double dflt) { return root.has(k) ? root.at(k).number() : dflt; };
  prv_intrinsics in{};
  in.w = w;
  in.h = h;
  if (root.has("fl_x")) in.fl_x = root.at("fl_x").number();
  else if (root.has("camera_angle_x")) in.fl_x = 0.5 * w / std::tan(0.5 * root.at("camera_angle_x").number());
  if (root.has("fl_y")) in.fl_y = root.at("fl_y").number();
  else if (root.has("camera_angle_y")) in.fl_y = 0.5 * h / std::tan(0.5 * root.at("camera_angle_y").number());
  else in.fl_y = in.fl_x;
  if (!(in.fl_x > 0.0)) in.fl_x = in.fl_y;
  if (!(in.fl_x > 0.0)) return fail(c, PRV_E_IO, "%s: no fl_x / fl_y / camera_angle_*", path);
  in.cx = num("cx", 0.5 * w);
  in.cy = num("cy", 0.5 * h);
  in.k1 = num("k1", 0.0);
  in.k2 = num("k2", 0.0);
  in.p1 = num("p1", 0.0);
  in.p2 = num("p2", 0.0);
  return build_camset(c, tm.data(), (int)frames.size(), in, true, scale, offset, out);
}

int prv_cameras_from_json(prv_ctx* c, const char* path, prv_camset** out) try {
  return cameras_from_json(c, path, false, out);
} catch (...) { return caught(c); }
int prv_cameras_from_dataset_json(prv_ctx* c, const char* path, prv_camset** out) try {
  return cameras_from_json(c, path, true, out);
} catch (...) { return caught(c); }

int prv_camset_lens(const prv_camset* cs, int i, float lens[4]) {
  if (!cs || i < 0 || i >= (int)cs->cams.size() || !lens) return PRV_E_INVALID;
  memcpy(lens, cs->cams[i].lens, sizeof(float) * 4);
  return PRV_OK;
}

int prv_camset_count(const prv_camset* cs) { return cs ? (int)cs->cams.size() : PRV_E_INVALID; }
int prv_camset_size(const prv_camset* cs, int* w, int* h) {
  if (!cs) return PRV_E_INVALID;
  if (w) *w = cs->width;
  if (h) *h = cs->height;
  return PRV_OK;
}
int prv_camset_get(const prv_camset* cs, int i, float c2w[12], float intr[4]) {
  if (!cs || i < 0 || i >= (int)cs->cams.size()) return PRV_E_INVALID;
  if (c2w) memcpy(c2w, cs->cams[i].c2w, sizeof(float) * 12);
  if (intr) {
    intr[0] = cs->cams[i].fx;
    intr[1] = cs->cams[i].fy;
    intr[2] = cs->cams[i].cx;
    intr[3] = cs->cams[i].cy;
  }
  return PRV_OK;
}
void prv_camset_destroy(prv_camset* cs) { delete cs; }

// ------------------------------------------------------------------ render

int prv_render(prv_ctx* c, int slot, const prv_camset* cs, const int* view_ids, int n_views, const prv_render_opts* o,
               float* out, prv_stats* st) try {
  if (!c) return PRV_E_INVALID;
  int rc;
  if ((rc = check_model(c, slot)) != PRV_OK || (rc = check_opts(c, o)) != PRV_OK) return rc;
  if (!cs || n_views < 0 || (!out && n_views > 0)) return fail(c, PRV_E_INVALID, "bad camset / view count / output");
  HIPCHK(c, hipSetDevice(c->device));
  if ((rc = check_device_ptr(c, out, "out_rgba_dev")) != PRV_OK) return rc;
  if ((rc = render_views(c, slot, cs, view_ids, n_views, o, out, nullptr, true)) != PRV_OK) return rc;
  return fetch_stats(c, o, n_views, 1, st);
} catch (...) { return caught(c); }

int prv_render_rgba8(prv_ctx* c, int slot, const prv_camset* cs, const int* view_ids, int n_views,
                     const prv_render_opts* o, uint8_t* out, prv_stats* st) try {
  if (!c) return PRV_E_INVALID;
  int rc;
  if ((rc = check_model(c, slot)) != PRV_OK || (rc = check_opts(c, o)) != PRV_OK) return rc;
  if (!cs || n_views < 0 || (!out && n_views > 0)) return fail(c, PRV_E_INVALID, "bad camset / view count / output");
  HIPCHK(c, hipSetDevice(c->device));
  if ((rc = check_device_ptr(c, out, "out_rgba8_dev")) != PRV_OK) return rc;
  const size_t npix = (size_t)o->width * o->height;
  if ((rc = ensure(c, c->img_f32, std::max<size_t>(16, (size_t)n_views * npix * 16))) != PRV_OK) return rc;
  if ((rc = render_views(c, slot, cs, view_ids, n_views, o, (float*)c->img_f32.p, out, true)) != PRV_OK) return rc;
  return fetch_stats(c, o, n_views, 1, st);
} catch (...) { return caught(c); }

int prv_first_hit(prv_ctx* c, int slot, const prv_camset* cs, const int* view_ids, int n_views, int W, int H,
                  float max_range, int32_t* out) try {
  if (!c) return PRV_E_INVALID;
  int rc;
  if ((rc = check_model(c, slot)) != PRV_OK) return rc;
  if (!cs || n_views < 0 || W < 1 || H < 1 || W > 16384 || H > 16384 || (!out && n_views > 0))
    return fail(c, PRV_E_INVALID, "bad argument");
  if (n_views == 0) return PRV_OK;
  HIPCHK(c, hipSetDevice(c->device));
  if ((rc = check_device_ptr(c, out, "out_voxel_dev")) != PRV_OK) return rc;
  std::vector<CamDev> cams(n_views);
  for (int i = 0; i < n_views; i++) {
    const int v = view_ids ? view_ids[i] : i;
    if (v < 0 || v >= (int)cs->cams.size()) return fail(c, PRV_E_INVALID, "view id %d out of range", v);
    cams[i] = cam_at(cs, v, W, H);
  }
  if ((rc = ensure(c, c->view_ids, (size_t)n_views * (sizeof(CamDev) + sizeof(int)))) != PRV_OK) return rc;
  HIPCHK(c, hipMemcpyAsync(c->view_ids.p, cams.data(), (size_t)n_views * sizeof(CamDev), hipMemcpyHostToDevice, c->stream));
  HIPCHK(c, hipStreamSynchronize(c->stream));
  HIPCHK(c, launch_first_hit(c->models[slot].dev, (const CamDev*)c->view_ids.p, n_views, W, H, max_range, out, c->stream));
  return PRV_OK;
} catch (...) { return caught(c); }

int prv_splat_points(prv_ctx* c, const float* xyz, const uint8_t* rgb, size_t n, double scale, const double offset[3],
                     const prv_camset* cs, const int* view_ids, int n_views, int W, int H, int point_size, int flip180,
                     uint8_t* out) try {
  if (!c) return PRV_E_INVALID;
  if (!cs || n_views < 0 || W < 1 || H < 1 || W > 16384 || H > 16384 || (!out && n_views > 0) || (n > 0 && (!xyz || !rgb)))
    return fail(c, PRV_E_INVALID, "bad argument");
  if (point_size < 1 || point_size > 64) return fail(c, PRV_E_INVALID, "point_size must be in [1,64], got %d", point_size);
  if (n_views == 0) return PRV_OK;
  HIPCHK(c, hipSetDevice(c->device));
  int rc;
  if ((rc = check_device_ptr(c, xyz, "xyz_dev")) != PRV_OK || (rc = check_device_ptr(c, rgb, "rgb_dev")) != PRV_OK ||
      (rc = check_device_ptr(c, out, "out_rgba8_dev")) != PRV_OK)
    return rc;
  std::vector<CamDev> cams(n_views);
  for (int i = 0; i < n_views; i++) {
    const int v = view_ids ? view_ids[i] : i;
    if (v < 0 || v >= (int)cs->cams.size()) return fail(c, PRV_E_INVALID, "view id %d out of range", v);
    cams[i] = cam_at(cs, v, W, H);
  }
  if ((rc = ensure(c, c->view_ids, (size_t)n_views * (sizeof(CamDev) + sizeof(int)))) != PRV_OK) return rc;
  if ((rc = ensure(c, c->stage, (size_t)n_views * W * H * 8)) != PRV_OK) return rc;
  HIPCHK(c, hipMemcpyAsync(c->view_ids.p, cams.data(), (size_t)n_views * sizeof(CamDev), hipMemcpyHostToDevice, c->stream));
  HIPCHK(c, hipStreamSynchronize(c->stream));
  const float off[3] = {(float)(offset ? offset[0] : 0.5), (float)(offset ? offset[1] : 0.5), (float)(offset ? offset[2] : 0.5)};
  HIPCHK(c, launch_splat_points(xyz, rgb, n, (float)scale, off, (const CamDev*)c->view_ids.p, n_views, W, H, point_size,
                                flip180, (unsigned long long*)c->stage.p, (uint32_t*)out, c->stream));
  return PRV_OK;
} catch (...) { return caught(c); }

int prv_precept(prv_ctx* c, int slot, const float* voxels, int n, const double c2w[16], const prv_rs2_intrinsics* k,
                float max_range, int32_t* out) try {
  if (!c) return PRV_E_INVALID;
  int rc;
  if ((rc = check_model(c, slot)) != PRV_OK) return rc;
  if (n < 0 || !c2w || !k || (n > 0 && (!voxels || !out))) return fail(c, PRV_E_INVALID, "bad argument");
  if (n == 0) return PRV_OK;
  HIPCHK(c, hipSetDevice(c->device));
  if ((rc = check_device_ptr(c, voxels, "voxels_dev")) != PRV_OK || (rc = check_device_ptr(c, out, "out_voxel_dev")) != PRV_OK) return rc;
  PreceptPose pose;
  memcpy(pose.c2w, c2w, sizeof(pose.c2w));
  { // w2c = inverse of the rigid c2w, in double (view_pose_world.inverse(), main.cpp:243)
    double a[4][8];
    for (int i = 0; i < 4; i++)
      for (int j = 0; j < 4; j++) {
        a[i][j] = c2w[i * 4 + j];
        a[i][4 + j] = i == j;
      }
    for (int col = 0; col < 4; col++) {
      int piv = col;
      for (int r = col + 1; r < 4; r++)
        if (std::fabs(a[r][col]) > std::fabs(a[piv][col])) piv = r;
      if (std::fabs(a[piv][col]) < 1e-300) return fail(c, PRV_E_INVALID, "singular camera pose");
      for (int j = 0; j < 8; j++) std::swap(a[col][j], a[piv][j]);
      const double dd = a[col][col];
      for (int j = 0; j < 8; j++) a[col][j] /= dd;
      for (int r = 0; r < 4; r++)
        if (r != col) {
          const double f = a[r][col];
          for (int j = 0; j < 8; j++) a[r][j] -= f * a[col][j];
        }
    }
    for (int i = 0; i < 4; i++)
      for (int j = 0; j < 4; j++) pose.w2c[i * 4 + j] = a[i][4 + j];
  }
  Rs2Intr in;
  in.ppx = k->ppx; in.ppy = k->ppy; in.fx = k->fx; in.fy = k->fy;
  for (int i = 0; i < 5; i++) in.c[i] = k->coeffs[i];
  in.width = k->width; in.height = k->height; in.model = k->model;
  HIPCHK(c, launch_precept(c->models[slot].dev, voxels, n, pose, in, max_range, out, c->stream));
  return PRV_OK;
} catch (...) { return caught(c); }

int prv_quantize_rgba8(prv_ctx* c, const float* rgba, size_t n, const float bg[4], uint8_t* out) try {
  if (!c) return PRV_E_INVALID;
  if (!rgba || !out || !bg) return fail(c, PRV_E_INVALID, "NULL argument");
  HIPCHK(c, hipSetDevice(c->device));
  int rc;
  if ((rc = check_device_ptr(c, rgba, "rgba_dev")) != PRV_OK || (rc = check_device_ptr(c, out, "out_rgba8_dev")) != PRV_OK) return rc;
  if (n) HIPCHK(c, launch_quantize(rgba, n, bg, out, c->stream));
  return PRV_OK;
} catch (...) { return caught(c); }

// ------------------------------------------------------------------ scores

static int score_ensemble_dev(prv_ctx* c, int method, const uint8_t* const* imgs, int E, int n_views, size_t npix,
                              prv_score_record* rec_dev) {
  // addends of a batch of views (8 B x 3 or 2 per pixel), then one sequential sum per view; batches bound the buffer
  const size_t per_view = npix * (method == PRV_SCORE_ENSEMBLE_RGB ? 3 : 2) * sizeof(double);
  const int batch = (int)std::min<size_t>((size_t)n_views, std::max<size_t>(1, ((size_t)1 << 30) / per_view));
  const int nblk = (int)std::min<size_t>(1024, std::max<size_t>(1, (npix + 255) / 256));
  int rc;
  if ((rc = ensure(c, c->partial, (size_t)batch * per_view)) != PRV_OK) return rc;
  EnsembleParams P;
  memset(&P, 0, sizeof(P));
  for (int e = 0; e < E; e++) P.imgs[e] = (const uint32_t*)imgs[e];
  P.E = E;
  P.pixels_per_view = npix;
  P.partial = (double*)c->partial.p;
  for (int v0 = 0; v0 < n_views; v0 += batch) {
    P.view0 = v0;
    HIPCHK(c, launch_score_ensemble(P, method, std::min(batch, n_views - v0), nblk, rec_dev, c->stream));
  }
  return PRV_OK;
}

static int score_psnr_dev(prv_ctx* c, const float* rgba, const float* gt, int n_views, size_t npix, const float bg[4],
                          double coverage_weight, prv_score_record* rec_dev, const CamDev* cams_dev = nullptr, int W = 0) {
  const int nblk = score_blocks(npix);
  int rc;
  if ((rc = ensure(c, c->partial, (size_t)n_views * nblk * 3 * sizeof(double))) != PRV_OK) return rc;
  PsnrParams P;
  memset(&P, 0, sizeof(P));
  P.rgba = rgba;
  P.gt = gt;
  P.pixels_per_view = npix;
  P.cams = cams_dev; // the render of a fused round wrote only inside the views' cull rectangles (render_views: private_output)
  P.W = W;
  memcpy(P.bg, bg, sizeof(P.bg));
  P.partial = (double*)c->partial.p;
  HIPCHK(c, launch_score_psnr(P, n_views, nblk, c->stream));
  HIPCHK(c, launch_score_finalize(P.partial, n_views, nblk, PRV_SCORE_PSNR_COVERAGE, npix, coverage_weight, rec_dev, c->stream));
  return PRV_OK;
}

int prv_score_ensemble_images(prv_ctx* c, int method, const uint8_t* const* imgs, int E, int n_views, size_t npix,
                              prv_score_record* rec_host) try {
  if (!c) return PRV_E_INVALID;
  if (method != PRV_SCORE_ENSEMBLE_RGB && method != PRV_SCORE_ENSEMBLE_RGB_DENSITY)
    return fail(c, PRV_E_INVALID, "method %d is not an ensemble score", method);
  if (!imgs || E < 1 || E > PRV_MAX_MODELS || n_views < 0 || !rec_host) return fail(c, PRV_E_INVALID, "bad argument");
  if (n_views == 0) return PRV_OK;
  if (npix == 0) return fail(c, PRV_E_INVALID, "empty images");
  HIPCHK(c, hipSetDevice(c->device));
  int rc;
  for (int e = 0; e < E; e++)
    if (!imgs[e]) return fail(c, PRV_E_INVALID, "image set %d is NULL", e);
    else if ((rc = check_device_ptr(c, imgs[e], "rgba8_dev")) != PRV_OK) return rc;
  if ((rc = ensure(c, c->records, (size_t)n_views * sizeof(prv_score_record))) != PRV_OK) return rc;
  if ((rc = score_ensemble_dev(c, method, imgs, E, n_views, npix, (prv_score_record*)c->records.p)) != PRV_OK) return rc;
  return prv_memcpy_d2h(c, rec_host, c->records.p, (size_t)n_views * sizeof(prv_score_record));
} catch (...) { return caught(c); }

int prv_score_psnr_images(prv_ctx* c, const float* rgba, const float* gt, int n_views, size_t npix, const float bg[4],
                          prv_score_record* rec_host) try {
  if (!c) return PRV_E_INVALID;
  if (!rgba || !gt || !bg || n_views < 0 || !rec_host) return fail(c, PRV_E_INVALID, "bad argument");
  if (n_views == 0) return PRV_OK;
  if (npix == 0) return fail(c, PRV_E_INVALID, "empty images");
  HIPCHK(c, hipSetDevice(c->device));
  int rc;
  if ((rc = check_device_ptr(c, rgba, "rgba_dev")) != PRV_OK || (rc = check_device_ptr(c, gt, "gt_rgba_dev")) != PRV_OK) return rc;
  if ((rc = ensure(c, c->records, (size_t)n_views * sizeof(prv_score_record))) != PRV_OK) return rc;
  if ((rc = score_psnr_dev(c, rgba, gt, n_views, npix, bg, c->coverage_weight, (prv_score_record*)c->records.p)) != PRV_OK) return rc;
  return prv_memcpy_d2h(c, rec_host, c->records.p, (size_t)n_views * sizeof(prv_score_record));
} catch (...) { return caught(c); }

int prv_evaluate_images(prv_ctx* c, const float* rgba, const float* gt, int n_views, int W, int H, const float bg[4],
                        double* psnr_host, double* ssim_host) try {
  if (!c) return PRV_E_INVALID;
  if (!rgba || !gt || !bg || n_views < 0 || W < 5 || H < 5) return fail(c, PRV_E_INVALID, "bad argument (images must be at least 5x5)");
  if (n_views == 0) return PRV_OK;
  HIPCHK(c, hipSetDevice(c->device));
  const size_t npix = (size_t)W * H;
  int rc;
  if ((rc = check_device_ptr(c, rgba, "rgba_dev")) != PRV_OK || (rc = check_device_ptr(c, gt, "gt_rgba_dev")) != PRV_OK) return rc;
  if (psnr_host) {
    std::vector<prv_score_record> rec(n_views);
    if ((rc = ensure(c, c->records, (size_t)n_views * sizeof(prv_score_record))) != PRV_OK) return rc;
    // coverage weight 0: the score is then exactly -psnr in double (the record's psnr field is a float)
    if ((rc = score_psnr_dev(c, rgba, gt, n_views, npix, bg, 0.0, (prv_score_record*)c->records.p)) != PRV_OK) return rc;
    if ((rc = prv_memcpy_d2h(c, rec.data(), c->records.p, (size_t)n_views * sizeof(prv_score_record))) != PRV_OK) return rc;
    for (int i = 0; i < n_views; i++) psnr_host[i] = -rec[i].score;
  }
  if (ssim_host) {
    const int nblk = score_blocks(npix);
    if ((rc = ensure(c, c->dbg[0], (size_t)n_views * npix * 4)) != PRV_OK || (rc = ensure(c, c->dbg[1], (size_t)n_views * npix * 4)) != PRV_OK ||
        (rc = ensure(c, c->partial, (size_t)n_views * nblk * 2 * sizeof(double))) != PRV_OK ||
        (rc = ensure(c, c->dbg[2], (size_t)n_views * sizeof(double))) != PRV_OK)
      return rc;
    HIPCHK(c, launch_ssim(rgba, gt, n_views, W, H, bg, (float*)c->dbg[0].p, (float*)c->dbg[1].p, (double*)c->partial.p, nblk,
                          (double*)c->dbg[2].p, c->stream));
    if ((rc = prv_memcpy_d2h(c, ssim_host, c->dbg[2].p, (size_t)n_views * sizeof(double))) != PRV_OK) return rc;
  }
  return PRV_OK;
} catch (...) { return caught(c); }

int prv_evaluate(prv_ctx* c, int slot, const prv_camset* cs, const int* view_ids, int n_views, const prv_render_opts* o,
                 const float* gt, double* mean_psnr, double* mean_ssim) try {
  if (!c) return PRV_E_INVALID;
  int rc;
  if ((rc = check_model(c, slot)) != PRV_OK || (rc = check_opts(c, o)) != PRV_OK) return rc;
  if (!cs || !gt || n_views < 1) return fail(c, PRV_E_INVALID, "bad argument");
  HIPCHK(c, hipSetDevice(c->device));
  if ((rc = check_device_ptr(c, gt, "gt_rgba_dev")) != PRV_OK) return rc;
  const size_t npix = (size_t)o->width * o->height;
  if ((rc = ensure(c, c->img_f32, (size_t)n_views * npix * 16)) != PRV_OK) return rc;
  if ((rc = render_views(c, slot, cs, view_ids, n_views, o, (float*)c->img_f32.p, nullptr, true)) != PRV_OK) return rc;
  std::vector<double> ps(n_views), ss(n_views);
  if ((rc = prv_evaluate_images(c, (const float*)c->img_f32.p, gt, n_views, o->width, o->height, o->background, ps.data(), ss.data())) != PRV_OK)
    return rc;
  double tp = 0, ts = 0; // totpsnr / totssim of run.py:261-264, in image order
  for (int i = 0; i < n_views; i++) {
    tp += ps[i];
    ts += ss[i];
  }
  if (mean_psnr) *mean_psnr = tp / n_views;
  if (mean_ssim) *mean_ssim = ts / n_views;
  return PRV_OK;
} catch (...) { return caught(c); }

int prv_score_views(prv_ctx* c, int method, const int* model_slots, int n_models, const prv_camset* cs,
                    const int* view_ids, int n_views, const prv_render_opts* o, const float* gt,
                    prv_score_record* rec_host, prv_score_record* rec_dev, prv_stats* st) try {
  if (!c) return PRV_E_INVALID;
  int rc;
  if ((rc = check_opts(c, o)) != PRV_OK) return rc;
  if (!cs || !model_slots || n_views < 0) return fail(c, PRV_E_INVALID, "bad argument");
  const bool ens = method == PRV_SCORE_ENSEMBLE_RGB || method == PRV_SCORE_ENSEMBLE_RGB_DENSITY;
  if (!ens && method != PRV_SCORE_PSNR_COVERAGE) return fail(c, PRV_E_INVALID, "unknown score method %d", method);
  if (ens && (n_models < 1 || n_models > PRV_MAX_MODELS)) return fail(c, PRV_E_INVALID, "ensemble size %d", n_models);
  if (!ens && (n_models != 1 || !gt)) return fail(c, PRV_E_INVALID, "method 5 needs one model and reference images");
  for (int e = 0; e < n_models; e++)
    if ((rc = check_model(c, model_slots[e])) != PRV_OK) return rc;
  HIPCHK(c, hipSetDevice(c->device));
  if ((rc = check_device_ptr(c, gt, "gt_rgba_dev")) != PRV_OK || (rc = check_device_ptr(c, rec_dev, "records_dev")) != PRV_OK) return rc;
  const size_t npix = (size_t)o->width * o->height;
  if ((rc = ensure(c, c->records, std::max<size_t>(16, (size_t)n_views * sizeof(prv_score_record)))) != PRV_OK) return rc;
  prv_score_record* rec = (prv_score_record*)c->records.p;
  if ((rc = ensure(c, c->img_f32, std::max<size_t>(16, (size_t)n_views * npix * 16))) != PRV_OK) return rc;
  if (ens) {
    const uint8_t* imgs[PRV_MAX_MODELS];
    uint8_t* out8[PRV_MAX_MODELS];
    for (int e = 0; e < n_models; e++) {
      if ((rc = ensure(c, c->img_u8[e], std::max<size_t>(16, (size_t)n_views * npix * 4))) != PRV_OK) return rc;
      imgs[e] = (const uint8_t*)c->img_u8[e].p;
      out8[e] = (uint8_t*)c->img_u8[e].p;
    }
    bool one_march = false; // the engine's rule, an ensemble size with an instance: ONE march launch for all members
    if ((rc = render_ensemble_ngp(c, model_slots, n_models, cs, view_ids, n_views, o, (float*)c->img_f32.p, out8, &one_march)) != PRV_OK) return rc;
    for (int e = 0; e < n_models && !one_march; e++)
      if ((rc = render_views(c, model_slots[e], cs, view_ids, n_views, o, (float*)c->img_f32.p, out8[e], e == 0)) != PRV_OK) return rc;
    if (n_views && (rc = score_ensemble_dev(c, method, imgs, n_models, n_views, npix, rec)) != PRV_OK) return rc;
  } else {
    // the image is a private temporary of the round: only the tiles inside each view's cull rectangle are rendered and
    // the score reads it through the same rectangles (view i of this call = camera i of the upload in c->view_ids)
    const bool priv = o->spp == 1;
    if ((rc = render_views(c, model_slots[0], cs, view_ids, n_views, o, (float*)c->img_f32.p, nullptr, true, priv)) != PRV_OK)
      return rc;
    if (n_views && (rc = score_psnr_dev(c, (const float*)c->img_f32.p, gt, n_views, npix, o->background, c->coverage_weight, rec,
                                        priv ? (const CamDev*)c->view_ids.p : nullptr, o->width)) != PRV_OK)
      return rc;
  }
  if (rec_dev && n_views)
    HIPCHK(c, hipMemcpyAsync(rec_dev, rec, (size_t)n_views * sizeof(prv_score_record), hipMemcpyDeviceToDevice, c->stream));
  if (rec_host && n_views) {
    HIPCHK(c, hipMemcpyAsync(rec_host, rec, (size_t)n_views * sizeof(prv_score_record), hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
  }
  return fetch_stats(c, o, n_views, n_models, st);
} catch (...) { return caught(c); }

// arg-max rule of main.cpp:1971-1972, 2088-2091: ascending ids, strict '>', start at -1e100
int prv_argmax(const prv_score_record* r, const int* ids, int n) {
  if (!r || !ids) return -1;
  double best = -1e100;
  int best_id = -1;
  for (int i = 0; i < n; i++)
    if (r[i].score > best) {
      best = r[i].score;
      best_id = ids[i];
    }
  return best_id;
}

int prv_rank(const prv_score_record* r, const int* ids, int n, int* order) try {
  if (n < 0 || (n > 0 && (!r || !ids || !order))) return PRV_E_INVALID;
  std::vector<int> idx(n);
  for (int i = 0; i < n; i++) idx[i] = i;
  // a strict weak order also when a score is NaN (a diverged ensemble member produces exactly that): NaN ranks
  // after every number -- the arg-max rule's strict '>' never picks it either -- and NaNs tie among themselves
  std::stable_sort(idx.begin(), idx.end(), [&](int a, int b) {
    const bool na = std::isnan(r[a].score), nb = std::isnan(r[b].score);
    if (na != nb) return nb;
    if (!na && r[a].score != r[b].score) return r[a].score > r[b].score;
    return ids[a] < ids[b];
  });
  for (int i = 0; i < n; i++) order[i] = ids[idx[i]];
  return PRV_OK;
} catch (...) { return caught(nullptr); }

// ------------------------------------------------------------------ stage hooks

int prv_debug_render_clock(prv_ctx* c, uint64_t* shader_cycles, uint64_t* ref_ticks, double* ref_hz) try {
  if (!c) return PRV_E_INVALID;
  HIPCHK(c, hipSetDevice(c->device)); // the calling thread may be a new one (its current device would be 0)
  if (!shader_cycles || !ref_ticks || !ref_hz) return fail(c, PRV_E_INVALID, "NULL output");
  if (!c->counters.p) return fail(c, PRV_E_STATE, "nothing rendered yet");
  unsigned long long v[2] = {0, 0};
  HIPCHK(c, hipMemcpyAsync(v, (char*)c->counters.p + kStatOffset + 16, 16, hipMemcpyDeviceToHost, c->stream));
  HIPCHK(c, hipStreamSynchronize(c->stream));
  int khz = 0;
  HIPCHK(c, hipDeviceGetAttribute(&khz, hipDeviceAttributeWallClockRate, c->device));
  *shader_cycles = v[0];
  *ref_ticks = v[1];
  *ref_hz = (double)khz * 1e3;
  return PRV_OK;
} catch (...) { return caught(c); }

int prv_debug_model_layout(prv_ctx* c, int slot, prv_model_layout* out) try {
  if (!c) return PRV_E_INVALID;
  int rc = check_model(c, slot);
  if (rc != PRV_OK) return rc;
  if (!out) return fail(c, PRV_E_INVALID, "out is NULL");
  const Model& m = c->models[slot];
  HostLevel lv[kMaxLevels];
  uint64_t total = 0;
  compute_levels(m.desc, lv, &total);
  out->table_bytes_canonical = m.table_halfs * 2;
  out->table_bytes_physical = m.phys.bytes;
  out->kernel_features = m.dev.n_features;
  out->kernel_dense_levels = render_instance_dense_levels(m.dev);
  out->kernel_slots = 64;
  out->n_dense_levels = m.dev.n_dense_levels;
  out->n_hashed_levels = 0;
  for (int l = 0; l < m.desc.n_levels; l++) out->n_hashed_levels += lv[l].hashed ? 1 : 0;
  return PRV_OK;
} catch (...) { return caught(c); }

int prv_debug_raygen(prv_ctx* c, const prv_camset* cs, int view, int W, int H, int spp_k, float* o, float* d, float* t) try {
  if (!c) return PRV_E_INVALID;
  if (!cs || view < 0 || view >= (int)cs->cams.size() || W < 1 || H < 1 || !o || !d || !t)
    return fail(c, PRV_E_INVALID, "bad argument");
  HIPCHK(c, hipSetDevice(c->device));
  const size_t n = (size_t)W * H;
  int rc;
  if ((rc = ensure(c, c->dbg[0], n * 12)) != PRV_OK || (rc = ensure(c, c->dbg[1], n * 12)) != PRV_OK ||
      (rc = ensure(c, c->dbg[2], n * 8)) != PRV_OK)
    return rc;
  HIPCHK(c, launch_debug_raygen(cam_at(cs, view, W, H), W, H, spp_k, (float*)c->dbg[0].p,
                                (float*)c->dbg[1].p, (float*)c->dbg[2].p, c->stream));
  HIPCHK(c, hipMemcpyAsync(o, c->dbg[0].p, n * 12, hipMemcpyDeviceToHost, c->stream));
  HIPCHK(c, hipMemcpyAsync(d, c->dbg[1].p, n * 12, hipMemcpyDeviceToHost, c->stream));
  HIPCHK(c, hipMemcpyAsync(t, c->dbg[2].p, n * 8, hipMemcpyDeviceToHost, c->stream));
  HIPCHK(c, hipStreamSynchronize(c->stream));
  return PRV_OK;
} catch (...) { return caught(c); }

static int debug_field_common(prv_ctx* c, int slot, const float* pos, const float* dir, int n, uint16_t* feat,
                              float* out36, int32_t* occ) {
  int rc;
  if ((rc = check_model(c, slot)) != PRV_OK) return rc;
  if (n < 0 || (n > 0 && !pos)) return fail(c, PRV_E_INVALID, "bad argument");
  if (n == 0) return PRV_OK;
  HIPCHK(c, hipSetDevice(c->device));
  const size_t N = (size_t)n;
  if ((rc = ensure(c, c->dbg[0], N * 12)) != PRV_OK || (rc = ensure(c, c->dbg[1], N * 12)) != PRV_OK ||
      (rc = ensure(c, c->dbg[3], N * 64)) != PRV_OK || (rc = ensure(c, c->dbg[4], N * 144)) != PRV_OK ||
      (rc = ensure(c, c->dbg[5], N * 4)) != PRV_OK)
    return rc;
  HIPCHK(c, hipMemcpyAsync(c->dbg[0].p, pos, N * 12, hipMemcpyHostToDevice, c->stream));
  if (dir) HIPCHK(c, hipMemcpyAsync(c->dbg[1].p, dir, N * 12, hipMemcpyHostToDevice, c->stream));
  HIPCHK(c, launch_debug_field(c->models[slot].dev, (const float*)c->dbg[0].p, dir ? (const float*)c->dbg[1].p : nullptr,
                               n, (uint16_t*)c->dbg[3].p, (float*)c->dbg[4].p, (int32_t*)c->dbg[5].p, c->stream));
  if (feat) HIPCHK(c, hipMemcpyAsync(feat, c->dbg[3].p, N * 64, hipMemcpyDeviceToHost, c->stream));
  if (out36) HIPCHK(c, hipMemcpyAsync(out36, c->dbg[4].p, N * 144, hipMemcpyDeviceToHost, c->stream));
  if (occ) HIPCHK(c, hipMemcpyAsync(occ, c->dbg[5].p, N * 4, hipMemcpyDeviceToHost, c->stream));
  HIPCHK(c, hipStreamSynchronize(c->stream));
  return PRV_OK;
}

int prv_debug_encode(prv_ctx* c, int slot, const float* pos, int n, uint16_t* feat) try {
  if (!c) return PRV_E_INVALID;
  if (!feat && n > 0) return fail(c, PRV_E_INVALID, "feat is NULL");
  return debug_field_common(c, slot, pos, nullptr, n, feat, nullptr, nullptr);
} catch (...) { return caught(c); }

int prv_debug_field(prv_ctx* c, int slot, const float* pos, const float* dir, int n, float* out36, int32_t* occ) try {
  if (!c) return PRV_E_INVALID;
  if (n > 0 && (!dir || !out36)) return fail(c, PRV_E_INVALID, "NULL argument");
  return debug_field_common(c, slot, pos, dir, n, nullptr, out36, occ);
} catch (...) { return caught(c); }

} // extern "C"

#include "prv_train_api.inc"
#include "prv_comm_api.inc"
