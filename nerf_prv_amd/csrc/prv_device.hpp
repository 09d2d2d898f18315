// prv_device.hpp -- device-side data layout and the per-lane building blocks of the
// fused render kernel (gfx950 / CDNA4 only; wave = 64 lanes).
//
// Arithmetic contract (shared with oracle/prv_oracle.c, which is the checker, not a
// dependency): all camera / ray / sample-position / interpolation arithmetic is IEEE
// fp32 in a fixed order with fusion only where fmaf() is written (the file is built
// with -ffp-contract=off), so sample positions, grid indices and fp16 features are
// bit-identical to the oracle.  Only the MFMA accumulation order and expf differ.
#pragma once
#include <utility>
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace prv {

constexpr int kMaxLevels = 16;
constexpr int kNumFrags = 24;      // MFMA A-operand fragments of the two MLPs
constexpr int kFragHalfs = 64 * 8; // one fragment: 64 lanes x 8 halfs
constexpr int kTile = 16;          // rays are dealt in 16x16-pixel chunks
constexpr int kChunkRays = kTile * kTile;

typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef _Float16 half4 __attribute__((ext_vector_type(4)));
typedef _Float16 half2v __attribute__((ext_vector_type(2)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

// One level of the PHYSICAL table layout (private to the kernels; the C ABI keeps the
// canonical level-major layout).  Dense levels are stored with power-of-two y/z strides and
// every level starts at a multiple of its own power-of-two size, so the byte offset of
// vertex (x,y,z) is a pure XOR of three per-axis terms:
//   ((x << esh) & m_b) ^ ((y * my_b) & m_b) ^ (((z * mz_b) & m_b) | off_b)
// dense : my_b = ebytes << sx, mz_b = ebytes << 2sx, m_b = ~0      (disjoint bit fields)
// hashed: my_b = ebytes * 2654435761 mod 2^24, mz_b = ebytes * 805459861 mod 2^24, m_b = (T-1) * ebytes < 2^24
// (AND and << distribute over XOR, so this equals ((x ^ y*p1 ^ z*p2) & (T-1)) * ebytes.)
struct LevelDev {
  float scale;     // pos = fmaf(scale, x, 0.5)
  uint32_t res_m1; // vertices per axis - 1 (clamp of the +1 corner)
  uint32_t my_b, mz_b;
  uint32_t m_b;
  uint32_t off_b;  // byte offset of the level, aligned to its size
  uint32_t myz_b;  // dense levels: my_b + mz_b
  // Everything but `scale` in ONE word, for the render kernel (which keeps the level constants in scalar registers and runs
  // out of them): off_b is a multiple of the level's (power-of-two) size, so its low bits are free --
  //   dense : off_b | sx      (y stride = ebytes << sx, z stride = ebytes << 2 sx)
  //   hashed: off_b | res_m1  (res_m1 < 4096; the hash constants are the field's shared three)
  uint32_t pack;
};

struct FieldDev {
  const uint16_t* table;  // PHYSICAL layout (see LevelDev), fp16 bit patterns, entries of F halfs
  const uint32_t* occ;    // occ_res^3 bits, x fastest
  const uint32_t* occ_coarse; // (occ_res/4)^3 bits: any occupied fine cell in the 4^3 block OR its 26 neighbours (or null)
  const half8* frags;     // kNumFrags * 64 half8, MFMA A fragments (prepacked weights) in the trainer's order (prv_train.hip)
  const half8* frags64;   // the render kernel's set (canonical first-layer k order, result copies in padding rows)
  LevelDev levels[kMaxLevels];
  int n_levels, n_features, occ_res;
  int n_dense_levels; // leading levels that are physically dense (clamp-free one-add neighbours)
  // Every hashed level has the same table size T and entry size, hence the same three hash constants (LevelDev::my_b,
  // mz_b, m_b): the render kernel reads THESE for its hashed levels -- three scalar registers for all of them instead of
  // three per level (the kernel keeps every level's constants in SGPRs and was spilling 45..117 of them).
  // hash_shared = 1: they are valid for every hashed level and no hashed level is finer than its table (res <= T, so
  // the x term (x << esh) lies inside the mask by itself).
  uint32_t hash_my_b, hash_mz_b, hash_m_b;
  int hash_shared;
  int wide_offsets; // some hashed level is larger than 16 MiB: its constants are full 32-bit, the (generic) gather multiplies in 32 bits
  float density_bias;
  float occ_lo[3], occ_hi[3]; // bounding box of the occupied cells, grown by one cell (march pass clips to it)
};

struct CamDev {  // engine-frame camera
  float c2w[12]; // row-major 3x4
  float fx, fy, cx, cy;
  float lens[4]; // k1, k2, p1, p2 (OpenCV model on normalised coordinates); all zero = pinhole
  // March pass only (set per render call, prv_api.cpp: set_cull_rect): the pixel rectangle [x0, x1) x [y0, y1) outside
  // which no ray of this view can meet the field's occupied box -- the bounding rectangle of the box's eight projected
  // corners, two pixels of margin.  x1 == 0: not set (lens cameras, a corner behind the camera): nothing is culled.
  int cull[4];
};

// ---------------------------------------------------------------- small helpers

__device__ __forceinline__ float clamp01(float x) { return fminf(fmaxf(x, 0.0f), 1.0f); }

// sub-pixel offset k of spp; k = 0 is the pixel centre (R2 sequence)
__device__ __forceinline__ void spp_offset(int k, float& ox, float& oy) {
  float fk = (float)k;
  float a = fmaf(fk, 0.7548776662466927f, 0.5f);
  float b = fmaf(fk, 0.5698402909980532f, 0.5f);
  ox = a - floorf(a);
  oy = b - floorf(b);
}

// pinhole ray through pixel (px+ox, py+oy); direction normalised
// OpenCV radial + tangential model and its Jacobian; the operation order is the oracle's (lens_eval)
__device__ __forceinline__ void lens_eval(const float L[4], float x, float y, float& fx, float& fy, float J[4]) {
  const float k1 = L[0], k2 = L[1], p1 = L[2], p2 = L[3];
  const float r2 = fmaf(x, x, y * y);
  const float kr = fmaf(k2, r2, k1);
  const float radial = fmaf(kr, r2, 1.0f);
  const float dk = 2.0f * fmaf(2.0f * k2, r2, k1);
  const float xy = x * y;
  fx = fmaf(x, radial, fmaf(2.0f * p1, xy, p2 * fmaf(2.0f * x, x, r2)));
  fy = fmaf(y, radial, fmaf(p1, fmaf(2.0f * y, y, r2), (2.0f * p2) * xy));
  const float a = 2.0f * fmaf(p1, x, p2 * y);
  J[0] = fmaf(x * x, dk, fmaf(2.0f * p1, y, fmaf(6.0f * p2, x, radial)));
  J[1] = fmaf(xy, dk, a);
  J[2] = J[1];
  J[3] = fmaf(y * y, dk, fmaf(6.0f * p1, y, fmaf(2.0f * p2, x, radial)));
}

constexpr int kLensIters = 8; // fixed count, no early exit: identical on every lane and in the oracle
__device__ __forceinline__ void lens_undistort(const float L[4], float& px, float& py) {
  const float xd = px, yd = py;
  float x = xd, y = yd;
  for (int it = 0; it < kLensIters; it++) {
    float fx, fy, J[4];
    lens_eval(L, x, y, fx, fy, J);
    const float ex = fx - xd, ey = fy - yd;
    const float det = fmaf(J[0], J[3], -(J[1] * J[2]));
    const float sx = fmaf(J[3], ex, -(J[1] * ey)) / det;
    const float sy = fmaf(J[0], ey, -(J[2] * ex)) / det;
    x -= sx;
    y -= sy;
  }
  px = x;
  py = y;
}

__device__ __forceinline__ bool has_lens(const CamDev& cam) {
  return cam.lens[0] != 0.0f || cam.lens[1] != 0.0f || cam.lens[2] != 0.0f || cam.lens[3] != 0.0f;
}

__device__ __forceinline__ void raygen(const CamDev& cam, int px, int py, float ox, float oy,
                                       float o[3], float d[3]) {
  float dx = (((float)px + ox) - cam.cx) / cam.fx;
  float dy = (((float)py + oy) - cam.cy) / cam.fy;
  if (has_lens(cam)) lens_undistort(cam.lens, dx, dy); // uniform per view
  float v[3];
#pragma unroll
  for (int r = 0; r < 3; r++) {
    const float* m = cam.c2w + r * 4;
    v[r] = fmaf(m[0], dx, fmaf(m[1], dy, m[2]));
    o[r] = m[3];
  }
  float n2 = fmaf(v[0], v[0], fmaf(v[1], v[1], v[2] * v[2]));
  float inv = 1.0f / sqrtf(n2);
#pragma unroll
  for (int r = 0; r < 3; r++) d[r] = v[r] * inv;
}

// slab test against the unit cube; returns hit
__device__ __forceinline__ bool ray_aabb(const float o[3], const float d[3], float& t0, float& t1) {
  float tmin = 0.0f, tmax = __builtin_inff();
#pragma unroll
  for (int a = 0; a < 3; a++) {
    float inv = 1.0f / d[a];
    float ta = (0.0f - o[a]) * inv;
    float tb = (1.0f - o[a]) * inv;
    tmin = fmaxf(tmin, fminf(ta, tb));
    tmax = fminf(tmax, fmaxf(ta, tb));
  }
  t0 = tmin;
  t1 = tmax;
  return tmax > tmin;
}

__device__ __forceinline__ bool occupied(const FieldDev& f, float px, float py, float pz) {
  int R = f.occ_res;
  float fR = (float)R;
  int cx = min((int)(clamp01(px) * fR), R - 1);
  int cy = min((int)(clamp01(py) * fR), R - 1);
  int cz = min((int)(clamp01(pz) * fR), R - 1);
  // R <= 1024: every factor is below 2^24, so the full-rate 24-bit multiply gives the same integers as v_mul_lo_u32
  uint32_t bit = (uint32_t)cx + __umul24((uint32_t)R, (uint32_t)cy + __umul24((uint32_t)R, (uint32_t)cz));
  return (f.occ[bit >> 5] >> (bit & 31)) & 1u;
}

// dilated coarse occupancy: 0 means no occupied fine cell within one coarse cell of p
__device__ __forceinline__ bool occupied_coarse(const FieldDev& f, float px, float py, float pz) {
  const int R = f.occ_res >> 2;
  const float fR = (float)R;
  const int cx = min((int)(clamp01(px) * fR), R - 1);
  const int cy = min((int)(clamp01(py) * fR), R - 1);
  const int cz = min((int)(clamp01(pz) * fR), R - 1);
  const uint32_t bit = (uint32_t)cx + __umul24((uint32_t)R, (uint32_t)cy + __umul24((uint32_t)R, (uint32_t)cz));
  return (f.occ_coarse[bit >> 5] >> (bit & 31)) & 1u;
}

// librealsense-style camera of the reference (Share_Data.hpp:79-196): intr = {ppx, ppy, fx, fy, c0..c4},
// coefficients in the YAML order k1,k2,k3,p1,p2 (Share_Data.hpp:395-399); model 2 = inverse Brown-Conrady
struct Rs2Intr {
  float ppx, ppy, fx, fy, c[5];
  int width, height, model;
};
__device__ __forceinline__ void rs2_project(float pixel[2], const Rs2Intr& in, const float point[3]) {
  float x = point[0] / point[2], y = point[1] / point[2];
  if (in.model == 1 || in.model == 2) { // Share_Data.hpp:96-108
    const float r2 = x * x + y * y;
    const float f = 1 + in.c[0] * r2 + in.c[1] * r2 * r2 + in.c[4] * r2 * r2 * r2;
    x *= f;
    y *= f;
    const float dx = x + 2 * in.c[2] * x * y + in.c[3] * (r2 + 2 * x * x);
    const float dy = y + 2 * in.c[3] * x * y + in.c[2] * (r2 + 2 * y * y);
    x = dx;
    y = dy;
  }
  pixel[0] = x * in.fx + in.ppx;
  pixel[1] = y * in.fy + in.ppy;
}
__device__ __forceinline__ void rs2_deproject(float point[3], const Rs2Intr& in, const float pixel[2], float depth) {
  float x = (pixel[0] - in.ppx) / in.fx;
  float y = (pixel[1] - in.ppy) / in.fy;
  if (in.model == 2) { // Share_Data.hpp:147-155
    const float r2 = x * x + y * y;
    const float f = 1 + in.c[0] * r2 + in.c[1] * r2 * r2 + in.c[4] * r2 * r2 * r2;
    const float ux = x * f + 2 * in.c[2] * x * y + in.c[3] * (r2 + 2 * x * x);
    const float uy = y * f + 2 * in.c[3] * x * y + in.c[2] * (r2 + 2 * y * y);
    x = ux;
    y = uy;
  }
  point[0] = depth * x;
  point[1] = depth * y;
  point[2] = depth;
}

// float -> half (round to nearest even) of a value that HAS been rounded to float.  Without the barrier clang folds
// `(half)(a * b)`, `(half)(a - b)` or `(half)fmaf(..)` into v_fma_mix{lo,hi}_f16, which rounds the exact result ONCE,
// to half; the arithmetic contract with the oracle is "round to float, then to half", and the two differ on one
// value in 2^13 (found by tests/test_gpu_sweep.py on tables with a non-power-of-two amplitude).
__device__ __forceinline__ _Float16 to_half(float x) {
  asm("" : "+v"(x));
  return (_Float16)x;
}

// real spherical harmonics degree 4; op order identical to the oracle
__device__ __forceinline__ void sh4(float x, float y, float z, float o[16]) {
  float xy = x * y, xz = x * z, yz = y * z, x2 = x * x, y2 = y * y, z2 = z * z;
  o[0] = 0.28209479177387814f;
  o[1] = -0.48860251190291987f * y;
  o[2] = 0.48860251190291987f * z;
  o[3] = -0.48860251190291987f * x;
  o[4] = 1.0925484305920792f * xy;
  o[5] = -1.0925484305920792f * yz;
  o[6] = 0.94617469575755997f * z2 - 0.31539156525251999f;
  o[7] = -1.0925484305920792f * xz;
  o[8] = 0.54627421529603959f * x2 - 0.54627421529603959f * y2;
  o[9] = (0.59004358992664352f * y) * (-3.0f * x2 + y2);
  o[10] = (2.8906114426405538f * xy) * z;
  o[11] = (0.45704579946446572f * y) * (1.0f - 5.0f * z2);
  o[12] = (0.3731763325901154f * z) * (5.0f * z2 - 3.0f);
  o[13] = (0.45704579946446572f * x) * (1.0f - 5.0f * z2);
  o[14] = (1.4453057213202769f * z) * (x2 - y2);
  o[15] = (0.59004358992664352f * x) * (-x2 + 3.0f * y2);
}

// ---------------------------------------------------------------- hash-grid gather
// one table entry = F halfs: an 8-byte (F = 4) or 4-byte (F = 2) load
template <int F> struct EntryWord;
template <> struct EntryWord<4> { typedef uint2 type; };
template <> struct EntryWord<2> { typedef uint32_t type; };
template <int F> struct Entry;
template <> struct Entry<4> {
  uint32_t w[2];
  __device__ __forceinline__ static Entry load(const char* p) {
    const uint2 v = *reinterpret_cast<const uint2*>(p);
    Entry e;
    e.w[0] = v.x;
    e.w[1] = v.y;
    return e;
  }
};
template <> struct Entry<2> {
  uint32_t w[1];
  __device__ __forceinline__ static Entry load(const char* p) {
    Entry e;
    e.w[0] = *reinterpret_cast<const uint32_t*>(p);
    return e;
  }
};

// the entries of vertices x and x+1 with ONE load (dense levels: contiguous in the physical layout, and
// the entry after the last vertex of a row duplicates it, so the +1 clamp is implicit)
template <int F> struct EntryPair;
template <> struct EntryPair<4> {
  uint32_t w[4];
  __device__ __forceinline__ static EntryPair load(const char* p) {
    const uint4 v = *reinterpret_cast<const uint4*>(p); // 8-byte aligned 16-byte load
    EntryPair e;
    e.w[0] = v.x; e.w[1] = v.y; e.w[2] = v.z; e.w[3] = v.w;
    return e;
  }
};
template <> struct EntryPair<2> {
  uint32_t w[2];
  __device__ __forceinline__ static EntryPair load(const char* p) {
    const uint2 v = *reinterpret_cast<const uint2*>(p);
    EntryPair e;
    e.w[0] = v.x; e.w[1] = v.y;
    return e;
  }
};

// ---------------------------------------------------------------- gather: one lane = one sample, every level
// One level, one sample, one lane: 8 corner values of F halfs and the trilinear blend in packed binary16 (as
// tiny-cuda-nn does for fp16 tables): weights rounded to fp16, w = fp16(fp16(wx*wy)*wz), acc = fp16 fma(w, v, acc),
// corners in order dx + 2dy + 4dz.  Every op is an IEEE RNE fp16 op (v_pk_mul_f16 / v_pk_fma_f16), so the result is
// bit-identical to the oracle.  out = F/2 packed pairs.
//  * DENSE levels (compile time: the leading levels of the field that are physically dense): the physical layout
//    duplicates the last vertex of every row, the last row of every plane and the last plane, so the +1 neighbours
//    need no clamp on any axis and are one add away: base, base + my, base + mz, base + my + mz (4 paired loads);
//  * the (1 - w, w) weight pairs of all three axes come out of one v_cvt_pk_f16_f32 each.
__device__ __forceinline__ half2v cvt_pk_f16(float lo, float hi) { // {RNE(lo), RNE(hi)}: two independent roundings, no fusing
  half2v r;
  asm("v_cvt_pk_f16_f32 %0, %1, %2" : "=v"(r) : "v"(lo), "v"(hi));
  return r;
}

struct HashConsts { // the field's shared hash constants, wave-uniform (FieldDev::hash_*)
  uint32_t my_b, mz_b, m_b;
  uint32_t wide; // FieldDev::wide_offsets (the generic gather's 32-bit multiplies)
};
enum { kLevelDense = 0, kLevelHashedShared = 1, kLevelGeneric = 2 };

// The last cell a LANE gathered on one hashed level and its eight corner entries.  Under the engine's stepping rule
// (dt = sqrt(3)/1024 = 0.43 of the finest cell of a 256^3 field) a ray's consecutive samples stay in the cell about
// half of the time; where a cohort's gathers are incoherent to begin with (the reference's 80x45 candidates: neighbouring
// pixels are six finest cells apart, every corner its own cache line) the launch is bound by the L2's request rate
// (profiles/NOTES.md, round 4), and the loads of a lane whose cell did not change are pure waste.  The cache maps a CELL to
// its entries, so it stays valid when the lane takes over another ray; values and blend are untouched: same pixels.
template <int F>
struct CornerCache {
  uint32_t key; // x | y << 10 | z << 20 of the cell (levels up to 1023 cells per axis); ~0u = nothing cached
  uint32_t v[8][F / 2];
};

template <int F, int KIND>
__device__ __forceinline__ void encode_level2(const uint16_t* __restrict__ table, const LevelDev& L, const HashConsts& H, float px,
                                              float py, float pz, half2v out[F / 2]) {
  constexpr bool DENSE = KIND == kLevelDense;
  constexpr int ESH = F == 4 ? 3 : 2;
  const float pos[3] = {fmaf(L.scale, px, 0.5f), fmaf(L.scale, py, 0.5f), fmaf(L.scale, pz, 0.5f)};
  uint32_t c0[3];
  half2v wa[3]; // (1-w, w) per axis, fp16
#pragma unroll
  for (int a = 0; a < 3; a++) {
    const float w1 = __builtin_amdgcn_fractf(pos[a]);
    wa[a] = cvt_pk_f16(1.0f - w1, w1);
    c0[a] = (uint32_t)(int)pos[a];
  }
  uint32_t vw[8][F / 2];
  // The level's packed word, made opaque once per use: everything derived from it (strides, the scalar base pointers) is
  // then recomputed by the scalar ALU every round -- a handful of SALU instructions per level -- instead of being hoisted
  // out of the loop into registers the kernel does not have (it was spilling 40..117 SGPRs, one v_readlane per use).
  uint32_t pk = L.pack;
  if (KIND != kLevelGeneric) asm volatile("" : "+s"(pk));
  if (DENSE) {
    // the level constants are wave-uniform here (scalar registers): the level offset and the +y neighbour go into two
    // SCALAR base pointers (SALU adds), so the four paired loads need one vector add between them -- (b, base),
    // (b, base + my), (b + mz, base), (b + mz, base + my)
    const uint32_t sxl = pk & 31u, off = pk & ~31u;
    const uint32_t my = (uint32_t)(F * 2) << sxl, mz = (uint32_t)(F * 2) << (2u * sxl);
    const char* base0 = reinterpret_cast<const char*>(table) + off;
    const char* base1 = base0 + my;
    const uint32_t b = (c0[0] << ESH) + __umul24(c0[1], my) + __umul24(c0[2], mz);
    const uint32_t bz = b + mz;
#pragma unroll
    for (int q = 0; q < 4; q++) {
      const EntryPair<F> e = EntryPair<F>::load(((q & 1) ? base1 : base0) + ((q >> 1) ? bz : b));
#pragma unroll
      for (int k = 0; k < F / 2; k++) {
        vw[2 * q][k] = e.w[k];
        vw[2 * q + 1][k] = e.w[F / 2 + k];
      }
    }
  } else if (KIND == kLevelHashedShared) {
    // hashed level, the field's shared constants: byte offset inside the level = ((x << esh) ^ (y * my) ^ (z * mz)) & m.
    // The x term needs no mask (res <= T, host-checked), the level's offset goes into a SCALAR base pointer, and the three
    // terms meet in one v_bitop3_b32 (xor3) per corner: per level 3 add + 3 min + 2 shift + 4 mul24 + 4 and + 8 xor3.
    const uint32_t res_m1 = pk & 4095u;
    uint32_t c1[3];
#pragma unroll
    for (int a = 0; a < 3; a++) c1[a] = min(c0[a] + 1u, res_m1);
    const char* base = reinterpret_cast<const char*>(table) + (pk & ~4095u);
    const uint32_t tx[2] = {c0[0] << ESH, c1[0] << ESH};
    const uint32_t ty[2] = {__umul24(c0[1], H.my_b) & H.m_b, __umul24(c1[1], H.my_b) & H.m_b};
    const uint32_t tz[2] = {__umul24(c0[2], H.mz_b) & H.m_b, __umul24(c1[2], H.mz_b) & H.m_b};
    {
#pragma unroll
      for (int c = 0; c < 8; c++) {
        const uint32_t byte_off = __builtin_amdgcn_bitop3_b32(tx[c & 1], ty[(c >> 1) & 1], tz[c >> 2], 0x96); // a ^ b ^ c
        const Entry<F> e = Entry<F>::load(base + byte_off);
#pragma unroll
        for (int k = 0; k < F / 2; k++) vw[c][k] = e.w[k];
      }
    }
  } else {
    uint32_t c1[3];
#pragma unroll
    for (int a = 0; a < 3; a++) c1[a] = min(c0[a] + 1u, L.res_m1);
    // (levels beyond 16 MiB -- FieldDev::wide_offsets, wave-uniform -- need more than the 24 low bits of the products)
    auto mul = [&](uint32_t a, uint32_t b) { return H.wide ? a * b : __umul24(a, b); };
    const uint32_t tx[2] = {(c0[0] << ESH) & L.m_b, (c1[0] << ESH) & L.m_b};
    const uint32_t ty[2] = {mul(c0[1], L.my_b) & L.m_b, mul(c1[1], L.my_b) & L.m_b};
    const uint32_t tz[2] = {(mul(c0[2], L.mz_b) & L.m_b) | L.off_b, (mul(c1[2], L.mz_b) & L.m_b) | L.off_b};
#pragma unroll
    for (int c = 0; c < 8; c++) {
      const uint32_t byte_off = tx[c & 1] ^ ty[(c >> 1) & 1] ^ tz[c >> 2];
      const Entry<F> e = Entry<F>::load(reinterpret_cast<const char*>(table) + byte_off);
#pragma unroll
      for (int k = 0; k < F / 2; k++) vw[c][k] = e.w[k];
    }
  }
  half2v wp[4];
#pragma unroll
  for (int q = 0; q < 4; q++) {
    const half2v wy = {wa[1][q & 1], wa[1][q & 1]}, wz = {wa[2][q >> 1], wa[2][q >> 1]};
    wp[q] = (wa[0] * wy) * wz;
  }
#pragma unroll
  for (int k = 0; k < F / 2; k++) out[k] = half2v{(_Float16)0.0f, (_Float16)0.0f};
#pragma unroll
  for (int c = 0; c < 8; c++) {
    const _Float16 w = wp[c >> 1][c & 1];
    const half2v ww = {w, w};
#pragma unroll
    for (int k = 0; k < F / 2; k++) out[k] = __builtin_elementwise_fma(ww, __builtin_bit_cast(half2v, vw[c][k]), out[k]);
  }
}

// ---- the gather in two explicit phases (CornerCache instances).  The plain path above leaves it to the scheduler to hoist
// all 44 loads of a sample to the top of one basic block; the cache's per-level branches cut that block into pieces and the
// scheduler then loads level after level, one memory round trip each.  Here phase 1 issues EVERY load of the sample -- the
// dense levels' four paired loads into a staging record, the hashed levels' eight loads (only in the lanes whose cell
// changed) into the cache -- and phase 2 blends; the arithmetic is encode_level2's, operation for operation.
template <int F>
struct LevelStage {
  uint32_t vw[8][F / 2]; // dense levels: the corner entries as loaded
  half2v wa[3];          // (1 - w, w) per axis, fp16
};

// (1 - w, w) per axis and the cell of one level
template <int F>
__device__ __forceinline__ void level_cell(const LevelDev& L, float px, float py, float pz, uint32_t c0[3], half2v wa[3]) {
  const float pos[3] = {fmaf(L.scale, px, 0.5f), fmaf(L.scale, py, 0.5f), fmaf(L.scale, pz, 0.5f)};
#pragma unroll
  for (int a = 0; a < 3; a++) {
    const float w1 = __builtin_amdgcn_fractf(pos[a]);
    wa[a] = cvt_pk_f16(1.0f - w1, w1);
    c0[a] = (uint32_t)(int)pos[a];
  }
}

template <int F>
__device__ __forceinline__ void stage_dense_level(const uint16_t* __restrict__ table, const LevelDev& L, float px, float py, float pz,
                                                  LevelStage<F>& st) {
  constexpr int ESH = F == 4 ? 3 : 2;
  uint32_t c0[3];
  level_cell<F>(L, px, py, pz, c0, st.wa);
  uint32_t pk = L.pack;
  asm volatile("" : "+s"(pk));
  const uint32_t sxl = pk & 31u, off = pk & ~31u;
  const uint32_t my = (uint32_t)(F * 2) << sxl, mz = (uint32_t)(F * 2) << (2u * sxl);
  const char* base0 = reinterpret_cast<const char*>(table) + off;
  const char* base1 = base0 + my;
  const uint32_t b = (c0[0] << ESH) + __umul24(c0[1], my) + __umul24(c0[2], mz);
  const uint32_t bz = b + mz;
#pragma unroll
  for (int q = 0; q < 4; q++) {
    const EntryPair<F> e = EntryPair<F>::load(((q & 1) ? base1 : base0) + ((q >> 1) ? bz : b));
#pragma unroll
    for (int k = 0; k < F / 2; k++) {
      st.vw[2 * q][k] = e.w[k];
      st.vw[2 * q + 1][k] = e.w[F / 2 + k];
    }
  }
}

// hashed level, shared hash constants: the lanes whose cell changed since their last sample issue the level's eight loads
// (the branch narrows EXEC; a wave none of whose lanes moved skips them); nobody waits here
template <int F>
__device__ __forceinline__ void stage_hashed_level(const uint16_t* __restrict__ table, const LevelDev& L, const HashConsts& H, float px,
                                                   float py, float pz, CornerCache<F>& cc, half2v wa[3]) {
  constexpr int ESH = F == 4 ? 3 : 2;
  uint32_t c0[3], c1[3];
  level_cell<F>(L, px, py, pz, c0, wa);
  uint32_t pk = L.pack;
  asm volatile("" : "+s"(pk));
  const uint32_t res_m1 = pk & 4095u;
  const uint32_t key = c0[0] | (c0[1] << 10) | (c0[2] << 20);
  if (key != cc.key) {
    cc.key = key;
#pragma unroll
    for (int a = 0; a < 3; a++) c1[a] = min(c0[a] + 1u, res_m1);
    const char* base = reinterpret_cast<const char*>(table) + (pk & ~4095u);
    const uint32_t tx[2] = {c0[0] << ESH, c1[0] << ESH};
    const uint32_t ty[2] = {__umul24(c0[1], H.my_b) & H.m_b, __umul24(c1[1], H.my_b) & H.m_b};
    const uint32_t tz[2] = {__umul24(c0[2], H.mz_b) & H.m_b, __umul24(c1[2], H.mz_b) & H.m_b};
#pragma unroll
    for (int c = 0; c < 8; c++) {
      const uint32_t byte_off = __builtin_amdgcn_bitop3_b32(tx[c & 1], ty[(c >> 1) & 1], tz[c >> 2], 0x96);
      const Entry<F> e = Entry<F>::load(base + byte_off);
#pragma unroll
      for (int k = 0; k < F / 2; k++) cc.v[c][k] = e.w[k];
    }
  }
}

// the trilinear blend of encode_level2, from eight corner entries and the axis weights
template <int F>
__device__ __forceinline__ void blend_level(const uint32_t vw[8][F / 2], const half2v wa[3], half2v out[F / 2]) {
  half2v wp[4];
#pragma unroll
  for (int q = 0; q < 4; q++) {
    const half2v wy = {wa[1][q & 1], wa[1][q & 1]}, wz = {wa[2][q >> 1], wa[2][q >> 1]};
    wp[q] = (wa[0] * wy) * wz;
  }
#pragma unroll
  for (int k = 0; k < F / 2; k++) out[k] = half2v{(_Float16)0.0f, (_Float16)0.0f};
#pragma unroll
  for (int c = 0; c < 8; c++) {
    const _Float16 w = wp[c >> 1][c & 1];
    const half2v ww = {w, w};
#pragma unroll
    for (int k = 0; k < F / 2; k++) out[k] = __builtin_elementwise_fma(ww, __builtin_bit_cast(half2v, vw[c][k]), out[k]);
  }
}

template <int F, int NDENSE, int... J>
__device__ __forceinline__ void encode_all_levels_cached(const uint16_t* __restrict__ table, const LevelDev* __restrict__ lv,
                                                         const HashConsts& H, float px, float py, float pz, half2v* out,
                                                         CornerCache<F>* cc, std::integer_sequence<int, J...>) {
  constexpr int NH = 32 / F - NDENSE;
  LevelStage<F> ds[NDENSE];
  half2v hwa[NH][3];
  // phase 1: every load of the sample (hashed levels first: theirs are the long ones)
  ((J >= NDENSE ? stage_hashed_level<F>(table, lv[J], H, px, py, pz, cc[J >= NDENSE ? J - NDENSE : 0], hwa[J >= NDENSE ? J - NDENSE : 0]) : (void)0), ...);
  ((J < NDENSE ? stage_dense_level<F>(table, lv[J], px, py, pz, ds[J < NDENSE ? J : 0]) : (void)0), ...);
  __builtin_amdgcn_sched_barrier(0);
  // phase 2: the blends
  ((J < NDENSE ? blend_level<F>(ds[J < NDENSE ? J : 0].vw, ds[J < NDENSE ? J : 0].wa, out + J * (F / 2))
               : blend_level<F>(cc[J >= NDENSE ? J - NDENSE : 0].v, hwa[J >= NDENSE ? J - NDENSE : 0], out + J * (F / 2))), ...);
}

// NDENSE > 0: the field has EXACTLY NDENSE leading dense levels and its hashed levels share their constants
// (FieldDev::hash_shared; the host picks the instance).  NDENSE = 0: every level through the generic path.
template <int F, int NDENSE, int... J>
__device__ __forceinline__ void encode_all_levels(const uint16_t* __restrict__ table, const LevelDev* __restrict__ lv, const HashConsts& H,
                                                  float px, float py, float pz, half2v* out, std::integer_sequence<int, J...>) {
  (encode_level2<F, (J < NDENSE ? kLevelDense : NDENSE > 0 ? kLevelHashedShared : kLevelGeneric)>(table, lv[J], H, px, py, pz, out + J * (F / 2)), ...);
}

// all 32 features of one sample in canonical order (feature F*l + f), as four half8 = the k rows [8s, 8s+8) of the first
// layer; render_queue64 turns them into MFMA B fragments with v_permlane32_swap
template <int F, int NDENSE, bool CACHE = false>
__device__ __forceinline__ void encode_sample(const uint16_t* __restrict__ table, const LevelDev* __restrict__ lv, const HashConsts& H,
                                              float px, float py, float pz, half8 f[4], CornerCache<F>* cc = nullptr) {
  px = clamp01(px);
  py = clamp01(py);
  pz = clamp01(pz);
  half2v out[16];
  if constexpr (CACHE && NDENSE > 0) encode_all_levels_cached<F, NDENSE>(table, lv, H, px, py, pz, out, cc, std::make_integer_sequence<int, 32 / F>{});
  else encode_all_levels<F, NDENSE>(table, lv, H, px, py, pz, out, std::make_integer_sequence<int, 32 / F>{});
#pragma unroll
  for (int s = 0; s < 4; s++)
#pragma unroll
    for (int k = 0; k < 4; k++) {
      f[s][2 * k] = out[4 * s + k][0];
      f[s][2 * k + 1] = out[4 * s + k][1];
    }
}

// lanes 32..63 of a <-> lanes 0..31 of b (gfx950 v_permlane32_swap), for every dword of a half8
__device__ __forceinline__ void swap_halves(half8& a, half8& b) {
  typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
  u32x4 x = __builtin_bit_cast(u32x4, a), y = __builtin_bit_cast(u32x4, b);
#pragma unroll
  for (int k = 0; k < 4; k++) {
    const auto r = __builtin_amdgcn_permlane32_swap(x[k], y[k], false, false); // {x.lo | y.lo, x.hi | y.hi}
    x[k] = r[0];
    y[k] = r[1];
  }
  a = __builtin_bit_cast(half8, x);
  b = __builtin_bit_cast(half8, y);
}

// stage the level table into LDS (call with all threads of the block, then __syncthreads)
__device__ __forceinline__ void stage_levels(const FieldDev& fd, LevelDev* lds_levels) {
  static_assert(sizeof(LevelDev) == 32, "LevelDev must be two 16-byte words");
  if (threadIdx.x < (unsigned)fd.n_levels) lds_levels[threadIdx.x] = fd.levels[threadIdx.x];
}

// ---------------------------------------------------------------- tiny MLPs on MFMA
// Samples sit on the MFMA column (lane & 31); hidden units on the accumulator rows, so
// every layer's output is already the next layer's B operand (no LDS round trip).
// 32x32x16 f16 MFMA: lane (r = lane&31, h = lane>>5) holds A[row r][k = 8h+j],
// B[k = 8h+j][col r]; D reg i = row (i&3) + 8(i>>2) + 4h, col r.

__device__ __forceinline__ f32x16 mfma(half8 a, half8 b, f32x16 c) {
  return __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0);
}

// accumulator rows [base, base+8) -> one fp16 B fragment.  ReLU is applied AFTER the fp16
// rounding (relu(round(x)) == round(relu(x))): the conversion result is canonical, so the
// compiler emits one v_pk_max_f16 per pair instead of two v_max_f32 per value.
template <bool RELU>
__device__ __forceinline__ half8 pack8(const f32x16& acc, int base) {
  half8 r;
#pragma unroll
  for (int j = 0; j < 8; j += 2) {
    half2v p;
    p[0] = (_Float16)acc[base + j];
    p[1] = (_Float16)acc[base + j + 1];
    if (RELU) {
      const half2v z = {(_Float16)0.0f, (_Float16)0.0f};
      p = __builtin_elementwise_max(p, z);
    }
    r[j] = p[0];
    r[j + 1] = p[1];
  }
  return r;
}

struct MlpOut {
  f32x16 dens; // density MLP outputs: rows 0..15 live in regs 0..7 (4 rows per lane half)
  f32x16 rgb;  // colour MLP outputs: rows 0..2 = r,g,b logits in regs 0..2 of lane half 0
};

// A tap sees the fp16 B fragments a lane holds between the layers (the trainer's forward pass stores them for its
// backward pass): tap(0, h1[4]), tap(1, &density_out), tap(2, h2[4]), tap(3, h3[4]).  Fragment t, element j of lane
// half h = unit 32 (t >> 1) + 16 (t & 1) + 8 (j >> 2) + (j & 3) + 4 h of a 64-unit layer; the density output's element
// j = unit (j & 3) + 8 (j >> 2) + 4 h.
struct MlpNoTap {
  __device__ __forceinline__ void operator()(int, const half8*) const {}
};

// frag order: D1 (mt,s)=4 | D2 s=4 | R1 (mt,s)=4 | R2 (mt,s)=8 | R3 s=4
template <class Tap = MlpNoTap>
__device__ __forceinline__ MlpOut mlp_forward(const half8* __restrict__ wl, int lane, half8 f0,
                                              half8 f1, half8 shfrag, Tap tap = Tap()) {
  const f32x16 zero = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
  MlpOut out;
  half8 hf[4];
  { // density layer 1: 32 -> 64
    f32x16 a0 = mfma(wl[0 * 64 + lane], f0, zero);
    f32x16 a1 = mfma(wl[2 * 64 + lane], f0, zero);
    a0 = mfma(wl[1 * 64 + lane], f1, a0);
    a1 = mfma(wl[3 * 64 + lane], f1, a1);
    hf[0] = pack8<true>(a0, 0);
    hf[1] = pack8<true>(a0, 8);
    hf[2] = pack8<true>(a1, 0);
    hf[3] = pack8<true>(a1, 8);
    tap(0, hf);
  }
  { // density layer 2: 64 -> 16 (rows 16..31 zero padding)
    f32x16 a = mfma(wl[4 * 64 + lane], hf[0], zero);
    a = mfma(wl[5 * 64 + lane], hf[1], a);
    a = mfma(wl[6 * 64 + lane], hf[2], a);
    a = mfma(wl[7 * 64 + lane], hf[3], a);
    out.dens = a;
  }
  half8 df = pack8<false>(out.dens, 0);
  tap(1, &df);
  { // colour layer 1: [density out 16 | SH 16] -> 64
    f32x16 a0 = mfma(wl[8 * 64 + lane], df, zero);
    f32x16 a1 = mfma(wl[10 * 64 + lane], df, zero);
    a0 = mfma(wl[9 * 64 + lane], shfrag, a0);
    a1 = mfma(wl[11 * 64 + lane], shfrag, a1);
    hf[0] = pack8<true>(a0, 0);
    hf[1] = pack8<true>(a0, 8);
    hf[2] = pack8<true>(a1, 0);
    hf[3] = pack8<true>(a1, 8);
    tap(2, hf);
  }
  { // colour layer 2: 64 -> 64
    f32x16 a0 = mfma(wl[12 * 64 + lane], hf[0], zero);
    f32x16 a1 = mfma(wl[16 * 64 + lane], hf[0], zero);
    a0 = mfma(wl[13 * 64 + lane], hf[1], a0);
    a1 = mfma(wl[17 * 64 + lane], hf[1], a1);
    a0 = mfma(wl[14 * 64 + lane], hf[2], a0);
    a1 = mfma(wl[18 * 64 + lane], hf[2], a1);
    a0 = mfma(wl[15 * 64 + lane], hf[3], a0);
    a1 = mfma(wl[19 * 64 + lane], hf[3], a1);
    hf[0] = pack8<true>(a0, 0);
    hf[1] = pack8<true>(a0, 8);
    hf[2] = pack8<true>(a1, 0);
    hf[3] = pack8<true>(a1, 8);
    tap(3, hf);
  }
  { // colour layer 3: 64 -> 16 (3 used)
    f32x16 a = mfma(wl[20 * 64 + lane], hf[0], zero);
    a = mfma(wl[21 * 64 + lane], hf[1], a);
    a = mfma(wl[22 * 64 + lane], hf[2], a);
    a = mfma(wl[23 * 64 + lane], hf[3], a);
    out.rgb = a;
  }
  return out;
}

// Both MLPs for TWO column groups of 32 samples (render_queue64: the wave's lanes 0..31 and 32..63 each own a sample):
// every weight fragment is read from LDS once and used for both groups; the groups' chains are independent, so one
// group's MFMAs run while the other group's accumulators are being packed.  Fragments: the `frags64` set (prv_api.cpp:
// first-layer k rows in canonical feature order; the density layer's unit 0 and the colour layer's units 0..2 repeated in
// the padding rows 20 / 20..22, which land in lane half 1: register 8 / registers 8..10).
struct MlpOut2 {
  f32x16 densA, rgbA, densB, rgbB;
};

__device__ __forceinline__ MlpOut2 mlp_forward2(const half8* __restrict__ wl, int lane, const half8 fA[2], const half8 fB[2], half8 shA,
                                                half8 shB) {
  const f32x16 zero = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
  MlpOut2 out;
  half8 hA[4], hB[4];
  // per layer: group A's MFMAs, group B's MFMAs, then the packs -- B's MFMAs are in the pipe while A's accumulators are
  // converted; at most two 32x32 accumulators per group are live
  auto layer64 = [&](int f0, const half8 in0A, const half8 in1A, const half8 in0B, const half8 in1B) { // K = 32 -> 64 units
    const half8 w0 = wl[(f0 + 0) * 64 + lane], w1 = wl[(f0 + 1) * 64 + lane], w2 = wl[(f0 + 2) * 64 + lane], w3 = wl[(f0 + 3) * 64 + lane];
    f32x16 a0 = mfma(w0, in0A, zero), a1 = mfma(w2, in0A, zero);
    a0 = mfma(w1, in1A, a0);
    a1 = mfma(w3, in1A, a1);
    f32x16 b0 = mfma(w0, in0B, zero), b1 = mfma(w2, in0B, zero);
    b0 = mfma(w1, in1B, b0);
    b1 = mfma(w3, in1B, b1);
    hA[0] = pack8<true>(a0, 0); hA[1] = pack8<true>(a0, 8); hA[2] = pack8<true>(a1, 0); hA[3] = pack8<true>(a1, 8);
    hB[0] = pack8<true>(b0, 0); hB[1] = pack8<true>(b0, 8); hB[2] = pack8<true>(b1, 0); hB[3] = pack8<true>(b1, 8);
  };
  auto layer16 = [&](int f0, f32x16& a, f32x16& b) { // K = 64 -> 16 units (+ copies in the padding rows)
    a = zero;
    b = zero;
#pragma unroll
    for (int s = 0; s < 4; s++) {
      const half8 w = wl[(f0 + s) * 64 + lane];
      a = mfma(w, hA[s], a);
      b = mfma(w, hB[s], b);
    }
  };
  layer64(0, fA[0], fA[1], fB[0], fB[1]);  // density layer 1: 32 -> 64
  layer16(4, out.densA, out.densB);         // density layer 2: 64 -> 16
  const half8 dfA = pack8<false>(out.densA, 0), dfB = pack8<false>(out.densB, 0);
  layer64(8, dfA, shA, dfB, shB);           // colour layer 1: [density out 16 | SH 16] -> 64
  { // colour layer 2: 64 -> 64
    f32x16 a0 = zero, a1 = zero, b0 = zero, b1 = zero;
#pragma unroll
    for (int s = 0; s < 4; s++) {
      const half8 w0 = wl[(12 + s) * 64 + lane], w1 = wl[(16 + s) * 64 + lane];
      a0 = mfma(w0, hA[s], a0);
      a1 = mfma(w1, hA[s], a1);
      b0 = mfma(w0, hB[s], b0);
      b1 = mfma(w1, hB[s], b1);
    }
    hA[0] = pack8<true>(a0, 0); hA[1] = pack8<true>(a0, 8); hA[2] = pack8<true>(a1, 0); hA[3] = pack8<true>(a1, 8);
    hB[0] = pack8<true>(b0, 0); hB[1] = pack8<true>(b0, 8); hB[2] = pack8<true>(b1, 0); hB[3] = pack8<true>(b1, 8);
  }
  layer16(20, out.rgbA, out.rgbB);          // colour layer 3: 64 -> 16 (3 used)
  return out;
}

// SH fragment of lane half h: coefficients [8h, 8h+8) as fp16
__device__ __forceinline__ half8 sh_fragment(int h, float dx, float dy, float dz) {
  float s[16];
  sh4(dx, dy, dz, s);
  half8 r;
#pragma unroll
  for (int j = 0; j < 8; j++) r[j] = to_half(h ? s[8 + j] : s[j]);
  return r;
}

// exp via the hardware exp2 (v_exp_f32) and a hardware reciprocal: ~1 ulp each
__device__ __forceinline__ float fast_exp(float x) { return __builtin_amdgcn_exp2f(x * 1.4426950408889634f); }
__device__ __forceinline__ float fast_sigmoid(float x) { return __builtin_amdgcn_rcpf(1.0f + fast_exp(-x)); }

__device__ __forceinline__ float linear_to_srgb(float x) {
  if (x <= 0.0031308f) return 12.92f * x;
  return 1.055f * powf(x, 0.41666666f) - 0.055f;
}

// composite over bg, un-premultiply, sRGB, clip, *255+0.5 truncate (upstream write_image recipe)
__device__ __forceinline__ uint32_t quantize_rgba8(float r, float g, float b, float a,
                                                   const float bg[4]) {
  float rem = 1.0f - a;
  float c[4] = {fmaf(rem, bg[0], r), fmaf(rem, bg[1], g), fmaf(rem, bg[2], b), fmaf(rem, bg[3], a)};
  uint32_t outw = 0;
#pragma unroll
  for (int k = 0; k < 3; k++) {
    float v = c[3] != 0.0f ? c[k] / c[3] : c[k];
    v = linear_to_srgb(v);
    v = fminf(fmaxf(v, 0.0f), 1.0f);
    outw |= ((uint32_t)(v * 255.0f + 0.5f)) << (8 * k);
  }
  float v = fminf(fmaxf(c[3], 0.0f), 1.0f);
  outw |= ((uint32_t)(v * 255.0f + 0.5f)) << 24;
  return outw;
}

} // namespace prv
