/*
 * prv_oracle.c -- CPU ORACLE (test infrastructure; see prv_oracle.h header for
 * scope and the "parity unpinned" statement).  Plain C, scalar, written to be
 * obviously correct.  Floating-point op ORDER is part of the specification:
 * compile with -ffp-contract=off so only the explicit fmaf() calls fuse.
 *
 * Citations "path:line" are relative to the reference tree (psc0628/NeRF-PRV).
 */
#include "prv_oracle.h"

#include <immintrin.h>
#include <math.h>
#include <pthread.h>
#include <stdlib.h>
#include <string.h>

/* ------------------------------------------------------------------ fp16 */

uint16_t orc_f2h(float f) {
  uint32_t x;
  memcpy(&x, &f, 4);
  uint32_t sign = (x >> 16) & 0x8000u;
  uint32_t absx = x & 0x7fffffffu;
  if (absx >= 0x7f800000u) /* inf / nan */
    return (uint16_t)(sign | 0x7c00u | ((absx > 0x7f800000u) ? 0x200u | ((absx >> 13) & 0x3ffu) : 0));
  if (absx >= 0x477ff000u) /* >= 65520 rounds to inf */
    return (uint16_t)(sign | 0x7c00u);
  if (absx < 0x38800000u) { /* subnormal half or zero: value < 2^-14 */
    if (absx < 0x33000000u) return (uint16_t)sign; /* < 2^-25 -> 0 */
    uint32_t e = absx >> 23;                       /* biased exp, 102..112 */
    uint32_t m = (absx & 0x7fffffu) | 0x800000u;   /* 24-bit significand */
    uint32_t shift = 126u - e;                     /* 14..24: m * 2^(e-150) -> units of 2^-24 */
    uint32_t r = m >> shift;
    uint32_t rem = m & ((1u << shift) - 1u);
    uint32_t half = 1u << (shift - 1);
    if (rem > half || (rem == half && (r & 1u))) r++;
    return (uint16_t)(sign | r);
  }
  uint32_t e = (absx >> 23) - 112u; /* rebias 127 -> 15 */
  uint32_t m = absx & 0x7fffffu;
  uint32_t r = (e << 10) | (m >> 13);
  uint32_t rem = m & 0x1fffu;
  if (rem > 0x1000u || (rem == 0x1000u && (r & 1u))) r++; /* may carry into exponent: fine */
  return (uint16_t)(sign | r);
}

float orc_h2f(uint16_t h) {
  uint32_t sign = ((uint32_t)h & 0x8000u) << 16;
  uint32_t e = (h >> 10) & 0x1fu;
  uint32_t m = h & 0x3ffu;
  uint32_t x;
  if (e == 0) {
    if (m == 0) {
      x = sign;
    } else { /* subnormal: normalise */
      int s = 0;
      while (!(m & 0x400u)) { m <<= 1; s++; }
      m &= 0x3ffu;
      x = sign | ((uint32_t)(113 - s) << 23) | (m << 13);
    }
  } else if (e == 31) {
    x = sign | 0x7f800000u | (m << 13);
  } else {
    x = sign | ((e + 112u) << 23) | (m << 13);
  }
  float f;
  memcpy(&f, &x, 4);
  return f;
}

/* double -> binary16, round-to-nearest-even in ONE rounding (used to model fp16 fma/mul exactly:
 * the exact product/sum of binary16 operands fits a double) */
uint16_t orc_d2h_soft(double v);
uint16_t orc_d2h(double v) {
  /* fast path, bit-identical to the general routine below (tests/test_oracle_known_answers.py sweeps both): for a
   * result in the normal binary16 range, adding and subtracting 2^(E+42) (E = exponent of |v|) rounds |v| to 11
   * significant bits with the hardware's round-to-nearest-even -- the one rounding wanted -- and the outcome
   * converts to float and on to binary16 exactly (F16C) */
  double a = fabs(v);
  if (a >= 6.103515625e-05 && a < 65520.0) {
    uint64_t x;
    memcpy(&x, &a, 8);
    uint64_t cb = ((x >> 52) + 42u) << 52;
    double c;
    memcpy(&c, &cb, 8);
    volatile double t = a + c; /* volatile: the pair must not be folded away */
    double r = t - c;
    uint16_t h = (uint16_t)_cvtss_sh((float)r, _MM_FROUND_TO_NEAREST_INT | _MM_FROUND_NO_EXC);
    return (uint16_t)(h | (signbit(v) ? 0x8000u : 0u));
  }
  return orc_d2h_soft(v);
}
uint16_t orc_d2h_soft(double v) {
  uint64_t x;
  memcpy(&x, &v, 8);
  uint16_t sign = (uint16_t)((x >> 48) & 0x8000u);
  uint64_t absx = x & 0x7fffffffffffffffull;
  if (absx >= 0x7ff0000000000000ull) return (uint16_t)(sign | 0x7c00u | (absx > 0x7ff0000000000000ull ? 0x200u : 0));
  int e = (int)(absx >> 52) - 1023;          /* unbiased exponent */
  uint64_t m = (absx & 0xfffffffffffffull) | (absx >> 52 ? 0x10000000000000ull : 0); /* 53-bit significand */
  if (absx == 0) return sign;
  if (e >= 16) return (uint16_t)(sign | 0x7c00u); /* >= 65536 */
  int shift;                                  /* bits to drop from the 53-bit significand */
  int he;                                     /* half exponent field */
  if (e >= -14) { shift = 42; he = e + 15; }  /* normal half: keep 11 bits */
  else { shift = 42 + (-14 - e); he = 0; }    /* subnormal half: value = r * 2^-24 */
  if (shift >= 64) return sign;
  uint64_t r = m >> shift;
  uint64_t rem = m & ((1ull << shift) - 1ull);
  uint64_t half = 1ull << (shift - 1);
  if (rem > half || (rem == half && (r & 1ull))) r++;
  /* r has up to 11 bits (+carry); assemble: for normals the implicit bit adds to the exponent field */
  uint32_t out = he ? (uint32_t)(((uint32_t)(he - 1) << 10) + (uint32_t)r) : (uint32_t)r;
  if (out >= 0x7c00u) out = 0x7c00u; /* rounded up to infinity */
  return (uint16_t)(sign | out);
}
static inline uint16_t h_mul(uint16_t a, uint16_t b) { return orc_d2h((double)orc_h2f(a) * (double)orc_h2f(b)); }
static inline uint16_t h_fma(uint16_t a, uint16_t b, uint16_t c) {
  return orc_d2h((double)orc_h2f(a) * (double)orc_h2f(b) + (double)orc_h2f(c));
}

/* ------------------------------------------------------------------ RNG */
/* counter-based: value i of stream s under seed; splitmix64 finaliser */
static uint64_t mix64(uint64_t z) {
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
  return z ^ (z >> 31);
}
static float rng_sym(uint64_t seed, uint64_t stream, uint64_t i, float amp) {
  uint64_t h = mix64(seed + (stream + 1) * 0xD1B54A32D192ED03ull + i * 0x9E3779B97F4A7C15ull);
  uint32_t u = (uint32_t)(h >> 40);                /* 24 bits */
  float v = (float)u * (1.0f / 8388608.0f) - 1.0f; /* exact, in [-1,1) */
  return v * amp;
}

/* ------------------------------------------------------------------ field */

int orc_field_levels(const orc_field_desc* d, orc_level* out, uint32_t* total) {
  if (d->n_levels < 1 || d->n_levels > ORC_MAX_LEVELS) return -1;
  if (d->n_levels * d->n_features != 32) return -1;
  if (d->log2_hashmap < 4 || d->log2_hashmap > 28) return -1;
  double b = (d->n_levels > 1)
                 ? exp((log((double)d->finest_res) - log((double)d->base_res)) / (double)(d->n_levels - 1))
                 : 1.0;
  uint64_t T = 1ull << d->log2_hashmap;
  uint32_t off = 0;
  for (int l = 0; l < d->n_levels; l++) {
    uint32_t res;
    if (d->per_level_scale > 0.0f) {
      /* tiny-cuda-nn grid.h: scale = exp2f(level * log2f(per_level_scale)) * base_resolution - 1.0f (float32),
       * resolution = ceilf(scale) + 1 */
      float sc = exp2f((float)l * log2f(d->per_level_scale)) * (float)d->base_res - 1.0f;
      out[l].scale = sc;
      res = (uint32_t)ceilf(sc) + 1u;
    } else {
      double s = (double)d->base_res * pow(b, (double)l) - 1.0;
      /* guard pow() round-off so the nominal integer resolutions are hit exactly */
      double sr = floor(s + 0.5);
      if (fabs(s - sr) < 1e-9) s = sr;
      out[l].scale = (float)s;
      res = (uint32_t)ceil(s) + 1u;
    }
    out[l].res = res;
    uint64_t dense = (uint64_t)res * res * res;
    if (dense <= T) {
      out[l].hashed = 0;
      out[l].size = (uint32_t)((dense + 7ull) & ~7ull);
    } else {
      out[l].hashed = 1;
      out[l].size = (uint32_t)T;
    }
    out[l].offset = off;
    off += out[l].size;
  }
  *total = off;
  return 0;
}

static const int kLayerIn[5] = {32, 64, 32, 64, 64};
static const int kLayerOut[5] = {64, 16, 64, 64, 16};

/* analytic occupancy: union of spheres, cell centre test */
static const float kBlobs[4][4] = {
    {0.50f, 0.50f, 0.50f, 0.35f},
    {0.80f, 0.50f, 0.62f, 0.13f},
    {0.36f, 0.80f, 0.45f, 0.11f},
    {0.40f, 0.24f, 0.78f, 0.10f},
};

static void field_finalize(orc_field* f) {
  for (int i = 0; i < ORC_MLP_HALFS; i++) f->mlp_f[i] = orc_h2f(f->mlp[i]);
}

static orc_field* field_alloc(const orc_field_desc* d) {
  orc_field* f = (orc_field*)calloc(1, sizeof(orc_field));
  if (!f) return NULL;
  f->desc = *d;
  if (orc_field_levels(d, f->levels, &f->total_entries) != 0) {
    free(f);
    return NULL;
  }
  f->table = (uint16_t*)malloc((size_t)f->total_entries * d->n_features * sizeof(uint16_t));
  size_t R = (size_t)d->occ_res;
  f->occ = (uint32_t*)calloc((R * R * R + 31) / 32, sizeof(uint32_t));
  if (!f->table || !f->occ) {
    orc_field_free(f);
    return NULL;
  }
  return f;
}

orc_field* orc_field_synthetic(const orc_field_desc* d, uint64_t seed) {
  orc_field* f = field_alloc(d);
  if (!f) return NULL;
  size_t n = (size_t)f->total_entries * d->n_features;
  for (size_t i = 0; i < n; i++) f->table[i] = orc_f2h(rng_sym(seed, 0, i, d->table_amp));
  size_t o = 0;
  for (int l = 0; l < 5; l++) {
    float amp = sqrtf(6.0f / (float)(kLayerIn[l] + kLayerOut[l]));
    size_t cnt = (size_t)kLayerIn[l] * kLayerOut[l];
    for (size_t i = 0; i < cnt; i++) f->mlp[o + i] = orc_f2h(rng_sym(seed, (uint64_t)(l + 1), i, amp));
    o += cnt;
  }
  int R = d->occ_res;
  float invR = 1.0f / (float)R;
  for (int z = 0; z < R; z++)
    for (int y = 0; y < R; y++)
      for (int x = 0; x < R; x++) {
        float cx = ((float)x + 0.5f) * invR, cy = ((float)y + 0.5f) * invR, cz = ((float)z + 0.5f) * invR;
        int in = 0;
        for (int b = 0; b < 4 && !in; b++) {
          float dx = cx - kBlobs[b][0], dy = cy - kBlobs[b][1], dz = cz - kBlobs[b][2];
          float d2 = fmaf(dx, dx, fmaf(dy, dy, dz * dz));
          in = d2 <= kBlobs[b][3] * kBlobs[b][3];
        }
        if (in) {
          size_t bit = (size_t)x + (size_t)R * ((size_t)y + (size_t)R * (size_t)z);
          f->occ[bit >> 5] |= 1u << (bit & 31);
        }
      }
  field_finalize(f);
  return f;
}

orc_field* orc_field_from_params(const orc_field_desc* d, const uint16_t* table, const uint16_t* mlp,
                                 const uint32_t* occ) {
  orc_field* f = field_alloc(d);
  if (!f) return NULL;
  memcpy(f->table, table, (size_t)f->total_entries * d->n_features * sizeof(uint16_t));
  memcpy(f->mlp, mlp, sizeof(f->mlp));
  size_t R = (size_t)d->occ_res;
  memcpy(f->occ, occ, ((R * R * R + 31) / 32) * sizeof(uint32_t));
  field_finalize(f);
  return f;
}

void orc_field_free(orc_field* f) {
  if (!f) return;
  free(f->table);
  free(f->occ);
  free(f);
}

static inline float clamp01(float x) { return fminf(fmaxf(x, 0.0f), 1.0f); }

int orc_occupied(const orc_field* f, const float p[3]) {
  int R = f->desc.occ_res;
  int c[3];
  for (int a = 0; a < 3; a++) {
    int v = (int)(clamp01(p[a]) * (float)R);
    c[a] = v > R - 1 ? R - 1 : v;
  }
  size_t bit = (size_t)c[0] + (size_t)R * ((size_t)c[1] + (size_t)R * (size_t)c[2]);
  return (f->occ[bit >> 5] >> (bit & 31)) & 1u;
}

/* multiresolution hash encoding (published instant-ngp / tiny-cuda-nn algorithm,
 * restated; NOT in the reference tree -- reached through pyngp, run.py:25).
 * Deviation, documented: corner coordinates are clamped to res-1 instead of the
 * dense index being wrapped modulo the level size. */
void orc_encode(const orc_field* f, const float p_in[3], uint16_t feat[32]) {
  const int F = f->desc.n_features;
  float p[3] = {clamp01(p_in[0]), clamp01(p_in[1]), clamp01(p_in[2])};
  for (int l = 0; l < f->desc.n_levels; l++) {
    const orc_level* L = &f->levels[l];
    uint32_t c0[3];
    float w[3];
    for (int a = 0; a < 3; a++) {
      float pos = fmaf(L->scale, p[a], 0.5f);
      float fl = floorf(pos);
      w[a] = pos - fl;
      c0[a] = (uint32_t)(int)fl;
    }
    /* trilinear blend in binary16, as tiny-cuda-nn does for fp16 tables: weights rounded to fp16,
     * w = fp16(fp16(wx*wy)*wz), acc = fp16 fma(w, v, acc) over the corners in order dx + 2dy + 4dz */
    uint16_t wh[3][2];
    for (int a = 0; a < 3; a++) {
      wh[a][0] = orc_f2h(1.0f - w[a]);
      wh[a][1] = orc_f2h(w[a]);
    }
    uint16_t acc[4] = {0, 0, 0, 0};
    for (int c = 0; c < 8; c++) {
      uint32_t cc[3];
      for (int a = 0; a < 3; a++) {
        uint32_t bit = (c >> a) & 1u;
        uint32_t v = c0[a] + bit;
        cc[a] = v > L->res - 1 ? L->res - 1 : v;
      }
      uint16_t weight = h_mul(h_mul(wh[0][c & 1], wh[1][(c >> 1) & 1]), wh[2][c >> 2]);
      uint32_t idx;
      if (L->hashed)
        idx = (cc[0] ^ (cc[1] * 2654435761u) ^ (cc[2] * 805459861u)) & (L->size - 1u);
      else
        idx = cc[0] + L->res * (cc[1] + L->res * cc[2]);
      const uint16_t* e = f->table + ((size_t)L->offset + idx) * F;
      for (int k = 0; k < F; k++) acc[k] = h_fma(weight, e[k], acc[k]);
    }
    for (int k = 0; k < F; k++) feat[l * F + k] = acc[k];
  }
}

/* real spherical harmonics, degree 4 (16 coefficients), direction in [-1,1]^3 */
void orc_sh4(const float d[3], float o[16]) {
  float x = d[0], y = d[1], z = d[2];
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

/* one fully-connected layer: fp16 inputs and weights, wide accumulation */
static void layer(const float* W, int n_in, int n_out, const float* in, float* out) {
  double acc[64];
  for (int o = 0; o < n_out; o++) acc[o] = 0.0;
  for (int k = 0; k < n_in; k++) { /* k ascending per output; products exact in double */
    double x = (double)in[k];
    const float* w = W + k * n_out;
    for (int o = 0; o < n_out; o++) acc[o] += x * (double)w[o];
  }
  for (int o = 0; o < n_out; o++) out[o] = (float)acc[o];
}
static void relu_round(float* v, int n) {
  for (int i = 0; i < n; i++) v[i] = orc_h2f(orc_f2h(v[i] > 0.0f ? v[i] : 0.0f));
}

static void eval_with_sh(const orc_field* f, const float p[3], const float sh[16], float* sigma,
                         float rgb[3], float* mlp_out) {
  uint16_t feat[32];
  orc_encode(f, p, feat);
  float in[32], h[64], h2[64], od[16], orr[16];
  for (int k = 0; k < 32; k++) in[k] = orc_h2f(feat[k]);
  const float* W = f->mlp_f;
  layer(W, 32, 64, in, h);
  relu_round(h, 64);
  layer(W + 2048, 64, 16, h, od);
  *sigma = expf(od[0] + f->desc.density_bias);
  for (int k = 0; k < 16; k++) in[k] = orc_h2f(orc_f2h(od[k]));
  for (int k = 0; k < 16; k++) in[16 + k] = orc_h2f(orc_f2h(sh[k]));
  layer(W + 3072, 32, 64, in, h);
  relu_round(h, 64);
  layer(W + 5120, 64, 64, h, h2);
  relu_round(h2, 64);
  layer(W + 9216, 64, 16, h2, orr);
  for (int k = 0; k < 3; k++) rgb[k] = 1.0f / (1.0f + expf(-orr[k]));
  if (mlp_out) {
    memcpy(mlp_out, od, sizeof(od));
    memcpy(mlp_out + 16, orr, sizeof(orr));
  }
}

void orc_eval(const orc_field* f, const float p[3], const float d[3], float* sigma, float rgb[3],
              float* mlp_out) {
  float sh[16];
  orc_sh4(d, sh);
  eval_with_sh(f, p, sh, sigma, rgb, mlp_out);
}

/* ------------------------------------------------------------------ 4x4 helpers (double) */

static void mat4_mul(const double* a, const double* b, double* o) {
  double t[16];
  for (int i = 0; i < 4; i++)
    for (int j = 0; j < 4; j++) {
      double s = 0;
      for (int k = 0; k < 4; k++) s += a[i * 4 + k] * b[k * 4 + j];
      t[i * 4 + j] = s;
    }
  memcpy(o, t, sizeof(t));
}
static void mat4_inv(const double* m, double* o) {
  double a[4][8];
  for (int i = 0; i < 4; i++)
    for (int j = 0; j < 4; j++) {
      a[i][j] = m[i * 4 + j];
      a[i][4 + j] = i == j;
    }
  for (int c = 0; c < 4; c++) {
    int piv = c;
    for (int r = c + 1; r < 4; r++)
      if (fabs(a[r][c]) > fabs(a[piv][c])) piv = r;
    if (piv != c)
      for (int j = 0; j < 8; j++) {
        double t = a[c][j];
        a[c][j] = a[piv][j];
        a[piv][j] = t;
      }
    double d = a[c][c]; /* singular -> inf/nan propagate, like Eigen */
    for (int j = 0; j < 8; j++) a[c][j] /= d;
    for (int r = 0; r < 4; r++)
      if (r != c) {
        double fct = a[r][c];
        if (fct != 0.0)
          for (int j = 0; j < 8; j++) a[r][j] -= fct * a[c][j];
      }
  }
  for (int i = 0; i < 4; i++)
    for (int j = 0; j < 4; j++) o[i * 4 + j] = a[i][4 + j];
}
static void vec3_normalized(double* v) { /* Eigen normalized(): zero stays zero */
  double n2 = v[0] * v[0] + v[1] * v[1] + v[2] * v[2];
  if (n2 > 0) {
    double n = sqrt(n2);
    v[0] /= n;
    v[1] /= n;
    v[2] /= n;
  }
}
static void cross3(const double* a, const double* b, double* o) {
  o[0] = a[1] * b[2] - a[2] * b[1];
  o[1] = a[2] * b[0] - a[0] * b[2];
  o[2] = a[0] * b[1] - a[1] * b[0];
}

/* View::get_next_camera_pos, type_of_pose = 0, now_camera_pose_world = I
 * (View_Space.hpp:67-140; the only live configuration, Share_Data.hpp:475). */
void orc_view_pose(const double init_pos[3], const double center[3], double pose[16]) {
  double view[3] = {init_pos[0], init_pos[1], init_pos[2]};
  double Z[3] = {center[0] - view[0], center[1] - view[1], center[2] - view[2]};
  vec3_normalized(Z); /* :79 */
  double X[3], Y[3];
  cross3(Z, view, X); /* :81 */
  vec3_normalized(X);
  cross3(Z, X, Y); /* :82 */
  vec3_normalized(Y);
  double T[16] = {1, 0, 0, -view[0], 0, 1, 0, -view[1], 0, 0, 1, -view[2], 0, 0, 0, 1}; /* :83-87 */
  double R[16] = {X[0], Y[0], Z[0], 0, X[1], Y[1], Z[1], 0, X[2], Y[2], Z[2], 0, 0, 0, 0, 1}; /* :88-92 */
  double Rz_min[16] = {1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1};
  double M[16], Mi[16], MT[16];
  mat4_inv(R, Mi);
  mat4_mul(Mi, T, MT); /* R^-1 * T */
  /* x_ray = MT*(1,0,0,1), y_ray = MT*(0,1,0,1)  :95-100 */
  double x0 = MT[0] + MT[3];
  double y1 = MT[5] + MT[7];
  double min_y = acos(y1); /* :101 */
  double min_x = acos(x0); /* :102 */
  for (double i = 5; i < 360; i += 5) { /* :103 */
    double a = i * acos(-1.0) / 180.0;
    /* AngleAxis product goes through a quaternion (0,0,sin(a/2); cos(a/2)) then
     * toRotationMatrix() */
    double qz = sin(a / 2), qw = cos(a / 2);
    double tz = 2 * qz, twz = tz * qw, tzz = tz * qz;
    double Rz[16] = {1 - tzz, -twz, 0, 0, twz, 1 - tzz, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1};
    mat4_mul(R, Rz, M);
    mat4_inv(M, Mi);
    mat4_mul(Mi, T, MT);
    double cx = acos(MT[0] + MT[3]); /* :118 */
    double cy = acos(MT[5] + MT[7]); /* :117 */
    if (cy < min_y) { /* :119 (NaN compares false) */
      memcpy(Rz_min, Rz, sizeof(Rz));
      min_y = cy;
      min_x = cx;
    } else if (fabs(cy - min_y) < 1e-6 && cx < min_x) { /* :124 */
      memcpy(Rz_min, Rz, sizeof(Rz));
      min_y = cy;
      min_x = cx;
    }
  }
  mat4_mul(R, Rz_min, M);
  mat4_inv(M, Mi);
  mat4_mul(Mi, T, pose); /* :137 */
}

/* main.cpp:1626-1641 with now_camera_pose_world = I */
void orc_transform_matrix(const double pose[16], double out[16]) {
  static const double P[16] = {0, 0, 1, 0, 1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 0, 1};
  static const double P1[16] = {1, 0, 0, 0, 0, -1, 0, 0, 0, 0, -1, 0, 0, 0, 0, 1};
  double vi[16], t[16];
  mat4_inv(pose, vi);
  mat4_mul(P, vi, t);
  mat4_mul(t, P1, out);
}

/* View_Space.hpp:550-556 */
int orc_view_space(const double* pt, int n, double radius, const double center[3], double* out_pos) {
  double pt_norm = sqrt(pt[0] * pt[0] + pt[1] * pt[1] + pt[2] * pt[2]); /* Share_Data.hpp:527-528 */
  int k = 0;
  for (int i = 0; i < n; i++) {
    if (pt[i * 3 + 2] < 0) continue;
    double scale = 1.0 / pt_norm * radius;
    for (int a = 0; a < 3; a++) out_pos[k * 3 + a] = pt[i * 3 + a] * scale + center[a];
    k++;
  }
  return k;
}

/* View_Space.hpp:534-548 */
void orc_bbx(const double* pts, int n, double center[3], double* predicted_size) {
  center[0] = center[1] = center[2] = 0;
  for (int i = 0; i < n; i++)
    for (int a = 0; a < 3; a++) center[a] += pts[i * 3 + a];
  for (int a = 0; a < 3; a++) center[a] /= n;
  double sz = 0;
  for (int i = 0; i < n; i++) {
    double dx = center[0] - pts[i * 3], dy = center[1] - pts[i * 3 + 1], dz = center[2] - pts[i * 3 + 2];
    double nn = sqrt(dx * dx + dy * dy + dz * dz);
    if (nn > sz) sz = nn;
  }
  *predicted_size = sz * (17.0 / 16.0);
}

/* ASSUMED (instant-ngp nerf_matrix_to_ngp, not in tree): negate columns 1,2,
 * t*scale+offset, cycle rows (y,z,x). */
void orc_nerf_to_ngp(const double tm[16], double scale, const double offset[3], float out[12]) {
  double m[3][4];
  for (int r = 0; r < 3; r++) {
    m[r][0] = tm[r * 4 + 0];
    m[r][1] = -tm[r * 4 + 1];
    m[r][2] = -tm[r * 4 + 2];
    m[r][3] = tm[r * 4 + 3] * scale + offset[r];
  }
  static const int src[3] = {1, 2, 0};
  for (int r = 0; r < 3; r++)
    for (int c = 0; c < 4; c++) out[r * 4 + c] = (float)m[src[r]][c];
}

/* intr = {ppx, ppy, fx, fy, c0..c4}; model 2 = inverse Brown-Conrady, 1 = modified, 0 = none
 * Share_Data.hpp:92-137 (only the Brown-Conrady branches are reachable: yaml color_model = 2) */
void orc_rs2_project(float pixel[2], const float in[9], int model, const float point[3]) {
  float x = point[0] / point[2], y = point[1] / point[2];
  const float* c = in + 4;
  if (model == 1 || model == 2) {
    float r2 = x * x + y * y;
    float f = 1 + c[0] * r2 + c[1] * r2 * r2 + c[4] * r2 * r2 * r2;
    x *= f;
    y *= f;
    float dx = x + 2 * c[2] * x * y + c[3] * (r2 + 2 * x * x);
    float dy = y + 2 * c[3] * x * y + c[2] * (r2 + 2 * y * y);
    x = dx;
    y = dy;
  }
  pixel[0] = x * in[2] + in[0];
  pixel[1] = y * in[3] + in[1];
}
/* Share_Data.hpp:140-196 */
void orc_rs2_deproject(float point[3], const float in[9], int model, const float pixel[2], float depth) {
  float x = (pixel[0] - in[0]) / in[2];
  float y = (pixel[1] - in[1]) / in[3];
  const float* c = in + 4;
  if (model == 2) {
    float r2 = x * x + y * y;
    float f = 1 + c[0] * r2 + c[1] * r2 * r2 + c[4] * r2 * r2 * r2;
    float ux = x * f + 2 * c[2] * x * y + c[3] * (r2 + 2 * x * x);
    float uy = y * f + 2 * c[3] * x * y + c[2] * (r2 + 2 * y * y);
    x = ux;
    y = uy;
  }
  point[0] = depth * x;
  point[1] = depth * y;
  point[2] = depth;
}

/* ------------------------------------------------------------------ rays */

/* R2 low-discrepancy sub-pixel sequence; k = 0 is the pixel centre.  The
 * reference's jitter comes from instant-ngp's RNG (unpinned); this is the
 * build's own deterministic choice. */
void orc_spp_offset(int k, float* ox, float* oy) {
  float fk = (float)k;
  float a = fmaf(fk, 0.7548776662466927f, 0.5f);
  float b = fmaf(fk, 0.5698402909980532f, 0.5f);
  *ox = a - floorf(a);
  *oy = b - floorf(b);
}

/* forward model and its Jacobian, every operation spelled out (fmaf order is part of the contract
 * with the device twin in prv_device.hpp) */
static void lens_eval(const float L[4], float x, float y, float* fx, float* fy, float J[4]) {
  const float k1 = L[0], k2 = L[1], p1 = L[2], p2 = L[3];
  const float r2 = fmaf(x, x, y * y);
  const float kr = fmaf(k2, r2, k1);         /* k1 + k2 r^2 */
  const float radial = fmaf(kr, r2, 1.0f);   /* 1 + k1 r^2 + k2 r^4 */
  const float dk = 2.0f * fmaf(2.0f * k2, r2, k1); /* d radial / d r^2, times 2 */
  const float xy = x * y;
  *fx = fmaf(x, radial, fmaf(2.0f * p1, xy, p2 * fmaf(2.0f * x, x, r2)));
  *fy = fmaf(y, radial, fmaf(p1, fmaf(2.0f * y, y, r2), (2.0f * p2) * xy));
  if (J) {
    const float a = 2.0f * fmaf(p1, x, p2 * y); /* shared off-diagonal tangential part */
    J[0] = fmaf(x * x, dk, fmaf(2.0f * p1, y, fmaf(6.0f * p2, x, radial)));
    J[1] = fmaf(xy, dk, a);
    J[2] = J[1];
    J[3] = fmaf(y * y, dk, fmaf(6.0f * p1, y, fmaf(2.0f * p2, x, radial)));
  }
}

void orc_lens_distort(const float lens[4], float x, float y, float* xd, float* yd) { lens_eval(lens, x, y, xd, yd, NULL); }

void orc_lens_undistort(const float lens[4], float* px, float* py) {
  const float xd = *px, yd = *py;
  float x = xd, y = yd;
  for (int it = 0; it < ORC_LENS_ITERS; it++) {
    float fx, fy, J[4];
    lens_eval(lens, x, y, &fx, &fy, J);
    const float ex = fx - xd, ey = fy - yd;
    const float det = fmaf(J[0], J[3], -(J[1] * J[2]));
    const float sx = fmaf(J[3], ex, -(J[1] * ey)) / det;
    const float sy = fmaf(J[0], ey, -(J[2] * ex)) / det;
    x -= sx;
    y -= sy;
  }
  *px = x;
  *py = y;
}

void orc_raygen(const orc_camera* cam, int px, int py, float ox, float oy, float o[3], float d[3]) {
  float dx = (((float)px + ox) - cam->cx) / cam->fx;
  float dy = (((float)py + oy) - cam->cy) / cam->fy;
  if (cam->lens[0] != 0.0f || cam->lens[1] != 0.0f || cam->lens[2] != 0.0f || cam->lens[3] != 0.0f)
    orc_lens_undistort(cam->lens, &dx, &dy);
  float v[3];
  for (int r = 0; r < 3; r++) {
    const float* m = cam->c2w + r * 4;
    v[r] = fmaf(m[0], dx, fmaf(m[1], dy, m[2]));
    o[r] = m[3];
  }
  float n2 = fmaf(v[0], v[0], fmaf(v[1], v[1], v[2] * v[2]));
  float inv = 1.0f / sqrtf(n2);
  for (int r = 0; r < 3; r++) d[r] = v[r] * inv;
}

int orc_ray_aabb(const float o[3], const float d[3], float* t0, float* t1) {
  float tmin = 0.0f, tmax = INFINITY;
  for (int a = 0; a < 3; a++) {
    float inv = 1.0f / d[a];
    float ta = (0.0f - o[a]) * inv;
    float tb = (1.0f - o[a]) * inv;
    tmin = fmaxf(tmin, fminf(ta, tb));
    tmax = fminf(tmax, fmaxf(ta, tb));
  }
  *t0 = tmin;
  *t1 = tmax;
  return tmax > tmin;
}

/* one ray, front-to-back compositing with early termination at T < min_T, in one of two sampling rules:
 *  ORC_STEP_FIXED_S : S uniform samples between AABB entry and exit (the BASELINE configs' fixed sample count);
 *  ORC_STEP_NGP     : what pyngp.Testbed.render does behind run.py:245-247, 304 for aabb_scale = 1 (SURVEY App. E;
 *                     the published instant-ngp render loop -- the engine itself is not in the reference tree):
 *                     a fixed step dt = sqrt(3)/1024 from the AABB entry, sample i at t0 + (i + 1/2) dt while that is
 *                     inside the box (at most ORC_NGP_MAX_STEPS = 1024, the cube's diagonal), every step tested
 *                     against the occupancy grid, alpha = 1 - exp(-sigma dt) with that same dt.
 * n_live counts the samples that pass the occupancy test BEFORE early termination (the march count: no field
 * evaluation needed, see orc_march_count_rows); n_eval the samples actually evaluated and composited. */
static inline int ray_steps(int step_mode, int S, float t0, float t1, float* dt) {
  if (step_mode == ORC_STEP_NGP) {
    *dt = sqrtf(3.0f) / 1024.0f;
    return ORC_NGP_MAX_STEPS;
  }
  *dt = (t1 - t0) / (float)S;
  return S;
}

static void march_ray(const orc_field* f, const float o[3], const float d[3], int step_mode, int S, float min_T,
                      float out[4], uint64_t* n_eval) {
  out[0] = out[1] = out[2] = out[3] = 0.0f;
  float t0, t1;
  if (!orc_ray_aabb(o, d, &t0, &t1)) return;
  float dt;
  const int n = ray_steps(step_mode, S, t0, t1, &dt);
  float sh[16];
  orc_sh4(d, sh);
  float T = 1.0f, r = 0.0f, g = 0.0f, b = 0.0f;
  for (int i = 0; i < n; i++) {
    float t = fmaf((float)i + 0.5f, dt, t0);
    if (step_mode == ORC_STEP_NGP && !(t < t1)) break; /* left the box */
    float p[3] = {fmaf(t, d[0], o[0]), fmaf(t, d[1], o[1]), fmaf(t, d[2], o[2])};
    if (!orc_occupied(f, p)) continue;
    float sigma, c[3];
    eval_with_sh(f, p, sh, &sigma, c, NULL);
    float alpha = 1.0f - expf(-(sigma * dt));
    float wgt = alpha * T;
    r = fmaf(wgt, c[0], r);
    g = fmaf(wgt, c[1], g);
    b = fmaf(wgt, c[2], b);
    T = T * (1.0f - alpha);
    (*n_eval)++;
    if (T < min_T) break;
  }
  out[0] = r;
  out[1] = g;
  out[2] = b;
  out[3] = 1.0f - T;
}

/* the march alone: how many samples of the ray lie in occupied cells (no field evaluation, no early termination) */
static uint64_t march_count_ray(const orc_field* f, const float o[3], const float d[3], int step_mode, int S) {
  float t0, t1;
  if (!orc_ray_aabb(o, d, &t0, &t1)) return 0;
  float dt;
  const int n = ray_steps(step_mode, S, t0, t1, &dt);
  uint64_t live = 0;
  for (int i = 0; i < n; i++) {
    float t = fmaf((float)i + 0.5f, dt, t0);
    if (step_mode == ORC_STEP_NGP && !(t < t1)) break;
    float p[3] = {fmaf(t, d[0], o[0]), fmaf(t, d[1], o[1]), fmaf(t, d[2], o[2])};
    live += orc_occupied(f, p) ? 1u : 0u;
  }
  return live;
}

typedef struct {
  const orc_field* f;
  const orc_camera* cam;
  int w, h, y0, y1, step_mode, S, spp, tid, nth;
  float min_T;
  float* rgba; /* NULL: march count only */
  uint64_t n_eval;
} render_job;

static void* render_worker(void* arg) {
  render_job* j = (render_job*)arg;
  float inv_spp = 1.0f / (float)j->spp;
  uint64_t n_eval = 0; /* thread-local: the jobs sit side by side in memory, a shared-line counter would throttle the baseline */
  for (int y = j->y0 + j->tid; y < j->y1; y += j->nth)
    for (int x = 0; x < j->w; x++) {
      float acc[4] = {0, 0, 0, 0};
      for (int k = 0; k < j->spp; k++) {
        float ox, oy, o[3], d[3], px[4];
        orc_spp_offset(k, &ox, &oy);
        orc_raygen(j->cam, x, y, ox, oy, o, d);
        if (!j->rgba) {
          n_eval += march_count_ray(j->f, o, d, j->step_mode, j->S);
          continue;
        }
        march_ray(j->f, o, d, j->step_mode, j->S, j->min_T, px, &n_eval);
        for (int c = 0; c < 4; c++) acc[c] += px[c];
      }
      if (!j->rgba) continue;
      float* dst = j->rgba + ((size_t)y * j->w + x) * 4;
      for (int c = 0; c < 4; c++) dst[c] = acc[c] * inv_spp;
    }
  j->n_eval = n_eval;
  return NULL;
}

static uint64_t run_rows(const orc_field* f, const orc_camera* cam, int w, int h, int y0, int y1, int step_mode, int S,
                         int spp, float min_T, float* rgba, int n_threads) {
  if (n_threads < 1) n_threads = 1;
  if (n_threads > 256) n_threads = 256;
  render_job jobs[256];
  pthread_t th[256];
  for (int t = 0; t < n_threads; t++) {
    render_job jb = {f, cam, w, h, y0, y1, step_mode, S, spp, t, n_threads, min_T, rgba, 0};
    jobs[t] = jb;
    if (t > 0) pthread_create(&th[t], NULL, render_worker, &jobs[t]);
  }
  render_worker(&jobs[0]);
  uint64_t tot = jobs[0].n_eval;
  for (int t = 1; t < n_threads; t++) {
    pthread_join(th[t], NULL);
    tot += jobs[t].n_eval;
  }
  return tot;
}

void orc_render_rows_mode(const orc_field* f, const orc_camera* cam, int w, int h, int y0, int y1, int step_mode, int S,
                          int spp, float min_T, float* rgba, uint64_t* n_evaluated, int n_threads) {
  const uint64_t tot = run_rows(f, cam, w, h, y0, y1, step_mode, S, spp, min_T, rgba, n_threads);
  if (n_evaluated) *n_evaluated = tot;
}

uint64_t orc_march_count_rows(const orc_field* f, const orc_camera* cam, int w, int h, int y0, int y1, int step_mode,
                              int S, int spp, int n_threads) {
  return run_rows(f, cam, w, h, y0, y1, step_mode, S, spp, 0.0f, NULL, n_threads);
}

void orc_render_rows(const orc_field* f, const orc_camera* cam, int w, int h, int y0, int y1, int S,
                     int spp, float min_T, float* rgba, uint64_t* n_evaluated, int n_threads) {
  orc_render_rows_mode(f, cam, w, h, y0, y1, ORC_STEP_FIXED_S, S, spp, min_T, rgba, n_evaluated, n_threads);
}

void orc_render(const orc_field* f, const orc_camera* cam, int w, int h, int S, int spp, float min_T,
                float* rgba, uint64_t* n_evaluated, int n_threads) {
  orc_render_rows(f, cam, w, h, 0, h, S, spp, min_T, rgba, n_evaluated, n_threads);
}

/* ------------------------------------------------------------------ image post + scores */

float orc_linear_to_srgb(float x) {
  if (x <= 0.0031308f) return 12.92f * x;
  return 1.055f * powf(x, 0.41666666f) - 0.055f; /* 1/2.4 */
}

/* ASSUMED from upstream scripts/common.py write_image (absent from the tree):
 * composite over background, un-premultiply, sRGB, clip, *255 + 0.5 truncate. */
void orc_quantize_rgba8(const float* rgba, size_t npix, const float bg[4], uint8_t* out) {
  for (size_t i = 0; i < npix; i++) {
    const float* s = rgba + i * 4;
    float a = s[3];
    float rem = 1.0f - a;
    float c[4] = {fmaf(rem, bg[0], s[0]), fmaf(rem, bg[1], s[1]), fmaf(rem, bg[2], s[2]), fmaf(rem, bg[3], a)};
    for (int k = 0; k < 3; k++) {
      float v = c[3] != 0.0f ? c[k] / c[3] : c[k];
      v = orc_linear_to_srgb(v);
      v = fminf(fmaxf(v, 0.0f), 1.0f);
      out[i * 4 + k] = (uint8_t)(v * 255.0f + 0.5f);
    }
    float v = fminf(fmaxf(c[3], 0.0f), 1.0f);
    out[i * 4 + 3] = (uint8_t)(v * 255.0f + 0.5f);
  }
}

/* main.cpp:2053-2086 -- channels 0..2 of a 4-channel uint8 image, E members */
double orc_score_ensemble_rgb(const uint8_t* const* imgs, int E, size_t npix) {
  double view_uncertainty = 0.0;
  for (size_t p = 0; p < npix; p++) {
    /* cv::imread(IMREAD_UNCHANGED) hands the reference BGRA pixels (:2049): its "r, g, b" = bytes 2, 1, 0 of the
     * PNG's RGBA order, and that is the order its three addends enter the sum */
    for (int c = 2; c >= 0; c--) {
      double mean = 0.0;
      for (int e = 0; e < E; e++) mean += imgs[e][p * 4 + c];
      mean /= E;
      double var = 0.0;
      for (int e = 0; e < E; e++) {
        double dlt = imgs[e][p * 4 + c] - mean;
        var += dlt * dlt;
      }
      var /= E;
      if (var > 1e-10) view_uncertainty += log(var); /* :2082-2084 */
    }
  }
  return view_uncertainty;
}

/* main.cpp:2113-2150 */
double orc_score_ensemble_rgbdensity(const uint8_t* const* imgs, int E, size_t npix) {
  double view_uncertainty = 0.0;
  for (size_t p = 0; p < npix; p++) {
    double var3[3]; /* reference channel k = byte 2 - k (BGRA, see above) */
    for (int k = 0; k < 3; k++) {
      const int c = 2 - k;
      double mean = 0.0;
      for (int e = 0; e < E; e++) mean += imgs[e][p * 4 + c];
      mean /= E;
      double var = 0.0;
      for (int e = 0; e < E; e++) {
        double dlt = imgs[e][p * 4 + c] - mean;
        var += dlt * dlt;
      }
      var3[k] = var / E;
    }
    double mean_density = 0.0;
    for (int e = 0; e < E; e++) mean_density += imgs[e][p * 4 + 3] / 255.0; /* :2127 */
    mean_density /= E;
    view_uncertainty += (var3[0] + var3[1] + var3[2]) / 3.0;         /* :2147 */
    view_uncertainty += (1.0 - mean_density) * (1.0 - mean_density); /* :2148 */
  }
  return view_uncertainty;
}

/* run.py:257-263: A = clip(srgb(img)), R = clip(srgb(ref)), mse over HxWx3, psnr = -10 log10(mse).
 * Images are first composited over the background (run.py:226).  coverage = mean opacity. */
void orc_score_psnr_coverage(const float* rgba, const float* gt, size_t npix, const float bg[4],
                             double* psnr, double* coverage) {
  double se = 0.0, cov = 0.0;
  for (size_t i = 0; i < npix; i++) {
    float ra = 1.0f - rgba[i * 4 + 3], rg = 1.0f - gt[i * 4 + 3];
    for (int k = 0; k < 3; k++) {
      float a = fmaf(ra, bg[k], rgba[i * 4 + k]);
      float r = fmaf(rg, bg[k], gt[i * 4 + k]);
      a = fminf(fmaxf(orc_linear_to_srgb(a), 0.0f), 1.0f);
      r = fminf(fmaxf(orc_linear_to_srgb(r), 0.0f), 1.0f);
      double dlt = (double)a - (double)r;
      se += dlt * dlt;
    }
    cov += rgba[i * 4 + 3];
  }
  double mse = se / (double)(npix * 3);
  *psnr = -10.0 * log10(mse);
  *coverage = cov / (double)npix;
}

/* the ranking key of PRV_SCORE_PSNR_COVERAGE: -PSNR + weight * mean((1 - alpha)^2) -- the second term is the
 * density term of main.cpp:2148 (there per pixel of the ensemble mean, weight 1), averaged over the pixels */
void orc_score_view(const float* rgba, const float* gt, size_t npix, const float bg[4], double coverage_weight,
                    double* score, double* psnr, double* coverage) {
  double unc = 0.0;
  orc_score_psnr_coverage(rgba, gt, npix, bg, psnr, coverage);
  for (size_t i = 0; i < npix; i++) {
    double u = 1.0 - (double)rgba[i * 4 + 3];
    unc += u * u;
  }
  *score = -*psnr + coverage_weight * (unc / (double)npix);
}

/* SSIM as run.py:260 calls it: compute_error("SSIM", A, R) on the sRGB-clipped images.
 * ASSUMED from upstream scripts/common.py (absent from the tree, unpinned): luminance
 * 0.2126 r' + 0.7152 g' + 0.0722 b' with c' = max(0,c)^0.4545454545, separable 5-tap blur
 * [0.120078 0.233881 0.292082 0.233881 0.120078] over the valid region, c1 = 0.01^2,
 * c2 = 0.03^2, mean of the SSIM map.  float32 arithmetic, taps accumulated in ascending order. */
static const float kSsimTap[5] = {0.120078f, 0.233881f, 0.292082f, 0.233881f, 0.120078f};
static float ssim_lum(const float* rgba, const float bg[4]) {
  float rem = 1.0f - rgba[3];
  float l = 0.0f;
  static const float kw[3] = {0.2126f, 0.7152f, 0.0722f};
  for (int k = 0; k < 3; k++) {
    float c = fmaf(rem, bg[k], rgba[k]);
    c = fminf(fmaxf(orc_linear_to_srgb(c), 0.0f), 1.0f); /* A / R of run.py:257-258 */
    l = fmaf(kw[k], powf(fmaxf(c, 0.0f), 0.4545454545f), l);
  }
  return l;
}
static float ssim_blur(const float* img, int w, int x, int y) { /* img = 5 rows window origin */
  float rows[5];
  for (int i = 0; i < 5; i++) {
    float acc = 0.0f;
    for (int j = 0; j < 5; j++) acc = fmaf(kSsimTap[j], img[(size_t)(y + i) * w + (x + j)], acc);
    rows[i] = acc;
  }
  float acc = 0.0f;
  for (int i = 0; i < 5; i++) acc = fmaf(kSsimTap[i], rows[i], acc);
  return acc;
}
double orc_ssim(const float* rgba, const float* gt, int w, int h, const float bg[4]) {
  if (w < 5 || h < 5) return 0.0;
  size_t n = (size_t)w * h;
  float* la = (float*)malloc(n * 5 * sizeof(float));
  float *lb = la + n, *aa = lb + n, *bb = aa + n, *ab = bb + n;
  for (size_t i = 0; i < n; i++) {
    la[i] = ssim_lum(rgba + i * 4, bg);
    lb[i] = ssim_lum(gt + i * 4, bg);
    aa[i] = la[i] * la[i];
    bb[i] = lb[i] * lb[i];
    ab[i] = la[i] * lb[i];
  }
  const float c1 = 0.01f * 0.01f, c2 = 0.03f * 0.03f;
  double sum = 0.0;
  for (int y = 0; y + 4 < h; y++)
    for (int x = 0; x + 4 < w; x++) {
      float mA = ssim_blur(la, w, x, y), mB = ssim_blur(lb, w, x, y);
      float sA = ssim_blur(aa, w, x, y) - mA * mA;
      float sB = ssim_blur(bb, w, x, y) - mB * mB;
      float sAB = ssim_blur(ab, w, x, y) - mA * mB;
      float p1 = (2.0f * mA * mB + c1) / (mA * mA + mB * mB + c1);
      float p2 = (2.0f * sAB + c2) / (sA + sB + c2);
      sum += (double)(p1 * p2);
    }
  free(la);
  return sum / (double)((size_t)(w - 4) * (h - 4));
}

/* arg-max with strict '>' over ascending ids, initial best -1e100 (main.cpp:1971, 2088-2091) */
int orc_argmax(const double* scores, const int* ids, int n) {
  double best = -1e100;
  int best_id = -1;
  for (int i = 0; i < n; i++)
    if (scores[i] > best) {
      best = scores[i];
      best_id = ids[i];
    }
  return best_id;
}

/* full ranking = repeated arg-max: descending score, ties -> lower id first; a NaN score (never selected by
 * the strict '>' of main.cpp:2088) ranks after every number, NaNs among themselves by id */
void orc_rank(const double* scores, const int* ids, int n, int* order) {
  for (int i = 0; i < n; i++) order[i] = i;
  for (int i = 1; i < n; i++) { /* insertion sort, stable */
    int k = order[i];
    int j = i - 1;
    while (j >= 0) {
      int q = order[j];
      int nk = scores[k] != scores[k], nq = scores[q] != scores[q];
      int before = nk != nq ? nq
                            : (!nk && scores[k] > scores[q]) || ((nk || scores[k] == scores[q]) && ids[k] < ids[q]);
      if (!before) break;
      order[j + 1] = q;
      j--;
    }
    order[j + 1] = k;
  }
  for (int i = 0; i < n; i++) order[i] = ids[order[i]];
}

/* first occupied cell along a ray (Amanatides-Woo DDA over the occupancy grid): the
 * build's analogue of Perception_3D::precept_thread_process's castRay (main.cpp:253-258) */
int orc_first_hit(const orc_field* f, const float o[3], const float d[3], float max_range, int cell[3]) {
  float t0, t1;
  if (!orc_ray_aabb(o, d, &t0, &t1)) return 0;
  if (t1 > max_range) t1 = max_range;
  if (t1 <= t0) return 0;
  int R = f->desc.occ_res;
  float fR = (float)R;
  float ts = t0 + 1e-6f;
  int c[3], step[3];
  float tmax[3], tdelta[3];
  for (int a = 0; a < 3; a++) {
    float p = clamp01(fmaf(ts, d[a], o[a])) * fR;
    int v = (int)p;
    c[a] = v > R - 1 ? R - 1 : v;
    if (d[a] > 0) {
      step[a] = 1;
      tmax[a] = (((float)(c[a] + 1)) / fR - o[a]) / d[a];
      tdelta[a] = 1.0f / (fR * d[a]);
    } else if (d[a] < 0) {
      step[a] = -1;
      tmax[a] = (((float)c[a]) / fR - o[a]) / d[a];
      tdelta[a] = -1.0f / (fR * d[a]);
    } else {
      step[a] = 0;
      tmax[a] = INFINITY;
      tdelta[a] = INFINITY;
    }
  }
  for (;;) {
    size_t bit = (size_t)c[0] + (size_t)R * ((size_t)c[1] + (size_t)R * (size_t)c[2]);
    if ((f->occ[bit >> 5] >> (bit & 31)) & 1u) {
      cell[0] = c[0];
      cell[1] = c[1];
      cell[2] = c[2];
      return 1;
    }
    int a = tmax[0] < tmax[1] ? (tmax[0] < tmax[2] ? 0 : 2) : (tmax[1] < tmax[2] ? 1 : 2);
    if (tmax[a] > t1) return 0;
    c[a] += step[a];
    if (c[a] < 0 || c[a] >= R) return 0;
    tmax[a] += tdelta[a];
  }
}

/* Perception_3D::precept_thread_process for every voxel (main.cpp:238-284), on the occupancy grid:
 * voxel -> camera frame (double) -> rs2_project (float) -> cull outside [0,w]x[0,h] (:248-251) ->
 * integer pixel (:253, the int parameters of project_pixel_to_ray_end) -> rs2_deproject at depth 1
 * (Share_Data.hpp:719-726) -> world (double, stored as float like octomap::point3d) -> direction =
 * end - origin -> first hit within max_range (castRay, :258).  out[i] = cell index or -1.
 * Coordinates are unit-cube coordinates; origin = camera position (the reference snaps it to a voxel
 * centre of its octree, which has no counterpart here). */
void orc_precept(const orc_field* f, const float* voxels, int n, const double w2c[16], const double c2w[16],
                 const float intr[9], int width, int height, int model, float max_range, int32_t* out) {
  const float origin[3] = {(float)c2w[3], (float)c2w[7], (float)c2w[11]};
  for (int i = 0; i < n; i++) {
    out[i] = -1;
    double v[3];
    for (int r = 0; r < 3; r++)
      v[r] = w2c[r * 4] * (double)voxels[i * 3] + w2c[r * 4 + 1] * (double)voxels[i * 3 + 1] + w2c[r * 4 + 2] * (double)voxels[i * 3 + 2] + w2c[r * 4 + 3];
    const float point_3d[3] = {(float)v[0], (float)v[1], (float)v[2]};
    float pixel[2];
    orc_rs2_project(pixel, intr, model, point_3d);
    if (pixel[0] < 0 || pixel[0] > (float)width || pixel[1] < 0 || pixel[1] > (float)height) continue;
    const float ipx[2] = {(float)(int)pixel[0], (float)(int)pixel[1]};
    float pt[3];
    orc_rs2_deproject(pt, intr, model, ipx, 1.0f);
    float end[3], d[3];
    for (int r = 0; r < 3; r++) {
      end[r] = (float)(c2w[r * 4] * (double)pt[0] + c2w[r * 4 + 1] * (double)pt[1] + c2w[r * 4 + 2] * (double)pt[2] + c2w[r * 4 + 3]);
      d[r] = end[r] - origin[r];
    }
    const float n2 = fmaf(d[0], d[0], fmaf(d[1], d[1], d[2] * d[2]));
    const float inv = 1.0f / sqrtf(n2);
    for (int r = 0; r < 3; r++) d[r] = d[r] * inv;
    int cell[3];
    if (orc_first_hit(f, origin, d, max_range, cell)) {
      const int R = f->desc.occ_res;
      out[i] = cell[0] + R * (cell[1] + R * cell[2]);
    }
  }
}

/* first-hit cell per pixel for image rows [y0, y1): out[(y - y0) * w + x] = cell index or -1 */
void orc_first_hit_rows(const orc_field* f, const orc_camera* cam, int w, int y0, int y1, float max_range, int32_t* out) {
  const int R = f->desc.occ_res;
  for (int y = y0; y < y1; y++)
    for (int x = 0; x < w; x++) {
      float o[3], d[3];
      int cell[3];
      orc_raygen(cam, x, y, 0.5f, 0.5f, o, d);
      out[(size_t)(y - y0) * w + x] = orc_first_hit(f, o, d, max_range, cell) ? cell[0] + R * (cell[1] + R * cell[2]) : -1;
    }
}

/* ------------------------------------------------------------------ ground-truth splats */

/* shared with the device twin (prv_kernels.hip: splat_project): engine-frame point -> pixel centre + depth */
static int splat_project(const orc_camera* cam, float scale, const float off[3], const float* p, float* u, float* v,
                         float* z) {
  const float q[3] = {fmaf(p[0], scale, off[0]), fmaf(p[1], scale, off[1]), fmaf(p[2], scale, off[2])};
  const float e[3] = {q[1], q[2], q[0]}; /* axes cycled as nerf_to_ngp cycles camera positions */
  const float d[3] = {e[0] - cam->c2w[3], e[1] - cam->c2w[7], e[2] - cam->c2w[11]};
  float c[3];
  for (int a = 0; a < 3; a++) c[a] = fmaf(cam->c2w[a], d[0], fmaf(cam->c2w[4 + a], d[1], cam->c2w[8 + a] * d[2]));
  if (!(c[2] > 1e-6f)) return 0;
  float x = c[0] / c[2], y = c[1] / c[2];
  if (cam->lens[0] != 0.0f || cam->lens[1] != 0.0f || cam->lens[2] != 0.0f || cam->lens[3] != 0.0f) {
    float xd, yd;
    orc_lens_distort(cam->lens, x, y, &xd, &yd);
    x = xd;
    y = yd;
  }
  *u = fmaf(cam->fx, x, cam->cx);
  *v = fmaf(cam->fy, y, cam->cy);
  *z = c[2];
  return 1;
}

void orc_splat_points(const float* xyz, const uint8_t* rgb, size_t n, float scale, const float offset[3],
                      const orc_camera* cam, int w, int h, int point_size, int flip180, uint8_t* out) {
  uint64_t* zb = (uint64_t*)malloc(sizeof(uint64_t) * (size_t)w * h);
  for (size_t i = 0; i < (size_t)w * h; i++) zb[i] = ~0ull;
  for (size_t i = 0; i < n; i++) {
    float u, v, z;
    if (!splat_project(cam, scale, offset, xyz + 3 * i, &u, &v, &z)) continue;
    uint32_t zbits;
    memcpy(&zbits, &z, 4);
    const uint32_t col = (uint32_t)rgb[3 * i] | ((uint32_t)rgb[3 * i + 1] << 8) | ((uint32_t)rgb[3 * i + 2] << 16);
    const uint64_t key = ((uint64_t)zbits << 32) | col;
    const int x0 = (int)floorf(u - 0.5f * (float)point_size + 0.5f), y0 = (int)floorf(v - 0.5f * (float)point_size + 0.5f);
    for (int dy = 0; dy < point_size; dy++)
      for (int dx = 0; dx < point_size; dx++) {
        const int x = x0 + dx, y = y0 + dy;
        if (x < 0 || y < 0 || x >= w || y >= h) continue;
        if (key < zb[(size_t)y * w + x]) zb[(size_t)y * w + x] = key;
      }
  }
  for (int y = 0; y < h; y++)
    for (int x = 0; x < w; x++) {
      const uint64_t k = zb[(size_t)y * w + x];
      uint8_t px[4] = {255, 255, 255, 0}; /* white background, made transparent by convertToAlpha */
      if (k != ~0ull) {
        px[0] = (uint8_t)(k & 255);
        px[1] = (uint8_t)((k >> 8) & 255);
        px[2] = (uint8_t)((k >> 16) & 255);
        px[3] = (px[0] == 255 && px[1] == 255 && px[2] == 255) ? 0 : 255; /* a white POINT turns transparent too */
      }
      const size_t o = flip180 ? ((size_t)(h - 1 - y) * w + (size_t)(w - 1 - x)) : ((size_t)y * w + x);
      memcpy(out + 4 * o, px, 4);
    }
  free(zb);
}
