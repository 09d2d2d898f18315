/*
 * prv_train.c -- CPU ORACLE of the in-process training step (test infrastructure, see prv_oracle.h).
 *
 * Replaces, on the reference path: the `while testbed.frame()` training loop that run.py:185-208 drives
 * for `--n_steps 2500` (main.cpp:1668).  The optimiser lives inside pyngp (NVlabs/instant-ngp +
 * tiny-cuda-nn, unvendored, unpinned), so this is a restatement of the PUBLISHED algorithm
 * (Mueller et al. 2022, sections 4-5): random training rays, occupancy-skipped samples, L2 loss on
 * linear colours over a random background, straight-through backward through the fp16 roundings,
 * Adam (beta 0.9/0.99, eps 1e-15) with sparse updates of the hash table, periodic density-grid refresh.
 * PARITY UNPINNED against the reference (nothing in its tree pins a loss value or a weight).
 * What IS checked: the backward pass against central finite differences of this file's own forward
 * pass in `exact` mode (no fp16 rounding), and the HIP trainer against this file.
 *
 * Two sampling rules (orc_train_opts.step_mode): ORC_STEP_FIXED_S -- S uniform samples between the AABB hits
 * with one random offset per ray (rounds 1-5) -- and ORC_STEP_NGP -- the engine's own marcher for aabb_scale = 1,
 * what upstream trains with (run.py:188): fixed step sqrt(3)/1024, per-ray random start, every step
 * tested against the occupancy grid (round 6).  Deliberate simplifications against upstream, stated: the sample
 * budget of a step is kept by a simple integer rule on the ray count (upstream smooths its own estimate);
 * fp32 gradients and master weights (upstream:
 * fp16 gradients with loss scaling); the density grid is refreshed over ALL cells at their centres.
 */
#include "prv_oracle.h"
#include <math.h>
#include <stdlib.h>
#include <string.h>

#define TR_MAX_S ORC_NGP_MAX_STEPS /* 128 under ORC_STEP_FIXED_S, 1024 steps under ORC_STEP_NGP */

static uint64_t tr_mix64(uint64_t z) {
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
  return z ^ (z >> 31);
}
/* 24 random bits: value i of stream s under seed (same construction as the synthetic field's RNG) */
uint32_t orc_rng_u24(uint64_t seed, uint64_t stream, uint64_t i) {
  return (uint32_t)(tr_mix64(seed + (stream + 1) * 0xD1B54A32D192ED03ull + i * 0x9E3779B97F4A7C15ull) >> 40);
}

static const int kIn[5] = {32, 64, 32, 64, 64}, kOut[5] = {64, 16, 64, 64, 16};
static const int kOff[5] = {0, 2048, 3072, 5120, 9216};

struct orc_trainer {
  orc_train_opts o;
  int exact; /* 1: no fp16 rounding anywhere (differentiable forward for the finite-difference check) */
  orc_field* f; /* working fp16 parameters + occupancy (what inference would load) */
  size_t n_table; /* table scalars */
  float *tab_w, *tab_m, *tab_v; /* fp32 master + Adam moments, canonical layout */
  double* tab_g;
  float mlp_w[ORC_MLP_HALFS], mlp_m[ORC_MLP_HALFS], mlp_v[ORC_MLP_HALFS];
  double mlp_g[ORC_MLP_HALFS];
  float* ema; /* density EMA per occupancy cell */
  const orc_camera* cams;
  int n_img, w, h;
  const uint8_t* rgba8;
  uint32_t step; /* completed steps */
  uint32_t n_active; /* rays of the next step */
  uint64_t n_samples_last;
};

static inline float q16(const orc_trainer* t, float x) { return t->exact ? x : orc_h2f(orc_f2h(x)); }
static inline float clamp01f(float x) { return fminf(fmaxf(x, 0.0f), 1.0f); }

/* parameters as the forward pass sees them */
static inline float tab_val(const orc_trainer* t, size_t i) { return t->exact ? t->tab_w[i] : orc_h2f(t->f->table[i]); }
static inline float mlp_val(const orc_trainer* t, int i) { return t->exact ? t->mlp_w[i] : t->f->mlp_f[i]; }

typedef struct {
  float t;
  float feat[32];        /* encoder output */
  uint32_t cidx[16][8];  /* canonical entry index of each corner */
  float cw[16][8];       /* blend weight of each corner as used */
  float h1[64], od[16], in2[32], h2[64], h3[64], orr[16];
  float sigma, rgb[3], alpha, T_before;
} tr_sample;

static void tr_encode(const orc_trainer* t, const float p_in[3], tr_sample* s) {
  const orc_field* f = t->f;
  const int F = f->desc.n_features;
  float p[3] = {clamp01f(p_in[0]), clamp01f(p_in[1]), clamp01f(p_in[2])};
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
    double acc_d[4] = {0, 0, 0, 0};
    uint16_t acc_h[4] = {0, 0, 0, 0};
    uint16_t wh[3][2];
    for (int a = 0; a < 3; a++) {
      wh[a][0] = orc_f2h(1.0f - w[a]);
      wh[a][1] = orc_f2h(w[a]);
    }
    for (int c = 0; c < 8; c++) {
      uint32_t cc[3];
      for (int a = 0; a < 3; a++) {
        uint32_t v = c0[a] + ((c >> a) & 1u);
        cc[a] = v > L->res - 1 ? L->res - 1 : v;
      }
      uint32_t idx = L->hashed ? ((cc[0] ^ (cc[1] * 2654435761u) ^ (cc[2] * 805459861u)) & (L->size - 1u))
                               : (cc[0] + L->res * (cc[1] + L->res * cc[2]));
      const size_t e = ((size_t)L->offset + idx) * F;
      s->cidx[l][c] = L->offset + idx;
      if (t->exact) {
        const double wc = (double)((c & 1) ? w[0] : 1.0f - w[0]) * (double)(((c >> 1) & 1) ? w[1] : 1.0f - w[1]) *
                          (double)((c >> 2) ? w[2] : 1.0f - w[2]);
        s->cw[l][c] = (float)wc;
        for (int k = 0; k < F; k++) acc_d[k] += (double)s->cw[l][c] * (double)t->tab_w[e + k];
      } else { /* binary16 blend, bit for bit orc_encode */
        const uint16_t a01 = orc_d2h((double)orc_h2f(wh[0][c & 1]) * (double)orc_h2f(wh[1][(c >> 1) & 1]));
        const uint16_t wc = orc_d2h((double)orc_h2f(a01) * (double)orc_h2f(wh[2][c >> 2]));
        s->cw[l][c] = orc_h2f(wc);
        for (int k = 0; k < F; k++)
          acc_h[k] = orc_d2h((double)orc_h2f(wc) * (double)orc_h2f(f->table[e + k]) + (double)orc_h2f(acc_h[k]));
      }
    }
    for (int k = 0; k < F; k++) s->feat[l * F + k] = t->exact ? (float)acc_d[k] : orc_h2f(acc_h[k]);
  }
}

static void tr_layer(const orc_trainer* t, int li, const float* in, float* out) {
  double acc[64];
  const int ni = kIn[li], no = kOut[li];
  for (int o = 0; o < no; o++) acc[o] = 0.0;
  for (int k = 0; k < ni; k++) {
    const double x = (double)in[k];
    for (int o = 0; o < no; o++) acc[o] += x * (double)mlp_val(t, kOff[li] + k * no + o);
  }
  for (int o = 0; o < no; o++) out[o] = (float)acc[o];
}
static void tr_relu(const orc_trainer* t, float* v, int n) {
  for (int i = 0; i < n; i++) v[i] = q16(t, v[i] > 0.0f ? v[i] : 0.0f);
}

static void tr_forward(const orc_trainer* t, const float p[3], const float sh[16], tr_sample* s) {
  tr_encode(t, p, s);
  tr_layer(t, 0, s->feat, s->h1);
  tr_relu(t, s->h1, 64);
  tr_layer(t, 1, s->h1, s->od);
  s->sigma = expf(s->od[0] + t->f->desc.density_bias);
  for (int k = 0; k < 16; k++) s->in2[k] = q16(t, s->od[k]);
  for (int k = 0; k < 16; k++) s->in2[16 + k] = q16(t, sh[k]);
  tr_layer(t, 2, s->in2, s->h2);
  tr_relu(t, s->h2, 64);
  tr_layer(t, 3, s->h2, s->h3);
  tr_relu(t, s->h3, 64);
  tr_layer(t, 4, s->h3, s->orr);
  for (int k = 0; k < 3; k++) s->rgb[k] = 1.0f / (1.0f + expf(-s->orr[k]));
}

/* dL/d(in) = W dL/d(out) and dW += in (x) dL/d(out) for layer li */
static void tr_layer_backward(orc_trainer* t, int li, const float* in, const double* dout, double* din) {
  const int ni = kIn[li], no = kOut[li];
  for (int k = 0; k < ni; k++) {
    double a = 0.0;
    for (int o = 0; o < no; o++) {
      a += (double)mlp_val(t, kOff[li] + k * no + o) * dout[o];
      t->mlp_g[kOff[li] + k * no + o] += (double)in[k] * dout[o];
    }
    if (din) din[k] = a;
  }
}

static float srgb_to_linear(float c) { return c <= 0.04045f ? c / 12.92f : powf((c + 0.055f) / 1.055f, 2.4f); }

/* one training ray of step `step`: returns its loss term; grad != 0 also accumulates gradients */
static double tr_ray(orc_trainer* t, uint32_t step, uint32_t j, int grad, tr_sample* S) {
  const orc_train_opts* o = &t->o;
  const uint64_t st = (uint64_t)step * 8u;
  /* the batch rule: ray j is pixel j % P of patch j / P (P = patch_w * patch_h adjacent pixels of one image, rows
   * walked in snake order); image, patch origin and the jitter are drawn per PATCH, the background per ray.
   * P = 1: every ray its own pixel, as published. */
  const uint32_t pw = o->patch_w > 1 ? (uint32_t)o->patch_w : 1u, ph = o->patch_h > 1 ? (uint32_t)o->patch_h : 1u;
  const uint32_t P = pw * ph, q = j / P, r = j % P;
  const uint32_t ry = r / pw, rx = (ry & 1u) ? pw - 1u - r % pw : r % pw;
  const uint32_t img = (uint32_t)(((uint64_t)orc_rng_u24(o->seed, st + 0, q) * (uint64_t)t->n_img) >> 24);
  const uint32_t px = (uint32_t)(((uint64_t)orc_rng_u24(o->seed, st + 1, q) * (uint64_t)(t->w - (int)pw + 1)) >> 24) + rx;
  const uint32_t py = (uint32_t)(((uint64_t)orc_rng_u24(o->seed, st + 2, q) * (uint64_t)(t->h - (int)ph + 1)) >> 24) + ry;
  const float jitter = (float)orc_rng_u24(o->seed, st + 3, q) * (1.0f / 16777216.0f);
  float bg[3] = {0, 0, 0};
  if (o->random_bg)
    for (int k = 0; k < 3; k++) bg[k] = (float)orc_rng_u24(o->seed, st + 4 + k, j) * (1.0f / 16777216.0f);
  const uint8_t* gp = t->rgba8 + (((size_t)img * t->h + py) * t->w + px) * 4;
  const float ga = (float)gp[3] * (1.0f / 255.0f);
  float target[3];
  for (int k = 0; k < 3; k++) target[k] = fmaf(srgb_to_linear((float)gp[k] * (1.0f / 255.0f)), ga, (1.0f - ga) * bg[k]);

  float ro[3], rd[3], t0, t1;
  orc_raygen(&t->cams[img], (int)px, (int)py, 0.5f, 0.5f, ro, rd);
  int n = 0;
  float T = 1.0f, C[3] = {0, 0, 0}, dt = 0.0f;
  if (orc_ray_aabb(ro, rd, &t0, &t1)) {
    float sh[16];
    orc_sh4(rd, sh);
    const int ngp = o->step_mode == ORC_STEP_NGP;
    dt = ngp ? sqrtf(3.0f) / 1024.0f : (t1 - t0) / (float)o->n_samples;
    for (int i = 0; i < o->n_samples; i++) {
      const float tt = fmaf((float)i + jitter, dt, t0);
      if (ngp && !(tt < t1)) break; /* left the box (as march_ray, prv_oracle.c) */
      const float p[3] = {fmaf(tt, rd[0], ro[0]), fmaf(tt, rd[1], ro[1]), fmaf(tt, rd[2], ro[2])};
      if (!orc_occupied(t->f, p)) continue;
      tr_sample* s = &S[n++];
      s->t = tt;
      tr_forward(t, p, sh, s);
      s->alpha = 1.0f - expf(-(s->sigma * dt));
      s->T_before = T;
      const float wgt = s->alpha * T;
      for (int k = 0; k < 3; k++) C[k] = fmaf(wgt, s->rgb[k], C[k]);
      T = T * (1.0f - s->alpha);
      if (T < o->min_T) break;
    }
  }
  t->n_samples_last += (uint64_t)n;
  float pred[3];
  double loss = 0.0, dC[3];
  for (int k = 0; k < 3; k++) {
    pred[k] = fmaf(T, bg[k], C[k]);
    const double e = (double)pred[k] - (double)target[k];
    loss += e * e;
    dC[k] = 2.0 * e / (3.0 * (double)t->n_active);
  }
  loss /= 3.0 * (double)t->n_active;
  if (!grad) return loss;

  /* suffix[k] = sum_{j>i} w_j c_j + T_final bg: walk the samples back to front */
  double suffix[3] = {(double)T * bg[0], (double)T * bg[1], (double)T * bg[2]};
  for (int i = n - 1; i >= 0; i--) {
    tr_sample* s = &S[i];
    const double wgt = (double)s->alpha * (double)s->T_before;
    const double T_after = (double)s->T_before * (1.0 - (double)s->alpha);
    double d_orr[16], d_h3[64], d_h2[64], d_in2[32], d_od[16], d_h1[64], d_feat[32];
    memset(d_orr, 0, sizeof(d_orr));
    double d_sigma = 0.0;
    for (int k = 0; k < 3; k++) {
      d_sigma += dC[k] * (double)dt * (T_after * (double)s->rgb[k] - suffix[k]);
      d_orr[k] = dC[k] * wgt * (double)s->rgb[k] * (1.0 - (double)s->rgb[k]); /* sigmoid' */
      suffix[k] += wgt * (double)s->rgb[k];
    }
    tr_layer_backward(t, 4, s->h3, d_orr, d_h3);
    for (int k = 0; k < 64; k++)
      if (!(s->h3[k] > 0.0f)) d_h3[k] = 0.0;
    tr_layer_backward(t, 3, s->h2, d_h3, d_h2);
    for (int k = 0; k < 64; k++)
      if (!(s->h2[k] > 0.0f)) d_h2[k] = 0.0;
    tr_layer_backward(t, 2, s->in2, d_h2, d_in2);
    for (int k = 0; k < 16; k++) d_od[k] = d_in2[k]; /* straight through the fp16 rounding of od */
    d_od[0] += d_sigma * (double)s->sigma;           /* sigma = exp(od0 + bias) */
    tr_layer_backward(t, 1, s->h1, d_od, d_h1);
    for (int k = 0; k < 64; k++)
      if (!(s->h1[k] > 0.0f)) d_h1[k] = 0.0;
    tr_layer_backward(t, 0, s->feat, d_h1, d_feat);
    const int F = t->f->desc.n_features;
    for (int l = 0; l < t->f->desc.n_levels; l++)
      for (int c = 0; c < 8; c++)
        for (int k = 0; k < F; k++) t->tab_g[(size_t)s->cidx[l][c] * F + k] += (double)s->cw[l][c] * d_feat[l * F + k];
  }
  return loss;
}

static void tr_refresh_fp16(orc_trainer* t) {
  for (size_t i = 0; i < t->n_table; i++) t->f->table[i] = orc_f2h(t->tab_w[i]);
  for (int i = 0; i < ORC_MLP_HALFS; i++) {
    t->f->mlp[i] = orc_f2h(t->mlp_w[i]);
    t->f->mlp_f[i] = orc_h2f(t->f->mlp[i]);
  }
}

orc_trainer* orc_train_create(const orc_field* init, const orc_train_opts* o, const orc_camera* cams, int n_img, int w,
                              int h, const uint8_t* rgba8, int exact) {
  if (!init || !o || o->n_samples < 1 || o->n_samples > (o->step_mode == ORC_STEP_NGP ? ORC_NGP_MAX_STEPS : 128) || o->n_rays < 1 || n_img < 1) return NULL;
  if (o->step_mode != ORC_STEP_FIXED_S && o->step_mode != ORC_STEP_NGP) return NULL;
  if (o->patch_w > w || o->patch_h > h || o->patch_w < 0 || o->patch_h < 0) return NULL;
  orc_trainer* t = (orc_trainer*)calloc(1, sizeof(orc_trainer));
  if (!t) return NULL;
  t->o = *o;
  t->exact = exact;
  t->f = orc_field_from_params(&init->desc, init->table, init->mlp, init->occ);
  t->n_table = (size_t)init->total_entries * init->desc.n_features;
  t->tab_w = (float*)malloc(t->n_table * sizeof(float));
  t->tab_m = (float*)calloc(t->n_table, sizeof(float));
  t->tab_v = (float*)calloc(t->n_table, sizeof(float));
  t->tab_g = (double*)calloc(t->n_table, sizeof(double));
  const size_t R = (size_t)init->desc.occ_res;
  t->ema = (float*)calloc(R * R * R, sizeof(float));
  for (size_t i = 0; i < t->n_table; i++) t->tab_w[i] = orc_h2f(init->table[i]);
  for (int i = 0; i < ORC_MLP_HALFS; i++) t->mlp_w[i] = orc_h2f(init->mlp[i]);
  t->n_active = (uint32_t)o->n_rays;
  if (o->target_samples > 0) {
    uint32_t a = (uint32_t)o->target_samples / (uint32_t)o->n_samples;
    if (a < 1u) a = 1u;
    if (a < t->n_active) t->n_active = a;
  }
  t->cams = cams;
  t->n_img = n_img;
  t->w = w;
  t->h = h;
  t->rgba8 = rgba8;
  return t;
}

void orc_train_free(orc_trainer* t) {
  if (!t) return;
  orc_field_free(t->f);
  free(t->tab_w);
  free(t->tab_m);
  free(t->tab_v);
  free(t->tab_g);
  free(t->ema);
  free(t);
}

const orc_field* orc_train_field(const orc_trainer* t) { return t->f; }
uint32_t orc_train_steps_done(const orc_trainer* t) { return t->step; }
uint64_t orc_train_samples_last(const orc_trainer* t) { return t->n_samples_last; }
uint32_t orc_train_active_rays(const orc_trainer* t) { return t->n_active; }

/* loss of the NEXT step's ray batch under the current parameters, no gradient, no update */
double orc_train_loss_only(orc_trainer* t) {
  tr_sample* S = (tr_sample*)malloc(sizeof(tr_sample) * TR_MAX_S);
  double loss = 0.0;
  t->n_samples_last = 0;
  for (uint32_t j = 0; j < t->n_active; j++) loss += tr_ray(t, t->step, j, 0, S);
  free(S);
  return loss;
}

/* gradients of the NEXT step's batch (no update); returns the loss */
double orc_train_gradients(orc_trainer* t, double* table_grad, double* mlp_grad) {
  tr_sample* S = (tr_sample*)malloc(sizeof(tr_sample) * TR_MAX_S);
  memset(t->tab_g, 0, t->n_table * sizeof(double));
  memset(t->mlp_g, 0, sizeof(t->mlp_g));
  double loss = 0.0;
  t->n_samples_last = 0;
  for (uint32_t j = 0; j < t->n_active; j++) loss += tr_ray(t, t->step, j, 1, S);
  free(S);
  if (table_grad) memcpy(table_grad, t->tab_g, t->n_table * sizeof(double));
  if (mlp_grad) memcpy(mlp_grad, t->mlp_g, sizeof(t->mlp_g));
  return loss;
}

static void adam(const orc_train_opts* o, float lr_t, float g, float* w, float* m, float* v) {
  *m = fmaf(o->beta1, *m, (1.0f - o->beta1) * g);
  *v = fmaf(o->beta2, *v, ((1.0f - o->beta2) * g) * g);
  *w = *w - (lr_t * *m) / (sqrtf(*v) + o->eps);
}

void orc_train_refresh_occupancy(orc_trainer* t) {
  const int R = t->f->desc.occ_res;
  const float invR = 1.0f / (float)R;
  tr_sample s;
  float sh[16] = {0};
  size_t nw = ((size_t)R * R * R + 31) / 32;
  memset(t->f->occ, 0, nw * sizeof(uint32_t));
  for (int z = 0; z < R; z++)
    for (int y = 0; y < R; y++)
      for (int x = 0; x < R; x++) {
        const float p[3] = {((float)x + 0.5f) * invR, ((float)y + 0.5f) * invR, ((float)z + 0.5f) * invR};
        tr_encode(t, p, &s);
        tr_layer(t, 0, s.feat, s.h1);
        tr_relu(t, s.h1, 64);
        tr_layer(t, 1, s.h1, s.od);
        (void)sh;
        const float sigma = expf(s.od[0] + t->f->desc.density_bias);
        const size_t c = (size_t)x + (size_t)R * ((size_t)y + (size_t)R * (size_t)z);
        t->ema[c] = fmaxf(t->ema[c] * t->o.occ_decay, sigma);
        if (t->ema[c] > t->o.occ_sigma_thresh) t->f->occ[c >> 5] |= 1u << (c & 31);
      }
}

/* one optimiser step; returns the loss of the batch under the parameters BEFORE the update */
double orc_train_step(orc_trainer* t) {
  const double loss = orc_train_gradients(t, NULL, NULL);
  const orc_train_opts* o = &t->o;
  const uint32_t n = t->step + 1;
  const float lr_t = (float)((double)o->lr * sqrt(1.0 - pow((double)o->beta2, (double)n)) / (1.0 - pow((double)o->beta1, (double)n)));
  for (size_t i = 0; i < t->n_table; i++) {
    const float g = (float)t->tab_g[i];
    if (g == 0.0f) continue; /* sparse update: untouched entries keep their moments */
    adam(o, lr_t, g, &t->tab_w[i], &t->tab_m[i], &t->tab_v[i]);
  }
  for (int i = 0; i < ORC_MLP_HALFS; i++) {
    const float g = fmaf(o->l2_reg, t->mlp_w[i], (float)t->mlp_g[i]);
    adam(o, lr_t, g, &t->mlp_w[i], &t->mlp_m[i], &t->mlp_v[i]);
  }
  if (!t->exact) tr_refresh_fp16(t);
  t->step = n;
  if (o->target_samples > 0) { /* the sample budget: integer arithmetic, mirrored by adam_mlp_kernel */
    const uint64_t used = t->n_samples_last ? t->n_samples_last : 1u;
    uint64_t a = (uint64_t)o->target_samples * (uint64_t)t->n_active / used;
    const uint64_t lo = t->n_active / 2u, hi = (uint64_t)t->n_active * 2u;
    if (a < lo) a = lo;
    if (a > hi) a = hi;
    if (a < 1u) a = 1u;
    if (a > (uint64_t)o->n_rays) a = (uint64_t)o->n_rays;
    t->n_active = (uint32_t)a;
  }
  if (o->occ_every > 0 && n % (uint32_t)o->occ_every == 0) orc_train_refresh_occupancy(t);
  return loss;
}

/* master weights, for tests (finite differences perturb them in exact mode) */
float* orc_train_master_table(orc_trainer* t) { return t->tab_w; }
float* orc_train_master_mlp(orc_trainer* t) { return t->mlp_w; }
size_t orc_train_table_size(const orc_trainer* t) { return t->n_table; }
