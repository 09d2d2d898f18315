// prv_train.hpp -- parameter blocks and launchers of the training step (prv_train.hip)
#pragma once
#include "../../include/prv.h"
#include "prv_device.hpp"

namespace prv {

constexpr int kMaxTrainSamples = 128;          // PRV_STEP_FIXED_S
constexpr int kMaxTrainSteps = PRV_NGP_MAX_STEPS; // PRV_STEP_NGP
constexpr uint32_t kNgpListCap = 1u << 24;       // samples a step may list under PRV_STEP_NGP (prv.h)

struct LevelCanon { // canonical (ABI) table layout of one level
  float scale;
  uint32_t res, offset, size, hashed;
};

struct TrainRay {
  float o[3], d[3], t0, dt, jitter;
  float target[3], bg[3];
  uint32_t offset, n_live, n_used;
  uint32_t pad[2];
};

// the per-step scalars live in device memory so that one captured HIP graph replays every step
struct TrainState {
  uint32_t step;  // completed steps
  uint32_t step0; // first step of the current prv_train_steps call (loss slot = step - step0)
  float lr_t;     // bias-corrected learning rate of the step in flight
  uint32_t n_active; // rays of the step in flight (<= n_rays; adapts to the sample budget)
  float* losses;  // where the call's per-step losses go
  float lr_cur;   // lr_t again, copied by the table's Adam launch: what the step's LAST kernel (the MLP's Adam pass) reads while its
                  // end_step already writes the next step's lr_t
  uint32_t overflow; // sticky: a ray batch did not fit the sample list (PRV_STEP_NGP's 2^24 cap); every later kernel sees an empty batch
};
static_assert(sizeof(TrainState) == 32, "scal: TrainState sits at [32, 64)");

struct TrainRaysParams {
  const CamDev* cams;    // n_img dataset cameras at (W, H)
  const uint8_t* images; // n_img * H * W * 4 straight-alpha sRGB bytes
  const uint32_t* occ;
  int n_img, W, H, S, n_rays, occ_res, random_bg;
  TrainState* state; // step number = state->step
  uint64_t seed;
  TrainRay* rays;
  uint2* samples; // (ray, sample index) of every live sample, grouped by ray (patch mode: by patch, depth step by depth step)
  uint32_t* sample_count; // two words: [step & 1]
  // patch mode (patch_w * patch_h = P > 1, <= 16): ray j = pixel j % P of patch j / P; slot_of[j * S + k] = the list
  // position of ray j's k-th live sample (a ray's samples are not contiguous in the list any more)
  // next = 1: the batch of the step AFTER the one in flight (state->step + 1), launched beside the table's Adam pass of the
  // step in flight (adam_table_kernel's extra blocks): its ray count is what end_step will set -- the same integer rule
  // on the same inputs (the backward launch's used-sample slices in loss_part) -- and its samples go to the OTHER counter
  // (sample_count[step & 1]: a step's live-sample count lives in the word of its parity)
  int next, target_samples;
  const double* loss_part;
  int patch_w, patch_h;
  int patch_ray_jitter; // dev only (PRV_TRAIN_PATCH_JITTER=ray): every ray of a patch its own jitter (the oracle has no such mode)
  uint32_t* slot_of;
  int step_mode;        // PRV_STEP_FIXED_S | PRV_STEP_NGP (then S = the most steps of a ray, <= 1024)
  uint32_t sample_cap;  // capacity of `samples`; a batch that does not fit sets state->overflow
  // deterministic = 1 (tests): the blocks append in block order -- block b waits for its ticket (scal word kOrderWord +
  // (step & 1), zeroed with the sample counters; blocks are dispatched in index order), so the sample list, the tiles and
  // every sum over them are the same run after run
  int deterministic;
};

struct TrainTileParams {
  const uint16_t* table; // canonical fp16
  const float* mlp;      // canonical [in][out], the fp16 working weights widened to f32
  LevelCanon levels[16];
  int n_levels, n_features;
  const TrainRay* rays;
  const uint2* samples;
  const uint32_t* sample_count; // two words: the step's count is [state->step & 1]
  const TrainState* state;
  float4* logits;        // forward: {density logit, r, g, b logits} per sample
  const float4* seeds;   // backward: {dL/d od0 via sigma, dL/d rgb logits}
  float* table_grad;     // canonical, f32
  float* mlp_grad;       // canonical, f32
  float* mlp_grad_partial; // n_blocks slots of PRV_MLP_HALFS floats (backward)
  unsigned long long* stamps; // dev only (PRV_TRAIN_ABLATE & 16)
  // activations of the forward pass, kept for the backward pass (train_forward_fast_kernel writes, the backward tile
  // kernel reads; NULL or a tile beyond act_cap: the backward pass recomputes them).  kActTileBytes per 32-sample tile:
  // [tile][slot 0..15][lane half][sample] x 8 halfs -- the B fragments the forward lanes hold, see act_row() -- then the
  // tile's 32 sample positions (float4 each)
  uint4* act;
  uint32_t act_cap; // samples covered by `act` (a multiple of 32)
  uint32_t sample_cap; // the most samples a step can list (n_rays x n_samples): act_cap >= sample_cap = no tile is ever recomputed
  // set by launch_train_tiles: the tiles [tile_begin, min(tile_limit, all)) are this launch's, block b owns slot
  // slot_base + b of mlp_grad_partial (the backward pass is two launches when activations are kept: the tiles the
  // buffer covers, and -- nearly always none -- the tiles beyond it, which recompute their forward pass)
  uint32_t tile_begin, tile_limit;
  int slot_base;
  // backward-pass A fragments of the fp16 weights, rebuilt with the forward ones after every optimiser step:
  // kBwdFrags x 64 lanes x 8 halfs; NULL: the LDS / f32-MFMA chain
  const uint4* bwd_frags;
  // the step's loss rides on the backward launch (no kernel of its own in every member's chain): block gridDim.x - 1 - s of
  // the launch that starts at tile 0 sums slice s (1024 rays) of the rays' loss terms and used-sample counts in a fixed
  // order before its own tiles -> loss_part[2 s], [2 s + 1]; the slices are added up by the step's last kernel (end_step)
  // or by launch_train_loss_finish (ray_loss == NULL: nobody does)
  const float* ray_loss;
  const uint32_t* ray_used;
  int n_rays;
  double* loss_part;
  // deterministic = 1 (tests): the table gradient is summed in 64-bit fixed point (table_grad_q, one word per scalar,
  // value * 2^kGradQBits; integer adds commute) and converted by the table's Adam launch; NULL: f32 atomics into table_grad
  long long* table_grad_q;
  // one byte per 32-sample tile, set by the compositing kernel where a tile holds a used sample, cleared by the backward block
  // that walks the tile: the backward pass skips the tiles behind the rays' terminations (NULL: every tile is walked)
  uint8_t* tile_live;
  // (entry, sum) pairs a scattering thread keeps while it walks a tile: 1 -- samples along one ray come back to an entry only on
  // consecutive steps -- or, > 1, the kernel's four for patch batches (a depth step's rays alternate between the cells the patch straddles)
  int scatter_ways;
};
constexpr int kGradQBits = 40;
constexpr size_t kActTileBytes = (16 * 64 + 32) * 16; // kept activations of a 32-sample tile: 16 slots x 64 lanes x 16 B, then 32 positions
constexpr int kBwdFrags = 20; // R3: 2 row tiles | R2: 2 x 4 k-steps | R1: 4 k-steps | D2: 2 row tiles | D1: 4 k-steps
// slots of mlp_grad_partial for a backward pass launched with n_blocks blocks (the stage of the reduction sits behind them)
inline int train_dw_slots(int n_blocks) { return n_blocks + n_blocks / 2; } // first launch + the tail launch (launch_train_tiles)

struct TrainCompositeParams {
  TrainRay* rays;
  const TrainState* state; // n_active rays take part
  int n_rays;
  const float4* logits;
  float4* seeds;
  float density_bias, min_T;
  float* ray_loss;
  uint32_t* ray_used;
  const uint32_t* slot_of; // patch mode: list position of (ray, k-th live sample), row stride S; NULL: offset + k
  int S;
  uint8_t* tile_live; // TrainTileParams::tile_live
};

struct AdamParams {
  TrainState* state; // lr_t read from it; adam_mlp_kernel advances state->step and adapts state->n_active
  float beta1, beta2, eps;
  unsigned long long* used; // samples composited by the step (sample budget rule); written by end_step from loss_part
  int target_samples, n_rays;
  const double* loss_part;  // the backward launch's slice sums of the rays' loss / used counts (NULL: `used` and the loss are final already)
};
// the slices' sums -> state->losses[step - step0], *used (the parity hook prv_train_gradients: no optimiser kernel follows)
hipError_t launch_train_loss_finish(const double* loss_part, int n_rays, TrainState* state, unsigned long long* used, hipStream_t s);

struct DensityParams {
  const uint16_t* table;
  const float* mlp;
  LevelCanon levels[16];
  int occ_res;
  float density_bias, decay, thresh;
  float* ema;
  uint32_t* occ;
};

size_t train_tile_lds_bytes(bool fwd, int mode);
hipError_t train_prepare_kernels();
hipError_t launch_train_rays(const TrainRaysParams& P, hipStream_t s);
// backward: finish_reduce = false leaves the second stage of the dW reduction to adam_mlp_kernel (stage = slots + n_blocks)
// n_slots_out != NULL: NO reduction launch at all -- the count of written slots is returned and the table's Adam launch does the
// first stage beside its own work (launch_adam_table's dw_* arguments)
hipError_t launch_train_tiles(const TrainTileParams& P, bool forward, int n_blocks, hipStream_t s, bool finish_reduce = true, int* n_slots_out = nullptr);
hipError_t launch_prepack_frags(const uint16_t* mlp, int n_features, uint16_t* frags, const AdamParams* end_of_step,
                                uint32_t* sample_count, float lr, hipStream_t s);
hipError_t launch_train_forward_fast(const TrainTileParams& P, const half8* frags, int n_blocks, hipStream_t s);
hipError_t launch_train_composite(const TrainCompositeParams& P, hipStream_t s);
hipError_t launch_train_begin(TrainState* state, uint32_t* sample_count, float lr, float beta1, float beta2, hipStream_t s);
// wmv: one {w[4], m[4], v[4]} record (48 B) per group of four table scalars, ceil(n / 4) records
// next_rays != NULL (single-pixel batches only: the patch kernel has another block size): the NEXT step's ray batch as further
// extra blocks of the same launch (TrainRaysParams::next = 1)
hipError_t launch_adam_table(const AdamParams& P, size_t n, float* grad, float* wmv, uint16_t* w16, hipStream_t s, const float* dw_partial = nullptr,
                             int dw_slots = 0, float* dw_stage = nullptr, const TrainRaysParams* next_rays = nullptr, long long* grad_q = nullptr);
hipError_t launch_grad_q_to_f32(long long* q, size_t n, float* out, hipStream_t s); // deterministic trainers: fixed-point gradient -> f32, cleared
hipError_t launch_widen_table(const uint16_t* in, size_t n, float* wmv, hipStream_t s);
hipError_t launch_narrow_table(const float* wmv, size_t n, float* out, hipStream_t s); // the records' w parts, contiguous
// frags != NULL: the step's last kernel -- every thread also puts its fresh fp16 weight into its place of the forward and of the
// backward MFMA fragments (frag_pos: launch_frag_positions) and thread 0 closes the step (end_step with sample_count / lr)
hipError_t launch_adam_mlp(const AdamParams& P, float l2_reg, float* grad, float* w, float* m, float* v, uint16_t* w16,
                           float* w16_as_f32, const float* stage, int end_of_step, hipStream_t s, uint16_t* frags = nullptr,
                           const int* frag_pos = nullptr, uint32_t* sample_count = nullptr, float lr = 0.f);
// frag_pos[w] / frag_pos[PRV_MLP_HALFS + w]: where canonical weight w sits in the forward / backward fragments (each map is a bijection)
hipError_t launch_frag_positions(int n_features, int* frag_pos, hipStream_t s);
hipError_t launch_widen(const uint16_t* in, size_t n, float* out, hipStream_t s);
hipError_t launch_density_refresh(const DensityParams& P, int n_features, hipStream_t s);
hipError_t launch_density_refresh_fast(const DensityParams& P, int n_features, const half8* frags, int n_blocks, hipStream_t s);

} // namespace prv
