// prv_kernels.hpp -- kernel parameter blocks and host-callable launchers.
#pragma once
#include "../../include/prv.h"
#include "prv_device.hpp"

namespace prv {

// one queue record = 96 bytes = 6 x uint4:
//   {o.x o.y o.z t0} {d.x d.y d.z dt} {mask0..3} {pixel, chunk info, 0, 0} {SH coeffs 0..7 fp16} {SH 8..15 fp16}
// The mask is one 128-step CHUNK of the ray's live-sample mask.  PRV_STEP_FIXED_S: the only one (chunk info = 1 << 16).
// PRV_STEP_NGP (up to 1024 steps): the chunk that starts at the ray's first non-empty 32-step word; chunk info = first
// step of that chunk | number of chunks << 16, and chunks 1..n-1 sit in the record's slot of the extension buffer
// (kExtChunks x uint4 per queue slot, chunk-major: chunk j of slot s at [(j - 1) * n_slots + s], so that the lanes of a
// march wave -- consecutive slots -- store a chunk as one contiguous block), read by the render kernel only when a ray gets that far.
constexpr size_t kRecordBytes = 96;
constexpr int kRecordWords = 6;   // uint4 per record
constexpr uint32_t kClaim = 64;   // records a wave claims per atomic on the queue head
constexpr int kMaxSamples = 128;  // PRV_STEP_FIXED_S: the live-sample mask is one 128-bit chunk
constexpr int kNgpMaxSteps = PRV_NGP_MAX_STEPS; // PRV_STEP_NGP: dt = sqrt(3)/1024, the cube's diagonal
constexpr int kExtChunks = kNgpMaxSteps / 128 - 1;
constexpr size_t kExtBytes = (size_t)kExtChunks * 16;

struct MarchParams {
  FieldDev field;
  const CamDev* cams;
  const int* view_ids; // n_views indices into cams
  int W, H, S, spp_k;
  uint32_t tiles_x, tiles_y;
  int tile_w_log2, tile_h_log2; // pixel tile of one 256-thread block
  int spp_inner_log2;           // > 0: sub-samples on adjacent lanes (spp = 2^n), 0: on grid.z
  int live_grid;                // 1: grid.x = the largest cull rectangle of the batch in tiles (CamDev::cull); pixels outside a
                                //    view's rectangle are NOT written (the caller consumes the image through the rectangles)
  uint32_t live_tiles_max;
  int step_mode; // PRV_STEP_FIXED_S | PRV_STEP_NGP
  void* queue;
  uint4* queue_ext;      // PRV_STEP_NGP: kExtChunks mask chunks per queue slot, chunk-major (n_slots = n_seg * seg_cap)
  unsigned long long* stat; // statistics block: [8 (1 + s)] += live samples (the march count), 8 shards a cache line apart
  uint32_t* queue_count; // n_seg counters, 64 bytes apart: records appended to region s of the queue
  int n_seg;             // the queue is n_seg regions of seg_cap records; a wave appends to the region of its first live ray's octant
  uint32_t seg_cap;      // (spatial_regions; 0: linear block id % n_seg, rounds 1-5), or to the next region that has room
  int spatial_regions;
  int n_sub;             // spatial_regions: the n_seg regions are n_seg / n_sub octant regions of n_sub sub-regions each (a counter per sub-region)
  float* out_f32; // n_views*H*W*4
  uint32_t* out_u8; // optional, n_views*H*W
  float inv_spp;
  int last_pass;
  float bg[4];
};

// the ensemble's march in one launch (march_multi_kernel): the common fields carry MarchParams' names
struct MarchMember {
  void* queue;
  uint4* queue_ext;
  uint32_t* queue_count; // this member's n_seg counters, 64 bytes apart
  float* out_f32;        // where this member's dead rays are written (its own staging / image)
  uint32_t* out_u8;
};
struct MarchMultiParams {
  const uint8_t* occ_bytes;        // occ_res^3 bytes, x fastest: bit e = member e's occupancy bit of the cell
  const uint8_t* occ_coarse_bytes; // (occ_res/4)^3 bytes of the members' dilated coarse grids, or null
  int occ_res;
  float occ_lo[3], occ_hi[3]; // the union of the members' occupied boxes
  const CamDev* cams;         // cull rectangles against the union box
  const int* view_ids;
  int W, H, spp_k;
  uint32_t tiles_x, tiles_y;
  int tile_w_log2, tile_h_log2, spp_inner_log2;
  unsigned long long* stat;
  int n_seg;
  uint32_t seg_cap;
  int spatial_regions, n_sub; // MarchParams::spatial_regions, n_sub
  float inv_spp;
  int last_pass;
  float bg[4];
  int n_members;
  MarchMember mem[PRV_MAX_MODELS];
};
struct OccInterleaveParams {
  const uint32_t* bits[PRV_MAX_MODELS];
  int n_members;
  uint32_t n_cells;
  uint8_t* out;
};
bool march_multi_supported(int n_members); // a compiled instance exists (2 and 5: the paper's ensembles)
hipError_t launch_march_multi(const MarchMultiParams& P, int n_views, int n_spp, hipStream_t s);
hipError_t launch_occ_interleave(const OccInterleaveParams& P, hipStream_t s);

struct RenderParams {
  FieldDev field;
  const void* queue;
  const uint4* queue_ext; // PRV_STEP_NGP: the mask chunks behind the record's own (see kRecordBytes)
  int step_mode;
  const uint32_t* queue_count; // n_segments counters, 64 bytes apart: records in region s
  uint32_t* queue_head; // n_segments heads, 64 bytes apart (region-relative record counts)
  int n_segments;
  int n_sub; // an XCD starts at sub-region 0 of ITS octant region: segment (XCC id % (n_segments / n_sub)) * n_sub
  uint32_t seg_cap;     // records per region: region s = [s * seg_cap, s * seg_cap + queue_count[16 s])
  unsigned long long* stat_evaluated;
  float* out_f32;
  uint32_t* out_u8;
  float min_T;
  float inv_spp;
  int spp_k;
  int last_pass;
  int merge_max; // render_queue64: a group down to <= this many rays hands them to the other group's idle slots (0 = never)
  int pool_on;   // render_queue64: ... or, failing that, to the block's LDS tail pool (any wave's idle slots adopt them)
  int cell_cache; // render_queue64: the instance whose lanes keep their last cell's corner entries per hashed level (incoherent gathers)
  float bg[4];
};

struct EnsembleParams {
  const uint32_t* imgs[PRV_MAX_MODELS];
  int E;
  int view0; // first view of this batch (the addend buffer holds one batch)
  size_t pixels_per_view;
  double* partial; // addends: [view - view0][pixel][k]
};

struct PsnrParams {
  const float* rgba;
  const float* gt;
  const CamDev* cams; // optional: view v's cull rectangle bounds what was rendered (see score_psnr_kernel)
  int W;
  size_t pixels_per_view;
  float bg[4];
  double* partial;
};

struct RepackLevel { // canonical -> physical copy of one level (entries, not bytes)
  uint32_t canon_off, phys_off, n, res, sx, hashed;
};
hipError_t launch_repack_level(const uint16_t* canon, uint16_t* phys, const RepackLevel& L, int F, hipStream_t s);

hipError_t launch_march(const MarchParams& P, int n_views, int n_spp, hipStream_t s);
hipError_t launch_spp_reduce(const float* stage, size_t n_pixels, int spp, const float bg[4], float* out, uint32_t* out_u8,
                             hipStream_t s);
hipError_t launch_render(const RenderParams& P, int n_blocks, hipStream_t s);
int render_instance_dense_levels(const FieldDev& fd); // NDENSE of the render_queue64_kernel<F, NDENSE> instance launch_render picks
struct PreceptPose {
  double w2c[16], c2w[16];
};
hipError_t launch_precept(const FieldDev& fd, const float* voxels, int n, const PreceptPose& pose, const Rs2Intr& in,
                          float max_range, int32_t* out, hipStream_t s);
hipError_t launch_first_hit(const FieldDev& fd, const CamDev* cams, int n_views, int W, int H, float max_range,
                            int32_t* out, hipStream_t s);
hipError_t launch_splat_points(const float* xyz, const uint8_t* rgb, size_t n, float scale, const float off[3],
                               const CamDev* cams, int n_views, int W, int H, int point_size, int flip180,
                               unsigned long long* zbuf, uint32_t* out, hipStream_t s);
hipError_t launch_quantize(const float* in, size_t n, const float bg[4], uint8_t* out, hipStream_t s);
hipError_t launch_score_ensemble(const EnsembleParams& P, int method, int n_views, int n_blocks, prv_score_record* rec,
                                 hipStream_t s);
hipError_t launch_score_psnr(const PsnrParams& P, int n_views, int n_blocks, hipStream_t s);
hipError_t launch_ssim(const float* img, const float* gt, int n_views, int W, int H, const float bg[4], float* la,
                       float* lb, double* partial, int n_blocks, double* out, hipStream_t s);
hipError_t launch_score_finalize(const double* partial, int n_views, int n_blocks, int method,
                                 size_t pixels_per_view, double coverage_weight, prv_score_record* rec, hipStream_t s);
hipError_t launch_synth_table(uint16_t* table, size_t n, uint64_t seed, float amp, hipStream_t s);
hipError_t launch_debug_raygen(const CamDev& cam, int W, int H, int spp_k, float* o, float* d, float* t,
                               hipStream_t s);
hipError_t launch_debug_field(const FieldDev& fd, const float* pos, const float* dir, int n, uint16_t* feat,
                              float* out36, int32_t* occ, hipStream_t s);

} // namespace prv
