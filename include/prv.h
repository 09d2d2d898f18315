/*
 * prv.h -- C ABI of the MI355X-native NeRF render + candidate-view scoring path.
 *
 * This is the drop-in boundary for the hot path of psc0628/NeRF-PRV.  In the
 * reference that boundary is a process + filesystem handshake:
 *   PRV_simulation/main.cpp:1658-1715  NBV_Net_Labeler::train_by_instantNGP writes
 *       interact/run_with_c++.py + ready_c++.txt and polls ready_py.txt;
 *   Instantngp_scripts/train_server.py:7-14 runs the script;
 *   Instantngp_scripts/run.py:284-309 renders every frame of --screenshot_transforms
 *       through pyngp.Testbed.render and writes PNGs;
 *   PRV_simulation/main.cpp:2045-2096 / 2105-2160 reads the PNGs back and scores.
 * Each entry point below names the reference interface it replaces.  Everything is
 * plain C: opaque handles, pointers and sizes, int error codes; nothing throws.
 *
 * Memory spaces: "host" pointers are ordinary memory; "dev" pointers are HIP device
 * memory on the context's GPU (from prv_malloc, or any other HIP allocation such as
 * a torch tensor's data_ptr).  All work is enqueued on the context's stream
 * (prv_set_stream) and functions that return results to host memory synchronise it.
 *
 * The library REQUIRES a gfx950 device: there is no CPU fallback.  prv_create fails
 * with PRV_E_NODEVICE when no GPU is visible.
 */
#ifndef PRV_H
#define PRV_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define PRV_ABI_VERSION 5 /* 2: prv_field_desc.per_level_scale; 3: prv_render_opts.step_mode, prv_stats.samples_live; 4: prv_train_opts.patch_w / patch_h; 5: prv_train_opts.step_mode / deterministic */

/* error codes (0 = ok, < 0 = error; message via prv_last_error) */
#define PRV_OK 0
#define PRV_E_INVALID (-1)  /* bad argument */
#define PRV_E_HIP (-2)      /* HIP runtime error */
#define PRV_E_IO (-3)       /* file / parse error */
#define PRV_E_NODEVICE (-4) /* no usable GPU */
#define PRV_E_STATE (-5)    /* e.g. model slot empty */
#define PRV_E_INTERNAL (-6) /* host allocation failure / unexpected internal error: no exception crosses the ABI */

/* scoring methods; 2 and 3 are the reference's method_of_IG values (Share_Data.hpp:198-202) */
#define PRV_SCORE_ENSEMBLE_RGB 2         /* main.cpp:2039-2097 */
#define PRV_SCORE_ENSEMBLE_RGB_DENSITY 3 /* main.cpp:2099-2161 */
#define PRV_SCORE_PSNR_COVERAGE 5        /* -PSNR (run.py:257-263, vs supplied images) + w * mean((1 - alpha)^2), the
                                            density term of main.cpp:2148; w = prv_set_coverage_weight */
#define PRV_COVERAGE_WEIGHT_DEFAULT 1.0  /* the reference adds its density term with unit weight too (main.cpp:2147-2148) */

#define PRV_MAX_MODELS 8  /* members of one ensemble (one scoring call) */
#define PRV_MAX_SLOTS 64  /* model slots of a context: several objects' ensembles side by side (prv_planner shard: members) */
#define PRV_MLP_HALFS 10240 /* 32*64 + 64*16 + 32*64 + 64*64 + 64*16, canonical [in][out] */

typedef struct prv_ctx prv_ctx;
typedef struct prv_camset prv_camset;

/* The NeRF field: multiresolution hash grid (L levels x F fp16 features, L*F = 32),
 * density MLP 32->64->16, SH-4 direction encoding, colour MLP 32->64->64->16, and an
 * occupancy bitfield.  Stands in for the instant-ngp network the reference trains
 * through run.py:185-208. */
typedef struct prv_field_desc {
  int32_t n_levels;
  int32_t n_features;
  int32_t log2_hashmap;
  int32_t base_res;
  int32_t finest_res;
  int32_t occ_res;
  float density_bias; /* sigma = exp(out0 + density_bias) */
  float table_amp;    /* synthetic generator only: table ~ U(-amp, amp) */
  float per_level_scale; /* 0: the levels grow geometrically from base_res to finest_res (double precision, nominal
                            integer resolutions hit exactly).  > 0: tiny-cuda-nn's recipe in float32 -- scale_l =
                            exp2f(l * log2f(per_level_scale)) * base_res - 1, res_l = ceilf(scale_l) + 1 -- the level
                            geometry an imported instant-ngp snapshot was trained with; finest_res is then only a label */
} prv_field_desc;

/* How a ray is sampled.
 * PRV_STEP_FIXED_S: samples_per_ray (<= 128) uniform samples between the AABB entry and exit -- the fixed sample
 *   count BASELINE configs[1] and [3] prescribe.
 * PRV_STEP_NGP: the rule pyngp.Testbed.render applies behind run.py:245-247, 304 for aabb_scale = 1 (the reference's
 *   ray_casting_aabb_scale, DefaultConfiguration.yaml:36): a fixed step dt = sqrt(3)/1024 from the AABB entry, sample i
 *   at t0 + (i + 1/2) dt while inside the box (at most PRV_NGP_MAX_STEPS, the cube's diagonal), every step tested against
 *   the occupancy grid, alpha = 1 - exp(-sigma dt) with that dt; samples_per_ray is ignored.  The engine is not in
 *   the reference tree: the rule is restated from the published algorithm, without its per-ray start
 *   jitter (a renderer-side dither the reference's scores do not depend on); parity unpinned. */
#define PRV_STEP_FIXED_S 0
#define PRV_STEP_NGP 1
#define PRV_NGP_MAX_STEPS 1024

/* render options == the knobs run.py sets on the Testbed before render():
 * w,h (run.py:304), screenshot_spp (run.py:48,304), render_min_transmittance
 * (run.py:235; the engine's default is 0.01), background_color (run.py:94,226).  samples_per_ray is the fixed
 * per-ray sample count of the BASELINE configs (step_mode PRV_STEP_FIXED_S). */
typedef struct prv_render_opts {
  int32_t width;
  int32_t height;
  int32_t samples_per_ray;
  int32_t spp;
  float min_transmittance;
  float background[4];
  int32_t step_mode; /* PRV_STEP_FIXED_S (0) | PRV_STEP_NGP */
} prv_render_opts;

typedef struct prv_score_record { /* 16 bytes: the unit of the multi-GPU all-gather */
  double score;                   /* ranking key: larger = chosen first; NaN ranks last */
  float psnr;                     /* dB (method 5), else 0 */
  float coverage;                 /* mean opacity of the render (method 5), else 0 */
} prv_score_record;

typedef struct prv_stats {
  uint64_t rays;              /* primary rays generated (pixels x spp) */
  uint64_t samples_nominal;   /* rays x samples_per_ray (PRV_STEP_NGP: rays x PRV_NGP_MAX_STEPS) */
  uint64_t samples_evaluated; /* field evaluations actually composited */
  uint64_t wave_rounds;       /* render_queue wave iterations (32 sample slots each): slot utilisation */
  uint64_t samples_live;      /* samples in occupied cells, before early termination: the march pass's own count */
} prv_stats;

/* ---- context ------------------------------------------------------------- */
/* replaces: starting train_server.py and importing pyngp (train_server.py:1-7, run.py:90) */
int prv_create(prv_ctx** out, int device_id);
void prv_destroy(prv_ctx* ctx);
/* replaces: nothing -- the reference returns 0 unconditionally (main.cpp:1714) and hangs
 * on failure (main.cpp:1695-1698).  ctx may be NULL for the last error of a failed create. */
const char* prv_last_error(const prv_ctx* ctx);
int prv_abi_version(void);
/* enqueue all work on this hipStream_t from now on; NULL is HIP's legacy default stream.
 * Until this is called the context uses a private non-blocking stream. */
int prv_set_stream(prv_ctx* ctx, void* hip_stream);
/* For a host process that OWNS the GPU runtime (prv_planner) and is about to return from main: ends its use of the GPU
 * in a defined order.  Every context must have been destroyed (PRV_E_STATE otherwise); synchronises and resets every
 * device a context was created on (hipDeviceReset), so the HIP runtime's own state -- its streams, signal pools and
 * worker threads -- is torn down HERE, while the process is still intact, and the static destructors that run after
 * main find nothing left to race with.  (A process in which a communicator loaded librccl is only synchronised, not
 * reset: that library stays loaded and releases device state of its own after main.)  (The reference's boundary had no such step: its GPU work lived in child Python
 * processes, train_server.py:12.)  A process that shares the runtime with another library (Python + torch) must NOT
 * call this. */
int prv_runtime_shutdown(void);
int prv_synchronize(prv_ctx* ctx);
int prv_device_count(void);
/* method 5's ranking key is  -PSNR_dB + weight * mean_pixels((1 - alpha)^2) : the worst-reconstructed and
 * least-covered view first.  The second term is the reference's own density term ((1 - mean_density)^2 per
 * pixel, main.cpp:2148, added with weight 1 to the colour term of method 3), here the mean over the view's
 * pixels so that it does not depend on the image size.  weight 0 ranks by PSNR alone. */
int prv_set_coverage_weight(prv_ctx* ctx, double weight);

/* Per-kernel timing with HIP events on the context's stream (for roofline accounting):
 * between begin and end every march_compact and render_queue launch is bracketed by an
 * event pair.  end synchronises and returns summed durations and launch counts. */
int prv_profile_begin(prv_ctx* ctx);
int prv_profile_end(prv_ctx* ctx, double* render_ms, int* render_launches, double* march_ms,
                    int* march_launches);
/* the render launches of the window prv_profile_end closed last, one duration (ms) each in launch order -- the scoring
 * round of an E-member ensemble launches E per round, member 0 first (main.cpp:2041-2043: one run.py per member).
 * Returns their number (>= 0; ms may be NULL or hold fewer than that) or an error code. */
int prv_profile_render_launches(prv_ctx* ctx, float* ms, int capacity);

/* thin device-memory helpers so a C/C++ host needs no HIP headers */
int prv_malloc(prv_ctx* ctx, void** dev_ptr, size_t bytes);
int prv_free(prv_ctx* ctx, void* dev_ptr);
int prv_memcpy_h2d(prv_ctx* ctx, void* dev_dst, const void* host_src, size_t bytes);
int prv_memcpy_d2h(prv_ctx* ctx, void* host_dst, const void* dev_src, size_t bytes);

/* ---- model --------------------------------------------------------------- */
/* element counts of the three parameter arrays for a descriptor */
int prv_model_sizes(const prv_field_desc* desc, uint64_t* table_halfs, uint64_t* mlp_halfs,
                    uint64_t* occ_words);
/* replaces: testbed.load_snapshot / the trained network state (run.py:123-129, 185-208).
 * host pointers; table = fp16 bit patterns, level-major; mlp = canonical [in][out] fp16,
 * layers density-1, density-2, rgb-1, rgb-2, rgb-3; occ = occ_res^3 bits, x fastest. */
int prv_model_load(prv_ctx* ctx, int slot, const prv_field_desc* desc, const uint16_t* table,
                   const uint16_t* mlp, const uint32_t* occ);
/* deterministic synthetic field (counter-based RNG), generated on the device */
int prv_model_synthetic(prv_ctx* ctx, int slot, const prv_field_desc* desc, uint64_t seed);
/* a field to start training from (replaces the network reset of a new Testbed, run.py:90): the same
 * counter-RNG parameters as prv_model_synthetic -- table ~ U(-table_amp, table_amp), use 1e-4 as upstream
 * does; Xavier-uniform MLP -- with EVERY occupancy cell set */
int prv_model_fresh(prv_ctx* ctx, int slot, const prv_field_desc* desc, uint64_t seed);
int prv_model_export(prv_ctx* ctx, int slot, uint16_t* table, uint16_t* mlp, uint32_t* occ);
/* replaces: testbed.save_snapshot / load_snapshot (run.py:123-127, 210-211).  File = "PRVF" magic,
 * ABI version, prv_field_desc, then table / mlp / occupancy arrays in the canonical layout. */
int prv_model_save_file(prv_ctx* ctx, int slot, const char* path);
int prv_model_load_file(prv_ctx* ctx, int slot, const char* path);
/* replaces: testbed.load_snapshot(<file>.ingp | .msgpack) / save_snapshot (run.py:123-127, 210-211) for snapshots
 * written by instant-ngp itself -- the weights BASELINE configs[2] names.  Reads the msgpack (gzip-compressed for
 * .ingp) network config + "snapshot" {params_binary fp16, density_grid_binary, nerf.aabb_scale}, maps tiny-cuda-nn's
 * parameter order (density MLP | rgb MLP | hash grid; FullyFusedMLP matrices [out][in], outputs padded to 16) to the
 * canonical layout above and the density grid (Morton order, optical thickness) to the occupancy bits; the level
 * geometry follows the file's per_level_scale (prv_field_desc.per_level_scale).  Anything this build cannot
 * represent is refused with PRV_E_INVALID and a message (aabb_scale != 1, L*F != 32, other MLP shapes, ...).
 * LAYOUT ASSUMED FROM UPSTREAM, UNPINNED: neither instant-ngp nor a snapshot exists in the reference tree or the
 * build container; save_ingp is the exact inverse and the pair is tested against an independent Python writer. */
int prv_model_load_ingp(prv_ctx* ctx, int slot, const char* path);
int prv_model_save_ingp(prv_ctx* ctx, int slot, const char* path);
/* the descriptor of the field in a slot (e.g. of a snapshot just loaded) */
int prv_model_desc(prv_ctx* ctx, int slot, prv_field_desc* out);

/* ---- cameras ------------------------------------------------------------- */
/* replaces: json.load(--screenshot_transforms) + set_nerf_camera_matrix + fov from
 * camera_angle_x (run.py:132-135, 285-286, 294-296).  Reads camera_angle_x, w, h, scale,
 * offset and frames[].transform_matrix of a transforms.json written by the planner
 * (main.cpp:1793-1811, 1885-1924). */
int prv_cameras_from_json(prv_ctx* ctx, const char* path, prv_camset** out);
/* same, from memory: tm = n row-major 4x4 transform_matrix */
int prv_cameras_from_matrices(prv_ctx* ctx, const double* tm, int n, double camera_angle_x,
                              int width, int height, double scale, const double offset[3],
                              prv_camset** out);
/* The cameras of a DATASET json, as the engine sees its training / test views: replaces
 * testbed.load_training_data(--test_transforms) + set_camera_to_training_view(i) with
 * render_with_lens_distortion (run.py:145, 238, 242) -- the per-file intrinsics the planner writes at
 * main.cpp:1585-1602 (fl_x, fl_y, cx, cy, w, h; k1, k2, p1, p2 read BY KEY = the yaml's color_k1, color_k2,
 * color_p1, color_p2; k3 is written too but is not a term of the engine's OpenCV lens and is ignored).  Missing fl_* fall
 * back to camera_angle_*, missing cx, cy to the image centre, missing lens terms to 0.  Rendering at
 * another size than (w,h) scales fl and the principal point per axis. */
typedef struct prv_intrinsics {
  double fl_x, fl_y, cx, cy; /* pixels, at (w,h) */
  double k1, k2, p1, p2;     /* OpenCV radial / tangential terms on normalised coordinates */
  int32_t w, h;
} prv_intrinsics;
int prv_cameras_from_dataset_json(prv_ctx* ctx, const char* path, prv_camset** out);
int prv_cameras_from_matrices_intr(prv_ctx* ctx, const double* tm, int n, const prv_intrinsics* intr,
                                   double scale, const double offset[3], prv_camset** out);
/* lens terms {k1,k2,p1,p2} of camera i (all 0 for the screenshot-style sets above) */
int prv_camset_lens(const prv_camset* cs, int i, float lens[4]);
int prv_camset_count(const prv_camset* cs);
int prv_camset_size(const prv_camset* cs, int* width, int* height);
/* engine-frame camera i: c2w[12] row-major 3x4, intr = {fx, fy, cx, cy} at the json w,h */
int prv_camset_get(const prv_camset* cs, int i, float c2w[12], float intr[4]);
void prv_camset_destroy(prv_camset* cs);

/* ---- render -------------------------------------------------------------- */
/* replaces: testbed.render(w, h, spp, linear=True) per frame (run.py:245-247, 304).
 * out_rgba_dev: n_views * h * w * 4 float32, linear premultiplied RGBA, NO background. */
int prv_render(prv_ctx* ctx, int model_slot, const prv_camset* cs, const int* view_ids, int n_views,
               const prv_render_opts* opts, float* out_rgba_dev, prv_stats* stats);
/* replaces: write_image(outname, image) PNG bytes (run.py:309) -- composited over
 * opts->background, un-premultiplied, sRGB, 8 bit.  out: n_views*h*w*4 uint8 (dev). */
int prv_render_rgba8(prv_ctx* ctx, int model_slot, const prv_camset* cs, const int* view_ids,
                     int n_views, const prv_render_opts* opts, uint8_t* out_rgba8_dev,
                     prv_stats* stats);
int prv_quantize_rgba8(prv_ctx* ctx, const float* rgba_dev, size_t n_pixels, const float bg[4],
                       uint8_t* out_rgba8_dev);

/* replaces: Perception_3D::precept / precept_thread_process, the reference's CPU render path
 * (main.cpp:98-284: project -> ray -> OctoMap castRay, max range main.cpp:258).  Per pixel the
 * first occupied voxel of the model's occupancy grid along the ray: out_cell_dev[n_views*h*w] =
 * x + R*(y + R*z), or -1 for no hit within max_range (in unit-cube units). */
int prv_first_hit(prv_ctx* ctx, int model_slot, const prv_camset* cs, const int* view_ids, int n_views,
                  int width, int height, float max_range, int32_t* out_cell_dev);

/* The reference's camera intrinsics (rs2_intrinsics, Share_Data.hpp:79-89); coeffs in the YAML order
 * k1,k2,k3,p1,p2 (Share_Data.hpp:395-399); model 2 = inverse Brown-Conrady (yaml color_model). */
typedef struct prv_rs2_intrinsics {
  int32_t width, height;
  float ppx, ppy, fx, fy;
  int32_t model;
  float coeffs[5];
} prv_rs2_intrinsics;
/* replaces: Perception_3D::precept + precept_thread_process (main.cpp:98-284) in full: for every
 * ground-truth voxel centre, rs2_project_point_to_pixel -> cull -> integer pixel ->
 * project_pixel_to_ray_end (rs2_deproject_pixel_to_point at depth 1, Share_Data.hpp:719-726) ->
 * castRay from the camera within max_range.  voxels_dev = n x 3 float (unit-cube coordinates),
 * c2w = camera-to-world 4x4 row-major double (+Z forward, the reference's view_pose_world).
 * out_cell_dev[n] = first occupied cell x + R*(y + R*z), or -1 (culled / no hit). */
int prv_precept(prv_ctx* ctx, int model_slot, const float* voxels_dev, int n, const double c2w[16],
                const prv_rs2_intrinsics* intr, float max_range, int32_t* out_cell_dev);

/* ---- scores -------------------------------------------------------------- */
/* replaces: the per-view loops main.cpp:2045-2094 (method 2) / 2105-2158 (method 3).
 * imgs_dev[e] = n_views*pixels_per_view*4 uint8 of ensemble member e. */
int prv_score_ensemble_images(prv_ctx* ctx, int method, const uint8_t* const* imgs_dev,
                              int n_members, int n_views, size_t pixels_per_view,
                              prv_score_record* records_host);
/* replaces: run.py:257-263 per image.  score = -psnr + coverage weight * mean((1 - alpha)^2), see
 * prv_set_coverage_weight (worst-reconstructed, least-covered view first). */
int prv_score_psnr_images(prv_ctx* ctx, const float* rgba_dev, const float* gt_rgba_dev,
                          int n_views, size_t pixels_per_view, const float bg[4],
                          prv_score_record* records_host);
/* replaces: the per-image metrics of the evaluation block, run.py:257-267:
 *   A,R = clip(srgb(img)), mse -> psnr = -10 log10(mse), ssim = SSIM(A,R)   (SSIM recipe assumed from
 *   instant-ngp's scripts/common.py, which the reference does not vendor).
 * Images: n_views * h * w * 4 float (dev).  psnr_host / ssim_host: n_views doubles each. */
int prv_evaluate_images(prv_ctx* ctx, const float* rgba_dev, const float* gt_rgba_dev, int n_views,
                        int width, int height, const float bg[4], double* psnr_host, double* ssim_host);
/* replaces: the evaluation loop run.py:240-277 for one model: render every listed view at opts,
 * compare with the reference images, return the MEAN of the per-image PSNRs and SSIMs (run.py:271-272;
 * what --save_metrics writes as "PSNR\t..\nSSIM\t..", consumed at main.cpp:1957-1961). */
int prv_evaluate(prv_ctx* ctx, int model_slot, const prv_camset* cs, const int* view_ids, int n_views,
                 const prv_render_opts* opts, const float* gt_rgba_dev, double* mean_psnr, double* mean_ssim);

/* The whole scoring round of nbv_loop for one shard of views, on the device end to end:
 * render every view with every model slot listed, reduce to one record per view.
 *   methods 2,3: model_slots = the ensemble (main.cpp:2041-2043 + 2045-2094);
 *   method 5   : model_slots[0] vs gt_rgba_dev (n_views*h*w*4 float, same view order).
 * records_host and/or records_dev (n_views records) receive the result; records_dev is
 * what a caller hands to its all-gather. */
int prv_score_views(prv_ctx* ctx, int method, const int* model_slots, int n_models,
                    const prv_camset* cs, const int* view_ids, int n_views,
                    const prv_render_opts* opts, const float* gt_rgba_dev,
                    prv_score_record* records_host, prv_score_record* records_dev,
                    prv_stats* stats);
/* replaces: the arg-max bookkeeping main.cpp:1971-1972, 2088-2091, 2096.
 * order = view ids sorted by (score descending, id ascending), NaN scores after every number (the reference's
 * strict '>' never selects a NaN either); host only. */
int prv_rank(const prv_score_record* records, const int* view_ids, int n, int* order);
int prv_argmax(const prv_score_record* records, const int* view_ids, int n);

/* replaces: Perception_3D::render (the PCL screenshot of the coloured ground-truth cloud with
 * points_size_cloud-pixel points on white, main.cpp:68-96) + convertToAlpha (Share_Data.hpp:771-784) +
 * cv::flip(-1) (main.cpp:1616) = the rgbaClip_<i>.png training images of get_coverage (main.cpp:1604-1618),
 * as a z-buffered square-splat rasteriser.  xyz_dev n*3 floats in world units, rgb_dev n*3 bytes; scale /
 * offset = the json's (the cloud goes where the cameras go); cameras = the dataset's.  out: n_views*h*w*4
 * RGBA bytes (device): nearest point's colour, alpha 255; background and exactly-white points 255,255,255,0. */
int prv_splat_points(prv_ctx* ctx, const float* xyz_dev, const uint8_t* rgb_dev, size_t n_points, double scale,
                     const double offset[3], const prv_camset* cs, const int* view_ids, int n_views, int width,
                     int height, int point_size, int flip180, uint8_t* out_rgba8_dev);

/* ---- several GPUs: view sharding + one all-gather ----------------------------- */
/* replaces: nothing in the reference -- its loops over the candidates are serial (main.cpp:2045-2094, 2105-2158;
 * run.py:293) and it has no multi-GPU path.  One process per GPU (RANK / WORLD_SIZE / LOCAL_RANK as torchrun exports
 * them): the candidate views of a round are dealt to the ranks, every rank scores its shard with the whole ensemble,
 * ONE all-gather of the 16-byte records gives every rank the whole round, the same prv_rank / prv_argmax on every
 * rank gives the same integer ranking.
 * transport "rccl" (default): ncclAllGather / ncclBroadcast on device buffers over xGMI, on the context's stream;
 *   librccl is loaded at run time, rank 0's ncclUniqueId reaches the others over a TCP star at `rendezvous`
 *   ("host:port"; NULL = $MASTER_ADDR : $PRV_COMM_PORT, else $MASTER_PORT + 23).
 * transport "socket": the same calls staged through host memory and that star -- for ranks that SHARE a GPU (RCCL
 *   refuses that), i.e. tests.  NULL transport = $PRV_COMM, else "rccl". */
typedef struct prv_comm prv_comm;
int prv_comm_create(prv_ctx* ctx, int rank, int world, const char* transport, const char* rendezvous, prv_comm** out);
void prv_comm_destroy(prv_comm* comm);
int prv_comm_rank(const prv_comm* comm);
int prv_comm_world(const prv_comm* comm);
const char* prv_comm_transport(const prv_comm* comm); /* "rccl" | "socket" */
/* which librccl the communicator's calls land in (transport "rccl"; empty strings / 0 for "socket"): the file's path as
 * the dynamic loader resolved it, ncclGetVersion's code, and how it was found -- "PRV_RCCL_LIB", "already mapped by the
 * process" (the host program's own copy, e.g. torch.distributed's: ONE RCCL per process) or "soname search" */
int prv_comm_library(const prv_comm* comm, char* path_out, int path_cap, int* version_out, char* how_out, int how_cap);
/* recv_dev = world blocks of bytes_per_rank in rank order, identical on every rank; enqueued on the context's stream */
int prv_comm_all_gather(prv_comm* comm, const void* send_dev, size_t bytes_per_rank, void* recv_dev);
int prv_comm_barrier(prv_comm* comm);
/* the views of `rank`: a contiguous block [rank*per, ...) or, interleaved, rank, rank+world, ... (a hemisphere set runs
 * pole -> equator and top views cost more).  ids_out (may be NULL) holds up to per entries; returns per = ceil(n/world). */
int prv_shard_views(int n_views, int rank, int world, int interleaved, int* ids_out, int* n_mine);
/* the sharded scoring round: prv_score_views on this rank's shard of views [0, n_views_total) of `cs`, one all-gather,
 * records_host = all n_views_total records in view order on every rank.  gt_shard_dev (method 5): the reference images
 * of THIS rank's views, in shard order.  comm NULL = one rank.  stats: this rank's share. */
int prv_score_views_sharded(prv_ctx* ctx, prv_comm* comm, int method, const int* model_slots, int n_models,
                            const prv_camset* cs, int n_views_total, int interleaved, const prv_render_opts* opts,
                            const float* gt_shard_dev, prv_score_record* records_host, prv_stats* stats);
/* the exchange step of a multi-GPU NBV iteration (the reference trains its members one after another in one process,
 * main.cpp:2041-2043): member e was trained in slot e of rank e % world; afterwards slot e of EVERY rank holds it bit
 * for bit.  Device to device: one group of broadcasts on the slots' canonical buffers (table | MLP | occupancy). */
int prv_model_exchange(prv_ctx* ctx, prv_comm* comm, int n_members, const prv_field_desc* desc);
/* the general form: member e lives in model slot slots[e] and was trained by rank owners[e] (prv_planner's `shard: members`
 * deals the (object, member) trainings of a round to the ranks round-robin, so an ensemble's members come from different
 * ranks and different objects' ensembles sit in different slots); prv_model_exchange = slots e, owners e % world */
int prv_model_exchange_slots(prv_ctx* ctx, prv_comm* comm, int n_members, const int* slots, const int* owners,
                             const prv_field_desc* desc);

/* ---- training ------------------------------------------------------------- */
/* replaces: the `while testbed.frame()` loop run.py:185-208 drives for `--train --n_steps 2500`
 * (main.cpp:1668) on the dataset of testbed.load_training_data (run.py:109): random rays over the dataset
 * images, L2 loss on linear colours over a random background, backward through both MLPs and the hash
 * grid, Adam with sparse table updates, periodic density-grid refresh -- the published instant-ngp
 * optimiser restated (the optimiser itself is inside pyngp, not in the reference tree).
 * The model slot is trained IN PLACE: after every prv_train_steps call the slot renders / scores with the
 * trained weights and occupancy. */
typedef struct prv_train_opts {
  int32_t n_rays;    /* rays per step (with target_samples: the cap of the adaptive count; default 2^16) */
  int32_t n_samples; /* PRV_STEP_FIXED_S: samples per ray between the AABB hits, <= 128.  PRV_STEP_NGP: the most steps a ray takes,
                        <= PRV_NGP_MAX_STEPS (1024 = the cube's diagonal, what prv_train_default_opts sets for that rule) */
  float lr, beta1, beta2, eps, l2_reg; /* Adam; l2_reg on the MLP weights only */
  float min_T;       /* early termination of a training ray */
  uint64_t seed;
  int32_t random_bg; /* 1: random background colour per ray */
  int32_t occ_every; /* refresh the density grid every N steps (0 = never) */
  float occ_decay, occ_sigma_thresh; /* ema = max(ema*decay, sigma); occupied iff ema > thresh (default 5.9 =
                                        upstream's optical thickness 0.01 over a sqrt(3)/1024 step) */
  int32_t target_samples; /* > 0: the ray count of a step adapts so that about this many samples are composited
                             (upstream keeps 2^18 samples per batch): after every step active = clamp(target *
                             active / used, active/2, 2*active) within [1, n_rays]; first step min(n_rays,
                             target / n_samples).  0: always n_rays */
  int32_t patch_w, patch_h; /* > 1: a step's rays are drawn as patches of patch_w x patch_h adjacent pixels of one image
                               (patch_w * patch_h <= 16; ray j = pixel j % P of patch j / P, rows walked in snake order)
                               that share one jitter, and the step's sample list is ordered depth step by depth step
                               inside a patch, so that the samples of a backward tile share table entries (fewer
                               memory-side atomic requests per step); 0 or 1: every ray its own pixel (the published
                               i.i.d. sampler).  prv_train_default_opts says which this build defaults to.  PRV_STEP_FIXED_S only */
  int32_t step_mode; /* how a training ray is sampled (the values of prv_render_opts.step_mode).  PRV_STEP_FIXED_S: n_samples
                        uniform samples between the AABB hits under one random offset per ray (rounds 1-5).  PRV_STEP_NGP: the
                        engine's own marcher, what upstream trains with behind run.py:188 `testbed.frame()` --
                        fixed step dt = sqrt(3)/1024 from the AABB entry, per-ray random start: sample i at t0 + (i + jitter) dt
                        while that is inside the box and i < n_samples, every step tested against the occupancy grid,
                        alpha = 1 - exp(-sigma dt) with that dt; the sample budget (target_samples) is unchanged, its first step
                        casts target_samples / n_samples rays.  A step lists at most 2^24 samples under this rule (64 x the
                        default budget); a batch beyond that makes prv_train_steps fail with PRV_E_STATE */
  int32_t deterministic; /* 1 (tests): bit-reproducible training -- the ray batches are listed in ray order and the table gradient
                            is summed in 64-bit fixed point (order-independent) instead of f32 atomics; slower.  0: the product path */
} prv_train_opts;
typedef struct prv_trainer prv_trainer;
int prv_train_default_opts(prv_train_opts* opts);
/* dataset = cameras of prv_cameras_from_dataset_json (own intrinsics + lens) and their images,
 * n * height * width * 4 straight-alpha sRGB bytes in device memory (caller keeps them alive); when
 * (width, height) differs from the dataset's size the intrinsics are scaled per axis */
int prv_train_create(prv_ctx* ctx, int model_slot, const prv_camset* dataset, const uint8_t* images_rgba8_dev,
                     int width, int height, const prv_train_opts* opts, prv_trainer** out);
/* n optimiser steps; losses_host (n floats, may be NULL) = the batch loss of each step before its update */
int prv_train_steps(prv_trainer* t, int n_steps, float* losses_host);
/* the members of an ensemble side by side: step i of every trainer (each on its own HIP stream) before step
 * i+1 of any; one slot per trainer, one context; losses_host = n_trainers rows of n_steps, may be NULL */
int prv_train_steps_multi(prv_trainer** trainers, int n_trainers, int n_steps, float* losses_host);
int prv_train_info(const prv_trainer* t, uint32_t* steps_done, uint64_t* samples_last_batch,
                   uint64_t* table_scalars);
/* rays the next step will cast (= n_rays unless target_samples is set) */
int prv_train_active_rays(const prv_trainer* t);
/* STREAMS: a trainer runs on a stream of its own that owns a hardware queue (hipExtStreamCreateWithCUMask naming every CU; the
 * pooled streams of hipStreamCreate share four queues, and five members stepping side by side sat on three of them).  That API
 * takes no flags, so the stream is a BLOCKING one: work an embedder puts on the LEGACY default stream (stream 0; PyTorch's
 * default stream is that one) serialises with every trainer's stream, which undoes the side-by-side overlap of
 * prv_train_steps_multi while such work is in flight.  Hosts that use the legacy default stream should move that work to a
 * stream created with hipStreamNonBlocking (or build with per-thread default streams); PRV_TRAIN_OWN_QUEUE=0 in the
 * environment gives the trainers pooled non-blocking streams instead (slower side by side: round of five 0.524 -> 0.503 ms
 * with own queues, profiles/NOTES.md round 5).
 * The trainer's stream (it owns a hardware queue) and device buffers stay with the CONTEXT for its next trainer of the same
 * sizes (at most 16 GiB / 512 buffers, oldest released first); prv_destroy releases them. */
void prv_train_destroy(prv_trainer* t);
/* parity hooks: gradients of the NEXT batch without an update (host arrays: table_scalars and
 * PRV_MLP_HALFS floats), the fp32 master weights, one density-grid refresh */
int prv_train_gradients(prv_trainer* t, float* table_grad_host, float* mlp_grad_host, float* loss);
int prv_train_master(prv_trainer* t, float* table_host, float* mlp_host);
int prv_train_refresh_occupancy(prv_trainer* t);
/* dev only: 64 phase time stamps of the tile kernels' block 0 (zeros unless built with PRV_TRAIN_ABLATE=16) */
int prv_train_debug_stamps(prv_trainer* t, unsigned long long* out64);

/* ---- stage hooks for parity tests (host in / host out, small n) ---------- */
/* how a loaded field is laid out for the kernels, and which render_queue64_kernel<F, NDENSE> instance prv_render
 * launches for it (tests assert that e.g. the 512^3 / F=2 field really runs the <2,10> instance) */
typedef struct prv_model_layout {
  uint64_t table_bytes_canonical; /* ABI layout: what prv_model_load / export carry */
  uint64_t table_bytes_physical;  /* kernel layout: power-of-two strides on the dense levels, size-aligned levels */
  int32_t kernel_features;        /* F of the instance */
  int32_t kernel_dense_levels;    /* NDENSE of the instance */
  int32_t n_hashed_levels;
  int32_t kernel_slots;           /* ray slots per wave of the render kernel (64) */
  int32_t n_dense_levels;         /* leading physically dense levels */
} prv_model_layout;
int prv_debug_model_layout(prv_ctx* ctx, int model_slot, prv_model_layout* out);
/* Shader clock the render launches actually ran at (roofline accounting: the VALU issue peak is per clock and the
 * clock under this load is well below the 2.4 GHz maximum).  One wave of every render_queue launch stamps the shader
 * cycle counter and the constant-rate reference counter at its start and end; this returns the two sums since the
 * statistics were last cleared (a prv_render / prv_score_views call with stats clears them) and the reference rate.
 * Average clock = shader_cycles / (ref_ticks / ref_hz).  Synchronises the stream. */
int prv_debug_render_clock(prv_ctx* ctx, uint64_t* shader_cycles, uint64_t* ref_ticks, double* ref_hz);
/* rays of view i at (w,h): o,d = n*3, t = n*2 (AABB entry/exit; exit<=entry => miss) */
int prv_debug_raygen(prv_ctx* ctx, const prv_camset* cs, int view, int width, int height,
                     int spp_index, float* o, float* d, float* t);
/* 32 fp16 features per position (bit patterns) */
int prv_debug_encode(prv_ctx* ctx, int model_slot, const float* pos, int n, uint16_t* feat);
/* per point: out[0]=sigma, out[1..3]=rgb, out[4..19]=density MLP outputs, out[20..35]=rgb MLP
 * outputs; occupied[n] = occupancy bit */
int prv_debug_field(prv_ctx* ctx, int model_slot, const float* pos, const float* dir, int n,
                    float* out36, int32_t* occupied);

#ifdef __cplusplus
}
#endif
#endif /* PRV_H */
