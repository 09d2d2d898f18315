/*
 * prv_host.h -- C ABI of the planner-side host library (libprv_host.so, pure C++17, no GPU).
 *
 * These are the pieces of PRV_simulation that sit on either side of the render boundary:
 * the candidate-view set, the per-view camera pose search, the transforms.json camera
 * interface, the config file, and the next-best-view loop.  The C++ classes behind this
 * ABI keep the reference's names (Share_Data, View, View_Space, NBV_Net_Labeler) in
 * nerf_prv_amd/host/; this header is what tests and non-C++ hosts bind.
 */
#ifndef PRV_HOST_H
#define PRV_HOST_H
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

/* No exception crosses this ABI: an int / count function that ran out of memory (or hit any other internal
 * error) returns PRVH_E_INTERNAL, a double-valued one -1.0, prvh_share_data_create NULL. */
#define PRVH_E_INTERNAL (-100)

/* View::get_next_camera_pos(now_camera_pose_world = I, object_center_world, type 0)
 * (View_Space.hpp:67-140): pose = world->camera 4x4, row-major */
void prvh_view_pose(const double init_pos[3], const double center[3], double pose[16]);
/* frame transform_matrix = P * pose^-1 * diag(1,-1,-1,1)  (main.cpp:1626-1641) */
void prvh_transform_matrix(const double pose[16], double tm[16]);
/* View_Space::get_view_space candidate positions (View_Space.hpp:550-556); returns count */
int prvh_view_space(const double* pt_sphere, int n, double radius, const double center[3], double* out_pos);
/* centroid and 17/16 x bounding radius of a cloud (View_Space.hpp:534-548) */
void prvh_bbx(const double* pts, int n, double center[3], double* predicted_size);
/* read Hemisphere/<N>.txt style files: n rows of 3 numbers (Share_Data.hpp:517-525); returns rows read */
int prvh_hemisphere_read(const char* path, int n, double* out_xyz);
/* generated view set for synthetic configs: Fibonacci hemisphere, row 0 = (0,0,1), z > 0 */
int prvh_hemisphere_generate(int n, double* out_xyz);

typedef struct prvh_intrinsics { /* Share_Data::color_intrinsics (Share_Data.hpp:79-89) */
  int32_t width, height;
  double ppx, ppy, fx, fy;
  double coeffs[5]; /* k1 k2 k3 p1 p2 in the YAML order (Share_Data.hpp:395-399) */
} prvh_intrinsics;

/* transforms.json as get_coverage / nbv_loop emit it (main.cpp:1584-1602, 1793-1811):
 * candidate_header != 0 selects the 1/16-resolution, zero-distortion header of the render json.
 * frames: init_pos[n*3] world positions -> pose search -> transform_matrix; file_path =
 * "<path_prefix><id>.png".  Returns 0 or a negative error. */
int prvh_write_transforms(const char* path, const prvh_intrinsics* intr, int candidate_header,
                          double candidate_divisor, int aabb_scale, double predicted_size,
                          const double center[3], const double* init_pos, const int* ids, int n,
                          const char* path_prefix);

/* the metrics file of --save_metrics: "PSNR\t<psnr>\nSSIM\t<ssim>" (run.py:274-277), read back with
 * fscanf / >> at main.cpp:1957-1961 and NeRF_fit_curve.cpp:103-115 */
int prvh_write_metrics(const char* path, double psnr, double ssim);
int prvh_read_metrics(const char* path, double* psnr, double* ssim);

/* ---- movement cost and visiting order (View_Space.hpp:206-305, main.cpp:398-594) ---- */
/* get_local_path: straight segment, or detour on the sphere (O, r); *type_out = 0 line, 1 arc, -1 wrong */
double prvh_local_path(const double M[3], const double N[3], const double O[3], double r, int* type_out);
/* Global_Path_Planner: shortest path through the n positions from start (end = -1: free end), edge cost
 * = prvh_local_path; order_out[n] = visiting order (indices), *exact_out = 1 when provably optimal
 * (n <= 20, Held-Karp; the reference uses Gurobi).  Returns the path length, or < 0. */
double prvh_global_path(const double* positions, int n, int start, int end, const double O[3], double r,
                        int* order_out, int* exact_out);

/* ---- stopping criterion (Origin_scripts/NeRF_fit_curve.cpp:56-212) ---- */
/* fit y = y0 + A*Phi((ln x - xc)/w) to (views, psnr); params_out = {y0, A, xc, w}; converged_out as in
 * label.txt (solver outcome AND no data point above max_psnr, :143-151).  Returns 0 or < 0. */
int prvh_fit_curve(const double* x, const double* y, int n, double max_psnr, double params_out[4], int* converged_out);
/* labels of :186-206 from a fitted curve: gap_out[11] (k = 0..10 %), gradient_out[20] (g = 0.01..0.20) */
void prvh_fit_labels(const double params[4], double max_psnr, int gap_out[11], int gradient_out[20]);
/* label.txt in the reference's format */
int prvh_write_label(const char* path, const double params[4], int converged, double max_psnr);

/* ---- config (Share_Data) ---- */
typedef struct prvh_share_data prvh_share_data;
/* Share_Data(config, name, num_of_views, id_of_batch, method) (Share_Data.hpp:334) */
prvh_share_data* prvh_share_data_create(const char* yaml_path, const char* test_name, int num_of_views,
                                        int id_of_batch, int test_method);
void prvh_share_data_destroy(prvh_share_data*);
const char* prvh_share_data_error(void);
/* string / number lookups by the reference's field names, e.g. "save_path", "ensemble_num" */
const char* prvh_share_data_string(const prvh_share_data*, const char* field);
double prvh_share_data_number(const prvh_share_data*, const char* field);
int prvh_share_data_views(const prvh_share_data*, double* out_xyz /* num_of_views*3 or NULL */);
void prvh_share_data_intrinsics(const prvh_share_data*, prvh_intrinsics* out);

/* ---- the planner loop (NBV_Net_Labeler::nbv_loop, main.cpp:1718-2277) ---- */
/* scorer: the render boundary as seen from the loop.  Given the render json of this
 * iteration and the candidate ids in it, fill one score per candidate (larger = better).
 * Production wires this to prv_score_views; tests may wire anything. */
typedef int (*prvh_score_fn)(void* user, int method, int iteration, const char* scene_json,
                             const char* render_json, const int* candidate_ids, int n, double* scores);
typedef struct prvh_loop_result {
  int n_chosen;
  int chosen[1024];
  double total_movement; /* sum of get_local_path costs between consecutive chosen views (main.cpp:2256-2264) */
} prvh_loop_result;
/* ---- PNG files of the boundary (rgbaClip_<i>.png written at main.cpp:1617, screenshots of run.py:309 read at
 * main.cpp:2047, 2107): 8-bit grey / RGB / RGBA, non-interlaced; pixels as RGBA8, top row first.
 * 0, or -1 io, -2 not a PNG, -3 unsupported flavour, -4 corrupt, -5 size mismatch */
int prvh_png_size(const char* path, int* width, int* height);
int prvh_png_read_rgba8(const char* path, int width, int height, uint8_t* out_rgba8);
int prvh_png_write_rgba8(const char* path, int width, int height, const uint8_t* rgba8);
/* replaces: the body of the per-view loops main.cpp:2045-2094 (method 2, EnsembleRGB) / 2105-2158 (method 3,
 * EnsembleRGBDensity) for ONE unchosen view, on the host as the reference has it: files[e] = the screenshot member e's
 * engine call left (render/<it>/ensemble_<e>/rgbaClip_<v>.png), *score = view_uncertainty in the reference's operation
 * order (BGRA channel order of cv::imread included).  What `score_path: png` of prv_planner runs; the fused device path
 * (prv_score_views, include/prv.h) returns the same doubles.  0, -1 bad argument, -18 a file is missing / unreadable /
 * of another size or the method is not 2 or 3. */
int prvh_score_view_pngs(int method, const char* const* files, int n_members, double* score);

/* ---- instant-ngp snapshots on the host (nerf_prv_amd/csrc/prv_ingp.hpp; what prv_model_load_ingp / save_ingp of
 * include/prv.h do, minus the GPU): <file>.ingp | .msgpack <-> descriptor + canonical arrays.  desc points at a
 * prv_field_desc (include/prv.h).  Read: call once with NULL arrays for the sizes, then with arrays of those sizes.
 * 0, or PRV_E_IO (-3) unreadable / malformed, PRV_E_INVALID (-1) not representable; message in err (may be NULL). */
struct prv_field_desc;
int prvh_ingp_read(const char* path, struct prv_field_desc* desc, uint64_t* n_table_halfs, uint64_t* n_occ_words,
                   uint16_t* table, uint16_t* mlp, uint32_t* occ, char* err, int err_cap);
int prvh_ingp_write(const char* path, const struct prv_field_desc* desc, const uint16_t* table, uint64_t n_table_halfs,
                    const uint16_t* mlp, const uint32_t* occ, uint64_t n_occ_words, char* err, int err_cap);

/* ---- the TCP star between the ranks of one job (nerf_prv_amd/csrc/prv_star.hpp) ----
 * the rendezvous of prv_comm (it carries rank 0's ncclUniqueId to the other ranks) and the host-staged `socket`
 * transport behind the same calls; exposed here so it can be exercised without a GPU.  Rank 0 listens on addr:port
 * (NULL / <= 0: $MASTER_ADDR, $PRV_COMM_PORT or $MASTER_PORT + 23), the others connect. */
typedef struct prvh_star prvh_star;
prvh_star* prvh_star_open(int rank, int world, const char* addr, int port, double timeout_s);
void prvh_star_close(prvh_star*);
/* recv = world blocks of `bytes` in rank order, identical on every rank; 0 or < 0 */
int prvh_star_all_gather(prvh_star*, const void* send, uint64_t bytes, void* recv);
int prvh_star_broadcast(prvh_star*, void* buf, uint64_t bytes, int root);
int prvh_star_barrier(prvh_star*);

int prvh_nbv_loop(prvh_share_data* sd, const double center[3], double predicted_size, int first_view_id,
                  int test_id, prvh_score_fn score, void* user, prvh_loop_result* out);
/* 1 when nbv_loop runs this method_of_IG (0, 2, 3, 5); 0 for RandomOneshot = 1 and PVBCoverage = 4 (main.cpp:1981-2037,
 * 2163-2242: the PRVNet pipeline, outside this build's scope) -- prvh_nbv_loop refuses those with -10 before it
 * writes anything */
int prvh_method_in_scope(int method_of_IG);

/* prv_planner `shard: members` (BASELINE configs[4] on more GPUs than objects): which rank trains member `member` of object
 * `object` in a lockstep round -- the n_objects x n_members trainings are dealt round-robin, pair object * n_members + member
 * to rank pair % world (main.cpp:2041-2043, 2101-2103 train them one after another).  Returns the rank, or -1 for arguments that make no sense */
int prvh_member_owner(int object, int member, int n_members, int world);

/* DEPRECATED, kept so that binaries linked against the round 1-3 library still load: both belonged to the out-of-scope
 * methods (the view budget of method 4; the PCD reader of the asset preparation).  Each returns PRVH_E_UNSUPPORTED and
 * touches nothing. */
#define PRVH_E_UNSUPPORTED (-95)
long long prvh_pcd_read(const char* path, float* xyz_out, uint8_t* rgb_out, long long capacity);
int prvh_nbv_loop_budget(prvh_share_data* sd, const double center[3], double predicted_size, int first_view_id,
                         int test_id, prvh_score_fn score, void* user, int view_budget, prvh_loop_result* out);

#ifdef __cplusplus
}
#endif
#endif
