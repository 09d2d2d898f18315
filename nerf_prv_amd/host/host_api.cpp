// host_api.cpp -- C ABI of the planner-side host library (include/prv_host.h).
#include "../../include/prv_host.h"

#include <cstring>
#include <string>

#include "fit_curve.hpp"
#include "path_planner.hpp"
#include "planner.hpp"
#include "png_io.hpp"
#include "../csrc/prv_star.hpp"
#include "../csrc/prv_ingp.hpp"

using namespace prvhost;

namespace {
thread_local std::string g_error;
}

struct prvh_star {
  prvstar::Star star;
};

struct prvh_share_data {
  std::shared_ptr<Share_Data> sd;
  std::string scratch;
};

extern "C" {

void prvh_view_pose(const double init_pos[3], const double center[3], double pose[16]) {
  View v(Vec3(init_pos[0], init_pos[1], init_pos[2]));
  v.get_next_camera_pos(Mat4::Identity(), Vec3(center[0], center[1], center[2]));
  memcpy(pose, v.pose.m.data(), sizeof(double) * 16);
}

void prvh_transform_matrix(const double pose[16], double tm[16]) {
  Mat4 p;
  memcpy(p.m.data(), pose, sizeof(double) * 16);
  Mat4 P = Mat4::Identity(), P1 = Mat4::Identity();
  P.m = {0, 0, 1, 0, 1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 0, 1};
  P1.m = {1, 0, 0, 0, 0, -1, 0, 0, 0, 0, -1, 0, 0, 0, 0, 1};
  const Mat4 r = P * p.inverse() * P1;
  memcpy(tm, r.m.data(), sizeof(double) * 16);
}

int prvh_view_space(const double* pt, int n, double radius, const double center[3], double* out_pos) try {
  if (n <= 0) return 0;
  const double pt_norm = std::sqrt(pt[0] * pt[0] + pt[1] * pt[1] + pt[2] * pt[2]);
  int k = 0;
  for (int i = 0; i < n; i++) {
    if (pt[i * 3 + 2] < 0) continue;
    const double scale = 1.0 / pt_norm * radius;
    for (int a = 0; a < 3; a++) out_pos[k * 3 + a] = pt[i * 3 + a] * scale + center[a];
    k++;
  }
  return k;
} catch (...) { return PRVH_E_INTERNAL; }

void prvh_bbx(const double* pts, int n, double center[3], double* predicted_size) {
  Vec3 c(0, 0, 0);
  for (int i = 0; i < n; i++) c = c + Vec3(pts[i * 3], pts[i * 3 + 1], pts[i * 3 + 2]);
  c = Vec3(c.x / n, c.y / n, c.z / n);
  double s = 0;
  for (int i = 0; i < n; i++) s = std::max(s, (c - Vec3(pts[i * 3], pts[i * 3 + 1], pts[i * 3 + 2])).norm());
  center[0] = c.x; center[1] = c.y; center[2] = c.z;
  *predicted_size = s * (17.0 / 16.0);
}

int prvh_hemisphere_read(const char* path, int n, double* out) try {
  std::ifstream f(path);
  if (!f.is_open()) return -1;
  int rows = 0;
  for (; rows < n; rows++) {
    double v[3];
    if (!(f >> v[0] >> v[1] >> v[2])) break;
    memcpy(out + rows * 3, v, sizeof(v));
  }
  return rows;
} catch (...) { return PRVH_E_INTERNAL; }

int prvh_hemisphere_generate(int n, double* out) try {
  if (n < 1) return 0;
  out[0] = 0; out[1] = 0; out[2] = 1; // the top view every reference set contains (main.cpp:2212)
  const double golden = std::acos(-1.0) * (3.0 - std::sqrt(5.0));
  for (int i = 1; i < n; i++) {
    const double z = 1.0 - (double)i / (double)n; // strictly above the equator
    const double r = std::sqrt(std::max(0.0, 1.0 - z * z));
    const double th = golden * i;
    out[i * 3] = r * std::cos(th);
    out[i * 3 + 1] = r * std::sin(th);
    out[i * 3 + 2] = z;
  }
  return n;
} catch (...) { return PRVH_E_INTERNAL; }

int prvh_write_transforms(const char* path, const prvh_intrinsics* in, int candidate_header, double divisor,
                          int aabb_scale, double predicted_size, const double center[3], const double* init_pos,
                          const int* ids, int n, const char* path_prefix) try {
  if (!path || !in || !center || (n > 0 && !init_pos)) return -1;
  rs2_intrinsics K;
  K.width = in->width; K.height = in->height;
  K.ppx = (float)in->ppx; K.ppy = (float)in->ppy; K.fx = (float)in->fx; K.fy = (float)in->fy;
  for (int i = 0; i < 5; i++) K.coeffs[i] = (float)in->coeffs[i];
  const Vec3 c(center[0], center[1], center[2]);
  Value root = transforms_header(K, aabb_scale, predicted_size, c, candidate_header ? divisor : 0.0);
  const Mat4 cam = Mat4::Identity();
  for (int i = 0; i < n; i++) {
    View v(Vec3(init_pos[i * 3], init_pos[i * 3 + 1], init_pos[i * 3 + 2]));
    Value view_image;
    view_image["file_path"] = Value(std::string(path_prefix ? path_prefix : "") + std::to_string(ids ? ids[i] : i) + ".png");
    view_image["transform_matrix"] = matrix_json(view_transform_matrix(v, cam, c));
    root["frames"].append(view_image);
  }
  return write_text(path, prvjson::to_styled_string(root)) ? 0 : -3;
} catch (...) { return PRVH_E_INTERNAL; }

int prvh_png_size(const char* path, int* width, int* height) try {
  if (!path || !width || !height) return -1;
  std::vector<uint8_t> img;
  return png_read_rgba8(path, width, height, img);
} catch (...) { return PRVH_E_INTERNAL; }

int prvh_png_read_rgba8(const char* path, int width, int height, uint8_t* out_rgba8) try {
  if (!path || !out_rgba8) return -1;
  std::vector<uint8_t> img;
  int w = 0, h = 0;
  const int rc = png_read_rgba8(path, &w, &h, img);
  if (rc != 0) return rc;
  if (w != width || h != height) return -5;
  memcpy(out_rgba8, img.data(), img.size());
  return 0;
} catch (...) { return PRVH_E_INTERNAL; }

int prvh_png_write_rgba8(const char* path, int width, int height, const uint8_t* rgba8) try {
  if (!path) return -1;
  return png_write_rgba8(path, width, height, rgba8);
} catch (...) { return PRVH_E_INTERNAL; }

int prvh_score_view_pngs(int method, const char* const* files, int n_members, double* score) try {
  if (!files || n_members < 1 || !score) return -1;
  std::vector<std::string> f;
  for (int e = 0; e < n_members; e++) {
    if (!files[e]) return -1;
    f.push_back(files[e]);
  }
  return ensemble_uncertainty_from_pngs(method, f, score);
} catch (...) { return PRVH_E_INTERNAL; }

int prvh_write_metrics(const char* path, double psnr, double ssim) try {
  if (!path) return -1;
  char buf[128];
  // python's str(float) is the shortest round-trip form; %.17g round-trips too and every reader
  // on the path parses with strtod
  snprintf(buf, sizeof(buf), "PSNR\t%.17g\nSSIM\t%.17g", psnr, ssim);
  return write_text(path, buf) ? 0 : -3;
} catch (...) { return PRVH_E_INTERNAL; }

int prvh_read_metrics(const char* path, double* psnr, double* ssim) try {
  if (!path) return -1;
  std::ifstream f(path);
  if (!f.is_open()) return -3;
  std::string name;
  double value;
  int got = 0;
  while (f >> name >> value) { // main.cpp:1958-1961
    if (name == "PSNR" && psnr) { *psnr = value; got |= 1; }
    if (name == "SSIM" && ssim) { *ssim = value; got |= 2; }
  }
  return got == 3 ? 0 : -3;
} catch (...) { return PRVH_E_INTERNAL; }

double prvh_local_path(const double M[3], const double N[3], const double O[3], double r, int* type_out) try {
  const auto lp = get_local_path(Vec3(M[0], M[1], M[2]), Vec3(N[0], N[1], N[2]), Vec3(O[0], O[1], O[2]), r);
  if (type_out) *type_out = lp.first;
  return lp.second;
} catch (...) { return -1.0; }

double prvh_global_path(const double* pos, int n, int start, int end, const double O[3], double r, int* order_out, int* exact_out) try {
  if (!pos || n < 1 || start < 0 || start >= n || end >= n || !O) return -1.0;
  std::vector<View> views;
  std::vector<int> label;
  for (int i = 0; i < n; i++) {
    views.emplace_back(Vec3(pos[i * 3], pos[i * 3 + 1], pos[i * 3 + 2]));
    label.push_back(i);
  }
  Global_Path_Planner gp(views, label, start, Vec3(O[0], O[1], O[2]), r, end);
  const double len = gp.solve();
  const auto path = gp.get_path_id_set();
  if (order_out) std::copy(path.begin(), path.end(), order_out);
  if (exact_out) *exact_out = gp.exact ? 1 : 0;
  return len;
} catch (...) { return -1.0; }

int prvh_fit_curve(const double* x, const double* y, int n, double max_psnr, double params_out[4], int* converged_out) try {
  if (!x || !y || n < 4 || !params_out) return -1;
  const FitResult r = fit_lognormal_cdf(std::vector<double>(x, x + n), std::vector<double>(y, y + n), max_psnr);
  params_out[0] = r.f.y0; params_out[1] = r.f.A; params_out[2] = r.f.xc; params_out[3] = r.f.w;
  if (converged_out) *converged_out = r.converged ? 1 : 0;
  return std::isfinite(r.rss) ? 0 : -2;
} catch (...) { return PRVH_E_INTERNAL; }

void prvh_fit_labels(const double params[4], double max_psnr, int gap_out[11], int gradient_out[20]) {
  LognormalCDF f;
  f.y0 = params[0]; f.A = params[1]; f.xc = params[2]; f.w = params[3];
  const Labels L = make_labels(f, max_psnr);
  if (gap_out) std::copy(L.gap, L.gap + 11, gap_out);
  if (gradient_out) std::copy(L.gradient, L.gradient + 20, gradient_out);
}

int prvh_write_label(const char* path, const double params[4], int converged, double max_psnr) try {
  if (!path || !params) return -1;
  FitResult r;
  r.f.y0 = params[0]; r.f.A = params[1]; r.f.xc = params[2]; r.f.w = params[3];
  r.converged = converged != 0;
  return write_label_file(path, r, max_psnr) ? 0 : -3;
} catch (...) { return PRVH_E_INTERNAL; }

prvh_share_data* prvh_share_data_create(const char* yaml, const char* name, int num_of_views, int id_of_batch, int method) try {
  auto sd = std::make_shared<Share_Data>(yaml ? yaml : "", name ? name : "", num_of_views, id_of_batch, method);
  if (!sd->ok) {
    g_error = sd->error;
    return nullptr;
  }
  auto* h = new prvh_share_data();
  h->sd = sd;
  return h;
} catch (...) { return nullptr; }
void prvh_share_data_destroy(prvh_share_data* h) { delete h; }
const char* prvh_share_data_error(void) { return g_error.c_str(); }

const char* prvh_share_data_string(const prvh_share_data* h, const char* f) {
  if (!h || !f) return "";
  const Share_Data& s = *h->sd;
  const std::string k = f;
  const std::string* v = nullptr;
  if (k == "pre_path") v = &s.pre_path; else if (k == "model_path") v = &s.model_path;
  else if (k == "viewspace_path") v = &s.viewspace_path; else if (k == "instant_ngp_path") v = &s.instant_ngp_path;
  else if (k == "name_of_pcd") v = &s.name_of_pcd; else if (k == "gt_path") v = &s.gt_path;
  else if (k == "save_path") v = &s.save_path; else if (k == "yaml_file_path") v = &s.yaml_file_path;
  else if (k == "shape_net") v = &s.shape_net; else if (k == "pvb_path") v = &s.pvb_path;
  else if (k == "orginalviews_path") v = &s.orginalviews_path;
  return v ? v->c_str() : "";
}

double prvh_share_data_number(const prvh_share_data* h, const char* f) try {
  if (!h || !f) return 0;
  const Share_Data& s = *h->sd;
  const std::string k = f;
#define NUM(name) if (k == #name) return (double)s.name;
  NUM(num_of_views) NUM(method_of_IG) NUM(num_of_thread) NUM(n_steps) NUM(evaluate) NUM(ensemble_num)
  NUM(num_of_max_iteration) NUM(id_of_batch) NUM(is_shape_net) NUM(coverage_view_num_max) NUM(coverage_view_num_add)
  NUM(ray_casting_aabb_scale) NUM(num_of_novel_test_views) NUM(view_space_radius) NUM(octomap_resolution)
  NUM(ground_truth_resolution) NUM(depth_scale) NUM(pt_norm) NUM(render_width) NUM(render_height)
  NUM(samples_per_ray) NUM(screenshot_spp) NUM(candidate_divisor) NUM(min_transmittance) NUM(cost_on) NUM(cost_rate) NUM(show)
  NUM(score_from_pngs)
#undef NUM
  return 0;
} catch (...) { return -1.0; }

int prvh_share_data_views(const prvh_share_data* h, double* out) try {
  if (!h) return -1;
  const auto& p = h->sd->pt_sphere;
  if (out)
    for (size_t i = 0; i < p.size(); i++)
      for (int j = 0; j < 3; j++) out[i * 3 + j] = p[i][j];
  return (int)p.size();
} catch (...) { return PRVH_E_INTERNAL; }

void prvh_share_data_intrinsics(const prvh_share_data* h, prvh_intrinsics* o) {
  if (!h || !o) return;
  const rs2_intrinsics& K = h->sd->color_intrinsics;
  o->width = K.width; o->height = K.height;
  o->ppx = K.ppx; o->ppy = K.ppy; o->fx = K.fx; o->fy = K.fy;
  for (int i = 0; i < 5; i++) o->coeffs[i] = K.coeffs[i];
}

int prvh_nbv_loop(prvh_share_data* h, const double center[3], double predicted_size, int first_view_id, int test_id,
                  prvh_score_fn score, void* user, prvh_loop_result* out) try {
  if (!h || !center || !out) return -1;
  Scorer s = [score, user](int method, int iteration, const std::string& scene, const std::string& render,
                           const std::vector<int>& ids, std::vector<double>& scores) -> int {
    if (!score) return -12;
    return score(user, method, iteration, scene.c_str(), render.c_str(), ids.data(), (int)ids.size(), scores.data());
  };
  NBV_Net_Labeler labeler(h->sd, Vec3(center[0], center[1], center[2]), predicted_size, s);
  const int rc = labeler.nbv_loop(first_view_id, test_id);
  out->n_chosen = (int)std::min<size_t>(labeler.chosen_nbvs.size(), 1024);
  for (int i = 0; i < out->n_chosen; i++) out->chosen[i] = labeler.chosen_nbvs[i];
  out->total_movement = labeler.total_movement_cost;
  return rc;
} catch (...) { return PRVH_E_INTERNAL; }


int prvh_method_in_scope(int method_of_IG) { return NBV_Net_Labeler::method_in_scope(method_of_IG) ? 1 : 0; }

int prvh_member_owner(int object, int member, int n_members, int world) {
  if (object < 0 || member < 0 || n_members < 1 || member >= n_members || world < 1) return -1;
  return member_pair_owner(object, member, n_members, world);
}

// deprecated entry points of the out-of-scope methods (include/prv_host.h): present, inert
long long prvh_pcd_read(const char*, float*, uint8_t*, long long) { return PRVH_E_UNSUPPORTED; }
int prvh_nbv_loop_budget(prvh_share_data*, const double*, double, int, int, prvh_score_fn, void*, int, prvh_loop_result* out) {
  if (out) out->n_chosen = 0;
  return PRVH_E_UNSUPPORTED;
}

static void put_err(char* err, int cap, const std::string& m) {
  if (err && cap > 0) snprintf(err, (size_t)cap, "%s", m.c_str());
}

int prvh_ingp_read(const char* path, prv_field_desc* desc, uint64_t* n_table, uint64_t* n_occ, uint16_t* table, uint16_t* mlp,
                   uint32_t* occ, char* err, int err_cap) try {
  if (!path) return -1;
  prvingp::Field f;
  std::string m;
  const int rc = prvingp::read_snapshot(path, f, m);
  if (rc != 0) {
    put_err(err, err_cap, m);
    return rc;
  }
  if (desc) *desc = f.desc;
  if (n_table) *n_table = f.table.size();
  if (n_occ) *n_occ = f.occ.size();
  if (table) memcpy(table, f.table.data(), f.table.size() * 2);
  if (mlp) memcpy(mlp, f.mlp.data(), f.mlp.size() * 2);
  if (occ) memcpy(occ, f.occ.data(), f.occ.size() * 4);
  return 0;
} catch (...) { return PRVH_E_INTERNAL; }

int prvh_ingp_write(const char* path, const prv_field_desc* desc, const uint16_t* table, uint64_t n_table, const uint16_t* mlp,
                    const uint32_t* occ, uint64_t n_occ, char* err, int err_cap) try {
  if (!path || !desc || !table || !mlp || !occ) return -1;
  prvingp::Field f;
  f.desc = *desc;
  f.table.assign(table, table + n_table);
  f.mlp.assign(mlp, mlp + prvingp::kMlpHalfs);
  f.occ.assign(occ, occ + n_occ);
  std::string m;
  const int rc = prvingp::write_snapshot(path, f, m);
  if (rc != 0) put_err(err, err_cap, m);
  return rc;
} catch (...) { return PRVH_E_INTERNAL; }

prvh_star* prvh_star_open(int rank, int world, const char* addr, int port, double timeout_s) try {
  std::unique_ptr<prvh_star> h(new prvh_star());
  if (!h->star.open(rank, world, addr ? addr : "", port, timeout_s > 0 ? timeout_s : 120.0)) {
    g_error = h->star.error;
    return nullptr;
  }
  return h.release();
} catch (...) { return nullptr; }

void prvh_star_close(prvh_star* h) { delete h; }

int prvh_star_all_gather(prvh_star* h, const void* send, uint64_t bytes, void* recv) try {
  if (!h || !send || !recv) return -1;
  return h->star.all_gather(send, (size_t)bytes, recv) ? 0 : -3;
} catch (...) { return PRVH_E_INTERNAL; }

int prvh_star_broadcast(prvh_star* h, void* buf, uint64_t bytes, int root) try {
  if (!h || !buf) return -1;
  return h->star.broadcast(buf, (size_t)bytes, root) ? 0 : -3;
} catch (...) { return PRVH_E_INTERNAL; }

int prvh_star_barrier(prvh_star* h) try {
  if (!h) return -1;
  return h->star.barrier() ? 0 : -3;
} catch (...) { return PRVH_E_INTERNAL; }

} // extern "C"
