// planner.hpp -- NBV_Net_Labeler: the slice of PRV_simulation/main.cpp:596-2277 that drives
// the render boundary.  Kept: transforms.json emission (get_coverage header + frames,
// nbv_loop's per-iteration json/render_json), the boundary call (train_by_instantNGP, now an
// in-process call through a scorer callback instead of the file-flag handshake), the
// per-iteration bookkeeping files, arg-max selection.  Not kept (out of scope):
// point-cloud / OctoMap asset preparation, PCL rendering, PRVNet itself (method 4 takes its view budget
// from a callback or from the file PRVNet's server would have written).
#pragma once
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <ctime>
#include <functional>
#include <iostream>
#include <memory>
#include <random>
#include <set>
#include <string>
#include <thread>
#include <vector>

#include "../csrc/prv_json.hpp"
#include "Share_Data.hpp"
#include "View_Space.hpp"
#include "path_planner.hpp"
#include "png_io.hpp"

namespace prvhost {

using prvjson::Value;

// transform_matrix of one view: P * (now_cam * pose^-1) * diag(1,-1,-1,1)  (main.cpp:1626-1641).
// The reference recomputes the pose inside the 4x4 element loops (16x per view, quirk F); once
// is enough, the result is identical.
inline Mat4 view_transform_matrix(View& view, const Mat4& now_camera_pose_world, const Vec3& object_center_world) {
  view.get_next_camera_pos(now_camera_pose_world, object_center_world);
  Mat4 view_pose_world = now_camera_pose_world * view.pose.inverse();
  Mat4 pose = Mat4::Identity(); // x,y,z -> z,x,y
  pose.m = {0, 0, 1, 0, 1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 0, 1};
  Mat4 pose_1 = Mat4::Identity(); // x,y,z -> x,-y,-z
  pose_1.m = {1, 0, 0, 0, 0, -1, 0, 0, 0, 0, -1, 0, 0, 0, 0, 1};
  return pose * view_pose_world * pose_1;
}

inline Value matrix_json(const Mat4& m) {
  Value rows;
  for (int k = 0; k < 4; k++) {
    Value row;
    for (int l = 0; l < 4; l++) row.append(Value(m(k, l)));
    rows.append(row);
  }
  return rows;
}

// header of a scene json (main.cpp:1585-1602) or, with divisor > 0, of the candidate render
// json (main.cpp:1794-1811: fl, c, w, h divided, distortion zeroed)
inline Value transforms_header(const rs2_intrinsics& K, int aabb_scale, double predicted_size, const Vec3& center,
                               double divisor) {
  Value root;
  root["camera_angle_x"] = Value(2.0 * std::atan(0.5 * K.width / (double)K.fx));
  root["camera_angle_y"] = Value(2.0 * std::atan(0.5 * K.height / (double)K.fy));
  if (divisor > 0) {
    root["fl_x"] = Value((double)K.fx / divisor);
    root["fl_y"] = Value((double)K.fy / divisor);
    root["k1"] = Value(0); root["k2"] = Value(0); root["k3"] = Value(0); root["p1"] = Value(0); root["p2"] = Value(0);
    root["cx"] = Value((double)K.ppx / divisor);
    root["cy"] = Value((double)K.ppy / divisor);
    root["w"] = Value((double)K.width / divisor);
    root["h"] = Value((double)K.height / divisor);
  } else {
    root["fl_x"] = Value((double)K.fx);
    root["fl_y"] = Value((double)K.fy);
    root["k1"] = Value((double)K.coeffs[0]); root["k2"] = Value((double)K.coeffs[1]);
    root["k3"] = Value((double)K.coeffs[2]); root["p1"] = Value((double)K.coeffs[3]);
    root["p2"] = Value((double)K.coeffs[4]);
    root["cx"] = Value((double)K.ppx);
    root["cy"] = Value((double)K.ppy);
    root["w"] = Value(K.width);
    root["h"] = Value(K.height);
  }
  root["aabb_scale"] = Value(aabb_scale);
  root["scale"] = Value(0.5 / predicted_size);
  root["offset"][0] = Value(0.5 + center.z);
  root["offset"][1] = Value(0.5 + center.x);
  root["offset"][2] = Value(0.5 + center.y);
  return root;
}

// elapsed seconds.  The reference divides clock() by CLOCKS_PER_SEC (main.cpp:1660, 1703): on its platform
// (MSVC) clock() is wall time; on Linux it is the process CPU time (every HIP runtime thread counted), so
// the meaning -- elapsed seconds -- is kept, not the call
inline double now_seconds() {
  return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count();
}

inline bool write_text(const std::string& path, const std::string& text) {
  FILE* f = fopen(path.c_str(), "wb");
  if (!f) return false;
  fwrite(text.data(), 1, text.size(), f);
  fclose(f);
  return true;
}

// the render boundary as the loop sees it: score every candidate of this iteration
using Scorer = std::function<int(int method, int iteration, const std::string& scene_json,
                                 const std::string& render_json, const std::vector<int>& candidate_ids,
                                 std::vector<double>& scores)>;

// What train_by_instantNGP hands run.py (the command line composed at main.cpp:1666-1685), as a struct: the in-process
// engine behind the boundary receives exactly the reference's arguments.  Empty strings = flag absent.
struct RunPyArgs {
  int n_steps = 0;                   // --n_steps
  std::string scene;                 // --scene <json>
  std::string test_transforms;       // --test_transforms <json>      (evaluation branch)
  std::string save_metrics;          // --save_metrics <txt>          (evaluation branch)
  std::string screenshot_transforms; // --screenshot_transforms <json> (candidate branch)
  std::string screenshot_dir;        // --screenshot_dir <dir>/        (candidate branch)
  int ensemble_id = -1;              // not a run.py flag: which member this call trains (the reference's members differ
                                     // by the engine's nondeterminism only; here member e has its own seed)
};
// the engine: train --n_steps on --scene, then evaluate or write the screenshots; 0 on success
using Engine = std::function<int(const RunPyArgs&)>;

// the final evaluation of `evaluate: 1` (main.cpp:1954-1965: train_by_instantNGP(<it>, "100", true) -> run.py's
// --test_transforms / --save_metrics): train on the scene json's views, score against the test view set
using EvalFn = std::function<int(const std::string& scene_json, double* mean_psnr, double* mean_ssim)>;

// The reference's score loops over the PNGs its engine left (main.cpp:2045-2094 EnsembleRGB, 2105-2158
// EnsembleRGBDensity): one unchosen view's E member images -> view_uncertainty, in the reference's own operation order.
// cv::imread(..., IMREAD_UNCHANGED) returns BGRA, so the reference's "r, g, b" are the PNG's B, G, R -- bytes 2, 1, 0 of
// an RGBA pixel -- and the order matters for the last bit of the sum.  Per pixel and channel: mean = (sum over members of
// the byte) / E, variance = (sum of (byte - mean)^2) / E, doubles, members in ascending order; method 2 adds
// ln(variance) where variance > 1e-10 (channel by channel into the running sum); method 3 adds (v0 + v1 + v2) / 3 and
// then (1 - mean alpha/255)^2, as two additions.  0, or -18 (a file is missing, unreadable or of another size).
inline int ensemble_uncertainty_from_pngs(int method, const std::vector<std::string>& files, double* out) {
  const int E = (int)files.size();
  if (E < 1 || (method != EnsembleRGB && method != EnsembleRGBDensity) || !out) return -18;
  std::vector<std::vector<uint8_t>> img((size_t)E);
  int w = 0, h = 0;
  for (int e = 0; e < E; e++) {
    int iw = 0, ih = 0;
    if (png_read_rgba8(files[(size_t)e], &iw, &ih, img[(size_t)e]) != 0 || (e > 0 && (iw != w || ih != h))) return -18;
    w = iw;
    h = ih;
  }
  static const int channel_order[3] = {2, 1, 0}; // the reference's rgba[0], rgba[1], rgba[2] of a BGRA pixel
  double score = 0.0;
  for (size_t p = 0; p < (size_t)w * (size_t)h; p++) {
    double var[3];
    for (int c = 0; c < 3; c++) {
      const size_t at = p * 4 + (size_t)channel_order[c];
      double mean = 0.0, v = 0.0;
      for (int e = 0; e < E; e++) mean += img[(size_t)e][at];
      mean /= E;
      for (int e = 0; e < E; e++) v += (img[(size_t)e][at] - mean) * (img[(size_t)e][at] - mean);
      var[c] = v / E;
    }
    if (method == EnsembleRGB) {
      for (int c = 0; c < 3; c++)
        if (var[c] > 1e-10) score += std::log(var[c]);
    } else {
      double opacity_mean = 0.0;
      for (int e = 0; e < E; e++) opacity_mean += img[(size_t)e][p * 4 + 3] / 255.0;
      opacity_mean /= E;
      score += (var[0] + var[1] + var[2]) / 3.0;
      score += (1.0 - opacity_mean) * (1.0 - opacity_mean);
    }
  }
  *out = score;
  return 0;
}

// `shard: members` (prv_planner): the member trainings of a lockstep round, n_objects x n_members
// (object, member) pairs, are dealt to the ranks round-robin -- pair p = object * n_members + member is trained by rank
// p % world (the reference trains them one run.py after another, main.cpp:2041-2043, 2101-2103)
inline int member_pair_owner(int object, int member, int n_members, int world) {
  return world > 0 ? (int)(((long long)object * n_members + member) % world) : 0;
}

class NBV_Net_Labeler {
public:
  std::shared_ptr<Share_Data> share_data;
  std::shared_ptr<View_Space> view_space;
  Scorer scorer;
  Engine engine;      // the in-process run.py behind train_by_instantNGP's reference signature
  EvalFn evaluator;   // empty: `evaluate: 1` is ignored
  std::vector<int> chosen_nbvs;
  std::vector<double> last_scores;
  // called after every scoring iteration with (iteration, unchosen view ids, their scores): logging / test dumps live in
  // the shell that installs it, not in the loop
  std::function<void(int, const std::vector<int>&, const std::vector<double>&)> on_scores;
  double total_movement_cost = 0.0;
  double final_psnr = -1.0, final_ssim = -1.0; // the final evaluation's metrics, when it ran

  NBV_Net_Labeler(const std::shared_ptr<Share_Data>& sd, const Vec3& center, double predicted_size, Scorer s)
      : share_data(sd), scorer(std::move(s)) {
    view_space = std::make_shared<View_Space>(share_data);
    view_space->set_view_space(center, predicted_size);
  }

  // scene json of the full candidate set, as get_coverage writes <gt_path>/<N>.json (main.cpp:1581-1651)
  // (the rgbaClip images themselves: prv_splat_points on the coloured cloud, or renders of a ground-truth field)
  int get_coverage() {
    Value root = transforms_header(share_data->color_intrinsics, share_data->ray_casting_aabb_scale,
                                   view_space->predicted_size, view_space->object_center_world, 0);
    const std::string n = std::to_string(share_data->num_of_views);
    for (size_t i = 0; i < view_space->views.size(); i++) {
      Value view_image;
      view_image["file_path"] = Value(n + "/rgbaClip_" + std::to_string(i) + ".png");
      view_image["transform_matrix"] =
          matrix_json(view_transform_matrix(view_space->views[i], view_space->now_camera_pose_world, view_space->object_center_world));
      root["frames"].append(view_image);
    }
    share_data->access_directory(share_data->gt_path);
    return write_text(share_data->gt_path + "/" + n + ".json", prvjson::to_styled_string(root)) ? 0 : -1;
  }

  // THE BOUNDARY, with the reference's own signature and branch structure (main.cpp:1658-1715).  The reference composes a
  // `python run.py ...` command line, writes interact/run_with_c++.py + ready_c++.txt and polls ready_py.txt once a
  // second; here the same arguments go to `engine` in process (the C ABI of include/prv.h behind it).  A maintainer
  // of the reference keeps every call site (main.cpp:1956, 2042, 2102, 2481) and swaps only this body: INTEGRATION.md.
  //   nbv_test == false              : scene <gt_path>/<train>.json, test <gt_path>/<test>.json -> <gt_path>/<train>.txt
  //   nbv_test, ensemble_id == -1    : scene <save>/json/<it>.json, test <gt_path>/<test>.json  -> <save>/metrics/<it>.txt
  //                                    (+ <save>/train_time/<it>.txt, :1707-1711)
  //   nbv_test, ensemble_id == e >= 0: scene <save>/json/<it>.json, screenshots of <save>/render_json/<it>.json ->
  //                                    <save>/render/<it>/ensemble_<e>/rgbaClip_<v>.png
  int train_by_instantNGP(std::string trian_json_file, std::string test_json_file = "100", bool nbv_test = false,
                          int ensemble_id = -1) {
    if (!engine) return -17;
    const double t0 = now_seconds();
    RunPyArgs a;
    a.n_steps = share_data->n_steps;
    a.ensemble_id = ensemble_id;
    if (!nbv_test) {
      a.scene = share_data->gt_path + "/" + trian_json_file + ".json";
      a.test_transforms = share_data->gt_path + "/" + test_json_file + ".json";
      a.save_metrics = share_data->gt_path + "/" + trian_json_file + ".txt";
    } else {
      a.scene = share_data->save_path + "/json/" + trian_json_file + ".json";
      if (ensemble_id == -1) {
        a.test_transforms = share_data->gt_path + "/" + test_json_file + ".json";
        a.save_metrics = share_data->save_path + "/metrics/" + trian_json_file + ".txt";
      } else {
        a.screenshot_transforms = share_data->save_path + "/render_json/" + trian_json_file + ".json";
        a.screenshot_dir = share_data->save_path + "/render/" + trian_json_file + "/ensemble_" + std::to_string(ensemble_id) + "/";
      }
    }
    const int rc = engine(a);
    const double cost_time = now_seconds() - t0;
    std::cout << "train and eval with executed time " << cost_time << " s." << std::endl; // :1705
    if (nbv_test && ensemble_id == -1)
      write_text(share_data->save_path + "/train_time/" + trian_json_file + ".txt", std::to_string(cost_time) + "\n"); // :1707-1711
    return rc;
  }

  // The fused form of the same boundary (`score_path: fused`, the default): every member trained and every candidate of
  // the iteration rendered + reduced on the device in ONE call; nothing but 16-byte records leaves the GPU.
  int score_candidates(const std::string& trian_json_file, const std::vector<int>& candidate_ids, std::vector<double>& scores) {
    const double t0 = now_seconds();
    const std::string scene = share_data->save_path + "/json/" + trian_json_file + ".json";
    const std::string render = share_data->save_path + "/render_json/" + trian_json_file + ".json";
    const int rc = scorer(share_data->method_of_IG, std::atoi(trian_json_file.c_str()), scene, render, candidate_ids, scores);
    const double cost_time = now_seconds() - t0;
    write_text(share_data->save_path + "/train_time/" + trian_json_file + ".txt", std::to_string(cost_time) + "\n");
    return rc;
  }

  // `score_path: png`: the score of one unchosen view from the PNGs the engine calls of this iteration left
  int view_uncertainty_from_pngs(int method, const std::string& iteration, int view_id, double* out) const {
    std::vector<std::string> files;
    for (int e = 0; e < share_data->ensemble_num; e++)
      files.push_back(share_data->save_path + "/render/" + iteration + "/ensemble_" + std::to_string(e) + "/rgbaClip_" +
                      std::to_string(view_id) + ".png");
    return ensemble_uncertainty_from_pngs(method, files, out);
  }

  // Methods 1 (RandomOneshot, main.cpp:1981-2037) and 4 (PVBCoverage, :2163-2242: PRVNet's view budget) never render and
  // are outside this build's scope: refused BEFORE the loop writes anything, with a message.
  static bool method_in_scope(int method) {
    return method == RandomIterative || method == EnsembleRGB || method == EnsembleRGBDensity || method == PSNRCoverage;
  }

  // main.cpp:1718-2277 for methods 0, 2, 3 (and 5, this build's single-model score); chosen views in `chosen_nbvs`.
  // The loop is three resumable pieces -- nbv_begin (everything before the reference's `while (true)`), nbv_prepare (an
  // iteration's json / render_json and the termination test, :1885-1966) and nbv_decide (score, pick, movement cost,
  // :1969-2264) -- so that a shell can walk SEVERAL objects' loops in lockstep (prv_planner `shard: members`: the
  // (object, member) trainings of a round are dealt to the ranks before any object of the round is scored).
  int nbv_loop(int first_view_id = -1, int test_id = 0) {
    int rc = nbv_begin(first_view_id, test_id);
    if (rc != 0) return rc < 0 ? rc : 0; // 1: a finished run was found on disk (idempotent resume)
    while ((rc = nbv_prepare()) == 0)
      if ((rc = nbv_decide()) != 0) return rc;
    return rc < 0 ? rc : 0;
  }

  // state of a loop in progress (nbv_begin .. the nbv_prepare that returns 1)
  struct NbvRun {
    Value root_nbvs, root_render;
    std::set<int> chosen_nbvs_set;
    std::mt19937 rng{12345}; // the reference seeds rand() with clock() (Share_Data.hpp:514): unreproducible by design
    double loop_t0 = 0.0;
    int iteration = 0;
    std::vector<int> candidates; // the unchosen views of the iteration in preparation / being decided
    std::string scene_json, render_json; // its <save>/json/<it>.json and <save>/render_json/<it>.json
  } run;

  // -> 0: the loop is set up; 1: nothing to do (a finished run on disk); < 0: error
  int nbv_begin(int first_view_id = -1, int test_id = 0) {
    if (first_view_id == -1) first_view_id = 0; // :1725-1728
    Share_Data& sd = *share_data;
    if (!method_in_scope(sd.method_of_IG)) {
      std::cerr << "nbv_loop: method_of_IG " << sd.method_of_IG << " is not built (RandomOneshot = 1 and PVBCoverage = 4 belong to the "
                   "reference's PRVNet pipeline, outside this build's scope); nothing was written" << std::endl;
      return -10;
    }
    { // :1735-1747: the methods run with the view budget a method-4 run of the REFERENCE left behind, when there is one
      std::ifstream fin(sd.pre_path + "Compare/ShapeNet/" + sd.name_of_pcd + "_m4_v1_t" + std::to_string(test_id) + "/view_budget.txt");
      int view_budget = 0;
      if (fin.is_open() && (fin >> view_budget) && view_budget > 0) sd.num_of_max_iteration = view_budget - 1;
    }
    sd.save_path += "_v1";                     // one initial view (init_view_ids.size() == 1, :1751)
    sd.save_path += "_t" + std::to_string(test_id);
    for (const char* sub : {"/json", "/render_json", "/metrics", "/render", "/train_time", "/infer_time", "/movement"})
      sd.access_directory(sd.save_path + sub); // :1753-1759
    { // idempotent resume: finished runs are skipped (:1761-1770)
      std::ifstream check(sd.save_path + "/run_time.txt");
      double run_time = -1;
      if (check.is_open() && (check >> run_time) && run_time >= 0) return 1;
    }
    run = NbvRun{};
    run.root_nbvs = transforms_header(sd.color_intrinsics, sd.ray_casting_aabb_scale, view_space->predicted_size,
                                      view_space->object_center_world, 0);
    run.root_render = transforms_header(sd.color_intrinsics, sd.ray_casting_aabb_scale, view_space->predicted_size,
                                        view_space->object_center_world, sd.candidate_divisor);
    write_text(sd.save_path + "/movement/-1.txt", std::to_string(first_view_id) + "\t0\t0\n"); // :1868-1870
    chosen_nbvs.assign(1, first_view_id);
    total_movement_cost = 0.0; // :1867
    run.chosen_nbvs_set = {first_view_id};
    run.loop_t0 = now_seconds();
    run.iteration = 0;
    return 0;
  }

  // the iteration's two json files; -> 0: a view is to be chosen (nbv_decide), 1: the loop has ended (run_time.txt and
  // the final evaluation are written), < 0: error
  int nbv_prepare() {
    Share_Data& sd = *share_data;
    const int n_views = (int)view_space->views.size();
    const std::string prefix = "../../../../Coverage_images/ShapeNet/" + sd.name_of_pcd + "/" + std::to_string(sd.num_of_views) + "/rgbaClip_";
    Value now_nbvs_json(run.root_nbvs), now_render_json(run.root_render);
    run.candidates.clear();
    for (int i = 0; i < n_views; i++) { // :1887-1916
      Value view_image;
      view_image["file_path"] = Value(prefix + std::to_string(i) + ".png");
      view_image["transform_matrix"] = matrix_json(
          view_transform_matrix(view_space->views[i], view_space->now_camera_pose_world, view_space->object_center_world));
      if (run.chosen_nbvs_set.count(i)) now_nbvs_json["frames"].append(view_image);
      else {
        now_render_json["frames"].append(view_image);
        run.candidates.push_back(i);
      }
    }
    const std::string it = std::to_string(run.iteration);
    run.scene_json = sd.save_path + "/json/" + it + ".json";
    run.render_json = sd.save_path + "/render_json/" + it + ".json";
    write_text(run.scene_json, prvjson::to_styled_string(now_nbvs_json));     // :1918-1920
    write_text(run.render_json, prvjson::to_styled_string(now_render_json)); // :1922-1924
    if (run.iteration == sd.num_of_max_iteration || run.candidates.empty()) { // :1946-1966
      const double loops_time = now_seconds() - run.loop_t0;
      write_text(sd.save_path + "/run_time.txt", std::to_string(loops_time) + "\n");
      if (sd.evaluate && evaluator) { // "final evaluating..." (:1954-1965): metrics/<it>.txt in run.py's format
        double psnr = 0, ssim = 0;
        const int rc = evaluator(run.scene_json, &psnr, &ssim);
        if (rc != 0) return rc < 0 ? rc : -12;
        char buf[128];
        snprintf(buf, sizeof(buf), "PSNR\t%.17g\nSSIM\t%.17g", psnr, ssim);
        write_text(sd.save_path + "/metrics/" + it + ".txt", buf);
        final_psnr = psnr;
        final_ssim = ssim;
      }
      return 1;
    }
    return 0;
  }

  // score the prepared iteration's candidates, keep the arg-max, account for the movement (:1969-2264); -> 0 or an error
  int nbv_decide() {
    Share_Data& sd = *share_data;
    const int n_views = (int)view_space->views.size();
    const std::vector<int>& candidates = run.candidates;
    const int iteration = run.iteration;
    const std::string it = std::to_string(iteration);
    const double infer_t0 = now_seconds();
    int next_view_id = -1;
    switch (sd.method_of_IG) {
      case RandomIterative: { // :1974-1979
        next_view_id = (int)(run.rng() % (unsigned)n_views);
        while (run.chosen_nbvs_set.count(next_view_id)) next_view_id = (int)(run.rng() % (unsigned)n_views);
        break;
      }
      case EnsembleRGB:
      case EnsembleRGBDensity:
      case PSNRCoverage: { // :2039-2161: score every unchosen view, keep the arg-max
        std::vector<double> scores(candidates.size(), 0.0);
        if (sd.score_from_pngs && sd.method_of_IG != PSNRCoverage) {
          // the reference's data flow, call for call: one engine run per member (:2041-2043, :2101-2103), then the PNGs
          // (no train_time/<it>.txt here: the reference writes it for ensemble_id == -1 only, :1707-1711)
          for (int ensemble_id = 0; ensemble_id < sd.ensemble_num; ensemble_id++) {
            const int rc = train_by_instantNGP(it, "100", true, ensemble_id);
            if (rc != 0) return rc;
          }
          for (size_t k = 0; k < candidates.size(); k++) {
            const int rc = view_uncertainty_from_pngs(sd.method_of_IG, it, candidates[k], &scores[k]);
            if (rc != 0) return rc;
          }
        } else {
          const int rc = score_candidates(it, candidates, scores);
          if (rc != 0) return rc;
        }
        last_scores = scores;
        if (on_scores) on_scores(iteration, candidates, scores); // an observer the shell may install (prv_planner: dump_scores)
        double largest_view_uncertainty = -1e100; // :1971
        int best_view_id = -1;
        for (size_t k = 0; k < candidates.size(); k++) // ascending ids, strict '>' (:2088-2091)
          if (scores[k] > largest_view_uncertainty) {
            largest_view_uncertainty = scores[k];
            best_view_id = candidates[k];
          }
        next_view_id = best_view_id;
        break;
      }
      default: // unreachable: method_in_scope() was checked before anything was written
        return -10;
    }
    if (next_view_id < 0) return -11;
    chosen_nbvs.push_back(next_view_id); // :2246-2247
    run.chosen_nbvs_set.insert(next_view_id);
    write_text(sd.save_path + "/infer_time/" + it + ".txt",
               std::to_string(now_seconds() - infer_t0) + "\n"); // :2250-2253
    // movement cost: view id \t local path \t running total (:2256-2264)
    const auto local_path = get_local_path(view_space->views[chosen_nbvs[iteration]].init_pos,
                                           view_space->views[next_view_id].init_pos,
                                           view_space->object_center_world + Vec3(1e-10, 1e-10, 1e-10), view_space->predicted_size);
    total_movement_cost += local_path.second;
    write_text(sd.save_path + "/movement/" + it + ".txt",
               std::to_string(next_view_id) + "\t" + std::to_string(local_path.second) + "\t" + std::to_string(total_movement_cost) + "\n");
    run.iteration++;
    return 0;
  }
};

} // namespace prvhost
