// planner.hpp -- NBV_Net_Labeler: the slice of PRV_simulation/main.cpp:596-2277 that drives
// the render boundary.  Kept: transforms.json emission (get_coverage header + frames,
// nbv_loop's per-iteration json/render_json), the boundary call (train_by_instantNGP, now an
// in-process call through a scorer callback instead of the file-flag handshake), the
// per-iteration bookkeeping files, arg-max selection.  Not kept (out of scope):
// point-cloud / OctoMap asset preparation, PCL rendering, Gurobi path planning, methods 1 and 4.
#pragma once
#include <cstdio>
#include <ctime>
#include <functional>
#include <memory>
#include <random>
#include <set>
#include <string>
#include <vector>

#include "../csrc/prv_json.hpp"
#include "Share_Data.hpp"
#include "View_Space.hpp"
#include "path_planner.hpp"

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

class NBV_Net_Labeler {
public:
  std::shared_ptr<Share_Data> share_data;
  std::shared_ptr<View_Space> view_space;
  Scorer scorer;
  std::vector<int> chosen_nbvs;
  std::vector<double> last_scores;
  double total_movement_cost = 0.0;

  NBV_Net_Labeler(const std::shared_ptr<Share_Data>& sd, const Vec3& center, double predicted_size, Scorer s)
      : share_data(sd), scorer(std::move(s)) {
    view_space = std::make_shared<View_Space>(share_data);
    view_space->set_view_space(center, predicted_size);
  }

  // scene json of the full candidate set, as get_coverage writes <gt_path>/<N>.json (main.cpp:1581-1651)
  // minus the PCL/OpenGL ground-truth screenshots (out of scope)
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

  // The boundary (main.cpp:1658-1715).  Reference: compose a run.py command line, write
  // interact/run_with_c++.py + ready_c++.txt, poll ready_py.txt.  Here: one in-process call.
  // Only the nbv_test / ensemble branch (the candidate-scoring use, :1676-1684) is on the path.
  int train_by_instantNGP(const std::string& trian_json_file, const std::vector<int>& candidate_ids,
                          std::vector<double>& scores) {
    const std::clock_t t0 = std::clock();
    const std::string scene = share_data->save_path + "/json/" + trian_json_file + ".json";
    const std::string render = share_data->save_path + "/render_json/" + trian_json_file + ".json";
    const int rc = scorer(share_data->method_of_IG, std::atoi(trian_json_file.c_str()), scene, render, candidate_ids, scores);
    const double cost_time = double(std::clock() - t0) / CLOCKS_PER_SEC;
    std::string t = std::to_string(cost_time) + "\n";
    write_text(share_data->save_path + "/train_time/" + trian_json_file + ".txt", t); // :1708-1710
    return rc;
  }

  // main.cpp:1718-2277, methods 0 (random), 2, 3 and 5; chosen views in `chosen_nbvs`
  int nbv_loop(int first_view_id = -1, int test_id = 0) {
    if (first_view_id == -1) first_view_id = 0; // :1725-1728
    Share_Data& sd = *share_data;
    sd.save_path += "_v1";                     // one initial view (init_view_ids.size() == 1, :1751)
    sd.save_path += "_t" + std::to_string(test_id);
    for (const char* sub : {"/json", "/render_json", "/metrics", "/render", "/train_time", "/infer_time", "/movement"})
      sd.access_directory(sd.save_path + sub); // :1753-1759
    { // idempotent resume: finished runs are skipped (:1761-1770)
      std::ifstream check(sd.save_path + "/run_time.txt");
      double run_time = -1;
      if (check.is_open() && (check >> run_time) && run_time >= 0) return 0;
    }
    const Value root_nbvs = transforms_header(sd.color_intrinsics, sd.ray_casting_aabb_scale, view_space->predicted_size,
                                              view_space->object_center_world, 0);
    const Value root_render = transforms_header(sd.color_intrinsics, sd.ray_casting_aabb_scale, view_space->predicted_size,
                                                view_space->object_center_world, sd.candidate_divisor);
    write_text(sd.save_path + "/movement/-1.txt", std::to_string(first_view_id) + "\t0\t0\n"); // :1868-1870
    chosen_nbvs.assign(1, first_view_id);
    total_movement_cost = 0.0; // :1867
    std::set<int> chosen_nbvs_set{first_view_id};
    std::mt19937 rng(12345); // the reference seeds rand() with clock() (Share_Data.hpp:514): unreproducible by design
    const std::clock_t loop_t0 = std::clock();
    const int n_views = (int)view_space->views.size();
    const std::string prefix = "../../../../Coverage_images/ShapeNet/" + sd.name_of_pcd + "/" + std::to_string(sd.num_of_views) + "/rgbaClip_";
    int iteration = 0;
    while (true) {
      Value now_nbvs_json(root_nbvs), now_render_json(root_render);
      std::vector<int> candidates;
      for (int i = 0; i < n_views; i++) { // :1887-1916
        Value view_image;
        view_image["file_path"] = Value(prefix + std::to_string(i) + ".png");
        view_image["transform_matrix"] = matrix_json(
            view_transform_matrix(view_space->views[i], view_space->now_camera_pose_world, view_space->object_center_world));
        if (chosen_nbvs_set.count(i)) now_nbvs_json["frames"].append(view_image);
        else {
          now_render_json["frames"].append(view_image);
          candidates.push_back(i);
        }
      }
      const std::string it = std::to_string(iteration);
      write_text(sd.save_path + "/json/" + it + ".json", prvjson::to_styled_string(now_nbvs_json));          // :1918-1920
      write_text(sd.save_path + "/render_json/" + it + ".json", prvjson::to_styled_string(now_render_json)); // :1922-1924
      if (iteration == sd.num_of_max_iteration || candidates.empty()) { // :1946-1966
        const double loops_time = double(std::clock() - loop_t0) / CLOCKS_PER_SEC;
        write_text(sd.save_path + "/run_time.txt", std::to_string(loops_time) + "\n");
        break;
      }
      const std::clock_t infer_t0 = std::clock();
      int next_view_id = -1;
      switch (sd.method_of_IG) {
        case RandomIterative: { // :1974-1979
          next_view_id = (int)(rng() % (unsigned)n_views);
          while (chosen_nbvs_set.count(next_view_id)) next_view_id = (int)(rng() % (unsigned)n_views);
          break;
        }
        case EnsembleRGB:
        case EnsembleRGBDensity:
        case PSNRCoverage: { // :2039-2161: score every unchosen view, keep the arg-max
          std::vector<double> scores(candidates.size(), 0.0);
          const int rc = train_by_instantNGP(it, candidates, scores);
          if (rc != 0) return rc;
          last_scores = scores;
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
        default:
          return -10; // RandomOneshot / PVBCoverage need the TSP planner / PRVNet: out of scope
      }
      if (next_view_id < 0) return -11;
      chosen_nbvs.push_back(next_view_id); // :2246-2247
      chosen_nbvs_set.insert(next_view_id);
      write_text(sd.save_path + "/infer_time/" + it + ".txt",
                 std::to_string(double(std::clock() - infer_t0) / CLOCKS_PER_SEC) + "\n"); // :2250-2253
      // movement cost: view id \t local path \t running total (:2256-2264)
      const auto local_path = get_local_path(view_space->views[chosen_nbvs[iteration]].init_pos,
                                             view_space->views[next_view_id].init_pos,
                                             view_space->object_center_world + Vec3(1e-10, 1e-10, 1e-10), view_space->predicted_size);
      total_movement_cost += local_path.second;
      write_text(sd.save_path + "/movement/" + it + ".txt",
                 std::to_string(next_view_id) + "\t" + std::to_string(local_path.second) + "\t" + std::to_string(total_movement_cost) + "\n");
      iteration++;
    }
    return 0;
  }
};

} // namespace prvhost
