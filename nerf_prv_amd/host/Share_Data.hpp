// Share_Data.hpp -- configuration + shared planner state.  Same class name, field names and
// constructor signature as the reference's PRV_simulation/Share_Data.hpp:204-537, reduced to
// what the render/score path and its planner loop read.  The config file keeps the
// reference's OpenCV-FileStorage YAML dialect ("%YAML:1.0", flat "key: value"); it is read
// by a small own parser because OpenCV is not a dependency here.  Keys that are absent
// read as 0 / "" exactly like cv::FileNode >> does.
#pragma once
#include <sys/stat.h>

#include <cmath>
#include <fstream>
#include <map>
#include <sstream>
#include <string>
#include <vector>

namespace prvhost {

// method_of_IG values (Share_Data.hpp:198-202) + the single-model score of this build
enum { RandomIterative = 0, RandomOneshot = 1, EnsembleRGB = 2, EnsembleRGBDensity = 3, PVBCoverage = 4, PSNRCoverage = 5 };

struct rs2_intrinsics { // Share_Data.hpp:79-89 (float fields, like librealsense)
  int width = 0, height = 0;
  float ppx = 0, ppy = 0, fx = 0, fy = 0;
  int model = 0;
  float coeffs[5] = {0, 0, 0, 0, 0};
};

// flat "key: value" reader for the OpenCV YAML dialect
class FileStorage {
public:
  bool open(const std::string& path) {
    std::ifstream f(path);
    if (!f.is_open()) return false;
    std::string line;
    while (std::getline(f, line)) {
      const size_t hash = line.find('#');
      if (line.rfind("%YAML", 0) == 0 || line.rfind("---", 0) == 0) continue;
      std::string s = hash == std::string::npos ? line : unquoted_prefix(line, hash);
      const size_t colon = s.find(':');
      if (colon == std::string::npos) continue;
      std::string k = trim(s.substr(0, colon)), v = trim(s.substr(colon + 1));
      if (k.empty()) continue;
      if (v.size() >= 2 && v.front() == '"' && v.back() == '"') v = v.substr(1, v.size() - 2);
      kv_[k] = v;
    }
    return true;
  }
  std::string str(const std::string& k) const {
    auto it = kv_.find(k);
    return it == kv_.end() ? std::string() : it->second;
  }
  double num(const std::string& k) const {
    auto it = kv_.find(k);
    if (it == kv_.end() || it->second.empty()) return 0.0;
    return std::strtod(it->second.c_str(), nullptr); // "0." and "1.0e-03" parse as OpenCV does
  }
  bool has(const std::string& k) const { return kv_.count(k) != 0; }

private:
  std::map<std::string, std::string> kv_;
  static std::string trim(const std::string& s) {
    size_t a = s.find_first_not_of(" \t\r\n"), b = s.find_last_not_of(" \t\r\n");
    return a == std::string::npos ? std::string() : s.substr(a, b - a + 1);
  }
  static std::string unquoted_prefix(const std::string& line, size_t hash) {
    // keep '#' inside a quoted value
    bool in_q = false;
    for (size_t i = 0; i < line.size(); i++) {
      if (line[i] == '"') in_q = !in_q;
      if (line[i] == '#' && !in_q) return line.substr(0, i);
    }
    (void)hash;
    return line;
  }
};

class Share_Data {
public:
  // paths
  std::string yaml_file_path, pre_path, model_path, viewspace_path, instant_ngp_path, orginalviews_path, pvb_path,
      shape_net, name_of_pcd, nbv_net_path;
  std::string gt_path, save_path;
  // planner settings
  int num_of_views = 0, method_of_IG = 0, num_of_thread = 0, n_steps = 0, evaluate = 0, ensemble_num = 0;
  int num_of_max_iteration = 0, id_of_batch = 0, is_shape_net = 0, coverage_view_num_max = 0, coverage_view_num_add = 0;
  int ray_casting_aabb_scale = 0, num_of_novel_test_views = 0, show = 0, cost_on = 0;
  double view_space_radius = 0, octomap_resolution = 0, ground_truth_resolution = 0, depth_scale = 0, cost_rate = 0;
  rs2_intrinsics color_intrinsics;
  // keys this build adds (absent from the reference's file; all optional)
  int render_width = 0, render_height = 0, samples_per_ray = 0, screenshot_spp = 0;
  double candidate_divisor = 0, min_transmittance = 0;
  // score_path: fused (default) = one device-resident scoring round per iteration (prv_score_views);
  //             png = the reference's own data flow: per member train_by_instantNGP(it, "100", true, e) writes
  //             render/<it>/ensemble_<e>/rgbaClip_<v>.png, then the planner's loops read the PNGs (main.cpp:2045-2094, 2105-2158)
  bool score_from_pngs = false;
  // state
  std::vector<std::vector<double>> pt_sphere;
  double pt_norm = 0;
  bool ok = false;
  std::string error;

  // Share_Data(config, name, num_of_views, id_of_batch, method)  (Share_Data.hpp:334)
  Share_Data(const std::string& _config_file_path, const std::string& test_name = "", int _num_of_views = -1,
             int _id_of_batch = -1, int test_method = -1) {
    yaml_file_path = _config_file_path;
    FileStorage fs;
    if (!fs.open(yaml_file_path)) {
      error = "cannot open " + yaml_file_path;
      return;
    }
    pre_path = fs.str("pre_path");
    model_path = fs.str("model_path");
    viewspace_path = fs.str("viewspace_path");
    instant_ngp_path = fs.str("instant_ngp_path");
    orginalviews_path = fs.str("orginalviews_path");
    pvb_path = fs.str("pvb_path");
    shape_net = fs.str("shape_net");
    name_of_pcd = fs.str("name_of_pcd");
    nbv_net_path = fs.str("nbv_net_path");
    method_of_IG = (int)fs.num("method_of_IG");
    num_of_thread = (int)fs.num("num_of_thread");
    octomap_resolution = fs.num("octomap_resolution");
    ground_truth_resolution = fs.num("ground_truth_resolution");
    is_shape_net = (int)fs.num("is_shape_net");
    n_steps = (int)fs.num("n_steps");
    id_of_batch = (int)fs.num("id_of_batch");
    evaluate = (int)fs.num("evaluate");
    ensemble_num = (int)fs.num("ensemble_num");
    cost_on = (int)fs.num("cost_on");
    cost_rate = fs.num("cost_rate");
    num_of_max_iteration = (int)fs.num("num_of_max_iteration");
    coverage_view_num_max = (int)fs.num("coverage_view_num_max");
    coverage_view_num_add = (int)fs.num("coverage_view_num_add");
    show = (int)fs.num("show");
    num_of_views = (int)fs.num("num_of_views");
    num_of_novel_test_views = (int)fs.num("num_of_novel_test_views");
    ray_casting_aabb_scale = (int)fs.num("ray_casting_aabb_scale");
    view_space_radius = fs.num("view_space_radius");
    color_intrinsics.width = (int)fs.num("color_width");
    color_intrinsics.height = (int)fs.num("color_height");
    color_intrinsics.fx = (float)fs.num("color_fx");
    color_intrinsics.fy = (float)fs.num("color_fy");
    color_intrinsics.ppx = (float)fs.num("color_ppx");
    color_intrinsics.ppy = (float)fs.num("color_ppy");
    color_intrinsics.model = (int)fs.num("color_model");
    color_intrinsics.coeffs[0] = (float)fs.num("color_k1"); // YAML order k1,k2,k3,p1,p2 (Share_Data.hpp:395-399)
    color_intrinsics.coeffs[1] = (float)fs.num("color_k2");
    color_intrinsics.coeffs[2] = (float)fs.num("color_k3");
    color_intrinsics.coeffs[3] = (float)fs.num("color_p1");
    color_intrinsics.coeffs[4] = (float)fs.num("color_p2");
    depth_scale = fs.num("depth_scale");
    render_width = (int)fs.num("render_width");
    render_height = (int)fs.num("render_height");
    // 0 (the default: the reference's yaml has no such key) = the engine's own stepping rule, what run.py:304 renders
    // with (PRV_STEP_NGP); N > 0 = N uniform samples per ray (PRV_STEP_FIXED_S, the BASELINE configs' rule)
    samples_per_ray = fs.has("samples_per_ray") ? (int)fs.num("samples_per_ray") : 0;
    screenshot_spp = fs.has("screenshot_spp") ? (int)fs.num("screenshot_spp") : 16; // run.py:48
    candidate_divisor = fs.has("candidate_divisor") ? fs.num("candidate_divisor") : 16.0; // main.cpp:1796
    min_transmittance = fs.has("min_transmittance") ? fs.num("min_transmittance") : 0.01;
    if (fs.has("score_path")) {
      const std::string how = fs.str("score_path");
      if (how != "fused" && how != "png") {
        error = "score_path must be 'fused' or 'png', not '" + how + "'";
        return;
      }
      score_from_pngs = how == "png";
    }
    // constructor overrides (Share_Data.hpp:402-409)
    if (test_name != "") name_of_pcd = test_name;
    if (test_method != -1) method_of_IG = test_method;
    if (_num_of_views != -1) num_of_views = _num_of_views;
    if (_id_of_batch != -1) id_of_batch = _id_of_batch;
    if (!is_shape_net) {
      coverage_view_num_max = 90;
      coverage_view_num_add = 1;
    }
    // derived paths (Share_Data.hpp:482-503)
    gt_path = pre_path + "Coverage_images/";
    save_path = pre_path + "Compare/";
    if (is_shape_net) {
      gt_path += "ShapeNet";
      save_path += "ShapeNet";
      if (id_of_batch >= 0) {
        gt_path += "_" + std::to_string(id_of_batch);
        save_path += "_" + std::to_string(id_of_batch);
      }
      gt_path += "/";
      save_path += "/";
    }
    gt_path += name_of_pcd;
    save_path += name_of_pcd;
    if (test_method != -1) save_path += "_m" + std::to_string(method_of_IG);
    if (method_of_IG == 2) ensemble_num = 2;      // the paper's values (Share_Data.hpp:505-510)
    else if (method_of_IG == 3) ensemble_num = 5;
    // view set: <viewspace_path><num_of_views>.txt, N rows of 3 numbers (Share_Data.hpp:517-528)
    if (num_of_views > 0) {
      std::ifstream fin_sphere(viewspace_path + std::to_string(num_of_views) + ".txt");
      pt_sphere.assign(num_of_views, std::vector<double>(3, 0.0));
      if (fin_sphere.is_open()) {
        for (int i = 0; i < num_of_views; i++)
          for (int j = 0; j < 3; j++) fin_sphere >> pt_sphere[i][j];
      } else {
        error = "cannot open view set " + viewspace_path + std::to_string(num_of_views) + ".txt";
        return;
      }
      pt_norm = std::sqrt(pt_sphere[0][0] * pt_sphere[0][0] + pt_sphere[0][1] * pt_sphere[0][1] + pt_sphere[0][2] * pt_sphere[0][2]);
    }
    ok = true;
  }

  // ranks > 0 of a views-sharded job run the same loop as rank 0 and must not write into rank 0's tree: every
  // output path moves from <pre_path> to <new_pre> (inputs -- model_path, viewspace_path -- stay)
  void relocate_outputs(const std::string& new_pre) {
    auto move = [&](std::string& p) {
      if (p.compare(0, pre_path.size(), pre_path) == 0) p = new_pre + p.substr(pre_path.size());
    };
    move(gt_path);
    move(save_path);
    pre_path = new_pre;
  }

  // create every directory level of cd (Share_Data.hpp:639-649, POSIX instead of <direct.h>)
  void access_directory(const std::string& cd) const {
    std::string temp;
    for (size_t i = 0; i < cd.length(); i++) {
      if (cd[i] == '/' && !temp.empty()) ::mkdir(temp.c_str(), 0777);
      temp += cd[i];
    }
    if (!temp.empty()) ::mkdir(temp.c_str(), 0777);
  }
};

} // namespace prvhost
