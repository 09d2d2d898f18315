// main.cpp -- prv_planner: the planner executable.  Mirrors the console protocol of the
// reference's main() (PRV_simulation/main.cpp:2294-2309): read a mode number, then object
// names until "-1", from stdin.  Mode 21 (ViewPlanning, main.cpp:3834-4004) is the one on
// the hot path; every render/score goes through the C ABI of include/prv.h in-process.
//
//   prv_planner [config.yaml]      (default "../DefaultConfiguration.yaml", main.cpp:2312)
// Mode 4 (InstantNGP, main.cpp:2463-2487) is here too: the PSNR-vs-#views curve files of an object.
//
// Ensemble members / the field under test come from model files
//   <model_path>/<object>/member_<e>.prvf  (prv_model_save_file format)
// or, when the config carries `synthetic_seed`, from the deterministic synthetic generator
// (member e = seed + e; reference images for method 5 from seed + 4096),
// or -- `train_steps: N` -- they are TRAINED in process every iteration, as the reference does through
// train_by_instantNGP (main.cpp:1658-1715, 2041-2043): a fresh field per member, N optimiser steps on the
// views chosen so far; the training images are rendered from the ground-truth field
// (`ground_truth_seed`, slot 6) with the dataset cameras of the iteration's json -- or, with `coverage_images: 1`
// and `train_images: files`, written once as <gt_path>/<N>/rgbaClip_<i>.png (what mode 3 leaves on disk) and read
// back through each iteration's json exactly as run.py's load_training_data does.
#include <cstdio>
#include <cstring>
#include <iostream>
#include <string>
#include <vector>

extern "C" {
#include "../../include/prv.h"
#include "../../include/prv_host.h"
}
#include "planner.hpp"
#include "fit_curve.hpp"
#include "extras/pcd_io.hpp" // mode 3 only: the coloured cloud the splat rasteriser turns into rgbaClip images (not exported)
#include "png_io.hpp"

using namespace prvhost;

namespace {

enum { GetCoverage = 3, InstantNGP = 4, GetPathPlan = 20, ViewPlanning = 21 }; // main.cpp:2284-2292

struct HipScorer {
  prv_ctx* ctx = nullptr;
  // views-sharded job (`shard: views` / PRV_SHARD=views under RANK / WORLD_SIZE): every rank runs the same loop on the
  // same object; the members are trained by their owners (e % world) and exchanged, the candidates of an iteration are
  // dealt to the ranks and ONE all-gather of the 16-byte records gives every rank every score
  prv_comm* comm = nullptr;
  int rank = 0, world = 1;
  // where the members live and who trains them.  Default (one object at a time): member e in model slot e, trained by rank
  // e % world.  `shard: members` walks several objects' loops in lockstep: object o's members sit in slots slot_base + e and
  // the (object, member) trainings of a round are dealt round-robin, member e to rank (pair_base + e) % world with
  // pair_base = o * n_members; the shell trains a round's pairs of ALL objects side by side before any object is scored
  // (members_pretrained: the scoring call then only brings the members together)
  int slot_base = 0, pair_base = 0;
  bool members_pretrained = false;
  int slot_of(int e) const { return slot_base + e; }
  int owner_of(int e) const { return member_pair_owner(0, pair_base + e, 1 << 30, world); } // (pair_base = object * n_members: planner.hpp's rule)
  std::shared_ptr<Share_Data> sd;
  int n_members = 1;
  float* gt_dev = nullptr; // method 5: reference images of ALL views at the candidate size
  int gt_w = 0, gt_h = 0;
  // training in the loop
  int train_steps = 0, train_rays = 0 /* 0: the library default */, train_w = 0, train_h = 0;
  bool dump_records = false; // view_planning sets it (yaml dump_scores: 1, or PRV_PLANNER_DUMP_RECORDS in the environment)
  int train_patch_w = 0, train_patch_h = 0; // yaml train_patch_w / train_patch_h: prv_train_opts.patch_w / patch_h (0: the library default, single pixels)
  int train_step_mode = -1; // yaml train_step_mode: prv_train_opts.step_mode (-1: the library default = the engine's marcher; 0: 128 uniform samples per ray, rounds 1-5)
  void apply_train_opts(prv_train_opts& to) const {
    if (train_step_mode == PRV_STEP_FIXED_S) {
      to.step_mode = PRV_STEP_FIXED_S;
      to.n_samples = 128;
    } else if (train_step_mode == PRV_STEP_NGP) {
      to.step_mode = PRV_STEP_NGP;
      to.n_samples = PRV_NGP_MAX_STEPS;
    }
    if (train_rays > 0) to.n_rays = train_rays;
    if (train_patch_w > 0) to.patch_w = train_patch_w;
    if (train_patch_h > 0) to.patch_h = train_patch_h;
    if (std::max(to.patch_w, 1) * std::max(to.patch_h, 1) > 1 && train_step_mode < 0) { // pixel patches are an option of the fixed rule
      to.step_mode = PRV_STEP_FIXED_S;
      to.n_samples = 128;
    }
  }
  prv_field_desc train_desc{};
  uint64_t train_seed = 0x1234;
  bool images_from_files = false; // train_images: files -> the json's file_path PNGs (the reference's data flow)

  // load_training_data (run.py:109) of one scene json: dataset cameras + the RGBA8 training images on the device.  Kept
  // until another scene is asked for: the reference's per-member run.py calls of one iteration (main.cpp:2041-2043) all
  // name the same json, and every one of them would reload the same files.
  std::string data_scene;
  prv_camset* data_cams = nullptr;
  uint8_t* data_imgs = nullptr;
  int data_w = 0, data_h = 0;
  void drop_training_data() {
    if (data_cams) prv_camset_destroy(data_cams);
    if (data_imgs) prv_free(ctx, data_imgs);
    data_cams = nullptr;
    data_imgs = nullptr;
    data_scene.clear();
  }
  int training_data(const std::string& scene_json) {
    if (data_cams && data_scene == scene_json) return PRV_OK;
    drop_training_data();
    prv_camset* ds = nullptr;
    if (prv_cameras_from_dataset_json(ctx, scene_json.c_str(), &ds) != PRV_OK) return -30;
    int w = 0, h = 0;
    prv_camset_size(ds, &w, &h);
    prv_render_opts o{};
    o.width = train_w > 0 ? train_w : w; // training images at the dataset size unless the config shrinks them
    o.height = train_h > 0 ? train_h : h;
    // the ground truth is rendered with the rule every other render of this run uses (yaml samples_per_ray; 0 = the
    // engine's own stepping): the PSNR evaluation compares like with like
    o.samples_per_ray = sd->samples_per_ray;
    o.step_mode = sd->samples_per_ray > 0 ? PRV_STEP_FIXED_S : PRV_STEP_NGP;
    o.spp = 1;
    o.min_transmittance = 1e-4f;
    const int n = prv_camset_count(ds);
    uint8_t* imgs = nullptr;
    int rc = prv_malloc(ctx, (void**)&imgs, (size_t)n * o.width * o.height * 4);
    if (rc == PRV_OK && images_from_files) {
      // every frame's file_path, relative to the json, as RGBA8
      std::string text, err;
      prvjson::Value root;
      if (!prvjson::read_file(scene_json, text) || !prvjson::Parser(text).parse(root, err)) rc = -31;
      const std::string base = scene_json.substr(0, scene_json.find_last_of('/') + 1);
      std::vector<uint8_t> px;
      for (int i = 0; rc == PRV_OK && i < n; i++) {
        std::string fp = root.at("frames").arr[(size_t)i].at("file_path").s;
        if (fp.size() < 4 || fp.substr(fp.size() - 4) != ".png") fp += ".png";
        int iw = 0, ih = 0;
        const int prc = png_read_rgba8(base + fp, &iw, &ih, px);
        if (prc != 0 || iw != o.width || ih != o.height) {
          std::cerr << "training image " << base + fp << ": " << (prc ? "cannot read (" + std::to_string(prc) + ")" : "size differs from the json") << std::endl;
          rc = -32;
          break;
        }
        rc = prv_memcpy_h2d(ctx, imgs + (size_t)i * o.width * o.height * 4, px.data(), px.size());
      }
    } else if (rc == PRV_OK) {
      rc = prv_render_rgba8(ctx, 6, ds, nullptr, n, &o, imgs, nullptr); // straight alpha over nothing
    }
    if (rc != PRV_OK) {
      std::cerr << "prv: " << prv_last_error(ctx) << std::endl;
      prv_camset_destroy(ds);
      if (imgs) prv_free(ctx, imgs);
      return rc;
    }
    data_scene = scene_json;
    data_cams = ds;
    data_imgs = imgs;
    data_w = o.width;
    data_h = o.height;
    return PRV_OK;
  }

  // trainers of the listed members on one scene json, for the members this rank owns: a fresh field per member
  // (seed = train_seed + e) in the member's slot; appended to `trs` (the caller steps and destroys them)
  int create_trainers(const std::string& scene_json, const std::vector<int>& members, std::vector<prv_trainer*>& trs, double* t_fresh = nullptr,
                      double* t_create = nullptr) {
    int rc = training_data(scene_json);
    for (size_t k = 0; rc == PRV_OK && k < members.size(); k++) {
      const int e = members[k];
      if (owner_of(e) != rank) continue;
      double t0 = now_seconds();
      rc = prv_model_fresh(ctx, slot_of(e), &train_desc, train_seed + (uint64_t)e);
      if (t_fresh) *t_fresh += now_seconds() - t0;
      prv_train_opts to;
      prv_train_default_opts(&to);
      apply_train_opts(to);
      to.seed += (uint64_t)e;
      prv_trainer* tr = nullptr;
      t0 = now_seconds();
      if (rc == PRV_OK) rc = prv_train_create(ctx, slot_of(e), data_cams, data_imgs, data_w, data_h, &to, &tr);
      if (t_create) *t_create += now_seconds() - t0;
      if (tr) trs.push_back(tr);
    }
    if (rc != PRV_OK) std::cerr << "prv: " << prv_last_error(ctx) << std::endl;
    return rc;
  }

  // `--train --n_steps N` for the listed members on one scene json, the members stepping side by side (prv_train_steps_multi);
  // with several ranks a member is trained by its owner (owner_of)
  int train_member_list(const std::string& scene_json, const std::vector<int>& members, int steps) {
    const bool timing = getenv("PRV_PLANNER_TIMING") != nullptr; // dev: where an iteration's seconds go
    const double t_start = now_seconds();
    double t_fresh = 0, t_create = 0;
    int rc = training_data(scene_json);
    if (rc != PRV_OK) return rc;
    if (timing) prv_synchronize(ctx);
    const double t_gt = now_seconds() - t_start;
    std::vector<prv_trainer*> trs;
    rc = create_trainers(scene_json, members, trs, &t_fresh, &t_create);
    const double t0 = now_seconds();
    if (rc == PRV_OK && !trs.empty()) rc = prv_train_steps_multi(trs.data(), (int)trs.size(), steps, nullptr);
    const double t_steps = now_seconds() - t0;
    uint64_t samples_last = 0;
    int active_rays = 0;
    if (timing && !trs.empty()) {
      (void)prv_train_info(trs[0], nullptr, &samples_last, nullptr);
      active_rays = prv_train_active_rays(trs[0]);
    }
    for (prv_trainer* tr : trs) prv_train_destroy(tr);
    if (timing)
      std::cerr << "train_members: views " << prv_camset_count(data_cams) << " gt " << t_gt << " s, fresh " << t_fresh << " s, create "
                << t_create << " s, steps " << t_steps << " s, total " << now_seconds() - t_start << " s; last batch of member " << (trs.empty() ? -1 : members[0])
                << ": " << samples_last << " samples, " << active_rays << " rays" << std::endl;
    if (rc != PRV_OK) std::cerr << "prv: " << prv_last_error(ctx) << std::endl;
    return rc;
  }

  // the members of the ensemble reach every rank (each trained by its owner): one group of broadcasts
  int exchange_members() {
    if (!comm) return PRV_OK;
    std::vector<int> slots(n_members), owners(n_members);
    for (int e = 0; e < n_members; e++) {
      slots[e] = slot_of(e);
      owners[e] = owner_of(e);
    }
    const int rc = prv_model_exchange_slots(ctx, comm, n_members, slots.data(), owners.data(), &train_desc);
    if (rc != PRV_OK) std::cerr << "prv: " << prv_last_error(ctx) << std::endl;
    return rc;
  }

  // the whole ensemble of an iteration in one go (the fused path): train, then exchange between ranks
  int train_members(const std::string& scene_json) {
    std::vector<int> all(n_members);
    for (int e = 0; e < n_members; e++) all[e] = e;
    int rc = members_pretrained ? PRV_OK : train_member_list(scene_json, all, train_steps);
    members_pretrained = false; // (the shell's lockstep round trained them: this call only brings them together)
    if (rc == PRV_OK) rc = exchange_members();
    drop_training_data(); // the next iteration's json differs
    return rc;
  }

  // The engine behind NBV_Net_Labeler::train_by_instantNGP's reference signature: ONE run.py invocation
  // (run.py:185-208 train, :226-277 evaluate, :284-309 screenshots) through include/prv.h, in process.
  //   --scene S --n_steps N                        train model slot max(ensemble_id, 0) on S (skipped when the run is
  //                                                configured with pretrained members: train_steps == 0)
  //   --screenshot_transforms J --screenshot_dir D  render every frame of J at its w x h, spp = screenshot_spp, opaque
  //                                                black background (run.py:94), write D/<basename(file_path)>
  //   --test_transforms T --save_metrics M          render every frame of T (spp 1, min_T 1e-4, run.py:231-235), PSNR /
  //                                                SSIM against renders of the ground-truth field, "PSNR\t..\nSSIM\t.." -> M
  int run_py(const RunPyArgs& a) {
    const int member = a.ensemble_id >= 0 ? a.ensemble_id : 0, slot = slot_of(member);
    if (train_steps > 0) {
      const int rc = train_member_list(a.scene, {member}, a.n_steps > 0 ? std::min(a.n_steps, train_steps) : train_steps);
      if (rc != PRV_OK) return rc;
    }
    if (!a.screenshot_transforms.empty()) {
      prv_camset* cams = nullptr;
      if (prv_cameras_from_json(ctx, a.screenshot_transforms.c_str(), &cams) != PRV_OK) {
        std::cerr << "prv: " << prv_last_error(ctx) << std::endl;
        return -20;
      }
      std::string text, err;
      prvjson::Value root;
      if (!prvjson::read_file(a.screenshot_transforms, text) || !prvjson::Parser(text).parse(root, err)) {
        std::cerr << "prv: cannot read " << a.screenshot_transforms << (err.empty() ? "" : ": " + err) << std::endl;
        prv_camset_destroy(cams);
        return -31;
      }
      const prv_render_opts o = candidate_opts(cams);
      const int n = prv_camset_count(cams);
      if (!root.has("frames") || (int)root.at("frames").arr.size() != n) { // the names below index the same array the cameras came from
        std::cerr << "prv: " << a.screenshot_transforms << ": " << n << " cameras but " << (root.has("frames") ? root.at("frames").arr.size() : 0) << " frames" << std::endl;
        prv_camset_destroy(cams);
        return -32;
      }
      const size_t bytes = (size_t)o.width * o.height * 4;
      uint8_t* dev = nullptr;
      std::vector<uint8_t> px(bytes);
      int rc = prv_malloc(ctx, (void**)&dev, (size_t)std::max(1, n) * bytes);
      if (rc == PRV_OK) rc = prv_render_rgba8(ctx, slot, cams, nullptr, n, &o, dev, nullptr);
      if (rc != PRV_OK) std::cerr << "prv: " << prv_last_error(ctx) << std::endl;
      sd->access_directory(a.screenshot_dir);
      for (int k = 0; rc == PRV_OK && k < n; k++) {
        std::string name = root.at("frames").arr[(size_t)k].at("file_path").s; // os.path.basename(f["file_path"]) (run.py:297)
        name = name.substr(name.find_last_of('/') + 1);
        if (name.find('.') == std::string::npos) name += ".png";
        rc = prv_memcpy_d2h(ctx, px.data(), dev + (size_t)k * bytes, bytes);
        if (rc != PRV_OK) std::cerr << "prv: " << prv_last_error(ctx) << std::endl;
        else if (png_write_rgba8(a.screenshot_dir + name, o.width, o.height, px.data()) != 0) {
          std::cerr << "prv: cannot write " << a.screenshot_dir + name << std::endl; // (not prv_last_error: the failure is the file's)
          rc = PRV_E_IO;
        }
      }
      if (dev) prv_free(ctx, dev);
      prv_camset_destroy(cams);
      return rc;
    }
    if (!a.test_transforms.empty()) {
      double psnr = 0, ssim = 0;
      const int rc = evaluate_on(a.test_transforms, slot, &psnr, &ssim);
      if (rc != PRV_OK) return rc;
      char buf[128];
      snprintf(buf, sizeof(buf), "PSNR\t%.17g\nSSIM\t%.17g", psnr, ssim); // run.py:275-277
      return write_text(a.save_metrics, buf) ? 0 : PRV_E_IO;
    }
    return 0;
  }

  // the candidate renders' options: the render json's own w x h (run.py:304), screenshot_spp sub-samples, the engine's
  // stepping rule unless the yaml fixes a sample count, opaque black background (run.py:94)
  prv_render_opts candidate_opts(const prv_camset* cams) const {
    int w = 0, h = 0;
    prv_camset_size(cams, &w, &h);
    prv_render_opts o{};
    o.width = sd->render_width > 0 ? sd->render_width : w;
    o.height = sd->render_height > 0 ? sd->render_height : h;
    o.samples_per_ray = sd->samples_per_ray;
    o.step_mode = sd->samples_per_ray > 0 ? PRV_STEP_FIXED_S : PRV_STEP_NGP;
    o.spp = sd->screenshot_spp;
    o.min_transmittance = (float)sd->min_transmittance;
    o.background[0] = o.background[1] = o.background[2] = 0.f;
    o.background[3] = 1.f;
    return o;
  }

  // `evaluate: 1`: train ONE field on the final view set and score it on the test view set
  // (<viewspace_path>/<evaluate_views>.txt, default 100 as in main.cpp:1956) against renders of the ground truth
  int evaluate(const std::string& scene_json, const Vec3& center, double size, double* psnr, double* ssim) {
    const int keep = n_members;
    n_members = 1;
    int rc = train_steps > 0 ? train_members(scene_json) : PRV_OK; // without training: member 0 as loaded
    n_members = keep;
    if (rc != PRV_OK) return rc;
    std::string test_json;
    if ((rc = write_test_json(center, size, &test_json)) != PRV_OK) return rc;
    return evaluate_on(test_json, slot_of(0), psnr, ssim);
  }

  // the test cameras' json, <gt_path>/<evaluate_views>.json: the test view set placed like the candidates, full-size
  // dataset header (what get_coverage leaves for the reference's "100" test set, main.cpp:1581-1651)
  int write_test_json(const Vec3& center, double size, std::string* path) {
    auto sd_test = std::make_shared<Share_Data>(sd->yaml_file_path, sd->name_of_pcd, eval_views, -1, sd->method_of_IG);
    if (!sd_test->ok) {
      std::cerr << sd_test->error << std::endl;
      return -40;
    }
    View_Space vs(sd_test);
    vs.set_view_space(center, size);
    Value root = transforms_header(sd->color_intrinsics, sd->ray_casting_aabb_scale, size, center, 0);
    for (size_t i = 0; i < vs.views.size(); i++) {
      Value v;
      v["file_path"] = Value(std::to_string(eval_views) + "/rgbaClip_" + std::to_string(i) + ".png");
      v["transform_matrix"] = matrix_json(view_transform_matrix(vs.views[i], Mat4::Identity(), center));
      root["frames"].append(v);
    }
    *path = sd->gt_path + "/" + std::to_string(eval_views) + ".json";
    sd->access_directory(sd->gt_path);
    return write_text(*path, prvjson::to_styled_string(root)) ? PRV_OK : PRV_E_IO;
  }

  // Mode 4's curve points trained SIDE BY SIDE: one fresh field per scene json on model slots 0, 1, ... (at most 5: slots
  // 6 and 7 hold the ground truth and the reference field), each on its own dataset, all stepped together by
  // prv_train_steps_multi -- the reference trains them one run.py after another (main.cpp:2469-2479); the fields are the
  // same fields either way (own slot, own seed, own data), the GPU is just not left idle between a single trainer's kernels.
  int train_scenes_side_by_side(const std::vector<std::string>& scenes) {
    struct Job {
      prv_camset* cams = nullptr;
      uint8_t* imgs = nullptr;
      prv_trainer* tr = nullptr;
    };
    std::vector<Job> jobs(scenes.size());
    int rc = PRV_OK;
    for (size_t k = 0; rc == PRV_OK && k < scenes.size(); k++) {
      drop_training_data();
      if ((rc = training_data(scenes[k])) != PRV_OK) break;
      jobs[k].cams = data_cams; // the job owns them from here
      jobs[k].imgs = data_imgs;
      data_cams = nullptr;
      data_imgs = nullptr;
      data_scene.clear();
      rc = prv_model_fresh(ctx, (int)k, &train_desc, train_seed); // every curve point starts from the same initial field, as `evaluate` does
      prv_train_opts to;
      prv_train_default_opts(&to);
      apply_train_opts(to);
      if (rc == PRV_OK) rc = prv_train_create(ctx, (int)k, jobs[k].cams, jobs[k].imgs, data_w, data_h, &to, &jobs[k].tr);
    }
    std::vector<prv_trainer*> trs;
    for (auto& j : jobs)
      if (j.tr) trs.push_back(j.tr);
    if (rc == PRV_OK && !trs.empty()) rc = prv_train_steps_multi(trs.data(), (int)trs.size(), train_steps, nullptr);
    if (rc != PRV_OK) std::cerr << "prv: " << prv_last_error(ctx) << std::endl;
    for (auto& j : jobs) {
      if (j.tr) prv_train_destroy(j.tr);
      if (j.cams) prv_camset_destroy(j.cams);
      if (j.imgs) prv_free(ctx, j.imgs);
    }
    return rc;
  }

  // run.py:226-277 for one model slot: every frame of the test json (full-size dataset cameras, lens included) at
  // spp 1 / min_T 1e-4 over opaque black, against renders of the ground-truth field (slot 6) -> mean PSNR, mean SSIM
  int evaluate_on(const std::string& test_json, int slot, double* psnr, double* ssim) {
    prv_camset* cams = nullptr;
    if (prv_cameras_from_dataset_json(ctx, test_json.c_str(), &cams) != PRV_OK) {
      std::cerr << "prv: " << prv_last_error(ctx) << std::endl;
      return -41;
    }
    int w = 0, h = 0;
    prv_camset_size(cams, &w, &h);
    prv_render_opts o{};
    o.width = train_w > 0 ? train_w : w;
    o.height = train_h > 0 ? train_h : h;
    o.samples_per_ray = sd->samples_per_ray;
    o.step_mode = sd->samples_per_ray > 0 ? PRV_STEP_FIXED_S : PRV_STEP_NGP;
    o.spp = 1; // snap_to_pixel_centers (run.py:231)
    o.min_transmittance = 1e-4f; // run.py:235
    o.background[3] = 1.f;       // black, opaque (run.py:226)
    const int n = prv_camset_count(cams);
    int rc = PRV_OK;
    // the reference images of the test set are rendered once per test json and size: mode 4 evaluates a dozen fields on them
    const std::string key = test_json + "@" + std::to_string(o.width) + "x" + std::to_string(o.height);
    if (!test_gt || test_gt_key != key) {
      drop_test_images();
      rc = prv_malloc(ctx, (void**)&test_gt, (size_t)n * o.width * o.height * 16);
      prv_render_opts og = o;
      og.background[3] = 0.f;
      if (rc == PRV_OK) rc = prv_render(ctx, 6, cams, nullptr, n, &og, test_gt, nullptr);
      if (rc == PRV_OK) test_gt_key = key;
      else drop_test_images();
    }
    if (rc == PRV_OK) rc = prv_evaluate(ctx, slot, cams, nullptr, n, &o, test_gt, psnr, ssim);
    if (rc != PRV_OK) std::cerr << "prv: " << prv_last_error(ctx) << std::endl;
    prv_camset_destroy(cams);
    return rc;
  }
  float* test_gt = nullptr;
  std::string test_gt_key;
  void drop_test_images() {
    if (test_gt) prv_free(ctx, test_gt);
    test_gt = nullptr;
    test_gt_key.clear();
  }
  int eval_views = 100;

  bool save_renders = false; // save_renders: 1 -> the PNG tree the reference's run.py leaves (main.cpp:1676-1684)

  int operator()(int method, int iteration, const std::string& scene_json, const std::string& render_json,
                 const std::vector<int>& ids, std::vector<double>& scores) {
    if (train_steps > 0) {
      const int trc = train_members(scene_json);
      if (trc != PRV_OK) return trc;
    }
    const bool timing = getenv("PRV_PLANNER_TIMING") != nullptr; // dev: where an iteration's seconds go
    const double t_round = now_seconds();
    prv_camset* cams = nullptr;
    if (prv_cameras_from_json(ctx, render_json.c_str(), &cams) != PRV_OK) {
      std::cerr << "prv: " << prv_last_error(ctx) << std::endl;
      return -20;
    }
    const double t_json = now_seconds() - t_round;
    const int n = prv_camset_count(cams);
    prv_render_opts o = candidate_opts(cams);
    std::vector<int> slots(n_members);
    for (int e = 0; e < n_members; e++) slots[e] = slot_of(e);
    std::vector<prv_score_record> rec(n);
    // this rank's shard of the frames of the render json (all of them with one rank): interleaved, as bench.py deals them
    std::vector<int> mine((size_t)std::max(1, (n + world - 1) / world));
    int n_mine = 0;
    prv_shard_views(n, rank, world, 1, mine.data(), &n_mine);
    int rc;
    if (method == PSNRCoverage) {
      // frame k of the render json is candidate ids[k]: pick its reference image out of the full set
      const size_t px = (size_t)o.width * o.height * 4;
      float* gt_sel = nullptr;
      if (prv_malloc(ctx, (void**)&gt_sel, (size_t)std::max(1, n_mine) * px * sizeof(float)) != PRV_OK) return -21;
      std::vector<float> tmp(px);
      for (int k = 0; k < n_mine; k++) { // device-to-device through the host keeps this file free of HIP headers
        prv_memcpy_d2h(ctx, tmp.data(), gt_dev + (size_t)ids[mine[k]] * px, px * sizeof(float));
        prv_memcpy_h2d(ctx, gt_sel + (size_t)k * px, tmp.data(), px * sizeof(float));
      }
      o.background[3] = 0.f;
      rc = prv_score_views_sharded(ctx, comm, PRV_SCORE_PSNR_COVERAGE, slots.data(), 1, cams, n, 1, &o, gt_sel, rec.data(), nullptr);
      prv_free(ctx, gt_sel);
    } else {
      rc = prv_score_views_sharded(ctx, comm, method, slots.data(), n_members, cams, n, 1, &o, nullptr, rec.data(), nullptr);
      if (rc == PRV_OK && save_renders) { // <save_path>/render/<it>/ensemble_<e>/rgbaClip_<view id>.png, the files :2047 reads
        uint8_t* dev = nullptr;
        std::vector<uint8_t> px((size_t)o.width * o.height * 4);
        rc = prv_malloc(ctx, (void**)&dev, (size_t)n * px.size());
        for (int e = 0; rc == PRV_OK && e < n_members; e++) {
          const std::string dir = sd->save_path + "/render/" + std::to_string(iteration) + "/ensemble_" + std::to_string(e);
          sd->access_directory(dir);
          rc = prv_render_rgba8(ctx, slot_of(e), cams, nullptr, n, &o, dev, nullptr);
          for (int k = 0; rc == PRV_OK && k < n; k++) {
            rc = prv_memcpy_d2h(ctx, px.data(), dev + (size_t)k * px.size(), px.size());
            if (rc == PRV_OK && png_write_rgba8(dir + "/rgbaClip_" + std::to_string(ids[k]) + ".png", o.width, o.height, px.data()) != 0) rc = PRV_E_IO;
          }
        }
        if (dev) prv_free(ctx, dev);
      }
    }
    prv_camset_destroy(cams);
    if (rc != PRV_OK) {
      std::cerr << "prv: " << prv_last_error(ctx) << std::endl;
      return rc;
    }
    if (timing) std::cerr << "score_round: views " << n << " cameras " << t_json << " s, total " << now_seconds() - t_round << " s" << std::endl;
    for (int k = 0; k < n; k++) scores[k] = rec[k].score;
    if (dump_records) { // yaml dump_scores / PRV_PLANNER_DUMP_RECORDS (view_planning): the gathered records of every iteration, byte for byte
      sd->access_directory(sd->save_path + "/records");
      write_text(sd->save_path + "/records/" + std::to_string(iteration) + ".bin",
                 std::string((const char*)rec.data(), rec.size() * sizeof(prv_score_record)));
    }
    return 0;
  }
};

// `train_steps` and friends -> the scorer's in-process training settings; the ground truth goes to slot 6
int configure_training(prv_ctx* ctx, const FileStorage& fs, const prv_field_desc& desc, int train_steps, HipScorer& scorer);

// Optimiser steps per member per iteration.  The reference retrains every ensemble member on the views chosen so
// far in EVERY iteration with `--n_steps <n_steps>` (main.cpp:1668, 2041-2043), so the yaml's `n_steps` is the
// default; `train_steps: N` overrides the count.  Scoring members that are NOT retrained (files / synthetic seeds:
// a ranking that does not depend on the acquired views) is an explicit opt-in: `train_steps: 0` or
// `pretrained_members: 1`.
int configured_train_steps(const FileStorage& fs, const Share_Data& sd) {
  if (fs.has("pretrained_members") && fs.num("pretrained_members") > 0) return 0;
  if (fs.has("train_steps")) return std::max(0, (int)fs.num("train_steps"));
  return std::max(0, sd.n_steps);
}
int write_coverage_images(prv_ctx* ctx, const std::shared_ptr<Share_Data>& sd);

prv_field_desc field_from_config(const FileStorage& fs) {
  prv_field_desc d{};
  auto get = [&](const char* k, double dflt) { return fs.has(k) ? fs.num(k) : dflt; };
  d.n_levels = (int)get("field_levels", 8);
  d.n_features = (int)get("field_features", 4);
  d.log2_hashmap = (int)get("field_log2_hashmap", 19);
  d.base_res = (int)get("field_base_res", 16);
  d.finest_res = (int)get("field_finest_res", 256);
  d.occ_res = (int)get("field_occ_res", 128);
  d.density_bias = (float)get("field_density_bias", 3.0);
  d.table_amp = (float)get("synthetic_table_amp", 4.0);
  d.per_level_scale = (float)get("field_per_level_scale", 0.0); // > 0: tiny-cuda-nn's level recipe (include/prv.h)
  return d;
}

// One object's planning run: everything view_planning does around NBV_Net_Labeler::nbv_loop, kept in one place so that the
// shell can hold SEVERAL objects' loops at once (`shard: members`: run_members_lockstep below).  slot_base / pair_base: where
// the object's members live and who trains them (HipScorer).
struct PlanningJob {
  prv_ctx* ctx = nullptr;
  std::shared_ptr<Share_Data> sd;
  std::unique_ptr<NBV_Net_Labeler> lab;
  HipScorer* state = nullptr; // the scorer object inside lab->scorer (engine, evaluator and the loop share it)
  float* gt_dev = nullptr;
  int first_view_id = -1;
  int setup(prv_ctx* ctx_, const std::string& cfg, const std::string& name, int method, prv_comm* comm, int slot_base = 0, int pair_base = 0);
  int finish(int rc);
};

int PlanningJob::setup(prv_ctx* ctx_, const std::string& cfg, const std::string& name, int method, prv_comm* comm, int slot_base, int pair_base) {
  ctx = ctx_;
  if (!NBV_Net_Labeler::method_in_scope(method)) { // before any directory, model or training: methods 1 / 4 are the PRVNet pipeline's
    std::cerr << "method_of_IG " << method << " is not built: this planner runs methods 0 (RandomIterative), 2 (EnsembleRGB), 3 "
                 "(EnsembleRGBDensity) and 5 (PSNRCoverage); 1 (RandomOneshot) and 4 (PVBCoverage) need the reference's PRVNet server" << std::endl;
    return -10;
  }
  sd = std::make_shared<Share_Data>(cfg, name, -1, -1, method); // main.cpp:3876
  if (!sd->ok) {
    std::cerr << sd->error << std::endl;
    return -1;
  }
  const int rank = comm ? prv_comm_rank(comm) : 0, world = comm ? prv_comm_world(comm) : 1;
  if (rank > 0) sd->relocate_outputs(sd->pre_path + "rank" + std::to_string(rank) + "/"); // same loop, own scratch tree
  FileStorage fs;
  fs.open(cfg);
  const prv_field_desc desc = field_from_config(fs);
  const int members = (method == EnsembleRGB || method == EnsembleRGBDensity) ? sd->ensemble_num : 1;
  const int train_steps = configured_train_steps(fs, *sd);
  if (train_steps == 0 && method != RandomIterative)
    std::cerr << "WARNING: the members are NOT retrained on the chosen views (train_steps: 0 / pretrained_members: 1): "
                 "the view ranking of this run does not depend on the views it acquires (the reference retrains "
                 "n_steps per member per iteration, main.cpp:1668)" << std::endl;
  if (train_steps == 0 && (slot_base != 0 || pair_base != 0)) {
    std::cerr << "shard: members deals TRAININGS to the ranks: it needs train_steps > 0" << std::endl;
    return -25;
  }
  for (int e = 0; e < members && train_steps == 0; e++) {
    int rc;
    if (fs.has("synthetic_seed")) rc = prv_model_synthetic(ctx, e, &desc, (uint64_t)fs.num("synthetic_seed") + (uint64_t)e);
    else { // <model_path>/<object>/member_<e>.prvf, or the instant-ngp snapshot of the same name (.ingp / .msgpack)
      const std::string stem = sd->model_path + name + "/member_" + std::to_string(e);
      rc = PRV_E_IO;
      for (const char* ext : {".prvf", ".ingp", ".msgpack"}) {
        if (!std::ifstream(stem + ext).is_open()) continue;
        rc = std::string(ext) == ".prvf" ? prv_model_load_file(ctx, e, (stem + ext).c_str()) : prv_model_load_ingp(ctx, e, (stem + ext).c_str());
        break;
      }
      if (rc == PRV_E_IO && !std::ifstream(stem + ".prvf").is_open() && !std::ifstream(stem + ".ingp").is_open() && !std::ifstream(stem + ".msgpack").is_open())
        rc = prv_model_load_file(ctx, e, (stem + ".prvf").c_str()); // for its error message
    }
    if (rc != PRV_OK) {
      std::cerr << "prv: " << prv_last_error(ctx) << std::endl;
      return rc;
    }
  }
  // synthetic objects sit at the origin (+1e-10, main.cpp:447) with the configured size
  const double size = fs.has("object_size") ? fs.num("object_size") : 0.1;
  const Vec3 center(1e-10, 1e-10, 1e-10);
  HipScorer scorer;
  scorer.ctx = ctx;
  scorer.sd = sd;
  scorer.comm = comm;
  scorer.rank = rank;
  scorer.world = world;
  scorer.n_members = members;
  scorer.slot_base = slot_base;
  scorer.pair_base = pair_base;
  scorer.save_renders = fs.has("save_renders") && fs.num("save_renders") > 0;
  if (train_steps > 0) { // members are trained from scratch every iteration
    const int rc = configure_training(ctx, fs, desc, train_steps, scorer);
    if (rc != PRV_OK) return rc;
  }
  if (fs.has("evaluate_views")) scorer.eval_views = (int)fs.num("evaluate_views");
  // diagnostics of the shell, not of the loop: every iteration's scores (raw doubles) and gathered records on disk
  const bool dump_scores = (fs.has("dump_scores") && fs.num("dump_scores") > 0) || getenv("PRV_PLANNER_DUMP_RECORDS") != nullptr;
  scorer.dump_records = dump_scores; // (before the labeler copies the scorer)
  lab.reset(new NBV_Net_Labeler(sd, center, size, scorer));
  NBV_Net_Labeler& labeler = *lab;
  auto sd = this->sd; // (the lambdas below capture the shared pointer by value)
  if (dump_scores) {
    labeler.on_scores = [sd](int iteration, const std::vector<int>&, const std::vector<double>& scores) {
      sd->access_directory(sd->save_path + "/scores");
      write_text(sd->save_path + "/scores/" + std::to_string(iteration) + ".bin", std::string((const char*)scores.data(), scores.size() * sizeof(double)));
    };
  }
  labeler.get_coverage(); // <gt_path>/<N>.json (main.cpp:3882-3978, json part)
  if (fs.has("coverage_images") && fs.num("coverage_images") > 0 && train_steps > 0) {
    const int rc = write_coverage_images(ctx, sd);
    if (rc != PRV_OK) return rc;
  }
  if (method == PSNRCoverage) { // reference images of every view, rendered once from the reference field
    int rc = fs.has("synthetic_seed") ? prv_model_synthetic(ctx, 7, &desc, (uint64_t)fs.num("synthetic_seed") + 4096)
                                      : prv_model_load_file(ctx, 7, (sd->model_path + name + "/reference.prvf").c_str());
    if (rc != PRV_OK) {
      std::cerr << "prv: " << prv_last_error(ctx) << std::endl;
      return rc;
    }
    // all views with the candidate header (divisor 16) so the images match the render json
    const std::string all_json = sd->gt_path + "/" + std::to_string(sd->num_of_views) + "_render.json";
    Value root = transforms_header(sd->color_intrinsics, sd->ray_casting_aabb_scale, size, center, sd->candidate_divisor);
    for (size_t i = 0; i < labeler.view_space->views.size(); i++) {
      Value v;
      v["file_path"] = Value("rgbaClip_" + std::to_string(i) + ".png");
      v["transform_matrix"] = matrix_json(view_transform_matrix(labeler.view_space->views[i], Mat4::Identity(), center));
      root["frames"].append(v);
    }
    write_text(all_json, prvjson::to_styled_string(root));
    prv_camset* cams = nullptr;
    if (prv_cameras_from_json(ctx, all_json.c_str(), &cams) != PRV_OK) return -22;
    int w, h;
    prv_camset_size(cams, &w, &h);
    prv_render_opts o{};
    o.width = sd->render_width > 0 ? sd->render_width : w;
    o.height = sd->render_height > 0 ? sd->render_height : h;
    o.samples_per_ray = sd->samples_per_ray;
    o.step_mode = sd->samples_per_ray > 0 ? PRV_STEP_FIXED_S : PRV_STEP_NGP;
    o.spp = sd->screenshot_spp;
    o.min_transmittance = (float)sd->min_transmittance;
    const int n = prv_camset_count(cams);
    if (prv_malloc(ctx, (void**)&scorer.gt_dev, (size_t)n * o.width * o.height * 16) != PRV_OK) return -23;
    rc = prv_render(ctx, 7, cams, nullptr, n, &o, scorer.gt_dev, nullptr);
    prv_camset_destroy(cams);
    if (rc != PRV_OK) {
      std::cerr << "prv: " << prv_last_error(ctx) << std::endl;
      return rc;
    }
    labeler.scorer = scorer;
  }
  // first view = the (0,0,r) top view of the set (main.cpp:3985-3995)
  first_view_id = -1;
  for (size_t i = 0; i < labeler.view_space->views.size(); i++) {
    const Vec3& p = labeler.view_space->views[i].init_pos;
    if (std::fabs(p.x) < 1e-6 && std::fabs(p.y) < 1e-6 && std::fabs(p.z - sd->view_space_radius) < 1e-6) first_view_id = (int)i;
  }
  if (first_view_id == -1) std::cout << "can not find now view id" << std::endl; // main.cpp:3993-3995
  // (after every assignment to labeler.scorer: the evaluator reaches into the scorer the loop will use)
  if (sd->evaluate) {
    if (train_steps == 0) { // evaluation needs a ground truth to compare with
      const int rc = prv_model_synthetic(ctx, 6, &desc, fs.has("ground_truth_seed") ? (uint64_t)fs.num("ground_truth_seed") : 0x5EED0002ull);
      if (rc != PRV_OK) return rc;
    }
    HipScorer* sc = labeler.scorer.target<HipScorer>();
    labeler.evaluator = [sc, center, size](const std::string& scene, double* p, double* q) {
      return sc ? sc->evaluate(scene, center, size, p, q) : -42;
    };
  }
  // train_by_instantNGP's reference signature -> the in-process engine (one run.py invocation per call); the engine
  // shares the loop's scorer object, so `score_path: png` and the fused path train and render the same members
  HipScorer* engine_state = labeler.scorer.target<HipScorer>();
  state = engine_state;
  gt_dev = scorer.gt_dev;
  labeler.engine = [engine_state](const RunPyArgs& a) { return engine_state ? engine_state->run_py(a) : -42; };
  if (sd->score_from_pngs && comm) {
    std::cerr << "score_path: png is the reference's single-process data flow; shard: views needs score_path: fused" << std::endl;
    return -24;
  }
  return 0;
}

int PlanningJob::finish(int rc) {
  NBV_Net_Labeler& labeler = *lab;
  HipScorer* engine_state = state;
  if (engine_state) {
    engine_state->drop_training_data();
    engine_state->drop_test_images();
  }
  if (gt_dev) prv_free(ctx, gt_dev);
  gt_dev = nullptr;
  if (labeler.final_psnr >= 0) std::cout << "final PSNR " << labeler.final_psnr << " SSIM " << labeler.final_ssim << std::endl;
  std::cout << "chosen_nbvs:";
  for (int v : labeler.chosen_nbvs) std::cout << ' ' << v;
  std::cout << std::endl;
  return rc;
}


int view_planning(prv_ctx* ctx, const std::string& cfg, const std::string& name, int method, prv_comm* comm) {
  PlanningJob job;
  int rc = job.setup(ctx, cfg, name, method, comm);
  if (rc != 0) return rc;
  return job.finish(job.lab->nbv_loop(job.first_view_id, 0));
}

// `shard: members` (BASELINE configs[4] on several GPUs; SURVEY section 8(e), main.cpp:2041-2043): the objects' loops in
// LOCKSTEP.  Round k of every object is prepared (its json pair written), then the E member-trainings of every object --
// objects x E (object, member) pairs, dealt to the ranks round-robin: pair p = o * E + e goes to rank p % world -- run side
// by side on their owners (prv_train_steps_multi over everything this rank owns), then object by object the members are
// brought together (prv_model_exchange_slots: one group of broadcasts), the candidates scored views-sharded (one all-gather)
// and the next view chosen.  With `shard: objects` 5 objects keep 5 of 8 GPUs busy; here 25 pairs keep all 8 busy with 3 or 4
// trainings each (22 % idle against 37.5 %).  Every rank walks every object's loop, so every collective is entered by all
// ranks in the same order; objects beyond the slots of one context (7 ensembles of up to 8 members) go in further batches.
int run_members_lockstep(prv_ctx* ctx, const std::string& cfg, const std::vector<std::string>& names, int method, prv_comm* comm) {
  constexpr int kBatch = (PRV_MAX_SLOTS - 8) / PRV_MAX_MODELS; // slots 0..7 stay the shared ones (ground truth 6, reference 7)
  int worst = 0;
  for (size_t b0 = 0; b0 < names.size(); b0 += (size_t)kBatch) {
    const size_t nb = std::min(names.size() - b0, (size_t)kBatch);
    std::vector<std::unique_ptr<PlanningJob>> jobs;
    std::vector<int> live; // 0: in the loop, 1: ended, < 0: failed
    for (size_t k = 0; k < nb; k++) {
      std::cout << "object " << names[b0 + k] << " method " << method << " (lockstep batch of " << nb << ")" << std::endl;
      jobs.emplace_back(new PlanningJob());
      PlanningJob& j = *jobs.back();
      int members = 1;
      {
        Share_Data probe(cfg, names[b0 + k], -1, -1, method);
        if (probe.ok && (method == EnsembleRGB || method == EnsembleRGBDensity)) members = probe.ensemble_num;
      }
      int rc = j.setup(ctx, cfg, names[b0 + k], method, comm, 8 + (int)k * PRV_MAX_MODELS, (int)(b0 + k) * members);
      if (rc == 0) rc = j.lab->nbv_begin(j.first_view_id, 0);
      live.push_back(rc);
    }
    for (;;) {
      bool any = false;
      for (size_t k = 0; k < nb; k++) // a round's json pairs (and, where a loop ends here, its final evaluation)
        if (live[k] == 0) {
          live[k] = jobs[k]->lab->nbv_prepare();
          any = any || live[k] == 0;
        }
      if (!any) break;
      // every (object, member) pair of the round this rank owns, side by side
      std::vector<prv_trainer*> trs;
      int rc = PRV_OK, steps = 0;
      const double t0 = now_seconds();
      for (size_t k = 0; k < nb && rc == PRV_OK; k++) {
        HipScorer* sc = jobs[k]->state;
        if (live[k] != 0 || !sc || sc->train_steps <= 0 || method == RandomIterative) continue;
        std::vector<int> all(sc->n_members);
        for (int e = 0; e < sc->n_members; e++) all[e] = e;
        rc = sc->create_trainers(jobs[k]->lab->run.scene_json, all, trs);
        steps = sc->train_steps;
        sc->members_pretrained = rc == PRV_OK;
      }
      const size_t n_mine = trs.size();
      if (rc == PRV_OK && !trs.empty()) rc = prv_train_steps_multi(trs.data(), (int)trs.size(), steps, nullptr);
      for (prv_trainer* tr : trs) prv_train_destroy(tr);
      if (getenv("PRV_PLANNER_TIMING"))
        std::cerr << "train_pairs: " << n_mine << " (object, member) trainings of this round on rank " << (comm ? prv_comm_rank(comm) : 0) << ", " << steps
                  << " steps side by side, " << now_seconds() - t0 << " s" << std::endl;
      if (rc != PRV_OK) {
        std::cerr << "prv: " << prv_last_error(ctx) << std::endl;
        for (size_t k = 0; k < nb; k++)
          if (live[k] == 0) live[k] = rc; // (every rank fails the same way or the next collective reports it)
        break;
      }
      for (size_t k = 0; k < nb; k++)
        if (live[k] == 0) live[k] = jobs[k]->lab->nbv_decide();
    }
    for (size_t k = 0; k < nb; k++) {
      const int rc = jobs[k]->lab ? jobs[k]->finish(live[k] < 0 ? live[k] : 0) : live[k];
      if (rc != 0) worst = rc;
    }
  }
  return worst;
}

int configure_training(prv_ctx* ctx, const FileStorage& fs, const prv_field_desc& desc, int train_steps, HipScorer& scorer) {
  const int rc = prv_model_synthetic(ctx, 6, &desc, fs.has("ground_truth_seed") ? (uint64_t)fs.num("ground_truth_seed") : 0x5EED0002ull);
  if (rc != PRV_OK) {
    std::cerr << "prv: " << prv_last_error(ctx) << std::endl;
    return rc;
  }
  scorer.train_steps = train_steps;
  if (fs.has("train_rays")) scorer.train_rays = (int)fs.num("train_rays"); // else prv_train_default_opts' batch
  if (fs.has("train_patch_w")) scorer.train_patch_w = (int)fs.num("train_patch_w"); // training rays as patches of adjacent pixels (a speed / quality trade)
  if (fs.has("train_patch_h")) scorer.train_patch_h = (int)fs.num("train_patch_h");
  if (fs.has("train_step_mode")) scorer.train_step_mode = (int)fs.num("train_step_mode"); // how a training ray is sampled (include/prv.h: prv_train_opts.step_mode)
  scorer.train_w = fs.has("train_width") ? (int)fs.num("train_width") : 0;
  scorer.train_h = fs.has("train_height") ? (int)fs.num("train_height") : 0;
  scorer.train_desc = desc;
  scorer.train_desc.density_bias = fs.has("train_density_bias") ? (float)fs.num("train_density_bias") : 0.0f;
  scorer.train_desc.table_amp = 1e-4f;
  if (fs.has("train_seed")) scorer.train_seed = (uint64_t)fs.num("train_seed");
  scorer.images_from_files = fs.has("train_images") && fs.str("train_images") == "files";
  return PRV_OK;
}

// get_coverage's images (main.cpp:1604-1618) from the ground-truth field: <gt_path>/<N>/rgbaClip_<i>.png for every
// view of the set, at the dataset size -- what mode 3 (GetCoverage) leaves on disk with PCL
int write_coverage_images(prv_ctx* ctx, const std::shared_ptr<Share_Data>& sd) {
  const std::string n = std::to_string(sd->num_of_views);
  prv_camset* ds = nullptr;
  if (prv_cameras_from_dataset_json(ctx, (sd->gt_path + "/" + n + ".json").c_str(), &ds) != PRV_OK) return -60;
  int w = 0, h = 0;
  prv_camset_size(ds, &w, &h);
  prv_render_opts o{};
  o.width = w;
  o.height = h;
  o.samples_per_ray = sd->samples_per_ray; // the rule of every other render of the run (0 = the engine's own stepping)
  o.step_mode = sd->samples_per_ray > 0 ? PRV_STEP_FIXED_S : PRV_STEP_NGP;
  o.spp = 1;
  o.min_transmittance = 1e-4f;
  const int count = prv_camset_count(ds);
  sd->access_directory(sd->gt_path + "/" + n);
  uint8_t* dev = nullptr;
  int rc = prv_malloc(ctx, (void**)&dev, (size_t)w * h * 4);
  std::vector<uint8_t> px((size_t)w * h * 4);
  for (int i = 0; rc == PRV_OK && i < count; i++) {
    rc = prv_render_rgba8(ctx, 6, ds, &i, 1, &o, dev, nullptr);
    if (rc == PRV_OK) rc = prv_memcpy_d2h(ctx, px.data(), dev, px.size());
    if (rc == PRV_OK && png_write_rgba8(sd->gt_path + "/" + n + "/rgbaClip_" + std::to_string(i) + ".png", w, h, px.data()) != 0) rc = PRV_E_IO;
  }
  if (rc != PRV_OK) std::cerr << "coverage images: " << prv_last_error(ctx) << std::endl;
  if (dev) prv_free(ctx, dev);
  prv_camset_destroy(ds);
  return rc;
}

// mode 4 (main.cpp:2463-2487): the PSNR-vs-#views curve of an object.  For n = 3, 3 + add, ... <= max: a field is
// trained on the n-view coverage set (<gt_path>/<n>.json, written here; view set <viewspace_path>/<n>.txt or, when
// that file does not exist, a generated n-point hemisphere) and evaluated on the test set; <gt_path>/<n>.txt gets
// run.py's two metrics lines -- the files NeRF_fit_curve.cpp:103-115 reads.  Existing files are kept (:2473).
int instant_ngp_curves(prv_ctx* ctx, const std::string& cfg, const std::string& name) {
  FileStorage fs;
  fs.open(cfg);
  const prv_field_desc desc = field_from_config(fs);
  const double size = fs.has("object_size") ? fs.num("object_size") : 0.1;
  const Vec3 center(1e-10, 1e-10, 1e-10);
  auto sd0 = std::make_shared<Share_Data>(cfg, name, -1, -1, 0);
  if (!sd0->ok) {
    std::cerr << sd0->error << std::endl;
    return -1;
  }
  const int train_steps = configured_train_steps(fs, *sd0);
  if (train_steps <= 0) {
    std::cerr << "mode 4 trains: n_steps / train_steps must be positive" << std::endl;
    return -50;
  }
  HipScorer scorer;
  scorer.ctx = ctx;
  scorer.sd = sd0;
  scorer.n_members = 1;
  int rc = configure_training(ctx, fs, desc, train_steps, scorer);
  if (rc != PRV_OK) return rc;
  if (fs.has("evaluate_views")) scorer.eval_views = (int)fs.num("evaluate_views");
  const int n_max = sd0->coverage_view_num_max > 0 ? sd0->coverage_view_num_max : 90;
  const int n_add = sd0->coverage_view_num_add > 0 ? sd0->coverage_view_num_add : 1;
  sd0->access_directory(sd0->gt_path);
  std::vector<int> counts;
  for (int n = 3; n <= n_max; n += n_add) counts.push_back(n);
  const int n_full = fs.has("coverage_view_num_full") ? (int)fs.num("coverage_view_num_full") : 100; // main.cpp:2480-2484
  counts.push_back(n_full); // the upper bound of the curve ("100.txt")
  std::vector<double> xs, ys;
  double max_psnr = 0.0;
  // pass 1: which curve points are missing (existing metrics files are kept, :2473), and their scene jsons
  std::vector<double> psnrs(counts.size(), 0.0);
  std::vector<size_t> missing;
  std::vector<std::string> scenes;
  for (size_t ci = 0; ci < counts.size(); ci++) {
    const int n = counts[ci];
    const std::string metrics = sd0->gt_path + "/" + std::to_string(n) + ".txt";
    double ssim = 0;
    if (prvh_read_metrics(metrics.c_str(), &psnrs[ci], &ssim) == 0) continue;
    // the n-view coverage set
    std::vector<std::vector<double>> pts((size_t)n, std::vector<double>(3, 0.0));
    std::ifstream fin(sd0->viewspace_path + std::to_string(n) + ".txt");
    if (fin.is_open()) {
      for (int i = 0; i < n; i++)
        for (int j = 0; j < 3; j++) fin >> pts[i][j];
    } else {
      std::vector<double> flat((size_t)n * 3);
      prvh_hemisphere_generate(n, flat.data());
      for (int i = 0; i < n; i++)
        for (int j = 0; j < 3; j++) pts[i][j] = flat[(size_t)i * 3 + j];
    }
    sd0->num_of_views = n;
    sd0->pt_sphere = pts;
    sd0->pt_norm = std::sqrt(pts[0][0] * pts[0][0] + pts[0][1] * pts[0][1] + pts[0][2] * pts[0][2]);
    View_Space vs(sd0);
    vs.set_view_space(center, size);
    Value root = transforms_header(sd0->color_intrinsics, sd0->ray_casting_aabb_scale, size, center, 0);
    for (size_t i = 0; i < vs.views.size(); i++) {
      Value v;
      v["file_path"] = Value(std::to_string(n) + "/rgbaClip_" + std::to_string(i) + ".png");
      v["transform_matrix"] = matrix_json(view_transform_matrix(vs.views[i], Mat4::Identity(), center));
      root["frames"].append(v);
    }
    // the test set's own json is <gt_path>/<evaluate_views>.json: keep the two apart when the sizes coincide
    const std::string scene = sd0->gt_path + "/" + std::to_string(n) + (n == scorer.eval_views ? "_train.json" : ".json");
    write_text(scene, prvjson::to_styled_string(root));
    missing.push_back(ci);
    scenes.push_back(scene);
  }
  // pass 2: the missing points, four fields at a time side by side (train_scenes_side_by_side), each evaluated on the test set
  // (a member-step costs 95 us in a round of four, 101 in a round of five, 98 of three: profiles/r05_member_queues.txt)
  std::string test_json;
  if (!missing.empty() && (rc = scorer.write_test_json(center, size, &test_json)) != PRV_OK) return rc;
  constexpr size_t kSideBySide = 4;
  for (size_t g0 = 0; g0 < missing.size(); g0 += kSideBySide) {
    const size_t g1 = std::min(missing.size(), g0 + kSideBySide);
    const bool timing = getenv("PRV_PLANNER_TIMING") != nullptr; // dev: where the curve's seconds go
    double t0 = now_seconds();
    if ((rc = scorer.train_scenes_side_by_side(std::vector<std::string>(scenes.begin() + (long)g0, scenes.begin() + (long)g1))) != PRV_OK) return rc;
    if (timing) std::cerr << "curve_group: " << g1 - g0 << " fields trained in " << now_seconds() - t0 << " s" << std::endl;
    for (size_t k = g0; k < g1; k++) {
      const int n = counts[missing[k]];
      double psnr = 0, ssim = 0;
      t0 = now_seconds();
      rc = scorer.evaluate_on(test_json, (int)(k - g0), &psnr, &ssim);
      if (timing) std::cerr << "curve_eval: " << n << " views, evaluated in " << now_seconds() - t0 << " s" << std::endl;
      if (rc != PRV_OK) return rc;
      prvh_write_metrics((sd0->gt_path + "/" + std::to_string(n) + ".txt").c_str(), psnr, ssim);
      std::cout << "views " << n << " PSNR " << psnr << " SSIM " << ssim << std::endl;
      psnrs[missing[k]] = psnr;
    }
  }
  scorer.drop_training_data();
  scorer.drop_test_images();
  for (size_t ci = 0; ci < counts.size(); ci++) {
    if (ci + 1 < counts.size()) {
      xs.push_back((double)counts[ci]);
      ys.push_back(psnrs[ci]);
    } else {
      max_psnr = psnrs[ci];
    }
  }
  // the label file Origin's fit script leaves next to the curve (NeRF_fit_curve.cpp:119-206)
  if (xs.size() >= 4) {
    const FitResult fit = fit_lognormal_cdf(xs, ys, max_psnr);
    write_label_file(sd0->gt_path + "/label.txt", fit, max_psnr);
    const Labels L = make_labels(fit.f, max_psnr);
    std::cout << "label: converged " << (fit.converged ? 1 : 0) << " gap 2% at " << L.gap[2] << " views, gradient 0.02 at "
              << L.gradient[1] << " views" << std::endl; // main.cpp:2641-2642 uses gradient index 1
  }
  return 0;
}

// mode 3 (GetCoverage, main.cpp:2343-2462 -> NBV_Net_Labeler's constructor :636-1035 + get_coverage :1581-1656) for
// an object given as a coloured cloud <model_path>/<name>.pcd: centre the cloud on its centroid, bounding radius x
// 17/16 (:827-833), bring it to the object size (the reference draws a random size in [0.075, 0.115] m and keeps it
// in <gt_path>/size.txt, :851-867; here: that file if it exists, else `object_size`), place the N views, write
// <gt_path>/<N>.json and the <N>/rgbaClip_<i>.png images (PCL screenshot + convertToAlpha + flip = prv_splat_points).
int get_coverage_from_cloud(prv_ctx* ctx, const std::string& cfg, const std::string& name) {
  FileStorage fs;
  fs.open(cfg);
  auto sd = std::make_shared<Share_Data>(cfg, name, -1, -1, 0);
  if (!sd->ok) {
    std::cerr << sd->error << std::endl;
    return -1;
  }
  std::vector<float> xyz;
  std::vector<uint8_t> rgb;
  const std::string cloud = sd->model_path + name + ".pcd";
  const int prc = pcd_read(cloud, xyz, rgb);
  if (prc != 0) {
    std::cerr << "cannot read " << cloud << " (" << prc << ")" << std::endl;
    return -70;
  }
  const size_t n = xyz.size() / 3;
  double c[3] = {0, 0, 0};
  for (size_t i = 0; i < n; i++)
    for (int a = 0; a < 3; a++) c[a] += xyz[3 * i + a];
  for (int a = 0; a < 3; a++) c[a] /= (double)n;
  double radius = 0.0;
  for (size_t i = 0; i < n; i++) {
    double d2 = 0;
    for (int a = 0; a < 3; a++) {
      xyz[3 * i + a] = (float)(xyz[3 * i + a] - c[a]); // "move to centre" (:793-797)
      d2 += (double)xyz[3 * i + a] * xyz[3 * i + a];
    }
    radius = std::max(radius, std::sqrt(d2));
  }
  const double predicted_raw = radius * 17.0 / 16.0;
  sd->access_directory(sd->gt_path);
  double size = fs.has("object_size") ? fs.num("object_size") : 0.1;
  {
    std::ifstream size_reader(sd->gt_path + "/size.txt");
    double v = -1;
    if (size_reader.is_open() && (size_reader >> v)) {
      if (v < 0) {
        std::cout << "no size. Skip." << std::endl; // :858-862
        return 0;
      }
      size = v;
    } else {
      write_text(sd->gt_path + "/size.txt", std::to_string(size) + "\n");
    }
  }
  std::vector<Vec3> pts(n);
  for (size_t i = 0; i < n; i++) {
    for (int a = 0; a < 3; a++) xyz[3 * i + a] = (float)(xyz[3 * i + a] * size / predicted_raw);
    pts[i] = Vec3(xyz[3 * i], xyz[3 * i + 1], xyz[3 * i + 2]);
  }
  HipScorer none;
  NBV_Net_Labeler labeler(sd, Vec3(0, 0, 0), size, none);
  labeler.view_space->get_view_space(pts); // centroid + 17/16 x radius of the scaled cloud (View_Space.hpp:517-558)
  if (labeler.get_coverage() != 0) return -71;
  const std::string N = std::to_string(sd->num_of_views);
  prv_camset* ds = nullptr;
  if (prv_cameras_from_dataset_json(ctx, (sd->gt_path + "/" + N + ".json").c_str(), &ds) != PRV_OK) return -72;
  int w = 0, h = 0;
  prv_camset_size(ds, &w, &h);
  const int count = prv_camset_count(ds);
  const int point_size = fs.has("points_size_cloud") ? (int)fs.num("points_size_cloud") : 5; // yaml:18
  float* xyz_dev = nullptr;
  uint8_t *rgb_dev = nullptr, *img_dev = nullptr;
  int rc = prv_malloc(ctx, (void**)&xyz_dev, xyz.size() * 4);
  if (rc == PRV_OK) rc = prv_malloc(ctx, (void**)&rgb_dev, rgb.size());
  if (rc == PRV_OK) rc = prv_malloc(ctx, (void**)&img_dev, (size_t)w * h * 4);
  if (rc == PRV_OK) rc = prv_memcpy_h2d(ctx, xyz_dev, xyz.data(), xyz.size() * 4);
  if (rc == PRV_OK) rc = prv_memcpy_h2d(ctx, rgb_dev, rgb.data(), rgb.size());
  const Vec3 oc = labeler.view_space->object_center_world;
  const double scale = 0.5 / labeler.view_space->predicted_size; // the json's (main.cpp:1599-1602)
  const double offset[3] = {0.5 + oc.z, 0.5 + oc.x, 0.5 + oc.y};
  sd->access_directory(sd->gt_path + "/" + N);
  std::vector<uint8_t> px((size_t)w * h * 4);
  for (int i = 0; rc == PRV_OK && i < count; i++) {
    rc = prv_splat_points(ctx, xyz_dev, rgb_dev, n, scale, offset, ds, &i, 1, w, h, point_size, 1, img_dev);
    if (rc == PRV_OK) rc = prv_memcpy_d2h(ctx, px.data(), img_dev, px.size());
    if (rc == PRV_OK && png_write_rgba8(sd->gt_path + "/" + N + "/rgbaClip_" + std::to_string(i) + ".png", w, h, px.data()) != 0) rc = PRV_E_IO;
  }
  if (rc != PRV_OK) std::cerr << "prv: " << prv_last_error(ctx) << std::endl;
  for (void* p : {(void*)xyz_dev, (void*)rgb_dev, (void*)img_dev})
    if (p) prv_free(ctx, p);
  prv_camset_destroy(ds);
  std::cout << "object " << name << ": " << n << " points, size " << size << " m, " << count << " views" << std::endl;
  return rc;
}

// mode 20 (main.cpp:3622-3832): for every view set <viewspace_path>/<N>.txt, N = 3..100, the shortest open tour
// over the unit view sphere from the (0,0,1) view, written to <N>_path.txt, one id per line (:3826-3830).
// No GPU involved.  Sets without a (0,0,1) view are reported and skipped ("can not find now view id", :3648).
int get_path_plan(const std::string& cfg) {
  FileStorage fs;
  if (!fs.open(cfg)) {
    std::cerr << "cannot open " << cfg << std::endl;
    return 5;
  }
  const std::string vs = fs.str("viewspace_path");
  const std::string out_dir = fs.has("path_plan_out") ? fs.str("path_plan_out") : vs; // the reference writes next to the inputs
  int written = 0;
  for (int n = 3; n <= 100; n++) {
    std::ifstream fin(vs + std::to_string(n) + ".txt");
    if (!fin.is_open()) continue;
    std::vector<View> views;
    std::vector<int> labels;
    int now_view_id = -1;
    for (int i = 0; i < n; i++) {
      double x = 0, y = 0, z = 0;
      fin >> x >> y >> z;
      views.emplace_back(Vec3(x, y, z));
      if (std::fabs(x) < 1e-6 && std::fabs(y) < 1e-6 && std::fabs(z - 1) < 1e-6) now_view_id = i;
      labels.push_back(i);
    }
    if (now_view_id == -1) {
      std::cout << "can not find now view id (" << n << ".txt)" << std::endl;
      continue;
    }
    Global_Path_Planner gpp(views, labels, now_view_id, Vec3(1e-10, 1e-10, 1e-10), 0.0);
    const double total_dis = gpp.solve();
    std::string text;
    for (int id : gpp.get_path_id_set()) text += std::to_string(id) + "\n";
    if (!write_text(out_dir + std::to_string(n) + "_path.txt", text)) return -3;
    std::cout << "view space " << n << " total dis is: " << total_dis << (gpp.exact ? "" : " (heuristic)") << std::endl;
    written++;
  }
  return written > 0 ? 0 : -1;
}

int env_int(const char* key, int fallback) {
  const char* v = getenv(key);
  return v && *v ? atoi(v) : fallback;
}

} // namespace

// PRV_SEGV_TRACE=1 (diagnostics): a crash prints where it happened before the process dies -- the phase of the process
// (running / shutting the runtime down / returning from main / exit handlers), the faulting thread, the fault address, the
// program counter and the mapping it lies in (a PC in no mapping = code that has been unloaded under a running thread)
#include <csignal>
#include <execinfo.h>
#include <sys/syscall.h>
#include <ucontext.h>
#include <unistd.h>
static volatile int g_phase = 0; // 0 run(), 1 inside prv_runtime_shutdown, 2 main is returning, 3 exit handlers have started
static void segv_trace(int sig, siginfo_t* info, void* uc_) {
  char buf[512];
  const ucontext_t* uc = (const ucontext_t*)uc_;
  const unsigned long long pc = uc ? (unsigned long long)uc->uc_mcontext.gregs[REG_RIP] : 0ull;
  const long tid = syscall(SYS_gettid);
  int n = snprintf(buf, sizeof(buf), "prv_planner: fatal signal %d in phase %d, thread %ld (%s), fault address %p, pc 0x%llx\n", sig,
                   g_phase, tid, tid == (long)getpid() ? "main" : "not main", info ? info->si_addr : nullptr, pc);
  (void)!write(2, buf, (size_t)n);
  if (FILE* maps = fopen("/proc/self/maps", "r")) { // not async-signal-safe; the process is dying anyway
    bool found = false;
    while (fgets(buf, sizeof(buf), maps)) {
      unsigned long long lo = 0, hi = 0;
      if (sscanf(buf, "%llx-%llx", &lo, &hi) == 2 && pc >= lo && pc < hi) {
        (void)!write(2, "  pc lies in: ", 14);
        (void)!write(2, buf, strlen(buf));
        found = true;
      }
    }
    fclose(maps);
    if (!found) (void)!write(2, "  pc lies in NO mapping (unloaded code)\n", 40);
  }
  void* frames[64];
  const int nf = backtrace(frames, 64);
  backtrace_symbols_fd(frames, nf, 2);
  signal(sig, SIG_DFL);
  raise(sig);
}
static void install_trace() {
  struct sigaction sa;
  memset(&sa, 0, sizeof(sa));
  sa.sa_sigaction = segv_trace;
  sa.sa_flags = SA_SIGINFO | SA_NODEFER;
  sigaction(SIGSEGV, &sa, nullptr);
  sigaction(SIGBUS, &sa, nullptr);
  sigaction(SIGABRT, &sa, nullptr);
  atexit([] { g_phase = 3; });
}

static int run(int argc, char** argv);

// Every context has been destroyed and every file closed when run() returns.  What an ordinary exit then adds is the
// static teardown of the HIP / HSA runtime, which this process does not control: 4 of 19,300 exits that shut the runtime
// down in order (prv_runtime_shutdown: synchronise + hipDeviceReset) and then returned from main still died AFTER main
// had returned, on a worker thread of the runtime itself (profiles/NOTES.md, round 3).  A planner run that finished
// correctly must not report SIGSEGV, so the DEFAULT exit is: ordered prv_runtime_shutdown, flush every stream, _exit(rc)
// -- the runtime's exit handlers never run.  PRV_PLANNER_EXIT=normal: the same shutdown, then an ordinary return (static
// destructors and atexit handlers run; chosen automatically under a profiler, which writes its files from an exit
// handler); =noreset: a plain return without the shutdown (round 2's behaviour, for the stress script only).
int main(int argc, char** argv) {
  if (getenv("PRV_SEGV_TRACE")) install_trace();
  const int rc = run(argc, argv);
  const char* how_c = getenv("PRV_PLANNER_EXIT");
  std::string how = how_c && *how_c ? how_c : "quick";
  if (how == "quick") {
    if (const char* pre = getenv("LD_PRELOAD"))
      if (std::string(pre).find("rocprof") != std::string::npos) how = "normal";
    for (char** e = environ; e && *e; e++)
      if (strncmp(*e, "ROCPROF", 7) == 0 || strncmp(*e, "ROCP_", 5) == 0 || strncmp(*e, "ROCTRACER", 9) == 0) how = "normal";
  }
  g_phase = 1;
  if (how != "noreset") (void)prv_runtime_shutdown();
  g_phase = 2;
  std::cout.flush();
  std::cerr.flush();
  fflush(nullptr);
  if (how == "quick") _exit(rc);
  return rc;
}

static int run(int argc, char** argv) {
  const std::string cfg = argc > 1 ? argv[1] : "../DefaultConfiguration.yaml";
  int mode = -1;
  std::cout << "input mode:" << std::endl;
  if (!(std::cin >> mode)) return 2;
  std::vector<std::string> names;
  std::cout << "input object names (-1 to stop):" << std::endl; // main.cpp:2299-2309
  std::string name;
  while (std::cin >> name && name != "-1") names.push_back(name);
  // Several GPUs, one process per GPU, launched by torchrun or by hand with RANK / WORLD_SIZE / LOCAL_RANK:
  //  shard objects (default; BASELINE config 5): the objects are independent, rank r takes the names i % world == r,
  //    no collective;
  //  shard views (`shard: views` in the yaml or PRV_SHARD=views; BASELINE config 4): every rank runs the loop of every
  //    object, the candidates of an iteration are dealt to the ranks, ONE all-gather of the score records per round
  //    (RCCL over xGMI; PRV_COMM=socket for ranks that share a GPU), trained members exchanged device to device;
  //  shard members (`shard: members` / PRV_SHARD=members; BASELINE config 5 with more GPUs than objects): the objects'
  //    loops in lockstep, the (object, member) trainings of a round dealt to the ranks round-robin (run_members_lockstep).
  const int world = env_int("WORLD_SIZE", 1), rank = env_int("RANK", 0);
  if (world < 1 || rank < 0 || rank >= world) {
    std::cerr << "RANK " << rank << " / WORLD_SIZE " << world << " make no sense" << std::endl;
    return 2;
  }
  bool shard_views = false, shard_members = false;
  {
    FileStorage fs0;
    const char* e = getenv("PRV_SHARD");
    std::string how = e && *e ? e : (fs0.open(cfg) && fs0.has("shard") ? fs0.str("shard") : "objects");
    if (how != "objects" && how != "views" && how != "members") {
      std::cerr << "shard must be 'objects', 'views' or 'members', not '" << how << "'" << std::endl;
      return 2;
    }
    shard_views = how == "views" && world > 1 && mode == ViewPlanning;
    shard_members = how == "members" && mode == ViewPlanning; // (with one rank: the lockstep walk alone, every training on this GPU)
  }
  if (world > 1 && !shard_views && !shard_members) {
    std::vector<std::string> mine;
    for (size_t i = 0; i < names.size(); i++)
      if ((int)(i % (size_t)world) == rank) mine.push_back(names[i]);
    names.swap(mine);
  }
  if (mode == GetPathPlan) return rank != 0 || get_path_plan(cfg) == 0 ? 0 : 1; // host only: before any GPU context exists
  if (mode != ViewPlanning && mode != InstantNGP && mode != GetCoverage) {
    std::cerr << "mode " << mode << " is outside the render/score path this build covers (21 = ViewPlanning, 4 = InstantNGP, "
                 "3 = GetCoverage, 20 = GetPathPlan)" << std::endl;
    return 3;
  }
  prv_ctx* ctx = nullptr;
  const int n_dev = prv_device_count();
  const int device = n_dev > 0 ? env_int("LOCAL_RANK", rank) % n_dev : 0; // fewer GPUs than ranks: they share
  if (prv_create(&ctx, device) != PRV_OK) {
    std::cerr << "prv: " << prv_last_error(nullptr) << std::endl;
    return 4;
  }
  FileStorage fs;
  if (!fs.open(cfg)) {
    std::cerr << "cannot open " << cfg << std::endl;
    return 5;
  }
  prv_comm* comm = nullptr;
  if (shard_views || (shard_members && world > 1)) {
    if (prv_comm_create(ctx, rank, world, nullptr, nullptr, &comm) != PRV_OK) {
      std::cerr << "prv: " << prv_last_error(ctx) << std::endl;
      prv_destroy(ctx);
      return 6;
    }
    std::cout << "rank " << rank << " of " << world << ": " << (shard_members ? "member trainings dealt round-robin, views sharded" : "views sharded") << ", transport "
              << prv_comm_transport(comm) << std::endl;
  }
  if (mode == InstantNGP || mode == GetCoverage) {
    int worst = 0;
    for (const auto& n : names) {
      const int rc = mode == InstantNGP ? instant_ngp_curves(ctx, cfg, n) : get_coverage_from_cloud(ctx, cfg, n);
      if (rc != 0) worst = rc;
    }
    prv_destroy(ctx);
    return worst == 0 ? 0 : 1;
  }
  // main.cpp:3838-3840 runs methods {4,0,1,2,3}; here: the configured one, or all that are in scope
  std::vector<int> methods;
  if (fs.has("method_of_IG") && fs.num("method_of_IG") >= 0) methods.push_back((int)fs.num("method_of_IG"));
  else methods = {RandomIterative, EnsembleRGB, EnsembleRGBDensity, PSNRCoverage};
  int worst = 0;
  if (shard_members) {
    for (int m : methods) {
      const int rc = run_members_lockstep(ctx, cfg, names, m, comm);
      if (rc != 0) worst = rc;
    }
  } else {
    for (const auto& n : names)
      for (int m : methods) {
        std::cout << "object " << n << " method " << m << std::endl;
        const int rc = view_planning(ctx, cfg, n, m, comm);
        if (rc != 0) worst = rc;
      }
  }
  if (comm) prv_comm_destroy(comm);
  prv_destroy(ctx);
  return worst == 0 ? 0 : 1;
}
