// robustness: corrupted PNG / PCD / JSON / YAML inputs must produce error codes, never crashes.  Build and run
// under the sanitizers from the repo root (host only, no GPU):
//   g++ -O1 -g -std=c++17 -fsanitize=address,undefined -o /tmp/fuzz scripts/fuzz_host_inputs.cpp \
//       nerf_prv_amd/host/host_api.cpp -lz && /tmp/fuzz 25000
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <string>
#include <vector>
extern "C" {
#include "../include/prv_host.h"
}
#include "../nerf_prv_amd/csrc/prv_json.hpp"
static uint64_t rng = 88172645463325252ull;
static uint32_t rnd() { rng ^= rng << 13; rng ^= rng >> 7; rng ^= rng << 17; return (uint32_t)(rng >> 11); }
static std::vector<uint8_t> slurp(const char* p) { std::ifstream f(p, std::ios::binary); return std::vector<uint8_t>((std::istreambuf_iterator<char>(f)), {}); }
static void spit(const char* p, const std::vector<uint8_t>& v) { std::ofstream f(p, std::ios::binary); f.write((const char*)v.data(), (std::streamsize)v.size()); }
static std::vector<uint8_t> mutate(std::vector<uint8_t> v) {
  const int kind = rnd() % 5;
  if (v.empty()) return v;
  if (kind == 4 && v.size() > 24) { // a PNG's IHDR width / height bytes (offsets 16..23): headers that promise more than the payload holds
    for (int k = 0; k < 1 + (int)(rnd() % 3); k++) v[16 + rnd() % 8] = (uint8_t)rnd();
    return v;
  }
  if (kind == 0) v.resize(rnd() % v.size());
  else if (kind == 1) for (int k = 0; k < 1 + (int)(rnd() % 8); k++) v[rnd() % v.size()] = (uint8_t)rnd();
  else if (kind == 2) { size_t a = rnd() % v.size(); size_t n = rnd() % 64; v.insert(v.begin() + a, n, (uint8_t)rnd()); }
  else { size_t a = rnd() % v.size(); size_t n = std::min<size_t>(v.size() - a, rnd() % 64); v.erase(v.begin() + a, v.begin() + a + n); }
  return v;
}
int main(int argc, char** argv) {
  const int iters = argc > 1 ? atoi(argv[1]) : 3000;
  // a valid PNG made by the library itself
  std::vector<uint8_t> img(37 * 23 * 4);
  for (auto& b : img) b = (uint8_t)rnd();
  prvh_png_write_rgba8("/tmp/prv_fuzz_ok.png", 37, 23, img.data());
  const auto png = slurp("/tmp/prv_fuzz_ok.png");
  int ok = 0, bad = 0;
  for (int i = 0; i < iters; i++) {
    spit("/tmp/prv_fuzz_m.png", mutate(png));
    int w = 0, h = 0;
    if (prvh_png_size("/tmp/prv_fuzz_m.png", &w, &h) == 0 && w > 0 && h > 0 && (long long)w * h < (1 << 24)) {
      std::vector<uint8_t> out((size_t)w * h * 4);
      (prvh_png_read_rgba8("/tmp/prv_fuzz_m.png", w, h, out.data()) == 0 ? ok : bad)++;
    } else bad++;
  }
  printf("png: %d decoded, %d refused\n", ok, bad);
  // PCD: ascii + binary
  {
    std::string a = "# .PCD v0.7\nVERSION 0.7\nFIELDS x y z rgb\nSIZE 4 4 4 4\nTYPE F F F U\nCOUNT 1 1 1 1\nWIDTH 5\nHEIGHT 1\nVIEWPOINT 0 0 0 1 0 0 0\nPOINTS 5\nDATA ascii\n";
    for (int k = 0; k < 5; k++) a += std::to_string(k * 0.1) + " 0.5 0.25 " + std::to_string(0x00ff8040 + k) + "\n";
    std::vector<uint8_t> pa(a.begin(), a.end());
    std::string bh = "VERSION 0.7\nFIELDS x y z rgb\nSIZE 4 4 4 4\nTYPE F F F F\nCOUNT 1 1 1 1\nWIDTH 4\nHEIGHT 1\nPOINTS 4\nDATA binary\n";
    std::vector<uint8_t> pb(bh.begin(), bh.end());
    for (int k = 0; k < 4 * 16; k++) pb.push_back((uint8_t)rnd());
    ok = bad = 0;
    for (int i = 0; i < iters; i++) {
      spit("/tmp/prv_fuzz_m.pcd", mutate(i & 1 ? pa : pb));
      long long n = prvh_pcd_read("/tmp/prv_fuzz_m.pcd", nullptr, nullptr, 0);
      if (n >= 0 && n < (1 << 20)) {
        std::vector<float> xyz((size_t)n * 3 + 3);
        std::vector<uint8_t> rgb((size_t)n * 3 + 3);
        (prvh_pcd_read("/tmp/prv_fuzz_m.pcd", xyz.data(), rgb.data(), n) >= 0 ? ok : bad)++;
      } else bad++;
    }
    printf("pcd: %d read, %d refused\n", ok, bad);
  }
  // JSON parser
  {
    const std::string j = "{\"camera_angle_x\": 1.2, \"w\": 80.0, \"h\": 45, \"scale\": 5.0, \"offset\": [0.5, 0.5, 0.5], \"frames\": [{\"file_path\": \"a/b_1.png\", \"transform_matrix\": [[1,0,0,0.1],[0,1,0,-2e-3],[0,0,1,3E+1],[0,0,0,1]]}], \"s\": \"x\\u00e9\\n\"}";
    std::vector<uint8_t> pj(j.begin(), j.end());
    ok = bad = 0;
    for (int i = 0; i < iters * 3; i++) {
      auto m = mutate(pj);
      std::string text(m.begin(), m.end()), err;
      prvjson::Value root;
      if (prvjson::Parser(text).parse(root, err)) {
        ok++;
        (void)root.at("frames").arr.size();
        (void)root.at("w").number();
      } else bad++;
    }
    printf("json: %d parsed, %d refused\n", ok, bad);
  }
  // YAML-dialect config reader through Share_Data
  {
    std::ifstream f("configs/DefaultConfiguration.yaml");
    std::string y((std::istreambuf_iterator<char>(f)), {});
    std::vector<uint8_t> py(y.begin(), y.end());
    ok = bad = 0;
    for (int i = 0; i < iters / 3; i++) {
      spit("/tmp/prv_fuzz_m.yaml", mutate(py));
      prvh_share_data* h = prvh_share_data_create("/tmp/prv_fuzz_m.yaml", "obj", -1, -1, -1);
      if (h) { ok++; prvh_share_data_destroy(h); } else bad++;
    }
    printf("yaml: %d accepted, %d refused\n", ok, bad);
  }
  return 0;
}
