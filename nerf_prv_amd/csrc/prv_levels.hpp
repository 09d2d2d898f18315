// prv_levels.hpp -- the level table of a hash-grid field from its descriptor (host only, no HIP): shared by the C ABI
// (prv_api.cpp), the trainer and the instant-ngp snapshot reader / writer (prv_ingp.hpp).  Same published recipe as
// the oracle's orc_field_levels, written independently.
#pragma once
#include <cmath>
#include <cstdint>

#include "../../include/prv.h"

namespace prv {

constexpr int kMaxFieldLevels = 16;

struct HostLevel {
  float scale;
  uint32_t res, offset, size, hashed;
};

// per_level_scale == 0: levels grow geometrically from base_res to finest_res, computed in double with the nominal
//   integer resolutions hit exactly (this build's synthetic fields).
// per_level_scale  > 0: tiny-cuda-nn's own recipe in float32 (grid.h: grid_scale / grid_resolution) --
//   scale_l = exp2f(l * log2f(per_level_scale)) * base_res - 1, res_l = ceilf(scale_l) + 1 -- what an imported
//   instant-ngp snapshot was trained with.
inline int compute_levels(const prv_field_desc& d, HostLevel* lv, uint64_t* total) {
  if (d.n_levels < 1 || d.n_levels > kMaxFieldLevels) return -1;
  if (d.n_features != 2 && d.n_features != 4) return -1;
  if (d.n_levels * d.n_features != 32) return -1;
  if (d.log2_hashmap < 4 || d.log2_hashmap > 28) return -1;
  if (d.base_res < 2 || d.finest_res < d.base_res || d.finest_res > 4096) return -1;
  if (d.occ_res < 1 || d.occ_res > 1024) return -1;
  if (!(d.per_level_scale >= 0.0f) || d.per_level_scale > 16.0f || (d.per_level_scale > 0.0f && d.per_level_scale < 1.0f)) return -1;
  const double growth =
      d.n_levels > 1 ? std::exp((std::log((double)d.finest_res) - std::log((double)d.base_res)) / (d.n_levels - 1)) : 1.0;
  const float log2_pls = d.per_level_scale > 0.0f ? std::log2(d.per_level_scale) : 0.0f;
  const uint64_t T = 1ull << d.log2_hashmap;
  uint64_t off = 0;
  for (int l = 0; l < d.n_levels; l++) {
    if (d.per_level_scale > 0.0f) {
      lv[l].scale = exp2f((float)l * log2_pls) * (float)d.base_res - 1.0f;
      lv[l].res = (uint32_t)ceilf(lv[l].scale) + 1u;
      if (lv[l].res > 4097u) return -1;
    } else {
      double s = (double)d.base_res * std::pow(growth, (double)l) - 1.0;
      const double nearest = std::floor(s + 0.5);
      if (std::fabs(s - nearest) < 1e-9) s = nearest;
      lv[l].scale = (float)s;
      lv[l].res = (uint32_t)std::ceil(s) + 1u;
    }
    const uint64_t dense = (uint64_t)lv[l].res * lv[l].res * lv[l].res;
    lv[l].hashed = dense > T;
    lv[l].size = lv[l].hashed ? (uint32_t)T : (uint32_t)((dense + 7) & ~7ull);
    lv[l].offset = (uint32_t)off;
    off += lv[l].size;
  }
  if (off * (uint64_t)d.n_features * 2ull >= (1ull << 32)) return -1; // 32-bit byte offsets in the gather
  *total = off;
  return 0;
}

} // namespace prv
