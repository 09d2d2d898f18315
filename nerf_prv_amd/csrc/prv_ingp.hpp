// prv_ingp.hpp -- instant-ngp snapshots (testbed.save_snapshot / load_snapshot, Instantngp_scripts/run.py:123-127,
// 210-211) <-> this build's canonical field arrays.  Host only (no HIP): used by libprv_hip.so (prv_model_load_ingp /
// prv_model_save_ingp) and by libprv_host.so (prvh_ingp_read / prvh_ingp_write, testable without a GPU).
//
// LAYOUT ASSUMED FROM UPSTREAM, UNPINNED.  instant-ngp / tiny-cuda-nn are not in the reference tree and no snapshot can
// be produced in the build container (no pyngp, no CUDA), so nothing here has been checked against a real file.  What
// is implemented is the published format as of the upstream the reference's run.py targets (".ingp" snapshots):
//
//   file      = msgpack of the network-config json (+ "snapshot"); ".ingp" = the same, gzip-compressed (zstr)
//   root      : "encoding" {otype "HashGrid", n_levels, n_features_per_level, log2_hashmap_size, base_resolution,
//               per_level_scale}, "network" {FullyFusedMLP, ReLU, None, n_neurons 64, n_hidden_layers 1},
//               "rgb_network" {..., n_hidden_layers 2}, "dir_encoding" {SphericalHarmonics degree 4 (possibly
//               nested in a Composite)}
//   snapshot  : "version", "mode" "nerf", "n_params", "params_type" "__half", "params_binary" (bin, fp16),
//               "density_grid_size" 128, "density_grid_binary" (bin, fp16, Morton order, cascade-major),
//               "nerf" {"aabb_scale": 1, ...}
//   params    : NerfNetwork::set_params_impl order -- density MLP | rgb MLP | hash grid | (direction encoding: none);
//               FullyFusedMLP matrices row-major [out][in] (first layer 64x32, hidden 64x64, last 16x64 with the
//               output padded to 16 rows); grid = level-major, entry-major, features innermost, levels of
//               min(next_multiple(res^3, 8), 2^log2_hashmap_size) entries, dense index x + y*res + z*res^2, hashed
//               index (x ^ y*2654435761 ^ z*805459861) mod size -- this build's canonical table layout exactly
//   density   : optical thickness per cell; occupied iff value > min(0.01, mean of the non-negative values)
//               (update_density_grid_mean_and_bitfield)
// Everything else in a real file (optimizer state, camera, dataset, ...) is ignored on read and written minimally.
// What this build cannot represent is REFUSED with a message: aabb_scale != 1 (cascaded grids, warped positions),
// n_levels * n_features_per_level != 32, other MLP shapes / activations, a direction encoding wider than 16.
#pragma once
#include <zlib.h>

#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <map>
#include <memory>
#include <string>
#include <vector>

#include "prv_levels.hpp"

namespace prvingp {

// ---------------------------------------------------------------- a small msgpack tree
struct Value {
  enum Type { Nil, Bool, Int, Float, Str, Bin, Array, Map } type = Nil;
  bool b = false;
  int64_t i = 0;
  double f = 0.0;
  std::string s;              // Str, and Bin payload
  std::vector<Value> arr;
  std::vector<std::pair<std::string, Value>> map; // insertion order kept

  static Value Int64(int64_t v) { Value x; x.type = Int; x.i = v; return x; }
  static Value Real(double v) { Value x; x.type = Float; x.f = v; return x; }
  static Value String(const std::string& v) { Value x; x.type = Str; x.s = v; return x; }
  static Value Boolean(bool v) { Value x; x.type = Bool; x.b = v; return x; }
  static Value Binary(const void* p, size_t n) { Value x; x.type = Bin; x.s.assign((const char*)p, n); return x; }
  static Value Object() { Value x; x.type = Map; return x; }
  static Value List() { Value x; x.type = Array; return x; }

  const Value* find(const std::string& k) const {
    if (type != Map) return nullptr;
    for (const auto& kv : map)
      if (kv.first == k) return &kv.second;
    return nullptr;
  }
  Value& set(const std::string& k, Value v) {
    type = Map;
    for (auto& kv : map)
      if (kv.first == k) return kv.second = std::move(v);
    map.emplace_back(k, std::move(v));
    return map.back().second;
  }
  bool is_number() const { return type == Int || type == Float; }
  double number() const { return type == Int ? (double)i : f; }
};

constexpr size_t kMaxMsgpackItems = (size_t)1 << 20; // values in the tree (~150 B each): real snapshots hold 10^2..10^5
class Parser {
public:
  Parser(const uint8_t* p, size_t n) : p_(p), n_(n) {}
  bool parse(Value& out, std::string& err) {
    if (!value(out, 0)) {
      err = err_.empty() ? "truncated msgpack" : err_;
      return false;
    }
    return true;
  }

private:
  const uint8_t* p_;
  size_t n_, at_ = 0, n_items_ = 0;
  std::string err_;
  bool need(size_t k) { return n_ - at_ >= k; }
  uint64_t be(int k) {
    uint64_t v = 0;
    for (int j = 0; j < k; j++) v = (v << 8) | p_[at_ + j];
    at_ += k;
    return v;
  }
  bool bytes(size_t len, std::string& out) {
    if (!need(len)) return false;
    out.assign((const char*)p_ + at_, len);
    at_ += len;
    return true;
  }
  bool items(size_t count, bool is_map, Value& out, int depth) {
    if (count > n_ - at_) return false; // every item takes at least a byte: a header can not promise more than the file holds
    n_items_ += count;
    if (n_items_ > kMaxMsgpackItems) { // a few MB of one-byte items would otherwise become GBs of tree
      err_ = "msgpack tree has more than " + std::to_string(kMaxMsgpackItems) + " values";
      return false;
    }
    out.type = is_map ? Value::Map : Value::Array;
    for (size_t k = 0; k < count; k++) {
      if (is_map) {
        Value key;
        if (!value(key, depth + 1)) return false;
        if (key.type != Value::Str) {
          err_ = "map key is not a string";
          return false;
        }
        Value v;
        if (!value(v, depth + 1)) return false;
        out.map.emplace_back(std::move(key.s), std::move(v));
      } else {
        out.arr.emplace_back();
        if (!value(out.arr.back(), depth + 1)) return false;
      }
    }
    return true;
  }
  bool value(Value& out, int depth) {
    if (depth > 64) {
      err_ = "msgpack nested too deep";
      return false;
    }
    if (!need(1)) return false;
    const uint8_t t = p_[at_++];
    if (t <= 0x7f) { out = Value::Int64(t); return true; }
    if (t >= 0xe0) { out = Value::Int64((int8_t)t); return true; }
    if ((t & 0xf0) == 0x80) return items(t & 0x0f, true, out, depth);
    if ((t & 0xf0) == 0x90) return items(t & 0x0f, false, out, depth);
    if ((t & 0xe0) == 0xa0) { out.type = Value::Str; return bytes(t & 0x1f, out.s); }
    switch (t) {
      case 0xc0: out.type = Value::Nil; return true;
      case 0xc2: out = Value::Boolean(false); return true;
      case 0xc3: out = Value::Boolean(true); return true;
      case 0xc4: case 0xc5: case 0xc6: { // bin 8 / 16 / 32
        const int k = 1 << (t - 0xc4);
        if (!need(k)) return false;
        const size_t len = (size_t)be(k);
        out.type = Value::Bin;
        return bytes(len, out.s);
      }
      case 0xc7: case 0xc8: case 0xc9: { // ext 8 / 16 / 32: kept as binary (nlohmann writes bin with a subtype this way)
        const int k = 1 << (t - 0xc7);
        if (!need(k + 1)) return false;
        const size_t len = (size_t)be(k);
        at_++; // type byte
        out.type = Value::Bin;
        return bytes(len, out.s);
      }
      case 0xca: { if (!need(4)) return false; const uint32_t u = (uint32_t)be(4); float f; memcpy(&f, &u, 4); out = Value::Real(f); return true; }
      case 0xcb: { if (!need(8)) return false; const uint64_t u = be(8); double f; memcpy(&f, &u, 8); out = Value::Real(f); return true; }
      case 0xcc: if (!need(1)) return false; out = Value::Int64((int64_t)be(1)); return true;
      case 0xcd: if (!need(2)) return false; out = Value::Int64((int64_t)be(2)); return true;
      case 0xce: if (!need(4)) return false; out = Value::Int64((int64_t)be(4)); return true;
      case 0xcf: if (!need(8)) return false; out = Value::Int64((int64_t)be(8)); return true;
      case 0xd0: if (!need(1)) return false; out = Value::Int64((int8_t)be(1)); return true;
      case 0xd1: if (!need(2)) return false; out = Value::Int64((int16_t)be(2)); return true;
      case 0xd2: if (!need(4)) return false; out = Value::Int64((int32_t)be(4)); return true;
      case 0xd3: if (!need(8)) return false; out = Value::Int64((int64_t)be(8)); return true;
      case 0xd4: case 0xd5: case 0xd6: case 0xd7: case 0xd8: { // fixext 1..16
        const size_t len = (size_t)1 << (t - 0xd4);
        if (!need(1 + len)) return false;
        at_++;
        out.type = Value::Bin;
        return bytes(len, out.s);
      }
      case 0xd9: case 0xda: case 0xdb: { // str 8 / 16 / 32
        const int k = 1 << (t - 0xd9);
        if (!need(k)) return false;
        const size_t len = (size_t)be(k);
        out.type = Value::Str;
        return bytes(len, out.s);
      }
      case 0xdc: if (!need(2)) return false; return items((size_t)be(2), false, out, depth);
      case 0xdd: if (!need(4)) return false; return items((size_t)be(4), false, out, depth);
      case 0xde: if (!need(2)) return false; return items((size_t)be(2), true, out, depth);
      case 0xdf: if (!need(4)) return false; return items((size_t)be(4), true, out, depth);
      default: err_ = "unknown msgpack type byte"; return false;
    }
  }
};

inline void put_be(std::vector<uint8_t>& o, uint64_t v, int k) {
  for (int j = k - 1; j >= 0; j--) o.push_back((uint8_t)(v >> (8 * j)));
}
inline void write_msgpack(const Value& v, std::vector<uint8_t>& o) {
  switch (v.type) {
    case Value::Nil: o.push_back(0xc0); break;
    case Value::Bool: o.push_back(v.b ? 0xc3 : 0xc2); break;
    case Value::Int:
      if (v.i >= 0 && v.i <= 0x7f) o.push_back((uint8_t)v.i);
      else if (v.i < 0 && v.i >= -32) o.push_back((uint8_t)(int8_t)v.i);
      else if (v.i >= 0 && v.i <= 0xffffffffll) { o.push_back(0xce); put_be(o, (uint64_t)v.i, 4); }
      else { o.push_back(0xd3); put_be(o, (uint64_t)v.i, 8); }
      break;
    case Value::Float: {
      const float f32 = (float)v.f;
      if ((double)f32 == v.f) { uint32_t u; memcpy(&u, &f32, 4); o.push_back(0xca); put_be(o, u, 4); } // as nlohmann does
      else { uint64_t u; memcpy(&u, &v.f, 8); o.push_back(0xcb); put_be(o, u, 8); }
      break;
    }
    case Value::Str:
      if (v.s.size() < 32) o.push_back((uint8_t)(0xa0 | v.s.size()));
      else if (v.s.size() < 256) { o.push_back(0xd9); put_be(o, v.s.size(), 1); }
      else if (v.s.size() < 65536) { o.push_back(0xda); put_be(o, v.s.size(), 2); }
      else { o.push_back(0xdb); put_be(o, v.s.size(), 4); }
      o.insert(o.end(), v.s.begin(), v.s.end());
      break;
    case Value::Bin:
      if (v.s.size() < 256) { o.push_back(0xc4); put_be(o, v.s.size(), 1); }
      else if (v.s.size() < 65536) { o.push_back(0xc5); put_be(o, v.s.size(), 2); }
      else { o.push_back(0xc6); put_be(o, v.s.size(), 4); }
      o.insert(o.end(), v.s.begin(), v.s.end());
      break;
    case Value::Array:
      if (v.arr.size() < 16) o.push_back((uint8_t)(0x90 | v.arr.size()));
      else if (v.arr.size() < 65536) { o.push_back(0xdc); put_be(o, v.arr.size(), 2); }
      else { o.push_back(0xdd); put_be(o, v.arr.size(), 4); }
      for (const auto& e : v.arr) write_msgpack(e, o);
      break;
    case Value::Map:
      if (v.map.size() < 16) o.push_back((uint8_t)(0x80 | v.map.size()));
      else if (v.map.size() < 65536) { o.push_back(0xde); put_be(o, v.map.size(), 2); }
      else { o.push_back(0xdf); put_be(o, v.map.size(), 4); }
      for (const auto& kv : v.map) {
        write_msgpack(Value::String(kv.first), o);
        write_msgpack(kv.second, o);
      }
      break;
  }
}

// ---------------------------------------------------------------- gzip (".ingp" = zstr-compressed msgpack)
// limit: the largest snapshot this build can represent is a < 4 GiB fp16 table + the density grid; upstream adds the
// optimiser's two fp32 moments when asked to (4 x the parameters).  2.5 GiB covers every configuration that passes the
// checks below several times over and keeps a hostile file from inflating into the host's memory.
constexpr size_t kMaxInflatedBytes = ((size_t)5 << 30) / 2;
inline bool gunzip(const std::vector<uint8_t>& in, std::vector<uint8_t>& out, size_t limit = kMaxInflatedBytes) {
  z_stream z{};
  if (inflateInit2(&z, 15 + 32) != Z_OK) return false; // zlib or gzip header, auto-detected
  z.next_in = const_cast<Bytef*>(in.data());
  z.avail_in = (uInt)std::min<size_t>(in.size(), 0xffffffffu);
  size_t fed = z.avail_in;
  std::vector<uint8_t> buf((size_t)1 << 20);
  int rc = Z_OK;
  while (rc != Z_STREAM_END) {
    z.next_out = buf.data();
    z.avail_out = (uInt)buf.size();
    rc = inflate(&z, Z_NO_FLUSH);
    if (rc != Z_OK && rc != Z_STREAM_END && rc != Z_BUF_ERROR) break;
    out.insert(out.end(), buf.data(), buf.data() + (buf.size() - z.avail_out));
    if (out.size() > limit) { rc = Z_MEM_ERROR; break; }
    if (z.avail_in == 0 && fed < in.size()) {
      z.next_in = const_cast<Bytef*>(in.data() + fed);
      z.avail_in = (uInt)std::min<size_t>(in.size() - fed, 0xffffffffu);
      fed += z.avail_in;
    } else if (rc == Z_BUF_ERROR && z.avail_in == 0) {
      break; // truncated stream
    }
  }
  inflateEnd(&z);
  return rc == Z_STREAM_END;
}
inline bool gzip(const std::vector<uint8_t>& in, std::vector<uint8_t>& out) {
  z_stream z{};
  if (deflateInit2(&z, Z_DEFAULT_COMPRESSION, Z_DEFLATED, 15 + 16, 8, Z_DEFAULT_STRATEGY) != Z_OK) return false;
  out.resize(deflateBound(&z, (uLong)in.size()) + 64);
  z.next_in = const_cast<Bytef*>(in.data());
  z.avail_in = (uInt)in.size();
  z.next_out = out.data();
  z.avail_out = (uInt)out.size();
  const int rc = deflate(&z, Z_FINISH);
  out.resize(out.size() - z.avail_out);
  deflateEnd(&z);
  return rc == Z_STREAM_END && in.size() < 0xffffffffu;
}

// ---------------------------------------------------------------- the field
struct Field {
  prv_field_desc desc{};
  std::vector<uint16_t> table, mlp; // canonical: table level-major; mlp [in][out], layers d1 d2 r1 r2 r3
  std::vector<uint32_t> occ;        // occ_res^3 bits, x fastest
};

constexpr int kLayerIn[5] = {32, 64, 32, 64, 64};
constexpr int kLayerOut[5] = {64, 16, 64, 64, 16};
constexpr int kMlpHalfs = 10240;
constexpr float kMinOpticalThickness = 0.01f; // NERF_MIN_OPTICAL_THICKNESS()

inline float half_to_float(uint16_t h) {
  const uint32_t s = (uint32_t)(h >> 15) << 31, e = (h >> 10) & 31u, m = h & 1023u;
  uint32_t u;
  if (e == 0) {
    if (m == 0) u = s;
    else { // subnormal
      int sh = 0;
      uint32_t mm = m;
      while (!(mm & 1024u)) { mm <<= 1; sh++; }
      u = s | ((uint32_t)(113 - sh) << 23) | ((mm & 1023u) << 13);
    }
  } else if (e == 31) u = s | 0x7f800000u | (m << 13);
  else u = s | ((e + 112u) << 23) | (m << 13);
  float f;
  memcpy(&f, &u, 4);
  return f;
}
inline uint16_t float_to_half(float f) { // round to nearest even
  uint32_t u;
  memcpy(&u, &f, 4);
  const uint32_t s = (u >> 16) & 0x8000u;
  const int32_t e = (int32_t)((u >> 23) & 255u) - 127 + 15;
  uint32_t m = u & 0x7fffffu;
  if (((u >> 23) & 255u) == 255u) return (uint16_t)(s | 0x7c00u | (m ? 0x200u : 0u));
  if (e >= 31) return (uint16_t)(s | 0x7c00u);
  if (e <= 0) {
    if (e < -10) return (uint16_t)s;
    m |= 0x800000u;
    const int sh = 14 - e;
    uint32_t h = m >> sh;
    const uint32_t rem = m & ((1u << sh) - 1u), half = 1u << (sh - 1);
    if (rem > half || (rem == half && (h & 1u))) h++;
    return (uint16_t)(s | h);
  }
  uint32_t h = ((uint32_t)e << 10) | (m >> 13);
  const uint32_t rem = m & 0x1fffu;
  if (rem > 0x1000u || (rem == 0x1000u && (h & 1u))) h++;
  return (uint16_t)(s | h);
}

inline uint32_t expand_bits(uint32_t v) { // instant-ngp's morton3D helper
  v = (v * 0x00010001u) & 0xFF0000FFu;
  v = (v * 0x00000101u) & 0x0F00F00Fu;
  v = (v * 0x00000011u) & 0xC30C30C3u;
  v = (v * 0x00000005u) & 0x49249249u;
  return v;
}
inline uint32_t morton3(uint32_t x, uint32_t y, uint32_t z) { return expand_bits(x) | (expand_bits(y) << 1) | (expand_bits(z) << 2); }

inline std::string lower(std::string s) {
  for (auto& c : s) c = (char)tolower((unsigned char)c);
  return s;
}

inline bool read_file(const std::string& path, std::vector<uint8_t>& out) {
  FILE* f = fopen(path.c_str(), "rb");
  if (!f) return false;
  uint8_t buf[1 << 16];
  size_t n;
  while ((n = fread(buf, 1, sizeof(buf), f)) > 0) out.insert(out.end(), buf, buf + n);
  fclose(f);
  return true;
}

// 0 ok; -3 (PRV_E_IO) unreadable / malformed; -1 (PRV_E_INVALID) well-formed but not representable here
inline int read_snapshot(const std::string& path, Field& out, std::string& err) {
  std::vector<uint8_t> file, raw;
  if (!read_file(path, file)) { err = "cannot read " + path; return PRV_E_IO; }
  if (file.size() >= 2 && file[0] == 0x1f && file[1] == 0x8b) {
    if (!gunzip(file, raw)) { err = path + ": corrupt gzip stream"; return PRV_E_IO; }
  } else raw.swap(file);
  Value root;
  std::string perr;
  if (!Parser(raw.data(), raw.size()).parse(root, perr) || root.type != Value::Map) {
    err = path + ": not a msgpack snapshot (" + (perr.empty() ? "top level is not a map" : perr) + ")";
    return PRV_E_IO;
  }
  auto num = [&](const Value* m, const char* k, double dflt) { const Value* v = m ? m->find(k) : nullptr; return v && v->is_number() ? v->number() : dflt; };
  auto str = [&](const Value* m, const char* k, const char* dflt) { const Value* v = m ? m->find(k) : nullptr; return v && v->type == Value::Str ? v->s : std::string(dflt); };
  const Value *enc = root.find("encoding"), *net = root.find("network"), *rgb = root.find("rgb_network"), *dir = root.find("dir_encoding"),
              *snap = root.find("snapshot");
  if (!enc || !net || !rgb || !snap) { err = path + ": encoding / network / rgb_network / snapshot missing"; return PRV_E_IO; }
  // ---- what this build can represent
  if (lower(str(enc, "otype", "")).find("grid") == std::string::npos) { err = "encoding otype '" + str(enc, "otype", "") + "' is not a hash grid"; return PRV_E_INVALID; }
  const std::string gtype = lower(str(enc, "type", "hash"));
  if (gtype != "hash") { err = "grid type '" + gtype + "' (only 'hash' is supported)"; return PRV_E_INVALID; }
  const std::string interp = lower(str(enc, "interpolation", "linear"));
  if (interp != "linear") { err = "grid interpolation '" + interp + "' (only 'linear')"; return PRV_E_INVALID; }
  prv_field_desc d{};
  d.n_features = (int)num(enc, "n_features_per_level", 2);
  d.n_levels = enc->find("n_levels") ? (int)num(enc, "n_levels", 16) : (d.n_features > 0 ? (int)num(enc, "n_features", 32) / d.n_features : 0);
  d.log2_hashmap = (int)num(enc, "log2_hashmap_size", 19);
  d.base_res = (int)num(enc, "base_resolution", 16);
  d.per_level_scale = (float)num(enc, "per_level_scale", 2.0); // tiny-cuda-nn's default
  d.occ_res = (int)num(snap, "density_grid_size", 128);
  d.density_bias = 0.0f; // upstream: density = exp(network output), no bias
  d.table_amp = 0.0f;
  if (d.n_levels * d.n_features != 32 || (d.n_features != 2 && d.n_features != 4)) {
    err = "n_levels " + std::to_string(d.n_levels) + " x n_features_per_level " + std::to_string(d.n_features) + " != 32 features (the fused kernels' width)";
    return PRV_E_INVALID;
  }
  if (d.n_levels == 1) d.per_level_scale = std::max(d.per_level_scale, 1.0f);
  {
    const double fin = (double)d.base_res * std::pow((double)d.per_level_scale, (double)(d.n_levels - 1));
    d.finest_res = (int)std::min(4096.0, std::max((double)d.base_res, std::floor(fin + 0.5))); // a label; the levels follow per_level_scale
    if (fin > 4096.5) { err = "finest level resolution " + std::to_string((long long)fin) + " exceeds 4096"; return PRV_E_INVALID; }
  }
  struct { const Value* v; const char* name; int hidden; } mlps[2] = {{net, "network", 1}, {rgb, "rgb_network", 2}};
  for (auto& m : mlps) {
    const std::string ot = lower(str(m.v, "otype", "fullyfusedmlp"));
    if (ot != "fullyfusedmlp" && ot != "cutlassmlp") { err = std::string(m.name) + " otype '" + ot + "'"; return PRV_E_INVALID; }
    if ((int)num(m.v, "n_neurons", 64) != 64 || (int)num(m.v, "n_hidden_layers", m.hidden) != m.hidden) {
      err = std::string(m.name) + ": only n_neurons 64 with " + std::to_string(m.hidden) + " hidden layer(s) is representable";
      return PRV_E_INVALID;
    }
    if (lower(str(m.v, "activation", "relu")) != "relu" || lower(str(m.v, "output_activation", "none")) != "none") {
      err = std::string(m.name) + ": activation must be ReLU, output_activation None";
      return PRV_E_INVALID;
    }
  }
  if (dir) { // SphericalHarmonics degree 4, directly or nested in a Composite whose other parts carry no dimensions
    const Value* sh = dir;
    int extra = 0;
    if (lower(str(dir, "otype", "")) == "composite") {
      sh = nullptr;
      const Value* nested = dir->find("nested");
      if (nested && nested->type == Value::Array)
        for (const auto& e : nested->arr) {
          if (lower(str(&e, "otype", "")) == "sphericalharmonics") sh = &e;
          else extra += (int)num(&e, "n_dims_to_encode", 0);
        }
    }
    if (!sh || lower(str(sh, "otype", "")) != "sphericalharmonics" || (int)num(sh, "degree", 4) != 4 || extra != 0) {
      err = "dir_encoding must be spherical harmonics of degree 4 (16 coefficients) with no extra dimensions";
      return PRV_E_INVALID;
    }
  }
  const std::string mode = lower(str(snap, "mode", "nerf"));
  if (mode != "nerf") { err = "snapshot mode '" + mode + "' is not nerf"; return PRV_E_INVALID; }
  const Value* nerf = snap->find("nerf");
  const int aabb_scale = (int)num(nerf, "aabb_scale", 1);
  if (aabb_scale != 1) {
    err = "aabb_scale " + std::to_string(aabb_scale) + ": cascaded density grids / warped positions are not representable (the reference sets ray_casting_aabb_scale: 1)";
    return PRV_E_INVALID;
  }
  if (str(snap, "params_type", "__half") != "__half") { err = "params_type '" + str(snap, "params_type", "") + "' (only __half)"; return PRV_E_INVALID; }
  if (d.occ_res != 128 && (d.occ_res < 8 || d.occ_res > 512 || (d.occ_res & (d.occ_res - 1)))) { err = "density_grid_size " + std::to_string(d.occ_res); return PRV_E_INVALID; }
  prv::HostLevel lv[prv::kMaxFieldLevels];
  uint64_t total = 0;
  if (prv::compute_levels(d, lv, &total) != 0) { err = "the grid configuration is outside this build's limits"; return PRV_E_INVALID; }
  // ---- parameters
  const Value* pb = snap->find("params_binary");
  if (!pb || pb->type != Value::Bin) { err = "snapshot.params_binary missing"; return PRV_E_IO; }
  const uint64_t n_grid = total * (uint64_t)d.n_features, n_params = (uint64_t)kMlpHalfs + n_grid;
  const uint64_t declared = (uint64_t)num(snap, "n_params", (double)n_params);
  if (pb->s.size() != n_params * 2 || declared != n_params) {
    err = "params_binary holds " + std::to_string(pb->s.size() / 2) + " values (n_params " + std::to_string(declared) + "), this configuration has " +
          std::to_string(n_params) + " (10240 MLP + " + std::to_string(n_grid) + " grid)";
    return PRV_E_IO;
  }
  const uint16_t* p = (const uint16_t*)pb->s.data();
  out.desc = d;
  out.mlp.assign(kMlpHalfs, 0);
  size_t src = 0, dst = 0;
  for (int l = 0; l < 5; l++) { // upstream [out][in] row-major -> canonical [in][out]
    for (int o = 0; o < kLayerOut[l]; o++)
      for (int i = 0; i < kLayerIn[l]; i++) out.mlp[dst + (size_t)i * kLayerOut[l] + o] = p[src + (size_t)o * kLayerIn[l] + i];
    src += (size_t)kLayerIn[l] * kLayerOut[l];
    dst += (size_t)kLayerIn[l] * kLayerOut[l];
  }
  out.table.assign(p + kMlpHalfs, p + kMlpHalfs + n_grid);
  // ---- occupancy from the density grid (cascade 0)
  const Value* dg = snap->find("density_grid_binary");
  const uint64_t R = (uint64_t)d.occ_res, cells = R * R * R;
  out.occ.assign((cells + 31) / 32, 0u);
  if (!dg || dg->type != Value::Bin || dg->s.size() < cells * 2) { err = "snapshot.density_grid_binary missing or shorter than density_grid_size^3"; return PRV_E_IO; }
  const uint16_t* g = (const uint16_t*)dg->s.data();
  // upstream's update_density_grid_mean_and_bitfield: mean = sum(max(v, 0)) / N over ALL N cells of cascade 0 -- the
  // never-seen cells (-1) add nothing to the sum but do count in the denominator
  double sum = 0.0;
  for (uint64_t i = 0; i < cells; i++) sum += (double)std::max(half_to_float(g[i]), 0.0f);
  const float mean = (float)(sum / (double)cells);
  const float thresh = std::min(kMinOpticalThickness, mean);
  for (uint32_t z = 0; z < R; z++)
    for (uint32_t y = 0; y < R; y++)
      for (uint32_t x = 0; x < R; x++)
        if (half_to_float(g[morton3(x, y, z)]) > thresh) {
          const uint64_t bit = x + R * (y + R * z);
          out.occ[bit >> 5] |= 1u << (bit & 31);
        }
  return 0;
}

// The inverse: a snapshot another reader of the same format (and this one) loads back.  occupied cells are written
// with a density-grid value of 1 (> any threshold), empty ones with 0.
inline int write_snapshot(const std::string& path, const Field& in, std::string& err) {
  const prv_field_desc& d = in.desc;
  prv::HostLevel lv[prv::kMaxFieldLevels];
  uint64_t total = 0;
  if (prv::compute_levels(d, lv, &total) != 0) { err = "invalid field descriptor"; return PRV_E_INVALID; }
  if (!(d.per_level_scale > 0.0f)) {
    err = "only fields whose levels follow tiny-cuda-nn's recipe (per_level_scale > 0) can be written as instant-ngp snapshots: "
          "a base_res/finest_res field has level scales no per_level_scale reproduces exactly";
    return PRV_E_INVALID;
  }
  if (d.density_bias != 0.0f) { err = "density_bias must be 0 (upstream: density = exp(network output))"; return PRV_E_INVALID; }
  const uint64_t n_grid = total * (uint64_t)d.n_features, R = (uint64_t)d.occ_res, cells = R * R * R;
  if (in.table.size() != n_grid || in.mlp.size() != (size_t)kMlpHalfs || in.occ.size() != (cells + 31) / 32) { err = "array sizes do not match the descriptor"; return PRV_E_INVALID; }
  if (d.occ_res & (d.occ_res - 1)) { err = "density grid size must be a power of two (Morton order)"; return PRV_E_INVALID; }
  std::vector<uint16_t> params((size_t)kMlpHalfs + n_grid);
  size_t src = 0, dst = 0;
  for (int l = 0; l < 5; l++) {
    for (int o = 0; o < kLayerOut[l]; o++)
      for (int i = 0; i < kLayerIn[l]; i++) params[dst + (size_t)o * kLayerIn[l] + i] = in.mlp[src + (size_t)i * kLayerOut[l] + o];
    src += (size_t)kLayerIn[l] * kLayerOut[l];
    dst += (size_t)kLayerIn[l] * kLayerOut[l];
  }
  memcpy(params.data() + kMlpHalfs, in.table.data(), n_grid * 2);
  std::vector<uint16_t> grid(cells, 0);
  const uint16_t one = float_to_half(1.0f);
  for (uint32_t z = 0; z < R; z++)
    for (uint32_t y = 0; y < R; y++)
      for (uint32_t x = 0; x < R; x++) {
        const uint64_t bit = x + R * (y + R * z);
        if ((in.occ[bit >> 5] >> (bit & 31)) & 1u) grid[morton3(x, y, z)] = one;
      }
  Value root = Value::Object();
  Value& enc = root.set("encoding", Value::Object());
  enc.set("otype", Value::String("HashGrid"));
  enc.set("n_levels", Value::Int64(d.n_levels));
  enc.set("n_features_per_level", Value::Int64(d.n_features));
  enc.set("log2_hashmap_size", Value::Int64(d.log2_hashmap));
  enc.set("base_resolution", Value::Int64(d.base_res));
  enc.set("per_level_scale", Value::Real(d.per_level_scale));
  auto mlp = [&](const char* key, int hidden) {
    Value& n = root.set(key, Value::Object());
    n.set("otype", Value::String("FullyFusedMLP"));
    n.set("activation", Value::String("ReLU"));
    n.set("output_activation", Value::String("None"));
    n.set("n_neurons", Value::Int64(64));
    n.set("n_hidden_layers", Value::Int64(hidden));
  };
  mlp("network", 1);
  mlp("rgb_network", 2);
  Value& de = root.set("dir_encoding", Value::Object());
  de.set("otype", Value::String("Composite"));
  Value& nested = de.set("nested", Value::List());
  {
    Value sh = Value::Object();
    sh.set("n_dims_to_encode", Value::Int64(3));
    sh.set("otype", Value::String("SphericalHarmonics"));
    sh.set("degree", Value::Int64(4));
    nested.arr.push_back(sh);
    Value id = Value::Object();
    id.set("otype", Value::String("Identity"));
    id.set("n_bins", Value::Int64(4));
    id.set("degree", Value::Int64(4));
    nested.arr.push_back(id);
  }
  Value& snap = root.set("snapshot", Value::Object());
  snap.set("version", Value::Int64(1));
  snap.set("mode", Value::String("nerf"));
  snap.set("n_params", Value::Int64((int64_t)params.size()));
  snap.set("params_type", Value::String("__half"));
  snap.set("params_binary", Value::Binary(params.data(), params.size() * 2));
  snap.set("density_grid_size", Value::Int64(d.occ_res));
  snap.set("density_grid_binary", Value::Binary(grid.data(), grid.size() * 2));
  Value& nerf = snap.set("nerf", Value::Object());
  nerf.set("aabb_scale", Value::Int64(1));
  snap.set("training_step", Value::Int64(0));
  std::vector<uint8_t> bytes, packed;
  write_msgpack(root, bytes);
  const bool zipped = path.size() >= 5 && lower(path.substr(path.size() - 5)) == ".ingp";
  if (zipped && !gzip(bytes, packed)) { err = "gzip failed"; return PRV_E_INTERNAL; }
  const std::vector<uint8_t>& o = zipped ? packed : bytes;
  FILE* f = fopen(path.c_str(), "wb");
  if (!f) { err = "cannot write " + path; return PRV_E_IO; }
  const bool ok = fwrite(o.data(), 1, o.size(), f) == o.size();
  if (fclose(f) != 0 || !ok) { err = "short write to " + path; return PRV_E_IO; }
  return 0;
}

} // namespace prvingp
