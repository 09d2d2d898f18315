// pcd_io.hpp -- the ground-truth cloud file of the reference (pcl::io::loadPCDFile<PointXYZRGB>, main.cpp:654):
// PCD v0.7, DATA ascii or binary (not binary_compressed), fields x y z [rgb | rgba] (+ others, skipped).
// rgb is PCL's packed 0x00RRGGBB, stored as a float (TYPE F) or an unsigned (TYPE U).
#pragma once
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <fstream>
#include <sstream>
#include <string>
#include <vector>

namespace prvhost {

// 0, or -1 io, -2 bad header, -3 unsupported (compressed / odd field sizes), -4 truncated
inline int pcd_read(const std::string& path, std::vector<float>& xyz, std::vector<uint8_t>& rgb) {
  std::ifstream f(path, std::ios::binary);
  if (!f.is_open()) return -1;
  std::vector<std::string> fields, types;
  std::vector<int> sizes, counts;
  size_t points = 0;
  std::string data_kind, line;
  while (std::getline(f, line)) {
    if (!line.empty() && line.back() == '\r') line.pop_back();
    if (line.empty() || line[0] == '#') continue;
    std::istringstream ss(line);
    std::string key;
    ss >> key;
    std::string tok;
    if (key == "FIELDS") while (ss >> tok) fields.push_back(tok);
    else if (key == "SIZE") while (ss >> tok) sizes.push_back(std::atoi(tok.c_str()));
    else if (key == "TYPE") while (ss >> tok) types.push_back(tok);
    else if (key == "COUNT") while (ss >> tok) counts.push_back(std::atoi(tok.c_str()));
    else if (key == "POINTS") ss >> points;
    else if (key == "DATA") {
      ss >> data_kind;
      break;
    }
  }
  const size_t nf = fields.size();
  if (nf == 0 || sizes.size() != nf || types.size() != nf || points == 0) return -2;
  if (counts.empty()) counts.assign(nf, 1);
  if (counts.size() != nf) return -2;
  if (data_kind != "ascii" && data_kind != "binary") return -3;
  for (size_t i = 0; i < nf; i++) // a header is input, not truth: sizes and counts decide offsets and column numbers
    if ((sizes[i] != 1 && sizes[i] != 2 && sizes[i] != 4 && sizes[i] != 8) || counts[i] < 1 || counts[i] > 1024) return -2;
  {
    // the body must be able to hold what the header announces (a point takes `stride` bytes, or at least two
    // characters per column as text) -- before anything is allocated for it
    size_t per_point = 0;
    for (size_t i = 0; i < nf; i++) per_point += data_kind == "binary" ? (size_t)sizes[i] * (size_t)counts[i] : 2u * (size_t)counts[i];
    const std::streampos here = f.tellg();
    f.seekg(0, std::ios::end);
    const std::streampos end = f.tellg();
    f.seekg(here);
    if (here < 0 || end < here || points > (size_t)(end - here) / per_point + 1) return -4;
  }
  int ix = -1, iy = -1, iz = -1, ic = -1;
  std::vector<size_t> offs(nf, 0);
  size_t stride = 0, cols = 0;
  std::vector<size_t> col0(nf, 0);
  for (size_t i = 0; i < nf; i++) {
    offs[i] = stride;
    col0[i] = cols;
    stride += (size_t)sizes[i] * counts[i];
    cols += (size_t)counts[i];
    if (fields[i] == "x") ix = (int)i;
    else if (fields[i] == "y") iy = (int)i;
    else if (fields[i] == "z") iz = (int)i;
    else if (fields[i] == "rgb" || fields[i] == "rgba") ic = (int)i;
  }
  if (ix < 0 || iy < 0 || iz < 0 || sizes[ix] != 4 || sizes[iy] != 4 || sizes[iz] != 4 || types[ix] != "F") return -3;
  if (ic >= 0 && sizes[ic] != 4) return -3;
  xyz.resize(points * 3);
  rgb.assign(points * 3, 200); // clouds without colour render light grey
  auto unpack = [&](uint32_t v, size_t i) {
    rgb[3 * i] = (uint8_t)(v >> 16);
    rgb[3 * i + 1] = (uint8_t)(v >> 8);
    rgb[3 * i + 2] = (uint8_t)v;
  };
  if (data_kind == "binary") {
    std::vector<uint8_t> rec(stride);
    for (size_t i = 0; i < points; i++) {
      if (!f.read((char*)rec.data(), (std::streamsize)stride)) return -4;
      memcpy(&xyz[3 * i], &rec[offs[ix]], 4);
      memcpy(&xyz[3 * i + 1], &rec[offs[iy]], 4);
      memcpy(&xyz[3 * i + 2], &rec[offs[iz]], 4);
      if (ic >= 0) {
        uint32_t v;
        memcpy(&v, &rec[offs[ic]], 4);
        unpack(v, i);
      }
    }
  } else {
    std::vector<std::string> tok(cols);
    for (size_t i = 0; i < points; i++) {
      if (!std::getline(f, line)) return -4;
      std::istringstream ss(line);
      for (size_t c = 0; c < cols; c++)
        if (!(ss >> tok[c])) return -4;
      xyz[3 * i] = std::strtof(tok[col0[ix]].c_str(), nullptr);
      xyz[3 * i + 1] = std::strtof(tok[col0[iy]].c_str(), nullptr);
      xyz[3 * i + 2] = std::strtof(tok[col0[iz]].c_str(), nullptr);
      if (ic >= 0) {
        uint32_t v;
        if (types[ic] == "F") { // the packed colour printed as a float: its BITS are the colour
          const float fv = std::strtof(tok[col0[ic]].c_str(), nullptr);
          memcpy(&v, &fv, 4);
        } else {
          v = (uint32_t)std::strtoul(tok[col0[ic]].c_str(), nullptr, 10);
        }
        unpack(v, i);
      }
    }
  }
  return 0;
}

} // namespace prvhost
