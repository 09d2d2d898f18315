// png_io.hpp -- the PNG files on the reference's boundary (rgbaClip_<i>.png training images written by
// cv::imwrite at main.cpp:1617, candidate screenshots written by run.py:309 and read back by cv::imread at
// main.cpp:2047, 2107): 8-bit RGB / RGBA / grey(+alpha), non-interlaced, read and written with zlib only.
#pragma once
#include <zlib.h>

#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

namespace prvhost {

inline uint32_t png_be32(const uint8_t* p) { return ((uint32_t)p[0] << 24) | ((uint32_t)p[1] << 16) | ((uint32_t)p[2] << 8) | p[3]; }
inline void png_put32(std::vector<uint8_t>& v, uint32_t x) {
  v.push_back((uint8_t)(x >> 24));
  v.push_back((uint8_t)(x >> 16));
  v.push_back((uint8_t)(x >> 8));
  v.push_back((uint8_t)x);
}

// -> RGBA8, row-major, top row first.  Returns 0 or a negative code (-1 io, -2 not a PNG, -3 unsupported, -4 corrupt)
inline int png_read_rgba8(const std::string& path, int* width, int* height, std::vector<uint8_t>& out) {
  FILE* f = fopen(path.c_str(), "rb");
  if (!f) return -1;
  std::vector<uint8_t> file;
  uint8_t buf[65536];
  size_t n;
  while ((n = fread(buf, 1, sizeof(buf), f)) > 0) file.insert(file.end(), buf, buf + n);
  fclose(f);
  static const uint8_t sig[8] = {0x89, 'P', 'N', 'G', 0x0d, 0x0a, 0x1a, 0x0a};
  if (file.size() < 33 || memcmp(file.data(), sig, 8) != 0) return -2;
  uint32_t w = 0, h = 0;
  int depth = 0, ctype = -1, interlace = 0;
  std::vector<uint8_t> idat;
  size_t pos = 8;
  while (pos + 12 <= file.size()) {
    const uint32_t len = png_be32(&file[pos]);
    const char* type = (const char*)&file[pos + 4];
    if (pos + 12 + (size_t)len > file.size()) return -4;
    const uint8_t* data = &file[pos + 8];
    if (!memcmp(type, "IHDR", 4) && len >= 13) {
      w = png_be32(data);
      h = png_be32(data + 4);
      depth = data[8];
      ctype = data[9];
      interlace = data[12];
    } else if (!memcmp(type, "IDAT", 4)) {
      idat.insert(idat.end(), data, data + len);
    } else if (!memcmp(type, "IEND", 4)) {
      break;
    }
    pos += 12 + (size_t)len;
  }
  int ch = ctype == 0 ? 1 : ctype == 2 ? 3 : ctype == 4 ? 2 : ctype == 6 ? 4 : 0;
  if (w == 0 || h == 0 || w > 65535 || h > 65535 || depth != 8 || ch == 0 || interlace != 0) return -3;
  const size_t stride = (size_t)w * ch;
  // the header is input, not truth: deflate expands at most 1032:1 (+ a few bytes of framing), so an IDAT payload that
  // cannot inflate to the size the header claims is refused BEFORE anything of that size is allocated
  if ((stride + 1) * (size_t)h > idat.size() * 1032 + 1024) return -4;
  std::vector<uint8_t> raw((stride + 1) * h);
  uLongf raw_len = (uLongf)raw.size();
  if (uncompress(raw.data(), &raw_len, idat.data(), (uLong)idat.size()) != Z_OK || raw_len != raw.size()) return -4;
  std::vector<uint8_t> img(stride * h);
  for (uint32_t y = 0; y < h; y++) { // undo the per-row filters
    const uint8_t ft = raw[(stride + 1) * y];
    const uint8_t* src = &raw[(stride + 1) * y + 1];
    uint8_t* dst = &img[stride * y];
    const uint8_t* up = y ? &img[stride * (y - 1)] : nullptr;
    for (size_t i = 0; i < stride; i++) {
      const int a = i >= (size_t)ch ? dst[i - ch] : 0, b = up ? up[i] : 0, c = (up && i >= (size_t)ch) ? up[i - ch] : 0;
      int pred = 0;
      switch (ft) {
        case 0: pred = 0; break;
        case 1: pred = a; break;
        case 2: pred = b; break;
        case 3: pred = (a + b) >> 1; break;
        case 4: {
          const int p = a + b - c, pa = std::abs(p - a), pb = std::abs(p - b), pc = std::abs(p - c);
          pred = (pa <= pb && pa <= pc) ? a : (pb <= pc ? b : c);
          break;
        }
        default: return -4;
      }
      dst[i] = (uint8_t)(src[i] + pred);
    }
  }
  out.resize((size_t)w * h * 4);
  for (size_t i = 0; i < (size_t)w * h; i++) {
    const uint8_t* p = &img[i * ch];
    uint8_t* o = &out[i * 4];
    if (ch >= 3) { o[0] = p[0]; o[1] = p[1]; o[2] = p[2]; o[3] = ch == 4 ? p[3] : 255; }
    else { o[0] = o[1] = o[2] = p[0]; o[3] = ch == 2 ? p[1] : 255; }
  }
  *width = (int)w;
  *height = (int)h;
  return 0;
}

inline int png_write_rgba8(const std::string& path, int w, int h, const uint8_t* rgba) {
  if (w < 1 || h < 1 || !rgba) return -3;
  const size_t stride = (size_t)w * 4;
  std::vector<uint8_t> raw((stride + 1) * (size_t)h);
  for (int y = 0; y < h; y++) {
    raw[(stride + 1) * y] = 0; // filter type None
    memcpy(&raw[(stride + 1) * y + 1], rgba + stride * y, stride);
  }
  uLongf clen = compressBound((uLong)raw.size());
  std::vector<uint8_t> comp(clen);
  if (compress2(comp.data(), &clen, raw.data(), (uLong)raw.size(), 6) != Z_OK) return -4;
  std::vector<uint8_t> file = {0x89, 'P', 'N', 'G', 0x0d, 0x0a, 0x1a, 0x0a};
  auto chunk = [&](const char* type, const uint8_t* data, uint32_t len) {
    png_put32(file, len);
    const size_t start = file.size();
    file.insert(file.end(), type, type + 4);
    if (len) file.insert(file.end(), data, data + len);
    png_put32(file, (uint32_t)crc32(0L, &file[start], (uInt)(4 + len)));
  };
  uint8_t ihdr[13];
  ihdr[0] = (uint8_t)(w >> 24); ihdr[1] = (uint8_t)(w >> 16); ihdr[2] = (uint8_t)(w >> 8); ihdr[3] = (uint8_t)w;
  ihdr[4] = (uint8_t)(h >> 24); ihdr[5] = (uint8_t)(h >> 16); ihdr[6] = (uint8_t)(h >> 8); ihdr[7] = (uint8_t)h;
  ihdr[8] = 8; ihdr[9] = 6; ihdr[10] = 0; ihdr[11] = 0; ihdr[12] = 0;
  chunk("IHDR", ihdr, 13);
  chunk("IDAT", comp.data(), (uint32_t)clen);
  chunk("IEND", nullptr, 0);
  FILE* f = fopen(path.c_str(), "wb");
  if (!f) return -1;
  const bool ok = fwrite(file.data(), 1, file.size(), f) == file.size();
  fclose(f);
  return ok ? 0 : -1;
}

} // namespace prvhost
