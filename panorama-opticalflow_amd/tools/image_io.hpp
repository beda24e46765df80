// Minimal image file I/O for the drop-in CLI: what the reference gets from
// cv::imread(..., -1) / cv::imwrite (CPU/util.cpp:19-34) for its inputs (Hugin TIFFs) and outputs (PNG).
//   * TIFF reader: baseline little/big-endian, 8-bit RGB or RGBA, chunky, strips, compression none / LZW /
//     Deflate / PackBits, predictor 1 or 2.  Tiles, 16-bit, palette and planar files are rejected with a message.
//   * PNG reader/writer: 8-bit RGB/RGBA, non-interlaced (zlib).
// Pixels are delivered as BGRA (CV_8UC4) like cv::imread(-1) followed by the reference's BGR->BGRA (main.cpp:58,68).
#ifndef PANO_IMAGE_IO_HPP_
#define PANO_IMAGE_IO_HPP_

#include <zlib.h>

#include <cstdio>
#include <cstring>
#include <string>
#include <vector>

#include "../include/util.hpp"

namespace pano_io {
using panocv::Mat;
using util::VrCamException;

static inline std::vector<unsigned char> read_file(const std::string& path) {
  FILE* f = fopen(path.c_str(), "rb");
  if (!f) throw VrCamException("failed to load image: " + path);
  fseek(f, 0, SEEK_END); const long n = ftell(f); fseek(f, 0, SEEK_SET);
  std::vector<unsigned char> b(n > 0 ? n : 0);
  if (n > 0 && fread(b.data(), 1, n, f) != (size_t)n) { fclose(f); throw VrCamException("short read: " + path); }
  fclose(f);
  return b;
}

// ---------------------------------------------------------------- TIFF
struct TiffReader {
  const std::vector<unsigned char>& b; bool le; std::string path;
  unsigned u16(size_t o) const { chk(o, 2); return le ? b[o] | (b[o + 1] << 8) : (b[o] << 8) | b[o + 1]; }
  unsigned u32(size_t o) const { chk(o, 4); return le ? b[o] | (b[o + 1] << 8) | (b[o + 2] << 16) | ((unsigned)b[o + 3] << 24)
                                                      : ((unsigned)b[o] << 24) | (b[o + 1] << 16) | (b[o + 2] << 8) | b[o + 3]; }
  void chk(size_t o, size_t n) const { if (o + n > b.size()) throw VrCamException("truncated TIFF: " + path); }
  std::vector<unsigned> values(size_t entry) const {   // values of one IFD entry (SHORT or LONG)
    const unsigned type = u16(entry + 2), count = u32(entry + 4);
    const size_t sz = type == 3 ? 2 : (type == 4 ? 4 : (type == 1 ? 1 : 0));
    if (!sz) throw VrCamException("unsupported TIFF field type in " + path);
    size_t off = entry + 8;
    if (sz * count > 4) off = u32(entry + 8);
    std::vector<unsigned> v(count);
    for (unsigned i = 0; i < count; ++i) v[i] = sz == 2 ? u16(off + 2 * i) : (sz == 4 ? u32(off + 4 * i) : b[off + i]);
    return v;
  }
};

static inline void lzw_decode(const unsigned char* src, size_t n, std::vector<unsigned char>& out, size_t expect) {
  // TIFF LZW: MSB-first codes, 9..12 bits, ClearCode 256, EOI 257, "early change"
  std::vector<int> prefix(4096), length(4096); std::vector<unsigned char> suffix(4096), first(4096);
  for (int i = 0; i < 256; ++i) { prefix[i] = -1; suffix[i] = first[i] = (unsigned char)i; length[i] = 1; }
  int next = 258, bits = 9, old = -1; unsigned acc = 0; int nacc = 0; size_t pos = 0;
  std::vector<unsigned char> tmp(4096);
  // a corrupt stream must not walk off the table: every code emitted is < next, its chain has exactly length[code] links
  auto emit = [&](int code) -> bool {
    const int len = length[code]; const size_t base = out.size(); out.resize(base + len); int c = code;
    for (int i = len - 1; i >= 0; --i) { if (c < 0) return false; out[base + i] = suffix[c]; c = prefix[c]; }
    return true;
  };
  while (out.size() < expect) {
    while (nacc < bits && pos < n) { acc = (acc << 8) | src[pos++]; nacc += 8; }
    if (nacc < bits) break;
    const int code = (acc >> (nacc - bits)) & ((1 << bits) - 1); nacc -= bits;
    if (code == 257) break;
    if (code == 256) { next = 258; bits = 9; old = -1; continue; }
    if (old < 0) { if (code >= 256) break; emit(code); old = code; continue; }
    if (code > next || code >= 4096) break;   // not a code the encoder can have produced
    if (code < next) {
      if (!emit(code)) break;
      if (next < 4096) { prefix[next] = old; suffix[next] = first[code]; first[next] = first[old]; length[next] = length[old] + 1; ++next; }
    } else {
      if (next < 4096) { prefix[next] = old; suffix[next] = first[old]; first[next] = first[old]; length[next] = length[old] + 1; ++next; }
      else break;                     // table full: "code == next" cannot occur
      if (!emit(next - 1)) break;
    }
    old = code;
    if (next + 1 >= (1 << bits) && bits < 12) ++bits;   // early change
  }
}
static inline void packbits_decode(const unsigned char* s, size_t n, std::vector<unsigned char>& out, size_t expect) {
  size_t i = 0;
  while (i < n && out.size() < expect) {
    const int c = (signed char)s[i++];
    if (c >= 0) { for (int k = 0; k <= c && i < n; ++k) out.push_back(s[i++]); }
    else if (c != -128) { if (i >= n) break; out.insert(out.end(), size_t(1 - c), s[i++]); }
  }
}
static inline void inflate_all(const unsigned char* s, size_t n, std::vector<unsigned char>& out, size_t expect, const std::string& path) {
  const size_t base = out.size(); out.resize(base + expect);
  uLongf dl = (uLongf)expect;
  const int rc = uncompress(out.data() + base, &dl, s, (uLong)n);
  if (rc != Z_OK && rc != Z_BUF_ERROR) throw VrCamException("zlib error in " + path);
  out.resize(base + dl);
}

static inline Mat read_tiff(const std::string& path) {
  const std::vector<unsigned char> b = read_file(path);
  if (b.size() < 8 || !((b[0] == 'I' && b[1] == 'I') || (b[0] == 'M' && b[1] == 'M'))) throw VrCamException("not a TIFF: " + path);
  TiffReader t{b, b[0] == 'I', path};
  if (t.u16(2) != 42) throw VrCamException("not a TIFF: " + path);
  const size_t ifd = t.u32(4); const unsigned n = t.u16(ifd);
  unsigned w = 0, h = 0, spp = 1, comp = 1, photo = 2, rps = 0xffffffffu, planar = 1, pred = 1, bps = 8;
  std::vector<unsigned> so, sc;
  for (unsigned i = 0; i < n; ++i) {
    const size_t e = ifd + 2 + 12 * i; const unsigned tag = t.u16(e);
    switch (tag) {
      case 256: w = t.values(e)[0]; break;       case 257: h = t.values(e)[0]; break;
      case 258: bps = t.values(e)[0]; break;     case 259: comp = t.values(e)[0]; break;
      case 262: photo = t.values(e)[0]; break;   case 273: so = t.values(e); break;
      case 277: spp = t.values(e)[0]; break;     case 278: rps = t.values(e)[0]; break;
      case 279: sc = t.values(e); break;         case 284: planar = t.values(e)[0]; break;
      case 317: pred = t.values(e)[0]; break;
      case 322: case 324: throw VrCamException("tiled TIFF not supported: " + path);
      default: break;
    }
  }
  if (!w || !h || bps != 8 || planar != 1 || (spp != 3 && spp != 4) || photo != 2 || so.empty() || so.size() != sc.size())
    throw VrCamException("unsupported TIFF layout (need 8-bit chunky RGB/RGBA strips): " + path);
  if (rps > h) rps = h;
  if (rps == 0 || so.size() != (size_t(h) + rps - 1) / rps) throw VrCamException("inconsistent TIFF strip table (RowsPerStrip / StripOffsets): " + path);
  if ((double)w * h > 4.0e9) throw VrCamException("TIFF too large: " + path);
  std::vector<unsigned char> pix; pix.reserve(size_t(w) * h * spp);
  for (size_t s = 0; s < so.size(); ++s) {
    const unsigned rows = (unsigned)std::min<size_t>(rps, h - s * rps);
    const size_t expect = size_t(rows) * w * spp, before = pix.size();
    t.chk(so[s], sc[s]);
    const unsigned char* src = b.data() + so[s];
    if (comp == 1) pix.insert(pix.end(), src, src + std::min<size_t>(sc[s], expect));
    else if (comp == 5) lzw_decode(src, sc[s], pix, before + expect);
    else if (comp == 8 || comp == 32946) inflate_all(src, sc[s], pix, expect, path);
    else if (comp == 32773) packbits_decode(src, sc[s], pix, before + expect);
    else throw VrCamException("unsupported TIFF compression in " + path);
    pix.resize(before + expect);
    if (pred == 2)
      for (unsigned y = 0; y < rows; ++y) { unsigned char* r = pix.data() + before + size_t(y) * w * spp; for (size_t i = spp; i < size_t(w) * spp; ++i) r[i] = (unsigned char)(r[i] + r[i - spp]); }
  }
  if (pix.size() != size_t(w) * h * spp) throw VrCamException("TIFF strips do not cover the image: " + path);
  Mat m(h, w, panocv::CV_8UC4);
  for (unsigned y = 0; y < h; ++y)
    for (unsigned x = 0; x < w; ++x) {
      const unsigned char* p = pix.data() + (size_t(y) * w + x) * spp; unsigned char* d = m.data + size_t(y) * m.step + x * 4;
      d[0] = p[2]; d[1] = p[1]; d[2] = p[0]; d[3] = spp == 4 ? p[3] : 255;   // RGB(A) -> BGRA
    }
  return m;
}

// ---------------------------------------------------------------- PNG
static inline void be32(unsigned char* p, unsigned v) { p[0] = v >> 24; p[1] = v >> 16; p[2] = v >> 8; p[3] = v; }
static inline void png_chunk(FILE* f, const char* type, const unsigned char* data, size_t n) {
  unsigned char hd[8]; be32(hd, (unsigned)n); memcpy(hd + 4, type, 4);
  fwrite(hd, 1, 8, f); if (n) fwrite(data, 1, n, f);
  uLong c = crc32(0L, hd + 4, 4); if (n) c = crc32(c, data, (uInt)n);
  unsigned char cb[4]; be32(cb, (unsigned)c); fwrite(cb, 1, 4, f);
}
static inline void write_png(const std::string& path, const Mat& bgra) {
  if (bgra.type() != panocv::CV_8UC4) throw VrCamException("write_png expects CV_8UC4");
  const size_t w = bgra.cols, h = bgra.rows;
  std::vector<unsigned char> raw((w * 4 + 1) * h);
  for (size_t y = 0; y < h; ++y) {
    unsigned char* r = raw.data() + y * (w * 4 + 1); r[0] = 0;
    const unsigned char* s = bgra.data + y * bgra.step;
    for (size_t x = 0; x < w; ++x) { r[1 + 4 * x] = s[4 * x + 2]; r[2 + 4 * x] = s[4 * x + 1]; r[3 + 4 * x] = s[4 * x]; r[4 + 4 * x] = s[4 * x + 3]; }
  }
  uLongf zn = compressBound((uLong)raw.size()); std::vector<unsigned char> z(zn);
  if (compress2(z.data(), &zn, raw.data(), (uLong)raw.size(), 1) != Z_OK) throw VrCamException("zlib compress failed");
  FILE* f = fopen(path.c_str(), "wb");
  if (!f) throw VrCamException("failed to write image: " + path);
  static const unsigned char sig[8] = {137, 80, 78, 71, 13, 10, 26, 10}; fwrite(sig, 1, 8, f);
  unsigned char ih[13]; be32(ih, (unsigned)w); be32(ih + 4, (unsigned)h); ih[8] = 8; ih[9] = 6; ih[10] = 0; ih[11] = 0; ih[12] = 0;
  png_chunk(f, "IHDR", ih, 13); png_chunk(f, "IDAT", z.data(), zn); png_chunk(f, "IEND", nullptr, 0);
  fclose(f);
}
static inline Mat read_png(const std::string& path) {
  const std::vector<unsigned char> b = read_file(path);
  static const unsigned char sig[8] = {137, 80, 78, 71, 13, 10, 26, 10};
  if (b.size() < 33 || memcmp(b.data(), sig, 8)) throw VrCamException("not a PNG: " + path);
  auto rd = [&](size_t o) { return ((unsigned)b[o] << 24) | (b[o + 1] << 16) | (b[o + 2] << 8) | b[o + 3]; };
  unsigned w = 0, h = 0, ct = 0; std::vector<unsigned char> z;
  for (size_t o = 8; o + 12 <= b.size();) {
    const unsigned n = rd(o); const char* ty = (const char*)&b[o + 4];
    if (o + 12 + n > b.size()) break;
    if (!memcmp(ty, "IHDR", 4)) { if (n != 13) throw VrCamException("bad PNG header: " + path); w = rd(o + 8); h = rd(o + 12); ct = b[o + 17]; if (b[o + 16] != 8 || b[o + 20] != 0 || (ct != 2 && ct != 6)) throw VrCamException("unsupported PNG (need 8-bit RGB/RGBA, non-interlaced): " + path); }
    else if (!memcmp(ty, "IDAT", 4)) z.insert(z.end(), b.begin() + o + 8, b.begin() + o + 8 + n);
    o += 12 + n;
  }
  const unsigned bpp = ct == 6 ? 4 : 3; const size_t stride = size_t(w) * bpp;
  std::vector<unsigned char> raw((stride + 1) * h); uLongf rn = (uLongf)raw.size();
  if (!w || !h || uncompress(raw.data(), &rn, z.data(), (uLong)z.size()) != Z_OK) throw VrCamException("bad PNG data: " + path);
  Mat m(h, w, panocv::CV_8UC4);
  std::vector<unsigned char> prev(stride, 0), cur(stride);
  for (unsigned y = 0; y < h; ++y) {
    const unsigned char* r = raw.data() + size_t(y) * (stride + 1); const int ft = r[0];
    for (size_t i = 0; i < stride; ++i) {
      const int a = i >= bpp ? cur[i - bpp] : 0, bb = prev[i], c = i >= bpp ? prev[i - bpp] : 0; int v = r[1 + i];
      if (ft == 1) v += a; else if (ft == 2) v += bb; else if (ft == 3) v += (a + bb) >> 1;
      else if (ft == 4) { const int p = a + bb - c, pa = abs(p - a), pb = abs(p - bb), pc = abs(p - c); v += (pa <= pb && pa <= pc) ? a : (pb <= pc ? bb : c); }
      cur[i] = (unsigned char)v;
    }
    unsigned char* d = m.data + size_t(y) * m.step;
    for (unsigned x = 0; x < w; ++x) { d[4 * x] = cur[x * bpp + 2]; d[4 * x + 1] = cur[x * bpp + 1]; d[4 * x + 2] = cur[x * bpp]; d[4 * x + 3] = bpp == 4 ? cur[x * bpp + 3] : 255; }
    prev.swap(cur);
  }
  return m;
}

// imreadExceptionOnFail / imwriteExceptionOnFail (CPU/util.cpp:19-34), by extension
static inline Mat imreadExceptionOnFail(const std::string& path) {
  const size_t dot = path.rfind('.'); std::string ext = dot == std::string::npos ? "" : path.substr(dot + 1);
  for (auto& c : ext) c = (char)tolower(c);
  if (ext == "tif" || ext == "tiff") return read_tiff(path);
  if (ext == "png") return read_png(path);
  throw VrCamException("failed to load image (unsupported extension): " + path);
}
static inline void imwriteExceptionOnFail(const std::string& path, const Mat& image) { write_png(path, image); }

}  // namespace pano_io
#endif
