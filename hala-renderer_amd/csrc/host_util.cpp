// host_util.cpp — error channel, JSON reader, and the host half of save_images / set_envmap:
// tonemap operators and PFM writer (src/rt_renderer.rs:1256-1334), Radiance .hdr / .pfm decoding for
// EnvMap::new_with_file (src/envmap.rs:48-60; the reference decodes through the `image` crate).
#include <zlib.h>

#include "host_util.h"

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <exception>

#include "hala_types.h"
#include "host_image.h"

namespace rt {

static thread_local std::string g_last_error;
void set_last_error(const std::string& msg) { g_last_error = msg; }
const char* get_last_error() { return g_last_error.c_str(); }

// ---- JSON ---------------------------------------------------------------------------------------------------
namespace {
struct JsonParser {
  const char* s;
  size_t i = 0;
  std::string err;
  void ws() { while (s[i] == ' ' || s[i] == '\t' || s[i] == '\n' || s[i] == '\r') ++i; }
  bool fail(const std::string& m) { if (err.empty()) err = m + " at byte " + std::to_string(i); return false; }
  bool parse_string(std::string* out) {
    if (s[i] != '"') return fail("expected string");
    ++i;
    out->clear();
    while (s[i] && s[i] != '"') {
      if (s[i] == '\\') {
        ++i;
        switch (s[i]) {
          case '"': out->push_back('"'); break;
          case '\\': out->push_back('\\'); break;
          case '/': out->push_back('/'); break;
          case 'b': out->push_back('\b'); break;
          case 'f': out->push_back('\f'); break;
          case 'n': out->push_back('\n'); break;
          case 'r': out->push_back('\r'); break;
          case 't': out->push_back('\t'); break;
          case 'u': {
            unsigned cp = 0;
            for (int k = 1; k <= 4; ++k) {
              char c = s[i + k];
              cp <<= 4;
              if (c >= '0' && c <= '9') cp |= c - '0';
              else if (c >= 'a' && c <= 'f') cp |= c - 'a' + 10;
              else if (c >= 'A' && c <= 'F') cp |= c - 'A' + 10;
              else return fail("bad \\u escape");
            }
            i += 4;
            if (cp < 0x80) out->push_back((char)cp);
            else if (cp < 0x800) { out->push_back((char)(0xC0 | (cp >> 6))); out->push_back((char)(0x80 | (cp & 0x3F))); }
            else { out->push_back((char)(0xE0 | (cp >> 12))); out->push_back((char)(0x80 | ((cp >> 6) & 0x3F))); out->push_back((char)(0x80 | (cp & 0x3F))); }
            break;
          }
          default: return fail("bad escape");
        }
        ++i;
      } else out->push_back(s[i++]);
    }
    if (s[i] != '"') return fail("unterminated string");
    ++i;
    return true;
  }
  bool parse_value(JsonValue* v, int depth) {
    if (depth > 64) return fail("nesting too deep");
    ws();
    const char c = s[i];
    if (c == '{') {
      v->kind = JsonValue::Object;
      ++i; ws();
      if (s[i] == '}') { ++i; return true; }
      for (;;) {
        ws();
        std::string key;
        if (!parse_string(&key)) return false;
        ws();
        if (s[i] != ':') return fail("expected ':'");
        ++i;
        JsonValue child;
        if (!parse_value(&child, depth + 1)) return false;
        v->members.emplace_back(std::move(key), std::move(child));
        ws();
        if (s[i] == ',') { ++i; continue; }
        if (s[i] == '}') { ++i; return true; }
        return fail("expected ',' or '}'");
      }
    }
    if (c == '[') {
      v->kind = JsonValue::Array;
      ++i; ws();
      if (s[i] == ']') { ++i; return true; }
      for (;;) {
        JsonValue child;
        if (!parse_value(&child, depth + 1)) return false;
        v->items.push_back(std::move(child));
        ws();
        if (s[i] == ',') { ++i; continue; }
        if (s[i] == ']') { ++i; return true; }
        return fail("expected ',' or ']'");
      }
    }
    if (c == '"') { v->kind = JsonValue::String; return parse_string(&v->str); }
    if (!strncmp(s + i, "true", 4)) { v->kind = JsonValue::Bool; v->b = true; i += 4; return true; }
    if (!strncmp(s + i, "false", 5)) { v->kind = JsonValue::Bool; v->b = false; i += 5; return true; }
    if (!strncmp(s + i, "null", 4)) { v->kind = JsonValue::Null; i += 4; return true; }
    if (c == '-' || (c >= '0' && c <= '9')) {
      char* end = nullptr;
      v->num = strtod(s + i, &end);
      if (end == s + i) return fail("bad number");
      v->kind = JsonValue::Number;
      i = (size_t)(end - s);
      return true;
    }
    return fail("unexpected character");
  }
};
}  // namespace

std::string json_parse(const char* text, JsonValue* out) {
  JsonParser p{text};
  if (!p.parse_value(out, 0)) return p.err;
  p.ws();
  if (text[p.i] != 0) return "trailing characters at byte " + std::to_string(p.i);
  return "";
}

// ---- tonemap: src/rt_renderer.rs:1256-1316 (host side, applied by save_images to the accum read-back) --------------
namespace {
struct C3 { float x, y, z; };
inline float lum709(C3 c) { return 0.212671f * c.x + 0.715160f * c.y + 0.072169f * c.z; }  // :1257-1259
inline float clamp01f(float v) { return v < 0.0f ? 0.0f : (v > 1.0f ? 1.0f : v); }
inline C3 clamp01(C3 c) { return C3{clamp01f(c.x), clamp01f(c.y), clamp01f(c.z)}; }
inline C3 mat3(const float c0[3], const float c1[3], const float c2[3], C3 v) {  // glam Mat3 * Vec3
  return C3{c0[0] * v.x + c1[0] * v.y + c2[0] * v.z, c0[1] * v.x + c1[1] * v.y + c2[1] * v.z, c0[2] * v.x + c1[2] * v.y + c2[2] * v.z};
}
inline float fit1(float v) {  // rrt_odt_fit :1260-1264, per channel
  const float a = v * (v + 0.0245786f) - 0.000090537f;
  const float b = v * (0.983729f * v + 0.432951f) + 0.238081f;
  return a / b;
}
inline C3 aces_fitted(C3 c) {  // :1265-1281
  static const float i0[3] = {0.59719f, 0.07600f, 0.02840f}, i1[3] = {0.35458f, 0.90834f, 0.13383f}, i2[3] = {0.04823f, 0.01566f, 0.83777f};
  static const float o0[3] = {1.60475f, -0.10208f, -0.00327f}, o1[3] = {-0.53108f, 1.10813f, -0.07276f}, o2[3] = {-0.07367f, -0.00605f, 1.07602f};
  c = mat3(i0, i1, i2, c);
  c = C3{fit1(c.x), fit1(c.y), fit1(c.z)};
  c = mat3(o0, o1, o2, c);
  return clamp01(c);
}
inline float aces1(float c) {  // :1282-1291, per channel
  return (c * (2.51f * c + 0.03f)) / (c * (2.43f * c + 0.59f) + 0.14f);
}
}  // namespace

void tonemap_pixels(float* rgba, size_t count, int enable_tonemap, int enable_aces, int use_simple_aces) {
  if (!enable_tonemap) return;  // :1299-1311
  for (size_t i = 0; i < count; ++i) {
    C3 c{rgba[4 * i], rgba[4 * i + 1], rgba[4 * i + 2]};
    if (enable_aces) {
      if (use_simple_aces) c = clamp01(C3{aces1(c.x), aces1(c.y), aces1(c.z)});
      else c = aces_fitted(c);
    } else {
      const float d = 1.0f + lum709(c) / 1.5f;  // tonemap(c, 1.5) = c * 1.0 / (1.0 + luminance(c) / limit)  :1292-1294
      c = C3{(c.x * 1.0f) / d, (c.y * 1.0f) / d, (c.z * 1.0f) / d};
    }
    rgba[4 * i] = c.x; rgba[4 * i + 1] = c.y; rgba[4 * i + 2] = c.z;
  }
}

// ---- PFM writer: src/rt_renderer.rs:1318-1334 ------------------------------------------------------------------------
std::string write_pfm(const char* path, const float* rgba, uint32_t width, uint32_t height) {
  FILE* f = fopen(path, "wb");
  if (!f) return std::string("Failed to create the image file: \"") + path + "\"";
  fprintf(f, "PF\n%u %u\n-1.0\n", width, height);  // writeln!("PF\n{} {}\n-1.0")
  std::vector<float> row((size_t)width * 3);
  for (uint32_t y = height; y-- > 0;) {  // rows bottom-to-top (.rev())
    for (uint32_t x = 0; x < width; ++x) memcpy(&row[3 * (size_t)x], rgba + 4 * ((size_t)y * width + x), 12);  // little-endian host
    if (fwrite(row.data(), 4, row.size(), f) != row.size()) { fclose(f); return std::string("Failed to write the image file: \"") + path + "\""; }
  }
  if (fclose(f) != 0) return std::string("Failed to flush the image file: \"") + path + "\"";
  return "";
}

// ---- decoders for set_envmap_file ---------------------------------------------------------------------------------------
static bool read_line(FILE* f, std::string* line) {
  line->clear();
  int c;
  while ((c = fgetc(f)) != EOF) {
    if (c == '\n') return true;
    line->push_back((char)c);
    if (line->size() > 4096) return false;
  }
  return !line->empty();
}

static std::string load_pfm(FILE* f, HostImage* img) {
  std::string l1, l2, l3;
  if (!read_line(f, &l1) || !read_line(f, &l2) || !read_line(f, &l3)) return "truncated PFM header";
  const int ch = l1 == "PF" ? 3 : (l1 == "Pf" ? 1 : 0);
  if (!ch) return "not a PFM file";
  unsigned w = 0, h = 0;
  if (sscanf(l2.c_str(), "%u %u", &w, &h) != 2 || !w || !h) return "bad PFM dimensions";
  const double scale = atof(l3.c_str());
  if (scale >= 0.0) return "big-endian PFM is not supported";
  img->width = w; img->height = h; img->channels = 3;
  img->pixels.assign((size_t)w * h * 3, 0.0f);
  std::vector<float> row((size_t)w * ch);
  for (unsigned y = h; y-- > 0;) {  // PFM stores rows bottom-to-top
    if (fread(row.data(), 4, row.size(), f) != row.size()) return "truncated PFM data";
    for (unsigned x = 0; x < w; ++x)
      for (int c = 0; c < 3; ++c) img->pixels[((size_t)y * w + x) * 3 + c] = row[(size_t)x * ch + (ch == 3 ? c : 0)];
  }
  return "";
}

static std::string load_hdr(FILE* f, HostImage* img) {
  std::string line;
  bool fmt_ok = false;
  if (!read_line(f, &line) || (line.rfind("#?", 0) != 0)) return "not a Radiance HDR file";
  for (;;) {
    if (!read_line(f, &line)) { if (feof(f)) return "truncated HDR header"; }
    if (line.empty()) break;
    if (line == "FORMAT=32-bit_rle_rgbe") fmt_ok = true;
  }
  if (!fmt_ok) return "unsupported HDR pixel format";
  if (!read_line(f, &line)) return "missing HDR resolution line";
  int w = 0, h = 0;
  if (sscanf(line.c_str(), "-Y %d +X %d", &h, &w) != 2 || w <= 0 || h <= 0) return "unsupported HDR orientation";
  img->width = (uint32_t)w; img->height = (uint32_t)h; img->channels = 3;
  img->pixels.assign((size_t)w * h * 3, 0.0f);
  std::vector<unsigned char> scan((size_t)w * 4);
  for (int y = 0; y < h; ++y) {
    unsigned char hd[4];
    if (fread(hd, 1, 4, f) != 4) return "truncated HDR data";
    if (hd[0] == 2 && hd[1] == 2 && !(hd[2] & 0x80) && ((hd[2] << 8) | hd[3]) == w && w >= 8 && w < 32768) {
      for (int c = 0; c < 4; ++c) {  // new-style RLE, channel-planar per scanline
        int x = 0;
        while (x < w) {
          int cnt = fgetc(f);
          if (cnt == EOF) return "truncated HDR data";
          if (cnt > 128) {
            cnt -= 128;
            const int val = fgetc(f);
            if (val == EOF || x + cnt > w) return "corrupt HDR run";
            while (cnt--) scan[(size_t)(x++) * 4 + c] = (unsigned char)val;
          } else {
            if (cnt == 0 || x + cnt > w) return "corrupt HDR run";
            while (cnt--) { const int val = fgetc(f); if (val == EOF) return "truncated HDR data"; scan[(size_t)(x++) * 4 + c] = (unsigned char)val; }
          }
        }
      }
    } else {  // flat scanline
      memcpy(scan.data(), hd, 4);
      if (w > 1 && fread(scan.data() + 4, 4, (size_t)w - 1, f) != (size_t)w - 1) return "truncated HDR data";
    }
    for (int x = 0; x < w; ++x) {
      const unsigned char* p = &scan[(size_t)x * 4];
      float* o = &img->pixels[((size_t)y * w + x) * 3];
      if (p[3] == 0) { o[0] = o[1] = o[2] = 0.0f; }
      else {
        const float sc = std::ldexp(1.0f, (int)p[3] - (128 + 8));  // image-rs hdr decoder: mantissa * 2^(e-136)
        o[0] = (float)p[0] * sc; o[1] = (float)p[1] * sc; o[2] = (float)p[2] * sc;
      }
    }
  }
  return "";
}

// ---- OpenEXR, scanline and single-level tiled images (the `image` crate's "exr" feature, Cargo.toml:21): compression NONE / RLE /
// ZIPS / ZIP / PIZ, HALF / FLOAT / UINT samples, channels R G B [A] or Y; mip / rip-mapped tiles, deep, multi-part and PXR24 / B44 / DWA
// files are refused ----
static float half_to_float(uint16_t h) {
  const uint32_t sign = (uint32_t)(h & 0x8000u) << 16, exp = (h >> 10) & 0x1fu, man = h & 0x3ffu;
  uint32_t bits;
  if (exp == 0) {
    if (man == 0) bits = sign;
    else {  // subnormal: normalise
      int e = -1; uint32_t m = man;
      do { ++e; m <<= 1; } while (!(m & 0x400u));
      bits = sign | ((uint32_t)(127 - 15 - e) << 23) | ((m & 0x3ffu) << 13);
    }
  } else if (exp == 31) bits = sign | 0x7f800000u | (man << 13);
  else bits = sign | ((exp + 127 - 15) << 23) | (man << 13);
  float f; memcpy(&f, &bits, 4);
  return f;
}
// ---- PIZ (OpenEXR compression 4): 16-bit samples -> range compaction through a bitmap of the values in use -> 2D Haar-like wavelet per
// channel (14-bit variant when the compacted range allows it, else the modulo-2^16 variant) -> canonical Huffman code over the 65 536 + 1
// symbols (the extra one: "repeat the previous symbol n times").  Restated from the published format (OpenEXR's ImfPizCompressor / ImfHuf /
// ImfWav, v2 file layout); the reference reads such files through the `image` crate's exr support.  No PIZ file written by another
// implementation was available to check against (tests round-trip through an encoder written from the same description): parity unpinned.
namespace {
struct PizBits {  // most significant bit first
  const unsigned char* p; const unsigned char* end;
  uint64_t c = 0; int lc = 0;
  bool get(int n, uint32_t* out) {
    while (lc < n) { if (p >= end) return false; c = (c << 8) | *p++; lc += 8; }
    lc -= n;
    *out = (uint32_t)((c >> lc) & ((1ull << n) - 1ull));
    return true;
  }
};
constexpr uint32_t kHufSymbols = 65537u;  // 16-bit values + the run symbol
constexpr int kHufMaxLen = 58;
// the Huffman part of a PIZ block: [im, iM, table bytes (unused), nBits, reserved] (five little-endian 32-bit words), the packed code lengths of
// symbols im..iM (6 bits each; 59..62 = a run of 2..5 zero lengths, 63 + 8 bits = a run of 6..261), then nBits of code
const char* piz_huf_decode(const unsigned char* in, size_t n_in, uint16_t* out, size_t n_out) {
  if (n_in == 0) return n_out == 0 ? nullptr : "not enough Huffman data";
  if (n_in < 20) return "truncated Huffman header";
  auto rd = [&](size_t at) { uint32_t v; memcpy(&v, in + at, 4); return v; };
  const uint32_t im = rd(0), iM = rd(4), n_bits = rd(12);
  if (im >= kHufSymbols || iM >= kHufSymbols || im > iM) return "bad Huffman symbol range";
  std::vector<unsigned char> len(kHufSymbols, 0);
  PizBits tb{in + 20, in + n_in};
  for (uint32_t k = im; k <= iM; ++k) {
    uint32_t l;
    if (!tb.get(6, &l)) return "truncated Huffman table";
    if (l == 63u) {
      uint32_t z;
      if (!tb.get(8, &z)) return "truncated Huffman table";
      z += 6u;
      if (k + z > iM + 1u) return "bad zero run in the Huffman table";
      k += z - 1u;
    } else if (l >= 59u) {
      const uint32_t z = l - 59u + 2u;
      if (k + z > iM + 1u) return "bad zero run in the Huffman table";
      k += z - 1u;
    } else len[k] = (unsigned char)l;
  }
  // canonical codes: the codes of one length are consecutive numbers in symbol order; the first code of length l follows from the counts of
  // the longer lengths
  uint64_t count[kHufMaxLen + 1] = {0}, first[kHufMaxLen + 1] = {0};
  for (uint32_t k = im; k <= iM; ++k) count[len[k]]++;
  { uint64_t c = 0; for (int l = kHufMaxLen; l > 0; --l) { const uint64_t nc = (c + count[l]) >> 1; first[l] = c; c = nc; } }
  std::vector<uint32_t> start(kHufMaxLen + 2, 0), sym;  // symbols sorted by (length, index)
  for (int l = 1; l <= kHufMaxLen; ++l) start[l + 1] = start[l] + (uint32_t)count[l];
  sym.resize(start[kHufMaxLen + 1]);
  { std::vector<uint32_t> fill(start.begin(), start.end()); for (uint32_t k = im; k <= iM; ++k) if (len[k]) sym[fill[len[k]]++] = k; }
  const unsigned char* data = tb.p;  // the table ends on a byte boundary of its own reader
  const size_t avail = (size_t)(in + n_in - data);
  if ((uint64_t)n_bits > (uint64_t)avail * 8u) return "bad Huffman bit count";
  PizBits db{data, data + (n_bits + 7u) / 8u};
  uint64_t left = n_bits;
  size_t o = 0;
  const uint32_t run_symbol = iM;
  while (o < n_out) {
    uint64_t v = 0;
    int l = 0;
    uint32_t s = kHufSymbols;
    while (l < kHufMaxLen) {
      uint32_t bit;
      if (left == 0 || !db.get(1, &bit)) return "not enough Huffman data";
      --left;
      v = (v << 1) | bit; ++l;
      if (count[l] && v >= first[l] && v - first[l] < count[l]) { s = sym[start[l] + (uint32_t)(v - first[l])]; break; }
    }
    if (s == kHufSymbols) return "bad Huffman code";
    if (s == run_symbol) {
      uint32_t n;
      if (left < 8 || !db.get(8, &n)) return "not enough Huffman data";
      left -= 8;
      if (o == 0 || o + n > n_out) return "bad run in the Huffman data";
      const uint16_t prev = out[o - 1];
      for (uint32_t k = 0; k < n; ++k) out[o++] = prev;
    } else out[o++] = (uint16_t)s;
  }
  return nullptr;
}
// inverse of the two-point transforms: 14-bit data (average / difference in signed 16-bit arithmetic), 16-bit data (modulo 2^16)
inline void piz_wdec14(uint16_t l, uint16_t h, uint16_t* a, uint16_t* b) {
  const int ls = (int16_t)l, hs = (int16_t)h;
  const int ai = ls + (hs & 1) + (hs >> 1);
  *a = (uint16_t)(int16_t)ai; *b = (uint16_t)(int16_t)(ai - hs);
}
inline void piz_wdec16(uint16_t l, uint16_t h, uint16_t* a, uint16_t* b) {
  const int m = l, d = h;
  const int bb = (m - (d >> 1)) & 0xffff;
  const int aa = (d + bb - 0x8000) & 0xffff;
  *b = (uint16_t)bb; *a = (uint16_t)aa;
}
void piz_wav2_decode(uint16_t* in, int nx, int ox, int ny, int oy, uint16_t mx) {
  const bool w14 = mx < (1u << 14);
  const int n = nx > ny ? ny : nx;
  int p = 1, p2;
  while (p <= n) p <<= 1;
  p >>= 1; p2 = p; p >>= 1;
  while (p >= 1) {
    uint16_t* py = in;
    uint16_t* const ey = in + (ptrdiff_t)oy * (ny - p2);
    const ptrdiff_t oy1 = (ptrdiff_t)oy * p, oy2 = (ptrdiff_t)oy * p2, ox1 = (ptrdiff_t)ox * p, ox2 = (ptrdiff_t)ox * p2;
    uint16_t i00, i01, i10, i11;
    for (; py <= ey; py += oy2) {
      uint16_t* px = py;
      uint16_t* const ex = py + (ptrdiff_t)ox * (nx - p2);
      for (; px <= ex; px += ox2) {
        uint16_t* const p01 = px + ox1; uint16_t* const p10 = px + oy1; uint16_t* const p11 = p10 + ox1;
        if (w14) { piz_wdec14(*px, *p10, &i00, &i10); piz_wdec14(*p01, *p11, &i01, &i11); piz_wdec14(i00, i01, px, p01); piz_wdec14(i10, i11, p10, p11); }
        else { piz_wdec16(*px, *p10, &i00, &i10); piz_wdec16(*p01, *p11, &i01, &i11); piz_wdec16(i00, i01, px, p01); piz_wdec16(i10, i11, p10, p11); }
      }
      if (nx & p) {  // a last column without a right neighbour
        uint16_t* const p10 = px + oy1;
        if (w14) piz_wdec14(*px, *p10, &i00, p10); else piz_wdec16(*px, *p10, &i00, p10);
        *px = i00;
      }
    }
    if (ny & p) {  // a last row without a lower neighbour
      uint16_t* px = py;
      uint16_t* const ex = py + (ptrdiff_t)ox * (nx - p2);
      for (; px <= ex; px += ox2) {
        uint16_t* const p01 = px + ox1;
        if (w14) piz_wdec14(*px, *p01, &i00, p01); else piz_wdec16(*px, *p01, &i00, p01);
        *px = i00;
      }
    }
    p2 = p; p >>= 1;
  }
}
// one PIZ block -> the bytes of its lines in the uncompressed layout (line after line, channel after channel within a line).
// types[k]: 0 UINT, 1 HALF, 2 FLOAT (32-bit samples travel as two 16-bit halves side by side)
const char* piz_decode(const unsigned char* src, size_t size, uint32_t cols, uint32_t lines, const std::vector<int>& types, std::vector<unsigned char>* raw) {
  struct Chan { size_t start; uint32_t halves; };
  std::vector<Chan> ch(types.size());
  size_t total = 0;
  for (size_t k = 0; k < types.size(); ++k) { ch[k].start = total; ch[k].halves = types[k] == 1 ? 1u : 2u; total += (size_t)cols * lines * ch[k].halves; }
  if (size < 4) return "truncated PIZ block";
  uint16_t min_nz, max_nz;
  memcpy(&min_nz, src, 2); memcpy(&max_nz, src + 2, 2);
  size_t p = 4;
  std::vector<unsigned char> bitmap(8192, 0);
  if (max_nz >= 8192) return "bad PIZ bitmap range";
  if (min_nz <= max_nz) {
    const size_t n = (size_t)max_nz - min_nz + 1;
    if (p + n > size) return "truncated PIZ bitmap";
    memcpy(&bitmap[min_nz], src + p, n);
    p += n;
  }
  std::vector<uint16_t> lut(65536, 0);
  uint32_t k = 0;
  for (uint32_t i = 0; i < 65536u; ++i)
    if (i == 0 || (bitmap[i >> 3] & (1u << (i & 7u)))) lut[k++] = (uint16_t)i;
  const uint16_t max_value = (uint16_t)(k - 1u);
  if (p + 4 > size) return "truncated PIZ block";
  uint32_t huf_len;
  memcpy(&huf_len, src + p, 4);
  p += 4;
  if ((size_t)huf_len > size - p) return "bad PIZ Huffman length";
  std::vector<uint16_t> tmp(total);
  if (const char* e = piz_huf_decode(src + p, huf_len, tmp.data(), total)) return e;
  for (size_t c = 0; c < ch.size(); ++c)
    for (uint32_t j = 0; j < ch[c].halves; ++j)
      piz_wav2_decode(tmp.data() + ch[c].start + j, (int)cols, (int)ch[c].halves, (int)lines, (int)(cols * ch[c].halves), max_value);
  for (uint16_t& v : tmp) v = lut[v];
  raw->resize(total * 2);
  size_t o = 0;
  std::vector<size_t> at(ch.size());
  for (size_t c = 0; c < ch.size(); ++c) at[c] = ch[c].start;
  for (uint32_t y = 0; y < lines; ++y)
    for (size_t c = 0; c < ch.size(); ++c) {
      const size_t n = (size_t)cols * ch[c].halves;
      memcpy(raw->data() + o, tmp.data() + at[c], n * 2);
      at[c] += n; o += n * 2;
    }
  return nullptr;
}
}  // namespace

static std::string load_exr(FILE* f, HostImage* img) {
  std::vector<unsigned char> d;
  { unsigned char buf[65536]; size_t n; while ((n = fread(buf, 1, sizeof(buf), f)) > 0) d.insert(d.end(), buf, buf + n); }
  size_t p = 0;
  auto need = [&](size_t n) { return p + n <= d.size(); };
  auto rd32 = [&](size_t at) { uint32_t v; memcpy(&v, &d[at], 4); return v; };
  if (!need(8) || rd32(0) != 20000630u) return "not an OpenEXR file";
  const uint32_t version = rd32(4);
  if ((version & 0xffu) != 2u || (version & 0x1800u)) return "deep and multi-part OpenEXR files are not supported";
  const bool tiled = (version & 0x200u) != 0u;
  uint32_t tile_w = 0, tile_h = 0, tile_mode = 0;
  p = 8;
  struct Chan { std::string name; int type; };
  std::vector<Chan> chans;
  int compression = -1, line_order = 0;
  int32_t win[4] = {0, 0, -1, -1};
  for (;;) {  // attributes: name\0 type\0 size value
    if (!need(1)) return "truncated header";
    if (d[p] == 0) { ++p; break; }
    std::string name, type;
    while (need(1) && d[p]) name += (char)d[p++];
    ++p;
    while (need(1) && d[p]) type += (char)d[p++];
    ++p;
    if (!need(4)) return "truncated header";
    const uint32_t size = rd32(p); p += 4;
    if (!need(size)) return "truncated header";
    if (name == "channels") {
      size_t q = p;
      while (q < p + size && d[q]) {
        Chan c;
        while (q < p + size && d[q]) c.name += (char)d[q++];
        ++q;
        if (q + 16 > p + size) return "bad channel list";
        c.type = (int)rd32(q);
        if (rd32(q + 8) != 1u || rd32(q + 12) != 1u) return "subsampled channels are not supported";
        q += 16;
        chans.push_back(c);
      }
    } else if (name == "compression" && size >= 1) compression = d[p];
    else if (name == "dataWindow" && size >= 16) memcpy(win, &d[p], 16);
    else if (name == "lineOrder" && size >= 1) line_order = d[p];
    else if (name == "tiles" && size >= 9) { tile_w = rd32(p); tile_h = rd32(p + 4); tile_mode = d[p + 8]; }
    p += size;
  }
  if (chans.empty() || win[2] < win[0] || win[3] < win[1]) return "missing channels or data window";
  if (compression < 0 || compression > 4) return "only NONE / RLE / ZIPS / ZIP / PIZ compressed OpenEXR files are supported";
  (void)line_order;  // chunks carry their own y; the offset table is not needed
  if ((int64_t)win[2] - win[0] >= (1 << 20) || (int64_t)win[3] - win[1] >= (1 << 20)) return "data window too large";
  const uint32_t W = (uint32_t)(win[2] - win[0] + 1), H = (uint32_t)(win[3] - win[1] + 1);
  if ((uint64_t)W * H > (1ull << 30)) return "data window too large";
  int ci[4] = {-1, -1, -1, -1};  // R G B A
  size_t line_bytes = 0;
  std::vector<size_t> chan_off(chans.size());
  for (size_t k = 0; k < chans.size(); ++k) {
    chan_off[k] = line_bytes;
    if (chans[k].type < 0 || chans[k].type > 2) return "bad pixel type";
    line_bytes += (size_t)W * (chans[k].type == 1 ? 2 : 4);
    if (chans[k].name == "R") ci[0] = (int)k; else if (chans[k].name == "G") ci[1] = (int)k;
    else if (chans[k].name == "B") ci[2] = (int)k; else if (chans[k].name == "A") ci[3] = (int)k;
    else if (chans[k].name == "Y" && ci[0] < 0) ci[0] = ci[1] = ci[2] = (int)k;
  }
  if (ci[0] < 0 || ci[1] < 0 || ci[2] < 0) return "no R, G, B (or Y) channels";
  if (tiled && (tile_w == 0 || tile_h == 0 || tile_w > (1u << 16) || tile_h > (1u << 16))) return "bad tile description";
  if (tiled && (tile_mode & 0x0fu) != 0u) return "mip-mapped and rip-mapped tiled OpenEXR files are not supported";
  const uint32_t block = compression == 3 ? 16u : (compression == 4 ? 32u : 1u);
  const uint32_t tiles_x = tiled ? (W + tile_w - 1) / tile_w : 1u, tiles_y = tiled ? (H + tile_h - 1) / tile_h : 0u;
  const uint32_t chunks = tiled ? tiles_x * tiles_y : (H + block - 1) / block;
  if ((size_t)chunks * 8 > d.size()) return "truncated offset table";
  p += (size_t)chunks * 8;  // offset table (chunks carry their own coordinates)
  img->width = W; img->height = H; img->channels = ci[3] >= 0 ? 4 : 3;
  img->pixels.assign((size_t)W * H * img->channels, 0.0f);
  std::vector<unsigned char> raw, tmp;
  for (uint32_t c = 0; c < chunks; ++c) {
    // a scanline chunk: y, size, data of `block` lines; a tile chunk: tile x, tile y, level x, level y, size, data of the tile's lines
    if (!need(tiled ? 20 : 8)) return "truncated pixel data";
    int32_t y0; uint32_t x_first = 0, cols = W;
    uint32_t lines;
    if (tiled) {
      int32_t tx, ty, lx, ly;
      memcpy(&tx, &d[p], 4); memcpy(&ty, &d[p + 4], 4); memcpy(&lx, &d[p + 8], 4); memcpy(&ly, &d[p + 12], 4);
      p += 16;
      if (tx < 0 || ty < 0 || (uint32_t)tx >= tiles_x || (uint32_t)ty >= tiles_y || lx != 0 || ly != 0) return "bad tile coordinates";
      x_first = (uint32_t)tx * tile_w; cols = std::min(tile_w, W - x_first);
      y0 = win[1] + (int32_t)((uint32_t)ty * tile_h);
      lines = std::min(tile_h, H - (uint32_t)ty * tile_h);
    } else {
      memcpy(&y0, &d[p], 4);
      p += 4;
      if (y0 < win[1] || y0 > win[3]) return "bad chunk";
      lines = std::min<uint32_t>(block, (uint32_t)(win[3] - y0 + 1));
    }
    const uint32_t size = rd32(p);
    p += 4;
    if (!need(size)) return "bad chunk";
    // bytes of one line of this chunk: every channel's samples of the chunk's columns, channel after channel
    size_t chunk_line = 0;
    std::vector<size_t> coff(chans.size());
    for (size_t k = 0; k < chans.size(); ++k) { coff[k] = chunk_line; chunk_line += (size_t)cols * (chans[k].type == 1 ? 2 : 4); }
    const size_t want = chunk_line * lines;
    if (compression == 0 || size == want) raw.assign(d.begin() + p, d.begin() + p + size);
    else if (compression == 4) {
      std::vector<int> types(chans.size());
      for (size_t k = 0; k < chans.size(); ++k) types[k] = chans[k].type;
      if (const char* e = piz_decode(&d[p], size, cols, lines, types, &raw)) return e;
    } else {
      tmp.resize(want);
      if (compression == 1) {  // RLE
        size_t o = 0, q = p;
        while (q < p + size && o < want) {
          const int n = (signed char)d[q++];
          if (n < 0) { const size_t k = (size_t)(-n); if (q + k > p + size || o + k > want) return "bad RLE data"; memcpy(&tmp[o], &d[q], k); q += k; o += k; }
          else { const size_t k = (size_t)n + 1; if (q >= p + size || o + k > want) return "bad RLE data"; memset(&tmp[o], d[q++], k); o += k; }
        }
        if (o != want) return "bad RLE data";
      } else {
        uLongf out_len = (uLongf)want;
        if (uncompress(tmp.data(), &out_len, &d[p], size) != Z_OK || out_len != want) return "bad ZIP data";
      }
      for (size_t i = 1; i < want; ++i) tmp[i] = (unsigned char)(tmp[i - 1] + tmp[i] - 128);  // predictor
      raw.resize(want);
      const size_t half = (want + 1) / 2;
      for (size_t i = 0; i < want; ++i) raw[i] = (i & 1) ? tmp[half + i / 2] : tmp[i / 2];  // de-interleave
    }
    if (raw.size() != want) return "bad chunk size";
    p += size;
    for (uint32_t l = 0; l < lines; ++l) {
      const uint32_t row = (uint32_t)(y0 - win[1]) + l;
      for (uint32_t ch = 0; ch < img->channels; ++ch) {
        const int k = ci[ch];
        const unsigned char* src = &raw[chunk_line * l + coff[k]];
        for (uint32_t x = 0; x < cols; ++x) {
          float v;
          if (chans[k].type == 1) { uint16_t h; memcpy(&h, src + 2 * (size_t)x, 2); v = half_to_float(h); }
          else if (chans[k].type == 2) memcpy(&v, src + 4 * (size_t)x, 4);
          else { uint32_t u; memcpy(&u, src + 4 * (size_t)x, 4); v = (float)u; }
          img->pixels[((size_t)row * W + x_first + x) * img->channels + ch] = v;
        }
      }
    }
  }
  return "";
}

std::string load_float_image(const char* path, HostImage* img) {
  FILE* f = fopen(path, "rb");
  if (!f) return std::string("Failed to open image \"") + path + "\".";  // src/envmap.rs:49
  unsigned char magic[4] = {0, 0, 0, 0};
  const size_t got = fread(magic, 1, 4, f);
  rewind(f);
  std::string e;
  try {  // nothing is thrown across the C ABI: a header that asks for an absurd image ends as a decode error, not as std::bad_alloc
    if (got >= 2 && magic[0] == 'P' && (magic[1] == 'F' || magic[1] == 'f')) e = load_pfm(f, img);
    else if (got >= 2 && magic[0] == '#' && magic[1] == '?') e = load_hdr(f, img);
    else if (got == 4 && magic[0] == 0x76 && magic[1] == 0x2f && magic[2] == 0x31 && magic[3] == 0x01) e = load_exr(f, img);
    else e = "unrecognised format";
  } catch (const std::exception& ex) { e = std::string("out of memory or malformed: ") + ex.what(); }
  fclose(f);
  if (!e.empty()) return std::string("Failed to decode image \"") + path + "\". (" + e + ")";  // src/envmap.rs:53
  return "";
}

}  // namespace rt

// ---- optional libraries, resolved on first use (dyn_api.h) -------------------------------------------------------------------------
#include <dlfcn.h>

#include <mutex>

#include "dyn_api.h"

namespace rt {
namespace {
void* open_first(const char* const* names, bool noload_first) {
  if (noload_first)
    for (const char* const* n = names; *n; ++n)
      if (void* h = dlopen(*n, RTLD_NOW | RTLD_NOLOAD)) return h;
  for (const char* const* n = names; *n; ++n)
    if (void* h = dlopen(*n, RTLD_NOW | RTLD_LOCAL)) return h;
  return nullptr;
}
}  // namespace

const RcclApi* rccl_api(std::string* err) {
  static RcclApi api{};
  static std::string failure;
  static std::once_flag once;
  std::call_once(once, [] {
    static const char* const names[] = {"librccl.so.1", "librccl.so", nullptr};
    void* h = open_first(names, true);
    if (!h) { failure = std::string("RCCL is not available (dlopen librccl.so.1: ") + (dlerror() ? dlerror() : "not found") + ")"; return; }
    struct { const char* name; void** slot; } syms[] = {
        {"ncclGetUniqueId", (void**)&api.GetUniqueId}, {"ncclCommInitRank", (void**)&api.CommInitRank}, {"ncclCommUserRank", (void**)&api.CommUserRank},
        {"ncclCommCount", (void**)&api.CommCount},     {"ncclCommDestroy", (void**)&api.CommDestroy},   {"ncclAllGather", (void**)&api.AllGather},
        {"ncclGetErrorString", (void**)&api.GetErrorString}};
    for (auto& s : syms) {
      *s.slot = dlsym(h, s.name);
      if (!*s.slot) { failure = std::string("RCCL lacks the symbol ") + s.name; return; }
    }
  });
  if (!failure.empty()) { if (err) *err = failure; return nullptr; }
  return &api;
}

namespace {
struct RoctxApi { int (*push)(const char*) = nullptr; int (*pop)() = nullptr; };
const RoctxApi& roctx_api() {
  static RoctxApi api;
  static std::once_flag once;
  std::call_once(once, [] {
    static const char* const names[] = {"librocprofiler-sdk-roctx.so.1", "librocprofiler-sdk-roctx.so", "libroctx64.so.4", "libroctx64.so", nullptr};
    void* h = open_first(names, true);
    if (!h) return;
    api.push = (int (*)(const char*))dlsym(h, "roctxRangePushA");
    api.pop = (int (*)())dlsym(h, "roctxRangePop");
    if (!api.push || !api.pop) api.push = nullptr, api.pop = nullptr;
  });
  return api;
}
}  // namespace
void roctx_push(const char* name) { const RoctxApi& a = roctx_api(); if (a.push) a.push(name); }
void roctx_pop() { const RoctxApi& a = roctx_api(); if (a.pop) a.pop(); }

}  // namespace rt
