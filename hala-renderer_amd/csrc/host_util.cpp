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
// ZIPS / ZIP, HALF / FLOAT / UINT samples, channels R G B [A] or Y; mip / rip-mapped tiles, deep, multi-part and PIZ/PXR24/B44/DWA
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
  if (compression < 0 || compression > 3) return "only NONE / RLE / ZIPS / ZIP compressed OpenEXR files are supported";
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
  const uint32_t block = compression == 3 ? 16u : 1u;
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
    else {
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
