// jpeg_decode.cpp — baseline sequential JPEG (SOF0 / SOF1, 8-bit, Huffman) to RGBA8 for the glTF loader: what the `image`
// crate does for `image/jpeg` textures behind gltf::import (src/scene/loader/gltf_loader.rs:123, :391-429).
// Greyscale and YCbCr with any 1x/2x sampling factors, restart intervals, 16-bit quantisation tables; chroma is upsampled
// by replication.  Progressive (SOF2), lossless, arithmetic-coded and CMYK files are refused.  The inverse DCT is the
// separable float form; results agree with libjpeg's to within the usual +-1..2 levels (tests compare against PIL).
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstring>
#include <vector>

namespace rt {

namespace {

struct Huff {
  uint8_t bits[17] = {0};
  uint8_t vals[256] = {0};
  int mincode[17], maxcode[18], valptr[17];
  bool present = false;
  void build() {
    int code = 0, k = 0;
    for (int l = 1; l <= 16; ++l) {
      valptr[l] = k; mincode[l] = code;
      code += bits[l]; k += bits[l];
      maxcode[l] = bits[l] ? code - 1 : -1;
      code <<= 1;
    }
    maxcode[17] = 0x7fffffff;
  }
};

struct BitReader {
  const uint8_t* p; const uint8_t* end;
  uint32_t acc = 0; int n = 0;
  bool marker = false;  // ran into a marker: feed zeros
  int bit() {
    if (n == 0) {
      uint8_t b = 0;
      if (!marker && p < end) {
        b = *p++;
        if (b == 0xff) {
          if (p < end && *p == 0x00) ++p;           // stuffed zero
          else { marker = true; --p; b = 0; }      // a marker: leave it for the caller
        }
      }
      acc = b; n = 8;
    }
    --n;
    return (acc >> n) & 1;
  }
  int receive(int s) { int v = 0; for (int i = 0; i < s; ++i) v = (v << 1) | bit(); return v; }
  void reset() { acc = 0; n = 0; marker = false; }
};

int decode_symbol(BitReader& br, const Huff& h) {
  int code = 0;
  for (int l = 1; l <= 16; ++l) {
    code = (code << 1) | br.bit();
    if (h.maxcode[l] >= 0 && code <= h.maxcode[l] && code >= h.mincode[l]) return h.vals[h.valptr[l] + code - h.mincode[l]];
  }
  return -1;
}
int extend(int v, int s) { return v < (1 << (s - 1)) ? v - (1 << s) + 1 : v; }

const uint8_t kZigzag[64] = {0, 1, 8, 16, 9, 2, 3, 10, 17, 24, 32, 25, 18, 11, 4, 5, 12, 19, 26, 33, 40, 48, 41, 34, 27, 20, 13, 6, 7, 14, 21, 28,
                             35, 42, 49, 56, 57, 50, 43, 36, 29, 22, 15, 23, 30, 37, 44, 51, 58, 59, 52, 45, 38, 31, 39, 46, 53, 60, 61, 54, 47, 55, 62, 63};

void idct8x8(const float* in, uint8_t* out, int stride) {
  static float c[8][8];
  static bool init = false;
  if (!init) {
    for (int x = 0; x < 8; ++x)
      for (int u = 0; u < 8; ++u) c[x][u] = (u == 0 ? 0.35355339059327379f : 0.5f) * std::cos((2 * x + 1) * u * 3.14159265358979323846 / 16.0);
    init = true;
  }
  float tmp[64];
  for (int y = 0; y < 8; ++y)      // rows: over u
    for (int x = 0; x < 8; ++x) {
      float s = 0.0f;
      for (int u = 0; u < 8; ++u) s += c[x][u] * in[y * 8 + u];
      tmp[y * 8 + x] = s;
    }
  for (int x = 0; x < 8; ++x)      // columns: over v
    for (int y = 0; y < 8; ++y) {
      float s = 0.0f;
      for (int v = 0; v < 8; ++v) s += c[y][v] * tmp[v * 8 + x];
      const int q = (int)std::lrintf(s + 128.0f);
      out[y * stride + x] = (uint8_t)(q < 0 ? 0 : (q > 255 ? 255 : q));
    }
}

struct Comp { int id = 0, h = 1, v = 1, tq = 0, td = 0, ta = 0, pred = 0, w = 0, hgt = 0; std::vector<uint8_t> plane; };

}  // namespace

// returns false if the data is not a JPEG this decoder handles
bool decode_jpeg(const std::vector<uint8_t>& raw, uint32_t* w, uint32_t* h, std::vector<uint8_t>* rgba) {
  if (raw.size() < 4 || raw[0] != 0xff || raw[1] != 0xd8) return false;
  uint16_t qt[4][64] = {};
  Huff dc[4], ac[4];
  std::vector<Comp> comps;
  int width = 0, height = 0, restart = 0;
  bool baseline = false;
  size_t p = 2;
  auto be16 = [&](size_t at) { return (int)((raw[at] << 8) | raw[at + 1]); };
  for (;;) {
    while (p < raw.size() && raw[p] != 0xff) ++p;
    while (p < raw.size() && raw[p] == 0xff) ++p;
    if (p >= raw.size()) return false;
    const uint8_t m = raw[p++];
    if (m == 0xd9) return false;  // EOI before any scan
    if (m == 0x01 || (m >= 0xd0 && m <= 0xd7)) continue;
    if (p + 2 > raw.size()) return false;
    const int len = be16(p);
    if (len < 2 || p + (size_t)len > raw.size()) return false;
    const size_t seg = p + 2, seg_end = p + (size_t)len;
    if (m == 0xdb) {  // DQT
      size_t q = seg;
      while (q < seg_end) {
        const int pq = raw[q] >> 4, tq = raw[q] & 15;
        ++q;
        if (tq > 3 || q + (size_t)(pq ? 128 : 64) > seg_end) return false;
        for (int i = 0; i < 64; ++i) { qt[tq][kZigzag[i]] = pq ? (uint16_t)be16(q + 2 * (size_t)i) : raw[q + (size_t)i]; }
        q += pq ? 128 : 64;
      }
    } else if (m == 0xc4) {  // DHT
      size_t q = seg;
      while (q < seg_end) {
        const int tc = raw[q] >> 4, th = raw[q] & 15;
        ++q;
        if (tc > 1 || th > 3 || q + 16 > seg_end) return false;
        Huff& t = tc ? ac[th] : dc[th];
        int total = 0;
        for (int l = 1; l <= 16; ++l) { t.bits[l] = raw[q + (size_t)l - 1]; total += t.bits[l]; }
        q += 16;
        if (total > 256 || q + (size_t)total > seg_end) return false;
        memcpy(t.vals, &raw[q], (size_t)total);
        q += (size_t)total;
        t.build(); t.present = true;
      }
    } else if (m == 0xc0 || m == 0xc1) {  // SOF0 / SOF1: baseline / extended sequential, Huffman
      if (seg + 6 > seg_end || raw[seg] != 8) return false;
      baseline = m == 0xc0;
      height = be16(seg + 1); width = be16(seg + 3);
      const int nc = raw[seg + 5];
      if ((nc != 1 && nc != 3) || width <= 0 || height <= 0 || width > 32768 || height > 32768 || seg + 6 + (size_t)nc * 3 > seg_end) return false;
      comps.resize((size_t)nc);
      for (int i = 0; i < nc; ++i) {
        comps[(size_t)i].id = raw[seg + 6 + (size_t)i * 3];
        comps[(size_t)i].h = raw[seg + 7 + (size_t)i * 3] >> 4; comps[(size_t)i].v = raw[seg + 7 + (size_t)i * 3] & 15;
        comps[(size_t)i].tq = raw[seg + 8 + (size_t)i * 3];
        if (comps[(size_t)i].h < 1 || comps[(size_t)i].h > 2 || comps[(size_t)i].v < 1 || comps[(size_t)i].v > 2 || comps[(size_t)i].tq > 3) return false;
      }
    } else if (m == 0xc2 || m == 0xc3 || (m >= 0xc5 && m <= 0xcf && m != 0xc8 && m != 0xcc)) {
      return false;  // progressive, lossless, differential, arithmetic
    } else if (m == 0xdd) {
      if (seg + 2 > seg_end) return false;
      restart = be16(seg);
    } else if (m == 0xda) {  // SOS: one interleaved scan with every component
      if (comps.empty() || seg + 1 > seg_end) return false;
      const int ns = raw[seg];
      if (ns != (int)comps.size() || seg + 1 + (size_t)ns * 2 + 3 > seg_end) return false;
      for (int i = 0; i < ns; ++i) {
        const int cid = raw[seg + 1 + (size_t)i * 2], tb = raw[seg + 2 + (size_t)i * 2];
        bool found = false;
        // table selectors index dc[4] / ac[4]; a baseline (SOF0) scan may only name tables 0 and 1 (T.81 B.2.3)
        if ((tb >> 4) > 3 || (tb & 15) > 3 || (baseline && ((tb >> 4) > 1 || (tb & 15) > 1))) return false;
        for (Comp& c : comps) if (c.id == cid) { c.td = tb >> 4; c.ta = tb & 15; found = true; }
        if (!found) return false;
      }
      p = seg_end;
      break;
    }
    p = seg_end;
  }
  int hmax = 1, vmax = 1;
  for (const Comp& c : comps) { hmax = std::max(hmax, c.h); vmax = std::max(vmax, c.v); }
  const int mcux = (width + 8 * hmax - 1) / (8 * hmax), mcuy = (height + 8 * vmax - 1) / (8 * vmax);
  for (Comp& c : comps) {
    if (!dc[c.td].present || !ac[c.ta].present) return false;
    c.w = mcux * c.h * 8; c.hgt = mcuy * c.v * 8;
    c.plane.assign((size_t)c.w * c.hgt, 128);
  }
  BitReader br{&raw[p], raw.data() + raw.size()};
  int until_restart = restart;
  for (int my = 0; my < mcuy; ++my)
    for (int mx = 0; mx < mcux; ++mx) {
      if (restart && until_restart == 0) {  // RSTn: byte-align, skip the marker, reset the predictors
        br.reset();
        const uint8_t* q = br.p;
        while (q + 1 < br.end && !(q[0] == 0xff && q[1] >= 0xd0 && q[1] <= 0xd7)) ++q;
        if (q + 1 < br.end) br.p = q + 2;
        for (Comp& c : comps) c.pred = 0;
        until_restart = restart;
      }
      for (Comp& c : comps)
        for (int by = 0; by < c.v; ++by)
          for (int bx = 0; bx < c.h; ++bx) {
            float blk[64] = {0};
            const int s = decode_symbol(br, dc[c.td]);
            if (s < 0 || s > 11) return false;
            const int diff = s ? extend(br.receive(s), s) : 0;
            c.pred += diff;
            blk[0] = (float)(c.pred * (int)qt[c.tq][0]);
            for (int k = 1; k < 64;) {
              const int rs = decode_symbol(br, ac[c.ta]);
              if (rs < 0) return false;
              const int r = rs >> 4, sz = rs & 15;
              if (sz == 0) { if (r == 15) { k += 16; continue; } break; }
              k += r;
              if (k > 63) return false;
              const int v = extend(br.receive(sz), sz);
              blk[kZigzag[k]] = (float)(v * (int)qt[c.tq][kZigzag[k]]);
              ++k;
            }
            idct8x8(blk, &c.plane[(size_t)((my * c.v + by) * 8) * c.w + (size_t)(mx * c.h + bx) * 8], c.w);
          }
      if (restart) --until_restart;
    }
  *w = (uint32_t)width; *h = (uint32_t)height;
  rgba->assign((size_t)width * height * 4, 255);
  for (int y = 0; y < height; ++y)
    for (int x = 0; x < width; ++x) {
      uint8_t* px = &(*rgba)[((size_t)y * width + x) * 4];
      auto at = [&](const Comp& c) { return (float)c.plane[(size_t)(y * c.v / vmax) * c.w + (size_t)(x * c.h / hmax)]; };
      if (comps.size() == 1) { px[0] = px[1] = px[2] = (uint8_t)at(comps[0]); continue; }
      const float Y = at(comps[0]), cb = at(comps[1]) - 128.0f, cr = at(comps[2]) - 128.0f;
      const float rgb[3] = {Y + 1.402f * cr, Y - 0.344136f * cb - 0.714136f * cr, Y + 1.772f * cb};
      for (int k = 0; k < 3; ++k) { const int q = (int)std::lrintf(rgb[k]); px[k] = (uint8_t)(q < 0 ? 0 : (q > 255 ? 255 : q)); }
    }
  return true;
}

}  // namespace rt
