// jpeg_decode.cpp — Huffman-coded 8-bit JPEG (SOF0 / SOF1 sequential, SOF2 progressive) to RGBA8 for the glTF loader: what the
// `image` crate does for `image/jpeg` textures behind gltf::import (src/scene/loader/gltf_loader.rs:123, :391-429).
// Greyscale and YCbCr with any 1x/2x sampling factors, restart intervals, 16-bit quantisation tables, any number of scans
// (interleaved or one component each; spectral selection and successive approximation, ITU-T T.81 annex G); chroma is upsampled
// by replication.  Lossless, differential, arithmetic-coded and CMYK files are refused.  Every scan only fills the coefficient
// arrays; the blocks are dequantised and transformed once, at the end of the image.  The inverse DCT is the separable float form;
// results agree with libjpeg's to within the usual +-1..2 levels (tests compare against PIL, and the progressive against the
// sequential encoding of the same coefficients byte for byte).
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

struct Comp {
  int id = 0, h = 1, v = 1, tq = 0, td = 0, ta = 0;
  long long pred = 0;  // DC predictor (64-bit: a hostile file cannot overflow it)
  int w = 0, hgt = 0;    // plane size in samples: whole MCUs
  int bw = 0, bh = 0;    // blocks the component really has: ceil(its size / 8) — what a scan of this component alone covers (T.81 A.2.3)
  bool seen = false;     // a scan has named it: its quantisation table is latched
  uint16_t q[64] = {0};
  std::vector<int16_t> coef;  // [block row][block column][64], natural order, whole MCUs
  std::vector<uint8_t> plane;
};

}  // namespace

// returns false if the data is not a JPEG this decoder handles
bool decode_jpeg(const std::vector<uint8_t>& raw, uint32_t* w, uint32_t* h, std::vector<uint8_t>* rgba) {
  if (raw.size() < 4 || raw[0] != 0xff || raw[1] != 0xd8) return false;
  uint16_t qt[4][64] = {};
  bool qt_present[4] = {false, false, false, false};
  Huff dc[4], ac[4];
  std::vector<Comp> comps;
  int width = 0, height = 0, restart = 0, hmax = 1, vmax = 1, mcux = 0, mcuy = 0, scans = 0;
  bool baseline = false, progressive = false;
  size_t p = 2;
  auto be16 = [&](size_t at) { return (int)((raw[at] << 8) | raw[at + 1]); };
  for (;;) {
    while (p < raw.size() && raw[p] != 0xff) ++p;
    while (p < raw.size() && raw[p] == 0xff) ++p;
    if (p >= raw.size()) break;   // no EOI: keep what the scans gave (truncated files decode as far as they go)
    const uint8_t m = raw[p++];
    if (m == 0xd9) break;         // EOI
    if (m == 0x00 || m == 0x01 || (m >= 0xd0 && m <= 0xd7)) continue;  // a stuffed byte left over from a scan, TEM, RSTn
    if (p + 2 > raw.size()) return false;
    const int len = be16(p);
    if (len < 2 || p + (size_t)len > raw.size()) return false;
    const size_t seg = p + 2, seg_end = p + (size_t)len;
    if (m == 0xdb) {  // DQT
      size_t q = seg;
      while (q < seg_end) {
        const int pq = raw[q] >> 4, tq = raw[q] & 15;
        ++q;
        if (tq > 3 || pq > 1 || q + (size_t)(pq ? 128 : 64) > seg_end) return false;
        for (int i = 0; i < 64; ++i) { qt[tq][kZigzag[i]] = pq ? (uint16_t)be16(q + 2 * (size_t)i) : raw[q + (size_t)i]; }
        qt_present[tq] = true;
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
    } else if (m == 0xc0 || m == 0xc1 || m == 0xc2) {  // SOF0 / SOF1 / SOF2: baseline / extended sequential / progressive, Huffman
      if (!comps.empty() || seg + 6 > seg_end || raw[seg] != 8) return false;
      baseline = m == 0xc0; progressive = m == 0xc2;
      height = be16(seg + 1); width = be16(seg + 3);
      const int nc = raw[seg + 5];
      if ((nc != 1 && nc != 3) || width <= 0 || height <= 0 || width > 32768 || height > 32768 || seg + 6 + (size_t)nc * 3 > seg_end) return false;
      comps.resize((size_t)nc);
      for (int i = 0; i < nc; ++i) {
        Comp& c = comps[(size_t)i];
        c.id = raw[seg + 6 + (size_t)i * 3];
        c.h = raw[seg + 7 + (size_t)i * 3] >> 4; c.v = raw[seg + 7 + (size_t)i * 3] & 15;
        c.tq = raw[seg + 8 + (size_t)i * 3];
        if (c.h < 1 || c.h > 2 || c.v < 1 || c.v > 2 || c.tq > 3) return false;
        for (int j = 0; j < i; ++j) if (comps[(size_t)j].id == c.id) return false;
        hmax = std::max(hmax, c.h); vmax = std::max(vmax, c.v);
      }
      if (nc == 1) { comps[0].h = comps[0].v = 1; hmax = vmax = 1; }  // a single component is never interleaved: its factors mean nothing (A.2.2)
      mcux = (width + 8 * hmax - 1) / (8 * hmax); mcuy = (height + 8 * vmax - 1) / (8 * vmax);
      size_t total = 0;
      for (Comp& c : comps) {
        c.w = mcux * c.h * 8; c.hgt = mcuy * c.v * 8;
        c.bw = ((width * c.h + hmax - 1) / hmax + 7) / 8; c.bh = ((height * c.v + vmax - 1) / vmax + 7) / 8;
        total += (size_t)c.w * c.hgt;
      }
      if (total > ((size_t)1 << 29)) return false;  // 1 GiB of coefficients
      for (Comp& c : comps) c.coef.assign((size_t)c.w * c.hgt, 0);
    } else if (m == 0xc3 || (m >= 0xc5 && m <= 0xcf && m != 0xc8 && m != 0xcc)) {
      return false;  // lossless, differential, arithmetic
    } else if (m == 0xdd) {
      if (seg + 2 > seg_end) return false;
      restart = be16(seg);
    } else if (m == 0xda) {  // SOS: one scan — all components interleaved, or a subset (one component: its own block raster)
      if (comps.empty() || seg + 1 > seg_end) return false;
      const int ns = raw[seg];
      if (ns < 1 || ns > (int)comps.size() || seg + 1 + (size_t)ns * 2 + 3 > seg_end) return false;
      Comp* sc[3] = {nullptr, nullptr, nullptr};
      for (int i = 0; i < ns; ++i) {
        const int cid = raw[seg + 1 + (size_t)i * 2], tb = raw[seg + 2 + (size_t)i * 2];
        // table selectors index dc[4] / ac[4]; a baseline (SOF0) scan may only name tables 0 and 1 (T.81 B.2.3)
        if ((tb >> 4) > 3 || (tb & 15) > 3 || (baseline && ((tb >> 4) > 1 || (tb & 15) > 1))) return false;
        for (Comp& c : comps) if (c.id == cid) { c.td = tb >> 4; c.ta = tb & 15; sc[i] = &c; }
        if (!sc[i]) return false;
        for (int j = 0; j < i; ++j) if (sc[j] == sc[i]) return false;
      }
      const size_t par = seg + 1 + (size_t)ns * 2;
      const int ss = raw[par], se = raw[par + 1], ah = raw[par + 2] >> 4, al = raw[par + 2] & 15;
      if (progressive) {
        if (ss > se || se > 63 || ah > 13 || al > 13 || (ss == 0 && se != 0) || (ss > 0 && ns != 1) || (ah != 0 && ah != al + 1)) return false;
      } else if (ss != 0 || se != 63 || ah != 0 || al != 0) return false;
      if (++scans > 1024) return false;
      for (int i = 0; i < ns; ++i) {
        Comp& c = *sc[i];
        if (!c.seen) { if (!qt_present[c.tq]) return false; memcpy(c.q, qt[c.tq], sizeof(c.q)); c.seen = true; }
        if ((ss == 0 && ah == 0 && !dc[c.td].present) || ((se > 0) && !ac[c.ta].present)) return false;
        c.pred = 0;
      }
      // MCUs of this scan: interleaved -> the frame's MCU grid, each holding h x v blocks per component; alone -> the component's blocks
      const bool alone = ns == 1;
      const int ux = alone ? sc[0]->bw : mcux, uy = alone ? sc[0]->bh : mcuy;
      BitReader br{&raw[seg_end], raw.data() + raw.size()};
      int until_restart = restart, eobrun = 0;
      const int p1 = 1 << al, m1 = -(1 << al);
      for (int my = 0; my < uy; ++my)
        for (int mx = 0; mx < ux; ++mx) {
          if (restart && until_restart == 0) {  // RSTn: byte-align, skip the marker, reset the predictors and the end-of-band run
            br.reset();
            const uint8_t* q = br.p;
            while (q + 1 < br.end && !(q[0] == 0xff && q[1] >= 0xd0 && q[1] <= 0xd7)) ++q;
            if (q + 1 < br.end) br.p = q + 2;
            for (int i = 0; i < ns; ++i) sc[i]->pred = 0;
            eobrun = 0;
            until_restart = restart;
          }
          for (int i = 0; i < ns; ++i) {
            Comp& c = *sc[i];
            const int nby = alone ? 1 : c.v, nbx = alone ? 1 : c.h;
            for (int by = 0; by < nby; ++by)
              for (int bx = 0; bx < nbx; ++bx) {
                const int brow = alone ? my : my * c.v + by, bcol = alone ? mx : mx * c.h + bx;
                int16_t* blk = &c.coef[((size_t)brow * (size_t)(c.w / 8) + (size_t)bcol) * 64];
                if (!progressive) {  // sequential: the whole block
                  const int s = decode_symbol(br, dc[c.td]);
                  if (s < 0 || s > 11) return false;
                  c.pred += s ? extend(br.receive(s), s) : 0;
                  blk[0] = (int16_t)c.pred;
                  for (int k = 1; k < 64;) {
                    const int rs = decode_symbol(br, ac[c.ta]);
                    if (rs < 0) return false;
                    const int r = rs >> 4, sz = rs & 15;
                    if (sz == 0) { if (r == 15) { k += 16; continue; } break; }
                    k += r;
                    if (k > 63) return false;
                    blk[kZigzag[k]] = (int16_t)extend(br.receive(sz), sz);
                    ++k;
                  }
                } else if (ss == 0) {
                  if (ah == 0) {  // DC, first pass: the difference of the point-transformed values (G.1.2.1)
                    const int s = decode_symbol(br, dc[c.td]);
                    if (s < 0 || s > 11) return false;
                    c.pred += s ? extend(br.receive(s), s) : 0;
                    blk[0] = (int16_t)(c.pred * (1 << al));
                  } else if (br.bit()) blk[0] = (int16_t)(blk[0] | p1);  // DC, refinement: one more bit
                } else if (ah == 0) {  // AC band, first pass (G.1.2.2)
                  if (eobrun > 0) { --eobrun; continue; }
                  for (int k = ss; k <= se;) {
                    const int rs = decode_symbol(br, ac[c.ta]);
                    if (rs < 0) return false;
                    const int r = rs >> 4, sz = rs & 15;
                    if (sz == 0) {
                      if (r == 15) { k += 16; continue; }
                      eobrun = (1 << r) - 1;  // this block ends here, and so do the next eobrun blocks
                      if (r) eobrun += br.receive(r);
                      break;
                    }
                    k += r;
                    if (k > se) return false;
                    blk[kZigzag[k]] = (int16_t)(extend(br.receive(sz), sz) * (1 << al));
                    ++k;
                  }
                } else {  // AC band, refinement (G.1.2.3): one more bit for every coefficient that is already non-zero, new +-1 << al ones in between
                  auto refine = [&](int16_t* cp) {
                    if (br.bit() && (*cp & p1) == 0) *cp = (int16_t)(*cp + (*cp >= 0 ? p1 : m1));
                  };
                  int k = ss;
                  if (eobrun == 0) {
                    for (; k <= se; ++k) {
                      const int rs = decode_symbol(br, ac[c.ta]);
                      if (rs < 0) return false;
                      int r = rs >> 4;
                      const int sz = rs & 15;
                      int value = 0;
                      if (sz) {
                        if (sz != 1) return false;
                        value = br.bit() ? p1 : m1;
                      } else if (r != 15) {
                        eobrun = 1 << r;  // counts this block too
                        if (r) eobrun += br.receive(r);
                        break;
                      }
                      // skip r zero coefficients (refining the non-zero ones on the way), stop at the position of the new one
                      for (; k <= se; ++k) {
                        int16_t* cp = &blk[kZigzag[k]];
                        if (*cp != 0) refine(cp);
                        else if (--r < 0) break;
                      }
                      if (sz) {
                        if (k > se) return false;
                        blk[kZigzag[k]] = (int16_t)value;
                      }
                    }
                  }
                  if (eobrun > 0) {
                    for (; k <= se; ++k) {
                      int16_t* cp = &blk[kZigzag[k]];
                      if (*cp != 0) refine(cp);
                    }
                    --eobrun;
                  }
                }
              }
          }
          if (restart) --until_restart;
        }
      // the parser goes on behind the entropy-coded data: at the marker the reader ran into, or from where it stopped
      p = (size_t)(br.p - raw.data());
      continue;
    }
    p = seg_end;
  }
  if (comps.empty() || scans == 0) return false;
  for (Comp& c : comps) {
    if (!c.seen) return false;
    c.plane.assign((size_t)c.w * c.hgt, 128);
    const int bpr = c.w / 8;
    for (int brow = 0; brow < c.hgt / 8; ++brow)
      for (int bcol = 0; bcol < bpr; ++bcol) {
        const int16_t* q = &c.coef[((size_t)brow * (size_t)bpr + (size_t)bcol) * 64];
        float blk[64];
        for (int k = 0; k < 64; ++k) blk[k] = (float)((int)q[k] * (int)c.q[k]);
        idct8x8(blk, &c.plane[(size_t)(brow * 8) * c.w + (size_t)bcol * 8], c.w);
      }
    std::vector<int16_t>().swap(c.coef);
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
