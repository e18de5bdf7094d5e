// texture.hip — K8 of SURVEY.md §2.1: the mip chain the reference gets from HalaImage::gen_mipmaps
// (src/scene/loader/gpu_uploader.rs:366-400) as a 2x2 box-filter kernel over linear RGBA32F texels.
// The fetch side (bilinear / trilinear, REPEAT) lives in shading.h::tex_sample.
#include <hip/hip_runtime.h>

#include "kernels.h"
#include "rt_math.h"

namespace rt {

// dst(x, y) = ((s(2x,2y) + s(2x+1,2y)) + (s(2x,2y+1) + s(2x+1,2y+1))) * 0.25, source coordinates clamped to the edge
// (odd sizes); RENDER_SPEC §7.4.
__global__ void __launch_bounds__(256) k_mip_downsample(const float4* __restrict__ src, uint32_t sw, uint32_t sh, float4* __restrict__ dst,
                                                         uint32_t dw, uint32_t dh) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= dw * dh) return;
  const uint32_t y = i / dw, x = i - y * dw;
  const uint32_t x0 = min(2u * x, sw - 1u), x1 = min(2u * x + 1u, sw - 1u);
  const uint32_t y0 = min(2u * y, sh - 1u), y1 = min(2u * y + 1u, sh - 1u);
  const float4 a = src[(size_t)y0 * sw + x0], b = src[(size_t)y0 * sw + x1];
  const float4 c = src[(size_t)y1 * sw + x0], d = src[(size_t)y1 * sw + x1];
  float4 r;
  r.x = ((a.x + b.x) + (c.x + d.x)) * 0.25f;
  r.y = ((a.y + b.y) + (c.y + d.y)) * 0.25f;
  r.z = ((a.z + b.z) + (c.z + d.z)) * 0.25f;
  r.w = ((a.w + b.w) + (c.w + d.w)) * 0.25f;
  dst[i] = r;
}

// ---- 8-bit images (RENDER_SPEC 7.4): RGBA bytes, 4 B per texel, tiled 4x4 -----------------------------------------------------------
// level 0: row-major bytes as uploaded -> tiled
__global__ void __launch_bounds__(256) k_tile8(const uint32_t* __restrict__ src, uint32_t w, uint32_t h, uint32_t* __restrict__ dst) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= w * h) return;
  const uint32_t y = i / w, x = i - y * w;
  dst[tex_tiled_index(x, y, w)] = src[i];
}
// one byte channel -> linear float: sRGB through the table, UNORM / 255; alpha is always UNORM
RT_DI float4 tex8_decode(uint32_t t, uint32_t format, const float* __restrict__ lut) {
  const uint32_t r = t & 0xffu, g = (t >> 8) & 0xffu, b = (t >> 16) & 0xffu, a = t >> 24;
  if (format == kTexSrgb8) return make_float4(lut[r], lut[g], lut[b], (float)a / 255.0f);
  return make_float4((float)r / 255.0f, (float)g / 255.0f, (float)b / 255.0f, (float)a / 255.0f);
}
// linear float -> byte.  UNORM: floor(x * 255 + 0.5) clamped.  sRGB: the code whose decoded value is nearest — the number of midpoints
// thr[k] = (lut[k] + lut[k + 1]) / 2 (k = 0 .. 254) that lie below x, found by bisection (the table is monotonic).
RT_DI uint32_t tex8_encode_unorm(float x) { const float v = floorf(x * 255.0f + 0.5f); return (uint32_t)(v < 0.0f ? 0.0f : (v > 255.0f ? 255.0f : v)); }
RT_DI uint32_t tex8_encode_srgb(float x, const float* __restrict__ thr) {
  uint32_t lo = 0, hi = 255;  // answer in [lo, hi]: thr[k] < x for all k < answer
  while (lo < hi) {
    const uint32_t mid = (lo + hi) >> 1;
    if (thr[mid] < x) lo = mid + 1u; else hi = mid;
  }
  return lo;
}
// mip level l from level l - 1 (both tiled): decode the four source texels, box-filter in linear space exactly like the float path
// (((a + b) + (c + d)) * 0.25, source coordinates clamped to the edge), encode back to bytes — like the 8-bit mip chains the
// reference's gen_mipmaps blits (gpu_uploader.rs:400)
__global__ void __launch_bounds__(256) k_mip_downsample8(const uint32_t* __restrict__ src, uint32_t sw, uint32_t sh, uint32_t* __restrict__ dst, uint32_t dw,
                                                          uint32_t dh, uint32_t format, const float* __restrict__ lut, const float* __restrict__ thr) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= dw * dh) return;
  const uint32_t y = i / dw, x = i - y * dw;
  const uint32_t x0 = min(2u * x, sw - 1u), x1 = min(2u * x + 1u, sw - 1u);
  const uint32_t y0 = min(2u * y, sh - 1u), y1 = min(2u * y + 1u, sh - 1u);
  const float4 a = tex8_decode(src[tex_tiled_index(x0, y0, sw)], format, lut), b = tex8_decode(src[tex_tiled_index(x1, y0, sw)], format, lut);
  const float4 c = tex8_decode(src[tex_tiled_index(x0, y1, sw)], format, lut), d = tex8_decode(src[tex_tiled_index(x1, y1, sw)], format, lut);
  const float r4[4] = {((a.x + b.x) + (c.x + d.x)) * 0.25f, ((a.y + b.y) + (c.y + d.y)) * 0.25f, ((a.z + b.z) + (c.z + d.z)) * 0.25f,
                       ((a.w + b.w) + (c.w + d.w)) * 0.25f};
  uint32_t out = tex8_encode_unorm(r4[3]) << 24;
#pragma unroll
  for (int k = 0; k < 3; ++k) out |= (format == kTexSrgb8 ? tex8_encode_srgb(r4[k], thr) : tex8_encode_unorm(r4[k])) << (8 * k);
  dst[tex_tiled_index(x, y, dw)] = out;
}
void launch_tile8(const uint32_t* src, uint32_t w, uint32_t h, uint32_t* dst, hipStream_t s) {
  hipLaunchKernelGGL(k_tile8, dim3((w * h + 255u) / 256u), dim3(256), 0, s, src, w, h, dst);
}
void launch_mip_downsample8(const uint32_t* src, uint32_t sw, uint32_t sh, uint32_t* dst, uint32_t dw, uint32_t dh, uint32_t format, const float* lut,
                            const float* thr, hipStream_t s) {
  hipLaunchKernelGGL(k_mip_downsample8, dim3((dw * dh + 255u) / 256u), dim3(256), 0, s, src, sw, sh, dst, dw, dh, format, lut, thr);
}

void launch_mip_downsample(const float4* src, uint32_t sw, uint32_t sh, float4* dst, uint32_t dw, uint32_t dh, hipStream_t s) {
  const uint32_t n = dw * dh;
  hipLaunchKernelGGL(k_mip_downsample, dim3((n + 255u) / 256u), dim3(256), 0, s, src, sw, sh, dst, dw, dh);
}

}  // namespace rt
