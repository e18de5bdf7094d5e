// texture.hip — K8 of SURVEY.md §2.1: the mip chain the reference gets from HalaImage::gen_mipmaps
// (src/scene/loader/gpu_uploader.rs:366-400) as a 2x2 box-filter kernel over linear RGBA32F texels.
// The fetch side (bilinear / trilinear, REPEAT) lives in shading.h::tex_sample.
#include <hip/hip_runtime.h>

#include "kernels.h"

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

void launch_mip_downsample(const float4* src, uint32_t sw, uint32_t sh, float4* dst, uint32_t dw, uint32_t dh, hipStream_t s) {
  const uint32_t n = dw * dh;
  hipLaunchKernelGGL(k_mip_downsample, dim3((n + 255u) / 256u), dim3(256), 0, s, src, sw, sh, dst, dw, dh);
}

}  // namespace rt
