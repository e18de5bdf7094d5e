// envmap.hip — A1 of SURVEY.md §8a: EnvMap::build_distribution_maps (src/envmap.rs:239-388) as HIP kernels.
//
// The reference builds the tables on the CPU with rayon; its results depend on the ORDER of the f32 additions
// (sequential fold over all pixels for total_sum :275, sequential running sum along each row :282-290, sequential
// prefix over the row sums :300-303).  To stay bit-exact the additions are kept sequential here and the
// parallelism comes from everything around them: rows are independent (one lane per row, tiles staged through LDS
// so that HBM reads stay coalesced), the divisions and the W*H binary searches are embarrassingly parallel, and
// the single-lane folds are fed from double-buffered LDS tiles so the adder never waits on HBM.
#include <hip/hip_runtime.h>

#include "kernels.h"

namespace rt {

namespace {

__device__ __forceinline__ float env_lum(float4 p) {
  // src/envmap.rs:249-251 — (0.212671*r + 0.715160*g) + 0.072169*b, never fused (-ffp-contract=off)
  return 0.212671f * p.x + 0.715160f * p.y + 0.072169f * p.z;
}

// total_sum = fold(0, +) over all pixels in row-major order (src/envmap.rs:275).  One workgroup; all 256 threads
// stream luminances into a double-buffered LDS tile, thread 0 folds the previous tile meanwhile.
constexpr int kFoldTile = 2048;
__global__ void __launch_bounds__(256) k_env_total(const float4* __restrict__ rgba, unsigned long long n, float* __restrict__ total_sum) {
  __shared__ float tile[2][kFoldTile];
  float acc = 0.0f;
  const unsigned long long tiles = (n + kFoldTile - 1) / kFoldTile;
  for (unsigned long long t = 0; t <= tiles; ++t) {
    if (t < tiles) {
      const unsigned long long base = t * kFoldTile;
      for (int k = threadIdx.x; k < kFoldTile; k += 256) {
        const unsigned long long i = base + k;
        tile[t & 1][k] = i < n ? env_lum(rgba[i]) : 0.0f;
      }
    }
    if (t > 0 && threadIdx.x == 0) {
      const unsigned long long base = (t - 1) * kFoldTile;
      const int cnt = (int)((n - base) < (unsigned long long)kFoldTile ? (n - base) : (unsigned long long)kFoldTile);
      const float* src = tile[(t - 1) & 1];
      for (int k = 0; k < cnt; ++k) acc = acc + src[k];
    }
    __syncthreads();
  }
  if (threadIdx.x == 0) *total_sum = acc;
}

// per-row running sums (src/envmap.rs:282-290): 64 rows per workgroup, 64-column tiles through LDS.
constexpr int kRowTile = 64;
__global__ void __launch_bounds__(256) k_env_row_scan(const float4* __restrict__ rgba, uint32_t W, uint32_t H, float* __restrict__ cdf_2d,
                                                       float* __restrict__ row_sum) {
  __shared__ float tile[kRowTile][kRowTile + 1];
  const uint32_t row0 = blockIdx.x * kRowTile;
  float run = 0.0f;  // meaningful in threads 0..63 (one per row)
  for (uint32_t col0 = 0; col0 < W; col0 += kRowTile) {
    // coalesced load: thread t reads column (t % 64) of rows (t / 64), (t / 64) + 4, ...
    for (uint32_t r = threadIdx.x / kRowTile; r < kRowTile; r += 256 / kRowTile) {
      const uint32_t c = threadIdx.x % kRowTile;
      const uint32_t y = row0 + r, x = col0 + c;
      tile[r][c] = (y < H && x < W) ? env_lum(rgba[(size_t)y * W + x]) : 0.0f;
    }
    __syncthreads();
    if (threadIdx.x < kRowTile) {
      const uint32_t r = threadIdx.x;
      const uint32_t cnt = min((uint32_t)kRowTile, W - col0);
      for (uint32_t c = 0; c < cnt; ++c) {
        run += tile[r][c];   // row_weight_sum += weight  (:288)
        tile[r][c] = run;    // cdf_2d_row[u] = row_weight_sum (:289)
      }
    }
    __syncthreads();
    for (uint32_t r = threadIdx.x / kRowTile; r < kRowTile; r += 256 / kRowTile) {
      const uint32_t c = threadIdx.x % kRowTile;
      const uint32_t y = row0 + r, x = col0 + c;
      if (y < H && x < W) cdf_2d[(size_t)y * W + x] = tile[r][c];
    }
    __syncthreads();
  }
  if (threadIdx.x < kRowTile && row0 + threadIdx.x < H) row_sum[row0 + threadIdx.x] = run;  // *pdf_1d_value = row_weight_sum (:298)
}

// cdf_2d_row[u] /= row_weight_sum (src/envmap.rs:293-296); 0/0 = NaN for an all-black row, unguarded as in the reference
__global__ void __launch_bounds__(256) k_env_row_normalize(float* __restrict__ cdf_2d, const float* __restrict__ row_sum, uint32_t W, uint32_t H) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= (size_t)W * H) return;
  cdf_2d[i] = cdf_2d[i] / row_sum[i / W];
}

// cdf_1d = sequential prefix of the row sums, then / last (src/envmap.rs:300-308); marginal table (:311-319)
__device__ __forceinline__ uint32_t lower_bound_f32(const float* a, uint32_t lower, uint32_t upper, float value) {
  while (lower < upper) {  // src/envmap.rs:252-265
    const uint32_t mid = (lower + upper) / 2;
    if (a[mid] < value) lower = mid + 1; else upper = mid;
  }
  return lower;
}
__global__ void __launch_bounds__(256) k_env_marginal(const float* __restrict__ row_sum, uint32_t H, float* __restrict__ cdf_1d,
                                                       float* __restrict__ marginal) {
  if (threadIdx.x == 0) {
    float col_weight_sum = 0.0f;
    for (uint32_t v = 0; v < H; ++v) {
      col_weight_sum = col_weight_sum + row_sum[v];
      cdf_1d[v] = col_weight_sum;
    }
  }
  __threadfence_block();
  __syncthreads();
  const float total = cdf_1d[H - 1];
  __syncthreads();
  for (uint32_t v = threadIdx.x; v < H; v += blockDim.x) cdf_1d[v] = cdf_1d[v] / total;
  __threadfence_block();
  __syncthreads();
  const float inv_height = 1.0f / (float)H;
  for (uint32_t v = threadIdx.x; v < H; v += blockDim.x) {
    const uint32_t row = lower_bound_f32(cdf_1d, 0, H, (float)(v + 1) * inv_height);
    marginal[v] = (float)row * inv_height;
  }
}
// conditional table (src/envmap.rs:321-331): one binary search per texel
__global__ void __launch_bounds__(256) k_env_conditional(const float* __restrict__ cdf_2d, uint32_t W, uint32_t H, float* __restrict__ conditional) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= (size_t)W * H) return;
  const uint32_t v = (uint32_t)(i / W), u = (uint32_t)(i - (size_t)v * W);
  const float inv_width = 1.0f / (float)W;
  const float* row = cdf_2d + (size_t)v * W;
  const uint32_t col = lower_bound_f32(row, 0, W, (float)(u + 1) * inv_width);
  conditional[i] = (float)col * inv_width;
}

}  // namespace

std::string envmap_build_distribution(const float4* d_rgba, uint32_t width, uint32_t height, float* d_total_sum, float* d_marginal,
                                      float* d_conditional, hipStream_t s) {
  if (width == 0 || height == 0) return "The environment map is empty!";
  float *cdf_2d = nullptr, *row_sum = nullptr, *cdf_1d = nullptr;
  const size_t n = (size_t)width * height;
  hipError_t e;
  if ((e = hipMalloc(&cdf_2d, n * 4)) != hipSuccess) return std::string("hipMalloc: ") + hipGetErrorString(e);
  if ((e = hipMalloc(&row_sum, (size_t)height * 4)) != hipSuccess) { (void)hipFree(cdf_2d); return std::string("hipMalloc: ") + hipGetErrorString(e); }
  if ((e = hipMalloc(&cdf_1d, (size_t)height * 4)) != hipSuccess) { (void)hipFree(cdf_2d); (void)hipFree(row_sum); return std::string("hipMalloc: ") + hipGetErrorString(e); }
  hipLaunchKernelGGL(k_env_total, dim3(1), dim3(256), 0, s, d_rgba, (unsigned long long)n, d_total_sum);
  hipLaunchKernelGGL(k_env_row_scan, dim3((height + kRowTile - 1) / kRowTile), dim3(256), 0, s, d_rgba, width, height, cdf_2d, row_sum);
  hipLaunchKernelGGL(k_env_row_normalize, dim3((uint32_t)((n + 255) / 256)), dim3(256), 0, s, cdf_2d, row_sum, width, height);
  hipLaunchKernelGGL(k_env_marginal, dim3(1), dim3(256), 0, s, row_sum, height, cdf_1d, d_marginal);
  hipLaunchKernelGGL(k_env_conditional, dim3((uint32_t)((n + 255) / 256)), dim3(256), 0, s, cdf_2d, width, height, d_conditional);
  e = hipStreamSynchronize(s);
  (void)hipFree(cdf_2d); (void)hipFree(row_sum); (void)hipFree(cdf_1d);
  if (e != hipSuccess) return std::string("envmap_build_distribution: ") + hipGetErrorString(e);
  e = hipGetLastError();
  if (e != hipSuccess) return std::string("envmap_build_distribution: ") + hipGetErrorString(e);
  return "";
}

}  // namespace rt
