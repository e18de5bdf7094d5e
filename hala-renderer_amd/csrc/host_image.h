// host_image.h — host image helpers (tonemap, PFM writer, .hdr/.pfm decoders); see host_util.cpp.
#pragma once
#include <cstddef>
#include <cstdint>
#include <string>
#include <vector>

namespace rt {

struct HostImage {
  uint32_t width = 0, height = 0, channels = 0;  // row 0 = top
  std::vector<float> pixels;
};

void tonemap_pixels(float* rgba, size_t count, int enable_tonemap, int enable_aces, int use_simple_aces);
std::string write_pfm(const char* path, const float* rgba, uint32_t width, uint32_t height);
std::string load_float_image(const char* path, HostImage* img);

}  // namespace rt
