// ASAN / UBSAN harness for the host-side loaders of libhalart.so (CPU only; scripts/fuzz/run.sh builds and feeds it).
//   harness jpeg <files...>   rt::decode_jpeg on raw bytes
//   harness gltf <files...>   hala_scene_load_gltf (JSON, buffers, PNG / JPEG data URIs)
//   harness image <files...>  rt::load_float_image (OpenEXR scanline / tiled, Radiance .hdr, .pfm: what set_envmap(path) opens)
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <string>
#include <vector>

#include "../../include/halart.h"
#include "host_image.h"
namespace rt { bool decode_jpeg(const std::vector<uint8_t>& raw, uint32_t* w, uint32_t* h, std::vector<uint8_t>* rgba); }

int main(int argc, char** argv) {
  if (argc < 3) return 2;
  int ok = 0, bad = 0;
  for (int i = 2; i < argc; ++i) {
    if (!strcmp(argv[1], "jpeg")) {
      FILE* f = fopen(argv[i], "rb");
      if (!f) continue;
      std::vector<uint8_t> raw; uint8_t buf[4096]; size_t n;
      while ((n = fread(buf, 1, sizeof buf, f)) > 0) raw.insert(raw.end(), buf, buf + n);
      fclose(f);
      uint32_t w = 0, h = 0; std::vector<uint8_t> out;
      if (rt::decode_jpeg(raw, &w, &h, &out)) ++ok; else ++bad;
    } else if (!strcmp(argv[1], "image")) {
      rt::HostImage img;
      if (rt::load_float_image(argv[i], &img).empty()) ++ok; else ++bad;
    } else {
      hala_scene* s = nullptr;
      if (hala_scene_load_gltf(argv[i], &s) == 0 && s) { ++ok; hala_scene_free(s); } else ++bad;
    }
  }
  printf("%s: accepted %d refused %d\n", argv[1], ok, bad);
  return 0;
}
