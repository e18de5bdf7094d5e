// host_scene.h — the renderer's copy of cpu::HalaScene (src/scene/cpu/scene.rs:17-26) and the records
// HalaSceneGPUUploader::upload derives from it (src/scene/loader/gpu_uploader.rs:63-545, :843-885).
#pragma once
#include <cstdint>
#include <string>
#include <vector>

#include "hala_types.h"

namespace rt {

struct Mat4 {
  float m[16];  // column-major (glam::Mat4)
  static Mat4 identity() {
    Mat4 r{};
    r.m[0] = r.m[5] = r.m[10] = r.m[15] = 1.0f;
    return r;
  }
};

struct HostNode {
  int32_t parent = -1;
  Mat4 local = Mat4::identity(), world = Mat4::identity();
  uint32_t mesh_index = HALA_INVALID_INDEX, camera_index = HALA_INVALID_INDEX, light_index = HALA_INVALID_INDEX;
};

struct HostPrimitive {
  std::vector<hala_vertex> vertices;
  std::vector<uint32_t> indices;
  uint32_t material_index = HALA_INVALID_INDEX;
};

// one cpu::HalaImageData as it is kept on the device (level 0 of its mip chain, RENDER_SPEC 7.4): float images as linear RGBA32F texels;
// 8-bit images stay 8-bit (RGBA byte order; B8G8R8A8-tagged data is swizzled here) and are decoded by the sampler
struct HostImage32F {
  uint32_t width = 0, height = 0;
  uint32_t format = 0;         // kTexFloat | kTexSrgb8 | kTexUnorm8 (hala_types.h)
  std::vector<float> rgba;     // format 0
  std::vector<uint8_t> rgba8;  // formats 1, 2
  bool has_alpha = false;  // some texel has alpha < 1: a base-colour map that cuts its surface out (RENDER_SPEC 7.1d)
};
const float* srgb_decode_lut();  // 256 entries: the sRGB EOTF in float (what the *_SRGB sampler of the reference computes)

struct HostScene {
  // cpu::HalaScene
  std::vector<HostImage32F> images;     // image_data, decoded
  std::vector<uint32_t> texture_image;  // textures[i] -> index into images (gpu_uploader.rs:336-338, BTreeMap key order)
  std::vector<HostNode> nodes;
  std::vector<HostPrimitive> prims;       // all primitives of all meshes, mesh-major
  std::vector<uint32_t> mesh_first_prim;  // [mesh_count + 1]
  std::vector<hala_material_desc> materials;
  std::vector<hala_light_desc> lights_cpu;
  std::vector<hala_camera_desc> cameras_cpu;
  // gpu::HalaScene (packed)
  std::vector<hala_gpu_camera> cameras;
  std::vector<hala_gpu_light> lights;
  std::vector<hala_aabb> light_aabbs;
  std::vector<hala_gpu_material> gpu_materials;
  std::vector<hala_gpu_mesh_data> instances;  // `primitives` of gpu_uploader.rs:843-871 (addresses filled at upload)
  std::vector<float> instance_3x4;            // VkAccelerationStructureInstanceKHR transforms (:854-858)
  std::vector<uint32_t> inst_first_tri;       // [instance_count + 1]
  std::vector<uint32_t> instance_node, instance_prim;
  uint32_t triangle_count = 0;

  std::string assign(const hala_scene_desc* d);  // copy + update_node_hierarchies + pack; "" on success
  void update_node_hierarchies();                // src/scene/cpu/scene.rs:99-114
  std::string pack();
  static hala_gpu_material pack_material(const hala_material_desc& m);
  static void primitive_bounds(const HostPrimitive& p, float center[3], float extents[3]);
};

}  // namespace rt
