/* A host in plain C99 over include/halart.h: the call sequence the reference's application runs against HalaRenderer
 * (src/rt_renderer.rs: new -> [set_envmap] -> set_scene -> commit -> { update -> render }* -> save_images), with the scene read by the
 * library's own glTF loader (what cpu::HalaScene::new(path) is in the reference).  No Python, no torch, no C++ on this side.
 *
 *   cc -std=c99 -Iinclude examples/render_gltf.c -Lhala-renderer_amd/lib -lhalart -Wl,-rpath,$PWD/hala-renderer_amd/lib -o render_gltf
 *   ./render_gltf scene.gltf out/frame 640 360 16 [env.hdr|env.exr|env.pfm [rotation_degrees [two-level]]]
 *
 * A last argument "two-level" asks for the reference's BLAS / TLAS split (hala_rt_set_build_options: instancing = 2 — primitives that
 * several nodes reference are stored once) instead of one tree over all triangles flattened to world space.
 *
 * writes out/frame_color.pfm, out/frame_albedo.pfm, out/frame_normal.pfm (the reference's save_images trio). */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "halart.h"

static int fail(const char* what) {
  fprintf(stderr, "%s: %s\n", what, hala_last_error_message());
  return 1;
}

int main(int argc, char** argv) {
  if (argc < 6) {
    fprintf(stderr, "usage: %s scene.gltf out_stem width height spp [envmap [rotation_degrees [two-level]]]\n", argv[0]);
    return 2;
  }
  const char* gltf = argv[1];
  const char* stem = argv[2];
  const uint32_t width = (uint32_t)atoi(argv[3]), height = (uint32_t)atoi(argv[4]), spp = (uint32_t)atoi(argv[5]);

  hala_scene* scene = NULL;
  if (hala_scene_load_gltf(gltf, &scene) != 0) return fail("hala_scene_load_gltf");

  hala_rt_renderer* r = NULL;
  if (hala_rt_create("render_gltf", width, height, /*device*/ 0, /*max_depth*/ 8, /*rr_depth*/ 3, /*tonemap*/ 0, /*aces*/ 0,
                     /*simple aces*/ 0, /*max_frames*/ 0, &r) != 0)
    return fail("hala_rt_create");
  if (argc > 6 && hala_rt_set_envmap_file(r, argv[6], argc > 7 ? (float)atof(argv[7]) : 0.0f) != 0) return fail("hala_rt_set_envmap_file");
  if (hala_rt_set_scene(r, hala_scene_get_desc(scene)) != 0) return fail("hala_rt_set_scene");
  hala_scene_free(scene); /* borrowed for the call only: the renderer has copied what it needs */
  if (argc > 8 && strcmp(argv[8], "two-level") == 0) {
    hala_rt_build_options options;
    memset(&options, 0, sizeof options); /* every field 0 = the default */
    options.instancing = 2;
    if (hala_rt_set_build_options(r, &options) != 0) return fail("hala_rt_set_build_options");
  }
  if (hala_rt_commit(r) != 0) return fail("hala_rt_commit");
  {
    hala_bvh_info info;
    if (hala_rt_get_bvh_info(r, &info) != 0) return fail("hala_rt_get_bvh_info");
    printf("triangles %u stored %u instance references %u tree bytes %llu\n", info.triangle_count, info.stored_triangle_count, info.instance_ref_count,
           (unsigned long long)info.tree_bytes);
  }

  for (uint32_t k = 0; k < spp; ++k) { /* one sample per pixel per update, like the reference's frame loop */
    if (hala_rt_update(r, 0.0, width, height) != 0) return fail("hala_rt_update");
    if (hala_rt_render(r) != 0) return fail("hala_rt_render");
  }
  if (hala_rt_save_images(r, stem) != 0) return fail("hala_rt_save_images");

  hala_rt_statistics st;
  if (hala_rt_get_statistics(r, &st) != 0) return fail("hala_rt_get_statistics");
  printf("frames %llu rays %llu gpu_ms_total %.3f\n", (unsigned long long)st.total_frames, (unsigned long long)st.rays_total, st.gpu_ms_total);
  hala_rt_destroy(r);
  return 0;
}
