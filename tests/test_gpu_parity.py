"""GPU tier (-m gpu): the HIP path of libhalart.so, called through its C ABI, against the CPU oracle on the same seeded
inputs.  Bar: bit-exact for everything integer/index valued and for everything the rendering spec pins to single IEEE
operations (which is all of it: docs/RENDER_SPEC.md §2); the only tolerance used is the libm tolerance of cos() in the
light records (the reference calls f32::cos there, gpu_uploader.rs:189,206).
"""
import ctypes as C
import os

import contextlib

import numpy as np
import pytest

import hala_renderer_amd as H
from conftest import GOLDEN
from hala_renderer_amd import scenes
from test_oracle_host import kat_images, light_scene
from test_oracle_render import furnace_scene, random_rays

pytestmark = pytest.mark.gpu
f32 = np.float32


def make_renderer(halart, scene, w, h, max_depth=5, rr_depth=3, tonemap=(False, False, False), max_frames=0, env=None, env_rot=0.0, build=None):
    r = halart.HalaRenderer("test", w, h, max_depth, rr_depth, *tonemap, max_frames)
    if build is not None:
        r.set_build_options(**build)
    if env is not None:
        r.set_envmap(env, env_rot)
    r.set_scene(scene)
    r.commit()
    return r


@contextlib.contextmanager
def two_level_trees(oracle):
    """RENDER_SPEC 4.5 on both sides: oracle scenes created inside intersect instanced primitives in object space; pass the yielded build
    options to make_renderer (hala_rt_build_options::instancing = 2).  The default everywhere else: every instance flattened."""
    oracle.set_instancing(True)
    try:
        yield dict(instancing=True)
    finally:
        oracle.set_instancing(False)


def validate_tree(oracle, osc, r):
    """structural check of the renderer's tree against the oracle's scene: (code, levels, two_level).  One-level trees: every child box
    contains what hangs below it, the triangles are the scene's bit for bit; two-level trees (RENDER_SPEC 4.5: scenes in which several
    instances reference one primitive): the same per tree, in its own space, plus the instance levels and references"""
    nodes, tris = r.download_bvh()
    refs = r.download_instance_refs()
    if len(refs):
        rc, depth = oracle.validate_bvh_two_level(osc, nodes, tris, refs)
        return rc, depth, True
    rc, depth = oracle.validate_bvh(nodes, tris, osc.triangles())
    return rc, depth, False


def struct_bytes(s):
    return bytes(memoryview(s))


# ---- A1: env tables ---------------------------------------------------------------------------------------------
@pytest.mark.parametrize("name", list(kat_images().keys()))
def test_envmap_tables_bit_exact_small(halart, oracle, name):
    img = kat_images()[name]
    h, w, _ = img.shape
    total = C.c_float(); m = np.empty(h, f32); c = np.empty((h, w), f32)
    halart.check(halart.load_library().hala_envmap_build_distribution(0, img.ctypes.data_as(C.POINTER(C.c_float)), w, h, C.byref(total),
                                                                       m.ctypes.data_as(C.POINTER(C.c_float)), c.ctypes.data_as(C.POINTER(C.c_float))))
    t0, m0, c0 = oracle.envmap_build_distribution(img)
    assert f32(total.value).tobytes() == t0.tobytes() and m.tobytes() == m0.tobytes() and c.tobytes() == c0.tobytes()
    g = np.load(os.path.join(GOLDEN, "envmap_kat.npz"))
    assert c.tobytes() == g[f"{name}_conditional"].tobytes()


@pytest.mark.parametrize("w,h", [(2048, 1024), (300, 77), (64, 1), (1, 64)])
def test_envmap_tables_bit_exact_full_size(halart, oracle, w, h):
    """config-3 size (2048x1024 = SURVEY §8d) and ragged shapes; sun disc 1e4x brighter exercises the table tails"""
    img = scenes.sky_sun_envmap(w, h) if h > 1 and w > 1 else (np.random.RandomState(2).rand(h, w, 4).astype(f32) + f32(0.01))
    total = C.c_float(); m = np.empty(h, f32); c = np.empty((h, w), f32)
    halart.check(halart.load_library().hala_envmap_build_distribution(0, img.ctypes.data_as(C.POINTER(C.c_float)), w, h, C.byref(total),
                                                                       m.ctypes.data_as(C.POINTER(C.c_float)), c.ctypes.data_as(C.POINTER(C.c_float))))
    t0, m0, c0 = oracle.envmap_build_distribution(img)
    assert f32(total.value).tobytes() == t0.tobytes()
    assert m.tobytes() == m0.tobytes()
    assert c.tobytes() == c0.tobytes()
    # size-independent properties: tables are monotone and lie in [0, 1]
    assert np.all(np.diff(c, axis=1) >= 0) and np.all(np.diff(m) >= 0) and c.min() >= 0 and c.max() <= 1


def test_envmap_rejects_nan_and_inf(halart):
    r = halart.HalaRenderer("env", 8, 8, 2, 1, False, False, False, 0)
    img = np.ones((4, 8, 3), f32)
    img[2, 3, 1] = np.nan
    with pytest.raises(halart.HalaRendererError, match="The pixel value is NaN!"):  # src/envmap.rs:64-66
        r.set_envmap(img)
    img[2, 3, 1] = np.inf
    with pytest.raises(halart.HalaRendererError, match="The pixel value is infinite!"):  # :67-69
        r.set_envmap(img)
    r.close()


# ---- upload(): packed records -------------------------------------------------------------------------------------
def test_packed_records_match_oracle(halart, oracle):
    s = light_scene()
    prim = scenes.cornell_box().meshes[1].primitives[0]
    s.materials = [H.HalaMaterial(type=0, roughness=r_, anisotropic=a_) for r_ in (0.0, 0.5, 1.0) for a_ in (0.0, 0.5, 1.0)]
    prim.material_index = 4
    s.meshes = [H.HalaMesh([prim])]
    s.nodes.append(H.HalaNode(name="m", parent=0, mesh_index=0))
    r = halart.HalaRenderer("pack", 16, 16, 2, 1, False, False, False, 0)
    r.set_scene(s)
    cams, ocams = r.packed_cameras(), oracle.pack_cameras(s)
    assert [struct_bytes(a) for a in cams] == [struct_bytes(b) for b in ocams]
    (lights, boxes), (olights, oboxes) = r.packed_lights(), oracle.pack_lights(s)
    assert len(lights) == len(olights) == 6
    for a, b in zip(lights, olights):
        for fld in ("intensity", "position", "u"):
            assert list(getattr(a, fld)) == list(getattr(b, fld))
        assert np.allclose(list(a.v), list(b.v), rtol=0, atol=1.2e-7)  # cos(): libm vs libm, <= 1 ulp
        assert (f32(a.radius), f32(a.area), a.type) == (f32(b.radius), f32(b.area), b.type)
    assert [struct_bytes(a) for a in boxes] == [struct_bytes(b) for b in oboxes]
    mats = r.packed_materials()
    assert [struct_bytes(a) for a in mats] == [struct_bytes(oracle.pack_material(m)) for m in s.materials]
    prims, t3x4 = r.packed_primitives()
    ot, omd = oracle.pack_instances(s)
    assert np.array_equal(t3x4, ot)
    for a, b in zip(prims, omd):
        assert list(a.transform) == list(b.transform) and a.material_index == b.material_index
        assert a.vertices != 0 and a.indices != 0 and a.vertices % 4 == 0 and a.indices % 16 == 0  # device addresses (:869-870)
    r.close()


def test_error_behaviour(halart):
    r = halart.HalaRenderer("err", 16, 16, 2, 1, False, False, False, 0)
    with pytest.raises(halart.HalaRendererError, match="The scene in GPU is none!"):  # src/rt_renderer.rs:138
        r.commit()
    with pytest.raises(halart.HalaRendererError):
        r.update()
    s = scenes.cornell_box()
    s.nodes[3].camera_index = H._abi.INVALID_INDEX
    with pytest.raises(halart.HalaRendererError, match="The camera node of the camera 0 is not found."):  # gpu_uploader.rs:113
        r.set_scene(s)
    s = scenes.cornell_box()
    s.materials[0].type = 7
    with pytest.raises(halart.HalaRendererError, match="Invalid material type."):
        r.set_scene(s)
    with pytest.raises(halart.HalaRendererError):
        halart.HalaRenderer("zero", 0, 16, 2, 1, False, False, False, 0)
    bad = scenes.cornell_box()
    bad.meshes[0].primitives[0].vertices["position"][1, 2] = np.nan  # would poison the scene bounds of the BVH build
    with pytest.raises(halart.HalaRendererError, match="not finite"):
        r.set_scene(bad)
    r.close()


# ---- K1/K3: BVH build, validated through structure and traversal results ---------------------------------------------
SCENES = {
    "cornell": lambda: scenes.cornell_box(),
    "blob_5k": lambda: scenes.bunny_class(subdivisions=4),
    "sponza_60k": lambda: scenes.sponza_class(target_triangles=60000, disney=False),
}


@pytest.mark.parametrize("name", list(SCENES.keys()) + ["sponza_60k:two_level"])
def test_bvh_structure_and_flattening(halart, oracle, name):
    want_two_level = name.endswith(":two_level")
    name = name.split(":")[0]
    s = SCENES[name]()
    with (two_level_trees(oracle) if want_two_level else contextlib.nullcontext()) as build:
        r = make_renderer(halart, s, 16, 16, build=build)
        osc = oracle.OracleScene(s)
    info = r.bvh_info()
    assert info.triangle_count == osc.triangle_count == s.triangle_count()
    omn, omx = osc.bounds()
    assert list(info.scene_min) == list(omn) and list(info.scene_max) == list(omx)
    assert info.node_width == 4  # compressed 4-wide nodes are the default format (RENDER_SPEC §4.1b)
    rc, depth, two_level = validate_tree(oracle, osc, r)  # also checks v0/e1/e2 bit-exact vs RENDER_SPEC §3 / 4.5
    assert rc == 0, f"validate_bvh code {rc}"
    assert two_level == want_two_level  # (the atrium's columns and arches are instanced primitives)
    if two_level:
        assert depth <= info.max_depth and info.instance_node_count > 0 and info.instance_ref_count == 42
        assert info.stored_triangle_count < 0.6 * info.triangle_count  # every instanced primitive is stored once
    else:
        assert depth == info.max_depth and info.stored_triangle_count == info.triangle_count and info.instance_node_count == 0
    r.close()


@pytest.mark.parametrize("name", list(SCENES.keys()))
def test_trace_rays_closest_and_any_bit_exact(halart, oracle, name):
    s = SCENES[name]()
    r = make_renderer(halart, s, 16, 16)
    osc = oracle.OracleScene(s)
    mn, mx = osc.bounds()
    pad = (mx - mn) * 0.25
    rays = np.concatenate([random_rays(60000, mn - pad, mx + pad, 11), osc.camera_rays(160, 90, 0)])
    got = r.trace_rays_host(rays, 0)
    want = osc.trace(rays, 0)  # oracle's own BVH: results must not depend on the BVH
    assert np.array_equal(got["prim"], want["prim"])
    assert got["t"].tobytes() == want["t"].tobytes() and got["u"].tobytes() == want["u"].tobytes() and got["v"].tobytes() == want["v"].tobytes()
    got_any = r.trace_rays_host(rays, 1)
    want_any = osc.trace(rays, 1)
    assert np.array_equal(got_any["t"], want_any["t"])
    # brute force on a sample (BVH-free truth)
    sub = rays[:: max(1, len(rays) // 3000)]
    assert np.array_equal(r.trace_rays_host(sub, 0)["prim"], osc.trace(sub, 0, brute=True)["prim"])
    r.close()


def test_traversal_step_counts_match_oracle_on_same_bvh(halart, oracle):
    """the inputs of roofline.achieved: nodes visited / triangles tested per ray must equal the oracle's count on the
    SAME (GPU-built) BVH for the same rays"""
    s = scenes.sponza_class(target_triangles=60000, disney=False)
    for instancing in (True, False):  # the two-level tree (instance levels + one tree per instanced primitive) and everything flattened
        r = make_renderer(halart, s, 16, 16, build=dict(instancing=instancing))
        oracle.set_instancing(instancing)
        try:
            osc = oracle.OracleScene(s)
        finally:
            oracle.set_instancing(False)
        rays = osc.camera_rays(320, 180, 0)
        nodes, tris = r.download_bvh()
        refs = r.download_instance_refs()
        assert (len(refs) > 0) == instancing
        for mode in (0, 1):
            hits, cnt = r.trace_rays_host(rays, mode, count_steps=True)
            ohits, ocnt = oracle.trace_on_bvh(nodes, tris, rays, mode, refs)
            assert cnt == ocnt, (instancing, mode)
            assert np.array_equal(hits["t"], ohits["t"])
            assert hits.tobytes() == osc.trace(rays, mode).tobytes()  # and the oracle's own tree, same instancing rule
        r.close()


def test_deep_stack_spills_to_global_scratch(halart, oracle):
    """overlapping sheets: rays hit every child of every node, the traversal stack outgrows its LDS entries; results must
    still equal brute force, and the step counts the oracle's on the same BVH"""
    s = scenes.stacked_sheets(count=4096)
    r = make_renderer(halart, s, 16, 16)
    osc = oracle.OracleScene(s)
    info = r.bvh_info()
    assert 3 * info.max_depth > 12  # more than the LDS part of the stack can hold (traverse.h kStackLds)
    rays = np.concatenate([osc.camera_rays(48, 48, 0), random_rays(2000, np.array([-1.5, -1.5, -2], dtype=f32), np.array([1.5, 1.5, 2], dtype=f32), 5)])
    got = r.trace_rays_host(rays, 0)
    want = osc.trace(rays, 0, brute=True)
    assert got.tobytes() == want.tobytes()
    nodes, tris = r.download_bvh()
    for mode in (1, 0):
        hits, cnt = r.trace_rays_host(rays, mode, count_steps=True)
        ohits, ocnt = oracle.trace_on_bvh(nodes, tris, rays, mode)
        assert cnt == ocnt and np.array_equal(hits["t"], ohits["t"])
    assert cnt[0] > 20 * len(rays)  # closest-hit rays visit dozens of nodes each here
    r.close()


@pytest.mark.parametrize("ntri", [1, 2, 3, 5])
def test_tiny_scenes_build_and_trace(halart, oracle, ntri):
    """fewer triangles than one leaf / one node holds: single-leaf root, one real node, and the first scene with two"""
    s = scenes.stacked_sheets(count=3)
    prim = s.meshes[0].primitives[0]
    prim.indices = prim.indices[: 3 * ntri].copy()
    r = make_renderer(halart, s, 16, 16)
    osc = oracle.OracleScene(s)
    info = r.bvh_info()
    assert info.triangle_count == ntri and info.node_count >= 1
    nodes, tris = r.download_bvh()
    assert oracle.validate_bvh(nodes, tris, osc.triangles())[0] == 0
    rays = np.concatenate([osc.camera_rays(32, 32, 0), random_rays(500, np.array([-1.5, -1.5, -2], dtype=f32), np.array([1.5, 1.5, 2], dtype=f32), 9)])
    for mode in (0, 1):
        hits, cnt = r.trace_rays_host(rays, mode, count_steps=True)
        assert hits.tobytes() == osc.trace(rays, mode, brute=True).tobytes()
        assert cnt == oracle.trace_on_bvh(nodes, tris, rays, mode)[1]
    r.update(); r.render()
    img, _ = osc.render(16, 16, frames=1)
    assert r.read_image(0).tobytes() == img[0].tobytes()
    r.close()


def test_empty_and_ragged_batches(halart, oracle):
    s = scenes.cornell_box()
    r = make_renderer(halart, s, 16, 16)
    osc = oracle.OracleScene(s)
    assert len(r.trace_rays_host(np.zeros(0, dtype=H._abi.RAY_DTYPE), 0)) == 0
    mn, mx = osc.bounds()
    for n in (1, 63, 64, 65, 1000):
        rays = random_rays(n, mn, mx, n)
        assert r.trace_rays_host(rays, 0).tobytes() == osc.trace(rays, 0).tobytes()
    # degenerate rays: zero direction components, tmax = 0, rays starting on geometry
    rays = np.zeros(4, dtype=H._abi.RAY_DTYPE)
    rays["origin"] = [(278, 273, -800), (278, 0, 279), (100, 100, 100), (278, 273, 279)]
    rays["direction"] = [(0, 0, 1), (0, 1, 0), (1, 0, 0), (0, -1, 0)]
    rays["tmax"] = [3e38, 3e38, 0.0, 3e38]
    assert r.trace_rays_host(rays, 0).tobytes() == osc.trace(rays, 0).tobytes()
    r.close()


@pytest.mark.parametrize("scene_name", ["cornell", "blob"])
def test_rays_with_odd_limits(halart, oracle, scene_name):
    """tmax = NaN / negative / +inf / 0 and negative tmin on a caller's batch, closest and any hit, on an LDS-staged tree and on a large one:
    a NaN or negative limit admits no hit (the kernels' leaf bookkeeping compares the BITS of the limit: trav_begin turns NaN into -1), +inf
    is an ordinary limit, a negative tmin starts at the origin (RENDER_SPEC 4.2)"""
    s = scenes.cornell_box() if scene_name == "cornell" else scenes.bunny_class(subdivisions=4)
    r = make_renderer(halart, s, 16, 16)
    osc = oracle.OracleScene(s)
    mn, mx = osc.bounds()
    rays = random_rays(512, mn, mx, 77)
    limits = np.array([np.nan, -1.0, np.inf, 0.0, -0.0, 1e-30, 3e38, -np.inf], dtype=np.float32)
    rays["tmax"] = limits[np.arange(512) % len(limits)]
    rays["tmin"] = np.where(np.arange(512) % 3 == 0, np.float32(-5.0), np.float32(0.0))
    for mode in (0, 1):
        got, want = r.trace_rays_host(rays, mode), osc.trace(rays, mode)
        assert got.tobytes() == want.tobytes()
        dead = ~(rays["tmax"] > 0)  # NaN, negatives, zeros
        assert np.all(got["prim"][dead] == 0xFFFFFFFF) and np.all(got["t"][dead] == -1.0)
        if mode == 0:
            assert np.any(got["prim"][~dead] != 0xFFFFFFFF)
    r.close()


def test_rtprog_trace_rays_and_indirect(halart, oracle):
    """HalaRayTracingProgram mirror (src/raytracing_program.rs:330-340) over device buffers"""
    import torch
    s = scenes.cornell_box()
    r = make_renderer(halart, s, 16, 16)
    osc = oracle.OracleScene(s)
    rays = osc.camera_rays(64, 32, 0)
    d_rays = torch.from_numpy(rays.view(np.uint8).copy()).cuda()
    d_hits = torch.zeros(len(rays) * 16, dtype=torch.uint8, device="cuda")
    torch.cuda.synchronize()
    desc = halart.HalaRayTracingProgramDesc(raygen_shader_file_paths=["builtin"], hit_shader_file_paths=[halart.HalaRayTracingHitShaderDesc("builtin")], push_constant_size=4)
    prog = halart.HalaRayTracingProgram(r, desc, "t")
    with pytest.raises(halart.HalaRendererError, match="not bound"):
        prog.trace_rays(64, 32, 1)
    with pytest.raises(halart.HalaRendererError, match="exceeds push_constant_size"):
        prog.push_constants(2, b"\0\0\0\0")
    with pytest.raises(halart.HalaRendererError, match="raygen shader list is empty"):
        halart.HalaRayTracingProgram(r, '{"raygen_shader_file_paths": [], "hit_shader_file_paths": []}', "e")
    with pytest.raises(halart.HalaRendererError, match="missing field"):
        halart.HalaRayTracingProgram(r, '{"hit_shader_file_paths": []}', "e")
    prog.bind(d_rays.data_ptr(), d_hits.data_ptr())
    prog.trace_rays(64, 32, 1)
    r.wait_idle()
    want = osc.trace(rays, 0)
    assert d_hits.cpu().numpy().tobytes() == want.tobytes()
    prog.push_constants(0, (1).to_bytes(4, "little"))  # any-hit
    cmd = torch.tensor([64, 16, 2], dtype=torch.int32, device="cuda")
    torch.cuda.synchronize()
    prog.trace_rays_indirect(cmd.data_ptr())
    r.wait_idle()
    got = np.frombuffer(d_hits.cpu().numpy().tobytes(), dtype=H._abi.HIT_DTYPE)
    assert np.array_equal(got["t"], osc.trace(rays, 1)["t"])
    prog.push_constants_f32(0, [0.0])  # bits 0: closest hit again
    prog.trace_rays(64, 32, 1)
    r.wait_idle()
    assert d_hits.cpu().numpy().tobytes() == want.tobytes()
    prog.close()
    r.close()


def test_launch_timing_period_and_pass_fusion(halart):
    """hala_rt_set_launch_timing_period: per-launch events on every n-th update only (default: none); ray totals and frame times are
    always collected, the *_timed ray counters follow the timed updates.  hala_rt_set_pass_fusion: 0 one launch per pass, 1 (default)
    fused launches except in timed updates, 2 always — timed updates then fill traverse_fused_* (the closest-hit pass of depth d + 1
    runs inside depth d's fused launch; the Cornell box has light connections only, so the last depth keeps its plain shadow launch).
    The image depends on neither."""
    s = scenes.cornell_box()
    imgs = []
    for period, fusion, timed_updates in ((1, 1, 6), (3, 1, 2), (0, 1, 0), (None, 1, 0), (1, 2, 6), (2, 2, 3), (1, 0, 6), (0, 0, 0)):
        r = make_renderer(halart, s, 64, 48)
        if period is not None:
            r.set_launch_timing_period(period)
        r.set_pass_fusion(fusion)
        for _ in range(6):
            r.update(); r.render()
        st = r.statistics()
        imgs.append(r.read_image(0))
        assert st.total_frames == 6 and st.rays_total == st.rays_closest_total + st.rays_shadow_total and st.gpu_ms_total > 0.0
        assert st.traverse_primary_launches == timed_updates and st.shade_launches == timed_updates * 5  # max_depth 5
        assert st.rays_primary_timed == timed_updates * 64 * 48
        if fusion == 2:  # timed AND fused: depth 0 has its own closest-hit launch, depths 0..3 end in a fused launch, depth 4 in a shadow launch
            assert st.traverse_closest_launches == timed_updates and st.traverse_fused_launches == timed_updates * 4
            assert st.traverse_shadow_launches == timed_updates
            assert st.rays_closest_timed == st.rays_primary_timed
            if period == 1:
                assert st.rays_closest_timed + st.rays_fused_closest_timed == st.rays_closest_total
                assert st.rays_shadow_timed + st.rays_fused_shadow_timed == st.rays_shadow_total
                assert st.traverse_fused_ms_total > 0.0 and st.rays_fused_shadow_timed > st.rays_shadow_timed > 0
        else:
            assert st.traverse_closest_launches == timed_updates * 5 and st.traverse_shadow_launches == timed_updates * 5
            assert st.traverse_fused_launches == 0 and st.traverse_fused_ms_total == 0.0 and st.rays_fused_closest_timed == 0
            if period == 1:
                assert (st.rays_closest_timed, st.rays_shadow_timed) == (st.rays_closest_total, st.rays_shadow_total)
                assert st.traverse_closest_ms_total > 0.0
        if not timed_updates:
            assert st.traverse_closest_ms_total == 0.0 and st.rays_closest_timed == 0
        r.close()
    assert all(i.tobytes() == imgs[0].tobytes() for i in imgs[1:])


def test_load_blue_noise_texture_from_file(halart, tmp_path):
    """load_blue_noise_texture(path) (src/rt_renderer.rs:1117-1156): PNG decoded by the library itself; the reference's messages for a
    missing file / an empty name; the texture is accepted and unused (RENDER_SPEC 2.3), so the render does not change"""
    from PIL import Image
    rng = np.random.RandomState(5)
    Image.fromarray(rng.randint(0, 256, (32, 32, 4), dtype=np.uint8), "RGBA").save(tmp_path / "blue.png")
    s = scenes.cornell_box()
    r = make_renderer(halart, s, 24, 24)
    r.update(); r.render()
    before = r.read_image(0)
    r.load_blue_noise_texture(str(tmp_path / "blue.png"))
    r.reset_accumulation(); r.update(); r.render()
    assert r.read_image(0).tobytes() == before.tobytes()
    with pytest.raises(Exception, match="Failed to open image"):
        r.load_blue_noise_texture(str(tmp_path / "missing.png"))
    with pytest.raises(Exception, match="The file name is none!"):
        r.load_blue_noise_texture("")
    r.close()


# ---- K4: refit -------------------------------------------------------------------------------------------------------
def test_refit_after_node_transform(halart, oracle):
    s = scenes.cornell_box()
    r = make_renderer(halart, s, 16, 16)
    m = np.eye(4, dtype=f32)
    m[:3, 3] = (60.0, 25.0, -40.0)
    c, sn = np.cos(0.4), np.sin(0.4)
    m[:3, :3] = np.array([[c, 0, sn], [0, 1, 0], [-sn, 0, c]], dtype=f32)
    r.update_node_transform(2, m)  # move the tall block
    r.refit()
    s.nodes[2].local_transform = m
    osc = oracle.OracleScene(s)
    nodes, tris = r.download_bvh()
    rc, _ = oracle.validate_bvh(nodes, tris, osc.triangles())
    assert rc == 0
    rays = osc.camera_rays(128, 128, 0)
    assert r.trace_rays_host(rays, 0).tobytes() == osc.trace(rays, 0).tobytes()
    r.update(); r.render()
    img, _ = osc.render(16, 16, frames=1)
    assert r.read_image(0).tobytes() == img[0].tobytes()
    r.close()


def test_refit_with_only_the_camera_moved(halart, oracle):
    """the interactive case: update_node_transform on the camera node + refit re-publishes the camera record and restarts the
    accumulation; the geometry did not move, so the tree is left alone (same bytes) and the new view matches the oracle"""
    s = scenes.cornell_box()
    cam = next(k for k, n in enumerate(s.nodes) if n.camera_index != 0xFFFFFFFF)
    r = make_renderer(halart, s, 40, 30)
    r.update(); r.render()
    n0, t0 = r.download_bvh()
    m = scenes.look_at_node_transform((150.0, 320.0, -650.0), (278.0, 250.0, 280.0))
    r.update_node_transform(cam, m)
    r.refit()
    n1, t1 = r.download_bvh()
    assert n0.tobytes() == n1.tobytes() and t0.tobytes() == t1.tobytes()
    r.update(); r.update(); r.render()
    s.nodes[cam].local_transform = m
    img, _ = oracle.OracleScene(s).render(40, 30, frames=2)
    assert r.read_image(0).tobytes() == img[0].tobytes()
    assert r.statistics().total_frames == 2
    r.close()


def test_refit_after_material_edit(halart, oracle):
    """hala_rt_update_material + refit: the packed material records are re-published (the tree is left alone), the accumulation
    restarts and the frame matches the oracle of the edited scene — diffuse to Disney glass with an absorbing medium"""
    s = scenes.cornell_box()
    r = make_renderer(halart, s, 40, 30, max_depth=8)
    r.update(); r.render()
    n0, t0 = r.download_bvh()
    before = r.read_image(0)
    new = H.HalaMaterial(type=1, base_color=(0.9, 0.95, 1.0), metallic=0.0, roughness=0.08, specular_transmission=1.0, ior=1.5,
                         medium=H.HalaMedium(1, (0.8, 0.9, 1.0), 0.004, 0.0))
    target = s.meshes[s.nodes[2].mesh_index].primitives[0].material_index  # the tall block's material
    with pytest.raises(Exception, match="The material does not exist"):
        r.update_material(len(s.materials), new)
    r.update_material(target, new)
    r.refit()
    n1, t1 = r.download_bvh()
    assert n0.tobytes() == n1.tobytes() and t0.tobytes() == t1.tobytes()
    r.update(); r.update(); r.render()
    s.materials[target] = new
    img, _ = oracle.OracleScene(s).render(40, 30, frames=2, max_depth=8)
    assert r.read_image(0).tobytes() == img[0].tobytes()
    assert np.abs(r.read_image(0) - before).max() > 0.05
    r.close()


@pytest.mark.parametrize("mesh,instancing", [(3, None), (3, True), (0, True)])
def test_refit_after_vertex_deformation(halart, oracle, request, mesh, instancing):
    """hala_rt_update_vertices + refit: a mesh of the atrium is rippled in place — the first drape (mesh 3: referenced
    once) or the first column mesh (mesh 0: 14 instances; in a two-level tree its ONE object-space tree is refitted and all instances
    follow); the refitted tree must bound the new triangles, and rays / a small render must match the oracle built from the deformed scene"""
    s = scenes.sponza_class(target_triangles=20_000)
    oracle.set_instancing(bool(instancing))
    request.addfinalizer(lambda: oracle.set_instancing(False))
    r = make_renderer(halart, s, 24, 16, build=dict(instancing=instancing))
    assert (r.bvh_info().instance_ref_count > 0) == bool(instancing)
    v = s.meshes[mesh].primitives[0].vertices.copy()
    pos = v["position"]
    pos[:, 2] += (0.35 * np.sin(3.0 * pos[:, 0] + 1.7 * pos[:, 1])).astype(f32)
    pos[:, 0] *= f32(1.1)
    with pytest.raises(Exception):
        r.update_vertices(mesh, 0, v[:-1])          # topology is kept: the count must match
    with pytest.raises(Exception):
        r.update_vertices(len(s.meshes), 0, v)
    r.update_vertices(mesh, 0, v)
    r.refit()
    s.meshes[mesh].primitives[0].vertices = v
    osc = oracle.OracleScene(s)
    rc, _, _ = validate_tree(oracle, osc, r)
    assert rc == 0
    rays = osc.camera_rays(160, 96, 0)
    assert r.trace_rays_host(rays, 0).tobytes() == osc.trace(rays, 0).tobytes()
    r.update(); r.update(); r.render()
    img, _ = osc.render(24, 16, frames=2)
    assert r.read_image(0).tobytes() == img[0].tobytes()
    r.close()


def test_ploc_drivers_build_the_same_tree(halart, oracle):
    """PLOC's rounds are driven three ways — every round looked at by the host (ploc_look_every = 1), several rounds between two
    looks with the counters on the device (the default, and an odd value), the last rounds (<= 512 clusters) inside one workgroup
    (k_ploc_tail) or not (ploc_tail = 2): all of them must emit byte-identical nodes and triangle orders (hala_rt_set_build_options)"""
    s = scenes.sponza_class(target_triangles=60_000)
    trees = []
    # builder "ploc": the fast large-scene build (the default at this size is the SAH one)
    for tail, look, collapse_look in ((1, 6, 4), (2, 1, 1), (1, 1, 3), (2, 7, 9)):
        r = make_renderer(halart, s, 16, 16, build=dict(builder="ploc", ploc_tail=tail, ploc_look_every=look, collapse_look_every=collapse_look, instancing=True))
        trees.append(r.download_bvh())
        info = r.bvh_info()
        if len(trees) == 1:
            with two_level_trees(oracle):
                rc, _, two_level = validate_tree(oracle, oracle.OracleScene(s), r)  # (a two-level tree: every tree of it is built by PLOC)
            assert rc == 0 and two_level
        r.close()
    assert info.triangle_count >= 4096
    for nodes, tris in trees[1:]:
        assert nodes.tobytes() == trees[0][0].tobytes() and tris.tobytes() == trees[0][1].tobytes()


def test_inner_children_stand_in_slots_by_descending_box_area(halart):
    """RENDER_SPEC 4.4c, build side: an any-hit ray takes the inner children of a node in slot order, and the builders fill the slots by
    descending surface area of the children's boxes (the larger box first: 9.09 instead of 9.40 node visits per connection on configs[3]).
    Read back from the 64-B nodes: pmin | exponents | near / far bytes per axis | four references (RENDER_SPEC 4.1b).  The builder orders
    by the exact boxes, the node holds them quantised to 8 bits: a pair may swap when its areas are within the quantisation error."""
    for builder in ("sah", "ploc"):
        r = make_renderer(halart, scenes.bunny_class(subdivisions=5), 16, 16, build=dict(builder=builder))
        nodes, _ = r.download_bvh()
        r.close()
        n = nodes.reshape(-1, 16)
        exps = n[:, 3]
        scale = np.stack([np.ldexp(1.0, ((exps >> (8 * a)) & 0xff).astype(np.int64) - 127) for a in range(3)], 1)  # [N, 3]
        ext = np.zeros((n.shape[0], 4, 3))
        for a in range(3):
            lo, hi = n[:, 4 + a], n[:, 7 + a]
            for c in range(4):
                ext[:, c, a] = (((hi >> (8 * c)) & 0xff).astype(np.int64) - ((lo >> (8 * c)) & 0xff).astype(np.int64)) * scale[:, a]
        area = ext[..., 0] * ext[..., 1] + ext[..., 1] * ext[..., 2] + ext[..., 2] * ext[..., 0]
        refs = n[:, 12:16]
        inner = (refs >> 31) == 0
        pairs = swapped = 0
        for k in range(3):
            both = inner[:, k] & inner[:, k + 1]
            pairs += int(both.sum())
            swapped += int((both & (area[:, k + 1] > area[:, k] * 1.25)).sum())
        assert pairs > 100 and swapped <= pairs // 100, (builder, pairs, swapped)


def test_builders_differ_in_trees_not_in_results(halart, oracle):
    """the three hierarchy builders (full-sweep SAH: the default from 4096 triangles; PLOC; LBVH) over one scene: every tree passes the
    structural check, is rebuilt byte for byte, gives the oracle's hits (the oracle traverses its OWN tree) and the oracle's step counts
    on that very tree; the SAH tree is the one with the fewest node visits per ray"""
    s = scenes.sponza_class(target_triangles=60_000, disney=False)
    osc = oracle.OracleScene(s)  # one tree over all 60 k triangles (every instance flattened): the builders are compared on it
    mn, mx = osc.bounds()
    rays = np.concatenate([random_rays(30000, mn, mx, 23), osc.camera_rays(160, 90, 0)])
    want = osc.trace(rays, 0)
    want_any = osc.trace(rays, 1)
    visits = {}
    for builder in ("sah", "ploc", "lbvh", None):
        r = make_renderer(halart, s, 16, 16, build=dict(builder=builder, instancing=False))
        nodes, tris = r.download_bvh()
        rc, depth = oracle.validate_bvh(nodes, tris, osc.triangles())
        assert rc == 0 and depth == r.bvh_info().max_depth, builder
        got, cnt = r.trace_rays_host(rays, 0, count_steps=True)
        assert got.tobytes() == want.tobytes(), builder
        ohits, ocnt = oracle.trace_on_bvh(nodes, tris, rays, 0)
        assert cnt == ocnt, builder
        got_any, cnt_any = r.trace_rays_host(rays, 1, count_steps=True)
        assert np.array_equal(got_any["t"], want_any["t"]), builder
        assert cnt_any == oracle.trace_on_bvh(nodes, tris, rays, 1)[1], builder
        r.close()
        r2 = make_renderer(halart, s, 16, 16, build=dict(builder=builder, instancing=False))  # deterministic: atomics only carry min / max / integer sums
        n2, t2 = r2.download_bvh()
        r2.close()
        assert n2.tobytes() == nodes.tobytes() and t2.tobytes() == tris.tobytes(), builder
        visits[builder] = (cnt[0], nodes.tobytes())
    assert visits[None][1] == visits["sah"][1]  # the default at this size
    assert visits["sah"][0] < min(visits["ploc"][0], visits["lbvh"][0]), {k: v[0] for k, v in visits.items()}


def test_sah_build_of_coincident_triangles(halart, oracle):
    """6000 copies of ONE triangle pair among ordinary geometry: every split of the pile costs the same, the sweep peels it a few
    triangles at a time until the round limit, then halves (bvh_build.hip: kSweepRounds) — the build must end, the tree must be valid
    and a ray into the pile must report the copy with the lowest id (RENDER_SPEC 4.2 tie rule), as brute force does"""
    s = scenes.cornell_box()
    quad = ((200.0, 100.0, 300.0), (350.0, 100.0, 300.0), (350.0, 250.0, 300.0), (200.0, 250.0, 300.0))  # inside the box, facing the camera
    pile = scenes._merge_quads([quad] * 3000); pile.material_index = 0
    s.meshes.append(H.HalaMesh([pile]))
    s.nodes.append(H.HalaNode(name="pile", mesh_index=len(s.meshes) - 1))
    osc = oracle.OracleScene(s)
    r = make_renderer(halart, s, 16, 16)
    info = r.bvh_info()
    assert info.triangle_count == osc.triangle_count >= 6000
    nodes, tris = r.download_bvh()
    rc, depth = oracle.validate_bvh(nodes, tris, osc.triangles())
    assert rc == 0 and depth == info.max_depth
    rays = osc.camera_rays(96, 96, 0)
    got = r.trace_rays_host(rays, 0)
    assert got.tobytes() == osc.trace(rays, 0, brute=True).tobytes()
    hits, cnt = r.trace_rays_host(rays, 0, count_steps=True)
    assert cnt == oracle.trace_on_bvh(nodes, tris, rays, 0)[1]
    imgs, _ = osc.render(48, 48, frames=1)
    r2 = make_renderer(halart, s, 48, 48)
    r2.update()
    assert_images_equal(r2, imgs)
    r.close(); r2.close()


def test_refit_without_change_reproduces_the_build(halart, oracle):
    """refit re-derives the 4-wide nodes level by level (k_refit_level) instead of fit + pack: with nothing moved the nodes, the
    triangle order and every box must come out byte for byte as the build (PLOC boxes for the large scene, k_fit for the small)"""
    for s in (scenes.sponza_class(target_triangles=20_000), scenes.cornell_box()):
        r = make_renderer(halart, s, 16, 16)
        n0, t0 = r.download_bvh()
        info0 = r.bvh_info()
        r.refit()
        n1, t1 = r.download_bvh()
        info1 = r.bvh_info()
        assert n0.tobytes() == n1.tobytes() and t0.tobytes() == t1.tobytes()
        assert (info0.node_count, info0.max_depth) == (info1.node_count, info1.max_depth)
        assert list(info0.scene_min) == list(info1.scene_min) and list(info0.scene_max) == list(info1.scene_max)
        r.close()


# ---- K5-K7: the integrator -----------------------------------------------------------------------------------------------
def assert_images_equal(r, imgs):
    for which, name in ((0, "accum"), (1, "albedo"), (2, "normal"), (3, "final")):
        got = r.read_image(which)
        want = imgs[which]
        if got.tobytes() != want.tobytes():
            bad = np.any(got != want, axis=-1)
            raise AssertionError(f"{name}: {bad.sum()} of {bad.size} pixels differ, max abs diff {np.abs(got - want).max()}")


def test_render_cornell_bit_exact_and_fixture(halart, oracle):
    """config 1 geometry at a test size; also against the frozen oracle render"""
    s = scenes.cornell_box()
    r = make_renderer(halart, s, 64, 64)
    for _ in range(2):
        r.update(); r.render()
    imgs, st = oracle.OracleScene(s).render(64, 64, frames=2)
    assert_images_equal(r, imgs)
    g = np.load(os.path.join(GOLDEN, "cornell_64x64_2spp.npz"))
    assert r.read_image(0).tobytes() == g["accum"].tobytes()
    stg = r.statistics()
    assert (stg.rays_closest_total, stg.rays_shadow_total) == (st.rays_closest, st.rays_shadow)
    u = r.global_uniform()  # src/rt_renderer.rs:408-427
    assert (u.frame_index, u.max_depth, u.rr_depth, u.num_of_lights, u.env_type, u.camera_index) == (1, 5, 3, 1, 0, 0)
    assert list(u.resolution) == [64.0, 64.0] and list(u.sky_color) == [f32(x) for x in (0.5, 0.7, 1.0, 1.0)] and list(u.ground_color) == [1.0] * 4
    r.close()


@pytest.mark.parametrize("tonemap", [(True, False, False), (True, True, False), (True, True, True)])
def test_render_tonemap_and_setters(halart, oracle, tonemap):
    s = scenes.cornell_box(aspect=1.5)
    r = make_renderer(halart, s, 96, 64, max_depth=4, rr_depth=1, tonemap=tonemap)
    r.set_exposure_value(1.7)
    r.set_env_intensity(0.8)
    r.set_sky_color((0.2, 0.3, 0.9, 1.0))
    r.set_ground_color((0.4, 0.3, 0.2, 1.0))
    for _ in range(3):
        r.update(); r.render()
    imgs, _ = oracle.OracleScene(s).render(96, 64, frames=3, max_depth=4, rr_depth=1, tonemap=tonemap, exposure=1.7, env_intensity=0.8,
                                           sky=(0.2, 0.3, 0.9, 1.0), ground=(0.4, 0.3, 0.2, 1.0))
    assert_images_equal(r, imgs)
    r.close()


def test_render_all_light_types_bit_exact(halart, oracle):
    s = light_scene()
    cb = scenes.cornell_box()
    s.materials = cb.materials
    s.meshes = cb.meshes
    s.nodes += [H.HalaNode(name="room", mesh_index=0), H.HalaNode(name="b1", mesh_index=1), H.HalaNode(name="b2", mesh_index=2)]
    s.nodes[7] = cb.nodes[3]  # the Cornell camera
    # place the analytic lights inside the room
    s.nodes[0].local_transform = np.eye(4, dtype=f32)
    for idx, pos in ((1, (150, 400, 200)), (2, (0, 0, 0)), (3, (100, 50, 0)), (4, (300, 300, 150)), (5, (0, -120, 80))):
        m = s.nodes[idx].local_transform.copy(); m[:3, 3] = pos; s.nodes[idx].local_transform = m
    for l, k in zip(s.lights, (4e4, 1.0, 2e5, 30.0, 60.0)):
        l.intensity = k
    s.lights[3].params = (120.0, 90.0); s.lights[4].params = (40.0, 0.0)
    r = make_renderer(halart, s, 80, 80, max_depth=4, rr_depth=2)
    for _ in range(3):
        r.update(); r.render()
    imgs, _ = oracle.OracleScene(s).render(80, 80, frames=3, max_depth=4, rr_depth=2)
    assert_images_equal(r, imgs)
    assert imgs[0][..., :3].max() > 0
    r.close()


def test_render_envmap_importance_sampling_bit_exact(halart, oracle):
    """config 3 at a test size: blob + ground, env map with a 1e4x sun, rotation, MIS env + BSDF"""
    env = scenes.sky_sun_envmap(256, 128)
    s = scenes.bunny_class(subdivisions=3, aspect=96 / 54)
    r = make_renderer(halart, s, 96, 54, max_depth=4, rr_depth=2, env=env, env_rot=75.0)
    r.set_env_intensity(1.25)
    for _ in range(3):
        r.update(); r.render()
    imgs, st = oracle.OracleScene(s, envmap=env).render(96, 54, frames=3, max_depth=4, rr_depth=2, env_rotation=75.0, env_intensity=1.25)
    assert_images_equal(r, imgs)
    t, m, c = r.env_distribution(256, 128)
    ot, om, oc = oracle.envmap_build_distribution(env)
    assert f32(t).tobytes() == ot.tobytes() and m.tobytes() == om.tobytes() and c.tobytes() == oc.tobytes()
    u = r.global_uniform()
    assert (u.env_type, u.env_map_width, u.env_map_height) == (1, 256, 128) and f32(u.env_rotation) == f32(f32(75.0) / f32(360.0))
    assert f32(u.env_total_sum) == ot
    r.close()


@pytest.mark.parametrize("glass", ["rough_clear", "smooth_tinted_partial"])
def test_render_disney_transmission_bit_exact(halart, oracle, glass):
    """RENDER_SPEC §7.1c: refraction through VNDF-sampled facets (entering and leaving, total internal reflection, NEE through
    the surface), on the blob over a ground plane under an env map and a quad light"""
    env = scenes.sky_sun_envmap(128, 64, sun_gain=300.0)
    s = scenes.bunny_class(subdivisions=3, aspect=80 / 48, disney=True)
    if glass == "rough_clear":
        s.materials[0] = H.HalaMaterial(type=1, base_color=(1.0, 1.0, 1.0), metallic=0.0, roughness=0.35, specular_transmission=1.0, ior=1.5)
    else:
        s.materials[0] = H.HalaMaterial(type=1, base_color=(0.9, 0.5, 0.4), metallic=0.0, roughness=0.05, specular_transmission=0.6, ior=1.33, clearcoat=0.5)
    s.lights = [H.HalaLight(color=(1.0, 0.9, 0.8), intensity=12.0, light_type=H.HalaLightType.QUAD, params=(1.0, 1.0))]
    s.nodes.append(H.HalaNode(name="light", light_index=0, local_transform=scenes.look_at_node_transform((1.5, 3.0, 1.0), (0.0, 0.0, 0.0))))
    r = make_renderer(halart, s, 80, 48, max_depth=8, rr_depth=3, env=env, env_rot=30.0)
    for _ in range(3):
        r.update(); r.render()
    imgs, st = oracle.OracleScene(s, envmap=env).render(80, 48, frames=3, max_depth=8, rr_depth=3, env_rotation=30.0)
    assert_images_equal(r, imgs)
    stg = r.statistics()
    assert (stg.rays_closest_total, stg.rays_shadow_total) == (st.rays_closest, st.rays_shadow)
    # something is seen THROUGH the blob: with the lobe switched off the centre of the image is a different picture
    s.materials[0].specular_transmission = 0.0
    opaque, _ = oracle.OracleScene(s, envmap=env).render(80, 48, frames=3, max_depth=8, rr_depth=3, env_rotation=30.0)
    assert np.abs(opaque[0][16:32, 30:50, :3] - imgs[0][16:32, 30:50, :3]).mean() > 0.01
    r.close()


def test_render_disney_materials_bit_exact(halart, oracle):
    """RENDER_SPEC §7.1b: GGX/VNDF + clearcoat + sheen paths, on the blob (env map MIS) and on the atrium (24 materials,
    quad lights, instanced meshes under a parent node)"""
    env = scenes.sky_sun_envmap(128, 64, sun_gain=300.0)
    s = scenes.bunny_class(subdivisions=3, aspect=80 / 48, disney=True)
    s.materials[0].clearcoat = 1.0; s.materials[0].clearcoat_roughness = 0.15; s.materials[0].sheen = 0.5
    s.materials[0].anisotropic = 0.6; s.materials[0].specular_tint = 0.4
    s.materials[1] = H.HalaMaterial(type=1, base_color=(0.9, 0.85, 0.7), metallic=1.0, roughness=0.3)
    r = make_renderer(halart, s, 80, 48, max_depth=5, rr_depth=2, env=env, env_rot=200.0)
    for _ in range(3):
        r.update(); r.render()
    imgs, _ = oracle.OracleScene(s, envmap=env).render(80, 48, frames=3, max_depth=5, rr_depth=2, env_rotation=200.0)
    assert_images_equal(r, imgs)
    r.close()
    s = scenes.sponza_class(target_triangles=30000, aspect=96 / 54, disney=True)
    r = make_renderer(halart, s, 96, 54, max_depth=5, rr_depth=3)
    for _ in range(2):
        r.update(); r.render()
    imgs, st = oracle.OracleScene(s).render(96, 54, frames=2, max_depth=5, rr_depth=3)
    assert_images_equal(r, imgs)
    assert imgs[0][..., :3].mean() > 1e-3
    stg = r.statistics()
    assert (stg.rays_closest_total, stg.rays_shadow_total) == (st.rays_closest, st.rays_shadow)
    r.close()


def test_textures_mips_fetch_and_render_bit_exact(halart, oracle):
    """K8 / A6 / A14: upload + GPU mip chain + software trilinear REPEAT fetch + the four material maps"""
    from test_oracle_render import textured_scene
    s = textured_scene(size=64, fmt_variant=True)
    r = make_renderer(halart, s, 72, 72)
    osc = oracle.OracleScene(s)
    rng = np.random.RandomState(3)
    for tex in range(8):
        assert r.texture_info(tex) == osc.texture_info(tex)
        for level in range(r.texture_info(tex)[2]):
            assert r.read_texture_level(tex, level).tobytes() == osc.texture_level(tex, level).tobytes(), (tex, level)
        q = np.concatenate([(rng.rand(3000, 2) * 4 - 1.5), rng.rand(3000, 1) * 9 - 1], 1).astype(f32)
        assert r.sample_texture(tex, q).tobytes() == osc.sample_texture(tex, q).tobytes()
    r.update_batch(3); r.render()
    imgs, _ = osc.render(72, 72, frames=3)
    assert_images_equal(r, imgs)
    r.close()
    # the atrium with 16 textures on DISNEY + DIFFUSE materials under instancing
    s = scenes.attach_textures(scenes.sponza_class(target_triangles=30000, aspect=96 / 54), sets=5, size=128)
    assert len(s.image_data) >= 15
    r = make_renderer(halart, s, 96, 54)
    r.update_batch(2); r.render()
    imgs, _ = oracle.OracleScene(s).render(96, 54, frames=2)
    assert_images_equal(r, imgs)
    r.close()


def test_gltf_scene_renders_like_the_procedural_one(halart, oracle, tmp_path):
    """config 1's plumbing: cpu::HalaScene::new(path) -> set_scene -> commit -> update, via the glTF mirror"""
    from gltf_writer import write_gltf
    s = scenes.cornell_box()
    scenes.attach_textures(s, sets=1, size=32)
    write_gltf(s, str(tmp_path / "cornell.gltf"))
    loaded = halart.HalaScene.new(str(tmp_path / "cornell.gltf"))
    r = make_renderer(halart, loaded, 48, 48)
    r.update_batch(2); r.render()
    imgs, _ = oracle.OracleScene(loaded).render(48, 48, frames=2)
    assert_images_equal(r, imgs)
    # (not equal to a render of `s` itself: the loader tags every 8-bit image *_SRGB like gltf_loader.rs:395-396, the
    # procedural normal / MR maps of `s` are UNORM)
    assert [i.format for i in loaded.image_data] == [1, 1, 1]
    r.close()


def test_texture_errors(halart):
    s = scenes.cornell_box()
    s.texture2image_mapping[0] = 5  # image 5 does not exist
    r = halart.HalaRenderer("tex", 16, 16, 2, 1, False, False, False, 0)
    with pytest.raises(halart.HalaRendererError, match="The image 5 is not found."):  # gpu_uploader.rs:337
        r.set_scene(s)
    r.close()


def test_render_orthographic_and_thin_lens(halart, oracle):
    s = scenes.cornell_box()
    s.cameras = [H.HalaOrthographicCamera(xmag=300.0, ymag=300.0)]
    r = make_renderer(halart, s, 48, 48)
    r.update(); r.render()
    imgs, _ = oracle.OracleScene(s).render(48, 48, frames=1)
    assert_images_equal(r, imgs)
    r.close()
    s = scenes.cornell_box()
    s.cameras[0].aperture = 12.0; s.cameras[0].focal_distance = 1000.0
    r = make_renderer(halart, s, 48, 48)
    r.update(); r.render()
    imgs, _ = oracle.OracleScene(s).render(48, 48, frames=1)
    assert_images_equal(r, imgs)
    r.close()


def test_furnace_on_gpu(halart):
    """analytic check on the GPU itself: white diffuse sphere in a uniform environment converges to the environment"""
    r = make_renderer(halart, furnace_scene(), 32, 32, max_depth=12, rr_depth=64)
    r.set_sky_color((0.7, 0.7, 0.7, 1)); r.set_ground_color((0.7, 0.7, 0.7, 1))
    for _ in range(64):
        r.update()
    r.render()
    assert abs(r.read_image(0)[8:24, 8:24, :3].mean() - 0.7) < 0.02
    r.close()


def test_max_frames_stops_accumulation(halart):
    """update() is a no-op once total_frames > max_frames (src/rt_renderer.rs:394-396)"""
    r = make_renderer(halart, scenes.cornell_box(), 32, 32, max_frames=2)
    for _ in range(2):
        r.update(); r.render()
    a = r.read_image(0).copy()
    for _ in range(3):
        r.update(); r.render()
    assert r.read_image(0).tobytes() == a.tobytes()
    st = r.statistics()
    assert st.total_frames == 5 and st.updates_rendered == 2
    r.close()


def test_update_batch_equals_single_updates(halart, oracle):
    """hala_rt_update_batch(n) == n x update(): same images bit for bit, same ray counts, same frame bookkeeping,
    including a batch that straddles max_frames and a batch larger than one chunk (16)"""
    env = scenes.sky_sun_envmap(64, 32)
    s = scenes.bunny_class(subdivisions=2, aspect=72 / 40, disney=True)
    a = make_renderer(halart, s, 72, 40, max_depth=4, rr_depth=2, env=env, max_frames=21)
    b = make_renderer(halart, s, 72, 40, max_depth=4, rr_depth=2, env=env, max_frames=21)
    for _ in range(25):
        a.update()
    a.render()
    b.update_batch(3)
    b.update_batch(18)  # 16 + 2, reaches max_frames exactly at the end
    b.update_batch(4)   # entirely beyond max_frames: no-op, but total_frames still advances
    b.render()
    for which in range(4):
        assert a.read_image(which).tobytes() == b.read_image(which).tobytes()
    sa, sb = a.statistics(), b.statistics()
    assert (sa.total_frames, sa.updates_rendered, sa.rays_closest_total, sa.rays_shadow_total) == \
           (sb.total_frames, sb.updates_rendered, sb.rays_closest_total, sb.rays_shadow_total)
    assert sa.total_frames == 25 and sa.updates_rendered == 21
    imgs, _ = oracle.OracleScene(s, envmap=env).render(72, 40, frames=21, max_depth=4, rr_depth=2)
    assert_images_equal(b, imgs)
    a.close(); b.close()


def test_save_images_pfm_trio(halart, oracle, tmp_path):
    s = scenes.cornell_box()
    r = make_renderer(halart, s, 40, 24, tonemap=(True, True, False))
    r.update(); r.render()
    r.save_images(str(tmp_path / "shot.png"))  # <stem>_color.pfm / _albedo.pfm / _normal.pfm (src/rt_renderer.rs:1235-1237)
    imgs, _ = oracle.OracleScene(s).render(40, 24, frames=1)
    color = oracle.tonemap_pixels(imgs[0], True, True, False)  # host tonemap of accum, no exposure (:1256-1316)
    assert (tmp_path / "shot_color.pfm").read_bytes() == oracle.pfm_bytes(color)
    assert (tmp_path / "shot_albedo.pfm").read_bytes() == oracle.pfm_bytes(imgs[1])
    assert (tmp_path / "shot_normal.pfm").read_bytes() == oracle.pfm_bytes(imgs[2])
    r.close()


def test_envmap_file_and_dist_cache(halart, oracle, tmp_path, monkeypatch):
    """set_envmap(path): PFM decode + ./out/<stem>.dist_cache written on first use, read on the second (src/envmap.rs:90-142)"""
    env = scenes.sky_sun_envmap(64, 32)
    lib = halart.load_library()
    monkeypatch.chdir(tmp_path)
    os.makedirs("out")
    halart.check(lib.hala_write_pfm(b"sky.pfm", env.ctypes.data_as(C.POINTER(C.c_float)), 64, 32))
    r = halart.HalaRenderer("envfile", 16, 16, 2, 1, False, False, False, 0)
    r.set_envmap("sky.pfm", 10.0)
    t, m, c = r.env_distribution(64, 32)
    ot, om, oc = oracle.envmap_build_distribution(env)
    assert f32(t).tobytes() == ot.tobytes() and c.tobytes() == oc.tobytes()
    blob = np.fromfile("out/sky.dist_cache", dtype=f32)
    assert blob.size == 1 + 32 + 64 * 32 and blob[0].tobytes() == ot.tobytes() and blob[1:33].tobytes() == om.tobytes() and blob[33:].tobytes() == oc.tobytes()
    # a stale cache is trusted blindly, exactly like the reference (no header / version)
    blob[0] = f32(123.0)
    blob.tofile("out/sky.dist_cache")
    r.set_envmap("sky.pfm", 0.0)
    assert r.env_distribution(64, 32)[0] == 123.0
    r.close()


def test_full_size_render_properties(halart):
    """BASELINE.json configs[1] at full size (1920x1080, 4 spp): size-independent properties instead of an oracle run"""
    s = scenes.cornell_box(aspect=1920 / 1080)
    r = make_renderer(halart, s, 1920, 1080)
    for _ in range(4):
        r.update()
    r.render()
    a = r.read_image(0)
    st = r.statistics()
    assert np.isfinite(a).all() and a[..., :3].min() >= 0 and a[..., 3].min() == 1.0
    assert 0.05 < a[..., :3].mean() < 2.0
    assert st.rays_total > 4 * 1920 * 1080 * 2  # at least primary + one more segment on average
    # idempotence: restarting the accumulation reproduces the same frame bit for bit (counter-based RNG, no atomics on data)
    r.reset_accumulation()
    for _ in range(4):
        r.update()
    r.render()
    assert r.read_image(0).tobytes() == a.tobytes()
    # linearity: every source of radiance (analytic light, emissive material, sky) scaled by 2 scales the image by 2
    # (power-of-two factor => exact in binary32; path construction does not depend on radiance)
    s.lights[0].intensity *= 2.0
    for m in s.materials:
        m.emission = tuple(2.0 * e for e in m.emission)
    r2 = make_renderer(halart, s, 1920, 1080)
    r2.set_env_intensity(2.0)
    for _ in range(4):
        r2.update()
    r2.render()
    assert np.array_equal(r2.read_image(0)[..., :3], a[..., :3] * f32(2.0))
    r.close(); r2.close()


def test_render_opacity_bit_exact(halart, oracle):
    """RENDER_SPEC §7.1d: stochastic pass-through of partly transparent surfaces (material opacity; the texture-alpha factor
    is covered by the BGRA-tagged texture of test_textures_*), same random number budget on both sides"""
    s = scenes.cornell_box()
    s.materials[0].opacity = 0.35   # white: floor, ceiling, back wall, blocks -> paths leak out of the box into the sky
    s.materials[2].opacity = 0.0    # one side wall: never shaded
    r = make_renderer(halart, s, 64, 64, max_depth=6, rr_depth=2)
    for _ in range(3):
        r.update(); r.render()
    imgs, st = oracle.OracleScene(s).render(64, 64, frames=3, max_depth=6, rr_depth=2)
    assert_images_equal(r, imgs)
    stg = r.statistics()
    assert (stg.rays_closest_total, stg.rays_shadow_total) == (st.rays_closest, st.rays_shadow)
    opaque, _ = oracle.OracleScene(scenes.cornell_box()).render(64, 64, frames=3, max_depth=6, rr_depth=2)
    assert np.abs(opaque[0][..., :3] - imgs[0][..., :3]).mean() > 0.02
    r.close()


def test_full_size_scene_bvh_and_rays(halart, oracle):
    """BASELINE configs[3] size (~1 M triangles, instanced meshes, PLOC builder): the tree passes the structural check with
    the bit-exact flattening, a sample of rays agrees with brute force over all triangles, and size-independent properties
    hold for a large batch: any-hit == (closest hit exists), hit ids are valid, a hit re-traced with tmax just below its t
    misses"""
    s = scenes.sponza_class(target_triangles=1_000_000, disney=False)
    r = make_renderer(halart, s, 16, 16)
    osc = oracle.OracleScene(s)
    info = r.bvh_info()
    assert info.triangle_count == osc.triangle_count > 900_000
    rc, depth, two_level = validate_tree(oracle, osc, r)
    assert rc == 0 and not two_level and depth == info.max_depth  # automatic: a scene of 1 M triangles is flattened
    rays = osc.camera_rays(640, 360, 0)
    got = r.trace_rays_host(rays, 0)
    sub = slice(0, None, len(rays) // 1500)
    assert got[sub].tobytes() == osc.trace(rays[sub], 0, brute=True).tobytes()
    hit = got["prim"] != 0xFFFFFFFF
    assert hit.mean() > 0.5 and got["prim"][hit].max() < info.triangle_count
    anyh = r.trace_rays_host(rays, 1)
    assert np.array_equal(anyh["t"] > 0, hit)
    again = rays[hit].copy()
    again["tmax"] = got["t"][hit] * np.float32(0.999)
    closer = r.trace_rays_host(again, 0)
    assert np.all(closer["t"][closer["prim"] != 0xFFFFFFFF] < again["tmax"][closer["prim"] != 0xFFFFFFFF])
    assert (closer["prim"] != 0xFFFFFFFF).mean() < 0.01  # nothing in front of a closest hit (up to grazing ties)
    r.close()


def test_headline_config_full_resolution_bit_exact(halart, oracle):
    """BASELINE configs[1] at its own size and sample count: Cornell box, 1920x1080, 4 spp, max_depth 5, rr_depth 3, rendered as one
    batch like bench.py does, against the oracle pixel for pixel (8.3 M paths), plus the ray totals"""
    s = scenes.cornell_box(aspect=1920 / 1080)
    r = make_renderer(halart, s, 1920, 1080)
    r.update_batch(4)
    r.render()
    imgs, st = oracle.OracleScene(s).render(1920, 1080, frames=4)
    assert_images_equal(r, imgs)
    stg = r.statistics()
    assert (stg.rays_closest_total, stg.rays_shadow_total) == (st.rays_closest, st.rays_shadow)
    assert stg.rays_primary_total == 4 * 1920 * 1080
    r.close()


def test_envmap_from_openexr_file(halart, oracle, tmp_path, monkeypatch):
    """set_envmap(path) with a ZIP-compressed half-float OpenEXR (the `exr` feature of the reference's image crate): the tables
    built from the decoded pixels equal the oracle's tables of the same (half-rounded) pixels"""
    from test_image_decoders import write_exr
    env = scenes.sky_sun_envmap(64, 32)[..., :3].astype(np.float16).astype(f32)
    monkeypatch.chdir(tmp_path)
    os.makedirs("out")  # the reference writes ./out/<stem>.dist_cache next to the process (src/envmap.rs:90-142)
    write_exr("sky.exr", env, "zip", True)
    r = halart.HalaRenderer("envexr", 16, 16, 2, 1, False, False, False, 0)
    r.set_envmap("sky.exr", 0.0)
    t, m, c = r.env_distribution(64, 32)
    rgba = np.concatenate([env, np.ones((32, 64, 1), f32)], axis=-1)
    ot, om, oc = oracle.envmap_build_distribution(rgba)
    assert f32(t).tobytes() == ot.tobytes() and m.tobytes() == om.tobytes() and c.tobytes() == oc.tobytes()
    r.close()


@pytest.mark.parametrize("medium", ["absorb", "emissive"])
def test_render_media_bit_exact(halart, oracle, medium):
    """RENDER_SPEC §7.1e: Beer-Lambert absorption / emission along the segment inside a glass object (polynomial exp, same bits)"""
    env = scenes.sky_sun_envmap(64, 32, sun_gain=100.0)
    s = scenes.bunny_class(subdivisions=3, aspect=80 / 48, disney=True)
    med = H.HalaMedium(1, (0.9, 0.4, 0.2), 2.5, 0.0) if medium == "absorb" else H.HalaMedium(3, (0.2, 0.5, 1.0), 0.8, 0.0)
    s.materials[0] = H.HalaMaterial(type=1, base_color=(1.0, 1.0, 1.0), metallic=0.0, roughness=0.15, specular_transmission=1.0, ior=1.45,
                                    emission=(0.05, 0.0, 0.0) if medium == "emissive" else (0.0, 0.0, 0.0), medium=med)
    r = make_renderer(halart, s, 80, 48, max_depth=8, rr_depth=3, env=env)
    for _ in range(3):
        r.update(); r.render()
    imgs, st = oracle.OracleScene(s, envmap=env).render(80, 48, frames=3, max_depth=8, rr_depth=3)
    assert_images_equal(r, imgs)
    s.materials[0].medium = H.HalaMedium()
    plain, _ = oracle.OracleScene(s, envmap=env).render(80, 48, frames=3, max_depth=8, rr_depth=3)
    assert np.abs(plain[0][16:32, 30:50, :3] - imgs[0][16:32, 30:50, :3]).mean() > 0.005
    r.close()


def test_media_glass_fixture_scene_bit_exact(halart, oracle):
    """the frozen media / glass frame (tests/golden/media_glass_48x32_2spp.npz): GPU == oracle bit for bit on this box's scene
    arrays, and both within the fixture's tolerance"""
    import sys
    sys.path.insert(0, GOLDEN)
    from make_golden import media_scene
    g = np.load(os.path.join(GOLDEN, "media_glass_48x32_2spp.npz"))
    s = media_scene()
    r = make_renderer(halart, s, 48, 32, max_depth=12, rr_depth=3, env=g["env"], env_rot=15.0)
    r.update_batch(2); r.render()
    imgs, st = oracle.OracleScene(s, envmap=g["env"]).render(48, 32, frames=2, max_depth=12, rr_depth=3, env_rotation=15.0)
    assert_images_equal(r, imgs)
    d = np.abs(r.read_image(0) - g["accum"])
    assert np.mean(d.max(axis=-1) > 1e-3) < 0.02
    r.close()


def test_invisible_surfaces_and_shadow_rays(halart, oracle):
    """RENDER_SPEC 7.1d: any-hit rays do not see opacity-0 surfaces (the any-hit launches traverse a triangle copy in which they are
    degenerate) while closest-hit rays do; a material edit that makes a surface (in)visible is picked up by refit"""
    from test_oracle_render import fog_over_floor_scene
    s = fog_over_floor_scene(True)
    r = make_renderer(halart, s, 48, 48, max_depth=6)
    osc = oracle.OracleScene(s)
    rays = random_rays(6000, np.array((-3, 0.05, -3.0)), np.array((3, 4, 3.0)), 3)
    for mode in (0, 1):
        assert r.trace_rays_host(rays, mode).tobytes() == osc.trace(rays, mode).tobytes(), mode
    assert (r.trace_rays_host(rays, 1)["t"] > 0).sum() < (r.trace_rays_host(rays, 0)["prim"] != 0xFFFFFFFF).sum()
    r.update_batch(3); r.render()
    imgs, st = osc.render(48, 48, frames=3, max_depth=6)
    assert_images_equal(r, imgs)
    lit = r.read_image(0)[20:28, 20:28, :3].mean()
    # the fog's boundary becomes an ordinary white surface: it shadows the floor again
    solid = H.HalaMaterial(type=0, base_color=(1.0, 1.0, 1.0), roughness=0.5, opacity=1.0)
    r.update_material(1, solid); r.refit()
    s.materials[1] = solid
    osc2 = oracle.OracleScene(s)
    assert r.trace_rays_host(rays, 1).tobytes() == osc2.trace(rays, 1).tobytes()
    r.update_batch(3); r.render()
    imgs2, _ = osc2.render(48, 48, frames=3, max_depth=6)
    assert_images_equal(r, imgs2)
    # ... and back
    fog = fog_over_floor_scene(True).materials[1]
    r.update_material(1, fog); r.refit()
    assert r.trace_rays_host(rays, 1).tobytes() == osc.trace(rays, 1).tobytes()
    r.update_batch(3); r.render()
    assert_images_equal(r, imgs)
    assert lit > 0.0
    r.close()


@pytest.mark.parametrize("big", [False, True])
def test_translucent_shadow_rays_bit_exact(halart, oracle, big):
    """RENDER_SPEC 7.1d, connections: sheets of opacity 0.6 with a cut-out (0 / 1 alpha checker) base-colour map between a quad light,
    an env map and a floor.  Whether a sheet blocks a connection is decided per (connection key, triangle) from opacity x alpha at the
    hit, so the image does not depend on the traversal: the LDS-staged kernels (each lane tests its own leaves) and the large-scene
    kernels (`big`: a 5 120-triangle blob pushes the tree past the LDS budget; the wave tests leaves cooperatively)
    must both equal the oracle bit for bit — images, any-hit ray batches, ray totals — also after a material edit + refit changes
    which triangles are translucent."""
    from test_oracle_render import sheet_over_floor_scene
    s = sheet_over_floor_scene(opacity=0.6, alpha_checker=True, sheets=3)
    if big:
        blob = scenes.blob_mesh(subdivisions=4)
        blob.material_index = 0
        s.meshes.append(H.HalaMesh([blob]))
        m = np.eye(4, dtype=np.float32); m[:3, :3] *= 0.5; m[:3, 3] = (1.5, 0.6, -1.0)
        s.nodes.append(H.HalaNode(name="blob", mesh_index=len(s.meshes) - 1, local_transform=m))
    env = scenes.sky_sun_envmap(64, 32, sun_gain=50.0)
    r = make_renderer(halart, s, 64, 64, max_depth=4, rr_depth=2, env=env)
    assert (r.bvh_info().lds_node_count == 0) == big
    osc = oracle.OracleScene(s, envmap=env)
    rays = random_rays(8000, np.array((-5, 0.05, -5.0)), np.array((5, 4, 5.0)), 7)
    for mode in (0, 1):
        assert r.trace_rays_host(rays, mode).tobytes() == osc.trace(rays, mode).tobytes(), mode
    blocked = (r.trace_rays_host(rays, 1)["t"] > 0).sum()
    assert 0 < blocked < (r.trace_rays_host(rays, 0)["prim"] != 0xFFFFFFFF).sum()  # some rays pass the sheets they hit
    r.update(); r.update_batch(3); r.render()
    imgs, st = osc.render(64, 64, frames=4, max_depth=4, rr_depth=2)
    assert_images_equal(r, imgs)
    stg = r.statistics()
    assert (stg.rays_closest_total, stg.rays_shadow_total) == (st.rays_closest, st.rays_shadow)
    # the sheets become ordinary opaque surfaces (no map, opacity 1): class 2 -> 0, the any-hit copy of the triangles is rewritten
    solid = H.HalaMaterial(type=0, base_color=(1.0, 1.0, 1.0), roughness=0.5)
    r.update_material(1, solid); r.refit()
    s.materials[1] = solid
    osc2 = oracle.OracleScene(s, envmap=env)
    assert r.trace_rays_host(rays, 1).tobytes() == osc2.trace(rays, 1).tobytes()
    r.update_batch(2); r.render()
    imgs2, _ = osc2.render(64, 64, frames=2, max_depth=4, rr_depth=2)
    assert_images_equal(r, imgs2)
    assert float(np.abs(imgs2[0][..., :3] - imgs[0][..., :3]).mean()) > 1e-3
    r.close()


@pytest.mark.parametrize("big", [False, True])
@pytest.mark.parametrize("opacity", [0.0, 0.5])
def test_media_on_connections_bit_exact(halart, oracle, big, opacity):
    """RENDER_SPEC 7.1g: connections that cross a slab of absorbing medium (behind an invisible or a half-opaque boundary) and a ball of
    scattering fog keep exp(-optical depth) of their contribution; the optical depth is an order-independent fixed-point sum over the
    boundary crossings, so the LDS-staged kernels, the wave-cooperative large-scene kernels (`big`) and the oracle agree bit for bit"""
    from test_oracle_render import slab_over_floor_scene
    s = slab_over_floor_scene(medium=H.HalaMedium(1, (0.2, 0.5, 0.8), 2.0, 0.0), opacity=opacity)
    fog = scenes.blob_mesh(subdivisions=2, amplitude=0.0)
    fog.material_index = len(s.materials)
    s.materials.append(H.HalaMaterial(type=0, base_color=(1.0, 1.0, 1.0), roughness=0.5, opacity=0.0, medium=H.HalaMedium(2, (0.95, 0.9, 0.8), 0.8, 0.3)))
    s.meshes.append(H.HalaMesh([fog]))
    m = np.eye(4, dtype=np.float32); m[:3, :3] *= 0.7; m[:3, 3] = (0.8, 1.0, -0.3)
    s.nodes.append(H.HalaNode(name="fog", mesh_index=len(s.meshes) - 1, local_transform=m))
    if big:
        blob = scenes.blob_mesh(subdivisions=4)
        blob.material_index = 0
        s.meshes.append(H.HalaMesh([blob]))
        m2 = np.eye(4, dtype=np.float32); m2[:3, :3] *= 0.4; m2[:3, 3] = (-1.2, 0.5, -0.8)
        s.nodes.append(H.HalaNode(name="blob", mesh_index=len(s.meshes) - 1, local_transform=m2))
    env = scenes.sky_sun_envmap(64, 32, sun_gain=50.0)
    r = make_renderer(halart, s, 64, 64, max_depth=5, rr_depth=2, env=env)
    assert (r.bvh_info().lds_node_count == 0) == big
    osc = oracle.OracleScene(s, envmap=env)
    rays = random_rays(6000, np.array((-5, 0.05, -5.0)), np.array((5, 4, 5.0)), 9)
    for mode in (0, 1):
        assert r.trace_rays_host(rays, mode).tobytes() == osc.trace(rays, mode).tobytes(), mode
    r.update(); r.update_batch(3); r.render()
    imgs, st = osc.render(64, 64, frames=4, max_depth=5, rr_depth=2)
    assert_images_equal(r, imgs)
    stg = r.statistics()
    assert (stg.rays_closest_total, stg.rays_shadow_total) == (st.rays_closest, st.rays_shadow)
    # without the media the frame is visibly brighter under the slab
    s.materials[1].medium = H.HalaMedium()
    plain, _ = oracle.OracleScene(s, envmap=env).render(64, 64, frames=4, max_depth=5, rr_depth=2)
    assert float(plain[0][..., :3].mean()) > float(imgs[0][..., :3].mean()) * 1.02
    r.close()


@pytest.mark.parametrize("boundary", ["glass", "invisible"])
def test_render_scattering_medium_bit_exact(halart, oracle, boundary):
    """RENDER_SPEC 7.1f: free-flight sampling (polynomial log), Henyey-Greenstein scattering, no NEE at scattering vertices (the
    next emitter hit counts in full) - inside a glass blob and inside an invisible (opacity 0) one, lights + env map, batched"""
    env = scenes.sky_sun_envmap(64, 32, sun_gain=100.0)
    s = scenes.bunny_class(subdivisions=3, aspect=80 / 48, disney=True)
    med = H.HalaMedium(2, (0.95, 0.7, 0.4), 3.0, 0.6)
    if boundary == "glass":
        s.materials[0] = H.HalaMaterial(type=1, base_color=(1.0, 1.0, 1.0), metallic=0.0, roughness=0.15, specular_transmission=1.0, ior=1.45, medium=med)
    else:
        s.materials[0] = H.HalaMaterial(type=0, base_color=(1.0, 1.0, 1.0), roughness=0.5, opacity=0.0, medium=med)
    r = make_renderer(halart, s, 80, 48, max_depth=24, rr_depth=3, env=env)
    r.update(); r.update_batch(3); r.render()
    imgs, st = oracle.OracleScene(s, envmap=env).render(80, 48, frames=4, max_depth=24, rr_depth=3)
    assert_images_equal(r, imgs)
    stg = r.statistics()
    assert (stg.rays_closest_total, stg.rays_shadow_total) == (st.rays_closest, st.rays_shadow)
    s.materials[0].medium = H.HalaMedium()
    plain, _ = oracle.OracleScene(s, envmap=env).render(80, 48, frames=4, max_depth=24, rr_depth=3)
    assert np.abs(plain[0][16:32, 30:50, :3] - imgs[0][16:32, 30:50, :3]).mean() > 0.005
    r.close()


@pytest.mark.parametrize("config,fusion,instancing", [(3, 1, None), (4, 1, None), (4, 0, None), (4, 1, True)])
def test_large_configs_full_resolution_bit_exact(halart, oracle, config, fusion, instancing):
    """BASELINE configs[2] (82 k-triangle Disney blob under a 2048x1024 sun/sky map, MIS; 16 spp) and configs[3] (1 M-triangle atrium:
    24 materials incl. glass and clearcoat, 18 mip-mapped textures, quad lights + env; 4 spp) at 1920x1080 and their own sample
    counts, pixel for pixel against the oracle ON ITS OWN TREE (binned-SAH BVH2: nothing of the product's builder is shared), plus
    the ray totals.  configs[3] runs with the fused launches every default host gets (k_trace_shadow_then_batch, large-scene variant)
    and with one launch per pass."""
    if config == 3:
        s = scenes.bunny_class(subdivisions=6, disney=True)
        env, spp = scenes.sky_sun_envmap(2048, 1024), 16
    else:
        s = scenes.sponza_class(target_triangles=1_000_000)
        scenes.attach_textures(s, sets=6, size=1024)
        env, spp = scenes.sky_sun_envmap(1024, 512, sun_gain=50.0), 4
    # instancing True: the atrium as a two-level tree (RENDER_SPEC 4.5: its 28 columns and 14 arches are instances of three primitives),
    # the oracle told the same; None: the automatic choice (flattened at this size)
    with (two_level_trees(oracle) if instancing else contextlib.nullcontext()) as build:
        r = make_renderer(halart, s, 1920, 1080, env=env, build=build)
        osc = oracle.OracleScene(s, envmap=env)  # its own tree (RENDER_SPEC 4.1b pads the boxes: results do not depend on the tree)
    assert (r.bvh_info().instance_ref_count > 0) == bool(instancing)
    r.set_pass_fusion(fusion)
    r.update_batch(spp)
    r.render()
    imgs, st = osc.render(1920, 1080, frames=spp)
    assert_images_equal(r, imgs)
    stg = r.statistics()
    assert (stg.rays_closest_total, stg.rays_shadow_total) == (st.rays_closest, st.rays_shadow)
    assert float(imgs[0][..., :3].mean()) > 0.01
    r.close()


def render_random_scene_both(halart, oracle, seed, big=False, instances=False):
    """one random scene (tests/random_scenes.py) on the GPU and in the oracle -> number of differing pixels per image.  instances: with
    instanced objects and a two-level tree (RENDER_SPEC 4.5) on both sides"""
    from random_scenes import random_scene
    s, env, kw = random_scene(seed, big, instances)
    r = halart.HalaRenderer("random", kw["width"], kw["height"], kw["max_depth"], kw["rr_depth"], *kw["tonemap"], 0)
    if instances:
        r.set_build_options(instancing=True)
        oracle.set_instancing(True)
    if env is not None:
        r.set_envmap(env, kw["env_rotation"])
        r.set_env_intensity(kw["env_intensity"])
    r.set_exposure_value(kw["exposure"])
    r.set_scene(s)
    r.commit()
    # the frames in two uneven batches (update_batch and update are held to the same images)
    first = kw["frames"] // 2
    if first:
        r.update_batch(first)
    for _ in range(kw["frames"] - first):
        r.update()
    r.render()
    osc = oracle.OracleScene(s, envmap=env)
    imgs, st = osc.render(kw["width"], kw["height"], frames=kw["frames"], max_depth=kw["max_depth"], rr_depth=kw["rr_depth"], tonemap=kw["tonemap"],
                          env_rotation=kw["env_rotation"] if env is not None else 0.0, env_intensity=kw["env_intensity"] if env is not None else 1.0,
                          exposure=kw["exposure"])
    bad = [int(np.any(r.read_image(k) != imgs[k], axis=-1).sum()) for k in range(4)]
    stg = r.statistics()
    rays_ok = (stg.rays_closest_total, stg.rays_shadow_total) == (st.rays_closest, st.rays_shadow)
    lit = float(imgs[0][..., :3].mean())
    if seed % 3 == 0:  # random edits without a rebuild: a node moved, a material replaced, a mesh deformed -> refit -> the same comparison
        from random_scenes import random_material, _xform
        rs = np.random.RandomState(seed + 77)
        node = 1 + int(rs.randint(len(s.meshes) - 1))  # an object node (0 is the root, objects follow)
        s.nodes[node].local_transform = _xform(rs, rs.uniform(-1.2, 1.2, 3))
        r.update_node_transform(node, s.nodes[node].local_transform)
        mi = int(rs.randint(len(s.materials)))
        keep_maps = s.materials[mi]
        m = random_material(rs)
        m.base_color_map_index, m.normal_map_index = keep_maps.base_color_map_index, keep_maps.normal_map_index
        m.metallic_roughness_map_index, m.emission_map_index = keep_maps.metallic_roughness_map_index, keep_maps.emission_map_index
        s.materials[mi] = m
        r.update_material(mi, m)
        v = s.meshes[0].primitives[0].vertices.copy()
        v["position"] += (0.05 * np.sin(5.0 * v["position"][:, ::-1])).astype(np.float32)
        s.meshes[0].primitives[0].vertices = v
        r.update_vertices(0, 0, v)
        r.refit()
        r.update_batch(2)
        r.render()
        osc2 = oracle.OracleScene(s, envmap=env)
        imgs2, st2 = osc2.render(kw["width"], kw["height"], frames=2, max_depth=kw["max_depth"], rr_depth=kw["rr_depth"], tonemap=kw["tonemap"],
                                 env_rotation=kw["env_rotation"] if env is not None else 0.0, env_intensity=kw["env_intensity"] if env is not None else 1.0,
                                 exposure=kw["exposure"])
        bad = [b + int(np.any(r.read_image(k) != imgs2[k], axis=-1).sum()) for k, b in enumerate(bad)]
    r.close()
    oracle.set_instancing(False)
    return bad, rays_ok, lit


@pytest.mark.parametrize("seed", list(range(12)))
def test_random_scenes_bit_exact(halart, oracle, seed):
    """every feature of the rendering spec drawn at random and combined (tests/random_scenes.py): all four images and the ray counts of
    the GPU render equal the oracle's; scripts/soak_random_scenes.py runs the same comparison over hundreds of seeds"""
    bad, rays_ok, lit = render_random_scene_both(halart, oracle, seed, big=seed % 6 == 5)
    assert bad == [0, 0, 0, 0] and rays_ok, (seed, bad, rays_ok)
    assert lit >= 0.0


@pytest.mark.parametrize("seed", [100, 101, 102, 103, 104, 105, 106, 107])
def test_random_scenes_with_instances_two_level_bit_exact(halart, oracle, seed):
    """the same with objects referenced by several nodes and a two-level tree on both sides (RENDER_SPEC 4.5): translucent / invisible /
    medium-bounding / textured / emissive materials on instanced primitives, mirrored and nested transforms, and (every third seed) a
    refit after a moved node, a replaced material and a deformed mesh"""
    try:
        bad, rays_ok, lit = render_random_scene_both(halart, oracle, seed, big=seed % 4 == 1, instances=True)
    finally:
        oracle.set_instancing(False)
    assert bad == [0, 0, 0, 0] and rays_ok, (seed, bad, rays_ok)


# ---- two-level trees (RENDER_SPEC 4.5) ---------------------------------------------------------------------------------------------
def _instanced_cornell(extra):
    """the Cornell box with further instances of the short block's mesh: extra = [(4x4 local transform, parent node or None)]"""
    s = scenes.cornell_box()
    for k, (m, parent) in enumerate(extra):
        s.nodes.append(H.HalaNode(name=f"copy_{k}", mesh_index=1, parent=parent, local_transform=np.asarray(m, dtype=f32)))
    return s


def _xf(t=(0, 0, 0), scale=(1, 1, 1), ry=0.0, rx=0.0):
    cy, sy, cx, sx = np.cos(ry), np.sin(ry), np.cos(rx), np.sin(rx)
    R = np.array([[cy, 0, sy], [0, 1, 0], [-sy, 0, cy]]) @ np.array([[1, 0, 0], [0, cx, -sx], [0, sx, cx]])
    m = np.eye(4)
    m[:3, :3] = R @ np.diag(scale)
    m[:3, 3] = t
    return m.astype(f32)


def test_two_level_instances_with_general_transforms(halart, oracle):
    """RENDER_SPEC 4.5 on small scenes (a two-level tree whatever the size): the short block's mesh referenced four more times —
    rotated + non-uniformly scaled, MIRRORED (negative determinant), under a parent that shears it, and squashed flat (determinant 0:
    that instance is flattened to world space, its siblings stay instanced).  Ray batches equal the oracle's own tree and brute force bit
    for bit, the render equals the oracle's, the tree passes the structural check, the step counts equal the oracle's on that tree."""
    shear = np.eye(4, dtype=f32); shear[0, 1] = 0.35; shear[:3, 3] = (60.0, 200.0, 120.0)
    s = _instanced_cornell([(_xf((250, 330, 230), (0.6, 1.4, 0.5), ry=0.7, rx=0.2), None),
                            (_xf((520, 60, 90), (-0.8, 0.9, 0.7), ry=-0.4), None),
                            (shear, None),
                            (_xf((-20, 150, 200), (0.5, 0.5, 0.5), ry=1.1), 6),     # child of the shearing node
                            (_xf((100, 400, 300), (0.7, 0.0, 0.7)), None)])        # flat: not invertible
    with two_level_trees(oracle) as build:
        r = make_renderer(halart, s, 64, 48, build=build)
        osc = oracle.OracleScene(s)
    info = r.bvh_info()
    assert info.instance_ref_count == 5 and info.instance_node_count >= 1  # the original + 4 invertible copies; the flat one is flattened
    assert info.stored_triangle_count == info.triangle_count - 4 * 10 and info.lds_node_count == 0
    rc, depth, two_level = validate_tree(oracle, osc, r)
    assert rc == 0 and two_level
    omn, omx = osc.bounds()
    assert list(info.scene_min) == list(omn) and list(info.scene_max) == list(omx)
    pad = (omx - omn) * 0.2
    rays = np.concatenate([random_rays(40000, omn - pad, omx + pad, 3), osc.camera_rays(160, 120, 0)])
    nodes, tris = r.download_bvh()
    refs = r.download_instance_refs()
    for mode in (0, 1):
        got, cnt = r.trace_rays_host(rays, mode, count_steps=True)
        assert got.tobytes() == osc.trace(rays, mode).tobytes()
        assert cnt == oracle.trace_on_bvh(nodes, tris, rays, mode, refs)[1]
    sub = rays[::13]
    assert r.trace_rays_host(sub, 0).tobytes() == osc.trace(sub, 0, brute=True).tobytes()
    r.update_batch(3); r.render()
    imgs, st = osc.render(64, 48, frames=3)
    assert_images_equal(r, imgs)
    stg = r.statistics()
    assert (stg.rays_closest_total, stg.rays_shadow_total) == (st.rays_closest, st.rays_shadow)
    # everything flattened (the default at this size): the other arithmetic, the same picture to rounding
    r2 = make_renderer(halart, s, 64, 48)
    flat = oracle.OracleScene(s)
    assert r2.bvh_info().instance_ref_count == 0
    r2.update_batch(3); r2.render()
    fimgs, _ = flat.render(64, 48, frames=3)
    assert_images_equal(r2, fimgs)
    assert np.abs(fimgs[0][..., :3] - imgs[0][..., :3]).mean() < 2e-3
    r.close(); r2.close()


def test_two_level_refit_of_a_moved_instance_touches_only_the_instance_levels(halart, oracle, request):
    """hala_rt_update_node_transform on a node of an INSTANCED primitive + refit: the primitives' trees and all triangles stay byte for
    byte, only the instance levels (the first instance_node_count nodes) and the instance references are rebuilt; a node of a primitive
    that is referenced once is flattened geometry: moving it refits the world tree.  Both times the frame equals the oracle's."""
    s = scenes.sponza_class(target_triangles=60_000)
    oracle.set_instancing(True)
    request.addfinalizer(lambda: oracle.set_instancing(False))
    r = make_renderer(halart, s, 48, 27, build=dict(instancing=True))
    info = r.bvh_info()
    T = info.instance_node_count
    assert T > 0
    n0, t0 = r.download_bvh()
    refs0 = r.download_instance_refs()
    col = next(k for k, n in enumerate(s.nodes) if n.name == "column_5")
    m = np.array(s.nodes[col].local_transform, dtype=f32).copy()
    m[:3, 3] += np.array([1.5, 0.0, -2.0], dtype=f32)
    r.update_node_transform(col, m)
    r.refit()
    n1, t1 = r.download_bvh()
    refs1 = r.download_instance_refs()
    assert t0.tobytes() == t1.tobytes()
    assert n0.reshape(-1, 16)[T:].tobytes() == n1.reshape(-1, 16)[T:].tobytes()      # every tree below the instance levels: untouched
    assert n0.reshape(-1, 16)[:T].tobytes() != n1.reshape(-1, 16)[:T].tobytes() and refs0.tobytes() != refs1.tobytes()
    s.nodes[col].local_transform = m
    osc = oracle.OracleScene(s)
    assert validate_tree(oracle, osc, r)[0] == 0
    r.update(); r.update(); r.render()
    img, _ = osc.render(48, 27, frames=2)
    assert r.read_image(0).tobytes() == img[0].tobytes()
    # a drape is referenced once: flattened into the world tree, which a move refits
    dr = next(k for k, n in enumerate(s.nodes) if n.name == "drape_2")
    m2 = np.array(s.nodes[dr].local_transform, dtype=f32).copy()
    m2[:3, 3] += np.array([0.0, 0.8, 1.0], dtype=f32)
    r.update_node_transform(dr, m2)
    r.refit()
    n2, t2 = r.download_bvh()
    assert t2.tobytes() != t1.tobytes() and n2.reshape(-1, 16)[T:].tobytes() != n1.reshape(-1, 16)[T:].tobytes()
    s.nodes[dr].local_transform = m2
    osc = oracle.OracleScene(s)
    assert validate_tree(oracle, osc, r)[0] == 0
    r.update(); r.update(); r.render()
    img, _ = osc.render(48, 27, frames=2)
    assert r.read_image(0).tobytes() == img[0].tobytes()
    r.close()


def test_two_level_tree_stores_instanced_primitives_once(halart):
    """the 1 M-triangle atrium: 28 columns + 14 arches are instances of three primitives.  Nodes + triangles + shading records of the
    two-level tree against everything flattened (hala_rt_build_options::instancing = 1)"""
    s = scenes.sponza_class(target_triangles=1_000_000, disney=False)
    sizes = {}
    for instancing in (True, False):
        r = make_renderer(halart, s, 16, 16, build=dict(instancing=instancing))
        i = r.bvh_info()
        sizes[instancing] = (i.tree_bytes, i.stored_triangle_count, i.triangle_count)
        r.close()
    assert sizes[False][1] == sizes[False][2] == sizes[True][2] > 900_000
    assert sizes[True][1] < 0.45 * sizes[True][2] and sizes[True][0] < 0.45 * sizes[False][0], sizes


def test_two_level_sharded_and_reclassified_on_refit(halart, oracle, request):
    """two-level trees under a tile shard (three emulated ranks; the gathered frame ≡ the oracle's) and through a refit that changes WHICH
    instances are instanced: a node's transform is squashed flat (determinant 0 → that instance is flattened: the trees are rebuilt), then
    made invertible again"""
    import torch
    oracle.set_instancing(True)
    request.addfinalizer(lambda: oracle.set_instancing(False))
    s = _instanced_cornell([(_xf((250, 330, 230), (0.6, 1.4, 0.5), ry=0.7), None), (_xf((520, 60, 90), (-0.8, 0.9, 0.7), ry=-0.4), None)])
    w, h, world, ts = 96, 64, 3, 16
    want, _ = oracle.OracleScene(s).render(w, h, frames=2)
    parts, last = [], None
    for rank in range(world):
        r = halart.HalaRenderer("shard2", w, h, 5, 3, False, False, False, 0)
        r.set_build_options(instancing=True)
        r.set_tile_shard(rank, world, ts)
        r.set_scene(s); r.commit()
        assert r.bvh_info().instance_ref_count == 3
        r.update_batch(2); r.render(); r.wait_idle()
        ptr, nbytes = r.tile_buffer(0)
        parts.append(torch.as_tensor(halart.dist._DeviceView(ptr, nbytes // 4), device="cuda:0").clone())
        if last is not None:
            last.close()
        last = r
    gathered = torch.cat(parts).contiguous()
    last.scatter_gathered_tiles(0, gathered.data_ptr(), gathered.numel() * 4)
    assert last.read_image(0).tobytes() == want[0].tobytes()
    last.close()
    # reclassification through refit
    r = make_renderer(halart, s, 64, 48, build=dict(instancing=True))
    node = len(s.nodes) - 1
    flat = _xf((520, 60, 90), (0.8, 0.0, 0.7))
    r.update_node_transform(node, flat)
    r.refit()
    s.nodes[node].local_transform = flat
    osc = oracle.OracleScene(s)
    assert r.bvh_info().instance_ref_count == 2 and validate_tree(oracle, osc, r)[0] == 0
    r.update_batch(2); r.render()
    img, _ = osc.render(64, 48, frames=2)
    assert r.read_image(0).tobytes() == img[0].tobytes()
    back = _xf((500, 80, 120), (0.5, 1.1, 0.9), ry=0.9)
    r.update_node_transform(node, back)
    r.refit()
    s.nodes[node].local_transform = back
    osc = oracle.OracleScene(s)
    assert r.bvh_info().instance_ref_count == 3 and validate_tree(oracle, osc, r)[0] == 0
    r.update_batch(2); r.render()
    img, _ = osc.render(64, 48, frames=2)
    assert r.read_image(0).tobytes() == img[0].tobytes()
    r.close()
