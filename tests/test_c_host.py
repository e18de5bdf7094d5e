"""examples/render_gltf.c: a host in plain C99 over include/halart.h (the drop-in boundary without Python or C++ on the caller's
side).  CPU: the header and the example are C99-clean, link against libhalart.so, and fail loudly without a GPU.  GPU: the images
the C host saves are byte for byte what the Python mirror of the same call sequence saves."""
import os
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))

from hala_renderer_amd import scenes  # noqa: E402
from gltf_writer import write_gltf  # noqa: E402


def build_c_host(halart, tmp_path):
    exe = tmp_path / "render_gltf"
    lib_dir = os.path.dirname(halart.LIB_PATH)
    cmd = ["gcc", "-std=c99", "-Wall", "-Wextra", "-Werror", "-pedantic", "-I", os.path.join(ROOT, "include"),
           os.path.join(ROOT, "examples", "render_gltf.c"), "-L", lib_dir, "-lhalart", "-Wl,-rpath," + lib_dir, "-o", str(exe)]
    subprocess.run(cmd, check=True, capture_output=True, text=True)
    return str(exe)


def test_c99_host_compiles_links_and_needs_a_gpu(halart, tmp_path):
    import torch
    exe = build_c_host(halart, tmp_path)
    write_gltf(scenes.cornell_box(aspect=1.5), str(tmp_path / "cornell.gltf"))
    p = subprocess.run([exe], capture_output=True, text=True)
    assert p.returncode == 2 and "usage" in p.stderr
    if torch.cuda.is_available():
        pytest.skip("the no-GPU behaviour is checked where there is no GPU")
    p = subprocess.run([exe, str(tmp_path / "cornell.gltf"), str(tmp_path / "frame"), "48", "32", "1"], capture_output=True, text=True)
    assert p.returncode == 1 and "HIP" in p.stderr  # no CPU path: the library says so instead of rendering on the host


@pytest.mark.gpu
def test_c99_host_saves_the_same_images_as_the_python_mirror(halart, tmp_path):
    from hala_renderer_amd.native_scene import NativeScene
    exe = build_c_host(halart, tmp_path)
    s = scenes.sponza_class(target_triangles=6000)
    gltf = str(tmp_path / "atrium.gltf")
    write_gltf(s, gltf)
    env = scenes.sky_sun_envmap(64, 32, sun_gain=50.0)
    env_path = str(tmp_path / "sky.pfm")
    with open(env_path, "wb") as f:  # PFM: bottom row first (src/rt_renderer.rs:1318-1334 is the writer's side of the same format)
        f.write(b"PF\n64 32\n-1.0\n")
        f.write(np.ascontiguousarray(env[::-1, :, :3], dtype="<f4").tobytes())
    os.makedirs("out", exist_ok=True)  # ./out/<stem>.dist_cache (src/envmap.rs:90-142)
    w, h, spp = 96, 54, 3
    p = subprocess.run([exe, gltf, str(tmp_path / "c_host"), str(w), str(h), str(spp), env_path, "30"], capture_output=True, text=True)
    assert p.returncode == 0, p.stderr
    assert p.stdout.startswith("triangles ") and "instance references 0" in p.stdout and "\nframes 3 rays " in p.stdout
    # the reference's BLAS / TLAS split through the C ABI (hala_rt_set_build_options from C99): fewer stored triangles, the same picture to rounding
    p2 = subprocess.run([exe, gltf, str(tmp_path / "c_host2"), str(w), str(h), str(spp), env_path, "30", "two-level"], capture_output=True, text=True)
    assert p2.returncode == 0, p2.stderr
    first = p2.stdout.splitlines()[0].split()
    assert int(first[1]) > int(first[3]) and int(first[6]) == 42, p2.stdout  # triangles > stored; 28 columns + 14 arches are instance references
    r = halart.HalaRenderer("py", w, h, 8, 3, False, False, False, 0)
    r.set_envmap(env_path, 30.0)
    r.set_scene(NativeScene(gltf))
    r.commit()
    for _ in range(spp):
        r.update(); r.render()
    r.save_images(str(tmp_path / "py_host"))
    r.close()
    for aov in ("color", "albedo", "normal"):
        a = open(tmp_path / f"c_host_{aov}.pfm", "rb").read()
        b = open(tmp_path / f"py_host_{aov}.pfm", "rb").read()
        assert len(a) > w * h * 12 and a == b, aov
    hdr = len(b"PF\n96 54\n-1.0\n")
    flat = np.frombuffer(open(tmp_path / "c_host_albedo.pfm", "rb").read()[hdr:], dtype="<f4")
    two = np.frombuffer(open(tmp_path / "c_host2_albedo.pfm", "rb").read()[hdr:], dtype="<f4")
    assert flat.shape == two.shape and np.abs(flat - two).mean() < 2e-3
