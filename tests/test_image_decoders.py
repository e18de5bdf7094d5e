"""CPU tier: the float-image decoders behind set_envmap(path) (`image::open` of src/envmap.rs:48-53): scanline OpenEXR written
by a small independent writer below (NONE / RLE-free / ZIPS / ZIP; half and float; RGB, RGBA, Y), Radiance .hdr and .pfm."""
import ctypes as C
import struct
import zlib

import numpy as np
import pytest

import hala_renderer_amd as H

f32 = np.float32


def write_exr(path, img, compression="zip", half=False, channels="RGB", data_window_origin=(0, 0), decreasing_y=False, tile=None):
    """img [H, W, len(channels)] float32.  OpenEXR 2 scanline file — or, tile = (tw, th), a single-level tiled one — channels stored
    alphabetically as the format requires."""
    h, w, nc = img.shape
    assert nc == len(channels)
    order = sorted(range(nc), key=lambda k: channels[k])
    comp = {"none": 0, "zips": 2, "zip": 3}[compression]
    block = 16 if comp == 3 else 1
    x0, y0 = data_window_origin

    def attr(name, typ, payload):
        return name.encode() + b"\0" + typ.encode() + b"\0" + struct.pack("<i", len(payload)) + payload

    chl = b"".join(channels[k].encode() + b"\0" + struct.pack("<iBxxxii", 1 if half else 2, 0, 1, 1) for k in order) + b"\0"
    box = struct.pack("<iiii", x0, y0, x0 + w - 1, y0 + h - 1)
    header = (struct.pack("<ii", 20000630, 2 | (0x200 if tile else 0)) + attr("channels", "chlist", chl) + attr("compression", "compression", bytes([comp]))
              + attr("dataWindow", "box2i", box) + attr("displayWindow", "box2i", box) + attr("lineOrder", "lineOrder", bytes([1 if decreasing_y else 0]))
              + attr("pixelAspectRatio", "float", struct.pack("<f", 1.0)) + attr("screenWindowCenter", "v2f", struct.pack("<ff", 0.0, 0.0))
              + attr("screenWindowWidth", "float", struct.pack("<f", 1.0))
              + (attr("tiles", "tiledesc", struct.pack("<IIB", tile[0], tile[1], 0)) if tile else b"") + b"\0")

    def pack(raw):
        if not comp:
            return raw
        a = np.frombuffer(raw, dtype=np.uint8)
        t = np.concatenate([a[0::2], a[1::2]]).astype(np.int32)  # interleave halves
        p = t.copy()
        p[1:] = (t[1:] - t[:-1] + 128 + 256) % 256  # predictor
        z = zlib.compress(p.astype(np.uint8).tobytes())
        return z if len(z) < len(raw) else raw

    def samples(a):
        return (a.astype(np.float16) if half else a.astype(np.float32)).tobytes()

    chunks = []
    if tile:
        tw, th = tile
        coords = [(tx, ty) for ty in range((h + th - 1) // th) for tx in range((w + tw - 1) // tw)]
        if decreasing_y:
            coords = coords[::-1]  # any order: a chunk carries its tile coordinates
        for tx, ty in coords:
            rows, cols = range(ty * th, min((ty + 1) * th, h)), slice(tx * tw, min((tx + 1) * tw, w))
            data = pack(b"".join(samples(img[y, cols, k]) for y in rows for k in order))
            chunks.append(struct.pack("<iiiii", tx, ty, 0, 0, len(data)) + data)
    starts = [] if tile else list(range(0, h, block))
    if decreasing_y:
        starts = starts[::-1]
    for ys in starts:
        rows = range(ys, min(ys + block, h))
        data = pack(b"".join(samples(img[y, :, k]) for y in rows for k in order))
        chunks.append(struct.pack("<ii", y0 + ys, len(data)) + data)
    table_off = len(header)
    offs, pos = [], table_off + 8 * len(chunks)
    for c in chunks:
        offs.append(pos); pos += len(c)
    with open(path, "wb") as f:
        f.write(header + b"".join(struct.pack("<Q", o) for o in offs) + b"".join(chunks))


def decode(path):
    lib = H.load_library()
    w, h, c = C.c_uint32(), C.c_uint32(), C.c_uint32()
    H.check(lib.hala_load_float_image(str(path).encode(), C.byref(w), C.byref(h), C.byref(c), None, C.c_size_t(0)))
    out = np.empty((h.value, w.value, c.value), dtype=f32)
    H.check(lib.hala_load_float_image(str(path).encode(), C.byref(w), C.byref(h), C.byref(c), out.ctypes.data_as(C.POINTER(C.c_float)), C.c_size_t(out.size)))
    return out


@pytest.mark.parametrize("compression", ["none", "zips", "zip"])
@pytest.mark.parametrize("half", [False, True])
def test_exr_scanline_decoding(tmp_path, compression, half):
    rng = np.random.RandomState(3)
    img = (rng.rand(37, 53, 3) * 4).astype(f32)  # 37 rows: a last ZIP block of 5 lines; 53 columns: odd byte counts
    img[5:9, 7:30] = 0.25  # runs for the compressor
    img[0, 0] = (1.0e4, 6.0e-5 if half else 1.0e-20, 0.0)
    p = tmp_path / "a.exr"
    write_exr(str(p), img, compression, half)
    want = img.astype(np.float16).astype(f32) if half else img
    got = decode(p)
    assert got.shape == want.shape and got.tobytes() == want.tobytes()


def test_exr_alpha_luminance_window_and_line_order(tmp_path):
    rng = np.random.RandomState(5)
    rgba = rng.rand(20, 9, 4).astype(f32)
    write_exr(str(tmp_path / "rgba.exr"), rgba, "zip", True, channels="RGBA", data_window_origin=(-3, 11), decreasing_y=True)
    assert decode(tmp_path / "rgba.exr").tobytes() == rgba.astype(np.float16).astype(f32).tobytes()
    y = rng.rand(6, 5, 1).astype(f32)
    write_exr(str(tmp_path / "y.exr"), y, "zips", False, channels="Y")
    got = decode(tmp_path / "y.exr")
    assert got.shape == (6, 5, 3) and np.array_equal(got[..., 0], y[..., 0]) and np.array_equal(got[..., 2], y[..., 0])


@pytest.mark.parametrize("compression,half,tile", [("none", False, (16, 16)), ("zip", True, (32, 8)), ("zips", False, (7, 5)), ("zip", False, (64, 64))])
def test_exr_tiled_decoding(tmp_path, compression, half, tile):
    """single-level tiled OpenEXR (what most tools write for environment maps): edge tiles are narrower / shorter, tiles come in any order"""
    rng = np.random.RandomState(11)
    img = (rng.rand(37, 53, 4) * 3).astype(f32)
    img[10:20, 5:40] = 0.5
    p = tmp_path / "t.exr"
    write_exr(str(p), img, compression, half, channels="RGBA", data_window_origin=(4, -2), decreasing_y=compression == "zip", tile=tile)
    want = img.astype(np.float16).astype(f32) if half else img
    got = decode(p)
    assert got.shape == want.shape and got.tobytes() == want.tobytes()


def test_exr_refusals_and_other_formats(tmp_path):
    img = np.ones((4, 4, 3), f32)
    write_exr(str(tmp_path / "ok.exr"), img, "none", False)
    raw = bytearray((tmp_path / "ok.exr").read_bytes())
    deep = bytearray(raw); deep[5] |= 0x08  # version flag 0x800: deep data
    (tmp_path / "deep.exr").write_bytes(bytes(deep))
    with pytest.raises(H.HalaRendererError, match="Failed to decode image.*not supported"):
        decode(tmp_path / "deep.exr")
    write_exr(str(tmp_path / "mip.exr"), img, "none", False, tile=(2, 2))
    mip = (tmp_path / "mip.exr").read_bytes().replace(struct.pack("<IIB", 2, 2, 0), struct.pack("<IIB", 2, 2, 1))  # mode 1: MIPMAP_LEVELS
    (tmp_path / "mip.exr").write_bytes(mip)
    with pytest.raises(H.HalaRendererError, match="mip-mapped and rip-mapped"):
        decode(tmp_path / "mip.exr")
    huge = bytes(raw).replace(struct.pack("<iiii", 0, 0, 3, 3), struct.pack("<iiii", 0, 0, 2**30, 2**30), 1)  # the data window
    (tmp_path / "huge.exr").write_bytes(huge)
    with pytest.raises(H.HalaRendererError, match="data window too large"):
        decode(tmp_path / "huge.exr")
    piz = bytes(raw).replace(b"compression\0compression\0\x01\0\0\0\x00", b"compression\0compression\0\x01\0\0\0\x04")
    (tmp_path / "piz.exr").write_bytes(piz)
    with pytest.raises(H.HalaRendererError, match="only NONE / RLE / ZIPS / ZIP"):
        decode(tmp_path / "piz.exr")
    with pytest.raises(H.HalaRendererError, match="Failed to open image"):  # src/envmap.rs:49
        decode(tmp_path / "missing.exr")
    (tmp_path / "junk.exr").write_bytes(b"hello world, not an image")
    with pytest.raises(H.HalaRendererError, match="Failed to decode image"):  # :53
        decode(tmp_path / "junk.exr")
    # .pfm written by the library itself decodes to what went in (alpha dropped by the format)
    rgba = np.random.RandomState(1).rand(5, 7, 4).astype(f32)
    H.check(H.load_library().hala_write_pfm(str(tmp_path / "x.pfm").encode(), rgba.ctypes.data_as(C.POINTER(C.c_float)), 7, 5))
    assert np.array_equal(decode(tmp_path / "x.pfm")[..., :3], rgba[..., :3])
