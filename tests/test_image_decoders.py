"""CPU tier: the float-image decoders behind set_envmap(path) (`image::open` of src/envmap.rs:48-53): scanline OpenEXR written
by a small independent writer below (NONE / RLE-free / ZIPS / ZIP; half and float; RGB, RGBA, Y), Radiance .hdr and .pfm."""
import ctypes as C
import struct
import zlib

import numpy as np
import pytest

import hala_renderer_amd as H

f32 = np.float32


# ---- a PIZ encoder for the tests, written from the published description of the format (bitmap range compaction, 2D wavelet, canonical
# Huffman code with a run symbol); no file from another implementation is at hand to pin it: the decoder and this encoder agree with each
# other and with the description, no more is claimed ----
def _piz_wenc14(a, b):
    a = a - 65536 if a >= 32768 else a
    b = b - 65536 if b >= 32768 else b
    m = (a + b) >> 1
    d = a - b
    return m & 0xffff, d & 0xffff


def _piz_wenc16(a, b):
    ao = (a + 0x8000) & 0xffff
    m = (ao + b) >> 1
    d = ao - b
    if d < 0:
        m = (m + 0x8000) & 0xffff
    return m & 0xffff, d & 0xffff


def _piz_wav2_encode(buf, at, nx, ox, ny, oy, mx):
    enc = _piz_wenc14 if mx < (1 << 14) else _piz_wenc16
    n = min(nx, ny)
    p, p2 = 1, 2
    while p2 <= n:
        oy1, oy2, ox1, ox2 = oy * p, oy * p2, ox * p, ox * p2
        py, ey = at, at + oy * (ny - p2)
        while py <= ey:
            px, ex = py, py + ox * (nx - p2)
            while px <= ex:
                p01, p10 = px + ox1, px + oy1
                p11 = p10 + ox1
                i00, i01 = enc(buf[px], buf[p01])
                i10, i11 = enc(buf[p10], buf[p11])
                buf[px], buf[p10] = enc(i00, i10)
                buf[p01], buf[p11] = enc(i01, i11)
                px += ox2
            if nx & p:
                p10 = px + oy1
                buf[px], buf[p10] = enc(buf[px], buf[p10])
            py += oy2
        if ny & p:
            px, ex = py, py + ox * (nx - p2)
            while px <= ex:
                p01 = px + ox1
                buf[px], buf[p01] = enc(buf[px], buf[p01])
                px += ox2
        p, p2 = p2, p2 << 1


class _BitsOut:
    def __init__(self):
        self.bytes = bytearray(); self.c = 0; self.lc = 0

    def put(self, n, v):
        self.c = (self.c << n) | (v & ((1 << n) - 1)); self.lc += n
        while self.lc >= 8:
            self.lc -= 8
            self.bytes.append((self.c >> self.lc) & 0xff)
        self.c &= (1 << self.lc) - 1

    def flush(self):
        if self.lc:
            self.bytes.append((self.c << (8 - self.lc)) & 0xff)
            self.c = 0; self.lc = 0
        return bytes(self.bytes)


def _piz_huf_compress(data):
    import heapq
    freq = np.bincount(np.asarray(data, dtype=np.int64), minlength=65537)
    used = np.nonzero(freq)[0]
    im, iM = int(used[0]), int(used[-1]) + 1  # + the run symbol
    freq[iM] = 1
    heap = [(int(freq[k]), int(k), (int(k),)) for k in np.nonzero(freq)[0]]
    heapq.heapify(heap)
    length = {k[1]: 0 for k in heap}
    while len(heap) > 1:
        fa, ka, sa = heapq.heappop(heap); fb, kb, sb = heapq.heappop(heap)
        for k in sa + sb:
            length[k] += 1
        heapq.heappush(heap, (fa + fb, min(ka, kb), sa + sb))
    assert max(length.values()) <= 58
    n = [0] * 59
    for l in length.values():
        n[l] += 1
    c = 0
    for i in range(58, 0, -1):
        nc = (c + n[i]) >> 1
        n[i] = c
        c = nc
    code = {}
    for k in sorted(length):
        code[k] = (n[length[k]], length[k]); n[length[k]] += 1
    out = _BitsOut()
    k = im
    while k <= iM:  # the table: 6-bit lengths, zero lengths in runs
        l = length.get(k, 0)
        if l == 0:
            z = 1
            while k + z <= iM and z < 255 + 6 and length.get(k + z, 0) == 0:
                z += 1
            if z >= 2:
                if z >= 6:
                    out.put(6, 63); out.put(8, z - 6)
                else:
                    out.put(6, 59 + z - 2)
                k += z
                continue
        out.put(6, l)
        k += 1
    table = out.flush()
    out = _BitsOut()
    nbits = 0

    def send(sym, run):
        nonlocal nbits
        (cs, ls), (cr, lr) = code[sym], code[iM]
        if ls + lr + 8 < ls * run:
            out.put(ls, cs); out.put(lr, cr); out.put(8, run); nbits += ls + lr + 8
        else:
            for _ in range(run + 1):
                out.put(ls, cs); nbits += ls
    s, run = int(data[0]), 0
    for v in data[1:]:
        v = int(v)
        if v == s and run < 255:
            run += 1
        else:
            send(s, run); run = 0
        s = v
    send(s, run)
    body = out.flush()
    return struct.pack("<IIIII", im, iM, len(table), nbits, 0) + table + body


def piz_pack(raw, cols, lines, nch, halves):
    a = np.frombuffer(raw, dtype="<u2").reshape(lines, nch, cols * halves)
    buf = [int(v) for c in range(nch) for v in a[:, c, :].reshape(-1)]
    used = np.zeros(65536, dtype=bool)
    used[np.asarray(buf, dtype=np.int64)] = True
    used[0] = False
    bitmap = np.packbits(used.reshape(-1, 8)[:, ::-1], axis=1).reshape(-1)  # bit i & 7 of byte i >> 3
    nzb = np.nonzero(bitmap)[0]
    mn, mx = (int(nzb[0]), int(nzb[-1])) if len(nzb) else (8191, 0)
    lut = np.zeros(65536, dtype=np.int64)
    k = 0
    for i in range(65536):
        if i == 0 or used[i]:
            lut[i] = k; k += 1
    max_value = k - 1
    buf = [int(lut[v]) for v in buf]
    per = cols * halves * lines
    for c in range(nch):
        for j in range(halves):
            _piz_wav2_encode(buf, c * per + j, cols, halves, lines, cols * halves, max_value)
    huf = _piz_huf_compress(buf)
    out = struct.pack("<HH", mn, mx) + (bitmap[mn:mx + 1].tobytes() if mn <= mx else b"") + struct.pack("<i", len(huf)) + huf
    return out if len(out) < len(raw) else raw



def write_exr(path, img, compression="zip", half=False, channels="RGB", data_window_origin=(0, 0), decreasing_y=False, tile=None):
    """img [H, W, len(channels)] float32.  OpenEXR 2 scanline file — or, tile = (tw, th), a single-level tiled one — channels stored
    alphabetically as the format requires."""
    h, w, nc = img.shape
    assert nc == len(channels)
    order = sorted(range(nc), key=lambda k: channels[k])
    comp = {"none": 0, "rle": 1, "zips": 2, "zip": 3, "piz": 4}[compression]
    block = 16 if comp == 3 else (32 if comp == 4 else 1)
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

    def pack(raw, cols=None, lines=None):
        if not comp:
            return raw
        if comp == 4:
            return piz_pack(raw, cols, lines, nc, 1 if half else 2)
        a = np.frombuffer(raw, dtype=np.uint8)
        t = np.concatenate([a[0::2], a[1::2]]).astype(np.int32)  # interleave halves
        p = t.copy()
        p[1:] = (t[1:] - t[:-1] + 128 + 256) % 256  # predictor
        if comp == 1:  # run-length: count byte n >= 0 -> the next byte n + 1 times; n < 0 -> -n literal bytes
            b, z, i = p.astype(np.uint8).tobytes(), bytearray(), 0
            while i < len(b):
                run = 1
                while i + run < len(b) and run < 128 and b[i + run] == b[i]:
                    run += 1
                if run >= 3:
                    z += bytes([run - 1, b[i]]); i += run
                else:
                    j = i
                    while j < len(b) and j - i < 127 and not (j + 2 < len(b) and b[j] == b[j + 1] == b[j + 2]):
                        j += 1
                    z += bytes([(256 - (j - i)) & 0xff]) + b[i:j]; i = j
            z = bytes(z)
        else:
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
            data = pack(b"".join(samples(img[y, cols, k]) for y in rows for k in order), len(range(*cols.indices(w))), len(rows))
            chunks.append(struct.pack("<iiiii", tx, ty, 0, 0, len(data)) + data)
    starts = [] if tile else list(range(0, h, block))
    if decreasing_y:
        starts = starts[::-1]
    for ys in starts:
        rows = range(ys, min(ys + block, h))
        data = pack(b"".join(samples(img[y, :, k]) for y in rows for k in order), w, len(rows))
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


@pytest.mark.parametrize("compression", ["none", "rle", "zips", "zip", "piz"])
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


def test_exr_piz_16_bit_wavelet_and_flat_blocks(tmp_path):
    """a block with more than 2^14 distinct 16-bit values takes the modulo-2^16 wavelet; a constant block compresses to a bitmap of one
    value and a Huffman stream of run symbols; 45 rows: a last block of 13 lines"""
    rng = np.random.RandomState(11)
    img = np.zeros((45, 400, 3), dtype=f32)
    img[:32] = (rng.rand(32, 400, 3) * 100).astype(f32)  # 32 x 400 x 3 x 2 halves = 76 800 samples, most low halves distinct
    img[32:] = 0.5
    p = tmp_path / "wide.exr"
    write_exr(str(p), img, "piz", False)
    got = decode(p)
    assert got.shape == img.shape and got.tobytes() == img.tobytes()
    half = (rng.rand(45, 400, 3) * 1000).astype(np.float16).astype(f32)  # HALF samples: 38 400 per block, > 2^14 distinct
    write_exr(str(p), half, "piz", True)
    assert decode(p).tobytes() == half.tobytes()


def test_exr_alpha_luminance_window_and_line_order(tmp_path):
    rng = np.random.RandomState(5)
    rgba = rng.rand(20, 9, 4).astype(f32)
    write_exr(str(tmp_path / "rgba.exr"), rgba, "zip", True, channels="RGBA", data_window_origin=(-3, 11), decreasing_y=True)
    assert decode(tmp_path / "rgba.exr").tobytes() == rgba.astype(np.float16).astype(f32).tobytes()
    y = rng.rand(6, 5, 1).astype(f32)
    write_exr(str(tmp_path / "y.exr"), y, "zips", False, channels="Y")
    got = decode(tmp_path / "y.exr")
    assert got.shape == (6, 5, 3) and np.array_equal(got[..., 0], y[..., 0]) and np.array_equal(got[..., 2], y[..., 0])


@pytest.mark.parametrize("compression,half,tile", [("none", False, (16, 16)), ("zip", True, (32, 8)), ("zips", False, (7, 5)), ("zip", False, (64, 64)),
                                                   ("piz", True, (32, 8)), ("piz", False, (7, 5)), ("piz", True, (64, 64))])
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
    b44 = bytes(raw).replace(b"compression\0compression\0\x01\0\0\0\x00", b"compression\0compression\0\x01\0\0\0\x06")  # B44: lossy, not supported
    (tmp_path / "b44.exr").write_bytes(b44)
    with pytest.raises(H.HalaRendererError, match="only NONE / RLE / ZIPS / ZIP / PIZ"):
        decode(tmp_path / "b44.exr")
    with pytest.raises(H.HalaRendererError, match="Failed to open image"):  # src/envmap.rs:49
        decode(tmp_path / "missing.exr")
    (tmp_path / "junk.exr").write_bytes(b"hello world, not an image")
    with pytest.raises(H.HalaRendererError, match="Failed to decode image"):  # :53
        decode(tmp_path / "junk.exr")
    # .pfm written by the library itself decodes to what went in (alpha dropped by the format)
    rgba = np.random.RandomState(1).rand(5, 7, 4).astype(f32)
    H.check(H.load_library().hala_write_pfm(str(tmp_path / "x.pfm").encode(), rgba.ctypes.data_as(C.POINTER(C.c_float)), 7, 5))
    assert np.array_equal(decode(tmp_path / "x.pfm")[..., :3], rgba[..., :3])
