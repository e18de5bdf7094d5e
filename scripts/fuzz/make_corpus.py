#!/usr/bin/env python3
"""Mutation corpus for scripts/fuzz/run.sh: sequential + progressive JPEGs, glTF documents (byte-level and structural mutations),
PNGs of every colour type, PGM / PPM and TGA files embedded in a glTF.  usage: make_corpus.py <outdir>"""
import base64, copy, io, json, os, sys
import numpy as np
from PIL import Image
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from hala_renderer_amd import scenes  # noqa: E402
from gltf_writer import write_gltf  # noqa: E402

out = sys.argv[1]
for d in ("jpeg", "gltf", "png", "image"):
    os.makedirs(os.path.join(out, d), exist_ok=True)
rs = np.random.RandomState(1)


def mutate(base, lo):
    b = bytearray(base)
    for _ in range(rs.randint(1, 4)):
        pos = rs.randint(lo, len(b)); mode = rs.randint(3)
        if mode == 0: b[pos] = rs.randint(256)
        elif mode == 1: b = b[:pos]
        else: b[pos:pos] = bytes(rs.randint(0, 256, rs.randint(1, 5)).tolist())
        if len(b) < lo + 2: break
    return bytes(b)


n = 0
for variant, (w, h) in enumerate([(64, 48), (61, 37), (3, 5), (40, 40), (17, 9), (128, 16)]):
    img = rs.randint(0, 256, (h, w, 3)).astype(np.uint8) if variant % 2 else (np.indices((h, w)).sum(0)[..., None] * np.array([1, 2, 3]) % 256).astype(np.uint8)
    for prog in (False, True):
        for sub in (0, 2):
            for grey in (False, True):
                bio = io.BytesIO(); pil = Image.fromarray(img, "RGB"); pil = pil.convert("L") if grey else pil
                pil.save(bio, format="JPEG", quality=80, subsampling=sub, progressive=prog, **({"restart_marker_blocks": 2} if variant == 3 else {}))
                for k in range(41):
                    open(os.path.join(out, "jpeg", f"{n:05d}.jpg"), "wb").write(bio.getvalue() if k == 0 else mutate(bio.getvalue(), 2)); n += 1
s = scenes.cornell_box(); scenes.attach_textures(s, sets=1, size=16)
base = os.path.join(out, "gltf", "base.gltf")
write_gltf(s, base)
doc = json.load(open(base)); raw = open(base, "rb").read()
for k in range(400):
    b = bytearray(raw)
    for _ in range(rs.randint(1, 4)):
        pos = rs.randint(0, len(b)); mode = rs.randint(3)
        if mode == 0: b[pos] = rs.randint(32, 127)
        elif mode == 1: del b[pos:pos + rs.randint(1, 20)]
        else: b[pos:pos] = bytes(rs.randint(32, 127, rs.randint(1, 6)).tolist())
    open(os.path.join(out, "gltf", f"m{k:04d}.gltf"), "wb").write(b)


def leaves(o, path=()):
    if isinstance(o, dict):
        for k, v in o.items(): yield from leaves(v, path + (k,))
    elif isinstance(o, list):
        for i, v in enumerate(o): yield from leaves(v, path + (i,))
    else: yield path, o


numeric = [p for p, v in leaves(doc) if isinstance(v, (int, float)) and not isinstance(v, bool)]
hostile = [-1, 0, 1, 2**31 - 1, 2**31, 2**32 - 1, 2**32, 2**53, -2**31, 1e30, -1e30, 0.5, 1e-30, 3.5]
for k in range(600):
    d = copy.deepcopy(doc)
    for _ in range(rs.randint(1, 3)):
        p = numeric[rs.randint(len(numeric))]; o = d
        for key in p[:-1]: o = o[key]
        o[p[-1]] = hostile[rs.randint(len(hostile))]
    json.dump(d, open(os.path.join(out, "gltf", f"s{k:04d}.gltf"), "w"))
# sparse accessors (glTF 2.0 3.6.2.3): the POSITION accessor of the first primitive as zeros + a substitution of 5 vertices, the NORMAL accessor with two
# elements overridden; then hostile values in every numeric leaf (counts, offsets, component types of the sparse blocks among them)
sd = copy.deepcopy(doc)
prim0 = sd["meshes"][0]["primitives"][0]
pa, na = sd["accessors"][prim0["attributes"]["POSITION"]], sd["accessors"][prim0["attributes"]["NORMAL"]]
blob = np.array([0, 1, 2, 3, 5], dtype=np.uint16).tobytes() + b"\0\0" + rs.rand(5, 3).astype(np.float32).tobytes() + bytes([0, 2, 0, 0]) + rs.rand(2, 3).astype(np.float32).tobytes()
sd["buffers"].append({"byteLength": len(blob), "uri": "data:application/octet-stream;base64," + base64.b64encode(blob).decode()})
bi, v0 = len(sd["buffers"]) - 1, len(sd["bufferViews"])
sd["bufferViews"] += [{"buffer": bi, "byteOffset": 0, "byteLength": 10}, {"buffer": bi, "byteOffset": 12, "byteLength": 60},
                      {"buffer": bi, "byteOffset": 72, "byteLength": 2}, {"buffer": bi, "byteOffset": 76, "byteLength": 24}]
pa.pop("bufferView", None); pa.pop("byteOffset", None)
pa["sparse"] = {"count": 5, "indices": {"bufferView": v0, "componentType": 5123}, "values": {"bufferView": v0 + 1}}
na["sparse"] = {"count": 2, "indices": {"bufferView": v0 + 2, "componentType": 5121}, "values": {"bufferView": v0 + 3}}
json.dump(sd, open(os.path.join(out, "gltf", "sparse_base.gltf"), "w"))
snum = [p for p, v in leaves(sd) if isinstance(v, (int, float)) and not isinstance(v, bool)]
sparse_first = [p for p in snum if "sparse" in p] * 6 + snum  # the sparse blocks' own numbers six times as often
for k in range(400):
    d = copy.deepcopy(sd)
    for _ in range(rs.randint(1, 3)):
        p = sparse_first[rs.randint(len(sparse_first))]; o = d
        for key in p[:-1]: o = o[key]
        o[p[-1]] = hostile[rs.randint(len(hostile))]
    json.dump(d, open(os.path.join(out, "gltf", f"p{k:04d}.gltf"), "w"))
n = 0
for mode, size, ch in (("RGB", (17, 9), 3), ("RGBA", (16, 16), 4), ("L", (5, 7), 1), ("LA", (8, 3), 2), ("P", (12, 12), 1), ("I;16", (6, 6), 1)):
    arr = rs.randint(0, 256, (size[1], size[0], ch)).astype(np.uint8)
    if mode == "P": im = Image.fromarray(arr[..., 0], "L").convert("P")
    elif mode == "I;16": im = Image.fromarray(arr[..., 0].astype(np.uint16) * 257)
    elif mode == "L": im = Image.fromarray(arr[..., 0], "L")
    else: im = Image.fromarray(arr, mode)
    bio = io.BytesIO(); im.save(bio, format="PNG")
    for k in range(120):
        b = bio.getvalue() if k == 0 else mutate(bio.getvalue(), 8)
        d = json.loads(json.dumps(doc)); d["images"][0] = {"uri": "data:image/png;base64," + base64.b64encode(b).decode()}
        json.dump(d, open(os.path.join(out, "png", f"p{n:04d}.gltf"), "w")); n += 1

# PGM / PPM (binary + plain) and TGA (raw, run-length, colour-mapped, 32-bit) embedded the same way
seeds = []
g = rs.randint(0, 256, (7, 11, 3)).astype(np.uint8); g[:, 2:8] = g[:, 2:3]
seeds.append(b"P6\n11 7\n255\n" + g.tobytes())
seeds.append(b"P5\n# c\n11 7\n255\n" + g[..., 0].tobytes())
seeds.append(b"P3\n11 7\n255\n" + b" ".join(str(int(v)).encode() for v in g.reshape(-1)) + b"\n")
seeds.append(b"P2\n11 7\n255\n" + b" ".join(str(int(v)).encode() for v in g[..., 0].reshape(-1)) + b"\n")
for im, kw in ((Image.fromarray(g, "RGB"), {}), (Image.fromarray(g, "RGB"), dict(compression="tga_rle")), (Image.fromarray(g[..., 0], "L"), dict(compression="tga_rle")),
               (Image.fromarray(g, "RGB").quantize(16), {}), (Image.fromarray(g, "RGB").quantize(16), dict(compression="tga_rle")),
               (Image.fromarray(np.dstack([g, g[..., :1]]), "RGBA"), dict(orientation=1))):
    bio = io.BytesIO(); im.save(bio, format="TGA", **kw); seeds.append(bio.getvalue())
from test_gltf_native import _png_bytes  # noqa: E402  (Adam7 writer)
for ctype, depth, ch in ((2, 8, 3), (0, 2, 1), (3, 4, 1), (6, 16, 4)):
    seeds.append(_png_bytes(rs.randint(0, 1 << depth, (9, 13, ch)), ctype, depth, True, rs.randint(0, 256, (1 << depth, 3)) if ctype == 3 else None))
for b0 in seeds:
    for k in range(120):
        b = b0 if k == 0 else mutate(b0, 0 if k % 3 else 18)  # two thirds of the mutations may hit the header
        d = json.loads(json.dumps(doc)); d["images"][0] = {"uri": "data:application/octet-stream;base64," + base64.b64encode(b).decode()}
        json.dump(d, open(os.path.join(out, "png", f"p{n:04d}.gltf"), "w")); n += 1

# float images (set_envmap(path)): OpenEXR scanline + tiled in every supported compression (NONE / ZIPS / ZIP / PIZ), .hdr, .pfm, each with byte-level mutations
from test_image_decoders import write_exr  # noqa: E402
n = 0
img = (rs.rand(21, 34, 4) * 3).astype(np.float32)
variants = [dict(compression=c, half=hf, tile=t) for c in ("none", "rle", "zips", "zip", "piz") for hf in (False, True) for t in (None, (8, 8), (16, 5))]
for v in variants:
    p0 = os.path.join(out, "image", "base.exr")
    write_exr(p0, img, v["compression"], v["half"], channels="RGBA", tile=v["tile"])
    raw = open(p0, "rb").read()
    for k in range(120 if v["compression"] == "piz" else 40):  # the PIZ blocks get three times the mutations: bitmap, code table, code, wavelet
        open(os.path.join(out, "image", f"e{n:05d}.exr"), "wb").write(raw if k == 0 else mutate(raw, 8 if k % 2 else 330)); n += 1
os.remove(os.path.join(out, "image", "base.exr"))
hdr = b"#?RADIANCE\nFORMAT=32-bit_rle_rgbe\n\n-Y 6 +X 9\n" + rs.randint(0, 256, 6 * 9 * 4).astype(np.uint8).tobytes()
pfm = b"PF\n7 5\n-1.0\n" + rs.rand(5 * 7 * 3).astype(np.float32).tobytes()
for base, ext in ((hdr, "hdr"), (pfm, "pfm")):
    for k in range(150):
        open(os.path.join(out, "image", f"x{n:05d}.{ext}"), "wb").write(base if k == 0 else mutate(base, 2)); n += 1
