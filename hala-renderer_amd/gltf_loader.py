"""HalaGltfLoader — host mirror of src/scene/loader/gltf_loader.rs: turns a .gltf file into the
`cpu::HalaScene` model of scene.py, restating the reference's loader rule by rule (each rule cites its lines).  The
reference delegates parsing to the `gltf` crate (`gltf::import`, gltf_loader.rs:123); here the JSON, the buffers
(external .bin or base64 data URIs) and the accessors are decoded with json + numpy, images with PIL.

Supported like the reference: one scene (first one, :128-133), u32-promoted indices, POSITION/NORMAL/TEXCOORD_0 required
(:242-253), optional TANGENT (xyz / w, :255-259) else per-triangle UV tangents (:260-286), pbrMetallicRoughness factors +
textures, KHR_materials_emissive_strength / _transmission / _ior (:335-344), KHR_lights_punctual with the `extras`
quad/sphere convention (:434-487), cameras with `extras` focal_dist / aperture (:492-538).
"""
import base64
import json
import math
import os
import struct
from collections import deque

import numpy as np

from . import _abi as A
from .scene import (HalaImageData, HalaLight, HalaLightType, HalaMaterial, HalaMedium, HalaMesh, HalaNode,
                    HalaOrthographicCamera, HalaPerspectiveCamera, HalaPrimitive, HalaScene)

_COMPONENT = {5120: np.int8, 5121: np.uint8, 5122: np.int16, 5123: np.uint16, 5125: np.uint32, 5126: np.float32}
_NCOMP = {"SCALAR": 1, "VEC2": 2, "VEC3": 3, "VEC4": 4, "MAT2": 4, "MAT3": 9, "MAT4": 16}
FORMAT_SRGB = 1  # HALA_FORMAT_R8G8B8A8_SRGB (gltf_loader.rs:395-396: 8-bit RGB(A) -> *_SRGB)
FORMAT_FLOAT = 2


class GltfError(Exception):
    pass


def _err(msg):
    from . import HalaRendererError
    return HalaRendererError(msg)


class _Doc:
    def __init__(self, path):
        self.dir = os.path.dirname(os.path.abspath(path))
        with open(path, "r") as f:
            self.j = json.load(f)
        self.buffers = [self._load_uri(b["uri"]) for b in self.j.get("buffers", [])]

    def _load_uri(self, uri):
        if uri.startswith("data:"):
            return base64.b64decode(uri.split(",", 1)[1])
        with open(os.path.join(self.dir, uri), "rb") as f:
            return f.read()

    def _elements(self, view, byte_offset, dt, nc, count, strided):
        bv = self.j["bufferViews"][view]
        off = bv.get("byteOffset", 0) + byte_offset
        stride = (bv.get("byteStride", 0) if strided else 0) or dt.itemsize * nc
        buf = self.buffers[bv["buffer"]]
        if stride == dt.itemsize * nc:
            return np.frombuffer(buf, dtype=dt, count=count * nc, offset=off).reshape(count, nc)
        return np.stack([np.frombuffer(buf, dtype=dt, count=nc, offset=off + i * stride) for i in range(count)])

    def accessor(self, idx):
        acc = self.j["accessors"][idx]
        dt = np.dtype(_COMPONENT[acc["componentType"]])
        nc = _NCOMP[acc["type"]]
        count = acc["count"]
        if "bufferView" in acc:
            arr = self._elements(acc["bufferView"], acc.get("byteOffset", 0), dt, nc, count, True)
        elif "sparse" in acc:
            arr = np.zeros((count, nc), dtype=dt)  # glTF 2.0 3.6.2.3: no buffer view -> zeros, then the sparse substitution
        else:
            raise ValueError("Accessor without a buffer view.")
        if "sparse" in acc:
            sp = acc["sparse"]
            ind = self._elements(sp["indices"]["bufferView"], sp["indices"].get("byteOffset", 0), np.dtype(_COMPONENT[sp["indices"]["componentType"]]), 1, sp["count"], False)[:, 0]
            val = self._elements(sp["values"]["bufferView"], sp["values"].get("byteOffset", 0), dt, nc, sp["count"], False)
            arr = np.array(arr)
            arr[ind.astype(np.int64)] = val
        if acc.get("normalized") and dt != np.float32:
            arr = arr.astype(np.float32) / float(np.iinfo(dt).max)
        return arr


def _node_matrix(n):
    """gltf::scene::Transform::matrix(): explicit matrix or T*R*S, column-major"""
    if "matrix" in n:
        return np.array(n["matrix"], dtype=np.float32).reshape(4, 4).T  # -> math convention m[row, col]
    t = np.array(n.get("translation", [0, 0, 0]), dtype=np.float64)
    q = np.array(n.get("rotation", [0, 0, 0, 1]), dtype=np.float64)
    s = np.array(n.get("scale", [1, 1, 1]), dtype=np.float64)
    x, y, z, w = q
    r = np.array([[1 - 2 * (y * y + z * z), 2 * (x * y - z * w), 2 * (x * z + y * w)],
                  [2 * (x * y + z * w), 1 - 2 * (x * x + z * z), 2 * (y * z - x * w)],
                  [2 * (x * z - y * w), 2 * (y * z + x * w), 1 - 2 * (x * x + y * y)]])
    m = np.eye(4)
    m[:3, :3] = r * s[None, :]
    m[:3, 3] = t
    return m.astype(np.float32)


class HalaGltfLoader:
    @staticmethod
    def load(path) -> HalaScene:
        """gltf_loader.rs:121-227"""
        try:
            doc = _Doc(path)
        except Exception as e:  # :123-124
            raise _err(f"Load glTF file \"{path}\" failed.") from e
        j = doc.j
        scenes = j.get("scenes", [])
        if not scenes:
            raise _err(f"No scene in glTF file \"{path}\".")  # :130
        scene = HalaScene()
        # nodes: BFS from the scene roots, parents before children (:134-173) -- what update_node_hierarchies relies on.
        # The reference loops over *all* scenes into one node list (:134) although it warns that only the first is used.
        for sc in scenes:
            queue = deque((None, idx) for idx in sc.get("nodes", []))
            while queue:
                parent, idx = queue.popleft()
                n = j["nodes"][idx]
                cur = len(scene.nodes)
                scene.nodes.append(HalaNode(name=n.get("name", "<Unnamed>"), parent=parent, local_transform=_node_matrix(n),
                                            mesh_index=n.get("mesh", A.INVALID_INDEX), camera_index=n.get("camera", A.INVALID_INDEX),
                                            light_index=n.get("extensions", {}).get("KHR_lights_punctual", {}).get("light", A.INVALID_INDEX)))
                queue.extend((cur, c) for c in n.get("children", []))
        scene.meshes = [HalaGltfLoader.load_mesh(doc, m) for m in j.get("meshes", [])]
        scene.materials = [HalaGltfLoader.load_material(m) for m in j.get("materials", [])]
        for i, t in enumerate(j.get("textures", [])):  # :188-192
            scene.texture2image_mapping[i] = t["source"]
        for i, _ in enumerate(j.get("images", [])):    # :193-197
            scene.image2data_mapping[i] = i
        scene.image_data = [HalaGltfLoader.load_image_data(doc, im) for im in j.get("images", [])]  # :198-201
        scene.lights = [HalaGltfLoader.load_light(l) for l in j.get("extensions", {}).get("KHR_lights_punctual", {}).get("lights", [])]
        scene.cameras = [HalaGltfLoader.load_camera(c) for c in j.get("cameras", [])]
        return scene

    @staticmethod
    def load_mesh(doc, mesh) -> HalaMesh:
        """gltf_loader.rs:232-313"""
        name = mesh.get("name", "<Unnamed>")
        prims = []
        for p in mesh["primitives"]:
            attr = p["attributes"]
            if "indices" not in p:
                raise _err(f"Read indices from mesh \"{name}\" failed.")  # :243
            idx = doc.accessor(p["indices"]).reshape(-1).astype(np.uint32)  # into_u32 :244
            for key, what in (("POSITION", "positions"), ("NORMAL", "normals"), ("TEXCOORD_0", "tex_coords")):
                if key not in attr:
                    raise _err(f"Read {what} from mesh \"{name}\" failed.")  # :246-252
            pos = doc.accessor(attr["POSITION"]).astype(np.float32)
            nrm = doc.accessor(attr["NORMAL"]).astype(np.float32)
            uv = doc.accessor(attr["TEXCOORD_0"]).astype(np.float32)  # into_f32 :253
            if "TANGENT" in attr:
                t4 = doc.accessor(attr["TANGENT"]).astype(np.float32)
                tan = (t4[:, :3] / t4[:, 3:4]).astype(np.float32)  # xyz / w (:255-259)
            else:  # per-triangle UV tangent, last writer wins (:260-286)
                tan = np.zeros_like(pos)
                for tri in idx.reshape(-1, 3):
                    v0, v1, v2 = pos[tri[0]], pos[tri[1]], pos[tri[2]]
                    uv0, uv1, uv2 = uv[tri[0]], uv[tri[1]], uv[tri[2]]
                    dp1, dp2 = (v1 - v0).astype(np.float32), (v2 - v0).astype(np.float32)
                    du1, du2 = (uv1 - uv0).astype(np.float32), (uv2 - uv0).astype(np.float32)
                    with np.errstate(divide="ignore", invalid="ignore"):
                        invdet = np.float32(1.0) / np.float32(np.float32(du1[0] * du2[1]) - np.float32(du1[1] * du2[0]))
                        t = ((dp1 * du2[1]).astype(np.float32) - (dp2 * du1[1]).astype(np.float32)).astype(np.float32) * invdet
                        t = (t / np.float32(np.sqrt(np.float32((t * t).sum(dtype=np.float32))))).astype(np.float32)  # normalize (:280)
                    tan[tri[0]] = tan[tri[1]] = tan[tri[2]] = t
            v = np.zeros(len(pos), dtype=A.VERTEX_DTYPE)
            v["position"], v["normal"], v["tangent"], v["tex_coord"] = pos, nrm, tan, uv
            prims.append(HalaPrimitive(indices=idx, vertices=v, material_index=p.get("material", A.INVALID_INDEX)))  # :298
        return HalaMesh(prims)

    @staticmethod
    def load_material(m) -> HalaMaterial:
        """gltf_loader.rs:318-385 (+ _MaterialCustomInfo :63-114)"""
        pbr = m.get("pbrMetallicRoughness", {})
        ext = m.get("extensions", {})
        if "extras" in m:
            ex = m["extras"]
            if "type" not in ex:  # `type` has no serde default (:65-66): a parse error in the reference
                raise _err("Parse material extras failed.")
            ci = dict(type=ex["type"], opacity=ex.get("opacity", 1.0), anisotropic=ex.get("anisotropic", 0.0), subsurface=ex.get("subsurface", 0.0),
                      specular_tint=ex.get("specular_tint", 0.0), sheen=ex.get("sheen", 0.0), sheen_tint=ex.get("sheen_tint", 0.0),
                      clearcoat=ex.get("clearcoat", 0.0), clearcoat_roughness=ex.get("clearcoat_roughness", 0.0),
                      clearcoat_tint=tuple(ex.get("clearcoat_tint", [0.0, 0.0, 0.0])),  # serde default [0,0,0] when extras exist (:83-84)
                      medium_type=ex.get("medium_type", 0), medium_color=tuple(ex.get("medium_color", [0.0, 0.0, 0.0])),
                      medium_density=ex.get("medium_density", 0.0), medium_anisotropy=ex.get("medium_anisotropy", 0.0))
        else:  # Default impl (:95-113)
            ci = dict(type=0, opacity=1.0, anisotropic=0.0, subsurface=0.0, specular_tint=0.0, sheen=0.0, sheen_tint=0.0, clearcoat=0.0,
                      clearcoat_roughness=0.0, clearcoat_tint=(1.0, 1.0, 1.0), medium_type=0, medium_color=(0.0, 0.0, 0.0), medium_density=0.0,
                      medium_anisotropy=0.0)
        if ci["type"] not in (0, 1):
            raise _err("Invalid material type.")
        emission = np.array(m.get("emissiveFactor", [0, 0, 0]), dtype=np.float32)
        if "KHR_materials_emissive_strength" in ext:  # :336-338
            emission = emission * np.float32(ext["KHR_materials_emissive_strength"].get("emissiveStrength", 1.0))
        tex = lambda d, k: d[k]["index"] if k in d else A.INVALID_INDEX  # noqa: E731  (:346-353)
        return HalaMaterial(
            type=ci["type"], base_color=tuple(pbr.get("baseColorFactor", [1, 1, 1, 1])[:3]), opacity=ci["opacity"], emission=tuple(float(x) for x in emission),
            anisotropic=ci["anisotropic"], metallic=pbr.get("metallicFactor", 1.0), roughness=pbr.get("roughnessFactor", 1.0), subsurface=ci["subsurface"],
            specular_tint=ci["specular_tint"], sheen=ci["sheen"], sheen_tint=ci["sheen_tint"], clearcoat=ci["clearcoat"],
            clearcoat_roughness=ci["clearcoat_roughness"], clearcoat_tint=ci["clearcoat_tint"],
            specular_transmission=ext.get("KHR_materials_transmission", {}).get("transmissionFactor", 0.0) if "KHR_materials_transmission" in ext else 0.0,
            ior=ext.get("KHR_materials_ior", {}).get("ior", 1.5) if "KHR_materials_ior" in ext else 1.5,  # :344
            medium=HalaMedium(ci["medium_type"], ci["medium_color"], ci["medium_density"], ci["medium_anisotropy"]),
            base_color_map_index=tex(pbr, "baseColorTexture"), emission_map_index=tex(m, "emissiveTexture"),
            normal_map_index=tex(m, "normalTexture"), metallic_roughness_map_index=tex(pbr, "metallicRoughnessTexture"))

    @staticmethod
    def load_image_data(doc, im) -> HalaImageData:
        """gltf_loader.rs:391-429: 8-bit RGB is padded to RGBA with alpha 255 and tagged *_SRGB"""
        from PIL import Image
        import io
        if "uri" in im:
            raw = doc._load_uri(im["uri"])
        else:
            bv = doc.j["bufferViews"][im["bufferView"]]
            raw = doc.buffers[bv["buffer"]][bv.get("byteOffset", 0): bv.get("byteOffset", 0) + bv["byteLength"]]
        img = Image.open(io.BytesIO(raw))
        if img.mode in ("F", "I;16", "I"):
            raise _err("Unsupported image format.")
        px = np.array(img.convert("RGBA"), dtype=np.uint8)  # RGB -> RGBA with 255 (:408-416)
        return HalaImageData(FORMAT_SRGB, px.shape[1], px.shape[0], px)

    @staticmethod
    def load_light(l) -> HalaLight:
        """gltf_loader.rs:434-487"""
        color = tuple(l.get("color", [1, 1, 1]))
        intensity = np.float32(l.get("intensity", 1.0))
        kind = l["type"]
        if kind == "directional":
            ltype, p0, p1 = HalaLightType.DIRECTIONAL, 0.0, 0.0
        elif kind == "point":
            ltype, p0, p1 = HalaLightType.POINT, 0.0, 0.0
        else:
            ltype = HalaLightType.SPOT
            p0, p1 = l.get("spot", {}).get("innerConeAngle", 0.0), l.get("spot", {}).get("outerConeAngle", math.pi / 4)
        if "extras" in l:  # :449-459
            ex = l["extras"]
            t = ex.get("type", 0)
            if t == 1:
                ltype = HalaLightType.QUAD
            elif t == 2:
                ltype = HalaLightType.SPHERE
            p0, p1 = ex.get("param0", 0.0), ex.get("param1", 0.0)
        p0, p1 = np.float32(p0), np.float32(p1)
        if ltype == HalaLightType.DIRECTIONAL:  # :461-464
            p0 = np.float32(np.deg2rad(np.float32(min(max(p0, np.float32(0)), np.float32(90)))))
        elif ltype == HalaLightType.SPOT:  # :465-471 (clamp of radians to [0, 90] is the reference's own quirk)
            p0 = np.float32(min(max(p0, np.float32(0)), np.float32(90)))
            p1 = np.float32(min(max(p1, np.float32(0)), np.float32(90)))
            if p0 > p1:
                p0, p1 = p1, p0
        elif ltype == HalaLightType.QUAD:  # :472-476
            intensity = np.float32(intensity / np.float32(np.float32(np.float32(0.5) * p0) * p1))
        return HalaLight(color=color, intensity=float(intensity), light_type=ltype, params=(float(p0), float(p1)))

    @staticmethod
    def load_camera(c):
        """gltf_loader.rs:492-538"""
        if c["type"] == "orthographic":
            o = c["orthographic"]
            return HalaOrthographicCamera(xmag=o["xmag"], ymag=o["ymag"])
        p = c["perspective"]
        ex = c.get("extras")
        focal, aperture = (ex.get("focal_dist", 10.0), ex.get("aperture", 0.0)) if ex is not None else (10.0, 0.0)  # :38-49, :519-525
        return HalaPerspectiveCamera(aspect=p.get("aspectRatio", 1.0), yfov=p["yfov"], znear=p["znear"], zfar=p.get("zfar", 1000.0),  # :511-514
                                     focal_distance=focal, aperture=aperture)


def scene_from_file(path) -> HalaScene:
    """cpu::HalaScene::new (src/scene/cpu/scene.rs:40-55): only `.gltf` is accepted"""
    ext = os.path.splitext(str(path))[1]
    if not ext:
        raise _err(f"Get file \"{path}\" extension failed.")  # :43-44
    if ext != ".gltf":
        raise _err(f"Unsupported file \"{path}\".")  # :49
    return HalaGltfLoader.load(path)
