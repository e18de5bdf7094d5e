"""Test helper: serialises a hala_renderer_amd HalaScene to a self-contained .gltf (base64 buffer, PNG images as data URIs)
using the conventions the reference's loader reads back (extras for materials / lights / cameras)."""
import base64
import io
import json

import numpy as np

from hala_renderer_amd import _abi as A
from hala_renderer_amd.scene import HalaLightType, HalaPerspectiveCamera


def write_gltf(scene, path, with_tangents=True, material_extras=True):
    buf = bytearray()
    views, accessors = [], []

    def add(arr, ctype, atype, target=None):
        arr = np.ascontiguousarray(arr)
        while len(buf) % 4:
            buf.append(0)
        off = len(buf)
        buf.extend(arr.tobytes())
        views.append({"buffer": 0, "byteOffset": off, "byteLength": arr.nbytes, **({"target": target} if target else {})})
        acc = {"bufferView": len(views) - 1, "componentType": ctype, "count": int(arr.shape[0]), "type": atype}
        if atype == "VEC3" and ctype == 5126:
            acc["min"] = arr.min(axis=0).tolist(); acc["max"] = arr.max(axis=0).tolist()
        accessors.append(acc)
        return len(accessors) - 1

    meshes = []
    for mesh in scene.meshes:
        prims = []
        for p in mesh.primitives:
            v = p.vertices
            attr = {"POSITION": add(v["position"].astype(np.float32), 5126, "VEC3", 34962), "NORMAL": add(v["normal"].astype(np.float32), 5126, "VEC3", 34962),
                    "TEXCOORD_0": add(v["tex_coord"].astype(np.float32), 5126, "VEC2", 34962)}
            if with_tangents:
                t4 = np.concatenate([v["tangent"].astype(np.float32), np.ones((len(v), 1), np.float32)], 1)
                attr["TANGENT"] = add(t4, 5126, "VEC4", 34962)
            prim = {"attributes": attr, "indices": add(p.indices.astype(np.uint32).reshape(-1, 1), 5125, "SCALAR", 34963)}
            if p.material_index != A.INVALID_INDEX:
                prim["material"] = int(p.material_index)
            prims.append(prim)
        meshes.append({"primitives": prims})
    materials = []
    for m in scene.materials:
        d = {"pbrMetallicRoughness": {"baseColorFactor": list(m.base_color) + [1.0], "metallicFactor": m.metallic, "roughnessFactor": m.roughness},
             "emissiveFactor": list(m.emission), "extensions": {"KHR_materials_ior": {"ior": m.ior}, "KHR_materials_transmission": {"transmissionFactor": m.specular_transmission}}}
        if m.base_color_map_index != A.INVALID_INDEX:
            d["pbrMetallicRoughness"]["baseColorTexture"] = {"index": m.base_color_map_index}
        if m.metallic_roughness_map_index != A.INVALID_INDEX:
            d["pbrMetallicRoughness"]["metallicRoughnessTexture"] = {"index": m.metallic_roughness_map_index}
        if m.normal_map_index != A.INVALID_INDEX:
            d["normalTexture"] = {"index": m.normal_map_index}
        if m.emission_map_index != A.INVALID_INDEX:
            d["emissiveTexture"] = {"index": m.emission_map_index}
        if material_extras:
            d["extras"] = {"type": m.type, "opacity": m.opacity, "anisotropic": m.anisotropic, "subsurface": m.subsurface, "specular_tint": m.specular_tint,
                           "sheen": m.sheen, "sheen_tint": m.sheen_tint, "clearcoat": m.clearcoat, "clearcoat_roughness": m.clearcoat_roughness,
                           "clearcoat_tint": list(m.clearcoat_tint), "medium_type": m.medium.type, "medium_color": list(m.medium.color),
                           "medium_density": m.medium.density, "medium_anisotropy": m.medium.anisotropy}
        materials.append(d)
    lights = []
    for l in scene.lights:
        d = {"color": list(l.color), "intensity": l.intensity}
        if l.light_type == HalaLightType.DIRECTIONAL:
            d["type"] = "directional"
            d["extras"] = {"param0": float(np.rad2deg(l.params[0]))}  # the loader converts degrees to radians (gltf_loader.rs:461-464)
        elif l.light_type == HalaLightType.SPOT:
            d["type"] = "spot"; d["spot"] = {"innerConeAngle": l.params[0], "outerConeAngle": l.params[1]}
        else:
            d["type"] = "point"
            if l.light_type == HalaLightType.QUAD:  # loader divides the intensity by 0.5*w*h (gltf_loader.rs:472-476)
                d["intensity"] = float(np.float32(l.intensity) * np.float32(np.float32(np.float32(0.5) * np.float32(l.params[0])) * np.float32(l.params[1])))
                d["extras"] = {"type": 1, "param0": l.params[0], "param1": l.params[1]}
            elif l.light_type == HalaLightType.SPHERE:
                d["extras"] = {"type": 2, "param0": l.params[0], "param1": l.params[1]}
        lights.append(d)
    cameras = []
    for c in scene.cameras:
        if isinstance(c, HalaPerspectiveCamera):
            cameras.append({"type": "perspective", "perspective": {"aspectRatio": c.aspect, "yfov": c.yfov, "znear": c.znear, "zfar": c.zfar},
                            "extras": {"focal_dist": c.focal_distance, "aperture": c.aperture}})
        else:
            cameras.append({"type": "orthographic", "orthographic": {"xmag": c.xmag, "ymag": c.ymag, "znear": 0.1, "zfar": 100.0}})
    # nodes: children lists from parents; roots in order
    children = {i: [] for i in range(len(scene.nodes))}
    roots = []
    for i, n in enumerate(scene.nodes):
        (roots if n.parent is None else children[n.parent]).append(i)
    nodes = []
    for i, n in enumerate(scene.nodes):
        d = {"name": n.name, "matrix": np.asarray(n.local_transform, dtype=np.float32).T.reshape(-1).tolist()}
        if children[i]:
            d["children"] = children[i]
        if n.mesh_index != A.INVALID_INDEX:
            d["mesh"] = int(n.mesh_index)
        if n.camera_index != A.INVALID_INDEX:
            d["camera"] = int(n.camera_index)
        if n.light_index != A.INVALID_INDEX:
            d["extensions"] = {"KHR_lights_punctual": {"light": int(n.light_index)}}
        nodes.append(d)
    images, textures = [], []
    from PIL import Image
    for im in scene.image_data:
        bio = io.BytesIO()
        Image.fromarray(np.asarray(im.data, dtype=np.uint8), "RGBA").save(bio, format="PNG")
        images.append({"uri": "data:image/png;base64," + base64.b64encode(bio.getvalue()).decode()})
    for k in sorted(scene.texture2image_mapping):
        textures.append({"source": scene.texture2image_mapping[k]})
    doc = {"asset": {"version": "2.0"}, "scene": 0, "scenes": [{"nodes": roots}], "nodes": nodes, "meshes": meshes, "materials": materials,
           "cameras": cameras, "accessors": accessors, "bufferViews": views,
           "buffers": [{"byteLength": len(buf), "uri": "data:application/octet-stream;base64," + base64.b64encode(bytes(buf)).decode()}],
           "extensions": {"KHR_lights_punctual": {"lights": lights}}, "extensionsUsed": ["KHR_lights_punctual", "KHR_materials_ior", "KHR_materials_transmission"]}
    if images:
        doc["images"] = images; doc["textures"] = textures
    with open(path, "w") as f:
        json.dump(doc, f)
