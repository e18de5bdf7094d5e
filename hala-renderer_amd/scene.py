"""Host mirror of the reference's CPU scene model (src/scene/cpu/*.rs) and its marshalling to the C ABI.

Class and field names follow the reference one to one (cpu::HalaScene src/scene/cpu/scene.rs:17-26,
HalaNode node.rs:2-12, HalaMesh/HalaPrimitive mesh.rs:6-19, HalaMaterial material.rs:24-50,
HalaLight light.rs:30-39, HalaCamera camera.rs:4-29) so that code written against `src/scene` reads the
same here.  Matrices are numpy (4,4) float32 in math convention (m[row, col]); `to_desc()` flattens them
column-major like glam::Mat4.
"""
import ctypes as C
from dataclasses import dataclass, field
from typing import Dict, List, Optional

import numpy as np

from . import _abi as A

INVALID = A.INVALID_INDEX


class HalaLightType:  # src/scene/cpu/light.rs:5-12
    POINT, DIRECTIONAL, SPOT, QUAD, SPHERE = 0, 1, 2, 3, 4


class HalaMaterialType:  # src/scene/cpu/material.rs:5-9
    DIFFUSE, DISNEY = 0, 1


class HalaMediumType:  # src/scene/cpu/material.rs:52-58
    NONE, ABSORB, SCATTER, EMISSIVE = 0, 1, 2, 3


@dataclass
class HalaNode:
    name: str = ""
    parent: Optional[int] = None
    local_transform: np.ndarray = field(default_factory=lambda: np.eye(4, dtype=np.float32))
    mesh_index: int = INVALID
    camera_index: int = INVALID
    light_index: int = INVALID


@dataclass
class HalaPrimitive:
    indices: np.ndarray          # uint32 [3*T]
    vertices: np.ndarray         # A.VERTEX_DTYPE [V]
    material_index: int = INVALID


@dataclass
class HalaMesh:
    primitives: List[HalaPrimitive] = field(default_factory=list)


@dataclass
class HalaMedium:
    type: int = HalaMediumType.NONE
    color: tuple = (0.0, 0.0, 0.0)
    density: float = 0.0
    anisotropy: float = 0.0


@dataclass
class HalaMaterial:
    # defaults = loader defaults without extras (src/scene/loader/gltf_loader.rs:95-113, :344)
    type: int = HalaMaterialType.DIFFUSE
    base_color: tuple = (1.0, 1.0, 1.0)
    opacity: float = 1.0
    emission: tuple = (0.0, 0.0, 0.0)
    anisotropic: float = 0.0
    metallic: float = 1.0
    roughness: float = 1.0
    subsurface: float = 0.0
    specular_tint: float = 0.0
    sheen: float = 0.0
    sheen_tint: float = 0.0
    clearcoat: float = 0.0
    clearcoat_roughness: float = 0.0
    clearcoat_tint: tuple = (1.0, 1.0, 1.0)
    specular_transmission: float = 0.0
    ior: float = 1.5
    medium: HalaMedium = field(default_factory=HalaMedium)
    base_color_map_index: int = INVALID
    emission_map_index: int = INVALID
    normal_map_index: int = INVALID
    metallic_roughness_map_index: int = INVALID


@dataclass
class HalaLight:
    color: tuple = (1.0, 1.0, 1.0)
    intensity: float = 1.0
    light_type: int = HalaLightType.POINT
    params: tuple = (0.0, 0.0)


@dataclass
class HalaPerspectiveCamera:
    aspect: float = 1.0
    yfov: float = 0.7
    znear: float = 0.1
    zfar: float = 1000.0           # gltf_loader.rs:514
    focal_distance: float = 10.0   # gltf_loader.rs:38-49
    aperture: float = 0.0


@dataclass
class HalaOrthographicCamera:
    xmag: float = 1.0
    ymag: float = 1.0


@dataclass
class HalaImageData:
    format: int
    width: int
    height: int
    data: np.ndarray


@dataclass
class HalaScene:
    nodes: List[HalaNode] = field(default_factory=list)
    meshes: List[HalaMesh] = field(default_factory=list)
    materials: List[HalaMaterial] = field(default_factory=list)
    texture2image_mapping: Dict[int, int] = field(default_factory=dict)
    image2data_mapping: Dict[int, int] = field(default_factory=dict)
    image_data: List[HalaImageData] = field(default_factory=list)
    lights: List[HalaLight] = field(default_factory=list)
    cameras: list = field(default_factory=list)

    @staticmethod
    def new(path) -> "HalaScene":
        """cpu::HalaScene::new(path) (src/scene/cpu/scene.rs:40-55): glTF only; world transforms are computed by the
        library when the scene is handed to set_scene (update_node_hierarchies, :99-114)."""
        from .gltf_loader import scene_from_file
        return scene_from_file(path)

    # src/scene/cpu/scene.rs:59-95
    def has_light(self) -> bool:
        return len(self.lights) > 0

    def has_medium(self) -> bool:
        return any(m.medium.type != HalaMediumType.NONE for m in self.materials)

    def has_transparent(self) -> bool:
        return any(m.opacity < 1.0 - float(np.finfo(np.float32).eps) for m in self.materials)

    def triangle_count(self) -> int:
        n = 0
        for node in self.nodes:
            if node.mesh_index != INVALID:
                n += sum(len(p.indices) // 3 for p in self.meshes[node.mesh_index].primitives)
        return n

    def to_desc(self) -> "SceneDescHolder":
        return SceneDescHolder(self)


def _f3(t):
    return (C.c_float * 3)(*[float(x) for x in t])


def fill_material_desc(d, m):
    """cpu::HalaMaterial -> the C ABI record (src/scene/cpu/material.rs:24-50)"""
    d.type = m.type
    d.base_color = _f3(m.base_color)
    d.opacity = m.opacity
    d.emission = _f3(m.emission)
    d.anisotropic = m.anisotropic
    d.metallic = m.metallic
    d.roughness = m.roughness
    d.subsurface = m.subsurface
    d.specular_tint = m.specular_tint
    d.sheen = m.sheen
    d.sheen_tint = m.sheen_tint
    d.clearcoat = m.clearcoat
    d.clearcoat_roughness = m.clearcoat_roughness
    d.clearcoat_tint = _f3(m.clearcoat_tint)
    d.specular_transmission = m.specular_transmission
    d.ior = m.ior
    d.medium_type = m.medium.type
    d.medium_color = _f3(m.medium.color)
    d.medium_density = m.medium.density
    d.medium_anisotropy = m.medium.anisotropy
    d.base_color_map_index = m.base_color_map_index
    d.emission_map_index = m.emission_map_index
    d.normal_map_index = m.normal_map_index
    d.metallic_roughness_map_index = m.metallic_roughness_map_index



class SceneDescHolder:
    """Owns the ctypes tree of a hala_scene_desc; keep it alive for the duration of the call."""

    def __init__(self, scene: HalaScene):
        self._keep = []
        n = len(scene.nodes)
        nodes = (A.NodeDesc * max(n, 1))()
        for i, nd in enumerate(scene.nodes):
            name = nd.name.encode()
            self._keep.append(name)
            nodes[i].name = name
            nodes[i].parent = -1 if nd.parent is None else int(nd.parent)
            m = np.asarray(nd.local_transform, dtype=np.float32)
            nodes[i].local_transform = (C.c_float * 16)(*m.T.reshape(-1).tolist())  # column-major
            nodes[i].mesh_index = nd.mesh_index
            nodes[i].camera_index = nd.camera_index
            nodes[i].light_index = nd.light_index
        meshes = (A.MeshDesc * max(len(scene.meshes), 1))()
        for i, mesh in enumerate(scene.meshes):
            prims = (A.PrimitiveDesc * max(len(mesh.primitives), 1))()
            for j, p in enumerate(mesh.primitives):
                idx = np.ascontiguousarray(p.indices, dtype=np.uint32)
                vtx = np.ascontiguousarray(p.vertices, dtype=A.VERTEX_DTYPE)
                self._keep += [idx, vtx]
                prims[j].indices = idx.ctypes.data_as(C.POINTER(C.c_uint32))
                prims[j].index_count = idx.size
                prims[j].vertices = C.cast(vtx.ctypes.data, C.POINTER(A.Vertex))
                prims[j].vertex_count = vtx.size
                prims[j].material_index = p.material_index
            self._keep.append(prims)
            meshes[i].primitives = prims
            meshes[i].primitive_count = len(mesh.primitives)
        mats = (A.MaterialDesc * max(len(scene.materials), 1))()
        for i, m in enumerate(scene.materials):
            fill_material_desc(mats[i], m)
        lights = (A.LightDesc * max(len(scene.lights), 1))()
        for i, l in enumerate(scene.lights):
            lights[i].color = _f3(l.color)
            lights[i].intensity = l.intensity
            lights[i].light_type = l.light_type
            lights[i].param0 = l.params[0]
            lights[i].param1 = l.params[1]
        cams = (A.CameraDesc * max(len(scene.cameras), 1))()
        for i, c in enumerate(scene.cameras):
            if isinstance(c, HalaPerspectiveCamera):
                cams[i].type = 0
                cams[i].aspect, cams[i].yfov, cams[i].znear, cams[i].zfar = c.aspect, c.yfov, c.znear, c.zfar
                cams[i].focal_distance, cams[i].aperture = c.focal_distance, c.aperture
            else:
                cams[i].type = 1
                cams[i].xmag, cams[i].ymag = c.xmag, c.ymag
        t2i = (A.IndexPair * max(len(scene.texture2image_mapping), 1))()
        for i, (k, v) in enumerate(sorted(scene.texture2image_mapping.items())):  # BTreeMap order
            t2i[i].key, t2i[i].value = k, v
        i2d = (A.IndexPair * max(len(scene.image2data_mapping), 1))()
        for i, (k, v) in enumerate(sorted(scene.image2data_mapping.items())):
            i2d[i].key, i2d[i].value = k, v
        imgs = (A.ImageDesc * max(len(scene.image_data), 1))()
        for i, im in enumerate(scene.image_data):
            data = np.ascontiguousarray(im.data)
            self._keep.append(data)
            imgs[i].format, imgs[i].width, imgs[i].height = im.format, im.width, im.height
            imgs[i].data = data.ctypes.data
            imgs[i].num_of_bytes = data.nbytes
        self._keep += [nodes, meshes, mats, lights, cams, t2i, i2d, imgs]
        d = A.SceneDesc()
        d.nodes, d.node_count = nodes, n
        d.meshes, d.mesh_count = meshes, len(scene.meshes)
        d.materials, d.material_count = mats, len(scene.materials)
        d.lights, d.light_count = lights, len(scene.lights)
        d.cameras, d.camera_count = cams, len(scene.cameras)
        d.texture2image_mapping, d.texture_count = t2i, len(scene.texture2image_mapping)
        d.image2data_mapping, d.image_count = i2d, len(scene.image2data_mapping)
        d.image_data, d.image_data_count = imgs, len(scene.image_data)
        self.desc = d

    def ptr(self):
        return C.byref(self.desc)
