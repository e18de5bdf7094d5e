"""NativeScene — cpu::HalaScene::new(path) done by the library itself (csrc/gltf_loader.cpp behind `hala_scene_load_gltf`):
the C++ restatement of src/scene/loader/gltf_loader.rs.  `HalaRenderer.set_scene` accepts it directly; `desc` exposes the
borrowed `hala_scene_desc` for inspection (tests compare it field by field with the Python mirror gltf_loader.py)."""
import ctypes as C
import os

from . import _abi as A


class NativeScene:
    def __init__(self, path):
        from . import check, load_library
        self._lib = load_library()
        self._lib.hala_scene_get_desc.restype = C.POINTER(A.SceneDesc)
        self._lib.hala_scene_get_desc.argtypes = [C.c_void_p]
        self._lib.hala_scene_free.argtypes = [C.c_void_p]
        h = C.c_void_p()
        check(self._lib.hala_scene_load_gltf(os.fsencode(path), C.byref(h)))
        self._h = h

    def desc_ptr(self):
        return self._lib.hala_scene_get_desc(self._h)

    @property
    def desc(self) -> A.SceneDesc:
        return self.desc_ptr().contents

    def close(self):
        if self._h:
            self._lib.hala_scene_free(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
