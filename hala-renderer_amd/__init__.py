"""hala-renderer_amd — host-side mirror of hala-renderer's ray-tracing renderer API over libhalart.so.

The product path is HIP only: importing the package never needs a GPU (so the CPU test tier can check the
ABI), but every compute entry point lives in libhalart.so and fails loudly if the library or a HIP device
is missing.  There is no CPU fallback anywhere in this package.
"""
import ctypes as _C
import os as _os

from . import _abi  # noqa: F401
from .scene import (HalaScene, HalaNode, HalaMesh, HalaPrimitive, HalaMaterial, HalaMedium, HalaLight,  # noqa: F401
                    HalaPerspectiveCamera, HalaOrthographicCamera, HalaImageData, HalaLightType,
                    HalaMaterialType, HalaMediumType)

_PKG_DIR = _os.path.dirname(_os.path.abspath(__file__))
LIB_PATH = _os.environ.get("HALART_LIB") or _os.path.join(_PKG_DIR, "lib", "libhalart.so")  # HALART_LIB: tuning builds
_lib = None


class HalaRendererError(RuntimeError):
    """reference: src/error.rs:5-22 (msg + optional source)."""

    def message(self):
        return self.args[0] if self.args else ""


def load_library():
    """dlopen libhalart.so (built in-tree by __graft_entry__.build()). Raises if it is missing."""
    global _lib
    if _lib is None:
        if not _os.path.exists(LIB_PATH):
            raise HalaRendererError(
                f"{LIB_PATH} not found: build the HIP library first (python -c 'import __graft_entry__ as g; g.build()')")
        # PyTorch-ROCm ships its own libamdhip64.so.7; whichever copy is loaded first serves the whole process.
        # Importing torch first makes libhalart.so bind to that same HIP runtime, so device memory, streams and
        # RCCL tensors can be shared with torch (loading the system runtime first leaves torch without a GPU).
        try:
            import torch  # noqa: F401
        except ImportError:
            pass
        _lib = _C.CDLL(LIB_PATH)
        _lib.hala_last_error_message.restype = _C.c_char_p
        _lib.hala_version.restype = _C.c_char_p
    return _lib


def last_error() -> str:
    return load_library().hala_last_error_message().decode(errors="replace")


def check(status: int):
    if status != 0:
        raise HalaRendererError(last_error())


from .renderer import HalaRenderer  # noqa: E402,F401
from .raytracing_program import (HalaRayTracingProgram, HalaRayTracingProgramDesc,  # noqa: E402,F401
                                 HalaRayTracingHitShaderDesc)

# reference: src/prelude.rs:17-18
HalaRayTracingRenderer = HalaRenderer
