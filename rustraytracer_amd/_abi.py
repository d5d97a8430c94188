"""ctypes mirror of include/rrt.h and the loader of the HIP backend library.

The library is built in-tree by ``__graft_entry__.build()`` (hipcc, gfx950).
There is no CPU fallback: if the shared object is missing or a symbol is
absent, importing the backend fails with an explicit error.
"""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "csrc", "librrt_hip.so")

# enums of include/rrt.h
RRT_OK = 0
PRIM_SPHERE, PRIM_PLANE, PRIM_TRIANGLE, PRIM_QUAD = 0, 1, 2, 3
MAT_EMISSIVE, MAT_LAMBERTIAN, MAT_PHONG, MAT_REFLECTOR, MAT_DIELECTRIC = 0, 1, 2, 3, 4
SAMPLER_RANDOM, SAMPLER_STRATIFIED = 0, 1
PRECISION_F32, PRECISION_F64 = 0, 1
ACCEL_AUTO, ACCEL_LINEAR, ACCEL_LBVH, ACCEL_SAH, ACCEL_SAH_GPU = 0, 1, 2, 3, 4
PIPELINE_AUTO, PIPELINE_FUSED, PIPELINE_WAVEFRONT = 0, 1, 2
RENDER_NEE, RENDER_EXACT_HITS = 1, 2  # rrt_render_flags
INTERSECT_STATIC_MAX_RAYS = 65536

# numpy views of the plain-data structs (bulk construction of large scenes)
PRIM_DTYPE = np.dtype(
    [("type", "<u4"), ("object_id", "<u4"), ("prim_id", "<u4"), ("reserved", "<u4"), ("v", "<f8", (12,))]
)
MATERIAL_DTYPE = np.dtype([("type", "<u4"), ("reserved", "<u4"), ("rgb", "<f8", (3,)), ("param", "<f8")])
assert PRIM_DTYPE.itemsize == 112 and MATERIAL_DTYPE.itemsize == 40


class Camera(C.Structure):
    _fields_ = [
        ("origin", C.c_double * 3),
        ("lower_left_corner", C.c_double * 3),
        ("horizontal", C.c_double * 3),
        ("vertical", C.c_double * 3),
    ]


class SceneDesc(C.Structure):
    _fields_ = [
        ("camera", Camera),
        ("n_prims", C.c_uint32),
        ("n_objects", C.c_uint32),
        ("prims", C.c_void_p),
        ("materials", C.c_void_p),
    ]


class RenderDesc(C.Structure):
    _fields_ = [
        ("width", C.c_uint32),
        ("height", C.c_uint32),
        ("sampler_type", C.c_uint32),
        ("spp", C.c_uint32),
        ("x_strata", C.c_uint32),
        ("y_strata", C.c_uint32),
        ("min_depth", C.c_uint32),
        ("max_depth", C.c_uint32),
        ("sample_begin", C.c_uint64),
        ("sample_end", C.c_uint64),
        ("seed", C.c_uint64),
        ("precision", C.c_uint32),
        ("accel", C.c_uint32),
        ("pipeline", C.c_uint32),
        ("flags", C.c_uint32),
    ]


class PostDesc(C.Structure):
    _fields_ = [
        ("tone_mapping", C.c_uint32),
        ("gamma_correction", C.c_uint32),
        ("has_white_balance", C.c_uint32),
        ("reserved", C.c_uint32),
        ("white_balance", C.c_double * 3),
    ]


class Stats(C.Structure):
    _fields_ = [
        ("samples", C.c_uint64),
        ("rays", C.c_uint64),
        ("prim_tests", C.c_uint64),
        ("node_tests", C.c_uint64),
        ("bounces", C.c_uint64 * 5),
        ("nonfinite", C.c_uint64),
        ("kernel_ms", C.c_double),
        ("reserved", C.c_double),
        ("accel_used", C.c_uint32),
        ("pipeline_used", C.c_uint32),
        ("tree_depth", C.c_uint32),
        ("tree_built", C.c_uint32),
        ("tree_build_ms", C.c_double),
        ("stack_overflows", C.c_uint64),
        ("exact_redos", C.c_uint64),
    ]

    def as_dict(self):
        return {
            "samples": int(self.samples),
            "rays": int(self.rays),
            "prim_tests": int(self.prim_tests),
            "node_tests": int(self.node_tests),
            "bounces": [int(b) for b in self.bounces],
            "nonfinite": int(self.nonfinite),
            "kernel_ms": float(self.kernel_ms),
            "reserved": float(self.reserved),
            "accel_used": int(self.accel_used),
            "pipeline_used": int(self.pipeline_used),
            "tree_depth": int(self.tree_depth),
            "tree_built": int(self.tree_built),
            "tree_build_ms": float(self.tree_build_ms),
            "stack_overflows": int(self.stack_overflows),
            "exact_redos": int(self.exact_redos),
        }


_dp = C.POINTER(C.c_double)
_i32p = C.POINTER(C.c_int32)

# name -> (restype, argtypes): every symbol include/rrt.h declares
SIGNATURES = {
    "rrt_last_error": (C.c_char_p, []),
    "rrt_camera_perspective": (C.c_int, [_dp, _dp, _dp, C.c_double, C.c_double, C.POINTER(Camera)]),
    "rrt_scene_create": (C.c_int, [C.POINTER(SceneDesc), C.POINTER(C.c_void_p)]),
    "rrt_scene_destroy": (None, [C.c_void_p]),
    "rrt_backend_create": (C.c_int, [C.c_char_p, C.c_int, C.POINTER(C.c_void_p)]),
    "rrt_backend_destroy": (None, [C.c_void_p]),
    "rrt_backend_set_stream": (C.c_int, [C.c_void_p, C.c_void_p]),
    "rrt_intersect": (
        C.c_int,
        [C.c_void_p, C.c_void_p, C.c_uint32, C.c_uint32, C.c_size_t, _dp, _dp, _i32p, _i32p, _dp],
    ),
    "rrt_intersect_static": (
        C.c_int,
        [C.c_void_p, C.c_void_p, C.c_uint32, C.c_uint32, C.c_size_t, _dp, _dp, _i32p, _i32p, _dp],
    ),
    "rrt_trace_primary": (
        C.c_int,
        [C.c_void_p, C.c_void_p, C.c_uint32, C.c_uint32, C.c_size_t, _dp, _dp, _i32p, _i32p, _dp],
    ),
    "rrt_render_accumulate": (
        C.c_int,
        [C.c_void_p, C.c_void_p, C.POINTER(RenderDesc), C.c_void_p, C.POINTER(Stats)],
    ),
    "rrt_postprocess": (
        C.c_int,
        [C.c_void_p, C.c_void_p, C.c_uint64, C.POINTER(PostDesc), C.c_uint32, C.c_uint32, C.c_void_p],
    ),
    "rrt_sampler_values": (C.c_int, [C.c_void_p, C.POINTER(RenderDesc), C.c_uint32, C.c_void_p]),
    "rrt_material_probe": (C.c_int, [C.c_void_p, C.c_void_p, C.c_uint32, C.c_size_t, C.c_void_p, C.c_void_p, C.c_void_p,
                                     C.c_void_p, C.c_void_p, C.c_void_p]),
    "rrt_render": (
        C.c_int,
        [C.c_void_p, C.c_void_p, C.POINTER(RenderDesc), C.POINTER(PostDesc), C.c_void_p, C.POINTER(Stats)],
    ),
}

_lib = None


class BackendMissing(RuntimeError):
    pass


def load_library(path=None):
    """dlopen the HIP backend and bind every declared symbol; raises if absent."""
    global _lib
    if _lib is not None and path is None:
        return _lib
    # RRT_HIP_LIBRARY: another build of the same library (kernel experiments)
    p = path or os.environ.get("RRT_HIP_LIBRARY") or LIB_PATH
    if not os.path.exists(p):
        raise BackendMissing(
            f"HIP backend library not found at {p}; run `python -c 'import __graft_entry__ as g; g.build()'` "
            "(there is no CPU fallback)"
        )
    # One process, one HIP runtime: PyTorch wheels bundle their own libamdhip64 / libhsa-runtime64 (same
    # SONAMEs as /opt/rocm's).  If this library pulled in /opt/rocm's copies first, a later `import torch`
    # would bring a second HSA runtime into the process and torch would see no GPU.  Importing torch first
    # makes the dynamic linker resolve this library's libamdhip64.so.7 to the copy that is already loaded.
    try:
        import torch  # noqa: F401
    except ImportError:  # a torch-free process simply uses the system ROCm runtime
        pass
    lib = C.CDLL(p)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)  # AttributeError if the symbol is not exported
        fn.restype = res
        fn.argtypes = args
    if path is None:
        _lib = lib
    return lib


class RrtError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__(f"rrt error {code}: {msg}")
        self.code = code


def check(lib, code):
    if code != RRT_OK:
        raise RrtError(code, (lib.rrt_last_error() or b"").decode("utf-8", "replace"))
