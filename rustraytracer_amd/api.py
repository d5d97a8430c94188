"""Python driver of the HIP backend through its C ABI (include/rrt.h).

Mirrors the reference's call structure (file:line in nero19960329/RustRayTracer):
    Scene::from_config   src/scene.rs:20     -> Scene(flat_scene)
    render(&cfg, &scene) src/renderer.rs:73  -> Backend.render(scene, render_config)
and adds the finer-grained entry points a device backend exposes
(closest-hit queries, radiance accumulation for a sample range, post-process).
PyTorch is optional plumbing: tensors are passed as raw device pointers.
"""
import ctypes as C

import numpy as np

from . import _abi
from ._abi import (ACCEL_AUTO, ACCEL_LBVH, ACCEL_LINEAR, ACCEL_SAH, ACCEL_SAH_GPU, PIPELINE_AUTO, PIPELINE_FUSED, PIPELINE_WAVEFRONT, RENDER_NEE, RENDER_EXACT_HITS,  # noqa: F401
                   PRECISION_F32, PRECISION_F64, RrtError)

_dp = C.POINTER(C.c_double)
_i32p = C.POINTER(C.c_int32)


class Scene:
    """Owns an rrt_scene handle built from a config.FlatScene."""

    def __init__(self, flat):
        self._lib = _abi.load_library()
        self.flat = flat
        desc = flat.desc()
        h = C.c_void_p()
        _abi.check(self._lib, self._lib.rrt_scene_create(C.byref(desc), C.byref(h)))
        self.handle = h

    def close(self):
        if getattr(self, "handle", None):
            self._lib.rrt_scene_destroy(self.handle)
            self.handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def _ptr(x):
    """Raw address of a numpy array, a torch tensor or an int."""
    if isinstance(x, np.ndarray):
        return C.c_void_p(x.ctypes.data)
    if hasattr(x, "data_ptr"):
        return C.c_void_p(x.data_ptr())
    return C.c_void_p(int(x))


class Backend:
    """One GPU (rrt_backend).  There is no CPU fallback."""

    def __init__(self, device_id=0, name="hip"):
        self._lib = _abi.load_library()
        h = C.c_void_p()
        _abi.check(self._lib, self._lib.rrt_backend_create(name.encode(), int(device_id), C.byref(h)))
        self.handle = h
        self.device_id = device_id

    def close(self):
        if getattr(self, "handle", None):
            self._lib.rrt_backend_destroy(self.handle)
            self.handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def set_stream(self, hip_stream):
        """hip_stream: integer hipStream_t (e.g. torch.cuda.current_stream().cuda_stream); 0 = default."""
        _abi.check(self._lib, self._lib.rrt_backend_set_stream(self.handle, C.c_void_p(int(hip_stream))))

    # -- closest hit ---------------------------------------------------------
    def _closest(self, fn, scene, a, b, precision, accel, per):
        a = np.ascontiguousarray(a, dtype=np.float64)
        b = np.ascontiguousarray(b, dtype=np.float64)
        if a.size != b.size or a.size % per:
            raise ValueError("mismatched ray arrays")
        n = a.size // per
        oid = np.empty(n, dtype=np.int32)
        pid = np.empty(n, dtype=np.int32)
        t = np.empty(n, dtype=np.float64)
        _abi.check(self._lib, fn(self.handle, scene.handle, precision, accel, n, a.ctypes.data_as(_dp),
                                 b.ctypes.data_as(_dp), oid.ctypes.data_as(_i32p), pid.ctypes.data_as(_i32p),
                                 t.ctypes.data_as(_dp)))
        return oid, pid, t

    def intersect(self, scene, origins, dirs, precision=PRECISION_F32, accel=ACCEL_AUTO):
        """Scene::intersect for n rays -> (object_id, prim_id, t); misses give -1, -1, inf."""
        return self._closest(self._lib.rrt_intersect, scene, origins, dirs, precision, accel, 3)

    def intersect_static(self, scene, origins, dirs, precision=PRECISION_F32, accel=ACCEL_AUTO):
        """rrt_intersect_static: Scene::intersect for at most _abi.INTERSECT_STATIC_MAX_RAYS rays without any
        device allocation (the scene must already be resident: call intersect once first)."""
        return self._closest(self._lib.rrt_intersect_static, scene, origins, dirs, precision, accel, 3)

    def trace_primary(self, scene, u, v, precision=PRECISION_F32, accel=ACCEL_AUTO):
        """Camera::create_ray(u, v) + Scene::intersect."""
        return self._closest(self._lib.rrt_trace_primary, scene, u, v, precision, accel, 1)

    # -- rendering -----------------------------------------------------------
    def render_accumulate(self, scene, render_desc, rgb_sum=None):
        """Adds the radiance of render_desc's sample range to rgb_sum.

        rgb_sum: None (a zeroed host array is created and returned), a float32
        numpy array [H, W, 3], or a float32 CUDA/HIP torch tensor of that shape.
        Returns (rgb_sum, stats dict).
        """
        if rgb_sum is None:
            rgb_sum = np.zeros((render_desc.height, render_desc.width, 3), dtype=np.float32)
        if isinstance(rgb_sum, np.ndarray):
            if rgb_sum.dtype != np.float32 or not rgb_sum.flags.c_contiguous:
                raise ValueError("rgb_sum must be a C-contiguous float32 array")
            if rgb_sum.size != render_desc.height * render_desc.width * 3:
                raise ValueError("rgb_sum has the wrong size")
        elif hasattr(rgb_sum, "data_ptr"):
            self._check_tensor(rgb_sum, "float32", render_desc.height * render_desc.width * 3, "rgb_sum")
        st = _abi.Stats()
        _abi.check(self._lib, self._lib.rrt_render_accumulate(self.handle, scene.handle, C.byref(render_desc),
                                                               _ptr(rgb_sum), C.byref(st)))
        return rgb_sum, st.as_dict()

    def _check_tensor(self, t, dtype, numel, what):
        """A torch tensor handed to the library by address must be what the kernels assume: dtype, dense layout, size,
        and (if it lives on a GPU) this backend's GPU."""
        if str(t.dtype) != "torch." + dtype:
            raise ValueError("%s must be a %s tensor, not %s" % (what, dtype, t.dtype))
        if not t.is_contiguous():
            raise ValueError("%s must be contiguous" % what)
        if t.numel() != numel:
            raise ValueError("%s has %d elements, expected %d" % (what, t.numel(), numel))
        if t.device.type == "cuda" and t.device.index != self.device_id:
            raise ValueError("%s is on %s, the backend renders on GPU %d" % (what, t.device, self.device_id))

    def sampler_values(self, render_desc, pixel):
        """Device sampler output for every sample index of the desc's range: float32 [n, 8]
        = get_2d, get_1d, get_2d, get_1d (the four stratified dimensions), get_2d (uniform)."""
        n = int(render_desc.sample_end - render_desc.sample_begin)
        out = np.empty((n, 8), dtype=np.float32)
        _abi.check(self._lib, self._lib.rrt_sampler_values(self.handle, C.byref(render_desc), int(pixel), _ptr(out)))
        return out

    def material_probe(self, material, d_in, normal, u, precision=PRECISION_F32):
        """Material::scatter + throughput factor for n samples of one material (a MATERIAL_DTYPE record):
        d_in, normal [n, 3]; u [n, 3] = (get_1d, get_2d.x, get_2d.y) -> (d_out [n, 3], weight [n, 3], ok [n])."""
        mat = np.ascontiguousarray(np.asarray(material, dtype=_abi.MATERIAL_DTYPE).reshape(1))
        d_in, normal, u = (np.ascontiguousarray(a, dtype=np.float64).reshape(-1, 3) for a in (d_in, normal, u))
        n = d_in.shape[0]
        if normal.shape[0] != n or u.shape[0] != n:
            raise ValueError("d_in, normal and u must have the same length")
        d_out, weight = np.zeros((n, 3)), np.zeros((n, 3))
        ok = np.zeros(n, dtype=np.int32)
        _abi.check(self._lib, self._lib.rrt_material_probe(self.handle, _ptr(mat), int(precision), n, _ptr(d_in), _ptr(normal),
                                                           _ptr(u), _ptr(d_out), _ptr(weight), _ptr(ok)))
        return d_out, weight, ok.astype(bool)

    def postprocess(self, rgb_sum, spp_divisor, post_desc, out=None):
        """mean -> tone map -> gamma -> white balance -> u8.  Returns uint8 [H, W, 3]."""
        if isinstance(rgb_sum, np.ndarray):
            rgb_sum = np.ascontiguousarray(rgb_sum, dtype=np.float32)
        h, w = int(rgb_sum.shape[0]), int(rgb_sum.shape[1])
        if hasattr(rgb_sum, "data_ptr"):
            self._check_tensor(rgb_sum, "float32", h * w * 3, "rgb_sum")
        if out is None:
            out = np.empty((h, w, 3), dtype=np.uint8)
        elif isinstance(out, np.ndarray):
            if out.dtype != np.uint8 or not out.flags.c_contiguous or out.size != h * w * 3:
                raise ValueError("out must be a C-contiguous uint8 array of %d elements" % (h * w * 3))
        elif hasattr(out, "data_ptr"):
            self._check_tensor(out, "uint8", h * w * 3, "out")
        _abi.check(self._lib, self._lib.rrt_postprocess(self.handle, _ptr(rgb_sum), int(spp_divisor),
                                                         C.byref(post_desc), w, h, _ptr(out)))
        return out

    def render(self, scene, render_config, seed=None, precision=PRECISION_F32, accel=ACCEL_AUTO,
               pipeline=PIPELINE_AUTO):
        """render(&RenderConfig, &Scene) -> RgbImage (src/renderer.rs:73): uint8 [H, W, 3] and stats."""
        d = render_config.render_desc(seed=seed, precision=precision, accel=accel, pipeline=pipeline)
        p = render_config.post_desc()
        out = np.empty((d.height, d.width, 3), dtype=np.uint8)
        st = _abi.Stats()
        _abi.check(self._lib, self._lib.rrt_render(self.handle, scene.handle, C.byref(d), C.byref(p),
                                                    _ptr(out), C.byref(st)))
        return out, st.as_dict()


def camera_perspective(look_from, look_at, vup, vfov, aspect):
    """PerspectiveCamera::new through the C ABI -> _abi.Camera."""
    lib = _abi.load_library()
    cam = _abi.Camera()
    a = np.ascontiguousarray(look_from, dtype=np.float64)
    b = np.ascontiguousarray(look_at, dtype=np.float64)
    c = np.ascontiguousarray(vup, dtype=np.float64)
    _abi.check(lib, lib.rrt_camera_perspective(a.ctypes.data_as(_dp), b.ctypes.data_as(_dp), c.ctypes.data_as(_dp),
                                               float(vfov), float(aspect), C.byref(cam)))
    return cam
