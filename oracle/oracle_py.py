"""ctypes binding of oracle/liboracle.so (the f64 CPU restatement).

TEST INFRASTRUCTURE: imported only by tests/, __graft_entry__.smoke() and the
cpu_baseline leg of bench.py.  Nothing under rustraytracer_amd/ imports it.
"""
import ctypes as C
import os
import subprocess

import numpy as np

from rustraytracer_amd import _abi

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "liboracle.so")
_dp = C.POINTER(C.c_double)
_i32p = C.POINTER(C.c_int32)
_lib = None


def build():
    subprocess.check_call(["make", "-s", "-C", _HERE, "liboracle.so"])


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            build()
        L = C.CDLL(LIB_PATH)
        L.oracle_fresnel.restype = C.c_double
        L.oracle_fresnel.argtypes = [C.c_double] * 3
        L.oracle_spherical_to_world.argtypes = [C.c_double, C.c_double, _dp, _dp]
        L.oracle_ray_at.argtypes = [_dp, _dp, C.c_double, _dp]
        L.oracle_refract.argtypes = [_dp, _dp, C.c_double, _dp]
        L.oracle_camera_perspective.argtypes = [_dp, _dp, _dp, C.c_double, C.c_double, C.POINTER(_abi.Camera)]
        L.oracle_create_ray.argtypes = [C.POINTER(_abi.Camera), C.c_double, C.c_double, _dp, _dp]
        L.oracle_prim_intersect.argtypes = [C.c_void_p, _dp, _dp, C.c_double, C.c_double, _dp, _dp, _dp]
        L.oracle_sphere_intersect_analytic.argtypes = [_dp, _dp, _dp, C.c_double, C.c_double, _dp]
        L.oracle_material_sample.argtypes = [C.c_void_p, _dp, _dp, _dp, C.c_double, C.c_double, C.c_double,
                                             _dp, _dp, _dp, _dp]
        L.oracle_stratified_tables.argtypes = [C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint64, _dp, _dp]
        L.oracle_intersect.argtypes = [C.POINTER(_abi.SceneDesc), C.c_size_t, _dp, _dp, _i32p, _i32p, _dp]
        L.oracle_trace_primary.argtypes = [C.POINTER(_abi.SceneDesc), C.c_size_t, _dp, _dp, _i32p, _i32p, _dp]
        L.oracle_render_accumulate.argtypes = [C.POINTER(_abi.SceneDesc), C.POINTER(_abi.RenderDesc), _dp,
                                               C.POINTER(_abi.Stats), C.c_int]
        L.oracle_render_accumulate_device_numbers.argtypes = L.oracle_render_accumulate.argtypes + [C.c_int]
        L.oracle_postprocess.argtypes = [_dp, C.c_uint64, C.POINTER(_abi.PostDesc), C.c_uint32, C.c_uint32,
                                         C.POINTER(C.c_uint8), _dp]
        L.oracle_trace_samples.argtypes = [C.POINTER(_abi.SceneDesc), C.POINTER(_abi.RenderDesc), C.c_size_t,
                                           _dp, _dp, _dp]
        L.oracle_accel_build.argtypes = [C.POINTER(_abi.SceneDesc)]
        L.oracle_accel_free.restype = None
        _lib = L
    return _lib


def _d(a):
    a = np.ascontiguousarray(a, dtype=np.float64)
    return a, a.ctypes.data_as(_dp)


def vec3_fn(name, *vecs_and_scalars):
    """Call an oracle function whose last argument is a 3-vector output."""
    out = np.zeros(3)
    args = []
    keep = []
    for a in vecs_and_scalars:
        if np.ndim(a) == 0:
            args.append(C.c_double(float(a)))
        else:
            arr, p = _d(a)
            keep.append(arr)
            args.append(p)
    getattr(lib(), name)(*args, out.ctypes.data_as(_dp))
    return out


def fresnel(cos_i, eta_i, eta_t):
    return lib().oracle_fresnel(cos_i, eta_i, eta_t)


def refract(v, n, eta):
    out = np.zeros(3)
    va, vp = _d(v)
    na, npp = _d(n)
    ok = lib().oracle_refract(vp, npp, eta, out.ctypes.data_as(_dp))
    return out if ok else None


def local_coordinate_system(n):
    u, v, w = np.zeros(3), np.zeros(3), np.zeros(3)
    na, npp = _d(n)
    lib().oracle_local_coordinate_system(npp, *(x.ctypes.data_as(_dp) for x in (u, v, w)))
    return u, v, w


def camera_perspective(look_from, look_at, vup, vfov, aspect):
    cam = _abi.Camera()
    a, ap = _d(look_from)
    b, bp = _d(look_at)
    c, cp = _d(vup)
    lib().oracle_camera_perspective(ap, bp, cp, vfov, aspect, C.byref(cam))
    return cam


def create_ray(cam, s, t):
    o, d = np.zeros(3), np.zeros(3)
    lib().oracle_create_ray(C.byref(cam), s, t, o.ctypes.data_as(_dp), d.ctypes.data_as(_dp))
    return o, d


def make_prim(ptype, values, object_id=0, prim_id=0):
    p = np.zeros(1, dtype=_abi.PRIM_DTYPE)
    p["type"] = ptype
    p["object_id"] = object_id
    p["prim_id"] = prim_id
    vals = np.asarray(values, dtype=np.float64).reshape(-1)
    p["v"][0, : len(vals)] = vals
    return p


def prim_intersect(prim, o, d, t_min, t_max):
    """-> (t, p, normal) or None"""
    t = C.c_double()
    p, n = np.zeros(3), np.zeros(3)
    oa, op = _d(o)
    da, dp = _d(d)
    ok = lib().oracle_prim_intersect(prim.ctypes.data, op, dp, t_min, t_max, C.byref(t),
                                     p.ctypes.data_as(_dp), n.ctypes.data_as(_dp))
    return (t.value, p, n) if ok else None


def sphere_intersect_analytic(sphere4, o, d, t_min, t_max):
    t = C.c_double()
    sa, sp = _d(sphere4)
    oa, op = _d(o)
    da, dp = _d(d)
    ok = lib().oracle_sphere_intersect_analytic(sp, op, dp, t_min, t_max, C.byref(t))
    return t.value if ok else None


def quad_predicates(q12):
    qa, qp = _d(q12)
    return bool(lib().oracle_are_points_coplanar(qp)), bool(lib().oracle_is_quadrilateral_convex(qp))


def material_sample(mat, d_in, p, normal, u1, u2a, u2b):
    """-> dict(d_out, pdf, bxdf, weight) or None"""
    d_out, bxdf, weight = np.zeros(3), np.zeros(3), np.zeros(3)
    pdf = C.c_double()
    a, ap = _d(d_in)
    b, bp = _d(p)
    c, cp = _d(normal)
    ok = lib().oracle_material_sample(mat.ctypes.data, ap, bp, cp, u1, u2a, u2b, d_out.ctypes.data_as(_dp),
                                      C.byref(pdf), bxdf.ctypes.data_as(_dp), weight.ctypes.data_as(_dp))
    return dict(d_out=d_out, pdf=pdf.value, bxdf=bxdf, weight=weight) if ok else None


def stratified_tables(spp, xs, ys, seed):
    o1 = np.zeros((4, spp))
    o2 = np.zeros((4, spp, 2))
    rc = lib().oracle_stratified_tables(spp, xs, ys, seed, o1.ctypes.data_as(_dp), o2.ctypes.data_as(_dp))
    if rc != 0:
        raise ValueError("x_strata * y_strata != samples_per_pixel")
    return o1, o2


def intersect(scene, origins, dirs):
    o, op = _d(origins)
    d, dp = _d(dirs)
    n = len(o.reshape(-1, 3))
    oid = np.zeros(n, dtype=np.int32)
    pid = np.zeros(n, dtype=np.int32)
    t = np.zeros(n)
    desc = scene.desc()
    lib().oracle_intersect(C.byref(desc), n, op, dp, oid.ctypes.data_as(_i32p), pid.ctypes.data_as(_i32p),
                           t.ctypes.data_as(_dp))
    return oid, pid, t


def trace_primary(scene, u, v):
    ua, up = _d(u)
    va, vp = _d(v)
    n = ua.size
    oid = np.zeros(n, dtype=np.int32)
    pid = np.zeros(n, dtype=np.int32)
    t = np.zeros(n)
    desc = scene.desc()
    lib().oracle_trace_primary(C.byref(desc), n, up, vp, oid.ctypes.data_as(_i32p), pid.ctypes.data_as(_i32p),
                               t.ctypes.data_as(_dp))
    return oid, pid, t


class accelerated:
    """with accelerated(scene): the oracle's closest-hit queries on `scene` go through its own bounding-volume hierarchy
    (same results as its scan, tests/test_oracle_accel.py) -- for scenes of many primitives."""

    def __init__(self, scene):
        self._desc = scene.desc()  # keeps the arrays the tree is keyed on alive

    def __enter__(self):
        if lib().oracle_accel_build(C.byref(self._desc)) != 0:
            raise MemoryError("oracle_accel_build failed")
        return self

    def __exit__(self, *exc):
        lib().oracle_accel_free()
        return False


def render_accumulate(scene, render_desc, n_threads=0, device_numbers=False):
    """-> (rgb_sum [H,W,3] f64, stats dict).  device_numbers: draw the random numbers from the restated
    generator of the HIP backend instead of the oracle's own, so that both trace the same paths (True or "f64":
    the numbers of the f64 kernels; "f32": those of the f32 kernels, and their lobe basis)."""
    if n_threads <= 0:
        n_threads = os.cpu_count() or 1
    rgb = np.zeros((render_desc.height, render_desc.width, 3))
    st = _abi.Stats()
    desc = scene.desc()
    args = [C.byref(desc), C.byref(render_desc), rgb.ctypes.data_as(_dp), C.byref(st), n_threads]
    if device_numbers:
        rc = lib().oracle_render_accumulate_device_numbers(*args, 1 if device_numbers == "f32" else 0)
    else:
        rc = lib().oracle_render_accumulate(*args)
    if rc != 0:
        raise ValueError("oracle_render_accumulate rejected the configuration")
    out = st.as_dict()
    out["rays_live"] = int(out.pop("reserved"))  # queries made with non-zero throughput
    return rgb, out


def postprocess(rgb_sum, spp_divisor, post_desc):
    """-> (rgb8 [H,W,3] u8, post-processed float image before quantisation)"""
    rgb_sum = np.ascontiguousarray(rgb_sum, dtype=np.float64)
    h, w = rgb_sum.shape[:2]
    out = np.zeros((h, w, 3), dtype=np.uint8)
    lin = np.zeros((h, w, 3))
    lib().oracle_postprocess(rgb_sum.ctypes.data_as(_dp), spp_divisor, C.byref(post_desc), w, h,
                             out.ctypes.data_as(C.POINTER(C.c_uint8)), lin.ctypes.data_as(_dp))
    return out, lin


def trace_samples(scene, render_desc, u, v):
    ua, up = _d(u)
    va, vp = _d(v)
    rgb = np.zeros((ua.size, 3))
    desc = scene.desc()
    lib().oracle_trace_samples(C.byref(desc), C.byref(render_desc), ua.size, up, vp, rgb.ctypes.data_as(_dp))
    return rgb
