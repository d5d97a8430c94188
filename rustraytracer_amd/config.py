"""Scene / render configuration: the reference's TOML schema, flattened for the C ABI.

Mirrors the serde structs of the reference (file:line in nero19960329/RustRayTracer):
  SceneConfig / ObjectConfig      src/scene.rs:13-17, src/object.rs:26-39
  CameraConfig (Perspective)      src/camera.rs:48-75
  ShapeConfig  (Sphere / Plane / Triangle / Quadrilateral / Mesh)   src/shapes/shape.rs:16-36
  MaterialConfig (5 variants)     src/material.rs:286-315
  RenderConfig and its tables     src/renderer.rs:12-37
  SamplerConfig / TracerConfig    src/sampler.rs:168-187, src/tracers/tracer.rs:11-24
  Matrix4DConfig (column-major)   src/math.rs:36-63
Unknown keys are ignored, as serde does without deny_unknown_fields.

The flattened form lists one rrt_prim per primitive in the order the
reference's linear scan visits them, with every `transform` already applied
the way each shape's `transform` method applies it.
"""
import os
import struct

import numpy as np

try:  # Python >= 3.11
    import tomllib as _toml
except ModuleNotFoundError:  # pragma: no cover
    import tomli as _toml

from . import _abi


class ConfigError(ValueError):
    """A config the reference would reject (serde error, unwrap/expect/panic)."""


def _as_config_error(what):
    """Decorator: whatever a loader trips over in a malformed file (a missing key, a short buffer, bytes that are not
    UTF-8, a count no file could hold) leaves it as ConfigError -- one kind of failure, as every such input is one panic
    in the reference."""
    import functools

    def wrap(fn):
        @functools.wraps(fn)
        def inner(path, *a, **kw):
            try:
                return fn(path, *a, **kw)
            except ConfigError:
                raise
            except (KeyError, IndexError, ValueError, TypeError, OverflowError, MemoryError, OSError, UnicodeError, AttributeError) as e:
                raise ConfigError(f"{what} {path}: {type(e).__name__}: {e}") from e
        return inner
    return wrap


# ----------------------------------------------------------------------------- math.rs
def _vec3(t, what):
    try:
        return np.array([float(t["x"]), float(t["y"]), float(t["z"])], dtype=np.float64)
    except (KeyError, TypeError) as e:
        raise ConfigError(f"{what}: expected inline table {{x, y, z}}") from e


def matrix_from_config(t):
    """Matrix4DConfig::to_matrix (src/math.rs:56-63): m<C><R> feeds cgmath's
    column-major Matrix4::new, so m41..m43 is the translation.  Returns the
    conventional 4x4 array M with M @ [x, y, z, 1] = transformed point."""
    if t is None:
        return np.eye(4)
    cols = []
    for c in range(1, 5):
        try:
            cols.append([float(t[f"m{c}{r}"]) for r in range(1, 5)])
        except KeyError as e:
            raise ConfigError(f"transform: missing field {e}") from e
    return np.array(cols, dtype=np.float64).T


def transform_point3(m, p):
    """src/math.rs:108-110 (homogeneous divide, src/math.rs:103-106)."""
    p = np.asarray(p, dtype=np.float64)
    v = p @ m[:3, :3].T + m[:3, 3]
    w = p @ m[3, :3] + m[3, 3]
    if np.any(w == 0.0):
        raise ConfigError("transform maps a point to w = 0")
    return v / (w[..., None] if np.ndim(w) else w)


def transform_vec3(m, v):
    """src/math.rs:112-115."""
    return np.asarray(v, dtype=np.float64) @ m[:3, :3].T


def _normalize(v):
    return v * (1.0 / np.sqrt(np.sum(v * v, axis=-1, keepdims=True)))


# ----------------------------------------------------------------------------- shapes/utils.rs
_PLY_SCALAR = {
    "char": ("i1", "Char"), "int8": ("i1", "Char"),
    "uchar": ("u1", "UChar"), "uint8": ("u1", "UChar"),
    "short": ("i2", "Short"), "int16": ("i2", "Short"),
    "ushort": ("u2", "UShort"), "uint16": ("u2", "UShort"),
    "int": ("i4", "Int"), "int32": ("i4", "Int"),
    "uint": ("u4", "UInt"), "uint32": ("u4", "UInt"),
    "float": ("f4", "Float"), "float32": ("f4", "Float"),
    "double": ("f8", "Double"), "float64": ("f8", "Double"),
}


@_as_config_error("mesh")
def load_ply(path):
    """PlyMeshLoader::load (src/shapes/utils.rs:16-82).

    Accepts ascii / binary_little_endian / binary_big_endian like ply-rs, then
    enforces what the reference enforces: vertex x,y,z,nx,ny,nz must be `float`
    and face.vertex_indices a list of `uint` (any count type).
    Returns (vertices [n,3] f64, normals [n,3] f64 normalised, faces list/array).
    """
    try:
        f = open(path, "rb")
    except OSError as e:
        raise ConfigError(f"cannot open mesh {path}: {e}") from e
    with f:
        if f.readline().strip() != b"ply":
            raise ConfigError(f"{path}: not a PLY file")
        fmt = None
        elements = []  # [name, count, [(kind, name, types...)]]
        while True:
            line = f.readline()
            if not line:
                raise ConfigError(f"{path}: unexpected end of header")
            tok = line.decode("ascii", "replace").split()
            if not tok or tok[0] in ("comment", "obj_info"):
                continue
            if tok[0] == "format":
                fmt = tok[1]
            elif tok[0] == "element":
                elements.append([tok[1], int(tok[2]), []])
                # (every record takes at least a byte: a count beyond the file's size is a lie, not a reason to allocate)
                if not 0 <= elements[-1][1] <= os.path.getsize(path):
                    raise ConfigError(f"{path}: element {tok[1]} claims {tok[2]} records in a file of {os.path.getsize(path)} bytes")
            elif tok[0] == "property":
                if tok[1] == "list":
                    elements[-1][2].append(("list", tok[4], tok[2], tok[3]))
                else:
                    elements[-1][2].append(("scalar", tok[2], tok[1]))
            elif tok[0] == "end_header":
                break
        if fmt not in ("ascii", "binary_little_endian", "binary_big_endian"):
            raise ConfigError(f"{path}: unsupported PLY format {fmt}")
        payload = {}
        endian = "<" if fmt != "binary_big_endian" else ">"
        for name, count, props in elements:
            has_list = any(p[0] == "list" for p in props)
            if fmt == "ascii":
                rows = [f.readline().split() for _ in range(count)]
                payload[name] = _ply_rows_ascii(rows, props)
            elif not has_list:
                dt = np.dtype([(p[1], endian + _PLY_SCALAR[p[2]][0]) for p in props])
                payload[name] = np.frombuffer(f.read(dt.itemsize * count), dtype=dt, count=count)
            else:
                payload[name] = _ply_lists_binary(f, count, props, endian)
        payload["__props__"] = {n: p for n, _, p in elements}

    vprops = {p[1]: p for p in payload["__props__"].get("vertex", [])}
    for k in ("x", "y", "z", "nx", "ny", "nz"):
        if k not in vprops:
            raise ConfigError(f"{path}: vertex property {k} missing")
        if vprops[k][0] != "scalar" or _PLY_SCALAR[vprops[k][2]][1] != "Float":
            raise ConfigError(f"{k}'s type unrecognized")  # panic text of the reference
    vert = payload["vertex"]
    vertices = np.stack([np.asarray(vert[k], dtype=np.float64) for k in ("x", "y", "z")], axis=-1)
    normals = np.stack([np.asarray(vert[k], dtype=np.float64) for k in ("nx", "ny", "nz")], axis=-1)
    with np.errstate(invalid="ignore", divide="ignore"):
        normals = _normalize(normals)
    fprops = {p[1]: p for p in payload["__props__"].get("face", [])}
    if "vertex_indices" not in fprops:
        raise ConfigError(f"{path}: face property vertex_indices missing")
    fp = fprops["vertex_indices"]
    if fp[0] != "list" or _PLY_SCALAR[fp[3]][1] != "UInt":
        raise ConfigError("vertex_indices's type unrecognized")
    faces = payload["face"]["vertex_indices"]
    return vertices, normals, faces


def _ply_rows_ascii(rows, props):
    out = {p[1]: [] for p in props}
    for r in rows:
        i = 0
        for p in props:
            if p[0] == "scalar":
                v = float(r[i])
                if _PLY_SCALAR[p[2]][1] == "Float":
                    v = float(np.float32(v))  # the file value is an f32 (ply-rs parses the token as f32)
                out[p[1]].append(v)
                i += 1
            else:
                n = int(r[i])
                out[p[1]].append(np.array([int(x) for x in r[i + 1 : i + 1 + n]], dtype=np.int64))
                i += 1 + n
    for p in props:
        if p[0] == "scalar":
            out[p[1]] = np.array(out[p[1]], dtype=np.float64)
        elif out[p[1]] and all(len(a) == len(out[p[1]][0]) for a in out[p[1]]):
            out[p[1]] = np.stack(out[p[1]])
    return out


def _ply_lists_binary(f, count, props, endian):
    """Binary element with list properties.  Fast path: one list property and
    every list the same length (the usual all-triangle / all-quad file)."""
    if len(props) == 1 and count > 0:
        _, name, cnt_t, idx_t = props[0]
        cdt = np.dtype(endian + _PLY_SCALAR[cnt_t][0])
        idt = np.dtype(endian + _PLY_SCALAR[idx_t][0])
        pos = f.tell()
        n0 = int(np.frombuffer(f.read(cdt.itemsize), dtype=cdt, count=1)[0])
        f.seek(pos)
        rec = np.dtype([("n", cdt), ("i", idt, (n0,))])
        buf = f.read(rec.itemsize * count)
        if len(buf) == rec.itemsize * count:
            arr = np.frombuffer(buf, dtype=rec, count=count)
            if np.all(arr["n"] == n0):
                return {name: arr["i"].astype(np.int64)}
        f.seek(pos)
    out = {p[1]: [] for p in props}
    for _ in range(count):
        for p in props:
            if p[0] == "scalar":
                dt = np.dtype(endian + _PLY_SCALAR[p[2]][0])
                out[p[1]].append(np.frombuffer(f.read(dt.itemsize), dtype=dt, count=1)[0])
            else:
                cdt = np.dtype(endian + _PLY_SCALAR[p[2]][0])
                idt = np.dtype(endian + _PLY_SCALAR[p[3]][0])
                n = int(np.frombuffer(f.read(cdt.itemsize), dtype=cdt, count=1)[0])
                out[p[1]].append(np.frombuffer(f.read(idt.itemsize * n), dtype=idt, count=n).astype(np.int64))
    return out


def are_points_coplanar(v0, v1, v2, v3):
    """src/shapes/quadrilateral.rs:23-33 (vectorised over leading axes)."""
    e01, e02, e03 = v1 - v0, v2 - v0, v3 - v0
    n1, n2 = np.cross(e01, e02), np.cross(e02, e03)
    with np.errstate(invalid="ignore", divide="ignore"):
        cos = np.sum(n1 * n2, -1) / (np.linalg.norm(n1, axis=-1) * np.linalg.norm(n2, axis=-1))
    return np.abs(np.abs(cos) - 1.0) < 1e-3


def is_quadrilateral_convex(v0, v1, v2, v3):
    """src/shapes/quadrilateral.rs:35-54."""
    e01, e12, e23, e30 = v1 - v0, v2 - v1, v3 - v2, v0 - v3
    c1, c2, c3, c4 = np.cross(e01, e12), np.cross(e12, e23), np.cross(e23, e30), np.cross(e30, e01)
    d = lambda a, b: np.sum(a * b, -1)
    return are_points_coplanar(v0, v1, v2, v3) & (d(c1, c2) >= 0) & (d(c2, c3) >= 0) & (d(c3, c4) >= 0) & (d(c4, c1) >= 0)


def load_mesh(path):
    """load_mesh (src/shapes/utils.rs:84-120): extension dispatch + validation.

    A face with more than four indices passes the reference's validation and
    panics at the first intersection (src/shapes/mesh.rs:75); here it is
    rejected at load time with that message.
    """
    if path.split(".")[-1] != "ply":
        raise ConfigError(f"Unsupported mesh format: {path}")
    vertices, normals, faces = load_ply(path)
    if isinstance(faces, np.ndarray) and faces.ndim == 2:
        groups = {faces.shape[1]: faces}
    else:
        groups = {}
        for fc in faces:
            groups.setdefault(len(fc), []).append(fc)
    for n, fs in groups.items():
        if n < 3:
            raise ConfigError(f"Invalid mesh: {list(fs[0])}")
        if n > 4:
            raise ConfigError("Mesh with non-triangle or non-quadrilateral face is not supported")
        fs = np.asarray(fs)
        if fs.size and (fs.min() < 0 or fs.max() >= len(vertices)):
            raise ConfigError("Invalid mesh: vertex index out of range")
        if n == 4 and fs.size:
            a, b, c, d = (vertices[fs[:, k]] for k in range(4))
            bad = ~are_points_coplanar(a, b, c, d)
            if bad.any():
                raise ConfigError(f"Invalid mesh: face {int(np.argmax(bad))}, Reason: coplanar")
            bad = ~is_quadrilateral_convex(a, b, c, d)
            if bad.any():
                raise ConfigError(f"Invalid mesh: face {int(np.argmax(bad))}, Reason: non-convex")
    return vertices, normals, faces


# ----------------------------------------------------------------------------- scene
_MATERIALS = {
    "Emissive": _abi.MAT_EMISSIVE,
    "Lambertian": _abi.MAT_LAMBERTIAN,
    "PhongSpecular": _abi.MAT_PHONG,
    "IdealReflector": _abi.MAT_REFLECTOR,
    "IdealDielectric": _abi.MAT_DIELECTRIC,
}


def perspective_camera(look_from, look_at, vup, vfov, aspect):
    """PerspectiveCamera::new (src/camera.rs:21-34) in f64; dict of the 4 state vectors."""
    look_from, look_at, vup = (np.asarray(a, dtype=np.float64) for a in (look_from, look_at, vup))
    theta = vfov * np.pi / 180.0
    half_height = np.tan(theta / 2.0)
    half_width = aspect * half_height
    w = _normalize(look_from - look_at)
    u = _normalize(np.cross(vup, w))
    v = np.cross(w, u)
    return {
        "origin": look_from,
        "lower_left_corner": look_from - half_width * u - half_height * v - w,
        "horizontal": 2.0 * half_width * u,
        "vertical": 2.0 * half_height * v,
    }


class FlatScene:
    """Flattened scene: camera state + rrt_prim array + rrt_material array."""

    def __init__(self, camera, prims, materials):
        self.camera = camera
        self.prims = np.ascontiguousarray(prims, dtype=_abi.PRIM_DTYPE)
        self.materials = np.ascontiguousarray(materials, dtype=_abi.MATERIAL_DTYPE)

    @property
    def n_prims(self):
        return len(self.prims)

    @property
    def n_objects(self):
        return len(self.materials)

    def desc(self):
        """ctypes rrt_scene_desc pointing into this object's arrays (keep self alive)."""
        d = _abi.SceneDesc()
        for k in ("origin", "lower_left_corner", "horizontal", "vertical"):
            getattr(d.camera, k)[:] = [float(x) for x in self.camera[k]]
        d.n_prims = self.n_prims
        d.n_objects = self.n_objects
        d.prims = self.prims.ctypes.data
        d.materials = self.materials.ctypes.data
        return d


def _resolve(path, base_dirs):
    if os.path.isabs(path) or os.path.exists(path):
        return path
    for b in base_dirs:
        p = os.path.join(b, path)
        if os.path.exists(p):
            return p
    return path


def _shape_prims(shape, object_id, base_dirs):
    typ = shape.get("type")
    m = matrix_from_config(shape.get("transform"))
    if typ == "Sphere":  # src/shapes/sphere.rs:104-119: only the centre is transformed
        p = np.zeros(1, dtype=_abi.PRIM_DTYPE)
        p["type"] = _abi.PRIM_SPHERE
        p["v"][0, 0:3] = transform_point3(m, _vec3(shape["center"], "center"))
        p["v"][0, 3] = float(shape["radius"])
    elif typ == "Plane":  # src/shapes/plane.rs:46-61
        p = np.zeros(1, dtype=_abi.PRIM_DTYPE)
        p["type"] = _abi.PRIM_PLANE
        n = _normalize(_vec3(shape["normal"], "normal"))
        p["v"][0, 0:3] = transform_point3(m, _vec3(shape["point"], "point"))
        p["v"][0, 3:6] = _normalize(transform_vec3(m, n))
    elif typ in ("Triangle", "Quadrilateral"):  # triangle.rs:90-111, quadrilateral.rs:192-215
        k = 3 if typ == "Triangle" else 4
        vs = shape["vertices"]
        if len(vs) != k:
            raise ConfigError(f"{typ}: expected {k} vertices")
        p = np.zeros(1, dtype=_abi.PRIM_DTYPE)
        p["type"] = _abi.PRIM_TRIANGLE if k == 3 else _abi.PRIM_QUAD
        for i, v in enumerate(vs):
            p["v"][0, 3 * i : 3 * i + 3] = transform_point3(m, _vec3(v, "vertex"))
    elif typ == "Mesh":  # src/shapes/mesh.rs:91-114
        vertices, _normals, faces = load_mesh(_resolve(shape["file"], base_dirs))
        vertices = transform_point3(m, vertices)
        if isinstance(faces, np.ndarray) and faces.ndim == 2:
            k = faces.shape[1]
            p = np.zeros(len(faces), dtype=_abi.PRIM_DTYPE)
            p["type"] = _abi.PRIM_TRIANGLE if k == 3 else _abi.PRIM_QUAD
            p["v"][:, : 3 * k] = vertices[faces].reshape(len(faces), 3 * k)
        else:
            p = np.zeros(len(faces), dtype=_abi.PRIM_DTYPE)
            for i, fc in enumerate(faces):
                k = len(fc)
                p["type"][i] = _abi.PRIM_TRIANGLE if k == 3 else _abi.PRIM_QUAD
                p["v"][i, : 3 * k] = vertices[np.asarray(fc)].reshape(3 * k)
        p["prim_id"] = np.arange(len(p), dtype=np.uint32)
    else:
        raise ConfigError(f"unknown shape type {typ!r}")
    p["object_id"] = object_id
    return p


def _material(mat):
    out = np.zeros(1, dtype=_abi.MATERIAL_DTYPE)
    typ = mat.get("type")
    if typ not in _MATERIALS:
        raise ConfigError(f"unknown material type {typ!r}")
    out["type"] = _MATERIALS[typ]
    if typ == "Emissive":
        out["rgb"][0] = _vec3(mat["color"], "color")
    elif typ == "Lambertian":
        out["rgb"][0] = _vec3(mat["albedo"], "albedo")
    elif typ == "PhongSpecular":
        out["rgb"][0] = _vec3(mat["specular"], "specular")
        out["param"] = float(mat["shininess"])
    elif typ == "IdealDielectric":
        out["param"] = float(mat["ior"])
    return out


def scene_from_dict(cfg, base_dirs=()):
    """Scene::from_config (src/scene.rs:20-33) on an already-parsed TOML dict."""
    cam = cfg.get("camera")
    if not cam or cam.get("type") != "Perspective":
        raise ConfigError("camera: type must be \"Perspective\"")
    camera = perspective_camera(
        _vec3(cam["look_from"], "look_from"), _vec3(cam["look_at"], "look_at"), _vec3(cam["vup"], "vup"),
        float(cam["vfov"]), float(cam["aspect"]),
    )
    prims, mats = [], []
    for oid, obj in enumerate(cfg.get("objects", [])):
        prims.append(_shape_prims(obj["shape"], oid, base_dirs))
        mats.append(_material(obj["material"]))
    prims = np.concatenate(prims) if prims else np.zeros(0, dtype=_abi.PRIM_DTYPE)
    mats = np.concatenate(mats) if mats else np.zeros(0, dtype=_abi.MATERIAL_DTYPE)
    return FlatScene(camera, prims, mats)


REPO_ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@_as_config_error("scene")
def load_scene(path, base_dirs=None):
    """Parse a scene TOML (src/main.rs:45-47).  Mesh paths are resolved like the
    reference does (relative to the working directory) and then against
    base_dirs (default: the repository root, where configs/ and assets/ live)."""
    with open(path, "rb") as f:
        cfg = _toml.load(f)
    if base_dirs is None:
        base_dirs = (REPO_ROOT, os.path.dirname(os.path.abspath(path)))
    return scene_from_dict(cfg, base_dirs)


class RenderConfig:
    """RenderConfig (src/renderer.rs:12-37)."""

    def __init__(self, cfg):
        try:
            tr, im, sa = cfg["tracer"], cfg["image"], cfg["sampler"]
            pp, perf = cfg["post_processing"], cfg["performance"]
        except KeyError as e:
            raise ConfigError(f"missing table {e}") from e
        if tr.get("type") != "mcpt":
            raise ConfigError("tracer.type must be \"mcpt\"")
        self.min_depth = int(tr["min_depth"])
        self.max_depth = int(tr["max_depth"])
        self.width = int(im["width"])
        self.height = int(im["height"])
        st = sa.get("type")
        if st == "Random":
            self.sampler_type = _abi.SAMPLER_RANDOM
            self.spp = int(sa["samples_per_pixel"])
            self.x_strata = self.y_strata = 0
        elif st == "Stratified":
            self.sampler_type = _abi.SAMPLER_STRATIFIED
            self.spp = int(sa["samples_per_pixel"])
            self.x_strata = int(sa["x_strata"])
            self.y_strata = int(sa["y_strata"])
            if self.x_strata * self.y_strata != self.spp:
                raise ConfigError("x_strata * y_strata != samples_per_pixel")  # src/sampler.rs:87-91
        else:
            raise ConfigError(f"unknown sampler type {st!r}")
        tm = pp.get("tone_mapping")
        self.tone_mapping = 1 if tm == "reinhard" else 0  # unknown names are the identity (renderer.rs:53-56)
        if "gamma_correction" not in pp:
            raise ConfigError("post_processing.gamma_correction missing")
        self.gamma_correction = bool(pp["gamma_correction"])
        wb = pp.get("white_balance")
        self.white_balance = None if wb is None else _vec3(wb, "white_balance")
        self.parallelism = perf.get("parallelism")
        self.seed = int(cfg.get("seed", 0))  # extension key; serde ignores unknown keys

    @property
    def samples_traced(self):
        """RandomSampler traces spp+1 samples per pixel (src/sampler.rs:48-55)."""
        return self.spp + 1 if self.sampler_type == _abi.SAMPLER_RANDOM else self.spp

    def render_desc(self, sample_begin=0, sample_end=None, seed=None, precision=_abi.PRECISION_F32,
                    accel=_abi.ACCEL_AUTO, pipeline=_abi.PIPELINE_AUTO, flags=0):
        d = _abi.RenderDesc()
        d.width, d.height = self.width, self.height
        d.sampler_type, d.spp = self.sampler_type, self.spp
        d.x_strata, d.y_strata = self.x_strata, self.y_strata
        d.min_depth, d.max_depth = self.min_depth, self.max_depth
        d.sample_begin = sample_begin
        d.sample_end = self.samples_traced if sample_end is None else sample_end
        d.seed = self.seed if seed is None else seed
        d.precision, d.accel, d.pipeline = precision, accel, pipeline
        d.flags = flags
        return d

    def post_desc(self):
        p = _abi.PostDesc()
        p.tone_mapping = self.tone_mapping
        p.gamma_correction = 1 if self.gamma_correction else 0
        p.has_white_balance = 0 if self.white_balance is None else 1
        if self.white_balance is not None:
            p.white_balance[:] = [float(x) for x in self.white_balance]
        return p


@_as_config_error("render config")
def load_render_config(path):
    with open(path, "rb") as f:
        return RenderConfig(_toml.load(f))


def stratified_config(width, height, x_strata, y_strata, min_depth=2, max_depth=6, seed=0):
    """A render config with the shipped configs' shape at another size."""
    return RenderConfig({
        "tracer": {"type": "mcpt", "min_depth": min_depth, "max_depth": max_depth},
        "image": {"width": width, "height": height},
        "sampler": {"type": "Stratified", "samples_per_pixel": x_strata * y_strata,
                    "x_strata": x_strata, "y_strata": y_strata},
        "post_processing": {"tone_mapping": "reinhard", "gamma_correction": True},
        "performance": {"parallelism": 0},
        "seed": seed,
    })


def random_config(width, height, spp, min_depth=2, max_depth=6, seed=0):
    return RenderConfig({
        "tracer": {"type": "mcpt", "min_depth": min_depth, "max_depth": max_depth},
        "image": {"width": width, "height": height},
        "sampler": {"type": "Random", "samples_per_pixel": spp},
        "post_processing": {"tone_mapping": "reinhard", "gamma_correction": True},
        "performance": {"parallelism": 0},
        "seed": seed,
    })
