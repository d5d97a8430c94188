"""Material::scatter and the throughput factor on the device, sample by sample against the f64 oracle
(src/material.rs:38-284, src/tracers/utils.rs:66-87), through rrt_material_probe: same incoming
direction, normal and random numbers on both sides.

What must agree: whether the path goes on, the throughput factor, and the new direction -- exactly
for IdealReflector and IdealDielectric; for the two lobe samplers the polar angle about the lobe axis
only, because the device builds another orthonormal basis around that axis than
local_coordinate_system (src/math.rs:162-172) and the lobes are symmetric in the azimuth.  The
reference's own material-related tests (reflect, refract, Fresnel: src/math.rs:209-284) are restated
on the device output as well."""
import numpy as np
import pytest

from rustraytracer_amd import _abi

pytestmark = pytest.mark.gpu

N = 1500


def _unit(v):
    return v / np.linalg.norm(v, axis=-1, keepdims=True)


def _material(kind, rgb=(0, 0, 0), param=0.0):
    m = np.zeros(1, dtype=_abi.MATERIAL_DTYPE)
    m["type"], m["rgb"], m["param"] = kind, rgb, param
    return m


def _inputs(seed, two_sided=False):
    rng = np.random.default_rng(seed)
    n = _unit(rng.normal(size=(N, 3)))
    d = _unit(rng.normal(size=(N, 3)))
    facing = np.einsum("ij,ij->i", d, n) < 0
    if not two_sided:
        d[~facing] = -d[~facing]  # arrive from the side the normal points to
    return d, n, rng.random((N, 3))


def _oracle(O, mat, d, n, u):
    out = [O.material_sample(mat, d[i], np.zeros(3), n[i], u[i, 0], u[i, 1], u[i, 2]) for i in range(N)]
    goes_on = np.array([r is not None and r["pdf"] > 1e-6 and np.all(np.isfinite(r["weight"])) and np.any(r["weight"] != 0)
                        for r in out])
    return out, goes_on


def _tols(api, prec):
    return (1e-12, 1e-12) if prec == api.PRECISION_F64 else (2e-5, 2e-5)


@pytest.mark.parametrize("prec_name", ["f64", "f32"])
def test_lambertian(gpu, O, prec_name):  # src/material.rs:67-91
    api, be = gpu
    prec = api.PRECISION_F64 if prec_name == "f64" else api.PRECISION_F32
    albedo = (0.8, 0.5, 0.25)
    mat = _material(_abi.MAT_LAMBERTIAN, albedo)
    d, n, u = _inputs(1)
    d_out, w, ok = be.material_probe(mat, d, n, u, precision=prec)
    ref, goes_on = _oracle(O, mat, d, n, u)
    atol, rtol = _tols(api, prec)
    clear = np.array([abs(r["pdf"] - 1e-6) > 1e-7 for r in ref])
    assert np.array_equal(ok[clear], goes_on[clear]) and ok.sum() > 0.99 * N
    for i in np.nonzero(ok & clear)[0]:
        assert np.allclose(w[i], ref[i]["weight"], rtol=max(rtol, 1e-9), atol=0)  # = albedo: cos * (albedo / pi) / (cos / pi)
        assert abs(np.linalg.norm(d_out[i]) - 1) <= 10 * atol + 1e-12
        if prec == api.PRECISION_F64:
            # the reference's mapping: cos(theta) = sqrt(1 - u), the same on both sides whatever the basis
            assert abs(d_out[i] @ n[i] - ref[i]["d_out"] @ n[i]) <= 10 * atol + 1e-12
            assert abs(d_out[i] @ n[i] - np.sqrt(1 - u[i, 1])) <= 10 * atol + 1e-9
    if prec == api.PRECISION_F32:
        # the f32 kernels draw the cosine-weighted distribution through another map of (u, v) (normal + uniform sphere
        # point): cos^2(theta) must be uniform on [0, 1] (Kolmogorov-Smirnov against the uniform distribution) and the
        # direction must be the one that map gives
        c2 = np.sort(np.einsum("ij,ij->i", d_out[ok], n[ok]) ** 2)
        ks = np.max(np.abs(c2 - (np.arange(len(c2)) + 0.5) / len(c2)))
        assert ks < 1.63 / np.sqrt(len(c2)), ks  # 1 % level
        r = 2 * np.sqrt(np.maximum(u[:, 1] - u[:, 1] ** 2, 0))
        s_pt = np.stack([r * np.cos(2 * np.pi * u[:, 2]), r * np.sin(2 * np.pi * u[:, 2]), 1 - 2 * u[:, 1]], axis=1)
        want = _unit(n + s_pt)
        assert np.max(np.abs(d_out[ok] - want[ok])) < 2e-4
    # the azimuth is uniform: the component along any fixed tangent has zero mean
    t = _unit(np.cross(n, np.roll(n, 1, axis=1)))
    assert abs(np.mean(np.einsum("ij,ij->i", d_out[ok], t[ok]))) < 4 * 0.5 / np.sqrt(N)


@pytest.mark.parametrize("prec_name", ["f64", "f32"])
def test_phong(gpu, O, prec_name):  # src/material.rs:104-142
    api, be = gpu
    prec = api.PRECISION_F64 if prec_name == "f64" else api.PRECISION_F32
    spec, s = np.array((0.9, 0.6, 0.3)), 20.0
    mat = _material(_abi.MAT_PHONG, spec, s)
    d, n, u = _inputs(2)
    d_out, w, ok = be.material_probe(mat, d, n, u, precision=prec)
    ref, _ = _oracle(O, mat, d, n, u)
    atol, rtol = _tols(api, prec)
    refl = d - 2 * np.einsum("ij,ij->i", d, n)[:, None] * n
    assert ok.sum() > 0.95 * N
    for i in np.nonzero(ok)[0]:
        assert abs(np.linalg.norm(d_out[i]) - 1) <= 10 * atol + 1e-12
        # polar angle about the mirror direction: cos = u^(1 / (s + 1))
        assert abs(d_out[i] @ refl[i] - ref[i]["d_out"] @ refl[i]) <= 20 * atol + 1e-12
        # cos * bxdf / pdf = specular |d.n| (s + 2) / (s + 1); |d.n| depends on the azimuth, so it is taken from the
        # device's own direction, and the oracle's factor must be the same function of ITS direction
        assert np.allclose(w[i], spec * abs(d_out[i] @ n[i]) * (s + 2) / (s + 1), rtol=50 * rtol + 1e-9, atol=1e-7)
        if ref[i]["pdf"] > 1e-6:
            assert np.allclose(ref[i]["weight"], spec * abs(ref[i]["d_out"] @ n[i]) * (s + 2) / (s + 1), rtol=1e-9, atol=1e-12)
    # lanes the device stops are those whose pdf is (next to) zero
    for i in np.nonzero(~ok)[0]:
        assert ref[i]["pdf"] <= 2e-6 or not np.all(np.isfinite(ref[i]["weight"])) or not np.any(ref[i]["weight"] != 0)


@pytest.mark.parametrize("prec_name", ["f64", "f32"])
def test_ideal_reflector(gpu, O, prec_name):  # src/material.rs:156-181, test_reflect src/math.rs:209-224
    api, be = gpu
    prec = api.PRECISION_F64 if prec_name == "f64" else api.PRECISION_F32
    mat = _material(_abi.MAT_REFLECTOR)
    d, n, u = _inputs(3)
    d_out, w, ok = be.material_probe(mat, d, n, u, precision=prec)
    ref, goes_on = _oracle(O, mat, d, n, u)
    atol, _ = _tols(api, prec)
    cos_out = np.einsum("ij,ij->i", d_out, n)
    clear = np.abs(-np.einsum("ij,ij->i", d, n) - 1e-6) > 1e-5
    assert np.array_equal(ok[clear], goes_on[clear]) and ok.sum() > 0.99 * N
    for i in np.nonzero(ok & clear)[0]:
        assert np.allclose(d_out[i], ref[i]["d_out"], atol=10 * atol, rtol=0)
        assert np.allclose(w[i], ref[i]["weight"], rtol=1e-5 if prec_name == "f32" else 1e-12)  # c * (1 / c) / 1
        assert np.array_equal(w[i], [1, 1, 1])
        # the reference's test: the reflected ray leaves on the normal's side, and (r - v) is along n
        assert cos_out[i] >= 0
        assert np.allclose(_unit(d_out[i] - d[i]), n[i], atol=1e-6 if prec_name == "f64" else 1e-3)


@pytest.mark.parametrize("ior", [1.5, 0.8])
@pytest.mark.parametrize("prec_name", ["f64", "f32"])
def test_ideal_dielectric(gpu, O, prec_name, ior):  # src/material.rs:193-284, src/math.rs:137-160,226-284
    api, be = gpu
    prec = api.PRECISION_F64 if prec_name == "f64" else api.PRECISION_F32
    mat = _material(_abi.MAT_DIELECTRIC, (0, 0, 0), ior)
    d, n, u = _inputs(4, two_sided=True)  # rays arrive from outside and from inside
    d_out, w, ok = be.material_probe(mat, d, n, u, precision=prec)
    ref, goes_on = _oracle(O, mat, d, n, u)
    atol, rtol = _tols(api, prec)
    dn = np.einsum("ij,ij->i", d, n)
    inside = dn > 0
    eta_i, eta_t = np.where(inside, ior, 1.0), np.where(inside, 1.0, ior)
    F = np.array([O.fresnel(abs(dn[i]), eta_i[i], eta_t[i]) for i in range(N)])
    # leave out samples whose reflect-or-refract decision or whose cut-offs hang on the last bits
    clear = (np.abs(u[:, 0] - F) > (1e-9 if prec_name == "f64" else 1e-4)) & (np.abs(dn) > 1e-3) & (np.abs(F - 1e-6) > 1e-7) \
        & (np.abs(1 - F - 1e-6) > 1e-7)
    assert np.array_equal(ok[clear], goes_on[clear]) and ok.sum() > 0.9 * N
    n_reflect = n_refract = 0
    for i in np.nonzero(ok & clear)[0]:
        assert np.allclose(d_out[i], ref[i]["d_out"], atol=20 * atol, rtol=0)
        assert np.allclose(w[i], ref[i]["weight"], rtol=20 * rtol + 1e-9)
        on = -n[i] if inside[i] else n[i]
        if u[i, 0] < F[i]:
            n_reflect += 1
            assert np.array_equal(w[i], [1, 1, 1])  # c * (F / c) / F
            assert d_out[i] @ on >= 0
        else:
            n_refract += 1
            eta = eta_i[i] / eta_t[i]
            assert np.allclose(w[i], 1 / eta ** 2, rtol=20 * rtol + 1e-12)  # c * ((1 - F) / (c eta^2)) / (1 - F)
            # Snell (the reference's test_refract): sin(theta_t) = eta sin(theta_i), on the far side
            sin_i = np.sqrt(max(0.0, 1 - dn[i] ** 2))
            cos_t = -(d_out[i] @ on)
            assert cos_t >= 0
            assert abs(np.sqrt(max(0.0, 1 - cos_t ** 2)) - eta * sin_i) <= (1e-3 if prec_name == "f32" else 1e-9)
    assert n_reflect > 30 and n_refract > 300
    # total internal reflection: Fresnel is 1 there (test_fresnel), so those samples always reflect
    tir = (eta_i / eta_t) ** 2 * (1 - dn ** 2) > 1 + 1e-6
    if ior != 1.0 and tir.any():
        assert np.all(np.abs(F[tir] - 1) < 1e-12)
        sel = tir & ok & clear
        assert np.allclose(np.einsum("ij,ij->i", d_out[sel], d_out[sel]), 1, atol=1e-5)
        assert np.all(np.einsum("ij,ij->i", d_out[sel], np.where(inside[sel, None], -n[sel], n[sel])) >= 0)


@pytest.mark.parametrize("ior", [1.0, 1.00000005])
def test_dielectric_grazing_ray_takes_the_reflect_formula(gpu, O, ior):
    """Material::bxdf tests `|reflect - out|^2 < 1e-6` BEFORE the refracted direction (src/material.rs:269-273).  Through a
    boundary whose two indices (nearly) agree, a refracted ray that grazes the surface (|d . n| < 5e-4) also passes that
    first test, so the reference weights it with F / cos_t although it was sampled with probability 1 - F: the path's
    throughput gets F / (1 - F), not 1 / eta^2.  The f64 kernels must follow, ray by ray; steeper rays are unaffected."""
    api, be = gpu
    rng = np.random.default_rng(77)
    n = _unit(rng.normal(size=(N, 3)))
    tangent = _unit(np.cross(n, rng.normal(size=(N, 3))))
    c = np.where(np.arange(N) % 2 == 0, 10 ** rng.uniform(-5, np.log10(4e-4), N), 10 ** rng.uniform(np.log10(2e-3), -0.3, N))
    c = c * rng.choice([-1.0, 1.0], N)                      # from outside and from inside
    d = _unit(tangent * np.sqrt(1 - c[:, None] ** 2) + n * c[:, None])
    u = rng.random((N, 3))
    mat = _material(_abi.MAT_DIELECTRIC, (0, 0, 0), ior)
    d_out, w, ok = be.material_probe(mat, d, n, u, precision=api.PRECISION_F64)
    ref, goes_on = _oracle(O, mat, d, n, u)
    dn = np.einsum("ij,ij->i", d, n)
    inside = dn > 0
    F = np.array([O.fresnel(abs(dn[i]), ior if inside[i] else 1.0, 1.0 if inside[i] else ior) for i in range(N)])
    clear = (np.abs(u[:, 0] - F) > 1e-9) & (np.abs(np.abs(dn) - 5e-4) > 1e-5)
    assert np.array_equal(ok[clear], goes_on[clear])
    grazing_refracted = 0
    for i in np.nonzero(clear)[0]:
        if ref[i] is None or not goes_on[i]:
            continue
        # (with equal indices F is pure rounding noise, ~1e-18: compared absolutely)
        assert np.allclose(w[i], ref[i]["weight"], rtol=1e-7, atol=1e-12), (i, dn[i], w[i], ref[i]["weight"])
        assert np.allclose(d_out[i], ref[i]["d_out"], atol=1e-11, rtol=0)
        if abs(dn[i]) < 4.5e-4 and u[i, 0] >= F[i]:
            grazing_refracted += 1
            assert np.allclose(w[i], F[i] / (1 - F[i]), rtol=1e-6, atol=1e-12)  # the reflect formula over the refract probability
            assert not np.allclose(w[i], 1.0, rtol=1e-3)                         # ... and not 1 / eta^2
    assert grazing_refracted > 100
    # the product kernels: same code, f32 -- F itself is rounding noise this close to the surface, so only sanity is asked
    d32, w32, ok32 = be.material_probe(mat, d, n, u, precision=api.PRECISION_F32)
    assert np.all(np.isfinite(w32)) and np.all(np.isfinite(d32)) and np.all(w32 >= 0)
    steep = clear & (np.abs(dn) > 2e-3) & (np.abs(u[:, 0] - F) > 1e-3)
    assert np.array_equal(ok32[steep], goes_on[steep])


def test_emissive_does_not_scatter(gpu):  # src/material.rs:44-46
    api, be = gpu
    d, n, u = _inputs(5)
    _, _, ok = be.material_probe(_material(_abi.MAT_EMISSIVE, (3, 2, 1)), d, n, u)
    assert not ok.any()


def test_probe_argument_errors(gpu):
    api, be = gpu
    d, n, u = _inputs(6)
    bad = _material(9)
    with pytest.raises(Exception, match="material type"):
        be.material_probe(bad, d, n, u)
    with pytest.raises(ValueError):
        be.material_probe(_material(_abi.MAT_REFLECTOR), d, n[:-1], u)
    out = be.material_probe(_material(_abi.MAT_REFLECTOR), d[:0], n[:0], u[:0])  # empty input
    assert out[0].shape == (0, 3) and out[2].shape == (0,)
