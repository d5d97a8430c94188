"""Oracle vs the reference's own math / camera tests (src/math.rs:192-313, src/camera.rs:77-103).

The two known answers are restated exactly; the property tests are restated with
a seeded generator and more draws than the reference's 10.
"""
import numpy as np


def _unit(rng):
    v = rng.random(3)
    return v / np.linalg.norm(v)


def _hemisphere_below(O, rng, n):
    # spherical_to_world(theta in [pi/2, pi), phi, n): the reference's way of
    # drawing an incoming direction with v.n <= 0
    return O.vec3_fn("oracle_spherical_to_world", rng.uniform(0.5 * np.pi, np.pi), rng.uniform(0, 2 * np.pi), n)


def test_ray_at_known_answer(O):  # src/math.rs:198-207
    p = O.vec3_fn("oracle_ray_at", [0.0, 0.0, 0.0], [1.0, 0.0, 0.0], 1.0)
    assert np.array_equal(p, [1.0, 0.0, 0.0])


def test_perspective_camera_known_answer(O):  # src/camera.rs:82-102
    cam = O.camera_perspective([0, 0, 0], [0, 0, -1], [0, 1, 0], 90.0, 2.0)
    o, d = O.create_ray(cam, 0.5, 0.5)
    assert np.linalg.norm(o - [0, 0, 0]) < 1e-6
    assert np.linalg.norm(d - [0, 0, -1]) < 1e-6
    # state vectors for this camera: half_height = tan(45 deg) = 1, half_width = 2
    assert np.allclose(cam.lower_left_corner[:], [-2, -1, -1], atol=1e-12)
    assert np.allclose(cam.horizontal[:], [4, 0, 0], atol=1e-12)
    assert np.allclose(cam.vertical[:], [0, 2, 0], atol=1e-12)


def test_camera_matches_python_loader(O, cfg):
    rng = np.random.default_rng(3)
    for _ in range(20):
        lf, la, up = rng.normal(size=3), rng.normal(size=3), _unit(rng)
        cam = O.camera_perspective(lf, la, up, 39.146, 1.33)
        py = cfg.perspective_camera(lf, la, up, 39.146, 1.33)
        for k in ("origin", "lower_left_corner", "horizontal", "vertical"):
            assert np.allclose(getattr(cam, k)[:], py[k], rtol=1e-13, atol=1e-13)


def test_reflect(O):  # src/math.rs:209-224
    rng = np.random.default_rng(1)
    for _ in range(200):
        n = _unit(rng)
        v = _hemisphere_below(O, rng, n)
        assert np.dot(v, n) <= 1e-12
        r = O.vec3_fn("oracle_reflect", v, n)
        assert np.dot(r, n) >= -1e-12
        assert np.linalg.norm(n - (r - v) / np.linalg.norm(r - v)) < 1e-6 or abs(np.dot(v, n)) < 1e-9


def test_refract_snell(O):  # src/math.rs:226-256
    rng = np.random.default_rng(2)
    seen_tir = seen_refr = 0
    for _ in range(400):
        n = _unit(rng)
        v = _hemisphere_below(O, rng, n)
        eta = rng.uniform(0.5, 2.0)
        r = O.refract(v, n, eta)
        cos1 = np.dot(-v, n)
        sin2 = np.sqrt(max(0.0, 1 - cos1 * cos1)) * eta
        if sin2 > 1.0:
            assert r is None
            seen_tir += 1
        else:
            assert r is not None and np.dot(r, n) <= 1e-12
            cos2 = np.dot(r, -n)
            assert abs(sin2 - np.sqrt(max(0.0, 1 - cos2 * cos2))) < 1e-3
            seen_refr += 1
    assert seen_tir > 10 and seen_refr > 10


def test_fresnel_reciprocity_and_tir(O):  # src/math.rs:258-284
    rng = np.random.default_rng(4)
    for _ in range(400):
        n = _unit(rng)
        v = _hemisphere_below(O, rng, n)
        eta_i, eta_t = rng.uniform(0.5, 2.0, size=2)
        cos_i = np.dot(-v, n)
        r = O.refract(v, n, eta_i / eta_t)
        if r is None:
            assert abs(O.fresnel(cos_i, eta_i, eta_t) - 1.0) < 1e-3
        else:
            cos_t = np.dot(r, -n)
            assert abs(O.fresnel(cos_i, eta_i, eta_t) - O.fresnel(cos_t, eta_t, eta_i)) < 1e-3


def test_fresnel_normal_incidence(O):
    # ((n-1)/(n+1))^2 at normal incidence: 0.04 for glass
    assert abs(O.fresnel(1.0, 1.0, 1.5) - 0.04) < 1e-12


def test_local_coordinate_system(O):  # src/math.rs:286-299
    rng = np.random.default_rng(5)
    for _ in range(200):
        n = _unit(rng) * rng.choice([-1, 1], size=3)
        n /= np.linalg.norm(n)
        u, v, w = O.local_coordinate_system(n)
        for a, b in ((u, v), (v, w), (w, u)):
            assert abs(np.dot(a, b)) < 1e-6
        for a in (u, v, w):
            assert abs(np.linalg.norm(a) - 1.0) < 1e-6
    # the |w.x| > 0.9 switch of the helper axis
    u, v, w = O.local_coordinate_system([1.0, 0.0, 0.0])
    assert np.allclose(u, np.cross([1, 0, 0], [0, 1, 0]))


def test_spherical_to_world(O):  # src/math.rs:301-312
    rng = np.random.default_rng(6)
    for _ in range(200):
        n = _unit(rng)
        theta, phi = rng.uniform(0, np.pi), rng.uniform(0, 2 * np.pi)
        v = O.vec3_fn("oracle_spherical_to_world", theta, phi, n)
        assert abs(np.linalg.norm(v) - 1.0) < 1e-6
        assert abs(np.dot(v, n) - np.cos(theta)) < 1e-6
