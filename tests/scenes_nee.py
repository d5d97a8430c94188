"""Scenes for the next-event-estimation tests: small area lights of every light type."""
import scenes_util as S
from rustraytracer_amd import _abi


def lit_room(light="pgram"):
    """A Lambertian floor and back wall, a Phong side wall, a mirror and a glass ball, lit by ONE small emitter
    overhead: a parallelogram, a triangle, a bent quad (two triangle lights) or a sphere."""
    prims = [
        # (wound so that the normals the reference reports, (v1 - v0) x (v2 - v0), point into the room: its lobes are
        #  sampled about that normal whichever side the ray arrives from)
        (3, 0, 0, [[-4, -1, 1], [4, -1, 1], [4, -1, -9], [-4, -1, -9]]),        # floor, normal +y
        (3, 0, 1, [[-4, -1, -9], [4, -1, -9], [4, 5, -9], [-4, 5, -9]]),        # back wall, normal +z
        (3, 1, 0, [[-4, -1, -9], [-4, 5, -9], [-4, 5, 1], [-4, -1, 1]]),        # Phong side wall, normal +x
        (0, 2, 0, [1.5, 0.0, -5.0, 1.0]),                                        # mirror ball
        (0, 3, 0, [-1.2, -0.3, -3.5, 0.7]),                                      # glass ball
    ]
    if light == "pgram":
        prims.append((3, 4, 0, [[-0.8, 4.0, -5.5], [0.8, 4.0, -5.5], [0.8, 4.0, -4.0], [-0.8, 4.0, -4.0]]))
    elif light == "triangle":
        prims.append((2, 4, 0, [[-1.0, 4.0, -5.5], [1.0, 4.0, -5.5], [0.0, 4.2, -3.8]]))
    elif light == "bent":
        prims.append((3, 4, 0, [[-0.8, 4.0, -5.5], [0.8, 4.0, -5.5], [0.9, 3.9, -3.9], [-0.8, 4.0, -4.0]]))
    elif light == "sphere":
        prims.append((0, 4, 0, [0.0, 3.6, -4.8, 0.5]))
    else:
        raise ValueError(light)
    mats = [(_abi.MAT_LAMBERTIAN, (0.7, 0.6, 0.5), 0.0), (_abi.MAT_PHONG, (0.6, 0.7, 0.6), 12.0),
            (_abi.MAT_REFLECTOR, (0, 0, 0), 0.0), (_abi.MAT_DIELECTRIC, (0, 0, 0), 1.5),
            (_abi.MAT_EMISSIVE, (30.0, 25.0, 20.0), 0.0)]
    return S.make_scene(prims, mats, look_from=(0.5, 1.5, 0.5), look_at=(0, 0.8, -5), vfov=60.0)


def emissive_plane_room():
    prims = [(1, 0, 0, [0, -1, 0, 0, 1, 0]), (1, 1, 0, [0, 5, 0, 0, -1, 0])]
    mats = [(_abi.MAT_LAMBERTIAN, (0.7, 0.7, 0.7), 0.0), (_abi.MAT_EMISSIVE, (1.0, 1.0, 1.0), 0.0)]
    return S.make_scene(prims, mats, look_from=(0, 1, 0), look_at=(0, 0.5, -5), vfov=60.0)
