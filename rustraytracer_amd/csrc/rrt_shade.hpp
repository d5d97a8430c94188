// Path state and the per-bounce shading step (materials, Russian roulette).
#pragma once
#include "rrt_device.hpp"

namespace rrt {

template <typename R>
struct Path {
    V3<R> o, d;     // current ray
    V3<R> beta;     // throughput (src/tracers/utils.rs:24)
    uint32_t depth; // loop counter of generate_camera_vertices (src/tracers/utils.rs:35)
    Sampler smp;
};

// reflect / refract / fresnel: src/math.rs:133-160
template <typename R>
RRT_DEV V3<R> reflect(V3<R> v, V3<R> n) {
    R k = R(2) * dot(v, n);
    return mk<R>(v.x - n.x * k, v.y - n.y * k, v.z - n.z * k);
}

// ratio = eta_i / eta_t, which the caller has at hand
template <typename R>
RRT_DEV R fresnel(R cos_i, R eta_i, R eta_t, R ratio) {
    R sin2_t = ratio * ratio * (R(1) - cos_i * cos_i);
    if (sin2_t > R(1)) return R(1);
    R cos_t = Num<R>::sqrt(R(1) - sin2_t);
#ifdef RRT_EXP_FRESNEL_PK  // experiment: the two amplitude ratios side by side in packed arithmetic
    if (sizeof(R) == 4) {
        typedef float F2 __attribute__((ext_vector_type(2)));
        const F2 a = {(float)eta_t, (float)eta_i}, b = {(float)eta_i, (float)eta_t};
        const F2 x = a * (float)cos_i, y = b * (float)cos_t;
        const F2 num = x - y, den = x + y;
        const float r0 = num.x * Num<float>::rcp(den.x), r1 = num.y * Num<float>::rcp(den.y);
        return (R)((r0 * r0 + r1 * r1) * 0.5f);
    }
#endif
    R r_ortho = (eta_t * cos_i - eta_i * cos_t) * Num<R>::rcp(eta_t * cos_i + eta_i * cos_t);
    R r_parallel = (eta_i * cos_i - eta_t * cos_t) * Num<R>::rcp(eta_i * cos_i + eta_t * cos_t);
    return (r_ortho * r_ortho + r_parallel * r_parallel) * R(0.5);
}

// spherical_to_world (src/math.rs:174-179) with sin/cos of theta and phi already known, over an
// orthonormal basis around the unit vector w.  The reference builds its basis with two cross
// products and two normalisations (local_coordinate_system, src/math.rs:162-172); both lobes that
// use it (cosine-weighted, Phong) are symmetric in phi and phi is uniform, so the distribution of
// the sampled direction does not depend on which basis is taken.  This one is the branch-free
// construction of Duff et al. ("Building an Orthonormal Basis, Revisited", JCGT 2017): one
// reciprocal and a dozen multiply-adds.
// The f64 kernels, whose renders are compared with the CPU restatement path by path, build the
// reference's basis instead.  (The f32 kernels use this one for the Phong lobe only; their cosine lobe needs no basis.)
template <typename R>
RRT_DEV V3<R> to_world(R sin_t, R cos_t, R sin_p, R cos_p, V3<R> w) {
    if (sizeof(R) == 8) {  // local_coordinate_system, src/math.rs:162-172
        const V3<R> a = Num<R>::abs(w.x) > R(0.9) ? mk<R>(0, 1, 0) : mk<R>(1, 0, 0);
        const V3<R> u = normalize(cross(w, a));
        const V3<R> v = normalize(cross(w, u));
        const R cu = sin_t * cos_p, cv = sin_t * sin_p;
        return mk<R>(u.x * cu + v.x * cv + w.x * cos_t, u.y * cu + v.y * cv + w.y * cos_t,
                     u.z * cu + v.z * cv + w.z * cos_t);
    }
    const R sign = w.z >= R(0) ? R(1) : R(-1);
    const R a = R(-1) * Num<R>::rcp(sign + w.z);
    const R b = w.x * w.y * a;
    const V3<R> u = mk<R>(R(1) + sign * w.x * w.x * a, sign * b, -sign * w.x);
    const V3<R> v = mk<R>(b, sign + w.y * w.y * a, -w.y);
    R cu = sin_t * cos_p, cv = sin_t * sin_p;
    return mk<R>(u.x * cu + v.x * cv + w.x * cos_t, u.y * cu + v.y * cv + w.y * cos_t,
                 u.z * cu + v.z * cv + w.z * cos_t);
}

struct BounceCounters {
    uint32_t bounce[5];
};

// What is left to do for a path after its closest-hit query.
//   MODE_NEW: the sample is over (miss, emitter, Russian roulette, depth limit);
//   otherwise the material type whose scatter gives the next ray.
enum : int { MODE_NEW = 0 };

// The part of the loop body of generate_camera_vertices between the closest-hit
// query and the scatter call (src/tracers/utils.rs:41-64) for a path that hit
// primitive g at distance t: emitted radiance goes to `acc` (connect with s = 0,
// utils.rs:115-121), Russian roulette may end the path, and otherwise the hit
// point becomes the new origin and the surface (normal, material) is handed to
// the scatter step.  Returns the mode for that step; `counted` receives the
// material type whose scatter the reference would now evaluate (-1 if none).
//
// A path at its last permitted vertex (depth + 1 == max_depth) ends here: the
// reference still scatters it (utils.rs:66-87) and then leaves its loop, so the
// new ray is never traced and nothing it computed can reach the image.
// `sh` is the ShadeRec of the primitive that was hit and `m` the MatRec of its object (the caller
// fetches them: from HBM, or from the copy a kernel keeps in LDS for a small scene).
// count_emission = false: the emitter still ends the path but adds nothing (next-event estimation has already
// counted the light from the previous vertex).
// JUDGED_POINT: the vertex is computed with every product and sum rounded, which makes it, bit for bit, the point the
// primitive test judged (plane_point).  Shadow rays leave their vertex towards ONE point, so on a finely tessellated
// surface a whole band of them (the terminator as the light sees it) grazes the vertex's own and the neighbouring
// triangles; with a fused multiply-add here -- the compiler's choice, and not the same choice in every kernel -- an f32
// render of the million-triangle scene left the f64 render's paths in three times as many pixels.  The kernels of the
// reference's estimator keep the fused form (three instructions less per hit).
template <typename R, bool JUDGED_POINT = false>
RRT_DEV int process_hit(const RenderParams& rp, Path<R>& p, const ShadeRec<R>& sh, const MatRec<R>& m, R t, V3<R>& acc,
                        V3<R>& n, bool& bad, int& counted, bool count_emission = true) {
    V3<R> P;
    if constexpr (JUDGED_POINT) {
#pragma clang fp contract(off)
        P = mk<R>(p.o.x + t * p.d.x, p.o.y + t * p.d.y, p.o.z + t * p.d.z);
    } else {
        P = mk<R>(p.o.x + t * p.d.x, p.o.y + t * p.d.y, p.o.z + t * p.d.z);
    }
    n = mk<R>(sh.nx, sh.ny, sh.nz);
    if (sh.type == PRIM_SPHERE) {  // (point - center) / radius, src/shapes/sphere.rs:87
        R inv_r = Num<R>::rcp(sh.w);
        n = mk<R>((P.x - sh.nx) * inv_r, (P.y - sh.ny) * inv_r, (P.z - sh.nz) * inv_r);
    }
    if (m.type == MAT_EMISSIVE) {
        // an Emissive surface ends the path either way: as an emitter
        // (utils.rs:52-54) or because its scatter returns None (material.rs:44-46)
        V3<R> c = mk<R>(p.beta.x * m.r, p.beta.y * m.g, p.beta.z * m.b);
        if (!count_emission) c = mk<R>(0, 0, 0);
        bad = !(isfinite(c.x) && isfinite(c.y) && isfinite(c.z));
        acc = acc + c;
        return MODE_NEW;
    }
    // Russian roulette (utils.rs:56-64): one 1D draw per bounce, used or not
    const bool rr = p.depth > rp.min_depth;
    R xi = get_1d<R>(p.smp, rp, rr);
    if (rr) {
        R q = fmin(fmax(fmax(p.beta.x, p.beta.y), p.beta.z), R(1));
        if (xi > q) return MODE_NEW;
        p.beta = p.beta * Num<R>::rcp(q);
    }
    counted = m.type;
    if (p.depth + 1u >= rp.max_depth) return MODE_NEW;  // `for depth in 0..max_depth`
    p.o = P;
    return m.type;
}

// Material::scatter + bxdf for one path (src/material.rs:67-284) given the
// surface found by process_hit and, for the two lobe samplers, the 2D sample
// (u, v) drawn for it.  On success the path carries the new direction and the
// updated throughput and the function returns true.
//
// The throughput factor cos * bxdf / pdf (src/tracers/utils.rs:76-82) is applied
// in its reduced form, which is the same number wherever pdf > 1e-6:
//   Lambertian      (|c| albedo/pi) / (c/pi)                 = albedo
//   PhongSpecular   |d.n| spec (s+2)/(2pi) c^s / (c^s (s+1)/(2pi)) = spec |d.n| (s+2)/(s+1)
//   IdealReflector  c (1/c) / 1                              = 1   (0 if d.n <= 1e-6)
//   IdealDielectric reflect: c (F/c) / F = 1; refract: c ((1-F)/(c eta^2)) / (1-F) = 1/eta^2, except where the refracted
//                   direction also passes bxdf's test for the reflected one, which comes first: c (F/c) / (1-F)
//                   (0 if |d.n| < 1e-6)
// A path whose throughput became exactly zero is dropped: it can only add zero.
// draw_1d() supplies the dielectric's reflect-or-refract draw (the path's sampler in the renderers,
// a given number in the material probe).
template <typename R, typename Draw1D>
RRT_DEV bool scatter_with(Path<R>& p, int mode, V3<R> n, const MatRec<R>& m, R u, R v, Draw1D draw_1d) {
    V3<R> dir;
    if (mode == MAT_LAMBERTIAN) {  // src/material.rs:67-91
        R pdf;
        if (sizeof(R) == 8) {
            R cos_t = Num<R>::sqrt(R(1) - u);  // theta = acos(sqrt(1 - u))
            R sin_t = Num<R>::sqrt(u);
            R sp, cp;
            Num<R>::sincos2pi(v, sp, cp);
            dir = to_world(sin_t, cos_t, sp, cp, n);
            pdf = dot(dir, n) * Num<R>::inv_pi();
            if (pdf <= Num<R>::eps()) return false;
        } else {
            // The f32 kernels draw the same cosine-weighted distribution without a basis: the unit normal plus a
            // uniform point s on the unit sphere, normalised (the density of that direction is cos / pi).  s has
            // height 1 - 2u and azimuth 2 pi v; both maps keep strata compact.  ~30 instructions instead of ~70.
            R sp, cp;
            Num<R>::sincos2pi(v, sp, cp);
            const R r = R(2) * Num<R>::sqrt(fmax(u - u * u, R(0)));
            const V3<R> w = mk<R>(n.x + r * cp, n.y + r * sp, n.z + (R(1) - R(2) * u));
            dir = w * Num<R>::rsqrt(dot(w, w));
            pdf = dot(dir, n) * Num<R>::inv_pi();
            if (!(pdf > Num<R>::eps())) return false;  // (also a NaN direction: s = -n)
        }
        p.beta = mk<R>(p.beta.x * m.r, p.beta.y * m.g, p.beta.z * m.b);
    } else if (mode == MAT_PHONG) {  // src/material.rs:104-142
        V3<R> refl = reflect(p.d, n);
        R s = m.param;
        R inv_s1 = Num<R>::rcp(s + R(1));
        R cos_t = Num<R>::pow(u, inv_s1);
        R sin_t = Num<R>::sqrt(fmax(R(1) - cos_t * cos_t, R(0)));
        R sp, cp;
        Num<R>::sincos2pi(v, sp, cp);
        dir = to_world(sin_t, cos_t, sp, cp, refl);
        R c = dot(dir, refl);
        R pdf = Num<R>::pow(c, s) * (s + R(1)) * Num<R>::inv_pi() * R(0.5);
        if (!(pdf > Num<R>::eps())) return false;
        R w = Num<R>::abs(dot(dir, n)) * (s + R(2)) * inv_s1;
        p.beta = mk<R>(p.beta.x * m.r * w, p.beta.y * m.g * w, p.beta.z * m.b * w);
    } else if (mode == MAT_REFLECTOR) {  // src/material.rs:156-181
        dir = reflect(p.d, n);
        if (!(dot(dir, n) > Num<R>::eps())) return false;
    } else {  // IdealDielectric, src/material.rs:193-284
        R dn = dot(p.d, n);
        const bool inside = dn > R(0);
        V3<R> on = inside ? mk<R>(-n.x, -n.y, -n.z) : n;
        R eta_i = inside ? m.param : R(1);
        R eta_t = inside ? R(1) : m.param;
        // eta_i / eta_t and its inverse square come with the material (the host divides once, in f64):
        // r = 1 / ior, g = 1 / (1/ior)^2, b = 1 / ior^2
        R eta = inside ? m.param : m.r;
        R cos_theta = -dot(p.d, on);
        R r = draw_1d();
        R F = fresnel<R>(cos_theta, eta_i, eta_t, eta);
        R pdf, w;
        if (r < F) {
            dir = reflect(p.d, on);
            pdf = F;
            w = R(1);
        } else {
            R sin2_t = (R(1) - cos_theta * cos_theta) * eta * eta;  // refract, src/math.rs:137-146
            if (sin2_t > R(1)) return false;
            R cos_t2 = Num<R>::sqrt(R(1) - sin2_t);
            R k = eta * cos_theta - cos_t2;
            dir = mk<R>(eta * p.d.x + k * on.x, eta * p.d.y + k * on.y, eta * p.d.z + k * on.z);
            pdf = R(1) - F;
            w = inside ? m.b : m.g;  // 1 / eta^2
            // Material::bxdf asks "is it the reflected direction?" FIRST (src/material.rs:269-273), so a refracted ray
            // within 1e-3 of the reflected one -- a grazing ray through a boundary with eta close to 1 -- is weighted
            // with the reflect formula F / cos although it was sampled with probability 1 - F: factor F / (1 - F).
            // |reflect - refract|^2 with reflect = d + 2 c on, refract = eta d + (eta c - cos_t) on, d . on = -c:
            const R a = R(1) - eta, q = cos_theta * (R(2) - eta) + cos_t2;
            if (a * a + q * (q - R(2) * a * cos_theta) < Num<R>::eps()) w = F * Num<R>::rcp(pdf);
        }
        if (pdf <= Num<R>::eps()) return false;
        if (Num<R>::abs(dot(dir, n)) < Num<R>::eps()) return false;
        p.beta = p.beta * w;
    }
    p.d = dir;
    p.depth += 1;
    return !(p.beta.x == R(0) && p.beta.y == R(0) && p.beta.z == R(0));
}

// The draw a path needs for its scatter step: the lobe samplers take a 2D sample, the dielectric one number
// (src/material.rs:193-284 draws it before anything can fail), the mirror nothing.
RRT_DEV bool mode_draws(int mode) { return mode == MAT_LAMBERTIAN || mode == MAT_PHONG || mode == MAT_DIELECTRIC; }
RRT_DEV bool mode_draws_2d(int mode) { return mode != MAT_DIELECTRIC; }

// (u, v) = the draw made for this mode with get_draw: the 2D lobe sample, or u = the dielectric's number
template <typename R>
RRT_DEV bool scatter(Path<R>& p, int mode, V3<R> n, const MatRec<R>& m, R u, R v) {
    return scatter_with<R>(p, mode, n, m, u, v, [&]() { return u; });
}

// ---------------------------------------------------------------- next-event estimation (RRT_RENDER_NEE)
// The strategy the reference's connect() leaves unimplemented (s = 1, src/tracers/utils.rs:122-124): at a vertex
// with a lobe material one point y on an emitter is sampled and joined to the vertex x by a shadow ray.
//
// sample_light: light `pick` in [0,1) selects through the cumulative table, (u, v) the point on it.  Returns the
// unit direction x -> y, the distance, and  Le |n_y . wi| / |y - x|^2 / pdf_area(y)  (emitters radiate from both
// sides, as Material::emission has no facing test, src/material.rs:52-54).
template <typename R>
RRT_DEV bool sample_light(const LightView<R>& lv, V3<R> x, R pick, R u, R v, V3<R>& wi, R& dist, V3<R>& le_w, int& prim) {
#pragma clang fp contract(off)  // (the same point and direction in every kernel that samples the lights)
    if (lv.n <= 0) return false;
    int lo = 0, hi = lv.n - 1;  // first light whose cumulative probability exceeds pick (the last if none)
    while (lo < hi) {
        const int mid = (lo + hi) >> 1;
        if (pick < (R)lv.cdf[mid]) hi = mid;
        else lo = mid + 1;
    }
    const LightRec<R> l = lv.lights[lo];
    prim = l.type == LIGHT_SPHERE ? -2 : l.prim;  // a sphere hides its own far side: no exemption for it
    V3<R> y, ny;
    if (l.type == LIGHT_SPHERE) {
        const R z = R(1) - R(2) * u;
        const R r = Num<R>::sqrt(fmax(R(1) - z * z, R(0)));
        R sp, cp;
        Num<R>::sincos2pi(v, sp, cp);
        ny = mk<R>(r * cp, r * sp, z);
        y = mk<R>(l.v0[0] + l.e1[0] * ny.x, l.v0[1] + l.e1[0] * ny.y, l.v0[2] + l.e1[0] * ny.z);
    } else {
        R a = u, b = v;
        if (l.type == LIGHT_TRIANGLE) {  // uniform in the triangle: (1 - sqrt(u)) v0 + sqrt(u) (1 - v) v1 + sqrt(u) v v2
            const R su = Num<R>::sqrt(u);
            a = su * (R(1) - v);
            b = su * v;
        }
        y = mk<R>(l.v0[0] + a * l.e1[0] + b * l.e2[0], l.v0[1] + a * l.e1[1] + b * l.e2[1], l.v0[2] + a * l.e1[2] + b * l.e2[2]);
        ny = mk<R>(l.n[0], l.n[1], l.n[2]);
    }
    const V3<R> to = y - x;
    const R d2 = dot(to, to);
    if (!(d2 > R(0))) return false;
    dist = Num<R>::sqrt(d2);
    wi = to * (R(1) / dist);
    const R w = Num<R>::abs(dot(ny, wi)) * l.inv_pdf / d2;
    le_w = mk<R>(l.le[0] * w, l.le[1] * w, l.le[2] * w);
    return true;
}

// bxdf(wo, wi) |cos| of the two lobe materials for a GIVEN direction wi, zero where Material::scatter could not
// have produced wi (its pdf <= 1e-6 rejects, src/tracers/utils.rs:72): the same light transport as sampling the lobe.
//   Lambertian     albedo / pi * (n . wi)                          for n . wi / pi > 1e-6   (src/material.rs:67-91)
//   PhongSpecular  spec (s + 2) / (2 pi) c^s |n . wi|, c = r . wi  for c^s (s + 1) / (2 pi) > 1e-6   (src/material.rs:104-142)
template <typename R>
RRT_DEV V3<R> lobe_bxdf_cos(int mode, V3<R> d_in, V3<R> n, const MatRec<R>& m, V3<R> wi) {
    R k = R(0);
    if (mode == MAT_LAMBERTIAN) {
        const R c = dot(wi, n) * Num<R>::inv_pi();
        if (c > Num<R>::eps()) k = c;
    } else {
        const R c = dot(wi, reflect(d_in, n));
        if (c > R(0)) {
            const R cs = Num<R>::pow(c, m.param);
            if (cs * (m.param + R(1)) * Num<R>::inv_pi() * R(0.5) > Num<R>::eps())
                k = cs * (m.param + R(2)) * Num<R>::inv_pi() * R(0.5) * Num<R>::abs(dot(wi, n));
        }
    }
    return mk<R>(m.r * k, m.g * k, m.b * k);
}

// A shadow ray is unblocked if nothing is on its way, if the first thing on its way is the flat light's own primitive
// (`prim` of sample_light: a bent quad reports the distance to its first triangle's plane, not to the sampled point),
// or if that thing is not nearer than dist (1 - slack).
template <typename R>
RRT_DEV R nee_shadow_slack() { return R(1e-4); }

}  // namespace rrt
