/*
 * oracle.c - f64 CPU restatement of the RustRayTracer hot path.
 *
 * TEST INFRASTRUCTURE, NOT PRODUCT.  Only tests/, __graft_entry__.smoke() and
 * the cpu_baseline leg of bench.py may load this library; the shipped backend
 * never calls into it.
 *
 * Every function restates the behaviour (not the structure) of the reference
 * code cited above it; citations are file:line in nero19960329/RustRayTracer.
 * Arithmetic is f64 and follows the reference's operation order so that hit
 * decisions at primitive edges fall the same way.
 *
 * Pinning.  The reference cannot be compiled here (Rust; no rustc/cargo), so
 * this restatement is pinned by the reference's own known-answer tests
 * (test_ray_at src/math.rs:198-207, test_perspective_camera
 * src/camera.rs:82-102, test_load_mesh counts src/shapes/utils.rs:126-133) and
 * by its property tests restated with seeded inputs (tests/test_oracle_*.py).
 * Tracer, sampler, material scatter/bxdf, Scene/Mesh closest-hit and
 * post-process have NO test in the reference: for those rows PARITY IS
 * UNPINNED and rests on this line-by-line restatement plus analytic scenes.
 *
 * Third-party arithmetic restated from its published semantics (the crates are
 * not in the reference tree and there is no Cargo.lock):
 *   cgmath 0.18  normalize = v * (1/|v|); Matrix4::new is column-major
 *   rand 0.8     gen::<f64>() = 53 random bits * 2^-53 in [0,1);
 *                SliceRandom::shuffle = Fisher-Yates from the back.
 * thread_rng is OS-seeded in the reference (never reproducible); here each
 * 16x16 tile owns a xoshiro256** stream seeded from (seed, tile index).
 */
#include "oracle.h"

#include <math.h>
#include <pthread.h>
#include <stdlib.h>
#include <string.h>

/* ------------------------------------------------------------------ vec3 */
typedef struct { double x, y, z; } v3;

static inline v3 V(double x, double y, double z) { v3 r = {x, y, z}; return r; }
static inline v3 vadd(v3 a, v3 b) { return V(a.x + b.x, a.y + b.y, a.z + b.z); }
static inline v3 vsub(v3 a, v3 b) { return V(a.x - b.x, a.y - b.y, a.z - b.z); }
static inline v3 vmul(v3 a, double s) { return V(a.x * s, a.y * s, a.z * s); }
static inline v3 vdivs(v3 a, double s) { return V(a.x / s, a.y / s, a.z / s); }
static inline v3 vneg(v3 a) { return V(-a.x, -a.y, -a.z); }
static inline v3 vmulv(v3 a, v3 b) { return V(a.x * b.x, a.y * b.y, a.z * b.z); }
static inline double vdot(v3 a, v3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
static inline v3 vcross(v3 a, v3 b) {
    return V(a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x);
}
static inline double vmag2(v3 a) { return vdot(a, a); }
static inline double vmag(v3 a) { return sqrt(vdot(a, a)); }
/* cgmath InnerSpace::normalize: self * (1 / magnitude) */
static inline v3 vnormalize(v3 a) { return vmul(a, 1.0 / vmag(a)); }
static inline v3 ld3(const double* p) { return V(p[0], p[1], p[2]); }
static inline void st3(double* p, v3 a) { p[0] = a.x; p[1] = a.y; p[2] = a.z; }
static inline int vfinite(v3 a) { return isfinite(a.x) && isfinite(a.y) && isfinite(a.z); }

typedef struct { v3 o, d; } ray_t;
/* Ray::at  src/math.rs:127-131 */
static inline v3 ray_at(const ray_t* r, double t) { return vadd(r->o, vmul(r->d, t)); }

/* ------------------------------------------------------------ math.rs */
/* reflect  src/math.rs:133-135 :  v - n * 2 * v.dot(n) */
static inline v3 reflect(v3 v, v3 n) { return vsub(v, vmul(vmul(n, 2.0), vdot(v, n))); }

/* refract  src/math.rs:137-146 ; returns 0 on total internal reflection */
static int refract(v3 v, v3 n, double eta, v3* out) {
    double cos_theta1 = vdot(vneg(v), n);
    double sin2_theta1 = 1.0 - cos_theta1 * cos_theta1;
    double sin2_theta2 = sin2_theta1 * eta * eta;
    if (sin2_theta2 > 1.0) return 0;
    double cos_theta2 = sqrt(1.0 - sin2_theta2);
    *out = vadd(vmul(v, eta), vmul(n, eta * cos_theta1 - cos_theta2));
    return 1;
}

/* fresnel  src/math.rs:148-160 */
static double fresnel(double cos_i, double eta_i, double eta_t) {
    double sin2_t = (eta_i / eta_t) * (eta_i / eta_t) * (1.0 - cos_i * cos_i);
    if (sin2_t > 1.0) return 1.0;
    double cos_t = sqrt(1.0 - sin2_t);
    double r_ortho = ((eta_t * cos_i) - (eta_i * cos_t)) / ((eta_t * cos_i) + (eta_i * cos_t));
    double r_parallel = ((eta_i * cos_i) - (eta_t * cos_t)) / ((eta_i * cos_i) + (eta_t * cos_t));
    return (r_ortho * r_ortho + r_parallel * r_parallel) / 2.0;
}

/* local_coordinate_system  src/math.rs:162-172 */
static void local_coordinate_system(v3 normal, v3* u, v3* v, v3* w) {
    *w = normal;
    v3 a = fabs(normal.x) > 0.9 ? V(0.0, 1.0, 0.0) : V(1.0, 0.0, 0.0);
    *u = vnormalize(vcross(*w, a));
    *v = vnormalize(vcross(*w, *u));
}

/* spherical_to_world  src/math.rs:174-179 */
static v3 spherical_to_world(double theta, double phi, v3 normal) {
    v3 u, v, w;
    local_coordinate_system(normal, &u, &v, &w);
    return vadd(vadd(vmul(u, sin(theta) * cos(phi)), vmul(v, sin(theta) * sin(phi))),
                vmul(w, cos(theta)));
}

/* max_component  src/math.rs:117-119 */
static inline double max_component(v3 v) { return fmax(fmax(v.x, v.y), v.z); }

/* ----------------------------------------------------------- camera.rs */
/* PerspectiveCamera::new  src/camera.rs:21-34 */
int oracle_camera_perspective(const double look_from[3], const double look_at[3],
                              const double vup[3], double vfov, double aspect, rrt_camera* out) {
    const double PI = 3.14159265358979323846264338327950288;
    double theta = vfov * PI / 180.0;
    double half_height = tan(theta / 2.0);
    double half_width = aspect * half_height;
    v3 from = ld3(look_from);
    v3 w = vnormalize(vsub(from, ld3(look_at)));
    v3 u = vnormalize(vcross(ld3(vup), w));
    v3 v = vcross(w, u);
    st3(out->origin, from);
    st3(out->lower_left_corner,
        vsub(vsub(vsub(from, vmul(u, half_width)), vmul(v, half_height)), w));
    st3(out->horizontal, vmul(u, 2.0 * half_width));
    st3(out->vertical, vmul(v, 2.0 * half_height));
    return 0;
}

/* Camera::create_ray  src/camera.rs:38-45 */
static ray_t create_ray(const rrt_camera* c, double s, double t) {
    ray_t r;
    r.o = ld3(c->origin);
    v3 p = vadd(vadd(ld3(c->lower_left_corner), vmul(ld3(c->horizontal), s)),
                vmul(ld3(c->vertical), t));
    r.d = vnormalize(vsub(p, r.o));
    return r;
}

/* --------------------------------------------------------- shapes */
/* Sphere::intersect_geometric  src/shapes/sphere.rs:58-96 */
static int sphere_intersect(const double* s, const ray_t* ray, double t_min, double t_max,
                            double* t_out) {
    v3 center = ld3(s);
    double radius = s[3];
    v3 l = vsub(center, ray->o);
    double t_ca = vdot(l, ray->d);
    double d2 = vmag2(l) - t_ca * t_ca;
    if (d2 < 0.0 || d2 > radius * radius) return 0;
    double t_hc = sqrt(radius * radius - d2);
    double t0 = t_ca - t_hc;
    double t1 = t_ca + t_hc;
    if (t0 > t1) { double tmp = t0; t0 = t1; t1 = tmp; }
    if (t0 < t_min) {
        t0 = t1;
        if (t0 < t_min) return 0;
    }
    if (t0 > t_max) return 0;
    *t_out = t0;
    return 1;
}

/* Plane::intersect  src/shapes/plane.rs:25-44 */
static int plane_intersect(const double* p, const ray_t* ray, double t_min, double t_max,
                           double* t_out) {
    v3 point = ld3(p), normal = ld3(p + 3);
    double denominator = vdot(normal, ray->d);
    if (fabs(denominator) < 1e-6) return 0;
    v3 v = vsub(point, ray->o);
    double distance = vdot(v, normal) / denominator;
    if (distance < t_min || distance > t_max) return 0;
    *t_out = distance;
    return 1;
}

/* triangle_intersect (Moller-Trumbore)  src/shapes/triangle.rs:22-61 */
static int triangle_intersect(v3 v0, v3 v1, v3 v2, const ray_t* ray, double t_min, double t_max,
                              double* t_out) {
    v3 e1 = vsub(v1, v0);
    v3 e2 = vsub(v2, v0);
    v3 h = vcross(ray->d, e2);
    double a = vdot(e1, h);
    if (fabs(a) < 1e-6) return 0;
    double f = 1.0 / a;
    v3 s = vsub(ray->o, v0);
    double u = f * vdot(s, h);
    if (u < 0.0 || u > 1.0) return 0;
    v3 q = vcross(s, e1);
    double v = f * vdot(ray->d, q);
    if (v < 0.0 || u + v > 1.0) return 0;
    double t = f * vdot(e2, q);
    if (t > t_min && t < t_max) { *t_out = t; return 1; }
    return 0;
}

/* quadrilateral_intersect (Lagae-Dutre)  src/shapes/quadrilateral.rs:56-117.
 * Lines 119-161 compute bilinear (u,v) that every caller discards
 * (quadrilateral.rs:166, mesh.rs:56) and are left out. */
static int quadrilateral_intersect(v3 v0, v3 v1, v3 v2, v3 v3_, const ray_t* ray, double t_min,
                                   double t_max, double* t_out) {
    v3 e01 = vsub(v1, v0);
    v3 e03 = vsub(v3_, v0);
    v3 p = vcross(ray->d, e03);
    double det = vdot(e01, p);
    if (fabs(det) < 1e-6) return 0;
    double inv_det = 1.0 / det;
    v3 t = vsub(ray->o, v0);
    double alpha = vdot(t, p) * inv_det;
    if (alpha < 0.0) return 0;
    v3 q = vcross(t, e01);
    double beta = vdot(ray->d, q) * inv_det;
    if (beta < 0.0) return 0;
    if ((alpha + beta) > 1.0) {
        v3 e23 = vsub(v3_, v2);
        v3 e21 = vsub(v1, v2);
        v3 p_prime = vcross(ray->d, e21);
        double det_prime = vdot(e23, p_prime);
        if (fabs(det_prime) < 1e-6) return 0;
        double inv_det_prime = 1.0 / det_prime;
        v3 t_prime = vsub(ray->o, v2);
        double alpha_prime = vdot(t_prime, p_prime) * inv_det_prime;
        if (alpha_prime < 0.0) return 0;
        v3 q_prime = vcross(t_prime, e23);
        double beta_prime = vdot(ray->d, q_prime) * inv_det_prime;
        if (beta_prime < 0.0) return 0;
    }
    double ray_t_ = vdot(e03, q) * inv_det;
    if (ray_t_ < t_min || ray_t_ > t_max) return 0;
    *t_out = ray_t_;
    return 1;
}

typedef struct {
    double t;
    v3 p, normal;
    int32_t object_id, prim_id;
} hit_t;

/* One primitive against one ray, with the hit point and the normal each shape
 * reports: sphere.rs:86-87, plane.rs:37-43, triangle.rs:77-80 / mesh.rs:48-51,
 * quadrilateral.rs:179-182 / mesh.rs:69-72. */
static int prim_intersect(const rrt_prim* pr, const ray_t* ray, double t_min, double t_max,
                          hit_t* h) {
    double t;
    const double* v = pr->v;
    switch (pr->type) {
    case RRT_PRIM_SPHERE:
        if (!sphere_intersect(v, ray, t_min, t_max, &t)) return 0;
        h->t = t;
        h->p = ray_at(ray, t);
        h->normal = vdivs(vsub(h->p, ld3(v)), v[3]);
        return 1;
    case RRT_PRIM_PLANE:
        if (!plane_intersect(v, ray, t_min, t_max, &t)) return 0;
        h->t = t;
        h->p = ray_at(ray, t);
        h->normal = ld3(v + 3);
        return 1;
    case RRT_PRIM_TRIANGLE:
        if (!triangle_intersect(ld3(v), ld3(v + 3), ld3(v + 6), ray, t_min, t_max, &t)) return 0;
        break;
    case RRT_PRIM_QUAD:
        if (!quadrilateral_intersect(ld3(v), ld3(v + 3), ld3(v + 6), ld3(v + 9), ray, t_min, t_max,
                                     &t))
            return 0;
        break;
    default:
        return 0;
    }
    h->t = t;
    h->p = ray_at(ray, t);
    h->normal = vnormalize(vcross(vsub(ld3(v + 3), ld3(v)), vsub(ld3(v + 6), ld3(v))));
    return 1;
}

/* Scene::intersect  src/scene.rs:35-47 with Object::intersect
 * (src/object.rs:14-23) and Mesh::intersect (src/shapes/mesh.rs:28-89) folded
 * in: the mesh loop starts from the scene's closest_so_far and shrinks it per
 * face, which is the same as one flat scan over the faces in scene order. */
static int scene_intersect_scan(const rrt_scene_desc* sc, const ray_t* ray, hit_t* best) {
    int found = 0;
    double closest_so_far = 1.7976931348623157e308; /* f64::MAX */
    for (uint32_t i = 0; i < sc->n_prims; ++i) {
        hit_t h;
        if (prim_intersect(&sc->prims[i], ray, 0.001, closest_so_far, &h)) {
            closest_so_far = h.t;
            h.object_id = (int32_t)sc->prims[i].object_id;
            h.prim_id = (int32_t)sc->prims[i].prim_id;
            *best = h;
            found = 1;
        }
    }
    return found;
}

/* ---- a bounding-volume hierarchy for the ORACLE (test infrastructure; the reference has none, README.md lists it
 * as an open item).  Scenes of a million triangles take the scan above ~10 ms per ray; with the tree the oracle can
 * render them and be compared with the device.  The tree must return what the scan returns: the same primitive and
 * the same t, including the scan's rule for equal t (sphere / plane / quad hits replace an earlier hit at equal t,
 * triangle hits do not; src/shapes/ *.rs range tests).  tests/test_oracle_accel.py checks tree == scan.
 *   - binary tree, median split of the centroids along the longest axis, leaves of <= 4 primitives;
 *   - boxes from the vertices (sphere: centre +- radius), padded by 1e-9 of the scene size: a hit point lies on its
 *     primitive up to rounding, far inside the padding;
 *   - planes (unbounded) and quads that are not parallelograms (the reference's test accepts rays outside the
 *     corners of a bent quad, see quadrilateral_intersect) stay in a list that is scanned for every ray. */
typedef struct { double lo[3], hi[3]; int32_t left, right, first, count; } bvh_node; /* leaf: count > 0 */
typedef struct {
    const rrt_prim* prims_key; uint32_t n_prims_key; /* the scene it belongs to */
    bvh_node* nodes; size_t n_nodes;
    uint32_t* order;   /* primitive indices, leaves own ranges of it */
    uint32_t* linear; size_t n_linear;
} bvh_t;
static bvh_t* g_accel = NULL;

static void prim_bounds(const rrt_prim* p, double lo[3], double hi[3]) {
    const double* v = p->v;
    if (p->type == RRT_PRIM_SPHERE) {
        for (int k = 0; k < 3; ++k) { lo[k] = v[k] - fabs(v[3]); hi[k] = v[k] + fabs(v[3]); }
        return;
    }
    int nv = p->type == RRT_PRIM_TRIANGLE ? 3 : 4;
    for (int k = 0; k < 3; ++k) { lo[k] = hi[k] = v[k]; }
    for (int i = 1; i < nv; ++i)
        for (int k = 0; k < 3; ++k) {
            if (v[3 * i + k] < lo[k]) lo[k] = v[3 * i + k];
            if (v[3 * i + k] > hi[k]) hi[k] = v[3 * i + k];
        }
}

static int prim_in_tree(const rrt_prim* p) {
    if (p->type == RRT_PRIM_PLANE) return 0;
    if (p->type == RRT_PRIM_QUAD) {
        v3 v0 = ld3(p->v), v1 = ld3(p->v + 3), v2 = ld3(p->v + 6), v3_ = ld3(p->v + 9);
        v3 gap = vsub(vadd(v0, v2), vadd(v1, v3_));
        double scale = vmag(vsub(v1, v0)) + vmag(vsub(v3_, v0));
        if (!(vmag(gap) <= 1e-12 * scale)) return 0;
    }
    for (int i = 0; i < 12; ++i) if (!isfinite(p->v[i])) return 0;
    return 1;
}

typedef struct { const double* cen; int axis; } bvh_sort_ctx;
static bvh_sort_ctx g_sort; /* build is single-threaded */
static int bvh_cmp(const void* a, const void* b) {
    double x = g_sort.cen[3 * (size_t)*(const uint32_t*)a + g_sort.axis], y = g_sort.cen[3 * (size_t)*(const uint32_t*)b + g_sort.axis];
    return x < y ? -1 : x > y ? 1 : (*(const uint32_t*)a < *(const uint32_t*)b ? -1 : 1);
}

static int32_t bvh_build_node(bvh_t* t, const double* plo, const double* phi, const double* cen, size_t first, size_t count) {
    int32_t id = (int32_t)t->n_nodes++;
    bvh_node* n = &t->nodes[id];
    for (int k = 0; k < 3; ++k) { n->lo[k] = INFINITY; n->hi[k] = -INFINITY; }
    double clo[3] = {INFINITY, INFINITY, INFINITY}, chi[3] = {-INFINITY, -INFINITY, -INFINITY};
    for (size_t i = first; i < first + count; ++i) {
        size_t g = t->order[i];
        for (int k = 0; k < 3; ++k) {
            if (plo[3 * g + k] < n->lo[k]) n->lo[k] = plo[3 * g + k];
            if (phi[3 * g + k] > n->hi[k]) n->hi[k] = phi[3 * g + k];
            if (cen[3 * g + k] < clo[k]) clo[k] = cen[3 * g + k];
            if (cen[3 * g + k] > chi[k]) chi[k] = cen[3 * g + k];
        }
    }
    n->left = n->right = -1;
    n->first = (int32_t)first;
    n->count = (int32_t)count;
    if (count <= 4) return id;
    int axis = 0;
    if (chi[1] - clo[1] > chi[axis] - clo[axis]) axis = 1;
    if (chi[2] - clo[2] > chi[axis] - clo[axis]) axis = 2;
    g_sort.cen = cen;
    g_sort.axis = axis;
    qsort(t->order + first, count, sizeof(uint32_t), bvh_cmp);
    size_t half = count / 2;
    int32_t l = bvh_build_node(t, plo, phi, cen, first, half);
    int32_t r = bvh_build_node(t, plo, phi, cen, first + half, count - half);
    n = &t->nodes[id];
    n->left = l; n->right = r; n->count = 0;
    return id;
}

void oracle_accel_free(void) {
    if (!g_accel) return;
    free(g_accel->nodes); free(g_accel->order); free(g_accel->linear);
    free(g_accel);
    g_accel = NULL;
}

/* Builds the tree for `scene` and makes every later closest-hit query on that scene (same primitive array) use it,
 * until oracle_accel_free().  Not thread-safe against concurrent builds; renders only read it. */
int oracle_accel_build(const rrt_scene_desc* scene) {
    oracle_accel_free();
    size_t n = scene->n_prims;
    bvh_t* t = (bvh_t*)calloc(1, sizeof(bvh_t));
    double* plo = (double*)malloc(sizeof(double) * 3 * (n + 1));
    double* phi = (double*)malloc(sizeof(double) * 3 * (n + 1));
    double* cen = (double*)malloc(sizeof(double) * 3 * (n + 1));
    if (!t || !plo || !phi || !cen) { free(t); free(plo); free(phi); free(cen); return -1; }
    t->prims_key = scene->prims;
    t->n_prims_key = scene->n_prims;
    t->order = (uint32_t*)malloc(sizeof(uint32_t) * (n + 1));
    t->linear = (uint32_t*)malloc(sizeof(uint32_t) * (n + 1));
    t->nodes = (bvh_node*)malloc(sizeof(bvh_node) * (2 * n + 2));
    if (!t->order || !t->linear || !t->nodes) { free(plo); free(phi); free(cen); g_accel = t; oracle_accel_free(); return -1; }
    size_t m = 0;
    double slo[3] = {INFINITY, INFINITY, INFINITY}, shi[3] = {-INFINITY, -INFINITY, -INFINITY};
    for (size_t i = 0; i < n; ++i) {
        if (!prim_in_tree(&scene->prims[i])) { t->linear[t->n_linear++] = (uint32_t)i; continue; }
        prim_bounds(&scene->prims[i], plo + 3 * i, phi + 3 * i);
        for (int k = 0; k < 3; ++k) {
            cen[3 * i + k] = 0.5 * (plo[3 * i + k] + phi[3 * i + k]);
            if (plo[3 * i + k] < slo[k]) slo[k] = plo[3 * i + k];
            if (phi[3 * i + k] > shi[k]) shi[k] = phi[3 * i + k];
        }
        t->order[m++] = (uint32_t)i;
    }
    double size = 0.0;
    for (int k = 0; k < 3; ++k) {
        if (m && shi[k] - slo[k] > size) size = shi[k] - slo[k];
        if (m && fabs(slo[k]) > size) size = fabs(slo[k]);
        if (m && fabs(shi[k]) > size) size = fabs(shi[k]);
    }
    double pad = 1e-9 * size + 1e-300;
    for (size_t j = 0; j < m; ++j) {
        size_t i = t->order[j];
        for (int k = 0; k < 3; ++k) { plo[3 * i + k] -= pad; phi[3 * i + k] += pad; }
    }
    if (m) bvh_build_node(t, plo, phi, cen, 0, m);
    free(plo); free(phi); free(cen);
    g_accel = t;
    return 0;
}

static int scene_intersect_scan(const rrt_scene_desc* sc, const ray_t* ray, hit_t* best);

/* the scan's winner among equal-t hits, reconstructed from visits in any order: see the header of this section */
typedef struct { double t; int64_t first, last_incl; hit_t h_first, h_last; } tie_acc;
static void tie_consider(tie_acc* a, const rrt_scene_desc* sc, const ray_t* ray, uint32_t g) {
    hit_t h;
    /* t_max a little above the current best, so that a triangle AT the best distance (open interval in the
     * reference) is still seen; anything beyond the best is then dropped here */
    double tmax = a->t >= 1.7976931348623157e308 ? a->t : nextafter(a->t, INFINITY);
    if (!prim_intersect(&sc->prims[g], ray, 0.001, tmax, &h)) return;
    if (h.t > a->t) return;
    h.object_id = (int32_t)sc->prims[g].object_id;
    h.prim_id = (int32_t)sc->prims[g].prim_id;
    int inclusive = sc->prims[g].type != RRT_PRIM_TRIANGLE;
    if (h.t < a->t) {
        a->t = h.t;
        a->first = g; a->h_first = h;
        a->last_incl = inclusive ? (int64_t)g : -1;
        if (inclusive) a->h_last = h;
    } else {
        if ((int64_t)g < a->first) { a->first = g; a->h_first = h; }
        if (inclusive && (int64_t)g > a->last_incl) { a->last_incl = g; a->h_last = h; }
    }
}

static int bvh_intersect(const bvh_t* t, const rrt_scene_desc* sc, const ray_t* ray, hit_t* best) {
    tie_acc a;
    a.t = 1.7976931348623157e308;
    a.first = a.last_incl = -1;
    for (size_t i = 0; i < t->n_linear; ++i) tie_consider(&a, sc, ray, t->linear[i]);
    if (t->n_nodes) {
        int32_t stack[128];
        int sp = 0;
        stack[sp++] = 0;
        const double o[3] = {ray->o.x, ray->o.y, ray->o.z}, d[3] = {ray->d.x, ray->d.y, ray->d.z};
        while (sp) {
            const bvh_node* n = &t->nodes[stack[--sp]];
            double t0 = 0.0, t1 = a.t;
            int miss = 0;
            for (int k = 0; k < 3 && !miss; ++k) {
                if (d[k] == 0.0) {
                    if (o[k] < n->lo[k] || o[k] > n->hi[k]) miss = 1;
                    continue;
                }
                double ta = (n->lo[k] - o[k]) / d[k], tb = (n->hi[k] - o[k]) / d[k];
                if (ta > tb) { double x = ta; ta = tb; tb = x; }
                /* widen by rounding of the two divisions */
                ta -= 1e-12 * fabs(ta); tb += 1e-12 * fabs(tb);
                if (ta > t0) t0 = ta;
                if (tb < t1) t1 = tb;
                if (t0 > t1) miss = 1;
            }
            if (miss) continue;
            if (n->count > 0) {
                for (int32_t i = 0; i < n->count; ++i) tie_consider(&a, sc, ray, t->order[n->first + i]);
            } else if (sp + 2 <= 128) {
                stack[sp++] = n->left;
                stack[sp++] = n->right;
            } else {
                return scene_intersect_scan(sc, ray, best); /* never with a median-split tree (depth <= 32) */
            }
        }
    }
    if (a.first < 0) return 0;
    *best = a.last_incl > a.first ? a.h_last : a.h_first;
    return 1;
}

static int scene_intersect(const rrt_scene_desc* sc, const ray_t* ray, hit_t* best) {
    if (g_accel && g_accel->prims_key == sc->prims && g_accel->n_prims_key == sc->n_prims)
        return bvh_intersect(g_accel, sc, ray, best);
    return scene_intersect_scan(sc, ray, best);
}

int oracle_intersect(const rrt_scene_desc* scene, size_t n, const double* origins,
                     const double* dirs, int32_t* object_id, int32_t* prim_id, double* t) {
    for (size_t i = 0; i < n; ++i) {
        ray_t r;
        r.o = ld3(origins + 3 * i);
        r.d = ld3(dirs + 3 * i);
        hit_t h;
        if (scene_intersect(scene, &r, &h)) {
            object_id[i] = h.object_id; prim_id[i] = h.prim_id; t[i] = h.t;
        } else {
            object_id[i] = -1; prim_id[i] = -1; t[i] = INFINITY;
        }
    }
    return 0;
}

int oracle_trace_primary(const rrt_scene_desc* scene, size_t n, const double* u, const double* v,
                         int32_t* object_id, int32_t* prim_id, double* t) {
    for (size_t i = 0; i < n; ++i) {
        ray_t r = create_ray(&scene->camera, u[i], v[i]);
        hit_t h;
        if (scene_intersect(scene, &r, &h)) {
            object_id[i] = h.object_id; prim_id[i] = h.prim_id; t[i] = h.t;
        } else {
            object_id[i] = -1; prim_id[i] = -1; t[i] = INFINITY;
        }
    }
    return 0;
}

/* ------------------------------------------------------------ sampler.rs */
typedef struct { uint64_t s[4]; } rng_t;

static inline uint64_t rotl64(uint64_t x, int k) { return (x << k) | (x >> (64 - k)); }
static uint64_t splitmix64(uint64_t* x) {
    uint64_t z = (*x += 0x9e3779b97f4a7c15ULL);
    z = (z ^ (z >> 30)) * 0xbf58476d1ce4e5b9ULL;
    z = (z ^ (z >> 27)) * 0x94d049bb133111ebULL;
    return z ^ (z >> 31);
}
static void rng_seed(rng_t* r, uint64_t seed, uint64_t stream) {
    uint64_t x = seed ^ (stream * 0xd1342543de82ef95ULL + 0x2545f4914f6cdd1dULL);
    for (int i = 0; i < 4; ++i) r->s[i] = splitmix64(&x);
}
static inline uint64_t rng_next(rng_t* r) { /* xoshiro256** */
    uint64_t* s = r->s;
    uint64_t result = rotl64(s[1] * 5, 7) * 9;
    uint64_t t = s[1] << 17;
    s[2] ^= s[0]; s[3] ^= s[1]; s[1] ^= s[2]; s[0] ^= s[3];
    s[2] ^= t;
    s[3] = rotl64(s[3], 45);
    return result;
}
/* rand 0.8 Standard for f64: 53 bits scaled by 2^-53, in [0,1) */
static inline double rng_f64(rng_t* r) { return (double)(rng_next(r) >> 11) * (1.0 / 9007199254740992.0); }
/* uniform integer in [0, bound), unbiased (rejection) */
static inline uint64_t rng_below(rng_t* r, uint64_t bound) {
    uint64_t threshold = (0 - bound) % bound;
    for (;;) {
        uint64_t x = rng_next(r);
        if (x >= threshold) return x % bound;
    }
}

#define SAMPLER_DIMS 4 /* SamplerConfig::to_sampler hard-codes 4  src/sampler.rs:179-184 */

typedef struct {
    int type;
    size_t spp, x_strata, y_strata;
    double* s1d[SAMPLER_DIMS];  /* [dim][spp] */
    double* s2d[SAMPLER_DIMS];  /* [dim][spp][2] */
    size_t current_sample;      /* RandomSampler.current_sample / Stratified.current_sample_index */
    size_t current_dimension;
    rng_t rng;
    uint64_t draws;
    /* device_numbers != 0: the numbers come from the HIP backend's counter-based generator (below);
     * 2: as the f32 kernels see them (24-bit uniforms), and the two lobe samplers build the orthonormal
     * basis those kernels build (to_world_f32_kernels) */
    int device_numbers;
    uint32_t seed_lo, seed_hi, pixel, pixkey, key, ctr, perm_mask;
} sampler_t;

/* ---- the HIP backend's number generator, restated (rustraytracer_amd/csrc/rrt_device.hpp) ----
 * NOT part of the reference: rand::thread_rng is seeded by the OS, so the reference never renders
 * the same image twice, and any source of uniform numbers behind the Sampler trait
 * (src/sampler.rs:6-23) is a valid one.  This source returns, call for call, what the device's
 * sampler returns for (seed, pixel, sample): a hash for every draw, and Kensler's hash permutation
 * ("Correlated Multi-Jittered Sampling", Pixar TM 13-01) in place of the shuffled stratum tables.
 * With it the oracle traces the very paths the f64 device kernels trace, and the two renders can
 * be compared pixel by pixel instead of statistically. */
static uint32_t dev_mix32(uint32_t x) {
    x ^= x >> 16; x *= 0x7feb352dU; x ^= x >> 15; x *= 0x846ca68bU; x ^= x >> 16;
    return x;
}
static uint32_t dev_permute(uint32_t i, uint32_t l, uint32_t w, uint32_t p) {
    do {
        i ^= p; i *= 0xe170893dU; i ^= p >> 16; i ^= (i & w) >> 4; i ^= p >> 8; i *= 0x0929eb3fU;
        i ^= p >> 23; i ^= (i & w) >> 1; i *= 1U | p >> 27; i *= 0x6935fa69U; i ^= (i & w) >> 11;
        i *= 0x74dcb303U; i ^= (i & w) >> 2; i *= 0x9e501cc3U; i ^= (i & w) >> 2; i *= 0xc860a3dfU;
        i &= w; i ^= i >> 5;
    } while (i >= l);
    uint32_t r = i + p;
    return (l & (l - 1)) == 0 ? (r & w) : (r % l);
}
static double dev_uniform(sampler_t* s) {
    uint32_t bits = dev_mix32(s->key + (s->ctr++) * 0x9E3779B9U);
    if (s->device_numbers == 2) return (double)(bits >> 8) * 5.9604644775390625e-8; /* 2^-24, the f32 kernels */
    return (double)bits * 2.3283064365386962890625e-10; /* 2^-32 */
}
static void dev_start_sample(sampler_t* s) {
    s->key = dev_mix32(s->pixkey + (uint32_t)s->current_sample * 0x9E3779B9U);
    s->ctr = 0;
    s->current_dimension = s->type == RRT_SAMPLER_STRATIFIED ? 0 : SAMPLER_DIMS;
}
static void dev_start_pixel(sampler_t* s) {
    s->pixkey = dev_mix32(s->seed_lo ^ dev_mix32(s->pixel * 0x9E3779B1U + (dev_mix32(s->seed_hi) | 1U)));
    uint32_t w = (uint32_t)s->spp - 1; /* next power of two - 1 */
    w |= w >> 1; w |= w >> 2; w |= w >> 4; w |= w >> 8; w |= w >> 16;
    s->perm_mask = w;
    s->current_sample = 0;
    dev_start_sample(s);
}
static double dev_get_1d(sampler_t* s) {
    if (s->current_dimension < SAMPLER_DIMS) {
        uint32_t d = (uint32_t)s->current_dimension++;
        uint32_t k = dev_mix32(s->pixkey ^ ((2u * d + 1u) * 0x9E3779B9U));
        uint32_t stratum = dev_permute((uint32_t)s->current_sample, (uint32_t)s->spp, s->perm_mask, k);
        return ((double)stratum + dev_uniform(s)) / (double)s->spp;
    }
    return dev_uniform(s);
}
static void dev_get_2d(sampler_t* s, double* a, double* b) {
    double j0 = dev_uniform(s), j1 = dev_uniform(s);
    if (s->current_dimension < SAMPLER_DIMS) {
        uint32_t d = (uint32_t)s->current_dimension++;
        uint32_t k = dev_mix32(s->pixkey ^ ((2u * d + 2u) * 0x9E3779B9U));
        uint32_t cell = dev_permute((uint32_t)s->current_sample, (uint32_t)s->spp, s->perm_mask, k);
        uint32_t yc = cell / (uint32_t)s->x_strata, xc = cell - yc * (uint32_t)s->x_strata;
        *a = ((double)xc + j0) / (double)s->x_strata;
        *b = ((double)yc + j1) / (double)s->y_strata;
        return;
    }
    *a = j0;
    *b = j1;
}

static int sampler_init(sampler_t* s, const rrt_render_desc* rd, uint64_t seed, uint64_t stream) {
    memset(s, 0, sizeof(*s));
    s->type = (int)rd->sampler_type;
    s->spp = rd->spp;
    s->x_strata = rd->x_strata;
    s->y_strata = rd->y_strata;
    rng_seed(&s->rng, seed, stream);
    if (s->type == RRT_SAMPLER_STRATIFIED) {
        for (int d = 0; d < SAMPLER_DIMS; ++d) {
            s->s1d[d] = (double*)malloc(sizeof(double) * s->spp);
            s->s2d[d] = (double*)malloc(sizeof(double) * 2 * s->spp);
            if (!s->s1d[d] || !s->s2d[d]) return -1;
        }
    }
    return 0;
}
static void sampler_free(sampler_t* s) {
    for (int d = 0; d < SAMPLER_DIMS; ++d) { free(s->s1d[d]); free(s->s2d[d]); }
}

/* StratifiedSampler::generate_samples  src/sampler.rs:105-125 */
static void generate_samples(sampler_t* s) {
    size_t n = s->spp;
    for (int d = 0; d < SAMPLER_DIMS; ++d) {
        double* a = s->s1d[d];
        for (size_t i = 0; i < n; ++i) a[i] = ((double)i + rng_f64(&s->rng)) / (double)n;
        for (size_t i = n; i-- > 1;) { /* shuffle: for i in (1..len).rev() swap(i, gen_range(0..=i)) */
            size_t j = (size_t)rng_below(&s->rng, (uint64_t)i + 1);
            double tmp = a[i]; a[i] = a[j]; a[j] = tmp;
        }
    }
    for (int d = 0; d < SAMPLER_DIMS; ++d) {
        double* a = s->s2d[d];
        for (size_t i = 0; i < n; ++i) {
            size_t x_cell = i % s->x_strata;
            size_t y_cell = i / s->x_strata;
            double x = ((double)x_cell + rng_f64(&s->rng)) / (double)s->x_strata;
            double y = ((double)y_cell + rng_f64(&s->rng)) / (double)s->y_strata;
            a[2 * i] = x;
            a[2 * i + 1] = y;
        }
        for (size_t i = n; i-- > 1;) {
            size_t j = (size_t)rng_below(&s->rng, (uint64_t)i + 1);
            double tx = a[2 * i], ty = a[2 * i + 1];
            a[2 * i] = a[2 * j]; a[2 * i + 1] = a[2 * j + 1];
            a[2 * j] = tx; a[2 * j + 1] = ty;
        }
    }
}

/* start_pixel  src/sampler.rs:36-38 (Random), 129-133 (Stratified) */
static void sampler_start_pixel(sampler_t* s) {
    if (s->device_numbers) { dev_start_pixel(s); return; }
    s->current_sample = 0;
    if (s->type == RRT_SAMPLER_STRATIFIED) {
        s->current_dimension = 0;
        generate_samples(s);
    }
}
/* get_1d  src/sampler.rs:40-42, 135-142 */
static double sampler_get_1d(sampler_t* s) {
    s->draws++;
    if (s->device_numbers) return dev_get_1d(s);
    if (s->type == RRT_SAMPLER_RANDOM || s->current_dimension >= SAMPLER_DIMS)
        return rng_f64(&s->rng);
    double v = s->s1d[s->current_dimension][s->current_sample];
    s->current_dimension += 1;
    return v;
}
/* get_2d  src/sampler.rs:44-46, 144-151 */
static void sampler_get_2d(sampler_t* s, double* a, double* b) {
    s->draws++;
    if (s->device_numbers) { dev_get_2d(s, a, b); return; }
    if (s->type == RRT_SAMPLER_RANDOM || s->current_dimension >= SAMPLER_DIMS) {
        *a = rng_f64(&s->rng);
        *b = rng_f64(&s->rng);
        return;
    }
    const double* p = s->s2d[s->current_dimension] + 2 * s->current_sample;
    *a = p[0];
    *b = p[1];
    s->current_dimension += 1;
}
/* start_next_sample  src/sampler.rs:48-55 (Random: spp+1 samples in total),
 * 153-161 (Stratified: exactly spp) */
static int sampler_start_next_sample(sampler_t* s) {
    if (s->type == RRT_SAMPLER_RANDOM) {
        if (s->current_sample < s->spp) {
            s->current_sample += 1;
            if (s->device_numbers) dev_start_sample(s);
            return 1;
        }
        return 0;
    }
    if (s->current_sample < s->spp - 1) {
        s->current_sample += 1;
        s->current_dimension = 0;
        if (s->device_numbers) dev_start_sample(s);
        return 1;
    }
    return 0;
}

/* Positions a freshly started pixel on sample index `index` (the head of a sample-index shard). */
static void sampler_seek(sampler_t* s, size_t index) {
    s->current_sample = index;
    if (s->device_numbers) dev_start_sample(s);
}

/* ----------------------------------------------------------- material.rs */
static const double ORACLE_PI = 3.14159265358979323846264338327950288;
static const double FRAC_1_PI = 0.318309886183790671537767526745028724;

/* Material::emission  src/material.rs:33-35,52-54 */
static v3 mat_emission(const rrt_material* m) {
    return m->type == RRT_MAT_EMISSIVE ? ld3(m->rgb) : V(0, 0, 0);
}

/* spherical_to_world over the basis the f32 device kernels build around w (Duff et al., "Building an
 * Orthonormal Basis, Revisited", JCGT 2017) instead of local_coordinate_system (src/math.rs:162-172).
 * Both lobe samplers are symmetric in phi and phi is uniform, so the sampled distribution is the
 * same; only used with the device's numbers (device_numbers == 2), to follow the f32 kernels' paths. */
static v3 to_world_f32_kernels(double theta, double phi, v3 w) {
    double sign = w.z >= 0.0 ? 1.0 : -1.0;
    double a = -1.0 / (sign + w.z);
    double b = w.x * w.y * a;
    v3 u = V(1.0 + sign * w.x * w.x * a, sign * b, -sign * w.x);
    v3 v = V(b, sign + w.y * w.y * a, -w.y);
    double cu = sin(theta) * cos(phi), cv = sin(theta) * sin(phi), cw = cos(theta);
    return V(u.x * cu + v.x * cv + w.x * cw, u.y * cu + v.y * cv + w.y * cw, u.z * cu + v.z * cv + w.z * cw);
}

/* Material::scatter; returns 0 for None.  src/material.rs:44-46 (Emissive),
 * 68-86 (Lambertian), 105-127 (PhongSpecular), 157-170 (IdealReflector),
 * 194-241 (IdealDielectric) */
static int mat_scatter(const rrt_material* m, const ray_t* ray_in, v3 hit_point, v3 normal,
                       sampler_t* sampler, ray_t* out, double* pdf) {
    switch (m->type) {
    case RRT_MAT_EMISSIVE:
        return 0;
    case RRT_MAT_LAMBERTIAN: {
        double u, v;
        sampler_get_2d(sampler, &u, &v);
        double theta = acos(sqrt(1.0 - u));
        double phi = 2.0 * ORACLE_PI * v;
        v3 nd;
        if (sampler->device_numbers == 2) {
            /* the f32 device kernels' cosine lobe: the normal plus a uniform point on the unit sphere (height 1 - 2u,
             * azimuth 2 pi v), normalised -- the same density cos / pi as the reference's mapping */
            double r = 2.0 * sqrt(fmax(u - u * u, 0.0));
            v3 w = V(normal.x + r * cos(phi), normal.y + r * sin(phi), normal.z + (1.0 - 2.0 * u));
            nd = vmul(w, 1.0 / sqrt(vdot(w, w)));
        } else {
            nd = spherical_to_world(theta, phi, normal);
        }
        out->o = hit_point; out->d = nd;
        *pdf = vdot(nd, normal) * FRAC_1_PI;
        return 1;
    }
    case RRT_MAT_PHONG: {
        double shininess = m->param;
        v3 reflected = reflect(ray_in->d, normal);
        double u, v;
        sampler_get_2d(sampler, &u, &v);
        double theta = acos(pow(u, 1.0 / (shininess + 1.0)));
        double phi = 2.0 * ORACLE_PI * v;
        v3 nd = sampler->device_numbers == 2 ? to_world_f32_kernels(theta, phi, reflected) : spherical_to_world(theta, phi, reflected);
        out->o = hit_point; out->d = nd;
        *pdf = pow(vdot(nd, reflected), shininess) * (shininess + 1.0) * FRAC_1_PI * 0.5;
        return 1;
    }
    case RRT_MAT_REFLECTOR: {
        out->o = hit_point; out->d = reflect(ray_in->d, normal);
        *pdf = 1.0;
        return 1;
    }
    case RRT_MAT_DIELECTRIC: {
        v3 outward_normal = normal;
        double eta_i = 1.0, eta_t = m->param;
        if (vdot(ray_in->d, normal) > 0.0) {
            eta_i = m->param; eta_t = 1.0;
            outward_normal = vneg(normal);
        }
        double eta = eta_i / eta_t;
        v3 unit_direction = vnormalize(ray_in->d);
        double cos_theta = vdot(vneg(unit_direction), outward_normal);
        double r = sampler_get_1d(sampler);
        double reflectance = fresnel(cos_theta, eta_i, eta_t);
        if (r < reflectance) {
            out->o = hit_point; out->d = reflect(unit_direction, outward_normal);
            *pdf = reflectance;
            return 1;
        }
        v3 refracted;
        if (!refract(unit_direction, outward_normal, eta, &refracted)) return 0;
        out->o = hit_point; out->d = refracted;
        *pdf = 1.0 - reflectance;
        return 1;
    }
    default:
        return 0;
    }
}

/* Material::bxdf  src/material.rs:48-50, 88-90, 129-141, 172-180, 243-283 */
static v3 mat_bxdf(const rrt_material* m, const ray_t* ray_in, const ray_t* ray_out, v3 normal) {
    switch (m->type) {
    case RRT_MAT_LAMBERTIAN:
        return vmul(ld3(m->rgb), FRAC_1_PI);
    case RRT_MAT_PHONG: {
        v3 reflected = reflect(ray_in->d, normal);
        double cos_theta = vdot(reflected, ray_out->d);
        if (cos_theta < 0.0) return V(0, 0, 0);
        return vmul(vmul(vmul(vmul(ld3(m->rgb), m->param + 2.0), FRAC_1_PI), 0.5),
                    pow(cos_theta, m->param));
    }
    case RRT_MAT_REFLECTOR: {
        v3 reflected = reflect(ray_in->d, normal);
        double cos_theta = vdot(ray_out->d, normal);
        if (cos_theta > 1e-6 && vmag2(vsub(ray_out->d, reflected)) < 1e-6)
            return vdivs(V(1, 1, 1), cos_theta);
        return V(0, 0, 0);
    }
    case RRT_MAT_DIELECTRIC: {
        v3 outward_normal = normal;
        double eta_i = 1.0, eta_t = m->param;
        if (vdot(ray_in->d, normal) > 0.0) {
            eta_i = m->param; eta_t = 1.0;
            outward_normal = vneg(normal);
        }
        double eta = eta_i / eta_t;
        double cos_theta_i = fabs(vdot(ray_in->d, normal));
        double cos_theta_t = fabs(vdot(ray_out->d, normal));
        if (cos_theta_t < 1e-6) return V(0, 0, 0);
        double reflectance = fresnel(cos_theta_i, eta_i, eta_t);
        double transmittance = 1.0 - reflectance;
        v3 reflect_dir = reflect(ray_in->d, outward_normal);
        v3 refract_dir;
        if (!refract(ray_in->d, outward_normal, eta, &refract_dir)) refract_dir = V(0, 0, 0);
        v3 bxdf = V(0, 0, 0);
        if (vmag2(vsub(reflect_dir, ray_out->d)) < 1e-6)
            bxdf = vdivs(vmul(V(1, 1, 1), reflectance), cos_theta_t);
        else if (vmag2(vsub(refract_dir, ray_out->d)) < 1e-6)
            bxdf = vdivs(vmul(V(1, 1, 1), transmittance), cos_theta_t * eta * eta);
        return bxdf;
    }
    default:
        return V(0, 0, 0);
    }
}

/* ------------------------------------------------------------ tracers/ */
typedef struct {
    uint64_t rays, bounces[5], nonfinite;
    uint64_t rays_live; /* closest-hit queries made with a non-zero throughput */
} trace_counters;

/* ---- next-event estimation (rrt_render_desc.flags & RRT_RENDER_NEE).  NOT part of the reference: its connect()
 * panics for s > 0 (src/tracers/utils.rs:122-124).  This is the CPU statement of the estimator include/rrt.h
 * describes, written for checking the device kernels: same light table, same draws in the same order. */
enum { LIGHT_TRIANGLE = 0, LIGHT_PGRAM = 1, LIGHT_SPHERE = 2 };
typedef struct {
    int type;
    v3 v0, e1, e2, n, le; /* sphere: v0 = centre, e1.x = radius */
    double inv_pdf;       /* area / probability of choosing this light */
    int32_t object_id, prim_id; /* the primitive it is (a part of) */
} light_t;
typedef struct {
    light_t* l;
    float* cdf;
    size_t n;
} light_table;

static void lights_push_planar(light_table* lt, double* weight, double* area, int type, v3 v0, v3 e1, v3 e2, const rrt_material* m, const rrt_prim* p) {
    v3 c = vcross(e1, e2);
    double len = vmag(c);
    double a = type == LIGHT_TRIANGLE ? 0.5 * len : len;
    if (!(a > 0.0)) return;
    light_t* l = &lt->l[lt->n];
    memset(l, 0, sizeof(*l));
    l->type = type;
    l->v0 = v0; l->e1 = e1; l->e2 = e2;
    l->n = vmul(c, 1.0 / len);
    l->le = ld3(m->rgb);
    l->object_id = (int32_t)p->object_id;
    l->prim_id = (int32_t)p->prim_id;
    area[lt->n] = a;
    weight[lt->n] = a * fmax(m->rgb[0], fmax(m->rgb[1], m->rgb[2]));
    lt->n++;
}

/* every primitive of an emitting object in scan order; a parallelogram is one light, any other quad the two triangles
 * (v0, v1, v3) and (v2, v3, v1) of the reference's test (src/shapes/quadrilateral.rs:56-117); chosen in proportion
 * to area x largest emission component */
static int lights_build(const rrt_scene_desc* sc, light_table* lt) {
    memset(lt, 0, sizeof(*lt));
    lt->l = (light_t*)calloc(2 * (size_t)sc->n_prims + 1, sizeof(light_t));
    lt->cdf = (float*)calloc(2 * (size_t)sc->n_prims + 1, sizeof(float));
    double* weight = (double*)calloc(2 * (size_t)sc->n_prims + 1, sizeof(double));
    double* area = (double*)calloc(2 * (size_t)sc->n_prims + 1, sizeof(double));
    if (!lt->l || !lt->cdf || !weight || !area) { free(weight); free(area); return -1; }
    for (uint32_t i = 0; i < sc->n_prims; ++i) {
        const rrt_prim* p = &sc->prims[i];
        const rrt_material* m = &sc->materials[p->object_id];
        if (m->type != RRT_MAT_EMISSIVE || !(vmag(ld3(m->rgb)) > 1e-6)) continue;
        const double* v = p->v;
        if (p->type == RRT_PRIM_PLANE) { free(weight); free(area); return -2; } /* no area to sample */
        if (p->type == RRT_PRIM_SPHERE) {
            double a = 4.0 * 3.14159265358979323846 * v[3] * v[3];
            if (!(a > 0.0)) continue;
            light_t* l = &lt->l[lt->n];
            memset(l, 0, sizeof(*l));
            l->type = LIGHT_SPHERE;
            l->v0 = ld3(v);
            l->e1 = V(v[3], 0, 0);
            l->le = ld3(m->rgb);
            l->object_id = (int32_t)p->object_id;
            l->prim_id = (int32_t)p->prim_id;
            area[lt->n] = a;
            weight[lt->n] = a * fmax(m->rgb[0], fmax(m->rgb[1], m->rgb[2]));
            lt->n++;
        } else if (p->type == RRT_PRIM_TRIANGLE) {
            lights_push_planar(lt, weight, area, LIGHT_TRIANGLE, ld3(v), vsub(ld3(v + 3), ld3(v)), vsub(ld3(v + 6), ld3(v)), m, p);
        } else {
            v3 v0 = ld3(v), v1 = ld3(v + 3), v2 = ld3(v + 6), v3_ = ld3(v + 9);
            v3 gap = vsub(vadd(v0, v2), vadd(v1, v3_));
            double scale = vmag(vsub(v1, v0)) + vmag(vsub(v3_, v0));
            if (vmag(gap) <= 1e-12 * scale) {
                lights_push_planar(lt, weight, area, LIGHT_PGRAM, v0, vsub(v1, v0), vsub(v3_, v0), m, p);
            } else {
                lights_push_planar(lt, weight, area, LIGHT_TRIANGLE, v0, vsub(v1, v0), vsub(v3_, v0), m, p);
                lights_push_planar(lt, weight, area, LIGHT_TRIANGLE, v2, vsub(v3_, v2), vsub(v1, v2), m, p);
            }
        }
    }
    double total = 0.0, run = 0.0;
    for (size_t i = 0; i < lt->n; ++i) total += weight[i];
    for (size_t i = 0; i < lt->n; ++i) {
        double prob = total > 0.0 ? weight[i] / total : 1.0 / (double)lt->n;
        run += prob;
        lt->l[i].inv_pdf = prob > 0.0 ? area[i] / prob : 0.0;
        lt->cdf[i] = i + 1 == lt->n ? 1.0f : (float)run;
    }
    free(weight);
    free(area);
    return 0;
}

static void lights_free(light_table* lt) {
    free(lt->l);
    free(lt->cdf);
    lt->l = NULL; lt->cdf = NULL; lt->n = 0;
}

/* f32_kernels: the device's f32 uniforms and its table are single precision; the point and the geometry terms are
 * still evaluated in f64 here */
static int light_sample(const light_table* lt, v3 x, double pick, double u, double v, v3* wi, double* dist, v3* le_w, const light_t** which) {
    if (lt->n == 0) return 0;
    size_t k = lt->n - 1;
    for (size_t i = 0; i < lt->n; ++i) if (pick < (double)lt->cdf[i]) { k = i; break; }
    const light_t* l = &lt->l[k];
    *which = l;
    v3 y, ny;
    if (l->type == LIGHT_SPHERE) {
        double z = 1.0 - 2.0 * u;
        double r = sqrt(fmax(1.0 - z * z, 0.0));
        double phi = 2.0 * ORACLE_PI * v;
        ny = V(r * cos(phi), r * sin(phi), z);
        y = vadd(l->v0, vmul(ny, l->e1.x));
    } else {
        double a = u, b = v;
        if (l->type == LIGHT_TRIANGLE) {
            double su = sqrt(u);
            a = su * (1.0 - v);
            b = su * v;
        }
        y = vadd(l->v0, vadd(vmul(l->e1, a), vmul(l->e2, b)));
        ny = l->n;
    }
    v3 to = vsub(y, x);
    double d2 = vdot(to, to);
    if (!(d2 > 0.0)) return 0;
    *dist = sqrt(d2);
    *wi = vmul(to, 1.0 / *dist);
    double w = fabs(vdot(ny, *wi)) * l->inv_pdf / d2;
    *le_w = vmul(l->le, w);
    return 1;
}

/* bxdf(wo, wi) |cos| of the lobe materials for a given wi, zero where mat_scatter could not have produced wi
 * (pdf <= 1e-6 is rejected by the tracer) */
static v3 lobe_bxdf_cos(const rrt_material* m, const ray_t* ray_in, v3 normal, v3 wi) {
    double k = 0.0;
    if (m->type == RRT_MAT_LAMBERTIAN) {
        double c = vdot(wi, normal) * FRAC_1_PI;
        if (c > 1e-6) k = c;
    } else {
        double c = vdot(wi, reflect(ray_in->d, normal));
        if (c > 0.0) {
            double cs = pow(c, m->param);
            if (cs * (m->param + 1.0) * FRAC_1_PI * 0.5 > 1e-6) k = cs * (m->param + 2.0) * FRAC_1_PI * 0.5 * fabs(vdot(wi, normal));
        }
    }
    return vmul(ld3(m->rgb), k);
}

/* MonteCarloPathTracer::trace  src/tracers/mcpt.rs:20-43 over
 * generate_camera_vertices  src/tracers/utils.rs:16-91 and connect (s = 0)
 * src/tracers/utils.rs:101-127.  Vertex generation stops at the first emissive
 * vertex, and connect(s=0, t) adds beta * emission only for an emissive vertex
 * t-1 (depth = t-2 never exceeds max_depth-1), so the sum over t is
 * beta * emission of the last vertex if it is emissive, else zero. */
static v3 trace(const rrt_scene_desc* sc, const ray_t* camera_ray, sampler_t* sampler,
                uint32_t min_depth, uint32_t max_depth, trace_counters* cnt, const light_table* lights) {
    v3 beta = V(1.0, 1.0, 1.0);
    ray_t ray = *camera_ray;
    v3 color = V(0, 0, 0);
    int lights_sampled = 0; /* NEE: the previous vertex has already counted the lights */
    for (uint32_t depth = 0; depth < max_depth; ++depth) {
        hit_t hit;
        cnt->rays++;
        if (beta.x != 0.0 || beta.y != 0.0 || beta.z != 0.0) cnt->rays_live++;
        if (!scene_intersect(sc, &ray, &hit)) break;
        const rrt_material* material = &sc->materials[hit.object_id];
        v3 emission = mat_emission(material);
        if (vmag(emission) > 1e-6) {
            if (!(lights && lights_sampled)) color = vadd(color, vmulv(beta, emission));
            break;
        }
        double continue_prob = depth > min_depth ? fmin(max_component(beta), 1.0) : 1.0;
        if (sampler_get_1d(sampler) > continue_prob) break;
        beta = vdivs(beta, continue_prob);
        ray_t sr;
        double pdf;
        cnt->bounces[material->type < 5 ? material->type : 0]++;
        if (lights && depth + 1 < max_depth) {
            /* (a vertex whose scattered ray would never be traced connects to nothing either) */
            if (material->type == RRT_MAT_LAMBERTIAN || material->type == RRT_MAT_PHONG) {
                double pick = sampler_get_1d(sampler), lu, lv;
                sampler_get_2d(sampler, &lu, &lv);
                v3 wi, le_w;
                double dist;
                const light_t* l;
                if (light_sample(lights, hit.p, pick, lu, lv, &wi, &dist, &le_w, &l)) {
                    v3 f = lobe_bxdf_cos(material, &ray, hit.normal, wi);
                    if (f.x != 0.0 || f.y != 0.0 || f.z != 0.0) {
                        ray_t shadow;
                        shadow.o = hit.p; shadow.d = wi;
                        hit_t sh;
                        cnt->rays++;
                        if (beta.x != 0.0 || beta.y != 0.0 || beta.z != 0.0) cnt->rays_live++;
                        /* unblocked: nothing on the way, or the first thing on the way is the flat light's own primitive (a
                         * bent quad reports the distance to its first triangle's plane, not to the sampled point; a sphere
                         * hides its own far side), or it is not nearer than the point */
                        if (!scene_intersect(sc, &shadow, &sh) || (l->type != LIGHT_SPHERE && sh.object_id == l->object_id && sh.prim_id == l->prim_id) ||
                            sh.t >= dist * (1.0 - 1e-4))
                            color = vadd(color, vmulv(vmulv(beta, f), le_w));
                    }
                }
                lights_sampled = 1;
            } else {
                lights_sampled = 0;
            }
        }
        if (!mat_scatter(material, &ray, hit.p, hit.normal, sampler, &sr, &pdf)) break;
        if (pdf <= 1e-6) break;
        double cos_theta = fabs(vdot(sr.d, hit.normal));
        v3 bxdf = mat_bxdf(material, &ray, &sr, hit.normal);
        beta = vmulv(beta, vdivs(vmul(bxdf, cos_theta), pdf));
        ray = sr;
    }
    if (!vfinite(color)) cnt->nonfinite++;
    return color;
}

/* ----------------------------------------------------------- renderer.rs */
typedef struct {
    const rrt_scene_desc* scene;
    const rrt_render_desc* rd;
    double* rgb_sum;
    size_t tiles_x, tile_count;
    size_t next_tile;
    pthread_mutex_t lock;
    uint64_t samples, rays, rays_live, bounces[5], nonfinite;
    int failed;
    int device_numbers;
    const light_table* lights; /* NULL: the reference's estimator */
} render_job;

/* One 16x16 tile: src/renderer.rs:104-140 */
static void render_tile(render_job* job, size_t tile_index, trace_counters* cnt, uint64_t* samples) {
    const rrt_render_desc* rd = job->rd;
    const size_t tile_size = 16;
    size_t tile_x = tile_index % job->tiles_x;
    size_t tile_y = tile_index / job->tiles_x;
    size_t x_start = tile_x * tile_size, y_start = tile_y * tile_size;
    size_t x_end = x_start + tile_size < rd->width ? x_start + tile_size : rd->width;
    size_t y_end = y_start + tile_size < rd->height ? y_start + tile_size : rd->height;
    sampler_t sampler;
    if (sampler_init(&sampler, rd, rd->seed, (uint64_t)tile_index) != 0) {
        job->failed = 1;
        sampler_free(&sampler);
        return;
    }
    sampler.device_numbers = job->device_numbers;
    sampler.seed_lo = (uint32_t)rd->seed;
    sampler.seed_hi = (uint32_t)(rd->seed >> 32);
    /* The reference's loop (src/renderer.rs:118-127) runs over every sample of the pixel: spp for the stratified
     * sampler, spp + 1 for the random one (src/sampler.rs:48-55).  A device backend shards that loop by sample index
     * (rrt_render_desc.sample_begin / sample_end); the same range is honoured here so that a shard can be checked
     * on its own.  sample_end == 0 means "all".  With the device's numbers every sample's draws depend on
     * (seed, pixel, sample index) only, so the shards of a pixel add up to its full loop exactly. */
    const uint64_t s_total = (uint64_t)rd->spp + (rd->sampler_type == RRT_SAMPLER_RANDOM ? 1u : 0u);
    const uint64_t s_end = rd->sample_end && rd->sample_end < s_total ? rd->sample_end : s_total;
    const uint64_t s_begin = rd->sample_begin;
    for (size_t y = y_start; y < y_end; ++y) {
        for (size_t x = x_start; x < x_end; ++x) {
            sampler.pixel = (uint32_t)(y * rd->width + x);
            sampler_start_pixel(&sampler);
            v3 color = V(0, 0, 0);
            double* out = job->rgb_sum + 3 * (y * rd->width + x);
            if (s_begin >= s_end) { out[0] = out[1] = out[2] = 0.0; continue; }
            sampler_seek(&sampler, (size_t)s_begin);
            for (;;) {
                double u_offset, v_offset;
                sampler_get_2d(&sampler, &u_offset, &v_offset);
                double u = ((double)x + u_offset + 0.5) / (double)rd->width;
                double v = 1.0 - ((double)y + v_offset + 0.5) / (double)rd->height;
                ray_t ray = create_ray(&job->scene->camera, u, v);
                color = vadd(color, trace(job->scene, &ray, &sampler, rd->min_depth, rd->max_depth, cnt, job->lights));
                (*samples)++;
                if (sampler.current_sample + 1 >= s_end) break;
                if (!sampler_start_next_sample(&sampler)) break;
            }
            out[0] = color.x; out[1] = color.y; out[2] = color.z;
        }
    }
    sampler_free(&sampler);
}

static void* render_worker(void* arg) {
    render_job* job = (render_job*)arg;
    trace_counters cnt;
    memset(&cnt, 0, sizeof(cnt));
    uint64_t samples = 0;
    for (;;) {
        pthread_mutex_lock(&job->lock);
        size_t tile = job->next_tile++;
        pthread_mutex_unlock(&job->lock);
        if (tile >= job->tile_count) break;
        render_tile(job, tile, &cnt, &samples);
    }
    pthread_mutex_lock(&job->lock);
    job->samples += samples;
    job->rays += cnt.rays;
    job->rays_live += cnt.rays_live;
    job->nonfinite += cnt.nonfinite;
    for (int i = 0; i < 5; ++i) job->bounces[i] += cnt.bounces[i];
    pthread_mutex_unlock(&job->lock);
    return NULL;
}

/* The tile loop of render  src/renderer.rs:92-141; rgb_sum receives the
 * per-pixel SUM over samples (the division by spp is in oracle_postprocess).
 * Threads pull tiles from a shared counter, as rayon's par_iter does. */
static int render_accumulate_impl(const rrt_scene_desc* scene, const rrt_render_desc* rd,
                                  double* rgb_sum, rrt_stats* stats, int n_threads, int device_numbers);
int oracle_render_accumulate(const rrt_scene_desc* scene, const rrt_render_desc* rd,
                             double* rgb_sum, rrt_stats* stats, int n_threads) {
    return render_accumulate_impl(scene, rd, rgb_sum, stats, n_threads, 0);
}
/* The same render with the numbers of the HIP backend's generator (see dev_mix32 above): the image the
 * f64 device kernels must reproduce pixel by pixel for the same seed (f32_kernels = 0), or the one the
 * f32 kernels follow up to their rounding (f32_kernels = 1: 24-bit uniforms, their lobe basis). */
int oracle_render_accumulate_device_numbers(const rrt_scene_desc* scene, const rrt_render_desc* rd,
                                            double* rgb_sum, rrt_stats* stats, int n_threads, int f32_kernels) {
    return render_accumulate_impl(scene, rd, rgb_sum, stats, n_threads, f32_kernels ? 2 : 1);
}
static int render_accumulate_impl(const rrt_scene_desc* scene, const rrt_render_desc* rd,
                                  double* rgb_sum, rrt_stats* stats, int n_threads, int device_numbers) {
    if (!scene || !rd || !rgb_sum || rd->width == 0 || rd->height == 0 || rd->spp == 0) return -1;
    if (rd->sampler_type == RRT_SAMPLER_STRATIFIED &&
        (uint64_t)rd->x_strata * rd->y_strata != rd->spp)
        return -1; /* assert_eq! in StratifiedSampler::new  src/sampler.rs:87-91 */
    render_job job;
    memset(&job, 0, sizeof(job));
    job.scene = scene;
    job.rd = rd;
    job.rgb_sum = rgb_sum;
    job.device_numbers = device_numbers;
    light_table lights;
    memset(&lights, 0, sizeof(lights));
    if (rd->flags & RRT_RENDER_NEE) {
        if (lights_build(scene, &lights) != 0) { lights_free(&lights); return -1; }
        job.lights = &lights;
    }
    job.tiles_x = (rd->width + 15) / 16;
    job.tile_count = job.tiles_x * ((rd->height + 15) / 16);
    pthread_mutex_init(&job.lock, NULL);
    if (n_threads < 1) n_threads = 1;
    if (n_threads > 1024) n_threads = 1024;
    pthread_t* th = (pthread_t*)malloc(sizeof(pthread_t) * (size_t)n_threads);
    int started = 0;
    for (int i = 0; i < n_threads; ++i) {
        if (pthread_create(&th[i], NULL, render_worker, &job) != 0) break;
        started++;
    }
    if (started == 0) render_worker(&job);
    for (int i = 0; i < started; ++i) pthread_join(th[i], NULL);
    free(th);
    pthread_mutex_destroy(&job.lock);
    lights_free(&lights);
    if (stats) {
        memset(stats, 0, sizeof(*stats));
        stats->samples = job.samples;
        stats->rays = job.rays;
        /* The reference keeps tracing a path whose throughput has become exactly zero (red wall then
         * green wall) until Russian roulette or max_depth stops it.  `reserved` reports the queries
         * made with non-zero throughput, for comparison with a backend that drops such paths. */
        stats->reserved = (double)job.rays_live;
        stats->nonfinite = job.nonfinite;
        for (int i = 0; i < 5; ++i) stats->bounces[i] = job.bounces[i];
    }
    return job.failed ? -1 : 0;
}

/* Rust `x as u8` on f64: saturating, NaN -> 0 */
static uint8_t f64_as_u8(double x) {
    if (isnan(x)) return 0;
    if (x <= 0.0) return 0;
    if (x >= 255.0) return 255;
    return (uint8_t)x;
}
/* Rust f64::min: if one side is NaN the other is returned */
static double rust_min(double a, double b) {
    if (isnan(a)) return b;
    if (isnan(b)) return a;
    return a < b ? a : b;
}

/* color /= spp, post_process, quantise: src/renderer.rs:39-71,128-136 */
int oracle_postprocess(const double* rgb_sum, uint64_t spp_divisor, const rrt_post_desc* post,
                       uint32_t width, uint32_t height, uint8_t* rgb8, double* rgb_linear) {
    size_t n = (size_t)width * height;
    for (size_t i = 0; i < n; ++i) {
        double c[3];
        for (int k = 0; k < 3; ++k) {
            double x = rgb_sum[3 * i + k] / (double)spp_divisor;
            if (post->tone_mapping == 1) x = x / (x + 1.0);
            if (post->gamma_correction) x = pow(x, 1.0 / 2.2);
            if (post->has_white_balance) x = x * post->white_balance[k];
            c[k] = x;
            if (rgb_linear) rgb_linear[3 * i + k] = x;
            rgb8[3 * i + k] = f64_as_u8(rust_min(c[k] * 255.0, 255.0));
        }
    }
    return 0;
}

/* ------------------------------------------- unit-level entry points (tests) */
void oracle_ray_at(const double o[3], const double d[3], double t, double out[3]) {
    ray_t r; r.o = ld3(o); r.d = ld3(d);
    st3(out, ray_at(&r, t));
}
void oracle_create_ray(const rrt_camera* cam, double s, double t, double o[3], double d[3]) {
    ray_t r = create_ray(cam, s, t);
    st3(o, r.o); st3(d, r.d);
}
void oracle_reflect(const double v[3], const double n[3], double out[3]) { st3(out, reflect(ld3(v), ld3(n))); }
int oracle_refract(const double v[3], const double n[3], double eta, double out[3]) {
    v3 r;
    if (!refract(ld3(v), ld3(n), eta, &r)) return 0;
    st3(out, r);
    return 1;
}
double oracle_fresnel(double cos_i, double eta_i, double eta_t) { return fresnel(cos_i, eta_i, eta_t); }
void oracle_local_coordinate_system(const double n[3], double u[3], double v[3], double w[3]) {
    v3 a, b, c;
    local_coordinate_system(ld3(n), &a, &b, &c);
    st3(u, a); st3(v, b); st3(w, c);
}
void oracle_spherical_to_world(double theta, double phi, const double n[3], double out[3]) {
    st3(out, spherical_to_world(theta, phi, ld3(n)));
}
/* One primitive, caller-chosen t range (the reference's shape tests use [0,100]). */
int oracle_prim_intersect(const rrt_prim* prim, const double o[3], const double d[3], double t_min,
                          double t_max, double* t, double p[3], double normal[3]) {
    ray_t r; r.o = ld3(o); r.d = ld3(d);
    hit_t h;
    if (!prim_intersect(prim, &r, t_min, t_max, &h)) return 0;
    *t = h.t; st3(p, h.p); st3(normal, h.normal);
    return 1;
}
/* Sphere::intersect_analytic  src/shapes/sphere.rs:26-56 (dead code in the
 * reference, kept only because test_sphere_intersect compares against it). */
int oracle_sphere_intersect_analytic(const double sphere[4], const double o[3], const double d[3],
                                     double t_min, double t_max, double* t) {
    v3 oc = vsub(ld3(o), ld3(sphere));
    v3 dir = ld3(d);
    double a = vmag2(dir);
    double half_b = vdot(oc, dir);
    double c = vmag2(oc) - sphere[3] * sphere[3];
    double discriminant = half_b * half_b - a * c;
    if (discriminant < 0.0) return 0;
    double sqrtd = sqrt(discriminant);
    double root = (-half_b - sqrtd) / a;
    if (root < t_min || root > t_max) {
        root = (-half_b + sqrtd) / a;
        if (root < t_min || root > t_max) return 0;
    }
    *t = root;
    return 1;
}
/* are_points_coplanar / is_quadrilateral_convex  src/shapes/quadrilateral.rs:23-54 */
int oracle_are_points_coplanar(const double q[12]) {
    v3 v0 = ld3(q), v1 = ld3(q + 3), v2 = ld3(q + 6), v3_ = ld3(q + 9);
    v3 e01 = vsub(v1, v0), e02 = vsub(v2, v0), e03 = vsub(v3_, v0);
    v3 n1 = vcross(e01, e02), n2 = vcross(e02, e03);
    double c = vdot(n1, n2) / (vmag(n1) * vmag(n2));
    return fabs(fabs(c) - 1.0) < 1e-3;
}
int oracle_is_quadrilateral_convex(const double q[12]) {
    if (!oracle_are_points_coplanar(q)) return 0;
    v3 v0 = ld3(q), v1 = ld3(q + 3), v2 = ld3(q + 6), v3_ = ld3(q + 9);
    v3 e01 = vsub(v1, v0), e12 = vsub(v2, v1), e23 = vsub(v3_, v2), e30 = vsub(v0, v3_);
    v3 c1 = vcross(e01, e12), c2 = vcross(e12, e23), c3 = vcross(e23, e30), c4 = vcross(e30, e01);
    return vdot(c1, c2) >= 0.0 && vdot(c2, c3) >= 0.0 && vdot(c3, c4) >= 0.0 && vdot(c4, c1) >= 0.0;
}
/* Material::scatter + bxdf with caller-supplied random numbers, for the
 * material tests: u1 is the get_1d value, (u2a,u2b) the get_2d value.  Returns
 * 0 for None.  weight = cos_theta * bxdf / pdf as src/tracers/utils.rs:76-82. */
int oracle_material_sample(const rrt_material* m, const double d_in[3], const double p[3],
                           const double normal[3], double u1, double u2a, double u2b,
                           double d_out[3], double* pdf, double bxdf[3], double weight[3]) {
    /* a one-sample "stratified" sampler whose tables hold the supplied numbers */
    sampler_t s;
    memset(&s, 0, sizeof(s));
    s.type = RRT_SAMPLER_STRATIFIED;
    s.spp = 1;
    double one_d[SAMPLER_DIMS] = {u1, u1, u1, u1};
    double two_d[SAMPLER_DIMS][2] = {{u2a, u2b}, {u2a, u2b}, {u2a, u2b}, {u2a, u2b}};
    for (int d = 0; d < SAMPLER_DIMS; ++d) { s.s1d[d] = &one_d[d]; s.s2d[d] = two_d[d]; }
    ray_t in, out;
    in.o = V(0, 0, 0); in.d = ld3(d_in);
    if (!mat_scatter(m, &in, ld3(p), ld3(normal), &s, &out, pdf)) return 0;
    v3 f = mat_bxdf(m, &in, &out, ld3(normal));
    double cos_theta = fabs(vdot(out.d, ld3(normal)));
    st3(d_out, out.d);
    st3(bxdf, f);
    st3(weight, vdivs(vmul(f, cos_theta), *pdf));
    return 1;
}
/* The stratified tables of one pixel, for the sampler tests:
 * out1d[4][spp], out2d[4][spp][2]. */
int oracle_stratified_tables(uint32_t spp, uint32_t x_strata, uint32_t y_strata, uint64_t seed,
                             double* out1d, double* out2d) {
    rrt_render_desc rd;
    memset(&rd, 0, sizeof(rd));
    rd.sampler_type = RRT_SAMPLER_STRATIFIED;
    rd.spp = spp; rd.x_strata = x_strata; rd.y_strata = y_strata;
    if ((uint64_t)x_strata * y_strata != spp) return -1;
    sampler_t s;
    if (sampler_init(&s, &rd, seed, 0) != 0) return -1;
    sampler_start_pixel(&s);
    for (int d = 0; d < SAMPLER_DIMS; ++d) {
        memcpy(out1d + (size_t)d * spp, s.s1d[d], sizeof(double) * spp);
        memcpy(out2d + (size_t)d * spp * 2, s.s2d[d], sizeof(double) * 2 * spp);
    }
    sampler_free(&s);
    return 0;
}
/* Radiance of single camera samples at given (u, v), for analytic-scene tests. */
int oracle_trace_samples(const rrt_scene_desc* scene, const rrt_render_desc* rd, size_t n,
                         const double* u, const double* v, double* rgb) {
    rrt_render_desc one = *rd;
    one.sampler_type = RRT_SAMPLER_RANDOM;
    sampler_t s;
    if (sampler_init(&s, &one, rd->seed, 0x5eed) != 0) return -1;
    trace_counters cnt;
    memset(&cnt, 0, sizeof(cnt));
    for (size_t i = 0; i < n; ++i) {
        ray_t ray = create_ray(&scene->camera, u[i], v[i]);
        v3 c = trace(scene, &ray, &s, rd->min_depth, rd->max_depth, &cnt, NULL);
        st3(rgb + 3 * i, c);
    }
    sampler_free(&s);
    return 0;
}
