// Device-side data layout of a flattened scene, shared by host set-up code and kernels.
//
// A scene is uploaded once per device in two precisions (R = float, double).
// Records are derived from the f64 scene description on the host, in f64, and
// rounded to R once, so the f32 kernels do not inherit f32 set-up error.
#pragma once
#include <stdint.h>

namespace rrt {

// Magnitudes of one primitive's record for the error bounds of a tracked f32 closest-hit query (rrt_device.hpp, HitTrack).
struct ErrRec {
    float e[8];
};

enum : int {
    PRIM_SPHERE = 0, PRIM_PLANE = 1, PRIM_TRIANGLE = 2, PRIM_QUAD = 3,
    PRIM_PGRAM = 4  // device-only: a quad whose corners form a parallelogram (PrimRec.type only)
};
enum : int { MAT_EMISSIVE = 0, MAT_LAMBERTIAN = 1, MAT_PHONG = 2, MAT_REFLECTOR = 3, MAT_DIELECTRIC = 4 };

// Intersection record of one primitive: everything the per-ray test reads, in
// the order the reference's linear scan visits primitives.  In the small-scene
// kernels the index into this table is wave-uniform, so a record travels
// through the scalar cache into SGPRs and costs no vector register.
//
//   triangle / quad (plane form of Moller-Trumbore / Lagae-Dutre, see intersect.hpp)
//     f[0..2] n = e01 x e03 (unnormalised)       f[3]  n . v0
//     f[4..6] A,  f[7]  cA   alpha = A . P + cA   (coefficient of P - v0 on e01)
//     f[8..10] B, f[11] cB   beta  = B . P + cB   (coefficient on e03)
//     parallelogram (v0 + v2 == v1 + v3): cA and cB carry an extra -1/2, nothing else is read
//     general quad only, second corner v2:
//     f[12..14] A2, f[15] cA2   alpha' on e23 = v3 - v2
//     f[16..18] B2, f[19] cB2   beta'  on e21 = v1 - v2
//     f[20..22] n2 = e23 x e21, f[23] n2 . v2: the plane of the triangle (v2, v3, v1), which the second
//               corner test intersects on its own (it differs from the first plane when the quad is bent)
//   sphere   f[0..2] centre, f[3] radius, f[4] radius^2
//   plane    f[0..2] unit normal, f[3] normal . point
template <typename R>
struct PrimRec {
    R f[24];
    int32_t type;
};

// Two records of one run side by side (parallelograms, triangles; planes and spheres in runs of more than one):
// g[k] = (f[k] of the first, f[k] of the second), k = 0..11.  A wave-uniform pair lands in 24 SGPRs as 12 aligned pairs, which is the operand
// shape of the packed FP32 instructions (v_pk_fma_f32 & co. do two FMAs per lane per issue
// slot).  An odd run is padded with a record that every ray rejects (all zero: n . d = 0; sphere: radius^2 = -1).
template <typename R>
struct PairRec {
    R g[12][2];
};

// What shading needs from the primitive that was hit (one gather per bounce).
//   triangle / quad / plane: n = the unit normal the reference reports
//   sphere: n = centre, w = radius (normal = (P - centre) / radius)
template <typename R>
struct ShadeRec {
    R nx, ny, nz, w;
    int32_t object_id;
    int32_t prim_id;
    int32_t type;
    int32_t pad;
};

// One material per object (src/material.rs:286-315).
template <typename R>
struct MatRec {
    R r, g, b;   // emission / albedo / specular; dielectric: 1/ior, 1/(1/ior)^2, 1/ior^2 (eta and 1/eta^2, entering and leaving)
    R param;     // shininess / index of refraction
    int32_t type;
    int32_t emitter;  // |emission| > 1e-6  (src/tracers/utils.rs:52)
    int32_t pad[2];
};

// Binary LBVH node as the builder makes it (rrt_lbvh.hip): the boxes of both children and the
// child links.  child >= 0: internal node index.  child < 0: a leaf, ~child = first sorted slot |
// (count - 1) << 28, i.e. 1..8 primitives that are neighbours in the sorted order (a subtree small
// enough is folded into one leaf).  Boxes are f32, rounded outward and padded on the host, so a
// slab test never rejects a box whose primitive the ray test would accept.
constexpr int BVH_SLOT_BITS = 28;
constexpr int BVH_SLOT_MASK = (1 << BVH_SLOT_BITS) - 1;
constexpr int BVH_MAX_PRIMS = BVH_SLOT_MASK;  // so that ~child never equals INT_MIN
// BvhNode::pad0 of a tree built on the host: which descendants the 4-wide node rooted here absorbs (at most two).
enum : int32_t {
    SAH_OPEN_LEFT = 1, SAH_OPEN_RIGHT = 2,       // the node's own children
    SAH_OPEN_LEFT_0 = 4, SAH_OPEN_LEFT_1 = 8,     // children of the left child (only with SAH_OPEN_LEFT)
    SAH_OPEN_RIGHT_0 = 16, SAH_OPEN_RIGHT_1 = 32  // children of the right child (only with SAH_OPEN_RIGHT)
};
struct alignas(64) BvhNode {
    float lo0[3];
    int32_t c0;
    float hi0[3];
    int32_t c1;
    float lo1[3];
    int32_t pad0;
    float hi1[3];
    int32_t pad1;
};

// A primitive as tree leaves read it, in sorted-slot order: the first 12 numbers of its PrimRec
// (all a triangle or parallelogram test reads), its record type and its scan-order index.  One
// 64-byte line in f32.  Spheres and general quads are tested from their full PrimRec.
template <typename R>
struct alignas(sizeof(R) == 4 ? 64 : 128) LeafRec {
    R f[12];
    int32_t type;
    int32_t g;
    int32_t pad[2];
};

// What rays traverse: binary node i with its internal children expanded one level, i.e. up to four
// grandchildren of i side by side (same child encoding, BVH_NO_CHILD pads a node that has fewer),
// with the child boxes quantised to 8 bits per bound on a grid local to the node:
//   bound = org[axis] + q * 2^(e[axis] - 127),   lower bounds rounded down, upper bounds up,
// checked at build time against the very expression the traversal evaluates.  64 bytes: a divergent
// fetch costs the texture-address unit per lane and per 16 bytes, so the traversal kernel's speed
// is set by how few bytes a step needs, and 4-wide nodes halve the dependent fetches per ray.
// Indexed like the binary nodes; node 0 is the root.
constexpr int BVH_NO_CHILD = (int)0x80000000;
// Traversal stacks (rrt_bvh.hpp) hold BVH_STACK_ENTRIES node links per lane: the first BVH_LDS_LEVELS in LDS, the rest
// in a private array.  A walk pushes at most three links per level of 4-wide nodes it descends, so a tree of up to
// BVH_STACK_ENTRIES / 3 levels can never overflow a stack; the host measures the depth of every tree it builds and
// rebuilds a deeper one with median cuts (rrt_api.cpp, ensure_bvh).
constexpr int BVH_LDS_LEVELS = 12;
constexpr int BVH_STACK_ENTRIES = 144;
constexpr int BVH_SPILL_LEVELS = BVH_STACK_ENTRIES - BVH_LDS_LEVELS;
struct alignas(64) WideNode {
    float org[3];
    uint8_t e[3];
    uint8_t pad0;
    uint32_t qlo[3];  // byte k of qlo[axis] = lower bound of child k
    uint32_t qhi[3];
    int32_t child[4];
    int32_t pad1[2];
};

// One area light = one emissive primitive (a general quad gives two triangles), for next-event estimation
// (RRT_RENDER_NEE; the reference's connect() stops at s = 0, src/tracers/utils.rs:115-124).
//   LIGHT_TRIANGLE  point = v0 + (1 - sqrt(u)) e1 ... uniform in the triangle (v0, v0 + e1, v0 + e2)
//   LIGHT_PGRAM     point = v0 + u e1 + v e2
//   LIGHT_SPHERE    v0 = centre, e1[0] = radius; point uniform on the surface
// inv_pdf = area / (probability of choosing this light); lights are chosen with probability proportional to
// area x largest emission component (LightView::cdf).
enum : int { LIGHT_TRIANGLE = 0, LIGHT_PGRAM = 1, LIGHT_SPHERE = 2 };
template <typename R>
struct LightRec {
    R v0[3], e1[3], e2[3];
    R n[3];       // unit normal (planar lights)
    R le[3];      // emitted radiance
    R inv_pdf;
    int32_t type;
    int32_t prim;  // scan-order index of the primitive it is (a part of)
};

template <typename R>
struct SceneView {
    R cam_o[3];    // camera origin
    R cam_ll[3];   // lower_left_corner - origin
    R cam_h[3];    // horizontal
    R cam_v[3];    // vertical
    const PrimRec<R>* prims;   // n_prims, scan order
    const ShadeRec<R>* shade;  // n_prims
    const MatRec<R>* mats;     // n_objects
    int32_t n_prims;
    int32_t n_objects;
    const int32_t* runs;       // n_runs x (type, first, count, first pair): maximal runs of equal PrimRec.type
    const PairRec<R>* pairs;   // PairRec table of the parallelogram and triangle runs
    int32_t n_runs;
    int32_t pad0;
    // acceleration structure (n_bvh_prims == 0: linear scan over all prims)
    const WideNode* nodes;      // n_bvh_prims - 1 nodes, node 0 is the root
    const int32_t* bvh_prim;    // sorted slot -> scan-order primitive index
    const LeafRec<R>* leaves;   // n_bvh_prims, sorted-slot order
    const int32_t* linear_prim; // primitives kept out of the tree (unbounded planes, bent quads), scan order
    int32_t n_bvh_prims;
    int32_t n_linear_prims;
    // where a walk reports a link it had to drop because its stack was full (never happens with a tree the host admitted;
    // counted instead of silently losing a subtree)
    unsigned long long* stack_overflow;
};

// The area lights of a scene: a kernel argument of its own, read by the next-event-estimation kernels only (as fields
// of SceneView they cost the other kernels scalar registers: +83 v_readlane in the fused kernel, 3-5 % of its speed).
template <typename R>
struct LightView {
    const LightRec<R>* lights;
    const float* cdf;     // n, non-decreasing, last = 1
    int32_t n;
    int32_t pad;
};

struct RenderParams {
    uint32_t width, height;
    uint32_t sampler_type, spp, x_strata, y_strata;
    uint32_t x_shift;          // log2(x_strata) if it is a power of two, else 0xffffffff
    uint32_t perm_mask;        // next_pow2(spp) - 1
    uint32_t min_depth, max_depth;
    uint32_t sample_begin;     // first per-pixel sample index of this call
    uint32_t n_samples;        // samples per pixel traced by this call
    // Work items of the fused kernel: per pixel n_big_chunks items of `chunk` consecutive sample indices, then the rest
    // in items of small_chunk.  Items are numbered chunk-major (item = chunk index * pixels + pixel), so the small
    // ones are handed out last and the lanes run dry within a small item's time of each other.
    uint32_t chunk;
    uint32_t n_big_chunks;
    uint64_t n_items;          // (n_big_chunks + small chunks per pixel) * width * height < 2^31
    uint32_t seed_lo, seed_hi;
    float inv_width, inv_height;          // 1 / width, 1 / height
    float inv_spp, inv_x_strata, inv_y_strata;
    uint32_t flags;            // RENDER_FLAG_*
    // floor(n / d) for n < 2^31 as umulhi(n, m) >> s (m = ceil(2^(32 + s) / d), s = ceil(log2 d) - 1; s = 0xffffffff marks
    // d = 1): the kernels divide item numbers by the pixel count and pixel numbers by the image width, and a 32-bit
    // division by a run-time divisor is ~35 VALU instructions
    uint32_t div_pixels_m, div_pixels_s;
    uint32_t div_width_m, div_width_s;
    uint32_t small_chunk, pad2;
    float* accum;              // width*height*3 radiance sums (atomicAdd target)
    unsigned long long* counters;  // see COUNTER_* below
    unsigned long long* work_counter;
};

enum : uint32_t { RENDER_FLAG_NEE = 1u, RENDER_FLAG_EXACT = 2u };

enum : int {
    COUNTER_RAYS = 0,
    COUNTER_PRIM_TESTS = 1,
    COUNTER_NODE_TESTS = 2,
    COUNTER_BOUNCE0 = 3,   // +material type, 5 entries
    COUNTER_NONFINITE = 8,
    COUNTER_SAMPLES = 9,
    COUNTER_STACK_OVERFLOW = 10,  // traversal-stack entries dropped (a tree deeper than the build admits)
    COUNTER_EXACT_REDOS = 11,     // closest-hit queries repeated in f64
    COUNTER_COUNT = 16
};
// The fused kernel's waves add their event counts to one of COUNTER_ROWS copies of the counter block (summed on the
// host): 8192 waves adding to ONE set of ten words at the end of a launch took 0.5 ms of same-address atomics.
constexpr int COUNTER_ROWS = 64;


}  // namespace rrt
