// Device functions of the path tracer: arithmetic helpers, counter-based RNG,
// the stateless stratified sampler, ray/primitive tests and closest-hit.
//
// Everything is templated on the arithmetic type R: float is the production
// path (FP32 VALU), double exists so that hit decisions can be compared with
// an f64 CPU implementation without f32 rounding in the way.
#pragma once
#include <hip/hip_runtime.h>

#include "rrt_types.hpp"

namespace rrt {

#define RRT_DEV __device__ __forceinline__

// Scene tables never change while a kernel runs.  Reading them through the
// constant address space lets the compiler use scalar loads (s_load_*) whenever
// the index is wave-uniform; through a plain global pointer it must assume the
// kernel's own atomics could alias the table and falls back to vector loads.
#define RRT_CONSTANT __attribute__((address_space(4)))
template <typename T>
RRT_DEV const T RRT_CONSTANT* as_constant(const T* p) {
    return (const T RRT_CONSTANT*)p;
}

// ---------------------------------------------------------------- arithmetic
template <typename R>
struct Num;

template <>
struct Num<float> {
    static RRT_DEV float tmin() { return 0.001f; }            // src/scene.rs:40
    static RRT_DEV float tmax() { return 3.402823466e+38f; }  // largest finite value of R
    static RRT_DEV float eps() { return 1e-6f; }
    static RRT_DEV float rcp(float x) { return __builtin_amdgcn_rcpf(x); }
    static RRT_DEV float rsqrt(float x) { return __builtin_amdgcn_rsqf(x); }
    static RRT_DEV float sqrt(float x) { return __builtin_amdgcn_sqrtf(x); }
    static RRT_DEV float abs(float x) { return __builtin_fabsf(x); }
    // sin / cos of 2*pi*x, x in [0,1): v_sin_f32 / v_cos_f32 take revolutions
    static RRT_DEV void sincos2pi(float x, float& s, float& c) {
        s = __builtin_amdgcn_sinf(x);
        c = __builtin_amdgcn_cosf(x);
    }
    // x^e for x in [0,1], e > 0
    static RRT_DEV float pow(float x, float e) {
        return __builtin_amdgcn_exp2f(e * __builtin_amdgcn_logf(x));
    }
    static RRT_DEV float uniform(uint32_t bits) { return (float)(bits >> 8) * 5.9604644775390625e-8f; }
    static RRT_DEV float inv_pi() { return 0.318309886183790671538f; }
};

template <>
struct Num<double> {
    static RRT_DEV double tmin() { return 0.001; }
    static RRT_DEV double tmax() { return 1.7976931348623157e308; }  // f64::MAX, src/scene.rs:37
    static RRT_DEV double eps() { return 1e-6; }
    static RRT_DEV double rcp(double x) { return 1.0 / x; }
    static RRT_DEV double rsqrt(double x) { return 1.0 / ::sqrt(x); }
    static RRT_DEV double sqrt(double x) { return ::sqrt(x); }
    static RRT_DEV double abs(double x) { return ::fabs(x); }
    static RRT_DEV void sincos2pi(double x, double& s, double& c) { ::sincospi(2.0 * x, &s, &c); }
    static RRT_DEV double pow(double x, double e) { return ::pow(x, e); }
    static RRT_DEV double uniform(uint32_t bits) { return (double)bits * 2.3283064365386962890625e-10; }
    static RRT_DEV double inv_pi() { return 0.318309886183790671538; }
};

template <typename R>
struct V3 {
    R x, y, z;
};
template <typename R>
RRT_DEV V3<R> mk(R x, R y, R z) {
    V3<R> r{x, y, z};
    return r;
}
template <typename R>
RRT_DEV R dot(V3<R> a, V3<R> b) {
    return a.x * b.x + a.y * b.y + a.z * b.z;
}
template <typename R>
RRT_DEV V3<R> cross(V3<R> a, V3<R> b) {
    return mk<R>(a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x);
}
template <typename R>
RRT_DEV V3<R> operator+(V3<R> a, V3<R> b) {
    return mk<R>(a.x + b.x, a.y + b.y, a.z + b.z);
}
template <typename R>
RRT_DEV V3<R> operator-(V3<R> a, V3<R> b) {
    return mk<R>(a.x - b.x, a.y - b.y, a.z - b.z);
}
template <typename R>
RRT_DEV V3<R> operator*(V3<R> a, R s) {
    return mk<R>(a.x * s, a.y * s, a.z * s);
}
template <typename R>
RRT_DEV V3<R> normalize(V3<R> a) {
    return a * Num<R>::rsqrt(dot(a, a));
}

// floor(n / d) for n < 2^31 with the divisor's precomputed (m, s), see RenderParams
RRT_DEV uint32_t fast_div(uint32_t n, uint32_t m, uint32_t s) {
    return s == 0xffffffffu ? n : (__umulhi(n, m) >> s);
}

// ---------------------------------------------------------------- RNG
// Counter-based: every number is a hash of (seed, pixel, sample, draw index);
// there is no generator state to carry in a queue or to advance in order.
RRT_DEV uint32_t mix32(uint32_t x) {
#ifdef RRT_EXP_CHEAPHASH
    return x ^ (x >> 7) ^ (x << 11);
#endif
    x ^= x >> 16;
    x *= 0x7feb352dU;
    x ^= x >> 15;
    x *= 0x846ca68bU;
    x ^= x >> 16;
    return x;
}

RRT_DEV uint32_t pixel_key(uint32_t seed_lo, uint32_t seed_hi, uint32_t pixel) {
    return mix32(seed_lo ^ mix32(pixel * 0x9E3779B1U + (mix32(seed_hi) | 1U)));
}
RRT_DEV uint32_t sample_key(uint32_t pixkey, uint32_t sample) {
    return mix32(pixkey + sample * 0x9E3779B9U);
}

// Kensler's hash-based permutation of {0..l-1} ("Correlated Multi-Jittered
// Sampling", Pixar TM 13-01): a bijection for every key p, evaluated for one
// index at a time, which replaces the reference's per-pixel shuffled tables
// (src/sampler.rs:105-125).  w = next_pow2(l) - 1.
// (i & w) >> s for w = 2^k - 1 is a bit-field extract: one instruction (v_bfe_u32) instead of two.
RRT_DEV uint32_t masked_shr(uint32_t i, uint32_t k, uint32_t s) {
    const int width = (int)k - (int)s;  // (signed max: scalar arithmetic has no saturating subtract)
    return __builtin_amdgcn_ubfe(i, s, (uint32_t)(width > 0 ? width : 0));
}

RRT_DEV uint32_t permute(uint32_t i, uint32_t l, uint32_t w, uint32_t p) {
#ifdef RRT_EXP_CHEAPHASH
    return (i ^ p) & w;
#endif
    // k = bits of w.  The field widths are wave-uniform scalar arithmetic; the empty asm keeps it next to its use
    // (hoisted out of a kernel's main loop the five widths would occupy scalar registers for the whole kernel).
    uint32_t k = (uint32_t)__builtin_popcount(w);
    asm volatile("" : "+s"(k));
    do {
        i ^= p;
        i *= 0xe170893dU;
        i ^= p >> 16;
        i ^= masked_shr(i, k, 4);
        i ^= p >> 8;
        i *= 0x0929eb3fU;
        i ^= p >> 23;
#ifndef RRT_EXP_PERM3  // experiment: three multiply rounds instead of six (measured on smallpt: see profiles)
        i ^= masked_shr(i, k, 1);
        i *= 1U | p >> 27;
        i *= 0x6935fa69U;
        i ^= masked_shr(i, k, 11);
        i *= 0x74dcb303U;
        i ^= masked_shr(i, k, 2);
        i *= 0x9e501cc3U;
        i ^= masked_shr(i, k, 2);
#endif
        i *= 0xc860a3dfU;
        i &= w;
        i ^= i >> 5;
    } while (i >= l);
    uint32_t r = i + p;
    return (l & (l - 1)) == 0 ? (r & w) : (r % l);
}

// Sampler of one camera sample.  `dim` is the reference's current_dimension:
// ONE counter shared by get_1d and get_2d (src/sampler.rs:135-151); the first
// four dimensions are stratified over the pixel's samples, later draws are
// plain uniforms.  The Random sampler is the same object with dim pinned at 4.
struct Sampler {
    uint32_t pixkey;  // hash(seed, pixel): keys the per-dimension permutations
    uint32_t key;     // hash(pixkey, sample): keys the uniforms of this sample
    uint32_t sample;  // sample index inside the pixel
    uint32_t ctr;     // uniforms drawn so far
    uint32_t dim;
};

RRT_DEV uint32_t next_bits(Sampler& s) { return mix32(s.key + (s.ctr++) * 0x9E3779B9U); }

// (cell + jitter) / cells, kept inside [cell/cells, (cell+1)/cells): cell + jitter can
// round up to cell + 1 in f32 when the jitter is within half an ulp of 1, which would
// move the sample into the next stratum (or to 1.0, outside [0,1)).  inv = 1 / cells.
RRT_DEV float stratum_value(uint32_t cell, float jitter, uint32_t cells, float inv) {
    float v = ((float)cell + jitter) * inv;
    float hi = (float)(cell + 1u) * inv;
    return v < hi ? v : __uint_as_float(__float_as_uint(hi) - 1u);
}
RRT_DEV double stratum_value(uint32_t cell, double jitter, uint32_t cells, float) {
    return ((double)cell + jitter) / (double)cells;  // jitter has 32 bits: no round-up in f64
}

RRT_DEV void sampler_start(Sampler& s, const RenderParams& rp, uint32_t pixkey, uint32_t sample) {
    s.pixkey = pixkey;
    s.sample = sample;
    s.key = sample_key(pixkey, sample);
    s.ctr = 0;
    s.dim = rp.sampler_type == 1 ? 0u : 4u;
}

// get_1d.  `needed` = false when the caller will not look at the value (the
// Russian-roulette draw against probability 1, src/tracers/utils.rs:56-63):
// the dimension and the draw are still consumed, as in the reference, so the
// numbers a sample sees depend on the sequence of calls only -- a CPU sampler
// that restates this generator then reproduces the device's paths.
template <typename R>
RRT_DEV R get_1d(Sampler& s, const RenderParams& rp, bool needed) {
    if (!needed) {
        s.dim += s.dim < 4u ? 1u : 0u;
        s.ctr += 1u;
        return R(0);
    }
    if (s.dim < 4u) {
        uint32_t d = s.dim++;
        uint32_t k = mix32(s.pixkey ^ ((2u * d + 1u) * 0x9E3779B9U));
        uint32_t stratum = permute(s.sample, rp.spp, rp.perm_mask, k);
        R j = Num<R>::uniform(next_bits(s));
        return stratum_value(stratum, j, rp.spp, rp.inv_spp);  // (i + xi) / spp, src/sampler.rs:108
    }
    return Num<R>::uniform(next_bits(s));
}

// Sampler::get_2d (two_d) or get_1d (!two_d) in one piece of code (src/sampler.rs:135-151): a path tracer's lanes need one draw per bounce, of either
// kind (pixel jitter and lobe samples are 2D, the dielectric's reflect-or-refract choice is 1D), and written as two
// calls the stratum permutation -- most of the cost -- runs once per kind whenever a wave holds lanes of both.
// A 2D draw takes two uniforms and ONE dimension.  a = the 1D value or the first component, b = the second (2D only).
template <typename R>
RRT_DEV void get_draw(Sampler& s, const RenderParams& rp, bool two_d, R& a, R& b) {
    const uint32_t base = s.key + s.ctr * 0x9E3779B9U;
    R j0 = Num<R>::uniform(mix32(base));
    R j1 = Num<R>::uniform(mix32(base + 0x9E3779B9U));  // the second uniform of a 2D draw (not consumed by a 1D one)
    s.ctr += two_d ? 2u : 1u;
    if (s.dim < 4u) {
        uint32_t d = s.dim++;
        uint32_t k = mix32(s.pixkey ^ ((2u * d + (two_d ? 2u : 1u)) * 0x9E3779B9U));
        uint32_t cell = permute(s.sample, rp.spp, rp.perm_mask, k);
        uint32_t xc, yc;  // x_cell = i % x_strata, y_cell = i / x_strata, src/sampler.rs:115-116
        if (rp.x_shift != 0xffffffffu) {
            xc = cell & (rp.x_strata - 1u);
            yc = cell >> rp.x_shift;
        } else {
            yc = cell / rp.x_strata;
            xc = cell - yc * rp.x_strata;
        }
        a = stratum_value(two_d ? xc : cell, j0, two_d ? rp.x_strata : rp.spp, two_d ? rp.inv_x_strata : rp.inv_spp);
        b = stratum_value(yc, j1, rp.y_strata, rp.inv_y_strata);
        return;
    }
    a = j0;
    b = j1;
}

// ---------------------------------------------------------------- camera
// Camera::create_ray (src/camera.rs:38-45); cam_ll already has the origin subtracted.
template <typename R>
RRT_DEV V3<R> camera_dir(const SceneView<R>& sv, R s, R t) {
    V3<R> d = mk<R>(sv.cam_ll[0] + s * sv.cam_h[0] + t * sv.cam_v[0],
                    sv.cam_ll[1] + s * sv.cam_h[1] + t * sv.cam_v[1],
                    sv.cam_ll[2] + s * sv.cam_h[2] + t * sv.cam_v[2]);
    return normalize(d);
}

// ---------------------------------------------------------------- primitive tests
// One primitive against one ray for t in the interval the reference accepts
// (closed [t_min, t_max] for sphere / plane / quad, open for triangles).
// Returns true and t when the primitive replaces the current closest hit.
//
// Triangles and quads use the plane form of the reference's tests
// (src/shapes/triangle.rs:22-61, src/shapes/quadrilateral.rs:56-117):
//   det   = e01 . (d x e03) = -(n . d)            -> same |det| < 1e-6 reject
//   ray_t = e03 . ((o-v0) x e01) / det = (n.v0 - n.o) / (n.d)
//   alpha = (o-v0) . (d x e03) / det = coefficient of (P - v0) on e01
//   beta  = d . ((o-v0) x e01) / det = coefficient of (P - v0) on e03
// with P = o + t d.  The coefficients are affine in P, so they are two dot
// products with vectors prepared on the host.  Same accept/reject conditions,
// about half the arithmetic.  The second corner test of a general quad is
// evaluated at the ray's intersection with the second triangle's own plane,
// as in the reference; for a parallelogram it folds into the first.
//
// The chain of early returns becomes ONE comparison: every condition is
// written as "term >= 0" (x >= c <=> x - c >= 0 holds exactly in floating
// point), the terms are folded with min / max (v_min3_f32 on gfx950) and the
// primitive is accepted iff the fold is >= 0.  "A or B" is max(A, B).  min and
// max drop a NaN operand, and the only source of NaN is n.d = 0, whose own
// term |n.d| - 1e-6 is negative, so a NaN can never turn a reject into an
// accept.  No branch, no exec-mask traffic: the scan is a straight run of FMAs.
template <typename R>
RRT_DEV R min3(R a, R b, R c) {
    return fmin(fmin(a, b), c);
}

// best - t with t already rounded.  t is a product (numerator x reciprocal); left to itself the
// compiler contracts "best - n * r" into one FMA, which subtracts the UNROUNDED product and can
// call two bit-identical distances different.  The tie rule needs the comparison the reference
// makes: between two stored f64/f32 values.
template <typename R>
RRT_DEV R headroom(R best_t, R t) {
#pragma clang fp contract(off)
    return best_t - t;
}

// The arithmetic of the plane-form tests, written ONCE with every rounding spelled out (explicit fused multiply-adds,
// contraction off for the rest) and used by the wave-uniform scan (prim_test_as), the packed pair scan (scan_run_pairs,
// T = two records side by side) and the tree's leaves (leaf_test_words): the scan and the tree must give a primitive the
// same t and the same verdict bit for bit (the tie rule compares distances for equality), and that must not depend on
// which products the compiler chooses to fuse at which site.
typedef float float2_t __attribute__((ext_vector_type(2)));
typedef double double2_t __attribute__((ext_vector_type(2)));
RRT_DEV float fma_r(float a, float b, float c) { return __builtin_fmaf(a, b, c); }
RRT_DEV double fma_r(double a, double b, double c) { return __builtin_fma(a, b, c); }
RRT_DEV float2_t fma_r(float2_t a, float2_t b, float2_t c) { return __builtin_elementwise_fma(a, b, c); }
RRT_DEV double2_t fma_r(double2_t a, double2_t b, double2_t c) { return __builtin_elementwise_fma(a, b, c); }

RRT_DEV void sqrt_each(float x, float& r) { r = Num<float>::sqrt(x); }
RRT_DEV void sqrt_each(double x, double& r) { r = Num<double>::sqrt(x); }
RRT_DEV void sqrt_each(float2_t x, float2_t& r) { r.x = Num<float>::sqrt(x.x); r.y = Num<float>::sqrt(x.y); }
RRT_DEV void sqrt_each(double2_t x, double2_t& r) { r.x = Num<double>::sqrt(x.x); r.y = Num<double>::sqrt(x.y); }
RRT_DEV void rcp_each(float x, float& r) { r = Num<float>::rcp(x); }
RRT_DEV void rcp_each(double x, double& r) { r = Num<double>::rcp(x); }
RRT_DEV void rcp_each(float2_t x, float2_t& r) { r.x = Num<float>::rcp(x.x); r.y = Num<float>::rcp(x.y); }
RRT_DEV void rcp_each(double2_t x, double2_t& r) { r.x = Num<double>::rcp(x.x); r.y = Num<double>::rcp(x.y); }
// out = a < limit ? yes : no, per component
RRT_DEV void select_lt(float a, float limit, float yes, float no, float& out) { out = a < limit ? yes : no; }
RRT_DEV void select_lt(double a, double limit, double yes, double no, double& out) { out = a < limit ? yes : no; }
RRT_DEV void select_lt(float2_t a, float limit, float2_t yes, float2_t no, float2_t& out) {
    out.x = a.x < limit ? yes.x : no.x;
    out.y = a.y < limit ? yes.y : no.y;
}
RRT_DEV void select_lt(double2_t a, double limit, double2_t yes, double2_t no, double2_t& out) {
    out.x = a.x < limit ? yes.x : no.x;
    out.y = a.y < limit ? yes.y : no.y;
}

// c0 x + c1 y + c2 z = fma(c2, z, fma(c1, y, c0 x)); T = R or two records side by side, the vector is per ray
template <typename T, typename R>
RRT_DEV T dot3_r(T c0, T c1, T c2, R x, R y, R z) {
#pragma clang fp contract(off)
    return fma_r(c2, (T)(z), fma_r(c1, (T)(y), c0 * (T)(x)));
}

// n . d and n . v0 - n . o of the plane (n, k)
template <typename T, typename R>
RRT_DEV void plane_nd_no(T n0, T n1, T n2, T k, V3<R> o, V3<R> d, T& nd, T& no) {
#pragma clang fp contract(off)
    nd = dot3_r<T, R>(n0, n1, n2, d.x, d.y, d.z);
    no = k - dot3_r<T, R>(n0, n1, n2, o.x, o.y, o.z);
}

// the hit point P = o + t d and one coefficient row:  (r0, r1, r2) . P + r3
template <typename T, typename R>
RRT_DEV void plane_point(T t, V3<R> o, V3<R> d, T& px, T& py, T& pz) {
    px = fma_r(t, (T)(d.x), (T)(o.x));
    py = fma_r(t, (T)(d.y), (T)(o.y));
    pz = fma_r(t, (T)(d.z), (T)(o.z));
}
template <typename T>
RRT_DEV T plane_coeff(T r0, T r1, T r2, T r3, T px, T py, T pz) {
#pragma clang fp contract(off)
    return fma_r(r2, pz, fma_r(r1, py, r0 * px)) + r3;
}

// The sphere test up to the choice of the root (src/shapes/sphere.rs:58-96): x = r^2 - d2 (negative: the ray misses),
// d2 = squared distance of the centre from the ray, t0 = the root the reference reports.
template <typename T, typename R>
RRT_DEV void sphere_roots(T cx, T cy, T cz, T r2, V3<R> o, V3<R> d, T& t0, T& x, T& d2) {
#pragma clang fp contract(off)
    const T lx = cx - (T)(o.x), ly = cy - (T)(o.y), lz = cz - (T)(o.z);
    const T t_ca = dot3_r<T, R>(lx, ly, lz, d.x, d.y, d.z);
    if (sizeof(R) == 4) {
        // |l - t_ca d|^2 equals |l|^2 - t_ca^2 for unit d; formed this way the
        // difference of two large squares is never taken, which in f32 would
        // lose the silhouette of a small sphere seen from far away
        const T qx = fma_r(-t_ca, (T)(d.x), lx), qy = fma_r(-t_ca, (T)(d.y), ly), qz = fma_r(-t_ca, (T)(d.z), lz);
        d2 = fma_r(qz, qz, fma_r(qy, qy, qx * qx));
    } else {
        d2 = fma_r(-t_ca, t_ca, fma_r(lz, lz, fma_r(ly, ly, lx * lx)));
    }
    x = r2 - d2;                    // r^2 - d2 >= 0 <=> d2 <= r^2
    T t_hc;                         // NaN when x < 0; the x term then rejects
    sqrt_each(x, t_hc);
    const T ta = t_ca - t_hc, tb = t_ca + t_hc;
    select_lt(ta, Num<R>::tmin(), tb, ta, t0);  // t0 < t_min ? t1 : t0
}

// ---------------------------------------------------------------- which f32 decisions can be trusted
// The f32 tests above decide by the sign of a fold of terms (edge coefficients, |n.d| - 1e-6, t - t_min, best_t - t).
// Evaluated in f64 from the f64 scene and the f64 ray they can come out the other way when the deciding term is
// smaller than the rounding error it carries.  A tracked query (TRACK = true; f32 only) bounds that error for every
// primitive it tests, from per-primitive magnitudes prepared by the host (ErrRec) and the ray's own numbers, and
// raises `amb` when a fold lies inside its bound; the caller then repeats the query in f64 (closest_hit_exact).
// The bounds are first-order running-error bounds.  With u = 2^-24: n.d carries at most 4 u sum |n_i d_i| (the three
// entries of n rounded from f64, one product and two fused multiply-adds), k - n.o at most 5 u (|k| + sum |n_i o_i|), the
// reciprocal 2 u (v_rcp_f32: one ulp) and the product one more; a ray rounded from f64 adds one u to each:
//   ku    = 1.25 (5 + e_in) u, e_in = error of the ray's components in units of u (0 for a ray that IS f32, 1 for one
//           rounded from f64); the 1.25 covers the second-order terms and the error of |1 / n.d| itself (see below)
//   planar primitive (n, k; rows A, B):   M = |k| + sum |n_i| |o_i|,  N1 = sum |n_i|,  A1 = max(sum |A_i|, sum |B_i|)
//     E_t  = ku (|1 / n.d| (M + N1 |t|) + |t|)                    error of t
//     E_ab = A1 (E_t + 2 ku (|o|_inf + |t|)) + ku (max(|cA|, |cB|) + 1)   error of a row's coefficient at the hit point
//     fold trusted iff |fold| > 2 E_ab + E_t + ku N1 + (error of the closest distance so far)
//   and never when |n.d| <= 16 ku N1 (its reciprocal is not known to 6 %) unless |n.d| - 1e-6 < -ku N1 (parallel for sure).
// The sphere's terms are bounded the same way, step by step, in sphere_track.
// (ErrRec: rrt_types.hpp -- planar: N1, |k|, A1, max(|cA|, |cB|) + 1 [, the same four of a general quad's second corner];
//  sphere: |c|_inf)
template <bool ON>
struct HitTrack {};
template <>
struct HitTrack<true> {
    bool amb;        // some decision of this query lies inside its error bound
    float best_err;  // error bound of the closest distance found so far
    float last_err;  // error bound of the distance the last primitive test reported
    float ku;
    float u1;        // 1.25 u: one rounding, with the margin for second-order terms
    float e_in;      // error of the ray's components in units of u (0: the ray IS f32; 1: rounded from f64)
    float om;        // |o|_inf of the ray
    const ErrRec* err;  // the scene's magnitudes, scan order
};
RRT_DEV void track_start(HitTrack<true>& tr, const ErrRec* err, V3<float> o, float e_in) {
    tr.err = err;
    tr.amb = false;
    tr.best_err = tr.last_err = 0.f;
    tr.ku = 1.25f * (5.f + e_in) * 5.9604644775390625e-8f;
    tr.u1 = 1.25f * 5.9604644775390625e-8f;
    tr.e_in = e_in;
    tr.om = fmaxf(fmaxf(__builtin_fabsf(o.x), __builtin_fabsf(o.y)), __builtin_fabsf(o.z));
}
RRT_DEV void track_accept(HitTrack<true>& tr) { tr.best_err = tr.last_err; }
RRT_DEV void track_accept(HitTrack<false>&) {}

// E_t and E_ab of one plane (n = f[0..2], k = f[3]) with magnitudes e[0..3]
RRT_DEV void plane_track(const HitTrack<true>& tr, float n0, float n1, float n2, const float RRT_CONSTANT* e, V3<float> o,
                         float rc, float t, float& et, float& eab) {
    const float at = __builtin_fabsf(t), arc = __builtin_fabsf(rc);
    const float m = __builtin_fmaf(__builtin_fabsf(n2), __builtin_fabsf(o.z),
                                   __builtin_fmaf(__builtin_fabsf(n1), __builtin_fabsf(o.y), __builtin_fmaf(__builtin_fabsf(n0), __builtin_fabsf(o.x), e[1])));
    et = tr.ku * __builtin_fmaf(arc, __builtin_fmaf(e[0], at, m), at);
    eab = __builtin_fmaf(e[2], __builtin_fmaf(2.f * tr.ku, tr.om + at, et), tr.ku * e[3]);
}

// Judges a planar test: fold = the number whose sign decided, nd = n.d, det_ok = |n.d| - 1e-6, bound = its error.
RRT_DEV void track_judge(HitTrack<true>& tr, float fold, float nd, float det_ok, float n1sum, float bound) {
    const float e_nd = tr.ku * n1sum;
    if (det_ok < -e_nd) return;  // parallel beyond doubt: rejected in any arithmetic
    const bool trusted = (__builtin_fabsf(nd) > 16.f * e_nd) && (__builtin_fabsf(fold) > bound + tr.best_err);
    if (!trusted) tr.amb = true;
}

// The sphere test's decisions (sphere_roots + the range tests): x = r^2 - d2 >= 0, ta < t_min (which root),
// t0 - t_min >= 0, best_t - t0 >= 0.
RRT_DEV void sphere_track(HitTrack<true>& tr, const float RRT_CONSTANT* f, const float RRT_CONSTANT* e, V3<float> o, V3<float> d,
                          float x, float d2, float t0, float best_t) {
#pragma clang fp contract(off)
    // Rounding by rounding (u1 = 1.25 u per rounding): the centre is the f64 centre rounded (u |c|), the origin carries
    // e_in u |o|, the subtraction one more rounding of l; t_ca = l . d is three products of a component of l with one of
    // d (|d_i| <= 1, d itself e_in u off) and two sums; q = l - t_ca d; d2 = |q|^2; x = r^2 - d2 with r^2 rounded.
    const float u1 = tr.u1, e_in = tr.e_in;
    const float lx = f[0] - o.x, ly = f[1] - o.y, lz = f[2] - o.z;
    const float l1 = __builtin_fabsf(lx) + __builtin_fabsf(ly) + __builtin_fabsf(lz);
    const float t_ca = lz * d.z + (ly * d.y + lx * d.x);
    const float el = u1 * (2.f * e[0] + (1.f + e_in) * tr.om);            // error of a component of l = c - o
    const float e_tca = 3.f * el + u1 * (3.f + e_in) * l1;
    const float e_q = el + e_tca + u1 * (l1 + (2.f + e_in) * __builtin_fabsf(t_ca));  // of a component of q = l - t_ca d
    const float q1 = 1.7320508f * __builtin_amdgcn_sqrtf(d2);           // >= |q|_1
    // (f64 forms d2 as |l|^2 - t_ca^2, which is |l - t_ca d|^2 only for a unit d; an f32 direction is unit to a few u)
    const float e_x = 2.f * q1 * e_q + u1 * (4.f * d2 + 2.f * f[4] + 8.f * t_ca * t_ca);
    if (x < -e_x) { tr.last_err = 0.f; return; }           // misses beyond doubt
    if (!(x > e_x)) { tr.amb = true; return; }
    const float t_hc = __builtin_amdgcn_sqrtf(x);
    const float e_t0 = e_tca + e_x * __builtin_amdgcn_rcpf(t_hc) + 2.f * u1 * (__builtin_fabsf(t_ca) + t_hc);
    tr.last_err = e_t0;
    const float ta = t_ca - t_hc;
    const float fold = fminf(t0 - Num<float>::tmin(), best_t - t0);
    if (!(__builtin_fabsf(ta - Num<float>::tmin()) > e_t0) || !(__builtin_fabsf(fold) > e_t0 + tr.best_err)) tr.amb = true;
}

// TYPE is a compile-time constant in the run loops of the linear scan (no per-primitive
// dispatch) and a per-lane value in tree leaves.
template <typename R, int TYPE, bool TRACK = false>
RRT_DEV bool prim_test_as(const PrimRec<R> RRT_CONSTANT* p, V3<R> o, V3<R> d, R best_t, R& t_out,
                          HitTrack<TRACK>* tr = nullptr, const ErrRec RRT_CONSTANT* er = nullptr) {
    static_assert(!TRACK || sizeof(R) == 4, "only the f32 tests are tracked");
    const R RRT_CONSTANT* f = p->f;
    if (TYPE >= PRIM_TRIANGLE) {
#pragma clang fp contract(off)
        R nd, no;
        plane_nd_no<R, R>(f[0], f[1], f[2], f[3], o, d, nd, no);
        R t = no * Num<R>::rcp(nd);
        t_out = t;
        R px, py, pz;
        plane_point<R, R>(t, o, d, px, py, pz);
        R a = plane_coeff<R>(f[4], f[5], f[6], f[7], px, py, pz);
        R b = plane_coeff<R>(f[8], f[9], f[10], f[11], px, py, pz);
        R det_ok = Num<R>::abs(nd) - Num<R>::eps();
        float et = 0.f, eab = 0.f;
        if constexpr (TRACK) {
            plane_track(*tr, f[0], f[1], f[2], er->e, o, Num<R>::rcp(nd), t, et, eab);
            tr->last_err = et;
        }
        if (TYPE == PRIM_PGRAM) {
            // parallelogram: alpha' = 1 - alpha and beta' = 1 - beta, so the two-triangle
            // test is 0 <= alpha, beta <= 1; a and b arrive shifted by -1/2
            R h = min3(R(0.5) - Num<R>::abs(a), R(0.5) - Num<R>::abs(b), det_ok);
            const R fold = min3(h, t - Num<R>::tmin(), headroom(best_t, t));
            if constexpr (TRACK) track_judge(*tr, fold, nd, det_ok, er->e[0], 2.f * eab + et + tr->ku * er->e[0]);
            return fold >= R(0);
        }
        if (TYPE == PRIM_QUAD) {
            // The corner at v2 (src/shapes/quadrilateral.rs:90-111): the reference intersects the ray with the
            // plane of the triangle (v2, v3, v1) ITSELF and takes the coefficients of that point on e23 and
            // e21.  For a planar quad it is the point P; the loader admits quads bent by up to 2.5 degrees
            // (quadrilateral.rs:23-33) and a Quadrilateral object is not checked at all (quadrilateral.rs:204-216),
            // and there the two points differ.  The distance stays the first plane's (quadrilateral.rs:114).
            R nd2, no2;
            plane_nd_no<R, R>(f[20], f[21], f[22], f[23], o, d, nd2, no2);
            R t2 = no2 * Num<R>::rcp(nd2);
            R qx, qy, qz;
            plane_point<R, R>(t2, o, d, qx, qy, qz);
            R a2 = plane_coeff<R>(f[12], f[13], f[14], f[15], qx, qy, qz);
            R b2 = plane_coeff<R>(f[16], f[17], f[18], f[19], qx, qy, qz);
            R second = min3(a2, b2, Num<R>::abs(nd2) - Num<R>::eps());  // the (v2, v3, v1) corner
            R inside = fmax(R(1) - (a + b), second);                      // alpha + beta <= 1, or that corner
            R h = min3(min3(a, b, inside), det_ok, t - Num<R>::tmin());
            const R fold = fmin(h, headroom(best_t, t));
            if constexpr (TRACK) {
                // the second corner's terms carry the error of their own plane; that plane being (nearly) parallel
                // to the ray makes them untrustworthy, not the test (its first corner may still decide)
                float et2, eab2;
                plane_track(*tr, f[20], f[21], f[22], er->e + 4, o, Num<R>::rcp(nd2), t2, et2, eab2);
                float bound = 2.f * eab + et + tr->ku * er->e[0] + 2.f * eab2 + tr->ku * er->e[4];
                if (!(__builtin_fabsf(nd2) > 16.f * tr->ku * er->e[4])) bound = 3.0e38f;
                track_judge(*tr, fold, nd, det_ok, er->e[0], bound);
            }
            return fold >= R(0);
        }
        // triangle: u, v >= 0, u <= 1, u + v <= 1, and an OPEN t interval
        R h = min3(min3(a, b, R(1) - a), R(1) - (a + b), det_ok);
        const R h2 = fmin(t - Num<R>::tmin(), headroom(best_t, t));
        if constexpr (TRACK) track_judge(*tr, fmin(h, h2), nd, det_ok, er->e[0], 2.f * eab + et + tr->ku * er->e[0]);
        return (h >= R(0)) && (h2 > R(0));
    }
    if (TYPE == PRIM_SPHERE) {  // src/shapes/sphere.rs:58-96
        R t0, x, d2;
        sphere_roots<R, R>(f[0], f[1], f[2], f[4], o, d, t0, x, d2);
        t_out = t0;
        if constexpr (TRACK) sphere_track(*tr, f, er->e, o, d, x, d2, t0, best_t);
        return min3(fmin(x, d2), t0 - Num<R>::tmin(), headroom(best_t, t0)) >= R(0);
    }
    // plane, src/shapes/plane.rs:25-44
    {
#pragma clang fp contract(off)
        R nd, no;
        plane_nd_no<R, R>(f[0], f[1], f[2], f[3], o, d, nd, no);
        R t = no * Num<R>::rcp(nd);
        t_out = t;
        const R det_ok = Num<R>::abs(nd) - Num<R>::eps();
        const R fold = min3(det_ok, t - Num<R>::tmin(), headroom(best_t, t));
        if constexpr (TRACK) {
            float et, eab;
            plane_track(*tr, f[0], f[1], f[2], er->e, o, Num<R>::rcp(nd), t, et, eab);
            tr->last_err = et;
            track_judge(*tr, fold, nd, det_ok, er->e[0], et + tr->ku * er->e[0]);
        }
        return fold >= R(0);
    }
}

template <typename R, bool TRACK = false>
RRT_DEV bool prim_test(const PrimRec<R> RRT_CONSTANT* p, V3<R> o, V3<R> d, R best_t, R& t_out,
                       HitTrack<TRACK>* tr = nullptr, const ErrRec RRT_CONSTANT* er = nullptr) {
    switch (p->type) {
    case PRIM_SPHERE: return prim_test_as<R, PRIM_SPHERE, TRACK>(p, o, d, best_t, t_out, tr, er);
    case PRIM_PLANE: return prim_test_as<R, PRIM_PLANE, TRACK>(p, o, d, best_t, t_out, tr, er);
    case PRIM_TRIANGLE: return prim_test_as<R, PRIM_TRIANGLE, TRACK>(p, o, d, best_t, t_out, tr, er);
    case PRIM_QUAD: return prim_test_as<R, PRIM_QUAD, TRACK>(p, o, d, best_t, t_out, tr, er);
    default: return prim_test_as<R, PRIM_PGRAM, TRACK>(p, o, d, best_t, t_out, tr, er);
    }
}

struct HitCounters {
    uint32_t prim_tests;
    uint32_t node_tests;
};

template <typename R, int TYPE, bool TRACK = false>
RRT_DEV void scan_run(const PrimRec<R> RRT_CONSTANT* prims, int first, int count, V3<R> o, V3<R> d,
                      R& best_t, int& best_g, HitTrack<TRACK>* tr = nullptr, const ErrRec RRT_CONSTANT* err = nullptr) {
#pragma unroll 2
    for (int k = 0; k < count; ++k) {
        R t;
        const bool ok = prim_test_as<R, TYPE, TRACK>(prims + first + k, o, d, best_t, t, tr, TRACK ? err + first + k : nullptr);
        best_t = ok ? t : best_t;
        best_g = ok ? first + k : best_g;
        if constexpr (TRACK) { if (ok) track_accept(*tr); }
    }
}

// Two parallelograms (or two triangles) per step with packed FP32 arithmetic: component 0 of every R2 belongs
// to the first record of the pair, component 1 to the second.  The dot products, t, the hit
// point and both coefficients are packed (one issue slot for two primitives); reciprocal,
// |x|, min3 and the selects have no packed form.  The second record is judged against the
// closest hit as updated by the first, exactly as in a sequential scan.
template <typename R, int TYPE>
RRT_DEV void scan_run_pairs(const PairRec<R> RRT_CONSTANT* pairs, int first, int count, V3<R> o, V3<R> d,
                            R& best_t, int& best_g) {
#pragma clang fp contract(off)
    typedef R R2 __attribute__((ext_vector_type(2)));
    const int n_pairs = (count + 1) >> 1;
    for (int k = 0; k < n_pairs; ++k) {
        const R2 RRT_CONSTANT* f = reinterpret_cast<const R2 RRT_CONSTANT*>(pairs[k].g);
        // (every packed operation reads at most one scalar register pair: the constant terms f[3], f[7], f[11]
        //  enter through an add of their own, not as the addend of an FMA whose multiplicand is a scalar pair
        //  too, which would cost a move into vector registers each)
        R2 nd, no;
        plane_nd_no<R2, R>(f[0], f[1], f[2], f[3], o, d, nd, no);
        R2 rc;
        rcp_each(nd, rc);
        R2 t = no * rc;
        R2 px, py, pz;
        plane_point<R2, R>(t, o, d, px, py, pz);
        R2 a = plane_coeff<R2>(f[4], f[5], f[6], f[7], px, py, pz);
        R2 b = plane_coeff<R2>(f[8], f[9], f[10], f[11], px, py, pz);
        R2 tm = t - Num<R>::tmin();
        if (TYPE == PRIM_PGRAM) {
            // |a| <= 1/2 <=> 1/4 - a^2 >= 0 exactly (the square of the f32/f64 next to 1/2 rounds to the
            // proper side of 1/4), and the square form is one packed FMA for both records
            R2 qa = fma_r(-a, a, (R2)(R(0.25)));
            R2 qb = fma_r(-b, b, (R2)(R(0.25)));
            R h0 = min3((R)qa.x, (R)qb.x, Num<R>::abs(nd.x) - Num<R>::eps());
            const bool ok0 = min3(h0, tm.x, headroom(best_t, (R)t.x)) >= R(0);
            best_t = ok0 ? t.x : best_t;
            R h1 = min3((R)qa.y, (R)qb.y, Num<R>::abs(nd.y) - Num<R>::eps());
            const bool ok1 = min3(h1, tm.y, headroom(best_t, (R)t.y)) >= R(0);
            best_t = ok1 ? t.y : best_t;
            // the index of the later of the two hits: one move, one add-with-carry, one select instead of two of each
            best_g = (ok0 || ok1) ? first + 2 * k + (ok1 ? 1 : 0) : best_g;
        } else {
            // triangle: u, v >= 0 and u + v <= 1 (u <= 1 follows: the rounded sum is never below u
            // when v >= 0), and an OPEN t interval
            R2 rest = R(1) - (a + b);
            R h0 = fmin(min3((R)a.x, (R)b.x, (R)rest.x), Num<R>::abs(nd.x) - Num<R>::eps());
            const bool ok0 = (h0 >= R(0)) && (fmin((R)tm.x, headroom(best_t, (R)t.x)) > R(0));
            best_t = ok0 ? t.x : best_t;
            R h1 = fmin(min3((R)a.y, (R)b.y, (R)rest.y), Num<R>::abs(nd.y) - Num<R>::eps());
            const bool ok1 = (h1 >= R(0)) && (fmin((R)tm.y, headroom(best_t, (R)t.y)) > R(0));
            best_t = ok1 ? t.y : best_t;
            best_g = (ok0 || ok1) ? first + 2 * k + (ok1 ? 1 : 0) : best_g;
        }
    }
}

// Two planes or two spheres per step, the same way (a run of one keeps the scalar loop).  The arithmetic of each
// record is prim_test_as's, operation by operation, so both give a primitive the same t and the same verdict.
template <typename R, int TYPE>
RRT_DEV void scan_run_pairs_round(const PairRec<R> RRT_CONSTANT* pairs, int first, int count, V3<R> o, V3<R> d,
                                  R& best_t, int& best_g) {
#pragma clang fp contract(off)
    typedef R R2 __attribute__((ext_vector_type(2)));
    const int n_pairs = (count + 1) >> 1;
    for (int k = 0; k < n_pairs; ++k) {
        const R2 RRT_CONSTANT* f = reinterpret_cast<const R2 RRT_CONSTANT*>(pairs[k].g);
        if (TYPE == PRIM_PLANE) {
            R2 nd, no;
            plane_nd_no<R2, R>(f[0], f[1], f[2], f[3], o, d, nd, no);
            R2 rc;
            rcp_each(nd, rc);
            R2 t = no * rc;
            R2 tm = t - Num<R>::tmin();
            const bool ok0 = min3(Num<R>::abs(nd.x) - Num<R>::eps(), (R)tm.x, headroom(best_t, (R)t.x)) >= R(0);
            best_t = ok0 ? t.x : best_t;
            const bool ok1 = min3(Num<R>::abs(nd.y) - Num<R>::eps(), (R)tm.y, headroom(best_t, (R)t.y)) >= R(0);
            best_t = ok1 ? t.y : best_t;
            best_g = (ok0 || ok1) ? first + 2 * k + (ok1 ? 1 : 0) : best_g;
        } else {  // PRIM_SPHERE
            R2 ts, x, d2;
            sphere_roots<R2, R>(f[0], f[1], f[2], f[4], o, d, ts, x, d2);
            R2 tm = ts - Num<R>::tmin();
            // (f32: d2 is a sum of squares, so min(x, d2) >= 0 exactly when x >= 0 and the d2 term of the
            //  scalar test is dropped; f64 forms d2 as a difference and keeps it)
            R in0 = sizeof(R) == 4 ? (R)x.x : fmin((R)x.x, (R)d2.x);
            const bool ok0 = min3(in0, (R)tm.x, headroom(best_t, (R)ts.x)) >= R(0);
            best_t = ok0 ? ts.x : best_t;
            R in1 = sizeof(R) == 4 ? (R)x.y : fmin((R)x.y, (R)d2.y);
            const bool ok1 = min3(in1, (R)tm.y, headroom(best_t, (R)ts.y)) >= R(0);
            best_t = ok1 ? ts.y : best_t;
            best_g = (ok0 || ok1) ? first + 2 * k + (ok1 ? 1 : 0) : best_g;
        }
    }
}

// Scene::intersect (src/scene.rs:35-47) as a scan over the primitive table in
// the reference's order with a shrinking t_max.  The table index is uniform
// across the wave, so records arrive through scalar loads.  The table is cut
// into runs of equal type (host side), each run is a tight loop of one test.
template <typename R, bool TRACK = false>
RRT_DEV void scan_linear(const SceneView<R>& sv, V3<R> o, V3<R> d, R& best_t, int& best_g,
                         HitCounters& hc, HitTrack<TRACK>* tr = nullptr) {
    const PrimRec<R> RRT_CONSTANT* prims = as_constant(sv.prims);
    const int32_t RRT_CONSTANT* runs = as_constant(sv.runs);
    const int n_runs = sv.n_runs;
    if constexpr (TRACK) {
        // a tracked scan tests every primitive through prim_test_as, whose arithmetic the pair scans repeat operation
        // by operation: same distances, same verdicts, plus the error bounds
        const ErrRec RRT_CONSTANT* err = as_constant(tr->err);
        for (int r = 0; r < n_runs; ++r) {
            const int type = runs[4 * r], first = runs[4 * r + 1], count = runs[4 * r + 2];
            switch (type) {
            case PRIM_SPHERE: scan_run<R, PRIM_SPHERE, true>(prims, first, count, o, d, best_t, best_g, tr, err); break;
            case PRIM_PLANE: scan_run<R, PRIM_PLANE, true>(prims, first, count, o, d, best_t, best_g, tr, err); break;
            case PRIM_TRIANGLE: scan_run<R, PRIM_TRIANGLE, true>(prims, first, count, o, d, best_t, best_g, tr, err); break;
            case PRIM_QUAD: scan_run<R, PRIM_QUAD, true>(prims, first, count, o, d, best_t, best_g, tr, err); break;
            default: scan_run<R, PRIM_PGRAM, true>(prims, first, count, o, d, best_t, best_g, tr, err); break;
            }
        }
        return;
    }
    for (int r = 0; r < n_runs; ++r) {
        const int type = runs[4 * r], first = runs[4 * r + 1], count = runs[4 * r + 2];
        switch (type) {
        case PRIM_SPHERE:
            if (count > 1) scan_run_pairs_round<R, PRIM_SPHERE>(as_constant(sv.pairs) + runs[4 * r + 3], first, count, o, d, best_t, best_g);
            else scan_run<R, PRIM_SPHERE>(prims, first, count, o, d, best_t, best_g);
            break;
        case PRIM_PLANE:
            if (count > 1) scan_run_pairs_round<R, PRIM_PLANE>(as_constant(sv.pairs) + runs[4 * r + 3], first, count, o, d, best_t, best_g);
            else scan_run<R, PRIM_PLANE>(prims, first, count, o, d, best_t, best_g);
            break;
        case PRIM_TRIANGLE: scan_run_pairs<R, PRIM_TRIANGLE>(as_constant(sv.pairs) + runs[4 * r + 3], first, count, o, d, best_t, best_g); break;
        case PRIM_QUAD: scan_run<R, PRIM_QUAD>(prims, first, count, o, d, best_t, best_g); break;
        default: scan_run_pairs<R, PRIM_PGRAM>(as_constant(sv.pairs) + runs[4 * r + 3], first, count, o, d, best_t, best_g); break;
        }
    }
}

}  // namespace rrt
