/*
 * rrt.h - C ABI of the MI355X (gfx950) path-tracing backend.
 *
 * The reference renderer (nero19960329/RustRayTracer) has no FFI of its own:
 * its only seam is the Rust function
 *     pub fn render(config: &RenderConfig, scene: &Scene) -> RgbImage     (src/renderer.rs:73)
 * called once from main (src/main.rs:48).  The entry points below are what a
 * Rust `extern "C"` block for that seam would bind: a flattened,
 * transform-baked scene goes in, radiance sums / 8-bit pixels come out.
 * Every entry point cites the reference code it replaces.
 *
 * Conventions
 *   - every function returns RRT_OK (0) or a negative rrt_status; nothing
 *     throws or aborts across the boundary; rrt_last_error() returns a
 *     thread-local message for the last failure on the calling thread
 *   - all arithmetic inputs are f64, as in the reference (src/math.rs:4-8);
 *     the device computes in f32 or f64 as rrt_render_desc.precision says
 *   - a backend handle is not thread-safe: one host thread drives one backend.  A scene handle may be used by
 *     several host threads at once, each with a backend of its own: what a scene builds on first use on a device
 *     (its tables, a tree) is built once, under the scene's own lock
 *   - pointers named *_dev_or_host may be device or host memory; the library
 *     asks the HIP runtime which one it was given
 */
#ifndef RRT_H
#define RRT_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef enum rrt_status {
    RRT_OK = 0,
    RRT_ERR_INVALID = -1,     /* bad argument / config the reference would panic on */
    RRT_ERR_HIP = -2,         /* a HIP runtime call failed */
    RRT_ERR_UNSUPPORTED = -3, /* valid request this build cannot serve */
    RRT_ERR_NO_DEVICE = -4,   /* no usable gfx950 device */
    RRT_ERR_IO = -5,          /* file could not be read / written / parsed */
    RRT_ERR_NOMEM = -6
} rrt_status;

/* src/shapes/shape.rs:16-24 (a Mesh is flattened into its faces). */
typedef enum rrt_prim_type {
    RRT_PRIM_SPHERE = 0,   /* v[0..2] centre, v[3] radius                 src/shapes/sphere.rs:12-15 */
    RRT_PRIM_PLANE = 1,    /* v[0..2] point, v[3..5] unit normal          src/shapes/plane.rs:12-15 */
    RRT_PRIM_TRIANGLE = 2, /* v[0..8] three vertices                      src/shapes/triangle.rs:12-14 */
    RRT_PRIM_QUAD = 3      /* v[0..11] four vertices                      src/shapes/quadrilateral.rs:13-15 */
} rrt_prim_type;

/* src/material.rs:286-294 */
typedef enum rrt_material_type {
    RRT_MAT_EMISSIVE = 0,   /* rgb = emitted colour                                          */
    RRT_MAT_LAMBERTIAN = 1, /* rgb = albedo                                                  */
    RRT_MAT_PHONG = 2,      /* rgb = specular, param = shininess                             */
    RRT_MAT_REFLECTOR = 3,  /* no parameters                                                 */
    RRT_MAT_DIELECTRIC = 4  /* param = index of refraction                                   */
} rrt_material_type;

/* src/sampler.rs:168-173 */
typedef enum rrt_sampler_type { RRT_SAMPLER_RANDOM = 0, RRT_SAMPLER_STRATIFIED = 1 } rrt_sampler_type;

/* F32 is the product path.  F64 computes in the reference's arithmetic type (src/math.rs:4-8) and builds the
 * reference's orthonormal basis in the lobe samplers (src/math.rs:162-172; F32 uses a cheaper one with the same
 * distribution): for a given seed it traces, sample by sample, the paths a CPU implementation of the reference
 * traces when it draws the same random numbers. */
typedef enum rrt_precision { RRT_PRECISION_F32 = 0, RRT_PRECISION_F64 = 1 } rrt_precision;

/* How closest-hit walks the primitives.  LINEAR is the reference's scan (src/scene.rs:35-47), wave-uniform.
 * The other three are builders of one tree format (4-wide nodes, 8-bit child boxes; unbounded planes and quads
 * that are not planar are scanned beside the tree), all returning the primitive and t the scan returns:
 *   SAH_GPU  binned surface-area-heuristic tree built top-down ON THE GPU, level by level (one work-group per range,
 *            16 bins along four cut coordinates, ranges of up to 32 primitives by an exact sweep);
 *   SAH      the same tree built by up to 16 host threads (~0.2 s per million primitives): the implementation the GPU
 *            builder is checked against (same cuts, same tree);
 *   LBVH     built on the GPU: primitives in Morton order, clustered bottom-up (neighbours whose union has the
 *            smallest box merge first); about a tenth more traversal steps per ray.
 * AUTO = LINEAR for scenes of up to 72 primitives (where it is the fastest), SAH_GPU otherwise.  A scene keeps every
 * tree it was asked for (per device) until rrt_scene_destroy.  rrt_stats.accel_used says which one a call walked. */
typedef enum rrt_accel { RRT_ACCEL_AUTO = 0, RRT_ACCEL_LINEAR = 1, RRT_ACCEL_LBVH = 2, RRT_ACCEL_SAH = 3, RRT_ACCEL_SAH_GPU = 4 } rrt_accel;

/* Which kernel organisation renders.  FUSED keeps a path in registers for
 * all of its bounces and refills finished lanes with new camera samples;
 * WAVEFRONT keeps SoA ray and hit queues in HBM and compacts them after every
 * bounce (max_depth <= 511; with RRT_PRECISION_F64 the queues hold doubles and the
 * tree is walked one ray per lane: a checking mode, chosen only by name).  AUTO:
 * WAVEFRONT when the closest hit goes through the tree of a scene with more than
 * 150000 primitives and the precision is f32, FUSED otherwise.  Both trace the same
 * paths for the same seed. */
typedef enum rrt_pipeline { RRT_PIPELINE_AUTO = 0, RRT_PIPELINE_FUSED = 1, RRT_PIPELINE_WAVEFRONT = 2 } rrt_pipeline;

/* Estimator switches.  0 everywhere in parity runs: the reference implements connect() for s = 0 only
 * (emission found by the camera path, src/tracers/utils.rs:115-121) and panics for s > 0 (utils.rs:122-124).
 * RRT_RENDER_NEE adds the s = 1 strategy the reference leaves open: at every Lambertian or Phong vertex one point
 * on an emissive primitive is sampled (lights chosen in proportion to area x emission, points uniform in area)
 * and connected through a shadow ray; emission found by a scattered ray then counts only after a mirror or
 * dielectric bounce or for the camera ray itself, so every light path is still counted once.  Same expected
 * image, less noise with small lights.  Both pipelines (the wavefront pipeline queues the shadow rays of a trip for a
 * closest-hit launch of their own); emissive planes (infinite area) are refused.
 * RRT_RENDER_EXACT_HITS: the f32 renderer tracks the rounding-error bound of every closest-hit decision (edge
 * coefficients, |n.d| - 1e-6, t - t_min, the order of two distances) and repeats in f64 the queries in which a decision
 * lies inside its bound, so that every ray it traces meets the primitive the f64 arithmetic of the reference
 * (the files under src/shapes) names for THAT ray; rrt_stats.exact_redos counts them.  The path state stays f32, so a path can
 * still part from the reference's where its f32 ray itself differs from the f64 one.  rrt_intersect and
 * rrt_trace_primary always work this way in f32 (their f64 input rays are what the repeated queries use).  Fused pipeline
 * only, f32 only (ignored for f64), not together with RRT_RENDER_NEE. */
typedef enum rrt_render_flags { RRT_RENDER_NEE = 1, RRT_RENDER_EXACT_HITS = 2 } rrt_render_flags;

/* One primitive in the order the reference's linear scan visits it:
 * objects in [[objects]] order (src/scene.rs:39), faces in file order inside a
 * mesh (src/shapes/mesh.rs:32).  object_id = index into scene.objects,
 * prim_id = face index inside the mesh, 0 for a non-mesh shape.
 * Transforms are already applied (the `transform` of each file under src/shapes). */
typedef struct rrt_prim {
    uint32_t type; /* rrt_prim_type */
    uint32_t object_id;
    uint32_t prim_id;
    uint32_t reserved;
    double v[12];
} rrt_prim;

typedef struct rrt_material {
    uint32_t type; /* rrt_material_type */
    uint32_t reserved;
    double rgb[3];
    double param;
} rrt_material;

/* State of PerspectiveCamera (src/camera.rs:13-18). */
typedef struct rrt_camera {
    double origin[3];
    double lower_left_corner[3];
    double horizontal[3];
    double vertical[3];
} rrt_camera;

typedef struct rrt_scene_desc {
    rrt_camera camera;
    uint32_t n_prims;
    uint32_t n_objects;
    const rrt_prim* prims;          /* n_prims, scan order */
    const rrt_material* materials;  /* n_objects, one per object (src/object.rs:8-11) */
} rrt_scene_desc;

/* [tracer] + [image] + [sampler] of a render config (src/renderer.rs:12-25,
 * src/tracers/mcpt.rs:14-18, src/sampler.rs:20-23,73-78) plus what a device
 * backend needs and the reference has no notion of: a seed, a sample-index
 * range (for sharding samples over GPUs), precision and kernel choices. */
typedef struct rrt_render_desc {
    uint32_t width, height;
    uint32_t sampler_type;     /* rrt_sampler_type */
    uint32_t spp;              /* samples_per_pixel as configured */
    uint32_t x_strata, y_strata; /* stratified only; x*y must equal spp (src/sampler.rs:87-91) */
    uint32_t min_depth, max_depth;
    uint64_t sample_begin;     /* per-pixel sample indices [begin, end) traced by this call */
    uint64_t sample_end;
    uint64_t seed;
    uint32_t precision;        /* rrt_precision */
    uint32_t accel;            /* rrt_accel */
    uint32_t pipeline;         /* rrt_pipeline */
    uint32_t flags;            /* rrt_render_flags, 0 = the reference's estimator */
} rrt_render_desc;

/* [post_processing] (src/renderer.rs:27-32). tone_mapping: 0 none, 1 reinhard. */
typedef struct rrt_post_desc {
    uint32_t tone_mapping;
    uint32_t gamma_correction;
    uint32_t has_white_balance;
    uint32_t reserved;
    double white_balance[3];
} rrt_post_desc;

typedef struct rrt_stats {
    uint64_t samples;        /* camera samples traced */
    uint64_t rays;           /* closest-hit queries = calls of Scene::intersect (src/tracers/utils.rs:36) */
    uint64_t prim_tests;     /* primitive tests executed */
    uint64_t node_tests;     /* tree node visits */
    uint64_t bounces[5];     /* scatter evaluations per rrt_material_type */
    uint64_t nonfinite;      /* samples whose radiance was NaN/Inf (src/tracers/utils.rs:78-85 only warns) */
    double kernel_ms;        /* device time of the tracing kernels (HIP events) */
    double reserved;
    /* What the call ran: AUTO resolved by the library, so that a caller reports the path taken, not a guess. */
    uint32_t accel_used;     /* rrt_accel: LINEAR, LBVH, SAH or SAH_GPU, never AUTO */
    uint32_t pipeline_used;  /* rrt_pipeline: FUSED or WAVEFRONT, never AUTO */
    uint32_t tree_depth;     /* levels of wide nodes of the tree that was walked (0: no tree) */
    uint32_t tree_built;     /* 1 when this call built the tree (the first call that needs it), else 0 */
    double tree_build_ms;    /* host wall time of that build, upload of its inputs included (0 when tree_built = 0) */
    uint64_t stack_overflows; /* traversal-stack entries dropped (always 0: the build refuses trees the stack cannot hold) */
    uint64_t exact_redos;    /* closest-hit queries repeated in f64 (RRT_RENDER_EXACT_HITS), see rrt_render_flags */
} rrt_stats;

typedef struct rrt_scene rrt_scene;
typedef struct rrt_backend rrt_backend;

/* Thread-local text of the last error on this thread ("" if none). */
const char* rrt_last_error(void);

/* PerspectiveCamera::new (src/camera.rs:21-34): basis and image-plane vectors
 * from look_from / look_at / vup / vertical fov in degrees / aspect. */
int rrt_camera_perspective(const double look_from[3], const double look_at[3], const double vup[3],
                           double vfov_deg, double aspect, rrt_camera* out);

/* Scene::from_config minus the TOML parsing (src/scene.rs:20-33): validates
 * and copies the description; the caller keeps ownership of `desc`. */
int rrt_scene_create(const rrt_scene_desc* desc, rrt_scene** out);
void rrt_scene_destroy(rrt_scene* scene);

/* Binds a backend to one GPU.  name must be "hip"; there is no CPU fallback.
 * Takes the place of the rayon pool set-up in render (src/renderer.rs:74-78). */
int rrt_backend_create(const char* name, int device_id, rrt_backend** out);
void rrt_backend_destroy(rrt_backend* backend);
/* Kernels and copies of later calls go to this hipStream_t (NULL = default). */
int rrt_backend_set_stream(rrt_backend* backend, void* hip_stream);

/* Scene::intersect (src/scene.rs:35-47) for n arbitrary rays.  origins/dirs are
 * 3n host doubles; results are host arrays of n.  A miss gives ids -1 and
 * t = +inf.  Directions must be unit length, as everywhere in the reference.
 * precision = F32 runs the product kernels' f32 tests and repeats in f64 the rays whose decisions f32 cannot settle
 * within its error bounds (see RRT_RENDER_EXACT_HITS), so the ids are those of the f64 query; t is the f32 distance. */
int rrt_intersect(rrt_backend* backend, const rrt_scene* scene, uint32_t precision, uint32_t accel,
                  size_t n, const double* origins, const double* dirs,
                  int32_t* object_id, int32_t* prim_id, double* t);

/* Camera::create_ray (src/camera.rs:38-45) followed by Scene::intersect, for n
 * image-plane coordinates (u, v) in [0,1]^2.  Host arrays. */
int rrt_trace_primary(rrt_backend* backend, const rrt_scene* scene, uint32_t precision, uint32_t accel,
                      size_t n, const double* u, const double* v,
                      int32_t* object_id, int32_t* prim_id, double* t);

/* rrt_intersect without any device allocation.  NO COUNTERPART IN THE REFERENCE: it exists for callers that must
 * not allocate device memory at the time of the call (tight serving loops, harnesses that forbid hipMalloc, a process
 * whose memory is spoken for).  rrt_intersect allocates five staging buffers per call; this variant stages the rays
 * and the results in a fixed array that lives in the library's device image (4 MiB per GPU the library runs on, held
 * until the process ends), so it takes at most RRT_INTERSECT_STATIC_MAX_RAYS rays per call and fails with
 * RRT_ERR_INVALID beyond that.  It still copies from and to the caller's HOST arrays and waits for the stream before it
 * returns, so it is NOT a call that can be recorded into a HIP graph.  The scene must already be resident on the
 * backend's device (and its tree built, if `accel` asks for one): one earlier rrt_intersect call with the same scene and
 * accel does both; otherwise RRT_ERR_INVALID.
 * Same arguments, results and kernels as rrt_intersect.  Every call on one GPU uses that GPU's one array: calls that
 * overlap on a GPU (two host threads, two backends on the same device) race and must be serialised by the caller. */
#define RRT_INTERSECT_STATIC_MAX_RAYS 65536
int rrt_intersect_static(rrt_backend* backend, const rrt_scene* scene, uint32_t precision, uint32_t accel,
                         size_t n, const double* origins, const double* dirs,
                         int32_t* object_id, int32_t* prim_id, double* t);

/* The pixel/sample loop of render (src/renderer.rs:101-128) with
 * MonteCarloPathTracer::trace (src/tracers/mcpt.rs:20-43) inside, for sample
 * indices [sample_begin, sample_end) of every pixel.  ADDS the radiance of the
 * traced samples to rgb_sum (width*height*3 floats, row-major, RGB), which the
 * caller owns and zeroes; stats may be NULL.
 * The sums are f32 (the reference adds in f64): with up to 2^20 samples per pixel in one sum the worst pixel is
 * within 5e-5 of the f64 sum (2^16: 3e-6), far inside the 8-bit output; at 2^24 the error reaches 7e-4 (fused) to
 * 2e-2 (wavefront, which adds every sample by itself).  A render with more samples per pixel than that belongs in
 * several calls over parts of the sample range, each into a sum of its own, added in f64 by the caller. */
int rrt_render_accumulate(rrt_backend* backend, const rrt_scene* scene, const rrt_render_desc* desc,
                          float* rgb_sum_dev_or_host, rrt_stats* stats);

/* color /= spp; post_process; quantise (src/renderer.rs:39-71,128-136):
 * reinhard -> gamma 1/2.2 -> white balance -> min(c*255,255) truncated to u8
 * (NaN becomes 255, negatives 0, as Rust's f64::min and `as u8` do). */
int rrt_postprocess(rrt_backend* backend, const float* rgb_sum_dev_or_host, uint64_t spp_divisor,
                    const rrt_post_desc* post, uint32_t width, uint32_t height,
                    uint8_t* rgb8_dev_or_host);

/* render (src/renderer.rs:73-150) in one call: zeroed accumulator, all samples
 * of desc (RandomSampler traces spp+1 samples and still divides by spp,
 * src/sampler.rs:48-55), post-process into rgb8 (host, width*height*3). */
int rrt_render(rrt_backend* backend, const rrt_scene* scene, const rrt_render_desc* desc,
               const rrt_post_desc* post, uint8_t* rgb8_host, rrt_stats* stats);

/* The numbers the device sampler hands to one camera sample, in the order the
 * path tracer asks for them on a diffuse path: get_2d (pixel jitter), get_1d,
 * get_2d, get_1d, get_2d -- i.e. the four stratified dimensions of
 * StratifiedSampler (src/sampler.rs:135-151) followed by one plain uniform
 * pair.  out receives 8 floats per sample index in [sample_begin, sample_end)
 * of pixel `pixel` (host memory).  Lets a test check on the device that every
 * stratum of every dimension is visited exactly once per pixel. */
int rrt_sampler_values(rrt_backend* backend, const rrt_render_desc* desc, uint32_t pixel, float* out_host);

/* Material::scatter (src/material.rs:38-284) and the throughput factor cos * bxdf / pdf of
 * generate_camera_vertices (src/tracers/utils.rs:66-87) for n independent samples of ONE material,
 * with the random numbers supplied by the caller: sample i has incoming direction d_in[3i..], surface
 * normal normal[3i..] (both unit vectors) and u[3i..] = (get_1d, get_2d.x, get_2d.y), the draws a
 * material may ask for (IdealDielectric uses the 1D draw, the two lobe samplers the 2D draw).
 * ok[i] = 1 when the material scatters (0 for the reference's None, and for a throughput of exactly
 * zero); then d_out[3i..] is the new direction and weight[3i..] the factor applied to the path
 * throughput.  All arrays are host memory.  A probe for tests, like rrt_sampler_values: the kernels
 * behind it are the device functions the renderers call per bounce. */
int rrt_material_probe(rrt_backend* backend, const rrt_material* material, uint32_t precision, size_t n,
                       const double* d_in, const double* normal, const double* u, double* d_out, double* weight,
                       int32_t* ok);

#ifdef __cplusplus
}
#endif
#endif /* RRT_H */
