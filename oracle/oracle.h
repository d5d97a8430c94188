/*
 * oracle.h - entry points of the f64 CPU restatement (see oracle.c).
 * TEST INFRASTRUCTURE: loaded only by tests/, __graft_entry__.smoke() and the
 * cpu_baseline leg of bench.py.  It shares the plain-data structs of
 * include/rrt.h with the product library and nothing else.
 */
#ifndef ORACLE_H
#define ORACLE_H
#include "../include/rrt.h"

#ifdef __cplusplus
extern "C" {
#endif

int oracle_camera_perspective(const double look_from[3], const double look_at[3],
                              const double vup[3], double vfov, double aspect, rrt_camera* out);
int oracle_intersect(const rrt_scene_desc* scene, size_t n, const double* origins,
                     const double* dirs, int32_t* object_id, int32_t* prim_id, double* t);
int oracle_trace_primary(const rrt_scene_desc* scene, size_t n, const double* u, const double* v,
                         int32_t* object_id, int32_t* prim_id, double* t);
int oracle_render_accumulate(const rrt_scene_desc* scene, const rrt_render_desc* rd,
                             double* rgb_sum, rrt_stats* stats, int n_threads);
/* the same render drawing its random numbers from the restated generator of the HIP backend; f32_kernels: as the
 * f32 kernels see them (24-bit uniforms) and with their lobe basis */
int oracle_render_accumulate_device_numbers(const rrt_scene_desc* scene, const rrt_render_desc* rd,
                             double* rgb_sum, rrt_stats* stats, int n_threads, int f32_kernels);
/* a bounding-volume hierarchy for the oracle's own closest-hit queries on `scene` (until oracle_accel_free): same
 * results as its scan, for scenes the scan is too slow for */
int oracle_accel_build(const rrt_scene_desc* scene);
void oracle_accel_free(void);
int oracle_postprocess(const double* rgb_sum, uint64_t spp_divisor, const rrt_post_desc* post,
                       uint32_t width, uint32_t height, uint8_t* rgb8, double* rgb_linear);

void oracle_ray_at(const double o[3], const double d[3], double t, double out[3]);
void oracle_create_ray(const rrt_camera* cam, double s, double t, double o[3], double d[3]);
void oracle_reflect(const double v[3], const double n[3], double out[3]);
int oracle_refract(const double v[3], const double n[3], double eta, double out[3]);
double oracle_fresnel(double cos_i, double eta_i, double eta_t);
void oracle_local_coordinate_system(const double n[3], double u[3], double v[3], double w[3]);
void oracle_spherical_to_world(double theta, double phi, const double n[3], double out[3]);
int oracle_prim_intersect(const rrt_prim* prim, const double o[3], const double d[3], double t_min,
                          double t_max, double* t, double p[3], double normal[3]);
int oracle_sphere_intersect_analytic(const double sphere[4], const double o[3], const double d[3],
                                     double t_min, double t_max, double* t);
int oracle_are_points_coplanar(const double q[12]);
int oracle_is_quadrilateral_convex(const double q[12]);
int oracle_material_sample(const rrt_material* m, const double d_in[3], const double p[3],
                           const double normal[3], double u1, double u2a, double u2b,
                           double d_out[3], double* pdf, double bxdf[3], double weight[3]);
int oracle_stratified_tables(uint32_t spp, uint32_t x_strata, uint32_t y_strata, uint64_t seed,
                             double* out1d, double* out2d);
int oracle_trace_samples(const rrt_scene_desc* scene, const rrt_render_desc* rd, size_t n,
                         const double* u, const double* v, double* rgb);

#ifdef __cplusplus
}
#endif
#endif
