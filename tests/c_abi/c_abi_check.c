/* A plain-C client of include/rrt.h: what a binding in any language has to do.
 *   c_abi_check cpu   : struct sizes, the host-only entry points, argument checking (no GPU needed)
 *   c_abi_check gpu   : a furnace scene end to end on device 0 (every pixel is exactly the emission)
 * Exit status 0 on success; the first failed check is printed. */
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "../../include/rrt.h"

#define CHECK(cond)                                                              \
    do {                                                                         \
        if (!(cond)) {                                                           \
            fprintf(stderr, "check failed at line %d: %s (last error: %s)\n", __LINE__, #cond, rrt_last_error()); \
            return 1;                                                            \
        }                                                                        \
    } while (0)

static int cpu_checks(void) {
    /* layouts the ctypes and Rust mirrors rely on */
    CHECK(sizeof(rrt_prim) == 112);
    CHECK(sizeof(rrt_material) == 40);
    CHECK(sizeof(rrt_camera) == 96);
    CHECK(sizeof(rrt_render_desc) == 72);
    /* test_perspective_camera, src/camera.rs:82-102 */
    const double from[3] = {0, 0, 0}, at[3] = {0, 0, -1}, up[3] = {0, 1, 0};
    rrt_camera cam;
    CHECK(rrt_camera_perspective(from, at, up, 90.0, 2.0, &cam) == RRT_OK);
    CHECK(fabs(cam.lower_left_corner[0] + 2.0) < 1e-12 && fabs(cam.lower_left_corner[1] + 1.0) < 1e-12 &&
          fabs(cam.lower_left_corner[2] + 1.0) < 1e-12);
    CHECK(fabs(cam.horizontal[0] - 4.0) < 1e-12 && fabs(cam.vertical[1] - 2.0) < 1e-12);
    /* argument checking never aborts */
    rrt_scene* scene = NULL;
    CHECK(rrt_scene_create(NULL, &scene) < 0 && strlen(rrt_last_error()) > 0);
    rrt_prim bad;
    memset(&bad, 0, sizeof(bad));
    bad.type = 9;
    rrt_material mat;
    memset(&mat, 0, sizeof(mat));
    rrt_scene_desc desc;
    memset(&desc, 0, sizeof(desc));
    desc.camera = cam;
    desc.n_prims = 1;
    desc.n_objects = 1;
    desc.prims = &bad;
    desc.materials = &mat;
    CHECK(rrt_scene_create(&desc, &scene) < 0 && scene == NULL);
    rrt_backend* be = NULL;
    CHECK(rrt_backend_create("cpu-f64", 0, &be) == RRT_ERR_UNSUPPORTED && be == NULL); /* no CPU path in this library */
    rrt_scene_destroy(NULL);
    rrt_backend_destroy(NULL);
    return 0;
}

static int gpu_checks(void) {
    const double from[3] = {0, 0, 0}, at[3] = {0, 0, -1}, up[3] = {0, 1, 0};
    rrt_prim sphere;
    memset(&sphere, 0, sizeof(sphere));
    sphere.type = RRT_PRIM_SPHERE;
    sphere.v[3] = 10.0; /* centre 0,0,0 radius 10: the camera sits inside */
    rrt_material light;
    memset(&light, 0, sizeof(light));
    light.type = RRT_MAT_EMISSIVE;
    light.rgb[0] = 3.0; light.rgb[1] = 2.0; light.rgb[2] = 1.0;
    rrt_scene_desc desc;
    memset(&desc, 0, sizeof(desc));
    CHECK(rrt_camera_perspective(from, at, up, 60.0, 1.0, &desc.camera) == RRT_OK);
    desc.n_prims = 1;
    desc.n_objects = 1;
    desc.prims = &sphere;
    desc.materials = &light;
    rrt_scene* scene = NULL;
    rrt_backend* be = NULL;
    CHECK(rrt_scene_create(&desc, &scene) == RRT_OK);
    CHECK(rrt_backend_create("hip", 0, &be) == RRT_OK);

    rrt_render_desc rd;
    memset(&rd, 0, sizeof(rd));
    rd.width = 24; rd.height = 16;
    rd.sampler_type = RRT_SAMPLER_STRATIFIED;
    rd.spp = 16; rd.x_strata = 4; rd.y_strata = 4;
    rd.min_depth = 2; rd.max_depth = 6;
    rd.sample_end = 16;
    rrt_post_desc pd;
    memset(&pd, 0, sizeof(pd));
    pd.tone_mapping = 1;
    pd.gamma_correction = 1;
    unsigned char img[24 * 16 * 3];
    rrt_stats st;
    CHECK(rrt_render(be, scene, &rd, &pd, img, &st) == RRT_OK);
    CHECK(st.samples == 24u * 16u * 16u && st.rays == st.samples && st.nonfinite == 0);
    /* L -> L/(1+L) -> ^(1/2.2) -> *255 truncated (src/renderer.rs:39-71,128-136) */
    const double L[3] = {3.0, 2.0, 1.0};
    for (int c = 0; c < 3; ++c) {
        unsigned char expect = (unsigned char)(pow(L[c] / (L[c] + 1.0), 1.0 / 2.2) * 255.0);
        for (int p = 0; p < 24 * 16; ++p) CHECK(img[3 * p + c] == expect);
    }
    /* closest-hit parity entry point: centre of the view hits the sphere at t = 10 */
    const double u = 0.5, v = 0.5;
    int32_t oid = -2, pid = -2;
    double t = 0;
    CHECK(rrt_trace_primary(be, scene, RRT_PRECISION_F64, RRT_ACCEL_AUTO, 1, &u, &v, &oid, &pid, &t) == RRT_OK);
    CHECK(oid == 0 && pid == 0 && fabs(t - 10.0) < 1e-12);
    rd.sample_end = 99; /* beyond spp for a stratified sampler */
    float sums[24 * 16 * 3];
    memset(sums, 0, sizeof(sums));
    CHECK(rrt_render_accumulate(be, scene, &rd, sums, &st) == RRT_ERR_INVALID);
    rd.sample_begin = 4; rd.sample_end = 12; /* a shard: 8 of the 16 sample indices of every pixel, into a host buffer */
    CHECK(rrt_render_accumulate(be, scene, &rd, sums, &st) == RRT_OK);
    CHECK(st.samples == 24u * 16u * 8u && sums[0] == 24.0f && sums[1] == 16.0f && sums[2] == 8.0f);
    rrt_backend_destroy(be);
    rrt_scene_destroy(scene);
    return 0;
}

int main(int argc, char** argv) {
    if (argc == 2 && strcmp(argv[1], "cpu") == 0) return cpu_checks();
    if (argc == 2 && strcmp(argv[1], "gpu") == 0) return gpu_checks();
    fprintf(stderr, "usage: c_abi_check cpu|gpu\n");
    return 2;
}
