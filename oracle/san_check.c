/*
 * san_check.c - drives the oracle's entry points on a small scene with every primitive and material
 * type, for a build with -fsanitize=address,undefined (make -C oracle san_check; run by
 * tests/test_oracle_render.py).  TEST INFRASTRUCTURE, like everything in this directory.
 */
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "oracle.h"

static void prim(rrt_prim* p, uint32_t type, uint32_t object, uint32_t id, const double* v, int n) {
    memset(p, 0, sizeof(*p));
    p->type = type;
    p->object_id = object;
    p->prim_id = id;
    memcpy(p->v, v, sizeof(double) * (size_t)n);
}

int main(void) {
    rrt_prim prims[6];
    const double sphere[4] = {0.0, 0.5, -4.0, 1.0}, plane[6] = {0, -1, 0, 0, 1, 0};
    const double tri[9] = {-3, -1, -6, -1, -1, -6, -2, 2, -6};
    const double quad[12] = {1, -1, -6, 3, -1, -6, 3, 1.5, -6, 1, 1.5, -6};
    const double mirror[4] = {-2.0, 0.0, -3.0, 0.6}, light[4] = {0, 0, 0, 30.0};
    prim(&prims[0], RRT_PRIM_SPHERE, 0, 0, sphere, 4);
    prim(&prims[1], RRT_PRIM_PLANE, 1, 0, plane, 6);
    prim(&prims[2], RRT_PRIM_TRIANGLE, 2, 0, tri, 9);
    prim(&prims[3], RRT_PRIM_QUAD, 2, 1, quad, 12);
    prim(&prims[4], RRT_PRIM_SPHERE, 3, 0, mirror, 4);
    prim(&prims[5], RRT_PRIM_SPHERE, 4, 0, light, 4);
    rrt_material mats[5];
    memset(mats, 0, sizeof(mats));
    mats[0].type = RRT_MAT_DIELECTRIC; mats[0].param = 1.5;
    mats[1].type = RRT_MAT_LAMBERTIAN; mats[1].rgb[0] = mats[1].rgb[1] = mats[1].rgb[2] = 0.6;
    mats[2].type = RRT_MAT_PHONG; mats[2].rgb[0] = 0.7; mats[2].rgb[1] = 0.7; mats[2].rgb[2] = 0.2; mats[2].param = 8.0;
    mats[3].type = RRT_MAT_REFLECTOR;
    mats[4].type = RRT_MAT_EMISSIVE; mats[4].rgb[0] = mats[4].rgb[1] = mats[4].rgb[2] = 1.0;
    rrt_scene_desc scene;
    memset(&scene, 0, sizeof(scene));
    const double from[3] = {0, 0.5, 2}, at[3] = {0, 0.2, -4}, up[3] = {0, 1, 0};
    if (oracle_camera_perspective(from, at, up, 55.0, 1.5, &scene.camera) != 0) return 1;
    scene.prims = prims; scene.n_prims = 6;
    scene.materials = mats; scene.n_objects = 5;

    enum { W = 45, H = 30 };  /* ragged against the 16x16 tiles */
    double* rgb = (double*)malloc(sizeof(double) * W * H * 3);
    uint8_t* out = (uint8_t*)malloc(W * H * 3);
    double* lin = (double*)malloc(sizeof(double) * W * H * 3);
    if (!rgb || !out || !lin) return 1;
    rrt_post_desc post;
    memset(&post, 0, sizeof(post));
    post.tone_mapping = 1; post.gamma_correction = 1;
    int fails = 0;
    for (int sampler = 0; sampler < 2; ++sampler) {
        for (int mode = 0; mode < 3; ++mode) {
            rrt_render_desc rd;
            memset(&rd, 0, sizeof(rd));
            rd.width = W; rd.height = H;
            rd.sampler_type = sampler ? RRT_SAMPLER_STRATIFIED : RRT_SAMPLER_RANDOM;
            rd.spp = 12; rd.x_strata = 4; rd.y_strata = 3;
            rd.min_depth = 1; rd.max_depth = 7; rd.seed = 5; rd.sample_end = 0; /* every sample of the pixel loop */
            rrt_stats st;
            memset(rgb, 0, sizeof(double) * W * H * 3);
            int rc = mode == 0 ? oracle_render_accumulate(&scene, &rd, rgb, &st, 3)
                               : oracle_render_accumulate_device_numbers(&scene, &rd, rgb, &st, 3, mode == 2);
            uint64_t want = (uint64_t)W * H * (rd.spp + (sampler ? 0 : 1));
            if (rc != 0 || st.samples != want || st.nonfinite != 0) { printf("render %d/%d: rc %d\n", sampler, mode, rc); fails++; }
            if (oracle_postprocess(rgb, rd.spp, &post, W, H, out, lin) != 0) fails++;
        }
    }
    double u[4] = {0.1, 0.5, 0.9, 0.5}, v[4] = {0.5, 0.5, 0.2, 0.95}, t[4];
    int32_t oid[4], pid[4];
    if (oracle_trace_primary(&scene, 4, u, v, oid, pid, t) != 0) fails++;
    free(rgb); free(out); free(lin);
    printf(fails ? "san_check: %d failures\n" : "san_check ok\n", fails);
    return fails ? 1 : 0;
}
