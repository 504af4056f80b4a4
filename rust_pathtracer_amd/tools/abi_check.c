/* abi_check.c -- a plain C11 consumer of include/ptgpu.h, linked against the STATIC archive lib/libptgpu.a with the link line
 * of INTEGRATION.md section 1 (what a build.rs host links: the archive + amdhip64 + stdc++ + dl).
 *
 * It stands where the reference's one call site stands (renderer::render, src/renderer.rs:5-13, called from
 * src/main.rs:56-64): flattens a world of two spheres by hand, uploads it, asks BvhNode::hit (bvh.rs:107-121) for three rays
 * whose answers follow from Sphere::hit (geometry.rs:132-163) on paper, and renders one small frame.
 *
 * Exit status: 0 = every check passed, 77 = no GPU visible (nothing could be checked beyond linking and the error path),
 * anything else = a failed check (the message says which). */
#include <math.h>
#include <stdio.h>
#include <string.h>

#include "ptgpu.h"

static int failures = 0;
#define CHECK(cond, ...)                         \
    do {                                         \
        if (!(cond)) {                           \
            fprintf(stderr, "abi_check: ");      \
            fprintf(stderr, __VA_ARGS__);        \
            fprintf(stderr, "\n");               \
            ++failures;                          \
        }                                        \
    } while (0)

int main(void) {
    /* the C view of the PODs must be what the library was compiled with */
    _Static_assert(sizeof(PtPrim) == 104, "PtPrim layout");
    _Static_assert(sizeof(PtNode) == 56, "PtNode layout");
    _Static_assert(sizeof(PtMaterial) == 16, "PtMaterial layout");
    _Static_assert(sizeof(PtTexture) == 48, "PtTexture layout");
    _Static_assert(sizeof(PtCamera) == 152, "PtCamera layout");
    _Static_assert(sizeof(PtHitOut) == 40, "PtHitOut layout");

    PtContext* ctx = NULL;
    const int n_dev = pt_device_count();
    if (n_dev <= 0) {
        const int rc = pt_context_create(0, &ctx);
        CHECK(rc == PT_ERR_NO_DEVICE && ctx == NULL, "pt_context_create without a GPU returned %d", rc);
        CHECK(pt_last_error(NULL) != NULL && pt_last_error(NULL)[0] != '\0', "no message for the missing device");
        printf("abi_check: no GPU visible (%s); linked and refused as documented\n", pt_last_error(NULL));
        return failures ? 1 : 77;
    }
    int rc = pt_context_create(0, &ctx);
    CHECK(rc == PT_OK && ctx != NULL, "pt_context_create: %d %s", rc, pt_last_error(NULL));
    if (rc != PT_OK) return 1;

    /* world: sphere A centre (0,0,-5) r 1, sphere B centre (3,0,-5) r 1; BvhNode::new of two objects = one node (bvh.rs:77-82) */
    PtTexture tex;
    memset(&tex, 0, sizeof tex);
    tex.type = PT_TEX_SOLID;
    tex.color[0] = tex.color[1] = tex.color[2] = 0.5;
    PtMaterial mat;
    memset(&mat, 0, sizeof mat);
    mat.type = PT_MAT_LAMBERTIAN;
    mat.texture = 0;
    PtPrim prims[2];
    memset(prims, 0, sizeof prims);
    prims[0].type = PT_PRIM_SPHERE;
    prims[0].d[0] = 0.0; prims[0].d[1] = 0.0; prims[0].d[2] = -5.0; prims[0].d[3] = 1.0;
    prims[1].type = PT_PRIM_SPHERE;
    prims[1].d[0] = 3.0; prims[1].d[1] = 0.0; prims[1].d[2] = -5.0; prims[1].d[3] = 1.0;
    PtNode node;
    memset(&node, 0, sizeof node);
    node.bmin[0] = -1.0; node.bmin[1] = -1.0; node.bmin[2] = -6.0;
    node.bmax[0] = 4.0;  node.bmax[1] = 1.0;  node.bmax[2] = -4.0;
    node.left = PT_REF_LEAF | 0u;
    node.right = PT_REF_LEAF | 1u;
    PtSceneDesc desc;
    memset(&desc, 0, sizeof desc);
    desc.nodes = &node; desc.n_nodes = 1; desc.root = 0;
    desc.prims = prims; desc.n_prims = 2;
    desc.materials = &mat; desc.n_materials = 1;
    desc.textures = &tex; desc.n_textures = 1;

    /* before the upload: the documented refusal */
    double o0[3] = {0, 0, 0}, d0[3] = {0, 0, -1};
    PtHitOut h0;
    rc = pt_trace_batch(ctx, o0, d0, 1, 0.001, INFINITY, &h0);
    CHECK(rc == PT_ERR_NO_SCENE, "pt_trace_batch before pt_scene_upload returned %d", rc);

    rc = pt_scene_upload(ctx, &desc);
    CHECK(rc == PT_OK, "pt_scene_upload: %d %s", rc, pt_last_error(ctx));

    /* ray 0: o (0,0,0), d (0,0,-1) at A: oc = (0,0,5), a = 1, half_b = -5, c = 24, dis = 1, t = (5 - 1)/1 = 4; p = (0,0,-4),
     *        n = (p - c)/r = (0,0,1), d.n < 0 = front face.
     * ray 1: o (3,0,0), d (0,0,-2) at B: a = 4, half_b = -10, c = 24, dis = 4, t = (10 - 2)/4 = 2; n = (0,0,1), front.
     * ray 2: o (0,5,0), d (0,0,-1): passes 4 above both spheres: miss. */
    const double origins[9] = {0, 0, 0, 3, 0, 0, 0, 5, 0};
    const double dirs[9] = {0, 0, -1, 0, 0, -2, 0, 0, -1};
    PtHitOut hits[3];
    memset(hits, 0xff, sizeof hits);
    rc = pt_trace_batch(ctx, origins, dirs, 3, 0.001, INFINITY, hits);
    CHECK(rc == PT_OK, "pt_trace_batch: %d %s", rc, pt_last_error(ctx));
    CHECK(hits[0].prim == 0 && hits[0].t == 4.0 && hits[0].front_face == 1u, "ray 0: prim %d t %.17g front %u", hits[0].prim, hits[0].t, hits[0].front_face);
    CHECK(hits[0].normal[0] == 0.0 && hits[0].normal[1] == 0.0 && hits[0].normal[2] == 1.0, "ray 0: normal %g %g %g", hits[0].normal[0], hits[0].normal[1], hits[0].normal[2]);
    CHECK(hits[1].prim == 1 && hits[1].t == 2.0 && hits[1].front_face == 1u, "ray 1: prim %d t %.17g front %u", hits[1].prim, hits[1].t, hits[1].front_face);
    CHECK(hits[1].normal[0] == 0.0 && hits[1].normal[1] == 0.0 && hits[1].normal[2] == 1.0, "ray 1: normal %g %g %g", hits[1].normal[0], hits[1].normal[1], hits[1].normal[2]);
    CHECK(hits[2].prim == -1, "ray 2: prim %d (expected a miss)", hits[2].prim);

    /* the entry point that allocates nothing returns the same records */
    PtHitOut hits_static[3];
    memset(hits_static, 0xee, sizeof hits_static);
    rc = pt_trace_batch_static(ctx, origins, dirs, 3, 0.001, INFINITY, hits_static);
    CHECK(rc == PT_OK, "pt_trace_batch_static: %d %s", rc, pt_last_error(ctx));
    CHECK(memcmp(hits, hits_static, sizeof hits) == 0, "pt_trace_batch_static differs from pt_trace_batch");

    /* one frame, the reference's call shape: look down -z at the spheres from the origin, white background */
    PtCamera cam;
    memset(&cam, 0, sizeof cam);
    cam.lower_left_corner[0] = -4.0; cam.lower_left_corner[1] = -2.0; cam.lower_left_corner[2] = -5.0;
    cam.horizontal[0] = 8.0;
    cam.vertical[1] = 4.0;
    cam.u[0] = 1.0;
    cam.v[1] = 1.0;
    const double white[3] = {1.0, 1.0, 1.0};
    enum { W = 16, H = 8, SPP = 4 };
    unsigned char rgb[W * H * 3];
    PtStats stats;
    rc = pt_accum_reset(ctx, W, H);
    CHECK(rc == PT_OK, "pt_accum_reset: %d %s", rc, pt_last_error(ctx));
    rc = pt_render_accumulate(ctx, &cam, W, H, white, 0, SPP, 8, 1u, &stats);
    CHECK(rc == PT_OK, "pt_render_accumulate: %d %s", rc, pt_last_error(ctx));
    CHECK(stats.paths == (uint64_t)W * H * SPP && stats.rays >= stats.paths, "stats: %llu paths, %llu rays", (unsigned long long)stats.paths,
          (unsigned long long)stats.rays);
    rc = pt_resolve(ctx, SPP, rgb);
    CHECK(rc == PT_OK, "pt_resolve: %d %s", rc, pt_last_error(ctx));
    /* top-right pixel: u >= 15/15, v >= 7/7: direction (4+, 2+, -5), past both spheres -> the white background, 255 */
    CHECK(rgb[(0 * W + (W - 1)) * 3] == 255 && rgb[(0 * W + (W - 1)) * 3 + 1] == 255 && rgb[(0 * W + (W - 1)) * 3 + 2] == 255,
          "top-right pixel %u %u %u (expected the white background)", rgb[(W - 1) * 3], rgb[(W - 1) * 3 + 1], rgb[(W - 1) * 3 + 2]);
    /* pixel x = 7, y = 3 from the bottom (row 4 from the top): u in [7/15, 8/15), v in [3/7, 4/7): the ray crosses z = -5 within 0.4 of
     * sphere A's centre, so every sample hits it and carries at most albedo 0.5 of the background: sqrt(0.5) * 255.999 = 181 */
    CHECK(rgb[(4 * W + 7) * 3] <= 181, "pixel on sphere A is %u (expected at most 181)", rgb[(4 * W + 7) * 3]);

    pt_context_destroy(ctx);
    if (failures) return 1;
    printf("abi_check: ok (static archive, C11 header, %d device(s)): t = %.17g / %.17g, %llu rays in the frame\n", n_dev, hits[0].t, hits[1].t,
           (unsigned long long)stats.rays);
    return 0;
}
