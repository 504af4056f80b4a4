// host_selfcheck.cpp -- exercises the host layer (csrc/host: scene builders, BvhNode::new, flattening, the OBJ/MTL reader,
// the PNG writer, the C API's error paths) without a GPU, so that it can run under AddressSanitizer + UBSan
// (`make -C rust_pathtracer_amd asan`; SURVEY.md section 5: "host: -fsanitize=address,undefined test build").
// Exit code 0 = every check passed; any sanitizer report aborts the run.
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <string>
#include <vector>

#include "../../include/pthost.h"

static int g_failed = 0;
#define CHECK(cond)                                                        \
    do {                                                                   \
        if (!(cond)) {                                                     \
            std::fprintf(stderr, "FAILED %s:%d: %s\n", __FILE__, __LINE__, #cond); \
            ++g_failed;                                                    \
        }                                                                  \
    } while (0)

static void write_text(const std::string& path, const std::string& text) {
    std::ofstream f(path, std::ios::binary);
    f << text;
}

// every table index a kernel would follow stays inside its table
static void check_desc(const PtSceneDesc* d) {
    CHECK(d != nullptr);
    if (!d) return;
    CHECK(d->n_nodes >= 1 && d->n_prims >= 2 && d->n_materials >= 1);
    CHECK(!(d->root & PT_REF_LEAF) && d->root < d->n_nodes);
    uint64_t leaves = 0;
    for (uint32_t i = 0; i < d->n_nodes; ++i) {
        for (uint32_t ref : {d->nodes[i].left, d->nodes[i].right}) {
            if (ref & PT_REF_LEAF) {
                const uint32_t first = ref & PT_REF_INDEX_MASK, count = (ref & PT_REF_GROUP6) ? 6u : 1u;
                CHECK((uint64_t)first + count <= d->n_prims);
                leaves += count;
            } else {
                CHECK(ref < d->n_nodes);
            }
        }
        for (int k = 0; k < 3; ++k) CHECK(d->nodes[i].bmin[k] <= d->nodes[i].bmax[k]);
    }
    CHECK(leaves == d->n_prims);
    for (uint32_t i = 0; i < d->n_prims; ++i) CHECK(d->prims[i].type <= PT_PRIM_TRIANGLE && d->prims[i].material < d->n_materials);
    for (uint32_t i = 0; i < d->n_materials; ++i)
        CHECK(d->materials[i].type <= PT_MAT_DIFFUSE_LIGHT && (d->materials[i].type == PT_MAT_DIELECTRIC || d->materials[i].texture < d->n_textures));
    for (uint32_t i = 0; i < d->n_textures; ++i) {
        const PtTexture& t = d->textures[i];
        CHECK(t.type <= PT_TEX_PERLIN);
        if (t.type == PT_TEX_CHECKER) CHECK(t.odd < d->n_textures && t.even < d->n_textures);
        if (t.type == PT_TEX_PERLIN) CHECK(t.perlin_table < d->n_perlin_tables);
    }
}

int main(int argc, char** argv) {
    const std::string tmp = argc > 1 ? argv[1] : "/tmp";
    // ---- the six scenes, several seeds, with image overrides; cameras over the whole animation
    for (const char* name : {"SphereField", "TwoSpheres", "LightTest", "CornellBox", "TriangleTest"}) {
        for (uint64_t seed = 1; seed <= 3; ++seed) {
            PthScene* s = pth_scene_create(name);
            CHECK(s != nullptr);
            if (!s) continue;
            pth_scene_set_seeds(s, seed, seed + 1, seed + 2);
            const double bg[3] = {0.5, 0.25, 1.0};
            pth_scene_set_image(s, 97, 31, 3, 9, bg);
            CHECK(pth_scene_build(s) == 0);
            check_desc(pth_scene_desc(s));
            for (double t : {0.0, 0.37, 0.999}) {
                PtCamera cam;
                CHECK(pth_scene_camera(s, t, &cam) == 0);
                CHECK(std::isfinite(cam.origin[0]) && std::isfinite(cam.horizontal[0]) && std::isfinite(cam.lens_radius));
            }
            uint32_t w = 0, h = 0, spp = 0, depth = 0;
            double b[3], fps = 0, dur = 0;
            int anim = 0;
            CHECK(pth_scene_settings(s, &w, &h, &spp, &depth, b, &anim, &fps, &dur) == 0);
            CHECK(w == 97 && h == 31 && spp == 3 && depth == 9 && b[2] == 1.0);
            CHECK(pth_scene_build(s) == 0); // rebuilding frees the previous world
            pth_scene_destroy(s);
        }
    }
    CHECK(pth_scene_create("NoSuchScene") == nullptr);
    CHECK(std::strlen(pth_last_error()) > 0);

    // ---- ModelTest: synthetic mesh, then OBJ + MTL files with the reader's edge cases
    {
        PthScene* s = pth_scene_create("ModelTest");
        const double c[3] = {0.0, 1.5, 0.0};
        CHECK(pth_scene_set_model_synthetic(s, 40, 21, 1.3, c, 1.0) == 0);
        CHECK(pth_scene_build(s) == 0);
        const PtSceneDesc* d = pth_scene_desc(s);
        check_desc(d);
        CHECK(d && d->n_prims == 1u + 2u * 40u * 20u);
        CHECK(pth_scene_mesh_count(s) >= 1);
        PthMeshDesc m;
        CHECK(pth_scene_mesh(s, 0, &m) == 0 && m.n_indices % 3 == 0 && m.n_positions % 3 == 0);
        CHECK(pth_scene_mesh(s, 99, &m) != 0);
        pth_scene_destroy(s);
    }
    {
        const std::string obj = tmp + "/selfcheck_model.obj", mtl = tmp + "/selfcheck_model.mtl";
        write_text(mtl, "newmtl glass\r\nillum 7\r\nNi 1.5\r\nKd 1 1 1\r\nnewmtl mirror\nillum 5\nKd 0.8 0.8 0.9\nNs 50\nnewmtl paint\nKd 0.7 0.2 0.2\n");
        write_text(obj,
                   "# comment\r\nmtllib selfcheck_model.mtl\r\no pyramid\r\nv 0 0 0\nv 2 0 0\nv 2 0 2\nv 0 0 2\nv 1 2 1\nvn 0 -1 0\nvt 0 0\n"
                   "usemtl glass\nf 1//1 2//1 3//1 4//1\nusemtl mirror\nf 1/1 5/1 2/1\nf 2/1 5/1 3/1\nusemtl paint\nf 3 5 4\nf -2 -1 -5\n"
                   "usemtl nobody\nf 4 5 1\no slab\nv -3 0.5 -1\nv -1 0.5 -1\nv -1 0.5 1\nv -2 0.5 2\nv -3 0.5 1\ng top\nf 6 10 9 8 7\n\n");
        uint32_t models = 0, mats = 0;
        uint64_t tris = 0;
        CHECK(pth_obj_probe(obj.c_str(), &models, &tris, &mats) == 0);
        CHECK(tris == 2 + 1 + 1 + 1 + 1 + 1 + 3 && mats == 3 && models >= 5);
        PthScene* s = pth_scene_create("ModelTest");
        CHECK(pth_scene_set_model_path(s, obj.c_str()) == 0);
        CHECK(pth_scene_build(s) == 0);
        check_desc(pth_scene_desc(s));
        CHECK(pth_scene_mtl_count(s) == 3);
        pth_scene_destroy(s);
        // broken inputs: an index past the vertices, a face with two corners, a missing file, a missing MTL, an empty file
        write_text(tmp + "/selfcheck_bad1.obj", "v 0 0 0\nv 1 0 0\nv 0 1 0\nf 1 2 9\n");
        write_text(tmp + "/selfcheck_bad2.obj", "v 0 0 0\nv 1 0 0\nf 1 2\n");
        write_text(tmp + "/selfcheck_bad3.obj", "mtllib does_not_exist.mtl\nv 0 0 0\nv 1 0 0\nv 0 1 0\nf 1 2 3\n");
        write_text(tmp + "/selfcheck_bad4.obj", "");
        write_text(tmp + "/selfcheck_bad5.obj", "v 0 0\nv nan x 1\nf 0 0 0\nf 1 -7 2\nvn 1\nusemtl\n");
        for (const char* bad : {"/selfcheck_bad1.obj", "/selfcheck_bad2.obj", "/selfcheck_bad3.obj", "/selfcheck_bad4.obj", "/selfcheck_bad5.obj", "/selfcheck_nope.obj"}) {
            uint32_t a = 0, b = 0;
            uint64_t t = 0;
            (void)pth_obj_probe((tmp + bad).c_str(), &a, &t, &b); // must fail or succeed, never touch memory it does not own
            PthScene* q = pth_scene_create("ModelTest");
            if (pth_scene_set_model_path(q, (tmp + bad).c_str()) == 0) {
                const int rb = pth_scene_build(q);
                if (rb == 0) check_desc(pth_scene_desc(q));
            }
            pth_scene_destroy(q);
        }
    }

    // ---- worlds from object lists (pth_world_create), reference error behaviour
    {
        PthTextureDesc tex[3];
        std::memset(tex, 0, sizeof(tex));
        tex[0].type = PT_TEX_SOLID; tex[0].color[0] = 0.5;
        tex[1].type = PT_TEX_CHECKER; tex[1].odd = 0; tex[1].even = 2;
        tex[2].type = PT_TEX_PERLIN; tex[2].scale = 4.0; tex[2].perlin_seed = 7;
        PtMaterial mat[3] = {{PT_MAT_LAMBERTIAN, 1, 0.0}, {PT_MAT_METAL, 2, 0.3}, {PT_MAT_DIELECTRIC, 0, 1.5}};
        std::vector<PthObjectDesc> objs(40);
        for (size_t i = 0; i < objs.size(); ++i) {
            PthObjectDesc& o = objs[i];
            std::memset(&o, 0, sizeof(o));
            o.material = (uint32_t)(i % 3);
            const double x = (double)(i % 7), y = (double)(i % 5), z = (double)(i % 3);
            switch (i % 6) {
                case 0: o.kind = 0; o.p[0] = x; o.p[1] = y; o.p[2] = z; o.p[3] = (i % 4 == 0) ? -0.4 : 0.5; break;
                case 1: o.kind = 1; o.p[0] = x; o.p[1] = y; o.p[2] = z; o.p[3] = x + 1; o.p[4] = y - 2; o.p[5] = z; o.p[6] = -2.5; break;
                case 2: o.kind = 2; o.p[0] = x; o.p[1] = y; o.p[2] = z; o.p[3] = x - 1; o.p[4] = y; o.p[5] = z + 2; o.p[6] = 1.0; break;
                case 3: o.kind = 3; o.p[0] = x; o.p[1] = y; o.p[2] = z; o.p[3] = x; o.p[4] = y + 1; o.p[5] = z + 1; o.p[6] = 1.0; break;
                case 4: o.kind = 4; o.p[0] = x; o.p[1] = y; o.p[2] = z; o.p[3] = x + 1; o.p[4] = y + 2; o.p[5] = z - 1; break;
                default: o.kind = 6; o.p[0] = x; o.p[1] = y; o.p[2] = z; o.p[3] = x + 1; o.p[4] = y; o.p[5] = z; o.p[6] = x; o.p[7] = y + 1; o.p[8] = z + 1; break;
            }
        }
        PthScene* w = pth_world_create(objs.data(), (uint32_t)objs.size(), mat, 3, tex, 3, nullptr, 0, nullptr, 0, 5);
        CHECK(w != nullptr);
        if (w) {
            check_desc(pth_scene_desc(w));
            CHECK(pth_scene_set_rebuild_per_frame(w, 1) == 0);
            pth_scene_destroy(w);
        }
        CHECK(pth_world_create(objs.data(), 1, mat, 3, tex, 3, nullptr, 0, nullptr, 0, 5) == nullptr); // bvh.rs:76
        CHECK(std::strstr(pth_last_error(), "empty object list") != nullptr);
        PthObjectDesc skew = objs[1];
        skew.p[5] = skew.p[2] + 1.0; // an XY rectangle whose ends differ in z: geometry.rs:173-206
        PthObjectDesc two[2] = {objs[0], skew};
        CHECK(pth_world_create(two, 2, mat, 3, tex, 3, nullptr, 0, nullptr, 0, 5) == nullptr);
        CHECK(std::strstr(pth_last_error(), "not aligned") != nullptr);
        PthObjectDesc badmat = objs[0];
        badmat.material = 17;
        PthObjectDesc two2[2] = {objs[0], badmat};
        CHECK(pth_world_create(two2, 2, mat, 3, tex, 3, nullptr, 0, nullptr, 0, 5) == nullptr);
    }

    // ---- PNG writer: sizes around the deflate block and row boundaries, and an unwritable path
    {
        for (uint32_t wh : {1u, 2u, 63u, 400u}) {
            std::vector<uint8_t> px((size_t)wh * (wh + 3) * 3);
            for (size_t i = 0; i < px.size(); ++i) px[i] = (uint8_t)(i * 131u + 7u);
            CHECK(pth_write_png((tmp + "/selfcheck.png").c_str(), wh, wh + 3, px.data()) == 0);
        }
        uint8_t one[3] = {1, 2, 3};
        CHECK(pth_write_png((tmp + "/no_such_dir_selfcheck/a.png").c_str(), 1, 1, one) != 0);
        CHECK(pth_write_png(nullptr, 1, 1, one) != 0);
    }

    // ---- rendering without a GPU fails cleanly (no CPU fallback), with a GPU it runs: either way no memory error
    {
        PthScene* s = pth_scene_create("CornellBox");
        const double bg[3] = {0, 0, 0};
        pth_scene_set_image(s, 16, 8, 1, 3, bg);
        CHECK(pth_scene_build(s) == 0);
        std::vector<uint8_t> img(16 * 8 * 3);
        PtStats st;
        const int rc = pth_render(s, 0, 0.0, 1, img.data(), &st);
        if (rc != 0) std::printf("pth_render without a usable GPU: \"%s\" (expected here)\n", pth_last_error());
        CHECK(pth_render(nullptr, 0, 0.0, 1, img.data(), &st) != 0);
        pth_scene_destroy(s);
    }
    {
        double a[3] = {0, 0, -1}, b[3] = {0, 0, 0}, up[3] = {0, 1, 0};
        PtCamera cam;
        pth_camera_new(a, b, up, 40.0, 1.5, 0.1, 10.0, &cam);
        CHECK(cam.lens_radius == 0.05);
    }
    if (g_failed) {
        std::fprintf(stderr, "host_selfcheck: %d check(s) failed\n", g_failed);
        return 1;
    }
    std::printf("host_selfcheck: all checks passed\n");
    return 0;
}
