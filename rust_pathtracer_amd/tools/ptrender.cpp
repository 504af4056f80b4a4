// ptrender -- the reference's main() (main.rs:19-75) on the MI355X backend: pick a scene, render its frames, write
// ./output/image_%04d.png.  main.rs hard-codes the scene and takes no arguments (its clap dependency is unused); the
// scene name and the ImageSettings overrides are arguments here.  Thin: everything happens in libpthost.
//
//   ptrender SCENE [--width W --height H --spp N --depth D] [--out DIR] [--device I] [--seed S]
//                  [--frames N] [--first-frame K --frame-stride M] [--rebuild-per-frame] [--model PATH | --synthetic-mesh NLON NLAT]
#include <sys/stat.h>

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>

#include "pthost.h"

static int fail(const char* what) {
    std::fprintf(stderr, "ptrender: %s: %s\n", what, pth_last_error());
    return 1;
}

int main(int argc, char** argv) {
    if (argc < 2) {
        std::fprintf(stderr,
                     "usage: ptrender SCENE [--width W --height H --spp N --depth D] [--out DIR] [--device I] [--seed S]\n"
                     "                [--frames N] [--first-frame K --frame-stride M] [--rebuild-per-frame] [--quiet] [--model PATH | --synthetic-mesh NLON NLAT]\n"
                     "scenes: SphereField TwoSpheres LightTest CornellBox TriangleTest ModelTest\n");
        return 2;
    }
    const char* name = argv[1];
    std::string out = "./output", model;
    long width = 0, height = 0, spp = 0, depth = 0, device = 0, frames = 0, first = 0, stride = 1, nlon = 0, nlat = 0;
    unsigned long long seed = 1;
    int rebuild = 0, quiet = 0;
    for (int i = 2; i < argc; ++i) {
        auto need = [&](int n) {
            if (i + n >= argc) {
                std::fprintf(stderr, "ptrender: %s needs %d value(s)\n", argv[i], n);
                std::exit(2);
            }
        };
        if (!std::strcmp(argv[i], "--width")) { need(1); width = std::atol(argv[++i]); }
        else if (!std::strcmp(argv[i], "--height")) { need(1); height = std::atol(argv[++i]); }
        else if (!std::strcmp(argv[i], "--spp")) { need(1); spp = std::atol(argv[++i]); }
        else if (!std::strcmp(argv[i], "--depth")) { need(1); depth = std::atol(argv[++i]); }
        else if (!std::strcmp(argv[i], "--out")) { need(1); out = argv[++i]; }
        else if (!std::strcmp(argv[i], "--device")) { need(1); device = std::atol(argv[++i]); }
        else if (!std::strcmp(argv[i], "--seed")) { need(1); seed = std::strtoull(argv[++i], nullptr, 0); }
        else if (!std::strcmp(argv[i], "--frames")) { need(1); frames = std::atol(argv[++i]); }
        else if (!std::strcmp(argv[i], "--first-frame")) { need(1); first = std::atol(argv[++i]); }
        else if (!std::strcmp(argv[i], "--frame-stride")) { need(1); stride = std::atol(argv[++i]); }
        else if (!std::strcmp(argv[i], "--rebuild-per-frame")) { rebuild = 1; }
        else if (!std::strcmp(argv[i], "--quiet")) { quiet = 1; }
        else if (!std::strcmp(argv[i], "--model")) { need(1); model = argv[++i]; }
        else if (!std::strcmp(argv[i], "--synthetic-mesh")) { need(2); nlon = std::atol(argv[++i]); nlat = std::atol(argv[++i]); }
        else {
            std::fprintf(stderr, "ptrender: unknown argument %s\n", argv[i]);
            return 2;
        }
    }
    PthScene* s = pth_scene_create(name);
    if (!s) return fail("unknown scene");
    pth_scene_set_seeds(s, seed, seed, seed);
    if (!model.empty() && pth_scene_set_model_path(s, model.c_str()) != 0) return fail("--model");
    const double center[3] = {0.0, 1.5, 0.0};
    if (nlon > 0 && pth_scene_set_model_synthetic(s, (uint32_t)nlon, (uint32_t)nlat, 1.2, center, 1.0) != 0) return fail("--synthetic-mesh");
    uint32_t w0, h0, spp0, depth0;
    double bg[3], fps, duration;
    int animation;
    if (pth_scene_settings(s, &w0, &h0, &spp0, &depth0, bg, &animation, &fps, &duration) != 0) return fail("settings");
    if (width > 0 || height > 0 || spp > 0 || depth > 0)
        pth_scene_set_image(s, width > 0 ? (uint32_t)width : w0, height > 0 ? (uint32_t)height : h0, spp > 0 ? (uint32_t)spp : spp0,
                            depth > 0 ? (uint32_t)depth : depth0, bg);
    if (pth_scene_build(s) != 0) return fail("building the world");
    if (pth_scene_set_rebuild_per_frame(s, rebuild) != 0) return fail("--rebuild-per-frame");
    mkdir(out.c_str(), 0777);  // main.rs expects ./output to exist; create it when it does not
    uint32_t total = 0, written = 0;
    double t_render = 0.0, t_write = 0.0;
    uint64_t rays = 0;
    // main.rs:42-49,73-75 draws a progress bar per frame; here one line per frame on stderr (--quiet turns it off)
    if (!quiet)
        pth_set_progress_callback(
            [](uint32_t done, uint32_t to_do, uint32_t frame, double seconds, void*) {
                std::fprintf(stderr, "[%u/%u] frame %04u rendered in %.3f s\n", done, to_do, frame, seconds);
            },
            nullptr);
    if (pth_render_frames(s, (int)device, seed, out.c_str(), (uint32_t)first, (uint32_t)stride, (uint32_t)frames, &total, &written, &t_render,
                          &t_write, &rays) != 0)
        return fail("rendering");
    std::printf("%s: %u of %u frame(s) -> %s/image_%%04d.png; render %.3f s (%.1f Mrays/s), png %.3f s (overlapped)\n", name, written, total,
                out.c_str(), t_render, t_render > 0 ? rays / t_render / 1e6 : 0.0, t_write);
    if (rebuild) {
        double t_build = 0.0, t_upload = 0.0;
        pth_scene_rebuild_seconds(s, &t_build, &t_upload);
        std::printf("%s: world rebuilt per frame: BvhNode::new %.3f s, flatten + upload %.3f s (not in the render time)\n", name, t_build, t_upload);
    }
    pth_scene_destroy(s);
    return 0;
}
