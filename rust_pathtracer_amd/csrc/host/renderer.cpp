// renderer.cpp -- see renderer.hpp.
#include "renderer.hpp"

#include <cstdlib>
#include <mutex>

namespace pt {

static void check(PtContext* ctx, int rc, const char* what) {
    if (rc != PT_OK) throw std::runtime_error(std::string(what) + ": " + pt_last_error(ctx));
}

GpuRenderer::GpuRenderer(int device) {
    int rc = pt_context_create(device, &ctx_);
    if (rc != PT_OK) throw std::runtime_error(std::string("pt_context_create: ") + pt_last_error(nullptr));
}
GpuRenderer::~GpuRenderer() { pt_context_destroy(ctx_); }

void GpuRenderer::upload(const BvhNode& world) {
    Flattener f;
    f.flatten_world(world);
    PtSceneDesc d = f.desc();
    check(ctx_, pt_scene_upload(ctx_, &d), "pt_scene_upload");
}

std::vector<uint8_t> GpuRenderer::render(const Camera& camera, size_t width, size_t height, const Color& background,
                                         size_t samples_per_pixel, size_t max_bounces, uint64_t seed, PtStats* stats) {
    if (width > 0xFFFFFFFFull || height > 0xFFFFFFFFull || samples_per_pixel > 0xFFFFFFFFull || max_bounces > 0xFFFFFFFFull)
        throw std::runtime_error("render: argument out of range");
    PtCamera cam = camera.to_pod();
    check(ctx_, pt_accum_reset(ctx_, (uint32_t)width, (uint32_t)height), "pt_accum_reset");
    check(ctx_,
          pt_render_accumulate(ctx_, &cam, (uint32_t)width, (uint32_t)height, background.e, 0, (uint32_t)samples_per_pixel,
                               (uint32_t)max_bounces, seed, stats),
          "pt_render_accumulate");
    std::vector<uint8_t> out(width * height * 3);
    if (samples_per_pixel == 0) return out; // 0/0 = NaN -> `as u8` = 0 in the reference too
    check(ctx_, pt_resolve(ctx_, (uint32_t)samples_per_pixel, out.data()), "pt_resolve");
    return out;
}

namespace {
std::mutex g_mutex;
GpuRenderer* g_renderer = nullptr; // released by render_shutdown(), registered with atexit AFTER the HIP runtime initialised:
                                   // it then runs before the runtime's own teardown (a static destructor might not)
uint64_t g_generation = 0;         // BvhNode::generation of the uploaded world; 0 = none
} // namespace

void render_shutdown() {
    std::lock_guard<std::mutex> lock(g_mutex);
    delete g_renderer;
    g_renderer = nullptr;
    g_generation = 0;
}

std::vector<uint8_t> render(const BvhNode& world, const Camera& camera, size_t width, size_t height, const Color& background,
                            size_t samples_per_pixel, size_t max_bounces, uint64_t seed, PtStats* stats) {
    std::lock_guard<std::mutex> lock(g_mutex);
    if (!g_renderer) {
        g_renderer = new GpuRenderer(0); // (pt_context_create has touched the device: HIP is up)
        std::atexit([] { render_shutdown(); });
    }
    if (g_generation != world.generation) {
        g_generation = 0;
        g_renderer->upload(world);
        g_generation = world.generation;
    }
    return g_renderer->render(camera, width, height, background, samples_per_pixel, max_bounces, seed, stats);
}

void render_invalidate_cache() {
    std::lock_guard<std::mutex> lock(g_mutex);
    g_generation = 0;
}

} // namespace pt
