// renderer.hpp -- host mirror of src/renderer.rs: render(world, camera, width, height, background,
// samples_per_pixel, max_bounces) -> Vec<u8>, executed on the GPU through the C ABI of include/ptgpu.h.
// There is no CPU fallback: every failure (no GPU, HIP error) throws std::runtime_error, as the reference
// panics on its own failures.
#pragma once
#include "scene.hpp"

namespace pt {

// One GPU's renderer: owns a PtContext and remembers which world it holds.
class GpuRenderer {
public:
    explicit GpuRenderer(int device = 0);
    ~GpuRenderer();
    GpuRenderer(const GpuRenderer&) = delete;
    GpuRenderer& operator=(const GpuRenderer&) = delete;

    // Flattens and uploads `world` (replaces the trait-object tree the reference walks on the CPU).
    void upload(const BvhNode& world);
    // renderer.rs:5-37.  `seed` keys the counter-based sampler (the reference is unseeded).
    std::vector<uint8_t> render(const Camera& camera, size_t width, size_t height, const Color& background,
                                size_t samples_per_pixel, size_t max_bounces, uint64_t seed = 0, PtStats* stats = nullptr);
    PtContext* context() { return ctx_; }

private:
    PtContext* ctx_ = nullptr;
};

// Drop-in for renderer::render: uploads `world` to the process-wide renderer on device 0 when it is not the
// world rendered last (the reference renders many frames from one world, main.rs:23,51-64), then renders.
// "The world rendered last" is decided by BvhNode::generation (assigned at construction), not by address.
std::vector<uint8_t> render(const BvhNode& world, const Camera& camera, size_t width, size_t height, const Color& background,
                            size_t samples_per_pixel, size_t max_bounces, uint64_t seed = 0, PtStats* stats = nullptr);
// Forget the cached world of the process-wide renderer (call after mutating a world's nodes in place).
void render_invalidate_cache();
// Destroys the process-wide renderer and its GPU context (also registered with atexit by its first use).
void render_shutdown();

} // namespace pt
