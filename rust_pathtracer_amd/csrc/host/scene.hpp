// scene.hpp -- host mirror of src/camera.rs and src/scene.rs: Camera, ImageSettings, OutputSettings, Scene and
// the six scenes.  Additions the reference lacks (it hard-codes everything and is unseeded): explicit seeds,
// and ImageSettings overrides so the benchmark configurations (other sizes / spp / depth) are expressible.
// The aspect ratio always follows the image settings (scene.rs:71-78 etc.), so overrides flow into the camera.
#pragma once
#include <optional>
#include <string>

#include "obj_model.hpp"

namespace pt {

struct Camera { // camera.rs:3-11; POD twin of PtCamera
    Vec3 origin, lower_left_corner, horizontal, vertical, u, v;
    double lens_radius = 0.0;
    Camera() {}
    // Camera::new, camera.rs:14-44
    Camera(const Vec3& lookfrom, const Vec3& lookat, const Vec3& up, double vertical_fov, double aspect_ratio,
           double aperture, double focus_dist);
    PtCamera to_pod() const;
};

struct ImageSettings { // scene.rs:19-25
    size_t width = 0, height = 0, samples_per_pixel = 0, max_bounces = 0;
    Color background;
};
struct OutputSettings { // scene.rs:27-36 (enum StaticImage | Animation)
    bool animation = false;
    ImageSettings image_settings;
    double fps = 0.0, duration = 0.0;
};

struct SceneSeeds {
    uint64_t layout = 1; // SphereField's draws, scene.rs:112-131
    uint64_t bvh = 1;    // axis draws, bvh.rs:67 (an ObjModel's nested tree draws first: it is built inside get_world)
    uint64_t perlin = 1; // texture.rs:55
};

struct Scene { // scene.rs:38-42
    SceneSeeds seeds;
    std::optional<ImageSettings> settings_override;
    virtual ~Scene() {}
    virtual std::shared_ptr<BvhNode> get_world() const = 0;
    virtual Camera get_camera_at(double t) const = 0;
    virtual OutputSettings get_output_settings() const = 0;
    double aspect_ratio() const {
        ImageSettings s = get_output_settings().image_settings;
        return (double)s.width / (double)s.height;
    }

protected:
    OutputSettings with_override(OutputSettings o) const {
        if (settings_override) o.image_settings = *settings_override;
        return o;
    }
};

struct SphereFieldScene : Scene { // scene.rs:44-171
    std::shared_ptr<BvhNode> get_world() const override;
    Camera get_camera_at(double t) const override;
    OutputSettings get_output_settings() const override;
};
struct TwoSphereCheckersScene : Scene { // scene.rs:173-236
    std::shared_ptr<BvhNode> get_world() const override;
    Camera get_camera_at(double t) const override;
    OutputSettings get_output_settings() const override;
};
struct LightTestScene : Scene { // scene.rs:238-326
    std::shared_ptr<BvhNode> get_world() const override;
    Camera get_camera_at(double t) const override;
    OutputSettings get_output_settings() const override;
};
struct CornellBoxScene : Scene { // scene.rs:328-465
    std::shared_ptr<BvhNode> get_world() const override;
    Camera get_camera_at(double t) const override;
    OutputSettings get_output_settings() const override;
};
struct TriangleTestScene : Scene { // scene.rs:467-595
    std::shared_ptr<BvhNode> get_world() const override;
    Camera get_camera_at(double t) const override;
    OutputSettings get_output_settings() const override;
};
struct ModelTestScene : Scene { // scene.rs:597-658
    std::string path_str;
    // Set instead of path_str to feed already-parsed / synthetic mesh arrays through the same construction.
    std::shared_ptr<ObjFile> arrays;
    // scene.rs:638-657 places a ground sphere of radius 1000 at (0,-1000,0) and the camera at (0,2.5,-7).
    // world_scale multiplies those constants (config 3 scales the world x100 so that the reference's
    // absolute det < 1e-4 cull, geometry.rs:526, does not swallow a million small triangles).
    double world_scale = 1.0;
    std::shared_ptr<BvhNode> get_world() const override;
    Camera get_camera_at(double t) const override;
    OutputSettings get_output_settings() const override;
};

// By-name factory: "SphereField", "TwoSpheres", "LightTest", "CornellBox", "TriangleTest", "ModelTest".
std::shared_ptr<Scene> make_scene(const std::string& name);

// A closed surface of revolution with exactly 2*n_lon*(n_lat-1) triangles wound counter-clockwise seen from
// outside (det > 0 for rays arriving from outside, geometry.rs:524-528), centred at `center`:
// the synthetic stand-in for model.obj (which the reference does not ship, .gitignore:4).
std::shared_ptr<ObjFile> make_synthetic_mesh(uint32_t n_lon, uint32_t n_lat, double radius, const Vec3& center);

} // namespace pt
