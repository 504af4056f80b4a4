// scene.cpp -- Camera::new (camera.rs:14-44) and the six worlds of scene.rs, object for object.
#include "scene.hpp"

namespace pt {

static const double PI = 3.14159265358979323846; // std::f64::consts::PI

Camera::Camera(const Vec3& lookfrom, const Vec3& lookat, const Vec3& up, double vertical_fov, double aspect_ratio,
               double aperture, double focus_dist) {
    double h = std::tan((vertical_fov * (PI / 180.0)) / 2.0); // f64::to_radians = x * (PI / 180)
    double viewport0 = aspect_ratio * 2.0 * h;
    double viewport1 = 2.0 * h;
    Vec3 w = (lookfrom - lookat).unit_vector();
    u = up.cross(w).unit_vector();
    v = w.cross(u);
    origin = lookfrom;
    horizontal = focus_dist * viewport0 * u;
    vertical = focus_dist * viewport1 * v;
    lower_left_corner = origin - horizontal / 2.0 - vertical / 2.0 - focus_dist * w;
    lens_radius = aperture / 2.0;
}

PtCamera Camera::to_pod() const {
    PtCamera c;
    for (int k = 0; k < 3; ++k) {
        c.origin[k] = origin.e[k];
        c.lower_left_corner[k] = lower_left_corner.e[k];
        c.horizontal[k] = horizontal.e[k];
        c.vertical[k] = vertical.e[k];
        c.u[k] = u.e[k];
        c.v[k] = v.e[k];
    }
    c.lens_radius = lens_radius;
    return c;
}

static ImageSettings image(size_t w, size_t h, size_t spp, size_t depth, const Color& bg) {
    ImageSettings s;
    s.width = w;
    s.height = h;
    s.samples_per_pixel = spp;
    s.max_bounces = depth;
    s.background = bg;
    return s;
}
static OutputSettings still(const ImageSettings& s) {
    OutputSettings o;
    o.animation = false;
    o.image_settings = s;
    return o;
}
static std::shared_ptr<Texture> checker_default() { // scene.rs:94-97, :215-218, :641-644
    return CheckerTexture::new_(SolidColorTexture::new_(Color(0.2, 0.3, 0.1)), SolidColorTexture::new_(Color(0.9, 0.9, 0.9)));
}
static const Vec3 UP(0.0, 1.0, 0.0);

// ---------------------------------------------------------------- SphereField, scene.rs:44-171
OutputSettings SphereFieldScene::get_output_settings() const {
    OutputSettings o;
    o.animation = true;
    o.image_settings = image(854, 480, 250, 20, Color(1.0, 1.0, 1.0));
    o.fps = 30.0;
    o.duration = 10.0;
    return with_override(o);
}
Camera SphereFieldScene::get_camera_at(double t) const { // scene.rs:61-89
    Vec3 lookfrom(12.0 * std::cos(2.0 * PI * t), 1.0 + 2.0 * std::sin(PI * t), 12.0 * std::sin(2.0 * PI * t));
    return Camera(lookfrom, Vec3(0.0, 0.5, 0.0), UP, 20.0, aspect_ratio(), 0.1, 10.0);
}
std::shared_ptr<BvhNode> SphereFieldScene::get_world() const { // scene.rs:91-170
    BuildRng rng(seeds.layout);
    std::vector<HittablePtr> world;
    world.push_back(Sphere::new_(Vec3(0.0, -1000.0, 0.0), 1000.0, LambertianMaterial::new_(checker_default())));
    for (int a = -11; a < 11; ++a) {
        for (int b = -11; b < 11; ++b) {
            if (-1 < b && b < 1 && -6 < a && a < 6) continue;
            double cx = (double)a + 0.5 * rng.f64();
            double cz = (double)b + 0.9 * rng.f64();
            Vec3 center(cx, 0.2, cz);
            double radius = 0.2;
            std::shared_ptr<Material> material;
            bool is_glass = false;
            double pick = rng.f64();
            if (pick < 0.6) {
                double r = rng.f64(), g = rng.f64(), bl = rng.f64(); // Color::random()
                material = LambertianMaterial::new_(SolidColorTexture::new_(Color(r, g, bl)));
            } else if (pick < 0.8) {
                double r = rng.range(0.5, 1.0), g = rng.range(0.5, 1.0), bl = rng.range(0.5, 1.0); // Color::random_range
                double fuzz = rng.f64();
                material = MetalMaterial::new_from_color(Color(r, g, bl), fuzz);
            } else {
                material = DielectricMaterial::new_(1.5);
                is_glass = true;
            }
            if (is_glass && rng.f64() < 0.5) { // hollow glass: a second sphere with radius -0.2 + 0.02
                world.push_back(Sphere::new_(center, radius, material));
                world.push_back(Sphere::new_(center, (radius * -1.0) + 0.02, material));
            } else {
                world.push_back(Sphere::new_(center, radius, material));
            }
        }
    }
    auto big1 = DielectricMaterial::new_(1.5);
    world.push_back(Sphere::new_(Vec3(-4.0, 1.0, 0.0), 1.0, big1));
    world.push_back(Sphere::new_(Vec3(-4.0, 1.0, 0.0), -0.95, big1));
    world.push_back(Sphere::new_(Vec3(4.0, 1.0, 0.0), 1.0, DielectricMaterial::new_(1.5)));
    world.push_back(Sphere::new_(Vec3(0.0, 1.0, 0.0), 1.0, MetalMaterial::new_from_color(Color(0.7, 0.6, 0.5), 0.0)));
    BuildRng bvh_rng(seeds.bvh);
    return BvhNode::new_(world, bvh_rng);
}

// ---------------------------------------------------------------- TwoSpheres, scene.rs:173-236
OutputSettings TwoSphereCheckersScene::get_output_settings() const {
    return with_override(still(image(854, 480, 250, 20, Color(1.0, 1.0, 1.0))));
}
Camera TwoSphereCheckersScene::get_camera_at(double) const {
    return Camera(Vec3(13.0, 2.0, 3.0), Vec3(0.0, 0.0, 0.0), UP, 20.0, aspect_ratio(), 0.0, 10.0);
}
std::shared_ptr<BvhNode> TwoSphereCheckersScene::get_world() const {
    std::vector<HittablePtr> world;
    world.push_back(Sphere::new_(Vec3(0.0, -10.0, 0.0), 10.0, LambertianMaterial::new_(checker_default())));
    world.push_back(Sphere::new_(Vec3(0.0, 10.0, 0.0), 10.0, LambertianMaterial::new_(PerlinNoiseTexture::new_(4.0, seeds.perlin))));
    BuildRng bvh_rng(seeds.bvh);
    return BvhNode::new_(world, bvh_rng);
}

// ---------------------------------------------------------------- LightTest, scene.rs:238-326
OutputSettings LightTestScene::get_output_settings() const {
    return with_override(still(image(854, 480, 2000, 50, Color(0.0, 0.0, 0.0))));
}
Camera LightTestScene::get_camera_at(double) const {
    return Camera(Vec3(26.0, 3.0, 6.0), Vec3(0.0, 2.0, 0.0), UP, 20.0, aspect_ratio(), 0.0, 10.0);
}
std::shared_ptr<BvhNode> LightTestScene::get_world() const {
    std::vector<HittablePtr> world;
    auto ground = LambertianMaterial::new_(PerlinNoiseTexture::new_(4.0, seeds.perlin)); // ONE material for both spheres
    world.push_back(Sphere::new_(Vec3(0.0, -1000.0, 0.0), 1000.0, ground));
    world.push_back(Sphere::new_(Vec3(0.0, 2.0, 0.0), 2.0, ground));
    auto light = DiffuseLightMaterial::new_from_color(Color(4.0, 4.0, 4.0));
    world.push_back(RectangleXY::new_(Vec3(3.0, 1.0, -2.0), Vec3(5.0, 3.0, -2.0), 1.0, light));
    world.push_back(RectangleXZ::new_(Vec3(-1.0, 6.0, -1.0), Vec3(1.0, 6.0, 1.0), -1.0, light));
    world.push_back(RectangleYZ::new_(Vec3(-6.0, 1.0, -2.0), Vec3(-6.0, 3.0, 2.0), 1.0, light));
    BuildRng bvh_rng(seeds.bvh);
    return BvhNode::new_(world, bvh_rng);
}

// ---------------------------------------------------------------- CornellBox / TriangleTest, scene.rs:328-595
static void cornell_shell(std::vector<HittablePtr>& world, std::shared_ptr<Material>& white) {
    auto red = LambertianMaterial::new_from_color(Color(0.65, 0.05, 0.05));
    white = LambertianMaterial::new_from_color(Color(0.73, 0.73, 0.73));
    auto green = LambertianMaterial::new_from_color(Color(0.12, 0.45, 0.15));
    auto light = DiffuseLightMaterial::new_from_color(Color(15.0, 15.0, 15.0));
    world.push_back(RectangleYZ::new_(Vec3(555.0, 0.0, 0.0), Vec3(555.0, 555.0, 555.0), -1.0, green));
    world.push_back(RectangleYZ::new_(Vec3(0.0, 0.0, 0.0), Vec3(0.0, 555.0, 555.0), 1.0, red));
    world.push_back(RectangleXZ::new_(Vec3(0.0, 555.0, 0.0), Vec3(555.0, 555.0, 555.0), -1.0, white));
    world.push_back(RectangleXZ::new_(Vec3(0.0, 0.0, 0.0), Vec3(555.0, 0.0, 555.0), 1.0, white));
    world.push_back(RectangleXZ::new_(Vec3(213.0, 554.0, 227.0), Vec3(343.0, 554.0, 332.0), -1.0, light));
    world.push_back(RectangleXY::new_(Vec3(0.0, 0.0, 555.0), Vec3(555.0, 555.0, 555.0), -1.0, white));
}
static Camera cornell_camera(double aspect) {
    return Camera(Vec3(278.0, 278.0, -800.0), Vec3(278.0, 278.0, 0.0), UP, 40.0, aspect, 0.0, 10.0);
}
OutputSettings CornellBoxScene::get_output_settings() const {
    return with_override(still(image(400, 400, 1000, 20, Color(0.0, 0.0, 0.0))));
}
Camera CornellBoxScene::get_camera_at(double) const { return cornell_camera(aspect_ratio()); }
std::shared_ptr<BvhNode> CornellBoxScene::get_world() const {
    std::vector<HittablePtr> world;
    std::shared_ptr<Material> white;
    cornell_shell(world, white);
    auto glass = DielectricMaterial::new_(1.5);
    world.push_back(AABox::new_(Vec3(130.0, 0.0, 65.0), Vec3(295.0, 165.0, 230.0), white));
    world.push_back(AABox::new_(Vec3(265.0, 0.0, 295.0), Vec3(430.0, 330.0, 460.0), white));
    world.push_back(Sphere::new_(Vec3(212.5, 255.0, 147.5), 90.0, glass));
    world.push_back(Sphere::new_(Vec3(347.5, 420.0, 377.5), 90.0, glass));
    BuildRng bvh_rng(seeds.bvh);
    return BvhNode::new_(world, bvh_rng);
}
OutputSettings TriangleTestScene::get_output_settings() const {
    return with_override(still(image(400, 400, 1000, 20, Color(0.0, 0.0, 0.0))));
}
Camera TriangleTestScene::get_camera_at(double) const { return cornell_camera(aspect_ratio()); }
std::shared_ptr<BvhNode> TriangleTestScene::get_world() const {
    std::vector<HittablePtr> world;
    std::shared_ptr<Material> white;
    cornell_shell(world, white);
    auto glass = DielectricMaterial::new_(1.5);
    world.push_back(Triangle::new_without_normal(Vec3(200.0, 100.0, 100.0), Vec3(300.0, 300.0, 500.0), Vec3(400.0, 100.0, 100.0), glass));
    world.push_back(Triangle::new_without_normal(Vec3(100.0, 300.0, 100.0), Vec3(150.0, 400.0, 250.0), Vec3(100.0, 300.0, 400.0), white));
    BuildRng bvh_rng(seeds.bvh);
    return BvhNode::new_(world, bvh_rng);
}

// ---------------------------------------------------------------- ModelTest, scene.rs:597-658
OutputSettings ModelTestScene::get_output_settings() const {
    return with_override(still(image(800, 800, 250, 20, Color(1.0, 1.0, 1.0))));
}
Camera ModelTestScene::get_camera_at(double) const {
    const double s = world_scale;
    return Camera(Vec3(0.0, 2.5 * s, -7.0 * s), Vec3(0.0, 1.5 * s, 0.0), UP, 60.0, aspect_ratio(), 0.0, 10.0);
}
std::shared_ptr<BvhNode> ModelTestScene::get_world() const {
    const double s = world_scale;
    std::vector<HittablePtr> world;
    BuildRng bvh_rng(seeds.bvh);
    world.push_back(Sphere::new_(Vec3(0.0, -1000.0 * s, 0.0), 1000.0 * s, LambertianMaterial::new_(checker_default())));
    if (arrays)
        world.push_back(ObjModel::new_from_arrays(*arrays, bvh_rng));
    else
        world.push_back(ObjModel::new_from_path(path_str, bvh_rng));
    return BvhNode::new_(world, bvh_rng);
}

std::shared_ptr<Scene> make_scene(const std::string& name) {
    if (name == "SphereField") return std::make_shared<SphereFieldScene>();
    if (name == "TwoSpheres") return std::make_shared<TwoSphereCheckersScene>();
    if (name == "LightTest") return std::make_shared<LightTestScene>();
    if (name == "CornellBox") return std::make_shared<CornellBoxScene>();
    if (name == "TriangleTest") return std::make_shared<TriangleTestScene>();
    if (name == "ModelTest") return std::make_shared<ModelTestScene>();
    return nullptr;
}

std::shared_ptr<ObjFile> make_synthetic_mesh(uint32_t n_lon, uint32_t n_lat, double radius, const Vec3& center) {
    if (n_lon < 3 || n_lat < 2) throw std::runtime_error("make_synthetic_mesh: need n_lon >= 3 and n_lat >= 2");
    auto file = std::make_shared<ObjFile>();
    file->models.emplace_back();
    ObjMesh& m = file->models.back();
    // vertex 0 = top pole, then (n_lat-1) rings of n_lon vertices, last = bottom pole
    auto radial = [&](double theta, double phi) { // a lumpy profile so the surface is not a plain sphere
        return radius * (1.0 + 0.18 * std::sin(3.0 * theta) * std::cos(5.0 * phi) + 0.07 * std::cos(9.0 * theta));
    };
    auto push = [&](double theta, double phi) {
        double r = radial(theta, phi);
        m.positions.push_back((float)(center.x() + r * std::sin(theta) * std::cos(phi)));
        m.positions.push_back((float)(center.y() + r * std::cos(theta)));
        m.positions.push_back((float)(center.z() + r * std::sin(theta) * std::sin(phi)));
    };
    push(0.0, 0.0);
    for (uint32_t i = 1; i < n_lat; ++i)
        for (uint32_t j = 0; j < n_lon; ++j) push(PI * (double)i / (double)n_lat, 2.0 * PI * (double)j / (double)n_lon);
    push(PI, 0.0);
    const uint32_t bottom = 1 + (n_lat - 1) * n_lon;
    auto ring = [&](uint32_t i, uint32_t j) { return 1 + (i - 1) * n_lon + (j % n_lon); };
    auto tri = [&](uint32_t a, uint32_t b, uint32_t c) {
        m.indices.push_back(a);
        m.indices.push_back(b);
        m.indices.push_back(c);
    };
    // With x = sin(theta)cos(phi), y = cos(theta), z = sin(theta)sin(phi), d/dtheta x d/dphi points inward, so
    // the outward (counter-clockwise from outside) order of a quad is (i,j) -> (i,j+1) -> (i+1,j+1) -> (i+1,j).
    for (uint32_t j = 0; j < n_lon; ++j) tri(0, ring(1, j + 1), ring(1, j));
    for (uint32_t i = 1; i + 1 < n_lat; ++i)
        for (uint32_t j = 0; j < n_lon; ++j) {
            tri(ring(i, j), ring(i, j + 1), ring(i + 1, j + 1));
            tri(ring(i, j), ring(i + 1, j + 1), ring(i + 1, j));
        }
    for (uint32_t j = 0; j < n_lon; ++j) tri(bottom, ring(n_lat - 1, j), ring(n_lat - 1, j + 1));
    return file;
}

} // namespace pt
