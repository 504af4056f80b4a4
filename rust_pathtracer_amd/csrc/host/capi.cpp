// capi.cpp -- the C view of the host layer declared in include/pthost.h.
#include "pthost.h"

#include <chrono>
#include <map>
#include <string>
#include <thread>

#include "output.hpp"
#include "renderer.hpp"

using namespace pt;

namespace {
thread_local std::string g_err;
PthProgressFn g_progress = nullptr; // pth_set_progress_callback
void* g_progress_user = nullptr;

struct CustomScene : Scene { // a world built from an object list; no camera / settings of its own
    std::shared_ptr<BvhNode> world;
    std::shared_ptr<BvhNode> get_world() const override { return world; }
    Camera get_camera_at(double) const override { throw std::runtime_error("a custom world has no camera"); }
    OutputSettings get_output_settings() const override { throw std::runtime_error("a custom world has no output settings"); }
};
} // namespace

struct PthScene {
    std::shared_ptr<Scene> scene;
    std::shared_ptr<BvhNode> world;
    Flattener flat;
    PtSceneDesc desc{};
    bool built = false;
    std::map<int, std::unique_ptr<GpuRenderer>> renderers; // per device
    std::map<int, bool> uploaded;
    std::shared_ptr<ObjFile> mesh; // ModelTest: the arrays the world was built from
    bool rebuild_per_frame = false;
    double seconds_build = 0.0, seconds_upload = 0.0; // of the last pth_render_frames call
};

template <class F>
static int guarded(F&& f) {
    try {
        f();
        return 0;
    } catch (const std::exception& e) {
        g_err = e.what();
        return -1;
    }
}

extern "C" {

const char* pth_last_error(void) { return g_err.c_str(); }

PthScene* pth_scene_create(const char* name) {
    if (!name) {
        g_err = "pth_scene_create: null name";
        return nullptr;
    }
    auto sc = make_scene(name);
    if (!sc) {
        g_err = std::string("unknown scene ") + name;
        return nullptr;
    }
    PthScene* s = new PthScene();
    s->scene = sc;
    return s;
}

void pth_scene_destroy(PthScene* s) { delete s; }

void pth_scene_set_seeds(PthScene* s, uint64_t layout, uint64_t bvh, uint64_t perlin) {
    if (!s) return;
    s->scene->seeds.layout = layout;
    s->scene->seeds.bvh = bvh;
    s->scene->seeds.perlin = perlin;
    s->built = false;
}

void pth_scene_set_image(PthScene* s, uint32_t width, uint32_t height, uint32_t spp, uint32_t max_bounces, const double background[3]) {
    if (!s || !background) return;
    ImageSettings im;
    im.width = width;
    im.height = height;
    im.samples_per_pixel = spp;
    im.max_bounces = max_bounces;
    im.background = Color(background[0], background[1], background[2]);
    s->scene->settings_override = im;
}

int pth_scene_set_model_path(PthScene* s, const char* path) {
    return guarded([&] {
        auto* m = s ? dynamic_cast<ModelTestScene*>(s->scene.get()) : nullptr;
        if (!m || !path) throw std::runtime_error("pth_scene_set_model_path: not a ModelTest scene");
        m->path_str = path;
        m->arrays = std::make_shared<ObjFile>(load_obj(path)); // parsed once; also what pth_scene_mesh hands out
        s->mesh = m->arrays;
        s->built = false;
    });
}

int pth_scene_set_model_synthetic(PthScene* s, uint32_t n_lon, uint32_t n_lat, double radius, const double center[3], double world_scale) {
    return guarded([&] {
        auto* m = s ? dynamic_cast<ModelTestScene*>(s->scene.get()) : nullptr;
        if (!m || !center) throw std::runtime_error("pth_scene_set_model_synthetic: not a ModelTest scene");
        m->arrays = make_synthetic_mesh(n_lon, n_lat, radius, Vec3(center[0], center[1], center[2]));
        m->world_scale = world_scale;
        s->mesh = m->arrays;
        s->built = false;
    });
}

int pth_scene_build(PthScene* s) {
    return guarded([&] {
        if (!s) throw std::runtime_error("pth_scene_build: null scene");
        s->world = s->scene->get_world();
        s->flat = Flattener();
        s->flat.flatten_world(*s->world);
        s->desc = s->flat.desc();
        s->built = true;
        s->uploaded.clear();
    });
}

const PtSceneDesc* pth_scene_desc(PthScene* s) {
    if (!s || !s->built) {
        g_err = "pth_scene_desc: scene not built";
        return nullptr;
    }
    return &s->desc;
}

int pth_scene_camera(PthScene* s, double t, PtCamera* out) {
    return guarded([&] {
        if (!s || !out) throw std::runtime_error("pth_scene_camera: null argument");
        *out = s->scene->get_camera_at(t).to_pod();
    });
}

int pth_scene_settings(PthScene* s, uint32_t* width, uint32_t* height, uint32_t* spp, uint32_t* max_bounces, double background[3],
                       int* animation, double* fps, double* duration) {
    return guarded([&] {
        if (!s) throw std::runtime_error("pth_scene_settings: null scene");
        OutputSettings o = s->scene->get_output_settings();
        if (width) *width = (uint32_t)o.image_settings.width;
        if (height) *height = (uint32_t)o.image_settings.height;
        if (spp) *spp = (uint32_t)o.image_settings.samples_per_pixel;
        if (max_bounces) *max_bounces = (uint32_t)o.image_settings.max_bounces;
        if (background)
            for (int k = 0; k < 3; ++k) background[k] = o.image_settings.background.e[k];
        if (animation) *animation = o.animation ? 1 : 0;
        if (fps) *fps = o.fps;
        if (duration) *duration = o.duration;
    });
}

int pth_scene_mesh_count(PthScene* s) { return (s && s->mesh) ? (int)s->mesh->models.size() : 0; }
int pth_scene_mesh(PthScene* s, int i, PthMeshDesc* out) {
    if (!s || !s->mesh || !out || i < 0 || i >= (int)s->mesh->models.size()) {
        g_err = "pth_scene_mesh: bad argument";
        return -1;
    }
    const ObjMesh& m = s->mesh->models[i];
    out->positions = m.positions.data();
    out->n_positions = m.positions.size();
    out->indices = m.indices.data();
    out->n_indices = m.indices.size();
    out->normals = m.normals.empty() ? nullptr : m.normals.data();
    out->n_normals = m.normals.size();
    out->material_id = m.material_id;
    out->pad = 0;
    return 0;
}
int pth_scene_mtl_count(PthScene* s) { return (s && s->mesh) ? (int)s->mesh->materials.size() : 0; }
int pth_scene_mtl(PthScene* s, int i, PthMtlDesc* out) {
    if (!s || !s->mesh || !out || i < 0 || i >= (int)s->mesh->materials.size()) {
        g_err = "pth_scene_mtl: bad argument";
        return -1;
    }
    const ObjMaterial& m = s->mesh->materials[i];
    out->illum = m.illumination_model;
    out->ni = m.optical_density;
    for (int k = 0; k < 3; ++k) out->kd[k] = m.diffuse[k];
    out->ns = m.shininess;
    return 0;
}

PthScene* pth_world_create(const PthObjectDesc* objects, uint32_t n_objects, const PtMaterial* materials, uint32_t n_materials,
                           const PthTextureDesc* textures, uint32_t n_textures, const PthMeshDesc* meshes, uint32_t n_meshes,
                           const PthMtlDesc* mtls, uint32_t n_mtls, uint64_t bvh_seed) {
    PthScene* s = new PthScene();
    int rc = guarded([&] {
        // textures: leaves first, then checkers whose children exist, until none is left
        std::vector<std::shared_ptr<Texture>> texs(n_textures);
        uint32_t remaining = n_textures;
        for (uint32_t pass = 0; pass <= n_textures && remaining > 0; ++pass) {
            for (uint32_t i = 0; i < n_textures; ++i) {
                if (texs[i]) continue;
                const PthTextureDesc& t = textures[i];
                if (t.type == PT_TEX_SOLID) {
                    texs[i] = SolidColorTexture::new_(Color(t.color[0], t.color[1], t.color[2]));
                } else if (t.type == PT_TEX_PERLIN) {
                    texs[i] = PerlinNoiseTexture::new_(t.scale, t.perlin_seed);
                } else if (t.type == PT_TEX_CHECKER) {
                    if (t.odd >= n_textures || t.even >= n_textures) throw std::runtime_error("checker child out of range");
                    if (!texs[t.odd] || !texs[t.even]) continue;
                    texs[i] = CheckerTexture::new_(texs[t.odd], texs[t.even]);
                } else {
                    throw std::runtime_error("bad texture type");
                }
                --remaining;
            }
        }
        if (remaining > 0) throw std::runtime_error("texture graph has a cycle");
        std::vector<std::shared_ptr<Material>> mats;
        for (uint32_t i = 0; i < n_materials; ++i) {
            const PtMaterial& m = materials[i];
            if (m.type != PT_MAT_DIELECTRIC && m.texture >= n_textures) throw std::runtime_error("material texture out of range");
            if (m.type == PT_MAT_LAMBERTIAN) mats.push_back(LambertianMaterial::new_(texs[m.texture]));
            else if (m.type == PT_MAT_METAL) mats.push_back(MetalMaterial::new_(texs[m.texture], m.param));
            else if (m.type == PT_MAT_DIELECTRIC) mats.push_back(DielectricMaterial::new_(m.param));
            else if (m.type == PT_MAT_DIFFUSE_LIGHT) mats.push_back(std::make_shared<DiffuseLightMaterial>(texs[m.texture]));
            else throw std::runtime_error("bad material type");
        }
        auto file = std::make_shared<ObjFile>();
        for (uint32_t i = 0; i < n_meshes; ++i) {
            ObjMesh m;
            m.positions.assign(meshes[i].positions, meshes[i].positions + meshes[i].n_positions);
            m.indices.assign(meshes[i].indices, meshes[i].indices + meshes[i].n_indices);
            if (meshes[i].n_normals) m.normals.assign(meshes[i].normals, meshes[i].normals + meshes[i].n_normals);
            m.material_id = meshes[i].material_id;
            file->models.push_back(std::move(m));
        }
        for (uint32_t i = 0; i < n_mtls; ++i) {
            ObjMaterial m;
            m.illumination_model = mtls[i].illum;
            m.optical_density = mtls[i].ni;
            for (int k = 0; k < 3; ++k) m.diffuse[k] = mtls[i].kd[k];
            m.shininess = mtls[i].ns;
            file->materials.push_back(m);
        }
        BuildRng rng(bvh_seed);
        std::vector<HittablePtr> world;
        for (uint32_t i = 0; i < n_objects; ++i) {
            const PthObjectDesc& o = objects[i];
            const double* p = o.p;
            if (o.kind == 7) {
                world.push_back(ObjModel::new_from_arrays(*file, rng));
                continue;
            }
            if (o.material >= mats.size()) throw std::runtime_error("object material out of range");
            auto m = mats[o.material];
            Vec3 a(p[0], p[1], p[2]), b(p[3], p[4], p[5]);
            switch (o.kind) {
                case 0: world.push_back(Sphere::new_(a, p[3], m)); break;
                case 1: world.push_back(RectangleXY::new_(a, b, p[6], m)); break;
                case 2: world.push_back(RectangleXZ::new_(a, b, p[6], m)); break;
                case 3: world.push_back(RectangleYZ::new_(a, b, p[6], m)); break;
                case 4: world.push_back(AABox::new_(a, b, m)); break;
                case 5: world.push_back(Triangle::new_(a, b, Vec3(p[6], p[7], p[8]), Vec3(p[9], p[10], p[11]), m)); break;
                case 6: world.push_back(Triangle::new_without_normal(a, b, Vec3(p[6], p[7], p[8]), m)); break;
                default: throw std::runtime_error("bad object kind");
            }
        }
        auto cs = std::make_shared<CustomScene>();
        cs->world = BvhNode::new_(world, rng);
        s->scene = cs;
        s->mesh = file;
        s->world = cs->world;
        s->flat.flatten_world(*s->world);
        s->desc = s->flat.desc();
        s->built = true;
    });
    if (rc != 0) {
        delete s;
        return nullptr;
    }
    return s;
}

void pth_camera_new(const double lookfrom[3], const double lookat[3], const double up[3], double vertical_fov, double aspect_ratio,
                    double aperture, double focus_dist, PtCamera* out) {
    if (!out) return;
    *out = Camera(Vec3(lookfrom[0], lookfrom[1], lookfrom[2]), Vec3(lookat[0], lookat[1], lookat[2]), Vec3(up[0], up[1], up[2]),
                  vertical_fov, aspect_ratio, aperture, focus_dist)
               .to_pod();
}

int pth_render(PthScene* s, int device, double t, uint64_t seed, uint8_t* rgb_out, PtStats* stats) {
    return guarded([&] {
        if (!s || !rgb_out) throw std::runtime_error("pth_render: null argument");
        if (!s->built) throw std::runtime_error("pth_render: scene not built");
        auto& r = s->renderers[device];
        if (!r) r.reset(new GpuRenderer(device));
        if (!s->uploaded[device]) {
            r->upload(*s->world);
            s->uploaded[device] = true;
        }
        OutputSettings o = s->scene->get_output_settings();
        const ImageSettings& im = o.image_settings;
        std::vector<uint8_t> px = r->render(s->scene->get_camera_at(t), im.width, im.height, im.background, im.samples_per_pixel,
                                            im.max_bounces, seed, stats);
        std::memcpy(rgb_out, px.data(), px.size());
    });
}

int pth_write_png(const char* path, uint32_t width, uint32_t height, const uint8_t* rgb) {
    return guarded([&] {
        if (!path || !rgb) throw std::runtime_error("pth_write_png: null argument");
        write_file(path, width, height, rgb, (size_t)width * height * 3);
    });
}

// main.rs:24-72: the frame loop.  Frame i of N is rendered at t = i/N and written to out_dir/image_{i:04}.png; the PNG of
// frame i is encoded and written by a second thread while frame i + stride renders.
int pth_render_frames(PthScene* s, int device, uint64_t seed, const char* out_dir, uint32_t first_frame, uint32_t frame_stride,
                      uint32_t max_frames, uint32_t* frames_total, uint32_t* frames_written, double* seconds_render, double* seconds_write,
                      uint64_t* rays) {
    return guarded([&] {
        if (!s || !out_dir) throw std::runtime_error("pth_render_frames: null argument");
        if (!s->built) throw std::runtime_error("pth_render_frames: scene not built");
        if (frame_stride == 0) throw std::runtime_error("pth_render_frames: frame_stride must be at least 1");
        auto& r = s->renderers[device];
        if (!r) r.reset(new GpuRenderer(device));
        if (!s->uploaded[device]) {
            r->upload(*s->world);
            s->uploaded[device] = true;
        }
        const OutputSettings o = s->scene->get_output_settings();
        const ImageSettings& im = o.image_settings;
        const uint64_t amount_of_frames = o.animation ? (uint64_t)(o.fps * o.duration) : 1;  // main.rs:25-32
        if (frames_total) *frames_total = (uint32_t)amount_of_frames;
        s->seconds_build = s->seconds_upload = 0.0;
        if (s->rebuild_per_frame && !s->scene) throw std::runtime_error("pth_render_frames: a world made by pth_world_create cannot be rebuilt");
        double t_render = 0.0, t_write = 0.0;
        uint32_t written = 0;
        uint64_t ray_sum = 0;
        auto now = [] { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
        std::string write_error;
        std::thread writer;
        // Joined on EVERY way out of this scope: an exception from the next frame's render or upload must not unwind past a
        // joinable std::thread (that is std::terminate, and the host process with it).
        struct Joiner {
            std::thread& t;
            ~Joiner() {
                if (t.joinable()) t.join();
            }
        } joiner{writer};
        for (uint64_t frame_index = first_frame; frame_index < amount_of_frames && written < (max_frames ? max_frames : 0xFFFFFFFFu);
             frame_index += frame_stride) {
            const double t = (double)frame_index / (double)amount_of_frames;  // main.rs:52
            PtStats st;
            if (s->rebuild_per_frame) { // BvhNode::new and the upload again for this frame: same seeds, so the same world and image
                const double b0 = now();
                s->world = s->scene->get_world();
                const double b1 = now();
                r->upload(*s->world); // flattening, record stream, guide tree, copies
                s->seconds_build += b1 - b0;
                s->seconds_upload += now() - b1;
            }
            const double t0 = now();
            std::vector<uint8_t> pixels = r->render(s->scene->get_camera_at(t), im.width, im.height, im.background,
                                                    im.samples_per_pixel, im.max_bounces, seed, &st);
            const double frame_seconds = now() - t0;
            t_render += frame_seconds;
            ray_sum += st.rays;
            if (g_progress) {
                const uint64_t remaining = (amount_of_frames - 1 - frame_index) / frame_stride; // frames after this one
                uint64_t to_do = (uint64_t)written + 1 + remaining;
                if (max_frames && to_do > max_frames) to_do = max_frames;
                g_progress(written + 1, (uint32_t)to_do, (uint32_t)frame_index, frame_seconds, g_progress_user);
            }
            if (writer.joinable()) writer.join();
            if (!write_error.empty()) throw std::runtime_error(write_error);
            const std::string path = frame_path(out_dir, (size_t)frame_index);
            writer = std::thread([&write_error, &t_write, now, path, w = im.width, h = im.height, px = std::move(pixels)] {
                const double w0 = now();
                try {
                    write_file(path, w, h, px);
                } catch (const std::exception& e) {
                    write_error = e.what();
                }
                t_write += now() - w0;
            });
            ++written;
        }
        if (writer.joinable()) writer.join();
        if (!write_error.empty()) throw std::runtime_error(write_error);
        if (frames_written) *frames_written = written;
        if (seconds_render) *seconds_render = t_render;
        if (seconds_write) *seconds_write = t_write;
        if (rays) *rays = ray_sum;
    });
}

void pth_set_progress_callback(PthProgressFn callback, void* user) {
    g_progress = callback;
    g_progress_user = user;
}

int pth_scene_set_rebuild_per_frame(PthScene* s, int on) {
    return guarded([&] {
        if (!s) throw std::runtime_error("pth_scene_set_rebuild_per_frame: null scene");
        s->rebuild_per_frame = on != 0;
    });
}

int pth_scene_rebuild_seconds(PthScene* s, double* seconds_build, double* seconds_upload) {
    return guarded([&] {
        if (!s) throw std::runtime_error("pth_scene_rebuild_seconds: null scene");
        if (seconds_build) *seconds_build = s->seconds_build;
        if (seconds_upload) *seconds_upload = s->seconds_upload;
    });
}

int pth_obj_probe(const char* path, uint32_t* n_models, uint64_t* n_triangles, uint32_t* n_materials) {
    return guarded([&] {
        if (!path) throw std::runtime_error("pth_obj_probe: null path");
        ObjFile f = load_obj(path);
        uint64_t tris = 0;
        for (const auto& m : f.models) tris += m.indices.size() / 3;
        if (n_models) *n_models = (uint32_t)f.models.size();
        if (n_triangles) *n_triangles = tris;
        if (n_materials) *n_materials = (uint32_t)f.materials.size();
    });
}

} // extern "C"
