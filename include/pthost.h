/* pthost.h -- C view of the C++ host layer (libpthost.so), for language bindings and the Python tests.
 *
 * The host layer (rust_pathtracer_amd/csrc/host, namespace pt) mirrors the reference's public surface name
 * for name: Vec3/Color (vec3.rs), Camera (camera.rs:14-56), Sphere / RectangleXY/XZ/YZ / AABox / Triangle
 * (geometry.rs), ObjModel (obj_model.rs), BvhNode (bvh.rs:65-103), the four materials (material.rs), the
 * three textures (texture.rs), ImageSettings / OutputSettings / Scene and the six scenes (scene.rs), and
 * render() (renderer.rs:5-37).  It builds worlds, flattens them into the POD tables of ptgpu.h and drives
 * the device library.  The functions below expose that to C callers; they add nothing of their own.
 *
 * Return 0 = success, -1 = failure with the message in pth_last_error() (thread-local).
 */
#ifndef PTHOST_H
#define PTHOST_H

#include "ptgpu.h"

#ifdef __cplusplus
extern "C" {
#endif

typedef struct PthScene PthScene;

const char* pth_last_error(void);

/* ---- the six scenes of scene.rs: "SphereField" (:44-171), "TwoSpheres" (:173-236), "LightTest" (:238-326),
 * "CornellBox" (:328-465), "TriangleTest" (:467-595), "ModelTest" (:597-658).  NULL for an unknown name. */
PthScene* pth_scene_create(const char* name);
void pth_scene_destroy(PthScene* s);
/* The reference is unseeded (thread_rng); here the layout (scene.rs:112-131), BVH axis (bvh.rs:67) and
 * Perlin (texture.rs:55) draws are seeded so that a world can be rebuilt identically anywhere. */
void pth_scene_set_seeds(PthScene* s, uint64_t layout, uint64_t bvh, uint64_t perlin);
/* Overrides get_output_settings()'s ImageSettings (scene.rs:19-25); the camera's aspect ratio follows. */
void pth_scene_set_image(PthScene* s, uint32_t width, uint32_t height, uint32_t spp, uint32_t max_bounces, const double background[3]);
/* ModelTest only: the OBJ path (scene.rs:597-599, main.rs:20-22) ... */
int pth_scene_set_model_path(PthScene* s, const char* path);
/* ... or a synthetic closed mesh of 2*n_lon*(n_lat-1) triangles centred at `center`, with the ground sphere
 * and camera of scene.rs:614-657 multiplied by world_scale. */
int pth_scene_set_model_synthetic(PthScene* s, uint32_t n_lon, uint32_t n_lat, double radius, const double center[3], double world_scale);
/* scene.get_world() (scene.rs:39) followed by flattening.  Must precede desc / render. */
int pth_scene_build(PthScene* s);
/* The flattened world; valid until the scene is rebuilt or destroyed. */
const PtSceneDesc* pth_scene_desc(PthScene* s);
/* scene.get_camera_at(t) (scene.rs:40). */
int pth_scene_camera(PthScene* s, double t, PtCamera* out);
/* scene.get_output_settings() (scene.rs:41). */
int pth_scene_settings(PthScene* s, uint32_t* width, uint32_t* height, uint32_t* spp, uint32_t* max_bounces, double background[3],
                       int* animation, double* fps, double* duration);

/* tobj-shaped arrays of a ModelTest scene's mesh (after pth_scene_build), so a checker can be fed the same data. */
typedef struct PthMeshDesc {
    const float* positions;
    uint64_t n_positions; /* floats */
    const uint32_t* indices;
    uint64_t n_indices;
    const float* normals;
    uint64_t n_normals; /* floats, 0 = none */
    int32_t material_id; /* -1 = none */
    int32_t pad;
} PthMeshDesc;
typedef struct PthMtlDesc {
    int32_t illum; /* -1 = absent */
    float ni;
    float kd[3];
    float ns;
} PthMtlDesc;
int pth_scene_mesh_count(PthScene* s);
int pth_scene_mesh(PthScene* s, int i, PthMeshDesc* out);
int pth_scene_mtl_count(PthScene* s);
int pth_scene_mtl(PthScene* s, int i, PthMtlDesc* out);

/* ---- a world given as an object list: the reference's Vec<Box<dyn Hittable>> handed to BvhNode::new. */
typedef struct PthTextureDesc {
    uint32_t type;      /* PT_TEX_* */
    uint32_t odd, even; /* checker children */
    uint32_t pad;
    double color[3];
    double scale;
    uint64_t perlin_seed;
} PthTextureDesc;
typedef struct PthObjectDesc {
    uint32_t kind;     /* 0 Sphere(center p[0..2], radius p[3])
                          1 RectangleXY / 2 RectangleXZ / 3 RectangleYZ (start p[0..2], end p[3..5], direction p[6])
                          4 AABox(start p[0..2], end p[3..5])
                          5 Triangle::new(p1, p2, p3, normal)   6 Triangle::new_without_normal(p1, p2, p3)
                          7 ObjModel built from the mesh / mtl arrays passed alongside */
    uint32_t material; /* index into the material list (ignored for kind 7) */
    double p[12];
} PthObjectDesc;
/* Errors mirror the reference: a rectangle whose start/end differ on its axis -> "rectangle axis is not
 * aligned" (geometry.rs:173-206); fewer than two objects -> "creating BVH node from empty object list" (bvh.rs:76). */
PthScene* pth_world_create(const PthObjectDesc* objects, uint32_t n_objects, const PtMaterial* materials, uint32_t n_materials,
                           const PthTextureDesc* textures, uint32_t n_textures, const PthMeshDesc* meshes, uint32_t n_meshes,
                           const PthMtlDesc* mtls, uint32_t n_mtls, uint64_t bvh_seed);

/* Camera::new (camera.rs:14-44). */
void pth_camera_new(const double lookfrom[3], const double lookat[3], const double up[3], double vertical_fov, double aspect_ratio,
                    double aperture, double focus_dist, PtCamera* out);

/* renderer::render (renderer.rs:5-37) of the built scene with its own settings and camera at time t, on
 * GPU `device`: rgb_out = width*height*3 bytes, top row first.  stats may be NULL. */
int pth_render(PthScene* s, int device, double t, uint64_t seed, uint8_t* rgb_out, PtStats* stats);

/* write_file (main.rs:78-91): an 8-bit RGB PNG, rows top first. */
int pth_write_png(const char* path, uint32_t width, uint32_t height, const uint8_t* rgb);
/* The frame loop of main.rs:24-72 for frames first_frame, first_frame + frame_stride, ... (< the scene's frame count =
 * fps * duration, or 1 for a static image; at most max_frames of them when max_frames > 0): frame i is rendered at t = i / count
 * with the scene's ImageSettings and written to out_dir/image_{i:04}.png (the directory must exist), the PNG being encoded
 * while the next frame renders.  A rank of an N-GPU job passes first_frame = rank, frame_stride = N.  Outputs may be NULL. */
int pth_render_frames(PthScene* s, int device, uint64_t seed, const char* out_dir, uint32_t first_frame, uint32_t frame_stride,
                      uint32_t max_frames, uint32_t* frames_total, uint32_t* frames_written, double* seconds_render, double* seconds_write,
                      uint64_t* rays);
/* Progress reporting of pth_render_frames (the reference draws an indicatif bar per frame, main.rs:42-49,73-75): `callback`
 * is called from the calling thread after every rendered frame with the number of frames done so far, the number this call
 * will render, the frame index just finished and the seconds its render took.  NULL (the default) = no reporting.
 * Process-wide; set it before the call. */
typedef void (*PthProgressFn)(uint32_t frames_done, uint32_t frames_to_do, uint32_t frame_index, double frame_seconds, void* user);
void pth_set_progress_callback(PthProgressFn callback, void* user);
/* BASELINE.json configs[4] asks for the world to be rebuilt and uploaded for every frame (the reference builds it once,
 * main.rs:23).  on = 1: pth_render_frames runs scene.get_world() (BvhNode::new, same seeds, hence the same tree) and the
 * upload before each frame; the time that takes is kept apart from seconds_render and read back with
 * pth_scene_rebuild_seconds after the call.  Fails for worlds made by pth_world_create (no scene to rebuild from). */
int pth_scene_set_rebuild_per_frame(PthScene* s, int on);
int pth_scene_rebuild_seconds(PthScene* s, double* seconds_build, double* seconds_upload);
/* OBJ reader probe: number of models / triangles / materials the reader finds in a file (-1 on error). */
int pth_obj_probe(const char* path, uint32_t* n_models, uint64_t* n_triangles, uint32_t* n_materials);

#ifdef __cplusplus
}
#endif
#endif /* PTHOST_H */
