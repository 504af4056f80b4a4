// obj_model.hpp -- host mirror of src/obj_model.rs: ObjModel::new_from_path.
// The reference parses with the tobj crate (3.2.3, LoadOptions{single_index:false, triangulate:true,
// ignore_lines, ignore_points}); tobj is not under /root/reference, so load_obj() below is our own reader
// that returns the same shapes: per model positions / indices / normals (f32, re-indexed per model in
// order of first use, positions and normals independently) and a material id, plus the MTL fields
// obj_model.rs:32-46 reads.
#pragma once
#include <string>

#include "world.hpp"

namespace pt {

struct ObjMesh { // tobj::Mesh fields used at obj_model.rs:56-96
    std::vector<float> positions;
    std::vector<uint32_t> indices;
    std::vector<float> normals;
    int material_id = -1; // Option<usize>
};
struct ObjMaterial { // tobj::Material fields used at obj_model.rs:32-46
    std::string name;
    int illumination_model = -1; // Option<u8>
    float optical_density = 1.0f;
    float diffuse[3] = {0.f, 0.f, 0.f};
    float shininess = 0.f;
};
struct ObjFile {
    std::vector<ObjMesh> models;
    std::vector<ObjMaterial> materials;
};

// Throws std::runtime_error where the reference's `.unwrap()`s would panic (obj_model.rs:26,29): unreadable
// OBJ, unreadable MTL named by mtllib, malformed face.
ObjFile load_obj(const std::string& path);

struct ObjModel : Hittable { // obj_model.rs:11-16
    std::shared_ptr<BvhNode> triangles;
    Vec3 minimum, maximum;
    // obj_model.rs:19-111.  The nested tree's axis draws come from `rng`.
    static std::shared_ptr<ObjModel> new_from_path(const std::string& path, BuildRng& rng);
    // The same construction from already-parsed arrays (used for synthetic meshes).
    static std::shared_ptr<ObjModel> new_from_arrays(const ObjFile& file, BuildRng& rng);
    Aabb bounding_box() const override { return Aabb(minimum, maximum); } // :115-117
    uint32_t flatten(Flattener& f) const override { return triangles->flatten(f); } // hit() forwards to the nested root, :119-121
};

} // namespace pt
