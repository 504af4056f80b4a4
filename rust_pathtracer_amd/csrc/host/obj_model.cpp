// obj_model.cpp -- OBJ/MTL reader (tobj-shaped output) and ObjModel construction (src/obj_model.rs:19-111).
#include "obj_model.hpp"

#include <cstdlib>
#include <fstream>
#include <sstream>
#include <unordered_map>

namespace pt {

namespace {

struct FaceVertex {
    long v = 0, vt = 0, vn = 0; // 1-based as written, 0 = absent, negative = relative
};

bool parse_face_vertex(const std::string& tok, FaceVertex& out) {
    // v, v/vt, v//vn, v/vt/vn
    out = FaceVertex();
    size_t s1 = tok.find('/');
    try {
        if (s1 == std::string::npos) {
            out.v = std::stol(tok);
            return out.v != 0;
        }
        out.v = std::stol(tok.substr(0, s1));
        size_t s2 = tok.find('/', s1 + 1);
        if (s2 == std::string::npos) {
            if (s1 + 1 < tok.size()) out.vt = std::stol(tok.substr(s1 + 1));
        } else {
            if (s2 > s1 + 1) out.vt = std::stol(tok.substr(s1 + 1, s2 - s1 - 1));
            if (s2 + 1 < tok.size()) out.vn = std::stol(tok.substr(s2 + 1));
        }
    } catch (...) {
        return false;
    }
    return out.v != 0;
}

// Resolves a 1-based / negative OBJ index against `count` elements seen so far; returns -1 if out of range.
long resolve_index(long idx, size_t count) {
    long r = idx > 0 ? idx - 1 : (long)count + idx;
    if (r < 0 || (size_t)r >= count) return -1;
    return r;
}

std::string dirname_of(const std::string& path) {
    size_t p = path.find_last_of("/\\");
    return p == std::string::npos ? std::string("") : path.substr(0, p + 1);
}

void load_mtl(const std::string& path, std::vector<ObjMaterial>& out, std::unordered_map<std::string, int>& by_name) {
    std::ifstream in(path);
    if (!in) throw std::runtime_error("could not open MTL file " + path);
    std::string line;
    ObjMaterial* cur = nullptr;
    while (std::getline(in, line)) {
        std::istringstream ss(line);
        std::string key;
        if (!(ss >> key) || key[0] == '#') continue;
        if (key == "newmtl") {
            std::string name;
            ss >> name;
            by_name[name] = (int)out.size();
            out.emplace_back();
            out.back().name = name;
            cur = &out.back();
        } else if (cur) {
            if (key == "Kd") ss >> cur->diffuse[0] >> cur->diffuse[1] >> cur->diffuse[2];
            else if (key == "Ns") ss >> cur->shininess;
            else if (key == "Ni") ss >> cur->optical_density;
            else if (key == "illum") { int v = 0; if (ss >> v) cur->illumination_model = v; }
        }
    }
}

// A model under construction: faces as global indices; exported with per-model re-indexing like tobj's
// multi-index path (positions and normals each numbered in order of first use).
struct PendingModel {
    std::vector<FaceVertex> corners; // triangulated, 3 per face, indices already resolved to 0-based (vn = -1 if absent)
    int material_id = -1;
};

void export_model(const PendingModel& pm, const std::vector<float>& pos, const std::vector<float>& nrm, ObjFile& file) {
    if (pm.corners.empty()) return;
    ObjMesh mesh;
    mesh.material_id = pm.material_id;
    std::unordered_map<long, uint32_t> pos_map, nrm_map;
    for (const FaceVertex& c : pm.corners) {
        auto it = pos_map.find(c.v);
        uint32_t idx;
        if (it == pos_map.end()) {
            idx = (uint32_t)(mesh.positions.size() / 3);
            pos_map[c.v] = idx;
            mesh.positions.insert(mesh.positions.end(), pos.begin() + c.v * 3, pos.begin() + c.v * 3 + 3);
        } else {
            idx = it->second;
        }
        mesh.indices.push_back(idx);
        if (c.vn >= 0 && nrm_map.find(c.vn) == nrm_map.end()) {
            nrm_map[c.vn] = (uint32_t)(mesh.normals.size() / 3);
            mesh.normals.insert(mesh.normals.end(), nrm.begin() + c.vn * 3, nrm.begin() + c.vn * 3 + 3);
        }
    }
    file.models.push_back(std::move(mesh));
}

} // namespace

ObjFile load_obj(const std::string& path) {
    std::ifstream in(path);
    if (!in) throw std::runtime_error("could not open OBJ file " + path);
    ObjFile file;
    std::unordered_map<std::string, int> mtl_by_name;
    std::vector<float> pos, nrm;
    PendingModel cur;
    std::string line;
    while (std::getline(in, line)) {
        std::istringstream ss(line);
        std::string key;
        if (!(ss >> key) || key[0] == '#') continue;
        if (key == "v") {
            float x = 0, y = 0, z = 0;
            ss >> x >> y >> z;
            pos.push_back(x); pos.push_back(y); pos.push_back(z);
        } else if (key == "vn") {
            float x = 0, y = 0, z = 0;
            ss >> x >> y >> z;
            nrm.push_back(x); nrm.push_back(y); nrm.push_back(z);
        } else if (key == "f") {
            std::vector<FaceVertex> poly;
            std::string tok;
            while (ss >> tok) {
                FaceVertex fv;
                if (!parse_face_vertex(tok, fv)) throw std::runtime_error("malformed face in " + path);
                long v = resolve_index(fv.v, pos.size() / 3);
                if (v < 0) throw std::runtime_error("face references a missing vertex in " + path);
                long vn = -1;
                if (fv.vn != 0) {
                    vn = resolve_index(fv.vn, nrm.size() / 3);
                    if (vn < 0) throw std::runtime_error("face references a missing normal in " + path);
                }
                FaceVertex r;
                r.v = v;
                r.vn = vn;
                poly.push_back(r);
            }
            if (poly.size() < 3) continue; // ignore_lines / ignore_points
            for (size_t i = 1; i + 1 < poly.size(); ++i) { // triangulate: fan around corner 0
                cur.corners.push_back(poly[0]);
                cur.corners.push_back(poly[i]);
                cur.corners.push_back(poly[i + 1]);
            }
        } else if (key == "o" || key == "g") {
            export_model(cur, pos, nrm, file);
            cur.corners.clear();
        } else if (key == "usemtl") {
            std::string name;
            ss >> name;
            export_model(cur, pos, nrm, file);
            cur.corners.clear();
            auto it = mtl_by_name.find(name);
            cur.material_id = it == mtl_by_name.end() ? -1 : it->second;
        } else if (key == "mtllib") {
            std::string name;
            ss >> name;
            load_mtl(dirname_of(path) + name, file.materials, mtl_by_name);
        }
    }
    export_model(cur, pos, nrm, file);
    return file;
}

std::shared_ptr<ObjModel> ObjModel::new_from_arrays(const ObjFile& file, BuildRng& rng) {
    // obj_model.rs:28-50
    std::vector<std::shared_ptr<Material>> materials_mapped;
    for (const ObjMaterial& m : file.materials) {
        if (m.illumination_model == 7)
            materials_mapped.push_back(DielectricMaterial::new_((double)m.optical_density));
        else if (m.illumination_model == 5)
            materials_mapped.push_back(MetalMaterial::new_from_color(
                Color((double)m.diffuse[0], (double)m.diffuse[1], (double)m.diffuse[2]), 1.0 / (double)m.shininess));
        else
            materials_mapped.push_back(
                LambertianMaterial::new_from_color(Color((double)m.diffuse[0], (double)m.diffuse[1], (double)m.diffuse[2])));
    }
    // obj_model.rs:52-100
    std::vector<HittablePtr> world;
    // obj_model.rs:82-84 makes a fresh Lambertian(0.2, 0.7, 0.2) per triangle; one shared instance renders the same
    std::shared_ptr<Material> default_material = LambertianMaterial::new_from_color(Color(0.2, 0.7, 0.2));
    for (const ObjMesh& mesh : file.models) {
        const size_t n_pos = mesh.positions.size() / 3;
        for (size_t tri = 0; tri < mesh.indices.size() / 3; ++tri) {
            size_t i0 = mesh.indices[tri * 3], i1 = mesh.indices[tri * 3 + 1], i2 = mesh.indices[tri * 3 + 2];
            if (i0 >= n_pos || i1 >= n_pos || i2 >= n_pos) throw std::runtime_error("mesh index out of range");
            Vec3 v0((double)mesh.positions[i0 * 3], (double)mesh.positions[i0 * 3 + 1], (double)mesh.positions[i0 * 3 + 2]);
            Vec3 v1((double)mesh.positions[i1 * 3], (double)mesh.positions[i1 * 3 + 1], (double)mesh.positions[i1 * 3 + 2]);
            Vec3 v2((double)mesh.positions[i2 * 3], (double)mesh.positions[i2 * 3 + 1], (double)mesh.positions[i2 * 3 + 2]);
            std::shared_ptr<Material> material;
            if (mesh.material_id >= 0) {
                if ((size_t)mesh.material_id >= materials_mapped.size()) throw std::runtime_error("mesh material out of range");
                material = materials_mapped[mesh.material_id];
            } else {
                material = default_material;
            }
            if (mesh.normals.empty()) {
                world.push_back(Triangle::new_without_normal(v0, v1, v2, material));
            } else {
                // obj_model.rs:90-95: the normal is looked up with vertex 0's POSITION index and kept verbatim
                if (i0 * 3 + 2 >= mesh.normals.size()) throw std::runtime_error("normal index out of range");
                Vec3 n((double)mesh.normals[i0 * 3], (double)mesh.normals[i0 * 3 + 1], (double)mesh.normals[i0 * 3 + 2]);
                world.push_back(Triangle::new_(v0, v1, v2, n, material));
            }
        }
    }
    if (world.empty()) throw std::runtime_error("called bounding box on empty hittable list"); // geometry.rs:86-88
    auto model = std::make_shared<ObjModel>();
    Aabb bb = world[0]->bounding_box(); // obj_model.rs:102 -> geometry.rs:85-101
    for (size_t i = 1; i < world.size(); ++i) bb = bb.surrounding_box(world[i]->bounding_box());
    model->minimum = bb.minimum;
    model->maximum = bb.maximum;
    model->triangles = BvhNode::new_(world, rng); // obj_model.rs:107
    return model;
}

std::shared_ptr<ObjModel> ObjModel::new_from_path(const std::string& path, BuildRng& rng) {
    return new_from_arrays(load_obj(path), rng);
}

} // namespace pt
