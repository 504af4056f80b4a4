// world.hpp -- host mirror of the reference's world types, names kept:
//   texture.rs   Texture, SolidColorTexture, CheckerTexture, PerlinNoiseTexture
//   material.rs  Material, LambertianMaterial, MetalMaterial, DielectricMaterial, DiffuseLightMaterial
//   geometry.rs  Hittable, Sphere, RectangleXY/XZ/YZ (+ RectangleNotAxisAlignedError), AABox, Triangle
//   bvh.rs       Aabb, BvhNode
// The reference's trait objects carry the CPU implementation (hit / scatter / value).  Here they carry only
// what the GPU backend needs from the host: bounding_box() for BvhNode::new, and flatten() which writes the
// object into the POD tables of include/ptgpu.h.  There is no CPU hit path: rendering requires the GPU.
#pragma once
#include <algorithm>
#include <cstring>
#include <map>
#include <memory>
#include <stdexcept>
#include <vector>

#include "ptgpu.h"
#include "vec3.hpp"

namespace pt {

class Flattener;

// ------------------------------------------------------------------ texture.rs
struct Texture { // texture.rs:5-7
    virtual ~Texture() {}
    virtual uint32_t flatten(Flattener& f) const = 0;
};
struct SolidColorTexture : Texture { // texture.rs:9-23
    Color color;
    explicit SolidColorTexture(const Color& c) : color(c) {}
    static std::shared_ptr<SolidColorTexture> new_(const Color& c) { return std::make_shared<SolidColorTexture>(c); }
    uint32_t flatten(Flattener& f) const override;
};
struct CheckerTexture : Texture { // texture.rs:25-45: `odd` where the sine product is negative
    std::shared_ptr<Texture> odd, even;
    CheckerTexture(std::shared_ptr<Texture> o, std::shared_ptr<Texture> e) : odd(o), even(e) {}
    static std::shared_ptr<CheckerTexture> new_(std::shared_ptr<Texture> o, std::shared_ptr<Texture> e) {
        return std::make_shared<CheckerTexture>(o, e);
    }
    uint32_t flatten(Flattener& f) const override;
};
// texture.rs:47-81.  The reference seeds noise::Perlin with rand::random(); here the seed is explicit and the
// permutation is a Fisher-Yates shuffle of 0..255 driven by BuildRng (noise 0.8.2 stand-in).
struct PerlinNoiseTexture : Texture {
    uint8_t perm[256];
    double scale;
    PerlinNoiseTexture(double scale_, uint64_t seed) : scale(scale_) {
        for (int i = 0; i < 256; ++i) perm[i] = (uint8_t)i;
        BuildRng rng(seed);
        for (int i = 255; i >= 1; --i) std::swap(perm[i], perm[rng.below((uint32_t)i + 1u)]);
    }
    static std::shared_ptr<PerlinNoiseTexture> new_(double scale, uint64_t seed) {
        return std::make_shared<PerlinNoiseTexture>(scale, seed);
    }
    uint32_t flatten(Flattener& f) const override;
};

// ------------------------------------------------------------------ material.rs
struct Material { // material.rs:15-22
    virtual ~Material() {}
    virtual uint32_t flatten(Flattener& f) const = 0;
};
struct LambertianMaterial : Material { // material.rs:24-38
    std::shared_ptr<Texture> albedo;
    explicit LambertianMaterial(std::shared_ptr<Texture> a) : albedo(a) {}
    static std::shared_ptr<LambertianMaterial> new_(std::shared_ptr<Texture> a) { return std::make_shared<LambertianMaterial>(a); }
    static std::shared_ptr<LambertianMaterial> new_from_color(const Color& c) { return new_(SolidColorTexture::new_(c)); }
    uint32_t flatten(Flattener& f) const override;
};
struct MetalMaterial : Material { // material.rs:58-74
    std::shared_ptr<Texture> albedo;
    double fuzz;
    MetalMaterial(std::shared_ptr<Texture> a, double f) : albedo(a), fuzz(f) {}
    static std::shared_ptr<MetalMaterial> new_(std::shared_ptr<Texture> a, double f) { return std::make_shared<MetalMaterial>(a, f); }
    static std::shared_ptr<MetalMaterial> new_from_color(const Color& c, double f) { return new_(SolidColorTexture::new_(c), f); }
    uint32_t flatten(Flattener& f) const override;
};
struct DielectricMaterial : Material { // material.rs:96-106
    double index_of_refraction;
    explicit DielectricMaterial(double ior) : index_of_refraction(ior) {}
    static std::shared_ptr<DielectricMaterial> new_(double ior) { return std::make_shared<DielectricMaterial>(ior); }
    uint32_t flatten(Flattener& f) const override;
};
struct DiffuseLightMaterial : Material { // material.rs:146-156
    std::shared_ptr<Texture> emit;
    explicit DiffuseLightMaterial(std::shared_ptr<Texture> e) : emit(e) {}
    static std::shared_ptr<DiffuseLightMaterial> new_from_color(const Color& c) {
        return std::make_shared<DiffuseLightMaterial>(SolidColorTexture::new_(c));
    }
    uint32_t flatten(Flattener& f) const override;
};

// ------------------------------------------------------------------ bvh.rs Aabb
struct Aabb {
    Vec3 minimum, maximum;
    Aabb() {}
    Aabb(const Vec3& a, const Vec3& b) : minimum(a), maximum(b) {}
    Aabb surrounding_box(const Aabb& o) const { // bvh.rs:37-50
        return Aabb(Vec3(std::fmin(minimum.x(), o.minimum.x()), std::fmin(minimum.y(), o.minimum.y()),
                         std::fmin(minimum.z(), o.minimum.z())),
                    Vec3(std::fmax(maximum.x(), o.maximum.x()), std::fmax(maximum.y(), o.maximum.y()),
                         std::fmax(maximum.z(), o.maximum.z())));
    }
};

// ------------------------------------------------------------------ geometry.rs
struct Hittable { // geometry.rs:46-49 (hit() runs on the GPU)
    virtual ~Hittable() {}
    virtual Aabb bounding_box() const = 0;
    virtual uint32_t flatten(Flattener& f) const = 0; // returns the child reference (PT_REF_*)
};
typedef std::shared_ptr<Hittable> HittablePtr;

struct Sphere : Hittable { // geometry.rs:104-171
    Vec3 center;
    double radius;
    std::shared_ptr<Material> material;
    Sphere(const Vec3& c, double r, std::shared_ptr<Material> m) : center(c), radius(r), material(m) {}
    static std::shared_ptr<Sphere> new_(const Vec3& c, double r, std::shared_ptr<Material> m) {
        return std::make_shared<Sphere>(c, r, m);
    }
    Aabb bounding_box() const override { // :165-170
        double r = std::fabs(radius);
        return Aabb(center - Vec3(r, r, r), center + Vec3(r, r, r));
    }
    uint32_t flatten(Flattener& f) const override;
};

struct RectangleNotAxisAlignedError : std::runtime_error { // geometry.rs:173-180
    RectangleNotAxisAlignedError() : std::runtime_error("rectangle axis is not aligned") {}
};

inline double signum(double v) { // f64::signum
    if (v != v) return v;
    return std::signbit(v) ? -1.0 : 1.0;
}

// RectangleXY / XZ / YZ (geometry.rs:182-369): AXIS = plane normal, A and B = in-plane axes in reference order.
// new_ throws RectangleNotAxisAlignedError where the reference returns Err (geometry.rs:197-199 etc.).
template <int AXIS, int A, int B, uint32_t TYPE>
struct Rectangle : Hittable {
    Vec3 start, end;
    double direction;
    std::shared_ptr<Material> material;
    Rectangle(const Vec3& s, const Vec3& e, double dir, std::shared_ptr<Material> m) : material(m) {
        if (s.e[AXIS] != e.e[AXIS]) throw RectangleNotAxisAlignedError();
        start.e[A] = std::fmin(s.e[A], e.e[A]);
        start.e[B] = std::fmin(s.e[B], e.e[B]);
        start.e[AXIS] = s.e[AXIS];
        end.e[A] = std::fmax(s.e[A], e.e[A]);
        end.e[B] = std::fmax(s.e[B], e.e[B]);
        end.e[AXIS] = s.e[AXIS];
        direction = signum(dir);
    }
    static std::shared_ptr<Rectangle> new_(const Vec3& s, const Vec3& e, double dir, std::shared_ptr<Material> m) {
        return std::make_shared<Rectangle>(s, e, dir, m);
    }
    Aabb bounding_box() const override { // :234-242 etc.: padded by 0.0001 on the thin axis
        Vec3 lo = start, hi = end;
        lo.e[AXIS] = start.e[AXIS] - 0.0001;
        hi.e[AXIS] = start.e[AXIS] + 0.0001;
        return Aabb(lo, hi);
    }
    PtPrim to_prim(uint32_t material_id) const {
        PtPrim p;
        std::memset(&p, 0, sizeof(p));
        p.type = TYPE;
        p.material = material_id;
        p.d[0] = start.e[AXIS];
        p.d[1] = start.e[A];
        p.d[2] = end.e[A];
        p.d[3] = start.e[B];
        p.d[4] = end.e[B];
        p.d[5] = direction;
        return p;
    }
    uint32_t flatten(Flattener& f) const override;
};
typedef Rectangle<2, 0, 1, PT_PRIM_RECT_XY> RectangleXY;
typedef Rectangle<1, 0, 2, PT_PRIM_RECT_XZ> RectangleXZ;
typedef Rectangle<0, 1, 2, PT_PRIM_RECT_YZ> RectangleYZ;

struct AABox : Hittable { // geometry.rs:371-464
    Vec3 minimum, maximum;
    std::shared_ptr<RectangleXY> xy0, xy1;
    std::shared_ptr<RectangleXZ> xz0, xz1;
    std::shared_ptr<RectangleYZ> yz0, yz1;
    AABox(const Vec3& s, const Vec3& e, std::shared_ptr<Material> m) {
        minimum = Vec3(std::fmin(s.x(), e.x()), std::fmin(s.y(), e.y()), std::fmin(s.z(), e.z()));
        maximum = Vec3(std::fmax(s.x(), e.x()), std::fmax(s.y(), e.y()), std::fmax(s.z(), e.z()));
        const Vec3& lo = minimum;
        const Vec3& hi = maximum;
        xy0 = RectangleXY::new_(Vec3(lo.x(), lo.y(), lo.z()), Vec3(hi.x(), hi.y(), lo.z()), -1.0, m);
        xy1 = RectangleXY::new_(Vec3(lo.x(), lo.y(), hi.z()), Vec3(hi.x(), hi.y(), hi.z()), 1.0, m);
        xz0 = RectangleXZ::new_(Vec3(lo.x(), lo.y(), lo.z()), Vec3(hi.x(), lo.y(), hi.z()), -1.0, m);
        xz1 = RectangleXZ::new_(Vec3(lo.x(), hi.y(), lo.z()), Vec3(hi.x(), hi.y(), hi.z()), 1.0, m);
        yz0 = RectangleYZ::new_(Vec3(lo.x(), lo.y(), lo.z()), Vec3(lo.x(), hi.y(), hi.z()), -1.0, m);
        yz1 = RectangleYZ::new_(Vec3(hi.x(), lo.y(), lo.z()), Vec3(hi.x(), hi.y(), hi.z()), 1.0, m);
    }
    static std::shared_ptr<AABox> new_(const Vec3& s, const Vec3& e, std::shared_ptr<Material> m) {
        return std::make_shared<AABox>(s, e, m);
    }
    Aabb bounding_box() const override { return Aabb(minimum, maximum); } // :461-463 (unpadded)
    uint32_t flatten(Flattener& f) const override;
};

struct Triangle : Hittable { // geometry.rs:466-589
    Vec3 point1, point2, point3, normal;
    std::shared_ptr<Material> material;
    Triangle(const Vec3& p1, const Vec3& p2, const Vec3& p3, const Vec3& n, std::shared_ptr<Material> m)
        : point1(p1), point2(p2), point3(p3), normal(n), material(m) {}
    static std::shared_ptr<Triangle> new_(const Vec3& p1, const Vec3& p2, const Vec3& p3, const Vec3& n,
                                          std::shared_ptr<Material> m) { // :486-500
        return std::make_shared<Triangle>(p1, p2, p3, n, m);
    }
    static std::shared_ptr<Triangle> new_without_normal(const Vec3& p1, const Vec3& p2, const Vec3& p3,
                                                        std::shared_ptr<Material> m) { // :502-515
        return std::make_shared<Triangle>(p1, p2, p3, (p2 - p1).cross(p3 - p1).unit_vector(), m);
    }
    Aabb bounding_box() const override { // :561-588
        Vec3 lo(std::fmin(std::fmin(point1.x(), point2.x()), point3.x()), std::fmin(std::fmin(point1.y(), point2.y()), point3.y()),
                std::fmin(std::fmin(point1.z(), point2.z()), point3.z()));
        Vec3 hi(std::fmax(std::fmax(point1.x(), point2.x()), point3.x()), std::fmax(std::fmax(point1.y(), point2.y()), point3.y()),
                std::fmax(std::fmax(point1.z(), point2.z()), point3.z()));
        if (lo.x() == hi.x()) { lo = lo + Vec3(-0.001, 0.0, 0.0); hi = hi + Vec3(0.001, 0.0, 0.0); }
        if (lo.y() == hi.y()) { lo = lo + Vec3(0.0, -0.001, 0.0); hi = hi + Vec3(0.0, 0.001, 0.0); }
        if (lo.z() == hi.z()) { lo = lo + Vec3(0.0, 0.0, -0.001); hi = hi + Vec3(0.0, 0.0, 0.001); }
        return Aabb(lo, hi);
    }
    uint32_t flatten(Flattener& f) const override;
};

// ------------------------------------------------------------------ bvh.rs BvhNode
struct BvhNode : Hittable {
    HittablePtr left, right;
    Aabb bbox;
    // Identifies this tree among all trees built by the process (a copy keeps it: same contents).  pt::render() keys its
    // upload cache on it -- never on an address, which a later world can reuse.
    uint64_t generation = next_generation();
    static uint64_t next_generation();
    // BvhNode::new (bvh.rs:65-103) with the axis draws taken from `rng` (the reference: thread_rng).
    // Throws std::runtime_error("creating BVH node from empty object list") where the reference panics (:76).
    BvhNode(const std::vector<HittablePtr>& objects, BuildRng& rng);
    static std::shared_ptr<BvhNode> new_(const std::vector<HittablePtr>& objects, BuildRng& rng) {
        return std::make_shared<BvhNode>(objects, rng);
    }
    Aabb bounding_box() const override { return bbox; }
    uint32_t flatten(Flattener& f) const override;
};

// ------------------------------------------------------------------ flattening into the ptgpu.h tables
// Walks the world the way BvhNode::hit does (left before right, AABox sides in list order, ObjModel -> its
// nested root), so primitives land in the table in depth-first order: the table index is the parity
// contract's "hit primitive index".
class Flattener {
public:
    std::vector<PtNode> nodes;
    std::vector<PtPrim> prims;
    std::vector<PtMaterial> materials;
    std::vector<PtTexture> textures;
    std::vector<uint8_t> perlin_tables;
    uint32_t root = 0;

    uint32_t material_id(const std::shared_ptr<Material>& m) {
        auto it = material_ids_.find(m.get());
        if (it != material_ids_.end()) return it->second;
        uint32_t id = m->flatten(*this);
        material_ids_[m.get()] = id;
        return id;
    }
    uint32_t texture_id(const std::shared_ptr<Texture>& t) {
        auto it = texture_ids_.find(t.get());
        if (it != texture_ids_.end()) return it->second;
        uint32_t id = t->flatten(*this);
        texture_ids_[t.get()] = id;
        return id;
    }
    void flatten_world(const BvhNode& world) { root = world.flatten(*this); }
    PtSceneDesc desc() const {
        PtSceneDesc d;
        std::memset(&d, 0, sizeof(d));
        d.nodes = nodes.data();
        d.n_nodes = (uint32_t)nodes.size();
        d.root = root;
        d.prims = prims.data();
        d.n_prims = (uint32_t)prims.size();
        d.materials = materials.data();
        d.n_materials = (uint32_t)materials.size();
        d.textures = textures.data();
        d.n_textures = (uint32_t)textures.size();
        d.perlin_tables = perlin_tables.data();
        d.n_perlin_tables = (uint32_t)(perlin_tables.size() / 256);
        return d;
    }

private:
    std::map<const Material*, uint32_t> material_ids_;
    std::map<const Texture*, uint32_t> texture_ids_;
};

template <int AXIS, int A, int B, uint32_t TYPE>
uint32_t Rectangle<AXIS, A, B, TYPE>::flatten(Flattener& f) const {
    uint32_t idx = (uint32_t)f.prims.size();
    f.prims.push_back(to_prim(f.material_id(material)));
    return PT_REF_LEAF | idx;
}

} // namespace pt
