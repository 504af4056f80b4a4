// world.cpp -- BvhNode::new and the flatten() bodies of world.hpp.
#include "world.hpp"

#include <atomic>

#include <numeric>

namespace pt {

uint32_t SolidColorTexture::flatten(Flattener& f) const {
    PtTexture t;
    std::memset(&t, 0, sizeof(t));
    t.type = PT_TEX_SOLID;
    for (int k = 0; k < 3; ++k) t.color[k] = color.e[k];
    f.textures.push_back(t);
    return (uint32_t)f.textures.size() - 1;
}
uint32_t CheckerTexture::flatten(Flattener& f) const {
    uint32_t o = f.texture_id(odd), e = f.texture_id(even);
    PtTexture t;
    std::memset(&t, 0, sizeof(t));
    t.type = PT_TEX_CHECKER;
    t.odd = o;
    t.even = e;
    f.textures.push_back(t);
    return (uint32_t)f.textures.size() - 1;
}
uint32_t PerlinNoiseTexture::flatten(Flattener& f) const {
    PtTexture t;
    std::memset(&t, 0, sizeof(t));
    t.type = PT_TEX_PERLIN;
    t.scale = scale;
    t.perlin_table = (uint32_t)(f.perlin_tables.size() / 256);
    f.perlin_tables.insert(f.perlin_tables.end(), perm, perm + 256);
    f.textures.push_back(t);
    return (uint32_t)f.textures.size() - 1;
}

static uint32_t push_material(Flattener& f, uint32_t type, uint32_t texture, double param) {
    PtMaterial m;
    std::memset(&m, 0, sizeof(m));
    m.type = type;
    m.texture = texture;
    m.param = param;
    f.materials.push_back(m);
    return (uint32_t)f.materials.size() - 1;
}
uint32_t LambertianMaterial::flatten(Flattener& f) const { return push_material(f, PT_MAT_LAMBERTIAN, f.texture_id(albedo), 0.0); }
uint32_t MetalMaterial::flatten(Flattener& f) const { return push_material(f, PT_MAT_METAL, f.texture_id(albedo), fuzz); }
uint32_t DielectricMaterial::flatten(Flattener& f) const { return push_material(f, PT_MAT_DIELECTRIC, 0, index_of_refraction); }
uint32_t DiffuseLightMaterial::flatten(Flattener& f) const { return push_material(f, PT_MAT_DIFFUSE_LIGHT, f.texture_id(emit), 0.0); }

uint32_t Sphere::flatten(Flattener& f) const {
    PtPrim p;
    std::memset(&p, 0, sizeof(p));
    p.type = PT_PRIM_SPHERE;
    p.material = f.material_id(material);
    for (int k = 0; k < 3; ++k) p.d[k] = center.e[k];
    p.d[3] = radius;
    f.prims.push_back(p);
    return PT_REF_LEAF | ((uint32_t)f.prims.size() - 1);
}

uint32_t AABox::flatten(Flattener& f) const { // side order of geometry.rs:391-446
    uint32_t first = (uint32_t)f.prims.size();
    xy0->flatten(f);
    xy1->flatten(f);
    xz0->flatten(f);
    xz1->flatten(f);
    yz0->flatten(f);
    yz1->flatten(f);
    return PT_REF_LEAF | PT_REF_GROUP6 | first;
}

uint32_t Triangle::flatten(Flattener& f) const {
    PtPrim p;
    std::memset(&p, 0, sizeof(p));
    p.type = PT_PRIM_TRIANGLE;
    p.material = f.material_id(material);
    for (int k = 0; k < 3; ++k) {
        p.d[k] = point1.e[k];
        p.d[3 + k] = point2.e[k];
        p.d[6 + k] = point3.e[k];
        p.d[9 + k] = normal.e[k];
    }
    f.prims.push_back(p);
    return PT_REF_LEAF | ((uint32_t)f.prims.size() - 1);
}

// f64::total_cmp (bvh.rs:52-54) as an integer key.
static inline int64_t total_order_key(double v) {
    int64_t b;
    std::memcpy(&b, &v, 8);
    b ^= (int64_t)((uint64_t)(b >> 63) >> 1);
    return b;
}

BvhNode::BvhNode(const std::vector<HittablePtr>& objects, BuildRng& rng) {
    const uint32_t axis = rng.below(3); // drawn before the length match (bvh.rs:67): pairs are sorted too
    const size_t len = objects.size();
    if (len < 2) throw std::runtime_error("creating BVH node from empty object list"); // bvh.rs:76
    std::vector<int64_t> keys(len);
    for (size_t i = 0; i < len; ++i) keys[i] = total_order_key(objects[i]->bounding_box().minimum.e[axis]);
    std::vector<uint32_t> order(len);
    std::iota(order.begin(), order.end(), 0u);
    std::stable_sort(order.begin(), order.end(), [&](uint32_t a, uint32_t b) { return keys[a] < keys[b]; }); // sort_by is stable
    if (len == 2) { // bvh.rs:77-82
        left = objects[order[0]];
        right = objects[order[1]];
    } else { // bvh.rs:83-101
        const size_t half_index = len / 2;
        std::vector<HittablePtr> left_list(half_index), right_list(len - half_index);
        for (size_t i = 0; i < half_index; ++i) left_list[i] = objects[order[i]];
        for (size_t i = half_index; i < len; ++i) right_list[i - half_index] = objects[order[i]];
        left = left_list.size() == 1 ? left_list[0] : HittablePtr(new BvhNode(left_list, rng));
        right = right_list.size() == 1 ? right_list[0] : HittablePtr(new BvhNode(right_list, rng));
    }
    bbox = left->bounding_box().surrounding_box(right->bounding_box());
}

uint32_t BvhNode::flatten(Flattener& f) const {
    uint32_t idx = (uint32_t)f.nodes.size();
    f.nodes.emplace_back();
    uint32_t l = left->flatten(f);  // left subtree first: depth-first primitive order
    uint32_t r = right->flatten(f);
    PtNode& n = f.nodes[idx];
    for (int k = 0; k < 3; ++k) {
        n.bmin[k] = bbox.minimum.e[k];
        n.bmax[k] = bbox.maximum.e[k];
    }
    n.left = l;
    n.right = r;
    return idx;
}

uint64_t BvhNode::next_generation() {
    static std::atomic<uint64_t> counter{0};
    return ++counter;
}

} // namespace pt
