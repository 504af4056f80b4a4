// oracle.cpp -- CPU restatement of ArcticXWolf/rust-pathtracer's render path.
//
// THIS FILE IS TEST INFRASTRUCTURE.  Only tests/, __graft_entry__.smoke() and
// bench.py's cpu_baseline leg may load it, and only as the checker / reported
// baseline.  Nothing in rust_pathtracer_amd/ (the product) links or calls it.
//
// What it is: an f64, recursive, pointer-tree restatement of the reference's
// algorithm, function by function, each citing the reference file:line it
// follows.  It is compiled with -ffp-contract=off because rustc never fuses
// a*b+c: every +,-,*,/ and sqrt below is one IEEE-754 binary64 operation in the
// reference's operand order.
//
// PARITY PINNING.  The reference holds exactly one test (vec3.rs:256-274, six
// known answers); tests/test_oracle.py checks those against this file.  The
// reference cannot be compiled here (Rust: no rustc/cargo in the image, crates
// not vendored), so everything else on the path -- hit tests, BVH, materials,
// textures, render -- is "parity unpinned" by reference fixtures and is pinned
// instead by analytic known answers derived by hand from scene.rs constants and
// by brute-force-vs-BVH equivalence (tests/test_oracle.py).
//
// Third-party arithmetic that is NOT under /root/reference, restated from the
// published algorithm:
//   rand 0.8.5   (thread_rng, OS-seeded ChaCha12): unseeded in the reference, so
//                only distributions matter.  Replaced by two documented
//                generators: SplitMix64 for build-time draws (BVH axis,
//                SphereField layout, Perlin permutation) and the counter-based
//                Philox4x32-10 for per-sample draws, keyed by
//                (pixel, sample, stream, block) so a sample is reproducible
//                wherever it is computed.
//   noise 0.8.2  (Perlin::new(seed).get([f64;3])): improved 3-D Perlin, 256-entry
//                permutation, 12 edge gradients chosen by hash&15, quintic fade,
//                polynomial trilinear blend, scaled by 2/sqrt(3) and clamped to
//                [-1,1].  Restated from memory of the crate; unpinned.
//   tobj 3.2.3   (OBJ parse): the oracle takes the arrays tobj would return
//                (positions/indices/normals per model, MTL fields) and restates
//                obj_model.rs:28-110 on them.
#include <algorithm>
#include <atomic>
#include <cmath>
#include <cstdint>
#include <cstring>
#include <limits>
#include <memory>
#include <stdexcept>
#include <string>
#include <thread>
#include <vector>

namespace {

// ---------------------------------------------------------------- vec3.rs
struct Vec3 {
    double e[3];
    Vec3() : e{0.0, 0.0, 0.0} {}                       // vec3.rs:14-18
    Vec3(double x, double y, double z) : e{x, y, z} {} // vec3.rs:21-23
    double x() const { return e[0]; }
    double y() const { return e[1]; }
    double z() const { return e[2]; }
    double len_squared() const { return x() * x() + y() * y() + z() * z(); } // :83-85
    double len() const { return std::sqrt(len_squared()); }                   // :79-81
    double dot(const Vec3& r) const { return x() * r.x() + y() * r.y() + z() * r.z(); } // :87-89
    Vec3 cross(const Vec3& r) const {                                                    // :91-97
        return Vec3(y() * r.z() - z() * r.y(), z() * r.x() - x() * r.z(), x() * r.y() - y() * r.x());
    }
    bool near_zero() const { // :110-112
        const double NEAR_ZERO = 1e-8;
        return std::fabs(x()) < NEAR_ZERO && std::fabs(y()) < NEAR_ZERO && std::fabs(z()) < NEAR_ZERO;
    }
};
inline Vec3 operator+(const Vec3& a, const Vec3& b) { return Vec3(a.x() + b.x(), a.y() + b.y(), a.z() + b.z()); } // :130-136
inline Vec3 operator-(const Vec3& a, const Vec3& b) { return Vec3(a.x() - b.x(), a.y() - b.y(), a.z() - b.z()); } // :146-152
inline Vec3 operator*(const Vec3& a, const Vec3& b) { return Vec3(a.x() * b.x(), a.y() * b.y(), a.z() * b.z()); } // :162-168
inline Vec3 operator*(const Vec3& a, double t) { return Vec3(a.x() * t, a.y() * t, a.z() * t); }                  // :178-184
inline Vec3 operator*(double t, const Vec3& a) { return Vec3(a.x() * t, a.y() * t, a.z() * t); }                  // :194-200
inline Vec3 operator/(const Vec3& a, double t) { return Vec3(a.x() / t, a.y() / t, a.z() / t); }                  // :218-224
inline Vec3 operator-(const Vec3& a) { return Vec3(a.x() * -1.0, a.y() * -1.0, a.z() * -1.0); }                   // :248-254
inline Vec3 unit_vector(const Vec3& a) { return a / a.len(); }                                                    // :99-101
inline Vec3 reflect(const Vec3& v, const Vec3& n) { return v - 2.0 * v.dot(n) * n; }                              // :114-116
inline Vec3 refract(const Vec3& v, const Vec3& n, double etai_over_etat) {                                         // :118-127
    double cos_theta = std::fmin((-v).dot(n), 1.0);
    Vec3 perp = etai_over_etat * (v + cos_theta * n);
    Vec3 parallel = (-std::sqrt(std::fabs(1.0 - perp.len_squared()))) * n;
    return parallel + perp;
}
// vec3.rs:278-290: `(c * 255.999) as u8` -- Rust's float->int `as` saturates and maps NaN to 0.
inline uint8_t to_u8(double c) {
    double v = c * 255.999;
    if (!(v == v)) return 0;
    if (v <= 0.0) return 0;
    if (v >= 255.0) return 255;
    return (uint8_t)v;
}
typedef Vec3 Color;

// ------------------------------------------------------ build-time RNG (ours)
struct SplitMix64 {
    uint64_t s;
    explicit SplitMix64(uint64_t seed) : s(seed) {}
    uint64_t next() {
        s += 0x9E3779B97F4A7C15ull;
        uint64_t z = s;
        z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
        z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
        return z ^ (z >> 31);
    }
    double f64() { return (double)(next() >> 11) * (1.0 / 9007199254740992.0); } // rand::random::<f64>() stand-in
    uint32_t below(uint32_t n) { return (uint32_t)(((next() >> 32) * (uint64_t)n) >> 32); } // gen_range(0..n) stand-in
};

// ------------------------------------------------- per-sample RNG (ours): Philox4x32-10
inline void philox4x32_10(const uint32_t ctr[4], const uint32_t key[2], uint32_t out[4]) {
    uint32_t c0 = ctr[0], c1 = ctr[1], c2 = ctr[2], c3 = ctr[3];
    uint32_t k0 = key[0], k1 = key[1];
    for (int r = 0; r < 10; ++r) {
        uint64_t p0 = (uint64_t)0xD2511F53u * c0;
        uint64_t p1 = (uint64_t)0xCD9E8D57u * c2;
        uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0;
        uint32_t n1 = (uint32_t)p1;
        uint32_t n2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1;
        uint32_t n3 = (uint32_t)p0;
        c0 = n0; c1 = n1; c2 = n2; c3 = n3;
        k0 += 0x9E3779B9u;
        k1 += 0xBB67AE85u;
    }
    out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
}

// Draw k of stream (pixel, sample, stream) is 64 bits: block k/2, half k%2.
struct SampleRng {
    uint32_t pixel, sample, stream, n;
    uint32_t key[2];
    uint32_t cached_block;
    uint32_t cache[4];
    SampleRng(uint64_t seed, uint32_t pixel_, uint32_t sample_, uint32_t stream_)
        : pixel(pixel_), sample(sample_), stream(stream_), n(0), cached_block(0xFFFFFFFFu), cache{0, 0, 0, 0} {
        key[0] = (uint32_t)seed;
        key[1] = (uint32_t)(seed >> 32);
    }
    double f64() { // rand::random::<f64>() stand-in: 53 bits in [0,1)
        uint32_t k = n++;
        uint32_t block = k >> 1;
        if (block != cached_block) {
            uint32_t ctr[4] = {pixel, sample, stream, block};
            philox4x32_10(ctr, key, cache);
            cached_block = block;
        }
        uint64_t bits = (k & 1u) ? (((uint64_t)cache[3] << 32) | cache[2]) : (((uint64_t)cache[1] << 32) | cache[0]);
        return (double)(bits >> 11) * (1.0 / 9007199254740992.0);
    }
    double range(double lo, double hi) { return lo + (hi - lo) * f64(); } // gen_range(lo..hi) stand-in
};

const int REJECTION_CAP = 256; // the reference loops forever; both sides here stop at the same draw

// vec3.rs:41-49 (three gen_range(-1..1) per try, vec3.rs:33-39)
inline Vec3 random_in_unitsphere(SampleRng& rng) {
    Vec3 p;
    for (int i = 0; i < REJECTION_CAP; ++i) {
        double x = rng.range(-1.0, 1.0);
        double y = rng.range(-1.0, 1.0);
        double z = rng.range(-1.0, 1.0);
        p = Vec3(x, y, z);
        if (p.len_squared() < 1.0) return p;
    }
    return p;
}
inline Vec3 random_on_unitsphere(SampleRng& rng) { return unit_vector(random_in_unitsphere(rng)); } // vec3.rs:51-53
// vec3.rs:55-67
inline Vec3 random_in_unitdisk_xy(SampleRng& rng) {
    Vec3 p;
    for (int i = 0; i < REJECTION_CAP; ++i) {
        double x = rng.range(-1.0, 1.0);
        double y = rng.range(-1.0, 1.0);
        p = Vec3(x, y, 0.0);
        if (p.len_squared() < 1.0) return p;
    }
    return p;
}

// ---------------------------------------------------------------- ray.rs
struct Ray {
    Vec3 origin, direction;
    Ray() {}
    Ray(const Vec3& o, const Vec3& d) : origin(o), direction(d) {}
    Vec3 at(double t) const { return origin + t * direction; } // ray.rs:16-18
};

// ---------------------------------------------------------------- texture.rs
struct Perlin { // noise 0.8.2 stand-in, see header
    uint8_t perm[256];
    explicit Perlin(uint64_t seed) {
        for (int i = 0; i < 256; ++i) perm[i] = (uint8_t)i;
        SplitMix64 rng(seed);
        for (int i = 255; i >= 1; --i) {
            uint32_t j = rng.below((uint32_t)i + 1u);
            std::swap(perm[i], perm[j]);
        }
    }
    int hash(long x, long y, long z) const {
        int a = perm[x & 0xff];
        int b = perm[a ^ (int)(y & 0xff)];
        return perm[b ^ (int)(z & 0xff)];
    }
    static double grad(int h, double x, double y, double z) {
        switch (h & 15) {
            case 0: case 12: return x + y;
            case 1: case 13: return -x + y;
            case 2: return x - y;
            case 3: return -x - y;
            case 4: return x + z;
            case 5: return -x + z;
            case 6: return x - z;
            case 7: return -x - z;
            case 8: return y + z;
            case 9: case 14: return -y + z;
            case 10: return y - z;
            default: return -y - z; // 11 | 15
        }
    }
    static double quintic(double t) { return t * t * t * (t * (t * 6.0 - 15.0) + 10.0); }
    double get(const double p[3]) const {
        double fx = std::floor(p[0]), fy = std::floor(p[1]), fz = std::floor(p[2]);
        long cx = (long)fx, cy = (long)fy, cz = (long)fz;
        double dx = p[0] - fx, dy = p[1] - fy, dz = p[2] - fz;
        double g000 = grad(hash(cx, cy, cz), dx, dy, dz);
        double g100 = grad(hash(cx + 1, cy, cz), dx - 1.0, dy, dz);
        double g010 = grad(hash(cx, cy + 1, cz), dx, dy - 1.0, dz);
        double g110 = grad(hash(cx + 1, cy + 1, cz), dx - 1.0, dy - 1.0, dz);
        double g001 = grad(hash(cx, cy, cz + 1), dx, dy, dz - 1.0);
        double g101 = grad(hash(cx + 1, cy, cz + 1), dx - 1.0, dy, dz - 1.0);
        double g011 = grad(hash(cx, cy + 1, cz + 1), dx, dy - 1.0, dz - 1.0);
        double g111 = grad(hash(cx + 1, cy + 1, cz + 1), dx - 1.0, dy - 1.0, dz - 1.0);
        double u = quintic(dx), v = quintic(dy), w = quintic(dz);
        double k0 = g000;
        double k1 = g100 - g000;
        double k2 = g010 - g000;
        double k3 = g001 - g000;
        double k4 = g000 + g110 - g100 - g010;
        double k5 = g000 + g101 - g100 - g001;
        double k6 = g000 + g011 - g010 - g001;
        double k7 = g100 + g010 + g001 + g111 - g000 - g110 - g101 - g011;
        double r = k0 + k1 * u + k2 * v + k3 * w + k4 * u * v + k5 * u * w + k6 * v * w + k7 * u * v * w;
        r = r * 1.1547005383792515; // 2/sqrt(3)
        if (r < -1.0) r = -1.0;
        if (r > 1.0) r = 1.0;
        return r;
    }
};

struct Texture { // texture.rs:5-7
    virtual ~Texture() {}
    virtual Color value(double u, double v, const Vec3& point) const = 0;
};
struct SolidColorTexture : Texture { // texture.rs:9-23
    Color color;
    explicit SolidColorTexture(const Color& c) : color(c) {}
    Color value(double, double, const Vec3&) const override { return color; }
};
struct CheckerTexture : Texture { // texture.rs:25-45
    std::shared_ptr<Texture> odd, even;
    CheckerTexture(std::shared_ptr<Texture> o, std::shared_ptr<Texture> e) : odd(o), even(e) {}
    Color value(double u, double v, const Vec3& point) const override {
        double sines = std::sin(10.0 * point.x()) * std::sin(10.0 * point.y()) * std::sin(10.0 * point.z());
        if (sines < 0.0) return odd->value(u, v, point);
        return even->value(u, v, point);
    }
};
struct PerlinNoiseTexture : Texture { // texture.rs:47-81
    Perlin noise;
    double scale;
    PerlinNoiseTexture(double scale_, uint64_t seed) : noise(seed), scale(scale_) {}
    double turbulance(const Vec3& point, int depth) const { // :60-72
        double accumulator = 0.0;
        Vec3 p = point;
        double weight = 1.0;
        for (int i = 0; i < depth; ++i) {
            accumulator += weight * noise.get(p.e);
            weight *= 0.5;
            p = Vec3(p.x() * 2.0, p.y() * 2.0, p.z() * 2.0);
        }
        return std::fabs(accumulator);
    }
    Color value(double, double, const Vec3& point) const override { // :76-80
        return Vec3(1.0, 1.0, 1.0) * 0.5 * (1.0 - std::sin(scale * point.z() + 10.0 * turbulance(point, 7)));
    }
};

// ---------------------------------------------------------------- material.rs / geometry.rs HitRecord
struct Material;
struct HitRecord { // geometry.rs:9-44
    double t;
    Vec3 point, normal;
    double u, v;
    bool front_face;
    const Material* material;
    int prim; // ours: rank of the primitive in depth-first order (for the parity tests)
    HitRecord() : t(0), u(0), v(0), front_face(false), material(nullptr), prim(-1) {}
    HitRecord(double t_, const Vec3& point_, const Ray& ray, const Vec3& outward_normal, double u_, double v_,
              const Material* m, int prim_)
        : t(t_), point(point_), u(u_), v(v_), material(m), prim(prim_) {
        front_face = ray.direction.dot(outward_normal) < 0.0; // :29
        normal = front_face ? outward_normal : -outward_normal; // :33-37
    }
};
struct Scatter { // material.rs:10-13
    Ray scattered_ray;
    Color attenuation;
};
struct Material { // material.rs:15-22
    virtual ~Material() {}
    virtual bool scatter(const Ray&, const HitRecord&, SampleRng&, Scatter&) const { return false; }
    virtual Color emits(const Ray&, const HitRecord&) const { return Color(); }
};
struct LambertianMaterial : Material { // material.rs:24-56
    std::shared_ptr<Texture> albedo;
    explicit LambertianMaterial(std::shared_ptr<Texture> a) : albedo(a) {}
    bool scatter(const Ray&, const HitRecord& hit, SampleRng& rng, Scatter& out) const override {
        Vec3 scatter_direction = hit.normal + random_on_unitsphere(rng);
        if (scatter_direction.near_zero()) scatter_direction = hit.normal; // (println dropped)
        out.scattered_ray = Ray(hit.point, scatter_direction);
        out.attenuation = albedo->value(hit.u, hit.v, hit.point);
        return true;
    }
};
struct MetalMaterial : Material { // material.rs:58-94
    std::shared_ptr<Texture> albedo;
    double fuzz;
    MetalMaterial(std::shared_ptr<Texture> a, double f) : albedo(a), fuzz(f) {}
    bool scatter(const Ray& ray_in, const HitRecord& hit, SampleRng& rng, Scatter& out) const override {
        Vec3 reflected_direction = reflect(unit_vector(ray_in.direction), hit.normal);
        if (reflected_direction.dot(hit.normal) > 0.0) {
            out.scattered_ray = Ray(hit.point, reflected_direction + fuzz * random_in_unitsphere(rng));
            out.attenuation = albedo->value(hit.u, hit.v, hit.point);
            return true;
        }
        return false;
    }
};
// f64::powi(n) lowers to compiler-rt __powidf2: square-and-multiply from the low bit.
inline double powi(double a, int n) {
    double r = 1.0;
    for (;;) {
        if (n & 1) r *= a;
        n /= 2;
        if (n == 0) break;
        a *= a;
    }
    return r;
}
inline double reflectance(double cosine, double refraction_index) { // material.rs:109-113
    double r0 = powi((1.0 - refraction_index) / (1.0 + refraction_index), 2);
    return r0 + (1.0 - r0) * powi(1.0 - cosine, 5);
}
struct DielectricMaterial : Material { // material.rs:96-144
    double index_of_refraction;
    explicit DielectricMaterial(double ior) : index_of_refraction(ior) {}
    bool scatter(const Ray& ray_in, const HitRecord& hit, SampleRng& rng, Scatter& out) const override {
        double refraction_ratio = hit.front_face ? 1.0 / index_of_refraction : index_of_refraction;
        Vec3 unit_direction = unit_vector(ray_in.direction);
        double cos_theta = std::fmin((-unit_direction).dot(hit.normal), 1.0);
        double sin_theta = std::sqrt(1.0 - cos_theta * cos_theta);
        Vec3 direction = unit_direction;
        if ((refraction_ratio * sin_theta > 1.0) || (reflectance(cos_theta, refraction_ratio) > rng.f64())) {
            direction = reflect(direction, hit.normal);
        } else {
            direction = refract(direction, hit.normal, refraction_ratio);
        }
        out.scattered_ray = Ray(hit.point, direction);
        out.attenuation = Color(1.0, 1.0, 1.0);
        return true;
    }
};
struct DiffuseLightMaterial : Material { // material.rs:146-167
    std::shared_ptr<Texture> emit;
    explicit DiffuseLightMaterial(std::shared_ptr<Texture> e) : emit(e) {}
    Color emits(const Ray&, const HitRecord& hit) const override {
        if (hit.front_face) return emit->value(hit.u, hit.v, hit.point);
        return Color();
    }
};

// ---------------------------------------------------------------- bvh.rs Aabb
struct Aabb {
    Vec3 minimum, maximum;
    Aabb() {}
    Aabb(const Vec3& a, const Vec3& b) : minimum(a), maximum(b) {}
    bool hit(const Ray& ray, double t_min, double t_max) const { // bvh.rs:18-35
        double found_t_min = t_min, found_t_max = t_max;
        for (int axis = 0; axis < 3; ++axis) {
            double inv_d = 1.0 / ray.direction.e[axis];
            double t0 = (minimum.e[axis] - ray.origin.e[axis]) * inv_d;
            double t1 = (maximum.e[axis] - ray.origin.e[axis]) * inv_d;
            if (inv_d < 0.0) std::swap(t0, t1);
            found_t_min = std::fmax(t0, found_t_min); // f64::max ignores a NaN operand, like fmax
            found_t_max = std::fmin(t1, found_t_max);
            if (found_t_max <= found_t_min) return false;
        }
        return true;
    }
    Aabb surrounding_box(const Aabb& o) const { // bvh.rs:37-50
        return Aabb(Vec3(std::fmin(minimum.x(), o.minimum.x()), std::fmin(minimum.y(), o.minimum.y()),
                         std::fmin(minimum.z(), o.minimum.z())),
                    Vec3(std::fmax(maximum.x(), o.maximum.x()), std::fmax(maximum.y(), o.maximum.y()),
                         std::fmax(maximum.z(), o.maximum.z())));
    }
};
// f64::total_cmp (bvh.rs:52-54): IEEE totalOrder via the sign-magnitude bit trick.
inline int64_t total_order_key(double v) {
    int64_t b;
    std::memcpy(&b, &v, 8);
    b ^= (int64_t)((uint64_t)(b >> 63) >> 1);
    return b;
}

// ---------------------------------------------------------------- geometry.rs
struct Hittable { // geometry.rs:46-49
    virtual ~Hittable() {}
    virtual bool hit(const Ray& ray, double t_min, double t_max, HitRecord& rec) const = 0;
    virtual Aabb bounding_box() const = 0;
    // ours: number primitives in depth-first order / list them for the brute-force scan / export the tree
    virtual void assign_ids(int& next, std::vector<const Hittable*>& flat) = 0;
    virtual int prim_id() const { return -1; }
};
typedef std::shared_ptr<Hittable> HittablePtr;

// geometry.rs:70-83: closest hit over a list, `t <= closest_so_far` accepted, so later wins ties.
inline bool list_hit(const std::vector<HittablePtr>& list, const Ray& ray, double t_min, double t_max, HitRecord& rec) {
    double closest_so_far = t_max;
    bool any = false;
    HitRecord tmp;
    for (const auto& object : list) {
        if (object->hit(ray, t_min, closest_so_far, tmp)) {
            closest_so_far = tmp.t;
            rec = tmp;
            any = true;
        }
    }
    return any;
}
inline Aabb list_bounding_box(const std::vector<HittablePtr>& list) { // geometry.rs:85-101
    Aabb out = list[0]->bounding_box();
    for (size_t i = 1; i < list.size(); ++i) out = out.surrounding_box(list[i]->bounding_box());
    return out;
}

struct Sphere : Hittable { // geometry.rs:104-171
    Vec3 center;
    double radius;
    std::shared_ptr<Material> material;
    int id = -1;
    Sphere(const Vec3& c, double r, std::shared_ptr<Material> m) : center(c), radius(r), material(m) {}
    static void get_sphere_uv(const Vec3& p, double& u, double& v) { // :120-128
        const double PI = 3.14159265358979323846;
        double theta = std::acos(-p.y());
        double phi = std::atan2(-p.z(), p.x()) + PI;
        u = phi / (2.0 * PI);
        v = theta / PI;
    }
    bool hit(const Ray& ray, double t_min, double t_max, HitRecord& rec) const override { // :132-163
        Vec3 oc = ray.origin - center;
        double a = ray.direction.len_squared();
        double half_b = ray.direction.dot(oc);
        double c = oc.len_squared() - radius * radius;
        double dis = half_b * half_b - a * c;
        if (dis < 0.0) return false;
        double root = (-half_b - std::sqrt(dis)) / a;
        if (root < t_min || root > t_max) {
            root = (-half_b + std::sqrt(dis)) / a;
            if (root < t_min || root > t_max) return false;
        }
        Vec3 point = ray.at(root);
        Vec3 outward_normal = (point - center) / radius;
        double u, v;
        get_sphere_uv(outward_normal, u, v);
        rec = HitRecord(root, point, ray, outward_normal, u, v, material.get(), id);
        return true;
    }
    Aabb bounding_box() const override { // :165-170
        double r = std::fabs(radius);
        return Aabb(center - Vec3(r, r, r), center + Vec3(r, r, r));
    }
    void assign_ids(int& next, std::vector<const Hittable*>& flat) override { id = next++; flat.push_back(this); }
    int prim_id() const override { return id; }
};

inline double signum(double v) { // f64::signum: 1.0 for +0.0 and positives, -1.0 for -0.0 and negatives, NaN for NaN
    if (v != v) return v;
    return std::signbit(v) ? -1.0 : 1.0;
}

// The three rectangle types (geometry.rs:182-369) differ only in which axis is the plane normal.
// AXIS = normal axis; A, B = the two in-plane axes in the reference's order.
template <int AXIS, int A, int B>
struct Rectangle : Hittable {
    Vec3 start, end;
    double direction;
    std::shared_ptr<Material> material;
    int id = -1;
    bool ok;
    Rectangle(const Vec3& s, const Vec3& e, double dir, std::shared_ptr<Material> m) : material(m) { // :191-206 etc.
        ok = (s.e[AXIS] == e.e[AXIS]);
        start.e[A] = std::fmin(s.e[A], e.e[A]);
        start.e[B] = std::fmin(s.e[B], e.e[B]);
        start.e[AXIS] = s.e[AXIS];
        end.e[A] = std::fmax(s.e[A], e.e[A]);
        end.e[B] = std::fmax(s.e[B], e.e[B]);
        end.e[AXIS] = s.e[AXIS];
        direction = signum(dir);
    }
    bool hit(const Ray& ray, double t_min, double t_max, HitRecord& rec) const override { // :210-232, :273-295, :336-358
        double t = (start.e[AXIS] - ray.origin.e[AXIS]) / ray.direction.e[AXIS];
        if (t < t_min || t > t_max) return false;
        double a = ray.origin.e[A] + t * ray.direction.e[A];
        double b = ray.origin.e[B] + t * ray.direction.e[B];
        if (a < start.e[A] || a > end.e[A] || b < start.e[B] || b > end.e[B]) return false;
        Vec3 n;
        n.e[AXIS] = direction;
        rec = HitRecord(t, ray.at(t), ray, n, (a - start.e[A]) / (end.e[A] - start.e[A]),
                        (b - start.e[B]) / (end.e[B] - start.e[B]), material.get(), id);
        return true;
    }
    Aabb bounding_box() const override { // :234-242, :297-305, :360-368
        Vec3 lo = start, hi = end;
        lo.e[AXIS] = start.e[AXIS] - 0.0001;
        hi.e[AXIS] = start.e[AXIS] + 0.0001;
        return Aabb(lo, hi);
    }
    void assign_ids(int& next, std::vector<const Hittable*>& flat) override { id = next++; flat.push_back(this); }
    int prim_id() const override { return id; }
};
typedef Rectangle<2, 0, 1> RectangleXY;
typedef Rectangle<1, 0, 2> RectangleXZ;
typedef Rectangle<0, 1, 2> RectangleYZ;

struct AABox : Hittable { // geometry.rs:371-464
    Vec3 minimum, maximum;
    std::vector<HittablePtr> sides;
    AABox(const Vec3& s, const Vec3& e, std::shared_ptr<Material> m) {
        minimum = Vec3(std::fmin(s.x(), e.x()), std::fmin(s.y(), e.y()), std::fmin(s.z(), e.z()));
        maximum = Vec3(std::fmax(s.x(), e.x()), std::fmax(s.y(), e.y()), std::fmax(s.z(), e.z()));
        const Vec3& lo = minimum;
        const Vec3& hi = maximum;
        sides.push_back(std::make_shared<RectangleXY>(Vec3(lo.x(), lo.y(), lo.z()), Vec3(hi.x(), hi.y(), lo.z()), -1.0, m));
        sides.push_back(std::make_shared<RectangleXY>(Vec3(lo.x(), lo.y(), hi.z()), Vec3(hi.x(), hi.y(), hi.z()), 1.0, m));
        sides.push_back(std::make_shared<RectangleXZ>(Vec3(lo.x(), lo.y(), lo.z()), Vec3(hi.x(), lo.y(), hi.z()), -1.0, m));
        sides.push_back(std::make_shared<RectangleXZ>(Vec3(lo.x(), hi.y(), lo.z()), Vec3(hi.x(), hi.y(), hi.z()), 1.0, m));
        sides.push_back(std::make_shared<RectangleYZ>(Vec3(lo.x(), lo.y(), lo.z()), Vec3(lo.x(), hi.y(), hi.z()), -1.0, m));
        sides.push_back(std::make_shared<RectangleYZ>(Vec3(hi.x(), lo.y(), lo.z()), Vec3(hi.x(), hi.y(), hi.z()), 1.0, m));
    }
    bool hit(const Ray& ray, double t_min, double t_max, HitRecord& rec) const override { // :457-459
        return list_hit(sides, ray, t_min, t_max, rec);
    }
    Aabb bounding_box() const override { return Aabb(minimum, maximum); } // :461-463
    void assign_ids(int& next, std::vector<const Hittable*>& flat) override {
        for (auto& s : sides) s->assign_ids(next, flat);
    }
};

struct Triangle : Hittable { // geometry.rs:466-589
    Vec3 point1, point2, point3, normal;
    std::shared_ptr<Material> material;
    int id = -1;
    Triangle(const Vec3& p1, const Vec3& p2, const Vec3& p3, const Vec3& n, std::shared_ptr<Material> m) // :486-500
        : point1(p1), point2(p2), point3(p3), normal(n), material(m) {}
    Triangle(const Vec3& p1, const Vec3& p2, const Vec3& p3, std::shared_ptr<Material> m) // :502-515
        : point1(p1), point2(p2), point3(p3), normal(unit_vector((p2 - p1).cross(p3 - p1))), material(m) {}
    bool hit(const Ray& ray, double t_min, double t_max, HitRecord& rec) const override { // :519-559
        Vec3 v0v1 = point2 - point1;
        Vec3 v0v2 = point3 - point1;
        Vec3 pvec = ray.direction.cross(v0v2);
        double det = v0v1.dot(pvec);
        if (det < 0.0001) return false;
        double inv_det = 1.0 / det;
        Vec3 tvec = ray.origin - point1;
        double u = tvec.dot(pvec) * inv_det;
        if (!(u >= 0.0 && u <= 1.0)) return false;
        Vec3 qvec = tvec.cross(v0v1);
        double v = ray.direction.dot(qvec) * inv_det;
        if (v < 0.0 || u + v > 1.0) return false;
        double t = v0v2.dot(qvec) * inv_det;
        if (!(t >= t_min && t <= t_max)) return false;
        rec = HitRecord(t, ray.at(t), ray, normal, 0.0, 0.0, material.get(), id);
        return true;
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
    void assign_ids(int& next, std::vector<const Hittable*>& flat) override { id = next++; flat.push_back(this); }
    int prim_id() const override { return id; }
};

// ---------------------------------------------------------------- bvh.rs BvhNode
struct BvhNode : Hittable {
    HittablePtr left, right;
    Aabb bbox;
    // bvh.rs:65-103.  The axis draw comes first, so two-element nodes are sorted too.
    BvhNode(std::vector<HittablePtr> objects, SplitMix64& rng) {
        uint32_t axis = rng.below(3);
        std::vector<int64_t> keys(objects.size());
        std::vector<size_t> order(objects.size());
        for (size_t i = 0; i < objects.size(); ++i) {
            keys[i] = total_order_key(objects[i]->bounding_box().minimum.e[axis]);
            order[i] = i;
        }
        std::stable_sort(order.begin(), order.end(), [&](size_t a, size_t b) { return keys[a] < keys[b]; });
        std::vector<HittablePtr> sorted(objects.size());
        for (size_t i = 0; i < objects.size(); ++i) sorted[i] = objects[order[i]];
        size_t len = sorted.size();
        if (len < 2) throw std::runtime_error("creating BVH node from empty object list"); // :76
        if (len == 2) {
            left = sorted[0];
            right = sorted[1];
        } else {
            size_t half_index = len / 2;
            std::vector<HittablePtr> left_list(sorted.begin(), sorted.begin() + half_index);
            std::vector<HittablePtr> right_list(sorted.begin() + half_index, sorted.end());
            left = left_list.size() == 1 ? left_list[0] : std::make_shared<BvhNode>(left_list, rng);
            right = right_list.size() == 1 ? right_list[0] : std::make_shared<BvhNode>(right_list, rng);
        }
        bbox = left->bounding_box().surrounding_box(right->bounding_box());
    }
    bool hit(const Ray& ray, double t_min, double t_max, HitRecord& rec) const override { // bvh.rs:107-121
        if (!bbox.hit(ray, t_min, t_max)) return false;
        HitRecord l, r;
        bool hl = left->hit(ray, t_min, t_max, l);
        double new_t_max = hl ? l.t : t_max;
        if (right->hit(ray, t_min, new_t_max, r)) {
            rec = r;
            return true;
        }
        if (hl) rec = l;
        return hl;
    }
    Aabb bounding_box() const override { return bbox; }
    void assign_ids(int& next, std::vector<const Hittable*>& flat) override {
        left->assign_ids(next, flat);
        right->assign_ids(next, flat);
    }
};

// ---------------------------------------------------------------- obj_model.rs
struct ObjModel : Hittable {
    std::shared_ptr<BvhNode> triangles;
    Vec3 minimum, maximum;
    bool hit(const Ray& ray, double t_min, double t_max, HitRecord& rec) const override { // :119-121
        return triangles->hit(ray, t_min, t_max, rec);
    }
    Aabb bounding_box() const override { return Aabb(minimum, maximum); } // :115-117
    void assign_ids(int& next, std::vector<const Hittable*>& flat) override { triangles->assign_ids(next, flat); }
};

// ---------------------------------------------------------------- camera.rs
struct Camera {
    Vec3 origin, lower_left_corner, horizontal, vertical, u, v;
    double lens_radius;
    Camera() : lens_radius(0) {}
    Camera(const Vec3& lookfrom, const Vec3& lookat, const Vec3& up, double vertical_fov, double aspect_ratio,
           double aperture, double focus_dist) { // camera.rs:14-44
        const double PI = 3.14159265358979323846;
        double h = std::tan((vertical_fov * (PI / 180.0)) / 2.0); // f64::to_radians = x * (PI/180)
        double viewport0 = aspect_ratio * 2.0 * h, viewport1 = 2.0 * h;
        Vec3 w = unit_vector(lookfrom - lookat);
        u = unit_vector(up.cross(w));
        v = w.cross(u);
        origin = lookfrom;
        horizontal = focus_dist * viewport0 * u;
        vertical = focus_dist * viewport1 * v;
        lower_left_corner = origin - horizontal / 2.0 - vertical / 2.0 - focus_dist * w;
        lens_radius = aperture / 2.0;
    }
    Ray ray_at(double s, double t, SampleRng& rng) const { // camera.rs:46-56
        Vec3 rd = lens_radius * random_in_unitdisk_xy(rng);
        Vec3 blur_offset = u * rd.x() + v * rd.y();
        return Ray(origin + blur_offset, lower_left_corner + s * horizontal + t * vertical - origin - blur_offset);
    }
};

// ---------------------------------------------------------------- ray.rs:20-41
// stream 0 of a sample feeds the pixel jitter and the lens; stream d+1 feeds the scatter at the
// d-th hit of the path (d = 0 for the camera ray's hit).
Color ray_color(const Ray& ray, const Hittable& world, const Color& background, uint32_t bounces_left, uint32_t depth,
                uint64_t seed, uint32_t pixel, uint32_t sample, uint64_t& rays) {
    if (bounces_left == 0) return Color();
    rays += 1; // one `hit` query: the unit of the Mrays/s metric
    HitRecord rec;
    if (world.hit(ray, 0.001, std::numeric_limits<double>::infinity(), rec)) {
        Color emitted = rec.material->emits(ray, rec);
        SampleRng rng(seed, pixel, sample, depth + 1u);
        Scatter sc;
        if (rec.material->scatter(ray, rec, rng, sc)) {
            return emitted + sc.attenuation * ray_color(sc.scattered_ray, world, background, bounces_left - 1, depth + 1,
                                                        seed, pixel, sample, rays);
        }
        return emitted;
    }
    return background;
}

// ---------------------------------------------------------------- scene.rs + custom worlds
struct OrcTextureDesc { // POD mirror used by custom scenes
    uint32_t type;      // 0 solid, 1 checker, 2 perlin
    uint32_t odd, even; // checker children (texture indices)
    uint32_t pad;
    double color[3];
    double scale;
    uint64_t perlin_seed;
};
struct OrcMaterialDesc {
    uint32_t type;    // 0 lambertian, 1 metal, 2 dielectric, 3 diffuse light
    uint32_t texture; // albedo / emit
    double param;     // fuzz / index of refraction
};
struct OrcObjectDesc {
    uint32_t kind;     // 0 sphere, 1 rectXY, 2 rectXZ, 3 rectYZ, 4 aabox, 5 triangle+normal, 6 triangle, 7 mesh
    uint32_t material; // (mesh: index of the mesh in the mesh list)
    double p[12];
};
struct OrcMeshDesc { // what tobj::load_obj would hand back for one model (obj_model.rs:53-55)
    const float* positions;
    uint64_t n_positions; // floats
    const uint32_t* indices;
    uint64_t n_indices;
    const float* normals;
    uint64_t n_normals; // floats, 0 = none
    int32_t material_id; // index into the MTL list, -1 = none
    int32_t pad;
};
struct OrcMtlDesc { // tobj::Material fields used by obj_model.rs:32-46
    int32_t illum; // -1 = absent
    float ni;
    float kd[3];
    float ns;
};

struct Scene {
    std::vector<std::shared_ptr<Texture>> textures;
    std::vector<std::shared_ptr<Material>> materials;
    std::shared_ptr<BvhNode> world;
    std::vector<const Hittable*> flat; // primitives in depth-first order
    std::string error;
};

std::shared_ptr<Material> lambertian_color(const Color& c) { // material.rs:33-37
    return std::make_shared<LambertianMaterial>(std::make_shared<SolidColorTexture>(c));
}
std::shared_ptr<Material> metal_color(const Color& c, double fuzz) { // material.rs:68-73
    return std::make_shared<MetalMaterial>(std::make_shared<SolidColorTexture>(c), fuzz);
}
std::shared_ptr<Material> light_color(const Color& c) { // material.rs:151-155
    return std::make_shared<DiffuseLightMaterial>(std::make_shared<SolidColorTexture>(c));
}
std::shared_ptr<Texture> checker_default() { // scene.rs:94-97, :215-218, :641-644
    return std::make_shared<CheckerTexture>(std::make_shared<SolidColorTexture>(Color(0.2, 0.3, 0.1)),
                                            std::make_shared<SolidColorTexture>(Color(0.9, 0.9, 0.9)));
}

// obj_model.rs:19-111 on tobj-shaped arrays.
std::shared_ptr<ObjModel> build_obj_model(const OrcMeshDesc* meshes, int n_meshes, const OrcMtlDesc* mtls, int n_mtls,
                                          SplitMix64& bvh_rng) {
    std::vector<std::shared_ptr<Material>> mapped;
    for (int i = 0; i < n_mtls; ++i) { // :28-50
        const OrcMtlDesc& m = mtls[i];
        if (m.illum == 7)
            mapped.push_back(std::make_shared<DielectricMaterial>((double)m.ni));
        else if (m.illum == 5)
            mapped.push_back(metal_color(Color((double)m.kd[0], (double)m.kd[1], (double)m.kd[2]), 1.0 / (double)m.ns));
        else
            mapped.push_back(lambertian_color(Color((double)m.kd[0], (double)m.kd[1], (double)m.kd[2])));
    }
    std::vector<HittablePtr> world;
    std::shared_ptr<Material> default_material = lambertian_color(Color(0.2, 0.7, 0.2)); // :82-84 (one instance, same values)
    for (int mi = 0; mi < n_meshes; ++mi) { // :53-100
        const OrcMeshDesc& mesh = meshes[mi];
        for (uint64_t tri = 0; tri < mesh.n_indices / 3; ++tri) {
            size_t i0 = mesh.indices[tri * 3], i1 = mesh.indices[tri * 3 + 1], i2 = mesh.indices[tri * 3 + 2];
            Vec3 v0((double)mesh.positions[i0 * 3], (double)mesh.positions[i0 * 3 + 1], (double)mesh.positions[i0 * 3 + 2]);
            Vec3 v1((double)mesh.positions[i1 * 3], (double)mesh.positions[i1 * 3 + 1], (double)mesh.positions[i1 * 3 + 2]);
            Vec3 v2((double)mesh.positions[i2 * 3], (double)mesh.positions[i2 * 3 + 1], (double)mesh.positions[i2 * 3 + 2]);
            std::shared_ptr<Material> material =
                mesh.material_id >= 0 ? mapped[mesh.material_id] : default_material;
            if (mesh.n_normals == 0) {
                world.push_back(std::make_shared<Triangle>(v0, v1, v2, material));
            } else { // :90-95 -- indexed by vertex 0's POSITION index, verbatim, not normalised
                Vec3 n((double)mesh.normals[i0 * 3], (double)mesh.normals[i0 * 3 + 1], (double)mesh.normals[i0 * 3 + 2]);
                world.push_back(std::make_shared<Triangle>(v0, v1, v2, n, material));
            }
        }
    }
    auto model = std::make_shared<ObjModel>();
    Aabb bb = list_bounding_box(world); // :102-104
    model->minimum = bb.minimum;
    model->maximum = bb.maximum;
    model->triangles = std::make_shared<BvhNode>(world, bvh_rng); // :107
    return model;
}

void finish_scene(Scene* s, std::vector<HittablePtr>& world, SplitMix64& bvh_rng) {
    s->world = std::make_shared<BvhNode>(world, bvh_rng);
    int next = 0;
    s->world->assign_ids(next, s->flat);
}

} // namespace

// ====================================================================== C interface for the tests
extern "C" {

struct OrcScene;

void orc_scene_free(OrcScene* s) { delete reinterpret_cast<Scene*>(s); }
const char* orc_scene_error(OrcScene* s) { return reinterpret_cast<Scene*>(s)->error.c_str(); }
int orc_scene_num_prims(OrcScene* s) { return (int)reinterpret_cast<Scene*>(s)->flat.size(); }

// One of scene.rs's six worlds.  Seeds: layout (SphereField draws, scene.rs:112-131), bvh (axis draws,
// bvh.rs:67; the ObjModel's nested tree draws first, as it is built inside get_world), perlin (texture.rs:55).
OrcScene* orc_scene_named(const char* name_c, uint64_t layout_seed, uint64_t bvh_seed, uint64_t perlin_seed,
                          const OrcMeshDesc* meshes, int n_meshes, const OrcMtlDesc* mtls, int n_mtls) {
    Scene* s = new Scene();
    std::string name(name_c);
    std::vector<HittablePtr> world;
    SplitMix64 bvh_rng(bvh_seed);
    try {
        if (name == "SphereField") { // scene.rs:91-170
            SplitMix64 rng(layout_seed);
            world.push_back(std::make_shared<Sphere>(Vec3(0.0, -1000.0, 0.0), 1000.0,
                                                     std::make_shared<LambertianMaterial>(checker_default())));
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
                        double r = rng.f64(), g = rng.f64(), bl = rng.f64(); // Color::random(), vec3.rs:25-31
                        material = lambertian_color(Color(r, g, bl));
                    } else if (pick < 0.8) {
                        double r = 0.5 + (1.0 - 0.5) * rng.f64(); // random_range(0.5, 1.0), vec3.rs:33-39
                        double g = 0.5 + (1.0 - 0.5) * rng.f64();
                        double bl = 0.5 + (1.0 - 0.5) * rng.f64();
                        double fuzz = rng.f64();
                        material = metal_color(Color(r, g, bl), fuzz);
                    } else {
                        material = std::make_shared<DielectricMaterial>(1.5);
                        is_glass = true;
                    }
                    if (is_glass && rng.f64() < 0.5) {
                        world.push_back(std::make_shared<Sphere>(center, radius, material));
                        world.push_back(std::make_shared<Sphere>(center, (radius * -1.0) + 0.02, material)); // radius.neg() + 0.02
                    } else {
                        world.push_back(std::make_shared<Sphere>(center, radius, material));
                    }
                }
            }
            auto big1 = std::make_shared<DielectricMaterial>(1.5);
            world.push_back(std::make_shared<Sphere>(Vec3(-4.0, 1.0, 0.0), 1.0, big1));
            world.push_back(std::make_shared<Sphere>(Vec3(-4.0, 1.0, 0.0), -0.95, big1));
            world.push_back(std::make_shared<Sphere>(Vec3(4.0, 1.0, 0.0), 1.0, std::make_shared<DielectricMaterial>(1.5)));
            world.push_back(std::make_shared<Sphere>(Vec3(0.0, 1.0, 0.0), 1.0, metal_color(Color(0.7, 0.6, 0.5), 0.0)));
        } else if (name == "TwoSpheres") { // scene.rs:212-235
            world.push_back(std::make_shared<Sphere>(Vec3(0.0, -10.0, 0.0), 10.0,
                                                     std::make_shared<LambertianMaterial>(checker_default())));
            world.push_back(std::make_shared<Sphere>(
                Vec3(0.0, 10.0, 0.0), 10.0,
                std::make_shared<LambertianMaterial>(std::make_shared<PerlinNoiseTexture>(4.0, perlin_seed))));
        } else if (name == "LightTest") { // scene.rs:277-325
            auto ground = std::make_shared<LambertianMaterial>(std::make_shared<PerlinNoiseTexture>(4.0, perlin_seed));
            world.push_back(std::make_shared<Sphere>(Vec3(0.0, -1000.0, 0.0), 1000.0, ground));
            world.push_back(std::make_shared<Sphere>(Vec3(0.0, 2.0, 0.0), 2.0, ground));
            auto light = light_color(Color(4.0, 4.0, 4.0));
            world.push_back(std::make_shared<RectangleXY>(Vec3(3.0, 1.0, -2.0), Vec3(5.0, 3.0, -2.0), 1.0, light));
            world.push_back(std::make_shared<RectangleXZ>(Vec3(-1.0, 6.0, -1.0), Vec3(1.0, 6.0, 1.0), -1.0, light));
            world.push_back(std::make_shared<RectangleYZ>(Vec3(-6.0, 1.0, -2.0), Vec3(-6.0, 3.0, 2.0), 1.0, light));
        } else if (name == "CornellBox" || name == "TriangleTest") { // scene.rs:367-464, :506-594
            auto red = lambertian_color(Color(0.65, 0.05, 0.05));
            auto white = lambertian_color(Color(0.73, 0.73, 0.73));
            auto green = lambertian_color(Color(0.12, 0.45, 0.15));
            auto light = light_color(Color(15.0, 15.0, 15.0));
            auto glass = std::make_shared<DielectricMaterial>(1.5);
            world.push_back(std::make_shared<RectangleYZ>(Vec3(555.0, 0.0, 0.0), Vec3(555.0, 555.0, 555.0), -1.0, green));
            world.push_back(std::make_shared<RectangleYZ>(Vec3(0.0, 0.0, 0.0), Vec3(0.0, 555.0, 555.0), 1.0, red));
            world.push_back(std::make_shared<RectangleXZ>(Vec3(0.0, 555.0, 0.0), Vec3(555.0, 555.0, 555.0), -1.0, white));
            world.push_back(std::make_shared<RectangleXZ>(Vec3(0.0, 0.0, 0.0), Vec3(555.0, 0.0, 555.0), 1.0, white));
            world.push_back(std::make_shared<RectangleXZ>(Vec3(213.0, 554.0, 227.0), Vec3(343.0, 554.0, 332.0), -1.0, light));
            world.push_back(std::make_shared<RectangleXY>(Vec3(0.0, 0.0, 555.0), Vec3(555.0, 555.0, 555.0), -1.0, white));
            if (name == "CornellBox") {
                world.push_back(std::make_shared<AABox>(Vec3(130.0, 0.0, 65.0), Vec3(295.0, 165.0, 230.0), white));
                world.push_back(std::make_shared<AABox>(Vec3(265.0, 0.0, 295.0), Vec3(430.0, 330.0, 460.0), white));
                world.push_back(std::make_shared<Sphere>(Vec3(212.5, 255.0, 147.5), 90.0, glass));
                world.push_back(std::make_shared<Sphere>(Vec3(347.5, 420.0, 377.5), 90.0, glass));
            } else {
                world.push_back(std::make_shared<Triangle>(Vec3(200.0, 100.0, 100.0), Vec3(300.0, 300.0, 500.0),
                                                           Vec3(400.0, 100.0, 100.0), glass));
                world.push_back(std::make_shared<Triangle>(Vec3(100.0, 300.0, 100.0), Vec3(150.0, 400.0, 250.0),
                                                           Vec3(100.0, 300.0, 400.0), white));
            }
        } else if (name == "ModelTest") { // scene.rs:638-657
            world.push_back(std::make_shared<Sphere>(Vec3(0.0, -1000.0, 0.0), 1000.0,
                                                     std::make_shared<LambertianMaterial>(checker_default())));
            world.push_back(build_obj_model(meshes, n_meshes, mtls, n_mtls, bvh_rng));
        } else {
            s->error = "unknown scene " + name;
            return reinterpret_cast<OrcScene*>(s);
        }
        finish_scene(s, world, bvh_rng);
    } catch (const std::exception& e) {
        s->error = e.what();
    }
    return reinterpret_cast<OrcScene*>(s);
}

// A world given as an object list (the reference's Vec<Box<dyn Hittable>> before BvhNode::new).
OrcScene* orc_scene_custom(const OrcObjectDesc* objs, int n_objs, const OrcMaterialDesc* mats, int n_mats,
                           const OrcTextureDesc* texs, int n_texs, const OrcMeshDesc* meshes, int n_meshes,
                           const OrcMtlDesc* mtls, int n_mtls, uint64_t bvh_seed) {
    Scene* s = new Scene();
    try {
        s->textures.resize(n_texs);
        // children may come after parents: build leaves first, then checkers until none is left
        int remaining = n_texs;
        for (int pass = 0; pass <= n_texs && remaining > 0; ++pass) {
            for (int i = 0; i < n_texs; ++i) {
                if (s->textures[i]) continue;
                const OrcTextureDesc& t = texs[i];
                if (t.type == 0) {
                    s->textures[i] = std::make_shared<SolidColorTexture>(Color(t.color[0], t.color[1], t.color[2]));
                } else if (t.type == 2) {
                    s->textures[i] = std::make_shared<PerlinNoiseTexture>(t.scale, t.perlin_seed);
                } else if (t.type == 1) {
                    if ((int)t.odd >= n_texs || (int)t.even >= n_texs) throw std::runtime_error("checker child out of range");
                    if (!s->textures[t.odd] || !s->textures[t.even]) continue;
                    s->textures[i] = std::make_shared<CheckerTexture>(s->textures[t.odd], s->textures[t.even]);
                } else {
                    throw std::runtime_error("bad texture type");
                }
                --remaining;
            }
        }
        if (remaining > 0) throw std::runtime_error("texture graph has a cycle");
        for (int i = 0; i < n_mats; ++i) {
            const OrcMaterialDesc& m = mats[i];
            if (m.type != 2 && (int)m.texture >= n_texs) throw std::runtime_error("material texture out of range");
            if (m.type == 0) s->materials.push_back(std::make_shared<LambertianMaterial>(s->textures[m.texture]));
            else if (m.type == 1) s->materials.push_back(std::make_shared<MetalMaterial>(s->textures[m.texture], m.param));
            else if (m.type == 2) s->materials.push_back(std::make_shared<DielectricMaterial>(m.param));
            else if (m.type == 3) s->materials.push_back(std::make_shared<DiffuseLightMaterial>(s->textures[m.texture]));
            else throw std::runtime_error("bad material type");
        }
        SplitMix64 bvh_rng(bvh_seed);
        std::vector<HittablePtr> world;
        for (int i = 0; i < n_objs; ++i) {
            const OrcObjectDesc& o = objs[i];
            const double* p = o.p;
            if (o.kind == 7) {
                (void)n_meshes;
                world.push_back(build_obj_model(meshes, n_meshes, mtls, n_mtls, bvh_rng));
                continue;
            }
            if ((int)o.material >= n_mats) throw std::runtime_error("object material out of range");
            auto m = s->materials[o.material];
            Vec3 a(p[0], p[1], p[2]), b(p[3], p[4], p[5]);
            switch (o.kind) {
                case 0: world.push_back(std::make_shared<Sphere>(a, p[3], m)); break;
                case 1: {
                    auto r = std::make_shared<RectangleXY>(a, b, p[6], m);
                    if (!r->ok) throw std::runtime_error("rectangle axis is not aligned"); // geometry.rs:178
                    world.push_back(r);
                    break;
                }
                case 2: {
                    auto r = std::make_shared<RectangleXZ>(a, b, p[6], m);
                    if (!r->ok) throw std::runtime_error("rectangle axis is not aligned");
                    world.push_back(r);
                    break;
                }
                case 3: {
                    auto r = std::make_shared<RectangleYZ>(a, b, p[6], m);
                    if (!r->ok) throw std::runtime_error("rectangle axis is not aligned");
                    world.push_back(r);
                    break;
                }
                case 4: world.push_back(std::make_shared<AABox>(a, b, m)); break;
                case 5: world.push_back(std::make_shared<Triangle>(a, b, Vec3(p[6], p[7], p[8]), Vec3(p[9], p[10], p[11]), m)); break;
                case 6: world.push_back(std::make_shared<Triangle>(a, b, Vec3(p[6], p[7], p[8]), m)); break;
                default: throw std::runtime_error("bad object kind");
            }
        }
        finish_scene(s, world, bvh_rng);
    } catch (const std::exception& e) {
        s->error = e.what();
    }
    return reinterpret_cast<OrcScene*>(s);
}

// Camera::new (camera.rs:14-44) -> 19 doubles: origin, lower_left_corner, horizontal, vertical, u, v, lens_radius.
void orc_camera_new(const double lookfrom[3], const double lookat[3], const double up[3], double vfov, double aspect,
                    double aperture, double focus_dist, double out[19]) {
    Camera c(Vec3(lookfrom[0], lookfrom[1], lookfrom[2]), Vec3(lookat[0], lookat[1], lookat[2]), Vec3(up[0], up[1], up[2]),
             vfov, aspect, aperture, focus_dist);
    const Vec3* v[6] = {&c.origin, &c.lower_left_corner, &c.horizontal, &c.vertical, &c.u, &c.v};
    for (int i = 0; i < 6; ++i)
        for (int k = 0; k < 3; ++k) out[i * 3 + k] = v[i]->e[k];
    out[18] = c.lens_radius;
}

// get_camera_at(t) of the six scenes (scene.rs:61-89, :188-210, :253-275, :343-365, :482-504, :614-636).
int orc_scene_camera(const char* name_c, double t, double aspect, double out[19]) {
    std::string name(name_c);
    const double PI = 3.14159265358979323846;
    double up[3] = {0.0, 1.0, 0.0};
    if (name == "SphereField") {
        double from[3] = {12.0 * std::cos(2.0 * PI * t), 1.0 + 2.0 * std::sin(PI * t), 12.0 * std::sin(2.0 * PI * t)};
        double at[3] = {0.0, 0.5, 0.0};
        orc_camera_new(from, at, up, 20.0, aspect, 0.1, 10.0, out);
    } else if (name == "TwoSpheres") {
        double from[3] = {13.0, 2.0, 3.0}, at[3] = {0.0, 0.0, 0.0};
        orc_camera_new(from, at, up, 20.0, aspect, 0.0, 10.0, out);
    } else if (name == "LightTest") {
        double from[3] = {26.0, 3.0, 6.0}, at[3] = {0.0, 2.0, 0.0};
        orc_camera_new(from, at, up, 20.0, aspect, 0.0, 10.0, out);
    } else if (name == "CornellBox" || name == "TriangleTest") {
        double from[3] = {278.0, 278.0, -800.0}, at[3] = {278.0, 278.0, 0.0};
        orc_camera_new(from, at, up, 40.0, aspect, 0.0, 10.0, out);
    } else if (name == "ModelTest") {
        double from[3] = {0.0, 2.5, -7.0}, at[3] = {0.0, 1.5, 0.0};
        orc_camera_new(from, at, up, 60.0, aspect, 0.0, 10.0, out);
    } else {
        return -1;
    }
    return 0;
}

static Camera camera_from(const double c[19]) {
    Camera cam;
    Vec3* v[6] = {&cam.origin, &cam.lower_left_corner, &cam.horizontal, &cam.vertical, &cam.u, &cam.v};
    for (int i = 0; i < 6; ++i)
        for (int k = 0; k < 3; ++k) v[i]->e[k] = c[i * 3 + k];
    cam.lens_radius = c[18];
    return cam;
}

// BvhNode::hit on a batch of rays.  prim = depth-first rank of the primitive hit, -1 = miss.
void orc_trace_batch(OrcScene* sc, const double* origins, const double* dirs, uint64_t n, double t_min, double t_max,
                     int32_t* prim, double* t, double* normal, uint8_t* front_face) {
    Scene* s = reinterpret_cast<Scene*>(sc);
    for (uint64_t i = 0; i < n; ++i) {
        Ray ray(Vec3(origins[i * 3], origins[i * 3 + 1], origins[i * 3 + 2]), Vec3(dirs[i * 3], dirs[i * 3 + 1], dirs[i * 3 + 2]));
        HitRecord rec;
        if (s->world->hit(ray, t_min, t_max, rec)) {
            prim[i] = rec.prim;
            t[i] = rec.t;
            for (int k = 0; k < 3; ++k) normal[i * 3 + k] = rec.normal.e[k];
            front_face[i] = rec.front_face ? 1 : 0;
        } else {
            prim[i] = -1;
            t[i] = 0.0;
            for (int k = 0; k < 3; ++k) normal[i * 3 + k] = 0.0;
            front_face[i] = 0;
        }
    }
}

// The same query with no tree: the list scan of geometry.rs:70-83 over all primitives in depth-first order.
// Equal to the BVH answer unless a box test prunes a primitive whose t ties the best so far.
void orc_trace_batch_brute(OrcScene* sc, const double* origins, const double* dirs, uint64_t n, double t_min, double t_max,
                           int32_t* prim, double* t) {
    Scene* s = reinterpret_cast<Scene*>(sc);
    for (uint64_t i = 0; i < n; ++i) {
        Ray ray(Vec3(origins[i * 3], origins[i * 3 + 1], origins[i * 3 + 2]), Vec3(dirs[i * 3], dirs[i * 3 + 1], dirs[i * 3 + 2]));
        double closest = t_max;
        int best = -1;
        HitRecord rec;
        for (const Hittable* h : s->flat) {
            if (h->hit(ray, t_min, closest, rec)) {
                closest = rec.t;
                best = rec.prim;
            }
        }
        prim[i] = best;
        t[i] = best >= 0 ? closest : 0.0;
    }
}

// Tree in depth-first pre-order, for comparison with the product's flattening.
// Each entry: kind (0 node, 1 primitive), for nodes the 6 box doubles, for primitives the rank.
// An AABox contributes its 6 sides, an ObjModel its nested tree (obj_model.rs:119-121).
static void export_rec(const Hittable* h, std::vector<int32_t>& kinds, std::vector<double>& boxes) {
    if (const BvhNode* n = dynamic_cast<const BvhNode*>(h)) {
        kinds.push_back(-1);
        boxes.insert(boxes.end(), n->bbox.minimum.e, n->bbox.minimum.e + 3);
        boxes.insert(boxes.end(), n->bbox.maximum.e, n->bbox.maximum.e + 3);
        export_rec(n->left.get(), kinds, boxes);
        export_rec(n->right.get(), kinds, boxes);
    } else if (const ObjModel* m = dynamic_cast<const ObjModel*>(h)) {
        export_rec(m->triangles.get(), kinds, boxes);
    } else if (const AABox* b = dynamic_cast<const AABox*>(h)) {
        for (const auto& sd : b->sides) export_rec(sd.get(), kinds, boxes);
    } else {
        kinds.push_back(h->prim_id());
    }
}
// Returns the number of entries; fills up to cap of them.  entries[i] = -1 for a node (its box is the next 6
// doubles of boxes, in order of appearance) or the primitive's rank.
int orc_tree_export(OrcScene* sc, int32_t* entries, int cap_entries, double* boxes, int cap_boxes) {
    Scene* s = reinterpret_cast<Scene*>(sc);
    std::vector<int32_t> kinds;
    std::vector<double> bx;
    export_rec(s->world.get(), kinds, bx);
    for (int i = 0; i < (int)kinds.size() && i < cap_entries; ++i) entries[i] = kinds[i];
    for (int i = 0; i < (int)bx.size() && i < cap_boxes; ++i) boxes[i] = bx[i];
    return (int)kinds.size();
}

// renderer.rs:14-36 for samples [sample_begin, sample_begin+sample_count) of every pixel.
// accum (w*h*3 doubles, rows top-first as the reference's output order) receives the SUM of the samples' colours;
// rays receives the number of hit queries.  threads <= 0: all cores (the reference uses rayon's global pool).
void orc_render_accumulate(OrcScene* sc, const double cam[19], uint32_t width, uint32_t height, const double bg[3],
                           uint32_t sample_begin, uint32_t sample_count, uint32_t max_bounces, uint64_t seed,
                           double* accum, uint64_t* rays, int threads) {
    Scene* s = reinterpret_cast<Scene*>(sc);
    Camera camera = camera_from(cam);
    Color background(bg[0], bg[1], bg[2]);
    int nthreads = threads > 0 ? threads : (int)std::thread::hardware_concurrency();
    if (nthreads < 1) nthreads = 1;
    std::atomic<uint32_t> next_row(0);
    std::atomic<uint64_t> total_rays(0);
    auto worker = [&]() {
        uint64_t my_rays = 0;
        for (;;) {
            uint32_t row = next_row.fetch_add(1); // row 0 = top of the image = reference y = height-1
            if (row >= height) break;
            uint32_t y = height - 1 - row;
            for (uint32_t x = 0; x < width; ++x) {
                uint32_t pixel = row * width + x;
                Color color_sampling;
                for (uint32_t sidx = sample_begin; sidx < sample_begin + sample_count; ++sidx) {
                    SampleRng rng(seed, pixel, sidx, 0);
                    double u = ((double)x + rng.f64()) / ((double)width - 1.0);  // renderer.rs:23
                    double v = ((double)y + rng.f64()) / ((double)height - 1.0); // renderer.rs:24
                    Ray ray = camera.ray_at(u, v, rng);
                    Color c = ray_color(ray, *s->world, background, max_bounces, 0, seed, pixel, sidx, my_rays);
                    color_sampling = color_sampling + c; // AddAssign, vec3.rs:138-144
                }
                for (int k = 0; k < 3; ++k) accum[(size_t)pixel * 3 + k] = color_sampling.e[k];
            }
        }
        total_rays.fetch_add(my_rays);
    };
    std::vector<std::thread> pool;
    for (int i = 1; i < nthreads; ++i) pool.emplace_back(worker);
    worker();
    for (auto& th : pool) th.join();
    *rays = total_rays.load();
}

// renderer.rs:30-33 + vec3.rs:278-290 on the accumulated sums.
void orc_resolve(const double* accum, uint32_t width, uint32_t height, uint32_t spp, uint8_t* rgb) {
    size_t n = (size_t)width * height * 3;
    for (size_t i = 0; i < n; ++i) rgb[i] = to_u8(std::sqrt(accum[i] / (double)spp));
}

// ---- probes of the small functions, for the known-answer tests
void orc_vec3_kat(double out[16]) { // vec3.rs:261-273
    Vec3 a(1.0, 0.0, 0.0), b(0.0, 1.0, 0.0), c(1.0, 1.0, 1.0), d(0.0, 10.0, 0.0);
    Vec3 r0 = a + b, r1 = a - b, r2 = -c, r4 = a.cross(b), r5 = unit_vector(d);
    double r3 = (a + b).dot(c);
    const Vec3* v[5] = {&r0, &r1, &r2, &r4, &r5};
    for (int i = 0; i < 5; ++i)
        for (int k = 0; k < 3; ++k) out[i * 3 + k] = v[i]->e[k];
    out[15] = r3;
}
void orc_reflect(const double v[3], const double n[3], double out[3]) {
    Vec3 r = reflect(Vec3(v[0], v[1], v[2]), Vec3(n[0], n[1], n[2]));
    for (int k = 0; k < 3; ++k) out[k] = r.e[k];
}
void orc_refract(const double v[3], const double n[3], double eta, double out[3]) {
    Vec3 r = refract(Vec3(v[0], v[1], v[2]), Vec3(n[0], n[1], n[2]), eta);
    for (int k = 0; k < 3; ++k) out[k] = r.e[k];
}
double orc_reflectance(double cosine, double ri) { return reflectance(cosine, ri); }
uint8_t orc_to_u8(double c) { return to_u8(c); }
void orc_philox(const uint32_t ctr[4], const uint32_t key[2], uint32_t out[4]) { philox4x32_10(ctr, key, out); }
void orc_sample_draws(uint64_t seed, uint32_t pixel, uint32_t sample, uint32_t stream, int n, double* out) {
    SampleRng rng(seed, pixel, sample, stream);
    for (int i = 0; i < n; ++i) out[i] = rng.f64();
}
void orc_splitmix(uint64_t seed, int n, uint64_t* out) {
    SplitMix64 r(seed);
    for (int i = 0; i < n; ++i) out[i] = r.next();
}
void orc_perlin_table(uint64_t seed, uint8_t out[256]) {
    Perlin p(seed);
    std::memcpy(out, p.perm, 256);
}
double orc_perlin_get(uint64_t seed, const double p[3]) {
    Perlin pn(seed);
    return pn.get(p);
}
// texture values: kind 0 checker(default colours), 1 marble(scale, seed)
void orc_texture_value(int kind, double scale, uint64_t seed, const double* points, int n, double* out) {
    std::shared_ptr<Texture> t;
    if (kind == 0) t = checker_default();
    else t = std::make_shared<PerlinNoiseTexture>(scale, seed);
    for (int i = 0; i < n; ++i) {
        Color c = t->value(0.0, 0.0, Vec3(points[i * 3], points[i * 3 + 1], points[i * 3 + 2]));
        for (int k = 0; k < 3; ++k) out[i * 3 + k] = c.e[k];
    }
}
// One scatter event, for the material known answers.  mat_type/param as OrcMaterialDesc with a solid albedo.
// Returns 1 if scattered; out = direction[3], attenuation[3], emitted[3].
int orc_scatter_probe(uint32_t mat_type, double param, const double albedo[3], const double ray_o[3], const double ray_d[3],
                      double t, const double outward_normal[3], uint64_t seed, uint32_t pixel, uint32_t sample,
                      uint32_t stream, double out[9]) {
    std::shared_ptr<Texture> tex = std::make_shared<SolidColorTexture>(Color(albedo[0], albedo[1], albedo[2]));
    std::shared_ptr<Material> m;
    if (mat_type == 0) m = std::make_shared<LambertianMaterial>(tex);
    else if (mat_type == 1) m = std::make_shared<MetalMaterial>(tex, param);
    else if (mat_type == 2) m = std::make_shared<DielectricMaterial>(param);
    else m = std::make_shared<DiffuseLightMaterial>(tex);
    Ray ray(Vec3(ray_o[0], ray_o[1], ray_o[2]), Vec3(ray_d[0], ray_d[1], ray_d[2]));
    HitRecord rec(t, ray.at(t), ray, Vec3(outward_normal[0], outward_normal[1], outward_normal[2]), 0.0, 0.0, m.get(), 0);
    SampleRng rng(seed, pixel, sample, stream);
    Scatter sc;
    Color e = m->emits(ray, rec);
    bool ok = m->scatter(ray, rec, rng, sc);
    for (int k = 0; k < 3; ++k) {
        out[k] = ok ? sc.scattered_ray.direction.e[k] : 0.0;
        out[3 + k] = ok ? sc.attenuation.e[k] : 0.0;
        out[6 + k] = e.e[k];
    }
    return ok ? 1 : 0;
}

} // extern "C"
