// vec3.hpp -- host mirror of the reference's Vec3 / Color (src/vec3.rs:9-291).
// Host side only builds scenes and cameras; the per-sample samplers (vec3.rs:25-67) live on the device
// (csrc/device/pt_device.h).  Compiled with -ffp-contract=off: one IEEE operation per operator, as in Rust.
#pragma once
#include <cmath>
#include <cstdint>

namespace pt {

struct Vec3 {
    double e[3];
    Vec3() : e{0.0, 0.0, 0.0} {}                       // vec3.rs:14-18
    Vec3(double x, double y, double z) : e{x, y, z} {} // vec3.rs:21-23
    double x() const { return e[0]; }
    double y() const { return e[1]; }
    double z() const { return e[2]; }
    double len_squared() const { return x() * x() + y() * y() + z() * z(); }
    double len() const { return std::sqrt(len_squared()); }
    double dot(const Vec3& r) const { return x() * r.x() + y() * r.y() + z() * r.z(); }
    Vec3 cross(const Vec3& r) const {
        return Vec3(y() * r.z() - z() * r.y(), z() * r.x() - x() * r.z(), x() * r.y() - y() * r.x());
    }
    Vec3 unit_vector() const { return Vec3(x() / len(), y() / len(), z() / len()); }
    double operator[](int i) const { return e[i]; }
    double& operator[](int i) { return e[i]; }
};
inline Vec3 operator+(const Vec3& a, const Vec3& b) { return Vec3(a.x() + b.x(), a.y() + b.y(), a.z() + b.z()); }
inline Vec3 operator-(const Vec3& a, const Vec3& b) { return Vec3(a.x() - b.x(), a.y() - b.y(), a.z() - b.z()); }
inline Vec3 operator*(const Vec3& a, double t) { return Vec3(a.x() * t, a.y() * t, a.z() * t); }
inline Vec3 operator*(double t, const Vec3& a) { return Vec3(a.x() * t, a.y() * t, a.z() * t); }
inline Vec3 operator/(const Vec3& a, double t) { return Vec3(a.x() / t, a.y() / t, a.z() / t); }
inline Vec3 operator-(const Vec3& a) { return Vec3(a.x() * -1.0, a.y() * -1.0, a.z() * -1.0); }

typedef Vec3 Color;

// Build-time random numbers.  The reference draws them from rand::thread_rng (OS-seeded, never
// reproducible: bvh.rs:67, scene.rs:112-131, texture.rs:55); here every such draw comes from a seeded
// SplitMix64 so a world can be rebuilt identically on every GPU and by the test oracle.
struct BuildRng {
    uint64_t s;
    explicit BuildRng(uint64_t seed) : s(seed) {}
    uint64_t next() {
        s += 0x9E3779B97F4A7C15ull;
        uint64_t z = s;
        z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
        z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
        return z ^ (z >> 31);
    }
    double f64() { return (double)(next() >> 11) * (1.0 / 9007199254740992.0); }            // rand::random::<f64>()
    uint32_t below(uint32_t n) { return (uint32_t)(((next() >> 32) * (uint64_t)n) >> 32); } // gen_range(0..n)
    double range(double lo, double hi) { return lo + (hi - lo) * f64(); }                   // gen_range(lo..hi)
};

} // namespace pt
