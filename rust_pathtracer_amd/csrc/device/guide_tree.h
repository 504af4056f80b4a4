// guide_tree.h -- host side: a surface-area-ordered binary tree over the leaves of the record stream.
//
// The reference's tree (bvh.rs:65-103: random axis, median split, children visited left before right) decides WHICH of
// two equally near primitives a ray reports, so the literal walk over it stays the definition of a hit (pt_device.h
// StreamWalk).  It culls poorly, though.  The guide tree is only used to find the nearest hit quickly; the ordered
// kernel (pt_device.h OrderedWalk) then proves that the reference's walk returns the same primitive -- or runs the
// literal walk for that ray.  Nothing here has a counterpart in the reference.
//
// Node = 64 bytes, laid out like a stream record so that one fetch routine serves both:
//   u32 child[2]            PT_GUIDE_LEAF | byte offset of a primitive record in the stream, or a node index
//   f32 box[2][6]           the children's boxes as (lo, hi) pairs per axis -- lo.x, hi.x, lo.y, hi.y, lo.z, hi.z: one packed
//                           f32 operation handles both planes of an axis -- rounded outwards, padded (see pad())
//   u32 unused[2]
#pragma once
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstring>
#include <vector>

#define PT_GUIDE_LEAF 0x80000000u

struct GuideLeaf {
    double lo[3], hi[3];
    uint32_t offset; // of its record in the stream
};

struct GuideNode {
    uint32_t child[2];
    float box[2][6];
    uint32_t unused[2];
};
static_assert(sizeof(GuideNode) == 64, "GuideNode is one 64-byte fetch");

namespace guide {

struct Box {
    double lo[3], hi[3];
    void clear() {
        for (int k = 0; k < 3; ++k) { lo[k] = INFINITY; hi[k] = -INFINITY; }
    }
    void grow(const double* l, const double* h) {
        for (int k = 0; k < 3; ++k) { lo[k] = std::min(lo[k], l[k]); hi[k] = std::max(hi[k], h[k]); }
    }
    double half_area() const {
        const double x = hi[0] - lo[0], y = hi[1] - lo[1], z = hi[2] - lo[2];
        return x * y + y * z + z * x;
    }
};

// Pads a leaf's box far beyond any rounding in the primitive tests (1e-5 of its largest coordinate against 1e-16), so
// that a primitive test that reports a hit at t implies the padded box is crossed at t.
inline void pad(GuideLeaf& l) {
    double m = 0.0;
    for (int k = 0; k < 3; ++k) m = std::max(m, std::max(std::fabs(l.lo[k]), std::fabs(l.hi[k])));
    const double eps = 1e-5 * m + 1e-30;
    for (int k = 0; k < 3; ++k) { l.lo[k] -= eps; l.hi[k] += eps; }
}

inline float round_down(double v) {
    float f = (float)v;
    return ((double)f > v) ? std::nextafterf(f, -INFINITY) : f;
}
inline float round_up(double v) {
    float f = (float)v;
    return ((double)f < v) ? std::nextafterf(f, INFINITY) : f;
}

// Whether a tree can be built over these leaves at all (by either builder): at least two, every box finite with lo <= hi and
// within what survives the rounding to f32, every stream offset within 31 bits.
inline bool usable(const std::vector<GuideLeaf>& leaves) {
    if (leaves.size() < 2) return false;
    for (const GuideLeaf& l : leaves) {
        if (l.offset & PT_GUIDE_LEAF) return false;
        for (int k = 0; k < 3; ++k)
            if (!(std::isfinite(l.lo[k]) && std::isfinite(l.hi[k]) && l.lo[k] <= l.hi[k])) return false;
        for (int k = 0; k < 3; ++k)
            if (std::fabs(l.lo[k]) > 1e37 || std::fabs(l.hi[k]) > 1e37) return false; // must survive the float rounding
    }
    return true;
}

// The host builder: binned surface-area splits.  Returns false (no tree) when the leaves are not usable().
inline bool build(std::vector<GuideLeaf>& leaves, std::vector<GuideNode>& nodes) {
    nodes.clear();
    const size_t n = leaves.size();
    if (!usable(leaves)) return false;
    nodes.reserve(n - 1);
    std::vector<uint32_t> idx(n);
    for (size_t i = 0; i < n; ++i) idx[i] = (uint32_t)i;
    std::vector<double> cen(n * 3);
    for (size_t i = 0; i < n; ++i)
        for (int k = 0; k < 3; ++k) cen[i * 3 + k] = 0.5 * (leaves[i].lo[k] + leaves[i].hi[k]);

    struct Task { uint32_t begin, end, parent, slot, depth; };
    std::vector<Task> todo;
    todo.push_back({0u, (uint32_t)n, 0xFFFFFFFFu, 0u, 0u});
    constexpr int BINS = 16;
    while (!todo.empty()) {
        const Task t = todo.back();
        todo.pop_back();
        Box whole;
        whole.clear();
        for (uint32_t i = t.begin; i < t.end; ++i) whole.grow(leaves[idx[i]].lo, leaves[idx[i]].hi);
        uint32_t ref;
        if (t.end - t.begin == 1) {
            ref = PT_GUIDE_LEAF | leaves[idx[t.begin]].offset;
        } else {
            ref = (uint32_t)nodes.size();
            nodes.emplace_back();
            std::memset(&nodes.back(), 0, sizeof(GuideNode));
            // binned surface-area split over the three axes; a median split when that fails or the tree gets deep
            double clo[3] = {INFINITY, INFINITY, INFINITY}, chi[3] = {-INFINITY, -INFINITY, -INFINITY};
            for (uint32_t i = t.begin; i < t.end; ++i)
                for (int k = 0; k < 3; ++k) {
                    clo[k] = std::min(clo[k], cen[(size_t)idx[i] * 3 + k]);
                    chi[k] = std::max(chi[k], cen[(size_t)idx[i] * 3 + k]);
                }
            int best_axis = -1, best_bin = -1;
            double best_cost = INFINITY;
            if (t.depth < 40 && t.end - t.begin > 2) {
                for (int axis = 0; axis < 3; ++axis) {
                    const double ext = chi[axis] - clo[axis];
                    if (!(ext > 0.0)) continue;
                    Box bb[BINS];
                    uint32_t cnt[BINS] = {0};
                    for (auto& b : bb) b.clear();
                    const double scale = BINS / ext;
                    for (uint32_t i = t.begin; i < t.end; ++i) {
                        int b = (int)((cen[(size_t)idx[i] * 3 + axis] - clo[axis]) * scale);
                        b = std::min(std::max(b, 0), BINS - 1);
                        bb[b].grow(leaves[idx[i]].lo, leaves[idx[i]].hi);
                        cnt[b] += 1;
                    }
                    double right_area[BINS];
                    uint32_t right_cnt[BINS];
                    Box acc;
                    acc.clear();
                    uint32_t c = 0;
                    for (int b = BINS - 1; b > 0; --b) {
                        if (cnt[b]) acc.grow(bb[b].lo, bb[b].hi);
                        c += cnt[b];
                        right_area[b] = c ? acc.half_area() : 0.0;
                        right_cnt[b] = c;
                    }
                    acc.clear();
                    c = 0;
                    for (int b = 0; b + 1 < BINS; ++b) { // split after bin b
                        if (cnt[b]) acc.grow(bb[b].lo, bb[b].hi);
                        c += cnt[b];
                        if (c == 0 || right_cnt[b + 1] == 0) continue;
                        const double cost = acc.half_area() * c + right_area[b + 1] * right_cnt[b + 1];
                        if (cost < best_cost) { best_cost = cost; best_axis = axis; best_bin = b; }
                    }
                }
            }
            uint32_t mid;
            if (best_axis >= 0) {
                const double scale = BINS / (chi[best_axis] - clo[best_axis]);
                auto in_left = [&](uint32_t i) {
                    int b = (int)((cen[(size_t)i * 3 + best_axis] - clo[best_axis]) * scale);
                    b = std::min(std::max(b, 0), BINS - 1);
                    return b <= best_bin;
                };
                mid = (uint32_t)(std::partition(idx.begin() + t.begin, idx.begin() + t.end, in_left) - idx.begin());
            } else {
                int axis = 0;
                for (int k = 1; k < 3; ++k)
                    if (chi[k] - clo[k] > chi[axis] - clo[axis]) axis = k;
                mid = t.begin + (t.end - t.begin) / 2;
                std::nth_element(idx.begin() + t.begin, idx.begin() + mid, idx.begin() + t.end,
                                 [&](uint32_t a, uint32_t b) { return cen[(size_t)a * 3 + axis] < cen[(size_t)b * 3 + axis]; });
            }
            todo.push_back({mid, t.end, ref, 1u, t.depth + 1u});
            todo.push_back({t.begin, mid, ref, 0u, t.depth + 1u}); // left next: a subtree stays together in memory
        }
        if (t.parent != 0xFFFFFFFFu) {
            GuideNode& p = nodes[t.parent];
            p.child[t.slot] = ref;
            for (int k = 0; k < 3; ++k) {
                p.box[t.slot][2 * k] = round_down(whole.lo[k]);
                p.box[t.slot][2 * k + 1] = round_up(whole.hi[k]);
            }
        }
    }
    return true;
}

} // namespace guide
