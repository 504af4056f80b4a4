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

// ---- Split references for thin diagonal triangles.
// A long thin triangle that runs diagonally through space has a box hundreds of times larger than itself; where many of
// them meet (the pole of a latitude-longitude mesh: a thousand slivers sharing a vertex) a ray near that place passes
// through hundreds of boxes and runs as many triangle tests -- and such a ray is the slowest of its launch.  The guide
// tree may reference a primitive more than once, each time with the box of a PIECE of it: the triangle is cut by
// axis-aligned planes (the longest side of the piece's box, at its middle) until a piece's box is no larger than
// `ratio` times the piece's own projected areas, or `max_depth` cuts.  The pieces cover the triangle (cut points are
// computed in f64, rounding 1e-16 relative, against a padding of 1e-5 of the triangle's largest coordinate), so the
// premise of the ordered walk stands: a test that reports a hit at t is crossed inside one of the primitive's padded
// boxes at t.  The walk meets such a primitive several times; pt_device.h OrderedWalk::consider ignores the repeats.
struct SplitPoly { double p[12][3]; int n; };
inline double poly_half_area_ideal(const SplitPoly& q) { // |Nx| + |Ny| + |Nz|, N = sum p_i x p_(i+1) (twice the area vector)
    double nx = 0, ny = 0, nz = 0;
    for (int i = 0; i < q.n; ++i) {
        const double* a = q.p[i];
        const double* b = q.p[(i + 1) % q.n];
        // relative to vertex 0: the sum is translation-invariant, the rounding is not
        const double ax = a[0] - q.p[0][0], ay = a[1] - q.p[0][1], az = a[2] - q.p[0][2];
        const double bx = b[0] - q.p[0][0], by = b[1] - q.p[0][1], bz = b[2] - q.p[0][2];
        nx += ay * bz - az * by; ny += az * bx - ax * bz; nz += ax * by - ay * bx;
    }
    return std::fabs(nx) + std::fabs(ny) + std::fabs(nz);
}
inline void clip_poly(const SplitPoly& in, int axis, double plane, bool keep_below, SplitPoly& out) { // Sutherland-Hodgman, one plane
    out.n = 0;
    for (int i = 0; i < in.n; ++i) {
        const double* a = in.p[i];
        const double* b = in.p[(i + 1) % in.n];
        const bool ia = keep_below ? a[axis] <= plane : a[axis] >= plane, ib = keep_below ? b[axis] <= plane : b[axis] >= plane;
        if (ia && out.n < 12) { for (int k = 0; k < 3; ++k) out.p[out.n][k] = a[k]; out.n += 1; }
        if (ia != ib && out.n < 12) {
            const double s = (plane - a[axis]) / (b[axis] - a[axis]);
            for (int k = 0; k < 3; ++k) out.p[out.n][k] = a[k] + (b[k] - a[k]) * s;
            out.p[out.n][axis] = plane;
            out.n += 1;
        }
    }
}
// Appends the boxes (unpadded) of the pieces of `poly` to `out` (6 doubles each: lo.xyz, hi.xyz); returns how many.
inline size_t split_pieces(const SplitPoly& poly, double ratio, int max_depth, std::vector<double>* out) {
    double lo[3] = {INFINITY, INFINITY, INFINITY}, hi[3] = {-INFINITY, -INFINITY, -INFINITY};
    for (int i = 0; i < poly.n; ++i)
        for (int k = 0; k < 3; ++k) { lo[k] = std::min(lo[k], poly.p[i][k]); hi[k] = std::max(hi[k], poly.p[i][k]); }
    const double x = hi[0] - lo[0], y = hi[1] - lo[1], z = hi[2] - lo[2];
    const double half = x * y + y * z + z * x;
    int axis = 0;
    if (y > x) axis = 1;
    if (z > (axis ? y : x)) axis = 2;
    const double mid = 0.5 * (lo[axis] + hi[axis]);
    const bool cut = max_depth > 0 && poly.n >= 3 && poly.n <= 10 && std::isfinite(half) && half > ratio * poly_half_area_ideal(poly) &&
                     lo[axis] < mid && mid < hi[axis];
    if (!cut) {
        if (out) { out->insert(out->end(), lo, lo + 3); out->insert(out->end(), hi, hi + 3); }
        return 1;
    }
    SplitPoly below, above;
    clip_poly(poly, axis, mid, true, below);
    clip_poly(poly, axis, mid, false, above);
    size_t n = 0;
    if (below.n >= 3) n += split_pieces(below, ratio, max_depth - 1, out);
    if (above.n >= 3) n += split_pieces(above, ratio, max_depth - 1, out);
    if (n == 0) { // (cannot happen: a cut through the inside of the box leaves both sides a polygon) keep the whole
        if (out) { out->insert(out->end(), lo, lo + 3); out->insert(out->end(), hi, hi + 3); }
        return 1;
    }
    return n;
}
inline SplitPoly split_poly_of(const double corners[9]) {
    SplitPoly q;
    q.n = 3;
    for (int i = 0; i < 3; ++i)
        for (int k = 0; k < 3; ++k) q.p[i][k] = corners[3 * i + k];
    return q;
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
