// guide_build.hip -- the ordered walk's guide tree built ON THE DEVICE (option "guide_build" = 1): for worlds that are rebuilt
// every frame (BASELINE.json configs[4]) the host's binned surface-area build -- 0.45 s for a million triangles -- is the
// wrong tool.  Nothing here has a counterpart in the reference (its tree, bvh.rs:65-103, stays the definition of a hit; the
// guide tree only decides how quickly the nearest hit is found, pt_device.h OrderedWalk), and results do not depend on which
// builder made the tree: the walk's proof needs nothing of it but boxes that contain their primitives.
//
// A linear BVH (Lauterbach et al. 2009) with the hierarchy of Karras 2012:
//   1. one 63-bit Morton key per leaf from the centre of its (padded, outward-rounded f32) box, 21 bits per axis -- the
//      scenes here put a 300-unit mesh on a 200 000-unit ground sphere: ten bits per axis would leave the mesh in two cells;
//   2. radix sort of (key, leaf index) -- rocPRIM;
//   3. one thread per internal node finds its range of sorted leaves and its split from the keys' common prefixes (equal
//      keys fall back to the sorted positions, so every leaf is distinguishable);
//   4. one thread per leaf walks up: the second thread to reach a node (an atomic counter per node) reads its two children's
//      boxes, stores them in the node -- as (lo, hi) pairs per axis, the layout guide_slab reads -- and carries their union
//      on.  Boxes only ever grow on the way up, in f32 min/max: containment is exact.
// Node 0 is the root (Karras' numbering), as the walk expects.
#include <hip/hip_runtime.h>

#include <cstring> // (rocPRIM's headers call memset without including it)

#include <rocprim/rocprim.hpp>

#include "guide_tree.h"

struct DevGuideLeaf { // 28 bytes: the padded box rounded outwards to f32, and the stream offset of the primitive record
    float lo[3], hi[3];
    uint32_t offset;
};

namespace {

__device__ __forceinline__ uint64_t spread21(uint32_t v) { // bit k of v -> bit 3k
    uint64_t x = v & 0x1FFFFFu;
    x = (x | x << 32) & 0x1F00000000FFFFull;
    x = (x | x << 16) & 0x1F0000FF0000FFull;
    x = (x | x << 8) & 0x100F00F00F00F00Full;
    x = (x | x << 4) & 0x10C30C30C30C30C3ull;
    x = (x | x << 2) & 0x1249249249249249ull;
    return x;
}

__global__ void guide_morton_kernel(const DevGuideLeaf* __restrict__ leaves, uint32_t n, float ox, float oy, float oz, float sx, float sy,
                                    float sz, uint64_t* __restrict__ keys, uint32_t* __restrict__ order) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const DevGuideLeaf l = leaves[i];
    const float cx = 0.5f * (l.lo[0] + l.hi[0]), cy = 0.5f * (l.lo[1] + l.hi[1]), cz = 0.5f * (l.lo[2] + l.hi[2]);
    const float qx = fminf(fmaxf((cx - ox) * sx, 0.0f), 2097151.0f), qy = fminf(fmaxf((cy - oy) * sy, 0.0f), 2097151.0f);
    const float qz = fminf(fmaxf((cz - oz) * sz, 0.0f), 2097151.0f);
    keys[i] = spread21((uint32_t)qx) << 2 | spread21((uint32_t)qy) << 1 | spread21((uint32_t)qz);
    order[i] = i;
}

// length of the common prefix of the keys at sorted positions i and j (-1 outside the array); equal keys are told apart by
// their positions
__device__ __forceinline__ int common_prefix(const uint64_t* __restrict__ keys, int n, int i, int j) {
    if (j < 0 || j >= n) return -1;
    const uint64_t a = keys[i], b = keys[j];
    if (a == b) return 64 + __clz((unsigned)(i ^ j));
    return __clzll((long long)(a ^ b));
}

// Karras 2012, section 4: internal node i covers the sorted leaves [min(i, j), max(i, j)]; children >= 0 are internal nodes,
// children < 0 are leaves (~position).
__global__ void guide_hierarchy_kernel(const uint64_t* __restrict__ keys, int n, int2* __restrict__ kids, int* __restrict__ parent_of_node,
                                       int* __restrict__ parent_of_leaf) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n - 1) return;
    const int d = common_prefix(keys, n, i, i + 1) - common_prefix(keys, n, i, i - 1) >= 0 ? 1 : -1;
    const int delta_min = common_prefix(keys, n, i, i - d);
    int l_max = 2;
    while (common_prefix(keys, n, i, i + l_max * d) > delta_min) l_max *= 2;
    int l = 0;
    for (int t = l_max / 2; t >= 1; t /= 2)
        if (common_prefix(keys, n, i, i + (l + t) * d) > delta_min) l += t;
    const int j = i + l * d;
    const int delta_node = common_prefix(keys, n, i, j);
    int s = 0;
    for (int t = (l + 1) / 2;; t = (t + 1) / 2) { // t = ceil(l / 2), ceil(l / 4), ..., 1
        if (common_prefix(keys, n, i, i + (s + t) * d) > delta_node) s += t;
        if (t <= 1) break;
    }
    const int gamma = i + s * d + min(d, 0);
    const int first = min(i, j), last = max(i, j);
    const int left = first == gamma ? ~gamma : gamma;
    const int right = last == gamma + 1 ? ~(gamma + 1) : gamma + 1;
    kids[i] = make_int2(left, right);
    if (left >= 0) parent_of_node[left] = i; else parent_of_leaf[~left] = i;
    if (right >= 0) parent_of_node[right] = i; else parent_of_leaf[~right] = i;
    if (i == 0) parent_of_node[0] = -1;
}

__global__ void guide_refit_kernel(const DevGuideLeaf* __restrict__ leaves, const uint32_t* __restrict__ order, int n, const int2* __restrict__ kids,
                                   const int* __restrict__ parent_of_node, const int* __restrict__ parent_of_leaf, unsigned* arrivals,
                                   float* own_box /* lo.xyz hi.xyz per internal node; written and read across workgroups */, GuideNode* __restrict__ nodes) {
    const int p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= n) return;
    int node = parent_of_leaf[p];
    // (a tree over 64-bit keys and 32-bit positions is at most 96 levels deep: the guard makes a malformed hierarchy end the loop
    // instead of hanging the GPU; the selftest and the upload's users would then see a broken tree, never a stuck kernel)
    for (int guard = 0; node >= 0 && guard < 128; ++guard) {
        // Release: what this thread finished below (its subtree's box, plain stores) is visible before the counter says so.
        // Acquire: the second arriver sees the first one's box (plain loads below) because it saw the first one's count.
        if (__hip_atomic_fetch_add(&arrivals[node], 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT) == 0u)
            return; // first to arrive: the other subtree is not finished, its thread goes on
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent"); // (only the second arriver reads what the first one published)
        const int2 k = kids[node];
        float box[2][6];
        uint32_t ref[2];
#pragma unroll
        for (int c = 0; c < 2; ++c) {
            const int kid = c == 0 ? k.x : k.y;
            if (kid < 0) {
                const DevGuideLeaf l = leaves[order[~kid]];
                for (int q = 0; q < 3; ++q) { box[c][q] = l.lo[q]; box[c][3 + q] = l.hi[q]; }
                ref[c] = PT_GUIDE_LEAF | l.offset;
            } else {
                for (int q = 0; q < 6; ++q) box[c][q] = own_box[(size_t)kid * 6 + q];
                ref[c] = (uint32_t)kid;
            }
        }
        GuideNode out;
        out.child[0] = ref[0];
        out.child[1] = ref[1];
        out.unused[0] = out.unused[1] = 0u;
        for (int c = 0; c < 2; ++c)
            for (int q = 0; q < 3; ++q) { out.box[c][2 * q] = box[c][q]; out.box[c][2 * q + 1] = box[c][3 + q]; }
        nodes[node] = out;
        for (int q = 0; q < 3; ++q) {
            own_box[(size_t)node * 6 + q] = fminf(box[0][q], box[1][q]);
            own_box[(size_t)node * 6 + 3 + q] = fmaxf(box[0][3 + q], box[1][3 + q]);
        }
        node = parent_of_node[node];
    }
}

struct Scratch { // freed on every way out
    void* p[8] = {};
    ~Scratch() {
        for (void* q : p)
            if (q) (void)hipFree(q);
    }
};

} // namespace

// Builds the guide tree over `leaves` (host array, padded boxes in f64; n >= 2) into d_nodes (device, n - 1 GuideNode records).
// Returns hipSuccess, or the failing call's error; *reason names it.  The stream is drained before returning.
hipError_t guide_build_on_device(hipStream_t stream, const std::vector<GuideLeaf>& leaves, GuideNode* d_nodes, const char** reason) {
    const size_t n = leaves.size();
    *reason = "";
    std::vector<DevGuideLeaf> h(n);
    float lo[3] = {INFINITY, INFINITY, INFINITY}, hi[3] = {-INFINITY, -INFINITY, -INFINITY};
    for (size_t i = 0; i < n; ++i) {
        for (int k = 0; k < 3; ++k) {
            h[i].lo[k] = guide::round_down(leaves[i].lo[k]);
            h[i].hi[k] = guide::round_up(leaves[i].hi[k]);
            const float c = 0.5f * (h[i].lo[k] + h[i].hi[k]);
            lo[k] = std::min(lo[k], c);
            hi[k] = std::max(hi[k], c);
        }
        h[i].offset = leaves[i].offset;
    }
    // One scale for the three axes (a cubic lattice over the longest side): a flat scene, e.g. a height field, then has no
    // high-order bits on its short axis and the upper levels split only where there is extent to split.
    float side = 0.0f;
    for (int k = 0; k < 3; ++k) side = std::max(side, hi[k] - lo[k]);
    float scale[3];
    for (int k = 0; k < 3; ++k) scale[k] = side > 0.0f && std::isfinite(side) ? 2097151.0f / side : 0.0f;

    Scratch s;
    auto alloc = [&](int slot, size_t bytes) { return hipMalloc(&s.p[slot], bytes ? bytes : 16); };
    hipError_t e;
#define GB_TRY(call, what)       \
    do {                         \
        e = (call);              \
        if (e != hipSuccess) {   \
            *reason = what;      \
            return e;            \
        }                        \
    } while (0)
    GB_TRY(alloc(0, n * sizeof(DevGuideLeaf)), "hipMalloc(leaves)");
    GB_TRY(alloc(1, n * sizeof(uint64_t) * 2), "hipMalloc(keys)");
    GB_TRY(alloc(2, n * sizeof(uint32_t) * 2), "hipMalloc(order)");
    GB_TRY(alloc(3, (n - 1) * sizeof(int2)), "hipMalloc(kids)");
    GB_TRY(alloc(4, (2 * n - 1) * sizeof(int)), "hipMalloc(parents)");
    GB_TRY(alloc(5, (n - 1) * (sizeof(unsigned) + 6 * sizeof(float))), "hipMalloc(arrivals, boxes)");
    DevGuideLeaf* d_leaves = (DevGuideLeaf*)s.p[0];
    uint64_t *keys_in = (uint64_t*)s.p[1], *keys_out = keys_in + n;
    uint32_t *order_in = (uint32_t*)s.p[2], *order_out = order_in + n;
    int2* kids = (int2*)s.p[3];
    int *parent_of_node = (int*)s.p[4], *parent_of_leaf = parent_of_node + (n - 1);
    unsigned* arrivals = (unsigned*)s.p[5];
    float* own_box = (float*)(arrivals + (n - 1));
    GB_TRY(hipMemcpyAsync(d_leaves, h.data(), n * sizeof(DevGuideLeaf), hipMemcpyHostToDevice, stream), "hipMemcpy(leaves)");
    GB_TRY(hipMemsetAsync(arrivals, 0, (n - 1) * sizeof(unsigned), stream), "hipMemset(arrivals)");
    const unsigned block = 256, grid_leaves = (unsigned)((n + block - 1) / block);
    hipLaunchKernelGGL(guide_morton_kernel, dim3(grid_leaves), dim3(block), 0, stream, d_leaves, (uint32_t)n, lo[0], lo[1], lo[2], scale[0], scale[1],
                       scale[2], keys_in, order_in);
    GB_TRY(hipGetLastError(), "guide_morton_kernel");
    size_t temp_bytes = 0;
    GB_TRY(rocprim::radix_sort_pairs(nullptr, temp_bytes, keys_in, keys_out, order_in, order_out, n, 0, 63, stream), "radix_sort_pairs(size)");
    GB_TRY(alloc(6, temp_bytes), "hipMalloc(sort)");
    GB_TRY(rocprim::radix_sort_pairs(s.p[6], temp_bytes, keys_in, keys_out, order_in, order_out, n, 0, 63, stream), "radix_sort_pairs");
    hipLaunchKernelGGL(guide_hierarchy_kernel, dim3(grid_leaves), dim3(block), 0, stream, keys_out, (int)n, kids, parent_of_node, parent_of_leaf);
    GB_TRY(hipGetLastError(), "guide_hierarchy_kernel");
    hipLaunchKernelGGL(guide_refit_kernel, dim3(grid_leaves), dim3(block), 0, stream, d_leaves, order_out, (int)n, kids, parent_of_node, parent_of_leaf,
                       arrivals, own_box, d_nodes);
    GB_TRY(hipGetLastError(), "guide_refit_kernel");
    // Completion: the root is written by the second of its two subtrees to arrive.  Anything else (a malformed hierarchy that
    // tripped the loop guard) would leave nodes unwritten in d_nodes, which is not zeroed: refuse the tree instead of uploading it.
    unsigned root_arrivals = 0;
    GB_TRY(hipMemcpyAsync(&root_arrivals, arrivals, sizeof(unsigned), hipMemcpyDeviceToHost, stream), "hipMemcpy(arrivals)");
    GB_TRY(hipStreamSynchronize(stream), "guide build");
#undef GB_TRY
    if (root_arrivals != 2u) {
        *reason = "guide_refit_kernel: the hierarchy did not close at the root (device-built guide tree refused)";
        return hipErrorUnknown;
    }
    return hipSuccess;
}
