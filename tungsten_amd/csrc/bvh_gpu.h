// Device-side BVH build for flat triangle soups (north star: "LBVH/SAH-BVH"; design model for what a builder has to deliver: the reference's
// src/core/bvh/BvhBuilder.cpp:99-185).  The host builder (bvh_build.h: binned SAH, one thread) is the better tree and stays the default for scenes of the
// benchmark's size; it is also O(n log n) on ONE core -- tens of seconds for ten million triangles.  Above TGPU_GPU_BVH_MIN triangles (default 1 Mi;
// TGPU_GPU_BVH=1 forces it, =0 forbids it) the scene's tree is built here instead:
//
//   1. k_lbvh_prepare   per triangle: padded box (include/tgpu_bounds.h), centroid; scene centroid bounds by wave reduction + atomics
//   2. k_lbvh_morton    30-bit Morton code of the centroid
//   3. hipcub radix sort of (code, triangle) pairs
//   4. k_lbvh_hierarchy Karras 2012: internal node i of the binary radix tree over the sorted codes (ties broken by position), all nodes in parallel
//   5. k_lbvh_boxes     bottom-up box union, one thread per leaf, the second arrival at a node continues upwards
//   6. k_lbvh_collapse  BVH2 -> the quantised BVH4 the traversal kernels read (same node format and the same "open the child with the largest
//                       area" rule as bvh_build.h's Collapser), breadth first: one launch per level, nodes numbered by an atomic counter;
//                       subtrees of <= 2 triangles become leaves (a leaf costs one traversal step per triangle)
//   7. k_lbvh_triangles the 64-byte triangle records (v0, e1, e2, Ng, soup id) in tree order
//
// Correctness does not depend on the shape of the tree: nearest hit = min (t, id) over every triangle whose binary32 test passes, and the boxes are
// conservative (padded triangle boxes, outward quantisation) -- so an image rendered through this tree is bit-identical to one rendered through the
// host's (tests/test_gpu_parity.py).  Quality: a Morton-order tree costs more node visits per ray than the SAH tree (measured).
#pragma once
#include <hip/hip_runtime.h>
#include <hipcub/hipcub.hpp>

#include <chrono>
#include <cstdint>
#include <vector>

namespace tg_lbvh {

struct Box3 { float lo[3], hi[3]; };

struct Result {
    float4 *nodes = nullptr;   // nodeCount x 4 float4 (quantised BVH4, kernels.h travApply)
    float4 *tris = nullptr;    // nTris x 4 float4, tree order
    uint32_t rootChild = 0x80000000u;
    uint32_t nodeCount = 0;
    int maxDepth4 = 0;
    double buildMs = 0.0;      // steps 1-7 on the device, after the positions are resident
    double copyMs = 0.0;       // host -> device copy of the positions
};

__device__ inline uint32_t expandBits10(uint32_t v) // 10 bits -> every third bit
{
    v = (v*0x00010001u) & 0xFF0000FFu;
    v = (v*0x00000101u) & 0x0F00F00Fu;
    v = (v*0x00000011u) & 0xC30C30C3u;
    v = (v*0x00000005u) & 0x49249249u;
    return v;
}
__device__ inline void atomicMinFloat(float *addr, float v) // v finite; works through the ordered-int trick for both signs
{
    if (v >= 0.0f) atomicMin(reinterpret_cast<int *>(addr), __float_as_int(v));
    else atomicMax(reinterpret_cast<unsigned int *>(addr), __float_as_uint(v));
}
__device__ inline void atomicMaxFloat(float *addr, float v)
{
    if (v >= 0.0f) atomicMax(reinterpret_cast<int *>(addr), __float_as_int(v));
    else atomicMin(reinterpret_cast<unsigned int *>(addr), __float_as_uint(v));
}

// centroidBounds: lo[3] initialised to +inf, hi[3] to -inf
static __global__ void __launch_bounds__(256) k_lbvh_prepare(const float *triPos, uint32_t n, float triPad, Box3 *triBox, float *centroid, float *centroidBounds)
{
    const uint32_t i = blockIdx.x*blockDim.x + threadIdx.x;
    float c[3] = {0.0f, 0.0f, 0.0f};
    const bool live = i < n;
    if (live) {
        const float *p = triPos + size_t(i)*9;
        Box3 b;
        for (int k = 0; k < 3; ++k) {
            const float lo = fminf(p[k], fminf(p[3 + k], p[6 + k])), hi = fmaxf(p[k], fmaxf(p[3 + k], p[6 + k]));
            c[k] = 0.5f*(lo + hi);
            b.lo[k] = lo - triPad;
            b.hi[k] = hi + triPad;
            centroid[size_t(i)*3 + k] = c[k];
        }
        triBox[i] = b;
    }
    for (int k = 0; k < 3; ++k) { // wave reduction, one atomic pair per wave and axis
        float lo = live ? c[k] : INFINITY, hi = live ? c[k] : -INFINITY;
        for (int off = 32; off > 0; off >>= 1) {
            lo = fminf(lo, __shfl_xor(lo, off));
            hi = fmaxf(hi, __shfl_xor(hi, off));
        }
        if ((threadIdx.x & 63) == 0 && lo <= hi) {
            atomicMinFloat(&centroidBounds[k], lo);
            atomicMaxFloat(&centroidBounds[3 + k], hi);
        }
    }
}

static __global__ void __launch_bounds__(256) k_lbvh_morton(const float *centroid, const float *centroidBounds, uint32_t n, uint32_t *code, uint32_t *index)
{
    const uint32_t i = blockIdx.x*blockDim.x + threadIdx.x;
    if (i >= n) return;
    uint32_t q[3];
    for (int k = 0; k < 3; ++k) {
        const float lo = centroidBounds[k], ext = centroidBounds[3 + k] - lo;
        float t = ext > 0.0f ? (centroid[size_t(i)*3 + k] - lo)/ext : 0.0f;
        t = fminf(fmaxf(t*1024.0f, 0.0f), 1023.0f);
        q[k] = uint32_t(t);
    }
    code[i] = (expandBits10(q[0]) << 2) | (expandBits10(q[1]) << 1) | expandBits10(q[2]);
    index[i] = i;
}

// Binary radix tree over n sorted codes: n - 1 internal nodes.  Child references: bit 31 set = leaf (sorted position), else internal node.
struct Radix {
    uint32_t *left, *right, *parent;   // internal nodes [0, n - 1)
    uint32_t *leafParent;              // leaves [0, n)
    uint32_t *first, *last;            // sorted positions covered by an internal node
};

__device__ inline int lbvhDelta(const uint32_t *code, uint32_t n, int i, int j)
{
    if (j < 0 || j >= int(n)) return -1;
    const uint32_t a = code[i], b = code[j];
    if (a == b) return 32 + __clz(uint32_t(i) ^ uint32_t(j));
    return __clz(a ^ b);
}

static __global__ void __launch_bounds__(256) k_lbvh_hierarchy(const uint32_t *code, uint32_t n, Radix r)
{
    const int i = int(blockIdx.x*blockDim.x + threadIdx.x);
    if (i >= int(n) - 1) return;
    const int d = lbvhDelta(code, n, i, i + 1) - lbvhDelta(code, n, i, i - 1) >= 0 ? 1 : -1;
    const int deltaMin = lbvhDelta(code, n, i, i - d);
    int lmax = 2;
    while (lbvhDelta(code, n, i, i + lmax*d) > deltaMin) lmax *= 2;
    int l = 0;
    for (int t = lmax/2; t >= 1; t /= 2)
        if (lbvhDelta(code, n, i, i + (l + t)*d) > deltaMin) l += t;
    const int j = i + l*d;
    const int deltaNode = lbvhDelta(code, n, i, j);
    int s = 0;
    for (int t = (l + 1)/2; ; t = (t + 1)/2) {
        if (lbvhDelta(code, n, i, i + (s + t)*d) > deltaNode) s += t;
        if (t <= 1) break;
    }
    const int gamma = i + s*d + (d < 0 ? d : 0);
    const int lo = i < j ? i : j, hi = i < j ? j : i;
    const bool leftLeaf = lo == gamma, rightLeaf = hi == gamma + 1;
    r.left[i] = leftLeaf ? (0x80000000u | uint32_t(gamma)) : uint32_t(gamma);
    r.right[i] = rightLeaf ? (0x80000000u | uint32_t(gamma + 1)) : uint32_t(gamma + 1);
    if (leftLeaf) r.leafParent[gamma] = uint32_t(i); else r.parent[gamma] = uint32_t(i);
    if (rightLeaf) r.leafParent[gamma + 1] = uint32_t(i); else r.parent[gamma + 1] = uint32_t(i);
    r.first[i] = uint32_t(lo);
    r.last[i] = uint32_t(hi);
    if (i == 0) r.parent[0] = 0xFFFFFFFFu;
}

__device__ inline Box3 unionBox(const Box3 &a, const Box3 &b)
{
    Box3 o;
    for (int k = 0; k < 3; ++k) { o.lo[k] = fminf(a.lo[k], b.lo[k]); o.hi[k] = fmaxf(a.hi[k], b.hi[k]); }
    return o;
}

// nodeBox of internal nodes; leaf boxes are triBox[index[position]].  visited: zeroed counters, one per internal node.
static __global__ void __launch_bounds__(256) k_lbvh_boxes(uint32_t n, Radix r, const uint32_t *index, const Box3 *triBox, Box3 *nodeBox, uint32_t *visited)
{
    const uint32_t leaf = blockIdx.x*blockDim.x + threadIdx.x;
    if (leaf >= n || n < 2) return;
    uint32_t node = r.leafParent[leaf];
    while (true) {
        // release our subtree's box, acquire the sibling's: the increment is acq_rel at device scope, so the stores before it are visible to
        // whoever sees the count and the loads after it bypass whatever this CU's L1 still holds of the line (a neighbouring box read earlier)
        if (__hip_atomic_fetch_add(&visited[node], 1u, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT) == 0u) return; // the sibling is not finished: its last thread continues
        const uint32_t a = r.left[node], b = r.right[node];
        const volatile Box3 *boxes = nodeBox;
        Box3 ba, bb;
        if (a & 0x80000000u) ba = triBox[index[a & 0x7FFFFFFFu]];
        else for (int k = 0; k < 3; ++k) { ba.lo[k] = boxes[a].lo[k]; ba.hi[k] = boxes[a].hi[k]; }
        if (b & 0x80000000u) bb = triBox[index[b & 0x7FFFFFFFu]];
        else for (int k = 0; k < 3; ++k) { bb.lo[k] = boxes[b].lo[k]; bb.hi[k] = boxes[b].hi[k]; }
        nodeBox[node] = unionBox(ba, bb);
        if (node == 0u) return;
        node = r.parent[node];
    }
}

struct WorkItem { uint32_t node2, node4; };

__device__ inline float halfAreaOf(const Box3 &b)
{
    const float dx = b.hi[0] - b.lo[0], dy = b.hi[1] - b.lo[1], dz = b.hi[2] - b.lo[2];
    return dx*dy + dy*dz + dz*dx;
}
__device__ inline void padBox(Box3 &b) // bvh_build.h Builder::pad: the fma-form slab test stays conservative
{
    for (int k = 0; k < 3; ++k) {
        const float m = fmaxf(fabsf(b.lo[k]), fabsf(b.hi[k]));
        const float eps = m*2e-6f + 1e-30f;
        b.lo[k] -= eps;
        b.hi[k] += eps;
    }
}

// One BVH4 node per work item.  counters[0] = nodes allocated so far, counters[1] = items written to `next`.
static __global__ void __launch_bounds__(256) k_lbvh_collapse(uint32_t nItems, const WorkItem *items, WorkItem *next, uint32_t *counters, Radix r, const uint32_t *index,
                                                              const Box3 *triBox, const Box3 *nodeBox, float4 *nodes4, uint32_t maxLeaf)
{
    const uint32_t w = blockIdx.x*blockDim.x + threadIdx.x;
    if (w >= nItems) return;
    const WorkItem item = items[w];
    uint32_t ref[4];   // bit 31: leaf-like (payload = first << 3 | count - 1), else BVH2 internal node still to be emitted
    Box3 box[4];
    auto childOf = [&](uint32_t c, uint32_t &outRef, Box3 &outBox) {
        if (c & 0x80000000u) { // a single triangle
            const uint32_t pos = c & 0x7FFFFFFFu;
            outRef = 0x80000000u | (pos << 3);
            outBox = triBox[index[pos]];
        } else {
            const uint32_t count = r.last[c] - r.first[c] + 1u;
            outBox = nodeBox[c];
            outRef = count <= maxLeaf ? (0x80000000u | (r.first[c] << 3) | (count - 1u)) : c;
        }
    };
    int n = 2;
    childOf(r.left[item.node2], ref[0], box[0]);
    childOf(r.right[item.node2], ref[1], box[1]);
    while (n < 4) { // open the inner child with the largest surface area until four slots are used (bvh_build.h, Collapser::emit)
        int best = -1;
        float bestArea = -1.0f;
        for (int i = 0; i < n; ++i)
            if (!(ref[i] & 0x80000000u) && halfAreaOf(box[i]) > bestArea) { bestArea = halfAreaOf(box[i]); best = i; }
        if (best < 0) break;
        const uint32_t open = ref[best];
        childOf(r.left[open], ref[best], box[best]);
        childOf(r.right[open], ref[n], box[n]);
        ++n;
    }
    uint32_t out[4] = {0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu};
    Box3 frame;
    for (int k = 0; k < 3; ++k) { frame.lo[k] = INFINITY; frame.hi[k] = -INFINITY; }
    for (int i = 0; i < n; ++i) {
        padBox(box[i]);
        frame = unionBox(frame, box[i]);
        if (ref[i] & 0x80000000u) {
            out[i] = ref[i];
        } else {
            const uint32_t idx = atomicAdd(&counters[0], 1u);
            const uint32_t slot = atomicAdd(&counters[1], 1u);
            next[slot] = WorkItem{ref[i], idx};
            out[i] = idx;
        }
    }
    uint32_t e[3];
    float scale[3];
    for (int a = 0; a < 3; ++a) {
        const float ext = frame.hi[a] - frame.lo[a];
        int ex = 1;
        if (ext > 0.0f) {
            int fe;
            (void)frexpf(ext/255.0f, &fe); // ext/255 = m * 2^fe, m in [0.5, 1)  -> 2^fe >= ext/255
            ex = fe + 127;
        }
        ex = min(max(ex, 1), 254);
        while (ex < 254 && frame.lo[a] + 255.0f*__uint_as_float(uint32_t(ex) << 23) < frame.hi[a]) ++ex; // 255 quanta really cover the extent
        e[a] = uint32_t(ex);
        scale[a] = __uint_as_float(uint32_t(ex) << 23);
    }
    uint32_t lo[3] = {0, 0, 0}, hi[3] = {0, 0, 0};
    for (int i = 0; i < 4; ++i)
        for (int a = 0; a < 3; ++a) {
            uint32_t ql = 255, qh = 0; // empty slot: inverted box
            if (i < n) {
                const float o = frame.lo[a], s = scale[a];
                int l = int(floorf((box[i].lo[a] - o)/s));
                l = min(max(l, 0), 255);
                while (l > 0 && o + float(l)*s > box[i].lo[a]) --l;
                int h = int(ceilf((box[i].hi[a] - o)/s));
                h = min(max(h, 0), 255);
                while (h < 255 && o + float(h)*s < box[i].hi[a]) ++h;
                ql = uint32_t(l);
                qh = uint32_t(h);
            }
            lo[a] |= ql << (8*i);
            hi[a] |= qh << (8*i);
        }
    float4 *q = nodes4 + size_t(item.node4)*4;
    q[0] = make_float4(frame.lo[0], frame.lo[1], frame.lo[2], __uint_as_float(e[0] | (e[1] << 8) | (e[2] << 16)));
    q[1] = make_float4(__uint_as_float(lo[0]), __uint_as_float(lo[1]), __uint_as_float(lo[2]), __uint_as_float(hi[0]));
    q[2] = make_float4(__uint_as_float(hi[1]), __uint_as_float(hi[2]), 0.0f, 0.0f);
    q[3] = make_float4(__uint_as_float(out[0]), __uint_as_float(out[1]), __uint_as_float(out[2]), __uint_as_float(out[3]));
}

// 64-byte triangle records in tree (= sorted) order: v0, e1 = v0 - v1, e2 = v2 - v0, Ng = e1 x e2 (Embree's precomputed form, the operation order of
// bvh_build.h's makeLeaf), then the soup triangle id
static __global__ void __launch_bounds__(256) k_lbvh_triangles(const float *triPos, const uint32_t *index, uint32_t n, float4 *tris)
{
    const uint32_t i = blockIdx.x*blockDim.x + threadIdx.x;
    if (i >= n) return;
    const uint32_t id = index[i];
    const float *p = triPos + size_t(id)*9;
    const float e1[3] = {p[0] - p[3], p[1] - p[4], p[2] - p[5]};
    const float e2[3] = {p[6] - p[0], p[7] - p[1], p[8] - p[2]};
    const float ng[3] = {e1[1]*e2[2] - e1[2]*e2[1], e1[2]*e2[0] - e1[0]*e2[2], e1[0]*e2[1] - e1[1]*e2[0]};
    float4 *t = tris + size_t(i)*4;
    t[0] = make_float4(p[0], p[1], p[2], e1[0]);
    t[1] = make_float4(e1[1], e1[2], e2[0], e2[1]);
    t[2] = make_float4(e2[2], ng[0], ng[1], ng[2]);
    t[3] = make_float4(__uint_as_float(id), ng[0]*ng[0] + (ng[1]*ng[1] + ng[2]*ng[2]), 0.0f, 0.0f); // id, |Ng|^2 (triTest's grazing test)
}

#define TG_LBVH_CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { err = hipGetErrorString(e_); goto done; } } while (0)

// Builds the tree for `n` triangles (host positions, 9 floats each).  nodes / tris of the result are device allocations the caller owns.
// Returns nullptr on success, else a message.
inline const char *build(const float *hostTriPos, uint32_t n, float triPad, uint32_t maxLeaf, hipStream_t stream, Result &out)
{
    const char *err = nullptr;
    std::vector<void *> scratch;
    auto alloc = [&](size_t bytes) -> void * { void *p = nullptr; if (hipMalloc(&p, bytes ? bytes : 4) != hipSuccess) return nullptr; scratch.push_back(p); return p; };
    const dim3 block(256), grid((n + 255u)/256u);
    float *dPos = nullptr, *dCentroid = nullptr, *dBounds = nullptr;
    Box3 *dTriBox = nullptr, *dNodeBox = nullptr;
    uint32_t *dCode = nullptr, *dIndex = nullptr, *dCodeSorted = nullptr, *dIndexSorted = nullptr, *dVisited = nullptr, *dCounters = nullptr;
    WorkItem *dItems[2] = {nullptr, nullptr};
    Radix r{};
    void *dTemp = nullptr;
    size_t tempBytes = 0;
    float4 *dNodes4 = nullptr, *dTris = nullptr;
    out = Result{};
    if (n < 3) return "device BVH build needs at least three triangles";
    {
        const auto t0 = std::chrono::steady_clock::now();
        dPos = static_cast<float *>(alloc(size_t(n)*9*sizeof(float)));
        if (!dPos) { err = "out of device memory"; goto done; }
        TG_LBVH_CHECK(hipMemcpyAsync(dPos, hostTriPos, size_t(n)*9*sizeof(float), hipMemcpyHostToDevice, stream));
        TG_LBVH_CHECK(hipStreamSynchronize(stream));
        out.copyMs = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
    }
    {
        const auto t0 = std::chrono::steady_clock::now();
        dCentroid = static_cast<float *>(alloc(size_t(n)*3*sizeof(float)));
        dBounds = static_cast<float *>(alloc(6*sizeof(float)));
        dTriBox = static_cast<Box3 *>(alloc(size_t(n)*sizeof(Box3)));
        dNodeBox = static_cast<Box3 *>(alloc(size_t(n)*sizeof(Box3)));
        dCode = static_cast<uint32_t *>(alloc(size_t(n)*4)); dIndex = static_cast<uint32_t *>(alloc(size_t(n)*4));
        dCodeSorted = static_cast<uint32_t *>(alloc(size_t(n)*4)); dIndexSorted = static_cast<uint32_t *>(alloc(size_t(n)*4));
        dVisited = static_cast<uint32_t *>(alloc(size_t(n)*4));
        dCounters = static_cast<uint32_t *>(alloc(2*4));
        dItems[0] = static_cast<WorkItem *>(alloc(size_t(n)*sizeof(WorkItem))); dItems[1] = static_cast<WorkItem *>(alloc(size_t(n)*sizeof(WorkItem)));
        r.left = static_cast<uint32_t *>(alloc(size_t(n)*4)); r.right = static_cast<uint32_t *>(alloc(size_t(n)*4)); r.parent = static_cast<uint32_t *>(alloc(size_t(n)*4));
        r.leafParent = static_cast<uint32_t *>(alloc(size_t(n)*4)); r.first = static_cast<uint32_t *>(alloc(size_t(n)*4)); r.last = static_cast<uint32_t *>(alloc(size_t(n)*4));
        if (!dCentroid || !dBounds || !dTriBox || !dNodeBox || !dCode || !dIndex || !dCodeSorted || !dIndexSorted || !dVisited || !dCounters || !dItems[0] || !dItems[1] ||
            !r.left || !r.right || !r.parent || !r.leafParent || !r.first || !r.last) { err = "out of device memory"; goto done; }
        // a BVH4 over n triangles with leaves of >= 1 triangle has fewer than n inner nodes
        if (hipMalloc(&dNodes4, size_t(n)*4*sizeof(float4)) != hipSuccess || hipMalloc(&dTris, size_t(n)*4*sizeof(float4)) != hipSuccess) { err = "out of device memory"; goto done; }
        const float inf = INFINITY;
        const float boundsInit[6] = {inf, inf, inf, -inf, -inf, -inf};
        TG_LBVH_CHECK(hipMemcpyAsync(dBounds, boundsInit, sizeof(boundsInit), hipMemcpyHostToDevice, stream));
        hipLaunchKernelGGL(k_lbvh_prepare, grid, block, 0, stream, dPos, n, triPad, dTriBox, dCentroid, dBounds);
        hipLaunchKernelGGL(k_lbvh_morton, grid, block, 0, stream, dCentroid, dBounds, n, dCode, dIndex);
        TG_LBVH_CHECK(hipcub::DeviceRadixSort::SortPairs(nullptr, tempBytes, dCode, dCodeSorted, dIndex, dIndexSorted, int(n), 0, 30, stream));
        dTemp = alloc(tempBytes);
        if (!dTemp) { err = "out of device memory"; goto done; }
        TG_LBVH_CHECK(hipcub::DeviceRadixSort::SortPairs(dTemp, tempBytes, dCode, dCodeSorted, dIndex, dIndexSorted, int(n), 0, 30, stream));
        hipLaunchKernelGGL(k_lbvh_hierarchy, grid, block, 0, stream, dCodeSorted, n, r);
        TG_LBVH_CHECK(hipMemsetAsync(dVisited, 0, size_t(n)*4, stream));
        hipLaunchKernelGGL(k_lbvh_boxes, grid, block, 0, stream, n, r, dIndexSorted, dTriBox, dNodeBox, dVisited);
        hipLaunchKernelGGL(k_lbvh_triangles, grid, block, 0, stream, dPos, dIndexSorted, n, dTris);
        // breadth-first collapse: level 0 is the radix tree's root -> BVH4 node 0
        const WorkItem rootItem{0u, 0u};
        uint32_t counters[2] = {1u, 0u};
        TG_LBVH_CHECK(hipMemcpyAsync(dItems[0], &rootItem, sizeof(rootItem), hipMemcpyHostToDevice, stream));
        uint32_t nItems = 1;
        int level = 0, cur = 0;
        while (nItems) {
            ++level;
            counters[1] = 0u;
            TG_LBVH_CHECK(hipMemcpyAsync(dCounters + 1, &counters[1], 4, hipMemcpyHostToDevice, stream));
            if (level == 1) TG_LBVH_CHECK(hipMemcpyAsync(dCounters, &counters[0], 4, hipMemcpyHostToDevice, stream));
            hipLaunchKernelGGL(k_lbvh_collapse, dim3((nItems + 255u)/256u), block, 0, stream, nItems, dItems[cur], dItems[cur ^ 1], dCounters, r, dIndexSorted, dTriBox, dNodeBox,
                               dNodes4, maxLeaf);
            TG_LBVH_CHECK(hipMemcpyAsync(counters, dCounters, 8, hipMemcpyDeviceToHost, stream));
            TG_LBVH_CHECK(hipStreamSynchronize(stream));
            nItems = counters[1];
            cur ^= 1;
            if (level > 4096) { err = "device BVH collapse did not terminate"; goto done; }
        }
        TG_LBVH_CHECK(hipGetLastError());
        out.nodeCount = counters[0];
        out.maxDepth4 = level;
        out.rootChild = 0u;
        out.nodes = dNodes4;
        out.tris = dTris;
        out.buildMs = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
    }
done:
    for (void *p : scratch) (void)hipFree(p);
    if (err) {
        if (dNodes4) (void)hipFree(dNodes4);
        if (dTris) (void)hipFree(dTris);
        out = Result{};
    }
    return err;
}
#undef TG_LBVH_CHECK

} // namespace tg_lbvh
