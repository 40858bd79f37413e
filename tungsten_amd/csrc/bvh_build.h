// Host-side SAH builder for the GPU BVH2 (replaces Embree's BVH4 builder; design model:
// the reference's own binned SAH, src/core/bvh/BvhBuilder.cpp:99-157 and
// BinnedSahSplitter.hpp:37-65 -- 32 bins, object splits, full bounds).
//
// GPU node layout (64 B = 4 x float4, one node holds BOTH children's boxes so a traversal
// step needs a single 64-byte fetch):
//   q0 = (c0.lo.x, c0.lo.y, c0.lo.z, c0.hi.x)
//   q1 = (c0.hi.y, c0.hi.z, c1.lo.x, c1.lo.y)
//   q2 = (c1.lo.z, c1.hi.x, c1.hi.y, c1.hi.z)
//   q3 = (bits child0, bits child1, 0, 0)
// child >= 0: inner node index; child < 0: leaf, payload = child & 0x7FFFFFFF =
// (firstTri << 3) | (count - 1), count in 1..8, triangles stored in BVH order.
// Triangle layout (48 B = 3 x float4), Embree's precomputed form
// (thirdparty/embree/kernels/geometry/triangle.h:53-54): v0, e1 = v0-v1, e2 = v2-v0, Ng = e1 x e2
//   t0 = (v0.x, v0.y, v0.z, e1.x)  t1 = (e1.y, e1.z, e2.x, e2.y)  t2 = (e2.z, Ng.x, Ng.y, Ng.z)
#pragma once
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <limits>
#include <vector>

namespace tgpu_bvh {

struct Float4 { float x, y, z, w; };

struct Bvh {
    std::vector<Float4> nodes;     // 4 per node; node 0 is the root
    std::vector<Float4> tris;      // 3 per triangle, BVH order
    std::vector<uint32_t> triId;   // BVH order -> index into the scene's triangle soup
    uint32_t rootChild = 0;        // encoded root (a leaf when the scene has <= 4 triangles)
    int maxDepth = 0;
    uint32_t nodeCount = 0;
};

struct Box {
    float lo[3], hi[3];
    void reset()
    {
        for (int k = 0; k < 3; ++k) { lo[k] = std::numeric_limits<float>::infinity(); hi[k] = -lo[k]; }
    }
    void grow(const Box &b)
    {
        for (int k = 0; k < 3; ++k) { lo[k] = std::min(lo[k], b.lo[k]); hi[k] = std::max(hi[k], b.hi[k]); }
    }
    float halfArea() const
    {
        float dx = hi[0] - lo[0], dy = hi[1] - lo[1], dz = hi[2] - lo[2];
        return dx*dy + dy*dz + dz*dx;
    }
};

class Builder {
public:
    // triPad: every triangle box grows by this much in all directions (include/tgpu_bounds.h: the walk must never prune a
    // triangle whose binary32 test passes)
    Builder(const float *triPos, uint32_t nTris, float triPad) : _pos(triPos), _n(nTris), _triPad(triPad) {}

    Bvh build()
    {
        Bvh out;
        _out = &out;
        _boxes.resize(_n);
        _cent.resize(size_t(_n)*3);
        _order.resize(_n);
        for (uint32_t i = 0; i < _n; ++i) {
            const float *p = _pos + size_t(i)*9;
            Box b;
            for (int k = 0; k < 3; ++k) {
                b.lo[k] = std::min(p[k], std::min(p[3 + k], p[6 + k]));
                b.hi[k] = std::max(p[k], std::max(p[3 + k], p[6 + k]));
                _cent[size_t(i)*3 + k] = 0.5f*(b.lo[k] + b.hi[k]);
                b.lo[k] -= _triPad;
                b.hi[k] += _triPad;
            }
            _boxes[i] = b;
            _order[i] = i;
        }
        out.triId.reserve(_n);
        out.tris.reserve(size_t(_n)*3);
        if (_n == 0) {
            out.rootChild = 0x80000000u; // never dereferenced: kernels test n_tris first
            return out;
        }
        Box rootBox;
        out.rootChild = uint32_t(recurse(0, _n, 1, rootBox));
        out.nodeCount = uint32_t(out.nodes.size()/4);
        return out;
    }

private:
    static constexpr int Bins = 32;
    // leaves hold 1..8 triangles (3 bits of the payload); TGPU_BVH_LEAF=1..8 sets the limit (experiments)
    uint32_t MaxLeaf = leafLimit();
    static uint32_t leafLimit()
    {
        const char *e = std::getenv("TGPU_BVH_LEAF");
        int v = e ? std::atoi(e) : 2; // 2 is the measured optimum once a triangle costs one traversal step (profiles/r01_ab_notes.txt)
        return uint32_t(v < 1 ? 1 : (v > 8 ? 8 : v));
    }

    // TGPU_BVH_ALIGN=1: a two-triangle leaf never straddles a 128-byte line (two 64-byte records): an unreferenced filler record goes in front of it
    // when it would (A/B, profiles/r02_ab_notes.txt)
    static bool alignLeaves()
    {
        const char *e = std::getenv("TGPU_BVH_ALIGN");
        return e ? std::atoi(e) != 0 : false;
    }
    int32_t makeLeaf(uint32_t first, uint32_t count, Box &box)
    {
        if (count == 2 && (_out->triId.size() & 1u) && alignLeaves()) {
            _out->tris.push_back({0.0f, 0.0f, 0.0f, 0.0f});
            _out->tris.push_back({0.0f, 0.0f, 0.0f, 0.0f});
            _out->tris.push_back({0.0f, 0.0f, 0.0f, 0.0f});
            _out->triId.push_back(0);
        }
        uint32_t start = uint32_t(_out->triId.size());
        box.reset();
        for (uint32_t i = first; i < first + count; ++i) {
            uint32_t id = _order[i];
            box.grow(_boxes[id]);
            const float *p = _pos + size_t(id)*9;
            float v0[3] = {p[0], p[1], p[2]};
            float e1[3] = {p[0] - p[3], p[1] - p[4], p[2] - p[5]};
            float e2[3] = {p[6] - p[0], p[7] - p[1], p[8] - p[2]};
            float ng[3] = {e1[1]*e2[2] - e1[2]*e2[1], e1[2]*e2[0] - e1[0]*e2[2], e1[0]*e2[1] - e1[1]*e2[0]};
            _out->tris.push_back({v0[0], v0[1], v0[2], e1[0]});
            _out->tris.push_back({e1[1], e1[2], e2[0], e2[1]});
            _out->tris.push_back({e2[2], ng[0], ng[1], ng[2]});
            _out->triId.push_back(id);
        }
        return int32_t(0x80000000u | (start << 3) | (count - 1));
    }

    int32_t recurse(uint32_t first, uint32_t count, int depth, Box &box)
    {
        _out->maxDepth = std::max(_out->maxDepth, depth);
        if (count <= MaxLeaf) return makeLeaf(first, count, box);

        Box cb;
        cb.reset();
        for (uint32_t i = first; i < first + count; ++i) {
            const float *c = &_cent[size_t(_order[i])*3];
            for (int k = 0; k < 3; ++k) { cb.lo[k] = std::min(cb.lo[k], c[k]); cb.hi[k] = std::max(cb.hi[k], c[k]); }
        }
        float bestCost = std::numeric_limits<float>::infinity();
        int bestAxis = -1, bestSplit = -1;
        for (int ax = 0; ax < 3; ++ax) {
            float ext = cb.hi[ax] - cb.lo[ax];
            if (!(ext > 0.0f)) continue;
            float scale = float(Bins)/ext;
            uint32_t cnt[Bins] = {0};
            Box bb[Bins];
            for (int b = 0; b < Bins; ++b) bb[b].reset();
            for (uint32_t i = first; i < first + count; ++i) {
                uint32_t id = _order[i];
                int b = std::min(Bins - 1, int((_cent[size_t(id)*3 + ax] - cb.lo[ax])*scale));
                cnt[b]++;
                bb[b].grow(_boxes[id]);
            }
            float rightArea[Bins];
            uint32_t rightCnt[Bins];
            Box acc;
            acc.reset();
            uint32_t c = 0;
            for (int b = Bins - 1; b >= 1; --b) {
                acc.grow(bb[b]);
                c += cnt[b];
                rightArea[b] = c ? acc.halfArea() : 0.0f;
                rightCnt[b] = c;
            }
            acc.reset();
            c = 0;
            for (int b = 0; b < Bins - 1; ++b) {
                acc.grow(bb[b]);
                c += cnt[b];
                if (c == 0 || rightCnt[b + 1] == 0) continue;
                float cost = acc.halfArea()*float(c) + rightArea[b + 1]*float(rightCnt[b + 1]);
                if (cost < bestCost) { bestCost = cost; bestAxis = ax; bestSplit = b; }
            }
        }
        uint32_t mid;
        if (bestAxis >= 0) {
            float ext = cb.hi[bestAxis] - cb.lo[bestAxis];
            float scale = float(Bins)/ext;
            auto it = std::partition(_order.begin() + first, _order.begin() + first + count, [&](uint32_t id) {
                int b = std::min(Bins - 1, int((_cent[size_t(id)*3 + bestAxis] - cb.lo[bestAxis])*scale));
                return b <= bestSplit;
            });
            mid = uint32_t(it - _order.begin());
            if (mid == first || mid == first + count) mid = first + count/2;
        } else {
            mid = first + count/2; // all centroids coincide
        }
        // depth guard: the traversal stack is sized from maxDepth, keep it bounded for pathological input
        if (depth >= 60) {
            std::nth_element(_order.begin() + first, _order.begin() + first + count/2, _order.begin() + first + count,
                             [&](uint32_t a, uint32_t b) { return _cent[size_t(a)*3] < _cent[size_t(b)*3]; });
            mid = first + count/2;
        }

        uint32_t nodeIdx = uint32_t(_out->nodes.size()/4);
        _out->nodes.resize(_out->nodes.size() + 4);
        Box b0, b1;
        int32_t c0 = recurse(first, mid - first, depth + 1, b0);
        int32_t c1 = recurse(mid, first + count - mid, depth + 1, b1);
        // pad the boxes by a few ulps so that the fma-based slab test stays conservative
        pad(b0);
        pad(b1);
        Float4 *q = &_out->nodes[size_t(nodeIdx)*4];
        q[0] = {b0.lo[0], b0.lo[1], b0.lo[2], b0.hi[0]};
        q[1] = {b0.hi[1], b0.hi[2], b1.lo[0], b1.lo[1]};
        q[2] = {b1.lo[2], b1.hi[0], b1.hi[1], b1.hi[2]};
        float f0, f1;
        std::memcpy(&f0, &c0, 4);
        std::memcpy(&f1, &c1, 4);
        q[3] = {f0, f1, 0.0f, 0.0f};
        box = b0;
        box.grow(b1);
        return int32_t(nodeIdx);
    }

    static void pad(Box &b)
    {
        for (int k = 0; k < 3; ++k) {
            float m = std::max(std::abs(b.lo[k]), std::abs(b.hi[k]));
            float eps = m*2e-6f + 1e-30f;
            b.lo[k] -= eps;
            b.hi[k] += eps;
        }
    }

    const float *_pos;
    uint32_t _n;
    float _triPad;
    Bvh *_out = nullptr;
    std::vector<Box> _boxes;
    std::vector<float> _cent;
    std::vector<uint32_t> _order;
};


// ---------------------------------------------------------------------------------------------------
// BVH4 with 8-bit quantised child boxes (the layout the traversal kernels use).
//
// The traversal kernels are bound by the number of divergent load instructions a CU's texture-address
// path can retire (every lane walks its own node), not by bytes.  A BVH2 step costs 4 x 16-byte loads
// for 2 child boxes; collapsing to 4 children with boxes quantised to 8 bits in the node's own frame keeps
// the node at 64 bytes = 4 loads while halving the number of steps per ray.
//
//   q0 = (origin.x, origin.y, origin.z, bits(ex | ey << 8 | ez << 16))     scale_a = 2^(e_a - 127)
//   q1 = bits(LOx, LOy, LOz, HIx)    one byte per child: child c's quantised bound = (W >> 8c) & 255
//   q2 = bits(HIy, HIz, 0, 0)
//   q3 = bits(child0, child1, child2, child3)   same encoding as the BVH2; 0xFFFFFFFF = empty slot
// box_c = [origin + LO*scale, origin + HI*scale] contains the (padded) child box; empty slots hold LO = 255,
// HI = 0 and can never be hit.  Box tests only steer the traversal: hits are decided by the exact triangle test.
struct Bvh4 {
    std::vector<Float4> nodes; // 4 per node
    uint32_t rootChild = 0x80000000u;
    int maxDepth = 0;          // in BVH4 levels
    uint32_t nodeCount = 0;
};

class Collapser {
public:
    explicit Collapser(const Bvh &b) : _b(b) {}

    Bvh4 run()
    {
        Bvh4 out;
        _out = &out;
        out.rootChild = _b.rootChild;
        if (!(_b.rootChild & 0x80000000u)) out.rootChild = emit(_b.rootChild, 1);
        out.nodeCount = uint32_t(out.nodes.size()/4);
        return out;
    }

private:
    struct Child { uint32_t ref; Box box; };

    static uint32_t bitsOf(float f) { uint32_t u; std::memcpy(&u, &f, 4); return u; }
    static float floatOf(uint32_t u) { float f; std::memcpy(&f, &u, 4); return f; }

    void children2(uint32_t node, Child out[2]) const
    {
        const Float4 *q = &_b.nodes[size_t(node)*4];
        out[0].box = Box{{q[0].x, q[0].y, q[0].z}, {q[0].w, q[1].x, q[1].y}};
        out[1].box = Box{{q[1].z, q[1].w, q[2].x}, {q[2].y, q[2].z, q[2].w}};
        out[0].ref = bitsOf(q[3].x);
        out[1].ref = bitsOf(q[3].y);
    }

    uint32_t emit(uint32_t node2, int depth)
    {
        _out->maxDepth = std::max(_out->maxDepth, depth);
        Child c[4];
        int n = 2;
        children2(node2, c);
        // open the inner child with the largest surface area until four slots are used
        while (n < 4) {
            int best = -1;
            float bestArea = -1.0f;
            for (int i = 0; i < n; ++i)
                if (!(c[i].ref & 0x80000000u) && c[i].box.halfArea() > bestArea) { bestArea = c[i].box.halfArea(); best = i; }
            if (best < 0) break;
            Child two[2];
            children2(c[best].ref, two);
            c[best] = two[0];
            c[n++] = two[1];
        }
        uint32_t idx = uint32_t(_out->nodes.size()/4);
        _out->nodes.resize(_out->nodes.size() + 4);
        uint32_t refs[4] = {0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu};
        for (int i = 0; i < n; ++i) refs[i] = (c[i].ref & 0x80000000u) ? c[i].ref : emit(c[i].ref, depth + 1);

        Box frame;
        frame.reset();
        for (int i = 0; i < n; ++i) frame.grow(c[i].box);
        uint32_t e[3];
        float scale[3];
        for (int a = 0; a < 3; ++a) {
            float ext = frame.hi[a] - frame.lo[a];
            int ex = 1;
            if (ext > 0.0f) {
                int fe;
                std::frexp(ext/255.0f, &fe); // ext/255 = m * 2^fe, m in [0.5, 1)  -> 2^fe >= ext/255
                ex = fe + 127;
            }
            ex = std::min(std::max(ex, 1), 254);
            // make sure 255 quanta really cover the extent after float rounding
            while (ex < 254 && frame.lo[a] + 255.0f*floatOf(uint32_t(ex) << 23) < frame.hi[a]) ++ex;
            e[a] = uint32_t(ex);
            scale[a] = floatOf(uint32_t(ex) << 23);
        }
        uint32_t lo[3] = {0, 0, 0}, hi[3] = {0, 0, 0};
        for (int i = 0; i < 4; ++i) {
            for (int a = 0; a < 3; ++a) {
                uint32_t ql = 255, qh = 0; // empty slot: inverted box
                if (i < n) {
                    float o = frame.lo[a], s = scale[a];
                    int l = int(std::floor((c[i].box.lo[a] - o)/s));
                    l = std::min(std::max(l, 0), 255);
                    while (l > 0 && o + float(l)*s > c[i].box.lo[a]) --l;
                    int h = int(std::ceil((c[i].box.hi[a] - o)/s));
                    h = std::min(std::max(h, 0), 255);
                    while (h < 255 && o + float(h)*s < c[i].box.hi[a]) ++h;
                    ql = uint32_t(l);
                    qh = uint32_t(h);
                }
                lo[a] |= ql << (8*i);
                hi[a] |= qh << (8*i);
            }
        }
        Float4 *q = &_out->nodes[size_t(idx)*4];
        q[0] = {frame.lo[0], frame.lo[1], frame.lo[2], floatOf(e[0] | (e[1] << 8) | (e[2] << 16))};
        q[1] = {floatOf(lo[0]), floatOf(lo[1]), floatOf(lo[2]), floatOf(hi[0])};
        q[2] = {floatOf(hi[1]), floatOf(hi[2]), 0.0f, 0.0f};
        q[3] = {floatOf(refs[0]), floatOf(refs[1]), floatOf(refs[2]), floatOf(refs[3])};
        return idx;
    }

    const Bvh &_b;
    Bvh4 *_out = nullptr;
};

// Renumber the nodes of a BVH4 so that the top `levels` levels (root = level 0) come first, in breadth-first order: those are the
// nodes the traversal kernels may stage in LDS ("nodelets").  Returns how many nodes the top levels hold.
inline uint32_t reorderTopLevels(Bvh4 &b, int levels)
{
    if (b.rootChild & 0x80000000u) return 0; // the root is a leaf: no inner nodes
    auto bitsOf = [](float f) { uint32_t u; std::memcpy(&u, &f, 4); return u; };
    auto floatOf = [](uint32_t u) { float f; std::memcpy(&f, &u, 4); return f; };
    const uint32_t n = uint32_t(b.nodes.size()/4);
    std::vector<uint32_t> order; // new index -> old index
    std::vector<uint32_t> frontier = {b.rootChild};
    for (int l = 0; l < levels && !frontier.empty(); ++l) {
        std::vector<uint32_t> next;
        for (uint32_t node : frontier) {
            order.push_back(node);
            const Float4 &q3 = b.nodes[size_t(node)*4 + 3];
            const float c[4] = {q3.x, q3.y, q3.z, q3.w};
            for (float f : c) {
                uint32_t ref = bitsOf(f);
                if (ref != 0xFFFFFFFFu && !(ref & 0x80000000u)) next.push_back(ref);
            }
        }
        frontier.swap(next);
    }
    const uint32_t top = uint32_t(order.size());
    std::vector<uint32_t> newIndex(n, 0xFFFFFFFFu);
    for (uint32_t i = 0; i < top; ++i) newIndex[order[i]] = i;
    uint32_t nextFree = top;
    for (uint32_t old = 0; old < n; ++old)
        if (newIndex[old] == 0xFFFFFFFFu) { newIndex[old] = nextFree++; order.push_back(old); }
    std::vector<Float4> nodes(b.nodes.size());
    for (uint32_t i = 0; i < n; ++i) {
        for (int k = 0; k < 4; ++k) nodes[size_t(i)*4 + k] = b.nodes[size_t(order[i])*4 + k];
        Float4 &q3 = nodes[size_t(i)*4 + 3];
        float *c[4] = {&q3.x, &q3.y, &q3.z, &q3.w};
        for (float *f : c) {
            uint32_t ref = bitsOf(*f);
            if (ref != 0xFFFFFFFFu && !(ref & 0x80000000u)) *f = floatOf(newIndex[ref]);
        }
    }
    b.nodes.swap(nodes);
    b.rootChild = newIndex[b.rootChild];
    return top;
}

} // namespace tgpu_bvh
