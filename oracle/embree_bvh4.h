// TEST INFRASTRUCTURE (part of the oracle; nothing under tungsten_amd/ includes this file).
//
// Restatement of the parts of Embree 2.11.0 (vendored by the reference under src/thirdparty/embree) that decide WHICH primitive the reference's
// TraceableScene::intersect reports when binary32 arithmetic leaves room: the BVH4 the reference's build of Embree constructs over the scene's
// finites (one user geometry, one primitive per leaf) and over every TriangleMesh's triangles (Triangle4 leaves), and the single-ray traversal
// that walks them.  Used only by the oracle's reference mode (rcp_mode = 1); the GPU contract does not depend on a tree.
//
//   builder    kernels/builders/bvh_builder_sah.h:84-322 (GeneralBVHBuilder::recurse / createLargeLeaf), heuristic_binning.h:29-385 (BinMapping, BinInfo::bin /
//              best), heuristic_binning_array_aligned.h:88-231 (sequential_find / sequential_split / splitFallback / deterministic_order),
//              algorithms/parallel_partition.h:23-65 (serial_partitioning), priminfo.h:25-131, bvh/bvh_builder_sah.cpp:44,150-215,810-815,
//              bvh/bvh.h:100-105,128-145, common/state.cpp:83-84
//   traversal  kernels/bvh/bvh_intersector1.cpp:46-150, bvh_intersector_node.h:28-47,162-190, bvh_traverser1.h:33-98, common/stack_item.h:39-60,
//              common/math/vec3fa.h:124-145 (rcp_safe), constants.h:30
//   leaves     geometry/object_intersector1.h (user geometry: the primitive's own intersect), geometry/triangle_intersector_moeller.h:76-113 +
//              intersector_epilog.h:201-270 (Triangle4: four triangles per block, smallest t of the block, lowest lane on a tie; a later block or leaf
//              must pass T < |den|*tfar, so the triangle found FIRST keeps an exact tie)
//
// What the reference's build selects (-march=core2 -mssse3, Embree's ISA cap SSE4.2): builders from the default-ISA library
// (no SSE4.1: floori(x) = cvtps2dq(x - 0.5), vfloat4_sse2.h:366-372), BVH4Triangle4Intersector1Moeller from the default ISA (float max/min in the node
// test), BVH4VirtualIntersector1 from the SSE4.2 library (integer max/min on the float bits, bvh_intersector_node.h:176-180).  Scenes are
// RTC_SCENE_STATIC | RTC_SCENE_INCOHERENT (TraceableScene.hpp:108, TriangleMesh.cpp:531): BVH4Triangle4ObjectSplit, sahBlockSize 4, leaves of 4..28
// triangles; user geometry: minLeafSize = maxLeafSize = 1.  Sets below 3072 primitives take the sequential find / split; above, the parallel ones
// give the same sets (bin merges are min / max / integer sums; the partition only changes the order inside a set, which the builder sorts away
// before it makes a leaf).
//
// Pinned by tests/golden/ref_kat5.* (visit orders and hits recorded from the compiled Embree of the survey-phase build, tests/tools/ref_embree_probe.cpp).
#pragma once
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstring>
#include <limits>
#include <vector>
#if defined(__SSE__)
#include <xmmintrin.h>
#include <emmintrin.h>
#endif

namespace embree_restated {

struct Box { float lo[3], hi[3]; };

inline Box emptyBox()
{
    Box b;
    for (int k = 0; k < 3; ++k) { b.lo[k] = std::numeric_limits<float>::infinity(); b.hi[k] = -std::numeric_limits<float>::infinity(); }
    return b;
}
inline void extend(Box &a, const Box &b) { for (int k = 0; k < 3; ++k) { a.lo[k] = std::min(a.lo[k], b.lo[k]); a.hi[k] = std::max(a.hi[k], b.hi[k]); } }
inline void extendPoint(Box &a, const float *p) { for (int k = 0; k < 3; ++k) { a.lo[k] = std::min(a.lo[k], p[k]); a.hi[k] = std::max(a.hi[k], p[k]); } }
inline float halfArea(const Box &b) // common/math/vec3fa.h:279 on b.size()
{
    float dx = b.hi[0] - b.lo[0], dy = b.hi[1] - b.lo[1], dz = b.hi[2] - b.lo[2];
    return dx*(dy + dz) + dy*dz;
}
inline float area(const Box &b) { return 2.0f*halfArea(b); }

inline int floori(float x) // vfloat4_sse2.h:366-372 without SSE4.1: vint4(a - 0.5f), a round-to-nearest-even conversion
{
#if defined(__SSE2__)
    return _mm_cvtss_si32(_mm_set_ss(x - 0.5f));
#else
    return int(std::lrintf(x - 0.5f));
#endif
}

inline float rcpNewton(float a) // common/math/vec3fa.h:124-133: RCPPS and one Newton-Raphson step (no FMA in the reference build)
{
#if defined(__SSE__)
    float r = _mm_cvtss_f32(_mm_rcp_ss(_mm_set_ss(a)));
    return r*(2.0f - r*a);
#else
    return 1.0f/a;
#endif
}

class Bvh4 {
public:
    static constexpr int32_t EMPTY = std::numeric_limits<int32_t>::min();
    struct Node {
        float lo[3][4], hi[3][4];
        int32_t child[4]; // >= 0: node, < 0 and != EMPTY: leaf ~child
    };
    struct Leaf { uint32_t first, count; }; // into prims(): primitive ids in the leaf's order
    struct Ray { float org[3], dir[3]; float tnear, tfar; };

    void build(const std::vector<Box> &boxes, size_t blockSize, size_t minLeafSize, size_t maxLeafSize)
    {
        _nodes.clear(); _leaves.clear(); _leafPrims.clear(); _refs.clear();
        _root = EMPTY;
        _logBlockSize = 0;
        while ((size_t(1) << (_logBlockSize + 1)) <= blockSize) _logBlockSize++;
        _minLeafSize = minLeafSize;
        _maxLeafSize = maxLeafSize;
        PrimInfo pinfo = emptyInfo(0, 0);
        for (uint32_t i = 0; i < boxes.size(); ++i) { // createPrimRefArray (builders/primrefgen.cpp:27-68): invalid primitives are left out
            const Box &b = boxes[i];
            bool valid = true;
            for (int k = 0; k < 3; ++k) valid = valid && std::isfinite(b.lo[k]) && std::isfinite(b.hi[k]);
            if (!valid) continue;
            _refs.push_back(Ref{b, i});
            extend(pinfo.geom, b);
            extendPoint(pinfo.cent, center2(b).data);
        }
        pinfo.end = _refs.size();
        if (_refs.empty()) return;
        Record br;
        br.depth = 1;
        br.pinfo = pinfo;
        br.split = find(pinfo);
        _root = recurse(br);
    }

    bool empty() const { return _root == EMPTY; }
    const std::vector<Node> &nodes() const { return _nodes; }
    const std::vector<Leaf> &leaves() const { return _leaves; }
    const std::vector<uint32_t> &prims() const { return _leafPrims; }
    int32_t root() const { return _root; }

    // BVHNIntersector1::intersect (bvh_intersector1.cpp:46-150).  leafFn(first, count, ray) intersects the leaf's primitives and lowers ray.tfar when it
    // finds a hit.  integerMinMax selects the SSE4.1 form of the node test (the user-geometry intersector of the reference build).
    template <class LeafFn>
    void intersect(Ray &ray, bool integerMinMax, LeafFn leafFn, uint64_t *nodeVisits = nullptr) const
    {
        if (_root == EMPTY) return;
        struct Item { int32_t ref; uint32_t dist; };
        Item stack[1 + 3*(48 + 48 + 32)]; // stackSize = 1+(N-1)*BVH::maxDepth
        int sp = 0;
        stack[sp++] = Item{_root, bits(-std::numeric_limits<float>::infinity())};
        float rdir[3];
        int nearIdx[3];
        for (int k = 0; k < 3; ++k) { // TravRay (bvh_intersector_node.h:28-47), rcp_safe (vec3fa.h:140-145)
            float a = ray.dir[k];
            if (std::fabs(a) < 1e-18f) a = 1e-18f;
            rdir[k] = rcpNewton(a);
            nearIdx[k] = rdir[k] >= 0.0f ? 0 : 1;
        }
        const float rayNear = std::max(ray.tnear, 0.0f);
        float rayFar = std::max(ray.tfar, 0.0f);
        while (sp) {
            Item it = stack[--sp];
            if (fromBits(it.dist) > ray.tfar) continue;
            int32_t cur = it.ref;
            bool popNext = false;
            while (cur >= 0) {
                const Node &n = _nodes[size_t(cur)];
                if (nodeVisits) ++*nodeVisits;
                float tNear[4];
                unsigned mask = 0;
                for (int c = 0; c < 4; ++c) {
                    float tn[3], tf[3];
                    for (int k = 0; k < 3; ++k) {
                        float nearPlane = nearIdx[k] ? n.hi[k][c] : n.lo[k][c], farPlane = nearIdx[k] ? n.lo[k][c] : n.hi[k][c];
                        tn[k] = (nearPlane - ray.org[k])*rdir[k];
                        tf[k] = (farPlane - ray.org[k])*rdir[k];
                    }
                    bool hit;
                    if (integerMinMax) { // maxi(maxi(x,y),maxi(z,tnear)), mini(...), asInt(tNear) > asInt(tFar) is a miss (:176-180)
                        int32_t a = std::max(std::max(ibits(tn[0]), ibits(tn[1])), std::max(ibits(tn[2]), ibits(rayNear)));
                        int32_t b = std::min(std::min(ibits(tf[0]), ibits(tf[1])), std::min(ibits(tf[2]), ibits(rayFar)));
                        hit = !(a > b);
                        tNear[c] = fromBits(uint32_t(a));
                    } else { // max(a,b,c,d) = max(max(a,b),max(c,d)) with MAXPS: the second operand on a NaN (:182-186)
                        float a = maxps(maxps(tn[0], tn[1]), maxps(tn[2], rayNear));
                        float b = minps(minps(tf[0], tf[1]), minps(tf[2], rayFar));
                        hit = a <= b;
                        tNear[c] = a;
                    }
                    if (hit) mask |= 1u << c;
                }
                if (mask == 0) { popNext = true; break; }
                // BVHNNodeTraverser1Hit<4,4>::traverseClosestHit (bvh_traverser1.h:40-95): children in ascending index, distances compared as unsigned bits
                int hitIdx[4], nHit = 0;
                for (int c = 0; c < 4; ++c) if (mask & (1u << c)) hitIdx[nHit++] = c;
                if (nHit == 1) { cur = n.child[hitIdx[0]]; continue; }
                Item c0{n.child[hitIdx[0]], bits(tNear[hitIdx[0]])}, c1{n.child[hitIdx[1]], bits(tNear[hitIdx[1]])};
                if (nHit == 2) {
                    if (c0.dist < c1.dist) { stack[sp++] = c1; cur = c0.ref; }
                    else { stack[sp++] = c0; cur = c1.ref; }
                    continue;
                }
                stack[sp++] = c0;
                stack[sp++] = c1;
                stack[sp++] = Item{n.child[hitIdx[2]], bits(tNear[hitIdx[2]])};
                if (nHit == 3) {
                    sort3(stack[sp - 1], stack[sp - 2], stack[sp - 3]);
                } else {
                    stack[sp++] = Item{n.child[hitIdx[3]], bits(tNear[hitIdx[3]])};
                    sort4(stack[sp - 1], stack[sp - 2], stack[sp - 3], stack[sp - 4]);
                }
                cur = stack[--sp].ref;
            }
            if (popNext) continue;
            const Leaf &leaf = _leaves[size_t(~cur)];
            leafFn(leaf.first, leaf.count, ray);
            rayFar = ray.tfar;
        }
    }

private:
    struct Vec3 { float data[3]; };
    struct Ref { Box b; uint32_t id; };
    struct PrimInfo { size_t begin, end; Box geom, cent; size_t size() const { return end - begin; } };
    struct Mapping { size_t num; float ofs[3], scale[3]; };
    struct Split { float sah = std::numeric_limits<float>::infinity(); int dim = -1; int pos = 0; Mapping mapping{}; bool valid() const { return dim != -1; } };
    struct Record { size_t depth = 0; PrimInfo pinfo{}; Split split; };

    static uint32_t bits(float f) { uint32_t u; std::memcpy(&u, &f, 4); return u; }
    static int32_t ibits(float f) { int32_t u; std::memcpy(&u, &f, 4); return u; }
    static float fromBits(uint32_t u) { float f; std::memcpy(&f, &u, 4); return f; }
    static float maxps(float a, float b) { return a > b ? a : b; } // MAXPS dst=a, src=b: a > b ? a : b (b on NaN)
    static float minps(float a, float b) { return a < b ? a : b; }
    template <class T> static void xchg(T &a, T &b) { T t = a; a = b; b = t; }
    template <class T> static void sort3(T &s1, T &s2, T &s3) // common/stack_item.h:45-50
    {
        if (s2.dist < s1.dist) xchg(s2, s1);
        if (s3.dist < s2.dist) xchg(s3, s2);
        if (s2.dist < s1.dist) xchg(s2, s1);
    }
    template <class T> static void sort4(T &s1, T &s2, T &s3, T &s4) // :53-60
    {
        if (s2.dist < s1.dist) xchg(s2, s1);
        if (s4.dist < s3.dist) xchg(s4, s3);
        if (s3.dist < s1.dist) xchg(s3, s1);
        if (s4.dist < s2.dist) xchg(s4, s2);
        if (s3.dist < s2.dist) xchg(s3, s2);
    }

    static Vec3 center2(const Box &b) { return Vec3{{b.lo[0] + b.hi[0], b.lo[1] + b.hi[1], b.lo[2] + b.hi[2]}}; }
    static PrimInfo emptyInfo(size_t begin, size_t end) { return PrimInfo{begin, end, emptyBox(), emptyBox()}; }

    Mapping makeMapping(const PrimInfo &pinfo) const // BinMapping (heuristic_binning.h:37-52)
    {
        Mapping m;
        m.num = std::min(size_t(32), size_t(4.0f + 0.05f*float(pinfo.size())));
        for (int k = 0; k < 3; ++k) {
            float diag = pinfo.cent.hi[k] - pinfo.cent.lo[k];
            m.scale[k] = diag > 1E-34f ? (0.99f*float(m.num))/diag : 0.0f;
            m.ofs[k] = pinfo.cent.lo[k];
        }
        return m;
    }
    static int binOf(const Mapping &m, const Vec3 &c2, int k) { return floori((c2.data[k] - m.ofs[k])*m.scale[k]); }

    Split find(const PrimInfo &pinfo) const // sequential_find :88-94, BinInfo::bin :215-258, BinInfo::best :321-385
    {
        const Mapping mapping = makeMapping(pinfo);
        const size_t B = 32;
        Box bounds[B][3];
        int counts[B][3];
        for (size_t i = 0; i < B; ++i) for (int k = 0; k < 3; ++k) { bounds[i][k] = emptyBox(); counts[i][k] = 0; }
        for (size_t i = pinfo.begin; i < pinfo.end; ++i) {
            const Box &b = _refs[i].b;
            Vec3 c2 = center2(b);
            for (int k = 0; k < 3; ++k) {
                unsigned bin = unsigned(binOf(mapping, c2, k));
                if (bin >= B) bin = bin > 0x7FFFFFFFu ? 0 : B - 1; // the reference asserts 0 <= bin < num; cannot happen with 0.99*num
                extend(bounds[bin][k], b);
                counts[bin][k]++;
            }
        }
        float rAreas[B][3];
        int rCounts[B][3];
        int count[3] = {0, 0, 0};
        Box bx[3] = {emptyBox(), emptyBox(), emptyBox()};
        for (size_t i = mapping.num - 1; i > 0; --i)
            for (int k = 0; k < 3; ++k) {
                count[k] += counts[i][k];
                rCounts[i][k] = count[k];
                extend(bx[k], bounds[i][k]);
                rAreas[i][k] = halfArea(bx[k]);
            }
        const int blocksAdd = (1 << _logBlockSize) - 1;
        float bestSAHk[3] = {std::numeric_limits<float>::infinity(), std::numeric_limits<float>::infinity(), std::numeric_limits<float>::infinity()};
        int bestPosk[3] = {0, 0, 0};
        for (int k = 0; k < 3; ++k) { count[k] = 0; bx[k] = emptyBox(); }
        for (size_t i = 1; i < mapping.num; ++i)
            for (int k = 0; k < 3; ++k) {
                count[k] += counts[i - 1][k];
                extend(bx[k], bounds[i - 1][k]);
                float lArea = halfArea(bx[k]), rArea = rAreas[i][k];
                int lCount = (count[k] + blocksAdd) >> _logBlockSize, rCount = (rCounts[i][k] + blocksAdd) >> _logBlockSize;
                float sah = lArea*float(lCount) + rArea*float(rCount);
                if (sah < bestSAHk[k]) { bestPosk[k] = int(i); bestSAHk[k] = sah; }
            }
        Split s;
        s.mapping = mapping;
        for (int k = 0; k < 3; ++k) {
            if (mapping.scale[k] == 0.0f) continue;
            if (bestSAHk[k] < s.sah && bestPosk[k] != 0) { s.dim = k; s.pos = bestPosk[k]; s.sah = bestSAHk[k]; }
        }
        return s;
    }

    void deterministicOrder(const PrimInfo &p) { std::sort(_refs.begin() + long(p.begin), _refs.begin() + long(p.end), [](const Ref &a, const Ref &b) { return a.id < b.id; }); }

    void splitFallback(const PrimInfo &p, PrimInfo &l, PrimInfo &r) const // :211-231
    {
        size_t center = (p.begin + p.end)/2;
        l = emptyInfo(p.begin, center);
        for (size_t i = p.begin; i < center; ++i) { extend(l.geom, _refs[i].b); extendPoint(l.cent, center2(_refs[i].b).data); }
        r = emptyInfo(center, p.end);
        for (size_t i = center; i < p.end; ++i) { extend(r.geom, _refs[i].b); extendPoint(r.cent, center2(_refs[i].b).data); }
    }

    void partition(const Record &rec, PrimInfo &l, PrimInfo &r) // sequential_split :125-165, serial_partitioning (parallel_partition.h:23-65)
    {
        const PrimInfo &p = rec.pinfo;
        if (!rec.split.valid()) {
            deterministicOrder(p);
            splitFallback(p, l, r);
            return;
        }
        const Split &s = rec.split;
        auto isLeft = [&](const Ref &ref) { return binOf(s.mapping, center2(ref.b), s.dim) < s.pos; };
        Box lg = emptyBox(), lc = emptyBox(), rg = emptyBox(), rc = emptyBox();
        auto addL = [&](const Ref &ref) { extend(lg, ref.b); extendPoint(lc, center2(ref.b).data); };
        auto addR = [&](const Ref &ref) { extend(rg, ref.b); extendPoint(rc, center2(ref.b).data); };
        long lo = long(p.begin), hi = long(p.end) - 1;
        while (true) {
            while (lo <= hi && isLeft(_refs[size_t(lo)])) { addL(_refs[size_t(lo)]); ++lo; }
            while (lo <= hi && !isLeft(_refs[size_t(hi)])) { addR(_refs[size_t(hi)]); --hi; }
            if (hi < lo) break;
            addL(_refs[size_t(hi)]);
            addR(_refs[size_t(lo)]);
            std::swap(_refs[size_t(lo)], _refs[size_t(hi)]);
            ++lo; --hi;
        }
        l = PrimInfo{p.begin, size_t(lo), lg, lc};
        r = PrimInfo{size_t(lo), p.end, rg, rc};
    }

    int32_t createNode(const Record *children, size_t n) // BVHN::CreateNode (bvh/bvh.h:128-145)
    {
        Node node;
        for (int c = 0; c < 4; ++c) {
            for (int k = 0; k < 3; ++k) { node.lo[k][c] = std::numeric_limits<float>::infinity(); node.hi[k][c] = -std::numeric_limits<float>::infinity(); }
            node.child[c] = EMPTY;
        }
        for (size_t i = 0; i < n; ++i)
            for (int k = 0; k < 3; ++k) { node.lo[k][i] = children[i].pinfo.geom.lo[k]; node.hi[k][i] = children[i].pinfo.geom.hi[k]; }
        _nodes.push_back(node);
        return int32_t(_nodes.size() - 1);
    }

    int32_t createLeaf(const Record &cur)
    {
        Leaf leaf{uint32_t(_leafPrims.size()), uint32_t(cur.pinfo.size())};
        for (size_t i = cur.pinfo.begin; i < cur.pinfo.end; ++i) _leafPrims.push_back(_refs[i].id);
        _leaves.push_back(leaf);
        return ~int32_t(_leaves.size() - 1);
    }

    int32_t createLargeLeaf(const Record &cur) // bvh_builder_sah.h:99-158
    {
        if (cur.pinfo.size() <= _maxLeafSize) return createLeaf(cur);
        Record children[16];
        size_t numChildren = 1;
        children[0] = cur;
        do {
            long bestChild = -1;
            size_t bestSize = 0;
            for (size_t i = 0; i < numChildren; ++i) {
                if (children[i].pinfo.size() <= _maxLeafSize) continue;
                if (children[i].pinfo.size() > bestSize) { bestSize = children[i].pinfo.size(); bestChild = long(i); }
            }
            if (bestChild == -1) break;
            Record left, right;
            left.depth = right.depth = cur.depth + 1;
            splitFallback(children[bestChild].pinfo, left.pinfo, right.pinfo);
            left.split = find(left.pinfo);
            right.split = find(right.pinfo);
            children[bestChild] = children[numChildren - 1];
            children[numChildren - 1] = left;
            children[numChildren] = right;
            numChildren++;
        } while (numChildren < 4);
        int32_t node = createNode(children, numChildren);
        for (size_t i = 0; i < numChildren; ++i) {
            int32_t ref = createLargeLeaf(children[i]);
            _nodes[size_t(node)].child[i] = ref;
        }
        return node;
    }

    int32_t recurse(Record &cur) // bvh_builder_sah.h:168-270; travCost = intCost = 1 (bvh_builder_sah.cpp:44,814)
    {
        const size_t maxDepth = 48, MIN_LARGE_LEAF_LEVELS = 8;
        const float leafSAH = 1.0f*(halfArea(cur.pinfo.geom)*float((cur.pinfo.size() + (size_t(1) << _logBlockSize) - 1) >> _logBlockSize));
        const float splitSAH = 1.0f*halfArea(cur.pinfo.geom) + 1.0f*cur.split.sah;
        if (cur.pinfo.size() <= _minLeafSize || cur.depth + MIN_LARGE_LEAF_LEVELS >= maxDepth || (cur.pinfo.size() <= _maxLeafSize && leafSAH <= splitSAH)) {
            deterministicOrder(cur.pinfo);
            return createLargeLeaf(cur);
        }
        Record children[16];
        children[0] = cur;
        size_t numChildren = 1;
        do {
            float bestSAH = -std::numeric_limits<float>::infinity();
            long bestChild = -1;
            for (size_t i = 0; i < numChildren; ++i) {
                if (children[i].pinfo.size() <= _minLeafSize) continue;
                if (area(children[i].pinfo.geom) > bestSAH) { bestChild = long(i); bestSAH = area(children[i].pinfo.geom); }
            }
            if (bestChild == -1) break;
            Record l, r;
            l.depth = r.depth = cur.depth + 1;
            partition(children[bestChild], l.pinfo, r.pinfo);
            l.split = find(l.pinfo);
            r.split = find(r.pinfo);
            children[bestChild] = l;
            children[numChildren] = r;
            numChildren++;
        } while (numChildren < 4);
        // std::sort(children, children + n, std::greater<>) on at most 4 records is libstdc++'s insertion sort: stable
        std::stable_sort(children, children + numChildren, [](const Record &a, const Record &b) { return a.pinfo.size() > b.pinfo.size(); });
        int32_t node = createNode(children, numChildren);
        for (long i = long(numChildren) - 1; i >= 0; --i) {
            int32_t ref = recurse(children[i]);
            _nodes[size_t(node)].child[i] = ref;
        }
        return node;
    }

    std::vector<Node> _nodes;
    std::vector<Leaf> _leaves;
    std::vector<uint32_t> _leafPrims;
    std::vector<Ref> _refs;
    int32_t _root = EMPTY;
    size_t _logBlockSize = 0, _minLeafSize = 1, _maxLeafSize = 1;
};

} // namespace embree_restated
