// Host-side build of the BVH2 an "instances" primitive traverses.
//
// Unlike the triangle BVH (bvh_build.h, free to take any shape because nearest-hit results do not depend on it)
// this tree is part of the RESULT: Instance::intersect (src/core/primitives/Instance.cpp:290-311) starts every local
// ray at the leaf's entry distance with farT = infinity and lets the instance that is hit LAST in traversal order
// win.  The tree therefore has to be the one Bvh::BinaryBvh(prims, 2) builds:
//   - Instance::prepareForRender (:405-424): world box of each instance from the 8 rotated corners of its master's box
//   - BvhBuilder(2).build (bvh/BvhBuilder.cpp:99-185): exact sweep SAH for <= 64 primitives (FullSahSplitter.hpp),
//     32-bin SAH above (BinnedSahSplitter.hpp), median fallbacks; IntersectionCost = TraversalCost = 1
//   - BinaryBvh::recursiveBuild (bvh/BinaryBvh.hpp:120-146): depth-first array layout, two-primitive leaves
// Work lists are explicit (no recursion, no per-node heap objects); every float expression keeps the reference's
// operand order (compiled with -ffp-contract=off like the rest of the core).
#pragma once
#include "../../include/tungsten_amd.h"

#include <algorithm>
#include <cstdint>
#include <cstring>
#include <limits>
#include <vector>

namespace tgpu_inst {

struct F3 { float v[3]; float &operator[](int i) { return v[i]; } float operator[](int i) const { return v[i]; } };

struct Aabb {
    F3 lo{{std::numeric_limits<float>::max(), std::numeric_limits<float>::max(), std::numeric_limits<float>::max()}};
    F3 hi{{std::numeric_limits<float>::lowest(), std::numeric_limits<float>::lowest(), std::numeric_limits<float>::lowest()}};
    void add(const F3 &p) { for (int k = 0; k < 3; ++k) { lo[k] = lo[k] < p[k] ? lo[k] : p[k]; hi[k] = hi[k] > p[k] ? hi[k] : p[k]; } }
    void add(const Aabb &b) { for (int k = 0; k < 3; ++k) { lo[k] = lo[k] < b.lo[k] ? lo[k] : b.lo[k]; hi[k] = hi[k] > b.hi[k] ? hi[k] : b.hi[k]; } }
    F3 extent() const // Box::diagonal: max(max - min, 0)
    {
        F3 d;
        for (int k = 0; k < 3; ++k) { float e = hi[k] - lo[k]; d[k] = e > 0.0f ? e : 0.0f; }
        return d;
    }
    float surface() const { F3 d = extent(); return (d[0]*d[1] + d[1]*d[2] + d[2]*d[0])*2.0f; } // Box::area
};

inline int widestAxis(const F3 &d) { int a = 0; if (d[1] > d[a]) a = 1; if (d[2] > d[a]) a = 2; return a; }

// Quaternion::operator*(Vec3) (math/Quaternion.hpp:74-84); q = w x y z
inline F3 rotate(const float *q, const F3 &o)
{
    float tx = 2.0f*(q[2]*o[2] - q[3]*o[1]);
    float ty = 2.0f*(q[3]*o[0] - q[1]*o[2]);
    float tz = 2.0f*(q[1]*o[1] - q[2]*o[0]);
    return F3{{o[0] + q[0]*tx + q[2]*tz - q[3]*ty, o[1] + q[0]*ty + q[3]*tx - q[1]*tz, o[2] + q[0]*tz + q[1]*ty - q[2]*tx}};
}

struct Item { Aabb box; F3 centroid; uint32_t id; float suffixArea; };

// One node of the array the kernels traverse: the two children's boxes side by side (TinyBvhNode::setJointBbox),
// child indices, or -- for a leaf (right < 0) -- the range in `order`.
struct Node { float bx[4], by[4], bz[4]; int32_t left, right; uint32_t first, count; };

struct Tree {
    std::vector<Node> nodes;
    std::vector<uint32_t> order; // BinaryBvh::_primIndices
    Aabb bounds;                 // BinaryBvh::_bounds
    int depth = 0;               // BvhBuilder::depth()
};

class TreeBuilder {
public:
    // world-space boxes of the instances (Instance.cpp:405-420)
    static std::vector<Item> instanceItems(const tgpu_instances &set)
    {
        std::vector<Item> items(set.n_instances);
        for (uint32_t i = 0; i < set.n_instances; ++i) {
            const tgpu_master &m = set.masters[set.master_id[i]];
            const float *q = set.rot + size_t(i)*4, *t = set.pos + size_t(i)*3;
            Aabb world;
            for (int cx = 0; cx < 2; ++cx)
                for (int cy = 0; cy < 2; ++cy)
                    for (int cz = 0; cz < 2; ++cz) {
                        float r[3] = {float(cx), float(cy), float(cz)};
                        F3 corner;
                        for (int k = 0; k < 3; ++k) corner[k] = m.bounds_min[k]*(1.0f - r[k]) + m.bounds_max[k]*r[k]; // lerp()
                        F3 w = rotate(q, corner);
                        world.add(F3{{t[0] + w[0], t[1] + w[1], t[2] + w[2]}});
                    }
            items[i].box = world;
            for (int k = 0; k < 3; ++k) items[i].centroid[k] = (world.lo[k] + world.hi[k])/2.0f; // Box::center
            items[i].id = i;
            items[i].suffixArea = world.surface();
        }
        return items;
    }

    Tree build(std::vector<Item> items)
    {
        Tree out;
        if (items.empty()) { // BinaryBvh.hpp:178-182: a lone inner node whose box never passes
            Node n{};
            Aabb e;
            joint(n, e, e);
            n.left = n.right = 0;
            out.nodes.push_back(n);
            return out;
        }
        _items.swap(items);
        Aabb geom, cent;
        for (const Item &it : _items) { geom.add(it.box); cent.add(it.centroid); }
        // ---- phase 1: the pointer tree of BvhBuilder::recursiveBuild as an array of build records
        _recs.clear();
        _recs.push_back(Rec{geom, cent, 0, uint32_t(_items.size() - 1), -1, -1, 0, 1});
        std::vector<uint32_t> todo{0};
        while (!todo.empty()) {
            uint32_t ri = todo.back();
            todo.pop_back();
            const uint32_t s = _recs[ri].start, e = _recs[ri].end, n = e - s + 1;
            if (n == 1) {
                _recs[ri].id = _items[s].id;
            } else if (n <= 2) {
                for (uint32_t i = s; i <= e; ++i) {
                    Rec leaf{_items[i].box, Aabb(), i, i, -1, -1, _items[i].id, 1};
                    (i == s ? _recs[ri].c0 : _recs[ri].c1) = int32_t(_recs.size());
                    _recs.push_back(leaf);
                }
            } else {
                Cut cut = n <= 64 ? sweepCut(s, e, _recs[ri].geom) : binnedCut(s, e, _recs[ri].geom, _recs[ri].cent);
                Rec l{cut.lBox, cut.lCent, s, cut.at - 1, -1, -1, 0, 1}, r{cut.rBox, cut.rCent, cut.at, e, -1, -1, 0, 1};
                _recs[ri].c0 = int32_t(_recs.size());
                _recs.push_back(l);
                _recs[ri].c1 = int32_t(_recs.size());
                _recs.push_back(r);
                // the reference recurses into child 0 first; the partition of [s, e] is final here, and the two halves
                // are disjoint, so the processing order of the halves does not change the result
                todo.push_back(uint32_t(_recs[ri].c1));
                todo.push_back(uint32_t(_recs[ri].c0));
            }
        }
        out.depth = depthOf(0);
        out.bounds = _recs[0].geom;
        // ---- phase 2: BinaryBvh::recursiveBuild
        out.order.resize(_items.size());
        out.nodes.assign(_recs.size(), Node{});
        uint32_t tail = 1, cursor = 0;
        flattenInto(out, 0, 0, tail, cursor);
        out.nodes.resize(tail);
        return out;
    }

private:
    struct Rec { Aabb geom, cent; uint32_t start, end; int32_t c0, c1; uint32_t id; int depth; };
    struct Cut { Aabb lBox, rBox, lCent, rCent; int axis; uint32_t at; float cost; };

    static void joint(Node &n, const Aabb &l, const Aabb &r)
    {
        n.bx[0] = l.lo[0]; n.bx[1] = r.lo[0]; n.bx[2] = l.hi[0]; n.bx[3] = r.hi[0];
        n.by[0] = l.lo[1]; n.by[1] = r.lo[1]; n.by[2] = l.hi[1]; n.by[3] = r.hi[1];
        n.bz[0] = l.lo[2]; n.bz[1] = r.lo[2]; n.bz[2] = l.hi[2]; n.bz[3] = r.hi[2];
    }

    int depthOf(uint32_t root) const
    {
        // result.depth = max(child depth + 1), leaves and two-primitive nodes count 1 (BvhBuilder.cpp:104,146)
        std::vector<int> d(_recs.size(), 1);
        for (size_t i = _recs.size(); i-- > 0;) {
            const Rec &r = _recs[i];
            if (r.c0 >= 0 && r.end - r.start + 1 > 2) d[i] = std::max(d[size_t(r.c0)], d[size_t(r.c1)]) + 1;
        }
        return d[root];
    }

    void orderBy(uint32_t s, uint32_t e, int axis)
    {
        std::sort(_items.begin() + s, _items.begin() + e + 1, [axis](const Item &a, const Item &b) {
            return a.centroid[axis] == b.centroid[axis] ? a.id < b.id : a.centroid[axis] < b.centroid[axis];
        });
    }

    // FullSahSplitter::twoWaySahSplit
    Cut sweepCut(uint32_t s, uint32_t e, const Aabb &geom)
    {
        Cut cut{};
        cut.axis = -1;
        cut.cost = geom.surface()*(float(e - s + 1)*1.0f - 1.0f);
        for (int axis = 0; axis < 3; ++axis) {
            orderBy(s, e, axis);
            Aabb suffix;
            for (uint32_t i = e; i > s; --i) { suffix.add(_items[i].box); _items[i].suffixArea = suffix.surface(); }
            suffix.add(_items[s].box);
            _items[s].suffixArea = suffix.surface();
            Aabb prefix = _items[s].box;
            for (uint32_t i = s + 1; i <= e; ++i) {
                float c = 1.0f*(prefix.surface()*float(i - s) + _items[i].suffixArea*float(e - i + 1));
                if (c < cut.cost) { cut.axis = axis; cut.at = i; cut.lBox = prefix; cut.cost = c; }
                prefix.add(_items[i].box);
            }
            if (cut.axis == axis) {
                Aabb r;
                for (uint32_t i = cut.at; i <= e; ++i) r.add(_items[i].box);
                cut.rBox = r;
            }
        }
        if (cut.axis == -1) {
            cut.axis = widestAxis(geom.extent());
            cut.at = (e - s + 1)/2 + s;
            orderBy(s, e, cut.axis);
            for (uint32_t i = s; i <= e; ++i) (i < cut.at ? cut.lBox : cut.rBox).add(_items[i].box);
        } else if (cut.axis != 2) {
            orderBy(s, e, cut.axis);
        }
        return cut; // centroid boxes stay empty: nodes of <= 64 primitives never read them
    }

    static int truncToInt(float f) // x86 cvttss2si: NaN / out of range -> INT_MIN
    {
        if (!(f > -2147483904.0f && f < 2147483648.0f)) return std::numeric_limits<int>::min();
        return int(f);
    }

    // BinnedSahSplitter::fullSplit
    Cut binnedCut(uint32_t s, uint32_t e, const Aabb &geom, const Aabb &cent)
    {
        const int B = 32;
        struct Bins { Aabb geom[32], cent[32]; int count[32]; };
        std::vector<Bins> bins(3);
        for (Bins &b : bins) std::memset(b.count, 0, sizeof(b.count));
        const F3 origin = cent.lo, span = cent.extent();
        auto binOf = [&](const Item &it, int axis) {
            int b = truncToInt(float(B)*((it.centroid[axis] - origin[axis])/span[axis]));
            return b < 0 ? 0 : (b > B - 1 ? B - 1 : b);
        };
        for (int axis = 0; axis < 3; ++axis)
            if (span[axis] > 0.0f)
                for (uint32_t i = s; i <= e; ++i) {
                    int b = binOf(_items[i], axis);
                    bins[size_t(axis)].geom[b].add(_items[i].box);
                    bins[size_t(axis)].cent[b].add(_items[i].centroid);
                    bins[size_t(axis)].count[b]++;
                }
        Cut cut{};
        cut.axis = -1;
        cut.cost = geom.surface()*(float(e - s + 1)*1.0f - 1.0f);
        for (int axis = 0; axis < 3; ++axis) {
            if (!(span[axis] > 0.0f)) continue;
            const Bins &bn = bins[size_t(axis)];
            int rCount = 0, rCounts[32];
            Aabb rBox, rBoxes[32];
            for (int i = B - 1; i > 0; --i) {
                rCount += bn.count[i];
                rBox.add(bn.geom[i]);
                rCounts[i] = rCount;
                rBoxes[i] = rBox;
            }
            int lCount = bn.count[0];
            Aabb lBox = bn.geom[0];
            for (int i = 1; i < B; ++i) {
                float c = 1.0f*(lBox.surface()*float(lCount) + rBoxes[i].surface()*float(rCounts[i]));
                if (c < cut.cost) { cut.axis = axis; cut.at = uint32_t(i); cut.cost = c; }
                lCount += bn.count[i];
                lBox.add(bn.geom[i]);
            }
        }
        if (cut.axis == -1) { cut.axis = widestAxis(geom.extent()); cut.at = B/2; }
        const int bin = int(cut.at);
        // sortByBin
        uint32_t left = s, right = e;
        while (left < right) {
            while (left < right && binOf(_items[left], cut.axis) < bin) left++;
            while (right > left && binOf(_items[right], cut.axis) >= bin) right--;
            if (left != right) std::swap(_items[left], _items[right]);
        }
        cut.at = (left == e || right == s) ? e + 1 : left;
        const bool flat = span[0] == 0.0f && span[1] == 0.0f && span[2] == 0.0f;
        if (cut.at > e || flat) {
            cut.at = s + (e - s + 1)/2;
            cut.lBox = _items[s].box;
            cut.rBox = _items[e].box;
            cut.lCent = Aabb(); cut.lCent.lo = cut.lCent.hi = _items[s].centroid;
            cut.rCent = Aabb(); cut.rCent.lo = cut.rCent.hi = _items[e].centroid;
            for (uint32_t i = s + 1; i < e; ++i) {
                if (i < cut.at) { cut.lBox.add(_items[i].box); cut.lCent.add(_items[i].centroid); }
                else { cut.rBox.add(_items[i].box); cut.rCent.add(_items[i].centroid); }
            }
        } else {
            const Bins &bn = bins[size_t(cut.axis)];
            cut.lBox = bn.geom[0];
            cut.rBox = bn.geom[B - 1];
            cut.lCent = bn.cent[0];
            cut.rCent = bn.cent[B - 1];
            for (int i = 1; i < B - 1; ++i) {
                if (i < bin) { cut.lBox.add(bn.geom[i]); cut.lCent.add(bn.cent[i]); }
                else { cut.rBox.add(bn.geom[i]); cut.rCent.add(bn.cent[i]); }
            }
        }
        return cut;
    }

    // BinaryBvh::recursiveBuild with an explicit stack: (record, array slot) pairs; children of a node take the next
    // two slots, a subtree of <= 2 primitives collapses into its root slot and gives its slots back.
    void flattenInto(Tree &out, uint32_t rootRec, uint32_t rootSlot, uint32_t &tail, uint32_t &cursor)
    {
        struct Frame { uint32_t rec, slot, childSlot; int stage; uint32_t lPrims; };
        std::vector<Frame> st{Frame{rootRec, rootSlot, 0, 0, 0}};
        uint32_t ret = 0;
        while (!st.empty()) {
            Frame &f = st.back();
            const Rec &r = _recs[f.rec];
            Node &n = out.nodes[f.slot];
            if (r.c0 < 0) { // NaiveBvhNode::isLeaf
                joint(n, r.geom, r.geom);
                n.first = cursor; n.count = 1; n.right = -1;
                out.order[cursor++] = r.id;
                ret = 1;
                st.pop_back();
                continue;
            }
            if (f.stage == 0) {
                f.childSlot = tail;
                joint(n, _recs[size_t(r.c0)].geom, _recs[size_t(r.c1)].geom);
                n.left = int32_t(tail);
                n.right = int32_t(tail + 1);
                tail += 2;
                f.stage = 1;
                Frame child{uint32_t(r.c0), f.childSlot, 0, 0, 0};
                st.push_back(child);
            } else if (f.stage == 1) {
                f.lPrims = ret;
                f.stage = 2;
                Frame child{uint32_t(r.c1), f.childSlot + 1, 0, 0, 0};
                st.push_back(child);
            } else {
                uint32_t total = f.lPrims + ret;
                if (total <= 2) {
                    n.count = total; n.first = out.nodes[f.childSlot].first; n.right = -1;
                    tail = f.childSlot;
                }
                ret = total;
                st.pop_back();
            }
        }
    }

    std::vector<Item> _items;
    std::vector<Rec> _recs;
};

} // namespace tgpu_inst
