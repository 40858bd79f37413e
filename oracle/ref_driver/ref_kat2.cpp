// Test-infrastructure driver (NOT product code), second set.  Compiled against the reference's own sources where they lie under
// /root/reference (no copies; oracle/Makefile.ref), it prints as JSON:
//   binary_bvh     bvh/BvhBuilder.cpp + bvh/BinaryBvh.hpp: for two box sets (50 boxes: full-sweep SAH; 300: binned SAH) and 64 rays
//                  each, the sequence of (primitive id, bits(tMin)) BinaryBvh::trace hands its intersector -- the order
//                  Instance::intersect depends on (the last hit wins) -- with a callback that shortens the ray like a hit would
//   filter         cameras/ReconstructionFilter.cpp: sample() of the tent filter at 256 points (it walks the private _cdf[] table)
//   envmap         thirdparty/stbi decoding assets/materialtest/envmap.hdr -> sha-like digest of the texel bits; the weights of
//                  BitmapTexture::makeSamplable(MAP_SPHERICAL) (restated here: BitmapTexture.cpp does not compile standalone) fed to the
//                  reference's sampling/Distribution2D.hpp -> digest of pdf(row, column) over all cells, warp() of 512 points
//   rot_yxz        math/Mat4f.cpp rotYXZ for 8 rotations (row-major 3x3 bits)
#include <algorithm>
#include <array>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <functional>
#include <iostream>
#include <memory>
#include <sstream>
#include <string>
#include <unordered_map>
#include <vector>
#include <math/Vec.hpp>
#include <math/MathUtil.hpp>
#include <math/Ray.hpp>
#include <StringableEnum.hpp>
#include <cameras/ReconstructionFilter.hpp>
#include <sampling/Distribution2D.hpp>
#include <bvh/BinaryBvh.hpp>
#include <sampling/UniformSampler.hpp>
#include <math/Mat4f.hpp>
#include <math/Angle.hpp>
#include <stbi/stb_image.h>

using namespace Tungsten;

static unsigned bits(float f) { unsigned u; std::memcpy(&u, &f, 4); return u; }
// FNV-1a over 32-bit words: the digest the test recomputes from our own tables
static unsigned long long digest(const float *p, size_t n)
{
    unsigned long long h = 1469598103934665603ull;
    for (size_t i = 0; i < n; ++i) { h ^= bits(p[i]); h *= 1099511628211ull; }
    return h;
}

static void bvhSet(UniformSampler &gen, int nBoxes, bool last)
{
    Bvh::PrimVector prims;
    std::vector<float> boxes;
    for (int i = 0; i < nBoxes; ++i) {
        Vec3f c(gen.next1D()*20.0f - 10.0f, gen.next1D()*20.0f - 10.0f, gen.next1D()*20.0f - 10.0f);
        Vec3f e(0.2f + gen.next1D()*1.5f, 0.2f + gen.next1D()*1.5f, 0.2f + gen.next1D()*1.5f);
        Box3f b(c - e, c + e);
        for (int k = 0; k < 3; ++k) boxes.push_back(b.min()[k]);
        for (int k = 0; k < 3; ++k) boxes.push_back(b.max()[k]);
        prims.emplace_back(Bvh::Primitive(b, b.center(), uint32(i)));
    }
    Bvh::BinaryBvh bvh(std::move(prims), 2);
    std::printf("{\"boxes\": [");
    for (size_t i = 0; i < boxes.size(); ++i) std::printf("%s%u", i ? ", " : "", bits(boxes[i]));
    std::printf("], \"rays\": [");
    for (int r = 0; r < 64; ++r) {
        Vec3f o(gen.next1D()*30.0f - 15.0f, gen.next1D()*30.0f - 15.0f, gen.next1D()*30.0f - 15.0f);
        Vec3f t(gen.next1D()*16.0f - 8.0f, gen.next1D()*16.0f - 8.0f, gen.next1D()*16.0f - 8.0f);
        Vec3f d = (t - o).normalized();
        if (r % 8 == 0) d = Vec3f(r % 16 == 0 ? 1.0f : 0.0f, r % 16 == 0 ? 0.0f : -1.0f, 0.0f); // axis-parallel: 1/0 lanes
        Ray ray(o, d, 1e-4f);
        std::printf("%s{\"o\": [%u, %u, %u], \"d\": [%u, %u, %u], \"visits\": [", r ? ", " : "", bits(o.x()), bits(o.y()), bits(o.z()), bits(d.x()), bits(d.y()), bits(d.z()));
        bool first = true;
        bvh.trace(ray, [&](Ray &rr, uint32 id, float tMin, const Vec3pf &) {
            std::printf("%s[%u, %u]", first ? "" : ", ", id, bits(tMin));
            first = false;
            if (id % 5 == 0) rr.setFarT(tMin + 0.75f); // a "hit": Instance::intersect overwrites farT unconditionally
        });
        std::printf("]}");
    }
    std::printf("]}%s\n", last ? "" : ",");
}

int main(int argc, char **argv)
{
    if (argc < 2) { std::fprintf(stderr, "usage: ref_kat2 envmap.hdr\n"); return 1; }
    UniformSampler gen(0xC0FFEEu);
    std::printf("{\n\"binary_bvh\": [\n");
    bvhSet(gen, 50, false);
    bvhSet(gen, 300, true);
    std::printf("],\n");

    // public surface only: sample() walks the private _cdf[] table and scales by _binSize
    ReconstructionFilter tent("tent");
    std::printf("\"filter\": {\"width\": %u, \"samples\": [", bits(tent.width()));
    for (int i = 0; i < 256; ++i) {
        Vec2f xi = gen.next2D();
        if (i < 33) xi = Vec2f(i/64.0f, 1.0f - i/64.0f); // bin edges of the half filter
        float pdf;
        Vec2f uv = tent.sample(xi, pdf);
        std::printf("%s[%u, %u, %u, %u, %u]", i ? ", " : "", bits(xi.x()), bits(xi.y()), bits(uv.x()), bits(uv.y()), bits(pdf));
    }
    std::printf("]},\n");

    int w = 0, h = 0, comp = 0;
    float *img = stbi_loadf(argv[1], &w, &h, &comp, 3);
    if (!img) { std::fprintf(stderr, "cannot decode %s\n", argv[1]); return 1; }
    std::vector<float> weights(size_t(w)*h);
    for (int y = 0, idx = 0; y < h; ++y) { // BitmapTexture::makeSamplable(MAP_SPHERICAL), textures/BitmapTexture.cpp:400-431 (restated)
        float rowWeight = std::sin((y*PI)/h);
        for (int x = 0; x < w; ++x, ++idx) {
            const float *c = img + size_t(idx)*3;
            weights[idx] = max(c[0], c[1], c[2])*rowWeight;
        }
    }
    for (int y = 0; y < h; ++y) {
        for (int x = 0; x < w - 1; ++x) weights[x + y*w] = max(weights[x + y*w], weights[x + 1 + y*w]);
        weights[y*w] = weights[w - 1 + y*w] = max(weights[w - 1 + y*w], weights[y*w]);
        for (int x = w - 1; x > 0; --x) weights[x + y*w] = max(weights[x + y*w], weights[x - 1 + y*w]);
    }
    for (int x = 0; x < w; ++x) {
        for (int y = 0; y < h - 1; ++y) weights[x + y*w] = max(weights[x + y*w], weights[x + (y + 1)*w]);
        weights[x] = weights[x + (h - 1)*w] = max(weights[x], weights[x + (h - 1)*w]);
        for (int y = h - 1; y > 0; --y) weights[x + y*w] = max(weights[x + y*w], weights[x + (y - 1)*w]);
    }
    const unsigned long long texDigest = digest(img, size_t(w)*h*3);
    Distribution2D dist(std::move(weights), w, h);
    // public surface only: pdf(row, column) = conditional pdf * marginal pdf for every cell; warp() inverts both CDFs
    std::vector<float> cell(size_t(w)*h);
    for (int y = 0; y < h; ++y)
        for (int x = 0; x < w; ++x) cell[size_t(x) + size_t(y)*w] = dist.pdf(y, x);
    std::printf("\"envmap\": {\"w\": %d, \"h\": %d, \"texels\": \"%016llx\", \"cell_pdf\": \"%016llx\",\n", w, h, texDigest, digest(cell.data(), cell.size()));
    std::printf(" \"warp\": [");
    for (int i = 0; i < 512; ++i) {
        Vec2f uv = gen.next2D();
        Vec2f in = uv;
        int row, col;
        dist.warp(uv, row, col);
        std::printf("%s[%u, %u, %d, %d, %u, %u, %u]", i ? ", " : "", bits(in.x()), bits(in.y()), row, col, bits(uv.x()), bits(uv.y()), bits(dist.pdf(row, col)));
    }
    std::printf("]");
    std::printf("},\n\"rot_yxz\": [");
    for (int i = 0; i < 8; ++i) {
        Vec3f rot(gen.next1D()*360.0f - 180.0f, gen.next1D()*360.0f - 180.0f, gen.next1D()*360.0f - 180.0f);
        if (i == 0) rot = Vec3f(0.0f, 46.1511f, 180.0f); // the floor quad of materialtest.json
        Mat4f m = Mat4f::rotYXZ(rot);
        std::printf("%s{\"rot\": [%u, %u, %u], \"m\": [", i ? ", " : "", bits(rot.x()), bits(rot.y()), bits(rot.z()));
        for (int r = 0; r < 3; ++r) for (int c = 0; c < 3; ++c) std::printf("%s%u", r + c ? ", " : "", bits(m(r, c)));
        std::printf("]}");
    }
    std::printf("]\n}\n");
    stbi_image_free(img);
    return 0;
}
