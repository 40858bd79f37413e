// Fixture generator, NOT built by any Makefile and not needed by any test: records what the COMPILED Embree 2.11 of the survey-phase reference build
// (static libraries left in /tmp/tungsten-build by the SURVEY.md section 8c recipe) does with (a) a set of boxes registered as ONE user geometry, exactly as
// TraceableScene registers the scene's finites (renderer/TraceableScene.hpp:107-129), and (b) a triangle mesh registered as TriangleMesh does
// (primitives/TriangleMesh.cpp:531-566).  tests/golden/make_ref_kat5.py writes the inputs, runs this program and stores inputs + outputs as
// tests/golden/ref_kat5.npz; tests/test_ref_kat5.py replays them through oracle/embree_bvh4.h.
//
//   user mode:  for every ray the sequence of primitives Embree asks (the intersect callback logs its item), the callback reporting a hit at a scripted
//               distance (one float per ray and primitive, negative = miss) under Quad::intersect's interval rule t < nearT || t > farT -> miss.
//   tris mode:  (primID, t, u, v) of rtcIntersect for every ray.
//
// Build (flags = the reference build's own, build.ninja):
//   B=/tmp/tungsten-build; S=/root/reference/src
//   g++ -std=c++0x -march=core2 -msse2 -msse3 -mssse3 -mno-sse4.1 -mno-sse4.2 -mno-avx -mno-fma -mno-avx2 -O3 -DNDEBUG -DEMBREE_STATIC_LIB=1 \
//       -I$S/thirdparty/embree/include -o ref_embree_probe tests/tools/ref_embree_probe.cpp \
//       $B/src/thirdparty/embree/lib{embree,embree_sse42,simd,lexers,sys}.a -ldl -lpthread
// Run:  ref_embree_probe user in.bin out.bin   |   ref_embree_probe tris in.bin out.bin      (little-endian binary, layouts below)
#include <embree2/rtcore.h>
#include <embree2/rtcore_ray.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

static std::vector<uint8_t> slurp(const char *path)
{
    FILE *f = std::fopen(path, "rb");
    if (!f) { std::perror(path); std::exit(1); }
    std::fseek(f, 0, SEEK_END);
    long n = std::ftell(f);
    std::fseek(f, 0, SEEK_SET);
    std::vector<uint8_t> b; b.resize(size_t(n));
    if (std::fread(b.data(), 1, size_t(n), f) != size_t(n)) std::exit(1);
    std::fclose(f);
    return b;
}

struct UserCtx {
    const float *boxes;          // n*6: lo xyz, hi xyz
    const float *scripted;       // per ray: n distances
    size_t n;
    std::vector<uint32_t> visits;
};
static UserCtx *g_user;

int main(int argc, char **argv)
{
    if (argc < 4) { std::printf("usage: ref_embree_probe user|tris in.bin out.bin\n"); return 1; }
    std::vector<uint8_t> in = slurp(argv[2]);
    FILE *out = std::fopen(argv[3], "wb");
    RTCDevice dev = rtcNewDevice(nullptr);
    RTCScene scene = rtcDeviceNewScene(dev, RTC_SCENE_STATIC | RTC_SCENE_INCOHERENT, RTC_INTERSECT1);
    const uint8_t *p = in.data();
    if (!std::strcmp(argv[1], "user")) {
        // in : u32 nBoxes, u32 nRays, f32 boxes[nBoxes*6], per ray { f32 org[3], dir[3], tnear, tfar, scripted[nBoxes] }
        // out: per ray { u32 nVisits, u32 visits[nVisits], u32 primID (0xFFFFFFFF = none), f32 tfar }
        uint32_t nBoxes, nRays;
        std::memcpy(&nBoxes, p, 4); std::memcpy(&nRays, p + 4, 4);
        const float *boxes = reinterpret_cast<const float *>(p + 8);
        const float *rays = boxes + size_t(nBoxes)*6;
        UserCtx ctx{boxes, nullptr, nBoxes, {}};
        g_user = &ctx;
        unsigned geom = rtcNewUserGeometry(scene, nBoxes);
        rtcSetUserData(scene, geom, &ctx);
        rtcSetBoundsFunction(scene, geom, [](void *ptr, size_t i, RTCBounds &b) {
            const float *bx = static_cast<UserCtx *>(ptr)->boxes + i*6;
            b.lower_x = bx[0]; b.lower_y = bx[1]; b.lower_z = bx[2]; b.align0 = 0.0f;
            b.upper_x = bx[3]; b.upper_y = bx[4]; b.upper_z = bx[5]; b.align1 = 0.0f;
        });
        rtcSetIntersectFunction(scene, geom, [](void *ptr, RTCRay &ray, size_t i) {
            UserCtx *c = static_cast<UserCtx *>(ptr);
            c->visits.push_back(uint32_t(i));
            float t = c->scripted[i];
            if (t < 0.0f || t < ray.tnear || t > ray.tfar) return;
            ray.tfar = t;
            ray.geomID = 0;
            ray.primID = unsigned(i);
        });
        rtcCommit(scene);
        const size_t stride = 8 + nBoxes;
        for (uint32_t r = 0; r < nRays; ++r) {
            const float *rr = rays + size_t(r)*stride;
            RTCRay ray;
            for (int k = 0; k < 3; ++k) { ray.org[k] = rr[k]; ray.dir[k] = rr[3 + k]; }
            ray.tnear = rr[6]; ray.tfar = rr[7];
            ray.geomID = RTC_INVALID_GEOMETRY_ID; ray.primID = RTC_INVALID_GEOMETRY_ID; ray.instID = RTC_INVALID_GEOMETRY_ID;
            ray.mask = 0xFFFFFFFF; ray.time = 0.0f;
            ctx.scripted = rr + 8;
            ctx.visits.clear();
            rtcIntersect(scene, ray);
            uint32_t nv = uint32_t(ctx.visits.size());
            std::fwrite(&nv, 4, 1, out);
            std::fwrite(ctx.visits.data(), 4, nv, out);
            uint32_t prim = ray.geomID == RTC_INVALID_GEOMETRY_ID ? 0xFFFFFFFFu : ray.primID;
            std::fwrite(&prim, 4, 1, out);
            std::fwrite(&ray.tfar, 4, 1, out);
        }
    } else {
        // in : u32 nVerts, u32 nTris, u32 nRays, f32 verts[nVerts*3], u32 tris[nTris*3], per ray { f32 org[3], dir[3], tnear, tfar }
        // out: per ray { u32 primID (0xFFFFFFFF = none), f32 t, u, v }
        uint32_t nVerts, nTris, nRays;
        std::memcpy(&nVerts, p, 4); std::memcpy(&nTris, p + 4, 4); std::memcpy(&nRays, p + 8, 4);
        const float *verts = reinterpret_cast<const float *>(p + 12);
        const uint32_t *tris = reinterpret_cast<const uint32_t *>(verts + size_t(nVerts)*3);
        const float *rays = reinterpret_cast<const float *>(tris + size_t(nTris)*3);
        unsigned geom = rtcNewTriangleMesh(scene, RTC_GEOMETRY_STATIC, nTris, nVerts, 1);
        float *vs = static_cast<float *>(rtcMapBuffer(scene, geom, RTC_VERTEX_BUFFER));
        uint32_t *ts = static_cast<uint32_t *>(rtcMapBuffer(scene, geom, RTC_INDEX_BUFFER));
        for (uint32_t i = 0; i < nVerts; ++i) { vs[i*4] = verts[i*3]; vs[i*4 + 1] = verts[i*3 + 1]; vs[i*4 + 2] = verts[i*3 + 2]; vs[i*4 + 3] = 0.0f; }
        std::memcpy(ts, tris, size_t(nTris)*12);
        rtcUnmapBuffer(scene, geom, RTC_VERTEX_BUFFER);
        rtcUnmapBuffer(scene, geom, RTC_INDEX_BUFFER);
        rtcCommit(scene);
        for (uint32_t r = 0; r < nRays; ++r) {
            const float *rr = rays + size_t(r)*8;
            RTCRay ray;
            for (int k = 0; k < 3; ++k) { ray.org[k] = rr[k]; ray.dir[k] = rr[3 + k]; }
            ray.tnear = rr[6]; ray.tfar = rr[7];
            ray.geomID = RTC_INVALID_GEOMETRY_ID; ray.primID = RTC_INVALID_GEOMETRY_ID; ray.instID = RTC_INVALID_GEOMETRY_ID;
            ray.mask = 0xFFFFFFFF; ray.time = 0.0f;
            rtcIntersect(scene, ray);
            uint32_t prim = ray.geomID == RTC_INVALID_GEOMETRY_ID ? 0xFFFFFFFFu : ray.primID;
            float o[3] = {ray.tfar, ray.u, ray.v};
            std::fwrite(&prim, 4, 1, out);
            std::fwrite(o, 4, 3, out);
        }
    }
    std::fclose(out);
    rtcDeleteScene(scene);
    rtcDeleteDevice(dev);
    return 0;
}
