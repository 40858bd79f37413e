// Investigation aid, NOT built by any Makefile and not needed by any test: a driver that links the survey-phase build of the UNMODIFIED
// reference (static libraries left in /tmp/tungsten-build by the SURVEY.md section 8c recipe: cmake + ninja, g++ -O3 -march=core2, no FMA) and
// walks one 16x16 tile exactly as PathTraceIntegrator::renderTile does (integrators/path_tracer/PathTraceIntegrator.cpp:27-42,136-156), tracing
// every other sample through PathTracer::traceSample and ONE chosen sample vertex by vertex through TraceBase's protected entry points, with every
// random draw logged.  Its output lines up with the oracle's ORACLE_TRACE=1 dump (oracle/oracle.cpp, "ORC ..." / "rnd ..." lines).
// This is how the one-tile residual of the Erlang / Davis-Weinstein / interpolated transmittance fixtures was root-caused in round 2: the first
// deviating vertex was a hit on a Cornell box's bottom face where the oracle hit the coplanar floor quad at the same binary32 distance
// (oracle_ctx::coplanarTieGoesToCube).
//
// Build (needs /root/reference and /tmp/tungsten-build; flags = the reference build's own, build.ninja):
//   B=/tmp/tungsten-build; S=/root/reference/src
//   g++ -std=c++0x -march=core2 -msse2 -msse3 -mssse3 -mno-sse4.1 -mno-sse4.2 -mno-avx -mno-fma -mno-avx2 -O3 -DNDEBUG -DCONSTEXPR=constexpr \
//       -DEMBREE_STATIC_LIB=1 -DINSTALL_PREFIX=\"/usr/local\" -DLODEPNG_NO_COMPILE_DISK=1 -DRAPIDJSON_HAS_STDSTRING=1 -DSTBI_NO_STDIO=1 \
//       -I$S/core -I$S/thirdparty -I$S/thirdparty/embree/include -I$S -o ref_path_dump tests/tools/ref_path_dump.cpp \
//       $B/libcore.a $B/libthirdparty.a $B/src/thirdparty/embree/lib{embree,embree_sse42,simd,lexers,sys}.a -ldl -lpthread
// Run:  ref_path_dump /abs/path/scene.json px py pass passes      (scene with the resolution already patched in; spp_step = 1 passes)
#include "integrators/path_tracer/PathTracer.hpp"
#include "sampling/SobolPathSampler.hpp"
#include "sampling/UniformPathSampler.hpp"
#include "sampling/UniformSampler.hpp"
#include "renderer/TraceableScene.hpp"
#include "primitives/EmbreeUtil.hpp"
#include "thread/ThreadUtils.hpp"
#include "io/DirectoryChange.hpp"
#include "io/JsonDocument.hpp"
#include "io/Scene.hpp"
#include "primitives/Cube.hpp"
#include "primitives/Quad.hpp"
#include <cstdio>
#include <cstring>
using namespace Tungsten;
static unsigned hx(float f) { unsigned u; std::memcpy(&u, &f, 4); return u; }
#define HX3(v) hx((v)[0]), hx((v)[1]), hx((v)[2])
struct LogSampler : PathSampleGenerator {
    PathSampleGenerator &in; int n = 0;
    LogSampler(PathSampleGenerator &i) : in(i) {}
    void startPath(uint32 a, uint32 b) override { in.startPath(a, b); }
    void advancePath() override { in.advancePath(); }
    void saveState(OutputStreamHandle &o) override { in.saveState(o); }
    void loadState(InputStreamHandle &i) override { in.loadState(i); }
    bool nextBoolean(float p) override { bool r = in.nextBoolean(p); std::printf("    rnd#%d bool(%08x)=%d\n", n++, hx(p), int(r)); return r; }
    int nextDiscrete(int k) override { int r = in.nextDiscrete(k); std::printf("    rnd#%d disc(%d)=%d\n", n++, k, r); return r; }
    float next1D() override { float r = in.next1D(); std::printf("    rnd#%d 1d=%08x\n", n++, hx(r)); return r; }
    Vec2f next2D() override { Vec2f r = in.next2D(); std::printf("    rnd#%d 2d=%08x %08x\n", n++, hx(r.x()), hx(r.y())); n++; return r; }
    UniformSampler &uniformGenerator() override { std::printf("    uniformGenerator()\n"); return in.uniformGenerator(); }
};
struct Dbg : PathTracer {
    PathTracerSettings _st;
    Dbg(TraceableScene *s, const PathTracerSettings &st) : PathTracer(s, st, 0), _st(st) {}
    Vec3f walk(Vec2u pixel, PathSampleGenerator &sampler)
    {
        PositionSample point; DirectionSample direction;
        _scene->cam().samplePosition(sampler, point);
        _scene->cam().sampleDirection(sampler, point, pixel, direction);
        Vec3f throughput = point.weight*direction.weight, emission(0.0f);
        Ray ray(point.p, direction.d); ray.setPrimaryRay(true);
        MediumSample ms; SurfaceScatterEvent ev; IntersectionTemporary data; Medium::MediumState state; state.reset(); IntersectionInfo info;
        const Medium *medium = _scene->cam().medium().get();
        int bounce = 0, mediumBounces = 0; bool wasSpecular = true;
        bool didHit = _scene->intersect(ray, data, info);
        std::printf("REF start O=%08x %08x %08x D=%08x %08x %08x\n", HX3(ray.pos()), HX3(ray.dir()));
        while ((didHit || medium) && bounce < _st.maxBounces) {
            std::printf("REF v%d hit=%d prim=%s t=%08x tp=%08x %08x %08x em=%08x %08x %08x\n", bounce, int(didHit), !didHit ? "-" : dynamic_cast<const Cube *>(info.primitive) ? "cube" : dynamic_cast<const Quad *>(info.primitive) ? "quad" : "other", hx(ray.farT()), HX3(throughput), HX3(emission));
            bool hitSurface = true;
            if (medium) {
                ms.continuedWeight = throughput;
                if (!medium->sampleDistance(sampler, ray, state, ms)) { std::printf("REF sampleDistance false\n"); return emission; }
                std::printf("  med t=%08x exited=%d w=%08x %08x %08x pdf=%08x\n", hx(ms.t), int(ms.exited), HX3(ms.weight), hx(ms.pdf));
                emission += throughput*ms.emission; throughput *= ms.weight; hitSurface = ms.exited;
                if (hitSurface && !didHit) break;
            }
            if (hitSurface) {
                ev = makeLocalScatterEvent(data, info, ray, &sampler);
                Vec3f tr(-1.0f);
                bool terminate = !handleSurface(ev, data, info, medium, bounce, false, _st.enableLightSampling && (mediumBounces > 0 || _st.includeSurfaces),
                                                ray, throughput, emission, wasSpecular, state, &tr);
                if (terminate) { std::printf("REF surface terminate em=%08x %08x %08x\n", HX3(emission)); return emission; }
            } else {
                mediumBounces++;
                if (!handleVolume(sampler, ms, medium, bounce, false, _st.enableVolumeLightSampling && (mediumBounces > 1 || _st.lowOrderScattering),
                                  ray, throughput, emission, wasSpecular)) { std::printf("REF volume terminate\n"); return emission; }
            }
            if (throughput.max() == 0.0f) break;
            float rp = std::abs(throughput).max();
            if (bounce > 2 && rp < 0.1f) {
                if (sampler.nextBoolean(rp)) throughput /= rp;
                else { std::printf("REF roulette end em=%08x %08x %08x\n", HX3(emission)); return emission; }
            }
            std::printf("REF   -> O=%08x %08x %08x D=%08x %08x %08x tp=%08x %08x %08x em=%08x %08x %08x\n", HX3(ray.pos()), HX3(ray.dir()), HX3(throughput), HX3(emission));
            bounce++;
            if (bounce < _st.maxBounces) didHit = _scene->intersect(ray, data, info);
        }
        if (bounce >= _st.minBounces && bounce < _st.maxBounces)
            handleInfiniteLights(data, info, _st.enableLightSampling, ray, throughput, wasSpecular, emission);
        std::printf("REF end bounce=%d em=%08x %08x %08x\n", bounce, HX3(emission));
        return emission;
    }
};
int main(int argc, char **argv)
{
    if (argc < 6) { std::printf("usage: refdbg scene.json px py pass passes\n"); return 1; }
    int tpx = std::atoi(argv[2]), tpy = std::atoi(argv[3]), tpass = std::atoi(argv[4]), passes = std::atoi(argv[5]);
    EmbreeUtil::initDevice();
    ThreadUtils::startThreads(1);
    Path scenePath(argv[1]);
    Path inputDir = scenePath.parent();
    std::unique_ptr<Scene> scene(Scene::load(scenePath, nullptr, &inputDir));
    scene->loadResources();
    DirectoryChange ctx(inputDir);
    std::unique_ptr<TraceableScene> flat(scene->makeTraceable(0xBA5EBA11));
    PathTracerSettings settings;
    JsonDocument doc(scenePath);
    settings.fromJson(doc["integrator"]);
    Dbg tracer(flat.get(), settings);
    const uint32 w = flat->cam().resolution().x(), h = flat->cam().resolution().y(), T = 16;
    UniformSampler smp(MathUtil::hash32(0xBA5EBA11u));
    for (uint32 y = 0; y < h; y += T) for (uint32 x = 0; x < w; x += T) {
        uint32 seed = MathUtil::hash32(smp.nextI());
        if (!(uint32(tpx) >= x && uint32(tpx) < x + T && uint32(tpy) >= y && uint32(tpy) < y + T)) continue;
        std::unique_ptr<PathSampleGenerator> s(scene->rendererSettings().useSobol() ? (PathSampleGenerator *)new SobolPathSampler(seed) : new UniformPathSampler(seed));
        for (int p = 0; p < passes; ++p)
            for (uint32 yy = y; yy < std::min(y + T, h); ++yy) for (uint32 xx = x; xx < std::min(x + T, w); ++xx) {
                s->startPath(xx + yy*w, p);
                if (int(xx) == tpx && int(yy) == tpy && p == tpass) { LogSampler ls(*s); Vec3f c = tracer.walk(Vec2u(xx, yy), ls); std::printf("REF result %08x %08x %08x\n", HX3(c)); return 0; }
                Vec3f c = tracer.traceSample(Vec2u(xx, yy), *s);
                (void)c;
            }
    }
    return 0;
}
