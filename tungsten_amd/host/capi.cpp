// C entry points of the host library (libtungsten_host.so) used by the Python
// test/bench harness through ctypes.  Pure host code: scene loading and
// flattening, shard construction, image I/O.  No GPU and no oracle here.
#include "adaptive.hpp"
#include "scene.hpp"

#include <cstring>
#include <memory>

using namespace tungsten_amd;

namespace {
void setErr(char *err, int errLen, const std::string &msg)
{
    if (err && errLen > 0) {
        std::strncpy(err, msg.c_str(), size_t(errLen) - 1);
        err[errLen - 1] = 0;
    }
}
} // namespace

extern "C" {

struct tgh_scene { Scene scene; };
struct tgh_shard { Shard shard; tgpu_shard_desc desc; };

tgh_scene *tgh_scene_load(const char *jsonPath, char *err, int errLen)
{
    try {
        std::unique_ptr<tgh_scene> s(new tgh_scene{Scene::load(jsonPath)});
        return s.release();
    } catch (const std::exception &e) {
        setErr(err, errLen, e.what());
        return nullptr;
    }
}

void tgh_scene_free(tgh_scene *s) { delete s; }

void tgh_scene_set_resolution(tgh_scene *s, uint32_t w, uint32_t h) { s->scene.setResolution(w, h); }
void tgh_scene_set_bounces(tgh_scene *s, int minBounces, int maxBounces)
{
    s->scene.settings.min_bounces = minBounces;
    s->scene.settings.max_bounces = maxBounces;
}
void tgh_scene_set_light_sampling(tgh_scene *s, int enable) { s->scene.settings.enable_light_sampling = enable; }
void tgh_scene_set_sobol(tgh_scene *s, int enable) { s->scene.renderer.useSobol = enable != 0; }
// one more entry of renderer "output_buffers" (before tgh_scene_flatten); type: tgpu_output_type
void tgh_scene_add_output_buffer(tgh_scene *s, int type, int twoBufferVariance, int sampleVariance)
{
    OutputBufferSettings b;
    b.type = type;
    b.twoBufferVariance = twoBufferVariance != 0;
    b.sampleVariance = sampleVariance != 0;
    s->scene.renderer.outputs.push_back(b);
}

int tgh_scene_flatten(tgh_scene *s, char *err, int errLen)
{
    try {
        s->scene.flatten();
        return 0;
    } catch (const std::exception &e) {
        setErr(err, errLen, e.what());
        return -1;
    }
}

const tgpu_scene_desc *tgh_scene_desc(const tgh_scene *s) { return &s->scene.desc(); }

// out[0..7] = resX, resY, spp, sppStep, useSobol, adaptive, minBounces, maxBounces
void tgh_scene_info(const tgh_scene *s, uint32_t *out)
{
    out[0] = s->scene.resX;
    out[1] = s->scene.resY;
    out[2] = s->scene.renderer.spp;
    out[3] = s->scene.renderer.sppStep;
    out[4] = s->scene.renderer.useSobol;
    out[5] = s->scene.renderer.useAdaptiveSampling;
    out[6] = uint32_t(s->scene.settings.min_bounces);
    out[7] = uint32_t(s->scene.settings.max_bounces);
}

const char *tgh_scene_tonemap(const tgh_scene *s) { return s->scene.tonemap.c_str(); }

tgh_shard *tgh_shard_create(uint32_t resX, uint32_t resY, uint32_t seed, uint32_t rank, uint32_t world)
{
    tgh_shard *s = new tgh_shard{makeShard(resX, resY, seed, rank, world), {}};
    s->desc = s->shard.desc();
    return s;
}
void tgh_shard_free(tgh_shard *s) { delete s; }
const tgpu_shard_desc *tgh_shard_desc(const tgh_shard *s) { return &s->desc; }
uint32_t tgh_shard_size(const tgh_shard *s) { return s->desc.n_pixels; }
const uint32_t *tgh_shard_pixels(const tgh_shard *s) { return s->desc.pixel_index; }

// AdaptiveScheduler (adaptive.hpp): generateWork of PathTraceIntegrator for the Python harness
struct tgh_adaptive { AdaptiveScheduler work; };
tgh_adaptive *tgh_adaptive_create(uint32_t resX, uint32_t resY, uint32_t seed)
{
    tgh_adaptive *a = new tgh_adaptive;
    a->work.init(resX, resY, seed);
    return a;
}
void tgh_adaptive_free(tgh_adaptive *a) { delete a; }
uint32_t tgh_adaptive_tiles(const tgh_adaptive *a) { return uint32_t(a->work.tileCount()); }
int tgh_adaptive_needs_statistics(int adaptive, uint32_t currentSpp) { return AdaptiveScheduler::needsStatistics(adaptive != 0, currentSpp); }
void tgh_adaptive_set_statistics(tgh_adaptive *a, const tgpu_sample_record *table) { a->work.setStatistics(table); }
// returns 1 and fills sampleIndex/sampleCount (one entry per variance tile) when the pass has work, else 0
int tgh_adaptive_generate(tgh_adaptive *a, uint32_t currentSpp, uint32_t nextSpp, int adaptive, uint32_t *sampleIndex, uint32_t *sampleCount)
{
    if (!a->work.generateWork(currentSpp, nextSpp, adaptive != 0)) return 0;
    std::vector<uint32_t> idx, cnt;
    a->work.passArrays(idx, cnt);
    std::memcpy(sampleIndex, idx.data(), idx.size()*sizeof(uint32_t));
    std::memcpy(sampleCount, cnt.data(), cnt.size()*sizeof(uint32_t));
    return 1;
}

// Mat4f::rotYXZ (math/Mat4f.cpp:118-130) as the transform parser uses it; out9 = row-major upper 3x3 (known-answer test)
void tgh_rot_yxz(const float *rotDeg3, float *out9)
{
    Mat4 m = Mat4::rotYXZ(Vec3(rotDeg3[0], rotDeg3[1], rotDeg3[2]));
    for (int r = 0; r < 3; ++r)
        for (int c = 0; c < 3; ++c) out9[r*3 + c] = m(r, c);
}

int tgh_save_pfm(const char *path, const float *rgb, int w, int h) { return savePfm(path, rgb, w, h) ? 0 : -1; }
int tgh_save_png_tonemapped(const char *path, const char *tonemap, const float *rgb, int w, int h)
{
    std::vector<uint8_t> ldr;
    tonemapToLdr(tonemap, rgb, w, h, ldr);
    return savePng(path, ldr.data(), w, h) ? 0 : -1;
}

// the loader's LDR decoders on their own (tests/test_ref_kat6.py): RGBA8, top row first.  Returns 0, or -1 with the reason in err.
int tgh_decode_ldr(const char *path, int *w, int *h, uint8_t *rgba, uint64_t capacity, char *err, int errLen)
{
    std::vector<uint8_t> px;
    std::string why, p(path);
    auto ends = [&](const char *suffix) { const size_t n = std::strlen(suffix); return p.size() >= n && p.compare(p.size() - n, n, suffix) == 0; };
    bool ok;
    if (ends(".png")) { ok = loadPng(p, *w, *h, px); why = "is not a PNG this loader reads"; }
    else if (ends(".bmp")) ok = loadBmp(p, *w, *h, px, why);
    else if (ends(".tga")) ok = loadTga(p, *w, *h, px, why);
    else ok = loadJpeg(p, *w, *h, px, why);
    if (!ok || px.size() > capacity) {
        std::snprintf(err, size_t(errLen), "%s", ok ? "buffer too small" : why.c_str());
        return -1;
    }
    std::memcpy(rgba, px.data(), px.size());
    return 0;
}

} // extern "C"
