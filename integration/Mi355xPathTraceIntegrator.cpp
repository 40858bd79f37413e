#include "Mi355xPathTraceIntegrator.hpp"

#include "cameras/Camera.hpp"
#include "cameras/OutputBuffer.hpp"
#include "primitives/Primitive.hpp"
#include "renderer/TraceableScene.hpp"
#include "sampling/UniformSampler.hpp"
#include "bsdfs/Bsdf.hpp"
#include "media/Medium.hpp"
#include "math/BitManip.hpp"
#include "io/JsonUtils.hpp"
#include "io/JsonObject.hpp"
#include "io/FileUtils.hpp"

#include <host/scene.hpp> // tungsten_amd/host: this repository's scene loader / flattener

#include <rapidjson/document.h>
#include <set>
#include <sstream>

namespace Tungsten {

Mi355xPathTraceIntegrator::Mi355xPathTraceIntegrator() {}
Mi355xPathTraceIntegrator::~Mi355xPathTraceIntegrator()
{
    if (_waiter.joinable()) _waiter.join();
    if (_ctx) tgpu_destroy(_ctx);
}

void Mi355xPathTraceIntegrator::fromJson(JsonPtr value, const Scene &/*scene*/) { _settings.fromJson(value); }
rapidjson::Value Mi355xPathTraceIntegrator::toJson(Allocator &allocator) const { return _settings.toJson(allocator); }

// FlattenedScene: the live objects write themselves out with their own toJson (exactly what Scene::toJson does, io/Scene.cpp:255-279) and
// this repository's loader bakes that text into a tgpu_scene_desc -- meshes, textures and environment maps are re-read from the files the
// objects name, as Scene::load would.  No private member of any primitive is touched.
void Mi355xPathTraceIntegrator::flatten(const TraceableScene &scene)
{
    rapidjson::Document document;
    document.SetObject();
    Allocator &allocator = document.GetAllocator();

    rapidjson::Value media(rapidjson::kArrayType);
    for (const std::shared_ptr<Medium> &m : scene.media())
        media.PushBack(m->toJson(allocator), allocator);

    rapidjson::Value bsdfs(rapidjson::kArrayType), primitives(rapidjson::kArrayType);
    std::set<const Bsdf *> seen;
    for (const std::shared_ptr<Primitive> &p : scene.primitives()) {
        for (int i = 0; i < p->numBsdfs(); ++i) {
            const std::shared_ptr<Bsdf> &b = p->bsdf(i);
            if (b && !b->unnamed() && seen.insert(b.get()).second) // named BSDFs are referenced by name from the primitives
                bsdfs.PushBack(b->toJson(allocator), allocator);
        }
        primitives.PushBack(p->toJson(allocator), allocator);
    }

    document.AddMember("media", std::move(media), allocator);
    document.AddMember("bsdfs", std::move(bsdfs), allocator);
    document.AddMember("primitives", std::move(primitives), allocator);
    document.AddMember("camera", scene.cam().toJson(allocator), allocator);
    document.AddMember("integrator", toJson(allocator), allocator);
    document.AddMember("renderer", scene.rendererSettings().toJson(allocator), allocator);

    const std::string text = JsonUtils::jsonToString(document);
    // resources are resolved in the current directory, as for the running renderer (Scene::load changes into the scene's directory)
    _flat.reset(new tungsten_amd::Scene(tungsten_amd::Scene::loadText(text, FileUtils::getCurrentDir().absolute().asString())));
    _flat->flatten();
}

void Mi355xPathTraceIntegrator::prepareForRender(TraceableScene &scene, uint32 seed)
{
    _currentSpp = 0;
    _scene = &scene;
    advanceSpp();
    scene.cam().requestColorBuffer();

    if (tgpu_init(0, &_ctx) != TGPU_OK)
        FAIL("No MI355X available");
    flatten(scene);

    // diceTiles (PathTraceIntegrator.cpp:27-42): 16x16 tiles in row-major order, one seed per tile from the integrator's sampler
    const uint32 w = scene.cam().resolution().x(), h = scene.cam().resolution().y();
    UniformSampler sampler(MathUtil::hash32(seed));
    _pixelIndex.clear();
    _tileSeed.clear();
    for (uint32 y = 0; y < h; y += 16) {
        for (uint32 x = 0; x < w; x += 16) {
            const uint32 tileSeed = MathUtil::hash32(sampler.nextI());
            for (uint32 ty = y; ty < min(y + 16u, h); ++ty) {
                for (uint32 tx = x; tx < min(x + 16u, w); ++tx) {
                    _pixelIndex.push_back(tx + ty*w);
                    _tileSeed.push_back(tileSeed);
                }
            }
        }
    }
    tgpu_shard_desc shard{uint32(_pixelIndex.size()), _pixelIndex.data(), _tileSeed.data()};
    if (tgpu_upload_scene(_ctx, &_flat->desc(), &shard, 0) != TGPU_OK)
        FAIL("%s", tgpu_last_error(_ctx));
}

void Mi355xPathTraceIntegrator::teardownAfterRender()
{
    if (_waiter.joinable()) _waiter.join();
    if (_ctx) tgpu_destroy(_ctx);
    _ctx = nullptr;
    _flat.reset();
}

// OutputBuffer has no bulk setter: the only writer besides addSample is deserialize(), which reads _bufferA and then _sampleCount -- the
// arrays tgpu_download returns -- from a stream (cameras/OutputBuffer.hpp:191-200)
void Mi355xPathTraceIntegrator::publish()
{
    const size_t n = _pixelIndex.size();
    std::vector<float> shardRgb(n*3), rgb(n*3);
    std::vector<uint32> shardCount(n), count(n);
    if (tgpu_download(_ctx, shardRgb.data(), shardCount.data()) != TGPU_OK)
        FAIL("%s", tgpu_last_error(_ctx));
    for (size_t i = 0; i < n; ++i) { // shard order (tile by tile) -> image order
        for (int c = 0; c < 3; ++c)
            rgb[size_t(_pixelIndex[i])*3 + c] = shardRgb[i*3 + c];
        count[_pixelIndex[i]] = shardCount[i];
    }
    std::string bytes(reinterpret_cast<const char *>(rgb.data()), rgb.size()*sizeof(float));
    bytes.append(reinterpret_cast<const char *>(count.data()), count.size()*sizeof(uint32));
    InputStreamHandle in(new std::istringstream(bytes, std::ios_base::in | std::ios_base::binary));
    _scene->cam().colorBuffer()->deserialize(in);
}

void Mi355xPathTraceIntegrator::startRender(std::function<void()> completionCallback)
{
    if (done()) {
        _currentSpp = _nextSpp;
        advanceSpp();
        completionCallback();
        return;
    }
    if (tgpu_render_pass(_ctx, _currentSpp, _nextSpp - _currentSpp, nullptr) != TGPU_OK)
        FAIL("%s", tgpu_last_error(_ctx));
    if (_waiter.joinable()) _waiter.join();
    _waiter = std::thread([this, completionCallback]() {
        try {
            if (tgpu_wait(_ctx) != TGPU_OK)
                throw std::runtime_error(tgpu_last_error(_ctx));
            publish();
            _currentSpp = _nextSpp;
            advanceSpp();
            completionCallback();
        } catch (...) {
            _error = std::current_exception();
        }
    });
}

void Mi355xPathTraceIntegrator::waitForCompletion()
{
    if (_waiter.joinable()) _waiter.join();
    if (_error) {
        std::exception_ptr e = _error;
        _error = nullptr;
        std::rethrow_exception(e);
    }
}

void Mi355xPathTraceIntegrator::abortRender()
{
    if (_ctx) tgpu_abort(_ctx);
    if (_waiter.joinable()) _waiter.join();
}

void Mi355xPathTraceIntegrator::saveState(OutputStreamHandle &/*out*/) {}
void Mi355xPathTraceIntegrator::loadState(InputStreamHandle &/*in*/) {}

}
