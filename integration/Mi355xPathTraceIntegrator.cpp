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
#include <functional>
#include <set>
#include <sstream>

namespace Tungsten {

Mi355xPathTraceIntegrator::Mi355xPathTraceIntegrator() {}
Mi355xPathTraceIntegrator::~Mi355xPathTraceIntegrator() { teardownAfterRender(); }

void Mi355xPathTraceIntegrator::check(int status) const
{
    if (status != TGPU_OK)
        FAIL("%s", _ctx ? tgpu_last_error(_ctx) : "no device context");
}

bool Mi355xPathTraceIntegrator::adaptive() const { return _scene->rendererSettings().useAdaptiveSampling(); }
bool Mi355xPathTraceIntegrator::supportsResumeRender() const { return true; } // PathTraceIntegrator.cpp:215-218

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
    teardownAfterRender(); // a second prepareForRender without a teardown in between must not leak the first context
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
    _seedOfTile.clear();
    for (uint32 y = 0; y < h; y += 16) {
        for (uint32 x = 0; x < w; x += 16) {
            const uint32 tileSeed = MathUtil::hash32(sampler.nextI());
            _seedOfTile.push_back(tileSeed);
            for (uint32 ty = y; ty < min(y + 16u, h); ++ty) {
                for (uint32 tx = x; tx < min(x + 16u, w); ++tx) {
                    _pixelIndex.push_back(tx + ty*w);
                    _tileSeed.push_back(tileSeed);
                }
            }
        }
    }
    _work.init(w, h, seed); // the same generator, advanced past the tile seeds: the draws generateWork's stochastic rounding takes
    tgpu_shard_desc shard{uint32(_pixelIndex.size()), _pixelIndex.data(), _tileSeed.data()};
    check(tgpu_upload_scene(_ctx, &_flat->desc(), &shard, 0));
}

void Mi355xPathTraceIntegrator::teardownAfterRender()
{
    if (_waiter.joinable()) _waiter.join();
    if (_ctx) tgpu_destroy(_ctx);
    _ctx = nullptr;
    _flat.reset();
    _error = nullptr;
    _aborted = false;
}

// One buffer of the camera, device -> OutputBuffer.  OutputBuffer has no bulk setter: the only writer besides addSample is deserialize(), which
// reads _bufferA, then _bufferB and _variance if the buffer has them, then _sampleCount from a stream (cameras/OutputBuffer.hpp:191-200).  Which
// parts exist is what the renderer settings asked for; the flattened scene carries the same flags, so the device holds exactly these parts.
void Mi355xPathTraceIntegrator::publishBuffer(int type, int channels, const std::function<void(InputStreamHandle &)> &deserialize)
{
    const tgpu_output_buffer &want = _flat->desc().output_buffers[type];
    const size_t n = _pixelIndex.size();
    std::string bytes;
    std::vector<uint32> shardWords, imageWords;
    for (int part : {TGPU_PART_A, TGPU_PART_B, TGPU_PART_VARIANCE, TGPU_PART_COUNT}) {
        if ((part == TGPU_PART_B && !want.two_buffer_variance) || (part == TGPU_PART_VARIANCE && !want.sample_variance))
            continue;
        const size_t width = part == TGPU_PART_COUNT ? 1 : size_t(channels);
        shardWords.resize(n*width);
        imageWords.resize(n*width);
        check(tgpu_download_output_buffer(_ctx, type, part, shardWords.data()));
        for (size_t i = 0; i < n; ++i) // shard order (tile by tile) -> image order
            for (size_t c = 0; c < width; ++c)
                imageWords[size_t(_pixelIndex[i])*width + c] = shardWords[i*width + c];
        bytes.append(reinterpret_cast<const char *>(imageWords.data()), imageWords.size()*sizeof(uint32));
    }
    InputStreamHandle in(new std::istringstream(bytes, std::ios_base::in | std::ios_base::binary));
    deserialize(in);
}

// The way back (resume): OutputBuffer::serialize writes the same parts in the same order (:202-211)
void Mi355xPathTraceIntegrator::uploadBuffer(int type, int channels, const std::function<void(OutputStreamHandle &)> &serialize)
{
    const tgpu_output_buffer &want = _flat->desc().output_buffers[type];
    const size_t n = _pixelIndex.size();
    std::shared_ptr<std::ostringstream> text(new std::ostringstream(std::ios_base::out | std::ios_base::binary));
    OutputStreamHandle out(text);
    serialize(out);
    const std::string bytes = text->str();
    size_t offset = 0;
    std::vector<uint32> shardWords;
    for (int part : {TGPU_PART_A, TGPU_PART_B, TGPU_PART_VARIANCE, TGPU_PART_COUNT}) {
        if ((part == TGPU_PART_B && !want.two_buffer_variance) || (part == TGPU_PART_VARIANCE && !want.sample_variance))
            continue;
        const size_t width = part == TGPU_PART_COUNT ? 1 : size_t(channels);
        if (offset + n*width*sizeof(uint32) > bytes.size())
            FAIL("Output buffer %d serialises to fewer parts than the renderer settings announce", type);
        const uint32 *imageWords = reinterpret_cast<const uint32 *>(bytes.data() + offset);
        shardWords.resize(n*width);
        for (size_t i = 0; i < n; ++i)
            for (size_t c = 0; c < width; ++c)
                shardWords[i*width + c] = imageWords[size_t(_pixelIndex[i])*width + c];
        check(tgpu_upload_output_buffer(_ctx, type, part, shardWords.data()));
        offset += n*width*sizeof(uint32);
    }
}

void Mi355xPathTraceIntegrator::publish()
{
    Camera &cam = _scene->cam();
    publishBuffer(TGPU_OUTPUT_COLOR, 3, [&](InputStreamHandle &in) { cam.colorBuffer()->deserialize(in); });
    if (cam.depthBuffer())      publishBuffer(TGPU_OUTPUT_DEPTH,      1, [&](InputStreamHandle &in) { cam.depthBuffer()->deserialize(in); });
    if (cam.normalBuffer())     publishBuffer(TGPU_OUTPUT_NORMAL,     3, [&](InputStreamHandle &in) { cam.normalBuffer()->deserialize(in); });
    if (cam.albedoBuffer())     publishBuffer(TGPU_OUTPUT_ALBEDO,     3, [&](InputStreamHandle &in) { cam.albedoBuffer()->deserialize(in); });
    if (cam.visibilityBuffer()) publishBuffer(TGPU_OUTPUT_VISIBILITY, 1, [&](InputStreamHandle &in) { cam.visibilityBuffer()->deserialize(in); });
}

// generateWork (PathTraceIntegrator.cpp:110-134): from 16 spp on the per-tile statistics the device accumulated steer the next pass
bool Mi355xPathTraceIntegrator::generateWork(std::vector<uint32> &sampleIndex, std::vector<uint32> &sampleCount)
{
    if (tungsten_amd::AdaptiveScheduler::needsStatistics(adaptive(), _currentSpp)) {
        std::vector<tgpu_sample_record> table(_work.tileCount());
        check(tgpu_download_records(_ctx, table.data()));
        _work.setStatistics(table.data());
    }
    if (!_work.generateWork(_currentSpp, _nextSpp, adaptive()))
        return false;
    _work.passArrays(sampleIndex, sampleCount);
    return true;
}

void Mi355xPathTraceIntegrator::startRender(std::function<void()> completionCallback)
{
    std::vector<uint32> sampleIndex, sampleCount;
    if (done() || !generateWork(sampleIndex, sampleCount)) { // :222-227
        _currentSpp = _nextSpp;
        advanceSpp();
        completionCallback();
        return;
    }
    _aborted = false;
    if (adaptive())
        check(tgpu_render_pass_adaptive(_ctx, sampleIndex.data(), sampleCount.data(), nullptr));
    else
        check(tgpu_render_pass(_ctx, _currentSpp, _nextSpp - _currentSpp, nullptr));
    if (_waiter.joinable()) _waiter.join();
    _waiter = std::thread([this, completionCallback]() {
        try {
            const int status = tgpu_wait(_ctx);
            if (_aborted) return; // abortRender: the pass is dropped, nothing is published, nobody is called back (TaskGroup::abort)
            if (status != TGPU_OK)
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

// PathTraceIntegrator::abortRender (:249-256) stops the task group and returns; it never throws, and neither does the next waitForCompletion
void Mi355xPathTraceIntegrator::abortRender()
{
    _aborted = true;
    if (_ctx) tgpu_abort(_ctx);
    if (_waiter.joinable()) _waiter.join();
    _error = nullptr;
}

// The resume file's integrator block in the reference's layout (:158-164): one 24-byte SampleRecord per variance tile
// (SampleRecord.hpp:57-66: sampleCount, nextSampleCount, sampleIndex, adaptiveWeight, mean, runningVariance), then every tile's sampler
// (SobolPathSampler: uint32 seed + the supplemental generator's 64-bit state; UniformPathSampler: the state alone).  The device keeps no
// tile-sequential generator, so the state written is the one a fresh sampler has -- what loadState expects back.
void Mi355xPathTraceIntegrator::saveState(OutputStreamHandle &out)
{
    if (adaptive()) {
        std::vector<tgpu_sample_record> table(_work.tileCount());
        check(tgpu_download_records(_ctx, table.data()));
        _work.setStatistics(table.data());
    }
    for (size_t i = 0; i < _work.tileCount(); ++i) {
        tungsten_amd::TileState t = _work.tile(i);
        if (!adaptive()) t.sampleCount = 0;
        FileUtils::streamWrite(out, t.sampleCount);
        FileUtils::streamWrite(out, t.nextSampleCount);
        FileUtils::streamWrite(out, t.sampleIndex);
        FileUtils::streamWrite(out, t.adaptiveWeight);
        FileUtils::streamWrite(out, t.mean);
        FileUtils::streamWrite(out, t.runningVariance);
    }
    const bool sobol = _scene->rendererSettings().useSobol();
    for (uint32 seed : _seedOfTile) {
        if (sobol) FileUtils::streamWrite(out, seed);
        FileUtils::streamWrite(out, uint64(seed));
    }
}

// Integrator::resumeRender (Integrator.cpp:131-162) has filled the camera's buffers from the file just before this call: they go to the device
// together with the SampleRecords, so that the next pass continues the running means.
void Mi355xPathTraceIntegrator::loadState(InputStreamHandle &in)
{
    std::vector<tgpu_sample_record> table(_work.tileCount());
    for (size_t i = 0; i < _work.tileCount(); ++i) {
        tungsten_amd::TileState t = tungsten_amd::TileState();
        FileUtils::streamRead(in, t.sampleCount);
        FileUtils::streamRead(in, t.nextSampleCount);
        FileUtils::streamRead(in, t.sampleIndex);
        FileUtils::streamRead(in, t.adaptiveWeight);
        FileUtils::streamRead(in, t.mean);
        FileUtils::streamRead(in, t.runningVariance);
        _work.setTile(i, t);
        table[i] = tgpu_sample_record{t.sampleCount, t.mean, t.runningVariance};
    }
    const bool sobol = _scene->rendererSettings().useSobol();
    for (uint32 expected : _seedOfTile) {
        uint32 seed = expected;
        uint64 state;
        if (sobol) FileUtils::streamRead(in, seed);
        FileUtils::streamRead(in, state);
        if (seed != expected)
            FAIL("The resume file was rendered with another seed");
    }
    if (adaptive()) check(tgpu_upload_records(_ctx, table.data()));
    Camera &cam = _scene->cam();
    uploadBuffer(TGPU_OUTPUT_COLOR, 3, [&](OutputStreamHandle &out) { cam.colorBuffer()->serialize(out); });
    if (cam.depthBuffer())      uploadBuffer(TGPU_OUTPUT_DEPTH,      1, [&](OutputStreamHandle &out) { cam.depthBuffer()->serialize(out); });
    if (cam.normalBuffer())     uploadBuffer(TGPU_OUTPUT_NORMAL,     3, [&](OutputStreamHandle &out) { cam.normalBuffer()->serialize(out); });
    if (cam.albedoBuffer())     uploadBuffer(TGPU_OUTPUT_ALBEDO,     3, [&](OutputStreamHandle &out) { cam.albedoBuffer()->serialize(out); });
    if (cam.visibilityBuffer()) uploadBuffer(TGPU_OUTPUT_VISIBILITY, 1, [&](OutputStreamHandle &out) { cam.visibilityBuffer()->serialize(out); });
}

}
