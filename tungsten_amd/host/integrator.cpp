#include "integrator.hpp"

#include <algorithm>
#include <cstring>
#include <stdexcept>

namespace tungsten_amd {

static void check(tgpu_ctx *ctx, int rc, const char *what)
{
    if (rc != TGPU_OK)
        throw std::runtime_error(std::string(what) + " failed (" + std::to_string(rc) + "): " + tgpu_last_error(ctx));
}

void Integrator::writeBuffers(const std::string &suffix, bool /*overwrite*/)
{
    std::vector<float> hdr;
    frameBuffer(hdr);
    const RendererSettings &r = _scene->renderer;
    auto withSuffix = [&](const std::string &file) {
        size_t dot = file.find_last_of('.');
        return dot == std::string::npos ? file + suffix : file.substr(0, dot) + suffix + file.substr(dot);
    };
    int w = int(_scene->resX), h = int(_scene->resY);
    if (!r.outputFile.empty()) {
        std::vector<uint8_t> ldr;
        tonemapToLdr(_scene->tonemap, hdr.data(), w, h, ldr);
        if (!savePng(withSuffix(r.outputFile), ldr.data(), w, h)) throw std::runtime_error("Unable to write " + r.outputFile);
    }
    if (!r.hdrOutputFile.empty()) {
        if (r.hdrOutputFile.size() < 4 || r.hdrOutputFile.substr(r.hdrOutputFile.size() - 4) != ".pfm")
            throw std::runtime_error("hdr_output_file: only .pfm is supported (no OpenEXR here, as in the reference build)");
        if (!savePfm(withSuffix(r.hdrOutputFile), hdr.data(), w, h)) throw std::runtime_error("Unable to write " + r.hdrOutputFile);
    }
}

void Integrator::saveOutputs() { writeBuffers("", _scene->renderer.overwriteOutputFiles); }
void Integrator::saveCheckpoint() { writeBuffers("_checkpoint", true); }

PathTraceIntegrator::PathTraceIntegrator(std::vector<int> devices, uint32_t rank, uint32_t world) : _rank(rank), _world(world)
{
    for (int d : devices) _devices.push_back(Device{d, nullptr, Shard{}});
}

PathTraceIntegrator::~PathTraceIntegrator() { teardownAfterRender(); }

void PathTraceIntegrator::prepareForRender(Scene &scene, uint32_t seed)
{
    teardownAfterRender();
    _scene = &scene;
    _currentSpp = 0;
    _spp = scene.renderer.spp;
    _sppStep = std::max<uint32_t>(scene.renderer.sppStep, 1);
    advanceSpp();
    _work.init(scene.resX, scene.resY, seed);
    if (scene.renderer.useAdaptiveSampling && _world > 1 && !recordExchange)
        throw std::runtime_error("adaptive_sampling across processes needs PathTraceIntegrator::recordExchange (all-gather of the sample records)");
    const uint32_t nLocal = uint32_t(_devices.size());
    for (uint32_t i = 0; i < nLocal; ++i) {
        Device &d = _devices[i];
        d.shard = makeShard(scene.resX, scene.resY, seed, _rank*nLocal + i, _world*nLocal);
        int rc = tgpu_init(d.id, &d.ctx);
        if (rc != TGPU_OK) throw std::runtime_error("tgpu_init(" + std::to_string(d.id) + ") failed: no usable MI355X (there is no CPU fallback)");
        tgpu_shard_desc sd = d.shard.desc();
        check(d.ctx, tgpu_upload_scene(d.ctx, &scene.desc(), &sd, 0), "tgpu_upload_scene");
    }
}

void PathTraceIntegrator::teardownAfterRender()
{
    if (_waiter.joinable()) _waiter.join();
    for (Device &d : _devices) {
        if (d.ctx) tgpu_destroy(d.ctx);
        d.ctx = nullptr;
    }
}

void PathTraceIntegrator::startRender(std::function<void()> completionCallback)
{
    if (_waiter.joinable()) _waiter.join();
    if (done()) { // PathTraceIntegrator.cpp:222-227
        _currentSpp = _nextSpp;
        advanceSpp();
        completionCallback();
        return;
    }
    _aborted = false;
    _error.clear();
    const uint32_t begin = _currentSpp, count = _nextSpp - _currentSpp;
    if (_scene->renderer.useAdaptiveSampling) {
        if (AdaptiveScheduler::needsStatistics(true, _currentSpp)) collectRecords();
        if (!_work.generateWork(_currentSpp, _nextSpp, true)) { // :229-232 -- converged: nothing to do in this pass
            _currentSpp = _nextSpp;
            advanceSpp();
            completionCallback();
            return;
        }
        std::vector<uint32_t> index, next;
        _work.passArrays(index, next);
        for (Device &d : _devices) check(d.ctx, tgpu_render_pass_adaptive(d.ctx, index.data(), next.data(), nullptr), "tgpu_render_pass_adaptive");
    } else {
        for (Device &d : _devices) check(d.ctx, tgpu_render_pass(d.ctx, begin, count, nullptr), "tgpu_render_pass");
    }
    // the completion callback fires from a host thread once every device has finished the pass
    _waiter = std::thread([this, completionCallback]() {
        for (Device &d : _devices) {
            int rc = tgpu_wait(d.ctx);
            if (rc == TGPU_ERR_ABORTED) _aborted = true;
            else if (rc != TGPU_OK && _error.empty()) _error = tgpu_last_error(d.ctx);
        }
        if (!_aborted && _error.empty()) {
            _currentSpp = _nextSpp;
            advanceSpp();
            completionCallback();
        }
    });
}

// Bring the SampleRecords the devices accumulated (k_accumulate_records) back to the host table.
void PathTraceIntegrator::collectRecords()
{
    std::vector<tgpu_sample_record> table(_work.samples().size(), tgpu_sample_record{0, 0.0f, 0.0f});
    for (Device &d : _devices) check(d.ctx, tgpu_download_records(d.ctx, table.data()), "tgpu_download_records");
    if (_world > 1) recordExchange(table);
    _work.setStatistics(table.data());
}

void PathTraceIntegrator::waitForCompletion()
{
    if (_waiter.joinable()) _waiter.join();
    if (!_error.empty()) { // TaskGroup::wait rethrows (TaskGroup.hpp:68-75)
        std::string e = _error;
        _error.clear();
        throw std::runtime_error(e);
    }
}

void PathTraceIntegrator::abortRender()
{
    for (Device &d : _devices)
        if (d.ctx) tgpu_abort(d.ctx);
    if (_waiter.joinable()) _waiter.join();
}

void PathTraceIntegrator::frameBuffer(std::vector<float> &rgb)
{
    rgb.assign(size_t(_scene->resX)*_scene->resY*3, 0.0f);
    std::vector<float> local;
    for (Device &d : _devices) {
        local.resize(d.shard.pixelIndex.size()*3);
        check(d.ctx, tgpu_download(d.ctx, local.data(), nullptr), "tgpu_download");
        for (size_t i = 0; i < d.shard.pixelIndex.size(); ++i)
            std::memcpy(&rgb[size_t(d.shard.pixelIndex[i])*3], &local[i*3], 3*sizeof(float));
    }
}

std::shared_ptr<Integrator> instantiateIntegrator(const std::string &type)
{
    if (type == "path_tracer") return std::make_shared<PathTraceIntegrator>();
    throw JsonLoadException("Unknown integrator '" + type + "': this core provides 'path_tracer' only");
}

} // namespace tungsten_amd
