// Host-side mirror of the reference's integrator plugin surface for this path:
//   Tungsten::Integrator            (src/core/integrators/Integrator.hpp:16-63)
//   Tungsten::PathTraceIntegrator   (src/core/integrators/path_tracer/PathTraceIntegrator.{hpp,cpp})
// Same member names, argument meaning and error behaviour (exceptions on the host side, FAIL ->
// std::runtime_error, Debug.hpp:26), implemented on top of the C ABI in include/tungsten_amd.h.  The
// reference's `TraceableScene&` argument becomes the flattened `Scene` of scene.hpp, which plays the
// role TraceableScene plays there (prepared scene + renderer settings + camera output buffers).
#pragma once
#include "../../include/tungsten_amd.h"
#include "adaptive.hpp"
#include "scene.hpp"

#include <atomic>
#include <functional>
#include <memory>
#include <string>
#include <thread>
#include <vector>

namespace tungsten_amd {

class Integrator {
public:
    virtual ~Integrator() {}

    virtual void prepareForRender(Scene &scene, uint32_t seed) = 0;
    virtual void teardownAfterRender() = 0;
    virtual void startRender(std::function<void()> completionCallback) = 0;
    virtual void waitForCompletion() = 0;
    virtual void abortRender() = 0;
    virtual void saveOutputs();
    virtual void saveCheckpoint();
    virtual bool supportsResumeRender() const { return false; }

    bool done() const { return _currentSpp >= _spp; }       // Integrator.hpp:52-55
    uint32_t currentSpp() const { return _currentSpp; }
    uint32_t nextSpp() const { return _nextSpp; }

    // Camera::colorBuffer() contents in image order (what the reference reads through Camera::getLinear)
    virtual void frameBuffer(std::vector<float> &rgb) = 0;

protected:
    void advanceSpp() { _nextSpp = std::min(_currentSpp + _sppStep, _spp); } // Integrator.cpp:51-54
    void writeBuffers(const std::string &suffix, bool overwrite);             // Integrator.cpp:56-80

    Scene *_scene = nullptr;
    uint32_t _currentSpp = 0, _nextSpp = 0, _spp = 0, _sppStep = 16;
};

// "path_tracer" on one or more MI355X: the image plane's 16x16 tiles are dealt round-robin to the
// devices of this process (one tgpu context per device); within a multi-process launch
// (bench.py / torch.distributed) each process passes its own rank/world instead.
class PathTraceIntegrator : public Integrator {
public:
    explicit PathTraceIntegrator(std::vector<int> devices = {0}, uint32_t rank = 0, uint32_t world = 1);
    ~PathTraceIntegrator() override;

    void prepareForRender(Scene &scene, uint32_t seed) override;   // PathTraceIntegrator.cpp:184-201
    void teardownAfterRender() override;                           // :203-213
    void startRender(std::function<void()> completionCallback) override; // :220-239 (asynchronous)
    void waitForCompletion() override;                             // :241-247 (rethrows worker errors)
    void abortRender() override;                                   // :249-256
    void frameBuffer(std::vector<float> &rgb) override;

    // SampleRecords (adaptive.hpp), one per 4x4 variance tile of the whole image
    const std::vector<SampleRecord> &sampleRecords() const { return _work.samples(); }
    // Multi-process renders with adaptive sampling: called once per pass with the whole-image
    // {sampleCount, mean, runningVariance} table holding this process' variance tiles; must fill in the
    // other ranks' tiles (an all-reduce/all-gather; tungsten_amd/dist.py does it over torch.distributed).
    std::function<void(std::vector<tgpu_sample_record> &)> recordExchange;

private:
    void collectRecords();

    AdaptiveScheduler _work;

    struct Device {
        int id;
        tgpu_ctx *ctx = nullptr;
        Shard shard;
    };
    std::vector<Device> _devices;
    uint32_t _rank, _world;
    std::thread _waiter;
    std::atomic<bool> _aborted{false};
    std::string _error;
};

// IntegratorFactory (src/core/integrators/IntegratorFactory.cpp:14-23): only "path_tracer" is in scope.
std::shared_ptr<Integrator> instantiateIntegrator(const std::string &type);

} // namespace tungsten_amd
