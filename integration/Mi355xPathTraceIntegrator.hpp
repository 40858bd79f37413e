// REFERENCE-SIDE binding: the Integrator subclass a Tungsten maintainer adds under
// src/core/integrators/mi355x/ and registers as "path_tracer" (IntegratorFactory.cpp:14-23).  It is compiled here against the reference's
// own headers where they lie (oracle/Makefile.ref, target integration-check: syntax and type check only -- the reference's libraries
// cannot be built in this image), so that the signatures of include/tungsten_amd.h and of Tungsten::Integrator are known to fit.
//
// Everything PathTraceIntegrator does for its callers is here: uniform and adaptive passes (generateWork, PathTraceIntegrator.cpp:110-134,220-239),
// the camera's colour buffer and auxiliary output buffers with all their parts (Camera.cpp:148-238), resume state (saveState / loadState :158-172,
// supportsResumeRender :215-218) and an abort that does not throw (:249-256).
#ifndef MI355XPATHTRACEINTEGRATOR_HPP_
#define MI355XPATHTRACEINTEGRATOR_HPP_

#include "integrators/Integrator.hpp"
#include "integrators/path_tracer/PathTracerSettings.hpp"

#include <tungsten_amd.h>
#include <host/adaptive.hpp> // tungsten_amd/host: the pass scheduler (generateWork) shared with this repository's own host

#include <atomic>
#include <exception>
#include <memory>
#include <thread>
#include <vector>

namespace tungsten_amd { class Scene; }

namespace Tungsten {

class Mi355xPathTraceIntegrator : public Integrator
{
    PathTracerSettings _settings;
    tgpu_ctx *_ctx = nullptr;
    std::unique_ptr<tungsten_amd::Scene> _flat;   // FlattenedScene: the tgpu_scene_desc and the buffers behind it
    std::vector<uint32> _pixelIndex, _tileSeed;   // the shard: every 16x16 tile of the image, pixel by pixel
    std::vector<uint32> _seedOfTile;              // one seed per tile, in diceTiles' order (saveState writes them)
    tungsten_amd::AdaptiveScheduler _work;        // SampleRecord columns + generateWork
    std::thread _waiter;
    std::exception_ptr _error;
    std::atomic<bool> _aborted{false};

    void flatten(const TraceableScene &scene);
    bool adaptive() const;
    bool generateWork(std::vector<uint32> &sampleIndex, std::vector<uint32> &sampleCount);
    void publish();
    void publishBuffer(int type, int channels, const std::function<void(InputStreamHandle &)> &deserialize);
    void uploadBuffer(int type, int channels, const std::function<void(OutputStreamHandle &)> &serialize);
    void check(int status) const;

    virtual void saveState(OutputStreamHandle &out) override;
    virtual void loadState(InputStreamHandle &in) override;

public:
    Mi355xPathTraceIntegrator();
    ~Mi355xPathTraceIntegrator();

    virtual void fromJson(JsonPtr value, const Scene &scene) override;
    virtual rapidjson::Value toJson(Allocator &allocator) const override;

    virtual void prepareForRender(TraceableScene &scene, uint32 seed) override;
    virtual void teardownAfterRender() override;

    virtual bool supportsResumeRender() const override;

    virtual void startRender(std::function<void()> completionCallback) override;
    virtual void waitForCompletion() override;
    virtual void abortRender() override;
};

}

#endif /* MI355XPATHTRACEINTEGRATOR_HPP_ */
