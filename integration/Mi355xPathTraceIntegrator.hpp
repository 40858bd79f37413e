// REFERENCE-SIDE binding: the Integrator subclass a Tungsten maintainer adds under
// src/core/integrators/mi355x/ and registers as "path_tracer" (IntegratorFactory.cpp:14-23).  It is compiled here against the reference's
// own headers where they lie (oracle/Makefile.ref, target integration-check: syntax and type check only -- the reference's libraries
// cannot be built in this image), so that the signatures of include/tungsten_amd.h and of Tungsten::Integrator are known to fit.
#ifndef MI355XPATHTRACEINTEGRATOR_HPP_
#define MI355XPATHTRACEINTEGRATOR_HPP_

#include "integrators/Integrator.hpp"
#include "integrators/path_tracer/PathTracerSettings.hpp"

#include <tungsten_amd.h>

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
    std::vector<uint32> _pixelIndex, _tileSeed;   // the shard: every 16x16 tile of the image
    std::thread _waiter;
    std::exception_ptr _error;

    void flatten(const TraceableScene &scene);
    void publish();

    virtual void saveState(OutputStreamHandle &out) override;
    virtual void loadState(InputStreamHandle &in) override;

public:
    Mi355xPathTraceIntegrator();
    ~Mi355xPathTraceIntegrator();

    virtual void fromJson(JsonPtr value, const Scene &scene) override;
    virtual rapidjson::Value toJson(Allocator &allocator) const override;

    virtual void prepareForRender(TraceableScene &scene, uint32 seed) override;
    virtual void teardownAfterRender() override;

    virtual void startRender(std::function<void()> completionCallback) override;
    virtual void waitForCompletion() override;
    virtual void abortRender() override;
};

}

#endif /* MI355XPATHTRACEINTEGRATOR_HPP_ */
