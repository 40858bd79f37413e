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
#include <iosfwd>
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
    // Integrator::saveRenderResumeData / resumeRender (Integrator.cpp:108-162) on the reference's file layout: a JSON header
    // {current_spp, adaptive_sampling, stratified_sampler} + NUL, a 64-bit scene hash, the colour buffer (RGB float32 per pixel,
    // then uint32 sample counts), then the integrator's saveState()/loadState() block.  The hash is BitManip::hash (65599
    // polynomial) of THIS loader's scene text without its "renderer" member -- the reference hashes its own re-serialisation of
    // the scene, which only it can produce.  A file whose hash differs is REFUSED, as the reference refuses it (Integrator.cpp:146-148): the
    // default file name is shared by every scene of a directory, and another scene's buffers of the same size would otherwise load -- and, if
    // complete, be saved as this scene's image.  `acceptForeignHash` (CLI: --accept-foreign-resume-file) is the explicit opt-in for a file
    // written by the reference renderer for the same scene; `hashMismatch` reports that the opt-in was needed.
    void saveRenderResumeData();
    bool resumeRender(bool acceptForeignHash = false, bool *hashMismatch = nullptr);

    bool done() const { return _currentSpp >= _spp; }       // Integrator.hpp:52-55
    uint32_t currentSpp() const { return _currentSpp; }
    uint32_t nextSpp() const { return _nextSpp; }

    // Camera::colorBuffer() contents in image order (what the reference reads through Camera::getLinear)
    virtual void frameBuffer(std::vector<float> &rgb) = 0;
    // OutputBuffer::_bufferA / _sampleCount in image order, out and back in (serialize / deserialize, OutputBuffer.hpp:191-211)
    virtual void downloadOutputBuffer(std::vector<float> &rgb, std::vector<uint32_t> &counts) = 0;
    virtual void uploadOutputBuffer(const std::vector<float> &rgb, const std::vector<uint32_t> &counts) = 0;
    // Any part (tgpu_output_part) of any output buffer (tgpu_output_type) as 4-byte words in image order: pixels x channels floats,
    // or pixels uint32 counts.  hasOutputPart: was it requested (renderer "output_buffers"; colour A / COUNT always are)?
    bool hasOutputPart(int type, int part) const;
    virtual void downloadOutputPart(int type, int part, std::vector<uint32_t> &words) = 0;
    virtual void uploadOutputPart(int type, int part, const std::vector<uint32_t> &words) = 0;
    // OutputBuffer::operator[] for every pixel (OutputBuffer.hpp:134-144): the mean, of both half buffers where there are two
    void outputBufferMean(int type, std::vector<float> &values);
    // Camera::saveOutputBuffers -> OutputBuffer::save (Camera.cpp:213-220, OutputBuffer.hpp:146-189)
    void saveOutputBuffers();

protected:
    virtual void saveState(std::ostream &out) = 0;  // Integrator.hpp:33-34
    virtual bool loadState(std::istream &in) = 0;
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
    void downloadOutputBuffer(std::vector<float> &rgb, std::vector<uint32_t> &counts) override;
    void uploadOutputBuffer(const std::vector<float> &rgb, const std::vector<uint32_t> &counts) override;
    void downloadOutputPart(int type, int part, std::vector<uint32_t> &words) override;
    void uploadOutputPart(int type, int part, const std::vector<uint32_t> &words) override;
    bool supportsResumeRender() const override { return true; }

    // SampleRecords (adaptive.hpp), one per 4x4 variance tile of the whole image
    const AdaptiveScheduler &scheduler() const { return _work; }   // the SampleRecord columns (tile(i)) and what the next pass will get
    // Multi-process renders with adaptive sampling: called once per pass with the whole-image
    // {sampleCount, mean, runningVariance} table holding this process' variance tiles; must fill in the
    // other ranks' tiles (an all-reduce/all-gather; tungsten_amd/dist.py does it over torch.distributed).
    std::function<void(std::vector<tgpu_sample_record> &)> recordExchange;

protected:
    // PathTraceIntegrator::saveState / loadState (PathTraceIntegrator.cpp:158-172): the SampleRecords (24 bytes each, SampleRecord.hpp:24-32),
    // then per 16x16 tile the sampler state -- {seed, PCG state} of a SobolPathSampler (SobolPathSampler.hpp:35-45), the PCG state
    // alone of a UniformPathSampler.  The supplemental stream of this core is path-local, so the PCG states carry no information
    // here: the tile's initial state is written, what a file holds is ignored.
    void saveState(std::ostream &out) override;
    bool loadState(std::istream &in) override;

private:
    void collectRecords();
    uint32_t _seed = 0;

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
