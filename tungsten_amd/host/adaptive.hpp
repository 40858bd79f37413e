// The work generator of PathTraceIntegrator (src/core/integrators/path_tracer/PathTraceIntegrator.cpp:44-134):
// per 4x4-pixel variance tile a SampleRecord (path_tracer/SampleRecord.hpp:11-66) whose running statistics the
// devices accumulate (k_accumulate_records); between passes the host turns them into the next pass' per-tile
// sample counts.  Float operations keep the reference's order and width (the library is built with
// -ffp-contract=off), the stochastic rounding consumes the integrator's own PCG stream after the tile seeds.
#pragma once
#include "../../include/tungsten_amd.h"
#include "scene.hpp"

#include <algorithm>
#include <cstdint>
#include <vector>

namespace tungsten_amd {

struct SampleRecord {
    uint32_t sampleCount = 0, nextSampleCount = 0, sampleIndex = 0;
    float adaptiveWeight = 0.0f, mean = 0.0f, runningVariance = 0.0f;
    float variance() const { return runningVariance/float(sampleCount - 1u); }
    float errorEstimate() const
    {
        float m = mean*mean;
        return variance()/(float(sampleCount)*(m > 1e-3f ? m : 1e-3f)); // max(mean*mean, 1e-3f)
    }
};

class AdaptiveScheduler {
public:
    static constexpr uint32_t TileSize = 16, VarianceTileSize = 4, AdaptiveThreshold = 16; // PathTraceIntegrator.hpp:26-29

    void init(uint32_t resX, uint32_t resY, uint32_t seed)
    {
        _w = resX;
        _h = resY;
        // the integrator's sampler: seeded at :187, then one draw per 16x16 tile for the tile seeds (diceTiles :29-38)
        _sampler = Pcg32(hash32(seed));
        const uint32_t nTiles = ((resX + TileSize - 1)/TileSize)*((resY + TileSize - 1)/TileSize);
        for (uint32_t i = 0; i < nTiles; ++i) (void)_sampler.nextI();
        _varW = (resX + VarianceTileSize - 1)/VarianceTileSize; // :190-193
        _varH = (resY + VarianceTileSize - 1)/VarianceTileSize;
        _samples.assign(size_t(_varW)*_varH, SampleRecord{});
    }

    uint32_t varW() const { return _varW; }
    uint32_t varH() const { return _varH; }
    std::vector<SampleRecord> &samples() { return _samples; }
    const std::vector<SampleRecord> &samples() const { return _samples; }

    // True when the coming pass [currentSpp, nextSpp) consults the statistics, i.e. the caller has to
    // bring the devices' records in with setStatistics() before generateWork().
    static bool needsStatistics(bool adaptive, uint32_t currentSpp) { return adaptive && currentSpp >= AdaptiveThreshold; }

    void setStatistics(const tgpu_sample_record *table)
    {
        for (size_t i = 0; i < _samples.size(); ++i) {
            _samples[i].sampleCount = table[i].sample_count;
            _samples[i].mean = table[i].mean;
            _samples[i].runningVariance = table[i].running_variance;
        }
    }

    // generateWork (:110-134); false = converged, the pass renders nothing
    bool generateWork(uint32_t currentSpp, uint32_t nextSpp, bool adaptive)
    {
        for (SampleRecord &r : _samples) r.sampleIndex += r.nextSampleCount;
        int sppCount = int(nextSpp - currentSpp);
        if (needsStatistics(adaptive, currentSpp)) {
            float maxError = errorPercentile95();
            if (maxError == 0.0f) return false;
            for (SampleRecord &r : _samples) r.adaptiveWeight = r.adaptiveWeight < maxError ? r.adaptiveWeight : maxError; // min()
            dilateAdaptiveWeights();
            distributeAdaptiveSamples(sppCount);
        } else {
            for (SampleRecord &r : _samples) r.nextSampleCount = uint32_t(sppCount);
        }
        return true;
    }

    void passArrays(std::vector<uint32_t> &sampleIndex, std::vector<uint32_t> &sampleCount) const
    {
        sampleIndex.resize(_samples.size());
        sampleCount.resize(_samples.size());
        for (size_t i = 0; i < _samples.size(); ++i) {
            sampleIndex[i] = _samples[i].sampleIndex;
            sampleCount[i] = _samples[i].nextSampleCount;
        }
    }

private:
    float errorPercentile95() // :44-59
    {
        std::vector<float> errors;
        errors.reserve(_samples.size());
        for (SampleRecord &r : _samples) {
            r.adaptiveWeight = r.errorEstimate();
            if (r.adaptiveWeight > 0.0f) errors.push_back(r.adaptiveWeight);
        }
        if (errors.empty()) return 0.0f;
        std::sort(errors.begin(), errors.end());
        return errors[(errors.size()*95)/100];
    }

    void dilateAdaptiveWeights() // :61-85
    {
        auto grow = [](float &a, float b) { a = a > b ? a : b; }; // Tungsten::max (MathUtil.hpp:22-26)
        for (uint32_t y = 0; y < _varH; ++y)
            for (uint32_t x = 0; x < _varW; ++x) {
                size_t idx = x + size_t(y)*_varW;
                if (y < _varH - 1) grow(_samples[idx].adaptiveWeight, _samples[idx + _varW].adaptiveWeight);
                if (x < _varW - 1) grow(_samples[idx].adaptiveWeight, _samples[idx + 1].adaptiveWeight);
            }
        for (int y = int(_varH) - 1; y >= 0; --y)
            for (int x = int(_varW) - 1; x >= 0; --x) {
                size_t idx = size_t(x) + size_t(y)*_varW;
                if (y > 0) grow(_samples[idx].adaptiveWeight, _samples[idx - _varW].adaptiveWeight);
                if (x > 0) grow(_samples[idx].adaptiveWeight, _samples[idx - 1].adaptiveWeight);
            }
    }

    void distributeAdaptiveSamples(int spp) // :87-108
    {
        double totalWeight = 0.0;
        for (SampleRecord &r : _samples) totalWeight += r.adaptiveWeight;
        int adaptiveBudget = (spp - 1)*int(_w)*int(_h);
        int budgetPerTile = adaptiveBudget/int(VarianceTileSize*VarianceTileSize);
        float weightToSampleFactor = float(double(budgetPerTile)/totalWeight);
        float pixelPdf = 0.0f;
        for (SampleRecord &r : _samples) {
            float fractionalSamples = r.adaptiveWeight*weightToSampleFactor;
            int adaptiveSamples = int(fractionalSamples);
            pixelPdf += fractionalSamples - float(adaptiveSamples);
            if (_sampler.next1D() < pixelPdf) {
                adaptiveSamples++;
                pixelPdf -= 1.0f;
            }
            r.nextSampleCount = uint32_t(adaptiveSamples + 1);
        }
    }

    uint32_t _w = 0, _h = 0, _varW = 0, _varH = 0;
    std::vector<SampleRecord> _samples;
    Pcg32 _sampler;
};

} // namespace tungsten_amd
