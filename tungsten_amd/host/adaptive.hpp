// Pass scheduler of the path tracer: how many samples every 4x4-pixel variance tile gets in the next pass.
//
// Behaviour = the reference's generateWork and its helpers (src/core/integrators/path_tracer/PathTraceIntegrator.cpp:44-134 with
// SampleRecord.hpp:11-66): below 16 spp every tile gets the pass's spp; from then on one sample per pixel is guaranteed and the rest of the
// budget follows the tiles' relative error (variance of the tile mean over its squared mean), clipped at the 95th percentile, spread to the
// four neighbours, and rounded stochastically with the integrator's own PCG stream (the draws after the tile seeds).  The devices own the
// running statistics (k_accumulate_records); the host only sees them through setStatistics() between passes.
//
// Results must equal the reference's bit for bit (the sample counts decide which Sobol' indices exist), so every float operation keeps the
// reference's order and width (-ffp-contract=off).  The data layout is this host's: one array per field (the device tables are exactly the
// three statistics columns, the pass arguments exactly the index / next columns) instead of an array of records.
#pragma once
#include "../../include/tungsten_amd.h"
#include "scene.hpp"

#include <algorithm>
#include <cstdint>
#include <vector>

namespace tungsten_amd {

// One tile's entry of the resume file (Integrator.cpp:108-129 writes the reference's 24-byte SampleRecord: these six words in this order).
struct TileState { // an aggregate (this header is also compiled as C++11 inside the reference tree): value-initialise with TileState()
    uint32_t sampleCount, nextSampleCount, sampleIndex;
    float adaptiveWeight, mean, runningVariance;
};

class AdaptiveScheduler {
public:
    static constexpr uint32_t TileSize = 16, VarianceTileSize = 4, AdaptiveThreshold = 16; // PathTraceIntegrator.hpp:26-29

    void init(uint32_t resX, uint32_t resY, uint32_t seed)
    {
        _pixels = int(resX)*int(resY);
        _cols = (resX + VarianceTileSize - 1)/VarianceTileSize;
        _rows = (resY + VarianceTileSize - 1)/VarianceTileSize;
        const size_t n = size_t(_cols)*_rows;
        _count.assign(n, 0u); _next.assign(n, 0u); _index.assign(n, 0u);
        _weight.assign(n, 0.0f); _mean.assign(n, 0.0f); _m2.assign(n, 0.0f);
        // the integrator's generator: seeded with hash32(seed), one draw per 16x16 image tile already spent on the tile seeds (:29-38,187)
        _rng = Pcg32(hash32(seed));
        const uint32_t imageTiles = ((resX + TileSize - 1)/TileSize)*((resY + TileSize - 1)/TileSize);
        for (uint32_t i = 0; i < imageTiles; ++i) (void)_rng.nextI();
    }

    uint32_t varW() const { return _cols; }
    uint32_t varH() const { return _rows; }
    size_t tileCount() const { return _count.size(); }

    // the pass [currentSpp, ...) is steered by the statistics: the caller brings the devices' tables in first
    static bool needsStatistics(bool adaptive, uint32_t currentSpp) { return adaptive && currentSpp >= AdaptiveThreshold; }

    void setStatistics(const tgpu_sample_record *table)
    {
        for (size_t i = 0; i < _count.size(); ++i) { _count[i] = table[i].sample_count; _mean[i] = table[i].mean; _m2[i] = table[i].running_variance; }
    }

    // Decides the pass [currentSpp, nextSpp).  false: nothing left to refine (every tile's error is zero), the pass renders nothing.
    bool generateWork(uint32_t currentSpp, uint32_t nextSpp, bool adaptive)
    {
        const size_t n = _count.size();
        for (size_t i = 0; i < n; ++i) _index[i] += _next[i];
        const int spp = int(nextSpp - currentSpp);
        if (!needsStatistics(adaptive, currentSpp)) {
            std::fill(_next.begin(), _next.end(), uint32_t(spp));
            return true;
        }
        // relative error of every tile; its 95th percentile over the tiles that have one caps the weights
        std::vector<float> ranked;
        ranked.reserve(n);
        for (size_t i = 0; i < n; ++i) {
            const float variance = _m2[i]/float(_count[i] - 1u);
            const float meanSq = _mean[i]*_mean[i];
            _weight[i] = variance/(float(_count[i])*(meanSq > 1e-3f ? meanSq : 1e-3f));
            if (_weight[i] > 0.0f) ranked.push_back(_weight[i]);
        }
        if (ranked.empty()) return false;
        std::sort(ranked.begin(), ranked.end());
        const float cap = ranked[(ranked.size()*95)/100];
        if (cap == 0.0f) return false;
        for (float &w : _weight) w = w < cap ? w : cap;
        // a tile takes the larger weight of its neighbours: one sweep pulling from below and right, one from above and left (in place, so weights travel)
        pullFromNeighbours(+1);
        pullFromNeighbours(-1);
        // one sample per pixel is fixed; the others are shared out in proportion to the weights, fractions carried from tile to tile
        double weightSum = 0.0;
        for (float w : _weight) weightSum += w;
        const int perTile = ((spp - 1)*_pixels)/int(VarianceTileSize*VarianceTileSize);
        const float samplesPerWeight = float(double(perTile)/weightSum);
        float carry = 0.0f;
        for (size_t i = 0; i < n; ++i) {
            const float share = _weight[i]*samplesPerWeight;
            int whole = int(share);
            carry += share - float(whole);
            if (_rng.next1D() < carry) {
                whole++;
                carry -= 1.0f;
            }
            _next[i] = uint32_t(whole + 1);
        }
        return true;
    }

    // arguments of tgpu_render_pass_adaptive
    void passArrays(std::vector<uint32_t> &sampleIndex, std::vector<uint32_t> &sampleCount) const { sampleIndex = _index; sampleCount = _next; }

    TileState tile(size_t i) const { return TileState{_count[i], _next[i], _index[i], _weight[i], _mean[i], _m2[i]}; }
    void setTile(size_t i, const TileState &t)
    {
        _count[i] = t.sampleCount; _next[i] = t.nextSampleCount; _index[i] = t.sampleIndex;
        _weight[i] = t.adaptiveWeight; _mean[i] = t.mean; _m2[i] = t.runningVariance;
    }

private:
    // dir = +1: rows top to bottom, tiles left to right, each taking max(self, tile below, tile to the right); dir = -1: the mirror image
    void pullFromNeighbours(int dir)
    {
        const int cols = int(_cols), rows = int(_rows);
        for (int r = 0; r < rows; ++r) {
            const int y = dir > 0 ? r : rows - 1 - r;
            const bool hasRow = dir > 0 ? y < rows - 1 : y > 0;
            for (int c = 0; c < cols; ++c) {
                const int x = dir > 0 ? c : cols - 1 - c;
                const bool hasCol = dir > 0 ? x < cols - 1 : x > 0;
                float &w = _weight[size_t(x) + size_t(y)*_cols];
                if (hasRow) { const float o = _weight[size_t(x) + size_t(y + dir)*_cols]; w = w > o ? w : o; }
                if (hasCol) { const float o = _weight[size_t(x + dir) + size_t(y)*_cols]; w = w > o ? w : o; }
            }
        }
    }

    int _pixels = 0;
    uint32_t _cols = 0, _rows = 0;
    std::vector<uint32_t> _count, _next, _index;   // samples a tile has, gets in the coming pass, and the first sample index of that pass
    std::vector<float> _weight, _mean, _m2;        // adaptive weight; running mean and sum of squared deviations of the tile's samples
    Pcg32 _rng;
};

} // namespace tungsten_amd
