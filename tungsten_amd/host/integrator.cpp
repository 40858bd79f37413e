#include "integrator.hpp"

#include <algorithm>
#include <cstdio>
#include <cmath>
#include <cstring>
#include <fstream>
#include <limits>
#include <sstream>
#include <stdexcept>

namespace tungsten_amd {

static void check(tgpu_ctx *ctx, int rc, const char *what)
{
    if (rc != TGPU_OK)
        throw std::runtime_error(std::string(what) + " failed (" + std::to_string(rc) + "): " + tgpu_last_error(ctx));
}

static int outputChannels(int type) { return (type == TGPU_OUTPUT_DEPTH || type == TGPU_OUTPUT_VISIBILITY) ? 1 : 3; }

bool Integrator::hasOutputPart(int type, int part) const
{
    const tgpu_output_buffer &b = _scene->desc().output_buffers[type];
    if (type != TGPU_OUTPUT_COLOR && !b.enabled) return false;
    if (part == TGPU_PART_B) return b.two_buffer_variance != 0;
    if (part == TGPU_PART_VARIANCE) return b.sample_variance != 0;
    return true;
}

void Integrator::outputBufferMean(int type, std::vector<float> &values)
{
    const int channels = outputChannels(type);
    std::vector<uint32_t> a, b, n;
    downloadOutputPart(type, TGPU_PART_A, a);
    values.resize(a.size());
    std::memcpy(values.data(), a.data(), a.size()*sizeof(float));
    if (!hasOutputPart(type, TGPU_PART_B)) return;
    downloadOutputPart(type, TGPU_PART_B, b);
    downloadOutputPart(type, TGPU_PART_COUNT, n);
    for (size_t i = 0; i < n.size(); ++i) {
        const uint32_t sampleIdx = n[i], sampleCountA = (sampleIdx + 1)/2, sampleCountB = sampleIdx/2;
        for (int c = 0; c < channels; ++c) {
            float fa, fb;
            std::memcpy(&fa, &a[i*channels + c], 4);
            std::memcpy(&fb, &b[i*channels + c], 4);
            values[i*channels + c] = (fa*float(sampleCountA) + fb*float(sampleCountB))/float(std::max(sampleIdx, 1u));
        }
    }
}

// OutputBuffer::saveLdr (OutputBuffer.hpp:57-88): depth is rescaled to its finite maximum, normals from [-1, 1], the rest as is
static void saveOutputLdr(int type, const float *hdr, int channels, int w, int h, const std::string &path, bool rescale)
{
    const size_t pixelCount = size_t(w)*h;
    float minimum[3] = {0.0f, 0.0f, 0.0f}, maximum[3] = {0.0f, 0.0f, 0.0f};
    auto average = [&](const float *p) { return channels == 1 ? p[0] : (p[0] + p[1] + p[2])*(1.0f/3.0f); };
    if (type == TGPU_OUTPUT_DEPTH) {
        for (size_t i = 0; i < pixelCount; ++i)
            if (average(hdr + i*channels) != std::numeric_limits<float>::infinity())
                for (int c = 0; c < channels; ++c) maximum[c] = std::max(maximum[c], hdr[i*channels + c]);
    } else if (type == TGPU_OUTPUT_NORMAL) {
        for (int c = 0; c < 3; ++c) { minimum[c] = -1.0f; maximum[c] = 1.0f; }
    } else {
        rescale = false;
    }
    std::vector<uint8_t> ldr(pixelCount*3);
    for (size_t i = 0; i < pixelCount; ++i) {
        float f[3] = {0.0f, 0.0f, 0.0f};
        for (int c = 0; c < channels; ++c) {
            f[c] = hdr[i*channels + c];
            if (rescale) f[c] = (f[c] - minimum[c])/(maximum[c] - minimum[c]);
        }
        const float avg = average(f);
        for (int c = 0; c < 3; ++c) {
            const float v = f[channels == 1 ? 0 : c];
            ldr[i*3 + c] = (std::isnan(avg) || std::isinf(avg)) ? uint8_t(255) : uint8_t(std::min(std::max(int(v*255.0f), 0), 255));
        }
    }
    if (!savePng(path, ldr.data(), w, h)) throw std::runtime_error("Unable to write " + path);
}

static void saveOutputHdr(const std::string &path, const float *img, int w, int h, int channels)
{
    if (path.size() < 4 || path.substr(path.size() - 4) != ".pfm")
        throw std::runtime_error("hdr_output_file '" + path + "': only .pfm is supported (no OpenEXR here, as in the reference build)");
    if (!savePfm(path, img, w, h, channels)) throw std::runtime_error("Unable to write " + path);
}

void Integrator::saveOutputBuffers()
{
    const int w = int(_scene->resX), h = int(_scene->resY);
    auto withTag = [](const std::string &file, const char *tag) { // Path::stripExtension() + tag + Path::extension()
        size_t dot = file.find_last_of('.');
        return dot == std::string::npos ? file + tag : file.substr(0, dot) + tag + file.substr(dot);
    };
    for (int type = 0; type < TGPU_OUTPUT_COUNT; ++type) { // colour, depth, normal, albedo, visibility (Camera.cpp:213-220)
        const OutputBufferSettings *settings = nullptr;
        for (const OutputBufferSettings &b : _scene->renderer.outputs)
            if (b.type == type) settings = &b;
        if (!settings) continue;
        const int channels = outputChannels(type);
        std::vector<float> mean;
        outputBufferMean(type, mean);
        const std::string &ldrFile = settings->ldrOutputFile, &hdrFile = settings->hdrOutputFile;
        if (!hdrFile.empty()) saveOutputHdr(hdrFile, mean.data(), w, h, channels);
        if (!ldrFile.empty()) saveOutputLdr(type, mean.data(), channels, w, h, ldrFile, true);
        if (hasOutputPart(type, TGPU_PART_B)) {
            for (int part : {TGPU_PART_A, TGPU_PART_B}) {
                std::vector<uint32_t> words;
                downloadOutputPart(type, part, words);
                const float *half = reinterpret_cast<const float *>(words.data());
                const char *tag = part == TGPU_PART_A ? "A" : "B";
                if (!hdrFile.empty()) saveOutputHdr(withTag(hdrFile, tag), half, w, h, channels);
                if (!ldrFile.empty()) saveOutputLdr(type, half, channels, w, h, withTag(ldrFile, tag), true);
            }
        }
        if (hasOutputPart(type, TGPU_PART_VARIANCE)) {
            std::vector<uint32_t> var, n;
            downloadOutputPart(type, TGPU_PART_VARIANCE, var);
            downloadOutputPart(type, TGPU_PART_COUNT, n);
            std::vector<float> scaled(var.size());
            for (size_t i = 0; i < n.size(); ++i)
                for (int c = 0; c < channels; ++c) {
                    float v;
                    std::memcpy(&v, &var[i*channels + c], 4);
                    scaled[i*channels + c] = v/float(n[i]*std::max(1u, n[i] - 1u));
                }
            if (!hdrFile.empty()) saveOutputHdr(withTag(hdrFile, "Variance"), scaled.data(), w, h, channels);
            if (!ldrFile.empty()) saveOutputLdr(type, scaled.data(), channels, w, h, withTag(ldrFile, "Variance"), false);
        }
    }
}

void Integrator::writeBuffers(const std::string &suffix, bool overwrite)
{
    std::vector<float> hdr;
    frameBuffer(hdr);
    const RendererSettings &r = _scene->renderer;
    // Where an image goes: the configured name with the suffix in front of its extension; with overwrite_output_files off, the first of
    // name, name001, name002, ... that does not exist yet (the stock scenes ship that way: a re-render keeps the earlier images).  Same names
    // as the reference's incrementalFilename (Integrator.cpp:24-38).
    auto target = [&](const std::string &file) {
        size_t slash = file.find_last_of("/\\");
        size_t dot = file.find_last_of('.');
        if (dot == std::string::npos || (slash != std::string::npos && dot < slash)) dot = file.size();
        const std::string stem = file.substr(0, dot) + suffix, extension = file.substr(dot);
        std::string path = stem + extension;
        if (overwrite) return path;
        char number[16];
        for (int index = 1; std::ifstream(path).good(); ++index) {
            std::snprintf(number, sizeof(number), "%03d", index);
            path = stem + number + extension;
        }
        return path;
    };
    auto withSuffix = target;
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
    if (suffix.empty() && !r.outputs.empty()) saveOutputBuffers(); // Integrator.cpp:78-79
}

// the scene text without its top-level "renderer" member (the reference leaves the renderer settings out of the hash so that spp or
// output names can change between runs, Integrator.cpp:92-106)
static std::string withoutRenderer(const std::string &text)
{
    int depth = 0;
    for (size_t i = 0; i < text.size(); ++i) {
        char c = text[i];
        if (c == '"') {
            size_t end = i + 1;
            while (end < text.size() && text[end] != '"') end += text[end] == '\\' ? 2 : 1;
            if (depth == 1 && text.compare(i, end - i + 1, "\"renderer\"") == 0) {
                size_t j = end + 1;
                while (j < text.size() && text[j] != '{' && text[j] != ',' && text[j] != '}') ++j;
                if (j < text.size() && text[j] == '{') {
                    int d = 0;
                    for (; j < text.size(); ++j) {
                        if (text[j] == '"') { ++j; while (j < text.size() && text[j] != '"') j += text[j] == '\\' ? 2 : 1; }
                        else if (text[j] == '{') ++d;
                        else if (text[j] == '}' && --d == 0) { ++j; break; }
                    }
                    return text.substr(0, i) + text.substr(j);
                }
            }
            i = end;
        } else if (c == '{' || c == '[') ++depth;
        else if (c == '}' || c == ']') --depth;
    }
    return text;
}
static uint64_t sceneHash(const Scene &scene) // BitManip::hash (math/BitManip.hpp:82-88)
{
    uint64_t result = 0;
    for (char c : withoutRenderer(scene.sceneJsonText)) result = result*65599ull + uint64_t(c);
    return result;
}

void Integrator::saveRenderResumeData()
{
    const std::string &path = _scene->renderer.resumeRenderFile;
    std::ofstream out(path, std::ios::binary);
    if (!out) {
        std::fprintf(stderr, "Failed to open render resume state at '%s'\n", path.c_str());
        return;
    }
    // rapidjson::PrettyWriter's default layout (io/JsonUtils.cpp: jsonToString): four spaces, one member per line
    std::ostringstream header;
    header << "{\n    \"current_spp\": " << _currentSpp << ",\n    \"adaptive_sampling\": " << (_scene->renderer.useAdaptiveSampling ? "true" : "false")
           << ",\n    \"stratified_sampler\": " << (_scene->renderer.useSobol ? "true" : "false") << "\n}";
    const std::string h = header.str();
    out.write(h.c_str(), std::streamsize(h.size() + 1)); // FileUtils::streamWrite<std::string>: the characters and a NUL
    const uint64_t hash = sceneHash(*_scene);
    out.write(reinterpret_cast<const char *>(&hash), sizeof(hash));
    // Camera::serializeOutputBuffers (Camera.cpp:222-229): colour, depth, normal, albedo, visibility; of each the parts it has, in the
    // order of OutputBuffer::serialize (OutputBuffer.hpp:216-226)
    std::vector<uint32_t> words;
    for (int type = 0; type < TGPU_OUTPUT_COUNT; ++type)
        for (int part : {TGPU_PART_A, TGPU_PART_B, TGPU_PART_VARIANCE, TGPU_PART_COUNT}) {
            if (!hasOutputPart(type, part)) continue;
            downloadOutputPart(type, part, words);
            out.write(reinterpret_cast<const char *>(words.data()), std::streamsize(words.size()*sizeof(uint32_t)));
        }
    saveState(out);
}

bool Integrator::resumeRender(bool acceptForeignHash, bool *hashMismatch)
{
    if (hashMismatch) *hashMismatch = false;
    std::ifstream in(_scene->renderer.resumeRenderFile, std::ios::binary);
    if (!in) return false;
    std::string header;
    std::getline(in, header, '\0');
    Json doc;
    try {
        doc = Json::parse(header);
    } catch (const std::exception &) {
        return false;
    }
    bool adaptiveSampling, stratifiedSampler;
    if (!doc.getField("adaptive_sampling", adaptiveSampling) || adaptiveSampling != _scene->renderer.useAdaptiveSampling) return false;
    if (!doc.getField("stratified_sampler", stratifiedSampler) || stratifiedSampler != _scene->renderer.useSobol) return false;
    int jsonSpp;
    if (!doc.getField("current_spp", jsonSpp) || jsonSpp < 0) return false;
    uint64_t jsonHash = 0;
    in.read(reinterpret_cast<char *>(&jsonHash), sizeof(jsonHash));
    if (!in) return false;
    if (jsonHash != sceneHash(*_scene)) { // Integrator.cpp:146-148: a state file of another scene is not this render's past
        if (!acceptForeignHash) return false;
        if (hashMismatch) *hashMismatch = true;
    }
    const size_t nPixels = size_t(_scene->resX)*_scene->resY;
    struct Part { int type, part; std::vector<uint32_t> words; };
    std::vector<Part> parts;
    for (int type = 0; type < TGPU_OUTPUT_COUNT; ++type)
        for (int part : {TGPU_PART_A, TGPU_PART_B, TGPU_PART_VARIANCE, TGPU_PART_COUNT}) {
            if (!hasOutputPart(type, part)) continue;
            parts.push_back(Part{type, part, std::vector<uint32_t>(nPixels*(part == TGPU_PART_COUNT ? 1 : outputChannels(type)))});
            in.read(reinterpret_cast<char *>(parts.back().words.data()), std::streamsize(parts.back().words.size()*sizeof(uint32_t)));
        }
    if (!in) return false; // another resolution or another set of output buffers
    if (!loadState(in)) return false;
    for (const Part &p : parts) uploadOutputPart(p.type, p.part, p.words);
    _currentSpp = uint32_t(jsonSpp);
    advanceSpp();
    return true;
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
    _seed = seed;
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
    std::vector<tgpu_sample_record> table(_work.tileCount(), tgpu_sample_record{0, 0.0f, 0.0f});
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
    if (hasOutputPart(TGPU_OUTPUT_COLOR, TGPU_PART_B)) { // Camera::getLinear reads OutputBuffer::operator[]
        outputBufferMean(TGPU_OUTPUT_COLOR, rgb);
        return;
    }
    rgb.assign(size_t(_scene->resX)*_scene->resY*3, 0.0f);
    std::vector<float> local;
    for (Device &d : _devices) {
        local.resize(d.shard.pixelIndex.size()*3);
        check(d.ctx, tgpu_download(d.ctx, local.data(), nullptr), "tgpu_download");
        for (size_t i = 0; i < d.shard.pixelIndex.size(); ++i)
            std::memcpy(&rgb[size_t(d.shard.pixelIndex[i])*3], &local[i*3], 3*sizeof(float));
    }
}

void PathTraceIntegrator::downloadOutputBuffer(std::vector<float> &rgb, std::vector<uint32_t> &counts)
{
    rgb.assign(size_t(_scene->resX)*_scene->resY*3, 0.0f);
    counts.assign(size_t(_scene->resX)*_scene->resY, 0u);
    std::vector<float> local;
    std::vector<uint32_t> localCnt;
    for (Device &d : _devices) {
        local.resize(d.shard.pixelIndex.size()*3);
        localCnt.resize(d.shard.pixelIndex.size());
        check(d.ctx, tgpu_download(d.ctx, local.data(), localCnt.data()), "tgpu_download");
        for (size_t i = 0; i < d.shard.pixelIndex.size(); ++i) {
            std::memcpy(&rgb[size_t(d.shard.pixelIndex[i])*3], &local[i*3], 3*sizeof(float));
            counts[d.shard.pixelIndex[i]] = localCnt[i];
        }
    }
}

void PathTraceIntegrator::uploadOutputBuffer(const std::vector<float> &rgb, const std::vector<uint32_t> &counts)
{
    std::vector<float> local;
    std::vector<uint32_t> localCnt;
    for (Device &d : _devices) {
        local.resize(d.shard.pixelIndex.size()*3);
        localCnt.resize(d.shard.pixelIndex.size());
        for (size_t i = 0; i < d.shard.pixelIndex.size(); ++i) {
            std::memcpy(&local[i*3], &rgb[size_t(d.shard.pixelIndex[i])*3], 3*sizeof(float));
            localCnt[i] = counts[d.shard.pixelIndex[i]];
        }
        check(d.ctx, tgpu_upload_framebuffer(d.ctx, local.data(), localCnt.data()), "tgpu_upload_framebuffer");
    }
}

void PathTraceIntegrator::downloadOutputPart(int type, int part, std::vector<uint32_t> &words)
{
    const size_t channels = part == TGPU_PART_COUNT ? 1 : size_t(outputChannels(type));
    words.assign(size_t(_scene->resX)*_scene->resY*channels, 0u);
    std::vector<uint32_t> local;
    for (Device &d : _devices) {
        local.resize(d.shard.pixelIndex.size()*channels);
        check(d.ctx, tgpu_download_output_buffer(d.ctx, type, part, local.data()), "tgpu_download_output_buffer");
        for (size_t i = 0; i < d.shard.pixelIndex.size(); ++i)
            std::memcpy(&words[size_t(d.shard.pixelIndex[i])*channels], &local[i*channels], channels*sizeof(uint32_t));
    }
}

void PathTraceIntegrator::uploadOutputPart(int type, int part, const std::vector<uint32_t> &words)
{
    const size_t channels = part == TGPU_PART_COUNT ? 1 : size_t(outputChannels(type));
    std::vector<uint32_t> local;
    for (Device &d : _devices) {
        local.resize(d.shard.pixelIndex.size()*channels);
        for (size_t i = 0; i < d.shard.pixelIndex.size(); ++i)
            std::memcpy(&local[i*channels], &words[size_t(d.shard.pixelIndex[i])*channels], channels*sizeof(uint32_t));
        check(d.ctx, tgpu_upload_output_buffer(d.ctx, type, part, local.data()), "tgpu_upload_output_buffer");
    }
}

void PathTraceIntegrator::saveState(std::ostream &out)
{
    if (_scene->renderer.useAdaptiveSampling) collectRecords(); // mean / variance live on the devices between passes
    for (size_t i = 0; i < _work.tileCount(); ++i) {
        // without adaptive_sampling the devices keep no SampleRecords (the reference updates them regardless but never reads
        // them then): the counters are filled in, mean and variance stay 0
        TileState w = _work.tile(i);
        if (!_scene->renderer.useAdaptiveSampling) w.sampleCount = 0;
        out.write(reinterpret_cast<const char *>(&w.sampleCount), 4);
        out.write(reinterpret_cast<const char *>(&w.nextSampleCount), 4);
        out.write(reinterpret_cast<const char *>(&w.sampleIndex), 4);
        out.write(reinterpret_cast<const char *>(&w.adaptiveWeight), 4);
        out.write(reinterpret_cast<const char *>(&w.mean), 4);
        out.write(reinterpret_cast<const char *>(&w.runningVariance), 4);
    }
    std::vector<uint32_t> seeds = tileSeeds(_scene->resX, _scene->resY, _seed);
    for (uint32_t seed : seeds) {
        const uint64_t state = seed; // UniformSampler(seed): _state = seed (UniformSampler.hpp:22-26)
        if (_scene->renderer.useSobol) out.write(reinterpret_cast<const char *>(&seed), 4);
        out.write(reinterpret_cast<const char *>(&state), 8);
    }
}

bool PathTraceIntegrator::loadState(std::istream &in)
{
    std::vector<TileState> records(_work.tileCount(), TileState());
    for (TileState &r : records) {
        in.read(reinterpret_cast<char *>(&r.sampleCount), 4);
        in.read(reinterpret_cast<char *>(&r.nextSampleCount), 4);
        in.read(reinterpret_cast<char *>(&r.sampleIndex), 4);
        in.read(reinterpret_cast<char *>(&r.adaptiveWeight), 4);
        in.read(reinterpret_cast<char *>(&r.mean), 4);
        in.read(reinterpret_cast<char *>(&r.runningVariance), 4);
    }
    std::vector<uint32_t> seeds = tileSeeds(_scene->resX, _scene->resY, _seed);
    for (uint32_t expect : seeds) {
        uint32_t seed = expect;
        uint64_t state;
        if (_scene->renderer.useSobol) in.read(reinterpret_cast<char *>(&seed), 4);
        in.read(reinterpret_cast<char *>(&state), 8);
        if (!in) return false;
        if (seed != expect) return false; // rendered with another seed: its samples do not continue this render's sequences
    }
    for (size_t i = 0; i < records.size(); ++i) _work.setTile(i, records[i]);
    if (_scene->renderer.useAdaptiveSampling) {
        std::vector<tgpu_sample_record> table(records.size());
        for (size_t i = 0; i < records.size(); ++i) table[i] = tgpu_sample_record{records[i].sampleCount, records[i].mean, records[i].runningVariance};
        for (Device &d : _devices) check(d.ctx, tgpu_upload_records(d.ctx, table.data()), "tgpu_upload_records");
    }
    return true;
}

std::shared_ptr<Integrator> instantiateIntegrator(const std::string &type)
{
    if (type == "path_tracer") return std::make_shared<PathTraceIntegrator>();
    throw JsonLoadException("Unknown integrator '" + type + "': this core provides 'path_tracer' only");
}

} // namespace tungsten_amd
