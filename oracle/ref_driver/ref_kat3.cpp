// Test-infrastructure driver (NOT product code), third set.  Compiled against the reference's own sources where they lie under
// /root/reference (no copies; oracle/Makefile.ref), it prints as JSON what cameras/OutputBuffer.hpp accumulates:
//   output_buffer  OutputBuffer<float> and OutputBuffer<Vec3f> (through the public addSample / serialize / operator[] only), for the four
//                  combinations of two_buffer_variance and sample_variance: 6 pixels fed 0..40 samples each from a fixed generator,
//                  some of them NaN or infinite (addSample drops those) -> the bits of _bufferA, _bufferB, _variance, _sampleCount and of
//                  the combined mean.  tests/test_ref_kat3.py replays the same samples through the oracle's accumulation.
#include <cstdio>
#include <cstring>
#include <limits>
#include <memory>
#include <sstream>
#include <string>
#include <vector>
#include <cameras/OutputBuffer.hpp>
#include <io/JsonDocument.hpp>

using namespace Tungsten;

static unsigned bits(float f) { unsigned u; std::memcpy(&u, &f, 4); return u; }

struct Gen { // the test restates this generator
    unsigned long long s;
    unsigned next() { s = s*6364136223846793005ull + 1442695040888963407ull; return unsigned(s >> 33); }
    float value()
    {
        unsigned r = next();
        unsigned kind = r % 23u;
        if (kind == 0) return std::numeric_limits<float>::quiet_NaN();
        if (kind == 1) return std::numeric_limits<float>::infinity();
        if (kind == 2) return -std::numeric_limits<float>::infinity();
        float mag = float(next() % 100000u)/1000.0f;      // 0 .. 100
        if (kind < 8) mag *= 1e-3f;
        return (r & 0x10000u) ? -mag : mag;
    }
};

template <typename T> struct Channels;
template <> struct Channels<float> { enum { N = 1 }; static float make(Gen &g) { return g.value(); } };
template <> struct Channels<Vec3f> { enum { N = 3 }; static Vec3f make(Gen &g) { float x = g.value(), y = g.value(), z = g.value(); return Vec3f(x, y, z); } };

static void printWords(const char *name, const std::string &blob, size_t &pos, size_t count, bool last = false)
{
    std::printf("\"%s\": [", name);
    for (size_t i = 0; i < count; ++i) {
        unsigned w;
        std::memcpy(&w, blob.data() + pos, 4);
        pos += 4;
        std::printf("%s%u", i ? ", " : "", w);
    }
    std::printf("]%s", last ? "" : ", ");
}

template <typename T>
static void runCase(const char *type, bool two, bool var, unsigned long long seed, bool lastCase)
{
    const unsigned nPixels = 6;
    std::string json = std::string("{\"type\": \"") + type + "\", \"two_buffer_variance\": " + (two ? "true" : "false") + ", \"sample_variance\": " + (var ? "true" : "false") + "}";
    JsonDocument doc(Path("kat3.json"), json);
    OutputBufferSettings settings;
    settings.fromJson(doc, *static_cast<const Scene *>(nullptr)); // the scene argument is unused (OutputBufferSettings.cpp:24)
    OutputBuffer<T> buffer(Vec2u(nPixels, 1), settings);
    Gen g{seed};
    const unsigned counts[nPixels] = {0, 1, 2, 7, 24, 40};
    for (unsigned p = 0; p < nPixels; ++p)
        for (unsigned i = 0; i < counts[p]; ++i) buffer.addSample(Vec2u(p, 0), Channels<T>::make(g));
    std::ostringstream *stream = new std::ostringstream();
    OutputStreamHandle out(stream);
    buffer.serialize(out);
    const std::string blob = stream->str();
    size_t pos = 0;
    const size_t n = size_t(nPixels)*Channels<T>::N;
    std::printf("    {\"type\": \"%s\", \"channels\": %d, \"two_buffer_variance\": %s, \"sample_variance\": %s, \"seed\": %llu, ", type, int(Channels<T>::N), two ? "true" : "false",
                var ? "true" : "false", seed);
    printWords("a", blob, pos, n);
    if (two) printWords("b", blob, pos, n);
    if (var) printWords("variance", blob, pos, n);
    printWords("count", blob, pos, nPixels);
    std::printf("\"mean\": [");
    for (unsigned p = 0; p < nPixels; ++p) {
        T m = buffer[p];
        const float *f = reinterpret_cast<const float *>(&m);
        for (int c = 0; c < Channels<T>::N; ++c) std::printf("%s%u", (p || c) ? ", " : "", bits(f[c]));
    }
    std::printf("]}%s\n", lastCase ? "" : ",");
}

int main()
{
    std::printf("{\n  \"output_buffer\": [\n");
    unsigned long long seed = 1;
    for (int two = 0; two < 2; ++two)
        for (int var = 0; var < 2; ++var) {
            runCase<float>("depth", two != 0, var != 0, seed++, false);
            runCase<Vec3f>("normal", two != 0, var != 0, seed++, two == 1 && var == 1);
        }
    std::printf("  ]\n}\n");
    return 0;
}
