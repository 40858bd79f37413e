// Test-infrastructure driver (NOT product code), fourth set.  Compiled against the reference's own sources where they lie under
// /root/reference (no copies; oracle/Makefile.ref), it prints as JSON the transmittance models the fork adds beyond the ones its example
// scene uses (transmittances/{Erlang,Davis,DavisWeinstein}Transmittance.cpp):
//   transmittance  per model: sigmaBar, the four forms (surfaceSurface, surfaceMedium, mediumSurface, mediumMedium) at 48 optical depths
//                  (three different channels each) and 192 draws each of sampleSurface / sampleMedium from a UniformPathSampler(seed)
// tests/test_ref_kat4.py evaluates the oracle's restatement on the same inputs and the same PCG stream.
#include <cstdio>
#include <cstring>
#include <string>
#include <transmittances/DavisTransmittance.hpp>
#include <transmittances/DavisWeinsteinTransmittance.hpp>
#include <transmittances/ErlangTransmittance.hpp>
#include <sampling/UniformPathSampler.hpp>
#include <io/JsonDocument.hpp>

using namespace Tungsten;

static unsigned bits(float f) { unsigned u; std::memcpy(&u, &f, 4); return u; }

static void dump(const char *name, const char *json, Transmittance &t, unsigned seed, bool last)
{
    if (json[0]) {
        JsonDocument doc(Path("kat4.json"), json);
        t.fromJson(doc, *static_cast<const Scene *>(nullptr)); // the scene argument is not used by these models
    }
    std::printf("    {\"name\": \"%s\", \"seed\": %u, \"sigma_bar\": %u,\n     \"forms\": [", name, seed, bits(t.sigmaBar()));
    for (int i = 0; i < 48; ++i) {
        float x = i == 0 ? 0.0f : 0.01f*float(i*i) + 1e-3f*float(i); // 0 .. 22
        Vec3f tau(x, x*0.5f + 0.125f, x*1.75f);
        Vec3f f[4] = {t.surfaceSurface(tau), t.surfaceMedium(tau), t.mediumSurface(tau), t.mediumMedium(tau)};
        for (int k = 0; k < 4; ++k)
            for (int c = 0; c < 3; ++c) std::printf("%s%u", (i || k || c) ? ", " : "", bits(f[k][c]));
    }
    std::printf("],\n     \"sample_surface\": [");
    UniformPathSampler s0(seed);
    for (int i = 0; i < 192; ++i) std::printf("%s%u", i ? ", " : "", bits(t.sampleSurface(s0)));
    std::printf("],\n     \"sample_medium\": [");
    UniformPathSampler s1(seed + 1);
    for (int i = 0; i < 192; ++i) std::printf("%s%u", i ? ", " : "", bits(t.sampleMedium(s1)));
    std::printf("]}%s\n", last ? "" : ",");
}

int main()
{
    std::printf("{\n  \"transmittance\": [\n");
    ErlangTransmittance erlang;
    dump("erlang", "{\"type\": \"erlang\", \"rate\": 2.5}", erlang, 101, false);
    DavisTransmittance davis;
    dump("davis", "{\"type\": \"davis\", \"alpha\": 1.6}", davis, 202, false);
    DavisWeinsteinTransmittance dw;
    dump("davis_weinstein", "{\"type\": \"davis_weinstein\", \"h\": 0.75, \"c\": 1.3}", dw, 303, false);
    DavisWeinsteinTransmittance dw2;
    dump("davis_weinstein_h1", "{\"type\": \"davis_weinstein\", \"h\": 1.0, \"c\": 0.7}", dw2, 404, true);
    // InterpolatedTransmittance.cpp needs Scene::fetchTransmittance, i.e. io/Scene.cpp and with it the whole renderer: not buildable here; it is
    // pinned by the reference-rendered images tests/golden/trans_interpolated*.pfm instead
    std::printf("  ]\n}\n");
    return 0;
}
