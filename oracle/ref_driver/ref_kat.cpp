// Test-infrastructure driver (NOT product code).  Compiled against the reference's own headers and
// sources where they lie under /root/reference (no copies), it prints known-answer vectors of the
// leaf functions on the hot path as JSON: hash32, PCG32, Sobol, normalizedUint, cosineHemisphere,
// Microfacet D/G1/sample/pdf (Beckmann, GGX), conductor/dielectric Fresnel, TangentFrame, powerHeuristic.
#include <math/MathUtil.hpp>
#include <math/BitManip.hpp>
#include <math/TangentFrame.hpp>
#include <sampling/UniformSampler.hpp>
#include <sampling/SampleWarp.hpp>
#include <bsdfs/Microfacet.hpp>
#include <bsdfs/Fresnel.hpp>
#include <sobol/sobol.h>
#include <math/FastMath.hpp>
#include <cstdio>
#include <cstring>

using namespace Tungsten;

static unsigned bits(float f) { unsigned u; std::memcpy(&u, &f, 4); return u; }

int main()
{
    UniformSampler gen(12345u);
    std::printf("{\n");
    std::printf("\"hash32\": [");
    for (unsigned i = 0; i < 32; ++i) std::printf("%s[%u, %u]", i ? ", " : "", i*2654435761u, MathUtil::hash32(i*2654435761u));
    std::printf("],\n\"pcg\": {\"seed\": 3735928559, \"out\": [");
    { UniformSampler s(3735928559u); for (int i = 0; i < 32; ++i) std::printf("%s%u", i ? ", " : "", s.nextI()); }
    std::printf("]},\n\"normalized_uint\": [");
    for (unsigned i = 0; i < 16; ++i) { unsigned v = MathUtil::hash32(i + 7u); std::printf("%s[%u, %u]", i ? ", " : "", v, bits(BitManip::normalizedUint(v))); }
    std::printf("],\n\"sobol\": [");
    for (unsigned i = 0; i < 64; ++i) {
        unsigned idx = MathUtil::hash32(i) & 0xFFFFu, dim = (MathUtil::hash32(i + 99u) % 1024u), scr = MathUtil::hash32(i + 1234u);
        std::printf("%s[%u, %u, %u, %u]", i ? ", " : "", idx, dim, scr, sobol::sample(idx, dim, scr));
    }
    std::printf("],\n\"cosine_hemisphere\": [");
    for (int i = 0; i < 16; ++i) {
        Vec2f xi = gen.next2D();
        Vec3f w = SampleWarp::cosineHemisphere(xi);
        std::printf("%s[%u, %u, %u, %u, %u]", i ? ", " : "", bits(xi.x()), bits(xi.y()), bits(w.x()), bits(w.y()), bits(w.z()));
    }
    std::printf("],\n\"microfacet\": [");
    for (int i = 0; i < 32; ++i) {
        const char *names[2] = {"beckmann", "ggx"};
        Microfacet::Distribution dist(names[i & 1]);
        float alpha = 0.05f + 0.5f*gen.next1D();
        Vec2f xi = gen.next2D();
        Vec3f m = Microfacet::sample(dist, alpha, xi);
        Vec3f v = SampleWarp::cosineHemisphere(gen.next2D());
        std::printf("%s[%d, %u, %u, %u, %u, %u, %u, %u, %u, %u, %u, %u, %u]", i ? ", " : "", i & 1, bits(alpha), bits(xi.x()), bits(xi.y()),
                    bits(m.x()), bits(m.y()), bits(m.z()), bits(Microfacet::D(dist, alpha, m)), bits(Microfacet::pdf(dist, alpha, m)),
                    bits(v.x()), bits(v.y()), bits(v.z()), bits(Microfacet::G1(dist, alpha, v, m)));
    }
    std::printf("],\n\"conductor\": [");
    for (int i = 0; i < 16; ++i) {
        float c = gen.next1D();
        Vec3f F = Fresnel::conductorReflectance(Vec3f(0.2004376970f, 0.9240334304f, 1.1022119527f), Vec3f(3.9129485033f, 2.4528477015f, 2.1421879552f), c);
        std::printf("%s[%u, %u, %u, %u]", i ? ", " : "", bits(c), bits(F.x()), bits(F.y()), bits(F.z()));
    }
    std::printf("],\n\"dielectric\": [");
    for (int i = 0; i < 16; ++i) {
        float c = gen.next1D()*2.0f - 1.0f, ct;
        float eta = (i & 1) ? 1.7f : 1.0f/1.7f;
        float F = Fresnel::dielectricReflectance(eta, c, ct);
        std::printf("%s[%u, %u, %u, %u]", i ? ", " : "", bits(eta), bits(c), bits(F), bits(ct));
    }
    std::printf("],\n\"tangent_frame\": [");
    for (int i = 0; i < 16; ++i) {
        Vec3f n = SampleWarp::uniformSphere(gen.next2D());
        TangentFrame f(n);
        std::printf("%s[%u, %u, %u, %u, %u, %u, %u, %u, %u]", i ? ", " : "", bits(n.x()), bits(n.y()), bits(n.z()), bits(f.tangent.x()), bits(f.tangent.y()),
                    bits(f.tangent.z()), bits(f.bitangent.x()), bits(f.bitangent.y()), bits(f.bitangent.z()));
    }
    // fmath::exp_ps as used by ExponentialTransmittance (math/FastMath.hpp:23-27): the 1024-entry mantissa table and
    // known answers (inputs include the clamped range and values next to table boundaries)
    std::printf("],\n\"fmath_exp_table\": [");
    for (int i = 0; i < 1024; ++i) std::printf("%s%u", i ? ", " : "", fmath::local::C<>::expVar.tbl[i]);
    std::printf("],\n\"fmath_exp\": [");
    for (int i = 0; i < 96; ++i) {
        float x = i < 64 ? -12.0f*gen.next1D() : (i < 80 ? -100.0f*gen.next1D() : 3.0f*gen.next1D());
        Vec3f e = FastMath::exp(Vec3f(x, x*0.5f, -x*0.25f));
        std::printf("%s[%u, %u, %u, %u]", i ? ", " : "", bits(x), bits(e.x()), bits(e.y()), bits(e.z()));
    }
    std::printf("]\n}\n");
    return 0;
}
