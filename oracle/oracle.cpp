// oracle.cpp -- TEST INFRASTRUCTURE, NOT PRODUCT CODE (see oracle.h).
//
// CPU restatement of daseyb/tungsten's path_tracer hot path.  Every function
// cites the reference file:line it follows (paths relative to
// /root/reference/src/core unless noted).  Arithmetic is written in the
// reference's operation order; build with -ffp-contract=off (the reference
// build has no FMA contraction).
#include "oracle.h"
#include "embree_bvh4.h"
#include "../include/tgpu_math.h"
#include "../include/tgpu_bounds.h"

#include <algorithm>
#include <atomic>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <limits>
#include <memory>
#include <thread>
#include <vector>
#if defined(__SSE__)
#include <xmmintrin.h>
#endif

namespace {

// ---- constants (math/Angle.hpp:8-16) ------------------------------------------------
constexpr float PI = 3.1415926536f;
constexpr float TWO_PI = PI*2.0f;
constexpr float INV_PI = 1.0f/PI;
constexpr float INV_TWO_PI = 0.5f*INV_PI;
constexpr float INV_FOUR_PI = 0.25f*INV_PI;
const float INF = std::numeric_limits<float>::infinity();
// Path tracing of ONE sample to stderr (ORACLE_TRACE=1): every vertex with the bits of ray, hit, throughput, radiance and
// sampler state, to be diffed against the GPU's slot dump (TGPU_TRACE_SLOTS=1); tests/tools/trace_path.py drives both.
static bool g_trace = std::getenv("ORACLE_TRACE") != nullptr;
// ORACLE_TRACE_AT="pixelIndex:sample": oracle_render_image traces only that sample (run with one thread); ORACLE_DUMP_TILE=t: prints the
// radiance bits of every sample of tile t in rendering order ("S pixelIndex sample r g b"), to be diffed against tests/tools/ref_path_dump.cpp
static const char *g_traceAt = std::getenv("ORACLE_TRACE_AT");
static const char *g_dumpTile = std::getenv("ORACLE_DUMP_TILE");
// ORACLE_BRUTE=1: triangle queries test EVERY triangle (no BVH pruning) -- the definition the BVH walk must reproduce:
// lexicographic minimum of (t, triangle id) over all triangles whose test passes.  Minutes per image; for single-path diagnosis.
static const bool g_brute = std::getenv("ORACLE_BRUTE") != nullptr;
#define TRC(...) do { if (g_trace) std::fprintf(stderr, __VA_ARGS__); } while (0)

// ---- Vec3f (math/Vec.hpp:127-193) ---------------------------------------------------
struct V3 {
    float x, y, z;
    V3() : x(0), y(0), z(0) {}
    explicit V3(float a) : x(a), y(a), z(a) {}
    V3(float a, float b, float c) : x(a), y(b), z(c) {}
    explicit V3(const float *p) : x(p[0]), y(p[1]), z(p[2]) {}
    float operator[](int i) const { return (&x)[i]; }
    float &operator[](int i) { return (&x)[i]; }
    V3 operator+(const V3 &o) const { return V3(x + o.x, y + o.y, z + o.z); }
    V3 operator-(const V3 &o) const { return V3(x - o.x, y - o.y, z - o.z); }
    V3 operator*(const V3 &o) const { return V3(x*o.x, y*o.y, z*o.z); }
    V3 operator/(const V3 &o) const { return V3(x/o.x, y/o.y, z/o.z); }
    V3 operator*(float s) const { return V3(x*s, y*s, z*s); }
    V3 operator/(float s) const { return V3(x/s, y/s, z/s); }
    V3 operator+(float s) const { return V3(x + s, y + s, z + s); }
    V3 operator-(float s) const { return V3(x - s, y - s, z - s); }
    V3 operator-() const { return V3(-x, -y, -z); }
    V3 &operator+=(const V3 &o) { x += o.x; y += o.y; z += o.z; return *this; }
    V3 &operator*=(const V3 &o) { x *= o.x; y *= o.y; z *= o.z; return *this; }
    V3 &operator*=(float s) { x *= s; y *= s; z *= s; return *this; }
    V3 &operator/=(float s) { x /= s; y /= s; z /= s; return *this; }
    bool operator==(float s) const { return x == s && y == s && z == s; }
    bool operator!=(float s) const { return x != s || y != s || z != s; }
    float dot(const V3 &o) const { float s = x*o.x; s += y*o.y; s += z*o.z; return s; }
    float lengthSq() const { float r = 0.0f; r += x*x; r += y*y; r += z*z; return r; }
    float length() const { return std::sqrt(lengthSq()); }
    V3 normalized() const { float inv = 1.0f/length(); return V3(x*inv, y*inv, z*inv); }
    V3 cross(const V3 &o) const { return V3(y*o.z - z*o.y, z*o.x - x*o.z, x*o.y - y*o.x); }
    float sum() const { float r = x; r += y; r += z; return r; }
    float avg() const { return sum()*(1.0f/3.0f); }
    float max() const { float m = x; if (y > m) m = y; if (z > m) m = z; return m; }
    float luminance() const { return x*0.2126f + y*0.7152f + z*0.0722f; }
};
inline V3 operator*(float s, const V3 &v) { return V3(v.x*s, v.y*s, v.z*s); }
inline uint32_t hx(float f) { return tgpu_float_to_bits(f); }
#define HX3(v) hx((v).x), hx((v).y), hx((v).z)
inline V3 vabs(const V3 &v) { return V3(std::abs(v.x), std::abs(v.y), std::abs(v.z)); }
inline float fmin2(float a, float b) { return a < b ? a : b; } // Tungsten::min (MathUtil.hpp)
inline float fmax2(float a, float b) { return a > b ? a : b; }
inline float clampf(float v, float lo, float hi) { return fmin2(fmax2(v, lo), hi); } // clamp = min(max(val,minVal),maxVal)
inline bool isnanV(const V3 &v) { return std::isnan(v.x) || std::isnan(v.y) || std::isnan(v.z); }
inline bool isinfV(const V3 &v) { return std::isinf(v.x) || std::isinf(v.y) || std::isinf(v.z); }

// Mat4f*Vec3f with zero translation column (math/Mat4f.hpp:320-327): the "+ 0" is kept.
inline V3 mulPoint0(const float *m, const V3 &b)
{
    return V3(m[0]*b.x + m[1]*b.y + m[2]*b.z + 0.0f, m[3]*b.x + m[4]*b.y + m[5]*b.z + 0.0f, m[6]*b.x + m[7]*b.y + m[8]*b.z + 0.0f);
}
inline V3 qrotate(const float *q, const V3 &o) // Quaternion::operator*(Vec3) (math/Quaternion.hpp:74-84), q = w x y z
{
    float tx = 2.0f*(q[2]*o.z - q[3]*o.y);
    float ty = 2.0f*(q[3]*o.x - q[1]*o.z);
    float tz = 2.0f*(q[1]*o.y - q[2]*o.x);
    return V3(o.x + q[0]*tx + q[2]*tz - q[3]*ty,
              o.y + q[0]*ty + q[3]*tx - q[1]*tz,
              o.z + q[0]*tz + q[1]*ty - q[2]*tx);
}
inline V3 mulVector(const float *m, const V3 &b) // Mat4f::transformVector (Mat4f.hpp:159-166)
{
    return V3(m[0]*b.x + m[1]*b.y + m[2]*b.z, m[3]*b.x + m[4]*b.y + m[5]*b.z, m[6]*b.x + m[7]*b.y + m[8]*b.z);
}

// ---- BsdfLobes (bsdfs/BsdfLobes.hpp) --------------------------------------------------
enum : uint32_t {
    GlossyReflectionLobe = 1u << 0, GlossyTransmissionLobe = 1u << 1, DiffuseReflectionLobe = 1u << 2,
    DiffuseTransmissionLobe = 1u << 3, SpecularReflectionLobe = 1u << 4, SpecularTransmissionLobe = 1u << 5,
    AnisotropicLobe = 1u << 6, ForwardLobe = 1u << 7,
    SpecularLobe = SpecularReflectionLobe | SpecularTransmissionLobe,
    TransmissiveLobe = GlossyTransmissionLobe | DiffuseTransmissionLobe | SpecularTransmissionLobe,
    ReflectiveLobe = GlossyReflectionLobe | DiffuseReflectionLobe | SpecularReflectionLobe,
    AllLobes = TransmissiveLobe | ReflectiveLobe | AnisotropicLobe,
    AllButSpecular = ~(SpecularLobe | ForwardLobe),
};

// ---- sampler --------------------------------------------------------------------------
inline uint32_t hash32(uint32_t x) // math/MathUtil.hpp:120-128
{
    x = ~x + (x << 15);
    x = x ^ (x >> 12);
    x = x + (x << 2);
    x = x ^ (x >> 4);
    x = x*2057;
    x = x ^ (x >> 16);
    return x;
}
inline float normalizedUint(uint32_t i) // math/BitManip.hpp:47-50
{
    return tgpu_bits_to_float((i >> 9u) | 0x3F800000u) - 1.0f;
}
struct Pcg32 { // sampling/UniformSampler.hpp:22-52
    uint64_t state = 0xBA5EBA11ull, sequence = 0;
    Pcg32() {}
    explicit Pcg32(uint64_t seed, uint64_t seq = 0) : state(seed), sequence(seq) {}
    uint32_t nextI()
    {
        uint64_t oldState = state;
        state = oldState*6364136223846793005ULL + (sequence | 1);
        uint32_t xorShifted = uint32_t(((oldState >> 18u) ^ oldState) >> 27u);
        uint32_t rot = uint32_t(oldState >> 59u);
        return (xorShifted >> rot) | (xorShifted << (uint32_t(-int32_t(rot)) & 31));
    }
    float next1D() { return normalizedUint(nextI()); }
};

struct Sampler { // sampling/SobolPathSampler.hpp:20-84 and sampling/UniformPathSampler.hpp
    const uint32_t *matrices = nullptr;
    bool sobol = true;
    Pcg32 *supplemental = nullptr; // the tile's stream (rng 0) or the path's own (rng 1)
    Pcg32 local;
    uint32_t seed = 0, scramble = 0, index = 0, dimension = 0;

    uint32_t permutedIndex() const { return (index & ~0xFFu) | ((index + scramble) & 0xFFu); }
    void startPath(uint32_t pixelId, uint32_t sample)
    {
        scramble = seed ^ hash32(pixelId);
        index = sample;
        dimension = 0;
    }
    bool nextBoolean(float pTrue) { bool r = supplemental->next1D() < pTrue; TRC("    rnd bool(%08x)=%d\n", hx(pTrue), int(r)); return r; }
    int nextDiscrete(int numChoices) { int r = int(supplemental->next1D()*float(numChoices)); TRC("    rnd disc(%d)=%d\n", numChoices, r); return r; }
    float next1D() { float r = next1Dq(); TRC("    rnd 1d=%08x\n", hx(r)); return r; }
    float next1Dq()
    {
        if (!sobol || dimension >= 1024) return supplemental->next1D();
        // thirdparty/sobol/sobol.h:40-53 (only the first 32 columns are reachable with a 32-bit index)
        uint32_t result = scramble;
        uint32_t idx = permutedIndex();
        for (uint32_t i = dimension*32u; idx; idx >>= 1, ++i)
            if (idx & 1) result ^= matrices[i];
        dimension++;
        return normalizedUint(result);
    }
};

// path-local supplemental stream: pcg32_srandom(seed = scramble:sampleIndex, inc = 1)
inline Pcg32 pathLocalPcg(uint32_t scramble, uint32_t sampleIndex)
{
    Pcg32 p(0, 0);
    p.nextI();
    p.state += (uint64_t(scramble) << 32) | uint64_t(sampleIndex);
    p.nextI();
    return p;
}

// ---- records ---------------------------------------------------------------------------
struct Ray { V3 pos, dir; float nearT, farT; };
enum PrimKind : int { PRIM_NONE = 0, PRIM_TRI = 1, PRIM_QUAD = 2, PRIM_CUBE = 3, PRIM_ENV = 4, PRIM_INSTANCE = 5, PRIM_SPHERE = 6, PRIM_DISK = 7 };
struct Temp { // IntersectionTemporary
    int kind = PRIM_NONE;
    uint32_t index = 0;
    float u = 0, v = 0;
    V3 p, w, Ng; // quad / disk hit point, env origin ; env direction ; mesh unnormalised geometric normal
    float rSq = 0; // DiskIntersection::rSq
    bool backSide = false;
    int32_t inst = -1, instSet = -1; // IntersectionTemporary::flags / primitive == an Instance (Instance.cpp:305-308)
};
struct Info { // primitives/IntersectionInfo.hpp
    V3 Ng, Ns, p, w;
    float uvx = 0, uvy = 0;
    float epsilon = 0;
    int kind = PRIM_NONE;
    uint32_t index = 0;
    int bsdf = -1;
};
struct Frame { V3 normal, tangent, bitangent; // math/TangentFrame.hpp:22-45
    V3 toLocal(const V3 &p) const { return V3(tangent.dot(p), bitangent.dot(p), normal.dot(p)); }
    V3 toGlobal(const V3 &p) const { return tangent*p.x + bitangent*p.y + normal*p.z; }
};
inline Frame duffFrame(const V3 &n)
{
    Frame f;
    f.normal = n;
    float sign = copysignf(1.0f, n.z);
    const float a = -1.0f/(sign + n.z);
    const float b = n.x*n.y*a;
    f.tangent = V3(1.0f + sign*n.x*n.x*a, sign*b, -sign*n.x);
    f.bitangent = V3(b, sign + n.y*n.y*a, -n.y);
    return f;
}
struct Event { // samplerecords/SurfaceScatterEvent.hpp:14-66
    const Info *info;
    Sampler *sampler;
    Frame frame;
    V3 wi, wo, weight;
    float pdf;
    uint32_t requestedLobe, sampledLobe;
    bool flippedFrame;
};
struct LightSample { V3 d; float dist, pdf; };
} // namespace
// media/Medium.hpp:30-47, samplerecords/MediumSample.hpp
// What one sample contributes to the auxiliary output buffers (PathTracer.cpp:78-96,133-140).  NaN = no sample for that buffer
// (OutputBuffer::addSample drops NaN and infinite values anyway, cameras/OutputBuffer.hpp:106-107).
struct AovSample {
    float depth = std::numeric_limits<float>::quiet_NaN();
    V3 normal = V3(std::numeric_limits<float>::quiet_NaN());
    V3 albedo = V3(std::numeric_limits<float>::quiet_NaN());
    float visibility = std::numeric_limits<float>::quiet_NaN();
};
struct MediumStateO { bool firstScatter; int bounce; void reset() { firstScatter = true; bounce = 0; } void advance() { firstScatter = false; bounce++; } };
struct MediumSampleO { V3 p; float t; V3 weight, emission; float pdf; bool exited; };
namespace {

struct BvhNode { float lo[3], hi[3]; int32_t left, right; uint32_t first, count; };
struct TriIsect { V3 v0, e1, e2, Ng; };
struct TriBvh { std::vector<BvhNode> nodes; std::vector<uint32_t> triOrder; }; // triOrder holds GLOBAL triangle ids

} // namespace

struct oracle_instances;
// ========================================================================================
struct oracle_ctx {
    tgpu_scene_desc sc;
    int rngMode, mathMode, rcpMode;
    bool brute = g_brute; // oracle_set_brute_force: triangle queries test every triangle (the definition the BVH walks must meet)
    float hitPad = 0.0f;             // GPU mode: TGPU_HIT_PAD_FRACTION * tgpu_bvh_pad (include/tgpu_bounds.h (3)); unused in reference mode
    TriBvh sceneBvh;                 // scene finites: triangles [0, n_tris)
    std::vector<TriIsect> tris;      // all triangles incl. the instance masters' (n_tris + n_master_tris)
    std::vector<Pcg32> tilePcg;
    uint32_t tileSeedBase = 0;
    bool tilesInit = false;
    std::atomic<uint64_t> stSamples{0}, stExtend{0}, stShadow{0}, stSurface{0}, stNodes{0}, stTris{0};

    // ---- libm dispatch -----------------------------------------------------------------
    float msin(float x) const { return mathMode ? tgpu_sinf(x) : std::sin(x); }
    float mcos(float x) const { return mathMode ? tgpu_cosf(x) : std::cos(x); }
    float mexp(float x) const { return mathMode ? tgpu_expf(x) : std::exp(x); }
    float mlog(float x) const { return mathMode ? tgpu_logf(x) : std::log(x); }
    float matan2(float y, float x) const { return mathMode ? tgpu_atan2f(y, x) : std::atan2(y, x); }
    float macos(float x) const { return mathMode ? tgpu_acosf(x) : std::acos(x); }
    float mtan(float x) const { return mathMode ? tgpu_tanf(x) : std::tan(x); }
    float mpow(float x, float y) const { return mathMode ? tgpu_powf(x, y) : std::pow(x, y); }

    struct Counters { uint64_t extend = 0, shadow = 0, surface = 0, nodes = 0, tris = 0, samples = 0; };
    oracle_outbuf out[TGPU_OUTPUT_COUNT] = {}; // output buffers bound by the test (oracle_bind_output_buffers)
    bool trackOutputs = false;

    // ---- BVH ---------------------------------------------------------------------------
    std::unique_ptr<oracle_instances> inst;
    void buildInstances();
    bool masterIntersect(uint32_t setIndex, uint32_t master, Ray &ray, Temp &data, Counters &c) const;
    bool instanceIntersect(uint32_t setIndex, Ray &ray, Temp &data, Counters &c) const;
    void prepareTris();
    void buildBvh(uint32_t firstTri, uint32_t nTris, TriBvh &out) const;
    void intersectTris(const TriBvh &bvh, const Ray &ray, float &bestT, float &bestU, float &bestV, uint32_t &bestPrim, Counters &c) const;
    bool occludedTris(const TriBvh &bvh, const Ray &ray, Counters &c) const;

    // ---- textures ----------------------------------------------------------------------
    V3 bitmapEval(const tgpu_bitmap &bm, float ux, float uy) const;
    V3 texEval(int tex, float ux, float uy) const;

    // ---- primitives / lights -----------------------------------------------------------
    bool quadIntersect(const tgpu_quad &q, uint32_t idx, Ray &ray, Temp &data) const;
    bool cubeIntersect(const tgpu_cube &cb, uint32_t idx, Ray &ray, Temp &data) const;
    bool sphereIntersect(const tgpu_sphere &sp, uint32_t idx, Ray &ray, Temp &data) const;
    bool diskIntersect(const tgpu_disk &dk, uint32_t idx, Ray &ray, Temp &data) const;
    void intersectionInfo(const Temp &data, Info &info) const;
    bool intersect(Ray &ray, Temp &data, Info &info, Counters &c) const;
    void directionToUV(const V3 &wi, float &ux, float &uy, float *sinTheta) const;
    V3 uvToDirection(float ux, float uy, float &sinTheta) const;
    float envPdf(float ux, float uy) const;

    bool lightSampleDirect(const tgpu_light &l, const V3 &p, Sampler &s, LightSample &out) const;
    bool lightIntersect(const tgpu_light &l, Ray &ray, Temp &data) const;
    float lightDirectPdf(const tgpu_light &l, const Temp &data, const Info &info, const V3 &p) const;
    V3 primEvalDirect(const Temp &data, const Info &info) const;
    bool primEmissive(int kind, uint32_t index) const;
    bool primSamplable(int kind, uint32_t index) const;
    float lightApproximateRadiance(const tgpu_light &l, const V3 &p) const;

    // ---- BSDFs -------------------------------------------------------------------------
    uint32_t bsdfLobes(int b) const;
    V3 bsdfEval(int b, const Event &e) const;
    bool bsdfSample(int b, Event &e) const;
    float bsdfPdf(int b, const Event &e) const;

    // ---- integrator --------------------------------------------------------------------
    Event makeLocalScatterEvent(const Temp &data, const Info &info, const Ray &ray, Sampler *sampler) const;
    V3 shadowRay(Ray &ray, int medium, int endKind, uint32_t endIndex, int bounce, bool startsOnSurface, bool endsOnSurface, Counters &c) const;
    V3 attenuatedEmission(const tgpu_light &l, int medium, float expectedDist, Temp &data, Info &info, int bounce, bool startsOnSurface, Ray &ray, Counters &c,
                          V3 *transmittance = nullptr) const;
    V3 lightSample(const tgpu_light &l, Event &event, int medium, int bounce, const Ray &parent, Counters &c, V3 *transmittance = nullptr) const;
    V3 bsdfSampleNee(const tgpu_light &l, Event &event, int medium, int bounce, const Ray &parent, Counters &c) const;
    V3 estimateDirect(Event &event, int medium, int bounce, const Ray &parent, Counters &c, V3 *transmittance = nullptr) const;
    const tgpu_light *chooseLight(const V3 &p, Sampler &sampler, float &weight) const;
    // RoughDielectricBsdf::sampleBase/evalBase/pdfBase (static helpers shared with RoughPlasticBsdf)
    bool rdSampleBase(Event &e, bool sampleR, bool sampleT, float roughness, float ior, int distribution) const;
    V3 rdEvalBase(const Event &e, bool sampleR, bool sampleT, float roughness, float ior, int distribution) const;
    float rdPdfBase(const Event &e, bool sampleR, bool sampleT, float roughness, float ior, int distribution) const;
    // media (media/HomogeneousMedium.cpp, transmittances/ExponentialTransmittance.cpp, phasefunctions/IsotropicPhaseFunction.cpp)
    V3 fastExp(const V3 &x) const;
    bool mediumSampleDistance(int medium, Sampler &sampler, const Ray &ray, struct MediumStateO &state, struct MediumSampleO &sample) const;
    V3 mediumTransmittance(int medium, const Ray &ray, bool startOnSurface, bool endOnSurface) const;
    V3 transEval(const tgpu_medium &m, const V3 &tau, bool startOnSurface, bool endOnSurface) const;
    V3 transForm(const tgpu_medium &m, int form, const V3 &tau) const; // 0 surfaceSurface 1 surfaceMedium 2 mediumSurface 3 mediumMedium
    float transSigmaBar(const tgpu_medium &m) const;
    float transSample(const tgpu_medium &m, Sampler &sampler, bool startOnSurface) const;
    // one model (type + parameters); the medium-level functions above add InterpolatedTransmittance on top
    V3 modelForm(int type, const float *param, int form, const V3 &tau) const;
    float modelSigmaBar(int type, const float *param) const;
    float modelSample(int type, const float *param, Sampler &sampler, bool startOnSurface) const;
    int selectMedium(const Info &info, int current, bool geometricBackside) const;
    float phaseEval(const tgpu_medium &m, const V3 &wi, const V3 &wo) const;
    V3 phaseSample(const tgpu_medium &m, Sampler &sampler, const V3 &wi, float &pdf) const;
    V3 volumeEstimateDirect(Sampler &sampler, const struct MediumSampleO &ms, int medium, int bounce, const Ray &parent, Counters &c) const;
    bool handleVolume(Sampler &sampler, const struct MediumSampleO &ms, int medium, int bounce, bool enableLightSampling, Ray &ray, V3 &throughput, V3 &emission,
                      bool &wasSpecular, Counters &c) const;
    uint32_t expTable[1024];
    float expA, expB;
    void buildExpTable();
    // reference mode (rcp_mode = 1): the reference's two levels of Embree BVH4, restated (oracle/embree_bvh4.h)
    struct RefFinite { int kind; uint32_t index; };
    std::vector<RefFinite> refFinites;                 // the primitives of TraceableScene's user geometry
    embree_restated::Bvh4 refTop;                      // BVH4 over their Primitive::bounds()
    std::vector<embree_restated::Bvh4> refMesh;        // per mesh: BVH4 with Triangle4 leaves, primitive ids local to the mesh
    bool refAccel = false;
    void buildReferenceAccel();
    bool refMeshIntersect(uint32_t mesh, Ray &ray, Temp &data, Counters &c) const;
    void refIntersect(Ray &ray, Temp &data, Counters &c) const;
    bool handleSurface(Event &event, Temp &data, Info &info, int &medium, int bounce, bool enableLightSampling, Ray &ray,
                       V3 &throughput, V3 &emission, bool &wasSpecular, struct MediumStateO &state, Counters &c, V3 *transmittance = nullptr) const;
    V3 traceSample(uint32_t px, uint32_t py, Sampler &sampler, Counters &c, struct AovSample *aov = nullptr) const;
    void cameraRay(uint32_t px, uint32_t py, float xi0, float xi1, Ray &ray) const;
};

// ========================================================================================
// BVH over the triangle soup (own builder: binned SAH, leaves <= 4; the reference's
// Embree BVH4 is replaced, nearest-hit results do not depend on the tree)
// ========================================================================================
void oracle_ctx::prepareTris()
{
    uint32_t n = sc.n_tris + sc.n_master_tris;
    tris.resize(n);
    for (uint32_t i = 0; i < n; ++i) {
        V3 p0(sc.tri_pos + size_t(i)*9), p1(sc.tri_pos + size_t(i)*9 + 3), p2(sc.tri_pos + size_t(i)*9 + 6);
        // Embree TriangleM: v0, e1 = v0 - v1, e2 = v2 - v0, Ng = cross(e1, e2)
        // (thirdparty/embree/kernels/geometry/triangle.h:53-54, triangle_intersector_moeller.h:118-120)
        tris[i].v0 = p0;
        tris[i].e1 = p0 - p1;
        tris[i].e2 = p2 - p0;
        tris[i].Ng = tris[i].e1.cross(tris[i].e2);
    }
}

void oracle_ctx::buildBvh(uint32_t firstTri, uint32_t n, TriBvh &out) const
{
    std::vector<BvhNode> &nodes = out.nodes;
    std::vector<uint32_t> &triOrder = out.triOrder;
    triOrder.resize(n);
    const float triPad = tgpu_bvh_pad(&sc);
    // per-triangle boxes, indexed by id - firstTri
    std::vector<float> cminL(size_t(n)*3), cmaxL(size_t(n)*3), cenL(size_t(n)*3);
    for (uint32_t i = 0; i < n; ++i) {
        triOrder[i] = firstTri + i;
        const float *tp = sc.tri_pos + size_t(firstTri + i)*9;
        V3 p0(tp), p1(tp + 3), p2(tp + 6);
        for (int k = 0; k < 3; ++k) {
            cminL[size_t(i)*3 + k] = std::min(p0[k], std::min(p1[k], p2[k]));
            cmaxL[size_t(i)*3 + k] = std::max(p0[k], std::max(p1[k], p2[k]));
            cenL[size_t(i)*3 + k] = 0.5f*(cminL[size_t(i)*3 + k] + cmaxL[size_t(i)*3 + k]);
            // include/tgpu_bounds.h: the binary32 triangle test can pass for a ray that misses the exact box by a few ulps
            cminL[size_t(i)*3 + k] -= triPad;
            cmaxL[size_t(i)*3 + k] += triPad;
        }
    }
    // views addressed by global id
    const float *cmin = cminL.data() - size_t(firstTri)*3, *cmax = cmaxL.data() - size_t(firstTri)*3, *cen = cenL.data() - size_t(firstTri)*3;
    nodes.clear();
    if (n == 0) return;
    nodes.reserve(size_t(n));
    struct Task { uint32_t node, first, count; };
    std::vector<Task> stack;
    nodes.push_back(BvhNode{});
    stack.push_back({0, 0, n});
    while (!stack.empty()) {
        Task t = stack.back();
        stack.pop_back();
        float lo[3] = {INF, INF, INF}, hi[3] = {-INF, -INF, -INF}, clo[3] = {INF, INF, INF}, chi[3] = {-INF, -INF, -INF};
        for (uint32_t i = t.first; i < t.first + t.count; ++i) {
            uint32_t id = triOrder[i];
            for (int k = 0; k < 3; ++k) {
                lo[k] = std::min(lo[k], cmin[size_t(id)*3 + k]);
                hi[k] = std::max(hi[k], cmax[size_t(id)*3 + k]);
                clo[k] = std::min(clo[k], cen[size_t(id)*3 + k]);
                chi[k] = std::max(chi[k], cen[size_t(id)*3 + k]);
            }
        }
        BvhNode nd{};
        for (int k = 0; k < 3; ++k) { nd.lo[k] = lo[k]; nd.hi[k] = hi[k]; }
        nd.left = nd.right = -1;
        nd.first = t.first;
        nd.count = t.count;
        uint32_t mid = 0;
        bool split = false;
        if (t.count > 4) {
            const int B = 16;
            float bestCost = INF;
            int bestAxis = -1, bestBin = -1;
            for (int ax = 0; ax < 3; ++ax) {
                float ext = chi[ax] - clo[ax];
                if (!(ext > 0.0f)) continue;
                uint32_t cnt[B] = {0};
                float blo[B][3], bhi[B][3];
                for (int b = 0; b < B; ++b)
                    for (int k = 0; k < 3; ++k) { blo[b][k] = INF; bhi[b][k] = -INF; }
                for (uint32_t i = t.first; i < t.first + t.count; ++i) {
                    uint32_t id = triOrder[i];
                    int b = std::min(B - 1, int((cen[size_t(id)*3 + ax] - clo[ax])/ext*B));
                    cnt[b]++;
                    for (int k = 0; k < 3; ++k) {
                        blo[b][k] = std::min(blo[b][k], cmin[size_t(id)*3 + k]);
                        bhi[b][k] = std::max(bhi[b][k], cmax[size_t(id)*3 + k]);
                    }
                }
                float rArea[B];
                uint32_t rCnt[B];
                float alo[3] = {INF, INF, INF}, ahi[3] = {-INF, -INF, -INF};
                uint32_t acc = 0;
                for (int b = B - 1; b > 0; --b) {
                    for (int k = 0; k < 3; ++k) { alo[k] = std::min(alo[k], blo[b][k]); ahi[k] = std::max(ahi[k], bhi[b][k]); }
                    acc += cnt[b];
                    float dx = ahi[0] - alo[0], dy = ahi[1] - alo[1], dz = ahi[2] - alo[2];
                    rArea[b] = acc ? dx*dy + dy*dz + dz*dx : 0.0f;
                    rCnt[b] = acc;
                }
                for (int k = 0; k < 3; ++k) { alo[k] = INF; ahi[k] = -INF; }
                acc = 0;
                for (int b = 0; b < B - 1; ++b) {
                    for (int k = 0; k < 3; ++k) { alo[k] = std::min(alo[k], blo[b][k]); ahi[k] = std::max(ahi[k], bhi[b][k]); }
                    acc += cnt[b];
                    if (acc == 0 || rCnt[b + 1] == 0) continue;
                    float dx = ahi[0] - alo[0], dy = ahi[1] - alo[1], dz = ahi[2] - alo[2];
                    float cost = (dx*dy + dy*dz + dz*dx)*acc + rArea[b + 1]*rCnt[b + 1];
                    if (cost < bestCost) { bestCost = cost; bestAxis = ax; bestBin = b; }
                }
            }
            if (bestAxis >= 0) {
                float ext = chi[bestAxis] - clo[bestAxis];
                auto it = std::partition(triOrder.begin() + t.first, triOrder.begin() + t.first + t.count, [&](uint32_t id) {
                    int b = std::min(B - 1, int((cen[size_t(id)*3 + bestAxis] - clo[bestAxis])/ext*B));
                    return b <= bestBin;
                });
                mid = uint32_t(it - triOrder.begin());
                split = mid > t.first && mid < t.first + t.count;
            }
            if (!split) { // degenerate: median split by index
                mid = t.first + t.count/2;
                split = true;
            }
        }
        if (split) {
            nd.left = int32_t(nodes.size());
            nd.right = nd.left + 1;
            nd.count = 0;
            nodes.push_back(BvhNode{});
            nodes.push_back(BvhNode{});
            stack.push_back({uint32_t(nd.left), t.first, mid - t.first});
            stack.push_back({uint32_t(nd.right), mid, t.first + t.count - mid});
        }
        nodes[t.node] = nd;
    }
}

static inline bool slab(const BvhNode &n, const V3 &o, const V3 &inv, float tnear, float tfar)
{
    // conservative slab test: a ray parallel to a slab only has to start inside it; the ray interval carries the relative
    // slack of include/tgpu_bounds.h (a triangle's computed t may lie a few ulps outside its box's [entry, exit])
    float t0 = tnear*(1.0f - TGPU_BVH_T_SLACK), t1 = tfar*(1.0f + TGPU_BVH_T_SLACK);
    for (int k = 0; k < 3; ++k) {
        if (std::isinf(inv[k])) {
            if (o[k] < n.lo[k] || o[k] > n.hi[k]) return false;
            continue;
        }
        float a = (n.lo[k] - o[k])*inv[k], b = (n.hi[k] - o[k])*inv[k];
        float mn = a < b ? a : b, mx = a < b ? b : a;
        if (mn > t0) t0 = mn;
        if (mx < t1) t1 = mx;
    }
    return t0 <= t1*1.0000004f;
}

namespace {
struct TriHit { bool valid; float t, u, v; };
}

// Embree MoellerTrumboreIntersector1 (thirdparty/embree/kernels/geometry/triangle_intersector_moeller.h:76-113,
// finalize :43-49; dot/cross order common/math/vec3.h:182,187)
// hitPad (GPU mode only, include/tgpu_bounds.h (3)): the reported point must lie inside the triangle's own box grown by that much
static inline TriHit triTest(const TriIsect &tri, const Ray &ray, float tfar0, int rcpMode, float hitPad = 0.0f)
{
    auto edot = [](const V3 &a, const V3 &b) { return a.x*b.x + (a.y*b.y + a.z*b.z); };
    TriHit h{false, 0, 0, 0};
    const V3 &O = ray.pos, &D = ray.dir;
    V3 C = tri.v0 - O;
    V3 R = D.cross(C);
    float den = edot(tri.Ng, D);
    float absDen = std::abs(den);
    uint32_t sgn = tgpu_float_to_bits(den) & 0x80000000u;
    float U = tgpu_bits_to_float(tgpu_float_to_bits(edot(R, tri.e2)) ^ sgn);
    float V = tgpu_bits_to_float(tgpu_float_to_bits(edot(R, tri.e1)) ^ sgn);
    if (!(den != 0.0f && U >= 0.0f && V >= 0.0f && U + V <= absDen)) return h;
    float T = tgpu_bits_to_float(tgpu_float_to_bits(edot(tri.Ng, C)) ^ sgn);
    if (!(T > absDen*ray.nearT && T < absDen*tfar0)) return h;
    h.valid = true;
#if defined(__SSE__)
    if (rcpMode) {
        float r = _mm_cvtss_f32(_mm_rcp_ss(_mm_set_ss(absDen)));
        float rcp = r*(2.0f - r*absDen);
        h.t = T*rcp; h.u = U*rcp; h.v = V*rcp;
        return h;
    }
#endif
    (void)rcpMode;
    h.t = T/absDen; h.u = U/absDen; h.v = V/absDen;
    if (!(absDen*absDen >= (TGPU_HIT_SURE_COS*TGPU_HIT_SURE_COS)*(edot(tri.Ng, tri.Ng)*edot(D, D)))) { // grazing: the box test decides
        const V3 P(O.x + D.x*h.t, O.y + D.y*h.t, O.z + D.z*h.t), v1 = tri.v0 - tri.e1, v2 = tri.v0 + tri.e2;
        for (int k = 0; k < 3; ++k) {
            const float lo = fmin2(tri.v0[k], fmin2(v1[k], v2[k])) - hitPad, hi = fmax2(tri.v0[k], fmax2(v1[k], v2[k])) + hitPad;
            if (!(P[k] >= lo && P[k] <= hi)) { h.valid = false; return h; }
        }
    }
    return h;
}

// nearest hit = lexicographic minimum of (t, triangle id) over all valid triangles
void oracle_ctx::intersectTris(const TriBvh &bvh, const Ray &ray, float &bestT, float &bestU, float &bestV, uint32_t &bestPrim, Counters &c) const
{
    const std::vector<BvhNode> &nodes = bvh.nodes;
    const std::vector<uint32_t> &triOrder = bvh.triOrder;
    if (nodes.empty()) return;
    const float tfar0 = bestT;
    V3 inv(1.0f/ray.dir.x, 1.0f/ray.dir.y, 1.0f/ray.dir.z);
    int stack[128];
    int sp = 0;
    stack[sp++] = 0;
    while (sp) {
        const BvhNode &n = nodes[size_t(stack[--sp])];
        c.nodes++;
        if (!brute && !slab(n, ray.pos, inv, ray.nearT, bestT)) continue;
        if (n.left < 0) {
            for (uint32_t i = n.first; i < n.first + n.count; ++i) {
                uint32_t id = triOrder[i];
                c.tris++;
                TriHit h = triTest(tris[id], ray, tfar0, rcpMode, hitPad);
                if (h.valid) TRC("    valid tri=%u t=%08x u=%08x v=%08x\n", id, hx(h.t), hx(h.u), hx(h.v));
                if (h.valid && (h.t < bestT || (h.t == bestT && id < bestPrim))) {
                    bestT = h.t; bestU = h.u; bestV = h.v; bestPrim = id;
                }
            }
        } else {
            if (sp + 2 > 128) continue; // cannot happen for depth < 64
            stack[sp++] = n.left;
            stack[sp++] = n.right;
        }
    }
}

bool oracle_ctx::occludedTris(const TriBvh &bvh, const Ray &ray, Counters &c) const
{
    const std::vector<BvhNode> &nodes = bvh.nodes;
    const std::vector<uint32_t> &triOrder = bvh.triOrder;
    if (nodes.empty()) return false;
    V3 inv(1.0f/ray.dir.x, 1.0f/ray.dir.y, 1.0f/ray.dir.z);
    int stack[128];
    int sp = 0;
    stack[sp++] = 0;
    while (sp) {
        const BvhNode &n = nodes[size_t(stack[--sp])];
        c.nodes++;
        if (!brute && !slab(n, ray.pos, inv, ray.nearT, ray.farT)) continue;
        if (n.left < 0) {
            for (uint32_t i = n.first; i < n.first + n.count; ++i) {
                c.tris++;
                if (triTest(tris[triOrder[i]], ray, ray.farT, rcpMode, hitPad).valid) return true;
            }
        } else {
            if (sp + 2 > 128) continue;
            stack[sp++] = n.left;
            stack[sp++] = n.right;
        }
    }
    return false;
}

// ========================================================================================
// textures
// ========================================================================================
V3 oracle_ctx::bitmapEval(const tgpu_bitmap &bm, float ux, float uy) const // textures/BitmapTexture.cpp:298-352
{
    int w = bm.w, h = bm.h;
    float u = ux*w;
    float v = (1.0f - uy)*h;
    bool linear = bm.linear != 0;
    if (linear) {
        u -= 0.5f;
        v -= 0.5f;
    }
    int iu0 = u < 0.0f ? -int(-u) - 1 : int(u);
    int iv0 = v < 0.0f ? -int(-v) - 1 : int(v);
    int iu1 = iu0 + 1;
    int iv1 = iv0 + 1;
    u -= iu0;
    v -= iv0;
    if (!bm.clamp) {
        iu0 = ((iu0 % w) + w) % w;
        iu1 = ((iu1 % w) + w) % w;
        iv0 = ((iv0 % h) + h) % h;
        iv1 = ((iv1 % h) + h) % h;
    } else {
        iu0 = std::min(std::max(iu0, 0), w - 1);
        iu1 = std::min(std::max(iu1, 0), w - 1);
        iv0 = std::min(std::max(iv0, 0), h - 1);
        iv1 = std::min(std::max(iv1, 0), h - 1);
    }
    auto rgb = [&](int x, int y) { return V3(bm.texels + (size_t(x) + size_t(y)*w)*3); };
    if (!linear) return rgb(iu0, iv0);
    V3 x00 = rgb(iu0, iv0), x01 = rgb(iu1, iv0), x10 = rgb(iu0, iv1), x11 = rgb(iu1, iv1);
    V3 l = (x00*(1.0f - u) + x01*u)*(1.0f - v) + (x10*(1.0f - u) + x11*u)*v; // :124-128
    return bm.scale*l;
}

V3 oracle_ctx::texEval(int tex, float ux, float uy) const
{
    const tgpu_texture &t = sc.textures[tex];
    switch (t.type) {
    case TGPU_TEX_CONSTANT: return V3(t.value);
    case TGPU_TEX_CHECKER: { // textures/CheckerTexture.cpp:64-69
        int ui = int(ux*float(t.res_u)), vi = int(uy*float(t.res_v));
        bool on = (ui ^ vi) & 1;
        return on ? V3(t.on_color) : V3(t.off_color);
    }
    default: return bitmapEval(sc.bitmaps[t.bitmap], ux, uy);
    }
}

// ========================================================================================
// primitives
// ========================================================================================
bool oracle_ctx::quadIntersect(const tgpu_quad &q, uint32_t idx, Ray &ray, Temp &data) const // primitives/Quad.cpp:71-98
{
    V3 normal(q.normal), base(q.base), edge0(q.edge0), edge1(q.edge1);
    float nDotW = ray.dir.dot(normal);
    if (std::abs(nDotW) < 1e-6f) return false;
    float t = normal.dot(base - ray.pos)/nDotW;
    if (t < ray.nearT || t > ray.farT) return false;
    V3 qp = ray.pos + t*ray.dir;
    V3 v = qp - base;
    float l0 = v.dot(edge0)*q.inv_uv_sq[0];
    float l1 = v.dot(edge1)*q.inv_uv_sq[1];
    if (l0 < 0.0f || l0 > 1.0f || l1 < 0.0f || l1 > 1.0f) return false;
    ray.farT = t;
    data.kind = PRIM_QUAD;
    data.index = idx;
    data.p = qp;
    data.u = l0;
    data.v = l1;
    data.backSide = nDotW >= 0.0f;
    return true;
}

bool oracle_ctx::cubeIntersect(const tgpu_cube &cb, uint32_t idx, Ray &ray, Temp &data) const // primitives/Cube.cpp:94-127
{
    V3 p = mulPoint0(cb.inv_rot, ray.pos - V3(cb.pos));
    V3 d = mulPoint0(cb.inv_rot, ray.dir);
    V3 invD = V3(1.0f/d.x, 1.0f/d.y, 1.0f/d.z);
    V3 scale(cb.scale);
    V3 relMin = -scale - p;
    V3 relMax = scale - p;
    float ttMin = ray.nearT, ttMax = ray.farT;
    for (int i = 0; i < 3; ++i) {
        if (invD[i] >= 0.0f) {
            ttMin = fmax2(ttMin, relMin[i]*invD[i]);
            ttMax = fmin2(ttMax, relMax[i]*invD[i]);
        } else {
            ttMax = fmin2(ttMax, relMin[i]*invD[i]);
            ttMin = fmax2(ttMin, relMax[i]*invD[i]);
        }
    }
    if (ttMin <= ttMax) {
        if (ttMin > ray.nearT && ttMin < ray.farT) {
            data.kind = PRIM_CUBE; data.index = idx;
            ray.farT = ttMin;
            data.backSide = false;
        } else if (ttMax > ray.nearT && ttMax < ray.farT) {
            data.kind = PRIM_CUBE; data.index = idx;
            ray.farT = ttMax;
            data.backSide = true;
        }
        return true;
    }
    return false;
}

bool oracle_ctx::sphereIntersect(const tgpu_sphere &sp, uint32_t idx, Ray &ray, Temp &data) const // primitives/Sphere.cpp:68-92
{
    V3 p = ray.pos - V3(sp.pos);
    float B = p.dot(ray.dir);
    float C = p.lengthSq() - sp.radius*sp.radius;
    float detSq = B*B - C;
    if (detSq >= 0.0f) {
        float det = std::sqrt(detSq);
        float t = -B - det;
        if (t < ray.farT && t > ray.nearT) {
            ray.farT = t;
            data.kind = PRIM_SPHERE; data.index = idx; data.backSide = false;
            return true;
        }
        t = -B + det;
        if (t < ray.farT && t > ray.nearT) {
            ray.farT = t;
            data.kind = PRIM_SPHERE; data.index = idx; data.backSide = true;
            return true;
        }
    }
    return false;
}

bool oracle_ctx::diskIntersect(const tgpu_disk &dk, uint32_t idx, Ray &ray, Temp &data) const // primitives/Disk.cpp:64-86
{
    V3 n(dk.n), center(dk.center);
    float nDotW = ray.dir.dot(n);
    float t = n.dot(center - ray.pos)/nDotW;
    if (t < ray.nearT || t > ray.farT) return false;
    V3 q = ray.pos + t*ray.dir;
    V3 v = q - center;
    float rSq = v.lengthSq();
    if (rSq > dk.r*dk.r) return false;
    ray.farT = t;
    data.kind = PRIM_DISK; data.index = idx;
    data.p = q;
    data.rSq = rSq;
    data.backSide = -nDotW < dk.cos_apex;
    return true;
}

void oracle_ctx::intersectionInfo(const Temp &data, Info &info) const
{
    info.kind = data.kind;
    info.index = data.index;
    switch (data.kind) {
    case PRIM_TRI: { // primitives/TriangleMesh.cpp:344-355, :90-106
        uint32_t i = data.index;
        info.Ng = data.Ng.normalized();
        if (sc.tri_smooth[i]) {
            const float *n = sc.tri_normal + size_t(i)*9;
            V3 n0(n), n1(n + 3), n2(n + 6);
            info.Ns = ((1.0f - data.u - data.v)*n0 + data.u*n1 + data.v*n2).normalized();
        } else {
            info.Ns = info.Ng;
        }
        const float *uv = sc.tri_uv + size_t(i)*6;
        float w0 = 1.0f - data.u - data.v;
        info.uvx = w0*uv[0] + data.u*uv[2] + data.v*uv[4];
        info.uvy = w0*uv[1] + data.u*uv[3] + data.v*uv[5];
        info.bsdf = sc.tri_bsdf[i];
        break;
    }
    case PRIM_QUAD: { // primitives/Quad.cpp:123-131
        const tgpu_quad &q = sc.quads[data.index];
        info.Ng = info.Ns = V3(q.normal);
        info.p = data.p;
        info.uvx = data.u; info.uvy = data.v;
        info.bsdf = q.bsdf;
        break;
    }
    case PRIM_CUBE: { // primitives/Cube.cpp:157-170
        const tgpu_cube &cb = sc.cubes[data.index];
        V3 p = mulPoint0(cb.inv_rot, info.p - V3(cb.pos));
        V3 n(0.0f);
        V3 a = vabs(p) - V3(cb.scale);
        int dim = 0; // Vec::maxDim
        if (a.y > a[dim]) dim = 1;
        if (a.z > a[dim]) dim = 2;
        n[dim] = p[dim] < 0.0f ? -1.0f : 1.0f;
        V3 uvw = (p/V3(cb.scale))*0.5f + 0.5f;
        info.Ns = info.Ng = mulPoint0(cb.rot, n);
        info.uvx = uvw[(dim + 1) % 3];
        info.uvy = uvw[(dim + 2) % 3];
        info.bsdf = cb.bsdf;
        break;
    }
    case PRIM_SPHERE: { // primitives/Sphere.cpp:119-128
        const tgpu_sphere &sp = sc.spheres[data.index];
        info.Ns = info.Ng = (info.p - V3(sp.pos))/sp.radius;
        V3 localN = mulVector(sp.inv_rot, info.Ng);
        info.uvx = matan2(localN.y, localN.x)*INV_TWO_PI + 0.5f;
        info.uvy = macos(clampf(localN.z, -1.0f, 1.0f))*INV_PI;
        if (std::isnan(info.uvx)) info.uvx = 0.0f;
        info.bsdf = sp.bsdf;
        break;
    }
    case PRIM_DISK: { // primitives/Disk.cpp:107-120
        const tgpu_disk &dk = sc.disks[data.index];
        info.Ng = info.Ns = V3(dk.n);
        info.p = data.p;
        V3 d = data.p - V3(dk.center);
        float x = d.dot(V3(dk.bitangent));
        float y = d.dot(V3(dk.tangent));
        float v = std::sqrt(data.rSq)/dk.r;
        float u = (x == 0.0f && y == 0.0f) ? 0.0f : (matan2(y, x)*INV_TWO_PI + 0.5f);
        info.uvx = u; info.uvy = v;
        info.bsdf = dk.bsdf;
        break;
    }
    case PRIM_ENV: { // primitives/InfiniteSphere.cpp:96-104
        info.Ng = info.Ns = -data.w;
        info.p = data.p;
        directionToUV(data.w, info.uvx, info.uvy, nullptr);
        info.bsdf = -1;
        break;
    }
    default: break;
    }
    if (data.inst >= 0) { // Instance::intersectionInfo (:335-344) on top of the master's
        const tgpu_instances &is = sc.instance_sets[data.instSet];
        const float *rot = is.rot + size_t(data.inst)*4;
        info.Ng = qrotate(rot, info.Ng);
        info.Ns = qrotate(rot, info.Ns);
        info.p = V3(is.pos + size_t(data.inst)*3) + qrotate(rot, info.p);
        info.kind = PRIM_INSTANCE;
        info.index = uint32_t(data.inst);
    }
}

// ========================================================================================
// "instances" primitives (primitives/Instance.cpp) -- restated WITH the reference's behaviour:
//  * BVH2 over the instance boxes built exactly as Bvh::BvhBuilder(2) + BinaryBvh(prims, 2) do
//    (bvh/BvhBuilder.cpp:99-185, FullSahSplitter.hpp, BinnedSahSplitter.hpp, BinaryBvh.hpp:120-146);
//  * BinaryBvh::trace's ordered traversal (BinaryBvh.hpp:188-283);
//  * Instance::intersect (:290-311): the local ray starts at the leaf's entry distance with farT = infinity, and
//    whichever instance is hit LAST in traversal order wins -- the tree shape is therefore part of the result;
//  * Instance::intersectionInfo (:335-344): the master sees the WORLD hit point, and info.p is transformed by the
//    instance a second time (only Quad masters, which overwrite info.p with their local point, end up correct).
// ========================================================================================
namespace {
const float BOX_MAX = std::numeric_limits<float>::max(), BOX_LOWEST = std::numeric_limits<float>::lowest();
inline float sseMax(float a, float b) { return a > b ? a : b; } // _mm_max_ps / Tungsten::max
inline float sseMin(float a, float b) { return a < b ? a : b; }
struct IBox { // math/Box.hpp
    V3 lo{BOX_MAX, BOX_MAX, BOX_MAX}, hi{BOX_LOWEST, BOX_LOWEST, BOX_LOWEST};
    void grow(const V3 &p) { for (int k = 0; k < 3; ++k) { lo[k] = sseMin(lo[k], p[k]); hi[k] = sseMax(hi[k], p[k]); } }
    void grow(const IBox &b) { for (int k = 0; k < 3; ++k) { lo[k] = sseMin(lo[k], b.lo[k]); hi[k] = sseMax(hi[k], b.hi[k]); } }
    V3 diagonal() const { return V3(sseMax(hi.x - lo.x, 0.0f), sseMax(hi.y - lo.y, 0.0f), sseMax(hi.z - lo.z, 0.0f)); }
    float area() const { V3 d = diagonal(); return (d.x*d.y + d.y*d.z + d.z*d.x)*2.0f; }
    V3 center() const { return (lo + hi)/2.0f; }
};
inline int maxDim(const V3 &v) { int idx = 0; float m = v.x; for (int i = 1; i < 3; ++i) if (v[i] > m) { m = v[i]; idx = i; } return idx; }
struct IPrim { IBox box; V3 centroid; uint32_t id; float area; }; // bvh/Primitive.hpp
struct ISplit { IBox lBox, rBox, lCentroidBox, rCentroidBox; int dim; uint32_t idx; float cost; };
typedef std::vector<IPrim> IPrims;

inline int cvttss2si(float f) // int(float) as x86 does it, NaN/overflow -> INT_MIN
{
    if (!(f > -2147483904.0f && f < 2147483648.0f)) return std::numeric_limits<int>::min();
    return int(f);
}

struct FullSahSplitter { // bvh/FullSahSplitter.hpp
    static void sortDim(uint32_t start, uint32_t end, int dim, IPrims &prims)
    {
        std::sort(prims.begin() + start, prims.begin() + end + 1, [&](const IPrim &a, const IPrim &b) {
            if (a.centroid[dim] == b.centroid[dim]) return a.id < b.id;
            return a.centroid[dim] < b.centroid[dim];
        });
    }
    static void findSahSplit(uint32_t start, uint32_t end, int dim, IPrims &prims, ISplit &split)
    {
        sortDim(start, end, dim, prims);
        IBox rBox; // computeAreas
        for (uint32_t i = end; i > start; --i) { rBox.grow(prims[i].box); prims[i].area = rBox.area(); }
        rBox.grow(prims[start].box);
        prims[start].area = rBox.area();
        IBox lBox = prims[start].box;
        for (uint32_t i = start + 1; i <= end; ++i) {
            float cost = 1.0f*(lBox.area()*float(i - start) + prims[i].area*float(end - i + 1));
            if (cost < split.cost) { split.dim = dim; split.idx = i; split.lBox = lBox; split.cost = cost; }
            lBox.grow(prims[i].box);
        }
        if (split.dim == dim) {
            IBox r;
            for (uint32_t i = split.idx; i <= end; ++i) r.grow(prims[i].box);
            split.rBox = r;
        }
    }
    static void twoWaySahSplit(uint32_t start, uint32_t end, IPrims &prims, const IBox &geomBox, ISplit &split)
    {
        split.dim = -1;
        split.cost = geomBox.area()*(float(end - start + 1)*1.0f - 1.0f);
        findSahSplit(start, end, 0, prims, split);
        findSahSplit(start, end, 1, prims, split);
        findSahSplit(start, end, 2, prims, split);
        if (split.dim == -1) {
            split.dim = maxDim(geomBox.diagonal());
            split.idx = (end - start + 1)/2 + start;
            sortDim(start, end, split.dim, prims);
            for (uint32_t i = start; i <= end; ++i) (i < split.idx ? split.lBox : split.rBox).grow(prims[i].box);
        } else if (split.dim != 2) {
            sortDim(start, end, split.dim, prims);
        }
    }
};

struct BinnedSahSplitter { // bvh/BinnedSahSplitter.hpp
    static const int BinCount = 32;
    IBox geomBounds[3][BinCount], centroidBounds[3][BinCount];
    V3 centroidMin, centroidSpan;
    int counts[3][BinCount];
    BinnedSahSplitter() { std::memset(counts, 0, sizeof(counts)); }
    int primitiveBin(const IPrim &p, int dim) const
    {
        int b = cvttss2si(float(BinCount)*((p.centroid[dim] - centroidMin[dim])/centroidSpan[dim]));
        return b < 0 ? 0 : (b > BinCount - 1 ? BinCount - 1 : b); // clamp = min(max(val, lo), hi)
    }
    void findSahSplit(int dim, ISplit &split)
    {
        int rCount = 0, rCounts[BinCount];
        IBox rBox, rBoxes[BinCount];
        for (int i = BinCount - 1; i > 0; --i) {
            rCount += counts[dim][i];
            rBox.grow(geomBounds[dim][i]);
            rCounts[i] = rCount;
            rBoxes[i] = rBox;
        }
        int lCount = counts[dim][0];
        IBox lBox = geomBounds[dim][0];
        for (int i = 1; i < BinCount; ++i) {
            float cost = 1.0f*(lBox.area()*float(lCount) + rBoxes[i].area()*float(rCounts[i]));
            if (cost < split.cost) { split.dim = dim; split.idx = uint32_t(i); split.cost = cost; }
            lCount += counts[dim][i];
            lBox.grow(geomBounds[dim][i]);
        }
    }
    uint32_t sortByBin(uint32_t start, uint32_t end, IPrims &prims, int dim, int bin)
    {
        uint32_t left = start, right = end;
        while (left < right) {
            while (left < right && primitiveBin(prims[left], dim) < bin) left++;
            while (right > left && primitiveBin(prims[right], dim) >= bin) right--;
            if (left != right) std::swap(prims[left], prims[right]);
        }
        if (left == end || right == start) return end + 1;
        return left;
    }
    void fullSplit(uint32_t start, uint32_t end, IPrims &prims, const IBox &box, const IBox &centroidBox, ISplit &split)
    {
        centroidMin = centroidBox.lo; // partialBin
        centroidSpan = centroidBox.diagonal();
        for (int d = 0; d < 3; ++d)
            if (centroidSpan[d] > 0.0f)
                for (uint32_t i = start; i <= end; ++i) {
                    int idx = primitiveBin(prims[i], d);
                    geomBounds[d][idx].grow(prims[i].box);
                    centroidBounds[d][idx].grow(prims[i].centroid);
                    counts[d][idx]++;
                }
        split.dim = -1; // twoWaySahSplit
        split.cost = box.area()*(float(end - start + 1)*1.0f - 1.0f);
        for (int d = 0; d < 3; ++d)
            if (centroidSpan[d] > 0.0f) findSahSplit(d, split);
        if (split.dim == -1) { split.dim = maxDim(box.diagonal()); split.idx = BinCount/2; }
        int bin = int(split.idx);
        split.idx = sortByBin(start, end, prims, split.dim, bin);
        if (split.idx > end || centroidSpan == 0.0f) {
            split.idx = start + (end - start + 1)/2;
            split.lBox = prims[start].box;
            split.rBox = prims[end].box;
            split.lCentroidBox = IBox(); split.lCentroidBox.lo = split.lCentroidBox.hi = prims[start].centroid;
            split.rCentroidBox = IBox(); split.rCentroidBox.lo = split.rCentroidBox.hi = prims[end].centroid;
            for (uint32_t i = start + 1; i < end; ++i) {
                if (i < split.idx) { split.lBox.grow(prims[i].box); split.lCentroidBox.grow(prims[i].centroid); }
                else { split.rBox.grow(prims[i].box); split.rCentroidBox.grow(prims[i].centroid); }
            }
        } else {
            split.lBox = geomBounds[split.dim][0];
            split.rBox = geomBounds[split.dim][BinCount - 1];
            split.lCentroidBox = centroidBounds[split.dim][0];
            split.rCentroidBox = centroidBounds[split.dim][BinCount - 1];
            for (int i = 1; i < BinCount - 1; ++i) {
                if (i < bin) { split.lBox.grow(geomBounds[split.dim][i]); split.lCentroidBox.grow(centroidBounds[split.dim][i]); }
                else { split.rBox.grow(geomBounds[split.dim][i]); split.rCentroidBox.grow(centroidBounds[split.dim][i]); }
            }
        }
    }
};

struct NaiveNode { // bvh/NaiveBvhNode.hpp
    std::unique_ptr<NaiveNode> child[2];
    IBox box;
    uint32_t id = 0;
    bool isLeaf() const { return !child[0]; }
};

// BvhBuilder.cpp:99-185 with branchFactor 2 (single-threaded paths; > 1M instances are refused at load)
void naiveBuild(NaiveNode &dst, uint32_t start, uint32_t end, IPrims &prims, const IBox &geomBox, const IBox &centroidBox, int &depth)
{
    depth = 1;
    dst.box = geomBox;
    uint32_t numPrims = end - start + 1;
    if (numPrims == 1) {
        dst.id = prims[start].id;
    } else if (numPrims <= 2) {
        for (uint32_t i = start; i <= end; ++i) {
            dst.child[i - start].reset(new NaiveNode());
            dst.child[i - start]->box = prims[i].box;
            dst.child[i - start]->id = prims[i].id;
        }
    } else {
        ISplit split;
        if (numPrims <= 64) FullSahSplitter::twoWaySahSplit(start, end, prims, geomBox, split);
        else { std::unique_ptr<BinnedSahSplitter> b(new BinnedSahSplitter()); b->fullSplit(start, end, prims, geomBox, centroidBox, split); }
        uint32_t starts[2] = {start, split.idx}, ends[2] = {split.idx - 1, end};
        IBox geomBoxes[2] = {split.lBox, split.rBox}, centroidBoxes[2] = {split.lCentroidBox, split.rCentroidBox};
        for (int i = 0; i < 2; ++i) dst.child[i].reset(new NaiveNode());
        for (int i = 0; i < 2; ++i) {
            int d;
            naiveBuild(*dst.child[i], starts[i], ends[i], prims, geomBoxes[i], centroidBoxes[i], d);
            depth = std::max(depth, d + 1);
        }
    }
}

struct TinyNode { float box[3][4]; int32_t l = -1, r = -1; uint32_t primIndex = 0, count = 0; bool isNode() const { return r >= 0; } };
struct InstSet {
    std::vector<TinyNode> nodes;
    std::vector<uint32_t> primIndices;
    IBox bounds;
    int depth = 0;
    std::vector<TriBvh> masterBvh; // per master; empty unless the master is a mesh
};

void setJointBbox(TinyNode &n, const IBox &l, const IBox &r)
{
    for (int k = 0; k < 3; ++k) { n.box[k][0] = l.lo[k]; n.box[k][1] = r.lo[k]; n.box[k][2] = l.hi[k]; n.box[k][3] = r.hi[k]; }
}

uint32_t tinyBuild(InstSet &s, const NaiveNode *node, uint32_t head, uint32_t &tail, uint32_t &primIndex, uint32_t maxPrimsPerLeaf)
{ // BinaryBvh.hpp:120-146
    if (node->isLeaf()) {
        setJointBbox(s.nodes[head], node->box, node->box);
        s.nodes[head].count = 1; s.nodes[head].primIndex = primIndex;
        s.nodes[head].r = -1;
        s.primIndices[primIndex++] = node->id;
        return 1;
    }
    uint32_t childIdx = tail;
    setJointBbox(s.nodes[head], node->child[0]->box, node->child[1]->box);
    s.nodes[head].l = int32_t(childIdx);
    s.nodes[head].r = int32_t(childIdx + 1);
    tail += 2;
    uint32_t lPrims = tinyBuild(s, node->child[0].get(), childIdx, tail, primIndex, maxPrimsPerLeaf);
    uint32_t rPrims = tinyBuild(s, node->child[1].get(), childIdx + 1, tail, primIndex, maxPrimsPerLeaf);
    if (lPrims + rPrims <= maxPrimsPerLeaf) {
        s.nodes[head].count = lPrims + rPrims; s.nodes[head].primIndex = s.nodes[childIdx].primIndex;
        s.nodes[head].r = -1;
        tail = childIdx;
    }
    return lPrims + rPrims;
}

inline V3 qrot(const float *q, const V3 &o) { return qrotate(q, o); }

inline bool bboxIntersection(const IBox &box, const V3 &o, const V3 &d, float &tMin, float &tMax) // BinaryBvh.hpp:148-171
{
    V3 invD(1.0f/d.x, 1.0f/d.y, 1.0f/d.z);
    V3 relMin = box.lo - o, relMax = box.hi - o;
    float ttMin = tMin, ttMax = tMax;
    for (int i = 0; i < 3; ++i) {
        if (invD[i] >= 0.0f) {
            ttMin = fmax2(ttMin, relMin[i]*invD[i]);
            ttMax = fmin2(ttMax, relMax[i]*invD[i]);
        } else {
            ttMax = fmin2(ttMax, relMin[i]*invD[i]);
            ttMin = fmax2(ttMin, relMax[i]*invD[i]);
        }
    }
    if (ttMin <= ttMax) { tMin = ttMin; tMax = ttMax; return true; }
    return false;
}
} // namespace

struct oracle_instances { std::vector<InstSet> sets; };

// BinaryBvh(prims, 2) (bvh/BinaryBvh.hpp:174-195) on top of BvhBuilder(2) (bvh/BvhBuilder.cpp:99-185)
static void buildBinaryBvh(InstSet &s, IPrims &prims)
{
    s.nodes.clear();
    if (prims.empty()) {
        s.depth = 0;
        s.nodes.emplace_back();
        setJointBbox(s.nodes[0], IBox(), IBox());
        s.nodes[0].l = 0; s.nodes[0].r = 0; // rChild = itself: an inner node that never passes the box test
        return;
    }
    IBox geomBounds, centroidBounds;
    for (const IPrim &p : prims) { geomBounds.grow(p.box); centroidBounds.grow(p.centroid); }
    NaiveNode root;
    size_t count = prims.size();
    naiveBuild(root, 0, uint32_t(count - 1), prims, geomBounds, centroidBounds, s.depth);
    s.bounds = root.box;
    s.primIndices.resize(count);
    s.nodes.resize(2*count);
    uint32_t tail = 1, primIndex = 0;
    tinyBuild(s, &root, 0, tail, primIndex, 2);
    s.nodes.resize(tail);
}

void oracle_ctx::buildInstances()
{
    inst.reset(new oracle_instances());
    for (uint32_t si = 0; si < sc.n_instance_sets; ++si) {
        const tgpu_instances &is = sc.instance_sets[si];
        inst->sets.emplace_back();
        InstSet &s = inst->sets.back();
        s.masterBvh.resize(is.n_masters);
        for (uint32_t m = 0; m < is.n_masters; ++m)
            if (is.masters[m].kind == TGPU_MASTER_MESH) buildBvh(is.masters[m].first, is.masters[m].count, s.masterBvh[m]);
        // Instance::prepareForRender :405-424
        IPrims prims;
        prims.reserve(is.n_instances);
        for (uint32_t i = 0; i < is.n_instances; ++i) {
            const tgpu_master &mm = is.masters[is.master_id[i]];
            V3 bmin(mm.bounds_min), bmax(mm.bounds_max), pos(is.pos + size_t(i)*3);
            IBox bGlobal;
            for (int xi = 0; xi < 2; ++xi)
                for (int yi = 0; yi < 2; ++yi)
                    for (int zi = 0; zi < 2; ++zi) {
                        V3 t{float(xi), float(yi), float(zi)};
                        V3 corner = bmin*(V3(1.0f, 1.0f, 1.0f) - t) + bmax*t; // lerp(a, b, ratio) = a*(1 - ratio) + b*ratio
                        bGlobal.grow(pos + qrot(is.rot + size_t(i)*4, corner));
                    }
            IPrim p;
            p.box = bGlobal; p.centroid = bGlobal.center(); p.id = i; p.area = bGlobal.area();
            prims.push_back(p);
        }
        buildBinaryBvh(s, prims);
    }
}

// A master primitive's intersect() on the instance-local ray (mesh: TriangleMesh.cpp:317-335 on the master's own
// Embree scene; quad / cube: the scene primitives' code)
bool oracle_ctx::masterIntersect(uint32_t setIndex, uint32_t master, Ray &ray, Temp &data, Counters &c) const
{
    const InstSet &s = inst->sets[setIndex];
    const tgpu_master &mm = sc.instance_sets[setIndex].masters[master];
    if (mm.kind == TGPU_MASTER_QUAD) return quadIntersect(sc.quads[mm.first], mm.first, ray, data);
    if (mm.kind == TGPU_MASTER_CUBE) return cubeIntersect(sc.cubes[mm.first], mm.first, ray, data);
    float t = ray.farT, u = 0, v = 0;
    uint32_t prim = 0xFFFFFFFFu;
    intersectTris(s.masterBvh[master], ray, t, u, v, prim, c);
    if (prim == 0xFFFFFFFFu) return false;
    TRC("  master %u: tri=%u t=%08x u=%08x v=%08x (local nearT=%08x)\n", master, prim, hx(t), hx(u), hx(v), hx(ray.nearT));
    ray.farT = t;
    data.kind = PRIM_TRI;
    data.index = prim;
    data.u = u;
    data.v = v;
    const float *p = sc.tri_pos + size_t(prim)*9;
    V3 p0(p), p1(p + 3), p2(p + 6);
    data.Ng = (p1 - p0).cross(p2 - p0);
    data.backSide = data.Ng.dot(ray.dir) > 0.0f; // with the LOCAL direction
    return true;
}

// BinaryBvh::trace (bvh/BinaryBvh.hpp:188-283), the SSE lanes written out per child; intersector(id, tMin) may shorten ray.farT
template <class F>
static void walkBinaryBvh(const InstSet &s, Ray &ray, uint64_t &nodeVisits, F intersector)
{
    float tMin = ray.nearT, tMax = ray.farT;
    if (!bboxIntersection(s.bounds, ray.pos, ray.dir, tMin, tMax)) return;
    struct StackNode { int32_t node; float tMin; };
    std::vector<StackNode> stack(size_t(s.depth) + 2);
    size_t sp = 0;
    const V3 o = ray.pos, d = ray.dir;
    const V3 invDir(1.0f/d.x, 1.0f/d.y, 1.0f/d.z);
    const float nearLane = ray.nearT;
    float negFarLane = -ray.farT;
    int32_t node = 0;
    while (true) {
        bool skipLeaf = false;
        while (s.nodes[size_t(node)].isNode()) {
            const TinyNode &n = s.nodes[size_t(node)];
            nodeVisits++;
            float mn[2], mx[2];
            for (int ch = 0; ch < 2; ++ch) {
                float tn[3], tfNeg[3];
                for (int k = 0; k < 3; ++k) {
                    float lo = n.box[k][ch], hi = n.box[k][2 + ch];
                    float nearPlane = d[k] >= 0.0f ? lo : hi, farPlane = d[k] >= 0.0f ? hi : lo;
                    tn[k] = (nearPlane - o[k])*invDir[k];
                    tfNeg[k] = (farPlane - o[k])*(-invDir[k]);
                }
                mn[ch] = sseMax(sseMax(sseMax(tn[0], tn[1]), tn[2]), nearLane);
                mx[ch] = -sseMax(sseMax(sseMax(tfNeg[0], tfNeg[1]), tfNeg[2]), negFarLane);
            }
            bool intersectsL = mn[0] <= mx[0], intersectsR = mn[1] <= mx[1];
            if (intersectsL && intersectsR) {
                if (mn[0] < mn[1]) { stack[sp++] = StackNode{n.r, mn[1]}; node = n.l; tMin = mn[0]; }
                else { stack[sp++] = StackNode{n.l, mn[0]}; node = n.r; tMin = mn[1]; }
            } else if (intersectsL) { node = n.l; tMin = mn[0]; }
            else if (intersectsR) { node = n.r; tMin = mn[1]; }
            else { skipLeaf = true; break; }
        }
        if (!skipLeaf) {
            const TinyNode &leaf = s.nodes[size_t(node)];
            for (uint32_t i = leaf.primIndex; i < leaf.primIndex + leaf.count; ++i) intersector(s.primIndices[i], tMin);
            tMax = fmin2(tMax, ray.farT);
            negFarLane = -tMax;
        }
        bool done = false;
        while (true) { // pop
            if (sp == 0) { done = true; break; }
            --sp;
            node = stack[sp].node;
            tMin = stack[sp].tMin;
            if (tMax < tMin) continue;
            break;
        }
        if (done) break;
    }
}

bool oracle_ctx::instanceIntersect(uint32_t setIndex, Ray &ray, Temp &data, Counters &c) const // Instance.cpp:290-311
{
    const InstSet &s = inst->sets[setIndex];
    const tgpu_instances &is = sc.instance_sets[setIndex];
    bool hit = false;
    uint32_t prim = 0;
    walkBinaryBvh(s, ray, c.nodes, [&](uint32_t id, float tMin) {
        float invRot[4] = {is.rot[size_t(id)*4], -is.rot[size_t(id)*4 + 1], -is.rot[size_t(id)*4 + 2], -is.rot[size_t(id)*4 + 3]};
        Ray localRay{qrot(invRot, ray.pos - V3(is.pos + size_t(id)*3)), qrot(invRot, ray.dir), tMin, INF}; // Ray::scatter: farT = infinity
        if (masterIntersect(setIndex, is.master_id[id], localRay, data, c)) {
            hit = true;
            prim = id;
            ray.farT = localRay.farT;
        }
    });
    if (hit) {
        data.inst = int32_t(prim);
        data.instSet = int32_t(setIndex);
    }
    return hit;
}

// Reference mode only (rcp_mode = 1): which primitive TraceableScene::intersect reports is decided by Embree wherever binary32 leaves room, and the
// oracle restates Embree for it (oracle/embree_bvh4.h; pinned by tests/golden/ref_kat5.npz against the compiled library).  The scene's finites are the
// primitives of ONE user geometry in a BVH4 with one primitive per leaf (renderer/TraceableScene.hpp:107-129); Primitive::intersect is only called for
// leaves the traversal reaches, with the ray interval the traversal has at that moment:
//   * a ray leaving the Cornell box through the edge between the ceiling quad and a wall can pass the ceiling's own test (Quad.cpp:71-98, u = 3e-8)
//     and miss the ceiling's flat box by one ulp of Embree's slab arithmetic -> the reference returns the wall, 3 ulp farther
//     (cornell-box/scene_sobol.json at 96x54, pixel (67,7) sample 41);
//   * two quads reached at the SAME binary32 distance (floor / wall edge): Quad.cpp:78 keeps t <= farT, so the leaf visited LAST wins unless its
//     box is culled against the shortened ray (64x36, pixel (47,33) sample 15);
//   * a quad and a cube face at the same distance (boxes standing on the floor): Cube.cpp:115 wants ttMin < farT, so the cube only wins when its
//     leaf comes first and the quad's box is then culled.
// Each TriangleMesh is one such primitive and owns a second BVH4 over its triangles (TriangleMesh.cpp:531-566) with Triangle4 leaves: four triangles
// are tested at once, the smallest t of a block wins (lowest lane on a tie), and a later block must pass T < |den|*tfar -- first found keeps a tie.
// The GPU contract (rcp_mode = 0) is tree-free: the primitive's own test, nearest hit = min (t, id).
void oracle_ctx::buildReferenceAccel()
{
    refAccel = false;
#if defined(__SSE__)
    if (!rcpMode) return;
    using embree_restated::Box;
    std::vector<Box> boxes;
    auto grow = [](Box &b, const V3 &p) { for (int k = 0; k < 3; ++k) { b.lo[k] = std::min(b.lo[k], p[k]); b.hi[k] = std::max(b.hi[k], p[k]); } };
    for (uint32_t i = 0; i < sc.n_instance_sets; ++i) { // Instance::bounds (Instance.cpp:380-383)
        Box b;
        for (int k = 0; k < 3; ++k) { b.lo[k] = inst->sets[i].bounds.lo[k]; b.hi[k] = inst->sets[i].bounds.hi[k]; }
        boxes.push_back(b);
        refFinites.push_back(RefFinite{PRIM_INSTANCE, i});
    }
    for (uint32_t i = 0; i < sc.n_quads; ++i) { // Quad::bounds (Quad.cpp:281-289)
        const tgpu_quad &q = sc.quads[i];
        Box b = embree_restated::emptyBox();
        grow(b, V3(q.base)); grow(b, V3(q.base) + V3(q.edge0)); grow(b, V3(q.base) + V3(q.edge1)); grow(b, V3(q.base) + V3(q.edge0) + V3(q.edge1));
        boxes.push_back(b);
        refFinites.push_back(RefFinite{PRIM_QUAD, i});
    }
    for (uint32_t i = 0; i < sc.n_cubes; ++i) { // Cube::bounds (Cube.cpp:333-344)
        const tgpu_cube &cb = sc.cubes[i];
        Box b = embree_restated::emptyBox();
        for (int c = 0; c < 8; ++c)
            grow(b, V3(cb.pos) + mulPoint0(cb.rot, V3(c & 1 ? cb.scale[0] : -cb.scale[0], c & 2 ? cb.scale[1] : -cb.scale[1], c & 4 ? cb.scale[2] : -cb.scale[2])));
        boxes.push_back(b);
        refFinites.push_back(RefFinite{PRIM_CUBE, i});
    }
    for (uint32_t i = 0; i < sc.n_spheres; ++i) { // Sphere::bounds (Sphere.cpp:273-276)
        const tgpu_sphere &sp = sc.spheres[i];
        Box b;
        for (int k = 0; k < 3; ++k) { b.lo[k] = sp.pos[k] - sp.radius; b.hi[k] = sp.pos[k] + sp.radius; }
        boxes.push_back(b);
        refFinites.push_back(RefFinite{PRIM_SPHERE, i});
    }
    for (uint32_t i = 0; i < sc.n_disks; ++i) { // Disk::bounds (Disk.cpp:298-306)
        const tgpu_disk &dk = sc.disks[i];
        Box b = embree_restated::emptyBox();
        V3 c(dk.center), t = V3(dk.tangent)*dk.r, bt = V3(dk.bitangent)*dk.r;
        grow(b, c - t - bt); grow(b, c + t - bt); grow(b, c + t + bt); grow(b, c - t + bt);
        boxes.push_back(b);
        refFinites.push_back(RefFinite{PRIM_DISK, i});
    }
    refMesh.resize(sc.n_meshes);
    for (uint32_t m = 0; m < sc.n_meshes; ++m) { // TriangleMesh::bounds (TriangleMesh.cpp:519-522)
        Box b;
        for (int k = 0; k < 3; ++k) { b.lo[k] = sc.mesh_bounds[size_t(m)*6 + k]; b.hi[k] = sc.mesh_bounds[size_t(m)*6 + 3 + k]; }
        boxes.push_back(b);
        refFinites.push_back(RefFinite{PRIM_TRI, m});
        uint32_t first = sc.mesh_first_tri[m], n = sc.mesh_first_tri[m + 1] - first;
        std::vector<Box> tb(n);
        for (uint32_t i = 0; i < n; ++i) { // TriangleMesh::bounds(i) (kernels/common/scene_triangle_mesh.h:154-161)
            tb[i] = embree_restated::emptyBox();
            for (int v = 0; v < 3; ++v) grow(tb[i], V3(sc.tri_pos + size_t(first + i)*9 + size_t(v)*3));
        }
        refMesh[m].build(tb, 4, 4, 28); // BVH4Triangle4SceneBuilderSAH (bvh_builder_sah.cpp:773: sahBlockSize 4, minLeafSize 4, maxLeafSize inf -> 4*7 blocks)
    }
    // The order of the primitives inside the user geometry (the scene file's) is not part of the flattened scene.  It only matters where Embree falls
    // back to a median split by position in the array (every centroid of a subtree coincides), which no scene on the path does.
    refTop.build(boxes, 4, 1, 1); // BVH4VirtualSceneBuilderSAH (bvh_builder_sah.cpp:811-815, common/state.cpp:83-84)
    refAccel = true;
#endif
}

bool oracle_ctx::refMeshIntersect(uint32_t mesh, Ray &ray, Temp &data, Counters &c) const // TriangleMesh::intersect (TriangleMesh.cpp:317-335)
{
    using embree_restated::Bvh4;
    const Bvh4 &bvh = refMesh[mesh];
    const uint32_t firstTri = sc.mesh_first_tri[mesh];
    Bvh4::Ray er;
    for (int k = 0; k < 3; ++k) { er.org[k] = ray.pos[k]; er.dir[k] = ray.dir[k]; }
    er.tnear = ray.nearT;
    er.tfar = ray.farT;
    uint32_t prim = 0xFFFFFFFFu;
    float u = 0, v = 0;
    bvh.intersect(er, false, [&](uint32_t first, uint32_t count, Bvh4::Ray &r) {
        for (uint32_t block = 0; block < count; block += 4) { // ArrayIntersector1 over Triangle4 blocks; Intersect1EpilogM: select_min(valid, vt)
            int best = -1;
            TriHit bestHit{false, 0, 0, 0};
            for (uint32_t lane = 0; lane < 4 && block + lane < count; ++lane) {
                uint32_t id = firstTri + bvh.prims()[first + block + lane];
                c.tris++;
                TriHit h = triTest(tris[id], ray, r.tfar, 1);
                if (h.valid && (best < 0 || h.t < bestHit.t)) { best = int(lane); bestHit = h; }
            }
            if (best >= 0) {
                r.tfar = bestHit.t;
                prim = firstTri + bvh.prims()[first + block + uint32_t(best)];
                u = bestHit.u;
                v = bestHit.v;
            }
        }
    }, &c.nodes);
    if (prim == 0xFFFFFFFFu) return false;
    ray.farT = er.tfar;
    data.inst = data.instSet = -1;
    data.kind = PRIM_TRI;
    data.index = prim;
    data.u = u;
    data.v = v;
    const float *p = sc.tri_pos + size_t(prim)*9;
    V3 p0(p), p1(p + 3), p2(p + 6);
    data.Ng = (p1 - p0).cross(p2 - p0); // unnormalizedGeometricNormalAt :81-88
    data.backSide = data.Ng.dot(ray.dir) > 0.0f;
    return true;
}

void oracle_ctx::refIntersect(Ray &ray, Temp &data, Counters &c) const // rtcIntersect(_scene, eRay) of TraceableScene::intersect (:175-177)
{
    using embree_restated::Bvh4;
    Bvh4::Ray er;
    for (int k = 0; k < 3; ++k) { er.org[k] = ray.pos[k]; er.dir[k] = ray.dir[k]; }
    er.tnear = ray.nearT;
    er.tfar = ray.farT;
    refTop.intersect(er, true, [&](uint32_t first, uint32_t /*count = 1*/, Bvh4::Ray &r) {
        const RefFinite &f = refFinites[refTop.prims()[first]];
        bool hit = false; // the intersect callback (TraceableScene.hpp:114-121): embreeRay.tfar = ray.farT() when the primitive says yes
        switch (f.kind) {
        case PRIM_INSTANCE: hit = instanceIntersect(f.index, ray, data, c); break;
        case PRIM_QUAD: hit = quadIntersect(sc.quads[f.index], f.index, ray, data); if (hit) data.inst = data.instSet = -1; break;
        case PRIM_CUBE: { float before = ray.farT; hit = cubeIntersect(sc.cubes[f.index], f.index, ray, data); if (ray.farT != before) data.inst = data.instSet = -1; break; }
        case PRIM_SPHERE: hit = sphereIntersect(sc.spheres[f.index], f.index, ray, data); if (hit) data.inst = data.instSet = -1; break;
        case PRIM_DISK: hit = diskIntersect(sc.disks[f.index], f.index, ray, data); if (hit) data.inst = data.instSet = -1; break;
        case PRIM_TRI: hit = refMeshIntersect(f.index, ray, data, c); break;
        default: break;
        }
        if (hit) r.tfar = ray.farT;
    }, &c.nodes);
}

bool oracle_ctx::intersect(Ray &ray, Temp &data, Info &info, Counters &c) const // renderer/TraceableScene.hpp:170-192
{
    info.kind = PRIM_NONE;
    data.kind = PRIM_NONE;
    data.inst = data.instSet = -1;
    // Order of the finites: "instances" primitives first, then everything else.  Embree visits its user geometries
    // front to back by box entry; an Instance is the only primitive whose result depends on what was found before
    // it (it ignores the incoming farT for its local rays), and its box is normally entered first.
    if (refAccel) {
        refIntersect(ray, data, c);
    } else {
    for (uint32_t i = 0; i < sc.n_instance_sets; ++i) instanceIntersect(i, ray, data, c);
    for (uint32_t i = 0; i < sc.n_quads; ++i)
        if (quadIntersect(sc.quads[i], i, ray, data)) data.inst = data.instSet = -1;
    for (uint32_t i = 0; i < sc.n_cubes; ++i) {
        float before = ray.farT;
        cubeIntersect(sc.cubes[i], i, ray, data);
        if (ray.farT != before) data.inst = data.instSet = -1;
    }
    for (uint32_t i = 0; i < sc.n_spheres; ++i)
        if (sphereIntersect(sc.spheres[i], i, ray, data)) data.inst = data.instSet = -1;
    for (uint32_t i = 0; i < sc.n_disks; ++i)
        if (diskIntersect(sc.disks[i], i, ray, data)) data.inst = data.instSet = -1;
    float t = ray.farT, u = 0, v = 0;
    uint32_t prim = 0xFFFFFFFFu;
    intersectTris(sceneBvh, ray, t, u, v, prim, c);
    if (prim != 0xFFFFFFFFu) { // primitives/TriangleMesh.cpp:317-335
        data.inst = data.instSet = -1;
        ray.farT = t;
        data.kind = PRIM_TRI;
        data.index = prim;
        data.u = u;
        data.v = v;
        const float *p = sc.tri_pos + size_t(prim)*9;
        V3 p0(p), p1(p + 3), p2(p + 6);
        data.Ng = (p1 - p0).cross(p2 - p0); // unnormalizedGeometricNormalAt :81-88
        data.backSide = data.Ng.dot(ray.dir) > 0.0f;
    }
    } // !refAccel
    if (data.kind != PRIM_NONE) {
        info.p = ray.pos + ray.dir*ray.farT;
        info.w = ray.dir;
        info.epsilon = 5e-4f;
        intersectionInfo(data, info);
        return true;
    }
    return false;
}

void oracle_ctx::directionToUV(const V3 &wi, float &ux, float &uy, float *sinTheta) const // InfiniteSphere.cpp:27-38
{
    V3 wLocal = mulPoint0(sc.env.inv_rot, wi);
    if (sinTheta) *sinTheta = std::sqrt(fmax2(1.0f - wLocal.y*wLocal.y, 0.0f));
    ux = matan2(wLocal.z, wLocal.x)*INV_TWO_PI + 0.5f;
    uy = macos(-wLocal.y)*INV_PI;
}

V3 oracle_ctx::uvToDirection(float ux, float uy, float &sinTheta) const // InfiniteSphere.cpp:40-50
{
    float phi = (ux - 0.5f)*TWO_PI;
    float theta = uy*PI;
    sinTheta = msin(theta);
    return mulPoint0(sc.env.rot, V3(mcos(phi)*sinTheta, -mcos(theta), msin(phi)*sinTheta));
}

float oracle_ctx::envPdf(float ux, float uy) const // BitmapTexture.cpp:452-455, Distribution2D.hpp:66-71
{
    int w = sc.env.dist_w, h = sc.env.dist_h;
    int row = int((1.0f - uy)*h), column = int(ux*w);
    row = std::min(std::max(row, 0), h - 1);
    column = std::min(std::max(column, 0), w - 1);
    return sc.env.pdf[size_t(row)*w + column]*sc.env.marginal_pdf[row]*w*h;
}

bool oracle_ctx::primEmissive(int kind, uint32_t index) const
{
    if (kind == PRIM_QUAD) return sc.quads[index].emission_tex >= 0;
    if (kind == PRIM_CUBE) return sc.cubes[index].emission_tex >= 0;
    if (kind == PRIM_SPHERE) return sc.spheres[index].emission_tex >= 0;
    if (kind == PRIM_DISK) return sc.disks[index].emission_tex >= 0;
    if (kind == PRIM_ENV) return true;
    return false;
}
bool oracle_ctx::primSamplable(int kind, uint32_t /*index*/) const
{
    if (kind == PRIM_QUAD) return true;
    if (kind == PRIM_ENV) return sc.env.samplable != 0;
    return true; // meshes/cubes: isSamplable() true, but never emissive in scope
}

V3 oracle_ctx::primEvalDirect(const Temp &data, const Info &info) const
{
    switch (data.kind) {
    case PRIM_QUAD: { // Quad.cpp:235-238
        const tgpu_quad &q = sc.quads[data.index];
        if (q.emission_tex < 0 || data.backSide) return V3(0.0f);
        return texEval(q.emission_tex, info.uvx, info.uvy);
    }
    case PRIM_SPHERE: { // Sphere.cpp:242-245
        const tgpu_sphere &sp = sc.spheres[data.index];
        if (sp.emission_tex < 0 || data.backSide) return V3(0.0f);
        return texEval(sp.emission_tex, info.uvx, info.uvy);
    }
    case PRIM_DISK: { // Disk.cpp:245-248
        const tgpu_disk &dk = sc.disks[data.index];
        if (dk.emission_tex < 0 || data.backSide) return V3(0.0f);
        return texEval(dk.emission_tex, info.uvx, info.uvy);
    }
    case PRIM_ENV: return texEval(sc.env.emission_tex, info.uvx, info.uvy); // InfiniteSphere.cpp:241-244
    default: return V3(0.0f); // Primitive::evalDirect with no emission (Primitive.cpp:104-111)
    }
}

bool oracle_ctx::lightSampleDirect(const tgpu_light &l, const V3 &p, Sampler &s, LightSample &out) const
{
    if (l.kind == TGPU_LIGHT_ENV) { // InfiniteSphere.cpp:161-176
        const tgpu_texture &tex = sc.textures[sc.env.emission_tex];
        if (tex.type == TGPU_TEX_CONSTANT) {
            float xi0 = s.next1D(), xi1 = s.next1D();
            float phi = xi0*TWO_PI; // SampleWarp::uniformSphere (SampleWarp.hpp:96-107)
            float z = xi1*2.0f - 1.0f;
            float r = std::sqrt(fmax2(1.0f - z*z, 0.0f));
            out.d = V3(mcos(phi)*r, msin(phi)*r, z);
            out.dist = INF;
            out.pdf = INV_FOUR_PI;
            return true;
        }
        float uvx = s.next1D(), uvy = s.next1D();
        // BitmapTexture::sample (:433-439) -> Distribution2D::warp (Distribution2D.hpp:66-76)
        int w = sc.env.dist_w, h = sc.env.dist_h;
        const float *mc = sc.env.marginal_cdf;
        int row = int(std::upper_bound(mc, mc + h + 1, uvy) - mc) - 1;
        uvy = clampf((uvy - mc[row])/sc.env.marginal_pdf[row], 0.0f, 1.0f);
        const float *rs = sc.env.cdf + size_t(row)*(w + 1);
        int column = int(std::upper_bound(rs, rs + w + 1, uvx) - rs) - 1;
        uvx = clampf((uvx - rs[column])/sc.env.pdf[size_t(row)*w + column], 0.0f, 1.0f);
        float ux = (uvx + column)/w;
        float uy = 1.0f - (uvy + row)/h;
        float sinTheta;
        out.d = uvToDirection(ux, uy, sinTheta);
        out.pdf = INV_PI*INV_TWO_PI*envPdf(ux, uy)/sinTheta;
        out.dist = INF;
        return out.pdf != 0.0f;
    }
    if (l.kind == TGPU_LIGHT_SPHERE) { // Sphere.cpp:173-194
        const tgpu_sphere &sp = sc.spheres[l.index];
        V3 L = V3(sp.pos) - p;
        float d = L.length();
        float C = d*d - sp.radius*sp.radius;
        if (C <= 0.0f) return false;
        L = L.normalized(); // L.normalize(): *= 1/length
        float cosTheta = std::sqrt(C)/d;
        float xi0 = s.next1D(), xi1 = s.next1D();
        float phi = xi0*TWO_PI; // SampleWarp::uniformSphericalCap (SampleWarp.hpp:119-129)
        float z = xi1*(1.0f - cosTheta) + cosTheta;
        float r = std::sqrt(fmax2(1.0f - z*z, 0.0f));
        V3 dl(mcos(phi)*r, msin(phi)*r, z);
        float B = d*dl.z;
        float det = std::sqrt(fmax2(B*B - C, 0.0f));
        out.dist = B - det;
        out.d = duffFrame(L).toGlobal(dl);
        out.pdf = INV_TWO_PI/(1.0f - cosTheta); // uniformSphericalCapPdf
        return true;
    }
    if (l.kind == TGPU_LIGHT_DISK) { // Disk.cpp:177-193
        const tgpu_disk &dk = sc.disks[l.index];
        V3 n(dk.n), center(dk.center);
        if (n.dot(p - center) < 0.0f) return false;
        float xi0 = s.next1D(), xi1 = s.next1D();
        float phi = xi0*TWO_PI; // SampleWarp::uniformDisk (:64-69)
        float rr = std::sqrt(xi1);
        float lqx = (mcos(phi)*rr)*dk.r, lqy = (msin(phi)*rr)*dk.r;
        V3 q = center + lqx*V3(dk.bitangent) + lqy*V3(dk.tangent);
        out.d = q - p;
        float rSq = out.d.lengthSq();
        out.dist = std::sqrt(rSq);
        out.d /= out.dist;
        if (-out.d.dot(n) < dk.cos_apex) return false;
        float cosTheta = -n.dot(out.d);
        out.pdf = rSq/(cosTheta*dk.r*dk.r*PI);
        return true;
    }
    // Quad.cpp:172-187
    const tgpu_quad &q = sc.quads[l.index];
    V3 normal(q.normal), base(q.base);
    if (normal.dot(p - base) <= 0.0f) return false;
    float xi0 = s.next1D(), xi1 = s.next1D();
    V3 qp = base + xi0*V3(q.edge0) + xi1*V3(q.edge1);
    out.d = qp - p;
    float rSq = out.d.lengthSq();
    out.dist = std::sqrt(rSq);
    out.d /= out.dist;
    float cosTheta = -normal.dot(out.d);
    out.pdf = rSq/(cosTheta*q.area);
    return true;
}

bool oracle_ctx::lightIntersect(const tgpu_light &l, Ray &ray, Temp &data) const
{
    if (l.kind == TGPU_LIGHT_ENV) { // InfiniteSphere.cpp:77-84 (farT is left untouched)
        data.kind = PRIM_ENV;
        data.index = 0;
        data.p = ray.pos;
        data.w = ray.dir;
        return true;
    }
    if (l.kind == TGPU_LIGHT_SPHERE) return sphereIntersect(sc.spheres[l.index], uint32_t(l.index), ray, data);
    if (l.kind == TGPU_LIGHT_DISK) return diskIntersect(sc.disks[l.index], uint32_t(l.index), ray, data);
    return quadIntersect(sc.quads[l.index], uint32_t(l.index), ray, data);
}

float oracle_ctx::lightDirectPdf(const tgpu_light &l, const Temp &data, const Info &info, const V3 &p) const
{
    if (l.kind == TGPU_LIGHT_ENV) { // InfiniteSphere.cpp:218-229
        if (sc.textures[sc.env.emission_tex].type == TGPU_TEX_CONSTANT) return INV_FOUR_PI;
        float sinTheta, ux, uy;
        directionToUV(data.w, ux, uy, &sinTheta);
        return INV_PI*INV_TWO_PI*envPdf(ux, uy)/sinTheta;
    }
    if (l.kind == TGPU_LIGHT_SPHERE) { // Sphere.cpp:224-230
        const tgpu_sphere &sp = sc.spheres[l.index];
        float dist = (V3(sp.pos) - p).length();
        float cosTheta = std::sqrt(fmax2(dist*dist - sp.radius*sp.radius, 0.0f))/dist;
        return INV_TWO_PI/(1.0f - cosTheta);
    }
    if (l.kind == TGPU_LIGHT_DISK) { // Disk.cpp:225-232
        const tgpu_disk &dk = sc.disks[l.index];
        V3 n(dk.n);
        float cosTheta = std::abs(n.dot(info.w));
        float t = n.dot(V3(dk.center) - p)/n.dot(info.w);
        return t*t/(cosTheta*dk.r*dk.r*PI);
    }
    const tgpu_quad &q = sc.quads[l.index]; // Quad.cpp:216-223
    V3 normal(q.normal);
    float cosTheta = std::abs(normal.dot(info.w));
    float t = normal.dot(V3(q.base) - p)/normal.dot(info.w);
    return t*t/(cosTheta*q.area);
}

float oracle_ctx::lightApproximateRadiance(const tgpu_light &l, const V3 &p) const
{
    auto avgMax = [&](int tex) { // Texture::average().max(): ConstantTexture.cpp:35-38, CheckerTexture.cpp:49-52, BitmapTexture.cpp:283-286
        const tgpu_texture &t = sc.textures[tex];
        V3 a;
        if (t.type == TGPU_TEX_CONSTANT) a = V3(t.value);
        else if (t.type == TGPU_TEX_CHECKER) a = (V3(t.on_color) + V3(t.off_color))*0.5f;
        else a = sc.bitmaps[t.bitmap].scale*V3(sc.bitmaps[t.bitmap].average);
        return a.max();
    };
    if (l.kind == TGPU_LIGHT_ENV) return TWO_PI*avgMax(sc.env.emission_tex); // InfiniteSphere.cpp:261-266 (listed lights are samplable)
    if (l.kind == TGPU_LIGHT_SPHERE) { // Sphere.cpp:33-40,266-271
        const tgpu_sphere &sp = sc.spheres[l.index];
        V3 L = V3(sp.pos) - p;
        float d = L.length();
        float cosTheta = std::sqrt(fmax2(d*d - sp.radius*sp.radius, 0.0f))/d;
        return (TWO_PI*(1.0f - cosTheta))*avgMax(sp.emission_tex);
    }
    if (l.kind == TGPU_LIGHT_DISK) { // Disk.cpp:268-296
        const tgpu_disk &dk = sc.disks[l.index];
        V3 n(dk.n);
        V3 coneD = p - V3(dk.cone_base);
        if (coneD.dot(n)/coneD.length() < dk.cos_apex) return 0.0f;
        V3 d = V3(dk.center) - p;
        V3 edge0 = V3(dk.tangent)*dk.r, edge1 = V3(dk.bitangent)*dk.r;
        V3 R0 = d - edge0 - edge1;
        V3 R1 = R0 + 2.0f*edge0;
        V3 R2 = R1 + 2.0f*edge1;
        V3 R3 = R0 + 2.0f*edge1;
        V3 n0 = R0.cross(R1).normalized(), n1 = R1.cross(R2).normalized(), n2 = R2.cross(R3).normalized(), n3 = R3.cross(R0).normalized();
        float Q = macos(n0.dot(n1)) + macos(n1.dot(n2)) + macos(n2.dot(n3)) + macos(n3.dot(n0));
        return (TWO_PI - std::abs(Q))*avgMax(dk.emission_tex);
    }
    const tgpu_quad &q = sc.quads[l.index]; // Quad.cpp:256-279
    V3 R0 = V3(q.base) - p;
    if (R0.dot(V3(q.normal)) >= 0.0f) return 0.0f;
    V3 R1 = R0 + V3(q.edge0), R2 = R1 + V3(q.edge1), R3 = R0 + V3(q.edge1);
    V3 n0 = R0.cross(R1).normalized(), n1 = R1.cross(R2).normalized(), n2 = R2.cross(R3).normalized(), n3 = R3.cross(R0).normalized();
    float Q = macos(n0.dot(n1)) + macos(n1.dot(n2)) + macos(n2.dot(n3)) + macos(n3.dot(n0));
    return (TWO_PI - std::abs(Q))*avgMax(q.emission_tex);
}

// ========================================================================================
// BSDFs
// ========================================================================================
uint32_t oracle_ctx::bsdfLobes(int b) const
{
    if (b < 0) return 0;
    const tgpu_bsdf &d = sc.bsdfs[b];
    switch (d.type) {
    case TGPU_BSDF_LAMBERT: return DiffuseReflectionLobe;
    case TGPU_BSDF_ROUGH_CONDUCTOR: return GlossyReflectionLobe;
    case TGPU_BSDF_OREN_NAYAR: return DiffuseReflectionLobe;                            // OrenNayarBsdf.cpp:20
    case TGPU_BSDF_PHONG: return GlossyReflectionLobe | DiffuseReflectionLobe;          // PhongBsdf.cpp:126
    case TGPU_BSDF_SMOOTH_COAT: return SpecularReflectionLobe | bsdfLobes(d.substrate);
    case TGPU_BSDF_MIXED: return bsdfLobes(d.substrate) | bsdfLobes(d.aux_bsdf);         // MixedBsdf.cpp:133-136
    case TGPU_BSDF_ROUGH_COAT: return GlossyReflectionLobe | bsdfLobes(d.substrate);      // RoughCoatBsdf.cpp:304
    case TGPU_BSDF_TRANSPARENCY: return ForwardLobe | bsdfLobes(d.substrate);             // TransparencyBsdf.cpp:66-69
    case TGPU_BSDF_THINSHEET: return SpecularReflectionLobe | ForwardLobe;                // ThinSheetBsdf.cpp:24
    case TGPU_BSDF_FORWARD: return ForwardLobe; // ForwardBsdf.cpp:9-12
    case TGPU_BSDF_MIRROR: case TGPU_BSDF_CONDUCTOR: return SpecularReflectionLobe;
    case TGPU_BSDF_PLASTIC: return SpecularReflectionLobe | DiffuseReflectionLobe;     // PlasticBsdf.cpp:19-24
    case TGPU_BSDF_ROUGH_PLASTIC: return GlossyReflectionLobe | DiffuseReflectionLobe; // RoughPlasticBsdf.cpp:21-29
    case TGPU_BSDF_DIELECTRIC: // DielectricBsdf.cpp:181-188
        return d.enable_refraction ? (SpecularReflectionLobe | SpecularTransmissionLobe) : SpecularReflectionLobe;
    case TGPU_BSDF_ROUGH_DIELECTRIC: // RoughDielectricBsdf.cpp:38-41
        return d.enable_refraction ? (GlossyReflectionLobe | GlossyTransmissionLobe) : GlossyReflectionLobe;
    default: return 0;
    }
}

namespace {
// bsdfs/Microfacet.hpp:27-130 (Beckmann and GGX; Phong is outside the configs)
inline float roughnessToAlpha(float roughness) { return fmax2(roughness, 1e-3f); }
struct MF {
    const oracle_ctx *o;
    int dist;
    float D(float alpha, const V3 &m) const
    {
        if (m.z <= 0.0f) return 0.0f;
        float alphaSq = alpha*alpha;
        float cosThetaSq = m.z*m.z;
        float tanThetaSq = fmax2(1.0f - cosThetaSq, 0.0f)/cosThetaSq;
        float cosThetaQu = cosThetaSq*cosThetaSq;
        if (dist == TGPU_DIST_BECKMANN)
            return INV_PI*o->mexp(-tanThetaSq/alphaSq)/(alphaSq*cosThetaQu);
        float s = alphaSq + tanThetaSq;
        return alphaSq*INV_PI/(cosThetaQu*(s*s));
    }
    float G1(float alpha, const V3 &v, const V3 &m) const
    {
        if (v.dot(m)*v.z <= 0.0f) return 0.0f;
        if (dist == TGPU_DIST_BECKMANN) {
            float cosThetaSq = v.z*v.z;
            float tanTheta = std::abs(std::sqrt(fmax2(1.0f - cosThetaSq, 0.0f))/v.z);
            float a = 1.0f/(alpha*tanTheta);
            if (a < 1.6f)
                return (3.535f*a + 2.181f*a*a)/(1.0f + 2.276f*a + 2.577f*a*a);
            else
                return 1.0f;
        }
        float alphaSq = alpha*alpha;
        float cosThetaSq = v.z*v.z;
        float tanThetaSq = fmax2(1.0f - cosThetaSq, 0.0f)/cosThetaSq;
        return 2.0f/(1.0f + std::sqrt(1.0f + alphaSq*tanThetaSq));
    }
    float G(float alpha, const V3 &i, const V3 &oo, const V3 &m) const { return G1(alpha, i, m)*G1(alpha, oo, m); }
    float pdf(float alpha, const V3 &m) const { return D(alpha, m)*m.z; }
    V3 sample(float alpha, float xix, float xiy) const
    {
        float phi = xiy*TWO_PI;
        float cosTheta = 0.0f;
        if (dist == TGPU_DIST_BECKMANN) {
            float tanThetaSq = -alpha*alpha*o->mlog(1.0f - xix);
            cosTheta = 1.0f/std::sqrt(1.0f + tanThetaSq);
        } else {
            float tanThetaSq = alpha*alpha*xix/(1.0f - xix);
            cosTheta = 1.0f/std::sqrt(1.0f + tanThetaSq);
        }
        float r = std::sqrt(fmax2(1.0f - cosTheta*cosTheta, 0.0f));
        return V3(o->mcos(phi)*r, o->msin(phi)*r, cosTheta);
    }
};

// bsdfs/Fresnel.hpp:102-125
inline float conductorReflectance(float eta, float k, float cosThetaI)
{
    float cosThetaISq = cosThetaI*cosThetaI;
    float sinThetaISq = fmax2(1.0f - cosThetaISq, 0.0f);
    float sinThetaIQu = sinThetaISq*sinThetaISq;
    float innerTerm = eta*eta - k*k - sinThetaISq;
    float aSqPlusBSq = std::sqrt(fmax2(innerTerm*innerTerm + 4.0f*eta*eta*k*k, 0.0f));
    float a = std::sqrt(fmax2((aSqPlusBSq + innerTerm)*0.5f, 0.0f));
    float Rs = ((aSqPlusBSq + cosThetaISq) - (2.0f*a*cosThetaI))/((aSqPlusBSq + cosThetaISq) + (2.0f*a*cosThetaI));
    float Rp = ((cosThetaISq*aSqPlusBSq + sinThetaIQu) - (2.0f*a*cosThetaI*sinThetaISq))/
               ((cosThetaISq*aSqPlusBSq + sinThetaIQu) + (2.0f*a*cosThetaI*sinThetaISq));
    return 0.5f*(Rs + Rs*Rp);
}
inline V3 conductorReflectance3(const float *eta, const float *k, float c)
{
    return V3(conductorReflectance(eta[0], k[0], c), conductorReflectance(eta[1], k[1], c), conductorReflectance(eta[2], k[2], c));
}
// bsdfs/Fresnel.hpp:75-92
inline float dielectricReflectance(float eta, float cosThetaI, float &cosThetaT)
{
    if (cosThetaI < 0.0f) {
        eta = 1.0f/eta;
        cosThetaI = -cosThetaI;
    }
    float sinThetaTSq = eta*eta*(1.0f - cosThetaI*cosThetaI);
    if (sinThetaTSq > 1.0f) {
        cosThetaT = 0.0f;
        return 1.0f;
    }
    cosThetaT = std::sqrt(fmax2(1.0f - sinThetaTSq, 0.0f));
    float Rs = (eta*cosThetaI - cosThetaT)/(eta*cosThetaI + cosThetaT);
    float Rp = (eta*cosThetaT - cosThetaI)/(eta*cosThetaT + cosThetaI);
    return (Rs*Rs + Rp*Rp)*0.5f;
}
inline float thinFilmReflectance(float eta, float cosThetaI, float &cosThetaT) // bsdfs/Fresnel.hpp:15-28
{
    float sinThetaTSq = eta*eta*(1.0f - cosThetaI*cosThetaI);
    if (sinThetaTSq > 1.0f) {
        cosThetaT = 0.0f;
        return 1.0f;
    }
    cosThetaT = std::sqrt(fmax2(1.0f - sinThetaTSq, 0.0f));
    float a = (eta*cosThetaI - cosThetaT)/(eta*cosThetaI + cosThetaT), b = (eta*cosThetaT - cosThetaI)/(eta*cosThetaT + cosThetaI);
    float Rs = a*a, Rp = b*b;
    return 1.0f - ((1.0f - Rs)/(1.0f + Rs) + (1.0f - Rp)/(1.0f + Rp))*0.5f;
}
inline float sgnE(float v) { return v < 0.0f ? -1.0f : 1.0f; } // RoughDielectricBsdf.cpp:13-17
// Bsdf::eval/sample(event, adjoint = false) multiply by sqr(eta(event)) (Bsdf.hpp:71-97, RoughDielectricBsdf.cpp:266-272)
inline float dielectricEtaSq(float ior, const V3 &wi, const V3 &wo)
{
    float eta = wi.z*wo.z >= 0.0f ? 1.0f : (wi.z < 0.0f ? ior : 1.0f/ior);
    return eta*eta;
}
inline bool checkReflectionConstraint(const V3 &wi, const V3 &wo) // bsdfs/Bsdf.hpp:44-47
{
    return std::abs(wi.z*wo.z - wi.x*wo.x - wi.y*wo.y - 1.0f) < 1e-3f;
}
inline bool checkRefractionConstraint(const V3 &wi, const V3 &wo, float eta, float cosThetaT) // bsdfs/Bsdf.hpp:50-54
{
    float dotP = -wi.x*wo.x*eta - wi.y*wo.y*eta - std::copysign(cosThetaT, wi.z)*wo.z;
    return std::abs(dotP - 1.0f) < 1e-3f;
}
inline Event warped(const Event &e, const V3 &wi, const V3 &wo) { Event c = e; c.wi = wi; c.wo = wo; return c; }
} // namespace

bool oracle_ctx::rdSampleBase(Event &e, bool sampleR, bool sampleT, float roughness, float ior, int distribution) const
{   // bsdfs/RoughDielectricBsdf.cpp:55-131
    MF mf{this, distribution};
    float wiDotN = e.wi.z;
    float eta = wiDotN < 0.0f ? ior : 1.0f/ior;
    float sampleRoughness = (1.2f - 0.2f*std::sqrt(std::abs(wiDotN)))*roughness;
    float alpha = roughnessToAlpha(roughness);
    float sampleAlpha = roughnessToAlpha(sampleRoughness);
    float xix = e.sampler->next1D(), xiy = e.sampler->next1D();
    V3 m = mf.sample(sampleAlpha, xix, xiy);
    float pm = mf.pdf(sampleAlpha, m);
    if (pm < 1e-10f) return false;
    float wiDotM = e.wi.dot(m);
    float cosThetaT = 0.0f;
    float F = dielectricReflectance(1.0f/ior, wiDotM, cosThetaT);
    float etaM = wiDotM < 0.0f ? ior : 1.0f/ior;
    bool reflect;
    if (sampleR && sampleT) {
        reflect = e.sampler->nextBoolean(F);
    } else if (sampleT) {
        if (F == 1.0f) return false;
        reflect = false;
    } else if (sampleR) {
        reflect = true;
    } else {
        return false;
    }
    if (reflect) e.wo = 2.0f*wiDotM*m - e.wi;
    else e.wo = (etaM*wiDotM - sgnE(wiDotM)*cosThetaT)*m - etaM*e.wi;
    float woDotN = e.wo.z;
    bool reflected = wiDotN*woDotN > 0.0f;
    if (reflected != reflect) return false;
    float woDotM = e.wo.dot(m);
    float G = mf.G(alpha, e.wi, e.wo, m);
    float D = mf.D(alpha, m);
    e.weight = V3(std::abs(wiDotM)*G*D/(std::abs(wiDotN)*pm));
    if (reflect) {
        e.pdf = pm*0.25f/std::abs(wiDotM);
        e.sampledLobe = GlossyReflectionLobe;
    } else {
        float q = eta*wiDotM + woDotM;
        e.pdf = pm*std::abs(woDotM)/(q*q);
        e.sampledLobe = GlossyTransmissionLobe;
    }
    if (sampleR && sampleT) {
        if (reflect) e.pdf *= F; else e.pdf *= 1.0f - F;
    } else {
        if (reflect) e.weight *= F; else e.weight *= 1.0f - F;
    }
    return true;
}

V3 oracle_ctx::rdEvalBase(const Event &e, bool sampleR, bool sampleT, float roughness, float ior, int distribution) const
{   // bsdfs/RoughDielectricBsdf.cpp:133-164
    float wiDotN = e.wi.z, woDotN = e.wo.z;
    bool reflect = wiDotN*woDotN >= 0.0f;
    if ((reflect && !sampleR) || (!reflect && !sampleT)) return V3(0.0f);
    MF mf{this, distribution};
    float alpha = roughnessToAlpha(roughness);
    float eta = wiDotN < 0.0f ? ior : 1.0f/ior;
    V3 m;
    if (reflect) m = sgnE(wiDotN)*(e.wi + e.wo).normalized();
    else m = -(e.wi*eta + e.wo).normalized();
    float wiDotM = e.wi.dot(m), woDotM = e.wo.dot(m);
    float cosThetaT;
    float F = dielectricReflectance(1.0f/ior, wiDotM, cosThetaT);
    float G = mf.G(alpha, e.wi, e.wo, m);
    float D = mf.D(alpha, m);
    if (reflect) {
        float fr = (F*G*D*0.25f)/std::abs(wiDotN);
        return V3(fr);
    }
    float q = eta*wiDotM + woDotM;
    float fs = std::abs(wiDotM*woDotM)*(1.0f - F)*G*D/((q*q)*std::abs(wiDotN));
    return V3(fs);
}

float oracle_ctx::rdPdfBase(const Event &e, bool sampleR, bool sampleT, float roughness, float ior, int distribution) const
{   // bsdfs/RoughDielectricBsdf.cpp:198-234
    float wiDotN = e.wi.z, woDotN = e.wo.z;
    bool reflect = wiDotN*woDotN >= 0.0f;
    if ((reflect && !sampleR) || (!reflect && !sampleT)) return 0.0f;
    MF mf{this, distribution};
    float sampleRoughness = (1.2f - 0.2f*std::sqrt(std::abs(wiDotN)))*roughness;
    float sampleAlpha = roughnessToAlpha(sampleRoughness);
    float eta = wiDotN < 0.0f ? ior : 1.0f/ior;
    V3 m;
    if (reflect) m = sgnE(wiDotN)*(e.wi + e.wo).normalized();
    else m = -(e.wi*eta + e.wo).normalized();
    float wiDotM = e.wi.dot(m), woDotM = e.wo.dot(m);
    float cosThetaT;
    float F = dielectricReflectance(1.0f/ior, wiDotM, cosThetaT);
    float pm = mf.pdf(sampleAlpha, m);
    float pdf;
    if (reflect) {
        pdf = pm*0.25f/std::abs(wiDotM);
    } else {
        float q = eta*wiDotM + woDotM;
        pdf = pm*std::abs(woDotM)/(q*q);
    }
    if (sampleR && sampleT) {
        if (reflect) pdf *= F; else pdf *= 1.0f - F;
    }
    return pdf;
}

V3 oracle_ctx::bsdfEval(int b, const Event &e) const
{
    if (b < 0) return V3(0.0f);
    const tgpu_bsdf &d = sc.bsdfs[b];
    switch (d.type) {
    case TGPU_BSDF_LAMBERT: { // bsdfs/LambertBsdf.cpp:40-47
        if (!(e.requestedLobe & DiffuseReflectionLobe)) return V3(0.0f);
        if (e.wi.z <= 0.0f || e.wo.z <= 0.0f) return V3(0.0f);
        return texEval(d.albedo_tex, e.info->uvx, e.info->uvy)*INV_PI*e.wo.z;
    }
    case TGPU_BSDF_OREN_NAYAR: { // bsdfs/OrenNayarBsdf.cpp:63-98
        if (!(e.requestedLobe & DiffuseReflectionLobe)) return V3(0.0f);
        if (e.wi.z <= 0.0f || e.wo.z <= 0.0f) return V3(0.0f);
        const V3 &wi = e.wi, &wo = e.wo;
        float thetaR = macos(wo.z);
        float thetaI = macos(wi.z);
        float alpha = fmax2(thetaR, thetaI);
        float beta = fmin2(thetaR, thetaI);
        float sinAlpha = msin(alpha);
        float denom = (wi.x*wi.x + wi.y*wi.y)*(wo.x*wo.x + wo.y*wo.y);
        float cosDeltaPhi = denom == 0.0f ? 1.0f : (wi.x*wo.x + wi.y*wo.y)/std::sqrt(denom);
        const float RoughnessToSigma = 1.0f/std::sqrt(2.0f);
        float sigma = RoughnessToSigma*texEval(d.roughness_tex, e.info->uvx, e.info->uvy).x;
        float sigmaSq = sigma*sigma;
        float C1 = 1.0f - 0.5f*sigmaSq/(sigmaSq + 0.33f);
        float C2 = 0.45f*sigmaSq/(sigmaSq + 0.09f);
        auto cube = [](float x) { return x*x*x; };
        auto sqr = [](float x) { return x*x; };
        if (cosDeltaPhi >= 0.0f) C2 *= sinAlpha;
        else C2 *= sinAlpha - cube((2.0f*INV_PI)*beta);
        float C3 = 0.125f*(sigmaSq/(sigmaSq + 0.09f))*sqr((4.0f*INV_PI*INV_PI)*alpha*beta);
        float fr1 = (C1 + cosDeltaPhi*C2*mtan(beta) + (1.0f - std::abs(cosDeltaPhi))*C3*mtan(0.5f*(alpha + beta)));
        float fr2 = 0.17f*sigmaSq/(sigmaSq + 0.13f)*(1.0f - cosDeltaPhi*sqr((2.0f*INV_PI)*beta));
        V3 diffuseAlbedo = texEval(d.albedo_tex, e.info->uvx, e.info->uvy);
        return (diffuseAlbedo*fr1 + diffuseAlbedo*diffuseAlbedo*fr2)*wo.z*INV_PI;
    }
    case TGPU_BSDF_PHONG: { // bsdfs/PhongBsdf.cpp:79-100 ; factors of prepareForRender :123-131
        bool evalGlossy = (e.requestedLobe & GlossyReflectionLobe) != 0, evalDiffuse = (e.requestedLobe & DiffuseReflectionLobe) != 0;
        if (!evalGlossy && !evalDiffuse) return V3(0.0f);
        if (e.wi.z <= 0.0f || e.wo.z <= 0.0f) return V3(0.0f);
        const float brdfFactor = (d.exponent + 2.0f)*INV_TWO_PI;
        float result = 0.0f;
        if (evalDiffuse) result += d.diffuse_ratio*INV_PI;
        if (evalGlossy) {
            float cosTheta = V3(-e.wi.x, -e.wi.y, e.wi.z).dot(e.wo);
            if (cosTheta > 0.0f) result += mpow(cosTheta, d.exponent)*brdfFactor*(1.0f - d.diffuse_ratio);
        }
        return texEval(d.albedo_tex, e.info->uvx, e.info->uvy)*e.wo.z*result;
    }
    case TGPU_BSDF_ROUGH_CONDUCTOR: { // bsdfs/RoughConductorBsdf.cpp:93-110
        if (!(e.requestedLobe & GlossyReflectionLobe)) return V3(0.0f);
        if (e.wi.z <= 0.0f || e.wo.z <= 0.0f) return V3(0.0f);
        MF mf{this, d.distribution};
        float roughness = texEval(d.roughness_tex, e.info->uvx, e.info->uvy).x;
        float alpha = roughnessToAlpha(roughness);
        V3 hr = (e.wi + e.wo).normalized();
        float cosThetaM = e.wi.dot(hr);
        V3 F = conductorReflectance3(d.eta, d.k, cosThetaM);
        float G = mf.G(alpha, e.wi, e.wo, hr);
        float D = mf.D(alpha, hr);
        float fr = (G*D*0.25f)/e.wi.z;
        return texEval(d.albedo_tex, e.info->uvx, e.info->uvy)*(F*fr);
    }
    case TGPU_BSDF_PLASTIC: { // bsdfs/PlasticBsdf.cpp:139-163
        if (e.wi.z <= 0.0f || e.wo.z <= 0.0f) return V3(0.0f);
        bool evalR = (e.requestedLobe & SpecularReflectionLobe) != 0;
        bool evalT = (e.requestedLobe & DiffuseReflectionLobe) != 0;
        float eta = 1.0f/d.ior;
        float ct;
        float Fi = dielectricReflectance(eta, e.wi.z, ct);
        float Fo = dielectricReflectance(eta, e.wo.z, ct);
        if (evalR && checkReflectionConstraint(e.wi, e.wo)) return V3(Fi);
        if (evalT) {
            V3 diffuseAlbedo = texEval(d.albedo_tex, e.info->uvx, e.info->uvy);
            V3 brdf = ((1.0f - Fi)*(1.0f - Fo)*eta*eta*e.wo.z*INV_PI)*(diffuseAlbedo/(V3(1.0f) - diffuseAlbedo*d.diffuse_fresnel));
            V3 ssa(d.scaled_sigma_a);
            if (ssa.max() > 0.0f) { V3 a = ssa*(-1.0f/e.wo.z - 1.0f/e.wi.z); brdf *= V3(mexp(a.x), mexp(a.y), mexp(a.z)); }
            return brdf;
        }
        return V3(0.0f);
    }
    case TGPU_BSDF_ROUGH_PLASTIC: { // bsdfs/RoughPlasticBsdf.cpp:128-157
        bool sampleR = (e.requestedLobe & GlossyReflectionLobe) != 0;
        bool sampleT = (e.requestedLobe & DiffuseReflectionLobe) != 0;
        if (!sampleR && !sampleT) return V3(0.0f);
        if (e.wi.z <= 0.0f || e.wo.z <= 0.0f) return V3(0.0f);
        float roughness = texEval(d.roughness_tex, e.info->uvx, e.info->uvy).x;
        V3 glossyR(0.0f);
        if (sampleR) glossyR = rdEvalBase(e, true, false, roughness, d.ior, d.distribution);
        V3 diffuseR(0.0f);
        if (sampleT) {
            float eta = 1.0f/d.ior;
            float ct;
            float Fi = dielectricReflectance(eta, e.wi.z, ct);
            float Fo = dielectricReflectance(eta, e.wo.z, ct);
            V3 diffuseAlbedo = texEval(d.albedo_tex, e.info->uvx, e.info->uvy);
            diffuseR = ((1.0f - Fi)*(1.0f - Fo)*eta*eta*e.wo.z*INV_PI)*(diffuseAlbedo/(V3(1.0f) - diffuseAlbedo*d.diffuse_fresnel));
            V3 ssa(d.scaled_sigma_a);
            if (ssa.max() > 0.0f) { V3 a = ssa*(-1.0f/e.wo.z - 1.0f/e.wi.z); diffuseR *= V3(mexp(a.x), mexp(a.y), mexp(a.z)); }
        }
        return glossyR + diffuseR;
    }
    case TGPU_BSDF_MIRROR: { // bsdfs/MirrorBsdf.cpp:40-47
        if ((e.requestedLobe & SpecularReflectionLobe) && checkReflectionConstraint(e.wi, e.wo)) return texEval(d.albedo_tex, e.info->uvx, e.info->uvy);
        return V3(0.0f);
    }
    case TGPU_BSDF_CONDUCTOR: { // bsdfs/ConductorBsdf.cpp:66-73
        if ((e.requestedLobe & SpecularReflectionLobe) && checkReflectionConstraint(e.wi, e.wo))
            return texEval(d.albedo_tex, e.info->uvx, e.info->uvy)*conductorReflectance3(d.eta, d.k, e.wi.z);
        return V3(0.0f);
    }
    case TGPU_BSDF_DIELECTRIC: { // bsdfs/DielectricBsdf.cpp:92-112 ; eval(event, false) *= sqr(eta)
        bool evalR = (e.requestedLobe & SpecularReflectionLobe) != 0;
        bool evalT = (e.requestedLobe & SpecularTransmissionLobe) != 0 && d.enable_refraction;
        float eta = e.wi.z < 0.0f ? d.ior : 1.0f/d.ior;
        float cosThetaT = 0.0f;
        float F = dielectricReflectance(eta, std::abs(e.wi.z), cosThetaT);
        V3 f(0.0f);
        if (e.wi.z*e.wo.z >= 0.0f) {
            if (evalR && checkReflectionConstraint(e.wi, e.wo)) f = F*texEval(d.albedo_tex, e.info->uvx, e.info->uvy);
        } else {
            if (evalT && checkRefractionConstraint(e.wi, e.wo, eta, cosThetaT)) f = (1.0f - F)*texEval(d.albedo_tex, e.info->uvx, e.info->uvy);
        }
        f *= dielectricEtaSq(d.ior, e.wi, e.wo);
        return f;
    }
    case TGPU_BSDF_FORWARD: { // bsdfs/ForwardBsdf.cpp:26-29
        bool isForward = (e.requestedLobe & ForwardLobe) != 0; // BsdfLobes::isForward
        V3 nwi = -e.wi;
        return (isForward && nwi.x == e.wo.x && nwi.y == e.wo.y && nwi.z == e.wo.z) ? V3(1.0f) : V3(0.0f);
    }
    case TGPU_BSDF_ROUGH_DIELECTRIC: { // bsdfs/RoughDielectricBsdf.cpp:133-164 (evalBase), :242-249
        bool sampleR = (e.requestedLobe & GlossyReflectionLobe) != 0;
        bool sampleT = (e.requestedLobe & GlossyTransmissionLobe) != 0 && d.enable_refraction;
        float roughness = texEval(d.roughness_tex, e.info->uvx, e.info->uvy).x;
        float wiDotN = e.wi.z, woDotN = e.wo.z;
        bool reflect = wiDotN*woDotN >= 0.0f;
        if ((reflect && !sampleR) || (!reflect && !sampleT)) return V3(0.0f);
        MF mf{this, d.distribution};
        float alpha = roughnessToAlpha(roughness);
        float eta = wiDotN < 0.0f ? d.ior : 1.0f/d.ior;
        V3 m;
        if (reflect) m = sgnE(wiDotN)*(e.wi + e.wo).normalized();
        else m = -(e.wi*eta + e.wo).normalized();
        float wiDotM = e.wi.dot(m), woDotM = e.wo.dot(m);
        float cosThetaT;
        float F = dielectricReflectance(1.0f/d.ior, wiDotM, cosThetaT);
        float G = mf.G(alpha, e.wi, e.wo, m);
        float D = mf.D(alpha, m);
        V3 f;
        if (reflect) {
            float fr = (F*G*D*0.25f)/std::abs(wiDotN);
            f = V3(fr);
        } else {
            float q = eta*wiDotM + woDotM;
            float fs = std::abs(wiDotM*woDotM)*(1.0f - F)*G*D/((q*q)*std::abs(wiDotN));
            f = V3(fs);
        }
        f = f*texEval(d.albedo_tex, e.info->uvx, e.info->uvy);
        f *= dielectricEtaSq(d.ior, e.wi, e.wo);
        return f;
    }
    case TGPU_BSDF_MIXED: { // bsdfs/MixedBsdf.cpp:104-108
        float ratio = texEval(d.aux_tex, e.info->uvx, e.info->uvy).x;
        return texEval(d.albedo_tex, e.info->uvx, e.info->uvy)*(bsdfEval(d.substrate, e)*ratio + bsdfEval(d.aux_bsdf, e)*(1.0f - ratio));
    }
    case TGPU_BSDF_TRANSPARENCY: { // bsdfs/TransparencyBsdf.cpp:47-53 ; BsdfLobes::isForward(): exactly the forward lobe
        if (e.requestedLobe == uint32_t(ForwardLobe)) {
            V3 nwi = -e.wi;
            return (nwi.x == e.wo.x && nwi.y == e.wo.y && nwi.z == e.wo.z) ? V3(1.0f - texEval(d.aux_tex, e.info->uvx, e.info->uvy).x) : V3(0.0f);
        }
        return bsdfEval(d.substrate, e);
    }
    case TGPU_BSDF_THINSHEET: { // bsdfs/ThinSheetBsdf.cpp:83-103 (enable_interference off), Fresnel::thinFilmReflectance (Fresnel.hpp:15-28)
        V3 nwi = -e.wi;
        if (e.requestedLobe != uint32_t(ForwardLobe) || !(nwi.x == e.wo.x && nwi.y == e.wo.y && nwi.z == e.wo.z)) return V3(0.0f);
        float thickness = texEval(d.aux_tex, e.info->uvx, e.info->uvy).x;
        float cosThetaT;
        V3 transmittance(1.0f - thinFilmReflectance(1.0f/d.ior, std::abs(e.wi.z), cosThetaT));
        V3 sigmaA(d.sigma_a);
        if (!(sigmaA == 0.0f) && cosThetaT > 0.0f) {
            V3 a = -sigmaA*(thickness*2.0f/cosThetaT);
            transmittance *= V3(mexp(a.x), mexp(a.y), mexp(a.z));
        }
        return transmittance;
    }
    case TGPU_BSDF_ROUGH_COAT: { // bsdfs/RoughCoatBsdf.cpp:159-198
        bool sampleR = (e.requestedLobe & GlossyReflectionLobe) != 0;
        bool sampleT = (e.requestedLobe & bsdfLobes(d.substrate)) != 0;
        if (!sampleT && !sampleR) return V3(0.0f);
        if (e.wi.z <= 0.0f || e.wo.z <= 0.0f) return V3(0.0f);
        float roughness = texEval(d.roughness_tex, e.info->uvx, e.info->uvy).x;
        V3 glossyR(0.0f);
        if (sampleR) glossyR = rdEvalBase(e, true, false, roughness, d.ior, d.distribution);
        V3 substrateR(0.0f);
        if (sampleT) {
            const V3 &wi = e.wi, &wo = e.wo;
            float eta = 1.0f/d.ior;
            float cosThetaTi, cosThetaTo;
            float Fi = dielectricReflectance(eta, wi.z, cosThetaTi);
            float Fo = dielectricReflectance(eta, wo.z, cosThetaTo);
            if (Fi == 1.0f || Fo == 1.0f) return glossyR;
            V3 wiSubstrate(wi.x*eta, wi.y*eta, std::copysign(cosThetaTi, wi.z));
            V3 woSubstrate(wo.x*eta, wo.y*eta, std::copysign(cosThetaTo, wo.z));
            float compressionProjection = eta*eta*wo.z/cosThetaTo;
            V3 substrateF = bsdfEval(d.substrate, warped(e, wiSubstrate, woSubstrate));
            V3 ssa(d.scaled_sigma_a);
            if (ssa.max() > 0.0f) { V3 a = ssa*(-1.0f/cosThetaTo - 1.0f/cosThetaTi); substrateF *= V3(mexp(a.x), mexp(a.y), mexp(a.z)); }
            substrateR = compressionProjection*(1.0f - Fi)*(1.0f - Fo)*substrateF;
        }
        return glossyR + substrateR;
    }
    case TGPU_BSDF_SMOOTH_COAT: { // bsdfs/SmoothCoatBsdf.cpp:146-178
        if (e.wi.z <= 0.0f || e.wo.z <= 0.0f) return V3(0.0f);
        uint32_t subLobes = bsdfLobes(d.substrate);
        bool evalR = (e.requestedLobe & SpecularReflectionLobe) != 0;
        bool evalT = (e.requestedLobe & subLobes) != 0;
        const V3 &wi = e.wi, &wo = e.wo;
        float eta = 1.0f/d.ior;
        float cosThetaTi, cosThetaTo;
        float Fi = dielectricReflectance(eta, wi.z, cosThetaTi);
        float Fo = dielectricReflectance(eta, wo.z, cosThetaTo);
        if (evalR && checkReflectionConstraint(e.wi, e.wo)) {
            return V3(Fi);
        } else if (evalT) {
            V3 wiSubstrate(wi.x*eta, wi.y*eta, std::copysign(cosThetaTi, wi.z));
            V3 woSubstrate(wo.x*eta, wo.y*eta, std::copysign(cosThetaTo, wo.z));
            float laplacian = eta*eta*wo.z/cosThetaTo;
            V3 substrateF = bsdfEval(d.substrate, warped(e, wiSubstrate, woSubstrate));
            V3 ss(d.scaled_sigma_a);
            if (ss.max() > 0.0f) {
                V3 a = ss*(-1.0f/cosThetaTo - 1.0f/cosThetaTi);
                substrateF *= V3(mexp(a.x), mexp(a.y), mexp(a.z));
            }
            return laplacian*(1.0f - Fi)*(1.0f - Fo)*substrateF;
        }
        return V3(0.0f);
    }
    default: return V3(0.0f);
    }
}

bool oracle_ctx::bsdfSample(int b, Event &e) const
{
    if (b < 0) return false;
    const tgpu_bsdf &d = sc.bsdfs[b];
    switch (d.type) {
    case TGPU_BSDF_LAMBERT: { // bsdfs/LambertBsdf.cpp:27-38
        if (!(e.requestedLobe & DiffuseReflectionLobe)) return false;
        if (e.wi.z <= 0.0f) return false;
        float xix = e.sampler->next1D(), xiy = e.sampler->next1D();
        float phi = xix*TWO_PI; // SampleWarp::cosineHemisphere (sampling/SampleWarp.hpp:42-52)
        float r = std::sqrt(xiy);
        e.wo = V3(mcos(phi)*r, msin(phi)*r, std::sqrt(fmax2(1.0f - xiy, 0.0f)));
        e.pdf = std::abs(e.wo.z)*INV_PI;
        e.weight = texEval(d.albedo_tex, e.info->uvx, e.info->uvy);
        e.sampledLobe = DiffuseReflectionLobe;
        return true;
    }
    case TGPU_BSDF_OREN_NAYAR: { // bsdfs/OrenNayarBsdf.cpp:41-61
        if (!(e.requestedLobe & DiffuseReflectionLobe)) return false;
        if (e.wi.z <= 0.0f) return false;
        float roughness = texEval(d.roughness_tex, e.info->uvx, e.info->uvy).x;
        float ratio = clampf(roughness, 0.01f, 1.0f);
        if (e.sampler->nextBoolean(ratio)) {
            float xix = e.sampler->next1D(), xiy = e.sampler->next1D();
            float phi = TWO_PI*xix; // SampleWarp::uniformHemisphere (SampleWarp.hpp:25-30)
            float r = std::sqrt(fmax2(1.0f - xiy*xiy, 0.0f));
            e.wo = V3(mcos(phi)*r, msin(phi)*r, xiy);
        } else {
            float xix = e.sampler->next1D(), xiy = e.sampler->next1D();
            float phi = xix*TWO_PI;
            float r = std::sqrt(xiy);
            e.wo = V3(mcos(phi)*r, msin(phi)*r, std::sqrt(fmax2(1.0f - xiy, 0.0f)));
        }
        e.pdf = INV_TWO_PI*ratio + (std::abs(e.wo.z)*INV_PI)*(1.0f - ratio);
        e.weight = bsdfEval(b, e)/e.pdf;
        e.sampledLobe = DiffuseReflectionLobe;
        return e.wo.z > 0.0f;
    }
    case TGPU_BSDF_PHONG: { // bsdfs/PhongBsdf.cpp:39-77
        bool evalGlossy = (e.requestedLobe & GlossyReflectionLobe) != 0, evalDiffuse = (e.requestedLobe & DiffuseReflectionLobe) != 0;
        if (!evalGlossy && !evalDiffuse) return false;
        if (e.wi.z <= 0.0f) return false;
        bool sampleGlossy = (evalGlossy && evalDiffuse) ? e.sampler->nextBoolean(1.0f - d.diffuse_ratio) : evalGlossy;
        if (sampleGlossy) {
            float xix = e.sampler->next1D(), xiy = e.sampler->next1D();
            float phi = xix*TWO_PI;
            float cosTheta = mpow(xiy, 1.0f/(1.0f + d.exponent));
            float sinTheta = std::sqrt(fmax2(0.0f, 1.0f - cosTheta*cosTheta));
            V3 woLocal(mcos(phi)*sinTheta, msin(phi)*sinTheta, cosTheta);
            Frame lobe = duffFrame(V3(-e.wi.x, -e.wi.y, e.wi.z));
            e.wo = lobe.toGlobal(woLocal);
            if (e.wo.z < 0.0f) return false;
            e.sampledLobe = GlossyReflectionLobe;
        } else {
            float xix = e.sampler->next1D(), xiy = e.sampler->next1D();
            float phi = xix*TWO_PI;
            float r = std::sqrt(xiy);
            e.wo = V3(mcos(phi)*r, msin(phi)*r, std::sqrt(fmax2(1.0f - xiy, 0.0f)));
            e.sampledLobe = DiffuseReflectionLobe;
        }
        e.pdf = bsdfPdf(b, e);
        e.weight = bsdfEval(b, e)/e.pdf;
        return true;
    }
    case TGPU_BSDF_ROUGH_CONDUCTOR: { // bsdfs/RoughConductorBsdf.cpp:60-91
        if (!(e.requestedLobe & GlossyReflectionLobe)) return false;
        if (e.wi.z <= 0.0f) return false;
        MF mf{this, d.distribution};
        float roughness = texEval(d.roughness_tex, e.info->uvx, e.info->uvy).x;
        float sampleRoughness = roughness;
        float alpha = roughnessToAlpha(roughness);
        float sampleAlpha = roughnessToAlpha(sampleRoughness);
        float xix = e.sampler->next1D(), xiy = e.sampler->next1D();
        V3 m = mf.sample(sampleAlpha, xix, xiy);
        float wiDotM = e.wi.dot(m);
        e.wo = 2.0f*wiDotM*m - e.wi;
        if (wiDotM <= 0.0f || e.wo.z <= 0.0f) return false;
        float G = mf.G(alpha, e.wi, e.wo, m);
        float D = mf.D(alpha, m);
        float mPdf = mf.pdf(sampleAlpha, m);
        float pdf = mPdf*0.25f/wiDotM;
        float weight = wiDotM*G*D/(e.wi.z*mPdf);
        V3 F = conductorReflectance3(d.eta, d.k, wiDotM);
        e.pdf = pdf;
        e.weight = texEval(d.albedo_tex, e.info->uvx, e.info->uvy)*(F*weight);
        e.sampledLobe = GlossyReflectionLobe;
        return true;
    }
    case TGPU_BSDF_PLASTIC: { // bsdfs/PlasticBsdf.cpp:45-91
        if (e.wi.z <= 0.0f) return false;
        bool sampleR = (e.requestedLobe & SpecularReflectionLobe) != 0;
        bool sampleT = (e.requestedLobe & DiffuseReflectionLobe) != 0;
        const V3 wi = e.wi;
        float eta = 1.0f/d.ior;
        float ct;
        float Fi = dielectricReflectance(eta, wi.z, ct);
        float substrateWeight = d.avg_transmittance*(1.0f - Fi);
        float specularWeight = Fi;
        float specularProbability;
        if (sampleR && sampleT) specularProbability = specularWeight/(specularWeight + substrateWeight);
        else if (sampleR) specularProbability = 1.0f;
        else if (sampleT) specularProbability = 0.0f;
        else return false;
        if (sampleR && e.sampler->nextBoolean(specularProbability)) {
            e.wo = V3(-wi.x, -wi.y, wi.z);
            e.pdf = specularProbability;
            e.weight = V3(Fi/specularProbability);
            e.sampledLobe = SpecularReflectionLobe;
        } else {
            float xix = e.sampler->next1D(), xiy = e.sampler->next1D();
            float phi = xix*TWO_PI; // SampleWarp::cosineHemisphere
            float r = std::sqrt(xiy);
            V3 wo(mcos(phi)*r, msin(phi)*r, std::sqrt(fmax2(1.0f - xiy, 0.0f)));
            float Fo = dielectricReflectance(eta, wo.z, ct);
            V3 diffuseAlbedo = texEval(d.albedo_tex, e.info->uvx, e.info->uvy);
            e.wo = wo;
            e.weight = ((1.0f - Fi)*(1.0f - Fo)*eta*eta)*(diffuseAlbedo/(V3(1.0f) - diffuseAlbedo*d.diffuse_fresnel));
            V3 ssa(d.scaled_sigma_a);
            if (ssa.max() > 0.0f) { V3 a = ssa*(-1.0f/e.wo.z - 1.0f/e.wi.z); e.weight *= V3(mexp(a.x), mexp(a.y), mexp(a.z)); }
            e.pdf = (std::abs(e.wo.z)*INV_PI)*(1.0f - specularProbability);
            e.weight /= 1.0f - specularProbability;
            e.sampledLobe = DiffuseReflectionLobe;
        }
        return true;
    }
    case TGPU_BSDF_ROUGH_PLASTIC: { // bsdfs/RoughPlasticBsdf.cpp:55-126
        if (e.wi.z <= 0.0f) return false;
        bool sampleR = (e.requestedLobe & GlossyReflectionLobe) != 0;
        bool sampleT = (e.requestedLobe & DiffuseReflectionLobe) != 0;
        if (!sampleR && !sampleT) return false;
        const V3 wi = e.wi;
        float eta = 1.0f/d.ior;
        float ct;
        float Fi = dielectricReflectance(eta, wi.z, ct);
        float substrateWeight = d.substrate_weight*d.avg_transmittance*(1.0f - Fi);
        float specularWeight = Fi;
        float specularProbability = specularWeight/(specularWeight + substrateWeight);
        if (sampleR && (e.sampler->nextBoolean(specularProbability) || !sampleT)) {
            float roughness = texEval(d.roughness_tex, e.info->uvx, e.info->uvy).x;
            if (!rdSampleBase(e, true, false, roughness, d.ior, d.distribution)) return false;
            if (sampleT) {
                V3 diffuseAlbedo = texEval(d.albedo_tex, e.info->uvx, e.info->uvy);
                float Fo = dielectricReflectance(eta, e.wo.z, ct);
                V3 brdfSubstrate = ((1.0f - Fi)*(1.0f - Fo)*eta*eta)*(diffuseAlbedo/(V3(1.0f) - diffuseAlbedo*d.diffuse_fresnel))*INV_PI*e.wo.z;
                V3 brdfSpecular = e.weight*e.pdf;
                float pdfSubstrate = (std::abs(e.wo.z)*INV_PI)*(1.0f - specularProbability);
                float pdfSpecular = e.pdf*specularProbability;
                e.weight = (brdfSpecular + brdfSubstrate)/(pdfSpecular + pdfSubstrate);
                e.pdf = pdfSpecular + pdfSubstrate;
            }
            return true;
        }
        float xix = e.sampler->next1D(), xiy = e.sampler->next1D();
        float phi = xix*TWO_PI;
        float r = std::sqrt(xiy);
        V3 wo(mcos(phi)*r, msin(phi)*r, std::sqrt(fmax2(1.0f - xiy, 0.0f)));
        float Fo = dielectricReflectance(eta, wo.z, ct);
        V3 diffuseAlbedo = texEval(d.albedo_tex, e.info->uvx, e.info->uvy);
        e.wo = wo;
        e.weight = ((1.0f - Fi)*(1.0f - Fo)*eta*eta)*(diffuseAlbedo/(V3(1.0f) - diffuseAlbedo*d.diffuse_fresnel));
        V3 ssa(d.scaled_sigma_a);
        if (ssa.max() > 0.0f) { V3 a = ssa*(-1.0f/e.wo.z - 1.0f/e.wi.z); e.weight *= V3(mexp(a.x), mexp(a.y), mexp(a.z)); }
        e.pdf = std::abs(e.wo.z)*INV_PI;
        if (sampleR) {
            float roughness = texEval(d.roughness_tex, e.info->uvx, e.info->uvy).x;
            V3 brdfSubstrate = e.weight*e.pdf;
            float pdfSubstrate = e.pdf*(1.0f - specularProbability);
            V3 brdfSpecular = rdEvalBase(e, true, false, roughness, d.ior, d.distribution);
            float pdfSpecular = rdPdfBase(e, true, false, roughness, d.ior, d.distribution);
            pdfSpecular *= specularProbability;
            e.weight = (brdfSpecular + brdfSubstrate)/(pdfSpecular + pdfSubstrate);
            e.pdf = pdfSpecular + pdfSubstrate;
        }
        e.sampledLobe = DiffuseReflectionLobe;
        return true;
    }
    case TGPU_BSDF_MIRROR: { // bsdfs/MirrorBsdf.cpp:29-38
        if (!(e.requestedLobe & SpecularReflectionLobe)) return false;
        e.wo = V3(-e.wi.x, -e.wi.y, e.wi.z);
        e.pdf = 1.0f;
        e.sampledLobe = SpecularReflectionLobe;
        e.weight = texEval(d.albedo_tex, e.info->uvx, e.info->uvy);
        return true;
    }
    case TGPU_BSDF_CONDUCTOR: { // bsdfs/ConductorBsdf.cpp:53-64
        if (!(e.requestedLobe & SpecularReflectionLobe)) return false;
        e.wo = V3(-e.wi.x, -e.wi.y, e.wi.z);
        e.pdf = 1.0f;
        e.weight = texEval(d.albedo_tex, e.info->uvx, e.info->uvy)*conductorReflectance3(d.eta, d.k, e.wi.z);
        e.sampledLobe = SpecularReflectionLobe;
        return true;
    }
    case TGPU_BSDF_DIELECTRIC: { // bsdfs/DielectricBsdf.cpp:50-90 ; sample(event, false) *= sqr(eta)
        bool sampleR = (e.requestedLobe & SpecularReflectionLobe) != 0;
        bool sampleT = (e.requestedLobe & SpecularTransmissionLobe) != 0 && d.enable_refraction;
        float eta = e.wi.z < 0.0f ? d.ior : 1.0f/d.ior;
        float cosThetaT = 0.0f;
        float F = dielectricReflectance(eta, std::abs(e.wi.z), cosThetaT);
        float reflectionProbability;
        if (sampleR && sampleT) reflectionProbability = F;
        else if (sampleR) reflectionProbability = 1.0f;
        else if (sampleT) reflectionProbability = 0.0f;
        else return false;
        if (e.sampler->nextBoolean(reflectionProbability)) {
            e.wo = V3(-e.wi.x, -e.wi.y, e.wi.z);
            e.pdf = reflectionProbability;
            e.sampledLobe = SpecularReflectionLobe;
            e.weight = sampleT ? V3(1.0f) : V3(F);
        } else {
            if (F == 1.0f) return false;
            e.wo = V3(-e.wi.x*eta, -e.wi.y*eta, -std::copysign(cosThetaT, e.wi.z));
            e.pdf = 1.0f - reflectionProbability;
            e.sampledLobe = SpecularTransmissionLobe;
            e.weight = sampleR ? V3(1.0f) : V3(1.0f - F);
        }
        e.weight *= texEval(d.albedo_tex, e.info->uvx, e.info->uvy);
        e.weight *= dielectricEtaSq(d.ior, e.wi, e.wo);
        return true;
    }
    case TGPU_BSDF_ROUGH_DIELECTRIC: { // bsdfs/RoughDielectricBsdf.cpp:55-131 (sampleBase), :232-240
        bool sampleR = (e.requestedLobe & GlossyReflectionLobe) != 0;
        bool sampleT = (e.requestedLobe & GlossyTransmissionLobe) != 0 && d.enable_refraction;
        float roughness = texEval(d.roughness_tex, e.info->uvx, e.info->uvy).x;
        const float ior = d.ior;
        MF mf{this, d.distribution};
        float wiDotN = e.wi.z;
        float eta = wiDotN < 0.0f ? ior : 1.0f/ior;
        float sampleRoughness = (1.2f - 0.2f*std::sqrt(std::abs(wiDotN)))*roughness;
        float alpha = roughnessToAlpha(roughness);
        float sampleAlpha = roughnessToAlpha(sampleRoughness);
        float xix = e.sampler->next1D(), xiy = e.sampler->next1D();
        V3 m = mf.sample(sampleAlpha, xix, xiy);
        float pm = mf.pdf(sampleAlpha, m);
        if (pm < 1e-10f) return false;
        float wiDotM = e.wi.dot(m);
        float cosThetaT = 0.0f;
        float F = dielectricReflectance(1.0f/ior, wiDotM, cosThetaT);
        float etaM = wiDotM < 0.0f ? ior : 1.0f/ior;
        bool reflect;
        if (sampleR && sampleT) {
            reflect = e.sampler->nextBoolean(F);
        } else if (sampleT) {
            if (F == 1.0f) return false;
            reflect = false;
        } else if (sampleR) {
            reflect = true;
        } else {
            return false;
        }
        if (reflect) e.wo = 2.0f*wiDotM*m - e.wi;
        else e.wo = (etaM*wiDotM - sgnE(wiDotM)*cosThetaT)*m - etaM*e.wi;
        float woDotN = e.wo.z;
        bool reflected = wiDotN*woDotN > 0.0f;
        if (reflected != reflect) return false;
        float woDotM = e.wo.dot(m);
        float G = mf.G(alpha, e.wi, e.wo, m);
        float D = mf.D(alpha, m);
        e.weight = V3(std::abs(wiDotM)*G*D/(std::abs(wiDotN)*pm));
        if (reflect) {
            e.pdf = pm*0.25f/std::abs(wiDotM);
            e.sampledLobe = GlossyReflectionLobe;
        } else {
            float q = eta*wiDotM + woDotM;
            e.pdf = pm*std::abs(woDotM)/(q*q);
            e.sampledLobe = GlossyTransmissionLobe;
        }
        if (sampleR && sampleT) {
            if (reflect) e.pdf *= F; else e.pdf *= 1.0f - F;
        } else {
            if (reflect) e.weight *= F; else e.weight *= 1.0f - F;
        }
        e.weight *= texEval(d.albedo_tex, e.info->uvx, e.info->uvy);
        e.weight *= dielectricEtaSq(d.ior, e.wi, e.wo);
        return true;
    }
    case TGPU_BSDF_MIXED: { // bsdfs/MixedBsdf.cpp:17-30 (adjustedRatio), :72-102
        bool sample0 = (e.requestedLobe & bsdfLobes(d.substrate)) != 0, sample1 = (e.requestedLobe & bsdfLobes(d.aux_bsdf)) != 0;
        float ratio;
        if (sample0 && sample1) ratio = texEval(d.aux_tex, e.info->uvx, e.info->uvy).x;
        else if (sample0) ratio = 1.0f;
        else if (sample1) ratio = 0.0f;
        else return false;
        if (e.sampler->nextBoolean(ratio)) {
            if (!bsdfSample(d.substrate, e)) return false;
            float pdf0 = e.pdf*ratio;
            float pdf1 = bsdfPdf(d.aux_bsdf, e)*(1.0f - ratio);
            V3 f = e.weight*e.pdf*ratio + bsdfEval(d.aux_bsdf, e)*(1.0f - ratio);
            e.pdf = pdf0 + pdf1;
            e.weight = f/e.pdf;
        } else {
            if (!bsdfSample(d.aux_bsdf, e)) return false;
            float pdf0 = bsdfPdf(d.substrate, e)*ratio;
            float pdf1 = e.pdf*(1.0f - ratio);
            V3 f = bsdfEval(d.substrate, e)*ratio + e.weight*e.pdf*(1.0f - ratio);
            e.pdf = pdf0 + pdf1;
            e.weight = f/e.pdf;
        }
        e.weight *= texEval(d.albedo_tex, e.info->uvx, e.info->uvy);
        return true;
    }
    case TGPU_BSDF_TRANSPARENCY: return bsdfSample(d.substrate, e); // bsdfs/TransparencyBsdf.cpp:42-45
    case TGPU_BSDF_THINSHEET: { // bsdfs/ThinSheetBsdf.cpp:50-81
        if (!(e.requestedLobe & SpecularReflectionLobe)) return false;
        e.wo = V3(-e.wi.x, -e.wi.y, e.wi.z);
        e.pdf = 1.0f;
        e.sampledLobe = SpecularReflectionLobe;
        V3 sigmaA(d.sigma_a);
        if (sigmaA == 0.0f) { // fast path / early out
            e.weight = V3(1.0f);
            return true;
        }
        float thickness = texEval(d.aux_tex, e.info->uvx, e.info->uvy).x;
        float cosThetaT;
        e.weight = V3(thinFilmReflectance(1.0f/d.ior, std::abs(e.wi.z), cosThetaT));
        V3 transmittance = V3(1.0f) - e.weight;
        if (cosThetaT > 0.0f) {
            V3 a = -sigmaA*(thickness*2.0f/cosThetaT);
            transmittance *= V3(mexp(a.x), mexp(a.y), mexp(a.z));
        }
        e.weight /= 1.0f - transmittance.avg();
        return true;
    }
    case TGPU_BSDF_ROUGH_COAT: { // bsdfs/RoughCoatBsdf.cpp:47-87 (substrateEvalAndPdf), :89-157
        if (e.wi.z <= 0.0f) return false;
        bool sampleR = (e.requestedLobe & GlossyReflectionLobe) != 0;
        bool sampleT = (e.requestedLobe & bsdfLobes(d.substrate)) != 0;
        if (!sampleR && !sampleT) return false;
        const V3 wi = e.wi;
        float eta = 1.0f/d.ior;
        float cosThetaTi;
        float Fi = dielectricReflectance(eta, wi.z, cosThetaTi);
        float substrateWeight = d.avg_transmittance*(1.0f - Fi);
        float specularWeight = Fi;
        float specularProbability = specularWeight/(specularWeight + substrateWeight);
        float roughness = texEval(d.roughness_tex, e.info->uvx, e.info->uvy).x;
        V3 ssa(d.scaled_sigma_a);
        if (sampleR && (e.sampler->nextBoolean(specularProbability) || !sampleT)) {
            if (!rdSampleBase(e, true, false, roughness, d.ior, d.distribution)) return false;
            if (sampleT) {
                V3 brdfSubstrate(0.0f), brdfSpecular = e.weight*e.pdf;
                float pdfSubstrate = 0.0f, pdfSpecular = e.pdf*specularProbability;
                { // substrateEvalAndPdf
                    const V3 &wo = e.wo;
                    float cosThetaTo;
                    float Fo = dielectricReflectance(eta, wo.z, cosThetaTo);
                    if (!(Fi == 1.0f || Fo == 1.0f)) {
                        V3 wiSubstrate(wi.x*eta, wi.y*eta, std::copysign(cosThetaTi, wi.z));
                        V3 woSubstrate(wo.x*eta, wo.y*eta, std::copysign(cosThetaTo, wo.z));
                        pdfSubstrate = bsdfPdf(d.substrate, warped(e, wiSubstrate, woSubstrate));
                        pdfSubstrate *= eta*eta*std::abs(wo.z/cosThetaTo);
                        float compressionProjection = eta*eta*wo.z/cosThetaTo;
                        V3 substrateF = bsdfEval(d.substrate, warped(e, wiSubstrate, woSubstrate));
                        if (ssa.max() > 0.0f) { V3 a = ssa*(-1.0f/cosThetaTo - 1.0f/cosThetaTi); substrateF *= V3(mexp(a.x), mexp(a.y), mexp(a.z)); }
                        brdfSubstrate = compressionProjection*(1.0f - Fi)*(1.0f - Fo)*substrateF;
                    }
                }
                pdfSubstrate *= 1.0f - specularProbability;
                e.weight = (brdfSpecular + brdfSubstrate)/(pdfSpecular + pdfSubstrate);
                e.pdf = pdfSpecular + pdfSubstrate;
            }
            return true;
        } else {
            V3 wiSubstrate(wi.x*eta, wi.y*eta, cosThetaTi);
            e.wi = wiSubstrate;
            bool success = bsdfSample(d.substrate, e);
            e.wi = wi;
            if (!success) return false;
            float cosThetaTo;
            float Fo = dielectricReflectance(d.ior, e.wo.z, cosThetaTo);
            if (Fo == 1.0f) return false;
            float cosThetaSubstrate = e.wo.z;
            e.wo = V3(e.wo.x*d.ior, e.wo.y*d.ior, cosThetaTo);
            e.weight *= (1.0f - Fi)*(1.0f - Fo);
            if (ssa.max() > 0.0f) { V3 a = ssa*(-1.0f/cosThetaSubstrate - 1.0f/cosThetaTi); e.weight *= V3(mexp(a.x), mexp(a.y), mexp(a.z)); }
            e.weight *= wi.z/wiSubstrate.z;
            e.pdf *= eta*eta*cosThetaTo/cosThetaSubstrate;
            if (sampleR) {
                V3 brdfSubstrate = e.weight*e.pdf;
                float pdfSubstrate = e.pdf*(1.0f - specularProbability);
                V3 brdfSpecular = rdEvalBase(e, true, false, roughness, d.ior, d.distribution);
                float pdfSpecular = rdPdfBase(e, true, false, roughness, d.ior, d.distribution);
                pdfSpecular *= specularProbability;
                e.weight = (brdfSpecular + brdfSubstrate)/(pdfSpecular + pdfSubstrate);
                e.pdf = pdfSpecular + pdfSubstrate;
            }
        }
        return true;
    }
    case TGPU_BSDF_SMOOTH_COAT: { // bsdfs/SmoothCoatBsdf.cpp:41-100
        if (e.wi.z <= 0.0f) return false;
        uint32_t subLobes = bsdfLobes(d.substrate);
        bool sampleR = (e.requestedLobe & SpecularReflectionLobe) != 0;
        bool sampleT = (e.requestedLobe & subLobes) != 0;
        if (!sampleR && !sampleT) return false;
        const V3 wi = e.wi;
        float eta = 1.0f/d.ior;
        float cosThetaTi;
        float Fi = dielectricReflectance(eta, wi.z, cosThetaTi);
        float substrateWeight = d.avg_transmittance*(1.0f - Fi);
        float specularWeight = Fi;
        float specularProbability;
        if (sampleR && sampleT)
            specularProbability = specularWeight/(specularWeight + substrateWeight);
        else if (sampleR)
            specularProbability = 1.0f;
        else
            specularProbability = 0.0f;
        if (sampleR && e.sampler->nextBoolean(specularProbability)) {
            e.wo = V3(-wi.x, -wi.y, wi.z);
            e.pdf = specularProbability;
            e.weight = V3(Fi/specularProbability);
            e.sampledLobe = SpecularReflectionLobe;
        } else {
            V3 originalWi(wi);
            V3 wiSubstrate(wi.x*eta, wi.y*eta, cosThetaTi);
            e.wi = wiSubstrate;
            bool success = bsdfSample(d.substrate, e);
            e.wi = originalWi;
            if (!success) return false;
            float cosThetaTo;
            float Fo = dielectricReflectance(d.ior, e.wo.z, cosThetaTo);
            if (Fo == 1.0f) return false;
            float cosThetaSubstrate = e.wo.z;
            e.wo = V3(e.wo.x*d.ior, e.wo.y*d.ior, cosThetaTo);
            e.weight *= (1.0f - Fi)*(1.0f - Fo);
            V3 ss(d.scaled_sigma_a);
            if (ss.max() > 0.0f) {
                V3 a = ss*(-1.0f/cosThetaSubstrate - 1.0f/cosThetaTi);
                e.weight *= V3(mexp(a.x), mexp(a.y), mexp(a.z));
            }
            e.weight /= 1.0f - specularProbability;
            e.pdf *= 1.0f - specularProbability;
            e.pdf *= eta*eta*cosThetaTo/cosThetaSubstrate;
        }
        return true;
    }
    default: return false;
    }
}

float oracle_ctx::bsdfPdf(int b, const Event &e) const
{
    if (b < 0) return 0.0f;
    const tgpu_bsdf &d = sc.bsdfs[b];
    switch (d.type) {
    case TGPU_BSDF_LAMBERT: // bsdfs/LambertBsdf.cpp:61-68
        if (!(e.requestedLobe & DiffuseReflectionLobe)) return 0.0f;
        if (e.wi.z <= 0.0f || e.wo.z <= 0.0f) return 0.0f;
        return std::abs(e.wo.z)*INV_PI;
    case TGPU_BSDF_OREN_NAYAR: { // bsdfs/OrenNayarBsdf.cpp:125-135
        if (!(e.requestedLobe & DiffuseReflectionLobe)) return 0.0f;
        if (e.wi.z <= 0.0f || e.wo.z <= 0.0f) return 0.0f;
        float ratio = clampf(texEval(d.roughness_tex, e.info->uvx, e.info->uvy).x, 0.01f, 1.0f);
        return INV_TWO_PI*ratio + (std::abs(e.wo.z)*INV_PI)*(1.0f - ratio);
    }
    case TGPU_BSDF_PHONG: { // bsdfs/PhongBsdf.cpp:102-121
        bool evalGlossy = (e.requestedLobe & GlossyReflectionLobe) != 0, evalDiffuse = (e.requestedLobe & DiffuseReflectionLobe) != 0;
        if (!evalGlossy && !evalDiffuse) return 0.0f;
        if (e.wi.z <= 0.0f || e.wo.z <= 0.0f) return 0.0f;
        const float pdfFactor = (d.exponent + 1.0f)*INV_TWO_PI;
        float result = 0.0f;
        if (evalGlossy) {
            float cosTheta = V3(-e.wi.x, -e.wi.y, e.wi.z).dot(e.wo);
            if (cosTheta > 0.0f) result += mpow(cosTheta, d.exponent)*pdfFactor;
        }
        if (evalDiffuse && evalGlossy) result = result*(1.0f - d.diffuse_ratio) + d.diffuse_ratio*(std::abs(e.wo.z)*INV_PI);
        else if (evalDiffuse) result = std::abs(e.wo.z)*INV_PI;
        return result;
    }
    case TGPU_BSDF_ROUGH_CONDUCTOR: { // bsdfs/RoughConductorBsdf.cpp:127-143
        if (!(e.requestedLobe & GlossyReflectionLobe)) return 0.0f;
        if (e.wi.z <= 0.0f || e.wo.z <= 0.0f) return 0.0f;
        MF mf{this, d.distribution};
        float roughness = texEval(d.roughness_tex, e.info->uvx, e.info->uvy).x;
        float sampleAlpha = roughnessToAlpha(roughness);
        V3 hr = (e.wi + e.wo).normalized();
        return mf.pdf(sampleAlpha, hr)*0.25f/e.wi.dot(hr);
    }
    case TGPU_BSDF_PLASTIC: { // bsdfs/PlasticBsdf.cpp:165-187
        if (e.wi.z <= 0.0f || e.wo.z <= 0.0f) return 0.0f;
        bool sampleR = (e.requestedLobe & SpecularReflectionLobe) != 0;
        bool sampleT = (e.requestedLobe & DiffuseReflectionLobe) != 0;
        if (sampleR && sampleT) {
            float ct;
            float Fi = dielectricReflectance(1.0f/d.ior, e.wi.z, ct);
            float substrateWeight = d.avg_transmittance*(1.0f - Fi);
            float specularWeight = Fi;
            float specularProbability = specularWeight/(specularWeight + substrateWeight);
            if (checkReflectionConstraint(e.wi, e.wo)) return specularProbability;
            return (std::abs(e.wo.z)*INV_PI)*(1.0f - specularProbability);
        } else if (sampleT) {
            return std::abs(e.wo.z)*INV_PI;
        } else if (sampleR) {
            return checkReflectionConstraint(e.wi, e.wo) ? 1.0f : 0.0f;
        }
        return 0.0f;
    }
    case TGPU_BSDF_ROUGH_PLASTIC: { // bsdfs/RoughPlasticBsdf.cpp:223-250
        bool sampleR = (e.requestedLobe & GlossyReflectionLobe) != 0;
        bool sampleT = (e.requestedLobe & DiffuseReflectionLobe) != 0;
        if (!sampleR && !sampleT) return 0.0f;
        if (e.wi.z <= 0.0f || e.wo.z <= 0.0f) return 0.0f;
        float glossyPdf = 0.0f;
        if (sampleR) glossyPdf = rdPdfBase(e, true, false, texEval(d.roughness_tex, e.info->uvx, e.info->uvy).x, d.ior, d.distribution);
        float diffusePdf = 0.0f;
        if (sampleT) diffusePdf = std::abs(e.wo.z)*INV_PI;
        if (sampleT && sampleR) {
            float ct;
            float Fi = dielectricReflectance(1.0f/d.ior, e.wi.z, ct);
            float substrateWeight = d.substrate_weight*d.avg_transmittance*(1.0f - Fi);
            float specularWeight = Fi;
            float specularProbability = specularWeight/(specularWeight + substrateWeight);
            diffusePdf *= (1.0f - specularProbability);
            glossyPdf *= specularProbability;
        }
        return glossyPdf + diffusePdf;
    }
    case TGPU_BSDF_MIRROR: case TGPU_BSDF_CONDUCTOR: // MirrorBsdf.cpp:58-65, ConductorBsdf.cpp:80-87
        return ((e.requestedLobe & SpecularReflectionLobe) && checkReflectionConstraint(e.wi, e.wo)) ? 1.0f : 0.0f;
    case TGPU_BSDF_DIELECTRIC: { // bsdfs/DielectricBsdf.cpp:148-168
        bool sampleR = (e.requestedLobe & SpecularReflectionLobe) != 0;
        bool sampleT = (e.requestedLobe & SpecularTransmissionLobe) != 0 && d.enable_refraction;
        float eta = e.wi.z < 0.0f ? d.ior : 1.0f/d.ior;
        float cosThetaT = 0.0f;
        float F = dielectricReflectance(eta, std::abs(e.wi.z), cosThetaT);
        if (e.wi.z*e.wo.z >= 0.0f) {
            if (sampleR && checkReflectionConstraint(e.wi, e.wo)) return sampleT ? F : 1.0f;
            return 0.0f;
        }
        if (sampleT && checkRefractionConstraint(e.wi, e.wo, eta, cosThetaT)) return sampleR ? 1.0f - F : 1.0f;
        return 0.0f;
    }
    case TGPU_BSDF_ROUGH_DIELECTRIC: { // bsdfs/RoughDielectricBsdf.cpp:198-234 (pdfBase)
        bool sampleR = (e.requestedLobe & GlossyReflectionLobe) != 0;
        bool sampleT = (e.requestedLobe & GlossyTransmissionLobe) != 0 && d.enable_refraction;
        float roughness = texEval(d.roughness_tex, e.info->uvx, e.info->uvy).x;
        float wiDotN = e.wi.z, woDotN = e.wo.z;
        bool reflect = wiDotN*woDotN >= 0.0f;
        if ((reflect && !sampleR) || (!reflect && !sampleT)) return 0.0f;
        MF mf{this, d.distribution};
        float sampleRoughness = (1.2f - 0.2f*std::sqrt(std::abs(wiDotN)))*roughness;
        float sampleAlpha = roughnessToAlpha(sampleRoughness);
        float eta = wiDotN < 0.0f ? d.ior : 1.0f/d.ior;
        V3 m;
        if (reflect) m = sgnE(wiDotN)*(e.wi + e.wo).normalized();
        else m = -(e.wi*eta + e.wo).normalized();
        float wiDotM = e.wi.dot(m), woDotM = e.wo.dot(m);
        float cosThetaT;
        float F = dielectricReflectance(1.0f/d.ior, wiDotM, cosThetaT);
        float pm = mf.pdf(sampleAlpha, m);
        float pdf;
        if (reflect) {
            pdf = pm*0.25f/std::abs(wiDotM);
        } else {
            float q = eta*wiDotM + woDotM;
            pdf = pm*std::abs(woDotM)/(q*q);
        }
        if (sampleR && sampleT) {
            if (reflect) pdf *= F; else pdf *= 1.0f - F;
        }
        return pdf;
    }
    case TGPU_BSDF_MIXED: { // bsdfs/MixedBsdf.cpp:125-131
        bool sample0 = (e.requestedLobe & bsdfLobes(d.substrate)) != 0, sample1 = (e.requestedLobe & bsdfLobes(d.aux_bsdf)) != 0;
        float ratio;
        if (sample0 && sample1) ratio = texEval(d.aux_tex, e.info->uvx, e.info->uvy).x;
        else if (sample0) ratio = 1.0f;
        else if (sample1) ratio = 0.0f;
        else return 0.0f;
        return bsdfPdf(d.substrate, e)*ratio + bsdfPdf(d.aux_bsdf, e)*(1.0f - ratio);
    }
    case TGPU_BSDF_TRANSPARENCY: return bsdfPdf(d.substrate, e); // bsdfs/TransparencyBsdf.cpp:60-63
    case TGPU_BSDF_THINSHEET: // bsdfs/ThinSheetBsdf.cpp:111-118
        return ((e.requestedLobe & SpecularReflectionLobe) && checkReflectionConstraint(e.wi, e.wo)) ? 1.0f : 0.0f;
    case TGPU_BSDF_ROUGH_COAT: { // bsdfs/RoughCoatBsdf.cpp:256-298
        bool sampleR = (e.requestedLobe & GlossyReflectionLobe) != 0;
        bool sampleT = (e.requestedLobe & bsdfLobes(d.substrate)) != 0;
        if (!sampleT && !sampleR) return 0.0f;
        if (e.wi.z <= 0.0f || e.wo.z <= 0.0f) return 0.0f;
        const V3 &wi = e.wi, &wo = e.wo;
        float eta = 1.0f/d.ior;
        float cosThetaTi, cosThetaTo;
        float Fi = dielectricReflectance(eta, wi.z, cosThetaTi);
        float Fo = dielectricReflectance(eta, wo.z, cosThetaTo);
        float specularProbability;
        if (sampleR && sampleT) {
            float substrateWeight = d.avg_transmittance*(1.0f - Fi);
            float specularWeight = Fi;
            specularProbability = specularWeight/(specularWeight + substrateWeight);
        } else {
            specularProbability = sampleR ? 1.0f : 0.0f;
        }
        float glossyPdf = 0.0f;
        if (sampleR) glossyPdf = rdPdfBase(e, true, false, texEval(d.roughness_tex, e.info->uvx, e.info->uvy).x, d.ior, d.distribution);
        float substratePdf = 0.0f;
        if (sampleT && Fi < 1.0f && Fo < 1.0f) {
            V3 wiSubstrate(wi.x*eta, wi.y*eta, std::copysign(cosThetaTi, wi.z));
            V3 woSubstrate(wo.x*eta, wo.y*eta, std::copysign(cosThetaTo, wo.z));
            substratePdf = bsdfPdf(d.substrate, warped(e, wiSubstrate, woSubstrate));
            substratePdf *= eta*eta*std::abs(wo.z/cosThetaTo);
        }
        return glossyPdf*specularProbability + substratePdf*(1.0f - specularProbability);
    }
    case TGPU_BSDF_SMOOTH_COAT: { // bsdfs/SmoothCoatBsdf.cpp:180-215
        if (e.wi.z <= 0.0f || e.wo.z <= 0.0f) return 0.0f;
        uint32_t subLobes = bsdfLobes(d.substrate);
        bool sampleR = (e.requestedLobe & SpecularReflectionLobe) != 0;
        bool sampleT = (e.requestedLobe & subLobes) != 0;
        const V3 &wi = e.wi, &wo = e.wo;
        float eta = 1.0f/d.ior;
        float cosThetaTi, cosThetaTo;
        float Fi = dielectricReflectance(eta, wi.z, cosThetaTi);
        dielectricReflectance(eta, wo.z, cosThetaTo);
        V3 wiSubstrate(wi.x*eta, wi.y*eta, std::copysign(cosThetaTi, wi.z));
        V3 woSubstrate(wo.x*eta, wo.y*eta, std::copysign(cosThetaTo, wo.z));
        if (sampleR && sampleT) {
            float substrateWeight = d.avg_transmittance*(1.0f - Fi);
            float specularWeight = Fi;
            float specularProbability = specularWeight/(specularWeight + substrateWeight);
            if (checkReflectionConstraint(e.wi, e.wo))
                return specularProbability;
            else
                return bsdfPdf(d.substrate, warped(e, wiSubstrate, woSubstrate))*(1.0f - specularProbability)*eta*eta*std::abs(wo.z/cosThetaTo);
        } else if (sampleT) {
            return bsdfPdf(d.substrate, warped(e, wiSubstrate, woSubstrate))*eta*eta*std::abs(wo.z/cosThetaTo);
        } else if (sampleR) {
            return checkReflectionConstraint(e.wi, e.wo) ? 1.0f : 0.0f;
        }
        return 0.0f;
    }
    default: return 0.0f;
    }
}

// ========================================================================================
// integrator (integrators/TraceBase.cpp, integrators/path_tracer/PathTracer.cpp)
// ========================================================================================
Event oracle_ctx::makeLocalScatterEvent(const Temp & /*data*/, const Info &info, const Ray &ray, Sampler *sampler) const
{   // TraceBase.cpp:24-51 ; Primitive::setupTangentFrame (Primitive.cpp:125-163) without bump/anisotropy
    Frame frame = duffFrame(info.Ns);
    bool hitBackside = frame.normal.dot(ray.dir) > 0.0f;
    bool isTransmissive = (bsdfLobes(info.bsdf) & TransmissiveLobe) != 0;
    bool flipFrame = sc.settings.enable_two_sided_shading && hitBackside && !isTransmissive;
    if (flipFrame) {
        frame.normal = -frame.normal;
        frame.tangent = -frame.tangent;
    }
    Event e;
    e.info = &info;
    e.sampler = sampler;
    e.frame = frame;
    e.wi = frame.toLocal(-ray.dir);
    e.wo = V3(0.0f);
    e.weight = V3(1.0f);
    e.pdf = 1.0f;
    e.requestedLobe = AllLobes;
    e.sampledLobe = 0;
    e.flippedFrame = flipFrame;
    return e;
}

V3 oracle_ctx::shadowRay(Ray &ray, int medium, int endKind, uint32_t endIndex, int bounce, bool startsOnSurface, bool endsOnSurface, Counters &c) const
{   // TraceBase.cpp:62-125 (generalizedShadowRayImpl<false>): surfaces with a forward lobe let the ray through (and may
    // switch the medium), anything else blocks; every segment is attenuated by the medium it runs in
    Temp data;
    Info info;
    float initialFarT = ray.farT;
    V3 throughput(1.0f);
    do {
        c.shadow++;
        bool didHit = intersect(ray, data, info, c) && !(info.kind == endKind && info.index == endIndex);
        if (didHit) {
            if (!(bsdfLobes(info.bsdf) & ForwardLobe)) return V3(0.0f);
            Event event = makeLocalScatterEvent(data, info, ray, nullptr);
            Event fwd = event; // makeForwardEvent
            fwd.wo = -fwd.wi;
            fwd.requestedLobe = ForwardLobe;
            V3 transparency = bsdfEval(info.bsdf, fwd);
            if (transparency == 0.0f) return V3(0.0f);
            throughput *= transparency;
            bounce++;
            if (bounce >= sc.settings.max_bounces) return V3(0.0f);
        }
        if (medium >= 0) throughput *= mediumTransmittance(medium, ray, startsOnSurface, endsOnSurface);
        if (info.kind == PRIM_NONE || (info.kind == endKind && info.index == endIndex))
            return bounce >= sc.settings.min_bounces ? throughput : V3(0.0f);
        medium = selectMedium(info, medium, !data.backSide); // selectMedium(medium, !hitBackside(data))
        startsOnSurface = true;
        ray.pos = ray.pos + ray.dir*ray.farT; // ray.hitpoint()
        initialFarT -= ray.farT;
        ray.nearT = info.epsilon;
        ray.farT = initialFarT;
    } while (true);
}

V3 oracle_ctx::attenuatedEmission(const tgpu_light &l, int medium, float expectedDist, Temp &data, Info &info, int bounce, bool startsOnSurface,
                                  Ray &ray, Counters &c, V3 *transmittance) const
{   // TraceBase.cpp:144-174
    constexpr float fudgeFactor = 1.0f + 1e-3f;
    if (!lightIntersect(l, ray, data) || ray.farT*fudgeFactor < expectedDist) return V3(0.0f);
    info.p = ray.pos + ray.dir*ray.farT;
    info.w = ray.dir;
    intersectionInfo(data, info);
    int endKind = l.kind == TGPU_LIGHT_ENV ? PRIM_ENV : (l.kind == TGPU_LIGHT_SPHERE ? PRIM_SPHERE : (l.kind == TGPU_LIGHT_DISK ? PRIM_DISK : PRIM_QUAD));
    V3 shadow = shadowRay(ray, medium, endKind, uint32_t(l.index), bounce, startsOnSurface, true, c);
    if (transmittance) *transmittance = shadow; // the 'visibility' output buffer (TraceBase.cpp:168-169)
    if (shadow == 0.0f) return V3(0.0f);
    return shadow*primEvalDirect(data, info);
}

V3 oracle_ctx::lightSample(const tgpu_light &l, Event &event, int medium, int bounce, const Ray &parent, Counters &c, V3 *transmittance) const
{   // TraceBase.cpp:246-285
    LightSample sample;
    if (!lightSampleDirect(l, event.info->p, *event.sampler, sample)) return V3(0.0f);
    event.wo = event.frame.toLocal(sample.d);
    if (sc.settings.enable_consistency_checks) {
        bool geometricBackside = sample.d.dot(event.info->Ng) < 0.0f;
        bool shadingBackside = (event.wo.z < 0.0f) ^ event.flippedFrame;
        if (geometricBackside != shadingBackside) return V3(0.0f);
    }
    medium = selectMedium(*event.info, medium, sample.d.dot(event.info->Ng) < 0.0f); // the medium on the light's side (:261-262)
    event.requestedLobe = AllButSpecular;
    V3 f = bsdfEval(event.info->bsdf, event); // eval(event, false)
    if (f == 0.0f) return V3(0.0f);
    Ray ray{event.info->p, sample.d, event.info->epsilon, INF};
    (void)parent;
    Temp data;
    Info info;
    V3 e = attenuatedEmission(l, medium, sample.dist, data, info, bounce, true, ray, c, transmittance);
    TRC("  neeL d=%08x %08x %08x dist=%08x pdf=%08x f=%08x %08x %08x e=%08x %08x %08x farT=%08x\n", HX3(sample.d), hx(sample.dist), hx(sample.pdf), HX3(f), HX3(e), hx(ray.farT));
    if (e == 0.0f) return V3(0.0f);
    V3 lightF = f*e/sample.pdf;
    float bp = bsdfPdf(event.info->bsdf, event);
    lightF *= (sample.pdf*sample.pdf)/(sample.pdf*sample.pdf + bp*bp); // SampleWarp::powerHeuristic (SampleWarp.hpp:189-192)
    TRC("  neeL bp=%08x L=%08x %08x %08x\n", hx(bp), HX3(lightF));
    return lightF;
}

V3 oracle_ctx::bsdfSampleNee(const tgpu_light &l, Event &event, int medium, int bounce, const Ray & /*parent*/, Counters &c) const
{   // TraceBase.cpp:287-321
    event.requestedLobe = AllButSpecular;
    if (!bsdfSample(event.info->bsdf, event)) return V3(0.0f);
    if (event.weight == 0.0f) return V3(0.0f);
    V3 wo = event.frame.toGlobal(event.wo);
    if (sc.settings.enable_consistency_checks) {
        bool geometricBackside = wo.dot(event.info->Ng) < 0.0f;
        bool shadingBackside = (event.wo.z < 0.0f) ^ event.flippedFrame;
        if (geometricBackside != shadingBackside) return V3(0.0f);
    }
    medium = selectMedium(*event.info, medium, wo.dot(event.info->Ng) < 0.0f); // :301-302
    Ray ray{event.info->p, wo, event.info->epsilon, INF};
    Temp data;
    Info info;
    V3 e = attenuatedEmission(l, medium, -1.0f, data, info, bounce, true, ray, c);
    TRC("  neeB wo=%08x %08x %08x w=%08x %08x %08x pdf=%08x e=%08x %08x %08x farT=%08x\n", HX3(wo), HX3(event.weight), hx(event.pdf), HX3(e), hx(ray.farT));
    if (e == 0.0f) return V3(0.0f);
    V3 bsdfF = e*event.weight;
    float lp = lightDirectPdf(l, data, info, event.info->p);
    bsdfF *= (event.pdf*event.pdf)/(event.pdf*event.pdf + lp*lp);
    TRC("  neeB lp=%08x B=%08x %08x %08x\n", hx(lp), HX3(bsdfF));
    return bsdfF;
}

const tgpu_light *oracle_ctx::chooseLight(const V3 &p, Sampler &sampler, float &weight) const
{   // TraceBase.cpp:416-459
    if (sc.n_lights == 0) return nullptr;
    const tgpu_light *light;
    if (sc.n_lights == 1) {
        weight = 1.0f;
        light = &sc.lights[0];
    } else {
        float lightPdf[64];
        uint32_t n = std::min<uint32_t>(sc.n_lights, 64);
        float total = 0.0f;
        unsigned numNonNegative = 0;
        for (uint32_t i = 0; i < n; ++i) {
            lightPdf[i] = lightApproximateRadiance(sc.lights[i], p);
            if (lightPdf[i] >= 0.0f) {
                total += lightPdf[i];
                numNonNegative++;
            }
        }
        if (numNonNegative == 0) {
            for (uint32_t i = 0; i < n; ++i) lightPdf[i] = 1.0f;
            total = float(n);
        } else if (numNonNegative < n) {
            for (uint32_t i = 0; i < n; ++i) {
                float uniformWeight = (total == 0.0f ? 1.0f : total)/numNonNegative;
                if (lightPdf[i] < 0.0f) {
                    lightPdf[i] = uniformWeight;
                    total += uniformWeight;
                }
            }
        }
        if (total == 0.0f) return nullptr;
        float t = sampler.next1D()*total;
        light = nullptr;
        for (uint32_t i = 0; i < n; ++i) {
            if (t < lightPdf[i] || i == n - 1) {
                weight = total/lightPdf[i];
                light = &sc.lights[i];
                break;
            } else {
                t -= lightPdf[i];
            }
        }
    }
    return light;
}

V3 oracle_ctx::estimateDirect(Event &event, int medium, int bounce, const Ray &parent, Counters &c, V3 *transmittance) const
{   // TraceBase.cpp:483-494 -> sampleDirect :383-400
    float weight;
    const tgpu_light *light = chooseLight(event.info->p, *event.sampler, weight);
    if (!light) return V3(0.0f);
    V3 result(0.0f);
    uint32_t lobes = bsdfLobes(event.info->bsdf);
    bool pureSpecular = lobes != 0 && (lobes & ~uint32_t(SpecularLobe)) == 0;
    if (pureSpecular || lobes == ForwardLobe) return V3(0.0f);
    result += lightSample(*light, event, medium, bounce, parent, c, transmittance);
    result += bsdfSampleNee(*light, event, medium, bounce, parent, c); // lights in scope are never Dirac
    return result*weight;
}

bool oracle_ctx::handleSurface(Event &event, Temp &data, Info &info, int &medium, int bounce, bool enableLightSampling, Ray &ray,
                               V3 &throughput, V3 &emission, bool &wasSpecular, MediumStateO &state, Counters &c, V3 *transmittance) const
{   // TraceBase.cpp:516-568 (adjoint = false)
    c.surface++;
    Event fwd = event; // makeForwardEvent
    fwd.wo = -fwd.wi;
    fwd.requestedLobe = ForwardLobe;
    V3 transparency = bsdfEval(info.bsdf, fwd);
    float transparencyScalar = transparency.avg();

    V3 wo;
    if (event.sampler->nextBoolean(transparencyScalar)) {
        wo = ray.dir;
        event.pdf = transparencyScalar;
        event.weight = transparency/transparencyScalar;
        event.sampledLobe = ForwardLobe;
        throughput *= event.weight;
    } else {
        if (enableLightSampling && bounce < sc.settings.max_bounces - 1)
            emission += estimateDirect(event, medium, bounce + 1, ray, c, transmittance)*throughput;
        if (primEmissive(info.kind, info.index) && bounce >= sc.settings.min_bounces) {
            if (!enableLightSampling || wasSpecular || !primSamplable(info.kind, info.index))
                emission += primEvalDirect(data, info)*throughput;
        }
        event.requestedLobe = AllLobes;
        if (!bsdfSample(info.bsdf, event)) return false;
        wo = event.frame.toGlobal(event.wo);
        if (sc.settings.enable_consistency_checks) {
            bool geometricBackside = wo.dot(info.Ng) < 0.0f;
            bool shadingBackside = (event.wo.z < 0.0f) ^ event.flippedFrame;
            if (geometricBackside != shadingBackside) return false;
        }
        throughput *= event.weight;
        wasSpecular = (event.sampledLobe & SpecularLobe) != 0;
    }
    bool geometricBackside = wo.dot(info.Ng) < 0.0f;
    medium = selectMedium(info, medium, geometricBackside);
    state.reset();
    V3 hitpoint = ray.pos + ray.dir*ray.farT;
    ray = Ray{hitpoint, wo, info.epsilon, INF};
    return true;
}

// ---- media ------------------------------------------------------------------------------
void oracle_ctx::buildExpTable()
{   // fmath::ExpVar<10> (thirdparty/fmath/fmath.hpp:90-127)
    float log_2 = ::logf(2.0f);
    expA = 1024/log_2;
    expB = log_2/1024;
    for (int i = 0; i < 1024; ++i) {
        float y = std::pow(2.0f, (float)i/1024);
        expTable[i] = tgpu_float_to_bits(y) & 0x7FFFFFu;
    }
}

V3 oracle_ctx::fastExp(const V3 &v) const
{   // FastMath::exp(Vec3f) = fmath::exp_ps on (x, y, z, 0) (math/FastMath.hpp:23-27, fmath.hpp:320-363)
    float x[3] = {v.x, v.y, v.z};
    bool over = false;
    for (int i = 0; i < 3; ++i)
        if (int32_t(tgpu_float_to_bits(x[i]) & 0x7FFFFFFFu) > int32_t(0x42B00000)) over = true;
    float out[3];
    for (int i = 0; i < 3; ++i) {
        float xi = x[i];
        if (over) {
            xi = xi < 88.0f ? xi : 88.0f;   // _mm_min_ps(x, maxX)
            xi = xi > -88.0f ? xi : -88.0f; // _mm_max_ps(x, minX)
        }
        int32_t r = _mm_cvtss_si32(_mm_set_ss(xi*expA)); // cvtps_epi32: round to nearest even
        float t = xi - float(r)*expB;
        t = t + 1.0f;
        uint32_t vIdx = uint32_t(r) & 1023u;
        uint32_t u = ((uint32_t(r) + (127u << 10)) >> 10) << 23;
        out[i] = t*tgpu_bits_to_float(expTable[vIdx] | u);
    }
    return V3(out[0], out[1], out[2]);
}

// ---- transmittances (transmittances/*.cpp; Transmittance.hpp:22-42 picks the form from where the segment starts / ends)
V3 oracle_ctx::transForm(const tgpu_medium &m, int form, const V3 &tau) const
{
    if (m.transmittance != TGPU_TRANS_INTERPOLATED) return modelForm(m.transmittance, m.param, form, tau);
    // InterpolatedTransmittance.cpp:43-71
    auto lerpV = [](const V3 &a, const V3 &b, float t) { return a*(1.0f - t) + b*t; }; // math/MathUtil.hpp lerp
    const float u = m.tr_ratio;
    const float sbA = modelSigmaBar(m.tr_a_type, m.tr_a_param), sbB = modelSigmaBar(m.tr_b_type, m.tr_b_param);
    const float sigmaBar = transSigmaBar(m);
    if (form == 0) return sigmaBar*lerpV(modelForm(m.tr_a_type, m.tr_a_param, 0, tau)/sbA, modelForm(m.tr_b_type, m.tr_b_param, 0, tau)/sbB, u);
    if (form == 1 || form == 2) {
        V3 ms = lerpV(modelForm(m.tr_a_type, m.tr_a_param, 2, tau), modelForm(m.tr_b_type, m.tr_b_param, 2, tau), u);
        return form == 1 ? ms*sigmaBar : ms;
    }
    V3 pdfA = modelForm(m.tr_a_type, m.tr_a_param, 3, tau), pdfB = modelForm(m.tr_b_type, m.tr_b_param, 3, tau), result;
    auto isDirac = [](int type) { return type == TGPU_TRANS_LINEAR || type == TGPU_TRANS_PULSE; }; // LinearTransmittance / PulseTransmittance::isDirac
    for (int i = 0; i < 3; ++i) {
        bool diracA = isDirac(m.tr_a_type) && pdfA[i] > 0.0f;
        bool diracB = isDirac(m.tr_b_type) && pdfB[i] > 0.0f;
        if (diracA ^ diracB) result[i] = diracA ? pdfA[i] : pdfB[i];
        else result[i] = pdfA[i]*(1.0f - u) + pdfB[i]*u;
    }
    return result;
}

V3 oracle_ctx::modelForm(int type, const float *param, int form, const V3 &tau) const
{
    struct { const float *param; } m{param};
    auto vmin1 = [](const V3 &a) { return V3(fmin2(a.x, 1.0f), fmin2(a.y, 1.0f), fmin2(a.z, 1.0f)); }; // min(a, Vec3f(1))
    auto ex = [&](const V3 &v) { return V3(mexp(v.x), mexp(v.y), mexp(v.z)); };          // std::exp(Vec3f): per component
    auto pw = [&](const V3 &v, float e) { return V3(mpow(v.x, e), mpow(v.y, e), mpow(v.z, e)); }; // std::pow(Vec3f, float)
    switch (type) {
    case TGPU_TRANS_ERLANG: { // ErlangTransmittance.cpp:32-47
        const float lambda = param[0];
        if (form == 0) return 0.5f*ex(-lambda*tau)*(V3(2.0f) + lambda*tau);
        V3 ms = ex(-lambda*tau)*(V3(1.0f) + lambda*tau);
        if (form == 1) return ms*lambda*0.5f;
        if (form == 2) return ms;
        return (lambda*lambda)*tau*ex(-lambda*tau);
    }
    case TGPU_TRANS_DAVIS: { // DavisTransmittance.cpp:31-46
        const float alpha = param[0];
        if (form == 0) return pw(V3(1.0f) + tau/alpha, -alpha);
        if (form == 1 || form == 2) return pw(V3(1.0f) + tau/alpha, -(alpha + 1.0f));
        return (1.0f + 1.0f/alpha)*pw(V3(1.0f) + tau/alpha, -(alpha + 2.0f));
    }
    case TGPU_TRANS_DAVIS_WEINSTEIN: { // DavisWeinsteinTransmittance.cpp:28-80: the first channel only
        const float h = param[0], c = param[1];
        const float beta = 2.0f*h - 1.0f;
        const float t = tau.x;
        const float alpha = mpow(t, 1 - beta)/(mpow(c, 1 + beta)); // computeAlpha
        const float base = 1.0f + t/alpha;
        float Tr;
        if (form == 0) {
            Tr = mpow(base, -alpha);
        } else if (form == 1 || form == 2) {
            float trSurface = mpow(base, -alpha);
            Tr = trSurface*(beta/base - (beta - 1.0f)*alpha/t*mlog(base));
        } else {
            float logBase = mlog(base);
            float trSurface = mpow(base, -alpha);
            float term1 = beta*(-1.0f + beta*(1.0f + t) + (-1.0f + 2.0f*beta)*t/alpha)/(t*base*base);
            float term2 = ((-1.0f + beta)*beta*alpha/(t*t)*(2.0f*t + base)*logBase)/base;
            float term3 = (beta - 1.0f)*alpha/t*logBase;
            Tr = trSurface*(term1 - term2 + term3*term3);
        }
        return V3(std::isnan(Tr) ? 0.0f : Tr);
    }
    default: break;
    }
    switch (type) {
    case TGPU_TRANS_EXPONENTIAL: return fastExp(-tau); // all four forms (ExponentialTransmittance.cpp:26-41)
    case TGPU_TRANS_LINEAR: { // LinearTransmittance.cpp:30-59
        const float maxT = m.param[0];
        if (form == 0) return V3(1.0f) - vmin1(tau/maxT);
        V3 r;
        for (int i = 0; i < 3; ++i) {
            if (form == 1) r[i] = tau[i] > maxT ? 0.0f : 1.0f/maxT;
            else if (form == 2) r[i] = tau[i] > maxT ? 0.0f : 1.0f;
            else r[i] = std::abs(tau[i] - maxT) < 1e-3f ? 1.0f : 0.0f;
        }
        return r;
    }
    case TGPU_TRANS_QUADRATIC: { // QuadraticTransmittance.cpp:30-51
        const float maxT = m.param[0];
        if (form == 0) { V3 t = vmin1(tau/maxT); return V3(1.0f) - 2.0f*t + t*t; }
        if (form == 1) return (2.0f/maxT)*(V3(1.0f) - vmin1(tau/maxT));
        if (form == 2) return V3(1.0f) - vmin1(tau/maxT);
        V3 r(1.0f/maxT);
        for (int i = 0; i < 3; ++i) if (tau[i] > maxT) r[i] = 0.0f;
        return r;
    }
    case TGPU_TRANS_DOUBLE_EXPONENTIAL: { // DoubleExponentialTransmittance.cpp:34-49 (std::exp per component)
        const float sa = m.param[0], sb = m.param[1];
        auto ex = [&](const V3 &v) { return V3(mexp(v.x), mexp(v.y), mexp(v.z)); };
        V3 ea = ex(-sa*tau), eb = ex(-sb*tau);
        if (form == 0) return 0.5f*(ea + eb);
        if (form == 1) return 0.5f*(sa*ea + sb*eb);
        if (form == 2) return (sa*ea + sb*eb)/(sa + sb);
        return ((sa*sa)*ea + (sb*sb)*eb)/(sa + sb);
    }
    case TGPU_TRANS_PULSE: { // PulseTransmittance.cpp:43-81
        const float a = m.param[0], b = m.param[1];
        const int numPulses = int(m.param[2]);
        if (form == 0) {
            V3 idxF = (float(numPulses)*(tau - a))/(b - a) + 0.5f;
            V3 res;
            for (int i = 0; i < 3; ++i) {
                float f = clampf(idxF[i], 0.0f, float(numPulses));
                int idx = int(f);
                float height = float(numPulses - idx)/float(numPulses);
                float cellIntegral = height*(f - float(idx));
                if (idx > 0) cellIntegral += (idx - 0.5f) - (idx*(idx - 1))/(2.0f*numPulses);
                else cellIntegral -= 0.5f;
                res[i] = 1.0f - (2.0f/numPulses)*cellIntegral;
            }
            return res;
        }
        if (form == 1 || form == 2) {
            V3 res;
            for (int i = 0; i < 3; ++i) {
                int idx = int((float(numPulses)*(tau[i] - a))/(b - a) + 0.5f);
                idx = std::min(std::max(idx, 0), numPulses);
                res[i] = 1.0f - float(idx)/float(numPulses);
            }
            return form == 1 ? 2.0f/(b - a)*res : res;
        }
        V3 res;
        for (int i = 0; i < 3; ++i) {
            float f = clampf((float(numPulses)*(tau[i] - a))/(b - a), 0.0f, float(numPulses));
            int idx = int(f);
            res[i] = std::abs(f - idx - 0.5f) < 1e-3f ? 1.0f : 0.0f;
        }
        return (1.0f/numPulses)*res;
    }
    default: return V3(0.0f);
    }
}

float oracle_ctx::transSigmaBar(const tgpu_medium &m) const
{
    if (m.transmittance != TGPU_TRANS_INTERPOLATED) return modelSigmaBar(m.transmittance, m.param);
    // InterpolatedTransmittance.cpp:73-76: 1/lerp(1/sigmaBarA, 1/sigmaBarB, u)
    const float a = 1.0f/modelSigmaBar(m.tr_a_type, m.tr_a_param), b = 1.0f/modelSigmaBar(m.tr_b_type, m.tr_b_param);
    return 1.0f/(a*(1.0f - m.tr_ratio) + b*m.tr_ratio);
}
float oracle_ctx::modelSigmaBar(int type, const float *param) const
{
    struct { const float *param; } m{param};
    switch (type) {
    case TGPU_TRANS_LINEAR: return 1.0f/m.param[0];
    case TGPU_TRANS_QUADRATIC: return 2.0f/m.param[0];
    case TGPU_TRANS_DOUBLE_EXPONENTIAL: return 0.5f*(m.param[0] + m.param[1]);
    case TGPU_TRANS_PULSE: return 2.0f/(m.param[1] - m.param[0]);
    case TGPU_TRANS_ERLANG: return m.param[0]*0.5f; // ErlangTransmittance.cpp:49-52
    default: return 1.0f;                            // exponential, Davis, Davis-Weinstein
    }
}

V3 oracle_ctx::transEval(const tgpu_medium &m, const V3 &tau, bool startOnSurface, bool endOnSurface) const
{   // Transmittance::eval (Transmittance.hpp:22-30)
    if (startOnSurface && endOnSurface) return transForm(m, 0, tau);
    if (!startOnSurface && !endOnSurface) return transForm(m, 3, tau)/transSigmaBar(m);
    return transForm(m, 2, tau);
}

float oracle_ctx::transSample(const tgpu_medium &m, Sampler &sampler, bool startOnSurface) const
{   // Transmittance::sample -> sampleSurface / sampleMedium
    if (m.transmittance != TGPU_TRANS_INTERPOLATED) return modelSample(m.transmittance, m.param, sampler, startOnSurface);
    // InterpolatedTransmittance.cpp:78-85
    return sampler.nextBoolean(m.tr_ratio) ? modelSample(m.tr_b_type, m.tr_b_param, sampler, startOnSurface)
                                           : modelSample(m.tr_a_type, m.tr_a_param, sampler, startOnSurface);
}
float oracle_ctx::modelSample(int type, const float *param, Sampler &sampler, bool startOnSurface) const
{
    struct { const float *param; } m{param};
    switch (type) {
    case TGPU_TRANS_ERLANG: { // ErlangTransmittance.cpp:54-67
        const float lambda = param[0];
        if (!startOnSurface) {
            float xi0 = sampler.next1D(), xi1 = sampler.next1D(); // (evaluation order of the two draws: see the test)
            return -1.0f/lambda*mlog(xi0*xi1);
        }
        float xi = sampler.next1D();
        float x = 0.5f;
        for (int i = 0; i < 10; ++i) { // Newton iterations on the CDF 1 - surfaceSurface
            x += (xi - (1.0f - modelForm(type, param, 0, V3(x)).x))/modelForm(type, param, 1, V3(x)).x;
            x = fmax2(x, 0.0f);
        }
        return x;
    }
    case TGPU_TRANS_DAVIS: { // DavisTransmittance.cpp:53-60
        const float alpha = param[0];
        return startOnSurface ? alpha*(mpow(1.0f - sampler.next1D(), -1.0f/alpha) - 1.0f)
                              : alpha*(mpow(1.0f - sampler.next1D(), -1.0f/(1.0f + alpha)) - 1.0f);
    }
    case TGPU_TRANS_DAVIS_WEINSTEIN: { // DavisWeinsteinTransmittance.cpp:87-117: bisection on the CDF
        float xi = sampler.next1D();
        float step = 1e6;
        float result = step*2;
        while (step > 1e-6) {
            float cdf = 1.0f - modelForm(type, param, startOnSurface ? 0 : 2, V3(result)).x;
            if (cdf > xi) result -= step;
            else result += step;
            step /= 2;
        }
        return result;
    }
    default: break;
    }
    switch (type) {
    case TGPU_TRANS_LINEAR: return startOnSurface ? m.param[0]*sampler.next1D() : m.param[0];
    case TGPU_TRANS_QUADRATIC:
        return startOnSurface ? m.param[0]*(1.0f - std::sqrt(1.0f - sampler.next1D())) : m.param[0]*sampler.next1D();
    case TGPU_TRANS_DOUBLE_EXPONENTIAL: {
        float t = -mlog(1.0f - sampler.next1D());
        float p = startOnSurface ? 0.5f : m.param[0]/(m.param[0] + m.param[1]);
        return sampler.nextBoolean(p) ? t/m.param[0] : t/m.param[1];
    }
    case TGPU_TRANS_PULSE: {
        const float a = m.param[0], b = m.param[1];
        const int numPulses = int(m.param[2]);
        if (!startOnSurface) return a + (0.5f + int(sampler.next1D()*numPulses))/float(numPulses)*(b - a);
        float xi = sampler.next1D()*numPulses*0.5f;
        float delta = 1.0f/numPulses;
        for (int i = 0; i < numPulses; ++i) {
            float h0 = 1.0f - (i + 0.0f)*delta;
            float h1 = 1.0f - (i + 1.0f)*delta;
            xi -= h0*0.5f;
            if (xi < 0.0f) return a + (i + 0.0f + 0.5f*sampler.next1D())*(b - a)*delta;
            xi -= h1*0.5f;
            if (xi < 0.0f) return a + (i + 0.5f + 0.5f*sampler.next1D())*(b - a)*delta;
        }
        return 0.0f;
    }
    default: return -mlog(1.0f - sampler.next1D());
    }
}

bool oracle_ctx::mediumSampleDistance(int medium, Sampler &sampler, const Ray &ray, MediumStateO &state, MediumSampleO &sample) const
{   // HomogeneousMedium.cpp:65-108
    const tgpu_medium &m = sc.media[medium];
    const V3 sigmaT(m.sigma_t), sigmaS(m.sigma_s);
    sample.emission = V3(0.0f);
    if (state.bounce > m.max_bounce) return false;
    float maxT = ray.farT;
    if (m.absorption_only) {
        if (maxT == INF) return false;
        sample.t = maxT;
        sample.weight = transEval(m, sample.t*sigmaT, state.firstScatter, true);
        sample.pdf = 1.0f;
        sample.exited = true;
    } else {
        int component = sampler.nextDiscrete(3);
        float sigmaTc = sigmaT[component];
        float t = transSample(m, sampler, state.firstScatter)/sigmaTc;
        sample.t = fmin2(t, maxT);
        sample.exited = t >= maxT;
        V3 tau = sample.t*sigmaT;
        sample.weight = transEval(m, tau, state.firstScatter, sample.exited);
        if (sample.exited) {
            sample.pdf = transForm(m, state.firstScatter ? 0 : 2, tau).avg();         // surfaceProbability
        } else {
            sample.pdf = (sigmaT*transForm(m, state.firstScatter ? 1 : 3, tau)).avg(); // sigmaT*mediumPdf
            sample.weight *= sigmaS*transSigmaBar(m);
        }
        sample.weight /= sample.pdf;
        state.advance();
        TRC("  med comp=%d t=%08x exited=%d w=%08x %08x %08x pdf=%08x\n", component, hx(sample.t), int(sample.exited), HX3(sample.weight), hx(sample.pdf));
    }
    sample.p = ray.pos + sample.t*ray.dir;
    return true;
}

V3 oracle_ctx::mediumTransmittance(int medium, const Ray &ray, bool startOnSurface, bool endOnSurface) const
{   // HomogeneousMedium.cpp:110-117
    if (ray.farT == INF) return V3(0.0f);
    return transEval(sc.media[medium], V3(sc.media[medium].sigma_t)*ray.farT, startOnSurface, endOnSurface);
}

float oracle_ctx::phaseEval(const tgpu_medium &m, const V3 &wi, const V3 &wo) const
{   // IsotropicPhaseFunction.cpp:24-27 / HenyeyGreensteinPhaseFunction.cpp:19-23,40-43 (eval == pdf for both)
    if (m.phase == 0) return INV_FOUR_PI;
    const float g = m.phase_g;
    float cosTheta = wi.dot(wo);
    float term = 1.0f + g*g - 2.0f*g*cosTheta;
    return INV_FOUR_PI*(1.0f - g*g)/(term*std::sqrt(term));
}

V3 oracle_ctx::phaseSample(const tgpu_medium &m, Sampler &sampler, const V3 &wi, float &pdf) const
{   // IsotropicPhaseFunction.cpp:29-35 / HenyeyGreensteinPhaseFunction.cpp:45-66 ; weight = 1
    float xi0 = sampler.next1D(), xi1 = sampler.next1D();
    const float g = m.phase_g;
    if (m.phase == 0 || g == 0.0f) {
        float phi = xi0*TWO_PI; // SampleWarp::uniformSphere (SampleWarp.hpp:96-107)
        float z = xi1*2.0f - 1.0f;
        float r = std::sqrt(fmax2(1.0f - z*z, 0.0f));
        pdf = INV_FOUR_PI;
        return V3(mcos(phi)*r, msin(phi)*r, z);
    }
    float phi = xi0*TWO_PI;
    float q = (1.0f - g*g)/(1.0f + g*(xi1*2.0f - 1.0f));
    float cosTheta = (1.0f + g*g - q*q)/(2.0f*g);
    float sinTheta = std::sqrt(fmax2(1.0f - cosTheta*cosTheta, 0.0f));
    Frame f = duffFrame(wi);
    float term = 1.0f + g*g - 2.0f*g*cosTheta;
    pdf = INV_FOUR_PI*(1.0f - g*g)/(term*std::sqrt(term));
    return f.toGlobal(V3(mcos(phi)*sinTheta, msin(phi)*sinTheta, cosTheta));
}

int oracle_ctx::selectMedium(const Info &info, int current, bool geometricBackside) const
{   // Primitive::selectMedium (primitives/Primitive.hpp:172-183)
    int im = -1, em = -1;
    if (info.kind == PRIM_TRI && sc.tri_media) { im = sc.tri_media[size_t(info.index)*2]; em = sc.tri_media[size_t(info.index)*2 + 1]; }
    else if (info.kind == PRIM_QUAD) { im = sc.quads[info.index].int_medium; em = sc.quads[info.index].ext_medium; }
    else if (info.kind == PRIM_CUBE) { im = sc.cubes[info.index].int_medium; em = sc.cubes[info.index].ext_medium; }
    else if (info.kind == PRIM_SPHERE) { im = sc.spheres[info.index].int_medium; em = sc.spheres[info.index].ext_medium; }
    else if (info.kind == PRIM_DISK) { im = sc.disks[info.index].int_medium; em = sc.disks[info.index].ext_medium; }
    if (im >= 0 || em >= 0) return geometricBackside ? im : em;
    return current;
}

V3 oracle_ctx::volumeEstimateDirect(Sampler &sampler, const MediumSampleO &ms, int medium, int bounce, const Ray &parent, Counters &c) const
{   // TraceBase.cpp:470-481 -> volumeSampleDirect :402-414 -> volumeLightSample :323-354, volumePhaseSample :356-381
    float weight;
    const tgpu_light *light = chooseLight(ms.p, sampler, weight);
    if (!light) return V3(0.0f);
    V3 result(0.0f);
    { // volumeLightSample
        LightSample ls;
        V3 lightF(0.0f);
        if (lightSampleDirect(*light, ms.p, sampler, ls)) {
            V3 f(phaseEval(sc.media[medium], parent.dir, ls.d)); // phase->eval(parentRay.dir(), lightSample.d)
            Ray ray{ms.p, ls.d, 0.0f, INF};
            Temp data;
            Info info;
            V3 e = attenuatedEmission(*light, medium, ls.dist, data, info, bounce, false, ray, c);
            TRC("  volL d=%08x %08x %08x dist=%08x pdf=%08x f=%08x e=%08x %08x %08x farT=%08x\n", HX3(ls.d), hx(ls.dist), hx(ls.pdf), hx(f.x), HX3(e), hx(ray.farT));
            if (!(e == 0.0f)) {
                lightF = f*e/ls.pdf;
                float pp = phaseEval(sc.media[medium], parent.dir, ls.d); // phase->pdf
                lightF *= (ls.pdf*ls.pdf)/(ls.pdf*ls.pdf + pp*pp);
            }
        }
        result = lightF;
    }
    { // volumePhaseSample (lights in scope are never Dirac)
        float pp;
        V3 w = phaseSample(sc.media[medium], sampler, parent.dir, pp);
        Ray ray{ms.p, w, 0.0f, INF};
        Temp data;
        Info info;
        V3 e = attenuatedEmission(*light, medium, -1.0f, data, info, bounce, false, ray, c);
        TRC("  volB w=%08x %08x %08x pp=%08x e=%08x %08x %08x farT=%08x\n", HX3(w), hx(pp), HX3(e), hx(ray.farT));
        if (!(e == 0.0f)) {
            V3 phaseF = e*V3(1.0f); // phaseSample.weight
            float lp = lightDirectPdf(*light, data, info, ms.p);
            phaseF *= (pp*pp)/(pp*pp + lp*lp);
            result += phaseF;
        }
    }
    return result*weight;
}

bool oracle_ctx::handleVolume(Sampler &sampler, const MediumSampleO &ms, int medium, int bounce, bool enableLightSampling, Ray &ray, V3 &throughput,
                              V3 &emission, bool &wasSpecular, Counters &c) const
{   // TraceBase.cpp:496-514
    wasSpecular = !enableLightSampling;
    if (enableLightSampling && bounce < sc.settings.max_bounces - 1)
        emission += throughput*volumeEstimateDirect(sampler, ms, medium, bounce + 1, ray, c);
    float pp;
    V3 w = phaseSample(sc.media[medium], sampler, ray.dir, pp);
    ray = Ray{ms.p, w, 0.0f, INF};
    throughput *= V3(1.0f);
    return true;
}

void oracle_ctx::cameraRay(uint32_t px, uint32_t py, float xi0, float xi1, Ray &ray) const
{   // cameras/PinholeCamera.cpp:53-86 ; ReconstructionFilter::sample (ReconstructionFilter.hpp:86-103,152-169)
    const tgpu_camera &cam = sc.camera;
    float uvx, uvy;
    if (cam.filter_type == TGPU_FILTER_DIRAC) {
        uvx = uvy = 0.0f;
    } else if (cam.filter_type == TGPU_FILTER_BOX) {
        uvx = xi0 - 0.5f;
        uvy = xi1 - 0.5f;
    } else {
        auto sample1 = [&](float xi) {
            const int R = 31;
            bool negative = xi < 0.5f;
            xi = negative ? xi*2.0f : (xi - 0.5f)*2.0f;
            int idx = R - 1;
            for (int i = 0; i < R - 1; ++i) {
                if (xi < cam.filter_cdf[i]) {
                    idx = i;
                    break;
                }
            }
            float pdf = cam.filter_cdf[idx] - cam.filter_cdf[idx - 1];
            float u = cam.filter_bin_size*(idx + (xi - cam.filter_cdf[idx - 1])/pdf);
            if (negative) u = -u;
            return u;
        };
        uvx = sample1(xi0);
        uvy = sample1(xi1);
    }
    V3 localD = V3(-1.0f + (float(px) + 0.5f + uvx)*2.0f*cam.pixel_size_x,
                   cam.ratio - (float(py) + 0.5f + uvy)*2.0f*cam.pixel_size_x,
                   cam.plane_dist).normalized();
    ray.pos = V3(cam.pos);
    ray.dir = mulVector(cam.rot, localD);
    ray.nearT = 1e-4f; // math/Ray.hpp:24
    ray.farT = INF;
}

V3 oracle_ctx::traceSample(uint32_t px, uint32_t py, Sampler &sampler, Counters &c, AovSample *aov) const
{   // integrators/path_tracer/PathTracer.cpp:14-149 (low_order_scattering = include_surfaces = true); aov: _trackOutputValues
    c.samples++;
    float xi0 = sampler.next1D(), xi1 = sampler.next1D();
    Ray ray;
    cameraRay(px, py, xi0, xi1, ray);
    V3 throughput(1.0f); // point.weight*direction.weight
    Temp data;
    Info info;
    V3 emission(0.0f);
    const int maxBounces = sc.settings.max_bounces, minBounces = sc.settings.min_bounces;
    const bool nee = sc.settings.enable_light_sampling != 0, volumeNee = sc.settings.enable_volume_light_sampling != 0;
    MediumSampleO mediumSample;
    MediumStateO state;
    state.reset();
    int medium = sc.n_media ? sc.camera.medium : -1;

    bool recordedOutputValues = false;
    float hitDistance = 0.0f;
    int mediumBounces = 0;
    int bounce = 0;
    c.extend++;
    bool didHit = intersect(ray, data, info, c);
    bool wasSpecular = true;
    TRC("ORC start px=%u py=%u idx=%u scr=%08x O=%08x %08x %08x D=%08x %08x %08x\n", px, py, sampler.index, sampler.scramble, HX3(ray.pos), HX3(ray.dir));
    while ((didHit || medium >= 0) && bounce < maxBounces) {
        TRC("ORC v%d hit=%d t=%08x u=%08x v=%08x kind=%d index=%u inst=%d med=%d tp=%08x %08x %08x em=%08x %08x %08x dim=%u pcg=%016llx\n", bounce, int(didHit), hx(ray.farT),
            hx(data.u), hx(data.v), didHit ? info.kind : 0, didHit ? info.index : 0u, didHit ? data.inst : -1, medium, HX3(throughput), HX3(emission), sampler.dimension,
            (unsigned long long)sampler.supplemental->state);
        bool hitSurface = true;
        if (medium >= 0) {
            if (!mediumSampleDistance(medium, sampler, ray, state, mediumSample)) return emission;
            emission += throughput*mediumSample.emission;
            throughput *= mediumSample.weight;
            hitSurface = mediumSample.exited;
            if (hitSurface && !didHit) break;
        }
        if (hitSurface) {
            hitDistance += ray.farT;
            Event surfaceEvent = makeLocalScatterEvent(data, info, ray, &sampler);
            V3 transmittance(-1.0f);
            bool terminate = !handleSurface(surfaceEvent, data, info, medium, bounce, nee, ray, throughput, emission, wasSpecular, state, c, &transmittance);
            if (aov && !recordedOutputValues && (!wasSpecular || terminate)) { // PathTracer.cpp:78-96
                aov->depth = hitDistance;
                aov->normal = info.Ns;
                const tgpu_bsdf &bd = sc.bsdfs[info.bsdf]; // a TransparencyBsdf shows its base's albedo
                V3 albedo = texEval(bd.type == TGPU_BSDF_TRANSPARENCY ? sc.bsdfs[bd.substrate].albedo_tex : bd.albedo_tex, info.uvx, info.uvy);
                if (primEmissive(info.kind, info.index)) albedo += primEvalDirect(data, info);
                aov->albedo = albedo;
                if (transmittance != -1.0f) aov->visibility = transmittance.avg();
                recordedOutputValues = true;
            }
            if (terminate) return emission;
        } else {
            mediumBounces++;
            if (!handleVolume(sampler, mediumSample, medium, bounce, volumeNee, ray, throughput, emission, wasSpecular, c)) return emission;
        }

        if (throughput.max() == 0.0f) break;

        float roulettePdf = vabs(throughput).max();
        if (bounce > 2 && roulettePdf < 0.1f) {
            if (sampler.nextBoolean(roulettePdf))
                throughput /= roulettePdf;
            else
                return emission;
        }
        if (std::isnan(ray.dir.sum() + ray.pos.sum())) return V3(0.0f);
        if (std::isnan(throughput.sum() + emission.sum())) return V3(0.0f);
        TRC("ORC   -> O=%08x %08x %08x D=%08x %08x %08x tp=%08x %08x %08x em=%08x %08x %08x dim=%u pcg=%016llx\n", HX3(ray.pos), HX3(ray.dir), HX3(throughput), HX3(emission),
            sampler.dimension, (unsigned long long)sampler.supplemental->state);

        bounce++;
        if (bounce < maxBounces) {
            c.extend++;
            didHit = intersect(ray, data, info, c);
        }
    }
    (void)mediumBounces;
    bool lastIsInfinite = false; // info.primitive && info.primitive->isInfinite() after handleInfiniteLights
    V3 infiniteEmission(0.0f);
    if (bounce >= minBounces && bounce < maxBounces) {
        // TraceBase::handleInfiniteLights (TraceBase.cpp:570-578) / TraceableScene::intersectInfinites (:194-209)
        if (sc.env.present) {
            Temp d2;
            Info i2;
            d2.kind = PRIM_ENV; d2.index = 0; d2.p = ray.pos; d2.w = ray.dir;
            i2.w = ray.dir;
            intersectionInfo(d2, i2);
            if (!nee || wasSpecular || !sc.env.samplable)
                emission += throughput*primEvalDirect(d2, i2);
            if (aov) {
                lastIsInfinite = true;
                infiniteEmission = primEvalDirect(d2, i2);
            }
        }
    }
    if (std::isnan(throughput.sum() + emission.sum())) return V3(0.0f);
    if (aov && !recordedOutputValues) { // PathTracer.cpp:133-140
        if (bounce == 0) aov->depth = 0.0f;
        aov->normal = -ray.dir;
        if (lastIsInfinite) aov->albedo = infiniteEmission;
    }
    TRC("ORC end bounce=%d em=%08x %08x %08x\n", bounce, HX3(emission));
    return emission;
}

// ========================================================================================
// C API
// ========================================================================================
// ray batches are split over the host's threads (the brute-force mode is O(rays x triangles))
template <class F>
static void parallelRanges(uint32_t n, F body)
{
    uint32_t nt = std::max(1u, std::min(std::thread::hardware_concurrency(), (n + 255u)/256u));
    std::vector<std::thread> th;
    for (uint32_t t = 1; t < nt; ++t) th.emplace_back([=] { body(uint32_t(uint64_t(n)*t/nt), uint32_t(uint64_t(n)*(t + 1)/nt)); });
    body(0u, uint32_t(uint64_t(n)/nt));
    for (auto &t : th) t.join();
}

extern "C" {

oracle_ctx *oracle_create(const tgpu_scene_desc *scene, int rng_mode, int math_mode, int rcp_mode)
{
    if (!scene || scene->abi_version != TGPU_ABI_VERSION) return nullptr;
    oracle_ctx *ctx = new oracle_ctx();
    ctx->sc = *scene; // shallow: caller keeps the buffers alive
    ctx->rngMode = rng_mode;
    ctx->mathMode = math_mode;
    ctx->rcpMode = rcp_mode;
    ctx->hitPad = TGPU_HIT_PAD_FRACTION*tgpu_bvh_pad(&ctx->sc);
    ctx->prepareTris();
    ctx->buildBvh(0, ctx->sc.n_tris, ctx->sceneBvh);
    ctx->buildInstances();
    ctx->buildExpTable();
    ctx->buildReferenceAccel();
    return ctx;
}

void oracle_destroy(oracle_ctx *ctx) { delete ctx; }
void oracle_set_brute_force(oracle_ctx *ctx, int on) { if (ctx) ctx->brute = on != 0; }

static void addSample(float *mean, uint32_t *count, const V3 &c) // cameras/OutputBuffer.hpp:104-132 (single buffer)
{
    if (isnanV(c) || isinfV(c)) return;
    uint32_t sampleIdx = (*count)++;
    V3 m(mean);
    V3 r = m + (c - m)/float(sampleIdx + 1);
    mean[0] = r.x; mean[1] = r.y; mean[2] = r.z;
}

// OutputBuffer<T>::addSample (cameras/OutputBuffer.hpp:104-132) for T = float (channels 1) and T = Vec3f (channels 3, elementwise):
// running means in one or two half buffers (two_buffer_variance), Welford sums against the combined mean (sample_variance)
static void obAddSample(const oracle_outbuf &ob, int channels, size_t idx, const float *c)
{
    for (int k = 0; k < channels; ++k)
        if (std::isnan(c[k]) || std::isinf(c[k])) return;
    uint32_t sampleIdx = ob.count[idx]++;
    float *A = ob.a + idx*channels, *B = ob.b ? ob.b + idx*channels : nullptr, *VAR = ob.variance ? ob.variance + idx*channels : nullptr;
    for (int k = 0; k < channels; ++k) {
        if (VAR) {
            float curr;
            if (B && sampleIdx > 0) {
                uint32_t sampleCountA = (sampleIdx + 1)/2, sampleCountB = sampleIdx/2;
                curr = (A[k]*float(sampleCountA) + B[k]*float(sampleCountB))/float(sampleIdx);
            } else {
                curr = A[k];
            }
            float delta = c[k] - curr;
            curr += delta/float(sampleIdx + 1);
            VAR[k] += delta*(c[k] - curr);
        }
        if (B) {
            float *feature = (sampleIdx & 1) ? B : A;
            uint32_t perBufferSampleCount = sampleIdx/2 + 1;
            feature[k] += (c[k] - feature[k])/float(perBufferSampleCount);
        } else {
            A[k] += (c[k] - A[k])/float(sampleIdx + 1);
        }
    }
}
static const int kOutputChannels[TGPU_OUTPUT_COUNT] = {3, 1, 3, 3, 1};
static void addOutputSamples(const oracle_ctx *ctx, size_t idx, const V3 &col, const AovSample &aov)
{
    const oracle_outbuf *ob = ctx->out;
    if (ob[TGPU_OUTPUT_COLOR].a) { float c[3] = {col.x, col.y, col.z}; obAddSample(ob[TGPU_OUTPUT_COLOR], 3, idx, c); }
    if (ob[TGPU_OUTPUT_DEPTH].a) obAddSample(ob[TGPU_OUTPUT_DEPTH], 1, idx, &aov.depth);
    if (ob[TGPU_OUTPUT_NORMAL].a) { float c[3] = {aov.normal.x, aov.normal.y, aov.normal.z}; obAddSample(ob[TGPU_OUTPUT_NORMAL], 3, idx, c); }
    if (ob[TGPU_OUTPUT_ALBEDO].a) { float c[3] = {aov.albedo.x, aov.albedo.y, aov.albedo.z}; obAddSample(ob[TGPU_OUTPUT_ALBEDO], 3, idx, c); }
    if (ob[TGPU_OUTPUT_VISIBILITY].a) obAddSample(ob[TGPU_OUTPUT_VISIBILITY], 1, idx, &aov.visibility);
}
/* Known-answer entry for the output-buffer accumulation (tests/test_ref_kat3.py against tests/golden/ref_kat3.json, produced by the
 * reference's own cameras/OutputBuffer.hpp): sample i goes to pixel pixel[i] with `channels` values; the part arrays are updated in place. */
void oracle_kat_output_buffer(int channels, uint32_t n_samples, const uint32_t *pixel, const float *values, float *a, float *b, float *variance, uint32_t *count)
{
    oracle_outbuf ob{a, b, variance, count};
    for (uint32_t i = 0; i < n_samples; ++i) obAddSample(ob, channels, pixel[i], values + size_t(i)*channels);
}

/* Known-answer entry for the transmittance models (tests/test_ref_kat4.py against tests/golden/ref_kat4.json, produced by the reference's own
 * transmittances/*.cpp): the four forms at n optical depths (forms_out: n x 4 x 3 floats), sigmaBar, and n_samples draws each of
 * sampleSurface / sampleMedium from UniformPathSampler(seed) / (seed + 1).  math_mode 0 = glibc (the reference), 1 = include/tgpu_math.h. */
void oracle_kat_transmittance(int type, const float *param3, int math_mode, const float *tau3, uint32_t n, float *forms_out, float *sigma_bar,
                              uint32_t seed, uint32_t n_samples, float *surface_out, float *medium_out)
{
    oracle_ctx ctx;
    ctx.mathMode = math_mode;
    ctx.buildExpTable();
    for (uint32_t i = 0; i < n; ++i)
        for (int form = 0; form < 4; ++form) {
            V3 f = ctx.modelForm(type, param3, form, V3(tau3 + size_t(i)*3));
            float *o = forms_out + (size_t(i)*4 + form)*3;
            o[0] = f.x; o[1] = f.y; o[2] = f.z;
        }
    *sigma_bar = ctx.modelSigmaBar(type, param3);
    for (int start = 0; start < 2; ++start) {
        Pcg32 pcg(seed + uint32_t(start)); // UniformPathSampler(seed): state = seed, sequence 0 (sampling/UniformSampler.hpp:22-26)
        Sampler s;
        s.sobol = false;
        s.supplemental = &pcg;
        for (uint32_t i = 0; i < n_samples; ++i) (start == 0 ? surface_out : medium_out)[i] = ctx.modelSample(type, param3, s, start == 0);
    }
}

void oracle_bind_output_buffers(oracle_ctx *ctx, const oracle_outbuf *bufs)
{
    ctx->trackOutputs = false;
    for (int i = 0; i < TGPU_OUTPUT_COUNT; ++i) {
        ctx->out[i] = bufs ? bufs[i] : oracle_outbuf{nullptr, nullptr, nullptr, nullptr};
        if (ctx->out[i].a) ctx->trackOutputs = true;
    }
    (void)kOutputChannels;
}

static void flushCounters(oracle_ctx *ctx, const oracle_ctx::Counters &c)
{
    ctx->stSamples += c.samples; ctx->stExtend += c.extend; ctx->stShadow += c.shadow;
    ctx->stSurface += c.surface; ctx->stNodes += c.nodes; ctx->stTris += c.tris;
}

void oracle_tile_seeds(uint32_t res_x, uint32_t res_y, uint32_t seed, uint32_t *out)
{   // PathTraceIntegrator.cpp:27-42,187
    Pcg32 s(hash32(seed));
    uint32_t i = 0;
    for (uint32_t y = 0; y < res_y; y += 16)
        for (uint32_t x = 0; x < res_x; x += 16) out[i++] = hash32(s.nextI());
}

int oracle_render_image(oracle_ctx *ctx, uint32_t seed, uint32_t spp_begin, uint32_t spp_count,
                        float *rgb_mean, uint32_t *sample_count, int n_threads)
{
    const uint32_t W = ctx->sc.camera.res_x, H = ctx->sc.camera.res_y, T = 16;
    uint32_t tilesX = (W + T - 1)/T, tilesY = (H + T - 1)/T, nTiles = tilesX*tilesY;
    std::vector<uint32_t> seeds(nTiles);
    oracle_tile_seeds(W, H, seed, seeds.data());
    if (!ctx->tilesInit || ctx->tileSeedBase != seed || spp_begin == 0) {
        ctx->tilePcg.assign(nTiles, Pcg32());
        for (uint32_t i = 0; i < nTiles; ++i) ctx->tilePcg[i] = Pcg32(seeds[i]); // SobolPathSampler ctor :26-33
        ctx->tilesInit = true;
        ctx->tileSeedBase = seed;
    }
    std::atomic<uint32_t> next{0};
    auto worker = [&]() {
        oracle_ctx::Counters c;
        while (true) {
            uint32_t t = next.fetch_add(1);
            if (t >= nTiles) break;
            uint32_t tx = (t % tilesX)*T, ty = (t/tilesX)*T;
            uint32_t tw = std::min(T, W - tx), th = std::min(T, H - ty);
            Sampler s;
            s.matrices = ctx->sc.sobol_matrices;
            s.sobol = ctx->sc.settings.use_sobol != 0;
            s.seed = seeds[t];
            const bool dumpTile = g_dumpTile && uint32_t(std::atoi(g_dumpTile)) == t;
            uint32_t traceP = ~0u, traceS = ~0u;
            if (g_traceAt) std::sscanf(g_traceAt, "%u:%u", &traceP, &traceS);
            for (uint32_t y = 0; y < th; ++y) { // PathTraceIntegrator::renderTile :136-156
                for (uint32_t x = 0; x < tw; ++x) {
                    uint32_t px = tx + x, py = ty + y, pixelIndex = px + py*W;
                    for (uint32_t i = 0; i < spp_count; ++i) {
                        s.startPath(pixelIndex, spp_begin + i);
                        if (ctx->rngMode == 0) {
                            s.supplemental = &ctx->tilePcg[t];
                        } else {
                            s.local = pathLocalPcg(s.scramble, spp_begin + i);
                            s.supplemental = &s.local;
                        }
                        AovSample aov;
                        if (g_traceAt) g_trace = pixelIndex == traceP && spp_begin + i == traceS;
                        V3 col = ctx->traceSample(px, py, s, c, ctx->trackOutputs ? &aov : nullptr);
                        if (dumpTile) std::fprintf(stderr, "S %u %u %08x %08x %08x\n", pixelIndex, spp_begin + i, hx(col.x), hx(col.y), hx(col.z));
                        addSample(rgb_mean + size_t(pixelIndex)*3, sample_count + pixelIndex, col);
                        if (ctx->trackOutputs) addOutputSamples(ctx, pixelIndex, col, aov);
                    }
                }
            }
        }
        flushCounters(ctx, c);
    };
    int nt = std::max(1, n_threads);
    std::vector<std::thread> th;
    for (int i = 1; i < nt; ++i) th.emplace_back(worker);
    worker();
    for (auto &t : th) t.join();
    return 0;
}

int oracle_render_shard(oracle_ctx *ctx, const tgpu_shard_desc *shard, uint32_t spp_begin, uint32_t spp_count,
                        float *rgb_mean, uint32_t *sample_count, int n_threads)
{
    if (ctx->rngMode != 1) return -1; // the tile-sequential stream is only defined for whole tiles in image order
    const uint32_t W = ctx->sc.camera.res_x;
    std::atomic<uint32_t> next{0};
    const uint32_t chunk = 64;
    auto worker = [&]() {
        oracle_ctx::Counters c;
        Sampler s;
        s.matrices = ctx->sc.sobol_matrices;
        s.sobol = ctx->sc.settings.use_sobol != 0;
        while (true) {
            uint32_t b = next.fetch_add(chunk);
            if (b >= shard->n_pixels) break;
            for (uint32_t k = b; k < std::min(b + chunk, shard->n_pixels); ++k) {
                uint32_t pixelIndex = shard->pixel_index[k];
                s.seed = shard->tile_seed[k];
                for (uint32_t i = 0; i < spp_count; ++i) {
                    s.startPath(pixelIndex, spp_begin + i);
                    s.local = pathLocalPcg(s.scramble, spp_begin + i);
                    s.supplemental = &s.local;
                    AovSample aov;
                    V3 col = ctx->traceSample(pixelIndex % W, pixelIndex/W, s, c, ctx->trackOutputs ? &aov : nullptr);
                    addSample(rgb_mean + size_t(k)*3, sample_count + k, col);
                    if (ctx->trackOutputs) addOutputSamples(ctx, k, col, aov);
                }
            }
        }
        flushCounters(ctx, c);
    };
    int nt = std::max(1, n_threads);
    std::vector<std::thread> th;
    for (int i = 1; i < nt; ++i) th.emplace_back(worker);
    worker();
    for (auto &t : th) t.join();
    return 0;
}

void oracle_kat_embree_user(const float *boxes6, uint32_t n_boxes, const float *rays, uint32_t n_rays, uint32_t *visits, uint32_t max_visits,
                            uint32_t *n_visits, uint32_t *prim, float *tfar)
{
    using namespace embree_restated;
    std::vector<Box> boxes(n_boxes);
    for (uint32_t i = 0; i < n_boxes; ++i) for (int k = 0; k < 3; ++k) { boxes[i].lo[k] = boxes6[i*6 + k]; boxes[i].hi[k] = boxes6[i*6 + 3 + k]; }
    Bvh4 bvh;
    bvh.build(boxes, 4, 1, 1);
    const size_t stride = 8 + size_t(n_boxes);
    for (uint32_t r = 0; r < n_rays; ++r) {
        const float *rr = rays + r*stride;
        Bvh4::Ray er;
        for (int k = 0; k < 3; ++k) { er.org[k] = rr[k]; er.dir[k] = rr[3 + k]; }
        er.tnear = rr[6];
        er.tfar = rr[7];
        uint32_t nv = 0, hitPrim = 0xFFFFFFFFu;
        bvh.intersect(er, true, [&](uint32_t first, uint32_t, Bvh4::Ray &ray) {
            uint32_t id = bvh.prims()[first];
            if (nv < max_visits) visits[size_t(r)*max_visits + nv] = id;
            nv++;
            float t = rr[8 + id];
            if (t < 0.0f || t < ray.tnear || t > ray.tfar) return;
            ray.tfar = t;
            hitPrim = id;
        });
        n_visits[r] = nv;
        prim[r] = hitPrim;
        tfar[r] = er.tfar;
    }
}

void oracle_kat_embree_tris(const float *verts3, uint32_t n_verts, const uint32_t *tris3, uint32_t n_tris, const float *rays8, uint32_t n_rays,
                            uint32_t *prim, float *tuv3)
{
    using namespace embree_restated;
    (void)n_verts;
    std::vector<Box> boxes(n_tris);
    std::vector<TriIsect> tri(n_tris);
    for (uint32_t i = 0; i < n_tris; ++i) {
        V3 p0(verts3 + size_t(tris3[i*3])*3), p1(verts3 + size_t(tris3[i*3 + 1])*3), p2(verts3 + size_t(tris3[i*3 + 2])*3);
        boxes[i] = emptyBox();
        for (int k = 0; k < 3; ++k) { boxes[i].lo[k] = std::min(p0[k], std::min(p1[k], p2[k])); boxes[i].hi[k] = std::max(p0[k], std::max(p1[k], p2[k])); }
        tri[i].v0 = p0; tri[i].e1 = p0 - p1; tri[i].e2 = p2 - p0; tri[i].Ng = tri[i].e1.cross(tri[i].e2);
    }
    Bvh4 bvh;
    bvh.build(boxes, 4, 4, 28);
    for (uint32_t r = 0; r < n_rays; ++r) {
        const float *rr = rays8 + size_t(r)*8;
        Ray ray{V3(rr), V3(rr + 3), rr[6], rr[7]};
        Bvh4::Ray er;
        for (int k = 0; k < 3; ++k) { er.org[k] = rr[k]; er.dir[k] = rr[3 + k]; }
        er.tnear = rr[6];
        er.tfar = rr[7];
        uint32_t hitPrim = 0xFFFFFFFFu;
        float u = 0, v = 0;
        bvh.intersect(er, false, [&](uint32_t first, uint32_t count, Bvh4::Ray &q) {
            for (uint32_t block = 0; block < count; block += 4) {
                int best = -1;
                TriHit bh{false, 0, 0, 0};
                for (uint32_t lane = 0; lane < 4 && block + lane < count; ++lane) {
                    TriHit h = triTest(tri[bvh.prims()[first + block + lane]], ray, q.tfar, 1);
                    if (h.valid && (best < 0 || h.t < bh.t)) { best = int(lane); bh = h; }
                }
                if (best >= 0) { q.tfar = bh.t; hitPrim = bvh.prims()[first + block + uint32_t(best)]; u = bh.u; v = bh.v; }
            }
        });
        prim[r] = hitPrim;
        tuv3[size_t(r)*3] = er.tfar; tuv3[size_t(r)*3 + 1] = u; tuv3[size_t(r)*3 + 2] = v;
    }
}

int oracle_trace_closest(oracle_ctx *ctx, const float *rays, uint32_t n, float *hits)
{
    parallelRanges(n, [=](uint32_t begin, uint32_t end) {
    oracle_ctx::Counters c;
    for (uint32_t i = begin; i < end; ++i) {
        const float *r = rays + size_t(i)*8;
        Ray ray{V3(r[0], r[1], r[2]), V3(r[4], r[5], r[6]), r[3], r[7]};
        Temp data;
        Info info;
        bool hit = ctx->intersect(ray, data, info, c);
        uint32_t prim = 0xFFFFFFFFu;
        float u = 0, v = 0;
        if (hit) {
            uint32_t kind = data.kind == PRIM_TRI ? 0u : data.kind == PRIM_QUAD ? 1u : data.kind == PRIM_SPHERE ? 4u : data.kind == PRIM_DISK ? 5u : 2u;
            prim = (kind << 29) | data.index; // tgpu_trace_closest's encoding: kind << 29 | index
            if (data.kind == PRIM_TRI || data.kind == PRIM_QUAD) { u = data.u; v = data.v; }
            if (data.kind == PRIM_DISK) u = data.rSq;
            if (data.inst >= 0) prim = (3u << 29) | uint32_t(data.inst); // same convention as tgpu_trace_closest
        }
        float *h = hits + size_t(i)*4;
        h[0] = hit ? ray.farT : r[7];
        h[1] = u;
        h[2] = v;
        std::memcpy(&h[3], &prim, 4);
    }
    flushCounters(ctx, c);
    });
    return 0;
}

int oracle_trace_occluded(oracle_ctx *ctx, const float *rays, uint32_t n, uint8_t *occluded)
{
    parallelRanges(n, [=](uint32_t begin, uint32_t end) {
    oracle_ctx::Counters c;
    for (uint32_t i = begin; i < end; ++i) {
        const float *r = rays + size_t(i)*8;
        Ray ray{V3(r[0], r[1], r[2]), V3(r[4], r[5], r[6]), r[3], r[7]};
        Temp data;
        Info info;
        occluded[i] = ctx->intersect(ray, data, info, c) ? 1 : 0;
    }
    flushCounters(ctx, c);
    });
    return 0;
}

uint32_t oracle_hash32(uint32_t x) { return hash32(x); }

void oracle_fast_exp(const oracle_ctx *ctx, const float *x3, float *out3)
{
    V3 e = ctx->fastExp(V3(x3[0], x3[1], x3[2]));
    out3[0] = e.x; out3[1] = e.y; out3[2] = e.z;
}
void oracle_exp_table(const oracle_ctx *ctx, uint32_t *out1024) { std::memcpy(out1024, ctx->expTable, sizeof(ctx->expTable)); }


uint32_t oracle_sobol(const oracle_ctx *ctx, uint32_t index, uint32_t dim, uint32_t scramble)
{
    uint32_t result = scramble;
    for (uint32_t i = dim*32u; index; index >>= 1, ++i)
        if (index & 1) result ^= ctx->sc.sobol_matrices[i];
    return result;
}

void oracle_pcg_stream(uint64_t state, uint64_t seq, uint32_t n, uint32_t *out)
{
    Pcg32 p(state, seq);
    for (uint32_t i = 0; i < n; ++i) out[i] = p.nextI();
}

void oracle_camera_ray(const oracle_ctx *ctx, uint32_t px, uint32_t py, float xi0, float xi1, float *o)
{
    Ray r;
    ctx->cameraRay(px, py, xi0, xi1, r);
    o[0] = r.pos.x; o[1] = r.pos.y; o[2] = r.pos.z; o[3] = r.dir.x; o[4] = r.dir.y; o[5] = r.dir.z;
}

/* Known-answer entry for the instance BVH2 (tests/test_oracle_kat.py against tests/golden/ref_kat2.json, produced by the reference's
 * own BvhBuilder / BinaryBvh): builds the tree over `n` boxes (6 floats each: min, max; primitive id = index, centroid = box centre)
 * and walks one ray (origin, direction, nearT 1e-4, farT infinity); every visit (id, bits(tMin)) goes to out[2*k], out[2*k + 1], and
 * a visit of an id divisible by 5 shortens the ray to tMin + 0.75 as a hit would.  Returns the number of visits. */
uint32_t oracle_kat_binary_bvh(const float *boxes6, uint32_t n, const float *origin3, const float *dir3, uint32_t *out, uint32_t maxVisits)
{
    IPrims prims;
    for (uint32_t i = 0; i < n; ++i) {
        IPrim p;
        p.box = IBox();
        p.box.grow(V3(boxes6 + size_t(i)*6));
        p.box.grow(V3(boxes6 + size_t(i)*6 + 3));
        p.centroid = p.box.center(); p.id = i; p.area = p.box.area();
        prims.push_back(p);
    }
    InstSet s;
    buildBinaryBvh(s, prims);
    Ray ray{V3(origin3), V3(dir3), 1e-4f, INF};
    uint64_t nodes = 0;
    uint32_t k = 0;
    walkBinaryBvh(s, ray, nodes, [&](uint32_t id, float tMin) {
        if (k < maxVisits) { out[2*k] = id; out[2*k + 1] = tgpu_float_to_bits(tMin); }
        ++k;
        if (id % 5 == 0) ray.farT = tMin + 0.75f;
    });
    return k;
}

void oracle_get_stats(oracle_ctx *ctx, oracle_stats *out)
{
    out->samples = ctx->stSamples; out->extend_rays = ctx->stExtend; out->shadow_rays = ctx->stShadow;
    out->surface_events = ctx->stSurface; out->node_visits = ctx->stNodes; out->tri_tests = ctx->stTris;
}

void oracle_reset_stats(oracle_ctx *ctx)
{
    ctx->stSamples = 0; ctx->stExtend = 0; ctx->stShadow = 0; ctx->stSurface = 0; ctx->stNodes = 0; ctx->stTris = 0;
}

} // extern "C"

// ---- leaf functions for the known-answer tests (tests/test_oracle_kat.py) ----------------------------
extern "C" {

static oracle_ctx *leafCtx(int math_mode)
{
    static oracle_ctx c0, c1;
    oracle_ctx *c = math_mode ? &c1 : &c0;
    c->mathMode = math_mode;
    return c;
}

float oracle_normalized_uint(uint32_t i) { return normalizedUint(i); }

void oracle_cosine_hemisphere(int math_mode, float xix, float xiy, float *out3)
{   // sampling/SampleWarp.hpp:42-52
    const oracle_ctx *o = leafCtx(math_mode);
    float phi = xix*TWO_PI;
    float r = std::sqrt(xiy);
    out3[0] = o->mcos(phi)*r;
    out3[1] = o->msin(phi)*r;
    out3[2] = std::sqrt(fmax2(1.0f - xiy, 0.0f));
}

// out: m.xyz, D(m), pdf(m), G1(v, m)
void oracle_microfacet(int math_mode, int dist, float alpha, float xix, float xiy, const float *v3, float *out6)
{
    MF mf{leafCtx(math_mode), dist};
    V3 m = mf.sample(alpha, xix, xiy);
    out6[0] = m.x; out6[1] = m.y; out6[2] = m.z;
    out6[3] = mf.D(alpha, m);
    out6[4] = mf.pdf(alpha, m);
    out6[5] = mf.G1(alpha, V3(v3), m);
}

void oracle_conductor_reflectance(const float *eta3, const float *k3, float cosThetaI, float *out3)
{
    V3 f = conductorReflectance3(eta3, k3, cosThetaI);
    out3[0] = f.x; out3[1] = f.y; out3[2] = f.z;
}

void oracle_dielectric_reflectance(float eta, float cosThetaI, float *out2)
{
    float ct;
    out2[0] = dielectricReflectance(eta, cosThetaI, ct);
    out2[1] = ct;
}

void oracle_tangent_frame(const float *n3, float *out6)
{
    Frame f = duffFrame(V3(n3));
    out6[0] = f.tangent.x; out6[1] = f.tangent.y; out6[2] = f.tangent.z;
    out6[3] = f.bitangent.x; out6[4] = f.bitangent.y; out6[5] = f.bitangent.z;
}

// include/tgpu_math.h evaluated on the host: kind as tgpu_eval_math
void oracle_eval_math(int kind, const float *a, const float *b, uint32_t n, float *out)
{
    for (uint32_t i = 0; i < n; ++i) {
        float x = a[i], y = b[i], r;
        switch (kind) {
        case 0: r = tgpu_sinf(x); break;
        case 1: r = tgpu_cosf(x); break;
        case 2: r = tgpu_expf(x); break;
        case 3: r = tgpu_logf(x); break;
        case 4: r = tgpu_atan2f(x, y); break;
        case 5: r = tgpu_acosf(x); break;
        case 6: r = x/y; break;
        case 7: r = std::sqrt(x); break;
        default: r = 0.0f; break;
        }
        out[i] = r;
    }
}

// Embree's rcp() on this CPU (RCPPS is implementation defined): lets a test decide whether the
// bit-exact reference fixtures (generated on an Intel Xeon) apply to this host.
float oracle_embree_rcp(float a)
{
#if defined(__SSE__)
    float r = _mm_cvtss_f32(_mm_rcp_ss(_mm_set_ss(a)));
    return r*(2.0f - r*a);
#else
    return 1.0f/a;
#endif
}

} // extern "C"

// ========================================================================================
// The whole StandaloneRenderer pass loop with the reference's adaptive sampling
// (src/tungsten/Shared.hpp:283-311 ; PathTraceIntegrator.cpp:44-134,136-156,184-239 ;
//  SampleRecord.hpp:44-66 ; Integrator.cpp:51-54)
// ========================================================================================
namespace {
struct SampleRecord { // path_tracer/SampleRecord.hpp:11-66
    uint32_t sampleCount = 0, nextSampleCount = 0, sampleIndex = 0;
    float adaptiveWeight = 0.0f, mean = 0.0f, runningVariance = 0.0f;
    void addSample(float x)
    {
        sampleCount++;
        float delta = x - mean;
        mean += delta/sampleCount;
        runningVariance += delta*(x - mean);
    }
    float variance() const { return runningVariance/(sampleCount - 1); }
    float errorEstimate() const { return variance()/(sampleCount*fmax2(mean*mean, 1e-3f)); }
};
} // namespace

extern "C" int oracle_render_shard_adaptive(oracle_ctx *ctx, const tgpu_shard_desc *shard, const uint32_t *vtIndex,
                                            const uint32_t *vtCount, float *rgb_mean, uint32_t *sample_count,
                                            tgpu_sample_record *records, int n_threads)
{
    if (ctx->rngMode != 1) return -1;
    const uint32_t W = ctx->sc.camera.res_x, H = ctx->sc.camera.res_y, VT = 4;
    const uint32_t varW = (W + VT - 1)/VT;
    // shard-local index of every image pixel, and the variance tiles the shard holds
    std::vector<int32_t> local(size_t(W)*H, -1);
    std::vector<uint32_t> tiles;
    for (uint32_t k = 0; k < shard->n_pixels; ++k) {
        uint32_t p = shard->pixel_index[k];
        local[p] = int32_t(k);
        tiles.push_back((p % W)/VT + ((p/W)/VT)*varW);
    }
    std::sort(tiles.begin(), tiles.end());
    tiles.erase(std::unique(tiles.begin(), tiles.end()), tiles.end());
    std::atomic<uint32_t> next{0};
    auto worker = [&]() {
        oracle_ctx::Counters c;
        Sampler s;
        s.matrices = ctx->sc.sobol_matrices;
        s.sobol = ctx->sc.settings.use_sobol != 0;
        while (true) {
            uint32_t t = next.fetch_add(1);
            if (t >= tiles.size()) break;
            uint32_t vt = tiles[t], vx = (vt % varW)*VT, vy = (vt/varW)*VT;
            SampleRecord record;
            record.sampleCount = records[vt].sample_count;
            record.mean = records[vt].mean;
            record.runningVariance = records[vt].running_variance;
            // renderTile order restricted to this variance tile: rows, then pixels, then the pixel's samples (:136-156)
            for (uint32_t y = vy; y < std::min(vy + VT, H); ++y)
                for (uint32_t x = vx; x < std::min(vx + VT, W); ++x) {
                    uint32_t pixelIndex = x + y*W;
                    int32_t k = local[pixelIndex];
                    if (k < 0) continue;
                    s.seed = shard->tile_seed[k];
                    for (uint32_t i = 0; i < vtCount[vt]; ++i) {
                        s.startPath(pixelIndex, vtIndex[vt] + i);
                        s.local = pathLocalPcg(s.scramble, vtIndex[vt] + i);
                        s.supplemental = &s.local;
                        AovSample aov;
                        V3 col = ctx->traceSample(x, y, s, c, ctx->trackOutputs ? &aov : nullptr);
                        record.addSample(col.luminance());
                        addSample(rgb_mean + size_t(k)*3, sample_count + k, col);
                        if (ctx->trackOutputs) addOutputSamples(ctx, size_t(k), col, aov);
                    }
                }
            records[vt] = tgpu_sample_record{record.sampleCount, record.mean, record.runningVariance};
        }
        flushCounters(ctx, c);
    };
    int nt = std::max(1, n_threads);
    std::vector<std::thread> th;
    for (int i = 1; i < nt; ++i) th.emplace_back(worker);
    worker();
    for (auto &t : th) t.join();
    return 0;
}

extern "C" int oracle_render_reference(oracle_ctx *ctx, uint32_t seed, uint32_t totalSpp, uint32_t sppStep, int adaptive,
                                       float *rgb_mean, uint32_t *sample_count, float *record_out /* varW*varH*3 or NULL */, int n_threads)
{
    const uint32_t W = ctx->sc.camera.res_x, H = ctx->sc.camera.res_y, T = 16, VT = 4, AdaptiveThreshold = 16;
    const uint32_t tilesX = (W + T - 1)/T, tilesY = (H + T - 1)/T, nTiles = tilesX*tilesY;
    const uint32_t varW = (W + VT - 1)/VT, varH = (H + VT - 1)/VT;
    Pcg32 integratorSampler(hash32(seed)); // PathTraceIntegrator::prepareForRender :187
    std::vector<uint32_t> seeds(nTiles);
    for (uint32_t i = 0; i < nTiles; ++i) seeds[i] = hash32(integratorSampler.nextI()); // diceTiles :29-38
    std::vector<Pcg32> tilePcg(nTiles);
    for (uint32_t i = 0; i < nTiles; ++i) tilePcg[i] = Pcg32(seeds[i]);
    std::vector<SampleRecord> samples(size_t(varW)*varH);

    uint32_t currentSpp = 0, nextSpp = std::min(sppStep, totalSpp);
    while (currentSpp < totalSpp) { // Shared.hpp:283 while (!integrator.done())
        // generateWork (:110-134)
        for (SampleRecord &r : samples) r.sampleIndex += r.nextSampleCount;
        int sppCount = int(nextSpp - currentSpp);
        bool work = true;
        if (adaptive && currentSpp >= AdaptiveThreshold) {
            // errorPercentile95 (:44-59)
            std::vector<float> errors;
            errors.reserve(samples.size());
            for (SampleRecord &r : samples) {
                r.adaptiveWeight = r.errorEstimate();
                if (r.adaptiveWeight > 0.0f) errors.push_back(r.adaptiveWeight);
            }
            float maxError = 0.0f;
            if (!errors.empty()) {
                std::sort(errors.begin(), errors.end());
                maxError = errors[(errors.size()*95)/100];
            }
            if (maxError == 0.0f) {
                work = false;
            } else {
                for (SampleRecord &r : samples) r.adaptiveWeight = fmin2(r.adaptiveWeight, maxError);
                // dilateAdaptiveWeights (:61-85)
                for (uint32_t y = 0; y < varH; ++y)
                    for (uint32_t x = 0; x < varW; ++x) {
                        size_t idx = x + size_t(y)*varW;
                        if (y < varH - 1) samples[idx].adaptiveWeight = fmax2(samples[idx].adaptiveWeight, samples[idx + varW].adaptiveWeight);
                        if (x < varW - 1) samples[idx].adaptiveWeight = fmax2(samples[idx].adaptiveWeight, samples[idx + 1].adaptiveWeight);
                    }
                for (int y = int(varH) - 1; y >= 0; --y)
                    for (int x = int(varW) - 1; x >= 0; --x) {
                        size_t idx = size_t(x) + size_t(y)*varW;
                        if (y > 0) samples[idx].adaptiveWeight = fmax2(samples[idx].adaptiveWeight, samples[idx - varW].adaptiveWeight);
                        if (x > 0) samples[idx].adaptiveWeight = fmax2(samples[idx].adaptiveWeight, samples[idx - 1].adaptiveWeight);
                    }
                // distributeAdaptiveSamples (:87-108)
                double totalWeight = 0.0;
                for (SampleRecord &r : samples) totalWeight += r.adaptiveWeight;
                int adaptiveBudget = (sppCount - 1)*int(W)*int(H);
                int budgetPerTile = adaptiveBudget/int(VT*VT);
                float weightToSampleFactor = float(double(budgetPerTile)/totalWeight);
                float pixelPdf = 0.0f;
                for (SampleRecord &r : samples) {
                    float fractionalSamples = r.adaptiveWeight*weightToSampleFactor;
                    int adaptiveSamples = int(fractionalSamples);
                    pixelPdf += fractionalSamples - float(adaptiveSamples);
                    if (integratorSampler.next1D() < pixelPdf) {
                        adaptiveSamples++;
                        pixelPdf -= 1.0f;
                    }
                    r.nextSampleCount = uint32_t(adaptiveSamples + 1);
                }
            }
        } else {
            for (SampleRecord &r : samples) r.nextSampleCount = uint32_t(sppCount);
        }
        if (work) {
            // renderTile for every tile (:136-156); tiles own disjoint records, so they can run concurrently
            std::atomic<uint32_t> next{0};
            auto worker = [&]() {
                oracle_ctx::Counters c;
                while (true) {
                    uint32_t t = next.fetch_add(1);
                    if (t >= nTiles) break;
                    uint32_t tx = (t % tilesX)*T, ty = (t/tilesX)*T;
                    uint32_t tw = std::min(T, W - tx), th = std::min(T, H - ty);
                    Sampler s;
                    s.matrices = ctx->sc.sobol_matrices;
                    s.sobol = ctx->sc.settings.use_sobol != 0;
                    s.seed = seeds[t];
                    for (uint32_t y = 0; y < th; ++y)
                        for (uint32_t x = 0; x < tw; ++x) {
                            uint32_t px = tx + x, py = ty + y, pixelIndex = px + py*W;
                            SampleRecord &record = samples[px/VT + size_t(py/VT)*varW];
                            int spp = int(record.nextSampleCount);
                            for (int i = 0; i < spp; ++i) {
                                s.startPath(pixelIndex, record.sampleIndex + uint32_t(i));
                                if (ctx->rngMode == 0) {
                                    s.supplemental = &tilePcg[t];
                                } else {
                                    s.local = pathLocalPcg(s.scramble, record.sampleIndex + uint32_t(i));
                                    s.supplemental = &s.local;
                                }
                                V3 col = ctx->traceSample(px, py, s, c);
                                record.addSample(col.luminance());
                                addSample(rgb_mean + size_t(pixelIndex)*3, sample_count + pixelIndex, col);
                            }
                        }
                }
                flushCounters(ctx, c);
            };
            int nt = std::max(1, n_threads);
            std::vector<std::thread> th;
            for (int i = 1; i < nt; ++i) th.emplace_back(worker);
            worker();
            for (auto &t : th) t.join();
        }
        currentSpp = nextSpp;                              // completion callback (:231-235)
        nextSpp = std::min(currentSpp + sppStep, totalSpp); // Integrator::advanceSpp
    }
    if (record_out)
        for (size_t i = 0; i < samples.size(); ++i) {
            record_out[i*3] = float(samples[i].sampleCount);
            record_out[i*3 + 1] = samples[i].mean;
            record_out[i*3 + 2] = samples[i].runningVariance;
        }
    return 0;
}
