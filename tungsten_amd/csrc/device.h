// Device-side building blocks of the wavefront path tracer: vector math, sampler,
// textures, lights, BSDFs.  Every function states the reference code it
// replaces (paths relative to /root/reference/src/core).  Compiled with
// -ffp-contract=off; the only fused multiply-adds are the explicit ones in
// include/tgpu_math.h and in the BVH slab test, so the arithmetic is bit-for-bit
// what the CPU oracle computes.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/tungsten_amd.h"
#include "../../include/tgpu_math.h"
#include "../../include/tgpu_bounds.h"

#define TG_DEV __device__ __forceinline__

namespace tg {

constexpr float PI = 3.1415926536f; // math/Angle.hpp:8-16
constexpr float TWO_PI = PI*2.0f;
constexpr float INV_PI = 1.0f/PI;
constexpr float INV_TWO_PI = 0.5f*INV_PI;
constexpr float INV_FOUR_PI = 0.25f*INV_PI;
#define TG_INF __builtin_huge_valf()

// ---- Vec3f (math/Vec.hpp:127-193): same operation order as the reference ---------------
struct V3 {
    float x, y, z;
    TG_DEV V3() : x(0), y(0), z(0) {}
    TG_DEV explicit V3(float a) : x(a), y(a), z(a) {}
    TG_DEV V3(float a, float b, float c) : x(a), y(b), z(c) {}
    TG_DEV explicit V3(const float *p) : x(p[0]), y(p[1]), z(p[2]) {}
    TG_DEV V3 operator+(const V3 &o) const { return V3(x + o.x, y + o.y, z + o.z); }
    TG_DEV V3 operator-(const V3 &o) const { return V3(x - o.x, y - o.y, z - o.z); }
    TG_DEV V3 operator*(const V3 &o) const { return V3(x*o.x, y*o.y, z*o.z); }
    TG_DEV V3 operator/(const V3 &o) const { return V3(x/o.x, y/o.y, z/o.z); }
    TG_DEV V3 operator*(float s) const { return V3(x*s, y*s, z*s); }
    TG_DEV V3 operator/(float s) const { return V3(x/s, y/s, z/s); }
    TG_DEV V3 operator+(float s) const { return V3(x + s, y + s, z + s); }
    TG_DEV V3 operator-(float s) const { return V3(x - s, y - s, z - s); }
    TG_DEV V3 operator-() const { return V3(-x, -y, -z); }
    TG_DEV V3 &operator+=(const V3 &o) { x += o.x; y += o.y; z += o.z; return *this; }
    TG_DEV V3 &operator*=(const V3 &o) { x *= o.x; y *= o.y; z *= o.z; return *this; }
    TG_DEV V3 &operator*=(float s) { x *= s; y *= s; z *= s; return *this; }
    TG_DEV V3 &operator/=(float s) { x /= s; y /= s; z /= s; return *this; }
    TG_DEV bool isZero() const { return x == 0.0f && y == 0.0f && z == 0.0f; }
    TG_DEV float dot(const V3 &o) const { float s = x*o.x; s += y*o.y; s += z*o.z; return s; }
    TG_DEV float lengthSq() const { float r = 0.0f; r += x*x; r += y*y; r += z*z; return r; }
    TG_DEV float length() const { return __builtin_sqrtf(lengthSq()); }
    TG_DEV V3 normalized() const { float inv = 1.0f/length(); return V3(x*inv, y*inv, z*inv); }
    TG_DEV V3 cross(const V3 &o) const { return V3(y*o.z - z*o.y, z*o.x - x*o.z, x*o.y - y*o.x); }
    TG_DEV float sum() const { float r = x; r += y; r += z; return r; }
    TG_DEV float avg() const { return sum()*(1.0f/3.0f); }
    TG_DEV float max() const { float m = x; if (y > m) m = y; if (z > m) m = z; return m; }
    TG_DEV float comp(int i) const { return i == 0 ? x : (i == 1 ? y : z); }
};
TG_DEV V3 operator*(float s, const V3 &v) { return V3(v.x*s, v.y*s, v.z*s); }
TG_DEV V3 vabs(const V3 &v) { return V3(__builtin_fabsf(v.x), __builtin_fabsf(v.y), __builtin_fabsf(v.z)); }
TG_DEV float fmin2(float a, float b) { return a < b ? a : b; } // Tungsten::min/max/clamp (math/MathUtil.hpp)
TG_DEV float fmax2(float a, float b) { return a > b ? a : b; }
TG_DEV float clampf(float v, float lo, float hi) { return fmin2(fmax2(v, lo), hi); }
TG_DEV bool isnanf_(float v) { return v != v; }
TG_DEV bool isinff_(float v) { return __builtin_fabsf(v) == TG_INF; }

TG_DEV V3 mulPoint0(const float *m, const V3 &b) // Mat4f*Vec3f, zero translation (math/Mat4f.hpp:320-327)
{
    return V3(m[0]*b.x + m[1]*b.y + m[2]*b.z + 0.0f, m[3]*b.x + m[4]*b.y + m[5]*b.z + 0.0f, m[6]*b.x + m[7]*b.y + m[8]*b.z + 0.0f);
}
TG_DEV V3 mulVector(const float *m, const V3 &b) // Mat4f::transformVector (math/Mat4f.hpp:159-166)
{
    return V3(m[0]*b.x + m[1]*b.y + m[2]*b.z, m[3]*b.x + m[4]*b.y + m[5]*b.z, m[6]*b.x + m[7]*b.y + m[8]*b.z);
}

// ---- BsdfLobes (bsdfs/BsdfLobes.hpp) ------------------------------------------------------
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

// ---- device scene -----------------------------------------------------------------------------
// texels: RGB padded to 16 B (one load per texel); rgbe: the same texels as 4-byte shared-exponent words when every one of them is
// exactly representable that way (a Radiance .hdr file is: ImageIO.cpp:340-365 decodes it with float(m)*2^(e-136)) -- then the
// float copy is not uploaded, the environment map shrinks from 8 MB to 2 MB (it fits the 4 MB L2) and a lookup decodes on the fly
struct DBitmap { int32_t w, h, clamp, linear; float scale; const float4 *texels; float average[3]; const uint32_t *rgbe; };

// "instances" primitives (primitives/Instance.cpp; host build in instance_bvh.h).
// BVH2 node = 4 x float4: x-, y-, z-boxes of both children as (lMin, rMin, lMax, rMax) -- the lanes of
// BinaryBvh's TinyBvhNode -- then bits(left, right, first, count); right == 0xFFFFFFFF marks a leaf.
struct DMaster { int32_t kind; uint32_t ref; }; // mesh: root child reference of its BVH4; quad / cube: index into quads / cubes
struct DInstSet {
    const float4 *nodes;
    const uint32_t *order;   // BinaryBvh::_primIndices
    const float4 *pos;       // xyz = _instancePos, w = bits(master id)
    const float4 *rot;       // _instanceRot as (w, x, y, z)
    const DMaster *masters;
    float boundsMin[3], boundsMax[3];
    uint32_t rootIsLeaf;
    // one 64-byte record per position of `order` (what a leaf's loop visits): (pos.xyz, bits(instance id)), rot (w, x, y, z),
    // (bits(master kind), bits(master ref), -, -), - : entering an instance is one memory round trip, not three (instStep)
    const float4 *entries;
};
constexpr int TG_TOP_STACK = 48; // BinaryBvh::trace needs depth + 1 entries; checked at upload

struct DScene {
    tgpu_camera cam;
    tgpu_settings settings;
    // BVH (bvh_build.h layout)
    const float4 *nodes;
    const float4 *tris;         // BVH-order triangle records, 4 x float4 each: v0 e1 e2 Ng + soup triangle id (64 B, like a node)
    uint32_t rootChild;
    uint32_t nTris;
    int32_t stackEntries;       // worst-case traversal stack depth (3 pushes per BVH4 level)
    int32_t stackLds;           // levels kept in LDS; the rest spills to stackSpill[level - stackLds][global thread]
    uint32_t ldsNodes;          // nodes [0, ldsNodes) of `nodes` (the top levels of the scene's BVH4, breadth first) are staged in LDS by the kernel
                                // this copy of the scene is launched with (0: none); ldsNodeWord: their offset in the dynamic LDS block, in words
    uint32_t ldsNodeWord;
    uint32_t ldsQueueWord;      // k_extend_p: offset (words) of the per-wave buffers of finished rays in the dynamic LDS block
    int32_t topStackLds;        // scenes with "instances": entries of the (node, tMin) stack of the top-level walk, in LDS behind the levels above
    uint32_t spillStride;       // threads of the largest grid
    uint32_t *stackSpill;
    // per-triangle shading records, soup order: 5 x float4
    //   s0 = (n0.xyz, n1.x) s1 = (n1.yz, n2.xy) s2 = (n2.z, uv0.xy, uv1.x) s3 = (uv1.y, uv2.xy, bits(bsdf | smooth<<31))
    //   s4 = (unnormalised Ng = (p1-p0)x(p2-p0), 0)
    const float4 *triShade;
    const uint8_t *triClass; // BSDF class of each triangle (queue selection in the extend kernel)
    const tgpu_quad *quads; uint32_t nQuads;
    const tgpu_cube *cubes; uint32_t nCubes;
    const tgpu_sphere *spheres; uint32_t nSpheres;
    const tgpu_disk *disks; uint32_t nDisks;
    const tgpu_bsdf *bsdfs;
    const tgpu_texture *textures;
    const DBitmap *bitmaps;
    const tgpu_light *lights; uint32_t nLights;
    tgpu_env env;          // pointers are device pointers
    int32_t envConstant;   // emission texture is constant
    const uint32_t *sobol; // 1024 x 32
    // guide records for the env CDF inversion (EnvGuide below): bucket g names the last CDF element <= g/N and carries the CDF / pdf values the
    // forward scan from there reads first, so the std::upper_bound of Distribution2D::warp is ONE 32-byte read in the common case
    const struct EnvGuide *envGuideMarginal;    // dist_h records
    const struct EnvGuide *envGuideConditional; // dist_h x dist_w records
    int32_t envGuideSlack;                      // 1: a bucket of slack absorbs the rounding of value*N (N not a power of two)
    const DInstSet *instSets; uint32_t nInstSets;
    // participating media (media/HomogeneousMedium.cpp) and the table of fmath::exp_ps (thirdparty/fmath/fmath.hpp:90-127)
    const tgpu_medium *media; uint32_t nMedia;
    const uint32_t *expTable; // 1024 mantissas of 2^(i/1024)
    float expA, expB;         // 1024/logf(2), logf(2)/1024
    int32_t hasForward;       // some BSDF has a ForwardLobe: shadow rays may pass through surfaces (TraceBase.cpp:62-125)
    float hitPad;             // a triangle hit counts only inside the triangle's box grown by this much (include/tgpu_bounds.h (3))
};

// prim id encoding shared with the C ABI: kind << 29 | index; 0xFFFFFFFF = miss
constexpr uint32_t PRIM_MISS = 0xFFFFFFFFu;
constexpr uint32_t KIND_TRI = 0u, KIND_QUAD = 1u, KIND_CUBE = 2u, KIND_ENV = 3u /* also: instance index in ray-level results */,
                   KIND_SPHERE = 4u, KIND_DISK = 5u;
constexpr uint32_t PRIM_SHIFT = 29u;
constexpr uint32_t INST_NONE = 0xFFFFFFFFu; // hitInst: set << 24 | instance, or none
TG_DEV uint32_t primKind(uint32_t p) { return p >> PRIM_SHIFT; }
TG_DEV uint32_t mkPrim(uint32_t kind, uint32_t index) { return (kind << PRIM_SHIFT) | index; }
TG_DEV uint32_t primIndex(uint32_t p) { return p & 0x1FFFFFFFu; }

// ---- sampler (sampling/SobolPathSampler.hpp:20-84, UniformSampler.hpp:40-52) --------------------
TG_DEV uint32_t hash32(uint32_t x) // math/MathUtil.hpp:120-128
{
    x = ~x + (x << 15);
    x = x ^ (x >> 12);
    x = x + (x << 2);
    x = x ^ (x >> 4);
    x = x*2057;
    x = x ^ (x >> 16);
    return x;
}
TG_DEV float normalizedUint(uint32_t i) { return tgpu_bits_to_float((i >> 9u) | 0x3F800000u) - 1.0f; } // BitManip.hpp:47-50

struct Sampler {
    const uint32_t *matrices;
    uint64_t pcg;       // path-local supplemental stream (sequence 0 -> increment 1)
    uint32_t scramble, index, dimension;
    int32_t sobol;

    TG_DEV uint32_t pcgNext()
    {
        uint64_t oldState = pcg;
        pcg = oldState*6364136223846793005ULL + 1ULL;
        uint32_t xorShifted = uint32_t(((oldState >> 18u) ^ oldState) >> 27u);
        uint32_t rot = uint32_t(oldState >> 59u);
        return (xorShifted >> rot) | (xorShifted << ((0u - rot) & 31u));
    }
    TG_DEV float pcg1D() { return normalizedUint(pcgNext()); }
    // pcg32_srandom(seed = scramble:sampleIndex, inc = 1): the path-local replacement of the
    // reference's tile-sequential stream (hard part 1)
    TG_DEV void seedPathLocal(uint32_t scr, uint32_t sampleIndex)
    {
        pcg = 0;
        pcgNext();
        pcg += (uint64_t(scr) << 32) | uint64_t(sampleIndex);
        pcgNext();
    }
    TG_DEV bool nextBoolean(float pTrue) { return pcg1D() < pTrue; }
    TG_DEV int nextDiscrete(int numChoices) { return int(pcg1D()*float(numChoices)); } // SobolPathSampler.hpp:62-65
    TG_DEV float next1D()
    {
        if (!sobol || dimension >= 1024u) return pcg1D();
        uint32_t result = scramble; // thirdparty/sobol/sobol.h:40-53
        uint32_t idx = (index & ~0xFFu) | ((index + scramble) & 0xFFu);
        const uint32_t *m = matrices + dimension*32u;
        // columns 0..7 with two 16-byte loads (the low byte of the index is the part that varies from
        // sample to sample), the rare higher bits one column at a time; XOR order is irrelevant
        const uint4 lo = *reinterpret_cast<const uint4 *>(m), hi = *reinterpret_cast<const uint4 *>(m + 4);
        result ^= (idx & 1u ? lo.x : 0u) ^ (idx & 2u ? lo.y : 0u) ^ (idx & 4u ? lo.z : 0u) ^ (idx & 8u ? lo.w : 0u);
        result ^= (idx & 16u ? hi.x : 0u) ^ (idx & 32u ? hi.y : 0u) ^ (idx & 64u ? hi.z : 0u) ^ (idx & 128u ? hi.w : 0u);
        idx >>= 8;
        m += 8;
        while (idx) {
            int b = __builtin_ctz(idx);
            result ^= m[b];
            idx &= idx - 1u;
        }
        dimension++;
        return normalizedUint(result);
    }
};

// ---- textures ---------------------------------------------------------------------------------
TG_DEV V3 loadRgb(const float4 *p) { float4 t = *p; return V3(t.x, t.y, t.z); }
// stb_image's hdr_convert, exact: byte*2^(e - 136); e == 0 is black; e < 10 (denormal scale) never reaches the device (upload keeps floats)
TG_DEV V3 decodeRgbe(uint32_t w)
{
    const uint32_t e = w >> 24;
    if (e == 0u) return V3(0.0f);
    const float f1 = __uint_as_float((e - 9u) << 23);
    return V3(float(w & 0xFFu)*f1, float((w >> 8) & 0xFFu)*f1, float((w >> 16) & 0xFFu)*f1);
}
TG_DEV V3 loadTexel(const DBitmap &bm, size_t idx) { return bm.rgbe ? decodeRgbe(bm.rgbe[idx]) : loadRgb(bm.texels + idx); }
// i mod n for the indices a lookup produces (almost always within one period of [0, n)); same value as ((i % n) + n) % n
TG_DEV int wrapIndex(int i, int n)
{
    if (i < 0) i += n;
    else if (i >= n) i -= n;
    if (i < 0 || i >= n) i = ((i % n) + n) % n;
    return i;
}

TG_DEV V3 bitmapEval(const DBitmap &bm, float ux, float uy) // textures/BitmapTexture.cpp:298-352
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
        iu0 = wrapIndex(iu0, w);
        iu1 = wrapIndex(iu1, w);
        iv0 = wrapIndex(iv0, h);
        iv1 = wrapIndex(iv1, h);
    } else {
        iu0 = min(max(iu0, 0), w - 1);
        iu1 = min(max(iu1, 0), w - 1);
        iv0 = min(max(iv0, 0), h - 1);
        iv1 = min(max(iv1, 0), h - 1);
    }
    if (!linear) return loadTexel(bm, size_t(iu0) + size_t(iv0)*w);
    V3 x00 = loadTexel(bm, size_t(iu0) + size_t(iv0)*w), x01 = loadTexel(bm, size_t(iu1) + size_t(iv0)*w);
    V3 x10 = loadTexel(bm, size_t(iu0) + size_t(iv1)*w), x11 = loadTexel(bm, size_t(iu1) + size_t(iv1)*w);
    V3 l = (x00*(1.0f - u) + x01*u)*(1.0f - v) + (x10*(1.0f - u) + x11*u)*v;
    return bm.scale*l;
}

template <class SC>
TG_DEV V3 texEval(const SC &sc, int tex, float ux, float uy)
{
    const tgpu_texture &t = sc.textures[tex];
    if (t.type == TGPU_TEX_CONSTANT) return V3(t.value);
    if (t.type == TGPU_TEX_CHECKER) { // textures/CheckerTexture.cpp:64-69
        int ui = int(ux*float(t.res_u)), vi = int(uy*float(t.res_v));
        bool on = (ui ^ vi) & 1;
        return on ? V3(t.on_color) : V3(t.off_color);
    }
    return bitmapEval(sc.bitmaps[t.bitmap], ux, uy);
}

// ---- environment light (primitives/InfiniteSphere.cpp) ---------------------------------------------
TG_DEV void directionToUV(const DScene &sc, const V3 &wi, float &ux, float &uy, float *sinTheta) // :27-38
{
    V3 wLocal = mulPoint0(sc.env.inv_rot, wi);
    if (sinTheta) *sinTheta = __builtin_sqrtf(fmax2(1.0f - wLocal.y*wLocal.y, 0.0f));
    ux = tgpu_atan2f(wLocal.z, wLocal.x)*INV_TWO_PI + 0.5f;
    uy = tgpu_acosf(-wLocal.y)*INV_PI;
}

TG_DEV V3 uvToDirection(const DScene &sc, float ux, float uy, float &sinTheta) // :40-50
{
    float phi = (ux - 0.5f)*TWO_PI;
    float theta = uy*PI;
    float sp, cp, st, ct;
    tgpu_sincosf(phi, &sp, &cp);
    tgpu_sincosf(theta, &st, &ct);
    sinTheta = st;
    return mulPoint0(sc.env.rot, V3(cp*sinTheta, -ct, sp*sinTheta));
}

TG_DEV float envPdf(const DScene &sc, float ux, float uy) // BitmapTexture.cpp:452-455, Distribution2D.hpp:66-71
{
    int w = sc.env.dist_w, h = sc.env.dist_h;
    int row = int((1.0f - uy)*h), column = int(ux*w);
    row = min(max(row, 0), h - 1);
    column = min(max(column, 0), w - 1);
    return sc.env.pdf[size_t(row)*w + column]*sc.env.marginal_pdf[row]*w*h;
}

// One bucket of the guide table of a CDF `a` (n entries) with its pdf `p`: idx = (std::upper_bound(a, a + n, g/buckets) - a) - 1, the last
// element <= the bucket's lower end, followed by what a forward scan from idx reads first: a[idx], a[idx + 1], a[idx + 2] and p[idx], p[idx + 1].
// With as many buckets as CDF elements a uniform value crosses half an element on average, so the scan usually ends inside the record; the
// dependent chain guide -> a[i + 1] -> a[i + 2] -> ... -> a[i], p[i] of the plain table (one round trip each) becomes a single read.
struct alignas(32) EnvGuide { uint32_t idx; float a0, a1, a2, p0, p1, pad0, pad1; };

// (std::upper_bound(a, a + n, value) - a) - 1 through the guide records; also hands back a[i] and p[i] of the element found, the values
// Distribution2D::warp reads next -- from the record when the scan ended inside it, from the tables otherwise (same bits either way).
// slack: value*buckets may round up to the next integer when buckets is not a power of two, so such tables start one bucket early.
TG_DEV int upperBoundGuided(const float *a, const float *p, int n, const EnvGuide *guide, int buckets, int slack, float value, float &ai, float &pi)
{
    int g = int(value*float(buckets));
    g = min(max(g - slack, 0), buckets - 1);
    const float4 *rec = reinterpret_cast<const float4 *>(guide + g);
    const float4 r0 = rec[0], r1 = rec[1]; // idx a0 a1 a2 | p0 p1 - -
    int i = int(__float_as_uint(r0.x));
    ai = r0.y;
    pi = r1.x;
    if (i + 1 < n && !(value < r0.z)) {
        ++i;
        ai = r0.z;
        pi = r1.y;
        if (i + 1 < n && !(value < r0.w)) {
            ++i;
            while (i + 1 < n && !(value < a[i + 1])) ++i;
            ai = a[i];
            pi = p[i];
        }
    }
    return i;
}

// std::upper_bound(first, first + n, value) - first
TG_DEV int upperBound(const float *a, int n, float value)
{
    int lo = 0, len = n;
    while (len > 0) {
        int half = len >> 1;
        if (!(value < a[lo + half])) {
            lo += half + 1;
            len -= half + 1;
        } else {
            len = half;
        }
    }
    return lo;
}

struct LightSample { V3 d; float dist, pdf; };

TG_DEV bool envSampleDirect(const DScene &sc, Sampler &s, LightSample &out) // InfiniteSphere.cpp:161-176
{
    if (sc.envConstant) {
        float xi0 = s.next1D(), xi1 = s.next1D();
        float phi = xi0*TWO_PI; // SampleWarp::uniformSphere (sampling/SampleWarp.hpp:96-107)
        float z = xi1*2.0f - 1.0f;
        float r = __builtin_sqrtf(fmax2(1.0f - z*z, 0.0f));
        float sp, cp;
        tgpu_sincosf(phi, &sp, &cp);
        out.d = V3(cp*r, sp*r, z);
        out.dist = TG_INF;
        out.pdf = INV_FOUR_PI;
        return true;
    }
    float uvx = s.next1D(), uvy = s.next1D();
    int w = sc.env.dist_w, h = sc.env.dist_h; // Distribution2D::warp (sampling/Distribution2D.hpp:66-76)
    const float *mc = sc.env.marginal_cdf;
    float cdfAt, pdfAt; // mc[row], marginal_pdf[row]; then rs[column], pdf[row][column]
    int row = upperBoundGuided(mc, sc.env.marginal_pdf, h + 1, sc.envGuideMarginal, h, sc.envGuideSlack & 1, uvy, cdfAt, pdfAt);
    const float pdfRow = pdfAt;
    uvy = clampf((uvy - cdfAt)/pdfAt, 0.0f, 1.0f);
    const float *rs = sc.env.cdf + size_t(row)*(w + 1);
    int column = upperBoundGuided(rs, sc.env.pdf + size_t(row)*w, w + 1, sc.envGuideConditional + size_t(row)*w, w, (sc.envGuideSlack >> 1) & 1, uvx, cdfAt, pdfAt);
    uvx = clampf((uvx - cdfAt)/pdfAt, 0.0f, 1.0f);
    float ux = (uvx + column)/w; // BitmapTexture::sample (:433-439)
    float uy = 1.0f - (uvy + row)/h;
    float sinTheta;
    out.d = uvToDirection(sc, ux, uy, sinTheta);
    // envPdf(sc, ux, uy): the texel it derives from (ux, uy) is nearly always the cell just sampled, whose two pdf values are in registers;
    // only a lane whose rounding lands in a neighbouring cell reads the tables again
    int pr = int((1.0f - uy)*h), pc = int(ux*w);
    pr = min(max(pr, 0), h - 1);
    pc = min(max(pc, 0), w - 1);
    const float texPdf = (pr == row && pc == column) ? pdfAt*pdfRow*w*h : sc.env.pdf[size_t(pr)*w + pc]*sc.env.marginal_pdf[pr]*w*h;
    out.pdf = INV_PI*INV_TWO_PI*texPdf/sinTheta;
    out.dist = TG_INF;
    return out.pdf != 0.0f;
}

// ---- quads (primitives/Quad.cpp) ----------------------------------------------------------------
struct QuadHit { float t, u, v; V3 p; bool backSide; };

TG_DEV bool quadIntersect(const tgpu_quad &q, const V3 &pos, const V3 &dir, float nearT, float farT, QuadHit &h) // :71-98
{
    V3 normal(q.normal), base(q.base);
    float nDotW = dir.dot(normal);
    if (__builtin_fabsf(nDotW) < 1e-6f) return false;
    float t = normal.dot(base - pos)/nDotW;
    if (t < nearT || t > farT) return false;
    V3 qp = pos + t*dir;
    V3 v = qp - base;
    float l0 = v.dot(V3(q.edge0))*q.inv_uv_sq[0];
    float l1 = v.dot(V3(q.edge1))*q.inv_uv_sq[1];
    if (l0 < 0.0f || l0 > 1.0f || l1 < 0.0f || l1 > 1.0f) return false;
    h.t = t; h.u = l0; h.v = l1; h.p = qp; h.backSide = nDotW >= 0.0f;
    return true;
}

TG_DEV bool quadSampleDirect(const tgpu_quad &q, const V3 &p, Sampler &s, LightSample &out) // :172-187
{
    V3 normal(q.normal), base(q.base);
    if (normal.dot(p - base) <= 0.0f) return false;
    float xi0 = s.next1D(), xi1 = s.next1D();
    V3 qp = base + xi0*V3(q.edge0) + xi1*V3(q.edge1);
    out.d = qp - p;
    float rSq = out.d.lengthSq();
    out.dist = __builtin_sqrtf(rSq);
    out.d /= out.dist;
    float cosTheta = -normal.dot(out.d);
    out.pdf = rSq/(cosTheta*q.area);
    return true;
}

// ---- tangent frame (math/TangentFrame.hpp:22-45) ---------------------------------------------------
struct Frame {
    V3 normal, tangent, bitangent;
    TG_DEV V3 toLocal(const V3 &p) const { return V3(tangent.dot(p), bitangent.dot(p), normal.dot(p)); }
    TG_DEV V3 toGlobal(const V3 &p) const { return tangent*p.x + bitangent*p.y + normal*p.z; }
};
TG_DEV Frame duffFrame(const V3 &n)
{
    Frame f;
    f.normal = n;
    float sign = __builtin_copysignf(1.0f, n.z);
    const float a = -1.0f/(sign + n.z);
    const float b = n.x*n.y*a;
    f.tangent = V3(1.0f + sign*n.x*n.x*a, sign*b, -sign*n.x);
    f.bitangent = V3(b, sign + n.y*n.y*a, -n.y);
    return f;
}

// ---- spheres (primitives/Sphere.cpp:68-92,173-194,224-230) -------------------------------------------
TG_DEV bool sphereIntersect(const tgpu_sphere &sp, const V3 &pos, const V3 &dir, float nearT, float farT, float &tHit, bool &backSide)
{
    V3 p = pos - V3(sp.pos);
    float B = p.dot(dir);
    float C = p.lengthSq() - sp.radius*sp.radius;
    float detSq = B*B - C;
    if (detSq >= 0.0f) {
        float det = __builtin_sqrtf(detSq);
        float t = -B - det;
        if (t < farT && t > nearT) { tHit = t; backSide = false; return true; }
        t = -B + det;
        if (t < farT && t > nearT) { tHit = t; backSide = true; return true; }
    }
    return false;
}
TG_DEV bool sphereSampleDirect(const tgpu_sphere &sp, const V3 &p, Sampler &s, LightSample &out)
{
    V3 L = V3(sp.pos) - p;
    float d = L.length();
    float C = d*d - sp.radius*sp.radius;
    if (C <= 0.0f) return false;
    L = L.normalized();
    float cosTheta = __builtin_sqrtf(C)/d;
    float xi0 = s.next1D(), xi1 = s.next1D();
    float phi = xi0*TWO_PI; // SampleWarp::uniformSphericalCap (sampling/SampleWarp.hpp:119-129)
    float z = xi1*(1.0f - cosTheta) + cosTheta;
    float r = __builtin_sqrtf(fmax2(1.0f - z*z, 0.0f));
    float sphi, cphi;
    tgpu_sincosf(phi, &sphi, &cphi);
    V3 dl(cphi*r, sphi*r, z);
    float B = d*dl.z;
    float det = __builtin_sqrtf(fmax2(B*B - C, 0.0f));
    out.dist = B - det;
    out.d = duffFrame(L).toGlobal(dl);
    out.pdf = INV_TWO_PI/(1.0f - cosTheta);
    return true;
}
TG_DEV float sphereDirectPdf(const tgpu_sphere &sp, const V3 &p)
{
    float dist = (V3(sp.pos) - p).length();
    float cosTheta = __builtin_sqrtf(fmax2(dist*dist - sp.radius*sp.radius, 0.0f))/dist;
    return INV_TWO_PI/(1.0f - cosTheta);
}
TG_DEV void sphereUv(const tgpu_sphere &sp, const V3 &Ng, float &u, float &v) // Sphere.cpp:121-125
{
    V3 localN = mulVector(sp.inv_rot, Ng);
    u = tgpu_atan2f(localN.y, localN.x)*INV_TWO_PI + 0.5f;
    v = tgpu_acosf(clampf(localN.z, -1.0f, 1.0f))*INV_PI;
    if (isnanf_(u)) u = 0.0f;
}

// ---- disks (primitives/Disk.cpp:64-86,107-120,177-193,225-232) ---------------------------------------
struct DiskHit { float t, rSq; V3 p; bool backSide; };
TG_DEV bool diskIntersect(const tgpu_disk &dk, const V3 &pos, const V3 &dir, float nearT, float farT, DiskHit &h)
{
    V3 n(dk.n), center(dk.center);
    float nDotW = dir.dot(n);
    float t = n.dot(center - pos)/nDotW;
    if (t < nearT || t > farT) return false;
    V3 q = pos + t*dir;
    V3 v = q - center;
    float rSq = v.lengthSq();
    if (rSq > dk.r*dk.r) return false;
    h.t = t; h.rSq = rSq; h.p = q; h.backSide = -nDotW < dk.cos_apex;
    return true;
}
TG_DEV void diskUv(const tgpu_disk &dk, const V3 &p, float rSq, float &u, float &v)
{
    V3 d = p - V3(dk.center);
    float x = d.dot(V3(dk.bitangent));
    float y = d.dot(V3(dk.tangent));
    v = __builtin_sqrtf(rSq)/dk.r;
    u = (x == 0.0f && y == 0.0f) ? 0.0f : (tgpu_atan2f(y, x)*INV_TWO_PI + 0.5f);
}
TG_DEV bool diskSampleDirect(const tgpu_disk &dk, const V3 &p, Sampler &s, LightSample &out)
{
    V3 n(dk.n), center(dk.center);
    if (n.dot(p - center) < 0.0f) return false;
    float xi0 = s.next1D(), xi1 = s.next1D();
    float phi = xi0*TWO_PI; // SampleWarp::uniformDisk (:64-69)
    float rr = __builtin_sqrtf(xi1);
    float sphi, cphi;
    tgpu_sincosf(phi, &sphi, &cphi);
    float lqx = (cphi*rr)*dk.r, lqy = (sphi*rr)*dk.r;
    V3 q = center + lqx*V3(dk.bitangent) + lqy*V3(dk.tangent);
    out.d = q - p;
    float rSq = out.d.lengthSq();
    out.dist = __builtin_sqrtf(rSq);
    out.d /= out.dist;
    if (-out.d.dot(n) < dk.cos_apex) return false;
    float cosTheta = -n.dot(out.d);
    out.pdf = rSq/(cosTheta*dk.r*dk.r*PI);
    return true;
}

// ---- cubes (primitives/Cube.cpp:94-127) ------------------------------------------------------------
TG_DEV bool cubeIntersect(const tgpu_cube &cb, const V3 &pos, const V3 &dir, float nearT, float farT, float &tHit, bool &backSide)
{
    V3 p = mulPoint0(cb.inv_rot, pos - V3(cb.pos));
    V3 d = mulPoint0(cb.inv_rot, dir);
    float ttMin = nearT, ttMax = farT;
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        float di = d.comp(i), pi = p.comp(i), sc = cb.scale[i];
        float invD = 1.0f/di;
        float relMin = -sc - pi, relMax = sc - pi;
        if (invD >= 0.0f) {
            ttMin = fmax2(ttMin, relMin*invD);
            ttMax = fmin2(ttMax, relMax*invD);
        } else {
            ttMax = fmin2(ttMax, relMin*invD);
            ttMin = fmax2(ttMin, relMax*invD);
        }
    }
    if (ttMin <= ttMax) {
        if (ttMin > nearT && ttMin < farT) { tHit = ttMin; backSide = false; return true; }
        if (ttMax > nearT && ttMax < farT) { tHit = ttMax; backSide = true; return true; }
    }
    return false;
}


// ---- BSDFs ---------------------------------------------------------------------------------------------
struct Event { // samplerecords/SurfaceScatterEvent.hpp:14-66 (info fields folded in)
    Frame frame;
    V3 wi, wo, weight;
    float pdf;
    float uvx, uvy;
    uint32_t requestedLobe, sampledLobe;
    bool flippedFrame;
};

TG_DEV V3 expV3(const V3 &a) { return V3(tgpu_expf(a.x), tgpu_expf(a.y), tgpu_expf(a.z)); }
TG_DEV float roughnessToAlpha(float roughness) { return fmax2(roughness, 1e-3f); } // bsdfs/Microfacet.hpp:27-36

// bsdfs/Microfacet.hpp:38-130 (Beckmann, GGX)
TG_DEV float mfD(int dist, float alpha, const V3 &m)
{
    if (m.z <= 0.0f) return 0.0f;
    float alphaSq = alpha*alpha;
    float cosThetaSq = m.z*m.z;
    float tanThetaSq = fmax2(1.0f - cosThetaSq, 0.0f)/cosThetaSq;
    float cosThetaQu = cosThetaSq*cosThetaSq;
    if (dist == TGPU_DIST_BECKMANN) return INV_PI*tgpu_expf(-tanThetaSq/alphaSq)/(alphaSq*cosThetaQu);
    float s = alphaSq + tanThetaSq;
    return alphaSq*INV_PI/(cosThetaQu*(s*s));
}
TG_DEV float mfG1(int dist, float alpha, const V3 &v, const V3 &m)
{
    if (v.dot(m)*v.z <= 0.0f) return 0.0f;
    if (dist == TGPU_DIST_BECKMANN) {
        float cosThetaSq = v.z*v.z;
        float tanTheta = __builtin_fabsf(__builtin_sqrtf(fmax2(1.0f - cosThetaSq, 0.0f))/v.z);
        float a = 1.0f/(alpha*tanTheta);
        if (a < 1.6f)
            return (3.535f*a + 2.181f*a*a)/(1.0f + 2.276f*a + 2.577f*a*a);
        else
            return 1.0f;
    }
    float alphaSq = alpha*alpha;
    float cosThetaSq = v.z*v.z;
    float tanThetaSq = fmax2(1.0f - cosThetaSq, 0.0f)/cosThetaSq;
    return 2.0f/(1.0f + __builtin_sqrtf(1.0f + alphaSq*tanThetaSq));
}
TG_DEV float mfG(int dist, float alpha, const V3 &i, const V3 &o, const V3 &m) { return mfG1(dist, alpha, i, m)*mfG1(dist, alpha, o, m); }
TG_DEV float mfPdf(int dist, float alpha, const V3 &m) { return mfD(dist, alpha, m)*m.z; }
TG_DEV V3 mfSample(int dist, float alpha, float xix, float xiy)
{
    float phi = xiy*TWO_PI;
    float cosTheta;
    if (dist == TGPU_DIST_BECKMANN) {
        float tanThetaSq = -alpha*alpha*tgpu_logf(1.0f - xix);
        cosTheta = 1.0f/__builtin_sqrtf(1.0f + tanThetaSq);
    } else {
        float tanThetaSq = alpha*alpha*xix/(1.0f - xix);
        cosTheta = 1.0f/__builtin_sqrtf(1.0f + tanThetaSq);
    }
    float r = __builtin_sqrtf(fmax2(1.0f - cosTheta*cosTheta, 0.0f));
    float sp, cp;
    tgpu_sincosf(phi, &sp, &cp);
    return V3(cp*r, sp*r, cosTheta);
}

TG_DEV float conductorReflectance(float eta, float k, float cosThetaI) // bsdfs/Fresnel.hpp:102-125
{
    float cosThetaISq = cosThetaI*cosThetaI;
    float sinThetaISq = fmax2(1.0f - cosThetaISq, 0.0f);
    float sinThetaIQu = sinThetaISq*sinThetaISq;
    float innerTerm = eta*eta - k*k - sinThetaISq;
    float aSqPlusBSq = __builtin_sqrtf(fmax2(innerTerm*innerTerm + 4.0f*eta*eta*k*k, 0.0f));
    float a = __builtin_sqrtf(fmax2((aSqPlusBSq + innerTerm)*0.5f, 0.0f));
    float Rs = ((aSqPlusBSq + cosThetaISq) - (2.0f*a*cosThetaI))/((aSqPlusBSq + cosThetaISq) + (2.0f*a*cosThetaI));
    float Rp = ((cosThetaISq*aSqPlusBSq + sinThetaIQu) - (2.0f*a*cosThetaI*sinThetaISq))/
               ((cosThetaISq*aSqPlusBSq + sinThetaIQu) + (2.0f*a*cosThetaI*sinThetaISq));
    return 0.5f*(Rs + Rs*Rp);
}
TG_DEV V3 conductorReflectance3(const float *eta, const float *k, float c)
{
    return V3(conductorReflectance(eta[0], k[0], c), conductorReflectance(eta[1], k[1], c), conductorReflectance(eta[2], k[2], c));
}
TG_DEV float dielectricReflectance(float eta, float cosThetaI, float &cosThetaT) // bsdfs/Fresnel.hpp:75-92
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
    cosThetaT = __builtin_sqrtf(fmax2(1.0f - sinThetaTSq, 0.0f));
    float Rs = (eta*cosThetaI - cosThetaT)/(eta*cosThetaI + cosThetaT);
    float Rp = (eta*cosThetaT - cosThetaI)/(eta*cosThetaT + cosThetaI);
    return (Rs*Rs + Rp*Rp)*0.5f;
}
TG_DEV bool checkReflectionConstraint(const V3 &wi, const V3 &wo) // bsdfs/Bsdf.hpp:44-47
{
    return __builtin_fabsf(wi.z*wo.z - wi.x*wo.x - wi.y*wo.y - 1.0f) < 1e-3f;
}

TG_DEV uint32_t baseLobes(int type)
{
    if (type == TGPU_BSDF_LAMBERT || type == TGPU_BSDF_OREN_NAYAR) return uint32_t(DiffuseReflectionLobe);
    if (type == TGPU_BSDF_PHONG) return uint32_t(GlossyReflectionLobe | DiffuseReflectionLobe);
    return type == TGPU_BSDF_ROUGH_CONDUCTOR ? uint32_t(GlossyReflectionLobe) : 0u;
}
TG_DEV uint32_t bsdfLobes(const DScene &sc, const tgpu_bsdf &d)
{
    if (d.type == TGPU_BSDF_SMOOTH_COAT) return SpecularReflectionLobe | baseLobes(sc.bsdfs[d.substrate].type);
    if (d.type == TGPU_BSDF_FORWARD) return uint32_t(ForwardLobe); // ForwardBsdf.cpp:9-12
    if (d.type == TGPU_BSDF_MIXED) return baseLobes(sc.bsdfs[d.substrate].type) | baseLobes(sc.bsdfs[d.aux_bsdf].type); // MixedBsdf.cpp:133-136
    if (d.type == TGPU_BSDF_ROUGH_COAT) return GlossyReflectionLobe | baseLobes(sc.bsdfs[d.substrate].type);           // RoughCoatBsdf.cpp:304
    if (d.type == TGPU_BSDF_TRANSPARENCY) return ForwardLobe | baseLobes(sc.bsdfs[d.substrate].type);                  // TransparencyBsdf.cpp:66-69
    if (d.type == TGPU_BSDF_THINSHEET) return uint32_t(SpecularReflectionLobe | ForwardLobe);                          // ThinSheetBsdf.cpp:24
    if (d.type == TGPU_BSDF_MIRROR || d.type == TGPU_BSDF_CONDUCTOR) return uint32_t(SpecularReflectionLobe);
    if (d.type == TGPU_BSDF_PLASTIC) return uint32_t(SpecularReflectionLobe | DiffuseReflectionLobe);
    if (d.type == TGPU_BSDF_ROUGH_PLASTIC) return uint32_t(GlossyReflectionLobe | DiffuseReflectionLobe);
    if (d.type == TGPU_BSDF_DIELECTRIC) // DielectricBsdf.cpp:181-188
        return d.enable_refraction ? uint32_t(SpecularReflectionLobe | SpecularTransmissionLobe) : uint32_t(SpecularReflectionLobe);
    if (d.type == TGPU_BSDF_ROUGH_DIELECTRIC) // RoughDielectricBsdf.cpp:38-41
        return d.enable_refraction ? uint32_t(GlossyReflectionLobe | GlossyTransmissionLobe) : uint32_t(GlossyReflectionLobe);
    return baseLobes(d.type);
}

// ---- RoughDielectricBsdf (bsdfs/RoughDielectricBsdf.cpp:55-280) --------------------------------------
TG_DEV float sgnE(float v) { return v < 0.0f ? -1.0f : 1.0f; } // :13-17
// Bsdf::eval/sample(event, adjoint = false) scale by sqr(eta(event)) (Bsdf.hpp:71-97, RoughDielectricBsdf.cpp:266-272)
TG_DEV float dielectricEtaSq(float ior, const V3 &wi, const V3 &wo)
{
    float eta = wi.z*wo.z >= 0.0f ? 1.0f : (wi.z < 0.0f ? ior : 1.0f/ior);
    return eta*eta;
}
// RoughDielectricBsdf::evalBase / pdfBase / sampleBase: static helpers RoughPlasticBsdf shares (:55-234)
TG_DEV V3 rdEvalBase(const V3 &wi, const V3 &wo, bool sampleR, bool sampleT, float roughness, float ior, int distribution)
{
    float wiDotN = wi.z, woDotN = wo.z;
    bool reflect = wiDotN*woDotN >= 0.0f;
    if ((reflect && !sampleR) || (!reflect && !sampleT)) return V3(0.0f);
    float alpha = roughnessToAlpha(roughness);
    float eta = wiDotN < 0.0f ? ior : 1.0f/ior;
    V3 m;
    if (reflect) m = sgnE(wiDotN)*(wi + wo).normalized();
    else m = -(wi*eta + wo).normalized();
    float wiDotM = wi.dot(m), woDotM = wo.dot(m);
    float cosThetaT;
    float F = dielectricReflectance(1.0f/ior, wiDotM, cosThetaT);
    float G = mfG(distribution, alpha, wi, wo, m);
    float D = mfD(distribution, alpha, m);
    if (reflect) {
        float fr = (F*G*D*0.25f)/__builtin_fabsf(wiDotN);
        return V3(fr);
    }
    float q = eta*wiDotM + woDotM;
    float fs = __builtin_fabsf(wiDotM*woDotM)*(1.0f - F)*G*D/((q*q)*__builtin_fabsf(wiDotN));
    return V3(fs);
}
TG_DEV float rdPdfBase(const V3 &wi, const V3 &wo, bool sampleR, bool sampleT, float roughness, float ior, int distribution)
{
    float wiDotN = wi.z, woDotN = wo.z;
    bool reflect = wiDotN*woDotN >= 0.0f;
    if ((reflect && !sampleR) || (!reflect && !sampleT)) return 0.0f;
    float sampleRoughness = (1.2f - 0.2f*__builtin_sqrtf(__builtin_fabsf(wiDotN)))*roughness;
    float sampleAlpha = roughnessToAlpha(sampleRoughness);
    float eta = wiDotN < 0.0f ? ior : 1.0f/ior;
    V3 m;
    if (reflect) m = sgnE(wiDotN)*(wi + wo).normalized();
    else m = -(wi*eta + wo).normalized();
    float wiDotM = wi.dot(m), woDotM = wo.dot(m);
    float cosThetaT;
    float F = dielectricReflectance(1.0f/ior, wiDotM, cosThetaT);
    float pm = mfPdf(distribution, sampleAlpha, m);
    float pdf;
    if (reflect) {
        pdf = pm*0.25f/__builtin_fabsf(wiDotM);
    } else {
        float q = eta*wiDotM + woDotM;
        pdf = pm*__builtin_fabsf(woDotM)/(q*q);
    }
    if (sampleR && sampleT) {
        if (reflect) pdf *= F; else pdf *= 1.0f - F;
    }
    return pdf;
}
TG_DEV V3 dielectricEval(const DScene &sc, const tgpu_bsdf &d, const V3 &wi, const V3 &wo, uint32_t requested, float uvx, float uvy)
{   // eval :242-249
    bool sampleR = (requested & GlossyReflectionLobe) != 0;
    bool sampleT = (requested & GlossyTransmissionLobe) != 0 && d.enable_refraction;
    float roughness = texEval(sc, d.roughness_tex, uvx, uvy).x;
    V3 f = rdEvalBase(wi, wo, sampleR, sampleT, roughness, d.ior, d.distribution);
    f = f*texEval(sc, d.albedo_tex, uvx, uvy);
    f *= dielectricEtaSq(d.ior, wi, wo);
    return f;
}
TG_DEV float dielectricPdf(const DScene &sc, const tgpu_bsdf &d, const V3 &wi, const V3 &wo, uint32_t requested, float uvx, float uvy)
{
    bool sampleR = (requested & GlossyReflectionLobe) != 0;
    bool sampleT = (requested & GlossyTransmissionLobe) != 0 && d.enable_refraction;
    float roughness = texEval(sc, d.roughness_tex, uvx, uvy).x;
    return rdPdfBase(wi, wo, sampleR, sampleT, roughness, d.ior, d.distribution);
}

// Lambert (bsdfs/LambertBsdf.cpp:27-68) and RoughConductor (bsdfs/RoughConductorBsdf.cpp:60-143):
// the two leaf BSDFs; SmoothCoat layers on top of one of them.
template <class SC>
TG_DEV V3 baseEval(const SC &sc, const tgpu_bsdf &d, const V3 &wi, const V3 &wo, uint32_t requested, float uvx, float uvy)
{
    if (d.type == TGPU_BSDF_LAMBERT) {
        if (!(requested & DiffuseReflectionLobe)) return V3(0.0f);
        if (wi.z <= 0.0f || wo.z <= 0.0f) return V3(0.0f);
        return texEval(sc, d.albedo_tex, uvx, uvy)*INV_PI*wo.z;
    }
    if (d.type == TGPU_BSDF_ROUGH_CONDUCTOR) {
        if (!(requested & GlossyReflectionLobe)) return V3(0.0f);
        if (wi.z <= 0.0f || wo.z <= 0.0f) return V3(0.0f);
        float roughness = texEval(sc, d.roughness_tex, uvx, uvy).x;
        float alpha = roughnessToAlpha(roughness);
        V3 hr = (wi + wo).normalized();
        float cosThetaM = wi.dot(hr);
        V3 F = conductorReflectance3(d.eta, d.k, cosThetaM);
        float G = mfG(d.distribution, alpha, wi, wo, hr);
        float D = mfD(d.distribution, alpha, hr);
        float fr = (G*D*0.25f)/wi.z;
        return texEval(sc, d.albedo_tex, uvx, uvy)*(F*fr);
    }
    if (d.type == TGPU_BSDF_OREN_NAYAR) { // bsdfs/OrenNayarBsdf.cpp:63-98
        if (!(requested & DiffuseReflectionLobe)) return V3(0.0f);
        if (wi.z <= 0.0f || wo.z <= 0.0f) return V3(0.0f);
        float thetaR = tgpu_acosf(wo.z);
        float thetaI = tgpu_acosf(wi.z);
        float alpha = fmax2(thetaR, thetaI);
        float beta = fmin2(thetaR, thetaI);
        float sinAlpha = tgpu_sinf(alpha);
        float denom = (wi.x*wi.x + wi.y*wi.y)*(wo.x*wo.x + wo.y*wo.y);
        float cosDeltaPhi = denom == 0.0f ? 1.0f : (wi.x*wo.x + wi.y*wo.y)/__builtin_sqrtf(denom);
        const float RoughnessToSigma = 1.0f/__builtin_sqrtf(2.0f);
        float sigma = RoughnessToSigma*texEval(sc, d.roughness_tex, uvx, uvy).x;
        float sigmaSq = sigma*sigma;
        float C1 = 1.0f - 0.5f*sigmaSq/(sigmaSq + 0.33f);
        float C2 = 0.45f*sigmaSq/(sigmaSq + 0.09f);
        const float b2 = (2.0f*INV_PI)*beta;
        if (cosDeltaPhi >= 0.0f) C2 *= sinAlpha;
        else C2 *= sinAlpha - b2*b2*b2;
        const float ab = (4.0f*INV_PI*INV_PI)*alpha*beta;
        float C3 = 0.125f*(sigmaSq/(sigmaSq + 0.09f))*(ab*ab);
        float fr1 = (C1 + cosDeltaPhi*C2*tgpu_tanf(beta) + (1.0f - __builtin_fabsf(cosDeltaPhi))*C3*tgpu_tanf(0.5f*(alpha + beta)));
        float fr2 = 0.17f*sigmaSq/(sigmaSq + 0.13f)*(1.0f - cosDeltaPhi*(b2*b2));
        V3 diffuseAlbedo = texEval(sc, d.albedo_tex, uvx, uvy);
        return (diffuseAlbedo*fr1 + diffuseAlbedo*diffuseAlbedo*fr2)*wo.z*INV_PI;
    }
    if (d.type == TGPU_BSDF_PHONG) { // bsdfs/PhongBsdf.cpp:79-100
        bool evalGlossy = (requested & GlossyReflectionLobe) != 0, evalDiffuse = (requested & DiffuseReflectionLobe) != 0;
        if (!evalGlossy && !evalDiffuse) return V3(0.0f);
        if (wi.z <= 0.0f || wo.z <= 0.0f) return V3(0.0f);
        const float brdfFactor = (d.exponent + 2.0f)*INV_TWO_PI;
        float result = 0.0f;
        if (evalDiffuse) result += d.diffuse_ratio*INV_PI;
        if (evalGlossy) {
            float cosTheta = V3(-wi.x, -wi.y, wi.z).dot(wo);
            if (cosTheta > 0.0f) result += tgpu_powf(cosTheta, d.exponent)*brdfFactor*(1.0f - d.diffuse_ratio);
        }
        return texEval(sc, d.albedo_tex, uvx, uvy)*wo.z*result;
    }
    return V3(0.0f);
}

template <class SC>
TG_DEV float basePdf(const SC &sc, const tgpu_bsdf &d, const V3 &wi, const V3 &wo, uint32_t requested, float uvx, float uvy)
{
    if (d.type == TGPU_BSDF_LAMBERT) {
        if (!(requested & DiffuseReflectionLobe)) return 0.0f;
        if (wi.z <= 0.0f || wo.z <= 0.0f) return 0.0f;
        return __builtin_fabsf(wo.z)*INV_PI;
    }
    if (d.type == TGPU_BSDF_ROUGH_CONDUCTOR) {
        if (!(requested & GlossyReflectionLobe)) return 0.0f;
        if (wi.z <= 0.0f || wo.z <= 0.0f) return 0.0f;
        float roughness = texEval(sc, d.roughness_tex, uvx, uvy).x;
        float sampleAlpha = roughnessToAlpha(roughness);
        V3 hr = (wi + wo).normalized();
        return mfPdf(d.distribution, sampleAlpha, hr)*0.25f/wi.dot(hr);
    }
    if (d.type == TGPU_BSDF_OREN_NAYAR) { // bsdfs/OrenNayarBsdf.cpp:125-135
        if (!(requested & DiffuseReflectionLobe)) return 0.0f;
        if (wi.z <= 0.0f || wo.z <= 0.0f) return 0.0f;
        float ratio = clampf(texEval(sc, d.roughness_tex, uvx, uvy).x, 0.01f, 1.0f);
        return INV_TWO_PI*ratio + (__builtin_fabsf(wo.z)*INV_PI)*(1.0f - ratio);
    }
    if (d.type == TGPU_BSDF_PHONG) { // bsdfs/PhongBsdf.cpp:102-121
        bool evalGlossy = (requested & GlossyReflectionLobe) != 0, evalDiffuse = (requested & DiffuseReflectionLobe) != 0;
        if (!evalGlossy && !evalDiffuse) return 0.0f;
        if (wi.z <= 0.0f || wo.z <= 0.0f) return 0.0f;
        const float pdfFactor = (d.exponent + 1.0f)*INV_TWO_PI;
        float result = 0.0f;
        if (evalGlossy) {
            float cosTheta = V3(-wi.x, -wi.y, wi.z).dot(wo);
            if (cosTheta > 0.0f) result += tgpu_powf(cosTheta, d.exponent)*pdfFactor;
        }
        if (evalDiffuse && evalGlossy) result = result*(1.0f - d.diffuse_ratio) + d.diffuse_ratio*(__builtin_fabsf(wo.z)*INV_PI);
        else if (evalDiffuse) result = __builtin_fabsf(wo.z)*INV_PI;
        return result;
    }
    return 0.0f;
}

// samples with incident direction e.wi; fills e.wo, e.pdf, e.weight, e.sampledLobe
template <class SC>
TG_DEV bool baseSample(const SC &sc, const tgpu_bsdf &d, Event &e, Sampler &s)
{
    if (d.type == TGPU_BSDF_LAMBERT) {
        if (!(e.requestedLobe & DiffuseReflectionLobe)) return false;
        if (e.wi.z <= 0.0f) return false;
        float xix = s.next1D(), xiy = s.next1D();
        float phi = xix*TWO_PI; // SampleWarp::cosineHemisphere (sampling/SampleWarp.hpp:42-52)
        float r = __builtin_sqrtf(xiy);
        float sp, cp;
        tgpu_sincosf(phi, &sp, &cp);
        e.wo = V3(cp*r, sp*r, __builtin_sqrtf(fmax2(1.0f - xiy, 0.0f)));
        e.pdf = __builtin_fabsf(e.wo.z)*INV_PI;
        e.weight = texEval(sc, d.albedo_tex, e.uvx, e.uvy);
        e.sampledLobe = DiffuseReflectionLobe;
        return true;
    }
    if (d.type == TGPU_BSDF_ROUGH_CONDUCTOR) {
        if (!(e.requestedLobe & GlossyReflectionLobe)) return false;
        if (e.wi.z <= 0.0f) return false;
        float roughness = texEval(sc, d.roughness_tex, e.uvx, e.uvy).x;
        float alpha = roughnessToAlpha(roughness);
        float sampleAlpha = alpha;
        float xix = s.next1D(), xiy = s.next1D();
        V3 m = mfSample(d.distribution, sampleAlpha, xix, xiy);
        float wiDotM = e.wi.dot(m);
        e.wo = 2.0f*wiDotM*m - e.wi;
        if (wiDotM <= 0.0f || e.wo.z <= 0.0f) return false;
        float G = mfG(d.distribution, alpha, e.wi, e.wo, m);
        float D = mfD(d.distribution, alpha, m);
        float mPdf = mfPdf(d.distribution, sampleAlpha, m);
        float pdf = mPdf*0.25f/wiDotM;
        float weight = wiDotM*G*D/(e.wi.z*mPdf);
        V3 F = conductorReflectance3(d.eta, d.k, wiDotM);
        e.pdf = pdf;
        e.weight = texEval(sc, d.albedo_tex, e.uvx, e.uvy)*(F*weight);
        e.sampledLobe = GlossyReflectionLobe;
        return true;
    }
    if (d.type == TGPU_BSDF_OREN_NAYAR) { // bsdfs/OrenNayarBsdf.cpp:41-61
        if (!(e.requestedLobe & DiffuseReflectionLobe)) return false;
        if (e.wi.z <= 0.0f) return false;
        float ratio = clampf(texEval(sc, d.roughness_tex, e.uvx, e.uvy).x, 0.01f, 1.0f);
        const bool uniform = s.nextBoolean(ratio);
        float xix = s.next1D(), xiy = s.next1D();
        float sp, cp;
        tgpu_sincosf(xix*TWO_PI, &sp, &cp);
        if (uniform) { // SampleWarp::uniformHemisphere (SampleWarp.hpp:25-30)
            float r = __builtin_sqrtf(fmax2(1.0f - xiy*xiy, 0.0f));
            e.wo = V3(cp*r, sp*r, xiy);
        } else {
            float r = __builtin_sqrtf(xiy);
            e.wo = V3(cp*r, sp*r, __builtin_sqrtf(fmax2(1.0f - xiy, 0.0f)));
        }
        e.pdf = INV_TWO_PI*ratio + (__builtin_fabsf(e.wo.z)*INV_PI)*(1.0f - ratio);
        e.weight = baseEval(sc, d, e.wi, e.wo, e.requestedLobe, e.uvx, e.uvy)/e.pdf;
        e.sampledLobe = DiffuseReflectionLobe;
        return e.wo.z > 0.0f;
    }
    if (d.type == TGPU_BSDF_PHONG) { // bsdfs/PhongBsdf.cpp:39-77
        bool evalGlossy = (e.requestedLobe & GlossyReflectionLobe) != 0, evalDiffuse = (e.requestedLobe & DiffuseReflectionLobe) != 0;
        if (!evalGlossy && !evalDiffuse) return false;
        if (e.wi.z <= 0.0f) return false;
        bool sampleGlossy = (evalGlossy && evalDiffuse) ? s.nextBoolean(1.0f - d.diffuse_ratio) : evalGlossy;
        float xix = s.next1D(), xiy = s.next1D();
        float sp, cp;
        tgpu_sincosf(xix*TWO_PI, &sp, &cp);
        if (sampleGlossy) {
            float cosTheta = tgpu_powf(xiy, 1.0f/(1.0f + d.exponent));
            float sinTheta = __builtin_sqrtf(fmax2(0.0f, 1.0f - cosTheta*cosTheta));
            Frame lobe = duffFrame(V3(-e.wi.x, -e.wi.y, e.wi.z));
            e.wo = lobe.toGlobal(V3(cp*sinTheta, sp*sinTheta, cosTheta));
            if (e.wo.z < 0.0f) return false;
            e.sampledLobe = GlossyReflectionLobe;
        } else {
            float r = __builtin_sqrtf(xiy);
            e.wo = V3(cp*r, sp*r, __builtin_sqrtf(fmax2(1.0f - xiy, 0.0f)));
            e.sampledLobe = DiffuseReflectionLobe;
        }
        e.pdf = basePdf(sc, d, e.wi, e.wo, e.requestedLobe, e.uvx, e.uvy);
        e.weight = baseEval(sc, d, e.wi, e.wo, e.requestedLobe, e.uvx, e.uvy)/e.pdf;
        return true;
    }
    return false;
}

TG_DEV bool rdSampleBase(Event &e, Sampler &s, bool sampleR, bool sampleT, float roughness, float ior, int distribution)
{
    float wiDotN = e.wi.z;
    float eta = wiDotN < 0.0f ? ior : 1.0f/ior;
    float sampleRoughness = (1.2f - 0.2f*__builtin_sqrtf(__builtin_fabsf(wiDotN)))*roughness;
    float alpha = roughnessToAlpha(roughness);
    float sampleAlpha = roughnessToAlpha(sampleRoughness);
    float xix = s.next1D(), xiy = s.next1D();
    V3 m = mfSample(distribution, sampleAlpha, xix, xiy);
    float pm = mfPdf(distribution, sampleAlpha, m);
    if (pm < 1e-10f) return false;
    float wiDotM = e.wi.dot(m);
    float cosThetaT = 0.0f;
    float F = dielectricReflectance(1.0f/ior, wiDotM, cosThetaT);
    float etaM = wiDotM < 0.0f ? ior : 1.0f/ior;
    bool reflect;
    if (sampleR && sampleT) {
        reflect = s.nextBoolean(F);
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
    float G = mfG(distribution, alpha, e.wi, e.wo, m);
    float D = mfD(distribution, alpha, m);
    e.weight = V3(__builtin_fabsf(wiDotM)*G*D/(__builtin_fabsf(wiDotN)*pm));
    if (reflect) {
        e.pdf = pm*0.25f/__builtin_fabsf(wiDotM);
        e.sampledLobe = GlossyReflectionLobe;
    } else {
        float q = eta*wiDotM + woDotM;
        e.pdf = pm*__builtin_fabsf(woDotM)/(q*q);
        e.sampledLobe = GlossyTransmissionLobe;
    }
    if (sampleR && sampleT) {
        if (reflect) e.pdf *= F; else e.pdf *= 1.0f - F;
    } else {
        if (reflect) e.weight *= F; else e.weight *= 1.0f - F;
    }
    return true;
}
TG_DEV bool dielectricSample(const DScene &sc, const tgpu_bsdf &d, Event &e, Sampler &s)
{   // sample :232-240
    bool sampleR = (e.requestedLobe & GlossyReflectionLobe) != 0;
    bool sampleT = (e.requestedLobe & GlossyTransmissionLobe) != 0 && d.enable_refraction;
    float roughness = texEval(sc, d.roughness_tex, e.uvx, e.uvy).x;
    if (!rdSampleBase(e, s, sampleR, sampleT, roughness, d.ior, d.distribution)) return false;
    e.weight *= texEval(sc, d.albedo_tex, e.uvx, e.uvy);
    e.weight *= dielectricEtaSq(d.ior, e.wi, e.wo);
    return true;
}

// ---- PlasticBsdf (bsdfs/PlasticBsdf.cpp:45-187) and RoughPlasticBsdf (bsdfs/RoughPlasticBsdf.cpp:55-250) ----------
TG_DEV V3 plasticSubstrate(const DScene &sc, const tgpu_bsdf &d, float Fi, float Fo, float eta, float uvx, float uvy)
{   // ((1 - Fi)*(1 - Fo)*eta*eta)*(albedo/(1 - albedo*diffuseFresnel))
    V3 diffuseAlbedo = texEval(sc, d.albedo_tex, uvx, uvy);
    return ((1.0f - Fi)*(1.0f - Fo)*eta*eta)*(diffuseAlbedo/(V3(1.0f) - diffuseAlbedo*d.diffuse_fresnel));
}
TG_DEV V3 plasticAbsorption(const tgpu_bsdf &d, float wiz, float woz)
{
    V3 ssa(d.scaled_sigma_a);
    return expV3(ssa*(-1.0f/woz - 1.0f/wiz));
}
TG_DEV V3 plasticEval(const DScene &sc, const tgpu_bsdf &d, const V3 &wi, const V3 &wo, uint32_t requested, float uvx, float uvy)
{
    const bool rough = d.type == TGPU_BSDF_ROUGH_PLASTIC;
    bool evalR = (requested & (rough ? GlossyReflectionLobe : SpecularReflectionLobe)) != 0;
    bool evalT = (requested & DiffuseReflectionLobe) != 0;
    if (rough && !evalR && !evalT) return V3(0.0f);
    if (wi.z <= 0.0f || wo.z <= 0.0f) return V3(0.0f);
    float eta = 1.0f/d.ior;
    float ct;
    float Fi = dielectricReflectance(eta, wi.z, ct);
    float Fo = dielectricReflectance(eta, wo.z, ct);
    V3 diffuseAlbedo = texEval(sc, d.albedo_tex, uvx, uvy);
    V3 diffuse = ((1.0f - Fi)*(1.0f - Fo)*eta*eta*wo.z*INV_PI)*(diffuseAlbedo/(V3(1.0f) - diffuseAlbedo*d.diffuse_fresnel));
    if (V3(d.scaled_sigma_a).max() > 0.0f) diffuse *= plasticAbsorption(d, wi.z, wo.z);
    if (!rough) {
        if (evalR && checkReflectionConstraint(wi, wo)) return V3(Fi);
        return evalT ? diffuse : V3(0.0f);
    }
    V3 glossyR(0.0f);
    if (evalR) glossyR = rdEvalBase(wi, wo, true, false, texEval(sc, d.roughness_tex, uvx, uvy).x, d.ior, d.distribution);
    return glossyR + (evalT ? diffuse : V3(0.0f));
}
TG_DEV float plasticPdf(const DScene &sc, const tgpu_bsdf &d, const V3 &wi, const V3 &wo, uint32_t requested, float uvx, float uvy)
{
    const bool rough = d.type == TGPU_BSDF_ROUGH_PLASTIC;
    bool sampleR = (requested & (rough ? GlossyReflectionLobe : SpecularReflectionLobe)) != 0;
    bool sampleT = (requested & DiffuseReflectionLobe) != 0;
    if (rough && !sampleR && !sampleT) return 0.0f;
    if (wi.z <= 0.0f || wo.z <= 0.0f) return 0.0f;
    float ct;
    float Fi = dielectricReflectance(1.0f/d.ior, wi.z, ct);
    if (!rough) {
        if (sampleR && sampleT) {
            float substrateWeight = d.avg_transmittance*(1.0f - Fi);
            float specularWeight = Fi;
            float specularProbability = specularWeight/(specularWeight + substrateWeight);
            if (checkReflectionConstraint(wi, wo)) return specularProbability;
            return (__builtin_fabsf(wo.z)*INV_PI)*(1.0f - specularProbability);
        } else if (sampleT) {
            return __builtin_fabsf(wo.z)*INV_PI;
        } else if (sampleR) {
            return checkReflectionConstraint(wi, wo) ? 1.0f : 0.0f;
        }
        return 0.0f;
    }
    float glossyPdf = 0.0f;
    if (sampleR) glossyPdf = rdPdfBase(wi, wo, true, false, texEval(sc, d.roughness_tex, uvx, uvy).x, d.ior, d.distribution);
    float diffusePdf = 0.0f;
    if (sampleT) diffusePdf = __builtin_fabsf(wo.z)*INV_PI;
    if (sampleT && sampleR) {
        float substrateWeight = d.substrate_weight*d.avg_transmittance*(1.0f - Fi);
        float specularWeight = Fi;
        float specularProbability = specularWeight/(specularWeight + substrateWeight);
        diffusePdf *= (1.0f - specularProbability);
        glossyPdf *= specularProbability;
    }
    return glossyPdf + diffusePdf;
}
TG_DEV V3 cosineHemisphereSample(Sampler &s)
{
    float xix = s.next1D(), xiy = s.next1D();
    float phi = xix*TWO_PI; // SampleWarp::cosineHemisphere (sampling/SampleWarp.hpp:42-52)
    float r = __builtin_sqrtf(xiy);
    float sp, cp;
    tgpu_sincosf(phi, &sp, &cp);
    return V3(cp*r, sp*r, __builtin_sqrtf(fmax2(1.0f - xiy, 0.0f)));
}
TG_DEV bool plasticSample(const DScene &sc, const tgpu_bsdf &d, Event &e, Sampler &s)
{
    if (e.wi.z <= 0.0f) return false;
    const bool rough = d.type == TGPU_BSDF_ROUGH_PLASTIC;
    bool sampleR = (e.requestedLobe & (rough ? GlossyReflectionLobe : SpecularReflectionLobe)) != 0;
    bool sampleT = (e.requestedLobe & DiffuseReflectionLobe) != 0;
    const V3 wi = e.wi;
    float eta = 1.0f/d.ior;
    float ct;
    float Fi = dielectricReflectance(eta, wi.z, ct);
    if (!rough) {
        float substrateWeight = d.avg_transmittance*(1.0f - Fi);
        float specularWeight = Fi;
        float specularProbability;
        if (sampleR && sampleT) specularProbability = specularWeight/(specularWeight + substrateWeight);
        else if (sampleR) specularProbability = 1.0f;
        else if (sampleT) specularProbability = 0.0f;
        else return false;
        if (sampleR && s.nextBoolean(specularProbability)) {
            e.wo = V3(-wi.x, -wi.y, wi.z);
            e.pdf = specularProbability;
            e.weight = V3(Fi/specularProbability);
            e.sampledLobe = SpecularReflectionLobe;
        } else {
            V3 wo = cosineHemisphereSample(s);
            float Fo = dielectricReflectance(eta, wo.z, ct);
            e.wo = wo;
            e.weight = plasticSubstrate(sc, d, Fi, Fo, eta, e.uvx, e.uvy);
            if (V3(d.scaled_sigma_a).max() > 0.0f) e.weight *= plasticAbsorption(d, e.wi.z, e.wo.z);
            e.pdf = (__builtin_fabsf(e.wo.z)*INV_PI)*(1.0f - specularProbability);
            e.weight /= 1.0f - specularProbability;
            e.sampledLobe = DiffuseReflectionLobe;
        }
        return true;
    }
    if (!sampleR && !sampleT) return false;
    float substrateWeight = d.substrate_weight*d.avg_transmittance*(1.0f - Fi);
    float specularWeight = Fi;
    float specularProbability = specularWeight/(specularWeight + substrateWeight);
    const float roughness = texEval(sc, d.roughness_tex, e.uvx, e.uvy).x;
    if (sampleR && (s.nextBoolean(specularProbability) || !sampleT)) {
        if (!rdSampleBase(e, s, true, false, roughness, d.ior, d.distribution)) return false;
        if (sampleT) {
            float Fo = dielectricReflectance(eta, e.wo.z, ct);
            V3 brdfSubstrate = plasticSubstrate(sc, d, Fi, Fo, eta, e.uvx, e.uvy)*INV_PI*e.wo.z;
            V3 brdfSpecular = e.weight*e.pdf;
            float pdfSubstrate = (__builtin_fabsf(e.wo.z)*INV_PI)*(1.0f - specularProbability);
            float pdfSpecular = e.pdf*specularProbability;
            e.weight = (brdfSpecular + brdfSubstrate)/(pdfSpecular + pdfSubstrate);
            e.pdf = pdfSpecular + pdfSubstrate;
        }
        return true;
    }
    V3 wo = cosineHemisphereSample(s);
    float Fo = dielectricReflectance(eta, wo.z, ct);
    e.wo = wo;
    e.weight = plasticSubstrate(sc, d, Fi, Fo, eta, e.uvx, e.uvy);
    if (V3(d.scaled_sigma_a).max() > 0.0f) e.weight *= plasticAbsorption(d, e.wi.z, e.wo.z);
    e.pdf = __builtin_fabsf(e.wo.z)*INV_PI;
    if (sampleR) {
        V3 brdfSubstrate = e.weight*e.pdf;
        float pdfSubstrate = e.pdf*(1.0f - specularProbability);
        V3 brdfSpecular = rdEvalBase(e.wi, e.wo, true, false, roughness, d.ior, d.distribution);
        float pdfSpecular = rdPdfBase(e.wi, e.wo, true, false, roughness, d.ior, d.distribution);
        pdfSpecular *= specularProbability;
        e.weight = (brdfSpecular + brdfSubstrate)/(pdfSpecular + pdfSubstrate);
        e.pdf = pdfSpecular + pdfSubstrate;
    }
    e.sampledLobe = DiffuseReflectionLobe;
    return true;
}

// ---- the specular BSDFs: MirrorBsdf.cpp:29-65, ConductorBsdf.cpp:53-87, DielectricBsdf.cpp:50-190 ----------------
TG_DEV bool checkRefractionConstraint(const V3 &wi, const V3 &wo, float eta, float cosThetaT) // bsdfs/Bsdf.hpp:50-54
{
    float dotP = -wi.x*wo.x*eta - wi.y*wo.y*eta - __builtin_copysignf(cosThetaT, wi.z)*wo.z;
    return __builtin_fabsf(dotP - 1.0f) < 1e-3f;
}
TG_DEV V3 specularEval(const DScene &sc, const tgpu_bsdf &d, const V3 &wi, const V3 &wo, uint32_t requested, float uvx, float uvy)
{
    if (d.type == TGPU_BSDF_DIELECTRIC) {
        bool evalR = (requested & SpecularReflectionLobe) != 0;
        bool evalT = (requested & SpecularTransmissionLobe) != 0 && d.enable_refraction;
        float eta = wi.z < 0.0f ? d.ior : 1.0f/d.ior;
        float cosThetaT = 0.0f;
        float F = dielectricReflectance(eta, __builtin_fabsf(wi.z), cosThetaT);
        V3 f(0.0f);
        if (wi.z*wo.z >= 0.0f) {
            if (evalR && checkReflectionConstraint(wi, wo)) f = F*texEval(sc, d.albedo_tex, uvx, uvy);
        } else {
            if (evalT && checkRefractionConstraint(wi, wo, eta, cosThetaT)) f = (1.0f - F)*texEval(sc, d.albedo_tex, uvx, uvy);
        }
        f *= dielectricEtaSq(d.ior, wi, wo);
        return f;
    }
    if (!((requested & SpecularReflectionLobe) && checkReflectionConstraint(wi, wo))) return V3(0.0f);
    if (d.type == TGPU_BSDF_MIRROR) return texEval(sc, d.albedo_tex, uvx, uvy);
    return texEval(sc, d.albedo_tex, uvx, uvy)*conductorReflectance3(d.eta, d.k, wi.z);
}
TG_DEV float specularPdf(const tgpu_bsdf &d, const V3 &wi, const V3 &wo, uint32_t requested)
{
    if (d.type == TGPU_BSDF_DIELECTRIC) {
        bool sampleR = (requested & SpecularReflectionLobe) != 0;
        bool sampleT = (requested & SpecularTransmissionLobe) != 0 && d.enable_refraction;
        float eta = wi.z < 0.0f ? d.ior : 1.0f/d.ior;
        float cosThetaT = 0.0f;
        float F = dielectricReflectance(eta, __builtin_fabsf(wi.z), cosThetaT);
        if (wi.z*wo.z >= 0.0f) {
            if (sampleR && checkReflectionConstraint(wi, wo)) return sampleT ? F : 1.0f;
            return 0.0f;
        }
        if (sampleT && checkRefractionConstraint(wi, wo, eta, cosThetaT)) return sampleR ? 1.0f - F : 1.0f;
        return 0.0f;
    }
    return ((requested & SpecularReflectionLobe) && checkReflectionConstraint(wi, wo)) ? 1.0f : 0.0f;
}
TG_DEV bool specularSample(const DScene &sc, const tgpu_bsdf &d, Event &e, Sampler &s)
{
    if (d.type == TGPU_BSDF_DIELECTRIC) {
        bool sampleR = (e.requestedLobe & SpecularReflectionLobe) != 0;
        bool sampleT = (e.requestedLobe & SpecularTransmissionLobe) != 0 && d.enable_refraction;
        float eta = e.wi.z < 0.0f ? d.ior : 1.0f/d.ior;
        float cosThetaT = 0.0f;
        float F = dielectricReflectance(eta, __builtin_fabsf(e.wi.z), cosThetaT);
        float reflectionProbability;
        if (sampleR && sampleT) reflectionProbability = F;
        else if (sampleR) reflectionProbability = 1.0f;
        else if (sampleT) reflectionProbability = 0.0f;
        else return false;
        if (s.nextBoolean(reflectionProbability)) {
            e.wo = V3(-e.wi.x, -e.wi.y, e.wi.z);
            e.pdf = reflectionProbability;
            e.sampledLobe = SpecularReflectionLobe;
            e.weight = sampleT ? V3(1.0f) : V3(F);
        } else {
            if (F == 1.0f) return false;
            e.wo = V3(-e.wi.x*eta, -e.wi.y*eta, -__builtin_copysignf(cosThetaT, e.wi.z));
            e.pdf = 1.0f - reflectionProbability;
            e.sampledLobe = SpecularTransmissionLobe;
            e.weight = sampleR ? V3(1.0f) : V3(1.0f - F);
        }
        e.weight *= texEval(sc, d.albedo_tex, e.uvx, e.uvy);
        e.weight *= dielectricEtaSq(d.ior, e.wi, e.wo);
        return true;
    }
    if (!(e.requestedLobe & SpecularReflectionLobe)) return false;
    e.wo = V3(-e.wi.x, -e.wi.y, e.wi.z);
    e.pdf = 1.0f;
    e.sampledLobe = SpecularReflectionLobe;
    e.weight = texEval(sc, d.albedo_tex, e.uvx, e.uvy);
    if (d.type == TGPU_BSDF_CONDUCTOR) e.weight = e.weight*conductorReflectance3(d.eta, d.k, e.wi.z);
    return true;
}


// ---- thin sheets and the forward ("pass straight through") lobe ---------------------------------------
TG_DEV float thinFilmReflectance(float eta, float cosThetaI, float &cosThetaT) // bsdfs/Fresnel.hpp:15-28
{
    float sinThetaTSq = eta*eta*(1.0f - cosThetaI*cosThetaI);
    if (sinThetaTSq > 1.0f) {
        cosThetaT = 0.0f;
        return 1.0f;
    }
    cosThetaT = __builtin_sqrtf(fmax2(1.0f - sinThetaTSq, 0.0f));
    float a = (eta*cosThetaI - cosThetaT)/(eta*cosThetaI + cosThetaT), b = (eta*cosThetaT - cosThetaI)/(eta*cosThetaT + cosThetaI);
    float Rs = a*a, Rp = b*b;
    return 1.0f - ((1.0f - Rs)/(1.0f + Rs) + (1.0f - Rp)/(1.0f + Rp))*0.5f;
}
// bsdf.eval(event.makeForwardEvent()) of handleSurface / generalizedShadowRay (TraceBase.cpp:522-535, :84-96): what fraction of
// the radiance passes straight through the surface.  ForwardBsdf: 1; TransparencyBsdf: 1 - alpha; ThinSheetBsdf: the film's
// transmittance at this angle; every other BSDF: 0.
template <class SC>
TG_DEV V3 forwardTransparency(const SC &sc, const tgpu_bsdf &d, float wiz, float uvx, float uvy)
{
    if (d.type == TGPU_BSDF_FORWARD) return V3(1.0f);
    if (d.type == TGPU_BSDF_TRANSPARENCY) return V3(1.0f - texEval(sc, d.aux_tex, uvx, uvy).x); // TransparencyBsdf.cpp:47-53
    if (d.type == TGPU_BSDF_THINSHEET) { // ThinSheetBsdf.cpp:83-103
        float thickness = texEval(sc, d.aux_tex, uvx, uvy).x;
        float cosThetaT;
        V3 transmittance(1.0f - thinFilmReflectance(1.0f/d.ior, __builtin_fabsf(wiz), cosThetaT));
        V3 sigmaA(d.sigma_a);
        if (!sigmaA.isZero() && cosThetaT > 0.0f) transmittance *= expV3(-sigmaA*(thickness*2.0f/cosThetaT));
        return transmittance;
    }
    return V3(0.0f);
}

// ---- the children of the nested BSDFs (Mixed, RoughCoat, Transparency), as real calls --------------------------------------------------
// The four nested types evaluate, sample and take the pdf of one or two single-layer reflectors each; with those inlined, one nested entry point is
// 120-270 KB of code and the whole-path tail kernel took 27 minutes to compile.  The children are therefore CALLED: three leaf functions
// (__noinline__, a few tens of KB each), the nested code around them small enough to inline anywhere -- the tail kernels included.
//
// Why the size matters beyond compile time (round 2 lost nested scenes in the tail kernel to this; root cause found in round 3 with rocgdb on the
// faulting wave, profiles/r03_nested_notes.txt): s_cbranch reaches +-128 KB.  In a larger function LLVM's branch relaxation expands far branches to
// s_getpc_b64 / s_add_u32 / s_addc_u32 / s_setpc_b64 through a scavenged SGPR pair, and in a LEAF device function (ROCm 7.2, gfx950) the pair it takes
// is s[30:31] -- the return address, which a leaf never saves.  The function then "returns" into the middle of itself with a stale EXEC mask, lanes
// that were never called wake up with garbage in their argument registers, and the first flat_load through them is the
// HSA_STATUS_ERROR_MEMORY_APERTURE_VIOLATION.  Kernels are immune (no return address), callers of other functions too (they spill s[30:31] in their
// prologue).  Rule kept here and checked by profiles/kernel_resources.sh: no leaf device function above 128 KB.
struct MaterialTables { // what the single-layer reflectors read of a scene: 24 bytes by value instead of a reference to the 600-byte kernel argument
    const tgpu_bsdf *bsdfs;
    const tgpu_texture *textures;
    const DBitmap *bitmaps;
};
template <class SC> TG_DEV MaterialTables materialTables(const SC &sc) { return MaterialTables{sc.bsdfs, sc.textures, sc.bitmaps}; }
static __device__ __noinline__ V3 childEval(MaterialTables mt, const tgpu_bsdf *d, V3 wi, V3 wo, uint32_t requested, float uvx, float uvy)
{
    return baseEval(mt, *d, wi, wo, requested, uvx, uvy);
}
static __device__ __noinline__ float childPdf(MaterialTables mt, const tgpu_bsdf *d, V3 wi, V3 wo, uint32_t requested, float uvx, float uvy)
{
    return basePdf(mt, *d, wi, wo, requested, uvx, uvy);
}
static __device__ __noinline__ bool childSample(MaterialTables mt, const tgpu_bsdf *d, Event *e, Sampler *s) { return baseSample(mt, *d, *e, *s); }

// RoughCoatBsdf (bsdfs/RoughCoatBsdf.cpp:47-307) over a base BSDF
template <class SC>
TG_DEV void roughCoatSubstrate(const SC &sc, const tgpu_bsdf &d, const tgpu_bsdf &sub, const V3 &wi, const V3 &wo, uint32_t requested, float uvx, float uvy,
                               float eta, float Fi, float cosThetaTi, float &pdf, V3 &brdf) // substrateEvalAndPdf :47-87
{
    float cosThetaTo;
    float Fo = dielectricReflectance(eta, wo.z, cosThetaTo);
    if (Fi == 1.0f || Fo == 1.0f) {
        pdf = 0.0f;
        brdf = V3(0.0f);
        return;
    }
    V3 wiSubstrate(wi.x*eta, wi.y*eta, __builtin_copysignf(cosThetaTi, wi.z));
    V3 woSubstrate(wo.x*eta, wo.y*eta, __builtin_copysignf(cosThetaTo, wo.z));
    pdf = childPdf(materialTables(sc), &sub, wiSubstrate, woSubstrate, requested, uvx, uvy);
    pdf *= eta*eta*__builtin_fabsf(wo.z/cosThetaTo);
    float compressionProjection = eta*eta*wo.z/cosThetaTo;
    V3 substrateF = childEval(materialTables(sc), &sub, wiSubstrate, woSubstrate, requested, uvx, uvy);
    V3 ss(d.scaled_sigma_a);
    if (ss.max() > 0.0f) substrateF *= expV3(ss*(-1.0f/cosThetaTo - 1.0f/cosThetaTi));
    brdf = compressionProjection*(1.0f - Fi)*(1.0f - Fo)*substrateF;
}
template <class SC>
TG_DEV V3 roughCoatEval(const SC &sc, const tgpu_bsdf &d, const V3 &wi, const V3 &wo, uint32_t requested, float uvx, float uvy)
{   // :159-198
    const tgpu_bsdf &sub = sc.bsdfs[d.substrate];
    bool sampleR = (requested & GlossyReflectionLobe) != 0, sampleT = (requested & baseLobes(sub.type)) != 0;
    if (!sampleT && !sampleR) return V3(0.0f);
    if (wi.z <= 0.0f || wo.z <= 0.0f) return V3(0.0f);
    V3 glossyR(0.0f);
    if (sampleR) glossyR = rdEvalBase(wi, wo, true, false, texEval(sc, d.roughness_tex, uvx, uvy).x, d.ior, d.distribution);
    V3 substrateR(0.0f);
    if (sampleT) {
        float eta = 1.0f/d.ior;
        float cosThetaTi, pdfUnused;
        float Fi = dielectricReflectance(eta, wi.z, cosThetaTi);
        roughCoatSubstrate(sc, d, sub, wi, wo, requested, uvx, uvy, eta, Fi, cosThetaTi, pdfUnused, substrateR);
    }
    return glossyR + substrateR;
}
template <class SC>
TG_DEV float roughCoatPdf(const SC &sc, const tgpu_bsdf &d, const V3 &wi, const V3 &wo, uint32_t requested, float uvx, float uvy)
{   // :256-298
    const tgpu_bsdf &sub = sc.bsdfs[d.substrate];
    bool sampleR = (requested & GlossyReflectionLobe) != 0, sampleT = (requested & baseLobes(sub.type)) != 0;
    if (!sampleT && !sampleR) return 0.0f;
    if (wi.z <= 0.0f || wo.z <= 0.0f) return 0.0f;
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
    if (sampleR) glossyPdf = rdPdfBase(wi, wo, true, false, texEval(sc, d.roughness_tex, uvx, uvy).x, d.ior, d.distribution);
    float substratePdf = 0.0f;
    if (sampleT && Fi < 1.0f && Fo < 1.0f) {
        V3 wiSubstrate(wi.x*eta, wi.y*eta, __builtin_copysignf(cosThetaTi, wi.z));
        V3 woSubstrate(wo.x*eta, wo.y*eta, __builtin_copysignf(cosThetaTo, wo.z));
        substratePdf = childPdf(materialTables(sc), &sub, wiSubstrate, woSubstrate, requested, uvx, uvy);
        substratePdf *= eta*eta*__builtin_fabsf(wo.z/cosThetaTo);
    }
    return glossyPdf*specularProbability + substratePdf*(1.0f - specularProbability);
}
template <class SC>
TG_DEV bool roughCoatSample(const SC &sc, const tgpu_bsdf &d, Event &e, Sampler &s)
{   // :89-157
    if (e.wi.z <= 0.0f) return false;
    const tgpu_bsdf &sub = sc.bsdfs[d.substrate];
    bool sampleR = (e.requestedLobe & GlossyReflectionLobe) != 0, sampleT = (e.requestedLobe & baseLobes(sub.type)) != 0;
    if (!sampleR && !sampleT) return false;
    const V3 wi = e.wi;
    float eta = 1.0f/d.ior;
    float cosThetaTi;
    float Fi = dielectricReflectance(eta, wi.z, cosThetaTi);
    float substrateWeight = d.avg_transmittance*(1.0f - Fi);
    float specularWeight = Fi;
    float specularProbability = specularWeight/(specularWeight + substrateWeight);
    const float roughness = texEval(sc, d.roughness_tex, e.uvx, e.uvy).x;
    if (sampleR && (s.nextBoolean(specularProbability) || !sampleT)) {
        if (!rdSampleBase(e, s, true, false, roughness, d.ior, d.distribution)) return false;
        if (sampleT) {
            V3 brdfSubstrate, brdfSpecular = e.weight*e.pdf;
            float pdfSubstrate, pdfSpecular = e.pdf*specularProbability;
            roughCoatSubstrate(sc, d, sub, wi, e.wo, e.requestedLobe, e.uvx, e.uvy, eta, Fi, cosThetaTi, pdfSubstrate, brdfSubstrate);
            pdfSubstrate *= 1.0f - specularProbability;
            e.weight = (brdfSpecular + brdfSubstrate)/(pdfSpecular + pdfSubstrate);
            e.pdf = pdfSpecular + pdfSubstrate;
        }
        return true;
    }
    V3 wiSubstrate(wi.x*eta, wi.y*eta, cosThetaTi);
    e.wi = wiSubstrate;
    bool success = childSample(materialTables(sc), &sub, &e, &s);
    e.wi = wi;
    if (!success) return false;
    float cosThetaTo;
    float Fo = dielectricReflectance(d.ior, e.wo.z, cosThetaTo);
    if (Fo == 1.0f) return false;
    float cosThetaSubstrate = e.wo.z;
    e.wo = V3(e.wo.x*d.ior, e.wo.y*d.ior, cosThetaTo);
    e.weight *= (1.0f - Fi)*(1.0f - Fo);
    V3 ss(d.scaled_sigma_a);
    if (ss.max() > 0.0f) e.weight *= expV3(ss*(-1.0f/cosThetaSubstrate - 1.0f/cosThetaTi));
    e.weight *= wi.z/wiSubstrate.z;
    e.pdf *= eta*eta*cosThetaTo/cosThetaSubstrate;
    if (sampleR) {
        V3 brdfSubstrate = e.weight*e.pdf;
        float pdfSubstrate = e.pdf*(1.0f - specularProbability);
        V3 brdfSpecular = rdEvalBase(e.wi, e.wo, true, false, roughness, d.ior, d.distribution);
        float pdfSpecular = rdPdfBase(e.wi, e.wo, true, false, roughness, d.ior, d.distribution);
        pdfSpecular *= specularProbability;
        e.weight = (brdfSpecular + brdfSubstrate)/(pdfSpecular + pdfSubstrate);
        e.pdf = pdfSpecular + pdfSubstrate;
    }
    return true;
}

// MixedBsdf (bsdfs/MixedBsdf.cpp:17-30,72-136) of two base BSDFs
template <class SC>
TG_DEV bool mixedRatio(const SC &sc, const tgpu_bsdf &d, uint32_t requested, float uvx, float uvy, float &ratio)
{
    bool sample0 = (requested & baseLobes(sc.bsdfs[d.substrate].type)) != 0, sample1 = (requested & baseLobes(sc.bsdfs[d.aux_bsdf].type)) != 0;
    if (sample0 && sample1) ratio = texEval(sc, d.aux_tex, uvx, uvy).x;
    else if (sample0) ratio = 1.0f;
    else if (sample1) ratio = 0.0f;
    else return false;
    return true;
}
template <class SC>
TG_DEV bool mixedSample(const SC &sc, const tgpu_bsdf &d, Event &e, Sampler &s)
{
    float ratio;
    if (!mixedRatio(sc, d, e.requestedLobe, e.uvx, e.uvy, ratio)) return false;
    const tgpu_bsdf &b0 = sc.bsdfs[d.substrate], &b1 = sc.bsdfs[d.aux_bsdf];
    if (s.nextBoolean(ratio)) {
        if (!childSample(materialTables(sc), &b0, &e, &s)) return false;
        float pdf0 = e.pdf*ratio;
        float pdf1 = childPdf(materialTables(sc), &b1, e.wi, e.wo, e.requestedLobe, e.uvx, e.uvy)*(1.0f - ratio);
        V3 f = e.weight*e.pdf*ratio + childEval(materialTables(sc), &b1, e.wi, e.wo, e.requestedLobe, e.uvx, e.uvy)*(1.0f - ratio);
        e.pdf = pdf0 + pdf1;
        e.weight = f/e.pdf;
    } else {
        if (!childSample(materialTables(sc), &b1, &e, &s)) return false;
        float pdf0 = childPdf(materialTables(sc), &b0, e.wi, e.wo, e.requestedLobe, e.uvx, e.uvy)*ratio;
        float pdf1 = e.pdf*(1.0f - ratio);
        V3 f = childEval(materialTables(sc), &b0, e.wi, e.wo, e.requestedLobe, e.uvx, e.uvy)*ratio + e.weight*e.pdf*(1.0f - ratio);
        e.pdf = pdf0 + pdf1;
        e.weight = f/e.pdf;
    }
    e.weight *= texEval(sc, d.albedo_tex, e.uvx, e.uvy);
    return true;
}

// ---- the nested BSDFs (mixtures, coats, cut-outs, thin sheets) over the children above
template <class SC>
TG_DEV V3 nestedEval(const SC &sc, const tgpu_bsdf &d, const V3 &wi, const V3 &wo, uint32_t requested, float uvx, float uvy)
{
    if (d.type == TGPU_BSDF_MIXED) { // MixedBsdf.cpp:104-108
        float ratio = texEval(sc, d.aux_tex, uvx, uvy).x;
        return texEval(sc, d.albedo_tex, uvx, uvy)*(childEval(materialTables(sc), &sc.bsdfs[d.substrate], wi, wo, requested, uvx, uvy)*ratio +
                                                      childEval(materialTables(sc), &sc.bsdfs[d.aux_bsdf], wi, wo, requested, uvx, uvy)*(1.0f - ratio));
    }
    if (d.type == TGPU_BSDF_ROUGH_COAT) return roughCoatEval(sc, d, wi, wo, requested, uvx, uvy);
    if (d.type == TGPU_BSDF_TRANSPARENCY || d.type == TGPU_BSDF_THINSHEET) { // TransparencyBsdf.cpp:47-53, ThinSheetBsdf.cpp:83-103
        if (requested == uint32_t(ForwardLobe)) { // BsdfLobes::isForward(): exactly the forward lobe
            V3 nwi = -wi;
            return (nwi.x == wo.x && nwi.y == wo.y && nwi.z == wo.z) ? forwardTransparency(sc, d, wi.z, uvx, uvy) : V3(0.0f);
        }
        if (d.type == TGPU_BSDF_THINSHEET) return V3(0.0f);
        return childEval(materialTables(sc), &sc.bsdfs[d.substrate], wi, wo, requested, uvx, uvy);
    }
    return V3(0.0f);
}
template <class SC>
TG_DEV float nestedPdf(const SC &sc, const tgpu_bsdf &d, const V3 &wi, const V3 &wo, uint32_t requested, float uvx, float uvy)
{
    if (d.type == TGPU_BSDF_MIXED) { // MixedBsdf.cpp:125-131
        float ratio;
        if (!mixedRatio(sc, d, requested, uvx, uvy, ratio)) return 0.0f;
        return childPdf(materialTables(sc), &sc.bsdfs[d.substrate], wi, wo, requested, uvx, uvy)*ratio +
               childPdf(materialTables(sc), &sc.bsdfs[d.aux_bsdf], wi, wo, requested, uvx, uvy)*(1.0f - ratio);
    }
    if (d.type == TGPU_BSDF_ROUGH_COAT) return roughCoatPdf(sc, d, wi, wo, requested, uvx, uvy);
    if (d.type == TGPU_BSDF_TRANSPARENCY) return childPdf(materialTables(sc), &sc.bsdfs[d.substrate], wi, wo, requested, uvx, uvy); // TransparencyBsdf.cpp:60-63
    if (d.type == TGPU_BSDF_THINSHEET) return ((requested & SpecularReflectionLobe) && checkReflectionConstraint(wi, wo)) ? 1.0f : 0.0f; // :111-118
    return 0.0f;
}
template <class SC>
TG_DEV bool nestedSample(const SC &sc, const tgpu_bsdf &d, Event &e, Sampler &s)
{
    if (d.type == TGPU_BSDF_MIXED) return mixedSample(sc, d, e, s);
    if (d.type == TGPU_BSDF_ROUGH_COAT) return roughCoatSample(sc, d, e, s);
    if (d.type == TGPU_BSDF_TRANSPARENCY) return childSample(materialTables(sc), &sc.bsdfs[d.substrate], &e, &s); // TransparencyBsdf.cpp:42-45
    if (d.type == TGPU_BSDF_THINSHEET) { // ThinSheetBsdf.cpp:50-81
        if (!(e.requestedLobe & SpecularReflectionLobe)) return false;
        e.wo = V3(-e.wi.x, -e.wi.y, e.wi.z);
        e.pdf = 1.0f;
        e.sampledLobe = SpecularReflectionLobe;
        V3 sigmaA(d.sigma_a);
        if (sigmaA.isZero()) {
            e.weight = V3(1.0f);
            return true;
        }
        float thickness = texEval(sc, d.aux_tex, e.uvx, e.uvy).x;
        float cosThetaT;
        e.weight = V3(thinFilmReflectance(1.0f/d.ior, __builtin_fabsf(e.wi.z), cosThetaT));
        V3 transmittance = V3(1.0f) - e.weight;
        if (cosThetaT > 0.0f) transmittance *= expV3(-sigmaA*(thickness*2.0f/cosThetaT));
        e.weight /= 1.0f - transmittance.avg();
        return true;
    }
    return false;
}
// KNOWN: the BSDF type when the caller knows it at compile time (the per-class shade kernels), -1 otherwise.  The nested types never have a class
// of their own, so a kernel that knows its type carries none of their code -- stated here rather than left to __builtin_assume on d.type, which
// stops propagating at the first real call (the children of the nested BSDFs are calls, see above).
// NESTED = false: the caller never brings a BSDF made of other BSDFs (types from TGPU_BSDF_MIXED on) -- their code, out-of-line calls into the
// single-layer reflectors, stays out of the kernel (k_volume: a kernel with call sites pays for them whether they are taken or not).
template <int KNOWN = -1, bool NESTED = true>
TG_DEV V3 bsdfEval(const DScene &sc, const tgpu_bsdf &d, const Event &e)
{
    if (NESTED && KNOWN < 0 && d.type >= TGPU_BSDF_MIXED) return nestedEval(sc, d, e.wi, e.wo, e.requestedLobe, e.uvx, e.uvy);
    if (d.type == TGPU_BSDF_ROUGH_DIELECTRIC) return dielectricEval(sc, d, e.wi, e.wo, e.requestedLobe, e.uvx, e.uvy);
    if (d.type == TGPU_BSDF_PLASTIC || d.type == TGPU_BSDF_ROUGH_PLASTIC) return plasticEval(sc, d, e.wi, e.wo, e.requestedLobe, e.uvx, e.uvy);
    if (d.type >= TGPU_BSDF_MIRROR && d.type <= TGPU_BSDF_CONDUCTOR) return specularEval(sc, d, e.wi, e.wo, e.requestedLobe, e.uvx, e.uvy);
    if (d.type != TGPU_BSDF_SMOOTH_COAT) return baseEval(sc, d, e.wi, e.wo, e.requestedLobe, e.uvx, e.uvy);
    // bsdfs/SmoothCoatBsdf.cpp:146-178
    if (e.wi.z <= 0.0f || e.wo.z <= 0.0f) return V3(0.0f);
    const tgpu_bsdf &sub = sc.bsdfs[d.substrate];
    bool evalR = (e.requestedLobe & SpecularReflectionLobe) != 0;
    bool evalT = (e.requestedLobe & baseLobes(sub.type)) != 0;
    const V3 &wi = e.wi, &wo = e.wo;
    float eta = 1.0f/d.ior;
    float cosThetaTi, cosThetaTo;
    float Fi = dielectricReflectance(eta, wi.z, cosThetaTi);
    float Fo = dielectricReflectance(eta, wo.z, cosThetaTo);
    if (evalR && checkReflectionConstraint(wi, wo)) {
        return V3(Fi);
    } else if (evalT) {
        V3 wiSubstrate(wi.x*eta, wi.y*eta, __builtin_copysignf(cosThetaTi, wi.z));
        V3 woSubstrate(wo.x*eta, wo.y*eta, __builtin_copysignf(cosThetaTo, wo.z));
        float laplacian = eta*eta*wo.z/cosThetaTo;
        V3 substrateF = baseEval(sc, sub, wiSubstrate, woSubstrate, e.requestedLobe, e.uvx, e.uvy);
        V3 ss(d.scaled_sigma_a);
        if (ss.max() > 0.0f) substrateF *= expV3(ss*(-1.0f/cosThetaTo - 1.0f/cosThetaTi));
        return laplacian*(1.0f - Fi)*(1.0f - Fo)*substrateF;
    }
    return V3(0.0f);
}

template <int KNOWN = -1, bool NESTED = true>
TG_DEV float bsdfPdf(const DScene &sc, const tgpu_bsdf &d, const Event &e)
{
    if (NESTED && KNOWN < 0 && d.type >= TGPU_BSDF_MIXED) return nestedPdf(sc, d, e.wi, e.wo, e.requestedLobe, e.uvx, e.uvy);
    if (d.type == TGPU_BSDF_ROUGH_DIELECTRIC) return dielectricPdf(sc, d, e.wi, e.wo, e.requestedLobe, e.uvx, e.uvy);
    if (d.type == TGPU_BSDF_PLASTIC || d.type == TGPU_BSDF_ROUGH_PLASTIC) return plasticPdf(sc, d, e.wi, e.wo, e.requestedLobe, e.uvx, e.uvy);
    if (d.type >= TGPU_BSDF_MIRROR && d.type <= TGPU_BSDF_CONDUCTOR) return specularPdf(d, e.wi, e.wo, e.requestedLobe);
    if (d.type != TGPU_BSDF_SMOOTH_COAT) return basePdf(sc, d, e.wi, e.wo, e.requestedLobe, e.uvx, e.uvy);
    // bsdfs/SmoothCoatBsdf.cpp:180-215
    if (e.wi.z <= 0.0f || e.wo.z <= 0.0f) return 0.0f;
    const tgpu_bsdf &sub = sc.bsdfs[d.substrate];
    bool sampleR = (e.requestedLobe & SpecularReflectionLobe) != 0;
    bool sampleT = (e.requestedLobe & baseLobes(sub.type)) != 0;
    const V3 &wi = e.wi, &wo = e.wo;
    float eta = 1.0f/d.ior;
    float cosThetaTi, cosThetaTo;
    float Fi = dielectricReflectance(eta, wi.z, cosThetaTi);
    dielectricReflectance(eta, wo.z, cosThetaTo);
    V3 wiSubstrate(wi.x*eta, wi.y*eta, __builtin_copysignf(cosThetaTi, wi.z));
    V3 woSubstrate(wo.x*eta, wo.y*eta, __builtin_copysignf(cosThetaTo, wo.z));
    if (sampleR && sampleT) {
        float substrateWeight = d.avg_transmittance*(1.0f - Fi);
        float specularWeight = Fi;
        float specularProbability = specularWeight/(specularWeight + substrateWeight);
        if (checkReflectionConstraint(wi, wo))
            return specularProbability;
        else
            return basePdf(sc, sub, wiSubstrate, woSubstrate, e.requestedLobe, e.uvx, e.uvy)*(1.0f - specularProbability)*eta*eta*__builtin_fabsf(wo.z/cosThetaTo);
    } else if (sampleT) {
        return basePdf(sc, sub, wiSubstrate, woSubstrate, e.requestedLobe, e.uvx, e.uvy)*eta*eta*__builtin_fabsf(wo.z/cosThetaTo);
    } else if (sampleR) {
        return checkReflectionConstraint(wi, wo) ? 1.0f : 0.0f;
    }
    return 0.0f;
}

template <int KNOWN = -1, bool NESTED = true>
TG_DEV bool bsdfSample(const DScene &sc, const tgpu_bsdf &d, Event &e, Sampler &s)
{
    if (NESTED && KNOWN < 0 && d.type >= TGPU_BSDF_MIXED) return nestedSample(sc, d, e, s);
    if (d.type == TGPU_BSDF_ROUGH_DIELECTRIC) return dielectricSample(sc, d, e, s);
    if (d.type == TGPU_BSDF_PLASTIC || d.type == TGPU_BSDF_ROUGH_PLASTIC) return plasticSample(sc, d, e, s);
    if (d.type >= TGPU_BSDF_MIRROR && d.type <= TGPU_BSDF_CONDUCTOR) return specularSample(sc, d, e, s);
    if (d.type != TGPU_BSDF_SMOOTH_COAT) return baseSample(sc, d, e, s);
    // bsdfs/SmoothCoatBsdf.cpp:41-100
    if (e.wi.z <= 0.0f) return false;
    const tgpu_bsdf &sub = sc.bsdfs[d.substrate];
    bool sampleR = (e.requestedLobe & SpecularReflectionLobe) != 0;
    bool sampleT = (e.requestedLobe & baseLobes(sub.type)) != 0;
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
    if (sampleR && s.nextBoolean(specularProbability)) {
        e.wo = V3(-wi.x, -wi.y, wi.z);
        e.pdf = specularProbability;
        e.weight = V3(Fi/specularProbability);
        e.sampledLobe = SpecularReflectionLobe;
    } else {
        e.wi = V3(wi.x*eta, wi.y*eta, cosThetaTi);
        bool success = baseSample(sc, sub, e, s);
        e.wi = wi;
        if (!success) return false;
        float cosThetaTo;
        float Fo = dielectricReflectance(d.ior, e.wo.z, cosThetaTo);
        if (Fo == 1.0f) return false;
        float cosThetaSubstrate = e.wo.z;
        e.wo = V3(e.wo.x*d.ior, e.wo.y*d.ior, cosThetaTo);
        e.weight *= (1.0f - Fi)*(1.0f - Fo);
        V3 ss(d.scaled_sigma_a);
        if (ss.max() > 0.0f) e.weight *= expV3(ss*(-1.0f/cosThetaSubstrate - 1.0f/cosThetaTi));
        e.weight /= 1.0f - specularProbability;
        e.pdf *= 1.0f - specularProbability;
        e.pdf *= eta*eta*cosThetaTo/cosThetaSubstrate;
    }
    return true;
}

} // namespace tg
