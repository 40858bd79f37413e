// C ABI of the MI355X path-tracing core (include/tungsten_amd.h): context, scene upload,
// wavefront pass driver, downloads.  Host code only orchestrates; all path work runs in the
// kernels of kernels.h.  No CPU fallback exists: every entry point fails loudly without a GPU.
#define TG_TU_MAIN
#include "kernels.h"
#include "bvh_build.h"
#include "bvh_gpu.h"
#include "instance_bvh.h"

#include <algorithm>
#include <atomic>
#include <chrono>
#include <condition_variable>
#include <deque>
#include <mutex>
#include <thread>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <cstdlib>
#include <string>
#include <vector>

using namespace tg;
static constexpr uint32_t TG_RING_ALL = TG_RING >= 32 ? 0xFFFFFFFFu : ((1u << (TG_RING & 31)) - 1u); // every slice of the result ring free

// instantiated in tu_*.hip
namespace tg {
TG_KERNELS_SHADE(extern, false)
TG_KERNELS_SHADE(extern, true)
TG_KERNELS_SHADOW(extern)
TG_KERNELS_INSTQ(extern)
TG_KERNELS_TAIL(extern, false, false, false)
TG_KERNELS_TAIL(extern, false, false, true)
TG_KERNELS_TAIL(extern, false, true, false)
TG_KERNELS_TAIL(extern, false, true, true)
TG_KERNELS_TAIL(extern, true, false, false)
TG_KERNELS_TAIL(extern, true, false, true)
TG_KERNELS_TAIL(extern, true, true, false)
TG_KERNELS_TAIL(extern, true, true, true)
TG_KERNELS_TAIL_LEAN(extern, false, false)
TG_KERNELS_TAIL_LEAN(extern, false, true)
TG_KERNELS_TAIL_LEAN(extern, true, false)
TG_KERNELS_TAIL_LEAN(extern, true, true)
TG_KERNELS_TAIL_COOP(extern, false, false)
TG_KERNELS_TAIL_COOP(extern, false, true)
TG_KERNELS_TAIL_COOP(extern, true, false)
TG_KERNELS_TAIL_COOP(extern, true, true)
}

namespace {

struct DevBuf {
    void *p = nullptr;
    size_t bytes = 0;
};

} // namespace

struct tgpu_ctx {
    int device = 0;
    hipStream_t stream = nullptr;
    std::string error;
    bool uploaded = false;
    std::atomic<bool> abortFlag{false};

    DScene sc{};
    DState st{};
    std::vector<void *> allocs;     // scene allocations
    std::vector<void *> stateAllocs;

    uint32_t nPixels = 0;
    uint64_t capacity = 0; // path slots in the pool
    uint32_t resultSpp = 16; // samples per pixel one result-ring slot holds
    uint32_t *hostCounters = nullptr; // pinned

    // job queue + worker thread: tgpu_render_pass enqueues, the worker drives the wavefront loop
    struct Job {
        uint32_t sppBegin, sppCount;
        unsigned long long start, total;
        uint32_t ringSlot;
        bool adaptive = false;
        std::vector<uint32_t> offsets, vtIndex; // adaptive jobs: per-pixel sample offsets, per-tile sample index
    };
    uint32_t classMask = 0;               // BSDF classes (shade kernels) the uploaded scene can hit
    bool generalGeometry = false;         // instances, spheres or disks: INST instantiation of the traversal kernels
    bool generalShading = false;          // media, spheres or disks (also as lights): general instantiation of the shade kernels
    bool nestedBsdfs = false;             // the scene has a BSDF made of other BSDFs (mixed, rough_coat, transparency, thinsheet): with media, k_volume_nested runs
    bool coopTail = true;                 // ... and, with plain geometry, in its four-lanes-per-path form (TGPU_TAIL_COOP=0: one lane per path, A/B)
    bool leanTail = true;                 // scenes with lean shading finish in the lean tail kernel (TGPU_TAIL_LEAN=0: the general one, A/B)
    bool trackRecords = false;            // settings.adaptive_sampling
    uint32_t varW = 0, varH = 0;
    std::vector<uint32_t> vtGlobal;       // local variance tile -> whole-image variance tile
    std::vector<uint32_t> vtOfPixelHost;  // shard pixel -> local variance tile
    uint32_t lastTileOfUpload = 0;
    std::vector<uint32_t> tileStart;      // first shard pixel of every 16x16 tile of the shard, + nPixels (adaptive passes split here)
    std::mutex mtx;
    std::condition_variable cvWork, cvIdle;
    std::deque<Job> pending, active;
    bool idle = true, shutdown = false;
    int lastRc = TGPU_OK;
    std::thread worker;
    unsigned long long sampleBase = 0, globalCursor = 0;
    uint32_t nCur = 0, freeCount = 0;
    uint32_t listCount[TG_LISTS] = {0};
    uint64_t tailMax = 65536; // switch to the tail kernel below this many live paths
    uint32_t tailSpread = 2;  // lanes per path in the tail kernel (power of two; TGPU_TAIL_SPREAD)
    uint64_t aliveCount = 0, hostExtendRays = 0;
    int curQ = 0;
    uint32_t ringFree = TG_RING_ALL;
    int numCUs = 256;
    int maxGrid = 2048;
    int extendQueue = 4;        // rays per lane of the extend kernel's queue form: 4 or 8; TGPU_EXTEND_Q=0: one ray per lane and claim (A/B)
    bool extendPersistent = true;  // the persistent-lane form of the queue kernel (k_extend_p); TGPU_EXTEND_P=0: k_extend_q (A/B)
    double uploadMs = 0.0, bvhBuildMs = 0.0; // of the last tgpu_upload_scene
    bool bvhOnDevice = false;
    uint32_t nodeletCount = 0;   // nodes of the top BVH4 levels staged in LDS by the queue kernels (TGPU_LDS_NODELETS)
    bool instExtendQueue = false; // TGPU_INST_EXTEND_Q=1: k_extend_iq + k_classify for scenes with instances / spheres / disks (A/B)
    int shadowQueue = 4;        // vertices (= 2 rays each) per lane of the shadow kernel's queue form: 4 or 8; TGPU_SHADOW_Q=0: k_shadow (A/B)
    size_t ldsBytes = 0;

    uint32_t profFlags = 0;
    bool traceSlots = false;            // TGPU_TRACE_SLOTS: dump the live slots after every iteration (debugging aid, small pools only)
    std::vector<uint32_t> tracePrev;
    uint64_t traceIter = 0;
    tgpu_stats stats{};
    hipEvent_t ev[2] = {nullptr, nullptr};
    std::vector<tgpu_quad> hostQuads; // for host-side checks only
};

namespace {

bool fail(tgpu_ctx *ctx, const char *fmt, ...)
{
    char buf[1024];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof(buf), fmt, ap);
    va_end(ap);
    if (ctx) ctx->error = buf;
    return false;
}

#define HIP_TRY(ctx, expr)                                                                              \
    do {                                                                                                \
        hipError_t _e = (expr);                                                                         \
        if (_e != hipSuccess) {                                                                         \
            fail(ctx, "%s failed: %s (%s:%d)", #expr, hipGetErrorString(_e), __FILE__, __LINE__);       \
            return TGPU_ERR_HIP;                                                                        \
        }                                                                                               \
    } while (0)

template <typename T>
int upload(tgpu_ctx *ctx, std::vector<void *> &pool, const T *src, size_t count, T **dst)
{
    *dst = nullptr;
    if (count == 0) count = 1; // keep pointers valid
    void *p = nullptr;
    HIP_TRY(ctx, hipMalloc(&p, count*sizeof(T)));
    pool.push_back(p);
    if (src) HIP_TRY(ctx, hipMemcpy(p, src, count*sizeof(T), hipMemcpyHostToDevice));
    else HIP_TRY(ctx, hipMemset(p, 0, count*sizeof(T)));
    *dst = static_cast<T *>(p);
    return TGPU_OK;
}

void freePool(std::vector<void *> &pool)
{
    for (void *p : pool) (void)hipFree(p);
    pool.clear();
}

int gridFor(const tgpu_ctx *ctx, uint64_t items)
{
    uint64_t blocks = (items + 255)/256;
    if (blocks < 1) blocks = 1;
    if (blocks > uint64_t(ctx->maxGrid)) blocks = uint64_t(ctx->maxGrid);
    return int(blocks);
}

struct Timer {
    tgpu_ctx *ctx;
    double *acc, *acc2; // acc2: a second total the same interval belongs to (the retire step inside "other")
    bool on;
    Timer(tgpu_ctx *c, double *a, double *b = nullptr) : ctx(c), acc(a), acc2(b), on((c->profFlags & 2u) != 0)
    {
        if (on) (void)hipEventRecord(ctx->ev[0], ctx->stream);
    }
    ~Timer()
    {
        if (on) {
            (void)hipEventRecord(ctx->ev[1], ctx->stream);
            (void)hipEventSynchronize(ctx->ev[1]);
            float ms = 0.0f;
            (void)hipEventElapsedTime(&ms, ctx->ev[0], ctx->ev[1]);
            *acc += ms;
            if (acc2) *acc2 += ms;
        }
    }
};

int validate(tgpu_ctx *ctx, const tgpu_scene_desc *s)
{
    if (!s || s->abi_version != TGPU_ABI_VERSION) { fail(ctx, "scene description has ABI version %u, expected %u", s ? s->abi_version : 0u, TGPU_ABI_VERSION); return TGPU_ERR_INVALID; }
    if (s->camera.res_x == 0 || s->camera.res_y == 0) { fail(ctx, "camera resolution is zero"); return TGPU_ERR_INVALID; }
    if (s->n_tris && (!s->tri_pos || !s->tri_normal || !s->tri_uv || !s->tri_bsdf || !s->tri_smooth)) { fail(ctx, "triangle arrays missing"); return TGPU_ERR_INVALID; }
    if ((s->n_tris || s->n_master_tris) && (!s->tri_pos || !s->tri_normal || !s->tri_uv || !s->tri_bsdf || !s->tri_smooth)) { fail(ctx, "triangle arrays missing"); return TGPU_ERR_INVALID; }
    if (uint64_t(s->n_tris) + s->n_master_tris >= (1u << 28)) { fail(ctx, "too many triangles (%u + %u)", s->n_tris, s->n_master_tris); return TGPU_ERR_UNSUPPORTED; }
    if (!s->sobol_matrices) { fail(ctx, "Sobol matrices missing"); return TGPU_ERR_INVALID; }
    if (s->n_lights > TG_MAX_LIGHTS) { fail(ctx, "more than %d samplable lights are not supported", TG_MAX_LIGHTS); return TGPU_ERR_UNSUPPORTED; }
    if (s->settings.max_bounces < 0 || s->settings.max_bounces > 0xFFFF) { fail(ctx, "max_bounces out of range"); return TGPU_ERR_INVALID; }
    auto texOk = [&](int t) { return t >= 0 && uint32_t(t) < s->n_textures; };
    for (uint32_t i = 0; i < s->n_textures; ++i) {
        const tgpu_texture &t = s->textures[i];
        if (t.type == TGPU_TEX_BITMAP && (t.bitmap < 0 || uint32_t(t.bitmap) >= s->n_bitmaps)) { fail(ctx, "texture %u references bitmap %d", i, t.bitmap); return TGPU_ERR_INVALID; }
        if (t.type < 0 || t.type > TGPU_TEX_BITMAP) { fail(ctx, "texture %u has unknown type", i); return TGPU_ERR_INVALID; }
    }
    for (uint32_t i = 0; i < s->n_bitmaps; ++i)
        if (s->bitmaps[i].w <= 0 || s->bitmaps[i].h <= 0 || !s->bitmaps[i].texels) { fail(ctx, "bitmap %u is empty", i); return TGPU_ERR_INVALID; }
    for (uint32_t i = 0; i < s->n_bsdfs; ++i) {
        const tgpu_bsdf &b = s->bsdfs[i];
        if (b.type < 0 || b.type >= TGPU_BSDF_NUM_TYPES) { fail(ctx, "bsdf %u has type %d outside the hot-path scope", i, b.type); return TGPU_ERR_UNSUPPORTED; }
        if (!texOk(b.albedo_tex)) { fail(ctx, "bsdf %u: bad albedo texture", i); return TGPU_ERR_INVALID; }
        if ((b.type == TGPU_BSDF_ROUGH_CONDUCTOR || b.type == TGPU_BSDF_OREN_NAYAR) && !texOk(b.roughness_tex)) { fail(ctx, "bsdf %u: bad roughness texture", i); return TGPU_ERR_INVALID; }
        if (b.type == TGPU_BSDF_ROUGH_CONDUCTOR && b.distribution == TGPU_DIST_PHONG) { fail(ctx, "bsdf %u: phong microfacet distribution is outside the hot-path scope", i); return TGPU_ERR_UNSUPPORTED; }
        // nested BSDFs (coats, mixtures, cut-outs): the children are the single-layer opaque reflectors
        auto childOk = [&](int c, const char *what) {
            if (c < 0 || uint32_t(c) >= s->n_bsdfs) { fail(ctx, "bsdf %u: bad %s", i, what); return TGPU_ERR_INVALID; }
            const int st = s->bsdfs[c].type;
            if (st != TGPU_BSDF_LAMBERT && st != TGPU_BSDF_ROUGH_CONDUCTOR && st != TGPU_BSDF_OREN_NAYAR && st != TGPU_BSDF_PHONG) {
                fail(ctx, "bsdf %u: %s must be lambert, rough_conductor, oren_nayar or phong", i, what);
                return TGPU_ERR_UNSUPPORTED;
            }
            return TGPU_OK;
        };
        if (b.type == TGPU_BSDF_SMOOTH_COAT || b.type == TGPU_BSDF_ROUGH_COAT || b.type == TGPU_BSDF_TRANSPARENCY || b.type == TGPU_BSDF_MIXED)
            if (int rc = childOk(b.substrate, b.type == TGPU_BSDF_MIXED ? "first mixture component" : (b.type == TGPU_BSDF_TRANSPARENCY ? "base" : "substrate"))) return rc;
        if (b.type == TGPU_BSDF_MIXED)
            if (int rc = childOk(b.aux_bsdf, "second mixture component")) return rc;
        if ((b.type == TGPU_BSDF_MIXED || b.type == TGPU_BSDF_TRANSPARENCY || b.type == TGPU_BSDF_THINSHEET) && !texOk(b.aux_tex)) {
            fail(ctx, "bsdf %u: bad ratio / alpha / thickness texture", i);
            return TGPU_ERR_INVALID;
        }
        if (b.type == TGPU_BSDF_ROUGH_COAT) {
            if (!texOk(b.roughness_tex)) { fail(ctx, "bsdf %u: bad roughness texture", i); return TGPU_ERR_INVALID; }
            if (b.distribution == TGPU_DIST_PHONG) { fail(ctx, "bsdf %u: phong microfacet distribution is outside the hot-path scope", i); return TGPU_ERR_UNSUPPORTED; }
        }
    }
    auto bsdfOk = [&](int b) { return b >= 0 && uint32_t(b) < s->n_bsdfs; };
    for (uint32_t i = 0; i < s->n_tris + s->n_master_tris; ++i)
        if (!bsdfOk(s->tri_bsdf[i])) { fail(ctx, "triangle %u references bsdf %d", i, s->tri_bsdf[i]); return TGPU_ERR_INVALID; }
    for (uint32_t i = 0; i < s->n_quads + s->n_master_quads; ++i) {
        if (!bsdfOk(s->quads[i].bsdf)) { fail(ctx, "quad %u references bsdf %d", i, s->quads[i].bsdf); return TGPU_ERR_INVALID; }
        if (s->quads[i].emission_tex >= 0 && !texOk(s->quads[i].emission_tex)) { fail(ctx, "quad %u: bad emission texture", i); return TGPU_ERR_INVALID; }
    }
    for (uint32_t i = 0; i < s->n_cubes + s->n_master_cubes; ++i) {
        if (!bsdfOk(s->cubes[i].bsdf)) { fail(ctx, "cube %u references bsdf %d", i, s->cubes[i].bsdf); return TGPU_ERR_INVALID; }
        if (s->cubes[i].emission_tex >= 0) { fail(ctx, "emissive cubes are outside the hot-path scope"); return TGPU_ERR_UNSUPPORTED; }
    }
    for (uint32_t i = 0; i < s->n_spheres; ++i) {
        const tgpu_sphere &sp = s->spheres[i];
        if (!bsdfOk(sp.bsdf) || (sp.emission_tex >= 0 && !texOk(sp.emission_tex)) || !(sp.radius > 0.0f)) { fail(ctx, "sphere %u: bad bsdf / emission / radius", i); return TGPU_ERR_INVALID; }
    }
    for (uint32_t i = 0; i < s->n_disks; ++i) {
        const tgpu_disk &dk = s->disks[i];
        if (!bsdfOk(dk.bsdf) || (dk.emission_tex >= 0 && !texOk(dk.emission_tex)) || !(dk.r > 0.0f)) { fail(ctx, "disk %u: bad bsdf / emission / radius", i); return TGPU_ERR_INVALID; }
    }
    for (uint32_t i = 0; i < s->n_lights; ++i) {
        const tgpu_light &l = s->lights[i];
        if (l.kind == TGPU_LIGHT_ENV) {
            if (!s->env.present) { fail(ctx, "light %u is the environment but env.present is 0", i); return TGPU_ERR_INVALID; }
        } else if (l.kind == TGPU_LIGHT_QUAD) {
            if (l.index < 0 || uint32_t(l.index) >= s->n_quads || s->quads[l.index].emission_tex < 0) { fail(ctx, "light %u: bad quad", i); return TGPU_ERR_INVALID; }
        } else if (l.kind == TGPU_LIGHT_SPHERE) {
            if (l.index < 0 || uint32_t(l.index) >= s->n_spheres || s->spheres[l.index].emission_tex < 0) { fail(ctx, "light %u: bad sphere", i); return TGPU_ERR_INVALID; }
        } else if (l.kind == TGPU_LIGHT_DISK) {
            if (l.index < 0 || uint32_t(l.index) >= s->n_disks || s->disks[l.index].emission_tex < 0) { fail(ctx, "light %u: bad disk", i); return TGPU_ERR_INVALID; }
        } else { fail(ctx, "light %u has unknown kind", i); return TGPU_ERR_INVALID; }
    }
    // participating media
    if (s->n_media > 254) { fail(ctx, "more than 254 media"); return TGPU_ERR_UNSUPPORTED; }
    if (s->n_media && !s->media) { fail(ctx, "media array missing"); return TGPU_ERR_INVALID; }
    auto mediumOk = [&](int m) { return m >= -1 && m < int(s->n_media); };
    if (!mediumOk(s->camera.medium)) { fail(ctx, "camera references medium %d", s->camera.medium); return TGPU_ERR_INVALID; }
    for (uint32_t i = 0; i < s->n_quads + s->n_master_quads; ++i)
        if (!mediumOk(s->quads[i].int_medium) || !mediumOk(s->quads[i].ext_medium)) { fail(ctx, "quad %u: bad medium", i); return TGPU_ERR_INVALID; }
    for (uint32_t i = 0; i < s->n_cubes + s->n_master_cubes; ++i)
        if (!mediumOk(s->cubes[i].int_medium) || !mediumOk(s->cubes[i].ext_medium)) { fail(ctx, "cube %u: bad medium", i); return TGPU_ERR_INVALID; }
    for (uint32_t i = 0; i < s->n_spheres; ++i)
        if (!mediumOk(s->spheres[i].int_medium) || !mediumOk(s->spheres[i].ext_medium)) { fail(ctx, "sphere %u: bad medium", i); return TGPU_ERR_INVALID; }
    for (uint32_t i = 0; i < s->n_disks; ++i)
        if (!mediumOk(s->disks[i].int_medium) || !mediumOk(s->disks[i].ext_medium)) { fail(ctx, "disk %u: bad medium", i); return TGPU_ERR_INVALID; }
    if (s->tri_media)
        for (uint32_t i = 0; i < s->n_tris + s->n_master_tris; ++i)
            if (!mediumOk(s->tri_media[size_t(i)*2]) || !mediumOk(s->tri_media[size_t(i)*2 + 1])) { fail(ctx, "triangle %u: bad medium", i); return TGPU_ERR_INVALID; }
    for (uint32_t i = 0; i < s->n_media; ++i) {
        const tgpu_medium &m = s->media[i];
        auto modelOk = [](int type, const float *p) { // one transmittance model: TransmittanceFactory.cpp:14-24 minus "interpolated"
            if (type < TGPU_TRANS_EXPONENTIAL || type >= TGPU_TRANS_INTERPOLATED) return false;
            if ((type == TGPU_TRANS_LINEAR || type == TGPU_TRANS_QUADRATIC) && !(p[0] > 0.0f)) return false;
            if (type == TGPU_TRANS_DOUBLE_EXPONENTIAL && !(p[0] > 0.0f && p[1] > 0.0f)) return false;
            if (type == TGPU_TRANS_PULSE && !(p[1] > p[0] && p[2] >= 1.0f && p[2] <= 4096.0f)) return false;
            if (type == TGPU_TRANS_ERLANG && !(p[0] > 0.0f)) return false;
            if (type == TGPU_TRANS_DAVIS && !(p[0] > 1.0f)) return false;
            if (type == TGPU_TRANS_DAVIS_WEINSTEIN && !(p[0] >= 0.5f && p[0] <= 1.0f && p[1] > 0.0f)) return false;
            return true;
        };
        bool ok = m.transmittance == TGPU_TRANS_INTERPOLATED ? (modelOk(m.tr_a_type, m.tr_a_param) && modelOk(m.tr_b_type, m.tr_b_param) && m.tr_ratio >= 0.0f && m.tr_ratio <= 1.0f)
                                                             : modelOk(m.transmittance, m.param);
        if (m.phase < 0 || m.phase > 1 || !(m.phase_g > -1.0f && m.phase_g < 1.0f)) ok = false;
        if (!ok) { fail(ctx, "medium %u: bad transmittance / phase model or parameters", i); return TGPU_ERR_INVALID; }
    }
    for (uint32_t i = 0; i < s->n_media; ++i)
        if (s->media[i].max_bounce < 0 || s->media[i].max_bounce >= (1 << 22)) { fail(ctx, "medium %u: max_bounces out of range", i); return TGPU_ERR_UNSUPPORTED; }
    for (uint32_t i = 0; i < s->n_bsdfs; ++i)
        if (s->bsdfs[i].type == TGPU_BSDF_ROUGH_DIELECTRIC) {
            if (!texOk(s->bsdfs[i].roughness_tex)) { fail(ctx, "bsdf %u: bad roughness texture", i); return TGPU_ERR_INVALID; }
            if (s->bsdfs[i].distribution == TGPU_DIST_PHONG) { fail(ctx, "bsdf %u: phong microfacet distribution is outside the hot-path scope", i); return TGPU_ERR_UNSUPPORTED; }
        }
    // "instances" primitives
    if (s->n_instance_sets > 255) { fail(ctx, "more than 255 'instances' primitives"); return TGPU_ERR_UNSUPPORTED; }
    if (s->n_instance_sets && !s->instance_sets) { fail(ctx, "instance_sets missing"); return TGPU_ERR_INVALID; }
    for (uint32_t si = 0; si < s->n_instance_sets; ++si) {
        const tgpu_instances &is = s->instance_sets[si];
        if (is.n_masters == 0 || is.n_masters > 256 || !is.masters) { fail(ctx, "instance set %u: needs 1..256 masters", si); return TGPU_ERR_INVALID; }
        if (is.n_instances && (!is.pos || !is.rot || !is.master_id)) { fail(ctx, "instance set %u: arrays missing", si); return TGPU_ERR_INVALID; }
        // > 2^20 instances would take the reference's multi-threaded binning path (BvhBuilder.cpp:40-61), whose bin
        // partition differs from the serial one restated in instance_bvh.h
        if (is.n_instances > (1u << 20)) { fail(ctx, "instance set %u: more than 2^20 instances", si); return TGPU_ERR_UNSUPPORTED; }
        for (uint32_t m = 0; m < is.n_masters; ++m) {
            const tgpu_master &mm = is.masters[m];
            bool ok = false;
            if (mm.kind == TGPU_MASTER_MESH) ok = mm.count > 0 && mm.first >= s->n_tris && uint64_t(mm.first) + mm.count <= uint64_t(s->n_tris) + s->n_master_tris;
            else if (mm.kind == TGPU_MASTER_QUAD) ok = mm.first >= s->n_quads && mm.first < s->n_quads + s->n_master_quads && s->quads[mm.first].emission_tex < 0;
            else if (mm.kind == TGPU_MASTER_CUBE) ok = mm.first >= s->n_cubes && mm.first < s->n_cubes + s->n_master_cubes;
            if (!ok) { fail(ctx, "instance set %u: master %u does not reference tail geometry of its kind", si, m); return TGPU_ERR_INVALID; }
        }
        for (uint32_t i = 0; i < is.n_instances; ++i)
            if (is.master_id[i] >= is.n_masters) { fail(ctx, "instance set %u: instance %u references master %u", si, i, unsigned(is.master_id[i])); return TGPU_ERR_INVALID; }
    }
    if (s->env.present) {
        if (!texOk(s->env.emission_tex)) { fail(ctx, "environment: bad emission texture"); return TGPU_ERR_INVALID; }
        const tgpu_texture &t = s->textures[s->env.emission_tex];
        if (t.type == TGPU_TEX_CHECKER) { fail(ctx, "checker environment emission is outside the hot-path scope"); return TGPU_ERR_UNSUPPORTED; }
        if (t.type == TGPU_TEX_BITMAP && s->env.samplable) {
            if (!s->env.pdf || !s->env.cdf || !s->env.marginal_pdf || !s->env.marginal_cdf || s->env.dist_w <= 0 || s->env.dist_h <= 0) { fail(ctx, "environment: sampling tables missing"); return TGPU_ERR_INVALID; }
        }
    }
    return TGPU_OK;
}

void resetDequeueHeads(tgpu_ctx *ctx)
{
    (void)hipMemsetAsync(ctx->st.counters, 0, CNT_TOTAL*TG_CNT_STRIDE*sizeof(uint32_t), ctx->stream);
}

} // namespace

extern "C" {

int tgpu_init(int device, tgpu_ctx **out)
{
    if (!out) return TGPU_ERR_INVALID;
    *out = nullptr;
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || n <= 0 || device < 0 || device >= n) return TGPU_ERR_NO_DEVICE;
    if (hipSetDevice(device) != hipSuccess) return TGPU_ERR_NO_DEVICE;
    tgpu_ctx *ctx = new tgpu_ctx();
    ctx->device = device;
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, device) == hipSuccess) ctx->numCUs = prop.multiProcessorCount;
    ctx->maxGrid = ctx->numCUs*8;
    if (hipStreamCreate(&ctx->stream) != hipSuccess || hipEventCreate(&ctx->ev[0]) != hipSuccess ||
        hipEventCreate(&ctx->ev[1]) != hipSuccess || hipHostMalloc(reinterpret_cast<void **>(&ctx->hostCounters), CNT_TOTAL*TG_CNT_STRIDE*sizeof(uint32_t)) != hipSuccess) {
        delete ctx;
        return TGPU_ERR_NO_DEVICE;
    }
    *out = ctx;
    return TGPU_OK;
}

static void workerMain(tgpu_ctx *ctx);
static int waitIdle(tgpu_ctx *ctx, bool consume = false)
{
    std::unique_lock<std::mutex> lk(ctx->mtx);
    ctx->cvIdle.wait(lk, [&] { return ctx->idle; });
    int rc = ctx->lastRc;
    if (consume) ctx->lastRc = TGPU_OK;
    return rc;
}

void tgpu_destroy(tgpu_ctx *ctx)
{
    if (!ctx) return;
    if (ctx->worker.joinable()) {
        ctx->abortFlag = true;
        {
            std::lock_guard<std::mutex> lk(ctx->mtx);
            ctx->shutdown = true;
        }
        ctx->cvWork.notify_all();
        ctx->worker.join();
    }
    (void)hipSetDevice(ctx->device);
    if (ctx->stream) (void)hipStreamSynchronize(ctx->stream);
    freePool(ctx->allocs);
    freePool(ctx->stateAllocs);
    if (ctx->hostCounters) (void)hipHostFree(ctx->hostCounters);
    if (ctx->ev[0]) (void)hipEventDestroy(ctx->ev[0]);
    if (ctx->ev[1]) (void)hipEventDestroy(ctx->ev[1]);
    if (ctx->stream) (void)hipStreamDestroy(ctx->stream);
    delete ctx;
}

const char *tgpu_last_error(const tgpu_ctx *ctx) { return ctx ? ctx->error.c_str() : "null context"; }

// TGPU_BVH_VALIDATE=1 (tests, bring-up of a builder): reads the scene's tree back and checks what every walk relies on -- each child slot's
// dequantised box contains every triangle below it (vertices, unpadded), every triangle is reachable exactly once, leaf payloads stay inside the
// record array.  Returns the number of violations (0 = sound); prints the first few.
static uint64_t validateTree(const float4 *dNodes, uint32_t nNodes, const float4 *dTris, uint32_t nTris, uint32_t rootChild)
{
    std::vector<float4> nodes(size_t(nNodes)*4), tris(size_t(nTris)*4);
    if (hipMemcpy(nodes.data(), dNodes, nodes.size()*sizeof(float4), hipMemcpyDeviceToHost) != hipSuccess) return 1;
    if (hipMemcpy(tris.data(), dTris, tris.size()*sizeof(float4), hipMemcpyDeviceToHost) != hipSuccess) return 1;
    auto bitsOf = [](float f) { uint32_t u; std::memcpy(&u, &f, 4); return u; };
    auto floatOf = [](uint32_t u) { float f; std::memcpy(&f, &u, 4); return f; };
    struct B { float lo[3], hi[3]; };
    std::vector<uint8_t> seen(nTris, 0);
    uint64_t bad = 0;
    auto complain = [&](const char *what, uint32_t a, uint32_t b) { if (bad++ < 8) fprintf(stderr, "[tgpu] BVH validation: %s (node/record %u, slot/triangle %u)\n", what, a, b); };
    // post-order over inner nodes with an explicit stack: (node, next child slot), child bounds accumulated per frame
    struct Frame { uint32_t node; int slot; B acc; B child[4]; bool has[4]; };
    auto emptyB = []() { B b; for (int k = 0; k < 3; ++k) { b.lo[k] = INFINITY; b.hi[k] = -INFINITY; } return b; };
    auto grow = [](B &a, const B &b) { for (int k = 0; k < 3; ++k) { a.lo[k] = std::min(a.lo[k], b.lo[k]); a.hi[k] = std::max(a.hi[k], b.hi[k]); } };
    auto leafBounds = [&](uint32_t ref, B &out) {
        const uint32_t payload = ref & 0x7FFFFFFFu, first = payload >> 3, count = (payload & 7u) + 1u;
        out = emptyB();
        for (uint32_t i = first; i < first + count; ++i) {
            if (i >= nTris) { complain("leaf payload outside the triangle records", first, i); return; }
            if (seen[i]++) complain("triangle record referenced twice", first, i);
            const float4 *t = &tris[size_t(i)*4];
            const float v0[3] = {t[0].x, t[0].y, t[0].z}, e1[3] = {t[0].w, t[1].x, t[1].y}, e2[3] = {t[1].z, t[1].w, t[2].x};
            for (int k = 0; k < 3; ++k) {
                const float p[3] = {v0[k], v0[k] - e1[k], v0[k] + e2[k]};
                for (float c : p) { out.lo[k] = std::min(out.lo[k], c); out.hi[k] = std::max(out.hi[k], c); }
            }
        }
    };
    if (rootChild & 0x80000000u) { B b; leafBounds(rootChild, b); }
    else {
        std::vector<Frame> stack;
        stack.push_back(Frame{rootChild, 0, emptyB(), {}, {false, false, false, false}});
        while (!stack.empty()) {
            Frame &f = stack.back();
            if (f.node >= nNodes) { complain("node index out of range", f.node, 0); stack.pop_back(); continue; }
            const float4 *q = &nodes[size_t(f.node)*4];
            if (f.slot < 4) {
                const float cr[4] = {q[3].x, q[3].y, q[3].z, q[3].w};
                const uint32_t ref = bitsOf(cr[f.slot]);
                const int slot = f.slot++;
                if (ref == 0xFFFFFFFFu) continue;
                if (ref & 0x80000000u) { leafBounds(ref, f.child[slot]); f.has[slot] = true; }
                else { f.has[slot] = true; f.child[slot] = emptyB(); stack.push_back(Frame{ref, 0, emptyB(), {}, {false, false, false, false}}); stack[stack.size() - 2].slot = slot + 1; }
                continue;
            }
            // all children done: compare each child's true bounds with its quantised box
            const uint32_t eb = bitsOf(q[0].w);
            const float scale[3] = {floatOf((eb & 0xFFu) << 23), floatOf(((eb >> 8) & 0xFFu) << 23), floatOf(((eb >> 16) & 0xFFu) << 23)};
            const float origin[3] = {q[0].x, q[0].y, q[0].z};
            const uint32_t LO[3] = {bitsOf(q[1].x), bitsOf(q[1].y), bitsOf(q[1].z)}, HI[3] = {bitsOf(q[1].w), bitsOf(q[2].x), bitsOf(q[2].y)};
            B whole = emptyB();
            for (int c = 0; c < 4; ++c) {
                if (!f.has[c]) continue;
                for (int k = 0; k < 3; ++k) {
                    const float lo = origin[k] + float((LO[k] >> (8*c)) & 0xFFu)*scale[k], hi = origin[k] + float((HI[k] >> (8*c)) & 0xFFu)*scale[k];
                    if (!(lo <= f.child[c].lo[k] && hi >= f.child[c].hi[k])) { complain("child box does not contain its triangles", f.node, uint32_t(c)); break; }
                }
                grow(whole, f.child[c]);
            }
            const B done = whole;
            stack.pop_back();
            if (!stack.empty()) { Frame &p = stack.back(); p.child[p.slot - 1] = done; }
        }
    }
    uint64_t unreachable = 0;
    for (uint32_t i = 0; i < nTris; ++i) if (!seen[i]) ++unreachable;
    if (unreachable) { fprintf(stderr, "[tgpu] BVH validation: %llu of %u triangle records unreachable\n", (unsigned long long)unreachable, nTris); bad += unreachable; }
    return bad;
}

int tgpu_upload_scene(tgpu_ctx *ctx, const tgpu_scene_desc *s, const tgpu_shard_desc *shard, uint64_t maxPaths)
{
    if (!ctx) return TGPU_ERR_INVALID;
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    int rc = validate(ctx, s);
    if (rc != TGPU_OK) return rc;
    if (!shard || shard->n_pixels == 0 || !shard->pixel_index || !shard->tile_seed) { fail(ctx, "empty shard"); return TGPU_ERR_INVALID; }
    const auto uploadStart = std::chrono::steady_clock::now();
    double bvhBuildMs = 0.0;
    const uint32_t nImagePixels = s->camera.res_x*s->camera.res_y;
    for (uint32_t i = 0; i < shard->n_pixels; ++i)
        if (shard->pixel_index[i] >= nImagePixels) { fail(ctx, "shard pixel %u out of range", i); return TGPU_ERR_INVALID; }

    (void)waitIdle(ctx);
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    freePool(ctx->allocs);
    freePool(ctx->stateAllocs);
    ctx->uploaded = false;
    ctx->sampleBase = ctx->globalCursor = 0;
    ctx->nCur = 0;
    ctx->aliveCount = 0;
    for (int l = 0; l < TG_LISTS; ++l) ctx->listCount[l] = 0;
    ctx->curQ = 0;
    ctx->ringFree = TG_RING_ALL;
    ctx->sc = DScene{};
    ctx->st = DState{};
    DScene &sc = ctx->sc;
    sc.cam = s->camera;
    sc.settings = s->settings;

    // ---- BVHs: the scene's triangles, then one per mesh master of the "instances" primitives ----
    const uint32_t nAllTris = s->n_tris + s->n_master_tris;
    std::vector<tgpu_bvh::Float4> allNodes, allTris;
    int maxDepth4 = 0;
    // appends one BVH over triangles [first, first + count) and returns its root child reference
    const float triPad = tgpu_bvh_pad(s); // include/tgpu_bounds.h
    sc.hitPad = TGPU_HIT_PAD_FRACTION*triPad;
    // TGPU_LDS_NODELETS=L: stage the top L levels of the scene's BVH4 (1 + 4 + 16 + ... nodes) in LDS in the queue kernels (A/B; default off:
    // no gain measured, profiles/r02_ab_notes.txt)
    const int nodeletLevels = getenv("TGPU_LDS_NODELETS") ? std::min(std::max(atoi(getenv("TGPU_LDS_NODELETS")), 0), 5) : 0;
    uint32_t nodeletCount = 0;
    // Large flat soups get their tree from the device builder (bvh_gpu.h): the host's binned SAH is one thread and O(n log n).
    // TGPU_GPU_BVH=1 / 0 forces / forbids it; TGPU_GPU_BVH_MIN sets the triangle count from which it is used (default 1 Mi).
    // Scenes with mesh masters keep the host path for everything (their BVHs share one node array).
    tg_lbvh::Result deviceBvh;
    bool useDeviceBvh = s->n_master_tris == 0 && s->n_tris >= 3;
    {
        uint32_t minTris = 1u << 20;
        if (const char *e = getenv("TGPU_GPU_BVH_MIN")) minTris = uint32_t(strtoul(e, nullptr, 10));
        if (const char *e = getenv("TGPU_GPU_BVH")) useDeviceBvh = useDeviceBvh && atoi(e) != 0;
        else useDeviceBvh = useDeviceBvh && s->n_tris >= minTris;
    }
    auto appendBvh = [&](uint32_t first, uint32_t count, const char *what) -> uint32_t {
        const auto buildStart = std::chrono::steady_clock::now();
        if (useDeviceBvh && first == 0) {
            uint32_t maxLeaf = 2;
            if (const char *e = getenv("TGPU_BVH_LEAF")) maxLeaf = uint32_t(std::min(std::max(atoi(e), 1), 8));
            if (const char *msg = tg_lbvh::build(s->tri_pos, count, triPad, maxLeaf, ctx->stream, deviceBvh)) {
                fail(ctx, "device BVH build failed: %s", msg);
                return 0xFFFFFFFEu;
            }
            ctx->allocs.push_back(deviceBvh.nodes);
            ctx->allocs.push_back(deviceBvh.tris);
            maxDepth4 = std::max(maxDepth4, deviceBvh.maxDepth4);
            bvhBuildMs += deviceBvh.buildMs;
            if (getenv("TGPU_DEBUG")) fprintf(stderr, "[tgpu] %s: %u tris, device LBVH -> bvh4 %u nodes depth %d in %.1f ms (+ %.1f ms copy)\n", what, count, deviceBvh.nodeCount,
                                              deviceBvh.maxDepth4, deviceBvh.buildMs, deviceBvh.copyMs);
            return deviceBvh.rootChild;
        }
        tgpu_bvh::Builder builder(s->tri_pos + size_t(first)*9, count, triPad);
        tgpu_bvh::Bvh bvh = builder.build();
        tgpu_bvh::Bvh4 bvh4 = tgpu_bvh::Collapser(bvh).run(); // the quantised BVH4 the kernels traverse (bvh_build.h)
        // TGPU_BVH_BFS=1: breadth-first node order -- the (up to four) children of a node are neighbours, so the second child a ray visits usually
        // shares a 128-byte line with the first; the depth-first default keeps a node next to its FIRST child.  Measured: no difference
        // (profiles/r02_ab_notes.txt), like TGPU_BVH_ALIGN (bvh_build.h)
        if (getenv("TGPU_BVH_BFS") && atoi(getenv("TGPU_BVH_BFS"))) (void)tgpu_bvh::reorderTopLevels(bvh4, 1 << 20);
        if (first == 0 && nodeletLevels > 0) nodeletCount = tgpu_bvh::reorderTopLevels(bvh4, nodeletLevels); // the scene's own BVH only
        if ((allTris.size()/4) & 1u) for (int k = 0; k < 4; ++k) allTris.push_back(tgpu_bvh::Float4{0.0f, 0.0f, 0.0f, 0.0f}); // every BVH's records start on a line
        const uint32_t nodeBase = uint32_t(allNodes.size()/4), triBase = uint32_t(allTris.size()/4);
        auto rebase = [&](uint32_t ref) -> uint32_t {
            if (ref == 0xFFFFFFFFu) return ref;                               // empty slot
            if (ref & 0x80000000u) {                                          // leaf: first << 3 | count - 1
                uint32_t payload = ref & 0x7FFFFFFFu;
                return 0x80000000u | ((((payload >> 3) + triBase) << 3) | (payload & 7u));
            }
            return ref + nodeBase;
        };
        for (size_t i = 0; i < bvh4.nodes.size(); i += 4) {
            tgpu_bvh::Float4 q3 = bvh4.nodes[i + 3];
            float *c[4] = {&q3.x, &q3.y, &q3.z, &q3.w};
            for (float *f : c) {
                uint32_t bits;
                std::memcpy(&bits, f, 4);
                bits = rebase(bits);
                std::memcpy(f, &bits, 4);
            }
            allNodes.push_back(bvh4.nodes[i]); allNodes.push_back(bvh4.nodes[i + 1]); allNodes.push_back(bvh4.nodes[i + 2]); allNodes.push_back(q3);
        }
        // 64-byte triangle records, the size of a node: v0 e1 e2 Ng (3 x float4) + the soup triangle id
        for (size_t i = 0; i < bvh.triId.size(); ++i) {
            allTris.push_back(bvh.tris[i*3]); allTris.push_back(bvh.tris[i*3 + 1]); allTris.push_back(bvh.tris[i*3 + 2]);
            tgpu_bvh::Float4 tail{0.0f, 0.0f, 0.0f, 0.0f};
            const uint32_t id = bvh.triId[i] + first;
            std::memcpy(&tail.x, &id, 4);
            const tgpu_bvh::Float4 &t2 = bvh.tris[i*3 + 2];
            tail.y = t2.y*t2.y + (t2.z*t2.z + t2.w*t2.w); // |Ng|^2 in edot's order: the grazing test of triTest (include/tgpu_bounds.h (3))
            allTris.push_back(tail);
        }
        maxDepth4 = std::max(maxDepth4, bvh4.maxDepth);
        bvhBuildMs += std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - buildStart).count();
        if (getenv("TGPU_DEBUG")) fprintf(stderr, "[tgpu] %s: %u tris, bvh2 %u nodes depth %d -> bvh4 %u nodes depth %d\n", what, count, bvh.nodeCount, bvh.maxDepth, bvh4.nodeCount, bvh4.maxDepth);
        return rebase(bvh4.rootChild);
    };
    sc.nTris = s->n_tris;
    sc.rootChild = appendBvh(0, s->n_tris, "scene");
    if (sc.rootChild == 0xFFFFFFFEu) return TGPU_ERR_HIP;
    // ---- "instances" (Instance::prepareForRender: the BVH2 over instance boxes is built in instance_bvh.h) ----
    auto bitsToFloat = [](uint32_t u) { float f; std::memcpy(&f, &u, 4); return f; };
    std::vector<DInstSet> instSets(s->n_instance_sets);
    int topDepth = 0; // entries of the top-level (node, tMin) stack: BinaryBvh::trace needs depth + 1
    for (uint32_t si = 0; si < s->n_instance_sets; ++si) {
        const tgpu_instances &is = s->instance_sets[si];
        std::vector<DMaster> masters(is.n_masters);
        for (uint32_t m = 0; m < is.n_masters; ++m) {
            const tgpu_master &mm = is.masters[m];
            masters[m].kind = mm.kind;
            masters[m].ref = mm.kind == TGPU_MASTER_MESH ? appendBvh(mm.first, mm.count, "instance master") : mm.first;
        }
        tgpu_inst::TreeBuilder tb;
        tgpu_inst::Tree tree = tb.build(tgpu_inst::TreeBuilder::instanceItems(is));
        if (tree.depth + 1 > TG_TOP_STACK) { fail(ctx, "instance BVH of depth %d exceeds the traversal stack (%d)", tree.depth, TG_TOP_STACK); return TGPU_ERR_UNSUPPORTED; }
        std::vector<float4> nodes(tree.nodes.size()*4), pos(is.n_instances), rot(is.n_instances);
        for (size_t i = 0; i < tree.nodes.size(); ++i) {
            const tgpu_inst::Node &n = tree.nodes[i];
            const bool leaf = n.right < 0;
            nodes[i*4 + 0] = make_float4(n.bx[0], n.bx[1], n.bx[2], n.bx[3]);
            nodes[i*4 + 1] = make_float4(n.by[0], n.by[1], n.by[2], n.by[3]);
            nodes[i*4 + 2] = make_float4(n.bz[0], n.bz[1], n.bz[2], n.bz[3]);
            nodes[i*4 + 3] = make_float4(bitsToFloat(uint32_t(n.left)), bitsToFloat(leaf ? 0xFFFFFFFFu : uint32_t(n.right)),
                                         bitsToFloat(n.first), bitsToFloat(n.count));
        }
        for (uint32_t i = 0; i < is.n_instances; ++i) {
            pos[i] = make_float4(is.pos[size_t(i)*3], is.pos[size_t(i)*3 + 1], is.pos[size_t(i)*3 + 2], bitsToFloat(uint32_t(is.master_id[i])));
            rot[i] = make_float4(is.rot[size_t(i)*4], is.rot[size_t(i)*4 + 1], is.rot[size_t(i)*4 + 2], is.rot[size_t(i)*4 + 3]);
        }
        float4 *dN, *dP, *dR;
        uint32_t *dO;
        DMaster *dM;
        if ((rc = upload(ctx, ctx->allocs, nodes.data(), nodes.size(), &dN))) return rc;
        if ((rc = upload(ctx, ctx->allocs, tree.order.data(), tree.order.size(), &dO))) return rc;
        if ((rc = upload(ctx, ctx->allocs, pos.data(), pos.size(), &dP))) return rc;
        if ((rc = upload(ctx, ctx->allocs, rot.data(), rot.size(), &dR))) return rc;
        if ((rc = upload(ctx, ctx->allocs, masters.data(), masters.size(), &dM))) return rc;
        std::vector<float4> entries(tree.order.size()*4);
        for (size_t i = 0; i < tree.order.size(); ++i) {
            const uint32_t id = tree.order[i];
            entries[i*4 + 0] = make_float4(pos[id].x, pos[id].y, pos[id].z, bitsToFloat(id));
            entries[i*4 + 1] = rot[id];
            entries[i*4 + 2] = make_float4(bitsToFloat(uint32_t(masters[is.master_id[id]].kind)), bitsToFloat(masters[is.master_id[id]].ref), 0.0f, 0.0f);
            entries[i*4 + 3] = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
        }
        float4 *dE;
        if ((rc = upload(ctx, ctx->allocs, entries.data(), entries.size(), &dE))) return rc;
        topDepth = std::max(topDepth, tree.depth + 1);
        DInstSet &d = instSets[si];
        d.entries = dE;
        d.nodes = dN; d.order = dO; d.pos = dP; d.rot = dR; d.masters = dM;
        for (int k = 0; k < 3; ++k) { d.boundsMin[k] = tree.bounds.lo[k]; d.boundsMax[k] = tree.bounds.hi[k]; }
        d.rootIsLeaf = tree.nodes[0].right < 0;
        if (getenv("TGPU_DEBUG")) fprintf(stderr, "[tgpu] instance set %u: %u instances, %zu bvh2 nodes, depth %d\n", si, is.n_instances, tree.nodes.size(), tree.depth);
    }
    DInstSet *dSets;
    if ((rc = upload(ctx, ctx->allocs, instSets.data(), instSets.size(), &dSets))) return rc;
    sc.instSets = dSets;
    sc.nInstSets = s->n_instance_sets;

    sc.stackEntries = 3*maxDepth4 + 2; // at most 3 pushes per level
    // the A/B knobs apply to THIS upload: each starts from its default again (they used to stick to the context, so one upload with
    // TGPU_TAIL_MAX=0 left every later render of that context without its tail kernel)
    ctx->extendQueue = 4; ctx->extendPersistent = true; ctx->shadowQueue = 4;
    ctx->leanTail = true; ctx->coopTail = true; ctx->tailSpread = 2; ctx->tailMax = 65536;
    if (const char *e = getenv("TGPU_EXTEND_Q")) ctx->extendQueue = atoi(e) >= 8 ? 8 : (atoi(e) ? 4 : 0);
    if (const char *e = getenv("TGPU_EXTEND_P")) ctx->extendPersistent = atoi(e) != 0;
    ctx->instExtendQueue = getenv("TGPU_INST_EXTEND_Q") && atoi(getenv("TGPU_INST_EXTEND_Q")) != 0;
    if (const char *e = getenv("TGPU_SHADOW_Q")) ctx->shadowQueue = atoi(e) >= 8 ? 8 : (atoi(e) ? 4 : 0);
    if (const char *e = getenv("TGPU_TAIL_LEAN")) ctx->leanTail = atoi(e) != 0;
    if (const char *e = getenv("TGPU_TAIL_COOP")) ctx->coopTail = atoi(e) != 0;
    if (const char *e = getenv("TGPU_TAIL_SPREAD")) { uint32_t v = uint32_t(atoi(e)); ctx->tailSpread = v >= 16 ? 16 : (v >= 8 ? 8 : (v >= 4 ? 4 : (v >= 2 ? 2 : 1))); }
    ctx->traceSlots = getenv("TGPU_TRACE_SLOTS") != nullptr;
    if (const char *e = getenv("TGPU_TAIL_MAX")) ctx->tailMax = strtoull(e, nullptr, 10); // 0 disables the tail kernel (A/B)
    float4 *dNodes, *dTris;
    if (deviceBvh.nodes) { // the scene's tree was built in place (no mesh masters: nothing else shares the arrays)
        dNodes = deviceBvh.nodes;
        dTris = deviceBvh.tris;
    } else {
        if ((rc = upload(ctx, ctx->allocs, reinterpret_cast<const float4 *>(allNodes.data()), allNodes.size(), &dNodes))) return rc;
        if ((rc = upload(ctx, ctx->allocs, reinterpret_cast<const float4 *>(allTris.data()), allTris.size(), &dTris))) return rc;
    }
    sc.nodes = dNodes;
    sc.tris = dTris;
    if (getenv("TGPU_BVH_VALIDATE") && atoi(getenv("TGPU_BVH_VALIDATE")) && deviceBvh.nodes) {
        HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
        if (uint64_t bad = validateTree(dNodes, deviceBvh.nodeCount, dTris, s->n_tris, sc.rootChild)) { fail(ctx, "BVH validation found %llu violation(s)", (unsigned long long)bad); return TGPU_ERR_HIP; }
    }
    // 20 levels of LDS stack = 20 KB per block: 8 blocks (32 waves) fit a CU, so registers, not LDS, set the occupancy
    // (profiles/r01_ab_notes.txt); deeper levels spill to global memory
    sc.stackLds = 20;
    if (const char *e = getenv("TGPU_STACK_LDS")) sc.stackLds = std::max(1, atoi(e));
    // scenes with "instances": the top-level stack lives in LDS too (8 bytes per entry and thread), the mesh levels shrink to 12
    sc.topStackLds = topDepth;
    if (topDepth && !getenv("TGPU_STACK_LDS")) sc.stackLds = 12;
    sc.stackLds = std::min(sc.stackLds, sc.stackEntries);
    sc.spillStride = uint32_t(ctx->maxGrid)*256u;
    sc.stackSpill = nullptr;
    if (sc.stackEntries > sc.stackLds) {
        uint32_t *dSpill;
        if ((rc = upload<uint32_t>(ctx, ctx->allocs, nullptr, size_t(sc.stackEntries - sc.stackLds)*sc.spillStride, &dSpill))) return rc;
        sc.stackSpill = dSpill;
    }
    ctx->ldsBytes = size_t(sc.stackLds)*256*sizeof(uint32_t) + size_t(sc.topStackLds)*256*2*sizeof(uint32_t);
    sc.ldsNodes = 0; // the scene copy of the queue kernels' launches carries the real count (driveIteration)
    sc.ldsNodeWord = uint32_t(ctx->ldsBytes/sizeof(uint32_t));
    ctx->nodeletCount = nodeletCount;

    // ---- shading records ----
    std::vector<tgpu_bvh::Float4> shade(size_t(nAllTris)*5);
    std::vector<uint8_t> triClass(nAllTris);
    for (uint32_t i = 0; i < nAllTris; ++i) {
        const float *n = s->tri_normal + size_t(i)*9, *uv = s->tri_uv + size_t(i)*6, *p = s->tri_pos + size_t(i)*9;
        uint32_t bits = uint32_t(s->tri_bsdf[i]) | (s->tri_smooth[i] ? 0x80000000u : 0u);
        float fb;
        std::memcpy(&fb, &bits, 4);
        // unnormalizedGeometricNormalAt: (p1 - p0) x (p2 - p0)  (TriangleMesh.cpp:81-88)
        float a[3] = {p[3] - p[0], p[4] - p[1], p[5] - p[2]}, b[3] = {p[6] - p[0], p[7] - p[1], p[8] - p[2]};
        float ng[3] = {a[1]*b[2] - a[2]*b[1], a[2]*b[0] - a[0]*b[2], a[0]*b[1] - a[1]*b[0]};
        tgpu_bvh::Float4 *q = &shade[size_t(i)*5];
        q[0] = {n[0], n[1], n[2], n[3]};
        q[1] = {n[4], n[5], n[6], n[7]};
        q[2] = {n[8], uv[0], uv[1], uv[2]};
        q[3] = {uv[3], uv[4], uv[5], fb};
        uint32_t mediaBits = 0; // (int_medium + 1) << 16 | (ext_medium + 1)
        if (s->tri_media) mediaBits = (uint32_t(int(s->tri_media[size_t(i)*2]) + 1) << 16) | uint32_t(int(s->tri_media[size_t(i)*2 + 1]) + 1);
        float fm;
        std::memcpy(&fm, &mediaBits, 4);
        q[4] = {ng[0], ng[1], ng[2], fm};
        triClass[i] = uint8_t(s->bsdfs[s->tri_bsdf[i]].type);
    }
    float4 *dShade;
    uint8_t *dClass;
    if ((rc = upload(ctx, ctx->allocs, reinterpret_cast<const float4 *>(shade.data()), shade.size(), &dShade))) return rc;
    if ((rc = upload(ctx, ctx->allocs, triClass.data(), triClass.size(), &dClass))) return rc;
    sc.triShade = dShade;
    sc.triClass = dClass;

    // ---- analytic primitives, materials, textures ----
    tgpu_quad *dQuads; tgpu_cube *dCubes; tgpu_bsdf *dBsdfs; tgpu_texture *dTex; tgpu_light *dLights;
    if ((rc = upload(ctx, ctx->allocs, s->quads, s->n_quads + s->n_master_quads, &dQuads))) return rc; // masters at the tail
    if ((rc = upload(ctx, ctx->allocs, s->cubes, s->n_cubes + s->n_master_cubes, &dCubes))) return rc;
    if ((rc = upload(ctx, ctx->allocs, s->bsdfs, s->n_bsdfs, &dBsdfs))) return rc;
    if ((rc = upload(ctx, ctx->allocs, s->textures, s->n_textures, &dTex))) return rc;
    if ((rc = upload(ctx, ctx->allocs, s->lights, s->n_lights, &dLights))) return rc;
    sc.quads = dQuads; sc.nQuads = s->n_quads;
    sc.cubes = dCubes; sc.nCubes = s->n_cubes;
    {
        tgpu_sphere *dSpheres; tgpu_disk *dDisks;
        if ((rc = upload(ctx, ctx->allocs, s->spheres, s->n_spheres, &dSpheres))) return rc;
        if ((rc = upload(ctx, ctx->allocs, s->disks, s->n_disks, &dDisks))) return rc;
        sc.spheres = dSpheres; sc.nSpheres = s->n_spheres;
        sc.disks = dDisks; sc.nDisks = s->n_disks;
    }
    sc.bsdfs = dBsdfs;
    sc.hasForward = 0;
    ctx->generalGeometry = s->n_instance_sets || s->n_spheres || s->n_disks;
    ctx->generalShading = s->n_media != 0 || s->n_spheres || s->n_disks;
    for (uint32_t i = 0; i < s->n_lights; ++i) if (s->lights[i].kind == TGPU_LIGHT_SPHERE || s->lights[i].kind == TGPU_LIGHT_DISK) ctx->generalShading = true;
    ctx->classMask = 0;
    ctx->nestedBsdfs = false;
    for (uint32_t i = 0; i < s->n_bsdfs; ++i) if (s->bsdfs[i].type >= TGPU_BSDF_MIXED) ctx->nestedBsdfs = true;
    for (uint32_t i = 0; i < s->n_bsdfs; ++i) ctx->classMask |= 1u << std::min<uint32_t>(uint32_t(s->bsdfs[i].type), uint32_t(TG_CLASS_OTHER));
    for (uint32_t i = 0; i < s->n_bsdfs; ++i) { // every BSDF with a ForwardLobe lets shadow rays through
        const int t = s->bsdfs[i].type;
        if (t == TGPU_BSDF_FORWARD || t == TGPU_BSDF_TRANSPARENCY || t == TGPU_BSDF_THINSHEET) sc.hasForward = 1;
    }
    if (sc.hasForward) ctx->generalShading = true; // the shadow walk needs the NEE factors the general shade kernels store
    for (int t = TGPU_OUTPUT_DEPTH; t < TGPU_OUTPUT_COUNT; ++t) // only the general shade kernels record the auxiliary output values
        if (s->output_buffers[t].enabled) ctx->generalShading = true;
    sc.textures = dTex;
    sc.lights = dLights; sc.nLights = s->n_lights;
    {   // media + the table of fmath::exp_ps (fmath.hpp:106-126: a = n/logf(2), b = logf(2)/n, tbl[i] = mantissa(pow(2.0f, i/n)))
        tgpu_medium *dMedia;
        if ((rc = upload(ctx, ctx->allocs, s->media, s->n_media, &dMedia))) return rc;
        sc.media = dMedia; sc.nMedia = s->n_media;
        std::vector<uint32_t> tbl(1024);
        const float log2f_ = ::logf(2.0f);
        sc.expA = 1024/log2f_;
        sc.expB = log2f_/1024;
        for (int i = 0; i < 1024; ++i) {
            float y = std::pow(2.0f, float(i)/1024);
            uint32_t bits;
            std::memcpy(&bits, &y, 4);
            tbl[size_t(i)] = bits & 0x7FFFFFu;
        }
        uint32_t *dTbl;
        if ((rc = upload(ctx, ctx->allocs, tbl.data(), tbl.size(), &dTbl))) return rc;
        sc.expTable = dTbl;
    }
    std::vector<DBitmap> bitmaps(s->n_bitmaps);
    for (uint32_t i = 0; i < s->n_bitmaps; ++i) {
        const tgpu_bitmap &b = s->bitmaps[i];
        // shared-exponent words when every texel survives the round trip bit for bit (device.h, decodeRgbe); TGPU_TEX_FLOAT=1 keeps floats (A/B)
        const size_t nTexels = size_t(b.w)*b.h;
        std::vector<uint32_t> rgbe(nTexels);
        bool exact = getenv("TGPU_TEX_FLOAT") == nullptr;
        for (size_t t = 0; t < nTexels && exact; ++t) {
            const float *c = b.texels + t*3;
            const float m = std::max(c[0], std::max(c[1], c[2]));
            uint32_t w = 0;
            if (!(c[0] >= 0.0f && c[1] >= 0.0f && c[2] >= 0.0f) || !std::isfinite(m)) { exact = false; break; }
            if (m > 0.0f) {
                int k;
                (void)std::frexp(m, &k); // m = f*2^k, f in [0.5, 1): the largest byte lands in [128, 255]
                const int e = k + 128;
                if (e < 10 || e > 255) { exact = false; break; }
                uint32_t scaleBits = uint32_t(e - 9) << 23;
                float f1;
                std::memcpy(&f1, &scaleBits, 4);
                uint32_t bytes[3];
                for (int ch = 0; ch < 3; ++ch) {
                    const float q = c[ch]/f1; // exact: a power-of-two scale
                    bytes[ch] = uint32_t(q);
                    if (q != float(bytes[ch]) || bytes[ch] > 255u || float(bytes[ch])*f1 != c[ch]) exact = false;
                }
                w = bytes[0] | (bytes[1] << 8) | (bytes[2] << 16) | (uint32_t(e) << 24);
            }
            rgbe[t] = w;
        }
        float4 *dTexels = nullptr;
        uint32_t *dRgbe = nullptr;
        if (exact) {
            if ((rc = upload(ctx, ctx->allocs, rgbe.data(), rgbe.size(), &dRgbe))) return rc;
        } else {
            std::vector<float4> padded(nTexels);
            for (size_t t = 0; t < padded.size(); ++t) padded[t] = make_float4(b.texels[t*3], b.texels[t*3 + 1], b.texels[t*3 + 2], 0.0f);
            if ((rc = upload(ctx, ctx->allocs, padded.data(), padded.size(), &dTexels))) return rc;
        }
        if (getenv("TGPU_DEBUG")) fprintf(stderr, "[tgpu] bitmap %u: %dx%d as %s\n", i, b.w, b.h, exact ? "RGBE words" : "float4 texels");
        bitmaps[i] = DBitmap{b.w, b.h, b.clamp, b.linear, b.scale, dTexels, {b.average[0], b.average[1], b.average[2]}, dRgbe};
    }
    DBitmap *dBitmaps;
    if ((rc = upload(ctx, ctx->allocs, bitmaps.data(), bitmaps.size(), &dBitmaps))) return rc;
    sc.bitmaps = dBitmaps;

    // ---- environment ----
    sc.env = s->env;
    sc.env.pdf = sc.env.cdf = sc.env.marginal_pdf = sc.env.marginal_cdf = nullptr;
    sc.envConstant = 1;
    if (s->env.present) {
        sc.envConstant = s->textures[s->env.emission_tex].type == TGPU_TEX_CONSTANT;
        if (!sc.envConstant && s->env.samplable) {
            size_t w = size_t(s->env.dist_w), h = size_t(s->env.dist_h);
            float *a, *b, *c, *d;
            if ((rc = upload(ctx, ctx->allocs, s->env.pdf, w*h, &a))) return rc;
            if ((rc = upload(ctx, ctx->allocs, s->env.cdf, (w + 1)*h, &b))) return rc;
            if ((rc = upload(ctx, ctx->allocs, s->env.marginal_pdf, h, &c))) return rc;
            if ((rc = upload(ctx, ctx->allocs, s->env.marginal_cdf, h + 1, &d))) return rc;
            sc.env.pdf = a; sc.env.cdf = b; sc.env.marginal_pdf = c; sc.env.marginal_cdf = d;
            // guide records (device.h: EnvGuide, upperBoundGuided)
            auto guideOf = [](const float *cdf, const float *pdf, const float *pdfEnd, size_t n, size_t buckets, size_t g) { // cdf: n entries
                EnvGuide e{};
                const size_t i = size_t((std::upper_bound(cdf, cdf + n, float(g)/float(buckets)) - cdf) - 1);
                e.idx = uint32_t(i);
                e.a0 = cdf[i];
                e.a1 = i + 1 < n ? cdf[i + 1] : 0.0f;
                e.a2 = i + 2 < n ? cdf[i + 2] : 0.0f;
                e.p0 = pdf + i < pdfEnd ? pdf[i] : 0.0f; // what the scan would read at p[i]: the next row's first element past a row's end, as in the plain tables
                e.p1 = pdf + i + 1 < pdfEnd ? pdf[i + 1] : 0.0f;
                return e;
            };
            std::vector<EnvGuide> gm(h), gc(w*h);
            for (size_t g = 0; g < h; ++g) gm[g] = guideOf(s->env.marginal_cdf, s->env.marginal_pdf, s->env.marginal_pdf + h, h + 1, h, g);
            for (size_t r = 0; r < h; ++r)
                for (size_t g = 0; g < w; ++g) gc[r*w + g] = guideOf(s->env.cdf + r*(w + 1), s->env.pdf + r*w, s->env.pdf + w*h, w + 1, w, g);
            EnvGuide *dgm, *dgc;
            if ((rc = upload(ctx, ctx->allocs, gm.data(), gm.size(), &dgm))) return rc;
            if ((rc = upload(ctx, ctx->allocs, gc.data(), gc.size(), &dgc))) return rc;
            sc.envGuideMarginal = dgm;
            sc.envGuideConditional = dgc;
            // value*N is exact when N is a power of two: int(value*N) is then the bucket that holds value and no slack is needed
            sc.envGuideSlack = ((h & (h - 1)) ? 1 : 0) | ((w & (w - 1)) ? 2 : 0);
        }
    }
    uint32_t *dSobol;
    if ((rc = upload(ctx, ctx->allocs, s->sobol_matrices, size_t(1024)*32, &dSobol))) return rc;
    sc.sobol = dSobol;

    // ---- shard + framebuffer ----
    DState &st = ctx->st;
    ctx->nPixels = shard->n_pixels;
    std::vector<uint32_t> scramble(shard->n_pixels);
    auto h32 = [](uint32_t x) { x = ~x + (x << 15); x = x ^ (x >> 12); x = x + (x << 2); x = x ^ (x >> 4); x = x*2057; x = x ^ (x >> 16); return x; };
    for (uint32_t i = 0; i < shard->n_pixels; ++i) scramble[i] = shard->tile_seed[i] ^ h32(shard->pixel_index[i]); // SobolPathSampler.hpp:47-52
    uint32_t *dPix, *dScr, *dCnt;
    float *dMean;
    if ((rc = upload(ctx, ctx->stateAllocs, shard->pixel_index, shard->n_pixels, &dPix))) return rc;
    if ((rc = upload(ctx, ctx->stateAllocs, scramble.data(), scramble.size(), &dScr))) return rc;
    if ((rc = upload<float>(ctx, ctx->stateAllocs, nullptr, size_t(shard->n_pixels)*3, &dMean))) return rc;
    if ((rc = upload<uint32_t>(ctx, ctx->stateAllocs, nullptr, shard->n_pixels, &dCnt))) return rc;
    st.pixelIndex = dPix; st.scramble = dScr; st.fbMean = dMean; st.fbCount = dCnt;
    // renderer "output_buffers" (Camera::requestOutputBuffers, Camera.cpp:148-166): the colour buffer's optional parts and the
    // auxiliary buffers
    bool anyAux = false;
    for (int t = 0; t < TGPU_OUTPUT_COUNT; ++t) {
        const tgpu_output_buffer &ob = s->output_buffers[t];
        DOutBuf &d = st.out[t];
        d = DOutBuf{nullptr, nullptr, nullptr, nullptr};
        if (t != TGPU_OUTPUT_COLOR && !ob.enabled) continue;
        const size_t n = size_t(shard->n_pixels)*((t == TGPU_OUTPUT_DEPTH || t == TGPU_OUTPUT_VISIBILITY) ? 1 : 3);
        if (t == TGPU_OUTPUT_COLOR) {
            d.a = dMean;
            d.count = dCnt;
        } else {
            anyAux = true;
            if ((rc = upload<float>(ctx, ctx->stateAllocs, nullptr, n, &d.a))) return rc;
            if ((rc = upload<uint32_t>(ctx, ctx->stateAllocs, nullptr, shard->n_pixels, &d.count))) return rc;
        }
        if (ob.two_buffer_variance && (rc = upload<float>(ctx, ctx->stateAllocs, nullptr, n, &d.b))) return rc;
        if (ob.sample_variance && (rc = upload<float>(ctx, ctx->stateAllocs, nullptr, n, &d.var))) return rc;
    }

    // ---- variance tiles (4x4 pixel blocks, PathTraceIntegrator.hpp:27-29) of the shard ----
    ctx->trackRecords = s->settings.adaptive_sampling != 0;
    ctx->varW = (s->camera.res_x + 3)/4;
    ctx->varH = (s->camera.res_y + 3)/4;
    ctx->vtGlobal.clear();
    ctx->vtOfPixelHost.clear();
    ctx->tileStart.clear();
    if (ctx->trackRecords) {
        for (uint32_t i = 0; i < shard->n_pixels; ++i) {
            const uint32_t x = shard->pixel_index[i] % s->camera.res_x, y = shard->pixel_index[i]/s->camera.res_x;
            const uint32_t tile = x/16 + (y/16)*((s->camera.res_x + 15)/16);
            if (i == 0 || tile != ctx->lastTileOfUpload) ctx->tileStart.push_back(i);
            ctx->lastTileOfUpload = tile;
        }
        ctx->tileStart.push_back(shard->n_pixels);
        std::vector<uint32_t> gids(shard->n_pixels);
        for (uint32_t i = 0; i < shard->n_pixels; ++i) {
            uint32_t x = shard->pixel_index[i] % s->camera.res_x, y = shard->pixel_index[i]/s->camera.res_x;
            gids[i] = x/4 + (y/4)*ctx->varW;
        }
        ctx->vtGlobal = gids;
        std::sort(ctx->vtGlobal.begin(), ctx->vtGlobal.end());
        ctx->vtGlobal.erase(std::unique(ctx->vtGlobal.begin(), ctx->vtGlobal.end()), ctx->vtGlobal.end());
        const uint32_t nVt = uint32_t(ctx->vtGlobal.size());
        std::vector<uint32_t> vtPix(size_t(nVt)*16, 0xFFFFFFFFu);
        ctx->vtOfPixelHost.resize(shard->n_pixels);
        for (uint32_t i = 0; i < shard->n_pixels; ++i) {
            uint32_t local = uint32_t(std::lower_bound(ctx->vtGlobal.begin(), ctx->vtGlobal.end(), gids[i]) - ctx->vtGlobal.begin());
            uint32_t x = shard->pixel_index[i] % s->camera.res_x, y = shard->pixel_index[i]/s->camera.res_x;
            ctx->vtOfPixelHost[i] = local;
            vtPix[size_t(local)*16 + (y % 4)*4 + (x % 4)] = i; // renderTile visits a block row by row (:136-156)
        }
        // a variance tile must be wholly inside the shard (16x16 tiles are dealt whole)
        for (uint32_t v = 0; v < nVt; ++v) {
            uint32_t gx = ctx->vtGlobal[v] % ctx->varW, gy = ctx->vtGlobal[v]/ctx->varW;
            uint32_t expect = std::min(4u, s->camera.res_x - gx*4)*std::min(4u, s->camera.res_y - gy*4), have = 0;
            for (int j = 0; j < 16; ++j) have += vtPix[size_t(v)*16 + j] != 0xFFFFFFFFu;
            if (have != expect) { fail(ctx, "adaptive sampling needs shards made of whole 4x4 variance tiles"); return TGPU_ERR_INVALID; }
        }
        uint32_t *dVtPix, *dVtOf, *dRecCount, *dJobOff, *dJobIdx;
        float *dRecMean, *dRecVar;
        if ((rc = upload(ctx, ctx->stateAllocs, vtPix.data(), vtPix.size(), &dVtPix))) return rc;
        if ((rc = upload(ctx, ctx->stateAllocs, ctx->vtOfPixelHost.data(), ctx->vtOfPixelHost.size(), &dVtOf))) return rc;
        if ((rc = upload<uint32_t>(ctx, ctx->stateAllocs, nullptr, nVt, &dRecCount))) return rc;
        if ((rc = upload<float>(ctx, ctx->stateAllocs, nullptr, nVt, &dRecMean))) return rc;
        if ((rc = upload<float>(ctx, ctx->stateAllocs, nullptr, nVt, &dRecVar))) return rc;
        if ((rc = upload<uint32_t>(ctx, ctx->stateAllocs, nullptr, size_t(TG_RING)*(shard->n_pixels + 1), &dJobOff))) return rc;
        if ((rc = upload<uint32_t>(ctx, ctx->stateAllocs, nullptr, size_t(TG_RING)*nVt, &dJobIdx))) return rc;
        st.nVt = nVt; st.vtPix = dVtPix; st.vtOfPixel = dVtOf; st.recCount = dRecCount; st.recMean = dRecMean; st.recVar = dRecVar;
        st.jobOffset = dJobOff; st.jobVtIndex = dJobIdx;
    }

    // ---- wavefront state: a pool of `cap` path slots + the per-sample result ring ----
    // default pool: up to 16 Mi slots (4 GiB of records + 1.3 GiB of lists out of 288 GB): the bigger the launches, the less
    // the drain of each kernel's last waves weighs (1 / 2 / 4 / 8 / 16 / 32 Mi slots: 350 / 512 / 643 / 733 / 789 / 797 Msamples/s)
    // (an eighth of the image, the shard of one of 8 ranks: 530 Msamples/s with 32 slots per pixel, 615 with 16 Mi slots)
    uint64_t cap = maxPaths ? maxPaths : std::min<uint64_t>(std::max<uint64_t>(uint64_t(shard->n_pixels)*256, 4096), 16ull << 20);
    if (cap < 64) cap = 64;
    if (cap > (1ull << 30)) { fail(ctx, "max_paths_in_flight too large"); return TGPU_ERR_INVALID; }
    ctx->capacity = cap;
    {
        // two 128-byte records per slot (kernels.h, DState)
        float4 *rec, *shrec;
        if ((rc = upload<float4>(ctx, ctx->stateAllocs, nullptr, cap*TG_REC_F4, &rec))) return rc;
        if ((rc = upload<float4>(ctx, ctx->stateAllocs, nullptr, cap*TG_REC_F4, &shrec))) return rc;
        st.rayO.base = rec; st.rayD.base = rec + 1; st.tp.base = rec + 2; st.hit.base = rec + 3; st.em.base = rec + 4;
        st.samp.base = reinterpret_cast<uint4 *>(rec + 5);
        uint32_t *misc = reinterpret_cast<uint32_t *>(rec + 6);
        st.med.base = misc; st.hitInst.base = misc + 1; st.state.base = misc + 2; st.sampleRef.base = misc + 3;
        st.tpSum.base = reinterpret_cast<float *>(rec + 7);
        st.shA.base = shrec; st.shB.base = shrec + 1; st.shO.base = shrec + 2; st.shX.base = shrec + 3;
        st.shC.base = shrec + 4; st.shD.base = shrec + 5; st.shE.base = shrec + 6; st.shF.base = shrec + 7;
    }
    if (sc.hasForward) {
        float4 **fw[] = {&st.shG, &st.shH, &st.shI};
        for (float4 **pp : fw)
            if ((rc = upload<float4>(ctx, ctx->stateAllocs, nullptr, cap, pp))) return rc;
    }
    {
        std::vector<uint32_t> ident(cap);
        for (uint64_t i = 0; i < cap; ++i) ident[i] = uint32_t(i);
        if ((rc = upload(ctx, ctx->stateAllocs, ident.data(), ident.size(), &st.freeQ))) return rc;
    }
    for (int set = 0; set < 2; ++set)
        for (int l = 0; l < TG_LISTS; ++l)
            if ((rc = upload<uint32_t>(ctx, ctx->stateAllocs, nullptr, cap, &st.lists[set][l]))) return rc;
    for (int set = 0; set < 2; ++set) // one byte per entry of the set's concatenated lists (their lengths add up to <= cap)
        if ((rc = upload<uint8_t>(ctx, ctx->stateAllocs, nullptr, cap + 256, &st.endFlag[set]))) return rc;
    if ((rc = upload<uint32_t>(ctx, ctx->stateAllocs, nullptr, CNT_TOTAL*TG_CNT_STRIDE, &st.counters))) return rc;
    if ((rc = upload<unsigned long long>(ctx, ctx->stateAllocs, nullptr, 8 + 128, &st.stats))) return rc; // [8..135]: tail-kernel timeline (profiling build)
    {
        uint32_t freeInit = uint32_t(cap);
        HIP_TRY(ctx, hipMemcpy(st.counters + CNT(CNT_FREE), &freeInit, sizeof(uint32_t), hipMemcpyHostToDevice));
        ctx->freeCount = freeInit;
    }
    ctx->resultSpp = 16;
    while (uint64_t(shard->n_pixels)*ctx->resultSpp*TG_RING > (1ull << 30) && ctx->resultSpp > 1) ctx->resultSpp /= 2; // <= 16 GiB of results
    st.nPixels = shard->n_pixels;
    st.resultSlice = shard->n_pixels*ctx->resultSpp;
    if (ctx->trackRecords) st.resultSlice += st.resultSlice/2 + 64; // adaptive passes move samples between regions/ranks
    if ((rc = upload<float4>(ctx, ctx->stateAllocs, nullptr, size_t(st.resultSlice)*TG_RING, &st.results))) return rc;
    if (anyAux) { // _trackOutputValues (PathTracer.cpp:11): one auxiliary record per sample next to its radiance
        if ((rc = upload<float4>(ctx, ctx->stateAllocs, nullptr, size_t(st.resultSlice)*TG_RING, &st.aov0))) return rc;
        if ((rc = upload<float4>(ctx, ctx->stateAllocs, nullptr, size_t(st.resultSlice)*TG_RING, &st.aov1))) return rc;
    }

    ctx->stats = tgpu_stats{};
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    ctx->uploadMs = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - uploadStart).count();
    ctx->bvhBuildMs = bvhBuildMs;
    ctx->bvhOnDevice = deviceBvh.nodes != nullptr;
    ctx->uploaded = true;
    ctx->abortFlag = false;
    if (!ctx->worker.joinable()) ctx->worker = std::thread(workerMain, ctx);
    return TGPU_OK;
}

// ---- the wavefront driver (runs on the context's worker thread) ---------------------------------------------
// One iteration = extend -> miss/shade (per BSDF class) -> shadow -> retire+regenerate, followed by one readback
// of the queue counters.  Jobs (spp ranges) are admitted while a slot of the result ring is free; a job's
// samples are folded into the framebuffer (k_accumulate) once all of them have retired, in job order.
// Poll instead of blocking: the per-iteration readback is latency critical, and a blocking
// hipStreamSynchronize from a non-main thread was measured at ~140 ms per call when another HIP user
// (torch) lives in the same process.
static hipError_t spinSync(hipStream_t stream)
{
    hipError_t e;
    while ((e = hipStreamQuery(stream)) == hipErrorNotReady) std::this_thread::yield();
    return e;
}

// every slot back on the free list, all counters cleared (after an abort or an error)
static void resetPool(tgpu_ctx *ctx)
{
    std::vector<uint32_t> ident(ctx->capacity);
    for (uint64_t i = 0; i < ctx->capacity; ++i) ident[i] = uint32_t(i);
    (void)hipStreamSynchronize(ctx->stream);
    (void)hipMemcpy(ctx->st.freeQ, ident.data(), ident.size()*sizeof(uint32_t), hipMemcpyHostToDevice);
    (void)hipMemset(ctx->st.counters, 0, CNT_TOTAL*TG_CNT_STRIDE*sizeof(uint32_t));
    uint32_t freeInit = uint32_t(ctx->capacity);
    (void)hipMemcpy(ctx->st.counters + CNT(CNT_FREE), &freeInit, sizeof(uint32_t), hipMemcpyHostToDevice);
    ctx->freeCount = freeInit;
    ctx->nCur = 0;
    ctx->aliveCount = 0;
    for (int l = 0; l < TG_LISTS; ++l) ctx->listCount[l] = 0;
}

static int driveIteration(tgpu_ctx *ctx)
{
    DScene &sc = ctx->sc;
    DState &st = ctx->st;
    hipStream_t stream = ctx->stream;
    const bool count = (ctx->profFlags & 1u) != 0;

    // admit pending jobs into free ring slots
    {
        std::lock_guard<std::mutex> lk(ctx->mtx);
        while (!ctx->pending.empty() && ctx->active.size() < TG_RING && ctx->ringFree) {
            tgpu_ctx::Job j = ctx->pending.front();
            ctx->pending.pop_front();
            j.ringSlot = uint32_t(__builtin_ctz(ctx->ringFree));
            ctx->ringFree &= ~(1u << j.ringSlot);
            j.start = ctx->globalCursor;
            ctx->globalCursor += j.total;
            ctx->active.push_back(std::move(j));
            const tgpu_ctx::Job &a = ctx->active.back();
            if (a.adaptive) {
                (void)hipMemcpyAsync(st.jobOffset + size_t(a.ringSlot)*(ctx->nPixels + 1), a.offsets.data(), a.offsets.size()*sizeof(uint32_t), hipMemcpyHostToDevice, stream);
                (void)hipMemcpyAsync(st.jobVtIndex + size_t(a.ringSlot)*st.nVt, a.vtIndex.data(), a.vtIndex.size()*sizeof(uint32_t), hipMemcpyHostToDevice, stream);
            }
        }
    }
    JobRing ring{};
    ring.n = uint32_t(ctx->active.size());
    unsigned long long limit = ctx->sampleBase;
    for (uint32_t i = 0; i < ring.n; ++i) {
        const tgpu_ctx::Job &j = ctx->active[i];
        ring.start[i] = j.start;
        ring.sppBegin[i] = j.sppBegin;
        ring.sppCount[i] = j.sppCount;
        ring.ringSlot[i] = j.ringSlot;
        ring.adaptive[i] = j.adaptive ? 1u : 0u;
        limit = j.start + j.total;
    }
    // jobs whose samples have all been handed out need no entry for g -> job lookups of NEW samples, but
    // the lookup walks start[] in order, so keep them: start[] is increasing and g >= sampleBase.

    const int cur = ctx->curQ, out = cur ^ 1;
    // zero the output set's list lengths, the two dequeue heads and the claim counter
    (void)hipMemsetAsync(st.counters + CNT(CNT_LIST0 + out*TG_LISTS), 0, TG_LISTS*TG_CNT_STRIDE*sizeof(uint32_t), stream);
    (void)hipMemsetAsync(st.counters + CNT(CNT_HEAD_EXT), 0, 3*TG_CNT_STRIDE*sizeof(uint32_t), stream);
    uint32_t nCur = ctx->nCur;
    // idle slots on the free list take admitted samples first and join THIS iteration's input (list 0)
    {
        unsigned long long room = limit - ctx->sampleBase;
        uint32_t nStart = uint32_t(std::min<unsigned long long>(room, ctx->freeCount));
        if (nStart) {
            Timer t(ctx, &ctx->stats.other_ms);
            uint32_t freeBase = ctx->freeCount - nStart;
            hipLaunchKernelGGL(k_refill, dim3(gridFor(ctx, nStart)), dim3(256), 0, stream, sc, st, ring, cur, ctx->sampleBase, nStart, freeBase, ctx->listCount[0]);
            ctx->freeCount = freeBase;
            ctx->listCount[0] += nStart;
            ctx->sampleBase += nStart;
            nCur += nStart;
        }
    }
    // tail mode: nothing left to start and only a few long paths alive -> one launch finishes them all
    bool tail = false;
    if (nCur > 0 && ctx->aliveCount > 0 && ctx->aliveCount <= ctx->tailMax && ctx->sampleBase == limit) {
        std::lock_guard<std::mutex> lk(ctx->mtx);
        // ... or nothing that could be admitted: every slice of the result ring is taken, and none comes back before the slowest path of the
        // oldest job has ended -- finish the jobs in flight now instead of draining the pool one bounce per iteration
        tail = ctx->pending.empty() || ctx->ringFree == 0u || ctx->active.size() >= TG_RING;
    }
    if (tail) {
        Timer t(ctx, &ctx->stats.other_ms);
        // lanes per path: as many as keep the grid within the resident waves (profiles/r01_ab_notes.txt)
        uint32_t spread = ctx->tailSpread;
        while (spread > 1 && uint64_t(nCur)*spread > uint64_t(ctx->maxGrid)*256) spread /= 2;
        int grid = gridFor(ctx, uint64_t(nCur)*spread);
        // variants: COUNT (visit counters), INST ("instances" in the scene), FWD (forward-scattering surfaces: shadow walk)
        // ... and LEAN: the scene shades with the lean kernels (never together with FWD: forward-scattering surfaces select the general ones)
        const int variant = (ctx->generalShading || !ctx->leanTail ? 0 : 8) | (count ? 4 : 0) | (ctx->generalGeometry ? 2 : 0) | (sc.hasForward ? 1 : 0);
#define TG_LAUNCH_TAIL(C, I, F) hipLaunchKernelGGL((k_tail<C, I, F>), dim3(grid), dim3(256), ctx->ldsBytes, stream, sc, st, cur, spread)
#define TG_LAUNCH_TAIL_LEAN(C, I) hipLaunchKernelGGL((k_tail<C, I, false, true>), dim3(grid), dim3(256), ctx->ldsBytes, stream, sc, st, cur, spread)
        // plain geometry on top of that: four lanes per path, the NEE rays traced beside the continuation ray (k_tail_coop)
        if (ctx->coopTail && (variant == 8 || variant == 12)) grid = gridFor(ctx, uint64_t(nCur)*4);
        switch (variant) {
#define TG_LAUNCH_TAIL_COOP(C) { if (ctx->nestedBsdfs) hipLaunchKernelGGL((k_tail_coop<C, true>), dim3(grid), dim3(256), ctx->ldsBytes, stream, sc, st, cur); \
                                 else hipLaunchKernelGGL((k_tail_coop<C, false>), dim3(grid), dim3(256), ctx->ldsBytes, stream, sc, st, cur); }
        case 8: if (ctx->coopTail) TG_LAUNCH_TAIL_COOP(false) else TG_LAUNCH_TAIL_LEAN(false, false); break;
        case 10: TG_LAUNCH_TAIL_LEAN(false, true); break;
        case 12: if (ctx->coopTail) TG_LAUNCH_TAIL_COOP(true) else TG_LAUNCH_TAIL_LEAN(true, false); break;
        case 14: TG_LAUNCH_TAIL_LEAN(true, true); break;
        case 0: TG_LAUNCH_TAIL(false, false, false); break;
        case 1: TG_LAUNCH_TAIL(false, false, true); break;
        case 2: TG_LAUNCH_TAIL(false, true, false); break;
        case 3: TG_LAUNCH_TAIL(false, true, true); break;
        case 4: TG_LAUNCH_TAIL(true, false, false); break;
        case 5: TG_LAUNCH_TAIL(true, false, true); break;
        case 6: TG_LAUNCH_TAIL(true, true, false); break;
        default: TG_LAUNCH_TAIL(true, true, true); break;
        }
#undef TG_LAUNCH_TAIL
#undef TG_LAUNCH_TAIL_LEAN
#undef TG_LAUNCH_TAIL_COOP
        // every listed slot ends on the free list: both list sets are empty afterwards
        (void)hipMemsetAsync(st.counters + CNT(CNT_LIST0), 0, 2*TG_LISTS*TG_CNT_STRIDE*sizeof(uint32_t), stream);
        ctx->stats.tail_launches++;
        if (count && getenv("TGPU_TAIL_TIMELINE")) { // per-vertex timeline of the tail kernel's slowest path (profiling build only)
            unsigned long long tl[128];
            if (hipStreamSynchronize(stream) == hipSuccess && hipMemcpy(tl, st.stats + 8, sizeof(tl), hipMemcpyDeviceToHost) == hipSuccess) {
                std::fprintf(stderr, "tail timeline (%u paths, spread %u), us at vertex n:", nCur, spread);
                for (int i = 0; i < 128 && (i == 0 || tl[i]); ++i) std::fprintf(stderr, " %.1f", double(tl[i])*0.01);
                std::fprintf(stderr, "\n");
                (void)hipMemset(st.stats + 8, 0, sizeof(tl));
            }
        }
    } else if (nCur > 0) {
        int grid = gridFor(ctx, nCur);
        {
            Timer t(ctx, &ctx->stats.extend_ms);
            // scenes with instances / spheres / disks: the nested-loop kernel stays the default for closest hits -- the persistent-lane
            // form around the flat state machine (k_extend_iq + k_classify, TGPU_INST_EXTEND_Q=1) doubles the lanes busy per
            // instruction but spends them on its own bookkeeping: 81 against 64 ms per step on config 4 (profiles/r02_ab_notes.txt).
            // For shadow rays the same state machine wins (k_shadow_iq: 29 against 49 ms).
            if (ctx->generalGeometry && ctx->instExtendQueue) {
                if (count) hipLaunchKernelGGL((k_extend_iq<true, 4>), dim3(grid), dim3(256), ctx->ldsBytes, stream, sc, st, cur);
                else hipLaunchKernelGGL((k_extend_iq<false, 4>), dim3(grid), dim3(256), ctx->ldsBytes, stream, sc, st, cur);
                hipLaunchKernelGGL(k_classify, dim3(grid), dim3(256), 0, stream, sc, st, cur);
            } else if (ctx->generalGeometry) {
                if (count) hipLaunchKernelGGL((k_extend<true, false, true>), dim3(grid), dim3(256), ctx->ldsBytes, stream, sc, st, cur, 0u);
                else hipLaunchKernelGGL((k_extend<false, false, true>), dim3(grid), dim3(256), ctx->ldsBytes, stream, sc, st, cur, 0u);
            } else {
                DScene scq = sc; // the queue kernels' copy: with the LDS nodelets, if any
                scq.ldsNodes = ctx->nodeletCount;
                const size_t ldsQ = ctx->ldsBytes + size_t(ctx->nodeletCount)*64;
                if (ctx->extendQueue && ctx->extendPersistent) { // persistent lanes + per-wave buffers of finished rays behind the stacks (and nodelets)
                    scq.ldsQueueWord = uint32_t(ldsQ/sizeof(uint32_t));
                    const size_t ldsP = ldsQ + size_t(4)*TG_EXTP_BUF*sizeof(uint2);
                    if (count) hipLaunchKernelGGL((k_extend_p<true>), dim3(grid), dim3(256), ldsP, stream, scq, st, cur);
                    else hipLaunchKernelGGL((k_extend_p<false>), dim3(grid), dim3(256), ldsP, stream, scq, st, cur);
                } else if (ctx->extendQueue == 8) { // queue form: eight (or four) rays per lane (kernels.h, k_extend_q)
                    if (count) hipLaunchKernelGGL((k_extend_q<true, 8>), dim3(grid), dim3(256), ldsQ, stream, scq, st, cur);
                    else hipLaunchKernelGGL((k_extend_q<false, 8>), dim3(grid), dim3(256), ldsQ, stream, scq, st, cur);
                } else if (ctx->extendQueue) {
                    if (count) hipLaunchKernelGGL((k_extend_q<true, 4>), dim3(grid), dim3(256), ldsQ, stream, scq, st, cur);
                    else hipLaunchKernelGGL((k_extend_q<false, 4>), dim3(grid), dim3(256), ldsQ, stream, scq, st, cur);
                } else if (count) hipLaunchKernelGGL((k_extend<true, false, false>), dim3(grid), dim3(256), ctx->ldsBytes, stream, sc, st, cur, 0u);
                else hipLaunchKernelGGL((k_extend<false, false, false>), dim3(grid), dim3(256), ctx->ldsBytes, stream, sc, st, cur, 0u);
            }
        }
        ctx->stats.extend_launches++;
        {
            Timer t(ctx, &ctx->stats.shade_ms);
            // one launch per BSDF class that occurs in the scene (ctx->classMask, set at upload)
#define TG_LAUNCH_SHADE(C) if (ctx->classMask & (1u << (C))) { \
                ctx->stats.shade_launches++; \
                if (ctx->generalShading) hipLaunchKernelGGL((k_shade<C, true>), dim3(grid), dim3(256), 0, stream, sc, st, out); \
                else hipLaunchKernelGGL((k_shade<C, false>), dim3(grid), dim3(256), 0, stream, sc, st, out); }
            TG_LAUNCH_SHADE(0) TG_LAUNCH_SHADE(1) TG_LAUNCH_SHADE(2) TG_LAUNCH_SHADE(3) TG_LAUNCH_SHADE(4) TG_LAUNCH_SHADE(5) TG_LAUNCH_SHADE(TG_CLASS_OTHER)
#undef TG_LAUNCH_SHADE
            if (sc.nMedia) { ctx->stats.shade_launches++; hipLaunchKernelGGL(k_volume, dim3(grid), dim3(256), 0, stream, sc, st, out); }
            if (sc.nMedia && ctx->nestedBsdfs) { ctx->stats.shade_launches++; hipLaunchKernelGGL(k_volume_nested, dim3(grid), dim3(256), 0, stream, sc, st, out); }
        }
        // the queue forms of the shadow kernel (visibility bits out; the next kernel that touches the path folds the NEE terms in)
        // serve every scene whose shadow rays are single segments; forward-scattering scenes keep k_shadow's multi-segment walk
        const bool shadowQ = ctx->shadowQueue && !sc.hasForward;
        {
            Timer t(ctx, &ctx->stats.shadow_ms);
            if (shadowQ && ctx->generalGeometry) {
                if (count) hipLaunchKernelGGL((k_shadow_iq<true, 4>), dim3(grid), dim3(256), ctx->ldsBytes, stream, sc, st, out);
                else hipLaunchKernelGGL((k_shadow_iq<false, 4>), dim3(grid), dim3(256), ctx->ldsBytes, stream, sc, st, out);
            } else if (shadowQ) {
                DScene scq = sc;
                scq.ldsNodes = ctx->nodeletCount;
                const size_t ldsQ = ctx->ldsBytes + size_t(ctx->nodeletCount)*64;
                if (ctx->shadowQueue == 8) {
                    if (count) hipLaunchKernelGGL((k_shadow_q<true, 8>), dim3(grid), dim3(256), ldsQ, stream, scq, st, out);
                    else hipLaunchKernelGGL((k_shadow_q<false, 8>), dim3(grid), dim3(256), ldsQ, stream, scq, st, out);
                } else {
                    if (count) hipLaunchKernelGGL((k_shadow_q<true, 4>), dim3(grid), dim3(256), ldsQ, stream, scq, st, out);
                    else hipLaunchKernelGGL((k_shadow_q<false, 4>), dim3(grid), dim3(256), ldsQ, stream, scq, st, out);
                }
            } else {
            const int variant = (count ? 4 : 0) | (ctx->generalGeometry ? 2 : 0) | (sc.hasForward ? 1 : 0);
#define TG_LAUNCH_SHADOW(C, I, F) hipLaunchKernelGGL((k_shadow<C, I, F>), dim3(grid), dim3(256), ctx->ldsBytes, stream, sc, st, out)
            switch (variant) {
            case 0: TG_LAUNCH_SHADOW(false, false, false); break;
            case 1: TG_LAUNCH_SHADOW(false, false, true); break;
            case 2: TG_LAUNCH_SHADOW(false, true, false); break;
            case 3: TG_LAUNCH_SHADOW(false, true, true); break;
            case 4: TG_LAUNCH_SHADOW(true, false, false); break;
            case 5: TG_LAUNCH_SHADOW(true, false, true); break;
            case 6: TG_LAUNCH_SHADOW(true, true, false); break;
            default: TG_LAUNCH_SHADOW(true, true, true); break;
            }
#undef TG_LAUNCH_SHADOW
            }
        }
        ctx->stats.shadow_launches++;
        {
            Timer t(ctx, &ctx->stats.other_ms, &ctx->stats.retire_ms);
            ctx->stats.retire_launches++;
            const uint64_t perRound = uint64_t(TG_RETIRE_BLOCK)*TG_RETIRE_K;
            int g1024 = int(std::min<uint64_t>((uint64_t(nCur) + perRound - 1)/perRound, uint64_t(ctx->numCUs)*(2048/TG_RETIRE_BLOCK)));
            hipLaunchKernelGGL(k_retire_regenerate, dim3(std::max(g1024, 1)), dim3(TG_RETIRE_BLOCK), 0, stream, sc, st, ring, out, ctx->sampleBase, limit);
        }
    }
    hipError_t e = hipMemcpyAsync(ctx->hostCounters, st.counters, CNT_TOTAL*TG_CNT_STRIDE*sizeof(uint32_t), hipMemcpyDeviceToHost, stream);
    if (e == hipSuccess) e = spinSync(stream);
    if (e != hipSuccess) {
        fail(ctx, "wavefront iteration failed: %s", hipGetErrorString(e));
        return TGPU_ERR_HIP;
    }
    const uint32_t *hc = ctx->hostCounters;
    // TGPU_TRACE_SLOTS=1 (pools of <= 4096 slots, i.e. the one-pixel shards of tests/tools/trace_path.py): both records of every
    // live slot after each iteration, as hex words, to be diffed against the oracle's ORACLE_TRACE output.  Host side only:
    // the kernels are the production binaries.
    if (ctx->traceSlots && ctx->capacity <= 4096) {
        std::vector<uint32_t> rec(size_t(ctx->capacity)*32), sh(size_t(ctx->capacity)*32);
        (void)hipMemcpy(rec.data(), st.rayO.base, rec.size()*4, hipMemcpyDeviceToHost);
        (void)hipMemcpy(sh.data(), st.shA.base, sh.size()*4, hipMemcpyDeviceToHost);
        if (ctx->tracePrev.size() != ctx->capacity) ctx->tracePrev.assign(size_t(ctx->capacity), uint32_t(END_DEAD));
        for (uint64_t s = 0; s < ctx->capacity; ++s) {
            const uint32_t *r = &rec[s*32], *h = &sh[s*32];
            const uint32_t end = (r[11] >> 24) & 0xFu;
            if (end == uint32_t(END_DEAD) && ctx->tracePrev[s] == uint32_t(END_DEAD)) continue;
            ctx->tracePrev[s] = end;
            fprintf(stderr, "GPU it=%llu slot=%llu%s hit t=%08x u=%08x v=%08x prim=%08x inst=%08x | O=%08x %08x %08x tn=%08x D=%08x %08x %08x | tp=%08x %08x %08x bounce=%u spec=%u end=%u "
                            "| em=%08x %08x %08x | dim=%u pcg=%08x%08x idx=%u med=%08x state=%02x\n",
                    (unsigned long long)ctx->traceIter, (unsigned long long)s, tail ? " (tail)" : "", r[12], r[13], r[14], r[15], r[25], r[0], r[1], r[2], r[3], r[4], r[5], r[6], r[8], r[9], r[10],
                    r[11] & 0xFFFFu, (r[11] >> 16) & 1u, end, r[16], r[17], r[18], r[20], r[22], r[21], r[23], r[24], r[26]);
            if (r[26] & 1u)
                fprintf(stderr, "    nee dL=%08x %08x %08x tfarL=%08x | dB=%08x %08x %08x tfarB=%08x | L=%08x %08x %08x pick=%08x | B=%08x %08x %08x vis=%08x | tpV=%08x %08x %08x bits=%08x | F=%08x %08x %08x %08x | cap=%08x\n",
                        h[0], h[1], h[2], h[3], h[4], h[5], h[6], h[7], h[16], h[17], h[18], h[19], h[20], h[21], h[22], h[23], h[24], h[25], h[26], h[27], h[28], h[29], h[30], h[31], h[12]);
        }
        ctx->traceIter++;
    }
    if (getenv("TGPU_DEBUG")) fprintf(stderr, "[tgpu] t=%.3f ms alive=%llu%s iter in=%u out=[%u %u %u %u %u] claims=%u free=%u base=%llu limit=%llu active=%zu\n",                                       std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count() - 1e3*std::floor(1e-3*std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count()/1e3)*1e3,
                                      (unsigned long long)ctx->aliveCount, tail ? " (tail)" : "", nCur,
                                      hc[CNT(CNT_LIST0 + out*TG_LISTS)], hc[CNT(CNT_LIST0 + out*TG_LISTS + 1)], hc[CNT(CNT_LIST0 + out*TG_LISTS + 2)], hc[CNT(CNT_LIST0 + out*TG_LISTS + 3)],
                                      hc[CNT(CNT_LIST0 + out*TG_LISTS + 4)], hc[CNT(CNT_REGEN)], hc[CNT(CNT_FREE)], ctx->sampleBase, limit, ctx->active.size());
    uint64_t nNext = 0;
    if (tail) {
        for (int l = 0; l < TG_LISTS; ++l) ctx->listCount[l] = 0;
    } else if (nCur > 0) {
        for (int l = 0; l < TG_LISTS; ++l) {
            ctx->listCount[l] = hc[CNT(CNT_LIST0 + out*TG_LISTS + l)];
            nNext += ctx->listCount[l];
        }
        ctx->curQ = out;
    } else {
        for (int l = 0; l < TG_LISTS; ++l) nNext += ctx->listCount[l];
    }
    const uint32_t dropped = hc[CNT(CNT_FREE)] - ctx->freeCount; // idle slots extend moved to the free list
    ctx->freeCount = hc[CNT(CNT_FREE)];
    if (nNext + ctx->freeCount != ctx->capacity) { fail(ctx, "internal error: slot accounting broke (%llu listed + %u free != %llu)", (unsigned long long)nNext, ctx->freeCount, (unsigned long long)ctx->capacity); return TGPU_ERR_HIP; }
    if (!tail) ctx->hostExtendRays += nCur - dropped; // the tail kernel counts its rays on the device
    unsigned long long room = limit - ctx->sampleBase;
    unsigned long long started = std::min<unsigned long long>(hc[CNT(CNT_REGEN)], room);
    ctx->sampleBase += started;
    // slots that asked for a sample and did not get one are idle: alive = listed - idle
    uint64_t idle = hc[CNT(CNT_REGEN)] - started;
    ctx->nCur = uint32_t(nNext);
    ctx->aliveCount = nNext - idle;

    bool retired = false;
    // retire finished jobs in order
    while (!ctx->active.empty()) {
        tgpu_ctx::Job &j = ctx->active.front();
        if (hc[CNT(CNT_JOB_DONE0 + j.ringSlot)] != j.total) break;
        {
            Timer t(ctx, &ctx->stats.other_ms);
            if (ctx->trackRecords) hipLaunchKernelGGL(k_accumulate_records, dim3(gridFor(ctx, st.nVt)), dim3(256), 0, stream, st, j.ringSlot, j.sppCount, j.adaptive ? 1u : 0u);
            else hipLaunchKernelGGL(k_accumulate, dim3(gridFor(ctx, ctx->nPixels)), dim3(256), 0, stream, st, j.ringSlot, j.sppCount);
        }
        (void)hipMemsetAsync(st.counters + CNT(CNT_JOB_DONE0 + j.ringSlot), 0, sizeof(uint32_t), stream);
        ctx->stats.samples += j.total;
        ctx->ringFree |= 1u << j.ringSlot;
        ctx->active.pop_front();
        retired = true;
    }
    // every iteration must start samples, advance paths or retire a job; anything else is a bookkeeping bug
    if (ctx->aliveCount == 0 && ctx->sampleBase == limit && !retired && !ctx->active.empty()) {
        fail(ctx, "internal error: wavefront stalled with %zu unfinished job(s)", ctx->active.size());
        return TGPU_ERR_HIP;
    }
    return TGPU_OK;
}

static void workerMain(tgpu_ctx *ctx)
{
    (void)hipSetDevice(ctx->device);
    while (true) {
        {
            std::unique_lock<std::mutex> lk(ctx->mtx);
            ctx->cvWork.wait(lk, [&] { return ctx->shutdown || !ctx->pending.empty(); });
            if (ctx->shutdown) return;
        }
        int rc = TGPU_OK;
        while (rc == TGPU_OK) {
            bool more;
            {
                std::lock_guard<std::mutex> lk(ctx->mtx);
                more = !ctx->pending.empty() || !ctx->active.empty() || ctx->aliveCount > 0;
            }
            if (!more) break;
            if (ctx->abortFlag.exchange(false)) { rc = TGPU_ERR_ABORTED; break; } // consumed here, and only here
            rc = driveIteration(ctx);
        }
        (void)hipStreamSynchronize(ctx->stream);
        {
            std::lock_guard<std::mutex> lk(ctx->mtx);
            if (rc != TGPU_OK) {
                // drop everything that was in flight; the framebuffer keeps the passes accumulated so far
                ctx->pending.clear();
                ctx->active.clear();
                ctx->nCur = 0;
                ctx->ringFree = TG_RING_ALL;
                ctx->sampleBase = ctx->globalCursor;
                resetPool(ctx);
                if (rc == TGPU_ERR_ABORTED) ctx->error = "render aborted";
                ctx->lastRc = rc;
            }
            ctx->idle = ctx->pending.empty();
            if (ctx->idle) {
                ctx->abortFlag = false; // an abort that arrived after the last iteration has nothing left to stop (tgpu_abort sets it under this mutex)
                ctx->cvIdle.notify_all();
            }
        }
    }
}

int tgpu_render_pass(tgpu_ctx *ctx, uint32_t sppBegin, uint32_t sppCount, void *streamPtr)
{
    if (!ctx) return TGPU_ERR_INVALID;
    if (!ctx->uploaded) { fail(ctx, "tgpu_render_pass before tgpu_upload_scene"); return TGPU_ERR_STATE; }
    if (sppCount == 0) return TGPU_OK;
    if (ctx->trackRecords && sppCount > ctx->resultSpp) {
        // a SampleRecord sees ALL samples of a pixel before the next pixel's (renderTile order), so the pass cannot be cut into spp
        // slices like below: it goes the way of the adaptive passes (pixel-major jobs, cut along 16x16 tiles) with one count for all
        std::vector<uint32_t> index(size_t(ctx->varW)*ctx->varH, sppBegin), cnt(size_t(ctx->varW)*ctx->varH, sppCount);
        return tgpu_render_pass_adaptive(ctx, index.data(), cnt.data(), streamPtr);
    }
    if (streamPtr) {
        // work submitted so far on the caller's stream must precede the pass: order the context stream after it
        HIP_TRY(ctx, hipSetDevice(ctx->device));
        hipEvent_t ev;
        HIP_TRY(ctx, hipEventCreateWithFlags(&ev, hipEventDisableTiming));
        HIP_TRY(ctx, hipEventRecord(ev, static_cast<hipStream_t>(streamPtr)));
        HIP_TRY(ctx, hipStreamWaitEvent(ctx->stream, ev, 0));
        HIP_TRY(ctx, hipEventDestroy(ev));
    }
    {
        std::lock_guard<std::mutex> lk(ctx->mtx);
        for (uint32_t s = 0; s < sppCount; s += ctx->resultSpp) { // one job per result-ring slice
            uint32_t n = std::min(ctx->resultSpp, sppCount - s);
            {
                tgpu_ctx::Job job;
                job.sppBegin = sppBegin + s; job.sppCount = n; job.start = 0; job.total = (unsigned long long)n*ctx->nPixels; job.ringSlot = 0;
                ctx->pending.push_back(std::move(job));
            }
        }
        ctx->idle = false;
    }
    ctx->cvWork.notify_one();
    return TGPU_OK;
}

int tgpu_render_pass_adaptive(tgpu_ctx *ctx, const uint32_t *vtSampleIndex, const uint32_t *vtSampleCount, void *streamPtr)
{
    if (!ctx || !vtSampleIndex || !vtSampleCount) return TGPU_ERR_INVALID;
    if (!ctx->uploaded) { fail(ctx, "tgpu_render_pass_adaptive before tgpu_upload_scene"); return TGPU_ERR_STATE; }
    if (!ctx->trackRecords) { fail(ctx, "tgpu_render_pass_adaptive needs settings.adaptive_sampling at upload"); return TGPU_ERR_STATE; }
    (void)streamPtr;
    // A pass whose samples do not fit one slice of the result ring (a shard that owns the noisy tiles of the image can be handed
    // several times its share, and spp_step may exceed 16) becomes several jobs along 16x16-tile boundaries: the pixels outside a
    // job have an empty range in it (offset[k + 1] == offset[k]), so the per-variance-tile sample order of k_accumulate_records
    // -- variance tiles never straddle a 16x16 tile -- is what a single job would give.
    std::vector<tgpu_ctx::Job> jobs;
    std::vector<uint32_t> vtIndex(ctx->vtGlobal.size());
    for (size_t v = 0; v < ctx->vtGlobal.size(); ++v) vtIndex[v] = vtSampleIndex[ctx->vtGlobal[v]];
    size_t t0 = 0; // first 16x16 tile of the job being filled
    while (t0 + 1 < ctx->tileStart.size()) {
        unsigned long long total = 0;
        size_t t1 = t0;
        while (t1 + 1 < ctx->tileStart.size()) {
            unsigned long long tileTotal = 0;
            for (uint32_t k = ctx->tileStart[t1]; k < ctx->tileStart[t1 + 1]; ++k) tileTotal += vtSampleCount[ctx->vtGlobal[ctx->vtOfPixelHost[k]]];
            if (total + tileTotal > ctx->st.resultSlice) break;
            total += tileTotal;
            ++t1;
        }
        if (t1 == t0) { fail(ctx, "adaptive pass: one 16x16 tile alone exceeds the result-ring slice (%u samples)", ctx->st.resultSlice); return TGPU_ERR_UNSUPPORTED; }
        if (total) {
            tgpu_ctx::Job job;
            job.adaptive = true;
            job.sppBegin = 0; job.sppCount = 0; job.start = 0; job.ringSlot = 0;
            job.offsets.resize(size_t(ctx->nPixels) + 1);
            uint32_t run = 0;
            for (uint32_t k = 0; k < ctx->nPixels; ++k) {
                job.offsets[k] = run;
                if (k >= ctx->tileStart[t0] && k < ctx->tileStart[t1]) run += vtSampleCount[ctx->vtGlobal[ctx->vtOfPixelHost[k]]];
            }
            job.offsets[ctx->nPixels] = run;
            job.total = total;
            job.vtIndex = vtIndex;
            jobs.push_back(std::move(job));
        }
        t0 = t1;
    }
    if (jobs.empty()) return TGPU_OK;
    {
        std::lock_guard<std::mutex> lk(ctx->mtx);
        for (tgpu_ctx::Job &j : jobs) ctx->pending.push_back(std::move(j));
        ctx->idle = false;
    }
    ctx->cvWork.notify_one();
    return TGPU_OK;
}

int tgpu_download_records(tgpu_ctx *ctx, tgpu_sample_record *records)
{
    if (!ctx || !records) return TGPU_ERR_INVALID;
    if (!ctx->uploaded || !ctx->trackRecords) { fail(ctx, "no sample records: upload a scene with settings.adaptive_sampling"); return TGPU_ERR_STATE; }
    (void)waitIdle(ctx);
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    const size_t n = ctx->vtGlobal.size();
    std::vector<uint32_t> cnt(n);
    std::vector<float> mean(n), var(n);
    HIP_TRY(ctx, hipMemcpy(cnt.data(), ctx->st.recCount, n*sizeof(uint32_t), hipMemcpyDeviceToHost));
    HIP_TRY(ctx, hipMemcpy(mean.data(), ctx->st.recMean, n*sizeof(float), hipMemcpyDeviceToHost));
    HIP_TRY(ctx, hipMemcpy(var.data(), ctx->st.recVar, n*sizeof(float), hipMemcpyDeviceToHost));
    for (size_t v = 0; v < n; ++v) records[ctx->vtGlobal[v]] = tgpu_sample_record{cnt[v], mean[v], var[v]};
    return TGPU_OK;
}

int tgpu_upload_records(tgpu_ctx *ctx, const tgpu_sample_record *records)
{
    if (!ctx || !records) return TGPU_ERR_INVALID;
    if (!ctx->uploaded || !ctx->trackRecords) { fail(ctx, "no sample records: upload a scene with settings.adaptive_sampling"); return TGPU_ERR_STATE; }
    (void)waitIdle(ctx);
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    const size_t n = ctx->vtGlobal.size();
    std::vector<uint32_t> cnt(n);
    std::vector<float> mean(n), var(n);
    for (size_t v = 0; v < n; ++v) {
        const tgpu_sample_record &r = records[ctx->vtGlobal[v]];
        cnt[v] = r.sample_count; mean[v] = r.mean; var[v] = r.running_variance;
    }
    HIP_TRY(ctx, hipMemcpy(ctx->st.recCount, cnt.data(), n*sizeof(uint32_t), hipMemcpyHostToDevice));
    HIP_TRY(ctx, hipMemcpy(ctx->st.recMean, mean.data(), n*sizeof(float), hipMemcpyHostToDevice));
    HIP_TRY(ctx, hipMemcpy(ctx->st.recVar, var.data(), n*sizeof(float), hipMemcpyHostToDevice));
    return TGPU_OK;
}

int tgpu_wait(tgpu_ctx *ctx)
{
    if (!ctx) return TGPU_ERR_INVALID;
    int rc = waitIdle(ctx, true);
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    return rc;
}

int tgpu_abort(tgpu_ctx *ctx)
{
    if (!ctx) return TGPU_ERR_INVALID;
    std::lock_guard<std::mutex> lk(ctx->mtx);
    if (!ctx->idle) ctx->abortFlag = true; // the worker consumes it at its next iteration; passes queued later on are not affected by it
    return TGPU_OK;
}

int tgpu_download(tgpu_ctx *ctx, float *rgb, uint32_t *cnt)
{
    if (!ctx) return TGPU_ERR_INVALID;
    if (!ctx->uploaded) { fail(ctx, "tgpu_download before tgpu_upload_scene"); return TGPU_ERR_STATE; }
    (void)waitIdle(ctx);
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    if (rgb) HIP_TRY(ctx, hipMemcpy(rgb, ctx->st.fbMean, size_t(ctx->nPixels)*3*sizeof(float), hipMemcpyDeviceToHost));
    if (cnt) HIP_TRY(ctx, hipMemcpy(cnt, ctx->st.fbCount, size_t(ctx->nPixels)*sizeof(uint32_t), hipMemcpyDeviceToHost));
    return TGPU_OK;
}

int tgpu_resolve_to_device(tgpu_ctx *ctx, void *deviceRgb, void *streamPtr)
{
    if (!ctx || !deviceRgb) return TGPU_ERR_INVALID;
    if (!ctx->uploaded) { fail(ctx, "tgpu_resolve_to_device before tgpu_upload_scene"); return TGPU_ERR_STATE; }
    (void)waitIdle(ctx);
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    hipStream_t stream = streamPtr ? static_cast<hipStream_t>(streamPtr) : ctx->stream;
    if (stream != ctx->stream) HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    HIP_TRY(ctx, hipMemcpyAsync(deviceRgb, ctx->st.fbMean, size_t(ctx->nPixels)*3*sizeof(float), hipMemcpyDeviceToDevice, stream));
    return TGPU_OK;
}

int tgpu_upload_framebuffer(tgpu_ctx *ctx, const float *rgb, const uint32_t *cnt)
{
    if (!ctx || !rgb || !cnt) return TGPU_ERR_INVALID;
    if (!ctx->uploaded) { fail(ctx, "tgpu_upload_framebuffer before tgpu_upload_scene"); return TGPU_ERR_STATE; }
    (void)waitIdle(ctx);
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    HIP_TRY(ctx, hipMemcpy(ctx->st.fbMean, rgb, size_t(ctx->nPixels)*3*sizeof(float), hipMemcpyHostToDevice));
    HIP_TRY(ctx, hipMemcpy(ctx->st.fbCount, cnt, size_t(ctx->nPixels)*sizeof(uint32_t), hipMemcpyHostToDevice));
    return TGPU_OK;
}

static int outputPart(tgpu_ctx *ctx, int type, int part, void **ptr, size_t *bytes)
{
    if (!ctx) return TGPU_ERR_INVALID;
    if (!ctx->uploaded) { fail(ctx, "output buffer access before tgpu_upload_scene"); return TGPU_ERR_STATE; }
    if (type < 0 || type >= TGPU_OUTPUT_COUNT || part < 0 || part > TGPU_PART_COUNT) { fail(ctx, "no output buffer %d / part %d", type, part); return TGPU_ERR_INVALID; }
    const DOutBuf &d = ctx->st.out[type];
    const size_t channels = (type == TGPU_OUTPUT_DEPTH || type == TGPU_OUTPUT_VISIBILITY) ? 1 : 3;
    *ptr = part == TGPU_PART_A ? static_cast<void *>(d.a) : part == TGPU_PART_B ? static_cast<void *>(d.b) : part == TGPU_PART_VARIANCE ? static_cast<void *>(d.var)
                                                                                                                                       : static_cast<void *>(d.count);
    *bytes = size_t(ctx->nPixels)*(part == TGPU_PART_COUNT ? sizeof(uint32_t) : channels*sizeof(float));
    if (!*ptr) { fail(ctx, "output buffer %d has no part %d (not requested at upload)", type, part); return TGPU_ERR_INVALID; }
    (void)waitIdle(ctx);
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    return TGPU_OK;
}

int tgpu_download_output_buffer(tgpu_ctx *ctx, int type, int part, void *dst)
{
    void *p;
    size_t bytes;
    if (!dst) return TGPU_ERR_INVALID;
    if (int rc = outputPart(ctx, type, part, &p, &bytes)) return rc;
    HIP_TRY(ctx, hipMemcpy(dst, p, bytes, hipMemcpyDeviceToHost));
    return TGPU_OK;
}

int tgpu_upload_output_buffer(tgpu_ctx *ctx, int type, int part, const void *src)
{
    void *p;
    size_t bytes;
    if (!src) return TGPU_ERR_INVALID;
    if (int rc = outputPart(ctx, type, part, &p, &bytes)) return rc;
    HIP_TRY(ctx, hipMemcpy(p, src, bytes, hipMemcpyHostToDevice));
    return TGPU_OK;
}

static int traceRaw(tgpu_ctx *ctx, const float *rays, uint32_t n, float *hits, uint8_t *occ)
{
    if (!ctx || !rays || (!hits && !occ)) return TGPU_ERR_INVALID;
    if (!ctx->uploaded) { fail(ctx, "ray query before tgpu_upload_scene"); return TGPU_ERR_STATE; }
    (void)waitIdle(ctx);
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    DState &st = ctx->st;
    std::vector<float> o, d;
    uint8_t *dOcc = nullptr;
    if (occ) HIP_TRY(ctx, hipMalloc(reinterpret_cast<void **>(&dOcc), size_t(std::min<uint64_t>(n, ctx->capacity)) + 1));
    int rc = TGPU_OK;
    for (uint64_t base = 0; base < n && rc == TGPU_OK; base += ctx->capacity) {
        uint32_t m = uint32_t(std::min<uint64_t>(ctx->capacity, n - base));
        o.resize(size_t(m)*4);
        d.resize(size_t(m)*4);
        for (uint32_t i = 0; i < m; ++i) {
            const float *r = rays + (base + i)*8;
            o[size_t(i)*4 + 0] = r[0]; o[size_t(i)*4 + 1] = r[1]; o[size_t(i)*4 + 2] = r[2]; o[size_t(i)*4 + 3] = r[3];
            d[size_t(i)*4 + 0] = r[4]; d[size_t(i)*4 + 1] = r[5]; d[size_t(i)*4 + 2] = r[6]; d[size_t(i)*4 + 3] = r[7];
        }
        hipError_t e = hipMemcpy2DAsync(st.rayO.base, TG_REC_F4*16, o.data(), 16, 16, m, hipMemcpyHostToDevice, ctx->stream);
        if (e == hipSuccess) e = hipMemcpy2DAsync(st.rayD.base, TG_REC_F4*16, d.data(), 16, 16, m, hipMemcpyHostToDevice, ctx->stream);
        if (e == hipSuccess) e = hipMemsetAsync(st.counters + CNT(CNT_HEAD_EXT), 0, 3*TG_CNT_STRIDE*sizeof(uint32_t), ctx->stream);
        if (e == hipSuccess) {
            if (hits) {
                if (ctx->generalGeometry) hipLaunchKernelGGL((k_extend<false, true, true>), dim3(gridFor(ctx, m)), dim3(256), ctx->ldsBytes, ctx->stream, ctx->sc, st, 0, m);
                else hipLaunchKernelGGL((k_extend<false, true, false>), dim3(gridFor(ctx, m)), dim3(256), ctx->ldsBytes, ctx->stream, ctx->sc, st, 0, m);
                e = hipMemcpy2DAsync(hits + base*4, 16, st.hit.base, TG_REC_F4*16, 16, m, hipMemcpyDeviceToHost, ctx->stream);
            } else {
                if (ctx->generalGeometry) hipLaunchKernelGGL(k_occluded_raw<true>, dim3(gridFor(ctx, m)), dim3(256), ctx->ldsBytes, ctx->stream, ctx->sc, st, m, dOcc);
                else hipLaunchKernelGGL(k_occluded_raw<false>, dim3(gridFor(ctx, m)), dim3(256), ctx->ldsBytes, ctx->stream, ctx->sc, st, m, dOcc);
                e = hipMemcpyAsync(occ + base, dOcc, m, hipMemcpyDeviceToHost, ctx->stream);
            }
        }
        if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
        if (e != hipSuccess) { fail(ctx, "ray query failed: %s", hipGetErrorString(e)); rc = TGPU_ERR_HIP; }
    }
    if (dOcc) (void)hipFree(dOcc);
    if (rc == TGPU_OK && hits) {
        // a miss reports the ray's own tfar
        for (uint32_t i = 0; i < n; ++i) {
            uint32_t prim;
            std::memcpy(&prim, &hits[size_t(i)*4 + 3], 4);
            if (prim == 0xFFFFFFFFu) hits[size_t(i)*4] = rays[size_t(i)*8 + 7];
        }
    }
    return rc;
}

int tgpu_trace_closest(tgpu_ctx *ctx, const float *rays, uint32_t n, float *hits) { return traceRaw(ctx, rays, n, hits, nullptr); }
int tgpu_trace_occluded(tgpu_ctx *ctx, const float *rays, uint32_t n, uint8_t *occ) { return traceRaw(ctx, rays, n, nullptr, occ); }

int tgpu_set_profiling(tgpu_ctx *ctx, uint32_t flags)
{
    if (!ctx) return TGPU_ERR_INVALID;
    (void)waitIdle(ctx);
    ctx->profFlags = flags;
    return TGPU_OK;
}

int tgpu_get_stats(tgpu_ctx *ctx, tgpu_stats *out)
{
    if (!ctx || !out) return TGPU_ERR_INVALID;
    (void)waitIdle(ctx);
    if (ctx->uploaded) {
        HIP_TRY(ctx, hipSetDevice(ctx->device));
        HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
        unsigned long long d[8];
        HIP_TRY(ctx, hipMemcpy(d, ctx->st.stats, sizeof(d), hipMemcpyDeviceToHost));
        ctx->stats.extend_node_visits = d[0];
        ctx->stats.extend_tri_tests = d[1];
        ctx->stats.shadow_node_visits = d[2];
        ctx->stats.shadow_tri_tests = d[3];
        ctx->stats.shadow_rays = d[4];
        ctx->stats.surface_events = d[5];
        ctx->stats.extend_rays = ctx->hostExtendRays + d[6];
        ctx->stats.shadow_vertices = d[7];
    }
    ctx->stats.upload_ms = ctx->uploadMs;
    ctx->stats.bvh_build_ms = ctx->bvhBuildMs;
    ctx->stats.bvh_on_device = ctx->bvhOnDevice ? 1u : 0u;
    *out = ctx->stats;
    return TGPU_OK;
}

int tgpu_reset_stats(tgpu_ctx *ctx)
{
    if (!ctx) return TGPU_ERR_INVALID;
    (void)waitIdle(ctx);
    ctx->stats = tgpu_stats{};
    ctx->hostExtendRays = 0;
    if (ctx->uploaded) {
        HIP_TRY(ctx, hipSetDevice(ctx->device));
        HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
        HIP_TRY(ctx, hipMemset(ctx->st.stats, 0, 8*sizeof(unsigned long long)));
    }
    return TGPU_OK;
}

int tgpu_eval_math(tgpu_ctx *ctx, int kind, const float *a, const float *b, uint32_t n, float *out)
{
    if (!ctx || !a || !b || !out) return TGPU_ERR_INVALID;
    (void)waitIdle(ctx);
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    float *da = nullptr, *db = nullptr, *dout = nullptr;
    size_t bytes = size_t(n ? n : 1)*sizeof(float);
    HIP_TRY(ctx, hipMalloc(reinterpret_cast<void **>(&da), bytes));
    HIP_TRY(ctx, hipMalloc(reinterpret_cast<void **>(&db), bytes));
    HIP_TRY(ctx, hipMalloc(reinterpret_cast<void **>(&dout), bytes));
    hipError_t e = hipMemcpy(da, a, size_t(n)*4, hipMemcpyHostToDevice);
    if (e == hipSuccess) e = hipMemcpy(db, b, size_t(n)*4, hipMemcpyHostToDevice);
    if (e == hipSuccess) {
        hipLaunchKernelGGL(k_eval_math, dim3(gridFor(ctx, n)), dim3(256), 0, ctx->stream, kind, da, db, n, dout);
        e = hipStreamSynchronize(ctx->stream);
    }
    if (e == hipSuccess) e = hipMemcpy(out, dout, size_t(n)*4, hipMemcpyDeviceToHost);
    (void)hipFree(da); (void)hipFree(db); (void)hipFree(dout);
    if (e != hipSuccess) { fail(ctx, "tgpu_eval_math: %s", hipGetErrorString(e)); return TGPU_ERR_HIP; }
    return TGPU_OK;
}

} // extern "C"
