// Wavefront kernels of the MI355X path tracer (gfx950, wave64).
//
//   K1 generate   camera rays + sampler init            (PathTraceIntegrator.cpp:136-156, PinholeCamera.cpp:70-86)
//   K2 extend     closest hit, persistent waves, LDS stacks (TraceableScene.hpp:170-192 + Embree BVH4 -> own BVH2)
//   K3 shade      per BSDF class, sorted queues          (TraceBase.cpp:516-568, 246-321 ; PathTracer.cpp:50-127)
//   K4 shadow     any-hit + NEE resolve                  (TraceBase.cpp:62-125, 144-174)
//   K5 miss       environment lookup                     (TraceBase.cpp:570-578)
//   K7 accumulate per-pixel running mean                 (OutputBuffer.hpp:104-132)
//
// Path state lives in HBM as SoA float4/uint4 streams indexed by path SLOT of a fixed pool; queues
// are compact arrays of slot indices with device-side counters.  A slot whose path ended is
// retired (its radiance goes to the per-sample result ring) and immediately re-used for the next
// camera sample (path regeneration, K1), so every wavefront iteration runs on a full pool and the
// long low-occupancy tail of a pass is paid once per render instead of once per pass.  All kernels
// are launched with a fixed, chip-filling grid and stride/pull over their queue.
#pragma once
#include "device.h"

namespace tg {

// ---- path-state streams ---------------------------------------------------------------------
// Slot lists: two sets (ping-pong) of 2 + TGPU_BSDF_NUM_CLASSES lists.  List 0 holds the slots whose ray
// escaped (and freshly started / just-died slots), list 1 + c the slots whose hit has BSDF class c, the last
// list the slots whose ray runs inside a participating medium (distance sampling comes before the surface).
// The extend kernel reads the concatenation of the current set and writes the other set; every other
// kernel works in place on the set extend just wrote, so a slot keeps its list entry while its path is
// retired and the next sample is started in it.
#define TG_LISTS (3 + TGPU_BSDF_NUM_CLASSES)
#define TG_CLASS_OTHER TGPU_BSDF_NUM_CLASSES        /* BSDF types without a shade kernel of their own share the generic one */
#define TG_CLASS_VOLUME (TGPU_BSDF_NUM_CLASSES + 1) /* paths whose ray travels inside a participating medium (hit or miss) */
enum { CNT_LIST0 = 0 /* 2 x TG_LISTS list lengths */, CNT_HEAD_EXT = 2*TG_LISTS, CNT_HEAD_SHADOW = CNT_HEAD_EXT + 1,
       CNT_REGEN = CNT_HEAD_EXT + 2 /* samples claimed this iteration */,
       CNT_FREE = CNT_HEAD_EXT + 3 /* persistent: length of the free-slot list */,
       CNT_JOB_DONE0 = CNT_HEAD_EXT + 4 /* TG_RING persistent per-job retire counters */ };
// Result-ring slices (= jobs that may be in flight).  A job can only be folded into the framebuffer
// once its slowest path has ended (~max_bounces iterations after its last sample started), and during
// that time the pool keeps starting samples of later jobs: to never run dry the ring must hold ~26 x pool size samples.
// 32 slices of 16 spp do for a whole 1280x720 image (the benchmark: 30 jobs); a small shard (an 8-rank run: 1/8 of the pixels per job)
// gets by as long as everything it was asked for fits into the ring -- when it does not, the launcher finishes the jobs in flight
// with the tail kernel instead of draining the pool one iteration per bounce (tgpu_api.hip, driveIteration).  With 16 slices a 20-pass
// request stalled exactly like that: 116 instead of 52 ms for an 8-rank shard, 7 % on the whole image (profiles/r03_ab_notes.txt).
#define TG_RING 32
static_assert(TG_RING <= 32, "tgpu_ctx::ringFree is a 32-bit mask");
enum { CNT_TOTAL = CNT_JOB_DONE0 + TG_RING };
// Every counter lives on its own 128-byte line: the hot ones (dequeue heads, list lengths) are hit by
// atomics from every CU, and an L2 channel serialises the requests to one line (~88 atomics/us).
#define TG_CNT_STRIDE 32
#define CNT(i) ((i)*TG_CNT_STRIDE)

// Samples are handed out in global order: job by job, inside a job sample-major then shard pixel
// (coalesced accumulate).  A job = one spp range over all shard pixels (one tgpu_render_pass chunk).
struct JobRing {
    unsigned long long start[TG_RING]; // global index of the job's first sample
    uint32_t sppBegin[TG_RING];
    uint32_t sppCount[TG_RING];
    uint32_t ringSlot[TG_RING];        // which slice of the result ring the job writes
    uint32_t adaptive[TG_RING];        // 1: per-variance-tile sample counts (jobOffset/jobVtIndex of the ring slot)
    uint32_t n;
};

enum { END_ALIVE = 0, END_RAW = 1, END_ZERO = 2, END_CHECKED = 3, END_DEAD = 4 /* slot idle: no sample */,
       END_DEFERRED = 0x7F /* only in the end-state bytes between k_volume and k_volume_nested: the entry's surface step is still to run */ };
// flags word (bits of tp.w): [15:0] bounce, [16] wasSpecular, [27:24] end state
TG_DEV uint32_t packFlags(int bounce, bool wasSpecular, int end) { return uint32_t(bounce) | (wasSpecular ? 1u << 16 : 0u) | (uint32_t(end) << 24); }
TG_DEV int flagsBounce(uint32_t f) { return int(f & 0xFFFFu); }
TG_DEV bool flagsSpecular(uint32_t f) { return (f >> 16) & 1u; }
TG_DEV int flagsEnd(uint32_t f) { return int((f >> 24) & 0xFu); }
// state word (second half of the path record): bit 0 = NEE rays queued for the current vertex, [7:4] = end state (copy of tp.w's)
TG_DEV uint32_t packState(bool nee, int end) { return (nee ? 1u : 0u) | (uint32_t(end) << 4); }
TG_DEV bool stateNee(uint32_t s) { return (s & 1u) != 0; }
TG_DEV int stateEnd(uint32_t s) { return int((s >> 4) & 0xFu); }

// Path state lives in two 128-byte records per slot (array of structures): the lists hand the kernels slot numbers in
// an order that decorrelates from memory order after a few bounces, so every field access of a lane is a random access.
// With one array per field (16-byte elements) each of them costs a 64-byte HBM sector; with records the fields a
// kernel needs share one or two sectors.  Strided<> keeps the `st.field[slot]` spelling.
//   path record   : [0] rayO [1] rayD [2] tp [3] hit | [4] em [5] samp [6] med, hitInst, state, sampleRef [7] tpSum
//   shadow record : [0] shA  [1] shB  [2] shO [3] shX | [4] shC [5] shD [6] shE [7] shF
// The halves are 64-byte sectors and are laid out by consumer: extend reads only the path record's first half, the shadow
// traversal only the shadow record's first half (rays, origin, end cap), and retire + NEE resolve only the two second halves
// (state, radiance, sample reference | contributions, visibility bits) -- a slot costs each kernel one or two sectors.
template <class T, int STRIDE>
struct Strided {
    T *base;
    TG_DEV T &operator[](size_t i) const { return base[i*STRIDE]; }
};
typedef Strided<float4, 8> RecF4;
typedef Strided<uint4, 8> RecU4;
typedef Strided<uint32_t, 32> RecU1;
#define TG_REC_F4 8 /* float4 slots per record */

// One OutputBuffer<float | Vec3f> (cameras/OutputBuffer.hpp:21-27) of the shard: nPixels x channels floats per part
struct DOutBuf { float *a, *b, *var; uint32_t *count; };
enum { AOV_RECORDED = 1u, AOV_VIS_PENDING = 2u }; // results[ref].w while the path runs

struct DState {
    RecF4 rayO;     // origin.xyz, tnear
    RecF4 rayD;     // dir.xyz, tfar
    RecF4 hit;                  // t, u, v, bits(prim)
    // only with forward-scattering surfaces (sc.hasForward): the factors of the two NEE terms, so that the shadow kernel can
    // apply the transmittance where the reference does (e = shadow*E before f*e/pdf): shG = (E_L, pdf_L), shH = (mis_L, mis_B),
    // shI = (weight_B); shC then holds f_L and shD E_B
    float4 *shG, *shH, *shI;
    RecU1 med;                  // medium state: [7:0] medium + 1, [8] MediumState::firstScatter, [31:9] MediumState::bounce; only with media
    RecU1 hitInst;              // instance of the hit (set << 24 | index) or INST_NONE; only with "instances" in the scene
    RecF4 tp;       // throughput.xyz, bits(flags)
    RecF4 em;       // emission (radiance so far).xyz, bits(Sobol scramble)
    RecU4 samp;     // sobol dimension, pcg lo, pcg hi, sample index
    // NEE records of the path's current vertex (origin = rayO.xyz, tnear = epsilon)
    RecF4 shA;      // light-sample dir.xyz, tfar (< 0: no ray)
    RecF4 shB;      // bsdf-sample dir.xyz, tfar (< 0: no ray)
    RecF4 shO;      // origin of the NEE rays.xyz (info.p), tnear -- a copy, so that the shadow kernel leaves the path record alone
    RecF4 shX;      // bits(end cap prim), bits(1 when the vertex has a second ray, in shB), -, -
    RecF4 shC;      // L.xyz (unoccluded light-sample term), light-pick weight
    RecF4 shD;      // B.xyz (unoccluded bsdf-sample term), bits(visibility of the NEE rays: bit 0 the ray in shA, bit 1 the ray in shB; written by k_shadow_q)
    RecF4 shE;      // throughput at the vertex.xyz, bits(has post-add)
    RecF4 shF;      // emitter term to add after the NEE term.xyz
    RecU1 sampleRef;     // per slot: index into `results` of the sample the slot is tracing
    float4 *results;     // TG_RING x nPixels x resultSpp: final radiance of every sample (w: AOV_* bits while the path runs)
    // auxiliary output buffers (renderer "output_buffers"; null otherwise), one record per sample next to `results`:
    //   aov0 = (depth, normal.xyz), aov1 = (albedo.xyz, visibility); NaN = the sample adds nothing to that buffer
    // (OutputBuffer::addSample drops NaN anyway).  Until the path records its first non-specular vertex aov0.x carries hitDistance.
    float4 *aov0, *aov1;
    DOutBuf out[TGPU_OUTPUT_COUNT]; // the output buffers themselves (k_accumulate*); out[0].a / .count alias fbMean / fbCount
    uint32_t *lists[2][TG_LISTS];
    uint8_t *endFlag[2]; // per set: the end state (END_*) the shade / volume step left for every entry of the concatenated lists
    uint32_t *freeQ;     // idle slots that were dropped from the lists (no sample admitted / end of render)
    RecU1 state;         // per slot: bit 0 = NEE terms of the slot's last vertex still have to be folded into its radiance (shC..shF valid),
                         // bits [7:4] = a copy of the end state of tp.w.  (A compact 4-byte array instead was slower: a random access costs a
                         // line whatever its size, and the kernels that need the word read this sector anyway.)
    Strided<float, 32> tpSum; // per slot: throughput.sum() of a path that ended END_CHECKED (traceSample's last NaN test, done at retire)
    uint32_t *counters;
    uint32_t nPixels;
    uint32_t resultSlice; // samples per ring slot = nPixels * resultSpp
    unsigned long long *stats; // [0] ext nodes [1] ext tris [2] shadow nodes [3] shadow tris [4] shadow rays [5] surface events [6] extend rays traced by the tail kernel
    // shard
    const uint32_t *pixelIndex;
    const uint32_t *scramble;
    float *fbMean;
    uint32_t *fbCount;
    // adaptive sampling (settings.adaptive_sampling): SampleRecords of the shard's 4x4 variance tiles
    uint32_t nVt;
    const uint32_t *vtPix;      // nVt x 16 shard pixel ids in the order renderTile visits them (row-major), 0xFFFFFFFF = outside
    const uint32_t *vtOfPixel;  // shard pixel -> local variance tile
    uint32_t *recCount;         // SampleRecord::sampleCount
    float *recMean, *recVar;    // SampleRecord::mean / runningVariance
    uint32_t *jobOffset;        // TG_RING x (nPixels + 1): first sample of each pixel inside an adaptive job
    uint32_t *jobVtIndex;       // TG_RING x nVt: SampleRecord::sampleIndex of the pass
};

// Accesses of the traversal kernels to the path state.  Until round 3 they were non-temporal ("keep the 4 MB L2 of each XCD for the BVH"): that
// dates from the one-array-per-field layout, where every access had a line of its own.  With records (section 4 of DESIGN.md) a kernel reads
// three or four sectors of ONE line, and a non-temporal load does not leave the line behind for the next sector's load: plain accesses are
// 5-6 % faster in both traversal kernels (profiles/r03_ab_notes.txt; in the shade step the hint costs a factor of three).
typedef float tg_f4 __attribute__((ext_vector_type(4)));
#ifndef TG_STATE_NT
#define TG_STATE_NT 0 /* flat scenes: bit 0: loads, bit 1: stores non-temporal (A/B switch) */
#endif
// Scenes with "instances" keep the hint in their traversal kernels: their working set (top-level tree, instance entries, the masters' trees) is what the
// L2 is for, and the plain form costs k_shadow_iq 7.5 % and k_extend<general> 1.2 % on config 4 (profiles/r03_ab_notes.txt).
#ifndef TG_STATE_NT_INST
#define TG_STATE_NT_INST 3
#endif
template <int MODE = TG_STATE_NT>
TG_DEV float4 ldStream(const float4 *p)
{
    if (MODE & 1) {
        tg_f4 v = __builtin_nontemporal_load(reinterpret_cast<const tg_f4 *>(p));
        return make_float4(v.x, v.y, v.z, v.w);
    }
    return *p;
}
template <int MODE = TG_STATE_NT>
TG_DEV uint32_t ldStream(const uint32_t *p)
{
    if (MODE & 1) return __builtin_nontemporal_load(p);
    return *p;
}
template <int MODE = TG_STATE_NT>
TG_DEV void stStream(float4 *p, const float4 &v)
{
    if (MODE & 2) {
        tg_f4 t = {v.x, v.y, v.z, v.w};
        __builtin_nontemporal_store(t, reinterpret_cast<tg_f4 *>(p));
    } else {
        *p = v;
    }
}
template <int MODE = TG_STATE_NT>
TG_DEV void stStream(uint32_t *p, uint32_t v)
{
    if (MODE & 2) __builtin_nontemporal_store(v, p);
    else *p = v;
}
TG_DEV float4 mk4(const V3 &v, float w) { return make_float4(v.x, v.y, v.z, w); }
TG_DEV float4 mk4u(const V3 &v, uint32_t w) { return make_float4(v.x, v.y, v.z, __uint_as_float(w)); }
TG_DEV V3 xyz(const float4 &f) { return V3(f.x, f.y, f.z); }

// wave-aggregated append of `path` to queue[counterIdx] for the lanes where pred holds
TG_DEV void queueAppend(bool pred, uint32_t *queue, uint32_t *counter, uint32_t value)
{
    unsigned long long mask = __ballot(pred);
    if (mask == 0) return;
    int lane = threadIdx.x & 63;
    int leader = __builtin_ctzll(mask);
    uint32_t base = 0;
    if (lane == leader) base = atomicAdd(counter, uint32_t(__builtin_popcountll(mask)));
    base = __shfl(base, leader);
    if (pred) {
        uint32_t off = uint32_t(__builtin_popcountll(mask & ((1ull << lane) - 1ull)));
        queue[base + off] = value;
    }
}

// List lengths of a set, read once per kernel (never per lane: the counter lines are hot)
struct ListCounts { uint32_t n[TG_LISTS]; };
TG_DEV ListCounts loadListCounts(const DState &st, int set)
{
    ListCounts c;
#pragma unroll
    for (int l = 0; l < TG_LISTS; ++l) c.n[l] = st.counters[CNT(CNT_LIST0 + set*TG_LISTS + l)];
    return c;
}
// Entry `idx` of the concatenation of lists [first, TG_LISTS) of `set`; false when idx is past the end.
TG_DEV bool listFetch(const DState &st, const ListCounts &lc, int set, int first, uint32_t idx, uint32_t &slot)
{
#pragma unroll
    for (int l = 0; l < TG_LISTS; ++l) {
        if (l < first) continue;
        uint32_t n = lc.n[l];
        if (idx < n) {
            slot = ldStream(&st.lists[set][l][idx]);
            return true;
        }
        idx -= n;
    }
    return false;
}
// position of list `l`'s first entry in the concatenation of the set's lists
TG_DEV uint32_t listBase(const DState &st, int set, int l)
{
    uint32_t n = 0;
    for (int j = 0; j < l; ++j) n += st.counters[CNT(CNT_LIST0 + set*TG_LISTS + j)];
    return n;
}
TG_DEV uint32_t listTotal(const ListCounts &lc, int first)
{
    uint32_t n = 0;
#pragma unroll
    for (int l = 0; l < TG_LISTS; ++l)
        if (l >= first) n += lc.n[l];
    return n;
}

// The NEE terms of a vertex whose shadow rays went through k_shadow_q are folded into the path's radiance by whoever touches
// the path next (the next vertex's shade / volume step, the environment lookup of an escaped path, the retire step of an
// ended one, the tail kernel) -- all of them read these records anyway -- in the reference's order
//   emission += ((0 + L*vis_L) + B*vis_B)*w*throughput   (TraceBase.cpp:383-400,483-494,537-538), then the emitter term (:540-543)
// 'visibility' output buffer: the vertex that recorded the sample's auxiliary values left the transmittance of its light sample
// as if the shadow ray were clear (AOV_VIS_PENDING); here the shadow ray's result is in.  shadow: the walked transmittance of
// scenes with forward-scattering surfaces (TraceBase.cpp:167-169), else null and `clear` says whether the ray was unoccluded.
// (out of line with plain values for arguments: nothing of the caller's private state is passed by address)
static __device__ __noinline__ void aovVisibilityCall(float4 *results, float4 *aov1, uint32_t ref, bool clear, bool walked, float walkedAverage)
{
    float *w = reinterpret_cast<float *>(&results[ref]) + 3;
    if (!(__float_as_uint(*w) & AOV_VIS_PENDING)) return;
    if (walked) reinterpret_cast<float *>(&aov1[ref])[3] = walkedAverage;
    else if (!clear) reinterpret_cast<float *>(&aov1[ref])[3] = 0.0f;
    *w = __uint_as_float(AOV_RECORDED);
}
TG_DEV void aovVisibility(const DState &st, uint32_t slot, bool clear, bool walked, const V3 &shadow)
{
    aovVisibilityCall(st.results, st.aov1, st.sampleRef[slot], clear, walked, walked ? shadow.avg() : 0.0f);
}

// AOV = false: the lean shade kernels, which never run with auxiliary output buffers (the launcher picks the general ones then)
template <bool AOV = true>
TG_DEV V3 resolveNeeTerms(const DState &st, uint32_t slot, V3 em)
{
    // (plain loads: the non-temporal form of this triple trips an 'even aligned vector registers' backend error in the general shade kernels)
    const float4 c = st.shC[slot], d = st.shD[slot], e = st.shE[slot];
    const uint32_t vis = __float_as_uint(d.w);
    if (AOV && st.aov0) aovVisibility(st, slot, (vis & 1u) != 0, false, V3(0.0f));
    // an occluded term enters as +0: x + 0 == x bit for bit, so this is ((0 + L*vis_L) + B*vis_B) without branches
    const V3 lTerm = (vis & 1u) ? xyz(c) : V3(0.0f), bTerm = (vis & 2u) ? xyz(d) : V3(0.0f);
    V3 result = (V3(0.0f) + lTerm) + bTerm;
    em += (result*c.w)*xyz(e);
    if (__float_as_uint(e.w) & 1u) em += xyz(st.shF[slot]);
    return em;
}

// =================================================================================================
// K1 generate
// =================================================================================================
TG_DEV void cameraRay(const tgpu_camera &cam, uint32_t px, uint32_t py, float xi0, float xi1, V3 &pos, V3 &dir)
{   // cameras/PinholeCamera.cpp:70-86 ; ReconstructionFilter::sample (ReconstructionFilter.hpp:86-103,152-169)
    float uvx, uvy;
    if (cam.filter_type == TGPU_FILTER_DIRAC) {
        uvx = uvy = 0.0f;
    } else if (cam.filter_type == TGPU_FILTER_BOX) {
        uvx = xi0 - 0.5f;
        uvy = xi1 - 0.5f;
    } else {
        float uv[2];
#pragma unroll
        for (int a = 0; a < 2; ++a) {
            float xi = a == 0 ? xi0 : xi1;
            bool negative = xi < 0.5f;
            xi = negative ? xi*2.0f : (xi - 0.5f)*2.0f;
            int idx = 30;
            for (int i = 0; i < 30; ++i) {
                if (xi < cam.filter_cdf[i]) {
                    idx = i;
                    break;
                }
            }
            float pdf = cam.filter_cdf[idx] - cam.filter_cdf[idx - 1];
            float u = cam.filter_bin_size*(idx + (xi - cam.filter_cdf[idx - 1])/pdf);
            uv[a] = negative ? -u : u;
        }
        uvx = uv[0];
        uvy = uv[1];
    }
    V3 localD = V3(-1.0f + (float(px) + 0.5f + uvx)*2.0f*cam.pixel_size_x,
                   cam.ratio - (float(py) + 0.5f + uvy)*2.0f*cam.pixel_size_x,
                   cam.plane_dist).normalized();
    pos = V3(cam.pos);
    dir = mulVector(cam.rot, localD);
}

// Start sample g of the render in path slot `slot`: sampler + camera ray
// (PathTraceIntegrator.cpp:143-147, PathTracer.cpp:23-31).
TG_DEV void startSample(const DScene &sc, const DState &st, const JobRing &jobs, unsigned long long g, uint32_t slot)
{
    uint32_t j = jobs.n ? jobs.n - 1u : 0u; // start[] increases and new samples belong to the latest jobs: search from the end
    while (j > 0u && g < jobs.start[j]) --j;
    uint32_t e = uint32_t(g - jobs.start[j]);
    uint32_t s, k, sampleIndex;
    if (jobs.adaptive[j]) {
        // pixel-major job: pixel k owns samples [offset[k], offset[k+1]); find k = last pixel with offset[k] <= e
        const uint32_t *off = st.jobOffset + size_t(jobs.ringSlot[j])*(st.nPixels + 1u);
        uint32_t lo = 0, hi = st.nPixels; // invariant: off[lo] <= e < off[hi]
        while (hi - lo > 1u) {
            uint32_t mid = (lo + hi) >> 1;
            if (off[mid] <= e) lo = mid; else hi = mid;
        }
        k = lo;
        s = e - off[k];
        sampleIndex = st.jobVtIndex[size_t(jobs.ringSlot[j])*st.nVt + st.vtOfPixel[k]] + s;
    } else {
        s = e/st.nPixels;
        k = e - s*st.nPixels;
        sampleIndex = jobs.sppBegin[j] + s;
    }
    uint32_t pixel = st.pixelIndex[k];
    Sampler smp;
    smp.matrices = sc.sobol;
    smp.sobol = sc.settings.use_sobol;
    smp.scramble = st.scramble[k];
    smp.index = sampleIndex;
    smp.dimension = 0;
    smp.seedPathLocal(smp.scramble, smp.index);
    float xi0 = smp.next1D(), xi1 = smp.next1D();
    V3 pos, dir;
    cameraRay(sc.cam, pixel % sc.cam.res_x, pixel/sc.cam.res_x, xi0, xi1, pos, dir);
    st.rayO[slot] = mk4(pos, 1e-4f); // math/Ray.hpp:24
    st.rayD[slot] = mk4(dir, TG_INF);
    st.tp[slot] = mk4u(V3(1.0f), packFlags(0, true, END_ALIVE));
    st.em[slot] = mk4u(V3(0.0f), smp.scramble);
    st.samp[slot] = make_uint4(smp.dimension, uint32_t(smp.pcg), uint32_t(smp.pcg >> 32), smp.index);
    const uint32_t ref = jobs.ringSlot[j]*st.resultSlice + e;
    st.sampleRef[slot] = ref;
    if (st.aov0) { // recordedOutputValues = false; hitDistance = 0 (PathTracer.cpp:41-42)
        const float nan = __uint_as_float(0x7FC00000u);
        st.aov0[ref] = make_float4(0.0f, nan, nan, nan);
        st.aov1[ref] = make_float4(nan, nan, nan, nan);
        reinterpret_cast<float *>(&st.results[ref])[3] = 0.0f;
    }
    st.state[slot] = packState(false, END_ALIVE);
    if (sc.nMedia) st.med[slot] = uint32_t(sc.cam.medium + 1) | (1u << 8); // cam().medium(), MediumState::reset()
}

// Retire + regenerate IN PLACE over every entry of the set extend just wrote:
//  - alive paths are left alone;
//  - an ended path is retired: its radiance, with traceSample's exit rules applied (PathTracer.cpp:119-131),
//    goes to the result ring and the job's completion counter is bumped;
//  - ended and idle slots then claim the next samples; when one is admitted (< limit) the
//    slot starts it and simply stays in its list (the next extend traces its camera ray), otherwise the
//    slot is marked idle and the next extend drops it onto the free list.
// Claims: ONE global atomic per block for everything it will ask for (counted first from the list lengths and the end-state bytes), job
// counters per block in LDS (flushed once at the end): a single hot counter word only sustains ~88 atomics/us on this chip.
TG_DEV void missPath(const DScene &sc, const DState &st, uint32_t path);
TG_DEV int missPathEx(const DScene &sc, const DState &st, uint32_t path, V3 &emOut, float &tpSumOut);
TG_DEV void aovFinish(const DScene &sc, const DState &st, uint32_t path, uint32_t ref, bool epilogue);

// Per shaded entry this kernel reads one byte of the end-state array the shade / volume step wrote next to the lists (coalesced); only
// ended paths cost record lines (radiance, pending NEE terms, sample reference), escaped ones the whole path record (environment lookup).
#ifdef TG_TU_MAIN
#define TG_RETIRE_K 1
#ifndef TG_RETIRE_BLOCK
#define TG_RETIRE_BLOCK 256
#endif
#ifndef TG_RETIRE_WAVES
#define TG_RETIRE_WAVES 5 /* waves per SIMD the retire kernel is compiled for (A/B: profiles/r02_ab_notes.txt) */
#endif
static __global__ void __launch_bounds__(TG_RETIRE_BLOCK, TG_RETIRE_WAVES) k_retire_regenerate(DScene sc, DState st, JobRing jobs, int set, unsigned long long sampleBase,
                                                            unsigned long long limit)
{
    __shared__ uint32_t sWave[TG_RETIRE_BLOCK/64];
    __shared__ uint32_t sHist[TG_RING];
    __shared__ uint32_t sBase;
    __shared__ uint2 sQueue[TG_RETIRE_BLOCK/64][128]; // per wave: (list entry, rank of its claim) of the entries that need work
    const ListCounts lc = loadListCounts(st, set);
    const uint32_t total = listTotal(lc, 0);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const unsigned long long below = (1ull << lane) - 1ull;
    const uint8_t *endFlag = st.endFlag[set];
    // a wave owns every (waves in the grid)-th group of 64 entries: escaped paths (list 0, the expensive entries) spread over all waves
    const uint32_t first = (blockIdx.x*(blockDim.x >> 6) + uint32_t(wave))*64u, stride = gridDim.x*blockDim.x;
    if (threadIdx.x < TG_RING) sHist[threadIdx.x] = 0;
    // pass 1: how many samples this block will ask for -- every entry of list 0 and every entry whose path ended -- from the list lengths
    // and the end-state bytes alone; ONE global claim per block (a claim per block round cost a dependent atomic and two barriers each)
    uint32_t mine = 0;
    for (uint32_t q0 = first; q0 < total; q0 += stride) {
        const uint32_t q = q0 + uint32_t(lane);
        const bool need = q < total && (q < lc.n[0] || endFlag[q] != END_ALIVE);
        mine += uint32_t(__builtin_popcountll(__ballot(need)));
    }
    if (lane == 0) sWave[wave] = mine;
    __syncthreads();
    if (threadIdx.x == 0) {
        uint32_t sum = 0;
        for (uint32_t w = 0; w < (blockDim.x >> 6); ++w) { uint32_t n = sWave[w]; sWave[w] = sum; sum += n; }
        sBase = sum ? atomicAdd(&st.counters[CNT(CNT_REGEN)], sum) : 0u;
    }
    __syncthreads();
    const unsigned long long waveBase = sampleBase + sBase + sWave[wave]; // the wave's claims, in the order pass 2 meets them
    // pass 2.  Only the entries that need work (escaped or ended paths: about a third of the lists) cost more than their end-state byte, so the
    // wave first packs them -- (entry, rank of its claim) -- into a small LDS queue and then works on 64 of them at a time with every lane busy:
    // twice the loads in flight per wave of a kernel that waits on memory for two thirds of its cycles (profiles/r02_sq_summary.txt).
    uint2 *queue = sQueue[wave];
    uint32_t queued = 0, claimed = 0; // wave-uniform
    auto work = [&](uint32_t count) { // the first `count` (<= 64) queue entries, one per lane
        uint32_t slot = 0, ring = 0xFFFFFFFFu;
        bool needs = false;
        unsigned long long g = 0;
        if (uint32_t(lane) < count) {
            const uint2 e = queue[lane];
            const uint32_t q = e.x;
            g = waveBase + e.y;
            if (listFetch(st, lc, set, 0, q, slot)) {
                int end;
                V3 c(0.0f);
                float tpSum = 0.0f;
                if (q < lc.n[0]) {
                    // entries of list 0 are the paths that left the scene: their environment lookup (K5 miss) happens here, so
                    // that their record is read and written once, not by a miss kernel first and by this one again
                    end = missPathEx(sc, st, slot, c, tpSum);
                } else {
                    end = int(endFlag[q]); // END_* of the entry, written by its shade / volume step
                    if (end != END_DEAD) { // (END_ALIVE entries are not queued: their NEE terms ride with their next vertex)
                        c = xyz(st.em[slot]);
                        if (stateNee(st.state[slot])) c = resolveNeeTerms(st, slot, c);
                        if (end == END_CHECKED) tpSum = st.tpSum[slot];
                    }
                }
                if (end != END_ALIVE) {
                    needs = true;
                    if (end != END_DEAD) {
                        const bool nanEnd = end == END_CHECKED && isnanf_(tpSum + c.sum()); // PathTracer.cpp:130-131
                        if (end == END_ZERO || nanEnd) c = V3(0.0f);
                        uint32_t ref = st.sampleRef[slot];
                        if (st.aov0) aovFinish(sc, st, slot, ref, end == END_CHECKED && !nanEnd);
                        st.results[ref] = mk4(c, 0.0f);
                        ring = ref/st.resultSlice;
                    }
                }
            }
        }
        unsigned long long todo = __ballot(ring != 0xFFFFFFFFu);
        while (todo) { // one LDS atomic per distinct job present in the wave
            int leader = __builtin_ctzll(todo);
            uint32_t r = __shfl(ring, leader);
            unsigned long long m = __ballot(ring == r);
            if (lane == leader) atomicAdd(&sHist[r], uint32_t(__builtin_popcountll(m)));
            todo &= ~m;
        }
        if (needs) {
            if (g < limit) {
                startSample(sc, st, jobs, g, slot);
            } else { // idle: the next extend drops the slot onto the free list (it reads tp.w)
                reinterpret_cast<uint32_t *>(&st.tp[slot])[3] = packFlags(0, false, END_DEAD);
                st.state[slot] = packState(false, END_DEAD);
            }
        }
        // close the gap: what is left moves to the front of the queue
        uint2 rest = make_uint2(0u, 0u);
        const bool moves = count + uint32_t(lane) < queued;
        if (moves) rest = queue[count + uint32_t(lane)];
        if (moves) queue[lane] = rest;
        queued -= count;
    };
    for (uint32_t q0 = first; q0 < total; q0 += stride) {
        const uint32_t q = q0 + uint32_t(lane);
        const bool need = q < total && (q < lc.n[0] || endFlag[q] != END_ALIVE);
        const unsigned long long m = __ballot(need);
        const uint32_t rank = uint32_t(__builtin_popcountll(m & below));
        if (need) queue[queued + rank] = make_uint2(q, claimed + rank); // the claim order of pass 1: by entry
        const uint32_t n = uint32_t(__builtin_popcountll(m));
        queued += n;
        claimed += n;
        if (queued >= 64u) work(64u);
    }
    if (queued) work(queued);
    __syncthreads();
    if (threadIdx.x < TG_RING && sHist[threadIdx.x]) atomicAdd(&st.counters[CNT(CNT_JOB_DONE0 + threadIdx.x)], sHist[threadIdx.x]);
}
#endif

// Refill from the free list: the host knows how many samples are admitted, so slot freeQ[freeBase + q]
// takes sample sampleBase + q and is appended to list 0 of the current set at listBase + q.
#ifdef TG_TU_MAIN
static __global__ void __launch_bounds__(256) k_refill(DScene sc, DState st, JobRing jobs, int set, unsigned long long sampleBase, uint32_t nStart,
                                                uint32_t freeBase, uint32_t listBase)
{
    for (uint32_t q = blockIdx.x*blockDim.x + threadIdx.x; q < nStart; q += gridDim.x*blockDim.x) {
        uint32_t slot = st.freeQ[freeBase + q];
        startSample(sc, st, jobs, sampleBase + q, slot);
        st.lists[set][0][listBase + q] = slot;
    }
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        st.counters[CNT(CNT_FREE)] = freeBase;
        st.counters[CNT(CNT_LIST0 + set*TG_LISTS)] = listBase + nStart;
    }
}
#endif

// =================================================================================================
// BVH traversal (K2 / K4 core)
// =================================================================================================
struct TraceCounters { uint32_t nodes, tris; };

// Embree MoellerTrumboreIntersector1 (thirdparty/embree/kernels/geometry/triangle_intersector_moeller.h:76-113)
// with IEEE division in finalize(); dot = a.x*b.x + (a.y*b.y + a.z*b.z) (common/math/vec3.h:182)
TG_DEV float edot(const V3 &a, const V3 &b) { return a.x*b.x + (a.y*b.y + a.z*b.z); }

// include/tgpu_bounds.h (3): a hit counts only when the point it reports lies inside the triangle's own box grown by hitPad -- a grazing ray's
// t = T/den can be far off, and a walk only finds what lies in the boxes the ray really enters.  Off the hot path on purpose: the traversal loops are
// register-bound (k_shadow_q: 72 VGPRs for 7 waves; inlined, this cost it 16 B more scratch and 15 % of its time), and it runs only for a candidate
// that passed every other test AND grazes (triTest below), i.e. for about one hit in a thousand.
static __device__ __noinline__ bool hitInsideOwnBox(float v0x, float v0y, float v0z, float e1x, float e1y, float e1z, float e2x, float e2y, float e2z,
                                                    float px, float py, float pz, float hitPad)
{
    const float ax = v0x - e1x, ay = v0y - e1y, az = v0z - e1z, bx = v0x + e2x, by = v0y + e2y, bz = v0z + e2z;
    return px >= fmin2(v0x, fmin2(ax, bx)) - hitPad && px <= fmax2(v0x, fmax2(ax, bx)) + hitPad &&
           py >= fmin2(v0y, fmin2(ay, by)) - hitPad && py <= fmax2(v0y, fmax2(ay, by)) + hitPad &&
           pz >= fmin2(v0z, fmin2(az, bz)) - hitPad && pz <= fmax2(v0z, fmax2(az, bz)) + hitPad;
}

// Embree's MoellerTrumboreIntersector1 on the precomputed record (v0, e1 = v0 - v1, e2 = v2 - v0, Ng = e1 x e2), plus rule (3) above.
// ngSq = |Ng|^2 from the record (t3.y).  Fast accept: a ray at least TGPU_HIT_SURE_COS off the triangle's plane reports a point within hitPad of the
// true one whatever the scene (include/tgpu_bounds.h), so only grazing candidates pay for the box test.
TG_DEV bool triTest(const float4 &t0, const float4 &t1, const float4 &t2, float ngSq, const V3 &O, const V3 &D, float tnear, float tfar0, float hitPad,
                    float &t, float &u, float &v)
{
    V3 v0(t0.x, t0.y, t0.z), e1(t0.w, t1.x, t1.y), e2(t1.z, t1.w, t2.x), Ng(t2.y, t2.z, t2.w);
    V3 C = v0 - O;
    V3 R = D.cross(C);
    float den = edot(Ng, D);
    float absDen = __builtin_fabsf(den);
    uint32_t sgn = __float_as_uint(den) & 0x80000000u;
    float U = __uint_as_float(__float_as_uint(edot(R, e2)) ^ sgn);
    float V = __uint_as_float(__float_as_uint(edot(R, e1)) ^ sgn);
    if (!(den != 0.0f && U >= 0.0f && V >= 0.0f && U + V <= absDen)) return false;
    float T = __uint_as_float(__float_as_uint(edot(Ng, C)) ^ sgn);
    if (!(T > absDen*tnear && T < absDen*tfar0)) return false;
    t = T/absDen;
    u = U/absDen;
    v = V/absDen;
    if (absDen*absDen >= (TGPU_HIT_SURE_COS*TGPU_HIT_SURE_COS)*(ngSq*edot(D, D))) return true;
    return hitInsideOwnBox(v0.x, v0.y, v0.z, e1.x, e1.y, e1.z, e2.x, e2.y, e2.z, O.x + D.x*t, O.y + D.y*t, O.z + D.z*t, hitPad);
}

// waves per SIMD the lean traversal kernels are compiled for (second __launch_bounds__ argument: 7 -> at most 72 VGPRs; A/B in profiles/r01_ab_notes.txt)
#ifndef TG_EXTEND_WAVES
#define TG_EXTEND_WAVES 7
#endif
#ifndef TG_SHADOW_WAVES
#define TG_SHADOW_WAVES 7
#endif

// Traversal stack: the first sc.stackLds levels live in LDS (stack[level][thread], bank = lane), deeper ones -- the
// worst case is 3 pushes per tree level, real rays stay far below it -- in a per-thread global spill area, so that
// the LDS footprint (and with it the number of resident waves) does not grow with the depth of the tree.
TG_DEV void stackPush(const DScene &sc, uint32_t *stack, int &sp, uint32_t value)
{
    if (sp < sc.stackLds) stack[sp*256] = value;
    else sc.stackSpill[size_t(sp - sc.stackLds)*sc.spillStride + blockIdx.x*256u + threadIdx.x] = value;
    ++sp;
}
TG_DEV uint32_t stackPop(const DScene &sc, uint32_t *stack, int &sp)
{
    --sp;
    if (sp < sc.stackLds) return stack[sp*256];
    return sc.stackSpill[size_t(sp - sc.stackLds)*sc.spillStride + blockIdx.x*256u + threadIdx.x];
}

// Traversal of the quantised BVH4 (bvh_build.h).
// Closest (ANY = false): lexicographic minimum of (t, soup triangle id) over all valid triangles.
// Any-hit (ANY = true): returns true at the first valid triangle.
// One ray's traversal state; travStep advances it by one BVH item.
struct TravState {
    V3 O, D, inv, noi;
    float tnear, tfar0, bestT, bestU, bestV;
    uint32_t bestPrim, node;
    int sp;
    bool nx, ny, nz;
};
enum { TRAV_MORE = 0, TRAV_DONE = 1, TRAV_ANYHIT = 2 };

TG_DEV void travInit(TravState &r, uint32_t root, const V3 &O, const V3 &D, float tnear, float tfar, uint32_t bestPrim)
{
    r.O = O; r.D = D;
    r.tnear = tnear; r.tfar0 = tfar; r.bestT = tfar; r.bestU = 0.0f; r.bestV = 0.0f; r.bestPrim = bestPrim;
    // keep 1/d finite so that 0*inf never appears in the slab test
    V3 ds(__builtin_fabsf(D.x) < 1e-20f ? __builtin_copysignf(1e-20f, D.x) : D.x,
          __builtin_fabsf(D.y) < 1e-20f ? __builtin_copysignf(1e-20f, D.y) : D.y,
          __builtin_fabsf(D.z) < 1e-20f ? __builtin_copysignf(1e-20f, D.z) : D.z);
    r.inv = V3(1.0f/ds.x, 1.0f/ds.y, 1.0f/ds.z);
    r.noi = V3(-O.x*r.inv.x, -O.y*r.inv.y, -O.z*r.inv.z);
    r.sp = 0;
    r.node = root;
    // per-axis choice of the near/far byte plane (bvh_build.h, Bvh4): for a negative direction the near
    // plane of the box is HI
    r.nx = r.inv.x < 0.0f; r.ny = r.inv.y < 0.0f; r.nz = r.inv.z < 0.0f;
}

// One BVH item per step and lane -- an inner node or ONE triangle of a leaf, both 64-byte records -- fetched by
// the same four loads at the top: every step costs a single memory round trip whichever mix of nodes and
// triangles the lanes hold (a per-leaf triangle loop cost the wave one round trip per triangle of its largest leaf).
// Closest (ANY = false): lexicographic minimum of (t, soup triangle id) over all valid triangles.
// SORTED = false (any-hit only): the children that are hit are visited in slot order -- no keys, no sorting network; the answer
// of an any-hit query does not depend on the order.
// The 64-byte record the next step of `r` needs (an inner node or one triangle of a leaf)
TG_DEV const float4 *travItem(const DScene &sc, const TravState &r)
{
    return (r.node & 0x80000000u) ? sc.tris + size_t((r.node & 0x7FFFFFFFu) >> 3)*4 : sc.nodes + size_t(r.node)*4;
}
// travApply: one step on the record q0..q3 fetched from travItem(); travStep = fetch + apply.  ANYDYN: ANY is decided per lane
// at run time (`any`) -- the instance state machine steps closest-hit and any-hit queries through one loop.
template <bool ANY, bool COUNT, bool SORTED = true, bool ANYDYN = false>
TG_DEV int travApply(const DScene &sc, uint32_t *stack, TravState &r, TraceCounters &tc, const float4 &q0, const float4 &q1, const float4 &q2, const float4 &q3,
                     bool any = ANY)
{
    static_assert(ANY || SORTED, "closest-hit traversal needs the near-to-far order");
    const int maxSp = sc.stackEntries;
    const V3 &O = r.O, &D = r.D, &inv = r.inv, &noi = r.noi;
    const float tnear = r.tnear, tfar0 = r.tfar0;
    const bool nx = r.nx, ny = r.ny, nz = r.nz;
    uint32_t &node = r.node;
    int &sp = r.sp;
    float &bestT = r.bestT, &bestU = r.bestU, &bestV = r.bestV;
    uint32_t &bestPrim = r.bestPrim;
    {
        const bool leaf = (node & 0x80000000u) != 0;
        const uint32_t payload = node & 0x7FFFFFFFu; // leaf: first << 3 | count - 1
        if (!leaf) {
            if (COUNT) tc.nodes++;
            // the interval the boxes are clipped to, with the relative slack of include/tgpu_bounds.h: the computed t of a
            // triangle may lie a few ulps outside its box's [entry, exit]
            const float tLo = tnear*(1.0f - TGPU_BVH_T_SLACK), tHi = bestT*(1.0f + TGPU_BVH_T_SLACK);
            const uint32_t eb = __float_as_uint(q0.w);
            // t = q*(scale*inv) + (origin - O)*inv, one fma per plane
            const float ax = __uint_as_float((eb & 0xFFu) << 23)*inv.x, ay = __uint_as_float(((eb >> 8) & 0xFFu) << 23)*inv.y,
                        az = __uint_as_float(((eb >> 16) & 0xFFu) << 23)*inv.z;
            const float bx = TGPU_FMA(q0.x, inv.x, noi.x), by = TGPU_FMA(q0.y, inv.y, noi.y), bz = TGPU_FMA(q0.z, inv.z, noi.z);
            const uint32_t LOx = __float_as_uint(q1.x), LOy = __float_as_uint(q1.y), LOz = __float_as_uint(q1.z);
            const uint32_t HIx = __float_as_uint(q1.w), HIy = __float_as_uint(q2.x), HIz = __float_as_uint(q2.y);
            const uint32_t Nx = nx ? HIx : LOx, Fx = nx ? LOx : HIx, Ny = ny ? HIy : LOy, Fy = ny ? LOy : HIy, Nz = nz ? HIz : LOz, Fz = nz ? LOz : HIz;
            uint32_t key[4];
            if (!SORTED) {
                const uint32_t cr[4] = {__float_as_uint(q3.x), __float_as_uint(q3.y), __float_as_uint(q3.z), __float_as_uint(q3.w)};
                uint32_t first = 0xFFFFFFFFu;
#pragma unroll
                for (int c = 3; c >= 0; --c) {
                    const int sh = 8*c;
                    float tn = TGPU_FMA(float((Nx >> sh) & 0xFFu), ax, bx);
                    tn = fmaxf(tn, TGPU_FMA(float((Ny >> sh) & 0xFFu), ay, by));
                    tn = fmaxf(tn, TGPU_FMA(float((Nz >> sh) & 0xFFu), az, bz));
                    tn = fmaxf(tn, tLo);
                    float tf = TGPU_FMA(float((Fx >> sh) & 0xFFu), ax, bx);
                    tf = fminf(tf, TGPU_FMA(float((Fy >> sh) & 0xFFu), ay, by));
                    tf = fminf(tf, TGPU_FMA(float((Fz >> sh) & 0xFFu), az, bz));
                    tf = fminf(tf, tHi);
                    if (tn <= tf) {
                        if (first != 0xFFFFFFFFu && sp < maxSp) stackPush(sc, stack, sp, first);
                        first = cr[c];
                    }
                }
                if (first != 0xFFFFFFFFu) {
                    node = first;
                    return TRAV_MORE;
                }
                if (sp == 0) return TRAV_DONE;
                node = stackPop(sc, stack, sp);
                return TRAV_MORE;
            }
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                const int sh = 8*c;
                float tn = TGPU_FMA(float((Nx >> sh) & 0xFFu), ax, bx);
                tn = fmaxf(tn, TGPU_FMA(float((Ny >> sh) & 0xFFu), ay, by));
                tn = fmaxf(tn, TGPU_FMA(float((Nz >> sh) & 0xFFu), az, bz));
                tn = fmaxf(tn, tLo);
                float tf = TGPU_FMA(float((Fx >> sh) & 0xFFu), ax, bx);
                tf = fminf(tf, TGPU_FMA(float((Fy >> sh) & 0xFFu), ay, by));
                tf = fminf(tf, TGPU_FMA(float((Fz >> sh) & 0xFFu), az, bz));
                tf = fminf(tf, tHi);
                // tn >= tLo >= 0: its bit pattern orders like the float; the low 2 bits carry the slot
                key[c] = tn <= tf ? ((__float_as_uint(tn) & ~3u) | uint32_t(c)) : 0xFFFFFFFFu;
            }
            // sort the four keys ascending (5-comparator network)
#define TG_CSWAP(i, j) { uint32_t lo_ = min(key[i], key[j]), hi_ = max(key[i], key[j]); key[i] = lo_; key[j] = hi_; }
            TG_CSWAP(0, 1) TG_CSWAP(2, 3) TG_CSWAP(0, 2) TG_CSWAP(1, 3) TG_CSWAP(1, 2)
#undef TG_CSWAP
            const uint32_t c0 = __float_as_uint(q3.x), c1 = __float_as_uint(q3.y), c2 = __float_as_uint(q3.z), c3 = __float_as_uint(q3.w);
#define TG_CHILD(k) (((k) & 3u) == 0u ? c0 : (((k) & 3u) == 1u ? c1 : (((k) & 3u) == 2u ? c2 : c3)))
            if (key[0] != 0xFFFFFFFFu) {
                // push the farther hits, farthest first, so that the nearest of them pops first
                if (key[3] != 0xFFFFFFFFu && sp < maxSp) stackPush(sc, stack, sp, TG_CHILD(key[3]));
                if (key[2] != 0xFFFFFFFFu && sp < maxSp) stackPush(sc, stack, sp, TG_CHILD(key[2]));
                if (key[1] != 0xFFFFFFFFu && sp < maxSp) stackPush(sc, stack, sp, TG_CHILD(key[1]));
                node = TG_CHILD(key[0]);
                return TRAV_MORE;
            }
#undef TG_CHILD
        } else {
            if (COUNT) tc.tris++;
            float t, u, v;
            if (triTest(q0, q1, q2, q3.y, O, D, tnear, tfar0, sc.hitPad, t, u, v)) {
                if (ANYDYN ? any : ANY) return TRAV_ANYHIT;
                const uint32_t id = __float_as_uint(q3.x);
                if (t < bestT || (t == bestT && id < bestPrim)) {
                    bestT = t; bestU = u; bestV = v; bestPrim = id;
                }
            }
            if (payload & 7u) { // next triangle of the leaf: first + 1, count - 1
                node += 7u;
                return TRAV_MORE;
            }
        }
    }
    if (sp == 0) return TRAV_DONE;
    node = stackPop(sc, stack, sp);
    return TRAV_MORE;
}
// LDS "nodelets" (BASELINE north_star): the kernel a scene copy with ldsNodes != 0 is launched with has copied the first ldsNodes
// nodes (the top levels of the BVH4, breadth first: bvh_build.h, reorderTopLevels) behind its traversal stacks; a lane whose next
// record is one of them reads it from LDS (~50 cycles) instead of L2 (~200).  Measured: profiles/r02_ab_notes.txt.
TG_DEV void stageNodelets(const DScene &sc)
{
    extern __shared__ uint32_t ldsStack[];
    if (sc.ldsNodes) {
        float4 *dst = reinterpret_cast<float4 *>(ldsStack + sc.ldsNodeWord);
        for (uint32_t i = threadIdx.x; i < sc.ldsNodes*4u; i += blockDim.x) dst[i] = sc.nodes[i];
        __syncthreads();
    }
}
template <bool ANY, bool COUNT, bool SORTED = true>
TG_DEV int travStep(const DScene &sc, uint32_t *stack, TravState &r, TraceCounters &tc)
{
    extern __shared__ uint32_t ldsStack[];
    float4 q0, q1, q2, q3;
    if (r.node < sc.ldsNodes) { // never true for a leaf reference (bit 31) nor when nothing is staged
        const float4 *n = reinterpret_cast<const float4 *>(ldsStack + sc.ldsNodeWord) + size_t(r.node)*4;
        q0 = n[0]; q1 = n[1]; q2 = n[2]; q3 = n[3];
    } else {
        const float4 *n = travItem(sc, r);
        q0 = n[0]; q1 = n[1]; q2 = n[2]; q3 = n[3];
    }
    return travApply<ANY, COUNT, SORTED>(sc, stack, r, tc, q0, q1, q2, q3);
}

template <bool ANY, bool COUNT>
TG_DEV bool traverse(const DScene &sc, uint32_t *stack, uint32_t root, const V3 &O, const V3 &D, float tnear, float &bestT, float &bestU, float &bestV,
                     uint32_t &bestPrim, TraceCounters &tc)
{
    TravState r;
    travInit(r, root, O, D, tnear, bestT, bestPrim);
    r.bestU = bestU; r.bestV = bestV;
    int state;
    do state = travStep<ANY, COUNT>(sc, stack, r, tc); while (state == TRAV_MORE);
    bestT = r.bestT; bestU = r.bestU; bestV = r.bestV; bestPrim = r.bestPrim;
    return state == TRAV_ANYHIT;
}

// =================================================================================================
// "instances" primitives: Instance::intersect (primitives/Instance.cpp:290-311) on top of BinaryBvh::trace
// (bvh/BinaryBvh.hpp:188-283), reference behaviour included -- the local ray of every candidate instance starts at
// the leaf's entry distance with farT = infinity and the LAST instance hit in traversal order wins, so the BVH2 is
// walked in exactly the reference's order (single ray, ordered, (node, tMin) stack; the SSE lanes written per child).
// The master is intersected through the same code scene primitives use: the quantised BVH4 of its mesh
// (closest: lexicographic (t, triangle id)), Quad.cpp:71-98, Cube.cpp:94-127.
// ANY = true: stop at the first instance whose master is hit (shadow rays towards infinite lights: the set of
// leaves reached before the first hit does not depend on the order).
// =================================================================================================
TG_DEV V3 quatRotate(const float4 &q, const V3 &o) // Quaternion::operator*(Vec3), q = (w, x, y, z) in .x .y .z .w
{
    float tx = 2.0f*(q.z*o.z - q.w*o.y);
    float ty = 2.0f*(q.w*o.x - q.y*o.z);
    float tz = 2.0f*(q.y*o.y - q.z*o.x);
    return V3(o.x + q.x*tx + q.z*tz - q.w*ty,
              o.y + q.x*ty + q.w*tx - q.y*tz,
              o.z + q.x*tz + q.y*ty - q.z*tx);
}
TG_DEV float sseMax(float a, float b) { return a > b ? a : b; } // _mm_max_ps operand order

// Cube::intersect as a master sees it: 0 = false, 1 = true without a recorded hit (the slab test passed but neither
// face is inside (nearT, farT): Cube.cpp:113-124 still returns true), 2 = hit
TG_DEV int cubeMasterIntersect(const tgpu_cube &cb, const V3 &pos, const V3 &dir, float nearT, float farT, float &tHit)
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
    if (!(ttMin <= ttMax)) return 0;
    if (ttMin > nearT && ttMin < farT) { tHit = ttMin; return 2; }
    if (ttMax > nearT && ttMax < farT) { tHit = ttMax; return 2; }
    return 1;
}

template <bool ANY, bool COUNT>
TG_DEV bool instanceIntersect(const DScene &sc, const DInstSet &is, uint32_t setIndex, uint32_t *stack, const V3 &O, const V3 &D,
                              float tnear, float &rayFarT, uint32_t &prim, float &u, float &v, uint32_t &inst, TraceCounters &tc)
{
    // bboxIntersection(_bounds, ...) :148-171
    float tMin = tnear, tMax = rayFarT;
    {
        float ttMin = tMin, ttMax = tMax;
#pragma unroll
        for (int i = 0; i < 3; ++i) {
            float invD = 1.0f/D.comp(i);
            float relMin = is.boundsMin[i] - O.comp(i), relMax = is.boundsMax[i] - O.comp(i);
            if (invD >= 0.0f) {
                ttMin = fmax2(ttMin, relMin*invD);
                ttMax = fmin2(ttMax, relMax*invD);
            } else {
                ttMax = fmin2(ttMax, relMin*invD);
                ttMin = fmax2(ttMin, relMax*invD);
            }
        }
        if (!(ttMin <= ttMax)) return false;
        tMin = ttMin; tMax = ttMax;
    }
    const V3 invDir(1.0f/D.x, 1.0f/D.y, 1.0f/D.z);
    const bool px = D.x >= 0.0f, py = D.y >= 0.0f, pz = D.z >= 0.0f;
    const float nearLane = tnear;
    float negFarLane = -rayFarT;
    // the (node, tMin) stack of this walk: in LDS behind the mesh levels (sc.topStackLds entries per thread, the tree's depth + 1), like the
    // queue kernels' -- as local arrays it lived in scratch memory, a memory round trip per push and pop
    uint2 *topStack = reinterpret_cast<uint2 *>((stack - threadIdx.x) + sc.stackLds*256) + threadIdx.x;
    int sp = 0;
    uint32_t node = 0;
    bool hit = false;
    while (true) {
        float4 q3 = is.nodes[size_t(node)*4 + 3];
        bool reachedLeaf = true;
        while (__float_as_uint(q3.y) != 0xFFFFFFFFu) {
            const float4 *n = is.nodes + size_t(node)*4;
            float4 bx = n[0], by = n[1], bz = n[2];
            if (COUNT) tc.nodes++;
            // child 0 = lanes (x: min, z: max), child 1 = lanes (y: min, w: max)
            float nx0 = px ? bx.x : bx.z, fx0 = px ? bx.z : bx.x, nx1 = px ? bx.y : bx.w, fx1 = px ? bx.w : bx.y;
            float ny0 = py ? by.x : by.z, fy0 = py ? by.z : by.x, ny1 = py ? by.y : by.w, fy1 = py ? by.w : by.y;
            float nz0 = pz ? bz.x : bz.z, fz0 = pz ? bz.z : bz.x, nz1 = pz ? bz.y : bz.w, fz1 = pz ? bz.w : bz.y;
            float mn0 = sseMax(sseMax(sseMax((nx0 - O.x)*invDir.x, (ny0 - O.y)*invDir.y), (nz0 - O.z)*invDir.z), nearLane);
            float mn1 = sseMax(sseMax(sseMax((nx1 - O.x)*invDir.x, (ny1 - O.y)*invDir.y), (nz1 - O.z)*invDir.z), nearLane);
            float mx0 = -sseMax(sseMax(sseMax((fx0 - O.x)*(-invDir.x), (fy0 - O.y)*(-invDir.y)), (fz0 - O.z)*(-invDir.z)), negFarLane);
            float mx1 = -sseMax(sseMax(sseMax((fx1 - O.x)*(-invDir.x), (fy1 - O.y)*(-invDir.y)), (fz1 - O.z)*(-invDir.z)), negFarLane);
            const bool hitL = mn0 <= mx0, hitR = mn1 <= mx1;
            const uint32_t l = __float_as_uint(q3.x), r = __float_as_uint(q3.y);
            if (hitL && hitR) {
                if (mn0 < mn1) { topStack[sp*256] = make_uint2(r, __float_as_uint(mn1)); ++sp; node = l; tMin = mn0; }
                else { topStack[sp*256] = make_uint2(l, __float_as_uint(mn0)); ++sp; node = r; tMin = mn1; }
            } else if (hitL) { node = l; tMin = mn0; }
            else if (hitR) { node = r; tMin = mn1; }
            else { reachedLeaf = false; break; }
            q3 = is.nodes[size_t(node)*4 + 3];
        }
        if (reachedLeaf) {
            const uint32_t first = __float_as_uint(q3.z), count = __float_as_uint(q3.w);
            for (uint32_t i = first; i < first + count; ++i) {
                const uint32_t id = is.order[i];
                const float4 pq = is.pos[id], rq = is.rot[id];
                const float4 inv = make_float4(rq.x, -rq.y, -rq.z, -rq.w); // conjugate
                const V3 lo = quatRotate(inv, O - V3(pq.x, pq.y, pq.z)), ld = quatRotate(inv, D);
                const DMaster m = is.masters[__float_as_uint(pq.w)];
                bool hitI = false;
                float localFarT = TG_INF; // Ray::scatter default (math/Ray.hpp:29)
                if (m.kind == TGPU_MASTER_MESH) {
                    float t = TG_INF, tu = 0.0f, tv = 0.0f;
                    uint32_t tri = PRIM_MISS;
                    bool any = traverse<ANY, COUNT>(sc, stack, m.ref, lo, ld, tMin, t, tu, tv, tri, tc);
                    if (ANY) {
                        hitI = any;
                    } else if (tri != PRIM_MISS) {
                        hitI = true; localFarT = t; prim = tri; u = tu; v = tv;
                    }
                } else if (m.kind == TGPU_MASTER_QUAD) {
                    QuadHit h;
                    if (quadIntersect(sc.quads[m.ref], lo, ld, tMin, TG_INF, h)) {
                        hitI = true; localFarT = h.t; prim = mkPrim(KIND_QUAD, m.ref); u = h.u; v = h.v;
                    }
                } else {
                    float t = 0.0f;
                    int r = cubeMasterIntersect(sc.cubes[m.ref], lo, ld, tMin, TG_INF, t);
                    if (r == 2) { localFarT = t; prim = mkPrim(KIND_CUBE, m.ref); u = 0.0f; v = 0.0f; }
                    hitI = r != 0;
                }
                if (hitI) {
                    if (ANY) return true;
                    hit = true;
                    inst = (setIndex << 24) | id;
                    rayFarT = localFarT;
                }
            }
            tMax = fmin2(tMax, rayFarT);
            negFarLane = -tMax;
        }
        // pop
        bool done = false;
        while (true) {
            if (sp == 0) { done = true; break; }
            --sp;
            const uint2 e = topStack[sp*256];
            node = e.x;
            tMin = __uint_as_float(e.y);
            if (tMax < tMin) continue;
            break;
        }
        if (done) break;
    }
    return hit;
}

// TraceableScene::intersect restated: analytic primitives first (each shrinking tfar, Quad.cpp:71-98,
// Cube.cpp:94-127), then the triangle BVH.  Returns encoded prim or PRIM_MISS.
template <bool COUNT, bool INST>
TG_DEV uint32_t closestHit(const DScene &sc, uint32_t *stack, const V3 &O, const V3 &D, float tnear, float &tfar, float &u, float &v, uint32_t &inst,
                           TraceCounters &tc)
{
    uint32_t prim = PRIM_MISS;
    inst = INST_NONE;
    // "instances" primitives come first: their result ignores what was found before them (see instanceIntersect),
    // the primitives after them compare against whatever farT they leave behind
    if (INST) {
        for (uint32_t s = 0; s < sc.nInstSets; ++s)
            instanceIntersect<false, COUNT>(sc, sc.instSets[s], s, stack, O, D, tnear, tfar, prim, u, v, inst, tc);
        if (prim == PRIM_MISS) inst = INST_NONE;
    }
    for (uint32_t i = 0; i < sc.nQuads; ++i) {
        QuadHit h;
        if (quadIntersect(sc.quads[i], O, D, tnear, tfar, h)) {
            tfar = h.t; u = h.u; v = h.v;
            prim = mkPrim(KIND_QUAD, i);
            inst = INST_NONE;
        }
    }
    for (uint32_t i = 0; i < sc.nCubes; ++i) {
        float t;
        bool back;
        if (cubeIntersect(sc.cubes[i], O, D, tnear, tfar, t, back)) {
            tfar = t; u = 0.0f; v = 0.0f;
            prim = mkPrim(KIND_CUBE, i);
            inst = INST_NONE;
        }
    }
    // spheres and disks: only in the general-geometry (INST) instantiation, so that scenes without them keep the lean kernels
    for (uint32_t i = 0; INST && i < sc.nSpheres; ++i) {
        float t;
        bool back;
        if (sphereIntersect(sc.spheres[i], O, D, tnear, tfar, t, back)) {
            tfar = t; u = 0.0f; v = 0.0f;
            prim = mkPrim(KIND_SPHERE, i);
            inst = INST_NONE;
        }
    }
    for (uint32_t i = 0; INST && i < sc.nDisks; ++i) {
        DiskHit h;
        if (diskIntersect(sc.disks[i], O, D, tnear, tfar, h)) {
            tfar = h.t; u = h.rSq; v = 0.0f; // u carries DiskIntersection::rSq
            prim = mkPrim(KIND_DISK, i);
            inst = INST_NONE;
        }
    }
    if (sc.nTris) {
        uint32_t tri = PRIM_MISS;
        float t = tfar, tu = 0.0f, tv = 0.0f;
        traverse<false, COUNT>(sc, stack, sc.rootChild, O, D, tnear, t, tu, tv, tri, tc);
        if (tri != PRIM_MISS) {
            tfar = t; u = tu; v = tv;
            prim = tri; // KIND_TRI == 0
            inst = INST_NONE;
        }
    }
    return prim;
}

// generalizedShadowRay for scenes without forward-lobe BSDFs: blocked iff any primitive other than
// the end cap is hit inside (tnear, tfar) (TraceBase.cpp:62-125)
template <bool COUNT, bool INST>
TG_DEV bool occluded(const DScene &sc, uint32_t *stack, const V3 &O, const V3 &D, float tnear, float tfar, uint32_t endCap, TraceCounters &tc)
{
    if (INST && sc.nInstSets) {
        if (endCap != PRIM_MISS && primKind(endCap) != KIND_ENV) {
            // finite end cap: an instance may leave farT BEYOND the light (its local rays run to infinity), in which case
            // the light quad itself becomes the final hit and the ray counts as unblocked -- only the full
            // closest-hit query of generalizedShadowRay (TraceBase.cpp:79) reproduces that
            float t = tfar, u, v;
            uint32_t inst;
            uint32_t prim = closestHit<COUNT, true>(sc, stack, O, D, tnear, t, u, v, inst, tc);
            return prim != PRIM_MISS && !(inst == INST_NONE && prim == endCap);
        }
        for (uint32_t s = 0; s < sc.nInstSets; ++s) {
            float t = tfar, u, v;
            uint32_t prim, inst;
            if (instanceIntersect<true, COUNT>(sc, sc.instSets[s], s, stack, O, D, tnear, t, prim, u, v, inst, tc)) return true;
        }
    }
    for (uint32_t i = 0; i < sc.nQuads; ++i) {
        if (mkPrim(KIND_QUAD, i) == endCap) continue;
        QuadHit h;
        if (quadIntersect(sc.quads[i], O, D, tnear, tfar, h)) return true;
    }
    for (uint32_t i = 0; i < sc.nCubes; ++i) {
        float t;
        bool back;
        if (cubeIntersect(sc.cubes[i], O, D, tnear, tfar, t, back)) return true;
    }
    for (uint32_t i = 0; INST && i < sc.nSpheres; ++i) {
        if (mkPrim(KIND_SPHERE, i) == endCap) continue;
        float t;
        bool back;
        if (sphereIntersect(sc.spheres[i], O, D, tnear, tfar, t, back)) return true;
    }
    for (uint32_t i = 0; INST && i < sc.nDisks; ++i) {
        if (mkPrim(KIND_DISK, i) == endCap) continue;
        DiskHit h;
        if (diskIntersect(sc.disks[i], O, D, tnear, tfar, h)) return true;
    }
    if (sc.nTris) {
        float t = tfar, u, v;
        uint32_t p = PRIM_MISS;
        if (traverse<true, COUNT>(sc, stack, sc.rootChild, O, D, tnear, t, u, v, p, tc)) return true;
    }
    return false;
}

// =================================================================================================
// The same queries as closestHit<.., true> / occluded<.., true> as a FLAT state machine: one 64-byte record per step and lane --
// a node of the instance BVH2, an instance entry, or a node / triangle of a mesh BVH4 -- so that the lanes of a wave advance
// together whatever level of the two-level structure each of them is in (the nested loops of instanceIntersect make a wave
// wait for its deepest lane at every level), and so that the per-lane ray queues of k_extend_q / k_shadow_q can drive it.
// Order, arithmetic and the reference's quirks (local rays start at the leaf's entry distance with farT = infinity, the last
// instance hit in traversal order wins: Instance.cpp:290-311, BinaryBvh.hpp:188-283) are instanceIntersect's, line by line.
// The (node, tMin) stack of the top-level walk lives in LDS: topStack[level*256] per thread, sc.topStackLds levels.
// =================================================================================================
enum { IP_TOP = 0, IP_ENTRY = 1, IP_BOTTOM = 2, IP_SCENE = 3, IP_DONE = 4 };
struct InstTrav {
    V3 O, D, invDir;
    float tnear, farT;          // the world ray; farT is what instance hits overwrite
    float tMin, tMax, negFarLane;
    uint32_t topNode, leafI, leafEnd, set, curInst;
    int topSp, phase;
    bool any, blocked;          // any: any-hit query (towards infinite lights); blocked: its answer
    uint32_t prim, inst, endCap;
    float u, v;
    TravState b;                // the mesh BVH4 walk of the instance being visited (local ray), then of the scene's own triangles
};

// next set (or the part after the sets) -- instanceIntersect's prologue: bboxIntersection(_bounds, ...) :148-171
TG_DEV void instBeginSet(const DScene &sc, InstTrav &r);
// closestHit / occluded after the instance sets: analytic primitives in place, then the scene's triangles as one more BVH walk
TG_DEV void instAfterSets(const DScene &sc, InstTrav &r)
{
    const V3 &O = r.O, &D = r.D;
    if (!r.any) {
        if (r.prim == PRIM_MISS) r.inst = INST_NONE;
        for (uint32_t i = 0; i < sc.nQuads; ++i) {
            QuadHit h;
            if (quadIntersect(sc.quads[i], O, D, r.tnear, r.farT, h)) { r.farT = h.t; r.u = h.u; r.v = h.v; r.prim = mkPrim(KIND_QUAD, i); r.inst = INST_NONE; }
        }
        for (uint32_t i = 0; i < sc.nCubes; ++i) {
            float t;
            bool back;
            if (cubeIntersect(sc.cubes[i], O, D, r.tnear, r.farT, t, back)) { r.farT = t; r.u = 0.0f; r.v = 0.0f; r.prim = mkPrim(KIND_CUBE, i); r.inst = INST_NONE; }
        }
        for (uint32_t i = 0; i < sc.nSpheres; ++i) {
            float t;
            bool back;
            if (sphereIntersect(sc.spheres[i], O, D, r.tnear, r.farT, t, back)) { r.farT = t; r.u = 0.0f; r.v = 0.0f; r.prim = mkPrim(KIND_SPHERE, i); r.inst = INST_NONE; }
        }
        for (uint32_t i = 0; i < sc.nDisks; ++i) {
            DiskHit h;
            if (diskIntersect(sc.disks[i], O, D, r.tnear, r.farT, h)) { r.farT = h.t; r.u = h.rSq; r.v = 0.0f; r.prim = mkPrim(KIND_DISK, i); r.inst = INST_NONE; }
        }
    } else {
        bool hit = false;
        for (uint32_t i = 0; i < sc.nQuads; ++i) {
            QuadHit h;
            if (mkPrim(KIND_QUAD, i) != r.endCap && quadIntersect(sc.quads[i], O, D, r.tnear, r.farT, h)) hit = true;
        }
        for (uint32_t i = 0; i < sc.nCubes; ++i) {
            float t;
            bool back;
            if (cubeIntersect(sc.cubes[i], O, D, r.tnear, r.farT, t, back)) hit = true;
        }
        for (uint32_t i = 0; i < sc.nSpheres; ++i) {
            float t;
            bool back;
            if (mkPrim(KIND_SPHERE, i) != r.endCap && sphereIntersect(sc.spheres[i], O, D, r.tnear, r.farT, t, back)) hit = true;
        }
        for (uint32_t i = 0; i < sc.nDisks; ++i) {
            DiskHit h;
            if (mkPrim(KIND_DISK, i) != r.endCap && diskIntersect(sc.disks[i], O, D, r.tnear, r.farT, h)) hit = true;
        }
        if (hit) {
            r.blocked = true;
            r.phase = IP_DONE;
            return;
        }
    }
    if (sc.nTris) {
        travInit(r.b, sc.rootChild, O, D, r.tnear, r.farT, PRIM_MISS);
        r.phase = IP_SCENE;
    } else {
        r.phase = IP_DONE;
    }
}
TG_DEV void instBeginSet(const DScene &sc, InstTrav &r)
{
    while (r.set < sc.nInstSets) {
        const DInstSet &is = sc.instSets[r.set];
        float ttMin = r.tnear, ttMax = r.farT;
#pragma unroll
        for (int i = 0; i < 3; ++i) {
            float invD = 1.0f/r.D.comp(i);
            float relMin = is.boundsMin[i] - r.O.comp(i), relMax = is.boundsMax[i] - r.O.comp(i);
            if (invD >= 0.0f) {
                ttMin = fmax2(ttMin, relMin*invD);
                ttMax = fmin2(ttMax, relMax*invD);
            } else {
                ttMax = fmin2(ttMax, relMin*invD);
                ttMin = fmax2(ttMin, relMax*invD);
            }
        }
        if (ttMin <= ttMax) {
            r.tMin = ttMin;
            r.tMax = ttMax;
            r.negFarLane = -r.farT;
            r.topNode = 0;
            r.topSp = 0;
            r.phase = IP_TOP;
            return;
        }
        r.set++;
    }
    instAfterSets(sc, r);
}
// a query for ray (O, D, tnear, tfar): closest hit, or any hit (`any`; endCap = the primitive the ray is aimed at)
TG_DEV void instInit(const DScene &sc, InstTrav &r, const V3 &O, const V3 &D, float tnear, float tfar, bool any, uint32_t endCap)
{
    r.O = O; r.D = D;
    r.invDir = V3(1.0f/D.x, 1.0f/D.y, 1.0f/D.z);
    r.tnear = tnear; r.farT = tfar;
    r.any = any; r.blocked = false; r.endCap = endCap;
    r.prim = PRIM_MISS; r.inst = INST_NONE; r.u = 0.0f; r.v = 0.0f;
    r.set = 0;
    instBeginSet(sc, r);
}
// the record the next step needs
TG_DEV const float4 *instItem(const DScene &sc, const InstTrav &r)
{
    if (r.phase == IP_TOP) return sc.instSets[r.set].nodes + size_t(r.topNode)*4;
    if (r.phase == IP_ENTRY) return sc.instSets[r.set].entries + size_t(r.leafI)*4;
    return travItem(sc, r.b);
}
// leave the current position of the top-level walk: pop until an entry is still in range, or finish the set
TG_DEV void instPop(const DScene &sc, InstTrav &r, const uint2 *topStack)
{
    while (true) {
        if (r.topSp == 0) {
            r.set++;
            instBeginSet(sc, r);
            return;
        }
        --r.topSp;
        const uint2 e = topStack[r.topSp*256];
        r.topNode = e.x;
        r.tMin = __uint_as_float(e.y);
        if (r.tMax < r.tMin) continue;
        r.phase = IP_TOP;
        return;
    }
}
// after the last instance of a leaf: BinaryBvh.hpp:262-264
TG_DEV void instLeafDone(const DScene &sc, InstTrav &r, const uint2 *topStack)
{
    r.tMax = fmin2(r.tMax, r.farT);
    r.negFarLane = -r.tMax;
    instPop(sc, r, topStack);
}
// an instance was hit at local distance t (Instance.cpp:301-308): it wins whatever was found before
TG_DEV void instRecordHit(InstTrav &r, float t, uint32_t prim, float u, float v)
{
    r.farT = t;
    r.prim = prim; r.u = u; r.v = v;
    r.inst = (r.set << 24) | r.curInst;
}

template <bool COUNT>
TG_DEV void instStep(const DScene &sc, uint32_t *stack, uint2 *topStack, InstTrav &r, TraceCounters &tc, const float4 &q0, const float4 &q1, const float4 &q2, const float4 &q3)
{
    if (r.phase == IP_TOP) {
        if (__float_as_uint(q3.y) == 0xFFFFFFFFu) { // a leaf: its instances next
            r.leafI = __float_as_uint(q3.z);
            r.leafEnd = r.leafI + __float_as_uint(q3.w);
            if (r.leafI < r.leafEnd) r.phase = IP_ENTRY;
            else instLeafDone(sc, r, topStack);
            return;
        }
        if (COUNT) tc.nodes++;
        const V3 &O = r.O, &invDir = r.invDir;
        const bool px = r.D.x >= 0.0f, py = r.D.y >= 0.0f, pz = r.D.z >= 0.0f;
        const float4 &bx = q0, &by = q1, &bz = q2;
        // child 0 = lanes (x: min, z: max), child 1 = lanes (y: min, w: max)
        float nx0 = px ? bx.x : bx.z, fx0 = px ? bx.z : bx.x, nx1 = px ? bx.y : bx.w, fx1 = px ? bx.w : bx.y;
        float ny0 = py ? by.x : by.z, fy0 = py ? by.z : by.x, ny1 = py ? by.y : by.w, fy1 = py ? by.w : by.y;
        float nz0 = pz ? bz.x : bz.z, fz0 = pz ? bz.z : bz.x, nz1 = pz ? bz.y : bz.w, fz1 = pz ? bz.w : bz.y;
        float mn0 = sseMax(sseMax(sseMax((nx0 - O.x)*invDir.x, (ny0 - O.y)*invDir.y), (nz0 - O.z)*invDir.z), r.tnear);
        float mn1 = sseMax(sseMax(sseMax((nx1 - O.x)*invDir.x, (ny1 - O.y)*invDir.y), (nz1 - O.z)*invDir.z), r.tnear);
        float mx0 = -sseMax(sseMax(sseMax((fx0 - O.x)*(-invDir.x), (fy0 - O.y)*(-invDir.y)), (fz0 - O.z)*(-invDir.z)), r.negFarLane);
        float mx1 = -sseMax(sseMax(sseMax((fx1 - O.x)*(-invDir.x), (fy1 - O.y)*(-invDir.y)), (fz1 - O.z)*(-invDir.z)), r.negFarLane);
        const bool hitL = mn0 <= mx0, hitR = mn1 <= mx1;
        const uint32_t l = __float_as_uint(q3.x), rr = __float_as_uint(q3.y);
        if (hitL && hitR) {
            if (mn0 < mn1) { topStack[r.topSp*256] = make_uint2(rr, __float_as_uint(mn1)); ++r.topSp; r.topNode = l; r.tMin = mn0; }
            else { topStack[r.topSp*256] = make_uint2(l, __float_as_uint(mn0)); ++r.topSp; r.topNode = rr; r.tMin = mn1; }
        } else if (hitL) { r.topNode = l; r.tMin = mn0; }
        else if (hitR) { r.topNode = rr; r.tMin = mn1; }
        else instPop(sc, r, topStack);
        return;
    }
    if (r.phase == IP_ENTRY) { // one instance of the leaf: Instance::intersect's lambda (:292-309)
        const V3 pos(q0.x, q0.y, q0.z);
        r.curInst = __float_as_uint(q0.w);
        const float4 inv = make_float4(q1.x, -q1.y, -q1.z, -q1.w); // conjugate
        const V3 lo = quatRotate(inv, r.O - pos), ld = quatRotate(inv, r.D);
        const int kind = int(__float_as_uint(q2.x));
        const uint32_t ref = __float_as_uint(q2.y);
        if (kind == TGPU_MASTER_MESH) {
            travInit(r.b, ref, lo, ld, r.tMin, TG_INF, PRIM_MISS); // Ray::scatter: farT = infinity (math/Ray.hpp:29)
            r.phase = IP_BOTTOM;
            return;
        }
        bool hitI = false;
        if (kind == TGPU_MASTER_QUAD) {
            QuadHit h;
            if (quadIntersect(sc.quads[ref], lo, ld, r.tMin, TG_INF, h)) {
                hitI = true;
                if (!r.any) instRecordHit(r, h.t, mkPrim(KIND_QUAD, ref), h.u, h.v);
            }
        } else {
            float t = 0.0f;
            const int c = cubeMasterIntersect(sc.cubes[ref], lo, ld, r.tMin, TG_INF, t);
            if (c != 0) {
                hitI = true;
                if (!r.any) {
                    if (c == 2) instRecordHit(r, t, mkPrim(KIND_CUBE, ref), 0.0f, 0.0f);
                    else { r.farT = TG_INF; r.inst = (r.set << 24) | r.curInst; } // Cube.cpp:113-124 returns true without a recorded hit: farT = the local ray's infinity, prim unchanged
                }
            }
        }
        if (hitI && r.any) { r.blocked = true; r.phase = IP_DONE; return; }
        if (++r.leafI == r.leafEnd) instLeafDone(sc, r, topStack);
        return;
    }
    // a mesh BVH4: the instance's master (local ray) or the scene's own triangles
    const int s = travApply<false, COUNT, true, true>(sc, stack, r.b, tc, q0, q1, q2, q3, r.any);
    if (s == TRAV_MORE) return;
    if (r.phase == IP_BOTTOM) {
        if (r.any) {
            if (s == TRAV_ANYHIT) { r.blocked = true; r.phase = IP_DONE; return; }
        } else if (r.b.bestPrim != PRIM_MISS) {
            instRecordHit(r, r.b.bestT, r.b.bestPrim, r.b.bestU, r.b.bestV);
        }
        if (++r.leafI == r.leafEnd) instLeafDone(sc, r, topStack);
        else r.phase = IP_ENTRY;
        return;
    }
    // IP_SCENE
    if (r.any) {
        r.blocked = s == TRAV_ANYHIT;
    } else if (r.b.bestPrim != PRIM_MISS) {
        r.farT = r.b.bestT; r.u = r.b.bestU; r.v = r.b.bestV;
        r.prim = r.b.bestPrim;
        r.inst = INST_NONE;
    }
    r.phase = IP_DONE;
}

TG_DEV void flushCounters(unsigned long long *stats, int slot, const TraceCounters &tc)
{
    // wave reduction, one atomic pair per wave
    uint32_t n = tc.nodes, t = tc.tris;
    for (int off = 32; off > 0; off >>= 1) {
        n += __shfl_down(n, off);
        t += __shfl_down(t, off);
    }
    if ((threadIdx.x & 63) == 0) {
        atomicAdd(&stats[slot], (unsigned long long)n);
        atomicAdd(&stats[slot + 1], (unsigned long long)t);
    }
}

TG_DEV uint32_t hitBsdfType(const DScene &sc, uint32_t prim) // tgpu_bsdf::type of the primitive's BSDF
{
    uint32_t kind = primKind(prim), idx = primIndex(prim);
    if (kind == KIND_TRI) return sc.triClass[idx];
    if (kind == KIND_QUAD) return uint32_t(sc.bsdfs[sc.quads[idx].bsdf].type);
    if (kind == KIND_SPHERE) return uint32_t(sc.bsdfs[sc.spheres[idx].bsdf].type);
    if (kind == KIND_DISK) return uint32_t(sc.bsdfs[sc.disks[idx].bsdf].type);
    return uint32_t(sc.bsdfs[sc.cubes[idx].bsdf].type);
}
TG_DEV uint32_t hitClass(const DScene &sc, uint32_t prim)
{
    if (prim == PRIM_MISS) return 0xFFu;
    uint32_t kind = primKind(prim), idx = primIndex(prim), type;
    if (kind == KIND_TRI) type = sc.triClass[idx];
    else if (kind == KIND_QUAD) type = uint32_t(sc.bsdfs[sc.quads[idx].bsdf].type);
    else if (kind == KIND_SPHERE) type = uint32_t(sc.bsdfs[sc.spheres[idx].bsdf].type);
    else if (kind == KIND_DISK) type = uint32_t(sc.bsdfs[sc.disks[idx].bsdf].type);
    else type = uint32_t(sc.bsdfs[sc.cubes[idx].bsdf].type);
    return type < uint32_t(TG_CLASS_OTHER) ? type : uint32_t(TG_CLASS_OTHER);
}

// =================================================================================================
// K2 extend: persistent waves pull 64 list entries per head atomic; per-lane traversal stack in LDS
// (stack[level][thread], bank = lane -> conflict free); after 4 such batches the wave appends its slots
// to the per-BSDF-class lists of the other set with ONE atomic per class.  Idle slots are
// dropped onto the free list.  RAW = true: ray-level entry point (tgpu_trace_closest): identity queue,
// no classification.
// =================================================================================================
template <bool COUNT, bool RAW, bool INST>
#ifndef TG_EXTEND_INST_WAVES
#define TG_EXTEND_INST_WAVES 4 /* waves per SIMD the general-geometry closest-hit kernel is compiled for: 4 = 128 VGPRs, no extra spills; config 4 140 -> 153 Msamples/s against 141 VGPRs / 3 waves, 5 waves (96 VGPRs) 133 (profiles/r02_ab_notes.txt) */
#endif
__global__ void __launch_bounds__(256, (INST ? TG_EXTEND_INST_WAVES : TG_EXTEND_WAVES)) k_extend(DScene sc, DState st, int set, uint32_t rawCount)
{
    extern __shared__ uint32_t ldsStack[];
    uint32_t *stack = ldsStack + threadIdx.x;
    const int lane = threadIdx.x & 63;
    ListCounts lc{};
    if (!RAW) lc = loadListCounts(st, set);
    const uint32_t count = RAW ? rawCount : listTotal(lc, 0);
    TraceCounters tc{0, 0};
    bool drained = false;
    while (!drained) {
        uint32_t p0 = 0, p1 = 0, p2 = 0, p3 = 0, cls4 = 0xFEFEFEFEu; // 0xFE = no entry
        for (int b = 0; b < 4; ++b) {
            // one head atomic per 64 entries keeps the waves balanced; the list appends below are the
            // expensive atomics (few hot words) and are shared by the 4 batches
            uint32_t base = 0;
            if (lane == 0) base = atomicAdd(&st.counters[CNT(CNT_HEAD_EXT)], 64u);
            base = __shfl(base, 0);
            if (base >= count) {
                drained = true;
                break;
            }
            uint32_t idx = base + lane;
            uint32_t path = 0, cls = 0xFEu;
            bool have = idx < count;
            if (have) {
                if (RAW) path = idx;
                else have = listFetch(st, lc, set, 0, idx, path);
            }
            if (have) {
                bool dead = !RAW && flagsEnd(__float_as_uint(ldStream<INST ? TG_STATE_NT_INST : TG_STATE_NT>(&st.tp[path]).w)) == END_DEAD;
                if (dead) {
                    cls = 0xFDu;
                } else {
                    float4 ro = ldStream<INST ? TG_STATE_NT_INST : TG_STATE_NT>(&st.rayO[path]), rd = ldStream<INST ? TG_STATE_NT_INST : TG_STATE_NT>(&st.rayD[path]);
                    V3 O = xyz(ro), D = xyz(rd);
                    float tfar = rd.w, u = 0.0f, v = 0.0f;
                    uint32_t inst;
                    uint32_t prim = closestHit<COUNT, INST>(sc, stack, O, D, ro.w, tfar, u, v, inst, tc);
                    if (INST) {
                        if (RAW) { if (inst != INST_NONE) prim = mkPrim(KIND_ENV, inst & 0xFFFFFFu); } // ray-level ABI: kind 3 = instance index
                        else stStream<INST ? TG_STATE_NT_INST : TG_STATE_NT>(&st.hitInst[path], inst);
                    }
                    stStream<INST ? TG_STATE_NT_INST : TG_STATE_NT>(&st.hit[path], make_float4(tfar, u, v, __uint_as_float(prim)));
                    if (!RAW) {
                        cls = hitClass(sc, prim);
                        if (sc.nMedia && (st.med[path] & 0xFFu)) cls = TG_CLASS_VOLUME;
                    }
                }
            }
            p0 = b == 0 ? path : p0; p1 = b == 1 ? path : p1; p2 = b == 2 ? path : p2; p3 = b == 3 ? path : p3;
            cls4 = (cls4 & ~(0xFFu << (8*b))) | (cls << (8*b));
        }
        if (!RAW) {
            const int out = set ^ 1;
            // class 0xFF (miss) -> list 0, class c -> list 1 + c, 0xFD (idle) -> free list
#pragma unroll
            for (int l = 0; l <= TG_LISTS; ++l) {
                const uint32_t want = l == 0 ? 0xFFu : (l == TG_LISTS ? 0xFDu : uint32_t(l - 1));
                unsigned long long m0 = __ballot((cls4 & 0xFFu) == want), m1 = __ballot(((cls4 >> 8) & 0xFFu) == want);
                unsigned long long m2 = __ballot(((cls4 >> 16) & 0xFFu) == want), m3 = __ballot((cls4 >> 24) == want);
                uint32_t n0 = uint32_t(__builtin_popcountll(m0)), n1 = uint32_t(__builtin_popcountll(m1));
                uint32_t n2 = uint32_t(__builtin_popcountll(m2)), n3 = uint32_t(__builtin_popcountll(m3));
                uint32_t n = n0 + n1 + n2 + n3;
                if (n == 0) continue;
                uint32_t *dst = l == TG_LISTS ? st.freeQ : st.lists[out][l];
                uint32_t *ctr = l == TG_LISTS ? &st.counters[CNT(CNT_FREE)] : &st.counters[CNT(CNT_LIST0 + out*TG_LISTS + l)];
                uint32_t wbase = 0;
                if (lane == 0) wbase = atomicAdd(ctr, n);
                wbase = __shfl(wbase, 0);
                const unsigned long long below = (1ull << lane) - 1ull;
                if ((m0 >> lane) & 1ull) stStream<INST ? TG_STATE_NT_INST : TG_STATE_NT>(&dst[wbase + uint32_t(__builtin_popcountll(m0 & below))], p0);
                if ((m1 >> lane) & 1ull) stStream<INST ? TG_STATE_NT_INST : TG_STATE_NT>(&dst[wbase + n0 + uint32_t(__builtin_popcountll(m1 & below))], p1);
                if ((m2 >> lane) & 1ull) stStream<INST ? TG_STATE_NT_INST : TG_STATE_NT>(&dst[wbase + n0 + n1 + uint32_t(__builtin_popcountll(m2 & below))], p2);
                if ((m3 >> lane) & 1ull) stStream<INST ? TG_STATE_NT_INST : TG_STATE_NT>(&dst[wbase + n0 + n1 + n2 + uint32_t(__builtin_popcountll(m3 & below))], p3);
            }
        }
    }
    if (COUNT) flushCounters(st.stats, 0, tc);
}

// =================================================================================================
// K2 extend, queue form (scenes without instances / spheres / disks): a wave claims 64*DEPTH list entries and every lane
// owns DEPTH (4 or 8) of them, traced back to back inside ONE traversal loop.  With one ray per lane a wave runs for as long as
// its slowest ray (measured: 25 % of the lanes busy per VALU instruction, SQ_THREAD_CYCLES_VALU); with a private
// queue per lane it runs for the slowest SUM of DEPTH rays, which varies much less.  Finished lanes are re-armed in
// groups (>= TG_EXTQ_THRESH waiting lanes, or nobody left tracing), so the re-arm code runs a few times per claim,
// not once per ray, and the next ray of each lane is already in registers (prefetched while the current one is
// traced), so re-arming never waits for memory.  Per-ray arithmetic is closestHit's, unchanged.
// =================================================================================================
#ifndef TG_EXTQ_THRESH
#define TG_EXTQ_THRESH 16
#endif
#ifndef TG_EXTQ_WAVES
#define TG_EXTQ_WAVES 6
#endif
#ifndef TG_EXTQ_WAVES_DEEP
#define TG_EXTQ_WAVES_DEEP 5
#endif
constexpr uint32_t PRIM_IDLE = 0xFFFFFFFDu; // queue entry whose slot holds no live path

template <bool COUNT, int DEPTH>
__global__ void __launch_bounds__(256, (DEPTH > 4 ? TG_EXTQ_WAVES_DEEP : TG_EXTQ_WAVES)) k_extend_q(DScene sc, DState st, int set)
{
    static_assert(DEPTH % 4 == 0, "the list appends work on groups of four entries");
    extern __shared__ uint32_t ldsStack[];
    uint32_t *stack = ldsStack + threadIdx.x;
    const int lane = threadIdx.x & 63;
    stageNodelets(sc);
    const ListCounts lc = loadListCounts(st, set);
    const uint32_t count = listTotal(lc, 0);
    TraceCounters tc{0, 0};
    while (true) {
        uint32_t base = 0;
        if (lane == 0) base = atomicAdd(&st.counters[CNT(CNT_HEAD_EXT)], 64u*DEPTH);
        base = __shfl(base, 0);
        if (base >= count) break;
        // the lane's queue: entries base + lane + 64*b (the b-th rays of the wave are 64 consecutive entries, as before);
        // p[] and h[] are only ever indexed by unrolled loops, so they stay in registers
        uint32_t p[DEPTH], h[DEPTH]; // slot of each entry, prim each ray ended on
        int n = 0;
#pragma unroll
        for (int q = 0; q < DEPTH; ++q) {
            p[q] = 0;
            h[q] = PRIM_IDLE;
            if (n == q && base + lane + 64u*q < count && listFetch(st, lc, set, 0, base + lane + 64u*q, p[q])) n = q + 1;
        }
        auto slotOf = [&](int i) {
            uint32_t v = p[0];
#pragma unroll
            for (int q = 1; q < DEPTH; ++q) v = i == q ? p[q] : v;
            return v;
        };
        float4 nro = make_float4(0.0f, 0.0f, 0.0f, 0.0f), nrd = nro; // ray b + 1, in flight while ray b is traced
        uint32_t nflags = 0;
        if (n > 0) {
            nro = ldStream(&st.rayO[p[0]]);
            nrd = ldStream(&st.rayD[p[0]]);
            nflags = ldStream(reinterpret_cast<const uint32_t *>(&st.tp[p[0]]) + 3);
        }
        int b = -1;
        bool active = false, fin = false;
        TravState r;
        r.node = 0; r.sp = 0; r.bestT = 0.0f; r.bestU = 0.0f; r.bestV = 0.0f; r.bestPrim = PRIM_MISS;
        while (true) {
            if (active) {
                if (travStep<false, COUNT>(sc, stack, r, tc) != TRAV_MORE) {
                    active = false;
                    fin = true;
                }
            }
            const bool waiting = !active && (fin || b + 1 < n);
            const unsigned long long w = __ballot(waiting), act = __ballot(active);
            if (w == 0 && act == 0) break;
            if (act != 0 && __builtin_popcountll(w) < TG_EXTQ_THRESH) continue;
            if (waiting) {
                if (fin) { // retire ray b
                    stStream(&st.hit[slotOf(b)], make_float4(r.bestT, r.bestU, r.bestV, __uint_as_float(r.bestPrim)));
#pragma unroll
                    for (int q = 0; q < DEPTH; ++q) h[q] = b == q ? r.bestPrim : h[q];
                    fin = false;
                }
                if (b + 1 < n) { // arm ray b + 1 from the prefetch registers
                    ++b;
                    const bool dead = flagsEnd(nflags) == END_DEAD;
                    if (!dead) {
                        const V3 O = xyz(nro), D = xyz(nrd);
                        const float tnear = nro.w;
                        float tfar = nrd.w, u = 0.0f, v = 0.0f;
                        uint32_t prim = PRIM_MISS;
                        for (uint32_t i = 0; i < sc.nQuads; ++i) { // closestHit's order: quads, cubes, triangles
                            QuadHit qh;
                            if (quadIntersect(sc.quads[i], O, D, tnear, tfar, qh)) {
                                tfar = qh.t; u = qh.u; v = qh.v;
                                prim = mkPrim(KIND_QUAD, i);
                            }
                        }
                        for (uint32_t i = 0; i < sc.nCubes; ++i) {
                            float t;
                            bool back;
                            if (cubeIntersect(sc.cubes[i], O, D, tnear, tfar, t, back)) {
                                tfar = t; u = 0.0f; v = 0.0f;
                                prim = mkPrim(KIND_CUBE, i);
                            }
                        }
                        // the triangles start from what the quads and cubes left: a triangle wins when it is nearer, or at
                        // the same distance (its id is below every quad / cube id and below PRIM_MISS, as in traverse())
                        travInit(r, sc.rootChild, O, D, tnear, tfar, prim);
                        r.bestU = u; r.bestV = v;
                        if (sc.nTris) active = true;
                        else fin = true;
                    }
                    if (b + 1 < n) {
                        const uint32_t next = slotOf(b + 1);
                        nro = ldStream(&st.rayO[next]);
                        nrd = ldStream(&st.rayD[next]);
                        nflags = ldStream(reinterpret_cast<const uint32_t *>(&st.tp[next]) + 3);
                    }
                }
            }
        }
        // classification and the list appends of k_extend, four entries at a time
        const int out = set ^ 1;
#pragma unroll
        for (int g = 0; g < DEPTH; g += 4) {
            uint32_t cls4 = 0;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                uint32_t cls = 0xFEu;
                if (g + q < n) {
                    if (h[g + q] == PRIM_IDLE) {
                        cls = 0xFDu;
                    } else {
                        cls = hitClass(sc, h[g + q]);
                        if (sc.nMedia && (st.med[p[g + q]] & 0xFFu)) cls = TG_CLASS_VOLUME;
                    }
                }
                cls4 |= cls << (8*q);
            }
            if (__ballot(g < n) == 0) break;
#pragma unroll
            for (int l = 0; l <= TG_LISTS; ++l) {
                const uint32_t want = l == 0 ? 0xFFu : (l == TG_LISTS ? 0xFDu : uint32_t(l - 1));
                unsigned long long m0 = __ballot((cls4 & 0xFFu) == want), m1 = __ballot(((cls4 >> 8) & 0xFFu) == want);
                unsigned long long m2 = __ballot(((cls4 >> 16) & 0xFFu) == want), m3 = __ballot((cls4 >> 24) == want);
                uint32_t n0 = uint32_t(__builtin_popcountll(m0)), n1 = uint32_t(__builtin_popcountll(m1));
                uint32_t n2 = uint32_t(__builtin_popcountll(m2)), n3 = uint32_t(__builtin_popcountll(m3));
                uint32_t nn = n0 + n1 + n2 + n3;
                if (nn == 0) continue;
                uint32_t *dst = l == TG_LISTS ? st.freeQ : st.lists[out][l];
                uint32_t *ctr = l == TG_LISTS ? &st.counters[CNT(CNT_FREE)] : &st.counters[CNT(CNT_LIST0 + out*TG_LISTS + l)];
                uint32_t wbase = 0;
                if (lane == 0) wbase = atomicAdd(ctr, nn);
                wbase = __shfl(wbase, 0);
                const unsigned long long below = (1ull << lane) - 1ull;
                if ((m0 >> lane) & 1ull) stStream(&dst[wbase + uint32_t(__builtin_popcountll(m0 & below))], p[g]);
                if ((m1 >> lane) & 1ull) stStream(&dst[wbase + n0 + uint32_t(__builtin_popcountll(m1 & below))], p[g + 1]);
                if ((m2 >> lane) & 1ull) stStream(&dst[wbase + n0 + n1 + uint32_t(__builtin_popcountll(m2 & below))], p[g + 2]);
                if ((m3 >> lane) & 1ull) stStream(&dst[wbase + n0 + n1 + n2 + uint32_t(__builtin_popcountll(m3 & below))], p[g + 3]);
            }
        }
    }
    if (COUNT) flushCounters(st.stats, 0, tc);
}

// =================================================================================================
// K2p: the closest-hit kernel with PERSISTENT lanes (k_shadow_q's scheme).  In k_extend_q a lane owns four fixed list entries and the wave
// moves on when its slowest lane has finished all four -- with ray lengths spread as they are, the lanes that finish early idle for a
// good part of every 256-entry chunk.  Here a wave claims chunks of 256 entries with one atomic and hands the entries to its lanes one
// at a time, whenever a group of lanes is waiting; a lane always has its next ray's record in registers (fetched while the current one is
// traced) and its next list entry in flight.  Finished rays -- (slot, prim) pairs -- collect in a per-wave LDS buffer and are classified and
// appended to the lists of the other set 256 at a time by k_extend_q's append code: one atomic per class and 256 rays, as before.
// Per-ray arithmetic is closestHit's, unchanged.
// =================================================================================================
#ifdef TG_TU_MAIN
#ifndef TG_EXTP_WAVES
#define TG_EXTP_WAVES 5
#endif
#ifndef TG_EXTP_THRESH
#define TG_EXTP_THRESH 24 /* waiting lanes that trigger a re-arm round: 8 / 16 / 24 / 32 -> extend 5.50 / 5.15 / 5.09 / 5.15 ms per step */
#endif
constexpr uint32_t TG_EXTP_BUF = 320; // pairs per wave: a flush leaves < 64 behind and a re-arm round adds <= 64
template <bool COUNT>
static __global__ void __launch_bounds__(256, TG_EXTP_WAVES) k_extend_p(DScene sc, DState st, int set)
{
    constexpr uint32_t CHUNK = 256u;
    extern __shared__ uint32_t ldsStack[];
    uint32_t *stack = ldsStack + threadIdx.x;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const unsigned long long below = (1ull << lane) - 1ull;
    stageNodelets(sc);
    uint2 *buf = reinterpret_cast<uint2 *>(ldsStack + sc.ldsQueueWord) + size_t(wave)*TG_EXTP_BUF;
    const ListCounts lc = loadListCounts(st, set);
    const uint32_t count = listTotal(lc, 0);
    const int out = set ^ 1;
    TraceCounters tc{0, 0};
    uint32_t chunkPos = 0, chunkEnd = 0; // the wave's private chunk of list entries (wave-uniform)
    bool exhausted = false;
    auto claim = [&](bool want, uint32_t &slot) -> bool { // hands one list entry to every lane that wants one (all lanes call)
        bool got = false;
        unsigned long long m = __ballot(want);
        while (m) {
            if (chunkPos == chunkEnd) {
                if (exhausted) break;
                uint32_t b = 0;
                if (lane == 0) b = atomicAdd(&st.counters[CNT(CNT_HEAD_EXT)], CHUNK);
                b = uint32_t(__builtin_amdgcn_readfirstlane(int(b)));
                if (b >= count) {
                    exhausted = true;
                    break;
                }
                chunkPos = b;
                chunkEnd = min(b + CHUNK, count);
            }
            const uint32_t rank = uint32_t(__builtin_popcountll(m & below)), avail = chunkEnd - chunkPos;
            const bool mine = ((m >> lane) & 1ull) && rank < avail;
            if (mine) got = listFetch(st, lc, set, 0, chunkPos + rank, slot);
            chunkPos += min(uint32_t(__builtin_popcountll(m)), avail);
            m &= ~__ballot(mine);
        }
        return got;
    };
    // classify the first nFlush (<= 256) buffered pairs and append their slots to the lists of the other set (k_extend_q's code)
    auto flush = [&](uint32_t nFlush) {
        uint32_t p[4], h[4];
        int n = 0;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            p[q] = 0;
            h[q] = PRIM_IDLE;
            if (uint32_t(lane) + 64u*q < nFlush) {
                const uint2 e = buf[uint32_t(lane) + 64u*q];
                p[q] = e.x;
                h[q] = e.y;
                n = q + 1;
            }
        }
        uint32_t cls4 = 0;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            uint32_t cls = 0xFEu;
            if (q < n) {
                if (h[q] == PRIM_IDLE) {
                    cls = 0xFDu;
                } else {
                    cls = hitClass(sc, h[q]);
                    if (sc.nMedia && (st.med[p[q]] & 0xFFu)) cls = TG_CLASS_VOLUME;
                }
            }
            cls4 |= cls << (8*q);
        }
#pragma unroll
        for (int l = 0; l <= TG_LISTS; ++l) {
            const uint32_t want = l == 0 ? 0xFFu : (l == TG_LISTS ? 0xFDu : uint32_t(l - 1));
            unsigned long long m0 = __ballot((cls4 & 0xFFu) == want), m1 = __ballot(((cls4 >> 8) & 0xFFu) == want);
            unsigned long long m2 = __ballot(((cls4 >> 16) & 0xFFu) == want), m3 = __ballot((cls4 >> 24) == want);
            uint32_t n0 = uint32_t(__builtin_popcountll(m0)), n1 = uint32_t(__builtin_popcountll(m1));
            uint32_t n2 = uint32_t(__builtin_popcountll(m2)), n3 = uint32_t(__builtin_popcountll(m3));
            uint32_t nn = n0 + n1 + n2 + n3;
            if (nn == 0) continue;
            uint32_t *dst = l == TG_LISTS ? st.freeQ : st.lists[out][l];
            uint32_t *ctr = l == TG_LISTS ? &st.counters[CNT(CNT_FREE)] : &st.counters[CNT(CNT_LIST0 + out*TG_LISTS + l)];
            uint32_t wbase = 0;
            if (lane == 0) wbase = atomicAdd(ctr, nn);
            wbase = __shfl(wbase, 0);
            if ((m0 >> lane) & 1ull) stStream(&dst[wbase + uint32_t(__builtin_popcountll(m0 & below))], p[0]);
            if ((m1 >> lane) & 1ull) stStream(&dst[wbase + n0 + uint32_t(__builtin_popcountll(m1 & below))], p[1]);
            if ((m2 >> lane) & 1ull) stStream(&dst[wbase + n0 + n1 + uint32_t(__builtin_popcountll(m2 & below))], p[2]);
            if ((m3 >> lane) & 1ull) stStream(&dst[wbase + n0 + n1 + n2 + uint32_t(__builtin_popcountll(m3 & below))], p[3]);
        }
    };
    // lane state: the ray being traced (slot cs), the ray whose record is in flight (pSlot -> nro, nrd, nflags), the next list entry (nvs)
    uint32_t cs = 0, pSlot = 0, nvs = 0, nflags = 0;
    bool pre = false, haveNext = false;
    float4 nro = make_float4(0.0f, 0.0f, 0.0f, 0.0f), nrd = nro;
    auto fetchRay = [&]() {
        nro = ldStream(&st.rayO[pSlot]);
        nrd = ldStream(&st.rayD[pSlot]);
        nflags = ldStream(reinterpret_cast<const uint32_t *>(&st.tp[pSlot]) + 3);
    };
    pre = claim(true, pSlot);
    if (pre) fetchRay();
    haveNext = claim(pre, nvs);
    bool active = false, fin = false, idle = false; // idle: the entry's slot holds no live path (no hit record, class 0xFD)
    uint32_t bufCount = 0; // wave-uniform
    TravState r;
    r.node = 0; r.sp = 0; r.bestT = 0.0f; r.bestU = 0.0f; r.bestV = 0.0f; r.bestPrim = PRIM_MISS;
    while (true) {
        if (active) {
            if (travStep<false, COUNT>(sc, stack, r, tc) != TRAV_MORE) {
                active = false;
                fin = true;
            }
        }
        const bool waiting = !active && (fin || pre);
        const unsigned long long w = __ballot(waiting), act = __ballot(active);
        if (w == 0 && act == 0) break;
        if (act != 0 && __builtin_popcountll(w) < TG_EXTP_THRESH) continue;
        // finished rays: hit record out, (slot, prim) into the wave's buffer
        const bool retire = waiting && fin;
        const unsigned long long rm = __ballot(retire);
        if (retire) {
            if (!idle) stStream(&st.hit[cs], make_float4(r.bestT, r.bestU, r.bestV, __uint_as_float(r.bestPrim)));
            buf[bufCount + uint32_t(__builtin_popcountll(rm & below))] = make_uint2(cs, idle ? PRIM_IDLE : r.bestPrim);
            fin = false;
        }
        bufCount += uint32_t(__builtin_popcountll(rm));
        bool wantEntry = false;
        if (waiting && pre) { // arm the ray whose record was fetched ahead
            cs = pSlot;
            idle = flagsEnd(nflags) == END_DEAD;
            if (!idle) {
                const V3 O = xyz(nro), D = xyz(nrd);
                const float tnear = nro.w;
                float tfar = nrd.w, u = 0.0f, v = 0.0f;
                uint32_t prim = PRIM_MISS;
                for (uint32_t i = 0; i < sc.nQuads; ++i) { // closestHit's order: quads, cubes, triangles
                    QuadHit qh;
                    if (quadIntersect(sc.quads[i], O, D, tnear, tfar, qh)) {
                        tfar = qh.t; u = qh.u; v = qh.v;
                        prim = mkPrim(KIND_QUAD, i);
                    }
                }
                for (uint32_t i = 0; i < sc.nCubes; ++i) {
                    float t;
                    bool back;
                    if (cubeIntersect(sc.cubes[i], O, D, tnear, tfar, t, back)) {
                        tfar = t; u = 0.0f; v = 0.0f;
                        prim = mkPrim(KIND_CUBE, i);
                    }
                }
                travInit(r, sc.rootChild, O, D, tnear, tfar, prim);
                r.bestU = u; r.bestV = v;
                if (sc.nTris) active = true;
                else fin = true;
            } else {
                fin = true;
            }
            if (haveNext) {
                pSlot = nvs;
                haveNext = false;
                wantEntry = true;
                fetchRay();
            } else {
                pre = false;
            }
        }
        uint32_t slot = 0;
        const bool got = claim(wantEntry, slot);
        if (wantEntry) {
            haveNext = got;
            nvs = slot;
        }
        if (bufCount >= 256u) { // wave-uniform
            flush(256u);
            const uint32_t rem = bufCount - 256u; // < 64
            uint2 e = make_uint2(0u, 0u);
            if (uint32_t(lane) < rem) e = buf[256u + uint32_t(lane)];
            if (uint32_t(lane) < rem) buf[lane] = e;
            bufCount = rem;
        }
    }
    if (bufCount) flush(bufCount);
    if (COUNT) flushCounters(st.stats, 0, tc);
}
#endif

// =================================================================================================
// K5 miss (runs inside k_retire_regenerate): TraceBase::handleInfiniteLights (TraceBase.cpp:570-578) + end of traceSample
// =================================================================================================
// Returns the path's end state with its radiance and throughput.sum() (what the retire step needs), after writing them back.
TG_DEV int missPathEx(const DScene &sc, const DState &st, uint32_t path, V3 &emOut, float &tpSumOut)
{
    float4 tpw = st.tp[path];
    uint32_t flags = __float_as_uint(tpw.w);
    float4 emw = st.em[path];
    V3 tp = xyz(tpw), em = xyz(emw);
    if (flagsEnd(flags) == END_ALIVE && stateNee(st.state[path])) { // NEE terms of the path's last vertex (k_shadow_q)
        em = resolveNeeTerms(st, path, em);
        st.em[path] = make_float4(em.x, em.y, em.z, emw.w);
    }
    emOut = em;
    tpSumOut = tp.sum();
    if (flagsEnd(flags) != END_ALIVE) return flagsEnd(flags); // slot without a live path (refilled later)
    int bounce = flagsBounce(flags);
    bool wasSpecular = flagsSpecular(flags);
    int end = END_CHECKED;
    // per-bounce NaN guard of the previous vertex (PathTracer.cpp:121-122), evaluated once its NEE is in
    if (bounce > 0 && isnanf_(tp.sum() + em.sum())) {
        end = END_ZERO;
    } else if (bounce >= sc.settings.min_bounces && bounce < sc.settings.max_bounces && sc.env.present) {
        bool nee = sc.settings.enable_light_sampling != 0;
        if (!nee || wasSpecular || !sc.env.samplable) {
            V3 D = xyz(st.rayD[path]);
            V3 E;
            if (sc.envConstant) {
                E = V3(sc.textures[sc.env.emission_tex].value);
            } else {
                float ux, uy;
                directionToUV(sc, D, ux, uy, nullptr);
                E = texEval(sc, sc.env.emission_tex, ux, uy);
            }
            em += tp*E;
            st.em[path] = make_float4(em.x, em.y, em.z, emw.w);
            emOut = em;
        }
    }
    st.tp[path] = mk4u(tp, packFlags(bounce, wasSpecular, end));
    st.state[path] = packState(false, end);
    st.tpSum[path] = tp.sum();
    return end;
}
TG_DEV void missPath(const DScene &sc, const DState &st, uint32_t path)
{
    V3 em;
    float tpSum;
    (void)missPathEx(sc, st, path, em, tpSum);
}

// The auxiliary values of a sample whose path ended without recording them at a surface.  epilogue: the path left traceSample's
// loop through its end (PathTracer.cpp:133-140: depth 0 for a primary ray that escaped, the direction it left in as the normal, the
// environment's radiance as the albedo when handleInfiniteLights looked it up, TraceBase.cpp:570-578); otherwise it returned early
// and adds nothing.  Either way the running hitDistance in aov0.x is no depth sample.
TG_DEV void aovFinish(const DScene &sc, const DState &st, uint32_t path, uint32_t ref, bool epilogue)
{
    if (__float_as_uint(st.results[ref].w) & AOV_RECORDED) return;
    const float nan = __uint_as_float(0x7FC00000u);
    float4 a0 = make_float4(nan, nan, nan, nan);
    if (epilogue) {
        const int bounce = flagsBounce(__float_as_uint(st.tp[path].w));
        const V3 D = xyz(st.rayD[path]);
        if (bounce == 0) a0.x = 0.0f;
        a0.y = -D.x; a0.z = -D.y; a0.w = -D.z;
        if (bounce >= sc.settings.min_bounces && bounce < sc.settings.max_bounces && sc.env.present) {
            V3 E;
            if (sc.envConstant) {
                E = V3(sc.textures[sc.env.emission_tex].value);
            } else {
                float ux, uy;
                directionToUV(sc, D, ux, uy, nullptr);
                E = texEval(sc, sc.env.emission_tex, ux, uy);
            }
            st.aov1[ref] = make_float4(E.x, E.y, E.z, nan);
        }
    }
    st.aov0[ref] = a0;
}


// =================================================================================================
// K3 shade
// =================================================================================================
struct HitInfo { V3 Ng, Ns, p; float uvx, uvy; int bsdf; bool emissive; int emissionTex; bool backSide; int intMedium, extMedium; };

TG_DEV void fetchHitInfo(const DScene &sc, uint32_t prim, const V3 &O, const V3 &D, float t, float u, float v, HitInfo &info)
{
    uint32_t kind = primKind(prim), idx = primIndex(prim);
    info.emissive = false;
    info.emissionTex = -1;
    info.backSide = false;
    info.intMedium = info.extMedium = -1;
    if (kind == KIND_TRI) { // primitives/TriangleMesh.cpp:344-355, :81-106
        const float4 *s = sc.triShade + size_t(idx)*5;
        float4 s0 = s[0], s1 = s[1], s2 = s[2], s3 = s[3], s4 = s[4];
        info.p = O + D*t;
        info.Ng = V3(s4.x, s4.y, s4.z).normalized();
        uint32_t bits = __float_as_uint(s3.w);
        if (bits >> 31) {
            V3 n0(s0.x, s0.y, s0.z), n1(s0.w, s1.x, s1.y), n2(s1.z, s1.w, s2.x);
            info.Ns = ((1.0f - u - v)*n0 + u*n1 + v*n2).normalized();
        } else {
            info.Ns = info.Ng;
        }
        float w0 = 1.0f - u - v;
        info.uvx = w0*s2.y + u*s2.w + v*s3.y;
        info.uvy = w0*s2.z + u*s3.x + v*s3.z;
        info.bsdf = int(bits & 0x7FFFFFFFu);
        const uint32_t mb = __float_as_uint(s4.w); // (int_medium + 1) << 16 | (ext_medium + 1)
        info.intMedium = int(mb >> 16) - 1;
        info.extMedium = int(mb & 0xFFFFu) - 1;
    } else if (kind == KIND_QUAD) { // primitives/Quad.cpp:71-98,123-131
        const tgpu_quad &q = sc.quads[idx];
        info.p = O + t*D;
        info.Ng = info.Ns = V3(q.normal);
        info.uvx = u;
        info.uvy = v;
        info.bsdf = q.bsdf;
        info.emissive = q.emission_tex >= 0;
        info.emissionTex = q.emission_tex;
        info.backSide = D.dot(V3(q.normal)) >= 0.0f;
        info.intMedium = q.int_medium;
        info.extMedium = q.ext_medium;
    } else if (kind == KIND_SPHERE) { // primitives/Sphere.cpp:119-128
        const tgpu_sphere &sp = sc.spheres[idx];
        info.p = O + D*t;
        info.Ns = info.Ng = (info.p - V3(sp.pos))/sp.radius;
        sphereUv(sp, info.Ng, info.uvx, info.uvy);
        info.bsdf = sp.bsdf;
        info.emissive = sp.emission_tex >= 0;
        info.emissionTex = sp.emission_tex;
        { // SphereIntersection::backSide: which root was taken (Sphere.cpp:74-90)
            V3 pp = O - V3(sp.pos);
            float B = pp.dot(D);
            float C = pp.lengthSq() - sp.radius*sp.radius;
            float det = __builtin_sqrtf(B*B - C);
            info.backSide = t != -B - det;
        }
        info.intMedium = sp.int_medium;
        info.extMedium = sp.ext_medium;
    } else if (kind == KIND_DISK) { // primitives/Disk.cpp:64-86,107-120
        const tgpu_disk &dk = sc.disks[idx];
        info.p = O + t*D;
        info.Ng = info.Ns = V3(dk.n);
        diskUv(dk, info.p, u, info.uvx, info.uvy); // u carries rSq
        info.bsdf = dk.bsdf;
        info.emissive = dk.emission_tex >= 0;
        info.emissionTex = dk.emission_tex;
        info.backSide = -D.dot(V3(dk.n)) < dk.cos_apex;
        info.intMedium = dk.int_medium;
        info.extMedium = dk.ext_medium;
    } else { // primitives/Cube.cpp:157-170
        const tgpu_cube &cb = sc.cubes[idx];
        info.p = O + D*t;
        V3 p = mulPoint0(cb.inv_rot, info.p - V3(cb.pos));
        V3 a = vabs(p) - V3(cb.scale);
        int dim = 0;
        float am = a.x;
        if (a.y > am) { dim = 1; am = a.y; }
        if (a.z > am) { dim = 2; }
        float pd = p.comp(dim);
        V3 n(0.0f);
        float sgn = pd < 0.0f ? -1.0f : 1.0f;
        if (dim == 0) n.x = sgn; else if (dim == 1) n.y = sgn; else n.z = sgn;
        V3 uvw = (p/V3(cb.scale))*0.5f + 0.5f;
        info.Ns = info.Ng = mulPoint0(cb.rot, n);
        info.uvx = uvw.comp((dim + 1) % 3);
        info.uvy = uvw.comp((dim + 2) % 3);
        info.bsdf = cb.bsdf;
        info.intMedium = cb.int_medium;
        info.extMedium = cb.ext_medium;
    }
}

// Instance::intersectionInfo (primitives/Instance.cpp:335-344) on top of the master's, as the reference does it: the
// master was handed the WORLD hit point (TraceableScene.hpp:184-187), and info.p is then moved by the instance
// transform once more.  Only a Quad master replaces info.p by its own (instance-local) hit point first
// (Quad.cpp:123-131), so only quads end up where the ray actually hit.
TG_DEV void applyInstance(const DScene &sc, uint32_t inst, uint32_t prim, const V3 &O, const V3 &D, float t, HitInfo &info)
{
    const DInstSet &is = sc.instSets[inst >> 24];
    const uint32_t id = inst & 0xFFFFFFu;
    const float4 pq = is.pos[id], rq = is.rot[id];
    const V3 pos(pq.x, pq.y, pq.z);
    if (primKind(prim) == KIND_QUAD) {
        const float4 inv = make_float4(rq.x, -rq.y, -rq.z, -rq.w);
        const V3 lo = quatRotate(inv, O - pos), ld = quatRotate(inv, D);
        info.p = lo + t*ld;
    }
    info.Ng = quatRotate(rq, info.Ng);
    info.Ns = quatRotate(rq, info.Ns);
    info.p = pos + quatRotate(rq, info.p);
    info.emissive = false; // the hit primitive is the Instance, whose masters' emission is never consulted
    info.emissionTex = -1;
    info.intMedium = info.extMedium = -1; // ... nor their media
}

// light.directPdf for the BSDF-sampled NEE ray (InfiniteSphere.cpp:218-229, Quad.cpp:216-223)
TG_DEV float envDirectPdf(const DScene &sc, const V3 &w, float &ux, float &uy)
{
    float sinTheta;
    directionToUV(sc, w, ux, uy, &sinTheta);
    if (sc.envConstant) return INV_FOUR_PI;
    return INV_PI*INV_TWO_PI*envPdf(sc, ux, uy)/sinTheta;
}

TG_DEV float powerHeuristic(float pdf0, float pdf1) { return (pdf0*pdf0)/(pdf0*pdf0 + pdf1*pdf1); } // SampleWarp.hpp:189-192

// Quad::approximateRadiance (Quad.cpp:256-279); env returns "unknown" (-1), see chooseLight
// ---- lights: environment, quad, sphere, disk -----------------------------------------------------------
TG_DEV uint32_t lightPrim(const tgpu_light &l)
{
    const uint32_t kind = l.kind == TGPU_LIGHT_ENV ? KIND_ENV : (l.kind == TGPU_LIGHT_SPHERE ? KIND_SPHERE : (l.kind == TGPU_LIGHT_DISK ? KIND_DISK : KIND_QUAD));
    return mkPrim(kind, l.kind == TGPU_LIGHT_ENV ? 0u : uint32_t(l.index));
}
TG_DEV bool lightSampleDirect(const DScene &sc, const tgpu_light &l, const V3 &p, Sampler &smp, LightSample &ls)
{
    if (l.kind == TGPU_LIGHT_ENV) return envSampleDirect(sc, smp, ls);
    if (l.kind == TGPU_LIGHT_SPHERE) return sphereSampleDirect(sc.spheres[l.index], p, smp, ls);
    if (l.kind == TGPU_LIGHT_DISK) return diskSampleDirect(sc.disks[l.index], p, smp, ls);
    return quadSampleDirect(sc.quads[l.index], p, smp, ls);
}
// attenuatedEmission (TraceBase.cpp:144-174) for a finite light, up to the shadow test: light.intersect(ray) with the
// 1.001 fudge against the expected distance (none when < 0), evalDirect at the hit, directPdf seen from the ray origin
TG_DEV bool finiteLightHit(const DScene &sc, const tgpu_light &l, const V3 &O, const V3 &D, float nearT, float expectedDist, float &farT, V3 &E,
                           float &pdf)
{
    if (l.kind == TGPU_LIGHT_SPHERE) {
        const tgpu_sphere &sp = sc.spheres[l.index];
        float t;
        bool back;
        if (!sphereIntersect(sp, O, D, nearT, TG_INF, t, back)) return false;
        if (t*(1.0f + 1e-3f) < expectedDist) return false;
        farT = t;
        if (back) {
            E = V3(0.0f);
        } else {
            V3 Ng = ((O + D*t) - V3(sp.pos))/sp.radius;
            float u, v;
            sphereUv(sp, Ng, u, v);
            E = texEval(sc, sp.emission_tex, u, v);
        }
        pdf = sphereDirectPdf(sp, O);
        return true;
    }
    if (l.kind == TGPU_LIGHT_DISK) {
        const tgpu_disk &dk = sc.disks[l.index];
        DiskHit h;
        if (!diskIntersect(dk, O, D, nearT, TG_INF, h)) return false;
        if (h.t*(1.0f + 1e-3f) < expectedDist) return false;
        farT = h.t;
        if (h.backSide) {
            E = V3(0.0f);
        } else {
            float u, v;
            diskUv(dk, h.p, h.rSq, u, v);
            E = texEval(sc, dk.emission_tex, u, v);
        }
        V3 n(dk.n); // Disk::directPdf :225-232
        float cosTheta = __builtin_fabsf(n.dot(D));
        float td = n.dot(V3(dk.center) - O)/n.dot(D);
        pdf = td*td/(cosTheta*dk.r*dk.r*PI);
        return true;
    }
    const tgpu_quad &lq = sc.quads[l.index];
    QuadHit qh;
    if (!quadIntersect(lq, O, D, nearT, TG_INF, qh)) return false;
    if (qh.t*(1.0f + 1e-3f) < expectedDist) return false;
    farT = qh.t;
    E = qh.backSide ? V3(0.0f) : texEval(sc, lq.emission_tex, qh.u, qh.v);
    V3 normal(lq.normal); // Quad::directPdf :216-223
    float cosTheta = __builtin_fabsf(normal.dot(D));
    float tq = normal.dot(V3(lq.base) - O)/normal.dot(D);
    pdf = tq*tq/(cosTheta*lq.area);
    return true;
}
TG_DEV float texAverageMax(const DScene &sc, int tex) // Texture::average().max()
{
    const tgpu_texture &t = sc.textures[tex];
    V3 a;
    if (t.type == TGPU_TEX_CONSTANT) a = V3(t.value);
    else if (t.type == TGPU_TEX_CHECKER) a = (V3(t.on_color) + V3(t.off_color))*0.5f;
    else a = sc.bitmaps[t.bitmap].scale*V3(sc.bitmaps[t.bitmap].average);
    return a.max();
}

template <bool LEAN>
TG_DEV float lightApproximateRadiance(const DScene &sc, const tgpu_light &l, const V3 &p)
{
    if (LEAN) __builtin_assume(l.kind == TGPU_LIGHT_ENV || l.kind == TGPU_LIGHT_QUAD);
    if (l.kind == TGPU_LIGHT_ENV) return TWO_PI*texAverageMax(sc, sc.env.emission_tex); // InfiniteSphere.cpp:261-266
    if (l.kind == TGPU_LIGHT_SPHERE) { // Sphere.cpp:33-40,266-271
        const tgpu_sphere &sp = sc.spheres[l.index];
        V3 L = V3(sp.pos) - p;
        float d = L.length();
        float cosTheta = __builtin_sqrtf(fmax2(d*d - sp.radius*sp.radius, 0.0f))/d;
        return (TWO_PI*(1.0f - cosTheta))*texAverageMax(sc, sp.emission_tex);
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
        float Q = tgpu_acosf(n0.dot(n1)) + tgpu_acosf(n1.dot(n2)) + tgpu_acosf(n2.dot(n3)) + tgpu_acosf(n3.dot(n0));
        return (TWO_PI - __builtin_fabsf(Q))*texAverageMax(sc, dk.emission_tex);
    }
    const tgpu_quad &q = sc.quads[l.index]; // Quad.cpp:256-279
    V3 R0 = V3(q.base) - p;
    if (R0.dot(V3(q.normal)) >= 0.0f) return 0.0f;
    V3 R1 = R0 + V3(q.edge0), R2 = R1 + V3(q.edge1), R3 = R0 + V3(q.edge1);
    V3 n0 = R0.cross(R1).normalized(), n1 = R1.cross(R2).normalized(), n2 = R2.cross(R3).normalized(), n3 = R3.cross(R0).normalized();
    float Q = tgpu_acosf(n0.dot(n1)) + tgpu_acosf(n1.dot(n2)) + tgpu_acosf(n2.dot(n3)) + tgpu_acosf(n3.dot(n0));
    return (TWO_PI - __builtin_fabsf(Q))*texAverageMax(sc, q.emission_tex);
}

// TraceBase::chooseLight (TraceBase.cpp:416-459); up to TG_MAX_LIGHTS samplable lights
#define TG_MAX_LIGHTS 8
template <bool LEAN>
TG_DEV int chooseLight(const DScene &sc, const V3 &p, Sampler &smp, float &weight)
{
    if (sc.nLights == 0) return -1;
    if (sc.nLights == 1) {
        weight = 1.0f;
        return 0;
    }
    float lightPdf[TG_MAX_LIGHTS];
    const uint32_t n = sc.nLights < uint32_t(TG_MAX_LIGHTS) ? sc.nLights : uint32_t(TG_MAX_LIGHTS);
    float total = 0.0f;
    uint32_t numNonNegative = 0;
#pragma unroll
    for (uint32_t i = 0; i < TG_MAX_LIGHTS; ++i) {
        lightPdf[i] = 0.0f;
        if (i < n) {
            lightPdf[i] = lightApproximateRadiance<LEAN>(sc, sc.lights[i], p);
            if (lightPdf[i] >= 0.0f) {
                total += lightPdf[i];
                numNonNegative++;
            }
        }
    }
    if (numNonNegative == 0) {
#pragma unroll
        for (uint32_t i = 0; i < TG_MAX_LIGHTS; ++i) lightPdf[i] = 1.0f;
        total = float(n);
    } else if (numNonNegative < n) {
#pragma unroll
        for (uint32_t i = 0; i < TG_MAX_LIGHTS; ++i) {
            if (i < n) {
                float uniformWeight = (total == 0.0f ? 1.0f : total)/numNonNegative;
                if (lightPdf[i] < 0.0f) {
                    lightPdf[i] = uniformWeight;
                    total += uniformWeight;
                }
            }
        }
    }
    if (total == 0.0f) return -1;
    float t = smp.next1D()*total;
    int result = -1;
#pragma unroll
    for (uint32_t i = 0; i < TG_MAX_LIGHTS; ++i) {
        if (i < n && result < 0) {
            if (t < lightPdf[i] || i == n - 1) {
                weight = total/lightPdf[i];
                result = int(i);
            } else {
                t -= lightPdf[i];
            }
        }
    }
    return result;
}

// ---- participating media ----------------------------------------------------------------------------
// FastMath::exp(Vec3f) = fmath::exp_ps on (x, y, z, 0) (math/FastMath.hpp:23-27, thirdparty/fmath/fmath.hpp:320-363):
// clamp to +-88 (all lanes, once any lane is out of range), r = round-to-nearest-even(x*1024/ln 2),
// 2^(r/1024) from a 1024-entry mantissa table and the exponent bits, times the first-order term 1 + (x - r*ln2/1024)
TG_DEV V3 fastExp(const DScene &sc, const V3 &v)
{
    float x[3] = {v.x, v.y, v.z};
    bool over = false;
#pragma unroll
    for (int i = 0; i < 3; ++i)
        if (int32_t(__float_as_uint(x[i]) & 0x7FFFFFFFu) > int32_t(0x42B00000)) over = true;
    float out[3];
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        float xi = x[i];
        if (over) {
            xi = xi < 88.0f ? xi : 88.0f;   // _mm_min_ps(x, maxX)
            xi = xi > -88.0f ? xi : -88.0f; // _mm_max_ps(x, minX)
        }
        int32_t r = __float2int_rn(xi*sc.expA);
        float t = xi - float(r)*sc.expB;
        t = t + 1.0f;
        uint32_t u = ((uint32_t(r) + (127u << 10)) >> 10) << 23;
        out[i] = t*__uint_as_float(sc.expTable[uint32_t(r) & 1023u] | u);
    }
    return V3(out[0], out[1], out[2]);
}

// ---- transmittance models (transmittances/*.cpp).  form: 0 surfaceSurface, 1 surfaceMedium, 2 mediumSurface,
// 3 mediumMedium; Transmittance.hpp:22-42 picks the form from where the segment starts and ends
TG_DEV V3 vmin1(const V3 &a) { return V3(fmin2(a.x, 1.0f), fmin2(a.y, 1.0f), fmin2(a.z, 1.0f)); }
// one model: type + parameters (the medium-level transForm / transSigmaBar / transSample below add InterpolatedTransmittance on top)
TG_DEV V3 powV3(const V3 &a, float e) { return V3(tgpu_powf(a.x, e), tgpu_powf(a.y, e), tgpu_powf(a.z, e)); }
TG_DEV V3 modelForm(const DScene &sc, int type, const float *param, int form, const V3 &tau)
{
    struct { const float *param; } m{param};
    switch (type) {
    case TGPU_TRANS_ERLANG: { // ErlangTransmittance.cpp:32-47
        const float lambda = param[0];
        if (form == 0) return 0.5f*expV3(-lambda*tau)*(V3(2.0f) + lambda*tau);
        V3 ms = expV3(-lambda*tau)*(V3(1.0f) + lambda*tau);
        if (form == 1) return ms*lambda*0.5f;
        if (form == 2) return ms;
        return (lambda*lambda)*tau*expV3(-lambda*tau);
    }
    case TGPU_TRANS_DAVIS: { // DavisTransmittance.cpp:31-46
        const float alpha = param[0];
        if (form == 0) return powV3(V3(1.0f) + tau/alpha, -alpha);
        if (form == 1 || form == 2) return powV3(V3(1.0f) + tau/alpha, -(alpha + 1.0f));
        return (1.0f + 1.0f/alpha)*powV3(V3(1.0f) + tau/alpha, -(alpha + 2.0f));
    }
    case TGPU_TRANS_DAVIS_WEINSTEIN: { // DavisWeinsteinTransmittance.cpp:28-80: the first channel only
        const float h = param[0], c = param[1];
        const float beta = 2.0f*h - 1.0f;
        const float t = tau.x;
        const float alpha = tgpu_powf(t, 1 - beta)/(tgpu_powf(c, 1 + beta)); // computeAlpha
        const float base = 1.0f + t/alpha;
        float Tr;
        if (form == 0) {
            Tr = tgpu_powf(base, -alpha);
        } else if (form == 1 || form == 2) {
            float trSurface = tgpu_powf(base, -alpha);
            Tr = trSurface*(beta/base - (beta - 1.0f)*alpha/t*tgpu_logf(base));
        } else {
            float logBase = tgpu_logf(base);
            float trSurface = tgpu_powf(base, -alpha);
            float term1 = beta*(-1.0f + beta*(1.0f + t) + (-1.0f + 2.0f*beta)*t/alpha)/(t*base*base);
            float term2 = ((-1.0f + beta)*beta*alpha/(t*t)*(2.0f*t + base)*logBase)/base;
            float term3 = (beta - 1.0f)*alpha/t*logBase;
            Tr = trSurface*(term1 - term2 + term3*term3);
        }
        return V3(isnanf_(Tr) ? 0.0f : Tr);
    }
    default: break;
    }
    switch (type) {
    case TGPU_TRANS_LINEAR: { // LinearTransmittance.cpp:30-59
        const float maxT = m.param[0];
        if (form == 0) return V3(1.0f) - vmin1(tau/maxT);
        float r[3], tv[3] = {tau.x, tau.y, tau.z};
#pragma unroll
        for (int i = 0; i < 3; ++i) {
            if (form == 1) r[i] = tv[i] > maxT ? 0.0f : 1.0f/maxT;
            else if (form == 2) r[i] = tv[i] > maxT ? 0.0f : 1.0f;
            else r[i] = __builtin_fabsf(tv[i] - maxT) < 1e-3f ? 1.0f : 0.0f;
        }
        return V3(r[0], r[1], r[2]);
    }
    case TGPU_TRANS_QUADRATIC: { // QuadraticTransmittance.cpp:30-51
        const float maxT = m.param[0];
        if (form == 0) { V3 t = vmin1(tau/maxT); return V3(1.0f) - 2.0f*t + t*t; }
        if (form == 1) return (2.0f/maxT)*(V3(1.0f) - vmin1(tau/maxT));
        if (form == 2) return V3(1.0f) - vmin1(tau/maxT);
        return V3(tau.x > maxT ? 0.0f : 1.0f/maxT, tau.y > maxT ? 0.0f : 1.0f/maxT, tau.z > maxT ? 0.0f : 1.0f/maxT);
    }
    case TGPU_TRANS_DOUBLE_EXPONENTIAL: { // DoubleExponentialTransmittance.cpp:34-49
        const float sa = m.param[0], sb = m.param[1];
        V3 ea = expV3(-sa*tau), eb = expV3(-sb*tau);
        if (form == 0) return 0.5f*(ea + eb);
        if (form == 1) return 0.5f*(sa*ea + sb*eb);
        if (form == 2) return (sa*ea + sb*eb)/(sa + sb);
        return ((sa*sa)*ea + (sb*sb)*eb)/(sa + sb);
    }
    case TGPU_TRANS_PULSE: { // PulseTransmittance.cpp:43-81
        const float a = m.param[0], b = m.param[1];
        const int numPulses = int(m.param[2]);
        float tv[3] = {tau.x, tau.y, tau.z}, res[3];
        if (form == 0) {
#pragma unroll
            for (int i = 0; i < 3; ++i) {
                float f = clampf((float(numPulses)*(tv[i] - a))/(b - a) + 0.5f, 0.0f, float(numPulses));
                int idx = int(f);
                float height = float(numPulses - idx)/float(numPulses);
                float cellIntegral = height*(f - float(idx));
                if (idx > 0) cellIntegral += (idx - 0.5f) - (idx*(idx - 1))/(2.0f*numPulses);
                else cellIntegral -= 0.5f;
                res[i] = 1.0f - (2.0f/numPulses)*cellIntegral;
            }
            return V3(res[0], res[1], res[2]);
        }
        if (form == 1 || form == 2) {
#pragma unroll
            for (int i = 0; i < 3; ++i) {
                int idx = int((float(numPulses)*(tv[i] - a))/(b - a) + 0.5f);
                idx = idx < 0 ? 0 : (idx > numPulses ? numPulses : idx);
                res[i] = 1.0f - float(idx)/float(numPulses);
            }
            V3 ms(res[0], res[1], res[2]);
            return form == 1 ? 2.0f/(b - a)*ms : ms;
        }
#pragma unroll
        for (int i = 0; i < 3; ++i) {
            float f = clampf((float(numPulses)*(tv[i] - a))/(b - a), 0.0f, float(numPulses));
            int idx = int(f);
            res[i] = __builtin_fabsf(f - idx - 0.5f) < 1e-3f ? 1.0f : 0.0f;
        }
        return (1.0f/numPulses)*V3(res[0], res[1], res[2]);
    }
    default: return fastExp(sc, -tau); // ExponentialTransmittance.cpp:26-41: all four forms
    }
}
TG_DEV float modelSigmaBar(int type, const float *param)
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
TG_DEV float transSigmaBar(const tgpu_medium &m)
{
    if (m.transmittance != TGPU_TRANS_INTERPOLATED) return modelSigmaBar(m.transmittance, m.param);
    // InterpolatedTransmittance.cpp:73-76: 1/lerp(1/sigmaBarA, 1/sigmaBarB, u)
    const float a = 1.0f/modelSigmaBar(m.tr_a_type, m.tr_a_param), b = 1.0f/modelSigmaBar(m.tr_b_type, m.tr_b_param);
    return 1.0f/(a*(1.0f - m.tr_ratio) + b*m.tr_ratio);
}
TG_DEV V3 transForm(const DScene &sc, const tgpu_medium &m, int form, const V3 &tau)
{
    if (m.transmittance != TGPU_TRANS_INTERPOLATED) return modelForm(sc, m.transmittance, m.param, form, tau);
    // InterpolatedTransmittance.cpp:43-71
    const float u = m.tr_ratio;
    const float sbA = modelSigmaBar(m.tr_a_type, m.tr_a_param), sbB = modelSigmaBar(m.tr_b_type, m.tr_b_param);
    const float sigmaBar = transSigmaBar(m);
    if (form == 0) {
        const V3 a = modelForm(sc, m.tr_a_type, m.tr_a_param, 0, tau)/sbA, b = modelForm(sc, m.tr_b_type, m.tr_b_param, 0, tau)/sbB;
        return sigmaBar*(a*(1.0f - u) + b*u);
    }
    if (form == 1 || form == 2) {
        const V3 a = modelForm(sc, m.tr_a_type, m.tr_a_param, 2, tau), b = modelForm(sc, m.tr_b_type, m.tr_b_param, 2, tau);
        const V3 ms = a*(1.0f - u) + b*u;
        return form == 1 ? ms*sigmaBar : ms;
    }
    const V3 pdfA = modelForm(sc, m.tr_a_type, m.tr_a_param, 3, tau), pdfB = modelForm(sc, m.tr_b_type, m.tr_b_param, 3, tau);
    const bool dA = m.tr_a_type == TGPU_TRANS_LINEAR || m.tr_a_type == TGPU_TRANS_PULSE; // Transmittance::isDirac
    const bool dB = m.tr_b_type == TGPU_TRANS_LINEAR || m.tr_b_type == TGPU_TRANS_PULSE;
    const float pa[3] = {pdfA.x, pdfA.y, pdfA.z}, pb[3] = {pdfB.x, pdfB.y, pdfB.z};
    float r[3];
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        const bool diracA = dA && pa[i] > 0.0f, diracB = dB && pb[i] > 0.0f;
        if (diracA != diracB) r[i] = diracA ? pa[i] : pb[i];
        else r[i] = pa[i]*(1.0f - u) + pb[i]*u;
    }
    return V3(r[0], r[1], r[2]);
}
TG_DEV V3 transEval(const DScene &sc, const tgpu_medium &m, const V3 &tau, bool startOnSurface, bool endOnSurface)
{   // Transmittance::eval (Transmittance.hpp:22-30)
    if (startOnSurface && endOnSurface) return transForm(sc, m, 0, tau);
    if (!startOnSurface && !endOnSurface) return transForm(sc, m, 3, tau)/transSigmaBar(m);
    return transForm(sc, m, 2, tau);
}
TG_DEV float modelSample(const DScene &sc, int type, const float *param, Sampler &smp, bool startOnSurface)
{   // sampleSurface / sampleMedium of the model
    struct { const float *param; } m{param};
    switch (type) {
    case TGPU_TRANS_ERLANG: { // ErlangTransmittance.cpp:54-67
        const float lambda = param[0];
        if (!startOnSurface) {
            const float xi0 = smp.next1D(), xi1 = smp.next1D();
            return -1.0f/lambda*tgpu_logf(xi0*xi1);
        }
        const float xi = smp.next1D();
        float x = 0.5f;
        for (int i = 0; i < 10; ++i) { // Newton iterations on the CDF 1 - surfaceSurface
            x += (xi - (1.0f - modelForm(sc, type, param, 0, V3(x)).x))/modelForm(sc, type, param, 1, V3(x)).x;
            x = fmax2(x, 0.0f);
        }
        return x;
    }
    case TGPU_TRANS_DAVIS: { // DavisTransmittance.cpp:53-60
        const float alpha = param[0];
        return startOnSurface ? alpha*(tgpu_powf(1.0f - smp.next1D(), -1.0f/alpha) - 1.0f)
                              : alpha*(tgpu_powf(1.0f - smp.next1D(), -1.0f/(1.0f + alpha)) - 1.0f);
    }
    case TGPU_TRANS_DAVIS_WEINSTEIN: { // DavisWeinsteinTransmittance.cpp:87-117: bisection on the CDF
        const float xi = smp.next1D();
        float step = 1e6;
        float result = step*2;
        while (double(step) > 1e-6) {
            const float cdf = 1.0f - modelForm(sc, type, param, startOnSurface ? 0 : 2, V3(result)).x;
            if (cdf > xi) result -= step;
            else result += step;
            step /= 2;
        }
        return result;
    }
    default: break;
    }
    switch (type) {
    case TGPU_TRANS_LINEAR: return startOnSurface ? m.param[0]*smp.next1D() : m.param[0];
    case TGPU_TRANS_QUADRATIC:
        return startOnSurface ? m.param[0]*(1.0f - __builtin_sqrtf(1.0f - smp.next1D())) : m.param[0]*smp.next1D();
    case TGPU_TRANS_DOUBLE_EXPONENTIAL: {
        float t = -tgpu_logf(1.0f - smp.next1D());
        float p = startOnSurface ? 0.5f : m.param[0]/(m.param[0] + m.param[1]);
        return smp.nextBoolean(p) ? t/m.param[0] : t/m.param[1];
    }
    case TGPU_TRANS_PULSE: {
        const float a = m.param[0], b = m.param[1];
        const int numPulses = int(m.param[2]);
        if (!startOnSurface) return a + (0.5f + int(smp.next1D()*numPulses))/float(numPulses)*(b - a);
        float xi = smp.next1D()*numPulses*0.5f;
        float delta = 1.0f/numPulses;
        for (int i = 0; i < numPulses; ++i) {
            float h0 = 1.0f - (i + 0.0f)*delta;
            float h1 = 1.0f - (i + 1.0f)*delta;
            xi -= h0*0.5f;
            if (xi < 0.0f) return a + (i + 0.0f + 0.5f*smp.next1D())*(b - a)*delta;
            xi -= h1*0.5f;
            if (xi < 0.0f) return a + (i + 0.5f + 0.5f*smp.next1D())*(b - a)*delta;
        }
        return 0.0f;
    }
    default: return -tgpu_logf(1.0f - smp.next1D());
    }
}
TG_DEV float transSample(const DScene &sc, const tgpu_medium &m, Sampler &smp, bool startOnSurface)
{
    if (m.transmittance != TGPU_TRANS_INTERPOLATED) return modelSample(sc, m.transmittance, m.param, smp, startOnSurface);
    // InterpolatedTransmittance.cpp:78-85
    return smp.nextBoolean(m.tr_ratio) ? modelSample(sc, m.tr_b_type, m.tr_b_param, smp, startOnSurface)
                                       : modelSample(sc, m.tr_a_type, m.tr_a_param, smp, startOnSurface);
}

// HomogeneousMedium::transmittance (HomogeneousMedium.cpp:110-117) over a shadow segment that ends (on a surface) at
// farT; no medium: 1
TG_DEV V3 neeTransmittance(const DScene &sc, int medium, float farT, bool startOnSurface)
{
    if (medium < 0) return V3(1.0f);
    if (farT == TG_INF) return V3(0.0f);
    const tgpu_medium &m = sc.media[medium];
    return transEval(sc, m, V3(m.sigma_t)*farT, startOnSurface, true);
}

template <int CLASS, bool MEDIA, bool NESTED = true>
TG_DEV int shadePath(const DScene &sc, const DState &st, uint32_t path, bool afterMedium = false); // returns the path's end state
TG_DEV void missPath(const DScene &sc, const DState &st, uint32_t path);

// phase functions: IsotropicPhaseFunction.cpp:24-44, HenyeyGreensteinPhaseFunction.cpp:19-23,40-66 (eval == pdf, weight 1)
TG_DEV float phaseEval(const tgpu_medium &m, const V3 &wi, const V3 &wo)
{
    if (m.phase == 0) return INV_FOUR_PI;
    const float g = m.phase_g;
    float cosTheta = wi.dot(wo);
    float term = 1.0f + g*g - 2.0f*g*cosTheta;
    return INV_FOUR_PI*(1.0f - g*g)/(term*__builtin_sqrtf(term));
}
TG_DEV V3 phaseSample(const tgpu_medium &m, Sampler &smp, const V3 &wi, float &pdf)
{
    float xi0 = smp.next1D(), xi1 = smp.next1D();
    const float g = m.phase_g;
    float sp, cp;
    if (m.phase == 0 || g == 0.0f) {
        float phi = xi0*TWO_PI; // SampleWarp::uniformSphere
        float z = xi1*2.0f - 1.0f;
        float r = __builtin_sqrtf(fmax2(1.0f - z*z, 0.0f));
        tgpu_sincosf(phi, &sp, &cp);
        pdf = INV_FOUR_PI;
        return V3(cp*r, sp*r, z);
    }
    float phi = xi0*TWO_PI;
    float q = (1.0f - g*g)/(1.0f + g*(xi1*2.0f - 1.0f));
    float cosTheta = (1.0f + g*g - q*q)/(2.0f*g);
    float sinTheta = __builtin_sqrtf(fmax2(1.0f - cosTheta*cosTheta, 0.0f));
    tgpu_sincosf(phi, &sp, &cp);
    Frame f = duffFrame(wi);
    float term = 1.0f + g*g - 2.0f*g*cosTheta;
    pdf = INV_FOUR_PI*(1.0f - g*g)/(term*__builtin_sqrtf(term));
    return f.toGlobal(V3(cp*sinTheta, sp*sinTheta, cosTheta));
}

// Every shaded vertex ends here: the state word (NEE flag + end-state copy) and throughput.sum() for the retire step, and -- when
// the vertex queued no NEE rays -- negative far ends, which is all the shadow traversal looks at.
TG_DEV void closeVertex(const DState &st, uint32_t path, bool nee, int end, float tpSum)
{
    st.state[path] = packState(nee, end);
    st.tpSum[path] = tpSum;
    if (!nee) {
        reinterpret_cast<float *>(&st.shA[path])[3] = -1.0f;
        reinterpret_cast<float *>(&st.shB[path])[3] = -1.0f;
    }
}

// One step of a path whose ray runs inside a medium: PathTracer.cpp:54-63 (HomogeneousMedium::sampleDistance,
// HomogeneousMedium.cpp:65-108), then either the surface / miss code with the attenuated throughput or
// TraceBase::handleVolume (TraceBase.cpp:496-514) with volumeEstimateDirect (:323-381, :402-414, :470-481) and the loop
// tail of traceSample (PathTracer.cpp:108-126).  The NEE rays go through the ordinary shadow kernel: their
// transmittance is known here (the segment length is the light's distance) and folded into the contributions.
// NESTED = false (k_volume): a path that leaves the medium onto a BSDF made of other BSDFs is left for k_volume_nested (END_DEFERRED); the surface
// code compiled in here then has no out-of-line calls.
template <bool NESTED = true>
TG_DEV int volumePath(const DScene &sc, const DState &st, uint32_t path) // returns the path's end state
{
    const int maxBounces = sc.settings.max_bounces, minBounces = sc.settings.min_bounces;
    float4 tpw = st.tp[path];
    uint32_t flags = __float_as_uint(tpw.w);
    if (flagsEnd(flags) != END_ALIVE) { closeVertex(st, path, false, flagsEnd(flags), xyz(tpw).sum()); return flagsEnd(flags); }
    float4 ro = st.rayO[path], rd = st.rayD[path], hw = st.hit[path], emw = st.em[path];
    uint4 sw = st.samp[path];
    V3 O = xyz(ro), D = xyz(rd), tp = xyz(tpw), em = xyz(emw);
    if (stateNee(st.state[path])) { // NEE terms of the previous vertex (k_shadow_q left the visibility bits)
        em = resolveNeeTerms(st, path, em);
        st.em[path] = make_float4(em.x, em.y, em.z, emw.w);
        st.state[path] = packState(false, END_ALIVE);
    }
    int bounce = flagsBounce(flags);
    bool wasSpecular = flagsSpecular(flags);
    if (bounce > 0 && isnanf_(tp.sum() + em.sum())) { // PathTracer.cpp:121-122 for the previous vertex
        st.tp[path] = mk4u(tp, packFlags(bounce, wasSpecular, END_ZERO));
        closeVertex(st, path, false, END_ZERO, tp.sum());
        return END_ZERO;
    }
    Sampler smp;
    smp.matrices = sc.sobol;
    smp.sobol = sc.settings.use_sobol;
    smp.dimension = sw.x;
    smp.pcg = (uint64_t(sw.z) << 32) | uint64_t(sw.y);
    smp.scramble = __float_as_uint(emw.w);
    smp.index = sw.w;

    const uint32_t med = st.med[path];
    const int medium = int(med & 0xFFu) - 1;
    bool firstScatter = (med >> 8) & 1u;
    uint32_t stateBounce = med >> 9;
    const tgpu_medium &m = sc.media[medium];
    const V3 sigmaT(m.sigma_t), sigmaS(m.sigma_s);
    const bool didHit = __float_as_uint(hw.w) != PRIM_MISS;
    const float maxT = didHit ? hw.x : TG_INF;

    // ---- sampleDistance ----
    bool ok = true, exited = true;
    float t = maxT;
    V3 weight(1.0f);
    if (int(stateBounce) > m.max_bounce) {
        ok = false;
    } else if (m.absorption_only) {
        if (maxT == TG_INF) ok = false;
        else weight = transEval(sc, m, t*sigmaT, firstScatter, true);
    } else {
        int component = smp.nextDiscrete(3);
        float sigmaTc = component == 0 ? sigmaT.x : (component == 1 ? sigmaT.y : sigmaT.z);
        float ts = transSample(sc, m, smp, firstScatter)/sigmaTc;
        t = fmin2(ts, maxT);
        exited = ts >= maxT;
        V3 tau = t*sigmaT;
        weight = transEval(sc, m, tau, firstScatter, exited);
        float pdf;
        if (exited) {
            pdf = transForm(sc, m, firstScatter ? 0 : 2, tau).avg();          // surfaceProbability
        } else {
            pdf = (sigmaT*transForm(sc, m, firstScatter ? 1 : 3, tau)).avg(); // sigmaT*mediumPdf
            weight *= sigmaS*transSigmaBar(m);
        }
        weight /= pdf;
        firstScatter = false; // MediumState::advance
        stateBounce++;
    }
    if (!ok) { // `return emission`
        st.tp[path] = mk4u(tp, packFlags(bounce, wasSpecular, END_RAW));
        closeVertex(st, path, false, END_RAW, tp.sum());
        return END_RAW;
    }
    tp *= weight;
    if (exited) {
        st.tp[path] = mk4u(tp, flags);
        st.samp[path] = make_uint4(smp.dimension, uint32_t(smp.pcg), uint32_t(smp.pcg >> 32), sw.w);
        st.med[path] = uint32_t(medium + 1) | (firstScatter ? 1u << 8 : 0u) | (stateBounce << 9);
        if (!didHit) { // `break`: handleInfiniteLights and the final NaN check
            closeVertex(st, path, false, END_ALIVE, tp.sum());
            V3 emUnused;
            float tpSumUnused;
            return missPathEx(sc, st, path, emUnused, tpSumUnused);
        }
        if (!NESTED && hitBsdfType(sc, __float_as_uint(hw.w)) >= uint32_t(TGPU_BSDF_MIXED)) return END_DEFERRED;
        return shadePath<-1, true, NESTED>(sc, st, path, true);
    }

    // ---- handleVolume ----
    const bool volumeNee = sc.settings.enable_volume_light_sampling != 0;
    const bool fwdScene = sc.hasForward != 0;
    const int neeBounce = bounce + 1;
    const V3 p = O + t*D;
    wasSpecular = !volumeNee;
    V3 L(0.0f), B(0.0f);
    float tfarL = -1.0f, tfarB = -1.0f, pickWeight = 1.0f;
    V3 dL(0.0f), dB(0.0f);
    V3 fL(0.0f), EL(0.0f), EB(0.0f), wB(0.0f);
    float pdfL = 1.0f, misL = 0.0f, misB = 0.0f;
    uint32_t endCap = PRIM_MISS;
    const V3 tpVertex = tp;
    if (volumeNee && bounce < maxBounces - 1 && sc.nLights > 0) {
        int lightIdx = chooseLight<false>(sc, p, smp, pickWeight);
        if (lightIdx >= 0) {
            const tgpu_light light = sc.lights[lightIdx];
            const bool isEnv = light.kind == TGPU_LIGHT_ENV;
            endCap = lightPrim(light);
            const bool shadowPasses = bounce + 1 >= minBounces;
            { // volumeLightSample
                LightSample ls;
                bool okL = lightSampleDirect(sc, light, p, smp, ls);
                if (okL) {
                    V3 f(phaseEval(m, D, ls.d)); // phase->eval(parentRay.dir(), lightSample.d)
                    V3 E(0.0f);
                    float farT = TG_INF;
                    bool lit = true;
                    if (isEnv) {
                        float ux, uy;
                        if (sc.envConstant) E = V3(sc.textures[sc.env.emission_tex].value);
                        else { directionToUV(sc, ls.d, ux, uy, nullptr); E = texEval(sc, sc.env.emission_tex, ux, uy); }
                    } else {
                        float lightPdfUnused;
                        lit = finiteLightHit(sc, light, p, ls.d, 0.0f, ls.dist, farT, E, lightPdfUnused);
                    }
                    if (!fwdScene) E = neeTransmittance(sc, medium, farT, false)*E;
                    if (lit && shadowPasses && !E.isZero()) {
                        V3 lightF = f*E/ls.pdf;
                        const float mis = powerHeuristic(ls.pdf, phaseEval(m, D, ls.d)); // phase->pdf
                        lightF *= mis;
                        L = lightF;
                        dL = ls.d;
                        tfarL = farT;
                        fL = f; EL = E; pdfL = ls.pdf; misL = mis;
                    }
                }
            }
            { // volumePhaseSample (the lights in scope are never Dirac)
                float phasePdf;
                V3 w = phaseSample(m, smp, D, phasePdf);
                V3 E(0.0f);
                float farT = TG_INF, lightPdf = 0.0f;
                bool lit = true;
                if (isEnv) {
                    float ux, uy;
                    lightPdf = envDirectPdf(sc, w, ux, uy);
                    E = sc.envConstant ? V3(sc.textures[sc.env.emission_tex].value) : texEval(sc, sc.env.emission_tex, ux, uy);
                } else {
                    lit = finiteLightHit(sc, light, p, w, 0.0f, -1.0f, farT, E, lightPdf); // expectedDist = -1 never rejects
                }
                if (!fwdScene) E = neeTransmittance(sc, medium, farT, false)*E;
                if (lit && shadowPasses && !E.isZero()) {
                    V3 phaseF = E*V3(1.0f);
                    const float mis = powerHeuristic(phasePdf, lightPdf);
                    phaseF *= mis;
                    B = phaseF;
                    dB = w;
                    tfarB = farT;
                    EB = E; wB = V3(1.0f); misB = mis;
                }
            }
        }
    }
    const bool pushShadow = tfarL >= 0.0f || tfarB >= 0.0f;

    // phase->sample(sampler, ray.dir(), phaseSample) -> the continuation ray (nearT = 0)
    float contPdf;
    const V3 wo = phaseSample(m, smp, D, contPdf);
    tp *= V3(1.0f);

    // PathTracer.cpp:108-126
    int end = END_ALIVE;
    if (tp.max() == 0.0f) {
        end = END_CHECKED;
    } else {
        float roulettePdf = vabs(tp).max();
        bool cont = true;
        if (bounce > 2 && roulettePdf < 0.1f) {
            if (smp.nextBoolean(roulettePdf))
                tp /= roulettePdf;
            else
                cont = false;
        }
        if (!cont) {
            end = END_RAW;
        } else if (isnanf_(wo.sum() + p.sum())) {
            end = END_ZERO;
        } else {
            bounce++;
            if (bounce >= maxBounces) end = END_CHECKED;
        }
    }
    st.rayO[path] = mk4(p, 0.0f);
    st.rayD[path] = mk4(wo, TG_INF);
    st.samp[path] = make_uint4(smp.dimension, uint32_t(smp.pcg), uint32_t(smp.pcg >> 32), sw.w);
    st.med[path] = uint32_t(medium + 1) | (firstScatter ? 1u << 8 : 0u) | (stateBounce << 9);
    if (pushShadow) {
        if (!fwdScene && tfarL < 0.0f) { // a single ray always sits in slot A (the resolve adds 0 + B either way): k_shadow_q never arms an absent ray
            dL = dB; tfarL = tfarB; L = B;
            tfarB = -1.0f; B = V3(0.0f);
        }
        st.shO[path] = mk4(p, 0.0f);
        st.shA[path] = mk4(dL, tfarL);
        st.shB[path] = mk4(dB, tfarB);
        st.shC[path] = mk4(fwdScene ? fL : L, pickWeight);
        st.shD[path] = mk4u(fwdScene ? EB : B, 0u);
        st.shX[path] = make_float4(__uint_as_float(endCap), __uint_as_float(tfarB >= 0.0f ? 1u : 0u), 0.0f, 0.0f);
        if (fwdScene) {
            st.shG[path] = mk4(EL, pdfL);
            st.shH[path] = make_float4(misL, misB, 0.0f, 0.0f);
            st.shI[path] = mk4(wB, 0.0f);
        }
        // both rays start inside `medium`, not on a surface (startsOnSurface bit 2 clear)
        st.shE[path] = mk4u(tpVertex, fwdScene ? ((uint32_t(medium + 1) << 8) | (uint32_t(medium + 1) << 16)) : 0u);
        if (fwdScene) st.shF[path] = mk4(V3(0.0f), __uint_as_float(uint32_t(neeBounce)));
    }
    st.em[path] = mk4u(em, smp.scramble);
    st.tp[path] = mk4u(tp, packFlags(bounce, wasSpecular, end));
    closeVertex(st, path, pushShadow, end, tp.sum());
    return end;
}

// One surface vertex of one path (the body of the shade kernels; the tail kernel calls it too):
// TraceBase::handleSurface + the loop tail of PathTracer::traceSample.
// CLASS >= 0: the caller guarantees the hit BSDF's type (the per-class shade kernels), which lets the compiler drop the
// code of the other BSDFs; MEDIA = false is the lean instantiation: no medium bookkeeping, only environment and quad
// lights (the launcher picks the general one for scenes with media or sphere / disk lights).
// NESTED = false: the caller has checked that the hit BSDF is not made of other BSDFs (volumePath<false>).
template <int CLASS, bool MEDIA, bool NESTED>
TG_DEV int shadePath(const DScene &sc, const DState &st, uint32_t path, bool afterMedium)
{
    const int maxBounces = sc.settings.max_bounces, minBounces = sc.settings.min_bounces;
    const bool neeEnabled = sc.settings.enable_light_sampling != 0;
    bool pushShadow = false;
    float4 ro = st.rayO[path], rd = st.rayD[path], hw = st.hit[path], tpw = st.tp[path], emw = st.em[path];
    uint4 sw = st.samp[path];
    V3 O = xyz(ro), D = xyz(rd), tp = xyz(tpw), em = xyz(emw);
    uint32_t flags = __float_as_uint(tpw.w);
    int bounce = flagsBounce(flags);
    bool wasSpecular = flagsSpecular(flags);
    int end = END_ALIVE;
    if (stateNee(st.state[path])) // NEE terms of the previous vertex (k_shadow_q left the visibility bits)
        em = resolveNeeTerms<MEDIA>(st, path, em); // (inline: an out-of-line call here cost the general kernels 600 B of scratch per lane, config 5 -9 %)

    // afterMedium: volumePath ran the check before the distance sampling and scaled tp by the medium weight since
    if (!afterMedium && bounce > 0 && isnanf_(tp.sum() + em.sum())) {
        end = END_ZERO; // PathTracer.cpp:121-122 for the previous vertex
    } else {
        Sampler smp;
        smp.matrices = sc.sobol;
        smp.sobol = sc.settings.use_sobol;
        smp.dimension = sw.x;
        smp.pcg = (uint64_t(sw.z) << 32) | uint64_t(sw.y);
        smp.scramble = __float_as_uint(emw.w); // em.w carries the Sobol scramble, samp.w the sample index
        smp.index = sw.w;

        uint32_t prim = __float_as_uint(hw.w);
        HitInfo info;
        if (!MEDIA) __builtin_assume(primKind(prim) <= KIND_CUBE); // lean instantiation: no spheres / disks in the scene
        fetchHitInfo(sc, prim, O, D, hw.x, hw.y, hw.z, info);
        if (sc.nInstSets) {
            const uint32_t inst = st.hitInst[path];
            if (inst != INST_NONE) applyInstance(sc, inst, prim, O, D, hw.x, info);
        }
        const tgpu_bsdf &bsdf = sc.bsdfs[info.bsdf];
        if (CLASS >= 0) __builtin_assume(bsdf.type == CLASS);
        const uint32_t lobes = bsdfLobes(sc, bsdf);
        const bool media = MEDIA && sc.nMedia != 0;
        const int curMedium = media ? int(st.med[path] & 0xFFu) - 1 : -1;
        const bool overridesMedia = info.intMedium >= 0 || info.extMedium >= 0; // Primitive::selectMedium (Primitive.hpp:172-183)

        // makeLocalScatterEvent (TraceBase.cpp:24-51)
        Event ev;
        ev.frame = duffFrame(info.Ns);
        bool hitBackside = ev.frame.normal.dot(D) > 0.0f;
        bool flipFrame = sc.settings.enable_two_sided_shading && hitBackside && !(lobes & TransmissiveLobe);
        if (flipFrame) {
            ev.frame.normal = -ev.frame.normal;
            ev.frame.tangent = -ev.frame.tangent;
        }
        ev.wi = ev.frame.toLocal(-D);
        ev.wo = V3(0.0f);
        ev.weight = V3(1.0f);
        ev.pdf = 1.0f;
        ev.uvx = info.uvx;
        ev.uvy = info.uvy;
        ev.flippedFrame = flipFrame;
        ev.sampledLobe = 0;

        // handleSurface (TraceBase.cpp:516-568): the forward-lobe test draws from the supplemental stream for every surface
        // hit.  transparency = bsdf.eval(forward event): (1,1,1) for a ForwardBsdf (ForwardBsdf.cpp:26-29; avg() == 1.0f),
        // zero for every other BSDF in scope
        // zero for a BSDF without that lobe; 1 - alpha for a TransparencyBsdf; the film's transmittance for a ThinSheetBsdf
        // auxiliary output buffers: is this sample still waiting for its first non-specular vertex (PathTracer.cpp:78)?
        // (general instantiation only: requesting auxiliary buffers selects it at upload, the lean kernels carry none of this)
        const bool aovTrack = MEDIA && st.aov0 != nullptr;
        const uint32_t aovRef = aovTrack ? st.sampleRef[path] : 0u;
        const bool aovOpen = aovTrack && !(__float_as_uint(st.results[aovRef].w) & AOV_RECORDED);
        bool visDefined = false; // the light sample reached its shadow test: `transmittance` of handleSurface is set (TraceBase.cpp:167-169)
        float visClear = 0.0f;   // ... to this value if the shadow ray turns out clear (scenes without forward-scattering surfaces)
        V3 transparency(0.0f);
        if (lobes & ForwardLobe) transparency = forwardTransparency(sc, bsdf, ev.wi.z, info.uvx, info.uvy);
        const float transparencyScalar = transparency.avg();
        const bool passThrough = smp.nextBoolean(transparencyScalar);
        const bool fwdScene = MEDIA && sc.hasForward != 0; // shadow rays may cross surfaces: the shadow kernel applies the media
        int mediumL = -1, mediumB = -1;

        const float eps = 5e-4f; // TraceableScene.hpp:39
        V3 L(0.0f), B(0.0f);
        float tfarL = -1.0f, tfarB = -1.0f, pickWeight = 1.0f;
        V3 dL(0.0f), dB(0.0f);
        V3 fL(0.0f), EL(0.0f), EB(0.0f), wB(0.0f);
        float pdfL = 1.0f, misL = 0.0f, misB = 0.0f;
        uint32_t endCap = PRIM_MISS;
        const V3 tpVertex = tp;
        const int neeBounce = bounce + 1; // estimateDirect(event, medium, bounce + 1, ...)

        if (!passThrough && neeEnabled && bounce < maxBounces - 1 && sc.nLights > 0) {
            // estimateDirect -> chooseLight (TraceBase.cpp:416-459)
            int lightIdx = chooseLight<!MEDIA>(sc, info.p, smp, pickWeight);
            bool haveLight = lightIdx >= 0;
            bool pureSpecular = lobes != 0 && (lobes & ~uint32_t(SpecularLobe)) == 0;
            if (haveLight && !pureSpecular && lobes != ForwardLobe) { // sampleDirect (TraceBase.cpp:383-400)
                const tgpu_light light = sc.lights[lightIdx];
                if (!MEDIA) __builtin_assume(light.kind == TGPU_LIGHT_ENV || light.kind == TGPU_LIGHT_QUAD); // lean instantiation
                const bool isEnv = light.kind == TGPU_LIGHT_ENV;
                endCap = lightPrim(light);
                const bool shadowPasses = bounce + 1 >= minBounces; // generalizedShadowRay's minBounces rule

                // ---- lightSample (TraceBase.cpp:246-285) ----
                {
                    LightSample ls;
                    bool ok = lightSampleDirect(sc, light, info.p, smp, ls);
                    if (ok) {
                        ev.wo = ev.frame.toLocal(ls.d);
                        bool consistent = true;
                        if (sc.settings.enable_consistency_checks) {
                            bool geometricBackside = ls.d.dot(info.Ng) < 0.0f;
                            bool shadingBackside = (ev.wo.z < 0.0f) ^ ev.flippedFrame;
                            consistent = geometricBackside == shadingBackside;
                        }
                        ev.requestedLobe = AllButSpecular;
                        V3 f = consistent ? bsdfEval<CLASS, NESTED>(sc, bsdf, ev) : V3(0.0f);
                        if (!f.isZero()) {
                            // attenuatedEmission (TraceBase.cpp:144-174) up to the shadow test
                            V3 E(0.0f);
                            float farT = TG_INF;
                            bool lit = true;
                            if (isEnv) {
                                float ux, uy;
                                if (sc.envConstant) E = V3(sc.textures[sc.env.emission_tex].value);
                                else { directionToUV(sc, ls.d, ux, uy, nullptr); E = texEval(sc, sc.env.emission_tex, ux, uy); }
                            } else {
                                float lightPdfUnused;
                                lit = finiteLightHit(sc, light, info.p, ls.d, eps, ls.dist, farT, E, lightPdfUnused);
                            }
                            // generalizedShadowRay: the medium on the light's side attenuates the unoccluded segment (:261-262, :100-108)
                            V3 clearShadow(1.0f);
                            if (media) {
                                mediumL = overridesMedia ? (ls.d.dot(info.Ng) < 0.0f ? info.intMedium : info.extMedium) : curMedium;
                                if (!fwdScene) {
                                    clearShadow = neeTransmittance(sc, mediumL, farT, true);
                                    E = clearShadow*E;
                                }
                            }
                            if (lit) {
                                visDefined = true;
                                visClear = clearShadow.avg();
                            }
                            // the reference traces the shadow ray whatever E is; here a ray that cannot add radiance is only traced
                            // when the 'visibility' buffer may want its result
                            if (lit && shadowPasses && (!E.isZero() || aovOpen)) {
                                if (!E.isZero()) { // (e == 0 returns before f*e/pdf, TraceBase.cpp:276-279)
                                    V3 lightF = f*E/ls.pdf;
                                    const float mis = powerHeuristic(ls.pdf, bsdfPdf<CLASS, NESTED>(sc, bsdf, ev));
                                    lightF *= mis;
                                    L = lightF;
                                    misL = mis;
                                }
                                dL = ls.d;
                                tfarL = farT;
                                fL = f; EL = E; pdfL = ls.pdf;
                            }
                        }
                    }
                }
                // ---- bsdfSample (TraceBase.cpp:287-321) ----
                {
                    ev.requestedLobe = AllButSpecular;
                    if (bsdfSample<CLASS, NESTED>(sc, bsdf, ev, smp) && !ev.weight.isZero()) {
                        V3 wo = ev.frame.toGlobal(ev.wo);
                        bool consistent = true;
                        if (sc.settings.enable_consistency_checks) {
                            bool geometricBackside = wo.dot(info.Ng) < 0.0f;
                            bool shadingBackside = (ev.wo.z < 0.0f) ^ ev.flippedFrame;
                            consistent = geometricBackside == shadingBackside;
                        }
                        if (consistent) {
                            V3 E(0.0f);
                            float farT = TG_INF, lightPdf = 0.0f;
                            bool lit = true;
                            if (isEnv) {
                                float ux, uy;
                                lightPdf = envDirectPdf(sc, wo, ux, uy);
                                E = sc.envConstant ? V3(sc.textures[sc.env.emission_tex].value) : texEval(sc, sc.env.emission_tex, ux, uy);
                            } else {
                                lit = finiteLightHit(sc, light, info.p, wo, eps, -1.0f, farT, E, lightPdf); // expectedDist = -1 never rejects
                            }
                            if (media) { // :301-302
                                mediumB = overridesMedia ? (wo.dot(info.Ng) < 0.0f ? info.intMedium : info.extMedium) : curMedium;
                                if (!fwdScene) E = neeTransmittance(sc, mediumB, farT, true)*E;
                            }
                            if (lit && shadowPasses && !E.isZero()) {
                                V3 bsdfF = E*ev.weight;
                                const float mis = powerHeuristic(ev.pdf, lightPdf);
                                bsdfF *= mis;
                                B = bsdfF;
                                dB = wo;
                                tfarB = farT;
                                EB = E; wB = ev.weight; misB = mis;
                            }
                        }
                    }
                }
            }
        }
        pushShadow = tfarL >= 0.0f || tfarB >= 0.0f;

        // emitter hit (TraceBase.cpp:540-543): added after the NEE term, as the reference does
        V3 postAdd(0.0f);
        bool hasPostAdd = false;
        if (!passThrough && info.emissive && bounce >= minBounces) {
            if (!neeEnabled || wasSpecular) { // every emissive primitive in scope is samplable
                V3 Ed = info.backSide ? V3(0.0f) : texEval(sc, info.emissionTex, info.uvx, info.uvy);
                postAdd = Ed*tp;
                hasPostAdd = true;
            }
        }
        if (hasPostAdd && !pushShadow) em += postAdd;

        // continuation (TraceBase.cpp:546-565)
        ev.requestedLobe = AllLobes;
        bool sampled = !passThrough && bsdfSample<CLASS, NESTED>(sc, bsdf, ev, smp);
        V3 wo(0.0f);
        bool alive = sampled;
        if (passThrough) { // wo = ray.dir(); weight = transparency/transparencyScalar; sampledLobe = ForwardLobe
            wo = D;
            alive = true;
            ev.weight = transparency/transparencyScalar;
        }
        if (sampled) {
            wo = ev.frame.toGlobal(ev.wo);
            if (sc.settings.enable_consistency_checks) {
                bool geometricBackside = wo.dot(info.Ng) < 0.0f;
                bool shadingBackside = (ev.wo.z < 0.0f) ^ ev.flippedFrame;
                alive = geometricBackside == shadingBackside;
            }
        }
        V3 hitpoint = O + D*hw.x;
        if (media && alive) { // medium = selectMedium(medium, geometricBackside); state.reset() (TraceBase.cpp:560-562)
            int next = overridesMedia ? (wo.dot(info.Ng) < 0.0f ? info.intMedium : info.extMedium) : curMedium;
            st.med[path] = uint32_t(next + 1) | (1u << 8);
        }
        if (!alive) {
            end = END_RAW; // handleSurface returned false -> `return emission`
        } else {
            tp *= ev.weight;
            if (!passThrough) wasSpecular = (ev.sampledLobe & SpecularLobe) != 0;
            // PathTracer.cpp:108-126
            if (tp.max() == 0.0f) {
                end = END_CHECKED; // break: the environment term is throughput(=0)*E
            } else {
                float roulettePdf = vabs(tp).max();
                bool cont = true;
                if (bounce > 2 && roulettePdf < 0.1f) {
                    if (smp.nextBoolean(roulettePdf))
                        tp /= roulettePdf;
                    else
                        cont = false;
                }
                if (!cont) {
                    end = END_RAW;
                } else if (isnanf_(wo.sum() + hitpoint.sum())) {
                    end = END_ZERO;
                } else {
                    bounce++;
                    if (bounce >= maxBounces) end = END_CHECKED;
                }
            }
        }
        if (aovOpen) { // PathTracer.cpp:64,78-96 (wasSpecular as handleSurface left it; terminate = !alive)
            const float hitDistance = st.aov0[aovRef].x + hw.x;
            if (!wasSpecular || !alive) {
                const tgpu_bsdf &shown = bsdf.type == TGPU_BSDF_TRANSPARENCY ? sc.bsdfs[bsdf.substrate] : bsdf; // :85-88
                V3 albedo = texEval(sc, shown.albedo_tex, info.uvx, info.uvy);
                if (info.emissive) albedo += info.backSide ? V3(0.0f) : texEval(sc, info.emissionTex, info.uvx, info.uvy);
                float vis = __uint_as_float(0x7FC00000u);
                const bool pending = visDefined && tfarL >= 0.0f;
                if (visDefined) vis = pending ? visClear : 0.0f; // a shadow ray below min_bounces returns 0 (TraceBase.cpp:113-116)
                st.aov0[aovRef] = mk4(V3(hitDistance, info.Ns.x, info.Ns.y), info.Ns.z);
                st.aov1[aovRef] = mk4(albedo, vis);
                reinterpret_cast<float *>(&st.results[aovRef])[3] = __uint_as_float(AOV_RECORDED | (pending ? AOV_VIS_PENDING : 0u));
            } else {
                reinterpret_cast<float *>(&st.aov0[aovRef])[0] = hitDistance;
            }
        }
        // NEE rays start at info.p.  For scene primitives that has the bits of the scatter origin (ray.hitpoint());
        // for a hit on an "instances" primitive it does not (applyInstance)
        const bool neeOriginMoved = sc.nInstSets && st.hitInst[path] != INST_NONE;
        st.rayO[path] = mk4(hitpoint, eps);
        st.rayD[path] = mk4(wo, TG_INF);
        st.samp[path] = make_uint4(smp.dimension, uint32_t(smp.pcg), uint32_t(smp.pcg >> 32), sw.w);
        if (pushShadow) {
            if (!fwdScene && tfarL < 0.0f) { // a single ray always sits in slot A (the resolve adds 0 + B either way)
                dL = dB; tfarL = tfarB; L = B;
                tfarB = -1.0f; B = V3(0.0f);
            }
            st.shO[path] = mk4(neeOriginMoved ? info.p : hitpoint, eps);
            st.shA[path] = mk4(dL, tfarL);
            st.shB[path] = mk4(dB, tfarB);
            st.shC[path] = mk4(fwdScene ? fL : L, pickWeight);
            st.shD[path] = mk4u(fwdScene ? EB : B, 0u);
        st.shX[path] = make_float4(__uint_as_float(endCap), __uint_as_float(tfarB >= 0.0f ? 1u : 0u), 0.0f, 0.0f);
            if (fwdScene) {
                st.shG[path] = mk4(EL, pdfL);
                st.shH[path] = make_float4(misL, misB, 0.0f, 0.0f);
                st.shI[path] = mk4(wB, 0.0f);
            }
            // with forward-scattering surfaces in the scene the shadow kernel walks the segments itself and needs the
            // media the two rays start in, startsOnSurface (bit 2) and the bounce argument (shF.w)
            const uint32_t fwdBits = fwdScene ? (4u | (uint32_t(mediumL + 1) << 8) | (uint32_t(mediumB + 1) << 16)) : 0u;
            st.shE[path] = mk4u(tpVertex, (hasPostAdd ? 1u : 0u) | fwdBits);
            const float bounceBits = __uint_as_float(uint32_t(neeBounce));
            // (always all eight sectors of the NEE record, the unused emitter term too: a whole-line write needs no merge with the line's old
            // contents on its way to HBM; +0.6 %, profiles/r03_ab_notes.txt)
            st.shF[path] = mk4(hasPostAdd ? postAdd : V3(0.0f), bounceBits);
        }
        st.em[path] = mk4u(em, smp.scramble);
    }
    st.tp[path] = mk4u(tp, packFlags(bounce, wasSpecular, end));
    closeVertex(st, path, pushShadow, end, tp.sum());
    return end;
}

template <int CLASS, bool MEDIA>
#ifndef TG_SHADE_WAVES
#define TG_SHADE_WAVES 4 /* waves per SIMD the shade kernels are compiled for (A/B: profiles/r02_ab_notes.txt) */
#endif
__global__ void __launch_bounds__(256, TG_SHADE_WAVES) k_shade(DScene sc, DState st, int set)
{
    const uint32_t count = st.counters[CNT(CNT_LIST0 + set*TG_LISTS + 1 + CLASS)];
    const uint32_t *list = st.lists[set][1 + CLASS];
    const uint32_t stride = gridDim.x*blockDim.x;
    constexpr int KNOWN = CLASS < TG_CLASS_OTHER ? CLASS : -1; // the shared list holds several BSDF types
    // the end state of every entry also goes to a byte array parallel to the concatenated lists: the retire step reads it coalesced and
    // only touches the records of paths that ended (a 4-byte state word per slot cost it a 128-byte line fill each, profiles/r02_ab_notes.txt)
    uint8_t *endFlag = st.endFlag[set] + listBase(st, set, 1 + CLASS);
    for (uint32_t q = blockIdx.x*blockDim.x + threadIdx.x; q < count; q += stride) endFlag[q] = uint8_t(shadePath<KNOWN, MEDIA>(sc, st, list[q]));
    if (threadIdx.x == 0 && blockIdx.x == 0) atomicAdd(&st.stats[5], (unsigned long long)count);
}

// K3v: the slots whose ray runs inside a medium (list 1 + TG_CLASS_VOLUME)
#ifdef TG_TU_MAIN
static __global__ void __launch_bounds__(256, 2) k_volume(DScene sc, DState st, int set)
{
    const uint32_t count = st.counters[CNT(CNT_LIST0 + set*TG_LISTS + 1 + TG_CLASS_VOLUME)];
    const uint32_t *list = st.lists[set][1 + TG_CLASS_VOLUME];
    uint8_t *endFlag = st.endFlag[set] + listBase(st, set, 1 + TG_CLASS_VOLUME);
    for (uint32_t q = blockIdx.x*blockDim.x + threadIdx.x; q < count; q += gridDim.x*blockDim.x) endFlag[q] = uint8_t(volumePath<false>(sc, st, list[q]));
}
// ... and the entries k_volume left behind: paths that left their medium onto a BSDF made of other BSDFs (launched only for scenes that have both)
static __global__ void __launch_bounds__(256, 2) k_volume_nested(DScene sc, DState st, int set)
{
    const uint32_t count = st.counters[CNT(CNT_LIST0 + set*TG_LISTS + 1 + TG_CLASS_VOLUME)];
    const uint32_t *list = st.lists[set][1 + TG_CLASS_VOLUME];
    uint8_t *endFlag = st.endFlag[set] + listBase(st, set, 1 + TG_CLASS_VOLUME);
    for (uint32_t q = blockIdx.x*blockDim.x + threadIdx.x; q < count; q += gridDim.x*blockDim.x)
        if (endFlag[q] == END_DEFERRED) endFlag[q] = uint8_t(shadePath<-1, true>(sc, st, list[q], true));
}
#endif

// =================================================================================================
// K4 shadow: both NEE rays of a vertex are traced by one lane, then resolved in the reference's order
//   emission += ((0 + L*visL) + B*visB) * weight * throughput   (TraceBase.cpp:383-400,483-494,537-538)
// =================================================================================================
// Both NEE rays of the path's current vertex, then the resolve (body of the shadow kernel and of the tail kernel)
// generalizedShadowRayImpl<false> (TraceBase.cpp:62-125) for scenes with forward-scattering surfaces: closest hit per
// segment; a surface with a ForwardLobe lets the ray through (ForwardBsdf::eval == 1) and may switch the medium, anything
// else blocks; every segment is attenuated by the medium it runs in (Transmittance::eval with the start / end flags).
template <bool COUNT, bool INST>
TG_DEV V3 shadowWalk(const DScene &sc, uint32_t *stack, V3 O, const V3 &D, float tnear, float tfar, uint32_t endCap, int medium, bool startsOnSurface,
                     int bounce, TraceCounters &tc, uint32_t &rays)
{
    float initialFarT = tfar;
    V3 throughput(1.0f);
    while (true) {
        rays++;
        float t = tfar, u = 0.0f, v = 0.0f;
        uint32_t inst;
        const uint32_t prim = closestHit<COUNT, INST>(sc, stack, O, D, tnear, t, u, v, inst, tc);
        const bool isEnd = prim == PRIM_MISS || (inst == INST_NONE && prim == endCap);
        int intMedium = -1, extMedium = -1;
        bool backSide = false;
        if (!isEnd) {
            const uint32_t kind = primKind(prim), idx = primIndex(prim);
            int bsdf;
            if (kind == KIND_TRI) {
                const float4 s3 = sc.triShade[size_t(idx)*5 + 3], s4 = sc.triShade[size_t(idx)*5 + 4];
                bsdf = int(__float_as_uint(s3.w) & 0x7FFFFFFFu);
                const uint32_t mb = __float_as_uint(s4.w);
                intMedium = int(mb >> 16) - 1;
                extMedium = int(mb & 0xFFFFu) - 1;
                backSide = V3(s4.x, s4.y, s4.z).dot(D) > 0.0f; // TriangleMesh.cpp:331
            } else if (kind == KIND_QUAD) {
                const tgpu_quad &q = sc.quads[idx];
                bsdf = q.bsdf; intMedium = q.int_medium; extMedium = q.ext_medium;
                backSide = D.dot(V3(q.normal)) >= 0.0f;        // Quad.cpp:93
            } else if (kind == KIND_SPHERE) {
                const tgpu_sphere &sp = sc.spheres[idx];
                bsdf = sp.bsdf; intMedium = sp.int_medium; extMedium = sp.ext_medium;
                float ts = 0.0f; // Sphere.cpp:74-90: the near root if it lies inside (nearT, farT), else the far one
                bool front = sphereIntersect(sp, O, D, tnear, TG_INF, ts, backSide);
                if (!front || ts != t) backSide = true;
            } else if (kind == KIND_DISK) {
                const tgpu_disk &dk = sc.disks[idx];
                bsdf = dk.bsdf; intMedium = dk.int_medium; extMedium = dk.ext_medium;
                backSide = -D.dot(V3(dk.n)) < dk.cos_apex;       // Disk.cpp:82
            } else {
                const tgpu_cube &cb = sc.cubes[idx];
                bsdf = cb.bsdf; intMedium = cb.int_medium; extMedium = cb.ext_medium;
                float tc2 = 0.0f; // Cube.cpp:113-124: the front face if it lies past nearT, else the back face
                bool front = cubeIntersect(cb, O, D, tnear, TG_INF, tc2, backSide);
                if (!front || tc2 != t) backSide = true;
            }
            if (inst != INST_NONE) { intMedium = extMedium = -1; }
            const tgpu_bsdf &bd = sc.bsdfs[bsdf];
            if (!(bsdfLobes(sc, bd) & ForwardLobe)) return V3(0.0f);
            if (bd.type != TGPU_BSDF_FORWARD) { // transparency = eval(forward event) (TraceBase.cpp:84-92); 1 for a ForwardBsdf
                HitInfo hi;
                fetchHitInfo(sc, prim, O, D, t, u, v, hi);
                if (inst != INST_NONE) applyInstance(sc, inst, prim, O, D, t, hi);
                const V3 transparency = forwardTransparency(sc, bd, duffFrame(hi.Ns).normal.dot(-D), hi.uvx, hi.uvy);
                if (transparency.isZero()) return V3(0.0f);
                throughput *= transparency;
            }
            bounce++;
            if (bounce >= sc.settings.max_bounces) return V3(0.0f);
        }
        if (medium >= 0) throughput *= neeTransmittance(sc, medium, t, startsOnSurface);
        if (isEnd) return bounce >= sc.settings.min_bounces ? throughput : V3(0.0f);
        if (intMedium >= 0 || extMedium >= 0) medium = !backSide ? intMedium : extMedium; // selectMedium(medium, !hitBackside(data))
        startsOnSurface = true;
        O = O + D*t; // ray.hitpoint()
        initialFarT -= t;
        tnear = 5e-4f;
        tfar = initialFarT;
    }
}

template <bool COUNT, bool INST, bool FWD>
TG_DEV void shadowPath(const DScene &sc, const DState &st, uint32_t *stack, uint32_t path, TraceCounters &tc, uint32_t &rays)
{
    float4 ro = ldStream(&st.shO[path]), a = ldStream(&st.shA[path]), b = ldStream(&st.shB[path]), c = ldStream(&st.shC[path]), d = ldStream(&st.shD[path]), e = ldStream(&st.shE[path]);
    V3 O = xyz(ro);
    uint32_t endCap = ldStream(reinterpret_cast<const uint32_t *>(&st.shX[path]));
    V3 result(0.0f);
    if (FWD) {
        const uint32_t bits = __float_as_uint(e.w);
        const bool startsOnSurface = (bits & 4u) != 0;
        const int mediumL = int((bits >> 8) & 0xFFu) - 1, mediumB = int((bits >> 16) & 0xFFu) - 1;
        const int bounce = int(__float_as_uint(ldStream(&st.shF[path]).w));
        const float4 g = ldStream(&st.shG[path]), h = ldStream(&st.shH[path]);
        if (a.w >= 0.0f) { // lightSample: e = shadow*E; lightF = f*e/pdf; lightF *= mis (TraceBase.cpp:272-283)
            const V3 shadowL = shadowWalk<COUNT, INST>(sc, stack, O, xyz(a), ro.w, a.w, endCap, mediumL, startsOnSurface, bounce, tc, rays);
            if (st.aov0) aovVisibility(st, path, false, true, shadowL);
            V3 eL = shadowL*xyz(g);
            if (!eL.isZero()) {
                V3 lightF = xyz(c)*eL/g.w;
                lightF *= h.x;
                result += lightF;
            }
        }
        if (b.w >= 0.0f) { // bsdfSample: bsdfF = e*weight; bsdfF *= mis (:309-318)
            V3 eB = shadowWalk<COUNT, INST>(sc, stack, O, xyz(b), ro.w, b.w, endCap, mediumB, startsOnSurface, bounce, tc, rays)*xyz(d);
            if (!eB.isZero()) {
                V3 bsdfF = eB*xyz(ldStream(&st.shI[path]));
                bsdfF *= h.y;
                result += bsdfF;
            }
        }
    } else {
        if (a.w >= 0.0f) {
            rays++;
            const bool clear = !occluded<COUNT, INST>(sc, stack, O, xyz(a), ro.w, a.w, endCap, tc);
            if (clear) result += xyz(c);
            if (st.aov0) aovVisibility(st, path, clear, false, V3(0.0f));
        }
        if (b.w >= 0.0f) {
            rays++;
            if (!occluded<COUNT, INST>(sc, stack, O, xyz(b), ro.w, b.w, endCap, tc)) result += xyz(d);
        }
    }
    float4 emw = ldStream(&st.em[path]);
    V3 em = xyz(emw);
    em += (result*c.w)*xyz(e);
    if (__float_as_uint(e.w) & 1u) em += xyz(ldStream(&st.shF[path]));
    stStream(&st.em[path], make_float4(em.x, em.y, em.z, emw.w));
    st.state[path] = st.state[path] & ~1u; // resolved here: nothing left for the next step that touches the path
}

template <bool COUNT, bool INST, bool FWD>
__global__ void __launch_bounds__(256, ((INST || FWD) ? 1 : TG_SHADOW_WAVES)) k_shadow(DScene sc, DState st, int set)
{
    extern __shared__ uint32_t ldsStack[];
    uint32_t *stack = ldsStack + threadIdx.x;
    const int lane = threadIdx.x & 63;
    const ListCounts lc = loadListCounts(st, set);
    const uint32_t count = listTotal(lc, 1); // every shaded slot; those without NEE rays are skipped
    TraceCounters tc{0, 0};
    uint32_t rays = 0;
    while (true) {
        uint32_t base = 0;
        if (lane == 0) base = atomicAdd(&st.counters[CNT(CNT_HEAD_SHADOW)], 64u);
        base = __shfl(base, 0);
        if (base >= count) break;
        uint32_t idx = base + lane;
        uint32_t path = 0;
        // no `continue` here: the dequeue above must be executed by the whole wave together
        if (idx < count && listFetch(st, lc, set, 1, idx, path) && stateNee(ldStream(&st.state[path]))) shadowPath<COUNT, INST, FWD>(sc, st, stack, path, tc, rays);
    }
    if (COUNT) flushCounters(st.stats, 2, tc);
    for (int off = 32; off > 0; off >>= 1) rays += __shfl_down(rays, off);
    if (lane == 0 && rays) atomicAdd(&st.stats[4], (unsigned long long)rays);
}

// =================================================================================================
// K4 shadow, queue form (scenes without instances / spheres / disks / forward-scattering surfaces): pure any-hit traversal with
// persistent lanes.  The old kernel traced both NEE rays of a vertex in one lane and resolved them there: 14 % of the lanes busy
// per VALU instruction (profiles/r01_ab_notes.txt) -- any-hit rays are short and bimodal, a wave lasted as long as its
// unluckiest vertex.  Here a ray is 36 bytes in (origin, direction, far end, end cap: the first half of the shadow record) and
// ONE BIT out.  A wave claims CHUNK = 64*DEPTH list entries with one atomic and hands them to its lanes one vertex at a time,
// whenever a lane runs out of rays (wave-private cursor, no atomics): no lane waits for the slowest lane of a batch, only for
// the next group re-arm (>= TG_SHADOWQ_THRESH waiting lanes, or nobody tracing).  Every lane keeps the next ray's record and the
// next vertex's list entry in flight while it traces, so re-arming never waits for memory; absent rays (far end < 0) cost a
// re-arm, not a traversal.  The visibility bits of a vertex go to shD.w; k_retire_regenerate folds the NEE terms into the
// radiance (it visits every slot anyway and reads the second halves of both records).
// =================================================================================================
#ifndef TG_SHADOWQ_THRESH
#define TG_SHADOWQ_THRESH 24 /* 16 / 24 / 32: shadow 4.08 / 4.02 / 4.01 ms per step */
#endif
#ifndef TG_SHADOWQ_WAVES
#define TG_SHADOWQ_WAVES 7
#endif
#ifndef TG_SHADOWQ_SORTED
#define TG_SHADOWQ_SORTED false
#endif
template <bool COUNT, int DEPTH>
__global__ void __launch_bounds__(256, TG_SHADOWQ_WAVES) k_shadow_q(DScene sc, DState st, int set)
{
    constexpr uint32_t CHUNK = 64u*DEPTH;
    extern __shared__ uint32_t ldsStack[];
    uint32_t *stack = ldsStack + threadIdx.x;
    const int lane = threadIdx.x & 63;
    const unsigned long long below = (1ull << lane) - 1ull;
    stageNodelets(sc);
    const ListCounts lc = loadListCounts(st, set);
    const uint32_t count = listTotal(lc, 1); // every shaded slot
    TraceCounters tc{0, 0};
    uint32_t rays = 0, verts = 0;
    // the wave's private chunk of list entries [chunkPos, chunkEnd) (wave-uniform)
    uint32_t chunkPos = 0, chunkEnd = 0;
    bool exhausted = false;
    // hands one list entry to every lane that wants one (all lanes call)
    auto claim = [&](bool want, uint32_t &slot) -> bool {
        bool got = false;
        unsigned long long m = __ballot(want);
        while (m) {
            if (chunkPos == chunkEnd) {
                if (exhausted) break;
                uint32_t b = 0;
                if (lane == 0) b = atomicAdd(&st.counters[CNT(CNT_HEAD_SHADOW)], CHUNK);
                b = uint32_t(__builtin_amdgcn_readfirstlane(int(b)));
                if (b >= count) {
                    exhausted = true;
                    break;
                }
                chunkPos = b;
                chunkEnd = min(b + CHUNK, count);
            }
            const uint32_t rank = uint32_t(__builtin_popcountll(m & below)), avail = chunkEnd - chunkPos;
            const bool mine = ((m >> lane) & 1ull) && rank < avail;
            if (mine) got = listFetch(st, lc, set, 1, chunkPos + rank, slot);
            chunkPos += min(uint32_t(__builtin_popcountll(m)), avail);
            m &= ~__ballot(mine);
        }
        return got;
    };
    // lane state: the vertex being traced (vs, ray cw), the ray whose record is in flight (pSlot, pWhich -> nro, nrd, ncap), the
    // lane's next vertex (nvs: its list entry is in flight)
    uint32_t vs = 0, pSlot = 0, nvs = 0, ncap = PRIM_MISS, nHasB = 0;
    int cw = 0, pWhich = 0;
    bool pre = false, haveNext = false;
    float4 nro = make_float4(0.0f, 0.0f, 0.0f, 0.0f), nrd = nro;
    auto fetchRay = [&]() {
        nrd = ldStream(pWhich ? &st.shB[pSlot] : &st.shA[pSlot]);
        nro = ldStream(&st.shO[pSlot]);
        ncap = ldStream(reinterpret_cast<const uint32_t *>(&st.shX[pSlot]));
        nHasB = ldStream(reinterpret_cast<const uint32_t *>(&st.shX[pSlot]) + 1);
    };
    pre = claim(true, pSlot);
    if (pre) fetchRay();
    haveNext = claim(pre, nvs);
    bool active = false, fin = false, blocked = false;
    uint32_t vis = 0; // of the vertex vs: bit 0 / 1 = ray unoccluded, bit 2 = the vertex has a ray at all
    bool last = true; // the ray being traced is its vertex's last one (its second ray, or its only one)
    auto complete = [&](bool unoccluded) {
        if (unoccluded) vis |= 1u << cw;
        if (last) {
            if (vis & 4u) {
                stStream(reinterpret_cast<uint32_t *>(&st.shD[vs]) + 3, vis & 3u);
                verts++;
            }
            vis = 0;
        }
    };
    TravState t;
    t.node = 0; t.sp = 0; t.bestT = 0.0f; t.bestU = 0.0f; t.bestV = 0.0f; t.bestPrim = PRIM_MISS;
    while (true) {
        if (active) {
            const int s = travStep<true, COUNT, TG_SHADOWQ_SORTED>(sc, stack, t, tc);
            if (s != TRAV_MORE) {
                active = false;
                fin = true;
                blocked = s == TRAV_ANYHIT;
            }
        }
        const bool waiting = !active && (fin || pre);
        const unsigned long long w = __ballot(waiting), act = __ballot(active);
        if (w == 0 && act == 0) break;
        if (act != 0 && __builtin_popcountll(w) < TG_SHADOWQ_THRESH) continue;
        bool wantVertex = false;
        if (waiting) {
            if (fin) {
                complete(!blocked);
                fin = false;
            }
            if (pre) { // arm the ray whose record was fetched ahead
                vs = pSlot;
                cw = pWhich;
                last = cw == 1 || !nHasB;
                if (nrd.w >= 0.0f) {
                    vis |= 4u;
                    rays++;
                    const V3 O = xyz(nro), D = xyz(nrd);
                    const float tnear = nro.w, tfar = nrd.w;
                    bool hitAnalytic = false; // occluded()'s order: quads (not the end cap), cubes, triangles
                    for (uint32_t i = 0; i < sc.nQuads; ++i) {
                        QuadHit qh;
                        if (mkPrim(KIND_QUAD, i) != ncap && quadIntersect(sc.quads[i], O, D, tnear, tfar, qh)) hitAnalytic = true;
                    }
                    for (uint32_t i = 0; i < sc.nCubes; ++i) {
                        float tc2;
                        bool back;
                        if (cubeIntersect(sc.cubes[i], O, D, tnear, tfar, tc2, back)) hitAnalytic = true;
                    }
                    if (hitAnalytic) {
                        complete(false);
                    } else if (sc.nTris) {
                        travInit(t, sc.rootChild, O, D, tnear, tfar, PRIM_MISS);
                        active = true;
                    } else {
                        complete(true);
                    }
                } else {
                    complete(false); // no ray: the bit is never looked at
                }
                // the next ray of this lane: the vertex's second ray if it has one, else the first one of the next vertex
                if (!last) {
                    pWhich = 1;
                    fetchRay();
                } else if (haveNext) {
                    pSlot = nvs;
                    pWhich = 0;
                    haveNext = false;
                    wantVertex = true;
                    fetchRay();
                } else {
                    pre = false;
                }
            }
        }
        uint32_t slot = 0;
        const bool got = claim(wantVertex, slot);
        if (wantVertex) {
            haveNext = got;
            nvs = slot;
        }
    }
    if (COUNT) flushCounters(st.stats, 2, tc);
    for (int off = 32; off > 0; off >>= 1) {
        rays += __shfl_down(rays, off);
        verts += __shfl_down(verts, off);
    }
    if (lane == 0 && rays) atomicAdd(&st.stats[4], (unsigned long long)rays);
    if (lane == 0 && verts) atomicAdd(&st.stats[7], (unsigned long long)verts);
}

// =================================================================================================
// The queue kernels for scenes with "instances" (and spheres / disks): k_extend_q's and k_shadow_q's scaffolding around the flat
// instance state machine (instStep).  Round 1 ran these scenes one ray per lane through nested loops with the top-level stack
// in scratch memory (400-1168 B per lane) at one wave per SIMD.
// =================================================================================================
#ifndef TG_INSTQ_WAVES
#define TG_INSTQ_WAVES 4 /* 128 VGPRs (3 waves: 135-138): config 4 shadow 28.3 -> 27.4 ms/step */
#endif
// Closest hit with persistent lanes: a wave claims CHUNK = 64*DEPTH entries of the input lists with one atomic and hands them to
// its lanes one ray at a time (k_shadow_q's scheme); the hit goes to the slot's record, the sort by material is k_classify's job
// (a ray of this scene family costs 5-7 thousand instructions, the extra pass over the list a few dozen).
template <bool COUNT, int DEPTH>
__global__ void __launch_bounds__(256, TG_INSTQ_WAVES) k_extend_iq(DScene sc, DState st, int set)
{
    constexpr uint32_t CHUNK = 64u*DEPTH;
    extern __shared__ uint32_t ldsStack[];
    uint32_t *stack = ldsStack + threadIdx.x;
    uint2 *topStack = reinterpret_cast<uint2 *>(ldsStack + sc.stackLds*256) + threadIdx.x;
    const int lane = threadIdx.x & 63;
    const unsigned long long below = (1ull << lane) - 1ull;
    const ListCounts lc = loadListCounts(st, set);
    const uint32_t count = listTotal(lc, 0);
    TraceCounters tc{0, 0};
    uint32_t chunkPos = 0, chunkEnd = 0;
    bool exhausted = false;
    auto claim = [&](bool want, uint32_t &slot) -> bool {
        bool got = false;
        unsigned long long m = __ballot(want);
        while (m) {
            if (chunkPos == chunkEnd) {
                if (exhausted) break;
                uint32_t b = 0;
                if (lane == 0) b = atomicAdd(&st.counters[CNT(CNT_HEAD_EXT)], CHUNK);
                b = uint32_t(__builtin_amdgcn_readfirstlane(int(b)));
                if (b >= count) {
                    exhausted = true;
                    break;
                }
                chunkPos = b;
                chunkEnd = min(b + CHUNK, count);
            }
            const uint32_t rank = uint32_t(__builtin_popcountll(m & below)), avail = chunkEnd - chunkPos;
            const bool mine = ((m >> lane) & 1ull) && rank < avail;
            if (mine) got = listFetch(st, lc, set, 0, chunkPos + rank, slot);
            chunkPos += min(uint32_t(__builtin_popcountll(m)), avail);
            m &= ~__ballot(mine);
        }
        return got;
    };
    // the ray being traced (slot vs), the ray whose record is in flight (pSlot -> nro, nrd, nflags), the lane's next list entry (nvs)
    uint32_t vs = 0, pSlot = 0, nvs = 0, nflags = 0;
    bool pre = false, haveNext = false;
    float4 nro = make_float4(0.0f, 0.0f, 0.0f, 0.0f), nrd = nro;
    auto fetchRay = [&]() {
        nro = ldStream<TG_STATE_NT_INST>(&st.rayO[pSlot]);
        nrd = ldStream<TG_STATE_NT_INST>(&st.rayD[pSlot]);
        nflags = ldStream<TG_STATE_NT_INST>(reinterpret_cast<const uint32_t *>(&st.tp[pSlot]) + 3);
    };
    pre = claim(true, pSlot);
    if (pre) fetchRay();
    haveNext = claim(pre, nvs);
    bool active = false, fin = false;
    InstTrav r;
    r.phase = IP_DONE; r.farT = 0.0f; r.u = 0.0f; r.v = 0.0f; r.prim = PRIM_MISS; r.inst = INST_NONE;
    // The lane's next record is fetched as soon as it is known (right after a step / on arming) and consumed when the lane's KIND
    // of step is the one the wave runs in an iteration: mesh-BVH steps (IP_BOTTOM / IP_SCENE) or top-level steps (IP_TOP /
    // IP_ENTRY), whichever has more lanes ready -- otherwise the wave pays the instructions of all code paths in every step.
    float4 q0 = make_float4(0.0f, 0.0f, 0.0f, 0.0f), q1 = q0, q2 = q0, q3 = q0;
    auto fetchItem = [&]() {
        const float4 *item = instItem(sc, r);
        q0 = item[0]; q1 = item[1]; q2 = item[2]; q3 = item[3];
    };
    while (true) {
        {
            const bool isMesh = active && r.phase >= IP_BOTTOM, isTop = active && r.phase < IP_BOTTOM;
            const bool runMesh = __builtin_popcountll(__ballot(isMesh)) >= __builtin_popcountll(__ballot(isTop));
            if (runMesh ? isMesh : isTop) {
                instStep<COUNT>(sc, stack, topStack, r, tc, q0, q1, q2, q3);
                if (r.phase == IP_DONE) {
                    active = false;
                    fin = true;
                } else {
                    fetchItem();
                }
            }
        }
        const bool waiting = !active && (fin || pre);
        const unsigned long long w = __ballot(waiting), act = __ballot(active);
        if (w == 0 && act == 0) break;
        if (act != 0 && __builtin_popcountll(w) < TG_EXTQ_THRESH) continue;
        bool wantEntry = false;
        if (waiting) {
            if (fin) { // retire the ray of slot vs
                stStream<TG_STATE_NT_INST>(&st.hit[vs], make_float4(r.farT, r.u, r.v, __uint_as_float(r.prim)));
                stStream<TG_STATE_NT_INST>(&st.hitInst[vs], r.prim == PRIM_MISS ? INST_NONE : r.inst);
                fin = false;
            }
            if (pre) {
                vs = pSlot;
                if (flagsEnd(nflags) != END_DEAD) { // idle slots are k_classify's business
                    instInit(sc, r, xyz(nro), xyz(nrd), nro.w, nrd.w, false, PRIM_MISS);
                    if (r.phase == IP_DONE) {
                        fin = true;
                    } else {
                        active = true;
                        fetchItem();
                    }
                }
                if (haveNext) {
                    pSlot = nvs;
                    haveNext = false;
                    wantEntry = true;
                    fetchRay();
                } else {
                    pre = false;
                }
            }
        }
        uint32_t slot = 0;
        const bool got = claim(wantEntry, slot);
        if (wantEntry) {
            haveNext = got;
            nvs = slot;
        }
    }
    if (COUNT) flushCounters(st.stats, 0, tc);
}

// Sort by material after k_extend_iq: every entry of the input lists goes to the list of its hit's BSDF class (miss: list 0, path in
// a medium: the volume list, idle slot: the free list) of the other set -- the tail of k_extend_q as a pass of its own.
#ifdef TG_TU_MAIN
static __global__ void __launch_bounds__(256) k_classify(DScene sc, DState st, int set)
{
    const int lane = threadIdx.x & 63;
    const ListCounts lc = loadListCounts(st, set);
    const uint32_t count = listTotal(lc, 0);
    const uint32_t waves = gridDim.x*(blockDim.x >> 6), wave = blockIdx.x*(blockDim.x >> 6) + (threadIdx.x >> 6);
    const int out = set ^ 1;
    for (uint32_t base = wave*256u; base < count; base += waves*256u) {
        uint32_t p[4], cls4 = 0;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            p[q] = 0;
            uint32_t cls = 0xFEu;
            if (base + lane + 64u*q < count && listFetch(st, lc, set, 0, base + lane + 64u*q, p[q])) {
                if (flagsEnd(__float_as_uint(ldStream<TG_STATE_NT_INST>(&st.tp[p[q]]).w)) == END_DEAD) {
                    cls = 0xFDu;
                } else {
                    cls = hitClass(sc, __float_as_uint(ldStream<TG_STATE_NT_INST>(&st.hit[p[q]]).w));
                    if (sc.nMedia && (st.med[p[q]] & 0xFFu)) cls = TG_CLASS_VOLUME;
                }
            }
            cls4 |= cls << (8*q);
        }
#pragma unroll
        for (int l = 0; l <= TG_LISTS; ++l) {
            const uint32_t want = l == 0 ? 0xFFu : (l == TG_LISTS ? 0xFDu : uint32_t(l - 1));
            unsigned long long m0 = __ballot((cls4 & 0xFFu) == want), m1 = __ballot(((cls4 >> 8) & 0xFFu) == want);
            unsigned long long m2 = __ballot(((cls4 >> 16) & 0xFFu) == want), m3 = __ballot((cls4 >> 24) == want);
            uint32_t n0 = uint32_t(__builtin_popcountll(m0)), n1 = uint32_t(__builtin_popcountll(m1));
            uint32_t n2 = uint32_t(__builtin_popcountll(m2)), n3 = uint32_t(__builtin_popcountll(m3));
            uint32_t nn = n0 + n1 + n2 + n3;
            if (nn == 0) continue;
            uint32_t *dst = l == TG_LISTS ? st.freeQ : st.lists[out][l];
            uint32_t *ctr = l == TG_LISTS ? &st.counters[CNT(CNT_FREE)] : &st.counters[CNT(CNT_LIST0 + out*TG_LISTS + l)];
            uint32_t wbase = 0;
            if (lane == 0) wbase = atomicAdd(ctr, nn);
            wbase = __shfl(wbase, 0);
            const unsigned long long below = (1ull << lane) - 1ull;
            if ((m0 >> lane) & 1ull) stStream<TG_STATE_NT_INST>(&dst[wbase + uint32_t(__builtin_popcountll(m0 & below))], p[0]);
            if ((m1 >> lane) & 1ull) stStream<TG_STATE_NT_INST>(&dst[wbase + n0 + uint32_t(__builtin_popcountll(m1 & below))], p[1]);
            if ((m2 >> lane) & 1ull) stStream<TG_STATE_NT_INST>(&dst[wbase + n0 + n1 + uint32_t(__builtin_popcountll(m2 & below))], p[2]);
            if ((m3 >> lane) & 1ull) stStream<TG_STATE_NT_INST>(&dst[wbase + n0 + n1 + n2 + uint32_t(__builtin_popcountll(m3 & below))], p[3]);
        }
    }
}
#endif

template <bool COUNT, int DEPTH>
__global__ void __launch_bounds__(256, TG_INSTQ_WAVES) k_shadow_iq(DScene sc, DState st, int set)
{
    constexpr uint32_t CHUNK = 64u*DEPTH;
    extern __shared__ uint32_t ldsStack[];
    uint32_t *stack = ldsStack + threadIdx.x;
    uint2 *topStack = reinterpret_cast<uint2 *>(ldsStack + sc.stackLds*256) + threadIdx.x;
    const int lane = threadIdx.x & 63;
    const unsigned long long below = (1ull << lane) - 1ull;
    const ListCounts lc = loadListCounts(st, set);
    const uint32_t count = listTotal(lc, 1);
    TraceCounters tc{0, 0};
    uint32_t rays = 0, verts = 0;
    uint32_t chunkPos = 0, chunkEnd = 0;
    bool exhausted = false;
    auto claim = [&](bool want, uint32_t &slot) -> bool {
        bool got = false;
        unsigned long long m = __ballot(want);
        while (m) {
            if (chunkPos == chunkEnd) {
                if (exhausted) break;
                uint32_t b = 0;
                if (lane == 0) b = atomicAdd(&st.counters[CNT(CNT_HEAD_SHADOW)], CHUNK);
                b = uint32_t(__builtin_amdgcn_readfirstlane(int(b)));
                if (b >= count) {
                    exhausted = true;
                    break;
                }
                chunkPos = b;
                chunkEnd = min(b + CHUNK, count);
            }
            const uint32_t rank = uint32_t(__builtin_popcountll(m & below)), avail = chunkEnd - chunkPos;
            const bool mine = ((m >> lane) & 1ull) && rank < avail;
            if (mine) got = listFetch(st, lc, set, 1, chunkPos + rank, slot);
            chunkPos += min(uint32_t(__builtin_popcountll(m)), avail);
            m &= ~__ballot(mine);
        }
        return got;
    };
    uint32_t vs = 0, pSlot = 0, nvs = 0, ncap = PRIM_MISS, nHasB = 0, curCap = PRIM_MISS;
    int cw = 0, pWhich = 0;
    bool pre = false, haveNext = false, closestMode = false;
    float4 nro = make_float4(0.0f, 0.0f, 0.0f, 0.0f), nrd = nro;
    auto fetchRay = [&]() {
        nrd = ldStream<TG_STATE_NT_INST>(pWhich ? &st.shB[pSlot] : &st.shA[pSlot]);
        nro = ldStream<TG_STATE_NT_INST>(&st.shO[pSlot]);
        ncap = ldStream<TG_STATE_NT_INST>(reinterpret_cast<const uint32_t *>(&st.shX[pSlot]));
        nHasB = ldStream<TG_STATE_NT_INST>(reinterpret_cast<const uint32_t *>(&st.shX[pSlot]) + 1);
    };
    pre = claim(true, pSlot);
    if (pre) fetchRay();
    haveNext = claim(pre, nvs);
    bool active = false, fin = false;
    uint32_t vis = 0;
    bool last = true;
    auto complete = [&](bool unoccluded) {
        if (unoccluded) vis |= 1u << cw;
        if (last) {
            if (vis & 4u) {
                stStream<TG_STATE_NT_INST>(reinterpret_cast<uint32_t *>(&st.shD[vs]) + 3, vis & 3u);
                verts++;
            }
            vis = 0;
        }
    };
    InstTrav r;
    r.phase = IP_DONE; r.blocked = false; r.prim = PRIM_MISS; r.inst = INST_NONE;
    float4 q0 = make_float4(0.0f, 0.0f, 0.0f, 0.0f), q1 = q0, q2 = q0, q3 = q0; // the lane's next record (see k_extend_iq)
    auto fetchItem = [&]() {
        const float4 *item = instItem(sc, r);
        q0 = item[0]; q1 = item[1]; q2 = item[2]; q3 = item[3];
    };
    while (true) {
        {
            const bool isMesh = active && r.phase >= IP_BOTTOM, isTop = active && r.phase < IP_BOTTOM;
            const bool runMesh = __builtin_popcountll(__ballot(isMesh)) >= __builtin_popcountll(__ballot(isTop));
            if (runMesh ? isMesh : isTop) {
                instStep<COUNT>(sc, stack, topStack, r, tc, q0, q1, q2, q3);
                if (r.phase == IP_DONE) {
                    active = false;
                    fin = true;
                } else {
                    fetchItem();
                }
            }
        }
        const bool waiting = !active && (fin || pre);
        const unsigned long long w = __ballot(waiting), act = __ballot(active);
        if (w == 0 && act == 0) break;
        if (act != 0 && __builtin_popcountll(w) < TG_SHADOWQ_THRESH) continue;
        bool wantVertex = false;
        if (waiting) {
            if (fin) {
                // a finite light behind "instances": generalizedShadowRay's full closest-hit query (occluded<.., true>)
                const bool blocked = closestMode ? (r.prim != PRIM_MISS && !(r.inst == INST_NONE && r.prim == curCap)) : r.blocked;
                complete(!blocked);
                fin = false;
            }
            if (pre) {
                vs = pSlot;
                cw = pWhich;
                last = cw == 1 || !nHasB;
                if (nrd.w >= 0.0f) {
                    vis |= 4u;
                    rays++;
                    curCap = ncap;
                    closestMode = sc.nInstSets && ncap != PRIM_MISS && primKind(ncap) != KIND_ENV;
                    instInit(sc, r, xyz(nro), xyz(nrd), nro.w, nrd.w, !closestMode, ncap);
                    if (r.phase == IP_DONE) {
                        fin = true;
                    } else {
                        active = true;
                        fetchItem();
                    }
                } else {
                    complete(false);
                }
                if (!last) {
                    pWhich = 1;
                    fetchRay();
                } else if (haveNext) {
                    pSlot = nvs;
                    pWhich = 0;
                    haveNext = false;
                    wantVertex = true;
                    fetchRay();
                } else {
                    pre = false;
                }
            }
        }
        uint32_t slot = 0;
        const bool got = claim(wantVertex, slot);
        if (wantVertex) {
            haveNext = got;
            nvs = slot;
        }
    }
    if (COUNT) flushCounters(st.stats, 2, tc);
    for (int off = 32; off > 0; off >>= 1) {
        rays += __shfl_down(rays, off);
        verts += __shfl_down(verts, off);
    }
    if (lane == 0 && rays) atomicAdd(&st.stats[4], (unsigned long long)rays);
    if (lane == 0 && verts) atomicAdd(&st.stats[7], (unsigned long long)verts);
}

// raw any-hit for tgpu_trace_occluded
template <bool INST>
__global__ void __launch_bounds__(256) k_occluded_raw(DScene sc, DState st, uint32_t count, uint8_t *out)
{
    extern __shared__ uint32_t ldsStack[];
    uint32_t *stack = ldsStack + threadIdx.x;
    TraceCounters tc{0, 0};
    for (uint32_t i = blockIdx.x*blockDim.x + threadIdx.x; i < count; i += gridDim.x*blockDim.x) {
        float4 ro = st.rayO[i], rd = st.rayD[i];
        out[i] = occluded<false, INST>(sc, stack, xyz(ro), xyz(rd), ro.w, rd.w, PRIM_MISS, tc) ? 1 : 0;
    }
}

// =================================================================================================
// Tail kernel.  When no sample is left to start and only a few thousand long paths remain, finishing them
// with one launch per bounce costs ~0.4 ms of latency per iteration for ~60 iterations.  Instead one
// launch lets every remaining path run to its end in its own lane: extend -> miss | shade -> shadow ->
// ..., using exactly the per-path functions of the wavefront kernels (so the bits do not change), then
// retires it.  Every slot of the lists ends on the free list.
// =================================================================================================
// LEAN: the scene runs the lean shade kernels (no media, no sphere / disk, no forward-scattering surface, no auxiliary buffers: tgpu_upload_scene's
// generalShading is false) -- the tail then carries the lean shading code only.  The general kernel is 1.9 MB of code of which a lean scene's paths
// visit a scattered tenth; at one wave per SIMD every instruction-cache miss is exposed (1.6 M misses per launch, profiles/r03_ab_notes.txt).
template <bool COUNT, bool INST, bool FWD, bool LEAN = false>
__global__ void __launch_bounds__(256) k_tail(DScene sc, DState st, int set, uint32_t spread)
{
    static_assert(!(LEAN && FWD), "scenes with forward-scattering surfaces shade with the general kernels");
    extern __shared__ uint32_t ldsStack[];
    uint32_t *stack = ldsStack + threadIdx.x;
    const int lane = threadIdx.x & 63;
    const ListCounts lc = loadListCounts(st, set);
    const uint32_t total = listTotal(lc, 0);
    const uint32_t stride = gridDim.x*blockDim.x;
    const uint32_t rounded = (total + 63u) & ~63u;
    TraceCounters tcE{0, 0}, tcS{0, 0};
    uint32_t extRays = 0, shRays = 0, events = 0;
    // only every `spread`-th lane takes a path: a wave lasts as long as the slowest of its 64 / spread paths, and the paths
    // spread over more waves per SIMD, which is what the last few thousand paths of a render need (latency, not throughput)
    for (uint32_t g = blockIdx.x*blockDim.x + threadIdx.x; g < rounded*spread; g += stride) {
        const uint32_t q = g/spread;
        uint32_t slot = 0, ring = 0xFFFFFFFFu;
        const bool have = g % spread == 0 && q < total && listFetch(st, lc, set, 0, q, slot);
        if (have) {
            uint32_t vertex = 0; // COUNT: when (10 ns ticks since this lane started) the slowest path reached its n-th vertex -> stats[8 + n]
            const unsigned long long t0 = COUNT ? wall_clock64() : 0ull;
            while (flagsEnd(__float_as_uint(st.tp[slot].w)) == END_ALIVE) {
                if (COUNT) {
                    atomicMax(&st.stats[8 + min(vertex, 127u)], wall_clock64() - t0);
                    ++vertex;
                }
                float4 ro = st.rayO[slot], rd = st.rayD[slot];
                float tfar = rd.w, u = 0.0f, v = 0.0f;
                uint32_t inst;
                uint32_t prim = closestHit<COUNT, INST>(sc, stack, xyz(ro), xyz(rd), ro.w, tfar, u, v, inst, tcE);
                if (INST) st.hitInst[slot] = inst;
                st.hit[slot] = make_float4(tfar, u, v, __uint_as_float(prim));
                extRays++;
                if (!LEAN && sc.nMedia && (st.med[slot] & 0xFFu)) {
                    volumePath(sc, st, slot);
                    events++;
                    if (stateNee(st.state[slot])) shadowPath<COUNT, INST, FWD>(sc, st, stack, slot, tcS, shRays);
                } else if (prim == PRIM_MISS) {
                    missPath(sc, st, slot);
                } else {
                    shadePath<-1, !LEAN>(sc, st, slot);
                    events++;
                    if (stateNee(st.state[slot])) shadowPath<COUNT, INST, FWD>(sc, st, stack, slot, tcS, shRays);
                }
            }
            float4 tpw = st.tp[slot];
            int end = flagsEnd(__float_as_uint(tpw.w));
            if (end != END_DEAD) { // retire (k_retire_regenerate's rules)
                V3 c = xyz(st.em[slot]);
                const bool nanEnd = end == END_CHECKED && isnanf_(xyz(tpw).sum() + c.sum());
                if (end == END_ZERO || nanEnd) c = V3(0.0f);
                uint32_t ref = st.sampleRef[slot];
                if (!LEAN && st.aov0) aovFinish(sc, st, slot, ref, end == END_CHECKED && !nanEnd);
                st.results[ref] = mk4(c, 0.0f);
                ring = ref/st.resultSlice;
                st.tp[slot] = make_float4(tpw.x, tpw.y, tpw.z, __uint_as_float(packFlags(0, false, END_DEAD)));
                st.state[slot] = packState(false, END_DEAD);
            }
        }
        unsigned long long todo = __ballot(ring != 0xFFFFFFFFu);
        while (todo) {
            int leader = __builtin_ctzll(todo);
            uint32_t r = __shfl(ring, leader);
            unsigned long long m = __ballot(ring == r);
            if (lane == leader) atomicAdd(&st.counters[CNT(CNT_JOB_DONE0 + r)], uint32_t(__builtin_popcountll(m)));
            todo &= ~m;
        }
        queueAppend(have, st.freeQ, &st.counters[CNT(CNT_FREE)], slot);
    }
    if (COUNT) {
        flushCounters(st.stats, 0, tcE);
        flushCounters(st.stats, 2, tcS);
    }
    for (int off = 32; off > 0; off >>= 1) {
        extRays += __shfl_down(extRays, off);
        shRays += __shfl_down(shRays, off);
        events += __shfl_down(events, off);
    }
    if (lane == 0) {
        if (shRays) atomicAdd(&st.stats[4], (unsigned long long)shRays);
        if (events) atomicAdd(&st.stats[5], (unsigned long long)events);
        if (extRays) atomicAdd(&st.stats[6], (unsigned long long)extRays);
    }
}

// =================================================================================================
// The tail kernel of scenes with plain geometry and lean shading (no instances / spheres / disks, no media, no forward-scattering surface, no
// auxiliary buffers -- the benchmark's kind), four lanes per path: lane 0 walks the path, lanes 1 and 2 trace the two NEE rays of the path's
// PREVIOUS vertex while lane 0 traces the continuation ray, all in one traversal loop.  In k_tail a bounce costs the path's lane three
// traversals back to back (closest hit, shadow ray, shadow ray), and at the end of a render nothing hides that latency; here it costs the
// longest of the three.  The arithmetic is the wavefront's: the helpers leave the visibility bits in shD.w as k_shadow_q does, and whoever
// touches the path next (shadePath, missPathEx, the retire code below) folds the NEE terms in with resolveNeeTerms -- same values, same order.
// =================================================================================================
// NESTED = false: the scene has no BSDF made of other BSDFs -- their code and its out-of-line calls stay out (bsdfEval's NESTED, device.h).
template <bool COUNT, bool NESTED>
__global__ void __launch_bounds__(256) k_tail_coop(DScene sc, DState st, int set)
{
    extern __shared__ uint32_t ldsStack[];
    uint32_t *stack = ldsStack + threadIdx.x;
    const int lane = threadIdx.x & 63, role = lane & 3, leader = lane & ~3;
    const ListCounts lc = loadListCounts(st, set);
    const uint32_t total = listTotal(lc, 0);
    const uint32_t waveIndex = (blockIdx.x*blockDim.x + threadIdx.x) >> 6, waves = (gridDim.x*blockDim.x) >> 6;
    TraceCounters tc{0, 0};
    uint32_t extRays = 0, shRays = 0, events = 0;
    for (uint32_t base = waveIndex*16u; base < total; base += waves*16u) { // wave-uniform trip count: 16 paths per wave and pass
        const uint32_t q = base + uint32_t(lane >> 2);
        uint32_t slot = 0, ring = 0xFFFFFFFFu;
        bool have = false;
        if (role == 0 && q < total) have = listFetch(st, lc, set, 0, q, slot);
        have = __shfl(int(have), leader) != 0;
        slot = __shfl(slot, leader);
        // alive: the path has a ray to trace; pending: its last vertex's NEE rays are still to be traced (group-uniform)
        bool alive = have && flagsEnd(__float_as_uint(st.tp[slot].w)) == END_ALIVE, pending = false;
        uint32_t vertex = 0; // COUNT: when (10 ns ticks since this lane started) the slowest path reached its n-th vertex -> stats[8 + n], as in k_tail
        const unsigned long long t0 = COUNT ? wall_clock64() : 0ull;
        while (__ballot(alive || pending)) {
            if (COUNT && role == 0 && alive) {
                atomicMax(&st.stats[8 + min(vertex, 127u)], wall_clock64() - t0);
                ++vertex;
            }
            TravState r;
            r.node = 0; r.sp = 0; r.bestT = 0.0f; r.bestU = 0.0f; r.bestV = 0.0f; r.bestPrim = PRIM_MISS;
            bool active = false, hasRay = false, blocked = false;
            if (role == 0 && alive) { // closestHit's order: quads, cubes, triangles
                const float4 ro = st.rayO[slot], rd = st.rayD[slot];
                const V3 O = xyz(ro), D = xyz(rd);
                float tfar = rd.w, u = 0.0f, v = 0.0f;
                uint32_t prim = PRIM_MISS;
                for (uint32_t i = 0; i < sc.nQuads; ++i) {
                    QuadHit qh;
                    if (quadIntersect(sc.quads[i], O, D, ro.w, tfar, qh)) {
                        tfar = qh.t; u = qh.u; v = qh.v;
                        prim = mkPrim(KIND_QUAD, i);
                    }
                }
                for (uint32_t i = 0; i < sc.nCubes; ++i) {
                    float t;
                    bool back;
                    if (cubeIntersect(sc.cubes[i], O, D, ro.w, tfar, t, back)) {
                        tfar = t; u = 0.0f; v = 0.0f;
                        prim = mkPrim(KIND_CUBE, i);
                    }
                }
                travInit(r, sc.rootChild, O, D, ro.w, tfar, prim);
                r.bestU = u; r.bestV = v;
                active = sc.nTris != 0;
            } else if ((role == 1 || role == 2) && pending) { // occluded()'s order: quads (not the end cap), cubes, triangles
                const float4 rd = ldStream(role == 1 ? &st.shA[slot] : &st.shB[slot]), ro = ldStream(&st.shO[slot]);
                const uint32_t endCap = ldStream(reinterpret_cast<const uint32_t *>(&st.shX[slot]));
                if (rd.w >= 0.0f) {
                    hasRay = true;
                    const V3 O = xyz(ro), D = xyz(rd);
                    for (uint32_t i = 0; i < sc.nQuads; ++i) {
                        QuadHit qh;
                        if (mkPrim(KIND_QUAD, i) != endCap && quadIntersect(sc.quads[i], O, D, ro.w, rd.w, qh)) blocked = true;
                    }
                    for (uint32_t i = 0; i < sc.nCubes; ++i) {
                        float t;
                        bool back;
                        if (cubeIntersect(sc.cubes[i], O, D, ro.w, rd.w, t, back)) blocked = true;
                    }
                    if (!blocked && sc.nTris) {
                        travInit(r, sc.rootChild, O, D, ro.w, rd.w, PRIM_MISS);
                        active = true;
                    }
                }
            }
            while (__ballot(active)) {
                if (active) {
                    const float4 *n = travItem(sc, r);
                    const float4 q0 = n[0], q1 = n[1], q2 = n[2], q3 = n[3];
                    const int s = travApply<false, COUNT, true, true>(sc, stack, r, tc, q0, q1, q2, q3, role != 0);
                    if (s != TRAV_MORE) {
                        active = false;
                        blocked = blocked || s == TRAV_ANYHIT;
                    }
                }
            }
            // the NEE rays' answers -> the visibility word of the vertex (k_shadow_q's `complete`)
            const uint32_t clearBit = hasRay && !blocked ? 1u : 0u;
            const uint32_t visA = __shfl(clearBit, leader + 1), visB = __shfl(clearBit, leader + 2);
            const bool rayA = __shfl(int(hasRay), leader + 1) != 0, rayB = __shfl(int(hasRay), leader + 2) != 0;
            if (hasRay) shRays++;
            if (role == 0 && pending && (rayA || rayB)) reinterpret_cast<uint32_t *>(&st.shD[slot])[3] = visA | (visB << 1);
            bool nextAlive = false, nextPending = false;
            if (role == 0 && alive) {
                st.hit[slot] = make_float4(r.bestT, r.bestU, r.bestV, __uint_as_float(r.bestPrim));
                extRays++;
                if (r.bestPrim == PRIM_MISS) {
                    missPath(sc, st, slot);
                } else {
                    shadePath<-1, false, NESTED>(sc, st, slot);
                    events++;
                }
                nextAlive = flagsEnd(__float_as_uint(st.tp[slot].w)) == END_ALIVE;
                nextPending = stateNee(st.state[slot]);
            }
            alive = __shfl(int(nextAlive), leader) != 0;
            pending = __shfl(int(nextPending), leader) != 0;
        }
        if (role == 0 && have) {
            const float4 tpw = st.tp[slot];
            const int end = flagsEnd(__float_as_uint(tpw.w));
            if (end != END_DEAD) { // retire (k_retire_regenerate's rules); the last vertex's NEE terms come in here
                V3 c = xyz(st.em[slot]);
                if (stateNee(st.state[slot])) c = resolveNeeTerms<false>(st, slot, c);
                const bool nanEnd = end == END_CHECKED && isnanf_(xyz(tpw).sum() + c.sum());
                if (end == END_ZERO || nanEnd) c = V3(0.0f);
                const uint32_t ref = st.sampleRef[slot];
                st.results[ref] = mk4(c, 0.0f);
                ring = ref/st.resultSlice;
                st.tp[slot] = make_float4(tpw.x, tpw.y, tpw.z, __uint_as_float(packFlags(0, false, END_DEAD)));
                st.state[slot] = packState(false, END_DEAD);
            }
        }
        unsigned long long todo = __ballot(ring != 0xFFFFFFFFu);
        while (todo) {
            const int first = __builtin_ctzll(todo);
            const uint32_t rr = __shfl(ring, first);
            const unsigned long long m = __ballot(ring == rr);
            if (lane == first) atomicAdd(&st.counters[CNT(CNT_JOB_DONE0 + rr)], uint32_t(__builtin_popcountll(m)));
            todo &= ~m;
        }
        queueAppend(role == 0 && have, st.freeQ, &st.counters[CNT(CNT_FREE)], slot);
    }
    if (COUNT) {
        TraceCounters tcE = tc, tcS = tc;
        if (role != 0) tcE.nodes = tcE.tris = 0;
        if (role == 0) tcS.nodes = tcS.tris = 0;
        flushCounters(st.stats, 0, tcE);
        flushCounters(st.stats, 2, tcS);
    }
    for (int off = 32; off > 0; off >>= 1) {
        extRays += __shfl_down(extRays, off);
        shRays += __shfl_down(shRays, off);
        events += __shfl_down(events, off);
    }
    if (lane == 0) {
        if (shRays) atomicAdd(&st.stats[4], (unsigned long long)shRays);
        if (events) atomicAdd(&st.stats[5], (unsigned long long)events);
        if (extRays) atomicAdd(&st.stats[6], (unsigned long long)extRays);
    }
}

// =================================================================================================
// K7 accumulate: OutputBuffer::addSample in sample order (cameras/OutputBuffer.hpp:104-132)
// =================================================================================================
#ifdef TG_TU_MAIN
// One pixel of one OutputBuffer in registers.  Buffers of one channel use .x only (the arithmetic is elementwise).
struct ObAcc { V3 a, b, var; uint32_t n; };
TG_DEV void obLoad(const DOutBuf &ob, int channels, uint32_t k, ObAcc &s)
{
    s.a = s.b = s.var = V3(0.0f);
    s.n = ob.count[k];
    if (channels == 3) {
        s.a = V3(ob.a + size_t(k)*3);
        if (ob.b) s.b = V3(ob.b + size_t(k)*3);
        if (ob.var) s.var = V3(ob.var + size_t(k)*3);
    } else {
        s.a.x = ob.a[k];
        if (ob.b) s.b.x = ob.b[k];
        if (ob.var) s.var.x = ob.var[k];
    }
}
TG_DEV void obStore(const DOutBuf &ob, int channels, uint32_t k, const ObAcc &s)
{
    ob.count[k] = s.n;
    for (int c = 0; c < channels; ++c) {
        const float a = c == 0 ? s.a.x : (c == 1 ? s.a.y : s.a.z), b = c == 0 ? s.b.x : (c == 1 ? s.b.y : s.b.z), v = c == 0 ? s.var.x : (c == 1 ? s.var.y : s.var.z);
        ob.a[size_t(k)*channels + c] = a;
        if (ob.b) ob.b[size_t(k)*channels + c] = b;
        if (ob.var) ob.var[size_t(k)*channels + c] = v;
    }
}
TG_DEV void obAdd(const DOutBuf &ob, ObAcc &s, const V3 &c) // OutputBuffer::addSample (cameras/OutputBuffer.hpp:104-132)
{
    if (isnanf_(c.x) || isnanf_(c.y) || isnanf_(c.z) || isinff_(c.x) || isinff_(c.y) || isinff_(c.z)) return;
    const uint32_t sampleIdx = s.n++;
    if (ob.var) {
        V3 curr = s.a;
        if (ob.b && sampleIdx > 0) {
            const uint32_t sampleCountA = (sampleIdx + 1u)/2u, sampleCountB = sampleIdx/2u;
            curr = (s.a*float(sampleCountA) + s.b*float(sampleCountB))/float(sampleIdx);
        }
        const V3 delta = c - curr;
        curr += delta/float(sampleIdx + 1u);
        s.var += delta*(c - curr);
    }
    if (ob.b) {
        const float perBufferSampleCount = float(sampleIdx/2u + 1u);
        if (sampleIdx & 1u) s.b += (c - s.b)/perBufferSampleCount;
        else s.a += (c - s.a)/perBufferSampleCount;
    } else {
        s.a += (c - s.a)/float(sampleIdx + 1u);
    }
}
// what sample `idx` of the ring slice adds to buffer `type` (one-channel values in .x)
TG_DEV V3 outputSample(const DState &st, int type, size_t idx)
{
    if (type == TGPU_OUTPUT_COLOR) return xyz(st.results[idx]);
    if (type == TGPU_OUTPUT_DEPTH) return V3(st.aov0[idx].x, 0.0f, 0.0f);
    if (type == TGPU_OUTPUT_NORMAL) { const float4 v = st.aov0[idx]; return V3(v.y, v.z, v.w); }
    if (type == TGPU_OUTPUT_ALBEDO) return xyz(st.aov1[idx]);
    return V3(st.aov1[idx].w, 0.0f, 0.0f);
}
TG_DEV int outputChannels(int type) { return (type == TGPU_OUTPUT_DEPTH || type == TGPU_OUTPUT_VISIBILITY) ? 1 : 3; }
// pixel k's samples of the job into every output buffer from `firstType` on: sample s sits at base + first + s*step
TG_DEV void accumulateOutputs(const DState &st, int firstType, uint32_t k, size_t base, size_t first, size_t step, uint32_t count)
{
    for (int type = firstType; type < TGPU_OUTPUT_COUNT; ++type) {
        const DOutBuf &ob = st.out[type];
        if (!ob.a) continue;
        const int channels = outputChannels(type);
        ObAcc acc;
        obLoad(ob, channels, k, acc);
        for (uint32_t s = 0; s < count; ++s) obAdd(ob, acc, outputSample(st, type, base + first + size_t(s)*step));
        obStore(ob, channels, k, acc);
    }
}

static __global__ void __launch_bounds__(256) k_accumulate(DState st, uint32_t ringSlot, uint32_t sppCount)
{
    const float4 *res = st.results + size_t(ringSlot)*st.resultSlice;
    const bool plainColor = !st.out[TGPU_OUTPUT_COLOR].b && !st.out[TGPU_OUTPUT_COLOR].var;
    for (uint32_t k = blockIdx.x*blockDim.x + threadIdx.x; k < st.nPixels; k += gridDim.x*blockDim.x) {
        if (plainColor) {
            V3 mean(st.fbMean + size_t(k)*3);
            uint32_t n = st.fbCount[k];
            for (uint32_t s = 0; s < sppCount; ++s) {
                V3 c = xyz(res[size_t(s)*st.nPixels + k]);
                if (isnanf_(c.x) || isnanf_(c.y) || isnanf_(c.z) || isinff_(c.x) || isinff_(c.y) || isinff_(c.z)) continue;
                mean = mean + (c - mean)/float(n + 1u);
                n++;
            }
            st.fbMean[size_t(k)*3 + 0] = mean.x;
            st.fbMean[size_t(k)*3 + 1] = mean.y;
            st.fbMean[size_t(k)*3 + 2] = mean.z;
            st.fbCount[k] = n;
        }
        if (!plainColor || st.aov0) accumulateOutputs(st, plainColor ? 1 : 0, k, size_t(ringSlot)*st.resultSlice, k, st.nPixels, sppCount);
    }
}
#endif

// Accumulate with SampleRecords (adaptive_sampling): one lane per variance tile replays renderTile's order --
// pixels of the 4x4 block row-major, the pixel's samples in sequence -- so that the Welford statistics of
// SampleRecord::addSample (SampleRecord.hpp:44-55) and the per-pixel means come out with the reference's bits.
// Uniform jobs (offset == nullptr) hold sample s of pixel k at s*nPixels + k, adaptive jobs at offset[k] + s.
#ifdef TG_TU_MAIN
static __global__ void __launch_bounds__(256) k_accumulate_records(DState st, uint32_t ringSlot, uint32_t sppCount, uint32_t adaptive)
{
    const float4 *res = st.results + size_t(ringSlot)*st.resultSlice;
    const uint32_t *off = st.jobOffset + size_t(ringSlot)*(st.nPixels + 1u);
    const bool plainColor = !st.out[TGPU_OUTPUT_COLOR].b && !st.out[TGPU_OUTPUT_COLOR].var;
    for (uint32_t vt = blockIdx.x*blockDim.x + threadIdx.x; vt < st.nVt; vt += gridDim.x*blockDim.x) {
        uint32_t sampleCount = st.recCount[vt];
        float mean = st.recMean[vt], runningVariance = st.recVar[vt];
        for (uint32_t j = 0; j < 16u; ++j) {
            uint32_t k = st.vtPix[size_t(vt)*16u + j];
            if (k == 0xFFFFFFFFu) continue;
            V3 pm(st.fbMean + size_t(k)*3);
            uint32_t n = st.fbCount[k];
            uint32_t first = adaptive ? off[k] : 0u, cnt = adaptive ? off[k + 1u] - off[k] : sppCount;
            for (uint32_t s = 0; s < cnt; ++s) {
                V3 c = xyz(res[adaptive ? size_t(first + s) : size_t(s)*st.nPixels + k]);
                float x = c.x*0.2126f + c.y*0.7152f + c.z*0.0722f; // Vec::luminance (math/Vec.hpp:195-199)
                sampleCount++;
                float delta = x - mean;
                mean += delta/float(sampleCount);
                runningVariance += delta*(x - mean);
                if (!plainColor || isnanf_(c.x) || isnanf_(c.y) || isnanf_(c.z) || isinff_(c.x) || isinff_(c.y) || isinff_(c.z)) continue;
                pm = pm + (c - pm)/float(n + 1u);
                n++;
            }
            if (plainColor) {
                st.fbMean[size_t(k)*3 + 0] = pm.x;
                st.fbMean[size_t(k)*3 + 1] = pm.y;
                st.fbMean[size_t(k)*3 + 2] = pm.z;
                st.fbCount[k] = n;
            }
            if (!plainColor || st.aov0)
                accumulateOutputs(st, plainColor ? 1 : 0, k, size_t(ringSlot)*st.resultSlice, adaptive ? size_t(first) : size_t(k), adaptive ? 1u : st.nPixels, cnt);
        }
        st.recCount[vt] = sampleCount;
        st.recMean[vt] = mean;
        st.recVar[vt] = runningVariance;
    }
}
#endif

// device evaluation of include/tgpu_math.h for the bit-exactness test
#ifdef TG_TU_MAIN
static __global__ void k_eval_math(int kind, const float *a, const float *b, uint32_t n, float *out)
{
    for (uint32_t i = blockIdx.x*blockDim.x + threadIdx.x; i < n; i += gridDim.x*blockDim.x) {
        float x = a[i], y = b[i], r;
        switch (kind) {
        case 0: r = tgpu_sinf(x); break;
        case 1: r = tgpu_cosf(x); break;
        case 2: r = tgpu_expf(x); break;
        case 3: r = tgpu_logf(x); break;
        case 4: r = tgpu_atan2f(x, y); break;
        case 5: r = tgpu_acosf(x); break;
        case 6: r = x/y; break;
        case 7: r = __builtin_sqrtf(x); break;
        default: r = 0.0f; break;
        }
        out[i] = r;
    }
}
#endif

// =================================================================================================
// Translation units.  The big kernel families are instantiated in files of their own (tu_*.hip) so that the library
// builds in parallel; tgpu_api.hip sees them as `extern template`.  PFX is `extern` or empty.
// =================================================================================================
#define TG_KERNELS_SHADE(PFX, M) \
    PFX template __global__ void k_shade<0, M>(DScene, DState, int); \
    PFX template __global__ void k_shade<1, M>(DScene, DState, int); \
    PFX template __global__ void k_shade<2, M>(DScene, DState, int); \
    PFX template __global__ void k_shade<3, M>(DScene, DState, int); \
    PFX template __global__ void k_shade<4, M>(DScene, DState, int); \
    PFX template __global__ void k_shade<5, M>(DScene, DState, int); \
    PFX template __global__ void k_shade<TG_CLASS_OTHER, M>(DScene, DState, int);
#define TG_KERNELS_INSTQ(PFX) \
    PFX template __global__ void k_extend_iq<false, 4>(DScene, DState, int); \
    PFX template __global__ void k_extend_iq<true, 4>(DScene, DState, int); \
    PFX template __global__ void k_shadow_iq<false, 4>(DScene, DState, int); \
    PFX template __global__ void k_shadow_iq<true, 4>(DScene, DState, int);
#define TG_KERNELS_SHADOW(PFX) \
    PFX template __global__ void k_shadow_q<false, 4>(DScene, DState, int); \
    PFX template __global__ void k_shadow_q<true, 4>(DScene, DState, int); \
    PFX template __global__ void k_shadow_q<false, 8>(DScene, DState, int); \
    PFX template __global__ void k_shadow_q<true, 8>(DScene, DState, int); \
    PFX template __global__ void k_shadow<false, false, false>(DScene, DState, int); \
    PFX template __global__ void k_shadow<false, false, true>(DScene, DState, int); \
    PFX template __global__ void k_shadow<false, true, false>(DScene, DState, int); \
    PFX template __global__ void k_shadow<false, true, true>(DScene, DState, int); \
    PFX template __global__ void k_shadow<true, false, false>(DScene, DState, int); \
    PFX template __global__ void k_shadow<true, false, true>(DScene, DState, int); \
    PFX template __global__ void k_shadow<true, true, false>(DScene, DState, int); \
    PFX template __global__ void k_shadow<true, true, true>(DScene, DState, int);
#define TG_KERNELS_TAIL(PFX, C, I, F) PFX template __global__ void k_tail<C, I, F>(DScene, DState, int, uint32_t);
#define TG_KERNELS_TAIL_LEAN(PFX, C, I) PFX template __global__ void k_tail<C, I, false, true>(DScene, DState, int, uint32_t);
#define TG_KERNELS_TAIL_COOP(PFX, C, N) PFX template __global__ void k_tail_coop<C, N>(DScene, DState, int);

} // namespace tg
