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
enum { CNT_EXT_NEXT = 0, CNT_MISS = 1, CNT_CLASS0 = 2, CNT_SHADOW = 2 + TGPU_BSDF_NUM_CLASSES,
       CNT_HEAD_EXT = 3 + TGPU_BSDF_NUM_CLASSES, CNT_HEAD_SHADOW = 4 + TGPU_BSDF_NUM_CLASSES,
       CNT_END = 5 + TGPU_BSDF_NUM_CLASSES, CNT_REGEN = 6 + TGPU_BSDF_NUM_CLASSES,
       CNT_PER_ITER = 7 + TGPU_BSDF_NUM_CLASSES /* everything below is zeroed every iteration */,
       CNT_FREE = CNT_PER_ITER /* persistent: length of the free-slot list */,
       CNT_JOB_DONE0 = CNT_FREE + 1 /* TG_RING persistent per-job retire counters */, CNT_TOTAL = 32 };
// Result-ring slices (= jobs that may be in flight).  A job can only be folded into the framebuffer
// once its slowest path has ended (~max_bounces iterations after its last sample started), and during
// that time the pool keeps starting samples of later jobs: the ring must hold ~26 x pool size samples.
#define TG_RING 16
static_assert(CNT_JOB_DONE0 + TG_RING <= CNT_TOTAL, "counter block too small");

// Samples are handed out in global order: job by job, inside a job sample-major then shard pixel
// (coalesced accumulate).  A job = one spp range over all shard pixels (one tgpu_render_pass chunk).
struct JobRing {
    unsigned long long start[TG_RING]; // global index of the job's first sample
    uint32_t sppBegin[TG_RING];
    uint32_t sppCount[TG_RING];
    uint32_t ringSlot[TG_RING];        // which slice of the result ring the job writes
    uint32_t n;
};

enum { END_ALIVE = 0, END_RAW = 1, END_ZERO = 2, END_CHECKED = 3 };
// flags word (bits of tp.w): [15:0] bounce, [16] wasSpecular, [27:24] end state
TG_DEV uint32_t packFlags(int bounce, bool wasSpecular, int end) { return uint32_t(bounce) | (wasSpecular ? 1u << 16 : 0u) | (uint32_t(end) << 24); }
TG_DEV int flagsBounce(uint32_t f) { return int(f & 0xFFFFu); }
TG_DEV bool flagsSpecular(uint32_t f) { return (f >> 16) & 1u; }
TG_DEV int flagsEnd(uint32_t f) { return int((f >> 24) & 0xFu); }

struct DState {
    float4 *rayO;   // origin.xyz, tnear
    float4 *rayD;   // dir.xyz, tfar
    float4 *hit;    // t, u, v, bits(prim)
    float4 *tp;     // throughput.xyz, bits(flags)
    float4 *em;     // emission (radiance so far).xyz, bits(Sobol scramble)
    uint4 *samp;    // sobol dimension, pcg lo, pcg hi, sample index
    // NEE records of the path's current vertex (origin = rayO.xyz, tnear = epsilon)
    float4 *shA;    // light-sample dir.xyz, tfar (< 0: no ray)
    float4 *shB;    // bsdf-sample dir.xyz, tfar (< 0: no ray)
    float4 *shC;    // L.xyz (unoccluded light-sample term), light-pick weight
    float4 *shD;    // B.xyz (unoccluded bsdf-sample term), bits(end cap prim | postAdd flag << 31 hmm)
    float4 *shE;    // throughput at the vertex.xyz, bits(has post-add)
    float4 *shF;    // emitter term to add after the NEE term.xyz
    uint32_t *sampleRef; // per slot: index into `results` of the sample the slot is tracing
    float4 *results;     // TG_RING x nPixels x resultSpp: final radiance of every sample (w unused)
    uint32_t *extQ[2];
    uint32_t *classQ[TGPU_BSDF_NUM_CLASSES];
    uint32_t *missQ;
    uint32_t *shadowQ;
    uint32_t *endQ;      // slots whose path ended this iteration
    uint32_t *freeQ;     // idle slots: ended while no sample was admitted (ring full / end of render)
    uint32_t *counters;
    uint32_t nPixels;
    uint32_t resultSlice; // samples per ring slot = nPixels * resultSpp
    unsigned long long *stats; // [0] ext nodes [1] ext tris [2] shadow nodes [3] shadow tris [4] shadow rays [5] surface events
    // shard
    const uint32_t *pixelIndex;
    const uint32_t *scramble;
    float *fbMean;
    uint32_t *fbCount;
};

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
    uint32_t j = 0;
    for (uint32_t t = 1; t < jobs.n; ++t)
        if (g >= jobs.start[t]) j = t;
    uint32_t e = uint32_t(g - jobs.start[j]);
    uint32_t s = e/st.nPixels, k = e - s*st.nPixels;
    uint32_t pixel = st.pixelIndex[k];
    Sampler smp;
    smp.matrices = sc.sobol;
    smp.sobol = sc.settings.use_sobol;
    smp.scramble = st.scramble[k];
    smp.index = jobs.sppBegin[j] + s;
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
    st.sampleRef[slot] = jobs.ringSlot[j]*st.resultSlice + e;
}

// Retire + regenerate, for every slot in the end queue:
//  1. retire: the finished path's radiance, with traceSample's exit rules applied (PathTracer.cpp:119-131),
//     goes to the result ring; the job's completion counter is bumped;
//  2. regenerate: the slot takes the next sample in global order if one is admitted (< limit) and joins the
//     next extend queue; otherwise it goes to the free list.
__global__ void __launch_bounds__(256) k_retire_regenerate(DScene sc, DState st, JobRing jobs, unsigned long long sampleBase,
                                                           unsigned long long limit, uint32_t *extNext)
{
    const uint32_t count = st.counters[CNT_END];
    const uint32_t stride = gridDim.x*blockDim.x;
    const uint32_t rounded = (count + 63u) & ~63u;
    const int lane = threadIdx.x & 63;
    for (uint32_t q = blockIdx.x*blockDim.x + threadIdx.x; q < rounded; q += stride) {
        const bool active = q < count;
        uint32_t slot = 0, ring = 0xFFFFFFFFu;
        if (active) {
            slot = st.endQ[q];
            float4 tpw = st.tp[slot], emw = st.em[slot];
            V3 c = xyz(emw);
            int end = flagsEnd(__float_as_uint(tpw.w));
            if (end == END_ZERO) c = V3(0.0f);
            else if (end == END_CHECKED && isnanf_(xyz(tpw).sum() + c.sum())) c = V3(0.0f); // PathTracer.cpp:130-131
            uint32_t ref = st.sampleRef[slot];
            st.results[ref] = mk4(c, 0.0f);
            ring = ref/st.resultSlice;
        }
        // one atomic per distinct job present in the wave
        unsigned long long todo = __ballot(active);
        while (todo) {
            int leader = __builtin_ctzll(todo);
            uint32_t r = __shfl(ring, leader);
            unsigned long long m = __ballot(active && ring == r);
            if (lane == leader) atomicAdd(&st.counters[CNT_JOB_DONE0 + r], uint32_t(__builtin_popcountll(m)));
            todo &= ~m;
        }
        // claim the next samples for the active lanes of this wave
        unsigned long long mask = __ballot(active);
        unsigned long long g = 0;
        if (mask) {
            int leader = __builtin_ctzll(mask);
            uint32_t base = 0;
            if (lane == leader) base = atomicAdd(&st.counters[CNT_REGEN], uint32_t(__builtin_popcountll(mask)));
            base = __shfl(base, leader);
            g = sampleBase + base + uint32_t(__builtin_popcountll(mask & ((1ull << lane) - 1ull)));
        }
        bool started = active && g < limit;
        if (started) startSample(sc, st, jobs, g, slot);
        queueAppend(started, extNext, &st.counters[CNT_EXT_NEXT], slot);
        queueAppend(active && !started, st.freeQ, &st.counters[CNT_FREE], slot);
    }
}

// Refill from the free list: the host knows how many samples are admitted, so slot freeQ[freeBase + q]
// takes sample sampleBase + q and is written straight into the current extend queue at queueBase + q.
__global__ void __launch_bounds__(256) k_refill(DScene sc, DState st, JobRing jobs, unsigned long long sampleBase, uint32_t nStart,
                                                uint32_t freeBase, uint32_t *queue, uint32_t queueBase)
{
    for (uint32_t q = blockIdx.x*blockDim.x + threadIdx.x; q < nStart; q += gridDim.x*blockDim.x) {
        uint32_t slot = st.freeQ[freeBase + q];
        startSample(sc, st, jobs, sampleBase + q, slot);
        queue[queueBase + q] = slot;
    }
    if (blockIdx.x == 0 && threadIdx.x == 0) st.counters[CNT_FREE] = freeBase;
}

// =================================================================================================
// BVH traversal (K2 / K4 core)
// =================================================================================================
struct TraceCounters { uint32_t nodes, tris; };

// Embree MoellerTrumboreIntersector1 (thirdparty/embree/kernels/geometry/triangle_intersector_moeller.h:76-113)
// with IEEE division in finalize(); dot = a.x*b.x + (a.y*b.y + a.z*b.z) (common/math/vec3.h:182)
TG_DEV float edot(const V3 &a, const V3 &b) { return a.x*b.x + (a.y*b.y + a.z*b.z); }

TG_DEV bool triTest(const float4 &t0, const float4 &t1, const float4 &t2, const V3 &O, const V3 &D, float tnear, float tfar0,
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
    return true;
}

// Closest (ANY = false): lexicographic minimum of (t, soup triangle id) over all valid triangles.
// Any-hit (ANY = true): returns true at the first valid triangle.
template <bool ANY, bool COUNT>
TG_DEV bool traverse(const DScene &sc, uint32_t *stack, const V3 &O, const V3 &D, float tnear, float &bestT, float &bestU, float &bestV,
                     uint32_t &bestPrim, TraceCounters &tc)
{
    const float tfar0 = bestT;
    // keep 1/d finite so that 0*inf never appears in the slab test
    V3 ds(__builtin_fabsf(D.x) < 1e-20f ? __builtin_copysignf(1e-20f, D.x) : D.x,
          __builtin_fabsf(D.y) < 1e-20f ? __builtin_copysignf(1e-20f, D.y) : D.y,
          __builtin_fabsf(D.z) < 1e-20f ? __builtin_copysignf(1e-20f, D.z) : D.z);
    V3 inv(1.0f/ds.x, 1.0f/ds.y, 1.0f/ds.z);
    V3 noi(-O.x*inv.x, -O.y*inv.y, -O.z*inv.z);
    const int maxSp = sc.stackEntries;
    int sp = 0;
    uint32_t node = sc.rootChild;
    while (true) {
        if (!(node & 0x80000000u)) {
            const float4 *n = sc.nodes + size_t(node)*4;
            float4 q0 = n[0], q1 = n[1], q2 = n[2], q3 = n[3];
            if (COUNT) tc.nodes++;
            float a, b;
            a = TGPU_FMA(q0.x, inv.x, noi.x); b = TGPU_FMA(q0.w, inv.x, noi.x);
            float mn0 = fminf(a, b), mx0 = fmaxf(a, b);
            a = TGPU_FMA(q0.y, inv.y, noi.y); b = TGPU_FMA(q1.x, inv.y, noi.y);
            mn0 = fmaxf(mn0, fminf(a, b)); mx0 = fminf(mx0, fmaxf(a, b));
            a = TGPU_FMA(q0.z, inv.z, noi.z); b = TGPU_FMA(q1.y, inv.z, noi.z);
            mn0 = fmaxf(mn0, fminf(a, b)); mx0 = fminf(mx0, fmaxf(a, b));
            a = TGPU_FMA(q1.z, inv.x, noi.x); b = TGPU_FMA(q2.y, inv.x, noi.x);
            float mn1 = fminf(a, b), mx1 = fmaxf(a, b);
            a = TGPU_FMA(q1.w, inv.y, noi.y); b = TGPU_FMA(q2.z, inv.y, noi.y);
            mn1 = fmaxf(mn1, fminf(a, b)); mx1 = fminf(mx1, fmaxf(a, b));
            a = TGPU_FMA(q2.x, inv.z, noi.z); b = TGPU_FMA(q2.w, inv.z, noi.z);
            mn1 = fmaxf(mn1, fminf(a, b)); mx1 = fminf(mx1, fmaxf(a, b));
            mn0 = fmaxf(mn0, tnear); mx0 = fminf(mx0, bestT);
            mn1 = fmaxf(mn1, tnear); mx1 = fminf(mx1, bestT);
            bool h0 = mn0 <= mx0, h1 = mn1 <= mx1;
            uint32_t c0 = __float_as_uint(q3.x), c1 = __float_as_uint(q3.y);
            if (h0 && h1) {
                bool swap = mn1 < mn0;
                uint32_t nearC = swap ? c1 : c0, farC = swap ? c0 : c1;
                if (sp < maxSp) stack[(sp++)*256] = farC;
                node = nearC;
                continue;
            } else if (h0) {
                node = c0;
                continue;
            } else if (h1) {
                node = c1;
                continue;
            }
        } else {
            uint32_t payload = node & 0x7FFFFFFFu;
            uint32_t first = payload >> 2, cnt = (payload & 3u) + 1u;
            for (uint32_t j = 0; j < cnt; ++j) {
                const float4 *tp = sc.tris + size_t(first + j)*3;
                float4 t0 = tp[0], t1 = tp[1], t2 = tp[2];
                if (COUNT) tc.tris++;
                float t, u, v;
                if (triTest(t0, t1, t2, O, D, tnear, tfar0, t, u, v)) {
                    if (ANY) return true;
                    uint32_t id = sc.triId[first + j];
                    if (t < bestT || (t == bestT && id < bestPrim)) {
                        bestT = t; bestU = u; bestV = v; bestPrim = id;
                    }
                }
            }
        }
        if (sp == 0) break;
        node = stack[(--sp)*256];
    }
    return false;
}

// TraceableScene::intersect restated: analytic primitives first (each shrinking tfar, Quad.cpp:71-98,
// Cube.cpp:94-127), then the triangle BVH.  Returns encoded prim or PRIM_MISS.
template <bool COUNT>
TG_DEV uint32_t closestHit(const DScene &sc, uint32_t *stack, const V3 &O, const V3 &D, float tnear, float &tfar, float &u, float &v, TraceCounters &tc)
{
    uint32_t prim = PRIM_MISS;
    for (uint32_t i = 0; i < sc.nQuads; ++i) {
        QuadHit h;
        if (quadIntersect(sc.quads[i], O, D, tnear, tfar, h)) {
            tfar = h.t; u = h.u; v = h.v;
            prim = (KIND_QUAD << 30) | i;
        }
    }
    for (uint32_t i = 0; i < sc.nCubes; ++i) {
        float t;
        bool back;
        if (cubeIntersect(sc.cubes[i], O, D, tnear, tfar, t, back)) {
            tfar = t; u = 0.0f; v = 0.0f;
            prim = (KIND_CUBE << 30) | i;
        }
    }
    if (sc.nTris) {
        uint32_t tri = PRIM_MISS;
        float t = tfar, tu = 0.0f, tv = 0.0f;
        traverse<false, COUNT>(sc, stack, O, D, tnear, t, tu, tv, tri, tc);
        if (tri != PRIM_MISS) {
            tfar = t; u = tu; v = tv;
            prim = tri; // KIND_TRI == 0
        }
    }
    return prim;
}

// generalizedShadowRay for scenes without forward-lobe BSDFs: blocked iff any primitive other than
// the end cap is hit inside (tnear, tfar) (TraceBase.cpp:62-125)
template <bool COUNT>
TG_DEV bool occluded(const DScene &sc, uint32_t *stack, const V3 &O, const V3 &D, float tnear, float tfar, uint32_t endCap, TraceCounters &tc)
{
    for (uint32_t i = 0; i < sc.nQuads; ++i) {
        if (((KIND_QUAD << 30) | i) == endCap) continue;
        QuadHit h;
        if (quadIntersect(sc.quads[i], O, D, tnear, tfar, h)) return true;
    }
    for (uint32_t i = 0; i < sc.nCubes; ++i) {
        float t;
        bool back;
        if (cubeIntersect(sc.cubes[i], O, D, tnear, tfar, t, back)) return true;
    }
    if (sc.nTris) {
        float t = tfar, u, v;
        uint32_t p = PRIM_MISS;
        if (traverse<true, COUNT>(sc, stack, O, D, tnear, t, u, v, p, tc)) return true;
    }
    return false;
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

TG_DEV uint32_t hitClass(const DScene &sc, uint32_t prim)
{
    if (prim == PRIM_MISS) return 0xFFu;
    uint32_t kind = primKind(prim), idx = primIndex(prim);
    if (kind == KIND_TRI) return sc.triClass[idx];
    if (kind == KIND_QUAD) return uint32_t(sc.bsdfs[sc.quads[idx].bsdf].type);
    return uint32_t(sc.bsdfs[sc.cubes[idx].bsdf].type);
}

// =================================================================================================
// K2 extend: persistent waves pull 64 rays at a time from the queue; per-lane traversal stack in LDS
// (stack[level][thread], bank = lane -> conflict free); hits are appended to per-BSDF-class queues.
// RAW = true: ray-level entry point (tgpu_trace_closest), no classification.
// =================================================================================================
template <bool COUNT, bool RAW>
__global__ void __launch_bounds__(256) k_extend(DScene sc, DState st, const uint32_t *queue, uint32_t count)
{
    extern __shared__ uint32_t ldsStack[];
    uint32_t *stack = ldsStack + threadIdx.x;
    const int lane = threadIdx.x & 63;
    TraceCounters tc{0, 0};
    while (true) {
        uint32_t base = 0;
        if (lane == 0) base = atomicAdd(&st.counters[CNT_HEAD_EXT], 64u);
        base = __shfl(base, 0);
        if (base >= count) break;
        uint32_t idx = base + lane;
        bool active = idx < count;
        uint32_t path = 0, cls = 0xFEu;
        if (active) {
            path = queue ? queue[idx] : idx;
            float4 ro = st.rayO[path], rd = st.rayD[path];
            V3 O = xyz(ro), D = xyz(rd);
            float tfar = rd.w, u = 0.0f, v = 0.0f;
            uint32_t prim = closestHit<COUNT>(sc, stack, O, D, ro.w, tfar, u, v, tc);
            st.hit[path] = make_float4(tfar, u, v, __uint_as_float(prim));
            if (!RAW) cls = hitClass(sc, prim);
        }
        if (!RAW) {
            queueAppend(active && cls == 0xFFu, st.missQ, &st.counters[CNT_MISS], path);
#pragma unroll
            for (uint32_t c = 0; c < TGPU_BSDF_NUM_CLASSES; ++c)
                queueAppend(active && cls == c, st.classQ[c], &st.counters[CNT_CLASS0 + c], path);
        }
    }
    if (COUNT) flushCounters(st.stats, 0, tc);
}

// =================================================================================================
// K5 miss: TraceBase::handleInfiniteLights (TraceBase.cpp:570-578) + end of traceSample
// =================================================================================================
__global__ void __launch_bounds__(256) k_miss(DScene sc, DState st)
{
    const uint32_t count = st.counters[CNT_MISS];
    const uint32_t stride = gridDim.x*blockDim.x;
    const uint32_t rounded = (count + 63u) & ~63u;
    for (uint32_t q = blockIdx.x*blockDim.x + threadIdx.x; q < rounded; q += stride) {
        const bool active = q < count;
        uint32_t path = 0;
        if (active) {
            path = st.missQ[q];
            float4 tpw = st.tp[path], emw = st.em[path];
            V3 tp = xyz(tpw), em = xyz(emw);
            uint32_t flags = __float_as_uint(tpw.w);
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
                }
            }
            st.tp[path] = mk4u(tp, packFlags(bounce, wasSpecular, end));
        }
        queueAppend(active, st.endQ, &st.counters[CNT_END], path);
    }
}

// =================================================================================================
// K3 shade
// =================================================================================================
struct HitInfo { V3 Ng, Ns, p; float uvx, uvy; int bsdf; bool emissive; int emissionTex; bool backSide; };

TG_DEV void fetchHitInfo(const DScene &sc, uint32_t prim, const V3 &O, const V3 &D, float t, float u, float v, HitInfo &info)
{
    uint32_t kind = primKind(prim), idx = primIndex(prim);
    info.emissive = false;
    info.emissionTex = -1;
    info.backSide = false;
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
    }
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
TG_DEV float lightApproximateRadiance(const DScene &sc, const tgpu_light &l, const V3 &p)
{
    if (l.kind == TGPU_LIGHT_ENV) return -1.0f;
    const tgpu_quad &q = sc.quads[l.index];
    V3 R0 = V3(q.base) - p;
    if (R0.dot(V3(q.normal)) >= 0.0f) return 0.0f;
    V3 R1 = R0 + V3(q.edge0), R2 = R1 + V3(q.edge1), R3 = R0 + V3(q.edge1);
    V3 n0 = R0.cross(R1).normalized(), n1 = R1.cross(R2).normalized(), n2 = R2.cross(R3).normalized(), n3 = R3.cross(R0).normalized();
    float Q = tgpu_acosf(n0.dot(n1)) + tgpu_acosf(n1.dot(n2)) + tgpu_acosf(n2.dot(n3)) + tgpu_acosf(n3.dot(n0));
    const tgpu_texture &t = sc.textures[q.emission_tex];
    V3 avg = t.type == TGPU_TEX_CONSTANT ? V3(t.value) : V3(0.5f);
    return (TWO_PI - __builtin_fabsf(Q))*avg.max();
}

// TraceBase::chooseLight (TraceBase.cpp:416-459); up to TG_MAX_LIGHTS samplable lights
#define TG_MAX_LIGHTS 8
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
            lightPdf[i] = lightApproximateRadiance(sc, sc.lights[i], p);
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

template <int CLASS>
__global__ void __launch_bounds__(256) k_shade(DScene sc, DState st, uint32_t *extNext)
{
    const uint32_t count = st.counters[CNT_CLASS0 + CLASS];
    const int maxBounces = sc.settings.max_bounces, minBounces = sc.settings.min_bounces;
    const bool neeEnabled = sc.settings.enable_light_sampling != 0;
    // round the trip count up to whole waves so that ballots in queueAppend see uniform control flow
    const uint32_t stride = gridDim.x*blockDim.x;
    const uint32_t rounded = (count + 63u) & ~63u;
    for (uint32_t q = blockIdx.x*blockDim.x + threadIdx.x; q < rounded; q += stride) {
        bool active = q < count;
        bool pushShadow = false, pushExt = false, ended = false;
        uint32_t path = 0;
        if (active) {
            path = st.classQ[CLASS][q];
            float4 ro = st.rayO[path], rd = st.rayD[path], hw = st.hit[path], tpw = st.tp[path], emw = st.em[path];
            uint4 sw = st.samp[path];
            V3 O = xyz(ro), D = xyz(rd), tp = xyz(tpw), em = xyz(emw);
            uint32_t flags = __float_as_uint(tpw.w);
            int bounce = flagsBounce(flags);
            bool wasSpecular = flagsSpecular(flags);
            int end = END_ALIVE;

            if (bounce > 0 && isnanf_(tp.sum() + em.sum())) {
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
                fetchHitInfo(sc, prim, O, D, hw.x, hw.y, hw.z, info);
                const tgpu_bsdf &bsdf = sc.bsdfs[info.bsdf];
                const uint32_t lobes = bsdfLobes(sc, bsdf);

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

                // handleSurface (TraceBase.cpp:516-568): forward-lobe test draws from the supplemental
                // stream for every surface hit; none of the BSDFs in scope has a forward lobe -> pTrue = 0
                (void)smp.nextBoolean(0.0f);

                const float eps = 5e-4f; // TraceableScene.hpp:39
                V3 L(0.0f), B(0.0f);
                float tfarL = -1.0f, tfarB = -1.0f, pickWeight = 1.0f;
                V3 dL(0.0f), dB(0.0f);
                uint32_t endCap = PRIM_MISS;
                const V3 tpVertex = tp;

                if (neeEnabled && bounce < maxBounces - 1 && sc.nLights > 0) {
                    // estimateDirect -> chooseLight (TraceBase.cpp:416-459)
                    int lightIdx = chooseLight(sc, info.p, smp, pickWeight);
                    bool haveLight = lightIdx >= 0;
                    bool pureSpecular = lobes != 0 && (lobes & ~uint32_t(SpecularLobe)) == 0;
                    if (haveLight && !pureSpecular && lobes != ForwardLobe) { // sampleDirect (TraceBase.cpp:383-400)
                        const tgpu_light light = sc.lights[lightIdx];
                        const bool isEnv = light.kind == TGPU_LIGHT_ENV;
                        endCap = isEnv ? ((KIND_ENV << 30) | 0u) : ((KIND_QUAD << 30) | uint32_t(light.index));
                        const bool shadowPasses = bounce + 1 >= minBounces; // generalizedShadowRay's minBounces rule

                        // ---- lightSample (TraceBase.cpp:246-285) ----
                        {
                            LightSample ls;
                            bool ok = isEnv ? envSampleDirect(sc, smp, ls) : quadSampleDirect(sc.quads[light.index], info.p, smp, ls);
                            if (ok) {
                                ev.wo = ev.frame.toLocal(ls.d);
                                bool consistent = true;
                                if (sc.settings.enable_consistency_checks) {
                                    bool geometricBackside = ls.d.dot(info.Ng) < 0.0f;
                                    bool shadingBackside = (ev.wo.z < 0.0f) ^ ev.flippedFrame;
                                    consistent = geometricBackside == shadingBackside;
                                }
                                ev.requestedLobe = AllButSpecular;
                                V3 f = consistent ? bsdfEval(sc, bsdf, ev) : V3(0.0f);
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
                                        QuadHit qh;
                                        const tgpu_quad &lq = sc.quads[light.index];
                                        lit = quadIntersect(lq, info.p, ls.d, eps, TG_INF, qh) && !(qh.t*(1.0f + 1e-3f) < ls.dist);
                                        if (lit) {
                                            farT = qh.t;
                                            E = qh.backSide ? V3(0.0f) : texEval(sc, lq.emission_tex, qh.u, qh.v);
                                        }
                                    }
                                    if (lit && shadowPasses && !E.isZero()) {
                                        V3 lightF = f*E/ls.pdf;
                                        lightF *= powerHeuristic(ls.pdf, bsdfPdf(sc, bsdf, ev));
                                        L = lightF;
                                        dL = ls.d;
                                        tfarL = farT;
                                    }
                                }
                            }
                        }
                        // ---- bsdfSample (TraceBase.cpp:287-321) ----
                        {
                            ev.requestedLobe = AllButSpecular;
                            if (bsdfSample(sc, bsdf, ev, smp) && !ev.weight.isZero()) {
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
                                        QuadHit qh;
                                        const tgpu_quad &lq = sc.quads[light.index];
                                        lit = quadIntersect(lq, info.p, wo, eps, TG_INF, qh); // expectedDist = -1 never rejects
                                        if (lit) {
                                            farT = qh.t;
                                            E = qh.backSide ? V3(0.0f) : texEval(sc, lq.emission_tex, qh.u, qh.v);
                                            V3 normal(lq.normal);
                                            float cosTheta = __builtin_fabsf(normal.dot(wo));
                                            float tq = normal.dot(V3(lq.base) - info.p)/normal.dot(wo);
                                            lightPdf = tq*tq/(cosTheta*lq.area);
                                        }
                                    }
                                    if (lit && shadowPasses && !E.isZero()) {
                                        V3 bsdfF = E*ev.weight;
                                        bsdfF *= powerHeuristic(ev.pdf, lightPdf);
                                        B = bsdfF;
                                        dB = wo;
                                        tfarB = farT;
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
                if (info.emissive && bounce >= minBounces) {
                    if (!neeEnabled || wasSpecular) { // every emissive primitive in scope is samplable
                        V3 Ed = info.backSide ? V3(0.0f) : texEval(sc, info.emissionTex, info.uvx, info.uvy);
                        postAdd = Ed*tp;
                        hasPostAdd = true;
                    }
                }
                if (hasPostAdd && !pushShadow) em += postAdd;

                // continuation (TraceBase.cpp:546-565)
                ev.requestedLobe = AllLobes;
                bool sampled = bsdfSample(sc, bsdf, ev, smp);
                V3 wo(0.0f);
                bool alive = sampled;
                if (sampled) {
                    wo = ev.frame.toGlobal(ev.wo);
                    if (sc.settings.enable_consistency_checks) {
                        bool geometricBackside = wo.dot(info.Ng) < 0.0f;
                        bool shadingBackside = (ev.wo.z < 0.0f) ^ ev.flippedFrame;
                        alive = geometricBackside == shadingBackside;
                    }
                }
                V3 hitpoint = O + D*hw.x;
                if (!alive) {
                    end = END_RAW; // handleSurface returned false -> `return emission`
                } else {
                    tp *= ev.weight;
                    wasSpecular = (ev.sampledLobe & SpecularLobe) != 0;
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
                            if (bounce < maxBounces) pushExt = true;
                            else end = END_CHECKED;
                        }
                    }
                }
                // NEE rays start at info.p, which for every primitive kind has the bits of the scatter origin
                st.rayO[path] = mk4(hitpoint, eps);
                st.rayD[path] = mk4(wo, TG_INF);
                st.samp[path] = make_uint4(smp.dimension, uint32_t(smp.pcg), uint32_t(smp.pcg >> 32), sw.w);
                if (pushShadow) {
                    st.shA[path] = mk4(dL, tfarL);
                    st.shB[path] = mk4(dB, tfarB);
                    st.shC[path] = mk4(L, pickWeight);
                    st.shD[path] = mk4u(B, endCap);
                    st.shE[path] = mk4u(tpVertex, hasPostAdd ? 1u : 0u);
                    if (hasPostAdd) st.shF[path] = mk4(postAdd, 0.0f);
                }
                st.em[path] = mk4u(em, smp.scramble);
            }
            st.tp[path] = mk4u(tp, packFlags(bounce, wasSpecular, end));
            ended = end != END_ALIVE;
        }
        queueAppend(pushShadow, st.shadowQ, &st.counters[CNT_SHADOW], path);
        queueAppend(pushExt, extNext, &st.counters[CNT_EXT_NEXT], path);
        queueAppend(ended, st.endQ, &st.counters[CNT_END], path);
    }
    if (threadIdx.x == 0 && blockIdx.x == 0) atomicAdd(&st.stats[5], (unsigned long long)count);
}

// =================================================================================================
// K4 shadow: both NEE rays of a vertex are traced by one lane, then resolved in the reference's order
//   emission += ((0 + L*visL) + B*visB) * weight * throughput   (TraceBase.cpp:383-400,483-494,537-538)
// =================================================================================================
template <bool COUNT>
__global__ void __launch_bounds__(256) k_shadow(DScene sc, DState st)
{
    extern __shared__ uint32_t ldsStack[];
    uint32_t *stack = ldsStack + threadIdx.x;
    const int lane = threadIdx.x & 63;
    const uint32_t count = st.counters[CNT_SHADOW];
    TraceCounters tc{0, 0};
    uint32_t rays = 0;
    while (true) {
        uint32_t base = 0;
        if (lane == 0) base = atomicAdd(&st.counters[CNT_HEAD_SHADOW], 64u);
        base = __shfl(base, 0);
        if (base >= count) break;
        uint32_t idx = base + lane;
        if (idx < count) {
            uint32_t path = st.shadowQ[idx];
            float4 ro = st.rayO[path], a = st.shA[path], b = st.shB[path], c = st.shC[path], d = st.shD[path], e = st.shE[path];
            V3 O = xyz(ro);
            uint32_t endCap = __float_as_uint(d.w);
            V3 result(0.0f);
            if (a.w >= 0.0f) {
                rays++;
                if (!occluded<COUNT>(sc, stack, O, xyz(a), ro.w, a.w, endCap, tc)) result += xyz(c);
            }
            if (b.w >= 0.0f) {
                rays++;
                if (!occluded<COUNT>(sc, stack, O, xyz(b), ro.w, b.w, endCap, tc)) result += xyz(d);
            }
            float4 emw = st.em[path];
            V3 em = xyz(emw);
            em += (result*c.w)*xyz(e);
            if (__float_as_uint(e.w) & 1u) em += xyz(st.shF[path]);
            st.em[path] = make_float4(em.x, em.y, em.z, emw.w);
        }
    }
    if (COUNT) flushCounters(st.stats, 2, tc);
    for (int off = 32; off > 0; off >>= 1) rays += __shfl_down(rays, off);
    if (lane == 0 && rays) atomicAdd(&st.stats[4], (unsigned long long)rays);
}

// raw any-hit for tgpu_trace_occluded
__global__ void __launch_bounds__(256) k_occluded_raw(DScene sc, DState st, uint32_t count, uint8_t *out)
{
    extern __shared__ uint32_t ldsStack[];
    uint32_t *stack = ldsStack + threadIdx.x;
    TraceCounters tc{0, 0};
    for (uint32_t i = blockIdx.x*blockDim.x + threadIdx.x; i < count; i += gridDim.x*blockDim.x) {
        float4 ro = st.rayO[i], rd = st.rayD[i];
        out[i] = occluded<false>(sc, stack, xyz(ro), xyz(rd), ro.w, rd.w, PRIM_MISS, tc) ? 1 : 0;
    }
}

// =================================================================================================
// K7 accumulate: OutputBuffer::addSample in sample order (cameras/OutputBuffer.hpp:104-132)
// =================================================================================================
__global__ void __launch_bounds__(256) k_accumulate(DState st, uint32_t ringSlot, uint32_t sppCount)
{
    const float4 *res = st.results + size_t(ringSlot)*st.resultSlice;
    for (uint32_t k = blockIdx.x*blockDim.x + threadIdx.x; k < st.nPixels; k += gridDim.x*blockDim.x) {
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
}

// device evaluation of include/tgpu_math.h for the bit-exactness test
__global__ void k_eval_math(int kind, const float *a, const float *b, uint32_t n, float *out)
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

} // namespace tg
