/*
 * oracle.h -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.
 *
 * CPU restatement of the reference's unidirectional path-tracing hot path
 * (PathTraceIntegrator::renderTile -> PathTracer::traceSample -> TraceBase::* ->
 * primitives / BSDFs / sampler), consuming the same flattened scene description
 * (tgpu_scene_desc) the HIP core takes.  Only tests/, __graft_entry__.smoke()
 * and bench.py's cpu_baseline leg may load this library; nothing under
 * tungsten_amd/ links or calls it.
 *
 * Parity status: pinned against full-frame PFMs rendered by the unmodified,
 * compiled reference (tests/golden/README.md) -- see DESIGN.md "Oracle".
 *
 * Modes (the reference behaviour is all zeros):
 *   rng  0 = tile-sequential supplemental PCG stream, exactly as the reference
 *            (SobolPathSampler.hpp:14,57-65; PathTraceIntegrator.cpp:36-38,136-156)
 *        1 = path-local supplemental stream (what a parallel wavefront can do)
 *   math 0 = glibc libm   1 = include/tgpu_math.h (bit-identical on gfx950)
 *   rcp  0 = IEEE division for the triangle hit (t,u,v)
 *        1 = Embree's rcp(): RCPPS + one Newton step (vfloat4_sse2.h:166-173)
 */
#ifndef TUNGSTEN_ORACLE_H_
#define TUNGSTEN_ORACLE_H_

#include "../include/tungsten_amd.h"

#ifdef __cplusplus
extern "C" {
#endif

typedef struct oracle_ctx oracle_ctx;

typedef struct oracle_stats {
    uint64_t samples, extend_rays, shadow_rays, surface_events;
    uint64_t node_visits, tri_tests;
} oracle_stats;

oracle_ctx *oracle_create(const tgpu_scene_desc *scene, int rng_mode, int math_mode, int rcp_mode);
void oracle_destroy(oracle_ctx *ctx);

/* Reference order: dice the whole image into 16x16 tiles seeded from `seed`,
 * trace samples [spp_begin, spp_begin+spp_count) of every pixel, accumulate into
 * rgb_mean / sample_count (res_x*res_y, image order).  Per-tile sampler state is
 * kept inside the context between calls (first call must use spp_begin == 0). */
int oracle_render_image(oracle_ctx *ctx, uint32_t seed, uint32_t spp_begin, uint32_t spp_count,
                        float *rgb_mean, uint32_t *sample_count, int n_threads);

/* Output buffers of the render calls below (cameras/OutputBuffer.hpp; indexed by tgpu_output_type, pixel order as rgb_mean): the
 * caller owns the zero-initialised arrays; a == NULL leaves that buffer out, b / variance select two_buffer_variance /
 * sample_variance.  Binding any buffer makes traceSample track the auxiliary values (PathTracer.cpp:78-96,133-140). */
typedef struct oracle_outbuf { float *a, *b, *variance; uint32_t *count; } oracle_outbuf;
void oracle_bind_output_buffers(oracle_ctx *ctx, const oracle_outbuf *bufs /* TGPU_OUTPUT_COUNT entries, or NULL */);

void oracle_kat_transmittance(int type, const float *param3, int math_mode, const float *tau3, uint32_t n, float *forms_out, float *sigma_bar,
                              uint32_t seed, uint32_t n_samples, float *surface_out, float *medium_out);
void oracle_kat_output_buffer(int channels, uint32_t n_samples, const uint32_t *pixel, const float *values, float *a, float *b /* or NULL */,
                              float *variance /* or NULL */, uint32_t *count);

/* Shard order (rng mode 1 only): same contract as tgpu_render_pass+tgpu_download. */
int oracle_render_shard(oracle_ctx *ctx, const tgpu_shard_desc *shard, uint32_t spp_begin, uint32_t spp_count,
                        float *rgb_mean, uint32_t *sample_count, int n_threads);

/* The whole render loop of the reference incl. generateWork / adaptive sampling (PathTraceIntegrator.cpp:44-156,
 * StandaloneRenderer loop Shared.hpp:283): passes of spp_step up to total_spp.  record_out (may be NULL) receives
 * {sampleCount, mean, runningVariance} as 3 floats per 4x4 variance tile. */
int oracle_render_reference(oracle_ctx *ctx, uint32_t seed, uint32_t total_spp, uint32_t spp_step, int adaptive,
                            float *rgb_mean, uint32_t *sample_count, float *record_out, int n_threads);

/* Shard order, per-variance-tile sample counts (rng mode 1 only): same contract as
 * tgpu_render_pass_adaptive + tgpu_download + tgpu_download_records.  vt_* and records are whole-image tables
 * (varW*varH entries); records of the shard's tiles are updated in place. */
int oracle_render_shard_adaptive(oracle_ctx *ctx, const tgpu_shard_desc *shard, const uint32_t *vt_sample_index,
                                 const uint32_t *vt_sample_count, float *rgb_mean, uint32_t *sample_count,
                                 tgpu_sample_record *records, int n_threads);

/* Ray-level queries, same layout as tgpu_trace_closest / tgpu_trace_occluded. */
int oracle_trace_closest(oracle_ctx *ctx, const float *rays, uint32_t n, float *hits);
int oracle_trace_occluded(oracle_ctx *ctx, const float *rays, uint32_t n, uint8_t *occluded);

/* Leaf functions exposed for known-answer tests. */
uint32_t oracle_hash32(uint32_t x);
void oracle_fast_exp(const oracle_ctx *ctx, const float *x3, float *out3); /* FastMath::exp(Vec3f) */
void oracle_exp_table(const oracle_ctx *ctx, uint32_t *out1024);
uint32_t oracle_sobol(const oracle_ctx *ctx, uint32_t index, uint32_t dim, uint32_t scramble);
void oracle_pcg_stream(uint64_t state, uint64_t seq, uint32_t n, uint32_t *out);
void oracle_camera_ray(const oracle_ctx *ctx, uint32_t px, uint32_t py, float xi0, float xi1, float *origin_dir6);
void oracle_tile_seeds(uint32_t res_x, uint32_t res_y, uint32_t seed, uint32_t *out);
uint32_t oracle_kat_binary_bvh(const float *boxes6, uint32_t n, const float *origin3, const float *dir3, uint32_t *out, uint32_t max_visits);

/* The Embree restatement of the reference mode (oracle/embree_bvh4.h) on its own, for tests/test_ref_kat5.py:
 * user form  -- n_boxes boxes as ONE user geometry; per ray {org3, dir3, tnear, tfar, scripted[n_boxes]}: a visited primitive reports a hit at
 *               its scripted distance unless that is negative or outside [tnear, tfar].  visits (capacity max_visits per ray) receives the
 *               primitives in the order the traversal asks them, n_visits their number, prim / tfar the result (prim 0xFFFFFFFF = none).
 * tri form   -- a triangle mesh (TriangleMesh.cpp:531-566); per ray {org3, dir3, tnear, tfar} -> prim, {t, u, v}.                              */
void oracle_kat_embree_user(const float *boxes6, uint32_t n_boxes, const float *rays, uint32_t n_rays, uint32_t *visits, uint32_t max_visits,
                            uint32_t *n_visits, uint32_t *prim, float *tfar);
void oracle_kat_embree_tris(const float *verts3, uint32_t n_verts, const uint32_t *tris3, uint32_t n_tris, const float *rays8, uint32_t n_rays,
                            uint32_t *prim, float *tuv3);

/* Triangle queries test EVERY triangle instead of walking the BVH: the definition (include/tgpu_bounds.h) both BVH walks
 * are checked against.  O(rays x triangles): ray batches and single paths only. */
void oracle_set_brute_force(oracle_ctx *ctx, int on);

void oracle_get_stats(oracle_ctx *ctx, oracle_stats *out);
void oracle_reset_stats(oracle_ctx *ctx);

#ifdef __cplusplus
}
#endif
#endif
