/*
 * tungsten_amd.h -- C ABI of the MI355X wavefront path-tracing core.
 *
 * This is the drop-in boundary for Tungsten's unidirectional `path_tracer`
 * integrator.  The reference interface it sits behind is the abstract class
 * Tungsten::Integrator (src/core/integrators/Integrator.hpp:16-63); each entry
 * point below names the reference member it replaces.  The host-side C++ class
 * `tungsten_amd::PathTraceIntegrator` (tungsten_amd/host/integrator.hpp) wraps
 * these calls back into the reference's Integrator shape; INTEGRATION.md shows
 * the binding a Tungsten maintainer would add.
 *
 * Conventions: plain pointers and sizes only, no C++/torch types; return 0 on
 * success, negative tgpu_status on error (never throws across the ABI);
 * tgpu_last_error() returns a human readable message for the last failure on a
 * context.  All scene buffers are caller-owned and copied during
 * tgpu_upload_scene().  One host thread drives one context; contexts are
 * independent.  All arithmetic is IEEE binary32 unless marked.
 */
#ifndef TUNGSTEN_AMD_H_
#define TUNGSTEN_AMD_H_

#include <stdint.h>
#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

#define TGPU_ABI_VERSION 4

typedef enum tgpu_status {
    TGPU_OK = 0,
    TGPU_ERR_INVALID = -1,     /* bad argument / inconsistent scene description   */
    TGPU_ERR_NO_DEVICE = -2,   /* no HIP device / HIP runtime failure at init      */
    TGPU_ERR_HIP = -3,         /* a HIP call failed (message in tgpu_last_error)   */
    TGPU_ERR_UNSUPPORTED = -4, /* scene uses a feature outside the hot-path scope  */
    TGPU_ERR_STATE = -5,       /* call order violated (e.g. render before upload)  */
    TGPU_ERR_ABORTED = -6      /* pass was aborted via tgpu_abort                  */
} tgpu_status;

/* ---- textures (src/core/textures: ConstantTexture, CheckerTexture.cpp:64-69,
 *      BitmapTexture.cpp:298-352) ------------------------------------------- */
enum { TGPU_TEX_CONSTANT = 0, TGPU_TEX_CHECKER = 1, TGPU_TEX_BITMAP = 2 };

typedef struct tgpu_texture {
    int32_t type;
    float   value[3];      /* constant                                     */
    float   on_color[3];   /* checker                                      */
    float   off_color[3];
    int32_t res_u, res_v;
    int32_t bitmap;        /* index into tgpu_scene_desc::bitmaps          */
} tgpu_texture;

typedef struct tgpu_bitmap {
    int32_t      w, h;
    int32_t      clamp;    /* 0 = wrap (reference default for bare string) */
    int32_t      linear;   /* bilinear interpolation                       */
    float        scale;
    const float *texels;   /* w*h*3 RGB f32, row 0 = top (ImageIO.cpp:340) */
    float        average[3]; /* BitmapTexture::_avg: sum of texel/(w*h) in row-major order (BitmapTexture.cpp:183-193) */
} tgpu_bitmap;

/* ---- BSDFs (src/core/bsdfs) -------------------------------------------- */
enum {
    TGPU_BSDF_LAMBERT = 0,          /* LambertBsdf.cpp:27-68            */
    TGPU_BSDF_ROUGH_CONDUCTOR = 1,  /* RoughConductorBsdf.cpp:60-143    */
    TGPU_BSDF_SMOOTH_COAT = 2,      /* SmoothCoatBsdf.cpp:41-223        */
    TGPU_BSDF_NULL = 3,             /* NullBsdf.cpp (light carriers)    */
    TGPU_BSDF_ROUGH_DIELECTRIC = 4, /* RoughDielectricBsdf.cpp:55-280   */
    TGPU_BSDF_FORWARD = 5,          /* ForwardBsdf.cpp:9-36 (index-matched medium boundaries) */
    TGPU_BSDF_NUM_CLASSES = 6,      /* types 0..5 get a shade kernel of their own; the ones below share one */
    TGPU_BSDF_MIRROR = 6,           /* MirrorBsdf.cpp:16-60              */
    TGPU_BSDF_DIELECTRIC = 7,       /* DielectricBsdf.cpp:50-190 (ior, enable_refraction) */
    TGPU_BSDF_CONDUCTOR = 8,        /* ConductorBsdf.cpp:53-90 (eta, k)  */
    TGPU_BSDF_PLASTIC = 9,          /* PlasticBsdf.cpp:45-195 (ior, thickness, sigma_a) */
    TGPU_BSDF_ROUGH_PLASTIC = 10,   /* RoughPlasticBsdf.cpp:55-260 (+ distribution, roughness) */
    TGPU_BSDF_OREN_NAYAR = 11,      /* OrenNayarBsdf.cpp:41-98 (roughness)   */
    TGPU_BSDF_PHONG = 12,           /* PhongBsdf.cpp:39-131 (exponent, diffuse_ratio) */
    TGPU_BSDF_MIXED = 13,           /* MixedBsdf.cpp:72-136: substrate = bsdf0, aux_bsdf = bsdf1, aux_tex = ratio */
    TGPU_BSDF_ROUGH_COAT = 14,      /* RoughCoatBsdf.cpp:47-307: substrate, ior, thickness, sigma_a, distribution, roughness */
    TGPU_BSDF_TRANSPARENCY = 15,    /* TransparencyBsdf.cpp:42-69: substrate = base, aux_tex = alpha (opacity) */
    TGPU_BSDF_THINSHEET = 16,       /* ThinSheetBsdf.cpp:50-120: ior, sigma_a, aux_tex = thickness; enable_interference is out of scope */
    TGPU_BSDF_NUM_TYPES = 17
};
enum { TGPU_DIST_BECKMANN = 0, TGPU_DIST_PHONG = 1, TGPU_DIST_GGX = 2 };

typedef struct tgpu_bsdf {
    int32_t type;
    int32_t albedo_tex;    /* index into textures                           */
    int32_t roughness_tex; /* rough_conductor: scalar (x channel)           */
    int32_t distribution;  /* TGPU_DIST_*                                   */
    float   eta[3], k[3];  /* conductor complex IOR                         */
    float   ior;           /* smooth_coat, rough_dielectric                 */
    float   thickness;
    float   sigma_a[3];
    int32_t substrate;     /* smooth_coat: index of substrate bsdf          */
    float   scaled_sigma_a[3];   /* thickness*sigma_a   (SmoothCoatBsdf.cpp:217-223, */
    float   avg_transmittance;   /* exp(-2*avg(scaled)) baked at prepareForRender)   */
    int32_t enable_refraction;   /* rough_dielectric "enable_refraction" (default 1)  */
    float   diffuse_fresnel;     /* plastics: Fresnel::computeDiffuseFresnel(ior, 1000000) (Fresnel.hpp:141-153) */
    float   substrate_weight;    /* rough_plastic: albedo->average().avg() (RoughPlasticBsdf.cpp:252-258) */
    float   exponent;            /* phong "exponent" (default 64)                                        */
    float   diffuse_ratio;       /* phong "diffuse_ratio" (default 0.2)                                  */
    int32_t aux_bsdf;            /* mixed: bsdf1                                                         */
    int32_t aux_tex;             /* mixed: ratio; transparency: alpha; thinsheet: thickness               */
} tgpu_bsdf;

/* ---- participating media (src/core/media) ------------------------------- */
/* HomogeneousMedium after prepareForRender (HomogeneousMedium.cpp:42-48) with
 * an isotropic or Henyey-Greenstein phase function and the
 * exponential transmittance (ExponentialTransmittance.cpp:26-55, evaluated
 * with fmath::exp_ps, thirdparty/fmath/fmath.hpp:320-363).                    */
/* Transmittance models of the fork (src/core/transmittances): what a free path looks like between two
 * collisions, with separate surface->surface, surface->medium, medium->surface and medium->medium forms
 * (Transmittance.hpp:22-60).  param[]: linear / quadratic: max_t; double_exponential: sigma_a, sigma_b;
 * pulse: min, max, num_pulses.                                               */
enum { TGPU_TRANS_EXPONENTIAL = 0,        /* ExponentialTransmittance.cpp:26-55       */
       TGPU_TRANS_LINEAR = 1,             /* LinearTransmittance.cpp:30-76            */
       TGPU_TRANS_QUADRATIC = 2,          /* QuadraticTransmittance.cpp:30-65         */
       TGPU_TRANS_DOUBLE_EXPONENTIAL = 3, /* DoubleExponentialTransmittance.cpp:34-65 */
       TGPU_TRANS_PULSE = 4,              /* PulseTransmittance.cpp:43-105            */
       TGPU_TRANS_ERLANG = 5,             /* ErlangTransmittance.cpp:32-67: param[0] = rate                     */
       TGPU_TRANS_DAVIS = 6,              /* DavisTransmittance.cpp:31-60: param[0] = alpha (clamped to > 1)     */
       TGPU_TRANS_DAVIS_WEINSTEIN = 7,    /* DavisWeinsteinTransmittance.cpp:34-117: param[0] = h in [0.5, 1], param[1] = c */
       TGPU_TRANS_INTERPOLATED = 8 };     /* InterpolatedTransmittance.cpp:43-83: tr_a / tr_b blended with tr_ratio */

typedef struct tgpu_medium {
    float   sigma_a[3], sigma_s[3], sigma_t[3];
    int32_t absorption_only;     /* sigma_s == 0                               */
    int32_t max_bounce;          /* Medium::_maxBounce (default 1024)          */
    int32_t transmittance;       /* TGPU_TRANS_*                               */
    float   param[3];
    int32_t phase;               /* 0 isotropic (IsotropicPhaseFunction.cpp:17-44), 1 Henyey-Greenstein (HenyeyGreensteinPhaseFunction.cpp:19-90) */
    float   phase_g;
    /* TGPU_TRANS_INTERPOLATED only: the two models it blends (any type but INTERPOLATED) and the ratio        */
    int32_t tr_a_type;  float tr_a_param[3];
    int32_t tr_b_type;  float tr_b_param[3];
    float   tr_ratio;
} tgpu_medium;

/* ---- analytic primitives ------------------------------------------------ */
/* Quad after prepareForRender (src/core/primitives/Quad.cpp:298-316).        */
typedef struct tgpu_quad {
    float   base[3], edge0[3], edge1[3];
    float   normal[3], tangent[3], bitangent[3]; /* TangentFrame(n, e0^, e1^) */
    float   inv_uv_sq[2];
    float   area;
    int32_t bsdf;
    int32_t emission_tex;  /* -1 = not emissive                             */
    int32_t int_medium, ext_medium; /* Primitive::_intMedium/_extMedium, -1 = none (Primitive.hpp:172-183) */
} tgpu_quad;

/* Cube after prepareForRender (src/core/primitives/Cube.cpp:353-370).        */
typedef struct tgpu_cube {
    float   rot[9];        /* row-major 3x3 rotation                        */
    float   inv_rot[9];
    float   pos[3];
    float   scale[3];
    int32_t bsdf;
    int32_t emission_tex;
    int32_t int_medium, ext_medium;
} tgpu_cube;

/* Sphere after prepareForRender (src/core/primitives/Sphere.cpp:285-295).      */
typedef struct tgpu_sphere {
    float   pos[3];
    float   radius;
    float   rot[9], inv_rot[9];
    int32_t bsdf;
    int32_t emission_tex;
    int32_t int_medium, ext_medium;
} tgpu_sphere;

/* Disk after prepareForRender (src/core/primitives/Disk.cpp:315-327).          */
typedef struct tgpu_disk {
    float   center[3];
    float   r;
    float   n[3], tangent[3], bitangent[3]; /* _n, TangentFrame(_n)           */
    float   cos_apex;      /* cos(cone_angle)                                */
    float   cone_base[3];  /* _center - _n/sin(cone_angle)                   */
    int32_t bsdf;
    int32_t emission_tex;
    int32_t int_medium, ext_medium;
} tgpu_disk;

/* Infinite sphere / environment light (InfiniteSphere.cpp:27-50,161-244) and
 * the Distribution2D of its emission bitmap (Distribution2D.hpp:18-83,
 * BitmapTexture.cpp:400-455).  Tables are NULL for a constant emission.      */
typedef struct tgpu_env {
    int32_t      present;
    int32_t      samplable;      /* "sample": true                          */
    int32_t      emission_tex;   /* constant or bitmap                      */
    float        rot[9];         /* _rotTransform, row-major                */
    float        inv_rot[9];     /* its transpose                           */
    int32_t      dist_w, dist_h;
    const float *pdf;            /* dist_w*dist_h                           */
    const float *cdf;            /* dist_h*(dist_w+1)                       */
    const float *marginal_pdf;   /* dist_h                                  */
    const float *marginal_cdf;   /* dist_h+1                                */
} tgpu_env;

/* ---- camera (PinholeCamera.cpp:28-35,70-86; ReconstructionFilter) ------- */
enum { TGPU_FILTER_DIRAC = 0, TGPU_FILTER_BOX = 1, TGPU_FILTER_TABULATED = 2 };

typedef struct tgpu_camera {
    float    pos[3];
    float    rot[9];          /* upper 3x3 of _transform (right flipped)     */
    uint32_t res_x, res_y;
    float    plane_dist;
    float    ratio;           /* res_y / float(res_x)                        */
    float    pixel_size_x;    /* 1 / res_x (used for both axes)              */
    int32_t  filter_type;
    float    filter_bin_size;
    float    filter_cdf[32];  /* RFILTER_RESOLUTION + 1                      */
    int32_t  medium;          /* Camera::_medium, -1 = none (Camera.hpp)     */
} tgpu_camera;

/* ---- integrator settings (TraceSettings.hpp:23-29, PathTracerSettings) --- */
typedef struct tgpu_settings {
    int32_t min_bounces, max_bounces;
    int32_t enable_light_sampling;
    int32_t enable_two_sided_shading;
    int32_t enable_consistency_checks;
    int32_t use_sobol;        /* renderer "stratified_sampler"               */
    int32_t adaptive_sampling;/* renderer "adaptive_sampling": keep per-4x4-tile SampleRecords (SampleRecord.hpp) */
    int32_t enable_volume_light_sampling; /* PathTracerSettings.hpp:14-27 (default 1) */
} tgpu_settings;

/* renderer "output_buffers" (cameras/OutputBufferSettings.hpp:13-19, Camera.cpp:148-166): the colour buffer and the four
 * auxiliary buffers the path tracer feeds when they are requested (PathTracer.cpp:78-96,133-140).  Every buffer is an
 * OutputBuffer<float | Vec3f> (cameras/OutputBuffer.hpp): a running mean per pixel, optionally split over two half
 * buffers (two_buffer_variance) and with Welford variance sums (sample_variance), plus a sample count per pixel. */
typedef enum tgpu_output_type {
    TGPU_OUTPUT_COLOR = 0,      /* 3 channels; always present                                      */
    TGPU_OUTPUT_DEPTH = 1,      /* 1 channel : distance along the path to the first non-specular hit */
    TGPU_OUTPUT_NORMAL = 2,     /* 3 channels: shading normal there (-ray.dir() for escaped paths)  */
    TGPU_OUTPUT_ALBEDO = 3,     /* 3 channels: BSDF albedo (+ emission) there                       */
    TGPU_OUTPUT_VISIBILITY = 4, /* 1 channel : mean transmittance of that vertex' light sample       */
    TGPU_OUTPUT_COUNT = 5
} tgpu_output_type;
typedef enum tgpu_output_part {
    TGPU_PART_A = 0,        /* _bufferA: the mean (of the even samples with two_buffer_variance)   */
    TGPU_PART_B = 1,        /* _bufferB: mean of the odd samples (two_buffer_variance only)          */
    TGPU_PART_VARIANCE = 2, /* _variance: running sum of squared deviations (sample_variance only) */
    TGPU_PART_COUNT = 3     /* _sampleCount: uint32 per pixel                                      */
} tgpu_output_part;
typedef struct tgpu_output_buffer {
    int32_t enabled;              /* ignored for TGPU_OUTPUT_COLOR */
    int32_t two_buffer_variance;
    int32_t sample_variance;
} tgpu_output_buffer;

/* A light is an emissive, samplable primitive, in scene order
 * (TraceableScene.hpp:86-102).                                              */
enum { TGPU_LIGHT_ENV = 0, TGPU_LIGHT_QUAD = 1, TGPU_LIGHT_SPHERE = 2, TGPU_LIGHT_DISK = 3 };
typedef struct tgpu_light { int32_t kind; int32_t index; } tgpu_light;

/* ---- "instances" primitives (src/core/primitives/Instance.cpp) -------------
 * Rigid copies (translation + unit quaternion) of master primitives.  A master
 * is a triangle mesh, a quad or a cube; its geometry sits at the TAIL of the
 * scene arrays (after the n_tris / n_quads / n_cubes scene finites), baked by
 * the master's own transform exactly like a scene primitive.  pos/rot are the
 * values Instance::prepareForRender (:392-403) leaves in _instancePos /
 * _instanceRot; bounds_* is Primitive::bounds() of the prepared master.  The
 * core builds the BVH2 over instance boxes itself (bvh/BvhBuilder.cpp:99-185,
 * BinaryBvh.hpp:120-146): its shape is part of the result, because
 * Instance::intersect (:290-311) lets the LAST hit instance in traversal
 * order win (local rays start at the node's entry distance with farT = inf).  */
enum { TGPU_MASTER_MESH = 0, TGPU_MASTER_QUAD = 1, TGPU_MASTER_CUBE = 2 };

typedef struct tgpu_master {
    int32_t  kind;
    uint32_t first, count;   /* mesh: triangle range [first, first+count) of tri_*; quad/cube: index `first` */
    float    bounds_min[3], bounds_max[3];
} tgpu_master;

typedef struct tgpu_instances {
    uint32_t           n_masters;    /* <= 256 (uint8 ids, Instance.hpp:23)       */
    const tgpu_master *masters;
    uint32_t           n_instances;
    const float       *pos;          /* n_instances*3                             */
    const float       *rot;          /* n_instances*4 : w x y z (Quaternion.hpp)  */
    const uint8_t     *master_id;    /* n_instances                               */
} tgpu_instances;

typedef struct tgpu_scene_desc {
    uint32_t abi_version;     /* TGPU_ABI_VERSION */

    tgpu_camera   camera;
    tgpu_settings settings;
    tgpu_output_buffer output_buffers[TGPU_OUTPUT_COUNT]; /* indexed by tgpu_output_type */

    /* triangle soup: every mesh baked to world space (TriangleMesh.cpp:542-552) */
    uint32_t       n_tris;
    const float   *tri_pos;     /* n_tris*9 : p0 p1 p2                        */
    const float   *tri_normal;  /* n_tris*9 : n0 n1 n2 (normal-matrix, unnormalised) */
    const float   *tri_uv;      /* n_tris*6                                   */
    const int32_t *tri_bsdf;    /* n_tris   : index into bsdfs                */
    const uint8_t *tri_smooth;  /* n_tris   : mesh "smooth" flag              */
    const int16_t *tri_media;   /* n_tris*2 : int_medium, ext_medium of the mesh, -1 = none; NULL = no media */
    /* the meshes the soup was baked from, as Primitives: mesh m owns triangles
     * [mesh_first_tri[m], mesh_first_tri[m+1]); mesh_bounds = TriangleMesh::bounds()
     * over EVERY vertex, referenced or not (TriangleMesh.cpp:233-239,519-522).
     * Descriptive: the traversal kernels work on the soup; the oracle's reference
     * mode and a per-mesh BVH build need the grouping.                          */
    uint32_t        n_meshes;
    const uint32_t *mesh_first_tri; /* n_meshes + 1                            */
    const float    *mesh_bounds;    /* n_meshes*6 : min xyz, max xyz           */

    uint32_t         n_quads;   const tgpu_quad    *quads;
    uint32_t         n_cubes;   const tgpu_cube    *cubes;
    uint32_t         n_spheres; const tgpu_sphere  *spheres;
    uint32_t         n_disks;   const tgpu_disk    *disks;
    uint32_t         n_bsdfs;   const tgpu_bsdf    *bsdfs;
    uint32_t         n_textures;const tgpu_texture *textures;
    uint32_t         n_bitmaps; const tgpu_bitmap  *bitmaps;
    uint32_t         n_lights;  const tgpu_light   *lights;
    uint32_t         n_media;   const tgpu_medium  *media;
    tgpu_env         env;

    /* instance masters: tri_* / quads / cubes hold n_tris + n_master_tris (etc.) entries; the tail
     * entries are reachable only through instance_sets                         */
    uint32_t n_master_tris, n_master_quads, n_master_cubes;
    uint32_t n_instance_sets;  const tgpu_instances *instance_sets;

    /* Joe-Kuo Sobol' generator matrices, 1024 dimensions x 32 columns
     * (thirdparty/sobol/sobol.h:28-53; columns >= 32 are unreachable with a
     * 32-bit sample index).                                                  */
    const uint32_t *sobol_matrices;
} tgpu_scene_desc;

/* The image-plane shard this context renders: a list of pixels in the order
 * the host wants them processed plus the seed of the 16x16 tile each pixel
 * belongs to (PathTraceIntegrator.cpp:27-42 dices tiles and draws the seeds;
 * SobolPathSampler.hpp:47-52 derives scramble = seed ^ hash32(pixelId)).     */
typedef struct tgpu_shard_desc {
    uint32_t        n_pixels;
    const uint32_t *pixel_index;  /* x + y*res_x                             */
    const uint32_t *tile_seed;    /* per listed pixel                        */
} tgpu_shard_desc;

typedef struct tgpu_stats {
    uint64_t samples;          /* camera paths traced since upload            */
    uint64_t extend_rays;      /* closest-hit rays                            */
    uint64_t shadow_rays;      /* any-hit rays                                */
    uint64_t extend_node_visits, extend_tri_tests;   /* 0 unless counting on  */
    uint64_t shadow_node_visits, shadow_tri_tests;
    uint64_t surface_events;
    double   extend_ms, shadow_ms, shade_ms, other_ms; /* HIP-event kernel time, when timing on */
    uint64_t extend_launches, shadow_launches;
    uint64_t tail_launches;    /* fused tail-kernel launches                  */
    uint64_t shadow_vertices;  /* path vertices whose NEE rays went through the shadow kernel's queue form */
    /* of the last tgpu_upload_scene (not reset by tgpu_reset_stats): wall time of the whole call, of the BVH build inside it
     * (host builder, or device builder without its input copy), and which builder made the scene's tree                  */
    double   retire_ms;        /* k_retire_regenerate alone (it is also part of other_ms), when timing on */
    uint64_t shade_launches, retire_launches; /* k_shade<class> / k_volume launches, k_retire_regenerate launches */
    double   upload_ms, bvh_build_ms;
    uint64_t bvh_on_device;    /* 1 = tungsten_amd/csrc/bvh_gpu.h built the scene's BVH (TGPU_GPU_BVH / TGPU_GPU_BVH_MIN) */
} tgpu_stats;

typedef struct tgpu_ctx tgpu_ctx;

/* Create a context on HIP device `device`.  Replaces nothing in the reference
 * (EmbreeUtil::initDevice, Shared.hpp:165 is the nearest analogue).          */
int tgpu_init(int device, tgpu_ctx **out);
void tgpu_destroy(tgpu_ctx *ctx);
const char *tgpu_last_error(const tgpu_ctx *ctx);

/* TraceableScene ctor + Integrator::prepareForRender
 * (TraceableScene.hpp:57-137, PathTraceIntegrator.cpp:184-201): copies the
 * flattened scene to HBM, builds the BVH, allocates the wavefront queues for
 * `max_paths_in_flight` paths (0 = the default pool: 16 Mi slots = 5.3 GiB
 * of HBM, fewer only for shards under 64 Ki pixels).                         */
int tgpu_upload_scene(tgpu_ctx *ctx, const tgpu_scene_desc *scene,
                      const tgpu_shard_desc *shard, uint64_t max_paths_in_flight);

/* Integrator::startRender (PathTraceIntegrator.cpp:220-239): enqueue one pass
 * tracing samples [spp_begin, spp_begin+spp_count) of every pixel of the shard
 * and folding them into the running per-pixel mean in the order
 * OutputBuffer::addSample would (OutputBuffer.hpp:104-132).  Asynchronous with
 * respect to the host up to internal queue-drain checks; `stream` is a
 * hipStream_t (NULL = the context's own stream).  Any number of passes may be
 * enqueued before tgpu_wait: the path pool stays full across them (one end-of-
 * render drain for all).  Up to 32 passes are in flight at a time (a pass holds
 * a slice of the sample ring until its slowest path has ended); with more, the
 * ones in flight are finished first.                                           */
int tgpu_render_pass(tgpu_ctx *ctx, uint32_t spp_begin, uint32_t spp_count, void *stream);

/* Adaptive pass (PathTraceIntegrator::generateWork/renderTile with adaptive_sampling, PathTraceIntegrator.cpp:110-156):
 * every pixel of variance tile v traces vt_sample_count[v] samples with sample indices vt_sample_index[v] + i.
 * Variance tiles are the 4x4 pixel blocks of the image, row-major, ceil(res_x/4) x ceil(res_y/4) of them; both
 * arrays cover the whole image, tiles outside the context's shard are ignored.  Needs settings.adaptive_sampling
 * at upload.  The host computes the two arrays from tgpu_download_records (errorPercentile95, dilate,
 * distribute: :44-108).                                                        */
int tgpu_render_pass_adaptive(tgpu_ctx *ctx, const uint32_t *vt_sample_index, const uint32_t *vt_sample_count, void *stream);

/* SampleRecord statistics (path_tracer/SampleRecord.hpp:44-66) of the shard's variance tiles, written into whole-image
 * arrays (ceil(res_x/4)*ceil(res_y/4) entries; entries of tiles outside the shard are left untouched).       */
typedef struct tgpu_sample_record { uint32_t sample_count; float mean; float running_variance; } tgpu_sample_record;
int tgpu_download_records(tgpu_ctx *ctx, tgpu_sample_record *records);
int tgpu_upload_records(tgpu_ctx *ctx, const tgpu_sample_record *records);

/* Integrator::waitForCompletion (PathTraceIntegrator.cpp:241-247).           */
int tgpu_wait(tgpu_ctx *ctx);
/* Integrator::abortRender (PathTraceIntegrator.cpp:249-256); callable from
 * another host thread while tgpu_render_pass runs.                           */
int tgpu_abort(tgpu_ctx *ctx);

/* Camera::colorBuffer() contents for the shard (OutputBuffer.hpp:134-144):
 * rgb_mean is n_pixels*3 floats, sample_count n_pixels, both in shard order.
 * Either pointer may be NULL.  Host destination.                             */
int tgpu_download(tgpu_ctx *ctx, float *rgb_mean, uint32_t *sample_count);
/* Same data written to DEVICE memory the caller owns (e.g. a torch tensor that
 * is then gathered over RCCL): n_pixels*3 floats.                            */
int tgpu_resolve_to_device(tgpu_ctx *ctx, void *device_rgb, void *stream);
/* Integrator::resumeRender support: restore running means + counts.          */
int tgpu_upload_framebuffer(tgpu_ctx *ctx, const float *rgb_mean, const uint32_t *sample_count);

/* Any part of any requested output buffer (Camera::depthBuffer() ... / OutputBuffer::serialize, OutputBuffer.hpp:206-226), shard
 * order, host memory: n_pixels*channels floats for parts A, B and VARIANCE, n_pixels uint32 for part COUNT.  Asking for a buffer
 * or part that was not requested at upload is TGPU_ERR_INVALID.  TGPU_OUTPUT_COLOR / A and COUNT are what tgpu_download returns. */
int tgpu_download_output_buffer(tgpu_ctx *ctx, int type, int part, void *dst);
int tgpu_upload_output_buffer(tgpu_ctx *ctx, int type, int part, const void *src);

/* Ray-level entry points used by the parity tests (TraceableScene::intersect /
 * generalizedShadowRay on dumped ray batches).  rays: n*8 floats
 * (ox,oy,oz,tnear,dx,dy,dz,tfar) in host memory; hits: n*4 floats
 * (t,u,v,bitcast prim id; prim id 0xFFFFFFFF = miss); occluded: n bytes.
 * prim id = kind << 29 | index, kind: 0 triangle (scene soup index), 1 quad, 2 cube,
 * 3 instance (index = instance number; u,v of the master hit), 4 sphere, 5 disk (u = r^2). */
int tgpu_trace_closest(tgpu_ctx *ctx, const float *rays, uint32_t n, float *hits);
int tgpu_trace_occluded(tgpu_ctx *ctx, const float *rays, uint32_t n, uint8_t *occluded);

/* Counters (rays, visits, kernel ms).  flags bit0 = count BVH visits (slower
 * kernels), bit1 = time kernels with HIP events.                             */
int tgpu_set_profiling(tgpu_ctx *ctx, uint32_t flags);
int tgpu_get_stats(tgpu_ctx *ctx, tgpu_stats *out);
int tgpu_reset_stats(tgpu_ctx *ctx);

/* Test hook: evaluates include/tgpu_math.h on the device so that the "bit-identical to the
 * host" claim can be checked.  kind: 0 sin, 1 cos, 2 exp, 3 log, 4 atan2(a,b), 5 acos,
 * 6 a/b, 7 sqrt(a).  a, b, out: n floats in host memory.                      */
int tgpu_eval_math(tgpu_ctx *ctx, int kind, const float *a, const float *b, uint32_t n, float *out);

#ifdef __cplusplus
}
#endif
#endif /* TUNGSTEN_AMD_H_ */
