/*
 * tgpu_bounds.h -- the conservativeness contract of every triangle BVH on this path (HIP core and test oracle).
 *
 * The nearest hit is DEFINED as the lexicographic minimum of (t, triangle id) over all triangles whose
 * Moeller-Trumbore test (Embree's formulation, thirdparty/embree/kernels/geometry/triangle_intersector_moeller.h:76-113)
 * passes; a BVH only prunes.  The test is evaluated in binary32: a ray can pass it while missing the triangle's exact
 * box by a few ulps of its coordinates, and the computed t can lie a few ulps outside the box's [entry, exit].  Round 1
 * pruned with exact boxes: 3 samples of 45 M at full size (rays through a shared triangle edge) found different triangles
 * on the two sides.  A BVH that honours the definition therefore
 *   (1) grows every TRIANGLE box by tgpu_bvh_pad(scene) in all directions before building, and
 *   (2) compares box distances against the ray interval / the best hit with a relative slack of TGPU_BVH_T_SLACK.
 * The error of the edge functions is <~ 5 eps L (L = largest distance between a ray origin and a triangle), that of an
 * fma-form slab test a few ulps of the largest coordinate; the pad is 2^-16 M = 512 eps M with M >= both.
 * oracle/: ORACLE_BRUTE=1 tests every triangle and is what both walks are checked against (tests/test_gpu_parity.py,
 * rays aimed at triangle edges and vertices).
 */
#ifndef TGPU_BOUNDS_H_
#define TGPU_BOUNDS_H_

#include "tungsten_amd.h"
#include <math.h>

#define TGPU_BVH_T_SLACK 0.0009765625f /* 2^-10 */
/* (3) Round 3: a triangle's binary32 test can pass with a distance that is far off when the ray grazes the triangle (t = T/den with den -> 0:
 * a ray 1e-5 rad off the plane gives |dt|/t ~ 6e-3, more than any fixed slack; found by tests/test_gpu_parity.py::test_device_side_bvh_build
 * on a ray whose reported hit lay 9e-4 in FRONT of its own padded box).  A walk prunes by the boxes the ray really enters, so such a hit is found or
 * lost depending on the tree -- the reference's too (Embree prunes with exact boxes).  The contract therefore counts a hit only when the point
 * O + t*D it reports lies inside the triangle's own box grown by TGPU_HIT_PAD_FRACTION * tgpu_bvh_pad: then every box that contains the padded
 * triangle box is entered before t and left after it, whatever the tree.  Evaluated for candidates that passed everything else (device:
 * kernels.h triTest, oracle: triTest in GPU mode and in brute-force mode); the oracle's reference mode keeps Embree's behaviour.               */
#define TGPU_HIT_PAD_FRACTION 0.5f
/* Candidates whose ray is at least this far off the triangle's plane (|cos| between ray and normal) skip the box test: with e = 2^-24, the reported
 * point is off by at most |D| dt <= 3e |Ng||D| (|C| + t|D|)/|den| <= 6 e M/|cos| (M = tgpu_scene_extent, the bound of every distance involved), and
 * 6 e M/|cos| <= hitPad = 2^-17 M  <=>  |cos| >= 6 * 2^-24 * 2^17 = 0.047.  0.0625 leaves a margin; below it (0.4 % of cosine-weighted hits) the
 * box test decides.  Both sides evaluate  den^2 >= cos^2 * (|Ng|^2 |D|^2)  in binary32 with |Ng|^2 = fl(Ng.x^2 + (Ng.y^2 + Ng.z^2)).           */
#define TGPU_HIT_SURE_COS 0.0625f

/* M: the diagonal of the box around everything a ray can start on or hit (scene geometry in world space, camera) or the
 * largest absolute coordinate of any of it or of an instance master's own frame, whichever is larger. */
static inline float tgpu_scene_extent(const tgpu_scene_desc *s)
{
    float lo[3] = {INFINITY, INFINITY, INFINITY}, hi[3] = {-INFINITY, -INFINITY, -INFINITY}, maxAbs = 0.0f;
#define TGPU_GROW_(px, py, pz)                                                        \
    do {                                                                              \
        const float p_[3] = {(px), (py), (pz)};                                       \
        for (int k_ = 0; k_ < 3; ++k_) {                                              \
            if (p_[k_] < lo[k_]) lo[k_] = p_[k_];                                     \
            if (p_[k_] > hi[k_]) hi[k_] = p_[k_];                                     \
        }                                                                             \
    } while (0)
    TGPU_GROW_(s->camera.pos[0], s->camera.pos[1], s->camera.pos[2]);
    for (size_t i = 0; i < (size_t)s->n_tris*3; ++i) TGPU_GROW_(s->tri_pos[i*3], s->tri_pos[i*3 + 1], s->tri_pos[i*3 + 2]);
    for (size_t i = (size_t)s->n_tris*9; i < ((size_t)s->n_tris + s->n_master_tris)*9; ++i) /* masters: their own frame */
        if (fabsf(s->tri_pos[i]) > maxAbs) maxAbs = fabsf(s->tri_pos[i]);
    for (uint32_t i = 0; i < s->n_quads + s->n_master_quads; ++i) {
        const tgpu_quad *q = &s->quads[i];
        for (int c = 0; c < 4; ++c) {
            const float a = (float)(c & 1), b = (float)(c >> 1);
            const float x = q->base[0] + a*q->edge0[0] + b*q->edge1[0], y = q->base[1] + a*q->edge0[1] + b*q->edge1[1],
                        z = q->base[2] + a*q->edge0[2] + b*q->edge1[2];
            if (i < s->n_quads) TGPU_GROW_(x, y, z);
            else { if (fabsf(x) > maxAbs) maxAbs = fabsf(x); if (fabsf(y) > maxAbs) maxAbs = fabsf(y); if (fabsf(z) > maxAbs) maxAbs = fabsf(z); }
        }
    }
    for (uint32_t i = 0; i < s->n_cubes + s->n_master_cubes; ++i) {
        const tgpu_cube *c = &s->cubes[i];
        const float r = sqrtf(c->scale[0]*c->scale[0] + c->scale[1]*c->scale[1] + c->scale[2]*c->scale[2]);
        if (i < s->n_cubes) { TGPU_GROW_(c->pos[0] - r, c->pos[1] - r, c->pos[2] - r); TGPU_GROW_(c->pos[0] + r, c->pos[1] + r, c->pos[2] + r); }
        else for (int k = 0; k < 3; ++k) if (fabsf(c->pos[k]) + r > maxAbs) maxAbs = fabsf(c->pos[k]) + r;
    }
    for (uint32_t i = 0; i < s->n_spheres; ++i) {
        const tgpu_sphere *p = &s->spheres[i];
        TGPU_GROW_(p->pos[0] - p->radius, p->pos[1] - p->radius, p->pos[2] - p->radius);
        TGPU_GROW_(p->pos[0] + p->radius, p->pos[1] + p->radius, p->pos[2] + p->radius);
    }
    for (uint32_t i = 0; i < s->n_disks; ++i) {
        const tgpu_disk *d = &s->disks[i];
        TGPU_GROW_(d->center[0] - d->r, d->center[1] - d->r, d->center[2] - d->r);
        TGPU_GROW_(d->center[0] + d->r, d->center[1] + d->r, d->center[2] + d->r);
    }
    for (uint32_t si = 0; si < s->n_instance_sets; ++si) {
        const tgpu_instances *is = &s->instance_sets[si];
        float r = 0.0f; /* radius of the largest master around its own origin */
        for (uint32_t m = 0; m < is->n_masters; ++m)
            for (int k = 0; k < 3; ++k) {
                const float a = fabsf(is->masters[m].bounds_min[k]), b = fabsf(is->masters[m].bounds_max[k]);
                if (a > r) r = a;
                if (b > r) r = b;
            }
        r *= 1.7320508f;
        for (uint32_t i = 0; i < is->n_instances; ++i) {
            const float *p = is->pos + (size_t)i*3;
            TGPU_GROW_(p[0] - r, p[1] - r, p[2] - r);
            TGPU_GROW_(p[0] + r, p[1] + r, p[2] + r);
        }
    }
#undef TGPU_GROW_
    float d2 = 0.0f;
    for (int k = 0; k < 3; ++k) {
        if (!(hi[k] >= lo[k])) continue;
        d2 += (hi[k] - lo[k])*(hi[k] - lo[k]);
        if (fabsf(lo[k]) > maxAbs) maxAbs = fabsf(lo[k]);
        if (fabsf(hi[k]) > maxAbs) maxAbs = fabsf(hi[k]);
    }
    const float diag = sqrtf(d2);
    return diag > maxAbs ? diag : maxAbs;
}

static inline float tgpu_bvh_pad(const tgpu_scene_desc *s) { return tgpu_scene_extent(s)*1.52587890625e-05f; /* 2^-16 */ }

#endif
