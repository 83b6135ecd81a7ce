/*
 * mi355rt.h -- C ABI of libmi355rt.so: MI355X-native (HIP, gfx950) per-pixel ray-sphere loop.
 *
 * Drop-in boundary for the RayTracing hot path of Asichurter/GPU-Computing-Course: the `kernel`
 * launch at RayTracing/anime_ray.cu:126 (body anime_ray.cu:41-88, Sphere::hit sphere.cuh:34-44).
 * Host pointers in / out, the library owns device memory behind an opaque context.
 * Every function returns int: 0 = ok, <0 = -(hipError_t) or RT_ERR_*.  No CPU fallback.
 */
#ifndef MI355RT_H
#define MI355RT_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* sphere.cuh:28-32 -- field order r, b, g (sic); 32 bytes. */
typedef struct RtSphere {
    float r, b, g;
    float radius;
    float x, y, z;
    int32_t idx;          /* index into the shift table (sphere.cuh:35) */
} RtSphere;

enum {
    RT_OK            = 0,
    RT_ERR_ARG       = -2001,
    RT_ERR_NO_DEVICE = -2003
};

/* How the per-pixel loop is organised.  Both produce identical pixels. */
enum {
    RT_MODE_BRUTE  = 0,   /* every pixel loops over all spheres, as anime_ray.cu:70-82 does            */
    RT_MODE_BINNED = 1    /* per image tile, spheres that cannot touch the tile are culled first with an */
                          /* exact test built from the same float operations as Sphere::hit             */
};

typedef struct rt_ctx rt_ctx;

typedef struct rt_stats {
    float    ms_render;           /* device time of the last render (HIP events on the context stream) */
    uint32_t mode;
    uint64_t sphere_tests;        /* hit() evaluations performed by the last render                    */
} rt_stats;

/* anime_ray.cu:226-236 + allocateSpheresOnConstant (anime_ray.cu:163-183): allocate the frame buffer
 * (dim x dim RGBA8) and upload n_spheres spheres.  The reference fixes DIM=1024, SPHERES=500 at
 * compile time (anime_ray.cu:24, sphere.cuh:22); here both are run-time.  dim must be a multiple of 64. */
int rt_create(rt_ctx **out, const RtSphere *spheres, int32_t n_spheres, int32_t dim);
void rt_destroy(rt_ctx *ctx);                                   /* anime_ray.cu:145-158 cleanup */
int rt_set_spheres(rt_ctx *ctx, const RtSphere *spheres);       /* re-upload (same count)       */
int rt_set_mode(rt_ctx *ctx, int mode);                         /* default RT_MODE_BINNED       */

/* anime_ray.cu:126-131: kernel<<<>>>(bitmap, c_shift_x, c_shift_y, shifts, spheres) + D2H of the
 * frame.  shifts4: n_spheres x 4 int32 (sphere.cuh:11: [0]=x shift, [1]=y shift, [2],[3] animation
 * state the pixel loop does not read).  rgba_out: dim*dim*4 bytes or NULL (render only, keep on device). */
int rt_render(rt_ctx *ctx, const int32_t *shifts4, int32_t c_shift_x, int32_t c_shift_y, uint8_t *rgba_out);

/* Same, image rows [y0, y1) only (multi-GPU: the image shards by rows, spheres replicated). rgba_out
 * receives (y1-y0)*dim*4 bytes.  y0, y1 multiples of 64. */
int rt_render_rows(rt_ctx *ctx, const int32_t *shifts4, int32_t c_shift_x, int32_t c_shift_y,
                   int32_t y0, int32_t y1, uint8_t *rgba_out);

/* sphere.cuh:50-61 initSpheres, the deterministic part: shifts = {0,0,(i%5+1)*5,(i%2)*2-1}, angles = 0. */
int rt_init_shifts(int32_t n_spheres, int32_t *shifts4, double *angles);

/* The same frame `frames` times back to back, as a loop that queues frames without a host round trip per frame submits
 * them (measurement aid: a time stamp costs a few idle microseconds next to the kernel that carries it, and a single frame
 * carries two).  rt_stats.ms_render = device time per frame, first kernel start to last kernel end / frames. */
int rt_render_repeat(rt_ctx *ctx, const int32_t *shifts4, int32_t c_shift_x, int32_t c_shift_y, int32_t frames, uint8_t *rgba_out);

/* Device-resident animation state.  After rt_anim_init, rt_render / rt_render_rows accept
 * shifts4 == NULL and read the state the kernels below maintain; generate_frame's frame counters
 * (anime_ray.cu:101-125: camera shake, SPHERE_FRAME_PER_SHAKE, SPHERE_SHAKE_TYPE) stay with the caller.
 * Random stream: XORWOW as cuRAND's curand_init(i, 0, 0) / curand() (see oracle/rt_oracle.c for its parity status). */
int rt_anim_init(rt_ctx *ctx);                                                     /* anime_ray.cu:251 initSpheres<<<128,1>>>, sphere.cuh:50-61 */
int rt_anim_axis_move(rt_ctx *ctx, int32_t shake_width);                           /* anime_ray.cu:119 updateSphereShiftsWithAxisMove, sphere.cuh:66-77 (SPHERE_SHAKE_WIDTH 35) */
int rt_anim_curve_move(rt_ctx *ctx);                                               /* anime_ray.cu:121 updateSphereShiftsWithCurveMove, sphere.cuh:82-97 */
int rt_anim_update_speed_angle(rt_ctx *ctx, int32_t update_prob, int32_t max_speed); /* anime_ray.cu:122 updateSphereCurveSpeedAngle, sphere.cuh:102-118 (1, 18) */
/* Read the state back (any pointer may be NULL): shifts4 n x 4 int32, angles n doubles, rng6 n x 6 uint32 {v[5], d}. */
int rt_anim_get_state(rt_ctx *ctx, int32_t *shifts4, double *angles, uint32_t *rng6);

/* generate_frame (anime_ray.cu:98-140) `frames` times, the frames staying on the device (the last one is copied to rgba_last unless NULL): per frame the
 * spheres move -- shake 0: not at all; 1: updateSphereShiftsWithAxisMove (anime_ray.cu:119; shake_width); 2: updateSphereShiftsWithCurveMove +
 * updateSphereCurveSpeedAngle (anime_ray.cu:121-122; update_prob, max_speed) -- and kernel<<<>>> (anime_ray.cu:126) renders them; the camera offsets are
 * the caller's (anime_ray.cu:101-108) and stay fixed over the call.  Needs rt_anim_init.  The state and the last frame are what `frames` rounds of
 * rt_anim_* + rt_render leave.  In binned mode the loop is ONE launch per frame (+ one in front): frame f renders while the spheres move on to frame
 * f + 1 and are binned for it in the same launch (two list sets).  rt_stats.ms_render = device time per frame over the call. */
int rt_anim_loop(rt_ctx *ctx, int32_t frames, int32_t shake, int32_t shake_width, int32_t update_prob, int32_t max_speed,
                 int32_t c_shift_x, int32_t c_shift_y, uint8_t *rgba_last);

int rt_get_stats(rt_ctx *ctx, rt_stats *out);
const char *rt_version(void);

#ifdef __cplusplus
}
#endif
#endif /* MI355RT_H */
