/*
 * rt_oracle.c -- CPU ORACLE for the RayTracing hot path.  TEST INFRASTRUCTURE ONLY.
 *
 * Plain-C restatement of the reference's per-pixel ray-sphere loop
 * (RayTracing/anime_ray.cu:41-88 `kernel`, RayTracing/sphere.cuh:28-44 `Sphere::hit`).
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load it.
 *
 * Pinning status: the reference has no CPU path and no saved frame for the ray tracer
 * ("parity unpinned by the reference"); the only recorded reference output is the
 * single hit() anchor in SURVEY.md Appendix A, checked in tests/test_oracle_pins.py.
 *
 * Build: gcc -O2 -std=c99 -ffp-contract=off (float math, no FMA contraction, IEEE sqrtf / divide).
 */
#include <stdint.h>
#include <math.h>
#include <stddef.h>

#define RT_INF 2e10f                       /* sphere.cuh:10 */
#define RT_SHIFT_W 4                       /* sphere.cuh:11 SPHERE_SHIFT_DATA_WIDTH */

/* sphere.cuh:28-32 -- note the field order r, b, g. 32 bytes. */
typedef struct {
    float r, b, g;
    float radius;
    float x, y, z;
    int32_t idx;
} rt_sphere;

/* Sphere::hit, sphere.cuh:34-44. */
float orc_rt_hit(const rt_sphere *s, float ox, float oy, float *n, const int32_t *shifts)
{
    int x_shift = shifts[RT_SHIFT_W * s->idx], y_shift = shifts[RT_SHIFT_W * s->idx + 1];
    float dx = ox - (s->x + x_shift);
    float dy = oy - (s->y + y_shift);
    if (dx * dx + dy * dy < s->radius * s->radius) {
        float dz = sqrtf(s->radius * s->radius - dx * dx - dy * dy);
        *n = dz / sqrtf(s->radius * s->radius);
        return dz + s->z;
    }
    return -RT_INF;
}

/* kernel, anime_ray.cu:61-87, for the pixel rows [y0, y1) of a dim x dim RGBA8 image.
 * The reference's DIM is compile-time 1024 (anime_ray.cu:24); here it is `dim`.
 * rgba points at the start of the FULL image (offset = x + y*dim, anime_ray.cu:64). */
void orc_rt_render_rows(const rt_sphere *s, int n_spheres, const int32_t *shifts,
                        int dim, int c_shift_x, int c_shift_y, int y0, int y1, uint8_t *rgba)
{
    for (int y = y0; y < y1; ++y) {
        for (int x = 0; x < dim; ++x) {
            int offset = x + y * dim;
            float ox = (float)(x - dim / 2 + c_shift_x);
            float oy = (float)(y - dim / 2 + c_shift_y);
            float r = 0, g = 0, b = 0;
            float maxz = -RT_INF;
            for (int i = 0; i < n_spheres; ++i) {
                float n;
                float t = orc_rt_hit(&s[i], ox, oy, &n, shifts);
                if (t > maxz) {                       /* strict: lowest index wins ties, anime_ray.cu:75 */
                    float fscale = n;
                    r = s[i].r * fscale;
                    g = s[i].g * fscale;
                    b = s[i].b * fscale;
                    maxz = t;
                }
            }
            rgba[(size_t)offset * 4 + 0] = (uint8_t)(int)(r * 255);
            rgba[(size_t)offset * 4 + 1] = (uint8_t)(int)(g * 255);
            rgba[(size_t)offset * 4 + 2] = (uint8_t)(int)(b * 255);
            rgba[(size_t)offset * 4 + 3] = 255;
        }
    }
}

void orc_rt_render(const rt_sphere *s, int n_spheres, const int32_t *shifts,
                   int dim, int c_shift_x, int c_shift_y, uint8_t *rgba)
{
    orc_rt_render_rows(s, n_spheres, shifts, dim, c_shift_x, c_shift_y, 0, dim, rgba);
}

/* Animation state, sphere.cuh:50-61 initSpheres (minus the cuRAND state, which only the
 * stochastic update kernels consume): shifts = {0, 0, (i%5+1)*5, (i%2)*2-1}, angle = 0. */
void orc_rt_init_shifts(int n_spheres, int32_t *shifts, double *angles)
{
    for (int i = 0; i < n_spheres; ++i) {
        shifts[RT_SHIFT_W * i] = shifts[RT_SHIFT_W * i + 1] = 0;
        shifts[RT_SHIFT_W * i + 2] = (i % 5 + 1) * 5;
        shifts[RT_SHIFT_W * i + 3] = (i % 2) * 2 - 1;
        angles[i] = 0.0;
    }
}
