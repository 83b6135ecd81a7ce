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

/* ------------------------------------------------------------------------------------------------
 * Animation state kernels, sphere.cuh:50-118.  TEST INFRASTRUCTURE like the rest.
 *
 * PARITY UNPINNED for the random stream: curandStateXORWOW / curand_init / curand come from NVIDIA's
 * <curand_kernel.h>, which is not in /root/reference and not in this image.  The generator below is a
 * restatement of the published XORWOW algorithm as that header implements it (Marsaglia's xorwow: five
 * 32-bit xorshift words + a Weyl counter stepping by 362437; curand_init(seed, 0, 0) scrambles the seed
 * with two odd multipliers and applies no skip-ahead for subsequence 0 / offset 0).  rocRAND's XORWOW,
 * the only other implementation on this box, seeds differently (different xor constants), so there is
 * nothing here to check the seeding constants against; everything downstream of curand() -- dev_rnd,
 * the shift / speed / direction / angle updates -- is plain IEEE double / int arithmetic and is exact.
 * cosf / sinf (sphere.cuh:88-89): CUDA's device cosf / sinf are not correctly rounded (<= 2 ulp); here
 * they are the correctly rounded float cosine / sine of (float)angle.  `speed * cosf(a)` is truncated to
 * int, so at exact boundaries (e.g. speed 10, a = pi/3) the reference may differ by one pixel.
 * ------------------------------------------------------------------------------------------------ */
typedef struct { uint32_t v[5]; uint32_t d; } orc_xorwow;

void orc_xorwow_init(orc_xorwow *s, uint64_t seed)                  /* curand_init(seed, 0, 0, s), sphere.cuh:53 */
{
    const uint32_t s0 = (uint32_t)seed ^ 0xaad26b49u, s1 = (uint32_t)(seed >> 32) ^ 0xf7dcefddu;
    const uint32_t t0 = 1099087573u * s0, t1 = 2591861531u * s1;
    s->d = 6615241u + t1 + t0;
    s->v[0] = 123456789u + t0; s->v[1] = 362436069u ^ t0; s->v[2] = 521288629u + t1;
    s->v[3] = 88675123u ^ t1;  s->v[4] = 5783321u + t0;
}

uint32_t orc_xorwow_next(orc_xorwow *s)                             /* curand(&state) */
{
    const uint32_t t = s->v[0] ^ (s->v[0] >> 2);
    s->v[0] = s->v[1]; s->v[1] = s->v[2]; s->v[2] = s->v[3]; s->v[3] = s->v[4];
    s->v[4] = (s->v[4] ^ (s->v[4] << 4)) ^ (t ^ (t << 1));
    s->d += 362437u;
    return s->v[4] + s->d;
}

/* dev_rnd(x, s), sphere.cuh:26: curand(&s) % 1000000 * 1.0 / 1000000 * x  (double) */
static double orc_dev_rnd(int x, orc_xorwow *s) { return orc_xorwow_next(s) % 1000000u * 1.0 / 1000000 * x; }

#define ORC_PI 3.1415926535898                                      /* sphere.cuh:19 */

void orc_rt_anim_init(int n, orc_xorwow *states, int32_t *shifts, double *angles)          /* initSpheres, sphere.cuh:50-61 */
{
    for (int i = 0; i < n; ++i) orc_xorwow_init(&states[i], (uint64_t)i);
    orc_rt_init_shifts(n, shifts, angles);
}

void orc_rt_anim_axis_move(int n, orc_xorwow *states, int32_t *shifts, int shake_width)    /* sphere.cuh:66-77 */
{
    for (int i = 0; i < n; ++i) {
        const int x_shift = (int)orc_dev_rnd(shake_width, &states[i]);
        const int y_shift = (int)orc_dev_rnd(shake_width, &states[i]);
        shifts[RT_SHIFT_W * i] = x_shift; shifts[RT_SHIFT_W * i + 1] = y_shift;
    }
}

void orc_rt_anim_curve_move(int n, int32_t *shifts, double *angles)                        /* sphere.cuh:82-97 */
{
    for (int i = 0; i < n; ++i) {
        const int speed = shifts[RT_SHIFT_W * i + 2];
        const float a = (float)angles[i];                           /* cosf(double) converts its argument */
        const int x_shift = (int)((float)speed * (float)cos((double)a));
        const int y_shift = (int)((float)speed * (float)sin((double)a));
        shifts[RT_SHIFT_W * i] += x_shift; shifts[RT_SHIFT_W * i + 1] += y_shift;
        angles[i] = fmod(angles[i] + ORC_PI / 12 * shifts[RT_SHIFT_W * i + 3], 2 * ORC_PI);
    }
}

void orc_rt_anim_speed_angle(int n, orc_xorwow *states, int32_t *shifts, double *angles,
                             int update_prob, int max_speed)                                /* sphere.cuh:102-118 */
{
    for (int i = 0; i < n; ++i) {
        const int p = (int)orc_dev_rnd(10, &states[i]);
        if (p >= update_prob) continue;
        shifts[RT_SHIFT_W * i + 2] = (int)orc_dev_rnd(max_speed, &states[i]);
        shifts[RT_SHIFT_W * i + 3] = ((int)orc_dev_rnd(2, &states[i])) * 2 - 1;
        angles[i] = fmod(angles[i] + ((int)orc_dev_rnd(2, &states[i])) * ORC_PI, 2 * ORC_PI);
    }
}
