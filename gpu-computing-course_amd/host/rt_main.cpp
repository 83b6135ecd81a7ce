// rt_main.cpp -- headless host harness of the ray-tracing path, the twin of RayTracing/anime_ray.cu's
// main() / generate_frame() (anime_ray.cu:99-139, 208-256) without the GLUT window: build the sphere
// scene with the reference's host rand() recipe, render frames through the C ABI (include/mi355rt.h),
// print the reference's per-frame timing line and optionally write the last frame as a binary PPM.
// Usage: rt_main [--dim D] [--spheres S] [--frames F] [--mode binned|brute] [--shake none|axis|curve] [--camera-shake] [--ppm out.ppm]
#include "mi355rt.h"

#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#define rnd(x) (x * rand() / RAND_MAX)            // anime_ray.cu:31

#define RT_CHECK(call)                                                                  \
    do {                                                                                \
        int rc_ = (call);                                                               \
        if (rc_ < 0) { std::printf("%s failed with status %d in %s at line %d\n", #call, rc_, __FILE__, __LINE__); std::exit(EXIT_FAILURE); } \
    } while (0)

int main(int argc, char **argv)
{
    int dim = 1024, n = 500, frames = 4, mode = RT_MODE_BINNED;      // anime_ray.cu:24, sphere.cuh:22
    const char *ppm = nullptr;
    bool camera_shake = false;
    int shake = 2;                                                    // SPHERE_SHAKE_TYPE 1 (curve), sphere.cuh:13; 0 none, 1 axis
    for (int i = 1; i < argc; ++i) {
        if (!std::strcmp(argv[i], "--dim") && i + 1 < argc) dim = std::atoi(argv[++i]);
        else if (!std::strcmp(argv[i], "--spheres") && i + 1 < argc) n = std::atoi(argv[++i]);
        else if (!std::strcmp(argv[i], "--frames") && i + 1 < argc) frames = std::atoi(argv[++i]);
        else if (!std::strcmp(argv[i], "--mode") && i + 1 < argc) mode = std::strcmp(argv[++i], "brute") ? RT_MODE_BINNED : RT_MODE_BRUTE;
        else if (!std::strcmp(argv[i], "--ppm") && i + 1 < argc) ppm = argv[++i];
        else if (!std::strcmp(argv[i], "--camera-shake")) camera_shake = true;
        else if (!std::strcmp(argv[i], "--shake") && i + 1 < argc) { ++i; shake = !std::strcmp(argv[i], "none") ? 0 : (!std::strcmp(argv[i], "axis") ? 1 : 2); }
    }
    // allocateSpheresOnConstant, anime_ray.cu:163-176 (ranges scale with the image: the reference's
    // 1000-wide world belongs to its 1024-pixel image)
    std::vector<RtSphere> s(n);
    const float world = 1000.0f * dim / 1024.0f;
    for (int i = 0; i < n; i++) {
        s[i].r = rnd(1.0f); s[i].g = rnd(1.0f); s[i].b = rnd(1.0f);
        s[i].x = rnd(world) - world / 2; s[i].y = rnd(world) - world / 2; s[i].z = rnd(world) - world / 2;
        s[i].radius = rnd(20.0f) + 8;
        s[i].idx = i;
    }
    rt_ctx *ctx = nullptr;
    RT_CHECK(rt_create(&ctx, s.data(), n, dim));                        // anime_ray.cu:226-248
    RT_CHECK(rt_set_mode(ctx, mode));
    RT_CHECK(rt_anim_init(ctx));                                        // anime_ray.cu:251 initSpheres<<<SPHERE_BLOCK, 1>>>
    std::vector<uint8_t> frame((size_t)dim * dim * 4);
    int c_shift_x = 0, c_shift_y = 0, camera_frame_count_loop = 0, sphere_frame_count_loop = 0;
    for (int f = 0; f < frames; ++f) {                                  // generate_frame, anime_ray.cu:99-139
        const auto t0 = std::chrono::steady_clock::now();
        if (camera_shake && (camera_frame_count_loop = ++camera_frame_count_loop % 2) == 0) {   // ENABLE_CAMERA_SHAKE (false), CAM_FRAME_PER_SHAKE 2, anime_ray.cu:25-27,101-108
            const int x_rd_val = rnd(15), y_rd_val = rnd(15);                                      // CAM_SHAKE_WIDTH 15
            c_shift_x += (x_rd_val - 15 / 2); c_shift_y += (y_rd_val - 15 / 2);
        }
        if (shake && (sphere_frame_count_loop = ++sphere_frame_count_loop % 4) == 0) { // SPHERE_FRAME_PER_SHAKE 4, anime_ray.cu:115-125
            if (shake == 1) RT_CHECK(rt_anim_axis_move(ctx, 35));                       // SPHERE_SHAKE_WIDTH
            else { RT_CHECK(rt_anim_curve_move(ctx)); RT_CHECK(rt_anim_update_speed_angle(ctx, 1, 18)); }   // SPHERE_UPDATE_CURVE_PROB, SPHERE_MAX_SPEED
        }
        RT_CHECK(rt_render(ctx, nullptr, c_shift_x, c_shift_y, frame.data()));   // kernel + D2H, anime_ray.cu:126-131; shifts stay on the device
        const double ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
        rt_stats st; RT_CHECK(rt_get_stats(ctx, &st));
        std::printf("Time to generate a frame:  %3.1f ms   (kernel %.3f ms, %llu sphere tests)\n", ms, st.ms_render, (unsigned long long)st.sphere_tests);
    }
    if (ppm) {
        FILE *f = std::fopen(ppm, "wb");
        if (f) {
            std::fprintf(f, "P6\n%d %d\n255\n", dim, dim);
            for (size_t p = 0; p < (size_t)dim * dim; ++p) std::fwrite(&frame[4 * p], 1, 3, f);
            std::fclose(f);
        }
    }
    rt_destroy(ctx);                                                    // cleanup, anime_ray.cu:145-158
    return 0;
}
