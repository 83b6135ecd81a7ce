// cd_main.cpp -- host harness of the collision path, the MI355X-side twin of the reference's
// CollisionDetection/main.cu:47-174: load -> sort -> hierarchy -> refit -> verify -> find collisions ->
// print, with the reference's stage lines and result format, every stage one call through the C ABI
// (include/mi355cd.h).  Usage: cd_main <file.obj> [--frame ref|auto] [--cap N] [--brute]
#include "mi355cd.h"

#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <set>
#include <string>
#include <vector>

// HANDLE_ERROR convention of common/book.h:21-30: print "<what> in <file> at line <n>" and exit.
#define CD_CHECK(call)                                                                  \
    do {                                                                                \
        int rc_ = (call);                                                               \
        if (rc_ < 0) { std::printf("%s failed with status %d in %s at line %d\n", #call, rc_, __FILE__, __LINE__); std::exit(EXIT_FAILURE); } \
    } while (0)

static void print_time(const char *opname, float ms) { std::printf("\nTime of %s:  %3.1f ms\n", opname, ms); }   // main.cu:19-24

// main.cu:33-45
static void make_and_print_set(const uint32_t *data, size_t num, const char *title)
{
    std::set<uint32_t> dset(data, data + num);
    std::printf("\n\n%s (%zu points in total):\n", title, dset.size());
    for (uint32_t v : dset) std::printf("%u\n", v);
}

int main(int argc, char **argv)
{
    if (argc < 2) { std::printf("usage: %s <file.obj> [--frame ref|auto] [--cap N] [--brute]\n", argv[0]); return 2; }
    int frame = CD_FRAME_REFERENCE;
    uint64_t cap = 1u << 20;
    bool brute = false;
    for (int i = 2; i < argc; ++i) {
        if (!std::strcmp(argv[i], "--frame") && i + 1 < argc) frame = std::strcmp(argv[++i], "auto") ? CD_FRAME_REFERENCE : CD_FRAME_AUTO;
        else if (!std::strcmp(argv[i], "--cap") && i + 1 < argc) cap = std::strtoull(argv[++i], nullptr, 10);
        else if (!std::strcmp(argv[i], "--brute")) brute = true;
    }
    const auto t_begin = std::chrono::steady_clock::now();                       // main.cu:55 m_start

    double *verts = nullptr; uint32_t *vidx = nullptr; uint32_t nv = 0, nt = 0;
    {   // main.cu:64 loadObj -- multi-threaded parse through the C ABI; Morton codes and the sort moved to the GPU
        const int rc = cd_load_obj(argv[1], &verts, &nv, &vidx, &nt, 0);
        if (rc == CD_ERR_IO) { std::printf("* ERROR: loading obj:(%s) file is not good\n", argv[1]); return 1; }                 // load_obj.h:33
        if (rc == CD_ERR_FORMAT) { std::printf("* ERROR: vertex / FaceMtl not in wanted format in OBJLoader\n"); return 1; }      // load_obj.h:59,71
        if (rc == CD_ERR_INDEX) { std::printf("* ERROR: Vertex of face out of bound\n"); return 1; }                             // load_obj.h:78
        CD_CHECK(rc);
    }
    float xmin = 1000, ymin = 1000, zmin = 1000;                                                      // load_obj.h:39,53-55
    for (uint32_t i = 0; i < nv; ++i) {
        if ((float)verts[3 * i] < xmin) xmin = (float)verts[3 * i];
        if ((float)verts[3 * i + 1] < ymin) ymin = (float)verts[3 * i + 1];
        if ((float)verts[3 * i + 2] < zmin) zmin = (float)verts[3 * i + 2];
    }
    std::printf("\nObj File Loaded:\n- %u vertexes loaded\n- %u triangles loaded\n", nv, nt);       // load_obj.h:117-119
    std::printf("- xmin=%f, ymin=%f, z=%f\n", xmin, ymin, zmin);                                       // load_obj.h:122

    cd_ctx *ctx = nullptr;
    CD_CHECK(cd_create(&ctx, verts, nv, vidx, nullptr, nt));                                          // main.cu:78-88
    cd_free_obj(verts, vidx);
    CD_CHECK(cd_set_morton_frame(ctx, frame, nullptr, nullptr));
    cd_stats st;

    CD_CHECK(cd_morton_sort(ctx));                                                                   // load_obj.h:89-107, on the GPU
    CD_CHECK(cd_get_stats(ctx, &st));
    print_time("mortonCodes", st.ms_morton);
    print_time("sortByKey", st.ms_sort);
    {
        std::vector<uint64_t> keys(nt);
        CD_CHECK(cd_export_keys(ctx, keys.data(), nullptr));
        uint32_t wrong = 0;
        for (uint32_t i = 0; i + 1 < nt; ++i) if (keys[i] >= keys[i + 1]) wrong++;                   // load_obj.h:109-115
        std::printf("- wrong morton sort count: %u\n", wrong);
        std::printf("- First morton code: %llu, last morton code: %llu\n\n", (unsigned long long)keys[0], (unsigned long long)keys[nt - 1]);
    }

    uint32_t parent_wrong = 0;
    CD_CHECK(cd_build_hierarchy(ctx, &parent_wrong));                                                // main.cu:92,99
    CD_CHECK(cd_get_stats(ctx, &st));
    print_time("generateHierarchyParallel", st.ms_hierarchy);
    std::printf("\n- generateHierarchyParallel check result: wrongParentNum = %u, with total nodes=%u\n\n", parent_wrong, nt - 1);   // main.cu:103

    CD_CHECK(cd_refit_boxes(ctx));                                                                   // main.cu:107
    CD_CHECK(cd_get_stats(ctx, &st));
    print_time("calBoundingBox", st.ms_refit);

    uint32_t ci[5], cl[4], ct = 0;
    CD_CHECK(cd_check_internal(ctx, ci));                                                            // main.cu:115
    CD_CHECK(cd_get_stats(ctx, &st)); print_time("checkInternalNodes", st.ms_check);
    std::printf("\n- Internal node check result: nullParentnum = %u, wrongBoundCount=%u, nullChildCount=%u, notInternalCount=%u, uninitBoxCount=%u, with total nodes=%u\n\n",
                ci[0], ci[1], ci[2], ci[3], ci[4], nt - 1);                                          // main.cu:119
    CD_CHECK(cd_check_leaves(ctx, cl));                                                              // main.cu:123
    CD_CHECK(cd_get_stats(ctx, &st)); print_time("checkLeafNodes", st.ms_check);
    std::printf("\n- Leaf node check result: nullParentnum = %u, nullTriangle=%u, notLeafCount=%u, illegalBoxCount=%u, with total nodes=%u\n\n",
                cl[0], cl[1], cl[2], cl[3], nt);                                                     // main.cu:127
    CD_CHECK(cd_check_triangle_idx(ctx, nv, &ct));                                                   // main.cu:131 (632674 -> nv)
    CD_CHECK(cd_get_stats(ctx, &st)); print_time("checkTriangleIdx", st.ms_check);
    std::printf("\n- Triangle check result: illegal triangle vidx num = %u, with total triangles=%u\n\n", ct, nt);      // main.cu:135
    std::printf("\n$ triangle num = %u, mortons num = %u, vertex num = %u\n\n", nt, nt, nv);         // main.cu:136

    std::vector<uint32_t> pairs(2 * cap);
    uint64_t n_pairs = 0;
    int rc = cd_find_collisions(ctx, pairs.data(), cap, &n_pairs);                                   // main.cu:142-146
    CD_CHECK(rc);
    CD_CHECK(cd_get_stats(ctx, &st));
    print_time("findCollisions", st.ms_traverse);
    std::printf("\n\n- contact val = %llu\n", (unsigned long long)n_pairs);                          // main.cu:147
    if (rc == CD_OVERFLOW) std::printf("* WARNING: %llu pairs found but capacity is %llu; rerun with --cap\n", (unsigned long long)n_pairs, (unsigned long long)cap);
    const uint64_t shown = n_pairs < cap ? n_pairs : cap;
    std::printf("\nCollision pair (%llu triangle pairs in total):\n", (unsigned long long)n_pairs);  // main.cu:149
    for (uint64_t i = 0; i < shown; ++i) std::printf("%07u - %07u\n", pairs[2 * i], pairs[2 * i + 1]);   // main.cu:151
    {   // main.cu:154 makeAndPrintSet -- the set is built on the device (sort + unique), printed in the reference's format
        std::vector<uint32_t> ids(2 * shown + 1);
        uint64_t n_ids = 0;
        const int rs = cd_collision_triangles(ctx, ids.data(), ids.size(), &n_ids);
        if (rs == CD_OK) {
            std::printf("\n\n%s (%llu points in total):\n", "Collision Triangles:", (unsigned long long)n_ids);
            for (uint64_t i = 0; i < n_ids; ++i) std::printf("%u\n", ids[i]);
        } else {
            make_and_print_set(pairs.data(), 2 * shown, "Collision Triangles:");     // truncated list: host fallback on what was returned
        }
    }
    std::printf("\n- pairs tested (leaf AABB hits) = %llu, node visits = %llu\n", (unsigned long long)st.pairs_tested, (unsigned long long)st.node_visits);

    if (brute) {                                                                                     // check.cuh:117-141
        uint64_t nb = 0;
        CD_CHECK(cd_brute_force(ctx, 1, nullptr, 0, &nb));
        std::printf("- brute force (checkDirectComp + leaf AABB filter) contact count = %llu\n", (unsigned long long)nb);
    }
    cd_destroy(ctx);                                                                                 // main.cu:156-163
    std::printf("- Successfully Return\n");                                                          // main.cu:168
    const double total_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_begin).count();
    print_time("Total Time", (float)total_ms);                                                       // main.cu:170-171
    return 0;
}
