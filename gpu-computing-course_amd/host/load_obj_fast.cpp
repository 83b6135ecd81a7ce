// load_obj_fast.cpp -- parallel OBJ ingest behind the C ABI (cd_load_obj / cd_free_obj, include/mi355cd.h).
// The step right before the hot path in the reference harness: CollisionDetection/load_obj.h:24-103 parses the
// file line by line with getline + sscanf on one thread ("Total Time" 187 ms vs 71 ms of kernels).
// Here the file is read once, cut into per-thread chunks at line boundaries, and parsed in two passes
// (count, then fill at prefix-summed offsets), so vertex / face order is exactly the file order.
// Dialect = the reference's: `v %f %f %f` (parsed as float, widened to double, load_obj.h:38,50-52) and
// `f %d/%d %d/%d %d/%d` with 1-based indices (load_obj.h:68,81-83); every other line is ignored.
// Morton codes and sorting are NOT done here -- they moved to the GPU (cd_morton_sort).
#include "../../include/mi355cd.h"

#include <cerrno>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <thread>
#include <vector>

namespace {

struct Chunk { size_t begin = 0, end = 0; size_t nv = 0, nf = 0; int err = 0; size_t err_line_pos = 0; };

inline const char *line_end(const char *p, const char *e) { while (p < e && *p != '\n') ++p; return p; }

// sscanf(buffer, "v %f %f %f") == 3
inline bool parse_v(const char *p, const char *e, float out[3])
{
    ++p;                                                   // 'v'
    for (int k = 0; k < 3; ++k) {
        while (p < e && (*p == ' ' || *p == '\t' || *p == '\r')) ++p;
        if (p >= e) return false;
        char *q = nullptr;
        out[k] = std::strtof(p, &q);                       // correctly rounded, as %f
        if (q == p) return false;
        p = q;
    }
    return true;
}
// sscanf(buffer, "f %d/%d %d/%d %d/%d") == 6
inline bool parse_f(const char *p, const char *e, long v[3])
{
    ++p;                                                   // 'f'
    for (int k = 0; k < 3; ++k) {
        while (p < e && (*p == ' ' || *p == '\t' || *p == '\r')) ++p;
        char *q = nullptr;
        v[k] = std::strtol(p, &q, 10);
        if (q == p || q >= e || *q != '/') return false;
        p = q + 1;
        (void)std::strtol(p, &q, 10);                      // the texture index is parsed and dropped (load_obj.h:68 nv.vIdx)
        if (q == p) return false;
        p = q;
    }
    return true;
}

}  // namespace

extern "C" {

int cd_load_obj(const char *path, double **verts_xyz, uint32_t *nv, uint32_t **vidx3, uint32_t *nt, int threads)
{
    if (!path || !verts_xyz || !nv || !vidx3 || !nt) return CD_ERR_ARG;
    *verts_xyz = nullptr; *vidx3 = nullptr; *nv = 0; *nt = 0;
    FILE *f = std::fopen(path, "rb");
    if (!f) return CD_ERR_IO;                              // load_obj.h:31-35 "file is not good"
    std::fseek(f, 0, SEEK_END);
    const long sz = std::ftell(f);
    std::fseek(f, 0, SEEK_SET);
    if (sz <= 0) { std::fclose(f); return CD_ERR_FORMAT; }
    std::vector<char> buf((size_t)sz + 2);
    const size_t got = std::fread(buf.data(), 1, (size_t)sz, f);
    std::fclose(f);
    if (got != (size_t)sz) return CD_ERR_IO;
    buf[(size_t)sz] = '\n';                                // a last line without a newline still ends in one ...
    buf[(size_t)sz + 1] = '\0';                            // ... and strtol / strtof, which skip newlines as white space, stop at the NUL
    const char *base = buf.data(), *end = base + sz;

    int nth = threads > 0 ? threads : (int)std::thread::hardware_concurrency();
    if (nth < 1) nth = 1;
    if (nth > 64) nth = 64;
    if ((size_t)sz < (size_t)nth * 65536) nth = (int)((size_t)sz / 65536) + 1;
    std::vector<Chunk> ch(nth);
    for (int t = 0; t < nth; ++t) {                        // cut at line boundaries
        size_t b = (size_t)sz * t / nth, e = (size_t)sz * (t + 1) / nth;
        if (t > 0) { const char *p = line_end(base + b - 1, end); b = (size_t)(p - base) + 1; if (b > (size_t)sz) b = (size_t)sz; }
        if (t + 1 < nth) { const char *p = line_end(base + e - 1, end); e = (size_t)(p - base) + 1; if (e > (size_t)sz) e = (size_t)sz; }
        ch[t].begin = b; ch[t].end = e;
    }
    auto for_lines = [&](const Chunk &c, auto &&fn) {
        const char *p = base + c.begin, *e = base + c.end;
        while (p < e) {
            const char *le = line_end(p, end);
            fn(p, le);
            p = le + 1;
        }
    };
    // pass 1: count
    {
        std::vector<std::thread> th;
        for (int t = 0; t < nth; ++t) th.emplace_back([&, t] {
            size_t cv = 0, cf = 0;
            for_lines(ch[t], [&](const char *p, const char *le) {
                if (le - p >= 2 && p[1] == ' ') { if (p[0] == 'v') ++cv; else if (p[0] == 'f') ++cf; }   // load_obj.h:48,64
            });
            ch[t].nv = cv; ch[t].nf = cf;
        });
        for (auto &x : th) x.join();
    }
    size_t tv = 0, tf = 0;
    std::vector<size_t> voff(nth), foff(nth);
    for (int t = 0; t < nth; ++t) { voff[t] = tv; foff[t] = tf; tv += ch[t].nv; tf += ch[t].nf; }
    if (tv == 0 || tf == 0 || tv > 0xfffffff0ull || tf > 0xfffffff0ull) return CD_ERR_FORMAT;
    double *V = (double *)std::malloc(sizeof(double) * 3 * tv);
    uint32_t *F = (uint32_t *)std::malloc(sizeof(uint32_t) * 3 * tf);
    if (!V || !F) { std::free(V); std::free(F); return CD_ERR_ARG; }
    // pass 2: fill
    {
        std::vector<std::thread> th;
        for (int t = 0; t < nth; ++t) th.emplace_back([&, t] {
            size_t iv = voff[t], jf = foff[t];
            for_lines(ch[t], [&](const char *p, const char *le) {
                if (ch[t].err || le - p < 2 || p[1] != ' ') return;
                if (p[0] == 'v') {
                    float x[3];
                    if (!parse_v(p, le, x)) { ch[t].err = CD_ERR_FORMAT; ch[t].err_line_pos = (size_t)(p - base); return; }   // load_obj.h:57-61
                    V[3 * iv] = (double)x[0]; V[3 * iv + 1] = (double)x[1]; V[3 * iv + 2] = (double)x[2]; ++iv;               // load_obj.h:52
                } else if (p[0] == 'f') {
                    long v[3];
                    if (!parse_f(p, le, v)) { ch[t].err = CD_ERR_FORMAT; ch[t].err_line_pos = (size_t)(p - base); return; }   // load_obj.h:69-74
                    // load_obj.h:76-79: a face may only use vertices already read (the reference warns, then reads out of bounds)
                    for (int k = 0; k < 3; ++k)
                        if (v[k] < 1 || (size_t)v[k] > iv) { ch[t].err = CD_ERR_INDEX; ch[t].err_line_pos = (size_t)(p - base); return; }
                    F[3 * jf] = (uint32_t)(v[0] - 1); F[3 * jf + 1] = (uint32_t)(v[1] - 1); F[3 * jf + 2] = (uint32_t)(v[2] - 1); ++jf;   // load_obj.h:81-83
                }
            });
        });
        for (auto &x : th) x.join();
    }
    for (int t = 0; t < nth; ++t) if (ch[t].err) { const int e = ch[t].err; std::free(V); std::free(F); return e; }
    *verts_xyz = V; *vidx3 = F; *nv = (uint32_t)tv; *nt = (uint32_t)tf;
    return CD_OK;
}

void cd_free_obj(double *verts_xyz, uint32_t *vidx3) { std::free(verts_xyz); std::free(vidx3); }

}  // extern "C"
