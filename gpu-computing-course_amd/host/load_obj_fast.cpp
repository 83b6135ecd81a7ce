// load_obj_fast.cpp -- parallel OBJ ingest behind the C ABI (cd_load_obj / cd_free_obj, include/mi355cd.h).
// The step right before the hot path in the reference harness: CollisionDetection/load_obj.h:24-103 parses the
// file line by line with getline + sscanf on one thread ("Total Time" 187 ms vs 71 ms of kernels).
// Here the file is read once, cut into per-thread chunks at line boundaries, and parsed in two passes
// (count, then fill at prefix-summed offsets), so vertex / face order is exactly the file order.
// Dialect = the reference's: `v %f %f %f` (parsed as float, widened to double, load_obj.h:38,50-52) and
// `f %d/%d %d/%d %d/%d` with 1-based indices (load_obj.h:68,81-83); every other line is ignored.
// Morton codes and sorting are NOT done here -- they moved to the GPU (cd_morton_sort).
#include "../../include/mi355cd.h"

// Round 5 (bench.py's from_obj leg: 35 ms of parsing in front of a 0.44 ms first step): the file is mapped, not copied (the page cache's pages are the
// buffer), and numbers go through parsers of their own -- strtof / strtol cost 60-100 ns a number, 10.5 M numbers in BASELINE config 3's file:
//   * an index is a run of digits;
//   * a `%f` field with at most 19 significant digits and a decimal exponent within +-22 is w * 10^k or w / 10^k with w and 10^k exact doubles: ONE
//     rounding, so the double d is the correctly rounded value of the decimal (Clinger's fast path).  (float)d is then the correctly rounded FLOAT of the
//     decimal unless d sits exactly on the midpoint of two floats (the decimal may lie a hair to either side of it, and the tie rule applies only if it
//     IS the midpoint): d and a midpoint are both doubles, so "the decimal and d on different sides of a midpoint" means d == midpoint.  Such fields, and
//     anything outside the fast path's range (long digit strings, large exponents, subnormal floats, inf / nan / hex), go to strtof as before.
#include <cerrno>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <thread>
#include <vector>
#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>

namespace {

struct Chunk { size_t begin = 0, end = 0; size_t nv = 0, nf = 0; int err = 0; size_t err_line_pos = 0; };

inline const char *line_end(const char *p, const char *e) { const void *q = p < e ? std::memchr(p, '\n', (size_t)(e - p)) : nullptr; return q ? (const char *)q : e; }

const double POW10[23] = {1e0, 1e1, 1e2, 1e3, 1e4, 1e5, 1e6, 1e7, 1e8, 1e9, 1e10, 1e11, 1e12, 1e13, 1e14, 1e15, 1e16, 1e17, 1e18, 1e19, 1e20, 1e21, 1e22};

// One `%f` field starting at p (white space already skipped), inside [p, e): the value strtof gives, *q behind the field; false: no number there.
inline bool parse_float(const char *p, const char *e, float &out, const char *&q)
{
    const char *s = p;
    bool neg = false;
    if (s < e && (*s == '-' || *s == '+')) { neg = *s == '-'; ++s; }
    unsigned long long w = 0; int digits = 0, exp10 = 0; bool any = false, slow = false;
    while (s < e && *s >= '0' && *s <= '9') { any = true; if (w || *s != '0') { if (digits < 19) { w = w * 10 + (unsigned)(*s - '0'); ++digits; } else { ++exp10; slow = true; } } ++s; }
    if (s < e && *s == '.') {
        ++s;
        while (s < e && *s >= '0' && *s <= '9') { any = true; if (w || *s != '0') { if (digits < 19) { w = w * 10 + (unsigned)(*s - '0'); ++digits; --exp10; } else slow = true; } else --exp10; ++s; }
    }
    if (!any) slow = true;                                  // inf, nan, hex floats, or nothing at all: strtof decides
    if (!slow && s < e && (*s == 'e' || *s == 'E')) {
        const char *t = s + 1; bool eneg = false;
        if (t < e && (*t == '-' || *t == '+')) { eneg = *t == '-'; ++t; }
        if (t < e && *t >= '0' && *t <= '9') {
            int x = 0;
            while (t < e && *t >= '0' && *t <= '9') { if (x < 10000) x = x * 10 + (*t - '0'); ++t; }
            exp10 += eneg ? -x : x; s = t;
        }
    }
    if (!slow && w == 0) { out = neg ? -0.0f : 0.0f; q = s; return true; }
    if (!slow && w < (1ull << 53) && exp10 >= -22 && exp10 <= 22) {
        double d = (double)w;                              // exact
        d = exp10 < 0 ? d / POW10[-exp10] : d * POW10[exp10];   // one rounding: the correctly rounded double of the decimal
        unsigned long long bits; std::memcpy(&bits, &d, 8);
        const int be = (int)((bits >> 52) & 0x7ff) - 1023;
        if (be >= -126 && be < 127 && (bits & 0x1fffffffull) != 0x10000000ull) {   // a normal float, and d is not the midpoint of two floats
            out = (float)(neg ? -d : d); q = s; return true;
        }
    }
    char *qq = nullptr;
    out = std::strtof(p, &qq);                              // correctly rounded, as %f.  p is at the field's first character (never white space), so strtof stops at the first character that is
                                                            // not part of a number -- at the latest the line's '\n': every line of the buffer ENDS in one (a mapped file because it ends in '\n', else the
                                                            // copy with one appended + NUL), so it never reads past the line, let alone the mapping
    if (qq == p) return false;
    q = qq;
    return true;
}
// sscanf(buffer, "v %f %f %f") == 3
inline bool parse_v(const char *p, const char *e, float out[3])
{
    ++p;                                                   // 'v'
    for (int k = 0; k < 3; ++k) {
        while (p < e && (*p == ' ' || *p == '\t' || *p == '\r')) ++p;
        if (p >= e) return false;
        const char *q = nullptr;
        if (!parse_float(p, e, out[k], q)) return false;
        p = q;
    }
    return true;
}
// one `%d`: optional sign + digits (what strtol(…, 10) accepts after white space)
inline bool parse_int(const char *p, const char *e, long &v, const char *&q)
{
    const char *s = p; bool neg = false;
    while (s < e && (*s == ' ' || *s == '\t' || *s == '\r')) ++s;    // %d / strtol skip white space in front of the number ("f 1/ 2 3/ 4 5/ 6" is six integers to sscanf): inside the line only
    if (s < e && (*s == '-' || *s == '+')) { neg = *s == '-'; ++s; }
    if (s >= e || *s < '0' || *s > '9') return false;
    unsigned long long w = 0;
    while (s < e && *s >= '0' && *s <= '9') { if (w < (1ull << 40)) w = w * 10 + (unsigned)(*s - '0'); ++s; }
    v = neg ? -(long)w : (long)w; q = s;
    return true;
}
// sscanf(buffer, "f %d/%d %d/%d %d/%d") == 6
inline bool parse_f(const char *p, const char *e, long v[3])
{
    ++p;                                                   // 'f'
    for (int k = 0; k < 3; ++k) {
        while (p < e && (*p == ' ' || *p == '\t' || *p == '\r')) ++p;
        const char *q = nullptr;
        if (!parse_int(p, e, v[k], q) || q >= e || *q != '/') return false;
        p = q + 1;
        long tex;
        if (!parse_int(p, e, tex, q)) return false;        // the texture index is parsed and dropped (load_obj.h:68 nv.vIdx)
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
    const int fd = ::open(path, O_RDONLY);
    if (fd < 0) return CD_ERR_IO;                          // load_obj.h:31-35 "file is not good"
    struct stat stt;
    if (::fstat(fd, &stt) != 0 || !S_ISREG(stt.st_mode)) { ::close(fd); return CD_ERR_IO; }
    const long sz = (long)stt.st_size;
    if (sz <= 0) { ::close(fd); return CD_ERR_FORMAT; }
    // The parsers never read past the end of a line, strtof (the slow path) stops at the white space or NUL behind a field: a file that ends in a newline is
    // parsed where the page cache holds it (a private, read-only mapping).  A last line WITHOUT a newline needs one appended: the copy of earlier rounds.
    std::vector<char> buf;
    const char *base = nullptr;
    void *map = ::mmap(nullptr, (size_t)sz, PROT_READ, MAP_PRIVATE | MAP_POPULATE, fd, 0);
    struct Unmap { void *p; size_t n; int fd; ~Unmap() { if (p && p != MAP_FAILED) ::munmap(p, n); ::close(fd); } } unmap{map, (size_t)sz, fd};
    if (map != MAP_FAILED && ((const char *)map)[sz - 1] == '\n') base = (const char *)map;
    else {
        buf.resize((size_t)sz + 2);
        size_t got = 0;
        while (got < (size_t)sz) { const ssize_t r = ::pread(fd, buf.data() + got, (size_t)sz - got, (off_t)got); if (r <= 0) break; got += (size_t)r; }
        if (got != (size_t)sz) return CD_ERR_IO;
        buf[(size_t)sz] = '\n';                            // a last line without a newline still ends in one ...
        buf[(size_t)sz + 1] = '\0';                        // ... and strtof, which skips newlines as white space, stops at the NUL
        base = buf.data();
    }
    const char *end = base + sz;

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
