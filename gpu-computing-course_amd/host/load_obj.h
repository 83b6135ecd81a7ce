// load_obj.h -- OBJ ingest for the collision harness, in the dialect the reference accepts
// (CollisionDetection/load_obj.h:41-103): `v x y z` read as float then widened to double, and
// `f a/ta b/tb c/tc` faces with 1-based indices.  Unlike the reference, the loader neither computes
// Morton codes nor sorts -- both moved to the GPU (cd_morton_sort) -- and it reports errors to the
// caller instead of calling exit().
#pragma once
#include <cstdint>
#include <cstdio>
#include <string>
#include <vector>

struct ObjMesh {
    std::vector<double> verts;      // nv x 3
    std::vector<uint32_t> vidx;     // nt x 3, 0-based
    float xmin = 1000, ymin = 1000, zmin = 1000;   // load_obj.h:39, printed at load_obj.h:122
};

// Returns 0 on success; on failure returns non-zero and fills `err`.
inline int load_obj(const std::string &path, ObjMesh &m, std::string &err)
{
    FILE *f = std::fopen(path.c_str(), "r");
    if (!f) { err = "* ERROR: loading obj:(" + path + ") file is not good"; return 1; }    // load_obj.h:31-35
    char buffer[256];
    while (std::fgets(buffer, 255, f)) {                                                   // load_obj.h:41 getline(buffer, 255)
        if (buffer[0] == 'v' && buffer[1] == ' ') {                                        // load_obj.h:48
            float f1, f2, f3;
            if (std::sscanf(buffer, "v %f %f %f", &f1, &f2, &f3) != 3) { err = "* ERROR: vertex not in wanted format in OBJLoader"; std::fclose(f); return 2; }
            m.verts.push_back((double)f1); m.verts.push_back((double)f2); m.verts.push_back((double)f3);   // load_obj.h:52
            if (f1 < m.xmin) m.xmin = f1;
            if (f2 < m.ymin) m.ymin = f2;
            if (f3 < m.zmin) m.zmin = f3;
        } else if (buffer[0] == 'f' && buffer[1] == ' ') {                                 // load_obj.h:64
            int v1, v2, v3, t1, t2, t3;
            const int nt = std::sscanf(buffer, "f %d/%d %d/%d %d/%d", &v1, &t1, &v2, &t2, &v3, &t3);   // load_obj.h:68
            if (nt != 6) {
                char msg[160];
                std::snprintf(msg, sizeof msg, "* ERROR: I don't know the format of that FaceMtl (while only read %d vertex of face)", nt);
                err = msg; std::fclose(f); return 3;
            }
            const int v_size = (int)(m.verts.size() / 3) + 1;                              // load_obj.h:76-79 (the reference only warns, then
            if (v1 >= v_size || v2 >= v_size || v3 >= v_size || v1 < 1 || v2 < 1 || v3 < 1) {   //  reads out of bounds; here it is an error)
                char msg[160];
                std::snprintf(msg, sizeof msg, "* ERROR: Vertex of face out of bound, v_size: %d, v_idx of face: %d,%d,%d", v_size, v1, v2, v3);
                err = msg; std::fclose(f); return 4;
            }
            m.vidx.push_back((uint32_t)(v1 - 1)); m.vidx.push_back((uint32_t)(v2 - 1)); m.vidx.push_back((uint32_t)(v3 - 1));   // load_obj.h:81-83
        }
    }
    std::fclose(f);
    if (m.vidx.empty() || m.verts.empty()) { err = "* ERROR: no faces or vertices in " + path; return 5; }
    return 0;
}
