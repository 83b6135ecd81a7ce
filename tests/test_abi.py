"""C-ABI surface checks that run without a GPU: the shared libraries load, export every symbol that
include/*.h declares (and nothing the header does not declare), and fail loudly -- not fall back --
when there is no HIP device."""
import ctypes as C
import os
import re
import subprocess

import numpy as np
import pytest

import mi355cd
import mi355rt
from conftest import ROOT, has_gpu


def _declared(header, prefix):
    text = open(os.path.join(ROOT, "include", header)).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(%s_[a-z0-9_]+)\s*\(" % prefix, text)))


def _exported(lib, prefix):
    out = subprocess.run(["nm", "-D", "--defined-only", lib], check=True, capture_output=True, text=True).stdout
    return sorted(l.split()[-1] for l in out.splitlines() if re.search(r" T %s_" % prefix, l))


def test_cd_header_and_library_agree():
    decl = _declared("mi355cd.h", "cd")
    assert decl == sorted(mi355cd.EXPORTS)
    assert _exported(mi355cd.LIB_PATH, "cd") == decl
    lib = mi355cd.load_library()
    for name in decl:
        assert getattr(lib, name) is not None
    assert mi355cd.version().startswith("mi355cd") and "gfx950" in mi355cd.version()


def test_rt_header_and_library_agree():
    decl = _declared("mi355rt.h", "rt")
    assert decl == sorted(mi355rt.EXPORTS)
    assert _exported(mi355rt.LIB_PATH, "rt") == decl
    lib = mi355rt.load_library()
    for name in decl:
        assert getattr(lib, name) is not None
    assert "gfx950" in mi355rt.version()


def test_struct_layouts_match_header():
    assert C.sizeof(mi355cd.CdStats) == 6 * 4 + 2 * 4 + 5 * 8 + 2 * 4 + 2 * 4 + 4 + 4          # ... ms_pipeline, ms_build_block, tail padding to 8
    assert C.sizeof(mi355cd.CdMultiInfo) == 6 * 4 + 6 * 8 + 8 * 4
    assert mi355cd.QUERY_DTYPE.itemsize == 88
    assert mi355rt.SPHERE_DTYPE.itemsize == 32
    assert [n for n in mi355rt.SPHERE_DTYPE.names][:3] == ["r", "b", "g"]     # sphere.cuh:29 field order
    assert C.sizeof(mi355rt.RtStats) == 16


def test_libraries_contain_gfx950_code_objects_only():
    for lib in (mi355cd.LIB_PATH, mi355rt.LIB_PATH):
        blob = open(lib, "rb").read()
        assert b"gfx950" in blob
        assert b"gfx942" not in blob and b"sm_" not in blob


def test_argument_errors_do_not_need_a_device():
    lib = mi355cd.load_library()
    ctx = C.c_void_p()
    assert lib.cd_create(C.byref(ctx), None, 0, None, None, 0) == mi355cd.CD_ERR_ARG
    assert lib.cd_morton_sort(None) == mi355cd.CD_ERR_ARG
    assert lib.cd_find_collisions(None, None, 0, None) == mi355cd.CD_ERR_ARG
    assert lib.cd_morton3d_points(None, 4, None, None, None) == mi355cd.CD_ERR_ARG
    assert lib.cd_expand64_values(None, 4, None) == mi355cd.CD_ERR_ARG
    assert lib.cd_alloc_host_pairs(0, None) == mi355cd.CD_ERR_ARG
    assert lib.cd_box_pairs(None, None, 4, None, None) == mi355cd.CD_ERR_ARG
    assert lib.cd_tri_contact_points(None, 4, None) == mi355cd.CD_ERR_ARG
    assert lib.cd_debug_option(None, 0, 0, None) == mi355cd.CD_ERR_ARG
    assert lib.cd_multi_step(None, None, 0, None, None) == mi355cd.CD_ERR_ARG
    rl = mi355rt.load_library()
    rctx = C.c_void_p()
    assert rl.rt_create(C.byref(rctx), None, 0, 0) == mi355rt.RT_ERR_ARG
    s = np.zeros(1, dtype=mi355rt.SPHERE_DTYPE)
    assert rl.rt_create(C.byref(rctx), s.ctypes.data_as(C.c_void_p), 1, 100) == mi355rt.RT_ERR_ARG   # dim % 64
    sh, ang = mi355rt.init_shifts(7)                        # host-only helper (sphere.cuh:54-57)
    assert sh[:, 2].tolist() == [5, 10, 15, 20, 25, 5, 10] and sh[:, 3].tolist() == [-1, 1, -1, 1, -1, 1, -1]
    assert (sh[:, :2] == 0).all() and (ang == 0).all()


@pytest.mark.skipif(has_gpu(), reason="only meaningful on a box without a GPU")
def test_no_device_is_an_error_not_a_fallback():
    verts = np.zeros((3, 3)); vidx = np.array([[0, 1, 2]], dtype=np.uint32)
    with pytest.raises(mi355cd.CdError) as e:
        mi355cd.CollisionDetector(verts, vidx)
    assert e.value.rc == mi355cd.CD_ERR_NO_DEVICE
    s = np.zeros(1, dtype=mi355rt.SPHERE_DTYPE)
    with pytest.raises(mi355rt.RtError) as e:
        mi355rt.RayTracer(s, 64)
    assert e.value.rc == mi355rt.RT_ERR_NO_DEVICE
    # the context-free entry points too: morton3D / expand64Bits on the device, the pinned pair buffer
    for fn, arg in ((mi355cd.morton3d_points, np.zeros((4, 3))), (mi355cd.expand64_values, np.arange(4, dtype=np.uint64)), (mi355cd.HostPairs, 16),
                    (mi355cd.tri_contact_points, np.zeros((2, 6, 3))), (lambda a: mi355cd.box_pairs(a, a), np.zeros((2, 6)))):
        with pytest.raises(mi355cd.CdError) as e:
            fn(arg)
        assert e.value.rc == mi355cd.CD_ERR_NO_DEVICE


def test_product_never_references_the_oracle():
    """The shipped libraries / harnesses / bindings must not link, load or mention anything under oracle/."""
    pkg = os.path.join(ROOT, "gpu-computing-course_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".hip", ".h", ".cpp", ".py", "Makefile")):
                text = open(os.path.join(dirpath, f), errors="replace").read()
                assert "liboracle" not in text and "orc_" not in text and "import oracle" not in text, os.path.join(dirpath, f)
    for lib in (mi355cd.LIB_PATH, mi355rt.LIB_PATH):
        out = subprocess.run(["ldd", lib], capture_output=True, text=True).stdout
        assert "oracle" not in out
