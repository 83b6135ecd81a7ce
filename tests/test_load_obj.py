"""OBJ ingest through the C ABI (cd_load_obj; reference: load_obj.h:24-103).  Host code only:
runs without a GPU.  Parity target: the reference's parse semantics -- `%f` floats widened to double, 1-based
`f a/ta b/tb c/tc` faces, file order kept -- independent of the number of parser threads."""
import os
import time

import numpy as np
import pytest

import mi355_synth as synth
import mi355cd


def test_generated_obj_round_trip(tmp_path):
    text = synth.grids_obj_text(48)
    p = tmp_path / "grids.obj"
    p.write_text(text)
    want_v, want_f = synth.parse_obj_text(text)
    for threads in (1, 3, 8, 0):
        v, f = mi355cd.load_obj(str(p), threads)
        assert v.dtype == np.float64 and f.dtype == np.uint32
        assert np.array_equal(v.view(np.uint64), want_v.view(np.uint64))     # float-parsed, widened: exact bits
        assert np.array_equal(f, want_f)


def test_float_parsing_matches_percent_f(tmp_path):
    """`%f` reads a float: 0.1 must become float32(0.1) widened, not the double 0.1 (load_obj.h:38,50-52)."""
    lines = ["# comment", "vt 0.5 0.5", "v 0.1 -0.2 3.0000001", "v 1e-3 2.5E2 -7", "v   4   5\t6  ", "vn 0 0 1",
             "f 1/1 2/2 3/3", "g group", "f 3/9 2/8 1/7",
             "f 1/ 2 3/ 4 2/\t6"]                 # white space after the slash: six integers to sscanf "%d/%d" (load_obj.h:68), as to strtol
    p = tmp_path / "a.obj"
    p.write_text("\n".join(lines))                                           # no trailing newline on purpose
    v, f = mi355cd.load_obj(str(p), 1)
    want = np.array([[0.1, -0.2, 3.0000001], [1e-3, 2.5e2, -7], [4, 5, 6]], dtype=np.float32).astype(np.float64)
    assert np.array_equal(v, want)
    assert f.tolist() == [[0, 1, 2], [2, 1, 0], [0, 2, 1]]


def _round_to_float32(text):
    """The correctly rounded float of a decimal string (what `%f` / strtof gives), by exact rational arithmetic: round-half-even on the float grid."""
    from fractions import Fraction
    x = Fraction(text)
    if x == 0:
        return np.float32(-0.0 if text.strip().startswith("-") else 0.0)
    sign = -1 if x < 0 else 1
    x = abs(x)
    e = 0                                                                   # 2^e <= x < 2^(e+1)
    while Fraction(2) ** (e + 1) <= x: e += 1
    while Fraction(2) ** e > x: e -= 1
    q = max(e, -126) - 23                                                   # the float grid around x is spaced 2^q (subnormals: 2^-149)
    n = x / Fraction(2) ** q
    lo = n.numerator // n.denominator
    r = n - lo
    if r > Fraction(1, 2) or (r == Fraction(1, 2) and lo % 2 == 1): lo += 1
    return np.float32(sign * float(Fraction(lo) * Fraction(2) ** q))       # exact: at most 24 significant bits


def test_fast_float_path_rounds_like_strtof(tmp_path):
    """Round 5: `%f` fields go through a parser of the loader's own (Clinger's fast path: w * 10^k in ONE double rounding, then to float unless the double sits
    on the midpoint of two floats; everything else -> strtof).  Against exact rational rounding on: decimals that ARE float midpoints and their neighbours
    (the one place where double rounding would show), 1-25 significant digits, exponents across the fast path's edge, subnormal and huge floats, signed zeros,
    leading zeros / bare dots."""
    rng = np.random.default_rng(7)
    texts = []
    from fractions import Fraction
    for _ in range(300):                                                     # exact midpoints of adjacent floats, written out in full, and +-1 in the last digit
        f = np.float32(rng.uniform(0.001, 1000.0)); g = np.nextafter(f, np.float32(np.inf))
        mid = (Fraction(float(f)) + Fraction(float(g))) / 2
        from decimal import Decimal, getcontext
        getcontext().prec = 60
        d = Decimal(mid.numerator) / Decimal(mid.denominator)               # finite: the denominator is a power of two
        t = format(d, "f")
        texts += [t, t + "1", t[:-1] + str((int(t[-1]) + 9) % 10) if t[-1] != "0" else t + "0"]
    for _ in range(1500):
        digits = int(rng.integers(1, 26)); ex = int(rng.integers(-30, 31))
        m = "".join(str(int(c)) for c in rng.integers(0, 10, digits))
        pos = int(rng.integers(0, digits + 1))
        t = (m[:pos] or "0") + "." + m[pos:] if rng.random() < 0.8 else m
        if rng.random() < 0.5: t += ("e%+d" % ex) if rng.random() < 0.5 else ("E%d" % ex)
        texts.append(("-" if rng.random() < 0.3 else "") + t)
    texts += ["0", "-0", "0.0", "-0.000", ".5", "5.", "-.25", "+1.5", "000123.4500", "1e-45", "1.4e-45", "7e-46", "3.39e38", "1e22", "1e23", "9007199254740993",
              "0.1", "0.2", "0.3", "16777217", "16777217.0000000000000001", "1.00000005960464477539062500000", "1.000000059604644775390625", "1.0000000596046447753906249999"]
    texts = [t for t in texts if abs(float(t)) < 3.4e38]                    # (overflow to inf is strtof's business and not a coordinate)
    while len(texts) % 3: texts.append("1")
    p = tmp_path / "floats.obj"
    with open(p, "w") as fh:
        for k in range(0, len(texts), 3):
            fh.write("v %s %s %s\n" % (texts[k], texts[k + 1], texts[k + 2]))
        fh.write("f 1/1 2/2 3/3\n")
    v, f = mi355cd.load_obj(str(p), 2)
    want = np.array([_round_to_float32(t) for t in texts], dtype=np.float32).astype(np.float64).reshape(-1, 3)
    bad = np.nonzero(v.view(np.uint64) != want.view(np.uint64))
    assert bad[0].size == 0, [(texts[3 * i + j], v[i, j], want[i, j]) for i, j in zip(*bad)][:5]


def test_large_file_many_threads_keeps_order(tmp_path):
    verts, vidx = synth.cloth_pair(150)                                       # 45 602 vertices, 90 000 faces, ~4 MB of text
    p = tmp_path / "cloth.obj"
    with open(p, "w") as fh:
        for v in verts:
            fh.write("v %.9g %.9g %.9g\n" % (v[0], v[1], v[2]))
        for t in vidx:
            fh.write("f %d/1 %d/2 %d/3\n" % (t[0] + 1, t[1] + 1, t[2] + 1))
    t0 = time.perf_counter(); v1, f1 = mi355cd.load_obj(str(p), 1); t1 = time.perf_counter()
    v8, f8 = mi355cd.load_obj(str(p), 8); t8 = time.perf_counter()
    assert np.array_equal(v1, verts) and np.array_equal(f1, vidx)           # %.9g round-trips a float exactly
    assert np.array_equal(v8, v1) and np.array_equal(f8, f1)
    print(f"cd_load_obj {os.path.getsize(p) / 1e6:.1f} MB: 1 thread {1e3 * (t1 - t0):.1f} ms, 8 threads {1e3 * (t8 - t1):.1f} ms")


@pytest.mark.parametrize("body,code", [
    ("v 1 2\nf 1/1 1/1 1/1\n", mi355cd.CD_ERR_FORMAT),                  # load_obj.h:57-61 vertex not in wanted format
    ("v 1 2 3\nf 1 1 1\n", mi355cd.CD_ERR_FORMAT),                      # load_obj.h:69-74 only `a/ta` faces are known
    ("v 1 2 3\nf 1/1 2/1 1/1\n", mi355cd.CD_ERR_INDEX),                 # load_obj.h:76-79 vertex of face out of bound
    ("v 1 2 3\nf 0/1 1/1 1/1\n", mi355cd.CD_ERR_INDEX),
    ("f 1/1 1/1 1/1\nv 1 2 3\n", mi355cd.CD_ERR_INDEX),                 # face before its vertices (load_obj.h:86 NOTE)
    ("# nothing\n", mi355cd.CD_ERR_FORMAT),
])
def test_malformed_files_are_errors_not_exits(tmp_path, body, code):
    p = tmp_path / "bad.obj"
    p.write_text(body)
    with pytest.raises(mi355cd.CdError) as e:
        mi355cd.load_obj(str(p), 2)
    assert e.value.rc == code


def test_missing_file():
    with pytest.raises(mi355cd.CdError) as e:
        mi355cd.load_obj("/nonexistent/dir/x.obj")
    assert e.value.rc == mi355cd.CD_ERR_IO
