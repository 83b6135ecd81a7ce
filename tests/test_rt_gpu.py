"""GPU parity tests for the ray-tracing path: frames rendered by libmi355rt.so (through the C ABI) must
equal the CPU oracle's frames byte for byte (float math, 2 sqrtf + 1 divide per hit, strict t > maxz)."""
import os

import numpy as np
import pytest

import mi355_synth as synth
import mi355rt
import oracle

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(__file__), "golden")
MODES = [mi355rt.RT_MODE_BINNED, mi355rt.RT_MODE_BRUTE]


@pytest.mark.parametrize("mode", MODES)
@pytest.mark.parametrize("dim,n,seed", [(64, 1, 1), (256, 64, 2), (512, 500, 3), (1024, 700, 4)])
def test_frames_pixel_exact(mode, dim, n, seed):
    spheres, shifts = synth.sphere_scene(n, dim, seed)
    with mi355rt.RayTracer(spheres, dim) as rt:
        rt.set_mode(mode)
        img = rt.render(shifts)
        st = rt.stats()
    want = oracle.rt_render(spheres, shifts, dim)
    assert np.array_equal(img, want)
    assert (img[..., 3] == 255).all()
    assert img[..., :3].any()
    if mode == mi355rt.RT_MODE_BRUTE:
        assert st.sphere_tests == n * dim * dim
    else:
        assert 0 < st.sphere_tests <= n * dim * dim


@pytest.mark.parametrize("mode", MODES)
def test_shifts_and_camera_offsets(mode):
    """Per-sphere integer shifts (sphere.cuh:35-37) and the camera offset (anime_ray.cu:65-66)."""
    spheres, shifts = synth.sphere_scene(96, 256, seed=11)
    shifts[:, 0] = (np.arange(96) * 7) % 23 - 11
    shifts[:, 1] = (np.arange(96) * 5) % 19 - 9
    g = np.load(os.path.join(GOLD, "rt_golden.npz"))["img"]
    with mi355rt.RayTracer(spheres, 256) as rt:
        rt.set_mode(mode)
        img = rt.render(shifts, 3, -2)
        assert np.array_equal(img, g)                                    # committed fixture
        img2 = rt.render(shifts, -40, 77)
    assert np.array_equal(img2, oracle.rt_render(spheres, shifts, 256, -40, 77))


@pytest.mark.parametrize("mode", MODES)
def test_ties_big_spheres_and_offscreen(mode):
    dim = 256
    s = np.zeros(8, dtype=mi355rt.SPHERE_DTYPE)
    s["idx"] = np.arange(8)
    # 0,1: identical geometry, different colour -> strict '>' keeps the lower index (anime_ray.cu:75)
    for i, col in ((0, (1.0, 0.0, 0.0)), (1, (0.0, 1.0, 0.0))):
        s[i]["x"], s[i]["y"], s[i]["z"], s[i]["radius"] = 10.0, -20.0, 5.0, 30.0
        s[i]["r"], s[i]["g"], s[i]["b"] = col
    # 2: huge sphere covering many tiles, behind everything
    s[2]["x"], s[2]["y"], s[2]["z"], s[2]["radius"] = 0.0, 0.0, -500.0, 200.0
    s[2]["r"], s[2]["g"], s[2]["b"] = 0.3, 0.6, 0.9
    # 3: centred exactly on a tile corner; 4: half off-screen; 5: completely off-screen
    s[3]["x"], s[3]["y"], s[3]["z"], s[3]["radius"] = -64.0, 64.0, 50.0, 17.5
    s[4]["x"], s[4]["y"], s[4]["z"], s[4]["radius"] = 128.0, 0.0, 60.0, 25.0
    s[5]["x"], s[5]["y"], s[5]["z"], s[5]["radius"] = 1000.0, 1000.0, 60.0, 25.0
    # 6: sub-pixel radius sitting between pixel centres (never hit); 7: radius just covering one pixel centre
    s[6]["x"], s[6]["y"], s[6]["z"], s[6]["radius"] = 40.5, 40.5, 90.0, 0.4
    s[7]["x"], s[7]["y"], s[7]["z"], s[7]["radius"] = -100.0, -100.0, 90.0, 0.75
    for i in range(3, 8):
        s[i]["r"], s[i]["g"], s[i]["b"] = 0.9, 0.8, 0.7
    shifts = np.zeros((8, 4), dtype=np.int32)
    with mi355rt.RayTracer(s, dim) as rt:
        rt.set_mode(mode)
        img = rt.render(shifts)
    want = oracle.rt_render(s, shifts, dim)
    assert np.array_equal(img, want)
    # the tie really went to sphere 0 (red), at its centre pixel
    px = img[dim // 2 - 20, dim // 2 + 10]
    assert px[0] > 200 and px[1] == 0


@pytest.mark.parametrize("mode", MODES)
def test_row_slabs_compose_the_frame(mode):
    """Multi-GPU sharding unit: rows [y0,y1) rendered separately equal the full frame's rows."""
    spheres, shifts = synth.sphere_scene(300, 512, seed=9)
    with mi355rt.RayTracer(spheres, 512) as rt:
        rt.set_mode(mode)
        full = rt.render(shifts)
        top = rt.render(shifts, rows=(0, 192))
        bot = rt.render(shifts, rows=(192, 512))
    assert np.array_equal(np.concatenate([top, bot], axis=0), full)
    assert np.array_equal(full, oracle.rt_render(spheres, shifts, 512))


@pytest.mark.parametrize("mode", MODES)
def test_repeated_frames_equal_the_single_frame(mode):
    """rt_render_repeat: the same frame several times back to back (time stamps on the first and last kernel only) leaves
    the same pixels and the same test count as one rt_render; the per-frame time is reported."""
    spheres, shifts = synth.sphere_scene(200, 512, seed=31)
    with mi355rt.RayTracer(spheres, 512) as rt:
        rt.set_mode(mode)
        one = rt.render(shifts, 2, -1)
        t1 = rt.stats().sphere_tests
        many = rt.render_repeat(shifts, 5, 2, -1)
        st = rt.stats()
        assert np.array_equal(one, many) and st.sphere_tests == t1 and st.ms_render > 0
        again = rt.render(shifts, 2, -1)                                     # (the list counters of the next frame were left clean)
        assert np.array_equal(one, again)
    assert np.array_equal(one, oracle.rt_render(spheres, shifts, 512, 2, -1))


def test_config5_shape_binned_equals_brute_and_oracle_rows():
    """BASELINE config 5 (4096^2, 4096 spheres): binned == brute on the full frame (size-independent
    property), and both equal the oracle on a 64-row slab (the oracle needs ~1 s per 64 rows here)."""
    dim, n = 4096, 4096
    spheres, shifts = synth.sphere_scene(n, dim, seed=7)
    with mi355rt.RayTracer(spheres, dim) as rt:
        rt.set_mode(mi355rt.RT_MODE_BINNED)
        a = rt.render(shifts)
        sa = rt.stats()
        rt.set_mode(mi355rt.RT_MODE_BRUTE)
        b = rt.render(shifts)
    assert np.array_equal(a, b)
    assert sa.sphere_tests < n * dim * dim // 100
    y0 = 2048
    want = oracle.rt_render(spheres, shifts, dim, rows=(y0, y0 + 64))[y0:y0 + 64]
    assert np.array_equal(a[y0:y0 + 64], want)


def test_config5_whole_frame_equals_the_oracle():
    """BASELINE config 5, the WHOLE 4096 x 4096 frame against the CPU oracle (sphere.cuh:34-44, anime_ray.cu:61-87) -- every one of the 16.8 M pixels, in slabs of
    rows spread over the host's cores (the oracle's loop is the reference's: 4096 spheres per pixel; about a minute of one core)."""
    import concurrent.futures as cf
    dim, n = 4096, 4096
    spheres, shifts = synth.sphere_scene(n, dim, seed=7)
    with mi355rt.RayTracer(spheres, dim) as rt:
        got = rt.render(shifts)
    workers = max(1, min(16, len(os.sched_getaffinity(0))))
    slabs = [(y, y + 128) for y in range(0, dim, 128)]
    oracle.rt_render(spheres, shifts, dim, rows=(0, 1))                      # (the library is loaded before the threads start)
    with cf.ThreadPoolExecutor(workers) as ex:                              # ctypes releases the GIL inside the C call
        outs = list(ex.map(lambda r: oracle.rt_render(spheres, shifts, dim, rows=r)[r[0]:r[1]].copy(), slabs))
    for (y0, y1), want in zip(slabs, outs):
        assert np.array_equal(got[y0:y1], want), (y0, y1)


def test_animation_state_kernels_follow_the_oracle():
    """sphere.cuh:50-118 on the device: initSpheres, axis move, curve move, speed / direction / angle update.
    State (XORWOW words, shifts, angles) must equal the oracle's after every step of a generate_frame-like schedule
    (anime_ray.cu:115-125), and a frame rendered from the device-resident state equals one rendered from the
    read-back shifts."""
    n, dim = 500, 512                                                      # SPHERES = 500, sphere.cuh:22
    spheres, _ = synth.sphere_scene(n, dim, seed=11)
    ref = oracle.RtAnim(n)
    with mi355rt.RayTracer(spheres, dim) as rt:
        with pytest.raises(mi355rt.RtError):
            rt.render(None)                                                # no device state yet
        rt.anim_init()
        sh, ang, rng = rt.anim_state()
        assert np.array_equal(sh, ref.shifts) and np.array_equal(ang, ref.angles) and np.array_equal(rng, ref.rng)
        assert sh[:6, 2].tolist() == [5, 10, 15, 20, 25, 5] and sh[:4, 3].tolist() == [-1, 1, -1, 1]
        for frame in range(1, 41):
            if frame % 4 == 0:                                             # SPHERE_FRAME_PER_SHAKE 4, SPHERE_SHAKE_TYPE 1
                rt.anim_curve_move(); ref.curve_move()
                rt.anim_update_speed_angle(1, 18); ref.speed_angle(1, 18)
            if frame % 10 == 0:                                            # exercise the other shake type too
                rt.anim_axis_move(35); ref.axis_move(35)
            sh, ang, rng = rt.anim_state()
            assert np.array_equal(rng, ref.rng), frame
            assert np.array_equal(sh, ref.shifts), frame
            assert np.array_equal(ang.view(np.uint64), ref.angles.view(np.uint64)), frame
        assert len(np.unique(sh[:, 2])) > 5 and (np.abs(sh[:, :2]) > 0).any()      # the state really moved
        img_dev = rt.render(None, 3, -2)
        img_host = rt.render(sh, 3, -2)
        assert np.array_equal(img_dev, img_host)
        assert np.array_equal(img_dev, oracle.rt_render(spheres, sh, dim, 3, -2))


@pytest.mark.parametrize("shake", [2, 1, 0])
def test_animation_loop_in_one_launch_per_frame_equals_the_kernel_sequence(shake):
    """rt_anim_loop (round 5): `frames` rounds of generate_frame (anime_ray.cu:115-131) with ONE launch per frame -- frame f renders while the workgroups of the
    launch's first grid row move the spheres on to frame f + 1 and bin them into the other list set.  The state after the loop and the last frame must be what
    the kernel sequence leaves (rt_anim_* + rt_render per frame) and what the oracle's restatement gives; loops of 1, 2, 3 and 7 frames (every phase of the two
    list sets and the three counter sets), single renders in between, a brute-mode loop and a scene with a permuted idx column (both take the plain sequence)."""
    n, dim = 700, 512
    spheres, _ = synth.sphere_scene(n, dim, seed=13)
    ref = oracle.RtAnim(n)

    def ref_step():
        if shake == 1:
            ref.axis_move(35)
        elif shake == 2:
            ref.curve_move(); ref.speed_angle(3, 18)

    with mi355rt.RayTracer(spheres, dim) as rt, mi355rt.RayTracer(spheres, dim) as seq:
        rt.anim_init(); seq.anim_init()
        total = 0
        for frames in (1, 2, 3, 7, 1, 4):
            img = rt.anim_loop(frames, shake, 35, 3, 18, 2, -3)
            assert rt.stats().ms_render > 0
            for _ in range(frames):
                ref_step()
                if shake == 1:
                    seq.anim_axis_move(35)
                elif shake == 2:
                    seq.anim_curve_move(); seq.anim_update_speed_angle(3, 18)
                want = seq.render(None, 2, -3)
            total += frames
            sh, ang, rng = rt.anim_state()
            assert np.array_equal(rng, ref.rng) and np.array_equal(sh, ref.shifts) and np.array_equal(ang.view(np.uint64), ref.angles.view(np.uint64)), total
            assert np.array_equal(img, want), total
            assert np.array_equal(rt.render(None, 2, -3), want)              # a single frame after a loop: the counter sets were left clean
        assert np.array_equal(img, oracle.rt_render(spheres, sh, dim, 2, -3))
        assert shake == 0 or (np.abs(sh[:, :2]) > 0).any()
        rt.set_mode(mi355rt.RT_MODE_BRUTE)                                   # brute mode: the plain kernel sequence behind the same entry point
        img = rt.anim_loop(2, shake, 35, 3, 18, 2, -3)
        ref_step(); ref_step()
        sh, ang, rng = rt.anim_state()
        assert np.array_equal(sh, ref.shifts) and np.array_equal(img, oracle.rt_render(spheres, sh, dim, 2, -3))
    # idx != position (sphere.cuh:35 reads the shift row idx names): the loop falls back to the kernel sequence
    perm = np.random.default_rng(5).permutation(n).astype(np.int32)
    sp2 = spheres.copy(); sp2["idx"] = perm
    ref2 = oracle.RtAnim(n)
    with mi355rt.RayTracer(sp2, dim) as rt:
        rt.anim_init()
        img = rt.anim_loop(3, 2, 35, 3, 18, 0, 0)
        for _ in range(3):
            ref2.curve_move(); ref2.speed_angle(3, 18)
        sh, _, _ = rt.anim_state()
        assert np.array_equal(sh, ref2.shifts) and np.array_equal(img, oracle.rt_render(sp2, sh, dim, 0, 0))


def test_rt_main_harness_animates(tmp_path):
    """host/rt_main.cpp, the headless twin of anime_ray.cu's main / generate_frame: init, per-frame animation kernels,
    render from the device-resident shifts, the reference's timing line per frame, last frame as PPM."""
    import subprocess
    exe = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpu-computing-course_amd", "bin", "rt_main")
    assert os.path.exists(exe), "run __graft_entry__.build()"
    ppm = tmp_path / "f.ppm"
    out = subprocess.run([exe, "--dim", "256", "--spheres", "120", "--frames", "9", "--shake", "curve", "--ppm", str(ppm)],
                         capture_output=True, text=True, timeout=120)
    assert out.returncode == 0, out.stdout + out.stderr
    assert out.stdout.count("Time to generate a frame:") == 9
    data = ppm.read_bytes()
    assert data.startswith(b"P6\n256 256\n255\n") and len(data) == len(b"P6\n256 256\n255\n") + 256 * 256 * 3
    assert any(data[15:])                                                  # not a black frame


def test_random_scenes_property():
    """Seeded random scenes: image sizes 64..512, 1..700 spheres with radii from sub-pixel to larger than the image,
    positions on and far off the screen, duplicate spheres (ties in t), random sphere shifts (also through shuffled
    idx), camera offsets and row slabs -- both modes must reproduce the oracle's pixels."""
    rng = np.random.default_rng(77)
    for case in range(30):
        dim = int(rng.choice([64, 128, 192, 256, 512]))
        n = int(rng.choice([1, 2, 7, 64, 65, 300, 700]))
        spheres, shifts = synth.sphere_scene(n, dim, int(rng.integers(1 << 30)))
        spheres["radius"] = (rng.random(n) ** 3 * float(rng.choice([4.0, 40.0, 2.0 * dim])) + 0.25).astype(np.float32)
        spheres["x"] = (spheres["x"] * float(rng.choice([1.0, 3.0]))).astype(np.float32)      # some far off the screen
        if n > 2:
            spheres[n // 2] = spheres[0]; spheres["idx"][n // 2] = n // 2                        # an exact duplicate: tie in t
        spheres["idx"] = rng.permutation(n).astype(np.int32)                                     # shifts looked up through idx
        shifts[:, :2] = rng.integers(-40, 40, size=(n, 2))
        csx, csy = int(rng.integers(-30, 30)), int(rng.integers(-30, 30))
        want = oracle.rt_render(spheres, shifts, dim, csx, csy)
        with mi355rt.RayTracer(spheres, dim) as rt:
            for mode in MODES:
                rt.set_mode(mode)
                assert np.array_equal(rt.render(shifts, csx, csy), want), (case, mode)
            if dim >= 128:
                rt.set_mode(mi355rt.RT_MODE_BINNED)
                y0 = 64 * int(rng.integers(0, dim // 64 - 1)); y1 = y0 + 64
                assert np.array_equal(rt.render(shifts, csx, csy, rows=(y0, y1)), want[y0:y1]), case
