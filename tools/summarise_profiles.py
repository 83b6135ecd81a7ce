"""Turn gpurun_out/prof_final/ (tools/refresh_profiles.sh) into the committed summaries under profiles/<round>/ and
profiles/traffic.json: HBM bytes per launch of every kernel of the step AND of the whole step.
Reads: what MI355X_MICROARCH.md prescribes is FETCH_SIZE doubled on gfx950 (it tallies 128-byte requests at 64 bytes) and WRITE_SIZE
as read; the guide calibrated that on wide streaming reads only and says "calibrate other access patterns yourself".  Done
(tools/calib/, profiles/r03_experiments/fetch_size_calibration.csv): FETCH_SIZE is 64 B x TCC_EA0_RDREQ whatever the request's
size -- EVERY vector-memory miss, streamed or gathered, is a 128-byte request (half counted), a scalar-cache miss a 64-byte one
(counted right).  So the bytes are taken from the SIZE-RESOLVED request counters, 32 n32 + 64 n64 + 128 n128 (a pass of its own:
pmc_rdreq), which equals 2 x FETCH_SIZE for a kernel without scalar-load misses and lies between 1 x and 2 x otherwise; both the raw
FETCH_SIZE and the doubled value are kept beside it.  rocprofv3 reports FETCH_SIZE / WRITE_SIZE in KiB."""
import collections, csv, json, os, shutil, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = os.path.join(ROOT, "gpurun_out", "prof_final")
rnd = sys.argv[1] if len(sys.argv) > 1 else "r03"
dst = os.path.join(ROOT, "profiles", rnd)
os.makedirs(dst, exist_ok=True)
shutil.copy(os.path.join(src, "trace", "run_kernel_stats.csv"), os.path.join(dst, "bench_kernel_stats.csv"))
if os.path.exists(os.path.join(src, "trace", "run_domain_stats.csv")):
    shutil.copy(os.path.join(src, "trace", "run_domain_stats.csv"), os.path.join(dst, "bench_domain_stats.csv"))
for name in ("bench_plain.json", "bench_under_rocprof.json"):
    lines = [l for l in open(os.path.join(src, name)).read().splitlines() if l.startswith("{")]
    json.dump(json.loads(lines[-1]), open(os.path.join(dst, name.replace("bench_plain", "bench_line")), "w"), indent=1)


def kname(full):
    """Kernel name without its argument list: cut at the LAST top-level '(' so that '(anonymous namespace)::' and
    template arguments such as k_render<true> survive (cutting at the first '(' merged every ray-tracer kernel)."""
    depth = 0
    for i in range(len(full) - 1, -1, -1):
        ch = full[i]
        if ch == ')':
            depth += 1
        elif ch == '(':
            depth -= 1
            if depth == 0:
                return full[:i].strip()
    return full.strip()


def per_kernel(path):
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(path)):
        acc[kname(r["Kernel_Name"])][r["Counter_Name"]].append(float(r["Counter_Value"]))
    return acc


hbm = collections.defaultdict(dict)
for sub in ("pmc_fetch", "pmc_write", "pmc_l2", "pmc_rdreq", "rt_pmc_fetch", "rt_pmc_write", "rt_pmc_l2", "rt_pmc_rdreq"):     # (rt_*: the ray tracer's kernels, tools/rt_time.py)
    if not os.path.exists(os.path.join(src, sub, "run_counter_collection.csv")):
        continue
    for k, cs in per_kernel(os.path.join(src, sub, "run_counter_collection.csv")).items():
        for c, v in cs.items():
            hbm[k][c] = sum(v) / len(v); hbm[k]["launches_" + c] = len(v)
with open(os.path.join(dst, "pmc_hbm_per_kernel.csv"), "w") as f:
    f.write("kernel,FETCH_SIZE_raw_per_launch,FETCH_SIZE_x2_per_launch,read_bytes_sized_per_launch(32*n32+64*n64+128*n128),RDREQ_32B,RDREQ_64B,RDREQ_128B,"
            "WRITE_SIZE_per_launch,TCC_HIT_per_launch,TCC_MISS_per_launch,L2_hit_rate,launches\n")
    for k in sorted(hbm):
        fr = hbm[k].get("FETCH_SIZE", 0.0); wr = hbm[k].get("WRITE_SIZE", 0.0)
        hit = hbm[k].get("TCC_HIT_sum", 0.0); miss = hbm[k].get("TCC_MISS_sum", 0.0)
        rate = f"{hit / (hit + miss):.3f}" if hit + miss > 0 else ""
        n32, n64, n128 = hbm[k].get("TCC_EA0_RDREQ_32B_sum", 0.0), hbm[k].get("TCC_EA0_RDREQ_64B_sum", 0.0), hbm[k].get("TCC_EA0_RDREQ_128B_sum", 0.0)
        hbm[k]["read_sized"] = 32 * n32 + 64 * n64 + 128 * n128 if (n32 + n64 + n128) > 0 else None
        sized = f"{hbm[k]['read_sized']:.0f}" if hbm[k]["read_sized"] is not None else ""
        f.write(f"\"{k}\",{fr * 1024:.0f},{2 * fr * 1024:.0f},{sized},{n32:.0f},{n64:.0f},{n128:.0f},{wr * 1024:.0f},{hit:.0f},{miss:.0f},{rate},{hbm[k].get('launches_FETCH_SIZE', 0)}\n")
sq = per_kernel(os.path.join(src, "pmc_sq", "run_counter_collection.csv"))
if os.path.exists(os.path.join(src, "rt_pmc_sq", "run_counter_collection.csv")):
    for k, cs in per_kernel(os.path.join(src, "rt_pmc_sq", "run_counter_collection.csv")).items():
        for c, v in cs.items():
            sq[k].setdefault(c, v)
# round 5: the 8 M step's tables and the ray tracer's kernel stats, as the refresh script left them
for name in ("8M_kernel_stats.csv", "8M_pmc_per_kernel.csv", "rt_time_plain.log"):
    if os.path.exists(os.path.join(src, name)):
        shutil.copy(os.path.join(src, name), os.path.join(dst, name))
if os.path.exists(os.path.join(src, "rt_trace", "run_kernel_stats.csv")):
    shutil.copy(os.path.join(src, "rt_trace", "run_kernel_stats.csv"), os.path.join(dst, "rt_kernel_stats.csv"))
if os.path.exists(os.path.join(src, "pmc_sq2", "run_counter_collection.csv")):      # scalar side + issue / wait split, a second pass (8 SQ slots per pass)
    for k, cs in per_kernel(os.path.join(src, "pmc_sq2", "run_counter_collection.csv")).items():
        for c, v in cs.items():
            sq[k].setdefault(c, v)
names = sorted({c for cs in sq.values() for c in cs})
with open(os.path.join(dst, "pmc_sq_per_kernel.csv"), "w") as f:
    f.write("kernel," + ",".join(names) + "\n")
    for k in sorted(sq):
        f.write("\"" + k + "\"," + ",".join(f"{sum(sq[k][c]) / len(sq[k][c]):.1f}" if c in sq[k] else "" for c in names) + "\n")

line = json.load(open(os.path.join(dst, "bench_line.json")))
dom = "k_descend_half"          # the top-level fields describe the descent; every kernel has its row in whole_path_per_kernel
kd = next(k for k in hbm if dom in k)
fetch = hbm[kd]["FETCH_SIZE"] * 1024; write = hbm[kd]["WRITE_SIZE"] * 1024
# whole step: every collision kernel (namespace cd) launched once per step -- the ray tracer's kernels are another path
steps = hbm[kd].get("launches_FETCH_SIZE", 1)
whole_f = whole_w = whole_sized = 0.0
per = {}
for k, v in hbm.items():
    if not k.startswith(("cd::", "void cd::")):
        continue
    n = v.get("launches_FETCH_SIZE", 0)
    if n == 0:
        continue
    scale = n / steps                                            # launches per step (k_os_pass runs twice)
    whole_f += v.get("FETCH_SIZE", 0.0) * 1024 * scale; whole_w += v.get("WRITE_SIZE", 0.0) * 1024 * scale
    rd = v.get("read_sized") if v.get("read_sized") is not None else 2 * v.get("FETCH_SIZE", 0.0) * 1024        # (no size-resolved pass: the guide's doubling)
    whole_sized += rd * scale
    per[k] = {"launches_per_step": scale,
              "hbm_bytes_per_step": (rd + v.get("WRITE_SIZE", 0.0) * 1024) * scale,                             # reads by request size + WRITE_SIZE
              "hbm_bytes_per_step_raw": (v.get("FETCH_SIZE", 0.0) + v.get("WRITE_SIZE", 0.0)) * 1024 * scale,     # FETCH_SIZE as reported + WRITE_SIZE
              "read_bytes": rd * scale, "read_bytes_fetch_size_raw": v.get("FETCH_SIZE", 0.0) * 1024 * scale, "write_bytes": v.get("WRITE_SIZE", 0.0) * 1024 * scale,
              "reads_are": "32*n32 + 64*n64 + 128*n128 (TCC_EA0_RDREQ_*)" if v.get("read_sized") is not None else "2 x FETCH_SIZE"}
l2 = (hbm[kd]["TCC_HIT_sum"] / (hbm[kd]["TCC_HIT_sum"] + hbm[kd]["TCC_MISS_sum"])) if hbm[kd].get("TCC_HIT_sum", 0) + hbm[kd].get("TCC_MISS_sum", 0) > 0 else None
kd_read = hbm[kd].get("read_sized") if hbm[kd].get("read_sized") is not None else 2 * fetch
json.dump({"workload": "cloth-vs-cloth 1M (bench.py default)", "triangles": line["config"]["triangles_per_gpu"], "kernel": kd,
           "fetch_size_bytes_raw": fetch, "fetch_size_bytes_x2": 2 * fetch, "read_bytes_by_request_size": kd_read, "write_size_bytes": write,
           "traverse_hbm_bytes_per_launch": kd_read + write, "traverse_hbm_bytes_per_launch_raw": fetch + write, "l2_hit_rate": l2,
           "whole_path_hbm_bytes_per_step": whole_sized + whole_w, "whole_path_hbm_bytes_per_step_raw": whole_f + whole_w, "whole_path_per_kernel": per,
           "note": "rocprofv3 --pmc passes, one counter group each (profiles/%s/pmc_hbm_per_kernel.csv).  Reads = 32 n32 + 64 n64 + 128 n128 from the "
                   "size-resolved TCC_EA0_RDREQ counters: FETCH_SIZE counts every request at 64 B, and every vector-memory miss on gfx950 is a 128-byte "
                   "request (calibrated on streaming AND gather kernels: profiles/r03_experiments/fetch_size_calibration.csv), so this equals the guide's "
                   "2 x FETCH_SIZE except for the scalar-cache misses (64-byte requests).  *_raw = FETCH_SIZE as reported.  WRITE_SIZE taken as read; "
                   "whole path = sum over the step's collision kernels (memsets excluded)" % rnd},
          open(os.path.join(ROOT, "profiles", "traffic.json"), "w"), indent=1)
print(open(os.path.join(ROOT, "profiles", "traffic.json")).read())
