"""Turn gpurun_out/prof_final/ (tools/refresh_profiles.sh) into the committed summaries under profiles/<round>/ and
profiles/traffic.json: HBM bytes per launch of the dominant kernel AND of the whole step, corrected as
MI355X_MICROARCH.md prescribes (FETCH_SIZE doubled on gfx950, WRITE_SIZE taken as read; rocprofv3 reports KiB)."""
import collections, csv, json, os, shutil, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = os.path.join(ROOT, "gpurun_out", "prof_final")
rnd = sys.argv[1] if len(sys.argv) > 1 else "r02"
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
for sub in ("pmc_fetch", "pmc_write", "pmc_l2"):
    if not os.path.exists(os.path.join(src, sub, "run_counter_collection.csv")):
        continue
    for k, cs in per_kernel(os.path.join(src, sub, "run_counter_collection.csv")).items():
        for c, v in cs.items():
            hbm[k][c] = sum(v) / len(v); hbm[k]["launches_" + c] = len(v)
with open(os.path.join(dst, "pmc_hbm_per_kernel.csv"), "w") as f:
    f.write("kernel,FETCH_SIZE_raw_per_launch,FETCH_SIZE_x2_per_launch,WRITE_SIZE_per_launch,TCC_HIT_per_launch,TCC_MISS_per_launch,L2_hit_rate,launches\n")
    for k in sorted(hbm):
        fr = hbm[k].get("FETCH_SIZE", 0.0); wr = hbm[k].get("WRITE_SIZE", 0.0)
        hit = hbm[k].get("TCC_HIT_sum", 0.0); miss = hbm[k].get("TCC_MISS_sum", 0.0)
        rate = f"{hit / (hit + miss):.3f}" if hit + miss > 0 else ""
        f.write(f"\"{k}\",{fr * 1024:.0f},{2 * fr * 1024:.0f},{wr * 1024:.0f},{hit:.0f},{miss:.0f},{rate},{hbm[k].get('launches_FETCH_SIZE', 0)}\n")
sq = per_kernel(os.path.join(src, "pmc_sq", "run_counter_collection.csv"))
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
whole_f = whole_w = 0.0
per = {}
for k, v in hbm.items():
    if not k.startswith(("cd::", "void cd::")):
        continue
    n = v.get("launches_FETCH_SIZE", 0)
    if n == 0:
        continue
    scale = n / steps                                            # launches per step (k_os_pass runs twice)
    whole_f += v.get("FETCH_SIZE", 0.0) * 1024 * scale; whole_w += v.get("WRITE_SIZE", 0.0) * 1024 * scale
    per[k] = {"launches_per_step": scale, "hbm_bytes_per_step": (2 * v.get("FETCH_SIZE", 0.0) + v.get("WRITE_SIZE", 0.0)) * 1024 * scale}
l2 = (hbm[kd]["TCC_HIT_sum"] / (hbm[kd]["TCC_HIT_sum"] + hbm[kd]["TCC_MISS_sum"])) if hbm[kd].get("TCC_HIT_sum", 0) + hbm[kd].get("TCC_MISS_sum", 0) > 0 else None
json.dump({"workload": "cloth-vs-cloth 1M (bench.py default)", "triangles": line["config"]["triangles_per_gpu"], "kernel": kd,
           "fetch_size_bytes_raw": fetch, "fetch_size_bytes_corrected": 2 * fetch, "write_size_bytes": write,
           "traverse_hbm_bytes_per_launch": 2 * fetch + write, "l2_hit_rate": l2,
           "whole_path_hbm_bytes_per_step": 2 * whole_f + whole_w, "whole_path_per_kernel": per,
           "note": "rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE in separate passes (profiles/%s/pmc_hbm_per_kernel.csv); FETCH_SIZE "
                   "doubled per MI355X_MICROARCH.md (gfx950 tallies 128-B requests at 64 B for 16-B/lane loads); WRITE_SIZE taken as read; "
                   "whole path = sum over the step's collision kernels (memsets excluded)" % rnd},
          open(os.path.join(ROOT, "profiles", "traffic.json"), "w"), indent=1)
print(open(os.path.join(ROOT, "profiles", "traffic.json")).read())
