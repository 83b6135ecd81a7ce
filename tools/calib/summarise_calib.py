"""Per calibration kernel: FETCH_SIZE (KiB as reported -> bytes) and 32 / 64 / 128-byte read requests against the useful bytes."""
import collections, csv, glob, os, sys
O = sys.argv[1]
useful = {"stream16": 1 << 30, "gather32": 1 << 30, "gather64": 1 << 30, "gather24": 24 << 25, "sload32": 32 << 22}
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(os.path.join(O, "**", "*counter_collection.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0]
        acc[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
print("kernel,useful_MB,FETCH_SIZE_MB,FETCH/useful,RDREQ,RDREQ_32B,RDREQ_64B,RDREQ_128B,sized_bytes_MB(32*n32+64*n64+128*n128),sized/useful,RDREQ_DRAM,TCC_HIT,TCC_MISS")
for k in ("stream16", "gather32", "gather64", "gather24", "sload32"):
    c = {n: sum(v) / len(v) for n, v in acc.get(k, {}).items()}
    if not c:
        continue
    u = useful[k]
    fetch = c.get("FETCH_SIZE", 0) * 1024
    n32, n64, n128 = c.get("TCC_EA0_RDREQ_32B_sum", 0), c.get("TCC_EA0_RDREQ_64B_sum", 0), c.get("TCC_EA0_RDREQ_128B_sum", 0)
    sized = 32 * n32 + 64 * n64 + 128 * n128
    print(f"{k},{u/1e6:.1f},{fetch/1e6:.1f},{fetch/u:.3f},{c.get('TCC_EA0_RDREQ_sum', 0):.0f},{n32:.0f},{n64:.0f},{n128:.0f},{sized/1e6:.1f},{sized/u:.3f},"
          f"{c.get('TCC_EA0_RDREQ_DRAM_sum', 0):.0f},{c.get('TCC_HIT_sum', 0):.0f},{c.get('TCC_MISS_sum', 0):.0f}")
