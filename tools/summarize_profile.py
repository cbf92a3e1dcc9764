"""Turn gpurun_out/prof_<tag>/ (written by tools/profile_bench.sh on the GPU box) into committed summaries:

    profiles/<tag>_kernel_stats.csv     rocprofv3 --kernel-trace --stats summary of `python bench.py`
    profiles/<tag>_pmc_hbm.csv          per-kernel averages of FETCH_SIZE / WRITE_SIZE (separate passes)
    profiles/k1_traffic.json            HBM bytes per K1 launch, corrected as MI355X_MICROARCH.md prescribes
                                        (FETCH_SIZE reads 1/2 of a wide coalesced stream on gfx950 -> x2;
                                        WRITE_SIZE exact; both in KiB)
"""
import collections
import csv
import glob
import json
import os
import sys

tag = sys.argv[1] if len(sys.argv) > 1 else "r02"
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = os.path.join(root, "gpurun_out", f"prof_{tag}")
dst = os.path.join(root, "profiles")
os.makedirs(dst, exist_ok=True)

stats = glob.glob(os.path.join(src, "trace", "*", "*_kernel_stats.csv"))[0]
rows = list(csv.reader(open(stats)))
with open(os.path.join(dst, f"{tag}_kernel_stats.csv"), "w", newline="") as f:
    w = csv.writer(f)
    for r in rows:
        r[0] = r[0][:120]
        w.writerow(r)

agg = collections.defaultdict(lambda: collections.defaultdict(list))
for name, sub in (("FETCH_SIZE", "pmc_fetch"), ("WRITE_SIZE", "pmc_write")):
    for f in glob.glob(os.path.join(src, sub, "*", "*_counter_collection.csv")):
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] == name:
                agg[r["Kernel_Name"][:120]][name].append(float(r["Counter_Value"]))
with open(os.path.join(dst, f"{tag}_pmc_hbm.csv"), "w", newline="") as f:
    w = csv.writer(f)
    w.writerow(["kernel", "dispatches", "FETCH_SIZE_KiB_avg", "WRITE_SIZE_KiB_avg", "hbm_bytes_corrected_avg"])
    k1 = tb = None
    for k, v in agg.items():
        fe = sum(v["FETCH_SIZE"]) / max(1, len(v["FETCH_SIZE"]))
        wr = sum(v["WRITE_SIZE"]) / max(1, len(v["WRITE_SIZE"]))
        tot = (2.0 * fe + wr) * 1024.0
        w.writerow([k, len(v["FETCH_SIZE"]), f"{fe:.1f}", f"{wr:.1f}", f"{tot:.0f}"])
        if "k_rank1" in k:
            k1 = dict(kernel=k, fetch_size_kib=fe, write_size_kib=wr, hbm_bytes_per_launch=tot,
                      correction="2*FETCH_SIZE + WRITE_SIZE, KiB -> bytes (MI355X_MICROARCH.md, HBM section)", tag=tag)
        if "k_tb_flush2" in k and (tb is None or tot > tb["hbm_bytes_per_launch"]):  # the instance that did the work (its twin leaves at once)
            tb = dict(kernel=k, fetch_size_kib=fe, write_size_kib=wr, hbm_bytes_per_launch=tot,
                      correction="2*FETCH_SIZE + WRITE_SIZE, KiB -> bytes (MI355X_MICROARCH.md, HBM section)", tag=tag)
if k1:
    json.dump(k1, open(os.path.join(dst, "k1_traffic.json"), "w"), indent=1)
    print(k1)
if tb:  # the pass of the deferred-update chain (one launch per 32 pivots)
    json.dump(tb, open(os.path.join(dst, "tb_flush_traffic.json"), "w"), indent=1)
    print(tb)

# config 2's trace (eager launches: the single-launch chain shows as one k_t_fused per pivot) and the bench lines
import shutil

t1k = glob.glob(os.path.join(src, "trace_tableau1k", "*", "*_kernel_stats.csv"))
if t1k:
    rows = list(csv.reader(open(t1k[0])))
    with open(os.path.join(dst, f"{tag}_tableau1k_kernel_stats.csv"), "w", newline="") as f:
        w = csv.writer(f)
        for r in rows:
            r[0] = r[0][:120]
            w.writerow(r)
for name in ("bench_default", "bench_tableau1k", "bench_batch", "bench_config5"):
    p = os.path.join(src, name + ".json")
    if os.path.exists(p):
        shutil.copy(p, os.path.join(dst, f"{tag}_{name}.json"))
