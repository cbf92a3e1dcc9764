"""Where does a short timed window of the pipelined chain spend its time?  Reads a rocprofv3 --kernel-trace CSV of
`bench.py --headline-only --steps K --warmup W` and prints, for every burst of passes (warm-up call, timed call, ...),
the first decision kernel's start, every pass's start / duration / distance to the previous pass, and the burst's span.

    python tools/window_timeline.py gpurun_out/<dir>
"""
import csv
import glob
import os
import sys

root = sys.argv[1]
files = glob.glob(os.path.join(root, "**", "*kernel_trace.csv"), recursive=True)
if not files:
    sys.exit("no kernel_trace.csv under " + root)
rows = []
for f in files:
    for r in csv.DictReader(open(f)):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"][:48]))
rows.sort()
tb = [(s, e, k) for s, e, k in rows if "k_tb_" in k or "k_t_price" in k or "k_tb_syncback" in k]
# bursts: a gap of more than 3 ms between consecutive lpx kernels separates two calls
bursts, cur = [], []
for s, e, k in tb:
    if cur and s - cur[-1][1] > 3_000_000:
        bursts.append(cur)
        cur = []
    cur.append((s, e, k))
if cur:
    bursts.append(cur)
for bi, b in enumerate(bursts):
    t0 = b[0][0]
    tend = max(e for s, e, k in b)
    passes = [(s, e, k) for s, e, k in b if "k_tb_flush2" in k and e - s > 200_000]
    dec = [(s, e, k) for s, e, k in b if "k_tb_column" in k or "k_tb_prepare" in k]
    print(f"burst {bi}: {len(b)} kernels, {len(passes)} passes, {len(dec)} decision launches, span {(tend - t0) / 1e6:.3f} ms")
    if len(passes) > 60:
        print("   (long burst: first and last 3 passes only)")
    prev = None
    for pi, (s, e, k) in enumerate(passes):
        if len(passes) > 60 and 3 <= pi < len(passes) - 3:
            prev = s
            continue
        d_in = [x for x in dec if s <= x[0] < e]
        print(f"   pass {pi:3d} start {(s - t0) / 1e3:9.1f} us  dur {(e - s) / 1e3:8.1f} us  since previous start "
              f"{'-' if prev is None else f'{(s - prev) / 1e3:8.1f}'}  decision launches inside {len(d_in)}  {k[12:40]}")
        prev = s
    if dec:
        print(f"   first decision kernel at {(dec[0][0] - t0) / 1e3:.1f} us, last ends {(dec[-1][1] - t0) / 1e3:.1f} us; "
              f"first 4: " + ", ".join(f"{(e - s) / 1e3:.1f}" for s, e, k in dec[:4]) + " us")
    others = [(s, e, k) for s, e, k in b if (s, e, k) not in passes and (s, e, k) not in dec]
    for s, e, k in others[:12]:
        print(f"   other: {k:48s} start {(s - t0) / 1e3:9.1f} us dur {(e - s) / 1e3:8.1f} us")
