"""Do the per-pivot decision kernels run BESIDE the pass?  Reads a rocprofv3 --kernel-trace CSV (kernel_trace.csv) and
prints, per kernel family, count / avg duration, the share of decision-kernel time that falls inside a pass's interval,
and a timeline of one steady-state block.

    python tools/trace_overlap.py gpurun_out/<dir>           (searches for *_kernel_trace.csv below it)
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
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"][:60], r.get("Queue_Id", "?")))
rows.sort()
fam = {}
for s, e, k, q in rows:
    d = fam.setdefault(k, [0, 0, set()])
    d[0] += 1
    d[1] += e - s
    d[2].add(q)
for k, (n, t, q) in sorted(fam.items(), key=lambda kv: -kv[1][1])[:12]:
    print(f"{n:7d} x {t / n / 1e3:9.2f} us  queues {sorted(q)}  {k}")
passes = [(s, e) for s, e, k, q in rows if "k_tb_flush2" in k and e - s > 200000]
dec = [(s, e, k) for s, e, k, q in rows if "k_tb_column" in k or "k_tb_prepare" in k]
inside = total = 0
pi = 0
for s, e, k in dec:
    total += e - s
    while pi < len(passes) and passes[pi][1] < s:
        pi += 1
    if pi < len(passes) and passes[pi][0] <= s and e <= passes[pi][1]:
        inside += e - s
# gaps between consecutive decision kernels (end -> next start), steady state
gaps = [dec[k + 1][0] - dec[k][1] for k in range(len(dec) - 1)]
gaps = sorted(g for g in gaps if 0 <= g < 50000)
if gaps:
    print(f"gap between consecutive decision kernels: median {gaps[len(gaps) // 2] / 1e3:.2f} us, mean {sum(gaps) / len(gaps) / 1e3:.2f} us, p90 {gaps[int(0.9 * len(gaps))] / 1e3:.2f} us")
print(f"passes {len(passes)}, decision launches {len(dec)}, decision time inside a pass: {100.0 * inside / max(1, total):.1f} %")
if len(passes) > 6:
    s0, e0 = passes[len(passes) // 2]
    s1, e1 = passes[len(passes) // 2 + 1]
    print(f"one block: pass {(e0 - s0) / 1e3:.0f} us, next pass starts {(s1 - s0) / 1e3:.0f} us after this one's start")
    n = 0
    for s, e, k in dec:
        if s0 - 100000 <= s <= s1 and n < 70:
            print(f"   {k[:28]:28s} start {(s - s0) / 1e3:9.1f} us  dur {(e - s) / 1e3:7.1f} us")
            n += 1
