"""Kernel list of ONE training step from a rocprofv3 --kernel-trace CSV: python scripts/trace_step.py <trace_kernel_trace.csv> [kernel marker] [marker launches per step]
Prints span / busy / gaps of the last complete step (default: two launches of the marker kernel per step; the nine-term
geometry step has four: clean + noisy pass) and its kernels by name."""
import collections, csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
marker = sys.argv[2] if len(sys.argv) > 2 else "level_fwd_train_bf16c"
per = int(sys.argv[3]) if len(sys.argv) > 3 else 2
idx = [i for i, r in enumerate(rows) if marker in r["Kernel_Name"]]
i0, i1 = idx[-2 * per], idx[-per]
seg = rows[i0:i1]
t0 = int(seg[0]["Start_Timestamp"])
busy = gaps = 0
last = t0
agg = collections.OrderedDict()
for r in seg:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    n = r["Kernel_Name"][:110]
    a = agg.setdefault(n, [0, 0]); a[0] += 1; a[1] += e - s
    if s > last: gaps += s - last
    last = max(last, e); busy += e - s
print(f"step span {(int(rows[i1]['Start_Timestamp']) - t0) / 1e3:.1f} us, busy {busy / 1e3:.1f} us, gaps {gaps / 1e3:.1f} us, {len(seg)} kernels")
for n, (c, t) in sorted(agg.items(), key=lambda x: -x[1][1]): print(f"{t / 1e3:9.1f} us {c:4d}  {n}")
