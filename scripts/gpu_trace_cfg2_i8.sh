#!/bin/bash
mkdir -p gpurun_out; export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d gpurun_out/trace_cfg2_i8 -o t -- python bench.py --accum i8 --steps 20 --warmup 3 --no-cpu-baseline --no-pca --no-alt > /dev/null 2>&1
python - <<'PY'
import csv, glob, collections
f = glob.glob("gpurun_out/trace_cfg2_i8/*kernel_trace.csv")[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
# find the last 20 steps: sequences starting at k_colmax ... take the last 300 kernels
rows = rows[-300:]
names = collections.OrderedDict()
prev_end = None
gap_tot = 0
for r in rows:
    n = r["Kernel_Name"].split("(")[0][-60:]
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    d = names.setdefault(n, [0, 0])
    d[0] += 1; d[1] += e - s
    if prev_end is not None: gap_tot += max(0, s - prev_end)
    prev_end = e
span = int(rows[-1]["End_Timestamp"]) - int(rows[0]["Start_Timestamp"])
print("span %.3f ms for %d kernels; gaps total %.3f ms" % (span / 1e6, len(rows), gap_tot / 1e6))
for n, (c, t) in names.items():
    print("%-62s calls %3d avg %.1f us total %.3f ms" % (n, c, t / c / 1e3, t / 1e6))
PY
