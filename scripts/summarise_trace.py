"""Per (kernel, grid size) launch statistics from a rocprofv3 kernel trace:  python scripts/summarise_trace.py <dir> > out.csv
The --stats summary averages a kernel over ALL its launches; the driver's bench command also solves small problems with the same
kernel instances (the 50,000 x 20,000 CPU-baseline solve, the CLI leg), so the per-grid rows are the ones to compare with
bench.py's `ms_dominant_kernel`."""
import csv, glob, sys, collections

rows = collections.defaultdict(list)
for f in glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        name = r["Kernel_Name"]
        name = name[name.find("k_"):] if "k_" in name else name
        rows[(name.split("(")[0], r.get("Grid_Size", r.get("Grid_Size_X", "")), r.get("Workgroup_Size", r.get("Workgroup_Size_X", "")))].append(
            (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6)
out = csv.writer(sys.stdout)
out.writerow(["kernel", "grid_size", "workgroup_size", "launches", "avg_ms", "min_ms", "max_ms", "total_ms"])
for (k, g, w), d in sorted(rows.items(), key=lambda kv: -sum(kv[1]))[:40]:
    out.writerow([k, g, w, len(d), "%.4f" % (sum(d) / len(d)), "%.4f" % min(d), "%.4f" % max(d), "%.2f" % sum(d)])
