"""Hardware counters of the GEMM kernels of one block apply, in any arithmetic: several rocprofv3 passes (a handful of SQ counters
each; counters serialise the kernels) of a child that runs `steps` block applies, summarised per kernel.
usage: python scripts/kernel_counters.py <accum> [b] [N] [P]        (child mode: ... child <accum> <b> <N> <P>)"""
import collections
import csv
import glob
import json
import os
import subprocess
import sys
import tempfile

if len(sys.argv) > 1 and sys.argv[1] == "child":
    sys.path.insert(0, os.getcwd())
    import flashpca_amd as fp

    accum, b, N, P = sys.argv[2], int(sys.argv[3]), int(sys.argv[4]), int(sys.argv[5])
    with fp.Context.synthetic(N, P, n_pop=40, accum=accum) as c:
        c.bench_apply(b=b, steps=3, warmup=1)
    sys.exit(0)

accum = sys.argv[1] if len(sys.argv) > 1 else "fp32"
b = int(sys.argv[2]) if len(sys.argv) > 2 else 16
N = int(sys.argv[3]) if len(sys.argv) > 3 else 500000
P = int(sys.argv[4]) if len(sys.argv) > 4 else 100000
PASSES = [
    ["SQ_VALU_MFMA_BUSY_CYCLES", "GRBM_GUI_ACTIVE", "SQ_INSTS_MFMA", "SQ_INSTS_VALU", "SQ_LDS_BANK_CONFLICT", "SQ_LDS_IDX_ACTIVE"],
    ["SQ_WAVES", "SQ_WAVE_CYCLES", "SQ_BUSY_CYCLES", "SQ_INSTS_LDS", "SQ_INSTS_SALU", "SQ_INSTS_VMEM_RD", "GRBM_GUI_ACTIVE"],
    ["SQ_WAIT_INST_ANY", "SQ_WAIT_ANY", "SQ_WAIT_INST_LDS", "SQ_ACTIVE_INST_LDS", "SQ_ACTIVE_INST_VALU", "SQ_ACTIVE_INST_ANY", "GRBM_GUI_ACTIVE"],
    ["SQ_INST_CYCLES_VMEM", "SQ_ACTIVE_INST_VMEM", "SQ_ACTIVE_INST_MISC", "SQ_ACTIVE_INST_SCA", "SQ_THREAD_CYCLES_VALU", "SQ_INSTS_WAVE32_LDS", "GRBM_GUI_ACTIVE"],
]
agg = collections.defaultdict(list)
dur = collections.defaultdict(list)
env = dict(os.environ, TMPDIR="/tmp")
for ctrs in PASSES:
    with tempfile.TemporaryDirectory(dir="/tmp") as tmp:
        cmd = ["rocprofv3", "--kernel-trace", "--pmc"] + ctrs + ["-d", tmp, "-o", "p", "--output-format", "csv", "--", sys.executable, os.path.abspath(__file__), "child",
               accum, str(b), str(N), str(P)]
        r = subprocess.run(cmd, capture_output=True, text=True, cwd="/tmp" if False else os.getcwd(), env=env)
        fs = glob.glob(tmp + "/**/*counter_collection.csv", recursive=True)
        if not fs:
            print("pass failed:", ctrs, r.stderr[-400:], flush=True)
            continue
        for row in csv.DictReader(open(fs[0])):
            n = row["Kernel_Name"]
            key = "K2 " + n[:44] if ("k_xt_b" in n) else "K3 " + n[:44] if "k_x_t" in n else ("GEMM " + n[:60]) if "k_gemm_i8" in n else None
            if key is None:
                continue
            agg[(key, row["Counter_Name"])].append(float(row["Counter_Value"]))
            if row["Counter_Name"] == "GRBM_GUI_ACTIVE" and row.get("End_Timestamp"):
                dur[key].append(int(row["End_Timestamp"]) - int(row["Start_Timestamp"]))
keys = sorted({k for k, _ in agg})
for k in keys:
    m = lambda c: (sum(agg[(k, c)]) / len(agg[(k, c)])) if agg.get((k, c)) else None
    g = m("GRBM_GUI_ACTIVE")
    cyc = g / 8 if g else None  # per-XCD cycles of the launch
    d = sum(dur[k]) / len(dur[k]) if dur[k] else None
    out = dict(kernel=k, launches=len(dur[k]) // max(1, len(PASSES)), ms=d / 1e6 if d else None, clock_ghz=cyc / d if cyc and d else None)
    simd = 1024.0
    for c, den, name in (("SQ_VALU_MFMA_BUSY_CYCLES", simd, "mfma_pipe_busy"), ("SQ_BUSY_CYCLES", 8.0 * 4, "sq_busy"),):
        if m(c) is not None and cyc:
            out[name] = m(c) / den / cyc
    im = m("SQ_INSTS_MFMA")
    for c in ("SQ_INSTS_VALU", "SQ_INSTS_LDS", "SQ_INSTS_SALU", "SQ_INSTS_VMEM_RD"):
        if im and m(c) is not None:
            out[c.lower() + "_per_mfma"] = m(c) / im
    if m("SQ_WAVE_CYCLES") and cyc:
        out["waves_per_simd_avg"] = m("SQ_WAVE_CYCLES") / simd / cyc * (4 if False else 1)
    for c in ("SQ_WAIT_INST_ANY", "SQ_WAIT_ANY", "SQ_WAIT_INST_LDS", "SQ_ACTIVE_INST_LDS", "SQ_ACTIVE_INST_VALU", "SQ_ACTIVE_INST_ANY", "SQ_INST_CYCLES_VMEM",
              "SQ_ACTIVE_INST_VMEM", "SQ_ACTIVE_INST_MISC", "SQ_ACTIVE_INST_SCA"):
        if m(c) is not None and m("SQ_WAVE_CYCLES" if False else "GRBM_GUI_ACTIVE"):
            out[c.lower() + "_per_simd_cycle"] = m(c) / simd / cyc
    if m("SQ_LDS_IDX_ACTIVE"):
        out["lds_bank_conflict_frac"] = m("SQ_LDS_BANK_CONFLICT") / m("SQ_LDS_IDX_ACTIVE")
        out["lds_idx_active_per_cu_cycle"] = m("SQ_LDS_IDX_ACTIVE") / 256.0 / cyc
    out["raw"] = {c: m(c) for (kk, c) in agg if kk == k}
    print(json.dumps(out), flush=True)
