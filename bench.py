#!/usr/bin/env python3
"""bench.py -- headline benchmark of the flashpca PCA hot path on MI355X.

  python bench.py --gpus N --steps K --warmup W          (N=1 directly; N>1 under torch.distributed.run)

A "step" is ONE pass of the hot path over the resident genotype matrix: the block operator
Y = sum_g X_g X_g' B on b columns, i.e. K2 (xt_b) + K3 (x_t) + the all-reduce of the N x b product when N > 1.  That is b
single-vector applications of the reference's perform_op (svdwide.cpp:21-68), so
value = N_samples * P_total * b * K / time   [genotype cells / s], the metric BASELINE.json names ("N x P x iters" with iters =
single-vector operator applications).  b IS THE WIDTH THE SOLVER RUNS for this k (`--blockvec 0`, the default: 16 columns for
k <= 64, flashpca_amd/csrc/pca_driver.cpp choose_blockvec) -- so `value` is the throughput of the pass `flashpca --ndim k`
actually makes (round 2 quoted the 32-column pass, whose 7 x 32 slice-columns fill the int8 GEMM's column tiles exactly and
which is 20 % faster per cell; that figure is the `apply_at_b32` side block now).  `pca` is the wall-clock of fpca_pca with its
default options -- the end-to-end figure: passes, time to the converged k, cells/s over the whole solve.

Workload (default, BASELINE.json configs[2]/[3] -- the configuration the metric is quoted on): synthetic 500,000 samples x
100,000 SNPs, k = 20, generated directly in HBM; with N > 1 ranks the SNP columns are sharded N ways (strong scaling,
rank r holds SNPs [P r / N, P (r+1) / N)) and the N x b product is all-reduced over RCCL.  `--workload cfg2` is the
50,000 x 20,000 matrix of configs[1] (per GPU: weak scaling), `cfg5` the 1M x 200k / k = 50 matrix of configs[4];
`cfg4shard` / `cfg5shard` are the one-GPU shards of configs[3] / configs[4] at 8 GPUs (12,500 / 25,000 SNPs of the same
sample counts): the per-GPU compute of the 8-GPU runs, measurable on one GPU.

Arithmetic: `--accum i8` (default; the product's default FPCA_ACCUM_AUTO resolves to it) runs the two GEMMs on the int8 matrix
cores -- integer genotype matrices x byte slices of the fp64 operand, exact int32 accumulation, fp64 recombination
-- with results equal to the fp64 MFMA kernels (`--accum fp64`) to ~3e-15; the line carries the other mode's timing and the
measured difference between the two as `fp64_mode` / `exact_int8_mode`.

The same JSON line also carries: the roofline of the dominant kernel (HIP events recorded live on the
kernels' stream inside the timed region), a bounded CPU baseline (the oracle = restated reference path, one
thread, like the shipped reference), and the wall-clock of one full k=20 PCA solve to convergence.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

FP64_MFMA_PEAK_TFLOPS = 78.6  # MI355X FP64 matrix peak (vendor datasheet); HBM3E 8 TB/s
FP32_MFMA_PEAK_TFLOPS = 157.3  # MI355X FP32 matrix peak (MI355X_MICROARCH.md)
I8_MFMA_PEAK_TOPS = 5033.0  # v_mfma_i32_32x32x32_i8 issue rate: 256 CUs x 4 SIMDs x 65536 op / 32 clk x 2.4 GHz (= 2x bf16 dense)
HBM_PEAK_GBS = 8000.0

WORKLOADS = {
    # name: (N samples, SNPs, k, block width, sharding)
    "cfg2": dict(N=50000, P=20000, k=20, scaling="weak",
                 desc="synthetic .bed-layout matrix, 50000 samples x 20000 SNPs per GPU, k=20 (BASELINE configs[1])"),
    "cfg3": dict(N=500000, P=100000, k=20, scaling="strong",
                 desc="synthetic 500000 samples x 100000 SNPs total, SNP-sharded across GPUs, k=20 (BASELINE configs[2]/[3])"),
    "cfg5": dict(N=1000000, P=200000, k=50, scaling="strong",
                 desc="synthetic 1000000 samples x 200000 SNPs total, SNP-sharded across GPUs, k=50 (BASELINE configs[4]; use --accum fp32)"),
    "cfg4shard": dict(N=500000, P=12500, k=20, scaling="weak", shards=8,
                      desc="one GPU's shard of BASELINE configs[3]: 500000 samples x 12500 of 100000 SNPs (1/8), k=20"),
    "cfg5shard": dict(N=1000000, P=25000, k=50, scaling="weak", shards=8,
                      desc="one GPU's shard of BASELINE configs[4]: 1000000 samples x 25000 of 200000 SNPs (1/8), k=50 (use --accum fp32)"),
    "tiny": dict(N=4000, P=3000, k=20, scaling="weak", desc="smoke-size workload"),
}
STEPS = {"cfg2": 400, "tiny": 400, "cfg3": 80, "cfg5": 24, "cfg4shard": 200, "cfg5shard": 60}


MISSING_PATH = {0: "dense (MFMA)", 1: "dense, empty blocks skipped", 2: "none missing", 3: "sparse fp64 gathers",
                4: "hybrid: sparse fp64 gathers + a compacted dense sub-matrix (MFMA) for the SNPs whose calls cost more to gather (above ~0.7 % missing at 7 slices; cost model over K1's counts)"}


def solver_blockvec(k):
    """fpca::choose_blockvec (pca_driver.cpp): the width fpca_pca picks when the caller leaves it open."""
    return 16 if k <= 64 else 32 if k <= 128 else 64
WARMUP = {"cfg2": 20, "tiny": 20, "cfg3": 5, "cfg5": 2, "cfg4shard": 10, "cfg5shard": 4}
# untimed clock spin-up applies before the caller's warm-up (reported in the JSON line as spinup_applies)
SPINUP = {"cfg2": 150, "tiny": 300, "cfg3": 4, "cfg5": 1, "cfg4shard": 20, "cfg5shard": 4}


def _child(args, prof_args, steps, tmp):
    """one child run of this bench (block applies only; args.accum may be overridden by the caller: the cheap passes run as
    --accum i8x4) under rocprofv3; returns the rows of the csv it wrote"""
    import csv
    import glob
    import subprocess

    cmd = ["rocprofv3"] + prof_args + ["--output-format", "csv", "-d", tmp, "-o", "pmc", "--", sys.executable,
           os.path.abspath(__file__), "--workload", args.workload, "--accum", args.accum, "--blockvec", str(args.blockvec),
           "--steps", str(steps), "--warmup", "1", "--no-cpu-baseline", "--no-pca", "--no-alt", "--no-e2e", "--no-cheap", "--traffic", "none"]
    subprocess.run(cmd, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, timeout=150, check=True,  # (a healthy child takes ~10 s)
                   env={k: v for k, v in dict(os.environ, TMPDIR="/tmp").items() if k != "LD_PRELOAD"})
    pat = "*kernel_trace.csv" if "--pmc" not in prof_args else "*counter_collection.csv"
    fs = glob.glob(os.path.join(tmp, "**", pat), recursive=True)
    if not fs:
        raise RuntimeError("rocprofv3 wrote no %s" % pat)
    key = "Dispatch_Id" if "--pmc" in prof_args else "Start_Timestamp"
    return sorted(csv.DictReader(open(fs[0])), key=lambda r: int(r[key]))


def _dominant_rows(rows, args, dom, id_key="Dispatch_Id"):
    """the launches of the dominant GEMM kernel among `rows` (one template serves K2 and K3 in the int8 mode: its launches
    alternate K2, K3, K2, ...)"""
    if args.accum.startswith("i8"):
        ids = sorted({int(r[id_key]) for r in rows if "k_gemm_i8" in r["Kernel_Name"]})
        want = {d for i, d in enumerate(ids) if i % 2 == (0 if dom == "xt_b" else 1)}
        return [r for r in rows if "k_gemm_i8" in r["Kernel_Name"] and int(r[id_key]) in want]
    key = "k_xt_b" if dom == "xt_b" else "k_x_t"
    return [r for r in rows if key in r["Kernel_Name"]]


def measure_counters(args, dom, hbm=True):
    """What hardware counters and the kernel trace say about the dominant GEMM kernel, from four child runs of this bench (block
    applies only) right after the timed region -- separate passes, as the guide prescribes:
      --pmc FETCH_SIZE, --pmc WRITE_SIZE       HBM bytes per launch (FETCH_SIZE counts 64 of every 128 B on gfx950: x2; KiB)
      --kernel-trace --pmc SQ_* GRBM_GUI_ACTIVE  effective clock, matrix-pipe busy fraction, VALU instructions per MFMA, LDS conflicts
      --kernel-trace                           the kernel's duration in EVENT-FREE steps (counters serialise kernels, a trace does not)
    """
    import tempfile

    res = dict(traffic=0.0 if hbm else None)
    try:
        tr = trace_gemms(args)
        if dom == "auto":  # the slower of the two in event-free launches
            dom = max(("xt_b", "x_t"), key=lambda d_: tr[d_][0] or 0.0)
        res["ms_trace"], res["trace_launches"] = tr[dom]
        res["ms_trace_both"] = {d_: tr[d_][0] for d_ in tr}
    except Exception as e:
        res["ms_trace"] = None
        res["trace_error"] = str(e)[:200]
        if dom == "auto":
            dom = "x_t"
    res["dom"] = dom
    alone_ns = []  # the dominant kernel's own duration in the counter runs: nothing runs beside it there
    for counter, scale in ((("FETCH_SIZE", 2048.0), ("WRITE_SIZE", 1024.0)) if hbm else ()):
        with tempfile.TemporaryDirectory(dir="/tmp") as tmp:
            rows = _dominant_rows(_child(args, ["--pmc", counter], 2, tmp), args, dom)
        vals = [float(r["Counter_Value"]) for r in rows if r["Counter_Name"] == counter]
        alone_ns += [int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in rows if r["Counter_Name"] == counter and r.get("End_Timestamp")]
        if not vals:
            raise RuntimeError("no launches of the dominant kernel in the counter file")
        res["traffic"] += sum(vals) / len(vals) * scale
    res["ms_alone"] = sum(alone_ns) / len(alone_ns) * 1e-6 if alone_ns else None
    try:
        sq = ["SQ_VALU_MFMA_BUSY_CYCLES", "GRBM_GUI_ACTIVE", "SQ_INSTS_MFMA", "SQ_INSTS_VALU", "SQ_LDS_BANK_CONFLICT", "SQ_LDS_IDX_ACTIVE"]
        with tempfile.TemporaryDirectory(dir="/tmp") as tmp:
            rows = _dominant_rows(_child(args, ["--kernel-trace", "--pmc"] + sq, 2, tmp), args, dom)
        mean = lambda c: (lambda v: sum(v) / len(v) if v else None)([float(r["Counter_Value"]) for r in rows if r["Counter_Name"] == c])
        dur = [int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in rows if r["Counter_Name"] == "GRBM_GUI_ACTIVE" and r.get("End_Timestamp")]
        busy, g, im, iv, lc, la = (mean(c) for c in sq)
        why = {}
        if g and dur:
            why["clock_ghz"] = g / 8 / (sum(dur) / len(dur))  # GRBM_GUI_ACTIVE sums the 8 XCDs
        if g and busy is not None:
            why["mfma_pipe_busy"] = busy / 1024 / (g / 8)     # SQ_VALU_MFMA_BUSY_CYCLES sums the 1024 SIMDs
        if im:
            why["valu_per_mfma"] = iv / im
        if la:
            why["lds_bank_conflict_frac"] = lc / la
        res["why"] = why
    except Exception as e:
        res["why"] = dict(error=str(e)[:200])
    return res


def trace_gemms(args):
    """event-free durations (ms) of the two GEMM kernels of a block apply: rocprofv3 --kernel-trace of a child run, last 10 launches each"""
    import tempfile

    with tempfile.TemporaryDirectory(dir="/tmp") as tmp:
        rows = _child(args, ["--kernel-trace"], 12, tmp)
    out = {}
    for dom in ("xt_b", "x_t"):
        d = [int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in _dominant_rows(rows, args, dom, id_key="Dispatch_Id")][-10:]
        out[dom] = (sum(d) / len(d) * 1e-6 if d else None, len(d))
    return out


def e2e_cli(fp, size, k, device):
    """wall-clock and phases of the flashpca CLI on a synthetic fileset in a fresh directory under /tmp"""
    import re
    import shutil
    import subprocess
    import tempfile

    N, P = (WORKLOADS[size]["N"], WORKLOADS[size]["P"])
    td = tempfile.mkdtemp(prefix="fpca_e2e_", dir="/tmp")
    try:
        pre = os.path.join(td, "syn")
        t0 = time.perf_counter()
        with open(pre + ".bed", "wb") as f:
            f.write(bytes([0x6C, 0x1B, 0x01]))
            step = max(1, (1 << 30) // ((N + 3) // 4))
            for j0 in range(0, P, step):
                with fp.Context.synthetic(N, min(step, P - j0), snp_begin=j0, n_pop=min(2 * k, 64), device=device) as c:
                    c.download_packed().tofile(f)
            f.flush()
            os.fsync(f.fileno())
        with open(pre + ".fam", "w") as f:
            f.write("".join("F%d I%d 0 0 0 -9\n" % (i, i) for i in range(N)))
        with open(pre + ".bim", "w") as f:
            f.write("".join("1 rs%d 0 %d A C\n" % (j, j + 1) for j in range(P)))
        t_write = time.perf_counter() - t0
        cmd = [fp.CLI_PATH, "--bfile", pre, "--ndim", str(k), "--outload", "load.txt", "--outmeansd", "ms.txt", "--device", str(device)]

        def run(label):
            t1 = time.perf_counter()
            r = subprocess.run(cmd, cwd=td, capture_output=True, text=True, env=dict(os.environ, FPCA_TIMING="1"), timeout=600)
            wall = time.perf_counter() - t1
            if r.returncode != 0:
                raise RuntimeError("flashpca CLI failed (%s): %s" % (label, (r.stderr or r.stdout)[-300:]))
            ph = {m.group(1).strip(): float(m.group(2)) for m in re.finditer(r"\[fpca-cli\] (.+?)\s+([0-9.]+) ms", r.stderr)}
            lib_ph = {m.group(1).strip(): float(m.group(2)) for m in re.finditer(r"\[fpca\] (device init \+ allocations|\.bed -> HBM|solver|run_pca|loadings, mean/sd)\s+([0-9.]+) ms", r.stderr)}
            return dict(wall_s=wall, phases_ms=ph, library_phases_ms=lib_ph)

        warm = run("warm")
        warm2 = run("warm")  # (the first run of a fresh binary also pages the executable and libamdhip64 in)
        if warm2["wall_s"] < warm["wall_s"]:
            warm = warm2
        dropped = False
        try:
            fd = os.open(pre + ".bed", os.O_RDONLY)
            os.posix_fadvise(fd, 0, 0, os.POSIX_FADV_DONTNEED)
            os.close(fd)
            dropped = True
        except Exception:
            pass
        cold = run("cold") if dropped else None
        bed_bytes = os.path.getsize(pre + ".bed")
        outs = {f: os.path.getsize(os.path.join(td, f)) for f in ("eigenvectors.txt", "pcs.txt", "eigenvalues.txt", "pve.txt", "load.txt", "ms.txt")}
        return dict(command="flashpca --bfile syn --ndim %d --outload load.txt --outmeansd ms.txt" % k, samples=N, snps=P, bed_bytes=bed_bytes,
                    output_bytes=sum(outs.values()), fileset_write_s=t_write, warm=warm,
                    cold=cold, cold_note="after fsync + posix_fadvise(DONTNEED) on the .bed: whether the pages really left the cache depends on the "
                                         "file system under /tmp (an overlay / tmpfs keeps them)")
    finally:
        shutil.rmtree(td, ignore_errors=True)


def power_sample(enqueue, seconds=2.5):
    """Package power and shader clock (rocm-smi --showpower --showclocks, as fast as it answers) WHILE `enqueue()` keeps the device
    busy for about `seconds`: the question is whether the int8 GEMMs sit at the package power cap (MI355X: 1,400 W) with the
    clock pulled down -- "throttled, not waiting" -- or run below it.  Samples under 60 % of the largest reading (ramp, drain) are
    dropped.  Returns None where rocm-smi is missing or says nothing parsable."""
    import re
    import shutil
    import subprocess
    import threading

    if not shutil.which("rocm-smi"):
        return None
    samples, stop = [], threading.Event()

    def poll():
        while not stop.is_set():
            try:
                o = subprocess.run(["rocm-smi", "--showpower", "--showclocks"], capture_output=True, text=True, timeout=10).stdout
            except Exception:
                return
            w = re.search(r"GPU\[0\].*?Package Power \(W\):\s*([0-9.]+)", o)
            c = re.search(r"GPU\[0\].*?sclk clock level:.*?\((\d+)Mhz\)", o)
            if w:
                samples.append((float(w.group(1)), float(c.group(1)) if c else None))

    th = threading.Thread(target=poll, daemon=True)
    t_end = time.perf_counter() + seconds
    th.start()
    n = 0
    while time.perf_counter() < t_end:
        n += enqueue()
    stop.set()
    th.join(timeout=15)
    cap = None
    try:
        m = re.search(r"GPU\[0\].*?Max Graphics Package Power \(W\):\s*([0-9.]+)", subprocess.run(["rocm-smi", "--showmaxpower"], capture_output=True, text=True, timeout=10).stdout)
        cap = float(m.group(1)) if m else None
    except Exception:
        pass
    if not samples:
        return None
    top = max(w for w, _ in samples)
    busy = sorted((w, c) for w, c in samples if w >= 0.6 * top)
    med = busy[len(busy) // 2]
    clk = sorted(c for _, c in busy if c)
    return dict(watts_median=med[0], watts_max=top, sclk_mhz_median=clk[len(clk) // 2] if clk else None, cap_watts=cap,
                frac_of_cap=(med[0] / cap) if cap else None, samples=len(busy), samples_dropped=len(samples) - len(busy), applies_during_sampling=n,
                source="rocm-smi --showpower --showclocks polled while the applies ran")


def e2e_size_auto(N, P):
    """The end-to-end CLI run takes the headline fileset (cfg3: a 12.5 GB .bed) when /tmp has room for it twice over and the machine
    has the memory to keep it in the page cache, else the 50,000 x 20,000 one."""
    import shutil

    need = (N + 3) // 4 * P
    try:
        free = shutil.disk_usage("/tmp").free
        mem = os.sysconf("SC_PAGE_SIZE") * os.sysconf("SC_AVPHYS_PAGES")
    except Exception:
        return "cfg2"
    return "cfg3" if (free > 2.2 * need and mem > 3 * need) else "cfg2"


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=None, help="timed block applies [60 / 400 / 12 for cfg3 / cfg2 / cfg5: a second or "
                                                             "two of device time -- the first ~50 ms after idle run at ramping clocks]")
    ap.add_argument("--warmup", type=int, default=None, help="untimed block applies before them [5 / 20 / 2]")
    ap.add_argument("--workload", default="cfg3", choices=sorted(WORKLOADS))
    ap.add_argument("--blockvec", type=int, default=0, choices=[0, 16, 32, 48, 64],
                    help="block width of the timed applies [0 = the width the solver picks for this k: 16 for k <= 64]")
    ap.add_argument("--accum", default="i8", choices=["fp64", "fp32", "i8"] + ["i8x%d" % s for s in range(4, 9)],
                    help="i8[xS] (default: the product's default mode) = exact-integer int8 MFMA on S (default 7) byte slices of the "
                         "fp64 operand, results equal to the fp64 path; fp64 = v_mfma_f64; fp32 = v_mfma_f32 products, fp64 long "
                         "accumulation")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-pca", action="store_true")
    ap.add_argument("--no-pca-hard", action="store_true", help="skip the second full PCA on a slowly converging spectrum (4 sub-populations)")
    ap.add_argument("--no-validate", action="store_true", help="several ranks: skip the self-validation against a one-context copy of the whole matrix on rank 0")
    ap.add_argument("--no-alt", action="store_true", help="skip the extra exact-int8-mode measurement of the same workload")
    ap.add_argument("--no-e2e", action="store_true", help="skip the end-to-end run of the flashpca CLI on a fileset in /tmp")
    ap.add_argument("--e2e-size", default="auto", choices=["auto", "cfg2", "cfg3"],
                    help="fileset of the end-to-end CLI run [auto: cfg3 -- the headline 500000 x 100000, a 12.5 GB .bed written to /tmp -- when "
                         "/tmp and the page cache have room for it, else cfg2: 50000 x 20000, 250 MB]")
    ap.add_argument("--no-cheap", action="store_true", help="skip the cheap_pass block (the eigensolver's 4-slice passes: kernel trace, counters, power sample)")
    ap.add_argument("--cpu-seconds", type=float, default=15.0, help="target CPU work of the bounded baseline sample")
    ap.add_argument("--traffic", default="auto", choices=["auto", "measure", "replay", "none"],
                    help="roofline.traffic: measure = two extra rocprofv3 --pmc passes (FETCH_SIZE; WRITE_SIZE) of a 2-step run of the "
                         "same workload as child processes after the timed region; replay = the committed passes under profiles/; "
                         "auto = measure when rocprofv3 is there and this is a one-GPU run, else replay")
    args = ap.parse_args()

    # the host driver of these boxes only supports dmabuf IPC; RCCL across processes needs this (already exported there)
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if os.environ.get("FPCA_BENCH_ONE_GPU"):
        local_rank = 0
    if args.gpus != world:
        if args.gpus > 1:
            sys.exit("bench.py --gpus %d must be launched with torch.distributed.run --nproc-per-node %d" % (args.gpus, args.gpus))
    import torch
    import torch.distributed as dist

    if not torch.cuda.is_available():
        sys.exit("bench.py needs an MI355X (torch.cuda.is_available() is False); there is no CPU path")
    torch.cuda.set_device(local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        # FPCA_BENCH_BACKEND=gloo lets two ranks share one GPU for a plumbing dry-run (the native RCCL communicator
        # then refuses the duplicate device and the torch.distributed fallback is exercised)
        backend = os.environ.get("FPCA_BENCH_BACKEND", "nccl")
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(backend)

    import flashpca_amd as fp

    w = WORKLOADS[args.workload]
    if args.steps is None:
        args.steps = STEPS[args.workload]
    if args.warmup is None:
        args.warmup = WARMUP[args.workload]
    steps_alt = min(args.steps, max(4, STEPS[args.workload] // 4))  # the other-mode comparison run
    N, k = w["N"], w["k"]
    b = args.blockvec or solver_blockvec(k)
    if w["scaling"] == "weak":
        # (the shard workloads keep the divisor of the full matrix they are a shard of)
        P_rank, P_total, snp_begin = w["P"], w["P"] * max(world, w.get("shards", 1)), rank * w["P"]
    else:
        lo, hi = w["P"] * rank // world, w["P"] * (rank + 1) // world
        P_rank, P_total, snp_begin = hi - lo, w["P"], lo

    t_gen = time.time()
    ctx = fp.Context.synthetic(N, P_rank, snp_begin=snp_begin, n_pop=min(2 * k, 64), device=local_rank, accum=args.accum)
    ctx.set_total_snps(P_total)
    ctx.stats()
    t_gen = time.time() - t_gen

    transport = "single"
    if world > 1:
        # RCCL inside the library (id broadcast over torch.distributed); falls back to torch.distributed's own
        # all-reduce (also RCCL) through the C-ABI hook if the native communicator cannot be created
        ids = [fp.Context.comm_unique_id() if rank == 0 else None]
        dist.broadcast_object_list(ids, src=0)
        ok = torch.ones(1, device="cuda")
        try:
            ctx.comm_init_rank(world, rank, ids[0])
            transport = "rccl-native"
        except Exception as e:  # pragma: no cover - needs multi-GPU
            ok.zero_()
            if rank == 0:
                print("native RCCL init failed (%s); using torch.distributed all-reduce" % e, file=sys.stderr)
        dist.all_reduce(ok, op=dist.ReduceOp.MIN)
        if ok.item() == 0:
            transport = "torch.distributed"

            class _Arr:
                def __init__(self, ptr, n):
                    self.__cuda_array_interface__ = dict(shape=(n,), typestr="<f8", data=(ptr, False), version=2)

            def allreduce(ptr, count, stream):
                ctx.synchronize()
                t = torch.as_tensor(_Arr(ptr, count), device="cuda")
                dist.all_reduce(t)
                torch.cuda.synchronize()
                return 0

            def allgather(snd, rcv, count, stream):
                ctx.synchronize()
                dist.all_gather_into_tensor(torch.as_tensor(_Arr(rcv, count * world), device="cuda"), torch.as_tensor(_Arr(snd, count), device="cuda"))
                torch.cuda.synchronize()
                return 0

            def reducescatter(snd, rcv, count, stream):
                ctx.synchronize()
                dist.reduce_scatter_tensor(torch.as_tensor(_Arr(rcv, count), device="cuda"), torch.as_tensor(_Arr(snd, count * world), device="cuda"))
                torch.cuda.synchronize()
                return 0

            ctx.set_allreduce(allreduce)
            ctx.set_rank(world, rank)  # (lets fpca_pca row-shard the solver over this transport too)
            if backend == "nccl":
                ctx.set_collectives(allgather, reducescatter)  # ... with the call sequence of the native path

    rows = ctx.block_rows()
    g = torch.Generator(device="cuda")
    g.manual_seed(1234)
    B = torch.zeros((rows, b), dtype=torch.float64, device="cuda")
    B[:N] = torch.rand((N, b), dtype=torch.float64, device="cuda", generator=g) - 0.5
    Y = torch.zeros((rows, b), dtype=torch.float64, device="cuda")
    torch.cuda.synchronize()

    def barrier():
        if world > 1:
            dist.barrier()

    # clock spin-up, untimed and on top of the W warm-up steps the caller asked for: the first ~50 ms of work after idle run
    # at ramping clocks whatever W is (a 20-step region at cfg2 read 0.62 ms per step right after start-up, 0.53 after)
    # (a fixed count, not a time limit: with several ranks every apply is a collective)
    for _ in range(SPINUP[args.workload]):
        ctx.apply_xxt_dev(B.data_ptr(), b, Y.data_ptr())
    ctx.synchronize()
    for _ in range(args.warmup):
        ctx.apply_xxt_dev(B.data_ptr(), b, Y.data_ptr())
    ctx.synchronize()
    torch.cuda.synchronize()
    barrier()
    torch.cuda.synchronize()
    # THE TIMED REGION: exactly K block applies, issued as the solver issues them -- no HIP events, no profiling hooks
    t0 = time.perf_counter()
    for _ in range(args.steps):
        ctx.apply_xxt_dev(B.data_ptr(), b, Y.data_ptr())
    ctx.synchronize()
    torch.cuda.synchronize()
    barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device="cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    # Right after it, an INSTRUMENTED region of the same applies: in-stream HIP events (on the kernels' own stream) around K2 /
    # K3 / their GEMM kernels / the all-reduce of every step.  Eight events per apply stretch a step by a few per cent, which
    # is why they are not in the region that sets `value`; the per-kernel times below belong to THESE steps
    # (ms_per_step_instrumented), and the kernel's duration in event-free steps comes from the kernel trace further down.
    steps_i = max(4, args.steps // 4)
    ctx.profile_begin(steps_i, sample_every=1)
    t0i = time.perf_counter()
    for _ in range(steps_i):
        ctx.apply_xxt_dev(B.data_ptr(), b, Y.data_ptr())
    ctx.synchronize()
    barrier()
    elapsed_i = time.perf_counter() - t0i
    prof = ctx.profile_end(b)

    # SNPs all ranks processed per step (the shard workloads process one shard; their divisor is the full matrix's)
    P_done = P_rank * world if w["scaling"] == "weak" else P_total
    cells = float(N) * float(P_done) * b * args.steps
    value = cells / elapsed

    _peaks = {}

    def mfma_stream_peak(pattern):
        """sustained rate of a bare MFMA stream with nothing else going on (fpca_debug_mfma_peak; ~50 ms): v_mfma_f64_16x16x4 in
        TFLOP/s (pattern 0, two waves per SIMD) or v_mfma_i32_32x32x32_i8 on random bytes in TOP/s (pattern 11, one wave per SIMD)
        -- the practical ceiling of the GEMM kernels: the int8 stream is itself limited by the package power cap"""
        if pattern not in _peaks:
            import ctypes

            v = ctypes.c_double(0)
            best = 0.0
            for _ in range(2):
                # ~50 ms per launch: long enough for the power governor to settle (a 3 ms launch reads 3-5 % high)
                fp._lib.check(fp.lib().fpca_debug_mfma_peak(1 if pattern >= 10 else 2, 300000 if pattern >= 10 else 100000, pattern, ctypes.byref(v)))
                best = max(best, v.value)
            _peaks[pattern] = best
        return _peaks[pattern]

    # roofline of the dominant kernel (both GEMMs carry 2 N P_g b flops per launch; the slower one dominates)
    # dominant kernel = the slower of the two GEMM kernels; its own launch duration from HIP events recorded around that
    # launch inside the timed region (ms_gemm_*); ms_xt_b / ms_x_t are the whole K2 / K3 stages (slicing, sparse gathers,
    # GEMM, combine)
    dom = "xt_b" if prof["ms_gemm_xt"] >= prof["ms_gemm_x"] else "x_t"
    ms_dom = max(prof["ms_gemm_xt"], prof["ms_gemm_x"])
    flops_launch = 2.0 * N * P_rank * b
    roofline = dict(bound="mfma", kernel=dom, achieved=flops_launch / (ms_dom * 1e-3) / 1e12,
                    peak=FP64_MFMA_PEAK_TFLOPS if args.accum == "fp64" else FP32_MFMA_PEAK_TFLOPS,
                    unit="TFLOP/s", traffic=None,
                    ms_xt_b=prof["ms_xt"], ms_x_t=prof["ms_x"], ms_allreduce=prof["ms_allreduce"],
                    ms_gemm_kernel_xt_b=prof["ms_gemm_xt"], ms_gemm_kernel_x_t=prof["ms_gemm_x"],
                    flops_per_launch=flops_launch,
                    packed_gbs=((N + 3) // 4) * P_rank / (ms_dom * 1e-3) / 1e9)
    if args.accum.startswith("i8"):
        S = int(args.accum[3:]) if len(args.accum) > 2 else 7
        mm = ctx.missing_mode(b)  # 0/1: two integer matrices on the matrix cores; 2/3: G.M alone (+ sparse gathers for E)
        nmat = 1 if mm in (2, 3, 4) else 2
        ops_launch = nmat * flops_launch * S  # integer matrices x S slices of every operand column
        roofline.update(achieved=ops_launch / (ms_dom * 1e-3) / 1e12, peak=I8_MFMA_PEAK_TOPS, unit="TOP/s", slices=S,
                        integer_matrices_on_mfma=nmat,
                        missing_call_path=MISSING_PATH[mm],
                        ops_per_launch=ops_launch, fp64_equivalent_tflops=flops_launch / (ms_dom * 1e-3) / 1e12,
                        peak_measured_pure_mfma_stream=mfma_stream_peak(11))  # random operands; power-limited (measured in this run)
        del roofline["flops_per_launch"]
    roofline["frac"] = roofline["achieved"] / roofline["peak"]
    roofline["duration_source"] = "HIP events around the launch in the instrumented steps"
    roofline["ms_dominant_kernel"] = ms_dom
    roofline["ms_per_step_instrumented"] = elapsed_i / steps_i * 1e3
    roofline["hip_events_on_steps"] = ("%d instrumented steps run right after the %d timed ones (the timed region carries no events); ms_xt_b / ms_x_t / "
                                       "ms_gemm_kernel_* are averages over the instrumented steps" % (prof["nsteps"], args.steps))
    if args.accum == "fp64":
        roofline["peak_measured_pure_mfma_stream"] = mfma_stream_peak(0)  # v_mfma_f64 stream, 2 waves/SIMD (measured in this run)
    # HBM traffic per launch of the dominant kernel: hardware counters serialise the kernels, so they cannot be read inside
    # the timed region.  `measure`: right after it, this process runs itself twice more under rocprofv3 --pmc (FETCH_SIZE,
    # then WRITE_SIZE: separate passes, as the guide prescribes; 2 block applies each, nothing else) and reads the dominant
    # kernel's counters per launch -- FETCH_SIZE x 2 (the gfx950 correction: it counts 64 of every 128 B) + WRITE_SIZE, KiB.
    # `replay`: the committed passes of the same workload under profiles/.  `traffic_source` says which.
    roofline["algorithmic_bytes"] = float((N + 3) // 4) * P_rank + 8.0 * b * (N + P_rank)
    roofline["traffic_measured_in_this_run"] = False
    under_profiler = any(k.startswith(("ROCPROF", "ROCP_", "ROCTX")) for k in os.environ) or "rocprof" in os.environ.get("LD_PRELOAD", "")
    mode = args.traffic
    if mode == "auto":
        import shutil

        mode = "measure" if (world == 1 and shutil.which("rocprofv3") and args.accum in ("i8", "fp64") and not under_profiler) else "replay"
    if mode == "measure" and world == 1:
        try:
            cnt = measure_counters(args, dom)
            roofline["traffic"] = cnt["traffic"]
            roofline["traffic_measured_in_this_run"] = True
            per_launch = roofline.get("ops_per_launch", roofline.get("flops_per_launch"))
            if cnt.get("ms_trace"):
                # THE figure: the kernel's own duration in event-free steps (rocprofv3 --kernel-trace of this command line with
                # block applies only: the trace reads the dispatch timestamps, nothing is serialised, the gathers of the same stage
                # run beside the kernel exactly as in the timed region)
                roofline["ms_dominant_kernel_instrumented_steps"] = ms_dom
                roofline["achieved_instrumented_steps"] = roofline["achieved"]
                roofline["ms_dominant_kernel"] = cnt["ms_trace"]
                roofline["achieved"] = per_launch / (cnt["ms_trace"] * 1e-3) / 1e12
                roofline["frac"] = roofline["achieved"] / roofline["peak"]
                roofline["duration_source"] = ("rocprofv3 --kernel-trace of a child run of this bench (event-free block applies, last %d launches of the "
                                               "dominant kernel); *_instrumented_steps: HIP events in this process" % cnt["trace_launches"])
            if cnt.get("ms_alone"):
                # secondary: the same kernel with the chip to itself (in the counter runs the kernels are serialised, so the
                # sparse gathers that normally run on the low-priority stream UNDER the GEMM do not)
                roofline["kernel_alone"] = dict(ms=cnt["ms_alone"], achieved=per_launch / (cnt["ms_alone"] * 1e-3) / 1e12,
                                                frac=per_launch / (cnt["ms_alone"] * 1e-3) / 1e12 / roofline["peak"],
                                                source="durations of the same launches in the two HBM-counter child runs (counters serialise the kernels)")
            roofline["why"] = cnt.get("why")
            roofline["traffic_source"] = ("rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (two child runs of this bench, 2 block applies "
                                          "each, right after the timed region): FETCH_SIZE x2 + WRITE_SIZE per launch of the dominant kernel")
        except Exception as e:  # no rocprofv3, counters busy, time-out: fall back to the committed passes
            print("counter measurement failed (%s); replaying profiles/" % e, file=sys.stderr)
            mode = "replay"
    if mode == "replay":
        for rnd in ("r04", "r03", "r02", "r01"):
            try:
                if world == 1 and args.accum == "i8":
                    fn = "profiles/%s_pmc_summary_i8.json" % rnd
                    pmc = json.load(open(os.path.join(ROOT, fn)))[args.workload]["gemm_i8_" + dom]
                elif world == 1 and args.accum == "fp64":
                    fn = "profiles/%s_pmc_summary.json" % rnd
                    pmc = json.load(open(os.path.join(ROOT, fn)))[args.workload][dom]
                else:
                    break
                roofline["traffic"] = pmc["hbm_read_bytes_per_launch"] + pmc["hbm_write_bytes_per_launch"]
                roofline["traffic_source"] = "replayed from " + fn + " (rocprofv3 --pmc passes of this workload: FETCH_SIZE x2 + WRITE_SIZE per launch)"
                break
            except Exception:
                continue

    out = dict(metric="genotype cells/sec (N x P x iters) for k=20 PCA", value=value, unit="cells/s", n_gpus=world,
               steps=args.steps, warmup=args.warmup, spinup_applies=SPINUP[args.workload],
               ms_per_step=elapsed / args.steps * 1e3, higher_is_better=True,
               scaling=w["scaling"], vs_baseline=None,
               dtype={"fp64": "f64", "fp32": "f32 (fp64 long accumulation)"}.get(
                   args.accum, "i8 (integer genotypes x 7 byte slices of the f64 operand; exact i32 accumulation, f64 recombination; "
                               "agrees with the f64 kernels to ~3e-15, see fp64_mode)"),
               data="synthetic",
               config=dict(workload=args.workload + ": " + w["desc"], samples=N, snps_total=P_total, snps_per_gpu=P_rank,
                           k=k, blockvec=b, solver_default_blockvec=solver_blockvec(k), missing_call_rate=0.001, parallelism="snp-shard x%d + all-reduce(N x b) [%s]" % (world, transport),
                           iters_per_step=b, generate_s=round(t_gen, 3),
                           solver_passes="`value` and every fpca_apply_* call: exact (7 slices).  fpca_pca (`pca*` blocks): exact passes until the measured "
                                         "decay predicts a long solve, then 4-slice passes verified by exact ones (see cheap_applies there)"),
               roofline=roofline)

    # ---- what the realistic solves actually run: the eigensolver's CHEAP passes (block apply on 4 byte slices of the operand: the
    # 2-column-tile instantiation of the int8 GEMM; 128 of the 141 passes of pca_realistic, 137 of 150 of pca_hard_spectrum) -- its own
    # roofline, priced both ways (at 64 slice-columns the ridge puts the kernel on the HBM side: 1.57 ms at 8 TB/s against 1.27 ms of
    # matrix work at the int8 issue peak), its counters, and the package power while it loops, beside the same sample of the exact
    # apply: "throttled at the power cap, not waiting on memory" is a claim a power reading can refute -------------------------
    if world == 1 and args.accum == "i8" and not args.no_cheap and not under_profiler:
        S4 = 4
        ops4 = 2.0 * N * P_rank * b * S4
        cp = dict(what="block apply on %d byte slices of the f64 operand (30 bits below each column's maximum): the eigensolver's cheap passes, "
                       "verified by exact ones before convergence is declared" % S4,
                  kernel="k_gemm_i8<I8Cfg<false, 2, 2, ...>>: 64 slice-columns = 2 column tiles of 32", slices=S4, blockvec=b,
                  ops_per_launch=ops4, algorithmic_bytes=roofline["algorithmic_bytes"], peak_mfma_tops=I8_MFMA_PEAK_TOPS, peak_hbm_gbs=HBM_PEAK_GBS)
        try:
            with fp.Context.synthetic(N, P_rank, snp_begin=snp_begin, n_pop=min(2 * k, 64), device=local_rank, accum="i8x%d" % S4) as c4:
                c4.set_total_snps(P_total)
                c4.stats()
                for _ in range(max(3, args.warmup)):
                    c4.apply_xxt_dev(B.data_ptr(), b, Y.data_ptr())
                c4.synchronize()
                steps4 = max(8, args.steps // 2)
                t1 = time.perf_counter()
                for _ in range(steps4):
                    c4.apply_xxt_dev(B.data_ptr(), b, Y.data_ptr())
                c4.synchronize()
                cp.update(steps=steps4, ms_per_step=(time.perf_counter() - t1) / steps4 * 1e3)
                cp["speedup_vs_exact_step"] = (elapsed / args.steps * 1e3) / cp["ms_per_step"]

                def enq4():
                    for _ in range(8):
                        c4.apply_xxt_dev(B.data_ptr(), b, Y.data_ptr())
                    c4.synchronize()
                    return 8

                cp["power"] = power_sample(enq4)

            def enq7():
                for _ in range(8):
                    ctx.apply_xxt_dev(B.data_ptr(), b, Y.data_ptr())
                ctx.synchronize()
                return 8

            roofline["power"] = power_sample(enq7)
            a4 = argparse.Namespace(**vars(args))
            a4.accum = "i8x%d" % S4
            cnt4 = measure_counters(a4, "auto", hbm=False)
            if cnt4.get("ms_trace"):
                ms4 = cnt4["ms_trace"]
                cp.update(dominant=cnt4["dom"], ms_dominant_kernel=ms4, ms_gemm_kernels=cnt4.get("ms_trace_both"),
                          achieved_tops=ops4 / (ms4 * 1e-3) / 1e12, frac_mfma=ops4 / (ms4 * 1e-3) / 1e12 / I8_MFMA_PEAK_TOPS,
                          achieved_gbs=cp["algorithmic_bytes"] / (ms4 * 1e-3) / 1e9, frac_hbm=cp["algorithmic_bytes"] / (ms4 * 1e-3) / 1e9 / HBM_PEAK_GBS,
                          duration_source="rocprofv3 --kernel-trace of a child run of this bench with --accum i8x%d (event-free block applies, last %d "
                                          "launches of the slower GEMM)" % (S4, cnt4["trace_launches"]))
            else:
                cp["trace_error"] = cnt4.get("trace_error")
            cp["why"] = cnt4.get("why")
        except Exception as e:  # never lose the line over a side block
            cp["error"] = str(e)[:300]
        out["cheap_pass"] = cp

    # ---- several ranks: the line validates itself ---------------------------------------------------------------------------
    # Rank 0 additionally holds the WHOLE matrix in one context without a communicator (12.5 GB at the headline size) and applies
    # it to the same block: multi_rank_parity = max |Y_N - Y_1| / max |Y_1| (Y_N: what the N ranks' shards + the all-reduce of
    # the timed region produced; the shards of svdwide.cpp:48-62 summed in another order: ~1e-15).  The eigenvalues of the
    # multi-rank solves (row-sharded solver, and the replicated one) are compared with the one-context solve of the same matrix
    # further down.  A line that is fast and wrong says so: multi_rank_validation.passed = false.
    whole = None
    if world > 1 and not args.no_validate:
        val = dict(passed=None)
        par = torch.zeros(1, dtype=torch.float64, device="cuda")
        if rank == 0:
            whole = fp.Context.synthetic(N, P_done, snp_begin=0, n_pop=min(2 * k, 64), device=local_rank, accum=args.accum)
            whole.set_total_snps(P_total)
            Y1 = torch.zeros_like(Y)
            whole.apply_xxt_dev(B.data_ptr(), b, Y1.data_ptr())
            whole.synchronize()
            par[0] = float((torch.max(torch.abs(Y - Y1)) / torch.max(torch.abs(Y1))).item())
            del Y1
        barrier()
        dist.broadcast(par, src=0)
        # every rank must hold the same all-reduced block: the largest deviation of any rank's Y from rank 0's
        Y0 = Y.clone()
        dist.broadcast(Y0, src=0)
        dev = torch.max(torch.abs(Y - Y0)).reshape(1) / torch.max(torch.abs(Y0)).clamp_min(1e-300)
        dist.all_reduce(dev, op=dist.ReduceOp.MAX)
        del Y0
        val["multi_rank_parity"] = float(par.item())
        val["max_deviation_between_ranks"] = float(dev.item())
        val["operator_ok"] = bool(val["multi_rank_parity"] < 1e-11 and val["max_deviation_between_ranks"] == 0.0)
        out["multi_rank_validation"] = val

    # ---- side measurements of the same operator, each a bounded number of block applies -----------------------------------
    def side_apply(c, bw, steps_s):
        """steps_s timed block applies of width bw on context c (random block, warm): wall, cells/s, GEMM kernel times, roofline"""
        Bs = torch.zeros((rows, bw), dtype=torch.float64, device="cuda")
        Bs[:N] = torch.rand((N, bw), dtype=torch.float64, device="cuda", generator=g) - 0.5
        Ys = torch.zeros((rows, bw), dtype=torch.float64, device="cuda")
        for _ in range(max(2, args.warmup)):
            c.apply_xxt_dev(Bs.data_ptr(), bw, Ys.data_ptr())
        c.synchronize()
        barrier()
        c.profile_begin(steps_s, sample_every=4 if steps_s >= 8 else 1)  # (side blocks: events on every 4th step)
        t1 = time.perf_counter()
        for _ in range(steps_s):
            c.apply_xxt_dev(Bs.data_ptr(), bw, Ys.data_ptr())
        c.synchronize()
        barrier()
        el_s = time.perf_counter() - t1
        ps = c.profile_end(bw)
        ms_s = max(ps["ms_gemm_xt"], ps["ms_gemm_x"])
        sw = dict(blockvec=bw, steps=steps_s, ms_per_step=el_s / steps_s * 1e3, value=float(N) * P_done * bw * steps_s / el_s,
                  unit="cells/s", ms_xt_b=ps["ms_xt"], ms_x_t=ps["ms_x"], ms_gemm_kernel_xt_b=ps["ms_gemm_xt"], ms_gemm_kernel_x_t=ps["ms_gemm_x"])
        if args.accum.startswith("i8"):
            S = int(args.accum[3:]) if len(args.accum) > 2 else 7
            mm = c.missing_mode(bw)
            nm = 1 if mm in (2, 3, 4) else 2
            ops = nm * 2.0 * N * P_rank * bw * S
            sw["missing_call_path"] = MISSING_PATH[mm]
            sw["roofline"] = dict(bound="mfma", achieved=ops / (ms_s * 1e-3) / 1e12, peak=I8_MFMA_PEAK_TOPS, unit="TOP/s",
                                  frac=ops / (ms_s * 1e-3) / 1e12 / I8_MFMA_PEAK_TOPS, ops_per_launch=ops, integer_matrices_on_mfma=nm,
                                  note="S b = %d slice-columns fill %.1f of %d column tiles" % (S * bw, S * bw / 32.0, -(-S * bw // 32)))
        del Bs, Ys
        return sw

    # (a) the widest tile-exact pass: 32 columns (7 x 32 slice-columns = exactly 7 column tiles of the int8 GEMM) -- what round 2
    #     quoted as `value`; more cells/s per pass, but a k = 20 solve needs 5 of them against 7 of the 16-column ones
    b_wide = 32 if k <= 64 else 64
    if b_wide != b and not args.no_pca:
        out["apply_at_b%d" % b_wide] = side_apply(ctx, b_wide, max(4, args.steps // 4))
    # (b) the same matrix with 2 % missing calls (real array data carries 1-2 %; the synthetic spec says 0.1 %): above 0.5 % the
    #     missing-call indicator takes the dense route, a second integer matrix on the matrix cores instead of sparse gathers
    if world == 1 and not args.no_alt and args.accum.startswith("i8"):
        with fp.Context.synthetic(N, P_rank, snp_begin=snp_begin, n_pop=min(2 * k, 64), missing_rate=0.02, device=local_rank,
                                  accum=args.accum) as cm:
            cm.set_total_snps(P_total)
            cm.stats()
            sm = side_apply(cm, b, max(4, args.steps // 4))
            sm["missing_call_rate"] = 0.02
            sm["slowdown_vs_value"] = sm["ms_per_step"] / (elapsed / args.steps * 1e3)
            out["apply_at_missing_2pct"] = sm
        # ... and what the eigensolver's 4-slice passes cost there (round 6: the two-matrix kernels exist from 2 column tiles up with
        # 64-row waves; rounds 2-5 multiplied two tiles of zero padding -- 15.3 ms per pass, profiles/r06_missing_cheap_probe.txt)
        try:
            with fp.Context.synthetic(N, P_rank, snp_begin=snp_begin, n_pop=min(2 * k, 64), missing_rate=0.02, device=local_rank, accum="i8x4") as cm4:
                cm4.set_total_snps(P_total)
                cm4.stats()
                s4 = side_apply(cm4, b, max(4, args.steps // 4))
                out["apply_at_missing_2pct"]["cheap_pass_ms_per_step"] = s4["ms_per_step"]
                out["apply_at_missing_2pct"]["cheap_pass_speedup_vs_exact_step"] = sm["ms_per_step"] / s4["ms_per_step"]
        except Exception as e:  # never lose the line over a side block
            out["apply_at_missing_2pct"]["cheap_pass_error"] = str(e)[:200]

    # (c) ... and with the SAME 2 % of the calls missing, but per-SNP rates log-normally distributed (sd of ln(rate) 1.5: most SNPs below
    #     1 %, a long tail of poor assays) -- where real arrays sit, between "uniform" and the concentrated profile of pca_realistic.  The
    #     route is chosen from a cost model over K1's per-SNP counts (missing_routes.hip hybrid_classify): SNPs whose calls cost more to
    #     gather than their indicator row costs on the matrix cores go dense, the shard goes hybrid if that beats the two-matrix kernels
    if world == 1 and not args.no_alt and args.accum.startswith("i8"):
        for tag, mean in (("apply_at_missing_2pct_lognormal", 0.02), ("apply_at_missing_1pct_lognormal", 0.01)):
            with fp.Context.synthetic(N, P_rank, snp_begin=snp_begin, n_pop=min(2 * k, 64), missing_rate=mean, missing_model=2, lognormal_sigma=1.5,
                                      device=local_rank, accum=args.accum) as cm:
                cm.set_total_snps(P_total)
                ms_l, _ = cm.stats()
                sm = side_apply(cm, b, max(4, args.steps // 8))
                sm["missing_call_rate_mean"] = mean
                sm["lognormal_sigma"] = 1.5
                sm["slowdown_vs_value"] = sm["ms_per_step"] / (elapsed / args.steps * 1e3)
                out[tag] = sm

    # ---- one full PCA solve to convergence (reported, not the timed region) -------------------------------
    watchdog = None
    if not args.no_pca and world > 1:
        # With several ranks fpca_pca runs the row-sharded solver (all-gather / reduce-scatter inside RCCL) -- a path no
        # multi-GPU box has executed yet.  The line the driver waits for must not depend on it: if the solves have not come
        # back after two minutes, every rank leaves and rank 0 prints the line with what has been measured so far.
        import threading

        def bail():
            if rank == 0:
                out["pca"] = dict(error="fpca_pca did not return within 180 s on %d ranks (collective hang?); the block apply above is unaffected" % world)
                sys.stdout.flush()
                print(json.dumps(out), flush=True)
            os._exit(0)

        watchdog = threading.Timer(180.0, bail)
        watchdog.daemon = True
        watchdog.start()
    if not args.no_pca:
        # two solves: the first one also allocates the Krylov basis blocks and the solver's scratch (kept by the context
        # afterwards), `wall_s` is the second
        import gc

        def solve(**kw):
            barrier()
            t1 = time.perf_counter()
            r0 = ctx.pca(ndim=k, allow_unconverged=True, **kw)
            ctx.synchronize()
            barrier()
            wf = time.perf_counter() - t1
            # (the first call's 160 MB of results are dropped BEFORE the clock starts: returning touched pages to the OS costs this
            # process ~10 ms on these boxes -- the harness's own housekeeping, not part of a solve)
            del r0
            gc.collect()
            t1 = time.perf_counter()
            r0 = ctx.pca(ndim=k, allow_unconverged=True, **kw)
            ctx.synchronize()
            barrier()
            return r0, wf, time.perf_counter() - t1

        solver_kw = {}
        if world > 1:
            # the row-sharded solver has never run over RCCL with a second GPU (its exchange tests itself at the first solve and
            # refuses with FPCA_ECOMM if a piece lands in the wrong place): whatever it does, the line must still come out, with a
            # solve in it -- the replicated solver (round 2's: one all-reduce per apply, the path `value` has just timed)
            okf = torch.ones(1, device="cuda")
            try:
                r, wall_first, wall = solve()
            except Exception as e:  # pragma: no cover - needs multi-GPU
                okf.zero_()
                out["pca_rowsharded_error"] = str(e)[:400]
                print("row-sharded fpca_pca failed on rank %d: %s" % (rank, e), file=sys.stderr, flush=True)
            dist.all_reduce(okf, op=dist.ReduceOp.MIN)
            if okf.item() == 0:
                out.setdefault("pca_rowsharded_error", "failed on another rank")
                solver_kw = dict(replicated_solver=True)
                r, wall_first, wall = solve(**solver_kw)
        else:
            r, wall_first, wall = solve()
        info = r["info"]
        out["pca"] = dict(wall_s=wall, first_call_wall_s=wall_first, blockvec=info["blockvec"], converged=bool(info["converged"]),
                          block_applies=info["block_applies"],
                          vector_ops=info["vector_ops"], restarts=info["restarts"],
                          cells_per_s=float(N) * P_done * info["vector_ops"] / wall,
                          seconds_apply=info["seconds_apply"], seconds_ortho=info["seconds_ortho"],
                          seconds_host=info["seconds_host"], seconds_download=info["seconds_download"], seconds_post=info["seconds_post"],
                          wall_minus_apply_s=wall - info["seconds_apply"],
                          eigenvalue_1=float(r["d"][0]), eigenvalue_k=float(r["d"][-1]),
                          max_rel_residual=info["max_residual"])
        out["pca"].update(cheap_applies=info["cheap_applies"], cheap_slices=info["cheap_slices"], seconds_apply_exact=info["seconds_exact"],
                          solver_path=fp._lib.SOLVER_PATH.get(info["solver_path"], str(info["solver_path"])))
        if world > 1:
            # (since round 5 the library demotes itself -- rank-agreed -- when the exchange self-test or a collective of the row-sharded
            #  solve fails: info.solver_path says which layout the solve ended on; the try / except above is the second line of defence)
            out["pca"]["solver"] = ("replicated (one all-reduce per apply; the row-sharded solver FAILED with an error, see pca_rowsharded_error)" if solver_kw else
                                    "row-sharded (all-gather -> K2, K3 -> reduce-scatter per apply; every rank orthogonalises N / %d rows)" % world
                                    if info["solver_path"] == 1 else "replicated by the library's own demotion: " + fp._lib.SOLVER_PATH.get(info["solver_path"], "?"))
            out["pca"]["collectives"] = dict(zip(("calls", "bytes"), ctx.collective_stats()))
            if not args.no_validate:
                # the same solve with round 2's replicated solver (one all-reduce per apply, every rank keeps the whole basis), and
                # on rank 0 alone with the whole matrix in one context: three routes to the same eigenvalues
                import numpy as np

                barrier()
                t1 = time.perf_counter()
                rr = ctx.pca(ndim=k, allow_unconverged=True, replicated_solver=True)
                ctx.synchronize()
                barrier()
                wall_r = time.perf_counter() - t1
                dd = torch.zeros(3, dtype=torch.float64, device="cuda")
                if rank == 0:
                    r1 = whole.pca(ndim=k, allow_unconverged=True)
                    t1 = time.perf_counter()
                    r1 = whole.pca(ndim=k, allow_unconverged=True)
                    whole.synchronize()
                    wall_1 = time.perf_counter() - t1
                    dd[0] = float(np.max(np.abs(r["d"] - r1["d"]) / r1["d"]))
                    dd[1] = float(np.max(np.abs(rr["d"] - r1["d"]) / r1["d"]))
                    # eigenvectors up to sign: |u_N . u_1| of the first and the k-th
                    dd[2] = float(min(abs(float(r["U"][:, j] @ r1["U"][:, j])) for j in (0, k - 1)))
                    val.update(one_context_pca_wall_s=wall_1, one_context_block_applies=r1["info"]["block_applies"])
                    del r1
                barrier()
                dist.broadcast(dd, src=0)
                val.update(eigenvalues_rowsharded_vs_one_context=float(dd[0].item()), eigenvalues_replicated_vs_one_context=float(dd[1].item()),
                           eigenvector_alignment_min=float(dd[2].item()), replicated_solver_wall_s=wall_r,
                           replicated_solver_block_applies=rr["info"]["block_applies"])
                val["solver_ok"] = bool(dd[0].item() < 1e-9 and dd[1].item() < 1e-9 and dd[2].item() > 1 - 1e-6 and info["converged"] and not solver_kw)
                del rr
        if world > 1 and not args.no_validate:
            val["passed"] = bool(val.get("operator_ok") and val.get("solver_ok", True))
            if not val["passed"] and rank == 0:
                print("MULTI-RANK VALIDATION FAILED: %s" % json.dumps(val), file=sys.stderr, flush=True)
    if watchdog is not None:
        watchdog.cancel()
    if world > 1 and not args.no_validate and out["multi_rank_validation"].get("passed") is None:
        out["multi_rank_validation"]["passed"] = bool(out["multi_rank_validation"].get("operator_ok"))  # (--no-pca: the operator only)

    # ---- the same solve on a slowly converging spectrum: 4 sub-populations, so that 17 of the 20 wanted eigenvalues sit in
    # the bulk (231 single-vector ops instead of 42 in the probe) -- the cost of a PCA whose k reaches past the
    # structure in the data; reported beside the easy one, not part of `value` ----------------------------------------------
    if world == 1 and not args.no_pca and not args.no_pca_hard:
        with fp.Context.synthetic(N, P_rank, snp_begin=snp_begin, n_pop=4, device=local_rank, accum=args.accum) as ch:
            ch.set_total_snps(P_total)
            ch.stats()
            ch.pca(ndim=k, allow_unconverged=True, max_applies=-(-k // solver_blockvec(k)) + 1)  # one-off set-up of the arithmetic mode, untimed like above
            ch.synchronize()
            t1 = time.perf_counter()
            rh = ch.pca(ndim=k, allow_unconverged=True)
            ch.synchronize()
            wall_h = time.perf_counter() - t1
            ih = rh["info"]
            out["pca_hard_spectrum"] = dict(n_pop=4, wall_s=wall_h, converged=bool(ih["converged"]), block_applies=ih["block_applies"],
                                            vector_ops=ih["vector_ops"], restarts=ih["restarts"], seconds_apply=ih["seconds_apply"],
                                            seconds_ortho=ih["seconds_ortho"], seconds_host=ih["seconds_host"],
                                            cheap_applies=ih["cheap_applies"], cheap_slices=ih["cheap_slices"], seconds_apply_exact=ih["seconds_exact"],
                                            eigenvalue_1=float(rh["d"][0]), eigenvalue_k=float(rh["d"][-1]),
                                            max_rel_residual=ih["max_residual"])

    # ---- ... and on the REALISTIC profile (flashpca_amd/csrc/synth.hpp, round 4): allele frequencies from a rare-variant spectrum
    # (per-SNP sd over a 16x range instead of 2.3x), missing calls concentrated in 5 % of the SNPs at 10-30 % (~1 % overall: the
    # dense missing-indicator route for those SNPs), 10 sub-populations = 9 structured eigenvalues with 11 of the 20 wanted ones in
    # the bulk -- what a PCA of array genotypes looks like (HapMap3's tenth eigenvalue sits 1 % above its bulk) ---------
    if world == 1 and not args.no_pca and not args.no_pca_hard:
        with fp.Context.synthetic(N, P_rank, snp_begin=snp_begin, n_pop=10, realistic=True, device=local_rank, accum=args.accum) as cr:
            cr.set_total_snps(P_total)
            ms_r, _ = cr.stats()
            cr.pca(ndim=k, allow_unconverged=True, max_applies=-(-k // solver_blockvec(k)) + 1)
            cr.synchronize()
            t1 = time.perf_counter()
            rr_ = cr.pca(ndim=k, allow_unconverged=True)
            cr.synchronize()
            wall_r = time.perf_counter() - t1
            ir = rr_["info"]
            sr = side_apply(cr, b, max(4, args.steps // 8))
            import numpy as np

            sd_r = ms_r[:, 1]
            out["pca_realistic"] = dict(n_pop=10, profile="rare-variant allele-frequency spectrum, missing calls concentrated in 5 % of the SNPs",
                                        wall_s=wall_r, converged=bool(ir["converged"]), block_applies=ir["block_applies"], cheap_applies=ir["cheap_applies"],
                                        cheap_slices=ir["cheap_slices"], vector_ops=ir["vector_ops"], restarts=ir["restarts"],
                                        seconds_apply=ir["seconds_apply"], seconds_apply_exact=ir["seconds_exact"], seconds_ortho=ir["seconds_ortho"],
                                        seconds_host=ir["seconds_host"], eigenvalue_1=float(rr_["d"][0]), eigenvalue_k=float(rr_["d"][-1]),
                                        max_rel_residual=ir["max_residual"],
                                        sd_min=float(np.nanmin(sd_r[sd_r > 1e-9])), sd_median=float(np.nanmedian(sd_r)), sd_max=float(np.nanmax(sd_r)),
                                        missing_call_path=sr.get("missing_call_path"), apply_ms_per_step=sr["ms_per_step"],
                                        apply_slowdown_vs_value=sr["ms_per_step"] / (elapsed / args.steps * 1e3))
            del rr_

    # ---- the same workload through the other exact path (fp64 MFMA kernels <-> int8 slices): timing and agreement ------
    # (the default run also carries the fp32-product twin: BASELINE configs[4]'s "fp32 accumulate" arithmetic, and with fp64 what
    # FPCA_ACCUM_AUTO drops to stage by stage when the exact-integer mode's buffers do not fit)
    for other in ((["fp64", "fp32"] if args.accum == "i8" else ["i8"]) if world == 1 and args.accum in ("fp64", "i8") and not args.no_alt else []):
        with fp.Context.synthetic(N, P_rank, snp_begin=snp_begin, n_pop=min(2 * k, 64), device=local_rank, accum=other) as c2:
            c2.set_total_snps(P_total)
            Y2 = torch.zeros_like(Y)
            for _ in range(max(2, args.warmup)):
                c2.apply_xxt_dev(B.data_ptr(), b, Y2.data_ptr())
            c2.synchronize()
            c2.profile_begin(steps_alt, sample_every=4 if steps_alt >= 8 else 1)
            t1 = time.perf_counter()
            for _ in range(steps_alt):
                c2.apply_xxt_dev(B.data_ptr(), b, Y2.data_ptr())
            c2.synchronize()
            el2 = time.perf_counter() - t1
            p2 = c2.profile_end(b)
            scale = float(torch.max(torch.abs(Y)).item())
            diff = float(torch.max(torch.abs(Y2 - Y)).item())
            ms2 = max(p2["ms_gemm_xt"], p2["ms_gemm_x"])
            if other == "i8":
                ops = (1 if c2.missing_mode(b) in (2, 3, 4) else 2) * flops_launch * 7
                rf = dict(bound="mfma", achieved=ops / (ms2 * 1e-3) / 1e12, peak=I8_MFMA_PEAK_TOPS, unit="TOP/s",
                          frac=ops / (ms2 * 1e-3) / 1e12 / I8_MFMA_PEAK_TOPS, peak_measured_pure_mfma_stream=mfma_stream_peak(11))
            elif other == "fp32":
                rf = dict(bound="mfma", achieved=flops_launch / (ms2 * 1e-3) / 1e12, peak=FP32_MFMA_PEAK_TFLOPS, unit="TFLOP/s",
                          frac=flops_launch / (ms2 * 1e-3) / 1e12 / FP32_MFMA_PEAK_TFLOPS,
                          note="fp32 products and short sums on v_mfma_f32_16x16x4, folded into fp64 accumulators every 4th chunk")
            else:
                rf = dict(bound="mfma", achieved=flops_launch / (ms2 * 1e-3) / 1e12, peak=FP64_MFMA_PEAK_TFLOPS, unit="TFLOP/s",
                          frac=flops_launch / (ms2 * 1e-3) / 1e12 / FP64_MFMA_PEAK_TFLOPS, peak_measured_pure_mfma_stream=mfma_stream_peak(0))
            alt = dict(accum=other, value=cells / args.steps * steps_alt / el2, unit="cells/s", steps=steps_alt, ms_per_step=el2 / steps_alt * 1e3, ms_xt_b=p2["ms_xt"],
                       ms_x_t=p2["ms_x"], max_abs_diff_between_modes_over_max_abs=diff / scale, roofline=rf)
            if not args.no_pca:
                t1 = time.perf_counter()
                r2 = c2.pca(ndim=k, allow_unconverged=True)
                c2.synchronize()
                alt["pca_wall_s"] = time.perf_counter() - t1
                alt["pca_block_applies"] = r2["info"]["block_applies"]
                if "pca" in out:
                    alt["pca_eigenvalue_max_rel_diff_between_modes"] = float(max(abs(a - c) / abs(c) for a, c in zip(r2["d"], r["d"])))
            out[{"fp64": "fp64_mode", "fp32": "fp32_mode"}.get(other, "exact_int8_mode")] = alt
            del Y2

    # ---- the whole program: `flashpca --bfile ... --ndim k --outload ... --outmeansd ...` on a fileset written to /tmp, the
    # reference's own process boundary (flashpca.cpp:589-604, 755-813): text parse, .bed upload, K1, solve, loadings, four
    # text files + loadings + mean/sd.  Phases as the CLI prints them under FPCA_TIMING=1; once with the .bed in the page
    # cache (just written), once after asking the kernel to drop it (fsync + posix_fadvise DONTNEED) ---------------------
    if world == 1 and not args.no_e2e and not under_profiler:  # (a profiler would follow the CLI child process too)
        try:
            size = args.e2e_size if args.e2e_size != "auto" else (e2e_size_auto(N, P_total) if args.workload == "cfg3" else "cfg2")
            out["e2e_cli"] = e2e_cli(fp, size, k, local_rank)
            out["e2e_cli"]["fileset"] = size + (": the headline fileset" if size == "cfg3" else "")
        except Exception as e:  # never lose the line over the side measurement
            out["e2e_cli"] = dict(error=str(e)[:300])

    # ---- CPU baseline: the oracle (restated reference path) on a bounded sample, rank 0, N=1 only -----------
    if world == 1 and not args.no_cpu_baseline:
        import numpy as np

        from oracle import oracle as O

        O.build()
        ncore = O.host_threads()  # logical CPUs this container may use at once (cgroup quota; 16 of 256 on the test boxes)
        P_s = min(P_rank, 1000)
        with fp.Context.synthetic(N, P_s, snp_begin=0, n_pop=min(2 * k, 64), device=local_rank) as sh:
            packed = sh.download_packed()
        od = O.OracleData(packed=packed, N=N, P=P_s, stand="binom2")
        bs = O.lib().orc_default_block_size(N, P_done, k, 0, 2048) or 1  # flashpca.cpp:636-686 on the FULL problem
        op = O.OracleOp(od, min(bs, P_s), nthreads=1)
        x = np.random.default_rng(0).standard_normal(N)
        op.perform_op(x)  # first visit computes mean/sd (not timed, like the GPU side's stats pass)
        nops, tc = 0, time.perf_counter()
        while True:
            op.perform_op(x)
            nops += 1
            if time.perf_counter() - tc > args.cpu_seconds * 0.5 or nops >= 5000:
                break
        tc = time.perf_counter() - tc
        out["cpu_baseline"] = dict(value=float(N) * P_s * nops / tc, unit="cells/s", cores=1, kind="port",
                                   sample="%d single-vector operator applications (decode->LUT->dense fp64 block->2 GEMV, "
                                          "svdwide.cpp:21-68) on the first %d SNPs x %d samples of the same synthetic matrix, "
                                          "block size %d, 1 thread (the shipped reference is single-threaded), %.1f s"
                                          % (nops, P_s, N, min(bs, P_s), tc),
                                   host_cores=os.cpu_count(), host_cores_usable=ncore)
        # generous variant (NOT what the shipped reference does): SNP sub-blocks dealt to all host
        # cores, every thread with its own dense block and partial y (oracle/fpca_oracle.c op_mt); a sample big enough to
        # give every core work
        P_a = min(P_rank, max(2000, 64 * ncore))
        with fp.Context.synthetic(N, P_a, snp_begin=0, n_pop=min(2 * k, 64), device=local_rank) as sh:
            packed_a = sh.download_packed()
        oda = O.OracleData(packed=packed_a, N=N, P=P_a, stand="binom2")
        opa = O.OracleOp(oda, min(bs, P_a), nthreads=ncore)
        opa.perform_op(x)
        na, ta = 0, time.perf_counter()
        while True:
            opa.perform_op(x)
            na += 1
            if time.perf_counter() - ta > args.cpu_seconds * 0.25 or na >= 5000:
                break
        ta = time.perf_counter() - ta
        out["cpu_baseline_allcores"] = dict(value=float(N) * P_a * na / ta, unit="cells/s", cores=ncore, kind="port",
                                            sample="%d operator applications on %d SNPs x %d samples, SNP sub-blocks over %d OpenMP "
                                                   "threads = the CPUs this container may use (cgroup quota) of the host's %d "
                                                   "(per-thread dense block + partial y), %.1f s" % (na, P_a, N, ncore, os.cpu_count() or 1, ta))
        # time to solution of the restated reference solver (Spectra-style IRLM, ncv = 2k+1, tol 1e-6, one operator
        # application per Lanczos column) on the WHOLE 50,000 x 20,000 matrix of BASELINE configs[1]: run for real on all
        # host cores (a 1-thread run takes ~ops x 0.45 s and would not fit the bench's time budget), next to this GPU's solve
        # of the same matrix
        if args.cpu_seconds >= 5:
            N2, P2 = WORKLOADS["cfg2"]["N"], WORKLOADS["cfg2"]["P"]
            with fp.Context.synthetic(N2, P2, snp_begin=0, n_pop=min(2 * k, 64), device=local_rank) as c2g:
                packed2 = c2g.download_packed()
                c2g.pca(ndim=k, allow_unconverged=True, max_applies=-(-k // solver_blockvec(k)) + 1)  # untimed set-up
                t1 = time.perf_counter()
                rg = c2g.pca(ndim=k)
                c2g.synchronize()
                gpu_wall = time.perf_counter() - t1
            od2 = O.OracleData(packed=packed2, N=N2, P=P2, stand="binom2")
            t1 = time.perf_counter()
            rc = O.pca_fast(od2, k, tol=1e-6, nthreads=ncore)
            cpu_wall = time.perf_counter() - t1
            op1 = O.OracleOp(O.OracleData(packed=packed2[:1000 * ((N2 + 3) // 4)], N=N2, P=1000, stand="binom2"), 1000, nthreads=1)
            x2 = np.random.default_rng(0).standard_normal(N2)
            op1.perform_op(x2)
            t1 = time.perf_counter()
            for _ in range(3):
                op1.perform_op(x2)
            s_per_op_1t = (time.perf_counter() - t1) / 3 * (P2 / 1000.0)
            out["cpu_baseline"]["full_solve_cfg2"] = dict(
                what="orc_pca_fast (restated Spectra IRLM, ncv=2k+1, tol=1e-6) on the full 50000 x 20000 synthetic matrix, k=%d" % k,
                total_ops=int(rc["nops"]), total_wall_s=cpu_wall, cores=ncore,
                cells_per_s=float(N2) * P2 * rc["nops"] / cpu_wall,
                one_thread_s_per_op=s_per_op_1t, one_thread_total_wall_s_extrapolated=s_per_op_1t * rc["nops"],
                gpu_wall_s_same_matrix=gpu_wall, gpu_block_applies=rg["info"]["block_applies"],
                max_rel_eigenvalue_diff_gpu_vs_cpu=float(np.max(np.abs(rg["d"] - rc["d"]) / np.abs(rc["d"]))))
            # "for cfg 3+ time >= 3 ops and extrapolate; state that".  The reference path's time to solution on THIS
            # workload = (operator applications of its IRLM) x (seconds per application at this size).  The applications are
            # those of the real solve above on the 50,000 x 20,000 matrix of the same generator (same 2k sub-populations, same k,
            # ncv, tol: the count depends on the spectrum's shape, not on the size -- 57 there, 51-58 on HapMap3); the seconds
            # per application are this run's measured samples (nops and na applications above) scaled to all P SNPs.
            s_op_1 = float(N) * P_done / out["cpu_baseline"]["value"]
            s_op_all = float(N) * P_done / out["cpu_baseline_allcores"]["value"]
            out["cpu_baseline"]["time_to_solution_extrapolated"] = dict(
                workload=args.workload, operator_applications_assumed=int(rc["nops"]),
                seconds_per_application_1_thread=s_op_1, seconds_per_application_all_cores=s_op_all, cores_all=ncore,
                total_s_1_thread=s_op_1 * rc["nops"], total_s_all_cores=s_op_all * rc["nops"],
                gpu_pca_wall_s=out.get("pca", {}).get("wall_s"),
                how="extrapolated: ops of the restated Spectra IRLM measured on the 50000 x 20000 matrix of the same generator x "
                    "seconds per operator application measured here on a %d-SNP (1 thread) / %d-SNP (%d threads) sample of this "
                    "matrix, scaled to its %d SNPs" % (P_s, P_a, ncore, P_done))

    # ---- the secondary results, as scalars inside `roofline` (the driver keeps that object whole; of the side blocks it keeps only
    # the names): fractions of the matrix peak of the fp64 / fp32 kernels, the cheap pass, the three solves, the CLI, the power sample
    def _g(d, *path):
        for q in path:
            d = d.get(q) if isinstance(d, dict) else None
        return d if isinstance(d, (int, float)) else None

    rl = out["roofline"]
    rl["fp64_frac"] = _g(out, "fp64_mode", "roofline", "frac") if args.accum != "fp64" else rl.get("frac")
    rl["fp64_ms_per_step"] = _g(out, "fp64_mode", "ms_per_step")
    rl["fp32_frac"] = _g(out, "fp32_mode", "roofline", "frac")
    rl["fp32_ms_per_step"] = _g(out, "fp32_mode", "ms_per_step")
    rl["cheap_pass_frac_mfma"] = _g(out, "cheap_pass", "frac_mfma")
    rl["cheap_pass_frac_hbm"] = _g(out, "cheap_pass", "frac_hbm")
    rl["cheap_pass_ms"] = _g(out, "cheap_pass", "ms_per_step")
    rl["cheap_pass_ms_kernel"] = _g(out, "cheap_pass", "ms_dominant_kernel")
    rl["pca_s"] = _g(out, "pca", "wall_s")
    rl["pca_hard_s"] = _g(out, "pca_hard_spectrum", "wall_s")
    rl["pca_hard_ortho_s"] = _g(out, "pca_hard_spectrum", "seconds_ortho")
    rl["pca_realistic_s"] = _g(out, "pca_realistic", "wall_s")
    rl["e2e_cli_warm_s"] = _g(out, "e2e_cli", "warm", "wall_s")
    rl["e2e_cli_cold_s"] = _g(out, "e2e_cli", "cold", "wall_s")
    rl["power_w"] = _g(rl, "power", "watts_median")
    rl["cheap_pass_power_w"] = _g(out, "cheap_pass", "power", "watts_median")
    rl["missing_2pct_slowdown"] = _g(out, "apply_at_missing_2pct", "slowdown_vs_value")
    rl["missing_2pct_cheap_pass_ms"] = _g(out, "apply_at_missing_2pct", "cheap_pass_ms_per_step")

    if rank == 0:
        # anything the C side buffered on stdout (RCCL prints a version banner there under NCCL_DEBUG=VERSION) goes out first:
        # the JSON line is the last line of rank 0's stdout
        import ctypes

        ctypes.CDLL(None).fflush(None)
        sys.stdout.flush()
        print(json.dumps(out), flush=True)
    if whole is not None:
        whole.close()
    ctx.close()
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
