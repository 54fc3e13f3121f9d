"""bench.py is what the driver runs: its one JSON line must keep its contract -- the metric of BASELINE.json, value = cells over
the timed region, `roofline` and `cpu_baseline` objects -- and, with several ranks, validate itself.  Run here on the smoke-size
workload (seconds), single rank and as two ranks sharing the one GPU over gloo (the plumbing path of a multi-GPU run; the native
RCCL communicator refuses two ranks on one device, so the torch.distributed hook transport carries it)."""
import json
import os
import socket
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pytestmark = pytest.mark.gpu


def last_json_line(text):
    lines = [l for l in text.splitlines() if l.startswith("{")]
    assert lines, text[-2000:]
    return json.loads(lines[-1])


def test_bench_line_contract_single_rank():
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--workload", "tiny", "--steps", "20", "--warmup", "3", "--cpu-seconds", "2",
                        "--no-e2e", "--traffic", "none"], capture_output=True, text=True, timeout=900, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-2000:]
    assert r.stdout.strip().splitlines()[-1].startswith("{")  # the JSON line is the LAST line of stdout
    d = last_json_line(r.stdout)
    base = json.load(open(os.path.join(ROOT, "BASELINE.json")))
    norm = lambda t: "".join(t.replace("\u00d7", "x").split()).lower()  # (BASELINE.json writes the multiplication sign, the line ASCII)
    assert norm(d["metric"]) in norm(base["metric"]) and d["unit"] == "cells/s" and d["n_gpus"] == 1 and d["steps"] == 20 and d["warmup"] == 3
    assert d["higher_is_better"] is True and d["vs_baseline"] is None and d["data"] == "synthetic" and d["scaling"] in ("weak", "strong")
    cfg = d["config"]
    assert "workload" in cfg and cfg["blockvec"] == cfg["solver_default_blockvec"] == 16 and "model" not in cfg
    # value = samples x SNPs x columns x steps / (ms_per_step x steps)
    cells = cfg["samples"] * cfg["snps_total"] * cfg["blockvec"] * d["steps"]
    assert abs(d["value"] - cells / (d["ms_per_step"] * 1e-3 * d["steps"])) <= 1e-6 * d["value"]
    rf = d["roofline"]
    for key in ("bound", "achieved", "peak", "unit", "frac", "traffic", "ms_dominant_kernel", "ms_per_step_instrumented", "duration_source"):
        assert key in rf, key
    assert rf["bound"] in ("hbm", "mfma") and abs(rf["frac"] - rf["achieved"] / rf["peak"]) < 1e-9
    # the instrumented steps' kernels cannot take longer than the instrumented steps
    assert rf["ms_gemm_kernel_xt_b"] + rf["ms_gemm_kernel_x_t"] <= rf["ms_per_step_instrumented"] * 1.001
    cb = d["cpu_baseline"]
    assert cb["kind"] in ("port", "reference") and cb["cores"] == 1 and cb["value"] > 0 and "sample" in cb
    for leg in ("pca", "pca_hard_spectrum", "pca_realistic"):
        assert d[leg]["converged"] is True and d[leg]["wall_s"] > 0, leg
    assert d["pca_realistic"]["missing_call_path"].startswith("hybrid")
    # the secondary results travel as scalars inside `roofline` (the object the driver keeps whole), equal to their side blocks
    for key in ("fp64_frac", "fp32_frac", "cheap_pass_frac_mfma", "cheap_pass_frac_hbm", "cheap_pass_ms", "pca_s", "pca_realistic_s", "pca_hard_s",
                "e2e_cli_warm_s", "power_w"):
        assert key in rf, key
    assert rf["pca_s"] == d["pca"]["wall_s"] and rf["pca_hard_s"] == d["pca_hard_spectrum"]["wall_s"] and rf["pca_realistic_s"] == d["pca_realistic"]["wall_s"]
    assert rf["fp64_frac"] == d["fp64_mode"]["roofline"]["frac"] and rf["fp32_frac"] == d["fp32_mode"]["roofline"]["frac"]
    assert rf["e2e_cli_warm_s"] is None  # (--no-e2e here; the driver's default run carries it)
    if "cheap_pass" in d and "ms_per_step" in d["cheap_pass"]:
        assert rf["cheap_pass_ms"] == d["cheap_pass"]["ms_per_step"]


def test_bench_two_ranks_on_one_gpu_validates_itself():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    env = dict(os.environ, FPCA_BENCH_BACKEND="gloo", FPCA_BENCH_ONE_GPU="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                        "--master-port", str(port), os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "5", "--warmup", "2", "--workload", "tiny"],
                       capture_output=True, text=True, timeout=900, cwd=ROOT, env=env)
    assert r.returncode == 0, (r.stdout[-1500:], r.stderr[-1500:])
    d = last_json_line(r.stdout)
    assert d["n_gpus"] == 2
    v = d["multi_rank_validation"]
    assert v["passed"] is True and v["operator_ok"] and v["solver_ok"]
    assert v["multi_rank_parity"] < 1e-12 and v["max_deviation_between_ranks"] == 0.0
    assert v["eigenvalues_rowsharded_vs_one_context"] < 1e-9 and v["eigenvalues_replicated_vs_one_context"] < 1e-9
    assert d["pca"]["solver"].startswith("row-sharded") and "pca_rowsharded_error" not in d
