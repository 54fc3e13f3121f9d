"""Kernel tuning harness: time K2 (xt_b) and K3 (x_t) separately through fpca_bench_apply under env-var variants."""
import json
import os
os.environ.setdefault("FPCA_LIB", "testhooks")  # the environment switches this script drives exist only in the -DFPCA_TEST_HOOKS build
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CHILD = r'''
import sys, json
sys.path.insert(0, %r)
import flashpca_amd as fp
N, P, b, steps = %d, %d, %d, %d
ctx = fp.Context.synthetic(N, P, n_pop=40)
r = ctx.bench_apply(b=b, steps=steps, warmup=2)
fl = 2.0 * N * P * b
print(json.dumps(dict(ms_xt=r["ms_xt"], ms_x=r["ms_x"], tf_xt=fl / r["ms_xt"] / 1e9, tf_x=fl / r["ms_x"] / 1e9)))
'''


def run(env, N, P, b, steps):
    e = dict(os.environ)
    e.update({k: str(v) for k, v in env.items()})
    out = subprocess.run([sys.executable, "-c", CHILD % (ROOT, N, P, b, steps)], env=e, capture_output=True, text=True)
    try:
        return json.loads(out.stdout.strip().splitlines()[-1])
    except Exception:
        return dict(error=out.stderr[-400:])


if __name__ == "__main__":
    configs = [("cfg2", 50000, 20000, 32, 20), ("cfg3", 500000, 100000, 32, 3)]
    variants = [dict(), dict(FPCA_XT_VARIANT=1)]
    for extra in sys.argv[1:]:
        variants.append(dict(kv.split("=") for kv in extra.split(",")))
    for name, N, P, b, steps in configs:
        for v in variants:
            r = run(v, N, P, b, steps)
            print(name, v, json.dumps(r))
