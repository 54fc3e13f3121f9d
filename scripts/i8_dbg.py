import json, os, subprocess, sys
os.environ.setdefault("FPCA_LIB", "testhooks")  # the environment switches this script drives exist only in the -DFPCA_TEST_HOOKS build
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CHILD = r'''
import sys, json
sys.path.insert(0, %r)
import flashpca_amd as fp
N, P = %d, %d
ctx = fp.Context.synthetic(N, P, n_pop=40, accum="i8")
r = ctx.bench_apply(b=32, steps=%d, warmup=2)
r = ctx.bench_apply(b=32, steps=%d, warmup=2)
print(json.dumps(dict(ms_xt=round(r["ms_xt"], 4), ms_x=round(r["ms_x"], 4))))
'''
for name, N, P, steps in (("cfg2", 50000, 20000, 20), ("cfg3", 500000, 100000, 3)):
    for env in [dict(), dict(FPCA_I8_SPLITS=1), dict(FPCA_I8_SPLITS=3), dict(FPCA_I8_SPLITS=5), dict()]:
        e = dict(os.environ); e.update({k: str(v) for k, v in env.items()})
        out = subprocess.run([sys.executable, "-c", CHILD % (ROOT, N, P, steps, steps)], env=e, capture_output=True, text=True)
        print(name, env, out.stdout.strip().splitlines()[-1] if out.stdout.strip() else out.stderr[-300:], flush=True)
