import json, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CHILD = r'''
import sys, json
sys.path.insert(0, %r)
import flashpca_amd as fp
ctx = fp.Context.synthetic(500000, 100000, n_pop=40, accum="i8")
r = ctx.bench_apply(b=32, steps=2, warmup=1)
print(json.dumps(dict(ms_xt=r["ms_xt"], ms_x=r["ms_x"])))
''' % ROOT
for env in [dict(), dict(FPCA_I8_DBG=1), dict(FPCA_I8_DBG=3), dict(FPCA_I8_DBG=7), dict(FPCA_I8_SPLITS=1), dict(FPCA_I8_SPLITS=3), dict(FPCA_I8_SPLITS=4), dict(FPCA_I8_SPLITS=5), dict(FPCA_I8_SPLITS=8)] :
    e = dict(os.environ); e.update({k: str(v) for k, v in env.items()})
    out = subprocess.run([sys.executable, "-c", CHILD], env=e, capture_output=True, text=True)
    print(env, out.stdout.strip().splitlines()[-1] if out.stdout.strip() else out.stderr[-300:], flush=True)
