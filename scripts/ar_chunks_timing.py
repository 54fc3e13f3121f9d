"""Cost of computing K3 in row chunks with the all-reduce on a second stream (nranks = 1: the all-reduce is a copy)."""
import json, os, subprocess, sys
os.environ.setdefault("FPCA_LIB", "testhooks")  # the environment switches this script drives exist only in the -DFPCA_TEST_HOOKS build
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CHILD = r'''
import sys, json
sys.path.insert(0, %r)
import flashpca_amd as fp
N, P = %d, %d
ctx = fp.Context.synthetic(N, P, n_pop=40, accum="i8")
if %d: ctx.comm_init_rank(1, 0, fp.Context.comm_unique_id())
r = ctx.bench_apply(b=32, steps=%d, warmup=3)
print("RESULT " + json.dumps({k: round(v, 4) for k, v in r.items() if k.startswith("ms")}))
'''
for name, N, P, steps in (("cfg2", 50000, 20000, 30), ("cfg3", 500000, 100000, 3)):
    for comm, env in [(0, {}), (1, dict(FPCA_AR_CHUNKS=1)), (1, {}), (1, dict(FPCA_AR_CHUNKS=4))]:
        e = dict(os.environ); e.update({k: str(v) for k, v in env.items()})
        out = subprocess.run([sys.executable, "-c", CHILD % (ROOT, N, P, comm, steps)], env=e, capture_output=True, text=True)
        print(name, "comm" if comm else "nocomm", env, ([l for l in out.stdout.splitlines() if l.startswith("RESULT")] or [out.stderr[-300:]])[-1], flush=True)
