"""Host-side probe (run on the GPU box): how the oracle's all-cores operator scales with the thread count there, and what
limits the container imposes (cgroup cpu quota, affinity)."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import oracle as O

print("cpu_count", os.cpu_count(), "affinity", len(os.sched_getaffinity(0)))
for f in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us", "/sys/fs/cgroup/cpu/cpu.cfs_period_us", "/sys/fs/cgroup/cpuset.cpus.effective"):
    try:
        print(f, open(f).read().strip())
    except Exception as e:
        print(f, "n/a")
os.system("lscpu | egrep 'Model name|Socket|NUMA node|Thread|Core' | head -12")
rng = np.random.default_rng(1)
for N, P in ((50000, 20000), (500000, 4096)):
    packed = rng.integers(0, 256, size=(P, (N + 3) // 4), dtype=np.uint8)
    x = rng.standard_normal(N)
    for nt in (1, 4, 16, 32, 64, 128, 256):
        if nt > (os.cpu_count() or 1):
            break
        od = O.OracleData(packed=packed, N=N, P=P, stand="binom2")
        op = O.OracleOp(od, 500, nthreads=nt)
        op.perform_op(x)
        t = time.time()
        reps = 1 if nt == 1 else 3
        for _ in range(reps):
            op.perform_op(x)
        dt = (time.time() - t) / reps
        print("N %d P %d threads %3d  %.3f s/op  %.2f Gcells/s" % (N, P, nt, dt, N * P / dt / 1e9), flush=True)
