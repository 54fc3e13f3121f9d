"""Write a synthetic PLINK fileset (prefix.bed/.bim/.fam) generated on the GPU (SURVEY.md 8d layout):
   python scripts/make_synth_bed.py N P prefix [n_pop]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np

import flashpca_amd as fp

N, P, prefix = int(sys.argv[1]), int(sys.argv[2]), sys.argv[3]
n_pop = int(sys.argv[4]) if len(sys.argv) > 4 else 40
with open(prefix + ".bed", "wb") as f:
    f.write(bytes([0x6C, 0x1B, 0x01]))
    step = max(1, (1 << 30) // ((N + 3) // 4))
    for j0 in range(0, P, step):
        pj = min(step, P - j0)
        with fp.Context.synthetic(N, pj, snp_begin=j0, n_pop=n_pop) as c:
            c.download_packed().tofile(f)
with open(prefix + ".fam", "w") as f:
    f.write("".join("F%d I%d 0 0 0 -9\n" % (i, i) for i in range(N)))
with open(prefix + ".bim", "w") as f:
    f.write("".join("1 rs%d 0 %d A C\n" % (j, j + 1) for j in range(P)))
print("wrote", prefix, N, P)
