#!/bin/bash
# the narrow-tile / occupancy matrix of scripts/r4_narrow_probe.py
export FPCA_LIB=testhooks
out=gpurun_out/r4_narrow_probe.txt
: > $out
run() { env "$@" python scripts/r4_narrow_probe.py $S $B >> $out 2>&1; }
for S in 4 5 3; do
  B=16
  run A=1
  run FPCA_I8_MT4=1
  run FPCA_I8_LDS_PAD=50000          # 2-tile kernel: 85 KB per workgroup -> one per CU
  run FPCA_I8_LDS_PAD=20000          # 55 KB -> two per CU
done
S=7; B=16
run A=1
run FPCA_I8_LDS_PAD=30000            # headline kernel (61 KB): 91 KB -> strictly one workgroup per CU
S=4; B=32
run A=1
run FPCA_I8_LDS_PAD=30000
cat $out
