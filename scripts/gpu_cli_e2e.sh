# end-to-end CLI run on a synthetic fileset in /tmp/e2e: bash scripts/gpu_cli_e2e.sh [N] [P] [runs]   (FPCA_TIMING=1 for the phases)
mkdir -p /tmp/e2e && cd /tmp/e2e
N=${1:-200000}; P=${2:-50000}; RUNS=${3:-2}
( time python $GRAFT_REPO_ROOT/scripts/make_synth_bed.py $N $P /tmp/e2e/syn ) 2>&1 | tail -4
ls -la /tmp/e2e/
for i in $(seq $RUNS); do
  echo "== run $i"
  ( time $GRAFT_REPO_ROOT/flashpca_amd/_build/flashpca --bfile /tmp/e2e/syn --ndim 20 --outload load.txt --outmeansd ms.txt -v ) 2>&1 | grep -v "apply  " | tail -40
done
ls -la /tmp/e2e/*.txt
