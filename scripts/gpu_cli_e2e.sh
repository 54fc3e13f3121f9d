mkdir -p /tmp/e2e && cd /tmp/e2e
N=${1:-200000}; P=${2:-50000}
( time python $GRAFT_REPO_ROOT/scripts/make_synth_bed.py $N $P /tmp/e2e/syn ) 2>&1 | tail -4
ls -la /tmp/e2e/
( time $GRAFT_REPO_ROOT/flashpca_amd/_build/flashpca --bfile /tmp/e2e/syn --ndim 20 --outload load.txt --outmeansd ms.txt -v ) 2>&1 | grep -v "^\[fpca\]" | tail -25
ls -la /tmp/e2e/*.txt
