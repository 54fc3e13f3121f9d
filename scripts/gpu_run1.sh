set -x
mkdir -p gpurun_out
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -5
python bench.py --steps 20 --warmup 3 2> gpurun_out/bench1.err | tee gpurun_out/bench1.json
tail -5 gpurun_out/bench1.err
mkdir -p gpurun_out/cli && cd gpurun_out/cli && ../../flashpca_amd/_build/flashpca --bfile ../../tests/golden/hapmap3_data --ndim 10 --outload loadings.txt --outmeansd meansd.txt -v 2>&1 | tail -30; head -3 eigenvalues.txt pve.txt; head -2 eigenvectors.txt | cut -c1-120; head -2 pcs.txt | cut -c1-100; head -2 loadings.txt | cut -c1-100; head -2 meansd.txt
../../flashpca_amd/_build/flashpca --bfile ../../tests/golden/hapmap3_data --check --notime 2>&1 | tail -4
cd ../..
export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d gpurun_out/prof_r1 -o bench -- python bench.py --steps 20 --warmup 3 --no-cpu-baseline > gpurun_out/bench_prof.json 2> gpurun_out/bench_prof.err
ls -R gpurun_out/prof_r1 | head -30
