# copies the summaries of gpurun_out/<round>/ (written by gpu_profile_round.sh on the GPU box) into profiles/:  bash scripts/copy_profiles.sh r01
R=${1:-r01}; S=gpurun_out/$R; D=profiles
for f in bench_cfg2_n1 bench_cfg3_n1 bench_cfg5_n1 bench_cfg2_n1_fp64 bench_cfg3_n1_fp64 bench_cfg2_n1_fp32 bench_cfg3_n1_fp32 bench_cfg2_n1_i8x6 bench_cfg3_n1_i8x6; do
  [ -s $S/$f.json ] && cp $S/$f.json $D/${R}_$f.json
done
for wl in cfg2 cfg3; do for a in i8 fp64; do
  f=$(ls $S/trace_${wl}_$a/*kernel_stats.csv 2>/dev/null | head -1); [ -n "$f" ] && cp $f $D/${R}_bench_${wl}_${a}_kernel_stats.csv
done; done
cp $S/pmc_summary.json $D/${R}_pmc_summary.json; cp $S/pmc_summary_i8.json $D/${R}_pmc_summary_i8.json
cp $S/mfma_i8_microbench.txt $D/${R}_mfma_i8_microbench.txt; cp $S/mfma_f64_microbench.txt $D/${R}_mfma_f64_microbench.txt
[ -s $S/mx_fp4_fp6_probe.txt ] && cp $S/mx_fp4_fp6_probe.txt $D/${R}_mx_fp4_fp6_probe.txt
ls -la $D
