#!/bin/bash
# Collects the evidence kept under profiles/ on a GPU box:  bash profiles/tools/collect.sh <tag>   (writes gpurun_out/<tag>_*)
set -u
cd "${GRAFT_REPO_ROOT:-.}"; export TMPDIR=/tmp
tag=${1:-rXX}; out=gpurun_out; mkdir -p $out
B="python bench.py --no-cpu-baseline"
timeout 900 python -m pytest tests -m gpu -q 2>&1 | tail -4 > $out/${tag}_pytest_gpu.log
python bench.py > $out/${tag}_bench_default.json 2> $out/${tag}_bench_default.stderr
rocprofv3 --kernel-trace --stats -d $out/prof_stats -o s -- $B > $out/${tag}_bench_under_rocprof.json 2>/dev/null
python profiles/tools/kernel_stats.py $out/prof_stats > $out/${tag}_kernel_stats_bench_default.txt
rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $out/prof_fetch -o f -- $B > /dev/null 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace -d $out/prof_write -o w -- $B > /dev/null 2>&1
{ python profiles/tools/pmc_summary.py $out/prof_fetch k_sweep; python profiles/tools/pmc_summary.py $out/prof_write k_sweep; } > $out/${tag}_pmc_sweep_bench_default.txt
python profiles/tools/k1_traffic.py $out/${tag}_pmc_sweep_bench_default.txt $out/${tag}_bench_default.json > $out/k1_pmc_traffic.json
$B --genomes-per-gpu 1250 --steps 3 --warmup 1 > $out/${tag}_bench_big_shard_1250x5k.json 2>/dev/null
find $out -name "*.db" -delete; rm -rf $out/prof_stats $out/prof_fetch $out/prof_write
cat $out/${tag}_pytest_gpu.log; cat $out/${tag}_bench_default.json; head -12 $out/${tag}_kernel_stats_bench_default.txt; cat $out/k1_pmc_traffic.json
