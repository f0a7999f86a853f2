#!/bin/bash
# Collects the evidence kept under profiles/ on a GPU box:  bash profiles/tools/collect.sh <tag>   (writes gpurun_out/<tag>_*)
set -u
cd "${GRAFT_REPO_ROOT:-.}"; export TMPDIR=/tmp
tag=${1:-rXX}; out=gpurun_out; mkdir -p $out
B="python bench.py --no-cpu-baseline --steps 2 --warmup 0"
timeout 1200 python -m pytest tests -m gpu -q 2>&1 | tail -n 4 > $out/${tag}_pytest_gpu.log
python bench.py > $out/${tag}_bench_default.json 2> $out/${tag}_bench_default.stderr
rocprofv3 --kernel-trace --stats -d $out/prof_stats -o s -- python bench.py --no-cpu-baseline --roofline-genomes 0 > $out/${tag}_bench_under_rocprof.json 2>/dev/null
python profiles/tools/kernel_stats.py $out/prof_stats > $out/${tag}_kernel_stats_bench_default.txt
rm -rf $out/prof_stats
rocprofv3 --kernel-trace --stats -d $out/prof_big -o s -- python bench.py --no-cpu-baseline --genomes-per-gpu 1250 --roofline-genomes 0 --steps 3 --warmup 1 > $out/${tag}_bench_big_shard_under_rocprof.json 2>/dev/null
python profiles/tools/kernel_stats.py $out/prof_big > $out/${tag}_kernel_stats_big_shard_1250x5k.txt
rm -rf $out/prof_big
rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $out/prof_fetch -o f -- $B > /dev/null 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace -d $out/prof_write -o w -- $B > /dev/null 2>&1
{ python profiles/tools/pmc_summary.py $out/prof_fetch k_sweep; python profiles/tools/pmc_summary.py $out/prof_write k_sweep; } > $out/${tag}_pmc_sweep_bench_default.txt
python profiles/tools/k1_traffic.py $out/prof_fetch $out/prof_write $out/${tag}_bench_default.json > $out/k1_pmc_traffic.json
find $out -name "*.db" -delete; rm -rf $out/prof_fetch $out/prof_write
cat $out/${tag}_pytest_gpu.log; cat $out/${tag}_bench_default.json; head -n 14 $out/${tag}_kernel_stats_bench_default.txt; tail -n 1 $out/${tag}_kernel_stats_bench_default.txt; head -n 12 $out/${tag}_kernel_stats_big_shard_1250x5k.txt; cat $out/k1_pmc_traffic.json
# extras: the configs[2] stand-in, a fresh-seed HIP-vs-oracle sweep, and the script's multi-rank flow with two ranks sharing the GPU
python bench.py --workload human47 --no-cpu-baseline --roofline-genomes 0 --steps 5 --warmup 2 > $out/${tag}_bench_human47.json 2>/dev/null
( echo "# python tests/fuzz_hip_vs_oracle.py 7400 24  (HIP vs oracle backend, both tie-order modes, 12 option variants)"; timeout 900 python tests/fuzz_hip_vs_oracle.py 7400 24 2>&1 | grep -v "^\[" | tail -n 5 ) > $out/${tag}_fuzz_sweep.txt
PANGENE_BENCH_ONE_GPU=1 timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29533 bench.py --gpus 2 --steps 3 --warmup 1 --no-cpu-baseline --roofline-genomes 0 2>/dev/null | tail -n 1 > $out/${tag}_bench_two_ranks_one_gpu_gloo.json
cut -c1-300 $out/${tag}_bench_human47.json; cat $out/${tag}_fuzz_sweep.txt; cut -c1-300 $out/${tag}_bench_two_ranks_one_gpu_gloo.json
