#!/bin/bash
# Collects the evidence kept under profiles/ on a GPU box:  bash profiles/tools/collect.sh <tag> <stage ...>   (writes gpurun_out/<tag>_*)
#   stage tests   the whole `pytest -m gpu` suite (log tail + wall time) and the default `python bench.py` line
#   stage stats   rocprofv3 --kernel-trace --stats of the bench workload, the 12 M-hit shard and the sharded route (forced exchange)
#   stage pmc     FETCH_SIZE / WRITE_SIZE passes: K1 per shard size and flavour (k1_pmc_traffic.json), every kernel of the 12 M-hit shard
#   stage pmc2    FETCH_SIZE / WRITE_SIZE of the arc round's kernels on the 12 M-hit shard (k2_pmc_traffic.json for bench.py's roofline.k2.*.frac_by_counters) + the per-kernel table
#   stage extra   configs[2] stand-in, fresh-seed HIP-vs-oracle sweep, two ranks sharing the GPU over gloo
#   stage human   rocprofv3 --kernel-trace --stats with the human-shaped leg left in
#   stage cal     calibration of FETCH_SIZE / WRITE_SIZE with kernels of known byte counts (profiles/tools/calibrate.py)
#   stage full    BASELINE configs[3] / configs[4] at their full stated size on this one GPU (bench.py --workload config3 | config4)
#   stage full4   configs[4] at full size under the profiler: kernel stats with the pass timeline, FETCH_SIZE / WRITE_SIZE per kernel, SQ counters of K1
#   stage k1full4  only K1's FETCH_SIZE / WRITE_SIZE at configs[4]'s full size (the entry of k1_pmc_traffic.json for that shard, after an edit of k_sweep.hpp)
#   stage full3   configs[3] at full size under the profiler (kernel stats with the pass timeline)
#   stage ranks8  eight ranks sharing the one GPU over gloo on the configs[3] / configs[4] per-GPU shards (pytest), with the wall time
set -u
cd "${GRAFT_REPO_ROOT:-.}"; export TMPDIR=/tmp
tag=${1:-rXX}; shift; out=gpurun_out; mkdir -p $out
Q="--no-cpu-baseline --roofline-genomes 0 --human-genomes 0 --no-extra-legs --cold-sets 0"
for stage in "$@"; do case $stage in
tests)
	t0=$(date +%s)
	timeout 1500 python -m pytest tests -m gpu -q 2>&1 | tail -n 6 > $out/${tag}_pytest_gpu.log
	echo "# wall time of the suite: $(( $(date +%s) - t0 )) s" >> $out/${tag}_pytest_gpu.log
	PANGENE_TIMING=1 python bench.py > $out/${tag}_bench_default.json 2> $out/${tag}_bench_default.stderr  # (stderr: the parts of every pass, the allocation / upload split of every context)
	cat $out/${tag}_pytest_gpu.log; cat $out/${tag}_bench_default.json;;
stats)
	rocprofv3 --kernel-trace --stats -d $out/prof_stats -o s -- python bench.py $Q > $out/${tag}_bench_under_rocprof.json 2>/dev/null
	python profiles/tools/kernel_stats.py $out/prof_stats > $out/${tag}_kernel_stats_bench_default.txt; rm -rf $out/prof_stats
	rocprofv3 --kernel-trace --stats -d $out/prof_big -o s -- python bench.py $Q --genomes-per-gpu 1250 --steps 3 --warmup 1 > $out/${tag}_bench_big_shard_under_rocprof.json 2>/dev/null
	python profiles/tools/kernel_stats.py $out/prof_big > $out/${tag}_kernel_stats_big_shard_1250x5k.txt; rm -rf $out/prof_big
	MASTER_ADDR=127.0.0.1 MASTER_PORT=29533 RANK=0 WORLD_SIZE=1 LOCAL_RANK=0 PANGENE_FORCE_EXCHANGE=1 rocprofv3 --kernel-trace --stats -d $out/prof_x -o s -- python bench.py --leg steps-only --steps 5 --warmup 2 > $out/${tag}_bench_forced_exchange_under_rocprof.json 2>/dev/null
	python profiles/tools/kernel_stats.py $out/prof_x > $out/${tag}_kernel_stats_forced_exchange.txt; rm -rf $out/prof_x
	head -n 14 $out/${tag}_kernel_stats_bench_default.txt; tail -n 3 $out/${tag}_kernel_stats_bench_default.txt; head -n 12 $out/${tag}_kernel_stats_big_shard_1250x5k.txt; tail -n 3 $out/${tag}_kernel_stats_forced_exchange.txt;;
pmc1|pmc)
	B="python bench.py --no-cpu-baseline --no-extra-legs --cold-sets 0 --steps 1 --warmup 0"
	rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $out/prof_fetch -o f -- $B > /dev/null 2>&1
	rocprofv3 --pmc WRITE_SIZE --kernel-trace -d $out/prof_write -o w -- $B > /dev/null 2>&1
	J=$out/${tag}_bench_default.json; [ -f $J ] || J=profiles/${tag}_bench_default.json  # (a fresh box only has what the repository holds)
	python profiles/tools/k1_traffic.py $out/prof_fetch $out/prof_write $J > $out/k1_pmc_traffic.json
	rm -rf $out/prof_fetch $out/prof_write; cat $out/k1_pmc_traffic.json
	[ $stage = pmc1 ] && continue
	B="python bench.py $Q --genomes-per-gpu 1250 --steps 1 --warmup 0"
	rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $out/prof_fetch -o f -- $B > /dev/null 2>&1
	rocprofv3 --pmc WRITE_SIZE --kernel-trace -d $out/prof_write -o w -- $B > /dev/null 2>&1
	python profiles/tools/pmc_traffic.py $out/prof_fetch $out/prof_write 12121149 > $out/${tag}_pmc_traffic_big_shard_1250x5k.txt
	rm -rf $out/prof_fetch $out/prof_write
	cat $out/k1_pmc_traffic.json; head -n 30 $out/${tag}_pmc_traffic_big_shard_1250x5k.txt;;
pmc2)
	B="python bench.py $Q --genomes-per-gpu 1250 --steps 1 --warmup 0"
	rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $out/prof_fetch -o f -- $B > /dev/null 2>&1
	rocprofv3 --pmc WRITE_SIZE --kernel-trace -d $out/prof_write -o w -- $B > /dev/null 2>&1
	python profiles/tools/pmc_traffic.py $out/prof_fetch $out/prof_write 12121149 > $out/${tag}_pmc_traffic_big_shard_1250x5k.txt
	python profiles/tools/k2_traffic.py $out/prof_fetch $out/prof_write 12121149 > $out/k2_pmc_traffic.json
	rm -rf $out/prof_fetch $out/prof_write
	cat $out/k2_pmc_traffic.json; head -n 30 $out/${tag}_pmc_traffic_big_shard_1250x5k.txt;;
extra)
	python bench.py --workload human47 --no-cpu-baseline --roofline-genomes 0 --human-genomes 0 --no-extra-legs --steps 5 --warmup 2 > $out/${tag}_bench_human47.json 2>/dev/null
	( echo "# python tests/fuzz_hip_vs_oracle.py 7600 ${FUZZ_SEEDS:-16}  (HIP vs oracle backend on fresh seeds: fuzz / bacterial / human-shaped / mutated sets, both tie-order modes, 12 option variants)"; timeout 900 python tests/fuzz_hip_vs_oracle.py 7600 ${FUZZ_SEEDS:-16} 2>&1 | grep -v "^\[" | tail -n 5 ) > $out/${tag}_fuzz_sweep.txt
	PANGENE_BENCH_ONE_GPU=1 timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29533 bench.py --gpus 2 --steps 3 --warmup 2 --no-cpu-baseline --roofline-genomes 0 --human-genomes 0 --no-extra-legs 2>/dev/null | tail -n 1 > $out/${tag}_bench_two_ranks_one_gpu_gloo.json
	cut -c1-300 $out/${tag}_bench_human47.json; cat $out/${tag}_fuzz_sweep.txt; cut -c1-400 $out/${tag}_bench_two_ranks_one_gpu_gloo.json;;
human)
	rocprofv3 --kernel-trace --stats -d $out/prof_h -o s -- python bench.py --no-cpu-baseline --roofline-genomes 0 --no-extra-legs --cold-sets 0 --steps 2 --warmup 1 > $out/${tag}_bench_human_shard_under_rocprof.json 2>/dev/null
	python profiles/tools/kernel_stats.py $out/prof_h > $out/${tag}_kernel_stats_human_shard_500x20k.txt; rm -rf $out/prof_h
	head -n 16 $out/${tag}_kernel_stats_human_shard_500x20k.txt;;
cal)
	N=33554432
	for w in 16384 0; do
		rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $out/cal_f -o f -- python profiles/tools/calibrate.py run $N $w > /dev/null 2>&1
		rocprofv3 --pmc WRITE_SIZE --kernel-trace -d $out/cal_w -o w -- python profiles/tools/calibrate.py run $N $w > /dev/null 2>&1
		( echo "# pga_selftest_traffic($N items, window $w)"; python profiles/tools/calibrate.py table $out/cal_f $out/cal_w $N | head -n 10 ) > $out/${tag}_calibration_window_$w.txt
		rm -rf $out/cal_f $out/cal_w; cat $out/${tag}_calibration_window_$w.txt
	done;;
full)
	for wl in ${FULL_WORKLOADS:-config4 config3}; do
		PANGENE_TIMING=1 timeout 1500 python bench.py --workload $wl --steps 3 --warmup 1 > $out/${tag}_bench_${wl}_full_size.json 2> $out/${tag}_bench_${wl}_full_size.stderr
		python - <<PY
import json
b = json.loads([l for l in open("$out/${tag}_bench_${wl}_full_size.json") if l.startswith("{")][-1])
print("$wl", json.dumps(b.get("full_size"))[:1500])
PY
	done;;
full4)
	B="python bench.py --workload config4 --steps 3 --warmup 1"
	PANGENE_TIMING=1 rocprofv3 --kernel-trace --stats -d $out/prof_c4 -o s -- $B > $out/${tag}_bench_config4_under_rocprof.json 2> $out/${tag}_bench_config4_under_rocprof.stderr
	python profiles/tools/kernel_stats.py $out/prof_c4 > $out/${tag}_kernel_stats_config4_full_size.txt; rm -rf $out/prof_c4
	B1="python bench.py --workload config4 --steps 1 --warmup 0"
	rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $out/prof_fetch -o f -- $B1 > /dev/null 2>&1
	rocprofv3 --pmc WRITE_SIZE --kernel-trace -d $out/prof_write -o w -- $B1 > /dev/null 2>&1
	python profiles/tools/pmc_traffic.py $out/prof_fetch $out/prof_write 21920400 > $out/${tag}_pmc_traffic_config4_full_size.txt
	python profiles/tools/k1_traffic.py $out/prof_fetch $out/prof_write $out/${tag}_bench_config4_under_rocprof.json > $out/k1_pmc_traffic_config4.json 2>/dev/null
	rm -rf $out/prof_fetch $out/prof_write
	rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY --kernel-trace -d $out/prof_sq -o q -- $B1 > /dev/null 2>&1
	python profiles/tools/pmc_summary.py $out/prof_sq "k_sweep<3" > $out/${tag}_pmc_sq_k1_config4.txt 2>&1; rm -rf $out/prof_sq
	rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY --kernel-trace -d $out/prof_sq -o q -- $B1 > /dev/null 2>&1
	python profiles/tools/pmc_summary.py $out/prof_sq "k_sweep<3" | grep -v "^dur" >> $out/${tag}_pmc_sq_k1_config4.txt 2>&1; rm -rf $out/prof_sq
	head -n 30 $out/${tag}_kernel_stats_config4_full_size.txt | cut -c1-160; head -n 12 $out/${tag}_pmc_traffic_config4_full_size.txt | cut -c1-160; cat $out/${tag}_pmc_sq_k1_config4.txt;;
k1full4)
	B1="python bench.py --workload config4 --steps 1 --warmup 0"
	$B1 > $out/${tag}_bench_config4_one_step.json 2>/dev/null
	rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $out/prof_fetch -o f -- $B1 > /dev/null 2>&1
	rocprofv3 --pmc WRITE_SIZE --kernel-trace -d $out/prof_write -o w -- $B1 > /dev/null 2>&1
	python profiles/tools/k1_traffic.py $out/prof_fetch $out/prof_write $out/${tag}_bench_config4_one_step.json > $out/k1_pmc_traffic_config4.json 2>/dev/null
	rm -rf $out/prof_fetch $out/prof_write; cat $out/k1_pmc_traffic_config4.json;;
full3)
	PANGENE_TIMING=1 rocprofv3 --kernel-trace --stats -d $out/prof_c3 -o s -- python bench.py --workload config3 --steps 2 --warmup 1 > $out/${tag}_bench_config3_under_rocprof.json 2> $out/${tag}_bench_config3_under_rocprof.stderr
	python profiles/tools/kernel_stats.py $out/prof_c3 > $out/${tag}_kernel_stats_config3_full_size.txt; rm -rf $out/prof_c3
	head -n 24 $out/${tag}_kernel_stats_config3_full_size.txt | cut -c1-160;;
ranks8)
	( time timeout 1500 python -m pytest tests/test_hip_parity.py -m gpu -q -k "eight_ranks_at_size" ) > $out/${tag}_pytest_eight_ranks_at_size.log 2>&1; tail -n 6 $out/${tag}_pytest_eight_ranks_at_size.log;;
esac; done
