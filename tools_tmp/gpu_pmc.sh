cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
B="python bench.py --no-cpu-baseline --genomes-per-gpu 400 --steps 1 --warmup 0"
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS --kernel-trace -d gpurun_out/pmcA -o a -- $B > /dev/null 2>&1
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAVES SQ_BUSY_CYCLES --kernel-trace -d gpurun_out/pmcB -o b -- $B > /dev/null 2>&1
rocprofv3 --pmc FETCH_SIZE --kernel-trace -d gpurun_out/pmcC -o c -- $B > /dev/null 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace -d gpurun_out/pmcD -o d -- $B > /dev/null 2>&1
for x in A B C D; do python profiles/tools/pmc_summary.py gpurun_out/pmc$x k_sweep; done > gpurun_out/pmc_sweep2.txt
find gpurun_out -name "*.db" -delete
grep "k_sweep<1\|k_sweep<0" gpurun_out/pmc_sweep2.txt
