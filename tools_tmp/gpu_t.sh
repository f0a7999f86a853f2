cd $GRAFT_REPO_ROOT
PANGENE_BENCH_ONE_GPU=1 timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29555 bench.py --gpus 2 --genomes-per-gpu 50 --steps 2 --warmup 1 2>gpurun_out/b2.err | tee gpurun_out/b2.json | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('world2', d['n_gpus'], d['ms_per_step'], d['gfa_sl_md5'], d['config']['workload'][-80:])"
tail -5 gpurun_out/b2.err
python bench.py --no-cpu-baseline --genomes-per-gpu 100 --steps 2 --warmup 1 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('world1', d['n_gpus'], d['ms_per_step'], d['gfa_sl_md5'], d['config']['workload'][-80:])"
