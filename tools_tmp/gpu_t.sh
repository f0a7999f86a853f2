cd $GRAFT_REPO_ROOT
timeout 600 python -m pytest tests -m gpu -x -q 2>&1 | tail -3
B="python bench.py --no-cpu-baseline --steps 1 --warmup 0"
for g in 400 100; do
PGA_SW_REPS=20 $B --genomes-per-gpu $g 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('genomes $g', round(d['roofline']['avg_launch_ms']/20*1000,1), 'us (20 x (sweep + slow-list kernel))', d['gfa_md5'])"; done
python bench.py --no-cpu-baseline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['roofline'])"
