cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -3
PANGENE_VTX_TIMING=1 python bench.py --no-cpu-baseline 2>gpurun_out/vt.err | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['gfa_md5'], d['host_phases_ms_per_step'])"
grep vtx gpurun_out/vt.err | tail -2
python bench.py --no-cpu-baseline --genomes-per-gpu 800 --steps 2 --warmup 1 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['gfa_md5'], d['host_phases_ms_per_step'])"
