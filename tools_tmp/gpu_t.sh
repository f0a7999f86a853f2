cd $GRAFT_REPO_ROOT
timeout 600 python -m pytest tests -m gpu -x -q -k "exchange or merge or primitives" 2>&1 | tail -5
PANGENE_FORCE_EXCHANGE=1 python bench.py --no-cpu-baseline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['config']['exchange'], d['ms_per_step'], d['gfa_md5'], d['host_phases_ms_per_step'])"
