cd $GRAFT_REPO_ROOT
B="python bench.py --no-cpu-baseline --steps 1 --warmup 0"
$B --genomes-per-gpu 400 2>&1 >/dev/null | grep "profile" | sort | uniq -c | sort -rn | head -4
$B --genomes-per-gpu 100 2>&1 >/dev/null | grep "profile" | head -3
timeout 600 python -m pytest tests -m gpu -x -q 2>&1 | tail -3
