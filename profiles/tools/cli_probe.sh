# the whole command in a fresh process, a few times, by parts (PANGENE_TIMING) and by the shell's clock:  bash profiles/tools/cli_probe.sh
set -u
cd "${GRAFT_REPO_ROOT:-.}"
python - <<'PY'
import sys; sys.path.insert(0,'.')
from pangene_amd import synth
synth.write_files_parallel("bact", "/tmp/c1", G=100, P=5000, seed=1)
PY
TIMEFORMAT="wall %R s user %U s sys %S s"
for i in 1 2 3 4; do
  { time PANGENE_TIMING=1 pangene_amd/bin/pangene /tmp/c1/* 2>/tmp/err >/dev/null ; } 2>&1; grep "cli_timing" /tmp/err
done
echo "--- with PANGENE_HOST_THREADS=4"
for i in 1 2; do
  { time PANGENE_HOST_THREADS=4 PANGENE_TIMING=1 pangene_amd/bin/pangene /tmp/c1/* 2>/tmp/err >/dev/null ; } 2>&1; grep "cli_timing" /tmp/err
done
echo "--- cpu.stat of the control group"; head -8 /sys/fs/cgroup/cpu.stat
