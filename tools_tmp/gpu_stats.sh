cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d gpurun_out/stats -o s -- python bench.py --no-cpu-baseline --steps 3 --warmup 1 > gpurun_out/stats_bench.json 2>/dev/null
python - <<'PY'
import sqlite3, glob
db = glob.glob('gpurun_out/stats/**/*.db', recursive=True)[0]
con = sqlite3.connect(db); cur = con.cursor()
rows = cur.execute("SELECT name, COUNT(*), SUM(duration), AVG(duration) FROM kernels GROUP BY name ORDER BY SUM(duration) DESC").fetchall()
tot = sum(r[2] for r in rows)
print("total kernel ns", tot, "launches", sum(r[1] for r in rows))
for r in rows[:28]: print("%-90s %6d %10.0f us total  %8.1f us avg  %5.1f%%" % (r[0][:90], r[1], r[2]/1e3, r[3]/1e3, 100*r[2]/tot))
PY
find gpurun_out -name "*.db" -delete
cat gpurun_out/stats_bench.json | head -c 600
