#!/usr/bin/env python3
"""Calibration of FETCH_SIZE / WRITE_SIZE per access pattern.

  run    : python profiles/tools/calibrate.py run <n items> <window>      (under `rocprofv3 --pmc FETCH_SIZE --kernel-trace` and again
           under `--pmc WRITE_SIZE`: the library's k_cal_* kernels with known byte counts, pga_selftest_traffic)
  table  : python profiles/tools/calibrate.py table <dir FETCH pass> <dir WRITE pass> <n items>   -> factors: bytes / (counter x 1024)
"""
import ctypes as C, glob, json, os, sqlite3, sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
KNOWN = {"k_cal_read16": (16, 0), "k_cal_read4": (4, 0), "k_cal_gather4": (4, 0), "k_cal_gather16": (16, 0), "k_cal_write16": (0, 16), "k_cal_write4": (0, 4),
         "k_cal_scatter4": (0, 4), "k_cal_scatter16": (0, 16)}  # useful bytes per item: (read, written)


def counters(root, counter):
    db = sorted(glob.glob(os.path.join(root, "**", "*.db"), recursive=True))[0]
    cur = sqlite3.connect(db).cursor()
    cur.execute("SELECT * FROM pmc_events LIMIT 1")
    c = [d[0] for d in cur.description]
    kn = ([x for x in c if x in ("name", "kernel_name")] or [x for x in c if "name" in x and "counter" not in x])[0]
    cn = ([x for x in c if x in ("counter_name", "pmc_name")] or [x for x in c if "counter" in x and "name" in x])[0]
    vn = ([x for x in c if x in ("counter_value", "value")] or [x for x in c if "value" in x])[0]
    dn = ([x for x in c if x in ("dispatch_id", "event_id")] or [x for x in c if "dispatch" in x])[0]
    acc = {}
    for name, ctr, val, did in cur.execute("SELECT %s, %s, %s, %s FROM pmc_events" % (kn, cn, vn, dn)):
        if ctr == counter and name.startswith("k_cal_"):
            acc[(name.split("(")[0], did)] = acc.get((name.split("(")[0], did), 0.0) + (val or 0.0)
    out = {}
    for (name, did), v in acc.items():
        out.setdefault(name, []).append(v)
    return {k: sum(v) / len(v) for k, v in out.items()}


if sys.argv[1] == "run":
    lib = C.CDLL(os.path.join(ROOT, "pangene_amd", "lib", "libpangene_amd.so"))
    lib.pga_selftest_traffic.argtypes = [C.c_int64, C.c_int64]
    rc = lib.pga_selftest_traffic(int(sys.argv[2]), int(sys.argv[3]))
    print("pga_selftest_traffic:", rc)
    sys.exit(0 if rc == 0 else 1)
f, w, n = counters(sys.argv[2], "FETCH_SIZE"), counters(sys.argv[3], "WRITE_SIZE"), int(sys.argv[4])
res = {}
print("# %-18s %12s %12s %14s %14s %10s %10s" % ("pattern", "useful rd MB", "useful wr MB", "FETCH_SIZE KB", "WRITE_SIZE KB", "rd factor", "wr factor"))
for k, (rb, wb) in KNOWN.items():
    fk, wk = f.get(k, 0.0), w.get(k, 0.0)
    rf = (rb * n / 1024.0 / fk) if (rb and fk) else None
    wf = (wb * n / 1024.0 / wk) if (wb and wk) else None
    res[k] = {"fetch_kb": fk, "write_kb": wk, "read_factor": rf, "write_factor": wf}
    print("%-20s %12.1f %12.1f %14.1f %14.1f %10s %10s" % (k, rb * n / 1e6, wb * n / 1e6, fk, wk, "%.3f" % rf if rf else "-", "%.3f" % wf if wf else "-"))
print(json.dumps(res))
