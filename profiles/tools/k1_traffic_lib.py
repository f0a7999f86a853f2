"""per-dispatch counter values out of a rocprofv3 --pmc run (shared by k1_traffic.py and k2_traffic.py)"""
import glob, os, sqlite3


def per_dispatch(root, counter, pat):
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
        if pat in name and ctr == counter:
            acc[did] = acc.get(did, 0.0) + (val or 0.0)
    return sorted(acc.values())


