import sys, os, time, ctypes as C, numpy as np, hashlib
sys.path.insert(0, os.getcwd())
from pangene_amd import capi, synth
lib = capi.load(); ora = capi.load(oracle_host=True)
raw = C.CDLL(capi.LIB_HIP)
# 1. primitives
rng = np.random.default_rng(1)
ok = True
for n, nb in [(1, 8), (63, 16), (2048, 24), (5000, 40), (1_000_003, 37)]:
    k = rng.integers(0, 1 << nb, size=n, dtype=np.uint64); k[: n // 3] &= np.uint64(0xff)  # many ties
    v = np.arange(n, dtype=np.uint32)
    k2, v2 = k.copy(), v.copy()
    rc = raw.pga_selftest_sort(k2.ctypes.data_as(C.c_void_p), v2.ctypes.data_as(C.c_void_p), C.c_int64(n), C.c_int32(nb))
    o = np.argsort(k, kind="stable")
    good = rc == 0 and np.array_equal(k2, k[o]) and np.array_equal(v2, v[o])
    print("sort", n, nb, "rc", rc, "OK" if good else "FAIL"); ok &= good
for n in [1, 100, 1024, 1025, 300000, 2_000_001]:
    a = rng.integers(-5, 50, size=n).astype(np.int32); seg = np.sort(rng.integers(0, max(1, n // 7), size=n)).astype(np.int32)
    out = np.zeros(n, dtype=np.int32)
    for mode in (0, 1, 2):
        rc = raw.pga_selftest_scan(a.ctypes.data_as(C.c_void_p), seg.ctypes.data_as(C.c_void_p), out.ctypes.data_as(C.c_void_p), C.c_int64(n), C.c_int32(mode))
        if mode == 0: exp = np.concatenate(([0], np.cumsum(a[:-1].astype(np.int64)))).astype(np.int32)
        elif mode == 1: exp = np.concatenate(([-1], np.maximum.accumulate(np.maximum(a, -1))[:-1])).astype(np.int32)
        else:
            exp = a.copy()
            for i in range(1, n):
                if seg[i] == seg[i - 1] and exp[i - 1] > exp[i]: exp[i] = exp[i - 1]
        good = rc == 0 and np.array_equal(out, exp)
        print("scan", n, mode, "rc", rc, "OK" if good else "FAIL"); ok &= good
# 2. end-to-end parity HIP vs oracle backend (same host driver, same canonical order)
def mk(name, gen):
    d = "/tmp/synth/" + name
    if not os.path.exists(d): synth.write_files(gen, d)
    return sorted(os.path.join(d, f) for f in os.listdir(d))
sets = {"C4": sorted("tests/golden/C4/" + f for f in os.listdir("tests/golden/C4")),
        "bact20": mk("bact20", synth.bact(20, 500, seed=1)), "human8": mk("human8", synth.human(8, 300, iso=3.0, seed=1, n_chr=6)),
        "human8f": mk("human8f", synth.human(8, 300, iso=3.0, seed=2, n_chr=4, frag=True))}
for s in range(4): sets["fuzz%d" % s] = mk("fuzz%d" % s, synth.fuzz(s, harsh=(s % 2 == 0)))
capi_verbose = C.c_int.in_dll(lib, "pg_verbose"); capi_verbose.value = 1
C.c_int.in_dll(ora, "pg_verbose").value = 1
os.makedirs("gpurun_out", exist_ok=True)
for name, files in sets.items():
    for args in (["--bed=raw"], ["--bed=flag"], [], ["-p0", "-a1"], ["-S"], ["-E", "-a2"]):
        try:
            a = capi.run(lib, files, args)
        except Exception as e:
            a = b"ERR " + str(e).encode()
        b = capi.run(ora, files, args)
        good = a == b
        print("parity", name, args, len(a), len(b), "OK" if good else "DIFF"); ok &= good
        if not good:
            tag = name + "_" + "_".join(x.strip("-=") for x in args)
            open("gpurun_out/%s.hip.out" % tag, "wb").write(a); open("gpurun_out/%s.ora.out" % tag, "wb").write(b)
big = mk("bact100", synth.bact(100, 5000, seed=1))
t = time.time(); a = capi.run(lib, big, []); t1 = time.time() - t
print("bact100 hip wall %.3fs path %.3fs hits %d md5 %s" % (t1, lib.pg_last_path_seconds(), lib.pg_last_path_hits(), hashlib.md5(a).hexdigest()))
t = time.time(); a2 = capi.run(lib, big, []); t1 = time.time() - t
print("bact100 hip (2nd) wall %.3fs path %.3fs" % (t1, lib.pg_last_path_seconds()))
t = time.time(); b = capi.run(ora, big, []); t2 = time.time() - t
print("bact100 oracle wall %.3fs path %.3fs md5 %s" % (t2, ora.pg_last_path_seconds(), hashlib.md5(b).hexdigest()), "OK" if a == b else "DIFF")
print("ALL OK" if ok and a == b else "SOME FAILED")
