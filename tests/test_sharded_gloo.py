"""world_size-2 run of the sharded path on CPU: genomes split across two processes, the exchange hook
backed by torch.distributed/gloo, the oracle backend doing the per-hit work.  Every rank must print the
same S/L lines as the single-process run, and the W lines of the two ranks together must equal its W
lines."""
import hashlib
import os
import socket
import sys

import pytest
import torch.multiprocessing as mp

from conftest import ROOT, golden_files


def _worker(rank, world, port, files, variant, q, cuts=None, verbose=None):
    import ctypes as C
    import torch.distributed as dist
    sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
    from pangene_amd import capi, exchange
    import oracle_host
    dist.init_process_group("gloo", init_method="tcp://127.0.0.1:%d" % port, rank=rank, world_size=world)
    lib = oracle_host.load()
    C.c_int.in_dll(lib, "pg_verbose").value = verbose[rank] if verbose else 0
    keep = exchange.install(lib)
    n = len(files)
    lo, hi = (cuts[rank], cuts[rank + 1]) if cuts else (n * rank // world, n * (rank + 1) // world)
    out = capi.run(lib, files, variant, scan_only=[not (lo <= k < hi) for k in range(n)])
    q.put((rank, out))
    dist.barrier()
    dist.destroy_process_group()
    del keep


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _sharded(files, variant, world, cuts=None, verbose=None):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, files, variant.split(), q, cuts, verbose)) for r in range(world)]
    for p in procs:
        p.start()
    res = dict(q.get(timeout=300) for _ in procs)
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    return res


@pytest.mark.parametrize("name,variant,world,cuts", [
    ("bact20", "", 2, None), ("human8", "", 2, None), ("C4", "", 2, None), ("human8f", "-p0 -a1", 2, None), ("fuzz2", "-F", 2, None),
    ("bact20", "", 3, None),             # uneven shards
    ("bact20", "", 2, [0, 20, 20]),      # a rank that owns no genome still takes part in every exchange
    ("human8", "", 3, [0, 0, 5, 8]),
    ("wide0", "", 2, None), ("wide3", "-S", 3, None),  # contig coordinates beyond 32 bits: every rank cuts its own contigs into virtual ones
])
def test_ranks_equal_single_process(built, expected, name, variant, world, cuts):
    files = golden_files(name)
    res = _sharded(files, variant, world, cuts)
    gold = expected[name][variant]
    sl = [b"\n".join(l for l in res[r].split(b"\n") if l[:1] in (b"S", b"L")) for r in range(world)]
    assert all(x == sl[0] for x in sl)
    w = b"\n".join(l for r in range(world) for l in res[r].split(b"\n") if l[:1] == b"W")
    whole = sl[0] + b"\n" + w + b"\n"
    assert hashlib.md5(whole).hexdigest() == gold["md5"], "sharded output differs from the reference's single-process GFA"


def test_ranks_with_different_log_levels(built, expected, capfd):
    """A per-rank log level must not select the route (and with it the collectives) of a sharded run: the level the routes follow
    is agreed once per upload (graph_driver.cpp route_v)."""
    files = golden_files("human8f")
    res = _sharded(files, "", 3, None, verbose=[3, 0, 1])
    capfd.readouterr()
    sl = [b"\n".join(l for l in res[r].split(b"\n") if l[:1] in (b"S", b"L")) for r in range(3)]
    assert all(x == sl[0] for x in sl)
    w = b"\n".join(l for r in range(3) for l in res[r].split(b"\n") if l[:1] == b"W")
    assert hashlib.md5(sl[0] + b"\n" + w + b"\n").hexdigest() == expected["human8f"][""]["md5"]
