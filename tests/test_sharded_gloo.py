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


def _worker(rank, world, port, files, variant, q):
    import ctypes as C
    import torch.distributed as dist
    sys.path.insert(0, ROOT)
    from pangene_amd import capi, exchange
    dist.init_process_group("gloo", init_method="tcp://127.0.0.1:%d" % port, rank=rank, world_size=world)
    lib = capi.load(oracle_host=True)
    C.c_int.in_dll(lib, "pg_verbose").value = 0
    keep = exchange.install(lib)
    n = len(files)
    lo, hi = n * rank // world, n * (rank + 1) // world
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


@pytest.mark.parametrize("name,variant", [("bact20", ""), ("human8", ""), ("C4", ""), ("human8f", "-p0 -a1"), ("fuzz2", "-F")])
def test_two_ranks_equal_single_process(built, expected, name, variant):
    files = golden_files(name)
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, files, variant.split(), q)) for r in range(2)]
    for p in procs:
        p.start()
    res = dict(q.get(timeout=300) for _ in procs)
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    import gzip
    gold = expected[name][variant]
    sl = [b"\n".join(l for l in res[r].split(b"\n") if l[:1] in (b"S", b"L")) for r in (0, 1)]
    assert sl[0] == sl[1]
    w = b"\n".join(l for r in (0, 1) for l in res[r].split(b"\n") if l[:1] == b"W")
    whole = sl[0] + b"\n" + w + b"\n"
    assert hashlib.md5(whole).hexdigest() == gold["md5"], "sharded output differs from the reference's single-process GFA"
