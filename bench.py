#!/usr/bin/env python3
"""bench.py -- throughput of the pangene graph-construction path (stages A+B+C) on MI355X.

    python bench.py --gpus N --steps K --warmup W
    (N > 1: python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 ... bench.py --gpus N ...)

Workload (BASELINE.json): configs[1] = synthetic bacterial pangenome, 100 genomes x 5000 proteins (~1 M PAF
hits) PER GPU; N GPUs process N x 100 genomes of the same seeded set (weak scaling: genomes shard
embarrassingly, ids are global, every round exchanges a few small integer vectors over RCCL).
A step = one full pass of the hot path over the shard that is already resident in HBM:
pg_post_process (device sort, stage A filters + interval sweeps, stage B) + pg_graph_gen (vertex selection,
17 arc rounds, 15 branch rounds) + the final per-hit state download.  PAF text parsing, the host->HBM
upload and GFA printing are outside the timed region (reported separately in the JSON line).
Rank 0 prints ONE JSON line.
"""
from __future__ import annotations

import argparse
import ctypes as C
import hashlib
import json
import os
import re
import subprocess
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak, /opt/skills/guides/MI355X_MICROARCH.md


def _pmc_traffic(hits_per_launch):
    """HBM bytes per launch of K1 from the PMC passes committed under profiles/ (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, separate
    runs of this very command, FETCH doubled as the gfx950 guide prescribes); null when the recorded passes are for another size."""
    try:
        with open(os.path.join(ROOT, "profiles", "k1_pmc_traffic.json")) as f:
            t = json.load(f)
        return int(t["bytes_per_launch"]) if t.get("hits_per_launch") == hits_per_launch else None
    except Exception:
        return None


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--genomes-per-gpu", type=int, default=100)
    ap.add_argument("--proteins", type=int, default=5000)
    ap.add_argument("--seed", type=int, default=1)
    ap.add_argument("--exact", default="auto", choices=["auto", "all", "off"])
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--keep", action="store_true", help="keep the generated PAF files")
    a = ap.parse_args()

    # RCCL / HIP print banners on fd 1; the contract is ONE JSON line on stdout: park fd 1 on stderr until the end
    real_stdout = os.dup(1)
    os.dup2(2, 1)

    import torch
    import torch.distributed as dist
    from pangene_amd import capi, synth

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != a.gpus and world > 1:
        raise SystemExit("--gpus %d but WORLD_SIZE=%d" % (a.gpus, world))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the product path has no CPU fallback")
    # test hook: PANGENE_BENCH_ONE_GPU=1 lets several ranks share device 0 and exchange over gloo (RCCL refuses two ranks on
    # one device); it exercises this script's multi-rank flow on a 1-GPU box and says nothing about performance
    one_gpu = os.environ.get("PANGENE_BENCH_ONE_GPU") == "1"
    if one_gpu:
        local = 0
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    lib = capi.load()
    C.c_int.in_dll(lib, "pg_verbose").value = 0
    lib.pg_set_exact_mode({"off": 0, "auto": 1, "all": 2}[a.exact])
    keep = None
    exchange_kind = "none (single process)"
    force_x = os.environ.get("PANGENE_FORCE_EXCHANGE") == "1"
    if world > 1 or force_x:
        if world == 1:
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29517")
            os.environ.setdefault("RANK", "0"); os.environ.setdefault("WORLD_SIZE", "1")
        if one_gpu:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=dev)
        from pangene_amd import exchange
        # RCCL called by the library itself, on the kernels' stream; PANGENE_EXCHANGE=torch keeps the collectives in
        # torch.distributed (Python callbacks) instead -- also the fallback when RCCL cannot be bound
        exchange_kind = "rccl-native"
        if one_gpu or os.environ.get("PANGENE_EXCHANGE") == "torch" or not exchange.install_native(lib):
            keep = exchange.install(lib, device=dev)
            exchange_kind = "torch.distributed(nccl) callbacks"

    # ---- synthetic input (not timed): every rank writes its own genomes, then registers the ids of the others
    G = a.genomes_per_gpu * world
    base = os.path.join(tempfile.gettempdir(), "pangene_bench_b%dx%d_s%d" % (G, a.proteins, a.seed))
    os.makedirs(base, exist_ok=True)
    t0 = time.time()
    lo, hi = rank * a.genomes_per_gpu, (rank + 1) * a.genomes_per_gpu
    mine = [os.path.join(base, "g%05d.paf" % j) for j in range(lo, hi)]
    if not all(os.path.exists(p + ".done") for p in mine):
        for name, text in synth.bact(G, a.proteins, seed=a.seed, first=lo, last=hi):
            p = os.path.join(base, name)
            with open(p, "w") as f:
                f.write(text)
            open(p + ".done", "w").close()
    if world > 1:
        dist.barrier()
    files = [os.path.join(base, "g%05d.paf" % j) for j in range(G)]
    t_gen = time.time() - t0

    opt = capi.parse_args(lib, [])
    d = lib.pg_data_init()
    t0 = time.time()
    capi.read_files(lib, opt, d, files, [not (lo <= j < hi) for j in range(G)])  # host threads; ids as in sequential reads
    t_parse = time.time() - t0

    def one_pass(first):
        if not first:
            lib.pg_rerun_resident(d)
        lib.pg_post_process(C.byref(opt), d)
        g = lib.pg_graph_init(d)
        lib.pg_graph_gen(C.byref(opt), g)
        if lib.pg_last_error():
            raise RuntimeError(lib.pg_last_error_str().decode())
        return g

    def sync():
        torch.cuda.synchronize(dev)
        if world > 1:
            dist.barrier()
            torch.cuda.synchronize(dev)

    # first pass: includes the upload (PCIe) -- reported, never part of `value`
    t0 = time.time()
    g = one_pass(True)
    t_first = time.time() - t0
    t_upload = lib.pg_last_upload_seconds()
    n_hits = lib.pg_last_path_hits()
    out = tempfile.mktemp(prefix="pangene_bench_", suffix=".gfa")
    lib.pg_set_output(out.encode())
    lib.pg_write_graph(g)
    lib.pg_write_walk(g)
    lib.pg_set_output(None)
    gfa = open(out, "rb").read()
    os.unlink(out)
    lib.pg_graph_destroy(g)
    for _ in range(max(0, a.warmup - 1)):
        lib.pg_graph_destroy(one_pass(False))
    lib.pg_kernel_timing_reset(d)
    sync()
    t0 = time.time()
    path_sec = 0.0
    phases = None
    for _ in range(a.steps):
        if lib.pg_rerun_resident(d) != 0:
            raise RuntimeError("pg_rerun_resident failed")
        lib.pg_graph_destroy(one_pass(False))
        path_sec += lib.pg_last_path_seconds()
        ph = (C.c_double * 16)()
        nph = lib.pg_phase_times(ph, 16)
        cur = [ph[i] for i in range(nph)]
        phases = cur if phases is None else [x + y for x, y in zip(phases, cur)]
    sync()
    dt = time.time() - t0
    if world > 1:
        xdev = torch.device("cpu") if one_gpu else dev
        t = torch.tensor([dt], dtype=torch.float64, device=xdev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
        h = torch.tensor([n_hits], dtype=torch.int64, device=xdev)
        dist.all_reduce(h)
        tot_hits = int(h.item())
    else:
        tot_hits = n_hits

    # ---- roofline of K1 = the stage-A interval-dominance sweep pg_shadow(cal_dom_sc=1), read.c:248 / overlap.c:101-178
    ms, nl, units = C.c_double(), C.c_int64(), C.c_int64()
    lib.pg_kernel_timing(d, 0, C.byref(ms), C.byref(nl), C.byref(units))
    E = 1.0  # exons per hit of the bacterial shape
    # algorithmic bytes of THIS kernel (the stage-A sweep, not the whole of stage A): SURVEY.md 8(d) gives 56 + 8E B/hit for a
    # pg_shadow sweep (reads cs ce cid pid gid score_adj rank flags n/off_exon + exons, writes flags pid_dom); the
    # cal_dom_sc=1 flavour also reads score_ori and writes score_dom => 64 + 8E
    bytes_per_hit = 64 + 8 * E
    roof = None
    if nl.value:
        avg_ms = ms.value / nl.value
        ach = bytes_per_hit * (units.value / nl.value) / (avg_ms * 1e-3) / 1e9
        roof = {"bound": "hbm", "kernel": "k_sweep<1, false> (pg_shadow cal_dom_sc=1, stage A; the flavour for shards without multi-exon hits)", "achieved": round(ach, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": round(ach / HBM_PEAK_GBS, 4), "traffic": _pmc_traffic(units.value // nl.value), "avg_launch_ms": round(avg_ms, 4), "launches": nl.value,
                "timing": "per-dispatch HIP start/stop events (hipExtLaunchKernelGGL) on the library's stream",
                "algorithmic_bytes_per_hit": bytes_per_hit, "hits_per_launch": units.value // nl.value}

    # ---- CPU baseline: the untouched reference binary on the same PAF files, 1 core (rank 0, N = 1 only)
    cpu = None
    ref = os.path.join(ROOT, "oracle", "_ref", "pangene_ref")
    ref_md5 = None
    if rank == 0 and world == 1 and not a.no_cpu_baseline and os.path.exists(ref):
        t0 = time.time()
        r = subprocess.run([ref] + files, stdout=subprocess.PIPE, stderr=subprocess.PIPE)
        t_ref = time.time() - t0
        ref_md5 = hashlib.md5(r.stdout).hexdigest()
        stamps = re.findall(rb"\[M::pg_read_paf::([0-9.]+)\*", r.stderr)
        done = re.findall(rb"\[M::pg_graph_gen::([0-9.]+)\*[0-9.]+\] round-3", r.stderr)
        t_path = float(done[-1]) if done else t_ref  # read+ingest are interleaved in the reference: count from 0
        cpu = {"value": round(n_hits / t_path / 1e6, 4), "unit": "M hits/s", "cores": 1, "kind": "reference",
               "sample": "whole workload (%d genomes, %d hits kept): reference binary wall until 'round-3 graph' %.2f s incl. its PAF parsing (stage A is interleaved with parsing there); host has %d cores, the reference is single-threaded"
                         % (G, n_hits, t_path, os.cpu_count() or 0),
               "total_wall_s": round(t_ref, 2)}
    if rank == 0:
        res = {
            "metric": "M PAF hits/sec through filter+overlap+graph", "value": round(tot_hits * a.steps / dt / 1e6, 4), "unit": "M hits/s",
            "n_gpus": world, "steps": a.steps, "warmup": a.warmup, "ms_per_step": round(dt / a.steps * 1e3, 3), "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": "int32", "data": "synthetic",
            "config": {"workload": "BASELINE configs[1]: synthetic bacterial pangenome, %d genomes x %d proteins per GPU (%d genomes, %d hits in total), default options"
                                   % (a.genomes_per_gpu, a.proteins, G, tot_hits),
                       "exact_order_mode": a.exact, "parallelism": "genomes sharded over %d GPU(s)" % world, "exchange": exchange_kind},
            "roofline": roof, "cpu_baseline": cpu,
            "gfa_md5": hashlib.md5(gfa).hexdigest() if world == 1 else None,
            # S and L lines are the same on every rank of a sharded run (W lines are per rank): comparable across --gpus N for equal G
            "gfa_sl_md5": hashlib.md5(b"\n".join(l for l in gfa.split(b"\n") if l[:1] in (b"S", b"L"))).hexdigest(),
            "gfa_identical_to_reference": (hashlib.md5(gfa).hexdigest() == ref_md5) if ref_md5 else None,
            "not_timed": {"paf_generate_s": round(t_gen, 2), "paf_parse_s": round(t_parse, 2), "first_pass_incl_upload_s": round(t_first, 3),
                          "pack_and_upload_s": round(t_upload, 3), "path_only_ms_per_step": round(path_sec / a.steps * 1e3, 3)},
            "host_phases_ms_per_step": {lib.pg_phase_name(i).decode(): round(v / a.steps * 1e3, 3) for i, v in enumerate(phases or [])},
        }
        os.write(real_stdout, (json.dumps(res) + "\n").encode())
    lib.pg_data_destroy(d)
    if world > 1 or force_x:
        dist.barrier()
        dist.destroy_process_group()
    del keep


if __name__ == "__main__":
    main()
