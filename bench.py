#!/usr/bin/env python3
"""bench.py -- throughput of the pangene graph-construction path (stages A+B+C) on MI355X.

    python bench.py --gpus N --steps K --warmup W
    (N > 1: python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 ... bench.py --gpus N ...)

Workload (BASELINE.json): configs[1] = synthetic bacterial pangenome, 100 genomes x 5000 proteins (~1 M PAF
hits) PER GPU; N GPUs process N x 100 genomes of the same seeded set (weak scaling: genomes shard
embarrassingly, ids are global, every round exchanges a few small integer vectors over RCCL).
`--scaling strong --genomes G` fixes the total instead (G / N genomes per rank).  `--workload human47` runs the
configs[2] stand-in (47 human-shaped haplotypes x 20 k multi-exon genes; real HPRC PAFs are not available offline).

A step = one full pass of the hot path over the shard that is already resident in HBM:
pg_post_process (device sort, stage A filters + interval sweeps, stage B) + pg_graph_gen (vertex selection,
17 arc rounds, 15 branch rounds) + the final per-hit state download.  `value` is that resident rate.
`cold_pass` (upload included) is the one pass a `pangene *.paf` invocation makes on a data set the
process has not seen: block packing (reader threads) + allocation + H2D + stages A+B+C; PAF text parsing and GFA
printing are reported separately.  A tiny data set is run first so that kernel code objects are loaded.

The roofline leg (N = 1 only) times K1 = the stage-A interval-dominance sweep on a shard that does not fit the 256 MiB
Infinity Cache (default 1250 x 5 k = the per-GPU shard of configs[3], ~12 M hits; SURVEY 8d asks for >= 10 M hits), and all
of stage A next to it.  Rank 0 prints ONE JSON line.
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
        for ent in (t if isinstance(t, list) else [t]):
            if ent.get("hits_per_launch") == hits_per_launch:
                return int(ent["bytes_per_launch"])
    except Exception:
        pass
    return None


def _gen(synth, kind, base, lo, hi, G, proteins, seed):
    """PAF files of genomes [lo, hi) of the seeded set (cached under the temp dir between runs on one box)"""
    os.makedirs(base, exist_ok=True)
    mine = [os.path.join(base, "g%05d.paf" % j) for j in range(lo, hi)]
    if all(os.path.exists(p + ".done") for p in mine):
        return
    gen = synth.bact(G, proteins, seed=seed, first=lo, last=hi) if kind == "bact" else synth.human(G, proteins, iso=1.0, seed=seed, first=lo, last=hi, frag=True)
    for k, (name, text) in enumerate(gen):
        p = mine[k]
        with open(p, "w") as f:
            f.write(text)
        open(p + ".done", "w").close()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--workload", default="bact", choices=["bact", "human47"])
    ap.add_argument("--genomes-per-gpu", type=int, default=0, help="default 100 (bact) / 47 (human47)")
    ap.add_argument("--genomes", type=int, default=0, help="--scaling strong: genomes in total")
    ap.add_argument("--scaling", default="weak", choices=["weak", "strong"])
    ap.add_argument("--proteins", type=int, default=0, help="default 5000 (bact) / 20000 genes (human47)")
    ap.add_argument("--seed", type=int, default=1)
    ap.add_argument("--exact", default="auto", choices=["auto", "all", "off"])
    ap.add_argument("--roofline-genomes", type=int, default=1250, help="size of the past-L3 shard of the roofline leg (0 = use the main workload)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    a = ap.parse_args()
    kind = "bact" if a.workload == "bact" else "human"
    if a.proteins == 0:
        a.proteins = 5000 if kind == "bact" else 20000
    if a.genomes_per_gpu == 0:
        a.genomes_per_gpu = 100 if kind == "bact" else 47

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

    def sync():
        torch.cuda.synchronize(dev)
        if world > 1:
            dist.barrier()
            torch.cuda.synchronize(dev)

    opt = capi.parse_args(lib, [])

    def one_pass(d, first):
        if not first and lib.pg_rerun_resident(d) != 0:
            raise RuntimeError("pg_rerun_resident failed")
        lib.pg_post_process(C.byref(opt), d)
        g = lib.pg_graph_init(d)
        lib.pg_graph_gen(C.byref(opt), g)
        if lib.pg_last_error():
            raise RuntimeError(lib.pg_last_error_str().decode())
        return g

    def gfa_of(g):
        out = tempfile.mktemp(prefix="pangene_bench_", suffix=".gfa")
        lib.pg_set_output(out.encode())
        lib.pg_write_graph(g)
        lib.pg_write_walk(g)
        lib.pg_set_output(None)
        b = open(out, "rb").read()
        os.unlink(out)
        return b

    def sl_md5(gfa):  # S and L lines are the same on every rank of a sharded run (W lines are per rank)
        return hashlib.md5(b"\n".join(l for l in gfa.split(b"\n") if l[:1] in (b"S", b"L"))).hexdigest()

    def k_timing(d, which):
        ms, nl, units = C.c_double(), C.c_int64(), C.c_int64()
        lib.pg_kernel_timing(d, which, C.byref(ms), C.byref(nl), C.byref(units))
        return ms.value, nl.value, units.value

    # ---- process warm-up (not timed): a tiny data set through the whole path, so that every kernel's code object is loaded
    with tempfile.TemporaryDirectory(prefix="pangene_bench_warm_") as td:
        wf = synth.write_files(synth.bact(8, 300, seed=7), td)
        dw = lib.pg_data_init()
        capi.read_files(lib, opt, dw, wf, [k % world != rank for k in range(len(wf))])
        lib.pg_graph_destroy(one_pass(dw, True))
        lib.pg_data_destroy(dw)

    # ---- synthetic input (not timed): every rank writes its own genomes, then registers the ids of the others
    if a.scaling == "strong":
        G = a.genomes or a.genomes_per_gpu * 8
        lo, hi = G * rank // world, G * (rank + 1) // world
    else:
        G = a.genomes_per_gpu * world
        lo, hi = rank * a.genomes_per_gpu, (rank + 1) * a.genomes_per_gpu
    base = os.path.join(tempfile.gettempdir(), "pangene_bench_%s%dx%d_s%d" % (kind[0], G, a.proteins, a.seed))
    t0 = time.time()
    _gen(synth, kind, base, lo, hi, G, a.proteins, a.seed)
    if world > 1:
        dist.barrier()
    files = [os.path.join(base, "g%05d.paf" % j) for j in range(G)]
    t_gen = time.time() - t0

    d = lib.pg_data_init()
    t0 = time.time()
    capi.read_files(lib, opt, d, files, [not (lo <= j < hi) for j in range(G)])  # host threads; ids as in sequential reads; packs the blocks
    t_parse = time.time() - t0

    # ---- cold pass: the first pass over a data set this process has not seen (allocation + H2D + A + B + C)
    sync()
    t0 = time.time()
    g = one_pass(d, True)
    sync()
    t_cold = time.time() - t0
    t_upload, t_pack = lib.pg_last_upload_seconds(), lib.pg_last_pack_seconds()
    n_hits = lib.pg_last_path_hits()
    nh, ne = C.c_int64(), C.c_int64()
    lib.pg_shard_counts(d, C.byref(nh), C.byref(ne))
    gfa = gfa_of(g)
    lib.pg_graph_destroy(g)
    for _ in range(max(0, a.warmup)):
        lib.pg_graph_destroy(one_pass(d, False))
    lib.pg_kernel_timing_reset(d)
    sync()
    t0 = time.time()
    path_sec = 0.0
    phases = None
    for _ in range(a.steps):
        lib.pg_graph_destroy(one_pass(d, False))
        path_sec += lib.pg_last_path_seconds()
        ph = (C.c_double * 16)()
        nph = lib.pg_phase_times(ph, 16)
        cur = [ph[i] for i in range(nph)]
        phases = cur if phases is None else [x + y for x, y in zip(phases, cur)]
    sync()
    dt = time.time() - t0
    if world > 1:
        xdev = torch.device("cpu") if one_gpu else dev
        t = torch.tensor([dt, t_cold + t_pack], dtype=torch.float64, device=xdev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt, t_cold_all = float(t[0].item()), float(t[1].item())
        h = torch.tensor([n_hits], dtype=torch.int64, device=xdev)
        dist.all_reduce(h)
        tot_hits = int(h.item())
    else:
        tot_hits, t_cold_all = n_hits, t_cold + t_pack

    # ---- roofline of K1 = the stage-A interval-dominance sweep pg_shadow(cal_dom_sc=1), read.c:248 / overlap.c:101-178.
    # Algorithmic bytes of THIS kernel: SURVEY 8(d) gives 56 + 8E B/hit for a pg_shadow sweep (reads cs ce cid pid gid score_adj rank
    # flags n/off_exon + exons, writes flags pid_dom); the cal_dom_sc=1 flavour also reads score_ori and writes score_dom => 64 + 8E.
    # Stage A as a whole (SURVEY 8d "K1 = ingest stage A"): 72 + 8E B/hit, timed from the first kernel of pga_begin to the last of
    # pga_ingest (sorts, per-hit constants, pg_flag_pseudo, both sweeps, isoform / chain / sub-optimal filters).
    def roofline_of(dd, hits, exons, note):
        E = exons / max(1, hits)
        ms, nl, units = k_timing(dd, 0)
        if not nl:
            return None
        multi = exons != hits
        bph = 64 + 8 * E
        avg_ms = ms / nl
        ach = bph * (units / nl) / (avg_ms * 1e-3) / 1e9
        r = {"bound": "hbm", "kernel": "k_sweep<1, %s> (pg_shadow cal_dom_sc=1, stage A)" % ("true" if multi else "false"), "achieved": round(ach, 1), "peak": HBM_PEAK_GBS,
             "unit": "GB/s", "frac": round(ach / HBM_PEAK_GBS, 4), "traffic": _pmc_traffic(units // nl), "avg_launch_ms": round(avg_ms, 4), "launches": nl,
             "timing": "per-dispatch HIP start/stop events (hipExtLaunchKernelGGL) on the library's stream",
             "algorithmic_bytes_per_hit": round(bph, 1), "hits_per_launch": units // nl, "shard": note}
        ms3, nl3, u3 = k_timing(dd, 3)
        if nl3:
            b3 = 72 + 8 * E
            a3 = b3 * (u3 / nl3) / (ms3 / nl3 * 1e-3) / 1e9
            r["stage_a"] = {"what": "all of stage A (pga_begin + pga_ingest: sorts, constants, pg_flag_pseudo, sweeps, filters), SURVEY 8(d) K1 as defined there",
                            "ms": round(ms3 / nl3, 4), "algorithmic_bytes_per_hit": round(b3, 1), "achieved": round(a3, 1), "frac": round(a3 / HBM_PEAK_GBS, 4)}
        return r

    roof = roofline_of(d, nh.value, ne.value, "the bench workload itself (fits the 256 MiB Infinity Cache: an L3 figure)")
    lib.pg_data_destroy(d)  # one context (and one HIP stream) at a time: the legs below bring their own
    d = None
    big = None
    if rank == 0 and world == 1 and a.roofline_genomes > 0 and kind == "bact":
        RG = a.roofline_genomes
        bbase = os.path.join(tempfile.gettempdir(), "pangene_bench_b%dx%d_s%d" % (RG, a.proteins, a.seed))
        t0 = time.time()
        _gen(synth, "bact", bbase, 0, RG, RG, a.proteins, a.seed)
        bfiles = [os.path.join(bbase, "g%05d.paf" % j) for j in range(RG)]
        tb_gen = time.time() - t0
        db = lib.pg_data_init()
        t0 = time.time()
        capi.read_files(lib, opt, db, bfiles)
        tb_parse = time.time() - t0
        torch.cuda.synchronize(dev)
        t0 = time.time()
        gb = one_pass(db, True)
        torch.cuda.synchronize(dev)
        tb_cold = time.time() - t0
        tb_pack, tb_up = lib.pg_last_pack_seconds(), lib.pg_last_upload_seconds()
        bh, be_ = C.c_int64(), C.c_int64()
        lib.pg_shard_counts(db, C.byref(bh), C.byref(be_))
        bgfa = gfa_of(gb)
        lib.pg_graph_destroy(gb)
        lib.pg_graph_destroy(one_pass(db, False))
        lib.pg_kernel_timing_reset(db)
        torch.cuda.synchronize(dev)
        t0 = time.time()
        nb = 3
        for _ in range(nb):
            lib.pg_graph_destroy(one_pass(db, False))
        torch.cuda.synchronize(dev)
        tb = (time.time() - t0) / nb
        note = "%d genomes x %d proteins, %d hits (configs[3] per-GPU shard): past the Infinity Cache" % (RG, a.proteins, bh.value)
        r2 = roofline_of(db, bh.value, be_.value, note)
        if r2:
            r2["also_at_bench_size"] = {k: roof[k] for k in ("achieved", "frac", "avg_launch_ms", "hits_per_launch", "traffic", "stage_a") if roof and k in roof}
            roof = r2
        big = {"workload": note, "ms_per_step": round(tb * 1e3, 2), "M_hits_per_s": round(bh.value / tb / 1e6, 2),
               "cold_pass_ms": round((tb_cold + tb_pack) * 1e3, 1), "cold_M_hits_per_s": round(bh.value / (tb_cold + tb_pack) / 1e6, 2),
               "pack_ms": round(tb_pack * 1e3, 1), "alloc_upload_ms": round(tb_up * 1e3, 1), "paf_generate_s": round(tb_gen, 1), "paf_parse_s": round(tb_parse, 2),
               "gfa_md5": hashlib.md5(bgfa).hexdigest(), "gfa_sl_md5": sl_md5(bgfa)}
        lib.pg_data_destroy(db)

    # ---- CPU baseline: the untouched reference binary on the same PAF files, 1 core (rank 0, N = 1 only)
    cpu = None
    ref = os.path.join(ROOT, "oracle", "_ref", "pangene_ref")
    ref_md5 = None
    if rank == 0 and world == 1 and not a.no_cpu_baseline and os.path.exists(ref):
        t0 = time.time()
        r = subprocess.run([ref] + files, stdout=subprocess.PIPE, stderr=subprocess.PIPE)
        t_ref = time.time() - t0
        ref_md5 = hashlib.md5(r.stdout).hexdigest()
        done = re.findall(rb"\[M::pg_graph_gen::([0-9.]+)\*[0-9.]+\] round-3", r.stderr)
        t_path = float(done[-1]) if done else t_ref  # read+ingest are interleaved in the reference: count from 0
        cpu = {"value": round(n_hits / t_path / 1e6, 4), "unit": "M hits/s", "cores": 1, "kind": "reference",
               "sample": "whole workload (%d genomes, %d hits kept): reference binary wall until 'round-3 graph' %.2f s incl. its PAF parsing (stage A is interleaved with parsing there); host has %d cores, the reference is single-threaded"
                         % (G, n_hits, t_path, os.cpu_count() or 0),
               "total_wall_s": round(t_ref, 2)}
    if rank == 0:
        cfg = "BASELINE configs[1]: synthetic bacterial pangenome" if kind == "bact" else "BASELINE configs[2] stand-in: synthetic human-shaped haplotypes (multi-exon, fragmented contigs)"
        res = {
            "metric": "M PAF hits/sec through filter+overlap+graph", "value": round(tot_hits * a.steps / dt / 1e6, 4), "unit": "M hits/s",
            "n_gpus": world, "steps": a.steps, "warmup": a.warmup, "ms_per_step": round(dt / a.steps * 1e3, 3), "higher_is_better": True,
            "scaling": a.scaling, "vs_baseline": None, "dtype": "int32", "data": "synthetic",
            "config": {"workload": "%s, %d genomes x %d proteins %s (%d genomes, %d hits in total), default options"
                                   % (cfg, hi - lo if a.scaling == "weak" else G, a.proteins, "per GPU" if a.scaling == "weak" else "in total", G, tot_hits),
                       "exact_order_mode": a.exact, "parallelism": "genomes sharded over %d GPU(s)" % world, "exchange": exchange_kind},
            # SURVEY 8(d)'s metric as defined there (upload included): ONE pass over a data set the process has not seen
            "cold_pass": {"value": round(tot_hits / t_cold_all / 1e6, 3), "unit": "M hits/s", "ms": round(t_cold_all * 1e3, 2),
                          "includes": "block packing in the reader threads (%.1f ms) + allocation, H2D and order-replay set-up (%.1f ms) + stages A+B+C; excludes PAF text parsing and GFA printing; kernels were loaded by a tiny warm-up data set"
                                      % (t_pack * 1e3, t_upload * 1e3)},
            "roofline": roof, "cpu_baseline": cpu, "big_shard": big,
            "gfa_md5": hashlib.md5(gfa).hexdigest() if world == 1 else None,
            "gfa_sl_md5": sl_md5(gfa),
            "gfa_identical_to_reference": (hashlib.md5(gfa).hexdigest() == ref_md5) if ref_md5 else None,
            "not_timed": {"paf_generate_s": round(t_gen, 2), "paf_parse_and_pack_s": round(t_parse, 2), "path_only_ms_per_step": round(path_sec / a.steps * 1e3, 3)},
            "host_phases_ms_per_step": {lib.pg_phase_name(i).decode(): round(v / a.steps * 1e3, 3) for i, v in enumerate(phases or [])},
        }
        os.write(real_stdout, (json.dumps(res) + "\n").encode())
    if d is not None:
        lib.pg_data_destroy(d)
    if world > 1 or force_x:
        dist.barrier()
        dist.destroy_process_group()
    del keep


if __name__ == "__main__":
    main()
