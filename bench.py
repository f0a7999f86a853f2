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
pg_post_process (stage A: both orders and the per-hit records per genome in LDS, filters + interval sweeps; stage B) +
pg_graph_gen (vertex selection, 17 arc rounds, 15 branch rounds) + the final per-hit state download.

Two rates, both in the line, each named for what it is:
  value / ms_per_step  the RESIDENT rate: the bench contract's definition ("inputs already resident in HBM when the timed region
                       starts"; a PCIe-inclusive figure is never `value`), K timed passes between barriers;
  upload_inclusive     SURVEY 8(d)'s metric as the survey words it ("device upload included"): ONE pass over a data set the
                       process has not seen = block packing (reader threads) + allocation + H2D + order-replay set-up + stages
                       A+B+C; PAF text parsing and GFA printing excluded.  Mean / min / max over FIVE data sets of the workload's
                       shape with different seeds (nothing of them is resident or cached), plus the first such pass of the
                       process (which also pays the driver's hipMalloc of the two arenas: 0.4-40 ms on this pool).
A tiny data set is run first so that kernel code objects are loaded.

Legs of the default run (N = 1 only; each brings its own data set and context, one context at a time):
  roofline / big_shard  K1 = the stage-A interval-dominance sweep, and all of stage A, on a shard past the 256 MiB Infinity
                        Cache (1250 x 5 k = the per-GPU shard of configs[3], ~12 M hits; SURVEY 8d asks for >= 10 M hits)
  human_shard           the same on a human-shaped shard (multi-exon hits, fragmented contigs: the k_sweep<3, true> flavour)
  exchange_overhead     the timed steps once more in a fresh process with every collective of the sharded route issued
                        (world size 1, native RCCL on the kernels' stream): what the exchange plumbing costs by itself
  cli                   `pangene_amd/bin/pangene files > /dev/null` in a fresh process (process start, HIP initialisation,
                        code-object load, parsing, path, GFA text) next to the reference binary's wall time
  cpu_baseline          the untouched reference binary on the same files, 1 core
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


def _sweep_src_sha():
    with open(os.path.join(ROOT, "pangene_amd", "csrc", "hip", "k_sweep.hpp"), "rb") as f:
        return hashlib.sha256(f.read()).hexdigest()[:16]


def _pmc_traffic(hits_per_launch, flavour):
    """HBM bytes per launch of K1 from the PMC passes committed under profiles/ (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, separate
    runs of this very command, FETCH doubled as the gfx950 guide prescribes).  An entry counts only for the shard size, the kernel
    flavour and the very source of the sweep (sha256 of k_sweep.hpp) it was measured with; null otherwise."""
    try:
        with open(os.path.join(ROOT, "profiles", "k1_pmc_traffic.json")) as f:
            t = json.load(f)
        sha = _sweep_src_sha()
        for ent in (t if isinstance(t, list) else [t]):
            if ent.get("hits_per_launch") == hits_per_launch and ent.get("flavour", "false") == flavour and ent.get("k_sweep_sha16") == sha:
                return int(ent["bytes_per_launch"])
    except Exception:
        pass
    return None


def _k2_pmc(hits):
    """HBM bytes per launch of the arc round's kernels on a shard of `hits` hits (profiles/k2_pmc_traffic.json: rocprofv3 --pmc FETCH_SIZE /
    WRITE_SIZE passes, profiles/tools/k2_traffic.py), while k_genes.hpp is the source they were measured with; {} otherwise."""
    try:
        with open(os.path.join(ROOT, "profiles", "k2_pmc_traffic.json")) as f:
            t = json.load(f)
        with open(os.path.join(ROOT, "pangene_amd", "csrc", "hip", "k_genes.hpp"), "rb") as f:
            sha = hashlib.sha256(f.read()).hexdigest()[:16]
        return {e["kernel"]: int(e["bytes_per_launch"]) for e in t if e.get("hits_of_the_shard") == hits and e.get("k_genes_sha16") == sha}
    except Exception:
        return {}


def _gen_range(args):
    kind, base, a, b, G, proteins, seed = args
    from pangene_amd import synth
    gen = synth.bact(G, proteins, seed=seed, first=a, last=b) if kind == "bact" else synth.human(G, proteins, iso=1.0, seed=seed, first=a, last=b, frag=True)
    for k, (name, text) in enumerate(gen):
        p = os.path.join(base, "g%05d.paf" % (a + k))
        with open(p, "w") as f:
            f.write(text)
        open(p + ".done", "w").close()
    return b - a


def _gen(kind, base, lo, hi, G, proteins, seed):
    """PAF files of genomes [lo, hi) of the seeded set (cached under the temp dir between runs on one box).  Every genome is
    seeded on its own, so the files are written by a pool of processes -- forked BEFORE this process touches the GPU."""
    os.makedirs(base, exist_ok=True)
    todo = [j for j in range(lo, hi) if not os.path.exists(os.path.join(base, "g%05d.paf.done" % j))]
    if not todo:
        return
    nproc = max(1, min(len(todo), min(os.cpu_count() or 1, 96)))
    per = max(1, (len(todo) + 4 * nproc - 1) // (4 * nproc))
    jobs = []
    i = 0
    while i < len(todo):  # runs of consecutive genomes
        k = i
        while k + 1 < len(todo) and todo[k + 1] == todo[k] + 1 and k + 1 - i < per:
            k += 1
        jobs.append((kind, base, todo[i], todo[k] + 1, G, proteins, seed))
        i = k + 1
    if nproc == 1:
        for jb in jobs:
            _gen_range(jb)
        return
    import multiprocessing as mp
    with mp.get_context("fork").Pool(nproc) as pool:
        list(pool.imap_unordered(_gen_range, jobs))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--workload", default="bact", choices=["bact", "human47", "config3", "config4"],
                    help="bact = BASELINE configs[1] (default); human47 = configs[2] stand-in; config3 / config4 = configs[3] (10 k x 5 k bacterial, ~100 M hits) / "
                         "configs[4] (200 human-shaped x ~110 k isoforms, -p0 -a1) at their FULL stated size on one GPU: one leg, md5 against the reference's")
    ap.add_argument("--cold-sets", type=int, default=5, help="data sets (different seeds) of the upload-inclusive series")
    ap.add_argument("--genomes-per-gpu", type=int, default=0, help="default 100 (bact) / 47 (human47)")
    ap.add_argument("--genomes", type=int, default=0, help="--scaling strong: genomes in total")
    ap.add_argument("--scaling", default="weak", choices=["weak", "strong"])
    ap.add_argument("--proteins", type=int, default=0, help="default 5000 (bact) / 20000 genes (human47)")
    ap.add_argument("--seed", type=int, default=1)
    ap.add_argument("--exact", default="auto", choices=["auto", "all", "off"])
    ap.add_argument("--roofline-genomes", type=int, default=1250, help="size of the past-L3 bacterial shard of the roofline leg (0 = no such leg)")
    ap.add_argument("--human-genomes", type=int, default=500, help="size of the human-shaped shard (x 20 k genes) of the human leg (0 = no such leg)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extra-legs", action="store_true", help="skip the exchange-overhead and command-line legs")
    ap.add_argument("--leg", default="", help=argparse.SUPPRESS)  # internal: "steps-only" prints {"ms_per_step": ...} for the exchange-overhead leg
    a = ap.parse_args()
    full = {"config3": ("bact", 10000, 5000, 1.0, [], "bact10000x5k", ""), "config4": ("human", 200, 20000, 5.5, ["-p0", "-a1"], "human200x20k_iso5.5", "-p0 -a1")}.get(a.workload)
    if full:  # one leg of its own (full_size_leg): the bench workload in front of it is the default one, with no other leg
        a.no_extra_legs = a.no_cpu_baseline = True
        a.roofline_genomes = a.human_genomes = 0
    kind = "human" if a.workload == "human47" else "bact"
    if a.proteins == 0:
        a.proteins = 5000 if kind == "bact" else 20000
    if a.genomes_per_gpu == 0:
        a.genomes_per_gpu = 100 if kind == "bact" else 47

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    force_x = os.environ.get("PANGENE_FORCE_EXCHANGE") == "1"
    solo = rank == 0 and world == 1 and not force_x and a.leg == ""
    if world != a.gpus and world > 1:
        raise SystemExit("--gpus %d but WORLD_SIZE=%d" % (a.gpus, world))

    # ---- synthetic inputs first (not timed): a pool of forked processes writes them, before this process touches the GPU
    if a.scaling == "strong":
        G = a.genomes or a.genomes_per_gpu * 8
        lo, hi = G * rank // world, G * (rank + 1) // world
    else:
        G = a.genomes_per_gpu * world
        lo, hi = rank * a.genomes_per_gpu, (rank + 1) * a.genomes_per_gpu
    tmp = tempfile.gettempdir()
    base = os.path.join(tmp, "pangene_bench_%s%dx%d_s%d" % (kind[0], G, a.proteins, a.seed))
    t0 = time.time()
    _gen(kind, base, lo, hi, G, a.proteins, a.seed)
    t_gen = time.time() - t0
    files = [os.path.join(base, "g%05d.paf" % j) for j in range(G)]
    legs = {}
    cold_sets = []  # the upload-inclusive series: the workload's shape, other seeds (N = 1 only)
    if solo:
        for k in range(1, max(0, a.cold_sets) + 1):
            bk = os.path.join(tmp, "pangene_bench_%s%dx%d_s%d" % (kind[0], G, a.proteins, a.seed + k))
            _gen(kind, bk, lo, hi, G, a.proteins, a.seed + k)
            cold_sets.append(bk)
    full_files = None
    if full and solo:
        from pangene_amd import synth as _synth
        fdir = os.path.join(tmp, "pangene_bench_full_%s" % full[5])
        t0 = time.time()
        if not os.path.exists(fdir + ".done"):
            kw = dict(G=full[1], seed=1)
            kw.update(dict(P=full[2]) if full[0] == "bact" else dict(Q=full[2], iso=full[3], frag=True))
            _synth.write_files_parallel(full[0], fdir, **kw)
            open(fdir + ".done", "w").close()
        full_files = (sorted(os.path.join(fdir, f) for f in os.listdir(fdir)), time.time() - t0)
    if solo and kind == "bact" and a.roofline_genomes > 0:
        b2 = os.path.join(tmp, "pangene_bench_b%dx%d_s%d" % (a.roofline_genomes, a.proteins, a.seed))
        t0 = time.time()
        _gen("bact", b2, 0, a.roofline_genomes, a.roofline_genomes, a.proteins, a.seed)
        big_cold = []
        for k in (1, 2):  # two more shards of the same shape for the leg's upload-inclusive figure
            bk = os.path.join(tmp, "pangene_bench_b%dx%d_s%d" % (a.roofline_genomes, a.proteins, a.seed + k))
            _gen("bact", bk, 0, a.roofline_genomes, a.roofline_genomes, a.proteins, a.seed + k)
            big_cold.append(bk)
        legs["big"] = (b2, a.roofline_genomes, time.time() - t0, big_cold)
    if solo and kind == "bact" and a.human_genomes > 0:
        b3 = os.path.join(tmp, "pangene_bench_h%dx%d_s%d" % (a.human_genomes, 20000, a.seed))
        t0 = time.time()
        _gen("human", b3, 0, a.human_genomes, a.human_genomes, 20000, a.seed)
        legs["human"] = (b3, a.human_genomes, time.time() - t0, [])

    # RCCL / HIP print banners on fd 1; the contract is ONE JSON line on stdout: park fd 1 on stderr until the end
    real_stdout = os.dup(1)
    os.dup2(2, 1)

    import torch
    import torch.distributed as dist
    from pangene_amd import capi, synth

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
        dist.barrier()  # (every rank's files are written)

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

    def gfa_of(g, timing=None):
        out = tempfile.mktemp(prefix="pangene_bench_", suffix=".gfa")
        t0 = time.time()
        lib.pg_set_output(out.encode())
        lib.pg_write_graph(g)
        lib.pg_write_walk(g)
        lib.pg_set_output(None)
        if timing is not None:
            timing.append(time.time() - t0)
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
    t_write = []
    gfa = gfa_of(g, t_write)
    lib.pg_graph_destroy(g)
    for _ in range(max(0, a.warmup)):
        lib.pg_graph_destroy(one_pass(d, False))
    lib.pg_kernel_timing_reset(d)
    sync()
    n_coll0 = lib.pg_collective_count()
    t0 = time.time()
    path_sec = 0.0
    phases = None
    attempts = 0
    for _ in range(a.steps):
        lib.pg_graph_destroy(one_pass(d, False))
        path_sec += lib.pg_last_path_seconds()
        attempts += lib.pg_last_attempts()
        ph = (C.c_double * 16)()
        nph = lib.pg_phase_times(ph, 16)
        cur = [ph[i] for i in range(nph)]
        phases = cur if phases is None else [x + y for x, y in zip(phases, cur)]
    sync()
    dt = time.time() - t0
    coll_per_step = (lib.pg_collective_count() - n_coll0) / a.steps
    waits_per_step = k_timing(d, 4)[1] / a.steps  # times the host waited for the stream
    if a.leg == "steps-only":  # the exchange-overhead leg of another bench.py: its own line, nothing else
        os.write(real_stdout, (json.dumps({"ms_per_step": round(dt / a.steps * 1e3, 3), "exchange": exchange_kind, "gfa_sl_md5": sl_md5(gfa), "collectives_per_step": coll_per_step, "host_waits_per_step": waits_per_step}) + "\n").encode())
        lib.pg_data_destroy(d)
        if world > 1 or force_x:
            dist.barrier()
            dist.destroy_process_group()
        return
    if world > 1:
        xdev = torch.device("cpu") if one_gpu else dev
        t = torch.tensor([dt, t_cold + t_pack], dtype=torch.float64, device=xdev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt, t_cold_all = float(t[0].item()), float(t[1].item())
        h = torch.tensor([n_hits], dtype=torch.int64, device=xdev)
        dist.all_reduce(h)
        tot_hits = int(h.item())
        # every rank must have built the same graph: the S/L lines' md5 of all ranks against rank 0's
        m = torch.tensor(list(bytes.fromhex(sl_md5(gfa))), dtype=torch.int64, device=xdev)
        lo_, hi_ = m.clone(), m.clone()
        dist.all_reduce(lo_, op=dist.ReduceOp.MIN); dist.all_reduce(hi_, op=dist.ReduceOp.MAX)
        ranks_agree = bool(torch.equal(lo_, hi_))
    else:
        tot_hits, t_cold_all = n_hits, t_cold + t_pack
        ranks_agree = None

    # ---- roofline of K1 = the stage-A interval-dominance sweep: pg_shadow(cal_dom_sc=1), read.c:248 / overlap.c:101-178, which since round 4
    # also does the reset behind it (read.c:249-253) and pg_flt_ov_isoform (overlap.c:58-93) in the same launch (k_sweep<3>).
    # Algorithmic bytes of THIS kernel: SURVEY 8(d) gives 56 + 8E B/hit for a pg_shadow sweep (reads cs ce cid pid gid score_adj rank
    # flags n/off_exon + exons, writes flags pid_dom); the cal_dom_sc=1 flavour also reads score_ori and writes score_dom => 64 + 8E
    # (kept for the fused kernel, which does strictly more: it also writes pid_dom0 and the isoform marks).
    # Stage A as a whole (SURVEY 8d "K1 = ingest stage A"): 72 + 8E B/hit, timed from the first kernel of pga_begin to the last of
    # pga_ingest (both orders + per-hit records, pg_flag_pseudo, both sweeps, isoform / chain / sub-optimal filters).
    def roofline_of(dd, hits, exons, note):
        E = exons / max(1, hits)
        ms, nl, units = k_timing(dd, 0)
        if not nl:
            return None
        multi = exons != hits
        bph = 64 + 8 * E
        avg_ms = ms / nl
        ach = bph * (units / nl) / (avg_ms * 1e-3) / 1e9
        dens, in_lds, tiles = k_timing(dd, 7)  # which build of K1 this upload got (pangene_hip.h): the lists of multi-exon hits in LDS, or left in global memory
        flavour = "false" if not multi else ("lean" if in_lds == 0 else "true")
        kname = "k_sweep_lean<3>" if flavour == "lean" else "k_sweep<3, %s>" % flavour
        r = {"bound": "hbm", "kernel": kname + " (stage A's fused sweep: pg_shadow cal_dom_sc=1 + the reset of read.c:249-253 + pg_flt_ov_isoform)", "achieved": round(ach, 1), "peak": HBM_PEAK_GBS,
             "unit": "GB/s", "frac": round(ach / HBM_PEAK_GBS, 4), "traffic": _pmc_traffic(units // nl, flavour), "avg_launch_ms": round(avg_ms, 4), "launches": nl,
             "timing": "per-dispatch HIP start/stop events (hipExtLaunchKernelGGL) on the library's stream",
             "algorithmic_bytes_per_hit": round(bph, 1), "hits_per_launch": units // nl, "shard": note}
        if multi and tiles:
            r["exon_lists"] = {"staged_in_lds": bool(in_lds), "exons_a_tile_to_stage": round(dens, 1), "tiles_sampled": tiles, "rule": "k_sweep (lists in LDS) from 1024 exons a tile on, k_sweep_lean below"}
        if r["traffic"]:  # what the kernel really moves (PMC passes of the same source), against the same peak: the honest fraction when the algorithmic bytes exceed it
            r["frac_by_counters"] = round(r["traffic"] / (avg_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4)
        ms3, nl3, u3 = k_timing(dd, 3)
        if nl3:
            b3 = 72 + 8 * E
            a3 = b3 * (u3 / nl3) / (ms3 / nl3 * 1e-3) / 1e9
            r["stage_a"] = {"what": "all of stage A (pga_begin + pga_ingest: both orders and the per-hit records, pg_flag_pseudo, sweeps, filters), SURVEY 8(d) K1 as defined there",
                            "ms": round(ms3 / nl3, 4), "algorithmic_bytes_per_hit": round(b3, 1), "achieved": round(a3, 1), "frac": round(a3 / HBM_PEAK_GBS, 4)}
        return r

    def k2_of(dd, hits, exons, gfa_bytes):
        """SURVEY 8(d)'s K2 = one pg_gen_arc round (sweep + walk in cm order + temp arcs + two-level collapse): 80 + 8E + 96w algorithmic
        B/hit, w = the share of hits a walk visits (taken from the W-lines of the graph that was written).  Timed in ONE more pass with
        an event pair around every round (PANGENE_TIME_ROUNDS=1: the events cost queue time, so not in a pass that is itself timed);
        the walk scan -- the time-dominant kernels of every pass -- also on its own.  Rounds that the fixed point of the branch rounds
        lets leave at once are not rounds: only those within a factor 4 of a typical long round (the median of the longer half) count."""
        os.environ["PANGENE_TIME_ROUNDS"] = "1"
        try:
            lib.pg_kernel_timing_reset(dd)
            lib.pg_graph_destroy(one_pass(dd, False))
            ev = {}
            for which in (5, 6):
                n_all = k_timing(dd, which)[1]
                v = [k_timing(dd, which | (i + 1) << 8)[0] for i in range(n_all)]  # (class | (k + 1) << 8: the k-th timed launch alone)
                # (the yardstick is the median of the longer half, not the longest: one round that met a hiccup of the box must not disqualify the others)
                top = sorted(v)[len(v) - 1 - (len(v) // 2) // 2] if v else 0.0
                live = [x for x in v if x * 4 >= top] if top > 0 else []
                ev[which] = (sum(live) / len(live) if live else None, len(live), len(v), (min(live), sorted(live)[len(live) // 2], max(live)) if live else None)
        finally:
            os.environ.pop("PANGENE_TIME_ROUNDS", None)
            lib.pg_kernel_timing_reset(dd)
        if not ev[5][0]:
            return None
        E = exons / max(1, hits)
        n_walk = sum(l.count(b">") + l.count(b"<") for l in gfa_bytes.split(b"\n") if l[:1] == b"W")
        w = n_walk / max(1, hits)
        b2 = 80 + 8 * E + 96 * w
        ach = b2 * hits / (ev[5][0] * 1e-3) / 1e9
        out = {"what": "K2 = one pg_gen_arc round (k_sweep<0> + walk scan with the half-arc output + the gene kernels), SURVEY 8(d)", "ms_per_round": round(ev[5][0], 4), "rounds_timed": ev[5][1], "rounds_queued": ev[5][2],
               "round_ms_min_median_max": [round(x, 4) for x in ev[5][3]], "walkable_share": round(w, 3), "algorithmic_bytes_per_hit": round(b2, 1), "achieved": round(ach, 1), "unit": "GB/s", "frac": round(ach / HBM_PEAK_GBS, 4)}
        if ev[6][0]:
            bw = 48 + 40 * w  # yperm + flag word + the two Y records + the gene-major position in; two 4-byte keys and two 16-byte payloads per walkable hit out
            out["walk_scan"] = {"what": "the walk scan alone (reduce / sums / output step): the time-dominant kernels of a pass", "ms": round(ev[6][0], 4), "algorithmic_bytes_per_hit": round(bw, 1),
                                "achieved": round(bw * hits / (ev[6][0] * 1e-3) / 1e9, 1), "frac": round(bw * hits / (ev[6][0] * 1e-3) / 1e9 / HBM_PEAK_GBS, 4)}
            pmc = _k2_pmc(hits)  # what the counters saw of these kernels (mean over the launches of a pass), over the walk's own time
            if "walk_scan" in pmc:
                out["walk_scan"]["traffic"] = pmc["walk_scan"]
                out["walk_scan"]["frac_by_counters"] = round(pmc["walk_scan"] / (ev[6][0] * 1e-3) / 1e9 / HBM_PEAK_GBS, 4)
            if pmc:
                out["traffic_by_kernel"] = {k: v for k, v in pmc.items()}
                tot = sum(v for k, v in pmc.items() if k in ("walk_scan", "gene_arcs_big", "gene_arcs_wave", "sweep0"))
                out["traffic"] = tot
                out["frac_by_counters"] = round(tot / (ev[5][0] * 1e-3) / 1e9 / HBM_PEAK_GBS, 4)
        return out

    roof = roofline_of(d, nh.value, ne.value, "the bench workload itself (fits the 256 MiB Infinity Cache: an L3 figure)")
    lib.pg_data_destroy(d)  # one context (and one HIP stream) at a time: the legs below bring their own
    d = None

    def cold_series(dirs, n_genomes, argv_opt=None):
        """SURVEY 8(d)'s metric: one pass each over data sets this process has never seen (pack + allocation + H2D + A + B + C)"""
        ms, hits = [], 0
        o = argv_opt or opt
        for dn in dirs:
            fl = [os.path.join(dn, "g%05d.paf" % j) for j in range(n_genomes)]
            dc = lib.pg_data_init()
            capi.read_files(lib, o, dc, fl)
            torch.cuda.synchronize(dev)
            t0 = time.time()
            lib.pg_post_process(C.byref(o), dc)
            gc_ = lib.pg_graph_init(dc)
            lib.pg_graph_gen(C.byref(o), gc_)
            torch.cuda.synchronize(dev)
            tc = time.time() - t0 + lib.pg_last_pack_seconds()
            if lib.pg_last_error():
                raise RuntimeError(lib.pg_last_error_str().decode())
            hits += lib.pg_last_path_hits()
            ms.append(tc * 1e3)
            lib.pg_graph_destroy(gc_)
            lib.pg_data_destroy(dc)
        if not ms:
            return None
        mean = sum(ms) / len(ms)
        return {"value": round(hits / len(ms) / (mean * 1e-3) / 1e6, 3), "unit": "M hits/s", "ms_mean": round(mean, 3), "ms_min": round(min(ms), 3), "ms_max": round(max(ms), 3),
                "spread": round((max(ms) - min(ms)) / mean, 4), "n_data_sets": len(ms), "ms_each": [round(x, 3) for x in ms]}

    upl = cold_series(cold_sets, G) if solo and cold_sets else None

    # the box's host link, as this process sees it (measured AFTER the timed passes: torch's allocations must not sit between the warm-up set and the cold pass)
    link_gbps = None
    try:
        hb = torch.empty(64 << 20, dtype=torch.uint8).pin_memory()
        db_ = torch.empty(64 << 20, dtype=torch.uint8, device=dev)
        db_.copy_(hb, non_blocking=True); torch.cuda.synchronize(dev)
        t0 = time.time()
        db_.copy_(hb, non_blocking=True); torch.cuda.synchronize(dev)
        link_gbps = round((64 << 20) / (time.time() - t0) / 1e9, 1)
        del hb, db_
    except Exception:
        pass


    def shard_leg(dirname, n_genomes, what, n_pass=3, more_dirs=()):
        """a leg on a data set of its own: parse, cold pass, one warm pass, n_pass timed passes; the K1 / stage-A roofline of that shard"""
        bfiles = [os.path.join(dirname, "g%05d.paf" % j) for j in range(n_genomes)]
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
        tw = []
        bgfa = gfa_of(gb, tw)
        lib.pg_graph_destroy(gb)
        lib.pg_graph_destroy(one_pass(db, False))
        lib.pg_kernel_timing_reset(db)
        torch.cuda.synchronize(dev)
        t0 = time.time()
        att = 0
        for _ in range(n_pass):
            lib.pg_graph_destroy(one_pass(db, False))
            att += lib.pg_last_attempts()
        torch.cuda.synchronize(dev)
        tb = (time.time() - t0) / n_pass
        note = "%s, %d hits, %.2f exons per hit: past the Infinity Cache" % (what, bh.value, be_.value / max(1, bh.value))
        r2 = roofline_of(db, bh.value, be_.value, note)
        if r2:
            try:
                r2["k2"] = k2_of(db, bh.value, be_.value, bgfa)
            except Exception as ex:  # (a leg's extra must not take the line down)
                r2["k2"] = {"error": str(ex)}
        info = {"workload": note, "ms_per_step": round(tb * 1e3, 2), "M_hits_per_s": round(bh.value / tb / 1e6, 2),
                "cold_pass_ms": round((tb_cold + tb_pack) * 1e3, 1), "cold_M_hits_per_s": round(bh.value / (tb_cold + tb_pack) / 1e6, 2),
                "pack_ms": round(tb_pack * 1e3, 1), "alloc_upload_ms": round(tb_up * 1e3, 1), "paf_parse_s": round(tb_parse, 3),
                "paf_parse_M_hits_per_s": round(bh.value / tb_parse / 1e6, 1), "gfa_write_s": round(tw[0], 3),
                "attempts_per_step": att / n_pass, "gfa_md5": hashlib.md5(bgfa).hexdigest(), "gfa_sl_md5": sl_md5(bgfa)}
        lib.pg_data_destroy(db)
        if more_dirs:
            info["upload_inclusive"] = cold_series(list(more_dirs), n_genomes)
            if info["upload_inclusive"]:
                info["upload_inclusive"]["first_data_set_ms"] = info["cold_pass_ms"]
        return r2, info

    big = human = None
    if "big" in legs:
        r2, big = shard_leg(legs["big"][0], legs["big"][1], "%d genomes x %d proteins (configs[3] per-GPU shard)" % (legs["big"][1], a.proteins), more_dirs=legs["big"][3])
        big["paf_generate_s"] = round(legs["big"][2], 1)
        if r2:
            r2["also_at_bench_size"] = {k: roof[k] for k in ("achieved", "frac", "avg_launch_ms", "hits_per_launch", "traffic", "stage_a") if roof and k in roof}
            roof = r2
    if "human" in legs:
        r3, human = shard_leg(legs["human"][0], legs["human"][1], "%d human-shaped haplotypes x 20000 multi-exon genes, fragmented contigs (configs[2] / [4] shape)" % legs["human"][1])
        human["paf_generate_s"] = round(legs["human"][2], 1)
        human["roofline"] = r3

    # ---- BASELINE configs[3] / configs[4] at their full stated size on ONE device, against the reference's md5 (tests/golden/expected_large.json)
    full_leg = None
    if full and solo and full_files:
        fl, t_fgen = full_files
        fopt = capi.parse_args(lib, full[4])
        df = lib.pg_data_init()
        t0 = time.time()
        capi.read_files(lib, fopt, df, fl)
        tf_reserve = lib.pg_last_reserve_seconds()  # device memory reserved before the parsers start (pga_reserve): inside the read call, not part of parsing
        tf_parse = time.time() - t0 - tf_reserve
        torch.cuda.synchronize(dev)
        t0 = time.time()
        lib.pg_post_process(C.byref(fopt), df)
        gf = lib.pg_graph_init(df)
        lib.pg_graph_gen(C.byref(fopt), gf)
        torch.cuda.synchronize(dev)
        tf_cold = time.time() - t0 + lib.pg_last_pack_seconds()
        if lib.pg_last_error():
            raise RuntimeError(lib.pg_last_error_str().decode())
        fh, fe = C.c_int64(), C.c_int64()
        lib.pg_shard_counts(df, C.byref(fh), C.byref(fe))
        att0 = lib.pg_last_attempts()
        tw = []
        fgfa = gfa_of(gf, tw)
        lib.pg_graph_destroy(gf)
        lib.pg_kernel_timing_reset(df)
        torch.cuda.synchronize(dev)
        t0 = time.time()
        n_res = 2
        for _ in range(n_res):
            lib.pg_rerun_resident(df)
            lib.pg_post_process(C.byref(fopt), df)
            gf = lib.pg_graph_init(df)
            lib.pg_graph_gen(C.byref(fopt), gf)
            lib.pg_graph_destroy(gf)
        torch.cuda.synchronize(dev)
        tf = (time.time() - t0) / n_res
        want = None
        try:
            with open(os.path.join(ROOT, "tests", "golden", "expected_large.json")) as f:
                want = json.load(f).get(full[5], {}).get(full[6])
        except Exception:
            pass
        full_leg = {"workload": "BASELINE %s at full size on one MI355X: %d genomes, %d hits, %.2f exons per hit, options %r" % (a.workload, len(fl), fh.value, fe.value / max(1, fh.value), " ".join(full[4])),
                    "ms_per_step": round(tf * 1e3, 2), "M_hits_per_s": round(fh.value / tf / 1e6, 2), "upload_inclusive_ms": round(tf_cold * 1e3, 1),
                    "upload_inclusive_M_hits_per_s": round(fh.value / tf_cold / 1e6, 2), "attempts_first_pass": att0, "paf_generate_s": round(t_fgen, 1), "paf_parse_s": round(tf_parse, 2), "device_reserve_s": round(tf_reserve, 2),
                    "upload_inclusive_plus_reserve_ms": round((tf_cold + tf_reserve) * 1e3, 1),
                    "gfa_write_s": round(tw[0], 2), "gfa_bytes": len(fgfa), "gfa_md5": hashlib.md5(fgfa).hexdigest(),
                    "reference_md5": want["md5"] if want else None, "gfa_identical_to_reference": (hashlib.md5(fgfa).hexdigest() == want["md5"]) if want else None,
                    "reference_wall_s": want.get("reference_wall_s") if want else None, "roofline": roofline_of(df, fh.value, fe.value, "the full-size shard")}
        del fgfa
        lib.pg_data_destroy(df)

    # HBM bandwidth of a plain copy kernel in this very process ("calibrate with a copy kernel in the same run")
    peak_meas = None
    if rank == 0:
        try:
            lib.pg_trim_host_cache(0)  # (the cached device blocks of the legs go back first)
            gb = lib.pg_device_copy_gbps(1 << 30, 5)
            peak_meas = round(gb, 1) if gb > 0 else None
        except Exception:
            pass
    for r_ in (roof, human.get("roofline") if human else None, full_leg.get("roofline") if full_leg else None):
        if r_ and peak_meas:
            r_["peak_measured"] = peak_meas
            r_["peak_guide"] = 6290.0  # what the guide (MI355X_MICROARCH.md) measured with a float4 copy: the practical ceiling whatever this box's copy kernel reaches
            fm = r_["achieved"] / max(peak_meas, 1.0)
            r_["frac_of_measured"] = round(fm, 4) if fm <= 1.0 else None  # (a fraction above 1 only says that the algorithmic bytes exceed what the kernel moves: see frac_by_counters)
            if "stage_a" in r_:
                r_["stage_a"]["frac_of_measured"] = round(r_["stage_a"]["achieved"] / peak_meas, 4)

    # ---- the exchange plumbing by itself: the same steps in a fresh process, world size 1, every collective of the sharded route issued
    xo = None
    if solo and not a.no_extra_legs:
        try:
            env = dict(os.environ, PANGENE_FORCE_EXCHANGE="1", MASTER_ADDR="127.0.0.1", MASTER_PORT="29519", RANK="0", WORLD_SIZE="1", LOCAL_RANK="0")
            cmd = [sys.executable, os.path.abspath(__file__), "--leg", "steps-only", "--steps", str(a.steps), "--warmup", str(a.warmup), "--workload", a.workload,
                   "--genomes-per-gpu", str(a.genomes_per_gpu), "--proteins", str(a.proteins), "--seed", str(a.seed), "--exact", a.exact]
            r = subprocess.run(cmd, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=600)
            line = [l for l in r.stdout.decode().splitlines() if l.startswith("{")]
            if r.returncode == 0 and line:
                x = json.loads(line[-1])
                plain = dt / a.steps * 1e3
                xo = {"ms_per_step": x["ms_per_step"], "plain_ms_per_step": round(plain, 3), "overhead": round(x["ms_per_step"] / plain - 1.0, 4), "exchange": x["exchange"],
                      "same_graph": x["gfa_sl_md5"] == sl_md5(gfa), "collectives_per_step": x.get("collectives_per_step"), "host_waits_per_step": x.get("host_waits_per_step"),
                      "what": "PANGENE_FORCE_EXCHANGE=1, world size 1: the sharded route with every collective issued (identities), in a fresh process"}
            else:
                xo = {"error": (r.stderr.decode()[-300:] or "no output")}
        except Exception as ex:  # the leg is informative: never fail the bench for it
            xo = {"error": str(ex)[:300]}

    # ---- the whole command in a fresh process, next to the reference's
    cli = None
    exe = os.path.join(ROOT, "pangene_amd", "bin", "pangene")
    if solo and not a.no_extra_legs and os.path.exists(exe):
        try:
            walls = []
            for _ in range(3):  # a fresh process each time: device start-up alone varies between 0.08 and 0.25 s on this box (profiles/r04_cli_probe.txt)
                t0 = time.time()
                r1 = subprocess.run([exe] + files, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=600)
                walls.append(time.time() - t0)
                if walls[-1] == min(walls):
                    r = r1
            t_cli = min(walls)
            rp = subprocess.run([exe] + files, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=600, env=dict(os.environ, PANGENE_TIMING="1"))  # one more, not timed: the wall time by parts
            cli = {"wall_s": round(t_cli, 3), "wall_s_each": [round(w, 3) for w in walls], "wall_s_is": "the fastest of three fresh processes", "M_hits_per_s": round(n_hits / t_cli / 1e6, 2), "rc": r.returncode, "gfa_md5": hashlib.md5(r.stdout).hexdigest(),
                   "same_bytes_as_the_library_run": hashlib.md5(r.stdout).hexdigest() == hashlib.md5(gfa).hexdigest(),
                   "what": "pangene_amd/bin/pangene <%d files> > pipe: process start + HIP initialisation + code-object load + parsing + path + GFA text" % len(files)}
            m = re.search(rb"\[cli_timing\] (\{.*\})", rp.stderr)
            if m:
                cli["parts"] = json.loads(m.group(1).decode())
        except Exception as ex:
            cli = {"error": str(ex)[:300]}

    # ---- CPU baseline: the untouched reference binary on the same PAF files, 1 core (rank 0, N = 1 only)
    cpu = None
    ref = os.path.join(ROOT, "oracle", "_ref", "pangene_ref")
    ref_md5 = None
    if solo and not a.no_cpu_baseline and os.path.exists(ref):
        t0 = time.time()
        r = subprocess.run([ref] + files, stdout=subprocess.PIPE, stderr=subprocess.PIPE)
        t_ref = time.time() - t0
        ref_md5 = hashlib.md5(r.stdout).hexdigest()
        done = re.findall(rb"\[M::pg_graph_gen::([0-9.]+)\*[0-9.]+\] round-3", r.stderr)
        t_path = float(done[-1]) if done else t_ref  # read+ingest are interleaved in the reference: count from 0
        cpu = {"value": round(n_hits / t_path / 1e6, 4), "unit": "M hits/s", "cores": 1, "kind": "reference",
               "sample": "whole workload (%d genomes, %d hits kept): reference binary wall until 'round-3 graph' %.2f s incl. its PAF parsing (stage A is interleaved with parsing there); host shows %d hardware threads and grants this process %d cores of CPU time, the reference is single-threaded"
                         % (G, n_hits, t_path, os.cpu_count() or 0, synth._cpu_budget()),
               "total_wall_s": round(t_ref, 2)}
        if cli and "wall_s" in cli:
            cli["reference_wall_s"] = round(t_ref, 2)
    if rank == 0:
        cfg = "BASELINE configs[1]: synthetic bacterial pangenome" if kind == "bact" else "BASELINE configs[2] stand-in: synthetic human-shaped haplotypes (multi-exon, fragmented contigs)"
        res = {
            "metric": "M PAF hits/sec through filter+overlap+graph", "value": round(tot_hits * a.steps / dt / 1e6, 4), "unit": "M hits/s",
            "n_gpus": world, "steps": a.steps, "warmup": a.warmup, "ms_per_step": round(dt / a.steps * 1e3, 3), "higher_is_better": True,
            "scaling": a.scaling, "vs_baseline": None, "dtype": "int32", "data": "synthetic",
            "config": {"workload": "%s, %d genomes x %d proteins %s (%d genomes, %d hits in total), default options"
                                   % (cfg, hi - lo if a.scaling == "weak" else G, a.proteins, "per GPU" if a.scaling == "weak" else "in total", G, tot_hits),
                       "exact_order_mode": a.exact, "parallelism": "genomes sharded over %d GPU(s)" % world, "exchange": exchange_kind,
                       "host_wait": "hipStreamSynchronize" if os.environ.get("PANGENE_WAIT") == "sync" else "hipStreamQuery polled for up to 200 us, then hipStreamSynchronize"},
            "value_definition": "resident rate (the bench contract: inputs resident in HBM when the timed region starts); SURVEY 8(d)'s upload-inclusive metric is `upload_inclusive`",
            "attempts_per_step": attempts / max(1, a.steps),
            # SURVEY 8(d)'s metric as the survey words it (device upload included), over data sets the process has never seen
            "upload_inclusive": (dict(upl, first_pass_of_the_process_ms=round(t_cold_all * 1e3, 2),
                                      includes="block packing in the reader threads + allocation, H2D and order-replay set-up + stages A+B+C; excludes PAF text parsing and GFA printing (round 5: and with the printing what the WRITERS fetch of the per-hit state -- one flt bit per hit and the two orders -- which pg_write_walk asks for itself, as the reference sorts inside pg_write_walk: gfa_write_s carries it now); every data set has its own seed: nothing is resident or cached except the device memory blocks the previous data set gave back (the library keeps two)")
                                 if upl else None),
            "full_size": full_leg,
            "cold_pass": {"value": round(tot_hits / t_cold_all / 1e6, 3), "unit": "M hits/s", "ms": round(t_cold_all * 1e3, 2),
                          "includes": "block packing in the reader threads (%.1f ms) + allocation, H2D and order-replay set-up (%.1f ms) + stages A+B+C; excludes PAF text parsing and GFA printing; kernels were loaded by a tiny warm-up data set"
                                      % (t_pack * 1e3, t_upload * 1e3),
                          "upload_MB": round((44 * nh.value + 8 * ne.value) / 1e6, 1), "host_to_device_GBps_of_this_box": link_gbps},
            "roofline": roof, "cpu_baseline": cpu, "big_shard": big, "human_shard": human, "exchange_overhead": xo, "cli": cli,
            "gfa_md5": hashlib.md5(gfa).hexdigest() if world == 1 else None,
            "gfa_sl_md5": sl_md5(gfa), "gfa_sl_lines_same_on_all_ranks": ranks_agree,
            "gfa_identical_to_reference": (hashlib.md5(gfa).hexdigest() == ref_md5) if ref_md5 else None,
            "not_timed": {"paf_generate_s": round(t_gen, 2), "paf_parse_and_pack_s": round(t_parse, 3), "gfa_write_s": round(t_write[0], 4), "path_only_ms_per_step": round(path_sec / a.steps * 1e3, 3)},
            "host_waits_per_step": waits_per_step, "collectives_per_step": coll_per_step,
            "host_phases_ms_per_step": {lib.pg_phase_name(i).decode(): round(v / a.steps * 1e3, 3) for i, v in enumerate(phases or [])},
        }
        os.write(real_stdout, (json.dumps(res) + "\n").encode())
    if world > 1 or force_x:
        dist.barrier()
        dist.destroy_process_group()
    del keep


if __name__ == "__main__":
    main()
